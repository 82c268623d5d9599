"""The oracle against everything the reference's own tests pin for this path
(SURVEY.md section 8c): the 181-row golden learning curve, plus known-answer values."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc-0.txt")


@pytest.mark.parametrize("math", [ob.MATH_LIBM, ob.MATH_PORTABLE])
def test_golden_curve_byte_exact(oracle, math):
    """`grld -s 1 tests/pendulum-sarsa-tc.yaml` (bin/runtests.py:21) -> tests/template/pendulum-sarsa-tc-0.txt"""
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=math), seed=1)
    rows, _ = e.run(2000)
    text = e.format_rows(rows)
    with open(GOLDEN) as f:
        golden = f.read()
    assert len(rows) == 181
    assert text == golden


def test_math_modes_agree_to_rounding(oracle):
    """libm and portable arithmetic differ by <= 1 ulp per call; over a whole run the
    discrete decisions are identical; returns agree to ~4e-9 relative (1-ulp differences
    are amplified by the unstable upright equilibrium), far inside the 1e-5 of the north star."""
    a = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_LIBM), seed=3)
    b = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_PORTABLE), seed=3)
    ra, _ = a.run(330)
    rb, _ = b.run(330)
    assert [r.steps for r in ra] == [r.steps for r in rb]
    np.testing.assert_allclose([r.reward for r in ra], [r.reward for r in rb], rtol=1e-6)
    assert list(a.rng()) == list(b.rng())
    # weights: the north star's 1e-5 relative, over the whole table (untouched slots are identical)
    wa, wb = a.all_weights(), b.all_weights()
    changed = np.nonzero(wa != wb)[0]
    assert changed.size < 20000
    np.testing.assert_allclose(wa[changed], wb[changed], rtol=1e-5, atol=1e-7)


def test_known_answers_seed1(oracle):
    """KATs of SURVEY.md section 8(c), re-derived: first weights, first Q-values, first step."""
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_LIBM, test_interval=-1), seed=1)
    np.testing.assert_array_equal(e.weights([0, 1, 2]),
                                  [0.033849041982652039, 0.27507035559708015, 0.79539919112916024])
    rows, taps = e.run(2, tap_cap=200)
    assert len(taps) == 200
    # tile indices of (obs=[0,0], a=-3): the projection updated by the first step
    expect_idx = [7880414, 393415, 3612154, 7585272, 7714558, 1817205, 6530852, 7084512,
                  3784258, 7288551, 1290213, 7444255, 3032162, 6105701, 5672678, 4342842]
    assert list(taps[0].p_idx[:16]) == expect_idx
    assert taps[0].reward == -57.783642145465336
    assert rows[0].reward == -4390.8915000856532
    assert rows[1].reward == -4452.4274919383552


def test_first_q_values_seed1(oracle):
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_LIBM), seed=1)
    ts = e.spec.projector
    q = []
    for a in (-3.0, 0.0, 3.0):
        idx = ob.tile_project(ts, [[0.0, 0.0, a]])[0]
        w = e.weights(idx)
        s = 0.0
        for v in w:
            s += v
        q.append(s / 16)
    assert q == [0.57214509158813298, 0.48896661473106495, 0.53637563628866447]


def test_stats_match_survey_probe(oracle):
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_LIBM), seed=1)
    e.run(2000)
    st = e.stats()
    assert st.learn_steps == 181900 and st.test_steps == 18100
    assert abs(st.weight_reads / st.learn_steps - 101.3) < 0.05
    assert abs(st.weight_rmws / st.learn_steps - 91.2) < 0.05
    assert abs(st.trace_entries_sum / st.learn_steps - 8.55) < 0.01


def test_q_learning_runs(oracle):
    """Q-learning predictor (advantage.cpp:71-110): parity unpinned by reference tests; sanity only."""
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_PORTABLE, agent=ob.AGENT_Q), seed=1)
    rows, _ = e.run(110)
    assert len(rows) == 10
    assert all(np.isfinite(r.reward) and r.reward < 0 for r in rows)


GOLDEN_PID = os.path.join(os.path.dirname(__file__), "golden", "cart_pole_balancing-pid-0.txt")


@pytest.mark.parametrize("math", [ob.MATH_LIBM, ob.MATH_PORTABLE])
def test_cart_pole_balancing_pid_golden_byte_exact(oracle, math):
    """`grld -s 1 tests/cart_pole_balancing-pid.yaml` (bin/runtests.py:21) -> tests/template/cart_pole_balancing-pid-0.txt:
    the reference's second golden file on this path.  It pins dynamics/cart_pole with end_stop = 1 under
    DynamicalModel::step (5 RK4 sub-steps of 0.01 s), the thread-local start draw of the task (cart_pole.cpp:264-273,
    the first RandGen::get() of the process), the balancing reward and the 200-step timeout."""
    e = ob.Experiment(ob.cart_pole_balancing_pid_spec(math=math), seed=1)
    rows, _ = e.run(10)
    assert len(rows) == 10 and all(r.time == 200 for r in rows)
    with open(GOLDEN_PID) as f:
        assert e.format_rows(rows) == f.read()


def test_cart_pole_pin_resolution(oracle):
    """What the 6-digit golden does and does not resolve: returns move by > 5e-4 (one unit of the
    last printed digit) for a 2 % change of the pole mass -- so the dynamics' constants and the integrator are pinned
    -- while the centrifugal term pole_mass_length * dtheta^2 * sin(theta), the one the cart_pole.cpp:65 quirk touches
    (dtheta = state[1], the ANGLE), stays below 1e-6 of the acceleration along the whole golden trajectory: the golden
    cannot tell theta^2 from thetad^2 there.  The quirk is therefore reproduced from the source text, not pinned."""
    spec = ob.cart_pole_balancing_pid_spec(math=ob.MATH_LIBM, tap_starts=1)
    e = ob.Experiment(spec, seed=1)
    _, taps = e.run(10, tap_cap=3000)
    st = np.array([list(t.state[:5]) for t in taps])
    assert st.shape[0] == 10 * 201
    theta, thetad = st[:, 1], st[:, 3]
    # size of the term under either reading, relative to gravity's g*sin(theta) on the same states
    quirk = 0.05 * theta ** 2 * np.abs(np.sin(theta))
    plain = 0.05 * thetad ** 2 * np.abs(np.sin(theta))
    grav = 9.8 * np.abs(np.sin(theta)) + 1e-300
    assert (quirk / grav).max() < 2e-5 and (plain / grav).max() < 2e-3
