"""Parity of the HIP path (through the C ABI) against the oracle: bit-exact for
indices, RNG streams and -- because both sides follow the same portable
arithmetic specification -- for weights, rewards and returns too."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc-0.txt")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    bad = np.nonzero(bits(a) != bits(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a.flat[bad[0]]!r} vs {b.flat[bad[0]]!r}"


# ---------------------------------------------------------------- math -----
@pytest.mark.parametrize("op,name,lo,hi", [(0, "orc_psin", -200.0, 200.0), (1, "orc_pcos", -200.0, 200.0),
                                           (2, "orc_plog", 1e-300, 1e3)])
def test_device_math_bit_exact(grlx, oracle, op, name, lo, hi):
    rng = np.random.default_rng(op)
    x = np.concatenate([rng.uniform(lo, hi, 200000), rng.uniform(-1e-3, 1e-3, 20000) if op < 2 else rng.uniform(0, 1, 20000) ** 8,
                        np.array([0.0, -0.0, 1e-30, 0.5, 1.0, np.pi, -np.pi, 2 * np.pi, 1e5, -1e5, 1048575.0])])
    if op == 2:
        x = np.abs(x) + 1e-308
    got = grlx.runner.device_math(op, x)
    f = getattr(oracle, name)
    want = np.array([f(float(v)) for v in x])
    assert_bit_equal(got, want, name)


def test_small_angle_forms_equal_the_general_ones(grlx, oracle):
    """psin_s / pcos_s / psincos_s take a short path when every lane of a wave has |x| within a quarter turn: it
    must give the bits of the general path.  Whole waves of small arguments, mixed waves, boundaries, zeros."""
    rng = np.random.default_rng(12)
    small = rng.uniform(-0.78, 0.78, 64 * 300)
    edge = np.concatenate([np.full(64, 0.0), np.full(64, -0.0), np.full(64, 0.7853981633974483), np.full(64, -0.7853981633974483),
                           np.full(64, 0.7853981633974484), np.full(64, 0.78539816339744828), np.full(64, 1e-300), np.full(64, -5e-324)])
    mixed = rng.uniform(-7, 7, 64 * 200)
    tiny = rng.uniform(-1e-9, 1e-9, 64 * 20)
    x = np.concatenate([small, edge, mixed, tiny, np.nextafter(0.7853981633974483, [0.0, 1.0] * 32)])
    want_s = np.array([oracle.orc_psin(float(v)) for v in x])
    want_c = np.array([oracle.orc_pcos(float(v)) for v in x])
    assert_bit_equal(grlx.runner.device_math(6, x), want_s, "psin_s")
    assert_bit_equal(grlx.runner.device_math(7, x), want_c, "pcos_s")
    assert_bit_equal(grlx.runner.device_math(8, x), want_s + want_c, "psincos_s")


def test_device_fmod_sqrt_exact(grlx):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-500, 500, 100000), rng.uniform(-7, 7, 100000), [0.0, -0.0, 2 * np.pi, -2 * np.pi, 1e15]])
    y = np.full_like(x, 2 * np.pi)
    assert_bit_equal(grlx.runner.device_math(3, x, y), np.fmod(x, y), "fmod")
    y2 = rng.uniform(1e-3, 50, x.size)
    assert_bit_equal(grlx.runner.device_math(3, x, y2), np.fmod(x, y2), "fmod general")
    z = rng.uniform(0, 1e6, 100000)
    assert_bit_equal(grlx.runner.device_math(4, z), np.sqrt(z), "sqrt")


def test_div6_equals_ieee_division(grlx):
    """RK4's (k1+2k2+2k3+k4)/6 (modeled.cpp:272) uses a 3-operation form proven to be correctly
    rounded; check it against the true division on random, structured and extreme inputs."""
    rng = np.random.default_rng(9)
    m = rng.integers(1 << 52, 1 << 53, 400000).astype(np.float64)          # every mantissa pattern class
    e = rng.integers(-300, 300, m.size)
    x = np.concatenate([np.ldexp(m, e - 52) * rng.choice([-1.0, 1.0], m.size), rng.uniform(-1e3, 1e3, 200000),
                        np.arange(-3000, 3000, dtype=np.float64), np.arange(1, 4000, dtype=np.float64) * (2.0 ** -60),
                        [0.0, -0.0, 1e-310, -1e-310, 5e-324, 1e308, -1e308, 6.0, 3.0, 1.0 / 3.0]])
    assert_bit_equal(grlx.runner.device_math(5, x), x / 6.0, "x/6")


def test_rand48_jump_on_device(grlx, oracle):
    skip = np.array([0, 1, 2, 1000, 8388607, 8388608, 8388609, 2**33 + 5, 2**47 - 1], dtype=np.uint64)
    for seed in (1, 77, 2**31 - 1):
        got = grlx.runner.rand48_at(seed, skip)
        want = []
        for s in skip:
            g = ob.Rand48()
            oracle.orc_srand48(C.byref(g), seed)
            oracle.orc_rand48_jump(C.byref(g), int(s))
            want.append(oracle.orc_drand48(C.byref(g)))
        assert_bit_equal(got, want, "rand48_at")


# --------------------------------------------------------- tile coding -----
def _tile_specs(grlx):
    out = []
    for (T, mem, res, wrap) in [
        (16, 8388608, [0.31415, 3.1415, 3], [6.283, 0, 0]),                               # pendulum sarsa_tc.yaml
        (16, 8388608, [2.5, 0.157075, 2.5, 1.57075], [0, 6.283, 0, 0]),                   # cart_pole ac_tc.yaml
        (8, 1000003, [0.5, 0.25], [0, 0]),                                                # non power-of-two memory
        (32, 4096, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6], [0, 0, 1.2, 0, 0, 0]),                 # many collisions
        (1, 65536, [1.0], [0]),
    ]:
        g = grlx.capi.TileSpec(); o = ob.TileSpec()
        for s in (g, o):
            s.tilings, s.memory, s.dims = T, mem, len(res)
            for i, (r, w) in enumerate(zip(res, wrap)):
                s.resolution[i] = r; s.wrapping[i] = w
        out.append((g, o))
    return out


def test_project_bit_exact(grlx):
    rng = np.random.default_rng(0)
    for g, o in _tile_specs(grlx):
        x = rng.uniform(-40, 40, (4000, g.dims))
        x[:50] = np.round(x[:50])                      # tile boundaries
        x[50:60] = 0.0
        x[60:70] = -1e-12
        got = grlx.runner.project(g, x)
        want = ob.tile_project(o, x)
        assert got.dtype == np.uint32 and (got == want).all()
    # empty batch
    g, _ = _tile_specs(grlx)[0]
    assert grlx.runner.project(g, np.zeros((0, 3))).shape == (0, 16)


def test_project_known_answer(grlx):
    g, _ = _tile_specs(grlx)[0]
    got = grlx.runner.project(g, [[0.0, 0.0, -3.0]])[0]
    assert list(got) == [7880414, 393415, 3612154, 7585272, 7714558, 1817205, 6530852, 7084512,
                         3784258, 7288551, 1290213, 7444255, 3032162, 6105701, 5672678, 4342842]


# ----------------------------------------------------------- env step ------
def test_env_step_bit_exact(grlx):
    rng = np.random.default_rng(1)
    n = 5000
    state = np.stack([rng.uniform(-30, 30, n), rng.uniform(-40, 40, n), rng.uniform(0, 2.97, n)], axis=1)
    state[0] = [np.pi, 0, 0]
    action = rng.choice([-3.0, 0.0, 3.0], n)
    cfg = grlx.pendulum_sarsa_config(1)
    spec = ob.pendulum_sarsa_spec(math=ob.MATH_PORTABLE)
    for _ in range(3):                                  # chained steps
        gs, gobs, grew, gterm = grlx.runner.env_step(cfg, state, action)
        os_, oobs, orew, oterm = ob.env_step(spec, state, action)
        assert_bit_equal(gs, os_, "state"); assert_bit_equal(gobs, oobs, "obs"); assert_bit_equal(grew, orew, "reward")
        assert (gterm == oterm).all()
        state = gs
    # KAT (libm == portable here): first step of seed 1
    s1, _, r1, _ = grlx.runner.env_step(cfg, [[np.pi, 0, 0]], [-3.0])
    assert list(s1[0]) == [3.1026908939925946, -2.5498788738732121, 0.030000000000000006]
    assert r1[0] == -57.783642145465336


# ------------------------------------------------- representation ops ------
def test_table_ops_match_dense_reference(grlx):
    """read/write/update against a dense numpy restatement of linear.cpp, incl. lazily
    initialised slots, duplicate indices inside a projection and invalid indices."""
    cfg = grlx.pendulum_sarsa_config(2, table_log2_capacity=10)
    r = grlx.Runner(cfg, [11, 12])
    e = [ob.Experiment(ob.pendulum_sarsa_spec(), seed=s) for s in (11, 12)]
    dense = [dict(), dict()]

    def w(rep, slot):
        if slot not in dense[rep]:
            dense[rep][slot] = float(e[rep].weights([slot])[0])
        return dense[rep][slot]

    rng = np.random.default_rng(2)
    INV = 0xFFFFFFFF
    for it in range(30):
        n = 6
        rep = rng.integers(0, 2, n).astype(np.int32)
        idx = rng.integers(0, 200, (n, 16)).astype(np.uint32)       # small range: duplicates + shared slots
        if it % 3 == 0:
            idx[0, 5] = idx[0, 2]
        op = it % 3
        if op == 0:
            got = r.read(rep, idx)
            want = []
            for k in range(n):
                s = 0.0
                for v in idx[k]:
                    s += w(rep[k], int(v))
                want.append(s / 16)
            assert_bit_equal(got, want, "read")
        elif op == 1:
            target = rng.uniform(-5, 5, n)
            r.write(rep, idx, target, 0.2)
            for k in range(n):
                s = 0.0
                for v in idx[k]:
                    s += w(rep[k], int(v))
                d = 0.2 * (target[k] - s / 16)
                for v in idx[k]:
                    dense[rep[k]][int(v)] = w(rep[k], int(v)) + d
        else:
            idx[1, 3] = INV
            delta = rng.uniform(-1, 1, n)
            # rows containing an invalid index are update-only (linear.cpp:207)
            r.update(rep, idx, delta)
            for k in range(n):
                for v in idx[k]:
                    if int(v) != INV:
                        dense[rep[k]][int(v)] = w(rep[k], int(v)) + delta[k]
    for rep in (0, 1):
        slots = np.array(sorted(dense[rep]), dtype=np.uint32)
        assert_bit_equal(r.weights(rep, slots), [dense[rep][int(s)] for s in slots], "final weights")
    assert r.table_load(0) + r.table_load(1) == len(dense[0]) + len(dense[1])
    r.close()


# ---------------------------------------------------------- fused path -----
def _compare_taps(gt, ot, A=3, D=2):
    assert gt.test == ot.test and gt.terminal == ot.terminal
    assert gt.action_index == ot.action_index, (gt.action_index, ot.action_index)
    assert list(gt.p_idx[:16]) == list(ot.p_idx[:16])
    assert_bit_equal(list(gt.obs[:D]), list(ot.obs[:D]), "obs")
    assert_bit_equal([gt.reward, gt.action, gt.delta], [ot.reward, ot.action, ot.delta], "reward/action/delta")
    if ot.terminal != 2:                               # no next action after an absorbing state (td.cpp:76-81)
        assert_bit_equal(list(gt.q[:A]), list(ot.q[:A]), "q")
    assert gt.trace_len == ot.trace_len


@pytest.mark.parametrize("agent", [0, 1, 3])
def test_fused_steps_bit_exact(grlx, agent):
    """Every step of the first 23 trials (learning + 2 test trials) of one replica: tile
    indices, Q-values, chosen actions, rewards, TD errors, trace lengths."""
    seed, trials, cap = 4, 23, 2400
    cfg = grlx.pendulum_sarsa_config(5, tap_replica=2, tap_capacity=cap, agent=agent)
    seeds = [seed + 10, seed + 11, seed, seed + 12, seed + 13]
    r = grlx.Runner(cfg, seeds)
    r.run(12); r.run(11)                                 # two launches: state persists across launches
    r.sync()
    e = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent), seed=seed)
    rows, otaps = e.run(trials, tap_cap=cap)
    gtaps = r.taps()
    assert len(gtaps) == len(otaps) == trials * 100
    for k, (gt, ot) in enumerate(zip(gtaps, otaps)):
        try:
            _compare_taps(gt, ot)
        except AssertionError as ex:
            raise AssertionError(f"step {k}: {ex}")
    t, s, rew = r.rows(2)
    assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows]
    assert_bit_equal(rew, [x.reward for x in rows], "returns")
    assert_bit_equal(r.env_state(2), e.state(), "env state")
    assert list(r.rng(2))[:3] == list(e.rng())[:3]
    r.close()


def test_advantage_learning_bit_exact(grlx):
    """predictor/critic/advantage (advantage.cpp:222-268, kappa = 0.2 as in cfg/pendulum/advantage_tc.yaml):
    every step of one replica (A(s', .), actions, TD errors, trace) plus rows, RNG and weights of all;
    and on the acrobot, whose episodes end in an absorbing state (target without the next-state term)."""
    from tests import configs
    seeds, trials, cap = [14, 15, 16, 17, 18], 23, 2400
    cfg = grlx.pendulum_sarsa_config(len(seeds), tap_replica=3, tap_capacity=cap, agent=4, kappa=0.2)
    r = grlx.Runner(cfg, seeds)
    r.run(12); r.run(11); r.sync()
    assert r.last_kernel() == 3
    rng = np.random.default_rng(2)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(agent=4, kappa=0.2), seed=seed)
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        if k == 3:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) == trials * 100
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    _compare_taps(gt, ot)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
            touched = np.unique(np.array([list(tp.p_idx[:16]) for tp in otaps if not tp.test]).ravel()).astype(np.uint32)
            assert_bit_equal(r.weights(k, touched), e.weights(touched), "touched weights")
        slots = rng.integers(0, 8388608, 1000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots), e.weights(slots), "weights")
    r.close()
    cfg, spec = configs.acrobot(grlx, 6, agent=4, kappa=0.5)
    spec.kappa = 0.5
    seeds = [21, 22, 23, 24, 25, 26]
    r = grlx.Runner(cfg, seeds)
    r.run(44); r.sync()
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, _ = e.run(44)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"acrobot returns of seed {seed}")
        assert_bit_equal(r.env_state(k), e.state(), "env state")
    r.close()
    with pytest.raises(grlx.capi.GrlxError, match="kappa"):
        grlx.Runner(grlx.pendulum_sarsa_config(1, agent=4), [1])


def test_qv_learning_bit_exact(grlx):
    """predictor/critic/qv (qv.cpp:74-108; cfg/pendulum/qv_tc.yaml): Q table read by the policy and written
    without a trace, V table with the trace.  Every step of one replica (both projections, Q(s', .), actions,
    TD errors, trace length), rows / RNG / sampled weights of both tables of all; two launches."""
    from tests import configs
    seeds, trials, cap = [71, 72, 73, 74, 75], 23, 2400
    cfg, spec = configs.pendulum_qv(grlx, len(seeds), tap_replica=1, tap_capacity=cap)
    r = grlx.Runner(cfg, seeds)
    r.run(12); r.run(11); r.sync()
    rng = np.random.default_rng(5)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        assert_bit_equal(r.env_state(k), e.state(), "env state")
        if k == 1:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) == trials * 100
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    assert list(gt.p_idx[16:32]) == list(ot.p_idx[16:32])
                    _compare_taps(gt, ot)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
            tq = np.unique(np.array([list(tp.p_idx[:16]) for tp in otaps if not tp.test]).ravel()).astype(np.uint32)
            tv = np.unique(np.array([list(tp.p_idx[16:32]) for tp in otaps if not tp.test]).ravel()).astype(np.uint32)
            assert_bit_equal(r.weights(k, tq, table=0), e.weights(tq, table=0), "Q weights touched")
            assert_bit_equal(r.weights(k, tv, table=1), e.weights(tv, table=1), "V weights touched")
        slots = rng.integers(0, 8388608, 1000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots, table=0), e.weights(slots, table=0), "Q weights")
        assert_bit_equal(r.weights(k, slots, table=1), e.weights(slots, table=1), "V weights")
    r.close()
    # many replicas, no taps, tiny hash memories (shared slots in both tables)
    seeds = list(range(80, 93))
    cfg, spec = configs.pendulum_qv(grlx, len(seeds))
    cfg.projector.memory = spec.projector.memory = 4096
    cfg.actor_projector.memory = spec.actor_projector.memory = 1024
    r = grlx.Runner(cfg, seeds)
    r.run(33); r.sync()
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, _ = e.run(33)
        assert_bit_equal(r.rows(k)[2], [x.reward for x in rows], f"small-memory returns of seed {seed}")
        assert_bit_equal(r.export_weights(k, table=0), e.all_weights(0), "Q table")
        assert_bit_equal(r.export_weights(k, table=1), e.all_weights(1), "V table")
    r.close()


@pytest.mark.parametrize("agent", [0, 1, 3])
def test_accumulating_trace_bit_exact(grlx, agent):
    """trace/enumerated/accumulating (trace.h:238-263): no ssub, up to 19 entries in which slots repeat, every
    occurrence updated in the reference's order.  Every step of one replica, rows / RNG / weights of all; then a
    tiny hash memory where most slots are shared between tilings (serialised updates) with complete dense tables."""
    seeds, trials, cap = [31, 32, 33, 34, 35], 23, 2400
    cfg = grlx.pendulum_sarsa_config(len(seeds), tap_replica=4, tap_capacity=cap, agent=agent, trace=2)
    r = grlx.Runner(cfg, seeds)
    r.run(12); r.run(11); r.sync()
    rng = np.random.default_rng(8)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent, trace=2), seed=seed)
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        if k == 4:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) == trials * 100
            assert max(t.trace_len for t in otaps) == 19
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    _compare_taps(gt, ot)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
            touched = np.unique(np.array([list(tp.p_idx[:16]) for tp in otaps if not tp.test]).ravel()).astype(np.uint32)
            assert_bit_equal(r.weights(k, touched), e.weights(touched), "touched weights")
        slots = rng.integers(0, 8388608, 1000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots), e.weights(slots), "weights")
    r.close()
    seeds = list(range(40, 51))
    cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent, trace=2)
    cfg.projector.memory = 2048
    r = grlx.Runner(cfg, seeds)
    r.run(33); r.sync()
    for k, seed in enumerate(seeds):
        spec = ob.pendulum_sarsa_spec(agent=agent, trace=2)
        spec.projector.memory = 2048
        e = ob.Experiment(spec, seed=seed)
        rows, _ = e.run(33)
        assert_bit_equal(r.rows(k)[2], [x.reward for x in rows], f"small-memory returns of seed {seed}")
        assert_bit_equal(r.export_weights(k), e.all_weights(), "dense table")
    r.close()


@pytest.mark.parametrize("agent", [0, 1])
def test_specialised_accumulating_trace_equals_generic_and_oracle(grlx, agent):
    """cfg/pendulum/{sarsa,q}_tc.yaml with trace/enumerated/accumulating runs the compile-time instantiation
    (SpecPendulumAcc); force_generic the run-time-parameter one: rows, RNG, weights of both equal the oracle's."""
    seeds, trials = [71, 72, 73, 74, 75, 76], 23
    rng = np.random.default_rng(4)
    slots = rng.integers(0, 8388608, 1500).astype(np.uint32)
    for force, want in ((0, 2), (1, 3)):
        cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent, trace=2, force_generic=force)
        r = grlx.Runner(cfg, seeds)
        r.run(12); r.run(11); r.sync()
        assert r.last_kernel() == want
        for k, seed in enumerate(seeds):
            e = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent, trace=2), seed=seed)
            rows, _ = e.run(trials)
            assert_bit_equal(r.rows(k)[2], [x.reward for x in rows], f"returns of seed {seed} (force_generic {force})")
            assert list(r.rng(k))[:3] == list(e.rng())[:3]
            assert_bit_equal(r.weights(k, slots), e.weights(slots), f"weights of seed {seed} (force_generic {force})")
            e.close()
        r.close()


def test_deployer_accumulating_trace(grlx, tmp_path):
    """grlxd on the golden yaml with trace/enumerated/accumulating instead of the replacing trace: rows = oracle's."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    assert "type: trace/enumerated/replacing" in text
    y = tmp_path / "acc.yaml"
    y.write_text(text.replace("type: trace/enumerated/replacing", "type: trace/enumerated/accumulating").replace("trials: 2000", "trials: 44"))
    res = subprocess.run([grlxd, "-s", "3", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(trace=2), seed=3)
    rows, _ = e.run(44)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)


def test_deployer_qv_learning(grlx, tmp_path):
    """grlxd with predictor/critic/qv (the predictor block of cfg/pendulum/qv_tc.yaml): rows = oracle's, and
    save_every: run writes one .dat per representation (Q and V) equal to the oracle's dense tables."""
    import subprocess
    from grl_amd import _build
    from tests import configs
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    old_block = "      projector: ../../policy/projector\n      representation: ../../policy/representation\n"
    new_block = ("      beta: 0.1\n      q_projector: ../../policy/projector\n      q_representation: ../../policy/representation\n"
                 "      v_projector:\n        type: projector/tile_coding\n        tilings: 16\n        memory: 8388608\n"
                 "        resolution: [ 0.31415, 3.1415 ]\n        wrapping: [ 6.283, 0 ]\n"
                 "      v_representation:\n        type: representation/parameterized/linear\n        init_min: [ 0 ]\n        init_max: [ 1 ]\n"
                 "        memory: ../../v_projector/memory\n        outputs: 1\n        output_min: [ ]\n        output_max: [ ]\n")
    assert old_block in text
    text = text.replace(old_block, new_block).replace("      type: predictor/sarsa\n", "      type: predictor/critic/qv\n")
    y = tmp_path / "qv.yaml"
    y.write_text(text.replace("trials: 2000", "trials: 33").replace("save_every: never", "save_every: run"))
    res = subprocess.run([grlxd, "-s", "8", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    _, spec = configs.pendulum_qv(None, 1)
    e = ob.Experiment(spec, seed=8)
    rows, _ = e.run(33)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)
    q = np.fromfile(tmp_path / "pendulum-sarsa-tc-run0-experiment_agent_policy_representation.dat", dtype="<f8")
    v = np.fromfile(tmp_path / "pendulum-sarsa-tc-run0-experiment_agent_predictor_v_representation.dat", dtype="<f8")
    assert_bit_equal(q, e.all_weights(0), "Q .dat")
    assert_bit_equal(v, e.all_weights(1), "V .dat")


def test_deployer_advantage_learning(grlx, tmp_path):
    """grlxd with predictor/critic/advantage (the predictor block of cfg/pendulum/advantage_tc.yaml): rows = oracle's."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    text = text.replace("      type: predictor/sarsa\n", "      kappa: 0.2\n      discretizer: ../../policy/discretizer\n      type: predictor/critic/advantage\n")
    y = tmp_path / "adv.yaml"
    y.write_text(text.replace("trials: 2000", "trials: 44"))
    res = subprocess.run([grlxd, "-s", "6", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(agent=4, kappa=0.2), seed=6)
    rows, _ = e.run(44)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)


def test_taps_with_trial_starts(grlx):
    """tap_starts: the tapped replica also records the start of every trial (first observation, first
    action; terminal = -1) -- together with the step records these are the rows of the reference's
    transition log (online_learning.cpp:183-184, 205-206).  Q agent and actor-critic."""
    from tests import configs
    cap = 1500
    cfg = grlx.pendulum_sarsa_config(3, tap_replica=1, tap_capacity=cap, tap_starts=1)
    r = grlx.Runner(cfg, [8, 9, 10])
    r.run(5); r.run(7); r.sync()
    e = ob.Experiment(ob.pendulum_sarsa_spec(tap_starts=1), seed=9)
    _, otaps = e.run(12, tap_cap=cap)
    gtaps = r.taps()
    assert len(gtaps) == len(otaps) == 12 * 101
    assert [t.terminal for t in otaps].count(-1) == 12
    for k, (gt, ot) in enumerate(zip(gtaps, otaps)):
        try:
            _compare_taps(gt, ot)
        except AssertionError as ex:
            raise AssertionError(f"record {k}: {ex}")
    r.close()
    cfg, spec = configs.cart_pole_ac(grlx, 2, tap_replica=0, tap_capacity=cap, tap_starts=1)
    spec.tap_starts = 1
    r = grlx.Runner(cfg, [51, 52])
    r.run(6); r.sync()
    e = ob.Experiment(spec, seed=51)
    _, otaps = e.run(6, tap_cap=cap)
    gtaps = r.taps()
    assert len(gtaps) == len(otaps) and [t.terminal for t in otaps].count(-1) == 6
    for k, (gt, ot) in enumerate(zip(gtaps, otaps)):
        try:
            _compare_taps(gt, ot, A=1, D=4)
        except AssertionError as ex:
            raise AssertionError(f"record {k}: {ex}")
    r.close()


def test_fused_many_replicas_vs_oracle(grlx):
    """N ragged (not a multiple of 4) replicas with different seeds, 10 test rows each,
    final weights of every slot the oracle touched."""
    seeds = list(range(1, 8))
    trials = 110
    cfg = grlx.pendulum_sarsa_config(len(seeds))
    r = grlx.Runner(cfg, seeds)
    r.run(trials); r.sync()
    assert r.n_rows() == 10
    rng = np.random.default_rng(3)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=seed)
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert_bit_equal(r.row_times(k), [x.time for x in rows], "episode time (column 4)")
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        slots = rng.integers(0, 8388608, 3000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots), e.weights(slots), "untouched/lazy weights")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
    learn, test = r.step_counts()
    assert learn == 100 * 100 * len(seeds) and test == 10 * 100 * len(seeds)
    r.close()


def test_replica0_reproduces_reference_golden(grlx):
    """Config 2 semantics: replica r is seeded srand48(1+r); replica 0 must print the
    reference's own 181-row golden file (tests/template/pendulum-sarsa-tc-0.txt)."""
    cfg = grlx.pendulum_sarsa_config(64)
    r = grlx.Runner(cfg, np.arange(1, 65))
    for _ in range(20):
        r.run(100)
    r.sync()
    t, s, rew = r.rows(0)
    text = "".join(grlx.runner.format_row(int(a), int(b), float(c)) for a, b, c in zip(t, s, rew))
    assert text == open(GOLDEN).read()
    # every touched slot of replica 0 equals the oracle's dense table
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=1)
    e.run(2000)
    dense = np.ctypeslib.as_array(e.L.orc_weights(e.h, 0), shape=(8388608,))
    allslots = np.arange(8388608, dtype=np.uint32)
    got = r.weights(0, allslots)
    assert_bit_equal(got, dense, "all 8,388,608 weights of replica 0")
    assert 10000 < r.table_load(0) < 30000
    r.close()


def test_table_overflow_is_reported(grlx):
    cfg = grlx.pendulum_sarsa_config(4, table_log2_capacity=8)      # 256 slots: too small on purpose
    r = grlx.Runner(cfg, [1, 2, 3, 4])
    r.run(5)
    with pytest.raises(grlx.capi.GrlxError) as ei:
        r.sync()
    assert ei.value.code == grlx.capi.ERR_TABLE_FULL
    r.close()


def test_curve_stats_device(grlx):
    torch = pytest.importorskip("torch")
    n = 37
    cfg = grlx.pendulum_sarsa_config(n)
    r = grlx.Runner(cfg, np.arange(100, 100 + n))
    r.run(22); r.sync()
    out = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
    r.curve_stats(out.data_ptr(), 0, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rew = np.stack([r.rows(k)[2] for k in range(n)])               # [n, 2]
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), rew.sum(0), rtol=1e-13)
    np.testing.assert_allclose(out[:, 1].cpu().numpy(), (rew ** 2).sum(0), rtol=1e-13)
    assert (out[:, 2].cpu().numpy() == n).all()
    r.close()


def test_deployer_reproduces_golden_file(grlx, tmp_path):
    """The reference's own regression test (bin/runtests.py:21-43): run the deployer on
    tests/pendulum-sarsa-tc.yaml with seed 1 and byte-compare `<output>-0.txt` with the template."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    yaml = os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")
    res = subprocess.run([grlxd, "-s", "1", "-l", "-q", yaml], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == open(GOLDEN).read()
    # clones: replica i is seeded seed+i and writes <output>-0@i.txt (multi.cpp:52-56)
    res = subprocess.run([grlxd, "-s", "1", "-l", "-q", "-r", "3", "-t", "110", yaml], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    first = (tmp_path / "pendulum-sarsa-tc-0@0.txt").read_text()
    assert first == "".join(open(GOLDEN).readlines()[:10])
    assert (tmp_path / "pendulum-sarsa-tc-0@2.txt").read_text() != first


def test_host_layer_steps_the_experiments_objects(grlx, tmp_path):
    """The host layer's per-step objects (grl_amd/csrc/host/objects.h: StepwiseEnvironment / StepwiseAgent of an experiment/online_learning):
    `grlx_ops stepwise` runs the loop of OnlineLearningExperiment::run on the HOST over them -- Environment::start / step and
    Agent::start / step / end forwarded to grlx_env_start / _advance and grlx_agent_start / _step / _end, all replicas per call -- on the
    reference's own yaml: the rows of replica 0 are the golden file's."""
    import subprocess
    from grl_amd import _build
    _build.build_host()
    yaml = os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")
    (tmp_path / "in.txt").write_text("1 3 33\n")
    res = subprocess.run([_build.GRLX_OPS, "stepwise", yaml, "experiment", str(tmp_path / "in.txt")], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    assert res.stdout == "".join(open(GOLDEN).readlines()[:3])


def test_deployer_experiment_multi(grlx, tmp_path):
    """experiment/multi (multi.cpp:36-75): `instances` clones side by side, outputs named <output>-<run>@<i>.txt.
    Here the clones are replicas of one device context, instance i seeded seed + i."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    assert text.startswith("experiment:\n")
    inner = "".join("  " + line + "\n" for line in text.splitlines()[1:])
    for sub in ("environment", "agent"):                 # absolute references move one level down
        inner = inner.replace(f": experiment/{sub}", f": experiment/experiment/{sub}")
    inner = inner.replace("trials: 2000", "trials: 22")
    y = tmp_path / "multi.yaml"
    y.write_text("experiment:\n  type: experiment/multi\n  instances: 3\n  experiment:\n" + inner)
    res = subprocess.run([grlxd, "-s", "11", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    for i in range(3):
        e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=11 + i)
        rows, _ = e.run(22)
        assert (tmp_path / f"pendulum-sarsa-tc-0@{i}.txt").read_text() == e.format_rows(rows)


# ------------------------------------------------------------- acrobot -----
def test_acrobot_env_step_bit_exact(grlx):
    from tests import configs
    cfg, spec = configs.acrobot(grlx, 1)
    rng = np.random.default_rng(11)
    n = 4000
    state = np.stack([np.pi + rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-15, 15, n),
                      rng.uniform(-30, 30, n), rng.uniform(0, 19.9, n)], axis=1)
    state[:200, 2] = rng.uniform(12, 14, 200)           # beyond the velocity limits (acrobot.cpp:68-72)
    state[200:400, 3] = -rng.uniform(28, 30, 200)
    action = rng.choice([-1.0, 0.0, 1.0], n)
    for _ in range(3):
        gs, gobs, grew, gterm = grlx.runner.env_step(cfg, state, action)
        os_, oobs, orew, oterm = ob.env_step(spec, state, action)
        assert_bit_equal(gs, os_, "state"); assert_bit_equal(gobs, oobs, "obs"); assert_bit_equal(grew, orew, "reward")
        assert (gterm == oterm).all() and set(np.unique(gterm)) >= {0, 2}
        state = gs


@pytest.mark.parametrize("agent", [0, 1])
def test_acrobot_fused_bit_exact(grlx, agent):
    """Episodes end by absorbing failure (terminal 2 -> TDAgent::end, td.cpp:76-81) at different
    lengths per replica; every step of one replica and the rows of all are compared."""
    from tests import configs
    seeds = [21, 22, 23, 24, 25, 26]
    trials, cap = 44, 20000
    cfg, spec = configs.acrobot(grlx, len(seeds), agent=agent, tap_replica=3, tap_capacity=cap)
    r = grlx.Runner(cfg, seeds)
    r.run(20); r.run(24); r.sync()
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert_bit_equal(r.row_times(k), [x.time for x in rows], "episode time (column 4)")
        assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        assert_bit_equal(r.env_state(k), e.state(), "env state")
        if k == 3:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) and len(otaps) > 100
            assert any(tp.terminal == 2 for tp in otaps)
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    _compare_taps(gt, ot, D=4)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
    r.close()


# ------------------------------------------------- cart-pole, actor-critic ---
def test_cart_pole_env_step_bit_exact(grlx):
    from tests import configs
    for esp in (0, 1):
        cfg, spec = configs.cart_pole_ac(grlx, 1, end_stop_penalty=esp, action_penalty=esp)
        rng = np.random.default_rng(13 + esp)
        n = 4000
        state = np.stack([rng.uniform(-2.6, 2.6, n), rng.uniform(-20, 20, n), rng.uniform(-8, 8, n),
                          rng.uniform(-15, 15, n), rng.uniform(0, 9.97, n)], axis=1)
        action = rng.uniform(-15, 15, n)
        for _ in range(3):
            gs, gobs, grew, gterm = grlx.runner.env_step(cfg, state, action)
            os_, oobs, orew, oterm = ob.env_step(spec, state, action)
            assert_bit_equal(gs, os_, "state"); assert_bit_equal(gobs, oobs, "obs"); assert_bit_equal(grew, orew, "reward")
            assert (gterm == oterm).all()
            state = gs
        if esp:
            assert 2 in set(np.unique(gterm))


def test_device_log_sqrt_for_box_muller(grlx, oracle):
    """Rand::getNormal (utils.h:120-125) = sqrt(-2 log U1) cos(2 pi U2): log on drand48 values k * 2^-48"""
    rng = np.random.default_rng(17)
    u = np.concatenate([rng.integers(1, 1 << 48, 200000).astype(np.float64) * 2.0 ** -48, [2.0 ** -48, 1 - 2.0 ** -48, 0.5]])
    got = grlx.runner.device_math(2, u)
    want = np.array([oracle.orc_plog(float(v)) for v in u])
    assert_bit_equal(got, want, "plog on uniform draws")
    assert_bit_equal(grlx.runner.device_math(4, -2 * want), np.sqrt(-2 * want), "sqrt")


@pytest.mark.parametrize("over", [dict(), dict(end_stop_penalty=1, ac_update_method=1, ac_step_limit=0.5)])
def test_actor_critic_fused_bit_exact(grlx, over):
    """cfg/cart_pole/ac_tc.yaml semantics: every step of one replica (actor and critic tile indices,
    actor output, noisy action, reward, TD error, trace length -- the critic trace is never cleared),
    rows / RNG / state / sampled weights of both tables of all replicas; four launches."""
    from tests import configs
    seeds = [31, 32, 33, 34, 35]
    trials, cap = 24, 6000
    cfg, spec = configs.cart_pole_ac(grlx, len(seeds), tap_replica=1, tap_capacity=cap, **over)
    r = grlx.Runner(cfg, seeds)
    for n in (5, 7, 1, 11):
        r.run(n)
    r.sync()
    rng = np.random.default_rng(19)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert_bit_equal(r.row_times(k), [x.time for x in rows], "episode time (column 4)")
        assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns seed {seed}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        assert_bit_equal(r.env_state(k), e.state(), "env state")
        if k == 1:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) and len(otaps) > 1000
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    assert list(gt.p_idx[16:32]) == list(ot.p_idx[16:32])
                    _compare_taps(gt, ot, A=1, D=4)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
            touched = np.unique(np.array([list(tp.p_idx[:16]) for tp in otaps if not tp.test]).ravel()).astype(np.uint32)
            touched_a = np.unique(np.array([list(tp.p_idx[16:32]) for tp in otaps if not tp.test]).ravel()).astype(np.uint32)
            assert_bit_equal(r.weights(k, touched, table=0), e.weights(touched, table=0), "critic weights touched")
            assert_bit_equal(r.weights(k, touched_a, table=1), e.weights(touched_a, table=1), "actor weights touched")
        slots = rng.integers(0, 8388608, 2000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots, table=0), e.weights(slots, table=0), "critic weights")
        assert_bit_equal(r.weights(k, slots, table=1), e.weights(slots, table=1), "actor weights")
    r.close()


def test_actor_critic_production_ordering_small_memory(grlx):
    """The actor-critic kernel without taps applies the critic's TD update one pass later (in the shadow of the
    next step's loads).  Tiny hash memories make most slots shared between tilings, so evictions, write-through
    entries and reloads of that ordering run constantly: rows, RNG and both complete dense tables vs the oracle."""
    from tests import configs
    seeds = list(range(201, 208))
    cfg, spec = configs.cart_pole_ac(grlx, len(seeds))
    cfg.projector.memory = spec.projector.memory = 4096
    cfg.actor_projector.memory = spec.actor_projector.memory = 2048
    r = grlx.Runner(cfg, seeds)
    for c in (25, 1, 30, 4):
        r.run(c)
    r.sync()
    assert r.last_kernel() == 1
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, _ = e.run(60)
        assert_bit_equal(r.rows(k)[2], [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        assert_bit_equal(r.export_weights(k, table=0), e.all_weights(0), "critic table")
        assert_bit_equal(r.export_weights(k, table=1), e.all_weights(1), "actor table")
    r.close()


def test_deployer_actor_critic_rows_equal_oracle(grlx, tmp_path):
    """grlxd on the actor-critic yaml: the rows it writes are the oracle's, digit for digit."""
    import subprocess
    from grl_amd import _build
    from tests import configs
    grlxd = _build.build_host()
    yaml = os.path.join(os.path.dirname(__file__), "golden", "cart_pole-ac-tc.yaml")
    res = subprocess.run([grlxd, "-s", "31", "-l", "-q", yaml], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    _, spec = configs.cart_pole_ac(grlx, 1)
    e = ob.Experiment(spec, seed=31)
    rows, _ = e.run(24)
    assert (tmp_path / "cart_pole-ac-tc-0.txt").read_text() == e.format_rows(rows)


# -------------------------------------------------------- compass walker ---
def test_walker_env_step_bit_exact(grlx):
    """model/compass_walker has its own RK4, angle wrapping and a secant search for the heel
    strike (SWModel.cpp); follow real gaits so that strikes and falls occur."""
    from tests import configs
    cfg, spec = configs.compass_walker(grlx, 1)
    rng = np.random.default_rng(23)
    n = 3000
    base = np.array([0.1534, 2.0 * 0.1534, -0.1561, -0.0073])
    state = np.zeros((n, 11))
    state[:, :4] = base * (1 + rng.uniform(-0.2, 0.2, (n, 4)))
    state[:, 6] = -np.sin(state[:, 0])
    state[:, 10] = 100.0
    strikes = falls = 0
    for it in range(12):
        action = rng.choice([-1.2, 0.0, 1.2], n)
        gs, gobs, grew, gterm = grlx.runner.env_step(cfg, state, action)
        os_, oobs, orew, oterm = ob.env_step(spec, state, action)
        assert_bit_equal(gs, os_, f"state it {it}"); assert_bit_equal(gobs, oobs, "obs"); assert_bit_equal(grew, orew, "reward")
        assert (gterm == oterm).all()
        strikes += int((gobs[:, 4] > 0.5).sum()); falls += int((gterm == 2).sum())
        state = gs
    assert strikes > 50 and falls > 100


@pytest.mark.parametrize("agent", [1, 0])
def test_walker_fused_bit_exact(grlx, agent):
    """qlearning_walk.yaml semantics: starts drawn by rejection from the GLOBAL drand48 stream,
    absorbing falls, doubled timeout in test trials."""
    from tests import configs
    seeds = [41, 42, 43, 44, 45]
    trials, cap = 33, 12000
    cfg, spec = configs.compass_walker(grlx, len(seeds), agent=agent, tap_replica=2, tap_capacity=cap)
    r = grlx.Runner(cfg, seeds)
    r.run(13); r.run(20); r.sync()
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert_bit_equal(r.row_times(k), [x.time for x in rows], "episode time (column 4)")
        assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        assert_bit_equal(r.env_state(k), e.state(), "env state")
        if k == 2:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) and len(otaps) > 200
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    _compare_taps(gt, ot, D=5)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
    r.close()


def _bench_two_ranks(extra, timeout=900):
    """bench.py launched as the driver launches it (torch.distributed.run, one process per rank);
    on this one-GPU box both ranks use the same device and gloo stands in for RCCL."""
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_share_the_replica_range(grlx):
    out = _bench_two_ranks(["--steps", "3", "--warmup", "1", "--replicas", "256", "--no-secondary"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["learn_steps"] == out["learn_steps_expected"] == 2 * 256 * 1000 * 3, (out["learn_steps"], out.get("env_server"))
    assert out["config"]["env_steps_per_step"] == 2 * 256 * 1100
    assert out["curve_replicas"] == 512                                   # the all-reduced curve counts both ranks' replicas
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"}


@pytest.mark.parametrize("workload,replicas", [("cart_pole_ac", 64), ("acrobot_q", 64), ("compass_walker_q", 64), ("acrobot_walker", 64), ("acrobot_walker_x2", 64)])
def test_bench_every_rollout_workload_on_two_ranks(grlx, workload, replicas):
    """BASELINE configs[2] and [3] have a multi-rank entry point: contiguous replica ids per rank, one all-reduce of the
    curve statistics, env-steps counted by the devices and summed over the ranks."""
    out = _bench_two_ranks(["--workload", workload, "--steps", "2", "--warmup", "1", "--replicas", str(replicas)])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["workload"] == workload
    assert out["replicas_per_gpu"] == replicas and out["learn_steps"] > 0 and out["test_steps"] >= 0
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_learn_step"}
    if workload.startswith("acrobot_walker"):
        assert [p["graph"] for p in out["parts"]] == ["acrobot_q", "compass_walker_q"]
        assert [p["replicas_per_wave"] for p in out["parts"]] == ([16, 16] if workload.endswith("x2") else [8, 8])
        assert [p["replicas"] for p in out["parts"]] == [replicas // 2, replicas // 2]
        assert all(p["curve_replicas"] == replicas for p in out["parts"])   # 2 ranks x replicas/2 per graph
        assert "both halves" in out["parallelism"]
    else:
        assert out["curve_replicas"] == 2 * replicas


def test_bench_batch_path_on_two_ranks(grlx):
    """BASELINE configs[4]: replicas only -- each rank runs its own independent-seed experiments, the per-batch test
    returns are all-reduced."""
    out = _bench_two_ranks(["--workload", "pendulum_fqi_ann", "--replicas", "2", "--fqi-batch-size", "2000", "--fqi-epochs", "20"])
    assert out["n_gpus"] == 2 and out["unit"] == "sample-epochs/s" and out["value"] > 0
    assert out["curve_replicas"] == 4 and out["roofline"]["bound"] == "valu" and "replicas only" in out["parallelism"]


def test_bench_default_line_with_all_secondaries_on_two_ranks(grlx):
    """What the driver's scaling run launches (no --workload): the headline plus every other configuration, all sharded."""
    out = _bench_two_ranks(["--steps", "2", "--warmup", "1", "--replicas", "128", "--secondary-replicas", "32", "--fqi-replicas", "1",
                            "--fqi-batch-size", "2000", "--fqi-epochs", "20"], timeout=1500)
    assert out["n_gpus"] == 2 and out["metric"].startswith("env-steps/sec")
    names = [s["workload"] for s in out["secondary"]]
    assert names == ["cart_pole_ac", "acrobot_q", "compass_walker_q", "acrobot_walker", "acrobot_walker_x2", "pendulum_fqi_ann"]
    assert all(s["n_gpus"] == 2 and s["value"] > 0 and "roofline" in s for s in out["secondary"])


# ------------------------------------------------ BASELINE.json full sizes ---
def _check_sampled_replicas(grlx, r, spec, seeds, sample, trials):
    for k in sample:
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")


@pytest.mark.parametrize("name,n,trials", [("pendulum", 4096, 44), ("cart_pole_ac", 16384, 22), ("cart_pole_ac", 13312, 22), ("compass_walker", 8192, 22),
                                           ("acrobot", 8192, 33), ("acrobot", 16384, 12), ("compass_walker", 32768, 11)])
def test_full_size_batches(grlx, name, n, trials):
    """The replica counts BASELINE.json quotes (configs[1..3], per-GPU share of configs[3]): replicas are
    independent, so ANY replica of the big batch must equal the scalar oracle run with its seed
    (first, last, and some in between), and no replica may raise a status flag."""
    from tests import configs
    make = {"pendulum": configs.pendulum, "cart_pole_ac": configs.cart_pole_ac, "compass_walker": configs.compass_walker,
            "acrobot": configs.acrobot}[name]
    cfg, spec = make(grlx, n)
    seeds = np.arange(1, n + 1)
    r = grlx.Runner(cfg, seeds)
    if name == "cart_pole_ac":                                 # 16 per SIMD: four sub-batches per wave; 13: twelve slots, rotated (grlx_rollout_ac_wide.h)
        assert r.replicas_per_wave() == (16 if n == 16384 else 12)
    else:                                                      # TD agents: 8 beyond 4 per SIMD; acrobot / walker 16 from 15 per SIMD on; the walker 32 from 30 on (the bench's layout)
        assert r.replicas_per_wave() == {4096: 4, 8192: 8, 16384: 16, 32768: 32}[n]
    r.run(trials // 2); r.run(trials - trials // 2)
    r.sync()                                                   # raises if any replica flagged an error
    learn, test = r.step_counts()
    assert learn > 0 and test > 0
    _check_sampled_replicas(grlx, r, spec, seeds, [0, 1, n // 3, n - 2, n - 1], trials)
    # the device-side curve statistics see every replica exactly once
    torch = pytest.importorskip("torch")
    rows = r.n_rows()
    out = torch.zeros((rows, 3), dtype=torch.float64, device="cuda")
    r.curve_stats(out.data_ptr(), 0, rows, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (out[:, 2].cpu().numpy() == n).all()
    assert np.isfinite(out.cpu().numpy()).all()
    r.close()


def test_wide_seed_sweep(grlx):
    """192 seeds x 66 trials of the headline configuration, every replica against its own scalar
    oracle run: rows, RNG positions, environment state and 400 random weights each.  Rare paths of
    the production ordering (evictions that alias a pending lookup, slots shared between tilings,
    colliding inserts) occur somewhere in a sweep of this width."""
    seeds = np.arange(1000, 1192)
    trials = 66
    cfg = grlx.pendulum_sarsa_config(len(seeds))
    r = grlx.Runner(cfg, seeds)
    r.run(40); r.run(26); r.sync()
    rng = np.random.default_rng(41)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=int(seed))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3], seed
        assert_bit_equal(r.env_state(k), e.state(), f"env state of seed {seed}")
        slots = rng.integers(0, 8388608, 400).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots), e.weights(slots), f"weights of seed {seed}")
        e.close()
    r.close()


def test_dat_policy_export_matches_dense_table(grlx, tmp_path):
    """save_every: run writes grl's raw .dat parameter files (representation.h:201-229): 8,388,608 doubles
    that must equal the oracle's dense table bit for bit -- the file a real grl build could load."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    y = tmp_path / "save.yaml"
    y.write_text(text.replace("save_every: never", "save_every: run").replace("trials: 2000", "trials: 33"))
    res = subprocess.run([grlxd, "-s", "7", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    f = tmp_path / "pendulum-sarsa-tc-run0-experiment_agent_policy_representation.dat"
    got = np.fromfile(f, dtype="<f8")
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=7)
    e.run(33)
    dense = np.ctypeslib.as_array(e.L.orc_weights(e.h, 0), shape=(8388608,))
    assert got.shape == (8388608,)
    assert_bit_equal(got, dense, ".dat parameters")


def test_deployer_load_file_round_trip(grlx, tmp_path):
    """experiment/online_learning:load_file (online_learning.cpp:140-150): the .dat a first run saved is
    loaded by a second run with another seed; its rows equal the oracle's after setParams() of the
    first run's table.  A missing file and a file of the wrong size are warnings, as in the reference."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    a = tmp_path / "a.yaml"
    a.write_text(text.replace("save_every: never", "save_every: run").replace("trials: 2000", "trials: 33")
                 .replace("output: pendulum-sarsa-tc", "output: first"))
    res = subprocess.run([grlxd, "-s", "7", "-q", str(a)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    assert (tmp_path / "first-run0-experiment_agent_policy_representation.dat").exists()
    b = tmp_path / "b.yaml"
    b.write_text(text.replace('load_file: ""', "load_file: first-run$run").replace("trials: 2000", "trials: 22")
                 .replace("output: pendulum-sarsa-tc", "output: second"))
    res = subprocess.run([grlxd, "-s", "9", "-l", "-q", str(b)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    src = ob.Experiment(ob.pendulum_sarsa_spec(), seed=7)
    src.run(33)
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_PORTABLE), seed=9)
    e.set_weights(src.all_weights())
    rows, _ = e.run(22)
    assert (tmp_path / "second-0.txt").read_text() == e.format_rows(rows)
    # missing file: warning, the run goes on with the random initialisation
    c = tmp_path / "c.yaml"
    c.write_text(text.replace('load_file: ""', "load_file: nowhere").replace("trials: 2000", "trials: 11")
                 .replace("output: pendulum-sarsa-tc", "output: third"))
    res = subprocess.run([grlxd, "-s", "9", "-l", "-q", str(c)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Could not open 'nowhere-experiment_agent_policy_representation.dat'" in res.stderr
    fresh = ob.Experiment(ob.pendulum_sarsa_spec(), seed=9)
    rows, _ = fresh.run(11)
    assert (tmp_path / "third-0.txt").read_text() == fresh.format_rows(rows)
    # wrong size: "Configuration mismatch" warning
    (tmp_path / "short-experiment_agent_policy_representation.dat").write_bytes(b"\0" * 800)
    d = tmp_path / "d.yaml"
    d.write_text(text.replace('load_file: ""', "load_file: short").replace("trials: 2000", "trials: 11")
                 .replace("output: pendulum-sarsa-tc", "output: fourth"))
    res = subprocess.run([grlxd, "-s", "9", "-l", "-q", str(d)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "Configuration mismatch" in res.stderr
    assert (tmp_path / "fourth-0.txt").read_text() == fresh.format_rows(rows)


def _expected_csv(otaps, D, style, fields, want_test):
    """The text exporter/csv writes for the taps of one variant (csv.cpp:159-206)."""
    names = ["time", "observation", "action", "reward", "terminal"]
    order = [names.index(f.strip()) for f in fields.split(",")] if fields else list(range(5))
    lines, rows, applied, t = [], [], 0.0, 0.0
    for tp in otaps:
        if tp.terminal == -1:
            t = 0.0
            row = [[t], list(tp.obs[:D]), [tp.action], [0.0], [0.0]]
        else:
            t += 1
            row = [[t], list(tp.obs[:D]), [applied], [tp.reward], [float(tp.terminal)]]
        applied = tp.action
        if bool(tp.test) == want_test:
            rows.append(row)
    if rows and style != "none":
        cols = [f"{names[i]}[{k}]" for i in order for k in range(len(rows[0][i]))]
        if style == "meshup":
            lines.append("COLUMNS:")
            lines += [c + (", " if n + 1 < len(cols) else "") for n, c in enumerate(cols)]
            lines.append("DATA:")
        else:
            lines.append(", ".join(cols))
    for row in rows:
        lines.append(", ".join(f"{v:11.6f}" for i in order for v in row[i]))
    return "".join(l + "\n" for l in lines)


@pytest.mark.parametrize("style,fields,variant", [("line", "", "all"), ("meshup", "time, reward,observation", "learn"), ("none", "", "test")])
def test_deployer_transition_log_csv(grlx, tmp_path, style, fields, variant):
    """experiment/online_learning:exporter (exporter/csv, csv.cpp): <file>-learn-0.csv / <file>-test-0.csv with
    one row per trial start and per step (time, observation, action applied, reward, terminal), in the
    reference's header styles and number format -- against the oracle's own step records."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    exporter = f"  exporter:\n    type: exporter/csv\n    file: translog\n    style: {style}\n    variant: {variant}\n"
    if fields:
        exporter += f"    fields: {fields}\n"
    y = tmp_path / "log.yaml"
    y.write_text(text.replace("trials: 2000", "trials: 23").replace('  load_file: ""\n', exporter + '  load_file: ""\n'))
    res = subprocess.run([grlxd, "-s", "4", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(tap_starts=1), seed=4)
    rows, otaps = e.run(23, tap_cap=3000)
    assert len(otaps) == 23 * 101
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)       # the run itself is unchanged
    for which, want_test in (("learn", False), ("test", True)):
        f = tmp_path / f"translog-{which}-0.csv"
        if variant in ("all", which):
            assert f.read_text() == _expected_csv(otaps, 2, style, fields, want_test), which
        else:
            assert not f.exists()


def test_deployer_compass_walker_rows_equal_oracle(grlx, tmp_path):
    """grlxd on the walker's Q-learning yaml (model/compass_walker + task/compass_walker/walk): rows = oracle's."""
    import subprocess
    from grl_amd import _build
    from tests import configs
    grlxd = _build.build_host()
    yaml = os.path.join(os.path.dirname(__file__), "golden", "compass_walker-q-tc.yaml")
    res = subprocess.run([grlxd, "-s", "17", "-l", "-q", yaml], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    _, spec = configs.compass_walker(grlx, 1)
    e = ob.Experiment(spec, seed=17)
    rows, _ = e.run(33)
    assert (tmp_path / "compass_walker-q-tc-0.txt").read_text() == e.format_rows(rows)


def test_deployer_environment_transition_log(grlx, tmp_path):
    """environment/modeled:exporter (modeled.cpp:67-71, 156-157, 200-203): time, state, observation, action, reward,
    terminal per step, where time is the environment's cumulative learn / test time before the step and state the
    model state the step started from -- against the oracle's records.  Together with the experiment's own log."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    old_env = "    exporter: 0\n    type: environment/modeled\n"
    assert old_env in text
    text = text.replace(old_env, "    exporter:\n      type: exporter/csv\n      file: envlog\n    type: environment/modeled\n")
    text = text.replace('  load_file: ""\n', '  exporter:\n    type: exporter/csv\n    file: explog\n    variant: test\n  load_file: ""\n')
    y = tmp_path / "envlog.yaml"
    y.write_text(text.replace("trials: 2000", "trials: 23"))
    res = subprocess.run([grlxd, "-s", "4", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(tap_starts=1), seed=4)
    rows, otaps = e.run(23, tap_cap=3000)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)
    names = ["time", "state", "observation", "action", "reward", "terminal"]
    width = {"time": 1, "state": 3, "observation": 2, "action": 1, "reward": 1, "terminal": 1}
    header = ", ".join(f"{n}[{k}]" for n in names for k in range(width[n]))
    want = {False: [header], True: [header]}
    clock = {False: 0.0, True: 0.0}
    before, applied, variant = None, 0.0, False
    for tp in otaps:
        if tp.terminal == -1:
            variant = bool(tp.test)
        else:
            vals = [clock[variant]] + before + list(tp.obs[:2]) + [applied, tp.reward, float(tp.terminal)]
            want[variant].append(", ".join(f"{v:11.6f}" for v in vals))
            clock[variant] += 0.03
        before, applied = list(tp.state[:3]), tp.action
    for which, key in (("learn", False), ("test", True)):
        assert (tmp_path / f"envlog-{which}-0.csv").read_text() == "".join(l + "\n" for l in want[key]), which
    assert (tmp_path / "explog-test-0.csv").read_text() == _expected_csv(otaps, 2, "line", "", True)
    assert not (tmp_path / "explog-learn-0.csv").exists()


# ----------------------------------------------------------- edge cases -----
def test_empty_batches_and_bad_arguments(grlx):
    """n = 0 is legal everywhere; out-of-range arguments come back as GRLX_ERR_INVALID, never a crash."""
    capi = grlx.capi
    cfg = grlx.pendulum_sarsa_config(2)
    st, obs, rew, term = grlx.runner.env_step(cfg, np.zeros((0, 3)), np.zeros(0))
    assert st.shape == (0, 3) and obs.shape == (0, 2) and rew.shape == (0,) and term.shape == (0,)
    assert grlx.runner.device_math(0, np.zeros(0)).shape == (0,)
    assert grlx.runner.rand48_at(1, np.zeros(0, np.uint64)).shape == (0,)
    r = grlx.Runner(cfg, [5, 6])
    assert r.read(np.zeros(0, np.int32), np.zeros((0, 16), np.uint32)).shape == (0,)
    r.run(0); r.sync()
    assert r.n_rows() == 0 and r.step_counts() == (0, 0)
    for bad in (lambda: r.rows(2), lambda: r.rows(0, 0, 10**6), lambda: r.weights(-1, [0]), lambda: r.weights(0, [0], table=1),
                lambda: r.read([2], np.zeros((1, 16), np.uint32)), lambda: r.read([0], np.full((1, 16), 8388608, np.uint32)),
                lambda: r.env_state(7)):
        with pytest.raises(capi.GrlxError) as ei:
            bad()
        assert ei.value.code == capi.ERR_INVALID
    r.close()
    # unsupported graphs are refused, not emulated
    for field, value in (("trace", 7), ("discrete_time", 0), ("action_steps", 9), ("lambda_", 0.99)):
        c = grlx.pendulum_sarsa_config(1)
        setattr(c, field, value)
        with pytest.raises(capi.GrlxError) as ei:
            grlx.Runner(c, [1])
        assert ei.value.code == capi.ERR_INVALID, field
    c = grlx.pendulum_sarsa_config(1)
    c.projector.tilings = 8
    with pytest.raises(capi.GrlxError):
        grlx.Runner(c, [1])


def test_more_rows_than_reserved_is_reported(grlx):
    cfg = grlx.pendulum_sarsa_config(3, max_rows=2)
    r = grlx.Runner(cfg, [1, 2, 3])
    r.run(33)
    with pytest.raises(grlx.capi.GrlxError) as ei:
        r.sync()
    assert ei.value.code == grlx.capi.ERR_ROWS_FULL
    r.close()


def test_kernel_selection(grlx):
    """Which instantiation a configuration runs: the reference yamls take their compile-time build, any
    deviation (or force_generic, or taps) takes the generic one -- never silently the wrong constants."""
    from tests import configs
    cases = [(grlx.pendulum_sarsa_config(4), 2), (grlx.pendulum_sarsa_config(4, agent=1), 2), (grlx.pendulum_sarsa_config(4, agent=3), 2),
             (grlx.pendulum_sarsa_config(4, force_generic=1), 1), (grlx.pendulum_sarsa_config(4, alpha=0.25), 1),
             (grlx.pendulum_sarsa_config(4, tap_replica=0, tap_capacity=10), 3),
             (configs.cart_pole_ac(grlx, 4)[0], 2), (configs.cart_pole_ac(grlx, 4, sigma=4.0)[0], 1),
             (configs.cart_pole_ac(grlx, 4, force_generic=1)[0], 1), (configs.acrobot(grlx, 4)[0], 2), (configs.acrobot(grlx, 4, agent=0)[0], 1),
             (configs.acrobot(grlx, 4, alpha=0.25)[0], 1), (configs.compass_walker(grlx, 4)[0], 2), (configs.compass_walker(grlx, 4, gamma=0.96)[0], 1)]
    for cfg, want in cases:
        r = grlx.Runner(cfg, [1, 2, 3, 4])
        assert r.last_kernel() == 0
        r.run(1); r.sync()
        assert r.last_kernel() == want, (cfg.env, cfg.agent, want, r.last_kernel())
        r.close()


def test_specialised_actor_critic_equals_generic(grlx):
    from tests import configs
    seeds = np.arange(60, 71)
    out = []
    for force in (0, 1):
        cfg, _ = configs.cart_pole_ac(grlx, len(seeds), force_generic=force)
        r = grlx.Runner(cfg, seeds)
        r.run(9); r.run(8); r.sync()
        assert r.last_kernel() == (1 if force else 2)
        slots = np.arange(0, 8388608, 211, dtype=np.uint32)
        out.append((np.stack([r.rows(k)[2] for k in range(len(seeds))]), r.weights(3, slots, table=0), r.weights(3, slots, table=1),
                    [list(r.rng(k)) for k in range(len(seeds))], np.stack([r.env_state(k) for k in range(len(seeds))])))
        r.close()
    assert_bit_equal(out[0][0], out[1][0], "rows")
    assert_bit_equal(out[0][1], out[1][1], "critic weights")
    assert_bit_equal(out[0][2], out[1][2], "actor weights")
    assert out[0][3] == out[1][3]
    assert_bit_equal(out[0][4], out[1][4], "env states")


@pytest.mark.parametrize("agent", [0, 1, 3])
def test_specialised_kernel_equals_generic(grlx, agent):
    """The headline configuration (and its Q / Expected-SARSA siblings) runs a compile-time specialised
    instantiation (parameters as literals); it must be indistinguishable from the generic kernel."""
    seeds = np.arange(1, 38)
    out = []
    for force in (0, 1):
        cfg = grlx.pendulum_sarsa_config(len(seeds), force_generic=force, agent=agent)
        r = grlx.Runner(cfg, seeds)
        r.run(55); r.sync()
        rows = np.stack([r.rows(k)[2] for k in range(len(seeds))])
        slots = np.arange(0, 8388608, 97, dtype=np.uint32)
        out.append((rows, r.weights(0, slots), r.weights(len(seeds) - 1, slots), [list(r.rng(k)) for k in range(len(seeds))]))
        r.close()
    assert_bit_equal(out[0][0], out[1][0], "rows")
    assert_bit_equal(out[0][1], out[1][1], "weights of replica 0")
    assert_bit_equal(out[0][2], out[1][2], "weights of the last replica")
    assert out[0][3] == out[1][3]


@pytest.mark.parametrize("memory,agent", [(8388608, 0), (4096, 0), (2048, 1), (4099, 3)])
def test_deferred_update_equals_in_place_and_oracle(grlx, memory, agent):
    """The production instantiation applies the TD update of a step one pass later, in the
    shadow of the next step's table loads; the diagnostic instantiation applies it in place.
    Both must give the oracle's rows, weights and RNG positions -- also with a tiny hash memory,
    where most slots are shared between tilings (write-through entries, cross-lane aliases,
    reloads) and the rarely taken paths of the update run all the time."""
    seeds = list(range(3, 12))
    trials = 44
    slots = np.arange(0, memory, 1 if memory < 10000 else 1021, dtype=np.uint32)
    got = []
    for inplace in (False, True):
        cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent)
        cfg.projector.memory = memory
        r = grlx.Runner(cfg, seeds)
        if inplace:
            r.set_diag(True)
        r.run(20); r.run(24); r.sync()
        got.append(([r.rows(k) for k in range(len(seeds))], [r.weights(k, slots) for k in range(len(seeds))],
                    [list(r.rng(k))[:3] for k in range(len(seeds))]))
        r.close()
    for k, seed in enumerate(seeds):
        spec = ob.pendulum_sarsa_spec(agent=agent)
        spec.projector.memory = memory
        e = ob.Experiment(spec, seed=seed)
        rows, _ = e.run(trials)
        for which, g in zip(("deferred", "in place"), got):
            t, s, rew = g[0][k]
            assert list(s) == [x.steps for x in rows], which
            assert_bit_equal(rew, [x.reward for x in rows], f"{which}: returns of seed {seed}")
            assert_bit_equal(g[1][k], e.weights(slots), f"{which}: weights of seed {seed}")
            assert g[2][k] == list(e.rng())[:3], which


# --------------------------------------------------- policy load (.dat) ----
@pytest.mark.parametrize("agent", [0, 1])
def test_load_weights_equals_oracle_set_params(grlx, agent):
    """ParameterizedRepresentation {action: load} (representation.h:231-263): a policy trained
    elsewhere is loaded into replicas 1..3 in mid-run (what they had learned is discarded, RNG
    streams and counters go on); replicas 0 and 4 are not touched.  Rows, RNG positions and the
    complete dense tables must equal the oracle's setParams() at the same point."""
    src = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent), seed=21)
    src.run(33)
    image = src.all_weights()
    seeds = [5, 6, 7, 8, 9]
    cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent)
    r = grlx.Runner(cfg, seeds)
    r.run(11); r.sync()
    r.load_weights(image, first_replica=1, n_replicas=3)
    r.run(22); r.sync()
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent), seed=seed)
        rows, _ = e.run(11)
        if 1 <= k <= 3:
            e.set_weights(image)
        rows2, _ = e.run(22)
        rows = list(rows) + list(rows2)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        if k in (0, 2):
            assert_bit_equal(r.export_weights(k), e.all_weights(), f"dense table of replica {k}")
    with pytest.raises(grlx.capi.GrlxError, match="mismatch"):
        r.load_weights(image[:-1])
    with pytest.raises(grlx.capi.GrlxError, match="replica range"):
        r.load_weights(image, first_replica=3, n_replicas=3)
    r.close()


def test_load_weights_actor_critic(grlx):
    """Both tables of an actor-critic graph loaded before the first run (critic = table 0, actor = 1)."""
    from tests import configs
    cfg, spec = configs.cart_pole_ac(grlx, 3)
    src = ob.Experiment(spec, seed=77)
    src.run(12)
    critic, actor = src.all_weights(0), src.all_weights(1)
    seeds = [41, 42, 43]
    r = grlx.Runner(cfg, seeds)
    r.load_weights(critic, table=0)
    r.load_weights(actor, table=1)
    r.run(12); r.sync()
    rng = np.random.default_rng(23)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=seed)
        e.set_weights(critic, 0); e.set_weights(actor, 1)
        rows, _ = e.run(12)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        slots = rng.integers(0, 8388608, 5000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots, table=0), e.weights(slots, table=0), "critic")
        assert_bit_equal(r.weights(k, slots, table=1), e.weights(slots, table=1), "actor")
    with pytest.raises(grlx.capi.GrlxError, match="before the first run"):
        r.load_weights(critic, table=0)
    r.close()
