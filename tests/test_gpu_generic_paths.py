"""Generic-kernel parameters of the fused path that the headline configuration never varies, each against the
oracle (rows, RNG positions, environment state, weights): `test_interval` (incl. the "row every trial" branch
of online_learning.cpp:160,238), `randomization` (pendulum.cpp:97-103), epsilon decay (greedy.cpp:144-149),
finite output limits on a Q table (linear.cpp:150-153, 207-215) -- and the regression test of the round-1
device fault (passes of the deferred-update kernel that evict nothing)."""
import os

import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

GOLDEN_PID = os.path.join(os.path.dirname(__file__), "golden", "cart_pole_balancing-pid-0.txt")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    bad = np.nonzero(bits(a) != bits(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a.flat[bad[0]]!r} vs {b.flat[bad[0]]!r}"


def _apply(obj, over):
    for k, v in over.items():
        if k.startswith("representation."):
            setattr(obj.representation, k.split(".", 1)[1], v)
        else:
            setattr(obj, k, v)


def _run_both(grlx, seeds, trials, over, agent=0, chunks=None, force_generic=0):
    """The production (deferred-update) kernel of `grlx` against one scalar oracle run per seed."""
    cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent, max_rows=trials + 1, force_generic=force_generic)
    _apply(cfg, over)
    r = grlx.Runner(cfg, seeds)
    for c in (chunks or [trials]):
        r.run(c)
    r.sync()                                                   # raises on any sticky status bit (ST_BAD_POS included)
    assert r.last_kernel() in (1, 2)                           # never the diagnostic in-place instantiation
    rng = np.random.default_rng(11)
    for k, seed in enumerate(seeds):
        spec = ob.pendulum_sarsa_spec(agent=agent)
        _apply(spec, over)
        e = ob.Experiment(spec, seed=int(seed))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert len(rows) == r.n_rows()
        assert list(t) == [x.trial for x in rows], f"trial column of seed {seed}"
        assert list(s) == [x.steps for x in rows], f"steps column of seed {seed}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert_bit_equal(r.row_times(k), [x.time for x in rows], f"episode times of seed {seed}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3], f"RNG positions of seed {seed}"
        assert_bit_equal(r.env_state(k), e.state(), f"env state of seed {seed}")
        slots = rng.integers(0, cfg.projector.memory, 2000).astype(np.uint32)
        assert_bit_equal(r.weights(k, slots), e.weights(slots), f"weights of seed {seed}")
        e.close()
    learn, test = r.step_counts()
    r.close()
    return learn, test


@pytest.mark.parametrize("agent", [0, 1])
def test_passes_without_eviction_on_a_ragged_batch(grlx, agent):
    """Round-1 device fault (gpurun_out/abort.log of that round): the first launch of the deferred-update kernel
    aborted at the stream sync.  On passes where the update evicts nothing the held eviction is "no position"
    (kInvalidPos); stored unguarded that is an address 64 GiB past the replica's table.  Here NO pass ever
    evicts (7-step episodes: the trace is cleared before it fills), the batch is ragged (7 replicas: one dead
    16-lane group whose lanes hold no position at all), and every table access is masked into the replica's
    table, so this must simply equal the oracle."""
    learn, test = _run_both(grlx, [5, 6, 7, 8, 9, 10, 11], 44, dict(timeout=0.2), agent=agent, chunks=[20, 24])
    assert learn == 7 * 40 * 7 and test == 7 * 4 * 7


@pytest.mark.parametrize("over,trials", [
    (dict(test_interval=-1), 24),                               # no test trials: a row for EVERY (learning) trial
    (dict(test_interval=0), 12),                                # every trial is a test trial: nothing is ever learned
    (dict(test_interval=3), 30),
    (dict(randomization=1.0), 33),                              # start angle drawn from the thread-local stream
    (dict(decay_rate=0.99, decay_min=0.1), 33),                 # epsilon decays at every episode start
    (dict(decay_rate=0.5, decay_min=0.3, epsilon=0.4), 22),     # ... down to decay_min
    ({"representation.output_min": -60.0, "representation.output_max": 0.5}, 33),   # finite limits: clamped reads AND clamped weights
    ({"representation.output_min": -60.0, "representation.output_max": 0.5, "representation.limit": 0}, 33),   # limit = 0: reads only
    (dict(test_interval=-1, randomization=1.0, decay_rate=0.9, decay_min=0.05), 24),
])
def test_generic_parameters_against_the_oracle(grlx, over, trials):
    seeds = [21, 22, 23, 24, 25]
    learn, test = _run_both(grlx, seeds, trials, over, chunks=[trials // 2, trials - trials // 2])
    ti = over.get("test_interval", 10)
    n_test = 0 if ti < 0 else sum(1 for tt in range(trials) if tt % (ti + 1) == ti)
    assert test == len(seeds) * n_test * 100 and learn == len(seeds) * (trials - n_test) * 100


def test_q_learning_with_limits_and_decay_on_five_actions(grlx):
    """The other instantiation of the pendulum kernel (5 actions), Q-learning, all the knobs at once."""
    over = dict(action_steps=5, test_interval=4, decay_rate=0.97, decay_min=0.2, randomization=1.0)
    over["representation.output_min"] = -500.0
    over["representation.output_max"] = 10.0
    _run_both(grlx, [31, 32, 33], 25, over, agent=1, chunks=[10, 15])


def test_cart_pole_dynamics_along_the_reference_golden_trajectory(grlx):
    """The trajectory that prints the reference's tests/template/cart_pole_balancing-pid-0.txt (oracle, portable
    arithmetic; tests/test_oracle_golden.py pins it byte for byte): every one of its 2000 transitions through
    grlx_env_step must give the oracle's next state bit for bit -- with the balancing task also its observation,
    reward and terminal flag; with the swing-up task of cfg/cart_pole/ac_tc.yaml (the task of BASELINE config 3)
    the same next state, since both tasks sit on the same dynamics/cart_pole + DynamicalModel::step."""
    from tests import configs
    spec = ob.cart_pole_balancing_pid_spec(math=ob.MATH_PORTABLE, tap_starts=1)
    e = ob.Experiment(spec, seed=1)
    rows, taps = e.run(10, tap_cap=3000)
    with open(GOLDEN_PID) as f:
        assert e.format_rows(rows) == f.read()
    assert len(taps) == 10 * 201
    prev_state, prev_action, want = [], [], []
    for a, b in zip(taps[:-1], taps[1:]):
        if b.terminal == -1:
            continue                                            # b starts a new trial: no transition a -> b
        prev_state.append(list(a.state[:5])); prev_action.append(a.action); want.append(b)
    assert len(want) == 2000
    cfg = grlx.pendulum_sarsa_config(1)
    cfg.env = grlx.capi.ENV_CART_POLE_BALANCING
    cfg.control_step, cfg.integration_steps, cfg.timeout = 0.05, 5, 9.99
    cfg.action_min, cfg.action_max = -15.0, 15.0
    cfg.projector.dims = 5
    for i in range(5):
        cfg.projector.resolution[i], cfg.projector.wrapping[i] = 1.0, 0.0
    st, obs, rew, term = grlx.runner.env_step(cfg, prev_state, prev_action)
    assert_bit_equal(st, [list(t.state[:5]) for t in want], "next state (balancing task)")
    assert_bit_equal(obs, [list(t.obs[:4]) for t in want], "observation")
    assert_bit_equal(rew, [t.reward for t in want], "reward")
    assert list(term) == [t.terminal for t in want] and list(term).count(1) == 10
    cfg2, _ = configs.cart_pole_ac(grlx, 1)
    st2, _, _, _ = grlx.runner.env_step(cfg2, prev_state, prev_action)
    assert_bit_equal(st2, st, "next state (swing-up task, same dynamics)")
    with pytest.raises(grlx.capi.GrlxError) as ei:              # no fused TD rollout exists for this task
        grlx.Runner(cfg, [1])
    assert ei.value.code == grlx.capi.ERR_INVALID


# ------------------------------------------------ wide waves (8 replicas per wave) ---
def _wide_vs_oracle(grlx, make, n, trials, sample, chunks, **over):
    """replicas_per_wave = 8 (two sub-batches of four share one environment phase, grlx_rollout_wide.h) against the
    scalar oracle AND against the 4-per-wave kernel: rows, RNG positions, environment state, weights, step counts."""
    got = {}
    for rpw in (8, 4):
        cfg, spec = make(grlx, n, **over)
        cfg.replicas_per_wave = rpw
        cfg.max_rows = trials + 1
        seeds = np.arange(101, 101 + n)
        r = grlx.Runner(cfg, seeds)
        assert r.replicas_per_wave() == rpw
        for c in chunks:
            r.run(c)
        r.sync()
        assert r.last_kernel() in (1, 2)
        rng = np.random.default_rng(5)
        slots = rng.integers(0, cfg.projector.memory, 1500).astype(np.uint32)
        got[rpw] = dict(rows=[r.rows(k) for k in sample], times=[r.row_times(k) for k in sample], rng=[list(r.rng(k))[:3] for k in sample],
                        state=[r.env_state(k) for k in sample], w=[r.weights(k, slots) for k in sample], counts=r.step_counts(),
                        load=[r.table_load(k) for k in sample])
        r.close()
    assert got[8]["counts"] == got[4]["counts"]
    assert got[8]["load"] == got[4]["load"]
    for i, k in enumerate(sample):
        e = ob.Experiment(spec, seed=int(101 + k))
        rows, _ = e.run(trials)
        for rpw in (8, 4):
            t, s, rew = got[rpw]["rows"][i]
            assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows], f"rpw {rpw} replica {k}"
            assert_bit_equal(rew, [x.reward for x in rows], f"rpw {rpw}: returns of replica {k}")
            assert_bit_equal(got[rpw]["times"][i], [x.time for x in rows], f"rpw {rpw}: episode times of replica {k}")
            assert got[rpw]["rng"][i] == list(e.rng())[:3], f"rpw {rpw}: RNG positions of replica {k}"
            assert_bit_equal(got[rpw]["state"][i], e.state(), f"rpw {rpw}: env state of replica {k}")
            assert_bit_equal(got[rpw]["w"][i], e.weights(slots), f"rpw {rpw}: weights of replica {k}")
        e.close()


@pytest.mark.parametrize("name,n,trials,agent,generic", [("pendulum", 21, 33, 0, 0), ("pendulum", 13, 22, 1, 0), ("pendulum", 9, 22, 3, 0),
                                                         ("acrobot", 19, 44, 1, 0), ("acrobot", 10, 33, 0, 0), ("compass_walker", 11, 22, 1, 0),
                                                         ("acrobot", 19, 44, 1, 1), ("compass_walker", 11, 22, 1, 1)])
def test_wide_waves_bit_exact(grlx, name, n, trials, agent, generic):
    """Ragged batches (n mod 8 != 0: a half-empty sub-batch, a dead 16-lane group), every replica checked; episodes of
    the acrobot and the walker end at different steps inside one wave (absorbing states), so sub-batches finish apart.
    Q-learning on the acrobot and on the walker with these parameters runs compile-time specialised instantiations
    (SpecAcrobotQ, SpecWalkerQ; both layouts); generic = 1 forces the run-time-parameter kernels on the same experiment."""
    from tests import configs
    make = {"pendulum": configs.pendulum, "acrobot": configs.acrobot, "compass_walker": configs.compass_walker}[name]
    if name != "pendulum" and agent == 1:
        cfg, _ = make(grlx, n, agent=agent, force_generic=generic)
        r = grlx.Runner(cfg, np.arange(n)); r.run(1); r.sync()
        assert r.last_kernel() == (1 if generic else 2)
        r.close()
    _wide_vs_oracle(grlx, lambda g, k, **o: make(g, k, agent=agent, force_generic=generic, **o), n, trials, list(range(n)),
                    [trials // 3, trials - trials // 3])


@pytest.mark.parametrize("name,n,agent,generic,wide", [("acrobot", 37, 1, 0, 16), ("compass_walker", 21, 1, 0, 16), ("acrobot", 19, 0, 0, 16),
                                                       ("acrobot", 18, 1, 1, 16), ("compass_walker", 17, 1, 1, 16),
                                                       ("compass_walker", 41, 1, 0, 32), ("compass_walker", 35, 0, 1, 32)])
def test_sixteen_replicas_per_wave_bit_exact(grlx, name, n, agent, generic, wide):
    """Round 4: four sub-batches per wave (rollout_wide_kernel<., 3, 4, .>: one environment phase per 16 replicas, the lane state of the third
    and fourth sub-batch parked in device memory between their turns).  Ragged batches (37 = 2 full waves + 5: a wave with one and a quarter
    sub-batches), trials in two launches, then two successive steps budgets with test trials of two episodes; every replica against the
    oracle: rows (returns and times), streams, environment state, sampled weights, step counts, table loads == the 8-per-wave layout's.
    wide = 32: EIGHT sub-batches per wave for the compass walker (two lanes per replica in the environment phase; one sub-batch parked in LDS,
    two in registers, five in device memory requested a turn ahead); 41 = one full wave + 9: two and a quarter sub-batches."""
    from tests import configs
    make = {"acrobot": configs.acrobot, "compass_walker": configs.compass_walker}[name]
    trials = 14 if name == "acrobot" else 9
    got = {}
    seeds = np.arange(201, 201 + n)
    for rpw in (wide, 8):
        cfg, spec = make(grlx, n, agent=agent, force_generic=generic, replicas_per_wave=rpw, max_rows=400, test_trials=2)
        spec.test_trials = 2
        spec.math = ob.MATH_PORTABLE
        r = grlx.Runner(cfg, seeds)
        assert r.replicas_per_wave() == rpw
        r.run(trials // 2); r.run(trials - trials // 2)
        for b in (500, 1100):
            r.run_steps(100000, b)
        r.sync()
        assert r.last_kernel() == (1 if (generic or agent != 1) else 2)
        rng = np.random.default_rng(5)
        slots = rng.integers(0, cfg.projector.memory, 1000).astype(np.uint32)
        got[rpw] = dict(rows=[r.rows(k) for k in range(n)], times=[r.row_times(k, 0, len(r.rows(k)[0])) for k in range(n)], rng=[list(r.rng(k))[:4] for k in range(n)],
                        state=[r.env_state(k) for k in range(n)], w=[r.weights(k, slots) for k in range(n)], counts=r.step_counts(),
                        load=[r.table_load(k) for k in range(n)])
        r.close()
    assert got[wide]["counts"] == got[8]["counts"] and got[wide]["load"] == got[8]["load"]
    learn = test = 0
    for k in range(n):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows = e.run(trials)[0]
        for b in (500, 1100):
            e.set_steps_budget(b)
            rows += e.run(100000)[0]
        t, s_, rew = got[wide]["rows"][k]
        assert list(t) == [x.trial for x in rows] and list(s_) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert_bit_equal(got[wide]["times"][k][:len(rows)], [x.time for x in rows], f"episode times of replica {k}")
        assert got[wide]["rng"][k] == list(e.rng())[:4], f"RNG positions of replica {k}"
        assert_bit_equal(got[wide]["state"][k], e.state(), f"env state of replica {k}")
        assert_bit_equal(got[wide]["w"][k], e.weights(slots), f"weights of replica {k}")
        st = e.stats()
        learn += int(st.learn_steps); test += int(st.test_steps)
        e.close()
    assert got[wide]["counts"] == (learn, test)


def test_sixteen_replicas_per_wave_is_chosen_and_refused(grlx):
    """Automatic layout: 16 for the TD agents on the acrobot / the walker from 15 replicas per SIMD on; an explicit 16 elsewhere is refused."""
    capi = grlx.capi
    r = grlx.Runner(grlx.acrobot_q_config(16384, table_log2_capacity=10), np.arange(16384)); assert r.replicas_per_wave() == 16; r.close()
    r = grlx.Runner(grlx.acrobot_q_config(8192, table_log2_capacity=10), np.arange(8192)); assert r.replicas_per_wave() == 8; r.close()
    for make, rpw in ((grlx.pendulum_sarsa_config, 16), (grlx.acrobot_q_config, 32)):
        with pytest.raises(capi.GrlxError) as ei:
            grlx.Runner(make(64, replicas_per_wave=rpw), np.arange(64))
        assert ei.value.code == capi.ERR_INVALID
    r = grlx.Runner(grlx.compass_walker_q_config(32768, table_log2_capacity=10), np.arange(32768)); assert r.replicas_per_wave() == 32; r.close()


def test_wide_waves_generic_parameters_and_tiny_memory(grlx):
    """The generic wide instantiation with the knobs of the generic-parameter tests, and a 2048-slot hash memory where
    nearly every slot is shared between tilings (write-through entries, cross-lane aliases, reloads after the update)."""
    from tests import configs

    def make(g, k, **o):
        cfg, spec = configs.pendulum(g, k, agent=1, **o)
        for obj in (cfg, spec):
            obj.projector.memory = 2048
            obj.test_interval, obj.randomization, obj.decay_rate, obj.decay_min = 3, 1.0, 0.95, 0.2
            obj.representation.output_min, obj.representation.output_max = -300.0, 5.0
        return cfg, spec
    _wide_vs_oracle(grlx, make, 12, 24, list(range(12)), [10, 14])


def test_wide_waves_are_chosen_for_batches_beyond_four_replicas_per_simd(grlx):
    """Automatic layout: 4 replicas per wave while the waves fit the SIMDs (4096 replicas on 1024 SIMDs), 8 beyond;
    actor-critic, QV, advantage and accumulating-trace contexts and tapped contexts stay at 4."""
    r = grlx.Runner(grlx.pendulum_sarsa_config(4096, table_log2_capacity=10), np.arange(4096)); assert r.replicas_per_wave() == 4; r.close()
    r = grlx.Runner(grlx.pendulum_sarsa_config(4100, table_log2_capacity=10), np.arange(4100)); assert r.replicas_per_wave() == 8; r.close()
    r = grlx.Runner(grlx.pendulum_sarsa_config(4100, table_log2_capacity=10, trace=2), np.arange(4100)); assert r.replicas_per_wave() == 4; r.close()
    r = grlx.Runner(grlx.pendulum_sarsa_config(8, replicas_per_wave=8, tap_replica=0, tap_capacity=16), np.arange(8)); assert r.replicas_per_wave() == 4; r.close()
    with pytest.raises(grlx.capi.GrlxError):
        grlx.Runner(grlx.pendulum_sarsa_config(8, replicas_per_wave=16), np.arange(8))


@pytest.mark.parametrize("over,n,wave_limit", [(dict(), 13, 0), (dict(end_stop_penalty=1, ac_update_method=1, ac_step_limit=0.5), 13, 0),
                                               (dict(), 29, 1), (dict(end_stop_penalty=1), 29, 2), (dict(), 16, 1)])
def test_wide_waves_actor_critic_bit_exact(grlx, over, n, wave_limit):
    """rollout_ac_wide_kernel: both actor update methods, the end-stop penalty (episodes of one wave end apart), the
    critic's trace kept across episodes AND launches (three launches), a ragged batch, every replica checked.
    wave_limit 1 / 2: ONE (two) wave(s) for 29 replicas -- the first 8 (16) start in its slots, the others are taken from the
    device-side queue as slots finish (grlx_rollout_ac_wide.h); 16 replicas on one wave: the queue runs dry with every slot
    busy.  12 slots: the wave OWNS its replicas (29, 15 + 14, 16 of them) and rotates them through its slots trial by trial.
    Who runs a replica, and when, must not matter."""
    from tests import configs
    trials = 24
    got = {}
    for rpw in (16, 12, 8, 4):
        cfg, spec = configs.cart_pole_ac(grlx, n, **over)
        cfg.replicas_per_wave = rpw
        cfg.wave_limit = wave_limit
        cfg.max_rows = trials + 1
        r = grlx.Runner(cfg, np.arange(201, 201 + n))
        assert r.replicas_per_wave() == rpw
        r.run(7); r.run(9); r.run(8); r.sync()
        rng = np.random.default_rng(9)
        slots = rng.integers(0, 8388608, 1500).astype(np.uint32)
        got[rpw] = [(r.rows(k), list(r.rng(k))[:2], r.env_state(k), r.weights(k, slots, 0), r.weights(k, slots, 1)) for k in range(n)]
        r.close()
    for k in range(n):
        e = ob.Experiment(spec, seed=201 + k)
        rows, _ = e.run(trials)
        for rpw in (16, 12, 8, 4):
            (t, s, rew), rg, st, w0, w1 = got[rpw][k]
            assert list(s) == [x.steps for x in rows], f"rpw {rpw} replica {k}"
            assert_bit_equal(rew, [x.reward for x in rows], f"rpw {rpw}: returns of replica {k}")
            assert rg == list(e.rng())[:2]
            assert_bit_equal(st, e.state(), f"rpw {rpw}: env state of replica {k}")
            assert_bit_equal(w0, e.weights(slots, 0), f"rpw {rpw}: critic weights of replica {k}")
            assert_bit_equal(w1, e.weights(slots, 1), f"rpw {rpw}: actor weights of replica {k}")
        e.close()


def test_host_layer_plugin_interfaces(grlx, tmp_path):
    """The C++ host layer's Projector::project, Environment::step and Representation::read/write/update
    (grl_amd/csrc/host/objects.h), instantiated from the reference's own yaml and driven through grl_amd/bin/grlx_ops:
    indices, transitions and representation outputs equal the oracle's bit for bit."""
    import subprocess
    from grl_amd import _build
    from tests import configs
    _build.build_host()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(17)
    # Projector::project
    yaml = os.path.join(root, "tests", "golden", "pendulum-sarsa-tc.yaml")
    x = np.column_stack([rng.uniform(0, 6.28, 200), rng.uniform(-30, 30, 200), rng.choice([-3.0, 0.0, 3.0], 200)])
    f = tmp_path / "in.txt"
    f.write_text("\n".join(" ".join(repr(float(v)) for v in row) for row in x) + "\n")
    res = subprocess.run([_build.GRLX_OPS, "project", yaml, "experiment/agent/policy/projector", str(f)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = np.array([[int(v) for v in line.split()] for line in res.stdout.splitlines()], dtype=np.uint32)
    want = ob.tile_project(ob.pendulum_sarsa_spec().projector, x)
    assert (got == want).all()
    # Environment::step: pendulum and cart-pole
    for yaml_name, spec, S, D, amax in (("pendulum-sarsa-tc.yaml", ob.pendulum_sarsa_spec(), 3, 2, 3.0),
                                        ("cart_pole-ac-tc.yaml", configs.cart_pole_ac(None, 1)[1], 5, 4, 15.0)):
        yaml = os.path.join(root, "tests", "golden", yaml_name)
        st = rng.uniform(-2, 2, (150, S)); st[:, -1] = rng.uniform(0, 1, 150)
        act = rng.uniform(-amax, amax, 150)
        f.write_text("\n".join(" ".join(repr(float(v)) for v in list(row) + [a]) for row, a in zip(st, act)) + "\n")
        res = subprocess.run([_build.GRLX_OPS, "envstep", yaml, "experiment/environment", str(f)], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr
        out = np.array([[float(v) for v in line.split()] for line in res.stdout.splitlines()])
        nst, nobs, nrew, nterm = ob.env_step(spec, st, act)
        assert_bit_equal(out[:, :S], nst, yaml_name + ": next state")
        assert_bit_equal(out[:, S:S + D], nobs, yaml_name + ": observation")
        assert_bit_equal(out[:, S + D], nrew, yaml_name + ": reward")
        assert list(out[:, S + D + 1].astype(int)) == list(nterm)
    # Representation::read / write / update (representation.h:60-83): a sequence of operations on the yaml's
    # representation/parameterized/linear object (seed 1), against a dense restatement of linear.cpp:136-216 on the oracle's
    # initial parameter vector -- lazily initialised slots, duplicate indices inside a projection
    yaml = os.path.join(root, "tests", "golden", "pendulum-sarsa-tc.yaml")
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=1)
    dense = {}

    def w(slot):
        if slot not in dense:
            dense[slot] = float(e.weights([slot])[0])
        return dense[slot]

    lines, want = [], []
    for it in range(60):
        idx = rng.integers(0, 300, 16)
        if it % 4 == 0:
            idx[5] = idx[2]
        op = it % 3
        arg = float(rng.uniform(-5, 5))
        lines.append(" ".join([str(op), repr(arg)] + [str(int(v)) for v in idx]))
        if op == 0:
            s = 0.0
            for v in idx:
                s += w(int(v))
            want.append(s / 16)
        elif op == 1:
            s = 0.0
            for v in idx:
                s += w(int(v))
            d = 0.2 * (arg - s / 16)
            for v in idx:
                dense[int(v)] = w(int(v)) + d
        else:
            for v in idx:
                dense[int(v)] = w(int(v)) + arg
    e.close()
    f.write_text("\n".join(lines) + "\n")
    res = subprocess.run([_build.GRLX_OPS, "represent", yaml, "experiment/agent/policy/representation", str(f)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = [float(v) for v in res.stdout.split()]
    assert_bit_equal(got, want, "Representation::read after writes and updates")


@pytest.mark.parametrize("env,agent,memory", [("pendulum", 0, 8388608), ("pendulum", 1, 8388608), ("pendulum", 3, 8388608),
                                              ("pendulum", 0, 2048), ("acrobot", 1, 8388608)])
def test_production_ordering_step_by_step(grlx, env, agent, memory):
    """Per-step records of the DEFERRED-update ordering (the one that is benchmarked; grlx_config.tap_deferred), not of
    the in-place diagnostic one: tile indices, Q-values, actions, rewards, TD errors and trace lengths of every step
    equal the oracle's -- a divergence that healed itself before the next row would show here.  2048-slot memory: nearly
    every slot is shared between tilings, so the reload / write-through paths of the deferred update run on every pass."""
    from tests import configs
    from tests.test_gpu_parity import _compare_taps
    make = {"pendulum": configs.pendulum, "acrobot": configs.acrobot}[env]
    seeds, trials, cap = [41, 42, 43, 44, 45, 46], 23, 2600
    cfg, spec = make(grlx, len(seeds), agent=agent, tap_replica=4, tap_capacity=cap, tap_deferred=1)
    cfg.projector.memory = memory
    spec.projector.memory = memory
    r = grlx.Runner(cfg, seeds)
    r.run(12); r.run(11); r.sync()
    assert r.last_kernel() == 1                                # generic instantiation, NOT the in-place diagnostic one (3)
    e = ob.Experiment(spec, seed=seeds[4])
    rows, otaps = e.run(trials, tap_cap=cap)
    gtaps = r.taps()
    assert len(gtaps) == len(otaps) and len(otaps) > 100
    D = 2 if env == "pendulum" else 4
    for k, (gt, ot) in enumerate(zip(gtaps, otaps)):
        try:
            _compare_taps(gt, ot, A=3, D=D)
        except AssertionError as ex:
            raise AssertionError(f"step {k}: {ex}")
    t, s, rew = r.rows(4)
    assert_bit_equal(rew, [x.reward for x in rows], "returns")
    r.close()


def test_read_back_waits_for_a_non_blocking_run_stream(grlx):
    """Stream contract of include/grlx.h: grlx_run is asynchronous on the caller's stream; every read-back entry point first
    waits for that stream.  A torch side stream is created non-blocking (it does not synchronise with the NULL stream), so
    without the wait these reads would see rollouts in flight."""
    torch = pytest.importorskip("torch")
    seeds = [61, 62, 63, 64, 65, 66, 67, 68]
    cfg = grlx.pendulum_sarsa_config(len(seeds))
    r = grlx.Runner(cfg, seeds)
    side = torch.cuda.Stream()
    r.run(55, side.cuda_stream)                               # no sync: the launches are still queued / running
    rows_now = [r.rows(k) for k in range(len(seeds))]         # grlx_rows + grlx_read_rows
    counts = r.step_counts()
    state = [r.env_state(k) for k in range(len(seeds))]
    slots = np.arange(0, 8388608, 4099, dtype=np.uint32)
    w = [r.weights(k, slots) for k in range(len(seeds))]
    assert counts == (len(seeds) * 50 * 100, len(seeds) * 5 * 100)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=seed)
        rows, _ = e.run(55)
        assert list(rows_now[k][1]) == [x.steps for x in rows]
        assert_bit_equal(rows_now[k][2], [x.reward for x in rows], f"returns of seed {seed}")
        assert_bit_equal(state[k], e.state(), f"env state of seed {seed}")
        assert_bit_equal(w[k], e.weights(slots), f"weights of seed {seed}")
        e.close()
    r.sync(side.cuda_stream)
    r.close()


def test_repeated_policy_loads_do_not_accumulate_device_memory(grlx):
    """grlx_load_weights keeps one dense image (64 MiB) per load on the device; an image no replica refers to any more is
    freed at the next load, so loading N times costs one image, not N."""
    torch = pytest.importorskip("torch")
    r = grlx.Runner(grlx.pendulum_sarsa_config(4), [1, 2, 3, 4])
    dense = np.random.default_rng(3).uniform(0, 1, 8388608)
    r.load_weights(dense)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(6):
        r.load_weights(dense + k)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 2 * 8388608 * 8, (free0, free1)    # at most one more image outstanding, not six
    r.run(3); r.sync()
    assert_bit_equal(r.weights(0, np.array([5, 77, 4242], np.uint32)) >= 0, [True] * 3)
    r.close()


@pytest.mark.parametrize("pattern", ["0xffffffff", "0x7ff80000"])
def test_results_do_not_depend_on_stale_register_content(grlx, monkeypatch, pattern):
    """ROCm 7.2's register allocator placed a live-range-split copy of the tiling key in
    front of a join block's exec restore in the 5-action kernel; the lanes masked off there got whatever the previous
    wave had left in a scratch register, and returns differed from the oracle's from the first update on.  With
    GRLX_POISON_REGISTERS every rollout launch is preceded by a kernel that fills all 512 vector registers per lane
    (and the SGPRs) of every SIMD with the pattern, so a read of a never-written register is no longer a matter of
    luck.  The build's assembly filter moves such copies behind the restore: the 5-action kernel, the generic
    3-action one and the headline specialisation must equal the oracle under any pattern."""
    monkeypatch.setenv("GRLX_POISON_REGISTERS", pattern)
    over = dict(action_steps=5, test_interval=4, decay_rate=0.97, decay_min=0.2, randomization=1.0)
    _run_both(grlx, [31, 32, 33], 15, over, agent=1, chunks=[7, 8])
    _run_both(grlx, [41, 42, 43, 44, 45], 12, dict(test_interval=3), chunks=[12], force_generic=1)
    _run_both(grlx, [51, 52, 53, 54], 12, {}, chunks=[5, 7])


# ------------------------------------------------ sparse tables that grow between launches ---
def test_tables_grow_between_the_launches_of_one_call(grlx):
    """A deployer runs a whole run as ONE grlx_run, which the library chunks into launches of 32 trials: the tables must grow BETWEEN those
    launches too (a launch cannot grow them).  The cart-pole actor-critic's tables start at 2^16 entries per replica (grlx_config_cart_pole_ac)
    and 32 trials create about 25 000 slots per table: 230 trials in one call ended with GRLX_ERR_TABLE_FULL before round 4."""
    from tests import configs
    n, trials = 5, 230
    cfg, spec = configs.cart_pole_ac(grlx, n)
    assert cfg.table_log2_capacity == 16
    cfg.max_rows = trials // 11 + 1
    seeds = np.arange(41, 41 + n)
    r = grlx.Runner(cfg, seeds)
    r.run(trials); r.sync()                                         # ONE call, no sync in between
    assert r.table_capacity() >= 18, r.table_capacity()
    for k in (0, 4):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        e.close()
    r.close()


@pytest.mark.parametrize("kind", ["pendulum_sarsa", "pendulum_wide", "cart_pole_ac", "cart_pole_ac_wide", "target_network", "accumulating"])
def test_tables_grow_between_launches(grlx, kind):
    """grlx_config.table_log2_capacity is the INITIAL size: started at 2^13 entries per replica, with a grlx_sync between
    launches the tables are re-hashed into larger ones whenever the fullest passes a quarter of its capacity
    (grlx_api.cpp grow_tables, rehash_kernel).  Positions are internal -- to the tables, to the actor-critic's persisted
    critic trace and to the target network's values, which are translated -- so rows, RNG positions, environment state and
    weights must equal the oracle's as with large tables, for every kernel family that keeps positions."""
    from tests import configs
    n, chunks = 6, [1, 1, 2, 3, 5, 10]
    over = {}
    if kind.startswith("pendulum") or kind in ("target_network", "accumulating"):
        if kind == "target_network":
            over = dict(target_interval=150, target_tau=0.3)
        if kind == "accumulating":
            over = dict(trace=2)
        cfg, spec = configs.pendulum(grlx, n, table_log2_capacity=13, **over)
        for k, v in over.items():
            setattr(spec, k, v)
        n_tables = 1
    else:
        cfg, spec = configs.cart_pole_ac(grlx, n, table_log2_capacity=13)
        n_tables = 2
    cfg.replicas_per_wave = 8 if kind.endswith("wide") else 4
    trials = sum(chunks)
    cfg.max_rows = trials + 1
    seeds = np.arange(301, 301 + n)
    r = grlx.Runner(cfg, seeds)
    caps = [r.table_capacity()]
    for c in chunks:
        r.run(c); r.sync()                                      # the sync is what lets the next run look at the load
        caps.append(r.table_capacity())
    assert caps[0] == 13 and caps[-1] >= 15 and caps == sorted(caps), caps        # it grew (the cart-pole tables more than once)
    rng = np.random.default_rng(3)
    slots = rng.integers(0, 8388608, 1200).astype(np.uint32)
    for k in range(n):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        for t_ in range(n_tables):
            assert_bit_equal(r.weights(k, slots, t_), e.weights(slots, t_), f"table {t_} of replica {k}")
        if kind == "target_network":
            tw, syncs = r.target_weights(k, slots)
            assert syncs == e.L.orc_target_syncs(e.h) and syncs > 0
            assert_bit_equal(tw, e.weights(slots, table=2), f"target table of replica {k}")
        assert r.table_load(k) * 4 <= (1 << caps[-1]) * 2        # never far beyond the growth threshold
        e.close()
    # a bounded context keeps its size and reports the overflow as before
    cfg2, _ = configs.pendulum(grlx, 2, table_log2_capacity=9, table_log2_max=9)
    r2 = grlx.Runner(cfg2, [1, 2])
    with pytest.raises(grlx.capi.GrlxError) as ei:
        for c in range(6):
            r2.run(3); r2.sync()
    assert ei.value.code == grlx.capi.ERR_TABLE_FULL and r2.table_capacity() == 9
    r2.close()
    r.close()


@pytest.mark.parametrize("family", ["acrobot_wide", "walker_wide", "cart_pole_ac_wide", "accumulating", "target_network", "qv", "advantage"])
def test_other_kernel_families_under_poisoned_registers(grlx, monkeypatch, family):
    """The same diagnostic (GRLX_POISON_REGISTERS) on one small case of every other kernel family, so
    that a register-allocation accident in any of them shows up in the default suite and not only in the full poisoned run."""
    from tests import configs
    monkeypatch.setenv("GRLX_POISON_REGISTERS", "0x7ff80000")
    n, trials = 9, 12
    over, rpw, tables = {}, 4, 1
    if family == "acrobot_wide":
        make, rpw = configs.acrobot, 8
    elif family == "walker_wide":
        make, rpw, trials = configs.compass_walker, 8, 8
    elif family == "cart_pole_ac_wide":
        make, rpw, tables = configs.cart_pole_ac, 8, 2
    elif family == "qv":
        make, tables = configs.pendulum_qv, 2
    else:
        make = configs.pendulum
        over = {"accumulating": dict(trace=2), "target_network": dict(target_interval=200, target_tau=0.5),
                "advantage": dict(agent=grlx.capi.AGENT_ADVANTAGE, kappa=0.2)}[family]
    cfg, spec = make(grlx, n, **{k: v for k, v in over.items() if k in ("agent", "kappa")})
    for k, v in over.items():
        if k != "agent":
            setattr(cfg, k, v); setattr(spec, k, v)
    cfg.replicas_per_wave = rpw
    cfg.max_rows = trials + 1
    seeds = np.arange(501, 501 + n)
    r = grlx.Runner(cfg, seeds)
    r.run(trials // 2); r.run(trials - trials // 2); r.sync()
    rng = np.random.default_rng(13)
    slots = rng.integers(0, 8388608, 800).astype(np.uint32)
    for k in (0, 3, 8):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows, _ = e.run(trials)
        assert_bit_equal(r.rows(k)[2], [x.reward for x in rows], f"{family}: returns of replica {k}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        assert_bit_equal(r.env_state(k), e.state(), f"{family}: env state of replica {k}")
        for t in range(tables):
            assert_bit_equal(r.weights(k, slots, t), e.weights(slots, t), f"{family}: table {t} of replica {k}")
        e.close()
    r.close()


# ------------------------------------------------ runs > 1: Experiment::reset() between runs ---
@pytest.mark.parametrize("graph", ["pendulum_sarsa", "pendulum_q", "cart_pole_ac", "pendulum_qv", "target_network", "target_network_tau0", "safe"])
def test_second_run_continues_the_streams_like_the_reference(grlx, graph):
    """`runs: 2` (online_learning.cpp:124, 307-308): after run 0 the experiment is RESET, not re-created -- the parameters are drawn
    again from the continuing thread-local stream (linear.cpp:104-125), traces are cleared, exploration decay returns to 1, the
    run's counters restart, and no stream is reseeded.  Rows, RNG positions, environment state and the complete dense tables of
    run 1 equal the oracle's; and they differ from what a fresh experiment gives (the deviation round 2 had)."""
    from tests import configs
    make = {"pendulum_sarsa": lambda n: configs.pendulum(grlx, n, agent=0), "pendulum_q": lambda n: configs.pendulum(grlx, n, agent=1),
            "cart_pole_ac": lambda n: configs.cart_pole_ac(grlx, n), "pendulum_qv": lambda n: configs.pendulum_qv(grlx, n),
            "target_network": lambda n: configs.pendulum(grlx, n, agent=1), "target_network_tau0": lambda n: configs.pendulum(grlx, n, agent=0),
            "safe": lambda n: configs.pendulum(grlx, n, agent=0)}[graph]
    seeds = [1, 2, 3, 4, 5]
    cfg, spec = make(len(seeds))
    if graph.startswith("target_network"):
        # round 4: the walk of Experiment::reset reaches the target network too (a provided object is a child configurator): it draws again
        # first, the representation next, synchronize() blends the two fresh vectors; count_ and the number of synchronisations restart
        for obj in (cfg, spec):
            obj.target_interval, obj.target_tau = 170, (0.0 if graph.endswith("tau0") else 0.35)
    if graph == "safe":
        cfg.projector.safe = 1; spec.safe = 1                     # the claims are dropped by the reset (tile_coding.cpp:82-89)
    spec.math = ob.MATH_PORTABLE
    trials = 23
    r = grlx.Runner(cfg, seeds)
    r.run(trials); r.sync()
    first_run = [r.rows(k)[2].copy() for k in range(len(seeds))]
    r.reset_run()
    assert r.n_rows() == 0 and r.step_counts() == (0, 0)
    r.run(trials); r.sync()
    n_tables = 2 if graph in ("cart_pole_ac", "pendulum_qv") else 1
    for k in (0, 4):
        e = ob.Experiment(spec, seed=seeds[k])
        rows0, _ = e.run(trials)
        assert_bit_equal(first_run[k], [x.reward for x in rows0], f"run 0, replica {k}")
        e.reset_run()
        rows1, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(t) == [x.trial for x in rows1] and list(s) == [x.steps for x in rows1], f"run 1, replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows1], f"returns of run 1, replica {k}")
        n_streams = 2 if graph == "cart_pole_ac" else 4          # the actor-critic graph has no samplers: global and thread-local stream only
        assert list(r.rng(k))[:n_streams] == list(e.rng())[:n_streams]
        assert_bit_equal(r.env_state(k), e.state(), f"env state after run 1, replica {k}")
        for tb in range(n_tables):
            assert_bit_equal(r.export_weights(k, tb), e.all_weights(tb), f"table {tb} after run 1, replica {k}")
        if graph.startswith("target_network"):
            slots = np.random.default_rng(23).integers(0, 8388608, 4000).astype(np.uint32)
            tw, nsync = r.target_weights(k, slots)
            assert nsync == e.L.orc_target_syncs(e.h) and nsync > 0
            assert_bit_equal(tw, e.weights(slots, table=2), f"target network after run 1, replica {k}")
        # ... and NOT the rows of a fresh experiment with the same seed (what re-creating per run would print)
        assert not np.array_equal(rew, first_run[k])
        e.close()
    r.close()


def _golden_yaml():
    return open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()


def test_deployer_two_runs_steps_budget_and_save_every(grlx, tmp_path):
    """The edges of OnlineLearningExperiment::run through grlxd and the reference's own yaml keys:
    runs: 2 (run 1 continues the streams, online_learning.cpp:307-308), steps (a run ends at the first trial boundary with
    ss >= steps, :154), save_every: test (a policy file after every test trial, :281-290)."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = _golden_yaml()
    assert "runs: 1" in text and "steps: 0" in text and "trials: 2000" in text
    # (a) runs: 2
    y = tmp_path / "two.yaml"
    y.write_text(text.replace("runs: 1", "runs: 2").replace("trials: 2000", "trials: 33"))
    res = subprocess.run([grlxd, "-s", "5", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=5)
    rows0, _ = e.run(33)
    e.reset_run()
    rows1, _ = e.run(33)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows0)
    assert (tmp_path / "pendulum-sarsa-tc-1.txt").read_text() == e.format_rows(rows1)
    assert e.format_rows(rows0) != e.format_rows(rows1)
    e.close()
    # (b) steps: 2450 with trials: 0 -- 100-step episodes, a test trial every 11th: learning trials 0..24 bring ss to 2500 >= 2450
    # at the boundary after trial index 26 (two test trials in between), so 27 trials run and two rows are written
    y = tmp_path / "steps.yaml"
    y.write_text(text.replace("steps: 0", "steps: 2450").replace("trials: 2000", "trials: 0"))
    res = subprocess.run([grlxd, "-s", "5", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=5)
    rows, _ = e.run(27)
    assert e.stats().learn_steps == 2500 and len(rows) == 2
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)
    e.close()
    # (c) save_every: test -- the file written after the first test trial (trial index 10) holds the table at that moment
    y = tmp_path / "save.yaml"
    y.write_text(text.replace("save_every: never", "save_every: test").replace("trials: 2000", "trials: 12"))
    res = subprocess.run([grlxd, "-s", "5", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    files = sorted(p.name for p in tmp_path.glob("pendulum-sarsa-tc-run0-trial*.dat"))
    assert files == ["pendulum-sarsa-tc-run0-trial10-experiment_agent_policy_representation.dat"]
    e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=5)
    e.run(11)
    assert_bit_equal(np.fromfile(tmp_path / files[0], dtype="<f8"), e.all_weights(0), "policy saved after the first test trial")
    e.close()


@pytest.mark.parametrize("graph,n,budgets", [("acrobot_q", 21, (700, 1500)), ("compass_walker_q", 13, (600, 1300)), ("cart_pole_ac", 9, (900, 2000)),
                                             ("pendulum_sarsa", 6, (450, 1250))])
@pytest.mark.parametrize("rpw", [4, 8])
def test_steps_budget_stops_every_replica_at_its_own_trial(grlx, graph, n, budgets, rpw):
    """experiment/online_learning:steps on the device (grlx_run_steps; online_learning.cpp:154): a replica starts no further trial once
    its learning steps have reached the budget.  With absorbing environments the replicas of a wave stop at different trials; rows (ragged),
    RNG positions, environment state and step counts equal the oracle's, in both wave layouts, over two successive budgets."""
    from tests import configs
    make = {"acrobot_q": configs.acrobot, "compass_walker_q": configs.compass_walker, "cart_pole_ac": configs.cart_pole_ac,
            "pendulum_sarsa": lambda g, k, **o: configs.pendulum(g, k, agent=0, **o)}[graph]
    over = dict(replicas_per_wave=rpw, max_rows=400)
    if graph == "cart_pole_ac":
        over["end_stop_penalty"] = 1                  # ragged episodes for the actor-critic graph too
    cfg, spec = make(grlx, n, **over)
    spec.math = ob.MATH_PORTABLE
    seeds = np.arange(1, n + 1)
    r = grlx.Runner(cfg, seeds)
    assert r.replicas_per_wave() == rpw
    oracles = [ob.Experiment(spec, seed=int(s)) for s in seeds]
    orows = [[] for _ in seeds]
    rows_seen = set()
    for budget in budgets:
        r.run_steps(100000, budget)
        r.sync()
        for k, e in enumerate(oracles):
            e.set_steps_budget(budget)
            orows[k] += e.run(100000)[0]
    total_learn = 0
    for k, e in enumerate(oracles):
        st = e.stats()
        total_learn += int(st.learn_steps)
        assert st.learn_steps >= budgets[-1]
        rows = orows[k]
        t, s, rew = r.rows(k)
        assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        n_streams = 2 if graph == "cart_pole_ac" else 4
        assert list(r.rng(k))[:n_streams] == list(e.rng())[:n_streams], f"replica {k}"
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        rows_seen.add(len(rows))
        e.close()
    assert r.step_counts()[0] == total_learn
    if graph != "pendulum_sarsa":
        assert len(rows_seen) > 1                     # the replicas did stop at different trials
    r.close()


@pytest.mark.parametrize("rpw", [4, 8])
def test_actor_critic_with_different_tile_codings(grlx, rpw):
    """cfg/cart_pole/ac_tc.yaml gives actor and critic the same tile coding, which the kernels exploit (twin tables: one key
    resolution and one creation path for both).  With DIFFERENT tile codings -- another resolution and memory for the critic --
    the two tables are independent again; rows, streams, states and both dense tables against the oracle, both wave layouts."""
    from tests import configs
    n, trials = 11, 23
    cfg, spec = configs.cart_pole_ac(grlx, n, replicas_per_wave=rpw, end_stop_penalty=1)
    for obj in (cfg, spec):
        obj.projector.memory = 4194304
        obj.projector.resolution[0] = 1.25
        obj.projector.resolution[2] = 5.0
    spec.math = ob.MATH_PORTABLE
    seeds = np.arange(3, 3 + n)
    r = grlx.Runner(cfg, seeds)
    r.run(12); r.run(trials - 12); r.sync()
    for k in (0, 5, 10):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2]
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        assert_bit_equal(r.export_weights(k, 0), e.all_weights(0), f"critic table of replica {k}")
        assert_bit_equal(r.export_weights(k, 1), e.all_weights(1), f"actor table of replica {k}")
        e.close()
    r.close()


@pytest.mark.parametrize("graph,n", [("pendulum_sarsa", 6), ("acrobot_q", 13), ("compass_walker_q", 9), ("cart_pole_ac", 9)])
@pytest.mark.parametrize("rpw", [4, 8])
def test_test_trials_runs_several_greedy_episodes_per_test_trial(grlx, graph, n, rpw):
    """experiment/online_learning:test_trials = 3 (online_learning.cpp:160-225): a test trial is three greedy episodes, each begun with
    environment start and agent start; reward and time keep adding up across them and the row holds their means (one running sum,
    then one division -- not the mean of three sums).  Rows (reward AND time), streams, states, step counts against the oracle, both wave
    layouts; the second half of the run under a steps budget."""
    from tests import configs
    make = {"acrobot_q": configs.acrobot, "compass_walker_q": configs.compass_walker, "cart_pole_ac": configs.cart_pole_ac,
            "pendulum_sarsa": lambda g, k, **o: configs.pendulum(g, k, agent=0, **o)}[graph]
    over = dict(replicas_per_wave=rpw, max_rows=200, test_trials=3)
    if graph == "cart_pole_ac":
        over["end_stop_penalty"] = 1
    cfg, spec = make(grlx, n, **over)
    spec.test_trials = 3
    spec.math = ob.MATH_PORTABLE
    seeds = np.arange(11, 11 + n)
    r = grlx.Runner(cfg, seeds)
    r.run(23)
    r.run_steps(1000, 3000)
    r.sync()
    learn = test = 0
    for k in range(n):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows = e.run(23)[0]
        e.set_steps_budget(3000)
        rows += e.run(1000)[0]
        t, s, rew = r.rows(k)
        assert list(t) == [x.trial for x in rows] and list(s) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"mean returns of replica {k}")
        assert_bit_equal(r.row_times(k, 0, len(rows)), [x.time for x in rows], f"mean episode times of replica {k}")
        n_streams = 2 if graph == "cart_pole_ac" else 4
        assert list(r.rng(k))[:n_streams] == list(e.rng())[:n_streams], f"replica {k}"
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        st = e.stats()
        learn += int(st.learn_steps); test += int(st.test_steps)
        e.close()
    assert r.step_counts() == (learn, test)
    r.close()


@pytest.mark.parametrize("family", ["qv", "advantage", "accumulating", "accumulating_acrobot", "target_network", "target_network_acrobot",
                                    "safe", "target_network_walker", "target_network_cart_pole"])
def test_steps_budget_and_test_trials_in_the_other_kernel_families(grlx, family):
    """Round 4: the second bound of the trial loop (`steps`, online_learning.cpp:154) and test trials of several greedy episodes
    (:160-225) in the kernels that did not have them -- predictor/critic/qv, advantage learning, the accumulating trace, and the plain
    kernel of target networks / `safe` projections (all four environments).  23 trials, then two successive budgets; rows (trial, steps,
    mean return AND mean time, ragged across replicas of one wave for the absorbing tasks), streams, states, step counts and table values
    against the oracle."""
    from tests import configs
    n, tables = 7, 1
    over, agent, make = {}, None, configs.pendulum
    if family == "qv":
        make, tables = configs.pendulum_qv, 2
    elif family == "advantage":
        agent, over = grlx.capi.AGENT_ADVANTAGE, dict(kappa=0.2)
    elif family.startswith("accumulating"):
        over = dict(trace=2)
    elif family.startswith("target_network"):
        over = dict(target_interval=200, target_tau=0.5)
    elif family == "safe":
        over = dict(safe=1)
    if family.endswith("acrobot"):
        make = configs.acrobot
    elif family.endswith("walker"):
        make = configs.compass_walker
    elif family.endswith("cart_pole"):
        make = lambda g, k, **o: configs.cart_pole_q(g, k, end_stop_penalty=1, **o)
    kw = dict(max_rows=300, test_trials=3)
    if agent is not None:
        kw.update(agent=agent, kappa=over["kappa"])
    cfg, spec = make(grlx, n, **kw)
    for k, v in over.items():
        if k == "safe":
            cfg.projector.safe = v; spec.safe = v
        elif k != "kappa":
            setattr(cfg, k, v); setattr(spec, k, v)
    if agent is not None:
        spec.kappa = over["kappa"]
    spec.test_trials = 3
    spec.math = ob.MATH_PORTABLE
    seeds = np.arange(71, 71 + n)
    r = grlx.Runner(cfg, seeds)
    budgets = (1200, 2600)
    r.run(12)
    for b in budgets:
        r.run_steps(100000, b)
    r.sync()
    rng = np.random.default_rng(17)
    slots = rng.integers(0, 8388608, 600).astype(np.uint32)
    learn = test = 0
    rows_seen = set()
    for k in range(n):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows = e.run(12)[0]
        for b in budgets:
            e.set_steps_budget(b)
            rows += e.run(100000)[0]
        t, s_, rew = r.rows(k)
        assert list(t) == [x.trial for x in rows] and list(s_) == [x.steps for x in rows], f"{family}: replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"{family}: mean returns of replica {k}")
        assert_bit_equal(r.row_times(k, 0, len(rows)), [x.time for x in rows], f"{family}: mean episode times of replica {k}")
        assert list(r.rng(k))[:4] == list(e.rng())[:4], f"{family}: replica {k}"
        assert_bit_equal(r.env_state(k), e.state(), f"{family}: env state of replica {k}")
        for tb in range(tables):
            assert_bit_equal(r.weights(k, slots, tb), e.weights(slots, tb), f"{family}: table {tb} of replica {k}")
        st = e.stats()
        assert st.learn_steps >= budgets[-1]
        learn += int(st.learn_steps); test += int(st.test_steps)
        rows_seen.add(len(rows))
        e.close()
    assert r.step_counts() == (learn, test)
    if family.endswith(("acrobot", "walker")):
        assert len(rows_seen) > 1                     # the replicas did stop at different trials
    r.close()


def test_deployer_test_trials(grlx, tmp_path):
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = _golden_yaml()
    assert "test_interval: 10" in text
    y = tmp_path / "tt.yaml"
    y.write_text(text.replace("test_interval: 10", "test_interval: 10\n  test_trials: 2").replace("trials: 2000", "trials: 33"))
    res = subprocess.run([grlxd, "-s", "9", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    spec = ob.pendulum_sarsa_spec()
    spec.test_trials = 2
    e = ob.Experiment(spec, seed=9)
    rows, _ = e.run(33)
    got = (tmp_path / "pendulum-sarsa-tc-0.txt").read_text().split("\n")
    assert len(got) == 4 and got[3] == ""
    for line, row in zip(got, rows):      # online_learning.cpp:243: trial, steps, reward, episode time (both means), reward / time, wall time
        f = line.split()
        assert int(f[0]) == row.trial and int(f[1]) == row.steps and f[2] == "%.3f" % row.reward and f[3] == "%.3f" % row.time
    e.close()


def test_deployer_one_process_per_gpu_reduces_the_curve_with_rccl(grlx, tmp_path):
    """`grlxd -g 1 -r 5`: the multi-GPU path of the C++ host on the one GPU this box has -- a world-1 RCCL communicator (ncclGetUniqueId,
    ncclCommInitRank), grlx_curve_stats into device memory, ONE ncclAllReduce per run, mean and standard deviation written by rank 0.  The
    per-clone files are the oracle's rows (`-l` layout), and <output>-0-mean.txt is their mean / population standard deviation.  More ranks
    need more GPUs (RCCL refuses two ranks on one device): unmeasured on hardware, as DESIGN.md section 6 says."""
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    y = tmp_path / "mg.yaml"
    y.write_text(_golden_yaml().replace("trials: 2000", "trials: 22"))
    res = subprocess.run([grlxd, "-g", "1", "-r", "5", "-s", "40", "-l", "-q", "-v", "-v", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr + res.stdout
    assert "communicator ready" in res.stdout + res.stderr
    returns = []
    for i in range(5):
        e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=40 + i)
        rows, _ = e.run(22)
        assert (tmp_path / f"pendulum-sarsa-tc-0@{i}.txt").read_text() == e.format_rows(rows)
        returns.append([x.reward for x in rows])
        e.close()
    returns = np.asarray(returns)
    mean = [ln.split() for ln in (tmp_path / "pendulum-sarsa-tc-0-mean.txt").read_text().strip().split("\n")]
    assert [int(f[0]) for f in mean] == [10, 20] and [int(f[1]) for f in mean] == [5, 5]
    np.testing.assert_allclose([float(f[2]) for f in mean], returns.mean(axis=0), rtol=1e-11)
    np.testing.assert_allclose([float(f[3]) for f in mean], returns.std(axis=0), rtol=1e-6)


@pytest.mark.parametrize("agent", [0, 1, 3])
def test_discrete_actions_on_the_cart_pole(grlx, agent):
    """The fused Q kernels on the fourth environment of the path with DISCRETE actions (SARSA / Q / Expected SARSA over 3 forces on the
    cart-pole swing-up task): an instantiation no other test reaches.  Rows, streams, state, weights against the oracle, both layouts."""
    from tests import configs
    n, trials = 9, 23
    seeds = np.arange(41, 41 + n)
    rng = np.random.default_rng(7)
    slots = rng.integers(0, 8388608, 1500).astype(np.uint32)
    for rpw in (4, 8):
        cfg, spec = configs.cart_pole_q(grlx, n, agent=agent, replicas_per_wave=rpw, end_stop_penalty=1)
        spec.math = ob.MATH_PORTABLE
        r = grlx.Runner(cfg, seeds)
        r.run(11); r.run(trials - 11); r.sync()
        for k in (0, 4, 8):
            e = ob.Experiment(spec, seed=int(seeds[k]))
            rows, _ = e.run(trials)
            t, s, rew = r.rows(k)
            assert list(s) == [x.steps for x in rows], f"rpw {rpw} replica {k}"
            assert_bit_equal(rew, [x.reward for x in rows], f"rpw {rpw}: returns of replica {k}")
            assert list(r.rng(k))[:3] == list(e.rng())[:3]
            assert_bit_equal(r.env_state(k), e.state(), f"rpw {rpw}: env state of replica {k}")
            assert_bit_equal(r.weights(k, slots), e.weights(slots), f"rpw {rpw}: weights of replica {k}")
            e.close()
        r.close()


def test_twelve_slots_rotate_under_a_steps_budget_and_test_trials(grlx):
    """The rotating actor-critic kernel (12 slots, grlx_rollout_ac_wide.h) with everything that ends a replica's turn at once: ONE wave
    owns 17 replicas (5 always waiting), episodes are ragged (end-stop penalty), test trials are two greedy episodes (no hand-over
    between them), and two successive steps budgets stop the replicas at trials of their own.  Rows, streams, states, step counts
    and both dense tables' samples against the oracle."""
    from tests import configs
    n, budgets = 17, (900, 2100)
    cfg, spec = configs.cart_pole_ac(grlx, n, replicas_per_wave=12, max_rows=400, test_trials=2, end_stop_penalty=1)
    cfg.wave_limit = 1
    spec.test_trials = 2
    spec.math = ob.MATH_PORTABLE
    seeds = np.arange(301, 301 + n)
    r = grlx.Runner(cfg, seeds)
    assert r.replicas_per_wave() == 12
    oracles = [ob.Experiment(spec, seed=int(s)) for s in seeds]
    orows = [[] for _ in seeds]
    for budget in budgets:
        r.run_steps(100000, budget)
        r.sync()
        for k, e in enumerate(oracles):
            e.set_steps_budget(budget)
            orows[k] += e.run(100000)[0]
    rng = np.random.default_rng(3)
    slots = rng.integers(0, 8388608, 1200).astype(np.uint32)
    total_learn, rows_seen = 0, set()
    for k, e in enumerate(oracles):
        total_learn += int(e.stats().learn_steps)
        t, s, rew = r.rows(k)
        assert list(t) == [x.trial for x in orows[k]] and list(s) == [x.steps for x in orows[k]], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in orows[k]], f"returns of replica {k}")
        assert_bit_equal(r.row_times(k, 0, len(orows[k])), [x.time for x in orows[k]], f"times of replica {k}")
        assert list(r.rng(k))[:2] == list(e.rng())[:2], f"replica {k}"
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        assert_bit_equal(r.weights(k, slots, 0), e.weights(slots, 0), f"critic weights of replica {k}")
        assert_bit_equal(r.weights(k, slots, 1), e.weights(slots, 1), f"actor weights of replica {k}")
        rows_seen.add(len(orows[k]))
        e.close()
    assert r.step_counts()[0] == total_learn
    assert len(rows_seen) > 1
    r.close()
