"""The environment server of the pendulum rollout kernels (grl_amd/csrc/grlx_env_server.h): the integration of the next step moves to a
second kernel that shares the SIMDs with the rollout waves.  Same operations on the same arguments, so every result must be the one of
the kernel that integrates itself (GRLX_ENV_SERVER=0) -- and the one of the oracle, which the rest of the GPU suite checks with the server on
by default.  Also: a server that is not there (every replica gives up waiting and integrates itself), ragged waves, chunked launches."""
import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _snapshot(grlx, seeds, chunks, agent=0, **over):
    cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent, max_rows=sum(chunks) + 1, **over)
    r = grlx.Runner(cfg, seeds)
    for c in chunks:
        r.run(c)
    r.sync()
    rng = np.random.default_rng(5)
    slots = rng.integers(0, cfg.projector.memory, 4000).astype(np.uint32)
    out = {"counts": r.env_server_counts(), "kernel": r.last_kernel(), "rows": [], "state": [], "rng": [], "w": []}
    for k in range(len(seeds)):
        t, s, rew = r.rows(k)
        out["rows"].append((list(t), list(s), np.asarray(rew, dtype=np.float64).view(np.uint64).tolist(),
                            np.asarray(r.row_times(k), dtype=np.float64).view(np.uint64).tolist()))
        out["state"].append(np.asarray(r.env_state(k), dtype=np.float64).view(np.uint64).tolist())
        out["rng"].append(list(r.rng(k)))
        out["w"].append(np.asarray(r.weights(k, slots), dtype=np.float64).view(np.uint64).tolist())
    r.close()
    return out


def _same(a, b):
    for key in ("rows", "state", "rng", "w"):
        assert a[key] == b[key], key


@pytest.mark.parametrize("agent", [0, 1, 3])          # SARSA, Q, Expected SARSA: the three specialised instantiations
@pytest.mark.parametrize("n", [1, 7, 64])
def test_server_on_equals_server_off(grlx, monkeypatch, agent, n):
    seeds = np.arange(1, n + 1)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _snapshot(grlx, seeds, [25, 10], agent=agent)
    assert off["counts"] == (0, 0)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = _snapshot(grlx, seeds, [25, 10], agent=agent)
    served, fell_back = on["counts"]
    assert served + fell_back == n and served > 0, on["counts"]
    _same(on, off)


def test_generic_instantiation_is_served_too(grlx, monkeypatch):
    seeds = np.arange(11, 19)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _snapshot(grlx, seeds, [5, 25], force_generic=1)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = _snapshot(grlx, seeds, [5, 25], force_generic=1)
    assert on["counts"][0] > 0
    _same(on, off)


@pytest.mark.parametrize("tune", ["64", "16", "7"])    # no server at all / no load ahead / wave priorities
def test_results_do_not_depend_on_the_server_s_timing(grlx, monkeypatch, tune):
    seeds = np.arange(3, 12)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _snapshot(grlx, seeds, [20, 5])
    monkeypatch.delenv("GRLX_ENV_SERVER")
    monkeypatch.setenv("GRLX_ENV_SERVER_TUNE", tune)
    on = _snapshot(grlx, seeds, [20, 5])
    if tune == "64":
        assert on["counts"] == (0, len(seeds)), on["counts"]       # every replica gave up waiting and integrated itself
    else:
        assert on["counts"][0] > 0
    _same(on, off)


def test_served_run_against_the_oracle(grlx):
    seeds = [2, 9, 31]
    trials = 45
    cfg = grlx.pendulum_sarsa_config(len(seeds), max_rows=trials + 1)
    r = grlx.Runner(cfg, np.asarray(seeds))
    r.run(trials); r.sync()
    assert r.env_server_counts()[0] > 0
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(), seed=int(seed))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert np.asarray(rew).view(np.uint64).tolist() == np.asarray([x.reward for x in rows]).view(np.uint64).tolist()
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        assert np.asarray(r.env_state(k)).view(np.uint64).tolist() == np.asarray(e.state()).view(np.uint64).tolist()
    r.close()


def test_not_served_where_it_is_not_built(grlx):
    """Five actions, the acrobot, the 8-replicas-per-wave layout: launches without the server."""
    cfg = grlx.pendulum_sarsa_config(4, action_steps=5, max_rows=12)
    r = grlx.Runner(cfg, np.arange(1, 5)); r.run(10); r.sync()
    assert r.env_server_counts() == (0, 0)
    r.close()
    cfg = grlx.acrobot_q_config(4, max_rows=12)
    r = grlx.Runner(cfg, np.arange(1, 5)); r.run(5); r.sync()
    assert r.env_server_counts() == (0, 0)
    r.close()


def test_every_replica_is_served_at_the_bench_size(grlx):
    """4096 replicas = one rollout wave AND one server wave per SIMD: the two kernels must fit a SIMD's register file together
    (416 + 96 of 512, grlx_rollout.h / grlx_env_server.h), or the server's waves wait for the rollout waves to finish and every
    replica falls back -- same results, none of the speed."""
    n = 4096
    cfg = grlx.pendulum_sarsa_config(n, max_rows=16)
    r = grlx.Runner(cfg, np.arange(1, n + 1))
    for _ in range(3):
        r.run(11)
    r.sync()
    assert r.last_kernel() == 2
    served, fell_back = r.env_server_counts()
    assert served + fell_back == n
    assert served >= n - n // 100, (served, fell_back)      # (a replica that waited 400 polls for one answer falls back: rare, and harmless)
    r.close()


def test_full_size_batches_server_on_equals_off(grlx, monkeypatch):
    """BASELINE configs[1] at its full size (4096 replicas) over more trials than the bench times, in the bench's launches of 11: every
    sampled replica's rows, streams, state and weights with the server equal those without it."""
    n, launches = 4096, 28                                   # 308 trials
    picks = [0, 1, 2, 3, 4, 1023, 1024, 2047, 2048, 3000, 4093, 4094, 4095]

    def run():
        cfg = grlx.pendulum_sarsa_config(n, max_rows=64)
        r = grlx.Runner(cfg, np.arange(1, n + 1))
        for _ in range(launches):
            r.run(11)
        r.sync()
        rng = np.random.default_rng(17)
        slots = rng.integers(0, cfg.projector.memory, 3000).astype(np.uint32)
        out = {"counts": r.env_server_counts(), "steps": r.step_counts()}
        for k in picks:
            t, s, rew = r.rows(k)
            out[k] = (list(t), list(s), np.asarray(rew, dtype=np.float64).view(np.uint64).tolist(),
                      np.asarray(r.env_state(k), dtype=np.float64).view(np.uint64).tolist(), list(r.rng(k)),
                      np.asarray(r.weights(k, slots), dtype=np.float64).view(np.uint64).tolist())
        r.close()
        return out

    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = run()
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = run()
    assert off["counts"] == (0, 0) and on["counts"][0] >= n - n // 100, on["counts"]
    assert on["steps"] == off["steps"]                       # learning and test steps summed over ALL replicas
    for k in picks:
        assert on[k] == off[k], f"replica {k}"


@pytest.mark.parametrize("over", [dict(timeout=0.05), dict(timeout=0.1, test_interval=0), dict(timeout=0.31, test_interval=-1, randomization=1.0),
                                  dict(timeout=0.2, test_interval=2, test_trials=3)])
def test_short_episodes_stress_the_start_of_trial_hand_off(grlx, monkeypatch, over):
    """Episodes of 2-11 steps: a RESET command every few passes, the last step of an episode without a command, test trials every trial
    / never / as several episodes -- the corners of the command sequence (grlx_rollout.h: mail_send_reset, the terminal pass)."""
    seeds = np.arange(21, 30)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _snapshot(grlx, seeds, [40, 23], **over)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = _snapshot(grlx, seeds, [40, 23], **over)
    assert on["counts"][0] > 0
    _same(on, off)


# ------------------------------------------------------------------------------------------------------------------------------------
# The environment server of the WIDE kernels (grl_amd/csrc/grlx_env_server_wide.h): acrobot and compass walker, 8 replicas per wave, one
# step of look-ahead for all three actions, a command per replica and pass.

def _wide_config(grlx, env, n, **over):
    make = {"acrobot": grlx.acrobot_q_config, "walker": grlx.compass_walker_q_config}[env]
    cfg = make(n, **over)
    cfg.replicas_per_wave = 8
    return cfg


@pytest.fixture(autouse=True)
def _walker_server_on(monkeypatch):
    """The walker's server is opt-in (slower than the rollout wave alone: DESIGN.md 4.1h); these tests exercise it."""
    monkeypatch.setenv("GRLX_ENV_SERVER_WALKER", "1")


def _wide_snapshot(grlx, env, seeds, chunks, budget=0, **over):
    cfg = _wide_config(grlx, env, len(seeds), max_rows=sum(chunks) + 8, **over)
    r = grlx.Runner(cfg, seeds)
    for k, c in enumerate(chunks):
        if budget:
            r.run_steps(c, budget * (k + 1))
        else:
            r.run(c)
    r.sync()
    rng = np.random.default_rng(5)
    slots = rng.integers(0, cfg.projector.memory, 3000).astype(np.uint32)
    out = {"counts": r.env_server_counts(), "kernel": r.last_kernel(), "rpw": r.replicas_per_wave(), "rows": [], "state": [], "rng": [], "w": [], "load": []}
    for k in range(len(seeds)):
        t, s, rew = r.rows(k)
        out["rows"].append((list(t), list(s), np.asarray(rew, dtype=np.float64).view(np.uint64).tolist(),
                            np.asarray(r.row_times(k, 0, len(t)), dtype=np.float64).view(np.uint64).tolist()))
        out["state"].append(np.asarray(r.env_state(k), dtype=np.float64).view(np.uint64).tolist())
        out["rng"].append(list(r.rng(k)))
        out["w"].append(np.asarray(r.weights(k, slots), dtype=np.float64).view(np.uint64).tolist())
        out["load"].append(r.table_load(k))
    r.close()
    return out


def _same_wide(a, b):
    for key in ("rows", "state", "rng", "w", "load"):
        assert a[key] == b[key], key


@pytest.mark.parametrize("env,chunks", [("acrobot", [30, 14]), ("walker", [9, 5])])
@pytest.mark.parametrize("n", [8, 13, 37])
def test_wide_server_on_equals_server_off(grlx, monkeypatch, env, chunks, n):
    """Full waves, a ragged last wave (13 = 8 + 5: a half-empty sub-batch and a dead 16-lane group), five waves; episodes of one wave end at
    different steps (absorbing states): trial starts, idle passes and steps of different replicas share one pass of the server."""
    seeds = np.arange(1, n + 1)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _wide_snapshot(grlx, env, seeds, chunks)
    assert off["counts"] == (0, 0) and off["rpw"] == 8
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = _wide_snapshot(grlx, env, seeds, chunks)
    served, fell_back = on["counts"]
    assert served + fell_back == n and served > 0, on["counts"]
    assert on["kernel"] == 2                                   # the specialised instantiations (SpecAcrobotQ / SpecWalkerQ)
    _same_wide(on, off)


@pytest.mark.parametrize("env", ["acrobot", "walker"])
@pytest.mark.parametrize("tune", ["64", "7"])                  # no server at all (every replica falls back at its first step) / wave priorities
def test_wide_results_do_not_depend_on_the_server_s_timing(grlx, monkeypatch, env, tune):
    seeds = np.arange(3, 14)
    chunks = [12, 5] if env == "acrobot" else [5, 3]
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _wide_snapshot(grlx, env, seeds, chunks)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    monkeypatch.setenv("GRLX_ENV_SERVER_TUNE", tune)
    on = _wide_snapshot(grlx, env, seeds, chunks)
    if tune == "64":
        assert on["counts"] == (0, len(seeds)), on["counts"]
    else:
        assert on["counts"][0] > 0
    _same_wide(on, off)


@pytest.mark.parametrize("env", ["acrobot", "walker"])
def test_wide_served_run_against_the_oracle(grlx, env):
    from tests import configs
    seeds = [2, 9, 31, 32, 33, 34, 35, 36, 40]
    trials = 34 if env == "acrobot" else 12
    make = configs.acrobot if env == "acrobot" else configs.compass_walker
    cfg, spec = make(grlx, len(seeds), max_rows=trials + 1)
    cfg.replicas_per_wave = 8
    r = grlx.Runner(cfg, np.asarray(seeds))
    r.run(trials); r.sync()
    assert r.env_server_counts()[0] > 0
    for k in (0, 4, 8):
        e = ob.Experiment(spec, seed=int(seeds[k]))
        rows, _ = e.run(trials)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert np.asarray(rew).view(np.uint64).tolist() == np.asarray([x.reward for x in rows]).view(np.uint64).tolist()
        assert list(r.rng(k))[:3] == list(e.rng())[:3]
        assert np.asarray(r.env_state(k)).view(np.uint64).tolist() == np.asarray(e.state()).view(np.uint64).tolist()
        e.close()
    r.close()


@pytest.mark.parametrize("env,budget,interval,kernel", [("acrobot", 700, 3, 1), ("acrobot", 700, 10, 2), ("walker", 900, 10, 2)])
def test_wide_server_under_a_steps_budget_and_with_test_trials(grlx, monkeypatch, env, budget, interval, kernel):
    """The launches the bench times: all trials of a call in one launch, bounded by learning steps (replicas finish at trials of their own:
    kMailExit at different passes), two successive budgets; test trials of two greedy episodes.  test_interval is one of the numbers a
    specialised kernel is compiled for: 3 runs the acrobot's generic instantiation beside its server (the walker's generic one has no room
    for a server: test_wide_generic_instantiations), 10 the specialised ones."""
    seeds = np.arange(50, 63)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _wide_snapshot(grlx, env, seeds, [1 << 20, 1 << 20], budget=budget, test_trials=2, test_interval=interval)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = _wide_snapshot(grlx, env, seeds, [1 << 20, 1 << 20], budget=budget, test_trials=2, test_interval=interval)
    assert on["kernel"] == kernel and on["counts"][0] > 0, (on["kernel"], on["counts"])
    _same_wide(on, off)


def test_wide_generic_instantiations(grlx, monkeypatch):
    """Parameters read at run time (force_generic): the acrobot's generic kernel and its server still fit a SIMD together and are served; the
    walker's generic kernel (378 registers) leaves no room for the server's wave, so the launcher leaves the server out -- asked of the runtime
    (hipFuncGetAttributes), never assumed."""
    seeds = np.arange(11, 27)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _wide_snapshot(grlx, "acrobot", seeds, [20], force_generic=1)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    on = _wide_snapshot(grlx, "acrobot", seeds, [20], force_generic=1)
    assert on["kernel"] == 1 and on["counts"][0] > 0, on["counts"]
    _same_wide(on, off)
    w = _wide_snapshot(grlx, "walker", seeds, [4], force_generic=1)
    assert w["kernel"] == 1 and w["counts"] == (0, 0), w["counts"]


@pytest.mark.parametrize("env", ["acrobot", "walker"])
def test_every_wide_replica_is_served_at_the_bench_size(grlx, env):
    """8192 replicas = one wide rollout wave AND one server wave per SIMD (344 + 168 / 336 + 112 of 512 registers)."""
    n = 8192
    cfg = _wide_config(grlx, env, n, max_rows=64, table_log2_capacity=(18 if env == "walker" else 0))
    r = grlx.Runner(cfg, np.arange(1, n + 1))
    r.run_steps(1 << 20, 600); r.sync()
    assert r.last_kernel() == 2 and r.replicas_per_wave() == 8
    served, fell_back = r.env_server_counts()
    assert served + fell_back == n
    # (the walker's server wave runs late on passes with several heel strikes: a few per cent of the replicas go on without it)
    assert served >= (n - n // 100 if env == "acrobot" else n - n // 8), (served, fell_back)
    r.close()


def test_the_walker_s_server_is_opt_in(grlx, monkeypatch):
    monkeypatch.delenv("GRLX_ENV_SERVER_WALKER")
    cfg = _wide_config(grlx, "walker", 16, max_rows=16)
    r = grlx.Runner(cfg, np.arange(1, 17))
    r.run(3); r.sync()
    assert r.last_kernel() == 2 and r.replicas_per_wave() == 8 and r.env_server_counts() == (0, 0)
    r.close()


@pytest.mark.parametrize("agent,quit_after", [(0, 37), (1, 101), (0, 250), (3, 777)])
def test_a_server_that_leaves_in_mid_episode(grlx, monkeypatch, agent, quit_after):
    """The bound of a fetch is there for a server that is late or gone; `GRLX_ENV_SERVER_TUNE` bits 8-23 make the server leave, unannounced, after
    it has answered that many commands of a replica: in the MIDDLE of an episode (37: the first; 101: exactly at a trial boundary, the reset is
    answered and nothing after it; 250: the third trial; 777: the eighth -- every launch counts its commands from 1, so both launches of 25 + 10
    trials lose their server).  The rollout wave's next fetch runs into its bound, the wave integrates that step itself from the state it holds and goes on alone.
    Same bits as with the server off; every replica reports that it fell back."""
    seeds = np.arange(21, 34)
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _snapshot(grlx, seeds, [25, 10], agent=agent)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    monkeypatch.setenv("GRLX_ENV_SERVER_TUNE", str((quit_after << 8) | 3))
    on = _snapshot(grlx, seeds, [25, 10], agent=agent)
    assert on["counts"] == (0, len(seeds)), on["counts"]
    _same(on, off)


@pytest.mark.parametrize("env,quit_after", [("acrobot", 45), ("acrobot", 88), ("walker", 29)])
def test_a_wide_server_that_leaves_in_mid_episode(grlx, monkeypatch, env, quit_after):
    seeds = np.arange(3, 16)
    chunks = [12, 5] if env == "acrobot" else [5, 3]
    monkeypatch.setenv("GRLX_ENV_SERVER", "0")
    off = _wide_snapshot(grlx, env, seeds, chunks)
    monkeypatch.delenv("GRLX_ENV_SERVER")
    monkeypatch.setenv("GRLX_ENV_SERVER_TUNE", str(quit_after << 8))
    on = _wide_snapshot(grlx, env, seeds, chunks)
    assert on["counts"][1] > 0, on["counts"]      # (a replica whose last launch had fewer passes than that was served to its end)
    _same_wide(on, off)
