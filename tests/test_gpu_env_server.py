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
