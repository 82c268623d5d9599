"""Target networks on the Q table (representation/parameterized/linear: interval, tau; representation.h:161-306):
rollout_tgt_kernel against the oracle -- per-step records, rows, RNG positions, the main table, the target table and
the number of synchronisations.  The reference ships no test for them: parity unpinned by reference tests."""
import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    bad = np.nonzero(bits(a).ravel() != bits(b).ravel())[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}"


@pytest.mark.parametrize("env,agent,interval,tau,trace,memory", [
    ("pendulum", 0, 1000, 1.0, 1, 8388608),       # hard update every 1000 update() calls (~ every 100 steps)
    ("pendulum", 1, 137, 0.0, 1, 8388608),        # tau = 0: setParams(params()); a synchronisation in the middle of most steps' trace updates
    ("pendulum", 0, 500, 0.25, 1, 8388608),       # Polyak averaging: untouched slots follow the K-fold recurrence from the target's own draw
    ("pendulum", 1, 64, 0.5, 0, 8388608),         # no trace: one update() call per step
    ("pendulum", 0, 300, 0.3, 1, 2048),           # tiny hash memory: nearly every slot shared between tilings
    ("acrobot", 1, 200, 0.1, 1, 8388608),
    ("cart_pole", 1, 250, 0.2, 1, 8388608),       # round 3: the other two environments of the path
    ("cart_pole", 0, 400, 0.0, 1, 8388608),
    ("compass_walker", 1, 150, 0.3, 1, 8388608),
])
def test_target_network_bit_exact(grlx, env, agent, interval, tau, trace, memory):
    from tests import configs
    from tests.test_gpu_parity import _compare_taps
    make = {"pendulum": configs.pendulum, "acrobot": configs.acrobot, "cart_pole": configs.cart_pole_q, "compass_walker": configs.compass_walker}[env]
    seeds, trials, cap = [71, 72, 73, 74, 75, 76, 77], 24, 2600
    cfg, spec = make(grlx, len(seeds), agent=agent, tap_replica=5, tap_capacity=cap)
    for obj in (cfg, spec):
        obj.target_interval, obj.target_tau, obj.trace = interval, tau, trace
        obj.projector.memory = memory
    r = grlx.Runner(cfg, seeds)
    r.run(7); r.run(9); r.run(8); r.sync()                    # three launches: count_ and the synchronisation number persist
    rng = np.random.default_rng(23)
    slots = np.unique(np.concatenate([rng.integers(0, memory, 1500), np.arange(0, min(memory, 4096))])).astype(np.uint32)
    D = {"pendulum": 2, "acrobot": 4, "cart_pole": 4, "compass_walker": 5}[env]
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=int(seed))
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3], f"RNG positions of replica {k}"
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        assert_bit_equal(r.weights(k, slots), e.weights(slots), f"main table of replica {k}")
        tw, syncs = r.target_weights(k, slots)
        assert syncs == e.L.orc_target_syncs(e.h) and syncs > 0, (syncs, e.L.orc_target_syncs(e.h))
        assert_bit_equal(tw, e.weights(slots, table=2), f"target table of replica {k}")
        if k == 5:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) and len(otaps) > 100
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    _compare_taps(gt, ot, A=3, D=D)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
        e.close()
    r.close()


def test_target_network_validation(grlx):
    capi = grlx.capi
    for over in (dict(target_interval=-1), dict(target_interval=10, target_tau=1.5), dict(target_interval=10, agent=3),
                 dict(target_interval=10, trace=2), dict(target_interval=10, action_steps=5)):
        with pytest.raises(capi.GrlxError) as ei:
            grlx.Runner(grlx.pendulum_sarsa_config(1, **over), [1])
        assert ei.value.code == capi.ERR_INVALID
    r = grlx.Runner(grlx.pendulum_sarsa_config(2), [1, 2])
    with pytest.raises(capi.GrlxError):
        r.target_weights(0, [1, 2, 3])                      # no target network in this context
    r.close()


def test_deployer_target_network(grlx, tmp_path):
    """grlxd on the reference's golden yaml with `interval: 700` / `tau: 0.4` on the policy's representation (the yaml keys of
    ParameterizedRepresentation, representation.h:173-176): rows = the oracle's."""
    import os
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    old = "        output_max: [  ]\n        type: representation/parameterized/linear"
    assert old in text
    y = tmp_path / "tgt.yaml"
    y.write_text(text.replace(old, "        output_max: [  ]\n        interval: 700\n        tau: 0.4\n        type: representation/parameterized/linear")
                 .replace("trials: 2000", "trials: 33"))
    res = subprocess.run([grlxd, "-s", "5", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.Experiment(ob.pendulum_sarsa_spec(target_interval=700, target_tau=0.4), seed=5)
    rows, _ = e.run(33)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)
    assert e.L.orc_target_syncs(e.h) > 10


# ---------------------------------------------------------------- projector/tile_coding: safe = 1 ---
@pytest.mark.parametrize("env,agent,memory,trace,target,safe", [
    ("pendulum", 0, 8388608, 1, 0, 1),     # SARSA: p, project(obs, action) and the critique's projection all claim
    ("pendulum", 1, 8388608, 1, 0, 1),     # Q-learning: p and the critique claim, the max runs over unclaimed batch projections
    ("pendulum", 0, 32768, 1, 0, 1),       # a memory of the size of the visited set: slots are contested, projections walk on
    ("pendulum", 1, 32768, 0, 0, 1),
    ("acrobot", 0, 65536, 1, 0, 1),
    ("pendulum", 0, 32768, 1, 400, 1),     # claim table AND target network
    ("pendulum", 0, 32768, 1, 0, 2),       # safe = 2 (claim always): the policy's batch projections claim too, variant after variant
    ("pendulum", 1, 32768, 1, 0, 2),       # ... and so do Q-learning's second batch projections inside criticize
    ("pendulum", 1, 8388608, 0, 0, 2),
    ("acrobot", 1, 65536, 1, 0, 2),
    ("cart_pole", 1, 65536, 1, 0, 1),      # round 3: the other two environments of the path
    ("compass_walker", 1, 65536, 1, 0, 2),
    ("pendulum", 1, 32768, 1, 300, 2),
])
def test_safe_tile_coding_bit_exact(grlx, env, agent, memory, trace, target, safe):
    """projector/tile_coding with safe = 1 (tile_coding.h:116-151): slots claimed by the hash sum of single projections,
    linear probing past slots claimed by another hash sum, claims of one projection made in tiling order.  Per-step records
    (the slot indices in them are the CLAIMED locations), rows, RNG, weights.  parity unpinned by reference tests."""
    from tests import configs
    from tests.test_gpu_parity import _compare_taps
    make = {"pendulum": configs.pendulum, "acrobot": configs.acrobot, "cart_pole": configs.cart_pole_q, "compass_walker": configs.compass_walker}[env]
    seeds, trials, cap = [81, 82, 83, 84, 85, 86], 24, 2600
    cfg, spec = make(grlx, len(seeds), agent=agent, tap_replica=3, tap_capacity=cap)
    for obj in (cfg, spec):
        obj.trace = trace
        obj.projector.memory = memory
        obj.target_interval, obj.target_tau = target, 0.3
    cfg.projector.safe = safe
    spec.safe = safe
    r = grlx.Runner(cfg, seeds)
    r.run(7); r.run(9); r.run(8); r.sync()
    rng = np.random.default_rng(29)
    slots = np.unique(np.concatenate([rng.integers(0, memory, 1500), np.arange(0, 4096)])).astype(np.uint32)
    D = {"pendulum": 2, "acrobot": 4, "cart_pole": 4, "compass_walker": 5}[env]
    for k, seed in enumerate(seeds):
        e = ob.Experiment(spec, seed=int(seed))
        rows, otaps = e.run(trials, tap_cap=cap)
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows], f"replica {k}"
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of replica {k}")
        assert list(r.rng(k))[:3] == list(e.rng())[:3], f"RNG positions of replica {k}"
        assert_bit_equal(r.env_state(k), e.state(), f"env state of replica {k}")
        assert_bit_equal(r.weights(k, slots), e.weights(slots), f"main table of replica {k}")
        if memory < 100000 and k == 0 and env == "pendulum":
            # the claim table matters here: the same experiment WITHOUT it takes another course
            spec0 = make(None, 1, agent=agent)[1]
            spec0.trace, spec0.projector.memory = trace, memory
            spec0.target_interval, spec0.target_tau = target, 0.3
            e0 = ob.Experiment(spec0, seed=int(seed))
            rows0, _ = e0.run(trials)
            assert [x.reward for x in rows0] != [x.reward for x in rows]
            e0.close()
        if target:
            tw, syncs = r.target_weights(k, slots)
            assert syncs == e.L.orc_target_syncs(e.h) and syncs > 0
            assert_bit_equal(tw, e.weights(slots, table=2), f"target table of replica {k}")
        if k == 3:
            gtaps = r.taps()
            assert len(gtaps) == len(otaps) and len(otaps) > 100
            for i, (gt, ot) in enumerate(zip(gtaps, otaps)):
                try:
                    _compare_taps(gt, ot, A=3, D=D)
                except AssertionError as ex:
                    raise AssertionError(f"step {i}: {ex}")
        e.close()
    r.close()


def test_safe_and_project_operator(grlx):
    """grlx_project is stateless: it refuses a spec with safe != 0; safe = 3 does not exist."""
    capi = grlx.capi
    spec = grlx.pendulum_sarsa_config(1).projector
    spec.safe = 1
    with pytest.raises(capi.GrlxError) as ei:
        grlx.runner.project(spec, [[0.1, 0.2, 0.0]])
    assert ei.value.code == capi.ERR_INVALID
    cfg = grlx.pendulum_sarsa_config(1)
    cfg.projector.safe = 3
    with pytest.raises(capi.GrlxError) as ei:
        grlx.Runner(cfg, [1])
    assert ei.value.code == capi.ERR_INVALID


def test_deployer_safe_tile_coding(grlx, tmp_path):
    """grlxd on the reference's golden yaml with `safe: 1` on the projector (and a 65536-slot memory, so that claims are
    contested): rows = the oracle's."""
    import os
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    text = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml")).read()
    assert "safe: 0" in text and "memory: 8388608" in text
    y = tmp_path / "safe.yaml"
    y.write_text(text.replace("safe: 0", "safe: 1").replace("memory: 8388608", "memory: 65536").replace("trials: 2000", "trials: 33"))
    res = subprocess.run([grlxd, "-s", "7", "-l", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    spec = ob.pendulum_sarsa_spec(safe=1)
    spec.projector.memory = 65536
    e = ob.Experiment(spec, seed=7)
    rows, _ = e.run(33)
    assert (tmp_path / "pendulum-sarsa-tc-0.txt").read_text() == e.format_rows(rows)


@pytest.mark.parametrize("agent,tau", [(0, 0.3), (1, 0.0), (1, 1.0)])
def test_load_into_a_representation_with_a_target_network(grlx, agent, tau):
    """Round 4: ParameterizedRepresentation {action: load} with a target network (representation.h:231-263): setParams(image) is followed
    by synchronize() -- target <- tau * image + (1 - tau) * target over the WHOLE parameter vector, created slots and untouched ones alike
    (tau = 0: the image).  A policy trained elsewhere is loaded into replicas 1 and 2 in mid-run, TWICE (the second load blends into what
    the first one and 12 more trials left); replica 0 is not touched.  Rows, streams, the main table, the target table at 6000 sampled
    slots and the number of synchronisations against the oracle's set_weights() at the same points."""
    src = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent), seed=77)
    src.run(25)
    image = src.all_weights()
    image2 = image[::-1].copy()
    seeds = [31, 32, 33]
    cfg = grlx.pendulum_sarsa_config(len(seeds), agent=agent, target_interval=190, target_tau=tau)
    r = grlx.Runner(cfg, seeds)
    r.run(9); r.sync()
    r.load_weights(image, first_replica=1, n_replicas=2)
    r.run(12); r.sync()
    r.load_weights(image2, first_replica=1, n_replicas=2)
    r.run(8); r.sync()
    slots = np.random.default_rng(9).integers(0, 8388608, 6000).astype(np.uint32)
    for k, seed in enumerate(seeds):
        e = ob.Experiment(ob.pendulum_sarsa_spec(agent=agent, target_interval=190, target_tau=tau), seed=seed)
        rows = list(e.run(9)[0])
        if k >= 1:
            e.set_weights(image)
        rows += list(e.run(12)[0])
        if k >= 1:
            e.set_weights(image2)
        rows += list(e.run(8)[0])
        t, s, rew = r.rows(k)
        assert list(s) == [x.steps for x in rows]
        assert_bit_equal(rew, [x.reward for x in rows], f"returns of seed {seed}")
        assert list(r.rng(k))[:4] == list(e.rng())[:4]
        assert_bit_equal(r.weights(k, slots), e.weights(slots), f"main table of replica {k}")
        tw, syncs = r.target_weights(k, slots)
        assert syncs == e.L.orc_target_syncs(e.h)
        assert_bit_equal(tw, e.weights(slots, table=2), f"target table of replica {k}")
        e.close()
    r.close()
