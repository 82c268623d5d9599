"""The per-step plug-in entry points of the C ABI (grlx_env_start / grlx_env_advance, grlx_agent_start / _step / _end: the reference's
Environment::start / step, environment.h:48-51, and Agent::start / step / end, agent.h:44-56, one call per call of
OnlineLearningExperiment::run, online_learning.cpp:172-213) against the fused run and against the oracle, bit for bit:

  * the GPU's agent beside the ORACLE's environment on the host  == the fused run (rows, streams, weights)
  * the GPU's environment beside the ORACLE's agent on the host  == the oracle's own run (rows, streams, state, weights)
  * both sides on the GPU, one call per step                      == the fused run (graphs whose agent and environment share a stream:
                                                                    actor-critic noise / start draws, the walker's rejection-sampled starts)
  * fused launches and per-step calls mixed on one context        == one fused run
Tolerance: 0 ulp."""
import numpy as np
import pytest

from tests import configs
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    bad = np.nonzero(bits(a).ravel() != bits(b).ravel())[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}"


def is_test(tt, interval):
    return interval >= 0 and tt % (interval + 1) == interval


class HostLoop:
    """OnlineLearningExperiment::run (online_learning.cpp:154-262) for N replicas in lock step, written on the per-step interfaces:
    `env` and `agent` are objects with start / step (/ end) over [N] rows and an `active` mask; replicas leave a trial at their own step."""

    def __init__(self, n, interval, env, agent):
        self.n, self.interval, self.env, self.agent = n, interval, env, agent
        self.tt = 0
        self.ss = np.zeros(n, np.int64)
        self.rows = [[] for _ in range(n)]

    def run(self, trials):
        n = self.n
        for _ in range(trials):
            test = 1 if is_test(self.tt, self.interval) else 0
            active = np.ones(n, np.int32)
            obs = self.env.start(test, active)
            action = self.agent.start(test, obs, active)
            total = np.zeros(n)
            steps = np.zeros(n)
            while active.any():
                obs, reward, terminal = self.env.step(action, active)
                total += np.where(active != 0, reward, 0.0)
                steps += active != 0
                action = self.agent.step(test, obs, reward, terminal, active, action)      # terminal == 2 rows: Agent::end
                if not test:
                    self.ss += active != 0
                active = np.where((active != 0) & (terminal == 0), 1, 0).astype(np.int32)
            if self.interval < 0 or test:
                for k in range(n):
                    trial = self.tt + 1 - (self.tt + 1) // (self.interval + 1) if self.interval >= 0 else self.tt
                    self.rows[k].append((trial, int(self.ss[k]), total[k], steps[k]))
            self.tt += 1


class GpuEnv:
    def __init__(self, runner):
        self.r = runner

    def start(self, test, active):
        return self.r.env_start(test, active)

    def step(self, action, active):
        return self.r.env_advance(action, active)


class GpuAgent:
    def __init__(self, runner):
        self.r = runner

    def start(self, test, obs, active):
        return self.r.agent_start(test, obs, active)

    def step(self, test, obs, reward, terminal, active, action):
        return self.r.agent_step(test, obs, reward, terminal, active, action=action.copy())


class OracleEnv:
    """the environments of N oracle experiments (their own model states and streams), on the host"""

    def __init__(self, exps, obs_dims):
        self.e, self.D = exps, obs_dims

    def start(self, test, active):
        return np.array([e.env_start(test) if a else np.zeros(self.D) for e, a in zip(self.e, active)])

    def step(self, action, active):
        n = len(self.e)
        obs = np.zeros((n, self.D)); reward = np.zeros(n); terminal = np.zeros(n, np.int32)
        for k, e in enumerate(self.e):
            if active[k]:
                _, obs[k], reward[k], terminal[k] = e.env_step(action[k])
        return obs, reward, terminal


class OracleAgent:
    """the agents of N oracle experiments, on the host"""

    def __init__(self, exps):
        self.e = exps

    def start(self, test, obs, active):
        return np.array([e.agent_start(test, o) if a else 0.0 for e, o, a in zip(self.e, obs, active)])

    def step(self, test, obs, reward, terminal, active, action):
        out = action.copy()
        for k, e in enumerate(self.e):
            if not active[k]:
                continue
            if terminal[k] == 2:
                e.agent_end(test, 1.0, obs[k], reward[k])
            else:
                out[k] = e.agent_step(test, 1.0, obs[k], reward[k])
        return out


def touched_slots(e, fresh, table=0, cap=6000):
    """slots whose weight differs from its initial draw in oracle `e` (`fresh`: the same experiment before any step)"""
    t = np.nonzero(e.all_weights(table) != fresh.all_weights(table))[0].astype(np.uint32)
    return t[:: max(1, t.size // cap)]


GRAPHS = {
    "pendulum_sarsa": (lambda g, n: configs.pendulum(g, n, agent=0), 22),
    "pendulum_q": (lambda g, n: configs.pendulum(g, n, agent=1), 22),
    "pendulum_expected_sarsa": (lambda g, n: configs.pendulum(g, n, agent=3), 13),
    "acrobot_q": (lambda g, n: configs.acrobot(g, n, agent=1), 33),
    "acrobot_sarsa": (lambda g, n: configs.acrobot(g, n, agent=0), 23),
    "cart_pole_ac": (lambda g, n: configs.cart_pole_ac(g, n), 12),
    "compass_walker_q": (lambda g, n: configs.compass_walker(g, n, agent=1), 12),
}


def make(grlx, name, n, **over):
    cfg, spec = GRAPHS[name][0](grlx, n)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg, spec


def five_actions(cfg, spec):
    for s in (cfg, spec):
        s.action_steps = 5
    return cfg, spec


def rows_of(runner, k):
    t, s, rew = runner.rows(k)
    return [(int(a), int(b), c, d) for a, b, c, d in zip(t, s, rew, runner.row_times(k, 0, len(t)))]


def assert_rows_equal(got, want, what):
    assert [(a, b) for a, b, _, _ in got] == [(a, b) for a, b, _, _ in want], what
    assert_bit_equal([c for _, _, c, _ in got], [c for _, _, c, _ in want], what + ": returns")
    assert_bit_equal([d for _, _, _, d in got], [d for _, _, _, d in want], what + ": episode times")


@pytest.mark.parametrize("name,five", [("pendulum_sarsa", False), ("pendulum_q", True), ("pendulum_expected_sarsa", False), ("acrobot_q", False), ("acrobot_sarsa", False)])
def test_agent_entries_beside_the_oracles_environment_equal_the_fused_run(grlx, name, five):
    """grlx_agent_start / _step / _end drive the GPU's agents; the environment of every replica is the ORACLE's, stepped on the host.
    Rows, the agents' streams (global, sampler) and the weights equal the fused run's; the environments' stream equals it too (the
    oracle's thread-local stream).  Seven replicas: a ragged batch, a dead 16-lane group; the acrobot's episodes end apart
    (Agent::end for some replicas while others step on)."""
    seeds = list(range(3, 10))
    n, trials = len(seeds), GRAPHS[name][1]
    cfg, spec = make(grlx, name, n)
    if five:
        five_actions(cfg, spec)
    fused = grlx.Runner(cfg, seeds)
    fused.run(trials); fused.sync()
    stepped = grlx.Runner(cfg, seeds)
    envs = [ob.Experiment(spec, seed=s) for s in seeds]
    loop = HostLoop(n, cfg.test_interval, OracleEnv(envs, stepped.obs_dims), GpuAgent(stepped))
    loop.run(trials)
    whole = ob.Experiment(spec, seed=seeds[2]); fresh = ob.Experiment(spec, seed=seeds[2])
    want_rows, _ = whole.run(trials)
    slots = touched_slots(whole, fresh)
    assert slots.size > 100
    for k in range(n):
        assert_rows_equal(loop.rows[k], rows_of(fused, k), f"{name} replica {k}")
        g, f = stepped.rng(k), fused.rng(k)
        assert (g[0], g[2]) == (f[0], f[2]), f"replica {k}: global / sampler streams"
        assert envs[k].rng()[1] == f[1], f"replica {k}: the environment's thread-local stream"
        assert_bit_equal(envs[k].state(), fused.env_state(k), f"replica {k}: model state")
        assert_bit_equal(stepped.weights(k, slots), fused.weights(k, slots), f"replica {k}: weights")
        assert stepped.table_load(k) == fused.table_load(k)
    assert_rows_equal(loop.rows[2], [(r.trial, r.steps, r.reward, r.time) for r in want_rows], "against the oracle's own run")
    assert_bit_equal(stepped.weights(2, slots), whole.weights(slots), "weights against the oracle's own run")
    for e in envs + [whole, fresh]:
        e.close()
    fused.close(); stepped.close()


@pytest.mark.parametrize("name", ["pendulum_sarsa", "acrobot_q"])
def test_environment_entries_beside_the_oracles_agent_equal_the_oracles_run(grlx, name):
    """grlx_env_start / grlx_env_advance drive the GPU's environments; the agent of every replica is the ORACLE's, on the host.  Rows,
    weights and streams of those agents equal the oracle's own run (orc_run), and the GPU's model state and thread-local stream
    equal that run's."""
    seeds = [5, 6, 7, 8, 9]
    n, trials = len(seeds), GRAPHS[name][1]
    cfg, spec = make(grlx, name, n)
    gpu = grlx.Runner(cfg, seeds)
    agents = [ob.Experiment(spec, seed=s) for s in seeds]
    loop = HostLoop(n, cfg.test_interval, GpuEnv(gpu), OracleAgent(agents))
    loop.run(trials)
    for k, seed in enumerate(seeds):
        whole = ob.Experiment(spec, seed=seed)
        want_rows, _ = whole.run(trials)
        assert_rows_equal(loop.rows[k], [(r.trial, r.steps, r.reward, r.time) for r in want_rows], f"{name} replica {k}")
        a, w = agents[k].rng(), whole.rng()
        assert (a[0], a[2]) == (w[0], w[2]), "the agent's streams"
        assert gpu.rng(k)[1] == w[1], "the environment's thread-local stream (on the GPU)"
        assert_bit_equal(gpu.env_state(k), whole.state(), "model state (on the GPU)")
        assert (agents[k].all_weights() == whole.all_weights()).all()
        whole.close()
    for e in agents:
        e.close()
    gpu.close()


@pytest.mark.parametrize("name,over", [("cart_pole_ac", {}), ("cart_pole_ac", {"different_tiles": True}), ("compass_walker_q", {}), ("pendulum_sarsa", {})])
def test_both_sides_on_the_gpu_one_call_per_step_equal_the_fused_run(grlx, name, over):
    """Environment AND agent through the per-step entry points (four launches per step): graphs whose two sides share a random stream
    -- the actor-critic's exploration noise and the cart-pole's start draw (thread-local), the walker's rejection-sampled starts and
    the sampler's tie breaks (global) -- cannot be split over two processes' streams, but must still equal the fused run call by call.
    Actor-critic with twin tables (equal tile codings) and with two independent tables; the critic's trace survives the trials."""
    seeds = [11, 12, 13, 14, 15, 16]
    n, trials = len(seeds), GRAPHS[name][1]
    cfg, spec = make(grlx, name, n)
    if over.get("different_tiles"):
        for s in (cfg, spec):
            s.projector.resolution[0] = 1.25
    fused = grlx.Runner(cfg, seeds)
    fused.run(trials); fused.sync()
    stepped = grlx.Runner(cfg, seeds)
    loop = HostLoop(n, cfg.test_interval, GpuEnv(stepped), GpuAgent(stepped))
    loop.run(trials)
    whole = ob.Experiment(spec, seed=seeds[1]); fresh = ob.Experiment(spec, seed=seeds[1])
    want_rows, _ = whole.run(trials)
    tables = (0, 1) if name == "cart_pole_ac" else (0,)
    slots = {t: touched_slots(whole, fresh, t) for t in tables}
    for k in range(n):
        assert_rows_equal(loop.rows[k], rows_of(fused, k), f"{name} replica {k}")
        assert list(stepped.rng(k)) == list(fused.rng(k)), f"replica {k}: streams"
        assert_bit_equal(stepped.env_state(k), fused.env_state(k), f"replica {k}: model state")
        for t in tables:
            assert_bit_equal(stepped.weights(k, slots[t], table=t), fused.weights(k, slots[t], table=t), f"replica {k}: table {t}")
            assert stepped.table_load(k, t) == fused.table_load(k, t)
    assert_rows_equal(loop.rows[1], [(r.trial, r.steps, r.reward, r.time) for r in want_rows], "against the oracle's own run")
    for t in tables:
        assert_bit_equal(stepped.weights(1, slots[t], table=t), whole.weights(slots[t], table=t), f"table {t} against the oracle's own run")
    whole.close(); fresh.close(); fused.close(); stepped.close()


@pytest.mark.parametrize("name", ["pendulum_sarsa", "cart_pole_ac"])
def test_fused_launches_and_per_step_calls_mix_on_one_context(grlx, name):
    """Per-step calls for some trials, grlx_run for the next ones, per-step calls again -- on ONE context: the state each side leaves
    is the state the other continues from (streams, decay, tables, the actor-critic's persisted critic trace, which is never
    cleared).  The trial counter belongs to the loop that runs the trials (the caller's, or the fused kernel's own), so the graph has
    no test trials here (test_interval = -1): every trial is a learning trial whichever side counts it."""
    seeds = [21, 22, 23, 24, 25]
    n = len(seeds)
    cfg, spec = make(grlx, name, n, test_interval=-1)
    spec.test_interval = -1
    a, b, c = 4, 7, 3
    mixed = grlx.Runner(cfg, seeds)
    loop = HostLoop(n, -1, GpuEnv(mixed), GpuAgent(mixed))
    loop.run(a)
    mixed.run(b); mixed.sync()
    loop.run(c)
    tables = (0, 1) if name == "cart_pole_ac" else (0,)
    for k in (0, 3):
        whole = ob.Experiment(spec, seed=seeds[k]); fresh = ob.Experiment(spec, seed=seeds[k])
        want_rows, _ = whole.run(a + b + c)
        ns = 2 if name == "cart_pole_ac" else 3                     # (the actor-critic graph has no samplers: global and thread-local streams)
        assert list(mixed.rng(k))[:ns] == list(whole.rng())[:ns], "streams"
        assert_bit_equal(mixed.env_state(k), whole.state(), "model state")
        for t in tables:
            sl = touched_slots(whole, fresh, t)
            assert_bit_equal(mixed.weights(k, sl, table=t), whole.weights(sl, table=t), f"replica {k}: table {t}")
        # the returns: the caller's trials from the loop, the fused ones from the context's rows
        got = [r[2] for r in loop.rows[k][:a]] + list(mixed.rows(k)[2]) + [r[2] for r in loop.rows[k][a:]]
        assert_bit_equal(got, [r.reward for r in want_rows], f"replica {k}: returns of all {a + b + c} trials")
        whole.close(); fresh.close()
    mixed.close()


@pytest.mark.parametrize("name", ["pendulum_q", "cart_pole_ac"])
def test_a_context_without_an_environment_serves_the_agent_entries(grlx, name):
    """GRLX_ENV_EXTERNAL: the environment is entirely the caller's (here: the oracle's), the context holds the agent only -- what a grl
    graph with a CPU environment (gym, a robot) binds to.  Same agents' streams, same weights as the context that also holds the environment;
    grlx_run and the environment entries refuse such a context.  (Actor-critic: its exploration noise draws from the thread-local stream the
    environment's start draws interleave with in ONE process; with the environment elsewhere the agent's stream is its own, so the
    actor-critic case is compared with an oracle agent driven the same way.)"""
    capi = grlx.capi
    seeds = [31, 32, 33]
    n, trials = len(seeds), 9
    cfg, spec = make(grlx, name, n)
    ext, _ = make(grlx, name, n)
    ext.env = capi.ENV_EXTERNAL
    ext.control_step, ext.timeout, ext.integration_steps = 0.0, 0.0, 0           # not read
    agent = grlx.Runner(ext, seeds)
    with pytest.raises(capi.GrlxError) as ei:
        agent.run(1)
    assert ei.value.code == capi.ERR_INVALID and "no environment" in str(ei.value)
    with pytest.raises(capi.GrlxError):
        agent.env_start(0)
    envs = [ob.Experiment(spec, seed=100 + s) for s in seeds]                      # environments of OTHER processes (their own seeds)
    loop = HostLoop(n, cfg.test_interval, OracleEnv(envs, agent.obs_dims), GpuAgent(agent))
    loop.run(trials)
    # the same environments (same seeds: same trajectories of start states) beside the ORACLE's agents, on the host
    envs2 = [ob.Experiment(spec, seed=100 + s) for s in seeds]
    agents2 = [ob.Experiment(spec, seed=s) for s in seeds]
    loop2 = HostLoop(n, cfg.test_interval, OracleEnv(envs2, agent.obs_dims), OracleAgent(agents2))
    loop2.run(trials)
    tables = (0, 1) if name == "cart_pole_ac" else (0,)
    for k in range(n):
        assert_rows_equal(loop.rows[k], loop2.rows[k], f"{name} replica {k}")
        fresh = ob.Experiment(spec, seed=seeds[k])
        for t in tables:
            sl = touched_slots(agents2[k], fresh, t)
            assert sl.size > 50
            assert_bit_equal(agent.weights(k, sl, table=t), agents2[k].weights(sl, table=t), f"replica {k}: table {t}")
        g, o = agent.rng(k), agents2[k].rng()
        ns = 2 if name == "cart_pole_ac" else 3
        assert list(g)[:ns] == list(o)[:ns], f"replica {k}: the agent's streams"
        fresh.close()
    for e in envs + envs2 + agents2:
        e.close()
    agent.close()


def test_per_step_entries_validate_their_arguments(grlx):
    capi = grlx.capi
    cfg, _ = configs.pendulum(grlx, 2, agent=4)
    cfg.kappa = 0.2
    r = grlx.Runner(cfg, [1, 2])
    obs = r.env_start(0)
    with pytest.raises(capi.GrlxError) as ei:
        r.agent_start(0, obs)
    assert ei.value.code == capi.ERR_INVALID and "per-step agent entry points" in str(ei.value)
    r.close()
    cfg, _ = configs.pendulum(grlx, 2)
    r = grlx.Runner(cfg, [1, 2])
    obs = r.env_start(0)
    act = r.agent_start(0, obs)
    obs, reward, terminal = r.env_advance(act)
    with pytest.raises(capi.GrlxError) as ei:
        r.agent_step(0, obs, reward, terminal, tau=0.03)
    assert ei.value.code == capi.ERR_INVALID and "tau" in str(ei.value)
    # inactive rows keep the caller's values
    mask = np.array([0, 1], np.int32)
    keep = np.array([123.0, 0.0])
    out = r.agent_step(0, obs, reward, terminal, active=mask, action=keep.copy())
    assert out[0] == 123.0 and out[1] in (-3.0, 0.0, 3.0)
    r.close()
