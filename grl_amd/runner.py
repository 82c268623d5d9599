"""Thin Python host over the C ABI: one `Runner` = one grlx context = the
replicas of one GPU.  All computation happens in the HIP kernels behind
include/grlx.h; this file only marshals arguments."""
import ctypes as C
import math

import numpy as np

from . import capi


def pendulum_sarsa_config(n_replicas=1, **overrides) -> capi.Config:
    """Config of the reference's tests/pendulum-sarsa-tc.yaml (cfg/pendulum/sarsa_tc.yaml semantics)."""
    lib = capi.load()
    cfg = capi.Config()
    lib.grlx_config_pendulum_sarsa(C.byref(cfg))
    cfg.n_replicas = n_replicas
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def cart_pole_ac_config(n_replicas=1, **overrides) -> capi.Config:
    """Config of the reference's cfg/cart_pole/ac_tc.yaml (actor-critic over two tile-coded tables)."""
    lib = capi.load()
    cfg = capi.Config()
    lib.grlx_config_cart_pole_ac(C.byref(cfg))
    cfg.n_replicas = n_replicas
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def _set_tile(ts, tilings, memory, resolution, wrapping):
    ts.tilings, ts.memory, ts.dims = tilings, memory, len(resolution)
    for i in range(capi.MAX_DIMS):
        ts.resolution[i] = resolution[i] if i < len(resolution) else 0.0
        ts.wrapping[i] = wrapping[i] if i < len(wrapping) else 0.0


def acrobot_q_config(n_replicas=1, agent=capi.AGENT_Q, **overrides) -> capi.Config:
    """dynamics/acrobot + task/acrobot/balancing (acrobot.cpp) under the TD agent block of the reference's
    cfg/pendulum/q_tc.yaml -- the reference ships no TD yaml for the acrobot: 3 torques
    over [-1, 1], control step 0.05 s, tile resolution [0.05, 0.05, 0.2, 0.4 | 1.0]."""
    cfg = pendulum_sarsa_config(n_replicas, agent=agent, **overrides)
    cfg.env = capi.ENV_ACROBOT
    cfg.control_step, cfg.integration_steps, cfg.timeout = 0.05, 5, 20.0
    cfg.action_min, cfg.action_max, cfg.action_steps = -1.0, 1.0, 3
    _set_tile(cfg.projector, 16, 8388608, [0.05, 0.05, 0.2, 0.4, 1.0], [0] * 5)
    return cfg


def compass_walker_q_config(n_replicas=1, agent=capi.AGENT_Q, **overrides) -> capi.Config:
    """The reference's cfg/compass_walker/qlearning_walk.yaml: model/compass_walker + task/compass_walker/walk,
    Q-learning over a 6-input tile coding, 3 hip torques over [-1.2, 1.2], control step 0.2 s (20 sub-steps)."""
    cfg = pendulum_sarsa_config(n_replicas, agent=agent, **overrides)
    cfg.env = capi.ENV_COMPASS_WALKER
    cfg.control_step, cfg.integration_steps, cfg.timeout = 0.2, 20, 100.0
    cfg.slope_angle, cfg.initial_state_variation, cfg.negative_reward = 0.004, 0.2, -100.0
    cfg.action_min, cfg.action_max, cfg.action_steps = -1.2, 1.2, 3
    _set_tile(cfg.projector, 16, 8388608, [0.0838, 0.1047, 0.1111, 0.2222, 10, 1.2], [0] * 6)
    return cfg


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class Runner:
    """N independent-seed replicas of one experiment graph on the current GPU
    (the reference's experiment/multi clones, base/src/experiments/multi.cpp:44-75)."""

    def __init__(self, cfg: capi.Config, seeds):
        self.lib = capi.load()
        self.cfg = cfg
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        if seeds.shape != (cfg.n_replicas,):
            raise ValueError("need one seed per replica")
        self.seeds = seeds
        self._ctx = C.c_void_p()
        capi.check(self.lib.grlx_create(C.byref(cfg), _ptr(seeds, C.c_int64), C.byref(self._ctx)))
        if cfg.env == capi.ENV_EXTERNAL:        # the caller's environment: the agent's observation = the projector's input (minus the action)
            self.state_dims, self.obs_dims = 0, cfg.projector.dims - (0 if cfg.agent == capi.AGENT_AC else 1)
        else:
            sd, od = C.c_int(), C.c_int()
            capi.check(self.lib.grlx_env_dims(cfg.env, C.byref(sd), C.byref(od)))
            self.state_dims, self.obs_dims = sd.value, od.value

    def close(self):
        if self._ctx:
            self.lib.grlx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # OnlineLearningExperiment::run for n_trials further trials (async on `stream`)
    def run(self, n_trials: int, stream: int = 0):
        capi.check(self.lib.grlx_run(self._ctx, int(n_trials), C.c_void_p(stream)))

    def run_steps(self, max_trials: int, steps: int, stream: int = 0):
        """The trial loop with both bounds (online_learning.cpp:154): at most max_trials further trials per replica, none started once
        the replica's learning steps of the run have reached `steps`."""
        capi.check(self.lib.grlx_run_steps(self._ctx, int(max_trials), int(steps), C.c_void_p(stream)))

    def sync(self, stream: int = 0):
        capi.check(self.lib.grlx_sync(self._ctx, C.c_void_p(stream)))

    def replica_rows(self, replica: int) -> int:
        return capi.check(self.lib.grlx_replica_rows(self._ctx, int(replica)))

    def set_diag(self, enable: bool = True):
        capi.check(self.lib.grlx_set_diag(self._ctx, int(enable)))

    def read_diag(self):
        waves = (self.cfg.n_replicas + 3) // 4
        out = np.zeros((waves, 8), np.uint64)
        n = C.c_int()
        capi.check(self.lib.grlx_read_diag(self._ctx, _ptr(out, C.c_uint64), waves, C.byref(n)))
        return out[: n.value]

    def n_rows(self) -> int:
        return capi.check(self.lib.grlx_rows(self._ctx))

    def rows(self, replica: int, first: int = 0, count: int = None):
        if count is None:
            count = self.replica_rows(replica) - first
        trial = np.zeros(count, np.int64)
        steps = np.zeros(count, np.int64)
        reward = np.zeros(count, np.float64)
        capi.check(self.lib.grlx_read_rows(self._ctx, replica, first, count, _ptr(trial, C.c_int64),
                                           _ptr(steps, C.c_int64), _ptr(reward, C.c_double)))
        return trial, steps, reward

    def last_kernel(self) -> int:
        """capi GRLX_KERNEL_*: 1 generic, 2 specialised (compile-time instantiation), 3 diagnostic in-place."""
        return self.lib.grlx_last_kernel(self._ctx)

    def env_server_counts(self):
        """(replicas served by the environment server to the end of the last launch that had it, replicas that fell back to integrating
        themselves); (0, 0) when no launch of this context had it -- diagnostic export, not part of include/grlx.h"""
        a, b = C.c_int(0), C.c_int(0)
        capi.check(self.lib.grlx_env_server_counts(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def replicas_per_wave(self) -> int:
        """4: one replica per 16 lanes; 8: two sub-batches per wave sharing the environment phase (wide kernels); 12 / 16: three / four
        sub-batches (actor-critic only; 12 rotates the wave's own replicas through its slots trial by trial)."""
        return self.lib.grlx_replicas_per_wave(self._ctx)

    def row_times(self, replica: int, first: int = 0, count: int = None):
        """Episode time of each row's trial (column 4, online_learning.cpp:243): steps under discrete_time."""
        if count is None:
            count = self.n_rows() - first
        out = np.zeros(count, np.float64)
        capi.check(self.lib.grlx_read_row_times(self._ctx, replica, first, count, _ptr(out, C.c_double)))
        return out

    def curve_stats(self, out_dev_ptr: int, first: int, count: int, stream: int = 0):
        capi.check(self.lib.grlx_curve_stats(self._ctx, first, count, C.c_void_p(out_dev_ptr), C.c_void_p(stream)))

    def step_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        capi.check(self.lib.grlx_step_counts(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def env_state(self, replica: int):
        st = np.zeros(capi.MAX_STATE, np.float64)
        capi.check(self.lib.grlx_get_env_state(self._ctx, replica, _ptr(st, C.c_double)))
        return st[: self.state_dims]

    def rng(self, replica: int):
        out = np.zeros(4, np.uint64)
        capi.check(self.lib.grlx_get_rng(self._ctx, replica, _ptr(out, C.c_uint64)))
        return out

    def weights(self, replica: int, slots, table: int = 0):
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.zeros(slots.size, np.float64)
        capi.check(self.lib.grlx_get_weights(self._ctx, table, replica, _ptr(slots, C.c_uint32), slots.size, _ptr(out, C.c_double)))
        return out

    def target_weights(self, replica: int, slots):
        """Values of the Q table's target network (representation `interval` / `tau`) and the synchronisations so far."""
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.zeros(slots.size, np.float64)
        n = C.c_uint32()
        capi.check(self.lib.grlx_get_target_weights(self._ctx, replica, _ptr(slots, C.c_uint32), slots.size, _ptr(out, C.c_double), C.byref(n)))
        return out, n.value

    def export_weights(self, replica: int, table: int = 0):
        """Dense double[memory] parameter vector, as grl's .dat files hold it (representation.h:201-229)."""
        t = self.cfg.actor_projector if table == 1 else self.cfg.projector
        out = np.zeros(t.memory, np.float64)
        capi.check(self.lib.grlx_export_weights(self._ctx, table, replica, _ptr(out, C.c_double)))
        return out

    def load_weights(self, dense, table: int = 0, first_replica: int = 0, n_replicas=None):
        """ParameterizedRepresentation {action: load} (representation.h:231-263): every weight of `table`
        of the given replicas becomes dense[slot]; nothing else of the experiment changes."""
        dense = np.ascontiguousarray(dense, dtype=np.float64)
        if n_replicas is None:
            n_replicas = self.cfg.n_replicas - first_replica
        capi.check(self.lib.grlx_load_weights(self._ctx, table, first_replica, n_replicas, _ptr(dense, C.c_double), dense.size))

    def table_load(self, replica: int, table: int = 0) -> int:
        n = C.c_uint32()
        capi.check(self.lib.grlx_table_load(self._ctx, table, replica, C.byref(n)))
        return n.value

    def table_capacity(self) -> int:
        """log2 of the entries per replica and table (the sparse tables grow between launches)."""
        n = C.c_uint32(0)
        capi.check(self.lib.grlx_table_capacity(self._ctx, C.byref(n)))
        return int(n.value)

    def reset_run(self):
        """Experiment::reset() between two runs (online_learning.cpp:307-308): see grlx_reset_run."""
        capi.check(self.lib.grlx_reset_run(self._ctx))

    def grow_tables(self, new_log2: int):
        capi.check(self.lib.grlx_grow_tables(self._ctx, new_log2))

    def taps(self):
        cap = max(int(self.cfg.tap_capacity), 1)
        buf = (capi.Tap * cap)()
        n = C.c_int()
        capi.check(self.lib.grlx_read_taps(self._ctx, buf, cap, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    # ---- the per-step plug-in interfaces: one call of the reference's interface for every replica (include/grlx.h) ----
    def _active(self, active):
        if active is None:
            return None, None
        a = np.ascontiguousarray(active, dtype=np.int32)
        if a.shape != (self.cfg.n_replicas,):
            raise ValueError("active: one entry per replica")
        return a, _ptr(a, C.c_int32)

    def _rows(self, x, cols=None):
        shape = (self.cfg.n_replicas,) if cols is None else (self.cfg.n_replicas, cols)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != shape:
            raise ValueError(f"expected an array of shape {shape}, got {x.shape}")
        return x

    def env_start(self, test: int, active=None, obs=None):
        """Environment::start (environment.h:48) for the replicas: first observations [n_replicas][obs_dims]."""
        obs = np.zeros((self.cfg.n_replicas, self.obs_dims), np.float64) if obs is None else self._rows(obs, self.obs_dims)
        keep, ap = self._active(active)
        capi.check(self.lib.grlx_env_start(self._ctx, int(test), ap, _ptr(obs, C.c_double)))
        return obs

    def env_advance(self, action, active=None, obs=None, reward=None, terminal=None):
        """Environment::step (environment.h:49-51) on the replicas' model states: (obs, reward, terminal); tau = 1."""
        n = self.cfg.n_replicas
        action = self._rows(action)
        obs = np.zeros((n, self.obs_dims), np.float64) if obs is None else self._rows(obs, self.obs_dims)
        reward = np.zeros(n, np.float64) if reward is None else self._rows(reward)
        terminal = np.zeros(n, np.int32) if terminal is None else np.ascontiguousarray(terminal, dtype=np.int32)
        keep, ap = self._active(active)
        capi.check(self.lib.grlx_env_advance(self._ctx, ap, _ptr(action, C.c_double), _ptr(obs, C.c_double), _ptr(reward, C.c_double), _ptr(terminal, C.c_int32)))
        return obs, reward, terminal

    def agent_start(self, test: int, obs, active=None, action=None):
        """Agent::start (agent.h:44-47): test = 0 the learning agent, 1 the test agent; returns the actions [n_replicas]."""
        obs = self._rows(obs, np.asarray(obs).shape[-1])
        action = np.zeros(self.cfg.n_replicas, np.float64) if action is None else self._rows(action)
        keep, ap = self._active(active)
        capi.check(self.lib.grlx_agent_start(self._ctx, int(test), ap, _ptr(obs, C.c_double), _ptr(action, C.c_double)))
        return action

    def agent_step(self, test: int, obs, reward, terminal=None, active=None, action=None, tau: float = 1.0):
        """Agent::step (agent.h:49-52); replicas whose `terminal` entry is 2 get Agent::end instead and keep their `action` row."""
        obs = self._rows(obs, np.asarray(obs).shape[-1])
        reward = self._rows(reward)
        action = np.zeros(self.cfg.n_replicas, np.float64) if action is None else self._rows(action)
        tp = None
        if terminal is not None:
            terminal = np.ascontiguousarray(terminal, dtype=np.int32)
            tp = _ptr(terminal, C.c_int32)
        keep, ap = self._active(active)
        capi.check(self.lib.grlx_agent_step(self._ctx, int(test), ap, float(tau), _ptr(obs, C.c_double), _ptr(reward, C.c_double), tp, _ptr(action, C.c_double)))
        return action

    def agent_end(self, test: int, obs, reward, active=None, tau: float = 1.0):
        """Agent::end (agent.h:54-56): the transition into an absorbing state."""
        obs = self._rows(obs, np.asarray(obs).shape[-1])
        reward = self._rows(reward)
        keep, ap = self._active(active)
        capi.check(self.lib.grlx_agent_end(self._ctx, int(test), ap, float(tau), _ptr(obs, C.c_double), _ptr(reward, C.c_double)))

    # Representation::read / write / update (batched rows, applied in order)
    def read(self, replica, idx, table: int = 0):
        replica = np.ascontiguousarray(replica, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = np.zeros(replica.size, np.float64)
        capi.check(self.lib.grlx_read(self._ctx, table, _ptr(replica, C.c_int32), _ptr(idx, C.c_uint32), replica.size, _ptr(out, C.c_double)))
        return out

    def write(self, replica, idx, target, alpha: float, table: int = 0):
        replica = np.ascontiguousarray(replica, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        target = np.ascontiguousarray(target, dtype=np.float64)
        capi.check(self.lib.grlx_write(self._ctx, table, _ptr(replica, C.c_int32), _ptr(idx, C.c_uint32), replica.size, _ptr(target, C.c_double), float(alpha)))

    def update(self, replica, idx, delta, table: int = 0):
        replica = np.ascontiguousarray(replica, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        delta = np.ascontiguousarray(delta, dtype=np.float64)
        capi.check(self.lib.grlx_update(self._ctx, table, _ptr(replica, C.c_int32), _ptr(idx, C.c_uint32), replica.size, _ptr(delta, C.c_double)))


def pendulum_fqi_config(n_replicas=1, **overrides) -> capi.FqiConfig:
    """Config of the reference's tests/pendulum-fqi-ann.yaml (experiment/batch_learning + predictor/fqi + ANN)."""
    lib = capi.load()
    cfg = capi.FqiConfig()
    lib.grlx_fqi_config_pendulum(C.byref(cfg))
    cfg.n_replicas = n_replicas
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


class FqiRunner:
    """N independent-seed replicas of the batch-learning experiment (fitted Q-iteration over a 3-H-1 network) on the
    current GPU: include/grlx.h, grlx_fqi_*."""

    def __init__(self, cfg: capi.FqiConfig, seeds):
        self.lib = capi.load()
        self.cfg = cfg
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        if seeds.shape != (cfg.n_replicas,):
            raise ValueError("need one seed per replica")
        self._ctx = C.c_void_p()
        capi.check(self.lib.grlx_fqi_create(C.byref(cfg), _ptr(seeds, C.c_int64), C.byref(self._ctx)))
        self.n_params = 4 * cfg.hidden + cfg.hidden + 1

    def close(self):
        if self._ctx:
            self.lib.grlx_fqi_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_batch(self, stream: int = 0):
        capi.check(self.lib.grlx_fqi_run_batch(self._ctx, C.c_void_p(stream)))

    def sync(self, stream: int = 0):
        capi.check(self.lib.grlx_fqi_sync(self._ctx, C.c_void_p(stream)))

    def rows(self, replica: int, count: int):
        b = np.zeros(count, np.int64); t = np.zeros(count, np.int64); rew = np.zeros(count, np.float64)
        capi.check(self.lib.grlx_fqi_read_rows(self._ctx, replica, 0, count, _ptr(b, C.c_int64), _ptr(t, C.c_int64), _ptr(rew, C.c_double)))
        return b, t, rew

    def params(self, replica: int):
        out = np.zeros(self.n_params, np.float64)
        capi.check(self.lib.grlx_fqi_get_params(self._ctx, replica, _ptr(out, C.c_double), out.size))
        return out

    def transitions(self, replica: int, first: int, count: int):
        inp = np.zeros((count, 3), np.float64); nobs = np.zeros((count, 2), np.float64)
        rew = np.zeros(count, np.float64); tgt = np.zeros(count, np.float64)
        capi.check(self.lib.grlx_fqi_get_transitions(self._ctx, replica, first, count, _ptr(inp, C.c_double), _ptr(nobs, C.c_double),
                                                     _ptr(rew, C.c_double), _ptr(tgt, C.c_double)))
        return inp, nobs, rew, tgt

    def info(self, replica: int):
        n = C.c_int64(); md = C.c_double(); it = C.c_int32(); err = C.c_double(); rng = (C.c_uint64 * 2)()
        capi.check(self.lib.grlx_fqi_info(self._ctx, replica, C.byref(n), C.byref(md), C.byref(it), C.byref(err), rng))
        return dict(n=n.value, maxdelta=md.value, iterations=it.value, error=err.value, rng=[rng[0], rng[1]])


# ---- stateless batched operators --------------------------------------------
def project(spec: capi.TileSpec, x):
    """Projector::project (TileCodingProjector::_project) for a batch of inputs."""
    lib = capi.load()
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, spec.dims)
    out = np.zeros((x.shape[0], spec.tilings), np.uint32)
    capi.check(lib.grlx_project(C.byref(spec), _ptr(x, C.c_double), x.shape[0], _ptr(out, C.c_uint32)))
    return out


def env_step(cfg: capi.Config, state, action):
    """Environment::step (ModeledEnvironment::step) for a batch of (state, action)."""
    lib = capi.load()
    sd, od = C.c_int(), C.c_int()
    capi.check(lib.grlx_env_dims(cfg.env, C.byref(sd), C.byref(od)))
    state = np.array(state, dtype=np.float64).reshape(-1, sd.value)
    action = np.ascontiguousarray(action, dtype=np.float64).reshape(-1)
    n = state.shape[0]
    obs = np.zeros((n, od.value), np.float64)
    reward = np.zeros(n, np.float64)
    terminal = np.zeros(n, np.int32)
    capi.check(lib.grlx_env_step(C.byref(cfg), _ptr(state, C.c_double), _ptr(action, C.c_double), n,
                                 _ptr(obs, C.c_double), _ptr(reward, C.c_double), _ptr(terminal, C.c_int32)))
    return state, obs, reward, terminal


MATH_SIN, MATH_COS, MATH_LOG, MATH_FMOD, MATH_SQRT = 0, 1, 2, 3, 4


def device_math(op: int, x, y=None):
    lib = capi.load()
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    out = np.zeros_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        yp = _ptr(y, C.c_double)
    capi.check(lib.grlx_math(op, _ptr(x, C.c_double), yp, x.size, _ptr(out, C.c_double)))
    return out


def rand48_at(seed: int, skip):
    lib = capi.load()
    skip = np.ascontiguousarray(skip, dtype=np.uint64).reshape(-1)
    out = np.zeros(skip.size, np.float64)
    capi.check(lib.grlx_rand48_at(int(seed), _ptr(skip, C.c_uint64), skip.size, _ptr(out, C.c_double)))
    return out


def format_row(trial: int, steps: int, reward: float) -> str:
    """One row in the layout of the reference's golden files (three setw(15)
    columns, default ostream precision; tests/template/pendulum-sarsa-tc-0.txt)."""
    return "%15d%15d%15s\n" % (trial, steps, "%g" % reward) if not math.isnan(reward) else "%15d%15d%15s\n" % (trial, steps, "nan")
