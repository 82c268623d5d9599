"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product (grl_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

MAX_DIMS, MAX_STATE = 8, 12
MATH_LIBM, MATH_PORTABLE = 0, 1
ENV_PENDULUM, ENV_CART_POLE, ENV_ACROBOT, ENV_COMPASS_WALKER, ENV_CART_POLE_BALANCING = 0, 1, 2, 3, 4
AGENT_SARSA, AGENT_Q, AGENT_AC, AGENT_EXPECTED_SARSA, AGENT_ADVANTAGE, AGENT_QV, AGENT_PID = 0, 1, 2, 3, 4, 5, 6
TRACE_NONE, TRACE_REPLACING, TRACE_ACCUMULATING = 0, 1, 2


class TileSpec(C.Structure):
    _fields_ = [("tilings", C.c_int), ("memory", C.c_int), ("dims", C.c_int),
                ("resolution", C.c_double * MAX_DIMS), ("wrapping", C.c_double * MAX_DIMS)]


class LinearSpec(C.Structure):
    _fields_ = [("init_min", C.c_double), ("init_max", C.c_double), ("output_min", C.c_double),
                ("output_max", C.c_double), ("limit", C.c_int)]


class Spec(C.Structure):
    _fields_ = [("test_interval", C.c_int), ("env", C.c_int), ("control_step", C.c_double),
                ("integration_steps", C.c_int), ("timeout", C.c_double), ("randomization", C.c_double),
                ("end_stop_penalty", C.c_int), ("action_penalty", C.c_int),
                ("slope_angle", C.c_double), ("initial_state_variation", C.c_double), ("negative_reward", C.c_double),
                ("action_min", C.c_double), ("action_max", C.c_double), ("action_steps", C.c_int),
                ("agent", C.c_int), ("projector", TileSpec), ("representation", LinearSpec),
                ("epsilon", C.c_double), ("decay_rate", C.c_double), ("decay_min", C.c_double),
                ("alpha", C.c_double), ("gamma", C.c_double), ("lambda_", C.c_double), ("trace", C.c_int),
                ("actor_projector", TileSpec), ("actor_representation", LinearSpec),
                ("actor_alpha", C.c_double), ("sigma", C.c_double), ("theta", C.c_double),
                ("ac_decay_rate", C.c_double), ("ac_decay_min", C.c_double),
                ("ac_update_method", C.c_int), ("ac_step_limit", C.c_double), ("math", C.c_int), ("tap_starts", C.c_int), ("kappa", C.c_double), ("beta", C.c_double),
                ("pid_p", C.c_double * MAX_DIMS), ("pid_setpoint", C.c_double * MAX_DIMS),
                ("target_interval", C.c_int), ("target_tau", C.c_double), ("safe", C.c_int), ("test_trials", C.c_int)]


class FqiSpec(C.Structure):
    _fields_ = [("base", Spec), ("batch_size", C.c_int), ("iterations", C.c_int), ("epochs", C.c_int), ("hidden", C.c_int),
                ("sum_order", C.c_int), ("gamma_tau", C.c_double), ("eta", C.c_double)]


class Row(C.Structure):
    _fields_ = [("trial", C.c_int64), ("steps", C.c_int64), ("reward", C.c_double), ("time", C.c_double)]


class Tap(C.Structure):
    _fields_ = [("test", C.c_int32), ("action_index", C.c_int32), ("obs", C.c_double * MAX_DIMS),
                ("action", C.c_double), ("reward", C.c_double), ("terminal", C.c_int32), ("trace_len", C.c_int32),
                ("q", C.c_double * 8), ("delta", C.c_double), ("p_idx", C.c_uint32 * 32), ("state", C.c_double * MAX_STATE)]


class Stats(C.Structure):
    _fields_ = [("learn_steps", C.c_uint64), ("test_steps", C.c_uint64), ("weight_reads", C.c_uint64),
                ("weight_rmws", C.c_uint64), ("trace_entries_sum", C.c_uint64), ("explorations", C.c_uint64),
                ("ties", C.c_uint64)]


class Rand48(C.Structure):
    _fields_ = [("x", C.c_uint64)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    return LIB


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB)
    P = C.POINTER
    L.orc_srand48.argtypes = [P(Rand48), C.c_long]
    L.orc_drand48.argtypes = [P(Rand48)]; L.orc_drand48.restype = C.c_double
    L.orc_lrand48.argtypes = [P(Rand48)]; L.orc_lrand48.restype = C.c_uint32
    L.orc_rand48_jump.argtypes = [P(Rand48), C.c_uint64]
    for f in ("orc_psin", "orc_pcos", "orc_plog"):
        getattr(L, f).argtypes = [C.c_double]; getattr(L, f).restype = C.c_double
    L.orc_spec_pendulum_sarsa.argtypes = [P(Spec)]
    L.orc_spec_cart_pole_balancing_pid.argtypes = [P(Spec)]
    L.orc_tile_project.argtypes = [P(TileSpec), P(C.c_double), P(C.c_uint32)]; L.orc_tile_project.restype = C.c_int
    L.orc_env_step.argtypes = [P(Spec), P(C.c_double), C.c_double, P(C.c_double), P(C.c_double), P(C.c_int)]
    L.orc_env_step.restype = C.c_double
    L.orc_env_state_dims.argtypes = [C.c_int]; L.orc_env_obs_dims.argtypes = [C.c_int]
    L.orc_create.argtypes = [P(Spec), C.c_long]; L.orc_create.restype = C.c_void_p
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_run.argtypes = [C.c_void_p, C.c_int, P(Row), C.c_int, P(Tap), C.c_int, P(C.c_int)]; L.orc_run.restype = C.c_int
    L.orc_get_stats.argtypes = [C.c_void_p, P(Stats)]
    L.orc_weights.argtypes = [C.c_void_p, C.c_int]; L.orc_weights.restype = P(C.c_double)
    L.orc_reset_run.argtypes = [C.c_void_p]; L.orc_reset_run.restype = C.c_int
    L.orc_set_steps_budget.argtypes = [C.c_void_p, C.c_uint64]; L.orc_set_steps_budget.restype = None
    L.orc_trials.argtypes = [C.c_void_p]; L.orc_trials.restype = C.c_int64
    L.orc_set_weights.argtypes = [C.c_void_p, C.c_int, P(C.c_double), C.c_size_t]; L.orc_set_weights.restype = C.c_int
    L.orc_target_syncs.argtypes = [C.c_void_p]; L.orc_target_syncs.restype = C.c_int64
    L.orc_get_state.argtypes = [C.c_void_p, P(C.c_double)]
    L.orc_set_state.argtypes = [C.c_void_p, P(C.c_double)]; L.orc_set_state.restype = None
    L.orc_exp_env_start.argtypes = [C.c_void_p, C.c_int, P(C.c_double)]; L.orc_exp_env_start.restype = None
    L.orc_exp_env_step.argtypes = [C.c_void_p, C.c_double, P(C.c_double), P(C.c_double), P(C.c_int)]; L.orc_exp_env_step.restype = C.c_double
    L.orc_exp_agent_start.argtypes = [C.c_void_p, C.c_int, P(C.c_double)]; L.orc_exp_agent_start.restype = C.c_double
    L.orc_exp_agent_step.argtypes = [C.c_void_p, C.c_int, C.c_double, P(C.c_double), C.c_double]; L.orc_exp_agent_step.restype = C.c_double
    L.orc_exp_agent_end.argtypes = [C.c_void_p, C.c_int, C.c_double, P(C.c_double), C.c_double]; L.orc_exp_agent_end.restype = None
    L.orc_rng_states.argtypes = [C.c_void_p, P(C.c_uint64)]
    L.orc_format_row.argtypes = [P(Row), C.c_char_p, C.c_size_t]; L.orc_format_row.restype = C.c_int
    L.orc_lazy_weight.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_double, C.c_double]
    L.orc_lazy_weight.restype = C.c_double
    L.orc_pexp.argtypes = [C.c_double]; L.orc_pexp.restype = C.c_double
    L.orc_logistic_exp.argtypes = [C.c_double]; L.orc_logistic_exp.restype = C.c_double
    L.orc_fqi_spec_pendulum.argtypes = [P(FqiSpec)]
    L.orc_fqi_create.argtypes = [P(FqiSpec), C.c_long]; L.orc_fqi_create.restype = C.c_void_p
    L.orc_fqi_destroy.argtypes = [C.c_void_p]
    L.orc_fqi_run_batch.argtypes = [C.c_void_p, P(Row)]; L.orc_fqi_run_batch.restype = C.c_int
    L.orc_fqi_params.argtypes = [C.c_void_p, P(C.c_int)]; L.orc_fqi_params.restype = P(C.c_double)
    L.orc_fqi_transitions.argtypes = [C.c_void_p] + [P(P(C.c_double))] * 4; L.orc_fqi_transitions.restype = C.c_size_t
    L.orc_fqi_info.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int), P(C.c_double)]
    L.orc_fqi_q.argtypes = [C.c_void_p, P(C.c_double), C.c_double]; L.orc_fqi_q.restype = C.c_double
    L.orc_fqi_rng.argtypes = [C.c_void_p, P(C.c_uint64)]
    _lib = L
    return L


def pendulum_sarsa_spec(math=MATH_PORTABLE, **over) -> Spec:
    s = Spec()
    load().orc_spec_pendulum_sarsa(C.byref(s))
    s.math = math
    for k, v in over.items():
        setattr(s, k, v)
    return s


def cart_pole_balancing_pid_spec(math=MATH_PORTABLE, **over) -> Spec:
    """The reference's tests/cart_pole_balancing-pid.yaml (golden: tests/golden/cart_pole_balancing-pid-0.txt)."""
    s = Spec()
    load().orc_spec_cart_pole_balancing_pid(C.byref(s))
    s.math = math
    for k, v in over.items():
        setattr(s, k, v)
    return s


class Experiment:
    """One scalar experiment instance of the oracle."""

    def __init__(self, spec: Spec, seed: int):
        self.L = load()
        self.spec = spec
        self.h = self.L.orc_create(C.byref(spec), seed)
        if not self.h:
            raise RuntimeError("orc_create failed (unsupported spec)")

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run(self, n_trials, tap_cap=0):
        rows = (Row * (n_trials + 1))()
        taps = (Tap * max(tap_cap, 1))()
        ntap = C.c_int(0)
        n = self.L.orc_run(self.h, n_trials, rows, n_trials + 1, taps if tap_cap else None, tap_cap, C.byref(ntap))
        return [rows[i] for i in range(n)], [taps[i] for i in range(ntap.value)]

    def format_rows(self, rows) -> str:
        buf = C.create_string_buffer(128)
        out = []
        for r in rows:
            self.L.orc_format_row(C.byref(r), buf, 128)
            out.append(buf.value.decode())
        return "".join(out)

    def set_steps_budget(self, steps: int):
        """experiment/online_learning:steps (online_learning.cpp:154); 0 = none."""
        self.L.orc_set_steps_budget(self.h, int(steps))

    def trials_run(self) -> int:
        return int(self.L.orc_trials(self.h))

    def reset_run(self):
        """Experiment::reset() between two runs (online_learning.cpp:307-308)."""
        if self.L.orc_reset_run(self.h) != 0:
            raise ValueError("orc_reset_run: not restated for this graph")

    def stats(self) -> Stats:
        s = Stats()
        self.L.orc_get_stats(self.h, C.byref(s))
        return s

    def weights(self, slots, table=0):
        p = self.L.orc_weights(self.h, table)
        return np.array([p[int(i)] for i in slots], dtype=np.float64)

    def all_weights(self, table=0):
        n = self.spec.actor_projector.memory if table == 1 else self.spec.projector.memory
        return np.ctypeslib.as_array(self.L.orc_weights(self.h, table), shape=(n,)).copy()

    def set_weights(self, w, table=0):
        w = np.ascontiguousarray(w, dtype=np.float64)
        if self.L.orc_set_weights(self.h, table, w.ctypes.data_as(C.POINTER(C.c_double)), w.size) != 0:
            raise ValueError("orc_set_weights: wrong table or size")

    def state(self):
        st = (C.c_double * MAX_STATE)()
        self.L.orc_get_state(self.h, st)
        return np.array(st[: self.L.orc_env_state_dims(self.spec.env)])

    def rng(self):
        out = (C.c_uint64 * 4)()
        self.L.orc_rng_states(self.h, out)
        return np.array(out[:], dtype=np.uint64)

    # ---- the per-step plug-in interfaces (orc_run is written on top of them) ----
    def env_start(self, test: int):
        """Environment::start (environment.h:48): first observation of a trial."""
        obs = (C.c_double * MAX_DIMS)()
        self.L.orc_exp_env_start(self.h, int(test), obs)
        return np.array(obs[: self.L.orc_env_obs_dims(self.spec.env)])

    def env_step(self, action: float):
        """Environment::step (environment.h:49-51): (tau, obs, reward, terminal)."""
        obs = (C.c_double * MAX_DIMS)(); rw = C.c_double(); tm = C.c_int()
        tau = self.L.orc_exp_env_step(self.h, float(action), obs, C.byref(rw), C.byref(tm))
        return tau, np.array(obs[: self.L.orc_env_obs_dims(self.spec.env)]), rw.value, tm.value

    def _obs(self, obs):
        buf = (C.c_double * MAX_DIMS)()
        for i, v in enumerate(obs):
            buf[i] = float(v)
        return buf

    def agent_start(self, test: int, obs) -> float:
        """Agent::start (agent.h:44-47): the first action of a trial."""
        return self.L.orc_exp_agent_start(self.h, int(test), self._obs(obs))

    def agent_step(self, test: int, tau: float, obs, reward: float) -> float:
        """Agent::step (agent.h:49-52): act, then learn from the transition."""
        return self.L.orc_exp_agent_step(self.h, int(test), float(tau), self._obs(obs), float(reward))

    def agent_end(self, test: int, tau: float, obs, reward: float):
        """Agent::end (agent.h:54-56): the transition into an absorbing state."""
        self.L.orc_exp_agent_end(self.h, int(test), float(tau), self._obs(obs), float(reward))

    def set_state(self, state):
        buf = (C.c_double * MAX_STATE)()
        for i, v in enumerate(state):
            buf[i] = float(v)
        self.L.orc_set_state(self.h, buf)


SUM_SEQUENTIAL, SUM_TREE = 0, 1


def pendulum_fqi_spec(math=MATH_PORTABLE, sum_order=SUM_TREE, **over) -> FqiSpec:
    """The reference's tests/pendulum-fqi-ann.yaml; gamma_tau = pow(gamma, control_step) is recomputed from the final values."""
    s = FqiSpec()
    load().orc_fqi_spec_pendulum(C.byref(s))
    s.base.math = math
    s.sum_order = sum_order
    for k, v in over.items():
        if hasattr(s, k):
            setattr(s, k, v)
        else:
            setattr(s.base, k, v)
    s.gamma_tau = float(s.base.gamma) ** float(s.base.control_step)
    return s


class FqiExperiment:
    """One scalar batch-learning experiment of the oracle (oracle/fqi.c)."""

    def __init__(self, spec: FqiSpec, seed: int):
        self.L = load()
        self.spec = spec
        self.h = self.L.orc_fqi_create(C.byref(spec), seed)
        if not self.h:
            raise RuntimeError("orc_fqi_create failed (unsupported spec)")

    def close(self):
        if self.h:
            self.L.orc_fqi_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run_batch(self) -> Row:
        row = Row()
        if self.L.orc_fqi_run_batch(self.h, C.byref(row)) != 0:
            raise MemoryError("orc_fqi_run_batch")
        return row

    def params(self):
        n = C.c_int()
        p = self.L.orc_fqi_params(self.h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def transitions(self):
        ptrs = [C.POINTER(C.c_double)() for _ in range(4)]
        n = self.L.orc_fqi_transitions(self.h, *[C.byref(p) for p in ptrs])
        shapes = [(n, 3), (n, 2), (n,), (n,)]
        return [np.ctypeslib.as_array(p, shape=sh).copy() for p, sh in zip(ptrs, shapes)]

    def info(self):
        md = C.c_double(); it = C.c_int(); err = C.c_double()
        self.L.orc_fqi_info(self.h, C.byref(md), C.byref(it), C.byref(err))
        return dict(maxdelta=md.value, iterations=it.value, error=err.value)

    def q(self, obs, action):
        o = (C.c_double * 2)(*obs)
        return self.L.orc_fqi_q(self.h, o, float(action))

    def rng(self):
        out = (C.c_uint64 * 3)()
        self.L.orc_fqi_rng(self.h, out)
        return [out[0], out[1], out[2]]

    def format_row(self, row) -> str:
        buf = C.create_string_buffer(128)
        self.L.orc_format_row(C.byref(row), buf, 128)
        return buf.value.decode()


def tile_project(ts: TileSpec, x):
    L = load()
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, ts.dims)
    out = np.zeros((x.shape[0], ts.tilings), np.uint32)
    tmp = (C.c_uint32 * ts.tilings)()
    for i in range(x.shape[0]):
        row = (C.c_double * ts.dims)(*x[i])
        if L.orc_tile_project(C.byref(ts), row, tmp) != 0:
            raise ValueError("invalid tile spec")
        out[i] = tmp[:]
    return out


def env_step(spec: Spec, state, action):
    L = load()
    S, D = L.orc_env_state_dims(spec.env), L.orc_env_obs_dims(spec.env)
    state = np.array(state, dtype=np.float64).reshape(-1, S)
    action = np.asarray(action, dtype=np.float64).reshape(-1)
    n = state.shape[0]
    obs = np.zeros((n, D)); reward = np.zeros(n); term = np.zeros(n, np.int32)
    for i in range(n):
        st = (C.c_double * S)(*state[i]); ob = (C.c_double * D)()
        rw = C.c_double(); tm = C.c_int()
        L.orc_env_step(C.byref(spec), st, float(action[i]), ob, C.byref(rw), C.byref(tm))
        state[i] = st[:]; obs[i] = ob[:]; reward[i] = rw.value; term[i] = tm.value
    return state, obs, reward, term


def timed_run(args):
    """Worker of bench.py's cpu_baseline (one process per host core): run `trials` trials of the
    pendulum SARSA oracle with libm arithmetic, return (env-steps, seconds) without the weight init."""
    import time
    seed, trials = args
    e = Experiment(pendulum_sarsa_spec(math=MATH_LIBM), seed=seed)
    t0 = time.perf_counter()
    e.run(trials)
    dt = time.perf_counter() - t0
    e.close()
    return trials * 100, dt
