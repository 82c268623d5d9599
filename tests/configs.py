"""Matching (grlx_config, oracle spec) pairs for the supported experiment graphs."""
import numpy as np

from tests import oracle_binding as ob

# tools/ (profiling helpers) reuse these builders for the grlx_config half only: with NO_ORACLE set the
# oracle is neither built nor loaded and the spec half is a placeholder that swallows assignments
NO_ORACLE = False


class _NoSpec:
    def __getattr__(self, name):
        return _NoSpec()

    def __setattr__(self, name, value):
        pass

    def __setitem__(self, key, value):
        pass


def _spec(**kw):
    return _NoSpec() if NO_ORACLE else ob.pendulum_sarsa_spec(**kw)


def _set_tile(ts, tilings, memory, resolution, wrapping):
    ts.tilings, ts.memory, ts.dims = tilings, memory, len(resolution)
    for i, (r, w) in enumerate(zip(resolution, wrapping)):
        ts.resolution[i] = r
        ts.wrapping[i] = w


def pendulum(grlx, n, agent=0, **over):
    """grlx = None: the oracle half only (bench.py's cpu_baseline leg)."""
    cfg = grlx.pendulum_sarsa_config(n, agent=agent, **over) if grlx is not None else None
    spec = _spec(agent=agent)
    return cfg, spec


def acrobot(grlx, n, agent=1, **over):
    """dynamics/acrobot + task/acrobot/balancing with the agent block of cfg/pendulum/q_tc.yaml
    (the reference ships no TD yaml for the acrobot)."""
    res = [0.05, 0.05, 0.2, 0.4, 1.0]
    wrap = [0, 0, 0, 0, 0]
    cfg = grlx.acrobot_q_config(n, agent=agent, **over) if grlx is not None else None
    spec = _spec(agent=agent)
    spec.env = 2
    spec.control_step, spec.integration_steps, spec.timeout = 0.05, 5, 20.0
    spec.action_min, spec.action_max, spec.action_steps = -1.0, 1.0, 3
    _set_tile(spec.projector, 16, 8388608, res, wrap)
    return cfg, spec


def cart_pole_ac(grlx, n, **over):
    """cfg/cart_pole/ac_tc.yaml: dynamics/cart_pole + task/cart_pole/swingup, mapping/policy/action,
    predictor/ac/action with a predictor/critic/td critic; two 8,388,608-slot tables."""
    cfg = grlx.cart_pole_ac_config(n, **over) if grlx is not None else None
    spec = _spec()
    spec.env, spec.agent = 1, ob.AGENT_AC
    spec.control_step, spec.integration_steps, spec.timeout, spec.randomization = 0.05, 5, 9.99, 0.0
    spec.end_stop_penalty, spec.action_penalty = (cfg.end_stop_penalty, cfg.action_penalty) if cfg is not None else (0, 0)
    spec.action_min, spec.action_max, spec.action_steps = -15.0, 15.0, 0
    res, wrap = [2.5, 0.157075, 2.5, 1.57075], [0, 6.283, 0, 0]
    for ts in (spec.projector, spec.actor_projector):
        for i in range(8):
            ts.resolution[i] = 0.0
            ts.wrapping[i] = 0.0
        _set_tile(ts, 16, 8388608, res, wrap)
    ar = spec.actor_representation
    ar.init_min, ar.init_max, ar.output_min, ar.output_max, ar.limit = 0.0, 1.0, -15.0, 15.0, 1
    spec.actor_alpha, spec.sigma, spec.theta = 0.01, 5.0, 1.0
    spec.ac_decay_rate, spec.ac_decay_min, spec.ac_update_method, spec.ac_step_limit = 1.0, 0.0, 0, -1.0
    for k, v in over.items():
        if not NO_ORACLE and hasattr(spec, k) and k not in ("tap_replica", "tap_capacity"):
            setattr(spec, k, v)
    return cfg, spec


def compass_walker(grlx, n, agent=1, **over):
    """cfg/compass_walker/qlearning_walk.yaml: model/compass_walker + task/compass_walker/walk, Q-learning."""
    res = [0.0838, 0.1047, 0.1111, 0.2222, 10, 1.2]
    wrap = [0] * 6
    cfg = grlx.compass_walker_q_config(n, agent=agent, **over) if grlx is not None else None
    spec = _spec(agent=agent)
    spec.env = 3
    spec.control_step, spec.integration_steps, spec.timeout = 0.2, 20, 100.0
    spec.slope_angle, spec.initial_state_variation, spec.negative_reward = 0.004, 0.2, -100.0
    spec.action_min, spec.action_max, spec.action_steps = -1.2, 1.2, 3
    _set_tile(spec.projector, 16, 8388608, res, wrap)
    return cfg, spec


def pendulum_qv(grlx, n, **over):
    """cfg/pendulum/qv_tc.yaml: policy/discrete/value/q over the Q table (table 0), predictor/critic/qv with a
    tile-coded state-value table V (table 1: the second projector / representation of the config), beta = 0.1."""
    DBL_MAX = 1.7976931348623157e308
    spec = _spec(agent=ob.AGENT_QV, beta=0.1)
    _set_tile(spec.actor_projector, 16, 8388608, [0.31415, 3.1415], [6.283, 0])
    ar = spec.actor_representation
    ar.init_min, ar.init_max, ar.output_min, ar.output_max, ar.limit = 0.0, 1.0, -DBL_MAX, DBL_MAX, 1
    cfg = None
    if grlx is not None:
        cfg = grlx.pendulum_sarsa_config(n, agent=5, beta=0.1, **over)
        _set_tile(cfg.actor_projector, 16, 8388608, [0.31415, 3.1415], [6.283, 0])
        cr = cfg.actor_representation
        cr.init_min, cr.init_max, cr.output_min, cr.output_max, cr.limit = 0.0, 1.0, -DBL_MAX, DBL_MAX, 1
    return cfg, spec


def cart_pole_q(grlx, n, agent=1, **over):
    """dynamics/cart_pole + task/cart_pole/swingup (cfg/cart_pole/ac_tc.yaml's environment block) under the TD agent block of
    cfg/pendulum/q_tc.yaml: 3 forces over [-15, 15], the actor-critic yaml's state resolution plus one action coordinate."""
    res = [2.5, 0.157075, 2.5, 1.57075, 15.0]
    wrap = [0, 6.283, 0, 0, 0]
    cfg = None
    if grlx is not None:
        cfg = grlx.pendulum_sarsa_config(n, agent=agent, **over)
        cfg.env = 1
        cfg.control_step, cfg.integration_steps, cfg.timeout, cfg.randomization = 0.05, 5, 9.99, 0.0
        cfg.action_min, cfg.action_max, cfg.action_steps = -15.0, 15.0, 3
        _set_tile(cfg.projector, 16, 8388608, res, wrap)
    spec = _spec(agent=agent)
    spec.env = 1
    spec.control_step, spec.integration_steps, spec.timeout, spec.randomization = 0.05, 5, 9.99, 0.0
    spec.end_stop_penalty, spec.action_penalty = (cfg.end_stop_penalty, cfg.action_penalty) if cfg is not None else (0, 0)
    spec.action_min, spec.action_max, spec.action_steps = -15.0, 15.0, 3
    _set_tile(spec.projector, 16, 8388608, res, wrap)
    return cfg, spec
