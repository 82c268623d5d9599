"""Matching (grlx_config, oracle spec) pairs for the supported experiment graphs."""
import numpy as np

from tests import oracle_binding as ob


def _set_tile(ts, tilings, memory, resolution, wrapping):
    ts.tilings, ts.memory, ts.dims = tilings, memory, len(resolution)
    for i, (r, w) in enumerate(zip(resolution, wrapping)):
        ts.resolution[i] = r
        ts.wrapping[i] = w


def pendulum(grlx, n, agent=0, **over):
    cfg = grlx.pendulum_sarsa_config(n, agent=agent, **over)
    spec = ob.pendulum_sarsa_spec(agent=agent)
    return cfg, spec


def acrobot(grlx, n, agent=1, **over):
    """dynamics/acrobot + task/acrobot/balancing with the agent block of cfg/pendulum/q_tc.yaml
    (the reference ships no TD yaml for the acrobot)."""
    res = [0.05, 0.05, 0.2, 0.4, 1.0]
    wrap = [0, 0, 0, 0, 0]
    cfg = grlx.pendulum_sarsa_config(n, agent=agent, **over)
    cfg.env = grlx.capi.ENV_ACROBOT
    cfg.control_step, cfg.integration_steps, cfg.timeout = 0.05, 5, 20.0
    cfg.action_min, cfg.action_max, cfg.action_steps = -1.0, 1.0, 3
    _set_tile(cfg.projector, 16, 8388608, res, wrap)
    spec = ob.pendulum_sarsa_spec(agent=agent)
    spec.env = 2
    spec.control_step, spec.integration_steps, spec.timeout = 0.05, 5, 20.0
    spec.action_min, spec.action_max, spec.action_steps = -1.0, 1.0, 3
    _set_tile(spec.projector, 16, 8388608, res, wrap)
    return cfg, spec
