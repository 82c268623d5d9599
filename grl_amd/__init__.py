"""grl_amd -- MI355X-native runner for grl's OnlineLearningExperiment hot path.

The product is the HIP library behind include/grlx.h (grl_amd/csrc); this
package is its ctypes host.  Importing does not load the library; the first use
does, and fails loudly when it has not been built (no CPU fallback).
"""
from . import capi, runner  # noqa: F401
from .runner import FqiRunner, Runner, acrobot_q_config, pendulum_fqi_config, cart_pole_ac_config, compass_walker_q_config, pendulum_sarsa_config  # noqa: F401

__all__ = ["capi", "runner", "Runner", "FqiRunner", "pendulum_fqi_config", "pendulum_sarsa_config", "cart_pole_ac_config", "acrobot_q_config", "compass_walker_q_config"]
