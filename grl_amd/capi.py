"""ctypes binding of the C ABI in include/grlx.h (the drop-in boundary).

The library is the product; there is no Python or CPU fallback.  If the shared
object is missing, loading raises -- build it with `python -m grl_amd._build`
(or __graft_entry__.build()).
"""
import ctypes as C
import os

from . import _build

MAX_DIMS, MAX_STATE, MAX_ACTIONS = 8, 12, 8

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_TABLE_FULL, ERR_DOMAIN, ERR_ROWS_FULL, ERR_OOM = 0, -1, -2, -3, -4, -5, -6, -7
ENV_PENDULUM, ENV_CART_POLE, ENV_ACROBOT, ENV_COMPASS_WALKER, ENV_CART_POLE_BALANCING, ENV_EXTERNAL = 0, 1, 2, 3, 4, 5
AGENT_SARSA, AGENT_Q, AGENT_AC, AGENT_EXPECTED_SARSA, AGENT_ADVANTAGE, AGENT_QV = 0, 1, 2, 3, 4, 5
TRACE_NONE, TRACE_REPLACING, TRACE_ACCUMULATING = 0, 1, 2


class TileSpec(C.Structure):
    _fields_ = [("tilings", C.c_int32), ("memory", C.c_int32), ("dims", C.c_int32), ("safe", C.c_int32),
                ("resolution", C.c_double * MAX_DIMS), ("wrapping", C.c_double * MAX_DIMS)]


class LinearSpec(C.Structure):
    _fields_ = [("init_min", C.c_double), ("init_max", C.c_double), ("output_min", C.c_double),
                ("output_max", C.c_double), ("limit", C.c_int32), ("reserved", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_replicas", C.c_int32), ("test_interval", C.c_int32), ("env", C.c_int32),
                ("control_step", C.c_double), ("integration_steps", C.c_int32), ("discrete_time", C.c_int32),
                ("timeout", C.c_double), ("randomization", C.c_double),
                ("action_min", C.c_double), ("action_max", C.c_double), ("action_steps", C.c_int32), ("agent", C.c_int32),
                ("projector", TileSpec), ("representation", LinearSpec),
                ("epsilon", C.c_double), ("decay_rate", C.c_double), ("decay_min", C.c_double),
                ("alpha", C.c_double), ("gamma", C.c_double), ("lambda_", C.c_double),
                ("trace", C.c_int32), ("tap_starts", C.c_int32),
                ("actor_projector", TileSpec), ("actor_representation", LinearSpec),
                ("actor_alpha", C.c_double), ("sigma", C.c_double), ("theta", C.c_double),
                ("ac_decay_rate", C.c_double), ("ac_decay_min", C.c_double), ("ac_step_limit", C.c_double),
                ("ac_update_method", C.c_int32),
                ("table_log2_capacity", C.c_int32), ("max_rows", C.c_int32), ("tap_replica", C.c_int32),
                ("tap_capacity", C.c_int32), ("end_stop_penalty", C.c_int32), ("action_penalty", C.c_int32),
                ("force_generic", C.c_int32),
                ("slope_angle", C.c_double), ("initial_state_variation", C.c_double), ("negative_reward", C.c_double),
                ("kappa", C.c_double), ("beta", C.c_double), ("replicas_per_wave", C.c_int32), ("tap_deferred", C.c_int32),
                ("target_interval", C.c_int32), ("wave_limit", C.c_int32), ("table_log2_max", C.c_int32), ("target_tau", C.c_double),
                ("test_trials", C.c_int32), ("reserved0", C.c_int32)]


class FqiConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_replicas", C.c_int32), ("env", C.c_int32), ("integration_steps", C.c_int32),
                ("control_step", C.c_double), ("timeout", C.c_double), ("action_min", C.c_double), ("action_max", C.c_double),
                ("action_steps", C.c_int32), ("batch_size", C.c_int32), ("gamma", C.c_double), ("iterations", C.c_int32),
                ("epochs", C.c_int32), ("hidden", C.c_int32), ("max_batches", C.c_int32), ("eta", C.c_double)]


class Tap(C.Structure):
    _fields_ = [("test", C.c_int32), ("action_index", C.c_int32), ("terminal", C.c_int32), ("trace_len", C.c_int32),
                ("obs", C.c_double * MAX_DIMS), ("action", C.c_double), ("reward", C.c_double), ("delta", C.c_double),
                ("q", C.c_double * MAX_ACTIONS), ("p_idx", C.c_uint32 * 32), ("state", C.c_double * MAX_STATE)]


class GrlxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"grlx error {code}: {msg}")
        self.code = code


_lib = None

_P = C.POINTER
_SIGS = {
    "grlx_last_error": (C.c_char_p, []),
    "grlx_abi_version": (C.c_int, []),
    "grlx_build_pipeline": (C.c_char_p, []),
    "grlx_device_count": (C.c_int, []),
    "grlx_config_pendulum_sarsa": (None, [_P(Config)]),
    "grlx_config_cart_pole_ac": (None, [_P(Config)]),
    "grlx_create": (C.c_int, [_P(Config), _P(C.c_int64), _P(C.c_void_p)]),
    "grlx_destroy": (C.c_int, [C.c_void_p]),
    "grlx_run": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "grlx_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grlx_set_diag": (C.c_int, [C.c_void_p, C.c_int]),
    "grlx_read_diag": (C.c_int, [C.c_void_p, _P(C.c_uint64), C.c_int, _P(C.c_int)]),
    "grlx_rows": (C.c_int, [C.c_void_p]),
    "grlx_read_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_int64), _P(C.c_int64), _P(C.c_double)]),
    "grlx_last_kernel": (C.c_int, [C.c_void_p]),
    "grlx_replicas_per_wave": (C.c_int, [C.c_void_p]),
    "grlx_read_row_times": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_double)]),
    "grlx_curve_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grlx_step_counts": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
    "grlx_get_env_state": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double)]),
    "grlx_get_rng": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_uint64)]),
    "grlx_get_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _P(C.c_uint32), C.c_int, _P(C.c_double)]),
    "grlx_table_load": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _P(C.c_uint32)]),
    "grlx_table_capacity": (C.c_int, [C.c_void_p, _P(C.c_uint32)]),
    "grlx_grow_tables": (C.c_int, [C.c_void_p, C.c_uint32]),
    "grlx_reset_run": (C.c_int, [C.c_void_p]),
    "grlx_run_steps": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]),
    "grlx_replica_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "grlx_get_target_weights": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_uint32), C.c_int, _P(C.c_double), _P(C.c_uint32)]),
    "grlx_export_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _P(C.c_double)]),
    "grlx_load_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_double), C.c_uint64]),
    "grlx_read_taps": (C.c_int, [C.c_void_p, _P(Tap), C.c_int, _P(C.c_int)]),
    "grlx_env_start": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), _P(C.c_double)]),
    "grlx_env_advance": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_int32)]),
    "grlx_agent_start": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), _P(C.c_double), _P(C.c_double)]),
    "grlx_agent_step": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), C.c_double, _P(C.c_double), _P(C.c_double), _P(C.c_int32), _P(C.c_double)]),
    "grlx_agent_end": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), C.c_double, _P(C.c_double), _P(C.c_double)]),
    "grlx_project": (C.c_int, [_P(TileSpec), _P(C.c_double), C.c_int, _P(C.c_uint32)]),
    "grlx_env_step": (C.c_int, [_P(Config), _P(C.c_double), _P(C.c_double), C.c_int, _P(C.c_double), _P(C.c_double), _P(C.c_int32)]),
    "grlx_env_dims": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_int)]),
    "grlx_read": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), _P(C.c_uint32), C.c_int, _P(C.c_double)]),
    "grlx_write": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), _P(C.c_uint32), C.c_int, _P(C.c_double), C.c_double]),
    "grlx_update": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int32), _P(C.c_uint32), C.c_int, _P(C.c_double)]),
    "grlx_math": (C.c_int, [C.c_int, _P(C.c_double), _P(C.c_double), C.c_int, _P(C.c_double)]),
    "grlx_rand48_at": (C.c_int, [C.c_int64, _P(C.c_uint64), C.c_int, _P(C.c_double)]),
    "grlx_fqi_config_pendulum": (None, [_P(FqiConfig)]),
    "grlx_fqi_create": (C.c_int, [_P(FqiConfig), _P(C.c_int64), _P(C.c_void_p)]),
    "grlx_fqi_destroy": (C.c_int, [C.c_void_p]),
    "grlx_fqi_run_batch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grlx_fqi_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "grlx_fqi_read_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_int64), _P(C.c_int64), _P(C.c_double)]),
    "grlx_fqi_get_params": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double), C.c_int]),
    "grlx_fqi_get_transitions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "grlx_fqi_info": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_int64), _P(C.c_double), _P(C.c_int32), _P(C.c_double), _P(C.c_uint64)]),
}

# include/grlx_diag.h: diagnostic exports (tools, tests, bench.py), not part of the boundary
_DIAG_SIGS = {
    "grlx_env_server_counts": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "grlx_env_server_debug": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "grlx_fqi_debug_stamps": (C.c_int, [C.c_void_p, _P(C.c_ulonglong), C.c_int]),
}
ABI_VERSION = 2         # include/grlx.h: GRLX_ABI_VERSION


def lib_path() -> str:
    """grl_amd/lib/libgrlx.so; GRLX_LIB names another build of the same library (A/B timing of two
    builds on one GPU box, tools/ab_bench.sh) -- never a fallback: a missing file is an error."""
    return os.environ.get("GRLX_LIB") or _build.LIB


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname as the system
    one).  Two HIP runtimes in one process cannot both own the GPU, so when torch is
    installed its runtime is loaded first and libgrlx.so binds to it (soname match);
    a later `import torch` then reuses the same object.  Without torch the system
    runtime in /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libgrlx.so (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build the HIP extension first (python -m grl_amd._build); "
                          "grl_amd has no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, (res, args) in list(_SIGS.items()) + list(_DIAG_SIGS.items()):
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.grlx_abi_version() != ABI_VERSION:
        raise ImportError(f"libgrlx.so ABI version {lib.grlx_abi_version()}, this binding is written for {ABI_VERSION}: rebuild (python -m grl_amd._build)")
    tag = (lib.grlx_build_pipeline() or b"").decode()
    if tag != _build.PIPELINE:
        raise ImportError(f"{path} was not built by grl_amd._build (pipeline tag {tag!r}, expected {_build.PIPELINE!r}): a plain hipcc build skips the "
                          "exec-prologue filter and its kernels can read stale lanes (DESIGN.md 4.1f); run `python -m grl_amd._build`")
    _lib = lib
    return lib


def check(code: int) -> int:
    if code < 0:
        raise GrlxError(code, load().grlx_last_error().decode())
    return code
