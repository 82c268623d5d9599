"""The C-ABI library builds, loads and exports every symbol include/grlx.h declares
(no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="grlx.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grlx_[a-z0-9_]+)\s*\(", text)))


def exported_functions(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("grlx_")})


def test_exports_every_declared_symbol(grlx):
    lib = C.CDLL(grlx.capi.lib_path())
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/grlx.h but not exported"
    # and the Python binding covers them all
    assert set(names) == set(grlx.capi._SIGS.keys())


def test_every_exported_symbol_is_declared(grlx):
    """exported == declared: the boundary in include/grlx.h, the diagnostics in include/grlx_diag.h, nothing else."""
    diag = declared_functions("grlx_diag.h")
    assert diag and not set(diag) & set(declared_functions())
    assert exported_functions(grlx.capi.lib_path()) == sorted(set(declared_functions()) | set(diag))
    assert set(diag) == set(grlx.capi._DIAG_SIGS.keys())


def test_struct_layout_matches_header(grlx):
    cfg = grlx.pendulum_sarsa_config(3)
    assert cfg.struct_size == C.sizeof(grlx.capi.Config)
    assert cfg.n_replicas == 3 and cfg.projector.tilings == 16 and cfg.projector.memory == 8388608
    assert cfg.alpha == 0.2 and cfg.gamma == 0.97 and cfg.lambda_ == 0.65 and cfg.epsilon == 0.05
    assert cfg.max_rows == 256 and cfg.tap_replica == -1


def test_validation_errors_without_device(grlx):
    """bad_param conditions surface as GRLX_ERR_INVALID before any device work."""
    capi = grlx.capi
    cfg = grlx.pendulum_sarsa_config(1)
    cfg.projector.wrapping[0] = 1.0          # 1.0*16/0.31415 is not an integer (tile_coding.cpp:72-78)
    with pytest.raises(capi.GrlxError) as ei:
        grlx.Runner(cfg, [1])
    assert ei.value.code == capi.ERR_INVALID and "wrapping" in str(ei.value)
    cfg = grlx.pendulum_sarsa_config(1)
    cfg.struct_size = 12
    with pytest.raises(capi.GrlxError) as ei:
        grlx.Runner(cfg, [1])
    assert ei.value.code == capi.ERR_INVALID


def test_no_cpu_fallback(grlx):
    """Without a device every compute entry point refuses: the product never computes on the CPU."""
    capi = grlx.capi
    if capi.load().grlx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.GrlxError) as ei:
        grlx.Runner(grlx.pendulum_sarsa_config(1), [1])
    assert ei.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.GrlxError) as ei:
        grlx.runner.device_math(0, [1.0])
    assert ei.value.code == capi.ERR_NO_DEVICE


@pytest.mark.parametrize("over,word", [(dict(timeout=float("nan")), "timeout"), (dict(timeout=float("inf")), "timeout"),
                                       (dict(timeout=-1.0), "timeout"), (dict(timeout=1e9), "GRLX_MAX_EPISODE_STEPS"),
                                       (dict(timeout=3100.0), "GRLX_MAX_EPISODE_STEPS"), (dict(action_steps=4), "not built"),
                                       (dict(action_steps=7), "not built")])
def test_unbounded_episodes_and_unbuilt_combinations_are_refused(grlx, over, word):
    """The fused kernels leave an episode only on the task's terminal state: a timeout that never comes would be a
    GPU hang, and an (environment, actions) pair without an instantiation a launch failure -- both are GRLX_ERR_INVALID
    at create, before any device work."""
    capi = grlx.capi
    cfg = grlx.pendulum_sarsa_config(1, **over)
    with pytest.raises(capi.GrlxError) as ei:
        grlx.Runner(cfg, [1])
    assert ei.value.code == capi.ERR_INVALID and word in str(ei.value), str(ei.value)


def test_longest_supported_episode_is_accepted_up_to_the_device_check(grlx):
    capi = grlx.capi
    cfg = grlx.pendulum_sarsa_config(1, timeout=2999.0)        # 99,968 steps of 0.03 s: inside the cap
    try:
        r = grlx.Runner(cfg, [1])
        r.close()
    except capi.GrlxError as ex:
        assert ex.code == capi.ERR_NO_DEVICE
