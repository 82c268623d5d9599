"""Build the HIP extension (libgrlx.so) in-tree for gfx950.

`hipcc` cross-compiles without a GPU, so this runs in the CPU container too.
-ffp-contract=off is REQUIRED: the reference's arithmetic is unfused IEEE double
(only grlx_math.h fuses, through explicit fma calls).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgrlx.so")
SOURCES = ["grlx_kernels.hip", "grlx_fqi.hip", "grlx_api.cpp"]
HEADERS = ["grlx_internal.h", "grlx_math.h", "grlx_rng.h", "grlx_tile.h", "grlx_table.h", "grlx_envs.h", "grlx_policy.h", "grlx_update.h",
           "grlx_rollout.h", "grlx_rollout_wide.h", "grlx_rollout_ac.h", "grlx_rollout_ac_wide.h", "grlx_rollout_qv.h", "grlx_rollout_acc.h", "grlx_rollout_tgt.h",
           os.path.join("..", "..", "include", "grlx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


FLAGS_FILE = LIB + ".flags"      # the flags the library was built with: a change of flags makes it stale


def _flags() -> str:
    return " ".join(FLAGS + os.environ.get("GRLX_EXTRA_FLAGS", "").split())


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    try:
        with open(FLAGS_FILE) as f:
            if f.read().strip() != _flags():
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libgrlx.so if missing or older than its sources; return its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP extension")
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("GRLX_EXTRA_FLAGS", "").split()       # experiments only; the default build uses FLAGS
    cmd = [hipcc] + FLAGS + extra + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    with open(FLAGS_FILE, "w") as f:
        f.write(_flags() + "\n")
    return LIB


HOST_SOURCES = ["configurable.cpp", "objects.cpp", "grlxd.cpp"]
OPS_SOURCES = ["configurable.cpp", "objects.cpp", "grlx_ops.cpp"]      # grlx_ops: Projector::project / Environment::step from the command line
HOST_HEADERS = ["configurable.h", "objects.h"]
BINDIR = os.path.join(HERE, "bin")
GRLXD = os.path.join(BINDIR, "grlxd")
GRLX_OPS = os.path.join(BINDIR, "grlx_ops")


def build_host(force: bool = False, verbose: bool = False) -> str:
    """Compile the C++ host layer + deployer (grlxd) against libgrlx.so."""
    build(force=False, verbose=verbose)
    hostdir = os.path.join(CSRC, "host")
    deps = [os.path.join(hostdir, f) for f in HOST_SOURCES + HOST_HEADERS + ["grlx_ops.cpp"]] + [LIB, os.path.join(HERE, "..", "include", "grlx.h")]
    if not force and os.path.exists(GRLXD) and os.path.exists(GRLX_OPS) and all(os.path.getmtime(d) <= min(os.path.getmtime(GRLXD), os.path.getmtime(GRLX_OPS)) for d in deps):
        return GRLXD
    os.makedirs(BINDIR, exist_ok=True)
    for out, sources in ((GRLXD, HOST_SOURCES), (GRLX_OPS, OPS_SOURCES)):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", out] + [os.path.join(hostdir, f) for f in sources] + \
              ["-L" + LIBDIR, "-lgrlx", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("g++ failed:\n" + res.stdout + res.stderr)
    return GRLXD


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
