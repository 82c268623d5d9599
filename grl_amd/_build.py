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
           "grlx_rollout.h", "grlx_rollout_wide.h", "grlx_rollout_ac.h", "grlx_rollout_ac_wide.h", "grlx_rollout_qv.h", "grlx_rollout_acc.h", "grlx_rollout_tgt.h", "grlx_env_server.h", "grlx_env_server_wide.h", "grlx_step.h",
           os.path.join("..", "..", "include", "grlx.h")]
FQI_HEADERS = ["grlx_math_batch.h"]          # included by grlx_fqi.hip only
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
TAG_DEFINE = "-DGRLX_BUILD_PIPELINE="
# per-source device flags.  The batch path runs ONE wave per SIMD (92 KB of LDS per block): nothing but instruction-level parallelism
# inside the wave hides its f64 latencies, so its translation unit is scheduled for that (the rollout kernels are not: they are
# scheduled for register pressure).
SOURCE_FLAGS = {"grlx_fqi.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
# The device code is not taken from hipcc as is: it is compiled to assembly together with the compiler's machine code
# after register allocation, passed through _exec_prologue (a work-around for a register-allocation bug of this
# compiler, see that file and DESIGN.md section 4.1f: the misplaced copies are FOUND in the machine code and MOVED in the
# assembly; anything it cannot handle stops the build), assembled, linked and bundled with the same tools and options
# the hipcc driver uses (`hipcc -###`), and handed to the host compilation of the same source with
# -fcuda-include-gpubinary.  The library carries the tag (grlx_build_pipeline()); capi.load() refuses one without it.
PIPELINE = "device-asm+mir-exec-prologue-fix/2"

FLAGS_FILE = LIB + ".flags"      # the flags the library was built with: a change of flags makes it stale


def _flags() -> str:
    return " ".join(FLAGS + os.environ.get("GRLX_EXTRA_FLAGS", "").split() + [PIPELINE] + [k + ":" + " ".join(v) for k, v in sorted(SOURCE_FLAGS.items())])


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
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + FQI_HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "_exec_prologue.py")]
    if not os.path.exists(LIB + ".asmfix"):
        return True
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose=False):
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("build step failed: " + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    return res


def _llvm_tool(name: str) -> str:
    for d in ("/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"):
        if os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    raise RuntimeError(name + " not found under /opt/rocm: cannot build the HIP extension")


def _build_hip_object(hipcc, src, tmp, flags, verbose, report):
    """One .hip source -> host object carrying the filtered device code."""
    from . import _exec_prologue
    stem = os.path.join(tmp, os.path.splitext(os.path.basename(src))[0])
    flags = flags + SOURCE_FLAGS.get(os.path.basename(src), [])
    cmd = [hipcc] + flags + ["--cuda-device-only", "-S", src, "-o", stem + ".s", "-mllvm", "-print-after=stack-slot-coloring"]
    if verbose:
        print(" ".join(cmd), "2>", stem + ".mir")
    with open(stem + ".mir", "w") as mir:        # the machine code after register allocation goes to stderr (hundreds of MB)
        rc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=mir, text=True).returncode
    if rc != 0:
        _run([hipcc] + flags + ["--cuda-device-only", "-S", src, "-o", stem + ".s"], verbose)      # again, for a readable message
        raise RuntimeError(f"{src}: hipcc failed only with -print-after=stack-slot-coloring")
    with open(stem + ".mir", errors="replace") as f:
        found, problems = _exec_prologue.find_misplaced(f)
    if os.environ.get("GRLX_KEEP_MIR"):          # diagnostic: keep the machine code the filter read (hundreds of MB)
        shutil.copy(stem + ".mir", os.path.join(os.environ["GRLX_KEEP_MIR"], os.path.basename(stem) + ".mir"))
    os.remove(stem + ".mir")
    with open(stem + ".s") as f:
        fixed, n_fixed, p2 = _exec_prologue.apply(f.read().split("\n"), found)
    with open(stem + ".fixed.s", "w") as f:
        f.write("\n".join(fixed))
    report[os.path.basename(src)] = (len(found), n_fixed)
    if problems or p2:
        raise RuntimeError(f"{src}: the exec-prologue filter cannot make this device code safe (DESIGN.md 4.1f):\n  " + "\n  ".join(problems + p2))
    _run([_llvm_tool("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", stem + ".fixed.s", "-o", stem + ".dev.o"], verbose)
    _run([_llvm_tool("lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", stem + ".out", stem + ".dev.o"], verbose)
    _run([_llvm_tool("clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + stem + ".out",
          "-output=" + stem + ".hipfb"], verbose)
    _run([hipcc] + flags + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb", "-o", stem + ".host.o"], verbose)
    return stem + ".host.o"


# what each translation unit includes (besides itself): an object whose inputs did not change is taken from the object cache
# (outside the tree: $GRLX_OBJCACHE or /tmp/grlx_objcache), so that editing grlx_api.cpp does not recompile the kernels
UNIT_DEPS = {
    "grlx_kernels.hip": [h for h in HEADERS],
    "grlx_fqi.hip": ["grlx_internal.h", "grlx_math.h", "grlx_math_batch.h", "grlx_rng.h", "grlx_tile.h", "grlx_table.h", "grlx_envs.h", os.path.join("..", "..", "include", "grlx.h")],
    "grlx_api.cpp": ["grlx_internal.h", os.path.join("..", "..", "include", "grlx.h")],
}


def _unit_key(name: str, flags) -> str:
    import hashlib
    h = hashlib.sha256()
    h.update((" ".join(flags) + "|" + PIPELINE + "|" + " ".join(SOURCE_FLAGS.get(name, []))).encode())
    for f in [name] + UNIT_DEPS[name] + [os.path.abspath(__file__), os.path.join(HERE, "_exec_prologue.py")]:
        with open(f if os.path.isabs(f) else os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:24]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libgrlx.so if missing or older than its sources; return its path."""
    if not force and not _stale():
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    cache = os.environ.get("GRLX_OBJCACHE", "/tmp/grlx_objcache")
    os.makedirs(cache, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP extension")
    os.makedirs(LIBDIR, exist_ok=True)
    flags = FLAGS + os.environ.get("GRLX_EXTRA_FLAGS", "").split()       # extra flags: experiments only
    report = {}
    with tempfile.TemporaryDirectory(prefix="grlx_build_") as tmp:
        def one(name):
            src = os.path.join(CSRC, name)
            key = os.path.join(cache, name + "." + _unit_key(name, flags))
            if not force and os.path.exists(key + ".o") and (not name.endswith(".hip") or os.path.exists(key + ".report")):
                if name.endswith(".hip"):
                    with open(key + ".report") as f:
                        report[name] = tuple(int(x) for x in f.read().split())
                return key + ".o"
            if name.endswith(".hip"):
                obj = _build_hip_object(hipcc, src, tmp, flags, verbose, report)
                with open(key + ".report", "w") as f:
                    f.write("%d %d" % report[name])
            else:
                obj = os.path.join(tmp, os.path.splitext(name)[0] + ".o")
                _run([hipcc] + flags + [TAG_DEFINE + '"' + PIPELINE + '"', "-c", src, "-o", obj], verbose)
            shutil.copy(obj, key + ".o")
            return key + ".o"
        with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
            objs = list(pool.map(one, SOURCES))
        _run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs, verbose)
    with open(FLAGS_FILE, "w") as f:
        f.write(_flags() + "\n")
    with open(LIB + ".asmfix", "w") as f:       # how many block heads the assembly filter rewrote, per source
        for k in sorted(report):
            f.write(f"{k}: {report[k][0]} misplaced block head(s) in the machine code after register allocation, {report[k][1]} rewritten in the assembly\n")
    return LIB


HOST_SOURCES = ["configurable.cpp", "objects.cpp", "multi_gpu.cpp", "grlxd.cpp"]       # multi_gpu.cpp: `grlxd -g N` (HIP runtime API + RCCL)
OPS_SOURCES = ["configurable.cpp", "objects.cpp", "multi_gpu.cpp", "grlx_ops.cpp"]    # grlx_ops: Projector::project / Environment::step from the command line
HOST_HEADERS = ["configurable.h", "objects.h", "multi_gpu.h"]
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
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", out] + \
              [os.path.join(hostdir, f) for f in sources] + \
              ["-L" + LIBDIR, "-lgrlx", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib",
               "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("g++ failed:\n" + res.stdout + res.stderr)
    return GRLXD


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
