"""Build the native library: hand-written HIP kernels (gfx950) + the C++ host side, one shared object.

    python -m libbsc_amd.build            # incremental
    python -m libbsc_amd.build --force
    python -m libbsc_amd.build --asan     # second library with AddressSanitizer + UBSan on the C++ host side (csrc/host/*.cpp,
                                          # device code unchanged): libbsc_amd/lib/asan/libbsc_mi355x.so, see asan_env()

Output: libbsc_amd/lib/libbsc_mi355x.so (git-ignored; travels to the GPU box with gpurun).
hipcc cross-compiles for gfx950 without a GPU present.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libbsc_mi355x.so")
ASAN_OBJ = os.path.join(HERE, "lib", "obj_asan")
ASAN_LIB = os.path.join(HERE, "lib", "asan", "libbsc_mi355x.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I", INCLUDE, "-I", CSRC, "-Wall", "-Wno-unused-result"]
DEVFLAGS = [f"--offload-arch={ARCH}"]
# Sanitizer build (SURVEY.md:248): the C++ translation units under csrc/host/ — container, pipeline, coder pool, the three coders and
# decoders, LZP, framing — are compiled by g++ with AddressSanitizer + UBSan and run on GCC's runtime; the .hip files (kernels and
# their launch code) are compiled as always.  Why not hipcc's own -fsanitize=address for everything: ROCm's ASan runtime intercepts
# hsa_amd_memory_pool_allocate to manage device memory itself, and on this image every HIP allocation then dies with "allocator is
# trying to allocate 0x400000 bytes" (with torch's HSA copy in the process and without; profiles/r04/asan.txt).  GCC's runtime knows
# nothing about HSA and leaves the GPU alone.  The runtime is preloaded so that an uninstrumented python can host the library.
GXX = shutil.which("g++") or "g++"
SANFLAGS = ["-fsanitize=address,undefined", "-fno-sanitize=alignment,vptr", "-fno-sanitize-recover=undefined",
            "-fno-omit-frame-pointer", "-g", "-DBSC_SANITIZE=1"]


def asan_runtime():
    out = []
    for name in ("libasan.so", "libubsan.so"):
        r = subprocess.run([GXX, "-print-file-name=" + name], capture_output=True, text=True)
        out.append(os.path.realpath(r.stdout.strip()))
    return ":".join(out)


def asan_env(extra_options=""):
    """Environment for running an uninstrumented host (python, the reference CLI) on the sanitizer build."""
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan_runtime() + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    # leaks: python itself never frees everything; the HSA runtime maps memory ASan has not seen (protect_shadow_gap)
    env["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0" + (":" + extra_options if extra_options else "")
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["BSC_LIB_OVERRIDE"] = ASAN_LIB
    # (under the preloaded runtime's dlopen interceptor torch's own RUNPATH is not honoured: "libcaffe2_nvrtc.so: cannot open shared
    # object file" at the first CUDA call)
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            tl = os.path.join(list(spec.submodule_search_locations)[0], "lib")
            env["LD_LIBRARY_PATH"] = tl + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    except Exception:
        pass
    return env


def _sources():
    out = []
    for sub in ("device", "host"):
        d = os.path.join(CSRC, sub)
        if not os.path.isdir(d):
            continue
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(d, f))
    return out


def _deps_hash(src, asan=False):
    """hash of the source plus every header under csrc/ and include/ (coarse but safe)."""
    h = hashlib.sha1()
    files = [src]
    for root in (CSRC, INCLUDE):
        for dp, _, fs in os.walk(root):
            for f in fs:
                if f.endswith((".h", ".hpp", ".inc")):
                    files.append(os.path.join(dp, f))
    root = os.path.dirname(HERE)
    for f in sorted(set(files)):
        with open(f, "rb") as fh:
            h.update(os.path.relpath(f, root).encode() + b"\0" + fh.read())      # (relative: the tree is built here and used on the GPU box under another path)
    h.update(" ".join(_command(src, "OBJ", asan)).replace(root, ".").encode())       # the full per-source command line: host-only flags included
    return h.hexdigest()


def _command(src, obj, asan=False):
    if asan and not src.endswith(".hip"):
        return [GXX] + COMMON + SANFLAGS + ["-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include"), "-march=x86-64-v3",
                                           "-Wno-unknown-pragmas", "-Wno-attributes", "-c", src, "-o", obj]
    cmd = [HIPCC] + COMMON + (["-g"] if asan else [])
    if src.endswith(".hip"):
        cmd += DEVFLAGS
    else:
        # host-only translation units: plain C++ through hipcc's clang (HIP runtime API headers only)
        # x86-64-v3 (AVX2/BMI2/LZCNT): same ISA floor as the reference's own makefile (-mavx2, makefile:20)
        # tuned (scheduling only, the ISA stays x86-64-v3) for the EPYC hosts MI355X boxes ship with: static coder -2.5 % per stream
        cmd += ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include"), "-march=x86-64-v3", "-mtune=znver4"]
    return cmd + ["-c", src, "-o", obj]


def _compile(src, force, asan=False):
    name = os.path.splitext(os.path.basename(src))[0]
    obj = os.path.join(ASAN_OBJ if asan else OBJ, name + ".o")
    stamp = obj + ".sha1"
    want = _deps_hash(src, asan)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj, False
    cmd = _command(src, obj, asan)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("compile failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    with open(stamp, "w") as fh:
        fh.write(want)
    return obj, True


def build(force=False, verbose=True, asan=False):
    LIB = ASAN_LIB if asan else globals()["LIB"]
    os.makedirs(ASAN_OBJ if asan else OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, asan), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or force or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-Wl,-Bsymbolic"]

        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[libbsc_amd.build] linked {LIB} ({len(objs)} objects)")
    elif verbose:
        print(f"[libbsc_amd.build] up to date: {LIB}")
    if not asan:
        _build_driver(force or changed)
    return LIB


DRIVER_SRC = os.path.join(CSRC, "driver", "bsc_mgpu.cpp")
DRIVER_EXE = os.path.join(HERE, "lib", "bsc_mgpu")
JOB_BENCH_SRC = os.path.join(os.path.dirname(HERE), "tools", "job_bench.cpp")
JOB_BENCH_EXE = os.path.join(HERE, "lib", "job_bench")


def _build_driver(force):
    """the multi-GPU file compressor (csrc/driver/bsc_mgpu.cpp): plain C++ over include/*.h, linked against the library next to it"""
    if not os.path.exists(DRIVER_SRC):
        return
    srcs = [DRIVER_SRC, LIB] + ([JOB_BENCH_SRC] if os.path.exists(JOB_BENCH_SRC) else [])
    exes = [DRIVER_EXE] + ([JOB_BENCH_EXE] if os.path.exists(JOB_BENCH_SRC) else [])
    if not force and all(os.path.exists(e) for e in exes) and min(os.path.getmtime(e) for e in exes) >= max(os.path.getmtime(x) for x in srcs):
        return
    cmd = [GXX, "-O2", "-std=c++17", "-I", INCLUDE, DRIVER_SRC, "-L", os.path.dirname(LIB), "-lbsc_mi355x",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", DRIVER_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("driver build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    # tools/job_bench.cpp: bench.py's workload through bscgpu_job_* alone (a C caller of the product's C ABI); built next to the library so
    # that it travels to the GPU box
    if os.path.exists(JOB_BENCH_SRC):
        cmd = [GXX, "-O2", "-std=c++17", "-I", INCLUDE, JOB_BENCH_SRC, "-L", os.path.dirname(LIB), "-lbsc_mi355x", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", JOB_BENCH_EXE]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("job_bench build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)


if __name__ == "__main__":
    build(force="--force" in sys.argv, asan="--asan" in sys.argv)
