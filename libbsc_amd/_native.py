"""ctypes bindings for libbsc_amd/lib/libbsc_mi355x.so (the C ABI declared in include/*.h).

The product path fails loudly when the native library is missing: there is no Python/CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSC_LIB_OVERRIDE") or os.path.join(HERE, "lib", "libbsc_mi355x.so")

u8p = C.POINTER(C.c_ubyte)
i32p = C.POINTER(C.c_int)
u32p = C.POINTER(C.c_uint32)

K_NAMES = ["radix_scatter", "radix_hist", "radix_scan", "pack", "seg", "gather", "emit", "misc",
           "dc_ctx", "dc_part", "dc_eval", "dc_pstream", "radix_hist_all", "radix_aux", "dc_static"]      # order of the BSCGPU_K_* enum (include/bscgpu.h)


class KStat(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_uint64), ("bytes", C.c_uint64), ("records", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; load it FIRST so that our DT_NEEDED "libamdhip64.so.7" binds to
    # the same runtime instead of pulling a second HIP runtime (/opt/rocm) into the process, which leaves
    # whichever loads second without a usable device.
    # BSC_NO_TORCH=1 (sanitizer runs, tools/asan_run.py): ROCm's ASan runtime intercepts the HSA allocation calls and dlopen()s
    # libhsa-runtime64.so itself; with torch's private copy of the runtime in the process that is a second, uninitialised HSA.
    if not os.environ.get("BSC_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - pure-C deployments have no torch; /opt/rocm's runtime is used
            pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"native library {LIB_PATH} is missing — build it with `python -m libbsc_amd.build` "
            "(hipcc --offload-arch=gfx950). libbsc_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.bscgpu_device_count.restype = C.c_int
    L.bscgpu_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int64]
    L.bscgpu_destroy.argtypes = [vp]
    L.bscgpu_destroy.restype = None
    L.bscgpu_arena_bytes.argtypes = [vp]
    L.bscgpu_arena_bytes.restype = C.c_int64
    L.bscgpu_bwt.argtypes = [vp, vp, vp, C.c_int64]
    L.bscgpu_bwt.restype = C.c_int64
    L.bscgpu_bwt_aux.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, u32p]
    L.bscgpu_bwt_aux.restype = C.c_int64
    L.bscgpu_bwt_device.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, u32p]
    L.bscgpu_bwt_device.restype = C.c_int64
    L.bscgpu_st_encode.argtypes = [vp, vp, C.c_int, C.c_int]
    L.bscgpu_st_encode_device.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.bscgpu_adler32_device.argtypes = [vp, vp, C.c_int64, u32p]
    L.bscgpu_radix_sort_u64.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, i32p]
    L.bscgpu_profile_enable.argtypes = [vp, C.c_int]
    L.bscgpu_profile_enable.restype = None
    L.bscgpu_profile_reset.argtypes = [vp]
    L.bscgpu_profile_reset.restype = None
    L.bscgpu_profile_get.argtypes = [vp, C.POINTER(KStat)]
    L.bscgpu_profile_scatter_launches.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.bscgpu_last_stage_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.bscgpu_last_error.argtypes = [vp]
    L.bscgpu_option_set.argtypes = [vp, C.c_int, C.c_int]
    L.bscgpu_option_get.argtypes = [vp, C.c_int]
    L.bscgpu_last_error.restype = C.c_char_p
    L.bscgpu_pipe_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.bscgpu_pipe_destroy.argtypes = [vp]
    L.bscgpu_pipe_destroy.restype = None
    L.bscgpu_coder_pool_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.bscgpu_coder_pool_stats.restype = None
    L.bscgpu_pipe_submit.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.bscgpu_pipe_wait.argtypes = [vp, C.c_int]
    L.bscgpu_pipe_submit_host.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    if hasattr(L, "bscgpu_compress_device"):
        L.bscgpu_compress_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L


def np_ptr(a):
    return C.c_void_p(a.ctypes.data)
