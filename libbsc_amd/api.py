"""Host-side mirror of the reference's C API (libbsc/libbsc.h:95-152 and the stage headers), bound to
libbsc_amd/lib/libbsc_mi355x.so.  Same names, same argument meaning, same error codes, so tests read like
calls into the reference.  Buffers are numpy uint8 arrays / bytes."""
import ctypes as C

import numpy as np

from . import _native as N

NO_ERROR, BAD_PARAMETER, NOT_ENOUGH_MEMORY, NOT_COMPRESSIBLE, NOT_SUPPORTED = 0, -1, -2, -3, -4
UNEXPECTED_EOB, DATA_CORRUPT, GPU_ERROR, GPU_NOT_SUPPORTED, GPU_NOT_ENOUGH_MEMORY = -5, -6, -7, -8, -9
BLOCKSORTER_BWT = 1
CODER_QLFC_STATIC, CODER_QLFC_ADAPTIVE, CODER_QLFC_FAST = 1, 2, 3
FEATURE_FASTMODE, FEATURE_MULTITHREADING, FEATURE_GPU = 1, 2, 8
HEADER_SIZE = 28

_bound = False


def _L():
    global _bound
    L = N.lib()
    if not _bound:
        vp, ci = C.c_void_p, C.c_int
        L.bsc_init.argtypes = [ci]
        L.bsc_compress.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci]
        L.bsc_store.argtypes = [vp, vp, ci, ci]
        L.bsc_block_info.argtypes = [vp, ci, N.i32p, N.i32p, ci]
        L.bsc_decompress.argtypes = [vp, ci, vp, ci, ci]
        L.bsc_bwt_encode.argtypes = [vp, ci, N.u8p, N.i32p, ci]
        L.bsc_bwt_decode.argtypes = [vp, ci, ci, C.c_ubyte, N.i32p, ci]
        L.bsc_st_encode.argtypes = [vp, ci, ci, ci]
        L.bsc_st_decode.argtypes = [vp, ci, ci, ci, ci]
        L.bsc_coder_compress.argtypes = [vp, vp, ci, ci, ci]
        L.bsc_coder_decompress.argtypes = [vp, vp, ci, ci]
        L.bsc_adler32.argtypes = [vp, ci, ci]
        L.bsc_adler32.restype = C.c_uint
        L.bsc_qlfc_encode_block.argtypes = [vp, vp, ci, ci, ci]
        L.bsc_qlfc_decode_block.argtypes = [vp, vp, ci]
        L.bsc_qlfc_ranks.argtypes = [vp, ci, vp, vp, N.i32p]
        L.bsc_synth_text_v1.argtypes = [C.c_ulonglong, vp, C.c_longlong]
        L.bsc_init(3)
        _bound = True
    return L


def _arr(data, copy=False):
    a = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data
    a = np.ascontiguousarray(a)
    return a.copy() if copy else a


def bsc_adler32(data):
    a = _arr(data)
    return int(_L().bsc_adler32(N.np_ptr(a), a.size, 0))


def bsc_coder_compress(data, coder=CODER_QLFC_STATIC, features=3):
    a = _arr(data)
    out = np.empty(a.size + 4096, np.uint8)
    r = _L().bsc_coder_compress(N.np_ptr(a), N.np_ptr(out), a.size, coder, features)
    return out[:r].tobytes() if r >= 0 else r


def bsc_coder_decompress(data, n, coder=CODER_QLFC_STATIC, features=3):
    a = _arr(data)
    out = np.empty(n + 64, np.uint8)
    r = _L().bsc_coder_decompress(N.np_ptr(a), N.np_ptr(out), coder, features)
    return out[:r].tobytes() if r >= 0 else r


def bsc_qlfc_encode_block(data, coder=CODER_QLFC_STATIC, out_size=None):
    a = _arr(data)
    out = np.empty(a.size + 4096, np.uint8)
    r = _L().bsc_qlfc_encode_block(N.np_ptr(a), N.np_ptr(out), a.size, a.size if out_size is None else out_size, coder)
    return out[:r].tobytes() if r >= 0 else r


def bsc_qlfc_decode_block(data, n, coder=CODER_QLFC_STATIC):
    a = _arr(data)
    out = np.empty(n + 64, np.uint8)
    r = _L().bsc_qlfc_decode_block(N.np_ptr(a), N.np_ptr(out), coder)
    return out[:r].tobytes() if r >= 0 else r


def bsc_bwt_decode(L, index, aux=(), features=3):
    T = _arr(L, copy=True)
    idx = (C.c_int * 256)(*aux)
    r = _L().bsc_bwt_decode(N.np_ptr(T), T.size, index, len(aux), idx, features)
    return T, r


def bsc_st_decode(data, k, index, features=3):
    T = _arr(data, copy=True)
    r = _L().bsc_st_decode(N.np_ptr(T), T.size, k, index, features)
    return T, r


def bsc_qlfc_ranks(data):
    a = _arr(data)
    ranks = np.empty(a.size + 8, np.uint8)
    first = np.empty(256, np.uint8)
    k = C.c_int(0)
    m = _L().bsc_qlfc_ranks(N.np_ptr(a), a.size, N.np_ptr(ranks), N.np_ptr(first), C.byref(k))
    return ranks[:m].copy(), first[:k.value].copy()


def bsc_store(data, features=3):
    a = _arr(data)
    out = np.empty(a.size + HEADER_SIZE, np.uint8)
    r = _L().bsc_store(N.np_ptr(a), N.np_ptr(out), a.size, features)
    return out[:r].tobytes()


def bsc_compress(data, sorter=BLOCKSORTER_BWT, coder=CODER_QLFC_STATIC, lzp_hash=0, lzp_min=0, features=3, inplace=False):
    a = _arr(data)
    if inplace:
        buf = np.empty(a.size + HEADER_SIZE, np.uint8)
        buf[:a.size] = a
        r = _L().bsc_compress(N.np_ptr(buf), N.np_ptr(buf), a.size, lzp_hash, lzp_min, sorter, coder, features)
        return buf[:r].tobytes() if r >= 0 else r
    out = np.empty(a.size + HEADER_SIZE, np.uint8)
    r = _L().bsc_compress(N.np_ptr(a), N.np_ptr(out), a.size, lzp_hash, lzp_min, sorter, coder, features)
    return out[:r].tobytes() if r >= 0 else r


def bsc_lzp_compress(data, hash_size, min_len, features=3):
    """LZP stage (lzp/lzp.h:50): -> bytes or the negative error code (LIBBSC_NOT_COMPRESSIBLE = -3)."""
    a = _arr(data)
    out = np.empty(a.size + 64, np.uint8)
    r = _L().bsc_lzp_compress(N.np_ptr(a), N.np_ptr(out), a.size, hash_size, min_len, features)
    return out[:r].tobytes() if r >= 0 else r


def bsc_lzp_decompress(data, orig_size, hash_size, min_len, features=3):
    a = _arr(data)
    out = np.empty(orig_size + 64, np.uint8)
    r = _L().bsc_lzp_decompress(N.np_ptr(a), N.np_ptr(out), a.size, hash_size, min_len, features)
    return out[:r].tobytes() if r >= 0 else r


def bsc_block_info(block, features=3):
    a = _arr(block)
    bs, ds = C.c_int(), C.c_int()
    r = _L().bsc_block_info(N.np_ptr(a), a.size, C.byref(bs), C.byref(ds), features)
    return r, bs.value, ds.value


def bsc_decompress(block, features=3):
    a = _arr(block)
    r, bs, ds = bsc_block_info(a, features)
    if r != 0:
        return r
    out = np.empty(max(ds, 1), np.uint8)
    r = _L().bsc_decompress(N.np_ptr(a), a.size, N.np_ptr(out), ds, features)
    return out[:ds].tobytes() if r == 0 else r


def bsc_bwt_encode(data, aux=True, features=3):
    T = _arr(data, copy=True)
    num = C.c_ubyte(0)
    idx = (C.c_int * 256)()
    if aux:
        r = _L().bsc_bwt_encode(N.np_ptr(T), T.size, C.byref(num), idx, features)
    else:
        r = _L().bsc_bwt_encode(N.np_ptr(T), T.size, None, None, features)
    return T, r, [idx[i] for i in range(num.value)]


def bsc_st_encode(data, k, features=3):
    T = _arr(data, copy=True)
    r = _L().bsc_st_encode(N.np_ptr(T), T.size, k, features)
    return T, r


def synth_text_v1(seed, n):
    """`synth-text v1` through the native generator (fast path of libbsc_amd.synth.synth_text_v1)."""
    out = np.empty(n, np.uint8)
    r = _L().bsc_synth_text_v1(seed, N.np_ptr(out), n)
    if r != 0:
        raise ValueError(r)
    return out
