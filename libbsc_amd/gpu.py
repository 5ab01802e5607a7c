"""Per-GPU context: thin object wrapper over the bscgpu_* C ABI (include/bscgpu.h)."""
import ctypes as C

import numpy as np

from . import _native as N


class GpuError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"bscgpu error {code}: {msg}")
        self.code = code


def _dptr(t):
    """device pointer of a torch CUDA tensor (or a raw int).  The context runs on its own non-blocking HIP stream,
    so pending torch work that produces the tensor (an async clone / copy on torch's stream) must have finished:
    synchronise torch's current stream before handing the pointer over."""
    if isinstance(t, int):
        return C.c_void_p(t)
    import torch
    torch.cuda.current_stream(t.device).synchronize()
    return C.c_void_p(t.data_ptr())


class GpuContext:
    """One context per GPU (mirrors libcubwt's device storage handle, libcubwt.cuh:60-71)."""

    def __init__(self, device=0, max_n=64 << 20):
        self.L = N.lib()
        h = C.c_void_p()
        rc = self.L.bscgpu_create(C.byref(h), int(device), int(max_n))
        if rc != 0:
            raise GpuError(rc, "bscgpu_create failed (no GPU / out of memory?)")
        self.h = h
        self.device = device
        self.max_n = max_n

    def close(self):
        if getattr(self, "h", None):
            self.L.bscgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise GpuError(rc, (self.L.bscgpu_last_error(self.h) or b"").decode())
        return rc

    @property
    def arena_bytes(self):
        return int(self.L.bscgpu_arena_bytes(self.h))

    # ---- host-pointer entry points (shape of the reference hooks) -----------------------------
    def bwt(self, data, aux_rate=None):
        """-> (L np.uint8[n], primary index, I list or None).  libcubwt_bwt / libcubwt_bwt_aux."""
        T = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data)
        n = T.size
        Lout = np.empty(max(n, 1), np.uint8)
        if aux_rate:
            cnt = (n - 1) // aux_rate + 1 if n > 0 else 0
            I = (C.c_uint32 * max(cnt, 1))()
            rc = self.L.bscgpu_bwt_aux(self.h, N.np_ptr(T), N.np_ptr(Lout), n, aux_rate, I)
            self._check(rc)
            return Lout[:n], int(I[0]) if cnt else 0, [int(I[t]) for t in range(cnt)]
        rc = self._check(self.L.bscgpu_bwt(self.h, N.np_ptr(T), N.np_ptr(Lout), n))
        return Lout[:n], int(rc), None

    def unbwt(self, L, index):
        """bscgpu_unbwt: inverse BWT of L (np.uint8) with the 1-based primary index -> (text np.uint8, rc)"""
        a = np.ascontiguousarray(L, dtype=np.uint8)
        out = np.empty(max(a.size, 1), np.uint8)
        f = self.L.bscgpu_unbwt
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        rc = f(self.h, N.np_ptr(a), N.np_ptr(out), a.size, int(index))
        return out[:a.size], int(rc)

    def st_encode(self, data, k):
        T = np.array(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data, copy=True)
        rc = self._check(self.L.bscgpu_st_encode(self.h, N.np_ptr(T), T.size, k))
        return T, int(rc)

    # ---- device-pointer entry points -------------------------------------------------------
    def bwt_device(self, dT, dL, n, aux_rate=None):
        if aux_rate:
            cnt = (n - 1) // aux_rate + 1
            I = (C.c_uint32 * cnt)()
            rc = self._check(self.L.bscgpu_bwt_device(self.h, _dptr(dT), _dptr(dL), n, aux_rate, I))
            return int(rc), [int(I[t]) for t in range(cnt)]
        rc = self._check(self.L.bscgpu_bwt_device(self.h, _dptr(dT), _dptr(dL), n, 0, None))
        return int(rc), None

    def st_encode_device(self, dT, dOut, n, k):
        return int(self._check(self.L.bscgpu_st_encode_device(self.h, _dptr(dT), _dptr(dOut), n, k)))

    def adler32_device(self, dT, n):
        out = C.c_uint32(0)
        self._check(self.L.bscgpu_adler32_device(self.h, _dptr(dT), n, C.byref(out)))
        return int(out.value)

    def radix_sort(self, keys, keys_alt, vals, vals_alt, n, begin_bit=0, end_bit=64):
        """torch int64 / int32 CUDA tensors (bit patterns are treated as unsigned). Returns (keys, vals) tensors
        that hold the result."""
        alt = C.c_int(0)
        self._check(self.L.bscgpu_radix_sort_u64(self.h, _dptr(keys), _dptr(keys_alt),
                                                 _dptr(vals) if vals is not None else None,
                                                 _dptr(vals_alt) if vals_alt is not None else None,
                                                 n, begin_bit, end_bit, C.byref(alt)))
        return (keys_alt, vals_alt) if alt.value else (keys, vals)

    def qlfc_static_pstream(self, L, debug=False):
        """bscgpu_qlfc_static_pstream: (entries u16[D], sub_start, sub_size, poff[nb+1], dbg [3,D] or None); raises GpuError
        with code -4 when the block has to take the host model."""
        a = np.ascontiguousarray(L, dtype=np.uint8)
        cap = 16 * a.size + 65536               # decisions: ~3 per byte on text, ~9 on random bytes
        out = np.empty(cap, np.uint16)
        dbg = np.empty((3, cap), np.uint16) if debug else None
        nb = C.c_int(0); st = (C.c_int * 8)(); sz = (C.c_int * 8)(); poff = (C.c_int64 * 9)()
        f = self.L.bscgpu_qlfc_static_pstream
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        D = f(self.h, N.np_ptr(a), a.size, out.ctypes.data, cap, C.byref(nb), st, sz, poff, dbg.ctypes.data if debug else None)
        self._check(D)
        if D > cap:
            raise GpuError(-2, f"{D} decisions exceed the buffer of {cap}")
        k = nb.value
        return out[:D], list(st[:k]), list(sz[:k]), list(poff[:k + 1]), (dbg[:, :D] if debug else None)

    def qlfc_static_pstream_packed(self, L):
        """bscgpu_qlfc_static_pstream_packed: (fields u16[D] unpacked from the 13-bit stream, sub_start, sub_size, poff, pbase, packed bytes);
        raises GpuError -4 when the block takes the host model or the packed form was not produced."""
        a = np.ascontiguousarray(L, dtype=np.uint8)
        cap = 26 * a.size + 65536 + 8 * 104
        out = np.zeros(cap + 8, np.uint8)
        nb = C.c_int(0); st = (C.c_int * 8)(); sz = (C.c_int * 8)(); poff = (C.c_int64 * 9)(); pbase = (C.c_int64 * 9)()
        f = self.L.bscgpu_qlfc_static_pstream_packed
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        D = f(self.h, N.np_ptr(a), a.size, out.ctypes.data, cap, C.byref(nb), st, sz, poff, pbase)
        self._check(D)
        k = nb.value
        po, pb = list(poff[:k + 1]), list(pbase[:k + 1])
        nbytes = pb[k] // 8 * 13
        if nbytes > cap:
            raise GpuError(-2, f"{nbytes} bytes exceed the buffer of {cap}")
        fields = np.empty(D, np.uint16)
        for b in range(k):                                         # field i of sub-block b: bits [13 i, 13 i + 13) from byte pbase[b] / 8 * 13 on
            cnt = po[b + 1] - po[b]
            base = pb[b] // 8 * 13
            bit = np.arange(cnt, dtype=np.int64) * 13
            byte = base + (bit >> 3)
            w = out[byte].astype(np.uint32) | (out[byte + 1].astype(np.uint32) << 8) | (out[byte + 2].astype(np.uint32) << 16)
            fields[po[b]:po[b + 1]] = ((w >> (bit & 7).astype(np.uint32)) & 0x1fff).astype(np.uint16)
        return fields, list(st[:k]), list(sz[:k]), po, pb, out[:nbytes]

    def compress_device(self, dInput, n, sorter=1, coder=1, features=3):
        out = np.empty(n + 28, np.uint8)
        rc = self._check(self.L.bscgpu_compress_device(self.h, _dptr(dInput), N.np_ptr(out), n, sorter, coder, features))
        return out[:rc]

    def pipe(self, depth=2, reuse_outputs=False):
        return Pipe(self, depth, reuse_outputs)

    # ---- measurement / test knobs (include/bscgpu.h: BSCGPU_OPT_*, BSCGPU_CNT_*) ---------------------------------
    OPT_RS_ONESWEEP, CNT_OS_RETRIES, OPT_DC_STREAM_STATIC, OPT_DC_PACKED_STREAM = 1, 2, 3, 4

    def option_set(self, key, value):
        return self._check(self.L.bscgpu_option_set(self.h, key, value))

    def option_get(self, key):
        return self._check(self.L.bscgpu_option_get(self.h, key))

    # ---- profiling ---------------------------------------------------------------------------
    def profile(self, on=True):
        self.L.bscgpu_profile_enable(self.h, 1 if on else 0)

    def profile_reset(self):
        self.L.bscgpu_profile_reset(self.h)

    def profile_get(self):
        arr = (N.KStat * len(N.K_NAMES))()
        self.L.bscgpu_profile_get(self.h, arr)
        return {N.K_NAMES[i]: dict(ms=arr[i].ms, launches=int(arr[i].launches), bytes=int(arr[i].bytes),
                                   records=int(arr[i].records)) for i in range(len(N.K_NAMES))}

    def scatter_launches(self, max_n=4096):
        ms = (C.c_double * max_n)()
        rec = (C.c_uint64 * max_n)()
        cnt = self.L.bscgpu_profile_scatter_launches(self.h, ms, rec, max_n)
        return [(ms[i], int(rec[i])) for i in range(cnt)]

    def last_stage_ms(self):
        out = (C.c_double * 6)()
        self.L.bscgpu_last_stage_ms(self.h, out)
        return list(out)


def coder_pool_stats(reset=False):
    """how the process's coder pool has coded the pipes' blocks: {scalar x8 tasks, pairs x4 tasks, eight_lanes x1 task, host_model}"""
    out = (C.c_uint64 * 4)()
    N.lib().bscgpu_coder_pool_stats(out, 1 if reset else 0)
    L = N.lib()
    L.bscgpu_coder_pool_x16_blocks.restype = C.c_uint64
    x16 = int(L.bscgpu_coder_pool_x16_blocks(1 if reset else 0))
    return {"scalar_tasks": int(out[0]), "pair_tasks": int(out[1]), "eight_lane_task": int(out[2]), "host_model": int(out[3]),
            "of_the_eight_lane_blocks_coded_in_pairs_of_sixteen_lanes": x16}


class Pipe:
    """Several blocks in flight on one GPU (bscgpu_pipe_*): submit() runs the GPU stage, the host coder of that
    block runs on worker threads while the next block is sorted."""

    def __init__(self, ctx, depth=2, reuse_outputs=False):
        # reuse_outputs: wait() returns a view into one of `depth` recycled output buffers, valid until `depth` further
        # submits (no 64 MiB allocation and no page faults per block); default: a fresh array per block
        self.reuse = reuse_outputs
        self._pool = {}
        self._seq = 0
        self.ctx = ctx
        self.L = ctx.L
        h = C.c_void_p()
        ctx._check(self.L.bscgpu_pipe_create(ctx.h, depth, C.byref(h)))
        self.h = h
        self.depth = depth
        self._out = {}

    def _new_out(self, n):
        if not self.reuse:
            return np.empty(n + 28, np.uint8)
        k = self._seq % self.depth
        self._seq += 1
        buf = self._pool.get(k)
        if buf is None or buf.size < n + 28:
            buf = self._pool[k] = np.empty(n + 28, np.uint8)
        return buf[:n + 28]

    def submit(self, dInput, n, sorter=1, coder=1, features=3):
        out = self._new_out(n)
        t = self.ctx._check(self.L.bscgpu_pipe_submit(self.h, _dptr(dInput), N.np_ptr(out), n, sorter, coder, features))
        self._out[t] = (out, dInput)          # keep both alive until wait()
        return t

    def submit_host(self, data, sorter=1, coder=1, lzp_hash=0, lzp_min=0, features=3):
        """Host-resident block (np.uint8) with bsc_compress's full parameter list, LZP included."""
        a = np.ascontiguousarray(data, dtype=np.uint8)
        out = self._new_out(a.size)
        t = self.ctx._check(self.L.bscgpu_pipe_submit_host(self.h, N.np_ptr(a), N.np_ptr(out), a.size, lzp_hash, lzp_min,
                                                            sorter, coder, features))
        self._out[t] = (out, a)
        return t

    def wait(self, ticket):
        rc = self.ctx._check(self.L.bscgpu_pipe_wait(self.h, ticket))
        out, _ = self._out.pop(ticket)
        return out[:rc]

    def close(self):
        if getattr(self, "h", None):
            self.L.bscgpu_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
