"""oracle/refbind.py — TEST INFRASTRUCTURE ONLY.

ctypes bindings for
  * oracle/_ref/libbsc_ref.so   — the real reference libbsc CPU path (built by oracle/Makefile
                                  from /root/reference, never copied), exported as ref_*;
  * oracle/_build/libbsc_oracle.so — the plain-C restatement (oracle/bsc_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libbsc_ref.so")
PORT_SO = os.path.join(_HERE, "_build", "libbsc_oracle.so")

u8p = C.POINTER(C.c_ubyte)
i32p = C.POINTER(C.c_int)


def _buf(b):
    """bytes/bytearray/numpy -> (ctypes pointer, keepalive)."""
    import numpy as np
    if isinstance(b, np.ndarray):
        assert b.dtype == np.uint8 and b.flags["C_CONTIGUOUS"]
        return b.ctypes.data_as(u8p), b
    if isinstance(b, (bytes, bytearray)):
        arr = (C.c_ubyte * len(b)).from_buffer_copy(b) if isinstance(b, bytes) else (C.c_ubyte * len(b)).from_buffer(b)
        return C.cast(arr, u8p), arr
    raise TypeError(type(b))


class Ref:
    """The compiled reference (libbsc 3.3.5 CPU path)."""

    def __init__(self, path=REF_SO, features=3):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` (needs /root/reference)")
        L = self.L = C.CDLL(path)
        L.ref_bsc_init.argtypes = [C.c_int]
        L.ref_bsc_compress.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_store.argtypes = [u8p, u8p, C.c_int, C.c_int]
        L.ref_bsc_block_info.argtypes = [u8p, C.c_int, i32p, i32p, C.c_int]
        L.ref_bsc_decompress.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int]
        L.ref_bsc_bwt_encode.argtypes = [u8p, C.c_int, u8p, i32p, C.c_int]
        L.ref_bsc_bwt_decode.argtypes = [u8p, C.c_int, C.c_int, C.c_ubyte, i32p, C.c_int]
        L.ref_bsc_st_encode.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_st_decode.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_coder_compress.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_coder_decompress.argtypes = [u8p, u8p, C.c_int, C.c_int]
        L.ref_bsc_qlfc_encode_block.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_qlfc_decode_block.argtypes = [u8p, u8p, C.c_int]
        L.ref_bsc_qlfc_transform.argtypes = [u8p, C.c_int, u8p, u8p]
        L.ref_bsc_adler32.argtypes = [u8p, C.c_int, C.c_int]
        L.ref_bsc_lzp_compress.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_lzp_decompress.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_bsc_adler32.restype = C.c_uint
        L.ref_wtime.restype = C.c_double
        self.features = features
        assert L.ref_bsc_init(features) == 0
        # The GPU box has 256 hardware threads; libgomp teams of that size are slow to start and can
        # fail to spawn under the container's pid limit.  Cap the reference's OpenMP team (callers that
        # time the reference set the count explicitly and report it).
        L.ref_omp_set_threads(min(os.cpu_count() or 1, int(os.environ.get("BSC_REF_THREADS", "16"))))

    # --- block API -------------------------------------------------------
    def compress(self, data, sorter=1, coder=1, lzp_hash=0, lzp_min=0, features=None):
        import numpy as np
        f = self.features if features is None else features
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = src.size
        out = np.empty(n + 28, dtype=np.uint8)
        r = self.L.ref_bsc_compress(src.ctypes.data_as(u8p), out.ctypes.data_as(u8p), n, lzp_hash, lzp_min, sorter, coder, f)
        if r < 0:
            return r
        return out[:r].tobytes()

    def decompress(self, block, features=None):
        import numpy as np
        f = self.features if features is None else features
        blk = np.frombuffer(bytes(block), dtype=np.uint8)
        bs, ds = C.c_int(), C.c_int()
        r = self.L.ref_bsc_block_info(blk.ctypes.data_as(u8p), blk.size, C.byref(bs), C.byref(ds), f)
        if r != 0:
            return r
        out = np.empty(max(ds.value, 1), dtype=np.uint8)
        r = self.L.ref_bsc_decompress(blk.ctypes.data_as(u8p), blk.size, out.ctypes.data_as(u8p), ds.value, f)
        if r != 0:
            return r
        return out[:ds.value].tobytes()

    # --- stage API -------------------------------------------------------
    def lzp_compress(self, data, hash_size, min_len, features=None):
        """bsc_lzp_compress (lzp.cpp:798) -> bytes, or the negative error code"""
        import numpy as np
        f = self.features if features is None else features
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = np.empty(src.size + 64, dtype=np.uint8)
        r = self.L.ref_bsc_lzp_compress(src.ctypes.data_as(u8p), out.ctypes.data_as(u8p), src.size, hash_size, min_len, f)
        return out[:r].tobytes() if r >= 0 else r

    def lzp_decompress(self, data, orig_size, hash_size, min_len, features=None):
        import numpy as np
        f = self.features if features is None else features
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = np.empty(orig_size + 64, dtype=np.uint8)
        r = self.L.ref_bsc_lzp_decompress(src.ctypes.data_as(u8p), out.ctypes.data_as(u8p), src.size, hash_size, min_len, f)
        return out[:r].tobytes() if r >= 0 else r

    def bwt_encode(self, data, aux=True, features=None):
        """-> (L bytes, primary index, [aux indexes])"""
        import numpy as np
        f = self.features if features is None else features
        T = np.frombuffer(bytes(data), dtype=np.uint8).copy() if not isinstance(data, np.ndarray) else data.copy()
        num = C.c_ubyte(0)
        idx = (C.c_int * 256)()
        if aux:
            r = self.L.ref_bsc_bwt_encode(T.ctypes.data_as(u8p), T.size, C.byref(num), idx, f)
        else:
            r = self.L.ref_bsc_bwt_encode(T.ctypes.data_as(u8p), T.size, None, None, f)
        return T, r, [idx[i] for i in range(num.value)]

    def st_encode(self, data, k, features=None):
        import numpy as np
        f = self.features if features is None else features
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        T = np.empty(src.size + 64, dtype=np.uint8)  # st.cpp:144 scribbles T[n..n+27]
        T[:src.size] = src
        r = self.L.ref_bsc_st_encode(T.ctypes.data_as(u8p), src.size, k, f)
        return T[:src.size].copy(), r

    def st_decode(self, data, k, index, features=None):
        import numpy as np
        f = self.features if features is None else features
        T = np.frombuffer(bytes(data), dtype=np.uint8).copy() if not isinstance(data, np.ndarray) else data.copy()
        r = self.L.ref_bsc_st_decode(T.ctypes.data_as(u8p), T.size, k, index, f)
        return T, r

    def bwt_decode(self, L, index, aux=(), features=None):
        import numpy as np
        f = self.features if features is None else features
        T = np.frombuffer(bytes(L), dtype=np.uint8).copy() if not isinstance(L, np.ndarray) else L.copy()
        idx = (C.c_int * 256)(*aux)
        r = self.L.ref_bsc_bwt_decode(T.ctypes.data_as(u8p), T.size, index, len(aux), idx, f)
        return T, r

    def coder_compress(self, data, coder=1, features=None):
        import numpy as np
        f = self.features if features is None else features
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = np.empty(src.size + 4096, dtype=np.uint8)
        r = self.L.ref_bsc_coder_compress(src.ctypes.data_as(u8p), out.ctypes.data_as(u8p), src.size, coder, f)
        return out[:r].tobytes() if r >= 0 else r

    def coder_decompress(self, data, n, coder=1, features=None):
        import numpy as np
        f = self.features if features is None else features
        src = np.frombuffer(bytes(data), dtype=np.uint8)
        out = np.empty(n + 64, dtype=np.uint8)
        r = self.L.ref_bsc_coder_decompress(src.ctypes.data_as(u8p), out.ctypes.data_as(u8p), coder, f)
        return out[:r].tobytes() if r >= 0 else r

    def qlfc_encode_block(self, data, coder=1, out_size=None):
        import numpy as np
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        osz = src.size if out_size is None else out_size
        out = np.empty(src.size + 4096, dtype=np.uint8)
        r = self.L.ref_bsc_qlfc_encode_block(src.ctypes.data_as(u8p), out.ctypes.data_as(u8p), src.size, osz, coder)
        return out[:r].tobytes() if r >= 0 else r

    def qlfc_transform(self, data):
        import numpy as np
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        ranks = np.empty(src.size + 64, dtype=np.uint8)
        mtf = np.empty(256, dtype=np.uint8)
        m = self.L.ref_bsc_qlfc_transform(src.ctypes.data_as(u8p), src.size, ranks.ctypes.data_as(u8p), mtf.ctypes.data_as(u8p))
        return ranks[:m].copy(), mtf

    def adler32(self, data):
        import numpy as np
        src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        return int(self.L.ref_bsc_adler32(src.ctypes.data_as(u8p), src.size, self.features))

    def max_threads(self):
        return int(self.L.ref_omp_max_threads())

    def set_threads(self, t):
        self.L.ref_omp_set_threads(int(t))


class Oracle:
    """The plain-C restatement (oracle/bsc_oracle.c).  Same method names as Ref where they overlap."""

    def __init__(self, path=PORT_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle port`")
        L = self.L = C.CDLL(path)
        L.orc_adler32.argtypes = [u8p, C.c_int]
        L.orc_adler32.restype = C.c_uint
        L.orc_bwt.argtypes = [u8p, C.c_int, u8p, i32p]
        L.orc_st.argtypes = [u8p, C.c_int, C.c_int]
        L.orc_qlfc_transform.argtypes = [u8p, u8p, C.c_int, u8p]
        L.orc_qlfc_encode.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.orc_coder_compress.argtypes = [u8p, u8p, C.c_int, C.c_int]
        L.orc_store.argtypes = [u8p, u8p, C.c_int]
        L.orc_compress.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
        L.orc_qlfc_static_pstream.argtypes = [u8p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        L.orc_qlfc_static_pstream.restype = C.c_long

    @staticmethod
    def _np(data, copy=False):
        import numpy as np
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        a = np.ascontiguousarray(a)
        return a.copy() if copy else a

    def static_pstream(self, data, counters=False):
        """trace of the static model over one sub-block -> (u16 entries, [n,3] counter values or None)"""
        import numpy as np
        a = self._np(data)
        cap = 16 * a.size + 64
        tr = np.empty(cap, np.uint16)
        ct = np.empty((cap, 3), np.uint16) if counters else None
        cnt = self.L.orc_qlfc_static_pstream(a.ctypes.data_as(u8p), a.size, tr.ctypes.data, ct.ctypes.data if counters else None, cap)
        assert 0 <= cnt <= cap
        return tr[:cnt], (ct[:cnt] if counters else None)

    def adler32(self, data):
        a = self._np(data)
        return int(self.L.orc_adler32(a.ctypes.data_as(u8p), a.size))

    def bwt_encode(self, data, aux=True):
        T = self._np(data, copy=True)
        num = C.c_ubyte(0)
        idx = (C.c_int * 256)()
        if aux:
            r = self.L.orc_bwt(T.ctypes.data_as(u8p), T.size, C.cast(C.byref(num), u8p), idx)
        else:
            r = self.L.orc_bwt(T.ctypes.data_as(u8p), T.size, None, None)
        return T, r, [idx[i] for i in range(num.value)]

    def st_encode(self, data, k):
        T = self._np(data, copy=True)
        r = self.L.orc_st(T.ctypes.data_as(u8p), T.size, k)
        return T, r

    def qlfc_transform(self, data):
        import numpy as np
        a = self._np(data)
        buf = np.empty(a.size + 16, np.uint8)
        mtf = np.empty(256, np.uint8)
        m = self.L.orc_qlfc_transform(a.ctypes.data_as(u8p), buf.ctypes.data_as(u8p), a.size, mtf.ctypes.data_as(u8p))
        return buf[a.size - m:a.size].copy(), mtf

    def qlfc_encode_block(self, data, coder=1, out_size=None):
        import numpy as np
        a = self._np(data)
        out = np.empty(a.size + 4096, np.uint8)
        r = self.L.orc_qlfc_encode(a.ctypes.data_as(u8p), out.ctypes.data_as(u8p), a.size, a.size if out_size is None else out_size, coder)
        return out[:r].tobytes() if r >= 0 else r

    def coder_compress(self, data, coder=1):
        import numpy as np
        a = self._np(data)
        out = np.empty(a.size + 4096, np.uint8)
        r = self.L.orc_coder_compress(a.ctypes.data_as(u8p), out.ctypes.data_as(u8p), a.size, coder)
        return out[:r].tobytes() if r >= 0 else r

    def compress(self, data, sorter=1, coder=1):
        import numpy as np
        a = self._np(data)
        out = np.empty(a.size + 28 + 4096, np.uint8)
        r = self.L.orc_compress(a.ctypes.data_as(u8p), out.ctypes.data_as(u8p), a.size, sorter, coder)
        return out[:r].tobytes() if r >= 0 else r
