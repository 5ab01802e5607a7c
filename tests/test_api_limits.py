"""Block-size limits of the public entry points (libbsc.cpp:124, :221, :259), checked before any GPU work: no GPU needed."""
import numpy as np

from libbsc_amd import api, _native as N


def test_bsc_compress_size_limits():
    L = api._L()
    big = np.empty((1 << 30) + 64, np.uint8)           # never touched: both calls return before reading it
    out = np.empty(64, np.uint8)
    # separate buffers: the format's maximum is 1 GiB (libbsc.cpp:221)
    assert L.bsc_compress(N.np_ptr(big), N.np_ptr(out), (1 << 30) + 1, 0, 0, 1, 1, 3) == api.BAD_PARAMETER
    # in place: the reference takes up to 2047 MiB (libbsc.cpp:124); this library declines above 1 GiB instead of running untested shapes
    assert L.bsc_compress(N.np_ptr(big), N.np_ptr(big), (1 << 30) + 1, 0, 0, 1, 1, 3) == api.NOT_SUPPORTED
    assert L.bsc_compress(N.np_ptr(big), N.np_ptr(big), 2146435073, 0, 0, 1, 1, 3) == api.BAD_PARAMETER
    assert L.bsc_compress(N.np_ptr(big), N.np_ptr(big), -1, 0, 0, 1, 1, 3) == api.BAD_PARAMETER


def test_bsc_bwt_encode_small_blocks_without_gpu():
    """n = 0 and the aux-rate rule are answered before a device is needed (bwt.cpp:178-231; libsais_bwt_aux refuses r < 2)."""
    T, r, I = api.bsc_bwt_encode(np.zeros(0, np.uint8), aux=True)
    assert r == api.BAD_PARAMETER
    T, r, I = api.bsc_bwt_encode(np.zeros(0, np.uint8), aux=False)
    assert r == 0
    T, r, I = api.bsc_bwt_encode(np.arange(15, dtype=np.uint8), aux=True)
    assert r == api.BAD_PARAMETER and np.array_equal(T, np.arange(15, dtype=np.uint8))
