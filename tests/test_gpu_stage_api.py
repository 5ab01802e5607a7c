"""GPU parity tests for the exported STAGE entry points the north star names: bsc_bwt_encode (bwt.h:45-56, bwt.cpp:178-231) and
bsc_st_encode (st.h:47-68, st.cpp:990-1012), called through ctypes on the product library exactly as the reference's callers
call them (in place on the caller's buffer, default GPU context inside the library), against the compiled reference:
(L, primary index, num_indexes, indexes[]) for the BWT, (L, index) for the sort transforms.  Edge sizes 0, 1, 15, 16, 17 (the
aux-rate rule: r = pow2floor(n / 8) < 2 is a BAD_PARAMETER in the reference's libsais call) included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch


def _cases():
    from test_gpu_device import _corpus
    rng = np.random.default_rng(23)
    cases = [("empty", np.zeros(0, np.uint8))]
    cases += _corpus(rng)
    return cases


def test_bsc_bwt_encode_symbol_matches_reference(torch_cuda, ref):
    from libbsc_amd import api
    bad = []
    for name, T in _cases():
        n = T.size
        for aux in (True, False):
            L, idx, I = api.bsc_bwt_encode(T, aux=aux)
            wL, widx, wI = ref.bwt_encode(T, aux=aux)
            # n < 16 with indexes asked for: both refuse (-1) and leave the text alone
            if not (idx == widx and list(I) == list(wI) and np.array_equal(L, wL)):
                bad.append((name, n, aux, idx, widx, len(I), len(wI)))
    assert not bad, bad


def test_bsc_bwt_encode_aux_rate_rule(torch_cuda, ref):
    """num_indexes = (n - 1) / r with r = pow2floor(n / 8): the sizes either side of a change of r, and the refusal below 16."""
    from libbsc_amd import api
    rng = np.random.default_rng(5)
    for n in [8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 4096 * 8 - 1, 4096 * 8, 4096 * 8 + 1]:
        T = rng.integers(0, 4, n, dtype=np.uint8)
        L, idx, I = api.bsc_bwt_encode(T, aux=True)
        wL, widx, wI = ref.bwt_encode(T, aux=True)
        assert idx == widx and list(I) == list(wI) and np.array_equal(L, wL), (n, idx, widx, len(I), len(wI))
        if n < 16:
            assert idx == api.BAD_PARAMETER and np.array_equal(L, T)


@pytest.mark.parametrize("k", [3, 4, 5, 6, 7, 8])
def test_bsc_st_encode_symbol_matches_reference(torch_cuda, ref, k):
    from libbsc_amd import api
    bad = []
    for name, T in _cases():
        n = T.size
        out, idx = api.bsc_st_encode(T, k)
        if k <= 6:
            want, widx = ref.st_encode(T, k)
            if not (idx == widx and np.array_equal(out, want)):
                bad.append((name, n, idx, widx))
        elif n >= 2:    # the reference's CPU encoder stops at k = 6 (st.cpp:1004-1009); its decoder (st.cpp:1491) is the judge
            back, rc = ref.st_decode(out, k, idx)
            if rc != 0 or not np.array_equal(back, T):
                bad.append((name, n, idx, rc))
        else:
            if idx != 0 or not np.array_equal(out, T):
                bad.append((name, n, idx, "n <= 1 must return 0 and leave T alone"))
    assert not bad, bad


def test_stage_entry_points_reject_bad_parameters(torch_cuda):
    """bwt.cpp / st.cpp parameter checks: k outside 3..8, negative n, null text."""
    import ctypes as C
    from libbsc_amd import api, _native as N
    L = api._L()
    T = np.zeros(64, np.uint8)
    assert L.bsc_st_encode(N.np_ptr(T), 64, 2, 3) == api.BAD_PARAMETER
    assert L.bsc_st_encode(N.np_ptr(T), 64, 9, 3) == api.BAD_PARAMETER
    assert L.bsc_st_encode(N.np_ptr(T), -1, 5, 3) == api.BAD_PARAMETER
    assert L.bsc_st_encode(None, 64, 5, 3) == api.BAD_PARAMETER
    assert L.bsc_bwt_encode(None, 64, None, None, 3) == api.BAD_PARAMETER
    assert L.bsc_bwt_encode(N.np_ptr(T), -1, None, None, 3) == api.BAD_PARAMETER
