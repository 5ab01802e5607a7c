"""GPU parity tests for the device layer, through the C ABI (include/bscgpu.h).

Oracle = the compiled reference (oracle/_ref) for BWT / ST, numpy stable sort for the radix engine,
zlib for Adler-32.  Bit-exact everywhere (integer / byte / index work).
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch


@pytest.fixture(scope="module")
def ctx(torch_cuda):
    from libbsc_amd import GpuContext
    c = GpuContext(0, max_n=(16 << 20) + 4096)
    yield c
    c.close()


def _aux_rate(n):
    m = n // 8
    if m == 0:
        return 1
    return 1 << (m.bit_length() - 1)


def _corpus(rng):
    from libbsc_amd.synth import synth_text_v1
    cases = []
    for n in [1, 2, 3, 7, 8, 9, 15, 16, 17, 29, 64, 255, 1000, 4095, 4096, 4097, 65535, 65536, 100003]:
        cases.append(("rand256", rng.integers(0, 256, n, dtype=np.uint8)))
        cases.append(("rand2", rng.integers(0, 2, n, dtype=np.uint8)))
        cases.append(("zeros", np.zeros(n, np.uint8)))
        cases.append(("ff", np.full(n, 255, np.uint8)))
        cases.append(("ab", (np.arange(n) % 2).astype(np.uint8)))
        x = rng.integers(0, 3, n, dtype=np.uint8)
        x[-min(n, 9):] = 0
        cases.append(("zero-tail", x))
    fib = [b"a", b"ab"]
    while len(fib[-1]) < 200000:
        fib.append(fib[-1] + fib[-2])
    cases.append(("fibonacci", np.frombuffer(fib[-1], dtype=np.uint8).copy()))
    cases.append(("text64k", synth_text_v1(5, 1 << 16)))
    cases.append(("text1m", synth_text_v1(1, 1 << 20)))
    rep = synth_text_v1(7, 50000)
    cases.append(("repeated-passage", np.concatenate([rep] * 8 + [rng.integers(0, 256, 1000, dtype=np.uint8)])))
    return cases


# --------------------------------------------------------------------------------------------
# Single-read digit passes (radix_onesweep.hip): BSC_RS_ONESWEEP=2 sends every sort of >= 4 tiles through them (the variable is
# read when a context is created).  Oracle = numpy's stable sort.
@pytest.fixture(scope="module")
def os_ctx(torch_cuda):
    import os
    from libbsc_amd import GpuContext
    old = os.environ.get("BSC_RS_ONESWEEP")
    os.environ["BSC_RS_ONESWEEP"] = "2"
    try:
        c = GpuContext(0, max_n=(24 << 20) + 4096)
    finally:
        if old is None:
            os.environ.pop("BSC_RS_ONESWEEP", None)
        else:
            os.environ["BSC_RS_ONESWEEP"] = old
    yield c
    c.close()


def _os_keys(rng, n, kind):
    if kind == "uniform":
        return rng.integers(0, 2**63, n, dtype=np.int64).view(np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    if kind == "skewed":            # few distinct bytes per digit, like text
        kb = rng.integers(0, 7, (n, 8), dtype=np.uint8) * 37
        kb[:, 3] = rng.integers(0, 256, n, dtype=np.uint8)
        return kb.view(np.uint64).reshape(-1).copy()
    if kind == "one-digit":         # every record in one bucket of the low passes, two buckets in the top one
        return (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)) | np.uint64(0x0101010101010101)
    # "sorted-runs": long runs of equal digits, the tile counts of a digit swing between 0 and a whole tile
    return (np.arange(n, dtype=np.uint64) // np.uint64(5000)) * np.uint64(0x0000010000010001)


def _os_check(torch, ctx, keys, pairs, b0, b1):
    n = keys.size
    vals = np.arange(n, dtype=np.uint32)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    dk2 = torch.empty_like(dk)
    dv = torch.from_numpy(vals.view(np.int32)).cuda() if pairs else None
    dv2 = torch.empty_like(dv) if pairs else None
    rk, rv = ctx.radix_sort(dk, dk2, dv, dv2, n, b0, b1)
    torch.cuda.synchronize()
    mask = np.uint64((((1 << (b1 - b0)) - 1) << b0) & 0xFFFFFFFFFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    assert np.array_equal(rk.cpu().numpy().view(np.uint64), keys[order]), f"keys mismatch n={n} bits=[{b0},{b1})"
    if pairs:
        assert np.array_equal(rv.cpu().numpy().view(np.uint32), vals[order]), f"vals mismatch n={n} bits=[{b0},{b1})"


@pytest.mark.parametrize("n", [4 * 7680, 4 * 7680 + 1, 40000, 65 * 7680 + 5, 64 * 7680, 64 * 9 * 7680 + 17, 5_000_000, 20_000_003])
@pytest.mark.parametrize("mode", ["pairs", "keys"])
def test_single_read_passes_match_stable_sort(os_ctx, torch_cuda, n, mode):
    rng = np.random.default_rng(n + 5)
    for kind, ranges in (("uniform", [(0, 64), (3, 30)]), ("skewed", [(0, 64), (8, 48), (32, 58)]), ("one-digit", [(0, 64)]), ("sorted-runs", [(0, 40)])):
        keys = _os_keys(rng, n, kind)
        for (b0, b1) in ranges:
            if n > 10_000_000 and (b0, b1) != ranges[0]:
                continue
            _os_check(torch_cuda, os_ctx, keys, mode == "pairs", b0, b1)


def test_single_read_passes_launch_tag_wraps(os_ctx, torch_cuda):
    """The tile rows carry a launch tag of 1..255 and are cleared when the sequence wraps: > 255 digit passes on one context,
    alternating a small and a larger input so that rows beyond the small input keep older tags."""
    rng = np.random.default_rng(99)
    small = _os_keys(rng, 5 * 7680 + 11, "skewed")
    large = _os_keys(rng, 70 * 7680 + 3, "uniform")
    for it in range(40):                                 # 40 x 8 = 320 passes
        _os_check(torch_cuda, os_ctx, large if it % 5 == 4 else small, True, 0, 64)


def test_single_read_passes_under_uneven_load(os_ctx, torch_cuda):
    """Look-back under partial residency and uneven load: other kernels occupy CUs on other streams while the digit
    passes run, and a second context sorts at the same time.  Every word of the output is checked."""
    import threading
    import os
    from libbsc_amd import GpuContext
    torch = torch_cuda
    stop = threading.Event()

    def hog():                                          # LDS-heavy and long-running kernels on torch's own streams
        s = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device="cuda")
        with torch.cuda.stream(s):
            while not stop.is_set():
                b = a @ a
                b = torch.sort(b.view(-1)[: 1 << 22])[0]
                s.synchronize()

    os.environ["BSC_RS_ONESWEEP"] = "2"
    try:
        other = GpuContext(0, max_n=(8 << 20) + 4096)
    finally:
        os.environ.pop("BSC_RS_ONESWEEP", None)
    errors = []

    def second():
        try:
            rng2 = np.random.default_rng(7)
            k2 = _os_keys(rng2, 6_000_011, "skewed")
            for _ in range(6):
                _os_check(torch, other, k2, True, 0, 64)
        except Exception as e:          # surfaced in the main thread
            errors.append(e)

    th = [threading.Thread(target=hog), threading.Thread(target=second)]
    for t in th:
        t.start()
    try:
        rng = np.random.default_rng(8)
        keys = _os_keys(rng, 9_000_017, "uniform")
        for rep in range(6):
            _os_check(torch, os_ctx, keys, rep % 2 == 0, 0, 64)
    finally:
        stop.set()
        for t in th:
            t.join()
        other.close()
    assert not errors, errors


def test_single_read_pass_give_up_is_retried_not_fatal(torch_cuda):
    """A scout wave that exhausts its bounded poll (pre-emption, a debugger, a hogged CU) must cost a retry, not the block and never a
    store outside the buffers: the sticky error word fails the sort, the transform is redone once through the three-kernel passes
    (bwt_device / st_device) and the context counts it.  The give-up is injected (BSC_RS_FAULT=<k>: the k-th check of the process
    reports one that did not happen; BSC_RS_FAULT_DEV=<k>: the k-th single-read sort finds the device word raised after its first pass, so
    its later passes take the early exit and its output is garbage), in child processes because the knobs are read once; BWT and ST8
    (key + value passes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from libbsc_amd import GpuContext, api
from oracle.refbind import Ref
ref = Ref()
ctx = GpuContext(0, max_n=(9 << 20) + 4096)
T = api.synth_text_v1(5, 9 << 20)
before = ctx.option_get(ctx.CNT_OS_RETRIES)
L, idx, _ = ctx.bwt(T)                                   # the first check of the process is the one behind the first sort: injected
wL, widx, _ = ref.bwt_encode(T, aux=False)
assert idx == widx and np.array_equal(L, wL), "BWT after a retried sort differs"
assert ctx.option_get(ctx.CNT_OS_RETRIES) == before + 1, ctx.option_get(ctx.CNT_OS_RETRIES)
L2, idx2, _ = ctx.bwt(T)                                 # and nothing sticks: the next transform runs the single-read passes again
assert idx2 == widx and np.array_equal(L2, wL) and ctx.option_get(ctx.CNT_OS_RETRIES) == before + 1
print("retried ok")
""" % root
    for knob in ("BSC_RS_FAULT", "BSC_RS_FAULT_DEV"):
        env = dict(os.environ, BSC_RS_ONESWEEP="2")
        env[knob] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root)
        assert r.returncode == 0 and "retried ok" in r.stdout, knob + r.stdout[-2000:] + r.stderr[-2000:]
    code_st = code.replace("L, idx, _ = ctx.bwt(T)", "out, idx = ctx.st_encode(T, 8); L = out").replace(
        "wL, widx, _ = ref.bwt_encode(T, aux=False)", "wL, widx = ctx.st_encode(T, 8)").replace(
        "L2, idx2, _ = ctx.bwt(T)", "L2, idx2 = ctx.st_encode(T, 8)")
    # ST8: the injected run is compared with an uninjected run of the same context (the reference's CPU build has no ST8 encoder)
    env = dict(os.environ, BSC_RS_FAULT_DEV="1", BSC_RS_ONESWEEP="2")
    r = subprocess.run([sys.executable, "-c", code_st], capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0 and "retried ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]



# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 5, 64, 4095, 4096, 4097, 100000, (1 << 20) + 123, 5_000_000])
@pytest.mark.parametrize("mode", ["pairs", "keys"])
def test_radix_sort_matches_stable_sort(ctx, torch_cuda, n, mode):
    torch = torch_cuda
    rng = np.random.default_rng(n + 17)
    # skewed 64-bit keys: few distinct bytes per digit, like text
    kb = rng.integers(0, 7, (max(n, 1), 8), dtype=np.uint8) * 37
    kb[:, 3] = rng.integers(0, 256, max(n, 1), dtype=np.uint8)
    keys = kb.view(np.uint64).reshape(-1)[:n].copy()
    vals = np.arange(n, dtype=np.uint32)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    dk2 = torch.empty_like(dk)
    if mode == "pairs":
        dv = torch.from_numpy(vals.view(np.int32)).cuda()
        dv2 = torch.empty_like(dv)
    else:
        dv = dv2 = None
    for (b0, b1) in [(0, 64), (8, 48), (0, 27), (32, 58)]:
        dk.copy_(torch.from_numpy(keys.view(np.int64)))
        if dv is not None:
            dv.copy_(torch.from_numpy(vals.view(np.int32)))
        rk, rv = ctx.radix_sort(dk, dk2, dv, dv2, n, b0, b1)
        torch.cuda.synchronize()
        mask = np.uint64(((1 << (b1 - b0)) - 1) << b0)
        order = np.argsort(keys & mask, kind="stable")
        got_k = rk.cpu().numpy().view(np.uint64)
        assert np.array_equal(got_k, keys[order]), f"keys mismatch n={n} bits=[{b0},{b1})"
        if dv is not None:
            got_v = rv.cpu().numpy().view(np.uint32)
            assert np.array_equal(got_v, vals[order]), f"vals mismatch n={n} bits=[{b0},{b1})"


def test_adler32_device(ctx, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(3)
    for n in [1, 15, 16, 17, 4096, 65521, 1 << 20, (3 << 20) + 5]:
        for data in (rng.integers(0, 256, n, dtype=np.uint8), np.full(n, 255, np.uint8)):
            d = torch.from_numpy(data).cuda()
            assert ctx.adler32_device(d, n) == (zlib.adler32(data.tobytes()) & 0xffffffff), n


def test_adler32_device_unaligned_input(ctx, torch_cuda):
    """A device pointer at an odd offset (slice of a caller's tensor) is accepted: the kernel's 16-byte loads run over the
    context's aligned copy."""
    import zlib
    torch = torch_cuda
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 300_007, dtype=np.uint8)
    d = torch.from_numpy(a).cuda()
    for off in (1, 3, 8, 13):
        assert ctx.adler32_device(d[off:], a.size - off) == zlib.adler32(a[off:].tobytes())


def test_bwt_matches_reference(ctx, ref):
    rng = np.random.default_rng(11)
    bad = []
    for name, T in _corpus(rng):
        n = T.size
        r = _aux_rate(n)
        want_L, want_idx, want_aux = ref.bwt_encode(T, aux=(n >= 16))
        if n >= 16:
            L, idx, I = ctx.bwt(T, aux_rate=r)
            aux = [x - 1 for x in I[1:]][: (n - 1) // r]
            ok = np.array_equal(L, want_L) and idx == want_idx and aux == want_aux
        else:
            L, idx, _ = ctx.bwt(T)
            ok = np.array_equal(L, want_L) and idx == want_idx
        if not ok:
            bad.append((name, n, idx, want_idx, int((L != want_L).sum())))
    assert not bad, bad


def test_bwt_device_resident_16m(ctx, ref, torch_cuda):
    """16 MiB text block, input and output resident in HBM; in-place (dL aliases dT)."""
    from libbsc_amd.synth import synth_text_v1
    torch = torch_cuda
    n = 16 << 20
    T = synth_text_v1(2, n)
    d = torch.from_numpy(T).cuda()
    r = _aux_rate(n)
    idx, I = ctx.bwt_device(d, d, n, aux_rate=r)
    L = d.cpu().numpy()
    want_L, want_idx, want_aux = ref.bwt_encode(T)
    assert idx == want_idx
    assert [x - 1 for x in I[1:]][: (n - 1) // r] == want_aux
    assert np.array_equal(L, want_L)


def test_bwt_long_groups_are_split_not_handed_over(torch_cuda):
    """Groups longer than one workgroup sorts (1024 records): the reference's cub::DeviceSegmentedSort takes any segment length
    (libcubwt.cu:1691); here the long groups are split by the top bits of the round's key and the segmented sort runs again.  A child
    process with shorter first-sort keys (BSC_BWT_W=11: hundreds of long groups on the 64 MiB text block) and the debug log on: the
    16 MiB text block, the edge corpus and the 64 MiB golden block must come out bit-exact, and the log must show long groups being
    split and sorted.  (A split that would not pay — most records in long groups, as with long repeats — is declined and the round is
    handed over to prefix doubling as before: the deep-LCP golden block and the periodic inputs of the corpus take that way.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSC_BWT_W="11", BSCGPU_DEBUG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_device.py"), os.path.join(root, "tests", "test_gpu_compress.py"),
                        "-q", "-x", "-s", "-k", "bwt_device_resident_16m or full_size_64m_block_golden or bwt_matches_reference"],
                       capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    log = r.stdout + r.stderr
    assert "split by the top key bits -> sorted" in log, log[-3000:]


def test_bwt_long_groups_with_default_keys(torch_cuda):
    """Long groups with the DEFAULT first-sort keys (12 characters of text): a 13-character phrase planted 3000 times, each time followed
    by different text — the suffixes at the phrase and at its first few characters agree on more than the key, so several groups of
    ~3000 records (> 1024, what one workgroup sorts) survive the first sort among otherwise ordinary text; the round's key then starts
    inside the phrase and its second character differs, so the split by the top key bits succeeds.  The same with a 40-character phrase:
    the split's buckets are still too long and the round hands over to prefix doubling.  Child process with the debug log on: both
    outcomes must appear in the log, and both blocks must equal libsais's output."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from libbsc_amd import GpuContext, api
from oracle.refbind import Ref
ref = Ref()
n = 6 << 20
ctx = GpuContext(0, max_n=n + 4096)
rng = np.random.default_rng(4)
for plen in (13, 40):
    T = api.synth_text_v1(31, n).copy()
    phrase = np.frombuffer(b"qzjxvkwpyfgmbqzjxvkwpyfgmbqzjxvkwpyfgmbhh"[:plen], np.uint8)
    for p in rng.choice((n - 64) // 64, 3000, replace=False) * 64:
        T[p:p + plen] = phrase
    L, idx, _ = ctx.bwt(T)
    wL, widx, _ = ref.bwt_encode(T, aux=False)
    assert idx == widx and np.array_equal(L, wL), plen
    print("planted phrase of", plen, "ok", flush=True)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, BSCGPU_DEBUG="1"), cwd=root)
    log = r.stdout + r.stderr
    assert r.returncode == 0 and "planted phrase of 40 ok" in log, log[-3000:]
    assert "split by the top key bits -> sorted" in log, log[-3000:]
    assert "handing over" in log or "not split" in log or "still too long" in log, log[-3000:]


@pytest.mark.parametrize("env", [{}, {"BSC_BWT_HYBRID": "0", "BSC_BWT_ISASKIP": "0"}, {"BSC_BWT_HYBRID_PCT": "30"}],
                         ids=["default", "no-hybrid-all-rank-stores", "hybrid-below-30pct"])
def test_bwt_hybrid_doubling_rounds(torch_cuda, env):
    """Prefix-doubling rounds with groups of more than 1024 records (bwt.hip: hybrid rounds — the large groups through the radix engine,
    the rest through the segmented sort, rank stores skipped where the rank did not change).  Inputs of the two classes that need them:
    indented source-like text over 256 symbols (runs of spaces, duplicated passages) and object-file-like data (zero runs, repeated
    tables, noise) — synthetic and, where the image holds them, its own *.py / *.so files.  Every variant of the switches must give
    libsais's output; the default must actually take hybrid rounds (debug log)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from libbsc_amd import GpuContext, api
from libbsc_amd.synth import image_corpus
from oracle.refbind import Ref
ref = Ref()
n = 6 << 20
ctx = GpuContext(0, max_n=n + 4096)
rng = np.random.default_rng(11)
def source_like(n):
    T = api.synth_text_v1(5, n).copy()
    T[rng.integers(0, n, n // 300)] = rng.integers(128, 256, n // 300).astype(np.uint8)      # > 128 symbols: 8-bit codes, 8-character keys
    for p in rng.integers(0, n - 64, n // 40):                                                # indentation: runs of 4 .. 40 spaces
        T[p:p + int(rng.integers(4, 41))] = 32
    for _ in range(40):                                                                       # duplicated passages (LCPs up to 60 000)
        L = int(rng.integers(2000, 60000)); a, b = (int(x) for x in rng.integers(0, n - L, 2))
        T[b:b + L] = T[a:a + L]
    return T
def object_like(n):
    T = rng.integers(0, 256, n).astype(np.uint8)
    for p in rng.integers(0, n - 5000, n // 2000):                                            # zero runs of 16 .. 4096 bytes
        T[p:p + int(rng.integers(16, 4097))] = 0
    tab = rng.integers(0, 256, 4096).astype(np.uint8)
    for p in rng.integers(0, n - 4096, 300):                                                  # the same table many times
        T[p:p + 4096] = tab
    return T
cases = [("source-like", source_like(n)), ("object-like", object_like(n))]
for kind in ("python-source", "binary"):
    T = image_corpus(kind, n)
    if T is not None: cases.append((kind, T))
for name, T in cases:
    L, idx, _ = ctx.bwt(T)
    wL, widx, _ = ref.bwt_encode(T, aux=False)
    assert idx == widx and np.array_equal(L, wL), name
    print(name, "ok", flush=True)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, BSCGPU_DEBUG="1", **env), cwd=root)
    log = r.stdout + r.stderr
    assert r.returncode == 0 and "source-like ok" in log and "object-like ok" in log, log[-3000:]
    if env.get("BSC_BWT_HYBRID") == "0":
        assert ": hybrid," not in log, log[-3000:]
    elif "BSC_BWT_HYBRID_PCT" not in env:
        assert ": hybrid," in log, log[-3000:]


@pytest.mark.parametrize("k", [3, 4, 5, 6, 7, 8])
def test_st_matches_reference(ctx, ref, k):
    rng = np.random.default_rng(k)
    bad = []
    for name, T in _corpus(rng):
        n = T.size
        if n < 2:
            continue
        out, idx = ctx.st_encode(T, k)
        if k <= 6:
            want, widx = ref.st_encode(T, k)
            if not (np.array_equal(out, want) and idx == widx):
                bad.append((name, n, idx, widx))
        else:   # reference CPU encoder stops at k = 6 (st.cpp:1004-1009); judge by its decoder (st.cpp:1491)
            back, rc = ref.st_decode(out, k, idx)
            if rc != 0 or not np.array_equal(back, T):
                bad.append((name, n, idx, rc))
    assert not bad, bad


def test_error_paths_and_two_contexts(torch_cuda, ref):
    """C-ABI misuse returns libbsc error codes instead of crashing; two contexts on one GPU work side by side."""
    import ctypes as C
    from libbsc_amd import GpuContext, GpuError
    from libbsc_amd.synth import synth_text_v1
    torch = torch_cuda
    a = GpuContext(0, max_n=1 << 20)
    b = GpuContext(0, max_n=1 << 20)
    try:
        T = synth_text_v1(3, 300_000)
        want = ref.compress(T, 1, 1)
        d = torch.from_numpy(T).cuda()
        assert a.compress_device(d, T.size, 1, 1).tobytes() == want
        assert b.compress_device(d, T.size, 1, 1).tobytes() == want
        big = torch.zeros((1 << 20) + 4097, dtype=torch.uint8, device="cuda")
        with pytest.raises(GpuError) as e:
            a.compress_device(big, big.numel(), 1, 1)
        assert e.value.code == -1                                   # n > max_n -> LIBBSC_BAD_PARAMETER
        with pytest.raises(GpuError):
            a.compress_device(d, T.size, 2, 1)                      # bad sorter
        with pytest.raises(GpuError):
            a.compress_device(d, T.size, 1, 7)                      # bad coder
        with pytest.raises(GpuError):
            a.bwt(np.zeros((1 << 20) + 8192, np.uint8))            # host-pointer hook: arena too small -> -9
        pipe = a.pipe(2)
        assert a.L.bscgpu_pipe_wait(pipe.h, 0) == -1                # nothing submitted yet
        t = pipe.submit(d, T.size, 1, 1)
        assert pipe.wait(t).tobytes() == want
        pipe.close()
        h = C.c_void_p()
        assert a.L.bscgpu_create(C.byref(h), 99, 1 << 20) == -1     # no such device
    finally:
        a.close(); b.close()


def test_contiguous_range_scatter_variants(torch_cuda):
    """The digit-pass kernels that give a workgroup one contiguous tile range (BSC_RS_ORDER=0: the write-combining kernel, forced
    on for every size with >= 4 chunks by BSC_RS_WC=2, and the plain 1024 x 8 kernel, BSC_RS_WC=0) must produce the same stable
    order as the default XCD-interleaved kernel: the radix parity cases and a 16 MiB BWT in child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for wc in ("2", "0"):
        env = dict(os.environ, BSC_RS_WC=wc, BSC_RS_ORDER="0")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_device.py"), "-q", "-x",
                            "-k", "radix_sort_matches or bwt_device_resident_16m"], capture_output=True, text=True, env=env, cwd=root)
        assert r.returncode == 0, (wc, r.stdout[-2000:] + r.stderr[-2000:])


def test_device_static_model_matches_oracle_trace(ctx):
    """bscgpu_qlfc_static_pstream (devcoder.hip): every probability of the static QLFC model computed on the GPU equals the
    oracle's trace of the reference model (oracle/bsc_oracle.c: encode_model1 with the trace hook), sub-block by sub-block,
    including the run-start marks; a block the device path cannot hold (more than 4 decisions per byte) is declined, never approximated."""
    from libbsc_amd import api
    from libbsc_amd.gpu import GpuError
    from oracle.refbind import Oracle, Ref
    orc, ref = Oracle(), Ref()
    rng = np.random.default_rng(11)
    bwt = lambda x: ref.bwt_encode(x)[0]
    cases = [("text300k", bwt(api.synth_text_v1(3, 300_000))), ("text1m", bwt(api.synth_text_v1(1, 1 << 20))),
             ("low1m", bwt(rng.integers(0, 3, 1 << 20, dtype=np.uint8))), ("zeros", np.zeros(500_000, np.uint8)),
             ("sym40", bwt((rng.geometric(0.15, 700_000) % 40).astype(np.uint8))),
             ("longruns", np.repeat(rng.integers(0, 6, 3000, dtype=np.uint8), rng.integers(1, 3000, 3000)).astype(np.uint8)),
             ("text3m", bwt(api.synth_text_v1(4, 3 << 20))),
             # 224 symbols, text-like structure (each 64 KiB segment in its own 32-symbol band): several hundred decision types
             ("text224", bwt(((api.synth_text_v1(6, 3 << 20) & 31) + ((np.arange(3 << 20) >> 16) % 7 * 32).astype(np.uint8)).astype(np.uint8))),
             ("rand600k", rng.integers(0, 256, 600_000, dtype=np.uint8)),           # all 8-bit ranks, escape coding (avg_rank >= 32), ~340 decision types
             ("skew1m", bwt((rng.geometric(0.02, 1 << 20) % 256).astype(np.uint8)))]
    for name, L in cases:
        ps, st, sz, poff, _ = ctx.qlfc_static_pstream(L)
        assert poff[0] == 0 and poff[-1] == len(ps), name
        for b in range(len(st)):
            tr, _ = orc.static_pstream(L[st[b]:st[b] + sz[b]])
            assert np.array_equal(tr, ps[poff[b]:poff[b + 1]]), (name, b)
    # capacity is the only thing the device path declines on such inputs: 4 decisions per byte of the CONTEXT's block size
    # (random bytes need ~9 per byte); a context sized for the block itself says LIBBSC_NOT_SUPPORTED, never approximates
    from libbsc_amd import GpuContext
    small = GpuContext(0, max_n=(1 << 20) + 4096)
    try:
        with pytest.raises(GpuError) as e:
            small.qlfc_static_pstream(rng.integers(0, 256, 1 << 20, dtype=np.uint8))
        assert e.value.code == -4
    finally:
        small.close()


def test_packed_probability_stream_equals_the_16_bit_entries(ctx):
    """Round 6 (BSCGPU_OPT_DC_PACKED_STREAM): the static coder's stream crosses PCIe as 13 bits per decision, eight decisions in 13 bytes,
    written by the wavefronts of dc_pstream (whole groups inside a wavefront's piece) and dc_p13_join_kernel (the group between two
    wavefronts).  Unpacked, it must be the 16-bit stream without its run-start marks — on one to eight sub-blocks, sub-block boundaries
    inside a wavefront (a block of a handful of runs), padding between sub-blocks zero, and it must be refused (not mis-written) where a
    wavefront's 64 runs have more decisions than its staging buffer holds."""
    from libbsc_amd import api
    from libbsc_amd.gpu import GpuError
    from oracle.refbind import Ref
    ref = Ref()
    rng = np.random.default_rng(41)
    bwt = lambda x: ref.bwt_encode(x)[0]
    cases = [("text300k", bwt(api.synth_text_v1(3, 300_000))), ("text1m", bwt(api.synth_text_v1(1, 1 << 20))), ("text5m", bwt(api.synth_text_v1(14, 5 << 20))),
             ("low1m", bwt(rng.integers(0, 3, 1 << 20, dtype=np.uint8))), ("tiny", np.array([1, 1, 2, 2, 2, 1, 3], np.uint8)),
             ("ab", (np.arange(600_000) % 2).astype(np.uint8)), ("sym40", bwt((rng.geometric(0.15, 700_000) % 40).astype(np.uint8))),
             ("rand600k", rng.integers(0, 256, 600_000, dtype=np.uint8)), ("skew3m", bwt((rng.geometric(0.02, 3 << 20) % 256).astype(np.uint8))),
             ("runs of tens", np.repeat(rng.integers(0, 6, 60_000, dtype=np.uint8), rng.integers(1, 40, 60_000)).astype(np.uint8))]
    assert ctx.option_get(ctx.OPT_DC_PACKED_STREAM) == 1
    for name, L in cases:
        ps, st, sz, poff, _ = ctx.qlfc_static_pstream(L)
        f, st2, sz2, poff2, pbase, raw = ctx.qlfc_static_pstream_packed(L)
        assert st == st2 and sz == sz2 and poff == poff2, name
        assert np.array_equal(f, ps & 0x1fff), (name, int(np.argmax(f != (ps & 0x1fff))))
        for b in range(len(st)):                                   # layout: 64-decision alignment of every sub-block, zero padding behind its last field
            assert pbase[b] % 64 == 0 and pbase[b + 1] - pbase[b] == (poff[b + 1] - poff[b] + 63) // 64 * 64, (name, b)
            cnt = poff[b + 1] - poff[b]
            end_bit = pbase[b] // 8 * 13 * 8 + cnt * 13
            tail = raw[(end_bit + 7) // 8: (pbase[b] // 8 + (cnt + 7) // 8) * 13]
            assert not tail.any(), (name, b)
            if end_bit % 8:
                assert raw[end_bit // 8] >> (end_bit % 8) == 0, (name, b)
    # runs of thousands: a wavefront's 64 runs hold more decisions than its staging buffer -> the packed form is refused for the block,
    # and bsc_compress (which then moves 16-bit entries) still gives the reference's bytes
    long_runs = np.repeat(rng.integers(0, 6, 3000, dtype=np.uint8), rng.integers(2000, 9000, 3000)).astype(np.uint8)
    with pytest.raises(GpuError) as e:
        ctx.qlfc_static_pstream_packed(long_runs)
    assert e.value.code == -4
    ps, *_ = ctx.qlfc_static_pstream(long_runs)
    assert len(ps) > 0
    # option off: refused as well, the 16-bit stage unaffected
    ctx.option_set(ctx.OPT_DC_PACKED_STREAM, 0)
    try:
        with pytest.raises(GpuError):
            ctx.qlfc_static_pstream_packed(cases[0][1])
    finally:
        ctx.option_set(ctx.OPT_DC_PACKED_STREAM, 1)


def test_stream_order_static_family_equals_partitioned_path(ctx):
    """devcoder_static.h (round 6): for blocks of at most 32 symbols per sub-block the static coder's context-free counter family is
    walked in stream order instead of being partitioned.  Same probability stream as the general path (BSCGPU_OPT_DC_STREAM_STATIC = 0)
    and as the oracle's trace of the reference model, on 1 / 2 / 4 sub-blocks, sub-blocks of different alphabet size (max_rank 0..4 and
    one with 40 symbols, which sends the whole block down the general path), a block of a handful of runs, and constant data (brackets
    that cannot close: declined or exact, never approximate)."""
    from libbsc_amd import api
    from libbsc_amd.gpu import GpuError
    from oracle.refbind import Oracle, Ref
    orc, ref = Oracle(), Ref()
    rng = np.random.default_rng(29)
    bwt = lambda x: ref.bwt_encode(x)[0]
    def alpha(n, k, p=0.3):         # k symbols, geometric
        return (rng.geometric(p, n) % k).astype(np.uint8)
    parts = [alpha(300_000, 2), alpha(700_000, 4), alpha(900_000, 7, 0.1), alpha(400_000, 32, 0.05), alpha(800_000, 17, 0.08), alpha(1_100_000, 3)]
    cases = [("text200k", bwt(api.synth_text_v1(12, 200_000))), ("text1m", bwt(api.synth_text_v1(13, 1 << 20))), ("text5m", bwt(api.synth_text_v1(14, 5 << 20))),
             ("mixed alphabets 4.2m", np.concatenate(parts)), ("mixed + 40 symbols", np.concatenate(parts[:3] + [alpha(500_000, 40, 0.03)] + parts[3:])),
             ("tiny", np.array([1, 1, 2, 2, 2, 1, 3], np.uint8)), ("one run", np.zeros(70_000, np.uint8)), ("ab", (np.arange(600_000) % 2).astype(np.uint8)),
             ("dna5m", bwt(rng.integers(0, 4, 5 << 20, dtype=np.uint8))), ("32 symbols uniform", rng.integers(0, 32, 2 << 20, dtype=np.uint8))]
    assert ctx.option_get(ctx.OPT_DC_STREAM_STATIC) == 0          # off by default (measured slower: profiles/r06)
    try:
        for name, L in cases:
            got = {}
            for mode in (1, 0):
                ctx.option_set(ctx.OPT_DC_STREAM_STATIC, mode)
                try:
                    got[mode] = ctx.qlfc_static_pstream(L, debug=(L.size <= (1 << 20)))
                except GpuError as e:
                    assert e.code == -4, (name, mode, e)            # declined: the block takes the host model
                    got[mode] = None
            if got[1] is None or got[0] is None:
                continue
            (ps1, st1, sz1, poff1, dbg1), (ps0, st0, sz0, poff0, dbg0) = got[1], got[0]
            assert st1 == st0 and sz1 == sz0 and poff1 == poff0, name
            assert np.array_equal(ps1, ps0), (name, int((ps1 != ps0).sum()))
            if dbg1 is not None:
                assert np.array_equal(dbg1, dbg0), name                 # the three counter values behind every probability
            for b in range(len(st1)):
                tr, _ = orc.static_pstream(L[st1[b]:st1[b] + sz1[b]])
                assert np.array_equal(tr, ps1[poff1[b]:poff1[b + 1]]), (name, b)
    finally:
        ctx.option_set(ctx.OPT_DC_STREAM_STATIC, 0)


def test_gpu_inverse_bwt(ctx, ref):
    """bscgpu_unbwt (unbwt.hip): LF mapping from one radix pass with destination positions, the LF cycle cut at ~n/128 marked rows
    and walked in parallel.  Against the texts the reference's forward BWT came from; wrong primary indexes and damaged columns
    must come back as DATA_CORRUPT or as a different text, never hang."""
    from libbsc_amd.synth import synth_text_v1, synth_repeat_v1
    rng = np.random.default_rng(3)
    cases = [synth_text_v1(7, n) for n in (1, 2, 3, 17, 127, 128, 129, 1000, 65536, 300_000, (1 << 20) + 17, 5 << 20)]
    cases += [rng.integers(0, 256, 200_000, dtype=np.uint8), np.zeros(70_000, np.uint8), (np.arange(100_000) % 2).astype(np.uint8),
              synth_repeat_v1(3, 2 << 20, 50_000), np.full(5, 255, np.uint8)]
    for T in cases:
        L, idx, _ = ref.bwt_encode(T, aux=False)
        back, rc = ctx.unbwt(L, idx)
        assert rc == 0 and np.array_equal(back, T), (T.size, rc)
    T = synth_text_v1(9, 400_000)
    L, idx, _ = ref.bwt_encode(T, aux=False)
    for bad in (1, idx - 1, idx + 1, T.size):
        if bad == idx or bad < 1 or bad > T.size:
            continue
        back, rc = ctx.unbwt(L, bad)
        assert rc == -6 or (rc == 0 and not np.array_equal(back, T)), bad
    L2 = L.copy(); L2[1000:1100] = L2[5000:5100]                # damaged column: symbol counts change, cycles break
    back, rc = ctx.unbwt(L2, idx)
    assert rc == -6 or (rc == 0 and not np.array_equal(back, T))
