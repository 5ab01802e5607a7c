"""CPU tests: the product's host-side coder / container against the compiled reference (oracle/_ref).
No GPU needed: inputs to the coder are produced by the reference's own BWT."""
import hashlib

import numpy as np
import pytest

from libbsc_amd import api


def _texts():
    from libbsc_amd.synth import synth_text_v1
    rng = np.random.default_rng(5)
    out = [("text300k", synth_text_v1(3, 300_000)), ("text1m", synth_text_v1(1, 1 << 20))]
    out.append(("rand16-200k", rng.integers(0, 16, 200_000, dtype=np.uint8)))
    out.append(("rand256-100k", rng.integers(0, 256, 100_000, dtype=np.uint8)))
    out.append(("zeros-70k", np.zeros(70_000, np.uint8)))
    out.append(("one-run-then-text", np.concatenate([np.full(100_000, 65, np.uint8), synth_text_v1(9, 50_000)])))
    long_runs = np.repeat(rng.integers(0, 200, 3000, dtype=np.uint8), rng.integers(1, 3000, 3000))
    out.append(("long-runs", long_runs))
    allsym = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 50_000, dtype=np.uint8) // 3])
    out.append(("all-256-symbols", allsym))
    for n in (1, 2, 3, 29, 100, 1000):
        out.append((f"tiny{n}", rng.integers(97, 100, n, dtype=np.uint8)))
    return out


def test_synth_generators_agree_and_match_survey_md5():
    from libbsc_amd.synth import synth_text_v1
    a = synth_text_v1(1, 1 << 20)
    b = api.synth_text_v1(1, 1 << 20)
    assert np.array_equal(a, b)
    assert hashlib.md5(a.tobytes()).hexdigest() == "7e493144d260fc285567cb4145cf4737"
    assert np.array_equal(synth_text_v1(7, 12345), api.synth_text_v1(7, 12345))


def test_adler32_matches_reference(ref):
    rng = np.random.default_rng(0)
    for n in (0, 1, 24, 5551, 5552, 5553, 100_000, 1 << 20):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert api.bsc_adler32(d) == ref.adler32(d)
    d = np.full(1 << 20, 255, np.uint8)
    assert api.bsc_adler32(d) == ref.adler32(d)


def test_qlfc_ranks_match_reference_transform(ref):
    for name, T in _texts():
        L, _, _ = ref.bwt_encode(T, aux=False)
        want_ranks, want_mtf = ref.qlfc_transform(L)
        ranks, first = api.bsc_qlfc_ranks(L)
        assert np.array_equal(ranks, want_ranks), name
        k = first.size
        assert np.array_equal(first, want_mtf[:k]), name          # MTFTable[0..K) = first-appearance order
        if k < 256:
            assert want_mtf[k] == want_mtf[k - 1], name          # terminator (qlfc.cpp:252)


@pytest.mark.parametrize("coder", [1, 2, 3])
def test_qlfc_encode_block_matches_reference(ref, coder):
    for name, T in _texts():
        L, _, _ = ref.bwt_encode(T, aux=False)
        want = ref.qlfc_encode_block(L, coder)
        got = api.bsc_qlfc_encode_block(L, coder)
        assert got == want, (name, coder, got if isinstance(got, int) else len(got), want if isinstance(want, int) else len(want))


@pytest.mark.parametrize("coder", [1, 2, 3])
@pytest.mark.parametrize("features", [1, 3])
def test_coder_compress_matches_reference(ref, coder, features):
    from libbsc_amd.synth import synth_text_v1
    cases = _texts() + [("text5m", synth_text_v1(4, 5 << 20))]        # 4 sub-blocks
    for name, T in cases:
        L, _, _ = ref.bwt_encode(T, aux=False)
        want = ref.coder_compress(L, coder, features=features)
        got = api.bsc_coder_compress(L, coder, features=features)
        assert got == want, (name, coder, features)


def test_store_and_block_info_match_reference(ref):
    rng = np.random.default_rng(1)
    for n in (0, 1, 28, 29, 1000):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        ours = api.bsc_store(d)
        out = np.empty(n + 28, np.uint8)
        import ctypes as C
        from oracle.refbind import u8p
        r = ref.L.ref_bsc_store(d.ctypes.data_as(u8p), out.ctypes.data_as(u8p), n, 3)
        assert ours == out[:r].tobytes()
        assert api.bsc_block_info(ours) == (0, n + 28, n)
        assert api.bsc_decompress(ours) == d.tobytes()
    bad = bytearray(api.bsc_store(b"hello world, hello world, hello"))
    bad[5] ^= 1
    assert api.bsc_block_info(bytes(bad))[0] == api.DATA_CORRUPT
    assert api.bsc_block_info(b"short")[0] == api.UNEXPECTED_EOB


def test_parameter_validation_matches_reference():
    d = np.zeros(1000, np.uint8)
    assert api.bsc_compress(d, sorter=2) == api.BAD_PARAMETER
    assert api.bsc_compress(d, sorter=9) == api.BAD_PARAMETER
    assert api.bsc_compress(d, coder=0) == api.BAD_PARAMETER
    assert api.bsc_compress(d, coder=4) == api.BAD_PARAMETER
    assert api.bsc_compress(d, lzp_hash=9, lzp_min=128) == api.BAD_PARAMETER
    assert api.bsc_compress(d, lzp_hash=15, lzp_min=3) == api.BAD_PARAMETER
    r = api.bsc_compress(d, lzp_hash=15, lzp_min=128)          # valid: a block on a GPU box, a loud GPU error without one
    assert isinstance(r, bytes) or r in (-7, -8, -9), r         # LIBBSC_GPU_ERROR / _NOT_SUPPORTED / _NOT_ENOUGH_MEMORY
    small = np.arange(20, dtype=np.uint8)
    assert api.bsc_compress(small) == api.bsc_store(small)                         # n <= 28 -> stored (libbsc.cpp:259)


def test_native_library_exports_every_declared_symbol():
    import re, os
    from libbsc_amd import _native as N
    L = N.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = []
    for h in ("include/libbsc.h", "include/bscgpu.h"):
        txt = open(os.path.join(root, h)).read()
        names += re.findall(r"(?:LIBBSC_API|BSCGPU_API)[^;(]*?\b(bsc\w*)\s*\(", txt)
    assert len(names) > 30
    for nm in names:
        assert hasattr(L, nm), nm


def test_avx2_eight_lane_range_coder_matches_scalar(tmp_path):
    """qlfc_encode_static_pstream_x8 / qlfc_encode_fast_pstream_x8 (eight sub-block streams in SIMD lanes, renormalisation log) against
    the scalar coders fed by the same probability streams: byte-identical outputs, and it must give up (never mis-code) when a stream
    reaches its output budget.  Every step the product can run is covered — round 5's AVX-512VL step, round 4's (BSC_RC_VSEL=0), the AVX2
    step (BSC_RC_AVX512=0) where the CPU has them — at stream lengths below one step, around it and across the replay chunk, built with
    g++ and with the product's own host compiler (the select in RangeEncoder::next_range is compiler-specific).
    tools/rc_x8_check.cpp compiles the host coder directly (no test hook in the product library)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    compilers = [("g++", ["-O2"])]
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if os.path.exists(clang):
        compilers.append((clang, ["-O3", "-mtune=znver4"]))
    for k, (cxx, flags) in enumerate(compilers):
        exe = str(tmp_path / f"rc_x8_check_{k}")
        subprocess.run([cxx, *flags, "-std=c++17", "-march=x86-64-v3", "-I", os.path.join(root, "libbsc_amd/csrc/host"), "-I", os.path.join(root, "include"),
                        os.path.join(root, "tools/rc_x8_check.cpp"), "-o", exe], check=True)
        # (BSC_RC_VBMI: the packed stream's unpack + transpose by one byte permute across two 512-bit registers; on by default on AMD hosts only)
        for extra in ({}, {"BSC_RC_VSEL": "0"}, {"BSC_RC_AVX512": "0"}, {"BSC_RC_PREFETCH": "0"}, {"BSC_RC_VBMI": "1"}, {"BSC_RC_VBMI": "1", "BSC_RC_VSEL": "0"}, {"BSC_RC_VBMI": "0"},
                      {"BSC_RC_VBMI16": "1"}, {"BSC_RC_VBMI16": "1", "BSC_RC_VSEL": "0"}):
            for size in ("1", "9", "400000"):
                r = subprocess.run([exe, size], capture_output=True, text=True, env={**os.environ, **extra})
                assert r.returncode == 0 and "all equal" in r.stdout and "fast coder, eight lanes: equal" in r.stdout, (cxx, extra, size, r.stdout + r.stderr)
                assert "packed stream: equal" in r.stdout, (cxx, extra, size, r.stdout + r.stderr)       # round 6: 13 bits per decision (single, pair, eight lanes)
                if size == "400000":
                    assert "gave up" in r.stdout            # the budget case bails out to the scalar coders
        # round 6: two blocks in the sixteen lanes of 512-bit registers (opt-in: BSC_RC_X16=1), static and fast entries, roomy and budget cases
        for size in ("9", "400000"):
            r = subprocess.run([exe, size], capture_output=True, text=True, env={**os.environ, "BSC_RC_X16": "1"})
            assert r.returncode == 0 and "all equal" in r.stdout, (cxx, size, r.stdout + r.stderr)
            assert "sixteen lanes: equal" in r.stdout or "sixteen lanes: not available" in r.stdout, r.stdout


@pytest.mark.parametrize("nphys", [1, 2, 4, 8])
@pytest.mark.parametrize("cpd", [1, 2])
def test_default_dispatch_spreads_callers_over_physical_devices_first(nphys, cpd):
    """The rule behind concurrent bsc_compress calls (block.cpp default_gpu_acquire), as the pure functions it is built from:
    N concurrent callers on an N-GPU node land on N different GPUs; the next N double up, one more per GPU; a released slot is
    preferred again; unusable slots are skipped; nothing usable -> -1."""
    import ctypes as C
    from libbsc_amd import _native
    L = _native.lib()
    pick = L.bscgpu_dispatch_pick
    pick.restype = C.c_int
    pick.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_ubyte), C.c_uint]
    dev = L.bscgpu_dispatch_device
    dev.restype = C.c_int
    dev.argtypes = [C.c_int, C.c_int]
    nslots = nphys * cpd
    users = (C.c_int * nslots)(*([0] * nslots))
    usable = (C.c_ubyte * nslots)(*([1] * nslots))
    rr = 0
    per_dev = [0] * nphys
    for caller in range(nslots):
        s = pick(nphys, cpd, users, usable, rr)
        assert 0 <= s < nslots and users[s] == 0, (caller, s)
        users[s] += 1
        rr = s + 1
        per_dev[dev(s, nphys)] += 1
        # after k callers no GPU has more than ceil(k / nphys) of them
        assert max(per_dev) == -(-(caller + 1) // nphys), (caller, per_dev)
    assert per_dev == [cpd] * nphys
    # one more caller queues behind the least-loaded GPU; with a slot released that GPU is the one chosen
    users[nslots - 1] -= 1
    s = pick(nphys, cpd, users, usable, rr)
    assert s == nslots - 1
    # among equally loaded slots one whose context already exists and fits (usable = 2) is preferred: a lone caller stays put
    for i in range(nslots):
        users[i] = 0
    usable[nslots - 1] = 2
    for start in range(nslots):
        assert pick(nphys, cpd, users, usable, start) == nslots - 1
    users[nslots - 1] = 1                                    # ... but never over an idle GPU / an idle slot
    if nslots > 1:
        assert pick(nphys, cpd, users, usable, 0) != nslots - 1
    users[nslots - 1] = 0
    # unusable slots are never chosen; all unusable -> -1
    for i in range(nslots):
        usable[i] = 0
    assert pick(nphys, cpd, users, usable, 0) == -1
    usable[0] = 1
    assert pick(nphys, cpd, users, usable, 3) == 0
    assert [dev(s, nphys) for s in range(nslots)] == [s % nphys for s in range(nslots)]


def test_coder_task_shape_rule():
    """How the host codes the eight sub-blocks of a device-model block (block.cpp: ps_group), as the pure function it is built on:
    one eight-lane SIMD task when the pool is busy, four pair tasks while >= 4 CPUs of its budget are idle, short tasks for blocks
    marked low-latency and for synchronous calls (sized by the caller's share of the CPUs), BSC_RC_SIMD overrides everything."""
    import ctypes as C
    from libbsc_amd import _native
    f = _native.lib().bscgpu_coder_task_shape
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 6
    shape = lambda forced=-1, low=0, free=-1, sync=16, wide=1, adaptive=1: f(forced, low, free, sync, wide, adaptive)
    # a pipe's block on a CPU with AVX-512VL: by the pool's idle CPUs
    assert [shape(free=k) for k in (0, 3, 4, 16)] == [8, 8, 2, 2]
    assert shape(free=16, adaptive=0) == 8                      # BSC_RC_ADAPTIVE=0
    assert shape(free=0, wide=0) == 2 and shape(free=16, wide=0) == 2      # no AVX-512VL: pairs (bench.py forces the AVX2 lanes where the CPU share is small)
    # marked low-latency in a pipe: never the eight-lane task; eight scalar tasks only when eight CPUs of the pool are idle
    assert [shape(low=1, free=k) for k in (0, 4, 7, 8, 24)] == [2, 2, 2, 1, 1]
    # synchronous calls: by the share of the CPUs per caller
    assert [shape(low=1, sync=c) for c in (2, 4, 7, 8, 16)] == [2, 2, 2, 1, 1]
    # forced
    assert shape(forced=8, low=1, free=24) == 8 and shape(forced=0, free=0) == 2


def test_fast_coder_chain_model_matches_the_host_coder(tmp_path):
    """The fast coder (-e0) as the device runs it — chains (decision type, symbol) with the shift updates of dcm::model_params_fast,
    entries of 13 / 11 bits, qlfc_encode_fast_pstream and its pair version — walked serially on the CPU: the bytes must be those of
    the host's own fast coder, which the tests above pin to the reference (tools/devcoder_fast_sim.cpp; random bytes, long runs,
    text-like and one-symbol data, with room and at the format's budget out_size = in_size)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fast_sim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-march=x86-64-v3", "-I", os.path.join(root, "libbsc_amd/csrc/host"), "-I", os.path.join(root, "libbsc_amd/csrc/device"),
                    "-I", os.path.join(root, "include"), os.path.join(root, "tools/devcoder_fast_sim.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout + r.stderr


def test_stream_order_static_family_lane_functions(tmp_path):
    """The static coder's context-free counter family evaluated in stream order (devcoder_static.h, round 6): descriptors derived from
    the model's own case analysis for max_rank 0..4 (checked rank by rank), rank bit planes, bracket walk per (slot, chunk), resolve,
    exact walk, values per (slot, sub-tile) — every lane function the HIP kernels call, run lane by lane on the CPU against a plain
    sequential walk of the chains (tools/devcoder_static_sim.cpp): 151 rank sequences incl. one-run blocks, constant ranks (brackets
    that never close: the path must decline, not approximate), sub-block starts crowded into one tile, on a tile's first lane and on
    a chunk's first lane, and sub-blocks with different max_rank; then the ranks of a real BWT block through the host front end."""
    import os
    import subprocess
    import numpy as np
    from libbsc_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "static_sim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-march=x86-64-v3", "-I", os.path.join(root, "libbsc_amd/csrc/host"), "-I", os.path.join(root, "include"),
                    os.path.join(root, "tools/devcoder_static_sim.cpp"), os.path.join(root, "libbsc_amd/csrc/host/coder.cpp"), "-o", exe, "-lpthread"], check=True)
    # a BWT-like block: text sorted by its 3 following characters (any byte sequence serves as the coder's input; this one has the
    # structure of a sorted block: long stretches of few symbols), 5 MiB = 4 sub-blocks
    T = api.synth_text_v1(9, 5 << 20)
    key = (T[np.r_[1:T.size, 0]].astype(np.uint32) << 16) | (T[np.r_[2:T.size, 0, 1]].astype(np.uint32) << 8) | T[np.r_[3:T.size, 0, 1, 2]]
    L = T[np.argsort(key, kind="stable")]
    f = tmp_path / "block.bin"
    L.tofile(f)
    r = subprocess.run([exe, str(f)], capture_output=True, text=True)
    assert r.returncode == 0 and "static stream-order evaluation OK" in r.stdout and "mismatches 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
