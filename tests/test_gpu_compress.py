"""GPU parity tests for the whole block path: bsc_compress (host API, GPU sorters) and
bscgpu_compress_device (input resident in HBM) must reproduce the reference's bsc_compress byte for byte."""
import hashlib

import numpy as np
import pytest

from libbsc_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch


def _inputs():
    from libbsc_amd.synth import synth_text_v1
    rng = np.random.default_rng(21)
    return [
        ("text1m", synth_text_v1(1, 1 << 20)),
        ("text300k", synth_text_v1(3, 300_000)),
        ("text65535", synth_text_v1(6, 65535)),          # below the aux-index threshold (libbsc.cpp:176)
        ("text65536", synth_text_v1(6, 65536)),
        ("n29", synth_text_v1(8, 29)),
        ("rand256", rng.integers(0, 256, 200_000, dtype=np.uint8)),     # incompressible -> bsc_store path
        ("zeros", np.zeros(100_000, np.uint8)),
        ("mixed", np.concatenate([synth_text_v1(2, 400_000), np.zeros(50_000, np.uint8), rng.integers(0, 4, 100_000, dtype=np.uint8)])),
        # the second sub-block of the sorted block is pure noise and is stored raw inside a block that still compresses
        # (raw sub-blocks are rebuilt on the host from the run arrays: the sorted block itself never crosses PCIe)
        ("zeros+noise", np.concatenate([np.zeros(600_000, np.uint8), rng.integers(0, 256, 400_000, dtype=np.uint8)])),
    ]


@pytest.mark.parametrize("sorter", [1, 5, 6])
@pytest.mark.parametrize("coder", [1, 2, 3])
def test_bsc_compress_matches_reference(ref, sorter, coder):
    for name, T in _inputs():
        want = ref.compress(T, sorter, coder)
        got = api.bsc_compress(T, sorter, coder)
        assert got == want, (name, sorter, coder)
        # in-place variant (libbsc.cpp:83): identical except that an incompressible block reports -3
        got_ip = api.bsc_compress(T, sorter, coder, inplace=True)
        if isinstance(want, bytes) and want[8:12] != b"\0\0\0\0":
            assert got_ip == want, (name, "inplace")
        else:
            assert got_ip == api.NOT_COMPRESSIBLE or got_ip == want, (name, "inplace-store")


@pytest.mark.parametrize("K", [2, 17, 32, 33, 48, 64, 65, 200])
def test_front_end_rank_paths_by_alphabet_size(ref, K):
    """qf_rank's three set layouts — 32-bit sets (<= 32 symbols), one 64-bit word (<= 64), four words — on skewed alphabets: frequent
    symbols come back inside the lifted tile, middling ones inside the 256-run halo behind it (round 5), rare ones only after thousands
    of runs (tile / super-tile sets, then the serial walk) or never again before the sub-block ends.  Four sub-blocks, so tiles straddle
    sub-block ends too.  Whole blocks against the reference, static and fast coder (device model) and the adaptive one (host model)."""
    rng = np.random.default_rng(1000 + K)
    p = 1.0 / np.arange(1, K + 1) ** 1.6
    p /= p.sum()
    n = 5 << 20
    T = rng.choice(K, size=n, p=p).astype(np.uint8)
    # runs, and order: sort inside short windows so that the BWT input has structure (the front end sees the SORTED block anyway)
    T = np.repeat(T[: n // 3 + 1], 3)[:n].copy()
    T[::97] = (K - 1)                                    # the rarest symbol at a fixed stride as well
    syms = rng.permutation(256)[:K].astype(np.uint8)     # not the low byte values only
    T = syms[T]
    for coder in (1, 3, 2):
        want = ref.compress(T, 1, coder)
        got = api.bsc_compress(T, 1, coder)
        assert got == want, (K, coder)


def test_reference_decoder_accepts_our_blocks(ref):
    for name, T in _inputs():
        for sorter, coder in ((1, 1), (1, 2), (5, 3), (6, 1), (3, 1), (4, 2), (7, 1), (8, 1)):
            blk = api.bsc_compress(T, sorter, coder)
            assert isinstance(blk, bytes), (name, sorter, coder, blk)
            assert ref.decompress(blk) == T.tobytes(), (name, sorter, coder)


def test_compress_device_resident(ref, torch_cuda):
    from libbsc_amd import GpuContext
    torch = torch_cuda
    ctx = GpuContext(0, max_n=(4 << 20) + 4096)
    try:
        for name, T in _inputs():
            d = torch.from_numpy(T).cuda()
            for sorter, coder in ((1, 1), (1, 2), (5, 1), (6, 3)):
                got = ctx.compress_device(d, T.size, sorter, coder).tobytes()
                assert got == ref.compress(T, sorter, coder), (name, sorter, coder)
    finally:
        ctx.close()


@pytest.mark.slow
def test_full_size_64m_block_golden(ref, torch_cuda):
    """BASELINE config 3: one 64 MiB block, BWT + QLFC static; md5 pinned in SURVEY.md §8c and re-derived here."""
    from libbsc_amd import GpuContext
    torch = torch_cuda
    n = 64 << 20
    T = api.synth_text_v1(2, n)
    assert hashlib.md5(T.tobytes()).hexdigest() == "968e72345b8eecbdf5e3574aba3e4127"
    ctx = GpuContext(0, max_n=n + 4096)
    try:
        d = torch.from_numpy(T).cuda()
        blk = ctx.compress_device(d, n, 1, 1).tobytes()
    finally:
        ctx.close()
    assert len(blk) == 15277890
    assert hashlib.md5(blk).hexdigest() == "0ae79c8172e7e5df11b4b5a2e5c53b58"
    want = ref.compress(T, 1, 1)
    assert blk == want
    # size-independent property: the reference decoder round-trips our block
    assert ref.decompress(blk) == T.tobytes()


def test_golden_fixtures_on_gpu(torch_cuda):
    """tests/golden/golden.json (reference outputs, generated by tests/golden/make_golden.py) — the parity pin
    that does not need /root/reference on the GPU box.  Includes the full 64 MiB BASELINE block."""
    import json, os
    from libbsc_amd import GpuContext
    from libbsc_amd.synth import synth_text_v1
    torch = torch_cuda
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
    ctx = GpuContext(0, max_n=(64 << 20) + 4096)
    try:
        for e in g["blocks"]:
            T = api.synth_text_v1(e["seed"], e["n"]) if e["kind"] == "synth" else np.frombuffer(bytes.fromhex(e["hex"]), np.uint8).copy()
            assert hashlib.md5(T.tobytes()).hexdigest() == e["input_md5"]
            d = torch.from_numpy(T).cuda()
            blk = ctx.compress_device(d, T.size, e["sorter"], e["coder"], 1).tobytes()
            assert (len(blk), hashlib.md5(blk).hexdigest()) == (e["size"], e["md5"]), {k: e[k] for k in ("kind", "n", "sorter", "coder")}
            if "bwt_index" in e:
                out = torch.empty_like(d)
                r = 1 << ((T.size // 8).bit_length() - 1)
                idx, I = ctx.bwt_device(d, out, T.size, aux_rate=r)
                assert idx == e["bwt_index"] and [x - 1 for x in I[1:]][: (T.size - 1) // r] == e["bwt_aux"]
                assert hashlib.md5(out.cpu().numpy().tobytes()).hexdigest() == e["bwt_md5"]
    finally:
        ctx.close()


def test_pipe_matches_sync_path(ref, torch_cuda):
    """bscgpu_pipe_*: several blocks in flight must give the same bytes as the synchronous call, in ticket order."""
    from libbsc_amd import GpuContext
    torch = torch_cuda
    ctx = GpuContext(0, max_n=(2 << 20) + 4096)
    try:
        blocks = [api.synth_text_v1(30 + i, (1 << 20) + 1000 * i) for i in range(5)]
        dev = [torch.from_numpy(b).cuda() for b in blocks]
        want = [ref.compress(b, 1, 1) for b in blocks]
        for depth in (1, 2, 3):
            pipe = ctx.pipe(depth)
            tickets = [pipe.submit(d, b.size, 1, 1) for d, b in zip(dev[:depth], blocks[:depth])]
            got = []
            for i in range(len(blocks)):
                got.append(pipe.wait(tickets[i]).tobytes())
                if i + depth < len(blocks):
                    tickets.append(pipe.submit(dev[i + depth], blocks[i + depth].size, 1, 1))
            pipe.close()
            assert got == want, depth
    finally:
        ctx.close()


def test_coder_pool_task_shapes_give_the_same_bytes(ref, torch_cuda):
    """The process's coder pool codes a device-model block as one eight-lane task, four pair tasks or eight scalar tasks depending on
    how busy it is and on BSCGPU_FEATURE_LOW_LATENCY (block.cpp: ps_group).  Every shape must give the reference's bytes, and
    bscgpu_coder_pool_stats must account for every block."""
    from libbsc_amd import GpuContext
    from libbsc_amd.gpu import coder_pool_stats
    torch = torch_cuda
    LOW_LATENCY = 0x10000
    n = (8 << 20) + 12345                                    # eight sub-blocks, device model
    T = api.synth_text_v1(55, n)
    d = torch.from_numpy(T).cuda()
    want = ref.compress(T, 1, 1)
    ctx = GpuContext(0, max_n=n + 4096)
    try:
        pipe = ctx.pipe(4)
        coder_pool_stats(reset=True)
        tickets = [pipe.submit(d, n, 1, 1, 3 | (LOW_LATENCY if i % 2 else 0)) for i in range(4)]      # a burst: the pool fills up while they queue
        got = [pipe.wait(t).tobytes() for t in tickets]
        tickets = [pipe.submit(d, n, 1, 1, 3 | LOW_LATENCY)]                                            # alone on an idle pool
        got += [pipe.wait(t).tobytes() for t in tickets]
        pipe.close()
        st = coder_pool_stats()
        assert all(g == want for g in got)
        assert st["scalar_tasks"] + st["pair_tasks"] + st["eight_lane_task"] + st["host_model"] == 5, st
        assert st["host_model"] == 0 and st["scalar_tasks"] + st["pair_tasks"] >= 3, st                 # the three marked blocks never take the eight-lane task
    finally:
        ctx.close()


def test_own_roundtrip(ref):
    """bsc_compress on the GPU -> our own bsc_decompress (host inverse BWT / inverse ST + QLFC decoders)."""
    for name, T in _inputs():
        for sorter, coder in ((1, 1), (1, 2), (1, 3), (3, 1), (4, 2), (5, 3), (6, 1), (7, 2), (8, 1)):
            blk = api.bsc_compress(T, sorter, coder)
            assert api.bsc_decompress(blk) == T.tobytes(), (name, sorter, coder)


def test_bsc_file_container_against_reference_cli(tmp_path, torch_cuda):
    """files we write unpack with the reference `bsc d`; files the reference writes with preprocessing off
    (`bsc e -p -t`) unpack with our driver, and hold the same blocks byte for byte."""
    import os, subprocess
    from libbsc_amd import cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bsc = os.path.join(root, "oracle", "_ref", "bsc")
    if not os.path.exists(bsc):
        pytest.skip("reference CLI not built (make -C oracle ref)")
    data = api.synth_text_v1(41, (3 << 20) + 12345)
    src = tmp_path / "in.txt"; data.tofile(src)
    ours = tmp_path / "ours.bsc"; theirs = tmp_path / "theirs.bsc"
    env = dict(os.environ, OMP_NUM_THREADS="8")
    for flags_cli, sorter, coder, lzp in (("-b1pt -m0 -e1", 1, 1, (0, 0)), ("-b1pt -m5 -e2", 5, 2, (0, 0)),
                                          ("-b1t -m0 -e1", 1, 1, (15, 128)), ("-b1t -m0 -e1 -H16 -M32", 1, 1, (16, 32))):
        if lzp[0]:                                        # give LZP something to find
            from libbsc_amd.synth import synth_repeat_v1
            data = synth_repeat_v1(42, (3 << 20) + 12345, 200_000)
            data.tofile(src)
        cli.compress_file(str(src), str(ours), 1 << 20, sorter, coder, lzp_hash=lzp[0], lzp_min=lzp[1])
        out = tmp_path / "back.bin"
        r = subprocess.run([bsc, "d", str(ours), str(out)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(out, np.uint8), data)
        r = subprocess.run([bsc, "e", str(src), str(theirs)] + flags_cli.split(), capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(ours, "rb").read() == open(theirs, "rb").read()
        out2 = tmp_path / "back2.bin"
        cli.decompress_file(str(theirs), str(out2))
        assert np.array_equal(np.fromfile(out2, np.uint8), data)


@pytest.mark.parametrize("n", [256 * 1024 - 1, 256 * 1024, 256 * 1024 + 1, (4 << 20) - 1, 4 << 20, (4 << 20) + 1,
                               (16 << 20) - 1, 16 << 20, (16 << 20) + 1])
def test_sub_block_count_boundaries(ref, n):
    """SURVEY §4 test plan: sizes around the 1/2/4/8 sub-block steps of coder.cpp:52-59."""
    T = api.synth_text_v1(50 + (n & 7), n)
    want = ref.compress(T, 1, 1)
    assert api.bsc_compress(T, 1, 1) == want, n
    if n <= (4 << 20) + 1:
        assert api.bsc_compress(T, 5, 3) == ref.compress(T, 5, 3), n


# ---- LZP on (the reference CLI's default is -H15 -M128, bsc.cpp:73-75) -----------------------------
def _lzp_inputs():
    from libbsc_amd.synth import synth_repeat_v1, synth_text_v1
    return [("repeat300k", synth_repeat_v1(3, 300_000, 4000)),
            ("repeat5m", synth_repeat_v1(5, 5 << 20, 70_000)),
            ("text300k", synth_text_v1(3, 300_000)),                      # LZP does not pay: block is written without it
            ("zeros70k", np.zeros(70_000, np.uint8)),                     # LZP output of 90 bytes, aux indexes of that size
            ("zeros300", np.zeros(300, np.uint8))]


@pytest.mark.parametrize("lzp", [(15, 128), (16, 32), (18, 8), (15, 4), (20, 255)])
def test_bsc_compress_with_lzp_matches_reference(ref, lzp):
    h, m = lzp
    for name, T in _lzp_inputs():
        for sorter, coder in ((1, 1), (5, 2), (1, 3)):
            for features in (1, 3):
                want = ref.compress(T, sorter, coder, lzp_hash=h, lzp_min=m, features=features)
                got = api.bsc_compress(T, sorter, coder, lzp_hash=h, lzp_min=m, features=features)
                assert got == want, (name, sorter, coder, h, m, features)
        blk = api.bsc_compress(T, 1, 1, lzp_hash=h, lzp_min=m)
        assert api.bsc_decompress(blk) == T.tobytes(), (name, h, m)
        assert ref.decompress(blk) == T.tobytes(), (name, h, m)
        assert api.bsc_compress(T, 1, 1, lzp_hash=h, lzp_min=m, inplace=True) == blk, (name, "inplace")


def test_lzp_golden_blocks_on_gpu(torch_cuda):
    """Blocks the reference wrote with LZP on (tests/golden/make_lzp_golden.py); needs no reference at run time."""
    import json
    import os
    from libbsc_amd.synth import synth_repeat_v1
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lzp_golden.json")))
    for e in gold["blocks"]:
        T = synth_repeat_v1(e["seed"], e["n"], e["period"])
        blk = api.bsc_compress(T, e["sorter"], e["coder"], lzp_hash=e["hash"], lzp_min=e["minlen"], features=e["features"])
        assert len(blk) == e["size"] and hashlib.md5(blk).hexdigest() == e["md5"], e
        assert blk[:28].hex() == e["header"]
        assert api.bsc_decompress(blk) == T.tobytes()


def test_bsc_compress_bad_lzp_parameters():
    T = np.zeros(1000, np.uint8)
    assert api.bsc_compress(T, 1, 1, lzp_hash=9, lzp_min=32) == api.BAD_PARAMETER      # libbsc.cpp:255-256
    assert api.bsc_compress(T, 1, 1, lzp_hash=15, lzp_min=3) == api.BAD_PARAMETER
    assert api.bsc_compress(T, 1, 1, lzp_hash=0, lzp_min=32) == api.BAD_PARAMETER


def test_concurrent_bsc_compress_calls(ref):
    """Callers may be concurrent (the reference CLI compresses blocks from an OpenMP team, bsc.cpp:197): GPU stages queue,
    host stages overlap, results stay byte-identical."""
    import threading
    from libbsc_amd.synth import synth_repeat_v1, synth_text_v1
    jobs = [(synth_text_v1(50 + i, 700_000 + 31 * i), 1, 1, 0, 0) for i in range(4)]
    jobs += [(synth_text_v1(60 + i, 400_000), 5, 2, 0, 0) for i in range(2)]
    jobs += [(synth_repeat_v1(70 + i, 900_000, 20_000), 1, 1, 15, 32) for i in range(2)]
    jobs += [(synth_text_v1(80, 5 << 20), 1, 1, 0, 0)]              # larger than the others: the default context is re-created
    want = [ref.compress(T, so, co, lzp_hash=h, lzp_min=m) for T, so, co, h, m in jobs]
    got = [None] * len(jobs)

    def work(i):
        T, so, co, h, m = jobs[i]
        got[i] = api.bsc_compress(T, so, co, lzp_hash=h, lzp_min=m)

    for rnd in range(2):
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        for t in ths: t.start()
        for t in ths: t.join()
        assert got == want


def test_reference_cli_relinked_against_product_library(tmp_path, torch_cuda):
    """INTEGRATION.md §A end to end: the reference's own CLI (bsc.cpp + its filters, compiled from /root/reference by
    `make -C oracle dropin`) linked against libbsc_mi355x.so instead of libbsc writes the same .bsc files as the unmodified
    reference binary, with the reference's default options (LZP on, parallel blocks) and a few others, and unpacks them."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ours = os.path.join(root, "oracle", "_ref", "bsc_mi355x")
    theirs = os.path.join(root, "oracle", "_ref", "bsc")
    if not (os.path.exists(ours) and os.path.exists(theirs)):
        pytest.skip("drop-in CLI not built (make -C oracle ref dropin; needs /root/reference)")
    from libbsc_amd.synth import synth_repeat_v1
    data = np.concatenate([api.synth_text_v1(91, (2 << 20) + 777), synth_repeat_v1(92, 3 << 20, 150_000)])
    src = tmp_path / "in.bin"; data.tofile(src)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    from libbsc_amd.cli import parse_container
    for flags in ("-b1", "-b1 -t", "-b2 -m5 -e2", "-b1 -p -e0 -t", "-b8 -H16 -M64 -m0"):
        a = tmp_path / "a.bsc"; b = tmp_path / "b.bsc"; back = tmp_path / "back.bin"
        r = subprocess.run([ours, "e", str(src), str(a)] + flags.split(), capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        r = subprocess.run([theirs, "e", str(src), str(b)] + flags.split(), capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        fa, fb = open(a, "rb").read(), open(b, "rb").read()
        if "-t" in flags.split():                      # blocks in order: the files are identical
            assert fa == fb, flags
        # with parallel blocks the CLI appends them in completion order (bsc.cpp:397-418): same set of blocks
        assert sorted(parse_container(fa)) == sorted(parse_container(fb)), flags
        r = subprocess.run([ours, "d", str(b), str(back)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(back, np.uint8), data), flags


@pytest.mark.parametrize("coder", [1, 2, 3])
def test_serial_framing_rules_with_parallel_execution(ref, coder):
    """features without MULTITHREADING (what the reference CLI passes when it compresses blocks in parallel, bsc.cpp:188)
    selects the serial framing rules of coder.cpp:111-155; sub-blocks are still coded concurrently and re-coded serially
    only when a sub-block's budget would have differed — incompressible and barely compressible blocks included."""
    rng = np.random.default_rng(33)
    extra = [("noise1m", rng.integers(0, 256, 1 << 20, dtype=np.uint8)),
             ("noise+text", np.concatenate([rng.integers(0, 256, 700_000, dtype=np.uint8), api.synth_text_v1(9, 300_000)])),
             ("lowent", rng.integers(0, 200, 600_000, dtype=np.uint8))]
    for name, T in _inputs() + extra:
        for features in (0, 1):
            want = ref.compress(T, 1, coder, features=features)
            assert api.bsc_compress(T, 1, coder, features=features) == want, (name, coder, features)


@pytest.mark.slow
def test_full_size_baseline_configs_golden(torch_cuda):
    """BASELINE configs 4 and 5 and one deep-LCP block at full size against reference outputs committed in
    tests/golden/golden_big.json (tests/golden/make_golden_big.py; SURVEY.md §8c pins the same config-5 md5s):
    seeds 10..17 at 64 MiB through BWT + QLFC static (the blocks `bench.py --gpus N` compresses), seed 3 at 128 MiB through
    ST5 / ST6, and 64 MiB of long repeated passages (many prefix-doubling rounds).  Needs no reference at run time."""
    import json, os
    from libbsc_amd import GpuContext
    from libbsc_amd.synth import synth_repeat_v1
    torch = torch_cuda
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_big.json")))
    ctx = GpuContext(0, max_n=(128 << 20) + 4096)
    try:
        cache = {}
        for e in g["blocks"]:
            if e["tag"] == "max-block":          # 1 GiB: test_max_block_1gib_golden
                continue
            key = json.dumps(e["gen"], sort_keys=True) + str(e["n"])
            if key not in cache:
                cache.clear()
                gen = e["gen"]
                cache[key] = api.synth_text_v1(gen["seed"], e["n"]) if gen["kind"] == "synth" else synth_repeat_v1(gen["seed"], e["n"], gen["period"])
            T = cache[key]
            assert hashlib.md5(T.tobytes()).hexdigest() == e["input_md5"]
            d = torch.from_numpy(T).cuda()
            blk = ctx.compress_device(d, T.size, e["sorter"], e["coder"], e["features"]).tobytes()
            assert (len(blk), hashlib.md5(blk).hexdigest()) == (e["size"], e["md5"]), (e["tag"], e["gen"], e["sorter"])
            if "bwt_index" in e:
                out = torch.empty_like(d)
                idx, _ = ctx.bwt_device(d, out, T.size)
                assert idx == e["bwt_index"] and hashlib.md5(out.cpu().numpy().tobytes()).hexdigest() == e["bwt_md5"], (e["tag"], e["gen"])
            del d
    finally:
        ctx.close()


@pytest.mark.slow
def test_max_block_1gib_golden(torch_cuda):
    """The format's maximum block, n = 2^30 (libbsc.cpp:221), BWT + QLFC static, against the reference's size + md5 committed in
    tests/golden/golden_big.json (row `max-block`, tests/golden/make_golden_big.py).  The device model's arena (230 bytes per block
    byte) does not fit beside a 1 GiB sorter arena, so the model runs on the host: sorter, QLFC front end and container at
    their largest shapes are what this pins."""
    import json, os
    from libbsc_amd import GpuContext
    torch = torch_cuda
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_big.json")))
    rows = [e for e in g["blocks"] if e["tag"] == "max-block"]
    assert rows, "golden_big.json has no max-block row"
    e = rows[0]
    n = e["n"]
    assert n == 1 << 30
    T = api.synth_text_v1(e["gen"]["seed"], n)
    assert hashlib.md5(T.tobytes()).hexdigest() == e["input_md5"]
    ctx = GpuContext(0, max_n=n)
    try:
        d = torch.from_numpy(T).cuda()
        out = torch.empty_like(d)
        idx, _ = ctx.bwt_device(d, out, n)
        assert idx == e["bwt_index"] and hashlib.md5(out.cpu().numpy().tobytes()).hexdigest() == e["bwt_md5"]
        del out
        blk = ctx.compress_device(d, n, e["sorter"], e["coder"], e["features"]).tobytes()
        assert (len(blk), hashlib.md5(blk).hexdigest()) == (e["size"], e["md5"])
    finally:
        ctx.close()


def test_two_contexts_driven_from_two_threads(ref, torch_cuda):
    """Two bscgpu contexts in one process, each driven by its own thread at the same time (what the per-device default
    contexts behind bsc_compress do on a multi-GPU node; with one GPU both land on device 0): per-context streams, arenas
    and kernel attributes must not interfere.  Sizes are chosen so both use the large-tile scatter shape."""
    import threading
    from libbsc_amd import GpuContext
    torch = torch_cuda
    ndev = torch.cuda.device_count()
    blocks = [api.synth_text_v1(70 + i, (6 << 20) + 4097 * i) for i in range(4)]
    want = [ref.compress(b, 1, 1) for b in blocks]
    got = [None] * len(blocks)
    errs = []

    def work(k):
        try:
            dev = k % ndev
            ctx = GpuContext(dev, max_n=(8 << 20))
            try:
                for i in range(k, len(blocks), 2):
                    d = torch.from_numpy(blocks[i]).to(f"cuda:{dev}")
                    got[i] = ctx.compress_device(d, blocks[i].size, 1, 1).tobytes()
            finally:
                ctx.close()
        except Exception as e:        # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errs, errs
    assert got == want


def test_eight_sub_block_inputs_both_range_coder_modes(ref, torch_cuda):
    """Blocks of >= 16 MiB have eight sub-blocks: the size from which the eight-lane AVX2 range coder (BSC_RC_SIMD=8) can take a
    block.  A 17 MiB text block, and one whose last part is noise — several of its sub-blocks cannot be coded inside their
    budget, the SIMD coder has to give up and the scalar coders / the host model decide exactly as the reference does.  This test
    runs in the default mode (pairs); test_simd_range_coder_mode re-runs it with the eight-lane coder."""
    from libbsc_amd import GpuContext
    torch = torch_cuda
    rng = np.random.default_rng(33)
    n = 17 << 20
    cases = [("text17m", api.synth_text_v1(5, n)),
             ("text+noise", np.concatenate([api.synth_text_v1(6, 15 << 20), rng.integers(0, 256, 2 << 20, dtype=np.uint8)]))]
    ctx = GpuContext(0, max_n=n + 4096)
    try:
        for name, T in cases:
            d = torch.from_numpy(T).cuda()
            got = ctx.compress_device(d, T.size, 1, 1).tobytes()
            assert got == ref.compress(T, 1, 1), name
    finally:
        ctx.close()


def test_close_sub_block_cuts_inside_one_chunk(ref, torch_cuda):
    """The adaptive split (coder.cpp:83-99) cuts where the sampled run starts reach total / nBlocks.  A block that is constant
    except for one short noisy stretch has a sorted image that is constant except for ~4 KiB at its end, so the cuts are ~0.5 KiB
    apart and ALL sub-blocks but the first lie inside one 16 KiB workgroup chunk of qf_apply_kernel (which once assumed at most
    two sub-blocks per chunk).  4 sub-blocks at 4 MiB, 8 at 16 MiB; BWT and ST5; every coder."""
    from libbsc_amd import GpuContext
    torch = torch_cuda
    rng = np.random.default_rng(77)
    for n in (4 << 20, 16 << 20):
        T = np.zeros(n, np.uint8)
        T[n // 2: n // 2 + 4096] = rng.integers(1, 256, 4096, dtype=np.uint8)
        ctx = GpuContext(0, max_n=n + 4096)
        try:
            d = torch.from_numpy(T).cuda()
            for sorter, coder in ((1, 1), (1, 2), (1, 3), (5, 1)):
                want = ref.compress(T, sorter, coder)
                got = ctx.compress_device(d, n, sorter, coder).tobytes()
                assert got == want, (n, sorter, coder)
        finally:
            ctx.close()
        assert api.bsc_compress(T, 1, 1) == ref.compress(T, 1, 1), (n, "host api")


def _child_env(**extra):
    """Environment for child EXECUTABLES that link the regular library themselves (bsc_mgpu, job_bench, the reference CLI): under
    tools/asan_run.py the parent python carries a preloaded sanitizer runtime, which an uninstrumented multi-threaded C++ program does
    not survive (round 4's one 'failed on its own assertion' under the sanitizer runtime was test_cxx_multi_gpu_file_compressor:
    'AddressSanitizer failed to deallocate ... UnsetAlternateSignalStack' at a thread's exit inside bsc_mgpu — the runtime's, not a report
    about the product)."""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS", "BSC_LIB_OVERRIDE")}
    env.update(extra)
    return env


def _rerun_with_env(selection, **env):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_compress.py"), os.path.join(root, "tests", "test_gpu_device.py"),
                        "-q", "-x", "-k", selection], capture_output=True, text=True, env=dict(os.environ, **env), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_simd_range_coder_mode(torch_cuda):
    """The opt-in eight-lane range coder (qlfc_encode_static_pstream_x8) end to end: eight-sub-block inputs, the 64 MiB golden
    block and the pipelined path, in a child process with BSC_RC_SIMD=8."""
    _rerun_with_env("eight_sub_block_inputs or full_size_64m_block_golden or pipe_matches_sync_path", BSC_RC_SIMD="8")


def test_device_coder_arena_that_does_not_fit_is_a_decline(torch_cuda):
    """BSC_DEVCODER_FAIL_ALLOC=1 makes the device coder's arena allocation fail: blocks must take the host model (same bytes),
    not return LIBBSC_GPU_NOT_ENOUGH_MEMORY, and the allocation must not be retried per block (pipelined path included)."""
    _rerun_with_env("eight_sub_block_inputs or pipe_matches_sync_path or close_sub_block_cuts", BSC_DEVCODER_FAIL_ALLOC="1")


def test_landing_zones_fall_back_to_hiphostmalloc_when_registration_is_refused(torch_cuda):
    """BSC_PIN_REGISTER_FAIL=1 makes hipHostRegister of a landing zone "fail" (RLIMIT_MEMLOCK, containers without large-page
    registration): run arrays and p stream must land in hipHostMalloc memory instead — same bytes, no error — on the host-model
    path, the device-model path and the pipelined path."""
    _rerun_with_env("eight_sub_block_inputs or pipe_matches_sync_path or sub_block_count_boundaries", BSC_PIN_REGISTER_FAIL="1")


def test_packed_stream_falls_back_to_16_bit_entries_for_runs_of_thousands(ref, torch_cuda):
    """A periodic block of 200 distinct symbols has a BWT of 200 runs of 16 384 bytes: ~44 decisions per run, 64 runs of a wavefront
    beyond the staging buffer of dc_pstream — the 13-bit packed stream is void for such a block (DM_P13_OVER) and the block's stream is
    written again as 16-bit entries.  Same bytes as the reference, on the device-model path (process counter) and without a redo."""
    import ctypes as C
    from libbsc_amd import GpuContext, _native
    torch = torch_cuda
    L = _native.lib()
    L.bscgpu_process_counter.restype = C.c_longlong
    L.bscgpu_process_counter.argtypes = [C.c_int]
    T = np.tile(np.arange(1, 201, dtype=np.uint8), 1 << 14)
    ctx = GpuContext(0, max_n=T.size + 4096)
    try:
        c0 = [L.bscgpu_process_counter(k) for k in (1, 2)]
        got = ctx.compress_device(torch.from_numpy(T).cuda(), T.size, 1, 1).tobytes()
        assert got == ref.compress(T, 1, 1)
        c1 = [L.bscgpu_process_counter(k) for k in (1, 2)]
        assert c1[0] == c0[0] + 1 and c1[1] == c0[1], (c0, c1)        # on the device model, not redone on the host model
    finally:
        ctx.close()


def test_p_stream_as_16_bit_entries_when_the_packed_form_is_off(torch_cuda):
    """BSC_PS13=0 (BSCGPU_OPT_DC_PACKED_STREAM = 0): the stream crosses as 16-bit entries with run-start marks, as before round 6 — same bytes
    on the synchronous path, the pipelined path and the 64 MiB golden block (the coders for that form stay in use for the fast coder and for
    blocks whose packed stream is void)."""
    _rerun_with_env("eight_sub_block_inputs or pipe_matches_sync_path or full_size_64m_block_golden or sub_block_count_boundaries", BSC_PS13="0")


def test_mixed_radix_first_sort_keys_variant(torch_cuda):
    """BSC_BWT_RADIX=1 (round 6, off by default: profiles/r06/first_sort_keys.txt): the BWT's first-sort key as the base-K number of one more
    character than bit packing holds (28 symbols: 13 instead of 12; 17-19 and 24-30 symbols likewise) — other digits, other group structure
    behind the first sort, same BWT: the 16 MiB text block, the edge corpus against the reference, alphabets of 17 .. 200 symbols and the
    64 MiB golden block."""
    _rerun_with_env("bwt_device_resident_16m or full_size_64m_block_golden or bwt_matches_reference or front_end_rank_paths", BSC_BWT_RADIX="1")


def test_p_stream_copies_through_the_hip_runtime_when_the_dma_path_is_off(torch_cuda):
    """The p stream normally leaves the device through the HSA runtime's DMA copy (dma_copy.h: HSA signals, host-side waits, host-side
    guard of the device buffer's reuse); BSC_D2H_DMA=0 keeps hipMemcpyAsync + events.  Both must give the same bytes on the synchronous
    path, the pipelined path (buffer reuse two blocks later) and the 64 MiB golden block."""
    _rerun_with_env("eight_sub_block_inputs or pipe_matches_sync_path or full_size_64m_block_golden", BSC_D2H_DMA="0")


def test_p_stream_dma_path_is_the_one_in_use(torch_cuda):
    """In this process (torch's HIP runtime) the DMA path must be available — a silent decline would put the copy kernels back."""
    import ctypes
    from libbsc_amd import _native
    lib = _native.lib()
    lib.bscgpu_d2h_dma_available.restype = ctypes.c_int
    assert lib.bscgpu_d2h_dma_available() == 1


def test_c_job_bench_verifies_against_the_committed_reference_output(torch_cuda):
    """tools/job_bench.cpp (built next to the library): bench.py's workload through bscgpu_job_* from a C++ caller — announced total,
    tapered head and tail, low-latency last blocks; its last block must be the committed reference output (its own md5 check, exit code 3
    on a mismatch), for the static and the fast coder."""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "libbsc_amd", "lib", "job_bench")
    assert os.path.exists(exe), "python -m libbsc_amd.build builds it"
    for coder in (1, 3):
        r = subprocess.run([exe, "--steps", "7", "--warmup", "1", "--contexts", "2", "--depth", "2", "--coder", str(coder)], capture_output=True, text=True, env=_child_env())
        assert r.returncode == 0, r.stdout + r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["verified"] is True and d["steps"] == 7 and d["value"] > 0, d


def test_prefix_doubling_from_the_first_round(torch_cuda):
    """BSC_BWT_TEXTROUNDS=0: the inverse suffix array is built by the first seg and every round doubles (the path that text
    blocks no longer take since the rounds on text keys): BWT parity cases and the 64 MiB golden block."""
    _rerun_with_env("bwt_matches_reference or bwt_device_resident_16m or full_size_64m_block_golden", BSC_BWT_TEXTROUNDS="0")


def test_job_driver_on_the_gpu(ref, torch_cuda):
    """bscgpu_job_* (include/bscgpu.h, csrc/host/job.cpp) with the library's own executor: two pipes on the one GPU of this box, blocks of
    every kind — sizes from tiny to several MiB, all three coders, a sort transform, LZP, an incompressible one — added in order and
    collected in order; every block must be the reference's bytes.  (The scheduler itself is tested on CPU with several stand-in devices:
    tests/test_job_driver.py.)"""
    import ctypes as C
    from libbsc_amd import _native as N
    from libbsc_amd.synth import synth_text_v1
    L = N.lib()
    vp, ci = C.c_void_p, C.c_int
    L.bscgpu_job_create.argtypes = [C.POINTER(vp), C.POINTER(ci), ci, ci, ci, C.c_int64]
    L.bscgpu_job_add.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci]
    L.bscgpu_job_wait.argtypes = [vp, ci]
    L.bscgpu_job_destroy.argtypes = [vp]
    L.bscgpu_job_destroy.restype = None
    rng = np.random.default_rng(3)
    blocks = [(synth_text_v1(40 + i, int(n)), so, co, lz) for i, (n, so, co, lz) in enumerate(
        [(5 << 20, 1, 1, (0, 0)), (20, 1, 1, (0, 0)), (3 << 20, 1, 2, (0, 0)), (1 << 20, 5, 3, (0, 0)), (2_500_000, 1, 1, (15, 32)),
         (300_000, 1, 1, (0, 0)), (6 << 20, 1, 1, (0, 0)), (70_000, 6, 1, (0, 0)), (1 << 20, 1, 3, (16, 128)), (4 << 20, 1, 1, (0, 0))])]
    blocks.insert(4, (rng.integers(0, 256, 400_000, dtype=np.uint8), 1, 1, (0, 0)))
    h = vp()
    assert L.bscgpu_job_create(C.byref(h), None, 0, 2, 2, (6 << 20) + 4096) == 0
    outs = []
    try:
        for b, (T, so, co, lz) in enumerate(blocks):
            outs.append(np.empty(T.size + 28, np.uint8))
            assert L.bscgpu_job_add(h, N.np_ptr(T), N.np_ptr(outs[-1]), T.size, lz[0], lz[1], so, co, 3) == b
        for b, (T, so, co, lz) in enumerate(blocks):
            r = L.bscgpu_job_wait(h, b)
            want = ref.compress(T, so, co, lzp_hash=lz[0], lzp_min=lz[1])
            assert r == len(want) and outs[b][:r].tobytes() == want, (b, T.size, so, co, lz, r)
    finally:
        L.bscgpu_job_destroy(h)


def test_lzp_blocks_take_the_device_model_and_survive_a_redo(ref, torch_cuda):
    """The reference CLI's default is LZP on (bsc.cpp:73-75): such a block is just another byte block to the GPU stage, so its static
    model runs on the device too.  (a) text with a long duplicate: LZP pays, the block takes the device model; (b) low-entropy bytes +
    noise + the same noise again: LZP removes the copy, the sorted block's last sub-block is pure noise and has to be stored raw, which
    the device path cannot do (it never copied the run arrays) — the block is redone with the model on the host, and that redo uploads
    the LZP OUTPUT again, not the original block.  Synchronous and pipelined entry points, both framings; bytes = the reference's."""
    import ctypes as C
    from libbsc_amd import GpuContext, _native as N
    from libbsc_amd.synth import synth_text_v1
    L = N.lib()
    L.bscgpu_process_counter.restype = C.c_longlong
    L.bscgpu_process_counter.argtypes = [C.c_int]
    from libbsc_amd.synth import synth_repeat_v1
    rng = np.random.default_rng(8)
    a = synth_repeat_v1(9, 8 << 20, 40_000, 48)                  # LZP output 4.7 MB: four sub-blocks, all of them compress
    lut = np.arange(256, dtype=np.uint8)                         # the same with its alphabet moved to the bottom of the byte order, then
    for i, v in enumerate([10, 32] + list(range(97, 123))):      # 2 MB of noise: the sorted block's third sub-block is pure noise (the
        lut[v] = i                                               # reference stores it raw: 768992 -> 768992 bytes)
    b = np.concatenate([lut[a], rng.integers(0, 256, 2_000_000, dtype=np.uint8)])
    ctx = GpuContext(0, max_n=b.size + 4096)
    pipe = ctx.pipe(3)
    try:
        for name, T, lz, want_redo in (("repeats", a, (16, 32), False), ("repeats+noise", b, (16, 32), True)):
            for feat in (3, 1):
                want = ref.compress(T, 1, 1, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
                assert not isinstance(want, int) and (want[9] != 0 or want[10] != 0), name       # the mode word says LZP stayed on
                c0 = [L.bscgpu_process_counter(k) for k in (1, 2, 3)]
                got = api.bsc_compress(T, 1, 1, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
                assert got == want, (name, feat, "sync")
                tk = pipe.submit_host(T, 1, 1, lz[0], lz[1], feat)
                assert pipe.wait(tk).tobytes() == want, (name, feat, "pipe")
                c1 = [L.bscgpu_process_counter(k) for k in (1, 2, 3)]
                assert c1[0] - c0[0] == 2 and c1[2] - c0[2] == 2, (name, c0, c1)                  # both took the device model, as LZP blocks
                assert (c1[1] - c0[1] == 2) == want_redo, (name, c0, c1)
    finally:
        pipe.close(); ctx.close()


def test_fast_coder_on_the_device_model(ref, torch_cuda):
    """-e0 (qlfc.cpp:1135-1336) behind the device model: its one counter per decision is the static coder's char family with shift
    updates (devcoder.hip: devcoder_pstream_fast; the chain model is pinned on CPU by tools/devcoder_fast_sim.cpp), the host codes the
    13- / 11-bit entries (qlfc_encode_fast_pstream).  Text with 2, 4 and 8 sub-blocks, 256-symbol data (all 7-bit rank exponents), long
    runs (run-length mantissas of more than 5 bits: the second rate class), a block with a raw sub-block (redo on the host model) and an
    LZP block; synchronous and pipelined, both framings; bytes = the reference's, and the counters say the device model ran."""
    import ctypes as C
    from libbsc_amd import GpuContext, _native as N
    from libbsc_amd.synth import synth_repeat_v1, synth_text_v1
    L = N.lib()
    L.bscgpu_process_counter.restype = C.c_longlong
    L.bscgpu_process_counter.argtypes = [C.c_int]
    rng = np.random.default_rng(17)
    skew = np.where(rng.random(3 << 20) < 0.85, 0, rng.integers(0, 256, 3 << 20)).astype(np.uint8)   # 256 symbols, ~0.3 runs per byte
    longruns = np.repeat(rng.integers(0, 9, 40_000, dtype=np.uint8), rng.integers(1, 400, 40_000)).astype(np.uint8)
    cases = [("text1m", synth_text_v1(1, 1 << 20), (0, 0)), ("text5m", synth_text_v1(3, 5 << 20), (0, 0)), ("text20m", synth_text_v1(4, 20 << 20), (0, 0)),
             ("skew256", skew, (0, 0)), ("longruns", longruns, (0, 0)),
             ("zeros+noise", np.concatenate([np.zeros(2_600_000, np.uint8), rng.integers(0, 256, 1_700_000, dtype=np.uint8)]), (0, 0)),
             ("repeats-lzp", synth_repeat_v1(9, 8 << 20, 40_000, 48), (16, 32))]
    ctx = GpuContext(0, max_n=(20 << 20) + 4096)
    pipe = ctx.pipe(3)
    try:
        for name, T, lz in cases:
            for feat in (3, 1):
                want = ref.compress(T, 1, 3, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
                c0 = L.bscgpu_process_counter(1)
                got = api.bsc_compress(T, 1, 3, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
                assert got == want, (name, feat, "sync")
                tk = pipe.submit_host(T, 1, 3, lz[0], lz[1], feat)
                assert pipe.wait(tk).tobytes() == want, (name, feat, "pipe")
                assert L.bscgpu_process_counter(1) - c0 == 2, (name, "the device model was not used")
                assert api.bsc_decompress(got) == T.tobytes()
        T = cases[2][1]                                                      # ST5 output through the same model
        assert api.bsc_compress(T, 5, 3) == ref.compress(T, 5, 3)
    finally:
        pipe.close(); ctx.close()
    # the eight-lane SIMD range coder on this stream (per-lane precision), forced on in a child process (the knob is read once)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
from libbsc_amd import GpuContext, api
from libbsc_amd.gpu import coder_pool_stats
from oracle.refbind import Ref
T = api.synth_text_v1(4, 20 << 20)
want = Ref().compress(T, 1, 3)
ctx = GpuContext(0, max_n=T.size + 4096); pipe = ctx.pipe(3)
for _ in range(3):
    assert pipe.wait(pipe.submit_host(T, 1, 3, 0, 0, 3)).tobytes() == want
assert coder_pool_stats()["eight_lane_task"] == 3, coder_pool_stats()
print("eight lanes ok")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, BSC_RC_SIMD="8"), cwd=root)
    assert r.returncode == 0 and "eight lanes ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_cxx_multi_gpu_file_compressor(tmp_path, torch_cuda):
    """csrc/driver/bsc_mgpu.cpp: the reference CLI's block loop in C++ over bscgpu_job_* (every GPU of the node, blocks streamed through a
    window of buffers, written in order).  Its files must be the reference's `bsc e ... -t` files byte for byte — defaults (LZP -H15
    -M128), -p, a sort transform with the fast coder —, unpack with the reference's `bsc d`, and it must unpack the reference's files."""
    import os
    import subprocess
    from libbsc_amd.synth import synth_repeat_v1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bsc = os.path.join(root, "oracle", "_ref", "bsc")
    exe = os.path.join(root, "libbsc_amd", "lib", "bsc_mgpu")
    if not os.path.exists(bsc):
        pytest.skip("reference CLI not built (make -C oracle ref)")
    assert os.path.exists(exe), "python -m libbsc_amd.build builds it"
    env = _child_env(OMP_NUM_THREADS="8")
    data = np.concatenate([api.synth_text_v1(43, (9 << 20) + 777), synth_repeat_v1(44, 5 << 20, 150_000)])      # 15 blocks of 1 MiB, the last one short
    src = tmp_path / "in.bin"; data.tofile(src)
    for flags in ("-b1", "-b1 -p", "-b2 -p -m5 -e0", "-b3 -e2 -H16 -M32"):
        ours = tmp_path / "ours.bsc"; theirs = tmp_path / "theirs.bsc"; back = tmp_path / "back.bin"
        r = subprocess.run([exe, "e", str(src), str(ours)] + flags.split() + ["-C2", "-D2"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        r = subprocess.run([bsc, "e", str(src), str(theirs)] + flags.split() + ["-t"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(ours, "rb").read() == open(theirs, "rb").read(), flags
        r = subprocess.run([bsc, "d", str(ours), str(back)], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and np.array_equal(np.fromfile(back, np.uint8), data), flags
        r = subprocess.run([exe, "d", str(theirs), str(back)], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and np.array_equal(np.fromfile(back, np.uint8), data), (flags, r.stdout + r.stderr)


@pytest.mark.slow
def test_large_odd_sized_block_takes_the_device_model(ref, torch_cuda):
    """One block well beyond the benchmark's size and not a power of two (200 000 033 bytes: 12 GB of sorter arena + 46 GB of device
    model): BWT and ST5 through the static coder against the compiled reference, with the process counter proving that the model ran on
    the GPU.  (The API's limit is 1 GiB per block, libbsc.cpp:259; there the device model's arena — 230 bytes per block byte — no longer
    fits next to the sorter's and the block takes the host model: a decline, covered by test_device_coder_arena_that_does_not_fit_...)"""
    from libbsc_amd import GpuContext, _native as N
    import ctypes as C
    torch = torch_cuda
    n = 200_000_033
    T = api.synth_text_v1(77, n)
    L = N.lib()
    L.bscgpu_process_counter.restype = C.c_longlong
    before = L.bscgpu_process_counter(1)
    ctx = GpuContext(0, max_n=n + 4096)
    try:
        d = torch.from_numpy(T).cuda()
        for sorter in (1, 5):
            got = ctx.compress_device(d, n, sorter, 1).tobytes()
            assert got == ref.compress(T, sorter, 1), sorter
    finally:
        ctx.close()
    assert L.bscgpu_process_counter(1) - before == 2
