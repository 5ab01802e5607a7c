"""CPU tests for the LZP preprocessor: the host encoder must reproduce the reference's stream byte for
byte for every encoder variant the reference selects by (hashSize, minLen) (lzp.cpp:537-557), in both framings
(serial / MULTITHREADING), and our decoder must invert it."""
import hashlib
import json
import os

import numpy as np
import pytest

from libbsc_amd import api
from libbsc_amd.synth import synth_repeat_v1, synth_text_v1

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lzp_golden.json")))

# minLen -> variant: 4/8 small<u32/u64>, 16 small2x, 5..7 medium<u32>, 9..15 medium<u64>, >16 large; hash > 17 generic
MINLENS = (4, 5, 7, 8, 9, 15, 16, 17, 40, 255)
HASHES = (10, 17, 18)


def _corpora():
    rng = np.random.default_rng(5)
    out = [("zeros", np.zeros(70_000, np.uint8)),
           ("repeat", synth_repeat_v1(3, 300_000, 4000)),
           ("repeat_dense_f2", synth_repeat_v1(4, 200_000, 900, noise_every=40)),
           ("text", synth_text_v1(3, 150_000)),
           ("random", rng.integers(0, 256, 100_000, dtype=np.uint8)),
           ("f2_runs", (rng.integers(0, 3, 120_000) * 0x79).astype(np.uint8))]      # bytes 0x00 0x79 0xF2
    return out


@pytest.mark.parametrize("features", [1, 3])
def test_lzp_stage_matches_reference(ref, features):
    for name, T in _corpora():
        for h in HASHES:
            for m in MINLENS:
                want = ref.lzp_compress(T, h, m, features=features)
                got = api.bsc_lzp_compress(T, h, m, features=features)
                assert got == want, (name, h, m, features)
                if not isinstance(got, int):
                    assert api.bsc_lzp_decompress(got, T.size, h, m) == T.tobytes(), (name, h, m)
                    assert ref.lzp_decompress(got, T.size, h, m) == T.tobytes(), (name, h, m)


def test_lzp_tiny_and_boundary_sizes(ref):
    """n - minLen < 32 is refused (lzp.cpp:531); sizes around the main-phase guard; chunk-count thresholds."""
    rng = np.random.default_rng(11)
    for it in range(1500):
        n = int(rng.integers(1, 300))
        T = [np.zeros(n, np.uint8), (rng.integers(0, 2, n) * 0xF2).astype(np.uint8),
             np.tile(rng.integers(0, 256, int(rng.integers(1, 7)), dtype=np.uint8), n)[:n].copy()][it % 3]
        h = int(rng.choice([10, 14, 18])); m = int(rng.choice(MINLENS)); f = int(rng.choice([0, 3]))
        assert api.bsc_lzp_compress(T, h, m, features=f) == ref.lzp_compress(T, h, m, features=f), (n, h, m, f)
    big = synth_repeat_v1(8, (4 << 20) + 5, 50_000)
    for n in (256 * 1024 - 1, 256 * 1024, (4 << 20) - 1, 4 << 20):
        for f in (1, 3):
            assert api.bsc_lzp_compress(big[:n], 15, 32, features=f) == ref.lzp_compress(big[:n], 15, 32, features=f), (n, f)


def test_lzp_bad_parameters():
    T = np.zeros(1000, np.uint8)
    assert api.bsc_lzp_compress(T, 9, 32) == api.BAD_PARAMETER
    assert api.bsc_lzp_compress(T, 29, 32) == api.BAD_PARAMETER
    assert api.bsc_lzp_compress(T, 15, 3) == api.BAD_PARAMETER
    assert api.bsc_lzp_compress(T, 15, 256) == api.BAD_PARAMETER
    assert api.bsc_lzp_compress(np.zeros(40, np.uint8), 15, 32) == api.NOT_COMPRESSIBLE


def test_lzp_golden_vectors():
    """Committed outputs of the reference's bsc_lzp_compress (tests/golden/make_lzp_golden.py); needs no reference."""
    cache = {}
    for e in GOLD["stage"]:
        if e["n"] > (6 << 20):
            continue                                       # the 17 MiB (8-chunk) entries run in the slow test below
        key = (e["seed"], e["n"], e["period"])
        if key not in cache:
            cache[key] = synth_repeat_v1(*key)
        got = api.bsc_lzp_compress(cache[key], e["hash"], e["minlen"], features=e["features"])
        if "error" in e:
            assert got == e["error"], e
        else:
            assert len(got) == e["size"] and hashlib.md5(got).hexdigest() == e["md5"], e


def test_lzp_golden_vectors_eight_chunks():
    T = None
    for e in GOLD["stage"]:
        if e["n"] <= (6 << 20) or e["hash"] != 15 or e["minlen"] not in (8, 32, 128):
            continue
        if T is None:
            T = synth_repeat_v1(e["seed"], e["n"], e["period"])
        got = api.bsc_lzp_compress(T, e["hash"], e["minlen"], features=e["features"])
        assert len(got) == e["size"] and hashlib.md5(got).hexdigest() == e["md5"], e
        assert api.bsc_lzp_decompress(got, T.size, e["hash"], e["minlen"]) == T.tobytes()


def test_lzp_concurrent_callers_share_the_kept_buffers(ref):
    """The parallel framing stages its chunks in a block-sized buffer that is kept between calls (par.h: bigbuf_get / bigbuf_put, one
    cache per process).  Eight threads compress and restore blocks of different sizes at the same time (ctypes releases the GIL): every
    result must be the reference's, whoever had the buffer before."""
    import threading
    blocks = [synth_repeat_v1(20 + k, (1 << 20) + k * 300_001, 20_000 + 1000 * k) for k in range(6)]
    want = [ref.lzp_compress(T, 15, 32, features=3) for T in blocks]
    errors = []

    def worker(t):
        try:
            for it in range(12):
                k = (t + it) % len(blocks)
                got = api.bsc_lzp_compress(blocks[k], 15, 32, features=3)
                assert got == want[k], (t, it, k)
                assert api.bsc_lzp_decompress(got, blocks[k].size, 15, 32) == blocks[k].tobytes(), (t, it, k)
        except Exception as e:                  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors[:3]
