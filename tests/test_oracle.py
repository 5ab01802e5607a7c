"""CPU tests pinning the plain-C oracle restatement (oracle/bsc_oracle.c) against the reference itself
(oracle/_ref, where it is built) and against the committed golden fixtures (tests/golden/)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def orc():
    from oracle.refbind import Oracle, PORT_SO
    if not os.path.exists(PORT_SO):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "port"], check=True)
    return Oracle()


def _cases():
    from libbsc_amd.synth import synth_text_v1
    rng = np.random.default_rng(9)
    out = [("text64k", synth_text_v1(5, 1 << 16)), ("text200k", synth_text_v1(11, 200_000)),
           ("rand4-50k", rng.integers(0, 4, 50_000, dtype=np.uint8)), ("rand256-30k", rng.integers(0, 256, 30_000, dtype=np.uint8)),
           ("zeros-20k", np.zeros(20_000, np.uint8)), ("ab-10k", (np.arange(10_000) % 2).astype(np.uint8)),
           ("runs", np.repeat(rng.integers(0, 50, 800, dtype=np.uint8), rng.integers(1, 400, 800)))]
    for n in (1, 2, 16, 17, 29, 100, 1000):
        out.append((f"tiny{n}", rng.integers(97, 101, n, dtype=np.uint8)))
    return out


def test_oracle_bwt_st_match_reference(orc, ref):
    for name, T in _cases():
        n = T.size
        a = orc.bwt_encode(T, aux=(n >= 16))
        b = ref.bwt_encode(T, aux=(n >= 16))
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2], name
        if n >= 2:
            for k in (3, 4, 5, 6):
                x, y = orc.st_encode(T, k), ref.st_encode(T, k)
                assert np.array_equal(x[0], y[0]) and x[1] == y[1], (name, k)


def test_oracle_coder_matches_reference(orc, ref):
    for name, T in _cases():
        L, _, _ = ref.bwt_encode(T, aux=False)
        r1, m1 = orc.qlfc_transform(L)
        r2, m2 = ref.qlfc_transform(L)
        assert np.array_equal(r1, r2), name
        k = len(set(L.tolist()))
        assert np.array_equal(m1[:min(k + 1, 256)], m2[:min(k + 1, 256)]), name
        for coder in (1, 2, 3):
            assert orc.qlfc_encode_block(L, coder) == ref.qlfc_encode_block(L, coder), (name, coder)
            assert orc.coder_compress(L, coder) == ref.coder_compress(L, coder, features=1), (name, coder)
        assert orc.adler32(T) == ref.adler32(T)


def test_oracle_compress_matches_reference(orc, ref):
    for name, T in _cases():
        for sorter, coder in ((1, 1), (1, 2), (1, 3), (5, 1), (6, 2), (3, 1), (4, 3)):
            assert orc.compress(T, sorter, coder) == ref.compress(T, sorter, coder, features=1), (name, sorter, coder)


def test_oracle_matches_golden_fixtures(orc):
    """Fixtures were produced by tests/golden/make_golden.py from the compiled reference (oracle/_ref);
    they travel with the repo, so this pins the oracle on machines without /root/reference."""
    import hashlib
    from libbsc_amd.synth import synth_text_v1
    g = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    for e in g["blocks"]:
        T = synth_text_v1(e["seed"], e["n"]) if e["kind"] == "synth" else np.frombuffer(bytes.fromhex(e["hex"]), np.uint8)
        assert hashlib.md5(T.tobytes()).hexdigest() == e["input_md5"]
        if e["n"] <= g["oracle_max_n"]:
            blk = orc.compress(T, e["sorter"], e["coder"])
            assert len(blk) == e["size"] and hashlib.md5(blk).hexdigest() == e["md5"], e


def test_long_group_counting_rule():
    """bwt.hip: seg_apply tells the host how many groups the next round's segmented sort cannot take (> 1024 records) and how many
    records lie beyond the first 1024 of such groups, from (SA slot - group rank) alone.  The numpy model of that rule against the
    group sizes counted directly, on keys with long runs of equal values (first seg and a later round's compacted set)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pipeline_model import seg, seg_long_counts
    rng = np.random.default_rng(12)
    for limit, m, kinds in ((1024, 200_000, 40), (64, 50_000, 300), (8, 5_000, 200)):
        keys = np.sort(rng.integers(0, kinds, m).astype(np.uint64) ** 3)           # sorted keys: groups of very different sizes
        sa = rng.permutation(m).astype(np.uint32)
        pos, rank, uns = seg(keys, sa, None, m, m + 100, True)                       # (n > m + 8: no tail suffixes among them)
        n_long, n_excess = seg_long_counts(pos, rank, uns, limit)
        _, sizes = np.unique(rank[uns], return_counts=True)
        assert n_long == int(np.count_nonzero(sizes > limit))
        assert n_excess == int(np.sum(np.maximum(sizes - limit, 0)))
        # a later round: the compacted set keeps SA slots (cpos) and is re-grouped by a finer key
        cpos = pos[uns].astype(np.uint32)
        finer = (rank[uns].astype(np.uint64) << np.uint64(20)) | rng.integers(0, 3, cpos.size).astype(np.uint64)
        order = np.argsort(finer, kind="stable")
        pos2, rank2, uns2 = seg(finer[order], sa[:cpos.size], cpos, cpos.size, m + 100, False)
        n_long2, n_excess2 = seg_long_counts(pos2, rank2, uns2, limit)
        _, sizes2 = np.unique(rank2[uns2], return_counts=True)
        assert n_long2 == int(np.count_nonzero(sizes2 > limit)) and n_excess2 == int(np.sum(np.maximum(sizes2 - limit, 0)))
