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
