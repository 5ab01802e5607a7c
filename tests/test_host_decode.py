"""CPU tests for the decode side: QLFC decoders, coder framing, inverse BWT, bsc_decompress —
judged against the compiled reference in both directions (we decode what the reference wrote)."""
import numpy as np
import pytest

from libbsc_amd import api
from test_host_coder import _texts


@pytest.mark.parametrize("coder", [1, 2, 3])
def test_qlfc_decode_block_inverts_reference_encoder(ref, coder):
    for name, T in _texts():
        L, _, _ = ref.bwt_encode(T, aux=False)
        enc = ref.qlfc_encode_block(L, coder)
        if isinstance(enc, int):
            continue                                  # incompressible sub-block: nothing to decode
        assert api.bsc_qlfc_decode_block(enc, L.size, coder) == L.tobytes(), (name, coder)


@pytest.mark.parametrize("coder", [1, 2, 3])
@pytest.mark.parametrize("features", [1, 3])
def test_coder_decompress_inverts_reference(ref, coder, features):
    from libbsc_amd.synth import synth_text_v1
    for name, T in _texts() + [("text5m", synth_text_v1(4, 5 << 20))]:
        L, _, _ = ref.bwt_encode(T, aux=False)
        enc = ref.coder_compress(L, coder, features=features)
        if isinstance(enc, int):
            continue
        assert api.bsc_coder_decompress(enc, L.size, coder, features=features) == L.tobytes(), (name, coder, features)
        assert api.bsc_coder_decompress(api.bsc_coder_compress(L, coder, features=features), L.size, coder) == L.tobytes()


def test_bwt_decode_matches_reference(ref):
    rng = np.random.default_rng(2)
    cases = [t for _, t in _texts()]
    for n in (2, 3, 7, 8, 9, 16, 17, 100):
        cases += [rng.integers(0, 256, n, dtype=np.uint8), np.zeros(n, np.uint8), (np.arange(n) % 2).astype(np.uint8)]
    for T in cases:
        L, idx, _ = ref.bwt_encode(T, aux=False)
        back, rc = api.bsc_bwt_decode(L, idx)
        assert rc == 0 and np.array_equal(back, T), T.size
    assert api.bsc_bwt_decode(np.zeros(10, np.uint8), 0)[1] == api.BAD_PARAMETER
    assert api.bsc_bwt_decode(np.zeros(10, np.uint8), 11)[1] == api.BAD_PARAMETER


def test_bsc_decompress_reads_reference_blocks(ref):
    for name, T in _texts():
        for coder in (1, 2, 3):
            blk = ref.compress(T, 1, coder)
            assert api.bsc_decompress(blk) == T.tobytes(), (name, coder)
    # damaged payload / header are detected (adler checks, libbsc.cpp:347,545,616)
    T = _texts()[0][1]
    blk = bytearray(ref.compress(T, 1, 1))
    blk[100] ^= 0x40
    assert api.bsc_decompress(bytes(blk)) == api.DATA_CORRUPT
    blk = bytearray(ref.compress(T, 1, 1)); blk[13] ^= 1
    assert api.bsc_decompress(bytes(blk)) == api.DATA_CORRUPT
    for k in (3, 4, 5, 6):
        assert api.bsc_decompress(ref.compress(T, k, 1)) == T.tobytes(), k


def test_st_decode_inverts_reference_encoder(ref):
    """inverse ST (row f4) against the reference's CPU encoder for k = 3..6, and against its decoder's verdict for
    k = 7, 8 via the numpy model of our GPU encoder (tests/pipeline_model.py)."""
    from pipeline_model import st_model
    rng = np.random.default_rng(4)
    cases = [t for _, t in _texts() if t.size <= 300_000]
    for n in (2, 3, 4, 5, 7, 8, 9, 16, 100, 1000):
        cases += [rng.integers(0, 256, n, dtype=np.uint8), rng.integers(0, 2, n, dtype=np.uint8), np.zeros(n, np.uint8),
                  (np.arange(n) % 3).astype(np.uint8)]
    for T in cases:
        for k in (3, 4, 5, 6):
            enc, idx = ref.st_encode(T, k)
            back, rc = api.bsc_st_decode(enc, k, idx)
            assert rc == 0 and np.array_equal(back, T), (T.size, k)
        if T.size <= 5000:
            for k in (7, 8):
                enc, idx = st_model(T, k)
                back, rc = api.bsc_st_decode(enc, k, idx)
                assert rc == 0 and np.array_equal(back, T), (T.size, k)
    assert api.bsc_st_decode(np.zeros(10, np.uint8), 2, 0)[1] == api.BAD_PARAMETER
    assert api.bsc_st_decode(np.zeros(10, np.uint8), 5, 10)[1] == api.BAD_PARAMETER


def test_bsc_decompress_reads_reference_blocks_with_lzp(ref):
    """LZP is not encoded here (row f3), but blocks the reference writes with LZP on — the CLI default -H15 -M128 —
    must still decode: literal runs, matches, escaped 0xF2 bytes, 1..4 LZP sub-blocks."""
    from libbsc_amd.synth import synth_text_v1
    rng = np.random.default_rng(8)
    base = synth_text_v1(12, 40_000)
    rep = np.concatenate([base, rng.integers(0, 256, 500, dtype=np.uint8), base, base[:20_000], np.full(3000, 0xF2, np.uint8), base])
    big = np.concatenate([synth_text_v1(13, 200_000)] * 3 + [rng.integers(0, 256, 1000, dtype=np.uint8)])      # >= 256 KiB: 2 LZP sub-blocks
    for T in (rep, big):
        for h, m in ((15, 128), (16, 32), (10, 4), (20, 255)):
            for sorter in (1, 5):
                blk = ref.compress(T, sorter, 1, lzp_hash=h, lzp_min=m)
                assert isinstance(blk, bytes)
                assert api.bsc_decompress(blk) == T.tobytes(), (T.size, h, m, sorter)


def test_bwt_decode_with_aux_indexes(ref):
    """The aux indexes (bwt.cpp:192-209) cut the inverse BWT into independent walks: same text with and without them, on
    one or several threads; inconsistent indexes are detected; a count that does not match n falls back to one walk."""
    from libbsc_amd.synth import synth_text_v1
    for n in (16, 17, 1000, 65536, 300_000, (1 << 20) + 17, 3 << 20):
        T = synth_text_v1(12, n)
        L, idx, aux = ref.bwt_encode(T)
        assert len(aux) == (n - 1) // (1 << ((n // 8).bit_length() - 1)) if n >= 16 else True
        for features in (1, 3):
            back, rc = api.bsc_bwt_decode(L, idx, aux, features=features)
            assert rc == 0 and np.array_equal(back, T), (n, features)
        back, rc = api.bsc_bwt_decode(L, idx)                       # no indexes: single walk
        assert rc == 0 and np.array_equal(back, T)
        if len(aux) >= 2:
            bad = list(aux); bad[1] = (bad[1] + 1) % n
            assert api.bsc_bwt_decode(L, idx, bad)[1] == api.DATA_CORRUPT
            back, rc = api.bsc_bwt_decode(L, idx, aux[:-1])          # wrong count: ignored
            assert rc == 0 and np.array_equal(back, T)


def test_decompress_survives_corrupt_streams(ref):
    """Blocks whose payload was damaged and whose checksums were then repaired reach the QLFC decoders, the inverse
    transforms and the LZP decoder with garbage: every one must come back as an error code (or, by luck, the right data),
    never as a crash or an overrun — all decoders here are bounded by the sizes the block header announces."""
    import struct
    from libbsc_amd.synth import synth_repeat_v1, synth_text_v1
    rng = np.random.default_rng(77)
    blocks = [(T, ref.compress(T, so, co, lzp_hash=lz[0], lzp_min=lz[1])) for T, so, co, lz in
              [(synth_text_v1(3, 200_000), 1, 1, (0, 0)), (synth_text_v1(4, 70_000), 1, 2, (0, 0)),
               (synth_text_v1(5, 300_000), 5, 3, (0, 0)), (synth_repeat_v1(6, 400_000, 7000), 1, 1, (15, 32)),
               (synth_text_v1(7, 2000), 1, 1, (0, 0))]]
    for it in range(150):
        T, blk = blocks[it % len(blocks)]
        b = bytearray(blk)
        for _ in range(int(rng.integers(1, 4))):
            mode = int(rng.integers(0, 4)); pos = int(rng.integers(0, len(b)))
            if mode == 0: b[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1: b[pos] = int(rng.integers(0, 256))
            elif mode == 2 and len(b) > 40: del b[pos]
            else: b.insert(pos, int(rng.integers(0, 256)))
        if rng.integers(0, 4):
            b[0:4] = struct.pack("<i", len(b))
        b[20:24] = struct.pack("<I", api.bsc_adler32(np.frombuffer(bytes(b[28:]), np.uint8)))
        b[24:28] = struct.pack("<I", api.bsc_adler32(np.frombuffer(bytes(b[:24]), np.uint8)))
        r = api.bsc_decompress(bytes(b))
        assert isinstance(r, int) and r < 0 or r == T.tobytes(), it


def test_decompress_rejects_frame_tables_that_leave_the_payload(ref):
    """A checksum-valid block whose sub-block table points past the payload (the reference has the same hole,
    coder.cpp:244-330): compressed sizes are checked against the payload length before anything is read, a truncated
    frame table and a 28-byte block with a non-zero mode are rejected too."""
    import struct
    from libbsc_amd.synth import synth_text_v1

    def seal(b):
        b[0:4] = struct.pack("<i", len(b))
        b[20:24] = struct.pack("<I", api.bsc_adler32(np.frombuffer(bytes(b[28:]), np.uint8)))
        b[24:28] = struct.pack("<I", api.bsc_adler32(np.frombuffer(bytes(b[:24]), np.uint8)))
        return bytes(b)

    T = synth_text_v1(5, 600_000)                       # >= 256 KiB: 2 sub-blocks
    blk = bytearray(ref.compress(T, 1, 1))
    assert blk[28] == 2 and api.bsc_decompress(bytes(blk)) == T.tobytes()
    for field, value in ((28 + 1 + 4, 16187462), (28 + 1 + 4, 0x7fffffff), (28 + 1 + 8 + 4, len(blk)), (28 + 1 + 4, len(blk) - 28 - 1)):
        b = bytearray(blk)
        b[field:field + 4] = struct.pack("<i", value)
        assert api.bsc_decompress(seal(b)) == api.DATA_CORRUPT, (field, value)
    # a 212-byte block announcing two sub-blocks, the first 16 MB long (the advisor's reproducer)
    small = bytearray(blk[:212])
    small[28 + 1 + 4:28 + 1 + 8] = struct.pack("<i", 16187462)
    small[211] = 0                                       # no aux indexes
    assert api.bsc_decompress(seal(small)) == api.DATA_CORRUPT
    # frame table cut short: nblocks = 8 but only a few bytes of payload
    tiny = bytearray(blk[:28 + 6]); tiny[28] = 8; tiny[-1] = 0
    assert api.bsc_decompress(seal(tiny)) == api.DATA_CORRUPT
    # header only, mode != 0
    hdr = bytearray(blk[:28])
    assert api.bsc_decompress(seal(hdr)) < 0
