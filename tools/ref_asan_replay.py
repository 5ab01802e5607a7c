"""Replay the case stream of tools/fuzz_gpu.py (same seed -> same sizes, contents and parameters: the generator only depends on
the draws, not on any result) through the REFERENCE alone, compiled with AddressSanitizer (make -C oracle ref_asan), to tell an
abort of the parity sweep that is the reference's own doing from one of ours.  CPU only.
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/ref_asan_replay.py [cases] [seed] [max block bytes]
Every reference call fuzz_gpu.py makes is made with the buffer sizes oracle/refbind.py uses there."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd.synth import synth_repeat_v1, synth_text_v1
from oracle import refbind

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 303
MAXN = int(sys.argv[3]) if len(sys.argv) > 3 else (12 << 20)
rng = np.random.default_rng(seed)
ref = refbind.Ref(os.path.join(os.path.dirname(refbind.REF_SO), "libbsc_ref_asan.so"))
EDGES = [1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 21, 1 << 22, (1 << 22) + (1 << 21), 1 << 23, 4096 * 1024, 8192 * 512, 8192 * 1024, 16384 * 256]

def draw_n():
    k = rng.integers(0, 4)
    if k == 0: return int(rng.integers(1, 5000))
    if k == 1: return int(max(1, rng.choice(EDGES) + rng.integers(-3, 4)))
    if k == 2: return int(rng.integers(1, 1 << 20))
    return int(rng.integers(1 << 20, MAXN))

def draw_data(n):
    k = rng.integers(0, 7)
    if k == 0: return synth_text_v1(int(rng.integers(1, 1 << 30)), n)
    if k == 1: return rng.integers(0, 256, n, dtype=np.uint8)
    if k == 2: return rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8)
    if k == 3: return np.tile(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8), n // 1 + 1)[:n].copy()
    if k == 4: return synth_repeat_v1(int(rng.integers(1, 1 << 20)), n, int(rng.integers(2, max(3, n // 3 + 3)))) if n >= 8 else np.zeros(n, np.uint8)
    if k == 5:
        x = np.zeros(n, np.uint8); m = max(1, n // 50); x[rng.integers(0, n, m)] = rng.integers(1, 256, m); return x
    return np.concatenate([synth_text_v1(7, n // 2), rng.integers(0, 256, n - n // 2, dtype=np.uint8)])

t0 = time.time()
for case in range(1, ncases + 1):
    n = draw_n(); T = np.ascontiguousarray(draw_data(n))
    what = int(rng.integers(0, 3))
    if what == 0:
        print("case", case, n, "bwt", flush=True)
        ref.bwt_encode(T, aux=(n >= 16))
    elif what == 1:
        k = int(rng.integers(3, 9))
        print("case", case, n, "st", k, flush=True)
        if k <= 6: ref.st_encode(T, k)
        # k = 7, 8: fuzz_gpu.py hands OUR transform to ref.st_decode; here the reference inverts a transform of its own order-6 output
        # shape instead (the decoder's buffer handling is what is being watched)
        else:
            L, idx = ref.st_encode(T, 6); ref.st_decode(L, 6, idx)
    else:
        sorter = int(rng.choice([1, 1, 1, 3, 4, 5, 6, 7, 8])); coder = int(rng.integers(1, 4)); feat = int(rng.choice([0, 1, 3]))
        lz = (0, 0) if rng.integers(0, 3) else (int(rng.integers(10, 20)), int(rng.choice([4, 6, 8, 12, 16, 32, 128])))
        print("case", case, n, "compress", sorter, coder, feat, lz, flush=True)
        blk = ref.compress(T, min(sorter, 6), coder, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
        if not isinstance(blk, int):
            assert ref.decompress(blk, features=feat) == T.tobytes()
print(f"ref replay: {ncases} cases of seed {seed} in {time.time() - t0:.0f} s, no sanitizer report")
