"""Randomised parity sweep on the GPU box: block sorters (index, aux indexes, bytes) and whole blocks against the compiled
reference, sizes and contents drawn at random with emphasis on the engine's shape thresholds.
    python tools/fuzz_gpu.py [seconds] [seed] [max block bytes]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import GpuContext, api
from libbsc_amd.synth import synth_repeat_v1
from oracle.refbind import Ref

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ref = Ref()
MAXN = int(sys.argv[3]) if len(sys.argv) > 3 else (12 << 20)
VERBOSE = bool(__import__("os").environ.get("FUZZ_VERBOSE"))       # one line per case BEFORE it runs (a crash then names its case)
def say(*a):
    if VERBOSE: print("case", *a, flush=True)
EDGES = [1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 21, 1 << 22, (1 << 22) + (1 << 21), 1 << 23, 4096 * 1024, 8192 * 512, 8192 * 1024, 16384 * 256]

def draw_n():
    k = rng.integers(0, 4)
    if k == 0: return int(rng.integers(1, 5000))
    if k == 1: return int(max(1, rng.choice(EDGES) + rng.integers(-3, 4)))
    if k == 2: return int(rng.integers(1, 1 << 20))
    return int(rng.integers(1 << 20, MAXN))

def draw_data(n):
    k = rng.integers(0, 7)
    if k == 0: return api.synth_text_v1(int(rng.integers(1, 1 << 30)), n)
    if k == 1: return rng.integers(0, 256, n, dtype=np.uint8)
    if k == 2: return rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8)
    if k == 3: return np.tile(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8), n // 1 + 1)[:n].copy()
    if k == 4: return synth_repeat_v1(int(rng.integers(1, 1 << 20)), n, int(rng.integers(2, max(3, n // 3 + 3)))) if n >= 8 else np.zeros(n, np.uint8)
    if k == 5:
        x = np.zeros(n, np.uint8); m = max(1, n // 50); x[rng.integers(0, n, m)] = rng.integers(1, 256, m); return x
    return np.concatenate([api.synth_text_v1(7, n // 2), rng.integers(0, 256, n - n // 2, dtype=np.uint8)])

t0 = time.time(); cases = 0; fails = 0
while time.time() - t0 < budget:
    n = draw_n(); T = np.ascontiguousarray(draw_data(n)); cases += 1
    what = int(rng.integers(0, 3))
    try:
        if what == 0:
            say(cases, n, "bwt")
            L, idx, aux = api.bsc_bwt_encode(T, aux=(n >= 16))
            wL, widx, waux = ref.bwt_encode(T, aux=(n >= 16))
            ok = np.array_equal(L, wL) and idx == widx and list(aux) == list(waux)
        elif what == 1:
            k = int(rng.integers(3, 9))
            say(cases, n, "st", k)
            L, idx = api.bsc_st_encode(T, k)
            if k <= 6:
                wL, widx = ref.st_encode(T, k)
                ok = np.array_equal(L, wL) and idx == widx
            else:       # the reference's CPU build encodes ST3..6 only (ST7/8 are CUDA-only there); its decoder takes all orders
                back, rc = ref.st_decode(L, k, idx)
                ok = rc == 0 and np.array_equal(back, T)
            what = (what, k)
        else:
            sorter = int(rng.choice([1, 1, 1, 3, 4, 5, 6, 7, 8])); coder = int(rng.integers(1, 4)); feat = int(rng.choice([0, 1, 3]))
            lz = (0, 0) if rng.integers(0, 3) else (int(rng.integers(10, 20)), int(rng.choice([4, 6, 8, 12, 16, 32, 128])))
            say(cases, n, "compress", sorter, coder, feat, lz)
            got = api.bsc_compress(T, sorter, coder, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
            if sorter <= 6:
                want = ref.compress(T, sorter, coder, lzp_hash=lz[0], lzp_min=lz[1], features=feat)
                ok = got == want and (isinstance(got, int) or api.bsc_decompress(got, features=feat) == T.tobytes())
            else:
                ok = (not isinstance(got, int)) and ref.decompress(got, features=feat) == T.tobytes() and api.bsc_decompress(got, features=feat) == T.tobytes()
            what = (what, sorter, coder, feat, lz)
    except Exception as e:
        ok = False; what = (what, repr(e))
    if not ok:
        fails += 1
        print("MISMATCH", n, what, flush=True)
        np.save(f"gpurun_out/fuzz_fail_{seed}_{cases}.npy", T) if n < (1 << 22) else None
print(f"fuzz: {cases} cases in {time.time()-t0:.0f} s, {fails} mismatches (seed {seed})")
sys.exit(1 if fails else 0)
