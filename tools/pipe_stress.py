"""Stress of the pipelined entry point on host input (bscgpu_pipe_submit_host: LZP, every sorter / coder, several blocks in flight,
the process-wide coder pool with all three task shapes), every block compared with the reference.  Needs no torch, so it also runs on
the sanitizer build (python tools/asan_run.py python tools/pipe_stress.py ...).
    python tools/pipe_stress.py [seconds] [seed] [depth] [max block bytes]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import GpuContext, api
from libbsc_amd.gpu import coder_pool_stats
from oracle.refbind import Ref

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
MAXN = int(sys.argv[4]) if len(sys.argv) > 4 else (24 << 20)
ref = Ref()
ctx = GpuContext(0, max_n=MAXN + 4096)
pipe = ctx.pipe(depth)

def draw():
    k = int(rng.integers(0, 4))
    n = int(rng.integers(1, 1 << 16)) if k == 0 else int(rng.integers(1 << 20, 4 << 20)) if k == 1 else int(rng.integers(16 << 20, MAXN)) if k == 2 else int(rng.integers(4 << 20, 17 << 20))
    kind = int(rng.integers(0, 5))
    if kind <= 1: T = api.synth_text_v1(int(rng.integers(1, 1 << 30)), n)
    elif kind == 2: T = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 3: T = np.concatenate([api.synth_text_v1(9, n // 2), rng.integers(0, 256, n - n // 2, dtype=np.uint8)])
    else:
        T = np.zeros(n, np.uint8); m = max(1, n // 40); T[rng.integers(0, n, m)] = rng.integers(1, 256, m)
    sorter = int(rng.choice([1, 1, 1, 1, 5, 6, 3])); coder = int(rng.choice([1, 1, 1, 2, 3])); feat = int(rng.choice([0, 1, 3]))
    if rng.integers(0, 5) == 0: feat |= 0x10000              # BSCGPU_FEATURE_LOW_LATENCY (include/bscgpu.h)
    lz = (0, 0) if rng.integers(0, 3) else (int(rng.integers(10, 20)), int(rng.choice([4, 16, 32, 128])))
    return T, sorter, coder, feat, lz

t0 = time.time(); cases = 0; bad = 0; inflight = []
def retire():
    global bad
    tk, T, sorter, coder, feat, lz = inflight.pop(0)
    got = pipe.wait(tk).tobytes()
    want = ref.compress(T, sorter, coder, lzp_hash=lz[0], lzp_min=lz[1], features=feat & 3)
    if got != want:
        bad += 1; print("MISMATCH", T.size, sorter, coder, feat, lz, flush=True)
while time.time() - t0 < budget:
    T, sorter, coder, feat, lz = draw()
    print("case", cases, T.size, sorter, coder, hex(feat), lz, flush=True)
    if len(inflight) == depth: retire()
    inflight.append((pipe.submit_host(T, sorter, coder, lz[0], lz[1], feat), T, sorter, coder, feat, lz))
    cases += 1
while inflight: retire()
print(f"pipe stress: {cases} blocks in {time.time() - t0:.0f} s, {bad} mismatches, task shapes {coder_pool_stats()}")
pipe.close(); ctx.close()
sys.exit(1 if bad else 0)
