"""Stress of the synchronous drop-in path (bsc_compress / bsc_decompress on host pointers, device model, one scalar coder per task):
random sizes around the sub-block and context-resize boundaries, output compared with the reference every time.
    python tools/sync_stress.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
from oracle.refbind import Ref
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref = Ref()
t0 = time.time(); cases = 0; bad = 0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(1 << 20, 3 << 20), rng.integers(3 << 20, 9 << 20), rng.integers(9 << 20, 26 << 20)]))
    T = api.synth_text_v1(int(rng.integers(1, 1 << 30)), n)
    feat = int(rng.choice([0, 1, 3]))
    print("case", cases, n, feat, flush=True)
    got = api.bsc_compress(T, 1, 1, features=feat)
    want = ref.compress(T, 1, 1, features=feat)
    ok = got == want and api.bsc_decompress(got, features=feat) == T.tobytes()
    cases += 1; bad += (not ok)
    if not ok: print("MISMATCH", n, feat, flush=True)
print(f"sync stress: {cases} cases in {time.time() - t0:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
