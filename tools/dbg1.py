import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from libbsc_amd import GpuContext
from oracle.refbind import Ref
from test_gpu_device import _corpus, _aux_rate
ref = Ref(); ctx = GpuContext(0, max_n=(1<<20)+4096)
rng = np.random.default_rng(11)
for name, T in _corpus(rng):
    n = T.size
    want_L, want_idx, _ = ref.bwt_encode(T, aux=False)
    try:
        L, idx, _ = ctx.bwt(T)
        ok = np.array_equal(L, want_L) and idx == want_idx
        if not ok:
            print("MISMATCH", name, n, idx, want_idx, int((L != want_L).sum()), np.flatnonzero(L != want_L)[:8])
    except Exception as e:
        print("ERROR", name, n, e)
print("done")
