"""Per-pass scatter times of the 64 MiB BWT's first sort: python tools/per_pass.py  (BSC_RS_WC=0/1 in the environment)."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=n + 4096)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
ctx.bwt_device(d, out, n, aux_rate=1 << 23)
acc = []
for rep in range(5):
    ctx.profile(True); ctx.profile_reset(); ctx.bwt_device(d, out, n, aux_rate=1 << 23); sl = ctx.scatter_launches(); ctx.profile(False)
    acc.append([m for m, rec in sl if rec == n])
a = np.median(np.array(acc), axis=0)
print("BSC_RS_ONESWEEP=%s per pass (ms): %s  mean %.3f" % (os.environ.get("BSC_RS_ONESWEEP", "default"), " ".join(f"{x:.3f}" for x in a), a.mean()))
