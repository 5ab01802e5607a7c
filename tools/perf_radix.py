"""Radix engine timing on synthetic keys: text-like skew and uniform digits."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext
n = 64 << 20
ctx = GpuContext(0, max_n=1 << 20)
rng = np.random.default_rng(0)
uni = torch.from_numpy(rng.integers(0, 2**63, n, dtype=np.int64)).cuda()
kb = rng.integers(0, 7, (n, 8), dtype=np.uint8) * 37
skew = torch.from_numpy(kb.view(np.int64).reshape(-1).copy()).cuda()
vals = torch.arange(n, dtype=torch.int32).cuda()
k2 = torch.empty_like(uni); v2 = torch.empty_like(vals)
for name, keys in (("uniform", uni), ("skewed", skew)):
    for pairs in (True, False):
        kk = keys.clone()
        ctx.profile(True); ctx.profile_reset()
        ctx.radix_sort(kk, k2, vals.clone() if pairs else None, v2 if pairs else None, n, 0, 64)
        st = ctx.profile_get(); sl = ctx.scatter_launches()
        ms = np.array([m for m, _ in sl])
        b = 24 if pairs else 16
        print(f"{name:8s} {'pairs' if pairs else 'keys '}: scatter avg {ms.mean():.3f} ms (min {ms.min():.3f}) -> {b*n/1e6/ms.mean():.0f} GB/s; hist avg {st['radix_hist']['ms']/8:.3f} ms -> {8*n/1e6/(st['radix_hist']['ms']/8):.0f} GB/s")
        ctx.profile(False)
