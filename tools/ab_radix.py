"""A/B timing of radix-engine variants on ONE box (box-to-box spread on the pool is ~20 %, so variants are only comparable
inside one gpurun call): python tools/ab_radix.py libA.so libB.so[:ENV=VAL,...] ...  — alternates the variants, 2 rounds each."""
import os, subprocess, sys, json
import numpy as np
CHILD = r'''
import sys, json; sys.path.insert(0, '.')
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
out = {}
for name, keys in (("uniform", uni), ("skewed", skew)):
    for pairs in (True, False):
        ms = []
        for rep in range(3):
            kk = keys.clone()
            ctx.profile(True); ctx.profile_reset()
            ctx.radix_sort(kk, k2, vals.clone() if pairs else None, v2 if pairs else None, n, 0, 64)
            ms += [m for m, _ in ctx.scatter_launches()]
            ctx.profile(False)
        out[name + (" pairs" if pairs else " keys")] = float(np.median(ms))
print("RESULT " + json.dumps(out))
'''
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        path, _, extra = l.partition(":")
        env = dict(os.environ, BSC_LIB_OVERRIDE=os.path.abspath(path))
        env.update(kv.split("=", 1) for kv in extra.split(",") if kv)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print("FAILED", l, r.stdout[-500:], r.stderr[-1500:]); continue
        res[l].append(json.loads(line[0][7:]))
for l in libs:
    if not res[l]: continue
    keys = res[l][0].keys()
    print(os.path.basename(l).ljust(40), "  ".join(f"{k}: {min(r[k] for r in res[l]):.3f} ms ({(24 if 'pairs' in k else 16) * (64 << 20) / 1e6 / min(r[k] for r in res[l]):.0f} GB/s)" for k in keys))
