"""A/B timing of engine variants on the real workload (one box, alternating): 64 MiB BWT and 128 MiB ST5.
    python tools/ab_bwt.py libA.so libB.so ..."""
import os, subprocess, sys, json
CHILD = r'''
import sys, json, time; sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=(128 << 20) + 4096)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
r = 1 << ((n // 8).bit_length() - 1)
ctx.bwt_device(d, out, n, aux_rate=r)
ts = []
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); idx, I = ctx.bwt_device(d, out, n, aux_rate=r); ts.append(time.perf_counter() - t0)
ctx.profile(True); ctx.profile_reset(); ctx.bwt_device(d, out, n, aux_rate=r); st = ctx.profile_get(); sl = ctx.scatter_launches(); ctx.profile(False)
full = [m for m, rec in sl if rec == n]; rest = [m for m, rec in sl if rec != n]
res = {"bwt_ms": min(ts) * 1e3, "idx": int(idx), "scatter_ms": st["radix_scatter"]["ms"], "hist_ms": st["radix_hist"]["ms"],
       "first_sort_pass_ms": float(np.mean(full)), "round_passes_ms": float(np.sum(rest))}
T2 = api.synth_text_v1(3, 128 << 20); d2 = torch.from_numpy(T2).cuda(); o2 = torch.empty_like(d2)
ctx.st_encode_device(d2, o2, 128 << 20, 5)
ts = []
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); i5 = ctx.st_encode_device(d2, o2, 128 << 20, 5); ts.append(time.perf_counter() - t0)
res["st5_128m_ms"] = min(ts) * 1e3; res["st5_idx"] = int(i5)
print("RESULT " + json.dumps(res))
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
    best = {k: min(r[k] for r in res[l]) for k in res[l][0]}
    print(os.path.basename(l).ljust(40), "  ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}" for k, v in best.items()))
