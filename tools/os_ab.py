"""A/B timing of digit-pass variants on the BWT's own first-sort keys (64 Mi records: 12 five-bit character codes per key, value =
suffix index), sort only — variants that give wrong results on purpose (ablation builds) can be timed too.
    python tools/os_ab.py libA.so[:ENV=VAL,...] libB.so ...        (alternating, 2 rounds; per-pass medians of the scatter kernel)"""
import os, subprocess, sys, json
CHILD = r'''
import sys, json, os; sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
used = np.unique(T); lut = np.zeros(256, np.uint64); lut[used] = np.arange(used.size, dtype=np.uint64)
codes = np.concatenate([lut[T], np.zeros(16, np.uint64)])
keys = np.zeros(n, np.uint64)
for t in range(12):
    keys |= codes[t:t + n] << np.uint64(64 - 5 * (t + 1))
ctx = GpuContext(0, max_n=n + 4096)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.arange(n, dtype=torch.int32).cuda()
k2 = torch.empty_like(dk); v2 = torch.empty_like(dv)
acc = []; hist = []
for rep in range(6):
    kk = dk.clone(); vv = dv.clone()
    ctx.profile(True); ctx.profile_reset()
    ctx.radix_sort(kk, k2, vv, v2, n, 4, 64)
    st = ctx.profile_get(); sl = ctx.scatter_launches(); ctx.profile(False)
    if rep: acc.append([m for m, rec in sl if rec == n]); hist.append(st["radix_hist"]["ms"] + st.get("radix_scan", {"ms": 0})["ms"])
a = np.median(np.array(acc), axis=0)
print("RESULT " + json.dumps({"passes": [round(float(x), 4) for x in a], "mean": float(a.mean()), "hist_scan_ms_per_sort": float(np.median(hist)), "sort_ms": float(a.sum() + np.median(hist))}))
'''
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        path, _, extra = l.partition(":")
        env = dict(os.environ)
        if path != "default": env["BSC_LIB_OVERRIDE"] = os.path.abspath(path)
        env.update(kv.split("=", 1) for kv in extra.split(",") if kv)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print("FAILED", l, r.stdout[-500:], r.stderr[-1500:]); continue
        res[l].append(json.loads(line[0][7:]))
for l in libs:
    if not res[l]: continue
    best = min(res[l], key=lambda r: r["sort_ms"])
    print(f"{l[-48:]:48s} scatter/pass mean {best['mean']:.4f} ms ({24 * (64 << 20) / 1e6 / best['mean']:.0f} GB/s)  hist+scan/sort {best['hist_scan_ms_per_sort']:.3f}  sort {best['sort_ms']:.3f} ms  passes {best['passes']}")
