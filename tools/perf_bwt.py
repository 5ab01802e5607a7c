"""Quick device-layer timing: 64 MiB BWT (and ST) with per-kernel HIP-event profile."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext
from libbsc_amd.synth import synth_text_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time(); T = synth_text_v1(seed, n); print("synth %.1fs" % (time.time() - t0))
ctx = GpuContext(0, max_n=n + 4096)
print("arena GiB", ctx.arena_bytes / 2**30)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
r = 1 << ((n // 8).bit_length() - 1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    idx, I = ctx.bwt_device(d, out, n, aux_rate=r)
    dt = time.time() - t0
    print(f"bwt iter {it}: {dt*1e3:.1f} ms  {n/1e6/dt:.0f} MB/s  idx={idx} rounds={ctx.last_stage_ms()[5]}")
ctx.profile(True); ctx.profile_reset()
t0 = time.time(); idx, I = ctx.bwt_device(d, out, n, aux_rate=r); dt = time.time() - t0
print(f"profiled: {dt*1e3:.1f} ms")
st = ctx.profile_get()
for k, v in st.items():
    if v['launches']:
        print(f"  {k:14s} {v['ms']:9.3f} ms  {v['launches']:4d} launches  {v['bytes']/1e9:8.2f} GB  -> {v['bytes']/1e6/max(v['ms'],1e-9):8.1f} GB/s")
sl = ctx.scatter_launches()
full = [(ms, rec) for ms, rec in sl if rec == n]
if full:
    ms = np.array([m for m, _ in full]); print(f"  scatter n={n}: {len(full)} launches avg {ms.mean():.3f} ms min {ms.min():.3f} max {ms.max():.3f} -> {24*n/1e6/ms.mean():.0f} GB/s ({24*n/1e6/ms.mean()/8000*100:.1f}% of 8 TB/s)")
by = {}
for ms, rec in sl: by.setdefault(rec, []).append(ms)
for rec, v in by.items(): print(f"  scatter records={rec}: {len(v)} launches avg {np.mean(v):.3f} ms -> {24*rec/1e6/np.mean(v):.0f} GB/s")
ctx.profile(False)
for k in (5, 6):
    ctx.profile(True); ctx.profile_reset()
    t0 = time.time(); i2 = ctx.st_encode_device(d, out, n, k); dt = time.time() - t0
    st = ctx.profile_get(); sl = ctx.scatter_launches()
    ms = np.array([m for m, _ in sl])
    print(f"st{k}: {dt*1e3:.1f} ms {n/1e6/dt:.0f} MB/s idx={i2}; scatter avg {ms.mean():.3f} ms -> {16*n/1e6/ms.mean():.0f} GB/s; hist {st['radix_hist']['ms']:.3f} ms total")
    ctx.profile(False)
