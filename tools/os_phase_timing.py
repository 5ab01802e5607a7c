"""Phase timing of the single-read digit pass (debug build: SRC=radix_onesweep bash tools/build_variant.sh osph -DOS_PHASE_TIMING=1).
    BSC_LIB_OVERRIDE=libbsc_amd/lib/variants/libbsc_osph.so python tools/os_phase_timing.py
Runs one 64 MiB BWT (the library dumps the stamps of the last first-sort pass) and prints the mean time between the phase stamps of
threads 0 and 960 of every workgroup, in s_memtime ticks (~2 per ns) and as a fraction of the iteration."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=n + 4096)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
os.makedirs("gpurun_out", exist_ok=True)
if os.path.exists("gpurun_out/os_phase_timing.bin"): os.remove("gpurun_out/os_phase_timing.bin")
ctx.bwt_device(d, out, n, aux_rate=1 << 23)
a = np.fromfile("gpurun_out/os_phase_timing.bin", dtype=np.uint64).reshape(256, 40, 2, 16)
names = ["ticket + look-back loads issued", "rank next (waits its keys)", "barrier 1", "values requested, digit scan, publish", "look-back sums (+polls), ticket read",
         "barrier 3", "stage next keys, request keys after next, adj", "barrier 4", "key write-out", "barrier 5", "value staging", "barrier 6", "value write-out"]
for who, label in ((0, "thread 0 (wave 0: tickets)"), (1, "thread 960 (wave 15)")):
    st = a[:, :, who, :14].astype(np.int64)
    ok = (st[:, :, 13] > 0) & (st[:, :, 0] > 0)
    ok[:, :3] = False                               # skip pipeline fill
    tiles = st[ok]
    nxt = a[:, :, who, 14][ok] & 0xffffffff
    tiles = tiles[nxt != 0xffffffff]                # steady iterations only
    dt = np.diff(tiles, axis=1)
    tot = (tiles[:, 13] - tiles[:, 0]).mean()
    print(f"{label}: {tiles.shape[0]} iterations, mean {tot:.0f} ticks")
    for i, nm in enumerate(names):
        print(f"  {nm:48s} {dt[:, i].mean():8.0f}  {100 * dt[:, i].mean() / tot:5.1f} %   (p90 {np.percentile(dt[:, i], 90):.0f})")
# tiles per workgroup and XCD spread of batches
cur = (a[:, :, 0, 14] >> 32).astype(np.int64)
cnt = ((cur != 0xffffffff) & (a[:, :, 0, 13] > 0)).sum(axis=1)
print("tiles per workgroup: min %d max %d" % (cnt.min(), cnt.max()))
