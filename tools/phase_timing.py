"""Phase timing of the scatter kernels (debug build: bash tools/build_variant.sh pht -DRS_PHASE_TIMING=1).
    BSC_LIB_OVERRIDE=libbsc_amd/lib/variants/libbsc_pht.so [BSC_RS_WC=1] python tools/phase_timing.py
Runs one 64 MiB BWT (the last first-sort pass is dumped by the library) and prints the mean time between the phase stamps
of thread 0 of every workgroup, in s_memtime ticks and as a fraction of the tile."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=n + 4096)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
os.makedirs("gpurun_out", exist_ok=True)
if os.path.exists("gpurun_out/phase_timing.bin"): os.remove("gpurun_out/phase_timing.bin")
ctx.bwt_device(d, out, n, aux_rate=1 << 23)
a = np.fromfile("gpurun_out/phase_timing.bin", dtype=np.uint64).reshape(256, 32, 16).astype(np.int64)
names = ["load issue+zero+bar", "rank (waits loads)", "barrier", "digit scan+bar", "staging wr (+flush)", "barrier", "key write-out", "bar+svals+bar", "value write-out", "barrier"]
tiles = a[:, 2:30, :11]                      # skip the first / last tiles of every workgroup
dt = np.diff(tiles, axis=2)
tot = (tiles[:, :, 10] - tiles[:, :, 0]).mean()
gap = (a[:, 3:30, 0] - a[:, 2:29, 10]).mean()
print(f"mode BSC_RS_WC={os.environ.get('BSC_RS_WC', '0')}: tile {tot:.0f} ticks (+ {gap:.0f} between tiles)")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {dt[:, :, i].mean():8.0f}  {100 * dt[:, :, i].mean() / tot:5.1f} %")
