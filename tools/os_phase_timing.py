"""Phase timing of the single-read digit pass (debug build: SRC=radix_onesweep bash tools/build_variant.sh os_ph -DOS_PHASE_TIMING=1).
    BSC_LIB_OVERRIDE=libbsc_amd/lib/variants/libbsc_os_ph.so python tools/os_phase_timing.py
Runs one 64 MiB BWT (the library dumps the stamps of the last first-sort pass) and prints the mean time between the phase stamps of
thread 0 (a streaming wave) and of the scout wave's lane 0 of every workgroup, in s_memtime ticks (~2 per ns)."""
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
stream = [(0, 1, "key write-out (tile i)"), (1, 2, "barrier 5"), (2, 3, "value staging"), (3, 4, "barrier 6"), (4, 5, "value write-out"),
          (5, 6, "rank tile i+2 (waits its keys)"), (6, 7, "barrier 1"), (7, 8, "values of i+1 requested, digit scan (barrier 2 inside)"), (8, 9, "barrier 3"),
          (9, 10, "stage keys of i+2, request keys of i+3"), (10, 11, "barrier 4 (scout: offsets of i+1)")]
scout = [(0, 1, "ticket + look-back loads of tile i+1 issued"), (1, 2, "barriers 5, 6, 1, 2, ticket read"), (2, 3, "barrier 3"),
         (3, 7, "wait for the rows, which are missing"), (7, 8, "re-requests (all missing rows at once)"),
         (8, 4, "sums, offsets"), (4, 11, "barrier 4")]
for who, label, phases in ((0, "thread 0 (streaming wave 0)", stream), (1, "scout wave", scout)):
    st = a[:, :, who, :12].astype(np.int64)
    ok = (st[:, :, 11] > 0) & (st[:, :, 0] > 0)
    ok[:, :3] = False                               # skip pipeline fill
    nxt = a[:, :, who, 14] & 0xffffffff
    cur = a[:, :, who, 14] >> 32
    ok &= (nxt != 0xffffffff) & (cur != 0xffffffff)  # steady iterations only
    tiles = st[ok]
    tot = (tiles[:, 11] - tiles[:, 0]).mean()
    print(f"{label}: {tiles.shape[0]} iterations, mean {tot:.0f} ticks")
    for i0, i1, nm in phases:
        dt = tiles[:, i1] - tiles[:, i0]
        print(f"  {nm:64s} {dt.mean():8.0f}  {100 * dt.mean() / tot:5.1f} %   (p90 {np.percentile(dt, 90):.0f})")
np_ = a[:, :, 1, 15].astype(np.int64)
print("scout: slow-path row reads (cumulative per workgroup over its first 40 tiles, all lanes counted once): mean %.1f max %d" % (np_.max(axis=1).mean(), np_.max()))
cnt = ((a[:, :, 0, 14] >> 32 != 0xffffffff) & (a[:, :, 0, 11] > 0)).sum(axis=1)
print("tiles per workgroup: min %d max %d" % (cnt.min(), cnt.max()))
