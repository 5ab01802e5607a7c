"""Per-kernel-class times of the device coder on the 64 MiB bench block (GPU box): python tools/devcoder_time.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import GpuContext, api
from oracle.refbind import Ref
L = Ref().bwt_encode(api.synth_text_v1(2, 64 << 20))[0]
ctx = GpuContext(0, max_n=L.size + 4096)
ctx.qlfc_static_pstream(L)
ctx.profile(True)
for rep in range(2):
    ctx.profile_reset()
    t0 = time.time()
    ctx.qlfc_static_pstream(L)
    print("call %.1f ms" % (1e3 * (time.time() - t0)), {k: round(v["ms"], 2) for k, v in ctx.profile_get().items() if v["launches"]}, flush=True)
