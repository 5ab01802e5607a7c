import sys
sys.path.insert(0, '.')
import torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=n + 4096)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
idx, _ = ctx.bwt_device(d, out, n, aux_rate=1 << 23)
print("idx", idx)
