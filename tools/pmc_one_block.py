"""One 64 MiB bench block through bscgpu_compress_device (Adler-32, BWT, QLFC front end, device coder) for the rocprofv3 PMC passes:
every kernel of the block's GPU stage exactly once per dispatch, no overlap with another context."""
import sys
sys.path.insert(0, '.')
import torch
from libbsc_amd import GpuContext, api
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=n + 4096)
d = torch.from_numpy(T).cuda()
import os
for _ in range(int(os.environ.get("ONE_BLOCK_REPS", "1"))):
    blk = ctx.compress_device(d, n, 1, 1)
print("compressed", blk.size)
