"""One large (default 300 MB, not a power of two) block: GPU BWT + full compress vs the compiled reference."""
import sys, time, hashlib
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
from oracle.refbind import Ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_007
T = api.synth_text_v1(77, n)
ref = Ref()
t = time.time(); want = ref.compress(T, 1, 1); t_ref = time.time() - t
ctx = GpuContext(0, max_n=n + 4096)
print("arena GiB", ctx.arena_bytes / 2**30)
d = torch.from_numpy(T).cuda()
t = time.time(); got = ctx.compress_device(d, n, 1, 1).tobytes(); t_gpu = time.time() - t
print(f"n={n}: ours {len(got)} B in {t_gpu*1e3:.0f} ms, reference {len(want)} B in {t_ref:.1f} s, identical={got == want}, stages={ctx.last_stage_ms()}")
got5 = ctx.compress_device(d, n, 5, 1).tobytes(); want5 = ref.compress(T, 5, 1)
print("ST5 identical:", got5 == want5, len(got5))
assert got == want and got5 == want5
