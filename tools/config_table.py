"""BASELINE.json configs 1-5 on one box: this library against the compiled reference (oracle/_ref) on the box's host cores.
Bit-exactness is asserted for every line.  python tools/config_table.py > gpurun_out/config_table.txt"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
from oracle.refbind import Ref

def eff_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": n = min(n, max(1, int(int(q) / int(p))))
    except Exception: pass
    return n
ncpu = eff_cpus()
os.environ["BSC_REF_THREADS"] = str(ncpu)
ref = Ref()
def best(f, reps=3):
    f(); b = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t)
    return b, r
print(f"host: {os.cpu_count()} logical CPUs visible, {ncpu} usable (cgroup quota); GPU: {torch.cuda.get_device_name(0)}")
MB = 1e6
# config 1: 1 MiB block, BWT + QLFC static
T1 = api.synth_text_v1(1, 1 << 20)
tr, want = best(lambda: ref.compress(T1, 1, 1))
to, got = best(lambda: api.bsc_compress(T1, 1, 1))
assert got == want
print(f"config 1  1 MiB block, bsc_compress(BWT, -e1), host pointers     ours {to*1e3:8.2f} ms ({T1.size/MB/to:7.1f} MB/s)   reference {tr*1e3:8.2f} ms ({T1.size/MB/tr:7.1f} MB/s)  identical")
# config 2: 64 MiB forward BWT only
n = 64 << 20
T = api.synth_text_v1(2, n)
ctx = GpuContext(0, max_n=(128 << 20) + 4096)
d = torch.from_numpy(T).cuda(); out = torch.empty_like(d)
r = 1 << ((n // 8).bit_length() - 1)
to, (idx, I) = best(lambda: ctx.bwt_device(d, out, n, aux_rate=r), 5)
tr, (Lr, idxr, auxr) = best(lambda: ref.bwt_encode(T), 2)
assert idx == idxr and np.array_equal(out.cpu().numpy(), Lr)
print(f"config 2  64 MiB forward BWT only (device resident)              ours {to*1e3:8.2f} ms ({n/MB/to:7.0f} MB/s)   reference libsais, {ncpu} threads {tr*1e3:8.1f} ms ({n/MB/tr:7.1f} MB/s)  identical")
# config 3: 64 MiB BWT + QLFC, single synchronous block and pipelined
to, got = best(lambda: ctx.compress_device(d, n, 1, 1).tobytes(), 3)
tr, want = best(lambda: ref.compress(T, 1, 1), 2)
assert got == want
pipe = ctx.pipe(4, reuse_outputs=True); tick = []
def run(k):
    for _ in range(k):
        tick.append(pipe.submit(d, n, 1, 1, 3))
        if len(tick) >= 4: pipe.wait(tick.pop(0))
    while tick: pipe.wait(tick.pop(0))
run(4); t = time.perf_counter(); run(24); tp = (time.perf_counter() - t) / 24; pipe.close()
print(f"config 3  64 MiB BWT + QLFC static (device resident)             ours {to*1e3:8.2f} ms one block, {tp*1e3:6.2f} ms/block pipelined ({n/MB/tp:7.0f} MB/s)   reference one call {tr*1e3:8.1f} ms ({n/MB/tr:6.1f} MB/s)  identical")
# config 5: ST5 / ST6 on 128 MiB
n2 = 128 << 20
T2 = api.synth_text_v1(3, n2); d2 = torch.from_numpy(T2).cuda(); o2 = torch.empty_like(d2)
for k in (5, 6):
    to, i5 = best(lambda: ctx.st_encode_device(d2, o2, n2, k), 3)
    tr, (Lr, ir) = best(lambda: ref.st_encode(T2, k), 1)
    assert i5 == ir and np.array_equal(o2.cpu().numpy(), Lr)
    tw, got = best(lambda: ctx.compress_device(d2, n2, k, 1).tobytes(), 1)
    print(f"config 5  128 MiB ST{k} sort transform only                          ours {to*1e3:8.2f} ms ({n2/MB/to:7.0f} MB/s)   reference {tr*1e3:8.1f} ms ({n2/MB/tr:7.1f} MB/s)  identical;  ST{k} + QLFC static one block {tw*1e3:7.1f} ms ({n2/MB/tw:6.0f} MB/s)")
# the reference CLI's default has LZP on (-H15 -M128, bsc.cpp:73-75): host input, LZP on host threads, then the same GPU stage (device model since round 4)
from libbsc_amd.synth import synth_repeat_v1
for name, TT in (("synth-text (LZP does not pay: dropped)", T), ("repeats (LZP pays)", synth_repeat_v1(9, n, 40_000, 48))):
    want = ref.compress(TT, 1, 1, lzp_hash=15, lzp_min=128)
    pipe = ctx.pipe(4, reuse_outputs=True); tick = []; last = [None]
    def runl(k):
        for _ in range(k):
            tick.append(pipe.submit_host(TT, 1, 1, 15, 128, 3))
            if len(tick) >= 4: last[0] = pipe.wait(tick.pop(0))
        while tick: last[0] = pipe.wait(tick.pop(0))
    runl(4); t = time.perf_counter(); runl(12); tp = (time.perf_counter() - t) / 12
    assert last[0].tobytes() == want
    pipe.close()
    print(f"config 3l 64 MiB BWT + QLFC static, LZP -H15 -M128, host input, {name:40s} ours {tp*1e3:6.2f} ms/block pipelined ({n/MB/tp:7.0f} MB/s)  identical")
# the other two coders (BASELINE's configs all say -e1): 64 MiB blocks pipelined through one context; -e0's model runs on the GPU since round 4, -e2's on host threads
for coder, name in ((2, "QLFC adaptive (-e2)"), (3, "QLFC fast (-e0)")):
    want = ref.compress(T, 1, coder)
    pipe = ctx.pipe(4, reuse_outputs=True); tick = []; last = [None]
    def runc(k):
        for _ in range(k):
            tick.append(pipe.submit(d, n, 1, coder, 3))
            if len(tick) >= 4: last[0] = pipe.wait(tick.pop(0))
        while tick: last[0] = pipe.wait(tick.pop(0))
    runc(4); t = time.perf_counter(); runc(12); tp = (time.perf_counter() - t) / 12
    assert last[0].tobytes() == want
    pipe.close()
    print(f"config 3' 64 MiB BWT + {name:20s} (device resident)  ours {tp*1e3:6.2f} ms/block pipelined ({n/MB/tp:7.0f} MB/s)  identical")
print("config 4  8 x 64 MiB across 8 GPUs: one process per GPU, `bench.py --gpus N` (this box has 1 GPU)")
