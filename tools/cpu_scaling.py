import sys, threading, time, os
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
from oracle.refbind import Ref
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, "->", open(f).read().strip().replace("\n", " | ")[:300])
    except Exception as e: print(f, "n/a")
print("sched_getaffinity:", len(os.sched_getaffinity(0)))
T = api.synth_text_v1(5, 4 << 20)
L, _, _ = Ref().bwt_encode(T, aux=False); L = np.ascontiguousarray(L)
api.bsc_qlfc_encode_block(L, 1)
for nt in (1, 4, 8, 16, 24, 32, 48, 64, 128):
    cnt = [0] * nt
    stop = time.time() + 3.0
    def work(i):
        while time.time() < stop:
            api.bsc_qlfc_encode_block(L, 1); cnt[i] += 1
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.time() - t0
    print(f"{nt:4d} threads: {sum(cnt)*L.size/1e6/dt:8.1f} MB/s total, {sum(cnt)*L.size/1e6/dt/nt:6.1f} MB/s per thread")
try: print("cpu.stat after:", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | ")[:300])
except Exception: pass
