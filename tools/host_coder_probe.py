import os, sys, time, subprocess
sys.path.insert(0, '.')
import numpy as np
print("affinity:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:16], "...")
print(subprocess.run("lscpu -e=CPU,CORE,SOCKET,NODE,CACHE | head -20; cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list; cat /sys/devices/system/cpu/cpu8/cache/index3/shared_cpu_list", shell=True, capture_output=True, text=True).stdout)
from libbsc_amd import api
from oracle.refbind import Ref
ref = Ref()
T = api.synth_text_v1(2, 8 << 20)
L, _, _ = ref.bwt_encode(T, aux=False); L = np.ascontiguousarray(L)
want = ref.qlfc_encode_block(L, 1)
mode = os.environ.get("BSC_QLFC_PIPELINE", "1")
for i in range(3):
    t = time.time(); got = api.bsc_qlfc_encode_block(L, 1); dt = time.time() - t
    print(f"pipeline={mode} one 8 MiB stream: {dt*1e3:.1f} ms ok={got == want}")
t = time.time(); r = api.bsc_qlfc_ranks(L); print(f"qlfc_runs only: {(time.time()-t)*1e3:.1f} ms")
t = time.time(); ref.qlfc_encode_block(L, 1); print(f"REF static one stream: {(time.time()-t)*1e3:.1f} ms")

import ctypes as C
from libbsc_amd import _native as N
Lb = N.lib(); Lb.bsc_qlfc_ablate.restype = C.c_ulonglong; Lb.bsc_qlfc_ablate.argtypes = [C.c_void_p, C.c_int, C.c_int]
for mode, name in ((0, "runs only"), (1, "runs + walk (count decisions)"), (2, "runs + walk + 3 counters (no range coder)")):
    Lb.bsc_qlfc_ablate(N.np_ptr(L), L.size, mode)
    t = time.time(); r = Lb.bsc_qlfc_ablate(N.np_ptr(L), L.size, mode); dt = time.time() - t
    print(f"ablate {name:45s} {dt*1e3:7.1f} ms   (ret {r})")
