"""How much does the static coder gain from a second hardware thread on the same core (upper bound for interleaving two
streams in one software thread)?  Host only."""
import os, sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
from oracle.refbind import Ref
ref = Ref()
T = api.synth_text_v1(2, 8 << 20)
L, _, _ = ref.bwt_encode(T, aux=False); L = np.ascontiguousarray(L)
sib = open("/sys/devices/system/cpu/cpu2/topology/thread_siblings_list").read().strip()
print("cpu2 siblings:", sib, "| affinity size", len(os.sched_getaffinity(0)))
a, b = [int(x) for x in sib.replace("-", ",").split(",")][:2] if ("," in sib or "-" in sib) else (2, 3)
def worker(cpu, reps, out, i):
    os.sched_setaffinity(0, {cpu})
    api.bsc_qlfc_encode_block(L, 1)
    t = time.time()
    for _ in range(reps): api.bsc_qlfc_encode_block(L, 1)
    out[i] = (time.time() - t) / reps
def run(cpus, reps=3):
    out = [0] * len(cpus)
    ths = [threading.Thread(target=worker, args=(c, reps, out, i)) for i, c in enumerate(cpus)]
    [t.start() for t in ths]; [t.join() for t in ths]
    return out
one = run([a])[0]
print(f"1 thread on cpu{a}: {one*1e3:.1f} ms per 8 MiB stream (includes the host run/rank front end)")
two_sib = run([a, b]); print(f"2 threads on siblings cpu{a},cpu{b}: {[round(x*1e3,1) for x in two_sib]} ms  -> throughput x{2*one/max(two_sib):.2f}")
two_sep = run([a, a + 2 if a + 2 != b else a + 4]); print(f"2 threads on separate cores: {[round(x*1e3,1) for x in two_sep]} ms -> throughput x{2*one/max(two_sep):.2f}")
