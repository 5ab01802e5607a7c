"""A/B of host coder builds on one box, one pinned core: python tools/ab_coder.py libA.so libB.so (BSC_LIB_OVERRIDE)."""
import os, subprocess, sys, json
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
from oracle.refbind import Ref
os.sched_setaffinity(0, {6})
ref = Ref()
T = api.synth_text_v1(2, 8 << 20)
L, _, _ = ref.bwt_encode(T, aux=False); L = np.ascontiguousarray(L)
t = time.time(); api.bsc_qlfc_ranks(L); tf = (time.time() - t) * 1e3
res = {"front_ms": tf}
for coder in (1, 2, 3):
    want = ref.qlfc_encode_block(L, coder); best = 1e9
    for i in range(5):
        t = time.time(); got = api.bsc_qlfc_encode_block(L, coder); best = min(best, (time.time() - t) * 1e3)
    res["coder%d_ms" % coder] = best - tf; res["ok%d" % coder] = (got == want)
print("RESULT " + json.dumps(res))
'''
libs = sys.argv[1:]
for rnd in range(2):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, BSC_LIB_OVERRIDE=os.path.abspath(l)))
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        print(os.path.basename(l).ljust(24), line[0][7:] if line else ("FAILED " + r.stderr[-400:]))
