"""Background host load equal to one bench rank's coder pool: 16 threads running the static QLFC coder in a loop."""
import sys, threading, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
from oracle.refbind import Ref
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20
T = api.synth_text_v1(5, 8 << 20)
L, _, _ = Ref().bwt_encode(T, aux=False); L = np.ascontiguousarray(L)
stop = time.time() + secs
def work():
    while time.time() < stop:
        api.bsc_qlfc_encode_block(L, 1)        # ctypes releases the GIL
ths = [threading.Thread(target=work) for _ in range(16)]
[t.start() for t in ths]; [t.join() for t in ths]
