import sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
d = np.load('/tmp/L8m.npy') if False else None
from oracle.refbind import Ref
ref = Ref()
T = api.synth_text_v1(2, 8 << 20)
L, _, _ = ref.bwt_encode(T, aux=False); L = np.ascontiguousarray(L)
t = time.time()
for i in range(10): api.bsc_qlfc_encode_block(L, 1)
print("10 x encode:", time.time() - t)
