import time, numpy as np, sys
sys.path.insert(0,'.')
from libbsc_amd import api
from libbsc_amd.synth import synth_repeat_v1
T=api.synth_text_v1(2, 64<<20)
for lz in ((0,0),(15,128)):
    for it in range(3):
        t0=time.time(); b=api.bsc_compress(T,1,1,lzp_hash=lz[0],lzp_min=lz[1]); dt=time.time()-t0
    print('bsc_compress host API 64MiB text lzp',lz,len(b),'%.1f ms -> %.0f MB/s'%(dt*1e3, T.size/dt/1e6))
R=synth_repeat_v1(5, 64<<20, 3_000_000)
for lz in ((0,0),(15,128)):
    for it in range(2):
        t0=time.time(); b=api.bsc_compress(R,1,1,lzp_hash=lz[0],lzp_min=lz[1]); dt=time.time()-t0
    print('bsc_compress host API 64MiB repeat lzp',lz,len(b),'%.1f ms -> %.0f MB/s'%(dt*1e3, R.size/dt/1e6))
# concurrent callers on the drop-in API (like the reference CLI's OpenMP team): 3 threads x 4 blocks each
import threading
blocks = [api.synth_text_v1(10 + i, 64 << 20) for i in range(3)]
def work(i):
    for _ in range(4): api.bsc_compress(blocks[i], 1, 1)
t0 = time.time(); ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
[t.start() for t in ths]; [t.join() for t in ths]; dt = time.time() - t0
print('bsc_compress from 3 concurrent threads, 12 x 64 MiB: %.0f MB/s' % (12 * (64 << 20) / dt / 1e6))
