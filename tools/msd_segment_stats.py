import sys, numpy as np
sys.path.insert(0,'/root/repo')
from libbsc_amd import api
n = 64<<20
T = api.synth_text_v1(2, n)
syms = np.unique(T)
lut = np.zeros(256, np.uint64); lut[syms] = np.arange(len(syms), dtype=np.uint64)
c = lut[T]
c = np.concatenate([c, np.zeros(16, np.uint64)])
def topbits(bits):
    # first `bits` bits of the 5-bit-per-char key
    nch = (bits + 4)//5
    k = np.zeros(n, np.uint64)
    for j in range(nch):
        k = (k << np.uint64(5)) | c[j:j+n]
    k >>= np.uint64(nch*5 - bits)
    return k
for bits in (24, 28, 32, 36, 40):
    k = topbits(bits)
    u, cnt = np.unique(k, return_counts=True)
    cnt = np.sort(cnt)[::-1]
    tot = cnt.sum()
    def frac(th): return cnt[cnt > th].sum()/tot
    print(f"top {bits} bits: {len(u)} segments, max {cnt[0]}, records in segments >1024: {frac(1024):.3f}  >4096: {frac(4096):.3f}  >8192: {frac(8192):.3f} >16384: {frac(16384):.3f}; segs>8192: {(cnt>8192).sum()}, sum s^2/n = {(cnt.astype(np.float64)**2).sum()/tot:.0f}")
