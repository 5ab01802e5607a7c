"""Device static coder vs the oracle's trace of the reference model, sub-block by sub-block, with a first-mismatch report
(run on the GPU box):  python tools/devcoder_check.py [quick|full]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import GpuContext, api
from libbsc_amd.gpu import GpuError
from oracle.refbind import Oracle, Ref

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
orc = Oracle(); ref = Ref()
rng = np.random.default_rng(11)


def bwt(x):
    return ref.bwt_encode(x)[0]


cases = [("text300k", bwt(api.synth_text_v1(3, 300_000))),
         ("text1m", bwt(api.synth_text_v1(1, 1 << 20))),
         ("low1m", bwt(rng.integers(0, 3, 1 << 20, dtype=np.uint8))),
         ("zeros", np.zeros(500_000, np.uint8)),
         ("n100", bwt(api.synth_text_v1(8, 100))),
         ("sym40", bwt((rng.geometric(0.15, 700_000) % 40).astype(np.uint8))),
         ("longruns", np.repeat(rng.integers(0, 6, 3000, dtype=np.uint8), rng.integers(1, 3000, 3000)).astype(np.uint8)),
         ("text5m", bwt(api.synth_text_v1(4, 5 << 20))),
         # 224 symbols with text-like structure: every 64 KiB segment of the text is shifted into its own 32-symbol band
         ("text224", bwt(((api.synth_text_v1(6, 3 << 20) & 31) + ((np.arange(3 << 20) >> 16) % 7 * 32).astype(np.uint8)).astype(np.uint8)))]
if mode == "full":
    cases += [("text20m", bwt(api.synth_text_v1(5, 20 << 20))), ("text64m", bwt(api.synth_text_v1(2, 64 << 20))),
              ("rand2m", rng.integers(0, 256, 2 << 20, dtype=np.uint8)), ("skew2m", bwt((rng.geometric(0.02, 2 << 20) % 256).astype(np.uint8)))]
maxn = max(c[1].size for c in cases)
ctx = GpuContext(0, max_n=maxn + 4096)
bad = 0
for name, L in cases:
    t0 = time.time()
    try:
        if L.size >= (16 << 20):
            ctx.qlfc_static_pstream(L, debug=False)          # warm-up (arena allocation), then a profiled call
            ctx.profile(True); ctx.profile_reset()
            t0 = time.time()
            ctx.qlfc_static_pstream(L, debug=False)
            print("   call %.1f ms; kernels:" % (1e3 * (time.time() - t0)), {k: round(v["ms"], 2) for k, v in ctx.profile_get().items() if v["launches"]})
            ctx.profile(False)
        t0 = time.time()
        ps, st, sz, poff, dbg = ctx.qlfc_static_pstream(L, debug=True)
    except GpuError as e:
        print(f"{name:10s} n={L.size}: declined/failed: {e}")
        continue
    t1 = time.time()
    ok = True
    for b in range(len(st)):
        sub = L[st[b]:st[b] + sz[b]]
        tr, ct = orc.static_pstream(sub, counters=True)
        mine = ps[poff[b]:poff[b + 1]]
        if len(tr) != len(mine):
            print(f"{name} sub {b}: {len(mine)} decisions vs oracle {len(tr)}"); ok = False
        k = min(len(tr), len(mine))
        neq = np.nonzero(tr[:k] != mine[:k])[0]
        if neq.size:
            ok = False
            i = int(neq[0])
            d = dbg[:, poff[b] + i]
            print(f"{name} sub {b}: {neq.size} of {k} entries differ; first at {i}: mine {mine[i]:#06x} oracle {tr[i]:#06x}; counters S/C/P mine {d.tolist()} oracle {ct[i].tolist()}")
            for fam, nm in enumerate("SCP"):
                dd = np.nonzero(dbg[fam, poff[b]:poff[b] + k] != ct[:k, fam])[0]
                print(f"    family {nm}: {dd.size} values differ" + (f", first at {int(dd[0])}: mine {int(dbg[fam, poff[b] + dd[0]])} oracle {int(ct[dd[0], fam])}" if dd.size else ""))
    bad += not ok
    print(f"{name:10s} n={L.size} sub-blocks={len(st)} decisions={len(ps)} gpu call {1e3 * (t1 - t0):.1f} ms  {'OK' if ok else 'MISMATCH'}", flush=True)
print("ALL OK" if not bad else f"{bad} case(s) FAILED")
