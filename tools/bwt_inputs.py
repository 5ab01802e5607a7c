"""BWT timing over input classes that are not synth-text v1 (the bench block): python tools/bwt_inputs.py [n_MiB] [class] [--lzp H,M]
Classes (64 MiB each by default, built from what the image holds — there is no network):
  synth-text v1     the bench block
  deep-lcp          3 MB passages repeated (the golden block of tests/golden/golden_big.json)
  python-source     concatenated *.py files of the image's Python installation (natural-language-like: identifiers, English comments)
  dna4              four-symbol pseudo-DNA with repeats (order-5 Markov source + copied segments)
  binary            concatenated shared objects (*.so) of the image
Every output is compared with the reference (oracle/_ref, libsais) — bit-exact or the line says MISMATCH; the time is the best of
three device-resident bscgpu_bwt_device calls, with the number of refinement rounds and the ratio to the synth-text time.
--lzp H,M: what the sorter sees under the CLI's defaults (bsc.cpp: -H15 -M128 unless -p is given) — every class first goes through
bsc_lzp_compress(H, M) on the host, as bsc_compress does (libbsc.cpp:82-96; a block LZP does not shrink is sorted as it is), and the
line gives the time for the (shorter) LZP output next to the size it has; the ratio is still against the 64 MiB synth-text block."""
import glob, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
from libbsc_amd.synth import synth_text_v1, synth_repeat_v1, image_corpus
from oracle.refbind import Ref

lzp = None
if "--lzp" in sys.argv:
    k = sys.argv.index("--lzp"); lzp = tuple(int(x) for x in sys.argv[k + 1].split(",")); del sys.argv[k:k + 2]
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 64) << 20
only = sys.argv[2] if len(sys.argv) > 2 else None          # one class only (with BSCGPU_DEBUG=1: the sorter's round-by-round trace)


def dna4(n, seed=5):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, n, dtype=np.uint8)
    # order-5 flavour: every position copies the symbol 5 back with probability 1/2, and ~2 % of the block are copies of earlier segments
    m = rng.integers(0, 2, n, dtype=np.uint8).astype(bool)
    base[5:][m[5:]] = base[:-5][m[5:]]
    for _ in range(200):
        L = int(rng.integers(1000, 20000)); src = int(rng.integers(0, n - L)); dst = int(rng.integers(0, n - L))
        base[dst:dst + L] = base[src:src + L]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[base]


classes = [("synth-text v1", synth_text_v1(2, n)),
           ("deep-lcp", synth_repeat_v1(4, n, 3_000_000)),
           ("python-source", image_corpus("python-source", n)),
           ("dna4", dna4(n)),
           ("binary", image_corpus("binary", n))]
ref = Ref()
ctx = GpuContext(0, max_n=n + 4096)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
base_ms = None
print(f"{'class':16s} {'ms':>8s} {'x synth':>8s} {'rounds':>6s} {'GB/s':>6s}  parity   (distinct bytes)")
for name, T in classes:
    if only and name != only and name != "synth-text v1":
        continue
    if T is None:
        print(f"{name:16s} (no source files in this image)"); continue
    note = ""
    if lzp:
        z = api.bsc_lzp_compress(T, lzp[0], lzp[1])
        if isinstance(z, int):
            note = f"  [LZP {lzp[0]},{lzp[1]}: not compressible, sorted as it is]"
        else:
            note = f"  [LZP {lzp[0]},{lzp[1]}: {T.size} -> {len(z)} bytes]"
            T = np.frombuffer(z, dtype=np.uint8).copy()
    n = T.size
    d = torch.from_numpy(T).cuda()
    r = 1 << ((n // 8).bit_length() - 1)
    best = None
    for rep in range(1 if (only and os.environ.get("BSCGPU_DEBUG")) else 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, I = ctx.bwt_device(d, out, n, aux_rate=r)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    rounds = ctx.last_stage_ms()[5]
    L_ref, idx_ref, aux_ref = ref.bwt_encode(T)
    ok = (idx == idx_ref) and np.array_equal(out[:n].cpu().numpy(), np.frombuffer(L_ref, dtype=np.uint8)) and [I[t + 1] - 1 for t in range(len(aux_ref))] == list(aux_ref)
    if base_ms is None: base_ms = best
    print(f"{name:16s} {best*1e3:8.2f} {best/base_ms:8.2f} {int(rounds):6d} {n/1e9/best:6.2f}  {'bit-exact' if ok else 'MISMATCH'}   ({np.unique(T).size}){note}", flush=True)
