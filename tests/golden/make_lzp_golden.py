#!/usr/bin/env python3
"""Generate tests/golden/lzp_golden.json from the compiled reference (oracle/_ref/libbsc_ref.so, built from
/root/reference).  Run in the build container: python tests/golden/make_lzp_golden.py
Entries: `synth-repeat v1` inputs (libbsc_amd/synth.py), the reference's bsc_lzp_compress output (size + md5, or the
error code) for every encoder variant the reference selects by (hashSize, minLen), and whole bsc_compress blocks
written with LZP on."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libbsc_amd.synth import synth_repeat_v1  # noqa: E402
from oracle.refbind import Ref  # noqa: E402

ref = Ref()
stage, blocks = [], []
INPUTS = ((3, 100_000, 5000), (4, 700_000, 30_000), (5, 5 << 20, 70_000), (6, 17 << 20, 400_000))
VARIANTS = ((15, 4), (15, 6), (15, 8), (15, 12), (15, 16), (15, 32), (15, 128), (18, 5), (18, 40), (20, 255))
for seed, n, period in INPUTS:
    T = synth_repeat_v1(seed, n, period)
    for h, m in VARIANTS:
        for f in (1, 3):
            r = ref.lzp_compress(T, h, m, features=f)
            e = dict(seed=seed, n=n, period=period, hash=h, minlen=m, features=f)
            e.update(dict(error=r) if isinstance(r, int) else dict(size=len(r), md5=hashlib.md5(r).hexdigest()))
            stage.append(e)
for seed, n, period in INPUTS[:3]:
    T = synth_repeat_v1(seed, n, period)
    for (h, m, sorter, coder) in ((15, 128, 1, 1), (16, 32, 1, 2), (18, 8, 5, 1), (15, 4, 1, 3)):
        blk = ref.compress(T, sorter, coder, lzp_hash=h, lzp_min=m, features=3)
        blocks.append(dict(seed=seed, n=n, period=period, hash=h, minlen=m, sorter=sorter, coder=coder, features=3,
                           size=len(blk), md5=hashlib.md5(blk).hexdigest(), header=blk[:28].hex()))
json.dump({"reference": "libbsc 3.3.5 (oracle/_ref)", "stage": stage, "blocks": blocks},
          open(os.path.join(ROOT, "tests/golden/lzp_golden.json"), "w"), indent=1)
print(len(stage), "stage entries,", len(blocks), "blocks")
