#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the compiled reference (oracle/_ref/libbsc_ref.so, built from
/root/reference).  Run in the build container: python tests/golden/make_golden.py
Each entry: input (synth-text v1 seed/n, or literal hex), sorter, coder, and the reference's
bsc_compress(features=1, lzp off) output size + md5 (+ BWT primary index and aux indexes for sorter 1)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libbsc_amd.synth import synth_text_v1  # noqa: E402
from oracle.refbind import Ref  # noqa: E402

ref = Ref()
blocks = []


def add(kind, T, sorter, coder, **meta):
    blk = ref.compress(T, sorter, coder, features=1)
    e = dict(kind=kind, n=int(T.size), sorter=sorter, coder=coder, size=len(blk), md5=hashlib.md5(blk).hexdigest(),
             input_md5=hashlib.md5(T.tobytes()).hexdigest(), header=blk[:28].hex(), **meta)
    if sorter == 1 and T.size >= 16:
        L, idx, aux = ref.bwt_encode(T)
        e.update(bwt_index=idx, bwt_aux=aux, bwt_md5=hashlib.md5(L.tobytes()).hexdigest())
    blocks.append(e)


for seed, n in ((1, 1 << 20), (5, 1 << 16), (6, 65535), (3, 300_000), (4, 5 << 20)):
    T = synth_text_v1(seed, n)
    for sorter, coder in ((1, 1), (1, 2), (1, 3), (5, 1), (6, 1)):
        add("synth", T, sorter, coder, seed=seed)
for lit in (b"abracadabra" * 9, bytes(100), bytes(range(256)) * 3, b"ab" * 77 + b"\0\0\0"):
    T = np.frombuffer(lit, np.uint8)
    for sorter, coder in ((1, 1), (5, 2), (3, 3)):
        add("literal", T, sorter, coder, hex=lit.hex())
# full-size block of BASELINE config 3 (checked on the GPU box only; too slow for the plain-C oracle)
T = synth_text_v1(2, 64 << 20)
add("synth", T, 1, 1, seed=2)
json.dump({"reference": "libbsc 3.3.5 (oracle/_ref)", "oracle_max_n": 300_000, "blocks": blocks},
          open(os.path.join(ROOT, "tests/golden/golden.json"), "w"), indent=1)
print(len(blocks), "entries")
