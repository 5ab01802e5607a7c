#!/usr/bin/env python3
"""Generate tests/golden/golden_big.json: reference outputs for the full-size BASELINE configurations the `-m gpu` suite
checks without /root/reference (the GPU box does not have it):
  config 3 — the 64 MiB seed-2 block of `bench.py` at N=1;
  config 4 — eight independent 64 MiB synth-text v1 blocks, seeds 10..17, BWT + QLFC static (the N>1 bench blocks);
  config 5 — one 128 MiB synth-text v1 block, seed 3, ST5 and ST6 + QLFC static (SURVEY.md §8c pins the same md5s);
  config3-e2 / -e0, config5-e0 — the config 3 block through the adaptive and the fast coder, the config 5 block through the fast one;
  deep-LCP — one 64 MiB block of long repeated passages (synth_repeat_v1), BWT + QLFC static: many doubling rounds;
  max-block — one block of the format's maximum size, 1 GiB (libbsc.cpp:221) of synth-text v1, seed 4, BWT + QLFC static (round 6).
Run in the build container: python tests/golden/make_golden_big.py [tag ...]   (needs oracle/_ref, i.e. /root/reference).
With tags given, only those rows are (re)generated and merged into the existing file."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libbsc_amd import api  # noqa: E402
from libbsc_amd.synth import synth_repeat_v1  # noqa: E402
from oracle.refbind import Ref  # noqa: E402

os.environ.setdefault("BSC_REF_THREADS", "8")
ref = Ref()
out = []
ONLY = set(sys.argv[1:])


def want(tag):
    return not ONLY or tag in ONLY


def add(tag, gen, T, sorter, coder, features):
    if not want(tag):
        return
    blk = ref.compress(T, sorter, coder, features=features)
    e = dict(tag=tag, gen=gen, n=int(T.size), sorter=sorter, coder=coder, features=features, size=len(blk),
             md5=hashlib.md5(blk).hexdigest(), input_md5=hashlib.md5(T.tobytes()).hexdigest())
    if sorter == 1:
        L, idx, aux = ref.bwt_encode(T)
        e.update(bwt_index=idx, bwt_md5=hashlib.md5(L.tobytes()).hexdigest())
    out.append(e)
    print(tag, gen, e["size"], e["md5"], flush=True)


T2 = api.synth_text_v1(2, 64 << 20) if (want("config3") or want("config3-e2") or want("config3-e0")) else None
add("config3", {"kind": "synth", "seed": 2}, T2, 1, 1, 3)      # the N=1 bench block
# the same block through the other two coders (round 5: `bench.py --coder 2 / 3` lines are checked at the size they are quoted on;
# SURVEY.md 8c pins the -e2 output: 15 148 620 B, md5 bea58a30...)
add("config3-e2", {"kind": "synth", "seed": 2}, T2, 1, 2, 3)
add("config3-e0", {"kind": "synth", "seed": 2}, T2, 1, 3, 3)
del T2
for seed in range(10, 18):
    if want("config4"):
        add("config4", {"kind": "synth", "seed": seed}, api.synth_text_v1(seed, 64 << 20), 1, 1, 3)
T = api.synth_text_v1(3, 128 << 20) if (want("config5") or want("config5-e0")) else None
add("config5", {"kind": "synth", "seed": 3}, T, 5, 1, 3)
add("config5", {"kind": "synth", "seed": 3}, T, 6, 1, 3)
add("config5-e0", {"kind": "synth", "seed": 3}, T, 5, 3, 3)
add("config5-e0", {"kind": "synth", "seed": 3}, T, 6, 3, 3)
if want("deep-lcp"):
    add("deep-lcp", {"kind": "repeat", "seed": 7, "period": 3_000_000}, synth_repeat_v1(7, 64 << 20, 3_000_000), 1, 1, 3)
if want("max-block"):
    add("max-block", {"kind": "synth", "seed": 4}, api.synth_text_v1(4, 1 << 30), 1, 1, 3)
path = os.path.join(ROOT, "tests/golden/golden_big.json")
if ONLY:
    old = json.load(open(path))["blocks"]
    out = [e for e in old if e["tag"] not in ONLY] + out
json.dump({"reference": "libbsc 3.3.5 (oracle/_ref)", "blocks": out}, open(path, "w"), indent=1)
