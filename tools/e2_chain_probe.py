"""Mixer-chain statistics of the adaptive coder (-e2) on the bench block: python tools/e2_chain_probe.py [out.txt]
Builds tools/e2_chain_probe.cpp, sorts the 64 MiB synth-text v1 block (seed 2) with the compiled reference's BWT and walks its eight
sub-blocks.  CPU only (needs oracle/_ref)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libbsc_amd import api
from oracle.refbind import Ref
exe = os.path.join(tempfile.gettempdir(), "e2_chain_probe")
subprocess.run(["g++", "-O2", "-std=c++17", "-march=x86-64-v3", "-I", os.path.join(ROOT, "libbsc_amd/csrc/host"), "-I", os.path.join(ROOT, "include"),
                os.path.join(ROOT, "tools/e2_chain_probe.cpp"), "-o", exe], check=True)
L = Ref().bwt_encode(api.synth_text_v1(2, 64 << 20))[0]
path = os.path.join(tempfile.gettempdir(), "bench_block.bwt")
L.tofile(path)
out = subprocess.run([exe, path, "8"], check=True, capture_output=True, text=True).stdout
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out)
