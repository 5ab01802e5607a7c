#!/bin/bash
# Host-side coder study on the GPU box's CPU (EPYC): BWT of the bench block through the compiled reference, then
# tools/coder_probe.cpp (decision statistics, bracket coalescence, range coder fed by a probability stream).
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from libbsc_amd import api
from oracle.refbind import Ref
x = api.synth_text_v1(2, 64 << 20)
L, idx, aux = Ref().bwt_encode(x)
L.tofile('/tmp/bwt_64m_s2.bin')
PY
g++ -O2 -std=c++17 -march=native -I libbsc_amd/csrc/host -I include tools/coder_probe.cpp libbsc_amd/csrc/host/coder.cpp -o /tmp/coder_probe -lpthread
/tmp/coder_probe /tmp/bwt_64m_s2.bin 8192 | grep -v "^  class"
