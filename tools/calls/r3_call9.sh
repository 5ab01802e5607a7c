#!/bin/bash
# round 3, GPU call 9: finer scout stamps; round-by-round traces of the BWT on long-group inputs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
echo "== phase stamps"; BSC_LIB_OVERRIDE=$V/libbsc_os_ph.so BSC_RS_ONESWEEP=1 timeout 300 python tools/os_phase_timing.py 2>&1 | tail -28
echo "== trace: synth text, 11-character first-sort keys"; BSCGPU_DEBUG=1 BSC_BWT_W=11 timeout 300 python tools/perf_bwt.py 2>&1 | grep "\[bwt\]\|bwt iter" | head -14
echo "== trace: synth text, 9-character keys"; BSCGPU_DEBUG=1 BSC_BWT_W=9 timeout 300 python tools/perf_bwt.py 2>&1 | grep "\[bwt\]\|bwt iter" | head -14
echo "== trace: python-source"; BSCGPU_DEBUG=1 timeout 300 python tools/bwt_inputs.py 64 python-source 2>&1 | grep "\[bwt\]\|python-source" | tail -30
echo "== trace: python-source, no long split"; BSC_BWT_LONGSPLIT=0 BSCGPU_DEBUG=1 timeout 300 python tools/bwt_inputs.py 64 python-source 2>&1 | grep "\[bwt\]\|python-source" | tail -30
} > gpurun_out/r3_call9.txt 2>&1
cat gpurun_out/r3_call9.txt
