#!/bin/bash
# round 3, GPU call 3: tile = global ticket (claim order = index order) against XCD batches; look-back cost; full GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
timeout 1200 python tools/os_ab.py default:BSC_RS_ONESWEEP=0 $V/libbsc_os_base.so $V/libbsc_os_order1.so $V/libbsc_os_abl1.so $V/libbsc_os_abl3.so 2>&1 | tail -8
echo "== phase stamps (ticket order)"; BSC_LIB_OVERRIDE=$V/libbsc_os_ph.so BSC_RS_ONESWEEP=1 timeout 300 python tools/os_phase_timing.py 2>&1 | tail -32
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} > gpurun_out/r3_call3.txt 2>&1
cat gpurun_out/r3_call3.txt
