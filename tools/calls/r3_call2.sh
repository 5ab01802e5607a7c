#!/bin/bash
# round 3, GPU call 2: where does the single-read pass lose 0.15 ms?  ablation builds + phase stamps, one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
timeout 1200 python tools/os_ab.py default:BSC_RS_ONESWEEP=0 $V/libbsc_os_base.so $V/libbsc_os_abl1.so $V/libbsc_os_abl3.so $V/libbsc_os_abl4.so $V/libbsc_os_abl7.so $V/libbsc_os_valsearly.so 2>&1 | tail -12
echo "== phase stamps"; BSC_LIB_OVERRIDE=$V/libbsc_os_ph.so BSC_RS_ONESWEEP=1 timeout 300 python tools/os_phase_timing.py 2>&1 | tail -40
} > gpurun_out/r3_call2.txt 2>&1
cat gpurun_out/r3_call2.txt
