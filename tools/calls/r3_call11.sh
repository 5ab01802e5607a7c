#!/bin/bash
# round 3, GPU call 11: what smaller tiles / fewer streaming waves cost the digit pass without any look-back (ablation builds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
timeout 1200 python tools/os_ab.py $V/libbsc_os_abl1.so $V/libbsc_os_abl1_sw13.so $V/libbsc_os_abl1_sw12.so $V/libbsc_os_abl1_sw11.so $V/libbsc_os_abl1_sw10.so 2>&1 | tail -8
} > gpurun_out/r3_call11.txt 2>&1
cat gpurun_out/r3_call11.txt
