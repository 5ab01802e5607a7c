#!/bin/bash
# round 5, GPU call 33: the eight-lane step with the low word out of the mask registers too (BSC_RC_VSEL=3), all four steps with prefetch 256
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call33; mkdir -p $O
timeout 100 libbsc_amd/lib/rc_host_bench 67108864 2 quick > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-200
