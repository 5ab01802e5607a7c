#!/bin/bash
# round 5, GPU call 31: timing builds of the eight-lane step (no log / no log and no low word / loads and transposes alone)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call31; mkdir -p $O
timeout 100 libbsc_amd/lib/rc_host_bench 67108864 2 quick > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-200
