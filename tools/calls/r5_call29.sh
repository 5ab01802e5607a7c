#!/bin/bash
# round 5, GPU call 29: the eight-lane host coder with the range's chain kept out of the mask registers, against the round-4 step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call29; mkdir -p $O
timeout 150 libbsc_amd/lib/rc_host_bench > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-200
