#!/bin/bash
# round 5, GPU call 32: the eight-lane step with the bit folded into the multiplier (chain: shift, multiply, select, add)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call32; mkdir -p $O
timeout 100 libbsc_amd/lib/rc_host_bench 67108864 2 quick > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-200
