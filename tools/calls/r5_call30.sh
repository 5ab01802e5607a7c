#!/bin/bash
# round 5, GPU call 30: where the eight-lane host coder's 17.6 cycles per step go — without the replay, without the log; the AVX2 step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call30; mkdir -p $O
{ timeout 150 libbsc_amd/lib/rc_host_bench; echo "=== BSC_RC_AVX512=0 (the AVX2 step)"; BSC_RC_AVX512=0 timeout 100 libbsc_amd/lib/rc_host_bench | grep -v "malloc\|single\|two-stream" | head -8; } > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-200
