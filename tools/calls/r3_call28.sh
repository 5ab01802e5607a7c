#!/bin/bash
# round 3, GPU call 28: the relinked CLI on a larger input (start-up amortised), a stress of the synchronous path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
echo "== CLI, 32 blocks"; timeout 600 python tools/cli_bench.py 32 2,4,8 2>&1 | tail -4
echo "== sync stress"; timeout 200 python -X faulthandler tools/sync_stress.py 100 11 > gpurun_out/sync_stress.log 2>&1; echo "exit $?"; grep -v "^case" gpurun_out/sync_stress.log | tail -8; grep "^case" gpurun_out/sync_stress.log | tail -2
} > gpurun_out/r3_call28.txt 2>&1
cat gpurun_out/r3_call28.txt | cut -c1-220
