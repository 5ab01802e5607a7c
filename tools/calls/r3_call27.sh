#!/bin/bash
# round 3, GPU call 27: the sweep's heap corruption again, default allocator, cases named
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
for s in 303 305; do
echo "== fuzz seed $s, verbose"; FUZZ_VERBOSE=1 timeout 300 python -X faulthandler tools/fuzz_gpu.py 170 $s 25165824 > gpurun_out/fuzz$s.log 2>&1; echo "exit $?"; grep -v "^case" gpurun_out/fuzz$s.log | tail -30; echo "last cases:"; grep "^case" gpurun_out/fuzz$s.log | tail -4
done
} > gpurun_out/r3_call27.txt 2>&1
cat gpurun_out/r3_call27.txt | cut -c1-250
