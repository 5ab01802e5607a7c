#!/bin/bash
# round 3, GPU call 26: the heap corruption the randomised sweep ran into (seed 303), case by case; the coder-pool test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
echo "== coder pool test"; timeout 600 python -m pytest tests/test_gpu_compress.py -x -q -k "coder_pool" 2>&1 | tail -3
echo "== fuzz seed 303, verbose"; FUZZ_VERBOSE=1 MALLOC_CHECK_=3 timeout 420 python -X faulthandler tools/fuzz_gpu.py 300 303 25165824 > gpurun_out/fuzz303.log 2>&1; echo "exit $?"; tail -40 gpurun_out/fuzz303.log
} > gpurun_out/r3_call26.txt 2>&1
cat gpurun_out/r3_call26.txt | cut -c1-250
