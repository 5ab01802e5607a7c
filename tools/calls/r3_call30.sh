#!/bin/bash
# round 3, GPU call 30: dispatch change (compress tests, CLI), a longer randomised sweep with cases named
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
echo "== compress tests"; timeout 900 python -m pytest tests/test_gpu_compress.py -x -q 2>&1 | tail -2
echo "== CLI"; timeout 600 python tools/cli_bench.py 2>&1 | tail -4 | tee gpurun_out/r3_final/cli_bench.txt; timeout 300 python tools/cli_bench.py 32 1,4 2>&1 | tail -2 | tee -a gpurun_out/r3_final/cli_bench.txt
echo "== fuzz seed 307, verbose"; FUZZ_VERBOSE=1 timeout 400 python -X faulthandler tools/fuzz_gpu.py 270 307 25165824 > gpurun_out/fuzz307.log 2>&1; echo "exit $?"; grep -v "^case" gpurun_out/fuzz307.log | tail -20; grep "^case" gpurun_out/fuzz307.log | tail -3
} > gpurun_out/r3_call30.txt 2>&1
cat gpurun_out/r3_call30.txt | cut -c1-220
