#!/bin/bash
# round 3, GPU call 35: bench lines of the final build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_final
{
echo "== bench, the driver's command"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final/bench_20.json 2>/dev/null; cut -c1-200 gpurun_out/r3_final/bench_20.json
echo "== bench, defaults"; timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3_final/bench_default.json 2>/dev/null; cut -c1-200 gpurun_out/r3_final/bench_default.json
} > gpurun_out/r3_call35.txt 2>&1
cat gpurun_out/r3_call35.txt | cut -c1-220
