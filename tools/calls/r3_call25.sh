#!/bin/bash
# round 3, GPU call 25: bench lines with the new default (6 contexts x 3), timeline, randomised parity sweep over this round's kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== bench, the driver's command"; BSC_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cut -c1-200 $O/bench_20.json; grep "\[trace\]" $O/bench_20.err > $O/bench_20_timeline.txt; tail -8 $O/bench_20_timeline.txt
echo "== again"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160
echo "== bench, defaults"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
echo "== fuzz"; timeout 400 python tools/fuzz_gpu.py 200 303 25165824 2>&1 | tail -4
echo "== fuzz, single-read passes for every sort of >= 4 tiles"; BSC_RS_ONESWEEP=2 timeout 300 python tools/fuzz_gpu.py 100 304 25165824 2>&1 | tail -3
} > gpurun_out/r3_call25.txt 2>&1
cat gpurun_out/r3_call25.txt | cut -c1-220
