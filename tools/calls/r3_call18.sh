#!/bin/bash
# round 3, GPU call 18: tail blocks marked low-latency, adaptive against eight-lanes-always over 320 steps, the CLI with the caller-aware thread count
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== compress tests"; timeout 900 python -m pytest tests/test_gpu_compress.py -x -q 2>&1 | tail -3
echo "== bench, the driver's command, with timeline"; BSC_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_trace.json 2> $O/bench_20_trace.err; cut -c1-140 $O/bench_20_trace.json; grep "\[trace\]" $O/bench_20_trace.err | tail -24
echo "== bench, the driver's command, tail not marked"; BSC_BENCH_TAIL=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-140
echo "== bench 320, adaptive"; timeout 900 python bench.py --no-cpu-baseline 2>/dev/null > $O/b320_a.json; cut -c1-140 $O/b320_a.json
echo "== bench 320, eight lanes always"; BSC_RC_ADAPTIVE=0 timeout 900 python bench.py --no-cpu-baseline 2>/dev/null > $O/b320_x8.json; cut -c1-140 $O/b320_x8.json
echo "== bench 320, adaptive again"; timeout 900 python bench.py --no-cpu-baseline 2>/dev/null > $O/b320_a2.json; cut -c1-140 $O/b320_a2.json
python - <<'PY'
import json
for f in ("bench_20_trace", "b320_a", "b320_x8", "b320_a2"):
    try:
        d = json.load(open("gpurun_out/r3_final/%s.json" % f))
        print(f, d["value"], d["host"]["blocks_by_coder_task_shape_rank0"], "cpu busy", d["host"]["cpu_busy_fraction_of_effective"], "cpu-s/block", d["host"]["cpu_seconds_per_block_rank0"])
    except Exception as e: print(f, "unreadable", e)
PY
echo "== CLI"; timeout 600 python tools/cli_bench.py 2>&1 | tail -6 | tee $O/cli_bench.txt
} > gpurun_out/r3_call18.txt 2>&1
cat gpurun_out/r3_call18.txt | cut -c1-200
