#!/bin/bash
# round 3, GPU call 16: shared coder pool with adaptive task shapes, staged p-stream writes, dc_ctx with independent neighbour loads,
# long groups counted by seg_apply (no sort launched that would give up) — full GPU suite, the driver's bench command, per-kernel times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== bench, the driver's command"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cut -c1-1500 $O/bench_20.json; tail -3 $O/bench_20.err
echo "== bench, the driver's command, eight lanes always (BSC_RC_ADAPTIVE=0)"; BSC_RC_ADAPTIVE=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160
echo "== bench, defaults"; timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cut -c1-160 $O/bench_default.json
python - <<'PY'
import json
for f in ("bench_20", "bench_default"):
    try:
        d = json.load(open("gpurun_out/r3_final/%s.json" % f))
        print(f, d["value"], d["host"]["blocks_by_coder_task_shape_rank0"], "cpu busy", d["host"]["cpu_busy_fraction_of_effective"], "frac", d["roofline"]["frac"])
    except Exception as e: print(f, "unreadable", e)
PY
echo "== one block: replays, kernel stats"; BSCGPU_DEBUG=1 timeout 300 python tools/pmc_one_block.py 2>&1 | grep "devcoder\|compressed" | tail -3
P=$(pwd)/gpurun_out/prof_r03b; mkdir -p $P
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/one -o b -- python tools/pmc_one_block.py > $P/one.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03b/one/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    for i, r in enumerate(csv.DictReader(open(f))):
        tot += float(r["TotalDurationNs"]) / 1e3
        if i < 26: print("%-60s calls %4s total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
    print("all kernels of the block: %.1f us" % tot)
PY
echo "== bwt input classes"; BSCGPU_DEBUG=1 timeout 600 python tools/bwt_inputs.py 64 > $O/bwt_inputs_trace.txt 2>&1; grep -v "^\[bwt\]\|^\[devcoder\]" $O/bwt_inputs_trace.txt | tail -8 | tee $O/bwt_inputs.txt
grep "round\|long" $O/bwt_inputs_trace.txt | awk 'NR>12' | cut -c1-150 | head -70
} > gpurun_out/r3_call16.txt 2>&1
rm -rf gpurun_out/prof_r03b/one/*/*_agent_info.csv 2>/dev/null
cat gpurun_out/r3_call16.txt | cut -c1-1600
