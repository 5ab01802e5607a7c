#!/bin/bash
# round 3, GPU call 17: where a 20-step run's time goes (bench timeline), dc_ctx without the contended atomics, replay with loads in flight
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== golden + device coder tests"; timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_device.py -x -q -k "golden or oracle_trace or device_coder or pstream" 2>&1 | tail -3
echo "== bench, the driver's command, with timeline"; BSC_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_trace.json 2> $O/bench_20_trace.err; cut -c1-140 $O/bench_20_trace.json; grep "\[trace\]" $O/bench_20_trace.err
echo "== the same with 2 contexts x 4"; BSC_BENCH_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --contexts 2 2> $O/t2.err | cut -c1-140; grep "\[trace\]" $O/t2.err | tail -12
echo "== the same with 4 contexts x 2"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --depth 2 2>/dev/null | cut -c1-140
echo "== one block: kernel stats"
P=$(pwd)/gpurun_out/prof_r03b; mkdir -p $P
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/one2 -o b -- python tools/pmc_one_block.py > $P/one2.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03b/one2/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    for i, r in enumerate(csv.DictReader(open(f))):
        tot += float(r["TotalDurationNs"]) / 1e3
        if i < 20: print("%-60s calls %4s total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
    print("all kernels of the block: %.1f us" % tot)
PY
} > gpurun_out/r3_call17.txt 2>&1
rm -rf gpurun_out/prof_r03b/one2/*/*_agent_info.csv 2>/dev/null
cat gpurun_out/r3_call17.txt | cut -c1-200
