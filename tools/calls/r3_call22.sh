#!/bin/bash
# round 3, GPU call 22: nine records per lane, second attempt (digit bytes of the ninth record); bare replay loop in dc_eval_b
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
V=libbsc_amd/lib/variants
{
echo "== single-read + bwt + golden tests, nine records per lane"; BSC_LIB_OVERRIDE=$(pwd)/$V/libbsc_os_items9.so timeout 900 python -m pytest tests/test_gpu_device.py tests/test_gpu_compress.py -x -q -k "single_read or radix_sort_matches or bwt or golden" 2>&1 | tail -3
echo "== device coder tests, default build"; timeout 900 python -m pytest tests/test_gpu_device.py tests/test_gpu_compress.py -x -q -k "oracle_trace or golden or device_coder" 2>&1 | tail -3
timeout 1200 python tools/os_ab.py default $V/libbsc_os_items9.so default:BSC_RS_ONESWEEP=0 2>&1 | tail -4
echo "== one block, nine records per lane: kernel stats"
P=$(pwd)/gpurun_out/prof_r03b; mkdir -p $P
BSC_LIB_OVERRIDE=$(pwd)/$V/libbsc_os_items9.so timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/one6 -o b -- python tools/pmc_one_block.py > $P/one6.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03b/one6/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if "onesweep" in r["Name"] or "eval" in r["Name"] or "textsort" in r["Name"]: print("%-60s calls %4s total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
PY
echo "== bench 20 steps, nine records per lane"; BSC_LIB_OVERRIDE=$(pwd)/$V/libbsc_os_items9.so timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_final/bench_20_items9.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_final/bench_20_items9.json")); r = d["roofline"]
print(d["value"], d["verified"], "frac", r["frac"], "sort_frac", r["sort_frac"], "avg", r["avg_launch_ms"])
PY
} > gpurun_out/r3_call22.txt 2>&1
rm -rf gpurun_out/prof_r03b/one6/*/*_agent_info.csv 2>/dev/null
cat gpurun_out/r3_call22.txt | cut -c1-330
