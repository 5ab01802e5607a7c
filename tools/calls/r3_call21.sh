#!/bin/bash
# round 3, GPU call 21: nine records per lane in the single-read pass (8640-record tiles, 159 KB of LDS): parity and time; text sort with batched loads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
V=libbsc_amd/lib/variants
{
echo "== single-read tests, nine records per lane"; BSC_LIB_OVERRIDE=$(pwd)/$V/libbsc_os_items9.so timeout 900 python -m pytest tests/test_gpu_device.py -x -q -k "single_read or radix_sort_matches or bwt" 2>&1 | tail -3
echo "== bwt + golden tests, default build (text sort with batched loads)"; timeout 900 python -m pytest tests/test_gpu_device.py tests/test_gpu_compress.py -x -q -k "bwt or golden" 2>&1 | tail -3
timeout 1200 python tools/os_ab.py default $V/libbsc_os_items9.so default:BSC_RS_ONESWEEP=0 2>&1 | tail -4
echo "== one block: kernel stats"
P=$(pwd)/gpurun_out/prof_r03b; mkdir -p $P
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/one4 -o b -- python tools/pmc_one_block.py > $P/one4.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03b/one4/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if "textsort" in r["Name"] or "seg_" in r["Name"] or i < 4: print("%-60s calls %4s total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
PY
echo "== one block, nine records per lane"
BSC_LIB_OVERRIDE=$(pwd)/$V/libbsc_os_items9.so timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/one5 -o b -- python tools/pmc_one_block.py > $P/one5.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03b/one5/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if "onesweep" in r["Name"]: print("%-60s calls %4s total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
PY
} > gpurun_out/r3_call21.txt 2>&1
rm -rf gpurun_out/prof_r03b/one4/*/*_agent_info.csv gpurun_out/prof_r03b/one5/*/*_agent_info.csv 2>/dev/null
cat gpurun_out/r3_call21.txt | cut -c1-330
