#!/bin/bash
# round 3, GPU call 20: qf_rank with binary lifting (parity + time); what the tile order is worth to the write pattern — the single-read
# pass with a static XCD-aware tile map (timing experiment) and the three-kernel scatter with shorter XCD runs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
V=libbsc_amd/lib/variants
{
echo "== compress + device tests"; timeout 1200 python -m pytest tests/test_gpu_compress.py tests/test_gpu_device.py -x -q 2>&1 | tail -3
echo "== one block: kernel stats"
P=$(pwd)/gpurun_out/prof_r03b; mkdir -p $P
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/one3 -o b -- python tools/pmc_one_block.py > $P/one3.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03b/one3/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if "qf_" in r["Name"] or i < 6: print("%-60s calls %4s total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
PY
echo "== tile order"
timeout 1500 python tools/os_ab.py default $V/libbsc_os_static32.so $V/libbsc_os_static8.so $V/libbsc_os_static1.so default:BSC_RS_ONESWEEP=0 $V/libbsc_rst_run8.so:BSC_RS_ONESWEEP=0 $V/libbsc_rst_run1.so:BSC_RS_ONESWEEP=0 2>&1 | tail -9
} > gpurun_out/r3_call20.txt 2>&1
rm -rf gpurun_out/prof_r03b/one3/*/*_agent_info.csv 2>/dev/null
cat gpurun_out/r3_call20.txt | cut -c1-330
