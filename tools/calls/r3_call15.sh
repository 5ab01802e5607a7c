#!/bin/bash
# round 3, GPU call 15: the round's rocprofv3 evidence (kernel trace of bench.py, FETCH / WRITE passes over one block), SQ / L2 counters
# of the same block for the device coder's kernels, and the sorter's round-by-round trace on the inputs that need prefix doubling
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== profile_round"; timeout 900 bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
P=$(pwd)/gpurun_out/prof_r03
echo "== SQ counters, one block"
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $P/pmc_sq -o s -- python tools/pmc_one_block.py > $P/pmc_sq.log 2>&1
python tools/pmc_table.py $P/pmc_sq > $O/pmc_sq.txt 2>&1; head -40 $O/pmc_sq.txt
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $P/pmc_sq2 -o s -- python tools/pmc_one_block.py > $P/pmc_sq2.log 2>&1
python tools/pmc_table.py $P/pmc_sq2 > $O/pmc_sq2.txt 2>&1; head -40 $O/pmc_sq2.txt
echo "== L2 counters, one block"
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $P/pmc_l2 -o l -- python tools/pmc_one_block.py > $P/pmc_l2.log 2>&1
python tools/pmc_table.py $P/pmc_l2 > $O/pmc_l2.txt 2>&1; head -40 $O/pmc_l2.txt
echo "== sorter trace: python-source"; BSCGPU_DEBUG=1 timeout 300 python tools/bwt_inputs.py 64 python-source 2>&1 | grep "\[bwt\]\|python-source" | tail -22
echo "== sorter trace: binary"; BSCGPU_DEBUG=1 timeout 300 python tools/bwt_inputs.py 64 binary 2>&1 | grep "\[bwt\]\|binary" | tail -22
echo "== sorter trace: deep-lcp"; BSCGPU_DEBUG=1 timeout 300 python tools/bwt_inputs.py 64 deep-lcp 2>&1 | grep "\[bwt\]\|deep-lcp" | tail -16
echo "== kernel stats: python-source"; timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $P/trace_py -o py -- python tools/bwt_inputs.py 64 python-source > $P/trace_py.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r03/trace_py/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 16: print("%-60s calls %6s total %10.1f us avg %9.1f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
} > gpurun_out/r3_call15.txt 2>&1
rm -rf gpurun_out/prof_r03/trace_py/*/*_agent_info.csv 2>/dev/null
du -sh gpurun_out/prof_r03 2>/dev/null
cat gpurun_out/r3_call15.txt | cut -c1-250
