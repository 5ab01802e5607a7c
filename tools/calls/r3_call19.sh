#!/bin/bash
# round 3, GPU call 19: the round's evidence with the final build — GPU suite, smoke, the driver's bench command, default bench,
# rocprofv3 kernel trace of bench.py + FETCH / WRITE passes over one block, config table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench, the driver's command"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cut -c1-200 $O/bench_20.json
echo "== bench, defaults"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
echo "== profile_round"; timeout 900 bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log | cut -c1-300
echo "== config table"; timeout 900 python tools/config_table.py > $O/config_table.txt 2>&1; cat $O/config_table.txt
echo "== bench config 5 (ST5, 128 MiB)"; timeout 600 python bench.py --sorter 5 --block 134217728 --steps 64 --no-cpu-baseline > $O/bench_config5_st5.json 2>/dev/null; cut -c1-200 $O/bench_config5_st5.json
} > gpurun_out/r3_call19.txt 2>&1
rm -rf gpurun_out/prof_r03/trace_py gpurun_out/prof_r03/pmc_sq gpurun_out/prof_r03/pmc_sq2 gpurun_out/prof_r03/pmc_l2 2>/dev/null
cat gpurun_out/r3_call19.txt | cut -c1-260
