#!/bin/bash
# round 3, GPU call 29: the final build once more — GPU suite, smoke, the driver's bench command, rocprofv3 evidence with the default (6 x 3) bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "== bench, the driver's command"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cut -c1-200 $O/bench_20.json
echo "== profile_round"; timeout 900 bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | cut -c1-200
} > gpurun_out/r3_call29.txt 2>&1
cat gpurun_out/r3_call29.txt | cut -c1-260
