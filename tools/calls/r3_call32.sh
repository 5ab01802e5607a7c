#!/bin/bash
# round 3, GPU call 32: last check of the tree as committed — GPU suite, smoke, the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_final
{
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "== bench, the driver's command"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final/bench_20_last.json 2>/dev/null; cut -c1-200 gpurun_out/r3_final/bench_20_last.json
} > gpurun_out/r3_call32.txt 2>&1
cat gpurun_out/r3_call32.txt | cut -c1-220
