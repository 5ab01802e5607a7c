#!/bin/bash
# round 3, GPU call 33: bwt_pack with its outputs staged through LDS — full GPU suite, pack time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bwt profile"; timeout 300 python tools/perf_bwt.py 2>&1 | grep "bwt iter 2\|profiled\|pack\|scatter n="
} > gpurun_out/r3_call33.txt 2>&1
cat gpurun_out/r3_call33.txt | cut -c1-220
