#!/bin/bash
# round 3, GPU call 34: seg_reduce with 16-byte loads of its side array — full GPU suite, seg time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bwt profile"; timeout 300 python tools/perf_bwt.py 2>&1 | grep "bwt iter 2\|profiled\|pack\|  seg"
} > gpurun_out/r3_call34.txt 2>&1
cat gpurun_out/r3_call34.txt | cut -c1-220
