#!/bin/bash
# round 3, GPU call 36: seg_apply with wide loads — full GPU suite, seg time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
{
echo "== full GPU suite"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bwt profile"; timeout 100 python tools/perf_bwt.py 2>&1 | grep "bwt iter 2\|  seg"
} > gpurun_out/r3_call36.txt 2>&1
cat gpurun_out/r3_call36.txt | cut -c1-220
