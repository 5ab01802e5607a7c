#!/bin/bash
# round 5, GPU call 27: rocprofv3 evidence of the LAST build (kernel-trace stats of bench.py, then FETCH / WRITE passes over one block)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call27; mkdir -p $O
timeout 400 bash tools/profile_round.sh r05b > $O/profile_round.log 2>&1
rm -rf gpurun_out/prof_r05b/*/*/*_agent_info.csv 2>/dev/null
tail -60 $O/profile_round.log | cut -c1-260
du -sh gpurun_out/prof_r05b
