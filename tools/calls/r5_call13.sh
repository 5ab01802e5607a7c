#!/bin/bash
# round 5, GPU call 13: randomised parity sweeps on the final build (fuzz_gpu: every sorter x coder x LZP on random input classes and sizes; pipe_stress:
# the pipelined entry point with low-latency marks and several blocks in flight), the new front-end test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp FUZZ_VERBOSE=1
O=gpurun_out/r5_call13; mkdir -p $O
{
echo "== qf_rank alphabet sizes"; timeout 900 python -m pytest tests/test_gpu_compress.py -q -m gpu -k "front_end_rank_paths" 2>&1 | tail -3
for s in 701 702 703; do (timeout 500 python tools/fuzz_gpu.py 220 $s $((12<<20)) > $O/fuzz$s.log 2>&1; echo "seed $s exit $?" >> $O/campaign.txt) & done
(timeout 500 python tools/pipe_stress.py 200 41 4 $((20<<20)) > $O/pipe41.log 2>&1; echo "pipe exit $?" >> $O/campaign.txt) &
wait
cat $O/campaign.txt; for s in 701 702 703; do tail -1 $O/fuzz$s.log | cut -c1-200; done; tail -1 $O/pipe41.log | cut -c1-300
} > $O/out.txt 2>&1
cut -c1-300 $O/out.txt | tail -20
