#!/bin/bash
# round 3, GPU call 37: bench.py after its robustness changes (thread errors propagate, fewer contexts when the set-up fails)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 50 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/b37.err | cut -c1-300 > gpurun_out/r3_call37.txt; tail -3 gpurun_out/b37.err >> gpurun_out/r3_call37.txt
cat gpurun_out/r3_call37.txt
