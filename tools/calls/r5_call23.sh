#!/bin/bash
# round 5, GPU call 23: the new large-block test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compress.py -q -m gpu -k "large_odd_sized or front_end_rank or c_job_bench" 2>&1 | tail -5
