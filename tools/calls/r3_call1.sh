#!/bin/bash
# round 3, GPU call 1: single-read digit passes — parity first, then A/B timing against the three-kernel pass on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests: single-read passes"; timeout 900 python -m pytest tests/test_gpu_device.py -x -q -k "single_read or radix_sort_matches" 2>&1 | tail -15
echo "== per pass, three-kernel (BSC_RS_ONESWEEP=0)"; BSC_RS_ONESWEEP=0 timeout 300 python tools/per_pass.py 2>&1 | tail -3
echo "== per pass, single-read (BSC_RS_ONESWEEP=1)"; BSC_RS_ONESWEEP=1 timeout 300 python tools/per_pass.py 2>&1 | tail -3
echo "== bwt profile, three-kernel"; BSC_RS_ONESWEEP=0 timeout 300 python tools/perf_bwt.py 2>&1 | grep -v "^st" | tail -16
echo "== bwt profile, single-read"; BSC_RS_ONESWEEP=1 timeout 300 python tools/perf_bwt.py 2>&1 | grep -v "^st" | tail -16
echo "== golden 64 MiB block + bwt parity through the new path"; timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_device.py -x -q -k "full_size_64m_block_golden or bwt_matches_reference or bwt_device_resident_16m" 2>&1 | tail -5
} > gpurun_out/r3_call1.txt 2>&1
tail -60 gpurun_out/r3_call1.txt
