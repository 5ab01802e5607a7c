#!/bin/bash
# round 3, GPU call 4: scout-wave digit pass — parity, then timing against the three-kernel pass and the ablations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
echo "== tests: single-read passes"; timeout 900 python -m pytest tests/test_gpu_device.py -x -q -k "single_read or radix_sort_matches" 2>&1 | tail -8
timeout 1200 python tools/os_ab.py default:BSC_RS_ONESWEEP=0 default $V/libbsc_os_abl1.so 2>&1 | tail -8
echo "== phase stamps"; BSC_LIB_OVERRIDE=$V/libbsc_os_ph.so BSC_RS_ONESWEEP=1 timeout 300 python tools/os_phase_timing.py 2>&1 | tail -24
echo "== golden + bwt parity"; timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_device.py -x -q -k "full_size_64m_block_golden or bwt_matches_reference or bwt_device_resident_16m" 2>&1 | tail -3
echo "== bwt profile"; timeout 300 python tools/perf_bwt.py 2>&1 | grep -v "^st" | tail -14
} > gpurun_out/r3_call4.txt 2>&1
cat gpurun_out/r3_call4.txt
