#!/bin/bash
# round 3, GPU call 5: digit pass with clean steady loop; device coder with XCD-contiguous run ranges (A/B); full GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
echo "== tests: single-read passes"; timeout 900 python -m pytest tests/test_gpu_device.py -x -q -k "single_read or radix_sort_matches" 2>&1 | tail -4
timeout 1200 python tools/os_ab.py default:BSC_RS_ONESWEEP=0 default $V/libbsc_os_abl1.so 2>&1 | tail -8
echo "== phase stamps"; BSC_LIB_OVERRIDE=$V/libbsc_os_ph.so BSC_RS_ONESWEEP=1 timeout 300 python tools/os_phase_timing.py 2>&1 | tail -24
echo "== bwt profile"; timeout 300 python tools/perf_bwt.py 2>&1 | grep -v "^st" | tail -12
echo "== device coder, XCD ranges (default)"; timeout 300 python tools/devcoder_time.py 2>&1 | tail -2
echo "== device coder, interleaved blocks"; BSC_LIB_OVERRIDE=$V/libbsc_dc_noxcd.so timeout 300 python tools/devcoder_time.py 2>&1 | tail -2
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
} > gpurun_out/r3_call5.txt 2>&1
cat gpurun_out/r3_call5.txt
