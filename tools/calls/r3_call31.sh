#!/bin/bash
# round 3, GPU call 31: does the alignment of the arena's buffers explain why a digit pass inside the BWT is slower than the same sort on
# torch buffers (tools/os_ab.py)?  kA / kB / vA / vB sit 512 B past a 4 KiB boundary today.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
V=$(pwd)/libbsc_amd/lib/variants
{
for rnd in 1 2; do
  for lib in default $V/libbsc_arena4k.so $V/libbsc_arena2m.so; do
    if [ "$lib" = default ]; then unset BSC_LIB_OVERRIDE; else export BSC_LIB_OVERRIDE=$lib; fi
    echo "-- $lib"; timeout 300 python tools/perf_bwt.py 2>&1 | grep "scatter n=\|profiled"
  done
done
unset BSC_LIB_OVERRIDE
timeout 600 python tools/os_ab.py default 2>&1 | tail -1
} > gpurun_out/r3_call31.txt 2>&1
cat gpurun_out/r3_call31.txt | cut -c1-250
