#!/bin/bash
which perf && perf --version
cat /proc/sys/kernel/perf_event_paranoid
export BSC_QLFC_PIPELINE=0
if which perf > /dev/null; then
  perf stat -e cycles,instructions,branches,branch-misses,L1-dcache-load-misses,cache-misses -- python tools/host_coder_only.py 2>&1 | tail -25
else
  echo "no perf"
fi
