#!/bin/bash
# round 5, GPU call 28: the host half of the static coder on the box's own CPU (tools/rc_host_bench.cpp): task shapes, software prefetch
# distances, landing zone against malloc memory, six tasks at once; the same program built against the previous commit's coder
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call28; mkdir -p $O
{
grep -m1 "model name" /proc/cpuinfo; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
echo "=== this build"; timeout 150 libbsc_amd/lib/rc_host_bench
echo "=== the previous commit's coder"; timeout 100 libbsc_amd/lib/rc_host_bench_old
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-200
