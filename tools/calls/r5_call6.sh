#!/bin/bash
# round 5, GPU call 6: does the CPU-time quota stop the GPU-driving threads?  coder threads 12..24 at 20 and 160 steps (idle test off), cgroup throttling counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call6; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d['host']
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'cpu_s/blk', h['cpu_seconds_per_block_rank0'], 'throttled ms', h['cgroup_throttled_ms_in_timed_region'], 'periods', h['cgroup_throttled_periods_in_timed_region'], h['blocks_by_coder_task_shape_rank0'])
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
{
cat /sys/fs/cgroup/cpu.max; nproc
for rep in 1 2; do
for th in 24 16 13 20; do
echo "== bench 20, threads $th (rep $rep)"; BSCGPU_HOST_THREADS=$th BSCGPU_HOST_CPUS=16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
done
done
for th in 24 16 13; do
echo "== bench 160, threads $th"; BSCGPU_HOST_THREADS=$th BSCGPU_HOST_CPUS=16 timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
done
echo "== bench 20, threads 24, idle test on"; BSC_RC_ADAPTIVE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 20, threads 16, idle test on"; BSC_RC_ADAPTIVE=1 BSCGPU_HOST_THREADS=16 BSCGPU_HOST_CPUS=16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
} > $O/out.txt 2>&1
cut -c1-500 $O/out.txt | tail -60
