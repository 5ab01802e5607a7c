#!/bin/bash
# round 5, GPU call 10: coding slots (at most budget - 1 tasks coding at once) against none, 20 and 24 coder threads, at the driver's 20 steps; cgroup throttling
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call10; mkdir -p $O
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
for rep in 1 2 3; do
echo "== bench 20: slots 15 (default), 20 threads"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 20: no slot limit, 20 threads"; BSCGPU_HOST_CODING=99 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 20: slots 15, 28 threads"; BSCGPU_HOST_THREADS=28 BSCGPU_HOST_CPUS=16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 20: slots 15, 20 threads, last 3 blocks low-latency"; BSC_BENCH_LL=3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 20: slots 15, 20 threads, last 2 blocks low-latency"; BSC_BENCH_LL=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 20: no slot limit, last 3 blocks low-latency"; BSCGPU_HOST_CODING=99 BSC_BENCH_LL=3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
done
echo "== bench 160: slots 15, 20 threads"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== bench 160: slots 15, 28 threads"; BSCGPU_HOST_THREADS=28 BSCGPU_HOST_CPUS=16 timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b.json 2> $O/b.err; line $O/b.json
echo "== job_bench 20 x2"; for i in 1 2; do libbsc_amd/lib/job_bench --steps 20 --warmup 5 | python -c "import json,sys;d=json.load(sys.stdin);print({k:d[k] for k in ('value','verified','cpu_seconds_per_block')})"; done
} > $O/out.txt 2>&1
cut -c1-400 $O/out.txt | tail -60
