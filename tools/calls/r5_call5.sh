#!/bin/bash
# round 5, GPU call 5: why the C job bench trails bench.py (pageable input? the in-order collector?); task-shape policy at 20 steps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call5; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'frac', d['roofline']['frac'], 'cpu_s/blk', d['per_rank'][0]['cpu_seconds_per_block'], d['host']['blocks_by_coder_task_shape_rank0'], d.get('boundary'))
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
jl() { python -c "import json;d=json.load(open('$1'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes')})" 2>&1 | tail -1; tail -3 ${1%.json}.err; }
J=libbsc_amd/lib/job_bench
{
for opt in "" "--pin-input" "--upfront" "--pin-input --upfront"; do
  echo "== job_bench 160 $opt"; timeout 300 $J --steps 160 $opt > $O/j.json 2> $O/j.err; jl $O/j.json
done
for opt in "" "--pin-input --upfront"; do
  echo "== job_bench 20 $opt"; timeout 300 $J --steps 20 --warmup 5 $opt > $O/j.json 2> $O/j.err; jl $O/j.json
done
echo "== bench 160"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160.json 2> $O/b160.err; line $O/b160.json
for i in 1 2 3; do
echo "== bench 20 default #$i"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_$i.json 2> $O/b20_$i.err; line $O/b20_$i.json
echo "== bench 20 BSC_RC_ADAPTIVE=0 #$i"; BSC_RC_ADAPTIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_na_$i.json 2> $O/b20_na_$i.err; line $O/b20_na_$i.json
done
echo "== bench 160 BSC_RC_ADAPTIVE=0"; BSC_RC_ADAPTIVE=0 timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_na.json 2> $O/b160_na.err; line $O/b160_na.json
echo "== bench 20 trace"; BSC_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_trace.json 2> $O/b20_trace.txt; line $O/b20_trace.json
} > $O/out.txt 2>&1
cut -c1-900 $O/out.txt | tail -60
