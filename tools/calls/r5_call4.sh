#!/bin/bash
# round 5, GPU call 4: qf_rank with a 256-run halo behind the lifted tile against the tile-only version; the C job bench with in-order
# collection through bscgpu_pipe_peek; the whole GPU suite on the current head
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call4; mkdir -p $O
V=$PWD/libbsc_amd/lib/variants
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get('kernels', {})
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'frac', d['roofline']['frac'], {n: k[n]['ms_per_block'] for n in ('gather', 'seg', 'misc')}, 'cpu_s/blk', d['per_rank'][0]['cpu_seconds_per_block'], d['host']['blocks_by_coder_task_shape_rank0'])
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
jl() { python -c "import json;d=json.load(open('$1'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes','create_s','setup_s')})" 2>&1 | tail -1; tail -3 ${1%.json}.err; }
{
echo "== front end, tile-only lifting (round 4)"; BSC_LIB_OVERRIDE=$V/libbsc_qfold.so timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== front end, tile + halo"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== job_bench 20"; timeout 300 libbsc_amd/lib/job_bench --steps 20 --warmup 5 > $O/job20.json 2> $O/job20.err; jl $O/job20.json
echo "== bench 20"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20.json 2> $O/b20.err; line $O/b20.json
echo "== job_bench 160"; timeout 300 libbsc_amd/lib/job_bench --steps 160 > $O/job160.json 2> $O/job160.err; jl $O/job160.json
echo "== bench 160"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160.json 2> $O/b160.err; line $O/b160.json
echo "== bench 160, old qf_rank"; BSC_LIB_OVERRIDE=$V/libbsc_qfold.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_qfold.json 2> $O/b160_qfold.err; line $O/b160_qfold.json
echo "== job_bench 20 again"; timeout 300 libbsc_amd/lib/job_bench --steps 20 --warmup 5 > $O/job20b.json 2> $O/job20b.err; jl $O/job20b.json
echo "== job_bench 160 -e0"; timeout 300 libbsc_amd/lib/job_bench --steps 160 --coder 3 > $O/job160e0.json 2> $O/job160e0.err; jl $O/job160e0.json
echo "== full GPU suite"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
} > $O/out.txt 2>&1
cut -c1-700 $O/out.txt | tail -60
