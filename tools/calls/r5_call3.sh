#!/bin/bash
# round 5, GPU call 3: p stream in rounds of eight decisions against the serial tail; the C job driver's bench (tools/job_bench.cpp) against bench.py;
# new tests (landing-zone fallback, job_bench, job driver)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call3; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get('kernels', {})
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'frac', d['roofline']['frac'], 'dc', {n: k[n]['ms_per_block'] for n in k if n.startswith('dc_')}, 'cpu_s/blk', d['per_rank'][0]['cpu_seconds_per_block'], d['host']['blocks_by_coder_task_shape_rank0'], d['per_rank'][0]['pcie_d2h_MB_per_block'])
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
{
echo "== tail"; BSC_DC_PSTREAM=tail timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== rounds"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== tests"; timeout 900 python -m pytest tests -x -q -m gpu -k "device_static_model or fast_coder_on_the_device or lzp_blocks_take or eight_sub_block or landing_zones or job_bench or job_driver or cxx_multi or golden_fixtures" 2>&1 | tail -4
echo "== bench 20 rounds"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20.json 2> $O/b20.err; line $O/b20.json
echo "== bench 20 tail"; BSC_DC_PSTREAM=tail timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_tail.json 2> $O/b20_tail.err; line $O/b20_tail.json
echo "== job_bench 20"; timeout 300 libbsc_amd/lib/job_bench --steps 20 --warmup 5 > $O/job20.json 2> $O/job20.err; cut -c1-400 $O/job20.json; python -c "import json;d=json.load(open('$O/job20.json'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes','create_s','setup_s')})"
echo "== bench 160 rounds"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160.json 2> $O/b160.err; line $O/b160.json
echo "== bench 160 tail"; BSC_DC_PSTREAM=tail timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_tail.json 2> $O/b160_tail.err; line $O/b160_tail.json
echo "== job_bench 160"; timeout 300 libbsc_amd/lib/job_bench --steps 160 > $O/job160.json 2> $O/job160.err; python -c "import json;d=json.load(open('$O/job160.json'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes','create_s','setup_s')})"
echo "== job_bench 20 again"; timeout 300 libbsc_amd/lib/job_bench --steps 20 --warmup 5 > $O/job20b.json 2> $O/job20b.err; python -c "import json;d=json.load(open('$O/job20b.json'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes','create_s','setup_s')})"
echo "== bench 20 rounds again"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20b.json 2> $O/b20b.err; line $O/b20b.json
} > $O/out.txt 2>&1
cut -c1-700 $O/out.txt | tail -60
