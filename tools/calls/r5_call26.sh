#!/bin/bash
# round 5, GPU call 26: the last build — whole GPU suite, smoke, bench at the driver's settings and the default, the C job bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call26; mkdir -p $O
{
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== whole GPU suite"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== python bench.py --gpus 1 --steps 20 --warmup 5"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_last_build.json 2> $O/bench_20.err; python -c "
import json; d=json.loads(open('$O/bench_20_last_build.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['frac'], d['cpu_baseline']['value'], {n:v['ms_per_block'] for n,v in d['kernels'].items()})"
echo "== python bench.py"; timeout 600 python bench.py --no-cpu-baseline > $O/bench_default_last_build.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default_last_build.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['frac'], {n:v['ms_per_block'] for n,v in d['kernels'].items()})"
echo "== job_bench"; timeout 300 libbsc_amd/lib/job_bench > $O/job_bench_320_last_build.json 2>/dev/null; python -c "import json;d=json.load(open('$O/job_bench_320_last_build.json'));print({k:d[k] for k in ('value','ms_per_step','verified')})"
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-400 | tail -20
