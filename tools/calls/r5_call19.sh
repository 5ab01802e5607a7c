#!/bin/bash
# round 5, GPU call 19: the final build once more — whole GPU suite, the sanitizer build on both GPU test files, a short fuzz, bench at the driver's settings and the default
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp FUZZ_VERBOSE=1
O=gpurun_out/r5_call19; mkdir -p $O
{
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== whole GPU suite"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== sanitizer build: both GPU test files"; timeout 1800 python tools/asan_run.py python -m pytest tests/test_gpu_compress.py tests/test_gpu_device.py -q -m gpu 2>&1 | tail -4 | cut -c1-300
for s in 801 802; do (timeout 300 python tools/fuzz_gpu.py 120 $s $((12<<20)) > $O/fuzz$s.log 2>&1; echo "seed $s exit $?" >> $O/campaign.txt) & done; wait
cat $O/campaign.txt; for s in 801 802; do tail -1 $O/fuzz$s.log | cut -c1-200; done
echo "== python bench.py --gpus 1 --steps 20 --warmup 5"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_final_build.json 2> $O/bench_20.err; python -c "
import json; d=json.loads(open('$O/bench_20_final_build.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['frac'], d['cpu_baseline']['value'], {n:v['ms_per_block'] for n,v in d['kernels'].items()})"
echo "== python bench.py"; timeout 600 python bench.py --no-cpu-baseline > $O/bench_default_final_build.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default_final_build.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['frac'], {n:v['ms_per_block'] for n,v in d['kernels'].items()})"
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-400 | tail -30
