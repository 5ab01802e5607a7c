#!/bin/bash
# round 5, GPU call 11: the final build — whole GPU suite, smoke, the sanitizer build on tests/test_gpu_compress.py (child executables without the preload), bench at the driver's settings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call11; mkdir -p $O
{
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== whole GPU suite"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo "== sanitizer build: tests/test_gpu_compress.py"; timeout 1500 python tools/asan_run.py python -m pytest tests/test_gpu_compress.py -q -m gpu 2>&1 | tail -6 | cut -c1-300
echo "== python bench.py --gpus 1 --steps 20 --warmup 5"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_final_build.json 2> $O/bench_20.err; python -c "
import json; d=json.loads(open('$O/bench_20_final_build.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['frac'], d['cpu_baseline']['value'])"
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-300 | tail -30
