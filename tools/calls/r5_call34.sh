#!/bin/bash
# round 5, GPU call 34: the build with the reworked host range coders — bench at the default and at the driver's settings (twice, alternating with the round-4
# eight-lane step), the C job bench, smoke, the whole GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call34; mkdir -p $O
show() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); h=d.get('host',{}); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['frac'], 'cpu_s/blk', h.get('cpu_seconds_per_block_rank0'), h.get('blocks_by_coder_task_shape_rank0'))"; }
{
echo "== python bench.py"; timeout 200 python bench.py > $O/bench_default_host_coder_build.json 2> $O/bench_default.err; show $O/bench_default_host_coder_build.json
for i in 1 2; do
echo "== python bench.py --gpus 1 --steps 20 --warmup 5 #$i"; timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_host_coder_build_$i.json 2> $O/bench_20.err; show $O/bench_20_host_coder_build_$i.json
echo "== the same with BSC_RC_VSEL=0 BSC_RC_PREFETCH=0 (round 4's eight-lane step) #$i"; BSC_RC_VSEL=0 BSC_RC_PREFETCH=0 timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_round4_step_$i.json 2> $O/bench_20.err; show $O/bench_20_round4_step_$i.json
done
echo "== job_bench"; timeout 100 libbsc_amd/lib/job_bench > $O/job_bench_320_host_coder_build.json 2>/dev/null; python -c "import json;d=json.load(open('$O/job_bench_320_host_coder_build.json'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block')})"
echo "== smoke"; timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== whole GPU suite"; timeout 225 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-400 | tail -24
