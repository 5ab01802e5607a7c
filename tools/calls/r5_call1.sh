#!/bin/bash
# round 5, GPU call 1: baseline of the round-4 head on this box; the decision-lane p-stream kernel against the run-lane one (per-kernel
# times, parity, whole job at 20 and 160 steps); p-stream D2H on SDMA against blit kernels; XCD-contiguous run ranges under the pipelined load
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call1; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get('kernels', {})
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'frac', d['roofline']['frac'], 'dc', {n: k[n]['ms_per_block'] for n in k if n.startswith('dc_')}, 'cpu_s/blk', d['per_rank'][0]['cpu_seconds_per_block'], d['host']['blocks_by_coder_task_shape_rank0'])
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
{
env | grep -i "sdma\|HSA_\|ROC\|HIP_" | head
echo "== device model per kernel class: run-lane p stream"; BSC_DC_PSTREAM=run timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== device model per kernel class: decision-lane p stream"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== parity (decision-lane)"; timeout 500 python -m pytest tests -x -q -m gpu -k "device_static_model or fast_coder_on_the_device or lzp_blocks_take or eight_sub_block or golden_fixtures or sub_block_count" 2>&1 | tail -4
echo "== bench 20, decision-lane"; timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_20_dl.json 2> $O/bench_20_dl.err; line $O/bench_20_dl.json
echo "== bench 20, run-lane"; BSC_DC_PSTREAM=run timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_run.json 2> $O/bench_20_run.err; line $O/bench_20_run.json
echo "== bench 160, decision-lane"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/bench_160_dl.json 2> $O/bench_160_dl.err; line $O/bench_160_dl.json
echo "== bench 160, run-lane"; BSC_DC_PSTREAM=run timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/bench_160_run.json 2> $O/bench_160_run.err; line $O/bench_160_run.json
echo "== bench 160, decision-lane, HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/bench_160_sdma0.json 2> $O/bench_160_sdma0.err; line $O/bench_160_sdma0.json
echo "== bench 160, decision-lane, HSA_ENABLE_SDMA=1"; HSA_ENABLE_SDMA=1 timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/bench_160_sdma1.json 2> $O/bench_160_sdma1.err; line $O/bench_160_sdma1.json
echo "== bench 160, decision-lane, DC_XCD_RANGES=1"; BSC_LIB_OVERRIDE=$PWD/libbsc_amd/lib/variants/libbsc_xcd.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/bench_160_xcd.json 2> $O/bench_160_xcd.err; line $O/bench_160_xcd.json
echo "== bench 20, decision-lane again"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_dl2.json 2> $O/bench_20_dl2.err; line $O/bench_20_dl2.json
} > $O/out.txt 2>&1
cut -c1-700 $O/out.txt | tail -60
