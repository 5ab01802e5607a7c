#!/bin/bash
# round 5, GPU call 2: decision-lane p stream with 2 / 4 / 8 entries per lane and trip, with and without XCD-contiguous run ranges
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call2; mkdir -p $O
V=$PWD/libbsc_amd/lib/variants
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
echo "== run-lane"; BSC_DC_PSTREAM=run timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== u4 (default build)"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
for v in u2 u8 u4x u8x; do echo "== $v"; BSC_LIB_OVERRIDE=$V/libbsc_$v.so timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300; done
echo "== run-lane + xcd ranges"; BSC_DC_PSTREAM=run BSC_LIB_OVERRIDE=$V/libbsc_u4x.so timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== parity (u4)"; timeout 500 python -m pytest tests -x -q -m gpu -k "device_static_model or fast_coder_on_the_device or lzp_blocks_take or eight_sub_block" 2>&1 | tail -3
echo "== bench 160 run-lane"; BSC_DC_PSTREAM=run timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_run.json 2> $O/b160_run.err; line $O/b160_run.json
echo "== bench 160 u4"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_u4.json 2> $O/b160_u4.err; line $O/b160_u4.json
echo "== bench 160 u8"; BSC_LIB_OVERRIDE=$V/libbsc_u8.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_u8.json 2> $O/b160_u8.err; line $O/b160_u8.json
echo "== bench 160 u4x"; BSC_LIB_OVERRIDE=$V/libbsc_u4x.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_u4x.json 2> $O/b160_u4x.err; line $O/b160_u4x.json
echo "== bench 160 u8x"; BSC_LIB_OVERRIDE=$V/libbsc_u8x.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_u8x.json 2> $O/b160_u8x.err; line $O/b160_u8x.json
echo "== bench 160 run-lane + xcd"; BSC_DC_PSTREAM=run BSC_LIB_OVERRIDE=$V/libbsc_u4x.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_runx.json 2> $O/b160_runx.err; line $O/b160_runx.json
echo "== bench 160 run-lane again"; BSC_DC_PSTREAM=run timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b160_run2.json 2> $O/b160_run2.err; line $O/b160_run2.json
} > $O/out.txt 2>&1
cut -c1-700 $O/out.txt | tail -60
