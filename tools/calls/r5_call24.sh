#!/bin/bash
# round 5, GPU call 24: as call 21, plus the state family's positions by wide loads in the p stream, and the text rounds' window in four words
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call24; mkdir -p $O
V=$PWD/libbsc_amd/lib/variants
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get('kernels', {})
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'dc', {n: k[n]['ms_per_block'] for n in k if n.startswith('dc_') or n == 'gather'})
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
{
for rep in 1 2; do
echo "== round 4's partition"; BSC_LIB_OVERRIDE=$V/libbsc_dcold.so timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== new partition"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
done
echo "== parity"; timeout 900 python -m pytest tests -x -q -m gpu -k "device_static_model or fast_coder_on_the_device or lzp_blocks_take or eight_sub_block or golden_fixtures or sub_block_count or front_end_rank or full_size_64m or matches_reference" 2>&1 | tail -3
echo "== bench 160 old"; BSC_LIB_OVERRIDE=$V/libbsc_dcold.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b_old.json 2> $O/b_old.err; line $O/b_old.json
echo "== bench 160 new"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b_new.json 2> $O/b_new.err; line $O/b_new.json
echo "== bench 160 old"; BSC_LIB_OVERRIDE=$V/libbsc_dcold.so timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b_old.json 2> $O/b_old.err; line $O/b_old.json
echo "== bench 160 new"; timeout 300 python bench.py --steps 160 --no-cpu-baseline > $O/b_new.json 2> $O/b_new.err; line $O/b_new.json
} > $O/out.txt 2>&1
cut -c1-400 $O/out.txt | tail -30
