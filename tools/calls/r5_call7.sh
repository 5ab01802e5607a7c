#!/bin/bash
# round 5, GPU call 7: the C++ file compressor by contexts per GPU (round 4: more contexts were slower), where its set-up time goes; bench.py and
# job_bench with the new defaults (20 coder threads, idle test off)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call7; mkdir -p $O
wall() { local s=$(date +%s%N); "$@" 2>&1 | tail -1 | sed 's/.*\(compressed\|encoded\)/\1/' | cut -c1-200; local e=$(date +%s%N); echo "   process wall $(( (e - s) / 1000000 )) ms"; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d['host']
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'frac', d['roofline']['frac'], 'cpu_s/blk', h['cpu_seconds_per_block_rank0'], 'throttled', h['cgroup_throttled_periods_in_timed_region'], h['blocks_by_coder_task_shape_rank0'])
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
jl() { python -c "import json;d=json.load(open('$1'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes')})" 2>&1 | tail -1; tail -3 ${1%.json}.err; }
{
python - <<'PY'
import sys
sys.path.insert(0, '.')
from libbsc_amd import api
with open('/dev/shm/mgpu_in.bin', 'wb') as f:
    eight = [api.synth_text_v1(seed, 64 << 20) for seed in range(10, 18)]
    for b in range(32): eight[b % 8].tofile(f)
PY
echo "== 32 x 64 MiB through bsc_mgpu"
for flags in "-b64 -p -e1 -C2 -D3" "-b64 -p -e1 -C3 -D3" "-b64 -p -e1 -C6 -D3" "-b64 -p -e1 -C4 -D3"; do
  echo "-- bsc_mgpu $flags"; for rep in 1 2 3; do wall ./libbsc_amd/lib/bsc_mgpu e /dev/shm/mgpu_in.bin /dev/shm/mgpu_out.bsc $flags; done
done
echo "== set-up timing, -C6"; BSCGPU_TIMING=1 ./libbsc_amd/lib/bsc_mgpu e /dev/shm/mgpu_in.bin /dev/shm/mgpu_out.bsc -b64 -p -e1 -C6 -D3 2>&1 | grep -v "pinned landing" | sort -t' ' -k1,1 | tail -40 | cut -c1-160
export OMP_NUM_THREADS=8 OMP_WAIT_POLICY=passive
echo "-- the reference CLI relinked against this library, 8 OpenMP threads"; for rep in 1 2 3; do wall ./oracle/_ref/bsc_mi355x e /dev/shm/mgpu_in.bin /dev/shm/mgpu_out2.bsc -b64 -p -e1; done
unset OMP_NUM_THREADS
OMP_NUM_THREADS=16 ./oracle/_ref/bsc d /dev/shm/mgpu_out.bsc /dev/shm/mgpu_back.bin > /dev/null 2>&1; cmp /dev/shm/mgpu_in.bin /dev/shm/mgpu_back.bin && echo "bsc_mgpu's last file unpacked by the reference bsc d: identical"
rm -f /dev/shm/mgpu_*
for i in 1 2 3; do
echo "== bench 20 #$i"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_$i.json 2> $O/b20_$i.err; line $O/b20_$i.json
echo "== job_bench 20 #$i"; timeout 300 libbsc_amd/lib/job_bench --steps 20 --warmup 5 > $O/job20_$i.json 2> $O/job20_$i.err; jl $O/job20_$i.json
done
echo "== bench 320"; timeout 300 python bench.py --no-cpu-baseline > $O/b320.json 2> $O/b320.err; line $O/b320.json
echo "== job_bench 320"; timeout 300 libbsc_amd/lib/job_bench > $O/job320.json 2> $O/job320.err; jl $O/job320.json
} > $O/out.txt 2>&1
grep -v amdgpu.ids $O/out.txt | cut -c1-400 | tail -80
