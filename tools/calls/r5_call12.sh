#!/bin/bash
# round 5, GPU call 12: the driver's setting (--steps 20 --warmup 5) eight times per configuration, interleaved, one box: which defaults, if any, are worth changing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call12; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > $O/b.json 2> $O/b.err
  python - "$name" <<'PY' >> gpurun_out/r5_call12/values.txt
import json, sys
try:
    d = json.loads(open('gpurun_out/r5_call12/b.json').read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], d['host']['cpu_seconds_per_block_rank0'], d['host']['cgroup_throttled_periods_in_timed_region'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
rm -f $O/values.txt
for rep in 1 2 3 4 5 6 7 8; do
  EXTRA="" run default X=1
  EXTRA="" run idle_test_on BSC_RC_ADAPTIVE=1
  EXTRA="" run ll3 BSC_BENCH_LL=3
  EXTRA="--contexts 5" run ctx5x3 X=1
  EXTRA="--contexts 4 --depth 4" run ctx4x4 X=1
  EXTRA="" run threads24 BSCGPU_HOST_THREADS=24 BSCGPU_HOST_CPUS=16
done
python - <<'PY'
import collections, statistics
v = collections.defaultdict(list)
for line in open('gpurun_out/r5_call12/values.txt'):
    p = line.split()
    if p[1] != 'FAILED': v[p[0]].append(float(p[1]))
for k, x in v.items():
    print(f"{k:14s} n={len(x)} mean {statistics.mean(x):7.1f} median {statistics.median(x):7.1f} min {min(x):7.1f} max {max(x):7.1f}  " + " ".join(f"{a:.0f}" for a in x))
PY
