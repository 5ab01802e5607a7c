#!/bin/bash
# round 3, GPU call 24: contexts per GPU at the driver's 20 steps; the N = 2 path on one GPU (gloo) after this round's bench.py changes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3_final; mkdir -p $O
{
for cfg in "4 4" "6 3" "5 3" "6 2" "8 2"; do set -- $cfg
  echo "== 20 steps, $1 contexts x $2"; for rep in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --contexts $1 --depth $2 2>/dev/null | cut -c1-160; done
done
echo "== N = 2 on one GPU (gloo)"; BSC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 24 --warmup 4 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; cut -c1-200 $O/bench_n2_gloo.json; tail -2 $O/bench_n2_gloo.err | cut -c1-200
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3_final/bench_n2_gloo.json")); print("verified", d["verified"], [ (r["rank"], r["verified"], r["pcie_d2h_MBps"]) for r in d["per_rank"] ])
except Exception as e: print("unreadable", e)
PY
} > gpurun_out/r3_call24.txt 2>&1
cat gpurun_out/r3_call24.txt | cut -c1-220
