#!/bin/bash
# round 5, GPU call 35: three more runs at the driver's settings on the final build (what is left of the budget)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5_call35; mkdir -p $O
for i in 1 2 3; do
timeout 30 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_final_$i.json 2> $O/err_$i.txt
python -c "
import json; d=json.loads(open('$O/bench_20_final_$i.json').read().strip().splitlines()[-1]); h=d['host']; print({k:d[k] for k in ('value','ms_per_step','verified')}, h.get('cpu_seconds_per_block_rank0'), h.get('blocks_by_coder_task_shape_rank0'))" >> $O/out.txt 2>&1
done
cat $O/out.txt
