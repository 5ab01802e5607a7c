#!/bin/bash
# round 3, GPU call 14: the chosen digit pass through the whole job — the driver's bench command, the default bench, the CLI, the config table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3_final; mkdir -p $O
{
echo "== tests: single-read passes + golden"; timeout 900 python -m pytest tests/test_gpu_device.py tests/test_gpu_compress.py -x -q -k "single_read or golden or sub_block_cuts or arena" 2>&1 | tail -4
echo "== bench, the driver's command"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; cat $O/bench_20.json
echo "== bench, defaults"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
echo "== bench, three-kernel digit pass (BSC_RS_ONESWEEP=0), the driver's command"; BSC_RS_ONESWEEP=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_three_kernel.json 2>/dev/null; cat $O/bench_20_three_kernel.json
echo "== CLI"; timeout 600 python tools/cli_bench.py 2>&1 | tail -8 | tee $O/cli_bench.txt
echo "== config table"; timeout 900 python tools/config_table.py > $O/config_table.txt 2>&1; cat $O/config_table.txt
} > gpurun_out/r3_call14.txt 2>&1
cat gpurun_out/r3_call14.txt
