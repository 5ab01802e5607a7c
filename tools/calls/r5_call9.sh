#!/bin/bash
# round 5, GPU call 9: the round's evidence on one box — rocprofv3 kernel trace of bench.py + FETCH / WRITE PMC passes over one block, SQ and L2 counters of the
# same block (digit pass: wait / issue / LDS-conflict / L2 hit figures), bench lines (default 320 steps, the driver's 20 steps, the other coders and config 5, the
# C job bench), input classes of the sorter, BASELINE configs against the reference, 8 ranks over gloo on the one GPU with an eighth of the CPUs each, the sanitizer
# build on tests/test_gpu_compress.py (which test was it that failed in round 4?), the whole GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_final; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d['host']; r = d['roofline']
    print({x: d[x] for x in ('value', 'ms_per_step', 'verified')}, 'frac', r['frac'], 'sort_frac', r.get('sort_frac'), 'ceiling', (r.get('pattern_ceiling') or {}).get('frac_of_peak'), 'cpu_s/blk', h['cpu_seconds_per_block_rank0'], h['blocks_by_coder_task_shape_rank0'])
except Exception as e:
    print('no line:', e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
}
jl() { python -c "import json;d=json.load(open('$1'));print({k:d[k] for k in ('value','ms_per_step','verified','cpu_seconds_per_block','coder_task_shapes','setup_s')})" 2>&1 | tail -1; tail -2 ${1%.json}.err; }
{
echo "== bench default (320 steps, with cpu_baseline)"; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; line $O/bench_default.json
for i in 1 2 3; do echo "== bench --steps 20 --warmup 5 #$i"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_$i.json 2> $O/bench_20_$i.err; line $O/bench_20_$i.json; done
echo "== job_bench 20 / 320"; timeout 300 libbsc_amd/lib/job_bench --steps 20 --warmup 5 > $O/job_bench_20.json 2> $O/job_bench_20.err; jl $O/job_bench_20.json
timeout 300 libbsc_amd/lib/job_bench > $O/job_bench_320.json 2> $O/job_bench_320.err; jl $O/job_bench_320.json
echo "== bench 20 with the timeline"; BSC_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_trace.json 2> $O/bench_20_timeline.txt; line $O/bench_20_trace.json
echo "== -e0 (fast coder), 320 steps"; timeout 300 python bench.py --coder 3 --no-cpu-baseline > $O/bench_e0.json 2> $O/bench_e0.err; line $O/bench_e0.json
echo "== -e2 (adaptive coder, host model), 96 steps"; timeout 600 python bench.py --coder 2 --steps 96 --no-cpu-baseline > $O/bench_e2.json 2> $O/bench_e2.err; line $O/bench_e2.json
echo "== config 5: ST5 / ST6 on 128 MiB blocks, 48 steps"; timeout 600 python bench.py --sorter 5 --block 134217728 --steps 48 --no-cpu-baseline > $O/bench_config5_st5.json 2> $O/bench_config5_st5.err; line $O/bench_config5_st5.json
timeout 600 python bench.py --sorter 6 --block 134217728 --steps 48 --no-cpu-baseline > $O/bench_config5_st6.json 2> $O/bench_config5_st6.err; line $O/bench_config5_st6.json
echo "== job_bench config 5 ST5, -e0"; timeout 300 libbsc_amd/lib/job_bench --sorter 5 --block 134217728 --seed 3 --steps 48 --contexts 4 > $O/job_bench_st5.json 2> $O/job_bench_st5.err; jl $O/job_bench_st5.json
timeout 300 libbsc_amd/lib/job_bench --coder 3 > $O/job_bench_e0.json 2> $O/job_bench_e0.err; jl $O/job_bench_e0.json
echo "== profile_round"; timeout 1200 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
P=$(pwd)/gpurun_out/prof_r05
echo "== SQ counters, one block"
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $P/pmc_sq -o s -- python tools/pmc_one_block.py > $P/pmc_sq.log 2>&1
python tools/pmc_table.py $P/pmc_sq > $O/pmc_sq_one_block.txt 2>&1; head -12 $O/pmc_sq_one_block.txt | cut -c1-230
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $P/pmc_sq2 -o s -- python tools/pmc_one_block.py > $P/pmc_sq2.log 2>&1
python tools/pmc_table.py $P/pmc_sq2 > $O/pmc_sq2_one_block.txt 2>&1; head -6 $O/pmc_sq2_one_block.txt | cut -c1-230
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $P/pmc_l2 -o l -- python tools/pmc_one_block.py > $P/pmc_l2.log 2>&1
python tools/pmc_table.py $P/pmc_l2 > $O/pmc_l2_one_block.txt 2>&1; head -6 $O/pmc_l2_one_block.txt | cut -c1-160
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum -d $P/pmc_ea -o l -- python tools/pmc_one_block.py > $P/pmc_ea.log 2>&1
python tools/pmc_table.py $P/pmc_ea > $O/pmc_ea_one_block.txt 2>&1; head -6 $O/pmc_ea_one_block.txt | cut -c1-200
echo "== sorter input classes"; timeout 600 python tools/bwt_inputs.py > $O/bwt_inputs.txt 2>&1; tail -7 $O/bwt_inputs.txt | cut -c1-220
echo "== BASELINE configs against the reference"; timeout 900 python tools/config_table.py > $O/config_table.txt 2>&1; tail -12 $O/config_table.txt | cut -c1-260
echo "== 8 ranks over gloo on one GPU, 2 CPUs of the quota each"
BSC_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 6 --warmup 2 --contexts 1 --no-cpu-baseline > $O/bench_n8_gloo_shared_gpu.json 2> $O/bench_n8_gloo.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r5_final/bench_n8_gloo_shared_gpu.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'n_gpus', 'verified')}, [(r['rank'], r['verified'], r['cpu_seconds_per_block'], r['coder_threads']) for r in d['per_rank']])
except Exception as e:
    print('n8 failed', e, open('gpurun_out/r5_final/bench_n8_gloo.err').read()[-800:])
PY
echo "== sanitizer build: tests/test_gpu_compress.py"; timeout 1500 python tools/asan_run.py python -m pytest tests/test_gpu_compress.py -q -m gpu 2>&1 | tail -25 | cut -c1-300
echo "== whole GPU suite"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5
} > $O/out.txt 2>&1
rm -rf gpurun_out/prof_r05/*/*/*_agent_info.csv 2>/dev/null
du -sh gpurun_out/prof_r05 gpurun_out/r5_final 2>/dev/null
grep -v amdgpu.ids $O/out.txt | cut -c1-420 | tail -110
