#!/bin/bash
# rocprofv3 evidence for the round: kernel-trace stats of bench.py, then separate PMC passes (HBM bytes).
# usage: bash tools/profile_round.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
# PMC passes: one whole 64 MiB block (sorter, QLFC front end, device coder), counters in their own runs
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python tools/pmc_one_block.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python tools/pmc_one_block.py > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
python tools/summarize_prof.py $OUT "python bench.py --steps 8 --warmup 2 --no-cpu-baseline" > $OUT/summary_$TAG.txt 2>&1
cat $OUT/summary_$TAG.txt
