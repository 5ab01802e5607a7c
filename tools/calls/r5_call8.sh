#!/bin/bash
# round 5, GPU call 8: the multi-second set-up stall of a 6-context process: does it depend on what the previous process left behind?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call8; mkdir -p $O
J=libbsc_amd/lib/job_bench
js() { python -c "import json;d=json.load(open('$1'));print({k:d[k] for k in ('value','create_s','setup_s')})" 2>&1 | tail -1; }
{
echo "== first process on a fresh box, 6 contexts"; $J --steps 8 --warmup 0 > $O/a.json 2>$O/a.err; js $O/a.json
echo "== back to back"; for i in 1 2 3; do $J --steps 8 --warmup 0 > $O/a.json 2>$O/a.err; js $O/a.json; done
echo "== 3 s apart"; for i in 1 2 3; do sleep 3; $J --steps 8 --warmup 0 > $O/a.json 2>$O/a.err; js $O/a.json; done
echo "== 2 contexts, back to back"; for i in 1 2 3; do $J --steps 8 --warmup 0 --contexts 2 > $O/a.json 2>$O/a.err; js $O/a.json; done
echo "== 6 contexts with timing"; BSCGPU_TIMING=1 $J --steps 8 --warmup 0 2>&1 >/dev/null | grep -v "pinned landing\|gpu_stage of one" | cut -c1-120 | head -40
rocm-smi --showmeminfo vram 2>/dev/null | head -8
} > $O/out.txt 2>&1
cut -c1-300 $O/out.txt | tail -70
