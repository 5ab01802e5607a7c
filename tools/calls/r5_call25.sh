#!/bin/bash
# round 5, GPU call 25: as call 24, plus the partition's positions flushed from LDS by 16-byte stores
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5_call25; mkdir -p $O
V=$PWD/libbsc_amd/lib/variants
{
echo "== round 4's device model"; BSC_LIB_OVERRIDE=$V/libbsc_dcold.so timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== new"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== new"; timeout 200 python tools/devcoder_time.py 2>&1 | tail -1 | cut -c1-300
echo "== parity"; timeout 900 python -m pytest tests -x -q -m gpu -k "device_static_model or fast_coder_on_the_device or lzp_blocks_take or eight_sub_block or golden_fixtures or sub_block_count or front_end_rank or full_size_64m or matches_reference" 2>&1 | tail -3
} > $O/out.txt 2>&1
cut -c1-400 $O/out.txt | tail -12
