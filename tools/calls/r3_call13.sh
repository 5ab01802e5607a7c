#!/bin/bash
# round 3, GPU call 13: the digit-pass variants of this round side by side on one box (alternating runs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
V=libbsc_amd/lib/variants
{
echo "A = three tiles in flight, 15 streaming waves, scout publishes, 8-byte loads of all rows (commit b3bf58d)"
echo "B = A + streaming waves publish, scout at raised priority, buffer loads of existing rows only (7dab301); C = B without the priority"
echo "E = four tiles in flight, 11 streaming waves x 5632-record tiles (350f867); F = E without the priority"
timeout 1500 python tools/os_ab.py default:BSC_RS_ONESWEEP=0 $V/libbsc_osv_A.so $V/libbsc_osv_B.so $V/libbsc_osv_C.so $V/libbsc_osv_E.so $V/libbsc_osv_F.so 2>&1 | tail -8
} > gpurun_out/r3_call13.txt 2>&1
cat gpurun_out/r3_call13.txt
