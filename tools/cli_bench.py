"""End-to-end CLI comparison on one box: the unmodified reference `bsc e` (CPU) against the same CLI relinked
against libbsc_mi355x.so (oracle/_ref/bsc_mi355x), on 8 x 64 MiB of synth-text v1 (seeds 10..17), -b64 -p -e1.
Wall time includes reading and writing the files (tmpfs).
    python tools/cli_bench.py [blocks] [callers,callers,...]     more blocks (the 8 repeated): the relinked CLI only, per OpenMP team size —
    process start-up (HIP, two contexts' arenas and pinned buffers: ~0.8 s) is then a smaller part of the run"""
import os, subprocess, sys, time
sys.path.insert(0, '.')
import numpy as np
from libbsc_amd import api
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
src = os.path.join(tmp, "cli_bench_in.bin")
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
teams = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4]
with open(src, "wb") as f:
    eight = [api.synth_text_v1(seed, 64 << 20) for seed in range(10, 18)]
    for b in range(nblocks):
        eight[b % 8].tofile(f)
size = os.path.getsize(src)
def cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": n = min(n, max(1, int(int(q) / int(p))))
    except Exception: pass
    return n
ncpu = cpus()
def run(binary, flags, threads, tag):
    out = os.path.join(tmp, "cli_bench_%s.bsc" % tag)
    best = None
    for _ in range(2):
        t0 = time.time()
        r = subprocess.run([os.path.join(root, "oracle", "_ref", binary), "e", src, out] + flags.split(),
                           capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="passive"))
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout + r.stderr
        best = dt if best is None or dt < best else best
    print(f"{tag:28s} {flags:18s} OMP_NUM_THREADS={threads:<3d} {best:7.2f} s  {size / 1e6 / best:8.1f} MB/s  -> {os.path.getsize(out)} B")
    return out
if nblocks != 8:
    for th in teams:
        run("bsc_mi355x", "-b64 -p -e1", th, f"relinked_mi355x, {nblocks} blocks")
    sys.exit(0)
a = run("bsc", "-b64 -p -e1", ncpu, "reference_cpu")
b = run("bsc_mi355x", "-b64 -p -e1", 4, "relinked_mi355x")
run("bsc_mi355x", "-b64 -p -e1 -t", 1, "relinked_mi355x_inorder")
from libbsc_amd.cli import parse_container
same = sorted(parse_container(open(a, "rb").read())) == sorted(parse_container(open(b, "rb").read()))
print("same blocks:", same, "| host cpus (effective):", ncpu)
