#!/usr/bin/env python3
"""bench.py — MB/s compress (BWT + QLFC) on 64 MiB blocks, 1/2/4/8 GPU (BASELINE.json metric).

A step = one pass of the hot path over one 64 MiB synthetic block per GPU, input already resident in HBM:
Adler-32 + forward BWT (LSD radix first sort on 12-character keys, refinement rounds as segmented sorts on text keys) +
QLFC front end + the static coder's adaptive model (-e1: every probability, devcoder.hip) on the MI355X, range coding
on host threads, block container — i.e. bsc_compress(lzp off, BWT, QLFC_STATIC) — then, for N > 1, the
compressed blocks are concatenated on rank 0 over RCCL/xGMI (send/recv of the variable-size blocks).
Blocks are independent, so N GPUs = N blocks per step (weak scaling), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     — the dominant kernel (the LSD digit pass of the BWT's first sort: rs_onesweep_kernel, records read once and
                 written once): algorithmic bytes 2*m*(8+4) per full-size launch over the HIP-event time of those launches on
                 the kernel's own stream, against 8 TB/s HBM3E; sort_frac charges the whole sort (one histogram read +
                 P passes, SURVEY 8d's B_sort) with every radix kernel's time;
  cpu_baseline — the reference libbsc CPU path (oracle/_ref, built from /root/reference) timed on this box's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 64 << 20
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)


def _dma_in_use():
    import ctypes
    from libbsc_amd import _native
    try:
        f = _native.lib().bscgpu_d2h_dma_available
        f.restype = ctypes.c_int
        return f() == 1
    except Exception:
        return False


def _process_affinity():
    """The WHOLE process (not only the library's coder pool, which does this by itself: block.cpp pool_cpu_set) on one hardware thread per core, of
    the GPUs' NUMA node if they all sit on one — when a cgroup CPU-time quota says the machine is not meant to be filled (the 1-GPU boxes: 16 CPUs
    of time on 256 hardware threads).  What numactl / taskset would do for any caller; BSC_BENCH_AFFINITY=0 leaves the process alone.
    profiles/r06/pool_affinity.txt: 20-step means 4472 against 4228 (call 40), 4291 against 4107 / 4221 (call 44) — at the edge of the spread."""
    if os.environ.get("BSC_BENCH_AFFINITY", "1") == "0":
        return
    import glob
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q == "max":
            return
        quota = int(q) // int(period)
        allowed = os.sched_getaffinity(0)
        nodes = set()
        for f in glob.glob("/sys/bus/pci/drivers/amdgpu/*/numa_node"):
            nodes.add(int(open(f).read()))
        def cpulist(t):
            out = set()
            for part in t.strip().split(","):
                a, _, b = part.partition("-")
                out.update(range(int(a), int(b or a) + 1))
            return out
        cand = allowed
        if len(nodes) == 1 and min(nodes) >= 0:
            cand = allowed & cpulist(open(f"/sys/devices/system/node/node{min(nodes)}/cpulist").read())
        keep = set()
        for c in cand:
            sib = cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()) & allowed
            if c == min(sib):
                keep.add(c)
        if quota >= 1 and len(keep) >= quota and len(keep) < len(allowed):
            os.sched_setaffinity(0, keep)
    except Exception as e:
        print(f"[bench] process affinity not set: {e!r}", file=sys.stderr)


def _narrow_affinity_to_gpu_node(torch, local):
    """After _process_affinity: the box's sysfs lists every GPU of the machine, on both NUMA nodes, although the process sees one; now that HIP says
    which one, the main thread (and every thread it starts from here on: the pipes' threads, the library's pool) keeps to that GPU's node."""
    if os.environ.get("BSC_BENCH_AFFINITY", "1") == "0":
        return
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] == "max":
            return
        quota = int(q[0]) // int(q[1])
        pr = torch.cuda.get_device_properties(local)
        bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read())
        if node < 0:
            return
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        keep = os.sched_getaffinity(0) & cpus
        if len(keep) >= max(quota, 1):
            os.sched_setaffinity(0, keep)
    except Exception as e:
        print(f"[bench] affinity not narrowed to the GPU's node: {e!r}", file=sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--block", type=int, default=BLOCK)
    ap.add_argument("--sorter", type=int, default=1)
    ap.add_argument("--coder", type=int, default=1)
    ap.add_argument("--depth", type=int, default=0, help="blocks in flight per GPU (0 = from the coder pool size); their sub-blocks feed the pool of coder threads")
    ap.add_argument("--contexts", type=int, default=5, help="GPU contexts (own stream, arena and pipe each) driven concurrently on every GPU: "
                    "kernels of different blocks interleave on the device, which fills what one block's latency-bound kernels leave idle "
                    "(round 4, one box, 320 / 20 steps: 3 x 4 4834, 4 x 4 4882 / 3655, 6 x 3 5054 / 3903, 5 x 3 - / 3640, 8 x 2 - / 3513 MB/s; "
                    "round 6, with the p-stream copies off the CUs, 160 / 20 steps, means of 3 / 5 interleaved runs: 4 x 4 5963, 5 x 4 5975 / 4466, "
                    "6 x 4 5980 / 4377, another box 5 x 4 6076, 6 x 4 5885, 7 x 4 5843, 8 x 4 5929: profiles/r06/contexts.txt)")
    ap.add_argument("--lzp", default="", help="H,M: LZP preprocessing as the reference CLI's default has it (-H15 -M128: --lzp 15,128).  The block then enters through "
                    "bscgpu_pipe_submit_host from host memory (LZP is host code); a separate, labelled line — BASELINE's configs have LZP off")
    ap.add_argument("--input", default="synth-text-v1", choices=["synth-text-v1", "python-source", "binary"],
                    help="what the blocks hold.  synth-text-v1 is BASELINE's workload and the only `value` that counts; python-source / binary are the image's "
                    "own *.py / *.so files (libbsc_amd.synth.image_corpus: long repeats, the sorter's hard classes) — separate, labelled lines, checked "
                    "against the compiled reference at run time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    _process_affinity()            # (BSC_BENCH_AFFINITY=0: off)
    import torch
    import torch.distributed as dist
    from libbsc_amd import GpuContext, api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BSC_BENCH_BACKEND=gloo is a functional-test mode for boxes with fewer GPUs than ranks: ranks share GPUs
    # (LOCAL_RANK modulo the device count) and the concatenation travels over gloo/CPU instead of RCCL/xGMI.
    backend = os.environ.get("BSC_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # host-thread budget of this rank: its share of the CPUs the job may use (affinity and cgroup quota); the native default is
    # "all of them", which is right for one rank per box
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    ncx = max(1, args.contexts)
    cpus_rank = max(1, effective_cpus() // max(local_world, 1))
    if "BSCGPU_HOST_THREADS" not in os.environ:         # ONE coder pool per process, shared by this rank's contexts
        # (a quarter more threads than the CPU share: a task spends part of its life asleep, waiting for its block's copy from the GPU; half
        # as many again — rounds 3-4 — overdraws a cgroup CPU quota at the tail of a job, and the quota then stops the whole process)
        os.environ["BSCGPU_HOST_THREADS"] = str(max(4, min(96, cpus_rank + cpus_rank // 4)))
        os.environ.setdefault("BSCGPU_HOST_CPUS", str(cpus_rank))
    coder_threads = int(os.environ["BSCGPU_HOST_THREADS"])
    # Range coding on the host: all eight sub-blocks of a block in SIMD lanes on one thread (one task of ~0.10 s with
    # AVX-512VL, 0.13 with AVX2) or pairs of sub-blocks per task (four tasks of ~0.05 s, 0.19 CPU-s per block).
    # The library picks per block (block.cpp: ps_group): pairs while at least four CPUs of the pool's budget are idle, eight lanes
    # when the coder threads are busy; without AVX-512VL it takes pairs, and then a rank whose share of the CPUs could not feed its
    # GPU with pairs (~68 blocks/s x 0.228 s = 15 CPUs) is given the AVX2 lanes here.  BSC_RC_SIMD=8 / 0 forces one of the two.
    try:
        has_avx512vl = "avx512vl" in open("/proc/cpuinfo").read()
    except OSError:
        has_avx512vl = False
    if "BSC_RC_SIMD" not in os.environ:
        if os.environ.get("BSC_RC_X8") == "1" or (args.coder == 1 and not has_avx512vl and cpus_rank < 14):
            os.environ["BSC_RC_SIMD"] = "8"
    rc_simd = int(os.environ.get("BSC_RC_SIMD", "-1"))
    rc_adaptive = rc_simd < 0 and has_avx512vl and os.environ.get("BSC_RC_ADAPTIVE", "0") != "0"
    rc_x8 = rc_simd == 8 or (rc_simd < 0 and has_avx512vl)
    if args.depth <= 0:                                 # blocks in flight per context
        args.depth = max(2, min(4, 8 // ncx))
        if rc_x8: args.depth = max(2, min(4, 24 // ncx))      # one longer task per block (~105 ms of a CPU against ~12 ms of the GPU): 24 blocks in
                                                              # flight per GPU (round 6: with 18 a pipe regularly sat in wait() on its oldest block
                                                              # while the GPU had nothing to do — 3991 -> 4089 MB/s at the driver's 20 steps, 5 runs each)
    torch.cuda.set_device(local)
    if world == 1:
        _narrow_affinity_to_gpu_node(torch, local)
    dev = torch.device("cuda", local)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    n = args.block

    # one 64 MiB synth-text v1 block per GPU: seed 2 at N=1 (BASELINE config 3), seeds 10..17 at N>1 (config 4)
    # (BASELINE config 5, the ST5 / ST6 ablation on 128 MiB blocks, has its committed reference output for seed 3)
    seed = (3 if (args.sorter in (5, 6) and n == (128 << 20)) else 2) if world == 1 else 10 + rank
    if args.input == "synth-text-v1":
        host_in = api.synth_text_v1(seed, n)
    else:
        from libbsc_amd.synth import image_corpus
        host_in = image_corpus(args.input, n)
        if host_in is None:
            sys.exit(f"bench.py: this image holds no files for --input {args.input}")
    d_in = torch.from_numpy(host_in).to(dev)
    ctxs = []

    # rank 0 receives world - 1 compressed blocks per round into one staging tensor; the others stage one block
    gather_buf = torch.empty((n + 64) * (max(world - 1, 1) if rank == 0 else 1), dtype=torch.uint8, device=comm_dev)

    from libbsc_amd.multigpu import Concatenator
    import threading
    pipes = []                                                              # compressed blocks land in recycled host buffers
    thread_errors = []
    stage = np.zeros(6)
    stage_lock = threading.Lock()

    concat = None

    def finish(pipe, ticket):
        blk = pipe.wait(ticket)
        if concat is not None:      # final concatenation on rank 0 over RCCL / xGMI, on a background thread per rank
            with stage_lock:
                concat.put(blk)
        return blk

    LOW_LATENCY = 0x10000                               # include/bscgpu.h
    # the last block of every pipe is marked BSCGPU_FEATURE_LOW_LATENCY (short host tasks whatever the pool's load): the pipeline's drain
    # is part of the timed region, and only the caller knows where a job ends
    tail_low_latency = os.environ.get("BSC_BENCH_TAIL", "1") != "0"
    ll_blocks = int(os.environ.get("BSC_BENCH_LL", "0"))     # how many of a job's last blocks are marked (0: one per context)

    trace = [] if os.environ.get("BSC_BENCH_TRACE") else None      # (pipe, block, what, seconds since the run started): where a short run's time goes
    t_run0 = [0.0]

    host_leg = [False]              # the boundary leg after the timed region: blocks enter through bscgpu_pipe_submit_host (pageable host memory)
    lzp = tuple(int(x) for x in args.lzp.split(",")) if args.lzp else (0, 0)

    # How the blocks of a run reach the contexts.  Steady state wants all contexts busy (kernels of different blocks interleave on the
    # GPU: +14 %), but the END of a job does not: six contexts that each hold one of the last six blocks finish their GPU stages in one
    # burst at the very end, and ~1.4 CPU-s of range coding are then left for 16 CPUs (round 3: 94 ms of drain behind 265 ms of GPU
    # work at the driver's 20 steps).  So the blocks come from ONE queue, and context k only takes a block while more than k blocks are
    # left: the last block goes to context 0 alone, the last two to contexts 0 and 1, ... — the tail's GPU stages finish one after
    # the other, and their coding overlaps the GPU work still to come.  BSC_BENCH_QUEUE=0: the static split of rounds 1-3.
    # The START of a job is tapered the same way: context k draws its first block once k GPU stages of the run have finished.  With all
    # contexts starting at once their first six GPU stages interleave and end together after ~6 x 13 ms, and the coder threads have
    # nothing to do until then (round 3's timeline: first block coded at 144 ms of 340); staggered, the first block reaches them after
    # one GPU stage, and the stages keep ending one at a time.
    use_queue = os.environ.get("BSC_BENCH_QUEUE", "1") != "0"
    tail_rule = int(os.environ.get("BSC_BENCH_TAILRULE", "0"))      # 0: context k draws only while more than k blocks are left; 1: any context draws; 2: while more than k // 2 are left
    head_rule = int(os.environ.get("BSC_BENCH_HEADRULE", "0"))      # 0: context k starts after k GPU stages of the run; 1: after k // 2; 2: at once
    urgent_blocks = int(os.environ.get("BSC_BENCH_URGENT", "0"))    # how many of the job's last blocks are coded as eight single-stream tasks whatever the pool's load
    URGENT = 0x20000
    queue_lock = threading.Condition()
    queue_state = {"next": 0, "total": 0, "stages_done": 0}

    def take_block(k, i_static, steps_static):
        """-> (take it?, is one of the job's last blocks?)"""
        if queue_state["total"] == 0:                     # static split: pipe k's own share
            return i_static < steps_static, (1 if i_static == steps_static - 1 else 0)
        with queue_lock:
            if i_static == 0:                             # this context's first block of the run
                need = k if head_rule == 0 else (k // 2 if head_rule == 1 else 0)
                queue_lock.wait_for(lambda: queue_state["stages_done"] >= need or queue_state["next"] >= queue_state["total"])
            left = queue_state["total"] - queue_state["next"]
            if left <= (k if tail_rule == 0 else (0 if tail_rule == 1 else k // 2)):
                return False, False
            queue_state["next"] += 1
            return True, (0 if announce else 2 if left <= urgent_blocks else 1 if left <= ll_blocks else 0)

    def stage_finished():
        if queue_state["total"]:
            with queue_lock:
                queue_state["stages_done"] += 1
                queue_lock.notify_all()

    def run_one(k, steps, record, out):
        """blocks through pipe k (`steps` of them with the static split, else from the queue): GPU stage of block i+1 overlaps the host
        coding of block i."""
        pipe, cx = pipes[k], ctxs[k]
        tickets, blk = [], None
        local_stage = np.zeros(6)
        done = 0
        i = -1
        while True:
            i += 1
            take, last = take_block(k, i, steps)
            if not take:
                break
            feat = 3 | ((LOW_LATENCY if last == 1 else URGENT | LOW_LATENCY) if (tail_low_latency and last) else 0)
            if trace is not None and record: trace.append((k, i, "submit", time.perf_counter() - t_run0[0]))
            tickets.append(pipe.submit_host(host_in, args.sorter, args.coder, lzp[0], lzp[1], feat) if (host_leg[0] or lzp[0]) else pipe.submit(d_in, n, args.sorter, args.coder, feat))
            stage_finished()
            if trace is not None and record: trace.append((k, i, "gpu stage done", time.perf_counter() - t_run0[0]))
            if record:
                local_stage += np.array(cx.last_stage_ms())
            if len(tickets) >= args.depth:
                blk = finish(pipe, tickets.pop(0))
                if trace is not None and record: trace.append((k, done, "coded", time.perf_counter() - t_run0[0]))
                done += 1
        while tickets:
            blk = finish(pipe, tickets.pop(0))
            if trace is not None and record: trace.append((k, done, "coded", time.perf_counter() - t_run0[0]))
            done += 1
        out[k] = blk
        if record:
            with stage_lock:
                stage[:] += local_stage

    announce = os.environ.get("BSC_BENCH_ANNOUNCE", "1") != "0"       # tell the coder pool how many blocks the job has (bscgpu_coder_pool_expect)

    def run(steps, record=False, static=False):
        out = [None] * ncx
        if announce and not static:
            import ctypes as C
            from libbsc_amd import _native as NN
            NN.lib().bscgpu_coder_pool_expect(C.c_longlong(steps), 1)
        share = [steps // ncx + (1 if k < steps % ncx else 0) for k in range(ncx)]
        queue_state["next"] = 0
        queue_state["stages_done"] = 0
        queue_state["total"] = 0 if (static or not use_queue or ncx == 1) else steps
        def guarded(k):
            try:
                run_one(k, share[k], record, out)
            except BaseException as e:                                      # a thread's exception must fail the run, not shorten it
                thread_errors.append(e)
                with queue_lock:                                            # ... nor leave the others waiting for its GPU stages
                    queue_state["next"] = queue_state["total"]
                    queue_lock.notify_all()
        if ncx == 1:
            guarded(0)
        else:
            ths = [threading.Thread(target=guarded, args=(k,)) for k in range(ncx)]
            for t in ths: t.start()
            for t in ths: t.join()
        if thread_errors:
            raise thread_errors[0]
        return next(b for b in reversed(out) if b is not None)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        concat = Concatenator(rank, world, comm_dev, staging=gather_buf)
    # setup, untimed: every pipeline slot is used once, so that its pinned landing zones (allocated on first use) and the
    # contexts' device arenas exist before the warm-up steps — otherwise those allocations land in the timed region
    # (six contexts want ~19 GB of HBM and ~2.7 GB of pinned host memory each at 64 MiB blocks: a box that cannot give that gets fewer
    # contexts instead of a crash, and the JSON line says how many ran)
    for cand in [ncx] + [k for k in (4, 2, 1) if k < ncx]:
        try:
            ncx = cand
            ctxs = [GpuContext(local, max_n=n + 4096) for _ in range(ncx)]
            pipes = [cx.pipe(args.depth, reuse_outputs=True) for cx in ctxs]
            run(ncx * args.depth, static=True)          # every slot of every pipe once
            break
        except Exception as e:
            print(f"[bench] rank {rank}: {cand} context(s) x {args.depth} could not be set up ({e!r})" + ("; trying fewer" if cand > 1 else ""), file=sys.stderr)
            del thread_errors[:]
            for p in pipes: p.close()
            for cx in ctxs: cx.close()
            pipes, ctxs = [], []
            if cand == 1:
                raise
    ctx = ctxs[0]
    if ll_blocks <= 0: ll_blocks = ncx
    blk = run(args.warmup)
    if concat is not None:
        concat.close()
        concat = Concatenator(rank, world, comm_dev, staging=gather_buf)
    for cx in ctxs:
        cx.profile(True)
        cx.profile_reset()
    sync()
    from libbsc_amd.gpu import coder_pool_stats
    coder_pool_stats(reset=True)
    cg0 = cgroup_cpu_stat()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    t_run0[0] = t0
    blk = run(args.steps, record=True)
    if concat is not None:
        concat.close()                                  # every block of the timed region has reached rank 0's host memory
    sync()
    dt = time.perf_counter() - t0
    cpu_used = time.process_time() - cpu0               # all threads of this rank
    cg1 = cgroup_cpu_stat()
    if trace is not None and rank == 0:
        try:                                            # the coder pool's own record of its tasks (BSCGPU_POOL_TRACE=1)
            import ctypes as C
            from libbsc_amd import _native as NN
            L = NN.lib()
            L.bscgpu_steady_now.restype = C.c_double
            off = time.perf_counter() - L.bscgpu_steady_now()           # both clocks are monotonic: one offset
            buf = (C.c_double * (6 * 4096))()
            cnt = L.bscgpu_coder_pool_trace(buf, 4096, 1)
            ids = {}
            for r in range(cnt):
                a, b, job, sub, shape, feat = (buf[6 * r + x] for x in range(6))
                if b + off < t0: continue
                jid = ids.setdefault(job, len(ids))
                trace.append((-1, jid, f"task sub {int(sub)} x{int(shape)}{' LL' if int(feat) & 0x10000 else ''} ran {(a + off - t0) * 1e3:.1f} .. {(b + off - t0) * 1e3:.1f} ms ({(b - a) * 1e3:.1f})", b + off - t0))
        except Exception as e:
            print(f"[trace] pool trace unavailable: {e!r}", file=sys.stderr)
        for k, i, what, t in sorted(trace, key=lambda x: x[3]):
            print(f"[trace] {t * 1e3:8.1f} ms  pipe {k} block {i}: {what}", file=sys.stderr)
        print(f"[trace] {dt * 1e3:8.1f} ms  end of the timed region", file=sys.stderr)
    pool_modes = coder_pool_stats()                     # how the timed blocks were coded (block.cpp: ps_group)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    for cx in ctxs:
        cx.profile(False)

    # ---- outside the timed region: is the output the reference's?  The last block every rank produced in the timed
    # region is checked against the reference output committed in tests/golden/golden_big.json (size + md5; generated
    # from the compiled reference by tests/golden/make_golden_big.py) — no reference is needed on this box.
    verified, verify_note = verify_block(blk, seed, n, args.sorter, args.coder)
    if lzp[0] or args.input != "synth-text-v1":           # no committed output for these runs: the compiled reference, where it travelled with the tree
        try:
            from oracle.refbind import Ref
            want = Ref().compress(host_in, args.sorter, args.coder, lzp_hash=lzp[0], lzp_min=lzp[1])
            verified = blk.tobytes() == want
            verify_note = "last timed block against the compiled reference's bsc_compress with the same parameters on the same input (oracle/_ref), outside the timed region"
        except Exception as e:
            verified, verify_note = None, f"no committed output for this run and the compiled reference is unavailable: {e}"

    stats = ctxs[0].profile_get()
    for cx in ctxs[1:]:
        for k, v in cx.profile_get().items():
            for f in ("ms", "launches", "bytes", "records"):
                stats[k][f] += v[f]
    launches_timed = [x for cx in ctxs for x in cx.scatter_launches(65536)]
    # ---- roofline leg.  With several contexts per GPU the kernels of different blocks run side by side in the timed region, so
    # the HIP-event duration of one launch there measures how the chip was shared, not the kernel.  The kernel's own rate is
    # therefore taken from a second, untimed pass of the same workload through ONE context (same block, same pipe, HIP events on
    # the context's stream); the timed region's (contended) numbers are reported next to it.  One context: the timed region itself.
    stats_iso, launches_iso, iso_blocks = stats, launches_timed, args.steps
    if ncx > 1 and rank == 0:
        iso_blocks = max(2, min(8, args.steps))
        concat = None                                   # (N > 1: these blocks are not part of the job's output)
        ctxs[0].profile(True)
        ctxs[0].profile_reset()
        queue_state["total"] = 0
        run_one(0, iso_blocks, False, [None] * ncx)
        torch.cuda.synchronize()
        ctxs[0].profile(False)
        stats_iso = ctxs[0].profile_get()
        launches_iso = ctxs[0].scatter_launches(65536)
    # ---- pattern ceiling of the digit pass: the same blocks once more through the same context with the three-kernel passes
    # (BSCGPU_OPT_RS_ONESWEEP = 0): their scatter kernel moves the same records into the same 256 runs per tile with the same ranking
    # code, but every offset is known before it starts (rs_hist + rs_scan) — no tickets, no look-back, no scout wave.  What it reaches
    # on THIS box is the denominator `frac` lacks besides the spec sheet: the boxes of the pool differ by ~15 % for this access pattern.
    launches_ceiling = []
    if rank == 0 and os.environ.get("BSC_RS_ONESWEEP", "3") != "0" and (args.sorter == 1 or os.environ.get("BSC_RS_ONESWEEP", "3") in ("2", "3")):
        try:
            prev = ctxs[0].option_set(ctxs[0].OPT_RS_ONESWEEP, 0)
            ctxs[0].profile(True)
            ctxs[0].profile_reset()
            concat = None
            queue_state["total"] = 0
            run_one(0, max(2, min(4, args.steps)), False, [None] * ncx)
            torch.cuda.synchronize()
            ctxs[0].profile(False)
            launches_ceiling = ctxs[0].scatter_launches(65536)
            ctxs[0].option_set(ctxs[0].OPT_RS_ONESWEEP, prev)
        except Exception as e:                            # reporting only
            print(f"[bench] pattern-ceiling leg failed: {e!r}", file=sys.stderr)
    free_b, total_b = torch.cuda.mem_get_info(local)
    # ---- the boundary C callers actually have: host pointers.  The same number of blocks twice through all pipes, once with the input
    # resident in HBM (as the timed region) and once entering through bscgpu_pipe_submit_host from ordinary (pageable) host memory —
    # one H2D copy per block in front of the same GPU stage.  Untimed legs, reported beside `value`, never as `value`.
    boundary = None
    if rank == 0 and world == 1:
        try:
            concat = None
            hb = max(2 * ncx * args.depth, min(args.steps, 48))
            legs = {}
            for name, flag in (("device_resident", False), ("host_input", True)):
                host_leg[0] = flag
                torch.cuda.synchronize()
                th0 = time.perf_counter()
                blk_h = run(hb)
                torch.cuda.synchronize()
                legs[name] = hb * n / 1e6 / (time.perf_counter() - th0)
                if flag:
                    ok_h, _ = verify_block(blk_h, seed, n, args.sorter, args.coder)
            host_leg[0] = False
            boundary = {"blocks": hb, "device_resident_MBps": round(legs["device_resident"], 1), "host_input_MBps": round(legs["host_input"], 1),
                        "host_over_device": round(legs["host_input"] / legs["device_resident"], 3), "host_input_verified": ok_h,
                        "what": "bscgpu_pipe_submit_host on pageable host memory (one H2D per block, overlapped across the contexts) against bscgpu_pipe_submit on the "
                                "resident block, same pipes, same number of blocks, fill and drain included in both"}
        except Exception as e:                            # reporting only
            print(f"[bench] boundary leg failed: {e!r}", file=sys.stderr)
            host_leg[0] = False
    # ---- BASELINE's other configurations, after the timed region (reported under `configs`, never as `value`): config 1 (1 MiB, -e1:
    # latency of one synchronous call), config 2 (forward BWT alone on the 64 MiB block), config 5 (the ST5 / ST6 ablation on 128 MiB
    # blocks).  Each is checked against the committed reference outputs (tests/golden/golden.json, golden_big.json).
    other_configs = None
    if rank == 0 and world == 1 and args.sorter == 1 and args.coder == 1 and not lzp[0] and args.input == "synth-text-v1" and n == BLOCK \
            and os.environ.get("BSC_BENCH_CONFIGS", "1") != "0":
        try:
            other_configs = baseline_configs(torch, dev, local, ctxs[0], d_in, n, GpuContext, api)
        except Exception as e:                            # reporting only
            print(f"[bench] configs leg failed: {e!r}", file=sys.stderr)
    # what leaves this rank's GPU over PCIe in the timed region: 13 (until round 6: 16) bits per binary decision of the device model (run arrays instead
    # for blocks on the host model: not counted) + nothing else of size (the input is resident, the sorted block never crosses);
    # the host's DRAM sees those bytes twice (DMA write, coder read) — with 8 ranks per node this, not xGMI, is the shared resource
    # (decisions per block from the p-stream kernel's own launches: the profile of the timed region folds a launch in at the context's next
    # sync, so its SUM misses the blocks still in flight at the end — round 4's line said 256.6 MB per block at 20 steps for that reason,
    # where every block of this workload has 183.2 M decisions = 366.4 MB)
    dps = stats.get("dc_pstream", {})
    decisions_per_block = dps.get("records", 0) / max(dps.get("launches", 0), 1)
    # (round 6: the static coder's stream crosses as 13 bits per decision, each sub-block's padded to 64 decisions — BSCGPU_OPT_DC_PACKED_STREAM)
    ps_packed = args.coder == 1 and ctxs[0].option_get(ctxs[0].OPT_DC_PACKED_STREAM) == 1
    bytes_per_decision = 13.0 / 8.0 if ps_packed else 2.0
    d2h_bytes = bytes_per_decision * decisions_per_block * args.steps
    mine = {"rank": rank, "verified": verified, "gpu_stage_total_ms": round(float(stage[0] + stage[1] + stage[2]) / args.steps, 2),
            "pcie_d2h_MBps": round(d2h_bytes / 1e6 / dt, 1), "pcie_d2h_MB_per_block": round(d2h_bytes / 1e6 / args.steps, 1),
            "host_dram_MBps_estimate": round(2 * d2h_bytes / 1e6 / dt, 1),
            # host DRAM traffic of one block (8 ranks per node share the host's memory system, not xGMI): the p stream is written once by the
            # DMA engine and read once by the range coder; the compressed block is written once; a host-resident input is read once for the H2D
            "pstream_bits_per_decision": 13 if ps_packed else 16,
            "pstream_copy": "HSA DMA copy, one signal per sub-block piece (csrc/device/dma_copy.cpp)" if _dma_in_use() else "hipMemcpyAsync (a copy kernel on torch's HIP runtime)",
            "host_dram_bytes_per_block": {"pstream_dma_write": int(bytes_per_decision * decisions_per_block), "pstream_coder_read": int(bytes_per_decision * decisions_per_block),
                                          "compressed_block_write": int(blk.size), "input_read_for_h2d": int(n if (lzp[0] or host_leg[0]) else 0)},
            "sorter_only_MBps": round(n / 1e6 / max(stage[1] / args.steps / 1e3, 1e-9), 1),
            "cpu_seconds_per_block": round(cpu_used / args.steps, 3), "coder_threads": coder_threads,
            "effective_cpus_of_process": effective_cpus()}
    per_rank = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered

    if rank == 0:
        value = world * args.steps * n / 1e6 / dt
        rec_bytes = 12 if args.sorter == 1 else 8

        os_env = os.environ.get("BSC_RS_ONESWEEP", "3")
        onesweep = os_env != "0" and (args.sorter == 1 or args.sorter == 8 or os_env in ("2", "3"))
        kernel_name = ("rs_onesweep_kernel<true> (one 8-bit LSD digit pass of the BWT's first sort, records read once and written once: u64 key + u32 value; "
                       "15 streaming waves x 7680-record tiles + a scout wave that collects the tile offsets by decoupled look-back)") if onesweep and args.sorter == 1 else \
                      ("rs_onesweep_kernel<false> (one 8-bit LSD digit pass of the sort transform, u64 keys only, records read once and written once; "
                       "tile offsets by decoupled look-back)") if onesweep else \
                      ("rs_scatter_tiled_kernel<true> (one 8-bit LSD digit pass, 1024 x 8 shape, tiles interleaved per XCD; offsets from rs_hist + rs_scan)"
                       if args.sorter == 1 else "rs_scatter_kernel<false, 256, 16, 1> (one 8-bit LSD digit pass of the sort transform: u64 keys only; offsets from rs_hist + rs_scan, "
                       "whose read of the keys per pass is charged in sort_frac, not here)")

        def scatter_rate(launches):
            """the graded kernel only: launches over all n records (the sorter's digit passes; the device coder's keys-only passes over
            the runs are booked under radix_aux and never enter this list)"""
            nfull = max((rec for _, rec in launches), default=n)   # (= n; with LZP the sorter's input is the shorter LZP output)
            full = [(ms, rec) for ms, rec in launches if rec == nfull]
            tot_ms = sum(ms for ms, _ in full) or 1e-9
            tot_bytes = sum(2 * rec_bytes * rec for _, rec in full)
            return full, tot_ms, tot_bytes, tot_bytes / 1e6 / tot_ms     # ..., GB/s

        def sort_rate(st, launches):
            """whole sort against SURVEY 8d's B_sort = m*kb + P*2*m*(kb+vb): every radix kernel of the sorter (the per-sort histogram
            read, per-pass histogram / scan kernels where the three-kernel pass runs, the digit passes) over the algorithmic bytes of
            the full-size passes + one key read per sort"""
            ms = sum(st[k]["ms"] for k in ("radix_hist_all", "radix_hist", "radix_scan", "radix_scatter") if k in st)
            passes = [rec for _, rec in launches]
            nsorts = max(st.get("radix_hist_all", {}).get("launches", 0), 1)
            byts = sum(2 * rec_bytes * rec for rec in passes) + (8 * n * nsorts if st.get("radix_hist_all", {}).get("launches", 0) else 0)
            return byts / 1e6 / max(ms, 1e-9)

        full, tot_ms, tot_bytes, achieved = scatter_rate(launches_iso)
        # HBM traffic of the same kernel from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE
        # runs, tools/profile_round.sh): per full-size launch, corrected as MI355X_MICROARCH.md prescribes
        # (KiB units; FETCH_SIZE x2 on gfx950 for wide coalesced loads).  Static artefact, not measured in this run.
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if args.sorter == 1 and n == pm.get("records") and pm.get("kernel", "").startswith("rs_onesweep") == onesweep:
                traffic = pm["rs_scatter_pairs"]["traffic_bytes_per_launch"]
        except Exception:
            pass
        sort_achieved = sort_rate(stats_iso, launches_iso)
        roofline = {
            "bound": "hbm", "kernel": kernel_name,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "sort_frac": round(sort_achieved / HBM_PEAK_GBPS, 4), "sort_achieved": round(sort_achieved, 1),
            "sort_note": "whole sort: every radix kernel of the sorter (histogram read(s), scans, digit passes) over SURVEY 8d's B_sort = m*8 + P*2*m*(8+4)",
            "measured_on": ("the timed region (one context per GPU: launches do not overlap)" if ncx == 1 else
                            f"{iso_blocks} more blocks of the same workload through ONE context right after the timed region (HIP events on its stream); in the timed "
                            f"region {ncx} contexts run side by side, so a launch's duration there would measure sharing of the chip"),
            "traffic": traffic, "traffic_note": "bytes per full-size launch = 2 x FETCH_SIZE + WRITE_SIZE from profiles/pmc_traffic.json "
                                                "(rocprofv3 PMC passes; algorithmic bytes per full-size launch = %d)" % (2 * rec_bytes * n),
            "launches": len(full), "avg_launch_ms": round(tot_ms / max(len(full), 1), 4),
            "bytes_per_launch": int(2 * rec_bytes * (full[0][1] if full else n)),
            "frac_of_copy_ceiling_6290": round(achieved / 6290.0, 4),
        }
        aux = stats_iso.get("radix_aux")
        if aux and aux["launches"]:
            roofline["other_radix_passes"] = {"what": "keys-only passes over the block's runs that also emit the permutation (device coder), histogram + scan + scatter, "
                                                      "20 B per record; not the graded kernel", "GBps": round(aux["bytes"] / 1e6 / max(aux["ms"], 1e-9), 1),
                                              "ms_per_block": round(aux["ms"] / iso_blocks, 3)}
        if launches_ceiling:
            cfull, c_ms, _, c_ach = scatter_rate(launches_ceiling)
            if cfull:
                roofline["pattern_ceiling"] = {
                    "achieved": round(c_ach, 1), "frac_of_peak": round(c_ach / HBM_PEAK_GBPS, 4), "avg_launch_ms": round(c_ms / len(cfull), 4), "launches": len(cfull),
                    "frac_of_ceiling": round(achieved / c_ach, 4),
                    "what": ("rs_scatter_tiled_kernel<true>" if args.sorter == 1 else "rs_scatter_kernel<false>") + " on the same blocks in the same run and context: the same "
                            "records into the same 256 runs per tile with the same ranking code, all offsets precomputed by rs_hist + rs_scan (whose 8 B per record and pass "
                            "are NOT charged here) - the scatter pattern without any cross-workgroup protocol; frac_of_ceiling = graded kernel / this"}
        # share of a block's GPU time that is digit passes (the kernel's own rate, not the contended in-region durations: with several
        # contexts per GPU launches of different blocks overlap, and the sum of their durations exceeds the wall time)
        roofline["digit_pass_ms_per_block"] = round(tot_ms / iso_blocks, 3)
        roofline["digit_pass_share_of_step"] = round(tot_ms / iso_blocks / (dt / args.steps * 1e3), 3)
        per_kernel = {k: {"ms_per_block": round(v["ms"] / iso_blocks, 3), "GBps": round(v["bytes"] / 1e6 / v["ms"], 1) if v["ms"] > 0 else None}
                      for k, v in stats_iso.items() if v["launches"]}
        simd = "AVX-512VL" if has_avx512vl else "AVX2"
        if rc_adaptive:
            coder_desc = (f"per block either all eight sub-blocks in the SIMD lanes of one task ({simd}; {pool_modes['eight_lane_task']} of this rank's {args.steps} timed blocks) "
                          f"or four tasks of two interleaved scalar coders — blocks queued while >= 4 CPUs of the pool's budget were idle"
                          + ((f", and the job's last {ll_blocks} blocks, marked low-latency" if (use_queue and ncx > 1) else f", and the last block of each of the {ncx} pipes, marked low-latency") if tail_low_latency else "") + f" ({pool_modes['pair_tasks']} blocks)"
                          + (f", or eight scalar tasks, a low-latency block that found >= 8 CPUs idle ({pool_modes['scalar_tasks']})" if pool_modes['scalar_tasks'] else ""))
        elif rc_x8:
            coder_desc = (f"all eight sub-blocks of a block in the SIMD lanes of one task, {simd} ({pool_modes['eight_lane_task']} of this rank's {args.steps} timed blocks)"
                          + ((f"; the job's last {ll_blocks} blocks" if (use_queue and ncx > 1) else f"; the last block of each of the {ncx} pipes") +
                             f", marked low-latency, as four tasks of two interleaved scalar coders ({pool_modes['pair_tasks']})"
                             + (f" or, finding >= 8 CPUs idle, eight scalar tasks ({pool_modes['scalar_tasks']})" if pool_modes['scalar_tasks'] else "") if tail_low_latency else ""))
        else:
            coder_desc = "two sub-blocks per task, interleaved scalar coders"
        out = {
            "metric": "MB/s compress (BWT+QLFC) on 64 MiB blocks", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic" if args.input == "synth-text-v1" else f"files of the container image ({args.input})",
            "config": {"workload": (f"{world} x {n >> 20} MiB synth-text-v1 block(s) per step (seed {'2' if world == 1 else '10..'+str(9+world)}), " if args.input == "synth-text-v1" else
                                    f"NOT BASELINE's workload: {world} x {n >> 20} MiB block(s) of the image's own {args.input} files per step (libbsc_amd.synth.image_corpus), ") +
                                   (f"bsc_compress(lzp off, sorter={args.sorter}, coder={args.coder}); input resident in HBM; " if not lzp[0] else
                                    f"bsc_compress(lzp -H{lzp[0]} -M{lzp[1]}, sorter={args.sorter}, coder={args.coder}); input in HOST memory (LZP is host code: bscgpu_pipe_submit_host), one H2D of the LZP output per block; ") +
                                   "Adler-32 + sorter + QLFC run/rank front end" + (" + the static coder's whole adaptive model on the GPU, " + ("13" if ps_packed else "16") + " bits per binary decision over PCIe, "
                                   "range coding on host threads (" + coder_desc + ")" if args.coder == 1 else
                                   " + the fast coder's model on the GPU (one counter per decision: the static coder's char family with shift updates), 16 bits per binary decision "
                                   "over PCIe, range coding on host threads (" + coder_desc + ")" if args.coder == 3 else
                                   "; adaptive model and range coding on host threads (one task per sub-block)") +
                                   f"; {ncx} GPU context(s) x {args.depth} = {ncx * args.depth} block(s) in flight per GPU feeding one pool of {coder_threads} coder threads"
                                   + ("; blocks are drawn from one queue, context k starts drawing when k GPU stages have finished and stops when k or fewer blocks are left (a job's "
                                      "head and tail run on fewer contexts, so GPU stages end one after the other instead of in bursts of six)" if (use_queue and ncx > 1) else "") + "; "
                                   "output checked against the reference's (see verified)",
                       "block_bytes": n, "blocks_per_step": world, "sorter": "BWT" if args.sorter == 1 else f"ST{args.sorter}",
                       "coder": {1: "QLFC static (-e1)", 2: "QLFC adaptive (-e2)", 3: "QLFC fast (-e0)"}[args.coder],
                       "parallelism": f"block-parallel x{world}", "compressed_bytes_rank0": int(blk.size)},
            "verified": all(r["verified"] is True for r in per_rank),
            "verified_note": verify_note,
            "per_rank": per_rank,
            "roofline": roofline,
            "stage_ms_per_step": {"adler32_gpu": round(stage[0] / args.steps, 2), "sort_transform_gpu": round(stage[1] / args.steps, 2),
                                  "qlfc_front_gpu_and_d2h": round(stage[2] / args.steps, 2),
                                  "gpu_stage_total": round((stage[0] + stage[1] + stage[2]) / args.steps, 2),
                                  "doubling_rounds": stage[5] / args.steps, "blocks_in_flight_per_context": args.depth, "contexts_per_gpu": ncx, "blocks_in_flight_per_gpu": ncx * args.depth},
            "sorter_only_MBps": round(n / 1e6 / max(stage[1] / args.steps / 1e3, 1e-9), 1),
            "hbm_bytes_in_use": int(total_b - free_b),
            "hbm_note": f"device memory in use on this GPU at the end of the run (hipMemGetInfo): {ncx} context arena(s), device-coder arenas, look-back tables, the resident input",
            "boundary": boundary,
            "configs": other_configs,
            "kernels": per_kernel,
            "kernels_note": "HIP-event time per kernel class and block, from the region named in roofline.measured_on",
            "host": {"cpus": os.cpu_count(), "effective_cpus": effective_cpus(), "coder_threads_per_gpu": coder_threads,
                     "range_coder": coder_desc, "blocks_by_coder_task_shape_rank0": pool_modes, "cpu_budget_of_pool": int(os.environ.get("BSCGPU_HOST_CPUS", cpus_rank)),
                     "cpu_seconds_per_block_rank0": round(cpu_used / args.steps, 3),
                     # the box grants CPU TIME (cgroup cpu.max), not cores: a process that runs more threads than its quota for part of a
                     # 100 ms period is stopped — GPU-driving threads included — for the rest of it
                     "cgroup_throttled_ms_in_timed_region": (round((cg1["throttled_usec"] - cg0["throttled_usec"]) / 1e3, 1) if cg0 and cg1 else None),
                     "cgroup_throttled_periods_in_timed_region": ((cg1["nr_throttled"] - cg0["nr_throttled"]) if cg0 and cg1 else None),
                     "cpu_busy_fraction_of_effective": round(cpu_used / (dt * max(effective_cpus() / max(local_world, 1), 1)), 3)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(host_in, args.sorter, args.coder)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def baseline_configs(torch, dev, local, ctx64, d_in64, n64, GpuContext, api):
    """BASELINE.json configs 1, 2 and 5 on this GPU, each checked against committed reference outputs.  A few seconds in all."""
    import hashlib
    t_leg = time.perf_counter()
    res = {}
    small = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["blocks"]
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_big.json")))["blocks"]

    def md5(x):
        return hashlib.md5(x.tobytes() if hasattr(x, "tobytes") else x).hexdigest()

    # config 1: one 1 MiB block (synth-text v1, seed 1: the stand-in for the enwik slice), BWT + QLFC static, one synchronous
    # call with the input resident in HBM: latency, best and median of 9
    e1 = next(e for e in small if e.get("kind") == "synth" and (e["seed"], e["n"], e["sorter"], e["coder"]) == (1, 1 << 20, 1, 1))
    d1 = torch.from_numpy(api.synth_text_v1(1, 1 << 20)).to(dev)
    blk = ctx64.compress_device(d1, 1 << 20, 1, 1)
    lat = []
    for _ in range(9):
        t0 = time.perf_counter()
        blk = ctx64.compress_device(d1, 1 << 20, 1, 1)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    res["config1_1MiB_bwt_qlfc_static"] = {
        "latency_ms_best": round(lat[0], 3), "latency_ms_median": round(lat[len(lat) // 2], 3), "MBps_at_best": round((1 << 20) / 1e3 / lat[0], 1),
        "compressed_bytes": int(blk.size), "verified": (int(blk.size), md5(blk)) == (e1["size"], e1["md5"]),
        "what": "bscgpu_compress_device on one 1 MiB synth-text v1 block (seed 1), synchronous, input in HBM, output on the host; checked against tests/golden/golden.json"}

    # config 2: forward BWT alone on the bench block (the bsc_bwt_encode-equivalent device entry point: bscgpu_bwt_device), best of 3
    e2 = next(e for e in big if e["tag"] == "config3")
    out = torch.empty_like(d_in64)
    idx, _ = ctx64.bwt_device(d_in64, out, n64)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        idx, _ = ctx64.bwt_device(d_in64, out, n64)
        dtb = time.perf_counter() - t0
        best = dtb if best is None or dtb < best else best
    res["config2_64MiB_bwt_only"] = {
        "ms": round(best * 1e3, 3), "GBps_of_input": round(n64 / 1e9 / best, 2), "MBps": round(n64 / 1e6 / best, 1), "primary_index": int(idx),
        "verified": int(idx) == e2["bwt_index"] and md5(out.cpu().numpy()) == e2["bwt_md5"],
        "what": "bscgpu_bwt_device on the 64 MiB bench block (seed 2), text and L resident in HBM, wall time of one synchronous call; index and md5(L) "
                "against tests/golden/golden_big.json (the reference's libsais path took 2.6 s on 8 threads, SURVEY 8a)"}
    del out

    # config 5: 128 MiB blocks (seed 3) through ST5 and ST6 + QLFC static, 8 blocks each through two contexts with two blocks in flight
    n5 = 128 << 20
    T3 = api.synth_text_v1(3, n5)
    d3 = torch.from_numpy(T3).to(dev)
    cxs = [GpuContext(local, max_n=n5 + 4096) for _ in range(2)]
    try:
        import threading
        pps = [cx.pipe(2, reuse_outputs=True) for cx in cxs]
        from libbsc_amd import _native as NN
        import ctypes as C

        def through(pipe, nblocks, k, out, slot):
            tickets, b5 = [], None
            for _ in range(nblocks):
                tickets.append(pipe.submit(d3, n5, k, 1, 3))
                if len(tickets) >= 2:
                    b5 = pipe.wait(tickets.pop(0))
            while tickets:
                b5 = pipe.wait(tickets.pop(0))
            out[slot] = b5
        for k in (5, 6):
            e5 = next(e for e in big if e["tag"] == "config5" and e["sorter"] == k)
            out = [None, None]
            for i, pp in enumerate(pps):
                through(pp, 2, k, out, i)                                # every buffer of the path once
            for cx in cxs:
                cx.profile(True); cx.profile_reset()
            torch.cuda.synchronize()
            NN.lib().bscgpu_coder_pool_expect(C.c_longlong(8), 1)
            t0 = time.perf_counter()
            ths = [threading.Thread(target=through, args=(pp, 4, k, out, i)) for i, pp in enumerate(pps)]
            for t in ths: t.start()
            for t in ths: t.join()
            dt5 = time.perf_counter() - t0
            b5 = out[0]
            st = None
            launches = []
            for cx in cxs:
                cx.profile(False)
                g = cx.profile_get()
                launches += cx.scatter_launches(65536)
                if st is None: st = g
                else:
                    for kk, vv in g.items():
                        for f in ("ms", "launches", "bytes", "records"): st[kk][f] += vv[f]
            ms_sort = sum(st[x]["ms"] for x in ("radix_hist_all", "radix_hist", "radix_scan", "radix_scatter") if x in st)
            full = [(ms, rec) for ms, rec in launches if rec == n5]
            b_sort = sum(16 * rec for _, rec in full) + 8 * n5 * max(st.get("radix_hist_all", {}).get("launches", 0), 0)
            res[f"config5_128MiB_st{k}_qlfc_static"] = {
                "MBps": round(8 * n5 / 1e6 / dt5, 1), "ms_per_block": round(dt5 / 8 * 1e3, 2), "blocks": 8,
                "sort_frac": round(b_sort / 1e6 / max(ms_sort, 1e-9) / HBM_PEAK_GBPS, 4),
                "digit_pass_frac": round(sum(16 * rec for _, rec in full) / 1e6 / max(sum(ms for ms, _ in full), 1e-9) / HBM_PEAK_GBPS, 4),
                "digit_passes_per_block": len(full) // 8,
                "compressed_bytes": int(b5.size), "verified": all((int(b.size), md5(b)) == (e5["size"], e5["md5"]) for b in out),
                "what": f"8 x 128 MiB synth-text v1 blocks (seed 3) through bscgpu_pipe_submit(ST{k}, QLFC static), two contexts x two blocks in flight, fill and "
                        f"drain included (a short job: the steady rate is higher); sort_frac = SURVEY 8d's B_sort (m*8 + {k}*2*m*8) over the time of every radix kernel "
                        f"(two contexts share the GPU while they are measured); checked against golden_big.json"}
        for pp in pps: pp.close()
    finally:
        for cx in cxs: cx.close()
    res["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
    return res


def verify_block(blk, seed, n, sorter, coder):
    """(True / False / None, note): size + md5 of a compressed block against tests/golden/golden_big.json."""
    import hashlib
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_big.json")))
    except Exception as e:
        return None, f"golden file unavailable: {e}"
    for e in g["blocks"]:
        if e["gen"] == {"kind": "synth", "seed": seed} and (e["n"], e["sorter"], e["coder"], e["features"]) == (n, sorter, coder, 3):
            ok = (int(blk.size), hashlib.md5(blk.tobytes()).hexdigest()) == (e["size"], e["md5"])
            return ok, ("last timed block of every rank: size + md5 equal the reference libbsc output committed in tests/golden/golden_big.json"
                        if ok else f"MISMATCH against tests/golden/golden_big.json for seed {seed}")
    return None, f"no committed reference output for seed {seed}, n {n}, sorter {sorter}, coder {coder}"


def cgroup_cpu_stat():
    """cgroup v2 CPU statistics of this container (usage and quota throttling), or None"""
    try:
        return {k: int(v) for k, v in (line.split() for line in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return None


def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup v2 cpu.max quota) — the 1-GPU box is a 16-CPU slice."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(host_in, sorter, coder):
    """The reference libbsc CPU path (oracle/_ref/libbsc_ref.so, compiled from /root/reference with its own flags)
    on the same block, best of 2 after a warm-up, MB = 1e6 bytes (bsc.cpp:427)."""
    try:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        from oracle.refbind import Ref
        threads = min(effective_cpus(), 32)
        os.environ["BSC_REF_THREADS"] = str(threads)
        ref = Ref()
        ref.compress(host_in[: 1 << 20], sorter, coder)            # warm-up (OpenMP team start)
        best = None
        for _ in range(2):                                          # (a) one call with the OpenMP team inside (features = 3)
            t0 = time.perf_counter()
            out = ref.compress(host_in, sorter, coder)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        single = host_in.size / 1e6 / best
        # (b) what the reference CLI does with many blocks (bsc.cpp:184-197): one block per thread, no threads inside a call
        import threading
        def one():
            ref.compress(host_in, sorter, coder, features=1)
        ths = [threading.Thread(target=one) for _ in range(threads)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dtp = time.perf_counter() - t0
        par = threads * host_in.size / 1e6 / dtp
        return {"value": round(max(single, par), 1), "unit": "MB/s", "cores": threads, "kind": "reference",
                "single_call_MBps": round(single, 1), "block_parallel_MBps": round(par, 1),
                "sample": f"the bench block ({host_in.size >> 20} MiB) through the reference's bsc_compress: (a) one call, features=3, "
                          f"{threads} OpenMP threads, best of 2 after warm-up: {best:.2f} s; (b) {threads} concurrent single-threaded calls "
                          f"(the CLI's block-parallel mode): {dtp:.2f} s; value = the better of the two; compressed {len(out)} B"}
    except Exception as e_ref:
        try:    # no compiled reference on this box: time the plain-C restatement (1 core) on a bounded sample
            import subprocess
            from oracle.refbind import Oracle, PORT_SO
            if not os.path.exists(PORT_SO):
                subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, capture_output=True)
            orc = Oracle()
            sample = host_in[: 2 << 20]
            t0 = time.perf_counter()
            out = orc.compress(sample, sorter, coder)
            dt = time.perf_counter() - t0
            return {"value": round(sample.size / 1e6 / dt, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                    "sample": f"first {sample.size >> 20} MiB of the bench block through oracle/bsc_oracle.c (qsort prefix-doubling BWT), "
                              f"{dt:.1f} s; compiled reference unavailable ({e_ref})"}
        except Exception as e:  # the baseline is reporting only; never fail the bench on it
            return {"value": None, "unit": "MB/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e_ref}; {e}"}


if __name__ == "__main__":
    main()
