"""The C multi-GPU job driver (include/bscgpu.h: bscgpu_job_*, csrc/host/job.cpp) on CPU: its scheduler — one queue of blocks, one worker
per (device, context) pipe, `depth` blocks in flight per pipe, results collected in block order — runs here against a stand-in
executor (bscgpu_job_backend: python callbacks instead of bscgpu_create / bscgpu_pipe_*), so ordering, depth limits, load balance over
unequal devices, error propagation and teardown are checked without a GPU.  The stand-in "compresses" with bsc_store."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from libbsc_amd import _native as N

vp, ci = C.c_void_p, C.c_int
CTX_CREATE = C.CFUNCTYPE(ci, vp, C.POINTER(vp), ci, C.c_int64)
CTX_DESTROY = C.CFUNCTYPE(None, vp, vp)
PIPE_CREATE = C.CFUNCTYPE(ci, vp, vp, ci, C.POINTER(vp))
PIPE_DESTROY = C.CFUNCTYPE(None, vp, vp)
SUBMIT = C.CFUNCTYPE(ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci)
WAIT = C.CFUNCTYPE(ci, vp, vp, ci)


class Backend(C.Structure):
    _fields_ = [("user", vp), ("ctx_create", CTX_CREATE), ("ctx_destroy", CTX_DESTROY), ("pipe_create", PIPE_CREATE),
                ("pipe_destroy", PIPE_DESTROY), ("pipe_submit_host", SUBMIT), ("pipe_wait", WAIT)]


class FakeNode:
    """devices whose pipes take `ms[device]` milliseconds per block; records everything the driver does"""

    def __init__(self, ms, fail_device=None, fail_sorter=99):
        self.L = N.lib()
        self.L.bsc_store.argtypes = [vp, vp, ci, ci]
        self.ms, self.fail_device, self.fail_sorter = ms, fail_device, fail_sorter
        self.lock = threading.Lock()
        self.ctx = {}          # ctx id -> device
        self.pipes = {}        # pipe id -> dict(device, depth, tickets {ticket: (in, out, n)}, next, max_inflight, blocks)
        self.destroyed = []
        self.ids = 0
        self.submit_ms = 0.0   # time a submit (= the GPU stage of a block) takes
        self.stages = {}       # device -> submits that have returned
        self.seen = {}         # input address -> (features, stages of the device finished when the submit began)
        self.cb = Backend(None, CTX_CREATE(self.ctx_create), CTX_DESTROY(self.ctx_destroy), PIPE_CREATE(self.pipe_create),
                          PIPE_DESTROY(self.pipe_destroy), SUBMIT(self.submit), WAIT(self.wait))

    def ctx_create(self, user, out, device, max_n):
        if device == self.fail_device:
            return -9
        with self.lock:
            self.ids += 1
            self.ctx[self.ids] = device
            out[0] = self.ids
        return 0

    def ctx_destroy(self, user, ctx):
        with self.lock:
            self.destroyed.append(("ctx", ctx))

    def pipe_create(self, user, ctx, depth, out):
        with self.lock:
            self.ids += 1
            self.pipes[self.ids] = dict(device=self.ctx[ctx], depth=depth, tickets={}, next=0, max_inflight=0, blocks=0)
            out[0] = self.ids
        return 0

    def pipe_destroy(self, user, pipe):
        with self.lock:
            assert not self.pipes[pipe]["tickets"], "pipe destroyed with blocks in flight"
            self.destroyed.append(("pipe", pipe))

    def submit(self, user, pipe, inp, out, n, lh, lm, sorter, coder, features):
        if sorter == self.fail_sorter:
            return -1
        with self.lock:
            dev = self.pipes[pipe]["device"]
            self.seen[inp] = (features, self.stages.get(dev, 0))
        if self.submit_ms:
            time.sleep(self.submit_ms / 1e3)
        with self.lock:
            self.stages[dev] = self.stages.get(dev, 0) + 1
            p = self.pipes[pipe]
            t = p["next"]; p["next"] += 1
            p["tickets"][t] = (inp, out, n)
            p["max_inflight"] = max(p["max_inflight"], len(p["tickets"]))
            p["blocks"] += 1
        return t

    def wait(self, user, pipe, ticket):
        with self.lock:
            p = self.pipes[pipe]
            inp, out, n = p["tickets"][ticket]
        time.sleep(self.ms[p["device"]] / 1e3)
        r = self.L.bsc_store(inp, out, n, 0)
        with self.lock:
            del p["tickets"][ticket]
        return r


def _job(node, devices, cpd, depth, max_block=1 << 20):
    L = N.lib()
    L.bscgpu_job_create_ex.argtypes = [C.POINTER(vp), C.POINTER(ci), ci, ci, ci, C.c_int64, C.POINTER(Backend)]
    L.bscgpu_job_add.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci]
    L.bscgpu_job_wait.argtypes = [vp, ci]
    L.bscgpu_job_block_worker.argtypes = [vp, ci, C.POINTER(ci)]
    L.bscgpu_job_destroy.argtypes = [vp]
    L.bscgpu_job_destroy.restype = None
    h = vp()
    devs = (ci * len(devices))(*devices)
    rc = L.bscgpu_job_create_ex(C.byref(h), devs, len(devices), cpd, depth, max_block, C.byref(node.cb))
    return L, h, rc


def test_job_collects_blocks_in_order_over_unequal_devices():
    node = FakeNode({0: 2.0, 1: 2.0, 2: 12.0, 3: 2.0})           # device 2 is six times slower
    L, h, rc = _job(node, [0, 1, 2, 3], 2, 3)
    assert rc == 0 and len(node.pipes) == 8
    rng = np.random.default_rng(1)
    ins = [rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8) for _ in range(120)]
    outs = [np.zeros(a.size + 28, np.uint8) for a in ins]
    for b, (a, o) in enumerate(zip(ins, outs)):
        assert L.bscgpu_job_add(h, N.np_ptr(a), N.np_ptr(o), a.size, 0, 0, 1, 1, 3) == b      # numbers follow the order of the calls
    per_dev = {}
    for b, (a, o) in enumerate(zip(ins, outs)):                 # collected in block order, whatever order they finished in
        assert L.bscgpu_job_wait(h, b) == a.size + 28
        assert o[28:].tobytes() == a.tobytes() and int.from_bytes(o[4:8].tobytes(), "little") == a.size
        dev = ci(-1)
        w = L.bscgpu_job_block_worker(h, b, C.byref(dev))
        assert 0 <= w < 8 and dev.value == [0, 1, 2, 3][w % 4]   # worker w lives on device w % ndev: first contexts before second ones
        per_dev[dev.value] = per_dev.get(dev.value, 0) + 1
    assert all(p["max_inflight"] <= 3 for p in node.pipes.values())
    assert max(p["max_inflight"] for p in node.pipes.values()) == 3          # and the depth is used
    assert set(per_dev) == {0, 1, 2, 3} and per_dev[2] < min(per_dev[0], per_dev[1], per_dev[3])     # the queue balances: the slow GPU takes fewer
    assert L.bscgpu_job_wait(h, 120) == -1 and L.bscgpu_job_wait(h, -1) == -1
    L.bscgpu_job_destroy(h)
    assert len([d for d in node.destroyed if d[0] == "pipe"]) == 8 and len([d for d in node.destroyed if d[0] == "ctx"]) == 8


def test_job_destroy_finishes_queued_blocks_and_errors_reach_the_caller():
    node = FakeNode({0: 3.0, 1: 3.0}, fail_sorter=7)
    L, h, rc = _job(node, [0, 1], 1, 2)
    assert rc == 0
    ins = [np.full(100 + i, i, np.uint8) for i in range(30)]
    outs = [np.zeros(a.size + 28, np.uint8) for a in ins]
    for b, (a, o) in enumerate(zip(ins, outs)):
        L.bscgpu_job_add(h, N.np_ptr(a), N.np_ptr(o), a.size, 0, 0, 7 if b == 11 else 1, 1, 3)
    assert L.bscgpu_job_wait(h, 11) == -1                        # the executor's error code is the block's result; the job goes on
    assert L.bscgpu_job_wait(h, 12) == ins[12].size + 28
    assert L.bscgpu_job_add(h, N.np_ptr(ins[0]), N.np_ptr(outs[0]), (1 << 20) + 1, 0, 0, 1, 1, 3) == -1     # larger than the job's contexts
    L.bscgpu_job_destroy(h)                                      # nothing waited for beyond 12: destroy = finish, then tear down
    for b, (a, o) in enumerate(zip(ins, outs)):
        if b != 11:
            assert o[28:].tobytes() == a.tobytes(), b
    assert sum(p["blocks"] for p in node.pipes.values()) == 29


def test_job_head_and_tail_are_tapered_when_the_total_is_announced():
    """bscgpu_job_expect: the k-th context of a device draws its FIRST block once k GPU stages have finished there, takes a block only
    while more than k x devices are left, and the last devices x contexts blocks are marked low-latency; without the call (or with more
    blocks than announced) every block is still processed."""
    LOW = 0x10000
    ndev, cpd, total = 2, 3, 26
    node = FakeNode({0: 4.0, 1: 4.0})
    node.submit_ms = 1.0
    L, h, rc = _job(node, [0, 1], cpd, 2)
    assert rc == 0
    L.bscgpu_job_expect.argtypes = [vp, ci]
    assert L.bscgpu_job_expect(h, total) == 0 and L.bscgpu_job_expect(h, -1) == -1
    ins = [np.full(64 + b, b, np.uint8) for b in range(total)]
    outs = [np.zeros(a.size + 28, np.uint8) for a in ins]
    for b, (a, o) in enumerate(zip(ins, outs)):
        assert L.bscgpu_job_add(h, N.np_ptr(a), N.np_ptr(o), a.size, 0, 0, 1, 1, 3) == b
    first = {}
    for b, (a, o) in enumerate(zip(ins, outs)):
        assert L.bscgpu_job_wait(h, b) == a.size + 28 and o[28:].tobytes() == a.tobytes()
        dev = ci(-1)
        w = L.bscgpu_job_block_worker(h, b, C.byref(dev))
        k = w // ndev                                            # the worker is its device's k-th context
        assert total - b > k * ndev, (b, w)                      # tail: the last block to a first context, the last 2 x ndev to first and second ...
        feat, stages_before = node.seen[a.ctypes.data]
        assert bool(feat & LOW) == (total - b <= ndev * cpd), b  # the last devices x contexts blocks are low-latency
        assert feat & 0xffff == 3
        if w not in first:
            first[w] = b                                          # (a context's very first block is exempt from the head rule: cold set-up runs side by side)
    assert set(first) == set(range(ndev * cpd))                  # the middle of the job uses every context
    # a second burst on the now warm contexts (the job ran dry above): a first context begins, the k-th joins after k stages of the burst
    base = dict(node.stages)
    total2 = total + 14
    assert L.bscgpu_job_expect(h, total2) == 0
    more = [np.full(70 + i, 3, np.uint8) for i in range(14)]
    mouts = [np.zeros(a.size + 28, np.uint8) for a in more]
    for i, (a, o) in enumerate(zip(more, mouts)):
        assert L.bscgpu_job_add(h, N.np_ptr(a), N.np_ptr(o), a.size, 0, 0, 1, 1, 3) == total + i
    started = set()
    for i, (a, o) in enumerate(zip(more, mouts)):
        assert L.bscgpu_job_wait(h, total + i) == a.size + 28
        dev = ci(-1)
        w = L.bscgpu_job_block_worker(h, total + i, C.byref(dev))
        if w not in started:
            started.add(w)
            assert node.seen[a.ctypes.data][1] - base[dev.value] >= w // ndev, (i, w)
    # more blocks than announced: the tail rule is dropped, nothing is left behind
    extra = [np.full(50, 7, np.uint8) for _ in range(5)]
    eouts = [np.zeros(78, np.uint8) for _ in extra]
    for i, (a, o) in enumerate(zip(extra, eouts)):
        assert L.bscgpu_job_add(h, N.np_ptr(a), N.np_ptr(o), a.size, 0, 0, 1, 1, 3) == total2 + i
    L.bscgpu_job_destroy(h)
    assert all(o[28:].tobytes() == a.tobytes() for a, o in zip(extra, eouts))
    assert sum(p["blocks"] for p in node.pipes.values()) == total + 5 + 14


def test_job_creation_fails_as_a_whole_when_a_device_cannot_be_set_up():
    node = FakeNode({0: 1.0, 1: 1.0, 2: 1.0}, fail_device=1)
    L, h, rc = _job(node, [0, 1, 2], 1, 2)
    assert rc == -9 and not h.value
    assert len([d for d in node.destroyed if d[0] == "ctx"]) == 2            # what had been created is released again
    node2 = FakeNode({0: 1.0})
    assert _job(node2, [], 1, 2)[2] == -1                                    # "every visible device" is the default executor's notion
    assert _job(node2, [0], 0, 2)[2] == -1 and _job(node2, [0], 1, 9)[2] == -1


def test_cxx_file_compressor_is_built_with_the_library():
    """csrc/driver/bsc_mgpu.cpp (the reference CLI's block loop over bscgpu_job_*) is compiled by libbsc_amd.build next to the library;
    without arguments it explains itself (its GPU test compares its files with the reference CLI's)."""
    import os
    import subprocess
    from libbsc_amd.build import DRIVER_EXE, build
    build(verbose=False)
    assert os.path.exists(DRIVER_EXE)
    r = subprocess.run([DRIVER_EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "bsc_mgpu e <in> <out>" in r.stderr


def test_job_scheduler_under_thread_sanitizer(tmp_path):
    """tools/job_tsan_check.cpp: job.cpp itself compiled with -fsanitize=thread against a stand-in executor (threads adding with pauses so
    that bursts restart, an in-order collector, announced totals on every other job, a failing block): no data race report, every block
    collected, head / tail rules hold."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "job_tsan_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-Wno-subobject-linkage", "-I", os.path.join(root, "include"),
                        os.path.join(root, "tools", "job_tsan_check.cpp"), "-o", exe, "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower():
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")}
    for _ in range(3):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "0 failure(s)" in r.stdout and "ThreadSanitizer:" not in r.stderr, r.stdout + r.stderr[-3000:]
