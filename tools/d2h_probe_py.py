"""Probe (not part of the product), round 6: the same eight 45 MB hipMemcpyAsync D2H copies into a registered landing zone as
tools/d2h_probe.hip, issued from a Python process through ctypes — with and without torch in the process — to find out why bench.py's
copies are copy KERNELS (__amd_rocclr_copyBuffer) while job_bench's take the DMA engine.  Run under rocprofv3 --kernel-trace --memory-copy-trace.
  python tools/d2h_probe_py.py MODE     MODE: notorch | import | init | tensor | thread"""
import ctypes, mmap, sys, time, threading
mode = sys.argv[1] if len(sys.argv) > 1 else "notorch"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # hipMemcpyKind: 2 = DeviceToHost, 4 = Default, 1024 = DeviceToDeviceNoCU
if mode == "notorch":
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=ctypes.RTLD_GLOBAL)
elif mode == "torchlib":
    hip = ctypes.CDLL("/usr/local/lib/python3.10/dist-packages/torch/lib/libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
else:
    import torch
    if mode in ("init", "tensor", "thread"):
        torch.cuda.init()
    if mode in ("tensor", "thread"):
        t = torch.zeros(1 << 20, device="cuda:0"); torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so.7")          # resolves to the one torch loaded (same SONAME)
def ck(r, what):
    if r != 0: raise SystemExit("%s -> %d" % (what, r))
piece = 45 << 20; total = piece * 8
dev = ctypes.c_void_p()
ck(hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(total)), "hipMalloc")
buf = mmap.mmap(-1, total)
buf.write(b"\0" * total)
host = ctypes.addressof(ctypes.c_char.from_buffer(buf))
ck(hip.hipHostRegister(ctypes.c_void_p(host), ctypes.c_size_t(total), 0), "hipHostRegister")
def run():
    s = ctypes.c_void_p()
    ck(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1), "hipStreamCreateWithFlags")
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(8):
            ck(hip.hipMemcpyAsync(ctypes.c_void_p(host + k * piece), ctypes.c_void_p(dev.value + k * piece), ctypes.c_size_t(piece), kind, s), "hipMemcpyAsync")
        ck(hip.hipStreamSynchronize(s), "sync")
        dt = time.perf_counter() - t0
        print("%s kind %d: 8 x 45 MB D2H %.2f ms = %.1f GB/s" % (mode, kind, dt * 1e3, total / 1e9 / dt), flush=True)
if mode == "thread":
    th = threading.Thread(target=run); th.start(); th.join()
else:
    run()
