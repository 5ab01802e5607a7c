// Probe (not part of the product), round 6: WHO moves a block's probability stream to the host, and what it costs the kernels beside it.
// The kernel trace of bench.py shows the eight 45 MB device-to-host copies of every block as __amd_rocclr_copyBuffer launches of 256
// workgroups x 1024 threads: the runtime's copy KERNEL, not a DMA engine — sixteen wavefronts resident on every CU for the ~0.8 ms
// the PCIe link needs per copy, i.e. for ~7 of the 11.4 ms a block takes.
//   part 1: hipMemcpyAsync D2H of 45 MB x 8 into (a) hipHostMalloc memory, (b) mmap + hipHostRegister memory (the product's landing
//           zone): GB/s; run it under `rocprofv3 --kernel-trace --memory-copy-trace --stats` to see which path each takes, and under the
//           runtime's environment knobs (GPU_FORCE_BLIT_COPY_SIZE, DEBUG_CLR_LIMIT_BLIT_WG, HSA_ENABLE_SDMA ...);
//   part 2: a copy kernel of our own writing to the device-mapped landing zone with W workgroups of 256 threads: how few saturate
//           the link;
//   part 3: an HBM-bound kernel (the digit pass's shape: 256 workgroups x 1024 threads, 24 B per record) timed alone, beside the
//           runtime's copy, and beside our W-workgroup copy.
// hipcc -O3 --offload-arch=gfx950 tools/d2h_probe.hip -o tools/bin/d2h_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
typedef unsigned long long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef u32 v4u __attribute__((ext_vector_type(4)));
// W workgroups of 256 threads, 16-byte accesses, four in flight per thread
__global__ __launch_bounds__(256) void k_d2h(const v4u* __restrict__ in, v4u* __restrict__ out, u64 n16)
{
    const u64 stride = (u64)gridDim.x * 256 * 4;
    u64 i = (u64)blockIdx.x * 256 * 4 + threadIdx.x;
    for (; i + 768 < n16; i += stride) {
        const v4u a = __builtin_nontemporal_load(&in[i]), b = __builtin_nontemporal_load(&in[i + 256]), c = __builtin_nontemporal_load(&in[i + 512]), d = __builtin_nontemporal_load(&in[i + 768]);
        out[i] = a; out[i + 256] = b; out[i + 512] = c; out[i + 768] = d;
    }
    for (; i < n16; i += 256) out[i] = in[i];
}
// the HBM-bound neighbour
__global__ __launch_bounds__(1024) void k_copy(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout, u32 n)
{
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (u64)gridDim.x * 1024) { kout[i] = __builtin_nontemporal_load(&kin[i]); vout[i] = __builtin_nontemporal_load(&vin[i]); }
}

// NUMA node a page lives on (move_pages in query mode), -1 if unknown
static int node_of(void* p)
{
    void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095); int status = -1;
    if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0) return -1;
    return status;
}

int main(int argc, char** argv)
{
    const size_t piece = (size_t)45 << 20, total = piece * 8;
    const int part = argc > 1 ? atoi(argv[1]) : 0;              // 0 = all
    const size_t off = argc > 2 ? (size_t)atoll(argv[2]) : 0;   // part 1: byte offset of every piece inside both buffers ...
    const size_t cut = argc > 3 ? (size_t)atoll(argv[3]) : 0;   // ... and bytes taken off every piece's length (alignment experiments)
    const int which = argc > 4 ? atoi(argv[4]) : 0;             // part 1: 1 = hipHostMalloc memory only, 2 = registered memory only (to tell them apart in a trace)
    unsigned char* dev; CHECK(hipMalloc(&dev, total)); CHECK(hipMemset(dev, 0x5a, total));
    void* hm = nullptr; CHECK(hipHostMalloc(&hm, total, hipHostMallocDefault));
    void* hr = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(hr, total, MADV_HUGEPAGE); memset(hr, 0, total);
    CHECK(hipHostRegister(hr, total, hipHostRegisterDefault));
    void* hr_dev = nullptr; CHECK(hipHostGetDevicePointer(&hr_dev, hr, 0));
    hipStream_t s, s2; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

    auto copy8 = [&](void* host) {
        for (int k = 0; k < 8; ++k) CHECK(hipMemcpyAsync((char*)host + k * piece + off, dev + k * piece + off, piece - off - cut, hipMemcpyDeviceToHost, s));
    };
    if (part == 0 || part == 1) {
        printf("this thread runs on CPU %d; host buffers live on NUMA node %d (hipHostMalloc) and %d (registered mmap; its last page: %d)\n", sched_getcpu(), node_of(hm), node_of(hr), node_of((char*)hr + total - 4096));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now(); if (which != 2) copy8(hm); CHECK(hipStreamSynchronize(s)); double t1 = now();
            if (which != 1) copy8(hr); CHECK(hipStreamSynchronize(s)); double t2 = now();
            printf("hipMemcpyAsync D2H 8 x 45 MB: hipHostMalloc %.2f ms = %.1f GB/s   registered mmap %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, total / 1e9 / (t1 - t0), (t2 - t1) * 1e3, total / 1e9 / (t2 - t1));
        }
        if (off || cut) printf("(pieces at byte offset %zu, %zu bytes short)\n", off, cut);
        // one 360 MB copy
        double t0 = now(); if (which != 1) CHECK(hipMemcpyAsync(hr, dev, total, hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s)); double t1 = now();
        printf("hipMemcpyAsync D2H 1 x 360 MB registered: %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, total / 1e9 / (t1 - t0));
    }
    if (part == 0 || part == 2) {
        for (int W : {1, 2, 4, 8, 16, 32, 64, 256}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                hipLaunchKernelGGL(k_d2h, dim3(W), dim3(256), 0, s, (const v4u*)dev, (v4u*)hr_dev, (u64)(total / 16));
                CHECK(hipStreamSynchronize(s));
                double t = now() - t0; if (t < best) best = t;
            }
            printf("own copy kernel to the mapped landing zone, %3d workgroups of 256: %.2f ms = %.1f GB/s\n", W, best * 1e3, total / 1e9 / best);
        }
        if (memcmp(hr, hm, 4096) != 0 && part == 0) printf("MISMATCH\n");
    }
    if (part == 0 || part == 3) {
        const u32 n = 64u << 20;
        u64 *ka, *kb; u32 *va, *vb;
        CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull)); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull));
        CHECK(hipMemset(ka, 1, n * 8ull)); CHECK(hipMemset(va, 1, n * 4ull));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto neighbour = [&](const char* what) {
            // 16 launches back to back on s2 while whatever was started on s runs
            CHECK(hipEventRecord(e0, s2));
            for (int r = 0; r < 16; ++r) hipLaunchKernelGGL(k_copy, dim3(256), dim3(1024), 0, s2, ka, va, kb, vb, n);
            CHECK(hipEventRecord(e1, s2)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("HBM-bound neighbour (256 x 1024 threads, 24 B per record, 64 Mi records) %-46s: %.3f ms per launch = %.0f GB/s\n", what, ms / 16, 24.0 * n / 1e6 / (ms / 16));
        };
        neighbour("alone"); neighbour("alone");
        for (int rep = 0; rep < 2; ++rep) {
            copy8(hr); copy8(hr); neighbour("beside the runtime's D2H copies"); CHECK(hipStreamSynchronize(s));
        }
        for (int W : {4, 16, 64}) {
            for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_d2h, dim3(W), dim3(256), 0, s, (const v4u*)dev, (v4u*)hr_dev, (u64)(total / 16));
            char what[64]; snprintf(what, sizeof what, "beside our copy kernel, %d workgroups", W);
            neighbour(what); CHECK(hipStreamSynchronize(s));
        }
    }
    if (part == 4) {
        // the DMA engine's rate over time while the chip is busy: 8 x 45 MB copies back to back on s for `secs` seconds, an HBM-bound kernel
        // looping on s2 the whole time; one line per 0.25 s.  (A job of 160 blocks ran at 11.1 ms or at 12.9 ms per block with the copies on the
        // DMA engine: 12.9 ms is what 366 MB per block take at HALF the link rate.)
        const double secs = argc > 2 ? atof(argv[2]) : 6.0;
        const int load = argc > 3 ? atoi(argv[3]) : 1;          // 0: no neighbour, 1: HBM-bound copy kernel, 2: two streams of it
        const u32 n = 64u << 20;
        u64 *ka, *kb; u32 *va, *vb;
        CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull)); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull));
        CHECK(hipMemset(ka, 1, n * 8ull)); CHECK(hipMemset(va, 1, n * 4ull));
        hipStream_t s3; CHECK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
        hipEvent_t ev[2]; CHECK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        const double t_begin = now(); double t_mark = t_begin; size_t bytes_mark = 0; long launches = 0, launches_mark = 0;
        int inflight = 0;
        while (now() - t_begin < secs) {
            if (load) {
                // keep ~8 launches queued on the neighbour stream(s)
                for (int k = 0; k < 4; ++k) { hipLaunchKernelGGL(k_copy, dim3(256), dim3(1024), 0, s2, ka, va, kb, vb, n); ++launches; if (load > 1) { hipLaunchKernelGGL(k_copy, dim3(256), dim3(1024), 0, s3, ka, va, kb, vb, n); ++launches; } }
                CHECK(hipEventRecord(ev[inflight & 1], s2)); ++inflight;
                if (inflight >= 2) CHECK(hipEventSynchronize(ev[inflight & 1]));
            }
            copy8(hr); CHECK(hipStreamSynchronize(s)); bytes_mark += total;
            const double t = now();
            if (t - t_mark >= 0.25) {
                printf("t %5.2f s: D2H %5.1f GB/s   neighbour launches %4ld (%.3f ms each if serial)\n", t - t_begin, bytes_mark / 1e9 / (t - t_mark), launches - launches_mark, (launches - launches_mark) ? (t - t_mark) * 1e3 / (launches - launches_mark) : 0.0);
                t_mark = t; bytes_mark = 0; launches_mark = launches;
            }
        }
        CHECK(hipDeviceSynchronize());
    }
    return 0;
}
