// Microbenchmarks for the memory shapes used by the radix engine (not part of the product).
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read16(const uint4* __restrict__ a, u32* __restrict__ out, size_t n16) {
    u32 acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345) out[0] = acc;
}
// chunked: WG c owns a contiguous chunk and walks it tile by tile (4096 keys), wave-striped 8-B loads like rs_scatter
template <int MODE>   // 0: read keys only; 1: read keys+vals; 2: copy keys+vals to same index; 3: read keys w/ LDS atomics (hist)
__global__ __launch_bounds__(256) void k_chunked(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                                 u32 n, u32 chunk_tiles, u32* __restrict__ sink) {
    __shared__ u32 h[1024];
    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;
    if (MODE == 3) { for (int i = t; i < 1024; i += 256) h[i] = 0; __syncthreads(); }
    u64 acc = 0;
    const u32 tile0 = blockIdx.x * chunk_tiles;
    for (u32 tile = tile0; tile < tile0 + chunk_tiles; ++tile) {
        const u64 tb = (u64)tile * 4096;
        if (tb >= n) break;
        u64 k[16]; u32 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) k[i] = kin[tb + w * 1024 + i * 64 + lane];
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = vin[tb + w * 1024 + i * 64 + lane];
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { kout[tb + w * 1024 + i * 64 + lane] = k[i]; vout[tb + w * 1024 + i * 64 + lane] = v[i]; }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&h[w * 256 + (u32)(k[i] >> 24 & 255)], 1u);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc += k[i]; if (MODE == 1) acc += v[i]; }
        }
    }
    if (MODE == 3) { __syncthreads(); if (h[t] == 0x7fffffff) sink[0] = 1; }
    if (acc == 0x1234567) sink[0] = (u32)acc;
}
// scatter-shaped writes: each WG writes runs of `run` keys to 256 streams (bucket b base = b * n/256), like a uniform digit pass
__global__ __launch_bounds__(256) void k_scatter_runs(const u64* __restrict__ kin, u64* __restrict__ kout, u32 n, u32 chunk_tiles, u32 num_chunks, u32 misalign) {
    const u32 t = threadIdx.x;
    const u32 per_bucket = n / 256 - 16;
    kout += misalign;
    for (u32 tt = 0; tt < chunk_tiles; ++tt) {
        const u32 tile = blockIdx.x * chunk_tiles + tt;
        const u64 tb = (u64)tile * 4096;
        if (tb >= n) break;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 q = j * 256 + t;               // position in the locally sorted tile
            const u32 bucket = q >> 4, r = q & 15;   // 16 keys per bucket per tile
            const u64 key = kin[tb + q];
            kout[(u64)bucket * per_bucket + (u64)tile * 16 + r] = key;
        }
    }
}
// pairs: per tile and bucket one aligned 128-B key line + one aligned 64-B value half-line (what a write-combining
// scatter would emit on uniform digits); LDSPAD bytes of dynamic LDS limit residency (81920 -> 1 WG/CU, 40960 -> 3)
extern __shared__ unsigned char pad_lds[];
__global__ __launch_bounds__(256) void k_scatter_pairs_aligned(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                                               u32 n, u32 chunk_tiles, u32 misalign, u32* sink) {
    const u32 t = threadIdx.x;
    if (pad_lds[t] == 77 && n == 1) sink[0] = 1;
    const u32 per_bucket = n / 256 - 16;
    kout += misalign; vout += misalign;
    for (u32 tt = 0; tt < chunk_tiles; ++tt) {
        const u32 tile = blockIdx.x * chunk_tiles + tt;
        const u64 tb = (u64)tile * 4096;
        if (tb >= n) break;
        u64 k[16]; u32 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { k[j] = __builtin_nontemporal_load(&kin[tb + j * 256 + t]); }
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = __builtin_nontemporal_load(&vin[tb + j * 256 + t]); }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 q = j * 256 + t; const u32 bucket = q >> 4, r = q & 15;
            const u64 o = (u64)bucket * per_bucket + (u64)tile * 16 + r;
            kout[o] = k[j]; vout[o] = v[j];
        }
    }
}
// what a write-combining scatter would emit: every (tile, bucket) flush is 32 records = 2 full key lines + 1 full value line;
// a bucket is flushed every other tile (256 buckets, 4096-record tiles, 16 records/bucket/tile on average)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_scatter_pairs_comb(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                                                u32 n, u32 chunk_tiles, u32* sink) {
    constexpr int ITEMS = 4096 / THREADS;
    const u32 t = threadIdx.x;
    if (pad_lds[t] == 77 && n == 1) sink[0] = 1;
    const u32 per_bucket = (n / 256) & ~31u;
    for (u32 tt = 0; tt < chunk_tiles; ++tt) {
        const u32 tile = blockIdx.x * chunk_tiles + tt;
        const u64 tb = (u64)tile * 4096;
        if (tb >= n) break;
        u64 k[ITEMS]; u32 v[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { k[j] = __builtin_nontemporal_load(&kin[tb + j * THREADS + t]); }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { v[j] = __builtin_nontemporal_load(&vin[tb + j * THREADS + t]); }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 q = j * THREADS + t; const u32 bucket = ((q >> 5) << 1) | (tt & 1), r = q & 31;
            const u64 o = (u64)bucket * per_bucket + (u64)(blockIdx.x * chunk_tiles + (tt & ~1u)) * 16 + r;
            kout[o] = k[j]; vout[o] = v[j];
        }
    }
}
// misaligned 16-record runs (what the plain scatter emits on uniform digits) at different residency: does L2 merge the
// partial lines of consecutive tiles when the open-line frontier (workgroups x 256 digits x 2 lines) fits in the 4 MB L2?
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_scatter_pairs_mis(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                                               u32 n, u32 chunk_tiles, u32 misalign, u32* sink) {
    constexpr int ITEMS = 4096 / THREADS;
    const u32 t = threadIdx.x;
    if (pad_lds[t] == 77 && n == 1) sink[0] = 1;
    const u32 per_bucket = n / 256 - 16;
    kout += misalign; vout += misalign;
    for (u32 tt = 0; tt < chunk_tiles; ++tt) {
        const u32 tile = blockIdx.x * chunk_tiles + tt;
        const u64 tb = (u64)tile * 4096;
        if (tb >= n) break;
        u64 k[ITEMS]; u32 v[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { k[j] = __builtin_nontemporal_load(&kin[tb + j * THREADS + t]); }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { v[j] = __builtin_nontemporal_load(&vin[tb + j * THREADS + t]); }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 q = j * THREADS + t; const u32 bucket = q >> 4, r = q & 15;
            const u64 o = (u64)bucket * per_bucket + (u64)tile * 16 + r;
            kout[o] = k[j]; vout[o] = v[j];
        }
    }
}
// inverse-permutation scatter dst[idx[i]] = i (what ISA[SA[j]] = rank does): fully random over 256 MB, and the same number
// of stores when the records have first been bucketed by destination so that each XCD (blockIdx % 8) works inside one
// window of `win` elements at a time (32 workgroups of the XCD share a window's records)
__global__ __launch_bounds__(256) void k_scatter_random(const u32* __restrict__ idx, u32* __restrict__ dst, u32 n) {
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[idx[i]] = i;
}
__global__ __launch_bounds__(256) void k_scatter_windowed(const u32* __restrict__ idx, u32* __restrict__ dst, u32 n, u32 win, u32 wgs_per_xcd) {
    const u32 xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const u32 nwin = n / win;
    for (u32 w = xcd; w < nwin; w += 8) {                         // records of window w are idx[w*win .. (w+1)*win)
        const u32 base = w * win;
        for (u32 i = slot * 256 + threadIdx.x; i < win; i += wgs_per_xcd * 256) dst[idx[base + i]] = base + i;
    }
}
// agent-scope hand-off chain: hop h (owned by workgroup h % gridDim.x) waits for flag[h-1], then sets flag[h].  With
// `load` > 0 the other waves of every workgroup stream-copy meanwhile.  Bounded spins: a stuck chain gives up, it cannot hang.
__global__ __launch_bounds__(256) void k_handoff(u32* flag, u32 hops, const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16, u32 load, u32* fail) {
    if (threadIdx.x >= 64) {                          // waves 1..3: background streaming traffic
        if (!load) return;
        const u32 t = threadIdx.x - 64;
        for (u32 rep = 0; rep < load; ++rep)
            for (size_t i = (size_t)blockIdx.x * 192 + t; i < n16; i += (size_t)gridDim.x * 192) b[i] = a[i];
        return;
    }
    if (threadIdx.x != 0) return;
    for (u32 h = blockIdx.x; h < hops; h += gridDim.x) {
        if (h > 0) {
            u32 spins = 0;
            while (__hip_atomic_load(&flag[h - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++spins > (1u << 24)) { *fail = h; return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(&flag[h], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <class F> static float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    const u32 n = 64u << 20;
    u64 *ka, *kb; u32 *va, *vb, *sink;
    CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull)); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(ka, 0x5a, n * 8ull)); CHECK(hipMemset(va, 1, n * 4ull));
    const size_t n16 = n * 8ull / 16;
    for (int grid : {1024, 2048, 8192}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)ka, (uint4*)kb, n16); });
        printf("copy16 grid-stride grid=%d: %.3f ms -> %.0f GB/s\n", grid, ms, 2.0 * n * 8 / 1e6 / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const uint4*)ka, sink, n16); });
        printf("read16 grid-stride grid=%d: %.3f ms -> %.0f GB/s\n", grid, ms, 1.0 * n * 8 / 1e6 / ms);
    }
    const u32 ct = 16, nc = 1024;
    float ms = timeit([&] { hipLaunchKernelGGL(k_chunked<0>, dim3(nc), dim3(256), 0, 0, ka, va, kb, vb, n, ct, sink); });
    printf("chunked read keys (8B/lane striped): %.3f ms -> %.0f GB/s\n", ms, 8.0 * n / 1e6 / ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_chunked<1>, dim3(nc), dim3(256), 0, 0, ka, va, kb, vb, n, ct, sink); });
    printf("chunked read keys+vals: %.3f ms -> %.0f GB/s\n", ms, 12.0 * n / 1e6 / ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_chunked<2>, dim3(nc), dim3(256), 0, 0, ka, va, kb, vb, n, ct, sink); });
    printf("chunked copy keys+vals (same index): %.3f ms -> %.0f GB/s\n", ms, 24.0 * n / 1e6 / ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_chunked<3>, dim3(nc), dim3(256), 0, 0, ka, va, kb, vb, n, ct, sink); });
    printf("chunked read keys + LDS atomics (all same digit): %.3f ms -> %.0f GB/s\n", ms, 8.0 * n / 1e6 / ms);
    for (u32 mis : {0u, 5u, 8u}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_runs, dim3(nc), dim3(256), 0, 0, ka, kb, n, ct, nc, mis); });
        printf("keys: read + write 128-B runs into 256 streams, misaligned by %u keys: %.3f ms -> %.0f GB/s\n", mis, ms, 16.0 * n / 1e6 / ms);
    }
    CHECK(hipFuncSetAttribute((const void*)k_scatter_pairs_aligned, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (u32 lds : {0u, 40000u, 64000u, 81000u}) for (u32 mis : {0u, 5u}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_aligned, dim3(nc), dim3(256), lds, 0, ka, va, kb, vb, n, ct, mis, sink); });
        printf("pairs: 16 records/bucket/tile, misalign %u records, dyn LDS %u B: %.3f ms -> %.0f GB/s\n", mis, lds, ms, 24.0 * n / 1e6 / ms);
    }
    CHECK(hipFuncSetAttribute((const void*)k_scatter_pairs_comb<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k_scatter_pairs_comb<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (u32 lds : {0u, 48000u, 78000u, 112000u}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_comb<256>, dim3(nc), dim3(256), lds, 0, ka, va, kb, vb, n, ct, sink); });
        printf("pairs write-combined (32-record aligned flushes), 256 thr, dyn LDS %u B: %.3f ms -> %.0f GB/s\n", lds, ms, 24.0 * n / 1e6 / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_comb<512>, dim3(nc), dim3(512), lds, 0, ka, va, kb, vb, n, ct, sink); });
        printf("pairs write-combined (32-record aligned flushes), 512 thr, dyn LDS %u B: %.3f ms -> %.0f GB/s\n", lds, ms, 24.0 * n / 1e6 / ms);
    }
    CHECK(hipFuncSetAttribute((const void*)k_scatter_pairs_mis<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k_scatter_pairs_mis<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (u32 mis : {0u, 5u}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_mis<1024>, dim3(256), dim3(1024), 100000, 0, ka, va, kb, vb, n, 64, mis, sink); });
        printf("pairs 16-record runs misalign %u: 256 WG x 1024 thr (1 WG/CU, frontier 2 MB/XCD): %.3f ms -> %.0f GB/s\n", mis, ms, 24.0 * n / 1e6 / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_mis<1024>, dim3(512), dim3(1024), 60000, 0, ka, va, kb, vb, n, 32, mis, sink); });
        printf("pairs 16-record runs misalign %u: 512 WG x 1024 thr (2 WG/CU, frontier 4 MB/XCD): %.3f ms -> %.0f GB/s\n", mis, ms, 24.0 * n / 1e6 / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_mis<256>, dim3(256), dim3(256), 100000, 0, ka, va, kb, vb, n, 64, mis, sink); });
        printf("pairs 16-record runs misalign %u: 256 WG x 256 thr (1 WG/CU): %.3f ms -> %.0f GB/s\n", mis, ms, 24.0 * n / 1e6 / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_pairs_mis<256>, dim3(1024), dim3(256), 0, 0, ka, va, kb, vb, n, 16, mis, sink); });
        printf("pairs 16-record runs misalign %u: 1024 WG x 256 thr (4 WG/CU, frontier 8 MB/XCD): %.3f ms -> %.0f GB/s\n", mis, ms, 24.0 * n / 1e6 / ms);
    }
    {   // what a decoupled look-back chain would pay per hop between workgroups on different XCDs
        const u32 hops = 8192; u32 *flag, *dfail; CHECK(hipMalloc(&flag, hops * 4)); CHECK(hipMalloc(&dfail, 4));
        for (u32 load : {0u, 4u}) {
            float best = 1e9; u32 hf = 0;
            for (int r = 0; r < 3; ++r) {
                CHECK(hipMemset(flag, 0, hops * 4)); CHECK(hipMemset(dfail, 0, 4)); CHECK(hipDeviceSynchronize());
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_handoff, dim3(256), dim3(256), 0, 0, flag, hops, (const uint4*)ka, (uint4*)kb, n16, load, dfail);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
                CHECK(hipMemcpy(&hf, dfail, 4, hipMemcpyDeviceToHost));
            }
            float copy_ms = 0;
            if (load) copy_ms = timeit([&] { hipLaunchKernelGGL(k_handoff, dim3(256), dim3(256), 0, 0, flag, 0u, (const uint4*)ka, (uint4*)kb, n16, load, dfail); });
            printf("hand-off chain, %u hops over 256 workgroups (8 XCDs), %s: %.3f ms total -> %.2f us per hop%s (copy alone %.3f ms)\n",
                   hops, load ? "under streaming load" : "idle machine", best, (best - copy_ms > 0 && load ? best : best) * 1e3 / hops, hf ? " [chain gave up]" : "", copy_ms);
        }
    }
    {   // idx: a random permutation (host-built), and one that only permutes inside windows
        std::vector<u32> h(n);
        for (u32 i = 0; i < n; ++i) h[i] = i;
        u64 st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (u32 i = n - 1; i > 0; --i) { u32 j = (u32)(rnd() % (i + 1)); u32 t2 = h[i]; h[i] = h[j]; h[j] = t2; }
        u32* didx; CHECK(hipMalloc(&didx, n * 4ull)); CHECK(hipMemcpy(didx, h.data(), n * 4ull, hipMemcpyHostToDevice));
        ms = timeit([&] { hipLaunchKernelGGL(k_scatter_random, dim3(2048), dim3(256), 0, 0, didx, va, n); });
        printf("random 4-B scatter of %u elements over %u MB: %.3f ms\n", n, n / 262144, ms);
        for (u32 win : {1u << 18, 1u << 20}) {                    // 1 MB and 4 MB windows
            for (u32 i = 0; i < n; ++i) h[i] = i;
            for (u32 b = 0; b < n; b += win) for (u32 i = win - 1; i > 0; --i) { u32 j = (u32)(rnd() % (i + 1)); u32 t2 = h[b + i]; h[b + i] = h[b + j]; h[b + j] = t2; }
            CHECK(hipMemcpy(didx, h.data(), n * 4ull, hipMemcpyHostToDevice));
            for (u32 wpx : {32u, 64u}) {
                ms = timeit([&] { hipLaunchKernelGGL(k_scatter_windowed, dim3(8 * wpx), dim3(256), 0, 0, didx, va, n, win, wpx); });
                printf("windowed 4-B scatter, window %u KB, %u WG per XCD: %.3f ms\n", win / 256, wpx, ms);
            }
        }
    }
    return 0;
}
