// Microbenchmark (not part of the product): store patterns of a (u64 key, u32 value) digit pass at the 1024 x 8 shape —
// 256 workgroups of 1024 threads, 8192-record tiles, 256 output streams, 32 records per stream and tile.
//   full     : every (tile, stream) run starts on a 32-record boundary and leaves in one instruction (2 full key lines + 1 full
//              value line): what a scatter emits when old pending records are merged with the new ones before the write-out;
//   split    : the same lines, but the first pk keys / pv values of every run leave in a separate instruction from 16- / 32-lane
//              groups (flush of old pending records) and the rest from the main sweep: what rs_scatter_wc emits;
//   misalign : runs shifted by 5 records (every run shares its first and last line with the neighbouring tiles' runs): what the
//              plain rs_scatter emits on uniform digits.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_wc.hip -o tools/bin/ubench_wc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ORDER 0: workgroup b owns `tiles` consecutive tiles (what rs_scatter does); 1: tiles interleaved over the workgroups so that at
// any time an XCD (workgroups b % 8) works on 32 consecutive tiles — neighbouring runs of a stream are written at about the same
// time by CUs that share an L2
template <int MODE, int ORDER = 0>
__global__ __launch_bounds__(1024) void k_wc(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                             u32 n, u32 tiles, u32 pk, u32 pv)
{
    const u32 t = threadIdx.x;
    const u32 per_bucket = n / 256;             // streams start at irregular (32-record aligned) offsets, as digit buckets do:
    const u32 shift = (MODE == 2) ? 5u : 0u;    // a regular 2 MB stride would put every stream on the same memory channel
#define STREAM(b) ((u64)(b) * per_bucket + ((((b) * 2654435761u) >> 21) & 2047u) * 32u)
    for (u32 tt = 0; tt < tiles; ++tt) {
        const u32 tile = ORDER == 0 ? blockIdx.x * tiles + tt : tt * 256 + (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);
        const u64 tb = (u64)tile * 8192;
        u64 k[8]; u32 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = __builtin_nontemporal_load(&kin[tb + j * 1024 + t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(&vin[tb + j * 1024 + t]);
        const u32 run0 = tile * 32 + shift;                              // this tile's offset inside every stream
        if (MODE == 1) {
            // flush-shaped instructions: keys 16 lanes per stream (4 steps), values 32 lanes per stream (8 steps)
#pragma unroll
            for (int s = 0; s < 4; ++s) { const u32 b = s * 64 + (t >> 4), r = t & 15; if (r < pk) kout[STREAM(b) + run0 + r] = k[s]; }
#pragma unroll
            for (int s = 0; s < 8; ++s) { const u32 b = s * 32 + (t >> 5), r = t & 31; if (r < pv) vout[STREAM(b) + run0 + r] = v[s]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 q = j * 1024 + t, b = q >> 5, r = q & 31;
            const u64 o = STREAM(b) + run0 + r;
            if (MODE != 1 || r >= pk) kout[o] = k[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 q = j * 1024 + t, b = q >> 5, r = q & 31;
            const u64 o = STREAM(b) + run0 + r;
            if (MODE != 1 || r >= pv) vout[o] = v[j];
        }
    }
}

template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t; }
    return best;
}

int main() {
    const u32 n = 64u << 20, tiles = n / 256 / 8192;
    u64 *ka, *kb; u32 *va, *vb;
    CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull + (1 << 20))); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull + (1 << 20)));
    CHECK(hipMemset(ka, 0x5a, n * 8ull)); CHECK(hipMemset(va, 1, n * 4ull));
    float ms = timeit([&] { hipLaunchKernelGGL(k_wc<0>, dim3(256), dim3(1024), 0, 0, ka, va, kb, vb, n, tiles, 0u, 0u); });
    printf("1024x8 pairs, full aligned lines: %.3f ms -> %.0f GB/s\n", ms, 24.0 * n / 1e6 / ms);
    for (u32 pk : {7u, 12u}) for (u32 pv : {15u, 24u}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_wc<1>, dim3(256), dim3(1024), 0, 0, ka, va, kb, vb, n, tiles, pk, pv); });
        printf("1024x8 pairs, split (first %u keys / %u values of a run in flush-shaped instructions): %.3f ms -> %.0f GB/s\n", pk, pv, ms, 24.0 * n / 1e6 / ms);
    }
    ms = timeit([&] { hipLaunchKernelGGL(k_wc<2>, dim3(256), dim3(1024), 0, 0, ka, va, kb, vb, n, tiles, 0u, 0u); });
    printf("1024x8 pairs, runs misaligned by 5 records: %.3f ms -> %.0f GB/s\n", ms, 24.0 * n / 1e6 / ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_wc<0, 1>), dim3(256), dim3(1024), 0, 0, ka, va, kb, vb, n, tiles, 0u, 0u); });
    printf("tiles interleaved per XCD, full aligned lines: %.3f ms -> %.0f GB/s\n", ms, 24.0 * n / 1e6 / ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_wc<2, 1>), dim3(256), dim3(1024), 0, 0, ka, va, kb, vb, n, tiles, 0u, 0u); });
    printf("tiles interleaved per XCD, runs misaligned by 5 records: %.3f ms -> %.0f GB/s\n", ms, 24.0 * n / 1e6 / ms);
    return 0;
}
