// Microbenchmark (not part of the product), round 6: what RUN LENGTH is worth to the write pattern of an 8-bit digit pass.
// A digit pass over n (u64 key, u32 value) records writes, per tile of T records, 256 runs of L = T / 256 records each (uniform
// digits, as the BWT's 5-bit character codes give them): run d of tile t goes to stream d at offset t * L.  This program issues
// exactly that pattern — coalesced reads of a tile, writes as runs of L records — with every offset known in advance and NO ranking,
// LDS or protocol: the ceiling of the pattern itself, as a function of L, on this box.
//   soa  : keys and values as two streams (runs of 8 L and 4 L bytes): the product's layout;
//   aos  : 12-byte records, one stream (runs of 12 L bytes): the review's variant (ii);
//   order: tiles handed out round-robin over the 256 workgroups (tile = round * 256 + workgroup: what ticket order amounts to when
//          all workgroups progress at the same rate) or as consecutive blocks per workgroup.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_runlen.hip -o tools/bin/ubench_runlen
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct __attribute__((packed, aligned(4))) Rec12 { u32 a, b, c; };

// streams start at irregular (line-aligned) offsets, as digit buckets do: a regular stride would put every stream on the same channel
__device__ __forceinline__ u64 stream_base(u32 b, u32 per_bucket) { return (u64)b * per_bucket + ((((b) * 2654435761u) >> 21) & 2047u) * 32u; }

template <int ITEMS, bool AOS, int ORDER, int WG>
__global__ __launch_bounds__(WG) void k_runs(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                             const Rec12* __restrict__ rin, Rec12* __restrict__ rout, u32 n, u32 tiles_per_wg, u32 nwg)
{
    constexpr u32 T = WG * ITEMS, L = T / 256;
    const u32 t = threadIdx.x;
    const u32 per_bucket = n / 256;
    for (u32 tt = 0; tt < tiles_per_wg; ++tt) {
        const u32 tile = ORDER != 1 ? tt * nwg + blockIdx.x : blockIdx.x * tiles_per_wg + tt;
        const u64 tb = (u64)tile * T;
        const u32 run0 = tile * L;
        if (!AOS) {
            u64 k[ITEMS]; u32 v[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) k[j] = __builtin_nontemporal_load(&kin[tb + j * WG + t]);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) v[j] = __builtin_nontemporal_load(&vin[tb + j * WG + t]);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) { const u32 q = j * WG + t, b = q / L, r = q % L; if (ORDER == 2) kout[tb + q] = k[j]; else kout[stream_base(b, per_bucket) + run0 + r] = k[j]; }
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) { const u32 q = j * WG + t, b = q / L, r = q % L; if (ORDER == 2) vout[tb + q] = v[j]; else vout[stream_base(b, per_bucket) + run0 + r] = v[j]; }
        } else {
            Rec12 x[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) x[j] = rin[tb + j * WG + t];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) { const u32 q = j * WG + t, b = q / L, r = q % L; rout[stream_base(b, per_bucket) + run0 + r] = x[j]; }
        }
    }
}

__global__ __launch_bounds__(1024) void k_copy(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout, u32 n)
{
    for (u64 i = (u64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (u64)gridDim.x * 1024) { kout[i] = __builtin_nontemporal_load(&kin[i]); vout[i] = __builtin_nontemporal_load(&vin[i]); }
}

// a better copy: 4 x 16 bytes per thread and trip, all loads first (1.5 GiB in all: the same 24 B per record)
typedef u32 v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const v4u* __restrict__ in, v4u* __restrict__ out, u64 n16)
{
    const u64 stride = (u64)gridDim.x * 256 * 4;
    for (u64 i = (u64)blockIdx.x * 256 * 4 + threadIdx.x; i + 768 < n16; i += stride) {
        const v4u a = __builtin_nontemporal_load(&in[i]), b = __builtin_nontemporal_load(&in[i + 256]), c = __builtin_nontemporal_load(&in[i + 512]), d = __builtin_nontemporal_load(&in[i + 768]);
        __builtin_nontemporal_store(a, &out[i]); __builtin_nontemporal_store(b, &out[i + 256]); __builtin_nontemporal_store(c, &out[i + 512]); __builtin_nontemporal_store(d, &out[i + 768]);
    }
}

template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 7; ++r) { CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t; }
    return best;
}

template <int ITEMS, bool AOS, int ORDER, int WG>
static void one(const char* what, const u64* ka, const u32* va, u64* kb, u32* vb, const Rec12* ra, Rec12* rb, u32 n)
{
    constexpr u32 T = WG * ITEMS;
    const u32 nwg = 256 * (1024 / WG);
    const u32 tiles = n / T / nwg;
    float ms = timeit([&] { hipLaunchKernelGGL((k_runs<ITEMS, AOS, ORDER, WG>), dim3(nwg), dim3(WG), 0, 0, ka, va, kb, vb, ra, rb, n, tiles, nwg); });
    const double bytes = 24.0 * (double)tiles * nwg * T;
    printf("%-58s tile %6u  run %4u records = %5u B%s: %.3f ms -> %5.0f GB/s = %.3f of 8 TB/s\n", what, T, T / 256, AOS ? 12 * (T / 256) : 8 * (T / 256), AOS ? "" : " + half", ms, bytes / 1e6 / ms, bytes / 1e6 / ms / 8000.0);
}

int main() {
    const u32 n = 64u << 20;
    u64 *ka, *kb; u32 *va, *vb; Rec12 *ra, *rb;
    CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull + (4 << 20))); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull + (4 << 20)));
    CHECK(hipMalloc(&ra, n * 12ull)); CHECK(hipMalloc(&rb, n * 12ull + (4 << 20)));
    CHECK(hipMemset(ka, 0x5a, n * 8ull)); CHECK(hipMemset(va, 1, n * 4ull)); CHECK(hipMemset(ra, 3, n * 12ull));
    float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(1024), 0, 0, ka, va, kb, vb, n); });
    printf("streaming copy of the same 24 B per record: %.3f ms -> %.0f GB/s = %.3f of 8 TB/s\n", ms, 24.0 * n / 1e6 / ms, 24.0 * n / 1e6 / ms / 8000.0);
    for (u32 blocks : {1024u, 2048u, 4096u, 8192u}) {
        const u64 n16 = (u64)n * 12 / 16;          // the 12-byte record array: 768 MiB in, 768 MiB out
        ms = timeit([&] { hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, 0, (const v4u*)ra, (v4u*)rb, n16); });
        printf("streaming copy, 16-byte accesses, 4 in flight per thread, %u workgroups of 256: %.3f ms -> %.0f GB/s = %.3f of 8 TB/s\n", blocks, ms, 2.0 * n16 * 16 / 1e6 / ms, 2.0 * n16 * 16 / 1e6 / ms / 8000.0);
    }
    printf("-- two streams (keys, values), tiles round-robin over 256 workgroups of 1024 threads\n");
    one<4,  false, 0, 1024>("soa", ka, va, kb, vb, ra, rb, n);
    one<8,  false, 0, 1024>("soa (the product's shape: 7680 in the single-read pass)", ka, va, kb, vb, ra, rb, n);
    one<16, false, 0, 1024>("soa", ka, va, kb, vb, ra, rb, n);
    one<32, false, 0, 1024>("soa", ka, va, kb, vb, ra, rb, n);
    printf("-- one stream of 12-byte records\n");
    one<4,  true, 0, 1024>("aos", ka, va, kb, vb, ra, rb, n);
    one<6,  true, 0, 1024>("aos (6144: what two staging buffers of 12-byte records allow)", ka, va, kb, vb, ra, rb, n);
    one<8,  true, 0, 1024>("aos", ka, va, kb, vb, ra, rb, n);
    one<16, true, 0, 1024>("aos", ka, va, kb, vb, ra, rb, n);
    printf("-- consecutive tiles per workgroup instead of round-robin\n");
    one<8,  false, 1, 1024>("soa, owned blocks of tiles", ka, va, kb, vb, ra, rb, n);
    one<16, false, 1, 1024>("soa, owned blocks of tiles", ka, va, kb, vb, ra, rb, n);
    printf("-- the same kernel, every tile written contiguously (no streams: what its loads and stores reach as a plain copy)\n");
    one<8,  false, 2, 1024>("soa, one stream", ka, va, kb, vb, ra, rb, n);
    one<16, false, 2, 1024>("soa, one stream", ka, va, kb, vb, ra, rb, n);
    printf("-- two workgroups of 512 threads per CU (same tile sizes: 8 / 16 / 32 records per thread)\n");
    one<8,  false, 0, 512>("soa, 512-thread workgroups", ka, va, kb, vb, ra, rb, n);
    one<16, false, 0, 512>("soa, 512-thread workgroups", ka, va, kb, vb, ra, rb, n);
    one<32, false, 0, 512>("soa, 512-thread workgroups", ka, va, kb, vb, ra, rb, n);
    return 0;
}
