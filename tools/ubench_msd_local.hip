// Microbenchmark (not part of the product), round 6, review item 4: what the LOCAL FINISH of a most-significant-digit first sort costs at best.
// The plan under test: sort the BWT's 64 Mi (u64 key, u32 suffix) records on the TOP 32 key bits with four single-read digit passes (known: 4 x
// 0.383 ms + 0.17 ms of histogram), then finish every segment on its low 28 bits inside one workgroup's LDS.  This program measures that finish in
// the most favourable form there is: every 8192-record tile is ONE segment (no segment boundaries to respect, no segment longer than a tile, no
// global fallback) — a LOWER bound of what the real thing costs.  A tile is read once (12 B per record), its (low 32 key bits, value) pairs go
// through PASSES stable 8-bit LDS digit passes with the product's own ranking code (rs_rank_wave: ballot match, per-wave counters), and it is
// written once (12 B per record).
//   hipcc -O3 --offload-arch=gfx950 -I libbsc_amd/csrc/device -I include tools/ubench_msd_local.hip -o tools/bin/ubench_msd_local
#include "../libbsc_amd/csrc/device/radix_dev.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int LWG = 1024, LWAVES = 16, ITEMS = 8, TILE = LWG * ITEMS;       // 8192 records per tile: 2 x 64 KB of (key32, value) + 16 KB of counters

template <int PASSES>
__global__ __launch_bounds__(LWG) void k_local(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout, u32 ntiles)
{
    extern __shared__ u32 lds[];
    u32* bufk[2] = {lds, lds + 2 * TILE};                 // [TILE] low key words, ping / pong
    u32* bufv[2] = {lds + TILE, lds + 3 * TILE};          // [TILE] values
    u32* wh = lds + 4 * TILE;                             // [LWAVES][256] per-wave digit counters
    u32* scr = wh + LWAVES * 256;                         // [LWAVES] scan scratch
    const u32 t = threadIdx.x, w = t >> 6, l = t & 63u;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u64 base = (u64)tile * TILE;
        // wave w owns records [w * 512, (w + 1) * 512) of the tile, item-major inside (item i of all lanes before item i + 1)
        u32 hi[ITEMS];                                     // the top key words stay in registers (one segment per tile: they do not move)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u32 r = w * 512u + (u32)i * 64u + l;
            const u64 k = __builtin_nontemporal_load(&kin[base + r]);
            hi[i] = (u32)(k >> 32);
            bufk[0][r] = (u32)k; bufv[0][r] = __builtin_nontemporal_load(&vin[base + r]);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const u32* sk = bufk[p & 1]; const u32* sv = bufv[p & 1];
            u32* dk = bufk[(p + 1) & 1]; u32* dv = bufv[(p + 1) & 1];
            u64 k[ITEMS]; u32 v[ITEMS], rk[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { const u32 r = w * 512u + (u32)i * 64u + l; k[i] = sk[r]; v[i] = sv[r]; }
            lds_vu32* mywh = (lds_vu32*)(wh + w * 256u);
            for (u32 d = l; d < 256u; d += 64u) mywh[d] = 0u;
            __builtin_amdgcn_wave_barrier();
            rs_rank_wave<ITEMS>(k, p * 8, 255u, mywh, rk);
            __syncthreads();
            // digit d (thread d < 256): its count over the waves, exclusive over the waves, then exclusive over the digits
            u32 tot = 0;
            if (t < 256u) {
#pragma unroll
                for (int ww = 0; ww < LWAVES; ++ww) { const u32 c = wh[ww * 256 + t]; wh[ww * 256 + t] = tot; tot += c; }
            }
            u32 total;
            const u32 dbase = rs_digit_excl_sum<LWAVES>(t < 256u ? tot : 0u, scr, &total);
            if (t < 256u) {
#pragma unroll
                for (int ww = 0; ww < LWAVES; ++ww) wh[ww * 256 + t] += dbase;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const u32 d = (u32)(k[i] >> (p * 8)) & 255u;
                const u32 dst = wh[w * 256u + d] + rk[i];
                dk[dst] = (u32)k[i]; dv[dst] = v[i];
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u32 r = w * 512u + (u32)i * 64u + l;
            kout[base + r] = ((u64)hi[i] << 32) | bufk[PASSES & 1][r];
            vout[base + r] = bufv[PASSES & 1][r];
        }
        __syncthreads();
    }
}

template <int PASSES>
static void run(const u64* ka, const u32* va, u64* kb, u32* vb, u32 n, const std::vector<u64>& hk, const std::vector<u32>& hv)
{
    const u32 ntiles = n / TILE;
    const size_t lds = (size_t)(4 * TILE + LWAVES * 256 + 64) * 4;
    CHECK(hipFuncSetAttribute((const void*)k_local<PASSES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_local<PASSES>, dim3(256), dim3(LWG), lds, 0, ka, va, kb, vb, ntiles); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_local<PASSES>, dim3(256), dim3(LWG), lds, 0, ka, va, kb, vb, ntiles); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    // check the first and the last tile against a stable sort on the low PASSES * 8 bits
    int bad = 0;
    for (u32 tile : {0u, ntiles - 1}) {
        std::vector<u64> gk(TILE); std::vector<u32> gv(TILE);
        CHECK(hipMemcpy(gk.data(), kb + (size_t)tile * TILE, TILE * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gv.data(), vb + (size_t)tile * TILE, TILE * 4, hipMemcpyDeviceToHost));
        // the kernel's record order inside a tile is (wave chunk, item-major): position r of the tile is input record r, so a stable sort of the tile's input is the reference
        std::vector<u32> idx(TILE); for (u32 i = 0; i < (u32)TILE; ++i) idx[i] = i;
        const u64 mask = PASSES >= 8 ? ~0ull : ((1ull << (PASSES * 8)) - 1);
        const size_t o = (size_t)tile * TILE;
        std::stable_sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return (hk[o + a] & mask) < (hk[o + b] & mask); });
        for (u32 i = 0; i < (u32)TILE; ++i) if ((u32)gk[i] != (u32)hk[o + idx[i]] || gv[i] != hv[o + idx[i]]) { ++bad; break; }
    }
    printf("local finish, %d LDS digit passes on 8192-record tiles (one segment each): %.3f ms for %u records = %.1f ps per record, %s\n", PASSES, best, ntiles * TILE,
           best * 1e9 / ((double)ntiles * TILE), bad ? "WRONG ORDER" : "order checked");
}

int main()
{
    const u32 n = 64u << 20;
    std::vector<u64> hk(n); std::vector<u32> hv(n);
    u64 x = 88172645463325252ull;
    for (u32 i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hk[i] = x & 0x0fffffffffffffffull; hv[i] = i; }
    u64 *ka, *kb; u32 *va, *vb;
    CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull)); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull));
    CHECK(hipMemcpy(ka, hk.data(), n * 8ull, hipMemcpyHostToDevice)); CHECK(hipMemcpy(va, hv.data(), n * 4ull, hipMemcpyHostToDevice));
    run<0>(ka, va, kb, vb, n, hk, hv);          // the tile through LDS and back: the floor of the shape
    run<1>(ka, va, kb, vb, n, hk, hv);
    run<3>(ka, va, kb, vb, n, hk, hv);
    run<4>(ka, va, kb, vb, n, hk, hv);          // 28 low key bits = four 8-bit passes (the plan: top 32 bits by global passes)
    printf("for comparison: one global single-read digit pass of the product over the same 64 Mi records takes 0.383 ms; the four passes this finish would replace: 1.53 ms\n");
    return 0;
}
