// Microbenchmark (not part of the product): what a decoupled look-back ("onesweep") digit pass would pay on MI355X.
// 256 persistent workgroups of 1024 threads (one per CU) take 8192-record tiles of (u64 key, u32 value) in ticket order.  Per tile:
// stream the tile in, build its 256-bin digit histogram (LDS atomics), publish it as an AGGREGATE row (agent-scope stores), look
// back over the predecessors' rows (agent-scope loads, four rows in flight per digit thread) until a row carrying an INCLUSIVE
// prefix is found, publish the own inclusive row, stream the tile out (same index: the look-back result only feeds a dependency).
// Compared with the same kernel without the publish / look-back steps: the difference is what replaces the rs_hist kernel
// (0.115 ms per 64 Mi-record pass) in a single-read design.  Spins are bounded: a stuck chain gives up and reports it.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_lookback.hip -o tools/bin/ubench_lookback
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int WG = 1024, ITEMS = 8, TILE = WG * ITEMS;
constexpr u32 FLAG_AGG = 1u << 30, FLAG_INC = 2u << 30, VAL_MASK = (1u << 30) - 1;

template <bool LOOKBACK>
__global__ __launch_bounds__(WG) void k_pass(const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout, u32* __restrict__ vout,
                                             u32 ntiles, u32* ticket, u32* status /*[ntiles][256]*/, u32* stats /* [0] steps, [1] max steps, [2] gave up, [3] polls */)
{
    extern __shared__ unsigned char pad_lds[];          // residency: one workgroup per CU
    __shared__ u32 hist[256];
    __shared__ u32 stile;
    const u32 t = threadIdx.x;
    if (pad_lds[t] == 77 && ntiles == 0xffffffffu) stats[4] = 1;
    u32 acc_steps = 0, acc_max = 0, acc_polls = 0, acc_giveup = 0;      // per thread, folded once at the end
    for (;;) {
        if (t == 0) stile = atomicAdd(ticket, 1u);
        if (t < 256) hist[t] = 0;
        __syncthreads();
        const u32 tile = stile;
        if (tile >= ntiles) break;
        const u64 tb = (u64)tile * TILE;
        u64 k[ITEMS]; u32 v[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) k[j] = __builtin_nontemporal_load(&kin[tb + j * WG + t]);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) v[j] = __builtin_nontemporal_load(&vin[tb + j * WG + t]);
        u32 dep = 0;
        if (LOOKBACK) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) atomicAdd(&hist[(u32)(k[j] >> 13) & 255u], 1u);
            __syncthreads();
            if (t < 256) {
                const u32 cnt = hist[t];
                u32* row = status + (size_t)tile * 256 + t;
                __hip_atomic_store(row, FLAG_AGG | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u32 excl = 0, steps = 0, polls = 0;
                bool gave_up = false;
                int p = (int)tile - 1;
                while (p >= 0 && !gave_up) {
                    // up to four predecessor rows in flight
                    u32 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = (p - q >= 0) ? __hip_atomic_load(status + (size_t)(p - q) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLAG_INC;
                    bool done = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (done) break;
                        u32 val = x[q];
                        u32 spins = 0;
                        while ((val >> 30) == 0u) {                                // not published yet: poll
                            if (++spins > (1u << 16)) { gave_up = true; break; }
                            __builtin_amdgcn_s_sleep(2);
                            val = __hip_atomic_load(status + (size_t)(p - q) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ++polls;
                        }
                        if (gave_up) break;
                        if (p - q >= 0) { excl += val & VAL_MASK; ++steps; }
                        if ((val >> 30) == 2u) done = true;
                    }
                    if (done) break;
                    p -= 4;
                }
                __hip_atomic_store(row, FLAG_INC | ((excl + cnt) & VAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hist[t] = excl;
                acc_steps += steps; acc_max = steps > acc_max ? steps : acc_max; acc_polls += polls; acc_giveup += gave_up ? 1u : 0u;
            }
            __syncthreads();
            dep = hist[t & 255u] & 0u;                                              // the write-out waits for the prefixes
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) kout[tb + j * WG + t + dep] = k[j];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) vout[tb + j * WG + t + dep] = v[j];
        __syncthreads();
    }
    if (LOOKBACK && t < 256) {
        for (int d = 32; d >= 1; d >>= 1) {
            acc_steps += __shfl_down(acc_steps, d, 64); acc_polls += __shfl_down(acc_polls, d, 64); acc_giveup += __shfl_down(acc_giveup, d, 64);
            const u32 o = __shfl_down(acc_max, d, 64); acc_max = o > acc_max ? o : acc_max;
        }
        if ((t & 63) == 0) { atomicAdd(&stats[0], acc_steps); atomicMax(&stats[1], acc_max); atomicAdd(&stats[3], acc_polls); atomicAdd(&stats[2], acc_giveup); }
    }
}

int main() {
    const u32 n = 64u << 20, ntiles = n / TILE;
    u64 *ka, *kb; u32 *va, *vb, *ticket, *status, *stats;
    CHECK(hipMalloc(&ka, n * 8ull)); CHECK(hipMalloc(&kb, n * 8ull)); CHECK(hipMalloc(&va, n * 4ull)); CHECK(hipMalloc(&vb, n * 4ull));
    CHECK(hipMalloc(&ticket, 4)); CHECK(hipMalloc(&status, (size_t)ntiles * 256 * 4)); CHECK(hipMalloc(&stats, 64));
    {   // pseudo-random keys (digits near uniform)
        u64* h = (u64*)malloc(n * 8ull); u64 s = 88172645463325252ull;
        for (u32 i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s; }
        CHECK(hipMemcpy(ka, h, n * 8ull, hipMemcpyHostToDevice)); free(h);
    }
    CHECK(hipMemset(va, 1, n * 4ull));
    const size_t lds = 96 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void*)k_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f; u32 hs[8] = {0};
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemset(ticket, 0, 4)); CHECK(hipMemset(status, 0, (size_t)ntiles * 256 * 4)); CHECK(hipMemset(stats, 0, 64));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_pass<false>, dim3(256), dim3(WG), lds, 0, ka, va, kb, vb, ntiles, ticket, status, stats);
            else           hipLaunchKernelGGL(k_pass<true>, dim3(256), dim3(WG), lds, 0, ka, va, kb, vb, ntiles, ticket, status, stats);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CHECK(hipMemcpy(hs, stats, 32, hipMemcpyDeviceToHost)); }
        }
        if (mode == 0) printf("ticket-ordered tile copy, 256 x 1024 threads, no look-back: %.3f ms -> %.0f GB/s\n", best, 24.0 * n / 1e6 / best);
        else printf("the same with histogram + publish + look-back + publish:       %.3f ms -> %.0f GB/s; rows summed per (tile, digit): mean %.1f, max %u; "
                    "polls of unpublished rows %u; chains that gave up %u\n", best, 24.0 * n / 1e6 / best, (double)hs[0] / ((double)ntiles * 256), hs[1], hs[3], hs[2]);
    }
    return 0;
}
