// Microbenchmark (not part of the product), round 6: what SHAPE a streaming kernel needs to reach the chip's copy rate — access width per
// lane (4 / 8 / 16 bytes), accesses in flight per lane, workgroup size, workgroups per CU.  1.5 GiB moved per launch (768 MiB in, 768 MiB
// out), loads of a trip first, then its stores, grid-stride over trips.  The digit pass's own shape is 256 workgroups of 1024 threads with
// 8-byte and 4-byte accesses, eight of each in flight: tools/ubench_runlen.hip shows that this shape, not the 256-stream write pattern, is
// what stops at 0.60-0.63 of 8 TB/s.
// hipcc -O3 --offload-arch=gfx950 tools/ubench_copyshape.hip -o tools/bin/ubench_copyshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <class T, int ITEMS, int WG>
__global__ __launch_bounds__(WG) void k_copy(const T* __restrict__ in, T* __restrict__ out, u64 n)
{
    const u64 trip = (u64)WG * ITEMS;
    for (u64 base = (u64)blockIdx.x * trip; base + trip <= n; base += (u64)gridDim.x * trip) {
        T x[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) x[j] = __builtin_nontemporal_load(&in[base + (u64)j * WG + threadIdx.x]);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) __builtin_nontemporal_store(x[j], &out[base + (u64)j * WG + threadIdx.x]);
    }
}
// mixed widths at the digit pass's shape (1024 threads, one workgroup per CU, 8 u64 per thread and trip): 16-byte loads with 8-byte stores
// (LW = true: the pairs go through LDS to come out one record per lane, as the ranking needs them) and 8-byte loads with 16-byte stores (SW)
template <bool LW, bool SW>
__global__ __launch_bounds__(1024) void k_mixed(const u64* __restrict__ in, u64* __restrict__ out, u64 n)
{
    __shared__ __attribute__((aligned(16))) u64 sm[1024 * 8];
    constexpr int ITEMS = 8;
    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63u;
    const u64 trip = 1024ull * ITEMS;
    u64* ws = sm + w * 512;                                           // the wavefront's 512 records
    for (u64 base = (u64)blockIdx.x * trip; base + trip <= n; base += (u64)gridDim.x * trip) {
        const u64 wb = base + (u64)w * 512;
        u64 x[ITEMS];
        if (LW) {
#pragma unroll
            for (int j = 0; j < ITEMS / 2; ++j) { const v4u q = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(in + wb + j * 128 + 2 * lane)); *reinterpret_cast<v4u*>(ws + j * 128 + 2 * lane) = q; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) x[j] = ws[j * 64 + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) x[j] = __builtin_nontemporal_load(&in[wb + j * 64 + lane]);
        }
        if (SW) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) ws[j * 64 + lane] = x[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < ITEMS / 2; ++j) { const v4u q = *reinterpret_cast<const v4u*>(ws + j * 128 + 2 * lane); *reinterpret_cast<v4u*>(out + wb + j * 128 + 2 * lane) = q; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) out[wb + j * 64 + lane] = x[j];
        }
    }
}

template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 7; ++r) { CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t; }
    return best;
}
template <class T, int ITEMS, int WG>
static void one(const void* a, void* b, u64 bytes, u32 blocks)
{
    const u64 n = bytes / sizeof(T);
    float ms = timeit([&] { hipLaunchKernelGGL((k_copy<T, ITEMS, WG>), dim3(blocks), dim3(WG), 0, 0, (const T*)a, (T*)b, n); });
    printf("%2zu-byte accesses x %2d in flight, %5u workgroups of %4d (%4.1f per CU): %.3f ms -> %5.0f GB/s = %.3f of 8 TB/s\n", sizeof(T), ITEMS, blocks, WG, blocks / 256.0, ms, 2.0 * bytes / 1e6 / ms, 2.0 * bytes / 1e6 / ms / 8000.0);
}
int main() {
    const u64 bytes = 768ull << 20;
    void *a, *b;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(a, 0x5a, bytes));
    printf("-- one workgroup of 1024 threads per CU (the single-read digit pass's shape)\n");
    one<u32, 8, 1024>(a, b, bytes, 256); one<u64, 8, 1024>(a, b, bytes, 256); one<v4u, 4, 1024>(a, b, bytes, 256); one<v4u, 8, 1024>(a, b, bytes, 256);
    one<u64, 16, 1024>(a, b, bytes, 256);
    {
        const u64 n = bytes / 8;
        float ms = timeit([&] { hipLaunchKernelGGL((k_mixed<false, false>), dim3(256), dim3(1024), 0, 0, (const u64*)a, (u64*)b, n); });
        printf(" 8-byte loads,  8-byte stores (wave-striped, the digit pass's own accesses): %.3f ms -> %5.0f GB/s = %.3f of 8 TB/s\n", ms, 2.0 * bytes / 1e6 / ms, 2.0 * bytes / 1e6 / ms / 8000.0);
        ms = timeit([&] { hipLaunchKernelGGL((k_mixed<true, false>), dim3(256), dim3(1024), 0, 0, (const u64*)a, (u64*)b, n); });
        printf("16-byte loads (through LDS to one record per lane), 8-byte stores:           %.3f ms -> %5.0f GB/s = %.3f of 8 TB/s\n", ms, 2.0 * bytes / 1e6 / ms, 2.0 * bytes / 1e6 / ms / 8000.0);
        ms = timeit([&] { hipLaunchKernelGGL((k_mixed<false, true>), dim3(256), dim3(1024), 0, 0, (const u64*)a, (u64*)b, n); });
        printf(" 8-byte loads, 16-byte stores (through LDS):                                %.3f ms -> %5.0f GB/s = %.3f of 8 TB/s\n", ms, 2.0 * bytes / 1e6 / ms, 2.0 * bytes / 1e6 / ms / 8000.0);
        ms = timeit([&] { hipLaunchKernelGGL((k_mixed<true, true>), dim3(256), dim3(1024), 0, 0, (const u64*)a, (u64*)b, n); });
        printf("16-byte loads, 16-byte stores (both through LDS):                           %.3f ms -> %5.0f GB/s = %.3f of 8 TB/s\n", ms, 2.0 * bytes / 1e6 / ms, 2.0 * bytes / 1e6 / ms / 8000.0);
    }
    printf("-- two workgroups of 1024 / four of 512 / eight of 256 per CU\n");
    one<u64, 8, 1024>(a, b, bytes, 512); one<u64, 8, 512>(a, b, bytes, 1024); one<u64, 8, 256>(a, b, bytes, 2048);
    one<v4u, 4, 1024>(a, b, bytes, 512); one<v4u, 4, 512>(a, b, bytes, 1024); one<v4u, 4, 256>(a, b, bytes, 2048);
    printf("-- many small workgroups\n");
    one<u32, 8, 256>(a, b, bytes, 4096); one<u64, 8, 256>(a, b, bytes, 4096); one<v4u, 4, 256>(a, b, bytes, 4096); one<v4u, 2, 256>(a, b, bytes, 8192); one<u64, 4, 256>(a, b, bytes, 8192);
    return 0;
}
