// startup_probe.cpp — where a short job's start-up goes (not part of the product): HIP initialisation, the arenas, pinned landing zones.
//   hipcc -O2 tools/startup_probe.cpp -o tools/bin/startup_probe -lpthread && tools/bin/startup_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
int main()
{
    auto t = clk::now();
    int n = 0; hipGetDeviceCount(&n); hipSetDevice(0); hipFree(nullptr);
    printf("hip init                         %8.1f ms (%d device(s))\n", ms(t), n);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void* a = nullptr;
    t = clk::now(); hipMalloc(&a, (size_t)3900 << 20); printf("hipMalloc 3.9 GB                 %8.1f ms\n", ms(t));
    t = clk::now(); hipMemsetAsync(a, 0, (size_t)3900 << 20, s); hipStreamSynchronize(s); printf("memset 3.9 GB                    %8.1f ms\n", ms(t));
    void* b = nullptr;
    t = clk::now(); hipMalloc(&b, (size_t)15000 << 20); printf("hipMalloc 15 GB                  %8.1f ms\n", ms(t));
    const size_t P = (size_t)366 << 20;
    for (int rep = 0; rep < 2; ++rep) {
        void* h = nullptr;
        t = clk::now(); hipHostMalloc(&h, P, hipHostMallocDefault); printf("hipHostMalloc 366 MB             %8.1f ms\n", ms(t));
        t = clk::now(); hipMemcpyAsync(h, b, P, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("  D2H 366 MB into it             %8.1f ms (%.1f GB/s)\n", ms(t), P / 1e6 / ms(t));
        t = clk::now(); hipMemcpyAsync(h, b, P, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("  D2H again                      %8.1f ms (%.1f GB/s)\n", ms(t), P / 1e6 / ms(t));
        t = clk::now(); hipHostFree(h); printf("  hipHostFree                    %8.1f ms\n", ms(t));
    }
    for (int threads : {1, 8}) {
        t = clk::now();
        void* m = mmap(nullptr, P, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(m, P, MADV_HUGEPAGE);
        std::vector<std::thread> th;
        for (int k = 0; k < threads; ++k) th.emplace_back([=] { char* p = (char*)m + P / threads * k; for (size_t i = 0; i < P / threads; i += 4096) p[i] = 0; });
        for (auto& x : th) x.join();
        const double t_touch = ms(t);
        auto t2 = clk::now();
        hipError_t e = hipHostRegister(m, P, hipHostRegisterDefault);
        printf("mmap + touch by %d thread(s) %6.1f ms, hipHostRegister %6.1f ms (%s)\n", threads, t_touch, ms(t2), hipGetErrorString(e));
        t2 = clk::now(); hipMemcpyAsync(m, b, P, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("  D2H 366 MB into it             %8.1f ms (%.1f GB/s)\n", ms(t2), P / 1e6 / ms(t2));
        t2 = clk::now(); hipHostUnregister(m); munmap(m, P); printf("  unregister + munmap            %8.1f ms\n", ms(t2));
    }
    {   // pageable destination, for comparison
        void* m = malloc(P); memset(m, 1, P);
        t = clk::now(); hipMemcpyAsync(m, b, P, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("D2H 366 MB into pageable memory  %8.1f ms (%.1f GB/s)\n", ms(t), P / 1e6 / ms(t));
        free(m);
    }
    // several pinned allocations at once (six contexts start together)
    t = clk::now();
    { std::vector<std::thread> th; std::vector<void*> hs(6, nullptr);
      for (int k = 0; k < 6; ++k) th.emplace_back([&hs, k, P] { hipSetDevice(0); hipHostMalloc(&hs[k], P, hipHostMallocDefault); });
      for (auto& x : th) x.join();
      printf("6 x hipHostMalloc 366 MB at once %8.1f ms\n", ms(t));
      for (void* h : hs) hipHostFree(h); }
    return 0;
}
