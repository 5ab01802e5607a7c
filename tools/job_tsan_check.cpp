// job_tsan_check.cpp — the job driver's scheduler (csrc/host/job.cpp) under ThreadSanitizer with a stand-in executor: no GPU, no HIP.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I include tools/job_tsan_check.cpp -o /tmp/job_tsan_check -lpthread && /tmp/job_tsan_check
// (run by tests/test_job_driver.py).  The default executor's entry points are stubbed: this program only ever uses bscgpu_job_create_ex.
#include "../libbsc_amd/csrc/host/job.cpp"
extern "C" int bscgpu_coder_pool_expect(long long, int) { return 0; }      // (block.cpp is not linked here: the stand-in executor has no coder pool)
#include <chrono>
#include <cstdio>
#include <random>

extern "C" {
int bscgpu_device_count(void) { return 0; }
int bscgpu_create(bscgpu_ctx**, int, int64_t) { return LIBBSC_GPU_NOT_SUPPORTED; }
void bscgpu_destroy(bscgpu_ctx*) {}
int bscgpu_pipe_create(bscgpu_ctx*, int, bscgpu_pipe**) { return LIBBSC_GPU_NOT_SUPPORTED; }
void bscgpu_pipe_destroy(bscgpu_pipe*) {}
int bscgpu_pipe_submit_host(bscgpu_pipe*, const uint8_t*, uint8_t*, int, int, int, int, int, int) { return LIBBSC_GPU_NOT_SUPPORTED; }
int bscgpu_pipe_wait(bscgpu_pipe*, int) { return LIBBSC_GPU_NOT_SUPPORTED; }
int bscgpu_pipe_peek(bscgpu_pipe*, int, int*) { return 0; }
}

namespace {
struct FakePipe { std::mutex mu; std::deque<std::pair<int, std::pair<const uint8_t*, std::pair<uint8_t*, int>>>> q; int next = 0; int device = 0; };
std::atomic<int> g_low{0}, g_blocks{0};
int f_ctx_create(void*, void** ctx, int device, int64_t) { *ctx = new int(device); return 0; }
void f_ctx_destroy(void*, void* ctx) { delete (int*)ctx; }
int f_pipe_create(void*, void* ctx, int, void** pipe) { FakePipe* p = new FakePipe; p->device = *(int*)ctx; *pipe = p; return 0; }
void f_pipe_destroy(void*, void* pipe) { delete (FakePipe*)pipe; }
int f_submit(void*, void* pipe, const uint8_t* in, uint8_t* out, int n, int, int, int sorter, int, int features)
{
    if (sorter == 7) return LIBBSC_NOT_COMPRESSIBLE;                                     // (the stand-in's error: any code but BAD_PARAMETER, which wait() uses for 'no such block')
    FakePipe* p = (FakePipe*)pipe;
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 100 * p->device));       // "GPU stage"
    if (features & BSCGPU_FEATURE_LOW_LATENCY) ++g_low;
    std::lock_guard<std::mutex> lk(p->mu);
    p->q.push_back({p->next, {in, {out, n}}});
    return p->next++;
}
int f_wait(void*, void* pipe, int ticket)
{
    FakePipe* p = (FakePipe*)pipe;
    const uint8_t* in; uint8_t* out; int n;
    { std::lock_guard<std::mutex> lk(p->mu); if (p->q.empty() || p->q.front().first != ticket) return -7; in = p->q.front().second.first; out = p->q.front().second.second.first; n = p->q.front().second.second.second; p->q.pop_front(); }
    std::this_thread::sleep_for(std::chrono::microseconds(300));                        // "host coding"
    memcpy(out, in, (size_t)n);
    ++g_blocks;
    return n;
}
}  // namespace

int main()
{
    const bscgpu_job_backend be{nullptr, f_ctx_create, f_ctx_destroy, f_pipe_create, f_pipe_destroy, f_submit, f_wait};
    std::mt19937 rng(7);
    int failures = 0;
    for (int round = 0; round < 6; ++round) {
        const int ndev = 1 + round % 3, cpd = 1 + round % 4, depth = 1 + round % 3, total = 40 + 17 * round;
        std::vector<int> devs; for (int d = 0; d < ndev; ++d) devs.push_back(d);
        bscgpu_job* job = nullptr;
        if (bscgpu_job_create_ex(&job, devs.data(), ndev, cpd, depth, 1 << 16, &be) != 0) { ++failures; continue; }
        std::vector<std::vector<uint8_t>> in((size_t)total), out((size_t)total);
        for (int b = 0; b < total; ++b) { in[(size_t)b].assign((size_t)(1 + rng() % 3000), (uint8_t)b); out[(size_t)b].assign(in[(size_t)b].size() + 28, 0); }
        g_low = 0;
        if (round % 2 == 0) bscgpu_job_expect(job, total);
        // one thread adds (with pauses, so that the job runs dry and bursts restart), another collects in order
        std::thread adder([&] {
            for (int b = 0; b < total; ++b) {
                const int id = bscgpu_job_add(job, in[(size_t)b].data(), out[(size_t)b].data(), (int)in[(size_t)b].size(), 0, 0, (b == 13) ? 7 : 1, 1, 3);
                if (id != b) ++failures;
                if (b % 11 == 10) std::this_thread::sleep_for(std::chrono::milliseconds(3));
            }
        });
        for (int b = 0; b < total; ++b) {
            int r;
            while ((r = bscgpu_job_wait(job, b)) == LIBBSC_BAD_PARAMETER) std::this_thread::yield();      // not added yet
            if (b == 13) { if (r != LIBBSC_NOT_COMPRESSIBLE) ++failures; continue; }
            if (r != (int)in[(size_t)b].size() || memcmp(out[(size_t)b].data(), in[(size_t)b].data(), in[(size_t)b].size()) != 0) ++failures;
            int dev = -1;
            const int w = bscgpu_job_block_worker(job, b, &dev);
            if (w < 0 || dev != devs[(size_t)(w % ndev)]) ++failures;
            if (round % 2 == 0 && total - b <= (w / ndev) * ndev) ++failures;           // the tail rule
        }
        adder.join();
        bscgpu_job_destroy(job);
        if (round % 2 == 0 && g_low.load() < 1) ++failures;
    }
    printf("job scheduler under ThreadSanitizer: %d failure(s), %d blocks\n", failures, g_blocks.load());
    return failures ? 1 : 0;
}
