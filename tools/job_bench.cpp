// job_bench.cpp — bench.py's workload through the C job driver alone: no Python, no torch, no file I/O.
//
// What the north star calls the product — "host code stays in C/C++ calling a thin C-ABI" — is include/libbsc.h + include/bscgpu.h;
// bench.py drives the same pipes from Python threads.  This program is the C caller: one bscgpu_job (contexts x depth blocks in flight
// on every GPU it is given, one queue, head and tail tapered by the job itself once the total is announced), K blocks of synth-text v1
// from a resident host buffer, results collected in order, wall time from the first add to the last wait.  It answers the round-4
// review's question whether job.cpp sustains what bench.py reports (profiles/r05/job_bench*.json); role: the reference CLI's block
// loop, bsc.cpp:182-199, minus the file.
//
//   job_bench [--steps K] [--warmup W] [--contexts C] [--depth D] [--gpus G] [--block BYTES] [--sorter S] [--coder E] [--seed N]
//             [--lzp H,M] [--dump FILE]
// Prints one JSON line.  The last block is checked against the reference outputs committed in tests/golden/golden_big.json for the
// BASELINE configurations it knows (size + md5, table below); --dump writes it out for any other check.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/resource.h>

#include "../include/libbsc.h"
#include "../include/bscgpu.h"

extern "C" int hipHostRegister(void* p, size_t bytes, unsigned flags);        // (--pin-input only; the HIP runtime the library is linked against)

// ---- MD5 (RFC 1321), for the golden check only ----------------------------------------------------------------------------------
namespace {
struct Md5 {
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const unsigned char* p)
    {
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        static uint32_t K[64]; static bool init = false;
        if (!init) { for (int i = 0; i < 64; ++i) { double v = __builtin_fabs(__builtin_sin((double)(i + 1))); K[i] = (uint32_t)(v * 4294967296.0); } init = true; }
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (b | ~d); g = (7 * i) & 15; }
            const uint32_t t = d; d = c; c = b; b = b + rol(a + f + K[i] + w[g], S[i]); a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    std::string hex(const unsigned char* data, size_t n)
    {
        size_t i = 0;
        for (; i + 64 <= n; i += 64) block(data + i);
        unsigned char tail[128]; size_t r = n - i;
        memcpy(tail, data + i, r); tail[r++] = 0x80;
        const size_t pad = (r <= 56) ? 56 - r : 120 - r;
        memset(tail + r, 0, pad); r += pad;
        const uint64_t bits = (uint64_t)n * 8;
        for (int k = 0; k < 8; ++k) tail[r++] = (unsigned char)(bits >> (8 * k));
        for (size_t q = 0; q < r; q += 64) block(tail + q);
        char out[33];
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) snprintf(out + 8 * k + 2 * j, 3, "%02x", (h[k] >> (8 * j)) & 0xffu);
        return std::string(out, 32);
    }
};

// tests/golden/golden_big.json (generated from the compiled reference by tests/golden/make_golden_big.py), the rows this program can meet
struct Golden { unsigned long long seed; long long n; int sorter, coder, size; const char* md5; };
const Golden kGolden[] = {
    {2, 64ll << 20, 1, 1, 15277890, "0ae79c8172e7e5df11b4b5a2e5c53b58"},
    {2, 64ll << 20, 1, 2, 15148620, "bea58a30c5fae2b503207644b6efe670"},
    {2, 64ll << 20, 1, 3, 15408618, "179dad28abbfe303a54b2ec364b99994"},
    {3, 128ll << 20, 5, 1, 31239874, "cbc1861a51692f7a6b3d57106f6726e8"},
    {3, 128ll << 20, 6, 1, 31235916, "64d571958b24d1e76c930232bb0dbbf5"},
    {3, 128ll << 20, 5, 3, 31469034, "f970435cf1d8ad506cdb70688ee65a6b"},
    {3, 128ll << 20, 6, 3, 31464832, "83666025817384e3d013e774ac5bb9dc"},
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double cpu_seconds() { rusage u; getrusage(RUSAGE_SELF, &u); return u.ru_utime.tv_sec + u.ru_utime.tv_usec * 1e-6 + u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6; }
}  // namespace

int main(int argc, char** argv)
{
    int steps = 320, warmup = 4, contexts = 6, depth = 4, gpus = 1, sorter = 1, coder = 1, lzpH = 0, lzpM = 0;
    long long n = 64ll << 20; unsigned long long seed = 2; const char* dump = nullptr;
    bool pin_input = false, upfront = false;
    for (int a = 1; a < argc; ++a) {
        auto val = [&]() -> const char* { return a + 1 < argc ? argv[++a] : "0"; };
        const std::string k = argv[a];
        if (k == "--steps") steps = atoi(val()); else if (k == "--warmup") warmup = atoi(val()); else if (k == "--contexts") contexts = atoi(val());
        else if (k == "--depth") depth = atoi(val()); else if (k == "--gpus") gpus = atoi(val()); else if (k == "--block") n = atoll(val());
        else if (k == "--sorter") sorter = atoi(val()); else if (k == "--coder") coder = atoi(val()); else if (k == "--seed") seed = strtoull(val(), nullptr, 10);
        else if (k == "--lzp") { const char* v = val(); lzpH = atoi(v); const char* c = strchr(v, ','); lzpM = c ? atoi(c + 1) : 0; }
        else if (k == "--dump") dump = val();
        else if (k == "--pin-input") pin_input = true;           // the input buffer page-locked (hipHostRegister): what an embedder with pinned I/O buffers has
        else if (k == "--upfront") upfront = true;               // every block of a phase added at once, own output buffer each: no collector in the way
        else { fprintf(stderr, "job_bench: unknown option %s\n", k.c_str()); return 2; }
    }
    if (steps < 1 || warmup < 0 || n < 1 || n > (1ll << 30)) return 2;
    const int features = LIBBSC_FEATURE_FASTMODE | LIBBSC_FEATURE_MULTITHREADING;
    if (bsc_init(features) != LIBBSC_NO_ERROR) return 1;
    int ndev = bscgpu_device_count();
    if (ndev <= 0) { fprintf(stderr, "job_bench: no usable GPU\n"); return 1; }
    if (gpus > 0 && gpus < ndev) ndev = gpus;
    std::vector<int> devs; for (int d = 0; d < ndev; ++d) devs.push_back(d);

    std::vector<unsigned char> input((size_t)n);
    if (bsc_synth_text_v1(seed, input.data(), n) != LIBBSC_NO_ERROR) return 1;
    if (pin_input && hipHostRegister(input.data(), (size_t)n, 0) != 0) { fprintf(stderr, "job_bench: hipHostRegister failed\n"); return 1; }
    // output buffers are recycled in a ring longer than what can be in flight: block b's buffer is free again once b has been waited for
    // (--upfront: one buffer per block of the longest phase; plain malloc — a buffer is touched only as far as its compressed block reaches)
    const int window = upfront ? (steps > ndev * contexts * depth + warmup ? steps : ndev * contexts * depth + warmup) + 1 : ndev * contexts * (depth + 1) + 2;
    std::vector<unsigned char*> outs((size_t)window);
    for (auto& o : outs) { o = (unsigned char*)malloc((size_t)n + LIBBSC_HEADER_SIZE); if (!o) return 1; }

    int last_size = 0; const unsigned char* last_block = nullptr;
    auto run = [&](bscgpu_job* job, int first, int count, bool announce) -> int {
        // adds run `window - 1` blocks ahead of the waits at most (the ring); returns 0 or the first error
        if (announce && bscgpu_job_expect(job, first + count) != LIBBSC_NO_ERROR) return -1;
        int added = first, waited = first;
        while (waited < first + count) {
            while (added < first + count && added - waited < window - 1) {
                const int id = bscgpu_job_add(job, input.data(), outs[(size_t)(added % window)], (int)n, lzpH, lzpM, sorter, coder, features);
                if (id != added) { fprintf(stderr, "job_bench: bscgpu_job_add -> %d\n", id); return id < 0 ? id : -1; }
                ++added;
            }
            const int r = bscgpu_job_wait(job, waited);
            if (r < 0) { fprintf(stderr, "job_bench: block %d -> %d\n", waited, r); return r; }
            last_size = r; last_block = outs[(size_t)(waited % window)];
            ++waited;
        }
        return 0;
    };

    // Set-up, untimed, as bench.py does it: every slot of every pipe once (arenas, pinned landing zones) plus the warm-up blocks.  Contexts
    // belong to a job, so set-up, warm-up and the timed region are ONE job; it has run dry when the timed region begins (everything was
    // waited for), so the timed blocks are a burst of their own — tapered head — and their total is announced — tapered tail.
    bscgpu_job* job = nullptr;
    const double t_create = now();
    int rc = bscgpu_job_create(&job, devs.data(), ndev, contexts, depth, n);
    if (rc != LIBBSC_NO_ERROR) { fprintf(stderr, "job_bench: bscgpu_job_create -> %d\n", rc); return 1; }
    const double create_s = now() - t_create;
    const int setup_blocks = ndev * contexts * depth + warmup;
    const double t_setup = now();
    if (run(job, 0, setup_blocks, false) != 0) return 1;
    const double setup_s = now() - t_setup;

    uint64_t shapes0[4]; bscgpu_coder_pool_stats(shapes0, 1);
    const double c0 = cpu_seconds(), t0 = now();
    if (run(job, setup_blocks, steps, true) != 0) return 1;
    const double dt = now() - t0, cpu = cpu_seconds() - c0;
    uint64_t shapes[4]; bscgpu_coder_pool_stats(shapes, 0);

    // the last block against the committed reference output
    const char* verified = "null"; std::string note = "no committed reference output for this configuration";
    const std::string md5 = Md5().hex(last_block, (size_t)last_size);
    if (lzpH == 0)
        for (const Golden& g : kGolden)
            if (g.seed == seed && g.n == n && g.sorter == sorter && g.coder == coder) {
                const bool ok = g.size == last_size && md5 == g.md5;
                verified = ok ? "true" : "false";
                note = ok ? "last timed block: size + md5 equal the reference libbsc output committed in tests/golden/golden_big.json" : "MISMATCH against tests/golden/golden_big.json";
            }
    if (dump) { if (FILE* f = fopen(dump, "wb")) { fwrite(last_block, 1, (size_t)last_size, f); fclose(f); } }
    bscgpu_job_destroy(job);

    const double mbps = (double)n * steps / 1e6 / dt;
    printf("{\"metric\": \"MB/s compress through the C job driver (bscgpu_job_*), host-resident input\", \"value\": %.1f, \"unit\": \"MB/s\", \"n_gpus\": %d, "
           "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.3f, \"config\": {\"workload\": \"%d x %lld-byte synth-text-v1 blocks (seed %llu), "
           "bsc_compress(lzp %d,%d sorter %d coder %d) through ONE bscgpu_job: %d GPU(s) x %d context(s) x %d in flight, total announced "
           "(tapered tail, low-latency last blocks), input in pageable host memory (one H2D per block)\", \"contexts\": %d, \"depth\": %d}, "
           "\"verified\": %s, \"verified_note\": \"%s\", \"compressed_bytes\": %d, \"md5\": \"%s\", \"cpu_seconds_per_block\": %.4f, "
           "\"coder_task_shapes\": {\"scalar_tasks\": %llu, \"pair_tasks\": %llu, \"eight_lane_task\": %llu, \"host_model\": %llu}, "
           "\"create_s\": %.3f, \"setup_s\": %.3f, \"setup_blocks\": %d}\n",
           mbps, ndev, steps, warmup, dt * 1e3 / steps, steps, n, seed, lzpH, lzpM, sorter, coder, ndev, contexts, depth, contexts, depth,
           verified, note.c_str(), last_size, md5.c_str(), cpu / steps,
           (unsigned long long)shapes[0], (unsigned long long)shapes[1], (unsigned long long)shapes[2], (unsigned long long)shapes[3],
           create_s, setup_s, setup_blocks);
    return strcmp(verified, "false") == 0 ? 3 : 0;
}
