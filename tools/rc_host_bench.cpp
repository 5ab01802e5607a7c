// rc_host_bench.cpp — what the HOST half of the static coder costs on the box's own CPU, one thread, real entries (not part of the product).
// The bench block (64 MiB synth-text v1, seed 2) goes through the library's BWT and device model once; its probability stream is then
// written into a pinned landing zone by the GPU's DMA engine before EVERY timed run (so every line comes from DRAM, as in the product),
// and coded by each task shape: one eight-lane task (round 4's and round 5's step, several software-prefetch distances), four two-stream tasks, eight
// single-stream tasks — run one after the other on this thread, i.e. the figures are CPU-seconds per block for each shape.
// The same streams from ordinary (malloc) memory, cache-cold, give the cost of the landing zone itself.
// (profiles/r05/host_coder_on_box_cpu.txt also holds the runs of intermediate builds: timing-only builds of the step without its log /
// without the low word / with loads and transposes alone, and two more variants of the step that were not kept.)
//   hipcc -O3 -std=c++17 -x c++ -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -march=x86-64-v3 -mtune=znver4 -I include tools/rc_host_bench.cpp -x none \
//         -L libbsc_amd/lib -lbsc_mi355x -L /opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -o libbsc_amd/lib/rc_host_bench
//   libbsc_amd/lib/rc_host_bench [n_bytes] [seed] [quick]
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include "bscgpu.h"
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <thread>

using namespace bschost;
#ifdef RC_HOST_BENCH_OLD_CODER                 // built against a round-4 checkout of qlfc.cpp for the comparison: no prefetch knob there
static int g_x8_prefetch_override = -1;
#endif
extern "C" int bsc_synth_text_v1(unsigned long long seed, unsigned char* out, long long n);

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : (64 << 20);
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 2;
    const bool quick = argc > 3;                       // third argument: the eight-lane legs on the landing zone only
    std::vector<uint8_t> T((size_t)n), L((size_t)n);
    if (bsc_synth_text_v1(seed, T.data(), n) != 0) { fprintf(stderr, "synth failed\n"); return 1; }
    bscgpu_ctx* c = nullptr;
    if (bscgpu_create(&c, 0, (int64_t)n + 4096) != 0) { fprintf(stderr, "no context\n"); return 1; }
    if (bscgpu_bwt(c, T.data(), L.data(), n) < 0) { fprintf(stderr, "bwt failed: %s\n", bscgpu_last_error(c)); return 1; }
    const int64_t cap = (int64_t)n * 4;
    uint16_t* pinned = nullptr; uint16_t* dcopy = nullptr;
    if (hipHostMalloc((void**)&pinned, (size_t)cap * 2, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "no pinned memory\n"); return 1; }
    int nb = 0, sub_start[8], sub_size[8]; int64_t poff[9];
    const int64_t D = bscgpu_qlfc_static_pstream(c, L.data(), n, pinned, cap, &nb, sub_start, sub_size, poff, nullptr);
    if (D < 0 || D > cap || nb != 8) { fprintf(stderr, "pstream failed: %lld (%s), nb %d\n", (long long)D, bscgpu_last_error(c), nb); return 1; }
    if (hipMalloc((void**)&dcopy, (size_t)D * 2) != hipSuccess || hipMemcpy(dcopy, pinned, (size_t)D * 2, hipMemcpyHostToDevice) != hipSuccess) return 1;
    std::vector<uint16_t> plain((size_t)D);
    memcpy(plain.data(), pinned, (size_t)D * 2);
    size_t starts = 0; for (int64_t i = 0; i < D; ++i) starts += (plain[(size_t)i] >> 13) & 1u;
    printf("block %d bytes, %lld decisions (%.2f per byte), run starts %.1f %%, sub-block streams:", n, (long long)D, (double)D / n, 100.0 * starts / D);
    for (int b = 0; b < 8; ++b) printf(" %lld", (long long)(poff[b + 1] - poff[b]));
    printf("\n");

    uint8_t first_seen[256]; for (int s = 0; s < 256; ++s) first_seen[s] = (uint8_t)s;
    std::vector<uint8_t> out[8], ref[8];
    for (int b = 0; b < 8; ++b) { out[b].resize((size_t)sub_size[b] + 4096); ref[b].resize((size_t)sub_size[b] + 4096); }
    std::vector<uint8_t> evict((size_t)512 << 20, 1);
    auto jobs = [&](const uint16_t* base, PstreamJob* J, std::vector<uint8_t>* o) {
        for (int b = 0; b < 8; ++b) J[b] = PstreamJob{first_seen, 96, sub_size[b], base + poff[b], (size_t)(poff[b + 1] - poff[b]), o[b].data(), sub_size[b]};
    };
    // cold: pinned <- DMA from the device copy; plain <- the CPU's own stores pushed out by half a gigabyte of other lines
    auto make_cold = [&](bool pin) {
        if (pin) { if (hipMemcpy(pinned, dcopy, (size_t)D * 2, hipMemcpyDeviceToHost) != hipSuccess) exit(1); }
        else { volatile unsigned long long sink = 0; for (size_t i = 0; i < evict.size(); i += 64) sink = sink + evict[i]; }
    };
    int refres[8];
    { PstreamJob J[8]; jobs(plain.data(), J, ref); for (int b = 0; b < 8; ++b) refres[b] = qlfc_encode_static_pstream(J[b].first_seen, J[b].nsym, J[b].in_size, J[b].ps, J[b].count, J[b].out, J[b].out_size); }
    long long total = 0; for (int b = 0; b < 8; ++b) total += refres[b] > 0 ? refres[b] : 0;
    printf("coded: %lld bytes over eight sub-blocks\n", total);
    int bad = 0;
    auto check = [&](const char* what, const int* r) {
        for (int b = 0; b < 8; ++b) if (r[b] != refres[b] || (r[b] > 0 && memcmp(out[b].data(), ref[b].data(), (size_t)r[b]) != 0)) { printf("MISMATCH %s sub-block %d: %d vs %d\n", what, b, r[b], refres[b]); ++bad; }
    };
    for (int pin = 1; pin >= (quick ? 1 : 0); --pin) {
        const uint16_t* base = pin ? pinned : plain.data();
        printf("== entries in %s\n", pin ? "the pinned landing zone, written by DMA before every run" : "malloc memory, evicted before every run");
#ifndef RC_HOST_BENCH_OLD_CODER
        for (int vsel = 0; vsel <= 2; vsel += 2) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                PstreamJob J[8]; jobs(base, J, out); int r[8];
                make_cold(pin != 0);
                g_x8_prefetch_override = 256; g_x8_vsel_override = vsel;
                const double t0 = now_ms();
                const bool ok = qlfc_encode_static_pstream_x8(J, r);
                const double ms = now_ms() - t0;
                if (!ok) { printf("x8 gave up\n"); ++bad; break; }
                if (rep == 0) check("x8", r);
                if (ms < best) best = ms;
            }
            printf("  eight-lane task, %s: %7.1f ms per block  (%.3f ns per decision)\n", vsel ? "round 5's step               " : "the round-4 step (BSC_RC_VSEL=0)", best, best * 1e6 / D);
        }
        g_x8_vsel_override = -1;
#endif
#ifdef RC_HOST_BENCH_OLD_CODER
        const int pfs[] = {0};
#else
        const int pfs[] = {0, 256, 1024};
#endif
        for (int pf : pfs) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                PstreamJob J[8]; jobs(base, J, out); int r[8];
                make_cold(pin != 0);
                g_x8_prefetch_override = pf;
                const double t0 = now_ms();
                const bool ok = qlfc_encode_static_pstream_x8(J, r);
                const double ms = now_ms() - t0;
                if (!ok) { printf("x8 gave up\n"); ++bad; break; }
                if (rep == 0) check("x8", r);
                if (ms < best) best = ms;
            }
            printf("  eight-lane task, prefetch %4d entries ahead: %7.1f ms per block  (%.3f ns per decision)\n", pf, best, best * 1e6 / D);
        }
        if (!quick) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                PstreamJob J[8]; jobs(base, J, out); int r[8];
                make_cold(pin != 0);
                const double t0 = now_ms();
                for (int b = 0; b < 8; b += 2) qlfc_encode_static_pstream_pair(J[b], J[b + 1], &r[b], &r[b + 1]);
                const double ms = now_ms() - t0;
                if (rep == 0) check("pairs", r);
                if (ms < best) best = ms;
            }
            printf("  four two-stream tasks, one after the other:   %7.1f ms per block  (%.3f ns per decision; a task %.1f ms)\n", best, best * 1e6 / D, best / 4);
        }
        if (!quick) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                PstreamJob J[8]; jobs(base, J, out); int r[8];
                make_cold(pin != 0);
                const double t0 = now_ms();
                for (int b = 0; b < 8; ++b) r[b] = qlfc_encode_static_pstream(J[b].first_seen, J[b].nsym, J[b].in_size, J[b].ps, J[b].count, J[b].out, J[b].out_size);
                const double ms = now_ms() - t0;
                if (rep == 0) check("singles", r);
                if (ms < best) best = ms;
            }
            printf("  eight single-stream tasks, one after the other: %7.1f ms per block  (%.3f ns per decision; a task %.1f ms)\n", best, best * 1e6 / D, best / 8);
        }
    }
#ifndef RC_HOST_BENCH_OLD_CODER
    // round 6: the same block's stream in its packed form (13 bits per decision, eight decisions in 13 bytes), in a pinned landing zone written
    // by DMA before every run: the three task shapes again
    {
        const int64_t cap13 = (int64_t)D * 13 / 8 + 8 * 104 + 64;
        uint8_t* pin13 = nullptr; uint8_t* d13 = nullptr;
        int nb2 = 0, st2[8], sz2[8]; int64_t poff2[9], pbase[9];
        if (hipHostMalloc((void**)&pin13, (size_t)cap13 + 64, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "no pinned memory\n"); return 1; }
        const int64_t D2 = bscgpu_qlfc_static_pstream_packed(c, L.data(), n, pin13, cap13, &nb2, st2, sz2, poff2, pbase);
        if (D2 != D || nb2 != 8) { fprintf(stderr, "packed pstream failed: %lld (%s)\n", (long long)D2, bscgpu_last_error(c)); return 1; }
        const size_t bytes13 = (size_t)pbase[8] / 8 * 13;
        if (hipMalloc((void**)&d13, bytes13) != hipSuccess || hipMemcpy(d13, pin13, bytes13, hipMemcpyHostToDevice) != hipSuccess) return 1;
        printf("== packed stream (13 bits per decision): %zu bytes against %lld as 16-bit entries; pinned landing zone, written by DMA before every run\n", bytes13, (long long)D * 2);
        auto jobs13 = [&](PstreamJob* J, std::vector<uint8_t>* o) {
            for (int b = 0; b < 8; ++b) J[b] = PstreamJob{first_seen, 96, sub_size[b], reinterpret_cast<const uint16_t*>(pin13 + (size_t)pbase[b] / 8 * 13), (size_t)(poff[b + 1] - poff[b]), o[b].data(), sub_size[b]};
        };
        auto cold13 = [&] { if (hipMemcpy(pin13, d13, bytes13, hipMemcpyDeviceToHost) != hipSuccess) exit(1); };
        for (int shape = 8; shape >= 1; shape = shape == 8 ? 2 : shape == 2 ? 1 : 0) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                PstreamJob J[8]; jobs13(J, out); int r[8];
                cold13();
                g_x8_prefetch_override = 256; g_x8_vsel_override = -1;
                const double t0 = now_ms();
                if (shape == 8) { if (!qlfc_encode_static_p13_x8(J, r)) { printf("packed x8 gave up\n"); ++bad; break; } }
                else if (shape == 2) for (int b = 0; b < 8; b += 2) qlfc_encode_static_p13_pair(J[b], J[b + 1], &r[b], &r[b + 1]);
                else for (int b = 0; b < 8; ++b) r[b] = qlfc_encode_static_p13(J[b].first_seen, J[b].nsym, J[b].in_size, reinterpret_cast<const uint8_t*>(J[b].ps), J[b].count, J[b].out, J[b].out_size);
                const double ms = now_ms() - t0;
                if (rep == 0) check(shape == 8 ? "packed x8" : shape == 2 ? "packed pairs" : "packed singles", r);
                if (ms < best) best = ms;
            }
            printf("  packed, %s: %7.1f ms per block  (%.3f ns per decision)\n", shape == 8 ? "eight-lane task                " : shape == 2 ? "four two-stream tasks          " : "eight single-stream tasks      ", best, best * 1e6 / D);
        }
        g_x8_prefetch_override = -1;
        hipFree(d13); hipHostFree(pin13);
    }
#endif
    // six eight-lane tasks at once, every one on its own landing zone (the steady state of a six-context job: the tasks share the
    // memory system, not the entries)
    if (!quick) {
        const int NT = 6;
        uint16_t* zone[NT] = {};
        std::vector<std::vector<uint8_t>> o((size_t)NT * 8);
        bool have = true;
        for (int t = 0; t < NT; ++t) {
            if (hipHostMalloc((void**)&zone[t], (size_t)D * 2, hipHostMallocDefault) != hipSuccess) { have = false; break; }
            for (int b = 0; b < 8; ++b) o[(size_t)t * 8 + b].resize((size_t)sub_size[b] + 4096);
        }
        if (have) {
            printf("== %d eight-lane tasks at once, each on its own DMA-written landing zone\n", NT);
            const int pfs[] = {0, 256, 1024};
            for (int pf : pfs) {
                double best_mean = 1e30, best_max = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    for (int t = 0; t < NT; ++t) if (hipMemcpy(zone[t], dcopy, (size_t)D * 2, hipMemcpyDeviceToHost) != hipSuccess) return 1;
                    g_x8_prefetch_override = pf;
                    double ms[NT]; bool okk[NT];
                    std::vector<std::thread> th;
                    for (int t = 0; t < NT; ++t) th.emplace_back([&, t] {
                        PstreamJob J[8]; int r[8];
                        for (int b = 0; b < 8; ++b) J[b] = PstreamJob{first_seen, 96, sub_size[b], zone[t] + poff[b], (size_t)(poff[b + 1] - poff[b]), o[(size_t)t * 8 + b].data(), sub_size[b]};
                        const double t0 = now_ms();
                        okk[t] = qlfc_encode_static_pstream_x8(J, r);
                        ms[t] = now_ms() - t0;
                        for (int b = 0; b < 8; ++b) if (r[b] != refres[b]) okk[t] = false;
                    });
                    for (auto& x : th) x.join();
                    double mean = 0, mx = 0; for (int t = 0; t < NT; ++t) { mean += ms[t] / NT; if (ms[t] > mx) mx = ms[t]; if (!okk[t]) ++bad; }
                    if (mean < best_mean) { best_mean = mean; best_max = mx; }
                }
                printf("  prefetch %4d entries ahead: %7.1f ms per task on average, slowest %7.1f ms\n", pf, best_mean, best_max);
            }
        }
        for (int t = 0; t < NT; ++t) if (zone[t]) hipHostFree(zone[t]);
    }
    printf(bad ? "FAILED\n" : "all outputs equal\n");
    hipFree(dcopy); hipHostFree(pinned); bscgpu_destroy(c);
    return bad != 0;
}
