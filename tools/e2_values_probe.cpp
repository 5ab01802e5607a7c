// e2_values_probe.cpp — feasibility probe (not part of the product): the adaptive coder (-e2) with the three counter values of every decision handed in, as a
// device share of its model would deliver them, against the full host coder.  g++ -O3 -std=c++17 -march=x86-64-v3 -I libbsc_amd/csrc/host -I include tools/e2_values_probe.cpp -o /tmp/e2_probe && /tmp/e2_probe <sorted block>
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace bschost;
struct V3 { short ch, st, sp; };
struct LogPolicy {          // full model, logs the values every decision sees
    RangeEncoder& rc; const QlfcTables& T; std::vector<V3>* log;
    using Live = RangeEncoder::Live;
    inline bool begin_run() { return !rc.full(); }
    inline Live enter() { return rc.enter(); } inline void leave(const Live& L) { rc.leave(L); }
    template <int CLS> inline void decide(Live& L, unsigned bit, short& st, short& ch, short& sp, Mixer* mx) { log->push_back(V3{ch, st, sp}); bschost::decide<CLS, true>(rc, L, T, bit, st, ch, sp, mx); }
};
struct ValuesPolicy {       // mixer + range coder only: counter values come from the stream
    RangeEncoder& rc; const QlfcTables& T; const V3* v;
    using Live = RangeEncoder::Live;
    inline bool begin_run() { return !rc.full(); }
    inline Live enter() { return rc.enter(); } inline void leave(const Live& L) { rc.leave(L); }
    template <int CLS> __attribute__((always_inline)) inline void decide(Live& L, unsigned bit, short&, short&, short&, Mixer* mx)
    {
        constexpr const short* P = kAdaptiveParams[CLS];
        const V3 x = *v++;
        const int s0 = T.stretch[x.ch], s1 = T.stretch[x.st], s2 = T.stretch[x.sp];
        short sp16 = (short)(wrap_add3(wrap_mul(s0, mx->w0), wrap_mul(s1, mx->w1), wrap_mul(s2, mx->w2)) >> 17);
        if (sp16 < -2047) sp16 = -2047;
        if (sp16 > 2047) sp16 = 2047;
        const int frac = sp16 & 255, idx = (sp16 + 2048) >> 8, sq = T.squash[2048 + sp16];
        const int mapped = mx->map[idx] + (((mx->map[idx + 1] - mx->map[idx]) * frac) >> 8);
        const int p = (3 * sq + mapped) >> 2;
        bump(mx->map[idx], bit, P[12], P[13], P[14], P[15]);
        bump(mx->map[idx + 1], bit, P[12], P[13], P[14], P[15]);
        const int eps = p - (bit ? 1 : 4095);
        mx->w0 = (int)((uint32_t)mx->w0 - (uint32_t)(wrap_mul(wrap_mul(P[16], eps), s0) >> 16));
        mx->w1 = (int)((uint32_t)mx->w1 - (uint32_t)(wrap_mul(wrap_mul(P[17], eps), s1) >> 16));
        mx->w2 = (int)((uint32_t)mx->w2 - (uint32_t)(wrap_mul(wrap_mul(P[18], eps), s2) >> 16));
        rc.encode_live<12>(L, bit, p);
    }
};
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> in(n); if (fread(in.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
    if (n > (8 << 20)) { n = 8 << 20; in.resize(n); }
    QlfcRuns R; qlfc_runs(in.data(), (int)n, R);
    const QlfcTables& T = qlfc_tables();
    std::vector<uint8_t> o1(n + 4096), o2(n + 4096);
    std::vector<V3> log; log.reserve(40000000);
    int r1 = 0, r2 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        r1 = qlfc_encode_runs(R.view, (int)n, o1.data(), (int)n, CODER_ADAPTIVE);
        printf("full adaptive coder: %.1f ms -> %d bytes\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), r1);
    }
    {   // log pass
        Counters1* Cn = tl_counters(); Mixers1* Mx = tl_mixers(T);
        RangeEncoder rc; rc.init(o2.data(), (int)n); rc.encode_word((uint32_t)n);
        const int max_rank = encode_alphabet(R.view, [&](unsigned b) { rc.encode_half(b); });
        LogPolicy pol{rc, T, &log};
        walk_model1<true>(R.view, T, max_rank, *Cn, Mx, pol);
        rc.finish();
    }
    for (int rep = 0; rep < 3; ++rep) {
        Counters1* Cn = tl_counters(); Mixers1* Mx = tl_mixers(T);
        auto t0 = std::chrono::steady_clock::now();
        RangeEncoder rc; rc.init(o2.data(), (int)n); rc.encode_word((uint32_t)n);
        const int max_rank = encode_alphabet(R.view, [&](unsigned b) { rc.encode_half(b); });
        ValuesPolicy pol{rc, T, log.data()};
        walk_model1<true>(R.view, T, max_rank, *Cn, Mx, pol);
        r2 = rc.finish();
        printf("mixer + range coder from handed-in counter values: %.1f ms -> %d bytes (%s), %zu decisions\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), r2,
               (r1 == r2 && memcmp(o1.data(), o2.data(), r1) == 0) ? "identical" : "DIFFERENT", log.size());
    }
    return 0;
}
