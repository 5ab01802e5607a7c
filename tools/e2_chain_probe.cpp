// e2_chain_probe.cpp — measurement (not part of the product): how the adaptive coder's (-e2) MIXER work is distributed over mixer
// instances.  The three counters behind a decision are independent chains with monotone update maps and run on the device model's
// machinery; the mixer (predictor.h:74-213) is neither monotone nor composable, so an instance is a strictly serial chain:
// step i needs the weights and map cells step i-1 left.  This probe walks sub-blocks with the product's own decision walker
// (csrc/host/qlfc.cpp, walk_model1) and counts, per sub-block, the steps of every mixer instance the walk touches
// (qlfc.cpp:590,623,635,651,692,728,759,771,786 of the reference choose the instance by symbol / history / bit position).
//   g++ -O2 -std=c++17 -march=x86-64-v3 -I libbsc_amd/csrc/host -I include tools/e2_chain_probe.cpp -o /tmp/e2_chain_probe
//   /tmp/e2_chain_probe <sorted block (BWT output)> <sub-blocks>        (tools/e2_chain_probe.py drives it and writes profiles/r05/)
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
using namespace bschost;
struct CountPolicy {
    const Mixer* base; std::vector<uint64_t>* steps; uint64_t decisions = 0;
    struct Live {};
    inline bool begin_run() { return true; }
    inline Live enter() { return Live(); } inline void leave(const Live&) {}
    template <int CLS> inline void decide(Live&, unsigned, short&, short&, short&, Mixer* mx) { ++(*steps)[(size_t)(mx - base)]; ++decisions; }
};
static const char* family(size_t i, size_t* rel)
{
    static const struct { const char* name; size_t count; } F[] = {{"rank[c]", 256}, {"rank_exp[h][b]", 64}, {"rank_mant[bits]", 8}, {"rank_esc[ctx]", 256},
                                                                   {"run[c]", 256}, {"run_exp[h][b]", 1024}, {"run_mant[bits]", 32}};
    for (const auto& f : F) { if (i < f.count) { *rel = i; return f.name; } i -= f.count; }
    *rel = i; return "?";
}
int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> in((size_t)n); if (fread(in.data(), 1, (size_t)n, f) != (size_t)n) return 1; fclose(f);
    const int nsub = atoi(argv[2]);
    const QlfcTables& T = qlfc_tables();
    const size_t NM = sizeof(Mixers1) / sizeof(Mixer);
    printf("# %ld bytes, %d sub-blocks cut as the coder cuts them (coder.cpp:70-109); %zu mixer instances in the model (Mixers1)\n", n, nsub, NM);
    // the coder's own split (coder.cpp:70-109): cut where the sampled run starts reach total / nsub
    std::vector<long> cut(1, 0);
    {
        long total = 0;
        for (long i = 1; i < n; i += 32) total += in[(size_t)i] != in[(size_t)i - 1];
        if (total > nsub) {
            const long per = total / nsub; long seen = 0;
            for (long i = 1; i < n && (int)cut.size() < nsub; i += 32) if (in[(size_t)i] != in[(size_t)i - 1] && ++seen == per) { seen = 0; cut.push_back(i); }
        } else for (int b = 1; b < nsub; ++b) cut.push_back(n / nsub * b);
        cut.push_back(n);
    }
    for (int b = 0; b < nsub; ++b) {
        const long lo = cut[(size_t)b], hi = cut[(size_t)b + 1];
        QlfcRuns R; qlfc_runs(in.data() + lo, (int)(hi - lo), R);
        Counters1* Cn = tl_counters(); Mixers1* Mx = tl_mixers(T);
        std::vector<uint64_t> steps(NM, 0);
        CountPolicy pol{reinterpret_cast<const Mixer*>(Mx), &steps};
        // max_rank as the stream header fixes it (qlfc.cpp:888): bsr(symbols - 1)
        int nsym = R.view.nsym; int max_rank = 0; while ((2 << max_rank) <= nsym - 1) ++max_rank; if (nsym < 2) max_rank = 0;
        walk_model1<true>(R.view, T, max_rank, *Cn, Mx, pol);
        std::vector<std::pair<uint64_t, size_t>> order;
        for (size_t i = 0; i < NM; ++i) if (steps[i]) order.push_back({steps[i], i});
        std::sort(order.rbegin(), order.rend());
        const uint64_t D = pol.decisions;
        auto pct = [&](double q) { return order[(size_t)((order.size() - 1) * q)].first; };
        printf("sub-block %d: %u runs, %llu decisions, %zu mixer instances touched; chain length longest %llu  p99 %llu  p90 %llu  p50 %llu  shortest %llu\n", b, R.view.count,
               (unsigned long long)D, order.size(), (unsigned long long)order[0].first, (unsigned long long)pct(0.01), (unsigned long long)pct(0.10),
               (unsigned long long)pct(0.50), (unsigned long long)order.back().first);
        uint64_t acc = 0;
        for (size_t k = 0; k < order.size() && k < 12; ++k) {
            size_t rel; const char* fam = family(order[k].second, &rel);
            acc += order[k].first;
            printf("    #%zu  %-16s [%3zu]  %9llu steps  %5.1f %% of the decisions, %5.1f %% with the longer ones\n", k + 1, fam, rel,
                   (unsigned long long)order[k].first, 100.0 * order[k].first / D, 100.0 * acc / D);
        }
        // what a device share could take: everything but the K longest chains, and what stays serial on the host
        for (int K : {1, 2, 4, 8, 16, 32}) {
            uint64_t host = 0; for (int k = 0; k < K && (size_t)k < order.size(); ++k) host += order[(size_t)k].first;
            const uint64_t longest_left = (size_t)K < order.size() ? order[(size_t)K].first : 0;
            printf("    K = %2d longest chains on the host: %5.1f %% of the decisions stay there; the longest chain left for a device lane has %llu steps\n", K,
                   100.0 * host / D, (unsigned long long)longest_left);
        }
    }
    return 0;
}
