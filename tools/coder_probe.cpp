// coder_probe.cpp — CPU study behind the device-side static coder: decision statistics of one sub-block,
// coalescence of the two-sided counter brackets per (chunk, slot), and the speed of a range coder fed by a precomputed
// probability stream.  Not part of the product.
//   g++ -O2 -std=c++17 -march=native -I libbsc_amd/csrc/host -I include tools/coder_probe.cpp libbsc_amd/csrc/host/coder.cpp -o /tmp/coder_probe
//   /tmp/coder_probe /tmp/bwt_64m_s2.bin [chunk_runs]
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include <chrono>
#include <cstdio>
#include <map>
#include <unordered_map>

using namespace bschost;

struct Dec { uint32_t st, ch, sp; uint8_t bit, cls; };
struct LogPolicy {
    Counters1* base; std::vector<Dec>* out; std::vector<uint32_t>* run_first;
    struct Live {}; inline Live enter() { return Live(); } inline void leave(const Live&) {}
    inline bool begin_run() { run_first->push_back((uint32_t)out->size()); return true; }
    template <int CLS> inline void decide(Live&, unsigned bit, short& st, short& ch, short& sp, Mixer*)
    {
        const short* b = reinterpret_cast<const short*>(base);
        out->push_back(Dec{(uint32_t)(&st - b), (uint32_t)(&ch - b), (uint32_t)(&sp - b), (uint8_t)bit, (uint8_t)CLS});
    }
};

static inline int step(int v, int bit, const short* P, int fam)
{
    const int th0 = P[4 * fam + 0], ar0 = P[4 * fam + 1], th1 = P[4 * fam + 2], ar1 = P[4 * fam + 3];
    if (bit) return v - (((v - th1) * ar1) >> 12);
    return v + (((4096 - th0 - v) * ar0) >> 12);
}

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> L(n);
    if (fread(L.data(), 1, n, f) != (size_t)n) return 1;
    fclose(f);
    const int chunk_runs = argc > 2 ? atoi(argv[2]) : 8192;
    const int nb = coder_num_blocks((int)n);
    int start[8], size[8];
    coder_split_blocks(L.data(), (int)n, nb, start, size);
    printf("n=%ld sub-blocks=%d first size=%d\n", n, nb, size[0]);

    const QlfcTables& T = qlfc_tables();
    QlfcRuns R; qlfc_runs(L.data() + start[0], size[0], R);
    std::unique_ptr<Counters1> K(new_counters());
    std::vector<Dec> D; std::vector<uint32_t> rf;
    D.reserve(30000000);
    const int max_rank = encode_alphabet(R.view, [](unsigned) {});
    LogPolicy pol{K.get(), &D, &rf};
    walk_model1<false>(R.view, T, max_rank, *K, nullptr, pol);
    rf.push_back((uint32_t)D.size());
    const size_t m = R.view.count;
    printf("runs=%zu decisions=%zu (%.2f per run) nsym=%d max_rank=%d\n", m, D.size(), (double)D.size() / m, R.view.nsym, max_rank);
    size_t per_cls[7] = {0};
    std::map<uint32_t, size_t> types, chs, sts;
    for (auto& d : D) { per_cls[d.cls]++; types[d.sp]++; chs[d.ch]++; sts[d.st]++; }
    for (int c = 0; c < 7; ++c) printf("  class %d: %zu\n", c, per_cls[c]);
    printf("distinct types(sp slots)=%zu ch slots=%zu st slots=%zu\n", types.size(), chs.size(), sts.size());
    {   // how many types cover 99.9 % of decisions
        std::vector<size_t> v; for (auto& t : types) v.push_back(t.second);
        std::sort(v.rbegin(), v.rend());
        size_t acc = 0; int k = 0;
        for (; k < (int)v.size() && acc < D.size() * 0.999; ++k) acc += v[k];
        printf("  types for 99.9%%: %d; top-5 counts: %zu %zu %zu %zu %zu\n", k, v[0], v[1], v[2], v[3], v[4]);
    }

    // exact p stream (for the rc timing) and true counter trajectories
    std::vector<uint16_t> pstream(D.size());
    {
        std::unique_ptr<Counters1> K2(new_counters());
        short* b = reinterpret_cast<short*>(K2.get());
        for (size_t i = 0; i < D.size(); ++i) {
            const Dec& d = D[i]; const short* P = kStaticParams[d.cls];
            const int p = (b[d.ch] * P[16] + b[d.st] * P[17] + b[d.sp] * P[18]) >> 5;
            pstream[i] = (uint16_t)(p | (d.bit << 15));
            b[d.st] = (short)step(b[d.st], d.bit, P, 0); b[d.ch] = (short)step(b[d.ch], d.bit, P, 1); b[d.sp] = (short)step(b[d.sp], d.bit, P, 2);
        }
    }
    // attainable ranges per class / family
    int vmin[7][3], vmax[7][3];
    for (int c = 0; c < 7; ++c) for (int fam = 0; fam < 3; ++fam) {
        int v = 2048; for (int i = 0; i < 100000; ++i) v = step(v, 0, kStaticParams[c], fam); vmax[c][fam] = v;
        v = 2048; for (int i = 0; i < 100000; ++i) v = step(v, 1, kStaticParams[c], fam); vmin[c][fam] = v;
        // the reachable set may exceed these through alternation; widen by scanning
        int lo = vmin[c][fam], hi = vmax[c][fam];
        for (int rep = 0; rep < 4; ++rep) for (int s = lo; s <= hi; ++s) for (int bit = 0; bit < 2; ++bit) { int w = step(s, bit, kStaticParams[c], fam); if (w < lo) lo = w; if (w > hi) hi = w; }
        vmin[c][fam] = lo; vmax[c][fam] = hi;
    }
    for (int c = 0; c < 7; ++c) printf("  class %d ranges: S[%d,%d] C[%d,%d] P[%d,%d]\n", c, vmin[c][0], vmax[c][0], vmin[c][1], vmax[c][1], vmin[c][2], vmax[c][2]);

    // family walks: partition key per decision's run
    // run-level keys: c = sym; state = st slot identifies (type,state) so state key is derived per decision, not per run;
    // to emulate the X-major chunking we chunk the *event list of each slot's partition*: partition of family fam =
    // runs sharing X.  X for ch = sym[run]; for st rank side / run side = the state; for sp = none.
    // Emulation: for each family, group decisions by (X, run order) -> chunk boundaries every chunk_runs runs of that X.
    std::vector<uint32_t> run_of(D.size());
    for (size_t r = 0; r < m; ++r) for (uint32_t i = rf[r]; i < rf[r + 1]; ++i) run_of[i] = (uint32_t)r;
    const char* famname[3] = {"state", "char", "static"};
    for (int fam = 0; fam < 3; ++fam) {
        // X key per decision: derive from slot index modulo structure is awkward; use: for ch -> sym, sp -> 0,
        // st -> (rank side or run side, state) = computed as slot-type pair: we key the partition by the pair (side, X)
        // where X is recovered as the value such that all decisions of one run+side share it: use the st slot of the
        // side's FIRST decision (RANK_FIRST / RUN_FIRST) which is unique per (side, state).
        std::vector<uint32_t> Xkey(D.size());
        for (size_t r = 0; r < m; ++r) {
            uint32_t xr = 0, xn = 0;
            for (uint32_t i = rf[r]; i < rf[r + 1]; ++i) { if (D[i].cls == RANK_FIRST || (D[i].cls == RANK_ESC && xr == 0)) xr = D[i].st + 1; if (D[i].cls == RUN_FIRST) xn = D[i].st + 1; }
            for (uint32_t i = rf[r]; i < rf[r + 1]; ++i) {
                if (fam == 2) Xkey[i] = 0;
                else if (fam == 1) Xkey[i] = R.view.sym[r];
                else Xkey[i] = (D[i].cls <= RANK_ESC) ? xr : (0x80000000u | xn);
            }
        }
        // per partition: ordinal of each run inside the partition -> chunk id
        std::unordered_map<uint32_t, uint32_t> part_runs;            // X -> runs seen so far
        std::unordered_map<uint64_t, std::vector<uint32_t>> chains;  // (slot) -> decision ids in order (already X-pure)
        std::vector<uint32_t> chunk_of(D.size());
        {
            std::unordered_map<uint32_t, uint32_t> last_run;         // X -> last run index counted
            for (size_t i = 0; i < D.size(); ++i) {
                const uint32_t x = Xkey[i];
                auto it = last_run.find(x);
                if (it == last_run.end() || it->second != run_of[i]) { last_run[x] = run_of[i]; part_runs[x]++; }
                chunk_of[i] = (part_runs[x] - 1) / chunk_runs;
            }
        }
        const uint32_t* slot = nullptr; (void)slot;
        size_t pairs = 0, pairs_uncoal = 0, pairs_uncoal_64 = 0, pairs_uncoal_256 = 0, ev_total = 0, ev_pre = 0, chunks_total = 0;
        for (auto& pr : part_runs) chunks_total += (pr.second + chunk_runs - 1) / chunk_runs;
        // iterate slots: build per-slot event lists
        std::unordered_map<uint32_t, std::vector<uint32_t>> ev;
        for (size_t i = 0; i < D.size(); ++i) ev[fam == 0 ? D[i].st : fam == 1 ? D[i].ch : D[i].sp].push_back((uint32_t)i);
        size_t worst_cnt = 0;
        for (auto& e : ev) {
            const std::vector<uint32_t>& ids = e.second;
            size_t a = 0;
            while (a < ids.size()) {
                size_t b = a; const uint32_t ck = chunk_of[ids[a]];
                while (b < ids.size() && chunk_of[ids[b]] == ck) ++b;
                const int cls = D[ids[a]].cls; const short* P = kStaticParams[cls];
                int lo = vmin[cls][fam], hi = vmax[cls][fam];
                size_t k = a;
                for (; k < b && lo != hi; ++k) { lo = step(lo, D[ids[k]].bit, P, fam); hi = step(hi, D[ids[k]].bit, P, fam); }
                ++pairs; ev_total += b - a; ev_pre += k - a;
                if (lo != hi) { ++pairs_uncoal; if (b - a > 64) ++pairs_uncoal_64; if (b - a > 256) { ++pairs_uncoal_256; if (b - a > worst_cnt) worst_cnt = b - a; } }
                a = b;
            }
        }
        printf("family %-6s: partitions=%zu chunks=%zu (chunk,slot) pairs=%zu  not coalesced: %zu (cnt>64: %zu, cnt>256: %zu, worst %zu)  events=%zu pre-coalescence=%zu (%.2f%%)\n",
               famname[fam], part_runs.size(), chunks_total, pairs, pairs_uncoal, pairs_uncoal_64, pairs_uncoal_256, worst_cnt, ev_total, ev_pre, 100.0 * ev_pre / ev_total);
    }


    // per-slot event chunking (explicit chain lists): chunk = EV events of ONE slot
    for (int EV : {4096, 16384, 65536}) {
        for (int fam = 0; fam < 3; ++fam) {
            std::unordered_map<uint32_t, std::vector<uint32_t>> ev;
            for (size_t i = 0; i < D.size(); ++i) ev[fam == 0 ? D[i].st : fam == 1 ? D[i].ch : D[i].sp].push_back((uint32_t)i);
            size_t hot_events = 0, cold_events = 0, cold_max = 0, chunks = 0, gap0 = 0, gapsmall = 0, gapbig = 0, cand_steps = 0, hot_slots = 0; int maxgap = 0;
            for (auto& e : ev) {
                const std::vector<uint32_t>& ids = e.second;
                if (ids.size() < (size_t)2 * EV) { cold_events += ids.size(); if (ids.size() > cold_max) cold_max = ids.size(); continue; }
                ++hot_slots; hot_events += ids.size();
                const int cls = D[ids[0]].cls; const short* P = kStaticParams[cls];
                int plo = 2048, phi = 2048;                   // candidate interval at the start of the chunk
                for (size_t a = 0; a < ids.size(); a += EV) {
                    const size_t b = std::min(ids.size(), a + (size_t)EV);
                    int lo = vmin[cls][fam], hi = vmax[cls][fam];
                    for (size_t k = a; k < b; ++k) { lo = step(lo, D[ids[k]].bit, P, fam); hi = step(hi, D[ids[k]].bit, P, fam); }
                    // candidate enumeration from [plo, phi]: distinct trajectories merge over time
                    std::vector<int> tr; for (int s = plo; s <= phi; ++s) tr.push_back(s);
                    for (size_t k = a; k < b && tr.size() > 1; ++k) {
                        for (auto& v : tr) v = step(v, D[ids[k]].bit, P, fam);
                        tr.erase(std::unique(tr.begin(), tr.end()), tr.end());
                        cand_steps += tr.size();
                    }
                    ++chunks; const int g = hi - lo; if (g == 0) ++gap0; else if (g <= 8) ++gapsmall; else ++gapbig; if (g > maxgap) maxgap = g;
                    plo = lo; phi = hi;
                }
            }
            printf("EV=%6d family %-6s: hot slots=%zu hot events=%zu chunks=%zu end gap==0: %zu, 1..8: %zu, >8: %zu (max %d); candidate steps=%zu (%.1f%% of hot events); cold events=%zu (longest cold chain %zu)\n",
                   EV, famname[fam], hot_slots, hot_events, chunks, gap0, gapsmall, gapbig, maxgap, cand_steps, 100.0 * cand_steps / std::max<size_t>(hot_events, 1), cold_events, cold_max);
        }
    }
    // range coder fed by the p stream
    {
        std::vector<uint8_t> out(size[0] + 1024);
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            RangeEncoder rc; rc.init(out.data(), (int)out.size());
            rc.encode_word((uint32_t)size[0]);
            encode_alphabet(R.view, [&](unsigned b) { rc.encode_half(b); });
            RangeEncoder::Live Lv = rc.enter();
            const uint16_t* ps = pstream.data(); const size_t nd = pstream.size();
            for (size_t i = 0; i < nd; ++i) { const unsigned x = ps[i]; rc.encode_live<12>(Lv, x >> 15, (int)(x & 0xfff)); }
            rc.leave(Lv);
            const int sz = rc.finish();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("rc from p stream: %d bytes, %.1f ms, %.2f ns/decision\n", sz, ms, ms * 1e6 / nd);
        }
        {   // two independent streams interleaved in one loop (the recurrence is latency-bound: range -> shift -> mul -> select)
            std::vector<uint8_t> oa(size[0] + 1024), ob(size[0] + 1024);
            for (int rep = 0; rep < 2; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                RangeEncoder ra, rb; ra.init(oa.data(), (int)oa.size()); rb.init(ob.data(), (int)ob.size());
                RangeEncoder::Live La = ra.enter(), Lb = rb.enter();
                const uint16_t* ps = pstream.data(); const size_t nd = pstream.size(), half = nd / 2;
                for (size_t i = 0; i < half; ++i) {
                    const unsigned x = ps[i], y = ps[half + i];
                    ra.encode_live<12>(La, x >> 15, (int)(x & 0xfff));
                    rb.encode_live<12>(Lb, y >> 15, (int)(y & 0xfff));
                }
                ra.leave(La); rb.leave(Lb);
                const int sa = ra.finish(), sb = rb.finish();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("rc, two interleaved streams: %d + %d bytes, %.1f ms, %.2f ns/decision\n", sa, sb, ms, ms * 1e6 / (2 * half));
            }
        }
        {   // four streams interleaved
            std::vector<uint8_t> o[4]; for (auto& v : o) v.resize(size[0] + 1024);
            for (int rep = 0; rep < 2; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                RangeEncoder r0, r1, r2, r3; r0.init(o[0].data(), (int)o[0].size()); r1.init(o[1].data(), (int)o[1].size()); r2.init(o[2].data(), (int)o[2].size()); r3.init(o[3].data(), (int)o[3].size());
                RangeEncoder::Live L0 = r0.enter(), L1 = r1.enter(), L2 = r2.enter(), L3 = r3.enter();
                const uint16_t* ps = pstream.data(); const size_t nd = pstream.size(), q = nd / 4;
                for (size_t i = 0; i < q; ++i) {
                    const unsigned a = ps[i], b = ps[q + i], c = ps[2 * q + i], d = ps[3 * q + i];
                    r0.encode_live<12>(L0, a >> 15, (int)(a & 0xfff));
                    r1.encode_live<12>(L1, b >> 15, (int)(b & 0xfff));
                    r2.encode_live<12>(L2, c >> 15, (int)(c & 0xfff));
                    r3.encode_live<12>(L3, d >> 15, (int)(d & 0xfff));
                }
                r0.leave(L0); r1.leave(L1); r2.leave(L2); r3.leave(L3);
                const int s0 = r0.finish() + r1.finish() + r2.finish() + r3.finish();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("rc, four interleaved streams: %d bytes, %.1f ms, %.2f ns/decision\n", s0, ms, ms * 1e6 / (4 * q));
            }
        }
        std::vector<uint8_t> out2(size[0] + 1024);
        auto t0 = std::chrono::steady_clock::now();
        const int sz2 = qlfc_encode_runs(R.view, size[0], out2.data(), size[0], CODER_STATIC);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("full static coder: %d bytes, %.1f ms\n", sz2, ms);
    }
    return 0;
}
