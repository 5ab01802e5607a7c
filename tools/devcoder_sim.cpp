// devcoder_sim.cpp — CPU check of devcoder_model.h against the host coder (qlfc.cpp): decision enumeration by canonical
// rounds, chain identity (tau, X) <-> the reference's counter slots, and the probability stream.  Not part of the product.
//   g++ -O2 -std=c++17 -march=native -I libbsc_amd/csrc/host -I include tools/devcoder_sim.cpp libbsc_amd/csrc/host/coder.cpp -o /tmp/devcoder_sim -lpthread
//   /tmp/devcoder_sim /tmp/bwt_64m_s2.bin [max sub-blocks]
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include "../libbsc_amd/csrc/device/devcoder_model.h"
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

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> L(n);
    if (fread(L.data(), 1, n, f) != (size_t)n) return 1;
    fclose(f);
    const int maxsb = argc > 2 ? atoi(argv[2]) : 8;
    const int nb = coder_num_blocks((int)n);
    int start[8], size[8];
    coder_split_blocks(L.data(), (int)n, nb, start, size);
    const QlfcTables& T = qlfc_tables();
    dcm::ModelParams M; dcm::model_params_from_table(kStaticParams, M);
    for (int c = 0; c < 7; ++c) printf("class %d: S[%d,%d] C[%d,%d] P[%d,%d]\n", c, M.vmin[c][0], M.vmax[c][0], M.vmin[c][1], M.vmax[c][1], M.vmin[c][2], M.vmax[c][2]);
    // tau_round consistency
    for (int tau = 0; tau < dcm::NUM_TAU; ++tau) { const int r = dcm::tau_round(tau); if (r < 0 || r >= dcm::NUM_ROUNDS) { printf("bad round for tau %d\n", tau); return 1; } }

    long total_bad = 0;
    for (int sb = 0; sb < nb && sb < maxsb; ++sb) {
        QlfcRuns R; qlfc_runs(L.data() + start[sb], size[sb], R);
        std::unique_ptr<Counters1> K(new_counters());
        std::vector<Dec> D; std::vector<uint32_t> rf;
        const int max_rank = encode_alphabet(R.view, [](unsigned) {});
        LogPolicy pol{K.get(), &D, &rf};
        walk_model1<false>(R.view, T, max_rank, *K, nullptr, pol);
        rf.push_back((uint32_t)D.size());
        const size_t m = R.view.count;
        // truth p stream
        std::vector<uint16_t> truth(D.size());
        {
            std::unique_ptr<Counters1> K2(new_counters());
            short* b = reinterpret_cast<short*>(K2.get());
            for (size_t i = 0; i < D.size(); ++i) {
                const Dec& d = D[i]; const short* P = kStaticParams[d.cls];
                truth[i] = (uint16_t)((b[d.ch] * P[16] + b[d.st] * P[17] + b[d.sp] * P[18]) >> 5);
                bump(b[d.st], d.bit, P[0], P[1], P[2], P[3]); bump(b[d.ch], d.bit, P[4], P[5], P[6], P[7]); bump(b[d.sp], d.bit, P[8], P[9], P[10], P[11]);
            }
        }
        // model: contexts, decisions by rounds, chains keyed by (tau, X)
        uint32_t ctx_rank0 = 0, ctx_rank4 = 0, ctx_run = 0, avg = 0;
        uint8_t rank_hist[256] = {0}, run_hist[256] = {0};
        std::unordered_map<uint64_t, int> chain[3];
        std::map<uint32_t, uint32_t> slot_of[3];              // chain key -> reference slot (bijection check)
        std::map<uint32_t, uint32_t> key_of[3];
        long bad = 0; size_t di = 0;
        for (size_t j = 0; j < m; ++j) {
            const uint32_t c = R.view.sym[j], rank = R.view.rank[j], run = R.view.len((uint32_t)j);
            dcm::Item it{rank, run, (uint32_t)sb, avg >= 32 ? 1u : 0u};
            const uint32_t state_rank = T.rank_state[dcm::rank_state_index(ctx_run, ctx_rank4, rank_hist[c])];
            const uint32_t state_run = T.run_state[dcm::run_state_index(ctx_rank0, ctx_run, rank, run_hist[c])];
            int cnt = 0;
            for (int r = 0; r < dcm::NUM_ROUNDS; ++r) {
                uint32_t bit = 0;
                const int tau = dcm::decision(it, max_rank, r, &bit);
                if (tau < 0) continue;
                if (dcm::tau_round(tau) != r) { if (bad++ < 10) printf("tau_round mismatch tau %d r %d\n", tau, r); }
                if (di >= D.size() || di >= rf[j + 1]) { if (bad++ < 10) printf("sb %d run %zu: too many decisions\n", sb, j); break; }
                const Dec& d = D[di];
                const int cls = dcm::tau_class(tau);
                if ((cls == dcm::CLS_NM2 ? (int)dcm::CLS_NM : cls) != d.cls || bit != d.bit) { if (bad++ < 10) printf("sb %d run %zu dec %d: cls/bit %d/%u vs truth %d/%d (tau %d rank %u run %u)\n", sb, j, cnt, cls, bit, d.cls, d.bit, tau, rank, run); }
                const uint32_t X[3] = {r < dcm::ROUND_NF ? state_rank : state_run, c, 0u};
                const uint32_t refslot[3] = {d.st, d.ch, d.sp};
                int v[3];
                for (int fam = 0; fam < 3; ++fam) {
                    const uint32_t key = (uint32_t)tau * 256u + X[fam];
                    auto a = slot_of[fam].find(key);
                    if (a == slot_of[fam].end()) slot_of[fam][key] = refslot[fam]; else if (a->second != refslot[fam]) { if (bad++ < 10) printf("fam %d key->slot not a function (tau %d X %u)\n", fam, tau, X[fam]); }
                    auto b2 = key_of[fam].find(refslot[fam]);
                    if (b2 == key_of[fam].end()) key_of[fam][refslot[fam]] = key; else if (b2->second != key) { if (bad++ < 10) printf("fam %d slot->key not a function (tau %d X %u slot %u)\n", fam, tau, X[fam], refslot[fam]); }
                    auto itc = chain[fam].find(key);
                    if (itc == chain[fam].end()) itc = chain[fam].emplace(key, 2048).first;
                    v[fam] = itc->second;
                    itc->second = dcm::step(v[fam], bit, M.rates[cls][fam]);
                    if (itc->second < M.vmin[cls][fam] || itc->second > M.vmax[cls][fam]) { if (bad++ < 10) printf("value outside the attainable range\n"); }
                }
                const int p = dcm::blend(v[dcm::FAM_CHAR], v[dcm::FAM_STATE], v[dcm::FAM_STATIC], M.lr[cls]);
                if (p != truth[di]) { if (bad++ < 10) printf("sb %d run %zu dec %d: p %d vs %d\n", sb, j, cnt, p, truth[di]); }
                ++di; ++cnt;
            }
            if (di != rf[j + 1]) { if (bad++ < 10) printf("sb %d run %zu: %d decisions vs truth %u\n", sb, j, cnt, rf[j + 1] - rf[j]); di = rf[j + 1]; }
            if (cnt != dcm::count_rank_side(it, max_rank) + dcm::count_run_side(it)) { if (bad++ < 10) printf("count mismatch\n"); }
            {   // nth_decision and enumerate agree with the round order
                const int n_rank = dcm::count_rank_side(it, max_rank);
                int k = 0;
                dcm::enumerate(it, max_rank, [&](int tau, uint32_t bit, bool rs) {
                    uint32_t b2 = 9; bool rs2 = !rs;
                    const int t2 = dcm::nth_decision(it, max_rank, n_rank, k, &b2, &rs2);
                    const Dec& d = D[rf[j] + k];
                    if (t2 != tau || b2 != bit || rs2 != rs || (dcm::tau_class(tau) == dcm::CLS_NM2 ? (int)dcm::CLS_NM : dcm::tau_class(tau)) != d.cls || bit != d.bit) { if (bad++ < 10) printf("nth/enumerate mismatch run %zu k %d\n", j, k); }
                    uint32_t b3 = 9; bool rs3 = !rs;                                        // the class-only form the p-stream kernel uses
                    const int c3 = dcm::nth_class(it, max_rank, n_rank, k, &b3, &rs3);
                    if (c3 != dcm::tau_class(tau) || b3 != bit || rs3 != rs) { if (bad++ < 10) printf("nth_class mismatch run %zu k %d: class %d vs %d\n", j, k, c3, dcm::tau_class(tau)); }
                    ++k;
                });
                if (k != cnt) { if (bad++ < 10) printf("enumerate count mismatch\n"); }
            }
            // context update (qlfc.cpp:978-989, :1063-1068)
            rank_hist[c] = (uint8_t)dcm::bsr(rank);
            avg = dcm::avg_rank_next(avg, rank);
            run_hist[c] = (uint8_t)dcm::run_hist_next(run_hist[c], run);
            ctx_rank0 = ((ctx_rank0 << 1) | (rank == 1 ? 1u : 0u)) & 7u;
            ctx_rank4 = ((ctx_rank4 << 2) | (rank - 1 < 3 ? rank - 1 : 3u)) & 0xffu;
            ctx_run = ((ctx_run << 1) | (run < 3 ? 1u : 0u)) & 0xfu;
        }
        printf("sub-block %d: runs %zu decisions %zu max_rank %d chains S/C/P %zu/%zu/%zu mismatches %ld\n", sb, m, D.size(), max_rank, chain[0].size(), chain[1].size(), chain[2].size(), bad);
        total_bad += bad;
    }
    printf("%s\n", total_bad ? "FAILED" : "model OK");
    return total_bad ? 1 : 0;
}
