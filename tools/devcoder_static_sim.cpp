// devcoder_static_sim.cpp — CPU check of the stream-order evaluation of the static coder's context-free family
// (libbsc_amd/csrc/device/devcoder_static.h: descriptors, bit planes, phase A / resolve / phase C / values) against a plain
// sequential walk of the same chains.  Every lane function the HIP kernels call is run here lane by lane.  Not part of the product.
//   g++ -O2 -std=c++17 -march=native -I libbsc_amd/csrc/host -I include tools/devcoder_static_sim.cpp libbsc_amd/csrc/host/coder.cpp -o /tmp/devcoder_static_sim -lpthread
//   /tmp/devcoder_static_sim [file holding a BWT output]     (without a file: synthetic rank sequences only)
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include "../libbsc_amd/csrc/device/devcoder_static.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <string>

using namespace dcs;

static ModelParams MP;
static SpDesc DESC[(SP_MAXR + 1) * SP_SLOTS];         // by max_rank

struct Case { std::string name; std::vector<uint8_t> rank; SpSub S; };

static long check(const Case& C, bool verbose)
{
    const uint32_t m = (uint32_t)C.rank.size();
    const SpGeom g = sp_geom(m);
    const SpSub& S = C.S;
    for (int mr = 0; mr <= SP_MAXR; ++mr) if (!sp_build_descs(mr, DESC + mr * SP_SLOTS)) { printf("descriptors not representable for max_rank %d\n", mr); return 1; }
    // truth: sequential chains (sub-block, tau)
    std::vector<uint16_t> truth((size_t)m * 8, 0xffff);
    {
        for (uint32_t b = 0; b < S.nb; ++b) {
            int val[NUM_TAU];
            for (int t = 0; t < NUM_TAU; ++t) val[t] = MP.init[tau_class(t)];
            for (uint32_t j = S.first[b]; j < S.first[b + 1]; ++j)
                sp_rank_side(C.rank[j], (int)S.maxr[b], [&](int tau, uint32_t bit, int k) {
                    truth[(size_t)j * 8 + k] = (uint16_t)val[tau];
                    val[tau] = step(val[tau], bit, MP.rates[tau_class(tau)][FAM_STATIC]);
                });
        }
    }
    // planes
    std::vector<sp_u64> planes((size_t)g.ntiles * SP_PLANES, 0);
    for (uint32_t j = 0; j < m; ++j) for (int b = 0; b < SP_PLANES; ++b) if ((C.rank[j] >> b) & 1) planes[(size_t)(j / 64) * SP_PLANES + b] |= 1ull << (j % 64);
    // phase A
    std::vector<SpSum> sums((size_t)SP_SLOTS * g.cstride);
    long open_chunks = 0, open_big = 0, all_chunks = 0;
    for (int s = 0; s < SP_SLOTS - 1; ++s) for (uint32_t c = 0; c < sp_nchunks(g, s); ++c) {
        sums[(size_t)s * g.cstride + c] = sp_phase_a(s, c, g, S, planes.data(), DESC, sp_params(MP, s));
        const SpSum& x = sums[(size_t)s * g.cstride + c];
        ++all_chunks;
        if (x.lo != x.hi) { ++open_chunks; if (x.cnt > SP_HIST) ++open_big; }
    }
    if (verbose) {
        for (int s = 0; s < SP_SLOTS - 1; ++s) {
            long op = 0, big = 0, ev = 0, gap = 0;
            const uint32_t nch = sp_nchunks(g, s);
            for (uint32_t c = 0; c < nch; ++c) { const SpSum& x = sums[(size_t)s * g.cstride + c]; ev += x.cnt; if (x.lo != x.hi) { ++op; gap += x.hi - x.lo; if (x.cnt > SP_HIST) ++big; } }
            if (ev) printf("  slot %2d: %6u chunks of %2u tiles, events/chunk %7.1f, open chunks %6ld (%.1f %%), of them > %d events: %5ld, mean gap of the open ones %.2f\n", s, nch, sp_slot_ct(s), (double)ev / nch, op, 100.0 * op / nch, SP_HIST, big, op ? (double)gap / op : 0.0);
        }
    }
    // resolve: group summaries, group start values (serial per slot), chunk start values
    std::vector<SpGroupSum> gsum((size_t)SP_SLOTS * g.gstride);
    std::vector<uint16_t> Sv((size_t)SP_SLOTS * g.cstride, 0), Gv((size_t)SP_SLOTS * g.gstride, 0);
    long failed = 0, open_groups = 0, big_groups = 0;
    for (int s = 0; s < SP_SLOTS - 1; ++s) for (uint32_t gr = 0; gr < sp_ngroups(g, s); ++gr) {
        if (!sp_resolve_group(s, gr, g, S, planes.data(), DESC, sp_params(MP, s), sums.data() + (size_t)s * g.cstride + (size_t)gr * SP_RG, &gsum[(size_t)s * g.gstride + gr])) ++failed;
        else { const SpGroupSum& o = gsum[(size_t)s * g.gstride + gr]; if (!o.closed) { ++open_groups; if (o.big) ++big_groups; } }
    }
    for (int s = 0; s < SP_SLOTS - 1 && !failed; ++s) {
        int v = MP.init[sp_slot_class(s)];
        for (uint32_t gr = 0; gr < sp_ngroups(g, s); ++gr) {
            Gv[(size_t)s * g.gstride + gr] = (uint16_t)v;
            if (!sp_after_group(&v, gsum[(size_t)s * g.gstride + gr], s, gr, g, S, planes.data(), DESC, sp_params(MP, s), sums.data() + (size_t)s * g.cstride + (size_t)gr * SP_RG)) { ++failed; break; }
        }
    }
    for (int s = 0; s < SP_SLOTS - 1 && !failed; ++s) for (uint32_t gr = 0; gr < sp_ngroups(g, s); ++gr)
        if (!sp_resolve_chunks(s, gr, Gv[(size_t)s * g.gstride + gr], g, S, planes.data(), DESC, sp_params(MP, s), sums.data() + (size_t)s * g.cstride + (size_t)gr * SP_RG,
                               [&](int i, int val) { Sv[(size_t)s * g.cstride + (size_t)gr * SP_RG + i] = (uint16_t)val; })) ++failed;
    if (failed) { printf("%-28s m %9u: resolve gave up in %ld lanes (open chunks %ld of %ld, of them with > %d events %ld): the block would take the host model\n", C.name.c_str(), m, failed, open_chunks, all_chunks, SP_HIST, open_big); return 0; }
    // phase C
    std::vector<uint16_t> state((size_t)g.ntp * SP_LANES + 64, 0xeeee);
    for (int s = 0; s < SP_SLOTS - 1; ++s) for (uint32_t c = 0; c < sp_nchunks(g, s); ++c) sp_phase_c(s, c, g, S, planes.data(), DESC, sp_params(MP, s), Sv.data(), state.data());
    // values
    std::vector<uint16_t> rec((size_t)g.ntiles * 64 * 8, 0xffff);
    const SpParams P3[3] = {sp_params(MP, 0), sp_params(MP, 1), sp_params(MP, 5)};
    for (uint32_t t = 0; t < g.ntiles; ++t) for (int lane = 0; lane < 64; ++lane)
        sp_values(t, lane, g, S, planes.data(), DESC, P3, state.data(), [&](int i, int k, int v) { rec[((size_t)t * 64 + i) * 8 + k] = (uint16_t)v; });
    long bad = 0;
    for (uint32_t j = 0; j < m; ++j) for (int k = 0; k < 8; ++k)
        if (truth[(size_t)j * 8 + k] != rec[(size_t)j * 8 + k]) { if (bad++ < 5) printf("  %s: run %u k %d: %u vs truth %u (rank %u)\n", C.name.c_str(), j, k, rec[(size_t)j * 8 + k], truth[(size_t)j * 8 + k], C.rank[j]); }
    if (verbose || bad) printf("%-28s m %9u nb %u: open chunks %ld of %ld (with > %d events: %ld), open groups %ld (big %ld), mismatches %ld\n", C.name.c_str(), m, S.nb, open_chunks, all_chunks, SP_HIST, open_big, open_groups, big_groups, bad);
    return bad;
}

static SpSub make_sub(const std::vector<uint8_t>& rank, std::vector<uint32_t> cuts)
{
    SpSub S; memset(&S, 0, sizeof S);
    S.nb = (uint32_t)cuts.size();
    for (uint32_t b = 0; b < 9; ++b) S.first[b] = b < S.nb ? cuts[b] : (uint32_t)rank.size();
    for (uint32_t b = 0; b < S.nb; ++b) {
        uint32_t mx = 1;
        for (uint32_t j = S.first[b]; j < S.first[b + 1]; ++j) if (rank[j] > mx) mx = rank[j];
        S.maxr[b] = (uint32_t)bsr(mx);                         // the least max_rank that holds the ranks (the real one is bsr(nsym - 1) >= this)
    }
    return S;
}

namespace bschost { void* bigbuf_get(unsigned long n) { return malloc(n); } void bigbuf_put(void* p) { free(p); } }

int main(int argc, char** argv)
{
    model_params_from_table(bschost::qlfc_static_params(), MP);
    for (int mr = 0; mr <= SP_MAXR; ++mr) {
        SpDesc D[SP_SLOTS];
        if (!sp_build_descs(mr, D)) { printf("descriptors: max_rank %d not representable\n", mr); return 1; }
        // every rank value, every slot: the masks say what sp_rank_side says
        for (uint32_t r = 0; r < (2u << mr); ++r) {
            sp_u64 pl[SP_PLANES]; for (int b = 0; b < SP_PLANES; ++b) pl[b] = ((r >> b) & 1) ? ~0ull : 0ull;
            uint32_t seen = 0;
            sp_rank_side(r, mr, [&](int tau, uint32_t bit, int k) {
                const int s = sp_slot_of_tau(tau); seen |= 1u << s;
                const bool on = sp_match(pl, D[s].on_care, D[s].on_pat, D[s].on_inv) & 1, bt = sp_match(pl, D[s].bit_care, D[s].bit_pat, D[s].bit_inv) & 1;
                if (!on || bt != (bit != 0) || D[s].k != k || !D[s].present) { printf("descriptor mismatch max_rank %d rank %u slot %d\n", mr, r, s); exit(1); }
            });
            for (int s = 0; s < SP_SLOTS; ++s) if (!((seen >> s) & 1) && D[s].present && (sp_match(pl, D[s].on_care, D[s].on_pat, D[s].on_inv) & 1)) { printf("descriptor: slot %d matches rank %u (max_rank %d) but has no decision there\n", s, r, mr); exit(1); }
        }
    }
    int lanes = 0;
    for (int l = 0; l < 64; ++l) {
        int s, q, s2, q2; sp_lane_map(l, &s, &q); sp_lane_map_ref(l, &s2, &q2);
        if (s != s2 || q != q2) { printf("lane map: lane %d -> (%d, %d), by definition (%d, %d)\n", l, s, q, s2, q2); return 1; }
        if (s >= 0) ++lanes;
    }
    for (int s = 0; s < SP_SLOTS; ++s) if (sp_slot_first_lane(s) != sp_slot_first_lane_ref(s)) { printf("first lane of slot %d: %d, by definition %d\n", s, sp_slot_first_lane(s), sp_slot_first_lane_ref(s)); return 1; }
    if (lanes != SP_LANES) { printf("lane map: %d lanes, expected %d\n", lanes, SP_LANES); return 1; }
    printf("descriptors OK for max_rank 0..%d, %d lanes\n", SP_MAXR, lanes);

    std::mt19937_64 rng(12345);
    std::vector<Case> cases;
    auto add = [&](const std::string& name, std::vector<uint8_t> r, std::vector<uint32_t> cuts) { Case c; c.name = name; c.rank = std::move(r); c.S = make_sub(c.rank, cuts); cases.push_back(std::move(c)); };
    auto gen = [&](uint32_t m, int kind, uint32_t top) {
        std::vector<uint8_t> r(m);
        for (uint32_t j = 0; j < m; ++j) {
            uint32_t v;
            switch (kind) {
            case 0: v = 1 + rng() % top; break;                                          // uniform
            case 1: v = 1; break;                                                        // constant (brackets never close)
            case 2: { const uint32_t u = rng() % 100; v = u < 80 ? 1 : u < 95 ? 2 + rng() % 2 : 1 + rng() % top; } break;   // skewed
            case 3: v = (j / 5000) % 2 ? 1 : 1 + rng() % top; break;                     // long constant stretches
            case 4: v = (rng() % 20000 == 0) ? top : 1 + rng() % 3; break;               // one rare rank
            default: v = 1 + (j % top); break;
            }
            r[j] = (uint8_t)(v > top ? top : v);
        }
        return r;
    };
    for (uint32_t m : {1u, 2u, 63u, 64u, 65u, 127u, 2047u, 2048u, 2049u, 4097u, 100000u, 300001u}) {
        for (int kind = 0; kind < 6; ++kind) {
            const uint32_t top = kind == 0 ? 31 : kind == 2 ? 27 : 15;
            auto r = gen(m, kind, top);
            std::vector<uint32_t> cuts = {0};
            add("synthetic k" + std::to_string(kind) + " one sub-block", r, cuts);
            if (m >= 16) {
                const uint32_t nb = m >= 64 ? 8 : 2;
                std::vector<uint32_t> c2 = {0};
                for (uint32_t b = 1; b < nb; ++b) { uint32_t x = c2.back() + 1 + (uint32_t)(rng() % (2 * m / nb)); if (x >= m - (nb - b)) x = m - (nb - b); if (x <= c2.back()) x = c2.back() + 1; c2.push_back(x); }
                add("synthetic k" + std::to_string(kind) + " " + std::to_string(nb) + " sub-blocks", r, c2);
                if (m >= 4097) {       // several boundaries inside one tile, one on a tile's first lane, one on a chunk's first lane
                    std::vector<uint32_t> c3 = {0, 64, 70, 71, 100, 2048, 2048 + 63, 4096};
                    add("synthetic k" + std::to_string(kind) + " crowded boundaries", r, c3);
                }
            }
        }
    }
    // mixed alphabets: sub-blocks with different max_rank
    {
        std::vector<uint8_t> r;
        std::vector<uint32_t> cuts;
        for (int b = 0; b < 8; ++b) { cuts.push_back((uint32_t)r.size()); const uint32_t top = b == 0 ? 1 : b == 1 ? 3 : b == 2 ? 7 : b == 3 ? 2 : b == 4 ? 31 : b == 5 ? 15 : b == 6 ? 5 : 27; auto x = gen(20000 + 777 * b, b % 2 ? 0 : 2, top); r.insert(r.end(), x.begin(), x.end()); }
        add("mixed max_rank", r, cuts);
    }
    long total = 0;
    for (auto& c : cases) total += check(c, false);
    printf("%zu synthetic cases: %ld mismatches\n", cases.size(), total);

    if (argc > 1) {       // a real block: sub-block split, runs and ranks through the host's own front end (coder.cpp, qlfc.cpp)
        FILE* f = fopen(argv[1], "rb");
        if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
        fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> L((size_t)n);
        if (fread(L.data(), 1, (size_t)n, f) != (size_t)n) return 1;
        fclose(f);
        const int nb = bschost::coder_num_blocks((int)n);
        int start[8], size[8];
        bschost::coder_split_blocks(L.data(), (int)n, nb, start, size);
        Case c; c.name = argv[1]; memset(&c.S, 0, sizeof c.S); c.S.nb = (uint32_t)nb;
        bool fits = true;
        for (int sb = 0; sb < nb; ++sb) {
            bschost::QlfcRuns R; bschost::qlfc_runs(L.data() + start[sb], size[sb], R);
            const int max_rank = bschost::encode_alphabet(R.view, [](unsigned) {});
            c.S.first[sb] = (uint32_t)c.rank.size(); c.S.maxr[sb] = (uint32_t)max_rank;
            if (max_rank > SP_MAXR) fits = false;
            for (size_t j = 0; j < R.view.count; ++j) c.rank.push_back(R.view.rank[j]);
        }
        for (int b = nb; b < 9; ++b) c.S.first[b] = (uint32_t)c.rank.size();
        if (!fits) printf("%s: a sub-block has more than 32 symbols: the path is not offered for this block\n", argv[1]);
        else total += check(c, true);
    }
    printf("%s\n", total ? "FAILED" : "static stream-order evaluation OK");
    return total ? 1 : 0;
}
