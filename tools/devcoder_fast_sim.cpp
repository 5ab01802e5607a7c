// devcoder_fast_sim.cpp — CPU check of the fast coder (-e0) as the device model sees it (not part of the product).
//   g++ -O2 -std=c++17 -march=x86-64-v3 -I libbsc_amd/csrc/host -I libbsc_amd/csrc/device -I include tools/devcoder_fast_sim.cpp -o /tmp/fast_sim && /tmp/fast_sim
// The device runs the fast coder as chains (sub-block, decision type, symbol) with the update maps of dcm::model_params_fast and hands
// the host 16-bit entries (devcoder_model.h PSF_*).  Here the same chains are walked serially: decisions by dcm::enumerate with
// max_rank 7, counters in a table indexed by (type, symbol) starting from ModelParams::init, dcm::step as the update, the entry
// stream coded by qlfc_encode_fast_pstream (and the pair coder) — and the bytes must be those of the host's own fast coder
// (encode_model2, which the CPU tests pin to the reference).  Also checks nth_decision against enumerate at max_rank 7.
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include "../libbsc_amd/csrc/device/devcoder_model.h"
#include <cstdio>
#include <random>
#include <vector>
using namespace bschost;

static std::vector<uint16_t> chain_stream(const RunView& R, const dcm::ModelParams& M, int* bad)
{
    std::vector<short> ctr((size_t)dcm::NUM_TAU * 256);
    for (int tau = 0; tau < dcm::NUM_TAU; ++tau) for (int c = 0; c < 256; ++c) ctr[(size_t)tau * 256 + c] = M.init[dcm::tau_class(tau)];
    std::vector<uint16_t> ps;
    for (uint32_t j = 0; j < R.count; ++j) {
        dcm::Item it; it.sb = 0; it.ge32 = 0; it.rank = R.rank[j]; it.run = R.len(j);
        const int c = R.sym[j];
        const int n_rank = dcm::count_rank_side(it, 7), n_run = dcm::count_run_side(it);
        int k = 0;
        dcm::enumerate(it, 7, [&](int tau, uint32_t bit, bool run_side) {
            uint32_t b2; bool rs2;
            const int t2 = dcm::nth_decision(it, 7, n_rank, k, &b2, &rs2);
            if (t2 != tau || b2 != bit || rs2 != run_side) { if ((*bad)++ < 5) printf("nth_decision mismatch run %u k %d\n", j, k); }
            short& v = ctr[(size_t)tau * 256 + c];
            const int cls = dcm::tau_class(tau);
            if (v < M.vmin[cls][dcm::FAM_CHAR] || v > M.vmax[cls][dcm::FAM_CHAR]) { if ((*bad)++ < 5) printf("value %d outside [%d, %d] class %d\n", v, M.vmin[cls][1], M.vmax[cls][1], cls); }
            ps.push_back((uint16_t)((unsigned)v | (bit << 13) | (k == 0 ? 0x4000u : 0u) | (run_side ? 0x8000u : 0u)));
            v = (short)dcm::step(v, bit, M.rates[cls][dcm::FAM_CHAR]);
            ++k;
        });
        if (k != n_rank + n_run) { if ((*bad)++ < 5) printf("decision count mismatch run %u\n", j); }
    }
    return ps;
}

int main()
{
    dcm::ModelParams M; dcm::model_params_fast(M);
    std::mt19937_64 rng(3);
    int bad = 0, cases = 0;
    std::vector<std::vector<uint8_t>> inputs;
    auto gen = [&](size_t n, int kind) {
        std::vector<uint8_t> v(n);
        if (kind == 0) for (auto& x : v) x = (uint8_t)(rng() & 255);                                          // every rank, short runs
        else if (kind == 1) { size_t i = 0; while (i < n) { const uint8_t c = (uint8_t)(rng() % 5); size_t l = 1 + (rng() % 3 == 0 ? rng() % 40000 : rng() % 4); while (l-- && i < n) v[i++] = c; } }   // long runs (chains of > 5 bits)
        else if (kind == 2) { uint8_t c = 0; for (auto& x : v) { if (rng() % 3 == 0) c = (uint8_t)(97 + (rng() % 26 < 20 ? rng() % 6 : rng() % 26)); x = c; } }      // text-like BWT output
        else { for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)((rng() % 100 < 97) ? 0 : rng() % 200); }        // one dominant symbol
        return v;
    };
    for (int kind = 0; kind < 4; ++kind) for (size_t n : {1000u, 70000u, 600000u}) inputs.push_back(gen(n, kind));
    inputs.push_back(std::vector<uint8_t>(300000, 7));                                                          // a single run
    for (auto& in : inputs) {
        QlfcRuns R; qlfc_runs(in.data(), (int)in.size(), R);
        std::vector<uint8_t> want(in.size() * 2 + 4096), got(in.size() * 2 + 4096), got2(in.size() * 2 + 4096);
        for (int budget_mode = 0; budget_mode < 2; ++budget_mode) {
            const int osz = budget_mode ? (int)in.size() : (int)want.size() - 64;                               // the format's budget: out_size = in_size
            const int rw = qlfc_encode_runs(R.view, (int)in.size(), want.data(), osz, CODER_FAST);
            const std::vector<uint16_t> ps = chain_stream(R.view, M, &bad);
            const int rg = qlfc_encode_fast_pstream(R.view.first_seen, R.view.nsym, (int)in.size(), ps.data(), ps.size(), got.data(), osz);
            PstreamJob A{R.view.first_seen, R.view.nsym, (int)in.size(), ps.data(), ps.size(), got2.data(), osz};
            PstreamJob B{R.view.first_seen, R.view.nsym, (int)in.size(), ps.data(), ps.size() / 2, got.data() + 0, osz};   // (B's output is not looked at)
            std::vector<uint8_t> scratchB(want.size()); B.out = scratchB.data();
            int r2 = 0, r3 = 0;
            qlfc_encode_fast_pstream_pair(A, B, &r2, &r3);
            ++cases;
            if (rw != rg || (rw > 0 && memcmp(want.data(), got.data(), (size_t)rw) != 0)) { ++bad; printf("MISMATCH n %zu budget %d: host %d chains %d\n", in.size(), budget_mode, rw, rg); }
            if (rw != r2 || (rw > 0 && memcmp(want.data(), got2.data(), (size_t)rw) != 0)) { ++bad; printf("PAIR MISMATCH n %zu budget %d: host %d pair %d\n", in.size(), budget_mode, rw, r2); }
        }
    }
    {   // eight streams in SIMD lanes (qlfc_encode_fast_pstream_x8) against eight single coders; then with one output too small: it must give up
        std::vector<std::vector<uint16_t>> ps(8); std::vector<QlfcRuns> R(8);
        const size_t pick[8] = {2, 5, 8, 11, 1, 4, 7, 10};
        for (int l = 0; l < 8; ++l) { qlfc_runs(inputs[pick[l]].data(), (int)inputs[pick[l]].size(), R[l]); ps[l] = chain_stream(R[l].view, M, &bad); }
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<std::vector<uint8_t>> oa(8), ob(8); PstreamJob J[8]; int ra[8], rb[8];
            for (int l = 0; l < 8; ++l) {
                const size_t n = inputs[pick[l]].size();
                const int osz = (mode == 1 && l == 0) ? 4096 : (int)n * 2 + 4096;      // lane 0 = 600 000 random bytes: far beyond 4 KB
                oa[l].assign((size_t)osz + 64, 0); ob[l].assign((size_t)osz + 64, 0);
                ra[l] = qlfc_encode_fast_pstream(R[l].view.first_seen, R[l].view.nsym, (int)n, ps[l].data(), ps[l].size(), oa[l].data(), osz);
                J[l] = PstreamJob{R[l].view.first_seen, R[l].view.nsym, (int)n, ps[l].data(), ps[l].size(), ob[l].data(), osz};
            }
            const bool ok = qlfc_encode_fast_pstream_x8(J, rb);
            ++cases;
            if (mode == 1) { if (ra[0] >= 0 || (ok && rb[0] >= 0)) { ++bad; printf("x8: a stream past its budget was not noticed (%d, %d)\n", ra[0], rb[0]); } else printf("x8 budget case: %s\n", ok ? "coded, lane 0 NOT_COMPRESSIBLE" : "gave up"); continue; }
            if (!ok) { ++bad; printf("x8 gave up with roomy outputs\n"); continue; }
            for (int l = 0; l < 8; ++l) if (ra[l] != rb[l] || memcmp(oa[l].data(), ob[l].data(), (size_t)ra[l]) != 0) { ++bad; printf("X8 MISMATCH lane %d: %d vs %d\n", l, ra[l], rb[l]); }
        }
    }
    printf("%d cases, %d problems%s\n", cases, bad, bad ? "" : ": all equal");
    return bad != 0;
}
