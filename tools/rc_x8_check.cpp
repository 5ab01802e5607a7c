// rc_x8_check.cpp — CPU check + timing of the 8-lane AVX2 range coder against the scalar one (not part of the product).
//   g++ -O2 -std=c++17 -march=x86-64-v3 -I libbsc_amd/csrc/host -I include tools/rc_x8_check.cpp -o /tmp/rc_x8_check && /tmp/rc_x8_check
#include "../libbsc_amd/csrc/host/qlfc.cpp"
#include <chrono>
#include <cstdio>
#include <random>
using namespace bschost;

int main(int argc, char** argv)
{
    const size_t base = argc > 1 ? (size_t)atol(argv[1]) : 4000000;
    std::mt19937_64 rng(7);
    std::vector<uint16_t> ps[8];
    uint8_t first_seen[8][256]; int nsym[8];
    for (int l = 0; l < 8; ++l) {
        const size_t cnt = base + (rng() % (base / 20 + 1)) - (l == 3 ? base / 50 : 0);
        ps[l].resize(cnt);
        for (size_t i = 0; i < cnt; ++i) {
            // skewed probabilities like the model's: mostly confident predictions
            unsigned p = 1 + (unsigned)(rng() % 4095);
            if (rng() & 3) p = (rng() & 1) ? 1 + p / 16 : 4095 - p / 16;
            const unsigned bit = ((rng() % 4096) >= p) ? 1u : 0u;        // P(bit = 0) = p / 4096
            ps[l][i] = (uint16_t)(p | (bit << 12) | ((i % 7 == 0) ? 0x2000u : 0u));
        }
        nsym[l] = 20 + l;
        for (int s = 0; s < nsym[l]; ++s) first_seen[l][s] = (uint8_t)(s * 3 + l);
    }
    int bad = 0;
    for (int mode = 0; mode < 2; ++mode) {          // 0: roomy outputs, 1: lane 5 too small (must give up)
        std::vector<uint8_t> oa[8], ob[8];
        PstreamJob J[8];
        int ra[8], rb[8];
        for (int l = 0; l < 8; ++l) {
            const int osz = (mode == 1 && l == 5) ? 4096 : (int)ps[l].size() * 2 + 1024;
            oa[l].assign(osz + 64, 0); ob[l].assign(osz + 64, 0);
            J[l] = PstreamJob{first_seen[l], nsym[l], (int)ps[l].size(), ps[l].data(), ps[l].size(), ob[l].data(), osz};
            ra[l] = qlfc_encode_static_pstream(first_seen[l], nsym[l], (int)ps[l].size(), ps[l].data(), ps[l].size(), oa[l].data(), osz);
        }
        auto t0 = std::chrono::steady_clock::now();
        const bool ok = qlfc_encode_static_pstream_x8(J, rb);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        size_t total = 0; for (int l = 0; l < 8; ++l) total += ps[l].size();
        printf("mode %d: x8 %s, %.1f ms, %.3f ns/decision\n", mode, ok ? "done" : "gave up", ms, ms * 1e6 / total);
        if (mode == 0) {
            if (!ok) { printf("FAIL: x8 gave up with roomy outputs\n"); ++bad; }
            else for (int l = 0; l < 8; ++l) {
                if (ra[l] != rb[l] || memcmp(oa[l].data(), ob[l].data(), (size_t)(ra[l] > 0 ? ra[l] : 0)) != 0) { printf("FAIL lane %d: scalar %d bytes, x8 %d bytes\n", l, ra[l], rb[l]); ++bad; }
            }
        } else {
            if (ok && rb[5] != ra[5]) { printf("FAIL: budget case, scalar %d x8 %d\n", ra[5], rb[5]); ++bad; }
            printf("  scalar result of the small lane: %d\n", ra[5]);
        }
    }
    // scalar pair timing for comparison
    {
        std::vector<uint8_t> o0(ps[0].size() * 2 + 1024), o1(ps[1].size() * 2 + 1024);
        PstreamJob A{first_seen[0], nsym[0], (int)ps[0].size(), ps[0].data(), ps[0].size(), o0.data(), (int)o0.size() - 64};
        PstreamJob B{first_seen[1], nsym[1], (int)ps[1].size(), ps[1].data(), ps[1].size(), o1.data(), (int)o1.size() - 64};
        int r0, r1;
        auto t0 = std::chrono::steady_clock::now();
        qlfc_encode_static_pstream_pair(A, B, &r0, &r1);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("scalar pair: %.1f ms, %.3f ns/decision\n", ms, ms * 1e6 / (ps[0].size() + ps[1].size()));
    }
    // the fast coder's entries (13-bit value, bit at 13, run start at 14, bit 15 = run side: 11-bit precision): eight lanes against the scalar coder
    {
        std::vector<uint16_t> fs[8];
        for (int l = 0; l < 8; ++l) {
            const size_t cnt = base + (rng() % (base / 20 + 1));
            fs[l].resize(cnt);
            for (size_t i = 0; i < cnt; ++i) {
                const unsigned side = (unsigned)(rng() & 1), sh = 13u - 2u * side;
                unsigned pv = 1 + (unsigned)(rng() % ((1u << sh) - 1));
                if (rng() & 3) pv = (rng() & 1) ? 1 + pv / 16 : ((1u << sh) - 1) - pv / 16;
                const unsigned bit = ((rng() % (1u << sh)) >= pv) ? 1u : 0u;
                fs[l][i] = (uint16_t)(pv | (bit << 13) | ((rng() % 3 == 0) ? 0x4000u : 0u) | (side << 15));
            }
        }
        std::vector<uint8_t> oa[8], ob[8];
        PstreamJob J[8]; int ra[8], rb[8];
        for (int l = 0; l < 8; ++l) {
            const int osz = (int)fs[l].size() * 2 + 1024;
            oa[l].assign(osz + 64, 0); ob[l].assign(osz + 64, 0);
            J[l] = PstreamJob{first_seen[l], nsym[l], (int)fs[l].size(), fs[l].data(), fs[l].size(), ob[l].data(), osz};
            ra[l] = qlfc_encode_fast_pstream(first_seen[l], nsym[l], (int)fs[l].size(), fs[l].data(), fs[l].size(), oa[l].data(), osz);
        }
        if (!qlfc_encode_fast_pstream_x8(J, rb)) { printf("FAIL: fast x8 gave up with roomy outputs\n"); ++bad; }
        else for (int l = 0; l < 8; ++l)
            if (ra[l] != rb[l] || memcmp(oa[l].data(), ob[l].data(), (size_t)(ra[l] > 0 ? ra[l] : 0)) != 0) { printf("FAIL fast lane %d: scalar %d bytes, x8 %d bytes\n", l, ra[l], rb[l]); ++bad; }
        printf("fast coder, eight lanes: %s\n", bad ? "differs" : "equal");
    }
    // the packed stream (13 bits per decision, eight decisions in 13 bytes; round 6): single, pair and eight-lane coders on the packed
    // form of the static streams above against the scalar coder on the 16-bit entries — same bytes; and the give-up rule (no run-start
    // mark: a stream whose budget is reached anywhere reports NOT_COMPRESSIBLE / the eight-lane coder gives up)
    {
        int pbad = 0;
        std::vector<uint8_t> pk[8];
        for (int l = 0; l < 8; ++l) { pk[l].assign((ps[l].size() + 7) / 8 * 13 + 64, 0xa5); qlfc_pack_p13(ps[l].data(), ps[l].size(), pk[l].data()); }
        std::vector<uint8_t> oa[8], ob[8], oc[8];
        PstreamJob J[8]; int ra[8], rb[8], rc1[8];
        for (int l = 0; l < 8; ++l) {
            const int osz = (int)ps[l].size() * 2 + 1024;
            oa[l].assign(osz + 64, 0); ob[l].assign(osz + 64, 0); oc[l].assign(osz + 64, 0);
            J[l] = PstreamJob{first_seen[l], nsym[l], (int)ps[l].size(), reinterpret_cast<const uint16_t*>(pk[l].data()), ps[l].size(), ob[l].data(), osz};
            ra[l] = qlfc_encode_static_pstream(first_seen[l], nsym[l], (int)ps[l].size(), ps[l].data(), ps[l].size(), oa[l].data(), osz);
            rc1[l] = qlfc_encode_static_p13(first_seen[l], nsym[l], (int)ps[l].size(), pk[l].data(), ps[l].size(), oc[l].data(), osz);
            if (ra[l] != rc1[l] || memcmp(oa[l].data(), oc[l].data(), (size_t)(ra[l] > 0 ? ra[l] : 0)) != 0) { printf("FAIL packed single, stream %d: %d against %d bytes\n", l, rc1[l], ra[l]); ++pbad; }
        }
        auto t0 = std::chrono::steady_clock::now();
        const bool ok = qlfc_encode_static_p13_x8(J, rb);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        size_t total = 0; for (int l = 0; l < 8; ++l) total += ps[l].size();
        printf("packed stream, eight lanes: %s, %.1f ms, %.3f ns/decision\n", ok ? "done" : "gave up", ms, ms * 1e6 / total);
        if (!ok) { printf("FAIL: packed x8 gave up with roomy outputs\n"); ++pbad; }
        else for (int l = 0; l < 8; ++l)
            if (ra[l] != rb[l] || memcmp(oa[l].data(), ob[l].data(), (size_t)(ra[l] > 0 ? ra[l] : 0)) != 0) { printf("FAIL packed x8 lane %d: scalar %d bytes, x8 %d bytes\n", l, ra[l], rb[l]); ++pbad; }
        for (int a = 0; a < 8; a += 2) {
            std::vector<uint8_t> o0(oa[a].size(), 0), o1(oa[a + 1].size(), 0);
            PstreamJob A = J[a], B = J[a + 1]; A.out = o0.data(); B.out = o1.data();
            int r0, r1;
            qlfc_encode_static_p13_pair(A, B, &r0, &r1);
            if (r0 != ra[a] || r1 != ra[a + 1] || memcmp(o0.data(), oa[a].data(), (size_t)(r0 > 0 ? r0 : 0)) != 0 || memcmp(o1.data(), oa[a + 1].data(), (size_t)(r1 > 0 ? r1 : 0)) != 0) { printf("FAIL packed pair %d\n", a); ++pbad; }
        }
        // budget: stream 5 into 4096 bytes
        {
            std::vector<uint8_t> small(4096 + 64, 0);
            const int r = qlfc_encode_static_p13(first_seen[5], nsym[5], (int)ps[5].size(), pk[5].data(), ps[5].size(), small.data(), 4096);
            const int rs = qlfc_encode_static_pstream(first_seen[5], nsym[5], (int)ps[5].size(), ps[5].data(), ps[5].size(), oa[5].data(), 4096);
            if (ps[5].size() > 40000 && r >= 0) { printf("FAIL: packed single coded %zu decisions into 4096 bytes (%d)\n", ps[5].size(), r); ++pbad; }
            if (rs >= 0 && r != rs) { printf("FAIL: packed single %d where the scalar coder fits (%d)\n", r, rs); ++pbad; }
            PstreamJob K[8]; for (int l = 0; l < 8; ++l) K[l] = J[l];
            K[5].out = small.data(); K[5].out_size = 4096;
            int rr[8];
            const bool ok2 = qlfc_encode_static_p13_x8(K, rr);
            if (ps[5].size() > 40000 && ok2 && rr[5] >= 0) { printf("FAIL: packed x8 did not give up on the small lane\n"); ++pbad; }
        }
        printf("packed stream: %s\n", pbad ? "differs" : "equal");
        bad += pbad;
    }
    // sixteen lanes (two blocks per task, 512-bit registers): static and fast entries against the scalar coders
    if (qlfc_x16_available()) {
        for (int fast = 0; fast < 2; ++fast) {
            std::vector<uint16_t> st[16];
            uint8_t fs16[16][256]; int ns16[16];
            for (int l = 0; l < 16; ++l) {
                const size_t cnt = base + (rng() % (base / 20 + 1)) - (l == 11 ? base / 40 : 0);
                st[l].resize(cnt);
                for (size_t i = 0; i < cnt; ++i) {
                    if (!fast) {
                        unsigned p = 1 + (unsigned)(rng() % 4095);
                        if (rng() & 3) p = (rng() & 1) ? 1 + p / 16 : 4095 - p / 16;
                        const unsigned bit = ((rng() % 4096) >= p) ? 1u : 0u;
                        st[l][i] = (uint16_t)(p | (bit << 12) | ((i % 7 == 0) ? 0x2000u : 0u));
                    } else {
                        const unsigned side = (unsigned)(rng() & 1), sh = 13u - 2u * side;
                        unsigned pv = 1 + (unsigned)(rng() % ((1u << sh) - 1));
                        if (rng() & 3) pv = (rng() & 1) ? 1 + pv / 16 : ((1u << sh) - 1) - pv / 16;
                        const unsigned bit = ((rng() % (1u << sh)) >= pv) ? 1u : 0u;
                        st[l][i] = (uint16_t)(pv | (bit << 13) | ((rng() % 3 == 0) ? 0x4000u : 0u) | (side << 15));
                    }
                }
                ns16[l] = 18 + l;
                for (int q = 0; q < ns16[l]; ++q) fs16[l][q] = (uint8_t)(q * 5 + l);
            }
            for (int mode = 0; mode < 2; ++mode) {                     // 0: roomy outputs, 1: lane 13 too small (must give up)
                std::vector<uint8_t> oa[16], ob[16];
                PstreamJob J[16]; int ra[16], rb[16];
                for (int l = 0; l < 16; ++l) {
                    const int osz = (mode == 1 && l == 13) ? 4096 : (int)st[l].size() * 2 + 1024;
                    oa[l].assign(osz + 64, 0); ob[l].assign(osz + 64, 0);
                    J[l] = PstreamJob{fs16[l], ns16[l], (int)st[l].size(), st[l].data(), st[l].size(), ob[l].data(), osz};
                    ra[l] = fast ? qlfc_encode_fast_pstream(fs16[l], ns16[l], (int)st[l].size(), st[l].data(), st[l].size(), oa[l].data(), osz)
                                 : qlfc_encode_static_pstream(fs16[l], ns16[l], (int)st[l].size(), st[l].data(), st[l].size(), oa[l].data(), osz);
                }
                auto t0 = std::chrono::steady_clock::now();
                const bool ok = fast ? qlfc_encode_fast_pstream_x16(J, rb) : qlfc_encode_static_pstream_x16(J, rb);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                size_t total = 0; for (int l = 0; l < 16; ++l) total += st[l].size();
                printf("%s coder, sixteen lanes, mode %d: %s, %.1f ms, %.3f ns/decision\n", fast ? "fast" : "static", mode, ok ? "done" : "gave up", ms, ms * 1e6 / total);
                if (mode == 0) {
                    if (!ok) { printf("FAIL: x16 gave up with roomy outputs\n"); ++bad; }
                    else for (int l = 0; l < 16; ++l)
                        if (ra[l] != rb[l] || memcmp(oa[l].data(), ob[l].data(), (size_t)(ra[l] > 0 ? ra[l] : 0)) != 0) { printf("FAIL x16 lane %d: scalar %d bytes, x16 %d bytes\n", l, ra[l], rb[l]); ++bad; }
                } else if (ok && rb[13] != ra[13]) { printf("FAIL: x16 budget case, scalar %d x16 %d\n", ra[13], rb[13]); ++bad; }
            }
        }
        printf("sixteen lanes: %s\n", bad ? "differs" : "equal");
    } else printf("sixteen lanes: not available on this CPU (needs AVX-512F/VL/BW)\n");
    printf(bad ? "FAILED\n" : "all equal\n");
    return bad != 0;
}
