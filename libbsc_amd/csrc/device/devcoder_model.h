// devcoder_model.h — the static QLFC model (-e1) restated as data-parallel pieces, shared by the HIP kernels
// (devcoder.hip) and by host code (host range coder, CPU checks).  Plain integer functions, no HIP dependency.
//
// What the reference does per run, serially (libbsc/coder/qlfc/qlfc.cpp:896-1126, model slots qlfc_model.h:38-176):
// a short list of binary decisions, each predicted by THREE adaptive counters — one indexed by a context state, one by
// the run's symbol, one by nothing but the decision's place in the code tree — whose values are blended into a 12-bit
// probability and then updated with the coded bit (predictor.h:53-61).  Which three counters a decision touches is a pure
// function of the run data; each counter only ever sees its own decisions.  So the model is a set of independent CHAINS,
//      chain = (sub-block, decision type tau, family in {state, char, static}, X = state | symbol | nothing),
// every chain the recurrence v <- step(v, bit) over its decisions in stream order, starting from 2048.
//
// Decision types ("tau", 1080 of them) enumerate the code tree positions:
//   RF                 rank == 1 ?                                   1        class 0  (qlfc.cpp:904)
//   RE s               unary exponent of the rank, position s        7        class 1  (:919-:935)
//   RM (B, ctx)        mantissa of a B-bit rank, tree node ctx       247      class 2  (:938-:953)
//   RP ctx             escape coding of the rank (avg_rank >= 32)    255      class 3  (:960-:975)
//   NF                 run length == 1 ?                             1        class 4  (:990)
//   NE s               unary exponent of the run length              31       class 5  (:1003-:1019)
//   NM (bits, ctx)     mantissa of the run length                    538      class 6  (:1022-:1060)
// Inside one run every type occurs at most once, always in the order of the "canonical rounds" below, which is what lets a
// wavefront emit the decisions of 64 runs round by round and rank them stably with ballots.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DC_HD __host__ __device__ __forceinline__
#else
#define DC_HD inline
#endif

namespace dcm {

// (CLS_NM2: run-length mantissas of more than 5 bits — one node per depth instead of a tree.  The static coder gives them the rates of
// CLS_NM; the fast coder, which runs on the same machinery, does not: qlfc.cpp:1316-1331)
enum : int { CLS_RF = 0, CLS_RE, CLS_RM, CLS_RP, CLS_NF, CLS_NE, CLS_NM, CLS_NM2, NUM_CLS };
enum : int { FAM_STATE = 0, FAM_CHAR = 1, FAM_STATIC = 2 };

// ---- type ids -------------------------------------------------------------------------------------------------------
constexpr int TAU_RF = 0, TAU_RE = 1, TAU_RM = 8, TAU_RP = 255, TAU_NF = 510, TAU_NE = 511, TAU_NM = 542, NUM_TAU = 1080;

DC_HD int bsr(uint32_t x) { return x ? 31 - __builtin_clz(x) : 0; }
// RM: B = 1..7, ctx in [1, 2^B): types of smaller B first
DC_HD int rm_off(int B) { return (1 << B) - 1 - B; }                       // sum_{b<B} (2^b - 1)
// NM: bits 1..5 are full trees (2^bits - 1 nodes), bits 6..31 are chains (ctx = 1..bits)
DC_HD int nm_off(int bits) { return bits <= 6 ? (1 << bits) - 1 - bits : 57 + (bits * (bits - 1)) / 2 - 15; }
constexpr int TAU_NM2 = TAU_NM + 57;                                         // = TAU_NM + nm_off(6): first type of the 6-bit chain
DC_HD int tau_class(int tau)
{
    return tau < TAU_RE ? CLS_RF : tau < TAU_RM ? CLS_RE : tau < TAU_RP ? CLS_RM : tau < TAU_NF ? CLS_RP
         : tau < TAU_NE ? CLS_NF : tau < TAU_NM ? CLS_NE : tau < TAU_NM2 ? CLS_NM : CLS_NM2;
}

// ---- canonical rounds ----------------------------------------------------------------------------------------------
// rank side: 0 RF | 1..7 RE s | 8..14 RM depth d | 15..22 RP depth d;   run side: 23 NF | 24..54 NE s | 55..85 NM depth d
constexpr int ROUND_RF = 0, ROUND_RE = 1, ROUND_RM = 8, ROUND_RP = 15, ROUND_NF = 23, ROUND_NE = 24, ROUND_NM = 55, NUM_ROUNDS = 86;
DC_HD int tau_round(int tau)
{
    if (tau < TAU_RE) return ROUND_RF;
    if (tau < TAU_RM) return ROUND_RE + (tau - TAU_RE);
    if (tau < TAU_RP) {                                        // depth of ctx inside its B-tree
        int t = tau - TAU_RM, B = 1;
        while (t >= (1 << B) - 1) { t -= (1 << B) - 1; ++B; }
        return ROUND_RM + bsr((uint32_t)t + 1);
    }
    if (tau < TAU_NF) return ROUND_RP + bsr((uint32_t)(tau - TAU_RP) + 1);
    if (tau < TAU_NE) return ROUND_NF;
    if (tau < TAU_NM) return ROUND_NE + (tau - TAU_NE);
    int t = tau - TAU_NM, bits = 1;
    for (;;) { const int cnt = bits <= 5 ? (1 << bits) - 1 : bits; if (t < cnt) break; t -= cnt; ++bits; }
    return ROUND_NM + (bits <= 5 ? bsr((uint32_t)t + 1) : t);
}

// ---- one run ("item") ----------------------------------------------------------------------------------------------
// packed item: [63:56] X (sort digit) | [55:53] sub-block | [52] avg_rank >= 32 | [51:44] rank | [43:13] run length
struct Item { uint32_t rank, run, sb, ge32; };
DC_HD uint64_t item_pack(uint32_t X, uint32_t sb, uint32_t ge32, uint32_t rank, uint32_t run)
{
    return ((uint64_t)X << 56) | ((uint64_t)sb << 53) | ((uint64_t)ge32 << 52) | ((uint64_t)rank << 44) | ((uint64_t)run << 13);
}
DC_HD Item item_unpack(uint64_t k)
{
    Item it;
    it.sb = (uint32_t)(k >> 53) & 7u; it.ge32 = (uint32_t)(k >> 52) & 1u; it.rank = (uint32_t)(k >> 44) & 0xffu; it.run = (uint32_t)(k >> 13) & 0x7fffffffu;
    return it;
}
DC_HD uint32_t item_X(uint64_t k) { return (uint32_t)(k >> 56); }

// number of decisions on each side (qlfc.cpp:904-975 / :990-1060)
DC_HD int count_rank_side(const Item& it, int max_rank)
{
    if (it.ge32) return max_rank + 1;
    if (it.rank == 1) return 1;
    const int B = bsr(it.rank);
    return 1 + (B - 1) + (B < max_rank ? 1 : 0) + B;
}
DC_HD int count_run_side(const Item& it)
{
    if (it.run == 1) return 1;
    const int bits = bsr(it.run);
    return 1 + bits + bits;
}

// The decision of canonical round r for this run, if it has one: returns tau (>= 0) and the coded bit, or -1.
DC_HD int decision(const Item& it, int max_rank, int r, uint32_t* bit)
{
    if (r < ROUND_NF) {
        const uint32_t rank = it.rank;
        if (r >= ROUND_RP) {                                   // escape coding: max_rank + 1 mantissa-like decisions
            const int d = r - ROUND_RP;
            if (!it.ge32 || d > max_rank) return -1;
            const uint32_t ctx = (1u << d) | ((rank >> (max_rank + 1 - d)) & ((1u << d) - 1u));
            *bit = (rank >> (max_rank - d)) & 1u;
            return TAU_RP + (int)ctx - 1;
        }
        if (it.ge32) return -1;
        if (r == ROUND_RF) { *bit = rank != 1u; return TAU_RF; }
        if (rank == 1u) return -1;
        const int B = bsr(rank);
        if (r < ROUND_RM) {
            const int s = r - ROUND_RE;                        // s = b - 1
            if (s <= B - 2) { *bit = 1u; return TAU_RE + s; }
            if (s == B - 1 && B < max_rank) { *bit = 0u; return TAU_RE + s; }
            return -1;
        }
        const int d = r - ROUND_RM;
        if (d >= B) return -1;
        const uint32_t ctx = rank >> (B - d);
        *bit = (rank >> (B - 1 - d)) & 1u;
        return TAU_RM + rm_off(B) + (int)ctx - 1;
    }
    const uint32_t run = it.run;
    if (r == ROUND_NF) { *bit = run != 1u; return TAU_NF; }
    if (run == 1u) return -1;
    const int bits = bsr(run);
    if (r < ROUND_NM) {
        const int s = r - ROUND_NE;
        if (s <= bits - 2) { *bit = 1u; return TAU_NE + s; }
        if (s == bits - 1) { *bit = 0u; return TAU_NE + s; }
        return -1;
    }
    const int d = r - ROUND_NM;
    if (d >= bits) return -1;
    const uint32_t ctx = bits <= 5 ? (run >> (bits - d)) : (uint32_t)(1 + d);
    *bit = (run >> (bits - 1 - d)) & 1u;
    return TAU_NM + nm_off(bits) + (int)ctx - 1;
}

// All decisions of a run in canonical (= stream) order: f(tau, bit, run_side).
template <class F>
DC_HD void enumerate(const Item& it, int max_rank, F&& f)
{
    const uint32_t rank = it.rank, run = it.run;
    if (it.ge32) {
        for (int d = 0; d <= max_rank; ++d) {
            const uint32_t ctx = (1u << d) | ((rank >> (max_rank + 1 - d)) & ((1u << d) - 1u));
            f(TAU_RP + (int)ctx - 1, (rank >> (max_rank - d)) & 1u, false);
        }
    } else {
        f(TAU_RF, rank != 1u ? 1u : 0u, false);
        if (rank != 1u) {
            const int B = bsr(rank);
            for (int s = 0; s <= B - 2; ++s) f(TAU_RE + s, 1u, false);
            if (B < max_rank) f(TAU_RE + B - 1, 0u, false);
            for (int d = 0; d < B; ++d) f(TAU_RM + rm_off(B) + (int)(rank >> (B - d)) - 1, (rank >> (B - 1 - d)) & 1u, false);
        }
    }
    f(TAU_NF, run != 1u ? 1u : 0u, true);
    if (run != 1u) {
        const int bits = bsr(run);
        for (int s = 0; s <= bits - 2; ++s) f(TAU_NE + s, 1u, true);
        f(TAU_NE + bits - 1, 0u, true);
        for (int d = 0; d < bits; ++d) {
            const uint32_t ctx = bits <= 5 ? (run >> (bits - d)) : (uint32_t)(1 + d);
            f(TAU_NM + nm_off(bits) + (int)ctx - 1, (run >> (bits - 1 - d)) & 1u, true);
        }
    }
}

// The k-th decision of a run (k counted over both sides, stream order), without walking the ones before it.
DC_HD int nth_decision(const Item& it, int max_rank, int n_rank, int k, uint32_t* bit, bool* run_side)
{
    if (k < n_rank) {
        const uint32_t rank = it.rank;
        *run_side = false;
        if (it.ge32) { return decision(it, max_rank, ROUND_RP + k, bit); }
        if (k == 0) { *bit = rank != 1u ? 1u : 0u; return TAU_RF; }
        const int B = bsr(rank);
        const int e = (B - 1) + (B < max_rank ? 1 : 0);
        if (k <= e) { const int sx = k - 1; *bit = sx + 1 < B ? 1u : 0u; return TAU_RE + sx; }
        const int d = k - 1 - e;
        *bit = (rank >> (B - 1 - d)) & 1u;
        return TAU_RM + rm_off(B) + (int)(rank >> (B - d)) - 1;
    }
    const int kk = k - n_rank;
    const uint32_t run = it.run;
    *run_side = true;
    if (kk == 0) { *bit = run != 1u ? 1u : 0u; return TAU_NF; }
    const int nb = bsr(run);
    if (kk <= nb) { const int sx = kk - 1; *bit = sx + 1 < nb ? 1u : 0u; return TAU_NE + sx; }
    const int d = kk - 1 - nb;
    const uint32_t ctx = nb <= 5 ? (run >> (nb - d)) : (uint32_t)(1 + d);
    *bit = (run >> (nb - 1 - d)) & 1u;
    return TAU_NM + nm_off(nb) + (int)ctx - 1;
}

// The CLASS (update rates, blend weights) and the coded bit of a run's k-th decision — what the probability stream needs; the type itself
// (which tree node) only matters to the partition.  Same case analysis as nth_decision without the node arithmetic
// (tools/devcoder_sim.cpp checks tau_class(nth_decision(...)) == nth_class(...) and the bits on every run it walks).
DC_HD int nth_class(const Item& it, int max_rank, int n_rank, int k, uint32_t* bit, bool* run_side)
{
    if (k < n_rank) {
        const uint32_t rank = it.rank;
        *run_side = false;
        if (it.ge32) { *bit = (rank >> (max_rank - k)) & 1u; return CLS_RP; }
        if (k == 0) { *bit = rank != 1u ? 1u : 0u; return CLS_RF; }
        const int B = bsr(rank);
        const int e = (B - 1) + (B < max_rank ? 1 : 0);
        if (k <= e) { *bit = k < B ? 1u : 0u; return CLS_RE; }
        *bit = (rank >> (B - 1 - (k - 1 - e))) & 1u;
        return CLS_RM;
    }
    const int kk = k - n_rank;
    const uint32_t run = it.run;
    *run_side = true;
    if (kk == 0) { *bit = run != 1u ? 1u : 0u; return CLS_NF; }
    const int nb = bsr(run);
    if (kk <= nb) { *bit = kk < nb ? 1u : 0u; return CLS_NE; }
    *bit = (run >> (nb - 1 - (kk - 1 - nb))) & 1u;
    return nb <= 5 ? CLS_NM : CLS_NM2;
}

// ---- counters --------------------------------------------------------------------------------------------------------
// predictor.h:53-61 with the family's tuned constants: bit 0 moves towards 4096 - th0, bit 1 towards th1 (arithmetic shifts)
// One form for both coders: v <- v + (((t_bit - v) * a_bit + r_bit) >> 12), arithmetic shift.
//   static coder (-e1): bit 0 moves towards t0 = 4096 - th0 with r0 = 0; bit 1 is v - (((v - t1) a1) >> 12), which is the same as
//                       v + (((t1 - v) a1 + 4095) >> 12) (minus the floor of a quotient = the ceiling of its negation): r1 = 4095;
//   fast coder (-e0):   p -= (p - target_bit) >> R (qlfc.cpp:1186-1331, shifts of 4..7) = v + ceil((target - v) / 2^R)
//                       = v + (((target - v) * 2^(12-R) + 4095) >> 12): a = 2^(12-R), r0 = r1 = 4095.
struct Rates { int t0, a0, t1, a1, r0, r1; };
DC_HD int step(int v, uint32_t bit, const Rates& R)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // both directions with the full-rate 24-bit multiply (|t - v| < 2^14, rates < 2^11: exact), then one select: the plain form
    // compiles to two exec-masked branches around quarter-rate 32-bit multiplies, twice the cycles of this on the serial chains
    const int up = v + ((__mul24(R.t0 - v, R.a0) + R.r0) >> 12);
    const int dn = v + ((__mul24(R.t1 - v, R.a1) + R.r1) >> 12);
    return bit ? dn : up;
#else
    return bit ? v + (((R.t1 - v) * R.a1 + R.r1) >> 12) : v + (((R.t0 - v) * R.a0 + R.r0) >> 12);
#endif
}

// Everything the kernels need from the tuned tables (filled on the host from qlfc_data.inc, passed by value).
struct ModelParams {
    Rates   rates[NUM_CLS][3];                               // [class][family]
    short   lr[NUM_CLS][3];                                  // blend weights: p = (ch * lr[0] + st * lr[1] + static * lr[2]) >> 5
    short   vmin[NUM_CLS][3], vmax[NUM_CLS][3];              // attainable counter range from the start value (tight two-sided brackets)
    short   init[NUM_CLS];                                   // value a chain starts from (2048; fast coder: 4096 rank side, 1024 run side)
};
// attainable range: closure of {init} under both maps (monotone maps -> an interval)
inline void model_closure(const Rates& R, int init, short* vmin, short* vmax)
{
    int lo = init, hi = init;
    for (bool grown = true; grown;) {
        grown = false;
        for (int s = lo; s <= hi; ++s) for (uint32_t b = 0; b < 2; ++b) {
            const int w = step(s, b, R);
            if (w < lo) { lo = w; grown = true; }
            if (w > hi) { hi = w; grown = true; }
        }
    }
    *vmin = (short)lo; *vmax = (short)hi;
}
// kStaticParams row layout (tools/gen_qlfc_data.py): S{th0,ar0,th1,ar1} C{...} P{...} mixer{4} LR0 LR1 LR2
inline void model_params_from_table(const short (*P)[19], ModelParams& M)
{
    for (int c = 0; c < NUM_CLS; ++c) {
        const int pc = c == CLS_NM2 ? CLS_NM : c;             // the table has one row for all run-length mantissas
        for (int f = 0; f < 3; ++f) {
            Rates& R = M.rates[c][f];
            R.t0 = 4096 - P[pc][4 * f + 0]; R.a0 = P[pc][4 * f + 1]; R.t1 = P[pc][4 * f + 2]; R.a1 = P[pc][4 * f + 3];
            R.r0 = 0; R.r1 = 4095;
            model_closure(R, 2048, &M.vmin[c][f], &M.vmax[c][f]);
        }
        M.lr[c][0] = P[pc][16]; M.lr[c][1] = P[pc][17]; M.lr[c][2] = P[pc][18];
        M.init[c] = 2048;
    }
}
// The fast coder (-e0, qlfc.cpp:1135-1336): ONE counter per decision, indexed by the run's symbol — the chains of the char family
// with other update maps (shifts instead of multiplications, targets per class and bit, qlfc.cpp:1186-1331; initial values
// qlfc_model.cpp:74-75) and nothing to blend.  Only [class][FAM_CHAR] is used; RP (escape coding) does not exist in this coder.
inline void model_params_fast(ModelParams& M)
{
    //                           RF     RE     RM     RP     NF     NE     NM    NM2
    static const short T0[] = {8016,  8114,  7999,  7999,  2025,  1962,  1951,  1987};     // target of a coded 0
    static const short T1[] = {  83,   122,   235,   235,    42,   142,   147,    46};     // target of a coded 1
    static const short S0[] = {   4,     4,     7,     7,     5,     4,     6,     5};     // shift (RF / NF: both bits the same)
    for (int c = 0; c < NUM_CLS; ++c) {
        for (int f = 0; f < 3; ++f) {
            Rates& R = M.rates[c][f];
            R.t0 = T0[c]; R.t1 = T1[c]; R.a0 = R.a1 = 1 << (12 - S0[c]); R.r0 = R.r1 = 4095;
            M.init[c] = (short)(c < CLS_NF ? 4096 : 1024);
            model_closure(R, M.init[c], &M.vmin[c][f], &M.vmax[c][f]);
            M.lr[c][f] = 0;
        }
    }
}
DC_HD int blend(int v_char, int v_state, int v_static, const short* lr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (__mul24(v_char, lr[0]) + __mul24(v_state, lr[1]) + __mul24(v_static, lr[2])) >> 5;
#else
    return (v_char * lr[0] + v_state * lr[1] + v_static * lr[2]) >> 5;
#endif
}

// ---- contexts of a run (qlfc.cpp:896-903, :978-989, :1063-1068) -------------------------------------------------------
DC_HD uint32_t avg_rank_next(uint32_t avg, uint32_t rank) { return (avg * 124u + rank * 4u) >> 7; }
DC_HD uint32_t run_hist_next(uint32_t h, uint32_t run) { return run == 1u ? (h + 2u) >> 2 : (h + 3u * (uint32_t)bsr(run) + 3u) >> 2; }
// window contexts from the previous runs of the same sub-block (prev[0] = run j-1, ...; absent runs contribute zero bits)
DC_HD uint32_t rank_state_index(uint32_t ctx_run, uint32_t ctx_rank4, uint32_t rank_hist) { return (ctx_run << 11) | (ctx_rank4 << 3) | rank_hist; }
DC_HD uint32_t run_state_index(uint32_t ctx_rank0, uint32_t ctx_run, uint32_t rank, uint32_t run_hist)
{
    const uint32_t r1 = rank - 1u;
    return (ctx_rank0 << 10) | (ctx_run << 6) | ((r1 < 7u ? r1 : 7u) << 3) | (run_hist < 7u ? run_hist : 7u);
}

// p-stream entry handed to the range coder: [11:0] probability, [12] bit, [13] first decision of a run
constexpr uint16_t PS_BIT = 1u << 12, PS_RUN = 1u << 13;
// ... of the fast coder: [12:0] probability (13 bits on the rank side, 11 on the run side), [13] bit, [14] first decision of a run,
// [15] run side (the range coder's precision follows it: qlfc.cpp:1186 / :1259)
constexpr uint16_t PSF_BIT = 1u << 13, PSF_RUN = 1u << 14, PSF_SIDE = 1u << 15;

}  // namespace dcm
