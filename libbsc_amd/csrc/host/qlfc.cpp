// qlfc.cpp — host-side QLFC encoders (static -e1, adaptive -e2, fast -e0), bit-exact with libbsc 3.3.5.
//
// Stream definition followed (reference file:line): header word + alphabet bits qlfc.cpp:852-891 (static),
// :486-525 (adaptive), :1146-1183 (fast); per-run decision tree :896-1126 / :530-820 / :1186-1331;
// counter update predictor.h:53-61; mixer predictor.h:121-183; range coder rangecoder.h:83-177.
// The organisation (flat run arrays -> templated decision walker -> range encoder) is ours.
#include "qlfc.h"
#include <immintrin.h>

#include <cstring>
#include <cstdlib>
#include <memory>
#include <algorithm>
#include <type_traits>
#include <emmintrin.h>
#include <tmmintrin.h>

#include "qlfc_data.inc"

namespace bschost {

// ------------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------------
static const QlfcTables* build_tables()
{
    QlfcTables* t = new QlfcTables();
    for (int i = 0; i < 4097; ++i) {
        t->stretch[i] = (short)((kStretchPacked[i >> 2] >> (16 * (i & 3))) & 0xffff);
        t->squash[i]  = (short)((kSquashPacked[i >> 2] >> (16 * (i & 3))) & 0xffff);
    }
    for (int i = 0; i < 32768; ++i) t->rank_state[i] = (uint8_t)(kRankStatePacked[i >> 3] >> (8 * (i & 7)));
    for (int i = 0; i < 8192; ++i)  t->run_state[i]  = (uint8_t)(kRunStatePacked[i >> 3] >> (8 * (i & 7)));
    return t;
}
const QlfcTables& qlfc_tables()
{
    static const QlfcTables* t = build_tables();
    return *t;
}

const short (*qlfc_static_params())[19] { return kStaticParams; }

static inline int bsr32(unsigned x) { return x ? 31 - __builtin_clz(x) : 0; }

// ------------------------------------------------------------------------------------------------
// run / rank front end
// ------------------------------------------------------------------------------------------------
void qlfc_runs(const uint8_t* in, int n, QlfcRuns& out)
{
    out.sym.clear(); out.rank.clear(); out.start.clear();
    RunView& V = out.view;
    V = RunView();
    if (n <= 0) return;
    // pass 1: maximal runs (count first so the arrays are allocated once)
    size_t m = 1;
    for (int i = 1; i < n; ++i) m += (in[i] != in[i - 1]);
    out.sym.resize(m); out.rank.resize(m); out.start.resize(m);
    {
        size_t j = 0; int i = 0;
        while (i < n) {
            const uint8_t c = in[i];
            int e = i + 1;
            const uint64_t pat = 0x0101010101010101ull * c;
            while (e + 8 <= n) {                     // word-at-a-time scan for long runs
                uint64_t w; memcpy(&w, in + e, 8);
                const uint64_t x = w ^ pat;
                if (x) { e += __builtin_ctzll(x) >> 3; goto done; }
                e += 8;
            }
            while (e < n && in[e] == c) ++e;
        done:
            out.sym[j] = c; out.start[j] = (uint32_t)i; ++j;
            i = e;
        }
    }
    // pass 2: the rank of a run is the move-to-front position its symbol has when it is seen NEXT
    uint8_t mtf[256 + 8];
    int     pos_of_last[256];
    int     nseen = 0;
    for (int c = 0; c < 256; ++c) pos_of_last[c] = -1;
    for (size_t j = 0; j < m; ++j) {
        const uint8_t c = out.sym[j];
        const int prev = pos_of_last[c];
        if (prev < 0) {
            V.first_seen[V.nsym++] = c;
            for (int p = nseen; p > 0; --p) mtf[p] = mtf[p - 1];
            mtf[0] = c; ++nseen;
        } else {
            uint8_t carry = mtf[0];                  // mtf[0] is the previous run's symbol, never c
            int p = 1;
            for (;; ++p) { const uint8_t t = mtf[p]; mtf[p] = carry; if (t == c) break; carry = t; }
            mtf[0] = c;
            out.rank[(size_t)prev] = (uint8_t)p;
        }
        pos_of_last[c] = (int)j;
    }
    for (int p = 0; p < nseen; ++p) out.rank[(size_t)pos_of_last[mtf[p]]] = (uint8_t)p;   // last occurrences
    out.rank[m - 1] = 1;                                                                  // qlfc.cpp:249/449
    V.sym = out.sym.data(); V.rank = out.rank.data(); V.start = out.start.data();
    V.count = (uint32_t)m; V.end = (uint32_t)n;
}

// ------------------------------------------------------------------------------------------------
// range encoder: 32-bit range, 64-bit low (bit 32 = carry), 16-bit little-endian output units
// ------------------------------------------------------------------------------------------------
class RangeEncoder {
public:
    void init(uint8_t* out, int out_size)
    {
        begin_ = out_ = out;
        limit_ = (ptrdiff_t)out_size - 16;          // rangecoder.h:127 (EOB margin)
        low_ = 0; range_ = 0xffffffffu; cache_ = 0; held_ = 0;
    }
    bool full() const { return (out_ - begin_) >= limit_; }

    // The coder's two hot words as plain locals of the caller: while a run is being coded they live in registers and are
    // written back to the object only around the (rare) renormalisation call and at the end of the run.
    struct Live { uint64_t low; uint32_t range; };
    __attribute__((always_inline)) inline Live enter() const { return Live{low_, range_}; }
    __attribute__((always_inline)) inline void leave(const Live& L) { low_ = L.low; range_ = L.range; }
    // The range after a decision: bit ? range - r : r.  Written as a select (cmov: the mantissa bits are close to coin flips, a branch
    // would mispredict) the chain a coder's next decision waits for is shift, multiply, subtract, select = 6 cycles; the arithmetic form
    // r + (m & (range - r - r)) of rounds 1-4 was 8.  The scalar and two-stream coders are bound by exactly that chain.
    static __attribute__((always_inline)) inline uint32_t next_range(uint32_t range, uint32_t r, unsigned bit)
    {
        const uint32_t alt = range - r;
#if defined(__clang__)
        return __builtin_unpredictable(bit != 0u) ? alt : r;
#else
        return bit != 0u ? alt : r;
#endif
    }
    template <int P> __attribute__((always_inline)) inline void encode_live(Live& L, unsigned bit, int p)
    {
        if (__builtin_expect(L.range < 0x10000u, 0)) { low_ = L.low; shift(); L.low = low_; L.range <<= 16; }
        const uint32_t r = (L.range >> P) * (uint32_t)p;
        const uint32_t m = 0u - bit;
        L.low  += (uint64_t)(r & m);
        L.range = next_range(L.range, r, bit);
    }
    // the same, keeping `is_full` (= full(), which only changes inside shift()) in a caller's register: the run-start budget test of
    // the p-stream coders costs three loads and a compare per decision otherwise
    template <int P> __attribute__((always_inline)) inline void encode_live_f(Live& L, unsigned bit, int p, unsigned& is_full)
    {
        if (__builtin_expect(L.range < 0x10000u, 0)) { low_ = L.low; shift(); L.low = low_; L.range <<= 16; is_full = (unsigned)full(); }
        const uint32_t r = (L.range >> P) * (uint32_t)p;
        const uint32_t m = 0u - bit;
        L.low  += (uint64_t)(r & m);
        L.range = next_range(L.range, r, bit);
    }
    template <int P> __attribute__((always_inline)) inline void encode(unsigned bit, int p)
    {
        if (__builtin_expect(range_ < 0x10000u, 0)) { shift(); range_ <<= 16; }
        const uint32_t r = (range_ >> P) * (uint32_t)p;
        const uint32_t m = 0u - bit;                 // branch-free: mantissa bits are close to coin flips
        low_  += (uint64_t)(r & m);
        range_ = next_range(range_, r, bit);
    }
    // precision picked per decision (the fast coder behind the device model: 13 bits on the rank side, 11 on the run side)
    __attribute__((always_inline)) inline void encode_live_var(Live& L, unsigned bit, unsigned p, unsigned prec, unsigned& is_full)
    {
        if (__builtin_expect(L.range < 0x10000u, 0)) { low_ = L.low; shift(); L.low = low_; L.range <<= 16; is_full = (unsigned)full(); }
        const uint32_t r = (L.range >> prec) * (uint32_t)p;
        const uint32_t m = 0u - bit;
        L.low  += (uint64_t)(r & m);
        L.range = next_range(L.range, r, bit);
    }
    inline void encode_half(unsigned bit) { encode<12>(bit, 2048); }     // rangecoder.h:179-182
    void encode_word(uint32_t w) { for (int b = 31; b >= 0; --b) encode_half((w >> b) & 1u); }

    // the output half of shift() for a coder whose low word lives elsewhere (qlfc_encode_static_pstream_x8):
    // top16 = bits 16..31 of low, carry = bit 32
    inline void emit_unit(uint32_t top16, uint32_t carry)
    {
        if (top16 != 0xffffu || carry) {
            put16(cache_ + carry);
            for (; held_; --held_) put16(carry - 1u);     // 0xffff without carry, 0x0000 after one
            cache_ = top16;
        } else {
            ++held_;
        }
    }

    int finish()
    {
        if (range_ < 0x10000u) shift();
        shift(); shift(); shift();
        return (int)(out_ - begin_);
    }

private:
    inline void put16(uint32_t v) { out_[0] = (uint8_t)v; out_[1] = (uint8_t)(v >> 8); out_ += 2; }
    void shift()
    {
        const uint32_t low32 = (uint32_t)low_;
        const uint32_t carry = (uint32_t)(low_ >> 32);
        if (low32 < 0xffff0000u || carry) {
            put16(cache_ + carry);
            for (; held_; --held_) put16(carry - 1u);     // 0xffff without carry, 0x0000 after one
            cache_ = low32 >> 16;
        } else {
            ++held_;
        }
        low_ = (uint64_t)(uint32_t)(low32 << 16);
    }

    uint64_t low_; uint32_t range_, cache_, held_;
    uint8_t *out_, *begin_; ptrdiff_t limit_;
};

// ------------------------------------------------------------------------------------------------
// model state
// ------------------------------------------------------------------------------------------------
enum { RANK_FIRST = 0, RANK_EXP, RANK_MANT, RANK_ESC, RUN_FIRST, RUN_EXP, RUN_MANT };

struct Mixer {                       // predictor.h:74-213
    int   w0, w1, w2;
    short map[17];
    void init(const QlfcTables& T)
    {
        w0 = w1 = 2048 << 5; w2 = 0;
        for (int p = 0; p < 17; ++p) map[p] = T.squash[2048 + (p - 8) * 256];
    }
};

#define BSC_ALWAYS_INLINE __attribute__((always_inline)) inline
// The mixer's weighted sum and weight updates (predictor.h:102-183) are plain `int` expressions in the reference and do overflow on
// some inputs (found by UBSan in round 4: fuzz seed 303); what the reference's binary computes is the two's-complement wrap, so that
// is what the format is.  Spelled out in unsigned arithmetic here: defined behaviour, same bits.
static BSC_ALWAYS_INLINE int wrap_mul(int a, int b) { return (int)((uint32_t)a * (uint32_t)b); }
static BSC_ALWAYS_INLINE int wrap_add3(int a, int b, int c) { return (int)((uint32_t)a + (uint32_t)b + (uint32_t)c); }
static BSC_ALWAYS_INLINE void bump(short& p, unsigned bit, int th0, int ar0, int th1, int ar1)
{
    // predictor.h:53-61 (the four-argument form at :44-50 is algebraically the same map)
    const int up   = ((4096 - th0 - p) * ar0) >> 12;       // bit 0: move towards 4096 - th0
    const int down = ((p - th1) * ar1) >> 12;               // bit 1: move towards th1
    p = (short)(p + (bit ? -down : up));                    // select, not branch (constant-folds when bit is a literal)
}

struct Counters1 {
    // rank side
    short rT_stat, rT_state[256], rT_chr[256];
    short rE_stat[8], rE_state[256][8], rE_chr[256][8];
    short rM_stat[8][256], rM_state[8][256][256], rM_chr[8][256][256];
    short rP_stat[256], rP_state[256][256], rP_chr[256][256];
    // run side
    short nT_stat, nT_state[256], nT_chr[256];
    short nE_stat[32], nE_state[256][32], nE_chr[256][32];
    short nM_stat[32][32], nM_state[32][256][32], nM_chr[32][256][32];
};
struct Mixers1 {
    Mixer rank[256], rank_exp[8][8], rank_mant[8], rank_esc[256], run[256], run_exp[32][32], run_mant[32];
};

static void fill_shorts(void* p, size_t bytes, short v)
{
    short* s = (short*)p;
    std::fill(s, s + bytes / 2, v);
}

// Per class and per coded bit: {target, rate << 4, sign} for the three counters [char, state, pos].  With s = +1 for a
// 0 bit (move up towards 4096 - th0) and s = -1 for a 1 bit (move down towards th1) both updates are
//     p += s * (((tgt - p) * s * rate) >> 12),
// i.e. psignw / pmulhw / psignw: one multiply instead of computing both directions and selecting.
struct alignas(16) StepTable { short lr[8]; short tgt[2][8]; short ar[2][8]; short sgn[2][8]; };
template <bool ADAPT, int CLS> struct StepConst {
    static constexpr const short* P = ADAPT ? kAdaptiveParams[CLS] : kStaticParams[CLS];
    static constexpr StepTable value = {
        {P[16], P[17], P[18], 0, 0, 0, 0, 0},
        {{(short)(4096 - P[4]), (short)(4096 - P[0]), (short)(4096 - P[8]), 0, 0, 0, 0, 0}, {P[6], P[2], P[10], 0, 0, 0, 0, 0}},
        {{(short)(P[5] << 4), (short)(P[1] << 4), (short)(P[9] << 4), 0, 0, 0, 0, 0}, {(short)(P[7] << 4), (short)(P[3] << 4), (short)(P[11] << 4), 0, 0, 0, 0, 0}},
        {{1, 1, 1, 1, 1, 1, 1, 1}, {-1, -1, -1, -1, -1, -1, -1, -1}}};
};
template <bool ADAPT, int CLS> constexpr StepTable StepConst<ADAPT, CLS>::value;

// One binary decision of class CLS: three counters (+ mixer), update, code.
//
// Static coder: the three counters are handled as one SSE vector [char, state, pos]: pmaddwd forms the weighted
// probability, pmulhw(d, rate << 4) is exactly (d * rate) >> 12 for |d| < 2^15 and rate < 2^11 (all tuned rates are
// <= 1364), so the update is bit-identical to predictor.h:53-61 at a third of the scalar instruction count.
template <int CLS, bool ADAPT = false>
static BSC_ALWAYS_INLINE int static_step(unsigned bit, short& st, short& ch, short& sp)
{
    constexpr const StepTable& P = StepConst<ADAPT, CLS>::value;
    __m128i v = _mm_cvtsi32_si128((int)((uint32_t)(uint16_t)ch | ((uint32_t)(uint16_t)st << 16)));
    v = _mm_insert_epi16(v, sp, 2);                                           // [ch, st, sp, 0, ...]
    const __m128i m  = _mm_madd_epi16(v, *reinterpret_cast<const __m128i*>(P.lr));     // [ch*LR0 + st*LR1, sp*LR2, 0, 0]
    const int p = (_mm_cvtsi128_si32(m) + _mm_cvtsi128_si32(_mm_srli_si128(m, 4))) >> 5;
    __m128i nv;
    if (!ADAPT) {       // one multiply, parameters picked by the bit (measured on the EPYC 9575F: static coder -2.5 %)
        const __m128i sg = *reinterpret_cast<const __m128i*>(P.sgn[bit]);
        const __m128i d  = _mm_sign_epi16(_mm_sub_epi16(*reinterpret_cast<const __m128i*>(P.tgt[bit]), v), sg);      // bit 0: tgt0 - v;  bit 1: v - tgt1
        nv = _mm_add_epi16(v, _mm_sign_epi16(_mm_mulhi_epi16(d, *reinterpret_cast<const __m128i*>(P.ar[bit])), sg));
    } else {            // both directions, then select (faster next to the mixer's scalar work: adaptive coder +5 % otherwise)
        const __m128i up   = _mm_add_epi16(v, _mm_mulhi_epi16(_mm_sub_epi16(*reinterpret_cast<const __m128i*>(P.tgt[0]), v), *reinterpret_cast<const __m128i*>(P.ar[0])));
        const __m128i down = _mm_sub_epi16(v, _mm_mulhi_epi16(_mm_sub_epi16(v, *reinterpret_cast<const __m128i*>(P.tgt[1])), *reinterpret_cast<const __m128i*>(P.ar[1])));
        const __m128i sel  = _mm_set1_epi16((short)(0 - (int)bit));
        nv = _mm_or_si128(_mm_and_si128(sel, down), _mm_andnot_si128(sel, up));
    }
    const uint32_t lo = (uint32_t)_mm_cvtsi128_si32(nv);
    ch = (short)(lo & 0xffffu); st = (short)(lo >> 16); sp = (short)_mm_extract_epi16(nv, 2);
    return p;
}

template <int CLS>
static BSC_ALWAYS_INLINE void decide_static(RangeEncoder& rc, RangeEncoder::Live& L, unsigned bit, short& st, short& ch, short& sp)
{
    rc.encode_live<12>(L, bit, static_step<CLS>(bit, st, ch, sp));
}
template <int CLS, bool ADAPT, class LiveT>
static BSC_ALWAYS_INLINE void decide(RangeEncoder& rc, LiveT& L, const QlfcTables& T, unsigned bit, short& st, short& ch, short& sp, Mixer* mx)
{
    if constexpr (!ADAPT) { decide_static<CLS>(rc, L, bit, st, ch, sp); return; } else {
    constexpr const short* P = kAdaptiveParams[CLS];
    const int p0 = ch, p1 = st, p2 = sp;
    (void)static_step<CLS, true>(bit, st, ch, sp);            // the three counter updates as one vector op
    const int s0 = T.stretch[p0], s1 = T.stretch[p1], s2 = T.stretch[p2];
    short sp16 = (short)(wrap_add3(wrap_mul(s0, mx->w0), wrap_mul(s1, mx->w1), wrap_mul(s2, mx->w2)) >> 17);
    if (sp16 < -2047) sp16 = -2047;
    if (sp16 >  2047) sp16 =  2047;
    const int frac = sp16 & 255;
    const int idx  = (sp16 + 2048) >> 8;
    const int sq   = T.squash[2048 + sp16];
    const int mapped = mx->map[idx] + (((mx->map[idx + 1] - mx->map[idx]) * frac) >> 8);
    const int p = (3 * sq + mapped) >> 2;
    bump(mx->map[idx],     bit, P[12], P[13], P[14], P[15]);
    bump(mx->map[idx + 1], bit, P[12], P[13], P[14], P[15]);
    const int eps = p - (bit ? 1 : 4095);
    mx->w0 = (int)((uint32_t)mx->w0 - (uint32_t)(wrap_mul(wrap_mul(P[16], eps), s0) >> 16));
    mx->w1 = (int)((uint32_t)mx->w1 - (uint32_t)(wrap_mul(wrap_mul(P[17], eps), s1) >> 16));
    mx->w2 = (int)((uint32_t)mx->w2 - (uint32_t)(wrap_mul(wrap_mul(P[18], eps), s2) >> 16));
    rc.encode_live<12>(L, bit, p);
    }
}

// Alphabet header shared by the three coders: for every symbol in order of first appearance emit only
// the bits not implied by the set of still-possible symbols; a repeated symbol terminates the list.
template <class EmitBit>
static int encode_alphabet(const RunView& R, EmitBit&& emit)
{
    bool used[256] = {false};
    int prev = -1;
    int max_rank = 7;
    for (int slot = 0; slot < 256; ++slot) {
        const int cur = (slot < R.nsym) ? R.first_seen[slot] : R.first_seen[R.nsym - 1];
        for (int bit = 7; bit >= 0; --bit) {
            bool can0 = false, can1 = false;
            for (int c = 0; c < 256 && !(can0 && can1); ++c) {
                if ((c == prev || !used[c]) && (cur >> (bit + 1)) == (c >> (bit + 1))) {
                    if (c & (1 << bit)) can1 = true; else can0 = true;
                }
            }
            if (can0 && can1) emit((unsigned)((cur >> bit) & 1));
        }
        if (cur == prev) { max_rank = bsr32((unsigned)(slot - 1)); break; }
        prev = cur; used[cur] = true;
    }
    return max_rank;
}

// The decision walker: enumerates, run by run, every binary decision of the static / adaptive coders with the
// three counter slots (state-, char- and position-indexed) and the mixer slot it uses.  All context indices are
// pure functions of the run data, never of the coder state — which is what lets the policies below either code
// directly or hand the three counter families to separate threads.  Returns false when the policy aborts.
// Per-stream context state of the walker (everything the context indices depend on besides the run data).
struct WalkState {
    int ctx_rank0 = 0, ctx_rank4 = 0, ctx_run = 0, avg_rank = 0;
    uint8_t rank_hist[256], run_hist[256];
    WalkState() { memset(rank_hist, 0, sizeof rank_hist); memset(run_hist, 0, sizeof run_hist); }
};

// One run: all its binary decisions, in stream order.
template <bool ADAPT, class Policy>
static BSC_ALWAYS_INLINE void walk_step(WalkState& W, const RunView& R, const QlfcTables& T, const int max_rank, Counters1& K, Mixers1* M,
                                        Policy& pol, typename Policy::Live& live, const uint32_t j)
{
    int ctx_rank0 = W.ctx_rank0, ctx_rank4 = W.ctx_rank4, ctx_run = W.ctx_run, avg_rank = W.avg_rank;
    uint8_t* const rank_hist = W.rank_hist; uint8_t* const run_hist = W.run_hist;
    {
        const int c = R.sym[j];
        int rank = R.rank[j];
        const int run = (int)R.len(j);

        // ---------------- rank ----------------
        int hist = rank_hist[c];
        int state = T.rank_state[(ctx_run << 11) | (ctx_rank4 << 3) | hist];
        if (avg_rank < 32) {
            pol.template decide<RANK_FIRST>(live, rank != 1, K.rT_state[state], K.rT_chr[c], K.rT_stat, ADAPT ? &M->rank[c] : nullptr);
            if (rank == 1) {
                rank_hist[c] = 0;
            } else {
                const int bits = bsr32((unsigned)rank);
                rank_hist[c] = (uint8_t)bits;
                // One indirect jump on the exponent instead of data-dependent loop exits: each case is straight-line
                // code (bits-1 ones, the optional terminating zero, then `bits` mantissa decisions).
                auto tail = [&](auto BITS_T) {
                    constexpr int BITS = decltype(BITS_T)::value;
#pragma GCC unroll 8
                    for (int b = 1; b < BITS; ++b)
                        pol.template decide<RANK_EXP>(live, 1, K.rE_state[state][b - 1], K.rE_chr[c][b - 1], K.rE_stat[b - 1],
                                                ADAPT ? &M->rank_exp[hist > b ? hist : b][b] : nullptr);
                    if (BITS < max_rank)
                        pol.template decide<RANK_EXP>(live, 0, K.rE_state[state][BITS - 1], K.rE_chr[c][BITS - 1], K.rE_stat[BITS - 1],
                                                ADAPT ? &M->rank_exp[hist > BITS ? hist : BITS][BITS] : nullptr);
                    short* ms = K.rM_state[BITS][state]; short* mc = K.rM_chr[BITS][c]; short* mp = K.rM_stat[BITS];
                    int ctx = 1;
#pragma GCC unroll 8
                    for (int b = BITS - 1; b >= 0; --b) {
                        const unsigned v = (unsigned)(rank >> b) & 1u;
                        pol.template decide<RANK_MANT>(live, v, ms[ctx], mc[ctx], mp[ctx], ADAPT ? &M->rank_mant[BITS] : nullptr);
                        ctx += ctx + (int)v;
                    }
                };
                switch (bits) {
                    case 1: tail(std::integral_constant<int, 1>()); break;
                    case 2: tail(std::integral_constant<int, 2>()); break;
                    case 3: tail(std::integral_constant<int, 3>()); break;
                    case 4: tail(std::integral_constant<int, 4>()); break;
                    case 5: tail(std::integral_constant<int, 5>()); break;
                    case 6: tail(std::integral_constant<int, 6>()); break;
                    default: tail(std::integral_constant<int, 7>()); break;
                }
            }
        } else {
            rank_hist[c] = (uint8_t)bsr32((unsigned)rank);
            short* es = K.rP_state[state]; short* ec = K.rP_chr[c]; short* ep = K.rP_stat;
            for (int ctx = 1, b = max_rank; b >= 0; --b) {
                const unsigned v = (unsigned)(rank >> b) & 1u;
                pol.template decide<RANK_ESC>(live, v, es[ctx], ec[ctx], ep[ctx], ADAPT ? &M->rank_esc[ctx] : nullptr);
                ctx += ctx + (int)v;
            }
        }

        // ---------------- run length ----------------
        avg_rank = (avg_rank * 124 + rank * 4) >> 7;
        rank -= 1;
        hist = run_hist[c];
        state = T.run_state[(ctx_rank0 << 10) | (ctx_run << 6) | ((rank < 7 ? rank : 7) << 3) | (hist < 7 ? hist : 7)];
        pol.template decide<RUN_FIRST>(live, run != 1, K.nT_state[state], K.nT_chr[c], K.nT_stat, ADAPT ? &M->run[c] : nullptr);
        if (run == 1) {
            run_hist[c] = (uint8_t)((run_hist[c] + 2) >> 2);
        } else {
            const int bits = bsr32((unsigned)run);
            run_hist[c] = (uint8_t)((run_hist[c] + 3 * bits + 3) >> 2);
            for (int b = 1; b < bits; ++b)
                pol.template decide<RUN_EXP>(live, 1, K.nE_state[state][b - 1], K.nE_chr[c][b - 1], K.nE_stat[b - 1],
                                       ADAPT ? &M->run_exp[hist > b ? hist : b][b] : nullptr);
            pol.template decide<RUN_EXP>(live, 0, K.nE_state[state][bits - 1], K.nE_chr[c][bits - 1], K.nE_stat[bits - 1],
                                   ADAPT ? &M->run_exp[hist > bits ? hist : bits][bits] : nullptr);
            short* ms = K.nM_state[bits][state]; short* mc = K.nM_chr[bits][c]; short* mp = K.nM_stat[bits];
            for (int ctx = 1, b = bits - 1; b >= 0; --b) {
                const unsigned v = (unsigned)(run >> b) & 1u;
                pol.template decide<RUN_MANT>(live, v, ms[ctx], mc[ctx], mp[ctx], ADAPT ? &M->run_mant[bits] : nullptr);
                ctx = (bits <= 5) ? (ctx + ctx + (int)v) : (ctx + 1);
            }
        }

        ctx_rank0 = ((ctx_rank0 << 1) | (rank == 0 ? 1 : 0)) & 0x7;
        ctx_rank4 = ((ctx_rank4 << 2) | (rank < 3 ? rank : 3)) & 0xff;
        ctx_run   = ((ctx_run   << 1) | (run < 3 ? 1 : 0)) & 0xf;
    }
    W.ctx_rank0 = ctx_rank0; W.ctx_rank4 = ctx_rank4; W.ctx_run = ctx_run; W.avg_rank = avg_rank;
}

template <bool ADAPT, class Policy>
static bool walk_model1(const RunView& R, const QlfcTables& T, const int max_rank, Counters1& K, Mixers1* M, Policy& pol)
{
    WalkState W;
    const uint32_t m = R.count;
    typename Policy::Live live = pol.enter();           // coder state that lives in registers for the whole stream
    for (uint32_t j = 0; j < m; ++j) {
        if (!pol.begin_run()) { pol.leave(live); return false; }
        walk_step<ADAPT>(W, R, T, max_rank, K, M, pol, live, j);
    }
    pol.leave(live);
    return true;
}

template <bool ADAPT>
struct DirectPolicy {
    RangeEncoder& rc; const QlfcTables& T;
    // the range coder's two hot words live in registers for the whole stream (static coder -3.5 % per stream on the EPYC 9575F,
    // adaptive coder unchanged)
    using Live = RangeEncoder::Live;
    inline bool begin_run() { return !rc.full(); }
    __attribute__((always_inline)) inline Live enter() { return rc.enter(); }
    __attribute__((always_inline)) inline void leave(const Live& L) { rc.leave(L); }
    template <int CLS> __attribute__((always_inline)) inline void decide(Live& L, unsigned bit, short& st, short& ch, short& sp, Mixer* mx) { bschost::decide<CLS, ADAPT>(rc, L, T, bit, st, ch, sp, mx); }
};

// Encoder model storage is kept per thread and re-initialised per sub-block: a fresh 3.5 MB allocation per sub-block costs
// page faults and kernel zeroing on top of the fill (the coder threads run under a CPU-time quota; every core-ms counts).
static Counters1* tl_counters()
{
    static thread_local std::unique_ptr<Counters1> k;
    if (!k) k.reset(new Counters1);
    fill_shorts(k.get(), sizeof(Counters1), 2048);
    return k.get();
}
static Mixers1* tl_mixers(const QlfcTables& T)
{
    static thread_local std::unique_ptr<Mixers1> m;
    if (!m) m.reset(new Mixers1);
    Mixer* all = reinterpret_cast<Mixer*>(m.get());
    for (size_t i = 0; i < sizeof(Mixers1) / sizeof(Mixer); ++i) all[i].init(T);
    return m.get();
}
static Counters1* new_counters()
{
    Counters1* k = new Counters1;
    fill_shorts(k, sizeof(Counters1), 2048);
    return k;
}

template <bool ADAPT>
static int encode_model1(const RunView& R, uint8_t* out, int in_size, int out_size)
{
    const QlfcTables& T = qlfc_tables();
    Counters1* Cn = tl_counters();
    Mixers1* Mx = ADAPT ? tl_mixers(T) : nullptr;
    RangeEncoder rc;
    rc.init(out, out_size);
    rc.encode_word((uint32_t)in_size);
    const int max_rank = encode_alphabet(R, [&](unsigned b) { rc.encode_half(b); });
    DirectPolicy<ADAPT> pol{rc, T};
    if (!walk_model1<ADAPT>(R, T, max_rank, *Cn, Mx, pol)) return NOT_COMPRESSIBLE;
    return rc.finish();
}

// ------------------------------------------------------------------------------------------------
// fast coder (-e0): one counter per context, shift updates, 13-bit (rank) / 11-bit (run) precision
// ------------------------------------------------------------------------------------------------
struct Counters2 {
    short r_exp[256][8];  short r_mant[256][8][256];
    short n_exp[256][32]; short n_mant[256][32][32];
};
template <int R> static inline void nudge(short& p, int target) { p = (short)(p - ((p - target) >> R)); }

static int encode_model2(const RunView& R, uint8_t* out, int in_size, int out_size)
{
    std::unique_ptr<Counters2> Cn(new Counters2);
    fill_shorts(Cn->r_exp, sizeof(Cn->r_exp) + sizeof(Cn->r_mant), 4096);      // qlfc_model.cpp:74
    fill_shorts(Cn->n_exp, sizeof(Cn->n_exp) + sizeof(Cn->n_mant), 1024);      // qlfc_model.cpp:75
    Counters2& K = *Cn;

    RangeEncoder rc;
    rc.init(out, out_size);
    rc.encode_word((uint32_t)in_size);
    encode_alphabet(R, [&](unsigned b) { rc.encode<1>(b, 1); });               // qlfc.cpp:1174

    const uint32_t m = R.count;
    for (uint32_t j = 0; j < m; ++j) {
        if (rc.full()) return NOT_COMPRESSIBLE;
        const int c = R.sym[j];
        const unsigned rank = R.rank[j];
        const unsigned run = R.len(j);
        {
            short* e = K.r_exp[c];
            if (rank == 1) { const int p = e[0]; nudge<4>(e[0], 8016); rc.encode<13>(0, p); }
            else {
                { const int p = e[0]; nudge<4>(e[0], 83); rc.encode<13>(1, p); }
                const int bits = bsr32(rank);
                for (int b = 1; b < bits; ++b) { const int p = e[b]; nudge<4>(e[b], 122); rc.encode<13>(1, p); }
                if (bits < 7) { const int p = e[bits]; nudge<4>(e[bits], 8114); rc.encode<13>(0, p); }
                short* mt = K.r_mant[c][bits];
                for (int ctx = 1, b = bits - 1; b >= 0; --b) {
                    const unsigned v = (rank >> b) & 1u;
                    const int p = mt[ctx]; nudge<7>(mt[ctx], v ? 235 : 7999); rc.encode<13>(v, p);
                    ctx += ctx + (int)v;
                }
            }
        }
        {
            short* e = K.n_exp[c];
            if (run == 1) { const int p = e[0]; nudge<5>(e[0], 2025); rc.encode<11>(0, p); }
            else {
                { const int p = e[0]; nudge<5>(e[0], 42); rc.encode<11>(1, p); }
                const int bits = bsr32(run);
                for (int b = 1; b < bits; ++b) { const int p = e[b]; nudge<4>(e[b], 142); rc.encode<11>(1, p); }
                { const int p = e[bits]; nudge<4>(e[bits], 1962); rc.encode<11>(0, p); }
                short* mt = K.n_mant[c][bits];
                if (bits <= 5) {
                    for (int ctx = 1, b = bits - 1; b >= 0; --b) {
                        const unsigned v = (run >> b) & 1u;
                        const int p = mt[ctx]; nudge<6>(mt[ctx], v ? 147 : 1951); rc.encode<11>(v, p);
                        ctx += ctx + (int)v;
                    }
                } else {
                    for (int ctx = 1, b = bits - 1; b >= 0; --b, ++ctx) {
                        const unsigned v = (run >> b) & 1u;
                        const int p = mt[ctx]; nudge<5>(mt[ctx], v ? 46 : 1987); rc.encode<11>(v, p);
                    }
                }
            }
        }
    }
    return rc.finish();
}

// ------------------------------------------------------------------------------------------------
// decoders (qlfc.cpp:1366-2127, rangecoder.h:200-270): the same models driven by decoded bits
// ------------------------------------------------------------------------------------------------
class RangeDecoder {
public:
    // in_end: one past the last byte this decoder may read (a corrupt stream must not walk off its payload; past the end
    // the input reads as zeros, which ends in a size / run-length check failing)
    void init(const uint8_t* in, const uint8_t* in_end)
    {
        in_ = in; end_ = in_end; code_ = 0; range_ = 0xffffffffu;
        code_ = (code_ << 16) | next16(); code_ = (code_ << 16) | next16(); code_ = (code_ << 16) | next16();
    }
    template <int P> inline unsigned decode(int p)
    {
        if (range_ < 0x10000u) { range_ <<= 16; code_ = (code_ << 16) | next16(); }
        const uint32_t r = (range_ >> P) * (uint32_t)p;
        const unsigned bit = code_ >= r;
        range_ = bit ? range_ - r : r;
        code_  = bit ? code_ - r : code_;
        return bit;
    }
    inline unsigned decode_half() { return decode<12>(2048); }
    uint32_t decode_word() { uint32_t w = 0; for (int b = 0; b < 32; ++b) w += w + decode_half(); return w; }
private:
    inline uint32_t next16()
    {
        if (__builtin_expect((uintptr_t)end_ - (uintptr_t)in_ < 2, 0)) { const uint32_t v = (in_ < end_) ? (uint32_t)in_[0] : 0u; in_ = end_; return v; }
        const uint32_t v = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8); in_ += 2; return v;
    }
    const uint8_t* in_; const uint8_t* end_; uint32_t code_, range_;
};

// Alphabet header: rebuilds the first-appearance list (with its terminator) and max_rank.
template <class GetBit>
static int decode_alphabet(uint8_t* mtf, GetBit&& get)
{
    bool used[256] = {false};
    int prev = -1, max_rank = 7;
    for (int slot = 0; slot < 256; ++slot) {
        int cur = 0;
        for (int bit = 7; bit >= 0; --bit) {
            bool can0 = false, can1 = false;
            for (int c = 0; c < 256 && !(can0 && can1); ++c)
                if ((c == prev || !used[c]) && cur == (c >> (bit + 1))) { if (c & (1 << bit)) can1 = true; else can0 = true; }
            if (can0 && can1) cur += cur + (int)get();
            else cur += cur + (can1 ? 1 : 0);
        }
        mtf[slot] = (uint8_t)cur;
        if (cur == prev) { max_rank = bsr32((unsigned)(slot - 1)); break; }
        prev = cur; used[cur] = true;
    }
    return max_rank;
}

// one decoded decision of class CLS
template <int CLS, bool ADAPT>
static BSC_ALWAYS_INLINE unsigned undecide(RangeDecoder& rd, const QlfcTables& T, short& st, short& ch, short& sp, Mixer* mx)
{
    constexpr const short* P = ADAPT ? kAdaptiveParams[CLS] : kStaticParams[CLS];
    const int p0 = ch, p1 = st, p2 = sp;
    unsigned bit;
    if (!ADAPT) {
        bit = rd.decode<12>((p0 * P[16] + p1 * P[17] + p2 * P[18]) >> 5);
    } else {
        const int s0 = T.stretch[p0], s1 = T.stretch[p1], s2 = T.stretch[p2];
        short sp16 = (short)(wrap_add3(wrap_mul(s0, mx->w0), wrap_mul(s1, mx->w1), wrap_mul(s2, mx->w2)) >> 17);
        if (sp16 < -2047) sp16 = -2047;
        if (sp16 >  2047) sp16 =  2047;
        const int frac = sp16 & 255, idx = (sp16 + 2048) >> 8, sq = T.squash[2048 + sp16];
        const int mapped = mx->map[idx] + (((mx->map[idx + 1] - mx->map[idx]) * frac) >> 8);
        const int p = (3 * sq + mapped) >> 2;
        bit = rd.decode<12>(p);
        bump(mx->map[idx],     bit, P[12], P[13], P[14], P[15]);
        bump(mx->map[idx + 1], bit, P[12], P[13], P[14], P[15]);
        const int eps = p - (bit ? 1 : 4095);
        mx->w0 = (int)((uint32_t)mx->w0 - (uint32_t)(wrap_mul(wrap_mul(P[16], eps), s0) >> 16));
        mx->w1 = (int)((uint32_t)mx->w1 - (uint32_t)(wrap_mul(wrap_mul(P[17], eps), s1) >> 16));
        mx->w2 = (int)((uint32_t)mx->w2 - (uint32_t)(wrap_mul(wrap_mul(P[18], eps), s2) >> 16));
    }
    bump(st, bit, P[0], P[1], P[2],  P[3]);
    bump(ch, bit, P[4], P[5], P[6],  P[7]);
    bump(sp, bit, P[8], P[9], P[10], P[11]);
    return bit;
}

static inline void requeue(uint8_t* mtf, int rank, uint8_t c)      // the head symbol will next be met at position `rank`
{
    for (int r = 0; r < rank; ++r) mtf[r] = mtf[r + 1];
    mtf[rank] = c;
}

template <bool ADAPT>
static int decode_model1(const uint8_t* in, const uint8_t* in_end, uint8_t* out, int max_out)
{
    const QlfcTables& T = qlfc_tables();
    std::unique_ptr<Counters1> Cn(new_counters());
    std::unique_ptr<Mixers1> Mx;
    if (ADAPT) {
        Mx.reset(new Mixers1);
        Mixer* all = reinterpret_cast<Mixer*>(Mx.get());
        for (size_t i = 0; i < sizeof(Mixers1) / sizeof(Mixer); ++i) all[i].init(T);
    }
    Counters1& K = *Cn; Mixers1* M = Mx.get();
    RangeDecoder rd; rd.init(in, in_end);
    const int n = (int)rd.decode_word();
    if (n < 0 || n > max_out) return DATA_CORRUPT;
    alignas(64) uint8_t mtf[256 + 16] = {0};
    const int max_rank = decode_alphabet(mtf, [&] { return rd.decode_half(); });

    int ctx_rank0 = 0, ctx_rank4 = 0, ctx_run = 0, avg_rank = 0;
    uint8_t rank_hist[256] = {0}, run_hist[256] = {0};
    for (int i = 0; i < n;) {
        const int c = mtf[0];
        int hist = rank_hist[c];
        int state = T.rank_state[(ctx_run << 11) | (ctx_rank4 << 3) | hist];
        int rank = 1;
        if (avg_rank < 32) {
            if (undecide<RANK_FIRST, ADAPT>(rd, T, K.rT_state[state], K.rT_chr[c], K.rT_stat, ADAPT ? &M->rank[c] : nullptr)) {
                int bits = 1;
                while (bits != max_rank &&
                       undecide<RANK_EXP, ADAPT>(rd, T, K.rE_state[state][bits - 1], K.rE_chr[c][bits - 1], K.rE_stat[bits - 1],
                                                 ADAPT ? &M->rank_exp[hist > bits ? hist : bits][bits] : nullptr))
                    ++bits;
                rank_hist[c] = (uint8_t)bits;
                short* ms = K.rM_state[bits][state]; short* mc = K.rM_chr[bits][c]; short* mp = K.rM_stat[bits];
                for (int b = bits - 1; b >= 0; --b)
                    rank += rank + (int)undecide<RANK_MANT, ADAPT>(rd, T, ms[rank], mc[rank], mp[rank], ADAPT ? &M->rank_mant[bits] : nullptr);
            } else {
                rank_hist[c] = 0;
            }
        } else {
            short* es = K.rP_state[state]; short* ec = K.rP_chr[c]; short* ep = K.rP_stat;
            rank = 0;
            for (int ctx = 1, b = max_rank; b >= 0; --b) {
                const int v = (int)undecide<RANK_ESC, ADAPT>(rd, T, es[ctx], ec[ctx], ep[ctx], ADAPT ? &M->rank_esc[ctx] : nullptr);
                ctx += ctx + v; rank += rank + v;
            }
            rank_hist[c] = (uint8_t)bsr32((unsigned)rank);
        }
        if (rank < 1 || rank > 255) return DATA_CORRUPT;      // 0 only from a damaged stream (escape code): it would index the state tables at -1
        requeue(mtf, rank, (uint8_t)c);

        avg_rank = (avg_rank * 124 + rank * 4) >> 7;
        rank -= 1;
        hist = run_hist[c];
        state = T.run_state[(ctx_rank0 << 10) | (ctx_run << 6) | ((rank < 7 ? rank : 7) << 3) | (hist < 7 ? hist : 7)];
        int run = 1;
        if (undecide<RUN_FIRST, ADAPT>(rd, T, K.nT_state[state], K.nT_chr[c], K.nT_stat, ADAPT ? &M->run[c] : nullptr)) {
            int bits = 1;
            while (bits < 31 &&
                   undecide<RUN_EXP, ADAPT>(rd, T, K.nE_state[state][bits - 1], K.nE_chr[c][bits - 1], K.nE_stat[bits - 1],
                                            ADAPT ? &M->run_exp[hist > bits ? hist : bits][bits] : nullptr))
                ++bits;
            run_hist[c] = (uint8_t)((run_hist[c] + 3 * bits + 3) >> 2);
            short* ms = K.nM_state[bits][state]; short* mc = K.nM_chr[bits][c]; short* mp = K.nM_stat[bits];
            for (int ctx = 1, b = bits - 1; b >= 0; --b) {
                const int v = (int)undecide<RUN_MANT, ADAPT>(rd, T, ms[ctx], mc[ctx], mp[ctx], ADAPT ? &M->run_mant[bits] : nullptr);
                run += run + v;
                ctx = (bits <= 5) ? (ctx + ctx + v) : (ctx + 1);
            }
        } else {
            run_hist[c] = (uint8_t)((run_hist[c] + 2) >> 2);
        }
        ctx_rank0 = ((ctx_rank0 << 1) | (rank == 0 ? 1 : 0)) & 0x7;
        ctx_rank4 = ((ctx_rank4 << 2) | (rank < 3 ? rank : 3)) & 0xff;
        ctx_run   = ((ctx_run   << 1) | (run < 3 ? 1 : 0)) & 0xf;
        if (run <= 0 || run > n - i) return DATA_CORRUPT;
        memset(out + i, c, (size_t)run);
        i += run;
    }
    return n;
}

static int decode_model2(const uint8_t* in, const uint8_t* in_end, uint8_t* out, int max_out)
{
    std::unique_ptr<Counters2> Cn(new Counters2);
    fill_shorts(Cn->r_exp, sizeof(Cn->r_exp) + sizeof(Cn->r_mant), 4096);
    fill_shorts(Cn->n_exp, sizeof(Cn->n_exp) + sizeof(Cn->n_mant), 1024);
    Counters2& K = *Cn;
    RangeDecoder rd; rd.init(in, in_end);
    const int n = (int)rd.decode_word();
    if (n < 0 || n > max_out) return DATA_CORRUPT;
    alignas(64) uint8_t mtf[256 + 16] = {0};
    decode_alphabet(mtf, [&] { return rd.decode<1>(1); });
    for (int i = 0; i < n;) {
        const int c = mtf[0];
        int rank = 1;
        {
            short* e = K.r_exp[c];
            const int p = e[0];
            if (rd.decode<13>(p)) {
                nudge<4>(e[0], 83);
                int bits = 1;
                for (;;) {
                    if (bits == 7) break;
                    const int q = e[bits];
                    if (rd.decode<13>(q)) { nudge<4>(e[bits], 122); ++bits; } else { nudge<4>(e[bits], 8114); break; }
                }
                short* mt = K.r_mant[c][bits];
                for (int b = bits - 1; b >= 0; --b) {
                    const int q = mt[rank];
                    const unsigned v = rd.decode<13>(q);
                    nudge<7>(mt[rank], v ? 235 : 7999);
                    rank += rank + (int)v;
                }
            } else nudge<4>(e[0], 8016);
        }
        if (rank > 255) return DATA_CORRUPT;
        requeue(mtf, rank, (uint8_t)c);
        int run = 1;
        {
            short* e = K.n_exp[c];
            const int p = e[0];
            if (rd.decode<11>(p)) {
                nudge<5>(e[0], 42);
                int bits = 1;
                for (;;) {
                    if (bits >= 31) break;
                    const int q = e[bits];
                    if (rd.decode<11>(q)) { nudge<4>(e[bits], 142); ++bits; } else { nudge<4>(e[bits], 1962); break; }
                }
                short* mt = K.n_mant[c][bits];
                for (int ctx = 1, b = bits - 1; b >= 0; --b) {
                    const int q = mt[ctx];
                    const unsigned v = rd.decode<11>(q);
                    if (bits <= 5) { nudge<6>(mt[ctx], v ? 147 : 1951); ctx += ctx + (int)v; }
                    else           { nudge<5>(mt[ctx], v ? 46 : 1987);  ctx += 1; }
                    run += run + (int)v;
                }
            } else nudge<5>(e[0], 2025);
        }
        if (run <= 0 || run > n - i) return DATA_CORRUPT;
        memset(out + i, c, (size_t)run);
        i += run;
    }
    return n;
}

int qlfc_decode_block_bounded(const uint8_t* in, long long in_size, uint8_t* out, int coder, int max_out)
{
    if (in_size < 0) return DATA_CORRUPT;
    const uint8_t* in_end = (in_size >= (long long)1 << 40) ? (const uint8_t*)UINTPTR_MAX : in + in_size;
    switch (coder) {
        case CODER_STATIC:   return decode_model1<false>(in, in_end, out, max_out);
        case CODER_ADAPTIVE: return decode_model1<true>(in, in_end, out, max_out);
        case CODER_FAST:     return decode_model2(in, in_end, out, max_out);
    }
    return BAD_PARAMETER;
}
// the reference's entry point carries no input size (qlfc.h:58): the caller vouches for the stream
int qlfc_decode_block(const uint8_t* in, uint8_t* out, int coder) { return qlfc_decode_block_bounded(in, UNBOUNDED_INPUT, out, coder, 0x7fffffff); }

// The static coder's back half alone: the probabilities come from the GPU (devcoder.hip), in stream order, with the first
// decision of every run marked so that the output-budget test sits where the reference has it (qlfc.cpp:894).
int qlfc_encode_static_pstream(const uint8_t* first_seen, int nsym, int in_size, const uint16_t* ps, size_t count, uint8_t* out, int out_size)
{
    if (in_size <= 0 || nsym <= 0) return BAD_PARAMETER;
    RunView H; H.nsym = nsym; memcpy(H.first_seen, first_seen, (size_t)nsym);
    RangeEncoder rc;
    rc.init(out, out_size);
    rc.encode_word((uint32_t)in_size);
    (void)encode_alphabet(H, [&](unsigned b) { rc.encode_half(b); });
    RangeEncoder::Live L = rc.enter();
    unsigned is_full = (unsigned)rc.full();                             // full() looks at the output cursor only: kept current by encode_live_f
    for (size_t i = 0; i < count; ++i) {
        const unsigned x = ps[i];
        // one test that practically never fires instead of a branch on the run-start mark itself (which is set on every third or fourth
        // decision, irregularly: mispredicted, it cost this loop more than its arithmetic)
        if (__builtin_expect(((x >> 13) & is_full) != 0u, 0)) return NOT_COMPRESSIBLE;
        rc.encode_live_f<12>(L, (x >> 12) & 1u, (int)(x & 0xfffu), is_full);
    }
    rc.leave(L);
    return rc.finish();
}

void qlfc_encode_static_pstream_pair(const PstreamJob& A, const PstreamJob& B, int* resA, int* resB)
{
    RunView HA, HB;
    HA.nsym = A.nsym; memcpy(HA.first_seen, A.first_seen, (size_t)A.nsym);
    HB.nsym = B.nsym; memcpy(HB.first_seen, B.first_seen, (size_t)B.nsym);
    RangeEncoder ra, rb;
    ra.init(A.out, A.out_size); rb.init(B.out, B.out_size);
    ra.encode_word((uint32_t)A.in_size); rb.encode_word((uint32_t)B.in_size);
    (void)encode_alphabet(HA, [&](unsigned b) { ra.encode_half(b); });
    (void)encode_alphabet(HB, [&](unsigned b) { rb.encode_half(b); });
    RangeEncoder::Live La = ra.enter(), Lb = rb.enter();
    const uint16_t* pa = A.ps; const uint16_t* pb = B.ps;
    const size_t both = A.count < B.count ? A.count : B.count;
    size_t i = 0;
    bool fa = false, fb = false;                              // a stream ran out of budget (reference: NOT_COMPRESSIBLE at a run start)
    unsigned fulla = (unsigned)ra.full(), fullb = (unsigned)rb.full();     // kept current by encode_live_f
    for (; i < both; ++i) {
        const unsigned x = pa[i], y = pb[i];
        // branch-free test (the budget check belongs to run starts only, qlfc.cpp:894; it practically never fires)
        const unsigned stop = ((x >> 13) & fulla) | ((y >> 13) & fullb);
        if (__builtin_expect(stop != 0, 0)) {
            if ((x & 0x2000u) && fulla) { fa = true; break; }
            fb = true; break;
        }
        ra.encode_live_f<12>(La, (x >> 12) & 1u, (int)(x & 0xfffu), fulla);
        rb.encode_live_f<12>(Lb, (y >> 12) & 1u, (int)(y & 0xfffu), fullb);
    }
    // what is left of either stream, singly
    if (!fa) for (size_t k = i; k < A.count; ++k) { const unsigned x = pa[k]; if ((x & 0x2000u) && fulla) { fa = true; break; } ra.encode_live_f<12>(La, (x >> 12) & 1u, (int)(x & 0xfffu), fulla); }
    if (!fb) for (size_t k = i; k < B.count; ++k) { const unsigned y = pb[k]; if ((y & 0x2000u) && fullb) { fb = true; break; } rb.encode_live_f<12>(Lb, (y >> 12) & 1u, (int)(y & 0xfffu), fullb); }
    ra.leave(La); rb.leave(Lb);
    *resA = fa ? NOT_COMPRESSIBLE : ra.finish();
    *resB = fb ? NOT_COMPRESSIBLE : rb.finish();
}

// ---- the packed stream (round 6; devcoder.hip DcP13): 13 bits per decision, eight decisions in 13 bytes -------------------------------
// field i at bits [13 i, 13 i + 13) of the sub-block's stream, little endian: {probability[11:0], coded bit}.  There is no run-start mark:
// the reference tests its output budget at run starts only (qlfc.cpp:894), here a stream whose budget is reached at ANY decision gives up
// (NOT_COMPRESSIBLE), and the caller redoes the block on the host model from the run arrays — which is what it does for every sub-block that
// does not compress, and which reproduces the reference's decision exactly.  (A 4-byte read at the last field reaches 3 bytes past the
// stream: the landing zones have that slack.)
static inline unsigned p13_get(const uint8_t* b, size_t i)
{
    const size_t bit = i * 13u;
    uint32_t w; memcpy(&w, b + (bit >> 3), 4);
    return (w >> (bit & 7u)) & 0x1fffu;
}
void qlfc_pack_p13(const uint16_t* ps, size_t count, uint8_t* out)
{
    const size_t bytes = (count + 7) / 8 * 13;
    memset(out, 0, bytes);
    for (size_t i = 0; i < count; ++i) {
        const size_t bit = i * 13u;
        const uint32_t f = ((uint32_t)ps[i] & 0x1fffu) << (bit & 7u);
        out[bit >> 3] |= (uint8_t)f; out[(bit >> 3) + 1] |= (uint8_t)(f >> 8); if (f >> 16) out[(bit >> 3) + 2] |= (uint8_t)(f >> 16);
    }
}
int qlfc_encode_static_p13(const uint8_t* first_seen, int nsym, int in_size, const uint8_t* ps, size_t count, uint8_t* out, int out_size)
{
    if (in_size <= 0 || nsym <= 0) return BAD_PARAMETER;
    RunView H; H.nsym = nsym; memcpy(H.first_seen, first_seen, (size_t)nsym);
    RangeEncoder rc;
    rc.init(out, out_size);
    rc.encode_word((uint32_t)in_size);
    (void)encode_alphabet(H, [&](unsigned b) { rc.encode_half(b); });
    RangeEncoder::Live L = rc.enter();
    unsigned is_full = (unsigned)rc.full();
    for (size_t i = 0; i < count; ++i) {
        if (__builtin_expect(is_full != 0u, 0)) return NOT_COMPRESSIBLE;
        const unsigned x = p13_get(ps, i);
        rc.encode_live_f<12>(L, x >> 12, (int)(x & 0xfffu), is_full);
    }
    rc.leave(L);
    return rc.finish();
}
void qlfc_encode_static_p13_pair(const PstreamJob& A, const PstreamJob& B, int* resA, int* resB)
{
    RunView HA, HB;
    HA.nsym = A.nsym; memcpy(HA.first_seen, A.first_seen, (size_t)A.nsym);
    HB.nsym = B.nsym; memcpy(HB.first_seen, B.first_seen, (size_t)B.nsym);
    RangeEncoder ra, rb;
    ra.init(A.out, A.out_size); rb.init(B.out, B.out_size);
    ra.encode_word((uint32_t)A.in_size); rb.encode_word((uint32_t)B.in_size);
    (void)encode_alphabet(HA, [&](unsigned b) { ra.encode_half(b); });
    (void)encode_alphabet(HB, [&](unsigned b) { rb.encode_half(b); });
    RangeEncoder::Live La = ra.enter(), Lb = rb.enter();
    const uint8_t* pa = (const uint8_t*)A.ps; const uint8_t* pb = (const uint8_t*)B.ps;
    const size_t both = A.count < B.count ? A.count : B.count;
    size_t i = 0;
    bool fa = false, fb = false;
    unsigned fulla = (unsigned)ra.full(), fullb = (unsigned)rb.full();
    for (; i < both; ++i) {
        if (__builtin_expect((fulla | fullb) != 0u, 0)) { if (fulla) fa = true; else fb = true; break; }
        const unsigned x = p13_get(pa, i), y = p13_get(pb, i);
        ra.encode_live_f<12>(La, x >> 12, (int)(x & 0xfffu), fulla);
        rb.encode_live_f<12>(Lb, y >> 12, (int)(y & 0xfffu), fullb);
    }
    if (!fa) for (size_t k = i; k < A.count; ++k) { if (fulla) { fa = true; break; } const unsigned x = p13_get(pa, k); ra.encode_live_f<12>(La, x >> 12, (int)(x & 0xfffu), fulla); }
    if (!fb) for (size_t k = i; k < B.count; ++k) { if (fullb) { fb = true; break; } const unsigned y = p13_get(pb, k); rb.encode_live_f<12>(Lb, y >> 12, (int)(y & 0xfffu), fullb); }
    ra.leave(La); rb.leave(Lb);
    *resA = fa ? NOT_COMPRESSIBLE : ra.finish();
    *resB = fb ? NOT_COMPRESSIBLE : rb.finish();
}

// The fast coder (-e0) behind the device model: one counter per decision, so an entry IS the probability; what differs from the
// static coder's stream is the precision, which follows the side of the run the decision belongs to (bit 15), and the alphabet header,
// whose bits go out at precision 1 (qlfc.cpp:1174).  The budget test sits on the run-start mark as in encode_model2 (qlfc.cpp:1191).
static inline unsigned psf_prec(unsigned x) { return 13u - ((x >> 15) << 1); }
int qlfc_encode_fast_pstream(const uint8_t* first_seen, int nsym, int in_size, const uint16_t* ps, size_t count, uint8_t* out, int out_size)
{
    if (in_size <= 0 || nsym <= 0) return BAD_PARAMETER;
    RunView H; H.nsym = nsym; memcpy(H.first_seen, first_seen, (size_t)nsym);
    RangeEncoder rc;
    rc.init(out, out_size);
    rc.encode_word((uint32_t)in_size);
    (void)encode_alphabet(H, [&](unsigned b) { rc.encode<1>(b, 1); });
    RangeEncoder::Live L = rc.enter();
    unsigned is_full = (unsigned)rc.full();
    for (size_t i = 0; i < count; ++i) {
        const unsigned x = ps[i];
        if (__builtin_expect(((x >> 14) & is_full) != 0u, 0)) return NOT_COMPRESSIBLE;
        rc.encode_live_var(L, (x >> 13) & 1u, x & 0x1fffu, psf_prec(x), is_full);
    }
    rc.leave(L);
    return rc.finish();
}

void qlfc_encode_fast_pstream_pair(const PstreamJob& A, const PstreamJob& B, int* resA, int* resB)
{
    RunView HA, HB;
    HA.nsym = A.nsym; memcpy(HA.first_seen, A.first_seen, (size_t)A.nsym);
    HB.nsym = B.nsym; memcpy(HB.first_seen, B.first_seen, (size_t)B.nsym);
    RangeEncoder ra, rb;
    ra.init(A.out, A.out_size); rb.init(B.out, B.out_size);
    ra.encode_word((uint32_t)A.in_size); rb.encode_word((uint32_t)B.in_size);
    (void)encode_alphabet(HA, [&](unsigned b) { ra.encode<1>(b, 1); });
    (void)encode_alphabet(HB, [&](unsigned b) { rb.encode<1>(b, 1); });
    RangeEncoder::Live La = ra.enter(), Lb = rb.enter();
    const uint16_t* pa = A.ps; const uint16_t* pb = B.ps;
    const size_t both = A.count < B.count ? A.count : B.count;
    size_t i = 0;
    bool fa = false, fb = false;
    unsigned fulla = (unsigned)ra.full(), fullb = (unsigned)rb.full();
    for (; i < both; ++i) {
        const unsigned x = pa[i], y = pb[i];
        const unsigned stop = ((x >> 14) & fulla) | ((y >> 14) & fullb);
        if (__builtin_expect((stop & 1u) != 0, 0)) {
            if ((x & 0x4000u) && fulla) { fa = true; break; }
            fb = true; break;
        }
        ra.encode_live_var(La, (x >> 13) & 1u, x & 0x1fffu, psf_prec(x), fulla);
        rb.encode_live_var(Lb, (y >> 13) & 1u, y & 0x1fffu, psf_prec(y), fullb);
    }
    if (!fa) for (size_t k = i; k < A.count; ++k) { const unsigned x = pa[k]; if ((x & 0x4000u) && fulla) { fa = true; break; } ra.encode_live_var(La, (x >> 13) & 1u, x & 0x1fffu, psf_prec(x), fulla); }
    if (!fb) for (size_t k = i; k < B.count; ++k) { const unsigned y = pb[k]; if ((y & 0x4000u) && fullb) { fb = true; break; } rb.encode_live_var(Lb, (y >> 13) & 1u, y & 0x1fffu, psf_prec(y), fullb); }
    ra.leave(La); rb.leave(Lb);
    *resA = fa ? NOT_COMPRESSIBLE : ra.finish();
    *resB = fb ? NOT_COMPRESSIBLE : rb.finish();
}

// ------------------------------------------------------------------------------------------------
// Eight sub-blocks of a device-model block, one per 32-bit lane of AVX2 registers.  The scalar loop above retires a decision
// in ~4.7 cycles however many streams are interleaved (4-way measured = 2-way: it is bound by instruction throughput, ~14
// micro-ops per decision); here one step codes a decision of EACH stream in ~38 micro-ops with no branch.
// range and the low 32 bits of low are vectors, bit 32 of low (the pending carry) a vector of 0 / 1.  A stream is renormalised
// exactly where the scalar coder would (range < 2^16 in front of a decision), by blends; what its shift() would have handed to
// the output — the top 16 bits of low and the carry — goes as one 32-bit record {lane, carry, unit} into a log (records of the
// lanes that renormalise in a step are left-packed with a permutation looked up by the 8-bit lane mask and stored with one
// unaligned store).  Every CHUNK steps the log is replayed through the lanes' scalar carry caches (RangeEncoder::emit_unit),
// the only code that touches the output: ~1 record per 30 decisions.
// The run-start budget test (qlfc.cpp:894) is not made per run: the output position only moves in emit_unit, so if it never
// reaches the limit there the test would never have fired; if it does, the function gives up (returns false, nothing
// about the outputs is defined) and the caller runs the scalar coders, which reproduce the reference's decision exactly.
// Measured on the EPYC 9575F in rounds 2-4 (CPU-seconds per 64 MiB block in the loaded pool, framing included; round 5's figures of the
// loops alone are at x8_steps_avx512): pairs 0.238, this 0.144; a four-lane SSE version
// of the same step (two tasks per block) 0.221 — a step costs about the same micro-ops whatever its width, so only the full
// eight lanes pay (that version was removed again).
// ------------------------------------------------------------------------------------------------
#if defined(__AVX2__)
struct PackLut { alignas(32) uint32_t idx[256][8]; };
static const PackLut& pack_lut()
{
    static const PackLut L = [] {
        PackLut t;
        for (int m = 0; m < 256; ++m) { int k = 0; for (int l = 0; l < 8; ++l) if (m & (1 << l)) t.idx[m][k++] = (uint32_t)l; for (; k < 8; ++k) t.idx[m][k] = 0; }
        return t;
    }();
    return L;
}

struct alignas(32) X8State { uint32_t R[8], LO[8], CY[8]; };

// 8 entries of each of the eight streams -> 8 vectors of one entry per stream (8 x 8 transpose of 16-bit words)
#define BSC_X8_TRANSPOSE(ps, i)                                                                                                   \
    const __m128i a0 = _mm_loadu_si128((const __m128i*)(ps[0] + i)), a1 = _mm_loadu_si128((const __m128i*)(ps[1] + i));            \
    const __m128i a2 = _mm_loadu_si128((const __m128i*)(ps[2] + i)), a3 = _mm_loadu_si128((const __m128i*)(ps[3] + i));            \
    const __m128i a4 = _mm_loadu_si128((const __m128i*)(ps[4] + i)), a5 = _mm_loadu_si128((const __m128i*)(ps[5] + i));            \
    const __m128i a6 = _mm_loadu_si128((const __m128i*)(ps[6] + i)), a7 = _mm_loadu_si128((const __m128i*)(ps[7] + i));            \
    const __m128i b0 = _mm_unpacklo_epi16(a0, a1), b1 = _mm_unpackhi_epi16(a0, a1), b2 = _mm_unpacklo_epi16(a2, a3), b3 = _mm_unpackhi_epi16(a2, a3); \
    const __m128i b4 = _mm_unpacklo_epi16(a4, a5), b5 = _mm_unpackhi_epi16(a4, a5), b6 = _mm_unpacklo_epi16(a6, a7), b7 = _mm_unpackhi_epi16(a6, a7); \
    const __m128i c0 = _mm_unpacklo_epi32(b0, b2), c1 = _mm_unpackhi_epi32(b0, b2), c2 = _mm_unpacklo_epi32(b1, b3), c3 = _mm_unpackhi_epi32(b1, b3); \
    const __m128i c4 = _mm_unpacklo_epi32(b4, b6), c5 = _mm_unpackhi_epi32(b4, b6), c6 = _mm_unpacklo_epi32(b5, b7), c7 = _mm_unpackhi_epi32(b5, b7); \
    const __m128i t0 = _mm_unpacklo_epi64(c0, c4), t1 = _mm_unpackhi_epi64(c0, c4), t2 = _mm_unpacklo_epi64(c1, c5), t3 = _mm_unpackhi_epi64(c1, c5); \
    const __m128i t4 = _mm_unpacklo_epi64(c2, c6), t5 = _mm_unpackhi_epi64(c2, c6), t6 = _mm_unpacklo_epi64(c3, c7), t7 = _mm_unpackhi_epi64(c3, c7)

// The packed stream (13 bits per decision): 8 decisions of a stream are 13 bytes; a 16-byte load, a byte shuffle that puts the 3-4 bytes of
// every field into a 32-bit lane and a per-lane shift give r_l = the 8 fields of stream l (with bits of the neighbouring field above bit
// 12, which the static coder's steps never look at: they mask the probability with 0xfff and test bit 12), then an 8 x 8 transpose of
// 32-bit words.  3 operations per stream and 24 for the transpose against 1 load, 24 + 8 widenings for the 16-bit entries.
#define BSC_X8_LOAD13(ps, i, l) _mm256_srlv_epi32(_mm256_shuffle_epi8(_mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)((const uint8_t*)(ps)[l] + ((i) >> 3) * 13))), shuf13), shift13)
#define BSC_X8_TRANSPOSE13(ps, i)                                                                                                 \
    const __m256i r0 = BSC_X8_LOAD13(ps, i, 0), r1 = BSC_X8_LOAD13(ps, i, 1), r2 = BSC_X8_LOAD13(ps, i, 2), r3 = BSC_X8_LOAD13(ps, i, 3); \
    const __m256i r4 = BSC_X8_LOAD13(ps, i, 4), r5 = BSC_X8_LOAD13(ps, i, 5), r6 = BSC_X8_LOAD13(ps, i, 6), r7 = BSC_X8_LOAD13(ps, i, 7); \
    const __m256i u0 = _mm256_unpacklo_epi32(r0, r1), u1 = _mm256_unpackhi_epi32(r0, r1), u2 = _mm256_unpacklo_epi32(r2, r3), u3 = _mm256_unpackhi_epi32(r2, r3); \
    const __m256i u4 = _mm256_unpacklo_epi32(r4, r5), u5 = _mm256_unpackhi_epi32(r4, r5), u6 = _mm256_unpacklo_epi32(r6, r7), u7 = _mm256_unpackhi_epi32(r6, r7); \
    const __m256i v0 = _mm256_unpacklo_epi64(u0, u2), v1 = _mm256_unpackhi_epi64(u0, u2), v2 = _mm256_unpacklo_epi64(u1, u3), v3 = _mm256_unpackhi_epi64(u1, u3); \
    const __m256i v4 = _mm256_unpacklo_epi64(u4, u6), v5 = _mm256_unpackhi_epi64(u4, u6), v6 = _mm256_unpacklo_epi64(u5, u7), v7 = _mm256_unpackhi_epi64(u5, u7); \
    const __m256i w0 = _mm256_permute2x128_si256(v0, v4, 0x20), w1 = _mm256_permute2x128_si256(v1, v5, 0x20), w2 = _mm256_permute2x128_si256(v2, v6, 0x20), w3 = _mm256_permute2x128_si256(v3, v7, 0x20); \
    const __m256i w4 = _mm256_permute2x128_si256(v0, v4, 0x31), w5 = _mm256_permute2x128_si256(v1, v5, 0x31), w6 = _mm256_permute2x128_si256(v2, v6, 0x31), w7 = _mm256_permute2x128_si256(v3, v7, 0x31)
#define BSC_X8_CONST13                                                                                                            \
    const __m256i shuf13 = _mm256_setr_epi8(0, 1, 2, 3, 1, 2, 3, 4, 3, 4, 5, 6, 4, 5, 6, 7, 6, 7, 8, 9, 8, 9, 10, 11, 9, 10, 11, 12, 11, 12, 13, 14); \
    const __m256i shift13 = _mm256_setr_epi32(0, 5, 2, 7, 4, 1, 6, 3)
#define BSC_X8_PREFETCH13(ps, i, pf) do {                                                                                          \
        if (pf) {                                                                                                                  \
            const unsigned l2 = ((unsigned)((i) >> 3) & 3u) * 2u;                                                                  \
            _mm_prefetch((const char*)(ps)[l2] + (((i) + (pf)) >> 3) * 13, _MM_HINT_T0);                                             \
            _mm_prefetch((const char*)(ps)[l2 + 1] + (((i) + (pf)) >> 3) * 13, _MM_HINT_T0);                                         \
        }                                                                                                                          \
    } while (0)

// A step reads 16 bytes of each stream, i.e. every stream crosses a cache line every fourth step and a 4 KiB page every 256th; the
// entries were written by the GPU's DMA engine, so every line comes from DRAM.  Two of the eight streams per step get a software
// prefetch `pf` entries ahead (each stream one per line): it runs across page boundaries, where the hardware stream prefetchers stop.
// pf = 0: none.  (A prefetch past the end of a stream is harmless: it does not fault.)
#define BSC_X8_PREFETCH(ps, i, pf) do {                                                                                            \
        if (pf) {                                                                                                                  \
            const unsigned l2 = ((unsigned)((i) >> 3) & 3u) * 2u;                                                                  \
            _mm_prefetch((const char*)((ps)[l2] + (i) + (pf)), _MM_HINT_T0);                                                        \
            _mm_prefetch((const char*)((ps)[l2 + 1] + (i) + (pf)), _MM_HINT_T0);                                                    \
        }                                                                                                                          \
    } while (0)

// steps [i, end) (end - i a multiple of 8) of all eight streams; appends the renormalisation records, returns the log's new end
// FAST: entries of the fast coder (13-bit value, bit at 13, precision 13 - 2 * bit 15: a per-lane shift count instead of the constant 12)
template <bool FAST, bool P13 = false>
static uint32_t* x8_steps_avx2(X8State& S, const uint16_t* const* ps, size_t i, size_t end, uint32_t* logp, size_t pf)
{
    static_assert(!(FAST && P13), "the packed stream is the static coder's");
    BSC_X8_CONST13;
    __m256i R = _mm256_load_si256((const __m256i*)S.R), LO = _mm256_load_si256((const __m256i*)S.LO), CY = _mm256_load_si256((const __m256i*)S.CY);
    const __m256i zero = _mm256_setzero_si256(), m12 = _mm256_set1_epi32(FAST ? 0x1fff : 0xfff), one = _mm256_set1_epi32(1);
    const __m256i c13 = _mm256_set1_epi32(13);
    const __m256i lane_id = _mm256_setr_epi32(0 << 17, 1 << 17, 2 << 17, 3 << 17, 4 << 17, 5 << 17, 6 << 17, 7 << 17);
    const PackLut& lut = pack_lut();
    // one decision of every stream; x = the eight 16-bit entries, zero-extended
    auto step = [&](const __m256i x) __attribute__((always_inline)) {
        const __m256i need = _mm256_cmpeq_epi32(_mm256_srli_epi32(R, 16), zero);                        // range < 2^16
        const unsigned mk = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(need));
        const __m256i rec = _mm256_or_si256(_mm256_or_si256(_mm256_srli_epi32(LO, 16), _mm256_slli_epi32(CY, 16)), lane_id);
        _mm256_storeu_si256((__m256i*)logp, _mm256_permutevar8x32_epi32(rec, _mm256_load_si256((const __m256i*)lut.idx[mk])));
        logp += __builtin_popcount(mk);
        LO = _mm256_blendv_epi8(LO, _mm256_slli_epi32(LO, 16), need);
        CY = _mm256_andnot_si256(need, CY);
        R  = _mm256_blendv_epi8(R, _mm256_slli_epi32(R, 16), need);
        const __m256i p = _mm256_and_si256(x, m12);
        const __m256i m = _mm256_sub_epi32(zero, _mm256_and_si256(_mm256_srli_epi32(x, FAST ? 13 : 12), one));      // all ones where the bit is 1
        const __m256i r = FAST ? _mm256_mullo_epi32(_mm256_srlv_epi32(R, _mm256_sub_epi32(c13, _mm256_slli_epi32(_mm256_srli_epi32(x, 15), 1))), p)
                               : _mm256_mullo_epi32(_mm256_srli_epi32(R, 12), p);
        const __m256i add = _mm256_and_si256(r, m);
        const __m256i lo2 = _mm256_add_epi32(LO, add);
        const __m256i ge = _mm256_cmpeq_epi32(_mm256_max_epu32(lo2, add), lo2);                        // all ones where lo2 >= add: no carry out
        CY = _mm256_add_epi32(CY, _mm256_andnot_si256(ge, one));
        LO = lo2;
        R = _mm256_add_epi32(r, _mm256_and_si256(m, _mm256_sub_epi32(_mm256_sub_epi32(R, r), r)));
    };
    if (P13) {
        for (; i < end; i += 8) {
            BSC_X8_PREFETCH13(ps, i, pf);
            BSC_X8_TRANSPOSE13(ps, i);
            step(w0); step(w1); step(w2); step(w3); step(w4); step(w5); step(w6); step(w7);
        }
    } else
    for (; i < end; i += 8) {
        BSC_X8_PREFETCH(ps, i, pf);
        BSC_X8_TRANSPOSE(ps, i);
        step(_mm256_cvtepu16_epi32(t0)); step(_mm256_cvtepu16_epi32(t1)); step(_mm256_cvtepu16_epi32(t2)); step(_mm256_cvtepu16_epi32(t3));
        step(_mm256_cvtepu16_epi32(t4)); step(_mm256_cvtepu16_epi32(t5)); step(_mm256_cvtepu16_epi32(t6)); step(_mm256_cvtepu16_epi32(t7));
    }
    _mm256_store_si256((__m256i*)S.R, R); _mm256_store_si256((__m256i*)S.LO, LO); _mm256_store_si256((__m256i*)S.CY, CY);
    return logp;
}

// The same step with AVX-512VL on 256-bit vectors (chosen at run time): compares write mask registers, the renormalisation and the
// two directions of the update are masked shifts / adds / subtracts, and the records are left-packed by vpcompressd: ~24
// micro-ops per step instead of ~45.
// VSEL: 0 the round-4 step (BSC_RC_VSEL=0), 2 round 5's
template <bool FAST, int VSEL, bool P13 = false>
__attribute__((target("avx512f,avx512vl")))
static uint32_t* x8_steps_avx512(X8State& S, const uint16_t* const* ps, size_t i, size_t end, uint32_t* logp, size_t pf)
{
    static_assert(!(FAST && P13), "the packed stream is the static coder's");
    BSC_X8_CONST13;
    __m256i R = _mm256_load_si256((const __m256i*)S.R), LO = _mm256_load_si256((const __m256i*)S.LO), CY = _mm256_load_si256((const __m256i*)S.CY);
    const __m256i m12 = _mm256_set1_epi32(FAST ? 0x1fff : 0xfff), one = _mm256_set1_epi32(1), lim = _mm256_set1_epi32(0x10000), b12 = _mm256_set1_epi32(FAST ? 0x2000 : 0x1000);
    const __m256i c13 = _mm256_set1_epi32(13), c16 = _mm256_set1_epi32(16), zero = _mm256_setzero_si256();
    const __m256i lane_id = _mm256_setr_epi32(0 << 17, 1 << 17, 2 << 17, 3 << 17, 4 << 17, 5 << 17, 6 << 17, 7 << 17);
    // (a macro, not a lambda: a lambda does not inherit the function's target attribute)
#define BSC_X8_STEP512(xv) do {                                                                                                    \
        const __m256i x = (xv);                                                                                                    \
        const __mmask8 need = _mm256_cmplt_epu32_mask(R, lim);                                         /* range < 2^16 */          \
        const __m256i rec = _mm256_ternarylogic_epi32(_mm256_srli_epi32(LO, 16), _mm256_slli_epi32(CY, 16), lane_id, 0xfe);         \
        _mm256_storeu_si256((__m256i*)logp, _mm256_maskz_compress_epi32(need, rec));                                               \
        logp += __builtin_popcount((unsigned)need);                                                                                \
        LO = _mm256_mask_slli_epi32(LO, need, LO, 16);                                                                             \
        CY = _mm256_maskz_mov_epi32((__mmask8)~need, CY);                                                                          \
        /* the step is bound by the latency of range -> compare -> shift -> multiply -> subtract: both products are started at   */ \
        /* once (range >> 12 and, for a renormalised lane, (range << 16) >> 12 = range << 4) and the compare only selects          */ \
        const __m256i p = _mm256_and_si256(x, m12);                                                                                \
        /* FAST: precision sh = 13 - 2 * (bit 15) per lane; a renormalised lane's (range << 16) >> sh = range << (16 - sh) */       \
        const __m256i sh = _mm256_sub_epi32(c13, _mm256_slli_epi32(_mm256_srli_epi32(x, 15), 1));                                  \
        const __m256i ra = FAST ? _mm256_mullo_epi32(_mm256_srlv_epi32(R, sh), p) : _mm256_mullo_epi32(_mm256_srli_epi32(R, 12), p); \
        const __m256i rb = FAST ? _mm256_mullo_epi32(_mm256_sllv_epi32(R, _mm256_sub_epi32(c16, sh)), p) : _mm256_mullo_epi32(_mm256_slli_epi32(R, 4), p); \
        R  = _mm256_mask_slli_epi32(R, need, R, 16);                                                                               \
        const __mmask8 kb = _mm256_test_epi32_mask(x, b12);                                            /* the coded bit */         \
        const __m256i r = _mm256_mask_mov_epi32(ra, need, rb);                                                                     \
        const __m256i lo2 = _mm256_mask_add_epi32(LO, kb, LO, r);                                                                  \
        CY = _mm256_mask_add_epi32(CY, _mm256_cmplt_epu32_mask(lo2, LO), CY, one);                     /* wrapped: carry out */    \
        LO = lo2;                                                                                                                  \
        R = _mm256_mask_sub_epi32(r, kb, R, r);                                                        /* bit ? range - r : r */   \
    } while (0)
    /* Round 5's step.  Measured on the EPYC 9575F hosts (tools/rc_host_bench.cpp, profiles/r05/host_coder_on_box_cpu.txt) the round-4 step */
    /* takes 17.6 cycles: the chain range -> compare into a mask register -> masked shift -> multiply -> subtract -> masked move, with     */
    /* two-cycle vector integer operations.  Here (16.4 cycles; the range's chain alone runs in 13)                                         */
    /*  - "range < 2^16" is a VECTOR mask m made by a VEX compare (inline assembly: written with intrinsics the compiler goes through a     */
    /*    mask register again), both products (range >> sh and, renormalised, range << (16 - sh)) start at once and m only selects;         */
    /*  - the coded bit is folded into the multiplier: range' - (range' >> sh) * p = (range' >> sh) * (2^sh - p) + (range' mod 2^sh), so    */
    /*    with q = bit ? 2^sh - p : p the next range is ONE product plus the low bits of range' where the bit is 1 (none for a              */
    /*    renormalised lane: (range << 16) mod 2^sh = 0) — no subtract and no select by the bit behind the multiply;                        */
    /*  - what the low word needs, r = range' - next range where the bit is 1, hangs off that chain.                                        */
    /* Variants that also kept the low word out of the mask registers, or only did the first item, measured the same or slower.            */
#define BSC_X8_STEP512W(xv) do {                                                                                                   \
        const __m256i x = (xv);                                                                                                    \
        __m256i m;                     /* all ones where range < 2^16, as a VECTOR (VEX compare; intrinsics would go through a mask register) */ \
        asm("vpsrld $16, %1, %0\n\tvpcmpeqd %2, %0, %0" : "=&x"(m) : "x"(R), "x"(zero));                                             \
        const __mmask8 need = _mm256_cmplt_epu32_mask(R, lim);                                         /* the same, for the log and the low word */ \
        const __m256i rec = _mm256_ternarylogic_epi32(_mm256_srli_epi32(LO, 16), _mm256_slli_epi32(CY, 16), lane_id, 0xfe);         \
        _mm256_storeu_si256((__m256i*)logp, _mm256_maskz_compress_epi32(need, rec));                                               \
        logp += __builtin_popcount((unsigned)need);                                                                                \
        LO = _mm256_mask_slli_epi32(LO, need, LO, 16);                                                                             \
        CY = _mm256_maskz_mov_epi32((__mmask8)~need, CY);                                                                          \
        /* from the entry alone: p, the bit as a vector, q, and the mask of the low bits that survive where the bit is 1 */          \
        const __m256i p = _mm256_and_si256(x, m12);                                                                                \
        const __m256i bv = _mm256_srai_epi32(_mm256_slli_epi32(x, FAST ? 18 : 19), 31);                                            \
        const __m256i sh = _mm256_sub_epi32(c13, _mm256_slli_epi32(_mm256_srli_epi32(x, 15), 1));                                  \
        const __m256i full = FAST ? _mm256_sllv_epi32(one, sh) : b12;                                  /* 2^sh (static coder: 4096 = the bit's own mask) */ \
        const __m256i q = _mm256_add_epi32(_mm256_xor_si256(p, bv), _mm256_and_si256(bv, _mm256_add_epi32(full, one)));            \
        const __m256i bm = _mm256_and_si256(bv, _mm256_sub_epi32(full, one));                                                      \
        const __m256i ra = FAST ? _mm256_mullo_epi32(_mm256_srlv_epi32(R, sh), q) : _mm256_mullo_epi32(_mm256_srli_epi32(R, 12), q); \
        const __m256i rb = FAST ? _mm256_mullo_epi32(_mm256_sllv_epi32(R, _mm256_sub_epi32(c16, sh)), q) : _mm256_mullo_epi32(_mm256_slli_epi32(R, 4), q); \
        const __m256i Rn = _mm256_ternarylogic_epi32(m, _mm256_slli_epi32(R, 16), R, 0xca);            /* m ? range << 16 : range */ \
        const __m256i keep = _mm256_ternarylogic_epi32(m, R, bm, 0x08);                                /* ~m & range & bm */       \
        R = _mm256_add_epi32(_mm256_ternarylogic_epi32(m, rb, ra, 0xca), keep);                                                    \
        const __mmask8 kb = _mm256_test_epi32_mask(x, b12);                                            /* the coded bit */         \
        const __m256i lo2 = _mm256_mask_add_epi32(LO, kb, LO, _mm256_sub_epi32(Rn, R));                /* bit: low += r = range' - next */ \
        CY = _mm256_mask_add_epi32(CY, _mm256_cmplt_epu32_mask(lo2, LO), CY, one);                     /* wrapped: carry out */    \
        LO = lo2;                                                                                                                  \
    } while (0)
    if (P13) {
        for (; i < end; i += 8) {
            BSC_X8_PREFETCH13(ps, i, pf);
            BSC_X8_TRANSPOSE13(ps, i);
            if (VSEL == 2) { BSC_X8_STEP512W(w0); BSC_X8_STEP512W(w1); BSC_X8_STEP512W(w2); BSC_X8_STEP512W(w3); BSC_X8_STEP512W(w4); BSC_X8_STEP512W(w5); BSC_X8_STEP512W(w6); BSC_X8_STEP512W(w7); }
            else           { BSC_X8_STEP512(w0); BSC_X8_STEP512(w1); BSC_X8_STEP512(w2); BSC_X8_STEP512(w3); BSC_X8_STEP512(w4); BSC_X8_STEP512(w5); BSC_X8_STEP512(w6); BSC_X8_STEP512(w7); }
        }
    } else
    for (; i < end; i += 8) {
        BSC_X8_PREFETCH(ps, i, pf);
        BSC_X8_TRANSPOSE(ps, i);
        if (VSEL == 2) {
            BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t0)); BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t1)); BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t2)); BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t3));
            BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t4)); BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t5)); BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t6)); BSC_X8_STEP512W(_mm256_cvtepu16_epi32(t7));
        } else {
            BSC_X8_STEP512(_mm256_cvtepu16_epi32(t0)); BSC_X8_STEP512(_mm256_cvtepu16_epi32(t1)); BSC_X8_STEP512(_mm256_cvtepu16_epi32(t2)); BSC_X8_STEP512(_mm256_cvtepu16_epi32(t3));
            BSC_X8_STEP512(_mm256_cvtepu16_epi32(t4)); BSC_X8_STEP512(_mm256_cvtepu16_epi32(t5)); BSC_X8_STEP512(_mm256_cvtepu16_epi32(t6)); BSC_X8_STEP512(_mm256_cvtepu16_epi32(t7));
        }
    }
    _mm256_store_si256((__m256i*)S.R, R); _mm256_store_si256((__m256i*)S.LO, LO); _mm256_store_si256((__m256i*)S.CY, CY);
    return logp;
}

// The packed stream on hosts with AVX-512 VBMI (the EPYC 9005 hosts of the MI355X boxes have it): the byte permute across two 512-bit
// registers does the unpacking AND the 8 x 8 transpose at once.  The 16 bytes of streams 0-3 and 4-7 are gathered into two registers
// (8 loads), and ONE vpermt2b per pair of steps puts bytes o_k .. o_k + 3 of every stream's chunk into the 32-bit lane of that stream for
// field k (low half) and field k + 1 (high half); a per-lane shift finishes it: 8 + 4 + 4 + 4 (the high halves) operations per 8 steps
// against 48 for the AVX2 form above and 40 for the 16-bit entries.
// P13 = false: the same for 16-bit entries (static and fast coder): the permute zero-extends every entry into its 32-bit lane, no shift — 8 + 4 + 4
// operations against 8 loads, a 24-operation transpose and 8 widenings.
template <bool FAST, int VSEL, bool P13>
__attribute__((target("avx512f,avx512vl,avx512bw,avx512vbmi")))
static uint32_t* x8_steps_avx512_vbmi(X8State& S, const uint16_t* const* ps, size_t i, size_t end, uint32_t* logp, size_t pf)
{
    static_assert(!(FAST && P13), "the packed stream is the static coder's");
    __m256i R = _mm256_load_si256((const __m256i*)S.R), LO = _mm256_load_si256((const __m256i*)S.LO), CY = _mm256_load_si256((const __m256i*)S.CY);
    const __m256i m12 = _mm256_set1_epi32(FAST ? 0x1fff : 0xfff), one = _mm256_set1_epi32(1), lim = _mm256_set1_epi32(0x10000), b12 = _mm256_set1_epi32(FAST ? 0x2000 : 0x1000);
    const __m256i c13 = _mm256_set1_epi32(13), c16 = _mm256_set1_epi32(16), zero = _mm256_setzero_si256();
    const __m256i lane_id = _mm256_setr_epi32(0 << 17, 1 << 17, 2 << 17, 3 << 17, 4 << 17, 5 << 17, 6 << 17, 7 << 17);
    (void)c13; (void)c16;
    // index / shift tables: output register q holds field 2 q of the eight streams in its low half, field 2 q + 1 in its high half
    alignas(64) static const struct Tab { uint8_t idx[4][64]; uint32_t sh[4][16]; } T = [] {
        Tab t;
        static const int off13[8] = {0, 1, 3, 4, 6, 8, 9, 11}, shf13[8] = {0, 5, 2, 7, 4, 1, 6, 3};
        for (int q = 0; q < 4; ++q)
            for (int d = 0; d < 16; ++d) {
                const int l = d & 7, k = 2 * q + (d >> 3);
                for (int b = 0; b < 4; ++b) t.idx[q][4 * d + b] = (uint8_t)(64 * (l >> 2) + 16 * (l & 3) + (P13 ? off13[k] + b : 2 * k + (b & 1)));   // (16-bit entries: bytes 2, 3 of a lane are masked to zero)
                t.sh[q][d] = P13 ? (uint32_t)shf13[k] : 0u;
            }
        return t;
    }();
    const __m512i ix0 = _mm512_load_si512(T.idx[0]), ix1 = _mm512_load_si512(T.idx[1]), ix2 = _mm512_load_si512(T.idx[2]), ix3 = _mm512_load_si512(T.idx[3]);
    const __m512i sh0 = _mm512_load_si512(T.sh[0]), sh1 = _mm512_load_si512(T.sh[1]), sh2 = _mm512_load_si512(T.sh[2]), sh3 = _mm512_load_si512(T.sh[3]);
#define BSC_X8_CHUNK13(l) (P13 ? _mm_loadu_si128((const __m128i*)((const uint8_t*)ps[l] + (i >> 3) * 13)) : _mm_loadu_si128((const __m128i*)(ps[l] + i)))
    const __mmask64 low2 = 0x3333333333333333ull;
    for (; i < end; i += 8) {
        if (P13) BSC_X8_PREFETCH13(ps, i, pf); else BSC_X8_PREFETCH(ps, i, pf);
        __m512i z0 = _mm512_castsi128_si512(BSC_X8_CHUNK13(0)), z1 = _mm512_castsi128_si512(BSC_X8_CHUNK13(4));
        z0 = _mm512_inserti32x4(z0, BSC_X8_CHUNK13(1), 1); z1 = _mm512_inserti32x4(z1, BSC_X8_CHUNK13(5), 1);
        z0 = _mm512_inserti32x4(z0, BSC_X8_CHUNK13(2), 2); z1 = _mm512_inserti32x4(z1, BSC_X8_CHUNK13(6), 2);
        z0 = _mm512_inserti32x4(z0, BSC_X8_CHUNK13(3), 3); z1 = _mm512_inserti32x4(z1, BSC_X8_CHUNK13(7), 3);
        __m512i q0, q1, q2, q3;
        if (P13) {
            q0 = _mm512_srlv_epi32(_mm512_permutex2var_epi8(z0, ix0, z1), sh0); q1 = _mm512_srlv_epi32(_mm512_permutex2var_epi8(z0, ix1, z1), sh1);
            q2 = _mm512_srlv_epi32(_mm512_permutex2var_epi8(z0, ix2, z1), sh2); q3 = _mm512_srlv_epi32(_mm512_permutex2var_epi8(z0, ix3, z1), sh3);
        } else {
            q0 = _mm512_maskz_permutex2var_epi8(low2, z0, ix0, z1); q1 = _mm512_maskz_permutex2var_epi8(low2, z0, ix1, z1);
            q2 = _mm512_maskz_permutex2var_epi8(low2, z0, ix2, z1); q3 = _mm512_maskz_permutex2var_epi8(low2, z0, ix3, z1);
        }
        const __m256i w0 = _mm512_castsi512_si256(q0), w1 = _mm512_extracti64x4_epi64(q0, 1), w2 = _mm512_castsi512_si256(q1), w3 = _mm512_extracti64x4_epi64(q1, 1);
        const __m256i w4 = _mm512_castsi512_si256(q2), w5 = _mm512_extracti64x4_epi64(q2, 1), w6 = _mm512_castsi512_si256(q3), w7 = _mm512_extracti64x4_epi64(q3, 1);
        if (VSEL == 2) { BSC_X8_STEP512W(w0); BSC_X8_STEP512W(w1); BSC_X8_STEP512W(w2); BSC_X8_STEP512W(w3); BSC_X8_STEP512W(w4); BSC_X8_STEP512W(w5); BSC_X8_STEP512W(w6); BSC_X8_STEP512W(w7); }
        else           { BSC_X8_STEP512(w0); BSC_X8_STEP512(w1); BSC_X8_STEP512(w2); BSC_X8_STEP512(w3); BSC_X8_STEP512(w4); BSC_X8_STEP512(w5); BSC_X8_STEP512(w6); BSC_X8_STEP512(w7); }
    }
#undef BSC_X8_CHUNK13
    _mm256_store_si256((__m256i*)S.R, R); _mm256_store_si256((__m256i*)S.LO, LO); _mm256_store_si256((__m256i*)S.CY, CY);
    return logp;
}
#undef BSC_X8_STEP512
#undef BSC_X8_STEP512W
#endif

// entries (of 2 bytes) the eight-lane coder prefetches ahead in every stream; BSC_RC_PREFETCH overrides, 0 = off
static int g_x8_prefetch_override = -1;                              // tools/rc_host_bench.cpp (which includes this file) varies both inside one process
static int g_x8_vsel_override = -1;
static int x8_vector_select()                                        // BSC_RC_VSEL=0: the round-4 step (see x8_steps_avx512)
{
    static const int env = [] { const char* e = getenv("BSC_RC_VSEL"); return e ? atoi(e) : 2; }();
    return g_x8_vsel_override >= 0 ? g_x8_vsel_override : env;
}
static int x8_prefetch_entries()
{
    static const int env = [] { const char* e = getenv("BSC_RC_PREFETCH"); return e ? atoi(e) : 256; }();      // 512 bytes ahead: -2 % (104 -> 102 -> 96 ms per block with the new step)
    return g_x8_prefetch_override >= 0 ? g_x8_prefetch_override : env;
}
template <bool FAST, bool P13 = false>
static bool encode_pstream_x8(const PstreamJob* J, int* res)
{
#if defined(__AVX2__)
    RunView H;
    RangeEncoder rc[8];
    size_t common = ~(size_t)0;
    for (int l = 0; l < 8; ++l) {
        H.nsym = J[l].nsym; memcpy(H.first_seen, J[l].first_seen, (size_t)J[l].nsym);
        rc[l].init(J[l].out, J[l].out_size);
        rc[l].encode_word((uint32_t)J[l].in_size);
        if (FAST) (void)encode_alphabet(H, [&](unsigned b) { rc[l].template encode<1>(b, 1); });
        else      (void)encode_alphabet(H, [&](unsigned b) { rc[l].encode_half(b); });
        if (J[l].count < common) common = J[l].count;
    }
    X8State S;
    for (int l = 0; l < 8; ++l) { const RangeEncoder::Live L = rc[l].enter(); S.R[l] = L.range; S.LO[l] = (uint32_t)L.low; S.CY[l] = (uint32_t)(L.low >> 32); }
    static const bool use512 = [] {
        if (const char* e = getenv("BSC_RC_AVX512")) return atoi(e) != 0 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
    }();

    // The VBMI form of the packed stream's unpacking: on by default on AMD hosts that have it (EPYC 9575F: the eight-lane task 101.6 -> 96.0 ms
    // per block, against 94.3 on 16-bit entries); off by default elsewhere — on the Xeon of the build container the 512-bit permutes
    // among 256-bit steps cost 40 % (1.03 -> 1.46 ns per decision).  BSC_RC_VBMI=1 / 0 overrides.
    static const bool use_vbmi = [] {
        const bool have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi");
        if (const char* e = getenv("BSC_RC_VBMI")) return atoi(e) != 0 && have;
        return have && __builtin_cpu_is("amd");
    }();
    // (16-bit entries through the same permute: measured SLOWER than the 128-bit transpose on the EPYC 9575F — 96.7 against 95.0 ms per block —
    // so only on request: BSC_RC_VBMI16=1)
    static const bool use_vbmi16 = [] {
        const char* e = getenv("BSC_RC_VBMI16");
        return e && atoi(e) != 0 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi");
    }();
    const size_t pf = (size_t)x8_prefetch_entries();
    const int vsel = x8_vector_select();
    constexpr size_t CHUNK = 32768;                                  // steps between two replays of the log (1 MiB of records at most)
    static thread_local std::unique_ptr<uint32_t[]> log_mem;
    if (!log_mem) log_mem.reset(new uint32_t[CHUNK * 8 + 16]);
    uint32_t* const log0 = log_mem.get();

    const uint16_t* ps[8];
    for (int l = 0; l < 8; ++l) ps[l] = J[l].ps;
    size_t i = 0;
    while (i + 8 <= common) {
        size_t end = i + CHUNK; if (end > common) end = common;
        end = i + ((end - i) & ~(size_t)7);
        uint32_t* const logp = !use512 ? x8_steps_avx2<FAST, P13>(S, ps, i, end, log0, pf)
                             : (P13 ? use_vbmi : use_vbmi16) ? (vsel != 0 ? x8_steps_avx512_vbmi<FAST, 2, P13>(S, ps, i, end, log0, pf) : x8_steps_avx512_vbmi<FAST, 0, P13>(S, ps, i, end, log0, pf))
                             : vsel != 0 ? x8_steps_avx512<FAST, 2, P13>(S, ps, i, end, log0, pf) : x8_steps_avx512<FAST, 0, P13>(S, ps, i, end, log0, pf);
        i = end;
        for (const uint32_t* q = log0; q < logp; ++q) {              // replay: the only code that touches the outputs
            const uint32_t rec = *q;
            RangeEncoder& e = rc[rec >> 17];
            if (e.full()) return false;
            e.emit_unit(rec & 0xffffu, (rec >> 16) & 1u);
        }
    }
    for (int l = 0; l < 8; ++l) if (rc[l].full()) return false;
    // the rest of every stream on its own (they differ in length by a few per cent), with the run-start test of the scalar coder
    for (int l = 0; l < 8; ++l) {
        RangeEncoder::Live L{(uint64_t)S.LO[l] | ((uint64_t)S.CY[l] << 32), S.R[l]};
        bool failed = false;
        const uint16_t* q = ps[l];
        unsigned is_full = (unsigned)rc[l].full();
        if (P13) {
            for (size_t k = i; k < J[l].count; ++k) {
                if (__builtin_expect(is_full != 0u, 0)) { failed = true; break; }
                const unsigned x = p13_get((const uint8_t*)q, k);
                rc[l].template encode_live_f<12>(L, x >> 12, (int)(x & 0xfffu), is_full);
            }
        } else
        for (size_t k = i; k < J[l].count; ++k) {
            const unsigned x = q[k];
            if (FAST) {
                if (__builtin_expect(((x >> 14) & is_full) != 0u, 0)) { failed = true; break; }
                rc[l].encode_live_var(L, (x >> 13) & 1u, x & 0x1fffu, psf_prec(x), is_full);
            } else {
                if (__builtin_expect(((x >> 13) & is_full) != 0u, 0)) { failed = true; break; }
                rc[l].template encode_live_f<12>(L, (x >> 12) & 1u, (int)(x & 0xfffu), is_full);
            }
        }
        rc[l].leave(L);
        res[l] = failed ? NOT_COMPRESSIBLE : rc[l].finish();
    }
    return true;
#else
    (void)J; (void)res;
    return false;
#endif
}
// ------------------------------------------------------------------------------------------------
// Sixteen sub-blocks — TWO device-model blocks — in the 32-bit lanes of 512-bit registers (round 6).  A step of the eight-lane coder
// above costs about the same micro-ops whatever its width (that is why a four-lane version lost), and the EPYC hosts of the MI355X
// boxes execute 512-bit integer operations at the rate of 256-bit ones: the same step on sixteen lanes codes two blocks in the time
// of one, i.e. HALF the CPU time per block (the range coder is the whole of the host's work behind the device model: ~0.10 CPU-s per
// 64 MiB block, 75-105 cores for an 8-GPU node by DESIGN.md 7's budget).  Latency per task is unchanged (~100 ms: 23 M dependent steps), and a
// task needs two blocks at once: the coder pool pairs eight-lane blocks that are waiting together (block.cpp).
// The step is the mask-register form of x8_steps_avx512 (VSEL = 0) — the vector-mask trick of round 5's step has no 512-bit encoding.
// Log records carry a 4-bit lane number; replay, tails and the give-up rule are those of the eight-lane coder.
// ------------------------------------------------------------------------------------------------
#if defined(__AVX2__)
struct alignas(64) X16State { uint32_t R[16], LO[16], CY[16]; };

template <bool FAST>
__attribute__((target("avx512f,avx512vl,avx512bw")))
static uint32_t* x16_steps_avx512(X16State& S, const uint16_t* const* ps, size_t i, size_t end, uint32_t* logp, size_t pf)
{
    __m512i R = _mm512_load_si512((const void*)S.R), LO = _mm512_load_si512((const void*)S.LO), CY = _mm512_load_si512((const void*)S.CY);
    const __m512i m12 = _mm512_set1_epi32(FAST ? 0x1fff : 0xfff), one = _mm512_set1_epi32(1), lim = _mm512_set1_epi32(0x10000), b12 = _mm512_set1_epi32(FAST ? 0x2000 : 0x1000);
    const __m512i c13 = _mm512_set1_epi32(13), c16 = _mm512_set1_epi32(16);
    const __m512i lane_id = _mm512_setr_epi32(0 << 17, 1 << 17, 2 << 17, 3 << 17, 4 << 17, 5 << 17, 6 << 17, 7 << 17,
                                              8 << 17, 9 << 17, 10 << 17, 11 << 17, 12 << 17, 13 << 17, 14 << 17, 15 << 17);
#define BSC_X16_STEP(xv) do {                                                                                                      \
        const __m512i x = (xv);                                                                                                    \
        const __mmask16 need = _mm512_cmplt_epu32_mask(R, lim);                                        /* range < 2^16 */          \
        const __m512i rec = _mm512_ternarylogic_epi32(_mm512_srli_epi32(LO, 16), _mm512_slli_epi32(CY, 16), lane_id, 0xfe);         \
        _mm512_storeu_si512((void*)logp, _mm512_maskz_compress_epi32(need, rec));                                                  \
        logp += __builtin_popcount((unsigned)need);                                                                                \
        LO = _mm512_mask_slli_epi32(LO, need, LO, 16);                                                                             \
        CY = _mm512_maskz_mov_epi32((__mmask16)~need, CY);                                                                         \
        const __m512i p = _mm512_and_si512(x, m12);                                                                                \
        const __m512i sh = _mm512_sub_epi32(c13, _mm512_slli_epi32(_mm512_srli_epi32(x, 15), 1));                                  \
        const __m512i ra = FAST ? _mm512_mullo_epi32(_mm512_srlv_epi32(R, sh), p) : _mm512_mullo_epi32(_mm512_srli_epi32(R, 12), p); \
        const __m512i rb = FAST ? _mm512_mullo_epi32(_mm512_sllv_epi32(R, _mm512_sub_epi32(c16, sh)), p) : _mm512_mullo_epi32(_mm512_slli_epi32(R, 4), p); \
        R  = _mm512_mask_slli_epi32(R, need, R, 16);                                                                               \
        const __mmask16 kb = _mm512_test_epi32_mask(x, b12);                                           /* the coded bit */         \
        const __m512i r = _mm512_mask_mov_epi32(ra, need, rb);                                                                     \
        const __m512i lo2 = _mm512_mask_add_epi32(LO, kb, LO, r);                                                                  \
        CY = _mm512_mask_add_epi32(CY, _mm512_cmplt_epu32_mask(lo2, LO), CY, one);                     /* wrapped: carry out */    \
        LO = lo2;                                                                                                                  \
        R = _mm512_mask_sub_epi32(r, kb, R, r);                                                        /* bit ? range - r : r */   \
    } while (0)
    const uint16_t* const* pa = ps; const uint16_t* const* pb = ps + 8;
    for (; i < end; i += 8) {
        if (pf) {
            const unsigned l2 = ((unsigned)(i >> 3) & 3u) * 2u;
            _mm_prefetch((const char*)(pa[l2] + i + pf), _MM_HINT_T0); _mm_prefetch((const char*)(pa[l2 + 1] + i + pf), _MM_HINT_T0);
            _mm_prefetch((const char*)(pb[l2] + i + pf), _MM_HINT_T0); _mm_prefetch((const char*)(pb[l2 + 1] + i + pf), _MM_HINT_T0);
        }
        __m512i xs[8];
        {
            BSC_X8_TRANSPOSE(pa, i);
            xs[0] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t0)); xs[1] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t1));
            xs[2] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t2)); xs[3] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t3));
            xs[4] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t4)); xs[5] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t5));
            xs[6] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t6)); xs[7] = _mm512_castsi256_si512(_mm256_cvtepu16_epi32(t7));
        }
        {
            BSC_X8_TRANSPOSE(pb, i);
            xs[0] = _mm512_inserti64x4(xs[0], _mm256_cvtepu16_epi32(t0), 1); xs[1] = _mm512_inserti64x4(xs[1], _mm256_cvtepu16_epi32(t1), 1);
            xs[2] = _mm512_inserti64x4(xs[2], _mm256_cvtepu16_epi32(t2), 1); xs[3] = _mm512_inserti64x4(xs[3], _mm256_cvtepu16_epi32(t3), 1);
            xs[4] = _mm512_inserti64x4(xs[4], _mm256_cvtepu16_epi32(t4), 1); xs[5] = _mm512_inserti64x4(xs[5], _mm256_cvtepu16_epi32(t5), 1);
            xs[6] = _mm512_inserti64x4(xs[6], _mm256_cvtepu16_epi32(t6), 1); xs[7] = _mm512_inserti64x4(xs[7], _mm256_cvtepu16_epi32(t7), 1);
        }
        BSC_X16_STEP(xs[0]); BSC_X16_STEP(xs[1]); BSC_X16_STEP(xs[2]); BSC_X16_STEP(xs[3]);
        BSC_X16_STEP(xs[4]); BSC_X16_STEP(xs[5]); BSC_X16_STEP(xs[6]); BSC_X16_STEP(xs[7]);
    }
#undef BSC_X16_STEP
    _mm512_store_si512((void*)S.R, R); _mm512_store_si512((void*)S.LO, LO); _mm512_store_si512((void*)S.CY, CY);
    return logp;
}
#endif

bool qlfc_x16_available()
{
#if defined(__AVX2__)
    static const bool ok = [] {
        // Opt-in (BSC_RC_X16=1).  Measured on the pool's EPYC 9575F hosts (profiles/r06/sixteen_lane_coder.txt): 0.114 -> 0.093 CPU-s per
        // block at the same throughput over 160 blocks (the step costs ~1.5 x the eight-lane step for twice the lanes there, not 1 x), but a
        // 20-block job is ~4 % slower: a block waits up to 15 ms for its partner, and that work then sits in the job's tail.
        const char* e = getenv("BSC_RC_X16");
        if (!e || atoi(e) == 0) return false;
        return (bool)(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw"));
    }();
    return ok;
#else
    return false;
#endif
}

template <bool FAST>
static bool encode_pstream_x16(const PstreamJob* J, int* res)
{
#if defined(__AVX2__)
    if (!qlfc_x16_available()) return false;
    RunView H;
    RangeEncoder rc[16];
    size_t common = ~(size_t)0;
    for (int l = 0; l < 16; ++l) {
        H.nsym = J[l].nsym; memcpy(H.first_seen, J[l].first_seen, (size_t)J[l].nsym);
        rc[l].init(J[l].out, J[l].out_size);
        rc[l].encode_word((uint32_t)J[l].in_size);
        if (FAST) (void)encode_alphabet(H, [&](unsigned b) { rc[l].template encode<1>(b, 1); });
        else      (void)encode_alphabet(H, [&](unsigned b) { rc[l].encode_half(b); });
        if (J[l].count < common) common = J[l].count;
    }
    X16State S;
    for (int l = 0; l < 16; ++l) { const RangeEncoder::Live L = rc[l].enter(); S.R[l] = L.range; S.LO[l] = (uint32_t)L.low; S.CY[l] = (uint32_t)(L.low >> 32); }
    const size_t pf = (size_t)x8_prefetch_entries();
    constexpr size_t CHUNK = 32768;                                  // steps between two replays of the log (2 MiB of records at most)
    static thread_local std::unique_ptr<uint32_t[]> log_mem;
    if (!log_mem) log_mem.reset(new uint32_t[CHUNK * 16 + 32]);
    uint32_t* const log0 = log_mem.get();
    const uint16_t* ps[16];
    for (int l = 0; l < 16; ++l) ps[l] = J[l].ps;
    size_t i = 0;
    while (i + 8 <= common) {
        size_t end = i + CHUNK; if (end > common) end = common;
        end = i + ((end - i) & ~(size_t)7);
        uint32_t* const logp = x16_steps_avx512<FAST>(S, ps, i, end, log0, pf);
        i = end;
        for (const uint32_t* q = log0; q < logp; ++q) {              // replay: the only code that touches the outputs
            const uint32_t rec = *q;
            RangeEncoder& e = rc[rec >> 17];
            if (e.full()) return false;
            e.emit_unit(rec & 0xffffu, (rec >> 16) & 1u);
        }
    }
    for (int l = 0; l < 16; ++l) if (rc[l].full()) return false;
    // the rest of every stream on its own, with the run-start test of the scalar coder (as in the eight-lane coder; the two blocks'
    // sub-blocks differ in length by a few per cent)
    for (int l = 0; l < 16; ++l) {
        RangeEncoder::Live L{(uint64_t)S.LO[l] | ((uint64_t)S.CY[l] << 32), S.R[l]};
        bool failed = false;
        const uint16_t* q = ps[l];
        unsigned is_full = (unsigned)rc[l].full();
        for (size_t k = i; k < J[l].count; ++k) {
            const unsigned x = q[k];
            if (FAST) {
                if (__builtin_expect(((x >> 14) & is_full) != 0u, 0)) { failed = true; break; }
                rc[l].encode_live_var(L, (x >> 13) & 1u, x & 0x1fffu, psf_prec(x), is_full);
            } else {
                if (__builtin_expect(((x >> 13) & is_full) != 0u, 0)) { failed = true; break; }
                rc[l].template encode_live_f<12>(L, (x >> 12) & 1u, (int)(x & 0xfffu), is_full);
            }
        }
        rc[l].leave(L);
        res[l] = failed ? NOT_COMPRESSIBLE : rc[l].finish();
    }
    return true;
#else
    (void)J; (void)res;
    return false;
#endif
}
bool qlfc_encode_static_pstream_x16(const PstreamJob* J, int* res) { return encode_pstream_x16<false>(J, res); }
bool qlfc_encode_fast_pstream_x16(const PstreamJob* J, int* res) { return encode_pstream_x16<true>(J, res); }

bool qlfc_encode_static_pstream_x8(const PstreamJob* J, int* res) { return encode_pstream_x8<false>(J, res); }
bool qlfc_encode_static_p13_x8(const PstreamJob* J, int* res) { return encode_pstream_x8<false, true>(J, res); }
bool qlfc_encode_fast_pstream_x8(const PstreamJob* J, int* res) { return encode_pstream_x8<true>(J, res); }

int qlfc_encode_runs(const RunView& R, int in_size, uint8_t* out, int out_size, int coder)
{
    if (in_size <= 0 || R.count == 0) return BAD_PARAMETER;
    switch (coder) {
        case CODER_STATIC:   return encode_model1<false>(R, out, in_size, out_size);
        case CODER_ADAPTIVE: return encode_model1<true>(R, out, in_size, out_size);
        case CODER_FAST:     return encode_model2(R, out, in_size, out_size);
    }
    return BAD_PARAMETER;
}

int qlfc_encode_block(const uint8_t* in, uint8_t* out, int in_size, int out_size, int coder)
{
    if (in_size <= 0) return BAD_PARAMETER;
    QlfcRuns R;
    qlfc_runs(in, in_size, R);
    return qlfc_encode_runs(R.view, in_size, out, out_size, coder);
}

}  // namespace bschost

