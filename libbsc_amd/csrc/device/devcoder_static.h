// devcoder_static.h — the static coder's CONTEXT-FREE counter family evaluated in stream order (round 6), shared by the HIP
// kernels (devcoder.hip) and by the CPU check (tools/devcoder_static_sim.cpp).  Plain integer functions, no HIP dependency.
//
// The context-free ("static") family of qlfc.cpp:829-1129 indexes its counters by nothing but the decision's place in the code
// tree: chain = (sub-block, decision type).  Its decisions therefore ARE the runs in stream order, filtered by a predicate on
// the run's rank, and nothing has to be brought into chain-major order to walk them:
//
//   * a TILE is 64 consecutive runs; the ranks of a tile are kept as bit planes (one 64-bit word per rank bit).  With at most 32
//     symbols per sub-block (max_rank <= 4: lower-case text, DNA, digits...) a rank has 5 bits and there are at most 31 rank-side
//     types (RF, RE 0..3, RM nodes of B <= 4): the SLOTS.  Which lanes of a tile hold a decision of a slot, and the coded bits,
//     are two 64-bit masks obtained from the planes by a prefix match (every type's rank set is a sub-cube of the rank bits or
//     the complement of one: SpDesc, built from the model's own case analysis and verified rank by rank).
//   * phase A: lane = (slot, chunk of SP_CT tiles) walks its events from both ends of the attainable range (the update maps are
//     monotone: devcoder.hip); the family's rates are fast (17-33 % per step), so a chunk nearly always closes the bracket.
//   * resolve: exact value at the start of every chunk (closed predecessor: its end value; open ones: replay of the <= 64 events
//     it recorded, or a walk of the chunk).
//   * phase C: exact walk, leaving the counter value at the start of every SUB-TILE (8 / 16 / 32 / 64 runs, by the slot's density).
//   * values: one wavefront per tile, lane = (slot, sub-tile): at most ~8 events each from the recorded value; the value every
//     decision sees goes to an 8 x u16 record per run (decision k of the rank side -> entry k), which the p-stream kernel reads
//     with ONE 16-byte load instead of two position loads and six 2-byte gathers.
//
// No partition, no positions, no chain-major copy of this family.  The run side's NE / NM decisions of the family (5 % of the
// decisions of text; NF has weight 0 in the blend and a rate of 0: qlfc_data.inc RUN_FIRST) stay on the general path.
#pragma once
#include "devcoder_model.h"

namespace dcs {
using namespace dcm;
typedef unsigned long long sp_u64;         // (= the device layer's u64)

constexpr int SP_TILE = 64;                 // runs per tile
#ifndef SP_CT_N
#define SP_CT_N 32
#endif
constexpr int SP_CT = SP_CT_N;              // tiles per chunk
constexpr int SP_PLANES = 5;                // rank bits (max_rank <= 4)
constexpr int SP_MAXR = 4;
constexpr int SP_SLOTS = 32;                // 0: RF | 1..4: RE 0..3 | 5..30: RM (B <= 4; 5 + tau - TAU_RM) | 31: unused
constexpr int SP_RG = 64;                   // chunks per resolve lane
constexpr int SP_HIST = 64;                 // events a chunk remembers

DC_HD int sp_slot_of_tau(int tau)
{
    if (tau == TAU_RF) return 0;
    if (tau < TAU_RM) return (tau - TAU_RE) < 4 ? 1 + (tau - TAU_RE) : -1;
    return (tau - TAU_RM) < 26 ? 5 + (tau - TAU_RM) : -1;
}
DC_HD int sp_slot_class(int slot) { return slot == 0 ? CLS_RF : slot < 5 ? CLS_RE : CLS_RM; }
// sub-tiles per tile of a slot (fixed by the slot's kind: RF / RE0 are nearly every run, the root nodes of the four mantissa trees a
// fifth of the runs each, the rest a few per tile) and the lane map of the values kernel: 55 (slot, sub-tile) pairs
DC_HD int sp_slot_sub(int slot)
{
    if (slot <= 1) return 8;                // RF, RE0: sub-tiles of 8 runs
    if (slot <= 3) return 4;                // RE1, RE2: 16 runs
    if (slot == 4) return 1;                // RE3 (cannot occur with max_rank <= 4; kept for the table's shape)
    const int t = slot - 5;                 // RM: roots of the B-trees are t = rm_off(B) = 0, 1, 4, 11
    return (t == 0 || t == 1 || t == 4 || t == 11) ? 2 : 1;
}
constexpr int SP_LANES = 8 + 8 + 4 + 4 + 1 + 4 * 2 + 22;      // = 55
// lane -> (slot, sub-tile): slots in order, sub-tiles in order.  The _ref forms say what the map is; the closed forms are what the
// kernels run (tools/devcoder_static_sim.cpp checks them against each other).
DC_HD void sp_lane_map_ref(int lane, int* slot, int* q)
{
    int l = lane;
    for (int s = 0; s < SP_SLOTS - 1; ++s) {
        const int sub = sp_slot_sub(s);
        if (l < sub) { *slot = s; *q = l; return; }
        l -= sub;
    }
    *slot = -1; *q = 0;
}
DC_HD int sp_slot_first_lane_ref(int slot)
{
    int l = 0;
    for (int s = 0; s < slot; ++s) l += sp_slot_sub(s);
    return l;
}
DC_HD void sp_lane_map(int lane, int* slot, int* q)
{
    const int l = lane;
    if (l < 16) { *slot = l >> 3; *q = l & 7; return; }
    if (l < 24) { *slot = 2 + ((l - 16) >> 2); *q = (l - 16) & 3; return; }
    if (l == 24) { *slot = 4; *q = 0; return; }
    const int x = l - 25;                                     // the tree nodes t = 0..25 (slot 5 + t); the roots t = 0, 1, 4, 11 have two lanes
    int t, qq = 0;
    if (x < 4) { t = x >> 1; qq = x & 1; }
    else if (x < 6) t = x - 2;
    else if (x < 8) { t = 4; qq = x - 6; }
    else if (x < 14) t = x - 3;
    else if (x < 16) { t = 11; qq = x - 14; }
    else if (x < 30) t = x - 4;
    else { *slot = -1; *q = 0; return; }
    *slot = 5 + t; *q = qq;
}
// first entry of a slot in the per-tile state record (entries of all slots of one tile are NOT interleaved in memory: see sp_state_base)
DC_HD int sp_slot_first_lane(int slot)
{
    if (slot < 2) return 8 * slot;
    if (slot < 4) return 16 + 4 * (slot - 2);
    if (slot == 4) return 24;
    const int t = slot - 5;
    return 25 + t + (t > 0 ? 1 : 0) + (t > 1 ? 1 : 0) + (t > 4 ? 1 : 0) + (t > 11 ? 1 : 0);
}

// How a slot's decisions are read off the rank planes: lanes whose rank matches `pat` on the `care` bits (all five planes: a
// sub-cube of the rank space), optionally complemented; the coded bit likewise (it only has to be right on lanes that match).
struct SpDesc { uint8_t on_care, on_pat, on_inv, bit_care, bit_pat, bit_inv, k, present; };

DC_HD sp_u64 sp_match(const sp_u64* p, uint32_t care, uint32_t pat, uint32_t inv)
{
    sp_u64 acc = ~0ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; b < SP_PLANES; ++b) {
        const sp_u64 x = ((pat >> b) & 1u) ? p[b] : ~p[b];
        acc &= ((care >> b) & 1u) ? x : ~0ull;
    }
    return inv ? ~acc : acc;
}

// The rank side of one run as the partition kernels see it (devcoder.hip dc_item_rounds, no escape coding): f(tau, bit, k)
template <class F>
DC_HD void sp_rank_side(uint32_t rank, int maxr, F&& f)
{
    int k = 0;
    f(TAU_RF, rank != 1u ? 1u : 0u, k++);
    if (rank == 1u) return;
    const int B = bsr(rank);
    const int e = B ? (B - 1) + (B < maxr ? 1 : 0) : 0;
    for (int sx = 0; sx < e && sx < 7; ++sx) f(TAU_RE + sx, sx + 1 < B ? 1u : 0u, k++);
    for (int d = 0; d < B; ++d) f(TAU_RM + rm_off(B) + (int)(rank >> (B - d)) - 1, (rank >> (B - 1 - d)) & 1u, k++);
}

// Descriptors of all slots for one max_rank, derived from sp_rank_side itself and checked on every rank value that can occur
// (rank < 2^(max_rank + 1)).  Returns false if some type's rank set is not a (complemented) sub-cube — the path is then not offered.
inline bool sp_build_descs(int maxr, SpDesc* D /*[SP_SLOTS]*/)
{
    if (maxr < 0 || maxr > SP_MAXR) return false;
    const uint32_t nr = 2u << maxr;                            // ranks 0 .. nr - 1
    uint32_t on[SP_SLOTS] = {0}, bit[SP_SLOTS] = {0};
    int kk[SP_SLOTS];
    for (int s = 0; s < SP_SLOTS; ++s) kk[s] = -1;
    bool ok = true;
    for (uint32_t r = 0; r < nr; ++r) {
        int nd = 0;
        sp_rank_side(r, maxr, [&](int tau, uint32_t b, int k) {
            const int s = sp_slot_of_tau(tau);
            if (s < 0 || k > 7) { ok = false; return; }
            on[s] |= 1u << r; if (b) bit[s] |= 1u << r;
            if (kk[s] >= 0 && kk[s] != k) ok = false;
            kk[s] = k; ++nd;
        });
        if (nd > 8) ok = false;
    }
    if (!ok) return false;
    const uint32_t dom = nr >= 32 ? 0xffffffffu : ((1u << nr) - 1u);
    // smallest sub-cube holding `set`; exact iff it holds nothing else of `within`
    auto cube = [&](uint32_t set, uint32_t within, uint8_t* care, uint8_t* pat) -> bool {
        uint32_t all1 = 31u, all0 = 31u;
        for (uint32_t r = 0; r < 32; ++r) if ((set >> r) & 1u) { all1 &= r; all0 &= ~r; }
        *care = (uint8_t)((all1 | all0) & 31u); *pat = (uint8_t)(all1 & 31u);
        for (uint32_t r = 0; r < 32; ++r) {
            if (!((within >> r) & 1u)) continue;
            const bool in = ((r ^ *pat) & *care) == 0;
            if (in != (((set >> r) & 1u) != 0)) return false;
        }
        return true;
    };
    auto fit = [&](uint32_t set, uint32_t within, uint8_t* care, uint8_t* pat, uint8_t* inv) -> bool {
        set &= within;
        if (set == 0) { *care = 0; *pat = 0; *inv = 1; return true; }               // nothing: the complement of everything
        if (set == within) { *care = 0; *pat = 0; *inv = 0; return true; }
        if (cube(set, within, care, pat)) { *inv = 0; return true; }
        if (cube(within & ~set, within, care, pat)) { *inv = 1; return true; }
        return false;
    };
    for (int s = 0; s < SP_SLOTS; ++s) {
        SpDesc d = {0, 0, 1, 0, 0, 1, 0, 0};
        if (on[s]) {
            d.present = 1; d.k = (uint8_t)kk[s];
            if (!fit(on[s], dom, &d.on_care, &d.on_pat, &d.on_inv)) return false;
            if (!fit(bit[s], on[s], &d.bit_care, &d.bit_pat, &d.bit_inv)) return false;
        }
        D[s] = d;
    }
    return true;
}

// ---- one tile ---------------------------------------------------------------------------------------------------------
struct SpSub { uint32_t nb; uint32_t first[9]; uint32_t maxr[8]; };        // = devcoder.hip DcSub (first[nb] = m)
DC_HD uint32_t sp_sb_of(uint32_t j, const SpSub& S)
{
    uint32_t sb = 0;
    for (uint32_t b = 1; b < 8; ++b) if (b < S.nb && j >= S.first[b]) sb = b;
    return sb;
}
DC_HD sp_u64 sp_low(uint32_t n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }        // lanes [0, n)
DC_HD int sp_ctz64(sp_u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
// does a sub-block start inside tiles [t0, t1)?  (its first run is where the family's chains restart)
DC_HD bool sp_has_boundary(uint32_t t0, uint32_t t1, const SpSub& S)
{
    bool h = false;
    for (uint32_t b = 0; b < 8; ++b) if (b < S.nb) { const uint32_t t = S.first[b] / SP_TILE; h = h || (t >= t0 && t < t1); }
    return h;
}

// The masks of one slot over one tile: lanes that hold a decision of the slot (`on`), their coded bits (`bm`), and the lanes at
// which the chains restart (`rst`: first run of a sub-block).  sb = sub-block of the tile's first run; desc = [SP_MAXR + 1][SP_SLOTS],
// indexed by the sub-block's max_rank.
struct SpTileMasks { sp_u64 on, bm, rst; };
// du: the slot's descriptor when every sub-block has the same max_rank (then it is one value for the whole launch — scalar registers
// on the GPU — instead of a load per tile), else null.
DC_HD SpTileMasks sp_tile_masks(const sp_u64* planes, uint32_t tile, uint32_t m, const SpSub& S, uint32_t sb, const SpDesc* desc, int slot, bool has_bnd,
                                const SpDesc* du = nullptr)
{
    SpTileMasks M;
    const uint32_t j0 = tile * SP_TILE;
    const sp_u64 valid = (m - j0 >= (uint32_t)SP_TILE) ? ~0ull : sp_low(m - j0);
    M.rst = 0;
    if (!has_bnd) {
        const SpDesc d = du ? *du : desc[S.maxr[sb] * SP_SLOTS + slot];
        M.on = sp_match(planes, d.on_care, d.on_pat, d.on_inv) & valid;
        M.bm = sp_match(planes, d.bit_care, d.bit_pat, d.bit_inv);
        return M;
    }
    // sub-block starts inside this tile: every piece with its own sub-block's descriptors
    M.on = 0; M.bm = 0;
    for (uint32_t b = 0; b < 8; ++b) {
        if (b >= S.nb) break;
        const uint32_t f = S.first[b], e = S.first[b + 1];
        const uint32_t lo = f > j0 ? f : j0, hi = e < j0 + SP_TILE ? e : j0 + SP_TILE;
        if (lo >= hi) continue;
        const sp_u64 piece = sp_low(hi - j0) & ~sp_low(lo - j0);
        const SpDesc d = desc[S.maxr[b] * SP_SLOTS + slot];
        M.on |= sp_match(planes, d.on_care, d.on_pat, d.on_inv) & valid & piece;
        M.bm |= sp_match(planes, d.bit_care, d.bit_pat, d.bit_inv) & piece;
        if (f >= j0) M.rst |= 1ull << (f - j0);
    }
    return M;
}

// Events of `on` in lane order from value v; a chain restarts (v = init) when a lane of `rst` is reached or passed.
// f(lane, value before the decision, bit).  Returns the value after the last event; `rst` keeps the restarts not yet applied.
// (32-bit halves: a 64-bit find-first / clear-lowest costs three times the 32-bit ones on the GPU)
DC_HD int sp_ctz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}
template <class F>
DC_HD int sp_walk(int v, sp_u64 on, sp_u64 bm, sp_u64& rst, int init, const Rates& R, F&& f)
{
    if (rst == 0) {
        for (int h = 0; h < 2; ++h) {
            uint32_t o = (uint32_t)(on >> (32 * h));
            const uint32_t b = (uint32_t)(bm >> (32 * h));
            while (o) {
                const int i = sp_ctz32(o);
                const uint32_t bit = (b >> i) & 1u;
                f(i + 32 * h, v, bit);
                v = step(v, bit, R);
                o &= o - 1u;
            }
        }
        return v;
    }
    while (on) {
        const int i = sp_ctz64(on);
        if (rst & sp_low((uint32_t)i + 1u)) { v = init; rst &= ~sp_low((uint32_t)i + 1u); }
        const uint32_t bit = (uint32_t)(bm >> i) & 1u;
        f(i, v, bit);
        v = step(v, bit, R);
        on &= on - 1;
    }
    return v;
}

// ---- phases -------------------------------------------------------------------------------------------------------------
// Tiles per chunk of a slot: a lane is a serial chain, so the slots that have an event on (nearly) every run get short chunks —
// the kernel's time is its longest lane — and the family's fast rates still close a bracket within one (RF / RE0: 8 tiles = 512
// runs; RE1 / RE2: 16; the tree nodes, a fifth of the runs or less each: 32).
DC_HD uint32_t sp_slot_ct(int slot) { return slot <= 1 ? 8u : slot <= 3 ? 16u : 32u; }
constexpr uint32_t SP_CT_MIN = 8;

// what phase A leaves per (slot, chunk): the two ends of the bracket after the chunk, the number of events since the chunk's start
// or its last restart (saturating), the last SP_HIST bits (newest = bit 0), flags bit 0 = a restart happened inside (the end is
// then exact whatever the start was)
struct SpSum { uint16_t lo, hi, cnt, flags; sp_u64 hist; };              // 16 bytes

struct SpGeom {
    uint32_t m, ntiles, ntp /* tiles, padded to a multiple of 8 */, cstride /* chunk slots per slot = chunks at SP_CT_MIN */, gstride /* group slots per slot */;
};
DC_HD SpGeom sp_geom(uint32_t m)
{
    SpGeom g; g.m = m;
    g.ntiles = (m + SP_TILE - 1) / SP_TILE; if (g.ntiles == 0) g.ntiles = 1;
    g.ntp = (g.ntiles + 7u) & ~7u;
    g.cstride = (g.ntiles + SP_CT_MIN - 1) / SP_CT_MIN;
    g.gstride = (g.cstride + SP_RG - 1) / SP_RG;
    return g;
}
DC_HD uint32_t sp_nchunks(const SpGeom& g, int slot) { const uint32_t ct = sp_slot_ct(slot); return (g.ntiles + ct - 1) / ct; }
DC_HD uint32_t sp_ngroups(const SpGeom& g, int slot) { return (sp_nchunks(g, slot) + SP_RG - 1) / SP_RG; }
// the family's constants of one slot, fetched ONCE per lane / wavefront (read through the ModelParams reference inside the loops they were
// a memory round trip per chunk: the resolve kernels spent their time there)
struct SpParams { Rates R; int init, vmin, vmax; };
DC_HD SpParams sp_params(const ModelParams& mp, int slot)
{
    const int cls = sp_slot_class(slot);
    SpParams P; P.R = mp.rates[cls][FAM_STATIC]; P.init = mp.init[cls]; P.vmin = mp.vmin[cls][FAM_STATIC]; P.vmax = mp.vmax[cls][FAM_STATIC];
    return P;
}
DC_HD size_t sp_state_base(const SpGeom& g, int slot) { return (size_t)g.ntp * (size_t)sp_slot_first_lane(slot); }   // u16 entries; (slot, tile, q) at base + tile * sub + q

// Phase A, one lane: bracket walk of (slot, chunk).
DC_HD SpSum sp_phase_a(int slot, uint32_t chunk, const SpGeom& g, const SpSub& S, const sp_u64* planes /*[ntiles][SP_PLANES]*/,
                       const SpDesc* desc, const SpParams& P, const SpDesc* du = nullptr)
{
    const Rates R = P.R;
    const int init = P.init;
    int lo = P.vmin, hi = P.vmax;
    uint32_t cnt = 0, flags = 0;
    sp_u64 hist = 0;
    const uint32_t ct = sp_slot_ct(slot);
    const uint32_t t0 = chunk * ct, t1 = (t0 + ct < g.ntiles) ? t0 + ct : g.ntiles;
    const bool cb = sp_has_boundary(t0, t1, S);
    uint32_t sb = sp_sb_of(t0 * SP_TILE, S);
    for (uint32_t t = t0; t < t1; ++t) {
        const bool tb = cb && sp_has_boundary(t, t + 1, S);
        if (cb) sb = sp_sb_of(t * SP_TILE, S);
        SpTileMasks M = sp_tile_masks(planes + (size_t)t * SP_PLANES, t, g.m, S, sb, desc, slot, tb, du);
        if (!tb) {
            for (int h = 0; h < 2; ++h) {
                uint32_t o = (uint32_t)(M.on >> (32 * h));
                const uint32_t b = (uint32_t)(M.bm >> (32 * h));
                while (o) {
                    const uint32_t bit = (b >> sp_ctz32(o)) & 1u;
                    lo = step(lo, bit, R); hi = step(hi, bit, R);
                    hist = (hist << 1) | bit; ++cnt;
                    o &= o - 1u;
                }
            }
            continue;
        }
        sp_u64 on = M.on;
        while (on) {
            const int i = sp_ctz64(on);
            if (M.rst & sp_low((uint32_t)i + 1u)) { lo = hi = init; cnt = 0; hist = 0; flags |= 1u; M.rst &= ~sp_low((uint32_t)i + 1u); }
            const uint32_t bit = (uint32_t)(M.bm >> i) & 1u;
            lo = step(lo, bit, R); hi = step(hi, bit, R);
            hist = (hist << 1) | bit; ++cnt;
            on &= on - 1;
        }
        if (M.rst) { lo = hi = init; cnt = 0; hist = 0; flags |= 1u; }          // restarts behind the tile's last event
    }
    SpSum s; s.lo = (uint16_t)lo; s.hi = (uint16_t)hi; s.cnt = (uint16_t)(cnt < 0xffffu ? cnt : 0xffffu); s.flags = (uint16_t)flags; s.hist = hist;
    return s;
}

// exact single-ended walk of a whole chunk (resolve: an open chunk that saw more events than it remembers)
DC_HD int sp_walk_chunk(int v, int slot, uint32_t chunk, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams& P)
{
    const Rates R = P.R;
    const int init = P.init;
    const uint32_t ct = sp_slot_ct(slot);
    const uint32_t t0 = chunk * ct, t1 = (t0 + ct < g.ntiles) ? t0 + ct : g.ntiles;
    const bool cb = sp_has_boundary(t0, t1, S);
    uint32_t sb = sp_sb_of(t0 * SP_TILE, S);
    for (uint32_t t = t0; t < t1; ++t) {
        const bool tb = cb && sp_has_boundary(t, t + 1, S);
        if (cb) sb = sp_sb_of(t * SP_TILE, S);
        SpTileMasks M = sp_tile_masks(planes + (size_t)t * SP_PLANES, t, g.m, S, sb, desc, slot, tb);
        v = sp_walk(v, M.on, M.bm, M.rst, init, R, [](int, int, uint32_t) {});
        if (M.rst) v = init;
    }
    return v;
}

DC_HD int sp_replay(int v, sp_u64 hist, uint32_t cnt, const Rates& R)          // cnt <= SP_HIST events, oldest first
{
    for (int i = (int)cnt - 1; i >= 0; --i) v = step(v, (uint32_t)(hist >> i) & 1u, R);
    return v;
}
// value after chunk c given the exact value v at its start
DC_HD int sp_after_chunk(int v, const SpSum& s, int slot, uint32_t c, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc,
                         const SpParams& P, uint32_t* rewalks)
{
    if (s.lo == s.hi) return s.lo;                                       // closed (or restarted inside): exact whatever v was
    if (s.cnt <= SP_HIST) return sp_replay(v, s.hist, s.cnt, P.R);
    ++*rewalks;
    return sp_walk_chunk(v, slot, c, g, S, planes, desc, P);
}

// ---- resolve: the exact value at the start of every chunk, in three small steps ---------------------------------------------
//   1. per (slot, group of SP_RG chunks): what the group does to a value — "ends at E whatever came in" when one of its chunks is
//      closed (nearly always) or the bracket closes over the events of several open ones, else the <= SP_HIST events it saw, else
//      (`big`) the chunks have to be gone through one by one;
//   2. per slot, serially over the groups (a few hundred): the exact value at every group's start;
//   3. per (slot, group): the exact value at every chunk's start.
// A step that would need more than SP_MAX_REWALKS chunk walks gives up (the block then goes through the host model: nothing
// approximate is ever produced).
constexpr uint32_t SP_MAX_REWALKS = 24;
struct SpGroupSum { uint16_t closed, value, cnt, big; sp_u64 hist; };          // 16 bytes

// gs = the sums of the group's chunks (gs[0] = chunk grp * SP_RG): global memory, or a copy the wavefront made (every lane may run these
// functions on the same data: the control flow is then uniform)
// (the _g forms take the summaries through get(i) = summary of the group's i-th chunk: on the GPU a lane holds one and the others read it
// with v_readlane — no memory at all in the serial part)
template <class Get>
DC_HD bool sp_resolve_group_g(int slot, uint32_t grp, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams& P,
                              Get&& get, SpGroupSum* out)
{
    const uint32_t nch = sp_nchunks(g, slot);
    const uint32_t c0 = grp * SP_RG, c1 = (c0 + SP_RG < nch) ? c0 + SP_RG : nch;
    const Rates R = P.R;
    bool exact = false, big = false;
    int v = 0, lo = P.vmin, hi = P.vmax;     // while nothing is exact yet: the bracket through the open chunks
    uint32_t cnt = 0, rewalks = 0;
    sp_u64 hist = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const SpSum s = get((int)(c - c0));
        if (s.lo == s.hi) { exact = true; v = s.lo; continue; }
        if (exact) { v = sp_after_chunk(v, s, slot, c, g, S, planes, desc, P, &rewalks); if (rewalks > SP_MAX_REWALKS) return false; continue; }
        // a few events per chunk close no chunk's bracket, but a group's worth of them usually closes the group's
        if (s.cnt <= SP_HIST) { lo = sp_replay(lo, s.hist, s.cnt, R); hi = sp_replay(hi, s.hist, s.cnt, R); }
        else {
            lo = sp_walk_chunk(lo, slot, c, g, S, planes, desc, P); hi = sp_walk_chunk(hi, slot, c, g, S, planes, desc, P);
            rewalks += 2; if (rewalks > SP_MAX_REWALKS) return false;
        }
        if (lo == hi) { exact = true; v = lo; continue; }
        if (big || s.cnt > SP_HIST || cnt + s.cnt > SP_HIST) big = true;
        else { hist = s.cnt >= 64 ? s.hist : ((hist << s.cnt) | s.hist); cnt += s.cnt; }        // (cnt + s.cnt <= 64: the shift is below 64 unless hist is empty)
    }
    SpGroupSum o; o.closed = exact ? 1 : 0; o.value = (uint16_t)v; o.cnt = (uint16_t)cnt; o.big = big ? 1 : 0; o.hist = hist;
    *out = o;
    return true;
}
DC_HD bool sp_resolve_group(int slot, uint32_t grp, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams& P,
                            const SpSum* gs, SpGroupSum* out)
{
    return sp_resolve_group_g(slot, grp, g, S, planes, desc, P, [&](int i) { return gs[i]; }, out);
}
// value after group `grp` given the exact value at its start (step 2; also what step 3 does chunk by chunk)
DC_HD bool sp_after_group(int* v, const SpGroupSum& o, int slot, uint32_t grp, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc,
                          const SpParams& P, const SpSum* gs)
{
    if (o.closed) { *v = o.value; return true; }
    if (!o.big) { *v = sp_replay(*v, o.hist, o.cnt, P.R); return true; }
    const uint32_t nch = sp_nchunks(g, slot);
    const uint32_t c0 = grp * SP_RG, c1 = (c0 + SP_RG < nch) ? c0 + SP_RG : nch;
    const SpSum* sm = gs - c0;
    uint32_t rewalks = 0;
    for (uint32_t c = c0; c < c1; ++c) { *v = sp_after_chunk(*v, sm[c], slot, c, g, S, planes, desc, P, &rewalks); if (rewalks > SP_MAX_REWALKS) return false; }
    return true;
}
template <class Get, class Put>
DC_HD bool sp_resolve_chunks_g(int slot, uint32_t grp, int v, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams& P,
                               Get&& get, Put&& put /* (chunk index inside the group, its exact start value) */)
{
    const uint32_t nch = sp_nchunks(g, slot);
    const uint32_t c0 = grp * SP_RG, c1 = (c0 + SP_RG < nch) ? c0 + SP_RG : nch;
    uint32_t rewalks = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        put((int)(c - c0), v);
        v = sp_after_chunk(v, get((int)(c - c0)), slot, c, g, S, planes, desc, P, &rewalks);
        if (rewalks > SP_MAX_REWALKS) return false;
    }
    return true;
}
template <class Put>
DC_HD bool sp_resolve_chunks(int slot, uint32_t grp, int v, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams& P,
                             const SpSum* gs, Put&& put)
{
    return sp_resolve_chunks_g(slot, grp, v, g, S, planes, desc, P, [&](int i) { return gs[i]; }, put);
}

// Phase C, one lane: exact walk of (slot, chunk) from Sv, leaving the value at the start of every sub-tile.
DC_HD void sp_phase_c(int slot, uint32_t chunk, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams& P,
                      const uint16_t* Sv, uint16_t* state, const SpDesc* du = nullptr)
{
    const Rates R = P.R;
    const int init = P.init;
    const int sub = sp_slot_sub(slot), glen = SP_TILE / sub;
    uint16_t* st = state + sp_state_base(g, slot);
    int v = Sv[(size_t)slot * g.cstride + chunk];
    const uint32_t ct = sp_slot_ct(slot);
    const uint32_t t0 = chunk * ct, t1 = (t0 + ct < g.ntiles) ? t0 + ct : g.ntiles;
    const bool cb = sp_has_boundary(t0, t1, S);
    uint32_t sb = sp_sb_of(t0 * SP_TILE, S);
    for (uint32_t t = t0; t < t1; ++t) {
        const bool tb = cb && sp_has_boundary(t, t + 1, S);
        if (cb) sb = sp_sb_of(t * SP_TILE, S);
        SpTileMasks M = sp_tile_masks(planes + (size_t)t * SP_PLANES, t, g.m, S, sb, desc, slot, tb, du);
        for (int q = 0; q < sub; ++q) {
            const uint32_t a = (uint32_t)(q * glen);
            if (M.rst & sp_low(a)) { v = init; M.rst &= ~sp_low(a); }           // restarts before this sub-tile: already behind us
            st[(size_t)t * sub + q] = (uint16_t)v;                              // (a restart AT lane a is applied by whoever starts here)
            const sp_u64 range = sp_low(a + (uint32_t)glen) & ~sp_low(a);
            v = sp_walk(v, M.on & range, M.bm, M.rst, init, R, [](int, int, uint32_t) {});
        }
        if (M.rst) v = init;
    }
}

// Values, one lane of the tile's wavefront: lane -> (slot, sub-tile); the value every decision of that piece sees goes to
// rec[run lane][k] (k = the decision's index inside its run's rank side).
template <class Put>
DC_HD void sp_values(uint32_t tile, int lane, const SpGeom& g, const SpSub& S, const sp_u64* planes, const SpDesc* desc, const SpParams* P3 /* by class: RF, RE, RM */,
                     const uint16_t* state, Put&& put /* (run lane, k, value) */)
{
    int slot, q;
    sp_lane_map(lane, &slot, &q);
    if (slot < 0) return;
    const SpParams& P = P3[slot == 0 ? 0 : slot < 5 ? 1 : 2];
    const Rates R = P.R;
    const int init = P.init;
    const int sub = sp_slot_sub(slot), glen = SP_TILE / sub;
    const bool tb = sp_has_boundary(tile, tile + 1, S);
    const uint32_t sb = sp_sb_of(tile * SP_TILE, S);
    SpTileMasks M = sp_tile_masks(planes + (size_t)tile * SP_PLANES, tile, g.m, S, sb, desc, slot, tb);
    const uint32_t a = (uint32_t)(q * glen);
    const sp_u64 range = sp_low(a + (uint32_t)glen) & ~sp_low(a);
    sp_u64 rst = M.rst & range;                                               // restarts before lane a are in the recorded value already
    int v = state[sp_state_base(g, slot) + (size_t)tile * sub + q];
    const int k0 = desc[S.maxr[sb] * SP_SLOTS + slot].k;
    sp_walk(v, M.on & range, M.bm, rst, init, R, [&](int i, int val, uint32_t) {
        const int k = tb ? (int)desc[S.maxr[sp_sb_of(tile * SP_TILE + (uint32_t)i, S)] * SP_SLOTS + slot].k : k0;
        put(i, k, val);
    });
}

}  // namespace dcs
