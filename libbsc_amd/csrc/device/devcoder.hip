// devcoder.hip — the adaptive model of the static QLFC coder (-e1) on MI355X: every probability the range coder will
// need, computed on the GPU, so that the host is left with the range coder's own carry chain and nothing else.
//
// The reference codes a sub-block as ONE serial loop (libbsc/coder/qlfc/qlfc.cpp:829-1129): per run ~7 binary decisions,
// each predicted from three adaptive counters and coded at once.  The counters, though, are independent chains (see
// devcoder_model.h): chain = (sub-block, decision type, family, X), v <- step(v, bit) over the chain's decisions in stream
// order.  This file lays the decisions of a whole block (8 sub-blocks, ~2e8 decisions for 64 MiB of text) out chain by
// chain and walks all chains in parallel:
//
//   1. contexts      avg_rank >= 32 flags (two-sided bracket walk with warm-up), per-run packed items; a stable 8-bit
//                    radix pass by symbol gives the symbol-major order in which "previous run of the same symbol"
//                    (rank_hist / run_hist, qlfc.cpp:900, :981-987) is the neighbouring element; context states from the
//                    reference's two state tables; two more radix passes order the runs by rank-state and by run-state.
//   2. partition     for each family (static: stream order; char: symbol-major; state: state-major) the decisions of the
//                    runs, generated on the fly round by round (64 runs per wavefront, ballot-match ranking, no atomics on
//                    the order-defining path), are scattered into chain-major order: row = decision type, inside a row
//                    X-major, inside that stream order.  Each decision also records where it went (pos).
//   3. evaluation    8192-event chunks, one lane each.  A chunk that starts inside a chain does not know the chain's value
//                    there, but the counter maps are monotone, so running them from the two ends of the attainable range
//                    brackets the truth, and once the two trajectories meet everything after is exact (chunks are long
//                    enough that they nearly always meet; a chunk whose predecessor did not coalesce replays it serially —
//                    exactness never depends on luck).  Second walk with the exact start values writes the counter value
//                    every decision sees.
//   4. p stream      back in stream order: gather the three values of every decision through pos, blend (predictor.h:121),
//                    emit 16 bits per decision: probability, coded bit, start-of-run mark.
//
// The host (qlfc.cpp: encode_from_pstream) then runs only the range coder.  Whenever something is outside what this path
// handles exactly (more than 256 distinct decision types in a block, an avg_rank bracket that does not decide, capacity),
// a flag is raised and the caller falls back to the host model; nothing approximate is ever emitted.
#include "dev_common.h"
#include "dma_copy.h"
#include <thread>
#include <system_error>
#include "devcoder_model.h"
#include "devcoder_static.h"
#include "../host/qlfc.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

using namespace dcm;

// Run-parallel kernels that gather (dc_ctx: the symbol-major neighbours; dc_pstream: the chain-major counter values) can give every
// XCD one contiguous eighth of the runs instead of every eighth workgroup (block b runs on XCD b % 8), so that a line of the gathered
// arrays is fetched by one XCD's L2 only.  Measured on the 64 MiB bench block (same box, round 3): dc_pstream 3.10 against 2.92 ms,
// dc_ctx 1.44 against 1.41 ms — no gain (the lines are mostly consumed by one workgroup or its neighbour in time, and eight
// separate streams per array cost more than the shared fetches save); kept as an A/B switch, off.
#ifndef DC_XCD_RANGES
#define DC_XCD_RANGES 0
#endif
__device__ __forceinline__ u32 dc_virtual_block() {
    if (!DC_XCD_RANGES) return blockIdx.x;
    return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
}

constexpr int DC_WCH_MAX  = 4096;      // wave-chunks of a partition job (one wavefront walks one chunk of runs)
#ifndef DC_EV_N
#define DC_EV_N 8192
#endif
constexpr int DC_EV       = DC_EV_N;   // events per evaluation chunk: the minimum (see devcoder_pstream: long enough for the brackets to meet)
constexpr int DC_AVG_CH   = 1024;      // runs per avg_rank lane
constexpr int DC_AVG_WARM = 768;       // warm-up runs in front of an avg_rank chunk
constexpr u32 DC_SIGMASK  = 0x7ffu;    // event = X | sub-block << 8 | bit << 11
constexpr int DC_ROWS     = 1088;      // rows of the chain-major layout = decision types (NUM_TAU = 1080), padded to whole wavefronts

// meta scalars (device u32 array)
enum { DM_FAIL = 0, DM_NTYPES, DM_NROUNDS, DM_AVG_UND, DM_REPLAYS, DM_D0, DM_D1, DM_D2, DM_D3, DM_HIST_FAIL, DM_DFULL /* = DM_D0 + 5 */, DM_SP_OPEN, DM_SP_REWALKS, DM_P13_OVER /* a wavefront's piece did not fit the staging buffer: the packed stream is void */, DM_COUNT = 16 };
enum { FAIL_TYPES = 1, FAIL_AVG = 2, FAIL_HIST = 4, FAIL_CAP = 8, FAIL_REPLAY = 16 };

typedef dcs::SpSub DcSub;                                       // { u32 nb; u32 first[9]; u32 maxr[8]; }: run index range and max_rank of each sub-block
struct DcRowBins { u16 lo1, hi1, lo2, hi2; };                   // counting pass: count(row) = P[hi1] - P[lo1] + P[hi2] - P[lo2] over bin prefix sums

struct DevCoder {
    size_t Mcap = 0, Dcap = 0;
    char*  arena = nullptr; size_t arena_bytes = 0;
    u64 *key_ch = nullptr, *key_ch_s = nullptr, *key_sr = nullptr, *key_sr_s = nullptr, *key_sn = nullptr, *key_sn_s = nullptr;
    u32 *inv_ch = nullptr, *inv_sr = nullptr, *inv_sn = nullptr;
    u8  *ge32 = nullptr;
    u32 *doff[4] = {nullptr, nullptr, nullptr, nullptr};       // per job: sp, ch, sr, sn
    u16 *events[4] = {nullptr, nullptr, nullptr, nullptr};     // per job, chain-major
    u32 *pos[4] = {nullptr, nullptr, nullptr, nullptr};        // per job: where each decision's event went
    u16 *V[4] = {nullptr, nullptr, nullptr, nullptr};          // per job: counter value seen by each event
    size_t nch_cap = 0;                                        // evaluation chunks per job (capacity)
    u16 *ps[2] = {nullptr, nullptr};                           // double buffer: block i's copy-out overlaps block i+1's kernels
    u32 *cnt = nullptr, *rowtot = nullptr, *rowstart = nullptr /*[4][257]*/, *wdec = nullptr, *wdecoff = nullptr;
    u16 *elo = nullptr, *ehi = nullptr, *S = nullptr;
    u32 *present = nullptr; u8 *rounds = nullptr; u32 *meta = nullptr; u32 *poff = nullptr;
    DcRowBins* rowbins = nullptr;                              // row -> bin ranges of the counting pass
    u16* sink = nullptr;                                       // where stores past the end of an array go (1 KB)
    u8  *tab_rank = nullptr, *tab_run = nullptr;
    ModelParams* mp = nullptr;                                 // device copy (static coder)
    ModelParams* mp_fast = nullptr;                            // device copy (fast coder: dcm::model_params_fast)
    // the static family in stream order (devcoder_static.h): rank planes, per-sub-block descriptors, chunk summaries, chunk start
    // values, sub-tile values, the 8 x u16 record of every run, and the runs' offsets in the p stream
    u64* sp_planes = nullptr; dcs::SpDesc* sp_desc = nullptr; dcs::SpSum* sp_sums = nullptr; dcs::SpGroupSum* sp_gsum = nullptr; u16* sp_gv = nullptr; u16* sp_sv = nullptr; u16* sp_state = nullptr;
    uint4* sp_rec = nullptr; u32* doff_full = nullptr;
    struct DcFrag* frag = nullptr;                              // packed stream: two fragment records per wavefront of dc_pstream (DcP13)
    char* sp_arena = nullptr; bool sp_alloc_failed = false;     // (own allocation, made when BSCGPU_OPT_DC_STREAM_STATIC is first used: dc_sp_ensure)
    dcs::SpDesc sp_desc_host[5][dcs::SP_SLOTS]; bool sp_ok = false;   // descriptors by max_rank (built once; sp_ok: every type representable)
    u32 *hmeta = nullptr;                                      // pinned: meta + poff
};

__device__ __forceinline__ u32 dc_sb_of(u32 j, const DcSub& S)
{
    u32 sb = 0;
#pragma unroll
    for (int b = 1; b < 8; ++b) if ((u32)b < S.nb && j >= S.first[b]) sb = b;
    return sb;
}
// max_rank of an item's sub-block WITHOUT a memory access: indexing the by-value kernel argument with a per-lane index is a vector load
// from the argument segment (a select chain over S.maxr[b] is turned back into a load of a selected address), and inside the partition
// kernels' tile loops the wait for it (a vmcnt(0): the stores around it cannot be counted) also waited out the request for the next
// tile's items that had just gone out — round 6, from the ISA.  So: the eight values packed into one scalar word at the kernel's top.
__device__ __forceinline__ u32 dc_maxr_pack(const DcSub& S)
{
    u32 p = 0;
#pragma unroll
    for (u32 b = 0; b < 8; ++b) p |= (S.maxr[b] & 15u) << (4u * b);
    return (u32)__builtin_amdgcn_readfirstlane((int)p);
}
__device__ __forceinline__ int dc_maxr_of(u32 sb, u32 packed) { return (int)((packed >> (4u * sb)) & 15u); }
__device__ __forceinline__ u32 dc_run_len(const u32* __restrict__ start, u32 j, u32 m, u32 n) { return ((j + 1 < m) ? start[j + 1] : n) - start[j]; }

// ---------------------------------------------------------------------------------------------------------------------
// 1a. avg_rank >= 32 (qlfc.cpp:903 / :978): avg' = (avg * 124 + rank * 4) >> 7, reset per sub-block.  One lane per chunk,
// two-sided bracket [0, 255] started DC_AVG_WARM runs early; the flag of a run is decided when both ends agree.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void dc_avg_kernel(const u8* __restrict__ rank, u32 m, DcSub S, u8* __restrict__ ge32, u32* __restrict__ meta)
{
    const u32 c = blockIdx.x * WG + threadIdx.x;
    const u64 j0 = (u64)c * DC_AVG_CH;
    if (j0 >= m) return;
    const u32 j1 = (u32)((j0 + DC_AVG_CH < m) ? j0 + DC_AVG_CH : m);
    u32 sb = dc_sb_of((u32)j0, S);
    const u32 sb_first = S.first[sb];
    u32 w0 = ((u32)j0 > sb_first + DC_AVG_WARM) ? (u32)j0 - DC_AVG_WARM : sb_first;
    u32 lo = 0, hi = (w0 == sb_first) ? 0u : 255u;
    for (u32 j = w0; j < (u32)j0; ++j) { const u32 r = rank[j]; lo = avg_rank_next(lo, r); hi = avg_rank_next(hi, r); }
    u32 next_first = (sb + 1 < S.nb) ? S.first[sb + 1] : 0xffffffffu;
    u32 und = 0;
    for (u32 j = (u32)j0; j < j1; ++j) {
        if (j == next_first) { lo = hi = 0; ++sb; next_first = (sb + 1 < S.nb) ? S.first[sb + 1] : 0xffffffffu; }
        const u32 f = lo >= 32u;
        und += (f != (u32)(hi >= 32u));
        ge32[j] = (u8)f;
        const u32 r = rank[j];
        lo = avg_rank_next(lo, r); hi = avg_rank_next(hi, r);
    }
    if (und) atomicAdd(&meta[DM_AVG_UND], und);
}

// 1b. packed items in stream order: X = symbol (the char family's sort digit; the static family ignores it)
// (+ the rank bit planes of every tile of 64 runs, when the static family is evaluated in stream order: devcoder_static.h)
__global__ __launch_bounds__(WG) void dc_items_kernel(const u8* __restrict__ sym, const u8* __restrict__ rank, const u32* __restrict__ start,
                                                      const u8* __restrict__ ge32, u32 m, u32 n, DcSub S, u64* __restrict__ key_ch, u64* __restrict__ planes)
{
    const u32 j = blockIdx.x * WG + threadIdx.x;
    const u32 r = (j < m) ? rank[j] : 0u;
    if (j < m) key_ch[j] = item_pack(sym[j], dc_sb_of(j, S), ge32[j], r, dc_run_len(start, j, m, n));
    if (planes) {                                                     // (wave-uniform; a wavefront is a tile: WG is a multiple of 64)
        const u32 lane = threadIdx.x & 63u;
        u64 mine = 0;
#pragma unroll
        for (int b = 0; b < dcs::SP_PLANES; ++b) { const u64 bal = __ballot((r >> b) & 1u); if (lane == (u32)b) mine = bal; }
        const u32 tile = j >> 6;
        if (lane < (u32)dcs::SP_PLANES && (tile << 6) < m) planes[(size_t)tile * dcs::SP_PLANES + lane] = mine;
    }
}

// kinds of run for the "which decision types occur" bitmap: 8 sub-blocks x escape flag x 256 ranks, then 96 run-length classes
constexpr u32 DC_KIND_RUN = 8 * 512, DC_KIND_WORDS = (DC_KIND_RUN + 96 + 31) / 32;

// 1c. context states + the state family's items + which kinds of run occur.  Thread per run (stream order).
__global__ __launch_bounds__(WG) void dc_ctx_kernel(const u64* __restrict__ key_ch, const u64* __restrict__ key_ch_s, const u32* __restrict__ inv_ch,
                                                    u32 m, DcSub S, const u8* __restrict__ tab_rank, const u8* __restrict__ tab_run,
                                                    u64* __restrict__ key_sr, u64* __restrict__ key_sn, u32* __restrict__ present, u32* __restrict__ meta)
{
    __shared__ u32 bits[DC_KIND_WORDS];
    for (u32 i = threadIdx.x; i < DC_KIND_WORDS; i += WG) bits[i] = 0;
    __syncthreads();
    const u32 j = dc_virtual_block() * WG + threadIdx.x;
    if (j < m) {
        const u64 key = key_ch[j];
        const Item it = item_unpack(key);
        const u32 j0 = S.first[it.sb];
        // window contexts: previous runs of the same sub-block (qlfc.cpp:1063-1068)
        u32 ctx_rank0 = 0, ctx_rank4 = 0, ctx_run = 0;
#pragma unroll
        for (u32 k = 1; k <= 4; ++k) {
            if (j >= j0 + k) {
                const Item p = item_unpack(key_ch[j - k]);
                if (k <= 3) ctx_rank0 |= (p.rank == 1u ? 1u : 0u) << (k - 1);
                ctx_rank4 |= (p.rank - 1u < 3u ? p.rank - 1u : 3u) << (2 * (k - 1));
                ctx_run |= (p.run < 3u ? 1u : 0u) << (k - 1);
            }
        }
        // symbol-major neighbours: the previous runs of this symbol in this sub-block, nearest first (loaded once)
        const u32 q = inv_ch[j];
        const u64 chain_id = key >> 53;                               // X and sub-block
        // (all NP loads are issued at once — the index does not depend on what the nearer neighbours turn out to be — and NP is large
        // enough that the bracket below almost always closes: on the bench block 1.9 % of the runs — a lane in 38 % of the wavefronts —
        // stay open after five predecessors (a symbol whose recent runs all have length 2 or 3 has two fixed points, 1 and 2), 0.015 %
        // after nine)
        constexpr int NP = 9;
        u64 pk[NP];
        if (q >= 10u) {
            // the nine predecessors are 72 consecutive bytes: five 16-byte loads instead of nine 8-byte ones (round 6: the kernel is bound by
            // the lines its gathers touch per instruction, and these nine touched the same one or two lines nine times)
            struct __attribute__((packed, aligned(8))) U2 { u64 a, b; };
            const U2* w = reinterpret_cast<const U2*>(key_ch_s + (q - 10u));
            const U2 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];            // elements q-10 .. q-1
            pk[8] = w0.b; pk[7] = w1.a; pk[6] = w1.b; pk[5] = w2.a; pk[4] = w2.b; pk[3] = w3.a; pk[2] = w3.b; pk[1] = w4.a; pk[0] = w4.b;
        } else {
#pragma unroll
            for (int t = 1; t <= NP; ++t) pk[t - 1] = key_ch_s[q >= (u32)t ? q - (u32)t : 0u];
        }
        u32 prun[NP], prank0 = 0; int np = 0; bool at_start = false;
#pragma unroll
        for (int t = 1; t <= NP; ++t) {
            bool have = false;
            if (!at_start && q >= (u32)t && (pk[t - 1] >> 53) == chain_id) {
                const Item pi = item_unpack(pk[t - 1]); prun[t - 1] = pi.run; if (t == 1) prank0 = pi.rank; have = true; np = t;
            }
            if (!have) { at_start = true; prun[t - 1] = 1; }
        }
        const u32 rank_hist = np > 0 ? (u32)bsr(prank0) : 0u;
        // run_hist (qlfc.cpp:981-987): h' = (h + x) >> 2 over the symbol's earlier runs; only min(h, 7) is used.  Walk a two-sided
        // bracket forward over the loaded predecessors; the start is exact (0) when the chain begins inside the window.
        u32 run_hist = 0;
        {
            u32 lo = 0, hi = at_start ? 0u : 63u;
#pragma unroll
            for (int t = NP; t >= 1; --t) if (t <= np) { lo = run_hist_next(lo, prun[t - 1]); hi = run_hist_next(hi, prun[t - 1]); }
            u32 cl = lo < 7u ? lo : 7u, ch = hi < 7u ? hi : 7u;
            if (cl != ch) {
                // rare: look further back (quadrupling) until the clamped values agree or the chain starts
                u32 K = 4 * NP;
                for (;;) {
                    lo = 0; hi = 63; u32 first = q; bool exact = false;
                    for (u32 t = 1; t <= K; ++t) {
                        if (q < t || (key_ch_s[q - t] >> 53) != chain_id) { exact = true; break; }
                        first = q - t;
                    }
                    if (exact) hi = 0;
                    for (u32 p = first; p < q; ++p) { const u32 r = item_unpack(key_ch_s[p]).run; lo = run_hist_next(lo, r); hi = run_hist_next(hi, r); }
                    cl = lo < 7u ? lo : 7u; ch = hi < 7u ? hi : 7u;
                    if (cl == ch) break;
                    if (K >= 4096) { atomicOr(&meta[DM_FAIL], (u32)FAIL_HIST); break; }
                    K *= 4;
                }
            }
            run_hist = cl;
        }
        const u32 state_rank = tab_rank[rank_state_index(ctx_run, ctx_rank4, rank_hist)];
        const u32 state_run = tab_run[run_state_index(ctx_rank0, ctx_run, it.rank, run_hist)];
        const u64 info = key & 0x00ffffffffffffffull;
        key_sr[j] = ((u64)state_rank << 56) | info;
        key_sn[j] = ((u64)state_run << 56) | info;

        // which kinds of run occur: the decision types of a run are a function of (sub-block, escape flag, rank) on the rank side
        // and of the run-length class on the run side (dc_setup_kernel expands the kinds that occur into types)
        auto mark = [&](u32 k) { const u32 w = k >> 5, b = 1u << (k & 31); if (!(bits[w] & b)) atomicOr(&bits[w], b); };
        mark((it.sb << 9) | (it.ge32 << 8) | it.rank);
        mark(DC_KIND_RUN + (it.run < 64u ? it.run : 64u + (u32)bsr(it.run)));
    }
    __syncthreads();
    // Bits are only ever set during the launch, so a plain look first is safe (a stale zero costs one atomic, a one is a one): without it
    // every workgroup — 110 K of them — sent ~20 atomics to the same few dozen words, and those serialise at ~11 ns apiece.
    for (u32 i = threadIdx.x; i < DC_KIND_WORDS; i += WG) {
        const u32 mine = bits[i];
        if (mine) { const u32 need = mine & ~__builtin_nontemporal_load(&present[i]); if (need) atomicOr(&present[i], need); }
    }
}

// 1d. the canonical rounds that occur (and how many decision types: diagnostics).  One workgroup.
__global__ __launch_bounds__(WG) void dc_setup_kernel(const u32* __restrict__ kinds, DcSub S, u8* __restrict__ rounds, u32* __restrict__ meta)
{
    const u32 mrp = dc_maxr_pack(S);                                  // max_rank of the eight sub-blocks, one scalar word
    __shared__ u32 present[(NUM_TAU + 31) / 32];
    __shared__ u32 rbits[3];
    __shared__ u32 ntypes;
    for (u32 i = threadIdx.x; i < (NUM_TAU + 31) / 32; i += WG) present[i] = 0;
    if (threadIdx.x < 3) rbits[threadIdx.x] = 0;
    if (threadIdx.x == 0) ntypes = 0;
    __syncthreads();
    auto mark = [&](int tau) { atomicOr(&present[(u32)tau >> 5], 1u << (tau & 31)); };
    for (u32 k = threadIdx.x; k < DC_KIND_RUN + 96; k += WG) {
        if (!(kinds[k >> 5] & (1u << (k & 31)))) continue;
        Item it; it.sb = 0; it.ge32 = 0; it.rank = 1; it.run = 1;
        if (k < DC_KIND_RUN) {                                                    // the rank side of a run of this kind
            it.sb = k >> 9; it.ge32 = (k >> 8) & 1u; it.rank = k & 255u;
            const int maxr = dc_maxr_of(it.sb, mrp);
            if (it.ge32) { for (int d = 0; d <= maxr; ++d) { u32 bit; mark(decision(it, maxr, ROUND_RP + d, &bit)); } }
            else {
                mark(TAU_RF);
                if (it.rank != 1u) {
                    const int B = bsr(it.rank);
                    for (int s = 0; s <= B - 2; ++s) mark(TAU_RE + s);
                    if (B < maxr) mark(TAU_RE + B - 1);
                    for (int d = 0; d < B; ++d) { u32 bit; mark(decision(it, maxr, ROUND_RM + d, &bit)); }
                }
            }
        } else {                                                                  // the run side: a representative length of the class
            const u32 cls = k - DC_KIND_RUN;
            it.run = cls < 64u ? cls : 1u << (cls - 64u);
            mark(TAU_NF);
            if (it.run != 1u) {
                const int nb = bsr(it.run);
                for (int s = 0; s < nb; ++s) mark(TAU_NE + s);
                for (int d = 0; d < nb; ++d) { u32 bit; mark(decision(it, 0, ROUND_NM + d, &bit)); }
            }
        }
    }
    __syncthreads();
    for (int tau = threadIdx.x; tau < NUM_TAU; tau += WG) {
        if (present[(u32)tau >> 5] & (1u << (tau & 31))) {
            const int r = tau_round(tau);
            atomicOr(&rbits[r >> 5], 1u << (r & 31));
            atomicAdd(&ntypes, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 nr = 0;
        for (int r = 0; r < NUM_ROUNDS; ++r) if (rbits[r >> 5] & (1u << (r & 31))) rounds[nr++] = (u8)r;
        meta[DM_NTYPES] = ntypes; meta[DM_NROUNDS] = nr;
        if (meta[DM_AVG_UND] != 0u) atomicOr(&meta[DM_FAIL], (u32)FAIL_AVG);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. partition: decisions of a job's items into chain-major order
// ---------------------------------------------------------------------------------------------------------------------
struct DcGeom { u32 m; u32 W; u32 per_wave; };                       // items, wave-chunks, items per wave-chunk (multiple of 64)
static DcGeom dc_geom(u32 m)
{
    DcGeom g; g.m = m;
    u32 W = (m + 63) / 64; if (W > DC_WCH_MAX) W = DC_WCH_MAX; if (W == 0) W = 1;
    u32 per = (m + W - 1) / W; per = (per + 63) / 64 * 64; if (per == 0) per = 64;
    g.per_wave = per; g.W = (m + per - 1) / per; if (g.W == 0) g.W = 1;
    return g;
}

typedef __attribute__((address_space(3))) volatile u32 dc_lds_vu32;

// Lanes among `on` whose NB-bit key equals this lane's: (lo, hi) halves of the peer mask.
template <int NB>
__device__ __forceinline__ void dc_peers(u32 key, bool on, u32& mlo, u32& mhi)
{
    const u64 act = __ballot(on);
    mlo = (u32)act; mhi = (u32)(act >> 32);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int bitm = __builtin_amdgcn_sbfe((int)key, b, 1);              // 0 or -1
        const u64 bal = __ballot(bitm != 0);
        const u32 nbm = ~(u32)bitm;
        mlo &= (u32)bal ^ nbm;
        mhi &= (u32)(bal >> 32) ^ nbm;
    }
}
// One level down a code tree: of the peers keep those that coded the same bit (`keep` = all ones: no refinement for this lane).
__device__ __forceinline__ void dc_refine(u32 bit, u32 keep, u32& mlo, u32& mhi)
{
    const u64 bal = __ballot(bit != 0);
    const u32 nbm = bit ? 0u : ~0u;
    mlo &= ((u32)bal ^ nbm) | keep;
    mhi &= ((u32)(bal >> 32) ^ nbm) | keep;
}

// The decisions of 64 items, round by round in canonical order, with the case analysis done once per item instead of once per
// round: single-row rounds (RF, RE s, NF, NE s: one decision type, hence one row, per round) go to es(slot, on, bit) with
// slot = 0 | 1 + s | 8 | 9 + s; mantissa / escape rounds (several rows per round) go to em(tau, on, bit, peers) where peers is
// the mask of the lanes whose decision of this round has the same type.  The rows of such a round are the nodes of one level
// of a code tree: two lanes are peers at depth d + 1 iff they were peers at depth d and coded the same bit there, so the mask
// costs one ballot per round after a single match on the tree's id (rank length B / run-length bits).  Control flow is
// wave-uniform; every lane calls with its own `on`.
constexpr int DC_SLOTS = 40;
__device__ __forceinline__ int dc_slot_tau(int slot) { return slot == 0 ? TAU_RF : slot < 8 ? TAU_RE + slot - 1 : slot == 8 ? TAU_NF : TAU_NE + slot - 9; }

template <int SIDES, class ES, class EM>
__device__ __forceinline__ void dc_item_rounds(const Item& it, bool valid, int maxr, ES&& es, EM&& em)
{
    if (SIDES & 1) {
        const u32 rank = it.rank;
        const bool ge = valid && it.ge32 != 0u, ng = valid && it.ge32 == 0u;
        const int B = (ng && rank != 1u) ? bsr(rank) : 0;
        const int e = B ? (B - 1) + (B < maxr ? 1 : 0) : 0;
        if (__ballot(ng)) es(0, ng, rank != 1u ? 1u : 0u);
        for (int sx = 0; sx < 7; ++sx) { const bool on = sx < e; if (!__ballot(on)) break; es(1 + sx, on, sx + 1 < B ? 1u : 0u); }
        if (__ballot(B != 0)) {
            u32 mlo, mhi;
            dc_peers<3>((u32)B, B != 0, mlo, mhi);
            for (int d = 0; d < 7; ++d) {
                const bool on = d < B;
                const u64 act = __ballot(on);
                if (!act) break;
                const u32 ctx = on ? (rank >> (B - d)) : 1u;
                const u32 bit = on ? (rank >> (B - 1 - d)) & 1u : 0u;
                em(on ? TAU_RM + rm_off(B) + (int)ctx - 1 : 0, on, bit, mlo & (u32)act, mhi & (u32)(act >> 32));
                dc_refine(bit, 0u, mlo, mhi);
            }
        }
        if (__ballot(ge)) {
            u32 mlo = ~0u, mhi = ~0u;                                           // depth 0: one row (ctx = 1)
            for (int d = 0; d < 8; ++d) {
                const bool on = ge && d <= maxr;
                const u64 act = __ballot(on);
                if (!act) break;
                const u32 ctx = on ? ((1u << d) | ((rank >> (maxr + 1 - d)) & ((1u << d) - 1u))) : 1u;
                const u32 bit = on ? (rank >> (maxr - d)) & 1u : 0u;
                em(TAU_RP + (int)ctx - 1, on, bit, mlo & (u32)act, mhi & (u32)(act >> 32));
                dc_refine(bit, 0u, mlo, mhi);
            }
        }
    }
    if (SIDES & 2) {
        const u32 run = it.run;
        const int nb = (valid && run != 1u) ? bsr(run) : 0;
        if (!(SIDES & 4) && __ballot(valid)) es(8, valid, run != 1u ? 1u : 0u);     // (SIDES & 4: the run side without its first decision, NF)
        for (int sx = 0; sx < 31; ++sx) { const bool on = sx < nb; if (!__ballot(on)) break; es(9 + sx, on, sx + 1 < nb ? 1u : 0u); }
        if (__ballot(nb != 0)) {
            u32 mlo, mhi;
            dc_peers<5>((u32)nb, nb != 0, mlo, mhi);
            const u32 keep = nb > 5 ? ~0u : 0u;                                 // more than 5 bits: one row per depth, no tree
            for (int d = 0; d < 31; ++d) {
                const bool on = d < nb;
                const u64 act = __ballot(on);
                if (!act) break;
                const u32 ctx = on ? (nb <= 5 ? (run >> (nb - d)) : (u32)(1 + d)) : 1u;
                const u32 bit = on ? (run >> (nb - 1 - d)) & 1u : 0u;
                em(on ? TAU_NM + nm_off(nb) + (int)ctx - 1 : 0, on, bit, mlo & (u32)act, mhi & (u32)(act >> 32));
                dc_refine(bit, keep, mlo, mhi);
            }
        }
    }
}

// 2a. per wave-chunk: decisions per row and in total.  Which rows a run contributes to is a function of (rank, B < max_rank)
// on the rank side and of the run length's class on the run side, and every row collects a contiguous range of those values
// (a tree node = a prefix of the value): so the wavefront only histograms its runs into DC_BINS bins (two LDS atomics per run)
// and turns the prefix sums of the bins into row counts at the end through a table built on the host from the model itself
// (dc_build_rowbins).  Runs under escape coding (avg_rank >= 32; their rows depend on max_rank) are counted directly.
constexpr int DC_BIN_RUN = 512;        // bins [0, 256): rank, B >= max_rank or rank < 2; [256, 512): rank, B < max_rank; [512, 608): run classes
constexpr int DC_BINS = 640;           // padded to 10 per lane
__host__ __device__ __forceinline__ u32 dc_rank_bin(u32 rank, int maxr) { const int B = rank != 1u ? bsr(rank) : 0; return rank | ((B != 0 && B < maxr) ? 256u : 0u); }
__host__ __device__ __forceinline__ u32 dc_run_bin(u32 run) { return (u32)DC_BIN_RUN + (run < 64u ? run : 64u + (u32)bsr(run)); }

template <int SIDES>
__global__ __launch_bounds__(WG) void dc_part_count_kernel(const u64* __restrict__ items, DcGeom g, DcSub S, const DcRowBins* __restrict__ rowbins,
                                                           u32* __restrict__ cnt /*[DC_ROWS][W]*/, u32* __restrict__ wdec)
{
    const u32 mrp = dc_maxr_pack(S);                                  // max_rank of the eight sub-blocks, one scalar word
    __shared__ u32 bins[WAVES][DC_BINS + 8];
    __shared__ u32 hrp[WAVES][256];                                             // escape rows (TAU_RP + ctx - 1), counted directly
    for (u32 i = threadIdx.x; i < WAVES * (DC_BINS + 8); i += WG) (&bins[0][0])[i] = 0;
    for (u32 i = threadIdx.x; i < WAVES * 256; i += WG) (&hrp[0][0])[i] = 0;
    __syncthreads();
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 wc = blockIdx.x * WAVES + w;
    u32* bw = &bins[w][0];
    u32 total = 0;
    if (wc < g.W) {
        const u64 i0 = (u64)wc * g.per_wave;
        u64 i1 = i0 + g.per_wave; if (i1 > g.m) i1 = g.m;
        // four items per lane and trip, their loads issued together: with one load in flight per wavefront the ~108 trips of a wave-chunk
        // each waited out a full memory latency (0.17 ms per job for 225 MB: round 5 counters, 80 % of the wave cycles parked)
        // (round 6: the NEXT trip's four loads are in flight while this trip's items are counted — nothing but LDS atomics between them, so
        // the compiler waits with an exact vmcnt; before, every one of a wave-chunk's ~27 trips waited out a full memory round trip)
        u64 nx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const u64 i = i0 + 64u * (u32)u + lane; nx[u] = items[i < i1 ? i : i0]; }
        for (u64 base = i0; base < i1; base += 256) {
            u64 kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kk[u] = nx[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const u64 i = base + 256u + 64u * (u32)u + lane; nx[u] = items[i < i1 ? i : i0]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + 64u * (u32)u + lane >= i1) continue;
                const Item it = item_unpack(kk[u]);
                const int maxr = dc_maxr_of(it.sb, mrp);
                if (SIDES & 1) {
                    total += (u32)count_rank_side(it, maxr);
                    if (it.ge32) {
                        for (int d = 0; d <= maxr; ++d) atomicAdd(&hrp[w][((1u << d) | ((it.rank >> (maxr + 1 - d)) & ((1u << d) - 1u))) - 1u], 1u);
                    } else atomicAdd(&bw[dc_rank_bin(it.rank, maxr)], 1u);
                }
                if (SIDES & 2) { total += (u32)count_run_side(it) - ((SIDES & 4) ? 1u : 0u); atomicAdd(&bw[dc_run_bin(it.run)], 1u); }
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix sums of the wave's bins, 10 consecutive bins per lane
        u32 loc[DC_BINS / 64], sum = 0;
#pragma unroll
        for (int q = 0; q < DC_BINS / 64; ++q) { loc[q] = bw[lane * (DC_BINS / 64) + q]; sum += loc[q]; }
        u32 run = wave_incl_sum(sum) - sum;
#pragma unroll
        for (int q = 0; q < DC_BINS / 64; ++q) { bw[lane * (DC_BINS / 64) + q] = run; run += loc[q]; }
    }
    total = wave_incl_sum(total);
    __syncthreads();
    if (wc < g.W) {
        for (u32 h = lane; h < (u32)DC_ROWS; h += 64) {
            const DcRowBins rb = rowbins[h];
            u32 v = bw[rb.hi1] - bw[rb.lo1] + bw[rb.hi2] - bw[rb.lo2];
            if (h >= (u32)TAU_RP && h < (u32)TAU_NF) v += hrp[w][h - TAU_RP];
            if ((SIDES & 4) && h == (u32)TAU_NF) v = 0;
            cnt[(size_t)h * g.W + wc] = v;
        }
        if (lane == 63) wdec[wc] = total;
    }
}

// 2b. scans: rows over wave-chunks (one workgroup per row), then row bases and the wave-chunks' decision offsets
__global__ __launch_bounds__(WG) void dc_scan_rows_kernel(u32* __restrict__ cnt, u32 W, u32* __restrict__ rowtot)
{
    __shared__ u32 scr[8];
    u32* row = cnt + (size_t)blockIdx.x * W;
    u32 carry = 0;
    for (u32 base = 0; base < W; base += WG) {
        const u32 i = base + threadIdx.x;
        const u32 v = (i < W) ? row[i] : 0u;
        u32 tot;
        const u32 ex = block_excl_sum(v, scr, &tot);
        if (i < W) row[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}
__global__ __launch_bounds__(WG) void dc_scan_misc_kernel(const u32* __restrict__ rowtot, u32* __restrict__ rowstart /*[DC_ROWS + 1]*/,
                                                          const u32* __restrict__ wdec, u32 W, u32* __restrict__ wdecoff /*[W+1]*/,
                                                          u32* __restrict__ meta, int job, u32 Dcap)
{
    __shared__ u32 scr[8];
    u32 rc = 0;
    for (u32 base = 0; base < (u32)DC_ROWS; base += WG) {
        const u32 i = base + threadIdx.x;
        const u32 v = (i < (u32)DC_ROWS) ? rowtot[i] : 0u;
        u32 t1;
        const u32 e1 = block_excl_sum(v, scr, &t1);
        if (i < (u32)DC_ROWS) rowstart[i] = rc + e1;
        rc += t1;
        __syncthreads();
    }
    if (threadIdx.x == 0) { rowstart[DC_ROWS] = rc; meta[DM_D0 + job] = rc; if (rc > Dcap) atomicOr(&meta[DM_FAIL], (u32)FAIL_CAP); }
    u32 carry = 0;
    for (u32 base = 0; base < W; base += WG) {
        const u32 i = base + threadIdx.x;
        const u32 v = (i < W) ? wdec[i] : 0u;
        u32 t2;
        const u32 e2 = block_excl_sum(v, scr, &t2);
        if (i < W) wdecoff[i] = carry + e2;
        carry += t2;
        __syncthreads();
    }
    if (threadIdx.x == 0) wdecoff[W] = carry;
}

// lane `lane` of v <- val, both wave-uniform (v_writelane takes one SGPR operand besides m0)
__device__ __forceinline__ int dc_writelane(int v, int val, int lane)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val), "s"(lane) : "m0");
    return v;
}

// 2c. scatter.  One wavefront per wave-chunk walks its items 64 at a time; per canonical round the lanes that have a decision
// are ranked stably inside their row (same type -> same round, so rows never interleave across rounds) and write
//   events[row offset]            = X | sub-block << 8 | bit << 11
//   pos[decision index of item]   = that offset (index into the job's value array)
// The running offsets of the 40 single-row types live in one VGPR (lane = slot; v_readlane with a uniform slot), those of the
// mantissa / escape rows in LDS.  pos is item-major, i.e. every lane owns a short run of it and a direct store per round would
// touch ~20 lines per wavefront instruction (the address path, not the bytes, was the kernel's limit): the tile's entries are
// collected in LDS and leave as whole lines when the tile is done (a tile with more decisions than the stage holds stores directly).
#ifndef DC_POS_STAGE_N
#define DC_POS_STAGE_N 1280
#endif
constexpr u32 DC_POS_STAGE = DC_POS_STAGE_N;
struct __attribute__((packed, aligned(4))) DcPos4 { u32 a, b, c, d; };      // four positions, 4-byte aligned: one dwordx4 store
template <int SIDES>
__global__ __launch_bounds__(WG) void dc_part_scatter_kernel(const u64* __restrict__ items, DcGeom g, DcSub S, const u32* __restrict__ meta,
                                                             const u32* __restrict__ cnt, const u32* __restrict__ rowstart,
                                                             const u32* __restrict__ wdecoff, u32 ignoreX,
                                                             u16* __restrict__ events, u32* __restrict__ pos, u32* __restrict__ doff)
{
    const u32 mrp = dc_maxr_pack(S);                                  // max_rank of the eight sub-blocks, one scalar word
    __shared__ u32 goff[WAVES][DC_ROWS];
    __shared__ u32 spos[WAVES][DC_POS_STAGE];
    if (meta[DM_FAIL] != 0u) return;
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 wc = blockIdx.x * WAVES + w;
    if (wc < g.W) for (u32 h = lane; h < (u32)DC_ROWS; h += 64) goff[w][h] = rowstart[h] + cnt[(size_t)h * g.W + wc];
    __syncthreads();
    if (wc >= g.W) return;
    dc_lds_vu32* vg = (dc_lds_vu32*)&goff[w][0];
    dc_lds_vu32* sp = (dc_lds_vu32*)&spos[w][0];
    int sreg = (lane < (u32)DC_SLOTS) ? (int)goff[w][dc_slot_tau((int)lane)] : 0;     // running offsets of the single-row types
    const u64 i0 = (u64)wc * g.per_wave;
    u64 i1 = i0 + g.per_wave; if (i1 > g.m) i1 = g.m;
    u32 running = wdecoff[wc];                                        // decision index of the tile's first item
    u64 knext = (i0 + lane < i1) ? items[i0 + lane] : 0ull;           // the next tile's items are requested one tile ahead
    for (u64 base = i0; base < i1; base += 64) {
        const u64 i = base + lane;
        const bool valid = i < i1;
        const u64 key = knext;
        // The request for the next tile's items goes out BEHIND the wait for this tile's.  (Round 6, from the ISA: the compiler cannot count the
        // stores of the rounds below — they sit in data-dependent loops — so the wait for `key` is a vmcnt(0); with the request in front of
        // it, as the source order had it, that wait covered the request itself: no prefetch at all, one full memory round trip per tile and
        // wavefront, 0.35 ms of every partition job whatever it had to scatter.)
        asm volatile("" :: "v"(key) : "memory");
        knext = (i + 64 < i1) ? items[i + 64] : 0ull;
        const Item it = item_unpack(key);
        const int maxr = dc_maxr_of(it.sb, mrp);
        u32 nd = 0;
        if (valid) { if (SIDES & 1) nd += (u32)count_rank_side(it, maxr); if (SIDES & 2) nd += (u32)count_run_side(it) - ((SIDES & 4) ? 1u : 0u); }
        const u32 incl = wave_incl_sum(nd);
        const u32 loc = incl - nd;                                    // first decision of this item inside the tile
        if (valid) doff[i] = running + loc;
        const u32 tile_total = (u32)__builtin_amdgcn_readlane((int)incl, 63);      // a scalar: `staged` below is then a scalar
        const bool staged = tile_total <= DC_POS_STAGE;               // branch around every round's position store, not an exec-mask dance
        const u32 sig = (ignoreX ? 0u : item_X(key)) | (it.sb << 8);
        u32* mypos = pos + running + loc;
        // (two copies of the rounds, chosen once per tile: with the test inside `put` every round carried a branch around its position store)
        auto rounds = [&](auto staged_tag) __attribute__((always_inline)) {
            constexpr bool STAGED = decltype(staged_tag)::value;
            u32 ord = 0;
            auto put = [&](u32 p) { if (STAGED) sp[loc + ord] = p; else mypos[ord] = p; ++ord; };
            dc_item_rounds<SIDES>(it, valid, maxr,
                [&](int slot, bool on, u32 bit) {
                    const u64 m = __ballot(on);
                    const u32 bs = (u32)__builtin_amdgcn_readlane(sreg, slot);
                    if (on) {
                        const u32 p = bs + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));   // peers in lower lanes
                        events[p] = (u16)(sig | (bit << 11));
                        put(p);
                    }
                    sreg = dc_writelane(sreg, (int)(bs + (u32)__popcll(m)), slot);                    // (was: compare, move, select)
                },
                [&](int tau, bool on, u32 bit, u32 mlo, u32 mhi) {
                    const u32 h = on ? (u32)tau : 0u;
                    const u32 before = vg[h];
                    const u32 rr = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                    const u32 cn = (u32)(__popc(mlo) + __popc(mhi));
                    if (on) {
                        const u32 p = before + rr;
                        vg[h] = before + cn;                                  // every peer writes the same value (was: the highest one, behind a compare and a branch)
                        events[p] = (u16)(sig | (bit << 11));
                        put(p);
                    }
                });
        };
        if (staged) rounds(std::true_type()); else rounds(std::false_type());
        // the tile's positions leave as 16-byte stores (four entries per lane: two store instructions for an average tile instead of seven)
        if (staged) {
            for (u32 t = 4u * lane; t < tile_total; t += 256u) {
                if (t + 4u <= tile_total) {
                    DcPos4 v; v.a = sp[t]; v.b = sp[t + 1]; v.c = sp[t + 2]; v.d = sp[t + 3];
                    *reinterpret_cast<DcPos4*>(pos + running + t) = v;
                } else for (u32 x = t; x < tile_total; ++x) pos[running + x] = sp[x];
            }
        }
        running += tile_total;
    }
    if (wc == g.W - 1 && lane == 0) doff[g.m] = running;
}

// 2d. the fast coder (-e0) has ONE family, so the stream-order job has no chains to lay out — but the p stream still needs to know
// where every run's entries go: doff[i] = decision index of item i's first decision, exactly what dc_part_scatter leaves behind.
template <int SIDES>
__global__ __launch_bounds__(WG) void dc_doff_kernel(const u64* __restrict__ items, DcGeom g, DcSub S, const u32* __restrict__ meta,
                                                     const u32* __restrict__ wdecoff, u32* __restrict__ doff)
{
    const u32 mrp = dc_maxr_pack(S);                                  // max_rank of the eight sub-blocks, one scalar word
    if (meta[DM_FAIL] != 0u) return;
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 wc = blockIdx.x * WAVES + w;
    if (wc >= g.W) return;
    const u64 i0 = (u64)wc * g.per_wave;
    u64 i1 = i0 + g.per_wave; if (i1 > g.m) i1 = g.m;
    u32 running = wdecoff[wc];
    u64 knext = (i0 + lane < i1) ? items[i0 + lane] : 0ull;          // the next tile's items are requested one tile ahead
    for (u64 base = i0; base < i1; base += 64) {
        const u64 i = base + lane;
        const bool valid = i < i1;
        const u64 key = knext;
        asm volatile("" :: "v"(key) : "memory");                     // (wait for this tile's items, THEN request the next tile's: dc_part_scatter_kernel)
        const Item it = item_unpack(key);
        knext = (i + 64 < i1) ? items[i + 64] : 0ull;
        const int maxr = dc_maxr_of(it.sb, mrp);
        u32 nd = 0;
        if (valid) { if (SIDES & 1) nd += (u32)count_rank_side(it, maxr); if (SIDES & 2) nd += (u32)count_run_side(it); }
        const u32 incl = wave_incl_sum(nd);
        if (valid) doff[i] = running + incl - nd;
        running += (u32)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (wc == g.W - 1 && lane == 0) doff[g.m] = running;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. chain evaluation over chain-major events
// ---------------------------------------------------------------------------------------------------------------------
struct DcEvalJob { const u16* events; u32 E; const u32* rowstart; int fam; };      // rowstart[DC_ROWS + 1]: row = decision type

template <class P>
__device__ __forceinline__ u32 dc_find_row(P rowstart, u32 k)
{
    u32 lo = 0, hi = DC_ROWS;                                          // largest row with rowstart[row] <= k
    while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (rowstart[mid] <= k) lo = mid; else hi = mid; }
    return lo;
}

// Phases a and c as one wavefront per 64 chunks: the lanes own one chunk each (a serial chain), but memory is touched as
// full lines — for every batch of 64 events per lane the wavefront loads the 64 lanes' 128-byte pieces cooperatively
// (8 x 16 B per lane, eight rows per instruction), transposes them through LDS, and (phase c) writes the values back the same
// way.  The next batch is in flight while the current one is walked.
#ifndef DC_EVAL_TIMING
#define DC_EVAL_TIMING 0
#endif
constexpr int DC_EB = 64;                       // events per lane per batch
constexpr int DC_EROW = DC_EB * 2 + 16;         // LDS row pitch in bytes (padded)
// all four jobs of a block in one launch (they are independent, and one job alone leaves most SIMDs idle: a lane is a serial chain)
struct DcEvalAll { DcEvalJob job[4]; u32 wstart[5]; u32 cstart[5]; u16* V[4]; u16* sink; u32 ev; };   // first wavefront / first chunk of each job; events per chunk

// First event of every non-empty row gets DC_ROWMARK: inside the rows of one class (same update rates) a walk then needs no row
// bookkeeping at all — a chain ends where the signature changes or a marked event begins.
constexpr u32 DC_ROWMARK = 0x1000u;
__global__ __launch_bounds__(WG) void dc_mark_rows_kernel(DcEvalAll A)
{
    const u32 g = blockIdx.x * WG + threadIdx.x;
    if (g >= 4u * DC_ROWS) return;
    const DcEvalJob J = A.job[g / DC_ROWS];
    if (J.E == 0) return;                                              // a job that does not run in this mode (its row table is not valid)
    const u32 r = g % DC_ROWS;
    const u32 rs = J.rowstart[r], re = J.rowstart[r + 1];
    if (re > rs) const_cast<u16*>(J.events)[rs] |= (u16)DC_ROWMARK;
}
// first row of each class, and one past the last
__device__ __forceinline__ int dc_class_first_row(int cls)
{
    return cls == CLS_RF ? TAU_RF : cls == CLS_RE ? TAU_RE : cls == CLS_RM ? TAU_RM : cls == CLS_RP ? TAU_RP : cls == CLS_NF ? TAU_NF
         : cls == CLS_NE ? TAU_NE : cls == CLS_NM ? TAU_NM : cls == CLS_NM2 ? TAU_NM2 : DC_ROWS;
}

// A wavefront is self-contained here (its own LDS slices, no workgroup barrier); four of them form a workgroup only so that the
// hardware puts them on the four SIMDs of one CU — launched as single-wave workgroups, a tenth of the ~1000 wavefronts ended up
// two to a SIMD while other SIMDs stayed empty, and the kernel took 1.6 x as long as its median wavefront.
constexpr int DC_EVAL_WAVES = 4;
constexpr int DC_EVAL_LDS_WAVE = 2 * 64 * DC_EROW;                                             // sin + sout
constexpr int DC_EVAL_LDS = DC_EVAL_WAVES * (DC_EVAL_LDS_WAVE + NUM_CLS * 4 + NUM_CLS * (int)sizeof(Rates));   // + class ends, rates
__device__ __forceinline__ void dc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // LDS operations of one wavefront execute in order: only the compiler has to be told
    __builtin_amdgcn_wave_barrier();
}

template <bool WRITE>
__global__ __launch_bounds__(64 * DC_EVAL_WAVES) void dc_eval_wave_kernel(DcEvalAll A, const ModelParams* __restrict__ mp, const u32* __restrict__ meta,
                                                          u16* __restrict__ elo_all, u16* __restrict__ ehi_all, const u16* __restrict__ Sv_all, u32* __restrict__ tdbg)
{
#if DC_EVAL_TIMING
    const u64 t_begin = wall_clock64();
    u32 slow_batches = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) u8 dc_eval_lds[];
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    u8* const sin = dc_eval_lds + wv * DC_EVAL_LDS_WAVE;
    u8* const sout = sin + 64 * DC_EROW;
    // where each class's rows end (in events) and the family's rates per class: all a walk needs besides the events themselves
    u32* const sclsend = reinterpret_cast<u32*>(dc_eval_lds + DC_EVAL_WAVES * DC_EVAL_LDS_WAVE) + wv * NUM_CLS;
    Rates* const srate = reinterpret_cast<Rates*>(dc_eval_lds + DC_EVAL_WAVES * (DC_EVAL_LDS_WAVE + NUM_CLS * 4)) + wv * NUM_CLS;
    const u32 gw = blockIdx.x * DC_EVAL_WAVES + wv;                    // this wavefront among all of the launch
    if (meta[DM_FAIL] != 0u || gw >= A.wstart[4]) return;
    int jb = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) if (gw >= A.wstart[q]) jb = q;
    const DcEvalJob J = A.job[jb];
    if (lane < (u32)NUM_CLS) {
        srate[lane] = mp->rates[lane][J.fam];
        sclsend[lane] = J.rowstart[dc_class_first_row((int)lane + 1)];
    }
    dc_wave_sync();
    u16* const elo = elo_all + A.cstart[jb];
    u16* const ehi = ehi_all + A.cstart[jb];
    const u16* const Sv = Sv_all + A.cstart[jb];
    u16* const Vout = A.V[jb];
    const u32 wave = gw - A.wstart[jb];
    const u32 c = wave * 64 + lane;
    const u32 EV = A.ev;                                               // multiple of DC_EB
    const u64 k0 = (u64)c * EV;
    const bool mine = k0 < J.E;
    const u32 k1 = mine ? (u32)((k0 + EV < J.E) ? k0 + EV : J.E) : 0u;
    // cooperative mapping: instruction i moves 16 bytes of row 8 i + lane / 8, column lane % 8
    const u32 crow = lane >> 3, ccol = lane & 7u;
    const u64 wave_k0 = (u64)wave * 64 * EV;

    u32 cls = 0, clsend = 0, prev = 0xffffu;
    int init = mp->init[0];                                            // the value a chain of the current class starts from
    int lo = init, hi = init;
    Rates R = srate[0];
    if (mine) {
        while (cls + 1 < (u32)NUM_CLS && sclsend[cls] <= (u32)k0) ++cls;                     // class of the row that owns event k0
        clsend = sclsend[cls];
        R = srate[cls];
        init = mp->init[cls];
        // a chunk that starts a row starts with a marked event; otherwise the chain may continue from the event before
        prev = (k0 > 0) ? ((u32)J.events[k0 - 1] & DC_SIGMASK) : 0xffffu;
        if (WRITE) { lo = Sv[c]; hi = lo; } else { lo = mp->vmin[cls][J.fam]; hi = mp->vmax[cls][J.fam]; }
    }
    auto fetch = [&](u32 b, uint4* q) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // no branch around a global load or store in this kernel (past the end: a clamped address / a sink slot): the compiler
            // then knows how many are outstanding and waits for the prefetched batch with an exact vmcnt, not for the stores behind it
            const u64 kk = wave_k0 + (u64)(8 * i + crow) * EV + (u64)b * DC_EB + ccol * 8;
            const u64 ka = kk < J.E ? kk : 0ull;
            const uint4 val = *reinterpret_cast<const uint4*>(J.events + ka);
            q[i].x = val.x; q[i].y = val.y; q[i].z = val.z; q[i].w = val.w;          // (member-wise: a whole-vector copy through the select made the array live in scratch)
        }
    };
    auto store_out = [&](u32 b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u64 kk = wave_k0 + (u64)(8 * i + crow) * EV + (u64)b * DC_EB + ccol * 8;
            *reinterpret_cast<uint4*>(kk < J.E ? Vout + kk : A.sink + lane * 8) = *reinterpret_cast<const uint4*>(sout + (8 * i + crow) * DC_EROW + ccol * 16);
        }
    };
    uint4 nxt[8];
    fetch(0, nxt);
    const u32 NB = EV / DC_EB;
    for (u32 b = 0; b < NB; ++b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sin + (8 * i + crow) * DC_EROW + ccol * 16) = nxt[i];
        dc_wave_sync();
        // the values of the batch before leave now, a whole walk ahead of the next wait on the memory counter (the loads below are
        // waited for at the top of the next iteration, and the counter is in order: stores issued after them would be waited for too)
        if (WRITE && b > 0) store_out(b - 1);
        fetch(b + 1 < NB ? b + 1 : b, nxt);                            // (the last batch is simply fetched again)
        const u32 kb = (u32)k0 + b * DC_EB;
        if (mine && kb < k1) {
            const u8* myrow = sin + lane * DC_EROW;
            u8* orow = sout + lane * DC_EROW;
            const u32 lim = k1 < clsend ? k1 : clsend;
            if (kb + DC_EB <= lim) {
                // whole batch inside one class (one set of rates): straight-line walk from registers
                const int c0 = R.t0 * R.a0 + R.r0, c1 = R.t1 * R.a1 + R.r1;
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) {
                    const uint4 q = *reinterpret_cast<const uint4*>(myrow + g8 * 16);
                    const u32 wds[4] = {q.x, q.y, q.z, q.w};
                    u32 outw[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const u32 e = (wds[x >> 1] >> (16 * (x & 1))) & 0xffffu;
                        const u32 sigm = e & (DC_SIGMASK | DC_ROWMARK);                      // a marked event never equals prev
                        if (sigm != prev) { lo = init; hi = init; }
                        prev = e & DC_SIGMASK;
                        // dcm::step with the target, rate and rounding picked by the bit first, so that both ends of the bracket
                        // share the selects (static coder: bit 1 is v - (((v - t1) a1) >> 12) = v + (((t1 - v) a1 + 4095) >> 12) —
                        // floor of a negated quotient = minus its ceiling —, bit 0 is v + (((t0 - v) a0) >> 12); devcoder_model.h)
                        // (round 5: (t - v) a + r = (t a + r) - v a with c = t a + r fixed per class and bit — two selects and three operations
                        // per step instead of three and four; all products stay below 2^24: |v| < 2^13, a < 2^11, c < 2^24)
                        const bool b1 = (e & 0x800u) != 0u;
                        const int Cc = b1 ? c1 : c0, Aa = b1 ? R.a1 : R.a0;
                        if (WRITE) outw[x >> 1] |= (u32)lo << (16 * (x & 1));
                        lo += (Cc - __mul24(lo, Aa)) >> 12;
                        if (!WRITE) hi += (Cc - __mul24(hi, Aa)) >> 12;
                    }
                    if (WRITE) *reinterpret_cast<uint4*>(orow + g8 * 16) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
                }
            } else {
                // a class boundary or the end of the chunk inside the batch (a handful of batches per launch)
#if DC_EVAL_TIMING
                ++slow_batches;
#endif
                const u32 cnt = (k1 - kb < (u32)DC_EB) ? k1 - kb : (u32)DC_EB;
                for (u32 x = 0; x < cnt; ++x) {
                    const u32 k = kb + x;
                    while (k >= clsend && cls + 1 < (u32)NUM_CLS) { ++cls; clsend = sclsend[cls]; R = srate[cls]; init = mp->init[cls]; }
                    const u32 e = *reinterpret_cast<const u16*>(myrow + 2 * x);
                    if ((e & (DC_SIGMASK | DC_ROWMARK)) != prev) { lo = init; hi = init; }
                    prev = e & DC_SIGMASK;
                    const u32 bt = (e >> 11) & 1u;
                    if (WRITE) *reinterpret_cast<u16*>(orow + 2 * x) = (u16)lo;
                    lo = step(lo, bt, R);
                    if (!WRITE) hi = step(hi, bt, R);
                }
            }
        }
        dc_wave_sync();
    }
    if (WRITE) store_out(NB - 1);
    if (!WRITE && mine) { elo[c] = (u16)lo; ehi[c] = (u16)hi; }
#if DC_EVAL_TIMING
    {
        u32 sb = slow_batches;
        for (int d = 32; d >= 1; d >>= 1) { const u32 o = (u32)__shfl_xor((int)sb, d, 64); sb = sb > o ? sb : o; }
        if (lane == 0) { tdbg[3 * gw] = (u32)(t_begin & 0xffffffffu); tdbg[3 * gw + 1] = (u32)(wall_clock64() - t_begin); tdbg[3 * gw + 2] = sb | ((u32)jb << 16); }
    }
#endif
}

__device__ __forceinline__ bool dc_chunk_continues(const DcEvalJob& J, u32 c, u32 EV)
{
    if (c == 0) return false;
    const u32 k0 = c * EV;
    u32 row = dc_find_row(J.rowstart, k0);
    if (J.rowstart[row] == k0) return false;                          // a row (hence a chain) starts exactly here
    // (an empty row cannot own k0: dc_find_row returns the last row starting at or before k0, which then is non-empty or k0 >= E)
    return (((u32)J.events[k0 - 1] ^ (u32)J.events[k0]) & DC_SIGMASK) == 0;
}

// exact value at the start of every chunk that begins inside a chain (all jobs in one launch; thread per chunk)
__global__ __launch_bounds__(WG) void dc_eval_b_kernel(DcEvalAll A, const ModelParams* __restrict__ mp, u32* __restrict__ meta,
                                                       const u16* __restrict__ elo_all, const u16* __restrict__ ehi_all, u16* __restrict__ Sv_all)
{
    if (meta[DM_FAIL] != 0u) return;
    const u32 g = blockIdx.x * WG + threadIdx.x;
    if (g >= A.cstart[4]) return;
    int jb = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) if (g >= A.cstart[q]) jb = q;
    const DcEvalJob J = A.job[jb];
    const u16* elo = elo_all + A.cstart[jb]; const u16* ehi = ehi_all + A.cstart[jb]; u16* Sv = Sv_all + A.cstart[jb];
    const u32 c = g - A.cstart[jb];
    const u32 EV = A.ev;
    const u64 k0 = (u64)c * EV;
    if (k0 >= J.E) return;
    const int init_c = mp->init[tau_class((int)dc_find_row(J.rowstart, (u32)k0))];     // what a chain of this chunk's first row starts from
    if (!dc_chunk_continues(J, c, EV)) { Sv[c] = (u16)init_c; return; }
    if (elo[c - 1] == ehi[c - 1]) { Sv[c] = elo[c - 1]; return; }
    // the predecessor did not coalesce: replay from the nearest chunk whose start value is known
    // (the chunks walked here hold no chain boundary — see below —, so they are all of this chunk's row and class)
    u32 j = c - 1;
    int start = init_c;
    u32 depth = 1;
    for (;;) {
        if (!dc_chunk_continues(J, j, EV)) { start = init_c; break; }
        if (elo[j - 1] == ehi[j - 1]) { start = elo[j - 1]; break; }
        --j; ++depth;
        if (depth > 64) { atomicOr(&meta[DM_FAIL], (u32)FAIL_REPLAY); Sv[c] = (u16)init_c; return; }
    }
    atomicAdd(&meta[DM_REPLAYS], depth);
    // Exact walk over chunks j .. c - 1.  None of them coalesced, so none contains a chain boundary (a boundary resets both ends of the
    // bracket to 2048 and they stay equal from there on): one chain, one row, one set of rates — a bare loop, eight events per 16-byte
    // load with four loads in flight (EV is a multiple of 8 and chunks start 16-byte aligned).  The GPU waits for this one lane:
    // 0.37 ms per replayed chunk with the general walk (row look-ups and signature tests per event), 0.32 ms like this (~80 cycles per
    // event: a lone wavefront issues the five dependent operations of a step ~8 cycles apart).  With several contexts on the GPU the
    // other blocks' kernels fill the machine meanwhile.
    const Rates R = mp->rates[tau_class((int)dc_find_row(J.rowstart, j * EV))][J.fam];
    int v = start;
    const uint4* ev = reinterpret_cast<const uint4*>(J.events + (size_t)j * EV);
    const u32 nq = (c - j) * (EV / 8u);
    uint4 qb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qb[i] = ev[(u32)i < nq ? (u32)i : 0u];
    for (u32 i = 0; i < nq; ++i) {
        const uint4 q = qb[0];
        qb[0] = qb[1]; qb[1] = qb[2]; qb[2] = qb[3];
        qb[3] = ev[i + 4u < nq ? i + 4u : i];
        const u32 wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int x = 0; x < 8; ++x) v = step(v, (wds[x >> 1] >> (16 * (x & 1) + 11)) & 1u, R);
    }
    Sv[c] = (u16)v;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3s. the static (context-free) family's rank side in STREAM ORDER (round 6; devcoder_static.h has the method and every lane
// function — the kernels below only map lanes to work).  Replaces, for blocks of at most 32 symbols per sub-block, the
// partition / evaluation / gather of that family: qlfc.cpp:829-1129's third counter of every rank decision.
// ---------------------------------------------------------------------------------------------------------------------
struct SpArgs {
    dcs::SpGeom g; DcSub S; u32 slots; u32 uniform;                     // slots: bit s = slot s occurs for some sub-block's max_rank; uniform: one max_rank for all sub-blocks
    const u64* planes; const dcs::SpDesc* desc; const ModelParams* mp;
    dcs::SpSum* sums; dcs::SpGroupSum* gsum; u16* gv; u16* sv; u16* state; uint4* rec; u32* meta;
};
// phase A (brackets) and phase C (exact walk): block = 256 consecutive chunks of one slot (blockIdx.y), heavy slots first
template <bool EXACT>
__global__ __launch_bounds__(WG) void sp_walk_kernel(SpArgs A)
{
    if (A.meta[DM_FAIL] != 0u) return;
    const int slot = (int)blockIdx.y;
    if (!((A.slots >> slot) & 1u)) return;
    const u32 chunk = blockIdx.x * WG + threadIdx.x;
    if (chunk >= dcs::sp_nchunks(A.g, slot)) return;
    // every sub-block with the same max_rank (one alphabet for the whole block: the common case): the slot's descriptor is one scalar value
    const dcs::SpDesc dval = A.desc[A.S.maxr[0] * dcs::SP_SLOTS + (u32)slot];
    const dcs::SpDesc* du = A.uniform ? &dval : nullptr;
    const dcs::SpParams P = dcs::sp_params(*A.mp, slot);
    if (EXACT) dcs::sp_phase_c(slot, chunk, A.g, A.S, A.planes, A.desc, P, A.sv, A.state, du);
    else {
        const dcs::SpSum s = dcs::sp_phase_a(slot, chunk, A.g, A.S, A.planes, A.desc, P, du);
        A.sums[(size_t)slot * A.g.cstride + chunk] = s;
    }
}
// Resolve steps 1 and 3: one WAVEFRONT per (slot, group).  The walk through a group's 64 chunk summaries is serial, but fetching them
// is not: the lanes load one summary each, and all of them then run the same uniform code, reading summary i out of lane i's registers
// (v_readlane; a lane per group waited out a memory latency per chunk: 0.9 ms for the step, nearly all of it in the lanes of rare types).
__device__ __forceinline__ u32 dc_readlane(u32 x, u32 i) { return (u32)__builtin_amdgcn_readlane((int)x, (int)i); }
template <int STEP>
__global__ __launch_bounds__(WG) void sp_resolve_kernel(SpArgs A)
{
    if (A.meta[DM_FAIL] != 0u) return;
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const int slot = (int)blockIdx.y;
    if (!((A.slots >> slot) & 1u)) return;
    const u32 grp = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));
    if (grp >= dcs::sp_ngroups(A.g, slot)) return;
    const u32 nch = dcs::sp_nchunks(A.g, slot);
    const u32 c = grp * dcs::SP_RG + lane;
    const uint4 mine4 = (c < nch) ? reinterpret_cast<const uint4*>(A.sums + (size_t)slot * A.g.cstride)[c] : make_uint4(0, 0, 0, 0);
    auto get = [&](int i) {
        const u32 ii = (u32)__builtin_amdgcn_readfirstlane(i);
        const u32 x = dc_readlane(mine4.x, ii), y = dc_readlane(mine4.y, ii), z = dc_readlane(mine4.z, ii), ww = dc_readlane(mine4.w, ii);
        dcs::SpSum s; s.lo = (u16)(x & 0xffffu); s.hi = (u16)(x >> 16); s.cnt = (u16)(y & 0xffffu); s.flags = (u16)(y >> 16); s.hist = (u64)z | ((u64)ww << 32);
        return s;
    };
    const dcs::SpParams P = dcs::sp_params(*A.mp, slot);
    if (STEP == 1) {
        dcs::SpGroupSum o;
        const bool ok = dcs::sp_resolve_group_g(slot, grp, A.g, A.S, A.planes, A.desc, P, get, &o);
        if (lane == 0) { if (ok) A.gsum[(size_t)slot * A.g.gstride + grp] = o; else atomicOr(&A.meta[DM_FAIL], (u32)FAIL_REPLAY); }
    } else {
        int mine = 0;
        const bool ok = dcs::sp_resolve_chunks_g(slot, grp, (int)A.gv[(size_t)slot * A.g.gstride + grp], A.g, A.S, A.planes, A.desc, P, get,
                                                 [&](int i, int val) { if ((u32)i == lane) mine = val; });
        if (!ok) { if (lane == 0) atomicOr(&A.meta[DM_FAIL], (u32)FAIL_REPLAY); }
        else if (c < nch) A.sv[(size_t)slot * A.g.cstride + c] = (u16)mine;
    }
}
// Resolve step 2: one wavefront per slot walks the slot's group summaries in order, 64 at a time out of the lanes' registers
__global__ __launch_bounds__(64) void sp_resolve_serial_kernel(SpArgs A)
{
    if (A.meta[DM_FAIL] != 0u) return;
    const int slot = (int)blockIdx.x;
    if (!((A.slots >> slot) & 1u)) return;
    const u32 lane = threadIdx.x;
    const u32 ng = dcs::sp_ngroups(A.g, slot);
    const dcs::SpParams P = dcs::sp_params(*A.mp, slot);
    int v = P.init;
    for (u32 base = 0; base < ng; base += 64) {
        const u32 cnt = (ng - base < 64u) ? ng - base : 64u;
        const uint4 g4 = (lane < cnt) ? reinterpret_cast<const uint4*>(A.gsum + (size_t)slot * A.g.gstride)[base + lane] : make_uint4(0, 0, 0, 0);
        int mine = 0;
        for (u32 i = 0; i < cnt; ++i) {
            if (i == lane) mine = v;
            const u32 x = dc_readlane(g4.x, i), y = dc_readlane(g4.y, i), z = dc_readlane(g4.z, i), ww = dc_readlane(g4.w, i);
            dcs::SpGroupSum o; o.closed = (u16)(x & 0xffffu); o.value = (u16)(x >> 16); o.cnt = (u16)(y & 0xffffu); o.big = (u16)(y >> 16); o.hist = (u64)z | ((u64)ww << 32);
            if (!dcs::sp_after_group(&v, o, slot, base + i, A.g, A.S, A.planes, A.desc, P, A.sums + (size_t)slot * A.g.cstride + (size_t)(base + i) * dcs::SP_RG)) {
                if (lane == 0) atomicOr(&A.meta[DM_FAIL], (u32)FAIL_REPLAY);
                return;
            }
        }
        if (lane < cnt) A.gv[(size_t)slot * A.g.gstride + base + lane] = (u16)mine;
    }
}
// values: one wavefront per tile, lane = (slot, sub-tile); the tile's 64 records (8 x u16 per run) leave as one 1 KB piece
__global__ __launch_bounds__(WG) void sp_values_kernel(SpArgs A)
{
    __shared__ __attribute__((aligned(16))) u16 rec[WAVES][64][8];
    __shared__ dcs::SpParams P3[3];                                     // by class: RF, RE, RM
    if (A.meta[DM_FAIL] != 0u) return;
    if (threadIdx.x < 3u) P3[threadIdx.x] = dcs::sp_params(*A.mp, threadIdx.x == 0 ? 0 : threadIdx.x == 1 ? 1 : 5);
    __syncthreads();
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 tile = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + w));
    if (tile >= A.g.ntiles) return;
    typedef __attribute__((address_space(3))) volatile u16 lds_vu16;
    lds_vu16* mine = (lds_vu16*)&rec[w][0][0];
    dcs::sp_values(tile, (int)lane, A.g, A.S, A.planes, A.desc, P3, A.state, [&](int i, int k, int v) { mine[i * 8 + k] = (u16)v; });
    dc_wave_sync();
    const u32 j = tile * 64u + lane;
    if (j < A.g.m) A.rec[j] = *reinterpret_cast<const uint4*>(&rec[w][lane][0]);
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. probability stream, stream order.  Thread per run.
// ---------------------------------------------------------------------------------------------------------------------
struct DcGather {
    const u64* key_ch; u32 m;
    const u32 *inv_ch, *inv_sr, *inv_sn;
    const u32 *doff_sp, *doff_ch, *doff_sr, *doff_sn;
    const u32 *pos_sp, *pos_ch, *pos_sr, *pos_sn;
    const u16 *V_sp, *V_ch, *V_sr, *V_sn;
    // static family in stream order (SPF): the 8 x u16 record of every run's rank side, the runs' offsets in the p stream; doff_sp /
    // pos_sp / V_sp then belong to the small job that holds the family's NE / NM decisions only (entry kk - 1 of the run's piece)
    const uint4* sp_rec; const u32* doff_full;
};
// pos entries of one run are contiguous but only 4-byte aligned: a packed struct makes the compiler use one dwordx4 load
// (global memory takes dword-aligned multi-dword accesses) instead of four dword loads
struct __attribute__((packed, aligned(4))) DcU4 { u32 a, b, c, d; };

// The entries of a run are 2 bytes each and 1..~20 per run: written straight from the lanes, every store instruction of a wavefront
// touches ~13 lines with 2 bytes per lane, and the same lines again in the next iteration (PMC: WRITE_SIZE 2.66 GB for 0.37 GB of
// entries, and the read-modify-write traffic behind it).  The entries of a wavefront's 64 runs are contiguous in the stream, so
// they are collected in LDS and leave as whole 128-byte pieces (DC_PS_STAGE entries per wavefront; a wavefront with more — 64 runs of
// > 28 decisions on average — stores directly).
#ifndef DC_PS_STAGE_N
#define DC_PS_STAGE_N 1792
#endif
constexpr u32 DC_PS_STAGE = DC_PS_STAGE_N;
typedef __attribute__((address_space(3))) volatile u16 dc_lds_vu16;

// ---- the packed stream (round 6): 13 bits per decision ---------------------------------------------------------------------------
// What the host's range coder needs of a static-coder entry is the probability (12 bits) and the coded bit; the run-start mark only
// placed the reference's budget test (qlfc.cpp:894), and a stream that reaches its budget is redone from the run arrays anyway.
// P13: field = entry & 0x1fff, eight decisions in 13 bytes (field e of a group at bits [13 e, 13 e + 13), little endian); a sub-block's
// stream starts at a multiple of 64 decisions (104 bytes) and is zero-padded to a whole group: 366 -> 298 MB per 64 MiB block over PCIe
// and through the host's DRAM.  The entries of a wavefront's runs are contiguous in the stream, but its piece starts and ends anywhere,
// so a wavefront writes the groups that lie wholly inside its piece (13 bytes per lane and trip) and leaves the fragment of a group
// at either end of it as a 24-byte record; dc_p13_join_kernel then writes the group between every two neighbouring wavefronts from
// the tail record of the one and the head record of the other — no atomics, no zeroed buffer.
struct DcFrag { u32 gidx; u32 slot_count; u16 e[8]; };               // group index (decisions / 8), first slot | count << 8 (count 0: none), the entries
struct DcP13 { u32 pad[8]; u8* out; DcFrag* frag; };                 // pad[sb]: packed index of a decision = its index in the block's stream + pad[its sub-block]
struct __attribute__((packed)) DcGroup13 { u32 a, b, c; u8 d; };
__device__ __forceinline__ DcGroup13 dc_pack13(const u32 (&f)[8])
{
    const u64 lo = (u64)f[0] | ((u64)f[1] << 13) | ((u64)f[2] << 26) | ((u64)f[3] << 39) | ((u64)f[4] << 52);
    const u64 hi = (u64)(f[4] >> 12) | ((u64)f[5] << 1) | ((u64)f[6] << 14) | ((u64)f[7] << 27);
    DcGroup13 g; g.a = (u32)lo; g.b = (u32)(lo >> 32); g.c = (u32)hi; g.d = (u8)(hi >> 32);
    return g;
}
// The wavefront's staged entries sg[0 .. wtotal) are the decisions wbase .. of the block's stream; loc / nd / sb / valid: this lane's run.
__device__ __forceinline__ void dc_flush13(dc_lds_vu16* sg, const u32 wbase, const u32 lane, const u32 loc, const u32 nd, const u32 sb, const bool valid,
                                           const DcP13& Q, const u32 wave_global)
{
    const u64 vmask = __ballot(valid);
    DcFrag head, tail; head.slot_count = 0; tail.slot_count = 0; head.gidx = 0; tail.gidx = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) { head.e[x] = 0; tail.e[x] = 0; }
    if (vmask != 0ull) {
        const u32 lastv = 63u - (u32)__builtin_clzll(vmask);
        const u32 sb_first = (u32)__builtin_amdgcn_readfirstlane((int)sb), sb_last = (u32)__builtin_amdgcn_readlane((int)sb, (int)lastv);
        for (u32 s2 = sb_first; s2 <= sb_last; ++s2) {                    // one trip, except where sub-blocks meet inside the wavefront
            const u64 mk = __ballot(valid && sb == s2);
            if (mk == 0ull) continue;
            const u32 l0 = (u32)__builtin_ctzll(mk), l1 = 63u - (u32)__builtin_clzll(mk);
            const u32 k0 = (u32)__builtin_amdgcn_readlane((int)loc, (int)l0), k1 = (u32)__builtin_amdgcn_readlane((int)(loc + nd), (int)l1);
            const u32 len = k1 - k0;
            if (len == 0u) continue;
            const u32 P0 = wbase + k0 + Q.pad[s2];
            const u32 g_full0 = (P0 + 7u) >> 3, g_full1 = (P0 + len) >> 3;         // groups [g_full0, g_full1) lie wholly inside
            for (u32 g = g_full0 + lane; g < g_full1; g += 64u) {
                const u32 k = k0 + (g * 8u - P0);
                u32 f[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) f[x] = (u32)sg[k + (u32)x] & 0x1fffu;
                *reinterpret_cast<DcGroup13*>(Q.out + (size_t)g * 13u) = dc_pack13(f);
            }
            // the fragment before the first group boundary: only the wavefront's first piece can have one (a sub-block starts on a group)
            u32 hcount = 0;
            if ((P0 & 7u) != 0u) {
                hcount = 8u - (P0 & 7u); if (hcount > len) hcount = len;
                head.gidx = P0 >> 3; head.slot_count = (P0 & 7u) | (hcount << 8);
#pragma unroll
                for (int x = 0; x < 8; ++x) head.e[x] = (u32)x < hcount ? (u16)(sg[k0 + (u32)x] & 0x1fffu) : (u16)0;
            }
            // the fragment behind the last one: the next wavefront completes it — unless the sub-block ends here, then it is a whole (padded) group
            if (((P0 + len) & 7u) != 0u && (g_full1 << 3) >= P0) {
                const u32 tcount = (P0 + len) & 7u, tk = k0 + ((g_full1 << 3) - P0);
                u32 f[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) f[x] = (u32)x < tcount ? ((u32)sg[tk + (u32)x] & 0x1fffu) : 0u;
                if (s2 != sb_last) { if (lane == 0u) *reinterpret_cast<DcGroup13*>(Q.out + (size_t)g_full1 * 13u) = dc_pack13(f); }
                else {
                    tail.gidx = g_full1; tail.slot_count = tcount << 8;
#pragma unroll
                    for (int x = 0; x < 8; ++x) tail.e[x] = (u16)f[x];
                }
            }
        }
    }
    // both records (every lane holds the same two: they were read from LDS at wave-uniform addresses) as ONE 48-byte store: lane t writes word t
    if (lane < 12u) {
        const DcFrag& r = lane < 6u ? head : tail;
        const u32 q = lane < 6u ? lane : lane - 6u;
        const u32 w = q == 0u ? r.gidx : q == 1u ? r.slot_count
                    : q == 2u ? ((u32)r.e[0] | ((u32)r.e[1] << 16)) : q == 3u ? ((u32)r.e[2] | ((u32)r.e[3] << 16))
                    : q == 4u ? ((u32)r.e[4] | ((u32)r.e[5] << 16)) : ((u32)r.e[6] | ((u32)r.e[7] << 16));
        reinterpret_cast<u32*>(Q.frag + 2u * wave_global)[lane] = w;
    }
}
// thread w: the group between wavefront w and w + 1 (tail of w, head of w + 1), and a head nobody's tail belongs to (never on blocks whose
// wavefronts hold 64 runs; written for completeness)
__global__ __launch_bounds__(WG) void dc_p13_join_kernel(const DcFrag* __restrict__ frag, u32 nwaves, u8* __restrict__ out, const u32* __restrict__ meta)
{
    if (meta[DM_FAIL] != 0u) return;
    const u32 w = blockIdx.x * WG + threadIdx.x;
    if (w >= nwaves) return;
    const DcFrag T = frag[2u * w + 1u];
    DcFrag H; H.slot_count = 0; H.gidx = 0;
    if (w + 1u < nwaves) H = frag[2u * (w + 1u)];
    const u32 tc = T.slot_count >> 8, hc = H.slot_count >> 8, hs = H.slot_count & 0xffu;
    if (tc != 0u) {
        u32 f[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) f[x] = (u32)x < tc ? (u32)T.e[x] : 0u;
        if (hc != 0u && H.gidx == T.gidx) {
#pragma unroll
            for (int x = 0; x < 8; ++x) if ((u32)x >= hs && (u32)x < hs + hc) f[x] = (u32)H.e[(u32)x - hs];
        }
        *reinterpret_cast<DcGroup13*>(out + (size_t)T.gidx * 13u) = dc_pack13(f);
    }
    if (hc != 0u && !(tc != 0u && H.gidx == T.gidx)) {
        u32 f[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) f[x] = ((u32)x >= hs && (u32)x < hs + hc) ? (u32)H.e[(u32)x - hs] : 0u;
        *reinterpret_cast<DcGroup13*>(out + (size_t)H.gidx * 13u) = dc_pack13(f);
    }
}

// FAST: the fast coder's stream — one counter per decision (the char family's value IS the probability, 13 / 11 bits), entries
// {value, bit << 13, run start << 14, run side << 15} (devcoder_model.h PSF_*).
#ifndef DC_PS_MINW
#define DC_PS_MINW 1            // minimum waves per SIMD the register allocator must leave room for (A/B: 82 VGPRs = 5 waves by default)
#endif
template <bool FAST, bool SPF, bool P13>
__global__ __launch_bounds__(WG, DC_PS_MINW) void dc_pstream_kernel(DcGather G, DcSub S, const ModelParams* __restrict__ mp,
                                                        u32* __restrict__ meta, u16* __restrict__ out, u16* __restrict__ dbg /*[3][D] or null*/, u32 dbgD, DcP13 Q)
{
    const u32 mrp = dc_maxr_pack(S);                                  // max_rank of the eight sub-blocks, one scalar word
    __shared__ u16 stage[WAVES][DC_PS_STAGE];
    __shared__ short s_lr[NUM_CLS][4];                                // blend weights per class: an LDS read per decision instead of global loads at a per-lane address
    if (meta[DM_FAIL] != 0u) return;
    if (!FAST) {
        if (threadIdx.x < (u32)NUM_CLS * 3u) s_lr[threadIdx.x / 3u][threadIdx.x % 3u] = mp->lr[threadIdx.x / 3u][threadIdx.x % 3u];
        __syncthreads();
    }
    const u32 j = dc_virtual_block() * WG + threadIdx.x;
    const u32 lane = threadIdx.x & 63u;
    const bool valid = j < G.m;                                       // (whole wavefronts past the end still take part in the shuffles below)
    const Item it = item_unpack(valid ? G.key_ch[j] : 0ull);
    const int maxr = dc_maxr_of(it.sb, mrp);
    const int n_rank = valid ? count_rank_side(it, maxr) : 0, n_run = valid ? count_run_side(it) : 0, nd = n_rank + n_run;
    const u32 b_sp = valid ? (SPF ? G.doff_full[j] : G.doff_sp[j]) : 0u;
    const u32 incl = wave_incl_sum((u32)nd);
    const u32 loc = incl - (u32)nd;                                   // this run's first entry inside the wavefront's piece of the stream
    const u32 wtotal = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    const u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)b_sp);                  // lane 0 is valid whenever any lane is
    const bool staged = wtotal <= DC_PS_STAGE;                        // wave-uniform
    dc_lds_vu16* sg = (dc_lds_vu16*)&stage[threadIdx.x >> 6][0];
    const u32* p_sp = G.pos_sp + (SPF ? (valid ? G.doff_sp[j] : 0u) : b_sp);
    const u32* p_ch = G.pos_ch + (valid ? G.doff_ch[G.inv_ch[j]] : 0u);
    const u32* p_sr = FAST ? G.pos_ch : G.pos_sr + (valid ? G.doff_sr[G.inv_sr[j]] : 0u);
    const u32* p_sn = FAST ? G.pos_ch : G.pos_sn + (valid ? G.doff_sn[G.inv_sn[j]] : 0u);
    u16* o = out + b_sp;
    // SPF: the family's values of the run side's NE / NM decisions (entries kk - 1 = 0..3 of the small job's piece of this run: all there
    // are for runs below 32) are fetched up front — looked up inside the rounds they were two dependent loads in the middle of every
    // round, and the rounds' other gathers waited behind them
    u32 vrs[4] = {2048u, 2048u, 2048u, 2048u};
    if (SPF && !FAST && n_run > 1) {
        const DcU4 pp = *reinterpret_cast<const DcU4*>(p_sp);
        vrs[0] = G.V_sp[pp.a];
        if (n_run > 2) vrs[1] = G.V_sp[pp.b];
        if (n_run > 3) vrs[2] = G.V_sp[pp.c];
        if (n_run > 4) vrs[3] = G.V_sp[pp.d];
    }
    // SPF: q_sp is the VALUE for a rank-side decision (from the run's record)
    auto emit = [&](int k, u32 q_sp, u32 q_ch, u32 q_st, bool run_side, int cls, u32 bit) {
        const int v_ch = G.V_ch[q_ch];
        u16 e;
        if (FAST) {
            e = (u16)((u32)v_ch | (bit << 13) | (k == 0 ? (u32)PSF_RUN : 0u) | (run_side ? (u32)PSF_SIDE : 0u));
        } else {
            int v_sp;
            if (SPF) {
                // NF's counter of this family never moves (rate 0, blend weight 0: qlfc_data.inc RUN_FIRST): 2048, as the general path reports it
                const int kk = k - n_rank;                          // run side: 0 = NF, then NE / NM
                const u32 pre = kk == 1 ? vrs[0] : kk == 2 ? vrs[1] : kk == 3 ? vrs[2] : vrs[3];
                v_sp = !run_side ? (int)q_sp : (kk == 0 ? 2048 : kk <= 4 ? (int)pre : (int)G.V_sp[p_sp[kk - 1]]);
            } else v_sp = G.V_sp[q_sp];
            const int v_st = run_side ? G.V_sn[q_st] : G.V_sr[q_st];
            const int p = blend(v_ch, v_st, v_sp, s_lr[cls]);
            e = (u16)((u32)p | (bit << 12) | (k == 0 ? (u32)PS_RUN : 0u));
            if (dbg) { dbg[b_sp + k] = (u16)v_st; dbg[(size_t)dbgD + b_sp + k] = (u16)v_ch; dbg[2 * (size_t)dbgD + b_sp + k] = (u16)v_sp; }
        }
        if (staged) sg[loc + (u32)k] = e; else if (!P13) o[k] = e;
    };
    // first 8 decisions: positions by wide loads (the arrays have slack behind their last entry), static register indices
    u32 qsp[8], qch[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { qsp[k] = 0; qch[k] = 0; }
    if (valid) {
        const DcU4 c0 = *reinterpret_cast<const DcU4*>(p_ch), c1 = *reinterpret_cast<const DcU4*>(p_ch + 4);
        qch[0] = c0.a; qch[1] = c0.b; qch[2] = c0.c; qch[3] = c0.d; qch[4] = c1.a; qch[5] = c1.b; qch[6] = c1.c; qch[7] = c1.d;
        if (!FAST && SPF) {
            const uint4 r = G.sp_rec[j];
            qsp[0] = r.x & 0xffffu; qsp[1] = r.x >> 16; qsp[2] = r.y & 0xffffu; qsp[3] = r.y >> 16;
            qsp[4] = r.z & 0xffffu; qsp[5] = r.z >> 16; qsp[6] = r.w & 0xffffu; qsp[7] = r.w >> 16;
        } else if (!FAST) {
            const DcU4 a0 = *reinterpret_cast<const DcU4*>(p_sp), a1 = *reinterpret_cast<const DcU4*>(p_sp + 4);
            qsp[0] = a0.a; qsp[1] = a0.b; qsp[2] = a0.c; qsp[3] = a0.d; qsp[4] = a1.a; qsp[5] = a1.b; qsp[6] = a1.c; qsp[7] = a1.d;
        }
    }
    // the state family's positions: the first eight of each side by two wide loads each (was: one 4-byte load per decision); which of the
    // sixteen a decision takes is a select chain on (side, index inside the side)
    u32 sr8[8], sn8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sr8[k] = 0; sn8[k] = 0; }
    if (!FAST && valid) {
        const DcU4 r0 = *reinterpret_cast<const DcU4*>(p_sr), r1 = *reinterpret_cast<const DcU4*>(p_sr + 4);
        const DcU4 n0 = *reinterpret_cast<const DcU4*>(p_sn), n1 = *reinterpret_cast<const DcU4*>(p_sn + 4);
        sr8[0] = r0.a; sr8[1] = r0.b; sr8[2] = r0.c; sr8[3] = r0.d; sr8[4] = r1.a; sr8[5] = r1.b; sr8[6] = r1.c; sr8[7] = r1.d;
        sn8[0] = n0.a; sn8[1] = n0.b; sn8[2] = n0.c; sn8[3] = n0.d; sn8[4] = n1.a; sn8[5] = n1.b; sn8[6] = n1.c; sn8[7] = n1.d;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < nd) {
            u32 bit; bool rs;
            const int cls = nth_class(it, maxr, n_rank, k, &bit, &rs);
            u32 q_st = 0;
            if (!FAST) {
                // rank side: index k (static); run side: k - n_rank in 0 .. k
                u32 qn = sn8[0];
#pragma unroll
                for (int j = 1; j <= k; ++j) qn = (k - n_rank == j) ? sn8[j] : qn;
                q_st = rs ? qn : sr8[k];
            }
            emit(k, qsp[k], qch[k], q_st, rs, cls, bit);
        }
    }
    for (int k = 8; k < nd; ++k) {
        u32 bit; bool rs;
        const int cls = nth_class(it, maxr, n_rank, k, &bit, &rs);
        emit(k, (FAST || SPF) ? 0u : p_sp[k], p_ch[k], FAST ? 0u : (rs ? p_sn[k - n_rank] : p_sr[k]), rs, cls, bit);     // (SPF: a rank side has at most 8 decisions, so this is a run-side one)
    }
    if (P13) {
        // (a piece that does not fit the staging buffer voids the packed stream: the host launches the 2-byte form for this block)
        if (!staged) { if (lane == 0u) atomicOr(&meta[DM_P13_OVER], 1u); }
        __builtin_amdgcn_wave_barrier();
        if (staged) dc_flush13(sg, wbase, lane, loc, (u32)nd, (u32)it.sb, valid, Q, j >> 6);
        else if (lane < 12u) reinterpret_cast<u32*>(Q.frag + 2u * (j >> 6))[lane] = 0u;
    } else if (staged) {
        // the wavefront's piece [wbase, wbase + wtotal) of the stream: 4-byte stores from the first even entry on, the odd ends singly
        __builtin_amdgcn_wave_barrier();
        u16* ow = out + wbase;
        const u32 head = wbase & 1u;                                  // 1: the piece starts on an odd entry
        if (head && lane == 0 && wtotal > 0) ow[0] = sg[0];
        const u32 body = (wtotal - (head < wtotal ? head : wtotal)) >> 1;     // whole 4-byte words after the head
        u32* ow32 = reinterpret_cast<u32*>(ow + head);
        for (u32 t = lane; t < body; t += 64) ow32[t] = (u32)sg[head + 2 * t] | ((u32)sg[head + 2 * t + 1] << 16);
        if (lane == 0 && wtotal > head && ((wtotal - head) & 1u)) ow[wtotal - 1] = sg[wtotal - 1];
    }
}

// The p stream with the static family in stream order (devcoder_static.h): rank side and run side as two sets of rounds with STATIC
// register indices.  What bounds this kernel is the number of dependent memory round trips a wavefront goes through (round 6: taking
// a third of the gathers away changed nothing; the tail loop's two round trips per decision beyond the eighth were ~10 of them), so
// everything a run of up to 8 + 9 decisions needs is requested in three waves of loads: per-run words, then all position pieces by
// 16-byte loads (the char family's positions once from the run's first decision and once from its first run-side decision, so that
// neither side needs a select chain), then all counter values.  Lanes without a decision in a round read entry 0 (no branch around a load).
__global__ __launch_bounds__(WG, DC_PS_MINW) void dc_pstream_spf_kernel(DcGather G, DcSub S, const ModelParams* __restrict__ mp,
                                                        const u32* __restrict__ meta, u16* __restrict__ out, u16* __restrict__ dbg /*[3][D] or null*/, u32 dbgD)
{
    const u32 mrp = dc_maxr_pack(S);                                  // max_rank of the eight sub-blocks, one scalar word
    __shared__ u16 stage[WAVES][DC_PS_STAGE];
    __shared__ short s_lr[NUM_CLS][4];
    if (meta[DM_FAIL] != 0u) return;
    if (threadIdx.x < (u32)NUM_CLS * 3u) s_lr[threadIdx.x / 3u][threadIdx.x % 3u] = mp->lr[threadIdx.x / 3u][threadIdx.x % 3u];
    __syncthreads();
    const u32 j = dc_virtual_block() * WG + threadIdx.x;
    const u32 lane = threadIdx.x & 63u;
    const bool valid = j < G.m;
    const u32 jj = valid ? j : 0u;
    // 1. per-run words (coalesced)
    const u64 key = G.key_ch[jj];
    const u32 b_sp = G.doff_full[jj], b_mini = G.doff_sp[jj];
    const u32 i_ch = G.inv_ch[jj], i_sr = G.inv_sr[jj], i_sn = G.inv_sn[jj];
    const uint4 rec = G.sp_rec[jj];
    const Item it = item_unpack(valid ? key : 0ull);
    const int maxr = dc_maxr_of(it.sb, mrp);
    const int n_rank = valid ? count_rank_side(it, maxr) : 0, n_run = valid ? count_run_side(it) : 0, nd = n_rank + n_run;
    const u32 incl = wave_incl_sum((u32)nd);
    const u32 loc = incl - (u32)nd;
    const u32 wtotal = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    const u32 wbase = (u32)__builtin_amdgcn_readfirstlane((int)b_sp);
    const bool staged = wtotal <= DC_PS_STAGE;
    dc_lds_vu16* sg = (dc_lds_vu16*)&stage[threadIdx.x >> 6][0];
    const u32* p_sp = G.pos_sp + b_mini;
    const u32* p_ch = G.pos_ch + G.doff_ch[i_ch];
    const u32* p_sr = G.pos_sr + G.doff_sr[i_sr];
    const u32* p_sn = G.pos_sn + G.doff_sn[i_sn];
    u16* o = out + b_sp;
    // 2. positions (the arrays have slack behind their last entry)
    const u32* p_chn = p_ch + n_rank;                                  // the char family's positions of the run side
    const DcU4 cr0 = *reinterpret_cast<const DcU4*>(p_ch), cr1 = *reinterpret_cast<const DcU4*>(p_ch + 4);
    const DcU4 sr0 = *reinterpret_cast<const DcU4*>(p_sr), sr1 = *reinterpret_cast<const DcU4*>(p_sr + 4);
    const DcU4 cn0 = *reinterpret_cast<const DcU4*>(p_chn), cn1 = *reinterpret_cast<const DcU4*>(p_chn + 4);
    const DcU4 sn0 = *reinterpret_cast<const DcU4*>(p_sn), sn1 = *reinterpret_cast<const DcU4*>(p_sn + 4);
    const DcU4 sp0 = *reinterpret_cast<const DcU4*>(p_sp), sp1 = *reinterpret_cast<const DcU4*>(p_sp + 4);
    const u32 cn8 = p_chn[8], sn8 = p_sn[8];
    const u32 qchR[8] = {cr0.a, cr0.b, cr0.c, cr0.d, cr1.a, cr1.b, cr1.c, cr1.d};
    const u32 qsr[8] = {sr0.a, sr0.b, sr0.c, sr0.d, sr1.a, sr1.b, sr1.c, sr1.d};
    const u32 qchN[9] = {cn0.a, cn0.b, cn0.c, cn0.d, cn1.a, cn1.b, cn1.c, cn1.d, cn8};
    const u32 qsn[9] = {sn0.a, sn0.b, sn0.c, sn0.d, sn1.a, sn1.b, sn1.c, sn1.d, sn8};
    const u32 qsp[8] = {sp0.a, sp0.b, sp0.c, sp0.d, sp1.a, sp1.b, sp1.c, sp1.d};       // the small job's entries kk - 1 = 0..7
    const u32 vrec[8] = {rec.x & 0xffffu, rec.x >> 16, rec.y & 0xffffu, rec.y >> 16, rec.z & 0xffffu, rec.z >> 16, rec.w & 0xffffu, rec.w >> 16};
    // 3. values: rank side (at most 8 decisions: RF, <= 3 RE, <= 4 RM with max_rank <= 4)
    const u32 rank = it.rank;
    const int B = (valid && rank != 1u) ? bsr(rank) : 0;
    const int e = B ? (B - 1) + (B < maxr ? 1 : 0) : 0;
    int vch[8], vst[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const bool on = k < n_rank; vch[k] = G.V_ch[on ? qchR[k] : 0u]; vst[k] = G.V_sr[on ? qsr[k] : 0u]; }
    // run side: NF, then NE / NM
    const u32 run = it.run;
    const int nb = (valid && run != 1u) ? bsr(run) : 0;
    int wch[5], wst[5], wsp[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const bool on = kk < n_run;
        wch[kk] = G.V_ch[on ? qchN[kk] : 0u]; wst[kk] = G.V_sn[on ? qsn[kk] : 0u];
        wsp[kk] = (kk == 0) ? 2048 : (int)G.V_sp[on ? qsp[kk - 1] : 0u];       // NF's counter of this family never moves (rate 0, weight 0)
    }
    auto put = [&](int k, int p, u32 bit, int v_st, int v_ch, int v_sp) {
        const u16 en = (u16)((u32)p | (bit << 12) | (k == 0 ? (u32)PS_RUN : 0u));
        if (dbg) { dbg[b_sp + k] = (u16)v_st; dbg[(size_t)dbgD + b_sp + k] = (u16)v_ch; dbg[2 * (size_t)dbgD + b_sp + k] = (u16)v_sp; }
        if (staged) sg[loc + (u32)k] = en; else o[k] = en;
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < n_rank) {
            // k = 0: RF; 1..e: RE (bit = more to come); then RM depth k - 1 - e
            const int cls = k == 0 ? CLS_RF : k <= e ? CLS_RE : CLS_RM;
            const u32 bit = k == 0 ? (rank != 1u ? 1u : 0u) : k <= e ? (k < B ? 1u : 0u) : (rank >> (B - 1 - (k - 1 - e))) & 1u;
            put(k, blend(vch[k], vst[k], (int)vrec[k], s_lr[cls]), bit, vst[k], vch[k], (int)vrec[k]);
        }
    }
    auto run_round = [&](int kk, int v_ch, int v_st, int v_sp) {
        const int cls = kk == 0 ? CLS_NF : kk <= nb ? CLS_NE : (nb <= 5 ? CLS_NM : CLS_NM2);
        const u32 bit = kk == 0 ? (run != 1u ? 1u : 0u) : kk <= nb ? (kk < nb ? 1u : 0u) : (run >> (nb - 1 - (kk - 1 - nb))) & 1u;
        put(n_rank + kk, blend(v_ch, v_st, v_sp, s_lr[cls]), bit, v_st, v_ch, v_sp);
    };
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) if (kk < n_run) run_round(kk, wch[kk], wst[kk], wsp[kk]);
    if (__ballot(n_run > 5)) {                                          // a run of 8 or more somewhere in the wavefront (a third of them on text)
        int xch[4], xst[4], xsp[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int kk = 5 + x; const bool on = kk < n_run;
            xch[x] = G.V_ch[on ? qchN[kk] : 0u]; xst[x] = G.V_sn[on ? qsn[kk] : 0u]; xsp[x] = (int)G.V_sp[on ? qsp[kk - 1] : 0u];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) if (5 + x < n_run) run_round(5 + x, xch[x], xst[x], xsp[x]);
        for (int kk = 9; kk < n_run; ++kk)                              // runs of 32 and more: from memory, one decision at a time
            run_round(kk, (int)G.V_ch[p_chn[kk]], (int)G.V_sn[p_sn[kk]], (int)G.V_sp[p_sp[kk - 1]]);
    }
    if (staged) {
        __builtin_amdgcn_wave_barrier();
        u16* ow = out + wbase;
        const u32 head = wbase & 1u;
        if (head && lane == 0 && wtotal > 0) ow[0] = sg[0];
        const u32 body = (wtotal - (head < wtotal ? head : wtotal)) >> 1;
        u32* ow32 = reinterpret_cast<u32*>(ow + head);
        for (u32 t = lane; t < body; t += 64) ow32[t] = (u32)sg[head + 2 * t] | ((u32)sg[head + 2 * t + 1] << 16);
        if (lane == 0 && wtotal > head && ((wtotal - head) & 1u)) ow[wtotal - 1] = sg[wtotal - 1];
    }
}

__global__ void dc_poff_kernel(const u32* __restrict__ doff_sp, DcSub S, u32 m, u32* __restrict__ poff)
{
    const u32 b = threadIdx.x;
    if (b <= S.nb) poff[b] = doff_sp[b < S.nb ? S.first[b] : m];
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// Row -> bin ranges of the counting pass, derived from the model's own case analysis (the one dc_item_rounds walks): for every
// bin the rows a run of that bin contributes to, then per row the (at most two) ranges of consecutive bins.  Returns false if a
// row ever needed more — the layout of the bins would be wrong, and the device coder is then not offered at all.
static bool dc_build_rowbins(DcRowBins* T)
{
    static unsigned char member[DC_ROWS][DC_BINS];
    memset(member, 0, sizeof member);
    for (int lt = 0; lt < 2; ++lt) {
        for (u32 rank = 0; rank < 256; ++rank) {
            const int B = rank != 1u ? bsr(rank) : 0;
            if (lt && B == 0) continue;                                         // dc_rank_bin never produces these
            const u32 bin = rank | (lt ? 256u : 0u);
            const int e = B ? (B - 1) + lt : 0;
            member[TAU_RF][bin] = 1;
            for (int sx = 0; sx < e && sx < 7; ++sx) member[TAU_RE + sx][bin] = 1;
            for (int d = 0; d < B; ++d) member[TAU_RM + rm_off(B) + (int)(rank >> (B - d)) - 1][bin] = 1;
        }
    }
    for (u32 cls = 1; cls < 96; ++cls) {
        if (cls >= 64 && (cls < 70 || cls > 94)) continue;                      // run lengths below 64 have their own bin; 2^31 > run
        const u32 run = cls < 64 ? cls : 1u << (cls - 64);
        const u32 bin = dc_run_bin(run);
        if (bin != DC_BIN_RUN + cls) return false;
        const int nb = run != 1u ? bsr(run) : 0;
        member[TAU_NF][bin] = 1;
        for (int sx = 0; sx < nb; ++sx) member[TAU_NE + sx][bin] = 1;
        for (int d = 0; d < nb; ++d) member[TAU_NM + nm_off(nb) + (int)(nb <= 5 ? (run >> (nb - d)) : (u32)(1 + d)) - 1][bin] = 1;
    }
    for (int h = 0; h < DC_ROWS; ++h) {
        u16 r[4] = {0, 0, 0, 0}; int nr = 0;
        for (int b = 0; b < DC_BINS;) {
            if (!member[h][b]) { ++b; continue; }
            int e = b; while (e < DC_BINS && member[h][e]) ++e;
            if (nr == 2) return false;
            r[2 * nr] = (u16)b; r[2 * nr + 1] = (u16)e; ++nr;
            b = e;
        }
        T[h].lo1 = r[0]; T[h].hi1 = r[1]; T[h].lo2 = r[2]; T[h].hi2 = r[3];
    }
    return true;
}

static size_t dc_align(size_t x) { return (x + 255) / 256 * 256; }

void devcoder_destroy(bscgpu_ctx* c)
{
    DevCoder* d = c->dc;
    if (!d) return;
    if (d->arena) hipFree(d->arena);
    if (d->sp_arena) (void)hipFree(d->sp_arena);
    if (d->hmeta) hipHostFree(d->hmeta);
    delete d;
    c->dc = nullptr;
}


// The two parameter sets with their attainable-value closures: function-local statics, so whoever comes first computes them and
// everybody else waits for that.
static const ModelParams& dc_model_static()
{
    static const ModelParams mp = [] { ModelParams m; model_params_from_table(bschost::qlfc_static_params(), m); return m; }();
    return mp;
}
static const ModelParams& dc_model_fast()
{
    static const ModelParams mpf = [] { ModelParams m; model_params_fast(m); return m; }();
    return mpf;
}
// Called at the top of the process's first bscgpu_create: the ~60 ms of closure computation run beside the ~190 ms the HIP runtime
// takes to come up, instead of in front of the first block (a file of a few dozen blocks is done in about a second: bsc_mgpu, the CLI).
void devcoder_warm_tables()
{
    struct Warm {
        std::thread th;
        Warm() { try { th = std::thread([] { (void)dc_model_static(); (void)dc_model_fast(); }); } catch (const std::system_error&) {} }
        ~Warm() { if (th.joinable()) th.join(); }
    };
    static Warm warm;
}

int devcoder_ensure(bscgpu_ctx* c)
{
    if (c->dc) return BSC_NO_ERROR;
    // An arena that does not fit (~220 bytes per block byte on top of the sorter's ~60) is a reason to DECLINE, not an error: the
    // block takes the host model, as it did before the device model existed, and the allocation is not retried for every block.
    if (c->dc_alloc_failed) return BSC_NOT_SUPPORTED;
    if (getenv("BSC_DEVCODER_FAIL_ALLOC")) { c->dc_alloc_failed = true; return BSC_NOT_SUPPORTED; }      // tests: an arena that does not fit
    CtxTimer tm("devcoder_ensure (arena, tables)");
    DevCoder* d = new DevCoder();
    const size_t N = ((size_t)c->max_n + 4096 + 4095) / 4096 * 4096;
    d->Mcap = N; d->Dcap = 4 * N + 65536;
    const size_t M = d->Mcap + 64, D = d->Dcap + 64, NCH = d->Dcap / DC_EV + 16;
    struct Carve { void** p; size_t bytes; };
    Carve carve[] = {
        {(void**)&d->key_ch, 8 * M}, {(void**)&d->key_ch_s, 8 * M}, {(void**)&d->key_sr, 8 * M}, {(void**)&d->key_sr_s, 8 * M},
        {(void**)&d->key_sn, 8 * M}, {(void**)&d->key_sn_s, 8 * M},
        {(void**)&d->inv_ch, 4 * M}, {(void**)&d->inv_sr, 4 * M}, {(void**)&d->inv_sn, 4 * M}, {(void**)&d->ge32, M},
        {(void**)&d->doff[0], 4 * M}, {(void**)&d->doff[1], 4 * M}, {(void**)&d->doff[2], 4 * M}, {(void**)&d->doff[3], 4 * M},
        {(void**)&d->events[0], 2 * D}, {(void**)&d->events[1], 2 * D}, {(void**)&d->events[2], 2 * D}, {(void**)&d->events[3], 2 * D},
        {(void**)&d->pos[0], 4 * D}, {(void**)&d->pos[1], 4 * D}, {(void**)&d->pos[2], 4 * D}, {(void**)&d->pos[3], 4 * D},
        {(void**)&d->V[0], 2 * D}, {(void**)&d->V[1], 2 * D}, {(void**)&d->V[2], 2 * D}, {(void**)&d->V[3], 2 * D},
        {(void**)&d->ps[0], 2 * D}, {(void**)&d->ps[1], 2 * D},
        {(void**)&d->cnt, (size_t)DC_ROWS * DC_WCH_MAX * 4}, {(void**)&d->rowtot, DC_ROWS * 4}, {(void**)&d->rowstart, 6 * (DC_ROWS + 8) * 4},
        {(void**)&d->wdec, (DC_WCH_MAX + 8) * 4}, {(void**)&d->wdecoff, (DC_WCH_MAX + 8) * 4},
        {(void**)&d->elo, 2 * 4 * NCH}, {(void**)&d->ehi, 2 * 4 * NCH}, {(void**)&d->S, 2 * 4 * NCH},
        {(void**)&d->present, (size_t)DC_KIND_WORDS * 4}, {(void**)&d->rounds, 256},
        {(void**)&d->meta, DM_COUNT * 4}, {(void**)&d->poff, 16 * 4}, {(void**)&d->frag, (M / 64 + 64) * 2 * sizeof(DcFrag)},
        {(void**)&d->tab_rank, 32768}, {(void**)&d->tab_run, 8192}, {(void**)&d->mp, sizeof(ModelParams)}, {(void**)&d->mp_fast, sizeof(ModelParams)},
        {(void**)&d->rowbins, DC_ROWS * sizeof(DcRowBins)}, {(void**)&d->sink, 4096},
        {(void**)&d->sp_desc, sizeof d->sp_desc_host},
    };
    size_t total = 0;
    for (auto& cv : carve) total += dc_align(cv.bytes);
    if (hipMalloc((void**)&d->arena, total) != hipSuccess) { (void)hipGetLastError(); delete d; c->dc_alloc_failed = true; return BSC_NOT_SUPPORTED; }
    d->arena_bytes = total; d->nch_cap = NCH;
    size_t off = 0;
    for (auto& cv : carve) { *cv.p = d->arena + off; off += dc_align(cv.bytes); }
    if (hipHostMalloc((void**)&d->hmeta, 64 * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); hipFree(d->arena); delete d; c->dc_alloc_failed = true; return BSC_NOT_SUPPORTED; }
    // (once per process: the attainable-range closures behind the brackets take ~60 ms to compute, and every context needs the same tables;
    // devcoder_warm_tables starts them on a thread of their own while the first context is still being created)
    const ModelParams& mp = dc_model_static();
    const ModelParams& mpf = dc_model_fast();
    // more than 64 KB of dynamic LDS is a per-device attribute of the function: set for every context's device
    if (hipFuncSetAttribute((const void*)dc_eval_wave_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DC_EVAL_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)dc_eval_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DC_EVAL_LDS) != hipSuccess) {
        (void)hipFree(d->arena); (void)hipHostFree(d->hmeta); delete d;
        return ctx_fail(c, BSC_GPU_ERROR, "device coder: LDS attribute", hipSuccess);
    }
    static DcRowBins rowbins[DC_ROWS];
    static const bool rowbins_ok = dc_build_rowbins(rowbins);
    if (!rowbins_ok) { (void)hipFree(d->arena); (void)hipHostFree(d->hmeta); delete d; return ctx_fail(c, BSC_GPU_ERROR, "device coder: bin layout", hipSuccess); }
    const bschost::QlfcTables& QT = bschost::qlfc_tables();
    const uint8_t *rs = QT.rank_state, *ns = QT.run_state;
    d->sp_ok = true;
    for (int mr = 0; mr <= dcs::SP_MAXR; ++mr) d->sp_ok = dcs::sp_build_descs(mr, d->sp_desc_host[mr]) && d->sp_ok;
    const bool ok = hipMemcpyAsync(d->mp, &mp, sizeof mp, hipMemcpyHostToDevice, c->stream) == hipSuccess
                 && hipMemcpyAsync(d->sp_desc, d->sp_desc_host, sizeof d->sp_desc_host, hipMemcpyHostToDevice, c->stream) == hipSuccess
                 && hipMemcpyAsync(d->mp_fast, &mpf, sizeof mpf, hipMemcpyHostToDevice, c->stream) == hipSuccess
                 && hipMemcpyAsync(d->tab_rank, rs, 32768, hipMemcpyHostToDevice, c->stream) == hipSuccess
                 && hipMemcpyAsync(d->tab_run, ns, 8192, hipMemcpyHostToDevice, c->stream) == hipSuccess
                 && hipMemcpyAsync(d->rowbins, rowbins, sizeof rowbins, hipMemcpyHostToDevice, c->stream) == hipSuccess
                 && ctx_sync(c) == hipSuccess;                      // mp is a stack object: the copies must have finished
    if (!ok) { hipFree(d->arena); hipHostFree(d->hmeta); delete d; return ctx_fail(c, BSC_GPU_ERROR, "device coder tables", hipSuccess); }
    c->dc = d;
    return BSC_NO_ERROR;
}

int64_t devcoder_arena_bytes(const bscgpu_ctx* c) { return c->dc ? (int64_t)c->dc->arena_bytes : 0; }

// The buffers of the stream-order evaluation of the static family (devcoder_static.h; ~23 bytes per block byte): allocated when the
// option is first used — it is off by default, and six contexts' worth of them would be 9 GB of HBM nobody reads.
static bool dc_sp_ensure(DevCoder* d)
{
    if (d->sp_arena) return true;
    if (d->sp_alloc_failed) return false;
    const size_t M = d->Mcap + 64;
    struct Carve { void** p; size_t bytes; };
    Carve carve[] = {
        {(void**)&d->sp_planes, (M / 64 + 2) * dcs::SP_PLANES * 8},
        {(void**)&d->sp_sums, (size_t)dcs::SP_SLOTS * (M / 64 / dcs::SP_CT_MIN + 2) * sizeof(dcs::SpSum)}, {(void**)&d->sp_sv, (size_t)dcs::SP_SLOTS * (M / 64 / dcs::SP_CT_MIN + 2) * 2},
        {(void**)&d->sp_gsum, (size_t)dcs::SP_SLOTS * (M / 64 / dcs::SP_CT_MIN / dcs::SP_RG + 2) * sizeof(dcs::SpGroupSum)}, {(void**)&d->sp_gv, (size_t)dcs::SP_SLOTS * (M / 64 / dcs::SP_CT_MIN / dcs::SP_RG + 2) * 2},
        {(void**)&d->sp_state, (M / 64 + 16) * dcs::SP_LANES * 2 + 256}, {(void**)&d->sp_rec, 16 * M}, {(void**)&d->doff_full, 4 * M},
    };
    size_t total = 0;
    for (auto& cv : carve) total += dc_align(cv.bytes);
    if (hipMalloc((void**)&d->sp_arena, total) != hipSuccess) { (void)hipGetLastError(); d->sp_arena = nullptr; d->sp_alloc_failed = true; return false; }
    size_t off = 0;
    for (auto& cv : carve) { *cv.p = d->sp_arena + off; off += dc_align(cv.bytes); }
    return true;
}

template <int SIDES>
static void dc_launch_partition(bscgpu_ctx* c, DevCoder* d, const u64* items, u32 m, const DcSub& S, int job, u32 ignoreX)
{
    const DcGeom g = dc_geom(m);
    const u32 grid = (g.W + WAVES - 1) / WAVES;
    u32* rowstart = d->rowstart + (DC_ROWS + 8) * job;
    prof_begin(c, BSCGPU_K_DC_PART, (u64)m * 8, m);
    hipLaunchKernelGGL(dc_part_count_kernel<SIDES>, dim3(grid), dim3(WG), 0, c->stream, items, g, S, d->rowbins, d->cnt, d->wdec);
    hipLaunchKernelGGL(dc_scan_rows_kernel, dim3(DC_ROWS), dim3(WG), 0, c->stream, d->cnt, g.W, d->rowtot);
    hipLaunchKernelGGL(dc_scan_misc_kernel, dim3(1), dim3(WG), 0, c->stream, d->rowtot, rowstart, d->wdec, g.W, d->wdecoff, d->meta, job, (u32)d->Dcap);
    hipLaunchKernelGGL(dc_part_scatter_kernel<SIDES>, dim3(grid), dim3(WG), 0, c->stream, items, g, S, d->meta, d->cnt, rowstart,
                       d->wdecoff, ignoreX, d->events[job], d->pos[job], d->doff[job]);
    prof_end(c);
}

// Probability stream of a whole block.  Inputs: the QLFC front end's run arrays on the device (sym / rank / start, m runs of
// the n-byte sorted block), the sub-blocks' run ranges and max_rank values.  On success *D_out decisions were written to the
// device p stream (d->ps) and poff[0..nb] (decision offsets of the sub-blocks) to hmeta[32..]; returns BSC_NOT_SUPPORTED when
// the block has to go through the host model instead.
// The fast coder (-e0, qlfc.cpp:1135-1336) on the same machinery: its one counter per decision is indexed by the run's symbol, i.e. its
// chains ARE the char family's — (sub-block, decision type, symbol) — with shift updates and per-class targets (dcm::model_params_fast),
// no escape coding and the exponent always closed below 7 bits (max_rank = 7 in the static coder's terms).  So: items, ONE radix pass
// (symbol-major order), ONE partition job, the stream offsets of the runs, evaluation of that one job, and a p stream whose entries
// are the counter values themselves.  No contexts, no state tables, no blend: about a third of the static coder's device work.
static int devcoder_pstream_fast(bscgpu_ctx* c, DevCoder* d, const u8* dsym, const u8* drank, const u32* dstart, u32 m, u32 n, int nb,
                                 const u32* run_first, u32* D_out, u32* poff_out, int psbuf);

int devcoder_pstream(bscgpu_ctx* c, const u8* dsym, const u8* drank, const u32* dstart, u32 m, u32 n, int nb, const u32* run_first,
                     const int* max_rank, u32* D_out, u32* poff_out, u16* dbg, int psbuf, int coder, int* packed_out)
{
    if (packed_out) *packed_out = 0;
    int rc = devcoder_ensure(c);
    if (rc < 0) return rc;
    DevCoder* d = c->dc;
    if (m == 0 || m > d->Mcap || nb < 1 || nb > 8) return BSC_NOT_SUPPORTED;
    if (coder == 3) return devcoder_pstream_fast(c, d, dsym, drank, dstart, m, n, nb, run_first, D_out, poff_out, psbuf);
    DcSub S; S.nb = (u32)nb;
    for (int b = 0; b < 9; ++b) S.first[b] = (b <= nb) ? run_first[b] : m;
    for (int b = 0; b < 8; ++b) S.maxr[b] = (b < nb) ? (u32)max_rank[b] : 0u;

    HIP_TRY(c, hipMemsetAsync(d->meta, 0, DM_COUNT * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(d->present, 0, (size_t)DC_KIND_WORDS * 4, c->stream));
    const u32 gm = (m + WG - 1) / WG;
    const u32 gm8 = (gm + 7u) / 8u * 8u;            // kernels that use dc_virtual_block()

    prof_begin(c, BSCGPU_K_DC_CTX, (u64)m * 40, m);
    // avg' = (124 avg + 4 rank) >> 7 <= max(avg, rank) and rank < nsym <= 2^(max_rank + 1): with at most 32 symbols in every
    // sub-block the average never reaches 32 and the escape coding (qlfc.cpp:960) cannot occur
    bool may_escape = false;
    for (int b = 0; b < nb; ++b) may_escape |= max_rank[b] > 4;
    if (may_escape)
        hipLaunchKernelGGL(dc_avg_kernel, dim3(((m + DC_AVG_CH - 1) / DC_AVG_CH + WG - 1) / WG), dim3(WG), 0, c->stream, drank, m, S, d->ge32, d->meta);
    else
        HIP_TRY(c, hipMemsetAsync(d->ge32, 0, m, c->stream));
    // Static family in stream order (devcoder_static.h) when no sub-block has more than 32 symbols (then no escape coding either)
    bool spf = c->dc_spf != 0 && d->sp_ok && !may_escape && dc_sp_ensure(d);
    u32 sp_slots = 0;
    for (int b = 0; b < nb && spf; ++b) {
        if (max_rank[b] < 0 || max_rank[b] > dcs::SP_MAXR) { spf = false; break; }
        for (int s2 = 0; s2 < dcs::SP_SLOTS; ++s2) if (d->sp_desc_host[max_rank[b]][s2].present) sp_slots |= 1u << s2;
    }
    hipLaunchKernelGGL(dc_items_kernel, dim3(gm), dim3(WG), 0, c->stream, dsym, drank, dstart, d->ge32, m, n, S, d->key_ch, spf ? d->sp_planes : (u64*)nullptr);
    prof_end(c);
    SpArgs SA;
    if (spf) {
        SA.g = dcs::sp_geom(m); SA.S = S; SA.slots = sp_slots;
        SA.uniform = 1u; for (int b = 1; b < nb; ++b) if (max_rank[b] != max_rank[0]) SA.uniform = 0u; SA.planes = d->sp_planes; SA.desc = d->sp_desc; SA.mp = d->mp;
        SA.sums = d->sp_sums; SA.gsum = d->sp_gsum; SA.gv = d->sp_gv; SA.sv = d->sp_sv; SA.state = d->sp_state; SA.rec = d->sp_rec; SA.meta = d->meta;
        prof_begin(c, BSCGPU_K_DC_STATIC, (u64)m * 16, m);
        const dim3 wgrid((SA.g.cstride + WG - 1) / WG, dcs::SP_SLOTS - 1), rgrid((SA.g.gstride + WAVES - 1) / WAVES, dcs::SP_SLOTS - 1);
        hipLaunchKernelGGL(sp_walk_kernel<false>, wgrid, dim3(WG), 0, c->stream, SA);
        hipLaunchKernelGGL(sp_resolve_kernel<1>, rgrid, dim3(WG), 0, c->stream, SA);
        hipLaunchKernelGGL(sp_resolve_serial_kernel, dim3(dcs::SP_SLOTS - 1), dim3(64), 0, c->stream, SA);
        hipLaunchKernelGGL(sp_resolve_kernel<3>, rgrid, dim3(WG), 0, c->stream, SA);
        hipLaunchKernelGGL(sp_walk_kernel<true>, wgrid, dim3(WG), 0, c->stream, SA);
        hipLaunchKernelGGL(sp_values_kernel, dim3((SA.g.ntiles + WAVES - 1) / WAVES), dim3(WG), 0, c->stream, SA);
        prof_end(c);
    }
    RadixPass top; top.shift = 56; top.bits = 8;
    int in_alt = 0;
    // one pass: the engine reads `keys` (left intact) and writes the sorted copy to the alt array
    rc = radix_sort_passes(c, d->key_ch, d->key_ch_s, nullptr, nullptr, m, &top, 1, &in_alt, d->inv_ch);
    if (rc < 0) return rc;
    prof_begin(c, BSCGPU_K_DC_CTX, (u64)m * 40, m);
    hipLaunchKernelGGL(dc_ctx_kernel, dim3(gm8), dim3(WG), 0, c->stream, d->key_ch, d->key_ch_s, d->inv_ch, m, S, d->tab_rank, d->tab_run,
                       d->key_sr, d->key_sn, d->present, d->meta);
    hipLaunchKernelGGL(dc_setup_kernel, dim3(1), dim3(WG), 0, c->stream, d->present, S, d->rounds, d->meta);
    prof_end(c);
    rc = radix_sort_passes(c, d->key_sr, d->key_sr_s, nullptr, nullptr, m, &top, 1, &in_alt, d->inv_sr);
    if (rc < 0) return rc;
    rc = radix_sort_passes(c, d->key_sn, d->key_sn_s, nullptr, nullptr, m, &top, 1, &in_alt, d->inv_sn);
    if (rc < 0) return rc;

    // families: static (stream order, X ignored), char (symbol-major), state (rank side by rank-state, run side by run-state):
    // four independent partition jobs, then ONE read-back of their sizes, then all chains of the block in one set of launches
    if (spf) {
        // job 0 holds the family's NE / NM decisions only; the runs' offsets in the p stream (all decisions) come from counts and scans alone
        dc_launch_partition<6>(c, d, d->key_ch, m, S, 0, 1u);
        const DcGeom g = dc_geom(m);
        const u32 grid = (g.W + WAVES - 1) / WAVES;
        prof_begin(c, BSCGPU_K_DC_PART, (u64)m * 8, m);
        hipLaunchKernelGGL(dc_part_count_kernel<3>, dim3(grid), dim3(WG), 0, c->stream, d->key_ch, g, S, d->rowbins, d->cnt, d->wdec);
        hipLaunchKernelGGL(dc_scan_rows_kernel, dim3(DC_ROWS), dim3(WG), 0, c->stream, d->cnt, g.W, d->rowtot);     // (the row totals: the block's decision count)
        hipLaunchKernelGGL(dc_scan_misc_kernel, dim3(1), dim3(WG), 0, c->stream, d->rowtot, d->rowstart + (DC_ROWS + 8) * 5, d->wdec, g.W, d->wdecoff, d->meta, 5, (u32)d->Dcap);
        hipLaunchKernelGGL(dc_doff_kernel<3>, dim3(grid), dim3(WG), 0, c->stream, d->key_ch, g, S, d->meta, d->wdecoff, d->doff_full);
        prof_end(c);
    } else dc_launch_partition<3>(c, d, d->key_ch, m, S, 0, 1u);
    dc_launch_partition<3>(c, d, d->key_ch_s, m, S, 1, 0u);
    dc_launch_partition<1>(c, d, d->key_sr_s, m, S, 2, 0u);
    dc_launch_partition<2>(c, d, d->key_sn_s, m, S, 3, 0u);
    // (the sub-blocks' offsets in the stream are known from here on: the packed stream's layout is made of them)
    hipLaunchKernelGGL(dc_poff_kernel, dim3(1), dim3(16), 0, c->stream, spf ? d->doff_full : d->doff[0], S, m, d->poff);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(d->hmeta, d->meta, DM_COUNT * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d->hmeta + 32, d->poff, 16 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    if (d->hmeta[DM_FAIL] != 0) { c->dc_last_fail = (int)d->hmeta[DM_FAIL]; return BSC_NOT_SUPPORTED; }
    u32 poff_h[9];
    for (int b = 0; b <= nb; ++b) poff_h[b] = d->hmeta[32 + b];
    u32 E[4];
    for (int job = 0; job < 4; ++job) E[job] = d->hmeta[DM_D0 + job];
    const u32 Efull = spf ? d->hmeta[DM_DFULL] : E[0];                 // decisions of the block (spf: job 0 is the small NE / NM job)
    if (Efull != E[1] || Efull != E[2] + E[3]) return ctx_fail(c, BSC_GPU_ERROR, "device coder: decision counts of the families differ", hipSuccess);
    {
        DcEvalAll A;
        A.wstart[0] = 0; A.cstart[0] = 0; A.sink = d->sink;
        // One lane is one serial chain, so the walks are VALU-issue bound with ONE wavefront per SIMD (1024 of them); a few
        // wavefronts more than that and some SIMDs get two, which doubles the kernel's time.  Chunks are therefore DC_EV events
        // (what the brackets need to meet) or as many as it takes to stay at <= 1000 wavefronts (+ <= 4 of padding) per launch.
        {
            const u64 Etot = (u64)E[0] + E[1] + E[2] + E[3];
            u64 ev = (Etot + 64ull * 1000 - 1) / (64ull * 1000);
            ev = (ev + DC_EB - 1) / DC_EB * DC_EB;
            A.ev = ev < (u64)DC_EV ? (u32)DC_EV : (u32)ev;
        }
        for (int job = 0; job < 4; ++job) {
            const int fam = job == 0 ? FAM_STATIC : job == 1 ? FAM_CHAR : FAM_STATE;
            A.job[job].events = d->events[job]; A.job[job].E = E[job]; A.job[job].rowstart = d->rowstart + (DC_ROWS + 8) * job;
            A.job[job].fam = fam;
            A.V[job] = d->V[job];
            const u32 nch = (E[job] + A.ev - 1) / A.ev;
            A.wstart[job + 1] = A.wstart[job] + (nch + 63) / 64;
            A.cstart[job + 1] = A.cstart[job] + (nch + 63) / 64 * 64;           // chunk slots padded to whole wavefronts
        }
        if (A.cstart[4] > 4 * d->nch_cap) return ctx_fail(c, BSC_GPU_ERROR, "device coder: chunk table too small", hipSuccess);
        prof_begin(c, BSCGPU_K_DC_EVAL, (u64)(E[0] + E[1] + E[2] + E[3]) * 6, (u64)E[0] + E[1] + E[2] + E[3]);
        if (A.wstart[4] > 0) {
            hipLaunchKernelGGL(dc_mark_rows_kernel, dim3((4 * DC_ROWS + WG - 1) / WG), dim3(WG), 0, c->stream, A);
            hipLaunchKernelGGL(dc_eval_wave_kernel<false>, dim3((A.wstart[4] + DC_EVAL_WAVES - 1) / DC_EVAL_WAVES), dim3(64 * DC_EVAL_WAVES), DC_EVAL_LDS, c->stream, A, d->mp, d->meta, d->elo, d->ehi, (const u16*)nullptr, d->cnt);
            hipLaunchKernelGGL(dc_eval_b_kernel, dim3((A.cstart[4] + WG - 1) / WG), dim3(WG), 0, c->stream, A, d->mp, d->meta, d->elo, d->ehi, d->S);
            hipLaunchKernelGGL(dc_eval_wave_kernel<true>, dim3((A.wstart[4] + DC_EVAL_WAVES - 1) / DC_EVAL_WAVES), dim3(64 * DC_EVAL_WAVES), DC_EVAL_LDS, c->stream, A, d->mp, d->meta, (u16*)nullptr, (u16*)nullptr, d->S, d->cnt + 3 * 4096);
        }
        prof_end(c);
    }

    DcGather G;
    G.key_ch = d->key_ch; G.m = m; G.inv_ch = d->inv_ch; G.inv_sr = d->inv_sr; G.inv_sn = d->inv_sn;
    G.doff_sp = d->doff[0]; G.doff_ch = d->doff[1]; G.doff_sr = d->doff[2]; G.doff_sn = d->doff[3];
    G.pos_sp = d->pos[0]; G.pos_ch = d->pos[1]; G.pos_sr = d->pos[2]; G.pos_sn = d->pos[3];
    G.V_sp = d->V[0]; G.V_ch = d->V[1]; G.V_sr = d->V[2]; G.V_sn = d->V[3];
    G.sp_rec = d->sp_rec; G.doff_full = d->doff_full;
    prof_begin(c, BSCGPU_K_DC_PSTREAM, (u64)Efull * 26, Efull);
    if (c->ps_guard[psbuf & 1]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ps_guard[psbuf & 1], 0));     // the buffer's previous copy-out
    for (int b = 0; b < 8; ++b) if (c->ps_guard_sig[psbuf & 1][b]) (void)dma_wait(c->ps_guard_sig[psbuf & 1][b]);   // ... when it went through the DMA engine directly (two blocks ago: long landed)
    // 13 bits per decision (DcP13) when the caller can take it: not with the debug planes, not with the stream-order static family
    bool packed = packed_out != nullptr && c->dc_p13 != 0 && !spf && dbg == nullptr;
    DcP13 Q{};
    if (packed) {
        u32 pbase = 0;
        for (int b = 0; b < nb; ++b) { Q.pad[b] = pbase - poff_h[b]; pbase += (poff_h[b + 1] - poff_h[b] + 63u) / 64u * 64u; }
        Q.out = reinterpret_cast<u8*>(d->ps[psbuf & 1]); Q.frag = d->frag;
        if ((u64)pbase / 8u * 13u > 2ull * (u64)(d->Dcap + 64)) packed = false;        // (cannot happen: 13 / 8 of D + 8 x 64 decisions of padding against 2 D)
    }
    if (spf)         hipLaunchKernelGGL(dc_pstream_spf_kernel, dim3(gm8), dim3(WG), 0, c->stream, G, S, d->mp, d->meta, d->ps[psbuf & 1], dbg, Efull);
    else if (packed) {
        hipLaunchKernelGGL((dc_pstream_kernel<false, false, true>), dim3(gm8), dim3(WG), 0, c->stream, G, S, d->mp, d->meta, d->ps[psbuf & 1], dbg, Efull, Q);
        hipLaunchKernelGGL(dc_p13_join_kernel, dim3((gm8 * WAVES + WG - 1) / WG), dim3(WG), 0, c->stream, d->frag, gm8 * WAVES, Q.out, d->meta);
    } else           hipLaunchKernelGGL((dc_pstream_kernel<false, false, false>), dim3(gm8), dim3(WG), 0, c->stream, G, S, d->mp, d->meta, d->ps[psbuf & 1], dbg, Efull, Q);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(d->hmeta, d->meta, DM_COUNT * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    if (d->hmeta[DM_FAIL] != 0) { c->dc_last_fail = (int)d->hmeta[DM_FAIL]; return BSC_NOT_SUPPORTED; }
    if (packed && d->hmeta[DM_P13_OVER] != 0) {
        // 64 consecutive runs with more decisions than a wavefront's staging buffer holds (runs of thousands): this block's stream in the 2-byte form
        packed = false;
        prof_begin(c, BSCGPU_K_DC_PSTREAM, (u64)Efull * 26, Efull);
        hipLaunchKernelGGL((dc_pstream_kernel<false, false, false>), dim3(gm8), dim3(WG), 0, c->stream, G, S, d->mp, d->meta, d->ps[psbuf & 1], dbg, Efull, Q);
        prof_end(c);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, ctx_sync(c));
    }
    if (packed_out) *packed_out = packed ? 1 : 0;
    c->dc_last_fail = 0;
    c->dc_replays = (int)d->hmeta[DM_REPLAYS];
#if DC_EVAL_TIMING
    {
        std::vector<u32> h(6 * 4096);
        (void)hipMemcpy(h.data(), d->cnt, h.size() * 4, hipMemcpyDeviceToHost);
        const u32 nw = 1100;
        for (int pass = 0; pass < 2; ++pass) {
            const u32* t = h.data() + pass * 3 * 4096;
            u32 t0 = 0xffffffffu; for (u32 i = 0; i < nw && i < 4096; ++i) if (t[3 * i + 1] && t[3 * i] < t0) t0 = t[3 * i];
            fprintf(stderr, "[eval timing pass %d] wave: start(us) dur(us) maxslowbatches job\n", pass);
            for (u32 i = 0; i < nw && i < 4096; ++i) if (t[3 * i + 1]) fprintf(stderr, "  %u %.1f %.1f %u %u\n", i, (t[3 * i] - t0) / 100.0, t[3 * i + 1] / 100.0, t[3 * i + 2] & 0xffff, t[3 * i + 2] >> 16);
        }
    }
#endif
    if (getenv("BSCGPU_DEBUG")) fprintf(stderr, "[devcoder] decisions %u, types %u, rounds %u, chunks replayed %u, static family %s\n", Efull, d->hmeta[DM_NTYPES], d->hmeta[DM_NROUNDS], d->hmeta[DM_REPLAYS], spf ? "in stream order" : "partitioned");
    *D_out = Efull;
    for (int b = 0; b <= nb; ++b) poff_out[b] = d->hmeta[32 + b];
    return BSC_NO_ERROR;
}

static int devcoder_pstream_fast(bscgpu_ctx* c, DevCoder* d, const u8* dsym, const u8* drank, const u32* dstart, u32 m, u32 n, int nb,
                                 const u32* run_first, u32* D_out, u32* poff_out, int psbuf)
{
    DcSub S; S.nb = (u32)nb;
    for (int b = 0; b < 9; ++b) S.first[b] = (b <= nb) ? run_first[b] : m;
    for (int b = 0; b < 8; ++b) S.maxr[b] = 7u;                        // `if (bits < 7)` closes the exponent (qlfc.cpp:1204), whatever the alphabet
    HIP_TRY(c, hipMemsetAsync(d->meta, 0, DM_COUNT * 4, c->stream));
    const u32 gm = (m + WG - 1) / WG;
    const u32 gm8 = (gm + 7u) / 8u * 8u;
    prof_begin(c, BSCGPU_K_DC_CTX, (u64)m * 14, m);
    HIP_TRY(c, hipMemsetAsync(d->ge32, 0, m, c->stream));              // no escape coding in this coder
    hipLaunchKernelGGL(dc_items_kernel, dim3(gm), dim3(WG), 0, c->stream, dsym, drank, dstart, d->ge32, m, n, S, d->key_ch, (u64*)nullptr);
    prof_end(c);
    RadixPass top; top.shift = 56; top.bits = 8;
    int in_alt = 0;
    int rc = radix_sort_passes(c, d->key_ch, d->key_ch_s, nullptr, nullptr, m, &top, 1, &in_alt, d->inv_ch);
    if (rc < 0) return rc;
    // stream offsets of the runs (job 0: counts and scans only), then the one family's chains (job 1, symbol-major items)
    {
        const DcGeom g = dc_geom(m);
        const u32 grid = (g.W + WAVES - 1) / WAVES;
        prof_begin(c, BSCGPU_K_DC_PART, (u64)m * 8, m);
        hipLaunchKernelGGL(dc_part_count_kernel<3>, dim3(grid), dim3(WG), 0, c->stream, d->key_ch, g, S, d->rowbins, d->cnt, d->wdec);
        hipLaunchKernelGGL(dc_scan_rows_kernel, dim3(DC_ROWS), dim3(WG), 0, c->stream, d->cnt, g.W, d->rowtot);
        hipLaunchKernelGGL(dc_scan_misc_kernel, dim3(1), dim3(WG), 0, c->stream, d->rowtot, d->rowstart, d->wdec, g.W, d->wdecoff, d->meta, 0, (u32)d->Dcap);
        hipLaunchKernelGGL(dc_doff_kernel<3>, dim3(grid), dim3(WG), 0, c->stream, d->key_ch, g, S, d->meta, d->wdecoff, d->doff[0]);
        prof_end(c);
    }
    dc_launch_partition<3>(c, d, d->key_ch_s, m, S, 1, 0u);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(d->hmeta, d->meta, DM_COUNT * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    if (d->hmeta[DM_FAIL] != 0) { c->dc_last_fail = (int)d->hmeta[DM_FAIL]; return BSC_NOT_SUPPORTED; }
    const u32 E = d->hmeta[DM_D0 + 1];
    if (d->hmeta[DM_D0 + 0] != E) return ctx_fail(c, BSC_GPU_ERROR, "device coder (fast): decision counts of stream and chain order differ", hipSuccess);
    {
        DcEvalAll A;
        A.sink = d->sink;
        u64 ev = ((u64)E + 64ull * 1000 - 1) / (64ull * 1000);
        ev = (ev + DC_EB - 1) / DC_EB * DC_EB;
        A.ev = ev < (u64)DC_EV ? (u32)DC_EV : (u32)ev;
        A.wstart[0] = 0; A.cstart[0] = 0;
        for (int job = 0; job < 4; ++job) {
            A.job[job].events = d->events[job]; A.job[job].E = job == 1 ? E : 0u; A.job[job].rowstart = d->rowstart + (DC_ROWS + 8) * job;
            A.job[job].fam = FAM_CHAR;
            A.V[job] = d->V[job];
            const u32 nch = (A.job[job].E + A.ev - 1) / A.ev;
            A.wstart[job + 1] = A.wstart[job] + (nch + 63) / 64;
            A.cstart[job + 1] = A.cstart[job] + (nch + 63) / 64 * 64;
        }
        if (A.cstart[4] > 4 * d->nch_cap) return ctx_fail(c, BSC_GPU_ERROR, "device coder: chunk table too small", hipSuccess);
        prof_begin(c, BSCGPU_K_DC_EVAL, (u64)E * 6, (u64)E);
        if (A.wstart[4] > 0) {
            hipLaunchKernelGGL(dc_mark_rows_kernel, dim3((4 * DC_ROWS + WG - 1) / WG), dim3(WG), 0, c->stream, A);
            hipLaunchKernelGGL(dc_eval_wave_kernel<false>, dim3((A.wstart[4] + DC_EVAL_WAVES - 1) / DC_EVAL_WAVES), dim3(64 * DC_EVAL_WAVES), DC_EVAL_LDS, c->stream, A, d->mp_fast, d->meta, d->elo, d->ehi, (const u16*)nullptr, d->cnt);
            hipLaunchKernelGGL(dc_eval_b_kernel, dim3((A.cstart[4] + WG - 1) / WG), dim3(WG), 0, c->stream, A, d->mp_fast, d->meta, d->elo, d->ehi, d->S);
            hipLaunchKernelGGL(dc_eval_wave_kernel<true>, dim3((A.wstart[4] + DC_EVAL_WAVES - 1) / DC_EVAL_WAVES), dim3(64 * DC_EVAL_WAVES), DC_EVAL_LDS, c->stream, A, d->mp_fast, d->meta, (u16*)nullptr, (u16*)nullptr, d->S, d->cnt + 3 * 4096);
        }
        prof_end(c);
    }
    DcGather G;
    G.key_ch = d->key_ch; G.m = m; G.inv_ch = d->inv_ch; G.inv_sr = d->inv_ch; G.inv_sn = d->inv_ch;
    G.doff_sp = d->doff[0]; G.doff_ch = d->doff[1]; G.doff_sr = d->doff[1]; G.doff_sn = d->doff[1];
    G.pos_sp = d->pos[1]; G.pos_ch = d->pos[1]; G.pos_sr = d->pos[1]; G.pos_sn = d->pos[1];
    G.V_sp = d->V[1]; G.V_ch = d->V[1]; G.V_sr = d->V[1]; G.V_sn = d->V[1];
    prof_begin(c, BSCGPU_K_DC_PSTREAM, (u64)E * 10, E);
    if (c->ps_guard[psbuf & 1]) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ps_guard[psbuf & 1], 0));     // the buffer's previous copy-out
    for (int b = 0; b < 8; ++b) if (c->ps_guard_sig[psbuf & 1][b]) (void)dma_wait(c->ps_guard_sig[psbuf & 1][b]);   // ... when it went through the DMA engine directly (two blocks ago: long landed)
    hipLaunchKernelGGL((dc_pstream_kernel<true, false, false>), dim3(gm8), dim3(WG), 0, c->stream, G, S, d->mp_fast, d->meta, d->ps[psbuf & 1], (u16*)nullptr, E, DcP13{});
    hipLaunchKernelGGL(dc_poff_kernel, dim3(1), dim3(16), 0, c->stream, d->doff[0], S, m, d->poff);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(d->hmeta, d->meta, DM_COUNT * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d->hmeta + 32, d->poff, 16 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    if (d->hmeta[DM_FAIL] != 0) { c->dc_last_fail = (int)d->hmeta[DM_FAIL]; return BSC_NOT_SUPPORTED; }
    c->dc_last_fail = 0;
    c->dc_replays = (int)d->hmeta[DM_REPLAYS];
    if (getenv("BSCGPU_DEBUG")) fprintf(stderr, "[devcoder fast] decisions %u, chunks replayed %u\n", E, d->hmeta[DM_REPLAYS]);
    *D_out = E;
    for (int b = 0; b <= nb; ++b) poff_out[b] = d->hmeta[32 + b];
    return BSC_NO_ERROR;
}

const u16* devcoder_pstream_ptr(const bscgpu_ctx* c, int psbuf) { return c->dc ? c->dc->ps[psbuf & 1] : nullptr; }

// ---- C ABI: the stage on its own (host block in, probability stream out) ---------------------------------------------
// What a maintainer would call next to bsc_qlfc_transform-style stage functions, and what the parity tests compare with the
// oracle's trace of the reference model: sub-block split (coder.cpp:70-109), run / rank front end and the model on the GPU.
// Returns the number of decisions (their 16-bit entries are in out[0..)), BSC_NOT_SUPPORTED when the block needs the host
// model (bscgpu_last_error tells why), or another negative libbsc code.
static int64_t qlfc_static_pstream_stage(bscgpu_ctx* c, const uint8_t* L, int n, uint16_t* out, int64_t cap, int* nblocks_out,
                                         int* sub_start /*[8]*/, int* sub_size /*[8]*/, int64_t* poff_out /*[9]*/, uint16_t* dbg_out /*[3][cap] or NULL*/,
                                         bool want_packed, int64_t* pbase_out /*[9], packed only*/)
{
    if (!c || !L || !out || n <= 0 || !nblocks_out || !sub_start || !sub_size || !poff_out) return BSC_BAD_PARAMETER;
    if (n > c->max_n) return BSC_GPU_NOT_ENOUGH_MEMORY;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    HIP_TRY(c, hipMemcpyAsync(c->dL, L, (size_t)n, hipMemcpyHostToDevice, c->stream));
    const int nb = bschost::coder_num_blocks(n);
    int rc = qlfc_front_split(c, c->dL, (u32)n, nb, sub_start, sub_size);
    if (rc < 0) return rc;
    u32 m = 0, run_first[9];
    static thread_local u32 first_run[8 * 256];
    rc = qlfc_front_runs(c, c->dL, (u32)n, nb, sub_start, &m, run_first, first_run, c->slots[0]);
    if (rc < 0) return rc;
    int max_rank[8];
    for (int b = 0; b < nb; ++b) {
        int nsym = 0;
        for (int s = 0; s < 256; ++s) nsym += first_run[b * 256 + s] != 0xffffffffu;
        max_rank[b] = bsr((uint32_t)(nsym - 1));
    }
    u16* ddbg = nullptr;
    if (dbg_out) { rc = devcoder_ensure(c); if (rc < 0) return rc; if (hipMalloc((void**)&ddbg, (size_t)c->dc->Dcap * 6) != hipSuccess) return BSC_GPU_NOT_ENOUGH_MEMORY; }
    u32 D = 0, poff[9];
    int packed = 0;
    rc = devcoder_pstream(c, reinterpret_cast<const u8*>(c->vA), reinterpret_cast<const u8*>(c->vB), c->SA, m, (u32)n, nb, run_first, max_rank, &D, poff, ddbg, 0, 1,
                          want_packed ? &packed : nullptr);
    if (rc == BSC_NO_ERROR && want_packed && !packed) { c->err = "the packed stream was not produced (option off, or a wavefront's piece beyond the staging buffer)"; rc = BSC_NOT_SUPPORTED; }
    else if (rc == BSC_NOT_SUPPORTED) {
        char buf[96]; snprintf(buf, sizeof buf, "device coder declined the block (reason mask %d)", c->dc_last_fail);
        c->err = buf;
    }
    if (rc < 0) { if (ddbg) hipFree(ddbg); return rc; }
    *nblocks_out = nb;
    for (int b = 0; b <= nb; ++b) poff_out[b] = poff[b];
    if (want_packed) {
        int64_t pb = 0;
        for (int b = 0; b < nb; ++b) { pbase_out[b] = pb; pb += ((int64_t)(poff[b + 1] - poff[b]) + 63) / 64 * 64; }
        pbase_out[nb] = pb;
        const int64_t bytes = pb / 8 * 13;
        if (bytes <= cap) { HIP_TRY(c, hipMemcpyAsync(out, devcoder_pstream_ptr(c), (size_t)bytes, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(c, ctx_sync(c)); }
    } else if ((int64_t)D <= cap) {
        HIP_TRY(c, hipMemcpyAsync(out, devcoder_pstream_ptr(c), (size_t)D * 2, hipMemcpyDeviceToHost, c->stream));
        if (dbg_out) for (int f = 0; f < 3; ++f) HIP_TRY(c, hipMemcpyAsync(dbg_out + (size_t)f * cap, ddbg + (size_t)f * D, (size_t)D * 2, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, ctx_sync(c));
    }
    if (ddbg) hipFree(ddbg);
    prof_collect(c);
    return (int64_t)D;
}
extern "C" int64_t bscgpu_qlfc_static_pstream(bscgpu_ctx* c, const uint8_t* L, int n, uint16_t* out, int64_t cap, int* nblocks_out,
                                              int* sub_start /*[8]*/, int* sub_size /*[8]*/, int64_t* poff_out /*[9]*/, uint16_t* dbg_out /*[3][cap] or NULL*/)
{
    return qlfc_static_pstream_stage(c, L, n, out, cap, nblocks_out, sub_start, sub_size, poff_out, dbg_out, false, nullptr);
}
// The same stage with the stream as it crosses PCIe since round 6 (DcP13): out[0 .. pbase[nb] / 8 * 13) bytes (cap in BYTES), sub-block b's
// fields from decision pbase[b] of the packed space on (13 bits each, eight in 13 bytes); poff as above.  BSC_NOT_SUPPORTED also when the
// packed form was not produced for this block (BSCGPU_OPT_DC_PACKED_STREAM off, or 64 consecutive runs with more decisions than a
// wavefront stages: bsc_compress then takes the 16-bit form for that block).
extern "C" int64_t bscgpu_qlfc_static_pstream_packed(bscgpu_ctx* c, const uint8_t* L, int n, uint8_t* out, int64_t cap_bytes, int* nblocks_out,
                                                     int* sub_start /*[8]*/, int* sub_size /*[8]*/, int64_t* poff_out /*[9]*/, int64_t* pbase_out /*[9]*/)
{
    if (!pbase_out) return BSC_BAD_PARAMETER;
    return qlfc_static_pstream_stage(c, L, n, reinterpret_cast<uint16_t*>(out), cap_bytes, nblocks_out, sub_start, sub_size, poff_out, nullptr, true, pbase_out);
}
