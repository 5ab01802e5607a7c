// radix_sort.hip — stable LSD radix sort of (u64 key [, u32 value]) records for gfx950 (MI355X).
//
// This is the engine under both block sorters: the BWT's suffix sort (initial 8-byte-prefix sort and
// every prefix-doubling round) and the Sort Transform.  It replaces the reference's calls into
// cub::DeviceRadixSort / cub::DeviceSegmentedSort (libcubwt.cu:718, :1691, :2136-2163; st.cu:187-193,
// :273-279) with a design written for CDNA4, not a translation of CUB:
//
//  * One digit pass = three launches on one stream: rs_hist (per-chunk digit counts of the pass's
//    input), rs_scan (256 workgroups, one per digit: exclusive scan over chunks), rs_scatter (the
//    graded kernel).  There is NO inter-workgroup communication inside a launch: on MI355X the eight
//    XCD L2s are not coherent and an agent-scope hand-off costs ~1-3 us under streaming load
//    (MI355X_MICROARCH.md, handoff rows; 4.6 us per hop in tools/ubench.hip), while the chip retires a
//    196 KB tile every ~50 ns: with 256 tiles in flight a decoupled look-back ("onesweep") has to sum ~20
//    predecessor rows per tile through L2-bypassing loads; measured under streaming load it adds 0.07-0.09 ms
//    per pass on the tiles' critical path (tools/ubench_lookback.hip) and gives up the chunk-contiguous
//    output segments write combining needs.
//    The price paid instead is one extra streaming read of the keys per pass (8 B/record, rs_hist, at
//    4.5 TB/s); the scatter pass itself moves exactly the algorithmic 2*(8+4) B/record.
//  * <= 1024 chunks for rs_hist (one 256-thread workgroup each, 4 per CU, all resident; block b runs on XCD b % 8 so
//    every XCD streams an equal contiguous share).  rs_scatter comes in two shapes: 256 threads x 16 records (4096-
//    record tiles, one chunk per workgroup) for keys-only passes and small inputs, and 1024 threads x 8 records
//    (8192-record tiles, one workgroup per CU walking four chunks) for large (key, value) passes, where the longer
//    per-digit runs halve the number of partially written lines.  A workgroup walks its records tile by tile keeping
//    its 256 running global bucket offsets in LDS, so the per-chunk offsets table is only 256 x 1024 u32.
//  * rs_scatter_tiled (large (key, value) passes, the default; BSC_RS_ORDER) gives up the contiguous range per workgroup:
//    the tiles are interleaved so that an XCD always works on 32 consecutive tiles, neighbouring runs of a digit are written
//    at the same time by CUs that share an L2, and the offsets are per tile.  0.60-0.62 of 8 TB/s on the BWT's keys.
//  * rs_scatter_wc (BSC_RS_ORDER=0) keeps the contiguous ranges and additionally holds the records of a digit that do not yet
//    fill a 128-B line in LDS: keys leave only as whole 16-key lines, values as whole 32-value lines (0.55).
//  * Inside a tile: wave-striped coalesced loads (each wave64 load instruction covers 512 contiguous
//    bytes of keys), 8-bit digit, stable in-wave ranking by wave64 ballot match (rs_match8: 8 ballots -> peer
//    mask, 32 hand-scheduled VALU instructions per record), per-wave 256-bin histograms in LDS, then the tile is locally
//    reordered through LDS (32 / 64 KB staging) so that every bucket leaves as one contiguous run:
//    consecutive lanes store consecutive addresses.  Integer/index work only — no MFMA.
#include "dev_common.h"
#include "radix_dev.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// Streamed-once inputs are read with non-temporal loads in rs_hist and rs_scatter (+10-20 %: they stop evicting the
// partially written output lines from L2); non-temporal *stores* were measured 20-50 % slower and are not used.
#ifndef RS_NT
#define RS_NT 5      // bit2: nt loads in rs_hist (A/B builds)
#endif
#ifndef RS_WG
#define RS_WG 256      // shape for keys-only passes and small inputs (the second shape below serves large pair passes)
#endif
#ifndef RS_WC_DEFAULT
#define RS_WC_DEFAULT 1        // default of BSC_RS_WC: write-combining scatter for large (key, value) passes
#endif
// Phase timing (debug builds, -DRS_PHASE_TIMING=1): thread 0 of every workgroup stamps s_memtime at the phase boundaries of
// its first 32 tiles into the context's scratch buffer; radix_sort_passes dumps the last pass to gpurun_out/phase_timing.bin.
#ifndef RS_PHASE_TIMING
#define RS_PHASE_TIMING 0
#endif
#if RS_PHASE_TIMING
#define RS_PH(i) do { if (t == 0 && tile_no < 32u) tdbg[((size_t)blockIdx.x * 32 + tile_no) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RS_PH(i) do { } while (0)
#endif
constexpr int RS_WAVES = RS_WG / 64;
#ifndef RS_ITEMS_N
#define RS_ITEMS_N 16
#endif
constexpr int RS_ITEMS = RS_ITEMS_N;
constexpr int RS_TILE  = RS_WG * RS_ITEMS;       // 4096 records per tile
constexpr int RS_LDS   = RS_TILE * 8 + RS_WAVES * 256 * 4 + 3 * 256 * 4 + 16 * 4;   // 40,000 B -> 4 WG (16 waves) / CU
constexpr int RS_MAX_CHUNKS = 256 * (1024 / RS_WG);       // all workgroups resident at once: 4 per CU
// second shape of rs_scatter for large (key, value) passes
#ifndef RS_BIG_PAIRS
#define RS_BIG_PAIRS 1
#endif
#ifndef RSB_ITEMS_N
#define RSB_ITEMS_N 8
#endif
#ifndef RSB_PIPE
#define RSB_PIPE 1      // software-pipelined loads in the 1024 x 8 shape (A/B builds: 0)
#endif
constexpr int RSB_WG = 1024, RSB_ITEMS = RSB_ITEMS_N, RSB_SPAN = 4;
constexpr int RSB_LDS = RSB_WG * RSB_ITEMS * 8 + (RSB_WG / 64) * 256 * 4 + 3 * 256 * 4 + 16 * 4;      // 85,056 B -> 1 WG (16 waves) / CU

static inline Chunking rs_chunking(u64 n) {
    // measured: 768 / 512 / 256 chunks (fewer open output lines per XCD, but fewer waves) are 1 / 5 / 40 % slower
    constexpr u32 RS_MAX_CHUNKS_RT = RS_MAX_CHUNKS;
    Chunking c;
    c.num_tiles   = (u32)((n + RS_TILE - 1) / RS_TILE);
    if (c.num_tiles == 0) c.num_tiles = 1;
    c.chunk_tiles = (c.num_tiles + RS_MAX_CHUNKS_RT - 1) / RS_MAX_CHUNKS_RT;
    c.num_chunks  = (c.num_tiles + c.chunk_tiles - 1) / c.chunk_tiles;
    return c;
}

// ---------------------------------------------------------------------------------------------
// rs_hist: per-chunk digit histogram.  counts layout [digit][chunk].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_WG) void rs_hist_kernel(const u64* __restrict__ keys, u32 n, int shift, u32 mask,
                                                        u32 chunk_tiles, u32 num_chunks, u32* __restrict__ counts)
{
    // BWT keys are text: a handful of digit values dominate, so a single 256-bin histogram per wave serialises its
    // LDS atomics (measured 2.4 TB/s with every lane on one bin).  Each wave keeps HREP replicas selected by the
    // low lane bits, which cuts the worst case from 64-way to 8-way conflicts; replicas are summed at the end.
    constexpr int WG = RS_WG, WAVES = RS_WAVES, HREP = 32 / RS_WAVES;
    __shared__ u32 h[WAVES * HREP * 256];
    const u32 t = threadIdx.x, w = t >> 6;
    for (u32 i = t; i < WAVES * HREP * 256; i += WG) h[i] = 0;
    __syncthreads();

    const u64 start = (u64)blockIdx.x * chunk_tiles * RS_TILE;
    u64 end = start + (u64)chunk_tiles * RS_TILE;
    if (end > n) end = n;
    u32* hw = h + (w * HREP + (t & (HREP - 1))) * 256;

    // 16-byte loads (2 keys per lane), 4 in flight per lane.
    u64 i = start + 2 * t;
    for (; i + 3 * 2 * WG + 1 < end; i += 4 * 2 * WG) {
        ulonglong2 a, b, c, d;
        if (RS_NT & 4) {
            a.x = __builtin_nontemporal_load(keys + i);          a.y = __builtin_nontemporal_load(keys + i + 1);
            b.x = __builtin_nontemporal_load(keys + i + 2 * WG); b.y = __builtin_nontemporal_load(keys + i + 2 * WG + 1);
            c.x = __builtin_nontemporal_load(keys + i + 4 * WG); c.y = __builtin_nontemporal_load(keys + i + 4 * WG + 1);
            d.x = __builtin_nontemporal_load(keys + i + 6 * WG); d.y = __builtin_nontemporal_load(keys + i + 6 * WG + 1);
        } else {
            a = *reinterpret_cast<const ulonglong2*>(keys + i);
            b = *reinterpret_cast<const ulonglong2*>(keys + i + 2 * WG);
            c = *reinterpret_cast<const ulonglong2*>(keys + i + 4 * WG);
            d = *reinterpret_cast<const ulonglong2*>(keys + i + 6 * WG);
        }
        atomicAdd(&hw[(u32)(a.x >> shift) & mask], 1u); atomicAdd(&hw[(u32)(a.y >> shift) & mask], 1u);
        atomicAdd(&hw[(u32)(b.x >> shift) & mask], 1u); atomicAdd(&hw[(u32)(b.y >> shift) & mask], 1u);
        atomicAdd(&hw[(u32)(c.x >> shift) & mask], 1u); atomicAdd(&hw[(u32)(c.y >> shift) & mask], 1u);
        atomicAdd(&hw[(u32)(d.x >> shift) & mask], 1u); atomicAdd(&hw[(u32)(d.y >> shift) & mask], 1u);
    }
    for (; i < end; i += 2 * WG) {
        atomicAdd(&hw[(u32)(keys[i] >> shift) & mask], 1u);
        if (i + 1 < end) atomicAdd(&hw[(u32)(keys[i + 1] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (t < 256) {
        u32 sum = 0;
#pragma unroll 8
        for (int r = 0; r < WAVES * HREP; ++r) sum += h[r * 256 + t];
        counts[(size_t)t * num_chunks + blockIdx.x] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// rs_scan: workgroup d turns row d of counts into exclusive per-chunk offsets and its row total.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void rs_scan_kernel(u32* __restrict__ counts, u32 num_chunks, u32* __restrict__ rowtot)
{
    __shared__ u32 scr[8];
    u32* row = counts + (size_t)blockIdx.x * num_chunks;
    u32 carry = 0;
    for (u32 base = 0; base < num_chunks; base += WG) {
        const u32 i = base + threadIdx.x;
        const u32 v = (i < num_chunks) ? row[i] : 0u;
        u32 tot;
        const u32 ex = block_excl_sum(v, scr, &tot);
        if (i < num_chunks) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}

// ---------------------------------------------------------------------------------------------
// rs_scatter: the digit pass.  Reads each record once, writes it once.
// ---------------------------------------------------------------------------------------------
// Two shapes of the same kernel (SHAPE = workgroup size x records per lane):
//   256 x 16 (4096-record tiles, 4 workgroups per CU, one rs_hist chunk each)   - keys-only passes and small inputs;
//   1024 x 8 (8192-record tiles, 1 workgroup per CU walking SPAN = 4 rs_hist chunks) - large (key, value) passes: twice the
//            records per digit and tile, i.e. half as many partially written lines per byte.  Same box, 64 MiB BWT:
//            first-sort pass 0.447 -> 0.410 ms, whole BWT 11.5 -> 10.8 ms; keys-only (ST) is 6 % slower with it.
// EMIT_POS: also write, for every input record (in input order, coalesced), the index it lands on — the inverse of the
// pass's permutation, which the device coder needs to find a run inside its sorted copies (devcoder.hip).
// PIPE (the 1024 x 8 shape): the next tile's keys are requested as soon as this tile's keys sit in the staging area, its
// values as soon as this tile's values do — ahead of the tile's stores in the wave's memory queue, into the registers the
// current tile has just given up.  Phase stamps (RS_PHASE_TIMING) showed a one-workgroup-per-CU tile spending 60 % of its time
// waiting: first for the previous tile's stores to drain before its own loads could even issue, then for those loads.
template <bool HAS_VAL, int WGSZ, int ITEMS, int SPAN, bool EMIT_POS = false, bool PIPE = false>
__global__ __launch_bounds__(WGSZ, 4) void rs_scatter_kernel(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                        const u32* __restrict__ vin, u32* __restrict__ vout,
                                                        u32 n, int shift, u32 mask,
                                                        u32 chunk_tiles, u32 num_chunks, u32 num_tiles,
                                                        const u32* __restrict__ offsets,
                                                        const u32* __restrict__ rowtot, u64* __restrict__ sink,
                                                        u32* __restrict__ dstpos = nullptr)
{
    constexpr int WG = WGSZ, WAVES = WGSZ / 64, TILE = WGSZ * ITEMS;
    u64* const tdbg = sink; (void)tdbg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* skeys  = reinterpret_cast<u64*>(smem);                       // [TILE] staging (reused as u32 for values)
    u32* whist  = reinterpret_cast<u32*>(smem + TILE * 8);            // [WAVES][256] per-wave digit counts / prefixes
    u32* goff   = whist + WAVES * 256;                                // [256] running global bucket offsets
    u32* adj    = goff + 256;                                         // [256] goff - tile-local bucket start
    u32* dstart = adj + 256;                                          // [256] (scratch)
    u32* scr    = dstart + 256;                                       // [8]
    lds_vu32* vwh = (lds_vu32*)whist;

    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;

    {   // global offset of this chunk's first record of digit t
        u32 tot;
        const u32 base = rs_digit_excl_sum<WAVES>(t < 256 ? rowtot[t] : 0u, scr, &tot);
        if (t < 256) goff[t] = base + offsets[(size_t)t * num_chunks + (size_t)blockIdx.x * SPAN];
    }
    // every wave owns (and re-zeroes, see below) its 256 counters
#pragma unroll
    for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;
    __syncthreads();

    // this workgroup's records: SPAN consecutive rs_hist chunks (chunk_tiles tiles of RS_TILE records each); n < 2^32 - 2^20
    const u32 rec0 = (u32)((u64)blockIdx.x * SPAN * chunk_tiles * RS_TILE);
    u32 rec1;
    { const u64 e = (u64)rec0 + (u64)SPAN * chunk_tiles * RS_TILE; rec1 = e > n ? n : (u32)e; }

    // wave-striped ownership: wave w holds records [w*64*ITEMS, (w+1)*64*ITEMS) of the tile, item i = 64 consecutive records
    const u32 wbase = w * (64 * ITEMS) + lane;
    u64 k[ITEMS];
    u32 v[ITEMS];
    // PIPE loads never sit behind a branch: a lane past the end of the workgroup's records reads its first record instead
    // (one cache line for the whole wave) and the value is replaced where it is used.
    auto prefetch_keys = [&](const u32 tb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = e < rec1 ? e : rec0; k[i] = __builtin_nontemporal_load(&kin[e]); }
    };
    auto prefetch_vals = [&](const u32 tb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = e < rec1 ? e : rec0; v[i] = __builtin_nontemporal_load(&vin[e]); }
    };
    if (PIPE) { prefetch_keys(rec0); if (HAS_VAL) prefetch_vals(rec0); }

    // Full tiles run without any branch around a global load or store, so the compiler can wait with exact vmcnt values
    // (a guarded memory instruction makes the number of outstanding operations unknown and later waits conservative);
    // the one partial tile of the whole input is handled by a second, guarded instantiation after the loop.
    // Barriers per tile: after ranking, inside and after the digit scan, after the staging writes, and two around the value
    // staging.  None at the end of a tile: what the next tile writes first (its own wave's counters, then — behind its first
    // barrier — adj / goff) is not read by anyone still in this tile's write-out.
    auto do_tile = [&](const u32 tbase, const u32 nvalid, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const u32 tile_no = (tbase - rec0) / TILE; (void)tile_no;
        RS_PH(0);

        if (!PIPE) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const u32 idx = wbase + i * 64;
                if (FULL) k[i] = __builtin_nontemporal_load(&kin[(u64)tbase + idx]);
                else k[i] = (idx < nvalid) ? __builtin_nontemporal_load(&kin[(u64)tbase + idx]) : ~0ull;
            }
        } else if (!FULL) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) if (wbase + i * 64 >= nvalid) k[i] = ~0ull;       // padding sorts last
        }
        RS_PH(1);

        // ---- stable in-wave ranking by ballot match (rs_rank_wave) --------------------------------
        u32 rk[ITEMS];
        rs_rank_wave<ITEMS>(k, shift, mask, vwh + w * 256, rk);
        RS_PH(2);
        // values are fetched only now (they are not needed for ranking): keeps the ranking loop's
        // register footprint at 4 waves/SIMD, and the loads fly under the bucket scan + key reorder.
        if (HAS_VAL && !PIPE) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const u32 idx = wbase + i * 64;
                if (FULL) v[i] = __builtin_nontemporal_load(&vin[(u64)tbase + idx]);
                else v[i] = (idx < nvalid) ? __builtin_nontemporal_load(&vin[(u64)tbase + idx]) : 0u;
            }
        }
        __syncthreads();
        RS_PH(3);

        // ---- per digit: wave prefixes, tile-local bucket start, global adjust -------------------
        {
            u32 c[WAVES];
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { c[i] = whist[i * 256 + t]; tot += c[i]; }
            }
            u32 all;
            const u32 ds = rs_digit_excl_sum<WAVES, false>(tot, scr, &all);
            if (t < 256) {
                u32 run = ds;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { whist[i * 256 + t] = run; run += c[i]; }
                const u32 g = goff[t];
                adj[t]  = g - ds;
                goff[t] = g + tot;
            }
        }
        __syncthreads();
        RS_PH(4);

        // ---- local reorder through LDS --------------------------------------------------------
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u32 d = (u32)(k[i] >> shift) & mask;
            const u32 pos = whist[w * 256 + d] + rk[i];
            rk[i] = pos;
            skeys[pos] = k[i];
            if (EMIT_POS) { const u32 idx = wbase + i * 64; if (FULL || idx < nvalid) dstpos[(u64)tbase + idx] = adj[d] + pos; }
        }
        if (PIPE) prefetch_keys(tbase + TILE);
        RS_PH(5);
        __syncthreads();
        RS_PH(6);
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;       // for the next tile's ranking (own wave's counters only)

        u32 dd[ITEMS / 4];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 q = j * WG + t;
            const u64 key = skeys[q];
            const u32 d = (u32)(key >> shift) & mask;
            if ((j & 3) == 0) dd[j >> 2] = d; else dd[j >> 2] |= d << (8 * (j & 3));
            if (FULL || q < nvalid) kout[adj[d] + q] = key;
        }
        RS_PH(7);

        if (HAS_VAL) {
            __syncthreads();
            u32* svals = reinterpret_cast<u32*>(skeys);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) svals[rk[i]] = v[i];
            if (PIPE) prefetch_vals(tbase + TILE);
            __syncthreads();
            RS_PH(8);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 q = j * WG + t;
                const u32 d = (dd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                if (FULL || q < nvalid) vout[adj[d] + q] = svals[q];
            }
        }
        RS_PH(9);
        RS_PH(10);
        };
    u32 tbase = rec0;
    for (; tbase + TILE <= rec1; tbase += TILE) do_tile(tbase, (u32)TILE, std::true_type());
    if (tbase < rec1) do_tile(tbase, rec1 - tbase, std::false_type());
}

// ---------------------------------------------------------------------------------------------
// rs_scatter_tiled (large (key, value) passes; BSC_RS_ORDER): the 1024 x 8 digit pass with the tiles INTERLEAVED over the workgroups instead
// of one contiguous range per workgroup: in round it workgroup b takes tile it * 256 + (b % 8) * 32 + b / 8, so an XCD
// (workgroups b % 8) always works on 32 consecutive tiles.  Neighbouring runs of a digit are then written at about the same
// time by CUs that share an L2, which can merge their partial lines, and the reads of a round cover one contiguous 50 MB
// stretch (tools/ubench_wc.hip: misaligned runs 0.541 -> 0.445 ms; tools/ubench_lookback.hip: tile copy 6.05 against 5.33 TB/s).
// The price: offsets per TILE instead of per chunk (counts table [256][tiles], rs_hist launched with one tile per workgroup),
// fetched per tile (one round ahead), and no state carried from tile to tile (no write combining).
// ---------------------------------------------------------------------------------------------
constexpr int RST_LDS = RSB_WG * RSB_ITEMS * 8 + (RSB_WG / 64) * 256 * 4 + 3 * 256 * 4 + 16 * 4;

template <bool HAS_VAL>
__global__ __launch_bounds__(RSB_WG) void rs_scatter_tiled_kernel(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                                  const u32* __restrict__ vin, u32* __restrict__ vout,
                                                                  u32 n, int shift, u32 mask, u32 num_tiles,
                                                                  const u32* __restrict__ offsets /*[256][num_tiles]*/,
                                                                  const u32* __restrict__ rowtot)
{
    constexpr int WG = RSB_WG, WAVES = RSB_WG / 64, ITEMS = RSB_ITEMS, TILE = RSB_WG * RSB_ITEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* skeys  = reinterpret_cast<u64*>(smem);
    u32* whist  = reinterpret_cast<u32*>(smem + TILE * 8);
    u32* adj    = whist + WAVES * 256;
    u32* scr    = adj + 3 * 256;
    lds_vu32* vwh = (lds_vu32*)whist;
    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;

    u32 dbase;                                            // first output position of digit t & 255
    {
        u32 tot;
        dbase = rs_digit_excl_sum<WAVES>(t < 256 ? rowtot[t] : 0u, scr, &tot);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;
    __syncthreads();

    const u32 wbase = w * (64 * ITEMS) + lane;
    u64 k[ITEMS];
    u32 v[ITEMS];
    auto prefetch_keys = [&](const u32 tile) __attribute__((always_inline)) {
        const u32 tb = tile * (u32)TILE;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = (tile < num_tiles && e < n) ? e : 0u; k[i] = __builtin_nontemporal_load(&kin[e]); }
    };
    auto prefetch_vals = [&](const u32 tile) __attribute__((always_inline)) {
        const u32 tb = tile * (u32)TILE;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = (tile < num_tiles && e < n) ? e : 0u; v[i] = __builtin_nontemporal_load(&vin[e]); }
    };
    auto fetch_offset = [&](const u32 tile) __attribute__((always_inline)) -> u32 {          // every thread loads (no branch around a load)
        return offsets[(size_t)(t & 255u) * num_tiles + (tile < num_tiles ? tile : num_tiles - 1)];
    };
#ifndef RST_XCD_RUN
#define RST_XCD_RUN 32          // consecutive tiles an XCD works on at a time (A/B builds: 8, 1 = plain round robin)
#endif
    const u32 first = ((blockIdx.x >> 3) / (u32)RST_XCD_RUN) * (8u * (u32)RST_XCD_RUN) + (blockIdx.x & 7u) * (u32)RST_XCD_RUN + ((blockIdx.x >> 3) % (u32)RST_XCD_RUN);
    u32 off_next = fetch_offset(first);
    prefetch_keys(first);
    if (HAS_VAL) prefetch_vals(first);

    auto do_tile = [&](const u32 tile, const u32 nvalid, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const u32 tbase = tile * (u32)TILE;
        const u32 off_cur = off_next;
        off_next = fetch_offset(tile + 256);
        if (!FULL) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) if (wbase + i * 64 >= nvalid) k[i] = ~0ull;
        }
        u32 rk[ITEMS];
        rs_rank_wave<ITEMS>(k, shift, mask, vwh + w * 256, rk);
        __syncthreads();
        {
            u32 c[WAVES];
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { c[i] = whist[i * 256 + t]; tot += c[i]; }
            }
            u32 all;
            const u32 ds = rs_digit_excl_sum<WAVES, false>(tot, scr, &all);
            if (t < 256) {
                u32 run = ds;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { whist[i * 256 + t] = run; run += c[i]; }
                adj[t] = dbase + off_cur - ds;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u32 d = (u32)(k[i] >> shift) & mask;
            const u32 pos = whist[w * 256 + d] + rk[i];
            rk[i] = pos;
            skeys[pos] = k[i];
        }
        prefetch_keys(tile + 256);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;
        u32 dd[ITEMS / 4];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 q = j * WG + t;
            const u64 key = skeys[q];
            const u32 d = (u32)(key >> shift) & mask;
            if ((j & 3) == 0) dd[j >> 2] = d; else dd[j >> 2] |= d << (8 * (j & 3));
            if (FULL || q < nvalid) kout[adj[d] + q] = key;
        }
        if (HAS_VAL) {
            __syncthreads();
            u32* svals = reinterpret_cast<u32*>(skeys);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) svals[rk[i]] = v[i];
            prefetch_vals(tile + 256);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 q = j * WG + t;
                const u32 d = (dd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                if (FULL || q < nvalid) vout[adj[d] + q] = svals[q];
            }
        }
        (void)tbase;
    };
    for (u32 tile = first; tile < num_tiles; tile += 256) {
        const u32 left = n - tile * (u32)TILE;
        if (left >= (u32)TILE) do_tile(tile, (u32)TILE, std::true_type());
        else do_tile(tile, left, std::false_type());
    }
}

// ---------------------------------------------------------------------------------------------
// rs_scatter_wc: the digit pass with write combining (large inputs).
//
// What limits rs_scatter above on uniform digits is not bytes but write *requests*: a tile leaves 256 runs of ~32
// records, i.e. ~256 B of keys and ~128 B of values at arbitrary 8-/4-byte alignment, so every run touches two
// partially written 128-B lines per array (tools/ubench.hip: pairs in aligned full lines 5.1 TB/s, the same bytes as
// misaligned 16-record runs 2.4 TB/s, aligned 128-B key lines with 64-B value half-lines 3.8-4.1 TB/s; HBM byte counters
// stay at the algorithmic volume in all cases).  Here a workgroup keeps, per digit, the records that do not yet fill a
// line: keys are written only as whole 16-key groups on 16-key boundaries of the output (one full 128-B line), values
// only as whole 32-value groups on 32-value boundaries (one full line); the unaligned head of a (chunk, digit) segment
// and its tail are written once each.  Output is identical to rs_scatter (same stable order).
// Measured (same box, 64 Mi pairs): BWT first-sort pass 0.370 ms against 0.389 ms for rs_scatter (the two lowest digits
// 0.47 -> 0.40 and 0.42 -> 0.38, the other six tie), uniform digits 0.403 against 0.414, text-skewed 0.307 against 0.284.
// A later version that merges the pending records in front of the new ones, so that every line leaves in ONE instruction
// (what tools/ubench_wc.hip says the memory system wants), is VALU-bound and no faster; it was removed in round 5 (history: NOTES_r01-r04.md).
//
// Shape: the 1024 x 8 shape of rs_scatter (8192-record tiles, one workgroup per CU walking four of rs_hist's chunks), so
// that the fixed per-tile work (digit scan, flushes, barriers) is spread over 8 records per lane; the first version of
// this kernel (4096-record tiles, 32-record groups for both arrays = 96 KB of pending records) was VALU-issue-bound.
// Pending records live in LDS rings indexed by the OUTPUT position (slot = position mod group size): a record never
// moves between becoming pending and being written.  State per digit (one thread each, in registers): E = output
// position behind the last record seen, gk / gv = first key / value position not yet written (gk = max(segment start,
// E rounded down to 16), gv likewise with 32).
// ---------------------------------------------------------------------------------------------
constexpr int WC_WG = 1024, WC_WAVES = WC_WG / 64, WC_ITEMS = 8, WC_TILE = WC_WG * WC_ITEMS, WC_GK = 16, WC_GV = 32, WC_SPAN = 4;
constexpr int WC_LDS_KEYS  = WC_TILE * 8 + 256 * WC_GK * 8 + WC_WAVES * 256 * 4 + 4 * 256 * 4 + 16 * 4;
constexpr int WC_LDS_PAIRS = WC_LDS_KEYS + 256 * WC_GV * 4 + 3 * 256 * 4;
static_assert(WC_LDS_PAIRS <= 160 * 1024, "rs_scatter_wc does not fit the CU's LDS");

template <bool HAS_VAL>
__global__ __launch_bounds__(WC_WG) void rs_scatter_wc_kernel(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                              const u32* __restrict__ vin, u32* __restrict__ vout,
                                                              u32 n, int shift, u32 mask,
                                                              u32 chunk_tiles, u32 num_chunks,
                                                              const u32* __restrict__ offsets,
                                                              const u32* __restrict__ rowtot, u64* __restrict__ tdbg)
{
    (void)tdbg;
    constexpr int WG = WC_WG, WAVES = WC_WAVES, ITEMS = WC_ITEMS, TILE = WC_TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* skeys = reinterpret_cast<u64*>(smem);                        // [TILE] staging (reused as u32 for values)
    u64* pendK = skeys + TILE;                                        // [256][16] pending keys, ring by output position
    u32* whist = reinterpret_cast<u32*>(pendK + 256 * WC_GK);         // [16][256] per-wave digit counts / prefixes
    u32* infoA = whist + WAVES * 256;                                 // [256] staging slot q of digit d is output position q + A
    u32* infoK = infoA + 256;                                         // [256] (first slot that stays pending) | (end of valid slots) << 16, keys
    u32* fkS   = infoK + 256;                                         // [256] old pending keys to write this tile: positions [fkS, fkE)
    u32* fkE   = fkS + 256;
    u32* scr   = fkE + 256;                                           // [16]
    u32* pendV = scr + 16;                                            // [256][32] pending values (pairs only)
    u32* infoV = pendV + 256 * WC_GV;                                 // [256] as infoK, values
    u32* fvS   = infoV + 256;
    u32* fvE   = fvS + 256;
    lds_vu32* vwh = (lds_vu32*)whist;

    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;

    u32 E = 0, gk = 0, gv = 0;              // digit state of thread t < 256
    {
        u32 tot;
        const u32 base = rs_digit_excl_sum<WAVES>(t < 256 ? rowtot[t] : 0u, scr, &tot);
        if (t < 256) { E = base + offsets[(size_t)t * num_chunks + (size_t)blockIdx.x * WC_SPAN]; gk = gv = E; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;          // every wave owns (and re-zeroes) its 256 counters
    __syncthreads();

    const u32 rec0 = (u32)((u64)blockIdx.x * WC_SPAN * chunk_tiles * RS_TILE);
    u32 rec1;
    { const u64 e = (u64)rec0 + (u64)WC_SPAN * chunk_tiles * RS_TILE; rec1 = e > n ? n : (u32)e; }

    // as in the 1024 x 8 shape of rs_scatter: the next tile's keys / values are requested as soon as this tile's sit in the
    // staging area, ahead of the tile's stores; a lane past the end reads the workgroup's first record instead
    const u32 wbase = w * (64 * ITEMS) + lane;
    u64 k[ITEMS];
    u32 v[ITEMS];
    auto prefetch_keys = [&](const u32 tb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = e < rec1 ? e : rec0; k[i] = __builtin_nontemporal_load(&kin[e]); }
    };
    auto prefetch_vals = [&](const u32 tb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = e < rec1 ? e : rec0; v[i] = __builtin_nontemporal_load(&vin[e]); }
    };
    prefetch_keys(rec0);
    if (HAS_VAL) prefetch_vals(rec0);

    // old pending records of the digits that reached a group boundary: 16 lanes per digit (keys), 32 per digit (values)
    auto flush_pending = [&]() __attribute__((always_inline)) {
#pragma unroll 2
        for (int s = 0; s < 256 / (WG / WC_GK); ++s) {
            const u32 b = s * (WG / WC_GK) + (t >> 4);
            const u32 pos = fkS[b] + (t & (WC_GK - 1));
            if (pos < fkE[b]) kout[pos] = pendK[b * WC_GK + (pos & (WC_GK - 1))];
        }
        if (HAS_VAL) {
#pragma unroll 2
            for (int s = 0; s < 256 / (WG / WC_GV); ++s) {
                const u32 b = s * (WG / WC_GV) + (t >> 5);
                const u32 pos = fvS[b] + (t & (WC_GV - 1));
                if (pos < fvE[b]) vout[pos] = pendV[b * WC_GV + (pos & (WC_GV - 1))];
            }
        }
    };

    auto do_tile = [&](const u32 tbase, const u32 nvalid, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const u32 tile_no = (tbase - rec0) / TILE; (void)tile_no;
        RS_PH(0);
        if (!FULL) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) if (wbase + i * 64 >= nvalid) k[i] = ~0ull;       // padding sorts last
        }
        RS_PH(1);

        // ---- stable in-wave ranking by ballot match (rs_rank_wave) --------------------------------
        u32 rk[ITEMS];
        rs_rank_wave<ITEMS>(k, shift, mask, vwh + w * 256, rk);
        RS_PH(2);
        __syncthreads();
        RS_PH(3);

        // ---- per digit: wave prefixes, tile-local start, what is written / kept this tile ----------
        {
            u32 c[WAVES];
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { c[i] = whist[i * 256 + t]; tot += c[i]; }
            }
            u32 all;
            const u32 ds = rs_digit_excl_sum<WAVES, false>(tot, scr, &all);
            if (t < 256) {
                u32 run = ds;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { whist[i * 256 + t] = run; run += c[i]; }
                const u32 cv = tot - ((!FULL && t == mask) ? ((u32)TILE - nvalid) : 0u);      // padding records sort last
                const u32 E0 = E, E1 = E0 + cv;
                const u32 ak = E1 & ~(u32)(WC_GK - 1), av = E1 & ~(u32)(WC_GV - 1);
                const u32 Fk = ak > gk ? ak : gk, Fv = av > gv ? av : gv;          // first positions still unwritten after this tile
                infoA[t] = E0 - ds;
                infoK[t] = (ds + (Fk > E0 ? Fk - E0 : 0u)) | ((ds + cv) << 16);
                fkS[t] = gk; fkE[t] = (Fk > gk) ? E0 : gk;                         // a boundary was reached: everything pending leaves
                if (HAS_VAL) {
                    infoV[t] = (ds + (Fv > E0 ? Fv - E0 : 0u)) | ((ds + cv) << 16);
                    fvS[t] = gv; fvE[t] = (Fv > gv) ? E0 : gv;
                }
                gk = Fk; gv = Fv; E = E1;
            }
        }
        __syncthreads();
        RS_PH(4);

        // ---- local reorder of the keys; old pending records of the digits that flush leave now -------
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u32 d = (u32)(k[i] >> shift) & mask;
            const u32 pos = whist[w * 256 + d] + rk[i];
            rk[i] = pos;
            skeys[pos] = k[i];
        }
        prefetch_keys(tbase + TILE);
        flush_pending();
        RS_PH(5);
        __syncthreads();
        RS_PH(6);
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;       // for the next tile's ranking (own wave's counters only)

        u32 dd[ITEMS / 4];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 q = j * WG + t;
            const u64 key = skeys[q];
            const u32 d = (u32)(key >> shift) & mask;
            if ((j & 3) == 0) dd[j >> 2] = d; else dd[j >> 2] |= d << (8 * (j & 3));
            const u32 B = infoK[d], pos = q + infoA[d];
            if (q < (B & 0xffffu)) kout[pos] = key;
            else if (q < (B >> 16)) pendK[d * WC_GK + (pos & (WC_GK - 1))] = key;
        }
        RS_PH(7);

        if (HAS_VAL) {
            __syncthreads();
            u32* svals = reinterpret_cast<u32*>(skeys);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) svals[rk[i]] = v[i];
            prefetch_vals(tbase + TILE);
            __syncthreads();
            RS_PH(8);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 q = j * WG + t;
                const u32 d = (dd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const u32 B = infoV[d], pos = q + infoA[d];
                const u32 val = svals[q];
                if (q < (B & 0xffffu)) vout[pos] = val;
                else if (q < (B >> 16)) pendV[d * WC_GV + (pos & (WC_GV - 1))] = val;
            }
        }
        RS_PH(9);
        RS_PH(10);            // no barrier here: the next tile's first writes (own counters; tables behind its first barrier) hurt nobody
    };
    u32 tbase = rec0;
    for (; tbase + TILE <= rec1; tbase += TILE) do_tile(tbase, (u32)TILE, std::true_type());
    if (tbase < rec1) do_tile(tbase, rec1 - tbase, std::false_type());

    // tails of this workgroup's segments
    __syncthreads();
    if (t < 256) { fkS[t] = gk; fkE[t] = E; if (HAS_VAL) { fvS[t] = gv; fvE[t] = E; } }
    __syncthreads();
    flush_pending();
}

// ---------------------------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------------------------
// Per-context (= per-device) setup, called from bscgpu_create with the context's device current: the dynamic-LDS limits
// of the large-tile kernels are a per-device attribute of the function, so every device that gets a context must be told
// (a process-wide `static` here would configure only the first device and race between threads).
// BSC_RS_WC: see radix_sort_passes.
int radix_engine_setup(bscgpu_ctx* c)
{
    const char* e = getenv("BSC_RS_WC");
    int mode = e ? atoi(e) : RS_WC_DEFAULT;
    if (hipFuncSetAttribute((const void*)rs_scatter_wc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WC_LDS_PAIRS) != hipSuccess ||
        hipFuncSetAttribute((const void*)rs_scatter_wc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WC_LDS_KEYS) != hipSuccess) {
        mode = 0;
        (void)hipGetLastError();               // do not leave a sticky error behind
    }
    HIP_TRY(c, hipFuncSetAttribute((const void*)rs_scatter_kernel<true, RSB_WG, RSB_ITEMS, RSB_SPAN, false, RSB_PIPE != 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RSB_LDS));
    c->rs_wc_mode = mode;
    return radix_onesweep_setup(c);
}

int radix_sort_passes(bscgpu_ctx* c, u64* keys, u64* keys_alt, u32* vals, u32* vals_alt, u64 n,
                      const RadixPass* passes, int npasses, int* in_alt, u32* emit_pos)
{
    if (emit_pos != nullptr && (npasses != 1 || vals != nullptr)) return BSC_BAD_PARAMETER;     // inverse of ONE keys-only pass
    *in_alt = 0;
    if (n == 0 || npasses == 0) return BSC_NO_ERROR;
    if (n >= 0xfff00000ull) return BSC_BAD_PARAMETER;          // record indexes are u32, with headroom for a tile past the end
    if ((((uintptr_t)keys) | ((uintptr_t)keys_alt)) & 15) return ctx_fail(c, BSC_BAD_PARAMETER, "radix keys not 16B aligned", hipSuccess);

    const Chunking ch = rs_chunking(n);
    const bool has_val = (vals != nullptr);
    const int wc_mode = c->rs_wc_mode;          // BSC_RS_WC, read once per context (radix_engine_setup)
    // large (key, value) sorts: one histogram read per sort + single-read digit passes (radix_onesweep.hip; BSC_RS_ONESWEEP)
    if (emit_pos == nullptr && wc_mode != 2 && radix_onesweep_wanted(c, n, npasses, has_val)) {
        const int rc = radix_onesweep_sort(c, keys, keys_alt, vals, vals_alt, n, passes, npasses);
        if (rc == BSC_NO_ERROR) *in_alt = (npasses & 1);
        return rc;
    }
    const bool big_pairs = RS_BIG_PAIRS && ch.num_chunks >= 512 && ch.chunk_tiles >= 2;   // enough records for 8192-record tiles on every CU
    // (with BSC_RS_ORDER=0) BSC_RS_WC: 0 = never, 1 (default) = large (key, value) passes — on the BWT's keys 0.370 ms per pass against 0.389 ms, the gain
    // sits in the two lowest digits (0.47 -> 0.40, 0.42 -> 0.38), the others tie; keys-only passes stay on the plain kernel (text-
    // skewed ST digits: 0.189 against 0.221 ms) —, 2 = every pass with >= 4 chunks (tests)
    const bool use_wc = emit_pos == nullptr && ((wc_mode == 2 && ch.num_chunks >= 4) || (wc_mode == 1 && has_val && ch.num_chunks >= 512 && ch.chunk_tiles >= 2));
    u64 *ksrc = keys, *kdst = keys_alt;
    u32 *vsrc = vals, *vdst = vals_alt;
    const u64 rec_bytes = 8 + (has_val ? 4 : 0);
    // XCD-interleaved tiles for large (key, value) passes (rs_scatter_tiled_kernel): BSC_RS_ORDER=1 (default); 0 = one contiguous
    // range per workgroup (the write-combining / plain kernels below); 2 = also for large keys-only passes (A/B)
    static const int order_mode = [] { const char* e = getenv("BSC_RS_ORDER"); return e ? atoi(e) : 1; }();
    const u32 num_tiles8k = (u32)((n + 8191) / 8192);
    const bool tiled = emit_pos == nullptr && big_pairs && wc_mode != 2 && ((order_mode >= 1 && has_val) || order_mode == 2);
    if (tiled && c->tile_counts_cap < (size_t)256 * num_tiles8k) {
        if (c->tile_counts) (void)hipFree(c->tile_counts);
        c->tile_counts = nullptr; c->tile_counts_cap = 0;
        const size_t want = (size_t)256 * ((size_t)(c->max_n > (int64_t)n ? c->max_n : (int64_t)n) / 8192 + 2);
        if (hipMalloc((void**)&c->tile_counts, want * 4) != hipSuccess) return ctx_fail(c, BSC_GPU_NOT_ENOUGH_MEMORY, "tile count table", hipSuccess);
        c->tile_counts_cap = want;
        HIP_TRY(c, hipFuncSetAttribute((const void*)rs_scatter_tiled_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RST_LDS));
        HIP_TRY(c, hipFuncSetAttribute((const void*)rs_scatter_tiled_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RST_LDS));
    }

    // passes that also emit the permutation serve the device coder / the inverse BWT: booked apart from the graded digit passes
    const int kind_hist = emit_pos ? BSCGPU_K_RADIX_AUX : BSCGPU_K_RADIX_HIST, kind_scan = emit_pos ? BSCGPU_K_RADIX_AUX : BSCGPU_K_RADIX_SCAN,
              kind_scatter = emit_pos ? BSCGPU_K_RADIX_AUX : BSCGPU_K_RADIX_SCATTER;
    for (int p = 0; p < npasses; ++p) {
        const int shift = passes[p].shift;
        const u32 mask  = (passes[p].bits >= 8) ? 0xffu : ((1u << passes[p].bits) - 1u);

        if (tiled) {
            prof_begin(c, BSCGPU_K_RADIX_HIST, n * 8, n);
            hipLaunchKernelGGL(rs_hist_kernel, dim3(num_tiles8k), dim3(RS_WG), 0, c->stream,
                               ksrc, (u32)n, shift, mask, 2u, num_tiles8k, c->tile_counts);
            prof_end(c);
            prof_begin(c, BSCGPU_K_RADIX_SCAN, (u64)256 * num_tiles8k * 8, 0);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(WG), 0, c->stream, c->tile_counts, num_tiles8k, c->rowtot);
            prof_end(c);
            prof_begin(c, BSCGPU_K_RADIX_SCATTER, 2 * n * rec_bytes, n);
            if (has_val)
                hipLaunchKernelGGL(rs_scatter_tiled_kernel<true>, dim3(256), dim3(RSB_WG), RST_LDS, c->stream,
                                   ksrc, kdst, vsrc, vdst, (u32)n, shift, mask, num_tiles8k, c->tile_counts, c->rowtot);
            else
                hipLaunchKernelGGL(rs_scatter_tiled_kernel<false>, dim3(256), dim3(RSB_WG), RST_LDS, c->stream,
                                   ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, shift, mask, num_tiles8k, c->tile_counts, c->rowtot);
            prof_end(c);
            HIP_TRY(c, hipGetLastError());
            u64* tk = ksrc; ksrc = kdst; kdst = tk;
            u32* tv = vsrc; vsrc = vdst; vdst = tv;
            continue;
        }
        prof_begin(c, kind_hist, n * 8, n);
        hipLaunchKernelGGL(rs_hist_kernel, dim3(ch.num_chunks), dim3(RS_WG), 0, c->stream,
                           ksrc, (u32)n, shift, mask, ch.chunk_tiles, ch.num_chunks, c->counts);
        prof_end(c);

        prof_begin(c, kind_scan, (u64)256 * ch.num_chunks * 8, 0);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(WG), 0, c->stream, c->counts, ch.num_chunks, c->rowtot);
        prof_end(c);

        prof_begin(c, kind_scatter, 2 * n * rec_bytes + (emit_pos ? 4 * n : 0), n);
        if (use_wc) {
            const u32 grid = (ch.num_chunks + WC_SPAN - 1) / WC_SPAN;
            if (has_val)
                hipLaunchKernelGGL(rs_scatter_wc_kernel<true>, dim3(grid), dim3(WC_WG), WC_LDS_PAIRS, c->stream,
                                   ksrc, kdst, vsrc, vdst, (u32)n, shift, mask, ch.chunk_tiles, ch.num_chunks,
                                   c->counts, c->rowtot, c->wc_sink);
            else
                hipLaunchKernelGGL(rs_scatter_wc_kernel<false>, dim3(grid), dim3(WC_WG), WC_LDS_KEYS, c->stream,
                                   ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, shift, mask,
                                   ch.chunk_tiles, ch.num_chunks, c->counts, c->rowtot, c->wc_sink);
        } else if (has_val && big_pairs) {
            hipLaunchKernelGGL((rs_scatter_kernel<true, RSB_WG, RSB_ITEMS, RSB_SPAN, false, RSB_PIPE != 0>), dim3((ch.num_chunks + RSB_SPAN - 1) / RSB_SPAN), dim3(RSB_WG), RSB_LDS, c->stream,
                               ksrc, kdst, vsrc, vdst, (u32)n, shift, mask, ch.chunk_tiles, ch.num_chunks,
                               ch.num_tiles, c->counts, c->rowtot, c->wc_sink);
        } else if (has_val)
            hipLaunchKernelGGL((rs_scatter_kernel<true, RS_WG, RS_ITEMS, 1>), dim3(ch.num_chunks), dim3(RS_WG), RS_LDS, c->stream,
                               ksrc, kdst, vsrc, vdst, (u32)n, shift, mask, ch.chunk_tiles, ch.num_chunks,
                               ch.num_tiles, c->counts, c->rowtot, c->wc_sink);
        else if (emit_pos != nullptr)
            hipLaunchKernelGGL((rs_scatter_kernel<false, RS_WG, RS_ITEMS, 1, true>), dim3(ch.num_chunks), dim3(RS_WG), RS_LDS, c->stream,
                               ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, shift, mask,
                               ch.chunk_tiles, ch.num_chunks, ch.num_tiles, c->counts, c->rowtot, c->wc_sink, emit_pos);
        else
            hipLaunchKernelGGL((rs_scatter_kernel<false, RS_WG, RS_ITEMS, 1>), dim3(ch.num_chunks), dim3(RS_WG), RS_LDS, c->stream,
                               ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, shift, mask,
                               ch.chunk_tiles, ch.num_chunks, ch.num_tiles, c->counts, c->rowtot, c->wc_sink, (u32*)nullptr);
        prof_end(c);
        HIP_TRY(c, hipGetLastError());

        u64* tk = ksrc; ksrc = kdst; kdst = tk;
        u32* tv = vsrc; vsrc = vdst; vdst = tv;
    }
#if RS_PHASE_TIMING
    if (n >= (1u << 24)) {      // debug builds: phase stamps of the last pass
        static std::vector<u64> host(256 * 32 * 16);
        if (hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(host.data(), c->wc_sink, host.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen("gpurun_out/phase_timing.bin", "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
        }
    }
#endif
    *in_alt = (npasses & 1);
    return BSC_NO_ERROR;
}
