// radix_onesweep.hip — the single-read LSD digit pass for large (u64 key, u32 value) sorts on gfx950 (MI355X).
//
// Role: cub::DeviceRadixSort::SortPairs as libcubwt calls it once per suffix sort (libcubwt.cu:718) — one histogram read per
// SORT, then one kernel per digit that reads every record once and writes it once (B_sort = m*kb + P*2*m*(kb+vb)).
// The three-kernel pass of radix_sort.hip re-reads all keys in rs_hist for every digit (+33 % traffic, 0.13 of 0.47 ms).
//
//   rs_hist_all      one streaming read of the keys, the global 256-bin histogram of every digit of the sort (<= 8).
//   rs_onesweep      persistent workgroups (one per CU, 16 waves).  FIFTEEN waves stream 7680-record tiles (the tile loop of
//                    rs_scatter_tiled: wave-striped loads, ballot-match ranking, tile-local reorder through LDS, every digit
//                    leaves as one run) and publish each tile's digit counts right after the digit scan; the SIXTEENTH wave —
//                    the scout — moves no records: it draws the tickets and collects the counts of all earlier tiles, i.e. the
//                    tile's output offsets, which a three-kernel pass gets from rs_hist + rs_scan.
//      order     tile = ticket (one returning atomic per tile): tiles are claimed in index order by workgroups that are RUNNING,
//                so every tile a wait can depend on has an owner that is executing and does not itself wait for a later tile —
//                the pass cannot deadlock under partial residency (several contexts share a GPU in bench.py; a static
//                assignment of tiles to workgroups could).  Claim order = index order also means a tile's predecessors published
//                before it did.  (Batches of 32 tiles owned by one XCD, for L2 write merging as in rs_scatter_tiled, were built and
//                measured: the batch before a tile is then claimed AFTER it by another XCD, half of the tiles wait for it.)
//      rows      a tile publishes ONE 512-byte row: 64 granules of 8 bytes, four 16-bit fields {digit count : 14, two bits of the launch
//                tag : 2}, each granule written by one lane with one agent-scope store — a granule validates itself, no fence, no
//                flag (MI355X_MICROARCH.md, hand-off table: data-tagged granules) —, and adds its counts to the row of its GROUP (8
//                tiles) and of its BATCH (8 groups) with agent-scope 64-bit atomics on {arrivals : 8, two 28-bit sums} words; such a
//                row is complete when arrivals == 8 / 64.  offsets(tile) = digit base + the complete batch rows below its batch (a
//                workgroup keeps a running sum; its next tile is ~256 tiles = 4-5 batches on) + the group rows of its batch below
//                its group (<= 7) + the tile rows of its group below it (<= 7): ~20 loads per lane.  Two levels with 32-tile batches
//                were measured first: 8-9 new batch rows per tile, and every row beyond the eight requested in bulk cost two
//                serial round trips of ~4 us.
//      latency   hidden by software pipelining, not avoided.  Three tiles are in flight per workgroup: tile i+2 is ranked, published and
//                staged into an LDS buffer; tile i+1 waits in the other buffer while the scout collects its offsets — its look-back
//                loads are issued at the top of the iteration, about half a tile time after the tile (and hence every tile claimed
//                before it) was published, and are looked at in front of the iteration's last barrier, a whole tile time (~8 us)
//                later; tile i is written out first thing in the iteration, at offsets that have been ready since the previous
//                one, and leaves its buffer to tile i+2.  Under streaming load an agent-scope load takes ~3 us on this chip: with
//                the offsets needed in the same iteration the scout sat on the critical path (0.39 ms per pass).  Rows that are
//                still missing when the scout looks are asked for again ALL AT ONCE (one round trip per retry, not one per row).
//                Four tiles in flight (11 streaming waves x 5632-record tiles, three staging buffers) took the scout off the
//                critical path entirely and were slower: smaller tiles write shorter runs (DESIGN.md 3.1 has the numbers).
//      the scout exists because vmcnt is an in-order counter: look-back loads issued by a streaming wave sit behind that wave's
//                own stores of the previous tile, and using them means waiting for those stores to drain (measured: 0.12 ms per
//                pass although the rows themselves were there).  The scout's queue holds protocol traffic only.  The two roles are
//                separate loops behind a scalar branch (own register allocation each) that execute the same barrier sequence.
//   Polls are bounded.  A scout that gives up (a predecessor's row did not arrive within ~16 K polls: wave pre-emption, a debugger, a
//   hogged CU) raises the context's STICKY error word and points its tile's stores at the tile's own place in the output (in bounds,
//   wrong order) instead of at offsets built from incomplete sums, which could lie up to 7 x 7680 records beyond the true ones — past
//   the output buffers.  Workgroups that see the word raised stop claiming tiles, later passes of the sort return at once, and the
//   host redoes the whole sort through the three-kernel passes (bwt_device / st_device; radix_onesweep_check).
// Stable: output order inside a digit = tile order, then the tile-local stable rank (rs_rank_wave), exactly as rs_scatter.
#include "dev_common.h"
#include "radix_dev.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <atomic>

constexpr int OS_WG = 1024, OS_WAVES = OS_WG / 64, OS_SW = OS_WAVES - 1 /* streaming waves */, OS_ST = OS_SW * 64 /* streaming threads */;
#ifndef OS_ITEMS_N
#define OS_ITEMS_N 8
#endif
constexpr int OS_ITEMS = OS_ITEMS_N, OS_TILE = OS_ST * OS_ITEMS /* 7680 */, OS_GRP = 8 /* tiles per group */, OS_GPB = 8 /* groups per batch */,
              OS_BATCH = OS_GRP * OS_GPB /* 64 tiles */, OS_MAXP = 8;
constexpr u32 OS_NONE = 0xffffffffu;
constexpr u32 OS_SPIN_LIMIT = 1u << 18;
// LDS: two key staging buffers, per-wave digit counters, five 256-entry tables, scratch
constexpr int OS_LDS = 2 * OS_TILE * 8 + OS_SW * 256 * 4 + 5 * 256 * 4 + 32 * 4;
static_assert(OS_TILE < (1 << 14), "a tile row stores 14-bit digit counts");
static_assert((OS_GRP - 1) * OS_TILE < (1 << 16), "the tile rows of a group are summed as packed 16-bit halves");
static_assert(OS_BATCH * OS_TILE < (1 << 28), "group / batch rows carry 28-bit sums");
static_assert(OS_LDS <= 160 * 1024, "one workgroup per CU: the staging buffers must fit the CU's LDS");
static_assert(OS_LDS <= 160 * 1024, "rs_onesweep does not fit the CU's LDS");
// per-pass control block (u32 words): [0] next ticket (the error word lives in the context's scalar area, dscal[OS_ERR_SLOT])
constexpr int OS_CTL_WORDS = 32;

struct OsPasses { int np; int shift[OS_MAXP]; u32 mask[OS_MAXP]; };

// Debug builds (tools/build_variant.sh): -DOS_PHASE_TIMING=1 stamps s_memtime at the phase boundaries of the first 40 tiles of every
// workgroup (thread 0 and the scout's lane 0) into the context's scratch buffer; -DOS_ABL=bits removes parts of the protocol for
// timing only (1: no look-back loads / polls, 2: no publishing) — results are wrong with any bit set.
#ifndef OS_PHASE_TIMING
#define OS_PHASE_TIMING 0
#endif
#ifndef OS_ABL
#define OS_ABL 0
#endif
#if OS_PHASE_TIMING
#define OS_PH(i) do { if ((t == 0 || t == (u32)OS_ST) && tile_no < 40u) tdbg[(((size_t)blockIdx.x * 40 + tile_no) * 2 + (t ? 1 : 0)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OS_PH(i) do { } while (0)
#endif

#define OS_LOAD(p)      __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OS_STORE(p, v)  __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#ifndef OS_STATIC_ORDER
#define OS_STATIC_ORDER 0
#endif
#define OS_ADD(p, v)    __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// The ablation / static-order switches produce WRONG results by design (timing only): such an object must never end up in the shipped
// library.  tools/build_variant.sh defines BSC_EXPERIMENT_BUILD for its side builds; libbsc_amd/build.py never does.
#if (OS_ABL || OS_STATIC_ORDER) && !defined(BSC_EXPERIMENT_BUILD)
#error "OS_ABL / OS_STATIC_ORDER are timing experiments with wrong results: build them with tools/build_variant.sh (-DBSC_EXPERIMENT_BUILD), never into the product"
#endif

// ---------------------------------------------------------------------------------------------
// rs_hist_all: totals[p][d] += number of keys whose digit p equals d, for every pass of the sort, in one read of the keys.
// 16 replicas of the np x 256 counters per workgroup (selected by lane and wave bits: text digits are skewed).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OS_WG) void rs_hist_all_kernel(const u64* __restrict__ keys, u32 n, OsPasses P,
                                                           u32* __restrict__ zero_base, u32 pass_stride_words)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32* h = reinterpret_cast<u32*>(smem);                     // [16][np][256]
    const u32 t = threadIdx.x, w = t >> 6;
    const u32 np = (u32)P.np;
    for (u32 i = t; i < 16u * np * 256u; i += OS_WG) h[i] = 0;
    __syncthreads();
    u32* hr = h + (((t & 3u) | ((w & 3u) << 2)) * np) * 256u;
    auto count = [&](const u64 key) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < OS_MAXP; ++p)
            if ((u32)p < np) atomicAdd(&hr[p * 256 + ((u32)(key >> P.shift[p]) & P.mask[p])], 1u);
    };
    constexpr u64 HA_TILE = (u64)OS_WG * 8;                    // 8 keys per thread and round
    const u64 stride = (u64)gridDim.x * HA_TILE;
    for (u64 base = (u64)blockIdx.x * HA_TILE; base < n; base += stride) {
        const u64 i = base + 2 * t;
        if (base + HA_TILE <= n) {
            ulonglong2 a, b, c, d;
            a.x = __builtin_nontemporal_load(keys + i);             a.y = __builtin_nontemporal_load(keys + i + 1);
            b.x = __builtin_nontemporal_load(keys + i + 2 * OS_WG); b.y = __builtin_nontemporal_load(keys + i + 2 * OS_WG + 1);
            c.x = __builtin_nontemporal_load(keys + i + 4 * OS_WG); c.y = __builtin_nontemporal_load(keys + i + 4 * OS_WG + 1);
            d.x = __builtin_nontemporal_load(keys + i + 6 * OS_WG); d.y = __builtin_nontemporal_load(keys + i + 6 * OS_WG + 1);
            count(a.x); count(a.y); count(b.x); count(b.y); count(c.x); count(c.y); count(d.x); count(d.y);
        } else {
            for (u64 e = base + t; e < n; e += OS_WG) count(keys[e]);
        }
    }
    __syncthreads();
    for (u32 i = t; i < np * 256u; i += OS_WG) {
        u32 sum = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += h[(u32)r * np * 256u + i];
        if (sum) atomicAdd(&zero_base[(size_t)(i >> 8) * pass_stride_words + OS_CTL_WORDS + (i & 255u)], sum);
    }
}

// ---------------------------------------------------------------------------------------------
// rs_onesweep: one digit pass, records read once and written once.
// ---------------------------------------------------------------------------------------------
// granule of a tile row: four 16-bit fields {count : 14 (<= 7680), two bits of the launch tag : 2} for digits 4l .. 4l+3 — a tag byte
//                        spread over the four fields, so that eight rows can be summed as packed 16-bit halves by plain 32-bit adds
//                        (8 x 7680 < 2^16) after masking the tag bits off;
// word of a batch row:   arrivals << 56 | sum_hi << 28 | sum_lo           (two words per lane: digits 4l, 4l+1 and 4l+2, 4l+3)
__host__ __device__ constexpr u64 os_tag_pattern(u32 tag8) {
    return ((u64)(tag8 & 3u) << 14) | ((u64)((tag8 >> 2) & 3u) << 30) | ((u64)((tag8 >> 4) & 3u) << 46) | ((u64)((tag8 >> 6) & 3u) << 62);
}

template <bool HAS_VAL>
__global__ __launch_bounds__(OS_WG) void rs_onesweep_kernel(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                            const u32* __restrict__ vin, u32* __restrict__ vout,
                                                            u32 n, int shift, u32 mask, u32 ntiles,
                                                            u32* ctl, u32* err, const u32* __restrict__ totals,
                                                            u64* bagg /*[batches][128]*/, u64* gagg /*[groups][128]*/, u64* agg /*[tiles][64]*/, u32 tag8, u64* tdbg)
{
    (void)tdbg;
    u32 tile_no = 0; (void)tile_no;
    constexpr int WAVES = OS_WAVES, SW = OS_SW, ST = OS_ST, ITEMS = OS_ITEMS, TILE = OS_TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* S      = reinterpret_cast<u64*>(smem);                         // [2][TILE] locally reordered keys (then values) of the two tiles in flight
    u32* whist  = reinterpret_cast<u32*>(smem + 2 * TILE * 8);          // [SW][256]
    u32* rbase  = whist + SW * 256;                                     // [256] first output position of every digit (prologue only)
    u32* adj    = rbase + 256;                                          // [2][256] per staging buffer: output position of slot q of digit d = adj[d] + q
    u32* dstart = adj + 512;                                            // [2][256] tile-local start of every digit, per staging buffer
    u32* scr    = dstart + 512;                                         // [16]
    u32* sclaim = scr + 16;                                             // [1] the ticket drawn at the top of the iteration
    lds_vu32* vwh = (lds_vu32*)whist;

    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;
    // The two roles are two separate loops behind a SCALAR branch (the wave number goes through readfirstlane), so neither role's
    // registers are live in the other's code; both execute the same sequence of workgroup barriers per iteration.
    const bool scout = (u32)__builtin_amdgcn_readfirstlane((int)w) == (u32)SW;
    const u64 tagpat = os_tag_pattern(tag8);
    u32 opaque0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));                   // a zero the compiler cannot see through: keeps the ticket atomic's
                                                                        // address "divergent", so the atomic optimiser does not rewrite it into
                                                                        // readfirstlane form and wait for it on the spot
    {
        u32 tot;
        const u32 base = rs_digit_excl_sum<WAVES, true, true>(t < 256 ? totals[t] : 0u, scr, &tot);
        if (t < 256) rbase[t] = base;
    }
    if (!scout) {
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;
    }
#if OS_STATIC_ORDER
    // EXPERIMENT (timing only, needs all workgroups resident): no tickets — workgroup b takes tiles k * 256 + map(b), an XCD (b % 8) working on
    // OS_STATIC_ORDER consecutive tiles at a time: what an XCD-aware claim order would be worth to the write pattern.
    const u32 os_map = ((blockIdx.x >> 3) / (u32)OS_STATIC_ORDER) * (8u * (u32)OS_STATIC_ORDER) + (blockIdx.x & 7u) * (u32)OS_STATIC_ORDER + ((blockIdx.x >> 3) % (u32)OS_STATIC_ORDER);
    u32 os_round = 1;
    if (t == (u32)ST) { const u32 tk = os_map + opaque0; sclaim[0] = tk < ntiles ? tk : OS_NONE; }
#else
    // (a launch that finds the error word raised — by an earlier pass of this sort, or by a workgroup of this launch — claims nothing:
    // its input is already unusable; the decision is taken by ONE lane, so it is uniform over the workgroup's barriers)
    if (t == (u32)ST) { const u32 e0 = OS_LOAD(err); const u32 tk = OS_ADD(ctl + opaque0, 1u); sclaim[0] = (tk < ntiles && e0 == 0u) ? tk : OS_NONE; }
#endif
    __syncthreads();
    // Three tiles are in flight per workgroup: t0 is written out (its offsets were collected during the previous iteration), t1 waits
    // in its staging buffer while the scout collects its offsets, t2 is ranked, published and staged into the buffer t0 leaves.
    u32 t0 = OS_NONE, t1 = OS_NONE, t2 = (u32)__builtin_amdgcn_readfirstlane((int)sclaim[0]);      // tile numbers are wave-uniform: scalar control flow
    if (t2 == OS_NONE) return;
    bool more = true;                                                   // tickets may still yield tiles
    u32 x = 0;                                                          // staging buffer of t0, and then of t2; t1 sits in x ^ 1
    __syncthreads();                                                    // everybody has read sclaim (and rbase is complete)

    if (scout) {
        // =========================================================================================================
        // The scout wave.  Lane l owns digits 4l .. 4l+3.
        // =========================================================================================================
        u32 gbase = 0;                                                  // batches [0, gbase) are in R
        u32 npolls = 0; (void)npolls;
        u32 R[4];                                                       // digit base + counts of all complete batches accounted so far
#pragma unroll
        for (int i = 0; i < 4; ++i) R[i] = rbase[4 * lane + i];
        // Rows are read with buffer loads (scalar base, 32-bit lane offset, sc1 = agent scope like the atomics that write them): one
        // 16-byte instruction per group / batch row, one 8-byte instruction per tile row — while fifteen waves stream through the
        // CU's memory pipeline, ISSUING an instruction costs the scout ~70 ns, so the number of instructions is what counts.
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        typedef u32 v2u __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t ragg = __builtin_amdgcn_make_buffer_rsrc((void*)agg, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rgag = __builtin_amdgcn_make_buffer_rsrc((void*)gagg, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rbag = __builtin_amdgcn_make_buffer_rsrc((void*)bagg, 0, 0x7fffffff, 0x00020000);
        constexpr int SC1 = 16;                                         // cache policy bit of the buffer builtins on gfx940+: sc1
        auto ld_tile = [&](const u32 row) __attribute__((always_inline)) -> v2u { return __builtin_amdgcn_raw_buffer_load_b64(ragg, (int)(row * 512u + lane * 8u), 0, SC1); };
        auto ld_grp  = [&](const u32 row) __attribute__((always_inline)) -> v4u { return __builtin_amdgcn_raw_buffer_load_b128(rgag, (int)(row * 1024u + lane * 16u), 0, SC1); };
        auto ld_bat  = [&](const u32 row) __attribute__((always_inline)) -> v4u { return __builtin_amdgcn_raw_buffer_load_b128(rbag, (int)(row * 1024u + lane * 16u), 0, SC1); };
        // {arrivals : 8, sum_hi : 28, sum_lo : 28} x 2 in 32-bit arithmetic
        auto row_ok  = [&](const v4u y, const u32 want) __attribute__((always_inline)) -> bool { return (y.y >> 24) == want && (y.w >> 24) == want; };
        auto add_row = [&](u32 (&acc)[4], const v4u y) __attribute__((always_inline)) {
            acc[0] += y.x & 0xfffffffu; acc[1] += __builtin_amdgcn_alignbit(y.y, y.x, 28) & 0xfffffffu;
            acc[2] += y.z & 0xfffffffu; acc[3] += __builtin_amdgcn_alignbit(y.w, y.z, 28) & 0xfffffffu;
        };
        const u32 tag_lo = (u32)tagpat, tag_hi = (u32)(tagpat >> 32);
        while (t0 != OS_NONE || t1 != OS_NONE || t2 != OS_NONE) {
            const bool v1 = t1 != OS_NONE, v2 = t2 != OS_NONE;
            // t1 = tile cj of group cg of batch cG
            const u32 cj = v1 ? (t1 & (u32)(OS_GRP - 1)) : 0u, cg = v1 ? ((t1 / (u32)OS_GRP) & (u32)(OS_GPB - 1)) : 0u, cG = v1 ? (t1 / (u32)OS_BATCH) : 0u;
            const u32 grp0 = v1 ? (t1 / (u32)OS_GRP - cg) : 0u;        // first group of its batch
            const u32 nb = (cG - gbase) < 6u ? (cG - gbase) : 6u;      // batch rows requested in bulk
            OS_PH(0);
            // Ticket for the tile after t2, and the look-back loads for t1 — only the rows that exist: the tile rows of its group below it
            // (<= 7), the group rows of its batch below its group (<= 7), the batch rows not yet in the running sum (a workgroup's
            // next tile is ~256 tiles = 4 or 5 batches further on; up to 6 in bulk, more one by one): ~11 instructions on average.
            // They are looked at behind barrier 3, when they have been in flight for most of the iteration.
            u32 ticket = 0;
#if OS_STATIC_ORDER
            if (more && lane == 0) ticket = os_round * gridDim.x + os_map;
            ++os_round;
#else
            if (more && lane == 0) ticket = OS_ADD(ctl + opaque0, 1u);        // ONE lane draws
#endif
            v2u a1[OS_GRP - 1]; v4u g1[OS_GPB - 1], b1[6];
            if (!(OS_ABL & 1)) {
#pragma unroll
                for (int q = 0; q < OS_GRP - 1; ++q) if ((u32)q < cj) a1[q] = ld_tile(t1 - cj + q);
#pragma unroll
                for (int q = 0; q < OS_GPB - 1; ++q) if ((u32)q < cg) g1[q] = ld_grp(grp0 + q);
#pragma unroll
                for (int q = 0; q < 6; ++q) if ((u32)q < nb) b1[q] = ld_bat(gbase + q);
            }
            OS_PH(1);
            if (HAS_VAL) {
                __syncthreads();                                                                  // B5
                __syncthreads();                                                                  // B6
            }
            __syncthreads();                                                                      // B1
            if (v2) __syncthreads();                                                              // B2 (the streaming waves' digit scan; they publish t2)
            if (lane == 0) sclaim[0] = (more && ticket < ntiles) ? ticket : OS_NONE;
            OS_PH(2);
            __syncthreads();                                                                      // B3
            OS_PH(3);
            const u32 nn = (u32)__builtin_amdgcn_readfirstlane((int)sclaim[0]);
            if (nn == OS_NONE) more = false;
            // offsets of t1 (it sits in staging buffer x ^ 1)
            bool ok = true;
            if (v1) {
                u32 sg[4] = {0, 0, 0, 0};                               // counts of the complete groups of t1's batch below its group
                u32 plo = 0, phi = 0;                                   // counts of the tiles of its group below it, packed 16-bit halves
                if (!(OS_ABL & 1)) {
                    // Rows that were not there yet (published less than a visibility latency before they were asked for) are asked for
                    // again ALL AT ONCE: one more round trip, however many they are.  Wave-uniform mask: bits 0.. tile rows, 8.. group
                    // rows, 16.. batch rows.
                    auto missing_rows = [&]() __attribute__((always_inline)) -> u32 {
                        u32 m = 0;
#pragma unroll
                        for (int q = 0; q < OS_GRP - 1; ++q) if ((u32)q < cj && __ballot((a1[q].x & 0xC000C000u) != tag_lo || (a1[q].y & 0xC000C000u) != tag_hi)) m |= 1u << q;
#pragma unroll
                        for (int q = 0; q < OS_GPB - 1; ++q) if ((u32)q < cg && __ballot(!row_ok(g1[q], (u32)OS_GRP))) m |= 1u << (8 + q);
#pragma unroll
                        for (int q = 0; q < 6; ++q) if ((u32)q < nb && __ballot(!row_ok(b1[q], (u32)OS_BATCH))) m |= 1u << (16 + q);
                        return m;
                    };
                    u32 missing = missing_rows(), tries = 0;
                    OS_PH(7);
                    while (missing != 0u && ok) {
                        __builtin_amdgcn_s_sleep(8);
#pragma unroll
                        for (int q = 0; q < OS_GRP - 1; ++q) if (missing & (1u << q)) a1[q] = ld_tile(t1 - cj + q);
#pragma unroll
                        for (int q = 0; q < OS_GPB - 1; ++q) if (missing & (1u << (8 + q))) g1[q] = ld_grp(grp0 + q);
#pragma unroll
                        for (int q = 0; q < 6; ++q) if (missing & (1u << (16 + q))) b1[q] = ld_bat(gbase + q);
                        missing = missing_rows();
                        if (++tries > (OS_SPIN_LIMIT >> 4) || ((tries & 63u) == 0u && OS_LOAD(err) != 0u)) ok = false;
                    }
#if OS_PHASE_TIMING
                    npolls += tries;
#endif
                    OS_PH(8);
#pragma unroll
                    for (int q = 0; q < 6; ++q) if ((u32)q < nb) add_row(R, b1[q]);
                    for (u32 gg = gbase + 6u; gg < cG; ++gg) {          // a workgroup that fell behind (or has just started)
                        v4u y = ld_bat(gg);
                        u32 spins = 0;
                        while (__ballot(!row_ok(y, (u32)OS_BATCH)) && ok) {
                            __builtin_amdgcn_s_sleep(8);
                            y = ld_bat(gg);
                            if (++spins > (OS_SPIN_LIMIT >> 4) || ((spins & 63u) == 0u && OS_LOAD(err) != 0u)) ok = false;
                        }
                        add_row(R, y);
                    }
#pragma unroll
                    for (int q = 0; q < OS_GPB - 1; ++q) if ((u32)q < cg) add_row(sg, g1[q]);
#pragma unroll
                    for (int q = 0; q < OS_GRP - 1; ++q) if ((u32)q < cj) { plo += a1[q].x & 0x3fff3fffu; phi += a1[q].y & 0x3fff3fffu; }     // 7 x 7680 < 2^16: no carry
                }
                gbase = cG;
                const uint4 d4 = *reinterpret_cast<const uint4*>(dstart + (x ^ 1u) * 256 + 4 * lane);
                uint4 o;
                o.x = R[0] + sg[0] + (plo & 0xffffu) - d4.x; o.y = R[1] + sg[1] + (plo >> 16) - d4.y;
                o.z = R[2] + sg[2] + (phi & 0xffffu) - d4.z; o.w = R[3] + sg[3] + (phi >> 16) - d4.w;
                // gave up: the sums are incomplete (a stale row may even have been added).  The tile is written over its own TILE
                // slots of the output — slot q of any digit goes to t1 * TILE + q — so no store can leave the buffers.
                if (!ok) { const u32 home = t1 * (u32)TILE; o.x = home; o.y = home; o.z = home; o.w = home; }
                *reinterpret_cast<uint4*>(adj + (x ^ 1u) * 256 + 4 * lane) = o;
            }
            if (!ok) (void)OS_ADD(err, 1u);
            OS_PH(4);
            __syncthreads();                                                                      // B4
            OS_PH(11);
#if OS_PHASE_TIMING
            if (t == (u32)ST && tile_no < 40u) { tdbg[(((size_t)blockIdx.x * 40 + tile_no) * 2 + 1) * 16 + 14] = ((u64)t1 << 32) | t2; tdbg[(((size_t)blockIdx.x * 40 + tile_no) * 2 + 1) * 16 + 15] = npolls; }
#endif
            ++tile_no;
            t0 = t1; t1 = t2; t2 = nn; x ^= 1u;
        }
        return;
    }

    // =============================================================================================================
    // The fifteen streaming waves.
    // =============================================================================================================
    const u32 wbase = w * (64 * ITEMS) + lane;
    u64 k[ITEMS];
    u32 v[ITEMS], rk[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { v[i] = 0; rk[i] = 0; }
    constexpr int HP = (ITEMS + 1) / 2;
    u32 pos0[HP], pos1[HP];                               // staging slots of t0's / t1's records, two 16-bit slots per word
    // loads never sit behind a branch: a missing tile or a lane past the end reads record 0 (one line for the whole wave)
    // (tile = OS_NONE wraps to record numbers >= n for every lane: TILE * 0xffffffff = -TILE)
    auto load_keys = [&](const u32 tile) __attribute__((always_inline)) {
        const u32 tb = tile * (u32)TILE;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = e < n ? e : 0u; k[i] = __builtin_nontemporal_load(&kin[e]); }
    };
    auto load_vals = [&](const u32 tile) __attribute__((always_inline)) {
        const u32 tb = tile * (u32)TILE;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { u32 e = tb + wbase + i * 64; e = e < n ? e : 0u; v[i] = __builtin_nontemporal_load(&vin[e]); }
    };
    load_keys(t2);
#pragma unroll
    for (int i = 0; i < HP; ++i) { pos0[i] = 0; pos1[i] = 0; }

    // One iteration: write t0 out at the offsets the scout found during the previous iteration, then rank t2 and publish it
    // and stage it into the buffer t0 has left.  FULL = t0 and t2 are full tiles (no guards around loads and stores: waits stay exact).
    auto iteration = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const bool v0 = FULL || t0 != OS_NONE, v2 = FULL || t2 != OS_NONE;
        u32 n0 = (u32)TILE, n2 = (u32)TILE;                            // valid records
        if (!FULL) {
            n0 = v0 ? ((n - t0 * (u32)TILE) < (u32)TILE ? (n - t0 * (u32)TILE) : (u32)TILE) : 0u;
            n2 = v2 ? ((n - t2 * (u32)TILE) < (u32)TILE ? (n - t2 * (u32)TILE) : (u32)TILE) : 0u;
        }
        u64* Sx = S + (size_t)x * TILE;
        const u32* adjx = adj + x * 256;
        u64 pubg = 0, puba = 0;                                         // data words of this iteration's publication (kept live to its end: on
                                                                        // gfx950 a store's data registers may be read late, and overwriting them
                                                                        // is preceded by a wait for the store itself)
        OS_PH(0);
        // ---- t0 leaves: every digit as one contiguous run, consecutive lanes -> consecutive addresses
        u32 dd[(ITEMS + 3) / 4];
#pragma unroll
        for (int j = 0; j < (ITEMS + 3) / 4; ++j) dd[j] = 0;
        if (v0) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 q = j * ST + t;
                const u64 key = Sx[q];
                const u32 d = (u32)(key >> shift) & mask;
                if ((j & 3) == 0) dd[j >> 2] = d; else dd[j >> 2] |= d << (8 * (j & 3));
                if (FULL || q < n0) kout[adjx[d] + q] = key;
            }
        }
        OS_PH(1);
        if (HAS_VAL) {
            __syncthreads();                                                                      // B5
            OS_PH(2);
            u32* svals = reinterpret_cast<u32*>(Sx);
            if (v0) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) svals[(pos0[i >> 1] >> (16 * (i & 1))) & 0xffffu] = v[i];
            }
            OS_PH(3);
            __syncthreads();                                                                      // B6
            OS_PH(4);
            if (v0) {
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const u32 q = j * ST + t;
                    const u32 d = (dd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    if (FULL || q < n0) vout[adjx[d] + q] = svals[q];
                }
            }
        }
        OS_PH(5);
        // ---- t2: rank inside the waves
        if (v2) {
            if (!FULL) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) if (wbase + i * 64 >= n2) k[i] = ~0ull;          // padding sorts last
            }
            rs_rank_wave<ITEMS>(k, shift, mask, vwh + w * 256, rk);
        }
        OS_PH(6);
        __syncthreads();                                                                          // B1
        OS_PH(7);
        if (HAS_VAL) load_vals(t1);                                     // for the next iteration's write-out: not live during the ranking
        if (v2) {
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < SW; ++i) tot += whist[i * 256 + t];
            }
            u32 all;
            const u32 ds = rs_digit_excl_sum<SW, false, true>(tot, scr, &all);                    // B2 inside
            if (t < 256) {
                u32 run = ds;
#pragma unroll
                for (int i = 0; i < SW; ++i) { const u32 ci = whist[i * 256 + t]; whist[i * 256 + t] = run; run += ci; }
                dstart[x * 256 + t] = ds;
            }
            // publish t2 — its row, and its counts into its group's and its batch's rows — from the four waves that hold the digit
            // counts, here and now: earlier than any other place, and the scout's path to the iteration's last barrier stays free of
            // memory instructions (their ISSUE alone costs ~70 ns each while fifteen waves stream through the same queue).
            if (t < 256 && !(OS_ABL & 2)) {
                const u32 cnt = tot - ((!FULL && t == mask) ? ((u32)TILE - n2) : 0u);
                const u32 c1 = (u32)__shfl_down((int)cnt, 1, 64), c2 = (u32)__shfl_down((int)cnt, 2, 64), c3 = (u32)__shfl_down((int)cnt, 3, 64);
                pubg = tagpat | (u64)(cnt | (c1 << 16)) | ((u64)(c2 | (c3 << 16)) << 32);       // digits t .. t+3 (used by lanes t % 4 == 0)
                puba = (1ull << 56) | ((u64)c1 << 28) | (u64)cnt;                               // digits t, t+1 (lanes t % 2 == 0)
                if ((t & 3u) == 0u) OS_STORE(&agg[(size_t)t2 * 64 + (t >> 2)], pubg);
                if ((t & 1u) == 0u) {
                    (void)OS_ADD(&gagg[(size_t)(t2 / (u32)OS_GRP) * 128 + (t >> 1)], puba);
                    (void)OS_ADD(&bagg[(size_t)(t2 / (u32)OS_BATCH) * 128 + (t >> 1)], puba);
                }
            }
        }
        OS_PH(8);
        __syncthreads();                                                                          // B3
        OS_PH(9);
        const u32 nn = (u32)__builtin_amdgcn_readfirstlane((int)sclaim[0]);
        // t2: tile-local reorder of the keys into the staging buffer t0 has left; then the keys of the tile after it are requested
        if (v2) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const u32 d = (u32)(k[i] >> shift) & mask;
                const u32 pos = whist[w * 256 + d] + rk[i];
                rk[i] = pos;
                Sx[pos] = k[i];
            }
        }
        load_keys(nn);
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;           // own wave's counters, for the next ranking
        OS_PH(10);
        asm volatile("" :: "v"(pubg), "v"(puba));
        __syncthreads();                                                                          // B4
#pragma unroll
        for (int i = 0; i < HP; ++i) { pos0[i] = pos1[i]; pos1[i] = rk[2 * i] | ((2 * i + 1 < ITEMS ? rk[(2 * i + 1 < ITEMS) ? 2 * i + 1 : 0] : 0u) << 16); }
        OS_PH(11);
#if OS_PHASE_TIMING
        if (t == 0 && tile_no < 40u) tdbg[(((size_t)blockIdx.x * 40 + tile_no) * 2 + 0) * 16 + 14] = ((u64)t0 << 32) | t2;
#endif
        ++tile_no;
        t0 = t1; t1 = t2; t2 = nn; x ^= 1u;
    };

    // Three loops in a row — fill, steady state, drain (a workgroup is never steady again once it has seen its last ticket or the
    // one partial tile) —, so that the steady state is a loop of its own: with both instantiations behind one loop header the register
    // allocator rotates the in-flight key / value registers through copies at the header, and a copy of a register that a load is
    // still filling means s_waitcnt vmcnt(0) at the top of every iteration.
    // (32-bit scalar compares only: a 64-bit compare goes through VALU registers, and the allocator has been seen to pick one that an
    // in-flight key load is about to write)
    const u32 nfull = n / (u32)TILE;                                    // tiles below this index are full; OS_NONE is not below it
    auto is_steady = [&]() __attribute__((always_inline)) { return t0 < nfull && t2 < nfull; };
    auto any_left = [&]() __attribute__((always_inline)) { return t0 != OS_NONE || t1 != OS_NONE || t2 != OS_NONE; };
    while (any_left() && !is_steady()) iteration(std::false_type());
    // Nothing may be in flight when the steady loop is entered: the wait-count pass merges the entry state into the loop header, and
    // loads pending into the fill loop's registers would turn into (static) waits at the top of every steady iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                 // vmcnt(0)
    while (is_steady()) iteration(std::true_type());
    while (any_left()) iteration(std::false_type());
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int radix_onesweep_setup(bscgpu_ctx* c)
{
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    c->num_cus = cus;
    const char* e = getenv("BSC_RS_ONESWEEP");
    c->os_mode = e ? atoi(e) : 3;
    if (hipFuncSetAttribute((const void*)rs_onesweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)rs_onesweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)rs_hist_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * OS_MAXP * 256 * 4) != hipSuccess) {
        (void)hipGetLastError();
        c->os_mode = 0;                        // the three-kernel pass serves every sort
        c->os_available = false;
    }
    return BSC_NO_ERROR;
}

bool radix_onesweep_wanted(const bscgpu_ctx* c, u64 n, int npasses, bool has_val)
{
    if (c->os_mode == 0 || npasses < 1 || npasses > OS_MAXP) return false;
    if (!has_val && c->os_mode < 2) return false;                       // mode 1 (round 3's default): keys-only passes (ST) keep the three-kernel passes.  Round 4,
                                                                        // ST5 / ST6 on 128 MiB blocks, one box: whole sort 0.38 -> 0.58 / 0.57 of 8 TB/s, job 4429 -> 4659 MB/s
    return n >= (u64)(c->os_mode == 2 ? 4 : 512) * OS_TILE;             // mode 2 (tests): every sort of >= 4 tiles; mode 3: large sorts, keys-only too
}

int radix_onesweep_sort(bscgpu_ctx* c, u64* keys, u64* keys_alt, u32* vals, u32* vals_alt, u64 n,
                        const RadixPass* passes, int npasses)
{
    const bool has_val = vals != nullptr;
    const u32 ntiles = (u32)((n + OS_TILE - 1) / OS_TILE);
    const u32 nbatches = (ntiles + OS_BATCH - 1) / OS_BATCH;
    if (c->os_tiles_cap < ntiles) {
        if (c->os_agg) (void)hipFree(c->os_agg);
        if (c->os_zero) (void)hipFree(c->os_zero);
        c->os_agg = nullptr; c->os_zero = nullptr; c->os_tiles_cap = 0;
        const u64 cap_n = (u64)c->max_n > n ? (u64)c->max_n : n;
        const u32 cap_tiles = (u32)((cap_n + OS_TILE - 1) / OS_TILE) + 1;
        const u32 cap_batches = (cap_tiles + OS_BATCH - 1) / OS_BATCH, cap_groups = (cap_tiles + OS_GRP - 1) / OS_GRP;
        c->os_batch_words = cap_batches * 256;
        c->os_pass_stride = OS_CTL_WORDS + 256 + cap_batches * 256 + cap_groups * 256;     // words: control block, digit totals, batch rows, group rows (128 x u64 each)
        if (hipMalloc((void**)&c->os_agg, (size_t)cap_tiles * 64 * 8) != hipSuccess ||
            hipMalloc((void**)&c->os_zero, (size_t)OS_MAXP * c->os_pass_stride * 4) != hipSuccess) {
            (void)hipGetLastError();
            if (c->os_agg) { (void)hipFree(c->os_agg); c->os_agg = nullptr; }
            return ctx_fail(c, BSC_GPU_NOT_ENOUGH_MEMORY, "digit-pass tables", hipSuccess);
        }
        c->os_tiles_cap = cap_tiles;
        c->os_epoch = 0;
    }
    // per sort: tickets, digit totals and batch rows of all passes start from zero
    HIP_TRY(c, hipMemsetAsync(c->os_zero, 0, (size_t)npasses * c->os_pass_stride * 4, c->stream));

    OsPasses P;
    P.np = npasses;
    for (int p = 0; p < OS_MAXP; ++p) {
        P.shift[p] = p < npasses ? passes[p].shift : 0;
        P.mask[p]  = p < npasses ? ((passes[p].bits >= 8) ? 0xffu : ((1u << passes[p].bits) - 1u)) : 0u;
    }
    const u32 grid = ntiles < (u32)c->num_cus ? ntiles : (u32)c->num_cus;
    prof_begin(c, BSCGPU_K_RADIX_HISTALL, n * 8, n);
    hipLaunchKernelGGL(rs_hist_all_kernel, dim3(grid), dim3(OS_WG), (size_t)16 * npasses * 256 * 4, c->stream,
                       keys, (u32)n, P, c->os_zero, c->os_pass_stride);
    prof_end(c);

    u64 *ksrc = keys, *kdst = keys_alt;
    u32 *vsrc = vals, *vdst = vals_alt;
    const u64 rec_bytes = 8 + (has_val ? 4 : 0);
    // BSC_RS_FAULT_DEV=<k> (tests): the k-th single-read sort of the process finds the error word raised after its first pass, as if a
    // scout of that pass had given up: the later passes must return at once and the caller's check must fail the sort.
    static const int fault_dev_at = [] { const char* e = getenv("BSC_RS_FAULT_DEV"); return e ? atoi(e) : 0; }();
    static std::atomic<int> sorts{0};
    const bool inject = fault_dev_at > 0 && sorts.fetch_add(1) + 1 == fault_dev_at;
    for (int p = 0; p < npasses; ++p) {
        if (inject && p == 1) HIP_TRY(c, hipMemsetAsync(c->dscal + OS_ERR_SLOT, 1, 4, c->stream));
        // launch tag: 1..255 in the top byte of every tile row; the rows are cleared when the sequence wraps
        if (c->os_epoch % 255u == 0u) HIP_TRY(c, hipMemsetAsync(c->os_agg, 0, (size_t)c->os_tiles_cap * 64 * 8, c->stream));
        const u32 tag = (c->os_epoch % 255u) + 1u;
        ++c->os_epoch;
        u32* ctl = c->os_zero + (size_t)p * c->os_pass_stride;
        prof_begin(c, BSCGPU_K_RADIX_SCATTER, 2 * n * rec_bytes, n);
        if (has_val)
            hipLaunchKernelGGL(rs_onesweep_kernel<true>, dim3(grid), dim3(OS_WG), OS_LDS, c->stream,
                               ksrc, kdst, vsrc, vdst, (u32)n, P.shift[p], P.mask[p], ntiles, ctl, c->dscal + OS_ERR_SLOT, ctl + OS_CTL_WORDS, reinterpret_cast<u64*>(ctl + OS_CTL_WORDS + 256), reinterpret_cast<u64*>(ctl + OS_CTL_WORDS + 256 + c->os_batch_words), reinterpret_cast<u64*>(c->os_agg), tag, c->wc_sink);
        else
            hipLaunchKernelGGL(rs_onesweep_kernel<false>, dim3(grid), dim3(OS_WG), OS_LDS, c->stream,
                               ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, P.shift[p], P.mask[p], ntiles, ctl, c->dscal + OS_ERR_SLOT, ctl + OS_CTL_WORDS, reinterpret_cast<u64*>(ctl + OS_CTL_WORDS + 256), reinterpret_cast<u64*>(ctl + OS_CTL_WORDS + 256 + c->os_batch_words), reinterpret_cast<u64*>(c->os_agg), tag, c->wc_sink);
        prof_end(c);
        HIP_TRY(c, hipGetLastError());
        u64* tk = ksrc; ksrc = kdst; kdst = tk;
        u32* tv = vsrc; vsrc = vdst; vdst = tv;
    }
#if OS_PHASE_TIMING
    if (n >= (1u << 24)) {      // debug builds: phase stamps of the last pass
        static std::vector<u64> host(256 * 40 * 2 * 16);
        if (hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(host.data(), c->wc_sink, host.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen("gpurun_out/os_phase_timing.bin", "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
        }
    }
#endif
    // The error word is the context's, not the sort's: it is only ever cleared by radix_onesweep_check, so a give-up of ANY sort since the
    // last check is still there when the check comes (several sorts in front of one sync lose nothing).  It travels to pinned memory
    // behind the last pass; radix_onesweep_check looks at it after the caller's next sync.
    HIP_TRY(c, hipMemcpyAsync(c->hscal + OS_ERR_SLOT, c->dscal + OS_ERR_SLOT, 4, hipMemcpyDeviceToHost, c->stream));
    c->os_check_pending = true;
    (void)nbatches;
    return BSC_NO_ERROR;
}

// After the stream has been synchronised: did any single-read pass since the last check give up a wait?  Then the output of every
// sort since then is unusable (in bounds, but not sorted): the word is cleared, c->os_gave_up tells the caller that a retry through
// the three-kernel passes is in order (bwt_device / st_device do that once), and the call fails with BSC_GPU_ERROR otherwise.
// BSC_RS_FAULT=<k> (tests): the k-th check of the process reports a give-up that did not happen.
int radix_onesweep_check(bscgpu_ctx* c)
{
    if (!c->os_check_pending) return BSC_NO_ERROR;
    c->os_check_pending = false;
    static const int fault_at = [] { const char* e = getenv("BSC_RS_FAULT"); return e ? atoi(e) : 0; }();
    static std::atomic<int> checks{0};
    const bool injected = fault_at > 0 && checks.fetch_add(1) + 1 == fault_at;
    if (c->hscal[OS_ERR_SLOT] != 0 || injected) {
        c->os_gave_up = true;
        (void)hipMemsetAsync(c->dscal + OS_ERR_SLOT, 0, 4, c->stream);
        return ctx_fail(c, BSC_GPU_ERROR, "digit pass gave up waiting for a predecessor tile", hipSuccess);
    }
    return BSC_NO_ERROR;
}
