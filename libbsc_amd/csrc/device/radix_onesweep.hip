// radix_onesweep.hip — the single-read LSD digit pass for large (u64 key, u32 value) sorts on gfx950 (MI355X).
//
// Role: cub::DeviceRadixSort::SortPairs as libcubwt calls it once per suffix sort (libcubwt.cu:718) — one histogram read per
// SORT, then one kernel per digit that reads every record once and writes it once (B_sort = m*kb + P*2*m*(kb+vb)).
// The three-kernel pass of radix_sort.hip re-reads all keys in rs_hist for every digit (+33 % traffic, 0.13 of 0.47 ms).
//
//   rs_hist_all      one streaming read of the keys, the global 256-bin histogram of every digit of the sort (<= 8).
//   rs_onesweep      persistent workgroups (one per CU, 1024 threads, 8192-record tiles).  A tile's output offsets are
//                    base(digit) + records of that digit in ALL EARLIER TILES, obtained without a second kernel:
//                      - tiles are grouped in batches of 32 consecutive tiles; a batch is claimed by ONE XCD (ticket), its tiles by
//                        the workgroups of that XCD (sub-tickets), so that — as in rs_scatter_tiled — neighbouring runs of a digit
//                        are written at about the same time by CUs that share an L2, and all CUs work inside one ~50 MB window;
//                      - a tile publishes its 256 digit counts as one row of 32-bit words {launch tag, count} (agent-scope stores:
//                        each word validates itself, no fence, no flag) and adds them to the batch's row {arrivals, sum} with
//                        agent-scope atomics (a batch row is complete when arrivals == 32);
//                      - offsets(tile j of batch G) = running sum of the complete batch rows < G (each workgroup keeps its own,
//                        typically 8 new rows per tile) + the tile rows j' < j of its own batch (<= 31 rows): ~24 rows of 1 KB on
//                        average, read by all 1024 threads at once (digit = t & 255, four row groups).
//                    The hand-off latency (1-3 us cross-XCD on this chip, MI355X_MICROARCH.md hand-off table) is hidden by software
//                    pipelining instead of being avoided: a workgroup ranks tile i+1 and publishes its counts BEFORE it finishes
//                    tile i, whose keys wait, locally reordered, in a second LDS staging buffer.  By the time tile i needs its
//                    predecessors' rows they have been visible for a whole tile time (~10 us); the loads are issued at the top of
//                    the iteration and consumed after the ranking of tile i+1.  Rows that are still missing are polled.
//   Progress: tickets are taken by RUNNING workgroups only, batches in global order, tiles of a batch in order, and a workgroup
//   publishes the counts of its next tile before it waits for anything; every wait is for tiles with a smaller index, whose
//   owners are running (or, for unclaimed tiles of an installed batch, will be claimed by the running workgroups of the XCD that
//   installed it before those wait on anything larger).  So the pass cannot deadlock under partial residency — several contexts
//   share a GPU in bench.py, and a statically partitioned look-back would.  Polls are bounded all the same: a give-up sets an error
//   word that the host turns into LIBBSC_GPU_ERROR instead of hanging the device.
//   Placement (XCC id) is used for speed only; any workgroup may run anywhere (MI355X_MICROARCH.md: dispatch contract).
// Stable: output order inside a digit = tile order, then the tile-local stable rank (rs_rank_wave), exactly as rs_scatter.
#include "dev_common.h"
#include "radix_dev.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

constexpr int OS_WG = 1024, OS_WAVES = OS_WG / 64, OS_ITEMS = 8, OS_TILE = OS_WG * OS_ITEMS, OS_BATCH = 32, OS_MAXP = 8;
constexpr u32 OS_NONE = 0xffffffffu;
constexpr u32 OS_SPIN_LIMIT = 1u << 18;
// LDS: two key staging buffers, per-wave digit counters, five 256-entry tables, scratch
constexpr int OS_LDS = 2 * OS_TILE * 8 + OS_WAVES * 256 * 4 + 6 * 256 * 4 + 32 * 4;
static_assert(OS_LDS <= 160 * 1024, "rs_onesweep does not fit the CU's LDS");
// per-pass control block (u32 words): [0] next batch ticket, [8 + x] per-XCD claim word ((batch + 1) << 16 | count); word [1] of the
// FIRST pass's block is the error word of the whole sort
constexpr int OS_CTL_WORDS = 32;

struct OsPasses { int np; int shift[OS_MAXP]; u32 mask[OS_MAXP]; };

// Debug builds (tools/build_variant.sh): -DOS_PHASE_TIMING=1 stamps s_memtime at the phase boundaries of the first 40 tiles of every
// workgroup (threads 0 and 960) into the context's scratch buffer; -DOS_ABL=bits removes parts of the protocol for timing only
// (1: no look-back loads / polls, 2: no publishing, 4: static tile order instead of tickets) — results are wrong with any bit set.
#ifndef OS_PHASE_TIMING
#define OS_PHASE_TIMING 0
#endif
#ifndef OS_ABL
#define OS_ABL 0
#endif
#ifndef OS_VALS_EARLY
#define OS_VALS_EARLY 0        // 1: the values of `cur` are requested at the top of the iteration (8 more registers live under the ranking)
#endif
#if OS_PHASE_TIMING
#define OS_PH(i) do { if ((t == 0 || t == 960) && tile_no < 40u) tdbg[(((size_t)blockIdx.x * 40 + tile_no) * 2 + (t ? 1 : 0)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OS_PH(i) do { } while (0)
#endif

#define OS_LOAD(p)      __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OS_STORE(p, v)  __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OS_ADD(p, v)    __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OS_XCHG(p, v)   __hip_atomic_exchange((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

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
    const u64 stride = (u64)gridDim.x * OS_TILE;
    for (u64 base = (u64)blockIdx.x * OS_TILE; base < n; base += stride) {
        const u64 i = base + 2 * t;
        if (base + OS_TILE <= n) {
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
// Ticket: next tile for a workgroup running on XCD x (thread 0 only).  W = ctl[8 + x] = (batch + 1) << 16 | sub-tickets handed
// out; batch field 0 = nothing installed yet, 0xffff = no batches left.
// ---------------------------------------------------------------------------------------------
// The common case is one returning atomic whose result is a tile; it is issued at the top of an iteration (os_claim_issue) and looked
// at several microseconds later (os_claim_finish), so the wave does not sit in s_waitcnt vmcnt(0) behind its own streaming loads.
// (The address is made to look divergent — `opaque0` is a zero the compiler cannot see through —, otherwise the atomic optimiser
// rewrites a uniform returning atomic into mbcnt / readfirstlane form and waits for it on the spot.)
#ifndef OS_ORDER
#define OS_ORDER 0        // 0: tile = global ticket (tiles are claimed in index order); 1: batches of 32 tiles per XCD (A/B builds)
#endif
__device__ __forceinline__ u32 os_claim_issue(u32* ctl, const u32 x, const u32 opaque0) { return OS_ADD(ctl + (OS_ORDER ? 8 + x : 0) + opaque0, 1u); }

__device__ __forceinline__ u32 os_claim_finish(u32 wv, u32* ctl, u32* err, const u32 x, const u32 ntiles, const u32 nbatches)
{
    if (!OS_ORDER) return wv < ntiles ? wv : OS_NONE;
    u32* W = ctl + 8 + x;
    for (u32 guard = 0; guard < 64; ++guard) {
        const u32 b = wv >> 16, j = wv & 0xffffu;
        if (b == 0xffffu) return OS_NONE;
        if (b != 0u && j < (u32)OS_BATCH) { const u32 T = (b - 1u) * OS_BATCH + j; return T < ntiles ? T : OS_NONE; }
        if ((b == 0u && j == 0u) || (b != 0u && j == (u32)OS_BATCH)) {        // this workgroup installs the XCD's next batch
            const u32 G = OS_ADD(&ctl[0], 1u);
            if (G >= nbatches) { OS_XCHG(W, 0xffff0000u); return OS_NONE; }
            OS_XCHG(W, ((G + 1u) << 16) | 1u);
            return G * OS_BATCH;
        }
        // another workgroup of this XCD is installing: wait for the word to change, then draw again
        u32 spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(4);
            const u32 w2 = OS_LOAD(W);
            if ((w2 >> 16) != b || (b != 0u && (w2 & 0xffffu) < (u32)OS_BATCH)) break;
            if (++spins > OS_SPIN_LIMIT) { OS_ADD(err, 1u); return OS_NONE; }
        }
        wv = OS_ADD(W, 1u);
    }
    OS_ADD(err, 1u);
    return OS_NONE;
}

// ---------------------------------------------------------------------------------------------
// rs_onesweep: one digit pass, records read once and written once.
// ---------------------------------------------------------------------------------------------
template <bool HAS_VAL>
__global__ __launch_bounds__(OS_WG) void rs_onesweep_kernel(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                            const u32* __restrict__ vin, u32* __restrict__ vout,
                                                            u32 n, int shift, u32 mask, u32 ntiles,
                                                            u32* ctl, u32* err, const u32* __restrict__ totals,
                                                            u32* bagg /*[batches][256]: arrivals << 24 | sum*/,
                                                            u32* agg /*[tiles][256]: tag | count*/, u32 tag, u64* tdbg)
{
    (void)tdbg;
    u32 tile_no = 0; (void)tile_no;
    constexpr int WG = OS_WG, WAVES = OS_WAVES, ITEMS = OS_ITEMS, TILE = OS_TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* S      = reinterpret_cast<u64*>(smem);                         // [2][TILE] locally reordered keys (then values) of the two tiles in flight
    u32* whist  = reinterpret_cast<u32*>(smem + 2 * TILE * 8);          // [WAVES][256]
    u32* Rrun   = whist + WAVES * 256;                                  // [256] digit base + counts of all complete batches accounted so far
    u32* adj    = Rrun + 256;                                           // [256] output position of staging slot q of digit d = adj[d] + q
    u32* dstart = adj + 256;                                            // [2][256] tile-local start of every digit, per staging buffer
    u32* accA   = dstart + 512;                                         // [256] look-back partial sums: tile rows of the own batch
    u32* accB   = accA + 256;                                           // [256] ... batch rows
    u32* scr    = accB + 256;                                           // [16]
    u32* sclaim = scr + 16;                                             // [1] the ticket thread 0 took at the top of the iteration
    lds_vu32* vwh = (lds_vu32*)whist;

    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;
    const u32 dig = t & 255u, grp = t >> 8;                             // look-back role: digit, row group
    const u32 nbatches = (ntiles + OS_BATCH - 1) / OS_BATCH;
    u32 xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    u32 opaque0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));

    {
        u32 tot;
        const u32 base = rs_digit_excl_sum<WAVES, true, true>(t < 256 ? totals[t] : 0u, scr, &tot);
        if (t < 256) { Rrun[t] = base; accA[t] = 0; accB[t] = 0; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;
    if (OS_ABL & 4) { if (t == 0) { const u32 f = (blockIdx.x & 7u) * 32u + (blockIdx.x >> 3); sclaim[0] = f < ntiles ? f : OS_NONE; } }
    else if (t == 0) sclaim[0] = os_claim_finish(os_claim_issue(ctl, xcc, opaque0), ctl, err, xcc, ntiles, nbatches);
    __syncthreads();
    u32 cur = OS_NONE, nxt = sclaim[0];
    if (nxt == OS_NONE) return;
    bool more = true;                                                   // tickets may still yield tiles
    u32 cb = 0;                                                         // staging buffer of `cur`; `nxt` goes to cb ^ 1
    u32 gbase = 0;                                                      // batches [0, gbase) are in Rrun

    const u32 wbase = w * (64 * ITEMS) + lane;
    u64 k[ITEMS];
    u32 v[ITEMS], rk[ITEMS];
    u32 posA[ITEMS / 2];                                                // staging slots of `cur`'s records, two 16-bit slots per word
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
    load_keys(nxt);
#pragma unroll
    for (int i = 0; i < ITEMS / 2; ++i) posA[i] = 0;
    __syncthreads();                                                    // everybody has read sclaim

    // a row word that was not there yet: poll until its top byte says so (bounded; a give-up poisons the sort's error word)
    auto poll_row = [&](u32* table, const u32 row, const u32 fmask, const u32 want, bool& ok) __attribute__((always_inline)) -> u32 {
        u32* p = &table[(size_t)row * 256 + dig];
        u32 x = OS_LOAD(p), spins = 0;
        while ((x & fmask) != want && ok) {
            __builtin_amdgcn_s_sleep(2);
            x = OS_LOAD(p);
            if (++spins > OS_SPIN_LIMIT || ((spins & 1023u) == 0u && OS_LOAD(err) != 0u)) ok = false;
        }
        return x;
    };

    // One iteration: rank + publish `nxt`, then look back for and write out `cur`.  STEADY = both are full tiles (no guards).
    auto iteration = [&](auto steady_tag) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_tag)::value;
        const bool cv = ST || cur != OS_NONE, nv = ST || nxt != OS_NONE;
        u32 cur_n = (u32)TILE, nxt_n = (u32)TILE;                      // valid records
        if (!ST) {
            cur_n = cv ? ((n - cur * (u32)TILE) < (u32)TILE ? (n - cur * (u32)TILE) : (u32)TILE) : 0u;
            nxt_n = nv ? ((n - nxt * (u32)TILE) < (u32)TILE ? (n - nxt * (u32)TILE) : (u32)TILE) : 0u;
        }
        u64* Sc = S + (size_t)cb * TILE;
        u64* Sn = S + (size_t)(cb ^ 1u) * TILE;

        OS_PH(0);
        // (0) the ticket for the tile after `nxt`: drawn now, looked at in front of the third barrier
        u32 ticket = 0;
        if (!(OS_ABL & 4) && t == 0 && more) ticket = os_claim_issue(ctl, xcc, opaque0);

        // (1) look-back loads for `cur`: <= 8 tile rows of its batch and <= 2 batch rows per thread, all in flight under the ranking
        const u32 cj = cv ? (cur & (u32)(OS_BATCH - 1)) : 0u, cG = cv ? (cur / (u32)OS_BATCH) : 0u;
        if (HAS_VAL && OS_VALS_EARLY) load_vals(cur);
        u32 la[8], lb[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32 jj = grp + 4u * q;
            const u32 row = (jj < cj) ? (cur - cj + jj) : 0u;
            la[q] = (OS_ABL & 1) ? tag : OS_LOAD(&agg[(size_t)row * 256 + dig]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32 gg = gbase + grp + 4u * q;
            lb[q] = (OS_ABL & 1) ? ((u32)OS_BATCH << 24) : OS_LOAD(&bagg[(size_t)(gg < cG ? gg : 0u) * 256 + dig]);
        }
        OS_PH(1);

        // (2) rank `nxt` inside its waves
        if (nv) {
            if (!ST) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) if (wbase + i * 64 >= nxt_n) k[i] = ~0ull;       // padding sorts last
            }
            rs_rank_wave<ITEMS>(k, shift, mask, vwh + w * 256, rk);
        }
        OS_PH(2);
        __syncthreads();                                                                          // B1
        OS_PH(3);
        if (HAS_VAL && !OS_VALS_EARLY) load_vals(cur);                  // needed behind B5: not live during the ranking
        if (nv) {
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < WAVES; ++i) tot += whist[i * 256 + t];
            }
            u32 all;
            const u32 ds = rs_digit_excl_sum<WAVES, false, true>(tot, scr, &all);                       // B2 inside
            if (t < 256) {
                u32 run = ds;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { const u32 ci = whist[i * 256 + t]; whist[i * 256 + t] = run; run += ci; }
                dstart[(cb ^ 1u) * 256 + t] = ds;
                const u32 cnt = tot - ((!ST && t == mask) ? ((u32)TILE - nxt_n) : 0u);
                if (!(OS_ABL & 2)) {
                    OS_STORE(&agg[(size_t)nxt * 256 + t], tag | cnt);
                    (void)OS_ADD(&bagg[(size_t)(nxt / (u32)OS_BATCH) * 256 + t], cnt | (1u << 24));
                }
            }
        }
        OS_PH(4);

        // (3) sum what the look-back loads brought; poll rows that were not there yet
        if (cv) {
            u32 sa = 0, sb = 0;
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const u32 jj = grp + 4u * q;
                if (jj < cj) {
                    u32 x = la[q];
                    if ((x & 0xff000000u) != tag) x = poll_row(agg, cur - cj + jj, 0xff000000u, tag, ok);
                    sa += x & 0xffffffu;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32 gg = gbase + grp + 4u * q;
                if (gg < cG) {
                    u32 x = lb[q];
                    if ((x >> 24) != (u32)OS_BATCH) x = poll_row(bagg, gg, 0xff000000u, (u32)OS_BATCH << 24, ok);
                    sb += x & 0xffffffu;
                }
            }
            for (u32 gg = gbase + 8u + grp; gg < cG && !(OS_ABL & 1); gg += 4u) {       // a workgroup that fell behind (or has just started)
                sb += poll_row(bagg, gg, 0xff000000u, (u32)OS_BATCH << 24, ok) & 0xffffffu;
            }
            if (!ok) (void)OS_ADD(err, 1u);
            if (sa) atomicAdd(&accA[dig], sa);
            if (sb) atomicAdd(&accB[dig], sb);
            gbase = cG;
        }
        if (OS_ABL & 4) {                                               // static order: rounds of gridDim.x tiles, 32 consecutive tiles per XCD
            if (t == 0) { const u32 s2 = (nxt != OS_NONE ? nxt : cur) + gridDim.x; sclaim[0] = (more && s2 < ntiles) ? s2 : OS_NONE; }
        } else if (t == 0) sclaim[0] = more ? os_claim_finish(ticket, ctl, err, xcc, ntiles, nbatches) : OS_NONE;
        OS_PH(5);
        __syncthreads();                                                                          // B3
        OS_PH(6);
        const u32 nn = sclaim[0];
        if (nn == OS_NONE) more = false;

        // (4) `nxt`: tile-local reorder of the keys into its staging buffer; then the keys of the tile after it are requested
        if (nv) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const u32 d = (u32)(k[i] >> shift) & mask;
                const u32 pos = whist[w * 256 + d] + rk[i];
                rk[i] = pos;
                Sn[pos] = k[i];
            }
        }
        load_keys(nn);
#pragma unroll
        for (int i = 0; i < 4; ++i) whist[w * 256 + i * 64 + lane] = 0;       // own wave's counters, for the next ranking
        if (cv && t < 256) {
            const u32 r = Rrun[t] + accB[t];
            Rrun[t] = r;
            adj[t] = r + accA[t] - dstart[cb * 256 + t];
            accA[t] = 0; accB[t] = 0;
        }
        OS_PH(7);
        __syncthreads();                                                                          // B4
        OS_PH(8);

        // (5) `cur` leaves: every digit as one contiguous run, consecutive lanes -> consecutive addresses
        u32 dd[ITEMS / 4];
        if (cv) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 q = j * WG + t;
                const u64 key = Sc[q];
                const u32 d = (u32)(key >> shift) & mask;
                if ((j & 3) == 0) dd[j >> 2] = d; else dd[j >> 2] |= d << (8 * (j & 3));
                if (ST || q < cur_n) kout[adj[d] + q] = key;
            }
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS / 4; ++j) dd[j] = 0;
        }
        OS_PH(9);
        if (HAS_VAL) {
            __syncthreads();                                                                      // B5
            OS_PH(10);
            u32* svals = reinterpret_cast<u32*>(Sc);
            if (cv) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) svals[(posA[i >> 1] >> (16 * (i & 1))) & 0xffffu] = v[i];
            }
            OS_PH(11);
            __syncthreads();                                                                      // B6
            OS_PH(12);
            if (cv) {
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const u32 q = j * WG + t;
                    const u32 d = (dd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    if (ST || q < cur_n) vout[adj[d] + q] = svals[q];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ITEMS / 2; ++i) posA[i] = rk[2 * i] | (rk[2 * i + 1] << 16);
        OS_PH(13);
#if OS_PHASE_TIMING
        if ((t == 0 || t == 960) && tile_no < 40u) tdbg[(((size_t)blockIdx.x * 40 + tile_no) * 2 + (t ? 1 : 0)) * 16 + 14] = ((u64)cur << 32) | nxt;
#endif
        ++tile_no;
        cur = nxt; nxt = nn; cb ^= 1u;
    };

    while (cur != OS_NONE || nxt != OS_NONE) {
        const bool steady = cur != OS_NONE && nxt != OS_NONE && (u64)(cur + 1u) * TILE <= n && (u64)(nxt + 1u) * TILE <= n;
        if (steady) iteration(std::true_type());
        else iteration(std::false_type());
    }
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
    c->os_mode = e ? atoi(e) : 1;
    if (hipFuncSetAttribute((const void*)rs_onesweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)rs_onesweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)rs_hist_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * OS_MAXP * 256 * 4) != hipSuccess) {
        (void)hipGetLastError();
        c->os_mode = 0;                        // the three-kernel pass serves every sort
    }
    return BSC_NO_ERROR;
}

bool radix_onesweep_wanted(const bscgpu_ctx* c, u64 n, int npasses, bool has_val)
{
    if (c->os_mode == 0 || npasses < 1 || npasses > OS_MAXP) return false;
    if (!has_val && c->os_mode != 2) return false;                      // keys-only passes (ST): the 256 x 16 kernel wins on text digits
    return n >= (u64)(c->os_mode == 2 ? 4 : 512) * OS_TILE;             // mode 2 (tests): every sort of >= 4 tiles
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
        const u32 cap_batches = (cap_tiles + OS_BATCH - 1) / OS_BATCH;
        c->os_pass_stride = OS_CTL_WORDS + 256 + cap_batches * 256;      // words: control block, digit totals, batch rows
        if (hipMalloc((void**)&c->os_agg, (size_t)cap_tiles * 256 * 4) != hipSuccess ||
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
    prof_begin(c, BSCGPU_K_RADIX_HIST, n * 8, n);
    hipLaunchKernelGGL(rs_hist_all_kernel, dim3(grid), dim3(OS_WG), (size_t)16 * npasses * 256 * 4, c->stream,
                       keys, (u32)n, P, c->os_zero, c->os_pass_stride);
    prof_end(c);

    u64 *ksrc = keys, *kdst = keys_alt;
    u32 *vsrc = vals, *vdst = vals_alt;
    const u64 rec_bytes = 8 + (has_val ? 4 : 0);
    for (int p = 0; p < npasses; ++p) {
        // launch tag: 1..255 in the top byte of every tile row; the rows are cleared when the sequence wraps
        if (c->os_epoch % 255u == 0u) HIP_TRY(c, hipMemsetAsync(c->os_agg, 0, (size_t)c->os_tiles_cap * 256 * 4, c->stream));
        const u32 tag = ((c->os_epoch % 255u) + 1u) << 24;
        ++c->os_epoch;
        u32* ctl = c->os_zero + (size_t)p * c->os_pass_stride;
        prof_begin(c, BSCGPU_K_RADIX_SCATTER, 2 * n * rec_bytes, n);
        if (has_val)
            hipLaunchKernelGGL(rs_onesweep_kernel<true>, dim3(grid), dim3(OS_WG), OS_LDS, c->stream,
                               ksrc, kdst, vsrc, vdst, (u32)n, P.shift[p], P.mask[p], ntiles, ctl, c->os_zero + 1, ctl + OS_CTL_WORDS, ctl + OS_CTL_WORDS + 256, c->os_agg, tag, c->wc_sink);
        else
            hipLaunchKernelGGL(rs_onesweep_kernel<false>, dim3(grid), dim3(OS_WG), OS_LDS, c->stream,
                               ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, P.shift[p], P.mask[p], ntiles, ctl, c->os_zero + 1, ctl + OS_CTL_WORDS, ctl + OS_CTL_WORDS + 256, c->os_agg, tag, c->wc_sink);
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
    // the sort's error word travels to pinned memory behind the last pass; radix_onesweep_check looks at it after the caller's next sync
    HIP_TRY(c, hipMemcpyAsync(c->hscal + OS_ERR_SLOT, c->os_zero + 1, 4, hipMemcpyDeviceToHost, c->stream));
    c->os_check_pending = true;
    (void)nbatches;
    return BSC_NO_ERROR;
}

// After the stream has been synchronised: did any pass of the last sort give up a wait?  (Cannot happen by construction;
// a non-zero word means corrupted tables, and the sort's output must not be used.)
int radix_onesweep_check(bscgpu_ctx* c)
{
    if (!c->os_check_pending) return BSC_NO_ERROR;
    c->os_check_pending = false;
    if (c->hscal[OS_ERR_SLOT] != 0) return ctx_fail(c, BSC_GPU_ERROR, "digit pass gave up waiting for a predecessor tile", hipSuccess);
    return BSC_NO_ERROR;
}
