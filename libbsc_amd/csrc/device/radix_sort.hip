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
//    (MI355X_MICROARCH.md, handoff rows), while a chip at 5 TB/s retires a 96 KB tile every ~20 ns —
//    a decoupled-look-back chain ("onesweep") would serialise on that latency.  The price is one
//    extra streaming read of the keys per pass (8 B/record, rs_hist); the scatter pass itself moves
//    exactly the algorithmic 2*(8+4) B/record.
//  * <= 1024 chunks (one workgroup each, 4 per CU, all resident; block b runs on XCD b % 8 so every
//    XCD streams an equal contiguous share).  A workgroup walks its chunk tile by tile (4096 records)
//    keeping its 256 running global bucket offsets in LDS, so the per-chunk offsets table is only
//    256 x 1024 u32.
//  * Inside a tile: wave-striped coalesced loads (each wave64 load instruction covers 512 contiguous
//    bytes of keys), 8-bit digit, stable in-wave ranking by wave64 ballot match (8 ballots -> peer
//    mask, popcount of lower peers), per-wave 256-bin histograms in LDS, then the tile is locally
//    reordered through LDS (32 KB staging) so that every bucket leaves as one contiguous run:
//    consecutive lanes store consecutive addresses.  Integer/index work only — no MFMA.
#include "dev_common.h"
#include <cstdlib>

#ifndef RS_NT
#define RS_NT 5      // bit0: non-temporal loads (+10-20 %: the streamed-once input stops evicting the partially
                     // written output lines from L2), bit1: non-temporal stores (measured 20-50 % SLOWER), bit2: nt loads in rs_hist
#endif
#ifndef RS_WG
#define RS_WG 256      // measured: 512 threads / 8192-record tiles are 10-25 % slower (barrier stalls, spills)
#endif
constexpr int RS_WAVES = RS_WG / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE  = RS_WG * RS_ITEMS;       // 4096 records per tile
constexpr int RS_LDS   = RS_TILE * 8 + RS_WAVES * 256 * 4 + 3 * 256 * 4 + 16 * 4;   // 40,000 B -> 4 WG (16 waves) / CU
constexpr int RS_MAX_CHUNKS = 256 * (1024 / RS_WG);       // all workgroups resident at once: 4 per CU

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

// Exclusive sum over the first 256 threads' values (one per digit); every thread of the workgroup calls it
// (threads >= 256 pass 0).  scr: RS_WAVES u32.
__device__ __forceinline__ u32 rs_digit_excl_sum(u32 v, u32* scr, u32* total) {
    const u32 incl = wave_incl_sum(v);
    const u32 w = threadIdx.x >> 6, l = lane_id();
    __syncthreads();
    if (l == 63) scr[w] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) { const u32 t = scr[i]; if ((u32)i < w) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
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
    constexpr int WG = RS_WG, WAVES = RS_WAVES, HREP = 8;
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
template <bool HAS_VAL>
__global__ __launch_bounds__(RS_WG, 4) void rs_scatter_kernel(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                        const u32* __restrict__ vin, u32* __restrict__ vout,
                                                        u32 n, int shift, u32 mask,
                                                        u32 chunk_tiles, u32 num_chunks, u32 num_tiles,
                                                        const u32* __restrict__ offsets,
                                                        const u32* __restrict__ rowtot)
{
    constexpr int WG = RS_WG, WAVES = RS_WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* skeys  = reinterpret_cast<u64*>(smem);                       // [RS_TILE] staging (reused as u32 for values)
    u32* whist  = reinterpret_cast<u32*>(smem + RS_TILE * 8);         // [4][256] per-wave digit counts / prefixes
    u32* goff   = whist + WAVES * 256;                                // [256] running global bucket offsets
    u32* adj    = goff + 256;                                         // [256] goff - tile-local bucket start
    u32* dstart = adj + 256;                                          // [256] (scratch)
    u32* scr    = dstart + 256;                                       // [8]
    volatile u32* vwh = whist;

    const u32 t = threadIdx.x, w = t >> 6, lane = t & 63;
    const u64 lt = lanemask_lt();

    {   // global offset of this chunk's first record of digit t
        u32 tot;
        const u32 base = rs_digit_excl_sum(t < 256 ? rowtot[t] : 0u, scr, &tot);
        if (t < 256) goff[t] = base + offsets[(size_t)t * num_chunks + blockIdx.x];
    }
    __syncthreads();

    const u32 tile0 = blockIdx.x * chunk_tiles;
    u32 tile1 = tile0 + chunk_tiles;
    if (tile1 > num_tiles) tile1 = num_tiles;

    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u64 tbase = (u64)tile * RS_TILE;
        const u32 rem = (u32)((u64)n - tbase);
        const u32 nvalid = rem < (u32)RS_TILE ? rem : (u32)RS_TILE;

        // ---- wave-striped loads: wave w owns records [w*1024, w*1024+1024) of the tile -------
        u64 k[RS_ITEMS];
        u32 v[RS_ITEMS];
        const u32 wbase = w * (64 * RS_ITEMS) + lane;
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i) {
            const u32 idx = wbase + i * 64;
            k[i] = (idx < nvalid) ? (RS_NT ? __builtin_nontemporal_load(&kin[tbase + idx]) : kin[tbase + idx]) : ~0ull;
        }
        for (u32 i = t; i < (u32)WAVES * 256; i += WG) whist[i] = 0;
        __syncthreads();

        // ---- stable in-wave ranking by ballot match ---------------------------------------------
        // peers(lane) = lanes whose digit equals this lane's.  Per digit bit: nb = all-ones if this lane's bit is
        // clear; the lanes agreeing with us on that bit are (ballot ^ nb), so the running mask is one 3-input
        // boolean op per 32-bit half (v_bitop3 on gfx950).
        u32 rk[RS_ITEMS];
        const u32 lt_lo = (u32)lt, lt_hi = (u32)(lt >> 32);
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i) {
            const u32 d = (u32)(k[i] >> shift) & mask;
            u32 mlo = ~0u, mhi = ~0u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int bitm = __builtin_amdgcn_sbfe((int)d, b, 1);         // -1 if bit b set, else 0
                const u64 bal = __ballot(bitm != 0);
                const u32 nb = ~(u32)bitm;
                mlo &= (u32)bal ^ nb;
                mhi &= (u32)(bal >> 32) ^ nb;
            }
            const u32 before = vwh[w * 256 + d];       // records of digit d seen by this wave so far
            const u32 r      = (u32)(__popc(mlo & lt_lo) + __popc(mhi & lt_hi));   // peers in lower lanes
            const u32 cnt    = (u32)(__popc(mlo) + __popc(mhi));
            rk[i] = before + r;
            if (r == cnt - 1) vwh[w * 256 + d] = before + cnt;   // highest peer lane publishes
        }
        // values are fetched only now (they are not needed for ranking): keeps the ranking loop's
        // register footprint at 4 waves/SIMD, and the loads fly under the bucket scan + key reorder.
        if (HAS_VAL) {
#pragma unroll
            for (int i = 0; i < RS_ITEMS; ++i) {
                const u32 idx = wbase + i * 64;
                v[i] = (idx < nvalid) ? (RS_NT ? __builtin_nontemporal_load(&vin[tbase + idx]) : vin[tbase + idx]) : 0u;
            }
        }
        __syncthreads();

        // ---- per digit: wave prefixes, tile-local bucket start, global adjust -------------------
        {
            u32 c[WAVES];
            u32 tot = 0;
            if (t < 256) {
#pragma unroll
                for (int i = 0; i < WAVES; ++i) { c[i] = whist[i * 256 + t]; tot += c[i]; }
            }
            u32 all;
            const u32 ds = rs_digit_excl_sum(tot, scr, &all);
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

        // ---- local reorder through LDS --------------------------------------------------------
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i) {
            const u32 d = (u32)(k[i] >> shift) & mask;
            const u32 pos = whist[w * 256 + d] + rk[i];
            rk[i] = pos;
            skeys[pos] = k[i];
        }
        __syncthreads();

        u32 dd[RS_ITEMS / 4];
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const u32 q = j * WG + t;
            const u64 key = skeys[q];
            const u32 d = (u32)(key >> shift) & mask;
            if ((j & 3) == 0) dd[j >> 2] = d; else dd[j >> 2] |= d << (8 * (j & 3));
            if (q < nvalid) { if (RS_NT & 2) __builtin_nontemporal_store(key, &kout[adj[d] + q]); else kout[adj[d] + q] = key; }
        }

        if (HAS_VAL) {
            __syncthreads();
            u32* svals = reinterpret_cast<u32*>(skeys);
#pragma unroll
            for (int i = 0; i < RS_ITEMS; ++i) svals[rk[i]] = v[i];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RS_ITEMS; ++j) {
                const u32 q = j * WG + t;
                const u32 d = (dd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                if (q < nvalid) { if (RS_NT & 2) __builtin_nontemporal_store(svals[q], &vout[adj[d] + q]); else vout[adj[d] + q] = svals[q]; }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------------------------
int radix_sort_passes(bscgpu_ctx* c, u64* keys, u64* keys_alt, u32* vals, u32* vals_alt, u64 n,
                      const RadixPass* passes, int npasses, int* in_alt)
{
    *in_alt = 0;
    if (n == 0 || npasses == 0) return BSC_NO_ERROR;
    if (n >= 0xffffffffull) return BSC_BAD_PARAMETER;
    if ((((uintptr_t)keys) | ((uintptr_t)keys_alt)) & 15) return ctx_fail(c, BSC_BAD_PARAMETER, "radix keys not 16B aligned", hipSuccess);

    const Chunking ch = rs_chunking(n);
    u64 *ksrc = keys, *kdst = keys_alt;
    u32 *vsrc = vals, *vdst = vals_alt;
    const bool has_val = (vals != nullptr);
    const u64 rec_bytes = 8 + (has_val ? 4 : 0);

    for (int p = 0; p < npasses; ++p) {
        const int shift = passes[p].shift;
        const u32 mask  = (passes[p].bits >= 8) ? 0xffu : ((1u << passes[p].bits) - 1u);

        prof_begin(c, BSCGPU_K_RADIX_HIST, n * 8, n);
        hipLaunchKernelGGL(rs_hist_kernel, dim3(ch.num_chunks), dim3(RS_WG), 0, c->stream,
                           ksrc, (u32)n, shift, mask, ch.chunk_tiles, ch.num_chunks, c->counts);
        prof_end(c);

        prof_begin(c, BSCGPU_K_RADIX_SCAN, (u64)256 * ch.num_chunks * 8, 0);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(WG), 0, c->stream, c->counts, ch.num_chunks, c->rowtot);
        prof_end(c);

        prof_begin(c, BSCGPU_K_RADIX_SCATTER, 2 * n * rec_bytes, n);
        if (has_val)
            hipLaunchKernelGGL(rs_scatter_kernel<true>, dim3(ch.num_chunks), dim3(RS_WG), RS_LDS, c->stream,
                               ksrc, kdst, vsrc, vdst, (u32)n, shift, mask, ch.chunk_tiles, ch.num_chunks,
                               ch.num_tiles, c->counts, c->rowtot);
        else
            hipLaunchKernelGGL(rs_scatter_kernel<false>, dim3(ch.num_chunks), dim3(RS_WG), RS_LDS, c->stream,
                               ksrc, kdst, (const u32*)nullptr, (u32*)nullptr, (u32)n, shift, mask,
                               ch.chunk_tiles, ch.num_chunks, ch.num_tiles, c->counts, c->rowtot);
        prof_end(c);
        HIP_TRY(c, hipGetLastError());

        u64* tk = ksrc; ksrc = kdst; kdst = tk;
        u32* tv = vsrc; vsrc = vdst; vdst = tv;
    }
    *in_alt = (npasses & 1);
    return BSC_NO_ERROR;
}
