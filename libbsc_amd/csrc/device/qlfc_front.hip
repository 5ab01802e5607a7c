// qlfc_front.hip — the data-parallel front half of the QLFC coder on MI355X.
//
// What the reference does on the CPU per sub-block before any entropy coding (coder.cpp:70-109 split,
// qlfc.cpp:177-255 / :398-455 run scan + backward move-to-front) is restated here as parallel kernels
// over the whole sorted block L that stays resident in HBM after the BWT / ST:
//
//   qf_split_flags  bit q = (L[1+32q] != L[32q])            -> 2^21 bits for a 64 MiB block; the host picks the
//                                                              <= 7 cut points from popcounts (coder.cpp:83-99)
//   qf_reduce/apply run heads (forced at sub-block starts), stream compaction -> sym[j], start[j] per run,
//                   first run index of every symbol per sub-block (alphabet order of the stream header)
//   qf_tile_masks   256-bit "symbols present" set per 256 runs, and per 65536 runs
//   qf_rank         rank[j] = number of distinct symbols between run j and the next run of the same symbol
//                   (or to the end of the sub-block): one lane per run walks forward, OR-ing whole tile /
//                   super-tile masks whenever they do not contain its symbol (SURVEY.md §4.3 restatement:
//                   this is embarrassingly parallel, unlike the reference's backward MTF scan)
//
// The host coder then consumes (sym, rank, start) directly; L itself never crosses PCIe unless a sub-block turns
// out incompressible and must be stored raw.
#include "dev_common.h"
#include <type_traits>

constexpr int QF_BYTES = 16;                 // bytes per thread per tile
constexpr int QF_TILE  = WG * QF_BYTES;      // 4096 bytes per tile

struct QfSplit { u32 nblocks; u32 start[9]; };     // start[nblocks] = n

__global__ __launch_bounds__(WG) void qf_split_flags_kernel(const u8* __restrict__ L, u32 n, u32 nq, u64* __restrict__ words)
{
    const u32 q = blockIdx.x * WG + threadIdx.x;
    bool f = false;
    if (q < nq) { const u32 i = 1u + 32u * q; f = L[i] != L[i - 1]; }
    const u64 b = __ballot(f);
    if ((threadIdx.x & 63) == 0 && (q >> 6) < ((nq + 63) >> 6)) words[q >> 6] = b;
}

__device__ __forceinline__ u32 qf_heads16(const u8* __restrict__ L, u32 i0, u32 n, const QfSplit& sp, uint4& bytes)
{
    // returns a 16-bit mask: bit p set iff position i0+p starts a run (and i0+p < n)
    bytes = make_uint4(0, 0, 0, 0);
    if (i0 + 16 <= n) bytes = *reinterpret_cast<const uint4*>(L + i0);
    else { u8 tmp[16]; for (int p = 0; p < 16; ++p) tmp[p] = (i0 + p < n) ? L[i0 + p] : 0; bytes = *reinterpret_cast<uint4*>(tmp); }
    const u32 prev = (i0 > 0) ? L[i0 - 1] : 0x100u;
    const u32 w[4] = {bytes.x, bytes.y, bytes.z, bytes.w};
    u32 mask = 0, last = prev;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const u32 c = (w[p >> 2] >> (8 * (p & 3))) & 0xffu;
        if (c != last) mask |= 1u << p;
        last = c;
    }
    if (i0 == 0) mask |= 1u;
#pragma unroll
    for (int b = 1; b < 8; ++b) {
        if ((u32)b < sp.nblocks) { const u32 s = sp.start[b]; if (s >= i0 && s < i0 + 16) mask |= 1u << (s - i0); }
    }
    if (i0 + 16 > n) mask &= (n > i0) ? ((1u << (n - i0)) - 1u) : 0u;
    return mask;
}

__global__ __launch_bounds__(WG) void qf_reduce_kernel(const u8* __restrict__ L, u32 n, QfSplit sp, u32 chunk_tiles, u32 num_tiles,
                                                       u32* __restrict__ segsum)
{
    __shared__ u32 scr[8];
    const u32 tile0 = blockIdx.x * chunk_tiles;
    u32 tile1 = tile0 + chunk_tiles; if (tile1 > num_tiles) tile1 = num_tiles;
    u32 cnt = 0;
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 i0 = tile * QF_TILE + threadIdx.x * QF_BYTES;
        if (i0 >= n) continue;
        uint4 bytes;
        cnt += __popc(qf_heads16(L, i0, n, sp, bytes));
    }
    u32 tot;
    block_excl_sum(cnt, scr, &tot);
    if (threadIdx.x == 0) { segsum[blockIdx.x] = tot; segsum[MAX_CHUNKS + blockIdx.x] = 0; }
}

__global__ __launch_bounds__(WG) void qf_apply_kernel(const u8* __restrict__ L, u32 n, QfSplit sp, u32 chunk_tiles, u32 num_tiles,
                                                      const u32* __restrict__ segoff, u8* __restrict__ sym, u32* __restrict__ start,
                                                      u32* __restrict__ first_run /*[8][256]*/)
{
    __shared__ u32 scr[8];
    // first run of every symbol per sub-block: minima in LDS first, one global atomicMin per (sub-block, symbol) the workgroup
    // has seen — not one L2 round trip per run.  All eight sub-blocks have their own row: the adaptive split (coder.cpp:83-99)
    // cuts where the sampled run starts reach total / nblocks, so on a block that is constant except for a short noisy stretch the
    // cuts can be as little as 32 bytes apart and one chunk can hold all of them.
    __shared__ u32 fmin[8 * 256];
    __shared__ u32 sstart[QF_TILE];                 // the tile's runs (a tile of QF_TILE bytes has at most QF_TILE runs)
    __shared__ u8  ssym[QF_TILE];
    __shared__ u32 scut[10];                        // cut points (a per-thread index into a kernel argument would go through scratch)
    for (u32 i = threadIdx.x; i < 8 * 256; i += WG) fmin[i] = 0xffffffffu;
    if (threadIdx.x < 10) scut[threadIdx.x] = (threadIdx.x <= 8 && threadIdx.x < sp.nblocks) ? sp.start[threadIdx.x < 9 ? threadIdx.x : 8] : 0xffffffffu;
    const u32 tile0 = blockIdx.x * chunk_tiles;
    u32 tile1 = tile0 + chunk_tiles; if (tile1 > num_tiles) tile1 = num_tiles;
    u32 off = segoff[blockIdx.x];
    __syncthreads();
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 i0 = tile * QF_TILE + threadIdx.x * QF_BYTES;
        uint4 bytes = make_uint4(0, 0, 0, 0);
        u32 mask = 0;
        if (i0 < n) mask = qf_heads16(L, i0, n, sp, bytes);
        u32 tot;
        u32 lj = block_excl_sum(__popc(mask), scr, &tot);      // run index inside the tile
        const u32 w[4] = {bytes.x, bytes.y, bytes.z, bytes.w};
        // sub-block of this thread's first byte, from ALL the cut points; runs further right move on as they pass a cut
        u32 b = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) if ((u32)q < sp.nblocks && i0 >= sp.start[q]) b = q;
        u32 b_next_start = scut[b + 1];             // 0xffffffff behind the last sub-block
        while (mask) {
            const u32 p = __ffs(mask) - 1; mask &= mask - 1;
            const u32 c = (w[p >> 2] >> (8 * (p & 3))) & 0xffu;
            const u32 pos = i0 + p;
            while (pos >= b_next_start) { ++b; b_next_start = scut[b + 1]; }
            ssym[lj] = (u8)c;
            sstart[lj] = pos;
            atomicMin(&fmin[b * 256u + c], off + lj);
            ++lj;
        }
        __syncthreads();
        // the tile's runs leave through LDS: consecutive lanes write consecutive entries of both arrays
        for (u32 i = threadIdx.x; i < tot; i += WG) { sym[off + i] = ssym[i]; start[off + i] = sstart[i]; }
        off += tot;
        __syncthreads();
    }
    for (u32 i = threadIdx.x; i < 8 * 256; i += WG) {
        const u32 v = fmin[i];
        if (v != 0xffffffffu) atomicMin(first_run + i, v);
    }
}

// Symbol sets per 256 runs ("tile") and per 65 536 runs ("super tile").  Two layouts:
//   DENSE (<= 64 distinct symbols in the block, e.g. any text): symbols are renumbered 0..K-1 (lut) and a set is ONE u64;
//   otherwise a set is four u64 indexed by the raw byte.
// One wavefront per tile of 256 runs: a lane takes four consecutive symbols with one load, the sets are OR-ed across the wavefront by
// shuffles — no LDS, no barrier, a quarter of the workgroups (round 5; one 256-thread workgroup per tile with an LDS atomic per run was
// bound by workgroup dispatch: 110 K workgroups for 0.12 ms of nothing).
template <bool DENSE>
__global__ __launch_bounds__(WG) void qf_tile_masks_kernel(const u8* __restrict__ sym, u32 m, const u8* __restrict__ lut, u64* __restrict__ masks)
{
    constexpr int W = DENSE ? 1 : 4;
    __shared__ u8 slut[256];
    if (DENSE) { slut[threadIdx.x] = lut[threadIdx.x]; __syncthreads(); }
    const u32 tile = blockIdx.x * WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const u32 j0 = tile * 256u + 4u * lane;
    if (tile * 256u >= m) return;                                      // (whole wavefronts only: the shuffles below need all 64 lanes)
    u32 word = 0;
    if (j0 + 4u <= m) word = *reinterpret_cast<const u32*>(sym + j0);   // sym is a carved arena buffer: 4-byte aligned at every multiple of four
    else for (u32 b = 0; b < 4u; ++b) if (j0 + b < m) word |= (u32)sym[j0 + b] << (8u * b);
    u64 v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = 0;
#pragma unroll
    for (u32 b = 0; b < 4u; ++b) {
        if (j0 + b < m) {
            const u32 raw = (word >> (8u * b)) & 0xffu;
            const u32 c = DENSE ? slut[raw] : raw;
            if (DENSE) v[0] |= 1ull << c;
            else {
#pragma unroll
                for (int k = 0; k < W; ++k) v[k] |= ((c >> 6) == (u32)k) ? (1ull << (c & 63u)) : 0ull;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] |= __shfl_xor(v[k], d, 64);
    }
    if (lane < (u32)W) {
        u64 out = v[0];
#pragma unroll
        for (int k = 1; k < W; ++k) if (lane == (u32)k) out = v[k];
        masks[(size_t)tile * W + lane] = out;
    }
}
template <bool DENSE>
__global__ __launch_bounds__(WG) void qf_super_masks_kernel(const u64* __restrict__ masks, u32 ntiles, u64* __restrict__ super)
{
    constexpr int W = DENSE ? 1 : 4;
    __shared__ u64 red[W][WAVES];
    const u32 t = blockIdx.x * WG + threadIdx.x;
    u64 v[W];
    for (int k = 0; k < W; ++k) v[k] = (t < ntiles) ? masks[(size_t)t * W + k] : 0ull;
    for (int k = 0; k < W; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] |= __shfl_xor(v[k], d, 64);
    }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < W; ++k) red[k][threadIdx.x >> 6] = v[k];
    __syncthreads();
    if (threadIdx.x < (u32)W) { u64 r = 0; for (int w = 0; w < WAVES; ++w) r |= red[threadIdx.x][w]; super[(size_t)blockIdx.x * W + threadIdx.x] = r; }
}

struct QfRuns { u32 nblocks; u32 first[9]; };      // run index range of each sub-block; first[nblocks] = m

// QLFC rank of run j = number of distinct symbols strictly between run j and the next run of the same symbol (or, for
// the last run of a symbol in its sub-block, all distinct symbols that still follow); the last run of a sub-block is 1
// (qlfc.cpp:249 / :449).  One lane per run walks forward: inside its own 256-run tile from an LDS copy, then over whole
// tiles / super tiles whose set does not contain its symbol, then inside the tile that holds the next occurrence.
// The walk lengths are extremely uneven (a frequent symbol comes back within a few runs, a rare one after thousands), and a
// wavefront is as slow as its slowest lane: so every lane first walks at most QF_SHORT runs; the few that are still looking
// are compacted through LDS and finished by the first wavefront(s) of the workgroup, lane per unfinished run, while the others
// leave.  The long walks read the tile sets four at a time and the destination tile sixteen symbols per load.
template <bool DENSE> struct QfSet;
template <> struct QfSet<true> {
    u64 a = 0;
    __device__ __forceinline__ void add(u32 s) { a |= 1ull << s; }
    __device__ __forceinline__ void merge(const u64* p) { a |= p[0]; }
    __device__ __forceinline__ static bool has(const u64* p, u32 c) { return (p[0] >> c) & 1ull; }
    __device__ __forceinline__ u32 count() const { return (u32)__popcll(a); }
    __device__ __forceinline__ void store(u64* q) const { q[0] = a; }
    __device__ __forceinline__ void load(const u64* q) { a = q[0]; }
};
template <> struct QfSet<false> {
    u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __device__ __forceinline__ void add(u32 s) {
        const u64 bit = 1ull << (s & 63);
        const u32 w = s >> 6;
        a0 |= (w == 0) ? bit : 0; a1 |= (w == 1) ? bit : 0; a2 |= (w == 2) ? bit : 0; a3 |= (w == 3) ? bit : 0;
    }
    __device__ __forceinline__ void merge(const u64* p) { a0 |= p[0]; a1 |= p[1]; a2 |= p[2]; a3 |= p[3]; }
    __device__ __forceinline__ static bool has(const u64* p, u32 c) { return (p[c >> 6] >> (c & 63)) & 1ull; }
    __device__ __forceinline__ u32 count() const { return (u32)(__popcll(a0) + __popcll(a1) + __popcll(a2) + __popcll(a3)); }
    __device__ __forceinline__ void store(u64* q) const { q[0] = a0; q[1] = a1; q[2] = a2; q[3] = a3; }
    __device__ __forceinline__ void load(const u64* q) { a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; }
};
#ifndef QF_SHORT_N
#define QF_SHORT_N 48
#endif
constexpr u32 QF_SHORT = QF_SHORT_N;

// NARROW (<= 32 distinct symbols, any lower-case text): the lifted tables hold 32-bit sets.
template <bool DENSE, bool NARROW>
__global__ __launch_bounds__(WG) void qf_rank_kernel(const u8* __restrict__ sym, u32 m, QfRuns rb, const u8* __restrict__ lut,
                                                     const u64* __restrict__ masks, const u64* __restrict__ super,
                                                     u8* __restrict__ rank)
{
    constexpr int W = DENSE ? 1 : 4;
    constexpr u32 R = 2 * WG;                   // a lifted tile looks at its own 256 runs and the 256 behind them
    __shared__ u8 scode[R];
    __shared__ u8 slut[256];
    __shared__ u32 qn;
    __shared__ u8 qt[WG];
    __shared__ u16 qi[WG];
    __shared__ u64 qset[WG * W];
    const u32 base = blockIdx.x * WG, t = threadIdx.x;
    auto sub_end = [&](u32 j) {
        u32 re = m;
#pragma unroll
        for (int b = 8; b >= 1; --b) if ((u32)b <= rb.nblocks && j < rb.first[b]) re = rb.first[b];
        return re;
    };
    const u32 tile_end = (base + WG < m) ? base + WG : m;
    const u32 re0 = sub_end(base);
    const u32 region_end = (base + R < re0) ? base + R : re0;             // the halo stops at the sub-block's end
    if (DENSE) slut[t] = lut[t];
    if (t == 0) qn = 0;
    {
        const u32 j = base + t, jh = base + WG + t;
        const u32 raw = (j < m) ? sym[j] : 0u, rawh = (DENSE && jh < region_end) ? sym[jh] : 0u;
        __syncthreads();
        scode[t] = (u8)(DENSE ? slut[raw] : raw);
        if (DENSE) scode[WG + t] = slut[rawh];                       // the halo is only read by the lifted tiles
    }
    __syncthreads();
    // Tiles that lie inside one sub-block, sets of one word: binary lifting instead of a walk.  win[k][p] = the symbols at region
    // positions (p, p + 2^k] (cut at the region's end), built level by level (one OR per entry and level); a lane then takes the windows
    // 256, 128, .. 1 that do not contain its symbol, greedily — nine uniform steps whatever the distance, where a walk is as slow as the
    // wavefront's slowest lane.  Round 5: the region is the tile AND the 256 runs behind it.  With the tile alone every lane near the
    // tile's end — a tenth of them on text — went on into the next tile with serial 16-byte loads, compacted into the workgroup's first
    // wavefront, which then took as long as ITS slowest lane: 83 % of the kernel's 5.4e8 VALU wave-instructions per 64 MiB block
    // (profiles/r03/pmc_sq_one_block.txt; the lifting itself is ~200 per wavefront).  Now only a symbol that does not come back within
    // 256 runs of the tile's end takes that path.
    const bool lifted = DENSE && WG == 256 && re0 >= tile_end;       // workgroup-uniform
    if (lifted) {
        typedef typename std::conditional<NARROW, u32, u64>::type MT;
        __shared__ MT win[DENSE ? 9 : 1][DENSE ? R : 1];           // (static LDS is reserved whether or not the branch runs: only the dense kernels pay for it)
#pragma unroll
        for (u32 q = 0; q < 2; ++q) {
            const u32 p = t + q * WG;
            win[0][p] = (base + p + 1 < region_end) ? (MT)1 << scode[p + 1 < R ? p + 1 : p] : (MT)0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 1; k < 9; ++k) {
            const u32 h = 1u << (k - 1);
#pragma unroll
            for (u32 q = 0; q < 2; ++q) {
                const u32 p = t + q * WG;
                win[k][p] = win[k - 1][p] | ((p + h < R) ? win[k - 1][p + h] : (MT)0);
            }
            __syncthreads();
        }
        const u32 j = base + t;
        if (j < m) {
            if (j + 1 == re0) rank[j] = 1;
            else {
                const u32 c = scode[t];
                MT set = 0; u32 p = t;
#pragma unroll
                for (int k = 8; k >= 0; --k) {
                    const MT w = (p < R) ? win[k][p] : (MT)0;
                    if (!((w >> c) & (MT)1)) { set |= w; p += 1u << k; }
                }
                // p = last region position known to be free of c (possibly past the end): the symbol comes back at p + 1, or not in this region
                const bool found = base + p + 1 < region_end;
                if (found || region_end == re0) rank[j] = (u8)(NARROW ? __popc((u32)set) : __popcll((u64)set));
                else {
                    const u32 slot = atomicAdd(&qn, 1u);
                    qt[slot] = (u8)t; qi[slot] = (u16)R; qset[slot * W] = (u64)set;            // goes on at the region's end (a multiple of 256)
                }
            }
        }
    } else {   // every lane: at most QF_SHORT runs ahead, inside the tile
        const u32 j = base + t;
        if (j < m) {
            const u32 re = sub_end(j);
            if (j + 1 == re) rank[j] = 1;
            else {
                const u32 c = scode[t];
                u32 lim = re < base + WG ? re : base + WG;
                if (lim > j + 1 + QF_SHORT) lim = j + 1 + QF_SHORT;
                QfSet<DENSE> set;
                u32 i = j + 1;
                bool found = false;
                while (i < lim) { const u32 s = scode[i - base]; if (s == c) { found = true; break; } set.add(s); ++i; }
                if (found || i == re) rank[j] = (u8)set.count();
                else {
                    const u32 slot = atomicAdd(&qn, 1u);
                    qt[slot] = (u8)t; qi[slot] = (u16)(i - base); set.store(&qset[slot * W]);
                }
            }
        }
    }
    __syncthreads();
    if (t >= qn) return;
    // the unfinished runs, one per lane
    const u32 tt = qt[t];
    const u32 j = base + tt;
    const u32 re = sub_end(j);
    const u32 c = scode[tt];
    u32 i = base + qi[t];
    QfSet<DENSE> set; set.load(&qset[t * W]);
    bool found = false;
    while (i < re && (i & 255u) != 0) { const u32 s = scode[i - base]; if (s == c) { found = true; break; } set.add(s); ++i; }
    if (!found) {
        while (i + 256 <= re) {
            if ((i & 65535u) == 0 && i + 65536 <= re) {
                const u64* sm = super + (size_t)(i >> 16) * W;
                if (!QfSet<DENSE>::has(sm, c)) { set.merge(sm); i += 65536; continue; }
            }
            if (DENSE && (i & 1023u) == 0 && i + 1024 <= re) {                 // four one-word tile sets per 32 bytes
                const ulonglong2 x = *reinterpret_cast<const ulonglong2*>(masks + (i >> 8));
                const ulonglong2 y = *reinterpret_cast<const ulonglong2*>(masks + (i >> 8) + 2);
                const u64 all = x.x | x.y | y.x | y.y;
                if (!QfSet<DENSE>::has(&all, c)) { set.merge(&all); i += 1024; continue; }
            }
            const u64* tm = masks + (size_t)(i >> 8) * W;
            if (QfSet<DENSE>::has(tm, c)) break;
            set.merge(tm);
            i += 256;
        }
        // the destination tile (or the sub-block's last, partial one): i is a multiple of 256, sixteen symbols per load
        while (i < re) {
            const uint4 q = *reinterpret_cast<const uint4*>(sym + i);
            const u32 wds[4] = {q.x, q.y, q.z, q.w};
            bool hit = false;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const u32 raw = (wds[b >> 2] >> (8 * (b & 3))) & 0xffu;
                const u32 s = DENSE ? slut[raw] : raw;
                if (!hit && i + b < re) { if (s == c) hit = true; else set.add(s); }
            }
            if (hit) break;
            i += 16;
        }
    }
    rank[j] = (u8)set.count();
}

// -------------------------------------------------------------------------------------------------
// host side of the front end
// -------------------------------------------------------------------------------------------------
int qlfc_front_split(bscgpu_ctx* c, const u8* dL, u32 n, int nblocks, int* start, int* size)
{
    // coder.cpp:70-109
    if (nblocks == 1) { start[0] = 0; size[0] = (int)n; return BSC_NO_ERROR; }
    const u32 nq = (n >= 2) ? (n - 1 + 31) / 32 : 0;
    const u32 nwords = (nq + 63) / 64;
    u64* dwords = reinterpret_cast<u64*>(c->kA);          // scratch: the sort buffers are free after the BWT
    prof_begin(c, BSCGPU_K_MISC, nq * 2, 0);
    hipLaunchKernelGGL(qf_split_flags_kernel, dim3((nq + WG - 1) / WG), dim3(WG), 0, c->stream, dL, n, nq, dwords);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->hsplit, dwords, (size_t)nwords * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    const u64* w = c->hsplit;
    u64 changes = 0;
    for (u32 k = 0; k < nwords; ++k) changes += (u64)__builtin_popcountll(w[k]);
    if (changes > (u64)nblocks) {
        const u64 per = changes / (u64)nblocks;
        int id = 0; u64 seen = 0;
        start[0] = 0;
        for (u32 k = 0; k < nwords && id < nblocks - 1; ++k) {
            u64 bits = w[k];
            const u64 pc = (u64)__builtin_popcountll(bits);
            if (seen + pc < per) { seen += pc; continue; }
            while (bits && id < nblocks - 1) {
                const int b = __builtin_ctzll(bits); bits &= bits - 1;
                if (++seen == per) {
                    seen = 0;
                    const int i = 1 + 32 * (int)(k * 64 + (u32)b);
                    size[id] = i - start[id];
                    start[++id] = i;
                }
            }
        }
        size[nblocks - 1] = (int)n - start[nblocks - 1];
    } else {
        const int each = (int)n / nblocks;
        for (int p = 0; p < nblocks; ++p) { start[p] = each * p; size[p] = (p != nblocks - 1) ? each : (int)n - each * (nblocks - 1); }
    }
    return BSC_NO_ERROR;
}

// Runs + ranks of all sub-blocks of dL.  Results in pinned host memory: c->hsym / c->hrank / c->hstart (m entries),
// run_first[0..nblocks] (run index range per sub-block) and first_run[8][256].
int qlfc_front_copy_runs(bscgpu_ctx* c, u32 m, HostSlot& slot)
{
    // the run arrays of the block whose front end ran last (they live in the sort buffers until the next block's sorter starts)
    { const int rc = ctx_ensure_run_slot(c, slot); if (rc < 0) return ctx_fail(c, rc, "pinned host memory for the run arrays", hipSuccess); }
    HIP_TRY(c, hipMemcpyAsync(slot.hsym, reinterpret_cast<u8*>(c->vA), m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(slot.hrank, reinterpret_cast<u8*>(c->vB), m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(slot.hstart, c->SA, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    return BSC_NO_ERROR;
}

int qlfc_front_runs(bscgpu_ctx* c, const u8* dL, u32 n, int nblocks, const int* start, u32* m_out, u32* run_first /*[9]*/,
                    u32* first_run_host /*[8*256]*/, HostSlot& slot, bool copy_runs)
{
    QfSplit sp; sp.nblocks = (u32)nblocks;
    for (int b = 0; b < 9; ++b) sp.start[b] = (b < nblocks) ? (u32)start[b] : n;
    u8*  dsym   = reinterpret_cast<u8*>(c->vA);
    u8*  drank  = reinterpret_cast<u8*>(c->vB);
    u32* dstart = c->SA;                                   // SA / ISA are dead once L has been emitted
    u32* dfirst = c->ISA;
    u64* dmask  = c->kB;                                   // m / 8 bytes
    u64* dsuper = reinterpret_cast<u64*>(c->cpos[0]);

    const Chunking ch = make_chunking(n, QF_TILE);
    HIP_TRY(c, hipMemsetAsync(dfirst, 0xff, 8 * 256 * 4, c->stream));
    prof_begin(c, BSCGPU_K_SEG, n, 0);
    hipLaunchKernelGGL(qf_reduce_kernel, dim3(ch.num_chunks), dim3(WG), 0, c->stream, dL, n, sp, ch.chunk_tiles, ch.num_tiles, c->segsum);
    prof_end(c);
    launch_seg_scan(c, ch.num_chunks);
    prof_begin(c, BSCGPU_K_SEG, n, 0);
    hipLaunchKernelGGL(qf_apply_kernel, dim3(ch.num_chunks), dim3(WG), 0, c->stream, dL, n, sp, ch.chunk_tiles, ch.num_tiles,
                       c->segoff, dsym, dstart, dfirst);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->hscal, c->dscal, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(first_run_host, dfirst, 8 * 256 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    const u32 m = c->hscal[0];
    *m_out = m;

    // run index range of every sub-block = smallest first_run of its symbols
    QfRuns rb; rb.nblocks = (u32)nblocks;
    for (int b = 0; b < nblocks; ++b) {
        u32 lo = 0xffffffffu;
        for (int s = 0; s < 256; ++s) if (first_run_host[b * 256 + s] < lo) lo = first_run_host[b * 256 + s];
        rb.first[b] = lo;
    }
    for (int b = nblocks; b < 9; ++b) rb.first[b] = m;
    for (int b = 0; b <= nblocks; ++b) run_first[b] = rb.first[b];

    // alphabet of the block = symbols that start a run somewhere; <= 64 of them -> one-word symbol sets
    u8* hlut = reinterpret_cast<u8*>(c->hscal + 640);      // pinned; the copy below is consumed before the next block
    u8* dlut = reinterpret_cast<u8*>(c->dscal + 640);
    u32 K = 0;
    for (int s = 0; s < 256; ++s) {
        bool present = false;
        for (int b = 0; b < nblocks; ++b) present |= (first_run_host[b * 256 + s] != 0xffffffffu);
        hlut[s] = (u8)(present ? (K < 255 ? K : 255) : 0);
        K += present;
    }
    const bool dense = K <= 64;
    if (dense) HIP_TRY(c, hipMemcpyAsync(dlut, hlut, 256, hipMemcpyHostToDevice, c->stream));

    const u32 ntiles = (m + 255) / 256, nsuper = (ntiles + 255) / 256;
    prof_begin(c, BSCGPU_K_MISC, m, 0);
    if (dense) {
        hipLaunchKernelGGL(qf_tile_masks_kernel<true>, dim3((ntiles + WAVES - 1) / WAVES), dim3(WG), 0, c->stream, dsym, m, dlut, dmask);
        hipLaunchKernelGGL(qf_super_masks_kernel<true>, dim3(nsuper), dim3(WG), 0, c->stream, dmask, ntiles, dsuper);
    } else {
        hipLaunchKernelGGL(qf_tile_masks_kernel<false>, dim3((ntiles + WAVES - 1) / WAVES), dim3(WG), 0, c->stream, dsym, m, dlut, dmask);
        hipLaunchKernelGGL(qf_super_masks_kernel<false>, dim3(nsuper), dim3(WG), 0, c->stream, dmask, ntiles, dsuper);
    }
    prof_end(c);
    prof_begin(c, BSCGPU_K_GATHER, (u64)m * 2, m);
    if (dense && K <= 32) hipLaunchKernelGGL((qf_rank_kernel<true, true>), dim3((m + WG - 1) / WG), dim3(WG), 0, c->stream, dsym, m, rb, dlut, dmask, dsuper, drank);
    else if (dense)       hipLaunchKernelGGL((qf_rank_kernel<true, false>), dim3((m + WG - 1) / WG), dim3(WG), 0, c->stream, dsym, m, rb, dlut, dmask, dsuper, drank);
    else                  hipLaunchKernelGGL((qf_rank_kernel<false, false>), dim3((m + WG - 1) / WG), dim3(WG), 0, c->stream, dsym, m, rb, dlut, dmask, dsuper, drank);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    (void)dstart;
    if (copy_runs) return qlfc_front_copy_runs(c, m, slot);
    return BSC_NO_ERROR;           // the caller feeds the device coder from the arrays in HBM (and may still ask for the copy)
}
