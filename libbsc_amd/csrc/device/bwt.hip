// bwt.hip — forward Burrows-Wheeler transform of one block on MI355X: suffix sort = one LSD radix sort on packed prefix
// keys (radix_sort.hip), refinement rounds as segmented sorts on text keys, prefix doubling only for blocks with long repeats;
// then the BWT byte emit.
//
// Result contract (what bsc_bwt_encode returns, bwt.cpp:178-231, via libsais_bwt_aux, libsais.c:6704):
//   SA  = suffixes of T[0..n) in lexicographic order, a proper prefix sorting first;
//   L   = [T[n-1]] ++ [T[SA[j]-1] for j = 0..n-1 if SA[j] != 0];  primary index = ISA[0] + 1;
//   aux = I[t] = ISA[t*r] + 1 for t = 0..(n-1)/r  (I[0] is the primary index).
// The reference GPU path (libcubwt.cu:2031-2223: DC3 2/3 sample + 64-bit prefix sort + segmented sort
// + merge) is NOT followed; this is a different algorithm with the same result:
//
//   1. bwt_pack:   key[s] = the first w characters of suffix i as dense alphabet codes (w = 8..16), value = i (+ the code of
//                  the character in front of it in the spare high bits).  The < w "tail" suffixes whose window crosses the
//                  block end are placed FIRST in input order, shortest first; the LSD sort is stable, so inside a group of
//                  equal padded keys they come out first and already in final order ("proper prefix is smaller"), and seg
//                  marks each of them as a finished singleton.
//   2. <= 8 radix passes over (u64 key, u32 suffix) -> order by w-character prefix.
//   3. seg (reduce / scan / apply): group heads, rank = position of the group head (so ranks are valid SA slots and only
//      ever grow under refinement), stream compaction of every suffix still sharing its rank ("unsorted").
//   4. rounds on text keys (bwt_round_textsort_kernel): the groups are refined by the next a characters of the text behind
//      the h characters they share — no inverse suffix array; h += a.  Blocks that converge this way (text: two rounds)
//      never build ISA.
//   5. otherwise ISA is built once from the current order and the rounds double h (bwt_round_segsort_kernel on ISA[sa+h],
//      radix fallback for groups longer than a workgroup sorts).  One 4-byte D2H + stream sync per round for the loop
//      test (libcubwt does the same, libcubwt.cu:1383).
//   6. bwt_find / bwt_emit: primary and aux indexes from a scan of SA, L from SA (predecessor codes) or SA/T.
#include "dev_common.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int SEG_ITEMS = 8;
constexpr int SEG_TILE  = WG * SEG_ITEMS;      // 2048 records per tile, 8 consecutive per thread

// ---------------------------------------------------------------------------------------------
// Byte histogram of the block (replicated LDS bins against skew) -> alphabet size K and dense codes.
__global__ __launch_bounds__(WG) void bwt_bytehist_kernel(const u8* __restrict__ T, u32 n, u32 chunk_bytes, u32* __restrict__ hist)
{
    __shared__ u32 h[8 * 256];
    for (u32 i = threadIdx.x; i < 8 * 256; i += WG) h[i] = 0;
    __syncthreads();
    u32* hw = h + (threadIdx.x & 7) * 256;
    const u64 start = (u64)blockIdx.x * chunk_bytes;
    u64 end = start + chunk_bytes; if (end > n) end = n;
    for (u64 i = start + 16ull * threadIdx.x; i < end; i += 16ull * WG) {
        if (i + 16 <= end) {
            const uint4 q = *reinterpret_cast<const uint4*>(T + i);
            const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                atomicAdd(&hw[w[x] & 255u], 1u); atomicAdd(&hw[(w[x] >> 8) & 255u], 1u);
                atomicAdd(&hw[(w[x] >> 16) & 255u], 1u); atomicAdd(&hw[w[x] >> 24], 1u);
            }
        } else for (u64 p = i; p < end; ++p) atomicAdd(&hw[T[p]], 1u);
    }
    __syncthreads();
    u32 sum = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += h[r * 256 + threadIdx.x];
    if (sum) atomicAdd(&hist[threadIdx.x], sum);
}

// Alphabet-compacted prefix keys.  codes[c] = order-preserving dense code of byte c (bits per code = cb), a key
// holds the first w = min(16, 64 / cb) characters of the suffix, most significant first, zero (= smallest code)
// padded past the block end.  Text with <= 32 distinct bytes thus sorts on 12-16 characters in the same 8 passes
// that raw bytes spend on 8, which removes most of the first doubling round's work.
// Suffixes whose window crosses the end ("tails", i + w > n, at most w-1 of them) go FIRST in input order, shortest
// first: the stable sort then leaves them first inside any group of equal padded keys, already in final order.
#ifndef BWT_RADIX_DEFAULT
#define BWT_RADIX_DEFAULT 0
#endif
struct PackParams { u32 cb; u32 w; u32 tc; u32 low_shift; u32 pred_shift; u32 radix; u64 top; };   // radix != 0: the key is the number with digits c_0 .. c_{w-1} in base `radix` (top = radix^(w-1)): one more character than 64 / cb where radix^(w) < 2^64 (28 symbols: 13 instead of 12);   // pred_shift: 0 = values are plain suffix indexes,
                                                                               // else value = index | code(T[i-1]) << pred_shift

__global__ __launch_bounds__(WG) void bwt_pack_kernel(const u8* __restrict__ T, u32 n, PackParams pp,
                                                      const u8* __restrict__ codes, u64* __restrict__ keys, u32* __restrict__ vals)
{
    // A thread builds the keys of four consecutive suffixes (their windows share all but three characters); written from there the
    // stores of a wavefront have a 32-byte lane pitch — every line of the output is touched by four instructions.  The workgroup's
    // 1024 records therefore go through LDS and leave lane-contiguous (the few tail suffixes, whose slots are elsewhere, directly).
    __shared__ u8 lut[256];
    __shared__ u64 skey[4 * WG];
    __shared__ u32 sval[4 * WG];
    lut[threadIdx.x] = codes[threadIdx.x];
    __syncthreads();
    const u32 b0 = 4u * blockIdx.x * WG;                                 // first suffix of the workgroup
    const u32 i0 = b0 + 4u * threadIdx.x;
    if (i0 < n) {
        // characters i0 .. i0 + w + 2 (<= 19 bytes; T is 16-B aligned at T[0] and zero padded for 32 bytes past n)
        const u32* T32 = reinterpret_cast<const u32*>(T + i0);
        u32 wds[5];
#pragma unroll
        for (int x = 0; x < 5; ++x) wds[x] = T32[x];
        u32 pcode = (pp.pred_shift && i0 > 0) ? (u32)lut[T[i0 - 1]] : 0u;        // code of the character in front of suffix i0
        u64 cd[19];                                     // codes of characters i0 .. i0+18 (0 past the end)
#pragma unroll
        for (u32 t = 0; t < 19; ++t) {
            const u32 c = (wds[t >> 2] >> (8 * (t & 3))) & 0xffu;
            cd[t] = (i0 + t < n) ? (u64)lut[c] : 0ull;
        }
        u64 rkey = 0;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            const u32 i = i0 + j;
            if (i < n) {
                u64 key = 0;
                if (pp.radix) {
                    // base-K number of the w characters: rolled from the previous suffix's key (drop its first digit, shift, append one)
                    if (j == 0) {
#pragma unroll
                        for (u32 t = 0; t < 16; ++t) if (t < pp.w) key = key * pp.radix + cd[t];
                    } else {
                        u64 nxt = 0;                                   // (static register indices: cd[j + w - 1] by a select chain)
#pragma unroll
                        for (u32 t = 0; t < 16; ++t) if (t + 1 == pp.w) nxt = cd[j + t];
                        key = (rkey - cd[j - 1] * pp.top) * pp.radix + nxt;
                    }
                    rkey = key;
                } else {
#pragma unroll
                for (u32 t = 0; t < 16; ++t) if (t < pp.w) key |= cd[j + t] << (64 - pp.cb * (t + 1));
                }
                const u32 val = pp.pred_shift ? (i | (pcode << pp.pred_shift)) : i;
                const bool tail = (u64)i + pp.w > (u64)n;
                if (tail) { keys[n - 1 - i] = key; vals[n - 1 - i] = val; }
                else { skey[4u * threadIdx.x + j] = key; sval[4u * threadIdx.x + j] = val; }
                pcode = (u32)cd[j];                                                 // this suffix's first character precedes the next one
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 li = k * WG + threadIdx.x, i = b0 + li;
        if (i < n && (u64)i + pp.w <= (u64)n) { keys[i + pp.tc] = skey[li]; vals[i + pp.tc] = sval[li]; }
    }
}

// ---------------------------------------------------------------------------------------------
// seg phase A: head / unsorted flags + per-chunk summaries.
//   head(x)  = x == 0 || key[x] != key[x-1] || (INITIAL && (tail(sa[x]) || tail(sa[x-1])))
//   uns(x)   = !(head(x) && head(x+1)), head(m) = 1
//   flags[x] = head | uns << 1 | oldhead << 2
//   oldhead (doubling rounds, grp_shift = the width of the key's next-rank field): x is also the first record of a group of the round's
//   INPUT — the group rank in the key's high bits changes there.  Records of a new group that starts at an old head keep their rank, and
//   seg_apply skips their (random, 4-byte) ISA stores: most of them while the unsorted set shrinks slowly.
// ---------------------------------------------------------------------------------------------
//   grp (rounds on text keys, see bwt_round_textsort_kernel): the group rank of every record is a separate array and a head is
//   also where it changes: head(x) |= grp[x] != grp[x-1]
template <bool INITIAL>
__global__ __launch_bounds__(WG) void seg_reduce_kernel(const u64* __restrict__ keys, const u32* __restrict__ sa,
                                                        u32 m, u32 tail_lo, u32 smask, u32 chunk_tiles, u32 num_tiles,
                                                        u8* __restrict__ flags, u32* __restrict__ segsum /*[2][MAX_CHUNKS]*/,
                                                        const u32* __restrict__ grp = nullptr, int grp_shift = 0)
{
    __shared__ u32 scr[8];
    const u32 t = threadIdx.x;
    const u32 tile0 = blockIdx.x * chunk_tiles;
    u32 tile1 = tile0 + chunk_tiles; if (tile1 > num_tiles) tile1 = num_tiles;
    // suffix i is a tail (window crosses the block end) iff i >= tail_lo

    u32 cnt = 0, last1 = 0;
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 j = tile * SEG_TILE + t * SEG_ITEMS;
        if (j >= m) continue;
        u64 k[SEG_ITEMS + 2];          // keys j-1 .. j+8
        u32 s[SEG_ITEMS + 2];
        const bool full = (j + SEG_ITEMS <= m);
        if (full) {
#pragma unroll
            for (int q = 0; q < SEG_ITEMS; q += 2) {
                const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(keys + j + q);
                k[1 + q] = kk.x; k[2 + q] = kk.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SEG_ITEMS; ++q) k[1 + q] = (j + q < m) ? keys[j + q] : 0;
        }
        k[0] = (j > 0) ? keys[j - 1] : 0;
        k[SEG_ITEMS + 1] = (j + SEG_ITEMS < m) ? keys[j + SEG_ITEMS] : 0;
        if (INITIAL || grp != nullptr) {
            // suffix numbers (first seg) / group ranks (text rounds) j-1 .. j+8: the thread's own eight as two 16-byte loads (ten
            // 4-byte loads at a 32-byte lane pitch put every line through the address path eight times)
            const u32* src = INITIAL ? sa : grp;
            const u32 msk = INITIAL ? smask : 0xffffffffu;
            if (full && SEG_ITEMS == 8) {
                const uint4 a = *reinterpret_cast<const uint4*>(src + j), b = *reinterpret_cast<const uint4*>(src + j + 4);
                s[1] = a.x & msk; s[2] = a.y & msk; s[3] = a.z & msk; s[4] = a.w & msk;
                s[5] = b.x & msk; s[6] = b.y & msk; s[7] = b.z & msk; s[8] = b.w & msk;
                s[0] = (j > 0) ? (src[j - 1] & msk) : 0;
                s[9] = (j + 8 < m) ? (src[j + 8] & msk) : 0;
            } else {
#pragma unroll
                for (int q = 0; q < SEG_ITEMS + 2; ++q) {
                    const long long x = (long long)j - 1 + q;
                    s[q] = (x >= 0 && x < (long long)m) ? (src[x] & msk) : 0;
                }
            }
        }
        u32 h[SEG_ITEMS + 1], oh[SEG_ITEMS + 1];
#pragma unroll
        for (int q = 0; q <= SEG_ITEMS; ++q) {
            const u32 x = j + q;
            bool hd, od = false;
            if (x == 0 || x >= m) { hd = true; od = grp_shift > 0; }
            else {
                hd = (k[q + 1] != k[q]);
                if (INITIAL) hd = hd || (s[q + 1] >= tail_lo) || (s[q] >= tail_lo);
                else if (grp != nullptr) hd = hd || (s[q + 1] != s[q]);
                else if (grp_shift > 0) od = (k[q + 1] >> grp_shift) != (k[q] >> grp_shift);
            }
            h[q] = hd ? 1u : 0u; oh[q] = od ? 1u : 0u;
        }
        u64 packed = 0;
#pragma unroll
        for (int q = 0; q < SEG_ITEMS; ++q) {
            const u32 x = j + q;
            if (x < m) {
                const u32 uns = (h[q] & h[q + 1]) ^ 1u;
                packed |= (u64)(h[q] | (uns << 1) | (oh[q] << 2)) << (8 * q);
                cnt += uns;
                if (h[q]) last1 = x + 1;
            }
        }
        if (full) *reinterpret_cast<u64*>(flags + j) = packed;
        else {
#pragma unroll
            for (int q = 0; q < SEG_ITEMS; ++q) if (j + q < m) flags[j + q] = (u8)(packed >> (8 * q));
        }
    }
    u32 tot, mx;
    block_excl_sum(cnt, scr, &tot);
    block_incl_max(last1, scr, &mx);
    if (t == 0) { segsum[blockIdx.x] = tot; segsum[MAX_CHUNKS + blockIdx.x] = mx; }
}

// Geometry of the segmented sorts of the refinement rounds (bwt_round_segsort_kernel / bwt_round_textsort_kernel below): a workgroup owns
// the groups whose head lies in its tile of RS_T records and sees RS_G records past the tile, so a group of up to RS_G + 1 records always
// fits; seg_apply counts the groups that do not (see there).
constexpr int RS_T = 2048, RS_G = 1024, RS_E = RS_T + RS_G;       // 24 KB of LDS per workgroup: 6 workgroups per CU hide the ISA gather
constexpr int DS_NLONG = 5, DS_EXCESS = 6;                        // dscal slots: groups of > RS_G records in the new unsorted set, their records past the first RS_G
// The same two counts for groups of more than HY_G records (hybrid doubling rounds, bwt_longrec_kernel): the segmented sort ranks a record by
// counting inside its group — s comparisons per record of a group of s —, the radix engine costs ~50 ps per record whatever the group.
constexpr int HY_G = 256, DS_NMID = 3, DS_MIDEXCESS = 4;
constexpr int DS_LRTOTAL = 7;                                     // what a hybrid round's extraction really found (must equal the host's count)

// seg phase B: one workgroup scans the <= 1024 chunk summaries.
//   segoff[c]            = exclusive sum of unsorted counts
//   segoff[MAX_CHUNKS+c] = max(last head + 1) over chunks < c
//   dscal[0]             = total unsorted
__global__ __launch_bounds__(WG) void seg_scan_kernel(const u32* __restrict__ segsum, u32 num_chunks,
                                                      u32* __restrict__ segoff, u32* __restrict__ dscal)
{
    __shared__ u32 scr[8];
    __shared__ u32 smax[WG];
    u32 carry = 0, carrymax = 0;
    for (u32 base = 0; base < num_chunks; base += WG) {
        const u32 i = base + threadIdx.x;
        const u32 v  = (i < num_chunks) ? segsum[i] : 0u;
        const u32 mx = (i < num_chunks) ? segsum[MAX_CHUNKS + i] : 0u;
        u32 tot, totmax;
        const u32 ex = block_excl_sum(v, scr, &tot);
        const u32 im = block_incl_max(mx, scr, &totmax);
        __syncthreads();
        smax[threadIdx.x] = im;
        __syncthreads();
        const u32 exmax = (threadIdx.x > 0) ? smax[threadIdx.x - 1] : 0u;
        if (i < num_chunks) {
            segoff[i] = carry + ex;
            segoff[MAX_CHUNKS + i] = (carrymax > exmax) ? carrymax : exmax;
        }
        carry += tot;
        carrymax = (carrymax > totmax) ? carrymax : totmax;
        __syncthreads();
    }
    if (threadIdx.x == 0) { dscal[0] = carry; dscal[DS_NLONG] = 0; dscal[DS_EXCESS] = 0; dscal[DS_NMID] = 0; dscal[DS_MIDEXCESS] = 0; }   // (seg_apply, the next launch, adds to the counters)
}

// seg phase C: ranks (position of the group head), SA / ISA write-back, compaction of unsorted records.
// WRITE_ISA = false (rounds on text keys): nobody reads ISA, the random 4-byte scatter is skipped.
// Also counts, for free, what the next round's segmented sort cannot take: an unsorted record knows its SA slot and its group's rank (the
// slot of the group's head), and members of a group are contiguous in SA — the record RS_G places behind its head proves a group of more
// than RS_G records (one such record per long group: dscal[DS_NLONG]), every record at or beyond that distance is counted in
// dscal[DS_EXCESS].  The host reads both with the size of the unsorted set and never launches a segmented sort that would give up.
template <bool INITIAL, bool WRITE_ISA>
__global__ __launch_bounds__(WG) void seg_apply_kernel(const u8* __restrict__ flags, const u32* sa_sorted,
                                                       const u32* __restrict__ cpos_in, u32 m, u32 smask,
                                                       u32 chunk_tiles, u32 num_tiles, const u32* __restrict__ segoff,
                                                       u32* SA, u32* __restrict__ ISA,
                                                       u32* __restrict__ cpos_out, u32* __restrict__ csa_out,
                                                       u32* __restrict__ cgrp_out, u32* __restrict__ dscal)
{
    __shared__ u32 scr[8];
    __shared__ u32 sprev[WG];
    __shared__ u32 cstage[3 * SEG_TILE];            // SA slot, suffix, group rank of the tile's unsorted records
    __shared__ u32 slong[4];
    const u32 t = threadIdx.x;
    // A single-read digit pass in front of this seg gave up a wait (radix_onesweep.hip): its output is in bounds but not sorted, and the
    // workgroups that stopped claiming tiles left stale records behind — suffix numbers that, masked, can exceed n.  Nothing of that may
    // be scattered into SA / ISA (the host finds the word raised right after this launch and redoes the transform).
    // (this relies on the word being STICKY: only radix_onesweep_check, on the host behind this launch, clears it — dev_common.h)
    if (dscal[OS_ERR_SLOT] != 0u) return;
    if (t < 4) slong[t] = 0;
    u32 n_long = 0, n_excess = 0, n_mid = 0, n_midx = 0;
    const u32 tile0 = blockIdx.x * chunk_tiles;
    u32 tile1 = tile0 + chunk_tiles; if (tile1 > num_tiles) tile1 = num_tiles;
    u32 off   = segoff[blockIdx.x];                 // next compacted slot
    // heads travel as (SA slot + 1) << 1 | oldhead: monotone in the slot, so the running maximum is still "the last head", and its low bit
    // says whether the group it opens keeps the rank its records already have in ISA (seg_reduce: oldhead)
    u32 carry = segoff[MAX_CHUNKS + blockIdx.x];    // (head index + 1) carried in from the left
    if (carry) carry = INITIAL ? carry << 1 : (((cpos_in[carry - 1] + 1) << 1) | ((flags[carry - 1] >> 2) & 1u));   // compacted index -> SA slot of that head

    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 j = tile * SEG_TILE + t * SEG_ITEMS;
        u32 f[SEG_ITEMS], pos[SEG_ITEMS], s[SEG_ITEMS];
        u32 lmax = 0, lcnt = 0;
        if (SEG_ITEMS == 8 && j + SEG_ITEMS <= m) {
            // a thread's eight records: flags as one 8-byte load, suffixes (and SA slots) as two 16-byte loads each — not 8 + 8 (+ 8) narrow ones
            const u64 fw = *reinterpret_cast<const u64*>(flags + j);
            const uint4 s0 = *reinterpret_cast<const uint4*>(sa_sorted + j), s1 = *reinterpret_cast<const uint4*>(sa_sorted + j + 4);
            s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
            if (INITIAL) {
#pragma unroll
                for (int q = 0; q < 8; ++q) pos[q] = j + q;
            } else {
                const uint4 p0 = *reinterpret_cast<const uint4*>(cpos_in + j), p1 = *reinterpret_cast<const uint4*>(cpos_in + j + 4);
                pos[0] = p0.x; pos[1] = p0.y; pos[2] = p0.z; pos[3] = p0.w; pos[4] = p1.x; pos[5] = p1.y; pos[6] = p1.z; pos[7] = p1.w;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f[q] = (u32)(fw >> (8 * q)) & 0xffu;
                if (f[q] & 1u) lmax = ((pos[q] + 1) << 1) | ((f[q] >> 2) & 1u);
                lcnt += (f[q] >> 1) & 1u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SEG_ITEMS; ++q) {
                const u32 x = j + q;
                if (x < m) {
                    f[q]   = flags[x];
                    pos[q] = INITIAL ? x : cpos_in[x];
                    s[q]   = sa_sorted[x];
                    if (f[q] & 1u) lmax = ((pos[q] + 1) << 1) | ((f[q] >> 2) & 1u);
                    lcnt += (f[q] >> 1) & 1u;
                } else { f[q] = 0; pos[q] = 0; s[q] = 0; }
            }
        }
        u32 totmax, totcnt;
        const u32 imax = block_incl_max(lmax, scr, &totmax);
        __syncthreads();
        sprev[t] = imax;
        __syncthreads();
        u32 run = (t > 0) ? sprev[t - 1] : 0u;       // exclusive max over lower threads
        if (carry > run) run = carry;
        u32 kslot = block_excl_sum(lcnt, scr, &totcnt);          // compacted index inside the tile
#pragma unroll
        for (int q = 0; q < SEG_ITEMS; ++q) {
            const u32 x = j + q;
            if (x < m) {
                if (f[q] & 1u) run = ((pos[q] + 1) << 1) | ((f[q] >> 2) & 1u);
                const u32 rank = (run >> 1) - 1;
                if (!INITIAL || SA != sa_sorted) SA[pos[q]] = s[q];          // the first seg may run in place: SA = the sort's value array
                if (WRITE_ISA && (INITIAL || !(run & 1u))) ISA[s[q] & smask] = rank;   // SA / csa keep the predecessor code in their high bits
                if (f[q] & 2u) {
                    cstage[kslot] = pos[q]; cstage[SEG_TILE + kslot] = s[q]; cstage[2 * SEG_TILE + kslot] = rank; ++kslot;
                    const u32 dist = pos[q] - rank;
                    n_long += (u32)(dist == (u32)RS_G); n_excess += (u32)(dist >= (u32)RS_G);
                    n_mid += (u32)(dist == (u32)HY_G); n_midx += (u32)(dist >= (u32)HY_G);
                }
            }
        }
        __syncthreads();
        // the tile's unsorted records leave through LDS: consecutive lanes write consecutive slots of the three arrays
        for (u32 i = t; i < totcnt; i += WG) {
            cpos_out[off + i] = cstage[i];
            csa_out[off + i]  = cstage[SEG_TILE + i];
            cgrp_out[off + i] = cstage[2 * SEG_TILE + i];
        }
        carry = (carry > totmax) ? carry : totmax;
        off += totcnt;
        __syncthreads();
    }
    if (n_midx) {
        atomicAdd(&slong[3], n_midx); if (n_mid) atomicAdd(&slong[2], n_mid);
        if (n_excess) { atomicAdd(&slong[1], n_excess); if (n_long) atomicAdd(&slong[0], n_long); }
    }
    __syncthreads();
    if (t == 0 && slong[3]) {
        atomicAdd(&dscal[DS_MIDEXCESS], slong[3]); if (slong[2]) atomicAdd(&dscal[DS_NMID], slong[2]);
        if (slong[1]) { atomicAdd(&dscal[DS_EXCESS], slong[1]); if (slong[0]) atomicAdd(&dscal[DS_NLONG], slong[0]); }
    }
}

// ---------------------------------------------------------------------------------------------
// doubling round key build: key = rank << 32 | (ISA[sa + h] + 1, 0 when sa + h == n)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void bwt_gather_kernel(const u32* __restrict__ csa, const u32* __restrict__ cgrp,
                                                        const u32* __restrict__ ISA, u32 U, u64 h, u64 n, int lo_bits, u32 smask,
                                                        u64* __restrict__ keys, u32* __restrict__ vals)
{
    const u32 stride = gridDim.x * WG;
    for (u32 k = blockIdx.x * WG + threadIdx.x; k < U; k += stride) {
        const u32 s = csa[k];
        const u64 p = (u64)(s & smask) + h;
        const u32 nxt = (p < n) ? (ISA[p] + 1u) : 0u;
        keys[k] = ((u64)cgrp[k] << lo_bits) | nxt;
        vals[k] = s;
    }
}


// ---------------------------------------------------------------------------------------------
// Doubling round as a SEGMENTED sort (what libcubwt does with cub::DeviceSegmentedSort, libcubwt.cu:1691): the compacted
// records arrive grouped (equal group rank = contiguous, in SA order), so ordering them by (group, next rank) only has to
// reorder each group internally.  One workgroup takes the groups whose head lies in its tile of RS_T records (plus the tail of
// the last one, up to RS_G records past the tile), fetches next = ISA[sa + h] + 1 for them (the gather of the round), and
// ranks every record inside its group by counting (groups are small: 99.9 % of the records of a text block sit in groups of
// <= 1024, three quarters in groups of <= 8; sum of squares ~ 23 comparisons per record).  Output = the same (key, value)
// arrays the radix path produced.  A group longer than RS_G raises `fallback` and the round is redone by the radix engine.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(WG) void bwt_round_segsort_kernel(const u32* __restrict__ csa, const u32* __restrict__ cgrp, const u32* __restrict__ ISA,
                                                               u32 U, u64 h, u64 n, int lo_bits, u32 smask,
                                                               u64* __restrict__ keys_out, u32* __restrict__ vals_out, u32* __restrict__ fallback,
                                                               u32 skip_long = 0u)
{
    __shared__ u32 snext[RS_E];                  // first the group ranks (for the head flags), then the gathered next ranks
    __shared__ short sgs[RS_E];                  // start of the record's group inside the window, -1: it started before the window
    __shared__ short sge[RS_E];                  // indexed by a group's start: one past its last record
    __shared__ u32 scr[8];
    __shared__ u32 sincl[WG];
    const u32 t = threadIdx.x;
    const u64 base = (u64)blockIdx.x * RS_T;
    const u32 ext = (u32)((base + RS_E <= U) ? (u64)RS_E : (U - base));     // records visible to this workgroup
    const u32 own = ext < (u32)RS_T ? ext : (u32)RS_T;                      // groups whose head is below `own` are ours
    const u32 prevg = (base > 0) ? cgrp[base - 1] : 0xffffffffu;            // group ranks are SA slots < n: never 0xffffffff
    for (u32 i = t; i < ext; i += WG) { snext[i] = cgrp[base + i]; sge[i] = (short)ext; }
    __syncthreads();
    // group start of every record = running maximum of head positions; each thread owns PER consecutive records
    constexpr int PER = RS_E / WG;
    const u32 i0 = t * PER;
    u32 hmask = 0;                               // head flags of this thread's records
    int last_head = -1;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 i = i0 + q;
        if (i < ext) { const bool head = (i == 0) ? (snext[0] != prevg) : (snext[i] != snext[i - 1]); if (head) { last_head = (int)i; hmask |= 1u << q; } }
    }
    u32 totmax;
    const u32 incl = block_incl_max((u32)(last_head + 1), scr, &totmax);    // 0 = no head so far
    sincl[t] = incl;
    __syncthreads();
    int run = (t > 0) ? (int)sincl[t - 1] - 1 : -1;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 i = i0 + q;
        if (i < ext) {
            if (hmask & (1u << q)) { if (run >= 0) sge[run] = (short)i; run = (int)i; }      // a head closes the group before it
            sgs[i] = (short)run;
        }
    }
    __syncthreads();
    // a group of ours that is still open at the end of the window (and continues behind it) is too long for this kernel.
    // skip_long (hybrid rounds, bwt_longrec_kernel): groups of more than HY_G records are somebody else's — the radix engine sorts them and
    // bwt_longrec_place_kernel puts them into the same output arrays; this kernel leaves their slots alone.  (An open group of ours has
    // its head below `own` and RS_G records of the window behind the tile: more than RS_G >= HY_G records, skipped like any closed one.)
    if (!skip_long && t == 0 && base + ext < U) {
        const int gs = sgs[ext - 1];
        if (gs >= 0 && (u32)gs < own && cgrp[base + ext] == snext[ext - 1]) atomicOr(fallback, 1u);
    }
    __syncthreads();
    // the round's gather, for the records we own (overwrites the group ranks in LDS; they are re-read from memory at the end)
    for (u32 i = t; i < ext; i += WG) {
        const int gs = sgs[i];
        u32 nx = 0;
        if (gs >= 0 && (u32)gs < own && !(skip_long && (u32)sge[gs] - (u32)gs > (u32)HY_G)) {
            const u64 p = (u64)(csa[base + i] & smask) + h;
            nx = (p < n) ? (ISA[p] + 1u) : 0u;
        }
        snext[i] = nx;
    }
    __syncthreads();
    // rank inside the group by counting (ties keep their order), write to the sorted position
    for (u32 i = t; i < ext; i += WG) {
        const int gs = sgs[i];
        if (gs < 0 || (u32)gs >= own) continue;
        const u32 mine = snext[i];
        const u32 ge = (u32)sge[gs];
        if (skip_long && ge - (u32)gs > (u32)HY_G) continue;
        u32 r = 0;
        for (u32 k = (u32)gs; k < ge; ++k) { const u32 o = snext[k]; r += (u32)((o < mine) || (o == mine && k < i)); }
        const u64 dst = base + (u32)gs + r;
        keys_out[dst] = ((u64)cgrp[base + i] << lo_bits) | mine;
        vals_out[dst] = csa[base + i];
    }
}

// ---------------------------------------------------------------------------------------------
// Hybrid doubling round (round 4): groups of more than HY_G records through the radix engine, everything else through the segmented sort.
// Rounds with a group of more than RS_G records used to send ALL unsorted records through seven radix passes on (group rank, next rank) —
// python sources, 64 MiB: 62 M records in the first doubling round, 18 M of them in 3783 such groups; the four rounds that have them cost
// 17.6 of the block's 25 ms (profiles/r04/bwt_rounds_python-source.txt).  The large groups' records are pulled out in order (count / scan
// / write: a stream compaction), keyed by (head index >> 8, next rank) — heads of groups of more than 256 records lie more than 256
// apart, so the shifted head index still tells the groups apart and keeps their order, in 18 bits instead of 26 —, sorted by the radix
// engine and put back: the j-th sorted record belongs into the slot the j-th extracted record came from (same groups, same sizes, same
// order).  The threshold is HY_G, not RS_G: the segmented sort ranks by counting inside the group, and with the first version's 1024
// the rounds of a block of shared objects (many groups of hundreds of records) got slower, not faster.  "More than HY_G" is what
// seg_apply counted for this unsorted set (the record HY_G places behind the head exists), so the host knows the number of extracted
// records without another round trip: dscal[DS_MIDEXCESS] + dscal[DS_NMID] * HY_G.
// ---------------------------------------------------------------------------------------------
constexpr int LR_ITEMS = 16, LR_TILE = WG * LR_ITEMS;
constexpr int LR_HEAD_SHIFT = 8;                                  // 2^8 <= HY_G
static_assert((1 << LR_HEAD_SHIFT) <= HY_G && HY_G <= RS_G, "heads of the extracted groups must differ after the shift; the segmented sort must be able to take the rest");
// i = index among the unsorted records (a group is contiguous there and in SA: head index = i - (SA slot - group rank))
__device__ __forceinline__ bool bwt_rec_is_long(const u32* __restrict__ cgrp, u32 i, u32 pos, u32 grp, u32 U)
{
    const u32 dist = pos - grp;
    if (dist >= (u32)HY_G) return true;
    const u64 far = (u64)(i - dist) + (u64)HY_G;
    return far < (u64)U && cgrp[far] == grp;
}

// WRITE = false: cnt[chunk] = extracted records of the chunk.  WRITE = true: off[chunk] = where the chunk's extracted records go; writes
// their index (lidx), key and suffix.  Lane-strided records (record = tile base + q * WG + t): loads are coalesced, a wavefront's
// extracted records of one q leave to consecutive slots (ballot prefix), and the 16 x 4 (q, wavefront) counts of a tile are scanned in
// LDS.  (The first version gave every thread 16 consecutive records and let it write its own run of slots: 64 lines per store
// instruction and sixteen dependent csa -> ISA loads in a row per thread; a block of shared objects lost 1.4 ms in its first round.)
template <bool WRITE>
__global__ __launch_bounds__(WG) void bwt_longrec_kernel(const u32* __restrict__ cpos, const u32* __restrict__ csa, const u32* __restrict__ cgrp,
                                                         const u32* __restrict__ ISA, u32 U, u64 h, u64 n, int lo_bits, u32 smask,
                                                         u32 chunk_tiles, u32 num_tiles, u32* __restrict__ cnt_or_off,
                                                         u32* __restrict__ lidx, u64* __restrict__ keys, u32* __restrict__ vals)
{
    __shared__ u32 scr[8];
    __shared__ u32 swc[LR_ITEMS * WAVES];            // extracted records per (q, wavefront) of the tile, then their exclusive sums
    __shared__ u32 stot;
    const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const u32 tile0 = blockIdx.x * chunk_tiles;
    u32 tile1 = tile0 + chunk_tiles; if (tile1 > num_tiles) tile1 = num_tiles;
    u32 off = WRITE ? cnt_or_off[blockIdx.x] : 0u;
    u32 mine_total = 0;
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 base = tile * (u32)LR_TILE;
        u32 pp[LR_ITEMS], gg[LR_ITEMS];
#pragma unroll
        for (int q = 0; q < LR_ITEMS; ++q) {
            const u32 i = base + (u32)q * WG + t;
            const bool in = i < U;
            pp[q] = in ? cpos[i] : 0u; gg[q] = in ? cgrp[i] : 0u;
        }
        u32 mask = 0, pre[LR_ITEMS];
#pragma unroll
        for (int q = 0; q < LR_ITEMS; ++q) {
            const u32 i = base + (u32)q * WG + t;
            const bool lg = i < U && bwt_rec_is_long(cgrp, i, pp[q], gg[q], U);
            const u64 bal = __ballot(lg);
            if (lg) mask |= 1u << q;
            if (WRITE) {
                pre[q] = (u32)__popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) swc[q * WAVES + wave] = (u32)__popcll(bal);
            }
        }
        if (!WRITE) { mine_total += (u32)__popc(mask); continue; }
        __syncthreads();
        if (t < 64) {                                   // LR_ITEMS * WAVES = 64 counts: one wavefront scans them
            const u32 v = swc[t];
            const u32 incl = wave_incl_sum(v);
            swc[t] = incl - v;
            if (t == 63) stot = incl;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LR_ITEMS; ++q) {
            if (mask & (1u << q)) {
                const u32 i = base + (u32)q * WG + t;
                const u32 o = off + swc[q * WAVES + wave] + pre[q];
                const u32 s = csa[i];
                const u64 p = (u64)(s & smask) + h;
                const u32 nxt = (p < n) ? (ISA[p] + 1u) : 0u;
                const u32 head = i - (pp[q] - gg[q]);
                keys[o] = ((u64)(head >> LR_HEAD_SHIFT) << lo_bits) | nxt;
                vals[o] = s;
                lidx[o] = i;
            }
        }
        off += stot;
        __syncthreads();                                // swc / stot are rewritten by the next tile
    }
    if (!WRITE) {
        u32 tot;
        (void)block_excl_sum(mine_total, scr, &tot);
        if (t == 0) cnt_or_off[blockIdx.x] = tot;
    }
}

// one workgroup: off[c] = exclusive sum of cnt[0..c); *total = their sum (the host sized the round's radix sort from seg_apply's counts:
// the two numbers are compared after the round, a difference fails the transform instead of leaving a wrong order behind)
__global__ __launch_bounds__(WG) void bwt_longrec_scan_kernel(const u32* __restrict__ cnt, u32 num_chunks, u32* __restrict__ off, u32* __restrict__ total)
{
    __shared__ u32 scr[8];
    u32 carry = 0;
    for (u32 base = 0; base < num_chunks; base += WG) {
        const u32 i = base + threadIdx.x;
        const u32 v = (i < num_chunks) ? cnt[i] : 0u;
        u32 tot;
        const u32 ex = block_excl_sum(v, scr, &tot);
        if (i < num_chunks) off[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// the j-th sorted long record goes where the j-th extracted one came from, with the key format of the round's other records
__global__ __launch_bounds__(WG) void bwt_longrec_place_kernel(const u64* __restrict__ ks, const u32* __restrict__ vs, const u32* __restrict__ lidx,
                                                               const u32* __restrict__ cgrp, u32 UL, int lo_bits,
                                                               u64* __restrict__ keys_out, u32* __restrict__ vals_out)
{
    const u32 stride = gridDim.x * WG;
    const u64 lomask = (1ull << lo_bits) - 1ull;
    for (u32 j = blockIdx.x * WG + threadIdx.x; j < UL; j += stride) {
        const u32 i = lidx[j];
        keys_out[i] = ((u64)cgrp[i] << lo_bits) | (ks[j] & lomask);
        vals_out[i] = vs[j];
    }
}

// The key of a refinement round on text keys: the packed codes of T[p .. p + a) (zero past the end) and, in the low 4 bits,
// min(n - p, a) — "a proper prefix sorts first" among equal padded keys, 0 for the empty suffix.  lut: byte -> code (LDS or global).
// (two halves so that a caller can have the loads of several records in flight before it packs the first key)
struct TextWin { u32 d[5]; };
__device__ __forceinline__ TextWin bwt_text_window(const u32* __restrict__ T32, u64 p64, u32 n, u32 a = 16)
{
    TextWin w;
    const u32 p = p64 < n ? (u32)p64 : 0u;                // (past the end: any valid address, the key is 0 anyway)
    const u32* q = T32 + (p >> 2);                         // T is 4-byte aligned at T[0] and zero padded for 32 bytes past n
#pragma unroll
    for (int i = 0; i < 4; ++i) w.d[i] = q[i];
    // the fifth word only holds characters 13.. of the window (a <= 12, any text: characters 0..11 + an offset of <= 3 bytes end inside
    // the fourth): one gather in five saved per record; `a` is a kernel argument, the branch is scalar
    w.d[4] = a > 12u ? q[4] : 0u;
    return w;
}
template <class LUT>
__device__ __forceinline__ u64 bwt_text_key_of(const TextWin& w, LUT lut, u64 p64, u32 n, u32 cb, u32 a)
{
    u64 key = 0;
    if (p64 < n) {
        const u32 p = (u32)p64, off = p & 3u;
        const u32 x[4] = { __builtin_amdgcn_alignbyte(w.d[1], w.d[0], off), __builtin_amdgcn_alignbyte(w.d[2], w.d[1], off),
                           __builtin_amdgcn_alignbyte(w.d[3], w.d[2], off), __builtin_amdgcn_alignbyte(w.d[4], w.d[3], off) };
        const u32 left = n - p;                          // characters the suffix has
#pragma unroll
        for (u32 c = 0; c < 15; ++c) {
            if (c < a) {
                const u32 byte = (x[c >> 2] >> (8 * (c & 3))) & 0xffu;
                const u64 code = (c < left) ? (u64)lut[byte] : 0ull;
                key |= code << (64 - cb * (c + 1));
            }
        }
        key |= (u64)(left < a ? left : a);
    }
    return key;
}
template <class LUT>
__device__ __forceinline__ u64 bwt_text_round_key(const u32* __restrict__ T32, LUT lut, u64 p64, u32 n, u32 cb, u32 a)
{
    if (p64 >= n) return 0;
    const TextWin w = bwt_text_window(T32, p64, n, a);
    return bwt_text_key_of(w, lut, p64, n, cb, a);
}

// ---------------------------------------------------------------------------------------------
// A refinement round WITHOUT the inverse suffix array.  Prefix doubling orders the suffixes of a group (equal first h
// characters) by the rank of suffix s + h, which needs ISA — a random 4-byte scatter of n entries after the first sort
// (1.7 ms per 64 MiB block, 5.6 GB of HBM writes for 0.27 GB of payload: the second-largest kernel of the sorter) and a random
// 4-byte gather per unsorted suffix.  But right after the first sort the rank of suffix s + h is just the order of ITS first
// a characters, and those can be read from the text: the round's key is the packed codes of T[s+h .. s+h+a) (zero padded
// past the end) with, in the low 4 bits, min(n - (s+h), a) — "a proper prefix sorts first" among equal padded keys, 0 for the
// empty suffix.  The text is 1/4 of ISA's size and the 16 bytes a record needs are one or two cache lines.  Such a round
// advances the sorted depth by a characters instead of doubling it, so it is used while it pays: text blocks are done after
// two of them (27.8 M unsorted -> 7 155 -> 0 on the bench block) and never build ISA at all; a block that does not converge
// (long repeats) or has a group too long for one workgroup builds ISA once from the current order (bwt_isa_fill / _fix) and
// continues with the doubling rounds below.  Same structure as bwt_round_segsort_kernel; the group rank of a record is not
// part of the output key (cgrp stays valid: records only move inside their group), seg_reduce takes it as a second array.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void bwt_round_textsort_kernel(const u8* __restrict__ T, const u8* __restrict__ codes,
                                                                const u32* __restrict__ csa, const u32* __restrict__ cgrp,
                                                                u32 U, u64 h, u32 n, u32 smask, u32 cb, u32 a,
                                                                u64* __restrict__ keys_out, u32* __restrict__ vals_out, u32* __restrict__ fallback,
                                                                const u8* __restrict__ subhead = nullptr)
{
    // subhead (optional): extra group boundaries inside long groups that bwt_long_* have split by the first characters of this round's
    // key (see there): a record with subhead[k] != 0 starts a group of its own although its group rank equals its left neighbour's
    __shared__ u64 snext[RS_E];                  // first (as u32) the group ranks for the head flags, then the round keys
    __shared__ short sgs[RS_E];
    __shared__ short sge[RS_E];
    __shared__ u32 scr[8];
    __shared__ u32 sincl[WG];
    __shared__ u8 lut[256];
    u32* sgrp = reinterpret_cast<u32*>(snext);
    const u32 t = threadIdx.x;
    lut[t] = codes[t];
    const u64 base = (u64)blockIdx.x * RS_T;
    const u32 ext = (u32)((base + RS_E <= U) ? (u64)RS_E : (U - base));
    const u32 own = ext < (u32)RS_T ? ext : (u32)RS_T;
    const u32 prevg = (base > 0) ? cgrp[base - 1] : 0xffffffffu;
    for (u32 i = t; i < ext; i += WG) { sgrp[i] = cgrp[base + i]; sge[i] = (short)ext; }
    __syncthreads();
    constexpr int PER = RS_E / WG;
    const u32 i0 = t * PER;
    u32 hmask = 0;
    int last_head = -1;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 i = i0 + q;
        if (i < ext) {
            bool head = (i == 0) ? (sgrp[0] != prevg) : (sgrp[i] != sgrp[i - 1]);
            if (subhead != nullptr && subhead[base + i]) head = true;
            if (head) { last_head = (int)i; hmask |= 1u << q; }
        }
    }
    u32 totmax;
    const u32 incl = block_incl_max((u32)(last_head + 1), scr, &totmax);
    sincl[t] = incl;
    __syncthreads();
    int run = (t > 0) ? (int)sincl[t - 1] - 1 : -1;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 i = i0 + q;
        if (i < ext) {
            if (hmask & (1u << q)) { if (run >= 0) sge[run] = (short)i; run = (int)i; }
            sgs[i] = (short)run;
        }
    }
    __syncthreads();
    if (t == 0 && base + ext < U) {
        const int gs = sgs[ext - 1];
        if (gs >= 0 && (u32)gs < own && cgrp[base + ext] == sgrp[ext - 1] && !(subhead != nullptr && subhead[base + ext])) atomicOr(fallback, 1u);
    }
    __syncthreads();
    // the round's keys, from the text (overwrites the group ranks in LDS)
    const u32* T32 = reinterpret_cast<const u32*>(T);
    // (issuing the text loads of four records at a time — 20 loads in flight per lane — was measured: 1.02 ms against 1.00, no gain; the
    // kernel waits on the gathers' sector traffic, 2 GB for 28 M records, not on their latency)
    for (u32 i = t; i < ext; i += WG) {
        const int gs = sgs[i];
        u64 key = 0;
        if (gs >= 0 && (u32)gs < own) key = bwt_text_round_key(T32, lut, (u64)(csa[base + i] & smask) + h, n, cb, a);
        snext[i] = key;
    }
    __syncthreads();
    for (u32 i = t; i < ext; i += WG) {
        const int gs = sgs[i];
        if (gs < 0 || (u32)gs >= own) continue;
        const u64 mine = snext[i];
        const u32 ge = (u32)sge[gs];
        u32 r = 0;
        for (u32 k = (u32)gs; k < ge; ++k) { const u64 o = snext[k]; r += (u32)((o < mine) || (o == mine && k < i)); }
        const u64 dst = base + (u32)gs + r;
        keys_out[dst] = mine;
        vals_out[dst] = csa[base + i];
    }
}

// ---------------------------------------------------------------------------------------------
// Groups longer than one workgroup sorts (RS_G records).  cub::DeviceSegmentedSort (libcubwt.cu:1691) takes segments of any
// length; bwt_round_textsort_kernel does not, and used to hand the WHOLE round over to prefix doubling as soon as one such group
// existed (w = 11 characters on the bench text: 318 long groups, BWT 6.7 -> 14.3 ms).  Now the long groups — and only they — are
// first split by the top LG_BITS bits of this round's key (the first two characters of a text) with a counting sort per group:
//   bwt_long_heads    every record knows where its group starts (its SA slot minus its group rank) and whether the group is long
//                     (the record RS_G places behind the head still has the head's rank); heads of long groups draw a dense id
//   bwt_long_count    histogram of the key's top bits per long group (global atomics: the long groups' records are few)
//   bwt_long_scan     one workgroup per long group: exclusive scan of its 1024 counts, sub-group heads marked at bucket starts
//   bwt_long_scatter  records of long groups move to their bucket (any order inside a bucket: they are about to be sorted by
//                     the full key), all others are copied through
// and the round's segmented sort runs again with the bucket starts as additional group boundaries.  Buckets are almost always short
// enough; if one is not (a long repeat: thousands of suffixes that agree far beyond the key), or there are more than LG_MAX long
// groups, the round is handed over to prefix doubling as before — the right tool for that shape.
// ---------------------------------------------------------------------------------------------
constexpr u32 LG_BITS = 10, LG_BUCKETS = 1u << LG_BITS, LG_MAX = 4096;
struct LongTables { u32* nlong; u32* khead; u32* cnt; };        // [1], [LG_MAX], [LG_MAX][LG_BUCKETS]

__device__ __forceinline__ bool bwt_group_is_long(const u32* __restrict__ cpos, const u32* __restrict__ cgrp, u32 U, u32 k, u32* kh_out)
{
    const u32 g = cgrp[k], kh = k - (cpos[k] - g);                                  // members of a group are contiguous, in SA order
    *kh_out = kh;
    return kh + (u32)RS_G < U && cgrp[kh + RS_G] == g;
}

__global__ __launch_bounds__(WG) void bwt_long_heads_kernel(const u32* __restrict__ cpos, const u32* __restrict__ cgrp, u32 U, LongTables L, u32* __restrict__ lgid)
{
    // heads of long groups draw a dense id (one atomic per long group; how many long groups and records there are the host already knows
    // from seg_apply — a count per wavefront here was 285 K atomics on one address, 3.3 ms on a block with 18 M such records)
    const u32 stride = gridDim.x * WG;
    for (u32 k = blockIdx.x * WG + threadIdx.x; k < U; k += stride) {
        u32 kh = 0;
        if (bwt_group_is_long(cpos, cgrp, U, k, &kh) && kh == k) {
            const u32 id = atomicAdd(L.nlong, 1u);
            lgid[k] = id;
            if (id < LG_MAX) L.khead[id] = k;
        }
    }
}
// The split pays only while the long groups hold a small part of the round's records: their records go through global atomics
// (counting sort per group), and on inputs with long repeats — where most unsorted suffixes sit in a few huge groups — that costs
// far more than the hand-over to prefix doubling it tries to avoid (python sources, 64 MiB: 127 ms against 44 ms).
__device__ __forceinline__ bool bwt_long_split_pays(const LongTables& L, u32 U) { (void)U; return L.nlong[0] != 0u && L.nlong[0] <= LG_MAX; }     // (the host decides; this only guards the tables)

__global__ __launch_bounds__(WG) void bwt_long_count_kernel(const u8* __restrict__ T, const u8* __restrict__ codes, const u32* __restrict__ csa,
                                                            const u32* __restrict__ cpos, const u32* __restrict__ cgrp, u32 U, u64 h, u32 n, u32 smask, u32 cb, u32 a,
                                                            LongTables L, const u32* __restrict__ lgid)
{
    if (!bwt_long_split_pays(L, U)) return;
    const u32* T32 = reinterpret_cast<const u32*>(T);
    const u32 stride = gridDim.x * WG;
    for (u32 k = blockIdx.x * WG + threadIdx.x; k < U; k += stride) {
        u32 kh;
        if (!bwt_group_is_long(cpos, cgrp, U, k, &kh)) continue;
        const u64 key = bwt_text_round_key(T32, codes, (u64)(csa[k] & smask) + h, n, cb, a);
        atomicAdd(&L.cnt[(size_t)lgid[kh] * LG_BUCKETS + (u32)(key >> (64 - LG_BITS))], 1u);
    }
}

__global__ __launch_bounds__(WG) void bwt_long_scan_kernel(LongTables L, u32 U, u8* __restrict__ subhead)
{
    __shared__ u32 scr[8];
    const u32 nl = *L.nlong;
    if (!bwt_long_split_pays(L, U) || blockIdx.x >= nl) return;
    u32* row = L.cnt + (size_t)blockIdx.x * LG_BUCKETS;
    const u32 kh = L.khead[blockIdx.x];
    u32 c[LG_BUCKETS / WG], sum = 0;
#pragma unroll
    for (u32 q = 0; q < LG_BUCKETS / WG; ++q) { c[q] = row[threadIdx.x * (LG_BUCKETS / WG) + q]; sum += c[q]; }
    u32 tot;
    u32 run = block_excl_sum(sum, scr, &tot);
#pragma unroll
    for (u32 q = 0; q < LG_BUCKETS / WG; ++q) {
        row[threadIdx.x * (LG_BUCKETS / WG) + q] = run;                           // becomes the bucket's cursor
        if (c[q]) subhead[kh + run] = 1;
        run += c[q];
    }
}

__global__ __launch_bounds__(WG) void bwt_long_scatter_kernel(const u8* __restrict__ T, const u8* __restrict__ codes, const u32* __restrict__ csa,
                                                              const u32* __restrict__ cpos, const u32* __restrict__ cgrp, u32 U, u64 h, u32 n, u32 smask, u32 cb, u32 a,
                                                              LongTables L, const u32* __restrict__ lgid, u32* __restrict__ csa_out)
{
    if (!bwt_long_split_pays(L, U)) return;
    const u32* T32 = reinterpret_cast<const u32*>(T);
    const u32 stride = gridDim.x * WG;
    for (u32 k = blockIdx.x * WG + threadIdx.x; k < U; k += stride) {
        u32 kh;
        const u32 s = csa[k];
        if (!bwt_group_is_long(cpos, cgrp, U, k, &kh)) { csa_out[k] = s; continue; }
        const u64 key = bwt_text_round_key(T32, codes, (u64)(s & smask) + h, n, cb, a);
        const u32 slot = atomicAdd(&L.cnt[(size_t)lgid[kh] * LG_BUCKETS + (u32)(key >> (64 - LG_BITS))], 1u);
        csa_out[kh + slot] = s;
    }
}

// ISA from the current order (when the text rounds hand over to prefix doubling): every suffix gets its SA slot, then the
// still unsorted ones the rank of their group (the slot of the group's head).
// (uns: the first seg's flags, while they are still there — a slot whose suffix is unsorted is left to bwt_isa_fix_kernel instead of being
// stored twice: with long repeats nearly every slot, and a random 4-byte store costs a whole line of HBM traffic)
__global__ __launch_bounds__(WG) void bwt_isa_fill_kernel(const u32* __restrict__ SA, u32 n, u32 smask, u32* __restrict__ ISA, const u8* __restrict__ uns)
{
    const u32 stride = gridDim.x * WG;
    for (u32 x = blockIdx.x * WG + threadIdx.x; x < n; x += stride) {
        if (uns != nullptr && (uns[x] & 2u)) continue;
        ISA[SA[x] & smask] = x;
    }
}
__global__ __launch_bounds__(WG) void bwt_isa_fix_kernel(const u32* __restrict__ csa, const u32* __restrict__ cgrp, u32 U, u32 smask, u32* __restrict__ ISA)
{
    const u32 stride = gridDim.x * WG;
    for (u32 k = blockIdx.x * WG + threadIdx.x; k < U; k += stride) ISA[csa[k] & smask] = cgrp[k];
}

// Primary index and aux indexes straight from SA (no ISA needed): dscal[1] = j + 1 where SA[j] = 0,
// dscal[8 + t] = j + 1 where SA[j] = t * r (libsais_bwt_aux semantics: I[t] = ISA[t * r] + 1).
__global__ __launch_bounds__(WG) void bwt_find_kernel(const u32* __restrict__ SA, u32 n, u32 smask, u32 rmask, u32 rshift, u32 cnt, u32* __restrict__ dscal)
{
    const u32 j0 = 4u * (blockIdx.x * WG + threadIdx.x);
    if (j0 >= n) return;
    u32 sv[4];
    if (j0 + 4 <= n) { const uint4 q = *reinterpret_cast<const uint4*>(SA + j0); sv[0] = q.x; sv[1] = q.y; sv[2] = q.z; sv[3] = q.w; }
    else { for (u32 q = 0; q < 4; ++q) sv[q] = (j0 + q < n) ? SA[j0 + q] : 0xffffffffu; }
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        if (j0 + q >= n) continue;
        const u32 sfx = sv[q] & smask;
        if (sfx == 0) dscal[1] = j0 + q + 1;
        if (cnt != 0 && (sfx & rmask) == 0 && (sfx >> rshift) < cnt) dscal[8 + (sfx >> rshift)] = j0 + q + 1;
    }
}

// ---------------------------------------------------------------------------------------------
// L[0] = T[n-1]; L[o] = T[SA[j]-1] with j = o-1 for o <= ISA[0], j = o for o > ISA[0].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void bwt_emit_kernel(const u8* __restrict__ T, const u32* __restrict__ SA,
                                                      u32 n, u8* __restrict__ L, const u32* __restrict__ dscal)
{
    const u32 o0 = 4u * (blockIdx.x * WG + threadIdx.x);
    if (o0 >= n) return;
    const u32 p = dscal[1] - 1;                  // ISA[0], from bwt_find_kernel
    u32 word = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const u32 o = o0 + q;
        u32 byte = 0;
        if (o < n) {
            if (o == 0) byte = T[n - 1];
            else {
                const u32 j = (o <= p) ? (o - 1) : o;
                byte = T[SA[j] - 1];
            }
        }
        word |= byte << (8 * q);
    }
    if (o0 + 4 <= n) *reinterpret_cast<u32*>(L + o0) = word;
    else for (u32 q = 0; o0 + q < n; ++q) L[o0 + q] = (u8)(word >> (8 * q));
}

// The same from values that carry the predecessor's code (SA[j] = index | code(T[index-1]) << pred_shift): no access to T at
// all, SA is read in order.  decode[code] = byte.
__global__ __launch_bounds__(WG) void bwt_emit_pred_kernel(const u8* __restrict__ T, const u32* __restrict__ SA, u32 n,
                                                           u32 pred_shift, const u8* __restrict__ decode, u8* __restrict__ L, const u32* __restrict__ dscal)
{
    __shared__ u8 dec[256];
    dec[threadIdx.x] = decode[threadIdx.x];
    __syncthreads();
    const u32 o0 = 4u * (blockIdx.x * WG + threadIdx.x);
    if (o0 >= n) return;
    const u32 p = dscal[1] - 1;                  // ISA[0], from bwt_find_kernel
    u32 word = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const u32 o = o0 + q;
        u32 byte = 0;
        if (o < n) {
            if (o == 0) byte = T[n - 1];
            else { const u32 j = (o <= p) ? (o - 1) : o; byte = dec[SA[j] >> pred_shift]; }
        }
        word |= byte << (8 * q);
    }
    if (o0 + 4 <= n) *reinterpret_cast<u32*>(L + o0) = word;
    else for (u32 q = 0; o0 + q < n; ++q) L[o0 + q] = (u8)(word >> (8 * q));
}

void launch_seg_scan(bscgpu_ctx* c, u32 num_chunks)
{
    hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(WG), 0, c->stream, c->segsum, num_chunks, c->segoff, c->dscal);
}

// ---------------------------------------------------------------------------------------------
static int bit_length(u64 x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

template <bool INITIAL, bool WRITE_ISA>
static int run_seg(bscgpu_ctx* c, const u64* keys, const u32* sa_sorted, const u32* cpos_in, u32 m, u32 tail_lo, u32 smask,
                   u32* cpos_out, u32* csa_out, u32* cgrp_out, u32* U_out, u32* SA, const u32* grp_in = nullptr, int grp_shift = 0)
{
    const Chunking ch = make_chunking(m, SEG_TILE);
    prof_begin(c, BSCGPU_K_SEG, (u64)m * (8 + (INITIAL ? 4 : 0) + 1), m);
    hipLaunchKernelGGL(seg_reduce_kernel<INITIAL>, dim3(ch.num_chunks), dim3(WG), 0, c->stream,
                       keys, sa_sorted, m, tail_lo, smask, ch.chunk_tiles, ch.num_tiles, c->flags, c->segsum, grp_in, grp_shift);
    prof_end(c);
    prof_begin(c, BSCGPU_K_SEG, 0, 0);
    hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(WG), 0, c->stream, c->segsum, ch.num_chunks, c->segoff, c->dscal);
    prof_end(c);
    prof_begin(c, BSCGPU_K_SEG, (u64)m * (1 + 4 + (INITIAL ? 0 : 4) + 4 + 4), m);
    hipLaunchKernelGGL((seg_apply_kernel<INITIAL, WRITE_ISA>), dim3(ch.num_chunks), dim3(WG), 0, c->stream,
                       c->flags, sa_sorted, cpos_in, m, smask, ch.chunk_tiles, ch.num_tiles, c->segoff,
                       SA, c->ISA, cpos_out, csa_out, cgrp_out, c->dscal);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->hscal, c->dscal, (DS_LRTOTAL + 1) * 4, hipMemcpyDeviceToHost, c->stream));  // U, (slots of other kernels), group counts, a hybrid round's extracted records
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    *U_out = c->hscal[0];
    return radix_onesweep_check(c);            // the sort in front of this seg, if it used the single-read passes
}

static int bwt_device_once(bscgpu_ctx* c, const u8* dT_user, u8* dL_user, int64_t n64, int64_t r, u32* I_host, int64_t* primary_out, bool reuse_text);

// A single-read digit pass that gave up a wait (radix_onesweep.hip: bounded polls; pre-emption, a debugger, a hogged CU) fails the
// sort, not the block: the transform is redone once with the three-kernel passes, which have no cross-workgroup protocol at all.
// Every sort is checked (run_seg) before anything is written to the caller's buffer; the private copy of the text is NOT relied on
// for the second attempt (it may have been clobbered by kernels that ran behind the failed sort): the text is copied again.
int bwt_device(bscgpu_ctx* c, const u8* dT_user, u8* dL_user, int64_t n64, int64_t r, u32* I_host, int64_t* primary_out)
{
    c->os_gave_up = false;
    int rc = bwt_device_once(c, dT_user, dL_user, n64, r, I_host, primary_out, false);
    if (rc == BSC_GPU_ERROR && c->os_gave_up) {
        const int mode = c->os_mode;
        c->os_mode = 0; c->os_gave_up = false; ++c->os_retries;
        // (the text is copied again: the caller's buffer is untouched until the last kernel of a transform, while the private copy may
        // not be — kernels that consume a failed sort's output run before the check, on stale keys and values: see seg_apply_kernel)
        rc = bwt_device_once(c, dT_user, dL_user, n64, r, I_host, primary_out, false);
        c->os_mode = mode;
    }
    return rc;
}

static int bwt_device_once(bscgpu_ctx* c, const u8* dT_user, u8* dL_user, int64_t n64, int64_t r, u32* I_host, int64_t* primary_out, bool reuse_text)
{
    if (n64 < 0 || n64 > c->max_n || n64 >= 0x7fffffffll) return BSC_BAD_PARAMETER;
    if (n64 == 0) { *primary_out = 0; return BSC_NO_ERROR; }
    const u32 n = (u32)n64;
    int rc;

    // private, padded copy of the text (emit may overwrite the user's buffer when dL aliases dT)
    if (!reuse_text) HIP_TRY(c, hipMemcpyAsync(c->dT, dT_user, n, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dT + n, 0, 32, c->stream));

    // alphabet: byte histogram -> dense order-preserving codes
    HIP_TRY(c, hipMemsetAsync(c->dscal + 300, 0, 256 * 4, c->stream));
    {
        const Chunking hc = make_chunking(n, 16 * WG * 16);
        prof_begin(c, BSCGPU_K_MISC, n, 0);
        hipLaunchKernelGGL(bwt_bytehist_kernel, dim3(hc.num_chunks), dim3(WG), 0, c->stream, c->dT, n, hc.chunk_tiles * 16u * WG * 16u, c->dscal + 300);
        prof_end(c);
    }
    HIP_TRY(c, hipMemcpyAsync(c->hscal + 300, c->dscal + 300, 256 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    u8 codes[256]; u32 K = 0;
    for (int b = 0; b < 256; ++b) { codes[b] = (u8)K; if (c->hscal[300 + b]) ++K; }
    PackParams pp;
    pp.cb = 1; while ((1u << pp.cb) < K) ++pp.cb;
    if (pp.cb < 4) pp.cb = 4;                                   // at most 16 characters per key
    pp.w = 64 / pp.cb;
    { static const int wmax = [] { const char* e = getenv("BSC_BWT_W"); return e ? atoi(e) : 0; }(); if (wmax >= 2 && (u32)wmax < pp.w) pp.w = (u32)wmax; }   // experiment: shorter first-sort keys
    // Mixed-radix keys (round 6, BSC_BWT_RADIX=1): with K symbols the base-K number of w' characters fits 64 bits for w' = floor(64 / log2 K),
    // which is one character more than 64 / cb for K = 17..19, 24..30 (text: 28 symbols -> 13 characters instead of 12; same eight digit passes).
    // Measured and NOT the default (profiles/r06/first_sort_keys.txt): on the bench block 20.9 M instead of 27.8 M suffixes are unsorted after the
    // first sort and the text round takes 0.78 instead of 1.13 ms — but the bytes of a base-28 number are uniform digits, where the bytes of
    // the 5-bit packing are skewed ones (long runs per bucket), and the eight digit passes take 3.65 instead of 3.07 ms (0.44 against 0.53 of 8 TB/s).
    pp.radix = 0; pp.top = 0;
    {
        static const int radix_on = [] { const char* e = getenv("BSC_BWT_RADIX"); return e ? atoi(e) : BWT_RADIX_DEFAULT; }();
        if (radix_on && K >= 3 && getenv("BSC_BWT_W") == nullptr) {            // (BSC_BWT_W, the other key-width experiment, keeps the bit-packed keys)
            u32 wr = 0; unsigned __int128 pw = 1;
            while (pw * K <= ((unsigned __int128)1 << 64) - 1 && wr < 16) { pw *= K; ++wr; }      // K^wr <= 2^64 - 1
            if (wr > pp.w && wr <= 16) { pp.radix = K; pp.w = wr; u64 t = 1; for (u32 x = 1; x < wr; ++x) t *= K; pp.top = t; }
        }
    }
    pp.tc = n < pp.w - 1 ? n : pp.w - 1;
    pp.low_shift = pp.radix ? 0u : 64 - pp.cb * pp.w;
    // The sort's values have spare high bits when the block is not huge: carry the code of the character in FRONT of the
    // suffix there, so that the final L = T[SA - 1] needs no random gather from the text (BSC_BWT_PRED=0 keeps the gather)
    const int idx_bits = bit_length(n - 1);
    static const int pred_on = [] { const char* e = getenv("BSC_BWT_PRED"); return e ? atoi(e) : 1; }();
    pp.pred_shift = (pred_on && idx_bits >= 1 && idx_bits + (int)pp.cb <= 32) ? (u32)idx_bits : 0u;
    const u32 smask = pp.pred_shift ? ((1u << pp.pred_shift) - 1u) : 0xffffffffu;
    const u32 tail_lo = (n >= pp.w) ? (n - (pp.w - 1)) : 0;
    u8* dcodes = reinterpret_cast<u8*>(c->dscal + 560);         // 256 bytes of the scalar area
    u8* ddecode = reinterpret_cast<u8*>(c->dscal + 720);        // 256 bytes: code -> byte (640..703 is the QLFC front end's symbol table)
    u8 decode[256]; memset(decode, 0, sizeof decode);
    for (int b = 255; b >= 0; --b) if (c->hscal[300 + b]) decode[codes[b]] = (u8)b;
    HIP_TRY(c, hipMemcpyAsync(dcodes, codes, 256, hipMemcpyHostToDevice, c->stream));
    if (pp.pred_shift) HIP_TRY(c, hipMemcpyAsync(ddecode, decode, 256, hipMemcpyHostToDevice, c->stream));
    prof_begin(c, BSCGPU_K_PACK, (u64)n * 13, n);
    hipLaunchKernelGGL(bwt_pack_kernel, dim3((n + 4 * WG - 1) / (4 * WG)), dim3(WG), 0, c->stream,
                       c->dT, n, pp, dcodes, c->kA, c->vA);
    prof_end(c);

    RadixPass passes[16]; int npass = 0;
    constexpr int dbits = 8;    // measured: 7- / 6-bit digits raise per-pass bandwidth (3.5 -> 3.8 / 4.0 TB/s) but the extra passes lose overall
    for (u32 sft = pp.low_shift; sft < 64; sft += dbits) { passes[npass].shift = (int)sft; passes[npass].bits = (64 - sft < (u32)dbits) ? (int)(64 - sft) : dbits; ++npass; }
    int in_alt = 0;
    rc = radix_sort_passes(c, c->kA, c->kB, c->vA, c->vB, n, passes, npass, &in_alt);
    if (rc < 0) return rc;
    const u64* ks = in_alt ? c->kB : c->kA;
    const u32* vs = in_alt ? c->vB : c->vA;

    // Rounds on text keys (bwt_round_textsort_kernel) need a characters + 4 bits in 64: a = w where the first-sort key leaves
    // 4 bits spare, else w - 1.  BSC_BWT_TEXTROUNDS=0 keeps the round-1 flow (ISA built by the first seg, doubling from h = w).
    static const int text_on = [] { const char* e = getenv("BSC_BWT_TEXTROUNDS"); return e ? atoi(e) : 1; }();
    const u32 ta = (pp.cb * pp.w + 4 <= 64) ? pp.w : pp.w - 1;
    bool isa_valid = !(text_on && ta >= 2 && ta <= 15);
    int cur = 0;
    u32 U = 0;
    // SA: the sort's value array itself when the rounds do not write to it (they put their output into kB / vB), else a copy
    u32* SA = (vs == c->vA) ? c->vA : c->SA;
    if (isa_valid) rc = run_seg<true, true >(c, ks, vs, nullptr, n, tail_lo, smask, c->cpos[cur], c->csa[cur], c->cgrp[cur], &U, SA);
    else           rc = run_seg<true, false>(c, ks, vs, nullptr, n, tail_lo, smask, c->cpos[cur], c->csa[cur], c->cgrp[cur], &U, SA);
    if (rc < 0) return rc;

    bool first_flags = true;                    // c->flags still holds the first seg's flags, indexed by SA slot (bit 1: the slot's suffix is unsorted)
    const int lo_bits = bit_length(n);          // next-rank field: values 0 .. n
    const int hi_bits = bit_length(n - 1);      // group rank field: values 0 .. n-1
    u64 h = pp.w;
    int rounds = 0, text_rounds = 0;
    const bool dbg = getenv("BSCGPU_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[bwt] n=%u initial unsorted=%u\n", n, U);
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_round = dbg ? now_ms() : 0.0;                       // debug trace only: wall time of a round (every round ends in a sync)
    auto lap = [&] { const double t = now_ms(), d = t - t_round; t_round = t; return d; };
    while (U > 0) {
        if (++rounds > 60) return ctx_fail(c, BSC_GPU_ERROR, "prefix doubling did not converge", hipSuccess);
        // what the seg that produced this unsorted set counted (seg_apply): groups no segmented sort of this round can take
        const u32 n_long = c->hscal[DS_NLONG], n_long_rec = c->hscal[DS_EXCESS] + c->hscal[DS_NLONG] * (u32)RS_G;
        const u32 n_mid = c->hscal[DS_NMID], n_mid_rec = c->hscal[DS_MIDEXCESS] + c->hscal[DS_NMID] * (u32)HY_G;      // groups of > HY_G records (hybrid rounds)
        if (!isa_valid) {
            // a round on text keys; it hands over to doubling when a group does not fit a workgroup or the round did not pay
            static const int long_split_on = [] { const char* e = getenv("BSC_BWT_LONGSPLIT"); return e ? atoi(e) : 1; }();
            bool round_done = false;
            if (n_long == 0) {
                HIP_TRY(c, hipMemsetAsync(c->dscal + 2, 0, 4, c->stream));
                prof_begin(c, BSCGPU_K_GATHER, (u64)U * (4 + 4 + 16 + 8 + 4), U);
                hipLaunchKernelGGL(bwt_round_textsort_kernel, dim3((U + RS_T - 1) / RS_T), dim3(WG), 0, c->stream,
                                   c->dT, dcodes, c->csa[cur], c->cgrp[cur], U, h, n, smask, pp.cb, ta, c->kB, c->vB, c->dscal + 2);
                prof_end(c);
                HIP_TRY(c, hipMemcpyAsync(c->hscal + 2, c->dscal + 2, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, ctx_sync(c));
                prof_collect(c);
                round_done = c->hscal[2] == 0;
            } else {
                // Groups too long for one workgroup.  While they hold a small part of the round's records they — and only they — are
                // split by the top bits of this round's key and the segmented sort runs with the bucket starts as extra boundaries;
                // otherwise (long repeats: most unsorted suffixes sit in a few huge groups, and their records would all go through
                // global atomics — python sources, 64 MiB: 127 ms against 44 ms) the round is handed over to prefix doubling at once,
                // without first running a sort that gives up (9.5 ms on that block).
                // (tables in an allocation of their own, made when a block first needs them — 16.8 MB per context; round 3 carved them
                // out of kA on the strength of a comment.  Group ids go to ISA, which does not exist yet on this path — isa_valid is
                // false, it is built at the hand-over —, sub-group heads to flags, which only the seg kernels behind the round rewrite;
                // the permuted suffixes to csa[cur ^ 1], the half the next seg writes.)
                constexpr size_t LT_BYTES = (size_t)(64 + LG_MAX + (size_t)LG_MAX * LG_BUCKETS) * 4;
                bool fits = n_long <= LG_MAX && n_long_rec <= U / 8u && long_split_on;
                if (fits && !c->long_tables) {
                    if (hipMalloc((void**)&c->long_tables, LT_BYTES) != hipSuccess) { (void)hipGetLastError(); c->long_tables = nullptr; fits = false; }
                }
                LongTables LT;
                LT.nlong = c->long_tables; LT.khead = LT.nlong + 64; LT.cnt = LT.khead + LG_MAX;
                const bool pays = fits;
                if (pays) {
                    u8* subhead = c->flags;
                    first_flags = false;
                    HIP_TRY(c, hipMemsetAsync(LT.nlong, 0, LT_BYTES, c->stream));
                    HIP_TRY(c, hipMemsetAsync(subhead, 0, (size_t)U + 1, c->stream));
                    HIP_TRY(c, hipMemsetAsync(c->dscal + 2, 0, 4, c->stream));
                    u32 blocks = (U + WG - 1) / WG; if (blocks > 4096) blocks = 4096;
                    prof_begin(c, BSCGPU_K_GATHER, (u64)U * (8 + (8 + 4 + 16 + 4) * 2), U);
                    hipLaunchKernelGGL(bwt_long_heads_kernel, dim3(blocks), dim3(WG), 0, c->stream, c->cpos[cur], c->cgrp[cur], U, LT, c->ISA);
                    hipLaunchKernelGGL(bwt_long_count_kernel, dim3(blocks), dim3(WG), 0, c->stream, c->dT, dcodes, c->csa[cur], c->cpos[cur], c->cgrp[cur],
                                       U, h, n, smask, pp.cb, ta, LT, c->ISA);
                    hipLaunchKernelGGL(bwt_long_scan_kernel, dim3(n_long), dim3(WG), 0, c->stream, LT, U, subhead);
                    hipLaunchKernelGGL(bwt_long_scatter_kernel, dim3(blocks), dim3(WG), 0, c->stream, c->dT, dcodes, c->csa[cur], c->cpos[cur], c->cgrp[cur],
                                       U, h, n, smask, pp.cb, ta, LT, c->ISA, c->csa[cur ^ 1]);
                    hipLaunchKernelGGL(bwt_round_textsort_kernel, dim3((U + RS_T - 1) / RS_T), dim3(WG), 0, c->stream,
                                       c->dT, dcodes, c->csa[cur ^ 1], c->cgrp[cur], U, h, n, smask, pp.cb, ta, c->kB, c->vB, c->dscal + 2, subhead);
                    prof_end(c);
                    HIP_TRY(c, hipMemcpyAsync(c->hscal + 2, c->dscal + 2, 4, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, ctx_sync(c));
                    prof_collect(c);
                    round_done = c->hscal[2] == 0;
                }
                if (dbg) fprintf(stderr, "[bwt] text round %d: %u long group(s) with %u of %u records: %s\n", rounds, n_long, n_long_rec, U,
                                 !pays ? "not split (too many / too large)" : round_done ? "split by the top key bits -> sorted" : "split, but a bucket is still too long");
            }
            if (round_done) {
                u32 U2 = 0;
                first_flags = false;
                rc = run_seg<false, false>(c, c->kB, c->vB, c->cpos[cur], U, 0, smask, c->cpos[cur ^ 1], c->csa[cur ^ 1], c->cgrp[cur ^ 1], &U2, SA, c->cgrp[cur]);
                if (rc < 0) return rc;
                cur ^= 1;
                ++text_rounds;
                h += ta;
                if (dbg) fprintf(stderr, "[bwt] text round %d depth %llu U %u -> %u  (%.2f ms)\n", rounds, (unsigned long long)h, U, U2, lap());
                const bool pays = U2 < 65536u || U2 <= U / 4;
                U = U2;
                if (U == 0 || (pays && text_rounds < 8)) continue;
            } else if (dbg) fprintf(stderr, "[bwt] text round %d: group too long, handing over\n", rounds);
            // hand over: ISA from the current order
            u32 blocks = (n + WG - 1) / WG; if (blocks > 8192) blocks = 8192;
            prof_begin(c, BSCGPU_K_SEG, (u64)n * 8, n);
            hipLaunchKernelGGL(bwt_isa_fill_kernel, dim3(blocks), dim3(WG), 0, c->stream, SA, n, smask, c->ISA, first_flags ? c->flags : (const u8*)nullptr);
            prof_end(c);
            u32 fb = (U + WG - 1) / WG; if (fb > 8192) fb = 8192;
            prof_begin(c, BSCGPU_K_SEG, (u64)U * 12, U);
            hipLaunchKernelGGL(bwt_isa_fix_kernel, dim3(fb), dim3(WG), 0, c->stream, c->csa[cur], c->cgrp[cur], U, smask, c->ISA);
            prof_end(c);
            isa_valid = true;
            if (round_done) continue;                // the round itself was done; the next one doubles
        }
        // segmented sort of the grouped records (falls back to the radix engine when a group is too long for one workgroup)
        static const int segsort_on = [] { const char* e = getenv("BSC_BWT_SEGSORT"); return e ? atoi(e) : 1; }();
        bool sorted = false;
        // A round with a group of > RS_G records is a hybrid round.  BSC_BWT_HYBRID=0: never (the whole round through the radix engine, as in
        // round 3); BSC_BWT_HYBRID_PCT=<p>: only while the groups of > HY_G records hold at most p % of the round's records (shared objects, first
        // rounds: 48 of 57 M records are extracted; same box, 64 MiB: 40.4 ms without, 36.3 at 50 %, 34.8 at 100 % = the default).
        static const int hybrid_on = [] { const char* e = getenv("BSC_BWT_HYBRID"); return e ? atoi(e) : 1; }();
        static const int hybrid_pct = [] { const char* e = getenv("BSC_BWT_HYBRID_PCT"); return e ? atoi(e) : 100; }();
        const bool hybrid = segsort_on && hybrid_on && n_long != 0 && n_mid != 0 && (u64)n_mid_rec * 100ull <= (u64)U * (u64)hybrid_pct;
        if (segsort_on && n_long == 0 && !hybrid) {        // (with a group of > RS_G records the kernel would only find out and give up)
            HIP_TRY(c, hipMemsetAsync(c->dscal + 2, 0, 4, c->stream));
            prof_begin(c, BSCGPU_K_GATHER, (u64)U * (4 + 4 + 4 + 8 + 4), U);
            hipLaunchKernelGGL(bwt_round_segsort_kernel, dim3((U + RS_T - 1) / RS_T), dim3(WG), 0, c->stream,
                               c->csa[cur], c->cgrp[cur], c->ISA, U, h, (u64)n, lo_bits, smask, c->kB, c->vB, c->dscal + 2);
            prof_end(c);
            HIP_TRY(c, hipMemcpyAsync(c->hscal + 2, c->dscal + 2, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, ctx_sync(c));
            prof_collect(c);
            sorted = (c->hscal[2] == 0);
            ks = c->kB; vs = c->vB;
        }
        int np = 0;
        if (!sorted && hybrid) {
            // hybrid round: the long groups' records through the radix engine, the rest through the segmented sort (see bwt_longrec_kernel)
            const u32 UL = n_mid_rec;                               // every record of a group of > HY_G records (seg_apply counted them)
            if (SA == c->vA) {
                HIP_TRY(c, hipMemcpyAsync(c->SA, c->vA, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
                SA = c->SA;
            }
            const Chunking lc = make_chunking(U, LR_TILE);
            u32* lidx = c->cpos[cur ^ 1];                          // the half the seg behind this round writes: free until then
            prof_begin(c, BSCGPU_K_GATHER, (u64)U * 16 + (u64)UL * (4 + 4 + 4 + 8 + 4), U);
            hipLaunchKernelGGL(bwt_longrec_kernel<false>, dim3(lc.num_chunks), dim3(WG), 0, c->stream, c->cpos[cur], c->csa[cur], c->cgrp[cur], c->ISA,
                               U, h, (u64)n, lo_bits, smask, lc.chunk_tiles, lc.num_tiles, c->segsum, lidx, c->kA, c->vA);
            hipLaunchKernelGGL(bwt_longrec_scan_kernel, dim3(1), dim3(WG), 0, c->stream, c->segsum, lc.num_chunks, c->segoff, c->dscal + DS_LRTOTAL);
            hipLaunchKernelGGL(bwt_longrec_kernel<true>, dim3(lc.num_chunks), dim3(WG), 0, c->stream, c->cpos[cur], c->csa[cur], c->cgrp[cur], c->ISA,
                               U, h, (u64)n, lo_bits, smask, lc.chunk_tiles, lc.num_tiles, c->segoff, lidx, c->kA, c->vA);
            prof_end(c);
            RadixPass rp[8];
            const int kbits = lo_bits + bit_length((u64)(U - 1) >> LR_HEAD_SHIFT);      // 27 + 18 at 64 M unsorted records: 6 passes
            for (int s = 0; s < kbits; s += 8) { rp[np].shift = s; rp[np].bits = (kbits - s < 8) ? kbits - s : 8; ++np; }
            rc = radix_sort_passes(c, c->kA, c->kB, c->vA, c->vB, UL, rp, np, &in_alt);
            if (rc < 0) return rc;
            const u64* lks = in_alt ? c->kB : c->kA;
            const u32* lvs = in_alt ? c->vB : c->vA;
            u64* ko = in_alt ? c->kA : c->kB;                       // the pair the sorted long records are NOT in takes the round's output
            u32* vo = in_alt ? c->vA : c->vB;
            prof_begin(c, BSCGPU_K_GATHER, (u64)U * (4 + 4 + 4 + 8 + 4) + (u64)UL * (8 + 4 + 4 + 4 + 8 + 4), U);
            hipLaunchKernelGGL(bwt_round_segsort_kernel, dim3((U + RS_T - 1) / RS_T), dim3(WG), 0, c->stream,
                               c->csa[cur], c->cgrp[cur], c->ISA, U, h, (u64)n, lo_bits, smask, ko, vo, c->dscal + 2, 1u);
            u32 pb = (UL + WG - 1) / WG; if (pb > 8192) pb = 8192;
            hipLaunchKernelGGL(bwt_longrec_place_kernel, dim3(pb), dim3(WG), 0, c->stream, lks, lvs, lidx, c->cgrp[cur], UL, lo_bits, ko, vo);
            prof_end(c);
            HIP_TRY(c, hipGetLastError());
            ks = ko; vs = vo;
            sorted = true;
            if (dbg) fprintf(stderr, "[bwt] round %d: hybrid, %u of %u records in %u group(s) of > 256\n", rounds, UL, U, n_mid);
        }
        if (!sorted) {
            if (SA == c->vA) {                       // the radix engine is about to use vA: SA moves to its own buffer
                HIP_TRY(c, hipMemcpyAsync(c->SA, c->vA, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
                SA = c->SA;
            }
            u32 blocks = (U + WG - 1) / WG; if (blocks > 8192) blocks = 8192;
            prof_begin(c, BSCGPU_K_GATHER, (u64)U * (4 + 4 + 4 + 8 + 4), U);
            hipLaunchKernelGGL(bwt_gather_kernel, dim3(blocks), dim3(WG), 0, c->stream,
                               c->csa[cur], c->cgrp[cur], c->ISA, U, h, (u64)n, lo_bits, smask, c->kA, c->vA);
            prof_end(c);

            RadixPass rp[8];
            const int kbits = lo_bits + hi_bits;            // key = (group rank << lo_bits) | next rank : 53 bits at n = 2^26 -> 7 passes
            for (int s = 0; s < kbits; s += 8) { rp[np].shift = s; rp[np].bits = (kbits - s < 8) ? kbits - s : 8; ++np; }
            rc = radix_sort_passes(c, c->kA, c->kB, c->vA, c->vB, U, rp, np, &in_alt);
            if (rc < 0) return rc;
            ks = in_alt ? c->kB : c->kA;
            vs = in_alt ? c->vB : c->vA;
        }

        u32 U2 = 0;
        // (the keys' group field starts at lo_bits: a new group that begins where an old one did keeps its rank, its ISA stores are skipped)
        static const int isa_skip = [] { const char* e = getenv("BSC_BWT_ISASKIP"); return e ? atoi(e) : 1; }();
        rc = run_seg<false, true>(c, ks, vs, c->cpos[cur], U, 0, smask, c->cpos[cur ^ 1], c->csa[cur ^ 1], c->cgrp[cur ^ 1], &U2, SA, nullptr, isa_skip ? lo_bits : 0);
        if (rc < 0) return rc;
        if (hybrid && c->hscal[DS_LRTOTAL] != n_mid_rec) return ctx_fail(c, BSC_GPU_ERROR, "hybrid round: extracted records differ from the counted ones", hipSuccess);
        cur ^= 1;
        if (dbg) fprintf(stderr, "[bwt] round %d h=%llu U %u -> %u (passes %d)  (%.2f ms)\n", rounds, (unsigned long long)h, U, U2, np, lap());
        U = U2;
        h <<= 1;
    }
    c->stage_ms[5] = rounds;

    // primary index and aux indexes from SA, then the last column
    u32 cnt = 0, rshift = 0;
    if (I_host != nullptr) {
        if (r <= 0 || (r & (r - 1)) != 0) return BSC_BAD_PARAMETER;
        cnt = (u32)((n64 - 1) / r) + 1;
        if (cnt > 256) return BSC_BAD_PARAMETER;
        while ((1ll << rshift) < r) ++rshift;
    }
    prof_begin(c, BSCGPU_K_EMIT, (u64)n * 4, n);
    hipLaunchKernelGGL(bwt_find_kernel, dim3((n + 4 * WG - 1) / (4 * WG)), dim3(WG), 0, c->stream,
                       SA, n, smask, cnt ? (u32)(r - 1) : 0u, rshift, cnt, c->dscal);
    prof_end(c);
    prof_begin(c, BSCGPU_K_EMIT, (u64)n * 6, n);
    if (pp.pred_shift)
        hipLaunchKernelGGL(bwt_emit_pred_kernel, dim3((n + 4 * WG - 1) / (4 * WG)), dim3(WG), 0, c->stream,
                           c->dT, SA, n, pp.pred_shift, ddecode, dL_user, c->dscal);
    else
        hipLaunchKernelGGL(bwt_emit_kernel, dim3((n + 4 * WG - 1) / (4 * WG)), dim3(WG), 0, c->stream,
                           c->dT, SA, n, dL_user, c->dscal);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->hscal, c->dscal, (8 + 256) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    *primary_out = c->hscal[1];
    for (u32 t = 0; t < cnt; ++t) I_host[t] = c->hscal[8 + t];
    return BSC_NO_ERROR;
}
