// unbwt.hip — inverse Burrows-Wheeler transform on MI355X (the reference's GPU hook: libcubwt_unbwt, libcubwt.cu:2953,
// reached from bsc_bwt_decode, bwt.cpp:233-281).
//
// Contract (bwt.cpp:283-334 via libsais_unbwt_aux): L[0..n) and the 1-based primary index `idx` as bsc_bwt_encode writes them;
// rows 0..n, row `idx` is the sentinel row; sym(row) = L[row] below idx, L[row-1] above; the text is read off backwards by
// row <- LF(row) from row 0 and the walk must close on row idx after n steps.
//
// The walk is one serial chain of n dependent random accesses — hopeless on one lane.  As the reference's GPU path does, it is
// cut into many independent pieces (this implementation is our own):
//   1. LF      the stable counting-sort position of every row's symbol = ONE keys-only pass of the radix engine with
//              destination positions (rs_scatter<EMIT_POS>): LF(row) = pos + 1; packed with the symbol into P[row] (8 B), so
//              a step is one random 8-byte load.
//   2. marks   S ~ n/128 rows are marked (one per stratum, hashed offset; row 0 and the sentinel row included); their P entry
//              is replaced by {MARK, segment id}, the original kept in the segment table.
//   3. survey  one lane per segment walks until it arrives at a marked row: length and successor segment.
//   4. order   the host follows the S successor links from segment 0 (row 0 = text end) and turns lengths into offsets
//              (a 4 MB table; 512 K dependent steps in L2 ~ 2 ms) — and checks that the links form ONE chain of n steps that
//              ends on the sentinel row: anything else is a corrupt block, reported, never walked blindly.
//   5. decode  the same walks again, every lane writing its bytes straight to their final place.
// Latency is hidden by the number of concurrent walks (S lanes, mean length 128), not by any single one being fast.
#include "dev_common.h"
#include <cstring>
#include <vector>

constexpr u64 UB_MARK = 1ull << 63;
constexpr u32 UB_END  = 0xffffffffu;            // successor of the segment that runs into the sentinel row
constexpr u32 UB_STEP_CAP = 1u << 22;           // a walk longer than this is a cycle without marks: corrupt input

struct UbSeg { u64 orig; u32 row; u32 pad; };   // original P entry of the marked row, and the row

__device__ __forceinline__ u32 ub_hash(u32 x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// keys for the LF pass: one u64 per non-sentinel row, in row order
__global__ __launch_bounds__(WG) void ub_keys_kernel(const u8* __restrict__ L, u32 n, u64* __restrict__ keys)
{
    const u32 i = blockIdx.x * WG + threadIdx.x;
    if (i < n) keys[i] = (u64)L[i];
}

// P[row] = LF(row) | sym(row) << 32; the sentinel row gets LF 0 and an END mark
__global__ __launch_bounds__(WG) void ub_pack_kernel(const u8* __restrict__ L, const u32* __restrict__ pos, u32 n, u32 idx, u64* __restrict__ P)
{
    const u32 row = blockIdx.x * WG + threadIdx.x;
    if (row > n) return;
    if (row == idx) { P[row] = UB_MARK | (u64)UB_END; return; }
    const u32 i = row < idx ? row : row - 1;
    P[row] = (u64)(pos[i] + 1u) | ((u64)L[i] << 32);
}

__device__ __forceinline__ u32 ub_seg_row(u32 s, u32 S, u32 rows)
{
    // stratum s of the rows 0..rows-1; segment 0 starts at row 0 (the walk's start)
    const u32 q = rows / S, r = rows % S;
    const u32 lo = s * q + (s < r ? s : r), len = q + (s < r ? 1u : 0u);
    return s == 0 ? 0u : lo + ub_hash(s * 2654435761u + rows) % len;
}

__global__ __launch_bounds__(WG) void ub_mark_kernel(u64* __restrict__ P, u32 S, u32 rows, UbSeg* __restrict__ seg)
{
    const u32 s = blockIdx.x * WG + threadIdx.x;
    if (s >= S) return;
    const u32 row = ub_seg_row(s, S, rows);
    const u64 p = P[row];
    seg[s].orig = p; seg[s].row = row; seg[s].pad = 0;
    if (!(p & UB_MARK)) P[row] = UB_MARK | (u64)s;                    // the sentinel row keeps its END mark
}

// one lane per segment: DECODE = false: length + successor; true: bytes to out[k], k descending from `start`
template <bool DECODE>
__global__ __launch_bounds__(WG) void ub_walk_kernel(const u64* __restrict__ P, const UbSeg* __restrict__ seg, u32 S,
                                                     u32* __restrict__ seg_len, u32* __restrict__ seg_next,
                                                     const u32* __restrict__ seg_start, u8* __restrict__ out, u32* __restrict__ bad)
{
    const u32 s = blockIdx.x * WG + threadIdx.x;
    if (s >= S) return;
    u64 p = seg[s].orig;
    if (p & UB_MARK) {                                                // the segment placed on the sentinel row: empty
        if (!DECODE) { seg_len[s] = 0; seg_next[s] = UB_END; }
        return;
    }
    long long k = DECODE ? (long long)seg_start[s] : 0;
    u32 len = 0, nxt = UB_END;
    for (;;) {
        if (DECODE) { if (k < 0) { atomicOr(bad, 1u); break; } out[k--] = (u8)(p >> 32); }
        ++len;
        const u64 q = P[(u32)p];
        if (q & UB_MARK) { nxt = (u32)q; break; }
        p = q;
        if (len >= UB_STEP_CAP) { atomicOr(bad, 2u); break; }
    }
    if (!DECODE) { seg_len[s] = len; seg_next[s] = nxt; }
}

// Host-pointer entry: L (n bytes) -> T (n bytes), both host memory; returns BSC_NO_ERROR, a libbsc error code,
// LIBBSC_DATA_CORRUPT (-6) when the rows do not form one cycle through the sentinel row, or LIBBSC_NOT_SUPPORTED (-4) when a
// piece of the cycle is longer than the walk kernel's step cap (T untouched: use the host walk).
extern "C" int bscgpu_unbwt(bscgpu_ctx* c, const uint8_t* L, uint8_t* T, int64_t n64, int64_t index)
{
    if (!c || !L || !T || n64 < 0 || index <= 0 || index > n64) return BSC_BAD_PARAMETER;
    if (n64 > c->max_n || n64 >= 0x7ffffff0ll) return BSC_GPU_NOT_ENOUGH_MEMORY;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    if (n64 == 0) return BSC_NO_ERROR;
    const u32 n = (u32)n64, idx = (u32)index, rows = n + 1;
    u8* dL = c->dT;                                 // input bytes
    u64* keys = c->kA; u64* P = c->kB;              // the pass's sorted keys land in kB and are overwritten by P afterwards
    u32* pos = c->vA;
    u32 S = rows / 128; if (S > (1u << 19)) S = 1u << 19; if (S < 1) S = 1;
    UbSeg* seg = reinterpret_cast<UbSeg*>(c->cpos[0]);             // 16 B x S <= 8 MB (cpos holds 4N bytes)
    u32* seg_len = c->csa[0]; u32* seg_next = c->csa[1]; u32* seg_start = c->cgrp[0];
    if ((size_t)S * 16 > (size_t)c->max_n * 4 + 4096 * 4) { S = (u32)(((size_t)c->max_n * 4) / 16); if (S < 1) S = 1; }

    HIP_TRY(c, hipMemcpyAsync(dL, L, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dscal + 4, 0, 4, c->stream));
    prof_begin(c, BSCGPU_K_PACK, (u64)n * 9, n);
    hipLaunchKernelGGL(ub_keys_kernel, dim3((n + WG - 1) / WG), dim3(WG), 0, c->stream, dL, n, keys);
    prof_end(c);
    RadixPass low; low.shift = 0; low.bits = 8;
    int in_alt = 0;
    int rc = radix_sort_passes(c, keys, P, nullptr, nullptr, n, &low, 1, &in_alt, pos);
    if (rc < 0) return rc;
    prof_begin(c, BSCGPU_K_PACK, (u64)rows * 13, rows);
    hipLaunchKernelGGL(ub_pack_kernel, dim3((rows + WG - 1) / WG), dim3(WG), 0, c->stream, dL, pos, n, idx, P);
    hipLaunchKernelGGL(ub_mark_kernel, dim3((S + WG - 1) / WG), dim3(WG), 0, c->stream, P, S, rows, seg);
    prof_end(c);
    prof_begin(c, BSCGPU_K_GATHER, (u64)n * 8, n);
    hipLaunchKernelGGL(ub_walk_kernel<false>, dim3((S + WG - 1) / WG), dim3(WG), 0, c->stream, P, seg, S, seg_len, seg_next,
                       (const u32*)nullptr, (u8*)nullptr, c->dscal + 4);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    std::vector<u32> hlen(S), hnext(S), hstart(S, 0u);
    HIP_TRY(c, hipMemcpyAsync(hlen.data(), seg_len, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hnext.data(), seg_next, (size_t)S * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->hscal + 4, c->dscal + 4, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    // A walk longer than UB_STEP_CAP is not proof of corruption — the marked rows are a fixed hash of the row number, and a valid
    // but adversarial block can put more than the cap between two of them —, so it is "not handled here" (the caller's host walk
    // decides, T is still untouched); LIBBSC_DATA_CORRUPT is reserved for the chain-closure checks below.
    if (c->hscal[4] & 2u) return BSC_NOT_SUPPORTED;
    if (c->hscal[4] != 0) return -6;
    // order of the segments: from segment 0 (row 0 = the text's end) along the successor links; every non-empty segment must
    // be met exactly once, the lengths must add up to n and the last link must be the sentinel row
    {
        u64 done = 0; u32 s = 0, met = 0;
        std::vector<u8> seen(S, 0);
        for (;;) {
            if (seen[s]) return -6;
            seen[s] = 1; ++met;
            if (done + hlen[s] > n) return -6;
            hstart[s] = (u32)(n - 1 - done);                           // first byte this segment writes (descending)
            done += hlen[s];
            const u32 nx = hnext[s];
            if (nx == UB_END) break;
            if (nx >= S) return -6;
            s = nx;
        }
        if (done != n) return -6;
        (void)met;
    }
    HIP_TRY(c, hipMemcpyAsync(seg_start, hstart.data(), (size_t)S * 4, hipMemcpyHostToDevice, c->stream));
    prof_begin(c, BSCGPU_K_GATHER, (u64)n * 9, n);
    hipLaunchKernelGGL(ub_walk_kernel<true>, dim3((S + WG - 1) / WG), dim3(WG), 0, c->stream, P, seg, S, (u32*)nullptr, (u32*)nullptr,
                       seg_start, c->dL, c->dscal + 4);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(T, c->dL, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->hscal + 4, c->dscal + 4, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    return c->hscal[4] != 0 ? -6 : BSC_NO_ERROR;
}
