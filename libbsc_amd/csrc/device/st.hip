// st.hip — Sort Transform of order k = 3..8 on MI355X, plus the device Adler-32.
//
// Result contract (bsc_st_encode, st.cpp:990; verified against the compiled reference):
// stable-sort the positions i in [0,n) by the k cyclic bytes T[i..i+k-1]; output byte T[i-1] (cyclic) in
// that order; return the 0-based sorted rank of position 0.  The reference's GPU path (st.cu:100-217)
// builds the same keys and calls cub::DeviceRadixSort; here the keys go through our own LSD engine:
//   k <= 7 : key = T[i-1] | T[i] .. T[i+6]  (output byte rides in the top byte, never sorted on),
//            keys-only passes over bits [(7-k)*8, 56)  -> k passes of 16 B/record;
//   k == 8 : key = T[i..i+7], value = T[i-1] | (i == 0) << 8, 8 passes of 24 B/record.
#include "dev_common.h"

// cyclic padding around the private text copy: dT[-1] = T[n-1], dT[n+j] = T[j mod n] (j < 32)
__global__ void st_pad_kernel(u8* __restrict__ dT, u32 n)
{
    const u32 j = threadIdx.x;
    if (j < 32) dT[n + j] = dT[j % n];
    if (j == 32) dT[-1] = dT[n - 1];
}

template <bool K8>
__global__ __launch_bounds__(WG) void st_pack_kernel(const u8* __restrict__ T, u32 n, u64* __restrict__ keys,
                                                     u32* __restrict__ vals, u64* __restrict__ key0)
{
    const u32 i0 = 4u * (blockIdx.x * WG + threadIdx.x);
    if (i0 >= n) return;
    const u32* T32 = reinterpret_cast<const u32*>(T + i0) - 1;         // bytes i0-4 .. i0+11
    const u64 w0 = ((u64)__builtin_bswap32(T32[0]) << 32) | __builtin_bswap32(T32[1]);
    const u64 w1 = ((u64)__builtin_bswap32(T32[2]) << 32) | __builtin_bswap32(T32[3]);
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
        const u32 i = i0 + j;
        if (i < n) {
            if (!K8) {
                const u32 s = 8 * (3 + j);                               // window starts at byte i-1
                const u64 key = (w0 << s) | (w1 >> (64 - s));
                keys[i] = key;
                if (i == 0) *key0 = key;
            } else {
                const u32 s = 8 * (4 + j);                               // window starts at byte i
                const u64 key = (s == 32) ? ((w0 << 32) | (w1 >> 32)) : ((w0 << s) | (w1 >> (64 - s)));
                keys[i] = key;
                const u32 prev = (u32)(w0 >> (8 * (4 - j))) & 0xffu;     // byte i-1
                vals[i] = prev | ((i == 0) ? 0x100u : 0u);
            }
        }
    }
}

template <bool K8>
__global__ __launch_bounds__(WG) void st_post_kernel(const u64* __restrict__ keys, const u32* __restrict__ vals,
                                                     u32 n, const u64* __restrict__ key0, u8* __restrict__ out,
                                                     u32* __restrict__ index)
{
    const u32 j0 = 4u * (blockIdx.x * WG + threadIdx.x);
    if (j0 >= n) return;
    const u64 k0 = K8 ? 0 : *key0;
    u32 word = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const u32 j = j0 + q;
        if (j < n) {
            if (!K8) {
                const u64 key = keys[j];
                word |= (u32)(key >> 56) << (8 * q);
                if (key == k0) atomicMin(index, j);
            } else {
                const u32 v = vals[j];
                word |= (v & 0xffu) << (8 * q);
                if (v & 0x100u) atomicMin(index, j);
            }
        }
    }
    if (j0 + 4 <= n) *reinterpret_cast<u32*>(out + j0) = word;
    else for (u32 q = 0; j0 + q < n; ++q) out[j0 + q] = (u8)(word >> (8 * q));
}

static int st_device_once(bscgpu_ctx* c, const u8* dT_user, u8* dOut_user, int n_, int k, int* index_out, bool reuse_text);

// as bwt_device: a single-read pass that gave up fails the sort, and the transform is redone once with the three-kernel passes from
// the private copy of the text (the caller's buffer may already hold the failed attempt's bytes when it is also the output)
int st_device(bscgpu_ctx* c, const u8* dT_user, u8* dOut_user, int n_, int k, int* index_out)
{
    c->os_gave_up = false;
    int rc = st_device_once(c, dT_user, dOut_user, n_, k, index_out, false);
    if (rc == BSC_GPU_ERROR && c->os_gave_up) {
        const int mode = c->os_mode;
        c->os_mode = 0; c->os_gave_up = false; ++c->os_retries;
        rc = st_device_once(c, dT_user, dOut_user, n_, k, index_out, true);
        c->os_mode = mode;
    }
    return rc;
}

static int st_device_once(bscgpu_ctx* c, const u8* dT_user, u8* dOut_user, int n_, int k, int* index_out, bool reuse_text)
{
    if (n_ < 0 || n_ > c->max_n) return BSC_BAD_PARAMETER;
    if (k < 3 || k > 8) return BSC_BAD_PARAMETER;
    const u32 n = (u32)n_;
    if (n <= 1) {                                   // st.cpp:994
        if (n == 1 && dOut_user != dT_user) HIP_TRY(c, hipMemcpyAsync(dOut_user, dT_user, 1, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, ctx_sync(c));
        *index_out = 0;
        return BSC_NO_ERROR;
    }
    if (!reuse_text) HIP_TRY(c, hipMemcpyAsync(c->dT, dT_user, n, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(st_pad_kernel, dim3(1), dim3(64), 0, c->stream, c->dT, n);
    HIP_TRY(c, hipMemsetAsync(c->dscal + 2, 0xff, 4, c->stream));

    const dim3 grid((n + 4 * WG - 1) / (4 * WG));
    int in_alt = 0, rc;
    RadixPass passes[8];
    if (k < 8) {
        prof_begin(c, BSCGPU_K_PACK, (u64)n * 9, n);
        hipLaunchKernelGGL(st_pack_kernel<false>, grid, dim3(WG), 0, c->stream, c->dT, n, c->kA, (u32*)nullptr, c->dscal64);
        prof_end(c);
        for (int p = 0; p < k; ++p) { passes[p].shift = (7 - k) * 8 + 8 * p; passes[p].bits = 8; }
        rc = radix_sort_passes(c, c->kA, c->kB, nullptr, nullptr, n, passes, k, &in_alt);
        if (rc < 0) return rc;
        prof_begin(c, BSCGPU_K_EMIT, (u64)n * 9, n);
        hipLaunchKernelGGL(st_post_kernel<false>, grid, dim3(WG), 0, c->stream, in_alt ? c->kB : c->kA, (const u32*)nullptr,
                           n, c->dscal64, dOut_user, c->dscal + 2);
        prof_end(c);
    } else {
        prof_begin(c, BSCGPU_K_PACK, (u64)n * 13, n);
        hipLaunchKernelGGL(st_pack_kernel<true>, grid, dim3(WG), 0, c->stream, c->dT, n, c->kA, c->vA, c->dscal64);
        prof_end(c);
        for (int p = 0; p < 8; ++p) { passes[p].shift = 8 * p; passes[p].bits = 8; }
        rc = radix_sort_passes(c, c->kA, c->kB, c->vA, c->vB, n, passes, 8, &in_alt);
        if (rc < 0) return rc;
        prof_begin(c, BSCGPU_K_EMIT, (u64)n * 5, n);
        hipLaunchKernelGGL(st_post_kernel<true>, grid, dim3(WG), 0, c->stream, in_alt ? c->kB : c->kA, in_alt ? c->vB : c->vA,
                           n, c->dscal64, dOut_user, c->dscal + 2);
        prof_end(c);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->hscal, c->dscal, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    *index_out = (int)c->hscal[2];
    return radix_onesweep_check(c);
}

// ---------------------------------------------------------------------------------------------
// Adler-32 (adler32.cpp:82-204): s1 = 1 + sum d, s2 = sum of running s1, both mod 65521.
// Each workgroup reduces one contiguous chunk to (a = sum d, b = sum (len - pos) * d); the host
// folds the <= 1024 partials in order: s2 += len * s1 + b ; s1 += a.
// ---------------------------------------------------------------------------------------------
constexpr u32 ADLER_TILE = 16 * WG;   // 4096 bytes per iteration (16-B load per lane)

__device__ __forceinline__ u64 wave_sum_u64(u64 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(WG) void adler_kernel(const u8* __restrict__ T, u64 n, u32 chunk_tiles,
                                                   u64* __restrict__ part /*[chunks][2]*/)
{
    __shared__ u64 red[2 * WAVES];
    const u64 start = (u64)blockIdx.x * chunk_tiles * ADLER_TILE;
    u64 end = start + (u64)chunk_tiles * ADLER_TILE; if (end > n) end = n;
    u64 a = 0, b = 0;
    for (u64 i = start + 16ull * threadIdx.x; i < end; i += ADLER_TILE) {
        if (i + 16 <= end) {
            const uint4 q = *reinterpret_cast<const uint4*>(T + i);
            const u32 w[4] = {q.x, q.y, q.z, q.w};
            u32 s = 0, ws = 0;                       // ws = sum (15 - pos) * d  within the 16 bytes
#pragma unroll
            for (int x = 0; x < 4; ++x) {
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const u32 d = (w[x] >> (8 * y)) & 0xffu;
                    s += d; ws += (u32)(15 - (4 * x + y)) * d;
                }
            }
            a += s;
            b += (u64)s * (end - i - 15) + ws;       // weight of byte at i+p is end - (i+p)
        } else {
            for (u64 p = i; p < end; ++p) { const u32 d = T[p]; a += d; b += (u64)d * (end - p); }
        }
    }
    a = wave_sum_u64(a); b = wave_sum_u64(b);
    const u32 w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[WAVES + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 sa = 0, sb = 0;
        for (int i = 0; i < WAVES; ++i) { sa += red[i]; sb += red[WAVES + i]; }
        part[2 * blockIdx.x]     = sa % 65521ull;
        part[2 * blockIdx.x + 1] = sb % 65521ull;
    }
}

int adler32_device(bscgpu_ctx* c, const u8* d, int64_t n, u32* out)
{
    if (n < 0) return BSC_BAD_PARAMETER;
    if (n == 0) { *out = 1; return BSC_NO_ERROR; }
    if (((uintptr_t)d) & 15) {
        // the kernel reads 16 bytes per lane: an unaligned input (a slice of a caller's tensor) goes through the context's
        // aligned text buffer first (the sorters copy the block there anyway, after this call)
        if (n > c->max_n) return ctx_fail(c, BSC_BAD_PARAMETER, "adler32: unaligned input larger than the context", hipSuccess);
        HIP_TRY(c, hipMemcpyAsync(c->dT, d, (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        d = c->dT;
    }
    const Chunking ch = make_chunking((u64)n, ADLER_TILE);
    prof_begin(c, BSCGPU_K_MISC, (u64)n, 0);
    hipLaunchKernelGGL(adler_kernel, dim3(ch.num_chunks), dim3(WG), 0, c->stream, d, (u64)n, ch.chunk_tiles, c->adler_part);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->hadler, c->adler_part, (size_t)ch.num_chunks * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    u64 s1 = 1, s2 = 0;
    const u64 chunk_bytes = (u64)ch.chunk_tiles * ADLER_TILE;
    for (u32 k = 0; k < ch.num_chunks; ++k) {
        const u64 start = (u64)k * chunk_bytes;
        u64 len = (u64)n - start; if (len > chunk_bytes) len = chunk_bytes;
        s2 = (s2 + (len % 65521ull) * s1 + c->hadler[2 * k + 1]) % 65521ull;
        s1 = (s1 + c->hadler[2 * k]) % 65521ull;
    }
    *out = (u32)(s1 | (s2 << 16));
    return BSC_NO_ERROR;
}
