// coder.cpp — block-level coder: split a sorted block into 1/2/4/8 sub-blocks at run boundaries, code each
// independently (one host thread per sub-block), frame them.  Wire format and split rule are the
// reference's (coder.cpp:52-59 counts, :70-109 split, :111-155 serial framing, :159-240 parallel framing).
#include "qlfc.h"

#include <cstring>
#include <immintrin.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <memory>
#include <thread>
#include "par.h"
#include <vector>

namespace bschost {

enum { FEATURE_MULTITHREADING = 2 };

int coder_num_blocks(int n)
{
    if (n < 256 * 1024)       return 1;
    if (n < 4 * 1024 * 1024)  return 2;
    if (n < 16 * 1024 * 1024) return 4;
    return 8;
}

// Sub-block boundaries: sample every 32nd position (1, 33, 65, ...), count the sampled positions that
// start a run, cut whenever the running count reaches total / nblocks, at that sampled position.
void coder_split_blocks(const uint8_t* in, int n, int nblocks, int* start, int* size)
{
    int changes = 0;
    for (int i = 1; i < n; i += 32) changes += (in[i] != in[i - 1]);

    if (changes > nblocks) {
        const int per_block = changes / nblocks;
        int id = 0, seen = 0;
        start[0] = 0;
        for (int i = 1; i < n && id < nblocks - 1; i += 32) {
            if (in[i] != in[i - 1] && ++seen == per_block) {
                seen = 0;
                size[id] = i - start[id];
                start[++id] = i;
            }
        }
        size[nblocks - 1] = n - start[nblocks - 1];
    } else {
        const int each = n / nblocks;
        for (int p = 0; p < nblocks; ++p) {
            start[p] = each * p;
            size[p]  = (p != nblocks - 1) ? each : n - each * (nblocks - 1);
        }
    }
}

static inline void put_i32(uint8_t* p, int v) { memcpy(p, &v, 4); }
static inline int  get_i32(const uint8_t* p) { int v; memcpy(&v, p, 4); return v; }

static int compress_serial(const uint8_t* in, uint8_t* out, int n, int coder)
{
    const int nblocks = coder_num_blocks(n);
    if (nblocks == 1) {
        const int r = qlfc_encode_block(in, out + 1, n, n - 1, coder);
        if (r < 0) return r;
        out[0] = 1;
        return r + 1;
    }
    int start[8], size[8];
    coder_split_blocks(in, n, nblocks, start, size);
    out[0] = (uint8_t)nblocks;
    int optr = 1 + 8 * nblocks;
    for (int b = 0; b < nblocks; ++b) {
        int room = size[b];
        if (room > n - optr) room = n - optr;
        int r = qlfc_encode_block(in + start[b], out + optr, size[b], room, coder);
        if (r < 0) {                                         // stored raw (coder.cpp:136-140)
            if (optr + size[b] >= n) return NOT_COMPRESSIBLE;
            r = size[b];
            memcpy(out + optr, in + start[b], (size_t)size[b]);
        }
        put_i32(out + 1 + 8 * b, size[b]);
        put_i32(out + 1 + 8 * b + 4, r);
        optr += r;
    }
    return optr;
}

static int compress_parallel(const uint8_t* in, uint8_t* out, int n, int coder)
{
    const int nblocks = coder_num_blocks(n);
    int start[8], size[8], res[8];
    coder_split_blocks(in, n, nblocks, start, size);
    std::unique_ptr<uint8_t, void (*)(void*)> scratch_buf((uint8_t*)bigbuf_get((size_t)n + 64), bigbuf_put);      // (not zeroed, not a fresh mapping per call)
    if (!scratch_buf) return NOT_ENOUGH_MEMORY;
    uint8_t* const scratch = scratch_buf.get();
    run_tasks(nblocks, [&](int b) {
        int r = qlfc_encode_block(in + start[b], scratch + start[b], size[b], size[b], coder);
        res[b] = (r < 0) ? size[b] : r;                     // failed sub-block is stored raw (coder.cpp:194)
    });
    int total = 1 + 8 * nblocks;
    for (int b = 0; b < nblocks; ++b) total += res[b];
    if (total >= n) return NOT_COMPRESSIBLE;

    out[0] = (uint8_t)nblocks;
    int optr = 1 + 8 * nblocks;
    for (int b = 0; b < nblocks; ++b) {
        put_i32(out + 1 + 8 * b, size[b]);
        put_i32(out + 1 + 8 * b + 4, res[b]);
        memcpy(out + optr, (res[b] != size[b] ? scratch : in) + start[b], (size_t)res[b]);
        optr += res[b];
    }
    return total;
}

int coder_compress_views(const RunView* views, int nblocks, const int* start, const int* size, int n,
                         uint8_t* out, int coder, int features, RawFetch& fetch_raw)
{
    if (coder != CODER_STATIC && coder != CODER_ADAPTIVE && coder != CODER_FAST) return BAD_PARAMETER;
    if (nblocks == 1) {
        const int r = qlfc_encode_runs(views[0], n, out + 1, n - 1, coder);
        if (r < 0) return r;
        out[0] = 1;
        return r + 1;
    }
    out[0] = (uint8_t)nblocks;
    if (features & FEATURE_MULTITHREADING) {               // coder.cpp:159-240
        std::vector<std::unique_ptr<uint8_t[]>> scratch((size_t)nblocks);
        int res[8];
        double tms[8];
        run_tasks(nblocks, [&](int b) {
            const auto t0 = std::chrono::steady_clock::now();
            scratch[(size_t)b].reset(new uint8_t[(size_t)size[b] + 64]);        // uninitialised on purpose
            int r = qlfc_encode_runs(views[b], size[b], scratch[(size_t)b].get(), size[b], coder);
            res[b] = (r < 0) ? size[b] : r;
            tms[b] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
        if (getenv("BSCGPU_DEBUG")) { fprintf(stderr, "[coder] sub-block ms:"); for (int b = 0; b < nblocks; ++b) fprintf(stderr, " %.1f", tms[b]); fprintf(stderr, "\n"); }
        int total = 1 + 8 * nblocks;
        for (int b = 0; b < nblocks; ++b) total += res[b];
        if (total >= n) return NOT_COMPRESSIBLE;
        int optr = 1 + 8 * nblocks;
        for (int b = 0; b < nblocks; ++b) {
            put_i32(out + 1 + 8 * b, size[b]);
            put_i32(out + 1 + 8 * b + 4, res[b]);
            if (res[b] != size[b]) memcpy(out + optr, scratch[(size_t)b].get(), (size_t)res[b]);
            else { int rc = fetch_raw(start[b], size[b], out + optr); if (rc < 0) return rc; }
            optr += res[b];
        }
        return total;
    }
    int optr = 1 + 8 * nblocks;                            // coder.cpp:111-155
    for (int b = 0; b < nblocks; ++b) {
        int room = size[b];
        if (room > n - optr) room = n - optr;
        int r = qlfc_encode_runs(views[b], size[b], out + optr, room, coder);
        if (r < 0) {
            if (optr + size[b] >= n) return NOT_COMPRESSIBLE;
            r = size[b];
            int rc = fetch_raw(start[b], size[b], out + optr); if (rc < 0) return rc;
        }
        put_i32(out + 1 + 8 * b, size[b]);
        put_i32(out + 1 + 8 * b + 4, r);
        optr += r;
    }
    return optr;
}

int coder_compress(const uint8_t* in, uint8_t* out, int n, int coder, int features)
{
    if (coder != CODER_STATIC && coder != CODER_ADAPTIVE && coder != CODER_FAST) return BAD_PARAMETER;
    if (coder_num_blocks(n) != 1 && (features & FEATURE_MULTITHREADING)) return compress_parallel(in, out, n, coder);
    return compress_serial(in, out, n, coder);
}

int coder_decompress_bounded(const uint8_t* in, long long in_size, uint8_t* out, int coder, int features, int max_out)
{
    if (coder != CODER_STATIC && coder != CODER_ADAPTIVE && coder != CODER_FAST) return BAD_PARAMETER;
    if (in_size < 1) return DATA_CORRUPT;
    const int nblocks = in[0];
    if (nblocks == 1) return qlfc_decode_block_bounded(in + 1, in_size - 1, out, coder, max_out);
    if (nblocks < 1 || nblocks > 8) return DATA_CORRUPT;   // the format never writes more than 8 (coder.cpp:52-59)
    if (in_size < 1 + 8 * nblocks) return DATA_CORRUPT;    // the frame table itself must lie inside the payload

    int res[8], iptr[8], optr[8], isz[8], osz[8];
    long long ip = 1 + 8 * nblocks, op = 0;
    for (int b = 0; b < nblocks; ++b) {
        osz[b] = get_i32(in + 1 + 8 * b);
        isz[b] = get_i32(in + 1 + 8 * b + 4);
        if (osz[b] < 0 || isz[b] < 0 || op + osz[b] > max_out) return DATA_CORRUPT;
        if (ip + isz[b] > in_size) return DATA_CORRUPT;    // a sub-block may not extend past the payload (checksum-valid corrupt tables)
        iptr[b] = (int)ip; optr[b] = (int)op;
        ip += isz[b]; op += osz[b];
    }
    auto one = [&](int b) {
        if (isz[b] != osz[b]) { res[b] = qlfc_decode_block_bounded(in + iptr[b], isz[b], out + optr[b], coder, osz[b]); if (res[b] >= 0 && res[b] != osz[b]) res[b] = DATA_CORRUPT; }
        else { res[b] = isz[b]; memcpy(out + optr[b], in + iptr[b], (size_t)isz[b]); }
    };
    if (features & FEATURE_MULTITHREADING) {
        run_tasks(nblocks, one);
    } else {
        for (int b = 0; b < nblocks; ++b) one(b);
    }
    int total = 0, err = OK;
    for (int b = 0; b < nblocks; ++b) { if (res[b] < 0) err = res[b]; total += res[b]; }
    return err == OK ? total : err;
}
int coder_decompress(const uint8_t* in, uint8_t* out, int coder, int features) { return coder_decompress_bounded(in, UNBOUNDED_INPUT, out, coder, features, 0x7fffffff); }

// Adler-32 (adler32.cpp:82-204): s1 = 1 + sum, s2 = sum of s1, mod 65521; deferred modulo every 5552 bytes.
// 32 bytes per step with AVX2: s2 += 32 * s1 + sum (32 - i) * d[i]  (pmaddubsw against the weights 32..1), s1 += sum d[i].
uint32_t adler32(const uint8_t* p, size_t n)
{
    uint32_t s1 = 1, s2 = 0;
    const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17,
                                             16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m256i ones16 = _mm256_set1_epi16(1);
    while (n > 0) {
        size_t k = n < 5536 ? n : 5536;                       // multiple of 32 below the 5552 overflow bound
        n -= k;
        if (k >= 32) {
            __m256i vs1 = _mm256_setzero_si256(), vs2 = _mm256_setzero_si256(), vacc = _mm256_setzero_si256();   // vacc: sum of s1 before each step
            const size_t steps = k / 32;
            for (size_t i = 0; i < steps; ++i, p += 32) {
                const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
                vacc = _mm256_add_epi32(vacc, vs1);
                vs1 = _mm256_add_epi32(vs1, _mm256_sad_epu8(d, _mm256_setzero_si256()));            // 4 partial byte sums (u64 lanes, low 32 bits used)
                vs2 = _mm256_add_epi32(vs2, _mm256_madd_epi16(_mm256_maddubs_epi16(d, weights), ones16));
            }
            k -= steps * 32;
            auto hsum = [](__m256i v) { __m128i x = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
                                        x = _mm_add_epi32(x, _mm_shuffle_epi32(x, 0x4e)); x = _mm_add_epi32(x, _mm_shuffle_epi32(x, 0xb1)); return (uint32_t)_mm_cvtsi128_si32(x); };
            const uint32_t sum_d = hsum(vs1), sum_w = hsum(vs2), sum_prev = hsum(vacc);
            s2 += (uint32_t)(steps * 32) * s1 + 32u * sum_prev + sum_w;      // every step adds 32 * (s1 before it) + its weighted bytes
            s1 += sum_d;
        }
        while (k--) { s1 += *p++; s2 += s1; }
        s1 %= 65521u; s2 %= 65521u;
    }
    return s1 | (s2 << 16);
}

}  // namespace bschost
