// decode.cpp — the decode side of the API: bsc_decompress, bsc_bwt_decode.
//
// Container checks follow libbsc.cpp:522-617 (bsc_decompress) and :420-519 (in-place twin).  The inverse BWT is a
// plain host LF-mapping walk (the reference uses libsais_unbwt[_aux], bwt.cpp:283-334; same result, no auxiliary
// index parallelism yet).  The inverse ST (reference: st.cpp:1014-1527) is our own algorithm from the definition.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include "par.h"
#include <algorithm>
#include <cstdint>
#include <memory>

#include "../../../include/libbsc.h"
#include "qlfc.h"
#include "lzp.h"

using namespace bschost;

static inline int get_i32(const unsigned char* p) { int v; memcpy(&v, p, 4); return v; }

// ---- LZP decoding (lzp.cpp:564-676 block decoder, :813-885 framing) -------------------------------------------
// The stream is literal bytes; wherever the 4-byte context has been seen before (hash table of last positions), an
// 0xF2 byte is an escape: 0xF2 0xFF = a literal 0xF2, otherwise 0xF2 l1 l2 .. = copy (minLen + sum of the length bytes,
// a length byte of 254 continues) bytes from the predicted position.
static int lzp_decode_block(const unsigned char* in, const unsigned char* in_end, unsigned char* out, int out_cap, int hashSize, int minLen)
{
    if (in_end - in < 4) return LIBBSC_UNEXPECTED_EOB;
    if (out_cap < 4) return LIBBSC_DATA_CORRUPT;
    std::vector<int> lookup((size_t)1 << hashSize, 0);
    const unsigned mask = (1u << hashSize) - 1u;
    unsigned char* const out0 = out;
    unsigned char* const out_end = out + out_cap;
    for (int i = 0; i < 4; ++i) *out++ = *in++;
    unsigned ctx = out[-1] | (out[-2] << 8) | (out[-3] << 16) | ((unsigned)out[-4] << 24);
    while (in < in_end) {
        const unsigned idx = ((ctx >> 15) ^ ctx ^ (ctx >> 3)) & mask;
        const int value = lookup[idx];
        lookup[idx] = (int)(out - out0);
        if (*in == 0xf2 && value > 0) {
            if (++in >= in_end) return LIBBSC_DATA_CORRUPT;
            if (*in != 255) {
                long long len = minLen;
                for (;;) { const unsigned char l = *in++; len += l; if (l != 254) break; if (in >= in_end) return LIBBSC_DATA_CORRUPT; }
                if (len > out_end - out) return LIBBSC_DATA_CORRUPT;
                const unsigned char* ref = out0 + value;
                for (long long k = 0; k < len; ++k) *out++ = *ref++;          // may overlap forward, byte by byte
                ctx = out[-1] | (out[-2] << 8) | (out[-3] << 16) | ((unsigned)out[-4] << 24);
            } else {
                ++in;
                if (out >= out_end) return LIBBSC_DATA_CORRUPT;
                *out++ = 0xf2; ctx = (ctx << 8) | 0xf2;
            }
        } else {
            if (out >= out_end) return LIBBSC_DATA_CORRUPT;
            const unsigned char c = *in++;
            *out++ = c; ctx = (ctx << 8) | c;
        }
    }
    return (int)(out - out0);
}

namespace bschost {
int lzp_decompress(const uint8_t* in, uint8_t* out, int n, int out_cap, int hashSize, int minLen)
{
    if (n < 1) return LIBBSC_DATA_CORRUPT;
    const int nblocks = in[0];
    if (nblocks == 1) return lzp_decode_block(in + 1, in + n, out, out_cap, hashSize, minLen);
    if (nblocks < 1 || nblocks > 8 || n < 1 + 8 * nblocks) return LIBBSC_DATA_CORRUPT;
    long long ip = 1 + 8 * nblocks, op = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int osz = get_i32(in + 1 + 8 * b), isz = get_i32(in + 1 + 8 * b + 4);
        if (osz < 0 || isz < 0 || ip + isz > n || op + osz > out_cap) return LIBBSC_DATA_CORRUPT;
        int r;
        if (isz != osz) r = lzp_decode_block(in + ip, in + ip + isz, out + op, osz, hashSize, minLen);
        else { r = isz; memcpy(out + op, in + ip, (size_t)isz); }
        if (r < 0) return r;
        if (r != osz) return LIBBSC_DATA_CORRUPT;
        ip += isz; op += osz;
    }
    return (int)op;
}
}  // namespace bschost

extern "C" {

// L = [T[n-1]] ++ [T[SA[j]-1] : SA[j] != 0]; `index` (1-based) is where the end-of-text row was removed.
// Rows 0..n: row 0 is the empty suffix, row j >= 1 is suffix SA[j-1]; the row of suffix 0 (`index`) carries the sentinel.
// One counting pass builds P[row] = LF(row) | symbol << 32, so a backward step is a single random access.  The aux
// indexes written by the encoder (bwt.cpp:192-209: the row of every suffix t * r) cut the text into num_indexes + 1
// independent backward walks; they run interleaved (several cache misses in flight per core) and, with
// LIBBSC_FEATURE_MULTITHREADING, on several threads.  The reference reaches the same result through
// libsais_unbwt_aux (bwt.cpp:283-334).
static int aux_rate_of(int n)          // largest power of two <= n / 8 (>= 1), bwt.cpp:192-197
{
    int mod = n / 8;
    mod |= mod >> 1; mod |= mod >> 2; mod |= mod >> 4; mod |= mod >> 8; mod |= mod >> 16; mod >>= 1;
    return mod + 1;
}

int bsc_bwt_decode_gpu(unsigned char* T, int n, int index);            // block.cpp: the default GPU context's bscgpu_unbwt

int bsc_bwt_decode(unsigned char* T, int n, int index, unsigned char num_indexes, int* indexes, int features)
{
    if (T == nullptr || n < 0 || index <= 0 || index > n) return LIBBSC_BAD_PARAMETER;     // bwt.cpp:285
    if (n <= 1) return LIBBSC_NO_ERROR;
    // large blocks: the GPU walks n/128 pieces of the LF cycle at once (unbwt.hip; the reference's hook is libcubwt_unbwt,
    // bwt.cpp:233-281).  Like the reference, the CPU walk below remains for small blocks and for machines without a GPU.
    static const int gpu_min = [] { const char* e = getenv("BSC_GPU_UNBWT_MIN_N"); return e ? atoi(e) : (1 << 20); }();
    if (gpu_min > 0 && n >= gpu_min) {
        const int r = bsc_bwt_decode_gpu(T, n, index);
        if (r == LIBBSC_NO_ERROR || r == LIBBSC_DATA_CORRUPT) return r;
    }
    const size_t N = (size_t)n;
    // 8 bytes per row, written once below: no zero fill
    struct FreeRaw { void operator()(void* q) const { free(q); } };
    std::unique_ptr<void, FreeRaw> pown(malloc((N + 1) * sizeof(uint64_t)));
    if (!pown) return LIBBSC_NOT_ENOUGH_MEMORY;
    uint64_t* const P = static_cast<uint64_t*>(pown.get());
    {
        unsigned cnt[256] = {0};
        for (size_t i = 0; i < N; ++i) cnt[T[i]]++;
        unsigned base[256], sum = 1;                          // row 0 is the sentinel-first suffix
        for (int c = 0; c < 256; ++c) { base[c] = sum; sum += cnt[c]; }
        const size_t idx = (size_t)index;
        for (size_t i = 0; i < idx; ++i) { const unsigned char c = T[i]; P[i] = (uint64_t)base[c]++ | ((uint64_t)c << 32); }
        P[idx] = 0;
        for (size_t i = idx + 1; i <= N; ++i) { const unsigned char c = T[i - 1]; P[i] = (uint64_t)base[c]++ | ((uint64_t)c << 32); }
    }

    // walks: chain s rebuilds text positions [s * r, min((s + 1) * r, n)) backwards from the row of its end position
    int chains = 1;
    const int r = aux_rate_of(n);
    if (num_indexes > 0 && indexes != nullptr && (int)num_indexes == (n - 1) / r) {
        chains = (int)num_indexes + 1;
        for (int t = 0; t < (int)num_indexes; ++t) if (indexes[t] < 0 || indexes[t] >= n) return LIBBSC_DATA_CORRUPT;
    }
    struct Chain { uint32_t row; long long k, stop; uint32_t expect; };
    std::vector<Chain> ch((size_t)chains);
    for (int s2 = 0; s2 < chains; ++s2) {
        Chain& c = ch[(size_t)s2];
        const long long lo = (long long)s2 * r, hi = (s2 == chains - 1) ? (long long)n : (long long)(s2 + 1) * r;
        c.k = hi - 1; c.stop = lo;
        c.row = (s2 == chains - 1) ? 0u : (uint32_t)indexes[s2] + 1u;
        c.expect = (s2 == 0) ? (uint32_t)index : (uint32_t)indexes[s2 - 1] + 1u;
    }
    std::vector<unsigned char> out(N);
    const uint64_t* Pp = P;
    unsigned char* op = out.data();
    auto walk = [&](int first, int step) {                   // chains first, first + step, ... interleaved on one thread
        Chain* mine[256]; int m = 0;
        for (int s2 = first; s2 < chains; s2 += step) mine[m++] = &ch[(size_t)s2];
        long long longest = 0;
        for (int i = 0; i < m; ++i) longest = std::max(longest, mine[i]->k - mine[i]->stop + 1);
        for (long long it = 0; it < longest; ++it)
            for (int i = 0; i < m; ++i) {
                Chain& c = *mine[i];
                if (c.k >= c.stop) { const uint64_t p = Pp[c.row]; op[c.k--] = (unsigned char)(p >> 32); c.row = (uint32_t)p; }
            }
    };
    int threads = 1;
    if ((features & LIBBSC_FEATURE_MULTITHREADING) && chains > 1 && n >= (1 << 20)) threads = chains < 8 ? chains : 8;
    if (threads == 1) walk(0, 1);
    else bschost::run_tasks(threads, [&](int t) { walk(t, threads); });
    for (int s2 = 0; s2 < chains; ++s2) if (ch[(size_t)s2].row != ch[(size_t)s2].expect) return LIBBSC_DATA_CORRUPT;   // inconsistent indexes
    memcpy(T, out.data(), N);
    return LIBBSC_NO_ERROR;
}

// Inverse Sort Transform of order k (contract: st.cpp:1491-1527; algorithm ours, from the definition).
// Rows are the positions sorted stably by their k cyclic bytes; L[row] is the byte before the position.
//  * img(j) = stable counting-sort image of row j by L[j]: it orders the predecessors (p-1) by (L[j], context(p), p),
//    i.e. by their (k+1)-context — in particular by their k-context, so every k-context group occupies the same index
//    range in image order as in row order.
//  * group starts for contexts of length d+1 follow from those of length d: image m = img(j) starts a group iff j is
//    the first row of its d-group carrying that byte (k-1 linear rounds).
//  * walking the text backwards visits positions in decreasing order, and rows inside a group are in increasing
//    position order, so the predecessor of the current row is the LAST unvisited row of the group holding img(row).
int bsc_st_decode(unsigned char* T, int n, int k, int index, int features)
{
    (void)features;
    if (T == nullptr || n < 0 || index < 0 || index > n) return LIBBSC_BAD_PARAMETER;
    if (k < 3 || k > 8) return LIBBSC_BAD_PARAMETER;
    if (n <= 1) return LIBBSC_NO_ERROR;
    if (index >= n) return LIBBSC_BAD_PARAMETER;
    const size_t N = (size_t)n;
    std::vector<unsigned> img(N), gstart(N);
    std::vector<unsigned char> head(N), next_head(N);
    unsigned base[256] = {0};
    {
        unsigned cnt[256] = {0};
        for (size_t j = 0; j < N; ++j) cnt[T[j]]++;
        unsigned sum = 0;
        for (int c = 0; c < 256; ++c) { base[c] = sum; if (cnt[c]) head[sum] = 1; sum += cnt[c]; }   // 1-context groups
        unsigned run[256];
        memcpy(run, base, sizeof run);
        for (size_t j = 0; j < N; ++j) img[j] = run[T[j]]++;
    }
    for (int d = 1; d < k; ++d) {                          // d-context groups -> (d+1)-context groups
        std::fill(next_head.begin(), next_head.end(), 0);
        long long last_group[256];
        for (int c = 0; c < 256; ++c) last_group[c] = -1;
        long long g = -1;
        for (size_t j = 0; j < N; ++j) {
            if (head[j]) g = (long long)j;
            const unsigned char c = T[j];
            if (last_group[c] != g) { last_group[c] = g; next_head[img[j]] = 1; }
        }
        head.swap(next_head);
    }
    // Dense group ids: gid[j] = number of group heads at or before j, minus one.  The walk is one serial chain, so what
    // matters is the number of dependent DRAM misses per step: fold (group of img(row), T[row]) into one 8-byte entry per row
    // (built with independent gathers, which overlap) and keep {first row, unvisited rows} per group in a dense table
    // indexed by group id, which is small enough to stay in cache (8 bytes per k-context group).  One DRAM miss per step
    // instead of four.
    std::vector<unsigned>& gid = gstart;
    unsigned ngroups = 0;
    for (size_t j = 0; j < N; ++j) { ngroups += head[j]; gid[j] = ngroups - 1; }
    struct Group { unsigned start, remaining; };
    std::vector<Group> grp(ngroups);
    for (size_t j = 0; j < N; ++j) { if (head[j]) { grp[gid[j]].start = (unsigned)j; grp[gid[j]].remaining = 0; } grp[gid[j]].remaining++; }
    std::vector<uint64_t> E(N);
    for (size_t j = 0; j < N; ++j) E[j] = (uint64_t)gid[img[j]] | ((uint64_t)T[j] << 32);
    { std::vector<unsigned>().swap(gstart); std::vector<unsigned>().swap(img); }

    std::vector<unsigned char> out(N);
    unsigned row = (unsigned)index;
    for (size_t t = N; t-- > 0;) {
        const uint64_t e = E[row];
        out[t] = (unsigned char)(e >> 32);
        Group& g = grp[(unsigned)e];
        if (g.remaining == 0) return LIBBSC_DATA_CORRUPT;
        row = g.start + --g.remaining;
    }
    if (row != (unsigned)index) return LIBBSC_DATA_CORRUPT;        // the walk must close on position 0's row
    memcpy(T, out.data(), N);
    return LIBBSC_NO_ERROR;
}

int bsc_decompress(const unsigned char* input, int inputSize, unsigned char* output, int outputSize, int features)
{
    int blockSize = 0, dataSize = 0;
    int info = bsc_block_info(input, inputSize, &blockSize, &dataSize, features);
    if (info != LIBBSC_NO_ERROR) return info;
    if (inputSize < blockSize || outputSize < dataSize) return LIBBSC_UNEXPECTED_EOB;
    if ((unsigned)get_i32(input + 20) != adler32(input + LIBBSC_HEADER_SIZE, (size_t)(blockSize - LIBBSC_HEADER_SIZE))) return LIBBSC_DATA_CORRUPT;
    const int mode = get_i32(input + 8);
    if (mode == 0) { memmove(output, input + LIBBSC_HEADER_SIZE, (size_t)dataSize); return LIBBSC_NO_ERROR; }

    const int index = get_i32(input + 12);
    const unsigned adler_data = (unsigned)get_i32(input + 16);
    const int lzpHashSize = (mode >> 16) & 0xff, lzpMinLen = (mode >> 8) & 0xff, coder = (mode >> 5) & 0x7, sorter = mode & 0x1f;

    // the coder writes dataSize bytes; decode through a scratch buffer when decompressing in place
    const bool inplace = (input == output);
    std::vector<unsigned char> copy;
    const unsigned char* src = input;
    if (inplace) { copy.assign(input, input + blockSize); src = copy.data(); }
    if (blockSize < LIBBSC_HEADER_SIZE + 2) return LIBBSC_DATA_CORRUPT;      // a coded block holds at least one payload byte and the index count
    int num_indexes = src[blockSize - 1];
    int indexes[256];
    const long long payload = (long long)blockSize - LIBBSC_HEADER_SIZE - 1 - 4LL * num_indexes;     // what the coder may read
    if (payload < 1) return LIBBSC_DATA_CORRUPT;
    if (num_indexes > 0) memcpy(indexes, src + blockSize - 1 - 4 * num_indexes, (size_t)4 * num_indexes);
    // guard the decoder's input and output before it reads / writes: the stream announces its own lengths
    int lzSize = coder_decompress_bounded(src + LIBBSC_HEADER_SIZE, payload, output, coder, features, dataSize);
    if (lzSize < LIBBSC_NO_ERROR) return lzSize;
    int rc = (sorter == LIBBSC_BLOCKSORTER_BWT) ? bsc_bwt_decode(output, lzSize, index, (unsigned char)num_indexes, indexes, features)
                                                : bsc_st_decode(output, lzSize, sorter, index, features);
    if (rc < LIBBSC_NO_ERROR) return rc;
    if (lzpHashSize != 0 || lzpMinLen != 0) {                  // undo LZP (libbsc.cpp:594-609); encoding it is not built (f3)
        std::unique_ptr<unsigned char, void (*)(void*)> tmp((unsigned char*)bschost::bigbuf_get((size_t)lzSize + 1), bschost::bigbuf_put);   // (par.h: kept buffers)
        if (!tmp) return LIBBSC_NOT_ENOUGH_MEMORY;
        memcpy(tmp.get(), output, (size_t)lzSize);
        const int r = lzp_decompress(tmp.get(), output, lzSize, dataSize, lzpHashSize, lzpMinLen);
        if (r < LIBBSC_NO_ERROR) return r;
        if (r != dataSize) return LIBBSC_DATA_CORRUPT;
    } else if (lzSize != dataSize) return LIBBSC_DATA_CORRUPT;
    return adler_data == adler32(output, (size_t)dataSize) ? LIBBSC_NO_ERROR : LIBBSC_DATA_CORRUPT;
}

}  // extern "C"
