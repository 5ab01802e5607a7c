// decode.cpp — the decode side of the API: bsc_decompress, bsc_bwt_decode.
//
// Container checks follow libbsc.cpp:522-617 (bsc_decompress) and :420-519 (in-place twin).  The inverse BWT is a
// plain host LF-mapping walk (the reference uses libsais_unbwt[_aux], bwt.cpp:283-334; same result, no auxiliary
// index parallelism yet).  Inverse ST (st.cpp:1014-1527) is row f4 and not built: ST blocks return
// LIBBSC_NOT_SUPPORTED from bsc_st_decode / bsc_decompress.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/libbsc.h"
#include "qlfc.h"

using namespace bschost;

static inline int get_i32(const unsigned char* p) { int v; memcpy(&v, p, 4); return v; }

extern "C" {

// L = [T[n-1]] ++ [T[SA[j]-1] : SA[j] != 0]; `index` (1-based) is where the end-of-text row was removed.
// Re-insert a virtual sentinel row at position `index`, build LF in one counting pass, walk it backwards from row 0.
int bsc_bwt_decode(unsigned char* T, int n, int index, unsigned char num_indexes, int* indexes, int features)
{
    (void)num_indexes; (void)indexes; (void)features;
    if (T == nullptr || n < 0 || index <= 0 || index > n) return LIBBSC_BAD_PARAMETER;     // bwt.cpp:285
    if (n <= 1) return LIBBSC_NO_ERROR;
    std::vector<unsigned> lf((size_t)n + 1);
    unsigned cnt[256] = {0};
    for (int i = 0; i < n; ++i) cnt[T[i]]++;
    unsigned base[256], sum = 1;                              // row 0 is the sentinel-first suffix
    for (int c = 0; c < 256; ++c) { base[c] = sum; sum += cnt[c]; }
    for (int i = 0; i <= n; ++i) {
        if (i == index) { lf[(size_t)i] = 0; continue; }
        const unsigned char c = T[i < index ? i : i - 1];
        lf[(size_t)i] = base[c]++;
    }
    std::vector<unsigned char> out((size_t)n);
    unsigned r = 0;
    for (int k = n - 1; k >= 0; --k) {
        if ((int)r == index) return LIBBSC_DATA_CORRUPT;     // walked into the sentinel early: inconsistent index
        out[(size_t)k] = T[(int)r < index ? r : r - 1];
        r = lf[r];
    }
    memcpy(T, out.data(), (size_t)n);
    return LIBBSC_NO_ERROR;
}

int bsc_st_decode(unsigned char*, int, int, int, int) { return LIBBSC_NOT_SUPPORTED; }

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
    if (lzpHashSize != 0 || lzpMinLen != 0) return LIBBSC_NOT_SUPPORTED;                  // LZP: out of scope (f3)
    if (sorter != LIBBSC_BLOCKSORTER_BWT) return LIBBSC_NOT_SUPPORTED;                    // inverse ST: row f4

    // the coder writes dataSize bytes; decode through a scratch buffer when decompressing in place
    const bool inplace = (input == output);
    std::vector<unsigned char> copy;
    const unsigned char* src = input;
    if (inplace) { copy.assign(input, input + blockSize); src = copy.data(); }
    int num_indexes = src[blockSize - 1];
    int indexes[256];
    if (num_indexes > 0) {
        if (blockSize - 1 - 4 * num_indexes < LIBBSC_HEADER_SIZE) return LIBBSC_DATA_CORRUPT;
        memcpy(indexes, src + blockSize - 1 - 4 * num_indexes, (size_t)4 * num_indexes);
    }
    // guard the decoder's output size before it writes: the stream announces its own length
    int lzSize = coder_decompress_bounded(src + LIBBSC_HEADER_SIZE, output, coder, features, dataSize);
    if (lzSize < LIBBSC_NO_ERROR) return lzSize;
    int rc = bsc_bwt_decode(output, lzSize, index, (unsigned char)num_indexes, indexes, features);
    if (rc < LIBBSC_NO_ERROR) return rc;
    if (lzSize != dataSize) return LIBBSC_DATA_CORRUPT;
    return adler_data == adler32(output, (size_t)dataSize) ? LIBBSC_NO_ERROR : LIBBSC_DATA_CORRUPT;
}

}  // extern "C"
