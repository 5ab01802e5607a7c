// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI wrapper around the *real* reference libbsc (compiled from the sources
// where they lie under /root/reference by oracle/Makefile into oracle/_ref/).
// The reference objects are compiled with -fvisibility=hidden so none of the
// reference's own bsc_* symbols escape; only the ref_* entry points below are
// exported.  That lets the test process load the product library (which exports
// the real bsc_* names) and this one side by side without symbol interposition.
//
// Every wrapper is a 1:1 forward to the reference function named in its comment.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <omp.h>

#include "libbsc.h"          // /root/reference/libbsc/libbsc.h
#include "bwt/bwt.h"
#include "st/st.h"
#include "coder/coder.h"
#include "coder/qlfc/qlfc.h"
#include "adler32/adler32.h"
#include "lzp/lzp.h"

// Internal (non-static) reference function, qlfc.cpp:398 (scalar) / :200 (SSE/AVX).
unsigned char* bsc_qlfc_transform(const unsigned char* input, unsigned char* buffer, int n, unsigned char* MTFTable);

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API int ref_bsc_init(int features) { return bsc_init(features); }                                   // libbsc.cpp:63
REF_API int ref_bsc_compress(const unsigned char* in, unsigned char* out, int n, int lzpHashSize,
                             int lzpMinLen, int sorter, int coder, int features)                        // libbsc.cpp:213
{ return bsc_compress(in, out, n, lzpHashSize, lzpMinLen, sorter, coder, features); }
REF_API int ref_bsc_store(const unsigned char* in, unsigned char* out, int n, int features)             // libbsc.cpp:68
{ return bsc_store(in, out, n, features); }
REF_API int ref_bsc_block_info(const unsigned char* hdr, int hdrSize, int* pBlockSize, int* pDataSize, int features)  // libbsc.cpp:340
{ return bsc_block_info(hdr, hdrSize, pBlockSize, pDataSize, features); }
REF_API int ref_bsc_decompress(const unsigned char* in, int inSize, unsigned char* out, int outSize, int features)    // libbsc.cpp:522
{ return bsc_decompress(in, inSize, out, outSize, features); }

REF_API int ref_bsc_bwt_encode(unsigned char* T, int n, unsigned char* num_indexes, int* indexes, int features)       // bwt.cpp:178
{ return bsc_bwt_encode(T, n, num_indexes, indexes, features); }
REF_API int ref_bsc_bwt_decode(unsigned char* T, int n, int index, unsigned char num_indexes, int* indexes, int features) // bwt.cpp:283
{ return bsc_bwt_decode(T, n, index, num_indexes, indexes, features); }

REF_API int ref_bsc_st_encode(unsigned char* T, int n, int k, int features) { return bsc_st_encode(T, n, k, features); }  // st.cpp:990
REF_API int ref_bsc_st_decode(unsigned char* T, int n, int k, int index, int features)                  // st.cpp:1491
{ return bsc_st_decode(T, n, k, index, features); }

REF_API int ref_bsc_coder_compress(const unsigned char* in, unsigned char* out, int n, int coder, int features)       // coder.cpp:244
{ return bsc_coder_compress(in, out, n, coder, features); }
REF_API int ref_bsc_coder_decompress(const unsigned char* in, unsigned char* out, int coder, int features)            // coder.cpp:273
{ return bsc_coder_decompress(in, out, coder, features); }

REF_API int ref_bsc_lzp_compress(const unsigned char* in, unsigned char* out, int n, int hashSize, int minLen, int features)   // lzp.cpp:798
{ return bsc_lzp_compress(in, out, n, hashSize, minLen, features); }
REF_API int ref_bsc_lzp_decompress(const unsigned char* in, unsigned char* out, int n, int hashSize, int minLen, int features) // lzp.cpp:813
{ return bsc_lzp_decompress(in, out, n, hashSize, minLen, features); }

// One QLFC sub-block (what coder.cpp:61 dispatches to); coder = 1 static / 2 adaptive / 3 fast.
REF_API int ref_bsc_qlfc_encode_block(const unsigned char* in, unsigned char* out, int inSize, int outSize, int coder)
{
    if (coder == LIBBSC_CODER_QLFC_STATIC)   return bsc_qlfc_static_encode_block(in, out, inSize, outSize);     // qlfc.cpp:2138
    if (coder == LIBBSC_CODER_QLFC_ADAPTIVE) return bsc_qlfc_adaptive_encode_block(in, out, inSize, outSize);   // qlfc.cpp:2133
    if (coder == LIBBSC_CODER_QLFC_FAST)     return bsc_qlfc_fast_encode_block(in, out, inSize, outSize);       // qlfc.cpp:2161
    return LIBBSC_BAD_PARAMETER;
}
REF_API int ref_bsc_qlfc_decode_block(const unsigned char* in, unsigned char* out, int coder)
{
    if (coder == LIBBSC_CODER_QLFC_STATIC)   return bsc_qlfc_static_decode_block(in, out);
    if (coder == LIBBSC_CODER_QLFC_ADAPTIVE) return bsc_qlfc_adaptive_decode_block(in, out);
    if (coder == LIBBSC_CODER_QLFC_FAST)     return bsc_qlfc_fast_decode_block(in, out);
    return LIBBSC_BAD_PARAMETER;
}

// QLFC rank transform (qlfc.cpp:200/398).  ranks_out receives the m rank bytes (forward order),
// mtf_out the 256-byte MTF table as the reference leaves it.  Returns m (number of runs).
REF_API int ref_bsc_qlfc_transform(const unsigned char* in, int n, unsigned char* ranks_out, unsigned char* mtf_out)
{
    if (n <= 0) return 0;
    unsigned char* buffer = new unsigned char[(size_t)n + 64];
    unsigned char* first = bsc_qlfc_transform(in, buffer, n, mtf_out);
    int m = (int)((buffer + n) - first);
    memcpy(ranks_out, first, (size_t)m);
    delete[] buffer;
    return m;
}

REF_API unsigned int ref_bsc_adler32(const unsigned char* T, int n, int features) { return bsc_adler32(T, n, features); } // adler32.cpp:82

REF_API int ref_omp_max_threads(void) { return omp_get_max_threads(); }
REF_API void ref_omp_set_threads(int t) { omp_set_num_threads(t); }
REF_API double ref_wtime(void) { return omp_get_wtime(); }
