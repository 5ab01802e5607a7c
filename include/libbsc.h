/*
 * libbsc.h — public block API of the MI355X-native block-sorting library.
 *
 * API-identical to the reference's public header (libbsc/libbsc.h:36-152, libbsc 3.3.5): same function
 * names, argument meaning, constants and error codes, so existing callers relink unchanged.  The text
 * of this header is ours.  Differences in behaviour are limited to what the hot-path scope states
 * (DESIGN.md): the block sorters run on the GPU only (no CPU sorter is shipped: without a usable GPU
 * the sorters return LIBBSC_GPU_NOT_SUPPORTED instead of silently falling back).  LZP preprocessing
 * (lzpHashSize / lzpMinLen) runs on the host, byte-identical to the reference's encoder.
 * Decode side: bsc_decompress is self-hosted (QLFC decoders, inverse BWT, inverse ST3..8 and LZP decoding on the
 * host), so it reads every block the reference can write.
 */
#ifndef LIBBSC_MI355X_LIBBSC_H
#define LIBBSC_MI355X_LIBBSC_H

#include <stddef.h>

#define LIBBSC_VERSION_MAJOR 3
#define LIBBSC_VERSION_MINOR 3
#define LIBBSC_VERSION_PATCH 5
#define LIBBSC_VERSION_STRING "3.3.5"

/* error codes (libbsc.h:41-51) */
#define LIBBSC_NO_ERROR                 0
#define LIBBSC_BAD_PARAMETER           -1
#define LIBBSC_NOT_ENOUGH_MEMORY       -2
#define LIBBSC_NOT_COMPRESSIBLE        -3
#define LIBBSC_NOT_SUPPORTED           -4
#define LIBBSC_UNEXPECTED_EOB          -5
#define LIBBSC_DATA_CORRUPT            -6
#define LIBBSC_GPU_ERROR               -7
#define LIBBSC_GPU_NOT_SUPPORTED       -8
#define LIBBSC_GPU_NOT_ENOUGH_MEMORY   -9

/* block sorters (libbsc.h:53-66) */
#define LIBBSC_BLOCKSORTER_NONE 0
#define LIBBSC_BLOCKSORTER_BWT  1
#define LIBBSC_BLOCKSORTER_ST3  3
#define LIBBSC_BLOCKSORTER_ST4  4
#define LIBBSC_BLOCKSORTER_ST5  5
#define LIBBSC_BLOCKSORTER_ST6  6
#define LIBBSC_BLOCKSORTER_ST7  7
#define LIBBSC_BLOCKSORTER_ST8  8

/* coders (libbsc.h:68-71) */
#define LIBBSC_CODER_NONE          0
#define LIBBSC_CODER_QLFC_STATIC   1
#define LIBBSC_CODER_QLFC_ADAPTIVE 2
#define LIBBSC_CODER_QLFC_FAST     3

/* features (libbsc.h:73-77); bit 8 selects the GPU like the reference's LIBBSC_FEATURE_CUDA */
#define LIBBSC_FEATURE_NONE           0
#define LIBBSC_FEATURE_FASTMODE       1
#define LIBBSC_FEATURE_MULTITHREADING 2
#define LIBBSC_FEATURE_LARGEPAGES     4
#define LIBBSC_FEATURE_CUDA           8
#define LIBBSC_FEATURE_GPU            LIBBSC_FEATURE_CUDA

#define LIBBSC_DEFAULT_LZPHASHSIZE 15
#define LIBBSC_DEFAULT_LZPMINLEN   128
#define LIBBSC_DEFAULT_BLOCKSORTER LIBBSC_BLOCKSORTER_BWT
#define LIBBSC_DEFAULT_CODER       LIBBSC_CODER_QLFC_STATIC
#define LIBBSC_DEFAULT_FEATURES    (LIBBSC_FEATURE_FASTMODE | LIBBSC_FEATURE_MULTITHREADING)

#define LIBBSC_HEADER_SIZE 28

#if defined(__GNUC__)
#define LIBBSC_API __attribute__((visibility("default")))
#else
#define LIBBSC_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- block API (libbsc.h:95-152) -------------------------------------------------------------------- */
LIBBSC_API int bsc_init(int features);
LIBBSC_API int bsc_init_full(int features, void* (*malloc_fn)(size_t), void* (*zero_malloc_fn)(size_t), void (*free_fn)(void*));
/* output must hold n + LIBBSC_HEADER_SIZE bytes; input == output selects the in-place variant. */
LIBBSC_API int bsc_compress(const unsigned char* input, unsigned char* output, int n, int lzpHashSize, int lzpMinLen,
                            int blockSorter, int coder, int features);
LIBBSC_API int bsc_store(const unsigned char* input, unsigned char* output, int n, int features);
LIBBSC_API int bsc_block_info(const unsigned char* blockHeader, int headerSize, int* pBlockSize, int* pDataSize, int features);
LIBBSC_API int bsc_decompress(const unsigned char* input, int inputSize, unsigned char* output, int outputSize, int features);

/* ---- stage API (bwt/bwt.h:45-68, st/st.h:47-68, coder/coder.h:45-66, adler32/adler32.h:47) ------------ */
LIBBSC_API int bsc_bwt_init(int features);
LIBBSC_API int bsc_bwt_encode(unsigned char* T, int n, unsigned char* num_indexes, int* indexes, int features);
LIBBSC_API int bsc_bwt_decode(unsigned char* T, int n, int index, unsigned char num_indexes, int* indexes, int features);
LIBBSC_API int bsc_st_init(int features);
LIBBSC_API int bsc_st_encode(unsigned char* T, int n, int k, int features);
LIBBSC_API int bsc_st_decode(unsigned char* T, int n, int k, int index, int features);
LIBBSC_API int bsc_coder_init(int features);
LIBBSC_API int bsc_coder_compress(const unsigned char* input, unsigned char* output, int n, int coder, int features);
LIBBSC_API int bsc_coder_decompress(const unsigned char* input, unsigned char* output, int coder, int features);
LIBBSC_API unsigned int bsc_adler32(const unsigned char* T, int n, int features);
/* LZP preprocessor (lzp/lzp.h:50-62); output must hold n bytes (compress) / the original size (decompress) */
LIBBSC_API int bsc_lzp_compress(const unsigned char* input, unsigned char* output, int n, int hashSize, int minLen, int features);
LIBBSC_API int bsc_lzp_decompress(const unsigned char* input, unsigned char* output, int n, int hashSize, int minLen, int features);

/* one QLFC sub-block (coder/qlfc/qlfc.h:44-101), exposed for stage-level parity tests */
LIBBSC_API int bsc_qlfc_encode_block(const unsigned char* input, unsigned char* output, int inputSize, int outputSize, int coder);
LIBBSC_API int bsc_qlfc_decode_block(const unsigned char* input, unsigned char* output, int coder);
/* QLFC rank transform of a sub-block: ranks[0..m) in run order, firstSeen[0..k) alphabet; returns m, *pK = k */
LIBBSC_API int bsc_qlfc_ranks(const unsigned char* input, int n, unsigned char* ranks, unsigned char* firstSeen, int* pK);

/* bench utility: `synth-text v1` generator */
LIBBSC_API int bsc_synth_text_v1(unsigned long long seed, unsigned char* out, long long n);

#ifdef __cplusplus
}
#endif
#endif
