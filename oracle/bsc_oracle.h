/* oracle/bsc_oracle.h — TEST INFRASTRUCTURE ONLY (see bsc_oracle.c). */
#ifndef BSC_ORACLE_H
#define BSC_ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
unsigned int orc_adler32(const unsigned char* T, int n);
int  orc_bwt(unsigned char* T, int n, unsigned char* num_indexes, int* indexes);
int  orc_st(unsigned char* T, int n, int k);
int  orc_qlfc_transform(const unsigned char* in, unsigned char* buffer, int n, unsigned char* mtf);
int  orc_qlfc_encode(const unsigned char* in, unsigned char* out, int n, int out_size, int coder);
void orc_split_blocks(const unsigned char* in, int n, int nb, int* start, int* size);
int  orc_coder_compress(const unsigned char* in, unsigned char* out, int n, int coder);
int  orc_store(const unsigned char* in, unsigned char* out, int n);
int  orc_compress(const unsigned char* in, unsigned char* out, int n, int sorter, int coder);
#ifdef __cplusplus
}
#endif
#endif
