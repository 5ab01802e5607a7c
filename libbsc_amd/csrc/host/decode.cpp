// decode.cpp — decode side of the API.  NOT YET IMPLEMENTED in this commit:
// stored blocks decode; compressed blocks return LIBBSC_NOT_SUPPORTED (tests judge our streams with the
// reference decoder until the QLFC decoders and the inverse BWT land).
#include <cstring>
#include "../../../include/libbsc.h"
#include "qlfc.h"
namespace bschost { int qlfc_decode_block(const uint8_t*, uint8_t*, int) { return NOT_SUPPORTED; } }
extern "C" {
int bsc_bwt_decode(unsigned char*, int, int, unsigned char, int*, int) { return LIBBSC_NOT_SUPPORTED; }
int bsc_st_decode(unsigned char*, int, int, int, int) { return LIBBSC_NOT_SUPPORTED; }
int bsc_decompress(const unsigned char* input, int inputSize, unsigned char* output, int outputSize, int features)
{
    int blockSize = 0, dataSize = 0;
    int info = bsc_block_info(input, inputSize, &blockSize, &dataSize, features);
    if (info != LIBBSC_NO_ERROR) return info;
    if (inputSize < blockSize || outputSize < dataSize) return LIBBSC_UNEXPECTED_EOB;
    unsigned a; memcpy(&a, input + 20, 4);
    if (a != bsc_adler32(input + LIBBSC_HEADER_SIZE, blockSize - LIBBSC_HEADER_SIZE, features)) return LIBBSC_DATA_CORRUPT;
    int mode; memcpy(&mode, input + 8, 4);
    if (mode == 0) { memmove(output, input + LIBBSC_HEADER_SIZE, (size_t)dataSize); return LIBBSC_NO_ERROR; }
    return LIBBSC_NOT_SUPPORTED;
}
}
