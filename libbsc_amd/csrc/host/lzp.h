// lzp.h — LZP preprocessor (encoder: lzp.cpp, decoder: decode.cpp), internal API.
#pragma once
#include <cstdint>

namespace bschost {

int lzp_num_chunks(int n);
// One chunk; returns the coded size or NOT_COMPRESSIBLE / NOT_ENOUGH_MEMORY (lzp.cpp:529).
int lzp_encode_chunk(const uint8_t* in, int n, uint8_t* out, int out_cap, int hashSize, int minLen);
// Whole block into an n-byte buffer (lzp.cpp:798); features & MULTITHREADING selects the concurrent framing.
int lzp_compress(const uint8_t* in, uint8_t* out, int n, int hashSize, int minLen, int features);
// Inverse (lzp.cpp:813); never writes more than out_cap bytes.
int lzp_decompress(const uint8_t* in, uint8_t* out, int n, int out_cap, int hashSize, int minLen);

}  // namespace bschost
