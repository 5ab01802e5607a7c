// qlfc.h — host-side QLFC entropy coder (bit-exact with libbsc 3.3.5's coder/qlfc), internal API.
//
// The bit stream is fixed by the reference (qlfc.cpp:463-1336 encoders, rangecoder.h:38-271,
// predictor.h:40-213, qlfc_model.h); the code organisation here is ours:
//   * a run/rank front end (`qlfc_runs`) that turns a sorted sub-block into three flat arrays
//     (symbol, rank, run length) — the part that is data-parallel and is meant to move to the GPU;
//   * one templated "model walker" that enumerates the binary decisions of a run with their context
//     slots, shared by the static (-e1) and adaptive (-e2) coders, specialised at compile time per
//     decision class so every threshold/rate is an immediate;
//   * the 32-bit-range / 16-bit-unit range encoder.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace bschost {

// libbsc.h error codes
enum : int {
    OK = 0, BAD_PARAMETER = -1, NOT_ENOUGH_MEMORY = -2, NOT_COMPRESSIBLE = -3, NOT_SUPPORTED = -4,
    UNEXPECTED_EOB = -5, DATA_CORRUPT = -6, GPU_ERROR = -7, GPU_NOT_SUPPORTED = -8, GPU_NOT_ENOUGH_MEMORY = -9
};
enum : int { CODER_STATIC = 1, CODER_ADAPTIVE = 2, CODER_FAST = 3 };

// Lookup tables (unpacked once from qlfc_data.inc).
struct QlfcTables {
    short   stretch[4097];
    short   squash[4097];
    uint8_t rank_state[32768];
    uint8_t run_state[8192];
};
const QlfcTables& qlfc_tables();
// tuned constants of the static model, one row per decision class (layout: tools/gen_qlfc_data.py)
const short (*qlfc_static_params())[19];

// Run decomposition of a sub-block + QLFC ranks, as flat arrays (the layout the GPU front end produces):
//   sym[j], start[j] : symbol and start position of the j-th maximal run (length = start[j+1] - start[j], the
//                      last run ends at `end`);
//   rank[j]          : the QLFC rank of run j (qlfc.cpp:398-455): number of distinct symbols between run j and
//                      the next run of the same symbol, or — for the last run of a symbol — the number of distinct
//                      symbols that still occur later; the rank of the final run is forced to 1;
//   first_seen[0..k) : distinct symbols in order of first appearance (the stream's alphabet header).
struct RunView {
    const uint8_t*  sym   = nullptr;
    const uint8_t*  rank  = nullptr;
    const uint32_t* start = nullptr;
    uint32_t count = 0;
    uint32_t end   = 0;
    uint8_t  first_seen[256];
    int      nsym = 0;
    inline uint32_t len(uint32_t j) const { return (j + 1 < count ? start[j + 1] : end) - start[j]; }
};
struct QlfcRuns {                       // owning storage for the host-side front end
    std::vector<uint8_t>  sym, rank;
    std::vector<uint32_t> start;
    RunView view;
};
void qlfc_runs(const uint8_t* in, int n, QlfcRuns& out);

// Encode one sub-block from its run arrays.  Returns bytes written or NOT_COMPRESSIBLE.
int qlfc_encode_runs(const RunView& R, int in_size, uint8_t* out, int out_size, int coder);
// Static coder (-e1) from a precomputed probability stream (devcoder_model.h: [11:0] p, [12] bit, [13] run start): header,
// alphabet and range coding only — the model ran on the GPU.  Returns bytes written or NOT_COMPRESSIBLE.
int qlfc_encode_static_pstream(const uint8_t* first_seen, int nsym, int in_size, const uint16_t* ps, size_t count, uint8_t* out, int out_size);
// Two independent sub-blocks coded in one loop: the range coder's recurrence (range -> shift -> multiply -> select) is latency
// bound, two chains in flight nearly double a core's rate (1.9 -> 1.1 ns per decision on an EPYC 9575F).  res[k] as above.
struct PstreamJob { const uint8_t* first_seen; int nsym; int in_size; const uint16_t* ps; size_t count; uint8_t* out; int out_size; };
void qlfc_encode_static_pstream_pair(const PstreamJob& A, const PstreamJob& B, int* resA, int* resB);
// Eight sub-blocks, one per AVX2 lane (J[8], res[8]).  false = not done (a stream reached its output budget, or no AVX2):
// the caller codes the sub-blocks with the functions above.
bool qlfc_encode_static_pstream_x8(const PstreamJob* J, int* res);
// The packed stream (devcoder.hip DcP13): 13 bits per decision {probability[11:0], coded bit}, eight decisions in 13 bytes, no run-start
// mark (a stream that reaches its budget anywhere returns NOT_COMPRESSIBLE / false and the block is redone on the host model).  PstreamJob::ps
// then points at the packed bytes.  qlfc_pack_p13 makes the packed form of a 16-bit stream (tests, tools): out has (count + 7) / 8 * 13 bytes.
int  qlfc_encode_static_p13(const uint8_t* first_seen, int nsym, int in_size, const uint8_t* ps, size_t count, uint8_t* out, int out_size);
void qlfc_encode_static_p13_pair(const PstreamJob& A, const PstreamJob& B, int* resA, int* resB);
bool qlfc_encode_static_p13_x8(const PstreamJob* J, int* res);
void qlfc_pack_p13(const uint16_t* ps, size_t count, uint8_t* out);
// The fast coder's back half (-e0, qlfc.cpp:1135-1336) behind the device model: entries {[12:0] probability, [13] bit, [14] first
// decision of a run, [15] run side = 11-bit precision, else 13} (devcoder_model.h PSF_*); header and alphabet as encode_model2 writes them.
int qlfc_encode_fast_pstream(const uint8_t* first_seen, int nsym, int in_size, const uint16_t* ps, size_t count, uint8_t* out, int out_size);
void qlfc_encode_fast_pstream_pair(const PstreamJob& A, const PstreamJob& B, int* resA, int* resB);
// sixteen sub-blocks = two blocks in the lanes of 512-bit registers (AVX-512F/VL/BW; false: not available or a stream near its budget)
bool qlfc_x16_available();
bool qlfc_encode_static_pstream_x16(const PstreamJob* J /*[16]*/, int* res /*[16]*/);
bool qlfc_encode_fast_pstream_x16(const PstreamJob* J /*[16]*/, int* res /*[16]*/);
bool qlfc_encode_fast_pstream_x8(const PstreamJob* J, int* res);      // eight sub-blocks in SIMD lanes, per-lane precision (as qlfc_encode_static_pstream_x8)
// Encode one sub-block (what coder.cpp:61 dispatches to).  Returns bytes written or NOT_COMPRESSIBLE.
int qlfc_encode_block(const uint8_t* in, uint8_t* out, int in_size, int out_size, int coder);
// Decode one sub-block; returns the decoded size or an error.
int qlfc_decode_block(const uint8_t* in, uint8_t* out, int coder);

// Block-level coder (coder.cpp:244 / :273): split into 1/2/4/8 sub-blocks, encode, frame.
int coder_compress(const uint8_t* in, uint8_t* out, int n, int coder, int features);
int coder_decompress(const uint8_t* in, uint8_t* out, int coder, int features);
// as above, but never reads more than in_size bytes of `in` nor writes more than max_out bytes (DATA_CORRUPT otherwise);
// UNBOUNDED_INPUT for callers whose API carries no input size (the reference's stage entry points)
constexpr long long UNBOUNDED_INPUT = (long long)1 << 62;
int coder_decompress_bounded(const uint8_t* in, long long in_size, uint8_t* out, int coder, int features, int max_out);
int qlfc_decode_block_bounded(const uint8_t* in, long long in_size, uint8_t* out, int coder, int max_out);
int coder_num_blocks(int n);
// Same framing, but the sub-blocks arrive as run arrays (GPU front end).  fetch_raw(start, size, dst) supplies the
// original bytes of a sub-block that has to be stored raw.
struct RawFetch { virtual int operator()(int start, int size, uint8_t* dst) = 0; virtual ~RawFetch() {} };
int coder_compress_views(const RunView* views, int nblocks, const int* start, const int* size, int n,
                         uint8_t* out, int coder, int features, RawFetch& fetch_raw);
void coder_split_blocks(const uint8_t* in, int n, int nblocks, int* start, int* size);

uint32_t adler32(const uint8_t* p, size_t n);

}  // namespace bschost
