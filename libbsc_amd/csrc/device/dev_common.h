// dev_common.h — shared declarations for the gfx950 device layer (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../../include/bscgpu.h"

typedef unsigned long long u64;
typedef unsigned int       u32;
typedef unsigned short     u16;
typedef unsigned char      u8;

// libbsc error codes (libbsc.h:41-51)
#define BSC_NO_ERROR                0
#define BSC_BAD_PARAMETER          -1
#define BSC_NOT_ENOUGH_MEMORY      -2
#define BSC_NOT_COMPRESSIBLE       -3
#define BSC_NOT_SUPPORTED          -4
#define BSC_GPU_ERROR              -7
#define BSC_GPU_NOT_SUPPORTED      -8
#define BSC_GPU_NOT_ENOUGH_MEMORY  -9

// ---------------------------------------------------------------------------------------------
// Work decomposition shared by all "chunked" kernels: the input is cut into at most MAX_CHUNKS
// contiguous chunks, one workgroup per chunk, each chunk a whole number of tiles.  ~1024 chunks
// = 4 workgroups of 256 threads on each of the 256 CUs (all resident at once, b % 8 -> XCD so
// every XCD streams an equal share).
// ---------------------------------------------------------------------------------------------
constexpr int   WG          = 256;      // threads per workgroup (4 wave64)
constexpr int   WAVES       = WG / 64;
#ifndef BSC_MAX_CHUNKS
#define BSC_MAX_CHUNKS 1024
#endif
constexpr int   MAX_CHUNKS  = BSC_MAX_CHUNKS;

struct Chunking {
    u32 num_tiles;     // total tiles
    u32 chunk_tiles;   // tiles per chunk
    u32 num_chunks;    // workgroups to launch
};
static inline Chunking make_chunking(u64 n, u32 tile) {
    Chunking c;
    c.num_tiles   = (u32)((n + tile - 1) / tile);
    if (c.num_tiles == 0) c.num_tiles = 1;
    c.chunk_tiles = (c.num_tiles + MAX_CHUNKS - 1) / MAX_CHUNKS;
    c.num_chunks  = (c.num_tiles + c.chunk_tiles - 1) / c.chunk_tiles;
    return c;
}

// ---------------------------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------------------------
struct ScatterLaunch { double ms; u64 records; };

// Pinned host buffers one block's QLFC front-end output lands in (sized for max_n).
struct HostSlot {
    u8*  hsym   = nullptr;   // run symbols
    u8*  hrank  = nullptr;   // QLFC ranks
    u32* hstart = nullptr;   // run start positions
    u8*  run_base = nullptr; size_t run_bytes = 0;   // the one pinned mapping behind hsym / hrank / hstart
    u16* hps    = nullptr;   // probability stream of the device coder (allocated when that path is first used)
    size_t hps_cap = 0;      // entries
    hipEvent_t copy_ev = nullptr;   // recorded on the copy stream behind the block's p-stream copy (its last piece); guards the device buffer's reuse
    hipEvent_t part_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // behind the piece of each sub-block: a coder task waits
                                    // for its own sub-blocks only (the stream leaves sub-block by sub-block, 366 MB in all for a 64 MiB text block)
    // the same pieces through the DMA engine directly (dma_copy.h): one HSA signal per piece, the landing zone's device address
    uint64_t part_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void* hps_dev = nullptr;
};
constexpr int MAX_SLOTS = 8;

struct DevCoder;                  // device-side static coder state (devcoder.hip), created on first use

struct bscgpu_ctx {
    int          device      = 0;
    hipStream_t  stream      = nullptr;
    hipStream_t  copy_stream = nullptr;   // D2H of the device coder's p stream, overlapped with the next block's GPU stage
    hipEvent_t   ps_guard[2] = {nullptr, nullptr};   // last copy out of each device p-stream buffer (not owned: a slot's copy_ev)
    uint64_t     ps_guard_sig[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};   // ... or, when the copy went through dma_copy.h, the signals of its pieces (host-side wait)
    int          ps_toggle   = 0;
    hipEvent_t   sync_ev     = nullptr;   // blocking-sync event: waiting threads sleep instead of spinning (host CPUs are the scarce resource)
    int64_t      max_n       = 0;
    char*        arena       = nullptr;
    size_t       arena_bytes = 0;

    // carved device buffers (sized for max_n)
    u8*  dT     = nullptr;   // text with 64 B zero/cyclic padding on both sides (points at T[0])
    u8*  dL     = nullptr;   // output bytes
    u64* kA     = nullptr;  u64* kB = nullptr;   // sort keys ping/pong
    u32* vA     = nullptr;  u32* vB = nullptr;   // sort values ping/pong
    u32* SA     = nullptr;
    u32* ISA    = nullptr;
    u32* cpos[2] = {nullptr, nullptr};
    u32* csa [2] = {nullptr, nullptr};
    u32* cgrp[2] = {nullptr, nullptr};
    u8*  flags  = nullptr;
    u32* counts = nullptr;   // [256][MAX_CHUNKS] radix per-chunk digit counts -> offsets
    u32* rowtot = nullptr;   // [256]
    u32* segsum = nullptr;   // [2][MAX_CHUNKS] per-chunk (unsorted count, last head+1)
    u32* segoff = nullptr;   // [2][MAX_CHUNKS] scanned
    u32* dscal  = nullptr;   // small device scalars (1024 u32, mirrored slot by slot in hscal)
    u64* dscal64 = nullptr;  // small device u64 scalars (16)
    u64* adler_part = nullptr; // [MAX_CHUNKS][2]
    u32* tile_counts = nullptr; size_t tile_counts_cap = 0;   // [256][tiles of 8192 records]: BSC_RS_ORDER=1 experiment, allocated on first use
    u32* long_tables = nullptr;   // BWT text rounds: counting-sort tables of the long-group split (bwt.hip LongTables), allocated on first use
    u64* wc_sink = nullptr;  // [512 * 1024] write sink for predicated-off lanes of rs_scatter_wc
    // single-read digit passes (radix_onesweep.hip), allocated on first use
    int  num_cus = 256;           // hipDeviceAttributeMultiprocessorCount of the context's device
    int  os_mode = 0;             // BSC_RS_ONESWEEP: 0 = off, 1 = large (key, value) sorts only, 2 = every sort of >= 4 tiles (tests), 3 = large sorts, keys-only too (default)
    u32* os_agg = nullptr;        // [tiles][256] tile rows {launch tag, digit count}
    u32* os_zero = nullptr;       // [8 passes] x {control block, digit totals, batch rows}: cleared per sort
    u32  os_tiles_cap = 0;
    u32  os_pass_stride = 0;      // words
    u32  os_batch_words = 0;      // words of the batch rows inside a pass's block (the group rows follow)
    u32  os_epoch = 0;            // launches so far (launch tag = epoch % 255 + 1)
    bool os_check_pending = false;   // hscal[OS_ERR_SLOT] has not been looked at since the last single-read sort
    bool os_available = true;        // the single-read kernels could be set up on this device (radix_onesweep_setup)
    int  os_retries = 0;             // transforms redone through the three-kernel passes after a give-up (bscgpu_debug_counter)
    int  dc_spf = 0;                 // BSCGPU_OPT_DC_STREAM_STATIC (context.hip reads BSC_DC_SPF at creation)
    int  dc_p13 = 1;                 // BSCGPU_OPT_DC_PACKED_STREAM: the static coder's p stream leaves as 13 bits per decision (BSC_PS13=0: 16-bit entries)
    bool os_gave_up = false;         // radix_onesweep_check found a give-up: the caller may redo its sorts through the three-kernel passes
    // pinned host
    u32* hscal  = nullptr;   // 1024 u32 (slot map: the users' comments; OS_ERR_SLOT = 1000)
    u64* hscal64 = nullptr;
    u64* hadler = nullptr;
    u64* hsplit = nullptr;   // pinned: split-flag words (max_n / 256 + 64 bytes)
    HostSlot slots[MAX_SLOTS];   // pinned landing zones for the QLFC front end; >1 when blocks are pipelined
    int      nslots = 0;
    DevCoder* dc = nullptr;
    bool     dc_alloc_failed = false;   // the device coder's arena did not fit: this context keeps the host model (not retried per block)
    int      dc_last_fail = 0;   // why the last block left the device coder (bit mask, devcoder.hip FAIL_*), 0 = it did not
    int      dc_replays = 0;     // evaluation chunks replayed serially in the last block
    int      rs_wc_mode = 0;     // BSC_RS_WC as read at context creation (radix_engine_setup)

    // profiling
    bool         prof        = false;
    bscgpu_kstat kstat[BSCGPU_K_COUNT];
    std::vector<ScatterLaunch> scatter_log;
    struct Pending { hipEvent_t a, b; int kind; u64 bytes; u64 records; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    double       stage_ms[6] = {0, 0, 0, 0, 0, 0};

    std::string  err;
};

int  ctx_fail(bscgpu_ctx* c, int code, const char* what, hipError_t e);
// BSCGPU_TIMING=1: one line on stderr per set-up step (context creation, arenas, landing zones) with its wall time — where a short
// job's start-up goes
bool ctx_timing_on();
struct CtxTimer {
    const char* what; double t0; bool on;
    explicit CtxTimer(const char* w);
    ~CtxTimer();
};
hipError_t ctx_sync(bscgpu_ctx* c);   // wait for everything queued on c->stream without burning a CPU
#define HIP_TRY(ctx, expr)                                                         \
    do { hipError_t _e = (expr);                                                   \
         if (_e != hipSuccess) return ctx_fail((ctx), BSC_GPU_ERROR, #expr, _e);   \
    } while (0)

// profiling brackets: PROF_BEGIN/END record events on ctx->stream around a launch.
void prof_begin(bscgpu_ctx* c, int kind, u64 bytes, u64 records);
void prof_end(bscgpu_ctx* c);
void prof_collect(bscgpu_ctx* c);   // after a stream sync: fold pending events into kstat

// ---- internal device-layer entry points (all asynchronous on ctx->stream) ---------------------
struct RadixPass { int shift; int bits; };
// Sort n records; passes applied in order (LSD).  Result lands in (keys, vals) if the number of
// passes is even, else in (keys_alt, vals_alt); *in_alt tells which.
// emit_pos (optional; one keys-only pass): emit_pos[i] = index in the output of input record i.
int radix_sort_passes(bscgpu_ctx* c, u64* keys, u64* keys_alt, u32* vals, u32* vals_alt, u64 n,
                      const RadixPass* passes, int npasses, int* in_alt, u32* emit_pos = nullptr);

constexpr int OS_ERR_SLOT = 1000;           // hscal / dscal word: sticky error word of the single-read digit passes (cleared by radix_onesweep_check only)
int  radix_onesweep_setup(bscgpu_ctx* c);
bool radix_onesweep_wanted(const bscgpu_ctx* c, u64 n, int npasses, bool has_val);
int  radix_onesweep_sort(bscgpu_ctx* c, u64* keys, u64* keys_alt, u32* vals, u32* vals_alt, u64 n, const RadixPass* passes, int npasses);
int  radix_onesweep_check(bscgpu_ctx* c);   // after the next stream sync: did a pass of the last such sort give up a wait?
int radix_engine_setup(bscgpu_ctx* c);     // per-device kernel attributes; bscgpu_create calls it with c->device current

int bwt_device(bscgpu_ctx* c, const u8* dT_user, u8* dL_user, int64_t n, int64_t r, u32* I_host,
               int64_t* primary_out);
int st_device(bscgpu_ctx* c, const u8* dT_user, u8* dOut_user, int n, int k, int* index_out);
int adler32_device(bscgpu_ctx* c, const u8* d, int64_t n, u32* out);
void launch_seg_scan(bscgpu_ctx* c, u32 num_chunks);
int qlfc_front_split(bscgpu_ctx* c, const u8* dL, u32 n, int nblocks, int* start, int* size);
int qlfc_front_runs(bscgpu_ctx* c, const u8* dL, u32 n, int nblocks, const int* start, u32* m_out, u32* run_first, u32* first_run_host,
                    HostSlot& slot, bool copy_runs = true);
int qlfc_front_copy_runs(bscgpu_ctx* c, u32 m, HostSlot& slot);
int ctx_ensure_slots(bscgpu_ctx* c, int count);
int ctx_ensure_pstream_slot(bscgpu_ctx* c, HostSlot& slot, size_t entries);     // pinned landing zone for a block's p stream
int ctx_ensure_run_slot(bscgpu_ctx* c, HostSlot& slot);                           // pinned landing zone for a block's run arrays
// device-side model of the static QLFC coder (devcoder.hip): probability stream of a whole block from the front end's run arrays
int  devcoder_pstream(bscgpu_ctx* c, const u8* dsym, const u8* drank, const u32* dstart, u32 m, u32 n, int nb, const u32* run_first,
                      const int* max_rank, u32* D_out, u32* poff_out, u16* dbg, int psbuf = 0, int coder = 1 /* 1 static (-e1), 3 fast (-e0) */,
                      int* packed_out = nullptr /* non-null: the caller takes the 13-bit packed stream (devcoder.hip DcP13); *packed_out = 1 if that is what was written */);
const u16* devcoder_pstream_ptr(const bscgpu_ctx* c, int psbuf = 0);
void devcoder_destroy(bscgpu_ctx* c);
int64_t devcoder_arena_bytes(const bscgpu_ctx* c);
void devcoder_warm_tables();               // starts the model tables' computation on a background thread (first context of the process)

// ---------------------------------------------------------------------------------------------
// Device helpers (wave64)
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ u32 lane_id() { return __lane_id(); }

__device__ __forceinline__ u64 lanemask_lt() {
    u32 l = lane_id();
    return (l == 0) ? 0ull : (~0ull >> (64 - l));
}

// Inclusive wave64 scans on the DPP network (row shifts inside the rows of 16, then the two row broadcasts): six dependent VALU operations.
// ALL 64 LANES MUST BE ACTIVE (every caller scans whole wavefronts).  Round 6: the ds_bpermute form these replace (kept below as *_bperm)
// is six LDS round trips, each waited for — ~700 cycles per scan on the critical path of every tile of the device coder's partition kernels
// (four wavefronts per SIMD: nothing to hide it behind) and of every block scan of the chunked kernels.
__device__ __forceinline__ u32 wave_incl_sum(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ u32 wave_incl_max(u32 v) {
    auto mx = [](u32 a, u32 b) { return a > b ? a : b; };                         // (lanes without a source read 0: the identity of an unsigned max)
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ u32 wave_incl_sum_bperm(u32 v) {
    u32 l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if (l >= (u32)d) v += t;
    }
    return v;
}

// Block (256 threads) exclusive sum scan.  lds must hold >= 8 u32.  Returns exclusive prefix of v;
// *total receives the block total.  Contains __syncthreads(); all threads must call.
__device__ __forceinline__ u32 block_excl_sum(u32 v, u32* lds, u32* total) {
    u32 incl = wave_incl_sum(v);
    u32 w = threadIdx.x >> 6, l = lane_id();
    __syncthreads();                 // protect lds reuse from a previous call
    if (l == 63) lds[w] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) { u32 t = lds[i]; if ((u32)i < w) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
}
// Block inclusive max scan.
__device__ __forceinline__ u32 block_incl_max(u32 v, u32* lds, u32* total) {
    u32 incl = wave_incl_max(v);
    u32 w = threadIdx.x >> 6, l = lane_id();
    __syncthreads();
    if (l == 63) lds[w] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) { u32 t = lds[i]; if ((u32)i < w) base = (t > base) ? t : base; tot = (t > tot) ? t : tot; }
    *total = tot;
    return (base > incl) ? base : incl;
}
#endif
