/*
 * bscgpu.h — thin C ABI between libbsc-style host code and the MI355X (gfx950) HIP kernels.
 *
 * This is the drop-in boundary for the reference's GPU plug points.  Each entry point names the
 * reference interface it replaces (paths relative to the reference tree, libbsc 3.3.5):
 *
 *   bscgpu_create / bscgpu_destroy   <->  libcubwt_allocate_device_storage / libcubwt_free_device_storage
 *                                         (libbsc/bwt/libcubwt/libcubwt.cuh:60-71; called from bwt.cpp:92-115)
 *   bscgpu_bwt                       <->  libcubwt_bwt      (libcubwt.cuh:73-80,  bwt.cpp:148-162)
 *   bscgpu_bwt_aux                   <->  libcubwt_bwt_aux  (libcubwt.cuh:82-89,  bwt.cpp:104-118)
 *   bscgpu_st_encode                 <->  bsc_st_encode_cuda (libbsc/st/st.cuh:57, st.cpp:998-1002)
 *
 * Like the reference hooks these take HOST pointers, do their own H2D/D2H and are synchronous on
 * return.  The *_device variants take DEVICE pointers (input already resident in HBM) and are what
 * bench.py times; they have no counterpart in the reference (its coder never leaves the CPU).
 *
 * Plain C, plain pointers and sizes.  Return values follow libbsc.h:41-51 (>= 0 ok, < 0 error code;
 * -7 GPU_ERROR, -8 GPU_NOT_SUPPORTED, -9 GPU_NOT_ENOUGH_MEMORY, -1 BAD_PARAMETER).
 */
#ifndef BSCGPU_H
#define BSCGPU_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define BSCGPU_API __attribute__((visibility("default")))
#else
#define BSCGPU_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bscgpu_ctx bscgpu_ctx;

/* Number of visible HIP devices (0 when there is no usable GPU). */
BSCGPU_API int bscgpu_device_count(void);

/* Create a per-device context with an HBM arena large enough for blocks of up to max_n bytes.
 * One context per GPU (one process per GPU in the multi-GPU driver); a context is not re-entrant:
 * serialise calls on it (the reference serialises on one global lock, bwt.cpp:50-52). */
BSCGPU_API int  bscgpu_create(bscgpu_ctx** ctx, int device, int64_t max_n);
BSCGPU_API void bscgpu_destroy(bscgpu_ctx* ctx);

/* Bytes of device memory held by the context's arena. */
BSCGPU_API int64_t bscgpu_arena_bytes(const bscgpu_ctx* ctx);

/* ---- forward BWT ------------------------------------------------------------------------- */
/* L may alias T.  Returns the primary index (1..n) like libcubwt_bwt, or a negative error. */
BSCGPU_API int64_t bscgpu_bwt(bscgpu_ctx* ctx, const uint8_t* T, uint8_t* L, int64_t n);
/* As above plus auxiliary indexes: r is a power of two, I[0..(n-1)/r] receives 1-based ranks
 * (I[0] = primary index), exactly libsais_bwt_aux / libcubwt_bwt_aux semantics.  Returns 0. */
BSCGPU_API int64_t bscgpu_bwt_aux(bscgpu_ctx* ctx, const uint8_t* T, uint8_t* L, int64_t n, int64_t r, uint32_t* I);
/* Device-resident variant: dT, dL are device pointers (may alias); I is a HOST array or NULL
 * (r ignored when I is NULL).  Returns the primary index. */
BSCGPU_API int64_t bscgpu_bwt_device(bscgpu_ctx* ctx, const void* dT, void* dL, int64_t n, int64_t r, uint32_t* I);

/* ---- inverse BWT (libcubwt_unbwt's role, libcubwt.cuh:91-104; reached from bsc_bwt_decode, bwt.cpp:233-281) ---------- */
/* L[0..n) and the 1-based primary index as bsc_bwt_encode writes them -> T[0..n); host pointers, may alias; synchronous.
 * Returns 0, LIBBSC_DATA_CORRUPT (-6) when the rows do not form one cycle through the sentinel row, or another libbsc code. */
BSCGPU_API int bscgpu_unbwt(bscgpu_ctx* ctx, const uint8_t* L, uint8_t* T, int64_t n, int64_t index);

/* ---- Sort Transform (order k = 3..8) ----------------------------------------------------- */
/* In place on host T[0..n).  Returns the 0-based primary index like bsc_st_encode (st.cpp:990). */
BSCGPU_API int bscgpu_st_encode(bscgpu_ctx* ctx, uint8_t* T, int n, int k);
BSCGPU_API int bscgpu_st_encode_device(bscgpu_ctx* ctx, const void* dT, void* dOut, int n, int k);

/* ---- Adler-32 of a device buffer (adler32.cpp:82) ---------------------------------------- */
BSCGPU_API int bscgpu_adler32_device(bscgpu_ctx* ctx, const void* dT, int64_t n, uint32_t* out);

/* ---- LSD radix sort primitive (the kernel the roofline is measured on) -------------------- */
/* Stable sort of n (u64 key, u32 value) records on key bits [begin_bit, end_bit), 8-bit digits.
 * All pointers are device pointers; *_alt are same-sized scratch (ping-pong).  vals may be NULL
 * (keys-only).  On return *result_in_alt is 1 when the sorted data lives in the *_alt buffers. */
BSCGPU_API int bscgpu_radix_sort_u64(bscgpu_ctx* ctx, void* keys, void* keys_alt, void* vals, void* vals_alt,
                          int64_t n, int begin_bit, int end_bit, int* result_in_alt);

/* ---- QLFC static coder (-e1): the adaptive model on the GPU --------------------------------- */
/* Stage function: sorted host block L[0..n) -> sub-block split (coder.cpp:70-109) + run/rank front end (qlfc.cpp:398-455)
 * + every probability of the static model (qlfc.cpp:896-1126, predictor.h:53-61,121) as a stream of 16-bit entries
 * {[11:0] probability, [12] coded bit, [13] first decision of a run}, stream order, all sub-blocks back to back
 * (poff[b]..poff[b+1] = sub-block b).  Returns the number of decisions, LIBBSC_NOT_SUPPORTED (-4) when the block has to take
 * the host model (more than 256 distinct decision types, ...; bscgpu_last_error says which), or a negative libbsc code.
 * dbg (optional, [3][cap]): the state- / char- / static-counter value behind every decision. */
BSCGPU_API int64_t bscgpu_qlfc_static_pstream(bscgpu_ctx* ctx, const uint8_t* L, int n, uint16_t* out, int64_t cap, int* nblocks,
                                   int* sub_start /*[8]*/, int* sub_size /*[8]*/, int64_t* poff /*[9]*/, uint16_t* dbg);
/* The same stage with the stream in the form that crosses PCIe since round 6 (BSCGPU_OPT_DC_PACKED_STREAM): 13 bits per decision
 * {[11:0] probability, [12] coded bit}, eight decisions in 13 bytes (field e of a group at bits [13 e, 13 e + 13), little endian); sub-block
 * b's fields start at decision pbase[b] of the packed space — a multiple of 64, i.e. at byte pbase[b] / 8 * 13 — and its last group is
 * zero-padded.  out takes pbase[nblocks] / 8 * 13 bytes (cap_bytes).  Returns the number of decisions; LIBBSC_NOT_SUPPORTED also when
 * the packed form was not produced for this block (option off; 64 consecutive runs with more decisions than a wavefront stages —
 * bsc_compress then moves that block's stream as 16-bit entries). */
BSCGPU_API int64_t bscgpu_qlfc_static_pstream_packed(bscgpu_ctx* ctx, const uint8_t* L, int n, uint8_t* out, int64_t cap_bytes, int* nblocks,
                                   int* sub_start /*[8]*/, int* sub_size /*[8]*/, int64_t* poff /*[9]*/, int64_t* pbase /*[9]*/);

/* ---- full block compression with the BWT/ST + coder split across GPU and host ------------- */
/* bsc_compress semantics (libbsc.cpp:213) for input already in HBM: Adler-32 + sort transform on
 * the GPU, QLFC coder on host threads.  output is a HOST buffer of n + 28 bytes. */
BSCGPU_API int bscgpu_compress_device(bscgpu_ctx* ctx, const void* dInput, uint8_t* output, int n,
                           int blockSorter, int coder, int features);

/* Pipelined variant: up to `depth` (<= 8) blocks in flight on one GPU.  submit() runs the GPU stage of a block
 * (Adler-32, sort transform, QLFC front end, D2H of the run arrays) on the calling thread and hands the host stage
 * (QLFC modelling + range coding, one task per sub-block; container) to the process's coder threads, so block i+1 sorts
 * while blocks i, i-1, ... are coded.
 * dInput and output must stay valid until wait() returns for that ticket.  wait() returns what
 * bscgpu_compress_device would have returned.  One submitting thread per pipe.  The host work is queued as tasks for the
 * process's pool of coder threads, shared by all pipes (default: the CPUs the process may use — affinity and cgroup quota —
 * clamped to 4..64; BSCGPU_HOST_THREADS overrides the thread count, BSCGPU_HOST_CPUS the CPU budget idle CPUs are counted
 * against), so a depth of 3-4 keeps those threads and the GPU busy.  A device-model block is one eight-lane SIMD task (half the
 * CPU time, ~100 ms) when the pool is busy and four tasks of two interleaved sub-blocks (~50 ms) while at least four CPUs of
 * the budget are idle if BSC_RC_ADAPTIVE=1 (round 5: off by default — every block that is not marked low-latency is the former;
 * BSC_RC_SIMD=8 / 0 forces one of the two). */
typedef struct bscgpu_pipe bscgpu_pipe;
/* Extra `features` bit for bscgpu_pipe_submit*: code this block's sub-blocks as several short host tasks (two interleaved scalar range
 * coders per task, ~50 ms for a 64 MiB block; one coder per task, ~35 ms, when eight CPUs of the pool are idle) instead of one
 * eight-lane SIMD task (~100 ms, half the CPU time).  For the LAST blocks
 * of a job, where latency — the drain of the pipeline — counts and the coder threads are running dry anyway.  Output is identical. */
#define BSCGPU_FEATURE_LOW_LATENCY 0x10000
/* ... and with this bit as well: one coder per task (eight tasks) whatever the pool's load — for the very last block(s) of a job, whose
 * coding time is the job's last 35 ms whatever else is still running. */
#define BSCGPU_FEATURE_URGENT 0x20000
BSCGPU_API int  bscgpu_pipe_create(bscgpu_ctx* ctx, int depth, bscgpu_pipe** out);
/* How the pool has coded the pipes' blocks so far: out[0] blocks as eight scalar tasks, [1] as four pair tasks, [2] as one eight-lane
 * task, [3] blocks on the host model (one task per sub-block).  reset != 0 clears the counts.  (bench.py reports them.) */
BSCGPU_API void bscgpu_coder_pool_stats(uint64_t out[4], int reset);
/* ... of the eight-lane blocks, how many were coded two at a time in the sixteen lanes of 512-bit registers (round 6: half the CPU time per
 * block at equal throughput on long jobs, a few per cent slower on 20-block jobs: opt-in with BSC_RC_X16=1 on AVX-512F/VL/BW hosts;
 * BSC_RC_X16_WAIT_MS is how long a block waits for a partner: 15) */
BSCGPU_API uint64_t bscgpu_coder_pool_x16_blocks(int reset);
/* Where a job ends: `blocks` more blocks will be submitted to the pipes of this process (all pipes together), which drive `gpus` GPUs.
 * The blocks whose GPU stages END last are then coded as short tasks — per GPU the last one as eight single-stream tasks, the few
 * before it as pairs (two fewer than the contexts that interleave on a GPU) (BSC_TAIL_SINGLES / BSC_TAIL_PAIRS) — whatever their order of submission (several contexts interleave on a GPU,
 * so the two orders differ by up to 100 ms), everything earlier as one eight-lane task.  For callers that know the total, instead of
 * marking blocks BSCGPU_FEATURE_LOW_LATENCY at submission; blocks < 0 withdraws the announcement.  Output is identical either way. */
BSCGPU_API int  bscgpu_coder_pool_expect(long long blocks, int gpus);
/* The pool's own record of its tasks (recorded when BSCGPU_POOL_TRACE=1 is in the environment): up to cap rows of six doubles — start, end
 * (seconds on the clock bscgpu_steady_now reads), the block's id inside its pipe, first sub-block, sub-blocks per task (1, 2, 8; 16 for a
 * sixteen-lane task), the block's features.  Returns the number of rows; reset != 0 clears the record.  (bench.py prints it with
 * BSC_BENCH_TRACE=1: where a short job's last 100 ms go.) */
BSCGPU_API int    bscgpu_coder_pool_trace(double* out, int cap, int reset);
BSCGPU_API double bscgpu_steady_now(void);
/* 1 when a block's probability stream leaves the device through the HSA runtime's DMA copy (csrc/device/dma_copy.h) in this process,
 * 0 when it goes through hipMemcpyAsync (BSC_D2H_DMA=0, or no usable HSA runtime in the process).  Which engine that is depends on the
 * HIP runtime: a DMA engine on ROCm 7.2's, a 256-workgroup copy kernel on the one torch 2.10 carries (profiles/r06/d2h_copy_path.txt). */
BSCGPU_API int  bscgpu_d2h_dma_available(void);
/* The rule behind those shapes as a pure function (unit-tested on CPU): sub-blocks per coder task — 8 (one SIMD task), 2 or 1 — from
 * forced (-1 none, 8 or 0: BSC_RC_SIMD), low_latency (synchronous call or BSCGPU_FEATURE_LOW_LATENCY), pool_free (idle CPUs of the
 * pool's budget; -1: a synchronous call), sync_cpus (CPUs / synchronous callers running), wide_simd (AVX-512VL), adaptive. */
BSCGPU_API int bscgpu_coder_task_shape(int forced, int low_latency, int pool_free, int sync_cpus, int wide_simd, int adaptive);
BSCGPU_API void bscgpu_pipe_destroy(bscgpu_pipe* pipe);
BSCGPU_API int  bscgpu_pipe_submit(bscgpu_pipe* pipe, const void* dInput, uint8_t* output, int n,
                                   int blockSorter, int coder, int features);       /* ticket >= 0 or error */
/* Host-resident block with bsc_compress's full parameter list (libbsc.cpp:213), LZP included: LZP runs on the calling
 * thread (+ up to 8 chunk threads), then one H2D copy feeds the same GPU stage.  input must stay valid until wait(). */
BSCGPU_API int  bscgpu_pipe_submit_host(bscgpu_pipe* pipe, const uint8_t* input, uint8_t* output, int n,
                                        int lzpHashSize, int lzpMinLen, int blockSorter, int coder, int features);
BSCGPU_API int  bscgpu_pipe_wait(bscgpu_pipe* pipe, int ticket);
/* From ANY thread: block until the host stage of `ticket` is over; 1 and *result (what bscgpu_pipe_wait will return) when the block is
 * complete — its output buffer is final —, 0 when it needs its submitting thread after all (redo on the host model) or has been retired
 * already.  Does not retire the ticket.  For in-order collectors running beside the submitting thread (the job driver). */
BSCGPU_API int  bscgpu_pipe_peek(bscgpu_pipe* pipe, int ticket, int* result);

/* ---- multi-GPU job: every GPU of a node from one process, C/C++ callers ----------------------------------------------------
 * The reference parallelises over blocks with the CLI's OpenMP team (bsc.cpp:182-199: next block under critical(input),
 * bsc_compress, write under critical(output)) and knows one GPU behind one lock (bwt.cpp:50-52).  A job is that loop for N GPUs:
 * `contexts_per_device` pipes per device (their kernels interleave on the GPU), `depth` blocks in flight per pipe, one worker thread
 * per pipe pulling the next block from ONE queue (blocks are independent, so block b -> whichever GPU is free next: on equal GPUs one
 * block per GPU per round, with load balancing for free), host coding on the process-wide coder pool.  The caller adds blocks in
 * order and collects them in order: bscgpu_job_wait(b) returns what bsc_compress would have returned for block b, and the bytes
 * are in that block's output buffer — the ordered host gather of the multi-GPU run (inside one process nothing has to travel
 * between GPUs; the RCCL concatenation belongs to the one-process-per-GPU layout, libbsc_amd/multigpu.py).
 *   devices / ndevices   device ordinals; ndevices = 0: every visible device
 *   input / output       host buffers, n and n + 28 bytes, valid until the block has been waited for
 * add() returns the block's number (0, 1, 2, ... in call order) or a negative code; it never blocks on the GPU.  One thread adds;
 * wait() may be called from another thread (for blocks that have been added); destroy() finishes what is queued, then frees every
 * context.  */
typedef struct bscgpu_job bscgpu_job;
BSCGPU_API int  bscgpu_job_create(bscgpu_job** job, const int* devices, int ndevices, int contexts_per_device, int depth, int64_t max_block_bytes);
BSCGPU_API int  bscgpu_job_add(bscgpu_job* job, const uint8_t* input, uint8_t* output, int n, int lzpHashSize, int lzpMinLen,
                               int blockSorter, int coder, int features);
/* Optional: how many blocks the job will have in all (a file's block count).  With the total known the job's LAST blocks are handled
 * for latency instead of throughput — the drain of the pipeline is the caller's time: the k-th context of a device takes a block only
 * while more than k x devices are left (the GPU stages of the tail end one after the other and their host coding overlaps the GPU work
 * still to come, instead of all contexts finishing one last block each in a burst), and the last (devices x contexts) blocks are
 * submitted with BSCGPU_FEATURE_LOW_LATENCY.  Adding more blocks than announced is allowed (the rule is dropped).  Independently of
 * this call the START of a job (and of every later burst, when the caller had let the job run dry) is tapered: a first context begins,
 * the k-th context of a device joins once k GPU stages of the burst have finished there. */
BSCGPU_API int  bscgpu_job_expect(bscgpu_job* job, int total_blocks);
/* (bscgpu_job_wait looks at the worker's pipe while it waits: it must have RETURNED before bscgpu_job_destroy is called on the same
 * job — destroy finishes the queued blocks, then frees the pipes a concurrent wait would still be peeking into) */
BSCGPU_API int  bscgpu_job_wait(bscgpu_job* job, int block);
/* which worker (= pipe; return value) on which device took the block — known once a worker has claimed it */
BSCGPU_API int  bscgpu_job_block_worker(bscgpu_job* job, int block, int* device);
BSCGPU_API void bscgpu_job_destroy(bscgpu_job* job);
/* The executor behind a job's pipes as a table of functions (default: bscgpu_create / bscgpu_pipe_* of this library).  Tests drive
 * the scheduler with a CPU stand-in (tests/test_job_driver.py); semantics of every entry = the bscgpu_* function it stands for. */
typedef struct bscgpu_job_backend {
    void* user;
    int  (*ctx_create)(void* user, void** ctx, int device, int64_t max_n);
    void (*ctx_destroy)(void* user, void* ctx);
    int  (*pipe_create)(void* user, void* ctx, int depth, void** pipe);
    void (*pipe_destroy)(void* user, void* pipe);
    int  (*pipe_submit_host)(void* user, void* pipe, const uint8_t* input, uint8_t* output, int n, int lzpHashSize, int lzpMinLen,
                             int blockSorter, int coder, int features);          /* ticket >= 0 or error */
    int  (*pipe_wait)(void* user, void* pipe, int ticket);
} bscgpu_job_backend;
BSCGPU_API int  bscgpu_job_create_ex(bscgpu_job** job, const int* devices, int ndevices, int contexts_per_device, int depth,
                                     int64_t max_block_bytes, const bscgpu_job_backend* backend /* NULL = this library */);

/* ---- profiling --------------------------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by HIP events on the context's stream (the stream
 * the kernels run on) and accumulated per kernel class. */
enum {
    BSCGPU_K_RADIX_SCATTER = 0, /* the graded kernel: one LSD digit pass (read + scatter) */
    BSCGPU_K_RADIX_HIST    = 1, /* per-chunk digit histogram of the next pass */
    BSCGPU_K_RADIX_SCAN    = 2,
    BSCGPU_K_PACK          = 3, /* key packing (BWT prefix keys / ST context keys) */
    BSCGPU_K_SEG           = 4, /* head flags, rank scans, compaction */
    BSCGPU_K_GATHER        = 5, /* ISA[SA+h] gathers */
    BSCGPU_K_EMIT          = 6, /* BWT / ST output byte emit */
    BSCGPU_K_MISC          = 7,
    BSCGPU_K_DC_CTX        = 8,  /* device coder: contexts, items, setup */
    BSCGPU_K_DC_PART       = 9,  /* device coder: decisions into chain-major order */
    BSCGPU_K_DC_EVAL       = 10, /* device coder: counter chains */
    BSCGPU_K_DC_PSTREAM    = 11, /* device coder: probability stream */
    BSCGPU_K_RADIX_HISTALL = 12, /* single-read sorts: the one histogram read per sort (all digits at once) */
    BSCGPU_K_RADIX_AUX     = 13, /* keys-only passes that also emit the permutation (device coder's orders, inverse BWT): not the graded kernel */
    BSCGPU_K_DC_STATIC     = 14, /* device coder: the context-free counter family walked in stream order (round 6) */
    BSCGPU_K_COUNT         = 15
};
typedef struct bscgpu_kstat {
    double   ms;        /* accumulated HIP-event time */
    uint64_t launches;
    uint64_t bytes;     /* algorithmic bytes moved (per kernel) */
    uint64_t records;   /* records processed (radix kernels) */
} bscgpu_kstat;
BSCGPU_API void bscgpu_profile_enable(bscgpu_ctx* ctx, int on);
BSCGPU_API void bscgpu_profile_reset(bscgpu_ctx* ctx);
BSCGPU_API int  bscgpu_profile_get(bscgpu_ctx* ctx, bscgpu_kstat* stats /* [BSCGPU_K_COUNT] */);
/* Per-launch durations (ms) of the most recent radix scatter launches, newest last; returns count. */
BSCGPU_API int  bscgpu_profile_scatter_launches(bscgpu_ctx* ctx, double* ms, uint64_t* records, int max);
/* Stage wall times (ms) of the last bscgpu_compress_device call: [0] adler, [1] sort transform,
 * [2] D2H, [3] host coder, [4] total; plus doubling rounds in [5]. */
BSCGPU_API int  bscgpu_last_stage_ms(bscgpu_ctx* ctx, double* out6);

BSCGPU_API const char* bscgpu_last_error(const bscgpu_ctx* ctx);

/* ---- measurement / test knobs of one context ------------------------------------------------
 * BSCGPU_OPT_RS_ONESWEEP   which large (key, value) sorts take the single-read digit passes: 0 none (three-kernel passes: histogram,
 *                          scan and a scatter whose offsets are all known before it starts — what bench.py times as the digit pass's
 *                          `pattern_ceiling`), 1 large (key, value) sorts only, 2 every sort of >= 4 tiles (tests), 3 large sorts including keys-only ones (the
 *                          sort transform; the default).
 *                          Results are identical.
 * BSCGPU_CNT_OS_RETRIES    (get only) transforms this context has redone through the three-kernel passes because a single-read pass
 *                          gave up a wait (bounded polls; the block still comes out right).
 * BSCGPU_OPT_DC_STREAM_STATIC  device model of the static coder: 1 (BSC_DC_SPF=1 in the environment) evaluates the context-free
 *                          counter family in stream order for blocks of at most 32 symbols per sub-block (devcoder_static.h);
 *                          0 (default) sends it through partition / evaluation / gather like the other two families.  Same
 *                          output either way; round 6 measured the stream-order form slower (profiles/r06/static_family_stream_order.txt).
 * BSCGPU_OPT_DC_PACKED_STREAM  1 (default; BSC_PS13=0 in the environment turns it off): the static coder's probability stream crosses
 *                          PCIe as 13 bits per decision — 12-bit probability + coded bit, eight decisions in 13 bytes — instead of
 *                          16-bit entries (298 instead of 366 MB per 64 MiB text block).  The run-start mark of the 16-bit entry only
 *                          placed the reference's output-budget test; a stream that reaches its budget is redone on the host model
 *                          either way.  Same output.
 * set returns the previous value or a negative libbsc error code; get the value or a negative error code. */
enum { BSCGPU_OPT_RS_ONESWEEP = 1, BSCGPU_CNT_OS_RETRIES = 2, BSCGPU_OPT_DC_STREAM_STATIC = 3, BSCGPU_OPT_DC_PACKED_STREAM = 4 };
BSCGPU_API int bscgpu_option_set(bscgpu_ctx* ctx, int key, int value);
BSCGPU_API int bscgpu_option_get(bscgpu_ctx* ctx, int key);
/* Process-wide counts since start (tests, reports): blocks whose static model ran on the GPU, how many of those were LZP-preprocessed,
 * and blocks that took the device model first and were redone with the model on the host (a sub-block that does not compress has to
 * be stored raw from the run arrays, which the device path never copies). */
enum { BSCGPU_PCNT_DEVICE_MODEL_BLOCKS = 1, BSCGPU_PCNT_REDONE_ON_HOST_MODEL = 2, BSCGPU_PCNT_DEVICE_MODEL_LZP_BLOCKS = 3 };
BSCGPU_API long long bscgpu_process_counter(int key);

/* ---- how the libbsc.h entry points spread concurrent callers over the GPUs of a node (pure functions, no GPU needed) -------------
 * The host-pointer API keeps ctx_per_dev default contexts per physical device = nphys * ctx_per_dev logical slots; slot s lives
 * on device s % nphys, so every GPU's first context comes before anybody's second one.  A call takes the usable slot whose GPU has
 * the fewest calls in flight (then the slot with the fewest, then a slot marked usable = 2 — its context exists and is large
 * enough — before one that would have to be created; remaining ties in round-robin order from `start`).  The reference has one
 * lock and one device (bwt.cpp:50-52, st.cu:56); its parallelism is the CLI's OpenMP team of concurrent bsc_compress calls
 * (bsc.cpp:184-199), which this rule maps to one block per GPU.  Returns the slot, -1 when no slot is usable. */
BSCGPU_API int bscgpu_dispatch_device(int slot, int nphys);
BSCGPU_API int bscgpu_dispatch_pick(int nphys, int ctx_per_dev, const int* users, const unsigned char* usable, unsigned start);

#ifdef __cplusplus
}
#endif
#endif /* BSCGPU_H */
