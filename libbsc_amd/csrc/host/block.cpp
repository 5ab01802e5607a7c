// block.cpp — the libbsc block container and public C API on top of the GPU sorters and the host coder.
//
// Follows the container semantics of libbsc/libbsc/libbsc.cpp: bsc_store :68-81, bsc_compress :213-338
// (in-place twin :83-211), bsc_block_info :340-418, bsc_decompress :522-617; header layout
// [0]blockSize [4]dataSize [8]mode [12]index [16]adler(data) [20]adler(payload) [24]adler(header[0..24)),
// trailer = indexes[num] + num byte.  Stage entry points mirror bwt.cpp:178, st.cpp:990, coder.cpp:244.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>
#include <algorithm>
#include <utility>

#include <sched.h>
#include <hip/hip_runtime.h>

#include "../../../include/libbsc.h"
#include "../../../include/bscgpu.h"
#include "../device/dev_common.h"
#include "../device/dma_copy.h"
#include "qlfc.h"
#include "lzp.h"
#include "par.h"

using namespace bschost;

// ---- allocator hooks (platform.cpp:173-190) ---------------------------------------------------------
static void* (*g_malloc)(size_t) = nullptr;
static void* (*g_zero_malloc)(size_t) = nullptr;
static void  (*g_free)(void*) = nullptr;
static void* bsc_malloc(size_t n) { return g_malloc ? g_malloc(n) : malloc(n); }
static void  bsc_free(void* p) { if (g_free) g_free(p); else free(p); }

// ---- process-wide default GPU contexts for the host-pointer API: one per visible device ----------------------------
// The reference keeps one cached arena behind one lock (bwt.cpp:50-52) and gets its parallelism from the CLI's OpenMP team
// calling bsc_compress concurrently, one block per thread (bsc.cpp:184-199).  Relinked against this library those calls
// are spread over ALL visible GPUs with no API change: every device has its own default contexts (two by default: their
// kernels interleave), GPU-stage locks and pinned slots.  Logical slot s lives on physical device s mod (number of devices), so
// the first context of every GPU comes before anybody's second one, and a call goes to the slot whose GPU has the fewest calls
// in flight (then the slot with the fewest; ties: round robin): N concurrent callers on an N-GPU node run one block per GPU, 2N
// callers two per GPU (bscgpu_dispatch_pick / bscgpu_dispatch_device below are that rule as pure functions, unit-tested on CPU).  BSC_GPU_DEVICE=<k> pins everything to device k; BSC_GPU_DEVICES=<n> uses the first n.
// The GPU stage of a call is serialised per device, but bsc_compress releases the lock before its host stage, so up to
// DEFAULT_SLOTS calls overlap per device — one on the GPU, the others coding on host threads.  A device's context is only
// re-created (for a larger block) when nobody is using it.
constexpr int       DEFAULT_SLOTS = 3;
constexpr int       MAX_DEVICES = 32;           // logical: contexts_per_device x physical devices
struct DefaultDevice {
    std::mutex   gpu_lock;                       // the GPU stage on this device
    bscgpu_ctx*  ctx = nullptr;
    int64_t      cap = 0;
    int          users = 0;
    bool         slot_busy[DEFAULT_SLOTS] = {false, false, false};
    int64_t      no_memory_for = -1;             // >= 0: creating this slot's context for a block this large failed for lack of HBM
};
static std::mutex   g_user_mu;                  // every DefaultDevice's ctx / cap / users / slot_busy, g_ndev, g_rr
static std::condition_variable g_user_cv;
static DefaultDevice g_dev[MAX_DEVICES];
static int          g_ndev = -1;                // logical devices the default path uses (-1: not probed yet)
static int          g_dev_first = 0;
static int          g_ctx_per_dev = 2;          // contexts per physical device: the kernels of two blocks interleave on the GPU and fill
                                                // the SIMDs that one block's serial chains leave idle (+14 % whole-job rate; BSC_GPU_CONTEXTS)
static unsigned     g_rr = 0;

// ---- the dispatch rule, as pure functions --------------------------------------------------------------------------
// Logical slots 0 .. nphys * ctx_per_dev - 1; slot s runs on physical device s % nphys.
extern "C" BSCGPU_API int bscgpu_dispatch_device(int slot, int nphys) { return nphys > 0 ? slot % nphys : 0; }
// users[s] = calls in flight on slot s; usable[s] != 0 when slot s can take this call now, 2 when its context already exists and is
// large enough (preferred among equally loaded slots: a lone caller then stays on one context instead of alternating between two and
// paying for two arenas, two sets of pinned buffers and every resize twice); start = round-robin cursor.
// Returns the slot to use, -1 when none is usable.
extern "C" BSCGPU_API int bscgpu_dispatch_pick(int nphys, int ctx_per_dev, const int* users, const unsigned char* usable, unsigned start)
{
    const int nslots = nphys * ctx_per_dev;
    if (nslots <= 0) return -1;
    int best = -1, best_dev_load = 0;
    for (int k = 0; k < nslots; ++k) {
        const int s = (int)((start + (unsigned)k) % (unsigned)nslots);
        if (!usable[s]) continue;
        int dev_load = 0;
        for (int q = s % nphys; q < nslots; q += nphys) dev_load += users[q];
        if (best < 0 || dev_load < best_dev_load ||
            (dev_load == best_dev_load && (users[s] < users[best] || (users[s] == users[best] && usable[s] > usable[best])))) { best = s; best_dev_load = dev_load; }
    }
    return best;
}

static int probe_devices_locked()
{
    if (g_ndev >= 0) return g_ndev;
    int n = bscgpu_device_count();
    if (n > MAX_DEVICES) n = MAX_DEVICES;
    g_dev_first = 0;
    if (const char* e = getenv("BSC_GPU_DEVICE")) { const int d = atoi(e); if (d >= 0 && d < n) { g_dev_first = d; n = 1; } else n = 0; }
    else if (const char* e2 = getenv("BSC_GPU_DEVICES")) { const int k = atoi(e2); if (k >= 1 && k < n) n = k; }
    if (const char* e3 = getenv("BSC_GPU_CONTEXTS")) { const int k = atoi(e3); if (k >= 1 && k <= 4) g_ctx_per_dev = k; }
    n *= g_ctx_per_dev;
    if (n > MAX_DEVICES) n = MAX_DEVICES / g_ctx_per_dev * g_ctx_per_dev;
    g_ndev = n;
    return n;
}

// Register as a user of a default context that can take n bytes; with want_slot also reserve a pinned slot.
static int default_gpu_acquire(int64_t n, bool want_slot, DefaultDevice** out, int* slot_out)
{
    std::unique_lock<std::mutex> lk(g_user_mu);
    const int ndev = probe_devices_locked();
    if (ndev <= 0) return LIBBSC_GPU_NOT_SUPPORTED;
    for (;;) {
        // the slot whose GPU is least loaded among those that can take the call right now (bscgpu_dispatch_pick)
        const int nphys = ndev / g_ctx_per_dev;
        int users[MAX_DEVICES]; unsigned char usable[MAX_DEVICES];
        bool any_ctx_fits = false;
        for (int d = 0; d < ndev; ++d) {
            DefaultDevice& D = g_dev[d];
            bool slot_free = !want_slot;
            for (int i = 0; i < DEFAULT_SLOTS && !slot_free; ++i) slot_free = !D.slot_busy[i];
            const bool fits = D.ctx && D.cap >= n;
            any_ctx_fits = any_ctx_fits || fits;
            users[d] = D.users;
            usable[d] = D.no_memory_for >= 0 && n >= D.no_memory_for ? 0 : (fits ? (slot_free ? 2 : 0) : (D.users == 0 ? 1 : 0));   // an idle slot can be (re)sized
        }
        const int best = bscgpu_dispatch_pick(nphys, g_ctx_per_dev, users, usable, g_rr);
        if (best >= 0) {
            DefaultDevice& D = g_dev[best];
            if (!(D.ctx && D.cap >= n)) {
                if (D.ctx) {
                    bscgpu_destroy(D.ctx); D.ctx = nullptr; D.cap = 0;
                    // HBM came back on this physical device: slots that had been taken out of the draw for lack of it are in again
                    for (int q = best % nphys; q < ndev; q += nphys) g_dev[q].no_memory_for = -1;
                }
                const int64_t cap = n + n / 32 + 4096;                    // headroom like bwt.cpp:106
                const int rc = bscgpu_create(&D.ctx, g_dev_first + bscgpu_dispatch_device(best, nphys), cap);
                if (rc != LIBBSC_NO_ERROR) {
                    D.ctx = nullptr;
                    // No HBM for one more arena of this size: if some other context can take the block, this slot is taken out
                    // of the draw for blocks this large and the caller queues for an existing context (large concurrent blocks
                    // queued before there were several contexts per GPU, too); otherwise the error is the caller's.
                    if (rc == LIBBSC_GPU_NOT_ENOUGH_MEMORY && any_ctx_fits) { D.no_memory_for = n; continue; }
                    return rc;
                }
                D.cap = cap;
                if (D.no_memory_for >= 0 && cap >= D.no_memory_for) D.no_memory_for = -1;     // it fits after all
            }
            int s = -1;
            if (want_slot) { for (int i = 0; i < DEFAULT_SLOTS; ++i) if (!D.slot_busy[i]) { s = i; break; } D.slot_busy[s] = true; }
            ++D.users;
            g_rr = (unsigned)best + 1u;
            *out = &D; if (slot_out) *slot_out = s;
            return LIBBSC_NO_ERROR;
        }
        // Nothing usable right now.  Somebody in flight will release a slot: wait.  Nobody in flight: nothing can change by itself — the
        // slots are all marked "no memory for this size" by an earlier failure — so the marks are dropped and creation is tried again
        // (the HBM may be free again: other processes, destroyed pipes); a second failure is the caller's error.
        bool in_flight = false, marked = false;
        for (int d = 0; d < ndev; ++d) { in_flight = in_flight || g_dev[d].users > 0; marked = marked || g_dev[d].no_memory_for >= 0; }
        if (!in_flight) {
            if (!marked) return LIBBSC_GPU_NOT_ENOUGH_MEMORY;
            for (int d = 0; d < ndev; ++d) g_dev[d].no_memory_for = -1;
            continue;
        }
        g_user_cv.wait(lk);
    }
}
static void default_gpu_release(DefaultDevice* D, int slot)
{
    { std::lock_guard<std::mutex> lk(g_user_mu); if (slot >= 0) D->slot_busy[slot] = false; --D->users; }
    g_user_cv.notify_all();
}
struct DefaultGpuUser {                         // RAII around acquire / release
    DefaultDevice* dev = nullptr; bscgpu_ctx* c = nullptr; int slot = -1; int rc;
    DefaultGpuUser(int64_t n, bool want_slot) { rc = default_gpu_acquire(n, want_slot, &dev, &slot); if (rc == LIBBSC_NO_ERROR) c = dev->ctx; }
    ~DefaultGpuUser() { if (rc == LIBBSC_NO_ERROR) default_gpu_release(dev, slot); }
};

static inline void put_i32(unsigned char* p, int v) { memcpy(p, &v, 4); }
static inline int  get_i32(const unsigned char* p) { int v; memcpy(&v, p, 4); return v; }

static int aux_rate(int n)          // largest power of two <= n / 8 (>= 1), bwt.cpp:192-197
{
    int mod = n / 8;
    mod |= mod >> 1; mod |= mod >> 2; mod |= mod >> 4; mod |= mod >> 8; mod |= mod >> 16; mod >>= 1;
    return mod + 1;
}

extern "C" {

int bsc_init_full(int features, void* (*malloc_fn)(size_t), void* (*zero_malloc_fn)(size_t), void (*free_fn)(void*))
{
    (void)features;
    g_malloc = malloc_fn; g_zero_malloc = zero_malloc_fn; g_free = free_fn;
    (void)qlfc_tables();
    return LIBBSC_NO_ERROR;
}
int bsc_init(int features) { return bsc_init_full(features, nullptr, nullptr, nullptr); }
int bsc_bwt_init(int) { return LIBBSC_NO_ERROR; }
int bsc_st_init(int) { return LIBBSC_NO_ERROR; }
int bsc_coder_init(int) { (void)qlfc_tables(); return LIBBSC_NO_ERROR; }

unsigned int bsc_adler32(const unsigned char* T, int n, int) { return adler32(T, n < 0 ? 0 : (size_t)n); }

int bsc_coder_compress(const unsigned char* in, unsigned char* out, int n, int coder, int features)
{ return coder_compress(in, out, n, coder, features); }
int bsc_coder_decompress(const unsigned char* in, unsigned char* out, int coder, int features)
{ return coder_decompress(in, out, coder, features); }
int bsc_qlfc_encode_block(const unsigned char* in, unsigned char* out, int inSize, int outSize, int coder)
{ return qlfc_encode_block(in, out, inSize, outSize, coder); }
int bsc_qlfc_decode_block(const unsigned char* in, unsigned char* out, int coder)
{ return qlfc_decode_block(in, out, coder); }
int bsc_qlfc_ranks(const unsigned char* in, int n, unsigned char* ranks, unsigned char* firstSeen, int* pK)
{
    QlfcRuns R;
    qlfc_runs(in, n, R);
    memcpy(ranks, R.rank.data(), R.rank.size());
    memcpy(firstSeen, R.view.first_seen, (size_t)R.view.nsym);
    if (pK) *pK = R.view.nsym;
    return (int)R.rank.size();
}

// ---- sorters: GPU only ---------------------------------------------------------------------------------
int bsc_bwt_encode(unsigned char* T, int n, unsigned char* num_indexes, int* indexes, int features)
{
    (void)features;
    if (T == nullptr || n < 0) return LIBBSC_BAD_PARAMETER;
    // n = 0: the reference hands libsais an empty text; with the secondary indexes asked for, r = 1 is rejected first (libsais.c:6711)
    if (n == 0) return (num_indexes != nullptr && indexes != nullptr) ? LIBBSC_BAD_PARAMETER : 0;
    if (num_indexes != nullptr && indexes != nullptr && aux_rate(n) < 2) return LIBBSC_BAD_PARAMETER;     // (no GPU needed to say so)
    DefaultGpuUser user(n, false);
    if (user.rc != LIBBSC_NO_ERROR) return user.rc;
    bscgpu_ctx* c = user.c;
    std::lock_guard<std::mutex> g(user.dev->gpu_lock);
    if (num_indexes != nullptr && indexes != nullptr) {
        const int r = aux_rate(n);
        if (r < 2) return LIBBSC_BAD_PARAMETER;               // libsais_bwt_aux rejects r < 2 (libsais.c:6711)
        uint32_t I[256];
        const int cnt = (n - 1) / r;
        if (cnt + 1 > 256) return LIBBSC_BAD_PARAMETER;
        int64_t res = bscgpu_bwt_aux(c, T, T, n, r, I);
        if (res < 0) return (int)res;
        *num_indexes = (unsigned char)cnt;
        for (int t = 0; t < cnt; ++t) indexes[t] = (int)I[t + 1] - 1;     // bwt.cpp:205-209
        return (int)I[0];
    }
    return (int)bscgpu_bwt(c, T, T, n);
}

// inverse BWT on the default GPU context (decode.cpp calls this for large blocks; < 0 other than DATA_CORRUPT = not available,
// the caller continues on the host)
int bsc_bwt_decode_gpu(unsigned char* T, int n, int index)
{
    DefaultGpuUser user(n, false);
    if (user.rc != LIBBSC_NO_ERROR) return user.rc;
    std::lock_guard<std::mutex> g(user.dev->gpu_lock);
    return bscgpu_unbwt(user.c, T, T, n, index);
}

int bsc_st_encode(unsigned char* T, int n, int k, int features)
{
    (void)features;
    if (T == nullptr || n < 0) return LIBBSC_BAD_PARAMETER;
    if (k < 3 || k > 8) return LIBBSC_BAD_PARAMETER;
    if (n <= 1) return 0;
    DefaultGpuUser user(n, false);
    if (user.rc != LIBBSC_NO_ERROR) return user.rc;
    std::lock_guard<std::mutex> g(user.dev->gpu_lock);
    return bscgpu_st_encode(user.c, T, n, k);
}

// ---- container ---------------------------------------------------------------------------------------------
int bsc_store(const unsigned char* input, unsigned char* output, int n, int)
{
    const unsigned a = adler32(input, (size_t)n);
    memmove(output + LIBBSC_HEADER_SIZE, input, (size_t)n);
    put_i32(output + 0, n + LIBBSC_HEADER_SIZE);
    put_i32(output + 4, n);
    put_i32(output + 8, 0);
    put_i32(output + 12, 0);
    put_i32(output + 16, (int)a);
    put_i32(output + 20, (int)a);
    put_i32(output + 24, (int)adler32(output, 24));
    return n + LIBBSC_HEADER_SIZE;
}

static int make_mode(int sorter, int coder, int lzpHashSize, int lzpMinLen, int* mode_out)
{
    int mode;
    if (sorter == LIBBSC_BLOCKSORTER_BWT || (sorter >= LIBBSC_BLOCKSORTER_ST3 && sorter <= LIBBSC_BLOCKSORTER_ST8)) mode = sorter;
    else return LIBBSC_BAD_PARAMETER;
    if (coder < LIBBSC_CODER_QLFC_STATIC || coder > LIBBSC_CODER_QLFC_FAST) return LIBBSC_BAD_PARAMETER;
    mode += coder << 5;
    if (lzpMinLen != 0 || lzpHashSize != 0) {
        if (lzpMinLen < 4 || lzpMinLen > 255) return LIBBSC_BAD_PARAMETER;
        if (lzpHashSize < 10 || lzpHashSize > 28) return LIBBSC_BAD_PARAMETER;
        mode += (lzpMinLen << 8) + (lzpHashSize << 16);
    }
    *mode_out = mode;
    return LIBBSC_NO_ERROR;
}

int bsc_block_info(const unsigned char* hdr, int headerSize, int* pBlockSize, int* pDataSize, int)
{
    if (headerSize < LIBBSC_HEADER_SIZE) return LIBBSC_UNEXPECTED_EOB;
    if ((unsigned)get_i32(hdr + 24) != adler32(hdr, 24)) return LIBBSC_DATA_CORRUPT;
    const int blockSize = get_i32(hdr + 0), dataSize = get_i32(hdr + 4), mode = get_i32(hdr + 8), index = get_i32(hdr + 12);
    const int lzpHashSize = (mode >> 16) & 0xff, lzpMinLen = (mode >> 8) & 0xff, coder = (mode >> 5) & 0x7, sorter = mode & 0x1f;

    int test = 0;
    if (sorter == LIBBSC_BLOCKSORTER_BWT || (sorter >= LIBBSC_BLOCKSORTER_ST3 && sorter <= LIBBSC_BLOCKSORTER_ST8)) test = sorter;
    else if (sorter > 0) return LIBBSC_DATA_CORRUPT;
    if (coder >= LIBBSC_CODER_QLFC_STATIC && coder <= LIBBSC_CODER_QLFC_FAST) test += coder << 5;
    else if (coder > 0) return LIBBSC_DATA_CORRUPT;
    if (lzpMinLen != 0 || lzpHashSize != 0) {
        if (lzpMinLen < 4 || lzpMinLen > 255) return LIBBSC_DATA_CORRUPT;
        if (lzpHashSize < 10 || lzpHashSize > 28) return LIBBSC_DATA_CORRUPT;
        test += (lzpMinLen << 8) + (lzpHashSize << 16);
    }
    if (test != mode) return LIBBSC_DATA_CORRUPT;
    if (blockSize < LIBBSC_HEADER_SIZE || blockSize > LIBBSC_HEADER_SIZE + dataSize) return LIBBSC_DATA_CORRUPT;
    if (index < 0 || index > dataSize) return LIBBSC_DATA_CORRUPT;
    if (pBlockSize) *pBlockSize = blockSize;
    if (pDataSize) *pDataSize = dataSize;
    return LIBBSC_NO_ERROR;
}

// ---- GPU-resident compress: Adler-32 + sorter + QLFC front end on the device, QLFC coding on host threads ----
// One block = a GPU stage (runs on the submitting thread, serialised per context) and a host stage (worker thread).
struct BlockJob {
    bscgpu_ctx* c = nullptr;
    const void* dInput = nullptr;
    const uint8_t* hInput = nullptr; // host copy of the original block (bsc_compress); null for device-resident callers
    uint8_t*    output = nullptr;
    int n_orig = 0;                  // size of the original block; n below is what the sorter sees (LZP output or the same)
    bool have_adler = false, inplace = false;
    std::unique_ptr<unsigned char, void (*)(void*)> lz{nullptr, nullptr};   // LZP output while the block is in flight
    int n = 0, coder = 0, features = 0, mode = 0, index = 0, num_indexes = 0, nblocks = 0;
    int indexes[256];
    uint32_t adler_data = 0;
    int start[8], size[8];
    u32 run_first[9];
    u32 first_run[8 * 256];
    HostSlot* slot = nullptr;        // pinned landing zones of this block (owned by the context)
    // device-side static model (devcoder.hip): the host codes from a probability stream instead of run arrays
    bool pipelined = false;        // submitted through a pipe (several blocks in flight): throughput over latency
    int  pipe_workers = 0;         // coder threads of the process-wide pool (0: a synchronous call, which starts its own threads)
    int  pool_free = -1;           // CPUs of the pool's budget with nothing to do when the block was queued (-1: not a pipe's block)
    int  tail_r = -1;              // blocks of the announced job whose GPU stage was still to END when this one's did (-1: no job announced)
    int  tail_gpus = 1;            // ... and how many GPUs feed the pool in that job
    int  tail_pipes = 6;           // ... and how many pipes (GPU contexts) per GPU: that many blocks' GPU stages overlap
    int  ps_g = 2;                 // device-model sub-blocks per coder task (ps_group), fixed when the block's host work starts
    bool use_ps = false; const uint16_t* ps = nullptr; u32 poff[9]; u32 ndec = 0; int sorter = 0;
    bool ps_packed = false; u32 pbase[9];          // the stream is the 13-bit packed form (devcoder.hip DcP13): sub-block b's starts at decision pbase[b] of the packed space (a multiple of 64)
    // sub-block b's stream as the coders take it: 16-bit entries, or the packed bytes behind the same pointer type
    const uint16_t* ps_of(int b) const { return ps_packed ? reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(ps) + (size_t)pbase[b] / 8u * 13u) : ps + poff[b]; }
    hipEvent_t ps_ready = nullptr;   // the p stream's copy to the host (copy stream), all of it
    hipEvent_t ps_part[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // ... up to and including sub-block b: what a coder task waits on
    bool ps_dma = false; uint64_t ps_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the pieces went through the DMA engine directly (dma_copy.h): signals instead of events
    bool ps_landed(int b) const { return ps_dma ? dma_wait(ps_sig[b]) == 0 : hipEventSynchronize(ps_part[b]) == hipSuccess; }
    std::atomic<bool> redo{false};   // a sub-block did not compress: the block goes through the host model again (raw sub-blocks need the run arrays)
    bool stored_small = false;       // n <= header size: finished in the GPU stage
    int  result = 0;
    // host stage, split into per-sub-block tasks for the pipe's worker pool
    RunView views[8];
    std::unique_ptr<uint8_t[]> scratch[8];          // coded sub-blocks; kept across the blocks of a pipe lane (no re-faulting)
    size_t scratch_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int  sub_res[8];
    std::atomic<int> remaining{0};
    bool done = false;
};

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// BSC_DEVICE_CODER=0 keeps the static coder's model on the host; BSC_DEVICE_CODER_MIN_N: smallest block that takes the device model
static bool devcoder_enabled() { static const int v = [] { const char* e = getenv("BSC_DEVICE_CODER"); return e ? atoi(e) : 1; }(); return v != 0; }
static int  devcoder_min_n() { static const int v = [] { const char* e = getenv("BSC_DEVICE_CODER_MIN_N"); return e ? atoi(e) : (1 << 20); }(); return v; }

static std::atomic<uint64_t> g_count_devmodel{0}, g_count_redo{0}, g_count_devmodel_lzp{0};     // bscgpu_process_counter

// Experiment (round 6, profiles/r06/sort_slots.txt): the sort is the HBM-bound half of a block's GPU stage, the device model the
// latency / VALU-bound half; contexts whose sorts overlap only share HBM, a sort beside another block's model overlaps for real.
// BSC_SORT_SLOTS=k lets at most k contexts of the process be inside their sort at once (0 = no limit, the default).
struct SortSlot {
    static int limit() { static const int v = [] { const char* e = getenv("BSC_SORT_SLOTS"); return e ? atoi(e) : 0; }(); return v; }
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::condition_variable& cv() { static std::condition_variable c; return c; }
    static int& busy() { static int b = 0; return b; }
    bool held = false;
    SortSlot() { if (limit() > 0) { std::unique_lock<std::mutex> lk(mu()); cv().wait(lk, [] { return busy() < limit(); }); ++busy(); held = true; } }
    void release() { if (held) { { std::lock_guard<std::mutex> lk(mu()); --busy(); } cv().notify_one(); held = false; } }
    ~SortSlot() { release(); }
};

static int gpu_stage(BlockJob& J, int blockSorter, bool allow_devcoder = true)
{
    CtxTimer tm_stage("gpu_stage of one block");
    J.sorter = blockSorter; J.use_ps = false; J.ps_packed = false; J.redo.store(false, std::memory_order_relaxed);
    bscgpu_ctx* c = J.c;
    const int n = J.n;
    if (hipSetDevice(c->device) != hipSuccess) return LIBBSC_GPU_ERROR;
    if (n <= LIBBSC_HEADER_SIZE) {
        unsigned char tmp[LIBBSC_HEADER_SIZE + 1];
        if (n > 0 && hipMemcpy(tmp, J.dInput, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return LIBBSC_GPU_ERROR;
        J.result = bsc_store(tmp, J.output, n, J.features);
        J.stored_small = true;
        return LIBBSC_NO_ERROR;
    }
    auto t0 = clk::now();
    int rc = J.have_adler ? LIBBSC_NO_ERROR : adler32_device(c, (const u8*)J.dInput, n, &J.adler_data);
    if (rc < 0) return rc;
    c->stage_ms[0] = ms_since(t0);

    t0 = clk::now();
    J.num_indexes = 0;
    SortSlot sort_slot;                                       // (experiment, BSC_SORT_SLOTS=k: at most k contexts of the process inside their sort at once)
    if (blockSorter == LIBBSC_BLOCKSORTER_BWT) {
        const int r = aux_rate(n);
        uint32_t I[256];
        int64_t primary = 0;
        rc = bwt_device(c, (const u8*)J.dInput, c->dL, n, r, I, &primary);
        if (rc < 0) return rc;
        J.index = (int)primary;
        J.num_indexes = (n - 1) / r;
        for (int t = 0; t < J.num_indexes; ++t) J.indexes[t] = (int)I[t + 1] - 1;
    } else {
        rc = st_device(c, (const u8*)J.dInput, c->dL, n, blockSorter, &J.index);
        if (rc < 0) return rc;
    }
    if (J.n_orig < 64 * 1024) J.num_indexes = 0;              // libbsc.cpp:290 (the original size decides)
    sort_slot.release();
    c->stage_ms[1] = ms_since(t0);

    // QLFC front half on the GPU (sub-block split, runs, ranks); only the run arrays (+ L for the rare raw
    // sub-block) cross PCIe, into this job's pinned slot.
    t0 = clk::now();
    J.nblocks = coder_num_blocks(n);
    rc = qlfc_front_split(c, c->dL, (u32)n, J.nblocks, J.start, J.size);
    if (rc < 0) return rc;
    u32 m = 0;
    // static coder on a block with several sub-blocks: the adaptive model runs on the GPU too (devcoder.hip) and only the
    // probability stream crosses PCIe; anything that path declines falls back to the run arrays + host model
    // (an LZP-preprocessed block — the reference CLI's default, bsc.cpp:73-75 — is just another byte block to the sorter and the model;
    // its LZP output stays alive until the block is done, because a redo on the host model uploads it again)
    // (the fast coder, -e0, runs on the same device machinery: its one counter per decision is the static coder's char family with
    // other update maps — devcoder.hip: devcoder_pstream_fast; BSC_DEVICE_CODER_FAST=0 keeps it on the host)
    static const bool dc_fast = [] { const char* e = getenv("BSC_DEVICE_CODER_FAST"); return e ? atoi(e) != 0 : true; }();
    const bool try_dc = allow_devcoder && devcoder_enabled() && (J.coder == LIBBSC_CODER_QLFC_STATIC || (J.coder == LIBBSC_CODER_QLFC_FAST && dc_fast))
                     && J.nblocks > 1 && n >= devcoder_min_n();
    rc = qlfc_front_runs(c, c->dL, (u32)n, J.nblocks, J.start, &m, J.run_first, J.first_run, *J.slot, !try_dc);
    if (rc < 0) return rc;
    if (try_dc) {
        bool ok = false;
        if ((double)m <= 0.70 * (double)n) {                        // nearly one run per byte: the sub-blocks will be stored raw anyway
            int maxr[8];
            for (int b = 0; b < J.nblocks; ++b) {
                int nsym = 0;
                for (int sy = 0; sy < 256; ++sy) nsym += J.first_run[b * 256 + sy] != 0xffffffffu;
                int k = 0; while ((2 << k) <= nsym - 1) ++k;           // bsr(nsym - 1), 0 for nsym <= 2 (qlfc.cpp:888)
                maxr[b] = nsym >= 2 ? k : 0;
            }
            u32 ndec = 0;
            const int pb = c->ps_toggle;
            int packed = 0;
            const int r2 = devcoder_pstream(c, reinterpret_cast<const u8*>(c->vA), reinterpret_cast<const u8*>(c->vB), c->SA, m, (u32)n, J.nblocks,
                                            J.run_first, maxr, &ndec, J.poff, nullptr, pb, J.coder, &packed);
            // byte range of every sub-block's piece in the device buffer = in the landing zone (the same layout on both sides)
            size_t piece_lo[8], piece_hi[8], zone_entries = (size_t)ndec + 64;
            J.ps_packed = packed != 0;
            if (r2 == LIBBSC_NO_ERROR) {
                u32 pb13 = 0;
                for (int b = 0; b < J.nblocks; ++b) {
                    const u32 cnt = J.poff[b + 1] - J.poff[b];
                    J.pbase[b] = pb13;
                    if (J.ps_packed) { piece_lo[b] = (size_t)pb13 / 8u * 13u; piece_hi[b] = piece_lo[b] + ((size_t)cnt + 7u) / 8u * 13u; pb13 += (cnt + 63u) / 64u * 64u; }
                    else             { piece_lo[b] = (size_t)J.poff[b] * 2u; piece_hi[b] = (size_t)J.poff[b + 1] * 2u; }
                }
                J.pbase[J.nblocks] = pb13;
                if (J.ps_packed) zone_entries = ((size_t)pb13 / 8u * 13u + 1u) / 2u + 64;
            }
            // (a pinned landing zone that cannot be had is a reason to take the host model, like an arena that does not fit)
            if (r2 == LIBBSC_NO_ERROR && ctx_ensure_pstream_slot(c, *J.slot, zone_entries) == LIBBSC_NO_ERROR) {
                // the stream has been synchronised behind the last kernel; the copy goes to the copy stream and is NOT waited for
                // here: the next block's sort overlaps it, the coder tasks wait on the event
                // (sub-block by sub-block, an event behind each piece: the task that codes sub-blocks b.. starts when ITS entries have landed —
                // 366 MB take 7-14 ms over PCIe, which the last block of a job and every synchronous call used to wait out in full)
                // Which engine moves them: hipMemcpyAsync is a DMA-engine copy on the system's HIP runtime but a 256-workgroup copy KERNEL on
                // the runtime a torch process carries (dma_copy.h), so the pieces are handed to the HSA runtime's DMA copy directly and
                // complete HSA signals; hipMemcpyAsync + events remain for a process where that is not possible (or BSC_D2H_DMA=0).
                bool dma = dma_available() && J.slot->hps_dev != nullptr;
                for (int b = 0; b < 8 && dma; ++b) dma = J.slot->part_sig[b] != 0;
                J.ps_dma = false;
                if (dma) {
                    const uint8_t* dps = reinterpret_cast<const uint8_t*>(devcoder_pstream_ptr(c, pb));
                    uint8_t* hdev = (uint8_t*)J.slot->hps_dev;
                    int issued = 0;
                    for (; issued < J.nblocks; ++issued) {
                        const size_t lo = piece_lo[issued], hi = piece_hi[issued];
                        if (dma_d2h(hdev + lo, dps + lo, hi - lo, J.slot->part_sig[issued]) != 0) break;
                        J.ps_sig[issued] = J.slot->part_sig[issued];
                    }
                    if (issued == J.nblocks) {
                        J.ps_dma = true;
                        for (int b = 0; b < 8; ++b) c->ps_guard_sig[pb][b] = b < J.nblocks ? J.slot->part_sig[b] : 0;
                        c->ps_guard[pb] = nullptr; J.ps_ready = nullptr;
                    } else {
                        for (int b = 0; b < issued; ++b) (void)dma_wait(J.slot->part_sig[b]);       // what was queued lands first; then the whole stream again through HIP
                    }
                }
                if (!J.ps_dma) {
                    const uint8_t* dps = reinterpret_cast<const uint8_t*>(devcoder_pstream_ptr(c, pb));
                    for (int b = 0; b < J.nblocks; ++b) {
                        const size_t lo = piece_lo[b], hi = piece_hi[b];
                        if ((hi > lo && hipMemcpyAsync((uint8_t*)J.slot->hps + lo, dps + lo, hi - lo, hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess) ||
                            hipEventRecord(J.slot->part_ev[b], c->copy_stream) != hipSuccess) return LIBBSC_GPU_ERROR;
                        J.ps_part[b] = J.slot->part_ev[b];
                    }
                    if (hipEventRecord(J.slot->copy_ev, c->copy_stream) != hipSuccess) return LIBBSC_GPU_ERROR;
                    c->ps_guard[pb] = J.slot->copy_ev; J.ps_ready = J.slot->copy_ev;
                    for (int b = 0; b < 8; ++b) c->ps_guard_sig[pb][b] = 0;
                }
                c->ps_toggle = pb ^ 1;
                J.use_ps = true; J.ps = J.slot->hps; J.ndec = ndec; ok = true;
                g_count_devmodel.fetch_add(1, std::memory_order_relaxed);
                if (J.lz) g_count_devmodel_lzp.fetch_add(1, std::memory_order_relaxed);
            } else if (r2 != LIBBSC_NOT_SUPPORTED && r2 != LIBBSC_NO_ERROR) return r2;
        }
        if (!ok) { rc = qlfc_front_copy_runs(c, m, *J.slot); if (rc < 0) return rc; }
    }
    c->stage_ms[2] = ms_since(t0);
    return LIBBSC_NO_ERROR;
}

static void host_prepare(BlockJob& J)
{
    for (int b = 0; b < J.nblocks; ++b) {
        RunView& V = J.views[b];
        V = RunView();
        V.sym = J.slot->hsym + J.run_first[b]; V.rank = J.slot->hrank + J.run_first[b]; V.start = J.slot->hstart + J.run_first[b];
        V.count = J.run_first[b + 1] - J.run_first[b];
        V.end = (u32)(J.start[b] + J.size[b]);
        // alphabet in order of first appearance = symbols sorted by the index of their first run
        std::pair<u32, int> order[256]; int k = 0;
        for (int s = 0; s < 256; ++s) if (J.first_run[b * 256 + s] != 0xffffffffu) order[k++] = {J.first_run[b * 256 + s], s};
        std::sort(order, order + k);
        V.nsym = k;
        for (int i = 0; i < k; ++i) V.first_seen[i] = (uint8_t)order[i].second;
    }
}

// The sorted bytes of a sub-block, rebuilt from its run arrays (needed only when a sub-block has to be stored raw; this is
// why the sorted block itself never crosses PCIe).
static void expand_runs(const RunView& V, int sub_start, uint8_t* dst)
{
    for (uint32_t j = 0; j < V.count; ++j) memset(dst + (V.start[j] - (uint32_t)sub_start), V.sym[j], V.len(j));
}

// parallel framing semantics (coder.cpp:159-240): every sub-block is coded with outputSize = its own size
static void host_encode_sub(BlockJob& J, int b)
{
    const size_t need = (size_t)J.size[b] + 64;
    if (J.scratch_cap[b] < need) { J.scratch[b].reset(new uint8_t[need + need / 8]); J.scratch_cap[b] = need + need / 8; }
    if (J.use_ps) {
        if (!J.ps_landed(b)) { J.redo.store(true, std::memory_order_relaxed); J.sub_res[b] = J.size[b]; return; }
        const int r = J.ps_packed ? qlfc_encode_static_p13(J.views[b].first_seen, J.views[b].nsym, J.size[b], reinterpret_cast<const uint8_t*>(J.ps_of(b)),
                                                           (size_t)(J.poff[b + 1] - J.poff[b]), J.scratch[b].get(), J.size[b])
                    : (J.coder == LIBBSC_CODER_QLFC_FAST ? qlfc_encode_fast_pstream : qlfc_encode_static_pstream)(
                          J.views[b].first_seen, J.views[b].nsym, J.size[b], J.ps + J.poff[b], (size_t)(J.poff[b + 1] - J.poff[b]), J.scratch[b].get(), J.size[b]);
        if (r < 0) J.redo.store(true, std::memory_order_relaxed);      // would be stored raw: that needs the run arrays
        J.sub_res[b] = (r < 0) ? J.size[b] : r;
        return;
    }
    const int r = qlfc_encode_runs(J.views[b], J.size[b], J.scratch[b].get(), J.size[b], J.coder);
    J.sub_res[b] = (r < 0) ? J.size[b] : r;
}

// two sub-blocks of a device-model block in one interleaved range-coder loop (b even)
static void host_encode_pair(BlockJob& J, int b)
{
    if (!J.use_ps || b + 1 >= J.nblocks) { host_encode_sub(J, b); if (b + 1 < J.nblocks) host_encode_sub(J, b + 1); return; }
    PstreamJob P[2];
    for (int k = 0; k < 2; ++k) {
        const int q = b + k;
        const size_t need = (size_t)J.size[q] + 64;
        if (J.scratch_cap[q] < need) { J.scratch[q].reset(new uint8_t[need + need / 8]); J.scratch_cap[q] = need + need / 8; }
        P[k] = PstreamJob{J.views[q].first_seen, J.views[q].nsym, J.size[q], J.ps_of(q), (size_t)(J.poff[q + 1] - J.poff[q]), J.scratch[q].get(), J.size[q]};
    }
    if (!J.ps_landed(b + 1)) { J.redo.store(true, std::memory_order_relaxed); J.sub_res[b] = J.size[b]; J.sub_res[b + 1] = J.size[b + 1]; return; }
    int r0, r1;
    if (J.ps_packed) qlfc_encode_static_p13_pair(P[0], P[1], &r0, &r1);
    else if (J.coder == LIBBSC_CODER_QLFC_FAST) qlfc_encode_fast_pstream_pair(P[0], P[1], &r0, &r1);
    else qlfc_encode_static_pstream_pair(P[0], P[1], &r0, &r1);
    if (r0 < 0 || r1 < 0) J.redo.store(true, std::memory_order_relaxed);
    J.sub_res[b] = r0 < 0 ? J.size[b] : r0;
    J.sub_res[b + 1] = r1 < 0 ? J.size[b + 1] : r1;
}

// How the eight sub-blocks of a device-model block are coded on the host (EPYC 9575F, the coding loops alone on one quiet thread, per 64 MiB
// bench block: tools/rc_host_bench.cpp, profiles/r05/host_coder_on_box_cpu.txt; round 4's figures in brackets):
//   8  all eight in the lanes of one SIMD range-coder loop (qlfc_encode_static_pstream_x8): one task of 96 ms [104] with AVX-512VL (126 with AVX2)
//   2  four tasks of two interleaved scalar coders: 48 ms each [54], 0.19 CPU-s
//   1  eight tasks of one scalar coder: 35 ms each [46], 0.28 CPU-s [0.37]
// bscgpu_coder_task_shape is the rule (a pure function, unit-tested on CPU); ps_group feeds it.  BSC_RC_SIMD=8 / 0 forces eight lanes /
// pairs everywhere, BSC_RC_ADAPTIVE=1 turns the idle test on (round 3-4's default: pairs for a block that finds >= 4 CPUs idle; off,
// every block that is not marked low-latency is one eight-lane task).
static int ps_simd_env()
{
    static const int mode = [] {
        if (const char* e = getenv("BSC_RC_SIMD")) return atoi(e) == 8 ? 8 : 0;
        if (const char* e = getenv("BSC_RC_X8")) return atoi(e) != 0 ? 8 : 0;
        return -1;
    }();
    return mode;
}
static int default_coder_threads();
static std::atomic<int> g_sync_callers{0};            // synchronous host stages running right now (bsc_compress / bscgpu_compress_device callers)
static bool cpu_has_avx512vl() { static const bool has = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"); return has; }
// forced: -1 none, 8 / 0 (BSC_RC_SIMD); low_latency: a synchronous call, or the caller marked the block (BSCGPU_FEATURE_LOW_LATENCY);
// pool_free: CPUs of the coder pool's budget with nothing to do when the block is queued, -1 for a synchronous call (which starts
// its own threads: sync_cpus = CPUs / synchronous callers running); wide_simd: AVX-512VL; adaptive: BSC_RC_ADAPTIVE.
extern "C" BSCGPU_API int bscgpu_coder_task_shape(int forced, int low_latency, int pool_free, int sync_cpus, int wide_simd, int adaptive)
{
    if (forced >= 0) return forced == 8 ? 8 : 2;
    if (low_latency) {
        // a pipe's block marked low-latency: pairs — the blocks around it are still being coded, eight more tasks would queue behind
        // them — unless eight CPUs of the pool have nothing to do: then eight single-stream tasks all start at once and the block is out
        // after 35 instead of 48 ms (round 5: the single-stream loop lost its mispredicted branch and two cycles of its chain; before that
        // its tasks took as long as the pairs' and the threshold was 12)
        if (pool_free >= 0) return pool_free >= 8 ? 1 : 2;
        // a synchronous call: eight threads only if its share of the CPUs has room for them (the reference CLI calls bsc_compress from
        // an OpenMP team: four callers x eight threads on 16 CPUs took 1.29 s for 8 x 64 MiB, sized by share 1.15 s)
        return sync_cpus >= 8 ? 1 : 2;
    }
    if (!wide_simd) return 2;
    // A pipe's block: the eight-lane task costs half the CPU time of four pair tasks but takes 96 instead of 48 ms.  While the pool has
    // four CPUs with nothing to do the pairs cost nothing and the block is out 40 ms earlier (a short job is mostly pipeline fill and
    // drain); when the coder threads are busy — many GPUs per host, a small quota — every block is one eight-lane task and the pool's
    // throughput is what counts.
    return (adaptive && pool_free >= 4) ? 2 : 8;
}
static int ps_group(const BlockJob& J)
{
    if (!J.use_ps || J.nblocks != 8) return 2;
    // Round 5: off by default.  Pairs for blocks that find idle CPUs double the CPU time of exactly the blocks at the HEAD of a job, and
    // that work is still queued when the tail arrives: at the driver's 20 steps, one box, three alternating runs each — 3396 / 3572 / 3971
    // MB/s with the idle test, 3841 / 3862 / 4018 without; 160 steps the same (5257 / 5205) at 0.127 instead of 0.164 CPU-s per block.
    // The job's tail is still coded as short tasks: its blocks are marked low-latency by whoever knows where the job ends.
    static const int adaptive = [] { const char* e = getenv("BSC_RC_ADAPTIVE"); return e ? atoi(e) : 0; }();
    static const int cpus = default_coder_threads();
    const int callers = g_sync_callers.load(std::memory_order_relaxed);
    if ((J.features & BSCGPU_FEATURE_URGENT) && ps_simd_env() < 0) return 1;       // the caller's word: eight single-stream tasks
    // An announced job (bscgpu_coder_pool_expect): the shape follows the order in which GPU stages END, which is what decides when a
    // block's coding can start — not the order in which the caller submitted them (with several contexts interleaving on the GPU the
    // two differ by 100 ms: round 6's task timeline had blocks submitted 8th..10th from last end their GPU stage 60 ms before the
    // job's last one, take an eight-lane task of 110-120 ms each, and finish the job).  The last block: eight single-stream tasks
    // (35 ms); the few before it: pairs (50 ms); everything earlier ends in time as one eight-lane task (half the CPU time).
    if (J.tail_r >= 0 && ps_simd_env() < 0 && cpu_has_avx512vl()) {
        static const int tail_singles = [] { const char* e = getenv("BSC_TAIL_SINGLES"); return e ? atoi(e) : 1; }();
        // how many blocks before the last as pairs: two fewer than the contexts that interleave on a GPU (measured: 5 contexts: 3 > 4 > 5 pair-blocks;
        // 6 contexts: 4 = 5 > 3 — profiles/r06/pool_affinity.txt, short_job_tail.txt); BSC_TAIL_PAIRS sets it by hand
        static const int tail_pairs_env = [] { const char* e = getenv("BSC_TAIL_PAIRS"); return e ? atoi(e) : -1; }();
        const int tail_pairs = tail_pairs_env >= 0 ? tail_pairs_env : (J.tail_pipes - 2 < 2 ? 2 : J.tail_pipes - 2 > 6 ? 6 : J.tail_pipes - 2);
        const int gp = J.tail_gpus > 0 ? J.tail_gpus : 1;               // GPUs feeding the pool: that many GPU stages end per block time
        if (J.tail_r < tail_singles * gp) return 1;
        if (J.tail_r < (tail_singles + tail_pairs) * gp) return 2;
        if (!(J.features & BSCGPU_FEATURE_LOW_LATENCY)) return 8;
    }
    return bscgpu_coder_task_shape(ps_simd_env(), ((J.features & BSCGPU_FEATURE_LOW_LATENCY) || !J.pipelined) ? 1 : 0, J.pool_free,
                                   cpus / (callers > 1 ? callers : 1), cpu_has_avx512vl() ? 1 : 0, adaptive);
}
// sub-blocks b .. b + g - 1 of a device-model block, g = ps_group(J)
static void host_encode_group(BlockJob& J, int b)
{
    const int g = J.ps_g;
    if (g == 2) { host_encode_pair(J, b); return; }
    if (g == 1) { host_encode_sub(J, b); return; }
    PstreamJob P[8];
    for (int k = 0; k < g; ++k) {
        const int q = b + k;
        const size_t need = (size_t)J.size[q] + 64;
        if (J.scratch_cap[q] < need) { J.scratch[q].reset(new uint8_t[need + need / 8]); J.scratch_cap[q] = need + need / 8; }
        P[k] = PstreamJob{J.views[q].first_seen, J.views[q].nsym, J.size[q], J.ps_of(q), (size_t)(J.poff[q + 1] - J.poff[q]), J.scratch[q].get(), J.size[q]};
    }
    if (!J.ps_landed(b + g - 1)) { J.redo.store(true, std::memory_order_relaxed); for (int k = 0; k < g; ++k) J.sub_res[b + k] = J.size[b + k]; return; }
    int r[8];
    if (!(J.ps_packed ? qlfc_encode_static_p13_x8(P, r) : J.coder == LIBBSC_CODER_QLFC_FAST ? qlfc_encode_fast_pstream_x8(P, r) : qlfc_encode_static_pstream_x8(P, r))) {
        for (int k = 0; k < g; k += 2) host_encode_pair(J, b + k);      // a stream near its budget: the exact scalar coders
        return;
    }
    for (int k = 0; k < g; ++k) {
        if (r[k] < 0) J.redo.store(true, std::memory_order_relaxed);
        J.sub_res[b + k] = r[k] < 0 ? J.size[b + k] : r[k];
    }
}

// All eight sub-blocks of TWO device-model blocks in the sixteen lanes of one SIMD range-coder loop (qlfc.cpp: encode_pstream_x16): half
// the CPU time per block.  Both blocks were queued as one eight-lane task each (ps_g = 8) and use the same coder.
static void host_encode_x16(BlockJob& A, BlockJob& B)
{
    BlockJob* JJ[2] = {&A, &B};
    PstreamJob P[16];
    for (int h = 0; h < 2; ++h) {
        BlockJob& J = *JJ[h];
        for (int q = 0; q < 8; ++q) {
            const size_t need = (size_t)J.size[q] + 64;
            if (J.scratch_cap[q] < need) { J.scratch[q].reset(new uint8_t[need + need / 8]); J.scratch_cap[q] = need + need / 8; }
            P[8 * h + q] = PstreamJob{J.views[q].first_seen, J.views[q].nsym, J.size[q], J.ps_of(q), (size_t)(J.poff[q + 1] - J.poff[q]), J.scratch[q].get(), J.size[q]};
        }
    }
    bool landed = true;
    for (int h = 0; h < 2; ++h) if (!JJ[h]->ps_landed(7)) landed = false;
    int r[16];
    if (!landed || !(A.coder == LIBBSC_CODER_QLFC_FAST ? qlfc_encode_fast_pstream_x16(P, r) : qlfc_encode_static_pstream_x16(P, r))) {
        host_encode_group(A, 0); host_encode_group(B, 0);            // a copy that failed, or a stream near its budget: each block on its own (eight lanes, then the exact scalar coders)
        return;
    }
    for (int h = 0; h < 2; ++h) for (int q = 0; q < 8; ++q) {
        BlockJob& J = *JJ[h];
        if (r[8 * h + q] < 0) J.redo.store(true, std::memory_order_relaxed);
        J.sub_res[q] = r[8 * h + q] < 0 ? J.size[q] : r[8 * h + q];
    }
}

static void write_stored(BlockJob& J)
{
    uint8_t* output = J.output; const int n = J.n_orig;
    if (J.hInput) { J.result = J.inplace ? LIBBSC_NOT_COMPRESSIBLE : bsc_store(J.hInput, output, n, J.features); return; }   // libbsc.cpp:188, :315
    if (hipSetDevice(J.c->device) != hipSuccess ||
        hipMemcpy(output + LIBBSC_HEADER_SIZE, J.dInput, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { J.result = LIBBSC_GPU_ERROR; return; }
    put_i32(output + 0, n + LIBBSC_HEADER_SIZE); put_i32(output + 4, n); put_i32(output + 8, 0); put_i32(output + 12, 0);
    put_i32(output + 16, (int)J.adler_data); put_i32(output + 20, (int)J.adler_data);
    put_i32(output + 24, (int)adler32(output, 24));
    J.result = n + LIBBSC_HEADER_SIZE;
}

static void write_header_and_trailer(BlockJob& J, int result)
{
    uint8_t* output = J.output; const int n = J.n_orig;
    if (result < LIBBSC_NO_ERROR || result + 1 + 4 * J.num_indexes >= n) { write_stored(J); return; }      // libbsc.cpp:315-318
    if (J.num_indexes > 0) memcpy(output + LIBBSC_HEADER_SIZE + result, J.indexes, (size_t)4 * J.num_indexes);
    output[LIBBSC_HEADER_SIZE + result + 4 * J.num_indexes] = (unsigned char)J.num_indexes;
    result += 1 + 4 * J.num_indexes;
    put_i32(output + 0, result + LIBBSC_HEADER_SIZE);
    put_i32(output + 4, n);
    put_i32(output + 8, J.mode);
    put_i32(output + 12, J.index);
    put_i32(output + 16, (int)J.adler_data);
    put_i32(output + 20, (int)adler32(output + LIBBSC_HEADER_SIZE, (size_t)result));
    put_i32(output + 24, (int)adler32(output, 24));
    J.result = result + LIBBSC_HEADER_SIZE;
}

// frame the independently coded sub-blocks (nblocks > 1, MULTITHREADING semantics)
static void host_finalize_parallel(BlockJob& J)
{
    uint8_t* out = J.output + LIBBSC_HEADER_SIZE; const int n = J.n, nb = J.nblocks;
    int total = 1 + 8 * nb;
    for (int b = 0; b < nb; ++b) total += J.sub_res[b];
    if (total >= n) { write_header_and_trailer(J, LIBBSC_NOT_COMPRESSIBLE); }
    else {
        out[0] = (uint8_t)nb;
        int optr = 1 + 8 * nb;
        for (int b = 0; b < nb; ++b) {
            put_i32(out + 1 + 8 * b, J.size[b]);
            put_i32(out + 1 + 8 * b + 4, J.sub_res[b]);
            if (J.sub_res[b] != J.size[b]) memcpy(out + optr, J.scratch[b].get(), (size_t)J.sub_res[b]);
            else expand_runs(J.views[b], J.start[b], out + optr);
            optr += J.sub_res[b];
        }
        write_header_and_trailer(J, total);
    }
}

// every block with more than one sub-block is coded as independent per-sub-block tasks, whatever the caller's
// MULTITHREADING flag says: the flag selects the reference's *framing rules* (coder.cpp:111 serial / :159 parallel), which
// differ only in corner cases, not whether this library may use its coder threads.
static bool job_uses_tasks(const BlockJob& J) { return !J.stored_small && J.nblocks > 1; }

// the block's sub-blocks through the strictly serial coder (coder.cpp:111-155) on this thread
static void host_code_serially(BlockJob& J)
{
    struct Fetch : RawFetch {
        const BlockJob* J;
        int operator()(int st, int sz, uint8_t* dst) override
        {
            for (int b = 0; b < J->nblocks; ++b) if (J->start[b] == st && J->size[b] == sz) { expand_runs(J->views[b], st, dst); return 0; }
            return LIBBSC_BAD_PARAMETER;
        }
    } fetch; fetch.J = &J;
    unsigned char* buffer = (unsigned char*)bsc_malloc((size_t)J.n + 4096);
    if (!buffer) { J.result = LIBBSC_NOT_ENOUGH_MEMORY; return; }
    const int result = coder_compress_views(J.views, J.nblocks, J.start, J.size, J.n, buffer, J.coder, J.features & ~LIBBSC_FEATURE_MULTITHREADING, fetch);
    if (result >= 0) memcpy(J.output + LIBBSC_HEADER_SIZE, buffer, (size_t)result);
    bsc_free(buffer);
    write_header_and_trailer(J, result);
}

// Serial framing rules (coder.cpp:111-155) applied to sub-blocks that were coded independently, each with an output
// budget of its own size.  The serial coder gives sub-block b the budget min(size_b, n - bytes written so far); when that
// is size_b for every b — always, unless the block is close to incompressible — its result is exactly what the
// independent run produced.  Otherwise the block is simply coded again serially.
static void host_finalize_serial(BlockJob& J)
{
    uint8_t* out = J.output + LIBBSC_HEADER_SIZE; const int n = J.n, nb = J.nblocks;
    int optr = 1 + 8 * nb;
    bool exact = true, incompressible = false;
    for (int b = 0; b < nb && exact && !incompressible; ++b) {
        if (n - optr < J.size[b]) { exact = false; break; }
        if (J.sub_res[b] == J.size[b] && optr + J.size[b] >= n) { incompressible = true; break; }       // coder.cpp:131-134
        optr += J.sub_res[b];
    }
    if (!exact) { if (J.use_ps) { J.redo.store(true, std::memory_order_relaxed); return; } host_code_serially(J); return; }
    if (incompressible) write_header_and_trailer(J, LIBBSC_NOT_COMPRESSIBLE);
    else {
        out[0] = (uint8_t)nb;
        optr = 1 + 8 * nb;
        for (int b = 0; b < nb; ++b) {
            put_i32(out + 1 + 8 * b, J.size[b]);
            put_i32(out + 1 + 8 * b + 4, J.sub_res[b]);
            if (J.sub_res[b] != J.size[b]) memcpy(out + optr, J.scratch[b].get(), (size_t)J.sub_res[b]);
            else expand_runs(J.views[b], J.start[b], out + optr);
            optr += J.sub_res[b];
        }
        write_header_and_trailer(J, optr);
    }
}

static void host_finalize(BlockJob& J)
{
    if (J.use_ps && J.redo.load(std::memory_order_relaxed)) { J.result = LIBBSC_GPU_ERROR; return; }   // finished later by redo_on_host_model(), which sets the result; an error until then
    if (J.features & LIBBSC_FEATURE_MULTITHREADING) host_finalize_parallel(J); else host_finalize_serial(J);
}

// whole host stage on the calling thread (+ its own sub-block threads): the synchronous entry points
static void host_stage(BlockJob& J)
{
    if (J.stored_small) return;
    host_prepare(J);
    if (job_uses_tasks(J)) {
        struct Caller { Caller() { g_sync_callers.fetch_add(1, std::memory_order_relaxed); } ~Caller() { g_sync_callers.fetch_sub(1, std::memory_order_relaxed); } } caller;
        if (J.use_ps) { const int g = J.ps_g = ps_group(J); run_tasks(J.nblocks / g, [&J, g](int t) { host_encode_group(J, g * t); }); }
        else run_tasks(J.nblocks, [&J](int b) { host_encode_sub(J, b); });
        host_finalize(J);
        return;
    }
    host_code_serially(J);
}

// Host-resident block -> job, part 1 (no GPU involved): optional LZP on the host (lzp.cpp:798 semantics).  `sorter` may be
// changed (a tiny LZP output is always BWT-sorted, libbsc.cpp:277-281).
static int stage_host_lzp(BlockJob& J, const unsigned char* input, unsigned char* output, int n,
                          int lzpHashSize, int lzpMinLen, int* sorter, int coder, int features, int mode)
{
    int lzSize = n;
    J.lz.reset();
    if (mode != (mode & 0xff)) {
        J.lz = std::unique_ptr<unsigned char, void (*)(void*)>((unsigned char*)bigbuf_get((size_t)n), bigbuf_put);   // (par.h: a few block-sized buffers are kept)
        if (!J.lz) return LIBBSC_NOT_ENOUGH_MEMORY;
        const int r = lzp_compress(input, J.lz.get(), n, lzpHashSize, lzpMinLen, features);
        if (r < LIBBSC_NO_ERROR) { mode &= 0xff; J.lz.reset(); }            // libbsc.cpp:266-269: the block goes on without LZP
        else lzSize = r;
    }
    if (lzSize <= LIBBSC_HEADER_SIZE) { *sorter = LIBBSC_BLOCKSORTER_BWT; mode = (mode & ~0x1f) | LIBBSC_BLOCKSORTER_BWT; }
    J.hInput = input; J.output = output; J.n = lzSize; J.n_orig = n; J.inplace = (input == output);
    J.coder = coder; J.features = features; J.mode = mode; J.stored_small = false; J.result = 0; J.have_adler = false;
    if (J.lz) { J.adler_data = adler32(input, (size_t)n); J.have_adler = true; }
    return LIBBSC_NO_ERROR;
}
// part 2: one H2D copy of what the sorter will see into the context's text buffer
static int stage_host_h2d(BlockJob& J, bscgpu_ctx* c)
{
    J.c = c; J.dInput = c->dL;
    if (hipSetDevice(c->device) != hipSuccess) return LIBBSC_GPU_ERROR;
    const unsigned char* data = J.lz ? J.lz.get() : J.hInput;
    if (hipMemcpyAsync(c->dL, data, (size_t)J.n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return LIBBSC_GPU_ERROR;
    return LIBBSC_NO_ERROR;
}

extern "C" BSCGPU_API long long bscgpu_process_counter(int key)
{
    switch (key) {
        case BSCGPU_PCNT_DEVICE_MODEL_BLOCKS:     return (long long)g_count_devmodel.load(std::memory_order_relaxed);
        case BSCGPU_PCNT_REDONE_ON_HOST_MODEL:    return (long long)g_count_redo.load(std::memory_order_relaxed);
        case BSCGPU_PCNT_DEVICE_MODEL_LZP_BLOCKS: return (long long)g_count_devmodel_lzp.load(std::memory_order_relaxed);
    }
    return LIBBSC_BAD_PARAMETER;
}

// A block that took the device model but has a sub-block that does not compress (or needs the strictly serial framing) is
// run again with the model on the host: raw sub-blocks are rebuilt from the run arrays, which that path never copied.  Rare
// (such blocks are mostly caught before by their run count); the caller must be the thread that owns the context's GPU stage.
// For an LZP-preprocessed block the sorter's input is the LZP output, which is why a device-model block keeps it until here.
static int redo_on_host_model(BlockJob& J)
{
    g_count_redo.fetch_add(1, std::memory_order_relaxed);
    J.redo.store(false, std::memory_order_relaxed);
    int rc = LIBBSC_NO_ERROR;
    if (J.hInput) rc = stage_host_h2d(J, J.c);
    if (rc >= 0) rc = gpu_stage(J, J.sorter, false);
    J.lz.reset();
    if (rc < 0) { J.result = rc; return rc; }
    host_stage(J);
    return J.result;
}

// bsc_compress (libbsc.cpp:213-338; input == output selects the in-place rules, :83-211): optional LZP on the host, then
// the same GPU stage + host coder as the device-resident entry point.
int bsc_compress(const unsigned char* input, unsigned char* output, int n, int lzpHashSize, int lzpMinLen,
                 int blockSorter, int coder, int features)
{
    const bool inplace = (input == output);
    int mode = 0;
    int rc = make_mode(blockSorter, coder, lzpHashSize, lzpMinLen, &mode);
    if (rc != LIBBSC_NO_ERROR) return rc;
    if (!input || !output) return LIBBSC_BAD_PARAMETER;
    if (n < 0 || n > (inplace ? 2146435072 : 1073741824)) return LIBBSC_BAD_PARAMETER;
    // The reference's in-place twin takes blocks up to 2047 MiB (libbsc.cpp:124).  The device sorter is sized and tested up to the
    // format's regular maximum, 1 GiB (tests/test_gpu_compress.py: test_max_block_1gib_golden); above it this library says so
    // instead of running an arena layout nobody has ever exercised.
    if (n > 1073741824) return LIBBSC_NOT_SUPPORTED;
    if (n <= LIBBSC_HEADER_SIZE) return bsc_store(input, output, n, features);

    std::unique_ptr<BlockJob> J(new BlockJob);
    rc = stage_host_lzp(*J, input, output, n, lzpHashSize, lzpMinLen, &blockSorter, coder, features, mode);
    if (rc < 0) return rc;
    DefaultGpuUser user(J->n, true);
    if (user.rc != LIBBSC_NO_ERROR) return user.rc;
    bscgpu_ctx* c = user.c;
    {
        std::lock_guard<std::mutex> g(user.dev->gpu_lock);      // GPU stage; concurrent callers on this device queue here
        if (hipSetDevice(c->device) != hipSuccess) return LIBBSC_GPU_ERROR;
        rc = ctx_ensure_slots(c, user.slot + 1);
        if (rc < 0) return rc;
        J->slot = &c->slots[user.slot];
        rc = stage_host_h2d(*J, c);
        if (rc < 0) return rc;
        rc = gpu_stage(*J, blockSorter);
        if (!J->use_ps) J->lz.reset();                          // (a device-model block may come back for a redo: redo_on_host_model)
        if (rc < 0) return rc;
    }
    host_stage(*J);                                             // host coder: overlaps the next caller's GPU stage
    if (J->redo.load(std::memory_order_relaxed)) {
        std::lock_guard<std::mutex> g(user.dev->gpu_lock);
        if (hipSetDevice(c->device) != hipSuccess) return LIBBSC_GPU_ERROR;
        return redo_on_host_model(*J);
    }
    return J->result;
}

int bsc_lzp_compress(const unsigned char* input, unsigned char* output, int n, int hashSize, int minLen, int features)
{
    if (!input || !output || n < 0 || minLen < 4 || minLen > 255 || hashSize < 10 || hashSize > 28) return LIBBSC_BAD_PARAMETER;
    return lzp_compress(input, output, n, hashSize, minLen, features);
}
int bsc_lzp_decompress(const unsigned char* input, unsigned char* output, int n, int hashSize, int minLen, int features)
{
    (void)features;
    if (!input || !output || n < 0 || minLen < 4 || minLen > 255 || hashSize < 10 || hashSize > 28) return LIBBSC_BAD_PARAMETER;
    return lzp_decompress(input, output, n, 0x7fffffff, hashSize, minLen);
}

static int prepare_job(BlockJob& J, bscgpu_ctx* c, const void* dInput, uint8_t* output, int n, int blockSorter, int coder, int features)
{
    if (!c || !dInput || !output) return LIBBSC_BAD_PARAMETER;
    int rc = make_mode(blockSorter, coder, 0, 0, &J.mode);
    if (rc != LIBBSC_NO_ERROR) return rc;
    if (n < 0 || n > c->max_n) return LIBBSC_BAD_PARAMETER;
    J.c = c; J.dInput = dInput; J.output = output; J.n = n; J.n_orig = n; J.coder = coder; J.features = features;
    J.hInput = nullptr; J.have_adler = false; J.inplace = false; J.lz.reset();
    J.stored_small = false; J.result = 0;
    return LIBBSC_NO_ERROR;
}

int bscgpu_compress_device(bscgpu_ctx* c, const void* dInput, uint8_t* output, int n, int blockSorter, int coder, int features)
{
    const auto t_all = clk::now();
    std::unique_ptr<BlockJob> J(new BlockJob);
    int rc = prepare_job(*J, c, dInput, output, n, blockSorter, coder, features);
    if (rc < 0) return rc;
    J->slot = &c->slots[0];
    rc = gpu_stage(*J, blockSorter);
    if (rc < 0) return rc;
    const auto t0 = clk::now();
    host_stage(*J);
    if (J->redo.load(std::memory_order_relaxed)) redo_on_host_model(*J);
    c->stage_ms[3] = ms_since(t0);
    c->stage_ms[4] = ms_since(t_all);
    return J->result;
}

// CPUs this process can actually use: min(affinity mask, cgroup v2 cpu.max quota), clamped to [4, 64].  The coder pool of a
// pipe defaults to this (a multi-rank driver divides it between its ranks through BSCGPU_HOST_THREADS).
// CPUs of CPU time the cgroup grants (cgroup v2 cpu.max), 0 if unlimited or unknown
static int cgroup_quota_cpus()
{
    int c = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) c = (int)(quota / period);
        }
        fclose(f);
    }
    return c;
}
static int default_coder_threads()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = c; }
    { const int c = cgroup_quota_cpus(); if (c >= 1 && c < n) n = c; }
    if (n < 4) n = 4;
    if (n > 64) n = 64;
    return n;
}

// Where the pool's threads may run (round 6, profiles/r06/pool_affinity.txt).  The 1-GPU boxes grant 16 CPUs of CPU TIME on a machine of
// 2 x 64 cores x 2 hardware threads, and leave the process free to run anywhere: coder threads then share cores with their SMT siblings
// and read a landing zone on the other socket whenever the scheduler says so.  Kept to one hardware thread per core of the GPU's own NUMA
// node a 20-block job runs 4472 MB/s against 4228 left alone (means of five interleaved runs, spread 4286-4663 against 3980-4553);
// long jobs are level.  Rule: of the CPUs the process may use, the first hardware thread of every core — on the GPU's node when the
// process sees ONE GPU (several GPUs: all nodes) — provided the machine is NOT meant to be filled: the cgroup grants a quota of CPU time
// and the whole quota fits on the chosen CPUs (every process of the cgroup can then follow the same rule: eight ranks of a node share
// one quota).  Without a quota the hardware threads are all the process's to use, and nothing is restricted.
// BSCGPU_HOST_AFFINITY=0 turns it off, =<cpu list> (e.g. 0-31,64-95) sets it by hand.
static bool parse_cpu_list(const char* t, cpu_set_t* out)
{
    CPU_ZERO(out);
    int n = 0;
    while (*t) {
        char* end = nullptr;
        const long a = strtol(t, &end, 10);
        if (end == t || a < 0 || a >= CPU_SETSIZE) return false;
        long b = a;
        if (*end == '-') { t = end + 1; b = strtol(t, &end, 10); if (end == t || b < a || b >= CPU_SETSIZE) return false; }
        for (long c = a; c <= b; ++c) { CPU_SET((int)c, out); ++n; }
        t = (*end == ',') ? end + 1 : end;
        if (*end != ',' && *end != 0 && *end != '\n') return false;
        if (*end == '\n') break;
    }
    return n > 0;
}
static bool read_cpu_list_file(const char* path, cpu_set_t* out)
{
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096]; const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    return got && parse_cpu_list(buf, out);
}
static bool pool_cpu_set(int budget, int device, cpu_set_t* out)
{
    const char* e = getenv("BSCGPU_HOST_AFFINITY");
    if (e && e[0] == '0' && e[1] == 0) return false;
    if (e && e[0] && !(e[0] == '1' && e[1] == 0)) return parse_cpu_list(e, out);
    const int quota = cgroup_quota_cpus();
    if (quota <= 0) return false;                                       // no CPU-time limit: every hardware thread is there to be used
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    // the GPU's NUMA node, if the process sees exactly one GPU
    cpu_set_t node; bool have_node = false;
    int ndev = 0;
    if (device >= 0 && hipGetDeviceCount(&ndev) == hipSuccess && ndev == 1) {
        char bus[64];
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) == hipSuccess) {
            for (char* q = bus; *q; ++q) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');
            char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
            int nd = -1;
            if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &nd) != 1) nd = -1; fclose(f); }
            if (nd >= 0) { snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd); have_node = read_cpu_list_file(path, &node); }
        }
    } else (void)hipGetLastError();
    for (int pass = have_node ? 0 : 1; pass < 2; ++pass) {              // pass 0: the GPU's node only; pass 1: every node
        CPU_ZERO(out);
        int n = 0;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &allowed) || (pass == 0 && !CPU_ISSET(c, &node))) continue;
            char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            cpu_set_t sib;
            if (!read_cpu_list_file(path, &sib)) return false;           // no topology: leave the threads alone
            int first = -1;
            for (int k = 0; k < CPU_SETSIZE; ++k) if (CPU_ISSET(k, &sib) && CPU_ISSET(k, &allowed)) { first = k; break; }
            if (first == c) { CPU_SET(c, out); ++n; }
        }
        if (n >= budget && n >= quota && n < CPU_COUNT(&allowed)) return true;        // (n == allowed: nothing to restrict)
    }
    return false;
}

// Optional task trace (BSCGPU_POOL_TRACE=1; bench.py BSC_BENCH_TRACE prints it): when each coder task ran — the only way to see where a
// short job's drain goes, since a pipe's caller learns of a finished block only when it next asks.
struct PoolTraceRec { double t0, t1; const void* job; int sub, shape, features; };
static std::mutex g_trace_mu;
static std::vector<PoolTraceRec> g_trace;
static const bool g_trace_on = [] { const char* e = getenv("BSCGPU_POOL_TRACE"); return e && e[0] == '1'; }();
static double trace_now() { return std::chrono::duration<double>(clk::now().time_since_epoch()).count(); }
extern "C" BSCGPU_API int bscgpu_coder_pool_trace(double* out /* [cap][6]: start, end (seconds, steady clock), job id, sub-block, sub-blocks per task, features */, int cap, int reset)
{
    std::lock_guard<std::mutex> g(g_trace_mu);
    int n = 0;
    for (const PoolTraceRec& r : g_trace) {
        if (n >= cap) break;
        out[6 * n + 0] = r.t0; out[6 * n + 1] = r.t1; out[6 * n + 2] = (double)(uintptr_t)r.job; out[6 * n + 3] = r.sub; out[6 * n + 4] = r.shape; out[6 * n + 5] = r.features;
        ++n;
    }
    if (reset) g_trace.clear();
    return n;
}
extern "C" BSCGPU_API double bscgpu_steady_now() { return trace_now(); }

// ---- pipe: several blocks in flight -------------------------------------------------------------------------
// submit() runs the GPU stage on the calling thread and queues the block's host work as tasks for the coder pool: ONE pool per
// process (the CPUs it may use: affinity and cgroup quota, default_coder_threads(); BSCGPU_HOST_THREADS overrides the number of
// threads, BSCGPU_HOST_CPUS the CPU budget the idle test below counts against — a multi-rank driver gives each rank its share),
// shared by every pipe — several contexts per GPU, as bench.py runs them, draw on the same threads, so a block whose own pipe
// is momentarily quiet is coded by whoever is free, and the pool knows how busy the process's CPUs are (ps_group).  FIFO; the
// worker that finishes a block's last task frames it.
static std::atomic<uint64_t> g_pool_x16{0};           // blocks coded two at a time in sixteen lanes
struct CoderPool {
    struct Task { BlockJob* job; int sub; };          // sub = -1: whole host stage of the block as one task; else first sub-block of the task
    std::mutex mu; std::condition_variable cv_work, cv_done;
    std::deque<Task> queue;
    std::vector<std::thread> workers;
    int active = 0, budget = 0, users = 0;
    // Pairing of eight-lane blocks into sixteen-lane tasks (host_encode_x16): a worker that draws an eight-lane block and finds another
    // one HELD takes both; else it holds its own for up to x16_wait_ms (asleep: no CPU) while more blocks are to come, and codes it alone
    // when nobody came.  At most one block is held at a time.
    Task held{nullptr, 0};
    std::condition_variable cv_hold;
    bool x16 = false; int x16_wait_ms = 15;
    long long to_come = -1;                           // blocks of the announced job not yet queued (bscgpu_coder_pool_expect); -1: none announced
    int to_come_gpus = 1;
    bool stop = false;

    void worker_loop()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;
                t = queue.front(); queue.pop_front();
                ++active;
                // an eight-lane block: with a partner it is half the CPU time
                BlockJob* partner = nullptr;
                if (x16 && t.sub == 0 && t.job->use_ps && !t.job->ps_packed && t.job->ps_g == 8 && t.job->nblocks == 8) {     // (the sixteen-lane coder reads 16-bit entries: BSC_PS13=0)
                    if (held.job && held.job->coder == t.job->coder) { partner = held.job; held.job = nullptr; cv_hold.notify_all(); }
                    else if (!held.job && (t.job->tail_r < 0 || t.job->tail_r >= 8 * t.job->tail_gpus)) {
                        // (a block of the job's tail is not held: its eight-lane task has to start now to end in time)
                        held = t; --active;
                        cv_hold.wait_for(lk, std::chrono::milliseconds(x16_wait_ms), [&] { return stop || held.job != t.job; });
                        if (held.job == t.job) { held.job = nullptr; ++active; }          // nobody came: alone after all
                        else continue;                                                     // taken: the worker that took it reports both blocks
                    }
                }
                if (partner) {
                    lk.unlock();
                    BlockJob& A = *partner; BlockJob& B = *t.job;
                    const double tr0 = g_trace_on ? trace_now() : 0.0;
                    host_encode_x16(A, B);
                    bool fa = false, fb = false;
                    if (A.remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) { host_finalize(A); fa = true; }
                    if (B.remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) { host_finalize(B); fb = true; }
                    g_pool_x16.fetch_add(2, std::memory_order_relaxed);
                    if (g_trace_on) { const double tr1 = trace_now(); std::lock_guard<std::mutex> g(g_trace_mu); g_trace.push_back({tr0, tr1, (const void*)&A, 0, 16, A.features}); g_trace.push_back({tr0, tr1, (const void*)&B, 0, 16, B.features}); }
                    { std::lock_guard<std::mutex> lk2(mu); --active; if (fa) A.done = true; if (fb) B.done = true; }
                    cv_done.notify_all();
                    continue;
                }
            }
            BlockJob& J = *t.job;
            bool finished = false;
            const double tr0 = g_trace_on ? trace_now() : 0.0;
            const int tr_shape = J.use_ps ? J.ps_g : 0, tr_feat = J.features;
            if (t.sub == -1) { host_stage(J); finished = true; }
            else {
                if (J.use_ps) host_encode_group(J, t.sub); else host_encode_sub(J, t.sub);
                if (J.remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) { host_finalize(J); finished = true; }
            }
            if (g_trace_on) { const double tr1 = trace_now(); std::lock_guard<std::mutex> g(g_trace_mu); g_trace.push_back({tr0, tr1, (const void*)&J, t.sub, tr_shape, tr_feat}); }
            { std::lock_guard<std::mutex> lk(mu); --active; if (finished) J.done = true; }
            if (finished) cv_done.notify_all();
        }
    }
    // CPUs of the budget that have neither a task nor one waiting for them (caller holds mu)
    int free_cpus() const { const int f = budget - active - (int)queue.size(); return f < 0 ? 0 : f; }
};
static std::mutex g_pool_mu;
static CoderPool* g_pool = nullptr;

static std::atomic<uint64_t> g_pool_mode[4];          // blocks queued as: 8 scalar tasks, 4 pair tasks, one eight-lane task, host-model tasks

// The caller's knowledge of where a job ENDS: `blocks` more blocks will be submitted to this process's pipes (any of them); the pool then
// shapes the host tasks of the blocks whose GPU stages end last for latency (ps_group).  blocks < 0: no announcement (the default).
static long long g_pending_expect = -1; static int g_pending_gpus = 1;      // an announcement made before the first pipe exists (under g_pool_mu)
extern "C" BSCGPU_API int bscgpu_coder_pool_expect(long long blocks, int gpus)
{
    if (gpus < 1) gpus = 1;
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (!g_pool) { g_pending_expect = blocks < 0 ? -1 : blocks; g_pending_gpus = gpus; return LIBBSC_NO_ERROR; }
    std::lock_guard<std::mutex> lk(g_pool->mu);
    g_pool->to_come = blocks < 0 ? -1 : blocks;
    g_pool->to_come_gpus = gpus;
    return LIBBSC_NO_ERROR;
}

extern "C" BSCGPU_API uint64_t bscgpu_coder_pool_x16_blocks(int reset)
{
    const uint64_t v = g_pool_x16.load(std::memory_order_relaxed);
    if (reset) g_pool_x16.store(0, std::memory_order_relaxed);
    return v;
}

void bscgpu_coder_pool_stats(uint64_t out[4], int reset)
{
    for (int i = 0; i < 4; ++i) { out[i] = g_pool_mode[i].load(std::memory_order_relaxed); if (reset) g_pool_mode[i].store(0, std::memory_order_relaxed); }
}

static CoderPool* pool_acquire(int device = -1)
{
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (!g_pool) {
        CtxTimer tm("coder pool threads");
        CoderPool* P = new CoderPool;
        // A quarter more threads than CPUs (round 5: the library's default; bench.py had been asking for half as many again since round
        // 3): a task spends part of its life asleep, waiting for its sub-blocks' copy from the GPU, so some surplus pays — but where the
        // CPUs are a cgroup QUOTA (CPU time, not cores: the 1-GPU boxes grant 16 of 256 hardware threads), 24 runnable threads at the tail of
        // a job overdraw it and the whole process, GPU-driving threads included, is stopped for the rest of the 100 ms period
        // (cpu.stat: one throttled period per 20-step run with 24 threads, none with 20; 3802 / 3844 MB/s against 4015 / 3941, 16
        // threads 3693 / 3701, one box, alternating runs, profiles/r05/coder_threads_and_quota.txt).  A cap on the tasks CODING at the same
        // time (budget - 1 slots, taken after a task's input has landed) was built and measured as well: no gain at 20 or 28 threads
        // (3565 against 3690 MB/s, means of three alternating runs, profiles/r05/coding_slots_and_tail_marks.txt) — removed.
        const int cpus = default_coder_threads();
        int nworkers = cpus + cpus / 4; if (nworkers > 96) nworkers = 96;
        bool forced = false;
        if (const char* e = getenv("BSCGPU_HOST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 256) { nworkers = v; forced = true; } }
        P->budget = (forced && nworkers < cpus) ? nworkers : cpus;
        if (const char* e = getenv("BSCGPU_HOST_CPUS")) { int v = atoi(e); if (v >= 1 && v <= 256) P->budget = v; }
        for (int i = 0; i < nworkers; ++i) P->workers.emplace_back([P] { P->worker_loop(); });
        cpu_set_t where;
        if (pool_cpu_set(P->budget, device, &where))
            for (auto& t : P->workers) (void)pthread_setaffinity_np(t.native_handle(), sizeof where, &where);
        P->to_come = g_pending_expect; P->to_come_gpus = g_pending_gpus; g_pending_expect = -1;
        P->x16 = qlfc_x16_available() && ps_simd_env() < 0;
        if (const char* e = getenv("BSC_RC_X16_WAIT_MS")) { const int v = atoi(e); if (v >= 0 && v <= 1000) P->x16_wait_ms = v; }
        g_pool = P;
    }
    ++g_pool->users;
    return g_pool;
}
static void pool_release(CoderPool* P)                 // the last pipe takes the threads with it
{
    std::unique_lock<std::mutex> g(g_pool_mu);
    if (--P->users > 0) return;
    g_pool = nullptr;
    g.unlock();
    { std::lock_guard<std::mutex> lk(P->mu); P->stop = true; }
    P->cv_work.notify_all();
    for (auto& t : P->workers) t.join();
    delete P;
}

struct bscgpu_pipe {
    bscgpu_ctx* c = nullptr;
    int depth = 1;
    int next_ticket = 0;
    struct Lane { std::unique_ptr<BlockJob> job; int ticket = -1; bool busy = false; bool joined = true; };   // ticket / joined / job->done: under the pool's mutex
    Lane lanes[MAX_SLOTS];
    CoderPool* pool = nullptr;
};

static void lane_join(bscgpu_pipe* p, bscgpu_pipe::Lane& L)
{
    if (!L.busy) return;
    {
        std::unique_lock<std::mutex> lk(p->pool->mu);
        p->pool->cv_done.wait(lk, [&] { return L.job->done; });
        L.joined = true;                            // (bscgpu_pipe_peek: from here on the lane may be reused; a peeker goes back to its caller's own bookkeeping)
    }
    p->pool->cv_done.notify_all();
    L.busy = false;
    // lane_join only runs on the pipe's submitting thread (submit / wait / destroy), which owns the GPU stage
    // (a redo that cannot be run must not leave the stale result of a block whose output was never written)
    if (L.job->redo.load(std::memory_order_relaxed)) {
        if (hipSetDevice(p->c->device) == hipSuccess) redo_on_host_model(*L.job);
        else L.job->result = LIBBSC_GPU_ERROR;
    }
    L.job->lz.reset();                              // a device-model block kept its LZP output for that redo only: a lane must not sit on a block-sized buffer until its next block
}

int bscgpu_pipe_create(bscgpu_ctx* c, int depth, bscgpu_pipe** out)
{
    if (!c || !out || depth < 1 || depth > MAX_SLOTS) return LIBBSC_BAD_PARAMETER;
    if (hipSetDevice(c->device) != hipSuccess) return LIBBSC_GPU_ERROR;
    int rc = ctx_ensure_slots(c, depth);
    if (rc < 0) return rc;
    bscgpu_pipe* p = new bscgpu_pipe;
    p->c = c; p->depth = depth;
    for (int i = 0; i < depth; ++i) p->lanes[i].job.reset(new BlockJob);
    p->pool = pool_acquire(c->device);
    *out = p;
    return LIBBSC_NO_ERROR;
}

void bscgpu_pipe_destroy(bscgpu_pipe* p)
{
    if (!p) return;
    for (int i = 0; i < p->depth; ++i) lane_join(p, p->lanes[i]);
    pool_release(p->pool);
    delete p;
}

// queue the host half of a block whose GPU stage has run
static int pipe_enqueue(bscgpu_pipe* p, bscgpu_pipe::Lane& L, int ticket)
{
    BlockJob& J = *L.job;
    L.busy = true;
    J.pipelined = p->depth >= 3;
    CoderPool* P = p->pool;
    J.pipe_workers = (int)P->workers.size();
    {
        std::lock_guard<std::mutex> lk(P->mu);
        L.ticket = ticket; L.joined = false;
        J.tail_r = -1;
        J.tail_gpus = P->to_come_gpus;
        J.tail_pipes = P->users / (P->to_come_gpus > 0 ? P->to_come_gpus : 1);
        if (P->to_come > 0) J.tail_r = (int)(--P->to_come < 0x7fffffff ? P->to_come : 0x7fffffff);
        else if (P->to_come == 0) P->to_come = -1;               // more blocks than announced: the rule is dropped
        J.done = false;                                          // (with the other fields a peeker reads, under the pool's mutex)
        p->next_ticket = ticket + 1;
        J.pool_free = P->free_cpus();
        if (job_uses_tasks(J)) {
            host_prepare(J);
            if (J.use_ps) {                                      // device model: 1 (scalar), 2 (interleaved scalar) or 8 (SIMD lanes) sub-blocks per task
                const int g = J.ps_g = ps_group(J);
                g_pool_mode[g == 1 ? 0 : g == 2 ? 1 : 2].fetch_add(1, std::memory_order_relaxed);
                J.remaining.store(J.nblocks / g, std::memory_order_release);
                for (int b = 0; b < J.nblocks; b += g) P->queue.push_back({&J, b});
            } else {
                g_pool_mode[3].fetch_add(1, std::memory_order_relaxed);
                J.remaining.store(J.nblocks, std::memory_order_release);
                for (int b = 0; b < J.nblocks; ++b) P->queue.push_back({&J, b});
            }
        } else {
            g_pool_mode[3].fetch_add(1, std::memory_order_relaxed);
            P->queue.push_back({&J, -1});
        }
    }
    P->cv_work.notify_all();
    return ticket;
}

int bscgpu_pipe_submit(bscgpu_pipe* p, const void* dInput, uint8_t* output, int n, int blockSorter, int coder, int features)
{
    if (!p) return LIBBSC_BAD_PARAMETER;
    const int ticket = p->next_ticket;
    bscgpu_pipe::Lane& L = p->lanes[ticket % p->depth];
    lane_join(p, L);                                // the slot's previous block must be finished before its buffers are reused
    BlockJob& J = *L.job;
    int rc = prepare_job(J, p->c, dInput, output, n, blockSorter, coder, features);
    if (rc < 0) return rc;
    J.slot = &p->c->slots[ticket % p->depth];
    rc = gpu_stage(J, blockSorter);
    J.lz.reset();                                   // (device-resident input: there is no LZP output)
    if (rc < 0) return rc;
    return pipe_enqueue(p, L, ticket);
}

// Host-resident block with the full bsc_compress parameter set (LZP included); `input` must stay valid until the
// ticket has been waited for (a block that does not compress is stored from it).
int bscgpu_pipe_submit_host(bscgpu_pipe* p, const uint8_t* input, uint8_t* output, int n, int lzpHashSize, int lzpMinLen,
                            int blockSorter, int coder, int features)
{
    if (!p || !input || !output) return LIBBSC_BAD_PARAMETER;
    int mode = 0;
    int rc = make_mode(blockSorter, coder, lzpHashSize, lzpMinLen, &mode);
    if (rc != LIBBSC_NO_ERROR) return rc;
    if (n < 0 || n > p->c->max_n) return LIBBSC_BAD_PARAMETER;
    const int ticket = p->next_ticket;
    bscgpu_pipe::Lane& L = p->lanes[ticket % p->depth];
    lane_join(p, L);
    BlockJob& J = *L.job;
    J.slot = &p->c->slots[ticket % p->depth];
    if (n <= LIBBSC_HEADER_SIZE) {
        J.c = p->c; J.n = J.n_orig = n; J.output = output; J.lz.reset();
        J.result = bsc_store(input, output, n, features);
        J.stored_small = true;
        return pipe_enqueue(p, L, ticket);
    }
    rc = stage_host_lzp(J, input, output, n, lzpHashSize, lzpMinLen, &blockSorter, coder, features, mode);
    if (rc < 0) return rc;
    rc = stage_host_h2d(J, p->c);
    if (rc < 0) return rc;
    rc = gpu_stage(J, blockSorter);
    if (!J.use_ps) J.lz.reset();                    // the LZP output lives in HBM from here on; a device-model block keeps it for a possible redo
    if (rc < 0) return rc;
    return pipe_enqueue(p, L, ticket);
}

// Any thread: block until the HOST stage of `ticket` has finished and, if the block is then complete, hand out its result — without
// retiring the ticket (bscgpu_pipe_wait on the submitting thread still does that).  Returns 1 with *result set; 0 when the block needs
// its submitting thread after all (a redo on the host model runs the GPU stage again) or the ticket has already been retired / its
// lane reused; a negative code for bad arguments.  For collectors that take blocks in order while the submitting thread is busy
// with the GPU stages of later blocks (job.cpp): a finished block's output buffer is final from here on.
int bscgpu_pipe_peek(bscgpu_pipe* p, int ticket, int* result)
{
    if (!p || !result || ticket < 0) return LIBBSC_BAD_PARAMETER;
    bscgpu_pipe::Lane& L = p->lanes[ticket % p->depth];
    std::unique_lock<std::mutex> lk(p->pool->mu);
    if (ticket >= p->next_ticket) return LIBBSC_BAD_PARAMETER;       // not submitted (yet): nothing to wait for
    p->pool->cv_done.wait(lk, [&] { return L.ticket != ticket || L.joined || L.job->done; });
    if (L.ticket != ticket || L.joined) return 0;
    if (L.job->redo.load(std::memory_order_relaxed)) return 0;
    *result = L.job->result;
    return 1;
}

int bscgpu_pipe_wait(bscgpu_pipe* p, int ticket)
{
    if (!p || ticket < 0 || ticket >= p->next_ticket) return LIBBSC_BAD_PARAMETER;
    bscgpu_pipe::Lane& L = p->lanes[ticket % p->depth];
    if (L.ticket != ticket) return LIBBSC_BAD_PARAMETER;   // already overwritten by a later submit
    lane_join(p, L);
    return L.job->result;
}

// ---- `synth-text v1` -----------------------------------------------------------------
int bsc_synth_text_v1(unsigned long long seed, unsigned char* out, long long n)
{
    if (seed == 0 || !out || n < 0) return LIBBSC_BAD_PARAMETER;
    unsigned long long s = seed;
    auto next = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821657736338717ull; };
    static thread_local unsigned char vocab[4096][10];
    static thread_local unsigned char vlen[4096];
    for (int w = 0; w < 4096; ++w) {
        const int len = 2 + (int)(next() % 8);
        vlen[w] = (unsigned char)len;
        for (int i = 0; i < len; ++i) vocab[w][i] = (unsigned char)('a' + next() % 26);
    }
    long long pos = 0, words = 0;
    while (pos < n) {
        const unsigned long long r = next();
        const unsigned idx = (unsigned)(((r & 0xfff) * ((r >> 12) & 0xfff)) >> 12);
        for (int i = 0; i < vlen[idx] && pos < n; ++i) out[pos++] = vocab[idx][i];
        ++words;
        if (pos < n) out[pos++] = (words % 16 == 0) ? '\n' : ' ';
    }
    return LIBBSC_NO_ERROR;
}

}  // extern "C"
