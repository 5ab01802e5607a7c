// context.hip — per-GPU context (HBM arena, stream, HIP-event profiling) and the host-pointer C ABI.
//
// Mirrors the role of libcubwt's device storage object (libcubwt.cu:2239-2395) but is sized for
// 288 GB of HBM3E: one hipMalloc of ~60 bytes per block byte, carved once, reused for every block.
#include "dev_common.h"
#include "dma_copy.h"
#include <sys/mman.h>
#include <system_error>
#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>

int ctx_fail(bscgpu_ctx* c, int code, const char* what, hipError_t e)
{
    if (c) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s", what, e == hipSuccess ? "failed" : hipGetErrorString(e));
        c->err = buf;
    }
    return code;
}

bool ctx_timing_on() { static const bool on = getenv("BSCGPU_TIMING") != nullptr; return on; }
static double ctx_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
CtxTimer::CtxTimer(const char* w) : what(w), t0(0), on(ctx_timing_on()) { if (on) t0 = ctx_now_ms(); }
CtxTimer::~CtxTimer() { if (on) fprintf(stderr, "[bscgpu timing] %-34s %8.1f ms (thread %zu)\n", what, ctx_now_ms() - t0, std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000); }

hipError_t ctx_sync(bscgpu_ctx* c)
{
    hipError_t e = hipEventRecord(c->sync_ev, c->stream);
    if (e != hipSuccess) return e;
    return hipEventSynchronize(c->sync_ev);
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int ctx_ensure_slots(bscgpu_ctx* c, int count)
{
    if (count > MAX_SLOTS) return BSC_BAD_PARAMETER;
    const size_t N = align_up((size_t)c->max_n + 4096, 4096);
    for (; c->nslots < count; ++c->nslots) {
        HostSlot& s = c->slots[c->nslots];
        // the landing zones themselves (6 N bytes of run arrays, 8 N of probability stream) are pinned on first use: a block of
        // the device model never copies its run arrays, a block of the host model has no probability stream
        (void)N;
        if (hipEventCreateWithFlags(&s.copy_ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) return BSC_NOT_ENOUGH_MEMORY;
        for (int b = 0; b < 8; ++b)
            if (hipEventCreateWithFlags(&s.part_ev[b], hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) return BSC_NOT_ENOUGH_MEMORY;
        if (dma_available())
            for (int b = 0; b < 8; ++b) s.part_sig[b] = dma_signal_create();       // (0 if the runtime has no more: that slot's copies then go through HIP)
    }
    return BSC_NO_ERROR;
}

// ---- large pinned landing zones: anonymous memory, touched in parallel, then registered ------------------------------------------
// hipHostMalloc of a block's landing zone (366 MB of probability stream for 64 MiB of text) takes 50-60 ms on this platform, the calls
// of different threads serialise (six at once: 370 ms), and hipHostFree another 30 ms — a job of a few dozen blocks spent more time
// allocating than compressing (round 4: 32 x 64 MiB through bsc_mgpu in 2.3 s, of which ~0.5 s of GPU work).  The same memory as an
// anonymous mapping with transparent huge pages, first-touched by eight threads (2.5 ms; zeroing pages is the cost, 14 ms on one
// thread) and then hipHostRegister'ed (0.7 ms) copies at the same 57 GB/s (tools/startup_probe.cpp, profiles/r04/startup_probe.txt).
static constexpr size_t PIN_ALIGN = (size_t)2 << 20;
static std::mutex g_hostmalloc_mu;
static std::vector<void*> g_hostmalloc;          // landing zones that came from hipHostMalloc (registration refused): freed accordingly
static void* pinned_alloc(size_t bytes)
{
    CtxTimer tm("pinned landing zone");
    bytes = align_up(bytes, PIN_ALIGN);
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return nullptr;
    static const bool thp = [] { const char* e = getenv("BSC_PIN_THP"); return e ? atoi(e) != 0 : true; }();
    if (thp) (void)madvise(m, bytes, MADV_HUGEPAGE);
    const int nth = bytes >= ((size_t)32 << 20) ? 8 : 1;
    auto touch = [m, bytes, nth](int k) {
        volatile char* p = (volatile char*)m;
        const size_t lo = bytes / (size_t)nth * (size_t)k, hi = (k == nth - 1) ? bytes : bytes / (size_t)nth * (size_t)(k + 1);
        for (size_t i = lo; i < hi; i += 4096) p[i] = 0;
    };
    std::vector<std::thread> th;
    int started = 0;
    for (; started < nth - 1; ++started) {
        try { th.emplace_back(touch, started); } catch (const std::system_error&) { break; }       // no more threads: the rest here
    }
    for (int k = started; k < nth; ++k) touch(k);
    for (auto& t : th) t.join();
    static const bool refuse = getenv("BSC_PIN_REGISTER_FAIL") != nullptr;                          // tests: a platform that refuses the registration
    if (!refuse && hipHostRegister(m, bytes, hipHostRegisterDefault) == hipSuccess) return m;
    // Registration refused (RLIMIT_MEMLOCK, a container without large-page registration, other THP settings): the runtime's own pinned
    // allocation — slower to get (50-60 ms), but the block keeps its landing zone instead of failing outright.
    (void)hipGetLastError();
    munmap(m, bytes);
    void* h = nullptr;
    if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (ctx_timing_on()) fprintf(stderr, "[bscgpu] hipHostRegister refused %zu bytes: hipHostMalloc instead\n", bytes);
    { std::lock_guard<std::mutex> g(g_hostmalloc_mu); g_hostmalloc.push_back(h); }
    return h;
}
static void pinned_free(void* p, size_t bytes)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_hostmalloc_mu);
        for (size_t i = 0; i < g_hostmalloc.size(); ++i)
            if (g_hostmalloc[i] == p) { g_hostmalloc[i] = g_hostmalloc.back(); g_hostmalloc.pop_back(); (void)hipHostFree(p); return; }
    }
    (void)hipHostUnregister(p);
    munmap(p, align_up(bytes, PIN_ALIGN));
}

int ctx_ensure_run_slot(bscgpu_ctx* c, HostSlot& s)
{
    if (s.hsym && s.hrank && s.hstart) return BSC_NO_ERROR;
    const size_t N = align_up((size_t)c->max_n + 4096, 4096);
    // one mapping for the three run arrays: symbol (N), rank (N), start (4 N)
    if (!s.run_base) {
        s.run_bytes = 6 * N;
        s.run_base = (u8*)pinned_alloc(s.run_bytes);
        if (!s.run_base) { s.run_bytes = 0; return BSC_NOT_ENOUGH_MEMORY; }
    }
    s.hsym = s.run_base; s.hrank = s.run_base + N; s.hstart = reinterpret_cast<u32*>(s.run_base + 2 * N);
    return BSC_NO_ERROR;
}

int ctx_ensure_pstream_slot(bscgpu_ctx* c, HostSlot& slot, size_t entries)
{
    (void)c;
    if (slot.hps_cap >= entries) return BSC_NO_ERROR;
    if (slot.hps) { pinned_free(slot.hps, slot.hps_cap * 2); slot.hps = nullptr; slot.hps_cap = 0; slot.hps_dev = nullptr; }
    entries += entries / 16;                                    // a little room: the next block's stream is rarely exactly this long
    slot.hps = (u16*)pinned_alloc(entries * 2);
    if (!slot.hps) return BSC_NOT_ENOUGH_MEMORY;
    slot.hps_cap = entries;
    slot.hps_dev = nullptr;
    if (hipHostGetDevicePointer(&slot.hps_dev, slot.hps, 0) != hipSuccess) { (void)hipGetLastError(); slot.hps_dev = nullptr; }
    return BSC_NO_ERROR;
}

extern "C" int bscgpu_d2h_dma_available(void) { return dma_available(); }

extern "C" int bscgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#ifndef ARENA_ALIGN
#define ARENA_ALIGN 256          // alignment of every buffer carved from the arena
#endif
extern "C" int bscgpu_create(bscgpu_ctx** out, int device, int64_t max_n)
{
    if (!out || max_n < 0 || max_n >= 0x7fffffffll) return BSC_BAD_PARAMETER;
    *out = nullptr;
    CtxTimer tm_all("bscgpu_create");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BSC_GPU_NOT_SUPPORTED;
    if (device < 0 || device >= ndev) return BSC_BAD_PARAMETER;
    devcoder_warm_tables();                    // (a thread: the model tables are ready by the time the first block needs them)
    if (hipSetDevice(device) != hipSuccess) return BSC_GPU_ERROR;
    // waits should sleep, not spin: the host CPUs belong to the entropy coder (BSCGPU_SPIN=1 keeps the runtime's default)
    if (!getenv("BSCGPU_SPIN")) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }

    bscgpu_ctx* c = new bscgpu_ctx();
    { CtxTimer tm("  first HIP call on this thread"); (void)hipFree(nullptr); }
    c->device = device;
    c->max_n  = max_n;
    { const char* e = getenv("BSC_DC_SPF"); c->dc_spf = (e && e[0] == '1') ? 1 : 0; }     // BSCGPU_OPT_DC_STREAM_STATIC (off by default: measured slower, profiles/r06)
    { const char* e = getenv("BSC_PS13"); c->dc_p13 = (e && e[0] == '0') ? 0 : 1; }       // BSCGPU_OPT_DC_PACKED_STREAM
    memset(c->kstat, 0, sizeof c->kstat);
    auto tm_streams = std::make_unique<CtxTimer>("  streams + events");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return BSC_GPU_ERROR; }
    if (hipEventCreateWithFlags(&c->sync_ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(c->stream); delete c; return BSC_GPU_ERROR; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { hipEventDestroy(c->sync_ev); hipStreamDestroy(c->stream); delete c; return BSC_GPU_ERROR; }

    tm_streams.reset();
    const size_t N = align_up((size_t)max_n + 4096, 4096);
    struct Carve { void** p; size_t bytes; size_t lead; };
    Carve carve[] = {
        {(void**)&c->dT,      N + 128, 64},      // 64 B of front padding (cyclic byte for ST), tail padding behind
        {(void**)&c->dL,      N + 64, 0},
        {(void**)&c->kA,      8 * N, 0}, {(void**)&c->kB, 8 * N, 0},
        {(void**)&c->vA,      4 * N, 0}, {(void**)&c->vB, 4 * N, 0},
        {(void**)&c->SA,      4 * N, 0}, {(void**)&c->ISA, 4 * N + 64, 0},
        {(void**)&c->cpos[0], 4 * N, 0}, {(void**)&c->cpos[1], 4 * N, 0},
        {(void**)&c->csa[0],  4 * N, 0}, {(void**)&c->csa[1],  4 * N, 0},
        {(void**)&c->cgrp[0], 4 * N, 0}, {(void**)&c->cgrp[1], 4 * N, 0},
        {(void**)&c->flags,   N + 64, 0},
        {(void**)&c->counts,  (size_t)256 * MAX_CHUNKS * 4, 0},
        {(void**)&c->rowtot,  256 * 4, 0},
        {(void**)&c->segsum,  2 * MAX_CHUNKS * 4, 0},
        {(void**)&c->segoff,  2 * MAX_CHUNKS * 4, 0},
        {(void**)&c->dscal,   1024 * 4, 0},
        {(void**)&c->dscal64, 16 * 8, 0},
        {(void**)&c->adler_part, (size_t)MAX_CHUNKS * 16, 0},
        {(void**)&c->wc_sink, (size_t)512 * 1024 * 8, 0},
    };
    size_t total = 0;
    for (auto& cv : carve) total += align_up(cv.bytes, ARENA_ALIGN);
    auto tm_arena = std::make_unique<CtxTimer>("  arena hipMalloc + memset + small pinned + engine setup");
    hipError_t e = hipMalloc((void**)&c->arena, total);
    if (e != hipSuccess) { hipStreamDestroy(c->stream); delete c; return BSC_GPU_NOT_ENOUGH_MEMORY; }
    c->arena_bytes = total;
    size_t off = 0;
    for (auto& cv : carve) { *cv.p = c->arena + off + cv.lead; off += align_up(cv.bytes, ARENA_ALIGN); }
    if (hipMemsetAsync(c->arena, 0, total, c->stream) != hipSuccess) { bscgpu_destroy(c); return BSC_GPU_ERROR; }

    bool ok = hipHostMalloc((void**)&c->hscal, 1024 * 4, hipHostMallocDefault) == hipSuccess
           && hipHostMalloc((void**)&c->hscal64, 16 * 8, hipHostMallocDefault) == hipSuccess
           && hipHostMalloc((void**)&c->hadler, (size_t)MAX_CHUNKS * 16, hipHostMallocDefault) == hipSuccess
           && hipHostMalloc((void**)&c->hsplit, N / 256 + 64, hipHostMallocDefault) == hipSuccess
           && ctx_ensure_slots(c, 1) == BSC_NO_ERROR;
    if (!ok || ctx_sync(c) != hipSuccess) { bscgpu_destroy(c); return BSC_GPU_NOT_ENOUGH_MEMORY; }
    if (radix_engine_setup(c) != BSC_NO_ERROR) { bscgpu_destroy(c); return BSC_GPU_ERROR; }
    tm_arena.reset();
    *out = c;
    return BSC_NO_ERROR;
}

extern "C" void bscgpu_destroy(bscgpu_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) ctx_sync(c);
    if (c->tile_counts) { (void)hipFree(c->tile_counts); c->tile_counts = nullptr; }
    if (c->os_agg) { (void)hipFree(c->os_agg); c->os_agg = nullptr; }
    if (c->os_zero) { (void)hipFree(c->os_zero); c->os_zero = nullptr; }
    if (c->long_tables) { (void)hipFree(c->long_tables); c->long_tables = nullptr; }
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (auto& p : c->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto& e : c->event_pool) hipEventDestroy(e);
    if (c->hscal) hipHostFree(c->hscal);
    if (c->hscal64) hipHostFree(c->hscal64);
    if (c->hadler) hipHostFree(c->hadler);
    if (c->hsplit) hipHostFree(c->hsplit);
    for (int i = 0; i < MAX_SLOTS; ++i) {
        HostSlot& s = c->slots[i];
        if (s.run_base) pinned_free(s.run_base, s.run_bytes);
        for (int b = 0; b < 8; ++b) if (s.part_sig[b]) (void)dma_wait(s.part_sig[b]);     // (a signal that was never armed reads 0)
        if (s.hps) pinned_free(s.hps, s.hps_cap * 2);
        if (s.copy_ev) hipEventDestroy(s.copy_ev);
        for (int b = 0; b < 8; ++b) if (s.part_ev[b]) hipEventDestroy(s.part_ev[b]);
        for (int b = 0; b < 8; ++b) if (s.part_sig[b]) dma_signal_destroy(s.part_sig[b]);
    }
    devcoder_destroy(c);
    if (c->arena) hipFree(c->arena);
    if (c->sync_ev) hipEventDestroy(c->sync_ev);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int64_t bscgpu_arena_bytes(const bscgpu_ctx* c) { return c ? (int64_t)c->arena_bytes : 0; }
extern "C" const char* bscgpu_last_error(const bscgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

// ---- profiling --------------------------------------------------------------------------------
static hipEvent_t take_event(bscgpu_ctx* c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
void prof_begin(bscgpu_ctx* c, int kind, u64 bytes, u64 records)
{
    if (!c->prof) return;
    bscgpu_ctx::Pending p;
    p.a = take_event(c); p.b = take_event(c); p.kind = kind; p.bytes = bytes; p.records = records;
    hipEventRecord(p.a, c->stream);
    c->pending.push_back(p);
}
void prof_end(bscgpu_ctx* c)
{
    if (!c->prof || c->pending.empty()) return;
    hipEventRecord(c->pending.back().b, c->stream);
}
void prof_collect(bscgpu_ctx* c)
{
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->kstat[p.kind].ms += ms;
            c->kstat[p.kind].launches += 1;
            c->kstat[p.kind].bytes += p.bytes;
            c->kstat[p.kind].records += p.records;
            if (p.kind == BSCGPU_K_RADIX_SCATTER && c->scatter_log.size() < 65536)
                c->scatter_log.push_back({(double)ms, p.records});
        }
        c->event_pool.push_back(p.a); c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}
extern "C" void bscgpu_profile_enable(bscgpu_ctx* c, int on) { if (c) c->prof = (on != 0); }
extern "C" void bscgpu_profile_reset(bscgpu_ctx* c)
{
    if (!c) return;
    memset(c->kstat, 0, sizeof c->kstat);
    c->scatter_log.clear();
}
extern "C" int bscgpu_profile_get(bscgpu_ctx* c, bscgpu_kstat* stats)
{
    if (!c || !stats) return BSC_BAD_PARAMETER;
    memcpy(stats, c->kstat, sizeof c->kstat);
    return BSC_NO_ERROR;
}
extern "C" int bscgpu_profile_scatter_launches(bscgpu_ctx* c, double* ms, uint64_t* records, int max)
{
    if (!c) return 0;
    int cnt = (int)c->scatter_log.size();
    int from = cnt > max ? cnt - max : 0;
    for (int i = from; i < cnt; ++i) { ms[i - from] = c->scatter_log[i].ms; records[i - from] = c->scatter_log[i].records; }
    return cnt - from;
}
extern "C" int bscgpu_option_set(bscgpu_ctx* c, int key, int value)
{
    if (!c) return BSC_BAD_PARAMETER;
    if (key == BSCGPU_OPT_RS_ONESWEEP && value >= 0 && value <= 3 && (value == 0 || c->os_available)) { const int old = c->os_mode; c->os_mode = value; return old; }
    if (key == BSCGPU_OPT_DC_STREAM_STATIC && (value == 0 || value == 1)) { const int old = c->dc_spf; c->dc_spf = value; return old; }
    if (key == BSCGPU_OPT_DC_PACKED_STREAM && (value == 0 || value == 1)) { const int old = c->dc_p13; c->dc_p13 = value; return old; }
    return BSC_BAD_PARAMETER;
}
extern "C" int bscgpu_option_get(bscgpu_ctx* c, int key)
{
    if (!c) return BSC_BAD_PARAMETER;
    if (key == BSCGPU_OPT_RS_ONESWEEP) return c->os_mode;
    if (key == BSCGPU_CNT_OS_RETRIES) return c->os_retries;
    if (key == BSCGPU_OPT_DC_STREAM_STATIC) return c->dc_spf;
    if (key == BSCGPU_OPT_DC_PACKED_STREAM) return c->dc_p13;
    return BSC_BAD_PARAMETER;
}
extern "C" int bscgpu_last_stage_ms(bscgpu_ctx* c, double* out6)
{
    if (!c || !out6) return BSC_BAD_PARAMETER;
    for (int i = 0; i < 6; ++i) out6[i] = c->stage_ms[i];
    return BSC_NO_ERROR;
}

// ---- C ABI: device-pointer entry points --------------------------------------------------------
extern "C" int64_t bscgpu_bwt_device(bscgpu_ctx* c, const void* dT, void* dL, int64_t n, int64_t r, uint32_t* I)
{
    if (!c || !dT || !dL) return BSC_BAD_PARAMETER;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    int64_t primary = 0;
    int rc = bwt_device(c, (const u8*)dT, (u8*)dL, n, r, I, &primary);
    return rc < 0 ? rc : primary;
}

extern "C" int bscgpu_st_encode_device(bscgpu_ctx* c, const void* dT, void* dOut, int n, int k)
{
    if (!c || !dT || !dOut) return BSC_BAD_PARAMETER;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    int index = 0;
    int rc = st_device(c, (const u8*)dT, (u8*)dOut, n, k, &index);
    return rc < 0 ? rc : index;
}

extern "C" int bscgpu_adler32_device(bscgpu_ctx* c, const void* dT, int64_t n, uint32_t* out)
{
    if (!c || !out || (!dT && n > 0)) return BSC_BAD_PARAMETER;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    return adler32_device(c, (const u8*)dT, n, out);
}

extern "C" int bscgpu_radix_sort_u64(bscgpu_ctx* c, void* keys, void* keys_alt, void* vals, void* vals_alt,
                                     int64_t n, int begin_bit, int end_bit, int* result_in_alt)
{
    if (!c || !result_in_alt || n < 0) return BSC_BAD_PARAMETER;
    if (n == 0) { *result_in_alt = 0; return BSC_NO_ERROR; }
    if (!keys || !keys_alt) return BSC_BAD_PARAMETER;
    if (begin_bit < 0 || end_bit > 64 || begin_bit > end_bit) return BSC_BAD_PARAMETER;
    if ((vals == nullptr) != (vals_alt == nullptr)) return BSC_BAD_PARAMETER;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    RadixPass passes[8]; int np = 0;
    for (int s = begin_bit; s < end_bit; s += 8) { passes[np].shift = s; passes[np].bits = (end_bit - s < 8) ? end_bit - s : 8; ++np; }
    int rc = radix_sort_passes(c, (u64*)keys, (u64*)keys_alt, (u32*)vals, (u32*)vals_alt, (u64)n, passes, np, result_in_alt);
    if (rc < 0) return rc;
    HIP_TRY(c, ctx_sync(c));
    prof_collect(c);
    return radix_onesweep_check(c);
}

// ---- C ABI: host-pointer entry points (the reference's hook shape: H2D, run, D2H, synchronous) ----
static int64_t bwt_host(bscgpu_ctx* c, const uint8_t* T, uint8_t* L, int64_t n, int64_t r, uint32_t* I)
{
    if (!c || !T || !L || n < 0) return BSC_BAD_PARAMETER;
    if (n > c->max_n) return BSC_GPU_NOT_ENOUGH_MEMORY;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    if (n == 0) return 0;
    HIP_TRY(c, hipMemcpyAsync(c->dL, T, (size_t)n, hipMemcpyHostToDevice, c->stream));
    int64_t primary = 0;
    int rc = bwt_device(c, c->dL, c->dL, n, r, I, &primary);
    if (rc < 0) return rc;
    HIP_TRY(c, hipMemcpyAsync(L, c->dL, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    return primary;
}
extern "C" int64_t bscgpu_bwt(bscgpu_ctx* c, const uint8_t* T, uint8_t* L, int64_t n) { return bwt_host(c, T, L, n, 0, nullptr); }
extern "C" int64_t bscgpu_bwt_aux(bscgpu_ctx* c, const uint8_t* T, uint8_t* L, int64_t n, int64_t r, uint32_t* I)
{
    if (!I) return BSC_BAD_PARAMETER;
    int64_t rc = bwt_host(c, T, L, n, r, I);
    return rc < 0 ? rc : 0;
}

extern "C" int bscgpu_st_encode(bscgpu_ctx* c, uint8_t* T, int n, int k)
{
    if (!c || !T || n < 0) return BSC_BAD_PARAMETER;
    if (k < 3 || k > 8) return BSC_BAD_PARAMETER;
    if (n <= 1) return 0;
    if (n > c->max_n) return BSC_GPU_NOT_ENOUGH_MEMORY;
    if (hipSetDevice(c->device) != hipSuccess) return BSC_GPU_ERROR;
    HIP_TRY(c, hipMemcpyAsync(c->dL, T, (size_t)n, hipMemcpyHostToDevice, c->stream));
    int index = 0;
    int rc = st_device(c, c->dL, c->dL, n, k, &index);
    if (rc < 0) return rc;
    HIP_TRY(c, hipMemcpyAsync(T, c->dL, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync(c));
    return index;
}
