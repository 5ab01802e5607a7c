// job.cpp — multi-GPU block driver in C++ behind a C ABI: N devices x contexts x blocks in flight, one queue of blocks, results
// collected in block order.
//
// Role: the reference's own block parallelism is the CLI's OpenMP team — every thread reads the next block under `critical(input)`,
// calls bsc_compress and writes under `critical(output)` (bsc.cpp:182-199, :218-221, :397-400); it knows one GPU and one lock
// (bwt.cpp:50-52).  Here the same shape is spread over every GPU of a node from ONE process: blocks are independent (own header,
// own Adler-32s, own model state), so block b simply goes to whichever pipe is free next — a work queue, which on equal GPUs is the
// north star's "one block per GPU" with load balancing for free — and the caller collects the compressed blocks in index order
// (bscgpu_job_wait), which is the `critical(output)` half.  Inside one process there is nothing to exchange between GPUs: a block's
// compressed bytes are produced by host threads of this process in host memory.  (The one-process-per-GPU layout that bench.py's
// contract prescribes does have an exchange step — variable-size blocks to rank 0 — and that one runs over RCCL: libbsc_amd/multigpu.py.)
//
// One worker thread per pipe = per (device, context): it owns its bscgpu context, submits blocks with bscgpu_pipe_submit_host
// (LZP on the host, one H2D copy, GPU stage) keeping `depth` of them in flight, and publishes each result as its ticket completes.
// The host coding of all pipes runs on the process-wide coder pool (block.cpp).  The executor behind a pipe is a table of function
// pointers: the default is the bscgpu_* entry points; tests drive the scheduler with a CPU stand-in (tests/test_job_driver.py), and
// an embedder can put its own transport there.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../include/libbsc.h"
#include "../../../include/bscgpu.h"

namespace {

struct Block {
    const uint8_t* input; uint8_t* output; int n, lzpHashSize, lzpMinLen, sorter, coder, features;
    int result = 0; int worker = -1; bool done = false;
    int ticket = -1;                    // on its worker's pipe, once the GPU stage has run
};

struct Job;
struct Worker {
    Job* job = nullptr; int id = 0, device = 0;
    int dev_index = 0, ordinal = 0;      // index of the device in the job's list; this worker is the device's ordinal-th context
    unsigned burst_seen = ~0u;           // the burst (see Job::burst) this worker last took a block in
    void* ctx = nullptr; void* pipe = nullptr;
    std::thread th;
    int setup_rc = 0; bool ready = false;
    uint64_t blocks = 0;
};

struct Job {
    bscgpu_job_backend be;
    int depth = 2;
    int64_t max_block = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_ready;
    std::deque<Block> blocks;          // by block number (a deque: references stay valid while blocks are appended)
    size_t next = 0;                    // first block nobody has taken yet
    bool closing = false;
    std::vector<Worker> workers;
    bool own_pipes = false;             // the default executor: a collector may look at a pipe's finished blocks itself (bscgpu_pipe_peek)
    int ndev = 1;
    long long expected = -1;            // blocks the caller has announced (bscgpu_job_expect); -1: unknown
    std::vector<unsigned> stages_done;  // per device: GPU stages (submits) that have returned in the current burst
    size_t active = 0;                  // blocks taken and not finished yet
    size_t in_gpu_stage = 0;            // blocks taken whose submit (GPU stage) has not returned yet
    unsigned burst = 0;                 // a burst begins when a block is taken while nothing is in flight anywhere (job start, or the caller let it run dry)

    // Which blocks a worker may take (caller holds mu).  Steady state wants every context busy — kernels of different blocks interleave on
    // a GPU —, the two ENDS of a job do not (measured with bench.py's block queue, rounds 3-4):
    //   head  all contexts of a GPU starting at once interleave their first GPU stages, which then end together after contexts x 13 ms
    //         while the coder threads idle: a burst is begun by a first context, and the k-th context of a device joins it once k GPU
    //         stages of the burst have finished there;
    //   tail  contexts that each hold one of the last blocks finish them in one burst, and the host coding of all of them is left for
    //         the end: with the total announced, the k-th context of a device takes a block only while more than k x devices are left
    //         (the last block of the job goes to a first context alone, the last 2 x devices to first and second contexts, ...).
    bool may_take(const Worker& w) const
    {
        if (next >= blocks.size()) return false;
        // (not for a context's very first block: its first GPU stage also allocates its device-coder arena and landing zones — ~90 ms
        // that the contexts of a cold process had better spend side by side, as a short file job does: bsc_mgpu)
        if (!closing && w.blocks != 0) {
            if (active == 0) { if (w.ordinal != 0) return false; }
            else if (w.burst_seen != burst && stages_done[(size_t)w.dev_index] < (unsigned)w.ordinal) return false;
        }
        if (expected >= 0 && (long long)blocks.size() <= expected && expected - (long long)next <= (long long)w.ordinal * ndev) return false;
        return true;
    }

    void run(Worker& w);
};

void Job::run(Worker& w)
{
    // the worker owns its context and pipe: created here so that context creation (arena, pinned buffers) runs in parallel over devices
    int rc = be.ctx_create(be.user, &w.ctx, w.device, max_block);
    if (rc == LIBBSC_NO_ERROR) {
        rc = be.pipe_create(be.user, w.ctx, depth, &w.pipe);
        if (rc != LIBBSC_NO_ERROR) { w.pipe = nullptr; be.ctx_destroy(be.user, w.ctx); w.ctx = nullptr; }     // a failed set-up leaves nothing behind
    } else w.ctx = nullptr;
    { std::lock_guard<std::mutex> lk(mu); w.setup_rc = rc; w.ready = true; }
    cv_ready.notify_all();
    if (rc != LIBBSC_NO_ERROR) return;

    std::deque<std::pair<int, size_t>> inflight;      // (ticket, block number), oldest first
    auto retire = [&] {
        const auto [ticket, b] = inflight.front(); inflight.pop_front();
        const int res = be.pipe_wait(be.user, w.pipe, ticket);
        bool idle;
        { std::lock_guard<std::mutex> lk(mu); blocks[b].result = res; blocks[b].done = true; blocks[b].ticket = -1; idle = --active == 0; }
        cv_done.notify_all();
        if (idle) cv_work.notify_all();                       // the next burst is a first context's to begin
    };
    for (;;) {
        // Room first, then a claim: a worker that claimed with `depth` blocks in flight would sit on a block no idle worker can take
        // while it waits for its own oldest one (the tail of a job, uneven block times).
        if ((int)inflight.size() == depth) { retire(); continue; }
        size_t b = 0; bool have = false; Block* Bp = nullptr; int features = 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            // with blocks of its own in flight a worker never sleeps on the queue: their results must reach the collector
            // (closing = no more blocks will be added; what is still queued is processed, by the contexts the tail rule leaves it to)
            const auto all_taken = [&] { return closing && next >= blocks.size(); };
            if (inflight.empty()) cv_work.wait(lk, [&] { return all_taken() || may_take(w); });
            // (the element's address is taken under the lock: a deque never moves its elements on push_back, but indexing it while
            // another thread appends is a race on its block map)
            if (may_take(w)) {
                if (active == 0) { ++burst; for (auto& sd : stages_done) sd = 0; }
                ++active; ++in_gpu_stage; w.burst_seen = burst;
                b = next++; Bp = &blocks[b]; Bp->worker = w.id; have = true; ++w.blocks;
                features = Bp->features;
                // the job's last blocks: short host tasks whatever the coder pool's load — the drain of the pipeline is the caller's time.
                // With the library's own pipes the coder pool knows the total and decides by the order in which GPU stages END
                // (bscgpu_job_expect -> bscgpu_coder_pool_expect); another executor gets the mark at submission.
                if (!own_pipes && expected >= 0 && (long long)blocks.size() <= expected && expected - (long long)b <= (long long)workers.size())
                    features |= BSCGPU_FEATURE_LOW_LATENCY;
            } else if (inflight.empty()) {
                if (all_taken()) break;
                continue;
            }
        }
        if (have) {
            Block& B = *Bp;
            const int ticket = be.pipe_submit_host(be.user, w.pipe, B.input, B.output, B.n, B.lzpHashSize, B.lzpMinLen, B.sorter, B.coder, features);
            { std::lock_guard<std::mutex> lk(mu); ++stages_done[(size_t)w.dev_index]; --in_gpu_stage; if (ticket < 0) { B.result = ticket; B.done = true; --active; } else B.ticket = ticket; }
            cv_work.notify_all();                             // a context waiting for this device's k-th stage may start now
            cv_done.notify_all();                             // a collector waiting for this block: it failed, or it has a ticket to look at now
            if (ticket >= 0) inflight.emplace_back(ticket, b);
        } else retire();                                      // nothing to take right now: drain the oldest, then look again
    }
    be.pipe_destroy(be.user, w.pipe); w.pipe = nullptr;
    be.ctx_destroy(be.user, w.ctx); w.ctx = nullptr;
}

// ---- the default executor: this library's own contexts and pipes ---------------------------------------------------------------
int d_ctx_create(void*, void** ctx, int device, int64_t max_n) { bscgpu_ctx* c = nullptr; const int rc = bscgpu_create(&c, device, max_n); *ctx = c; return rc; }
void d_ctx_destroy(void*, void* ctx) { bscgpu_destroy((bscgpu_ctx*)ctx); }
int d_pipe_create(void*, void* ctx, int depth, void** pipe) { bscgpu_pipe* p = nullptr; const int rc = bscgpu_pipe_create((bscgpu_ctx*)ctx, depth, &p); *pipe = p; return rc; }
void d_pipe_destroy(void*, void* pipe) { bscgpu_pipe_destroy((bscgpu_pipe*)pipe); }
int d_submit(void*, void* pipe, const uint8_t* in, uint8_t* out, int n, int lh, int lm, int sorter, int coder, int features)
{ return bscgpu_pipe_submit_host((bscgpu_pipe*)pipe, in, out, n, lh, lm, sorter, coder, features); }
int d_wait(void*, void* pipe, int ticket) { return bscgpu_pipe_wait((bscgpu_pipe*)pipe, ticket); }

}  // namespace

struct bscgpu_job { Job j; };

extern "C" {

int bscgpu_job_create_ex(bscgpu_job** out, const int* devices, int ndevices, int contexts_per_device, int depth, int64_t max_block_bytes,
                         const bscgpu_job_backend* backend)
{
    if (!out || ndevices < 0 || contexts_per_device < 1 || contexts_per_device > 8 || depth < 1 || depth > 8 || max_block_bytes < 0) return LIBBSC_BAD_PARAMETER;
    *out = nullptr;
    std::vector<int> devs;
    if (ndevices == 0) {                                   // every visible device (the default executor's view)
        const int n = backend ? 0 : bscgpu_device_count();
        if (n <= 0) return backend ? LIBBSC_BAD_PARAMETER : LIBBSC_GPU_NOT_SUPPORTED;
        for (int d = 0; d < n; ++d) devs.push_back(d);
    } else {
        if (!devices) return LIBBSC_BAD_PARAMETER;
        devs.assign(devices, devices + ndevices);
    }
    bscgpu_job* J = new bscgpu_job;
    Job& j = J->j;
    if (backend) j.be = *backend;
    else { j.be = bscgpu_job_backend{nullptr, d_ctx_create, d_ctx_destroy, d_pipe_create, d_pipe_destroy, d_submit, d_wait}; j.own_pipes = true; }
    j.depth = depth; j.max_block = max_block_bytes;
    // worker w: context w / ndev of device w % ndev — the first context of every device comes before anybody's second
    j.workers.resize(devs.size() * (size_t)contexts_per_device);
    j.ndev = (int)devs.size();
    j.stages_done.assign(devs.size(), 0u);
    for (size_t w = 0; w < j.workers.size(); ++w) {
        Worker& W = j.workers[w];
        W.job = &j; W.id = (int)w; W.dev_index = (int)(w % devs.size()); W.ordinal = (int)(w / devs.size()); W.device = devs[(size_t)W.dev_index];
    }
    for (auto& w : j.workers) w.th = std::thread([&j, &w] { j.run(w); });
    // all contexts up before the first block: a device that cannot be set up fails the job here, not in the middle of it
    int rc = LIBBSC_NO_ERROR;
    {
        std::unique_lock<std::mutex> lk(j.mu);
        j.cv_ready.wait(lk, [&] { for (auto& w : j.workers) if (!w.ready) return false; return true; });
        for (auto& w : j.workers) if (w.setup_rc != LIBBSC_NO_ERROR && rc == LIBBSC_NO_ERROR) rc = w.setup_rc;
    }
    if (rc != LIBBSC_NO_ERROR) { bscgpu_job_destroy(J); return rc; }
    *out = J;
    return LIBBSC_NO_ERROR;
}

int bscgpu_job_create(bscgpu_job** out, const int* devices, int ndevices, int contexts_per_device, int depth, int64_t max_block_bytes)
{ return bscgpu_job_create_ex(out, devices, ndevices, contexts_per_device, depth, max_block_bytes, nullptr); }

int bscgpu_job_add(bscgpu_job* J, const uint8_t* input, uint8_t* output, int n, int lzpHashSize, int lzpMinLen, int blockSorter, int coder, int features)
{
    if (!J || !input || !output || n < 0 || (int64_t)n > J->j.max_block) return LIBBSC_BAD_PARAMETER;
    Job& j = J->j;
    int number;
    {
        std::lock_guard<std::mutex> lk(j.mu);
        if (j.closing) return LIBBSC_BAD_PARAMETER;
        number = (int)j.blocks.size();
        j.blocks.push_back(Block{input, output, n, lzpHashSize, lzpMinLen, blockSorter, coder, features});
    }
    j.cv_work.notify_all();             // (all: the worker a notify_one picks may be one the head / tail rule keeps waiting)
    return number;
}

int bscgpu_job_expect(bscgpu_job* J, int total_blocks)
{
    if (!J || total_blocks < 0) return LIBBSC_BAD_PARAMETER;
    Job& j = J->j;
    long long to_come;
    { std::lock_guard<std::mutex> lk(j.mu); j.expected = total_blocks; to_come = total_blocks - (long long)j.next + (long long)j.in_gpu_stage; }
    if (j.own_pipes) (void)bscgpu_coder_pool_expect(to_come > 0 ? to_come : -1, j.ndev);      // blocks whose host work has not been queued yet
    j.cv_work.notify_all();
    return LIBBSC_NO_ERROR;
}

int bscgpu_job_wait(bscgpu_job* J, int block)
{
    if (!J || block < 0) return LIBBSC_BAD_PARAMETER;
    Job& j = J->j;
    std::unique_lock<std::mutex> lk(j.mu);
    if ((size_t)block >= j.blocks.size()) return LIBBSC_BAD_PARAMETER;
    Block& B = j.blocks[(size_t)block];
    // A worker publishes a block when it RETIRES it — when it needs the lane again or has nothing else to do —, which for a block that
    // was coded long ago can be a GPU stage or two later: an in-order collector (a file writer, job_bench) then holds its buffers back
    // and the queue runs dry (round 5: 4005 MB/s through the job against 5100 through bench.py's independent pipes).  So the collector
    // looks at the pipe itself: the block is final as soon as its host stage is over.
    while (!B.done) {
        if (j.own_pipes && B.ticket >= 0 && !j.closing) {        // (a closing job's workers tear their pipes down: no peeking then)
            bscgpu_pipe* pipe = (bscgpu_pipe*)j.workers[(size_t)B.worker].pipe;
            const int ticket = B.ticket;
            lk.unlock();
            int res = 0;
            const int got = bscgpu_pipe_peek(pipe, ticket, &res);
            lk.lock();
            if (got == 1) return res;                     // (the worker still retires the ticket and publishes the same result)
            if (B.done) break;
            // needs its worker (redo) or was retired meanwhile: the worker's publication follows
            j.cv_done.wait(lk, [&] { return B.done; });
            break;
        }
        j.cv_done.wait(lk, [&] { return B.done || (j.own_pipes && B.ticket >= 0 && !j.closing); });
    }
    return B.result;
}

int bscgpu_job_block_worker(bscgpu_job* J, int block, int* device)
{
    if (!J || block < 0) return LIBBSC_BAD_PARAMETER;
    Job& j = J->j;
    std::lock_guard<std::mutex> lk(j.mu);
    if ((size_t)block >= j.blocks.size() || j.blocks[(size_t)block].worker < 0) return LIBBSC_BAD_PARAMETER;
    const int w = j.blocks[(size_t)block].worker;
    if (device) *device = j.workers[(size_t)w].device;
    return w;
}

void bscgpu_job_destroy(bscgpu_job* J)
{
    if (!J) return;
    Job& j = J->j;
    { std::lock_guard<std::mutex> lk(j.mu); j.closing = true; }
    j.cv_work.notify_all();
    for (auto& w : j.workers) if (w.th.joinable()) w.th.join();     // queued blocks are still processed: destroy = finish, then tear down
    delete J;
}

}  // extern "C"
