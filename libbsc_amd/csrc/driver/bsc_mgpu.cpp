// bsc_mgpu.cpp — file compressor on every GPU of a node, in C++ over the C ABI (include/libbsc.h + include/bscgpu.h): the reference
// CLI's block loop (bsc.cpp:115-430: 'bsc1', nBlocks, per block {int64 offset, int8 recordSize, int8 sortingContexts} + bsc_compress
// output; OpenMP team, next block under critical(input), write under critical(output)) with the team replaced by a bscgpu_job — N
// devices x contexts x blocks in flight behind one queue — and the blocks written in order.  Files are byte-identical to
// `bsc e ... -t` of the reference (same defaults: LZP on -H15 -M128, fast mode, multithreading; no segmentation / reordering, which
// are CLI-side filters outside the hot path) and unpack with the reference's `bsc d`.
//
//   bsc_mgpu e <in> <out> [-b<MiB>] [-m0|-m3..8] [-e0|-e1|-e2] [-H<bits>] [-M<len>] [-p] [-G<gpus>] [-C<contexts per gpu>] [-D<depth>]
//   bsc_mgpu d <in> <out>
//
// The input is streamed: at most (devices x contexts x depth + 4) blocks are in memory; reader threads fill their buffers straight from
// the file, the main thread adds them to the job in file order, a writer appends them — in file order — as they are done.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../../include/libbsc.h"
#include "../../../include/bscgpu.h"

#pragma pack(push, 1)
struct BlockHeader { long long offset; signed char recordSize; signed char sortingContexts; };      // bsc.cpp:52-57
#pragma pack(pop)

static int usage()
{
    fprintf(stderr, "usage: bsc_mgpu e <in> <out> [-b<MiB>] [-m0|-m3..8] [-e0|-e1|-e2] [-H<bits>] [-M<len>] [-p] [-G<gpus>] [-C<contexts>] [-D<depth>]\n"
                    "       bsc_mgpu d <in> <out>\n");
    return 2;
}

static int decompress(const char* in, const char* out)
{
    FILE* fi = fopen(in, "rb"); if (!fi) { perror(in); return 1; }
    FILE* fo = fopen(out, "wb"); if (!fo) { perror(out); fclose(fi); return 1; }
    unsigned char sign[4]; int nblocks = 0;
    if (fread(sign, 1, 4, fi) != 4 || memcmp(sign, "bsc1", 4) != 0 || fread(&nblocks, 4, 1, fi) != 1 || nblocks < 0) { fprintf(stderr, "not a bsc1 file\n"); return 1; }
    std::vector<unsigned char> buf;
    long long total = 0;
    for (int b = 0; b < nblocks; ++b) {
        BlockHeader h; unsigned char hdr[LIBBSC_HEADER_SIZE];
        if (fread(&h, sizeof h, 1, fi) != 1 || fread(hdr, 1, LIBBSC_HEADER_SIZE, fi) != LIBBSC_HEADER_SIZE) { fprintf(stderr, "unexpected end of file\n"); return 1; }
        if (h.recordSize != 1 || h.sortingContexts != 1) { fprintf(stderr, "block %d uses a CLI-side filter (record size %d, contexts %d): not handled here\n", b, h.recordSize, h.sortingContexts); return 1; }
        int blockSize = 0, dataSize = 0;
        if (bsc_block_info(hdr, LIBBSC_HEADER_SIZE, &blockSize, &dataSize, 0) != LIBBSC_NO_ERROR) { fprintf(stderr, "bad block header\n"); return 1; }
        const size_t need = (size_t)(blockSize > dataSize ? blockSize : dataSize) + 64;
        if (buf.size() < need) buf.resize(need);
        memcpy(buf.data(), hdr, LIBBSC_HEADER_SIZE);
        if (fread(buf.data() + LIBBSC_HEADER_SIZE, 1, (size_t)blockSize - LIBBSC_HEADER_SIZE, fi) != (size_t)blockSize - LIBBSC_HEADER_SIZE) { fprintf(stderr, "unexpected end of file\n"); return 1; }
        const int rc = bsc_decompress(buf.data(), blockSize, buf.data(), dataSize, LIBBSC_FEATURE_FASTMODE | LIBBSC_FEATURE_MULTITHREADING);
        if (rc != LIBBSC_NO_ERROR) { fprintf(stderr, "bsc_decompress: %d\n", rc); return 1; }
        if (fseeko(fo, (off_t)h.offset, SEEK_SET) != 0 || fwrite(buf.data(), 1, (size_t)dataSize, fo) != (size_t)dataSize) { perror(out); return 1; }
        total += dataSize;
    }
    fclose(fi); fclose(fo);
    printf("%lld bytes\n", total);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 4 || (strcmp(argv[1], "e") != 0 && strcmp(argv[1], "d") != 0)) return usage();
    if (bsc_init(LIBBSC_FEATURE_FASTMODE | LIBBSC_FEATURE_MULTITHREADING) != LIBBSC_NO_ERROR) return 1;
    if (argv[1][0] == 'd') return decompress(argv[2], argv[3]);

    long long block = 25ll << 20;                              // bsc.cpp:62
    int sorter = LIBBSC_BLOCKSORTER_BWT, coder = LIBBSC_CODER_QLFC_STATIC, lzpHash = 15, lzpMin = 128, gpus = 0, contexts = 2, depth = 3;
    bool lzp = true;
    for (int a = 4; a < argc; ++a) {
        const char* s = argv[a];
        if (s[0] != '-') return usage();
        const int v = atoi(s + 2);
        switch (s[1]) {
            case 'b': block = (long long)v << 20; break;
            case 'm': sorter = v == 0 ? LIBBSC_BLOCKSORTER_BWT : v; break;
            case 'e': coder = v == 0 ? LIBBSC_CODER_QLFC_FAST : v == 2 ? LIBBSC_CODER_QLFC_ADAPTIVE : LIBBSC_CODER_QLFC_STATIC; break;
            case 'H': lzpHash = v; break;
            case 'M': lzpMin = v; break;
            case 'p': lzp = false; break;
            case 'G': gpus = v; break;
            case 'C': contexts = v; break;
            case 'D': depth = v; break;
            default: return usage();
        }
    }
    if (block < 1 << 20 || block > 1 << 30) { fprintf(stderr, "block size out of range\n"); return 2; }

    FILE* fi = fopen(argv[2], "rb"); if (!fi) { perror(argv[2]); return 1; }
    if (fseeko(fi, 0, SEEK_END) != 0) { perror(argv[2]); return 1; }
    const long long size = (long long)ftello(fi);
    fseeko(fi, 0, SEEK_SET);
    const int nblocks = (int)((size + block - 1) / block);
    FILE* fo = fopen(argv[3], "wb"); if (!fo) { perror(argv[3]); return 1; }
    fwrite("bsc1", 1, 4, fo); fwrite(&nblocks, 4, 1, fo);

    const auto t0 = std::chrono::steady_clock::now();
    int ndev = bscgpu_device_count();
    if (gpus > 0 && gpus < ndev) ndev = gpus;
    if (ndev <= 0) { fprintf(stderr, "no usable GPU\n"); return 1; }
    std::vector<int> devs; for (int d = 0; d < ndev; ++d) devs.push_back(d);
    const long long max_block = block < size ? block : (size > 0 ? size : 1);
    bscgpu_job* job = nullptr;
    int rc = bscgpu_job_create(&job, devs.data(), ndev, contexts, depth, max_block);
    if (rc != LIBBSC_NO_ERROR) { fprintf(stderr, "bscgpu_job_create: %d\n", rc); return 1; }
    (void)bscgpu_job_expect(job, nblocks);                    // a file's block count is known: the job tapers its tail and marks the last blocks low-latency

    // Three roles around the job, so that none of them waits for the others' system calls: READERS fill block buffers straight from the
    // file (pread: a 64 MiB block costs 15-25 ms of page-cache copy and page faults, which one thread in front of the GPUs cannot
    // hide: 32 blocks were 0.6 s of a 1.2 s run), the main thread ADDS blocks to the job in file order as they become ready, a WRITER
    // collects them in file order (bscgpu_job_wait) and appends them to the output; a buffer is reused once its block has been written.
    const int window = ndev * contexts * depth + 4;             // blocks in memory
    // (plain malloc, not zero-filled containers: a 64 MiB output buffer is touched only as far as the compressed block reaches)
    struct Buf { unsigned char* p = nullptr; size_t cap = 0; void need(size_t n) { if (cap < n) { free(p); p = (unsigned char*)malloc(n); cap = p ? n : 0; } } ~Buf() { free(p); } };
    std::vector<Buf> ibuf((size_t)window), obuf((size_t)window);
    std::mutex mu; std::condition_variable cv;
    int next_read = 0, added = 0, written = 0; bool failed = false;
    std::vector<char> ready((size_t)nblocks > 0 ? (size_t)nblocks : 1, 0);
    const int fd = fileno(fi);
    auto block_len = [&](int b) -> long long { return (b == nblocks - 1) ? size - (long long)b * block : block; };
    auto reader = [&] {
        for (;;) {
            int b;
            {
                std::unique_lock<std::mutex> lk(mu);
                b = next_read;
                if (b >= nblocks || failed) return;
                ++next_read;
                cv.wait(lk, [&] { return failed || b < written + window; });        // its buffers are free again
                if (failed) return;
            }
            const size_t s = (size_t)(b % window);
            const long long n = block_len(b);
            ibuf[s].need((size_t)n); obuf[s].need((size_t)n + LIBBSC_HEADER_SIZE);
            bool ok = ibuf[s].p && obuf[s].p;
            for (long long got = 0; ok && got < n;) {
                const ssize_t r = pread(fd, ibuf[s].p + got, (size_t)(n - got), (off_t)((long long)b * block + got));
                if (r <= 0) ok = false; else got += r;
            }
            { std::lock_guard<std::mutex> lk(mu); if (ok) ready[(size_t)b] = 1; else failed = true; }
            cv.notify_all();
        }
    };
    long long out_bytes = 8;
    auto writer = [&] {
        for (int b = 0; b < nblocks; ++b) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || added > b; }); if (failed) return; }
            const int r = bscgpu_job_wait(job, b);
            bool ok = r >= 0;
            if (!ok) fprintf(stderr, "block %d: error %d\n", b, r);
            if (ok) {
                BlockHeader h; h.offset = (long long)b * block; h.recordSize = 1; h.sortingContexts = 1;
                ok = fwrite(&h, sizeof h, 1, fo) == 1 && fwrite(obuf[(size_t)(b % window)].p, 1, (size_t)r, fo) == (size_t)r;
                if (!ok) perror(argv[3]);
                out_bytes += (long long)sizeof h + r;
            }
            { std::lock_guard<std::mutex> lk(mu); if (ok) written = b + 1; else failed = true; }
            cv.notify_all();
            if (!ok) return;
        }
    };
    std::vector<std::thread> threads;
    const int nreaders = nblocks < 3 ? (nblocks > 0 ? nblocks : 1) : 3;
    for (int k = 0; k < nreaders; ++k) threads.emplace_back(reader);
    threads.emplace_back(writer);
    for (int b = 0; b < nblocks; ++b) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || ready[(size_t)b]; }); if (failed) break; }
        const size_t s = (size_t)(b % window);
        const int id = bscgpu_job_add(job, ibuf[s].p, obuf[s].p, (int)block_len(b), lzp ? lzpHash : 0, lzp ? lzpMin : 0, sorter, coder,
                                      LIBBSC_FEATURE_FASTMODE | LIBBSC_FEATURE_MULTITHREADING);
        { std::lock_guard<std::mutex> lk(mu); if (id == b) added = b + 1; else { fprintf(stderr, "bscgpu_job_add: %d\n", id); failed = true; } }
        cv.notify_all();
    }
    for (auto& t : threads) t.join();
    rc = failed ? -1 : 0;
    bscgpu_job_destroy(job);
    fclose(fi); fclose(fo);
    if (rc < 0) return 1;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s compressed %lld into %lld in %.3f seconds (%d block(s), %d GPU(s) x %d context(s) x %d in flight, %.1f MB/s).\n", argv[2], size, out_bytes, dt,
           nblocks, ndev, contexts, depth, dt > 0 ? size / 1e6 / dt : 0.0);
    return 0;
}
