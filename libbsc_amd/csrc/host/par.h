// par.h — run n independent tasks, one per thread where threads can be had (internal).
//
// Thread creation can fail (pid / thread limits of a container: std::system_error out of std::thread's constructor).
// Behind an extern "C" API and inside pipe workers that must never escape: whatever could not get its own thread runs
// on the calling thread instead, so the result is the same and only the parallelism degrades.  The last task always
// runs on the caller (one thread fewer to create).
#pragma once
#include <cstddef>
#include <system_error>
#include <thread>
#include <vector>

namespace bschost {

template <class F>
inline void run_tasks(int n, F&& fn)
{
    if (n <= 0) return;
    std::vector<std::thread> pool;
    pool.reserve((size_t)n);
    int next = 0;
    for (; next < n - 1; ++next) {
        try { pool.emplace_back([&fn, next] { fn(next); }); }
        catch (const std::system_error&) { break; }         // no more threads: the rest runs here
    }
    for (int b = next; b < n; ++b) fn(b);
    for (auto& t : pool) t.join();
}

// Block-sized scratch that lives for one block (the LZP output, the staging area of the parallel LZP encoder).  A fresh allocation of that
// size is an mmap per block: 16 K page faults and 64 MiB of page zeroing each time, under the process's mmap lock with thirty threads around.
// A few idle buffers are kept instead (lzp.cpp; at most 8 and 1 GiB, BSC_HOST_BUFFER_CACHE=0: none).  Plain malloc / free underneath.
void* bigbuf_get(size_t bytes);          // nullptr: out of memory
void  bigbuf_put(void* p);               // nullptr is fine

}  // namespace bschost
