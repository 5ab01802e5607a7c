// lzp.cpp — LZP preprocessor, encoder side.  Host code: the predictor is one sequential hash chain per
// chunk, at most 8 chunks per block, so there is nothing here for the GPU.
//
// Stream format and match decisions are fixed by the reference (libbsc/lzp/lzp.cpp): the decoder mirrors the
// encoder's hash table, so *which* positions become matches is part of the format in practice — a different but valid
// choice would still decode, but would not be byte-identical to libbsc's output.  What has to be reproduced:
//   * context hash of the previous 4 bytes c (big-endian):  ((c >> 15) ^ c ^ (c >> 3)) & (2^hashSize - 1); the table
//     holds the last position seen per slot, 0 = empty, positions 0..3 are never entered           (lzp.cpp:69, :151);
//   * token grammar: literal | 0xF2 255 (a literal 0xF2 where the slot was occupied) |
//     0xF2 {254}* r (match of minLen + 254*k + r bytes)                                             (lzp.cpp:118, :153);
//   * six encoder variants picked by (hashSize, minLen) (lzp.cpp:537-557) that differ in how a candidate is probed,
//     where the main phase stops, and — for minLen > 16 or hashSize > 17 — a "do not retry before" heuristic.
// The reference spells the first five out as five hand-unrolled functions; here they are one scanner parametrised by a
// probe width, a main-phase guard and a verify flag, plus a separate scanner for wide tables.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include "par.h"
#include <vector>
#include <memory>
#include <mutex>
#include <new>

#include "qlfc.h"
#include "lzp.h"

namespace bschost {

namespace {

constexpr uint8_t FLAG = 0xF2;

inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t ctx_at(const uint8_t* p) { return __builtin_bswap32(ld32(p - 4)); }        // previous four bytes, newest lowest
inline uint32_t slot_of(uint32_t c, uint32_t mask) { return ((c >> 15) ^ c ^ (c >> 3)) & mask; }

// Match length counted in 8-byte strides from `len` on, while the compared window starts before `limit`
// (lzp.cpp:103-110): a stride that starts before the limit is compared in full.
inline int64_t extend8(const uint8_t* p, const uint8_t* ref, int64_t len, const uint8_t* limit)
{
    while (p + len < limit) {
        const uint64_t x = ld64(p + len) ^ ld64(ref + len);
        if (x) return len + (__builtin_ctzll(x) >> 3);
        len += 8;
    }
    return len;
}

// 0xF2 then the excess over minLen in base 254 (lzp.cpp:114).  False when the output budget is exhausted — the block
// is then reported as not compressible whatever else happens, so the caller stops.
inline bool put_match(uint8_t*& o, const uint8_t* eob, int64_t excess)
{
    *o++ = FLAG;
    while (excess >= 254) { excess -= 254; *o++ = 254; if (o >= eob) return false; }
    *o++ = (uint8_t)excess;
    return true;
}

// Last phase of every variant (lzp.cpp:136-150): no matching any more, only the escape after a literal 0xF2 whose
// slot was occupied.
inline int finish_tail(const uint8_t* base, const uint8_t* p, const uint8_t* end, uint8_t* out, uint8_t* o, const uint8_t* eob,
                       int32_t* table, uint32_t mask)
{
    if (p < end && o < eob) {
        uint32_t c = ctx_at(p);
        while (p < end && o < eob) {
            const uint32_t s = slot_of(c, mask);
            const int32_t seen = table[s]; table[s] = (int32_t)(p - base);
            const uint8_t b = *p++; *o++ = b; c = (c << 8) | b;
            if (b == FLAG && seen > 0) *o++ = 255;
        }
    }
    return o >= eob ? NOT_COMPRESSIBLE : (int)(o - out);
}

// Narrow tables (hashSize <= 17).  W = probe word (4 or 8 bytes): a candidate is accepted when the words at offsets 0
// and minLen - W agree with the predicted position — that covers all minLen bytes for minLen <= 2W (lzp.cpp:72, :182,
// :292).  VERIFY (minLen > 16): the two words do not cover the match, so its real length is measured from byte 8 and a
// short one is turned into a literal; `retry_after` then suppresses probing for groups that start at or before the
// point where the comparison failed (lzp.cpp:336-337, :394).
// The main phase re-checks its bounds once per group of 4 positions, exactly like the reference loop (lzp.cpp:63), so
// up to three positions past `limit` are still probed.
template <int W, bool VERIFY>
int scan_narrow(const uint8_t* in, const uint8_t* end, uint8_t* out, uint8_t* out_end, int32_t* table, uint32_t mask,
                int minLen, int guard)
{
    const uint8_t* const base = in;
    const uint8_t* const limit = end - guard;
    const uint8_t* const eob = out_end - 8;
    const uint8_t* p = in + 4;
    uint8_t* o = out;
    memcpy(o, in, 4); o += 4;
    const uint8_t* retry_after = in;
    const int far = minLen - W;

    auto same = [&](const uint8_t* q, const uint8_t* ref) -> bool {
        if (W == 4) return ld32(q + far) == ld32(ref + far) && ld32(q) == ld32(ref);
        return ld64(q + far) == ld64(ref + far) && ld64(q) == ld64(ref);
    };

    while (p < limit && o < eob) {
        const bool probing = !VERIFY || p > retry_after;
        int j = 0, kind = 0;                     // kind: 0 nothing in this group, 1 match at p + j, 2 escaped literal at p + j
        int32_t seen = 0;
        for (; j < 4; ++j) {
            const uint8_t* q = p + j;
            const uint32_t s = slot_of(ctx_at(q), mask);
            seen = table[s]; table[s] = (int32_t)(q - base);
            if (seen > 0) {
                if (probing && same(q, base + seen)) { kind = 1; break; }
                if (*q == FLAG) { kind = 2; break; }
            }
        }
        if (kind == 0) { memcpy(o, p, 4); o += 4; p += 4; continue; }
        memcpy(o, p, 4); o += j; p += j;          // literals in front of the event (4-byte store: slack is inside the 8-byte reserve)
        if (kind == 2) { *o++ = FLAG; *o++ = 255; ++p; continue; }

        const uint8_t* ref = base + seen;
        const int64_t len = extend8(p, ref, VERIFY ? 8 : minLen, limit);
        if (VERIFY && len < minLen) {
            retry_after = p + len;
            const uint8_t b = *p++; *o++ = b;
            if (b == FLAG) *o++ = 255;
            continue;
        }
        p += len;
        if (!put_match(o, eob, len - minLen)) return NOT_COMPRESSIBLE;
    }
    return finish_tail(base, p, end, out, o, eob, table, mask);
}

// Wide tables (hashSize > 17): one position per step with a rolling context; candidates are probed with two 4-byte
// words, measured in 4-byte strides plus a 2-byte and a 1-byte step, and rejected early when the word at the last
// failure point disagrees (lzp.cpp:447-519).  A literal 0xF2 is escaped whenever its slot was occupied.
int scan_wide(const uint8_t* in, const uint8_t* end, uint8_t* out, uint8_t* out_end, int32_t* table, uint32_t mask, int minLen)
{
    const uint8_t* const base = in;
    const uint8_t* const limit = end - minLen - 32;
    const uint8_t* const eob = out_end - 8;
    const uint8_t* p = in + 4;
    uint8_t* o = out;
    memcpy(o, in, 4); o += 4;
    const uint8_t* fail_at = in;
    uint32_t c = (p < limit) ? ctx_at(p) : 0;

    while (p < limit && o < eob) {
        const uint32_t s = slot_of(c, mask);
        const int32_t seen = table[s]; table[s] = (int32_t)(p - base);
        if (seen <= 0) { const uint8_t b = *p++; *o++ = b; c = (c << 8) | b; continue; }

        const uint8_t* ref = base + seen;
        bool take = ld32(p + minLen - 4) == ld32(ref + minLen - 4) && ld32(p) == ld32(ref);
        if (take && fail_at > p && ld32(fail_at) != ld32(ref + (fail_at - p))) take = false;
        int64_t len = 0;
        if (take) {
            len = 4;
            while (p + len < limit && ld32(p + len) == ld32(ref + len)) len += 4;
            if (len < minLen) { if (fail_at < p + len) fail_at = p + len; take = false; }
        }
        if (!take) {
            const uint8_t b = *p++; *o++ = b; c = (c << 8) | b;
            if (b == FLAG) *o++ = 255;
            continue;
        }
        uint16_t a2, b2; memcpy(&a2, p + len, 2); memcpy(&b2, ref + len, 2);
        if (a2 == b2) len += 2;
        if (p[len] == ref[len]) len += 1;
        p += len; c = ctx_at(p);
        if (!put_match(o, eob, len - minLen)) return NOT_COMPRESSIBLE;
    }
    return finish_tail(base, p, end, out, o, eob, table, mask);
}

// Where the main phase of the narrow scanner stops, in bytes before the end of the chunk (lzp.cpp:56, :166, :276, :330).
int narrow_guard(int minLen)
{
    if (minLen == 4) return 4 + 32;
    if (minLen <= 8) return 8 + 32;
    if (minLen <= 16) return 16 + 32;
    return minLen + 32;
}

}  // namespace

// ---- bigbuf: see par.h ----------------------------------------------------------------------------------------------------------------
namespace {
struct BigBuf { void* p; size_t cap; bool used; };
struct BigBufCache {
    std::mutex mu;
    std::vector<BigBuf> all;                     // buffers handed out (used) and idle ones
    const bool on = [] { const char* e = getenv("BSC_HOST_BUFFER_CACHE"); return e ? atoi(e) != 0 : true; }();
    ~BigBufCache() { for (auto& b : all) if (!b.used) free(b.p); }      // (a buffer still handed out at exit belongs to its holder)
};
BigBufCache& bigbufs() { static BigBufCache c; return c; }
constexpr size_t BIG_MIN = (size_t)1 << 20, BIG_IDLE_MAX = 8, BIG_IDLE_BYTES = (size_t)1 << 30;
}  // namespace

void* bigbuf_get(size_t bytes)
{
    BigBufCache& C = bigbufs();
    if (!C.on || bytes < BIG_MIN) return malloc(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(C.mu);
        BigBuf* best = nullptr;
        for (auto& b : C.all)
            if (!b.used && b.cap >= bytes && b.cap / 4 <= bytes && (!best || b.cap < best->cap)) best = &b;     // (no 1 GiB buffer for a 2 MiB request)
        if (best) { best->used = true; return best->p; }
    }
    const size_t cap = bytes + bytes / 16 + 4096;                     // the next block is rarely exactly this long
    void* p = malloc(cap);
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(C.mu);
    C.all.push_back(BigBuf{p, cap, true});
    return p;
}

void bigbuf_put(void* p)
{
    if (!p) return;
    BigBufCache& C = bigbufs();
    {
        std::lock_guard<std::mutex> lk(C.mu);
        size_t idle = 0, idle_bytes = 0; long at = -1;
        for (size_t i = 0; i < C.all.size(); ++i) {
            if (C.all[i].p == p) at = (long)i;
            else if (!C.all[i].used) { ++idle; idle_bytes += C.all[i].cap; }
        }
        if (at >= 0) {
            if (idle < BIG_IDLE_MAX && idle_bytes + C.all[(size_t)at].cap <= BIG_IDLE_BYTES) { C.all[(size_t)at].used = false; return; }
            C.all.erase(C.all.begin() + at);                           // enough idle ones already: this one goes back to the system
        }
    }
    free(p);                                                            // (also: a small request that never entered the cache)
}

int lzp_num_chunks(int n)                                   // lzp.cpp:44-51
{
    if (n < 256 * 1024) return 1;
    if (n < 4 * 1024 * 1024) return 2;
    if (n < 16 * 1024 * 1024) return 4;
    return 8;
}

int lzp_encode_chunk(const uint8_t* in, int n, uint8_t* out, int out_cap, int hashSize, int minLen)   // lzp.cpp:529
{
    if ((int64_t)n - minLen < 32) return NOT_COMPRESSIBLE;
    if (out_cap < 16) return NOT_COMPRESSIBLE;              // the reference would run off its buffer here; cannot compress anyway
    int32_t* table = (int32_t*)calloc((size_t)1 << hashSize, sizeof(int32_t));
    if (!table) return NOT_ENOUGH_MEMORY;
    const uint32_t mask = ((uint32_t)1 << hashSize) - 1;
    int r;
    if (hashSize <= 17) {
        const int guard = narrow_guard(minLen);
        if (minLen > 16)     r = scan_narrow<8, true >(in, in + n, out, out + out_cap, table, mask, minLen, guard);
        else if (minLen >= 8) r = scan_narrow<8, false>(in, in + n, out, out + out_cap, table, mask, minLen, guard);
        else                  r = scan_narrow<4, false>(in, in + n, out, out + out_cap, table, mask, minLen, guard);
    } else {
        r = scan_wide(in, in + n, out, out + out_cap, table, mask, minLen);
    }
    free(table);
    return r;
}

static inline void put_le32(uint8_t* p, int v) { memcpy(p, &v, 4); }

// Block framing (lzp.cpp:687-811): [nChunks] then, for more than one chunk, {i32 rawSize, i32 codedSize} per chunk and
// the chunk payloads; a chunk that does not shrink is kept raw (codedSize == rawSize).
int lzp_compress(const uint8_t* in, uint8_t* out, int n, int hashSize, int minLen, int features)
{
    const int nc = lzp_num_chunks(n);
    if (nc == 1) {                                          // lzp.cpp:689-695
        const int r = lzp_encode_chunk(in, n, out + 1, n - 2, hashSize, minLen);
        if (r < 0) return r;
        out[0] = 1;
        return r + 1;
    }
    const int chunk = n / nc;
    if (features & 2 /* LIBBSC_FEATURE_MULTITHREADING */) {
        // concurrent chunks, each with a budget of its own size (lzp.cpp:736-790)
        std::unique_ptr<uint8_t, void (*)(void*)> tmp_buf((uint8_t*)bigbuf_get((size_t)n), bigbuf_put);   // (not a zeroed vector, not a fresh mapping per block)
        if (!tmp_buf) return NOT_ENOUGH_MEMORY;
        uint8_t* const tmp = tmp_buf.get();
        int res[8];
        run_tasks(nc, [&](int b) {
            const int st = b * chunk, sz = (b != nc - 1) ? chunk : n - st;
            const int r = lzp_encode_chunk(in + st, sz, tmp + st, sz, hashSize, minLen);
            res[b] = (r < 0) ? sz : r;
        });
        int64_t total = 1 + 8 * nc;
        for (int b = 0; b < nc; ++b) total += res[b];
        if (total >= n) return NOT_COMPRESSIBLE;
        out[0] = (uint8_t)nc;
        int optr = 1 + 8 * nc;
        for (int b = 0; b < nc; ++b) {
            const int st = b * chunk, sz = (b != nc - 1) ? chunk : n - st;
            put_le32(out + 1 + 8 * b, sz); put_le32(out + 1 + 8 * b + 4, res[b]);
            memcpy(out + optr, (res[b] != sz) ? tmp + st : in + st, (size_t)res[b]);
            optr += res[b];
        }
        return optr;
    }
    // one chunk after the other, each limited by what is left of the n-byte output (lzp.cpp:697-731)
    out[0] = (uint8_t)nc;
    int optr = 1 + 8 * nc;
    for (int b = 0; b < nc; ++b) {
        const int st = b * chunk, sz = (b != nc - 1) ? chunk : n - st;
        int cap = sz; if (cap > n - optr) cap = n - optr;
        int r = lzp_encode_chunk(in + st, sz, out + optr, cap, hashSize, minLen);
        if (r < 0) {
            if (optr + sz >= n) return NOT_COMPRESSIBLE;
            r = sz; memcpy(out + optr, in + st, (size_t)sz);
        }
        put_le32(out + 1 + 8 * b, sz); put_le32(out + 1 + 8 * b + 4, r);
        optr += r;
    }
    return optr;
}

}  // namespace bschost
