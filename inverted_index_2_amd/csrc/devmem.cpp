// devmem.cpp — device memory for SEGMENTS: a size-class cache in front of hipMalloc / hipFree.
//
// A Shard.Merge writes a new segment and drops the segments it replaces (reference shard.go:163-225, segments.go:48-75); the
// strong-scaling exchange receives N - 1 segments per step and frees them.  Each segment is six device arrays: through
// hipMalloc / hipFree that is a dozen driver calls per segment, and a hipFree of a large array waits for the whole device and
// unmaps it — milliseconds, and much more on a box whose allocator is slow (one 64-segment merge step of the bench measured
// 31 ms on one box and 101 ms on another with the same kernels).  Freed arrays are kept here, by size class, and handed out
// again: a steady stream of merges allocates nothing.
//
// Contract (unchanged): a segment is freed only when no call that reads it is still running — the library's calls are
// synchronous, so that is "after the call returned"; ii2_intersect_async users wait for their query first.  hipFree used to
// enforce this by waiting for the device; the cache does not wait — so the library's OWN early exits (a launch that failed
// with earlier kernels of the call still enqueued) wait for their stream before they hand arrays back (seg_release of a
// segment that was never finished, the error paths of ii2_seg_select_aligned and ii2_merge_small).
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

#include "internal.h"

namespace ii2 {

namespace {
struct DevCache {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;            // arrays handed out: their size class
    std::multimap<size_t, void *> idle;                  // arrays waiting for reuse, by size class
    size_t idle_bytes = 0;
    size_t cap = 0;                                      // 0: not initialised
    int users = 0;                                       // contexts alive (the cache is emptied with the last one)
};
DevCache g_cache;        // one cache for the process: a pointer knows its device, and a class is only reused on the device it came from

struct Key { int device; size_t cls; };

size_t size_class(size_t bytes) {
    if (bytes < 4096) return 4096;
    size_t p = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));        // largest power of two <= bytes
    const size_t g = p / 8;                                                            // classes are an eighth of an octave apart
    return (bytes + g - 1) / g * g;
}
size_t cache_cap() {
    const char *e = std::getenv("II2_DEVMEM_CACHE_MB");
    if (e && *e) return (size_t)std::strtoull(e, nullptr, 10) << 20;
    return (size_t)16 << 30;
}
// per-device separation: the class is tagged with the device in its low bits (classes are multiples of 512)
size_t tagged(size_t cls, int device) { return cls + (size_t)(device & 0xFF); }
}  // namespace

hipError_t dm_alloc(void **p, size_t bytes) {
    int device = 0;
    (void)hipGetDevice(&device);
    const size_t cls = size_class(bytes ? bytes : 16);
    const size_t key = tagged(cls, device);
    {
        std::lock_guard<std::mutex> g(g_cache.mu);
        auto it = g_cache.idle.find(key);
        if (it != g_cache.idle.end()) {
            *p = it->second;
            g_cache.idle.erase(it);
            g_cache.idle_bytes -= cls;
            g_cache.live[*p] = key;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {                               // make room: give the idle arrays back and try once more
        (void)hipGetLastError();
        dm_trim(0);
        e = hipMalloc(p, cls);
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    }
    std::lock_guard<std::mutex> g(g_cache.mu);
    g_cache.live[*p] = key;
    return hipSuccess;
}

// hipMalloc for everything that is NOT a segment array (workspaces, staging pools, dictionaries, user buffers): when the driver
// has no room, the idle segment arrays in the cache are the first thing to give back
hipError_t dm_malloc_retry(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dm_trim(0);
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) (void)hipGetLastError();      // (the runtime keeps a failed call's error for the next hipGetLastError: a later launch check would report it)
    }
    return e;
}

void dm_free(void *p) {
    if (!p) return;
    size_t key = 0;
    {
        std::lock_guard<std::mutex> g(g_cache.mu);
        auto it = g_cache.live.find(p);
        if (it != g_cache.live.end()) {
            key = it->second;
            g_cache.live.erase(it);
            if (!g_cache.cap) g_cache.cap = cache_cap();
            const size_t cls = key & ~(size_t)0xFF;
            if (g_cache.idle_bytes + cls <= g_cache.cap) {
                g_cache.idle.emplace(key, p);
                g_cache.idle_bytes += cls;
                return;
            }
        }
    }
    (void)hipFree(p);                                    // not from the cache, or the cache is full
}

// gives idle arrays back to the driver until at most `keep_bytes` are cached
void dm_trim(size_t keep_bytes) {
    std::multimap<size_t, void *> out;
    {
        std::lock_guard<std::mutex> g(g_cache.mu);
        while (g_cache.idle_bytes > keep_bytes && !g_cache.idle.empty()) {
            auto it = std::prev(g_cache.idle.end());      // the largest first
            g_cache.idle_bytes -= it->first & ~(size_t)0xFF;
            out.insert(*it);
            g_cache.idle.erase(it);
        }
    }
    for (auto &kv : out) (void)hipFree(kv.second);
}

void dm_stats(uint64_t *live_bytes, uint64_t *idle_bytes) {
    std::lock_guard<std::mutex> g(g_cache.mu);
    if (live_bytes) {
        uint64_t t = 0;
        for (auto &kv : g_cache.live) t += kv.second & ~(size_t)0xFF;
        *live_bytes = t;
    }
    if (idle_bytes) *idle_bytes = g_cache.idle_bytes;
}

void dm_user(int delta) {
    bool last = false;
    {
        std::lock_guard<std::mutex> g(g_cache.mu);
        g_cache.users += delta;
        last = g_cache.users <= 0;
    }
    if (last && delta < 0) dm_trim(0);
}

}  // namespace ii2
