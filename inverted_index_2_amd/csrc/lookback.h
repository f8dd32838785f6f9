// lookback.h — single-pass ordered output offsets across workgroup tiles (decoupled
// look-back).  Tile t publishes its survivor count as one 8-byte descriptor
// {epoch:22 | status:2 | value:40}; a tile sums its predecessors' descriptors until it
// meets an INCLUSIVE one.  Each descriptor is one naturally aligned 8-byte agent-scope
// atomic, so a reader sees either the old or the new word (never a torn one) and nothing
// else is handed over between workgroups — no release/acquire pairing is needed
// (MI355X_MICROARCH.md "Valid forms": the 8-byte {data, tag} granule).
// Descriptors of earlier launches carry another epoch and read as "not yet published".
// Forward progress is the caller's contract: every tile < t must be held by a workgroup
// that is resident and working on it (the posting kernels run persistent grids no larger
// than what is co-resident, walking tiles in ascending order), so the spin terminates.
#pragma once
#include "dv1_device.h"

namespace ii2 {

constexpr unsigned long long LB_VALUE_MASK = (1ull << 40) - 1ull;
constexpr uint32_t LB_AGG = 1u, LB_INCL = 2u;

__device__ __forceinline__ unsigned long long lb_pack(uint32_t epoch, uint32_t status, unsigned long long value) {
    return ((unsigned long long)(epoch & 0x3FFFFFu) << 42) | ((unsigned long long)status << 40) | (value & LB_VALUE_MASK);
}
__device__ __forceinline__ uint32_t lb_status(unsigned long long d, uint32_t epoch) {
    return (uint32_t)(d >> 42) == (epoch & 0x3FFFFFu) ? (uint32_t)((d >> 40) & 3u) : 0u;
}

// Called by ALL lanes of ONE wave of the tile (wave-uniform arguments).  Publishes this
// tile's count and returns the exclusive prefix (sum of the counts of tiles < tile).
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long *desc, uint32_t tile, uint32_t count,
                                                                 uint32_t epoch) {
    const int l = lane_id();
    if (tile == 0) {
        if (l == 0) __hip_atomic_store(&desc[0], lb_pack(epoch, LB_INCL, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0ull;
    }
    if (l == 0) __hip_atomic_store(&desc[tile], lb_pack(epoch, LB_AGG, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long base = 0ull;
    long long idx = (long long)tile - 1;
    for (;;) {
        const long long i = idx - l;
        unsigned long long d = 0ull;
        uint32_t st;
        for (;;) {
            if (i >= 0) {
                d = __hip_atomic_load(&desc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st = lb_status(d, epoch);
            } else {
                st = LB_INCL; d = 0ull;      // before tile 0: inclusive prefix 0
            }
            if (__ballot(st == 0u) == 0ull) break;
            __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long incl_mask = __ballot(st == LB_INCL);
        const uint32_t v32 = (uint32_t)(d & 0xFFFFFFFFull);   // aggregates fit 32 bits
        if (incl_mask) {
            const int first = __ffsll((long long)incl_mask) - 1;          // nearest predecessor with a full prefix
            const uint32_t part = wave_sum(l < first ? v32 : 0u);
            const uint32_t lo = wave_bcast((uint32_t)(d & 0xFFFFFFFFull), first);
            const uint32_t hi = wave_bcast((uint32_t)((d & LB_VALUE_MASK) >> 32), first);
            base += (unsigned long long)part + (((unsigned long long)hi << 32) | lo);
            break;
        }
        base += (unsigned long long)wave_sum(v32);
        idx -= 64;
    }
    if (l == 0)
        __hip_atomic_store(&desc[tile], lb_pack(epoch, LB_INCL, base + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return base;
}

}  // namespace ii2
