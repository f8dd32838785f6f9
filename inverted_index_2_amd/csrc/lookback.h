// lookback.h — output offsets inside ONE launch: how many units (result ids, payload bytes) the workgroups before mine produce.
//
// Used by kernels whose workgroups each make a variable amount of output that has to land in workgroup order (k_and2_fused: the
// ids of a two-list AND; k_enc_stream: the payload bytes of a DV1 segment).  A workgroup publishes its own amount as soon as it
// knows it and later asks for the sum of all amounts before it; in between it does whatever does not need the offset (staging
// its output in LDS).  The records live in HBM, one 8-byte word each, {epoch : 24 | value : 40}: value and ready flag travel in
// one relaxed agent-scope store (no fence: the per-XCD L2s are not coherent with each other and a release would write one back),
// and the epoch is the launch's number, so the records are never cleared between launches (the host clears them when the 24
// bits wrap).
//   agg[g]          amount of workgroup g
//   grp[2 G]        amount of group G = 64 consecutive workgroups, published by the group's last workgroup once the others have
//   grp[2 G + 1]    amount of all groups up to and including G, published by the same workgroup when it knows its own prefix
// A workgroup reads the records of the groups before its own 64 at a time, nearest first, and stops at the first that already
// carries a prefix; then the members of its own group before it: two hops behind the slowest workgroup it depends on.
//
// Every wait is for a workgroup with a SMALLER index.  The hardware starts the workgroups of a launch in index order on every
// XCD, so the lowest unfinished workgroup is always running and waits for nobody — observed behaviour, not a HIP guarantee:
// every wait is therefore bounded (`spin` polls), a wait that runs out stores the epoch into *err and the caller's kernel
// leaves without writing output; the host then repeats the call on a path without inter-workgroup waits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dv1_device.h"
#include "internal.h"      // struct LookBack

namespace ii2 {

constexpr uint32_t LB_GROUP = 64;
constexpr uint32_t LB_SPIN = 1u << 21;                       // polls (each >= ~1 us): seconds in all
constexpr uint32_t LB_VALUE_BITS = 40;
constexpr unsigned long long LB_VALUE_MASK = (1ull << LB_VALUE_BITS) - 1ull;
constexpr uint32_t LB_EPOCH_MAX = (1u << (64u - LB_VALUE_BITS)) - 1u;

__device__ __forceinline__ unsigned long long lb_ld(const unsigned long long *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_st(unsigned long long *q, unsigned long long v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long lb_tag(const LookBack &lb) { return (unsigned long long)lb.epoch << LB_VALUE_BITS; }
__device__ __forceinline__ bool lb_ready(const LookBack &lb, unsigned long long rec) { return (uint32_t)(rec >> LB_VALUE_BITS) == lb.epoch; }
__device__ __forceinline__ uint32_t lb_limit(const LookBack &lb) { return lb.spin ? lb.spin : LB_SPIN; }

// sum over the wave of values < 2^40 (all lanes active): two 20-bit halves through the DPP scan - a 64-bit shuffle butterfly is
// twelve LDS round trips on the one wave every other wave of the workgroup is waiting for
__device__ __forceinline__ unsigned long long lb_wave_sum64(unsigned long long x) {
    const uint32_t lo = wave_sum((uint32_t)x & 0xFFFFFu), hi = wave_sum((uint32_t)(x >> 20) & 0xFFFFFu);
    return (unsigned long long)lo + ((unsigned long long)hi << 20);
}

// ONE lane of workgroup g: my amount is known
__device__ __forceinline__ void lb_publish(const LookBack &lb, uint32_t g, unsigned long long amount) { lb_st(&lb.agg[g], lb_tag(lb) | (amount & LB_VALUE_MASK)); }

__device__ __forceinline__ bool lb_is_leader(uint32_t g, uint32_t n_wg) { return g % LB_GROUP == LB_GROUP - 1u || g == n_wg - 1u; }

// ONE WAVE (all 64 lanes) of a group's last workgroup: the group's amount, once its other members have published theirs.
// Returns false when the wait ran out (nothing published).
__device__ __forceinline__ bool lb_group_publish(const LookBack &lb, uint32_t g, unsigned long long own) {
    const uint32_t l = (uint32_t)lane_id(), G = g / LB_GROUP, gi = g % LB_GROUP;
    const uint32_t limit = lb_limit(lb);
    uint32_t spins = 0;
    unsigned long long v = lb_tag(lb);
    bool have = l >= gi;
    for (;;) {
        if (!have) { v = lb_ld(&lb.agg[(size_t)G * LB_GROUP + l]); have = lb_ready(lb, v); }
        if (__ballot(!have) == 0ull) break;
        if (++spins > limit) return false;
        __builtin_amdgcn_s_sleep(16);
    }
    const unsigned long long tot = lb_wave_sum64(v & LB_VALUE_MASK) + own;
    if (l == 0) lb_st(&lb.grp[2 * (size_t)G], lb_tag(lb) | (tot & LB_VALUE_MASK));
    return true;
}

// ONE WAVE (all 64 lanes) of workgroup g: the amounts of all workgroups before g.  A group's last workgroup also publishes the
// group's inclusive prefix (`own` = its own amount).  Returns false when a wait ran out.
// Polls are uncached loads, one fabric transaction each, and a launch may have hundreds of thousands of workgroups: a record
// that has been seen ready is not read again, and the groups are looked at 32 at a time (the nearest one that carries a prefix
// is normally one or two groups back) - the first version read 64 + 2 x 64 records per poll and its polls alone were as much
// traffic as the payload of k_enc_stream.
constexpr uint32_t LB_WINDOW = 32;
__device__ __forceinline__ bool lb_prefix(const LookBack &lb, uint32_t g, uint32_t n_wg, unsigned long long own, unsigned long long *prefix,
                                          uint32_t *polls = nullptr, uint32_t *polls_members = nullptr) {
    const uint32_t l = (uint32_t)lane_id(), G = g / LB_GROUP, gi = g % LB_GROUP;
    const uint32_t limit = lb_limit(lb);
    if (lb.spin == 0xFFFFFFFFu && g == 1u) return false;
    // Three sets of records, ALL requested in the same round (an uncached load under a streaming kernel's traffic is ~3 us: three
    // rounds one after the other were most of a workgroup's life in k_enc_stream):
    //   a   members of my group before me;
    //   b   all members of the group right before mine - by its members' amounts (one hop behind them) rather than by the group's
    //       record (two: its last workgroup has to collect them first);
    //   ga / gp   amount and inclusive prefix of the LB_WINDOW groups before that, nearest first.
    unsigned long long a = lb_tag(lb), b = lb_tag(lb);
    bool have_a = l >= gi, have_b = G == 0u;
    int top = (int)G - 2;
    unsigned long long accg = 0ull;
    bool groups_done = top < 0;
    uint32_t spins = 0;
    for (;;) {                                  // one pass per window of groups (normally one)
        const int j = top - (int)l;
        const bool inr = !groups_done && j >= 0 && l < LB_WINDOW;
        unsigned long long ga = lb_tag(lb), gp = 0ull;
        bool have_ga = !inr, have_gp = false;
        bool next = false;
        for (;;) {
            if (polls) ++*polls;                // (diagnostics)
            if (!have_a) a = lb_ld(&lb.agg[(size_t)G * LB_GROUP + l]);
            if (!have_b) b = lb_ld(&lb.agg[(size_t)(G - 1u) * LB_GROUP + l]);
            if (inr && !have_gp) gp = lb_ld(&lb.grp[2 * (size_t)j + 1]);
            if (!have_ga) ga = lb_ld(&lb.grp[2 * (size_t)j]);
            have_a = have_a || lb_ready(lb, a);
            have_b = have_b || lb_ready(lb, b);
            have_gp = inr && (have_gp || lb_ready(lb, gp));
            have_ga = have_ga || lb_ready(lb, ga);
            const bool mem_ok = __ballot(!have_a || !have_b) == 0ull;
            if (polls_members && !mem_ok) ++*polls_members;      // (diagnostics)
            if (!groups_done) {
                const unsigned long long gav = __ballot(have_ga);
                const unsigned long long gpv = __ballot(have_gp);
                if (gpv != 0ull) {              // nearest group that already carries its prefix: the groups between it and me by their own amounts
                    const uint32_t d = (uint32_t)__ffsll((long long)gpv) - 1u;
                    const unsigned long long need = (1ull << d) - 1ull;
                    if ((gav & need) == need) {
                        const unsigned long long part = lb_wave_sum64(l < d ? (ga & LB_VALUE_MASK) : 0ull);
                        const unsigned long long gl = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gp >> 32), (int)d) << 32) |
                                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gp, (int)d);
                        accg += part + (gl & LB_VALUE_MASK);
                        groups_done = true;
                    }
                } else if (gav == ~0ull) {      // no prefix among them, but all their amounts: on to the next window
                    accg += lb_wave_sum64(inr ? (ga & LB_VALUE_MASK) : 0ull);
                    next = true;
                }
            }
            if (next || (groups_done && mem_ok)) break;
            if (++spins > limit) return false;
            __builtin_amdgcn_s_sleep(16);
        }
        if (!next) break;
        top -= (int)LB_WINDOW;
        if (top < 0) groups_done = true;
        if (groups_done && __ballot(!have_a || !have_b) == 0ull) break;
        spins = 0;
    }
    const unsigned long long mem = lb_wave_sum64(a & LB_VALUE_MASK) + lb_wave_sum64(b & LB_VALUE_MASK);
    *prefix = accg + mem;
    if (l == 0 && lb_is_leader(g, n_wg)) lb_st(&lb.grp[2 * (size_t)G + 1], lb_tag(lb) | ((accg + mem + own) & LB_VALUE_MASK));
    return true;
}

// ONE lane: a wait ran out
__device__ __forceinline__ void lb_fail(const LookBack &lb) { lb_st(lb.err, (unsigned long long)lb.epoch); }
__device__ __forceinline__ bool lb_failed(const LookBack &lb) { return lb_ld(lb.err) == (unsigned long long)lb.epoch; }

}  // namespace ii2
