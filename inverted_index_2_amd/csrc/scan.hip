// scan.hip — device-wide exclusive prefix sums for the planners' offset tables (gfx950, wave64).
// Hand-written reduce-then-scan, three launches, no library:
//   k_scan_reduce   one workgroup per tile of SCAN_TILE elements -> the tile's sum
//   k_scan_tiles    one workgroup turns the tile sums into exclusive tile bases (in place)
//   k_scan_apply    every workgroup rescans its tile from its base and writes the result
// (an input of one tile - up to 4096 elements - takes k_scan_apply alone)
// Every thread reads all of its inputs before it writes, so `out` may be `in` (same element type).
#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_PER_THREAD = 16;
constexpr uint32_t SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;

__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t x) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = (uint64_t)__shfl_up((unsigned long long)x, d, 64);
        if (l >= d) x += y;
    }
    return x;
}

// exclusive scan of one u64 per thread over a 256-thread workgroup; *tot = the workgroup's sum
__device__ __forceinline__ uint64_t wg_excl_scan64(uint64_t v, uint64_t *wsum, uint64_t *tot) {
    const int l = lane_id(), wv = (int)threadIdx.x >> 6;
    const uint64_t incl = wave_incl_scan64(v);
    __syncthreads();
    if (l == 63) wsum[wv] = incl;
    __syncthreads();
    uint64_t pre = 0, t = 0;
    for (int w = 0; w < (int)(SCAN_THREADS / 64); w++) { if (w < wv) pre += wsum[w]; t += wsum[w]; }
    *tot = t;
    return pre + incl - v;
}

template <class TI>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const TI *__restrict__ in, uint64_t n, uint64_t *__restrict__ part) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_PER_THREAD; j++)
        if (base + j < n) s += (uint64_t)in[base + j];
    uint64_t tot;
    (void)wg_excl_scan64(s, wsum, &tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(uint64_t *__restrict__ part, uint64_t n_tiles) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    uint64_t carry = 0;
    for (uint64_t i0 = 0; i0 < n_tiles; i0 += SCAN_THREADS) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t v = i < n_tiles ? part[i] : 0;
        uint64_t tot;
        const uint64_t ex = wg_excl_scan64(v, wsum, &tot);
        if (i < n_tiles) part[i] = carry + ex;
        carry += tot;
    }
}

template <class TI, class TO>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const TI *in, TO *out, uint64_t n, const uint64_t *__restrict__ part,
                                                              const uint64_t *__restrict__ guard, uint64_t guard_max) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    if (guard && *guard > guard_max) return;        // all-or-nothing calls: the result is only written when it is wanted
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t v[SCAN_PER_THREAD];
    uint64_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_PER_THREAD; j++) {
        v[j] = base + j < n ? (uint64_t)in[base + j] : 0;
        s += v[j];
    }
    uint64_t tot;
    uint64_t run = (part ? part[blockIdx.x] : 0ull) + wg_excl_scan64(s, wsum, &tot);      // (part == nullptr: the input is one tile)
#pragma unroll
    for (uint32_t j = 0; j < SCAN_PER_THREAD; j++) {
        if (base + j < n) out[base + j] = (TO)run;
        run += v[j];
    }
}

// ---- three scans of equal length in the launches of one (blockIdx.y = which array): the merge's plan scans its terms' weights
// and posting counts (to u64) and their tile counts (to u32) one after the other - nine dependent launches of ~6 us each on a
// merge of any size, a third of the whole call for the ones between a few thousand and a million postings
__global__ __launch_bounds__(SCAN_THREADS) void k_scan3_reduce(const uint32_t *__restrict__ in0, const uint32_t *__restrict__ in1,
                                                               const uint32_t *__restrict__ in2, uint64_t n, uint64_t *__restrict__ part, uint64_t tiles) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    const uint32_t *in = blockIdx.y == 0 ? in0 : blockIdx.y == 1 ? in1 : in2;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_PER_THREAD; j++)
        if (base + j < n) s += (uint64_t)in[base + j];
    uint64_t tot;
    (void)wg_excl_scan64(s, wsum, &tot);
    if (threadIdx.x == 0) part[(uint64_t)blockIdx.y * tiles + blockIdx.x] = tot;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan3_tiles(uint64_t *__restrict__ part3, uint64_t n_tiles) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    uint64_t *part = part3 + (uint64_t)blockIdx.x * n_tiles;
    uint64_t carry = 0;
    for (uint64_t i0 = 0; i0 < n_tiles; i0 += SCAN_THREADS) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t v = i < n_tiles ? part[i] : 0;
        uint64_t tot;
        const uint64_t ex = wg_excl_scan64(v, wsum, &tot);
        if (i < n_tiles) part[i] = carry + ex;
        carry += tot;
    }
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan3_apply(const uint32_t *in0, uint64_t *out0, const uint32_t *in1, uint64_t *out1, const uint32_t *in2,
                                                              uint32_t *out2, uint64_t n, const uint64_t *__restrict__ part, uint64_t tiles) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    const uint32_t y = blockIdx.y;
    const uint32_t *in = y == 0 ? in0 : y == 1 ? in1 : in2;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t v[SCAN_PER_THREAD];
    uint64_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_PER_THREAD; j++) {
        v[j] = base + j < n ? (uint64_t)in[base + j] : 0;
        s += v[j];
    }
    uint64_t tot;
    uint64_t run = (part ? part[(uint64_t)y * tiles + blockIdx.x] : 0ull) + wg_excl_scan64(s, wsum, &tot);
#pragma unroll
    for (uint32_t j = 0; j < SCAN_PER_THREAD; j++) {
        if (base + j < n) {
            if (y == 0) out0[base + j] = run;
            else if (y == 1) out1[base + j] = run;
            else out2[base + j] = (uint32_t)run;
        }
        run += v[j];
    }
}

size_t scan_temp_bytes(size_t n) {
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE + 1;
    return (tiles * sizeof(uint64_t) + 255) & ~(size_t)255;
}

template <class TI, class TO>
static hipError_t scan_excl(void *tmp, size_t tmp_bytes, const TI *in, TO *out, size_t n, hipStream_t s, const uint64_t *guard = nullptr,
                            uint64_t guard_max = 0) {
    if (n == 0) return hipSuccess;
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tmp_bytes < tiles * sizeof(uint64_t)) return hipErrorInvalidValue;
    uint64_t *part = (uint64_t *)tmp;
    if (tiles == 1) {       // one launch instead of three: a dependent kernel costs ~5 us whatever it does, and the planners scan a lot of short arrays
        hipLaunchKernelGGL((k_scan_apply<TI, TO>), dim3(1), dim3(SCAN_THREADS), 0, s, in, out, (uint64_t)n, (const uint64_t *)nullptr, guard, guard_max);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_scan_reduce<TI>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, (uint64_t)n, part);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_THREADS), 0, s, part, (uint64_t)tiles);
    hipLaunchKernelGGL((k_scan_apply<TI, TO>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, in, out, (uint64_t)n, (const uint64_t *)part, guard, guard_max);
    return hipGetLastError();
}

hipError_t scan_excl_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    return scan_excl<uint32_t, uint32_t>(tmp, tmp_bytes, in, out, n, s);
}
hipError_t scan_excl_u32_to_u64(void *tmp, size_t tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t s) {
    return scan_excl<uint32_t, uint64_t>(tmp, tmp_bytes, in, out, n, s);
}
// writes `out` only while *guard <= guard_max (read on the device when the scan runs)
hipError_t scan_excl_u32_to_u64_guarded(void *tmp, size_t tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, const uint64_t *guard,
                                        uint64_t guard_max, hipStream_t s) {
    return scan_excl<uint32_t, uint64_t>(tmp, tmp_bytes, in, out, n, s, guard, guard_max);
}
hipError_t scan_excl_u64(void *tmp, size_t tmp_bytes, const uint64_t *in, uint64_t *out, size_t n, hipStream_t s) {
    return scan_excl<uint64_t, uint64_t>(tmp, tmp_bytes, in, out, n, s);
}

// tmp: 3 x scan_temp_bytes(n).  The outputs may be their inputs' arrays only where the element types agree (out2 / in2).
hipError_t scan3_excl(void *tmp, size_t tmp_bytes, const uint32_t *in0, uint64_t *out0, const uint32_t *in1, uint64_t *out1, const uint32_t *in2,
                      uint32_t *out2, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tmp_bytes < 3 * tiles * sizeof(uint64_t)) return hipErrorInvalidValue;
    uint64_t *part = (uint64_t *)tmp;
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan3_apply, dim3(1, 3), dim3(SCAN_THREADS), 0, s, in0, out0, in1, out1, in2, out2, (uint64_t)n, (const uint64_t *)nullptr, (uint64_t)1);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_scan3_reduce, dim3((unsigned)tiles, 3), dim3(SCAN_THREADS), 0, s, in0, in1, in2, (uint64_t)n, part, (uint64_t)tiles);
    hipLaunchKernelGGL(k_scan3_tiles, dim3(3), dim3(SCAN_THREADS), 0, s, part, (uint64_t)tiles);
    hipLaunchKernelGGL(k_scan3_apply, dim3((unsigned)tiles, 3), dim3(SCAN_THREADS), 0, s, in0, out0, in1, out1, in2, out2, (uint64_t)n, (const uint64_t *)part, (uint64_t)tiles);
    return hipGetLastError();
}

}  // namespace ii2
