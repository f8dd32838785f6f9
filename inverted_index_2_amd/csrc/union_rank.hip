// union_rank.hip — OR of a few lists of medium size (up to 8 lists, up to 2^20 postings in all): four launches.
// Between the one-launch kernel for short lists (setop_small.hip) and the streaming / tiled kernels for dense ones the
// union used to fall to the merge passes (~40 launches, 150-180 us whatever the size) or to OR tiles over the whole doc
// range (one tile per 16k docs: ~100 us of mostly empty tiles for sparse lists).  Same idea as the short-list kernel,
// with global arrays instead of LDS:
//   1. k_ur_decode: one wave per DV1 block -> raw[lpre[j] + 256 bi ...] (the host knows every list's size, so list j is
//      one ascending stretch of raw);
//   2. k_ur_rank: an id's rank among all ids = its position in its own list + one bisection per other list (ties broken
//      by the list number, so the ranks are a permutation); sorted[rank] = id;
//   3. k_ur_count: 2048 ids per workgroup — the first id of every run of equal ids survives unless the tombstone bitmap
//      has it; survivors per workgroup;
//   4. k_ur_write: the same flags again, positions = the counts of the workgroups before (summed by the workgroup itself:
//      at most 512 words) + a block scan; the last workgroup stores the total.
#include <hip/hip_runtime.h>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t UR_IDS_PER_WG = 2048;        // steps 3 / 4: 256 threads x 8 ids

__global__ __launch_bounds__(256) void k_ur_decode(UnionRankParams p) {
    const uint32_t l = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= p.n_blocks) return;                                           // (wave-uniform)
    uint32_t j = 0;
#pragma unroll
    for (uint32_t c = 1; c < UNION_RANK_MAXL; c++) j += (c < p.n_lists && p.blk_base[c] <= b) ? 1u : 0u;
    uint32_t base = 0, lp = 0, lend = 0;
    ListView L = p.lists[0];
#pragma unroll
    for (uint32_t c = 0; c < UNION_RANK_MAXL; c++)
        if (c == j) { L = p.lists[c]; base = p.blk_base[c]; lp = p.lpre[c]; lend = p.lpre[c + 1u]; }
    const uint32_t bi = b - base;
    const ii2_skip e0 = L.skip[bi], e1 = L.skip[bi + 1u];
    const uint32_t at = lp + bi * II2_DV1_BLOCK;
    (void)l;
    decode_block_wave(L.payload, e0.byte_off, e1.byte_off, e0.first_doc,
                      [&](uint32_t ix, uint32_t id) { if (at + ix < lend) p.raw[at + ix] = id; });
}

__global__ __launch_bounds__(256) void k_ur_rank(UnionRankParams p) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_total = p.lpre[p.n_lists];
    if (e >= n_total) return;
    uint32_t j = 0;
#pragma unroll
    for (uint32_t c = 1; c < UNION_RANK_MAXL; c++) j += (c < p.n_lists && p.lpre[c] <= e) ? 1u : 0u;
    uint32_t lp = 0;
#pragma unroll
    for (uint32_t c = 0; c < UNION_RANK_MAXL; c++) if (c == j) lp = p.lpre[c];
    const uint32_t x = p.raw[e];
    uint32_t r = e - lp;
#pragma unroll
    for (uint32_t c = 0; c < UNION_RANK_MAXL; c++) {
        if (c >= p.n_lists || c == j) continue;
        const uint32_t *B = p.raw + p.lpre[c];
        uint32_t lo = 0, hi = p.lpre[c + 1u] - p.lpre[c];                  // first index with B[i] > x (c < j) or >= x (c > j)
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t y = B[mid];
            if (y < x || (c < j && y == x)) lo = mid + 1u; else hi = mid;
        }
        r += lo;
    }
    p.sorted[r] = x;
}

// the eight ids of this thread that survive: bit q of the mask, ids in v[]
__device__ __forceinline__ uint32_t ur_flags(const UnionRankParams &p, uint32_t n_total, uint32_t a0, uint32_t *v) {
    uint32_t mask = 0;
    if (a0 >= n_total) return 0u;
    uint32_t prev = a0 ? p.sorted[a0 - 1u] : 0u;
#pragma unroll
    for (uint32_t q = 0; q < 8u; q++) {
        const uint32_t i = a0 + q;
        v[q] = 0;
        if (i >= n_total) continue;
        const uint32_t x = p.sorted[i];
        v[q] = x;
        bool keep = i == 0u || prev != x;                                  // first of its run
        prev = x;
        if (keep && p.tomb && (x >> 5) < p.tomb_nwords) keep = ((p.tomb[x >> 5] >> (x & 31u)) & 1u) == 0u;
        if (keep) mask |= 1u << q;
    }
    return mask;
}

__global__ __launch_bounds__(256) void k_ur_count(UnionRankParams p) {
    __shared__ uint32_t ws[4];
    const uint32_t n_total = p.lpre[p.n_lists];
    uint32_t v[8];
    const uint32_t mask = ur_flags(p, n_total, blockIdx.x * UR_IDS_PER_WG + 8u * threadIdx.x, v);
    const uint32_t s = wave_sum((uint32_t)__popc(mask));
    if ((threadIdx.x & 63u) == 0u) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) p.wg_cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_ur_write(UnionRankParams p) {
    __shared__ unsigned long long wsum[4];
    __shared__ uint32_t ws[4];
    const uint32_t tid = threadIdx.x, l = tid & 63u, wv = tid >> 6;
    const uint32_t n_total = p.lpre[p.n_lists];
    uint32_t v[8];
    const uint32_t mask = ur_flags(p, n_total, blockIdx.x * UR_IDS_PER_WG + 8u * tid, v);
    // survivors of the workgroups before mine
    unsigned long long mine = 0;
    for (uint32_t g = tid; g < blockIdx.x; g += 256u) mine += p.wg_cnt[g];
    for (int d = 32; d >= 1; d >>= 1) mine += (unsigned long long)__shfl_xor((long long)mine, d, 64);
    const uint32_t c = (uint32_t)__popc(mask);
    const uint32_t incl = wave_incl_scan(c);
    if (l == 0) wsum[wv] = mine;
    if (l == 63u) ws[wv] = incl;
    __syncthreads();
    unsigned long long pos = wsum[0] + wsum[1] + wsum[2] + wsum[3] + (incl - c);
    uint32_t total_wg = 0;
    for (uint32_t w = 0; w < 4u; w++) { if (w < wv) pos += ws[w]; total_wg += ws[w]; }
#pragma unroll
    for (uint32_t q = 0; q < 8u; q++)
        if ((mask >> q) & 1u) { if (pos < p.out_cap) p.out[pos] = v[q]; pos++; }
    if (blockIdx.x == gridDim.x - 1u && tid == 0) *p.d_count = wsum[0] + wsum[1] + wsum[2] + wsum[3] + total_wg;
}

hipError_t launch_union_rank(const UnionRankParams &p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, s);
    const uint32_t n_total = p.lpre[p.n_lists];
    const uint32_t nwg = (n_total + UR_IDS_PER_WG - 1u) / UR_IDS_PER_WG;
    hipLaunchKernelGGL(k_ur_decode, dim3((p.n_blocks + 3u) / 4u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_ur_rank, dim3((n_total + 255u) / 256u), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_ur_count, dim3(nwg), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_ur_write, dim3(nwg), dim3(256), 0, s, p);
    if (ev1) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}

}  // namespace ii2
