// setop_small.hip — AND / OR of posting lists that together hold at most 8192 postings (in at most 128 DV1 blocks): one launch.
// The general paths cut a query into tiles over four (intersection) or some forty (union through the merge passes)
// launches; below a few thousand postings that is all launch and dependency latency — 14-20 us per AND, 110-150 us
// per OR — and short lists are what most terms of a real dictionary have (PrefixSearch, inverted_index.go:274-292,
// unions the lists of every matching term).
//
//   1. every workgroup decodes all blocks into its LDS, one wave per block (redundant across workgroups, but cheaper than
//      a second launch); the host knows every list's size, so list j is one ascending stretch raw[lpre[j] ...];
//   2. an id's rank among all ids = its position in its own list + one bisection per other list (ties broken by the
//      list number, so the ranks are a permutation) — no sort network.  One workgroup stores every id at its rank in
//      LDS; when the bisections are too many for one CU (many lists) the ids are shared out over up to 32 workgroups,
//      which store into a small global array, and the workgroup that finishes last (a device-wide ticket) goes on;
//   3. over the ascending ids: the first id of every run
//      of equal ids survives — for an AND only when the run is n_lists long (each list holds an id once, so
//      id[i + n_lists - 1] == id[i] says it all) — unless the tombstone bitmap has it; block scan, write-out, count.
#include <hip/hip_runtime.h>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t SS_THREADS = 1024;
constexpr uint32_t SS_WAVES = SS_THREADS / 64u;

__global__ __launch_bounds__(SS_THREADS) void k_setop_small(SmallSetParams p) {
    __shared__ uint32_t raw[SMALL_SET_POSTINGS];                    // list j decoded at raw[lpre[j] ...], ascending
    __shared__ uint32_t lcnt[MAX_LISTS], lpre[MAX_LISTS + 1];       // postings of every list (the host knows them), their prefix
    __shared__ uint32_t lbase[MAX_LISTS + 1];                       // first block of every list in the concatenated block list
    __shared__ uint32_t wsum[SS_WAVES];
    __shared__ uint32_t last_s;
    __shared__ uint8_t blist[SMALL_SET_BLOCKS];                     // the list every block belongs to
    const uint32_t tid = threadIdx.x, l = tid & 63u, wv = tid >> 6;
    if (tid <= p.n_lists) {
        lpre[tid] = p.lpre[tid];
        lbase[tid] = p.blk_base[tid];
        if (tid < p.n_lists) {
            lcnt[tid] = p.lpre[tid + 1u] - p.lpre[tid];
            for (uint32_t b = p.blk_base[tid]; b < p.blk_base[tid + 1u]; b++) blist[b] = (uint8_t)tid;
        }
    }
    __syncthreads();
    // 1. decode: block b of the concatenated block list, one wave each (wave w: blocks w, w + 16, ...); every block of
    // a list but its last is full, so block bi of list j starts at raw[lpre[j] + 256 bi].  The skip entries of all the
    // wave's blocks are requested first, then the first 256 payload bytes of all of them, then they are decoded: two
    // memory round trips per wave instead of two per block.
    constexpr uint32_t PER_WAVE = SMALL_SET_BLOCKS / SS_WAVES;
    uint32_t bj[PER_WAVE], q0[PER_WAVE], q1[PER_WAVE], f0[PER_WAVE], pw[PER_WAVE];
#pragma unroll
    for (uint32_t t = 0; t < PER_WAVE; t++) {
        const uint32_t b = wv + t * SS_WAVES;
        bj[t] = 0xFFFFFFFFu; q0[t] = 0; q1[t] = 0; f0[t] = 0;
        if (b < p.n_blocks) {
            const uint32_t j = blist[b];                                  // (wave-uniform)
            const ii2_skip *sk = p.lists[j].skip + (b - lbase[j]);
            const ii2_skip e0 = sk[0], e1 = sk[1];
            bj[t] = j; q0[t] = e0.byte_off; q1[t] = e1.byte_off; f0[t] = e0.first_doc;
        }
    }
#pragma unroll
    for (uint32_t t = 0; t < PER_WAVE; t++) {
        pw[t] = 0;
        if (bj[t] != 0xFFFFFFFFu && q0[t] + 4u * l < q1[t]) pw[t] = load_u32_unaligned(p.lists[bj[t]].payload + q0[t] + 4u * l);
    }
#pragma unroll
    for (uint32_t t = 0; t < PER_WAVE; t++) {
        if (bj[t] == 0xFFFFFFFFu) continue;                               // (wave-uniform)
        const uint32_t b = wv + t * SS_WAVES, j = bj[t];
        const uint32_t at = lpre[j] + (b - lbase[j]) * II2_DV1_BLOCK, end = lpre[j + 1u];
        const uint8_t *pl = p.lists[j].payload;
        const uint32_t first_q = q0[t], pre = pw[t];
        decode_block_wave([&](uint32_t myq) -> uint32_t { return myq == first_q + 4u * l ? pre : load_u32_unaligned(pl + myq); },
                          q0[t], q1[t], f0[t], [&](uint32_t ix, uint32_t id) { if (at + ix < end) raw[at + ix] = id; });
    }
    __syncthreads();
    const uint32_t n_total = lpre[p.n_lists];
    // 2. ranks: this workgroup's share of the ids (list-major numbering e = 0 .. n_total), at most eight per thread
    const uint32_t per_wg = (n_total + gridDim.x - 1u) / gridDim.x;
    const uint32_t e_end = (blockIdx.x + 1u) * per_wg < n_total ? (blockIdx.x + 1u) * per_wg : n_total;
    uint32_t top = 1;                                // the largest power of two <= the longest list
    for (uint32_t c = l; c < p.n_lists; c += 64u) top = lcnt[c] > top ? lcnt[c] : top;
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)top, d, 64); top = o > top ? o : top; }
    top = 1u << (31u - (uint32_t)__clz((int)top));
    uint32_t rk[8], xv[8];
#pragma unroll 4
    for (uint32_t q = 0; q < 8u; q++) {
        const uint32_t e = blockIdx.x * per_wg + tid + q * SS_THREADS;
        rk[q] = 0xFFFFFFFFu;
        xv[q] = 0;
        if (e >= e_end) continue;
        uint32_t j = 0;                                                   // my list: the last j with lpre[j] <= e
        for (uint32_t st = 32u; st > 0u; st >>= 1) if (j + st < p.n_lists && lpre[j + st] <= e) j += st;
        const uint32_t i = e - lpre[j];
        const uint32_t x = raw[lpre[j] + i];
        uint32_t r = i;
        if (p.n_lists <= 8u) {
            for (uint32_t c = 0; c < p.n_lists; c++) {
                if (c == j) continue;
                const uint32_t *B = raw + lpre[c];
                uint32_t lo = 0, hi = lcnt[c];                            // first index with B[i] > x (c < j) or >= x (c > j)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t y = B[mid];
                    if (y < x || (c < j && y == x)) lo = mid + 1u; else hi = mid;
                }
                r += lo;
            }
        } else {
            // many lists: a bisection per list is a chain of dependent LDS reads, and the chains of 63 lists one after the
            // other were most of the kernel's time — branch-free bisections with the same steps for every list, eight
            // lists (eight independent chains) at a time
            for (uint32_t c0 = 0; c0 < p.n_lists; c0 += 8u) {
                uint32_t pos[8], n[8], base[8];
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) {
                    const uint32_t c = c0 + u;
                    const bool on = c < p.n_lists && c != j;
                    n[u] = on ? lcnt[c] : 0u;
                    base[u] = on ? lpre[c] : 0u;
                    pos[u] = 0;
                }
                for (uint32_t st = top; st > 0u; st >>= 1) {
#pragma unroll
                    for (uint32_t u = 0; u < 8u; u++) {
                        const uint32_t cand = pos[u] + st;
                        if (cand <= n[u]) {
                            const uint32_t y = raw[base[u] + cand - 1u];
                            if (y < x || (c0 + u < j && y == x)) pos[u] = cand;     // ties: lists before mine go first
                        }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) r += pos[u];
            }
        }
        rk[q] = r;
        xv[q] = x;
    }
    const bool solo = gridDim.x == 1u;               // one workgroup: the ascending ids replace the decoded blocks in LDS
    if (solo) {
        __syncthreads();                             // (every rank is computed: raw may be overwritten)
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) if (rk[q] != 0xFFFFFFFFu) raw[rk[q]] = xv[q];
        __syncthreads();
    } else {
        // 3. several workgroups: the ids go to a small global array; the last workgroup to get here (a device-wide
        // ticket) filters them
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) if (rk[q] != 0xFFFFFFFFu) p.sorted[rk[q]] = xv[q];
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const uint32_t t = atomicAdd(p.ticket, 1u);
            last_s = t == gridDim.x - 1u ? 1u : 0u;
            if (last_s) *p.ticket = 0u;                                   // ready for the next launch (stream order)
        }
        __syncthreads();
        if (!last_s) return;
        __threadfence();
    }
    // written by other CUs: read at device scope (past this CU's L1), relaxed, so the loads still overlap
    auto key = [&](uint32_t i) -> uint32_t {
        return solo ? raw[i] : __hip_atomic_load(p.sorted + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const uint32_t a0 = 8u * tid;
    uint32_t kept[8];
    uint32_t keepmask = 0, cnt = 0;
    if (a0 < n_total) {
        uint32_t prev = a0 ? key(a0 - 1u) : 0u;
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) {
            const uint32_t i = a0 + q;
            kept[q] = 0;
            if (i >= n_total) continue;
            const uint32_t v = key(i);
            kept[q] = v;
            bool keep = i == 0u || prev != v;                             // first of its run
            prev = v;
            if (keep && !p.is_union) keep = i + p.n_lists - 1u < n_total && key(i + p.n_lists - 1u) == v;
            if (keep && p.tomb && (v >> 5) < p.tomb_nwords) keep = ((p.tomb[v >> 5] >> (v & 31u)) & 1u) == 0u;
            if (keep) { keepmask |= 1u << q; cnt++; }
        }
    }
    const uint32_t incl = wave_incl_scan(cnt);
    if (l == 63u) wsum[wv] = incl;
    __syncthreads();
    uint32_t pos = incl - cnt, total = 0;
    for (uint32_t w = 0; w < SS_WAVES; w++) { if (w < wv) pos += wsum[w]; total += wsum[w]; }
#pragma unroll
    for (uint32_t q = 0; q < 8u; q++)
        if ((keepmask >> q) & 1u) { if (pos < p.out_cap) p.out[pos] = kept[q]; pos++; }
    if (tid == 0) *p.d_count = total;
}

hipError_t launch_setop_small(const SmallSetParams &p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, s);
    // one workgroup ranks everything in LDS (no global round trips: ~6 us) unless the bisections — one per id and other
    // list — are too many for one CU; then the ids are shared out (at most 8 K ids per workgroup: eight per thread)
    const uint64_t work = (uint64_t)p.lpre[p.n_lists] * (p.n_lists > 1u ? p.n_lists - 1u : 1u);
    uint32_t grid = (uint32_t)((work + 16383u) / 16384u);
    grid = grid < 1u ? 1u : grid > 32u ? 32u : grid;
    hipLaunchKernelGGL(k_setop_small, dim3(grid), dim3(SS_THREADS), 0, s, p);
    if (ev1) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}

}  // namespace ii2
