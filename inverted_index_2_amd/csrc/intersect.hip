// intersect.hip — multi-term intersection over DV1 lists (gfx950, wave64, no MFMA).
//
// Build-defined operator on top of the reference's Read lists:
// AND(t1..tn) = ascending ids present in every list, optionally minus the tombstone bitmap
// (shard.go:181-190 semantics as a bit test).
//
// Tiling.  The shortest list is the DRIVER.  A tile = G consecutive driver blocks
// (G*256 candidate ids, doc range [lo, hi]); a pre-pass finds, per tile and per other list,
// the block range whose ids can fall in [lo, hi] (binary search on the skip tables).  One
// 256-thread workgroup per tile:
//   byte-map path (hi-lo < 16 Ki docs — dense lists): the tile keeps one LDS byte per doc of
//     its range.  Driver postings write 1; list j's postings turn a j into j+1 (lists are
//     duplicate-free, so no two lanes ever race on a byte); bytes equal to n are the
//     result.  Every block is decoded by one wave straight from HBM (4 bytes per lane, one
//     DPP prefix sum), so a posting costs one LDS byte access and nothing is sorted,
//     merged or searched.
//   gallop path (sparse / skewed tiles): candidates sit in LDS as a sorted array; for each
//     other list every live candidate binary-searches that list's skip table for the one
//     block that could hold it, a wave decodes just those blocks into LDS and the
//     candidates search them.  Blocks without a candidate are never touched.
// Output order: tiles are in doc order and a tile's survivors are in doc order, so the
// result is the concatenation of the tiles' survivors; offsets come from a decoupled
// look-back over the tile counts (single pass) or, with lookback off, from a count scan
// and a copy pass.
#include "dv1_device.h"
#include "internal.h"
#include "lookback.h"

namespace ii2 {

constexpr uint32_t NONE = 0xFFFFFFFFu;

// ---- pre-pass: tile doc ranges and per-list block ranges --------------------------------
// ranges layout per tile: [lo, hi, bl_1, bh_1, ..., bl_{n-1}, bh_{n-1}]
__global__ void k_isect_partition(IntersectParams p) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = p.n_lists;
    if (gid >= (uint64_t)p.n_tiles * n) return;
    const uint32_t t = (uint32_t)(gid / n), j = (uint32_t)(gid % n);
    const ListView d = p.lists[0];
    const uint32_t b0 = t * p.G;
    const uint32_t b1 = b0 + p.G < d.nblk ? b0 + p.G : d.nblk;
    const uint32_t lo = d.skip[b0].first_doc;
    uint32_t hi;
    if (b1 < d.nblk) {
        hi = d.skip[b1].first_doc - 1u;
    } else {                                     // last tile: walk the last block to its last id
        const uint32_t b = d.nblk - 1u;
        uint32_t q = d.skip[b].byte_off;
        const uint32_t qe = d.skip[b + 1].byte_off;
        uint32_t cur = d.skip[b].first_doc;
        while (q < qe) {
            uint32_t v = 0, sh = 0, c;
            do { c = d.payload[q++]; v |= (c & 0x7Fu) << sh; sh += 7; } while ((c & 0x80u) && q < qe && sh < 35);
            cur += v;
        }
        hi = cur;
    }
    uint32_t *r = p.ranges + (uint64_t)t * 2u * n;
    if (j == 0) { r[0] = lo; r[1] = hi; return; }
    const ListView L = p.lists[j];
    // first block that may hold ids >= lo: the last block whose first_doc <= lo
    uint32_t ub = skip_upper_bound(L.skip, 0u, L.nblk, lo);
    const uint32_t bl = ub ? ub - 1u : 0u;
    const uint32_t bh = skip_upper_bound(L.skip, bl, L.nblk, hi);   // first block starting after hi
    r[2 * j] = bl;
    r[2 * j + 1] = bh;
}

// ---- the tile kernel ----------------------------------------------------------------------
struct __align__(16) IsectSmem {
    uint8_t map[ISECT_SMAX];                 // byte map | gallop: cand[GMAX*256] u32 + hit[GMAX*256] u8
    uint32_t stage[ISECT_GMAX * 256];        // survivors | gallop: 4 x 256 decoded block (one per wave)
    uint32_t wcnt[4];
    uint32_t ncand;
    uint32_t tile;
    unsigned long long base;
};

__device__ __forceinline__ bool tomb_hit(const uint32_t *__restrict__ tomb, uint32_t nwords, uint32_t doc) {
    const uint32_t w = doc >> 5;
    return w < nwords && ((tomb[w] >> (doc & 31u)) & 1u);
}

__global__ __launch_bounds__(256) void k_isect_tiles(IntersectParams p) {
    __shared__ IsectSmem sm;
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t n = p.n_lists;

    uint32_t tile;
    if (p.lookback) {
        if (tid == 0) sm.tile = atomicAdd(p.ticket, 1u) - p.ticket_base;
        __syncthreads();
        tile = sm.tile;
    } else {
        tile = blockIdx.x;
    }
    const uint32_t *r = p.ranges + (uint64_t)tile * 2u * n;
    const uint32_t lo = r[0], hi = r[1];
    const uint32_t span = hi - lo;
    const ListView drv = p.lists[0];
    const uint32_t b0 = tile * p.G;
    const uint32_t b1 = b0 + p.G < drv.nblk ? b0 + p.G : drv.nblk;
    uint32_t total = 0;

    if (span < ISECT_SMAX) {
        // ================= byte-map path =================
        const uint32_t nbytes = (span + 1u + 15u) & ~15u;
        for (uint32_t i = (uint32_t)tid * 16u; i < nbytes; i += 256u * 16u)
            *reinterpret_cast<uint4 *>(&sm.map[i]) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        for (uint32_t b = b0 + (uint32_t)wv; b < b1; b += 4u) {
            decode_block_wave(drv.payload, drv.skip[b].byte_off, drv.skip[b + 1].byte_off, drv.skip[b].first_doc,
                              [&](uint32_t, uint32_t id) {
                                  const uint32_t off = id - lo;
                                  if (off <= span) sm.map[off] = 1;
                              });
        }
        __syncthreads();
        for (uint32_t j = 1; j < n; j++) {
            const ListView L = p.lists[j];
            const uint32_t bl = r[2 * j], bh = r[2 * j + 1];
            const uint8_t want = (uint8_t)j;
            for (uint32_t b = bl + (uint32_t)wv; b < bh; b += 4u) {
                decode_block_wave(L.payload, L.skip[b].byte_off, L.skip[b + 1].byte_off, L.skip[b].first_doc,
                                  [&](uint32_t, uint32_t id) {
                                      const uint32_t off = id - lo;
                                      if (off <= span && sm.map[off] == want) sm.map[off] = (uint8_t)(want + 1u);
                                  });
            }
            __syncthreads();
        }
        // pass 1: finalise (tombstones) and count; wave w owns a contiguous quarter of the map
        const uint32_t quarter = ((nbytes / 4u) + 255u) & ~255u;
        const uint32_t w0 = (uint32_t)wv * quarter;
        const uint32_t w1 = w0 + quarter < nbytes ? w0 + quarter : nbytes;
        const uint32_t full = n * 0x01010101u;
        uint32_t mine = 0;
        for (uint32_t base = w0; base < w1; base += 256u) {
            const uint32_t off = base + 4u * (uint32_t)l;
            if (off < w1) {
                uint32_t w = *reinterpret_cast<uint32_t *>(&sm.map[off]);
                uint32_t out = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (((w >> (8 * k)) & 0xFFu) == (full & 0xFFu)) {
                        const uint32_t doc = lo + off + (uint32_t)k;
                        if (!(p.tomb && tomb_hit(p.tomb, p.tomb_nwords, doc))) { out |= 0xFFu << (8 * k); mine++; }
                    }
                }
                *reinterpret_cast<uint32_t *>(&sm.map[off]) = out;
            }
        }
        mine = wave_sum(mine);
        if (l == 0) sm.wcnt[wv] = mine;
        __syncthreads();
        uint32_t pos = 0;
        for (int w = 0; w < wv; w++) pos += sm.wcnt[w];
        total = sm.wcnt[0] + sm.wcnt[1] + sm.wcnt[2] + sm.wcnt[3];
        // pass 2: ordered write into stage
        for (uint32_t base = w0; base < w1; base += 256u) {
            const uint32_t off = base + 4u * (uint32_t)l;
            uint32_t w = off < w1 ? *reinterpret_cast<uint32_t *>(&sm.map[off]) : 0u;
            const uint32_t m = (uint32_t)__popc(w & 0x01010101u);
            const uint32_t incl = wave_incl_scan(m);
            uint32_t q = pos + incl - m;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((w >> (8 * k)) & 1u) sm.stage[q++] = lo + off + (uint32_t)k;
            pos += wave_bcast(incl, 63);
        }
        __syncthreads();
    } else {
        // ================= gallop path =================
        uint32_t *cand = reinterpret_cast<uint32_t *>(sm.map);
        uint8_t *hit = sm.map + ISECT_GMAX * 256u * 4u;
        uint32_t *wbuf = sm.stage + (uint32_t)wv * 256u;
        for (uint32_t b = b0 + (uint32_t)wv; b < b1; b += 4u) {
            const uint32_t pb = (b - b0) * 256u;
            const uint32_t c = decode_block_wave(drv.payload, drv.skip[b].byte_off, drv.skip[b + 1].byte_off,
                                                 drv.skip[b].first_doc, [&](uint32_t ix, uint32_t id) {
                                                     cand[pb + ix] = id;
                                                     hit[pb + ix] = 1;
                                                 });
            if (b == b1 - 1u && l == 0) sm.ncand = pb + c;
        }
        __syncthreads();
        const uint32_t ncand = sm.ncand;
        for (uint32_t j = 1; j < n; j++) {
            const ListView L = p.lists[j];
            const uint32_t bl = r[2 * j], bh = r[2 * j + 1];
            for (uint32_t base = (uint32_t)wv * 64u; base < ncand; base += 256u) {
                const uint32_t pi = base + (uint32_t)l;
                const bool alive = pi < ncand && hit[pi] == (uint8_t)j;
                const uint32_t c = alive ? cand[pi] : 0u;
                uint32_t blk = NONE;
                if (alive && bl < bh) {
                    const uint32_t ub = skip_upper_bound(L.skip, bl, bh, c);
                    if (ub > bl) blk = ub - 1u;
                }
                unsigned long long pending = __ballot(blk != NONE);
                while (pending) {
                    const int leader = __ffsll((long long)pending) - 1;
                    const uint32_t cur = wave_bcast(blk, leader);
                    const uint32_t cnt = decode_block_wave(L.payload, L.skip[cur].byte_off, L.skip[cur + 1].byte_off,
                                                           L.skip[cur].first_doc,
                                                           [&](uint32_t ix, uint32_t id) { wbuf[ix] = id; });
                    __threadfence_block();
                    if (blk == cur) {
                        uint32_t a = 0, e = cnt;
                        while (a < e) {
                            const uint32_t mid = (a + e) >> 1;
                            if (wbuf[mid] < c) a = mid + 1u; else e = mid;
                        }
                        if (a < cnt && wbuf[a] == c) hit[pi] = (uint8_t)(j + 1u);
                    }
                    __threadfence_block();
                    pending &= ~__ballot(blk == cur);
                }
            }
            __syncthreads();
        }
        // finalise + count: wave w owns candidates [w*Q, (w+1)*Q)
        const uint32_t quarter = (((ncand + 3u) / 4u) + 63u) & ~63u;
        const uint32_t w0 = (uint32_t)wv * quarter;
        const uint32_t w1 = w0 + quarter < ncand ? w0 + quarter : ncand;
        uint32_t mine = 0;
        for (uint32_t base = w0; base < w1; base += 64u) {
            const uint32_t pi = base + (uint32_t)l;
            if (pi < w1) {
                bool keep = hit[pi] == (uint8_t)n;
                if (keep && p.tomb && tomb_hit(p.tomb, p.tomb_nwords, cand[pi])) keep = false;
                if (keep && pi > 0 && cand[pi - 1] == cand[pi]) keep = false;   // tolerate a duplicated driver id
                hit[pi] = keep ? 0xFF : 0;
                mine += keep;
            }
        }
        mine = wave_sum(mine);
        if (l == 0) sm.wcnt[wv] = mine;
        __syncthreads();               // also: all waves are done with wbuf (aliases stage)
        uint32_t pos = 0;
        for (int w = 0; w < wv; w++) pos += sm.wcnt[w];
        total = sm.wcnt[0] + sm.wcnt[1] + sm.wcnt[2] + sm.wcnt[3];
        for (uint32_t base = w0; base < w1; base += 64u) {
            const uint32_t pi = base + (uint32_t)l;
            const uint32_t m = (pi < w1 && hit[pi]) ? 1u : 0u;
            const uint32_t incl = wave_incl_scan(m);
            if (m) sm.stage[pos + incl - 1u] = cand[pi];
            pos += wave_bcast(incl, 63);
        }
        __syncthreads();
    }

    // ---- ordered output ----
    if (p.lookback) {
        if (wv == 0) {
            const unsigned long long base = lookback_exclusive(p.desc, tile, total, p.epoch);
            if (l == 0) {
                sm.base = base;
                if (tile == p.n_tiles - 1u) *p.d_count = base + total;
            }
        }
        __syncthreads();
        const unsigned long long ob = sm.base;
        for (uint32_t i = (uint32_t)tid; i < total; i += 256u)
            if (ob + i < p.out_cap) p.out[ob + i] = sm.stage[i];
    } else {
        uint32_t *dst = p.tmp + (uint64_t)tile * p.G * 256u;
        for (uint32_t i = (uint32_t)tid; i < total; i += 256u) dst[i] = sm.stage[i];
        if (tid == 0) p.tile_count[tile] = total;
    }
}

// ---- non-lookback epilogue: offsets from a serial-chunked scan, then a copy ------------
__global__ __launch_bounds__(1024) void k_isect_scan_counts(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *__restrict__ off,
                                                            uint64_t *__restrict__ d_count) {
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long carry;
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t v = i < n ? cnt[i] : 0u;
        const uint32_t incl = wave_incl_scan(v);
        if (l == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t pre = 0;
        for (int w = 0; w < wv; w++) pre += wsum[w];
        if (i < n) off[i] = carry + pre + incl - v;
        __syncthreads();
        if (tid == 1023) carry += (unsigned long long)pre + incl;
        __syncthreads();
    }
    if (tid == 0) { off[n] = carry; *d_count = carry; }
}

__global__ __launch_bounds__(256) void k_isect_copy(IntersectParams p, const uint64_t *__restrict__ off) {
    const uint32_t tile = blockIdx.x;
    const uint32_t c = p.tile_count[tile];
    const uint32_t *src = p.tmp + (uint64_t)tile * p.G * 256u;
    const uint64_t ob = off[tile];
    for (uint32_t i = threadIdx.x; i < c; i += 256u)
        if (ob + i < p.out_cap) p.out[ob + i] = src[i];
}

hipError_t launch_intersect(const IntersectParams &p, uint64_t *d_tile_off, hipStream_t s) {
    if (p.n_tiles == 0) return hipSuccess;
    const uint64_t nthr = (uint64_t)p.n_tiles * p.n_lists;
    hipLaunchKernelGGL(k_isect_partition, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_isect_tiles, dim3(p.n_tiles), dim3(256), 0, s, p);
    if (!p.lookback) {
        hipLaunchKernelGGL(k_isect_scan_counts, dim3(1), dim3(1024), 0, s, (const uint32_t *)p.tile_count, p.n_tiles, d_tile_off, p.d_count);
        hipLaunchKernelGGL(k_isect_copy, dim3(p.n_tiles), dim3(256), 0, s, p, (const uint64_t *)d_tile_off);
    }
    return hipGetLastError();
}

}  // namespace ii2
