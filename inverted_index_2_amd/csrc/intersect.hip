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

// ---- pre-pass: tile doc ranges and per-list phase descriptors ----------------------------
// desc layout per tile (DESC_STRIDE(n) = 2 + 4n words): [lo, hi] then per list j (0 = driver)
// {bl, bh, qlo, qhi}: block range and payload byte range of the phase (tile, j).
__host__ __device__ constexpr uint32_t desc_stride(uint32_t n) { return 2u + 4u * n; }

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
    } else {
        hi = *d.last_doc;                        // last tile ends at the list's last id
    }
    uint32_t *r = p.ranges + (uint64_t)t * desc_stride(n);
    if (j == 0) {
        r[0] = lo; r[1] = hi;
        r[2] = b0; r[3] = b1; r[4] = d.skip[b0].byte_off; r[5] = d.skip[b1].byte_off;
        return;
    }
    const ListView L = p.lists[j];
    // first block that may hold ids >= lo: the last block whose first_doc <= lo
    uint32_t ub = skip_upper_bound(L.skip, 0u, L.nblk, lo);
    const uint32_t bl = ub ? ub - 1u : 0u;
    const uint32_t bh = skip_upper_bound(L.skip, bl, L.nblk, hi);   // first block starting after hi
    r[2 + 4 * j] = bl;
    r[3 + 4 * j] = bh;
    r[4 + 4 * j] = L.skip[bl].byte_off;
    r[5 + 4 * j] = L.skip[bh].byte_off;
}

// ---- the tile kernel ----------------------------------------------------------------------
// Persistent workgroups: the grid never exceeds what is co-resident (host: <= 4 per CU), and
// workgroup w walks tiles w, w+grid, w+2*grid, ...  Every tile a workgroup waits on in the
// look-back is therefore held by a resident workgroup that is working on it — forward
// progress does not depend on dispatch order and needs no atomic ticket.
//
// Software pipeline.  A phase = (tile, list).  Its payload bytes and skip entries are
// contiguous in HBM, so all 256 threads fetch them with 16-byte loads into registers one
// phase AHEAD (while the previous phase decodes out of LDS), then park them in LDS; the
// waves decode from LDS.  Phases too large for the staging buffer decode straight from HBM.
constexpr uint32_t RAWCAP = 8192;     // staged payload bytes per phase (2 x uint4 per thread)
constexpr uint32_t SKIPCAP = 256;     // staged skip entries per phase (1 per thread)
constexpr uint32_t DESC_WORDS = 2u + 4u * MAX_LISTS;

struct __align__(16) IsectSmem {
    uint8_t map[ISECT_SMAX];                 // byte map | gallop: cand[GMAX*256] u32 + hit[GMAX*256] u8
    uint32_t stage[ISECT_GMAX * 256];        // survivors | gallop: 4 x 256 decoded block (one per wave)
    uint8_t raw[RAWCAP + 16];
    ii2_skip skipbuf[SKIPCAP + 8];
    uint32_t desc[2][DESC_WORDS];
    uint32_t wcnt[4];
    uint32_t ncand;
    uint32_t pad;
    unsigned long long base;
};

__device__ __forceinline__ bool tomb_hit(const uint32_t *__restrict__ tomb, uint32_t nwords, uint32_t doc) {
    const uint32_t w = doc >> 5;
    return w < nwords && ((tomb[w] >> (doc & 31u)) & 1u);
}

struct Phase { uint32_t bl, bh, qlo, qhi; };
__device__ __forceinline__ bool can_stage(const Phase &d) {
    return d.bh > d.bl && d.bh - d.bl < SKIPCAP && d.qhi - (d.qlo & ~15u) <= RAWCAP;
}
struct Prefetch { uint4 r0, r1; ii2_skip sk; };

__device__ __forceinline__ void prefetch_issue(Prefetch &pf, const Phase &d, const ListView &L, int tid) {
    const uint32_t base16 = d.qlo & ~15u;
    const uint32_t nch = (d.qhi - base16 + 15u) >> 4;
    const uint8_t *src = L.payload + base16;
    pf.r0 = make_uint4(0, 0, 0, 0);
    pf.r1 = make_uint4(0, 0, 0, 0);
    pf.sk.first_doc = 0; pf.sk.byte_off = 0;
    if ((uint32_t)tid < nch) pf.r0 = *reinterpret_cast<const uint4 *>(src + 16u * (uint32_t)tid);
    if ((uint32_t)tid + 256u < nch) pf.r1 = *reinterpret_cast<const uint4 *>(src + 16u * ((uint32_t)tid + 256u));
    if ((uint32_t)tid <= d.bh - d.bl) pf.sk = L.skip[d.bl + (uint32_t)tid];
}
__device__ __forceinline__ void prefetch_commit(IsectSmem &sm, const Prefetch &pf, const Phase &d, int tid) {
    const uint32_t base16 = d.qlo & ~15u;
    const uint32_t nch = (d.qhi - base16 + 15u) >> 4;
    if ((uint32_t)tid < nch) *reinterpret_cast<uint4 *>(&sm.raw[16u * (uint32_t)tid]) = pf.r0;
    if ((uint32_t)tid + 256u < nch) *reinterpret_cast<uint4 *>(&sm.raw[16u * ((uint32_t)tid + 256u)]) = pf.r1;
    if ((uint32_t)tid <= d.bh - d.bl) sm.skipbuf[tid] = pf.sk;
}

__global__ __launch_bounds__(256) void k_isect_tiles(IntersectParams p) {
    __shared__ IsectSmem sm;
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t n = p.n_lists;
    const uint32_t stride = desc_stride(n);
    Prefetch pf;
    pf.r0 = make_uint4(0, 0, 0, 0); pf.r1 = pf.r0; pf.sk.first_doc = 0; pf.sk.byte_off = 0;

    uint32_t tile = blockIdx.x;
    if (tile >= p.n_tiles) return;
    // diagnostics only: thread 0 sums the cycles spent in each part of the tile loop
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();
    // prime the pipeline: descriptors of the first tile, then its driver phase
    if ((uint32_t)tid < stride) sm.desc[0][tid] = p.ranges[(uint64_t)tile * stride + tid];
    __syncthreads();
    {
        const uint32_t *D = sm.desc[0];
        const Phase d0 = {D[2], D[3], D[4], D[5]};
        if (D[1] - D[0] < ISECT_SMAX && can_stage(d0)) prefetch_issue(pf, d0, p.lists[0], tid);
    }

    for (uint32_t it = 0; tile < p.n_tiles; it++, tile += gridDim.x) {
        const uint32_t *D = sm.desc[it & 1u];
        const uint32_t lo = D[0], hi = D[1];
        const uint32_t span = hi - lo;
        const uint32_t next_tile = tile + gridDim.x;
        const bool has_next = next_tile < p.n_tiles;
        // descriptors of this workgroup's next tile: one word per thread, in flight during the tile
        uint32_t dreg = 0;
        if (has_next && (uint32_t)tid < stride) dreg = p.ranges[(uint64_t)next_tile * stride + tid];
        const ListView drv = p.lists[0];
        uint32_t total = 0;

        if (span < ISECT_SMAX) {
            // ================= byte-map path =================
            const uint32_t nbytes = (span + 1u + 15u) & ~15u;
            for (uint32_t i = (uint32_t)tid * 16u; i < nbytes; i += 256u * 16u)
                *reinterpret_cast<uint4 *>(&sm.map[i]) = make_uint4(0, 0, 0, 0);
            for (uint32_t j = 0; j < n; j++) {
                const ListView L = p.lists[j];
                const Phase d = {D[2 + 4 * j], D[3 + 4 * j], D[4 + 4 * j], D[5 + 4 * j]};
                const bool staged = can_stage(d);
                if (j == n - 1u && has_next && (uint32_t)tid < stride) sm.desc[(it + 1u) & 1u][tid] = dreg;
                if (staged) prefetch_commit(sm, pf, d, tid);
                __syncthreads();
                // fetch the next phase while this one decodes
                if (j + 1u < n) {
                    const Phase dn = {D[6 + 4 * j], D[7 + 4 * j], D[8 + 4 * j], D[9 + 4 * j]};
                    if (can_stage(dn)) prefetch_issue(pf, dn, p.lists[j + 1u], tid);
                } else if (has_next) {
                    const uint32_t *DN = sm.desc[(it + 1u) & 1u];
                    const Phase dn = {DN[2], DN[3], DN[4], DN[5]};
                    if (DN[1] - DN[0] < ISECT_SMAX && can_stage(dn)) prefetch_issue(pf, dn, drv, tid);
                }
                II2_STAMP(0)      // clear + commit + barrier + prefetch issue
                const uint32_t nblk = d.bh - d.bl;
                const uint8_t want = (uint8_t)j;
                auto mark = [&](uint32_t, uint32_t id) {
                    const uint32_t off = id - lo;
                    if (off <= span && sm.map[off] == want) sm.map[off] = (uint8_t)(want + 1u);
                };
                if (staged) {
                    const uint32_t base16 = d.qlo & ~15u;
                    for (uint32_t i = (uint32_t)wv; i < nblk; i += 4u) {
                        const ii2_skip e0 = sm.skipbuf[i], e1 = sm.skipbuf[i + 1u];
                        decode_block_wave(LdsBytes{sm.raw}, e0.byte_off - base16, e1.byte_off - base16, e0.first_doc, mark);
                    }
                } else {
                    for (uint32_t b = d.bl + (uint32_t)wv; b < d.bh; b += 4u)
                        decode_block_wave(L.payload, L.skip[b].byte_off, L.skip[b + 1].byte_off, L.skip[b].first_doc, mark);
                }
                II2_STAMP(1)      // decode
                __syncthreads();
                II2_STAMP(2)      // barrier after decode
            }
            // pass 1: finalise (tombstones) and count; wave w owns a contiguous quarter of the map
            const uint32_t quarter = ((nbytes / 4u) + 255u) & ~255u;
            const uint32_t w0 = (uint32_t)wv * quarter;
            const uint32_t w1 = w0 + quarter < nbytes ? w0 + quarter : nbytes;
            uint32_t mine = 0;
            for (uint32_t base = w0; base < w1; base += 256u) {
                const uint32_t off = base + 4u * (uint32_t)l;
                if (off < w1) {
                    uint32_t w = *reinterpret_cast<uint32_t *>(&sm.map[off]);
                    uint32_t out = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (((w >> (8 * k)) & 0xFFu) == n) {
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
            II2_STAMP(3)          // pass 1
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
            II2_STAMP(4)          // pass 2
        } else {
            // ================= gallop path =================
            const uint32_t b0 = D[2], b1 = D[3];
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
                const uint32_t bl = D[2 + 4 * j], bh = D[3 + 4 * j];
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
            // restart the pipeline for the next tile
            if (has_next && (uint32_t)tid < stride) sm.desc[(it + 1u) & 1u][tid] = dreg;
            __syncthreads();
            if (has_next) {
                const uint32_t *DN = sm.desc[(it + 1u) & 1u];
                const Phase dn = {DN[2], DN[3], DN[4], DN[5]};
                if (DN[1] - DN[0] < ISECT_SMAX && can_stage(dn)) prefetch_issue(pf, dn, drv, tid);
            }
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
        __syncthreads();      // stage / map / base are reused by the next tile
        II2_STAMP(5)              // look-back + output
    }
    if (stamps && tid == 0)
        for (int i = 0; i < 8; i++) p.debug[(uint64_t)blockIdx.x * 8u + i] = tacc[i];
#undef II2_STAMP
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

hipError_t launch_intersect(const IntersectParams &p, uint64_t *d_tile_off, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (p.n_tiles == 0) return hipSuccess;
    const uint64_t nthr = (uint64_t)p.n_tiles * p.n_lists;
    hipLaunchKernelGGL(k_isect_partition, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p);
    const uint32_t grid = p.n_tiles < p.max_grid ? p.n_tiles : p.max_grid;
    if (ev0) (void)hipEventRecord(ev0, s);
    hipLaunchKernelGGL(k_isect_tiles, dim3(grid), dim3(256), 0, s, p);
    if (ev1) (void)hipEventRecord(ev1, s);
    if (!p.lookback) {
        hipLaunchKernelGGL(k_isect_scan_counts, dim3(1), dim3(1024), 0, s, (const uint32_t *)p.tile_count, p.n_tiles, d_tile_off, p.d_count);
        hipLaunchKernelGGL(k_isect_copy, dim3(p.n_tiles), dim3(256), 0, s, p, (const uint64_t *)d_tile_off);
    }
    return hipGetLastError();
}

}  // namespace ii2
