// intersect.hip — multi-term intersection over DV1 lists (gfx950, wave64, no MFMA).
//
// Build-defined operator on top of the reference's Read lists:
// AND(t1..tn) = ascending ids present in every list, optionally minus the tombstone bitmap
// (shard.go:181-190 semantics as a bit test).
//
// Tiling.  The shortest list is the DRIVER.  A tile = G consecutive driver blocks
// (G*256 candidate ids, doc range [lo, hi]); a pre-pass finds, per tile and per other list,
// the block range whose ids can fall in [lo, hi] (binary search on the skip tables).
//   byte-map path (hi-lo < 16 Ki docs — dense lists): the tile keeps one LDS byte per doc of
//     its range.  Driver postings write 1; list j's postings turn a j into j+1 (lists are
//     duplicate-free, so no two lanes ever race on a byte); bytes equal to n are the result.
//     Every block is decoded by one wave (4 bytes per lane, one DPP prefix sum), so a posting
//     costs one LDS byte access and nothing is sorted, merged or searched.  The tile leaves its
//     result as a BITMAP (1 bit per doc of its range, tombstones already cleared) plus a count.
//   gallop path (sparse / skewed tiles): candidates sit in LDS as a sorted array; for each
//     other list every live candidate binary-searches that list's skip table for the one
//     block that could hold it, a wave decodes just those blocks into LDS and the
//     candidates search them.  Blocks without a candidate are never touched.  The tile leaves
//     its survivors as a short id list plus a count.
// Output order.  Tiles are in doc order and independent of each other (no inter-workgroup
// hand-off at all): a scan of the tile counts gives every tile its output offset and an
// expand pass turns bitmaps / lists into the final ascending id array.  Intermediate traffic
// is 1 bit per doc of the driver's range instead of a second copy of the ids.
#include <algorithm>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t LIST_FLAG = 0x80000000u;       // tile_count: the tile left an id list, not a bitmap
constexpr uint32_t MAP_BYTES = ISECT_SMAX + 32u;  // byte map origin is rounded down to a multiple of 32 docs

// ---- pre-pass: tile doc ranges and per-list phase descriptors ----------------------------
// desc layout per tile (2 + 4n words): [lo, hi] then per list j (0 = driver)
// {bl, bh, qlo, qhi}: block range and payload byte range of the phase (tile, j).
__host__ __device__ constexpr uint32_t desc_stride(uint32_t n) { return 2u + 4u * n; }

// Two upper bounds on one skip table at once, each searched 64 ways per round by the whole wave:
// r = first index in [0, n) with first_doc > x.  The two probe sequences are independent, so their
// loads overlap and the pair costs the latency of one search (3 rounds for 250k blocks).
__device__ __forceinline__ void wave_skip_upper_bound2(const ii2_skip *__restrict__ skip, uint32_t n, uint32_t xa, uint32_t xb,
                                                       uint32_t &ra, uint32_t &rb) {
    const uint32_t l = (uint32_t)lane_id();
    uint32_t loa = 0, hia = n, lob = 0, hib = n;
    bool donea = false, doneb = false;
    while (!(donea && doneb)) {
        const uint32_t spa = hia - loa, spb = hib - lob;
        const uint32_t sta = (spa + 63u) >> 6, stb = (spb + 63u) >> 6;
        const uint32_t pa = loa + l * sta, pb = lob + l * stb;
        const bool ina = !donea && pa < hia, inb = !doneb && pb < hib;
        const uint32_t fa = ina ? skip[pa].first_doc : 0u;
        const uint32_t fb = inb ? skip[pb].first_doc : 0u;
        if (!donea) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(ina && fa <= xa));
            const uint32_t nin = (uint32_t)__popcll(__ballot(ina));
            if (spa == 0u) { donea = true; }
            else if (sta == 1u) { loa = cnt < nin ? loa + cnt : hia; hia = loa; donea = true; }
            else {
                const uint32_t nlo = cnt ? loa + (cnt - 1u) * sta + 1u : loa;
                hia = cnt < nin ? loa + cnt * sta : hia;
                loa = nlo;
            }
        }
        if (!doneb) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(inb && fb <= xb));
            const uint32_t nin = (uint32_t)__popcll(__ballot(inb));
            if (spb == 0u) { doneb = true; }
            else if (stb == 1u) { lob = cnt < nin ? lob + cnt : hib; hib = lob; doneb = true; }
            else {
                const uint32_t nlo = cnt ? lob + (cnt - 1u) * stb + 1u : lob;
                hib = cnt < nin ? lob + cnt * stb : hib;
                lob = nlo;
            }
        }
    }
    ra = loa;
    rb = lob;
}

__global__ __launch_bounds__(256) void k_isect_partition(IntersectParams p) {
    const uint64_t gw = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // one wave per (tile, list)
    const uint32_t n = p.n_lists;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < p.n_sums) p.sums[gid] = 0;                      // level sums of the tile counts (filled by the tile kernel)
    if (p.op_union) {
        // union: fixed doc-range tiles, one wave per (tile, list)
        if (gw >= (uint64_t)p.n_tiles * n) return;
        const uint32_t t = (uint32_t)(gw / n), j = (uint32_t)(gw % n);
        const uint64_t lo64 = (uint64_t)p.u_base + (uint64_t)t * p.u_span;
        uint64_t hi64 = lo64 + p.u_span - 1u;
        if (hi64 > p.u_max) hi64 = p.u_max;
        const uint32_t lo = (uint32_t)lo64, hi = (uint32_t)hi64;
        uint32_t *r = p.ranges + (uint64_t)t * desc_stride(n);
        if (j == 0u && lane_id() == 0) { r[0] = lo; r[1] = hi; }
        const ListView L = p.lists[j];
        uint32_t ub, bh;
        wave_skip_upper_bound2(L.skip, L.nblk, lo, hi, ub, bh);
        const uint32_t bl = ub ? ub - 1u : 0u;
        if (bh < bl) bh = bl;
        if (lane_id() == 0) {
            r[2 + 4 * j] = bl;
            r[3 + 4 * j] = bh;
            r[4 + 4 * j] = L.skip[bl].byte_off;
            r[5 + 4 * j] = L.skip[bh].byte_off;
        }
        return;
    }
    // n - 1 waves per tile (one per non-driver list); the first of them also writes the driver's descriptor
    const uint32_t m = n > 1u ? n - 1u : 1u;
    if (gw >= (uint64_t)p.n_tiles * m) return;
    const uint32_t t = (uint32_t)(gw / m), j = (uint32_t)(gw % m) + 1u;
    const ListView d = p.lists[0];
    const uint32_t b0 = (p.sub > 1u ? t / p.sub : t) * p.G;   // sub > 1 (tiny drivers, G = 1): several tiles share a driver block
    const uint32_t b1 = b0 + p.G < d.nblk ? b0 + p.G : d.nblk;
    const uint32_t lo = d.skip[b0].first_doc;
    const uint32_t hi = b1 < d.nblk ? d.skip[b1].first_doc - 1u : *d.last_doc;
    uint32_t *r = p.ranges + (uint64_t)t * desc_stride(n);
    if (j == 1u && lane_id() == 0) {
        r[0] = lo; r[1] = hi;
        r[2] = b0; r[3] = b1; r[4] = d.skip[b0].byte_off; r[5] = d.skip[b1].byte_off;
    }
    if (j >= n) return;                         // single-list query
    const ListView L = p.lists[j];
    // first block that may hold ids >= lo: the last block whose first_doc <= lo
    uint32_t ub, bh;                            // bh: first block starting after hi
    wave_skip_upper_bound2(L.skip, L.nblk, lo, hi, ub, bh);
    const uint32_t bl = ub ? ub - 1u : 0u;
    if (bh < bl) bh = bl;
    if (lane_id() == 0) {
        r[2 + 4 * j] = bl;
        r[3 + 4 * j] = bh;
        r[4 + 4 * j] = L.skip[bl].byte_off;
        r[5 + 4 * j] = L.skip[bh].byte_off;
    }
}

// ---- the tile kernel ----------------------------------------------------------------------
// Workgroup w walks tiles w, w+grid, ... so that it can fetch its next tile's bytes while it
// works on the current one.
//
// Software pipeline.  A phase = (tile, list).  Its payload bytes and skip entries are
// contiguous in HBM, so all 256 threads fetch them with 16-byte loads into registers one
// phase AHEAD (while the previous phase decodes out of LDS), then park them in LDS; the
// waves decode from LDS.  Phases too large for the staging buffer decode straight from HBM.
constexpr uint32_t RAWCAP = 8192;     // staged payload bytes per phase (2 x uint4 per thread)
constexpr uint32_t SKIPCAP = 256;     // staged skip entries per phase (1 per thread)
constexpr uint32_t DESC_WORDS = 2u + 4u * MAX_LISTS;

// RAW: staged payload bytes per phase; DW: descriptor words per tile (both smaller in the two-list instantiation,
// which then fits six workgroups per CU)
template <uint32_t RAW, uint32_t DW>
struct __align__(16) IsectSmemT {
    uint8_t map[MAP_BYTES];                  // byte map | gallop: cand[GMAX*256] u32 + hit[GMAX*256] u8
    uint8_t raw[RAW + 32];                   // staged payload | gallop: 4 x 256 decoded block (one per wave)
    ii2_skip skipbuf[SKIPCAP + 8];
    uint32_t desc[2][DW];
    uint32_t wcnt[4];
    uint32_t ncand;
    uint32_t pad;
};
constexpr uint32_t BM_MAXL = 7;            // lists a bitmap-mode tile can hold (7 x 522 words fit the map)
constexpr uint32_t GALLOP_SUB = 8;         // driver blocks the gallop path handles per round
static_assert(GALLOP_SUB * 256u * 5u <= MAP_BYTES, "gallop candidates + flags must fit the byte map");

struct Phase { uint32_t bl, bh, qlo, qhi; };
template <uint32_t RAW>
__device__ __forceinline__ bool can_stage_t(const Phase &d) {
    return d.bh > d.bl && d.bh - d.bl < SKIPCAP && d.qhi - (d.qlo & ~15u) <= RAW;
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
template <class Smem>
__device__ __forceinline__ void prefetch_commit(Smem &sm, const Prefetch &pf, const Phase &d, int tid) {
    const uint32_t base16 = d.qlo & ~15u;
    const uint32_t nch = (d.qhi - base16 + 15u) >> 4;
    if ((uint32_t)tid < nch) *reinterpret_cast<uint4 *>(&sm.raw[16u * (uint32_t)tid]) = pf.r0;
    if ((uint32_t)tid + 256u < nch) *reinterpret_cast<uint4 *>(&sm.raw[16u * ((uint32_t)tid + 256u)]) = pf.r1;
    if ((uint32_t)tid <= d.bh - d.bl) sm.skipbuf[tid] = pf.sk;
}

// bytes of w equal to the byte replicated in n4 -> 4-bit mask (bit k = byte k)
__device__ __forceinline__ uint32_t bytes_eq_mask(uint32_t w, uint32_t n4) {
    const uint32_t x = w ^ n4;                                   // zero byte <=> match
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    const uint32_t z = ~(t | x | 0x7F7F7F7Fu);                   // 0x80 in every zero byte, exact
    return (((z >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// bit 7 of every byte of w -> 4-bit mask (bit k = byte k)
__device__ __forceinline__ uint32_t bytes_msb_mask(uint32_t w) {
    return ((((w & 0x80808080u) >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// ---- gallop helpers: one block of a long list held in registers (up to 5 x 256 payload bytes, 4 per lane) -------
struct GallopBlock { uint32_t cur, q0, q1, first; uint32_t w[5]; };
struct RegBytes5 {
    uint32_t w0, w1, w2, w3, w4, q0;
    __device__ __forceinline__ uint32_t operator()(uint32_t myq) const {      // myq = q0 + 256 k + 4 lane
        const uint32_t k = (myq - q0) >> 8;                                   // wave-uniform
        return k == 0u ? w0 : k == 1u ? w1 : k == 2u ? w2 : k == 3u ? w3 : w4;
    }
};
// fetch the payload of lane `leader`'s block (its skip entries are already in ea / eb of that lane)
__device__ __forceinline__ void gallop_fetch(GallopBlock &g, const uint8_t *__restrict__ payload, uint32_t blk, const ii2_skip &ea,
                                             const ii2_skip &eb, int leader) {
    g.cur = wave_bcast(blk, leader);
    g.q0 = wave_bcast(ea.byte_off, leader);
    g.q1 = wave_bcast(eb.byte_off, leader);
    g.first = wave_bcast(ea.first_doc, leader);
    const uint32_t l4 = 4u * (uint32_t)lane_id();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t q = g.q0 + 256u * (uint32_t)k + l4;
        g.w[k] = q < g.q1 ? load_u32_unaligned(payload + q) : 0u;
    }
}

// WIDE: 64 lists — 258 descriptor words per tile, two more than threads (kept out of the common instantiation:
// the kernel sits at its register limit and even two extra loads cost 2.5 % there)
// NFIX: list count known at compile time (0 = read it from the parameters) — the two-term query is by far the most common
// UNION: ids of ANY list (every list marks like the driver, the finalise tests for a mark) — dense unions, host-selected
// SUBT: driver blocks are split over several tiles (p.sub > 1; gallop only)
template <bool WIDE, uint32_t NFIX, bool UNION, bool SUBT = false>
__global__ __launch_bounds__(256, 5) void k_isect_tiles(IntersectParams p) {
    constexpr uint32_t RAW = NFIX == 2u ? 7680u : RAWCAP;
    constexpr uint32_t DW = NFIX ? 2u + 4u * NFIX : DESC_WORDS;
    __shared__ IsectSmemT<RAW, DW> sm;
    auto can_stage = [](const Phase &d) { return can_stage_t<RAW>(d); };
    // byte/bit map tiles pay per DOC of the tile's range (clear + finalise), the gallop path per driver posting: sparse
    // tiles (many docs per driver block) gallop even when their range would fit the map
    const uint32_t map_docs_per_block = p.map_docs_per_block < (1u << 27) ? p.map_docs_per_block : (1u << 27);
    constexpr bool whole_blocks = !SUBT;
    auto use_map = [&](const uint32_t *DD) { return whole_blocks && DD[1] - DD[0] < ISECT_SMAX && (UNION || DD[1] - DD[0] < map_docs_per_block * (DD[3] - DD[2])); };   // (G <= 16 blocks: no overflow below 2^28 docs per block)
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t n = NFIX ? NFIX : p.n_lists;
    const bool shiftenc = UNION || n <= 8u;
    const uint32_t stride = desc_stride(n);
    Prefetch pf;
    pf.r0 = make_uint4(0, 0, 0, 0); pf.r1 = pf.r0; pf.sk.first_doc = 0; pf.sk.byte_off = 0;

    const uint32_t n_items = p.n_tiles;
    uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    uint32_t tile = item;
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
    if (WIDE && (uint32_t)tid + 256u < stride) sm.desc[0][256u + tid] = p.ranges[(uint64_t)tile * stride + 256u + tid];
    __syncthreads();
    {
        const uint32_t *D = sm.desc[0];
        const Phase d0 = {D[2], D[3], D[4], D[5]};
        if (use_map(D) && can_stage(d0)) prefetch_issue(pf, d0, p.lists[0], tid);
    }

    for (uint32_t it = 0; item < n_items; it++) {
        const uint32_t *D = sm.desc[it & 1u];
        const uint32_t lo = D[0], hi = D[1];
        const uint32_t span = hi - lo;
        const uint32_t next_item = item + gridDim.x;
        const bool has_next = next_item < n_items;
        const uint32_t next_tile = next_item;
        // descriptors of this workgroup's next tile: one word per thread, in flight during the tile
        uint32_t dreg = 0;
        if (has_next && (uint32_t)tid < stride) dreg = p.ranges[(uint64_t)next_tile * stride + tid];
        const ListView drv = p.lists[0];
        uint32_t *slot = p.tmp + (uint64_t)tile * p.slot_words;     // this tile's bitmap words / id list
        bool map_tile = false;                                      // a map tile: its count is published after the end barrier

        // Very dense tiles (every phase: full blocks of single-byte gaps, at most ~3 docs per posting) keep one
        // BITMAP per list instead of the byte map: a lane turns four gap bytes into a 32-bit mask in registers
        // ((M << gap) | 1, one instruction per posting) and ORs it into its list's bitmap with two LDS atomics per
        // four postings; the result is the AND of the bitmaps.  Purely a fast path: any block is still decoded right.
        const bool mapped = use_map(D);          // (use_map is false for every tile when driver blocks are split: p.sub > 1)
        bool bm = mapped && n <= BM_MAXL && p.bitmap_mode != 0u;
        for (uint32_t j = 0; j < n && bm; j++) {
            const uint32_t nb = D[3 + 4 * j] - D[2 + 4 * j], bytes = D[5 + 4 * j] - D[4 + 4 * j];
            bm = nb > 0u && bytes == 255u * nb && (uint64_t)bytes * 13u >= (uint64_t)span * 4u;
        }
        if (bm) {
            // ================= bitmap path =================
            const uint32_t mlo = lo & ~31u;
            const uint32_t mspan = hi - mlo;
            const uint32_t nwords = (mspan >> 5) + 1u;
            constexpr uint32_t GU = 128u;                    // guard bits below and above the tile's range
            const uint32_t bstride = nwords + 9u;            // words per list bitmap: 4 guard + nwords + 4 guard + 1 spill
            const uint32_t lim = mspan + 252u;               // highest (guard-shifted) position ever written
            uint32_t *bmw = reinterpret_cast<uint32_t *>(sm.map);
            for (uint32_t i = (uint32_t)tid; 4u * i < n * bstride; i += 256u)
                reinterpret_cast<uint4 *>(bmw)[i] = make_uint4(0, 0, 0, 0);
            for (uint32_t j = 0; j < n; j++) {
                const ListView L = p.lists[j];
                const Phase d = {D[2 + 4 * j], D[3 + 4 * j], D[4 + 4 * j], D[5 + 4 * j]};
                const bool staged = can_stage(d);
                if (j == n - 1u && has_next && (uint32_t)tid < stride) sm.desc[(it + 1u) & 1u][tid] = dreg;
                // 64 lists: 258 descriptor words, two more than threads — fetched here, latency exposed (rare)
                if (WIDE && j == n - 1u && has_next && (uint32_t)tid + 256u < stride) sm.desc[(it + 1u) & 1u][256u + tid] = p.ranges[(uint64_t)next_tile * stride + 256u + tid];
                if (staged) prefetch_commit(sm, pf, d, tid);
                II2_STAMP(4)
                lds_barrier();
                if (j + 1u < n) {
                    const Phase dn = {D[6 + 4 * j], D[7 + 4 * j], D[8 + 4 * j], D[9 + 4 * j]};
                    if (can_stage(dn)) prefetch_issue(pf, dn, p.lists[j + 1u], tid);
                } else if (has_next) {
                    const uint32_t *DN = sm.desc[(it + 1u) & 1u];
                    const Phase dn = {DN[2], DN[3], DN[4], DN[5]};
                    if (use_map(DN) && can_stage(dn)) prefetch_issue(pf, dn, drv, tid);
                }
                II2_STAMP(0)
                const uint32_t nblk = d.bh - d.bl;
                uint32_t *bmj = bmw + j * bstride;
                auto setbit = [&](uint32_t id, bool valid) {
                    const uint32_t pos = id - mlo + GU;
                    if (valid && pos <= lim) atomicOr(&bmj[pos >> 5], 1u << (pos & 31u));
                };
                auto mark4b = [&](uint32_t, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                    setbit(id0, mask & 1u); setbit(id1, mask & 2u); setbit(id2, mask & 4u); setbit(id3, mask & 8u);
                };
                const uint32_t rl = (uint32_t)l & 15u, row = (uint32_t)l >> 4;
                auto mark16b = [&](uint32_t base, const uint4 &w, bool rv, uint32_t first_doc) {
                    const uint32_t u = base - mlo + GU;              // (guard-shifted) position of the posting before my bytes
                    const bool ok = rv && u <= mspan + GU;           // else the lane lies wholly outside the tile's range
                    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
                    uint32_t ss[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) ss[g] = __builtin_amdgcn_sad_u8(ww[g], 0u, 0u);
                    const uint32_t smax = max(max(ss[0], ss[1]), max(ss[2], ss[3]));
                    const bool fast = ok && smax <= 31u;             // four postings fit one 32-bit mask
                    const uint32_t seed = fast ? 1u : 0u;
                    uint32_t q = u;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const uint32_t x = ww[g];
                        uint32_t M = (seed << ((x >> 24) & 31u)) | seed;        // postings 3, 2 of the group
                        M = (M << ((x >> 16) & 31u)) | seed;                    // posting 1
                        M = (M << ((x >> 8) & 31u)) | seed;                     // posting 0 (bit 0)
                        uint32_t P = q + (x & 0xFFu);
                        P = P < lim ? P : lim;
                        q += ss[g];
                        const unsigned long long MM = (unsigned long long)M << (P & 31u);
                        uint32_t *dst = bmj + (P >> 5);
                        atomicOr(dst, (uint32_t)MM);
                        if ((uint32_t)(MM >> 32)) atomicOr(dst + 1, (uint32_t)(MM >> 32));     // LDS atomics cost per active lane
                    }
                    // lanes with a wide group walk their postings one by one if they can reach the tile's range — also a lane
                    // that STARTS more than the guard below the range (not ok) and jumps into it; a lane of narrow groups
                    // spans < 128 docs, so one that is not ok lies wholly outside
                    if (__ballot(rv && smax > 31u) != 0ull) {          // rare: the test for reaching the range is only made here
                        const bool slow = rv && smax > 31u && (ok || (u + (ss[0] + ss[1] + ss[2] + ss[3])) < u);   // second case: the sum wrapped past 2^32
                        uint32_t pp = u;
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            pp += (ww[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                            if (slow && pp <= lim) atomicOr(&bmj[pp >> 5], 1u << (pp & 31u));
                        }
                    }
                    setbit(first_doc, rv && rl == 0u);
                };
                const uint32_t ngroups = (nblk + 3u) >> 2;
                if (staged) {
                    const uint32_t base16 = d.qlo & ~15u;
                    for (uint32_t g = (uint32_t)wv; g < ngroups; g += 4u) {
                        const uint32_t bi = 4u * g + row;
                        const bool rv = bi < nblk;
                        ii2_skip e0 = {0u, 0u}, e1 = {0u, 0u};
                        if (rv) { e0 = sm.skipbuf[bi]; e1 = sm.skipbuf[bi + 1u]; }
                        uint32_t base;
                        uint4 w;
                        if (decode_rows16(LdsBytes16{sm.raw}, e0.byte_off - base16, e1.byte_off - base16, e0.first_doc, rv, base, w)) {
                            mark16b(base, w, rv, e0.first_doc);
                        } else {
                            for (uint32_t i = 4u * g; i < 4u * g + 4u && i < nblk; i++) {
                                const ii2_skip f0 = sm.skipbuf[i], f1 = sm.skipbuf[i + 1u];
                                decode_block_wave4(LdsBytes{sm.raw}, f0.byte_off - base16, f1.byte_off - base16, f0.first_doc, mark4b);
                            }
                        }
                    }
                } else {
                    for (uint32_t g = (uint32_t)wv; g < ngroups; g += 4u) {
                        const uint32_t bi = 4u * g + row;
                        const bool rv = bi < nblk;
                        ii2_skip e0 = {0u, 0u}, e1 = {0u, 0u};
                        if (rv) { e0 = L.skip[d.bl + bi]; e1 = L.skip[d.bl + bi + 1u]; }
                        uint32_t base;
                        uint4 w;
                        if (decode_rows16(GlobalBytes16{L.payload}, e0.byte_off, e1.byte_off, e0.first_doc, rv, base, w)) {
                            mark16b(base, w, rv, e0.first_doc);
                        } else {
                            for (uint32_t b = d.bl + 4u * g; b < d.bl + 4u * g + 4u && b < d.bh; b++)
                                decode_block_wave4(GlobalBytes{L.payload}, L.skip[b].byte_off, L.skip[b + 1].byte_off,
                                                   L.skip[b].first_doc, mark4b);
                        }
                    }
                }
                II2_STAMP(1)
                lds_barrier();
                II2_STAMP(2)
            }
            // finalise: AND of the list bitmaps, tombstones cleared, survivors counted
            uint32_t mine = 0;
            for (uint32_t wi = (uint32_t)tid; wi < nwords; wi += 256u) {
                uint32_t word = bmw[GU / 32u + wi];
                for (uint32_t j = 1; j < n; j++) word = UNION ? (word | bmw[j * bstride + GU / 32u + wi]) : (word & bmw[j * bstride + GU / 32u + wi]);
                if (wi == nwords - 1u && (mspan & 31u) != 31u) word &= (2u << (mspan & 31u)) - 1u;
                if (p.tomb) {
                    const uint32_t tw = (mlo >> 5) + wi;
                    if (tw < p.tomb_nwords) word &= ~p.tomb[tw];
                }
                slot[wi] = word;
                mine += (uint32_t)__popc(word);
            }
            mine = wave_sum(mine);
            if (l == 0) sm.wcnt[wv] = mine;        // summed up after the end-of-tile barrier (one barrier fewer per tile)
            map_tile = true;
            II2_STAMP(3)
        } else if (mapped) {
            // ================= byte-map path =================
            const uint32_t mlo = lo & ~31u;                  // map origin: 32-doc aligned
            const uint32_t mspan = hi - mlo;                 // last valid map offset
            const uint32_t nwords = (mspan >> 5) + 1u;       // <= MAP_BYTES / 32
            for (uint32_t i = (uint32_t)tid * 16u; i < nwords * 32u; i += 256u * 16u)
                *reinterpret_cast<uint4 *>(&sm.map[i]) = make_uint4(0, 0, 0, 0);
            for (uint32_t j = 0; j < n; j++) {
                const ListView L = p.lists[j];
                const Phase d = {D[2 + 4 * j], D[3 + 4 * j], D[4 + 4 * j], D[5 + 4 * j]};
                const bool staged = can_stage(d);
                if (j == n - 1u && has_next && (uint32_t)tid < stride) sm.desc[(it + 1u) & 1u][tid] = dreg;
                // 64 lists: 258 descriptor words, two more than threads — fetched here, latency exposed (rare)
                if (WIDE && j == n - 1u && has_next && (uint32_t)tid + 256u < stride) sm.desc[(it + 1u) & 1u][256u + tid] = p.ranges[(uint64_t)next_tile * stride + 256u + tid];
                if (staged) prefetch_commit(sm, pf, d, tid);
                II2_STAMP(4)      // clear + commit (waits for the prefetched bytes)
                lds_barrier();
                // fetch the next phase while this one decodes
                if (j + 1u < n) {
                    const Phase dn = {D[6 + 4 * j], D[7 + 4 * j], D[8 + 4 * j], D[9 + 4 * j]};
                    if (can_stage(dn)) prefetch_issue(pf, dn, p.lists[j + 1u], tid);
                } else if (has_next) {
                    const uint32_t *DN = sm.desc[(it + 1u) & 1u];
                    const Phase dn = {DN[2], DN[3], DN[4], DN[5]};
                    if (use_map(DN) && can_stage(dn)) prefetch_issue(pf, dn, drv, tid);
                }
                II2_STAMP(0)      // clear + commit + barrier + prefetch issue
                const uint32_t nblk = d.bh - d.bl;
                const uint8_t want = UNION ? (uint8_t)0 : (uint8_t)j;
                // n <= 8 (shift code): the driver writes 1 << (8 - n) and every further list shifts the bytes it
                // hits left by one, so a doc of all n lists ends at 0x80 and nothing else reaches bit 7 — no
                // compare per posting and a one-AND test when the map is finalised.  n > 8: counting code
                // (driver 1, list j turns j into j + 1).
                const uint8_t s0 = UNION ? (uint8_t)0x80 : shiftenc ? (uint8_t)(1u << (8u - n)) : (uint8_t)1;
                // driver: plain stores of 1.  list j: read the four candidate bytes, then bump the ones at j.
                auto mark4 = [&](uint32_t, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                    const uint32_t o0 = id0 - mlo, o1 = id1 - mlo, o2 = id2 - mlo, o3 = id3 - mlo;
                    const bool v0 = (mask & 1u) && o0 <= mspan, v1 = (mask & 2u) && o1 <= mspan;
                    const bool v2 = (mask & 4u) && o2 <= mspan, v3 = (mask & 8u) && o3 <= mspan;
                    if (want == 0) {
                        if (v0) sm.map[o0] = s0;
                        if (v1) sm.map[o1] = s0;
                        if (v2) sm.map[o2] = s0;
                        if (v3) sm.map[o3] = s0;
                    } else if (shiftenc) {
                        const uint8_t m0 = v0 ? sm.map[o0] : (uint8_t)0, m1 = v1 ? sm.map[o1] : (uint8_t)0;
                        const uint8_t m2 = v2 ? sm.map[o2] : (uint8_t)0, m3 = v3 ? sm.map[o3] : (uint8_t)0;
                        if (v0) sm.map[o0] = (uint8_t)(m0 << 1);
                        if (v1) sm.map[o1] = (uint8_t)(m1 << 1);
                        if (v2) sm.map[o2] = (uint8_t)(m2 << 1);
                        if (v3) sm.map[o3] = (uint8_t)(m3 << 1);
                    } else {
                        const uint8_t m0 = v0 ? sm.map[o0] : (uint8_t)0xFE, m1 = v1 ? sm.map[o1] : (uint8_t)0xFE;
                        const uint8_t m2 = v2 ? sm.map[o2] : (uint8_t)0xFE, m3 = v3 ? sm.map[o3] : (uint8_t)0xFE;
                        const uint8_t nx = (uint8_t)(want + 1u);
                        if (m0 == want) sm.map[o0] = nx;
                        if (m1 == want) sm.map[o1] = nx;
                        if (m2 == want) sm.map[o2] = nx;
                        if (m3 == want) sm.map[o3] = nx;
                    }
                };
                // four blocks per wave round: one 16-lane row per block, 16 gap bytes per lane;
                // all 17 candidate bytes of a lane are read before any is written
                const uint32_t dummy = mspan + 1u;           // harmless byte: masked out when the map is finalised
                const uint32_t rl = (uint32_t)l & 15u, row = (uint32_t)l >> 4;
                auto mark16 = [&](uint32_t base, const uint4 &w, bool rv, uint32_t first_doc) {
                    uint32_t acc = rv ? base - mlo : 0xFFFFFFFFu;
                    uint32_t o[16];
                    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        acc += (ww[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                        o[k] = acc < dummy ? acc : dummy;
                    }
                    const uint32_t f = first_doc - mlo;
                    const uint32_t of = (rv && rl == 0u && f < dummy) ? f : dummy;
                    if (want == 0) {
#pragma unroll
                        for (int k = 0; k < 16; k++) sm.map[o[k]] = s0;
                        sm.map[of] = s0;
                    } else if (shiftenc) {
                        uint8_t m[16];
#pragma unroll
                        for (int k = 0; k < 16; k++) m[k] = sm.map[o[k]];
                        const uint8_t mf = sm.map[of];
                        // bytes hit twice (zero-gap padding repeats the lane's last id) were read before any write: same value
#pragma unroll
                        for (int k = 0; k < 16; k++) sm.map[o[k]] = (uint8_t)(m[k] << 1);
                        sm.map[of] = (uint8_t)(mf << 1);
                    } else {
                        uint8_t m[16];
#pragma unroll
                        for (int k = 0; k < 16; k++) m[k] = sm.map[o[k]];
                        const uint8_t mf = sm.map[of];
                        const uint8_t nx = (uint8_t)(want + 1u);
                        // no exec-mask juggling: lanes without a match store onto the dummy byte instead
#pragma unroll
                        for (int k = 0; k < 16; k++) sm.map[m[k] == want ? o[k] : dummy] = nx;
                        sm.map[mf == want ? of : dummy] = nx;
                    }
                };
                const uint32_t ngroups = (nblk + 3u) >> 2;
                if (staged) {
                    const uint32_t base16 = d.qlo & ~15u;
                    for (uint32_t g = (uint32_t)wv; g < ngroups; g += 4u) {
                        const uint32_t bi = 4u * g + row;
                        const bool rv = bi < nblk;
                        ii2_skip e0 = {0u, 0u}, e1 = {0u, 0u};
                        if (rv) { e0 = sm.skipbuf[bi]; e1 = sm.skipbuf[bi + 1u]; }
                        uint32_t base;
                        uint4 w;
                        if (decode_rows16(LdsBytes16{sm.raw}, e0.byte_off - base16, e1.byte_off - base16, e0.first_doc, rv, base, w)) {
                            mark16(base, w, rv, e0.first_doc);
                        } else {
                            for (uint32_t i = 4u * g; i < 4u * g + 4u && i < nblk; i++) {
                                const ii2_skip f0 = sm.skipbuf[i], f1 = sm.skipbuf[i + 1u];
                                decode_block_wave4(LdsBytes{sm.raw}, f0.byte_off - base16, f1.byte_off - base16, f0.first_doc, mark4);
                            }
                        }
                    }
                } else {
                    for (uint32_t g = (uint32_t)wv; g < ngroups; g += 4u) {
                        const uint32_t bi = 4u * g + row;
                        const bool rv = bi < nblk;
                        ii2_skip e0 = {0u, 0u}, e1 = {0u, 0u};
                        if (rv) { e0 = L.skip[d.bl + bi]; e1 = L.skip[d.bl + bi + 1u]; }
                        uint32_t base;
                        uint4 w;
                        if (decode_rows16(GlobalBytes16{L.payload}, e0.byte_off, e1.byte_off, e0.first_doc, rv, base, w)) {
                            mark16(base, w, rv, e0.first_doc);
                        } else {
                            for (uint32_t b = d.bl + 4u * g; b < d.bl + 4u * g + 4u && b < d.bh; b++)
                                decode_block_wave4(GlobalBytes{L.payload}, L.skip[b].byte_off, L.skip[b + 1].byte_off,
                                                   L.skip[b].first_doc, mark4);
                        }
                    }
                }
                II2_STAMP(1)      // decode
                lds_barrier();
                II2_STAMP(2)      // barrier after decode
            }
            // finalise: byte map -> bitmap words (32 docs each), tombstones cleared, survivors counted
            const uint32_t n4 = n * 0x01010101u;
            uint32_t mine = 0;
            for (uint32_t wi = (uint32_t)tid; wi < nwords; wi += 256u) {
                const uint4 a = *reinterpret_cast<const uint4 *>(&sm.map[32u * wi]);
                const uint4 b = *reinterpret_cast<const uint4 *>(&sm.map[32u * wi + 16u]);
                uint32_t word;
                if (shiftenc)
                    word = bytes_msb_mask(a.x) | (bytes_msb_mask(a.y) << 4) | (bytes_msb_mask(a.z) << 8) | (bytes_msb_mask(a.w) << 12) |
                           (bytes_msb_mask(b.x) << 16) | (bytes_msb_mask(b.y) << 20) | (bytes_msb_mask(b.z) << 24) | (bytes_msb_mask(b.w) << 28);
                else
                    word = bytes_eq_mask(a.x, n4) | (bytes_eq_mask(a.y, n4) << 4) | (bytes_eq_mask(a.z, n4) << 8) |
                           (bytes_eq_mask(a.w, n4) << 12) | (bytes_eq_mask(b.x, n4) << 16) | (bytes_eq_mask(b.y, n4) << 20) |
                           (bytes_eq_mask(b.z, n4) << 24) | (bytes_eq_mask(b.w, n4) << 28);
                if (wi == nwords - 1u && (mspan & 31u) != 31u) word &= (2u << (mspan & 31u)) - 1u;   // drop the dummy byte and beyond
                if (p.tomb) {
                    const uint32_t tw = (mlo >> 5) + wi;
                    if (tw < p.tomb_nwords) word &= ~p.tomb[tw];
                }
                slot[wi] = word;
                mine += (uint32_t)__popc(word);
            }
            mine = wave_sum(mine);
            if (l == 0) sm.wcnt[wv] = mine;        // summed up after the end-of-tile barrier
            map_tile = true;
            II2_STAMP(3)          // finalise
        } else {
            // ================= gallop path =================
            uint32_t *cand = reinterpret_cast<uint32_t *>(sm.map);
            uint8_t *hit = sm.map + GALLOP_SUB * 256u * 4u;
            uint32_t *fd = reinterpret_cast<uint32_t *>(sm.map + GALLOP_SUB * 256u * 5u);      // rest of the map: staged first docs
            constexpr uint32_t FDCAP = (MAP_BYTES - GALLOP_SUB * 256u * 5u) / 4u;
            uint32_t *wbuf = reinterpret_cast<uint32_t *>(sm.raw) + (uint32_t)wv * 256u;
            uint32_t total = 0;
            for (uint32_t b0 = D[2]; b0 < D[3]; b0 += GALLOP_SUB) {
                const uint32_t b1 = b0 + GALLOP_SUB < D[3] ? b0 + GALLOP_SUB : D[3];
                __syncthreads();
                for (uint32_t b = b0 + (uint32_t)wv; b < b1; b += 4u) {
                    const uint32_t pb = (b - b0) * 256u;
                    const uint32_t c = decode_block_wave(GlobalBytes{drv.payload}, drv.skip[b].byte_off, drv.skip[b + 1].byte_off,
                                                         drv.skip[b].first_doc, [&](uint32_t ix, uint32_t id) {
                                                             cand[pb + ix] = id;
                                                             hit[pb + ix] = 1;
                                                         });
                    if (b == b1 - 1u && l == 0) sm.ncand = pb + c;
                }
                __syncthreads();
                const uint32_t ncand = sm.ncand;
                if (SUBT) {
                    // this tile owns one slice of its (single) driver block: the other candidates are dead from the start
                    const uint32_t si = tile % p.sub;
                    const uint32_t c0 = (uint32_t)(((uint64_t)ncand * si) / p.sub), c1 = (uint32_t)(((uint64_t)ncand * (si + 1u)) / p.sub);
                    for (uint32_t pi = (uint32_t)tid; pi < ncand; pi += 256u)
                        if (pi < c0 || pi >= c1) hit[pi] = 0;
                    __syncthreads();
                }
                // Split driver blocks with four or more lists (round 4): the second-shortest list is tested first, as before -
                // it is what thins the candidates out - and then ALL the longer lists at once: every (surviving candidate,
                // list) pair is one test, the tests of a tile share their rounds of dependent loads (block search, skip
                // entries, payload) instead of taking them list after list.  A candidate that would have died at the third
                // list is still tested against the others - a few wasted blocks against five fewer chains of round trips.
                const bool combine = SUBT && NFIX == 0u && n >= 4u;
                for (uint32_t j = 1; j < (combine ? 2u : n); j++) {
                    const ListView L = p.lists[j];
                    const uint32_t bl = D[2 + 4 * j], bh = D[3 + 4 * j];
                    // the first docs of the blocks in range go to LDS in one coalesced fetch: a candidate's binary search
                    // then costs LDS latency per step instead of an HBM round trip (~2 us) per step
                    const bool fd_staged = bh > bl && bh - bl <= FDCAP;
                    if (fd_staged) {
                        for (uint32_t i = (uint32_t)tid; i < bh - bl; i += 256u) fd[i] = L.skip[bl + i].first_doc;
                        __syncthreads();
                    }
                    if (bh > bl && bh - bl <= 4u) {
                        // a handful of blocks in range (two sparse lists): wave w decodes block bl + w ONCE into LDS — in
                        // parallel, one fetch latency for the tile — and every candidate searches its block there
                        uint32_t *allbuf = reinterpret_cast<uint32_t *>(sm.raw);
                        if ((uint32_t)wv < bh - bl) {
                            const uint32_t b = bl + (uint32_t)wv;
                            const uint32_t cntb = decode_block_wave(GlobalBytes{L.payload}, L.skip[b].byte_off, L.skip[b + 1u].byte_off,
                                                                    L.skip[b].first_doc, [&](uint32_t ix, uint32_t id) { allbuf[(uint32_t)wv * 256u + ix] = id; });
                            if (l == 0) sm.wcnt[wv] = cntb;
                        }
                        __syncthreads();
                        for (uint32_t pi = (uint32_t)tid; pi < ncand; pi += 256u) {
                            if (hit[pi] != (uint8_t)j) continue;
                            const uint32_t c = cand[pi];
                            uint32_t a = 0, e = bh - bl;
                            while (a < e) { const uint32_t mid = a + ((e - a) >> 1); if (fd[mid] <= c) a = mid + 1u; else e = mid; }
                            if (a == 0u) continue;                         // before the first block in range
                            const uint32_t wb = a - 1u;
                            const uint32_t *ids = allbuf + wb * 256u;
                            uint32_t x = 0, y = sm.wcnt[wb];
                            const uint32_t cntb = y;
                            while (x < y) { const uint32_t mid = (x + y) >> 1; if (ids[mid] < c) x = mid + 1u; else y = mid; }
                            if (x < cntb && ids[x] == c) hit[pi] = (uint8_t)(j + 1u);
                        }
                        __syncthreads();
                        continue;
                    }
                    // split driver blocks leave a run of consecutive candidates alive: deal them out to the four waves
                    // candidate by candidate, or one wave would decode all their blocks one after the other
                    for (uint32_t base = SUBT ? 0u : (uint32_t)wv * 64u; base < ncand; base += 256u) {
                        const uint32_t pi = SUBT ? base + 4u * (uint32_t)l + (uint32_t)wv : base + (uint32_t)l;
                        const bool alive = pi < ncand && hit[pi] == (uint8_t)j;
                        const uint32_t c = alive ? cand[pi] : 0u;
                        uint32_t blk = NONE;
                        if (alive && bl < bh) {
                            uint32_t ub;
                            if (fd_staged) {
                                uint32_t a = 0, e = bh - bl;
                                while (a < e) { const uint32_t mid = a + ((e - a) >> 1); if (fd[mid] <= c) a = mid + 1u; else e = mid; }
                                ub = bl + a;
                            } else {
                                // a long list (thousands of blocks in the tile's doc range) against a handful of candidates: a plain
                                // bisection of its skip table in HBM is a dozen dependent round trips per candidate (~20 us a list, the
                                // largest part of a tile's life on BASELINE configs[4]); posting lists are close to uniform over a
                                // tile's range, so a linear guess + a short walk finds the block in two to four
                                auto get = [&](uint32_t jj) { return L.skip[jj].first_doc; };
                                ub = upper_bound_guess(get, bl, bh, c, D[0], D[1]);
                            }
                            if (ub > bl) blk = ub - 1u;
                        }
                        if (NFIX == 0u && bh - bl >= 16u) {       // many blocks in range: candidates mostly hit distinct ones
                        // Every live lane fetches the skip entries of ITS block at once (one round trip for the wave instead of
                        // one per block).  Then FOUR candidates' blocks per round, one per row of 16 lanes (round 4): a wave that
                        // decodes one block at a time for one candidate is a chain of dependent fetch + decode steps - thirteen of
                        // them per wave against the second-shortest list of BASELINE configs[4], two against each longer one -
                        // and the tile's whole life is that chain.  A row walks its block 16 bytes per lane (any gap width) and
                        // only looks for its candidate: nothing is written to LDS.
                        ii2_skip ea = {0u, 0u}, eb = {0u, 0u};
                        if (blk != NONE) { ea = L.skip[blk]; eb = L.skip[blk + 1u]; }
                        unsigned long long pending = __ballot(blk != NONE);
                        const uint32_t rw = (uint32_t)l >> 4;
                        while (pending) {
                            int src[4];
                            unsigned long long m = pending;
#pragma unroll
                            for (int g4 = 0; g4 < 4; g4++) {
                                src[g4] = m ? __ffsll((long long)m) - 1 : -1;
                                if (m) m &= m - 1ull;
                            }
                            const int sr = rw == 0u ? src[0] : rw == 1u ? src[1] : rw == 2u ? src[2] : src[3];
                            const bool rowv = sr >= 0;
                            const int ss = rowv ? sr : 0;
                            const uint32_t rc = (uint32_t)__shfl((int)c, ss, 64);
                            const uint32_t rq0 = (uint32_t)__shfl((int)ea.byte_off, ss, 64), rq1 = (uint32_t)__shfl((int)eb.byte_off, ss, 64);
                            const uint32_t rf = (uint32_t)__shfl((int)ea.first_doc, ss, 64);
                            bool found = false;
                            decode_rows16_any(L.payload, rq0, rq1, rf, rowv, [&](uint32_t, uint32_t id) { found = found || id == rc; });
                            const unsigned long long fm = __ballot(found);
#pragma unroll
                            for (int g4 = 0; g4 < 4; g4++)
                                if (l == src[g4] && ((fm >> (16 * g4)) & 0xFFFFull) != 0ull) hit[pi] = (uint8_t)(j + 1u);
                            pending = m;
                        }
                        } else {       // the dense two-list instantiation keeps the plain loop: it sits at its register limit
                        unsigned long long pending = __ballot(blk != NONE);
                        while (pending) {
                            const int leader = __ffsll((long long)pending) - 1;
                            const uint32_t cur = wave_bcast(blk, leader);
                            const uint32_t cnt = decode_block_wave(GlobalBytes{L.payload}, L.skip[cur].byte_off,
                                                                   L.skip[cur + 1].byte_off, L.skip[cur].first_doc,
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
                    }
                    __syncthreads();
                }
                if (combine) {
                    uint16_t *al = reinterpret_cast<uint16_t *>(fd);                  // surviving candidates (the staging area is free: nothing is staged here)
                    uint32_t *okc = fd + 128u;                                        // per survivor: lists that hold it (a split driver block: <= 256 candidates)
                    uint32_t *na_p = okc + 256u;
                    static_assert(128u + 256u + 1u <= FDCAP, "survivor list + counters fit the staging area");
                    if (tid == 0) *na_p = 0u;
                    __syncthreads();
                    for (uint32_t pi = (uint32_t)tid; pi < ncand; pi += 256u)
                        if (hit[pi] == 2u) { const uint32_t z = atomicAdd(na_p, 1u); al[z] = (uint16_t)pi; okc[z] = 0u; }
                    __syncthreads();
                    const uint32_t na = *na_p, nl = n > 2u ? n - 2u : 1u, ntest = na * nl;
                    const uint32_t rw = (uint32_t)l >> 4;
                    for (uint32_t t0 = 0; t0 < ntest; t0 += 256u) {
                        const uint32_t t = t0 + 4u * (uint32_t)l + (uint32_t)wv;           // dealt out to the four waves test by test: a handful of tests must not all land in one wave
                        const bool tv = t < ntest;
                        const uint32_t ai = tv ? t / nl : 0u, jj = 2u + (tv ? t % nl : 0u);
                        const uint32_t c = tv ? cand[al[ai]] : 0u;
                        const ii2_skip *__restrict__ sk = p.lists[jj].skip;
                        const uint8_t *pay = p.lists[jj].payload;
                        const uint32_t bl = D[2 + 4 * jj], bh = D[3 + 4 * jj];
                        uint32_t blk = NONE;
                        if (tv && bl < bh) {
                            auto get = [&](uint32_t x) { return sk[x].first_doc; };
                            const uint32_t ub = upper_bound_guess(get, bl, bh, c, D[0], D[1]);
                            if (ub > bl) blk = ub - 1u;
                        }
                        ii2_skip ea = {0u, 0u}, eb = {0u, 0u};
                        if (blk != NONE) { ea = sk[blk]; eb = sk[blk + 1u]; }
                        const unsigned long long pay64 = (unsigned long long)(uintptr_t)pay;
                        unsigned long long pending = __ballot(blk != NONE);
                        while (pending) {
                            int src[4];
                            unsigned long long m = pending;
#pragma unroll
                            for (int g4 = 0; g4 < 4; g4++) {
                                src[g4] = m ? __ffsll((long long)m) - 1 : -1;
                                if (m) m &= m - 1ull;
                            }
                            const int sr = rw == 0u ? src[0] : rw == 1u ? src[1] : rw == 2u ? src[2] : src[3];
                            const bool rowv = sr >= 0;
                            const int ss = rowv ? sr : 0;
                            const uint32_t rc = (uint32_t)__shfl((int)c, ss, 64);
                            const uint32_t rq0 = (uint32_t)__shfl((int)ea.byte_off, ss, 64), rq1 = (uint32_t)__shfl((int)eb.byte_off, ss, 64);
                            const uint32_t rf = (uint32_t)__shfl((int)ea.first_doc, ss, 64);
                            const uint8_t *rp = (const uint8_t *)(uintptr_t)(unsigned long long)__shfl((long long)pay64, ss, 64);
                            bool found = false;
                            decode_rows16_any(rp, rq0, rq1, rf, rowv, [&](uint32_t, uint32_t id) { found = found || id == rc; });
                            const unsigned long long fm = __ballot(found);
#pragma unroll
                            for (int g4 = 0; g4 < 4; g4++)
                                if (l == src[g4] && ((fm >> (16 * g4)) & 0xFFFFull) != 0ull) atomicAdd(&okc[ai], 1u);
                            pending = m;
                        }
                    }
                    __syncthreads();
                    for (uint32_t z = (uint32_t)tid; z < na; z += 256u)
                        if (okc[z] == nl) hit[al[z]] = (uint8_t)n;
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
                        if (keep && p.tomb) {
                            const uint32_t v = cand[pi], w = v >> 5;
                            if (w < p.tomb_nwords && ((p.tomb[w] >> (v & 31u)) & 1u)) keep = false;
                        }
                        if (keep && pi > 0 && cand[pi - 1] == cand[pi]) keep = false;   // tolerate a duplicated driver id
                        hit[pi] = keep ? 0xFF : 0;
                        mine += keep;
                    }
                }
                mine = wave_sum(mine);
                if (l == 0) sm.wcnt[wv] = mine;
                __syncthreads();
                uint32_t pos = total;
                for (int w = 0; w < wv; w++) pos += sm.wcnt[w];
                for (uint32_t base = w0; base < w1; base += 64u) {
                    const uint32_t pi = base + (uint32_t)l;
                    const uint32_t m = (pi < w1 && hit[pi]) ? 1u : 0u;
                    const uint32_t incl = wave_incl_scan(m);
                    if (m) slot[pos + incl - 1u] = cand[pi];
                    pos += wave_bcast(incl, 63);
                }
                total += sm.wcnt[0] + sm.wcnt[1] + sm.wcnt[2] + sm.wcnt[3];
            }
            if (tid == 0) {
                p.tile_count[tile] = total | LIST_FLAG;
            }
            // restart the pipeline for the next tile
            if (has_next && (uint32_t)tid < stride) sm.desc[(it + 1u) & 1u][tid] = dreg;
            if (WIDE && has_next && (uint32_t)tid + 256u < stride) sm.desc[(it + 1u) & 1u][256u + tid] = p.ranges[(uint64_t)next_tile * stride + 256u + tid];
            __syncthreads();
            if (has_next) {
                const uint32_t *DN = sm.desc[(it + 1u) & 1u];
                const Phase dn = {DN[2], DN[3], DN[4], DN[5]};
                if (use_map(DN) && can_stage(dn)) prefetch_issue(pf, dn, drv, tid);
            }
        }
        lds_barrier();        // map / wcnt are reused by the next tile
        if (map_tile && tid == 0) p.tile_count[tile] = sm.wcnt[0] + sm.wcnt[1] + sm.wcnt[2] + sm.wcnt[3];
        item = next_item;
        tile = next_tile;
    }
    if (stamps && tid == 0)
        for (int i = 0; i < 8; i++) p.debug[(uint64_t)blockIdx.x * 8u + i] = tacc[i];
#undef II2_STAMP
}

// ---- per-64-tile sums of the tile counts (expand reads them to place its tiles) ------------
// A kernel of its own: an atomicAdd per tile from the tile kernel put 64 same-address device atomics in flight
// per sum at once, and every workgroup then waited for its own at its next s_waitcnt — ~20 us per pass.
__global__ __launch_bounds__(256) void k_isect_sums(IntersectParams p) {
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;      // one wave per sum: 64 counts, one load each
    if (g >= p.n_sums) return;
    const uint32_t t = 64u * g + (uint32_t)lane_id();
    const uint32_t s = wave_sum(t < p.n_tiles ? (p.tile_count[t] & ~LIST_FLAG) : 0u);
    if (lane_id() == 0) p.sums[g] = s;
}

// ---- expand: bitmaps / id lists -> the final ascending id array -----------------------------
// A tile's output offset = sum of the counts of the tiles before it, read off the per-64-tile
// sums the tile kernel accumulated plus the counts inside the tile's own group of 64 — a few
// 64-wide loads, once per expand workgroup (it then walks consecutive tiles), instead of a
// separate scan launch.  (Sums per 64 tiles keep the atomics spread: 64 adds per address.)
constexpr uint32_t SUMS_FROM_TILES = 4096;      // up to this many tiles the expand pass adds the counts up itself: no sums launch

__device__ __forceinline__ unsigned long long tile_offset(const IntersectParams &p, uint32_t tile) {
    const uint32_t l = (uint32_t)lane_id();
    unsigned long long off = 0;
    if (p.n_tiles <= SUMS_FROM_TILES) {           // (a query of a few thousand tiles is launch-bound: 64 loads per lane at most)
        uint32_t acc = 0;
        for (uint32_t i = l; i < tile; i += 64u) acc += p.tile_count[i] & ~LIST_FLAG;
        return wave_sum(acc);                      // (<= 4096 tiles of <= 4096 ids)
    }
    const uint32_t e1 = tile >> 6;
    for (uint32_t i = 0; i < e1; i += 64u) off += wave_sum(i + l < e1 ? p.sums[i + l] : 0u);
    const uint32_t a0 = e1 << 6;
    off += wave_sum(a0 + l < tile ? (p.tile_count[a0 + l] & ~LIST_FLAG) : 0u);
    return off;
}

__global__ __launch_bounds__(256) void k_isect_expand(IntersectParams p) {
    __shared__ unsigned long long s_off;
    __shared__ uint32_t stage[ISECT_GMAX * 256u];
    __shared__ uint32_t wsum[4];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t stride = p.desc_words;
    // workgroup w expands the consecutive tiles [w*per, (w+1)*per): one offset lookup, then a running sum
    const uint32_t per = (p.n_tiles + gridDim.x - 1u) / gridDim.x;
    const uint32_t tbeg = blockIdx.x * per;
    const uint32_t tend = tbeg + per < p.n_tiles ? tbeg + per : p.n_tiles;
    if (tbeg >= tend) return;
    if (wv == 0) {
        const unsigned long long o = tile_offset(p, tbeg);
        if (l == 0) s_off = o;
    }
    __syncthreads();
    unsigned long long running = s_off;
    for (uint32_t tile = tbeg; tile < tend; tile++) {
        const uint32_t cf = p.tile_count[tile];
        const uint32_t c = cf & ~LIST_FLAG;
        const uint64_t ob = running;
        running += c;
        if (tile == p.n_tiles - 1u && tid == 0) *p.d_count = running;
        if (c == 0) continue;
        const uint32_t *slot = p.tmp + (uint64_t)tile * p.slot_words;
        if (cf & LIST_FLAG) {
            for (uint32_t i = (uint32_t)tid; i < c; i += 256u)
                if (ob + i < p.out_cap) p.out[ob + i] = slot[i];
            continue;
        }
        const uint32_t *r = p.ranges + (uint64_t)tile * stride;
        const uint32_t mlo = r[0] & ~31u;
        const uint32_t nwords = ((r[1] - mlo) >> 5) + 1u;
        uint32_t done = 0;                       // ids written by earlier rounds
        // words per round: an intersection tile holds at most GMAX * 256 ids in all; a union tile can be full,
        // so its rounds cover no more docs than the staging buffer has slots
        const uint32_t rw = p.op_union ? ISECT_GMAX * 256u / 32u : 512u;       // two consecutive words per thread
        for (uint32_t w0 = 0; w0 < nwords; w0 += rw) {
            const uint32_t wi = w0 + 2u * (uint32_t)tid;
            const bool mine = 2u * (uint32_t)tid < rw;
            uint32_t wordA = (mine && wi < nwords) ? slot[wi] : 0u;
            uint32_t wordB = (mine && wi + 1u < nwords) ? slot[wi + 1u] : 0u;
            const uint32_t pc = (uint32_t)__popc(wordA) + (uint32_t)__popc(wordB);
            const uint32_t incl = wave_incl_scan(pc);
            __syncthreads();                      // stage / wsum are free (previous round fully written out)
            if (l == 63) wsum[wv] = incl;
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for (int w = 0; w < 4; w++) { if (w < wv) pre += wsum[w]; tot += wsum[w]; }
            uint32_t q = pre + incl - pc;
            uint32_t basedoc = mlo + 32u * wi;
            while (wordA) {
                const uint32_t bit = (uint32_t)__ffs((int)wordA) - 1u;
                stage[q++] = basedoc + bit;
                wordA &= wordA - 1u;
            }
            basedoc += 32u;
            while (wordB) {
                const uint32_t bit = (uint32_t)__ffs((int)wordB) - 1u;
                stage[q++] = basedoc + bit;
                wordB &= wordB - 1u;
            }
            __syncthreads();
            for (uint32_t i = (uint32_t)tid; i < tot; i += 256u)
                if (ob + done + i < p.out_cap) p.out[ob + done + i] = stage[i];
            done += tot;
        }
        __syncthreads();
    }
}

hipError_t launch_intersect(const IntersectParams &p, uint64_t *d_tile_off, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (p.n_tiles == 0) return hipSuccess;
    const uint64_t nthr = (uint64_t)p.n_tiles * (p.op_union ? p.n_lists : p.n_lists > 1u ? p.n_lists - 1u : 1u);
    const uint64_t pthr = std::max<uint64_t>(std::max<uint64_t>(nthr * 64u, p.n_sums), p.n_tiles);
    if (ev0) (void)hipEventRecord(ev0, s);          // the events bracket the whole pass: partition + tiles + expand
    hipLaunchKernelGGL(k_isect_partition, dim3((unsigned)((pthr + 255) / 256)), dim3(256), 0, s, p);
    {
        const uint32_t grid = p.n_tiles < p.max_grid ? p.n_tiles : p.max_grid;
        if (p.op_union && desc_stride(p.n_lists) > 256u) hipLaunchKernelGGL((k_isect_tiles<true, 0u, true>), dim3(grid), dim3(256), 0, s, p);
        else if (p.op_union) hipLaunchKernelGGL((k_isect_tiles<false, 0u, true>), dim3(grid), dim3(256), 0, s, p);
        else if (p.sub > 1u && desc_stride(p.n_lists) > 256u) hipLaunchKernelGGL((k_isect_tiles<true, 0u, false, true>), dim3(grid), dim3(256), 0, s, p);
        else if (p.sub > 1u) hipLaunchKernelGGL((k_isect_tiles<false, 0u, false, true>), dim3(grid), dim3(256), 0, s, p);
        else if (desc_stride(p.n_lists) > 256u) hipLaunchKernelGGL((k_isect_tiles<true, 0u, false>), dim3(grid), dim3(256), 0, s, p);
        else if (p.n_lists == 2u && !p.sparse_driver) hipLaunchKernelGGL((k_isect_tiles<false, 2u, false>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_isect_tiles<false, 0u, false>), dim3(grid), dim3(256), 0, s, p);
    }
    if (p.n_tiles > SUMS_FROM_TILES) hipLaunchKernelGGL(k_isect_sums, dim3((p.n_sums + 3u) / 4u), dim3(256), 0, s, p);
    const uint32_t egrid = p.n_tiles < 4096u ? p.n_tiles : 4096u;
    hipLaunchKernelGGL(k_isect_expand, dim3(egrid), dim3(256), 0, s, p);
    if (ev1) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}

}  // namespace ii2
