// merge.hip — segmented k-way union of DV1 posting lists with fused tombstone filter
// (gfx950, wave64, no MFMA).  One kernel family serves
//   * the segment merge: the body of Shard.Merge's loop (reference shard.go:163-212) and the
//     k-way merging iterator it drains (shard.go:253-278) folding equal terms with
//     file.MergeTermValues (file/types.go:14-22) — for every aligned term the sorted,
//     duplicate-free union of the k segments' lists, minus the tombstones (shard.go:181-190),
//     empty terms reported with count 0 (shard.go:192-194);
//   * the multi-term union of PrefixSearch (inverted_index.go:274-292) — one "term", k lists.
//
// Pass 1 decodes every input list once, streaming, into a raw u32 scratch array (one wave per
// DV1 block, all segments in one launch) and derives exact per-(segment, term) offsets.  Pass 2
// cuts the work into TILES that fit LDS (<= CAP postings):
//   small terms are packed, in term order, into batches of consecutive terms;
//   a large term is cut into doc-id ranges at exact quantiles of its longest list; where each
//   segment's list enters and leaves a range is found by binary search in the planning pass.
// A 512-thread workgroup handles a tile: it copies the tile's slices of the k raw lists into
// LDS as k runs sorted by (term, doc), folds the runs pairwise (log2 k levels; single-term tiles
// by merge-path + sequential two-way merges, batches by one binary search per element), then
// drops duplicates and tombstoned ids and compacts.  Tiles are independent: each parks its
// survivors in a scratch array at the input rank of its first posting; a scan of the tile counts
// and a packing pass then produce the CSR the reference's writer would have been fed: terms
// ascending, ids ascending.
#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t MCAP = MERGE_CAP;              // postings per tile
constexpr uint32_t OFFMAX = MERGE_OFFMAX;         // (nt+1) * k table entries
constexpr uint32_t MT = MERGE_THREADS;            // threads per workgroup of the tile kernel
constexpr uint32_t MW = MT / 64u;                 // waves per workgroup
constexpr uint32_t NBK = 4096;                    // buckets of the single-term fold (8 per thread)
constexpr uint32_t BKT_LIMIT = 24;                // fullest bucket the bucket fold accepts
constexpr uint32_t BKT_LIMIT_MT = 14;             // ... for batches (the slot shares 16 bits with the 12-bit bucket)
static_assert(NBK == 8u * MT && NBK + 4u <= 2u * OFFMAX, "bucket counters alias the list-offset table");

// ---- pass 1: decode everything once --------------------------------------------------------
// global block g of the merge input -> (segment, block of that segment)
__device__ __forceinline__ uint32_t seg_of_gblock(const MergeSegs &p, uint32_t g) {
    const uint32_t *cum = p.segtab + p.k;
    uint32_t lo = 0, hi = p.k;          // cum[lo] <= g < cum[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cum[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// lc[s * (T+1) + t] = postings of list (s, t) (0 for t == T); its exclusive scan is poff
__global__ void k_mlist_counts(MergeSegs p, uint32_t *__restrict__ lc) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n1 = p.n_terms + 1;
    if (i >= (uint64_t)p.k * n1) return;
    const uint32_t s = (uint32_t)(i / n1);
    const uint64_t t = i % n1;
    lc[i] = t < p.n_terms ? p.segs[s].cnt[t] : 0u;
}

// segtab: [s] = first block of segment s's term range, [k + s] = blocks of the segments before s, [2k] = all blocks (one wave)
__global__ void k_mseg_blocks(MergeSegs p, uint32_t *__restrict__ segtab) {
    const uint32_t s = threadIdx.x;
    uint32_t b0 = 0, cnt = 0;
    if (s < p.k) { b0 = p.segs[s].blk_off[0]; cnt = p.segs[s].blk_off[p.n_terms] - b0; }
    const uint32_t incl = wave_incl_scan(cnt);
    if (s < p.k) { segtab[s] = b0; segtab[p.k + s] = incl - cnt; }
    if (s == 63u) segtab[2u * p.k] = incl;
}

// Every block but a list's last holds II2_DV1_BLOCK postings, so block j of list (s, t) decodes to
// raw[poff[s, t] + 256 j ...].  Most lists of a Zipf index are tiny (a handful of postings per segment
// and term): blocks with up to TINY_BYTES of payload are decoded one per LANE from registers (all seven
// possible dwords fetched at once); the others are appended to a list and decoded one per 16-lane ROW.
constexpr uint32_t TINY_BYTES = 28;

// which (segment, list, block) a global block is, where it decodes to, and whether it is left to the row kernel
struct BlkRef { bool valid, big; uint32_t s, q0, q1, first; unsigned long long pos; };
__device__ __forceinline__ BlkRef blk_ref(const MergeSegs &p, const unsigned long long *__restrict__ poff, uint64_t g, bool want_pos) {
    BlkRef r;
    r.valid = false; r.big = false; r.s = 0; r.q0 = 0; r.q1 = 0; r.first = 0; r.pos = 0;
    if (g >= p.segtab[2u * p.k]) return r;
    const uint32_t s = seg_of_gblock(p, (uint32_t)g);
    const SegView sv = p.segs[s];
    const uint32_t b = p.segtab[s] + ((uint32_t)g - p.segtab[p.k + s]);
    const uint32_t t = sv.blk_list[b] - sv.list_base;
    if (t >= p.n_terms) return r;
    const ii2_skip e0 = sv.skip[b];
    r.valid = true;
    r.s = s;
    r.q0 = e0.byte_off;
    r.q1 = sv.skip[b + 1].byte_off;
    r.first = e0.first_doc;
    r.big = r.q1 - r.q0 > TINY_BYTES;
    if (want_pos) r.pos = poff[(uint64_t)s * (p.n_terms + 1) + t] + (unsigned long long)(b - sv.blk_off[t]) * II2_DV1_BLOCK;
    return r;
}

// non-tiny blocks per workgroup of k_mdec_lane (their scan gives every workgroup its slice of the work list:
// one address sustains only ~90 atomics/us, far too few for an append per wave)
__global__ __launch_bounds__(256) void k_mbig_count(MergeSegs p, uint32_t *__restrict__ wgcnt) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const BlkRef r = blk_ref(p, nullptr, g, false);
    const int c = __syncthreads_count(r.big ? 1 : 0);
    if (threadIdx.x == 0) wgcnt[blockIdx.x] = (uint32_t)c;
}

__global__ __launch_bounds__(256) void k_mdec_lane(MergeSegs p, const unsigned long long *__restrict__ poff, uint32_t *__restrict__ raw,
                                                    const uint32_t *__restrict__ wgbase, uint4 *__restrict__ ent0, uint2 *__restrict__ ent1) {
    __shared__ uint32_t wcnt[4];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const BlkRef r = blk_ref(p, poff, g, true);
    if (r.valid && !r.big) {
        const uint32_t len = r.q1 - r.q0;
        const uint8_t *pl = p.segs[r.s].payload + r.q0;
        uint32_t w[7];
#pragma unroll
        for (int j = 0; j < 7; j++) w[j] = (uint32_t)(4 * j) < len ? load_u32_unaligned(pl + 4u * j) : 0u;
        uint32_t *out = raw + r.pos;
        uint32_t cur = r.first, acc = 0, sh = 0;
        *out++ = cur;
#pragma unroll
        for (int j = 0; j < 28; j++) {
            if ((uint32_t)j < len) {
                const uint32_t c = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                acc += (c & 0x7Fu) << sh;
                if (c & 0x80u) sh = sh < 28u ? sh + 7u : 28u;
                else { cur += acc; *out++ = cur; acc = 0; sh = 0; }
            }
        }
    }
    // work list entries of the blocks left to the row kernel, in block order
    const unsigned long long m = __ballot(r.big);
    const int wv = (int)threadIdx.x >> 6;
    if (lane_id() == 0) wcnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (r.big) {
        uint32_t at = wgbase[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
        for (int w2 = 0; w2 < wv; w2++) at += wcnt[w2];
        ent0[at] = make_uint4((uint32_t)r.pos, (uint32_t)(r.pos >> 32), r.q0, r.q1);
        ent1[at] = make_uint2(r.first, r.s);
    }
}

__global__ __launch_bounds__(256) void k_mdec_rows(MergeSegs p, uint32_t *__restrict__ raw, const uint4 *__restrict__ ent0,
                                                    const uint2 *__restrict__ ent1, const uint32_t *__restrict__ nbig) {
    // a row's postings are collected in LDS and leave as 16-byte stores: a block is contiguous in raw, but the lanes of its
    // row hold 16 payload bytes each — a varying number of postings — so storing from the decoder would be one partly
    // filled 4-byte store instruction per posting of the fullest lane
    __shared__ __align__(16) uint32_t stage[4][4][II2_DV1_BLOCK];
    const uint32_t n = *nbig;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    const uint32_t wv = threadIdx.x >> 6, row = ((uint32_t)threadIdx.x >> 4) & 3u, rl = (uint32_t)threadIdx.x & 15u;
    uint32_t *st = stage[wv][row];
    for (uint64_t z = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; __ballot(z < n) != 0ull; z += stride) {
        const bool rv = z < n;
        uint32_t q0 = 0, q1 = 0, first = 0;
        uint32_t *out = raw;
        const uint8_t *pl = nullptr;     // rows of one wave may read different segments
        if (rv) {
            const uint4 e0 = ent0[z];
            const uint2 e1 = ent1[z];
            q0 = e0.z;
            q1 = e0.w;
            first = e1.x;
            out = raw + ((unsigned long long)e0.x | ((unsigned long long)e0.y << 32));
            pl = p.segs[e1.y].payload;
        }
        const uint32_t cnt = decode_rows16_any(pl, q0, q1, first, rv, [&](uint32_t ix, uint32_t id) { st[ix & (II2_DV1_BLOCK - 1u)] = id; });
        if (rv) {
            const uint32_t c = cnt < II2_DV1_BLOCK ? cnt : II2_DV1_BLOCK;       // (imported segments are validated: a block holds <= 256)
            for (uint32_t i = 4u * rl; i < c; i += 64u) {
                if (i + 4u <= c) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(&st[i]);
                    __builtin_memcpy(out + i, &v, 16);
                } else {
                    for (uint32_t j = i; j < c; j++) out[j] = st[j];
                }
            }
        }
    }
}

// ---- plan -------------------------------------------------------------------------------
// exact input postings of every term, its packing weight, and the tiles of a large term
__global__ void k_merge_term_ub(MergeParams p, uint32_t *__restrict__ ub, uint32_t *__restrict__ weight,
                                uint32_t *__restrict__ ntiles_large) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.n_terms) return;
    if (t == p.n_terms) { ub[t] = 0; weight[t] = 0; ntiles_large[t] = 0; return; }
    const uint64_t n1 = p.n_terms + 1;
    uint64_t u = 0;
    for (uint32_t s = 0; s < p.k; s++) u += p.poff[s * n1 + t + 1] - p.poff[s * n1 + t];
    const uint32_t u32 = u > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)u;
    ub[t] = u32;
    if (u32 > p.small_max) {
        weight[t] = 0;
        ntiles_large[t] = (uint32_t)((u + p.large_tile - 1) / p.large_tile);
    } else {
        weight[t] = u32 > p.wmin ? u32 : p.wmin;
        ntiles_large[t] = 0;
    }
}

// head[t] = 1 when small term t opens a new batch
__global__ void k_merge_heads(MergeParams p, const uint32_t *__restrict__ ntl, const uint64_t *__restrict__ wpre,
                              uint32_t *__restrict__ head) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.n_terms) return;
    if (t == p.n_terms || ntl[t] > 0) { head[t] = 0; return; }
    bool h = t == 0 || ntl[t - 1] > 0;
    if (!h) h = (wpre[t] / p.batch_q) != (wpre[t - 1] / p.batch_q);
    head[t] = h ? 1u : 0u;
}

// term_tile[t] = id of the (first) tile of term t; monotone in t
__global__ void k_merge_term_tile(MergeParams p, const uint32_t *__restrict__ ntl, const uint32_t *__restrict__ head,
                                  const uint32_t *__restrict__ hpre, const uint32_t *__restrict__ lpre,
                                  uint32_t *__restrict__ term_tile) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.n_terms) return;
    if (t == p.n_terms) { term_tile[t] = hpre[t] + lpre[t]; return; }
    if (ntl[t] > 0) term_tile[t] = hpre[t] + lpre[t];
    else term_tile[t] = hpre[t] + head[t] - 1u + lpre[t];
}

// tile descriptors {t0, t1, lo, hi}
__global__ void k_merge_tile_desc(MergeParams p, const uint32_t *__restrict__ ntl, const uint32_t *__restrict__ term_tile,
                                  uint4 *__restrict__ desc) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= *p.n_tiles_dev) return;
    // last term whose first tile <= tile
    uint64_t lo = 0, hi = p.n_terms;           // term_tile[lo] <= tile < term_tile[hi] (term_tile[n_terms] = n_tiles)
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (term_tile[mid] <= tile) lo = mid; else hi = mid;
    }
    const uint64_t tl = lo;
    if (ntl[tl] > 0) {                          // tile j of large term tl
        const uint32_t m = ntl[tl], j = tile - term_tile[tl];
        // splitters: exact quantiles of the term's longest list (raw ids)
        const uint64_t n1 = p.n_terms + 1;
        uint32_t best_s = 0;
        uint64_t best_n = 0;
        for (uint32_t s = 0; s < p.k; s++) {
            const uint64_t len = p.poff[s * n1 + tl + 1] - p.poff[s * n1 + tl];
            if (len > best_n) { best_n = len; best_s = s; }
        }
        const uint32_t *lst = p.raw + p.poff[best_s * n1 + tl];
        auto splitter = [&](uint32_t jj) -> uint32_t { return lst[((uint64_t)jj * best_n) / m]; };
        // tile j covers [S_j, S_{j+1}) with S_0 = 0 and S_m = 2^32; S is non-decreasing in j
        const uint64_t lo64 = j > 0 ? (uint64_t)splitter(j) : 0ull;
        const uint64_t hi64 = j + 1u < m ? (uint64_t)splitter(j + 1u) : (1ull << 32);
        uint32_t dlo, dhi;
        if (hi64 <= lo64) {                                   // empty range (dlo > dhi); dhi + 1 still is the upper splitter
            if (hi64 > 0) { dlo = (uint32_t)hi64; dhi = (uint32_t)(hi64 - 1ull); }
            else { dlo = 1u; dhi = 0u; }
        }
        else { dlo = (uint32_t)lo64; dhi = (uint32_t)(hi64 - 1ull); }
        desc[tile] = make_uint4((uint32_t)tl, (uint32_t)tl + 1u, dlo, dhi);
    } else {
        // batch: terms [first with term_tile == tile, last with term_tile == tile]
        uint64_t a = 0, b = tl;                 // find first term with term_tile >= tile
        if (term_tile[0] >= tile) b = 0;
        else {
            a = 0;                              // term_tile[a] < tile <= term_tile[b]
            while (b - a > 1) {
                const uint64_t mid = a + ((b - a) >> 1);
                if (term_tile[mid] < tile) a = mid; else b = mid;
            }
        }
        desc[tile] = make_uint4((uint32_t)b, (uint32_t)tl + 1u, 0u, 0xFFFFFFFFu);
    }
}

// ends[tile * k + s] = postings of list (s, t0) with doc <= the tile's upper bound (large-term tiles only).
// Two levels: the block is found in the segment's skip table (8 bytes per 256 postings, cache-resident), only the last
// eight steps touch the decoded list itself — the searches over the raw arrays used to fetch 0.8 GB per merge.
// first index i in [lo, hi) with get(i) > x, for an ascending sequence that is close to uniform between vlo (a lower bound of
// get(lo)) and vhi (an upper bound of get(hi - 1)): a linear guess, a doubling walk away from it until x is bracketed, then
// bisection inside the bracket — a handful of dependent loads instead of log2(hi - lo).  Exact for any ascending input.
template <class Get>
__device__ __forceinline__ uint32_t upper_bound_guess(Get get, uint32_t lo, uint32_t hi, uint32_t x, uint32_t vlo, uint32_t vhi) {
    if (lo >= hi) return lo;
    if (hi - lo > 4u && vhi > vlo) {
        const uint64_t rel = x > vlo ? (uint64_t)(x - vlo) : 0ull;
        uint64_t g64 = (uint64_t)lo + rel * (uint64_t)(hi - lo) / ((uint64_t)(vhi - vlo) + 1ull);
        uint32_t g = g64 >= hi ? hi - 1u : (uint32_t)g64;
        if (get(g) <= x) {                                  // answer in (g, hi]: walk up
            uint32_t a = g + 1u, step = 1u;
            while (a < hi) {
                const uint32_t pr = a + step - 1u < hi ? a + step - 1u : hi - 1u;
                if (get(pr) <= x) { a = pr + 1u; step <<= 1; } else { hi = pr; break; }
            }
            lo = a;                                         // get(i) <= x for i < lo; get(hi) > x or hi is the end
        } else {                                            // answer in [lo, g]: walk down
            uint32_t b = g, step = 1u;
            while (b > lo) {
                const uint32_t pr = b - lo > step ? b - step : lo;
                if (get(pr) > x) { b = pr; step <<= 1; } else { lo = pr + 1u; break; }
            }
            hi = b;
        }
    }
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (get(mid) <= x) lo = mid + 1u; else hi = mid; }
    return lo;
}

__global__ void k_merge_tile_ends(MergeParams p, MergeSegs ms, const uint4 *__restrict__ desc, uint32_t *__restrict__ ends) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)*p.n_tiles_dev * p.k) return;
    const uint32_t tile = (uint32_t)(i / p.k), s = (uint32_t)(i % p.k);
    const uint4 td = desc[tile];
    if (td.z == 0u && td.w == 0xFFFFFFFFu) return;           // whole lists: nothing to search
    const uint64_t n1 = p.n_terms + 1;
    const uint64_t beg = p.poff[s * n1 + td.x], end = p.poff[s * n1 + td.y];
    const uint32_t len = (uint32_t)(end - beg);
    const SegView sv = ms.segs[s];
    const uint32_t b_lo = sv.blk_off[td.x], b_hi = sv.blk_off[td.x + 1u];
    uint32_t res = 0;
    if (b_hi > b_lo) {
        const uint32_t f0 = sv.skip[b_lo].first_doc, fl = sv.skip[b_hi - 1u].first_doc;
        // first block of the list whose first doc is > td.w (the doc ids of a large term's list are close to uniform)
        const uint32_t a = upper_bound_guess([&](uint32_t j) { return sv.skip[j].first_doc; }, b_lo, b_hi, td.w, f0, fl);
        if (a > b_lo) {                                       // block a - 1 starts at or before td.w: the boundary lies inside it (or at its end)
            const uint32_t base = (a - 1u - b_lo) * II2_DV1_BLOCK;
            const uint32_t *lst = p.raw + beg + base;
            const uint32_t cntb = len - base < II2_DV1_BLOCK ? len - base : II2_DV1_BLOCK;
            const uint32_t fb = sv.skip[a - 1u].first_doc;
            const uint32_t fn = a < b_hi ? sv.skip[a].first_doc : (cntb ? lst[cntb - 1u] : fb);
            res = base + upper_bound_guess([&](uint32_t j) { return lst[j]; }, 0u, cntb, td.w, fb, fn);
        }
    }
    ends[i] = res;
}

// where every segment's list enters and leaves a tile's doc range: rng[2 * (tile * k + s)] =
// (position in raw of the first posting inside the range: lo, hi; postings inside; postings of list (s, t0) before it),
// rng[.. + 1] = (first doc, last doc of the slice, -, -).  A range starts where the previous tile of the term ended.
__global__ void k_merge_tile_ranges(MergeParams p, const uint4 *__restrict__ desc, const uint32_t *__restrict__ ends, uint4 *__restrict__ rng) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)*p.n_tiles_dev * p.k) return;
    const uint32_t tile = (uint32_t)(i / p.k), s = (uint32_t)(i % p.k);
    const uint4 td = desc[tile];
    const uint64_t n1 = p.n_terms + 1;
    const uint64_t beg = p.poff[s * n1 + td.x], end = p.poff[s * n1 + td.y];
    const uint32_t len = (uint32_t)(end - beg);
    const uint32_t *lst = p.raw + beg;
    uint32_t a = 0, e = len;
    if (td.z > td.w) { a = 0; e = 0; }                       // empty doc range
    else if (!(td.z == 0u && td.w == 0xFFFFFFFFu)) {
        e = ends[i];
        a = td.z == 0u ? 0u : ends[i - p.k];                 // td.z > 0: the tile before belongs to the same term and ends at td.z - 1
    }
    const uint64_t rs = beg + a;
    rng[2 * i] = make_uint4((uint32_t)rs, (uint32_t)(rs >> 32), e - a, a);
    rng[2 * i + 1] = e > a ? make_uint4(lst[a], lst[e - 1u], 0u, 0u) : make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
}

// ---- the tile kernel ----------------------------------------------------------------------
struct __align__(16) MergeSmem {
    uint32_t vals[2][MCAP];
    uint16_t tids[2][MCAP];             // (run << 10 | term) of each element
    uint32_t offs[2][OFFMAX];           // per run: nt+1 list offsets inside the run
    uint32_t runbase[2][MAX_LISTS + 2];
    unsigned long long rs[MAX_LISTS];   // position in raw of each run's first posting
    uint32_t spre[MAX_LISTS + 1];       // run lengths
    uint32_t sbl_rank[MAX_LISTS];       // postings of each list that precede the range
    uint2 tterm[MT];                    // batches: per term {smallest doc, float bits of buckets per doc}
    uint16_t ttb[MT + 2];               // batches: per term its first bucket = its first position after the fold
    uint32_t rmin[MAX_LISTS], rmax[MAX_LISTS];   // first / last doc of each run (single-term tiles)
    uint32_t wsum[MW];
    uint32_t wmax[MW];
    uint32_t vmin, vmax;                // doc range the tile's postings really span
    uint32_t n_in;
    uint32_t rank;                      // input postings of the tile's term(s) that precede the tile's doc range
    uint32_t stk[70][2];                // bisection stack of doc ranges (oversized tiles)
};

// block-wide exclusive scan of one value per thread (MT threads); returns exclusive prefix, total in *tot
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *tot) {
    const int l = lane_id(), wv = (int)threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();
    if (l == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = 0;
    uint32_t t = 0;
    for (int w = 0; w < (int)MW; w++) { if (w < wv) pre += wsum[w]; t += wsum[w]; }
    *tot = t;
    return pre + incl - v;
}

// KFIX: segment count known at compile time (0 = read it from the parameters); 16-way merges get their own instantiation
template <uint32_t KFIX>
__global__ __launch_bounds__(MERGE_THREADS, 4) void k_merge_tiles(MergeParams p, const uint4 *__restrict__ tile_desc) {
    __shared__ MergeSmem sm;
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t k = KFIX ? KFIX : p.k;
    // diagnostics only: thread 0 sums the cycles spent in each step of the tile loop
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

    const uint32_t n_tiles = *p.n_tiles_dev;       // computed by the plan kernels; the host only knows an upper bound
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint4 td = tile_desc[tile];
        const uint32_t t0 = td.x, t1 = td.y;
        const uint32_t nt = t1 - t0;
        const uint32_t stride = nt + 1u;
        const uint64_t n1 = p.n_terms + 1;

        // ---- step A: the k runs of the doc range [dlo, dhi] — where they start in raw, how long they
        // are, where they go in LDS.  root: the planning pass already searched the range (rng);
        // otherwise (a bisected leaf) every segment's list is binary-searched here.
        // Returns false when the range does not fit LDS.
        auto load_range = [&](uint32_t dlo, uint32_t dhi, bool root) -> bool {
            __syncthreads();
            if (nt == 1u && k > 1u) {       // bucket counters of the single-term fold (they live in the unused list-offset table)
                uint4 *z = reinterpret_cast<uint4 *>(&sm.offs[0][0]);
                for (uint32_t i = (uint32_t)tid; i < (NBK + 4u) / 4u; i += MT) z[i] = make_uint4(0, 0, 0, 0);
            } else if (k > 1u) {            // batches: 4096 16-bit bucket counters in the second list-offset table
                reinterpret_cast<uint4 *>(&sm.offs[1][0])[tid] = make_uint4(0, 0, 0, 0);
            }
            if ((uint32_t)tid < k) {
                if (root) {
                    const uint4 r0 = p.rng[2u * ((uint64_t)tile * k + (uint32_t)tid)], r1 = p.rng[2u * ((uint64_t)tile * k + (uint32_t)tid) + 1u];
                    sm.rs[tid] = (unsigned long long)r0.x | ((unsigned long long)r0.y << 32);
                    sm.spre[tid] = r0.z;
                    sm.sbl_rank[tid] = r0.w;
                    sm.rmin[tid] = r1.x;
                    sm.rmax[tid] = r1.y;
                } else {
                    const uint64_t beg = p.poff[(uint32_t)tid * n1 + t0], end = p.poff[(uint32_t)tid * n1 + t1];
                    const uint32_t len = (uint32_t)(end - beg);
                    const uint32_t *lst = p.raw + beg;
                    uint32_t lo = 0, hi = len;
                    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (lst[mid] < dlo) lo = mid + 1u; else hi = mid; }
                    const uint32_t a = lo;
                    hi = len;
                    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (lst[mid] <= dhi) lo = mid + 1u; else hi = mid; }
                    const uint32_t e = lo;
                    sm.rs[tid] = beg + a;
                    sm.spre[tid] = e - a;
                    sm.sbl_rank[tid] = a;
                    sm.rmin[tid] = e > a ? lst[a] : 0xFFFFFFFFu;
                    sm.rmax[tid] = e > a ? lst[e - 1u] : 0u;
                }
            }
            __syncthreads();
            if (wv == 0) {
                const uint32_t c = (uint32_t)l < k ? sm.spre[l] : 0u;
                const uint32_t incl = wave_incl_scan(c);
                if ((uint32_t)l < k) sm.runbase[0][l] = incl - c;
                if (l == 63) { sm.runbase[0][k] = incl; sm.n_in = incl; }
                const uint32_t r = wave_sum((uint32_t)l < k ? sm.sbl_rank[l] : 0u);
                if (l == 0) sm.rank = r;
                uint32_t mn = (uint32_t)l < k ? sm.rmin[l] : 0xFFFFFFFFu, mx = (uint32_t)l < k ? sm.rmax[l] : 0u;
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint32_t on = (uint32_t)__shfl_xor((int)mn, d, 64), ox = (uint32_t)__shfl_xor((int)mx, d, 64);
                    mn = on < mn ? on : mn;
                    mx = ox > mx ? ox : mx;
                }
                if (l == 0) { sm.vmin = mn; sm.vmax = mx; }
            }
            // list offsets inside each run (batches of several terms)
            if (nt > 1u) {
                for (uint32_t e = (uint32_t)tid; e < k * stride; e += MT) {
                    const uint32_t s = e / stride, t = e % stride;
                    sm.offs[0][e] = (uint32_t)(p.poff[s * n1 + t0 + t] - p.poff[s * n1 + t0]);
                }
            }
            __syncthreads();
            II2_STAMP(0)      // A: run ranges, list table
            return sm.n_in <= MCAP;
        };

        // ---- steps D-F on a loaded range: copy the runs into LDS, fold them, dedupe + tombstones + compact.
        // Survivors land in sm.vals[*outbuf][0..return) (*outbuf = 2: in the tag arrays); per-term counts go to out_counts when asked.
        auto merge_range = [&](uint32_t *outbuf, bool emit_counts, bool atomic_counts) -> uint32_t {
            const uint32_t n_in = sm.n_in;
            *outbuf = 0;
            if (n_in == 0) return 0u;
            // Single-term tiles whose doc range fits a bitmap in the two value arrays (2 * MCAP words = 262144 docs: terms
            // with >= 1 posting per 64 docs, a third of a Zipf workload's postings): mark, clear the tombstoned bits word
            // by word (coalesced loads of exactly the tile's range — no per-posting probe), count, extract.  The union,
            // the dedupe and the order come for free; ~5x fewer instructions per posting than the bucket fold below.
            if (p.bitmap_tiles && nt == 1u && k > 1u && sm.vmax - (sm.vmin & ~31u) < 2u * MCAP * 32u) {
                const uint32_t lo32 = sm.vmin & ~31u;
                const uint32_t nw = ((sm.vmax - lo32) >> 5) + 1u;                 // <= 2 * MCAP
                uint32_t *bm = &sm.vals[0][0];                                    // vals[0] and vals[1] are contiguous
                for (uint32_t i = 4u * (uint32_t)tid; i < nw; i += 4u * MT) *reinterpret_cast<uint4 *>(&bm[i]) = make_uint4(0, 0, 0, 0);
                __syncthreads();
                {   // every wave marks whole runs (run s to wave s mod MW), loads four deep
                    const uint32_t *RB = sm.runbase[0];
                    for (uint32_t s2 = (uint32_t)wv; s2 < k; s2 += MW) {
                        const uint32_t len = RB[s2 + 1u] - RB[s2];
                        const uint32_t *src = p.raw + sm.rs[s2];
                        for (uint32_t i0 = 0; i0 < len; i0 += 256u) {
                            uint32_t v4[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)l;
                                v4[j] = i < len ? src[i] : 0u;
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)l;
                                if (i < len) atomicOr(&bm[(v4[j] - lo32) >> 5], 1u << (v4[j] & 31u));
                            }
                        }
                    }
                }
                __syncthreads();
                if (p.tomb) {     // the tombstone words of exactly this range, coalesced and four in flight per thread
                    const uint32_t twb = lo32 >> 5;
                    for (uint32_t i0 = (uint32_t)tid; i0 < nw; i0 += 4u * MT) {
                        uint32_t t4[4];
#pragma unroll
                        for (uint32_t j = 0; j < 4u; j++) {
                            const uint32_t i = i0 + j * MT;
                            t4[j] = (i < nw && twb + i < p.tomb_nwords) ? p.tomb[twb + i] : 0u;
                        }
#pragma unroll
                        for (uint32_t j = 0; j < 4u; j++)
                            if (t4[j]) bm[i0 + j * MT] &= ~t4[j];
                    }
                    __syncthreads();
                }
                II2_STAMP(3)      // D: gather (here: mark, tombstones)
                // consecutive words per thread, as few as cover the range (a dense tile spans few words: one each).  Two
                // passes over the thread's own words in LDS instead of sixteen registers: this kernel has none to spare.  The ids go to the tag arrays (4096 words, unused by this path): the
                // bitmap occupies both value arrays.
                const uint32_t wpt = (nw + MT - 1u) / MT;                         // 1 .. 16
                const uint32_t w0 = wpt * (uint32_t)tid;
                const uint32_t w1 = w0 + wpt < nw ? w0 + wpt : nw;
                uint32_t cnt = 0;
#pragma unroll 1
                for (uint32_t w = w0; w < w1; w++) cnt += (uint32_t)__popc(bm[w]);
                uint32_t tot;
                uint32_t pos = block_excl_scan(cnt, sm.wsum, &tot);
                uint32_t *V2 = reinterpret_cast<uint32_t *>(&sm.tids[0][0]);
#pragma unroll 1
                for (uint32_t w = w0; w < w1; w++) {
                    uint32_t x = bm[w];
                    const uint32_t base = lo32 + 32u * w;
                    while (x) {
                        V2[pos++] = base + (uint32_t)__ffs((int)x) - 1u;
                        x &= x - 1u;
                    }
                }
                *outbuf = 2u;     // (the tag arrays)
                if (emit_counts && tid == 0 && tot) {
                    if (atomic_counts) atomicAdd(&p.out_counts[t0], tot);
                    else p.out_counts[t0] = tot;
                }
                __syncthreads();
                II2_STAMP(1)      // E1: single-term fold (here: tombstones, count, extract)
                return tot;
            }
            // Single-term tiles (the tiles of large terms — most of the postings) are folded by a bucket sort:
            // a monotone map of the doc id onto NBK buckets, slot inside the bucket from an LDS counter,
            // then every posting ranks itself among the few that share its bucket.  Docs clustered so that a
            // bucket overflows BKT_LIMIT send the tile to the pairwise merge below instead.
            const bool bucketed = nt == 1u && k > 1u;
            uint32_t *bkt = &sm.offs[0][0];
            const uint32_t vmin = sm.vmin;
            const float binv = (float)NBK / ((float)(sm.vmax - vmin) + 1.0f);
            auto bucket_of = [&](uint32_t v) -> uint32_t {
                const uint32_t b = (uint32_t)((float)(v - vmin) * binv);
                return b < NBK - 1u ? b : NBK - 1u;
            };
            // ---- D. gather the runs into vals[0] (coalesced inside every run) ----
            // MCAP = 8 * MT: eight postings per thread, all global loads issued before the first LDS write
            if (bucketed) {
                // single-term tile: every wave copies whole runs (run s to wave s mod MW), lane i of the wave the run's
                // elements i, i + 64, ... — no search for the run an element belongs to, loads four deep
                const uint32_t *RB = sm.runbase[0];
                for (uint32_t s2 = (uint32_t)wv; s2 < k; s2 += MW) {
                    const uint32_t base = RB[s2], len = RB[s2 + 1u] - base;
                    const uint32_t *src = p.raw + sm.rs[s2];
                    for (uint32_t i0 = 0; i0 < len; i0 += 256u) {
                        uint32_t v4[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)l;
                            v4[j] = i < len ? src[i] : 0u;
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)l;
                            if (i < len) {
                                sm.vals[0][base + i] = v4[j];
                                sm.tids[0][base + i] = (uint16_t)atomicAdd(&bkt[bucket_of(v4[j])], 1u);
                            }
                        }
                    }
                }
            } else {
                // batches (and the odd single-run tile): the same wave-by-wave copy of whole runs; the (run, term) tag of
                // every element comes from markers instead of a search per element — the first element of every
                // non-empty list gets its tag, and since the tags ascend along the run-major order an inclusive
                // max-scan over the positions fills in the rest
                const uint32_t *RB = sm.runbase[0];
                uint32_t *tg32 = reinterpret_cast<uint32_t *>(&sm.tids[0][0]);
                for (uint32_t i = (uint32_t)tid; i < (n_in + 1u) / 2u; i += MT) tg32[i] = 0u;
                __syncthreads();
                for (uint32_t s2 = (uint32_t)wv; s2 < k; s2 += MW) {
                    const uint32_t base = RB[s2], len = RB[s2 + 1u] - base;
                    if (len == 0u) continue;                                   // (wave-uniform)
                    if (nt > 1u) {
                        const uint32_t *O = sm.offs[0] + s2 * stride;
                        for (uint32_t t = (uint32_t)l; t < nt; t += 64u) {
                            const uint32_t o0 = O[t], o1 = O[t + 1u];
                            if (o1 > o0) sm.tids[0][base + o0] = (uint16_t)((s2 << 10) | t);
                        }
                    } else if (l == 0) sm.tids[0][base] = (uint16_t)(s2 << 10);
                    const uint32_t *src = p.raw + sm.rs[s2];
                    for (uint32_t i0 = 0; i0 < len; i0 += 256u) {
                        uint32_t v4[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)l;
                            v4[j] = i < len ? src[i] : 0u;
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)l;
                            if (i < len) sm.vals[0][base + i] = v4[j];
                        }
                    }
                }
                __syncthreads();
                {   // inclusive max-scan of the tags: eight consecutive positions per thread, then across the workgroup
                    uint4 *tg4 = reinterpret_cast<uint4 *>(&sm.tids[0][0]);
                    const uint4 q = tg4[tid];
                    uint32_t g[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
#pragma unroll
                    for (int j = 1; j < 8; j++) g[j] = g[j] > g[j - 1] ? g[j] : g[j - 1];
                    uint32_t m = g[7];                                         // inclusive max over the wave's threads up to mine
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o = (uint32_t)__shfl_up((int)m, d, 64);
                        if (l >= d) m = o > m ? o : m;
                    }
                    if (l == 63) sm.wmax[wv] = m;
                    uint32_t before = (uint32_t)__shfl_up((int)m, 1, 64);      // exclusive: the threads before mine in the wave
                    if (l == 0) before = 0u;
                    __syncthreads();
                    for (int w2 = 0; w2 < wv; w2++) before = sm.wmax[w2] > before ? sm.wmax[w2] : before;
#pragma unroll
                    for (int j = 0; j < 8; j++) g[j] = g[j] > before ? g[j] : before;
                    tg4[tid] = make_uint4(g[0] | (g[1] << 16), g[2] | (g[3] << 16), g[4] | (g[5] << 16), g[6] | (g[7] << 16));
                }
            }
            __syncthreads();
            II2_STAMP(3)      // D: gather
            // ---- E. fold the runs ----
            uint32_t cur = 0, nruns = k;
            bool flagged = false;          // the bucket folds already marked duplicates and tombstoned ids (bit 15 of the tag)
            if (bucketed) {
                // exclusive scan of the bucket counters in place (8 per thread) and the fullest bucket
                uint32_t c[8], sum = 0, mxc = 0;
                {
                    const uint4 a4 = *reinterpret_cast<const uint4 *>(&bkt[8u * (uint32_t)tid]);
                    const uint4 b4 = *reinterpret_cast<const uint4 *>(&bkt[8u * (uint32_t)tid + 4u]);
                    c[0] = a4.x; c[1] = a4.y; c[2] = a4.z; c[3] = a4.w; c[4] = b4.x; c[5] = b4.y; c[6] = b4.z; c[7] = b4.w;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) { sum += c[j]; mxc = c[j] > mxc ? c[j] : mxc; }
                for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mxc, d, 64); mxc = o > mxc ? o : mxc; }
                if (l == 0) sm.wmax[wv] = mxc;
                uint32_t tot_;
                uint32_t run = block_excl_scan(sum, sm.wsum, &tot_);
                mxc = 0;
                for (int w = 0; w < (int)MW; w++) mxc = sm.wmax[w] > mxc ? sm.wmax[w] : mxc;
                if (mxc <= BKT_LIMIT) {
                    uint32_t ex[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { ex[j] = run; run += c[j]; }
                    *reinterpret_cast<uint4 *>(&bkt[8u * (uint32_t)tid]) = make_uint4(ex[0], ex[1], ex[2], ex[3]);
                    *reinterpret_cast<uint4 *>(&bkt[8u * (uint32_t)tid + 4u]) = make_uint4(ex[4], ex[5], ex[6], ex[7]);
                    if (tid == (int)MT - 1) bkt[NBK] = run;
                    __syncthreads();
                    // scatter into bucket order
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t e = (uint32_t)tid + (uint32_t)j * MT;
                        if (e < n_in) {
                            const uint32_t v = sm.vals[0][e];
                            sm.vals[1][bkt[bucket_of(v)] + sm.tids[0][e]] = v;
                        }
                    }
                    __syncthreads();
                    // rank inside the bucket (equal ids keep their bucket order: the dedupe below wants them adjacent)
#pragma unroll 4
                    for (uint32_t q = (uint32_t)tid; q < n_in; q += MT) {
                        const uint32_t v = sm.vals[1][q];
                        const uint32_t b = bucket_of(v);
                        const uint32_t lo = bkt[b], hi = bkt[b + 1u];
                        const uint32_t ts = (p.tomb && (v >> 5) < p.tomb_nwords) ? p.tomb_summary[v >> 9] : 0u;   // in flight during the loop
                        uint32_t r = 0, dup = 0;
                        {   // the first four of the bucket without a loop (buckets hold one posting on average: lanes that
                            // loop make the whole wave wait for the fullest bucket among its 64); reading past the bucket
                            // is harmless (masked), past vals[1] lands in the tag arrays
                            const uint32_t *B4 = &sm.vals[1][lo];
                            const uint32_t u0 = B4[0], u1 = B4[1], u2 = B4[2], u3 = B4[3];
                            const uint32_t nb = hi - lo;
                            const uint32_t e0 = (u0 == v && lo < q) ? 1u : 0u;
                            const uint32_t e1 = (nb > 1u && u1 == v && lo + 1u < q) ? 1u : 0u;
                            const uint32_t e2 = (nb > 2u && u2 == v && lo + 2u < q) ? 1u : 0u;
                            const uint32_t e3 = (nb > 3u && u3 == v && lo + 3u < q) ? 1u : 0u;
                            r = (u0 < v ? 1u : 0u) + ((nb > 1u && u1 < v) ? 1u : 0u) + ((nb > 2u && u2 < v) ? 1u : 0u) + ((nb > 3u && u3 < v) ? 1u : 0u) +
                                e0 + e1 + e2 + e3;
                            dup = e0 | e1 | e2 | e3;
                        }
                        for (uint32_t m = lo + 4u; m < hi; m++) {
                            const uint32_t u = sm.vals[1][m];
                            const uint32_t eq = (u == v && m < q) ? 1u : 0u;
                            r += (u < v ? 1u : 0u) + eq;
                            dup |= eq;
                        }
                        sm.vals[0][lo + r] = v;
                        uint32_t dead = dup;
                        if ((ts >> ((v >> 4) & 31u)) & 1u) dead |= (p.tomb[v >> 5] >> (v & 31u)) & 1u;     // rarely: the bitmap itself
                        sm.tids[0][lo + r] = (uint16_t)(dead << 15);   // dead: duplicate or tombstoned
                    }
                    __syncthreads();
                    nruns = 1u;
                    flagged = true;
                    II2_STAMP(1)      // E1: bucket fold
                }
            }
            // Batches of several terms: the same bucket sort with one bucket per posting, shared out among the
            // terms in proportion to their sizes — term t owns buckets [ttb[t], ttb[t+1]) (= its positions
            // after the fold) and maps its own doc range onto them, so the bucket order is the (term, doc) order.
            if (nt > 1u && k > 1u && nt <= MT) {
                const uint32_t *O = sm.offs[0];
                const uint32_t *RB = sm.runbase[0];
                uint32_t *c32 = &sm.offs[1][0];                 // two 16-bit counters per word
                {
                    uint32_t n_t = 0, mn = 0xFFFFFFFFu, mx = 0u;
                    if ((uint32_t)tid < nt) {
                        for (uint32_t r = 0; r < k; r++) {
                            const uint32_t o0 = O[r * stride + (uint32_t)tid], o1 = O[r * stride + (uint32_t)tid + 1u];
                            if (o1 > o0) {
                                n_t += o1 - o0;
                                const uint32_t f = sm.vals[0][RB[r] + o0], la = sm.vals[0][RB[r] + o1 - 1u];
                                mn = f < mn ? f : mn;
                                mx = la > mx ? la : mx;
                            }
                        }
                    }
                    uint32_t tot_;
                    const uint32_t tb = block_excl_scan(n_t, sm.wsum, &tot_);
                    if ((uint32_t)tid < nt) {
                        sm.ttb[tid] = (uint16_t)tb;
                        sm.tterm[tid] = make_uint2(mn, __float_as_uint(n_t ? (float)n_t / ((float)(mx - mn) + 1.0f) : 0.0f));
                    }
                    if (tid == 0) sm.ttb[nt] = (uint16_t)n_in;
                    __syncthreads();
                }
                auto bucket_mt = [&](uint32_t t, uint32_t v) -> uint32_t {
                    const uint32_t tb = sm.ttb[t], nb = sm.ttb[t + 1u] - tb;
                    const uint2 te = sm.tterm[t];
                    const uint32_t b = (uint32_t)((float)(v - te.x) * __uint_as_float(te.y));
                    return tb + (b < nb - 1u ? b : nb - 1u);
                };
                auto base_of = [&](uint32_t b) -> uint32_t { return (c32[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu; };
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t e = (uint32_t)tid + (uint32_t)j * MT;
                    if (e < n_in) {
                        const uint32_t t = sm.tids[0][e] & 1023u;
                        const uint32_t b = bucket_mt(t, sm.vals[0][e]);
                        const uint32_t sh = 16u * (b & 1u);
                        uint32_t slot = (atomicAdd(&c32[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
                        slot = slot < 15u ? slot : 15u;           // a fuller bucket sends the tile to the pairwise fold anyway
                        // from here on an element is known by its bucket (12 bits; the bucket implies the term): the
                        // flagged fold below and step F need no term tag, so the bucket is computed once
                        sm.tids[0][e] = (uint16_t)((slot << 12) | b);
                    }
                }
                __syncthreads();
                // exclusive scan of the counters in place (8 per thread) and the fullest bucket
                uint32_t c[8], sum = 0, mxc = 0;
                {
                    const uint4 w4 = reinterpret_cast<const uint4 *>(c32)[tid];
                    c[0] = w4.x & 0xFFFFu; c[1] = w4.x >> 16; c[2] = w4.y & 0xFFFFu; c[3] = w4.y >> 16;
                    c[4] = w4.z & 0xFFFFu; c[5] = w4.z >> 16; c[6] = w4.w & 0xFFFFu; c[7] = w4.w >> 16;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) { sum += c[j]; mxc = c[j] > mxc ? c[j] : mxc; }
                for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mxc, d, 64); mxc = o > mxc ? o : mxc; }
                if (l == 0) sm.wmax[wv] = mxc;
                uint32_t tot_;
                uint32_t run = block_excl_scan(sum, sm.wsum, &tot_);
                mxc = 0;
                for (int w = 0; w < (int)MW; w++) mxc = sm.wmax[w] > mxc ? sm.wmax[w] : mxc;
                if (mxc <= BKT_LIMIT_MT) {
                    uint32_t ex[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { ex[j] = run; run += c[j]; }
                    reinterpret_cast<uint4 *>(c32)[tid] = make_uint4(ex[0] | (ex[1] << 16), ex[2] | (ex[3] << 16), ex[4] | (ex[5] << 16), ex[6] | (ex[7] << 16));
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t e = (uint32_t)tid + (uint32_t)j * MT;
                        if (e < n_in) {
                            const uint32_t v = sm.vals[0][e];
                            const uint32_t tag = sm.tids[0][e];
                            const uint32_t dst = base_of(tag & 4095u) + (tag >> 12);
                            sm.vals[1][dst] = v;
                            sm.tids[1][dst] = (uint16_t)(tag & 4095u);
                        }
                    }
                    __syncthreads();
#pragma unroll 4
                    for (uint32_t q = (uint32_t)tid; q < n_in; q += MT) {
                        const uint32_t v = sm.vals[1][q];
                        const uint32_t b = sm.tids[1][q];
                        const uint32_t lo = base_of(b), hi = b + 1u < MCAP ? base_of(b + 1u) : n_in;
                        const uint32_t ts = (p.tomb && (v >> 5) < p.tomb_nwords) ? p.tomb_summary[v >> 9] : 0u;   // in flight during the loop
                        uint32_t r = 0, dup = 0;
                        {   // the first four of the bucket without a loop (buckets hold one posting on average: lanes that
                            // loop make the whole wave wait for the fullest bucket among its 64); reading past the bucket
                            // is harmless (masked), past vals[1] lands in the tag arrays
                            const uint32_t *B4 = &sm.vals[1][lo];
                            const uint32_t u0 = B4[0], u1 = B4[1], u2 = B4[2], u3 = B4[3];
                            const uint32_t nb = hi - lo;
                            const uint32_t e0 = (u0 == v && lo < q) ? 1u : 0u;
                            const uint32_t e1 = (nb > 1u && u1 == v && lo + 1u < q) ? 1u : 0u;
                            const uint32_t e2 = (nb > 2u && u2 == v && lo + 2u < q) ? 1u : 0u;
                            const uint32_t e3 = (nb > 3u && u3 == v && lo + 3u < q) ? 1u : 0u;
                            r = (u0 < v ? 1u : 0u) + ((nb > 1u && u1 < v) ? 1u : 0u) + ((nb > 2u && u2 < v) ? 1u : 0u) + ((nb > 3u && u3 < v) ? 1u : 0u) +
                                e0 + e1 + e2 + e3;
                            dup = e0 | e1 | e2 | e3;
                        }
                        for (uint32_t m = lo + 4u; m < hi; m++) {
                            const uint32_t u = sm.vals[1][m];
                            const uint32_t eq = (u == v && m < q) ? 1u : 0u;
                            r += (u < v ? 1u : 0u) + eq;
                            dup |= eq;
                        }
                        sm.vals[0][lo + r] = v;
                        uint32_t dead = dup;
                        if ((ts >> ((v >> 4) & 31u)) & 1u) dead |= (p.tomb[v >> 5] >> (v & 31u)) & 1u;     // rarely: the bitmap itself
                        sm.tids[0][lo + r] = (uint16_t)(dead << 15);         // bit 15: duplicate or tombstoned
                    }
                    __syncthreads();
                    nruns = 1u;
                    flagged = true;
                } else {
                    // clustered docs: back to (run, term) tags for the pairwise fold
                    __syncthreads();
                    for (uint32_t e = (uint32_t)tid; e < n_in; e += MT) {
                        uint32_t sa = 0, sb = k;
                        while (sb - sa > 1u) { const uint32_t sm_ = (sa + sb) >> 1; if (RB[sm_] <= e) sa = sm_; else sb = sm_; }
                        const uint32_t i = e - RB[sa];
                        const uint32_t *Os = O + sa * stride;
                        uint32_t ta = 0, tb2 = nt;
                        while (tb2 - ta > 1u) { const uint32_t tm = (ta + tb2) >> 1; if (Os[tm] <= i) ta = tm; else tb2 = tm; }
                        sm.tids[0][e] = (uint16_t)((sa << 10) | ta);
                    }
                    __syncthreads();
                }
            }
            // One term in the tile (the tiles of large terms — most of the postings): plain two-way merges.
            // Each thread produces a few consecutive outputs: one merge-path search to find where its
            // chunk starts in the two runs, then a sequential merge (A first on ties) — a handful of
            // instructions per posting and level instead of a binary search per posting.
            while (nt == 1u && nruns > 1u) {
                const uint32_t *V = sm.vals[cur];
                const uint32_t *RB = sm.runbase[cur];
                uint32_t *V2 = sm.vals[cur ^ 1u];
                const uint32_t npairs = (nruns + 1u) >> 1;
                const uint32_t VT = (n_in + MT - 1u) / MT;
                uint32_t o = (uint32_t)tid * VT;
                const uint32_t oe = o + VT < n_in ? o + VT : n_in;
                while (o < oe) {
                    uint32_t ja = 0, jb = npairs;             // pair j with RB[2j] <= o < RB[2j+2]
                    while (jb - ja > 1u) { const uint32_t jm = (ja + jb) >> 1; if (RB[2u * jm] <= o) ja = jm; else jb = jm; }
                    const uint32_t ra = 2u * ja, rb = ra + 1u;
                    const uint32_t abase = RB[ra], bbase = RB[rb];
                    const uint32_t la = bbase - abase;
                    const uint32_t lb = rb < nruns ? RB[rb + 1u] - bbase : 0u;
                    const uint32_t *A = V + abase, *B = V + bbase;
                    const uint32_t d = o - abase;
                    uint32_t lo = d > lb ? d - lb : 0u, hi = d < la ? d : la;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (A[mid] <= B[d - 1u - mid]) lo = mid + 1u; else hi = mid;
                    }
                    uint32_t ia = lo, ib = d - lo;
                    const uint32_t pend = abase + la + lb;
                    const uint32_t stop = oe < pend ? oe : pend;
                    uint32_t va = ia < la ? A[ia] : 0u, vb = ib < lb ? B[ib] : 0u;
                    for (; o < stop; o++) {
                        const bool takeA = ib >= lb || (ia < la && va <= vb);
                        V2[o] = takeA ? va : vb;
                        if (takeA) { ia++; va = ia < la ? A[ia] : 0u; }
                        else { ib++; vb = ib < lb ? B[ib] : 0u; }
                    }
                }
                __syncthreads();
                uint32_t *RB2 = sm.runbase[cur ^ 1u];
                for (uint32_t r2 = (uint32_t)tid; r2 <= npairs; r2 += MT) RB2[r2] = r2 < npairs ? RB[2u * r2] : n_in;
                __syncthreads();
                cur ^= 1u;
                nruns = npairs;
            }
            II2_STAMP(2)      // E2: pairwise merge of a single term (bucket overflow)
            while (nruns > 1u) {
                const uint32_t *V = sm.vals[cur];
                const uint16_t *T = sm.tids[cur];
                const uint32_t *O = sm.offs[cur];
                const uint32_t *RB = sm.runbase[cur];
                uint32_t *V2 = sm.vals[cur ^ 1u];
                uint16_t *T2 = sm.tids[cur ^ 1u];
                for (uint32_t e = (uint32_t)tid; e < n_in; e += MT) {
                    const uint32_t v = V[e];
                    const uint32_t tag = T[e];
                    const uint32_t r = tag >> 10, t = tag & 1023u;
                    const uint32_t ro = r ^ 1u;
                    const uint32_t myoff = O[r * stride + t];
                    const uint32_t i = e - RB[r] - myoff;
                    uint32_t rank = 0, poff = 0;
                    if (ro < nruns) {
                        poff = O[ro * stride + t];
                        const uint32_t plen = O[ro * stride + t + 1u] - poff;
                        const uint32_t *P = V + RB[ro] + poff;
                        uint32_t a = 0, b = plen;
                        if (r & 1u) {            // odd run: count partner elements <= v
                            while (a < b) { const uint32_t mid = (a + b) >> 1; if (P[mid] <= v) a = mid + 1u; else b = mid; }
                        } else {                 // even run: count partner elements < v
                            while (a < b) { const uint32_t mid = (a + b) >> 1; if (P[mid] < v) a = mid + 1u; else b = mid; }
                        }
                        rank = a;
                    }
                    const uint32_t dst = RB[r & ~1u] + myoff + poff + i + rank;
                    V2[dst] = v;
                    T2[dst] = (uint16_t)(((r >> 1) << 10) | t);
                }
                __syncthreads();
                const uint32_t nr2 = (nruns + 1u) >> 1;
                uint32_t *O2 = sm.offs[cur ^ 1u];
                uint32_t *RB2 = sm.runbase[cur ^ 1u];
                for (uint32_t e = (uint32_t)tid; e < nr2 * stride; e += MT) {
                    const uint32_t r2 = e / stride, t = e % stride;
                    const uint32_t ra = 2u * r2, rb = ra + 1u;
                    O2[e] = O[ra * stride + t] + (rb < nruns ? O[rb * stride + t] : 0u);
                }
                for (uint32_t r2 = (uint32_t)tid; r2 <= nr2; r2 += MT) RB2[r2] = r2 < nr2 ? RB[2u * r2] : n_in;
                __syncthreads();
                cur ^= 1u;
                nruns = nr2;
            }
            II2_STAMP(4)      // E: fold
            // ---- F. dedupe, tombstones, compact ----
            const uint32_t *V = sm.vals[cur];
            const uint16_t *T = sm.tids[cur];
            uint32_t *V2 = sm.vals[cur ^ 1u];
            uint16_t *T2 = sm.tids[cur ^ 1u];
            // eight consecutive postings per thread (MCAP = 8 * MT), read as vectors; the tombstone words
            // of all eight are fetched before any is tested
            const uint32_t a = 8u * (uint32_t)tid;
            uint32_t fv[8], ft[8];
            uint32_t keepmask = 0, cnt = 0;
            if (a < n_in && flagged) {
                const uint4 v0 = *reinterpret_cast<const uint4 *>(&V[a]), v1 = *reinterpret_cast<const uint4 *>(&V[a + 4u]);
                fv[0] = v0.x; fv[1] = v0.y; fv[2] = v0.z; fv[3] = v0.w; fv[4] = v1.x; fv[5] = v1.y; fv[6] = v1.z; fv[7] = v1.w;
                const uint4 t4 = *reinterpret_cast<const uint4 *>(&T[a]);
                // bit 15 of every 16-bit tag -> one bit per posting
                const uint32_t dead = ((t4.x >> 15) & 1u) | ((t4.x >> 30) & 2u) | (((t4.y >> 15) & 1u) << 2) | (((t4.y >> 30) & 2u) << 2) |
                                      (((t4.z >> 15) & 1u) << 4) | (((t4.z >> 30) & 2u) << 4) | (((t4.w >> 15) & 1u) << 6) | (((t4.w >> 30) & 2u) << 6);
                const uint32_t have = n_in - a >= 8u ? 0xFFu : ((1u << (n_in - a)) - 1u);
                keepmask = ~dead & have;
                cnt = (uint32_t)__popc(keepmask);
#pragma unroll
                for (int j = 0; j < 8; j++) ft[j] = 0;
            } else if (a < n_in) {
                const uint4 v0 = *reinterpret_cast<const uint4 *>(&V[a]), v1 = *reinterpret_cast<const uint4 *>(&V[a + 4u]);
                fv[0] = v0.x; fv[1] = v0.y; fv[2] = v0.z; fv[3] = v0.w; fv[4] = v1.x; fv[5] = v1.y; fv[6] = v1.z; fv[7] = v1.w;
                uint32_t pv = a ? V[a - 1u] : ~fv[0], pt = 0;
                if (nt > 1u) {
                    const uint4 t4 = *reinterpret_cast<const uint4 *>(&T[a]);
                    ft[0] = t4.x & 1023u; ft[1] = (t4.x >> 16) & 1023u; ft[2] = t4.y & 1023u; ft[3] = (t4.y >> 16) & 1023u;
                    ft[4] = t4.z & 1023u; ft[5] = (t4.z >> 16) & 1023u; ft[6] = t4.w & 1023u; ft[7] = (t4.w >> 16) & 1023u;
                    pt = a ? (T[a - 1u] & 1023u) : 0u;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) ft[j] = 0;
                }
                uint32_t tw[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t w = fv[j] >> 5;
                    tw[j] = 0u;
                    if (p.tomb && a + (uint32_t)j < n_in && w < p.tomb_nwords && ((p.tomb_summary[fv[j] >> 9] >> ((fv[j] >> 4) & 31u)) & 1u)) tw[j] = p.tomb[w];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool dup = fv[j] == pv && ft[j] == pt && (a + (uint32_t)j) > 0u;
                    const bool keep = a + (uint32_t)j < n_in && !dup && !((tw[j] >> (fv[j] & 31u)) & 1u);
                    if (keep) { keepmask |= 1u << j; cnt++; }
                    pv = fv[j];
                    pt = ft[j];
                }
            }
            uint32_t tot;
            uint32_t pos = block_excl_scan(cnt, sm.wsum, &tot);
            const uint32_t pos0 = pos;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if ((keepmask >> j) & 1u) { V2[pos] = fv[j]; if (!flagged) T2[pos] = (uint16_t)ft[j]; pos++; }
            }
            *outbuf = cur ^ 1u;
            if (emit_counts && nt == 1u) {
                if (tid == 0 && tot) {
                    if (atomic_counts) atomicAdd(&p.out_counts[t0], tot);
                    else p.out_counts[t0] = tot;
                }
            } else if (emit_counts && flagged) {
                // term t sat at [ttb[t], ttb[t+1]) before the compaction: survivors = difference of the keep prefix there
                uint2 *kp = reinterpret_cast<uint2 *>(&sm.offs[0][0]);
                kp[tid] = make_uint2(pos0, keepmask);
                __syncthreads();
                if ((uint32_t)tid < nt) {
                    auto kept_before = [&](uint32_t x) -> uint32_t {
                        if (x >= MCAP) return tot;
                        const uint2 e = kp[x >> 3];
                        return e.x + (uint32_t)__popc(e.y & ((1u << (x & 7u)) - 1u));
                    };
                    const uint32_t c = kept_before(sm.ttb[tid + 1]) - kept_before(sm.ttb[tid]);
                    if (c) {
                        if (atomic_counts) atomicAdd(&p.out_counts[t0 + (uint32_t)tid], c);
                        else p.out_counts[t0 + (uint32_t)tid] = c;
                    }
                }
            } else if (emit_counts) {
                // per-term survivor counts from the boundaries of the compacted array
                uint32_t *tstart = sm.offs[0], *tend = sm.offs[1];
                __syncthreads();
                for (uint32_t t = (uint32_t)tid; t < nt; t += MT) { tstart[t] = 0; tend[t] = 0; }
                __syncthreads();
                for (uint32_t q = (uint32_t)tid; q < tot; q += MT) {
                    const uint32_t t = T2[q];
                    if (q == 0 || T2[q - 1] != t) tstart[t] = q;
                    if (q + 1u == tot || T2[q + 1] != t) tend[t] = q + 1u;
                }
                __syncthreads();
                for (uint32_t t = (uint32_t)tid; t < nt; t += MT) {
                    const uint32_t c = tend[t] - tstart[t];
                    if (c) {
                        if (atomic_counts) atomicAdd(&p.out_counts[t0 + t], c);
                        else p.out_counts[t0 + t] = c;
                    }
                }
            }
            __syncthreads();
            II2_STAMP(5)      // F: dedupe / filter / compact / counts
            return tot;
        };

        const uint32_t dlo = td.z, dhi = td.w;
        const bool root_full = dlo == 0u && dhi == 0xFFFFFFFFu;
        // The tile parks its survivors in the scratch array at the input rank of its first posting
        // (term slots start at the prefix of the terms' input counts), which no other tile can reach:
        // survivors never outnumber the inputs that precede the next tile.  A later pass packs them.
        // (Tried in round 2: writing survivors straight to their final place through a chained scan over the tiles — with
        // ~500 tiles in flight every tile waits for its slowest predecessor; the scan + write-out took 61 % of a workgroup's
        // time and the merge went from 4.3 to 7.3 ms.  Tiles must stay independent.)
        const unsigned long long term_slot = p.ub_prefix[t0];
        unsigned long long slot = term_slot;
        uint32_t total = 0;
        if (dlo <= dhi) {
            uint32_t outbuf = 0;
            const bool fits = load_range(dlo, dhi, true);
            slot = term_slot + sm.rank;                  // rank of the range start (0 for whole-term tiles)
            if (fits) {
                // range tiles of a large term leave its count to k_merge_large_counts: an atomicAdd per tile would put
                // thousands of same-address device atomics in flight (the top terms own most tiles)
                total = merge_range(&outbuf, root_full, false);
                const uint32_t *V = outbuf == 2u ? reinterpret_cast<const uint32_t *>(&sm.tids[0][0]) : sm.vals[outbuf];
                for (uint32_t q = (uint32_t)tid; q < total; q += MT) p.tmp[slot + q] = V[q];
            } else {
                // the range holds more than LDS (a term whose lists are clustered differently): bisect the
                // doc range; leaves are handled in doc order and appended to the tile's slot.
                uint32_t sp = 1;
                __syncthreads();
                if (tid == 0) { sm.stk[0][0] = dlo; sm.stk[0][1] = dhi; }
                while (sp > 0) {
                    __syncthreads();
                    const uint32_t lo = sm.stk[sp - 1][0], hi = sm.stk[sp - 1][1];
                    sp--;
                    if (!load_range(lo, hi, false)) {
                        // lo < hi here: a single doc id never exceeds k postings
                        const uint32_t mid = lo + ((hi - lo) >> 1);
                        __syncthreads();
                        if (tid == 0) {
                            sm.stk[sp][0] = mid + 1u; sm.stk[sp][1] = hi;
                            sm.stk[sp + 1][0] = lo;   sm.stk[sp + 1][1] = mid;
                        }
                        sp += 2;
                        continue;
                    }
                    uint32_t ob2 = 0;
                    const uint32_t c = merge_range(&ob2, nt > 1u, true);       // single-term leaves: counted from the tile totals
                    const uint32_t *V = ob2 == 2u ? reinterpret_cast<const uint32_t *>(&sm.tids[0][0]) : sm.vals[ob2];
                    for (uint32_t q = (uint32_t)tid; q < c; q += MT) p.tmp[slot + total + q] = V[q];
                    total += c;
                }
            }
        }
        if (tid == 0) { p.tile_count[tile] = total; p.tile_slot[tile] = slot; }
        __syncthreads();
        II2_STAMP(6)          // G: park survivors
    }
    if (stamps && tid == 0)
        for (int i = 0; i < 8; i++) p.debug[(uint64_t)blockIdx.x * 8u + i] = tacc[i];
#undef II2_STAMP
}

// packs the parked survivors: tile t's ids go to out[off[t] ...]
__global__ __launch_bounds__(256) void k_merge_pack(const uint32_t *__restrict__ tmp, const unsigned long long *__restrict__ slot,
                                                    const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ off, const uint32_t *__restrict__ n_tiles_dev,
                                                    uint32_t *__restrict__ out, uint64_t out_cap, uint64_t *__restrict__ d_total) {
    const uint32_t n_tiles = *n_tiles_dev;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t c = cnt[tile];
        const uint64_t ob = off[tile];
        const uint32_t *src = tmp + slot[tile];
        for (uint32_t q = threadIdx.x; q < c; q += 256u)
            if (ob + q < out_cap) out[ob + q] = src[q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_total = off[n_tiles];
}

// survivors of every large term = survivors of its tiles (tile_off = exclusive scan of the tile counts)
__global__ void k_merge_large_counts(const uint32_t *__restrict__ ntl, const uint32_t *__restrict__ term_tile, const uint64_t *__restrict__ tile_off,
                                     uint64_t n_terms, uint32_t *__restrict__ out_counts) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms || ntl[t] == 0u) return;
    const uint32_t a = term_tile[t];
    out_counts[t] = (uint32_t)(tile_off[a + ntl[t]] - tile_off[a]);
}

// one atomic per workgroup, few workgroups: a single address sustains only ~90 device atomics per microsecond
__global__ __launch_bounds__(256) void k_count_nonzero(const uint32_t *__restrict__ v, uint64_t n, uint64_t *__restrict__ out) {
    __shared__ uint32_t wsum[4];
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += v[i] != 0;
    c = wave_sum(c);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd((unsigned long long *)out, (unsigned long long)t);
    }
}

static unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_mseg_blocks(const MergeSegs &p, uint32_t *segtab, hipStream_t s) {
    hipLaunchKernelGGL(k_mseg_blocks, dim3(1), dim3(64), 0, s, p, segtab);
    return hipGetLastError();
}
hipError_t launch_mlist_counts(const MergeSegs &p, uint32_t *lc, hipStream_t s) {
    hipLaunchKernelGGL(k_mlist_counts, dim3(grid_for((uint64_t)p.k * (p.n_terms + 1))), dim3(256), 0, s, p, lc);
    return hipGetLastError();
}
hipError_t launch_mbig_count(const MergeSegs &p, uint32_t *wgcnt, hipStream_t s) {
    const uint64_t total = p.total_ub;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mbig_count, dim3(grid_for(total)), dim3(256), 0, s, p, wgcnt);
    return hipGetLastError();
}
hipError_t launch_mdec_write(const MergeSegs &p, const unsigned long long *poff, uint32_t *raw, const uint32_t *wgbase, void *ent0, void *ent1,
                             uint32_t grid_rows, hipStream_t s) {
    const uint64_t total = p.total_ub;
    if (total == 0) return hipSuccess;
    const unsigned nwg = grid_for(total);
    hipLaunchKernelGGL(k_mdec_lane, dim3(nwg), dim3(256), 0, s, p, poff, raw, wgbase, (uint4 *)ent0, (uint2 *)ent1);
    hipLaunchKernelGGL(k_mdec_rows, dim3(grid_rows), dim3(256), 0, s, p, raw, (const uint4 *)ent0, (const uint2 *)ent1, wgbase + nwg);
    return hipGetLastError();
}
hipError_t launch_merge_tile_ranges(const MergeParams &p, const MergeSegs &ms, const void *desc, uint32_t *ends, void *rng, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    const unsigned g = grid_for((uint64_t)p.n_tiles_ub * p.k);
    hipLaunchKernelGGL(k_merge_tile_ends, dim3(g), dim3(256), 0, s, p, ms, (const uint4 *)desc, ends);
    hipLaunchKernelGGL(k_merge_tile_ranges, dim3(g), dim3(256), 0, s, p, (const uint4 *)desc, (const uint32_t *)ends, (uint4 *)rng);
    return hipGetLastError();
}

hipError_t launch_merge_plan1(const MergeParams &p, uint32_t *ub, uint32_t *weight, uint32_t *ntl, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_term_ub, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, p, ub, weight, ntl);
    return hipGetLastError();
}
hipError_t launch_merge_heads(const MergeParams &p, const uint32_t *ntl, const uint64_t *wpre, uint32_t *head, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_heads, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, p, ntl, wpre, head);
    return hipGetLastError();
}
hipError_t launch_merge_term_tile(const MergeParams &p, const uint32_t *ntl, const uint32_t *head, const uint32_t *hpre,
                                  const uint32_t *lpre, uint32_t *term_tile, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_term_tile, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, p, ntl, head, hpre, lpre, term_tile);
    return hipGetLastError();
}
hipError_t launch_merge_tile_desc(const MergeParams &p, const uint32_t *ntl, const uint32_t *term_tile, void *desc, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_tile_desc, dim3(grid_for(p.n_tiles_ub)), dim3(256), 0, s, p, ntl, term_tile, (uint4 *)desc);
    return hipGetLastError();
}
hipError_t launch_merge_large_counts(const MergeParams &p, const uint32_t *ntl, const uint32_t *term_tile, const uint64_t *tile_off, hipStream_t s) {
    if (p.n_terms == 0 || p.n_tiles_ub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_large_counts, dim3(grid_for(p.n_terms)), dim3(256), 0, s, ntl, term_tile, tile_off, p.n_terms, p.out_counts);
    return hipGetLastError();
}
hipError_t launch_merge_pack(const MergeParams &p, const uint64_t *tile_off, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    const uint32_t g = p.n_tiles_ub < 16384u ? p.n_tiles_ub : 16384u;
    hipLaunchKernelGGL(k_merge_pack, dim3(g), dim3(256), 0, s, (const uint32_t *)p.tmp, (const unsigned long long *)p.tile_slot,
                       (const uint32_t *)p.tile_count, tile_off, p.n_tiles_dev, p.out_values, p.out_cap, p.d_total);
    return hipGetLastError();
}
hipError_t launch_merge_tiles(const MergeParams &p, const void *tile_desc, uint32_t grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    if (ev0) (void)hipEventRecord(ev0, s);
    if (p.k == 16u) hipLaunchKernelGGL(k_merge_tiles<16u>, dim3(grid < p.n_tiles_ub ? grid : p.n_tiles_ub), dim3(MERGE_THREADS), 0, s, p, (const uint4 *)tile_desc);
    else hipLaunchKernelGGL(k_merge_tiles<0u>, dim3(grid < p.n_tiles_ub ? grid : p.n_tiles_ub), dim3(MERGE_THREADS), 0, s, p, (const uint4 *)tile_desc);
    if (ev1) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}
hipError_t launch_count_nonzero(const uint32_t *v, uint64_t n, uint64_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = grid_for(n);
    if (g > 128) g = 128;
    hipLaunchKernelGGL(k_count_nonzero, dim3(g), dim3(256), 0, s, v, n, out);
    return hipGetLastError();
}

}  // namespace ii2
