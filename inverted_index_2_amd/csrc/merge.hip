// merge.hip — segmented k-way union of DV1 posting lists with fused tombstone filter
// (gfx950, wave64, no MFMA).  One kernel family serves
//   * the segment merge: the body of Shard.Merge's loop (reference shard.go:163-212) and the
//     k-way merging iterator it drains (shard.go:253-278) folding equal terms with
//     file.MergeTermValues (file/types.go:14-22) — for every aligned term the sorted,
//     duplicate-free union of the k segments' lists, minus the tombstones (shard.go:181-190),
//     empty terms reported with count 0 (shard.go:192-194);
//   * the multi-term union of PrefixSearch (inverted_index.go:274-292) — one "term", k lists.
//
// Round 3: every posting crosses HBM once on its way in.  There is no decode pass and no raw scratch array: the plan
// is made from what the segments already hold (per-list posting counts, first / last docs, skip tables) and the tile
// kernel decodes the DV1 blocks of its tile straight into LDS.  Three kinds of tile:
//   batch    consecutive small terms, whole lists (<= MERGE_CAP postings in all);
//   range    a doc-id range of a large term (ranges are cut at block boundaries of the term's longest list, so they
//            need no decoded data; every list's blocks that overlap a range are found in its skip table);
//   bitmap   a fixed doc-id range (MERGE_BM_DOCS docs) of a term dense enough for a bitmap over the range.
// Batch and range tiles sort by buckets: a monotone map of the doc id (per term) onto ~1 bucket per posting, slot
// inside the bucket from an LDS counter, exclusive scan of the counters, then every posting ranks itself among the few
// that share its bucket; duplicates and tombstoned ids set a bit in a "dead" mask over the sorted positions and every
// survivor goes straight from its register to its final rank in the output (no compaction pass, no sorted copy).
// Bitmap tiles mark, clear the tombstoned words and extract.  A bucket that overflows (clustered ids) or a range that
// holds more than LDS sends the range to a bisection whose leaves are small enough for the bitmap — exact for any input.
// Tiles are independent: each parks its survivors in a scratch array (batches at the input rank of their first term,
// tiles of a large term through a bump allocator inside the term's region); a scan of the tile counts and a packing
// pass then produce the CSR the reference's writer would have been fed: terms ascending, ids ascending.
#include <type_traits>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t MCAP = MERGE_CAP;
constexpr uint32_t MT = MERGE_THREADS;
constexpr uint32_t MW = MT / 64u;                 // waves per workgroup
constexpr uint32_t EPT = MCAP / MT;               // elements per thread in the sort passes (14)
constexpr uint32_t PCAP = 1024;                   // 16-byte payload pieces decoded per chunk of blocks
constexpr uint32_t BKT_LIMIT = 15;                // fullest bucket the bucket sort accepts (slot numbers are 4 bits)
constexpr uint32_t BMW = MERGE_BM_WORDS;
static_assert(MCAP % MT == 0 && EPT * 4u <= 64u && (MCAP / 2u) % MT == 0, "sort passes: EPT elements and EPT / 2 counter words per thread");

// ---- plan ----------------------------------------------------------------------------------
// per term: input postings, doc range, longest list, and how it will be merged
__global__ __launch_bounds__(256) void k_mp_terms(const MergeSegs *__restrict__ ms, MergeParams p) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.n_terms) return;
    if (t == p.n_terms) { p.tn[t] = 0; p.tmin[t] = 0; p.tmax[t] = 0; p.tinfo[t] = 0; p.weight[t] = 0; p.ntl[t] = 0; return; }
    uint64_t n = 0;
    uint32_t mn = 0xFFFFFFFFu, mx = 0u, best = 0, best_nb = 0;
    for (uint32_t s = 0; s < p.k; s++) {
        const SegView &sv = ms->segs[s];
        const uint32_t b0 = sv.blk_off[t], b1 = sv.blk_off[t + 1];
        if (b1 > b0) {
            n += sv.cnt[t];
            const uint32_t f = sv.skip[b0].first_doc, la = sv.last_doc[t];
            mn = f < mn ? f : mn;
            mx = la > mx ? la : mx;
            if (b1 - b0 > best_nb) { best_nb = b1 - b0; best = s; }
        }
    }
    const uint32_t n32 = n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;
    if (n == 0) { mn = 0; mx = 0; }
    if (mx < mn) mx = mn;                              // (lists that are not ascending: keep the range well-formed)
    p.tn[t] = n32;
    p.tmin[t] = mn;
    p.tmax[t] = mx;
    uint32_t info = best, w = 0, tiles = 0;
    if (n32 > p.small_max) {
        const uint64_t span = (uint64_t)mx - (mn & ~31u) + 1ull;
        if (p.bitmap_tiles && span <= (uint64_t)n32 * p.bitmap_sparsity) {
            info |= 1u << 8;
            tiles = (uint32_t)((span + MERGE_BM_DOCS - 1) / MERGE_BM_DOCS);
        } else {
            tiles = (uint32_t)(((uint64_t)n32 + p.range_target - 1) / p.range_target);
            if (best_nb < 2u * tiles) info |= 1u << 9;     // too few blocks to cut at: splitters uniform in doc space
        }
    } else {
        w = n32 > p.wmin ? n32 : p.wmin;
    }
    p.tinfo[t] = info;
    p.weight[t] = w;
    p.ntl[t] = tiles;
}

// head[t] = 1 when small term t opens a new batch
__global__ void k_merge_heads(MergeParams p, const uint64_t *__restrict__ wpre, uint32_t *__restrict__ head) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.n_terms) return;
    if (t == p.n_terms || p.ntl[t] > 0) { head[t] = 0; return; }
    bool h = t == 0 || p.ntl[t - 1] > 0;
    if (!h) h = (wpre[t] / p.batch_q) != (wpre[t - 1] / p.batch_q);
    head[t] = h ? 1u : 0u;
}

// term_tile[t] = id of the (first) tile of term t; monotone in t; term_tile[T] = number of tiles
__global__ void k_merge_term_tile(MergeParams p, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hpre,
                                  const uint32_t *__restrict__ lpre, uint32_t *__restrict__ term_tile) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.n_terms) return;
    if (t == p.n_terms || p.ntl[t] > 0) term_tile[t] = hpre[t] + lpre[t];
    else term_tile[t] = hpre[t] + head[t] - 1u + lpre[t];
}

// tile descriptors {t0, t1 | flags, dlo, dhi}
__global__ void k_merge_tile_desc(const MergeSegs *__restrict__ ms, MergeParams p) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= *p.n_tiles_dev) return;
    const uint32_t *term_tile = p.term_tile;
    uint64_t lo = 0, hi = p.n_terms;           // term_tile[lo] <= tile < term_tile[hi] (term_tile[n_terms] = n_tiles)
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (term_tile[mid] <= tile) lo = mid; else hi = mid;
    }
    const uint64_t tl = lo;
    if (p.ntl[tl] > 0) {                        // tile j of large term tl
        const uint32_t m = p.ntl[tl], j = tile - term_tile[tl], info = p.tinfo[tl];
        const uint32_t mn = p.tmin[tl], mx = p.tmax[tl];
        uint32_t flags = MERGE_DESC_LARGE;
        uint64_t lo64, hi64;                    // the tile covers [lo64, hi64)
        if (info & (1u << 8)) {                 // bitmap tiles: fixed windows from the term's first doc (rounded down to a word)
            flags |= MERGE_DESC_BITMAP;
            lo64 = (uint64_t)(mn & ~31u) + (uint64_t)j * MERGE_BM_DOCS;
            hi64 = lo64 + MERGE_BM_DOCS;
            if (hi64 > (1ull << 32)) hi64 = 1ull << 32;
        } else if (m == 1u) {
            lo64 = 0; hi64 = 1ull << 32;
        } else {
            auto splitter = [&](uint32_t jj) -> uint64_t {
                if (jj == 0) return 0ull;
                if (jj >= m) return 1ull << 32;
                if (info & (1u << 9)) return (uint64_t)mn + ((uint64_t)jj * ((uint64_t)mx - mn + 1ull)) / m;
                const SegView &sv = ms->segs[info & 0xFFu];
                const uint32_t b0 = sv.blk_off[tl], nb = sv.blk_off[tl + 1] - b0;
                return (uint64_t)sv.skip[b0 + (uint32_t)(((uint64_t)jj * nb) / m)].first_doc;      // first docs ascend: non-decreasing in jj
            };
            lo64 = splitter(j);
            hi64 = splitter(j + 1u);
        }
        uint32_t dlo, dhi;
        if (hi64 <= lo64) { dlo = 1u; dhi = 0u; }                                 // empty range
        else { dlo = (uint32_t)lo64; dhi = (uint32_t)(hi64 - 1ull); }
        p.desc[tile] = make_uint4((uint32_t)tl, ((uint32_t)tl + 1u) | flags, dlo, dhi);
    } else {
        // batch: terms [first with term_tile == tile, last with term_tile == tile]
        uint64_t a = 0, b = tl;                 // find first term with term_tile >= tile
        if (term_tile[0] >= tile) b = 0;
        else {
            while (b - a > 1) {                 // term_tile[a] < tile <= term_tile[b]
                const uint64_t mid = a + ((b - a) >> 1);
                if (term_tile[mid] < tile) a = mid; else b = mid;
            }
        }
        p.desc[tile] = make_uint4((uint32_t)b, (uint32_t)tl + 1u, 0u, 0xFFFFFFFFu);
    }
}

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

// blocks [b0, b1) of a list (blocks [b_lo, b_hi) of its segment) that may hold docs of [dlo, dhi]: from the last block
// that starts at or before dlo up to the last block that starts at or before dhi
__device__ __forceinline__ uint2 blocks_of_range(const ii2_skip *__restrict__ skip, uint32_t b_lo, uint32_t b_hi, uint32_t dlo, uint32_t dhi) {
    if (b_hi <= b_lo || dlo > dhi) return make_uint2(b_lo, b_lo);
    const uint32_t f0 = skip[b_lo].first_doc, fl = skip[b_hi - 1u].first_doc;
    auto get = [&](uint32_t j) { return skip[j].first_doc; };
    const uint32_t a = upper_bound_guess(get, b_lo, b_hi, dlo, f0, fl);
    const uint32_t b1 = dhi == 0xFFFFFFFFu ? b_hi : upper_bound_guess(get, a, b_hi, dhi, dlo, fl);
    return make_uint2(a > b_lo ? a - 1u : b_lo, b1);
}

// runs[tile * k + s]: the blocks of list (s, t0) a range tile has to decode (tiles that take whole lists read blk_off)
__global__ void k_merge_tile_runs(const MergeSegs *__restrict__ ms, MergeParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)*p.n_tiles_dev * p.k) return;
    const uint32_t tile = (uint32_t)(i / p.k), s = (uint32_t)(i % p.k);
    const uint4 td = p.desc[tile];
    if (td.z == 0u && td.w == 0xFFFFFFFFu) return;            // whole lists: nothing to search
    const SegView &sv = ms->segs[s];
    p.runs[i] = blocks_of_range(sv.skip, sv.blk_off[td.x], sv.blk_off[td.x + 1u], td.z, td.w);
}

// ---- the tile kernel ----------------------------------------------------------------------
struct __align__(16) MergeSmem {
    union {
        struct {
            uint32_t V[MCAP];               // the tile's postings: arrival order, then bucket order
            uint16_t TG[MCAP];              // per posting: term slot (while decoding), then its bucket
            uint32_t C32[MCAP / 2u + 4u];   // bucket counters, then exclusive bucket bases: two 16-bit values per word
        } s;
        uint32_t bm[BMW];                   // bitmap tiles: one bit per doc of the tile's range
    } u;
    union {
        struct {                            // while a chunk of blocks is decoded
            uint32_t PX[PCAP];              // exclusive prefix of the pieces' gap sums
            uint32_t BF[MT], BQ[MT], BI[MT];   // per block: first doc, payload offset, run | term slot << 6 | payload bytes << 15
            uint16_t PB[MT + 2u];           // per block: its first piece
        } d;
        struct {                            // while a decoded tile is sorted
            uint2 TT[MERGE_NT_MAX];         // per term of a batch: {smallest doc, float bits of buckets per doc}
            uint16_t TB[MERGE_NT_MAX + 4u]; // per term: its first bucket
            uint32_t DB[MCAP / 32u + 1u];   // dead mask over the sorted positions: duplicates and tombstoned ids
            uint32_t DP[MCAP / 32u + 2u];   // exclusive prefix of the dead mask's popcounts
        } f;
    } x;
    const uint8_t *pay[MAX_LISTS];
    const ii2_skip *skp[MAX_LISTS];
    const uint32_t *bls[MAX_LISTS];
    const uint32_t *bof[MAX_LISTS];
    uint32_t lbase[MAX_LISTS];
    uint32_t RR0[MAX_LISTS], RR1[MAX_LISTS];   // block range of each run for the tile's root doc range
    uint32_t R0[MAX_LISTS];                 // first block of each run for the range being merged
    uint32_t RB[MAX_LISTS + 2u];            // exclusive prefix of the runs' block counts
    uint32_t wsum[MW];
    uint32_t fill;                          // postings of the range that arrived in V (may exceed MCAP: the range is then split)
    uint32_t ovf;                           // a bucket overflowed
    uint32_t tp;                            // pieces of the chunk
    uint32_t ab;                            // allocation inside the term's parking region
    uint32_t stk[72][2];                    // bisection stack of doc ranges
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

__global__ __launch_bounds__(MERGE_THREADS, 4) void k_merge_tiles(const MergeSegs *__restrict__ ms, MergeParams p) {
    __shared__ MergeSmem sm;
    const int tid = (int)threadIdx.x, l = tid & 63;
    const uint32_t k = p.k;
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

    if ((uint32_t)tid < k) {
        const SegView &sv = ms->segs[tid];
        sm.pay[tid] = sv.payload;
        sm.skp[tid] = sv.skip;
        sm.bls[tid] = sv.blk_list;
        sm.bof[tid] = sv.blk_off;
        sm.lbase[tid] = sv.list_base;
    }
    __syncthreads();

    // ---- the runs of a doc range: which blocks of every list have to be decoded.
    // mode 0: whole lists of the terms [t0, t1); 1: the tile's root range (searched by the plan); 2: a sub-range of the root
    auto setup_runs = [&](uint32_t mode, uint32_t t0, uint32_t t1, uint32_t tile, uint32_t lo, uint32_t hi) {
        __syncthreads();
        if ((uint32_t)tid < k) {
            uint2 r;
            if (mode == 0u) r = make_uint2(sm.bof[tid][t0], sm.bof[tid][t1]);
            else if (mode == 1u) r = p.runs[(uint64_t)tile * k + (uint32_t)tid];
            else r = blocks_of_range(sm.skp[tid], sm.RR0[tid], sm.RR1[tid], lo, hi);
            if (mode != 2u) { sm.RR0[tid] = r.x; sm.RR1[tid] = r.y; }
            sm.R0[tid] = r.x;
            sm.RB[tid] = r.y - r.x;            // (count; scanned below)
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t c = (uint32_t)l < k ? sm.RB[l] : 0u;
            const uint32_t incl = wave_incl_scan(c);
            if ((uint32_t)l < k) sm.RB[l] = incl - c;
            if (l == 63) sm.RB[k] = incl;
            if (l == 0) { sm.fill = 0u; sm.ovf = 0u; }
        }
        __syncthreads();
    };

    // ---- decode the runs' blocks; every posting with lo <= id <= hi goes
    //   BM = false: to sm.u.s.V[arrival order] (and its term slot to TG when the tile is a batch),
    //   BM = true:  into the bitmap of the range (bit id - lo32).
    // A chunk = as many consecutive blocks (one per thread) as have PCAP 16-byte payload pieces between them; a thread then
    // walks one piece: the 16 partial gap sums and which bytes end a posting; a scan over the chunk's pieces gives every
    // piece the sum before it, and the difference to its block's first piece the id it starts from.
    auto decode = [&](auto bm_tag, uint32_t lo, uint32_t hi, uint32_t lo32, bool batch, uint32_t t0, uint32_t nt, bool filter) {
        constexpr bool BM = decltype(bm_tag)::value;
        const uint32_t NB = sm.RB[k];
        uint32_t g0 = 0;
        while (g0 < NB) {
            __syncthreads();                     // the chunk before is done with the tables
            const uint32_t g = g0 + (uint32_t)tid;
            const bool valid = g < NB;
            uint32_t s = 0, np = 0, first = 0, q0 = 0, ti = 0, len = 0;
            if (valid) {
                uint32_t a = 0, e = k;            // RB[a] <= g < RB[e]
                while (e - a > 1u) { const uint32_t m = (a + e) >> 1; if (sm.RB[m] <= g) a = m; else e = m; }
                s = a;
                const uint32_t b = sm.R0[s] + (g - sm.RB[s]);
                const ii2_skip *sk = sm.skp[s];
                const ii2_skip e0 = sk[b];
                first = e0.first_doc;
                q0 = e0.byte_off;
                len = sk[b + 1u].byte_off - q0;
                if (len > 1280u) len = 1280u;     // (a block holds <= 256 postings of <= 5 bytes; imported segments are validated)
                np = (len + 15u) >> 4;
                if (batch) { ti = sm.bls[s][b] - sm.lbase[s] - t0; ti = ti < nt ? ti : nt - 1u; }
            }
            uint32_t tot;
            const uint32_t pex = block_excl_scan(np, sm.wsum, &tot);
            const bool ok = valid && pex + np <= PCAP;          // a prefix of the threads (pex ascends); never empty (np <= 80)
            const uint32_t nchunk = (uint32_t)__syncthreads_count(ok ? 1 : 0);
            if (ok) {
                sm.x.d.BF[tid] = first;
                sm.x.d.BQ[tid] = q0;
                sm.x.d.BI[tid] = s | (ti << 6) | (len << 15);
                sm.x.d.PB[tid] = (uint16_t)pex;
                if ((uint32_t)tid == nchunk - 1u) sm.tp = pex + np;
            }
            {   // the blocks' first postings (their ids are in the skip entries)
                const bool in = ok && first >= lo && first <= hi;
                if (BM) {
                    if (in) atomicOr(&sm.u.bm[(first - lo32) >> 5], 1u << (first & 31u));
                } else {
                    const unsigned long long m = __ballot(in);
                    if (m != 0ull) {
                        uint32_t wb = 0;
                        if (l == 0) wb = atomicAdd(&sm.fill, (uint32_t)__popcll(m));
                        wb = wave_bcast(wb, 0);
                        const uint32_t pos = wb + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
                        if (in && pos < MCAP) { sm.u.s.V[pos] = first; if (batch) sm.u.s.TG[pos] = (uint16_t)ti; }
                    }
                }
            }
            __syncthreads();
            const uint32_t tp = sm.tp;
            uint32_t carry = 0;
            for (uint32_t it = 0; it < tp; it += MT) {
                const uint32_t pc = it + (uint32_t)tid;
                const bool pv = pc < tp;
                uint32_t jb = 0, val[16], tmask = 0, bi = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) val[i] = 0;
                if (pv) {
                    uint32_t a = 0, e = nchunk;   // last block with PB <= pc (blocks without payload share their successor's PB)
                    while (e - a > 1u) { const uint32_t m = (a + e) >> 1; if ((uint32_t)sm.x.d.PB[m] <= pc) a = m; else e = m; }
                    jb = a;
                    bi = sm.x.d.BI[jb];
                    const uint32_t off = 16u * (pc - (uint32_t)sm.x.d.PB[jb]);
                    const uint32_t blen = bi >> 15;
                    const uint32_t nb = blen - off < 16u ? blen - off : 16u;
                    const uint8_t *pp = sm.pay[bi & 63u] + sm.x.d.BQ[jb] + off;
                    uint4 w4;
                    __builtin_memcpy(&w4, pp, 16);                        // (segments carry 16 bytes of padding)
                    const uint32_t prev = off ? load_u32_unaligned(pp - 4) : 0u;
                    uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
                    if (nb < 16u) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t nj = nb > 4u * j ? (nb - 4u * j < 4u ? nb - 4u * j : 4u) : 0u;
                            w[j] &= nj >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nj)) - 1u);
                        }
                    }
                    // continuation bytes pending right before my first byte (varints are <= 5 bytes)
                    uint32_t sh = 7u * ((uint32_t)__clz((int)~(prev | 0x7F7F7F7Fu)) >> 3);
                    uint32_t sum = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                        sum += (c & 0x7Fu) << sh;
                        if (c & 0x80u) sh = sh < 28u ? sh + 7u : 28u;
                        else { sh = 0u; tmask |= 1u << i; }
                        val[i] = sum;
                    }
                    tmask &= (1u << nb) - 1u;
                }
                uint32_t tots;
                const uint32_t pex2 = carry + block_excl_scan(val[15], sm.wsum, &tots);
                carry += tots;
                if (pv) sm.x.d.PX[pc] = pex2;
                __syncthreads();
                if (pv) {
                    const uint32_t base = sm.x.d.BF[jb] + pex2 - sm.x.d.PX[sm.x.d.PB[jb]];
                    if (BM) {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const uint32_t id = base + val[i];
                            if (((tmask >> i) & 1u) && id >= lo && id <= hi) atomicOr(&sm.u.bm[(id - lo32) >> 5], 1u << (id & 31u));
                        }
                    } else if (filter) {
                        uint32_t im = 0;
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const uint32_t id = base + val[i];
                            if (id >= lo && id <= hi) im |= 1u << i;
                        }
                        tmask &= im;
                    }
                }
                if (!BM) {
                    const uint32_t c = (uint32_t)__popc(tmask);
                    const uint32_t incl = wave_incl_scan(c);
                    const uint32_t wtot = wave_bcast(incl, 63);
                    if (wtot) {
                        uint32_t wb = 0;
                        if (l == 0) wb = atomicAdd(&sm.fill, wtot);
                        wb = wave_bcast(wb, 0);
                        uint32_t pos = wb + incl - c;
                        if (wb + wtot <= MCAP) {
                            const uint32_t base = pv ? sm.x.d.BF[jb] + pex2 - sm.x.d.PX[sm.x.d.PB[jb]] : 0u;
                            const uint16_t tg = (uint16_t)((bi >> 6) & 511u);
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                if ((tmask >> i) & 1u) {
                                    sm.u.s.V[pos] = base + val[i];
                                    if (batch) sm.u.s.TG[pos] = tg;
                                    pos++;
                                }
                            }
                        }       // else: the range holds more than LDS; the caller sees fill > MCAP and splits it
                    }
                }
            }
            g0 += nchunk;
        }
        __syncthreads();
    };

    // ---- sort the decoded postings of [lo, hi] (terms [t0, t0 + nt)) by buckets and write the survivors to `out` in
    // (term, id) order.  Returns false when the range has to be split (more postings than LDS, or clustered ids that
    // overflow a bucket) — nothing was written then.  alloc(n) is called once, by every thread, with the survivor count,
    // before anything is written, and returns where they go.  Batches also store per-term survivor counts.
    auto sort_range = [&](uint32_t lo, uint32_t hi, bool batch, uint32_t t0, uint32_t nt, auto alloc, uint32_t *n_out) -> bool {
        const uint32_t n = sm.fill;
        *n_out = 0;
        if (n > MCAP) return false;
        uint32_t *C32 = sm.u.s.C32;
        for (uint32_t i = (uint32_t)tid; i < MCAP / 2u + 4u; i += MT) C32[i] = 0u;
        if ((uint32_t)tid < MCAP / 32u + 1u) sm.x.f.DB[tid] = 0u;
        // bucket maps: term t of a batch owns floor(n_t * MCAP / n) buckets, a range tile all MCAP of them
        uint32_t u_mn = 0, u_nbm1 = 0;
        float u_scale = 0.0f;
        if (batch) {
            uint32_t nbk = 0, mn = 0, mx = 0;
            if ((uint32_t)tid < nt) {
                const uint32_t n_t = p.tn[t0 + (uint32_t)tid];
                mn = p.tmin[t0 + (uint32_t)tid];
                mx = p.tmax[t0 + (uint32_t)tid];
                nbk = n_t ? (uint32_t)(((uint64_t)n_t * MCAP) / n) : 0u;
            }
            uint32_t totb;
            const uint32_t tb = block_excl_scan(nbk, sm.wsum, &totb);
            if ((uint32_t)tid < nt) {
                sm.x.f.TB[tid] = (uint16_t)tb;
                sm.x.f.TT[tid] = make_uint2(mn, __float_as_uint(nbk ? (float)nbk / ((float)(mx - mn) + 1.0f) : 0.0f));
            }
            if (tid == 0) sm.x.f.TB[nt] = (uint16_t)totb;
        } else {
            const uint32_t t_mn = p.tmin[t0], t_mx = p.tmax[t0];
            u_mn = lo > t_mn ? lo : t_mn;
            const uint32_t mxr = hi < t_mx ? hi : t_mx;
            u_nbm1 = MCAP - 1u;
            u_scale = (float)MCAP / ((float)((mxr > u_mn ? mxr : u_mn) - u_mn) + 1.0f);
        }
        __syncthreads();
        // ---- bucket of every posting, slot inside the bucket
        unsigned long long slots = 0ull;
        bool over = false;
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) {
            const uint32_t i = (uint32_t)tid + j * MT;
            if (i < n) {
                const uint32_t v = sm.u.s.V[i];
                uint32_t mn = u_mn, nbm1 = u_nbm1, tb = 0;
                float sc = u_scale;
                if (batch) {
                    const uint32_t ti = sm.u.s.TG[i];
                    const uint2 te = sm.x.f.TT[ti];
                    tb = sm.x.f.TB[ti];
                    nbm1 = (uint32_t)sm.x.f.TB[ti + 1u] - tb - 1u;
                    mn = te.x;
                    sc = __uint_as_float(te.y);
                }
                const uint32_t bq = (uint32_t)((float)(v - mn) * sc);
                const uint32_t b = tb + (bq < nbm1 ? bq : nbm1);
                sm.u.s.TG[i] = (uint16_t)b;
                const uint32_t sh = 16u * (b & 1u);
                uint32_t slot = (atomicAdd(&C32[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
                if (slot > BKT_LIMIT) { over = true; slot = BKT_LIMIT; }
                slots |= (unsigned long long)slot << (4u * j);
            }
        }
        if (over) sm.ovf = 1u;
        __syncthreads();
        if (sm.ovf) return false;
        // ---- exclusive scan of the counters in place: EPT buckets = EPT / 2 words per thread
        {
            uint32_t c[EPT], sum = 0;
#pragma unroll
            for (uint32_t j = 0; j < EPT / 2u; j++) {
                const uint32_t w = C32[(EPT / 2u) * (uint32_t)tid + j];
                c[2u * j] = w & 0xFFFFu;
                c[2u * j + 1u] = w >> 16;
                sum += c[2u * j] + c[2u * j + 1u];
            }
            uint32_t tot_;
            uint32_t run = block_excl_scan(sum, sm.wsum, &tot_);
#pragma unroll
            for (uint32_t j = 0; j < EPT / 2u; j++) {
                const uint32_t e0 = run;
                run += c[2u * j];
                const uint32_t e1 = run;
                run += c[2u * j + 1u];
                C32[(EPT / 2u) * (uint32_t)tid + j] = e0 | (e1 << 16);
            }
            if (tid == (int)MT - 1) C32[MCAP / 2u] = run;        // base of the bucket past the last one
        }
        __syncthreads();
        auto base_of = [&](uint32_t b) -> uint32_t { return (C32[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu; };
        // ---- scatter into bucket order (in place: all reads, then all writes)
        {
            uint32_t v[EPT], pk[EPT];
#pragma unroll
            for (uint32_t j = 0; j < EPT; j++) {
                const uint32_t i = (uint32_t)tid + j * MT;
                v[j] = 0; pk[j] = 0;
                if (i < n) {
                    v[j] = sm.u.s.V[i];
                    const uint32_t b = sm.u.s.TG[i];
                    pk[j] = (base_of(b) + (uint32_t)((slots >> (4u * j)) & 15ull)) | (b << 16);
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < EPT; j++) {
                const uint32_t i = (uint32_t)tid + j * MT;
                if (i < n) { sm.u.s.V[pk[j] & 0xFFFFu] = v[j]; sm.u.s.TG[pk[j] & 0xFFFFu] = (uint16_t)(pk[j] >> 16); }
            }
        }
        __syncthreads();
        // ---- every posting ranks itself inside its bucket: final position, duplicate / tombstone flag
        uint32_t fv[EPT], fp[EPT];
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) {
            const uint32_t q = (uint32_t)tid + j * MT;
            fv[j] = 0; fp[j] = 0x80000000u;
            if (q < n) {
                const uint32_t v = sm.u.s.V[q];
                const uint32_t b = sm.u.s.TG[q];
                const uint32_t blo = base_of(b), bhi = base_of(b + 1u);
                const uint32_t ts = (p.tomb && (v >> 5) < p.tomb_nwords) ? p.tomb_summary[v >> 9] : 0u;   // in flight during the ranking
                uint32_t r = 0, dup = 0;
                {   // the first four of the bucket without a loop (buckets hold one posting on average: lanes that loop make the
                    // whole wave wait for the fullest bucket among its 64); reading past the bucket is harmless (masked)
                    const uint32_t *B4 = &sm.u.s.V[blo];
                    const uint32_t u0 = B4[0], u1 = B4[1], u2 = B4[2], u3 = B4[3];
                    const uint32_t nb = bhi - blo;
                    const uint32_t e0 = (u0 == v && blo < q) ? 1u : 0u;
                    const uint32_t e1 = (nb > 1u && u1 == v && blo + 1u < q) ? 1u : 0u;
                    const uint32_t e2 = (nb > 2u && u2 == v && blo + 2u < q) ? 1u : 0u;
                    const uint32_t e3 = (nb > 3u && u3 == v && blo + 3u < q) ? 1u : 0u;
                    r = (u0 < v ? 1u : 0u) + ((nb > 1u && u1 < v) ? 1u : 0u) + ((nb > 2u && u2 < v) ? 1u : 0u) + ((nb > 3u && u3 < v) ? 1u : 0u) +
                        e0 + e1 + e2 + e3;
                    dup = e0 | e1 | e2 | e3;
                }
                for (uint32_t m = blo + 4u; m < bhi; m++) {
                    const uint32_t u = sm.u.s.V[m];
                    const uint32_t eq = (u == v && m < q) ? 1u : 0u;
                    r += (u < v ? 1u : 0u) + eq;
                    dup |= eq;
                }
                uint32_t dead = dup;
                if ((ts >> ((v >> 4) & 31u)) & 1u) dead |= (p.tomb[v >> 5] >> (v & 31u)) & 1u;     // rarely: the bitmap itself
                const uint32_t P = blo + r;
                if (dead) atomicOr(&sm.x.f.DB[P >> 5], 1u << (P & 31u));
                fv[j] = v;
                fp[j] = P | (dead << 31);
            }
        }
        __syncthreads();
        // ---- dead ids before every 32 sorted positions
        const uint32_t nw = (n + 31u) >> 5;
        uint32_t totdead;
        {
            const uint32_t x = (uint32_t)tid < nw ? (uint32_t)__popc(sm.x.f.DB[tid]) : 0u;
            const uint32_t ex = block_excl_scan(x, sm.wsum, &totdead);
            if ((uint32_t)tid <= nw) sm.x.f.DP[tid] = ex;       // (DP[nw] = all of them)
        }
        const uint32_t nout = n - totdead;
        uint32_t *out = alloc(nout);                            // (barriers inside)
        __syncthreads();
        auto dead_before = [&](uint32_t x) -> uint32_t {
            return sm.x.f.DP[x >> 5] + (uint32_t)__popc(sm.x.f.DB[x >> 5] & ((1u << (x & 31u)) - 1u));
        };
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) {
            if (!(fp[j] >> 31)) out[fp[j] - dead_before(fp[j])] = fv[j];       // lanes hold neighbouring ranks: coalesced
        }
        if (batch && (uint32_t)tid < nt) {
            // term t sits at the sorted positions [base of its first bucket, base of the next term's first bucket)
            const uint32_t sp0 = base_of(sm.x.f.TB[tid]), sp1 = base_of(sm.x.f.TB[tid + 1]);
            p.out_counts[t0 + (uint32_t)tid] = (sp1 - sp0) - (dead_before(sp1) - dead_before(sp0));
        }
        *n_out = nout;
        return true;
    };

    // ---- a doc range of at most BMW * 32 docs (from lo & ~31) through the LDS bitmap: exact for any input
    auto bitmap_range = [&](uint32_t lo, uint32_t hi, uint32_t t0, auto alloc) -> uint32_t {
        const uint32_t lo32 = lo & ~31u;
        const uint32_t nw = ((hi - lo32) >> 5) + 1u;                      // <= BMW
        uint32_t *bm = sm.u.bm;
        for (uint32_t i = 4u * (uint32_t)tid; i < nw; i += 4u * MT) *reinterpret_cast<uint4 *>(&bm[i]) = make_uint4(0, 0, 0, 0);
        decode(std::true_type{}, lo, hi, lo32, false, t0, 1u, true);    // (starts with a barrier)
        II2_STAMP(1)      // decode + mark
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
        // consecutive words per thread, as few as cover the range
        const uint32_t wpt = (nw + MT - 1u) / MT;
        const uint32_t w0 = wpt * (uint32_t)tid;
        const uint32_t w1 = w0 + wpt < nw ? w0 + wpt : nw;
        uint32_t cnt = 0;
#pragma unroll 1
        for (uint32_t w = w0; w < w1; w++) cnt += (uint32_t)__popc(bm[w]);
        uint32_t tot;
        uint32_t pos = block_excl_scan(cnt, sm.wsum, &tot);
        uint32_t *out = alloc(tot);
#pragma unroll 1
        for (uint32_t w = w0; w < w1; w++) {
            uint32_t x = bm[w];
            const uint32_t base = lo32 + 32u * w;
            while (x) {
                out[pos++] = base + (uint32_t)__ffs((int)x) - 1u;
                x &= x - 1u;
            }
        }
        II2_STAMP(2)      // tombstones, count, extract
        return tot;
    };

    const uint32_t n_tiles = *p.n_tiles_dev;       // computed by the plan kernels; the host only knows an upper bound
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint4 td = p.desc[tile];
        const uint32_t t0 = td.x, t1 = td.y & 0x3FFFFFFFu;
        const uint32_t nt = t1 - t0;
        const bool large = (td.y & MERGE_DESC_LARGE) != 0u, bm_tile = (td.y & MERGE_DESC_BITMAP) != 0u;
        const uint32_t dlo = td.z, dhi = td.w;
        const bool whole = dlo == 0u && dhi == 0xFFFFFFFFu;
        unsigned long long slot = p.npre[t0];
        uint32_t total = 0;

        // One term's doc range [lo, hi], bisected until every piece fits: leaves in doc order, appended at dst + *acc.
        // from_root: the runs of [lo, hi] are already set up.
        auto single_term = [&](uint32_t t, uint32_t lo, uint32_t hi, bool from_root, uint32_t *dst, uint32_t *acc) {
            uint32_t sp = 1;
            bool first = from_root;
            __syncthreads();
            if (tid == 0) { sm.stk[0][0] = lo; sm.stk[0][1] = hi; }
            while (sp > 0) {
                __syncthreads();
                const uint32_t a = sm.stk[sp - 1][0], b = sm.stk[sp - 1][1];
                sp--;
                if (!first) setup_runs(2u, t, t + 1u, 0u, a, b);
                first = false;
                auto alloc = [&](uint32_t) -> uint32_t * { return dst + *acc; };
                if (sm.RB[k] == 0u) continue;
                if (b - (a & ~31u) < BMW * 32u) { *acc += bitmap_range(a, b, t, alloc); continue; }
                decode(std::false_type{}, a, b, 0u, false, t, 1u, true);
                uint32_t c = 0;
                if (sort_range(a, b, false, t, 1u, alloc, &c)) { *acc += c; continue; }
                const uint32_t mid = a + ((b - a) >> 1);     // a < b here: a range of one doc fits the bitmap
                __syncthreads();
                if (tid == 0) {
                    sm.stk[sp][0] = mid + 1u; sm.stk[sp][1] = b;
                    sm.stk[sp + 1][0] = a;    sm.stk[sp + 1][1] = mid;
                }
                sp += 2;
            }
        };

        if (!large) {
            // ---- batch of small terms: whole lists; survivors go to the batch's own region of the parking array
            setup_runs(0u, t0, t1, tile, 0u, 0xFFFFFFFFu);
            II2_STAMP(0)
            decode(std::false_type{}, 0u, 0xFFFFFFFFu, 0u, true, t0, nt, false);
            II2_STAMP(1)
            uint32_t *dst = p.tmp + slot;
            auto alloc = [&](uint32_t) -> uint32_t * { return dst; };
            if (!sort_range(0u, 0xFFFFFFFFu, true, t0, nt, alloc, &total)) {
                // clustered ids: term by term, each through the bisection
                total = 0;
                for (uint32_t t = t0; t < t1; t++) {
                    const uint32_t before = total;
                    setup_runs(0u, t, t + 1u, tile, 0u, 0xFFFFFFFFu);
                    single_term(t, 0u, 0xFFFFFFFFu, true, dst, &total);
                    __syncthreads();
                    if (tid == 0) p.out_counts[t] = total - before;
                }
            }
            II2_STAMP(3)
        } else if (dlo <= dhi) {
            // ---- a doc range of a large term; its place inside the term's region comes from the term's bump allocator
            // (range tiles finish in any order; the packing pass only needs every tile's slot and count)
            setup_runs(whole ? 0u : 1u, t0, t1, tile, dlo, dhi);
            II2_STAMP(0)
            auto alloc = [&](uint32_t c) -> uint32_t * {
                __syncthreads();
                if (tid == 0) sm.ab = c ? atomicAdd(&p.term_alloc[t0], c) : 0u;
                __syncthreads();
                slot = p.npre[t0] + sm.ab;
                return p.tmp + slot;
            };
            if (sm.RB[k] == 0u) {
                total = 0;
            } else if (bm_tile) {
                total = bitmap_range(dlo, dhi, t0, alloc);
            } else {
                decode(std::false_type{}, dlo, dhi, 0u, false, t0, 1u, !whole);
                II2_STAMP(1)
                if (!sort_range(dlo, dhi, false, t0, 1u, alloc, &total)) {
                    // more postings than LDS or clustered ids: reserve room for all the range's postings, then bisect
                    uint32_t *dst = alloc(sm.fill);
                    total = 0;
                    single_term(t0, dlo, dhi, false, dst, &total);
                }
                II2_STAMP(3)
            }
        }
        __syncthreads();
        if (tid == 0) { p.tile_count[tile] = total; p.tile_slot[tile] = slot; }
        II2_STAMP(6)
    }
    if (stamps && tid == 0)
        for (int i = 0; i < 8; i++) p.debug[(uint64_t)blockIdx.x * 8u + i] = tacc[i];
#undef II2_STAMP
}

// packs the parked survivors: tile t's ids go to out[off[t] ...].  Nothing is written when the result does not fit the
// caller's buffer (the call then fails with II2_ECAPACITY: all-or-nothing).
__global__ __launch_bounds__(256) void k_merge_pack(const uint32_t *__restrict__ tmp, const unsigned long long *__restrict__ slot,
                                                    const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ off, const uint32_t *__restrict__ n_tiles_dev,
                                                    uint32_t *__restrict__ out, uint64_t out_cap, uint64_t *__restrict__ d_total) {
    const uint32_t n_tiles = *n_tiles_dev;
    const uint64_t total = off[n_tiles];
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_total = total;
    if (total > out_cap) return;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t c = cnt[tile];
        const uint64_t ob = off[tile];
        const uint32_t *src = tmp + slot[tile];
        for (uint32_t q = threadIdx.x; q < c; q += 256u) out[ob + q] = src[q];
    }
}

// survivors of every large term = survivors of its tiles (tile_off = exclusive scan of the tile counts)
__global__ void k_merge_large_counts(MergeParams p, const uint64_t *__restrict__ tile_off) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_terms || p.ntl[t] == 0u) return;
    const uint32_t a = p.term_tile[t];
    p.out_counts[t] = (uint32_t)(tile_off[a + p.ntl[t]] - tile_off[a]);
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

hipError_t launch_merge_plan_terms(const MergeSegs *ms, const MergeParams &p, hipStream_t s) {
    hipLaunchKernelGGL(k_mp_terms, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, ms, p);
    return hipGetLastError();
}
hipError_t launch_merge_heads(const MergeParams &p, const uint64_t *wpre, uint32_t *head, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_heads, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, p, wpre, head);
    return hipGetLastError();
}
hipError_t launch_merge_term_tile(const MergeParams &p, const uint32_t *head, const uint32_t *hpre, const uint32_t *lpre,
                                  uint32_t *term_tile, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_term_tile, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, p, head, hpre, lpre, term_tile);
    return hipGetLastError();
}
hipError_t launch_merge_tile_desc(const MergeSegs *ms, const MergeParams &p, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_tile_desc, dim3(grid_for(p.n_tiles_ub)), dim3(256), 0, s, ms, p);
    return hipGetLastError();
}
hipError_t launch_merge_tile_runs(const MergeSegs *ms, const MergeParams &p, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_tile_runs, dim3(grid_for((uint64_t)p.n_tiles_ub * p.k)), dim3(256), 0, s, ms, p);
    return hipGetLastError();
}
hipError_t launch_merge_tiles(const MergeSegs *ms, const MergeParams &p, uint32_t grid, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_tiles, dim3(grid < p.n_tiles_ub ? grid : p.n_tiles_ub), dim3(MERGE_THREADS), 0, s, ms, p);
    return hipGetLastError();
}
hipError_t launch_merge_large_counts(const MergeParams &p, const uint64_t *tile_off, hipStream_t s) {
    if (p.n_terms == 0 || p.n_tiles_ub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_large_counts, dim3(grid_for(p.n_terms)), dim3(256), 0, s, p, tile_off);
    return hipGetLastError();
}
hipError_t launch_merge_pack(const MergeParams &p, const uint64_t *tile_off, hipStream_t s) {
    if (p.n_tiles_ub == 0) return hipSuccess;
    const uint32_t g = p.n_tiles_ub < 16384u ? p.n_tiles_ub : 16384u;
    hipLaunchKernelGGL(k_merge_pack, dim3(g), dim3(256), 0, s, (const uint32_t *)p.tmp, (const unsigned long long *)p.tile_slot,
                       (const uint32_t *)p.tile_count, tile_off, p.n_tiles_dev, p.out_values, p.out_cap, p.d_total);
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
