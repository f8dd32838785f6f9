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
// A range tile decodes only its own part of every block: the plan cuts each list at the tiles' doc bounds (cut_for /
// k_merge_tile_runs*: the block that straddles a bound is walked once, there).
// Tiles are independent: each parks its survivors in a scratch array (batches at the input rank of their first term,
// tiles of a large term through a bump allocator inside the term's region).  When the caller's buffer is known to hold
// any result the tiles then move them to their final place themselves (ticket order, a scanner workgroup turns the
// published counts into offsets: "direct placement" below); otherwise a scan of the tile counts and a packing pass
// follow.  Either way the result is the CSR the reference's writer would have been fed: terms ascending, ids ascending.
#include <type_traits>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t MCAP = MERGE_CAP;
constexpr uint32_t MT = MERGE_THREADS;
constexpr uint32_t MW = MT / 64u;                 // waves per workgroup
constexpr uint32_t EPT = MCAP / MT;               // elements per thread in the sort passes (14)
constexpr uint32_t PCAP = 768;                   // 16-byte payload pieces decoded per chunk of blocks
constexpr uint32_t BKT_LIMIT = 15;                // fullest bucket the bucket sort accepts (slot numbers are 4 bits)
constexpr uint32_t BMW = MERGE_BM_WORDS;
constexpr uint32_t MERGE_PQ = 16;                 // parked tiles a workgroup may have waiting for their output offset
constexpr uint32_t SPIN_LIMIT = 4000000;          // bounded waits (each ~1.5 us: seconds in all): a bug must not hang the GPU
static_assert(MCAP % MT == 0 && EPT * 4u <= 64u && (MCAP / 2u) % MT == 0, "sort passes: EPT elements and EPT / 2 counter words per thread");

// payload bytes live in global memory: say so (a pointer that went through LDS would be loaded with flat instructions)
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ uint4 gload16(const uint8_t *p) {
    const u32x4_unaligned v = *(const __attribute__((address_space(1))) u32x4_unaligned *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t gload4(const uint8_t *p) { return *(const __attribute__((address_space(1))) u32_unaligned *)p; }

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
            else {
                // cuts at block boundaries of the longest list give the fullest tile ceil(nb / tiles) of its nb blocks: with a
                // handful of blocks per tile (few segments: a tile is several blocks of every list) that is well over the
                // average - 7 blocks into 2 tiles are 3 + 4, the second tile overflows and is bisected, decoding its blocks
                // again and again (one such tile was 185 of a 250 us merge).  Up to two tiles more bring the fullest one under
                // the target (the plan's upper bound of the tile count has room for them).
                for (uint32_t extra = 0; extra < 2u; extra++) {
                    const uint32_t most = (best_nb + tiles - 1u) / tiles;
                    if ((uint64_t)n32 * most <= (uint64_t)p.range_target * best_nb) break;
                    tiles++;
                }
            }
        }
    } else {
        w = n32 > p.wmin ? n32 : p.wmin;
    }
    p.tinfo[t] = info;
    p.weight[t] = w;
    p.ntl[t] = tiles;
}

// the same for FEW terms: 16 lanes per term share its k lists (a thread that walks 64 segments alone is a chain of 64 x 5
// dependent loads: 110 us for a 3000-term chunk of C4; with many terms the thread-per-term kernel above reads every
// segment's arrays coalesced and is 3.5x faster than this one)
__global__ __launch_bounds__(256) void k_mp_terms_few(const MergeSegs *__restrict__ ms, MergeParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t t = i >> 4;
    const uint32_t sub = (uint32_t)i & 15u;
    const bool live = t < p.n_terms;                    // (lanes past the last term take part in the shuffles with nothing)
    uint64_t n = 0;
    uint32_t mn = 0xFFFFFFFFu, mx = 0u, best = 0, best_nb = 0;
    if (live) {
        for (uint32_t s = sub; s < p.k; s += 16u) {
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
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const uint64_t n2 = (uint64_t)__shfl_xor((long long)n, d, 16);
        const uint32_t mn2 = (uint32_t)__shfl_xor((int)mn, d, 16), mx2 = (uint32_t)__shfl_xor((int)mx, d, 16);
        const uint32_t bn2 = (uint32_t)__shfl_xor((int)best_nb, d, 16), b2 = (uint32_t)__shfl_xor((int)best, d, 16);
        n += n2;
        mn = mn2 < mn ? mn2 : mn;
        mx = mx2 > mx ? mx2 : mx;
        if (bn2 > best_nb || (bn2 == best_nb && b2 < best)) { best_nb = bn2; best = b2; }      // (the first of the longest lists, as a serial walk finds it)
    }
    if (sub != 0u || t > p.n_terms) return;
    if (t == p.n_terms) { p.tn[t] = 0; p.tmin[t] = 0; p.tmax[t] = 0; p.tinfo[t] = 0; p.weight[t] = 0; p.ntl[t] = 0; return; }
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
            else {
                // cuts at block boundaries of the longest list give the fullest tile ceil(nb / tiles) of its nb blocks: with a
                // handful of blocks per tile (few segments: a tile is several blocks of every list) that is well over the
                // average - 7 blocks into 2 tiles are 3 + 4, the second tile overflows and is bisected, decoding its blocks
                // again and again (one such tile was 185 of a 250 us merge).  Up to two tiles more bring the fullest one under
                // the target (the plan's upper bound of the tile count has room for them).
                for (uint32_t extra = 0; extra < 2u; extra++) {
                    const uint32_t most = (best_nb + tiles - 1u) / tiles;
                    if ((uint64_t)n32 * most <= (uint64_t)p.range_target * best_nb) break;
                    tiles++;
                }
            }
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

// Where a doc range begins inside a list: the CUT before the first posting >= x of the list that owns the segment's blocks
// [b_lo, b_hi).  {block | CUT_INSIDE, payload byte, doc before}: without CUT_INSIDE the cut is in front of that block (or at the
// list's end, block == b_hi); with it the cut is inside the block, in front of the gap at `payload byte` (absolute offset in
// the segment's payload), and `doc before` is the posting that gap is added to.  Found in the skip table and then by walking
// the one block that straddles x - once, here, so that the tiles on both sides of x decode only their own part of it (a
// range tile holds less than one block per list: decoding the straddling blocks whole was more than half of its decode work).
constexpr uint32_t CUT_INSIDE = 1u << 31;

// lane 15's x of my group of 16 lanes (grp = lane / 16): four scalar reads and a select instead of a trip through the LDS crossbar
__device__ __forceinline__ uint32_t row_last(uint32_t x, uint32_t grp) {
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return grp == 0u ? a : grp == 1u ? b : grp == 2u ? c : d;
}

// cut0[s * n_tiles_ub + tile] / cut1[...]: where list s's part of the tile begins and ends (tiles that take whole lists: the
// lists' first block and the block after their last).  A tile's end is the next tile's beginning, so the cut at a tile's lower
// bound is computed once and handed to the tile(s) before it as their end.  A thread per (list, tile) - neighbouring lanes hold
// neighbouring tiles of ONE list - finds the block that straddles its bound in the skip table.  The walks through those blocks
// are then done by the wave together: 16 lanes per block, 16 payload bytes per lane, 256 bytes of a block in one round of
// loads, four blocks at a time, and a block is walked ONCE for all the bounds that fall into it (with 64 lists a tile takes a
// fifth of a block from each: five bounds per block).  A thread walking its block alone is a chain of dependent loads
// (~1 ms on C3 against 0.1 ms for everything else the plan does); one shared walk per bound was 3.3 ms of C4's 14.7.
__global__ __launch_bounds__(256) void k_merge_tile_runs_shared(const MergeSegs *__restrict__ ms, MergeParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = lane_id();
    const uint32_t hl = (uint32_t)l & 15u, grp = (uint32_t)l >> 4;
    const uint32_t n_tiles = *p.n_tiles_dev;
    const uint64_t n_pairs = (uint64_t)n_tiles * p.k;
    const bool live = i < n_pairs;                              // (no early exit of single lanes: the wave works together below)
    if (__ballot(live) == 0ull) return;                         // (the grid is sized by the tiles' upper bound: most waves have nothing)
    const uint32_t s = live ? (uint32_t)(i / n_tiles) : 0u, tile = live ? (uint32_t)(i % n_tiles) : 0u;
    const uint4 td = live ? p.desc[tile] : make_uint4(0u, 0u, 1u, 0u);
    const bool nonempty = live && td.z <= td.w;                 // (an empty range: the tile decodes nothing)
    const SegView &sv = ms->segs[s];
    const uint32_t b_lo = live ? sv.blk_off[td.x] : 0u, b_hi = live ? sv.blk_off[td.y & 0x3FFFFFFFu] : 0u;
    const uint32_t x = td.z;
    uint4 c0 = make_uint4(b_lo, 0u, 0u, 0u);
    // ---- the block that straddles x (if any): the last one that starts below x
    bool walk = false;
    uint32_t a = b_lo, start = 0, len = 0, first = 0;
    if (nonempty && x != 0u && b_hi > b_lo && !(p.pad0 & 32u)) {
        const ii2_skip *__restrict__ skip = sv.skip;
        const uint32_t f0 = skip[b_lo].first_doc;
        if (x > f0) {
            auto get = [&](uint32_t j) { return skip[j].first_doc; };
            a = upper_bound_guess(get, b_lo, b_hi, x - 1u, f0, skip[b_hi - 1u].first_doc);      // first block that starts at or after x (> b_lo)
            const ii2_skip e0 = skip[a - 1u];
            start = e0.byte_off;
            len = skip[a].byte_off - start;
            if (len > 1280u) len = 1280u;                       // (a block holds <= 256 postings of <= 5 bytes; imported segments are validated)
            first = e0.first_doc;
            c0 = make_uint4(a, 0u, 0u, 0u);                      // unless the walk finds x inside the block
            walk = len != 0u && !(p.pad0 & 16u);
        }
    }
    // (block a of segment s: two lanes share a walk when both agree; lanes of a wave may belong to two lists)
    // The walk, shared: a group of 16 lanes decodes 256 bytes of a block into LDS - per byte the doc-id sum up to and including it
    // (non-decreasing; at a varint's last byte: the posting's doc id) and the last-byte marks - and EVERY lane whose bound falls
    // into that block then finds its crossing by itself, all bounds at once: bisection over the 256 sums for the first byte at
    // or above its bound (that byte lies in the first posting >= the bound), the last mark before it is where that posting
    // starts and holds the doc id before it.  (The first version took the bounds one after the other, each with a 16-way
    // compare of every lane's sums and a dozen shuffles: ~900 instructions per block and step against ~300.)
    __shared__ __align__(16) uint32_t sh_doc[4][4][256];     // [wave][group][byte]
    __shared__ uint32_t sh_end[4][4][16];        // [wave][group][piece] bit q: byte q of the piece ends a posting
    __shared__ __align__(16) uint32_t sh_tot[4][4][16];      // [wave][group][piece] the sum at the piece's last byte
    __shared__ uint2 sh_last[4][4];              // [wave][group] {payload offset behind the step's last posting end, its doc id}
    const uint32_t wv = threadIdx.x >> 6;
    const unsigned long long blk_id = ((unsigned long long)s << 32) | a;
    for (unsigned long long need = __ballot(walk); need != 0ull;) {
        // up to four different blocks, one per group of 16 lanes, each with the lanes whose bounds fall into it (they are
        // neighbours: ascending tiles, ascending bounds)
        unsigned long long bm[4] = {0ull, 0ull, 0ull, 0ull};
        int lead[4] = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (need) {
                lead[g] = __ffsll((long long)need) - 1;
                const unsigned long long id = (unsigned long long)__shfl((long long)blk_id, lead[g], 64);
                bm[g] = __ballot(walk && blk_id == id) & need;
                need &= ~bm[g];
            } else (void)__shfl((long long)blk_id, 0, 64);       // (every lane takes part in every shuffle)
        }
        // my role as a member of group grp: its block
        const int src = grp == 0u ? lead[0] : grp == 1u ? lead[1] : grp == 2u ? lead[2] : lead[3];
        const unsigned long long gmask = grp == 0u ? bm[0] : grp == 1u ? bm[1] : grp == 2u ? bm[2] : bm[3];
        const uint32_t w_start = (uint32_t)__shfl((int)start, src, 64), w_len = (uint32_t)__shfl((int)len, src, 64);
        const uint32_t w_first = (uint32_t)__shfl((int)first, src, 64);
        const uint8_t *w_pay = ms->segs[(uint32_t)__shfl((int)s, src, 64)].payload;
        const uint32_t np = gmask ? (w_len + 15u) >> 4 : 0u;
        // my role as the owner of a bound: which group walks my block (if any, this round)
        const bool b0 = (bm[0] >> l) & 1ull, b1 = (bm[1] >> l) & 1ull, b2 = (bm[2] >> l) & 1ull, b3 = (bm[3] >> l) & 1ull;
        bool asking = b0 || b1 || b2 || b3;
        const uint32_t og = b0 ? 0u : b1 ? 1u : b2 ? 2u : 3u;
        uint32_t prev_start = start, prev_doc = first;          // the posting end before my step: the block's beginning so far
        uint32_t docbase = w_first;                             // sum of all gap bits before the step's first piece
        for (uint32_t pb = 0;; pb += 16u) {
            const unsigned long long askmask = __ballot(asking);
            if (__ballot((askmask & gmask) != 0ull && pb < np) == 0ull) break;       // (wave-uniform: a group that is done idles)
            const uint32_t pc = pb + hl;
            const bool pv = (askmask & gmask) != 0ull && pc < np;
            uint32_t val[16], tmask = 0, w[4] = {0, 0, 0, 0}, prev = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) val[q] = 0;
            const uint32_t off = 16u * pc;
            if (pv) {
                const uint32_t rem = w_len - off;
                const uint8_t *pp = w_pay + w_start + off;
                const uint4 w4 = gload16(pp);                    // (segments carry 16 bytes of padding)
                prev = off ? gload4(pp - 4) : 0u;
                w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                if (rem < 16u) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t nj = rem > 4u * j ? (rem - 4u * j < 4u ? rem - 4u * j : 4u) : 0u;
                        w[j] &= nj >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nj)) - 1u);
                    }
                }
                uint32_t sh = 7u * ((uint32_t)__clz((int)~(prev | 0x7F7F7F7Fu)) >> 3);      // continuation bytes pending before my first byte
                uint32_t sum = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t c7 = __builtin_amdgcn_ubfe(w[q >> 2], 8 * (q & 3), 7);
                    const uint32_t cm = (uint32_t)__builtin_amdgcn_sbfe((int)w[q >> 2], 8 * (q & 3) + 7, 1);
                    sum += c7 << (sh & 31u);
                    sh = (sh + 7u) & cm;
                    tmask |= ~cm & (1u << q);
                    val[q] = sum;
                }
                if (rem < 16u) tmask &= (1u << rem) - 1u;
            }
            uint32_t incl = val[15];                              // prefix sum inside the group of 16 lanes = a DPP row
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
            const uint32_t base = docbase + incl - val[15];
            // ---- the step's 256 bytes in LDS (a piece past the block's end: all ones - never below a bound, and past `valid`)
            {
                uint4 *d4 = reinterpret_cast<uint4 *>(&sh_doc[wv][grp][16u * hl]);
                if (pv) {
                    d4[0] = make_uint4(base + val[0], base + val[1], base + val[2], base + val[3]);
                    d4[1] = make_uint4(base + val[4], base + val[5], base + val[6], base + val[7]);
                    d4[2] = make_uint4(base + val[8], base + val[9], base + val[10], base + val[11]);
                    d4[3] = make_uint4(base + val[12], base + val[13], base + val[14], base + val[15]);
                } else {
                    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                    d4[0] = ones; d4[1] = ones; d4[2] = ones; d4[3] = ones;
                }
                sh_end[wv][grp][hl] = tmask;
                sh_tot[wv][grp][hl] = pv ? base + val[15] : 0xFFFFFFFFu;
                // the step's last posting end, for the bounds that go on to the next step: the highest lane of the group with a mark
                const uint32_t marks = (uint32_t)(__ballot(tmask != 0u) >> (16u * grp)) & 0xFFFFu;
                if (marks != 0u && hl == 31u - (uint32_t)__clz((int)marks)) {
                    const uint32_t q = 31u - (uint32_t)__clz((int)tmask);
                    sh_last[wv][grp] = make_uint2(w_start + off + q + 1u, sh_doc[wv][grp][16u * hl + q]);
                }
            }
            // ---- every lane whose bound falls into one of the four blocks: its crossing, if it lies in this step
            if (asking) {
                const uint32_t done_bytes = 16u * pb;
                const uint32_t valid = len - done_bytes < 256u ? len - done_bytes : 256u;       // (my block's bytes in this step; len > done_bytes while I ask)
                const uint32_t *dv = sh_doc[wv][og];
                // first byte in [0, valid) whose sum is >= x: the sums ascend, so it is the number of sums below x - counted over
                // the 16 pieces' last sums, then over that piece's 16 bytes (two LDS round trips; a bisection is nine)
                uint32_t pi = 0, qi = 0;
                {
                    const uint4 *t4 = reinterpret_cast<const uint4 *>(sh_tot[wv][og]);
                    const uint4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
                    pi = (t0.x < x) + (t0.y < x) + (t0.z < x) + (t0.w < x) + (t1.x < x) + (t1.y < x) + (t1.z < x) + (t1.w < x) +
                         (t2.x < x) + (t2.y < x) + (t2.z < x) + (t2.w < x) + (t3.x < x) + (t3.y < x) + (t3.z < x) + (t3.w < x);
                    const uint4 *d4 = reinterpret_cast<const uint4 *>(dv + 16u * (pi & 15u));
                    const uint4 d0 = d4[0], d1 = d4[1], d2 = d4[2], d3 = d4[3];
                    qi = (d0.x < x) + (d0.y < x) + (d0.z < x) + (d0.w < x) + (d1.x < x) + (d1.y < x) + (d1.z < x) + (d1.w < x) +
                         (d2.x < x) + (d2.y < x) + (d2.z < x) + (d2.w < x) + (d3.x < x) + (d3.y < x) + (d3.z < x) + (d3.w < x);
                }
                const uint32_t lo = pi < 16u ? 16u * pi + qi : 256u;      // (pi < 16: the piece's last sum is >= x, so qi < 16)
                if (lo < valid) {
                    // the posting that holds byte lo is the first one >= x; it starts behind the last posting end before lo
                    uint32_t m = sh_end[wv][og][pi] & ((1u << qi) - 1u);
                    uint32_t tp = 16u * pi;
                    if (m == 0u && pi != 0u) { m = sh_end[wv][og][pi - 1u]; tp -= 16u; }
                    if (m != 0u) {
                        tp += 31u - (uint32_t)__clz((int)m);
                        c0 = make_uint4((a - 1u) | CUT_INSIDE, start + done_bytes + tp + 1u, dv[tp], 0u);
                    } else {
                        c0 = make_uint4((a - 1u) | CUT_INSIDE, prev_start, prev_doc, 0u);
                    }
                    asking = false;
                } else if (done_bytes + 256u >= len) {
                    asking = false;                              // the block lies wholly below my bound: the cut stays in front of the next block
                } else {
                    const uint2 lt = sh_last[wv][og];
                    prev_start = lt.x; prev_doc = lt.y;
                }
            }
            docbase += row_last(incl, grp);
        }
        // (bounds still pending: the block lies wholly below them - their cut stays in front of the next block)
    }
    if (!nonempty) return;
    if (x != 0u) {
        // the tiles of the same term right before this one: the empty ones in between and the one that ends where I begin
        for (uint32_t j = tile; j-- > 0u;) {
            const uint4 pd = p.desc[j];
            if (pd.x != td.x || !(pd.y & MERGE_DESC_LARGE)) break;
            if (pd.z > pd.w) continue;
            p.cut1[(uint64_t)s * p.cut_ss + (uint64_t)j * p.cut_st] = make_uint2(c0.x, c0.y);
            break;
        }
    }
    p.cut0[(uint64_t)s * p.cut_ss + (uint64_t)tile * p.cut_st] = c0;
    // the last tile of its term(s) ends where the lists end (a bitmap term's last window stops short of 2^32, and nothing follows it)
    bool last = tile + 1u == *p.n_tiles_dev;
    if (!last) { const uint4 nd = p.desc[tile + 1u]; last = nd.x != td.x || !(nd.y & MERGE_DESC_LARGE) || !(td.y & MERGE_DESC_LARGE); }
    if (last) p.cut1[(uint64_t)s * p.cut_ss + (uint64_t)tile * p.cut_st] = make_uint2(b_hi, 0u);
}


// Few lists (k < 32): a bound rarely shares its block with another one - a thread per (tile, list), neighbouring lanes hold the
// lists of one tile (cut0[tile * k + s]), every bound walks its block.
// cut0 / cut1: where list s's part of the tile begins and ends (tiles that take whole lists: the
// lists' first block and the block after their last).  A tile's end is the next tile's beginning, so the cut at a tile's lower
// bound is computed once and handed to the tile(s) before it as their end.  A thread per (tile, list) finds the block that
// straddles the bound in the skip table; the walks through those blocks are then done by the wave together, four at a time:
// 16 lanes per block, 16 payload bytes per lane, 256 bytes of a block in one round of loads (a thread walking its block
// alone is a chain of dependent loads: ~1 ms on C3 against 0.1 ms for everything else the plan does).
__global__ __launch_bounds__(256) void k_merge_tile_runs_few(const MergeSegs *__restrict__ ms, MergeParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = lane_id();
    const uint32_t hl = (uint32_t)l & 15u, grp = (uint32_t)l >> 4;
    const uint64_t n_pairs = (uint64_t)*p.n_tiles_dev * p.k;
    const bool live = i < n_pairs;                              // (no early exit of single lanes: the wave works together below)
    if (__ballot(live) == 0ull) return;                         // (the grid is sized by the tiles' upper bound: most waves have nothing)
    const uint32_t tile = live ? (uint32_t)(i / p.k) : 0u, s = live ? (uint32_t)(i % p.k) : 0u;
    const uint4 td = live ? p.desc[tile] : make_uint4(0u, 0u, 1u, 0u);
    const bool nonempty = live && td.z <= td.w;                 // (an empty range: the tile decodes nothing)
    const SegView &sv = ms->segs[s];
    const uint32_t b_lo = live ? sv.blk_off[td.x] : 0u, b_hi = live ? sv.blk_off[td.y & 0x3FFFFFFFu] : 0u;
    const uint32_t x = td.z;
    uint4 c0 = make_uint4(b_lo, 0u, 0u, 0u);
    // ---- the block that straddles x (if any): the last one that starts below x
    bool walk = false;
    uint32_t a = b_lo, start = 0, len = 0, first = 0;
    if (nonempty && x != 0u && b_hi > b_lo && !(p.pad0 & 32u)) {
        const ii2_skip *__restrict__ skip = sv.skip;
        const uint32_t f0 = skip[b_lo].first_doc;
        if (x > f0) {
            auto get = [&](uint32_t j) { return skip[j].first_doc; };
            a = upper_bound_guess(get, b_lo, b_hi, x - 1u, f0, skip[b_hi - 1u].first_doc);      // first block that starts at or after x (> b_lo)
            const ii2_skip e0 = skip[a - 1u];
            start = e0.byte_off;
            len = skip[a].byte_off - start;
            if (len > 1280u) len = 1280u;                       // (a block holds <= 256 postings of <= 5 bytes; imported segments are validated)
            first = e0.first_doc;
            c0 = make_uint4(a, 0u, 0u, 0u);                      // unless the walk finds x inside the block
            walk = len != 0u && !(p.pad0 & 16u);
        }
    }
    const unsigned long long pay64 = (unsigned long long)(uintptr_t)sv.payload;
    for (unsigned long long need = __ballot(walk); need != 0ull;) {
        // the four lowest lanes that need a walk: one per group of 16 lanes (a block is walked 16 pieces = 256 bytes at a time,
        // and most walks end in their first or second step: the crossing is half a block in on average)
        int sl[4];
        uint32_t ng = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            sl[g] = need ? __ffsll((long long)need) - 1 : 0;
            if (need) { need &= need - 1ull; ng++; }
        }
        const int src = grp == 0u ? sl[0] : grp == 1u ? sl[1] : grp == 2u ? sl[2] : sl[3];
        const bool mine = grp < ng;                             // my group has a block to walk
        const uint32_t w_start = (uint32_t)__shfl((int)start, src, 64), w_len = (uint32_t)__shfl((int)len, src, 64);
        const uint32_t w_first = (uint32_t)__shfl((int)first, src, 64), w_x = (uint32_t)__shfl((int)x, src, 64), w_a = (uint32_t)__shfl((int)a, src, 64);
        const uint8_t *w_pay = (const uint8_t *)(uintptr_t)(unsigned long long)__shfl((long long)pay64, src, 64);
        const uint32_t np = mine ? (w_len + 15u) >> 4 : 0u;
        uint32_t docbase = w_first;                             // sum of all gap bits before the step's first piece
        bool open = mine;
        uint4 res = make_uint4(w_a, 0u, 0u, 0u);
        for (uint32_t pb = 0; __ballot(open && pb < np) != 0ull; pb += 16u) {       // (wave-uniform: a group that is done idles)
            const uint32_t pc = pb + hl;
            const bool pv = open && pc < np;
            uint32_t val[16], tmask = 0, w[4] = {0, 0, 0, 0}, prev = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) val[q] = 0;
            const uint32_t off = 16u * pc;
            if (pv) {
                const uint32_t rem = w_len - off;
                const uint8_t *pp = w_pay + w_start + off;
                const uint4 w4 = gload16(pp);                    // (segments carry 16 bytes of padding)
                prev = off ? gload4(pp - 4) : 0u;
                w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                if (rem < 16u) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t nj = rem > 4u * j ? (rem - 4u * j < 4u ? rem - 4u * j : 4u) : 0u;
                        w[j] &= nj >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nj)) - 1u);
                    }
                }
                uint32_t sh = 7u * ((uint32_t)__clz((int)~(prev | 0x7F7F7F7Fu)) >> 3);      // continuation bytes pending before my first byte
                uint32_t sum = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t c7 = __builtin_amdgcn_ubfe(w[q >> 2], 8 * (q & 3), 7);
                    const uint32_t cm = (uint32_t)__builtin_amdgcn_sbfe((int)w[q >> 2], 8 * (q & 3) + 7, 1);
                    sum += c7 << (sh & 31u);
                    sh = (sh + 7u) & cm;
                    tmask |= ~cm & (1u << q);
                    val[q] = sum;
                }
                if (rem < 16u) tmask &= (1u << rem) - 1u;
            }
            uint32_t incl = val[15];                              // prefix sum inside the group of 16 lanes = a DPP row
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
            const uint32_t base = docbase + incl - val[15];
            uint32_t cm = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) cm |= (base + val[q] >= w_x) ? 1u << q : 0u;
            cm &= tmask;
            const uint32_t hits = (uint32_t)(__ballot(pv && cm != 0u) >> (16u * grp)) & 0xFFFFu;
            uint4 r = make_uint4(0, 0, 0, 0);
            if (pv && cm != 0u) {
                // the first posting >= x ends at byte q of my piece: its varint back to front (the byte before its first one ends
                // the posting before, or is the 0 in front of the block's first byte), most significant group first
                const int q = __ffs((int)cm) - 1;
                uint32_t gap = 0, nbytes = 0;
                for (int j = q; j > q - 5; j--) {
                    const uint32_t wj = j < 0 ? prev : j < 4 ? w[0] : j < 8 ? w[1] : j < 12 ? w[2] : w[3];
                    const uint32_t c = (wj >> (8u * ((uint32_t)j & 3u))) & 0xFFu;
                    if (j != q && !(c & 0x80u)) break;
                    gap = (gap << 7) | (c & 0x7Fu);
                    nbytes++;
                }
                uint32_t vq = val[0];
#pragma unroll
                for (int t = 1; t < 16; t++) vq = t == q ? val[t] : vq;
                r = make_uint4((w_a - 1u) | CUT_INSIDE, w_start + off + (uint32_t)q + 1u - nbytes, base + vq - gap, 0u);
            }
            if (hits != 0u) {                                    // (uniform in the group)
                const int hsrc = __ffs((int)hits) - 1;
                res = make_uint4((uint32_t)__shfl((int)r.x, hsrc, 16), (uint32_t)__shfl((int)r.y, hsrc, 16), (uint32_t)__shfl((int)r.z, hsrc, 16), 0u);
                open = false;
            }
            docbase += row_last(incl, grp);
        }
        // the groups' results go to the lanes whose blocks they walked
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)res.x, 16 * g), ry = (uint32_t)__builtin_amdgcn_readlane((int)res.y, 16 * g),
                           rz = (uint32_t)__builtin_amdgcn_readlane((int)res.z, 16 * g);
            if ((uint32_t)g < ng && l == sl[g]) c0 = make_uint4(rx, ry, rz, 0u);
        }
    }
    if (!nonempty) return;
    if (x != 0u) {
        // the tiles of the same term right before this one: the empty ones in between and the one that ends where I begin
        for (uint32_t j = tile; j-- > 0u;) {
            const uint4 pd = p.desc[j];
            if (pd.x != td.x || !(pd.y & MERGE_DESC_LARGE)) break;
            if (pd.z > pd.w) continue;
            p.cut1[(uint64_t)s * p.cut_ss + (uint64_t)j * p.cut_st] = make_uint2(c0.x, c0.y);
            break;
        }
    }
    p.cut0[(uint64_t)s * p.cut_ss + (uint64_t)tile * p.cut_st] = c0;
    // the last tile of its term(s) ends where the lists end (a bitmap term's last window stops short of 2^32, and nothing follows it)
    bool last = tile + 1u == *p.n_tiles_dev;
    if (!last) { const uint4 nd = p.desc[tile + 1u]; last = nd.x != td.x || !(nd.y & MERGE_DESC_LARGE) || !(td.y & MERGE_DESC_LARGE); }
    if (last) p.cut1[(uint64_t)s * p.cut_ss + (uint64_t)tile * p.cut_st] = make_uint2(b_hi, 0u);
}

// ---- the tile kernel ----------------------------------------------------------------------
constexpr uint32_t PSLOT = PCAP / MT;             // piece slots per thread when the piece -> block map is built
static_assert(PCAP % MT == 0 && BMW % (2u * MT) == 0, "piece map: PSLOT slots per thread; bitmap tiles: BMW / MT words per thread");
static_assert(MERGE_NT_MAX <= 1024u && PCAP < 2048u, "BI packs term slot (10 bits), payload bytes (11 bits) and first piece (11 bits)");

struct __align__(16) MergeSmem {
    union {
        struct {
            uint32_t V[MCAP + 4u];          // the tile's postings: arrival order, then bucket order (+ 4: the ranking reads four entries from any bucket's base)
            uint16_t TG[MCAP];              // per posting: term slot (while decoding), then its bucket
            uint32_t C32[MCAP / 2u + 4u];   // bucket counters, then exclusive bucket bases: two 16-bit values per word
        } s;
        uint32_t bm[BMW];                   // bitmap tiles: one bit per doc of the tile's range
    } u;
    union {
        struct {                            // while a chunk of blocks is decoded
            uint32_t PX[PCAP];              // exclusive prefix of the pieces' gap sums
            uint16_t PJ[PCAP];              // block (thread of the chunk) that owns each piece
            const uint8_t *BP[MT];          // per block: its payload
            uint32_t BF[MT], BI[MT];        // per block: first doc; term slot | payload bytes << 10 | first piece << 21
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
    uint32_t lbase[MAX_LISTS];
    uint32_t RR0[MAX_LISTS], RR1[MAX_LISTS];   // block range of each run for the root doc range of the term(s) being merged
    uint32_t R0[MAX_LISTS];                 // first block of each run for the range being merged
    uint32_t RB[MAX_LISTS + 2u];            // exclusive prefix of the runs' block counts
    uint32_t wsum[MW], wmax[MW];
    uint32_t fill;                          // postings of the range that arrived in V (may exceed MCAP: the range is then split)
    uint32_t ovf;                           // a bucket overflowed
    uint32_t tp;                            // pieces of the chunk
    uint32_t ab;                            // allocation inside the term's parking region
    uint32_t tk;                            // the tile this workgroup works on next (ticket)
    uint32_t pq_n;                          // parked tiles that still wait for their place in the output
    uint32_t abort;                         // a bounded wait ran out somewhere: stop
    uint32_t pq_tile[MERGE_PQ], pq_cnt[MERGE_PQ];
    unsigned long long pq_slot[MERGE_PQ];
    unsigned long long pq_off;              // output offset of the queue's first tile
    uint32_t stk[36][2];                    // bisection stack of doc ranges (<= 32 levels, two pushes per pop)
};

// block-wide exclusive scan of one value per thread (MT threads); returns exclusive prefix, total in *tot
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *tot) {
    const int l = lane_id(), wv = (int)threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    lds_barrier();
    if (l == 63) wsum[wv] = incl;
    lds_barrier();
    uint32_t pre = 0;
    uint32_t t = 0;
    for (int w = 0; w < (int)MW; w++) { if (w < wv) pre += wsum[w]; t += wsum[w]; }
    *tot = t;
    return pre + incl - v;
}

// number of threads of the workgroup whose predicate holds
__device__ __forceinline__ uint32_t block_count(bool pred, uint32_t *wsum) {
    const unsigned long long m = __ballot(pred);
    lds_barrier();
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    lds_barrier();
    uint32_t t = 0;
    for (int w = 0; w < (int)MW; w++) t += wsum[w];
    return t;
}

// inclusive prefix maximum over the 64 lanes of a wave (same DPP steps as wave_incl_scan; lanes without a source see 0)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); x = y > x ? y : x;
    return x;
}

__global__ __launch_bounds__(MERGE_THREADS, 4) void k_merge_tiles(const MergeSegs *__restrict__ ms, MergeParams p) {
    __shared__ MergeSmem sm;
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t k = p.k;
    // diagnostics only: thread 0 sums the cycles spent in each step of the tile loop
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr;
    const uint32_t xskip = p.pad0;             // timing experiments only (option debug.merge_skip): phases left out, results wrong
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();

    const uint32_t *my_bo = nullptr;               // blk_off of "my" run (threads 0 .. k-1)
    if ((uint32_t)tid < k) {
        const SegView &sv = ms->segs[tid];
        my_bo = sv.blk_off;
        sm.pay[tid] = sv.payload;
        sm.skp[tid] = sv.skip;
        sm.bls[tid] = sv.blk_list;
        sm.lbase[tid] = sv.list_base;
    }
    lds_barrier();
    uint16_t *C16 = reinterpret_cast<uint16_t *>(sm.u.s.C32);

    const uint32_t n_tiles = *p.n_tiles_dev;       // computed by the plan kernels; the host only knows an upper bound
    MergeSync *sy = p.sync;
    // ---- direct placement (p.direct): the FIRST workgroup of the grid is the scanner (workgroups are dispatched in ascending
    // order: whenever a worker of this launch runs, its scanner runs too).  Tiles are claimed in ticket order and publish their
    // survivor counts (count + 1; 0 = not yet); the scanner's one wave turns the counts into output offsets as far as the
    // counts are contiguous (tile_off starts as all-ones: an entry is its own "ready" flag).  A tile workgroup parks its
    // survivors as always, goes on with its next tile and moves a parked tile to its final place once its offset is there -
    // normally one tile later, when everything that was in flight beside it has counted too.  No packing pass, no scan launch.
    // Only those two scalars cross workgroups (relaxed device-scope accesses, no fences: the L2s of the eight XCDs are not
    // coherent with each other, and a release would write a whole L2 back); the parked ids stay inside their workgroup's CU.
    // Every wait is bounded.
    const uint32_t spin_limit = p.spin_limit ? p.spin_limit : SPIN_LIMIT;
    if (p.direct && blockIdx.x == 0u) {
        if (xskip & 64u) return;                  // (tests: a scanner that never runs - the workers' bounded waits must end the launch)
        // all eight waves: 512 tiles per step (one wave alone - 64 device-scope loads and as many write-through stores per
        // step - tops out below the rate at which a full machine counts tiles, and the workers then queue up behind it)
        uint32_t base = 0, spins = 0;
        unsigned long long run = 0;
        unsigned long long it_all = 0, it_zero = 0, t_start = __builtin_amdgcn_s_memtime();
        uint32_t *sk = sm.RR0, *ss = sm.RR1;          // per wave: counted tiles in a row from the wave's first, their survivors
        while (base < n_tiles) {
            const uint32_t i = base + (uint32_t)tid;
            const uint32_t c1 = i < n_tiles ? __hip_atomic_load(&p.tile_count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            const unsigned long long m = __ballot(c1 != 0u);
            const uint32_t kk = m == ~0ull ? 64u : (uint32_t)__ffsll((long long)~m) - 1u;
            const uint32_t cnt = (uint32_t)l < kk && i < n_tiles ? c1 - 1u : 0u;
            const uint32_t incl = wave_incl_scan(cnt);
            if (l == 0) { sk[wv] = kk; ss[wv] = kk ? wave_bcast(incl, 0) : 0u; }
            const uint32_t wsum = kk ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)kk - 1) : 0u;
            if (l == 0) ss[wv] = wsum;
            lds_barrier();
            bool mine = true;                         // the waves before mine are complete: my counted tiles extend the row
            uint32_t pre = 0, K = 0, R = 0;
            bool open = true;
            for (uint32_t w = 0; w < MW; w++) {
                if (w < (uint32_t)wv) { mine = mine && sk[w] == 64u; pre += ss[w]; }
                if (open) { K += sk[w]; R += ss[w]; open = sk[w] == 64u; }
            }
            if (mine && (uint32_t)l < kk && i < n_tiles)
                __hip_atomic_store(&p.tile_off[i], (uint64_t)(run + pre + incl - cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            run += R;
            base += K;
            it_all++;
            if (K == 0u) it_zero++;
            lds_barrier();
            if (K == 0u) {
                if (++spins > spin_limit || __hip_atomic_load(&sy->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (tid == 0) __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(10);
            } else spins = 0;
        }
        if (tid == 0) { p.tile_off[n_tiles] = run; *p.d_total = run; }
        if (stamps && tid == 0) { p.debug[0] = it_all; p.debug[1] = it_zero; p.debug[2] = __builtin_amdgcn_s_memtime() - t_start; p.debug[3] = n_tiles; }
        return;
    }
    // moves parked tiles whose output offset is known to their final place; block: wait (bounded) until none is left waiting
    // probe: thread 0's early look at the first queued tile's offset (taken mid-tile, so that its latency is hidden); used once
    auto drain = [&](bool block, unsigned long long probe, uint32_t probe_tile) {
        uint32_t spins = 0;
        while (true) {
            lds_barrier();
            const uint32_t nq = sm.pq_n;
            if (nq == 0u) return;
            if (tid == 0) {
                if (probe != ~0ull && probe_tile == sm.pq_tile[0]) sm.pq_off = probe;
                else sm.pq_off = __hip_atomic_load(&p.tile_off[sm.pq_tile[0]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                probe = ~0ull;
            }
            __syncthreads();          // (also: the parked ids were stored by all waves of this workgroup - their stores have landed)
            if (sm.pq_off != ~0ull) {
                const uint32_t c = sm.pq_cnt[0];
                const uint32_t *src = p.tmp + sm.pq_slot[0];
                uint32_t *dstp = p.out_values + sm.pq_off;
                // 16 bytes per thread and step, four steps in flight (a plain element loop is one dependent load -> store per
                // thread at a time: latency-bound)
                for (uint32_t q0 = 0; q0 < c; q0 += 16u * MT) {
                    uint4 v4[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) {
                        const uint32_t q = q0 + 4u * ((uint32_t)tid + j * MT);
                        v4[j] = make_uint4(0, 0, 0, 0);
                        if (q + 4u <= c) __builtin_memcpy(&v4[j], src + q, 16);
                        else if (q < c) { v4[j].x = src[q]; if (q + 1u < c) v4[j].y = src[q + 1u]; if (q + 2u < c) v4[j].z = src[q + 2u]; }
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) {
                        const uint32_t q = q0 + 4u * ((uint32_t)tid + j * MT);
                        if (q + 4u <= c) __builtin_memcpy(dstp + q, &v4[j], 16);
                        else if (q < c) { dstp[q] = v4[j].x; if (q + 1u < c) dstp[q + 1u] = v4[j].y; if (q + 2u < c) dstp[q + 2u] = v4[j].z; }
                    }
                }
                lds_barrier();
                if (tid == 0) {
                    for (uint32_t z = 1; z < nq; z++) { sm.pq_tile[z - 1u] = sm.pq_tile[z]; sm.pq_cnt[z - 1u] = sm.pq_cnt[z]; sm.pq_slot[z - 1u] = sm.pq_slot[z]; }
                    sm.pq_n = nq - 1u;
                }
                spins = 0;
                continue;
            }
            if (!block) return;
            // a wait that runs out - mine, or (seen in the same round as my tile's offset) any other workgroup's or the scanner's -
            // ends this workgroup's part of the launch: the queue is dropped, the tile loop below stops claiming tiles
            if (tid == 0) sm.abort = (++spins > spin_limit || __hip_atomic_load(&sy->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1u : 0u;
            lds_barrier();
            if (sm.abort) {
                if (tid == 0) { __hip_atomic_store(&sy->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sm.pq_n = 0u; }
                continue;
            }
            __builtin_amdgcn_s_sleep(20);
        }
    };
    if (tid == 0) { sm.pq_n = 0u; sm.abort = 0u; sm.tk = p.direct ? atomicAdd(&sy->ticket, 1u) : blockIdx.x; }
    lds_barrier();
    const uint32_t n_workers = p.direct ? gridDim.x - 1u : gridDim.x;
    uint32_t tk_next = 0;                           // thread 0: the ticket after this one (claimed early: its latency hides behind the tile)
    // what a tile needs from global memory before it can start is fetched ahead: its descriptor, its runs and (one step later,
    // when the descriptor is there) its terms' plan entries
    uint32_t tile_nx = sm.tk;
    uint4 td_nx = tile_nx < n_tiles ? p.desc[tile_nx] : make_uint4(0, 0, 1, 0);
    uint4 c0_nx = make_uint4(0, 0, 0, 0);          // threads 0 .. k-1: where my list's part of the tile begins ...
    uint2 c1_nx = make_uint2(0, 0);                // ... and ends (plan: k_merge_tile_runs)
    auto fetch_cuts = [&]() {
        if (tile_nx < n_tiles && (uint32_t)tid < k) { c0_nx = p.cut0[(uint64_t)tid * p.cut_ss + (uint64_t)tile_nx * p.cut_st]; c1_nx = p.cut1[(uint64_t)tid * p.cut_ss + (uint64_t)tile_nx * p.cut_st]; }
    };
    fetch_cuts();
    for (uint32_t tile = tile_nx; tile < n_tiles; tile = tile_nx) {
        if (sm.abort) break;                       // (uniform: written before a barrier every thread has passed)
        const uint4 td = td_nx;
        const uint4 c0 = c0_nx;
        const uint2 c1 = c1_nx;
        bool nx_known = false;                      // tile_nx / td_nx / c0_nx / c1_nx hold the next tile
        if (!p.direct) {                            // static order: the next tile is known now
            tile_nx = tile + n_workers;
            nx_known = true;
            if (tile_nx < n_tiles) td_nx = p.desc[tile_nx];
            fetch_cuts();
        }
        // The next ticket is claimed now (its latency hides behind this tile) only while few parked tiles wait: a workgroup
        // that may have to wait for room in its queue must not own a tile it has not started - the scanner's frontier
        // would stand still at that tile, and every other queue would fill behind it.
        const bool claim_early = p.direct && sm.pq_n <= 2u;
        // thread 0 looks up the first queued tile's offset now; mid-tile - when the stores that parked it have long landed and
        // this tile's decode has hidden the look-up's latency - the queued tiles whose offsets are there move to their place
        unsigned long long probe = ~0ull;
        uint32_t probe_tile = 0xFFFFFFFFu;
        if (p.direct && tid == 0 && sm.pq_n) {
            probe_tile = sm.pq_tile[0];
            probe = __hip_atomic_load(&p.tile_off[probe_tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool drained_mid = false;
        if (claim_early && tid == 0) tk_next = atomicAdd(&sy->ticket, 1u);
        const uint32_t t0 = td.x, t1 = td.y & 0x3FFFFFFFu;
        // plan entries of the tile's terms (a batch: one term per thread), in flight while the tile is decoded
        const uint32_t my_t = t0 + ((uint32_t)tid < t1 - t0 ? (uint32_t)tid : 0u);
        const uint32_t pf_tn = p.tn[my_t], pf_mn = p.tmin[my_t], pf_mx = p.tmax[my_t];
        const uint32_t pf_mn0 = p.tmin[t0], pf_mx0 = p.tmax[t0];
        const bool large = (td.y & MERGE_DESC_LARGE) != 0u, bm_tile = (td.y & MERGE_DESC_BITMAP) != 0u;
        // Work list of the tile: doc ranges of cur_t0 .. cur_t0 + cur_nt, handled one after the other in doc order, their
        // survivors appended at dst + acc.  Normally one range (the tile's).  A range that does not fit LDS or whose ids are
        // clustered is bisected until its pieces fit the bitmap (exact for any input); a batch whose sort fails is redone
        // term by term.
        unsigned long long slot = p.npre[t0];
        uint32_t *dst = p.tmp + slot;
        bool allocated = !large;                   // tiles of a large term take their room from the term's bump allocator
        uint32_t acc = 0;
        uint32_t cur_t0 = t0, cur_nt = t1 - t0;
        bool cur_batch = !large;
        uint32_t root_mode = 1u;                   // how the runs of a root range are found (see below)
        bool root = true;
        uint32_t fb_next = 0, fb_end = 0, fb_acc0 = 0;
        bool fb_active = false;
        uint32_t sp = td.z <= td.w ? 1u : 0u;
        lds_barrier();
        if (tid == 0) { sm.stk[0][0] = td.z; sm.stk[0][1] = td.w; }

        // room for c survivors: every thread calls it, once per range, before the range's survivors are written
        auto alloc = [&](uint32_t c) -> uint32_t * {
            if (!allocated) {
                lds_barrier();
                if (tid == 0) sm.ab = c ? atomicAdd(&p.term_alloc[t0], c) : 0u;
                lds_barrier();
                slot = p.npre[t0] + sm.ab;
                dst = p.tmp + slot;
                allocated = true;
            }
            return dst + acc;
        };

        // Direct placement, at the tile's end: its count is published, the older parked tiles whose offsets have arrived move to
        // their final place, and the tile joins the queue.  (Tried: doing this the moment the count is known, before the
        // tile's own survivors are stored - the scanner sees counts earlier and the barrier in drain() has nothing recent to wait
        // for, but the tile's stores then start later: 12.7 -> 13.7 ms on C3.)
        bool published = false;
        auto publish = [&](uint32_t cnt) {
            lds_barrier();
            if (tid == 0) __hip_atomic_store(&p.tile_count[tile], cnt + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // counted: the scanner may pass this tile
            if (sm.pq_n >= MERGE_PQ || !drained_mid) drain(sm.pq_n >= MERGE_PQ, probe, probe_tile);     // (normally done mid-tile; a full queue waits for its oldest entry)
            probe = ~0ull;
            lds_barrier();
            if (tid == 0 && cnt) { const uint32_t z = sm.pq_n; sm.pq_tile[z] = tile; sm.pq_cnt[z] = cnt; sm.pq_slot[z] = slot; sm.pq_n = z + 1u; }
            published = true;
        };
        while (true) {
            if (sp == 0u) {
                if (fb_active) {                       // a term of a batch that is redone term by term is complete
                    lds_barrier();
                    if (tid == 0) p.out_counts[cur_t0] = acc - fb_acc0;
                }
                if (fb_next >= fb_end) break;
                cur_t0 = fb_next++;
                cur_nt = 1u;
                cur_batch = false;
                fb_acc0 = acc;
                fb_active = true;
                root = true;
                root_mode = 0u;
                lds_barrier();
                if (tid == 0) { sm.stk[0][0] = 0u; sm.stk[0][1] = 0xFFFFFFFFu; }
                sp = 1u;
            }
            lds_barrier();
            const uint32_t lo = sm.stk[sp - 1u][0], hi = sm.stk[sp - 1u][1];
            sp--;
            // ---- the runs of [lo, hi]: which blocks of every list have to be decoded.  mode 1: the tile's own range, between
            // the cuts the plan made (the first and the last block of a run may be partial); 0: the whole lists of a term (a
            // batch redone term by term); 2: a sub-range of the root (searched here: whole blocks, ids outside are dropped)
            const bool cutmode = root && root_mode == 1u;
            {
                const uint32_t mode = root ? root_mode : 2u;
                lds_barrier();
                if ((uint32_t)tid < k) {
                    uint2 r;
                    if (mode == 0u) r = make_uint2(my_bo[cur_t0], my_bo[cur_t0 + cur_nt]);
                    else if (mode == 1u) {
                        const uint32_t bs = c0.x & ~CUT_INSIDE, be = (c1.x & ~CUT_INSIDE) + (c1.x >> 31);
                        r = make_uint2(bs, be > bs ? be : bs);
                    } else r = blocks_of_range(sm.skp[tid], sm.RR0[tid], sm.RR1[tid], lo, hi);
                    if (mode != 2u) { sm.RR0[tid] = r.x; sm.RR1[tid] = r.y; }
                    sm.R0[tid] = r.x;
                    sm.RB[tid] = r.y - r.x;            // (count; scanned below)
                }
                lds_barrier();
                if (tid < 64) {
                    const uint32_t c = (uint32_t)l < k ? sm.RB[l] : 0u;
                    const uint32_t incl = wave_incl_scan(c);
                    if ((uint32_t)l < k) sm.RB[l] = incl - c;
                    if (l == 63) sm.RB[k] = incl;
                    if (l == 0) { sm.fill = 0u; sm.ovf = 0u; }
                }
                lds_barrier();
            }
            const bool was_root = root;
            root = false;
            const uint32_t NB = sm.RB[k];
            if (NB == 0u || (xskip & 8u)) continue;
            const bool BM = (was_root && bm_tile) || (!was_root && hi - (lo & ~31u) < BMW * 32u);
            const uint32_t lo32 = lo & ~31u;
            const bool filter = !cutmode && !(lo == 0u && hi == 0xFFFFFFFFu);       // (between cuts every decoded id is the tile's)
            const uint32_t bm_nw = BM ? ((hi - lo32) >> 5) + 1u : 0u;                      // <= BMW
            if (BM) {
                for (uint32_t i = 4u * (uint32_t)tid; i < bm_nw; i += 4u * MT) *reinterpret_cast<uint4 *>(&sm.u.bm[i]) = make_uint4(0, 0, 0, 0);
            }
            II2_STAMP(0)

            // ---- decode the runs' blocks; every posting with lo <= id <= hi goes
            //   BM: into the bitmap of the range (bit id - lo32);  else: to V[arrival order] (its term slot to TG in a batch).
            // A chunk = as many consecutive blocks (one per thread) as have PCAP 16-byte payload pieces between them; a thread
            // then walks one piece: the 16 partial gap sums and which bytes end a posting; a scan over the chunk's pieces gives
            // every piece the sum before it, and the difference to its block's first piece the id it starts from.
            for (uint32_t g0 = 0; g0 < NB;) {
                lds_barrier();                     // the chunk before is done with the tables
                const uint32_t g = g0 + (uint32_t)tid;
                const bool valid = g < NB;
                uint32_t np = 0, first = 0, ti = 0, len = 0;
                bool has_first = true;                // the block's first posting (the one in its skip entry) is the tile's
                const uint8_t *bp = nullptr;
                if (valid) {
                    uint32_t a = 0, e = k;            // RB[a] <= g < RB[e]
                    while (e - a > 1u) { const uint32_t m = (a + e) >> 1; if (sm.RB[m] <= g) a = m; else e = m; }
                    const uint32_t b = sm.R0[a] + (g - sm.RB[a]);
                    const ii2_skip *sk = sm.skp[a];
                    const ii2_skip e0 = sk[b];
                    uint32_t from = e0.byte_off, to = sk[b + 1u].byte_off;
                    first = e0.first_doc;
                    if (cutmode) {                    // the run's first / last block may be cut (all four loads are in flight together)
                        if (g == sm.RB[a]) {
                            const uint4 cs = p.cut0[(uint64_t)a * p.cut_ss + (uint64_t)tile * p.cut_st];
                            if (cs.x & CUT_INSIDE) { from = cs.y; first = cs.z; has_first = false; }
                        }
                        if (g + 1u == sm.RB[a + 1u]) {
                            const uint2 ce = p.cut1[(uint64_t)a * p.cut_ss + (uint64_t)tile * p.cut_st];
                            if (ce.x & CUT_INSIDE) to = ce.y;
                        }
                    }
                    len = to > from ? to - from : 0u;
                    if (len > 1280u) len = 1280u;     // (a block holds <= 256 postings of <= 5 bytes; imported segments are validated)
                    np = (len + 15u) >> 4;
                    bp = sm.pay[a] + from;
                    if (cur_batch) { ti = sm.bls[a][b] - sm.lbase[a] - cur_t0; ti = ti < cur_nt ? ti : cur_nt - 1u; }
                }
#pragma unroll
                for (uint32_t j = 0; j < PSLOT; j++) sm.x.d.PJ[(uint32_t)tid + j * MT] = 0;
                uint32_t tot;
                const uint32_t pex = block_excl_scan(np, sm.wsum, &tot);
                const bool ok = valid && pex + np <= PCAP;          // a prefix of the threads (pex ascends); never empty (np <= 80)
                const uint32_t nchunk = block_count(ok, sm.wsum);
                if (ok) {
                    sm.x.d.BF[tid] = first;
                    sm.x.d.BP[tid] = bp;
                    sm.x.d.BI[tid] = ti | (len << 10) | (pex << 21);
                    if (np) sm.x.d.PJ[pex] = (uint16_t)(tid + 1);     // marks the block's first piece
                    if ((uint32_t)tid == nchunk - 1u) sm.tp = pex + np;
                }
                {   // the blocks' first postings (their ids are in the skip entries)
                    const bool in = ok && has_first && first >= lo && first <= hi;
                    if (BM) {
                        if (in) atomicOr(&sm.u.bm[(first - lo32) >> 5], 1u << (first & 31u));
                    } else {
                        const unsigned long long m = __ballot(in);
                        if (m != 0ull) {
                            uint32_t wb = 0;
                            if (l == 0) wb = atomicAdd(&sm.fill, (uint32_t)__popcll(m));
                            wb = wave_bcast(wb, 0);
                            const uint32_t pos = wb + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
                            if (in && pos < MCAP) { sm.u.s.V[pos] = first; if (cur_batch) sm.u.s.TG[pos] = (uint16_t)ti; }
                        }
                    }
                }
                lds_barrier();
                const uint32_t tp = sm.tp;
                {   // piece -> block: inclusive prefix maximum over the marks, PSLOT consecutive pieces per thread
                    uint32_t m[PSLOT], run = 0;
#pragma unroll
                    for (uint32_t j = 0; j < PSLOT; j++) { m[j] = sm.x.d.PJ[PSLOT * (uint32_t)tid + j]; run = m[j] > run ? m[j] : run; m[j] = run; }
                    const uint32_t wi = wave_incl_max(run);
                    if (l == 63) sm.wmax[wv] = wi;
                    uint32_t before = (uint32_t)__shfl_up((int)wi, 1, 64);
                    if (l == 0) before = 0u;
                    lds_barrier();
                    for (int w2 = 0; w2 < wv; w2++) before = sm.wmax[w2] > before ? sm.wmax[w2] : before;
#pragma unroll
                    for (uint32_t j = 0; j < PSLOT; j++) sm.x.d.PJ[PSLOT * (uint32_t)tid + j] = (uint16_t)(m[j] > before ? m[j] : before);
                }
                lds_barrier();
                uint32_t carry = 0;
                for (uint32_t it = 0; it < tp; it += MT) {
                    const uint32_t pc = it + (uint32_t)tid;
                    const bool pv = pc < tp;
                    uint32_t jb = 0, val[16], tmask = 0, bi = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) val[i] = 0;
                    if (pv) {
                        jb = (uint32_t)sm.x.d.PJ[pc] - 1u;
                        bi = sm.x.d.BI[jb];
                        const uint32_t off = 16u * (pc - (bi >> 21));
                        const uint32_t rem = ((bi >> 10) & 0x7FFu) - off;                 // payload bytes from my first one on (>= 1)
                        const uint8_t *pp = sm.x.d.BP[jb] + off;
                        const uint4 w4 = gload16(pp);                          // (segments carry 16 bytes of padding)
                        const uint32_t prev = off ? gload4(pp - 4) : 0u;
                        uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
                        if (rem < 16u) {      // a block's last piece: the bytes past its end count as zero gaps (the sums stay exact ids)
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t nj = rem > 4u * j ? (rem - 4u * j < 4u ? rem - 4u * j : 4u) : 0u;
                                w[j] &= nj >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nj)) - 1u);
                            }
                        }
                        // continuation bytes pending right before my first byte (varints are <= 5 bytes)
                        uint32_t sh = 7u * ((uint32_t)__clz((int)~(prev | 0x7F7F7F7Fu)) >> 3);
                        uint32_t sum = 0;
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const uint32_t c7 = __builtin_amdgcn_ubfe(w[i >> 2], 8 * (i & 3), 7);
                            const uint32_t cm = (uint32_t)__builtin_amdgcn_sbfe((int)w[i >> 2], 8 * (i & 3) + 7, 1);    // -1: continuation byte
                            sum += c7 << (sh & 31u);
                            sh = (sh + 7u) & cm;
                            tmask |= ~cm & (1u << i);
                            val[i] = sum;
                        }
                        if (rem < 16u) tmask &= (1u << rem) - 1u;      // bytes past the block's end are not postings
                    }
                    if (xskip & 1u) tmask = 0;
                    uint32_t tots;
                    const uint32_t pex2 = carry + block_excl_scan(val[15], sm.wsum, &tots);
                    carry += tots;
                    if (pv) sm.x.d.PX[pc] = pex2;
                    lds_barrier();
                    const uint32_t base = pv ? sm.x.d.BF[jb] + pex2 - sm.x.d.PX[bi >> 21] : 0u;
                    if (BM) {
                        if (pv) {
                            const uint32_t rb = base - lo32, span = hi - lo32;
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                const uint32_t rel = rb + val[i];
                                if (((tmask >> i) & 1u) && rel <= span) atomicOr(&sm.u.bm[rel >> 5], 1u << (rel & 31u));
                            }
                        }
                    } else {
                        // pieces that reach over an end of the range drop the ids outside it (the others skip the test)
                        if (filter && __ballot(pv && (base + val[0] < lo || base + val[15] > hi)) != 0ull) {
                            uint32_t im = 0;
#pragma unroll
                            for (int i = 0; i < 16; i++) im |= (base + val[i] - lo <= hi - lo) ? 1u << i : 0u;
                            tmask &= im;
                        }
                        const uint32_t c = (uint32_t)__popc(tmask);
                        const uint32_t incl = wave_incl_scan(c);
                        const uint32_t wtot = wave_bcast(incl, 63);
                        if (wtot) {
                            uint32_t wb = 0;
                            if (l == 0) wb = atomicAdd(&sm.fill, wtot);
                            wb = wave_bcast(wb, 0);
                            uint32_t pos = wb + incl - c;
                            if (wb + wtot <= MCAP) {
                                const uint16_t tg = (uint16_t)(bi & 0x3FFu);
#pragma unroll
                                for (int i = 0; i < 16; i++) {
                                    if ((tmask >> i) & 1u) {
                                        sm.u.s.V[pos] = base + val[i];
                                        if (cur_batch) sm.u.s.TG[pos] = tg;
                                        pos++;
                                    }
                                }
                            }       // else: the range holds more than LDS (fill > MCAP): it is split below
                        }
                    }
                }
                g0 += nchunk;
            }
            lds_barrier();
            II2_STAMP(BM ? 5 : (cur_batch ? 1 : 3))
            if (was_root) {
                if (p.direct && !drained_mid) { drain(false, probe, probe_tile); probe = ~0ull; drained_mid = true; }
                if (!nx_known && claim_early) {    // the next ticket has arrived by now: hand it round and fetch that tile's descriptor and runs
                    if (tid == 0) sm.tk = tk_next;
                    lds_barrier();
                    tile_nx = sm.tk;
                    nx_known = true;
                    if (tile_nx < n_tiles) td_nx = p.desc[tile_nx];
                    fetch_cuts();
                }
            }

            if (BM) {
                // ---- bitmap: the union, the dedupe and the order came with the representation; clear the tombstoned docs
                // word by word (coalesced loads of exactly the range's words), count, extract
                uint32_t *bm = sm.u.bm;
                if (tid == 0) {     // a sub-range need not start or end at a word
                    if (lo & 31u) bm[0] &= ~((1u << (lo & 31u)) - 1u);
                    if ((hi & 31u) != 31u) bm[bm_nw - 1u] &= (2u << (hi & 31u)) - 1u;
                }
                if (p.tomb) {
                    // all of a thread's tombstone words are requested before the first is used (one round of memory latency per tile)
                    constexpr uint32_t TW = BMW / MT / 2u;      // (two rounds: all at once would spill registers)
                    const uint32_t twb = lo32 >> 5;
#pragma unroll 1
                    for (uint32_t h = 0; h < 2u; h++) {
                        uint32_t tw[TW];
#pragma unroll
                        for (uint32_t j = 0; j < TW; j++) {
                            const uint32_t i = (uint32_t)tid + (h * TW + j) * MT;
                            tw[j] = (i < bm_nw && twb + i < p.tomb_nwords) ? p.tomb[twb + i] : 0u;
                        }
#pragma unroll
                        for (uint32_t j = 0; j < TW; j++)
                            if (tw[j]) bm[(uint32_t)tid + (h * TW + j) * MT] &= ~tw[j];
                    }
                }
                lds_barrier();
                // a wave takes a stretch of consecutive words and goes through it 64 words at a time, a word per lane: lanes
                // next to each other write ids next to each other (a thread that extracts a stretch of its own writes 4 bytes
                // per lane to 64 different cache lines with every store)
                const uint32_t wpw = ((bm_nw + 64u * MW - 1u) / (64u * MW)) * 64u;      // words per wave (a multiple of 64)
                const uint32_t ww0 = wpw * (uint32_t)wv < bm_nw ? wpw * (uint32_t)wv : bm_nw;
                const uint32_t ww1 = ww0 + wpw < bm_nw ? ww0 + wpw : bm_nw;
                uint32_t cnt = 0;
                for (uint32_t w = ww0 + (uint32_t)l; w < ww1; w += 64u) cnt += (uint32_t)__popc(bm[w]);
                uint32_t tot;
                const uint32_t tpos = block_excl_scan(cnt, sm.wsum, &tot);
                uint32_t off = wave_bcast(tpos, 0);               // ids before my wave's stretch
                uint32_t *out = alloc(tot);
                for (uint32_t r = ww0; r < ww1; r += 64u) {
                    const uint32_t w = r + (uint32_t)l;
                    uint32_t x = w < ww1 ? bm[w] : 0u;
                    const uint32_t c = (uint32_t)__popc(x);
                    const uint32_t incl = wave_incl_scan(c);
                    uint32_t q = off + incl - c;
                    const uint32_t base = lo32 + 32u * w;
                    while (x) {
                        out[q++] = base + (uint32_t)__ffs((int)x) - 1u;
                        x &= x - 1u;
                    }
                    off += wave_bcast(incl, 63);
                }
                acc += tot;
                II2_STAMP(6)
                continue;
            }

            // ---- sort the decoded postings by buckets and write the survivors in (term, id) order
            if (xskip & 2u) { acc += 0; continue; }
            const uint32_t n = sm.fill;
            bool sorted = n <= MCAP;
            // a large term's range takes room for all its input postings from the term's bump allocator (inputs never
            // outnumber the region); the atomic is issued here and its answer is only needed when the survivors are written
            uint32_t ab_reg = 0;                    // (thread 0 parks the answer in LDS only when it is about to be needed)
            if (!allocated && tid == 0) ab_reg = n ? atomicAdd(&p.term_alloc[t0], n) : 0u;
            const uint32_t nj = (n + MT - 1u) / MT;               // sort passes that have postings (<= EPT)
            uint32_t *C32 = sm.u.s.C32;
            uint32_t u_mn = 0, u_nbm1 = 0;
            float u_scale = 0.0f;
            unsigned long long slots = 0ull;
            if (sorted) {
                for (uint32_t i = (uint32_t)tid; i < MCAP / 2u + 4u; i += MT) C32[i] = 0u;
                if ((uint32_t)tid < MCAP / 32u + 1u) sm.x.f.DB[tid] = 0u;
                // bucket maps: term t of a batch owns floor(n_t * MCAP / n) buckets, a range tile all MCAP of them
                if (cur_batch) {
                    uint32_t nbk = 0, mn = 0, mx = 0;
                    if ((uint32_t)tid < cur_nt) {
                        const uint32_t n_t = pf_tn;            // (a batch is always the tile's own terms)
                        mn = pf_mn;
                        mx = pf_mx;
                        nbk = n_t ? (uint32_t)(((uint64_t)n_t * MCAP) / n) : 0u;
                    }
                    uint32_t totb;
                    const uint32_t tb = block_excl_scan(nbk, sm.wsum, &totb);
                    if ((uint32_t)tid < cur_nt) {
                        sm.x.f.TB[tid] = (uint16_t)tb;
                        sm.x.f.TT[tid] = make_uint2(mn, __float_as_uint(nbk ? (float)nbk / ((float)(mx - mn) + 1.0f) : 0.0f));
                    }
                    if (tid == 0) sm.x.f.TB[cur_nt] = (uint16_t)totb;
                } else {
                    const uint32_t t_mn = cur_t0 == t0 ? pf_mn0 : p.tmin[cur_t0], t_mx = cur_t0 == t0 ? pf_mx0 : p.tmax[cur_t0];
                    u_mn = lo > t_mn ? lo : t_mn;
                    const uint32_t mxr = hi < t_mx ? hi : t_mx;
                    u_nbm1 = MCAP - 1u;
                    u_scale = (float)MCAP / ((float)((mxr > u_mn ? mxr : u_mn) - u_mn) + 1.0f);
                }
                lds_barrier();
                // ---- bucket of every posting, slot inside the bucket.  No per-lane branches: a lane past the tile's last
                // posting works on a dummy (term slot 0, counts nothing, its bucket id lands in an unused TG entry) - the
                // exec-mask bookkeeping of a predicated body was a quarter of this pass's issue slots
                bool over = false;
#pragma unroll
                for (uint32_t j = 0; j < EPT; j++) {
                    const uint32_t i = (uint32_t)tid + j * MT;
                    if (j < nj) {
                        const bool valid = i < n;
                        const uint32_t v = sm.u.s.V[i];
                        uint32_t mn = u_mn, nbm1 = u_nbm1, tb = 0;
                        float sc = u_scale;
                        if (cur_batch) {
                            const uint32_t tg0 = sm.u.s.TG[i];
                            const uint32_t ti = valid ? tg0 : 0u;
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
                        uint32_t slot_ = (atomicAdd(&C32[b >> 1], valid ? 1u << sh : 0u) >> sh) & 0xFFFFu;
                        over = over || (valid && slot_ > BKT_LIMIT);
                        slot_ = slot_ < BKT_LIMIT ? slot_ : BKT_LIMIT;
                        slots |= (unsigned long long)slot_ << (4u * j);
                    }
                }
                if (over) sm.ovf = 1u;
                lds_barrier();
                sorted = sm.ovf == 0u;
            }
            if (sorted) {
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
                lds_barrier();
                // ---- scatter into bucket order (in place: all reads, then all writes)
                {
                    uint32_t v[EPT], pk[EPT], ts[EPT];
#pragma unroll
                    for (uint32_t j = 0; j < EPT; j++) {
                        const uint32_t i = (uint32_t)tid + j * MT;
                        v[j] = 0; pk[j] = 0; ts[j] = 0;
                        if (j < nj) {           // (no per-lane branch: a lane past the last posting moves its unused entry onto itself)
                            const bool valid = i < n;
                            v[j] = sm.u.s.V[i];
                            const uint32_t b0 = sm.u.s.TG[i];
                            const uint32_t b = valid ? b0 : 0u;
                            const uint32_t to = (uint32_t)C16[b] + (uint32_t)((slots >> (4u * j)) & 15ull);
                            pk[j] = (valid ? to : i) | (b << 16);
                            // tombstones: most tests stop at the summary (1 bit per 1 << TOMB_SUM_SHIFT docs, mostly L2-resident)
                            if (p.tomb && (v[j] >> 5) < p.tomb_nwords) ts[j] = p.tomb_summary[v[j] >> (5u + TOMB_SUM_SHIFT)];
                        }
                    }
                    lds_barrier();
#pragma unroll
                    for (uint32_t j = 0; j < EPT; j++) {
                        if (j < nj) {
                            uint32_t tag = pk[j] >> 16;
                            if ((ts[j] >> ((v[j] >> TOMB_SUM_SHIFT) & 31u)) & 1u) tag |= ((p.tomb[v[j] >> 5] >> (v[j] & 31u)) & 1u) << 15;     // rarely: the bitmap itself
                            sm.u.s.V[pk[j] & 0xFFFFu] = v[j];
                            sm.u.s.TG[pk[j] & 0xFFFFu] = (uint16_t)tag;
                        }
                    }
                }
                lds_barrier();
                if (xskip & 4u) continue;
                // ---- every posting ranks itself inside its bucket: final position, duplicate / tombstone flag
                uint32_t fv[EPT], fp[EPT];
#pragma unroll
                for (uint32_t j = 0; j < EPT; j++) {
                    const uint32_t q = (uint32_t)tid + j * MT;
                    fv[j] = 0; fp[j] = 0x80000000u;
                    if (j < nj) {               // (no per-lane branch: a lane past the last posting ranks a dummy in bucket 0 and drops it)
                        const bool valid = q < n;
                        const uint32_t v = sm.u.s.V[q];
                        const uint32_t tg0 = sm.u.s.TG[q];
                        const uint32_t tg = valid ? tg0 : 0u;                     // bucket | tombstoned << 15
                        const uint32_t b = tg & 0x7FFFu;
                        const uint32_t blo = C16[b], nb = (uint32_t)C16[b + 1u] - blo;
                        const uint32_t qi = q - blo;                              // my place in the bucket
                        // the first four of the bucket without a loop (buckets hold one posting on average: lanes that loop make
                        // the whole wave wait for the fullest bucket among its 64).  A place past the bucket counts as my own id
                        // at a later place: neither smaller nor an earlier copy.
                        const uint32_t *B4 = &sm.u.s.V[blo];
                        // (all four are read whatever the bucket holds and the ones past the bucket are replaced afterwards: selects
                        // instead of predicated loads)
                        const uint32_t r0 = B4[0], r1 = B4[1], r2 = B4[2], r3 = B4[3];
                        const uint32_t u0 = r0, u1 = nb > 1u ? r1 : v, u2 = nb > 2u ? r2 : v, u3 = nb > 3u ? r3 : v;
                        const uint32_t e0 = (u0 == v && 0u < qi) ? 1u : 0u, e1 = (u1 == v && 1u < qi) ? 1u : 0u;
                        const uint32_t e2 = (u2 == v && 2u < qi) ? 1u : 0u, e3 = (u3 == v && 3u < qi) ? 1u : 0u;
                        uint32_t r = (u0 < v ? 1u : 0u) + (u1 < v ? 1u : 0u) + (u2 < v ? 1u : 0u) + (u3 < v ? 1u : 0u) + e0 + e1 + e2 + e3;
                        uint32_t dead = e0 | e1 | e2 | e3 | (tg >> 15);
#pragma unroll 1
                        for (uint32_t m = 4u; m < nb; m++) {
                            const uint32_t u = B4[m];
                            const uint32_t eq = (u == v && m < qi) ? 1u : 0u;
                            r += (u < v ? 1u : 0u) + eq;
                            dead |= eq;
                        }
                        const uint32_t P = blo + r;
                        if (dead && valid) atomicOr(&sm.x.f.DB[P >> 5], 1u << (P & 31u));
                        fv[j] = v;
                        fp[j] = valid ? P | (dead << 31) : 0x80000000u;
                    }
                }
                lds_barrier();
                // ---- dead ids before every 32 sorted positions
                const uint32_t nw = (n + 31u) >> 5;
                uint32_t totdead;
                {
                    const uint32_t x = (uint32_t)tid < nw ? (uint32_t)__popc(sm.x.f.DB[tid]) : 0u;
                    const uint32_t ex = block_excl_scan(x, sm.wsum, &totdead);
                    if ((uint32_t)tid <= nw) sm.x.f.DP[tid] = ex;       // (DP[nw] = all of them)
                }
                const uint32_t nout = n - totdead;
                if (!allocated && tid == 0) sm.ab = ab_reg;
                lds_barrier();
                if (!allocated) { slot = p.npre[t0] + sm.ab; dst = p.tmp + slot; allocated = true; }
                uint32_t *out = dst + acc;
                auto dead_before = [&](uint32_t x) -> uint32_t {
                    return sm.x.f.DP[x >> 5] + (uint32_t)__popc(sm.x.f.DB[x >> 5] & ((1u << (x & 31u)) - 1u));
                };
#pragma unroll
                for (uint32_t j = 0; j < EPT; j++) {
                    if (!(fp[j] >> 31)) out[fp[j] - dead_before(fp[j])] = fv[j];       // lanes hold neighbouring ranks: coalesced
                }
                if (cur_batch && (uint32_t)tid < cur_nt) {
                    // term t sits at the sorted positions [base of its first bucket, base of the next term's first bucket)
                    const uint32_t sp0 = C16[sm.x.f.TB[tid]], sp1 = C16[sm.x.f.TB[tid + 1]];
                    p.out_counts[cur_t0 + (uint32_t)tid] = (sp1 - sp0) - (dead_before(sp1) - dead_before(sp0));
                }
                acc += nout;
                II2_STAMP(cur_batch ? 2 : 4)
                continue;
            }
            // ---- the range does not fit LDS, or its ids are clustered so that a bucket overflowed
            if (cur_batch) {
                fb_next = cur_t0;                  // the batch again, term by term (nothing of it was written)
                fb_end = cur_t0 + cur_nt;
                acc = 0;
                continue;
            }
            const uint32_t mid = lo + ((hi - lo) >> 1);     // lo < hi here: a range of one doc fits the bitmap
            if (!allocated && tid == 0) sm.ab = ab_reg;
            lds_barrier();
            if (!allocated) { slot = p.npre[t0] + sm.ab; dst = p.tmp + slot; allocated = true; }     // (room for all n postings, see above)
            if (tid == 0) {
                sm.stk[sp][0] = mid + 1u; sm.stk[sp][1] = hi;
                sm.stk[sp + 1u][0] = lo;  sm.stk[sp + 1u][1] = mid;
            }
            sp += 2u;
            II2_STAMP(7)
        }
        auto take_next = [&]() {                    // the next ticket goes round; that tile's descriptor and runs are fetched
            lds_barrier();
            tile_nx = sm.tk;
            nx_known = true;
            if (tile_nx < n_tiles) td_nx = p.desc[tile_nx];
            fetch_cuts();
        };
        if (!nx_known && claim_early) {             // (a tile without a root range to decode: the early ticket was not handed round yet)
            if (tid == 0) sm.tk = tk_next;
            take_next();
        }
        if (!p.direct) {
            if (tid == 0) { p.tile_count[tile] = acc; p.tile_slot[tile] = slot; }
        } else {
            if (!published) publish(acc);
            lds_barrier();
            if (!nx_known) {                        // no early claim: the next tile is claimed only now, with room in the queue
                if (tid == 0) sm.tk = atomicAdd(&sy->ticket, 1u);
                take_next();
            }
        }
        II2_STAMP(7)
    }
    if (p.direct) drain(true, ~0ull, 0xFFFFFFFFu);
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
// (and what else the host reads at the end of a merge, so that ONE copy fetches it: mail[3] = the direct placement's error word,
// mail[4] = the number of tiles)
__global__ __launch_bounds__(256) void k_count_nonzero(const uint32_t *__restrict__ v, uint64_t n, uint64_t *__restrict__ out,
                                                       const uint32_t *__restrict__ err, const uint32_t *__restrict__ n_tiles, uint64_t *__restrict__ mail) {
    __shared__ uint32_t wsum[4];
    if (mail && blockIdx.x == 0 && threadIdx.x == 0) {
        mail[3] = err ? (uint64_t)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        mail[4] = n_tiles ? (uint64_t)*n_tiles : 0ull;
    }
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
    if (p.n_terms < 65536u) hipLaunchKernelGGL(k_mp_terms_few, dim3(grid_for(((uint64_t)p.n_terms + 1) * 16u)), dim3(256), 0, s, ms, p);
    else hipLaunchKernelGGL(k_mp_terms, dim3(grid_for(p.n_terms + 1)), dim3(256), 0, s, ms, p);
    return hipGetLastError();
}
// everything a call clears before its tile kernel, in one launch (four memsets were four dependent launches of ~5 us each): the
// result words, the per-term output counts, the tickets / tile counts / bump allocators, and - direct placement - the tiles'
// offsets, all ones = "not known yet"
__global__ __launch_bounds__(256) void k_merge_init(uint32_t *__restrict__ mail, uint32_t n_mail, uint32_t *__restrict__ cnt, uint64_t n_cnt,
                                                    uint32_t *__restrict__ aux, uint64_t n_aux, uint32_t *__restrict__ off, uint64_t n_off) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_mail) mail[t] = 0u;
    for (uint64_t i = t; i < n_cnt; i += stride) cnt[i] = 0u;
    for (uint64_t i = t; i < n_aux; i += stride) aux[i] = 0u;
    for (uint64_t i = t; i < n_off; i += stride) off[i] = 0xFFFFFFFFu;
}
hipError_t launch_merge_init(uint64_t *mail, uint32_t n_mail_words64, uint32_t *cnt, uint64_t n_cnt, void *aux, uint64_t aux_bytes, uint64_t *tile_off,
                             uint64_t n_off, hipStream_t s) {
    const uint64_t most = std::max<uint64_t>(std::max<uint64_t>(n_cnt, aux_bytes / 4u), 2u * n_off);
    const unsigned grid = (unsigned)std::min<uint64_t>((most + 1023u) / 1024u + 1u, 4096u);      // (four words per thread and round)
    hipLaunchKernelGGL(k_merge_init, dim3(grid), dim3(256), 0, s, (uint32_t *)mail, 2u * n_mail_words64, cnt, n_cnt, (uint32_t *)aux, aux_bytes / 4u,
                       (uint32_t *)tile_off, 2u * n_off);
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
    if (p.cut_st == 1u) hipLaunchKernelGGL(k_merge_tile_runs_shared, dim3(grid_for((uint64_t)p.n_tiles_ub * p.k)), dim3(256), 0, s, ms, p);
    else hipLaunchKernelGGL(k_merge_tile_runs_few, dim3(grid_for((uint64_t)p.n_tiles_ub * p.k)), dim3(256), 0, s, ms, p);
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
hipError_t launch_count_nonzero(const uint32_t *v, uint64_t n, uint64_t *out, hipStream_t s, const uint32_t *err, const uint32_t *n_tiles, uint64_t *mail) {
    if (n == 0 && !mail) return hipSuccess;
    unsigned g = grid_for(n ? n : 1);
    if (g > 128) g = 128;
    hipLaunchKernelGGL(k_count_nonzero, dim3(g), dim3(256), 0, s, v, n, out, err, n_tiles, mail);
    return hipGetLastError();
}

}  // namespace ii2
