// merge_small.hip — the common Shard.Merge in ONE launch (gfx950, wave64).
//
// Every Shard.Put writes a direct segment with one posting per term (reference shard.go:33-67) and Shard.Merge picks the
// smallest segments first (shard.go:135-146), so the usual merge is tiny: a handful of segments of a few terms.  Through
// the general entry points such a merge is dozens of launches, several allocations and several host round trips (term
// alignment, k aligned views, plan + tile + pack kernels, offset download, empty-term compaction, encode): 0.5 - 1.4 ms
// for a few hundred bytes of data.  Here one workgroup does all of it:
//   1. term alignment: every term of the k dictionaries finds its place in their k-way merge by bisection in the other
//      dictionaries (file.CompareTermValues = bytes.Compare order, file/types.go:24-26; same ranking as align.hip, in LDS);
//      equal terms collapse into one union term (the merging iterator's fold, shard.go:253-278);
//   2. the lists are decoded into LDS in (term, segment) order;
//   3. same-term union (file.MergeTermValues, file/types.go:14-22): every posting ranks itself inside its term — its
//      index in its own list plus one bisection per other list of the term, ties by segment — duplicates and removed ids
//      (slices.BinarySearch in the sorted removed list, shard.go:181-190) are flagged and the survivors compacted;
//   4. terms without survivors are dropped (shard.go:192-194);
//   5. the result is DV1-encoded (the w.Append of shard.go:207) into a segment that is ready to use.
// One upload (dictionaries + removed list, one pinned block), one launch, one download (sizes, kept terms, host mirrors).
#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t SM_T = II2_SMALL_MERGE_TERMS;          // input terms (= input lists) in all
constexpr uint32_t SM_P = II2_SMALL_MERGE_POSTINGS;       // input postings in all
constexpr uint32_t SM_R = II2_SMALL_MERGE_REMOVED;        // removed ids
constexpr uint32_t SM_TB = 16384;                         // bytes of all terms
constexpr uint32_t SM_THREADS = 512;
constexpr uint32_t SM_EPT = SM_P / SM_THREADS;            // postings per thread (16)
constexpr uint32_t SM_NB = SM_T + SM_P / II2_DV1_BLOCK;   // output blocks at most (every list has one, full blocks add to that)

// what the host uploads in one block
struct SmallIn {
    uint64_t key[SM_T];          // first 8 bytes of every term, big-endian, zero padded
    uint32_t len[SM_T];          // term lengths
    uint32_t toff[SM_T + 1];     // byte offsets of the terms in tbytes
    uint32_t dfirst[MAX_LISTS + 1];   // first term of each dictionary
    uint32_t lfirst[MAX_LISTS];       // the segment's list that holds a dictionary's first term (a range-restricted read: a slice of the lists)
    uint32_t removed[SM_R];      // ascending
    uint8_t tbytes[SM_TB];
};
// what the kernel hands back in one block
struct SmallOut {
    uint64_t n_terms_out, n_out, n_blocks, n_bytes, n_in, error;
    uint32_t kept[SM_T];         // per output list: an input term equal to its term
    uint32_t blk_off[SM_T + 1];  // host mirror of the new segment's list table
    uint32_t spans[3 * SM_T];    // per output list {first doc, first doc of the last block, last doc}
    uint32_t values[SM_P];       // raw mode (ii2_read_small): the merged ids, list after list (blk_off then holds posting offsets)
};
struct SmallSeg { const uint32_t *blk_off; const ii2_skip *skip; const uint8_t *payload; const uint32_t *cnt; };
struct SmallParams {
    SmallSeg seg[MAX_LISTS];
    const SmallIn *in;
    SmallOut *out;
    uint32_t k, n_terms, n_removed, long_terms;
    uint32_t raw;                 // 1: hand the merged lists back as they are (every union term, empty ones too), no segment
    // the new segment's arrays (sized for the limits)
    uint32_t *o_blk_off; ii2_skip *o_skip; uint8_t *o_payload; uint32_t *o_cnt; uint32_t *o_last; uint32_t *o_blk_list;
};

struct __align__(16) SmallSmem {
    uint32_t E[SM_P];                    // the postings: list by list, then term by term in id order, then the survivors
    uint64_t key[SM_T];
    uint32_t len[SM_T];
    uint16_t dict[SM_T];                 // dictionary (= segment) of every input term
    uint16_t place[SM_T];                // its place in the k-way merge of the dictionaries
    uint16_t order[SM_T];                // inverse: the term at each place
    uint16_t uof[SM_T + 1];              // union term of each place
    uint16_t uhead[SM_T + 1];            // first place of every union term (+ end)
    uint16_t lbase[SM_T + 1];            // first posting of the list at each place (term-major order)
    uint16_t tcnt[SM_T + 1];             // survivors per union term, then their exclusive prefix over the kept terms
    uint16_t keptu[SM_T + 1];            // union term of every output list
    uint32_t dead[SM_P / 32 + 1], deadpre[SM_P / 32 + 2];
    uint32_t wsum[SM_THREADS / 64];
    uint32_t dfirst[MAX_LISTS + 1], lfirst[MAX_LISTS];
    uint32_t n, nu, err;
};

__device__ __forceinline__ uint32_t sm_scan(uint32_t v, uint32_t *wsum, uint32_t *tot) {      // exclusive scan over the workgroup
    const int l = lane_id(), wv = (int)threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();
    if (l == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, t = 0;
    for (int w = 0; w < (int)(SM_THREADS / 64); w++) { if (w < wv) pre += wsum[w]; t += wsum[w]; }
    *tot = t;
    return pre + incl - v;
}

__global__ __launch_bounds__(SM_THREADS) void k_merge_small(SmallParams p) {
    __shared__ SmallSmem sm;
    const uint32_t tid = threadIdx.x, wv = tid >> 6;
    const SmallIn *in = p.in;
    const uint32_t nt = p.n_terms, k = p.k;
    // ---- 1. the dictionaries: keys and lengths into LDS, then every term's place in their k-way merge
    if (tid <= k) sm.dfirst[tid] = in->dfirst[tid];
    if (tid < k) sm.lfirst[tid] = in->lfirst[tid];
    if (tid == 0) { sm.err = 0u; sm.n = 0u; }
    __syncthreads();
    uint32_t my_dict = 0;
    if (tid < nt) {
        sm.key[tid] = in->key[tid];
        sm.len[tid] = in->len[tid];
        uint32_t lo = 0, hi = k;                  // dfirst[lo] <= tid < dfirst[hi]
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (sm.dfirst[mid] <= tid) lo = mid; else hi = mid; }
        my_dict = lo;
        sm.dict[tid] = (uint16_t)lo;
    }
    for (uint32_t i = tid; i < SM_P / 32 + 1; i += SM_THREADS) sm.dead[i] = 0u;
    __syncthreads();
    // bytes.Compare of term a with term b (-1 / 0 / 1)
    auto tcmp = [&](uint32_t a, uint32_t b) -> int {
        const uint64_t ka = sm.key[a], kb = sm.key[b];
        if (ka != kb) return ka < kb ? -1 : 1;
        const uint32_t la = sm.len[a], lb = sm.len[b];
        if (p.long_terms) {
            const uint8_t *pa = in->tbytes + in->toff[a], *pb = in->tbytes + in->toff[b];
            const uint32_t m = la < lb ? la : lb;
            for (uint32_t j = 8; j < m; j++) if (pa[j] != pb[j]) return pa[j] < pb[j] ? -1 : 1;
        }
        return la == lb ? 0 : (la < lb ? -1 : 1);
    };
    bool is_head = true;
    uint32_t my_place = 0;
    if (tid < nt) {
        my_place = tid - sm.dfirst[my_dict];          // my index in my own dictionary + the terms before me in every other one
        for (uint32_t sp = 0; sp < k; sp++) {
            if (sp == my_dict) continue;
            uint32_t lo = sm.dfirst[sp], hi = sm.dfirst[sp + 1];
            const uint32_t base = lo;
            const bool upper = sp < my_dict;          // earlier dictionaries: count the terms <= mine; later ones: the terms < mine
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                const int c = tcmp(mid, tid);
                if (c < 0 || (upper && c == 0)) lo = mid + 1u; else hi = mid;
            }
            if (upper && lo > base && tcmp(lo - 1u, tid) == 0) is_head = false;   // an earlier dictionary holds my term
            my_place += lo - base;
        }
        sm.place[tid] = (uint16_t)my_place;
        sm.order[my_place] = (uint16_t)tid;
    }
    __syncthreads();
    // union terms: heads of runs of equal terms, numbered by prefix sum over the places
    {
        // (a thread knows whether ITS term is a head; the flag has to sit at the term's place)
        if (tid < nt) sm.uof[my_place] = is_head ? 1u : 0u;
        __syncthreads();
        const uint32_t hf = tid < nt ? sm.uof[tid] : 0u;
        uint32_t tot;
        const uint32_t ex = sm_scan(hf, sm.wsum, &tot);
        __syncthreads();
        if (tid < nt) {
            sm.uof[tid] = (uint16_t)(ex + hf - 1u);             // the place's union term
            if (hf) sm.uhead[ex] = (uint16_t)tid;               // its first place
        }
        if (tid == 0) { sm.nu = tot; sm.uhead[tot] = (uint16_t)nt; }
    }
    __syncthreads();
    const uint32_t nu = sm.nu;
    // ---- 2. the lists, in (term, segment) order = order of the places: sizes, then decode
    uint32_t my_cnt = 0, my_b0 = 0, my_nb = 0, my_s = 0, my_li = 0;
    if (tid < nt) {               // thread = place
        const uint32_t g = sm.order[tid];
        my_s = sm.dict[g];
        my_li = g - sm.dfirst[my_s] + sm.lfirst[my_s];
        const SmallSeg sg = p.seg[my_s];
        my_b0 = sg.blk_off[my_li];
        my_nb = sg.blk_off[my_li + 1u] - my_b0;
        my_cnt = my_nb ? sg.cnt[my_li] : 0u;
    }
    {
        uint32_t tot;
        const uint32_t ex = sm_scan(my_cnt, sm.wsum, &tot);
        if (tid < nt) sm.lbase[tid] = (uint16_t)(ex < SM_P ? ex : SM_P);
        if (tid == 0) { sm.lbase[nt] = (uint16_t)(tot < SM_P ? tot : SM_P); sm.n = tot; if (tot > SM_P) sm.err = 1u; }
    }
    __syncthreads();
    if (sm.err) { if (tid == 0) p.out->error = 1u; return; }
    const uint32_t n = sm.n;
    // single-posting lists (what Shard.Put writes) by their own thread; the others one wave per list, block by block
    if (tid < nt && my_cnt == 1u) sm.E[sm.lbase[tid]] = p.seg[my_s].skip[my_b0].first_doc;
    for (uint32_t pl = wv; pl < nt; pl += SM_THREADS / 64u) {
        const uint32_t base = sm.lbase[pl], cnt = sm.lbase[pl + 1u] - base;
        if (cnt <= 1u) continue;                      // (decided in LDS: no memory round trip for the lists done above)
        const uint32_t g = sm.order[pl], s = sm.dict[g], li = g - sm.dfirst[s] + sm.lfirst[s];
        const SmallSeg sg = p.seg[s];
        const uint32_t b0 = sg.blk_off[li], nb = sg.blk_off[li + 1u] - b0;
        for (uint32_t j = 0; j < nb; j++) {
            const ii2_skip e0 = sg.skip[b0 + j];
            const uint32_t q1 = sg.skip[b0 + j + 1u].byte_off;
            const uint32_t room = cnt - j * II2_DV1_BLOCK;        // postings the list still has from this block on
            decode_block_wave(sg.payload, e0.byte_off, q1, e0.first_doc, [&](uint32_t ix, uint32_t id) {
                if (ix < room && ix < II2_DV1_BLOCK) sm.E[base + j * II2_DV1_BLOCK + ix] = id;
            });
        }
    }
    __syncthreads();
    // ---- 3. every posting ranks itself inside its term; duplicates and removed ids are flagged
    uint32_t ev[SM_EPT], ed[SM_EPT];            // my postings and where they go (bit 31: dead)
#pragma unroll
    for (uint32_t j = 0; j < SM_EPT; j++) {
        const uint32_t e = tid + j * SM_THREADS;
        ev[j] = 0; ed[j] = 0xFFFFFFFFu;
        if (e < n) {
            uint32_t lo = 0, hi = nt;               // the list (place) that holds posting e: lbase[lo] <= e < lbase[hi]
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)sm.lbase[mid] <= e) lo = mid; else hi = mid; }
            const uint32_t pl = lo, u = sm.uof[pl];
            const uint32_t v = sm.E[e];
            uint32_t r = e - sm.lbase[pl];
            bool dup = false;
            for (uint32_t q = sm.uhead[u]; q < (uint32_t)sm.uhead[u + 1u]; q++) {
                if (q == pl) continue;
                uint32_t a = sm.lbase[q], b = sm.lbase[q + 1u];
                const uint32_t a0 = a;
                const bool upper = q < pl;            // earlier lists of the term: count ids <= mine; later: ids < mine
                while (a < b) { const uint32_t mid = a + ((b - a) >> 1); const uint32_t y = sm.E[mid]; if (y < v || (upper && y == v)) a = mid + 1u; else b = mid; }
                if (upper && a > a0 && sm.E[a - 1u] == v) dup = true;
                r += a - a0;
            }
            bool gone = dup;
            if (!gone && p.n_removed) {             // slices.BinarySearch(removedValues, v)
                uint32_t a = 0, b = p.n_removed;
                while (a < b) { const uint32_t mid = a + ((b - a) >> 1); if (in->removed[mid] < v) a = mid + 1u; else b = mid; }
                gone = a < p.n_removed && in->removed[a] == v;
            }
            const uint32_t dst = (uint32_t)sm.lbase[sm.uhead[u]] + r;
            if (gone) atomicOr(&sm.dead[dst >> 5], 1u << (dst & 31u));
            ev[j] = v;
            ed[j] = dst | (gone ? 0x80000000u : 0u);
        }
    }
    __syncthreads();
    // dead ids before every 32 positions
    uint32_t n_dead;
    {
        const uint32_t nw = (n + 31u) >> 5;
        const uint32_t x = tid < nw ? (uint32_t)__popc(sm.dead[tid]) : 0u;
        const uint32_t ex = sm_scan(x, sm.wsum, &n_dead);
        if (tid <= nw) sm.deadpre[tid] = ex;
    }
    __syncthreads();
    auto dead_before = [&](uint32_t x) -> uint32_t { return sm.deadpre[x >> 5] + (uint32_t)__popc(sm.dead[x >> 5] & ((1u << (x & 31u)) - 1u)); };
    const uint32_t n_out = n - n_dead;
    // survivors per union term; the kept terms
    {
        uint32_t c = 0;
        if (tid < nu) {
            const uint32_t a = sm.lbase[sm.uhead[tid]], b = sm.lbase[sm.uhead[tid + 1u]];
            c = (b - a) - (dead_before(b) - dead_before(a));
        }
        const bool keep = tid < nu && (c != 0u || p.raw);       // (a read hands every term back, a merge drops the emptied ones)
        uint32_t n_kept;
        const uint32_t kx = sm_scan(keep ? 1u : 0u, sm.wsum, &n_kept);
        __syncthreads();
        if (keep) {
            sm.keptu[kx] = (uint16_t)tid;
            sm.tcnt[kx] = (uint16_t)c;                               // (<= 8192)
            p.out->kept[kx] = sm.order[sm.uhead[tid]];               // the term of the union term's first place stands for it
        }
        if (tid == 0) sm.nu = n_kept;            // from here on: output lists
    }
    // the survivors, compacted in place (all reads are done: they sit in registers)
#pragma unroll
    for (uint32_t j = 0; j < SM_EPT; j++)
        if (!(ed[j] >> 31)) sm.E[ed[j] - dead_before(ed[j])] = ev[j];
    __syncthreads();
    const uint32_t T2 = sm.nu;
    if (p.raw) {                 // ---- a read: the merged lists as they are
        SmallOut *o = p.out;
        for (uint32_t e = tid; e < n_out; e += SM_THREADS) o->values[e] = sm.E[e];
        const uint32_t c_r = tid < T2 ? sm.tcnt[tid] : 0u;
        uint32_t tot_r;
        const uint32_t off_r = sm_scan(c_r, sm.wsum, &tot_r);
        if (tid < T2) o->blk_off[tid] = off_r;
        if (tid == 0) { o->blk_off[T2] = tot_r; o->n_terms_out = T2; o->n_out = n_out; o->n_blocks = 0; o->n_bytes = 0; o->n_in = n; o->error = 0; }
        return;
    }
    // ---- 5. encode.  Output list j: c_j postings from fo_j on; its blocks; gaps as varints
    uint32_t c_j = tid < T2 ? sm.tcnt[tid] : 0u;
    uint32_t fo_tot, bo_tot;
    const uint32_t fo_j = sm_scan(c_j, sm.wsum, &fo_tot);                        // first survivor of list j
    const uint32_t nb_j = (c_j + II2_DV1_BLOCK - 1u) / II2_DV1_BLOCK;
    const uint32_t bo_j = sm_scan(nb_j, sm.wsum, &bo_tot);                       // first block of list j
    __syncthreads();
    // (tcnt / lbase are free now: list starts and block starts of the output lists)
    if (tid < T2) { sm.lbase[tid] = (uint16_t)fo_j; sm.uhead[tid] = (uint16_t)bo_j; }
    if (tid == 0) { sm.lbase[T2] = (uint16_t)(fo_tot < SM_P ? fo_tot : SM_P); sm.uhead[T2] = (uint16_t)bo_tot; }
    __syncthreads();
    // bytes of every posting's gap (0 for the first posting of a block), their prefix = byte offsets
    uint32_t glen[SM_EPT], gap[SM_EPT], mylen = 0;
    const uint32_t e0 = tid * SM_EPT;              // consecutive postings per thread here
#pragma unroll
    for (uint32_t j = 0; j < SM_EPT; j++) {
        const uint32_t e = e0 + j;
        glen[j] = 0; gap[j] = 0;
        if (e < n_out) {
            uint32_t lo = 0, hi = T2;               // list of survivor e
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)sm.lbase[mid] <= e) lo = mid; else hi = mid; }
            const uint32_t i = e - sm.lbase[lo];
            if (i & (II2_DV1_BLOCK - 1u)) { gap[j] = sm.E[e] - sm.E[e - 1u]; glen[j] = varint_len(gap[j]); }
            else {                                   // first posting of block (i / 256) of list lo
                const uint32_t b = (uint32_t)sm.uhead[lo] + i / II2_DV1_BLOCK;
                p.o_skip[b].first_doc = sm.E[e];
                p.o_blk_list[b] = lo;
                glen[j] = 0x80000000u | b;           // (marks the block start; its byte offset is written below)
            }
            if (!(glen[j] >> 31)) mylen += glen[j];
        }
    }
    uint32_t n_bytes;
    uint32_t q = sm_scan(mylen, sm.wsum, &n_bytes);
#pragma unroll
    for (uint32_t j = 0; j < SM_EPT; j++) {
        const uint32_t e = e0 + j;
        if (e >= n_out) break;
        if (glen[j] >> 31) { p.o_skip[glen[j] & 0x7FFFFFFFu].byte_off = q; continue; }
        uint32_t v = gap[j];
        while (v >= 0x80u) { p.o_payload[q++] = (uint8_t)(v | 0x80u); v >>= 7; }
        p.o_payload[q++] = (uint8_t)v;
    }
    // list tables, closing entries, host mirrors, sizes
    SmallOut *o = p.out;
    if (tid < T2) {
        const uint32_t last = sm.E[fo_j + c_j - 1u], lastblk_first = sm.E[fo_j + ((c_j - 1u) & ~(II2_DV1_BLOCK - 1u))];
        p.o_blk_off[tid] = bo_j;
        p.o_cnt[tid] = c_j;
        p.o_last[tid] = last;
        o->blk_off[tid] = bo_j;
        o->spans[3u * tid] = sm.E[fo_j];
        o->spans[3u * tid + 1u] = lastblk_first;
        o->spans[3u * tid + 2u] = last;
    }
    if (tid == 0) {
        p.o_blk_off[T2] = bo_tot;
        o->blk_off[T2] = bo_tot;
        p.o_skip[bo_tot].first_doc = n_out ? sm.E[n_out - 1u] : 0u;
        p.o_skip[bo_tot].byte_off = n_bytes;
        for (uint32_t z = 0; z < 16u; z++) p.o_payload[n_bytes + z] = 0;
        o->n_terms_out = T2; o->n_out = n_out; o->n_blocks = bo_tot; o->n_bytes = n_bytes; o->n_in = n; o->error = 0;
    }
}

}  // namespace ii2

using namespace ii2;

#define HIP_TRY(ctx, expr)                                                                 \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return II2_EHIP;                                                               \
        }                                                                                  \
    } while (0)

static int fail(ii2_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}
static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Both entry points: the merge (out: a new segment; emptied terms dropped) and the read (raw: post_off / values on the host,
// every union term).  list_first (may be NULL): the list of segment s that holds its dictionary's first term.
static int small_core(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const uint8_t *term_bytes, const uint64_t *term_off,
                      const uint64_t *seg_first, const uint64_t *list_first, const uint32_t *removed, uint64_t n_removed, bool raw,
                      ii2_seg **out, uint64_t *kept, uint64_t *n_kept, ii2_merge_stats *stats, uint64_t *post_off, uint32_t *values, uint64_t cap) {
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (out) *out = nullptr;
    *n_kept = 0;
    const uint64_t n = seg_first[k];
    if (seg_first[0] != 0) return fail(ctx, II2_EINVAL, "ii2_merge_small: seg_first[0] must be 0");
    if (n && term_off[n] && !term_bytes) return fail(ctx, II2_EINVAL, "ii2_merge_small: term_bytes is NULL");
    if (n > SM_T || n_removed > SM_R) return fail(ctx, II2_ERANGE, "ii2_merge_small: too many terms or removed ids for the one-launch merge");
    uint64_t n_post = 0;
    for (uint32_t s = 0; s < k; s++) {
        if (!segs[s] || segs[s]->device != ctx->device) return fail(ctx, II2_EINVAL, "ii2_merge_small: a segment is NULL or lives on another device");
        const uint64_t lf = list_first ? list_first[s] : 0;
        if (seg_first[s + 1] < seg_first[s] || lf > segs[s]->n_lists || seg_first[s + 1] - seg_first[s] > segs[s]->n_lists - lf ||
            (!list_first && segs[s]->n_lists != seg_first[s + 1] - seg_first[s]))
            return fail(ctx, II2_EINVAL, "ii2_merge_small: every segment must have one list per term of its dictionary");
        n_post += segs[s]->n_postings;
    }
    if (n_post > SM_P) return fail(ctx, II2_ERANGE, "ii2_merge_small: too many postings for the one-launch merge");
    if (n && (term_off[0] != 0 || term_off[n] > SM_TB)) return fail(ctx, term_off[0] ? II2_EINVAL : II2_ERANGE, "ii2_merge_small: term bytes out of range");
    // staging blocks (made once per context): the upload, the download
    if (!ctx->h_small_in) {
        if (hipHostMalloc((void **)&ctx->h_small_in, sizeof(SmallIn)) != hipSuccess || hipHostMalloc((void **)&ctx->h_small_out, sizeof(SmallOut)) != hipSuccess ||
            ii2::dm_malloc_retry((void **)&ctx->d_small_in, sizeof(SmallIn)) != hipSuccess || ii2::dm_malloc_retry((void **)&ctx->d_small_out, sizeof(SmallOut)) != hipSuccess)
            return fail(ctx, II2_ENOMEM, "ii2_merge_small: staging allocation failed");
    }
    SmallIn *hi = (SmallIn *)ctx->h_small_in;
    uint32_t long_terms = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (term_off[i + 1] < term_off[i]) return fail(ctx, II2_EINVAL, "ii2_merge_small: term_off must be non-decreasing");
        const uint64_t b = term_off[i], len = term_off[i + 1] - b;
        uint64_t key = 0;
        for (int j = 0; j < 8; j++) { key <<= 8; if ((uint64_t)j < len) key |= term_bytes[b + j]; }
        hi->key[i] = key;
        hi->len[i] = (uint32_t)len;
        hi->toff[i] = (uint32_t)b;
        long_terms |= len > 8 ? 1u : 0u;
    }
    hi->toff[n] = n ? (uint32_t)term_off[n] : 0u;
    for (uint32_t s = 0; s <= k; s++) hi->dfirst[s] = (uint32_t)seg_first[s];
    for (uint32_t s = 0; s < k; s++) hi->lfirst[s] = list_first ? (uint32_t)list_first[s] : 0u;
    if (n && term_off[n]) std::memcpy(hi->tbytes, term_bytes, term_off[n]);
    if (n_removed) {
        std::memcpy(hi->removed, removed, n_removed * sizeof(uint32_t));
        std::sort(hi->removed, hi->removed + n_removed);            // RemovedLists.Values() is sorted (removed_list.go:44-54); any order is accepted here
    }
    hipStream_t st = ctx->stream;
    // only what is used travels: everything up to the end of the removed list, then the term bytes
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_small_in, hi, offsetof(SmallIn, removed) + n_removed * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (long_terms) HIP_TRY(ctx, hipMemcpyAsync((uint8_t *)ctx->d_small_in + offsetof(SmallIn, tbytes), hi->tbytes, term_off[n], hipMemcpyHostToDevice, st));
    // the new segment's arrays: one allocation sized for the limits
    const size_t o_blk = 0, o_skip = o_blk + up256((SM_T + 1) * sizeof(uint32_t)), o_pay = o_skip + up256((SM_NB + 1) * sizeof(ii2_skip)),
                 o_cnt = o_pay + up256((size_t)SM_P * 5 + 16), o_last = o_cnt + up256((SM_T + 1) * sizeof(uint32_t)), o_bl = o_last + up256((SM_T + 1) * sizeof(uint32_t)),
                 slab_bytes = o_bl + up256((SM_NB + 1) * sizeof(uint32_t));
    uint8_t *slab = nullptr;
    if (!raw && dm_alloc((void **)&slab, slab_bytes) != hipSuccess) return fail(ctx, II2_ENOMEM, "ii2_merge_small: segment allocation failed");
    SmallParams p;
    std::memset(&p, 0, sizeof p);
    for (uint32_t s = 0; s < k; s++) p.seg[s] = SmallSeg{segs[s]->d_blk_off, segs[s]->d_skip, segs[s]->d_payload, segs[s]->d_cnt};
    p.in = (const SmallIn *)ctx->d_small_in;
    p.out = (SmallOut *)ctx->d_small_out;
    p.k = k; p.n_terms = (uint32_t)n; p.n_removed = (uint32_t)n_removed; p.long_terms = long_terms;
    p.raw = raw ? 1u : 0u;
    p.o_blk_off = (uint32_t *)(slab + o_blk); p.o_skip = (ii2_skip *)(slab + o_skip); p.o_payload = slab + o_pay;
    p.o_cnt = (uint32_t *)(slab + o_cnt); p.o_last = (uint32_t *)(slab + o_last); p.o_blk_list = (uint32_t *)(slab + o_bl);
    hipLaunchKernelGGL(k_merge_small, dim3(1), dim3(SM_THREADS), 0, st, p);
    SmallOut *ho = (SmallOut *)ctx->h_small_out;
    hipError_t e = hipGetLastError();
    // sizes first (they say how much of the rest matters) would be a second round trip: the whole block is 12 KB, take it in one
    // (a read also takes the merged ids: at most as many as the lists hold)
    if (e == hipSuccess) e = hipMemcpyAsync(ho, ctx->d_small_out, offsetof(SmallOut, values) + (raw ? n_post * sizeof(uint32_t) : 0), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); dm_free(slab); ctx->err = std::string("ii2_merge_small: ") + hipGetErrorString(e); return II2_EHIP; }
    if (ho->error) { dm_free(slab); return fail(ctx, II2_ERANGE, "ii2_merge_small: the lists hold more postings than the one-launch merge takes"); }
    if (stats) { stats->n_in = ho->n_in; stats->n_out = ho->n_out; stats->n_terms_out = ho->n_terms_out; stats->n_tiles = 1; }
    if (raw && ho->n_out > cap) return fail(ctx, II2_ECAPACITY, "ii2_read_small: the values buffer is too small; nothing was written");
    *n_kept = ho->n_terms_out;
    for (uint64_t j = 0; j < ho->n_terms_out; j++) kept[j] = ho->kept[j];
    if (raw) {
        for (uint64_t j = 0; j <= ho->n_terms_out; j++) post_off[j] = ho->blk_off[j];
        if (ho->n_out) std::memcpy(values, ho->values, ho->n_out * sizeof(uint32_t));
        return II2_OK;
    }
    if (ho->n_terms_out == 0) { dm_free(slab); return II2_OK; }        // shard.go:219-225: nothing survives, no segment is written
    ii2_seg *seg = new (std::nothrow) ii2_seg();
    if (!seg) { dm_free(slab); return II2_ENOMEM; }
    seg->device = ctx->device;
    seg->store = std::make_shared<ii2_seg_store>();
    seg->store->slab = slab;                     // owns every array of this segment
    seg->store->d_skip = nullptr;
    seg->store->d_payload = nullptr;
    seg->in_slab = true;
    seg->n_lists = ho->n_terms_out; seg->n_postings = ho->n_out; seg->n_blocks = ho->n_blocks; seg->n_bytes = ho->n_bytes;
    seg->d_blk_off = p.o_blk_off; seg->d_skip = p.o_skip; seg->d_payload = p.o_payload;
    seg->d_cnt = p.o_cnt; seg->d_last_doc = p.o_last; seg->d_blk_list = p.o_blk_list;
    seg->h_blk_off.assign(ho->blk_off, ho->blk_off + seg->n_lists + 1);
    seg->h_spans.assign(ho->spans, ho->spans + 3 * seg->n_lists);
    *out = seg;
    return II2_OK;
}

extern "C" int ii2_merge_small(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const uint8_t *term_bytes, const uint64_t *term_off,
                               const uint64_t *seg_first, const uint32_t *removed, uint64_t n_removed, ii2_seg **out, uint64_t *kept,
                               uint64_t *n_kept, ii2_merge_stats *stats) {
    if (!ctx || !segs || !term_off || !seg_first || !out || !kept || !n_kept || k == 0 || k > MAX_LISTS || (n_removed && !removed))
        return fail(ctx, II2_EINVAL, "ii2_merge_small: bad argument");
    return small_core(ctx, k, segs, term_bytes, term_off, seg_first, nullptr, removed, n_removed, false, out, kept, n_kept, stats, nullptr, nullptr, 0);
}

// The small Shard.Read (reference shard.go:72-75 -> makeIterator shard.go:253-278, a3 without the tombstone filter): the
// merged lists of k small segments, range-restricted by the caller's dictionaries, straight to host memory in one launch.
extern "C" int ii2_read_small(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const uint8_t *term_bytes, const uint64_t *term_off,
                              const uint64_t *seg_first, const uint64_t *list_first, uint64_t *rep, uint64_t *post_off, uint32_t *values,
                              uint64_t cap, uint64_t *n_union) {
    if (!ctx || !segs || !term_off || !seg_first || !rep || !post_off || !n_union || (cap && !values) || k == 0 || k > MAX_LISTS)
        return fail(ctx, II2_EINVAL, "ii2_read_small: bad argument");
    return small_core(ctx, k, segs, term_bytes, term_off, seg_first, list_first, nullptr, 0, true, nullptr, rep, n_union, nullptr, post_off, values, cap);
}
