// ops.cpp — segment merge and union entry points, and the host-buffer convenience calls
// (include/ii2.h).  Planning (which terms share a tile) runs on the device; the host only
// reads back three scalars per call.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "internal.h"

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
static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return ii2::dm_malloc_retry(&p, bytes ? bytes : 16); }
    template <class T> T *as() const { return (T *)p; }
};

int ii2_ws_reserve(ii2_ctx *ctx, size_t bytes);          // api.cpp

template <class T> static T *carve(uint8_t *&cursor, size_t count) {
    T *p = (T *)cursor;
    cursor += align_up(count * sizeof(T));
    return p;
}

// Core: k SegViews over n_terms aligned term slots -> d_out_off (u64[n_terms+1], may be null), d_out_values.
// blocks_ub / postings_ub: host-side upper bounds of the views' blocks and postings — every grid and every scratch
// array is sized from them, so the call enqueues all its kernels without a single host round trip and only reads three
// scalars back at the end.  Nothing is decoded ahead of the tile kernel: the plan works on the segments' per-list counts,
// first / last docs and skip tables.
static int merge_core(ii2_ctx *ctx, uint32_t k, const SegView *views, uint64_t n_terms, uint64_t blocks_ub, uint64_t postings_ub,
                      const ii2_tomb *tomb, uint64_t *d_out_off, uint32_t *d_out_values, uint64_t out_cap, ii2_merge_stats *stats) {
    hipStream_t st = ctx->stream;
    const uint64_t T = n_terms;
    if (T >= (1ull << 30) - 2) return fail(ctx, II2_ERANGE, "merge: 2^30 or more term slots in one call");
    if (blocks_ub >= (1ull << 31)) return fail(ctx, II2_ERANGE, "merge: too many input blocks");
    if (postings_ub >= (1ull << 32)) return fail(ctx, II2_ERANGE, "merge: 2^32 or more input postings in one call");
    if (tomb && tomb->device != ctx->device) return fail(ctx, II2_EINVAL, "tombstones live on another device");
    MergeParams p;
    std::memset(&p, 0, sizeof p);
    // the views travel through a pinned staging block (the call ends with a stream sync, so the next call may reuse it)
    if (!ctx->h_segs) {
        if (hipHostMalloc((void **)&ctx->h_segs, sizeof(MergeSegs)) != hipSuccess || ii2::dm_malloc_retry((void **)&ctx->d_segs, sizeof(MergeSegs)) != hipSuccess)
            return fail(ctx, II2_ENOMEM, "merge: staging allocation failed");
    }
    MergeSegs *hs = (MergeSegs *)ctx->h_segs;
    std::memset(hs, 0, sizeof *hs);
    for (uint32_t s = 0; s < k; s++) hs->segs[s] = views[s];
    hs->k = k;
    hs->n_terms = T;
    const MergeSegs *d_ms = (const MergeSegs *)ctx->d_segs;
    p.k = k;
    p.n_terms = T;
    p.tomb = tomb ? tomb->d_words : nullptr;
    p.tomb_summary = tomb ? tomb->d_summary : nullptr;
    p.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
    const uint32_t cap = MERGE_CAP;
    // batches: small terms are packed in term order; a batch ends when its weight passes a multiple of batch_q, so it holds
    // less than batch_q + small_max <= cap postings and, every term weighing at least wmin, at most MERGE_NT_MAX terms
    p.small_max = (cap * 5u) / 14u;         // 2560
    p.batch_q = cap - p.small_max;          // 4608
    p.wmin = (cap + MERGE_NT_MAX - 1u) / MERGE_NT_MAX;
    p.range_target = ctx->opt_merge_large_tile > 0 ? std::min<uint32_t>((uint32_t)ctx->opt_merge_large_tile, cap) : (cap / 20u) * 19u;
    p.bitmap_tiles = ctx->opt_merge_bitmap ? 1u : 0u;
    p.bitmap_sparsity = ctx->opt_merge_bitmap > 1 ? (uint32_t)std::min<int64_t>(ctx->opt_merge_bitmap, 4096) : 80u;
    // upper bound of the tile count (the exact one is computed on the device and stays there): a large term has more than
    // small_max postings and takes ceil(n / range_target) range tiles or, as a bitmap term with >= 1 posting per
    // `sparsity` docs, ceil(span / MERGE_BM_DOCS) <= n * sparsity / MERGE_BM_DOCS + 1 bitmap tiles; a batch ends when its
    // weight passes batch_q or a large term interrupts the run of small ones
    const uint64_t n_large_ub = postings_ub / ((uint64_t)p.small_max + 1u);
    const uint64_t tiles_ub64 = postings_ub / p.range_target + (postings_ub * p.bitmap_sparsity) / MERGE_BM_DOCS + 4 * n_large_ub +      // (+ 2 per large term: block-granular cuts, k_mp_terms)
                                (postings_ub + T * p.wmin) / p.batch_q + n_large_ub + 4;
    if (tiles_ub64 >= (1ull << 31) || tiles_ub64 * k >= (1ull << 32)) return fail(ctx, II2_ERANGE, "merge: too many tiles");
    p.n_tiles_ub = (uint32_t)tiles_ub64;

    hipEvent_t e0 = nullptr, e1 = nullptr;          // option profile.events: the pair brackets the WHOLE call, first launch to last
    if (ii2_profile_pair(ctx, &e0, &e1)) (void)hipEventRecord(e0, st);
    auto grow = [&](uint8_t *&buf, size_t &bcap, size_t bytes) -> bool {
        if (bytes <= bcap) return true;
        (void)hipStreamSynchronize(st);
        if (buf) (void)hipFree(buf);
        buf = nullptr;
        bcap = 0;
        const size_t want = align_up(bytes + bytes / 8, 1 << 20);
        if (ii2::dm_malloc_retry((void **)&buf, want) != hipSuccess) return false;
        bcap = want;
        return true;
    };
    const size_t n1 = (size_t)T + 1;
    const size_t scan_b = scan_temp_bytes(n1 + 1);
    // ws: per-term plan arrays
    size_t need = 11 * align_up(n1 * sizeof(uint32_t)) + 2 * align_up(n1 * sizeof(uint64_t)) + 3 * scan_b + 4096;
    int rc = ii2_ws_reserve(ctx, need);
    if (rc) return rc;
    uint8_t *cur = ctx->ws;
    p.tn = carve<uint32_t>(cur, n1);
    p.tmin = carve<uint32_t>(cur, n1);
    p.tmax = carve<uint32_t>(cur, n1);
    p.tinfo = carve<uint32_t>(cur, n1);
    p.weight = carve<uint32_t>(cur, n1);
    p.ntl = carve<uint32_t>(cur, n1);
    uint32_t *d_head = carve<uint32_t>(cur, n1);
    uint32_t *d_hpre = carve<uint32_t>(cur, n1);
    uint32_t *d_lpre = carve<uint32_t>(cur, n1);
    uint32_t *d_tt = carve<uint32_t>(cur, n1);
    uint32_t *d_cnt = carve<uint32_t>(cur, n1);
    uint64_t *d_wpre = carve<uint64_t>(cur, n1);
    uint64_t *d_npre = carve<uint64_t>(cur, n1);
    void *d_scan = cur;
    // aux: per-tile arrays, the large terms' bump allocators and the parking array
    const size_t nt1 = (size_t)p.n_tiles_ub + 1;
    const size_t scan_t = scan_temp_bytes(nt1);
    const size_t aux_need = 256 + align_up(nt1 * sizeof(uint32_t)) + align_up(n1 * sizeof(uint32_t)) + align_up(nt1 * 16) + 2 * align_up(nt1 * sizeof(uint64_t)) + scan_t +
                            align_up(nt1 * k * 16) + align_up(nt1 * k * sizeof(uint2)) + align_up((postings_ub + 64) * sizeof(uint32_t)) + 4096;
    if (!grow(ctx->aux, ctx->aux_cap, aux_need)) return fail(ctx, II2_ENOMEM, "merge scratch allocation failed");
    uint8_t *ac = ctx->aux;
    p.sync = (MergeSync *)carve<uint8_t>(ac, sizeof(MergeSync));      // (first in the allocation: zeroed together with the two arrays after it by one memset)
    p.tile_count = carve<uint32_t>(ac, nt1);
    p.term_alloc = carve<uint32_t>(ac, n1);
    const size_t zero_bytes = (size_t)(ac - ctx->aux);
    p.tile_slot = (unsigned long long *)carve<uint64_t>(ac, nt1);
    uint64_t *d_tile_off = carve<uint64_t>(ac, nt1);
    void *d_scan_t = carve<uint8_t>(ac, scan_t);
    p.desc = (uint4 *)carve<uint8_t>(ac, nt1 * 16);
    p.cut0 = (uint4 *)carve<uint8_t>(ac, nt1 * k * 16);
    p.cut1 = carve<uint2>(ac, nt1 * k);
    if (k >= 32u) { p.cut_ss = p.n_tiles_ub; p.cut_st = 1u; }      // many lists: bounds share blocks, the plan walks a block once for all of them (merge.hip)
    else { p.cut_ss = 1u; p.cut_st = k; }
    p.tmp = carve<uint32_t>(ac, postings_ub + 64);

    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_segs, hs, sizeof(MergeSegs), hipMemcpyHostToDevice, st));
    // ---- plan (device): per-term counts and doc ranges, batches, tiles, the blocks every range tile has to decode ----
    HIP_TRY(ctx, launch_merge_plan_terms(d_ms, p, st));
    HIP_TRY(ctx, scan3_excl(d_scan, 3 * scan_b, p.weight, d_wpre, p.tn, d_npre, p.ntl, d_lpre, n1, st));      // (three scans, the launches of one)
    HIP_TRY(ctx, launch_merge_heads(p, d_wpre, d_head, st));
    HIP_TRY(ctx, scan_excl_u32(d_scan, scan_b, d_head, d_hpre, n1, st));
    HIP_TRY(ctx, launch_merge_term_tile(p, d_head, d_hpre, d_lpre, d_tt, st));
    p.term_tile = d_tt;
    p.n_tiles_dev = d_tt + T;                   // term_tile[T] = number of tiles
    p.npre = d_npre;
    p.pad0 = (uint32_t)ctx->opt_merge_skip;
    p.spin_limit = ctx->opt_merge_spin > 0 ? (uint32_t)std::min<int64_t>(ctx->opt_merge_spin, 0x7FFFFFFF) : 0u;
    HIP_TRY(ctx, launch_merge_tile_desc(d_ms, p, st));
    HIP_TRY(ctx, launch_merge_tile_runs(d_ms, p, st));
    p.out_counts = d_cnt;
    p.out_values = d_out_values;
    p.out_cap = out_cap;
    p.d_total = ctx->d_mail;                    // [0] total, [2] surviving terms
    p.debug = nullptr;
    if (ctx->opt_debug_stamps) {
        if (!ctx->d_debug && ii2::dm_malloc_retry((void **)&ctx->d_debug, (size_t)2048 * 8 * sizeof(unsigned long long)) != hipSuccess)
            return fail(ctx, II2_ENOMEM, "debug buffer allocation failed");
        p.debug = ctx->d_debug;
    }
    // 2 workgroups of ~75 KB LDS per CU.  When the caller's buffer is known to hold any result (out_cap >= all input postings)
    // the tiles put their survivors in place themselves (ticket order, one scanner workgroup: merge.hip) - every posting
    // crosses HBM once on its way in and once on its way out.  Otherwise (the fit is only known at the end, and the call is
    // all-or-nothing) the tiles park and a packing pass follows the scan of their counts.
    p.tile_off = d_tile_off;
    p.direct = (postings_ub <= out_cap && ctx->opt_merge_direct) ? 1u : 0u;
    // one launch clears the result words, the per-term counts, the tickets / frontier / tile counts (tiles past the real count
    // contribute 0) / bump allocators and - direct placement - sets the tiles' offsets to all ones: "not known yet"
    HIP_TRY(ctx, launch_merge_init(ctx->d_mail, 4, d_cnt, n1, ctx->aux, zero_bytes, d_tile_off, p.direct ? nt1 : 0, st));
    if (p.direct) {
        // its workers wait for its scanner workgroup: never beside a look-back kernel (api.cpp).  Beside another tile kernel it is
        // safe, and worth it for small merges (a few tiles per workgroup: one kernel's tail hides behind the other's body - the
        // chunks of a rank's term range, 3 contexts: 2.5 - 3.2 ms a step against 2.8 - 3.7); big ones only get in each other's way
        // (three chunks of C4 on one GPU: 18.1 ms side by side, 16.1 ms one after the other).
        const bool alone = postings_ub > (uint64_t)(ctx->opt_merge_alone > 0 ? ctx->opt_merge_alone : 64) << 20;      // (option merge.alone: the threshold in Mi postings)
        if (int rcq = ii2_lookback_launch(ctx, alone, [&] { return launch_merge_tiles(d_ms, p, (uint32_t)ctx->cu_count * (1024u / MERGE_THREADS), st); })) return rcq;
    } else {
        HIP_TRY(ctx, launch_merge_tiles(d_ms, p, (uint32_t)ctx->cu_count * (1024u / MERGE_THREADS), st));
    }
    if (!p.direct) {
        HIP_TRY(ctx, scan_excl_u32_to_u64(d_scan_t, scan_t, p.tile_count, d_tile_off, nt1, st));
        HIP_TRY(ctx, launch_merge_pack(p, d_tile_off, st));
    }
    HIP_TRY(ctx, launch_merge_large_counts(p, d_tile_off, st));
    // (the same launch leaves the direct placement's error word and the tile count in the mailbox: one copy fetches everything)
    HIP_TRY(ctx, launch_count_nonzero(d_cnt, T, ctx->d_mail + 2, st, p.direct ? &p.sync->error : nullptr, p.n_tiles_dev, ctx->d_mail));
    // all-or-nothing: like the packing pass, the offsets are only written when the result fits the caller's buffer
    if (d_out_off) HIP_TRY(ctx, scan_excl_u32_to_u64_guarded(d_scan, scan_b, d_cnt, d_out_off, n1, ctx->d_mail, out_cap, st));
    if (e1) (void)hipEventRecord(e1, st);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint32_t n_tiles = (uint32_t)(ctx->h_mail[4] & 0xFFFFFFFFull);
    if (n_tiles > p.n_tiles_ub) return fail(ctx, II2_EHIP, "merge: internal error (tile bound exceeded)");
    if (p.direct && (ctx->h_mail[3] & 0xFFFFFFFFull)) {
        // A bounded wait of the direct placement ran out (its scanner workgroup and its workers did not run side by side: the
        // scheme rests on workgroups starting in index order, which the hardware does and HIP does not promise).  Nothing is
        // wrong with the inputs: the same merge again through the parking + packing pass, which has no waits between
        // workgroups and rewrites every offset and id the failed attempt may have left in the caller's buffers.
        ctx->merge_fallbacks++;
        const int64_t keep = ctx->opt_merge_direct, keep_skip = ctx->opt_merge_skip;
        ctx->opt_merge_direct = 0;
        ctx->opt_merge_skip &= ~64ll;
        const int rc2 = merge_core(ctx, k, views, n_terms, blocks_ub, postings_ub, tomb, d_out_off, d_out_values, out_cap, stats);
        ctx->opt_merge_direct = keep;
        ctx->opt_merge_skip = keep_skip;
        return rc2;
    }
    if (n_tiles == 0) ctx->h_mail[0] = 0;
    if (ctx->h_mail[0] > out_cap) return fail(ctx, II2_ECAPACITY, "merge: output buffer too small; nothing was written");
    if (stats) {
        stats->n_out = ctx->h_mail[0];
        stats->n_terms_out = ctx->h_mail[2];
        stats->n_tiles = n_tiles;
    }
    return II2_OK;
}

static int check_segs(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs) {
    if (k == 0 || k > MAX_LISTS || !segs) return fail(ctx, II2_EINVAL, "segment count must be 1..64");
    for (uint32_t s = 0; s < k; s++) {
        if (!segs[s] || segs[s]->device != ctx->device) return fail(ctx, II2_EINVAL, "segment is NULL or lives on another device");
        if (segs[s]->n_lists != segs[0]->n_lists) return fail(ctx, II2_EINVAL, "segments must be term-aligned (same n_lists)");
    }
    return II2_OK;
}

static int merge_unlocked(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const ii2_tomb *tomb, uint64_t *d_out_off,
                          uint32_t *d_out_values, uint64_t out_cap, ii2_merge_stats *stats) {
    int rc = check_segs(ctx, k, segs);
    if (rc) return rc;
    if (!d_out_values) return fail(ctx, II2_EINVAL, "merge: output buffer is NULL");
    std::vector<SegView> views(k);
    uint64_t n_in = 0, n_blk = 0;
    for (uint32_t s = 0; s < k; s++) {
        views[s] = SegView{segs[s]->d_blk_off, segs[s]->d_skip, segs[s]->d_payload, segs[s]->d_cnt, segs[s]->d_blk_list, segs[s]->d_last_doc, 0u, 0u};
        n_in += segs[s]->n_postings;       // (exact for whole segments and for views made by ii2_seg_select; views adopted from the device alignment carry their store's total)
        n_blk += segs[s]->merge_blocks();
    }
    ii2_merge_stats local;
    std::memset(&local, 0, sizeof local);
    rc = merge_core(ctx, k, views.data(), segs[0]->n_lists, n_blk, n_in, tomb, d_out_off, d_out_values, out_cap, &local);
    if (rc) return rc;
    local.n_in = n_in;
    if (stats) *stats = local;
    return II2_OK;
}

extern "C" {

int ii2_merge_segments(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const ii2_tomb *tomb, uint64_t *d_out_off,
                       uint32_t *d_out_values, uint64_t out_cap, ii2_merge_stats *stats) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return merge_unlocked(ctx, k, segs, tomb, d_out_off, d_out_values, out_cap, stats);
}

int ii2_merge_segments_to_seg(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const ii2_tomb *tomb, ii2_seg **out,
                              ii2_merge_stats *stats) {
    if (!ctx || !out) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    int rc = check_segs(ctx, k, segs);
    if (rc) return rc;
    uint64_t n_in = 0, bytes_in = 0, blocks_in = 0;
    for (uint32_t s = 0; s < k; s++) { n_in += segs[s]->n_postings; bytes_in += segs[s]->merge_bytes(); blocks_in += segs[s]->merge_blocks(); }
    const uint64_t T = segs[0]->n_lists;
    // the merged CSR waits for the encoder in the context's grow-only staging buffers (no hipMalloc per Shard.Merge)
    uint64_t *off = (uint64_t *)ii2_pool_get(ctx, 2, (T + 1) * sizeof(uint64_t));
    uint32_t *vals = (uint32_t *)ii2_pool_get(ctx, 3, (n_in + 1) * sizeof(uint32_t));
    if (!off || !vals) return fail(ctx, II2_ENOMEM, "merge output allocation failed");
    ii2_merge_stats local;
    rc = merge_unlocked(ctx, k, segs, tomb, off, vals, n_in, &local);
    if (rc) return rc;
    if (stats) *stats = local;
    if (local.n_terms_out == 0) return II2_OK;      // shard.go:219-225: nothing survives, no segment is written
    // (payload bound: a merged gap is never longer than the gap its posting had in its input list; the inputs' block-first ids had none)
    return ii2_seg_encode_stream_unlocked(ctx, T, off, vals, local.n_out, local.n_terms_out, bytes_in + 5 * blocks_in, out);
}

int ii2_union(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx, const ii2_tomb *tomb,
              uint32_t *d_out, uint64_t cap, uint64_t *count) {
    if (!ctx || !count) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0 || n > MAX_LISTS || !segs) return fail(ctx, II2_EINVAL, "ii2_union: list count must be 1..64");
    std::vector<SegView> views(n);
    bool any = false;
    uint64_t blocks_ub = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t li = list_idx ? list_idx[i] : 0;
        if (!segs[i] || segs[i]->device != ctx->device || li >= segs[i]->n_lists) return fail(ctx, II2_EINVAL, "ii2_union: bad list");
        if (int rc0 = ii2_seg_host_blk_off(ctx, segs[i])) return rc0;
        // a one-term view of the segment: blk_off shifted to the list
        views[i] = SegView{segs[i]->d_blk_off + li, segs[i]->d_skip, segs[i]->d_payload, segs[i]->d_cnt + li, segs[i]->d_blk_list, segs[i]->d_last_doc + li, (uint32_t)li, 0u};
        any |= segs[i]->h_blk_off[li + 1] > segs[i]->h_blk_off[li];
        blocks_ub += segs[i]->h_blk_off[li + 1] - segs[i]->h_blk_off[li];
    }
    if (!any) { *count = 0; return II2_OK; }
    if (!d_out) return fail(ctx, II2_EINVAL, "ii2_union: output buffer is NULL");
    {   // lists dense together: OR over byte-map tiles instead of the merge passes
        bool taken = false;
        uint64_t *d_cnt = ii2_mapped_mail(ctx, II2_MAIL_COUNT);      // the count goes straight into the pinned host mailbox
        int rc = ii2_union_dense_unlocked(ctx, n, segs, list_idx, tomb, d_out, cap, d_cnt ? d_cnt : ctx->d_mail, &taken);
        if (rc) return rc;
        if (taken) {
            if (!d_cnt) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mail + II2_MAIL_COUNT, ctx->d_mail, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            *count = ctx->h_mail[II2_MAIL_COUNT];
            if (*count > cap) return fail(ctx, II2_ECAPACITY, "ii2_union: result does not fit the output buffer (content unspecified)");
            return II2_OK;
        }
    }
    ii2_merge_stats st;
    std::memset(&st, 0, sizeof st);
    int rc = merge_core(ctx, n, views.data(), 1, blocks_ub, blocks_ub * II2_DV1_BLOCK, tomb, nullptr, d_out, cap, &st);
    if (rc) return rc;
    *count = st.n_out;
    return II2_OK;
}

// ---- host-buffer convenience -----------------------------------------------------------------
int ii2_merge_host(ii2_ctx *ctx, uint32_t k, uint64_t n_terms, const uint64_t *seg_off, const uint64_t *seg_base,
                   const uint32_t *values, const uint32_t *removed, uint64_t n_removed, uint64_t *out_off,
                   uint32_t *out_values, uint64_t out_cap, ii2_merge_stats *stats) {
    if (!ctx || !seg_off || !seg_base || !out_off || (k && seg_base[k] && !values))
        return fail(ctx, II2_EINVAL, "ii2_merge_host: bad argument");
    if (k == 0 || k > MAX_LISTS) return fail(ctx, II2_EINVAL, "segment count must be 1..64");
    std::vector<ii2_seg *> segs(k, nullptr);
    ii2_tomb *tomb = nullptr;
    int rc = II2_OK;
    uint64_t n_in = 0;
    for (uint32_t s = 0; s < k && !rc; s++) {
        rc = ii2_seg_encode(ctx, n_terms, seg_off + (size_t)s * (n_terms + 1), values + seg_base[s], II2_HOST, &segs[s]);
        n_in += seg_base[s + 1] - seg_base[s];
    }
    if (!rc && n_removed) rc = ii2_tomb_create(ctx, removed, n_removed, II2_HOST, &tomb);
    if (!rc && !out_values && n_in) rc = fail(ctx, II2_EINVAL, "ii2_merge_host: out_values is NULL");
    if (!rc) {
        std::lock_guard<std::mutex> g(ctx->mu);
        ii2_merge_stats local;
        std::memset(&local, 0, sizeof local);
        uint64_t *off = (uint64_t *)ii2_pool_get(ctx, 2, (n_terms + 1) * sizeof(uint64_t));
        uint32_t *vals = (uint32_t *)ii2_pool_get(ctx, 3, (n_in + 1) * sizeof(uint32_t));
        if (!off || !vals) rc = fail(ctx, II2_ENOMEM, "merge output allocation failed");
        if (!rc) rc = merge_unlocked(ctx, k, segs.data(), tomb, off, vals, n_in, &local);
        if (!rc && local.n_out > out_cap) rc = fail(ctx, II2_ECAPACITY, "ii2_merge_host: out_values too small; nothing was written");
        if (!rc) {
            hipError_t e = hipMemcpyAsync(out_off, off, (n_terms + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && local.n_out)
                e = hipMemcpyAsync(out_values, vals, local.n_out * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->err = std::string("merge download: ") + hipGetErrorString(e); rc = II2_EHIP; }
            if (stats) *stats = local;
        }
    }
    for (ii2_seg *s : segs) ii2_seg_free(s);
    ii2_tomb_free(tomb);
    return rc;
}

static int lists_host(ii2_ctx *ctx, bool is_union, uint32_t n, const uint64_t *list_off, const uint32_t *values,
                      const uint32_t *removed, uint64_t n_removed, uint32_t *out, uint64_t cap, uint64_t *count) {
    if (!ctx || !list_off || !count || n == 0 || n > MAX_LISTS) return fail(ctx, II2_EINVAL, "bad argument");
    ii2_seg *seg = nullptr;
    ii2_tomb *tomb = nullptr;
    int rc = ii2_seg_encode(ctx, n, list_off, values, II2_HOST, &seg);
    if (!rc && n_removed) rc = ii2_tomb_create(ctx, removed, n_removed, II2_HOST, &tomb);
    if (!rc) {
        uint64_t bound = 0;
        if (is_union) bound = list_off[n] - list_off[0];
        else {
            bound = ~0ull;
            for (uint32_t i = 0; i < n; i++) bound = std::min<uint64_t>(bound, list_off[i + 1] - list_off[i]);
        }
        DevBuf d_out;
        if (d_out.alloc((bound + 1) * sizeof(uint32_t)) != hipSuccess) rc = fail(ctx, II2_ENOMEM, "result allocation failed");
        std::vector<const ii2_seg *> segs(n, seg);
        std::vector<uint64_t> idx(n);
        for (uint32_t i = 0; i < n; i++) idx[i] = i;
        uint64_t c = 0;
        if (!rc)
            rc = is_union ? ii2_union(ctx, n, segs.data(), idx.data(), tomb, d_out.as<uint32_t>(), bound + 1, &c)
                          : ii2_intersect(ctx, n, segs.data(), idx.data(), tomb, d_out.as<uint32_t>(), bound + 1, &c);
        if (!rc && c > cap) rc = fail(ctx, II2_ECAPACITY, "output buffer too small; nothing was written");
        if (!rc && c) {
            if (!out) rc = fail(ctx, II2_EINVAL, "output buffer is NULL");
            else rc = ii2_copy_d2h(ctx, out, d_out.p, c * sizeof(uint32_t));
        }
        if (!rc) *count = c;
    }
    ii2_seg_free(seg);
    ii2_tomb_free(tomb);
    return rc;
}

int ii2_intersect_host(ii2_ctx *ctx, uint32_t n, const uint64_t *list_off, const uint32_t *values, const uint32_t *removed,
                       uint64_t n_removed, uint32_t *out, uint64_t cap, uint64_t *count) {
    return lists_host(ctx, false, n, list_off, values, removed, n_removed, out, cap, count);
}

int ii2_union_host(ii2_ctx *ctx, uint32_t n, const uint64_t *list_off, const uint32_t *values, const uint32_t *removed,
                   uint64_t n_removed, uint32_t *out, uint64_t cap, uint64_t *count) {
    return lists_host(ctx, true, n, list_off, values, removed, n_removed, out, cap, count);
}

}  // extern "C"
