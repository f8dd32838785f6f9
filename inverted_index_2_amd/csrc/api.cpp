// api.cpp — the C ABI of libii2_hip.so (include/ii2.h): contexts, segments, tombstones,
// intersect / union / merge entry points and the host-buffer convenience calls.
// There is deliberately no CPU implementation behind any entry point.
#include <algorithm>
#include <functional>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>

#include "internal.h"

using namespace ii2;

static thread_local std::string g_create_err;

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

// temp device allocation freed at scope exit (cold paths only: encode / import / host calls)
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return ii2::dm_malloc_retry(&p, bytes ? bytes : 16); }
    template <class T> T *as() const { return (T *)p; }
};

static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Workspace: reserve the total up front, then carve.
int ii2_ws_reserve(ii2_ctx *ctx, size_t bytes) {
    ctx->ws_used = 0;
    if (bytes <= ctx->ws_cap) return II2_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_cap = 0;
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    if (ii2::dm_malloc_retry((void **)&ctx->ws, want) != hipSuccess) return fail(ctx, II2_ENOMEM, "workspace allocation failed");
    ctx->ws_cap = want;
    return II2_OK;
}
template <class T> static T *ws_take(ii2_ctx *ctx, size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    T *p = (T *)(ctx->ws + ctx->ws_used);
    ctx->ws_used += bytes;
    return p;
}

void *ii2_pool_get(ii2_ctx *ctx, int slot, size_t bytes) {
    if (bytes <= ctx->pool_cap[slot] && ctx->pool[slot]) return ctx->pool[slot];
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pool[slot]) (void)hipFree(ctx->pool[slot]);
    ctx->pool[slot] = nullptr;
    ctx->pool_cap[slot] = 0;
    const size_t want = align_up(bytes + bytes / 8 + 256, 1 << 16);
    if (ii2::dm_malloc_retry((void **)&ctx->pool[slot], want) != hipSuccess) return nullptr;
    ctx->pool_cap[slot] = want;
    return ctx->pool[slot];
}

bool ii2_profile_pair(ii2_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1) {
    if (ctx->opt_profile_events <= 0) return false;
    if (ctx->prof_calls++ % (uint64_t)ctx->opt_profile_events != 0) return false;   // a timed event pair costs ~10 us of stream idle time
    std::pair<hipEvent_t, hipEvent_t> pr;
    if (!ctx->prof_pool.empty()) { pr = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
    else if (hipEventCreate(&pr.first) != hipSuccess || hipEventCreate(&pr.second) != hipSuccess) return false;
    ctx->prof_events.push_back(pr);
    *e0 = pr.first;
    *e1 = pr.second;
    return true;
}

extern "C" {

int ii2_profile_read(ii2_ctx *ctx, double *total_ms, uint64_t *launches) {
    if (!ctx || !total_ms || !launches) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    uint64_t n = 0;
    for (auto &pr : ctx->prof_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { tot += ms; n++; }
        ctx->prof_pool.push_back(pr);
    }
    ctx->prof_events.clear();
    *total_ms = tot;
    *launches = n;
    return II2_OK;
}

int ii2_profile_region(ii2_ctx *ctx, int begin) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t &e = ctx->region_ev[begin ? 0 : 1];
    if (!e) HIP_TRY(ctx, hipEventCreate(&e));
    HIP_TRY(ctx, hipEventRecord(e, ctx->stream));
    return II2_OK;
}

int ii2_profile_region_ms(ii2_ctx *ctx, double *ms) {
    if (!ctx || !ms || !ctx->region_ev[0] || !ctx->region_ev[1]) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipEventSynchronize(ctx->region_ev[1]));
    float f = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&f, ctx->region_ev[0], ctx->region_ev[1]));
    *ms = f;
    return II2_OK;
}

int ii2_abi_version(void) { return II2_ABI_VERSION; }

int ii2_ctx_create(int device, uint32_t flags, ii2_ctx **out) {
    (void)flags;
    if (!out) return II2_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_create_err = "no HIP device visible (libii2_hip has no CPU path)";
        return II2_ENODEVICE;
    }
    if (device < 0 || device >= n) { g_create_err = "device ordinal out of range"; return II2_EINVAL; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_err = "hipGetDeviceProperties failed"; return II2_EHIP; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return II2_ENODEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return II2_EHIP; }
    ii2_ctx *ctx = new (std::nothrow) ii2_ctx();
    if (!ctx) return II2_ENOMEM;
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    dm_user(+1);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_mail, II2_MAIL_WORDS * sizeof(uint64_t)) != hipSuccess ||
        ii2::dm_malloc_retry((void **)&ctx->d_mail, II2_MAIL_WORDS * sizeof(uint64_t)) != hipSuccess) {
        g_create_err = "context resource allocation failed";
        ii2_ctx_destroy(ctx);
        return II2_EHIP;
    }
    *out = ctx;
    return II2_OK;
}

void ii2_ctx_destroy(ii2_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ii2_lookback_forget(ctx);
    ii2_comm_destroy_internal(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->aux) (void)hipFree(ctx->aux);
    if (ctx->h_small_in) (void)hipHostFree(ctx->h_small_in);
    if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
    if (ctx->d_small_in) (void)hipFree(ctx->d_small_in);
    if (ctx->d_small_out) (void)hipFree(ctx->d_small_out);
    if (ctx->h_segs) (void)hipHostFree(ctx->h_segs);
    if (ctx->d_segs) (void)hipFree(ctx->d_segs);
    for (uint8_t *q : ctx->pool) if (q) (void)hipFree(q);
    if (ctx->d_debug) (void)hipFree(ctx->d_debug);
    if (ctx->d_lb) (void)hipFree(ctx->d_lb);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->d_mail) (void)hipFree(ctx->d_mail);
    if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
    for (hipEvent_t e : ctx->region_ev) if (e) (void)hipEventDestroy(e);
    for (auto *v : {&ctx->prof_events, &ctx->prof_pool})
        for (auto &pr : *v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    dm_user(-1);                 // the last context gives the cached segment arrays back (devmem.cpp)
    delete ctx;
}

const char *ii2_last_error(const ii2_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int ii2_ctx_sync(ii2_ctx *ctx) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->lb_pending && ctx->d_lb) {
        // asynchronous two-list ANDs since the last look: did a bounded wait of any of them run out?  (its count is all ones)
        unsigned long long e = 0;
        const uint32_t first = ctx->lb_pending;
        ctx->lb_pending = 0;
        HIP_TRY(ctx, hipMemcpy(&e, ctx->d_lb, sizeof e, hipMemcpyDeviceToHost));
        if (e >= first && e <= ctx->lb_epoch)
            return fail(ctx, II2_EHIP, "ii2_intersect_async: a workgroup's bounded wait ran out (that call's count is all ones); repeat it with ii2_intersect");
    }
    return II2_OK;
}

void *ii2_ctx_stream(ii2_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int ii2_ctx_device(const ii2_ctx *ctx) { return ctx ? ctx->device : -1; }

int ii2_dev_alloc(ii2_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ii2::dm_malloc_retry(dptr, bytes ? bytes : 16) != hipSuccess) return fail(ctx, II2_ENOMEM, "hipMalloc failed");
    return II2_OK;
}
int ii2_dev_free(ii2_ctx *ctx, void *dptr) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(dptr));
    return II2_OK;
}
int ii2_copy_h2d(ii2_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return II2_OK;
}
int ii2_copy_d2h(ii2_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return II2_OK;
}

}  // extern "C"

// ---- segments -----------------------------------------------------------------------------
static void seg_release(ii2_seg *s) {
    if (!s) return;
    if (s->born) (void)hipStreamSynchronize(s->born);     // never finished (an early exit): kernels of that call may still be writing its arrays
    if (s->in_slab) { delete s; return; }       // the store's slab holds every array (ii2_merge_small)
    // (devmem.cpp: the arrays go back to the size-class cache; no driver call, no device-wide wait)
    dm_free(s->d_blk_off);
    if (!s->store) {          // not yet handed to a store: still owned directly
        dm_free(s->d_skip);
        dm_free(s->d_payload);
    }
    dm_free(s->d_last_doc);
    dm_free(s->d_cnt);
    dm_free(s->d_blk_list);
    delete s;
}

// reads back the driver-choice statistics of single-list segments and the host blk_off mirror
// the three per-list / per-block arrays every segment carries next to its DV1 arrays
static int seg_alloc_meta(ii2_ctx *ctx, ii2_seg *seg) {
    if (dm_alloc((void **)&seg->d_last_doc, (seg->n_lists + 1) * sizeof(uint32_t)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    if (dm_alloc((void **)&seg->d_cnt, (seg->n_lists + 1) * sizeof(uint32_t)) != hipSuccess ||
        dm_alloc((void **)&seg->d_blk_list, (seg->n_blocks + 1) * sizeof(uint32_t)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    return II2_OK;
}

// have_meta: the encoder filled d_cnt / d_last_doc / d_blk_list from the CSR it encoded (no list has to be decoded for them)
// everything a finished segment still needs, enqueued (no wait): the caller waits for the stream and then calls seg_finish_done
static int seg_finish_enqueue(ii2_ctx *ctx, ii2_seg *seg, bool have_meta, bool *count_view_out) {
    if (!seg->store) {
        seg->store = std::make_shared<ii2_seg_store>();
        seg->store->d_skip = seg->d_skip;
        seg->store->d_payload = seg->d_payload;
    }
    if (!have_meta) {
        if (int rcm = seg_alloc_meta(ctx, seg)) return rcm;
        HIP_TRY(ctx, hipMemsetAsync(seg->d_blk_list, 0xFF, (seg->n_blocks + 1) * sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, launch_list_last_doc(seg->d_blk_off, seg->d_skip, seg->d_payload, seg->n_lists, seg->d_cnt, seg->d_blk_list, seg->d_last_doc,
                                          ctx->stream));
    }
    if (seg->n_lists <= ii2_seg::SPAN_MIRROR_MAX) {      // (a segment of many lists mirrors them when a query first asks: ii2_seg_host_blk_off)
        seg->h_blk_off.resize(seg->n_lists + 1);
        HIP_TRY(ctx, hipMemcpyAsync(seg->h_blk_off.data(), seg->d_blk_off, (seg->n_lists + 1) * sizeof(uint32_t),
                                    hipMemcpyDeviceToHost, ctx->stream));
    }
    // the lists' spans (what a query's path choice looks at), mirrored now so that no query has to fetch them
    if (seg->n_lists && seg->n_lists <= ii2_seg::SPAN_MIRROR_MAX) {
        if (int rcw = ii2_ws_reserve(ctx, align_up(3 * seg->n_lists * sizeof(uint32_t)) + 256)) return rcw;
        uint32_t *d_sp = ws_take<uint32_t>(ctx, 3 * seg->n_lists);
        HIP_TRY(ctx, launch_list_spans(seg->d_blk_off, seg->d_skip, seg->d_last_doc, seg->n_lists, d_sp, ctx->stream));
        seg->h_spans.resize(3 * seg->n_lists);
        HIP_TRY(ctx, hipMemcpyAsync(seg->h_spans.data(), d_sp, 3 * seg->n_lists * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    // a view's postings: the sum of its lists' counts (the store's total would make every merge of a few of its lists plan,
    // clear and launch for the whole store: milliseconds of fills and empty workgroups per call)
    const bool count_view = seg->is_view && seg->n_lists && seg->d_cnt;
    if (count_view) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_mail + 16, 0, sizeof(uint64_t), ctx->stream));
        HIP_TRY(ctx, launch_sum_u32(seg->d_cnt, seg->n_lists, ctx->d_mail + 16, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mail + 16, ctx->d_mail + 16, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    *count_view_out = count_view;
    return II2_OK;
}
static void seg_finish_done(ii2_ctx *ctx, ii2_seg *seg, bool count_view) {
    if (count_view) seg->n_postings = std::min<uint64_t>(seg->n_postings, ctx->h_mail[16]);
    seg->born = nullptr;                                    // finished: from here on the caller's contract governs its lifetime
}
static int seg_finish(ii2_ctx *ctx, ii2_seg *seg, bool have_meta = false) {
    bool count_view = false;
    if (int rc = seg_finish_enqueue(ctx, seg, have_meta, &count_view)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    seg_finish_done(ctx, seg, count_view);
    return II2_OK;
}

// encode from device-resident CSR (d_post_off, d_values) — unlocked
int ii2_seg_encode_dev_unlocked(ii2_ctx *ctx, uint64_t n_lists, const uint64_t *d_post_off, const uint32_t *d_values,
                                uint64_t n_postings, ii2_seg **out) {
    hipStream_t st = ctx->stream;
    std::unique_ptr<ii2_seg, void (*)(ii2_seg *)> seg(new (std::nothrow) ii2_seg(), seg_release);
    if (!seg) return II2_ENOMEM;
    seg->device = ctx->device;
    seg->born = ctx->stream;
    seg->n_lists = n_lists;
    seg->n_postings = n_postings;
    const uint64_t nb_bound = n_postings / II2_DV1_BLOCK + n_lists + 1;
    if (nb_bound >= (1ull << 31)) return fail(ctx, II2_ERANGE, "too many DV1 blocks for one segment");
    if (dm_alloc((void **)&seg->d_blk_off, (n_lists + 1) * sizeof(uint32_t)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    // scratch comes out of the context's grow-only workspace (sized by the upper bound of the block count)
    const size_t tmpb = scan_temp_bytes((size_t)std::max<uint64_t>(n_lists + 1, nb_bound + 1));
    if (int rcw = ii2_ws_reserve(ctx, align_up((n_lists + 1) * sizeof(uint32_t)) + align_up(tmpb) + align_up((nb_bound + 1) * sizeof(uint32_t)) +
                                          align_up((nb_bound + 1) * sizeof(uint64_t)) + 4096))
        return rcw;
    uint32_t *d_nblk = ws_take<uint32_t>(ctx, n_lists + 1);
    void *d_scan_tmp = ws_take<uint8_t>(ctx, tmpb);
    uint32_t *d_sizes = ws_take<uint32_t>(ctx, nb_bound + 1);
    uint64_t *d_boff = ws_take<uint64_t>(ctx, nb_bound + 1);
    HIP_TRY(ctx, launch_enc_list_blocks(d_post_off, n_lists, d_nblk, st));
    HIP_TRY(ctx, scan_excl_u32(d_scan_tmp, tmpb, d_nblk, seg->d_blk_off, n_lists + 1, st));
    uint32_t nb32 = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nb32, seg->d_blk_off + n_lists, sizeof nb32, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint64_t nb = nb32;
    seg->n_blocks = nb;
    if (dm_alloc((void **)&seg->d_skip, (nb + 1) * sizeof(ii2_skip)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    HIP_TRY(ctx, launch_enc_block_sizes(d_post_off, seg->d_blk_off, n_lists, d_values, nb, d_sizes, seg->d_skip, st));
    HIP_TRY(ctx, scan_excl_u32_to_u64(d_scan_tmp, tmpb, d_sizes, d_boff, nb + 1, st));
    uint64_t nbytes = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nbytes, d_boff + nb, sizeof nbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (nbytes >= 0xFFFFFFF0ull) return fail(ctx, II2_ERANGE, "segment payload exceeds the 4 GiB DV1 limit; split the segment");
    seg->n_bytes = nbytes;
    if (dm_alloc((void **)&seg->d_payload, nbytes + 16) != hipSuccess) return fail(ctx, II2_ENOMEM, "segment allocation failed");
    HIP_TRY(ctx, hipMemsetAsync(seg->d_payload + nbytes, 0, 16, st));
    if (int rcm = seg_alloc_meta(ctx, seg.get())) return rcm;
    HIP_TRY(ctx, hipMemsetAsync(seg->d_blk_list + nb, 0xFF, sizeof(uint32_t), st));         // (the entry past the last block, as k_list_last_doc leaves it)
    HIP_TRY(ctx, launch_enc_write(d_post_off, seg->d_blk_off, n_lists, d_values, nb, d_boff, seg->d_skip,
                                  seg->d_payload, n_postings, seg->d_blk_list, st));
    HIP_TRY(ctx, launch_enc_list_meta(d_post_off, d_values, n_lists, seg->d_cnt, seg->d_last_doc, st));
    int rc = seg_finish(ctx, seg.get(), true);
    if (rc) return rc;
    seg->born = nullptr;
    *out = seg.release();
    return II2_OK;
}

// encode from device-resident CSR in ONE pass over the ids (encode_stream.hip) - the tail of Shard.Merge (shard.go:207 ->
// file/writer.go:32-59).  The caller knows the number of non-empty lists and an upper bound of the payload (a merge: the inputs'
// payload bytes + 5 per input block - a merged gap is never longer than the gap its posting had in its input list, and only
// the inputs' block-first ids had none), so every array is allocated before the first kernel and nothing is read back until the
// end.  Falls back to the two-pass encoder above when a bounded wait of the look-back runs out or the bound does not hold.
int ii2_seg_encode_stream_unlocked(ii2_ctx *ctx, uint64_t n_lists, const uint64_t *d_post_off, const uint32_t *d_values, uint64_t n_postings,
                                   uint64_t n_nonempty, uint64_t payload_bound, ii2_seg **out) {
    hipStream_t st = ctx->stream;
    if (!ctx->opt_encode_stream || n_postings == 0) return ii2_seg_encode_dev_unlocked(ctx, n_lists, d_post_off, d_values, n_postings, out);
    std::unique_ptr<ii2_seg, void (*)(ii2_seg *)> seg(new (std::nothrow) ii2_seg(), seg_release);
    if (!seg) return II2_ENOMEM;
    seg->device = ctx->device;
    seg->born = ctx->stream;
    seg->n_lists = n_lists;
    seg->n_postings = n_postings;
    const uint64_t nb_bound = n_postings / II2_DV1_BLOCK + std::min<uint64_t>(n_nonempty, n_lists) + 1;
    if (nb_bound >= (1ull << 31)) return fail(ctx, II2_ERANGE, "too many DV1 blocks for one segment");
    const uint64_t cap = std::min<uint64_t>(std::min<uint64_t>(payload_bound, 5ull * n_postings), 0xFFFFFFEFull);
    seg->n_blocks = nb_bound;                      // (sizes the per-block arrays; the exact count comes back with the result)
    if (dm_alloc((void **)&seg->d_blk_off, (n_lists + 1) * sizeof(uint32_t)) != hipSuccess ||
        dm_alloc((void **)&seg->d_skip, (nb_bound + 1) * sizeof(ii2_skip)) != hipSuccess ||
        dm_alloc((void **)&seg->d_payload, cap + 16) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    if (int rcm = seg_alloc_meta(ctx, seg.get())) return rcm;
    const size_t tmpb = scan_temp_bytes((size_t)n_lists + 1);
    const size_t n_part = (size_t)((n_postings + 1023) / 1024) + 1;       // one word per 1024 output positions (k_enc_partition)
    if (int rcw = ii2_ws_reserve(ctx, align_up((n_lists + 1) * sizeof(uint32_t)) + align_up(tmpb) + align_up(n_part * sizeof(uint32_t)) + 4096)) return rcw;
    uint32_t *d_nblk = ws_take<uint32_t>(ctx, n_lists + 1);
    void *d_scan_tmp = ws_take<uint8_t>(ctx, tmpb);
    uint32_t *d_part = ws_take<uint32_t>(ctx, n_part);
    HIP_TRY(ctx, launch_enc_list_blocks(d_post_off, n_lists, d_nblk, st));
    HIP_TRY(ctx, scan_excl_u32(d_scan_tmp, tmpb, d_nblk, seg->d_blk_off, n_lists + 1, st));
    LookBack lb;
    if (int rcl = ii2_lookback_prepare(ctx, (size_t)enc_stream_workgroups(n_postings), &lb)) return rcl;
    ctx->lb_pending = 0;                           // (this call looks at its own result below)
    if (ctx->opt_encode_stream < 0) lb.spin = 0xFFFFFFFFu;      // tests: a wait runs out, the two-pass encoder takes over
    uint64_t *d_res = ctx->d_mail + 8;
    unsigned long long *d_dbg = nullptr;
    if (ctx->opt_debug_stamps == 3) {              // diagnostics: the encoder's cycle counters
        if (!ctx->d_debug && ii2::dm_malloc_retry((void **)&ctx->d_debug, (size_t)2048 * 8 * sizeof(unsigned long long)) != hipSuccess)
            return fail(ctx, II2_ENOMEM, "debug buffer allocation failed");
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_debug, 0, (size_t)2048 * 8 * sizeof(unsigned long long), st));
        d_dbg = ctx->d_debug;
    }
    if (int rcq = ii2_lookback_launch(ctx, true, [&] {
            return launch_enc_stream(d_post_off, d_values, seg->d_blk_off, n_lists, n_postings, seg->d_skip, seg->d_payload, cap, seg->d_blk_list,
                                     d_part, d_res, lb, d_dbg, st);
        })) return rcq;
    HIP_TRY(ctx, launch_enc_list_meta(d_post_off, d_values, n_lists, seg->d_cnt, seg->d_last_doc, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mail + 8, d_res, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    // the mirrors a finished segment keeps (list spans of small segments) depend on nothing the host still has to learn: they are
    // enqueued behind the encoder and ONE wait serves the byte count and them
    bool count_view = false;
    if (int rcf = seg_finish_enqueue(ctx, seg.get(), true, &count_view)) return rcf;
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint64_t nbytes = ctx->h_mail[8], nb = ctx->h_mail[9];
    if (nbytes == ~0ull || nb > nb_bound) {        // a bounded wait ran out, or the bound did not hold: the exact two-pass form
        ctx->lb_fallbacks++;
        seg.reset();
        return ii2_seg_encode_dev_unlocked(ctx, n_lists, d_post_off, d_values, n_postings, out);
    }
    seg->n_blocks = nb;
    seg->n_bytes = nbytes;
    seg_finish_done(ctx, seg.get(), count_view);
    *out = seg.release();
    return II2_OK;
}

// decode into device buffers — unlocked.  d_post_off may be null.
int ii2_seg_decode_dev_unlocked(ii2_ctx *ctx, const ii2_seg *seg, uint64_t *d_post_off, uint32_t *d_values) {
    hipStream_t st = ctx->stream;
    const uint64_t nb = seg->n_blocks;
    const size_t tmpb = scan_temp_bytes((size_t)nb + 1);
    if (int rcw = ii2_ws_reserve(ctx, align_up((nb + 1) * sizeof(uint32_t)) + align_up((nb + 1) * sizeof(uint64_t)) + align_up(tmpb) + 4096)) return rcw;
    uint32_t *d_counts = ws_take<uint32_t>(ctx, nb + 1);
    uint64_t *d_bpo = ws_take<uint64_t>(ctx, nb + 1);
    void *d_scan_tmp = ws_take<uint8_t>(ctx, tmpb);
    HIP_TRY(ctx, launch_dec_block_counts(seg->d_skip, seg->d_payload, nb, d_counts, st));
    HIP_TRY(ctx, scan_excl_u32_to_u64(d_scan_tmp, tmpb, d_counts, d_bpo, nb + 1, st));
    if (d_values) HIP_TRY(ctx, launch_dec_write(seg->d_skip, seg->d_payload, nb, d_bpo, d_values, st));
    if (d_post_off) HIP_TRY(ctx, launch_gather_post_off(seg->d_blk_off, d_bpo, seg->n_lists, d_post_off, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return II2_OK;
}

extern "C" {

int ii2_seg_encode(ii2_ctx *ctx, uint64_t n_lists, const uint64_t *post_off, const uint32_t *values, int where, ii2_seg **out) {
    if (!ctx || !post_off || !out || (where != II2_HOST && where != II2_DEVICE)) return fail(ctx, II2_EINVAL, "ii2_seg_encode: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    if (where == II2_DEVICE) {
        uint64_t n = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&n, post_off + n_lists, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return ii2_seg_encode_dev_unlocked(ctx, n_lists, post_off, values, n, out);
    }
    const uint64_t n = post_off[n_lists];
    for (uint64_t l = 0; l < n_lists; l++)
        if (post_off[l + 1] < post_off[l]) return fail(ctx, II2_EINVAL, "ii2_seg_encode: post_off must be non-decreasing");
    if (n && !values) return fail(ctx, II2_EINVAL, "ii2_seg_encode: values is NULL");
    uint64_t *dpo = (uint64_t *)ii2_pool_get(ctx, 0, (n_lists + 1) * sizeof(uint64_t));
    uint32_t *dv = (uint32_t *)ii2_pool_get(ctx, 1, n * sizeof(uint32_t));
    if (!dpo || !dv) return fail(ctx, II2_ENOMEM, "encode staging allocation failed");
    HIP_TRY(ctx, hipMemcpyAsync(dpo, post_off, (n_lists + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(dv, values, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    return ii2_seg_encode_dev_unlocked(ctx, n_lists, dpo, dv, n, out);
}

int ii2_seg_import(ii2_ctx *ctx, uint64_t n_lists, uint64_t n_postings, uint64_t n_blocks, uint64_t n_bytes, const uint32_t *blk_off,
                   const ii2_skip *skip, const uint8_t *payload, int where, ii2_seg **out) {
    if (!ctx || !blk_off || !skip || !out) return fail(ctx, II2_EINVAL, "ii2_seg_import: bad argument");
    if (n_blocks > 0xFFFFFFFEull || n_bytes > 0xFFFFFFFFull) return fail(ctx, II2_EINVAL, "ii2_seg_import: a segment holds < 2^32 blocks and payload bytes");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    const hipMemcpyKind kind = where == II2_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    std::unique_ptr<ii2_seg, void (*)(ii2_seg *)> seg(new (std::nothrow) ii2_seg(), seg_release);
    if (!seg) return II2_ENOMEM;
    seg->device = ctx->device;
    seg->born = ctx->stream;
    seg->n_lists = n_lists;
    seg->n_postings = n_postings;
    // the closing entries must agree with the stated array lengths: nothing below reads past those lengths
    uint32_t nb = 0;
    ii2_skip last;
    if (where == II2_HOST) {
        nb = blk_off[n_lists];
        last = skip[n_blocks];
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(&nb, blk_off + n_lists, sizeof nb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&last, skip + n_blocks, sizeof last, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (nb != n_blocks) return fail(ctx, II2_EINVAL, "ii2_seg_import: malformed DV1 segment (blk_off[n_lists] differs from n_blocks)");
    if (last.byte_off != n_bytes) return fail(ctx, II2_EINVAL, "ii2_seg_import: malformed DV1 segment (skip[n_blocks].byte_off differs from n_bytes)");
    seg->n_blocks = nb;
    seg->n_bytes = last.byte_off;
    if (seg->n_bytes && !payload) return fail(ctx, II2_EINVAL, "ii2_seg_import: payload is NULL");
    if (dm_alloc((void **)&seg->d_blk_off, (n_lists + 1) * sizeof(uint32_t)) != hipSuccess ||
        dm_alloc((void **)&seg->d_skip, ((uint64_t)nb + 1) * sizeof(ii2_skip)) != hipSuccess ||
        dm_alloc((void **)&seg->d_payload, seg->n_bytes + 16) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    HIP_TRY(ctx, hipMemcpyAsync(seg->d_blk_off, blk_off, (n_lists + 1) * sizeof(uint32_t), kind, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(seg->d_skip, skip, ((uint64_t)nb + 1) * sizeof(ii2_skip), kind, ctx->stream));
    if (seg->n_bytes) HIP_TRY(ctx, hipMemcpyAsync(seg->d_payload, payload, seg->n_bytes, kind, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(seg->d_payload + seg->n_bytes, 0, 16, ctx->stream));
    {   // refuse structurally broken input before any kernel walks it
        uint32_t *d_bad = (uint32_t *)ctx->d_mail;
        uint32_t bad = 0;
        HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, launch_validate_seg(seg->d_blk_off, n_lists, seg->d_skip, nb, seg->n_bytes, d_bad, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) {
            // the arrays are not yet owned by a store: seg_release frees them
            return fail(ctx, II2_EINVAL, "ii2_seg_import: malformed DV1 segment (offsets not monotone / out of range)");
        }
    }
    int rc = seg_finish(ctx, seg.get());
    if (rc) return rc;
    {   // block fill rule (every block full but a list's last): the merge places decoded blocks by it
        uint32_t *d_bad = (uint32_t *)ctx->d_mail;
        uint32_t bad = 0;
        HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, launch_validate_counts(seg->d_blk_off, seg->d_blk_list, seg->d_skip, seg->d_payload, nb, d_bad, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) return fail(ctx, II2_EINVAL, "ii2_seg_import: malformed DV1 segment (a block that is not its list's last must hold 256 postings)");
    }
    {   // the caller's n_postings sizes the decode buffers (ii2_seg_decode): it must be the count the blocks really hold
        uint64_t *d_sum = ctx->d_mail + 8;
        uint64_t sum = 0;
        HIP_TRY(ctx, hipMemsetAsync(d_sum, 0, sizeof(uint64_t), ctx->stream));
        HIP_TRY(ctx, launch_sum_u32(seg->d_cnt, n_lists, d_sum, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&sum, d_sum, sizeof sum, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (sum != n_postings) return fail(ctx, II2_EINVAL, "ii2_seg_import: n_postings differs from the number of postings the blocks hold");
    }
    seg->born = nullptr;
    *out = seg.release();
    return II2_OK;
}

int ii2_seg_decode(ii2_ctx *ctx, const ii2_seg *seg, uint64_t *post_off, uint32_t *values, int where) {
    if (!ctx || !seg) return fail(ctx, II2_EINVAL, "ii2_seg_decode: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (where == II2_DEVICE) return ii2_seg_decode_dev_unlocked(ctx, seg, post_off, values);
    uint64_t *dpo = (uint64_t *)ii2_pool_get(ctx, 0, (seg->n_lists + 1) * sizeof(uint64_t));
    uint32_t *dv = (uint32_t *)ii2_pool_get(ctx, 1, seg->n_postings * sizeof(uint32_t));
    if (!dpo || !dv) return fail(ctx, II2_ENOMEM, "decode staging allocation failed");
    int rc = ii2_seg_decode_dev_unlocked(ctx, seg, dpo, values ? dv : nullptr);
    if (rc) return rc;
    if (post_off) HIP_TRY(ctx, hipMemcpyAsync(post_off, dpo, (seg->n_lists + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (values && seg->n_postings)
        HIP_TRY(ctx, hipMemcpyAsync(values, dv, seg->n_postings * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return II2_OK;
}

int ii2_seg_export(ii2_ctx *ctx, const ii2_seg *seg, uint32_t *blk_off, ii2_skip *skip, uint8_t *payload) {
    if (!ctx || !seg) return fail(ctx, II2_EINVAL, "ii2_seg_export: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (blk_off) HIP_TRY(ctx, hipMemcpyAsync(blk_off, seg->d_blk_off, (seg->n_lists + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (skip) HIP_TRY(ctx, hipMemcpyAsync(skip, seg->d_skip, (seg->n_blocks + 1) * sizeof(ii2_skip), hipMemcpyDeviceToHost, ctx->stream));
    if (payload && seg->n_bytes) HIP_TRY(ctx, hipMemcpyAsync(payload, seg->d_payload, seg->n_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return II2_OK;
}

int ii2_seg_get_info(const ii2_seg *seg, ii2_seg_info *info) {
    if (!seg || !info) return II2_EINVAL;
    info->n_lists = seg->n_lists;
    info->n_postings = seg->n_postings;
    info->n_blocks = seg->n_blocks;
    info->n_bytes = seg->n_bytes;
    return II2_OK;
}

int ii2_seg_select(ii2_ctx *ctx, const ii2_seg *src, uint64_t n_out, const int64_t *src_list, ii2_seg **out) {
    if (!ctx || !src || !out || (n_out && !src_list) || src->device != ctx->device) return fail(ctx, II2_EINVAL, "ii2_seg_select: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    if (int rc0 = ii2_seg_host_blk_off(ctx, src)) return rc0;
    // selected lists must ascend, and only empty lists may be skipped between two selected ones
    std::vector<uint32_t> blk(n_out + 1, 0);
    int64_t prev = -1;
    uint32_t prev_end = 0;
    bool any = false;
    for (uint64_t i = 0; i < n_out; i++) {
        const int64_t j = src_list[i];
        if (j < 0) continue;
        if ((uint64_t)j >= src->n_lists || j <= prev) return fail(ctx, II2_EINVAL, "ii2_seg_select: list indices must ascend");
        const uint32_t b0 = src->h_blk_off[j], b1 = src->h_blk_off[j + 1];
        if (any && b0 != prev_end) return fail(ctx, II2_EINVAL, "ii2_seg_select: a non-empty list lies between two selected lists");
        prev = j;
        prev_end = b1;
        any = true;
    }
    uint32_t next = any ? prev_end : 0u;
    blk[n_out] = next;
    for (uint64_t i = n_out; i-- > 0;) {
        const int64_t j = src_list[i];
        if (j >= 0) next = src->h_blk_off[j];
        blk[i] = next;
    }
    std::unique_ptr<ii2_seg, void (*)(ii2_seg *)> seg(new (std::nothrow) ii2_seg(), seg_release);
    if (!seg) return II2_ENOMEM;
    seg->device = ctx->device;
    seg->born = ctx->stream;
    seg->store = src->store;
    seg->is_view = true;
    seg->d_skip = src->d_skip;
    seg->d_payload = src->d_payload;
    seg->n_lists = n_out;
    seg->n_blocks = src->n_blocks;
    seg->n_bytes = src->n_bytes;
    seg->n_postings = src->n_postings;      // upper bound: postings of unselected window lists are still counted
    if (dm_alloc((void **)&seg->d_blk_off, (n_out + 1) * sizeof(uint32_t)) != hipSuccess) return fail(ctx, II2_ENOMEM, "segment allocation failed");
    HIP_TRY(ctx, hipMemcpyAsync(seg->d_blk_off, blk.data(), (n_out + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    // the view's window of the store: blocks [blk[0], blk[n_out]) and the payload bytes between their skip entries
    ii2_skip w0, w1;
    std::memset(&w0, 0, sizeof w0); std::memset(&w1, 0, sizeof w1);
    if (n_out && blk[n_out] > blk[0]) {
        HIP_TRY(ctx, hipMemcpyAsync(&w0, src->d_skip + blk[0], sizeof w0, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&w1, src->d_skip + blk[n_out], sizeof w1, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_out && blk[n_out] > blk[0]) {
        seg->win_blocks = (uint64_t)blk[n_out] - blk[0];
        seg->win_bytes = std::max<uint64_t>((uint64_t)w1.byte_off - w0.byte_off, 1);
    } else {
        seg->win_blocks = 1; seg->win_bytes = 1;             // (nothing selected: a window of nothing; 0 would mean "the totals")
    }
    int rc = seg_finish(ctx, seg.get());
    if (rc) return rc;
    seg->born = nullptr;
    *out = seg.release();
    return II2_OK;
}

}  // extern "C"

// an empty segment shell with room for the given shape (ii2_seg_allgather fills it): ctx->mu held
int ii2_seg_alloc_internal(ii2_ctx *ctx, uint64_t n_lists, uint64_t n_postings, uint64_t n_blocks, uint64_t n_bytes, ii2_seg **out, bool with_meta) {
    std::unique_ptr<ii2_seg, void (*)(ii2_seg *)> seg(new (std::nothrow) ii2_seg(), seg_release);
    if (!seg) return II2_ENOMEM;
    seg->device = ctx->device;
    seg->born = ctx->stream;
    seg->n_lists = n_lists;
    seg->n_postings = n_postings;
    seg->n_blocks = n_blocks;
    seg->n_bytes = n_bytes;
    if (dm_alloc((void **)&seg->d_blk_off, (n_lists + 1) * sizeof(uint32_t)) != hipSuccess ||
        dm_alloc((void **)&seg->d_skip, (n_blocks + 1) * sizeof(ii2_skip)) != hipSuccess ||
        dm_alloc((void **)&seg->d_payload, n_bytes + 16) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    if (with_meta) { if (int rcm = seg_alloc_meta(ctx, seg.get())) return rcm; }
    seg->born = nullptr;
    *out = seg.release();
    return II2_OK;
}

// the parts of a gathered segment (rank r: lists [lo[r], lo[r+1]), blocks [bo[r], bo[r+1]), bytes [qo[r], qo[r+1])) still
// number their blocks and bytes from 0: shift them, write the closing entries and derive the per-list arrays.  ctx->mu held
int ii2_seg_rebase_internal(ii2_ctx *ctx, ii2_seg *seg, int world, const uint64_t *lo, const uint64_t *bo, const uint64_t *qo) {
    hipStream_t st = ctx->stream;
    const bool have_meta = seg->d_blk_list != nullptr;           // the derived arrays travelled with the parts (ii2_seg_allgather)
    for (int r = 0; r < world; r++) {
        HIP_TRY(ctx, launch_seg_rebase(seg->d_blk_off + lo[r], lo[r + 1] - lo[r], (uint32_t)bo[r], seg->d_skip + bo[r], bo[r + 1] - bo[r], (uint32_t)qo[r],
                                       have_meta ? seg->d_blk_list + bo[r] : nullptr, (uint32_t)lo[r], st));
    }
    HIP_TRY(ctx, launch_seg_close(seg->d_blk_off + lo[world], (uint32_t)bo[world], seg->d_skip + bo[world], (uint32_t)qo[world], seg->d_payload + qo[world],
                                  have_meta ? seg->d_blk_list + bo[world] : nullptr, st));
    return seg_finish(ctx, seg, have_meta);                      // (one host wait: the segment is ready when this returns)
}

// host mirror of blk_off, fetched on first use (views built on the device do not have one)
int ii2_seg_host_blk_off(ii2_ctx *ctx, const ii2_seg *seg) {
    std::lock_guard<std::mutex> sg(seg->span_mu);
    if (seg->h_blk_off.size() == seg->n_lists + 1) return II2_OK;
    std::vector<uint32_t> h(seg->n_lists + 1);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), seg->d_blk_off, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const_cast<ii2_seg *>(seg)->h_blk_off.swap(h);
    return II2_OK;
}

int ii2_seg_host_cnt(ii2_ctx *ctx, const ii2_seg *seg) {
    std::lock_guard<std::mutex> sg(seg->span_mu);
    if (seg->h_cnt.size() == seg->n_lists) return II2_OK;
    std::vector<uint32_t> h(seg->n_lists);
    if (seg->n_lists) {
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), seg->d_cnt, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const_cast<ii2_seg *>(seg)->h_cnt.swap(h);
    return II2_OK;
}

// a view over src's store whose four per-slot arrays were built on the device (align.hip); takes ownership of them
int ii2_seg_adopt_view(ii2_ctx *ctx, const ii2_seg *src, uint64_t n_out, uint32_t *d_blk_off, uint32_t *d_cnt, uint32_t *d_last_doc,
                       uint32_t *d_blk_list, ii2_seg **out) {
    ii2_seg *seg = new (std::nothrow) ii2_seg();
    if (!seg) {
        dm_free(d_blk_off); dm_free(d_cnt); dm_free(d_last_doc); dm_free(d_blk_list);
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    }
    seg->device = ctx->device;
    seg->born = ctx->stream;
    seg->store = src->store;
    seg->is_view = true;
    seg->d_skip = src->d_skip;
    seg->d_payload = src->d_payload;
    seg->n_lists = n_out;
    seg->n_blocks = src->n_blocks;
    seg->n_bytes = src->n_bytes;
    seg->n_postings = src->n_postings;      // upper bound, as for ii2_seg_select
    seg->d_blk_off = d_blk_off;
    seg->d_cnt = d_cnt;
    seg->d_last_doc = d_last_doc;
    seg->d_blk_list = d_blk_list;
    seg->born = nullptr;
    *out = seg;
    return II2_OK;
}

extern "C" {

void ii2_devmem_stats(uint64_t *live_bytes, uint64_t *idle_bytes) { dm_stats(live_bytes, idle_bytes); }

void ii2_seg_free(ii2_seg *seg) {
    if (!seg) return;
    // the caller guarantees no call that reads the segment is still running (as with any free); the arrays go back to the
    // size-class cache (devmem.cpp), which does not wait for the device
    (void)hipSetDevice(seg->device);
    seg_release(seg);
}

// ---- tombstones ---------------------------------------------------------------------------
int ii2_tomb_create(ii2_ctx *ctx, const uint32_t *removed, uint64_t n, int where, ii2_tomb **out) {
    if (!ctx || !out || (n && !removed)) return fail(ctx, II2_EINVAL, "ii2_tomb_create: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    const uint32_t *d_rem = removed;
    if (where == II2_HOST && n) {
        uint32_t *staged = (uint32_t *)ii2_pool_get(ctx, 1, n * sizeof(uint32_t));
        if (!staged) return fail(ctx, II2_ENOMEM, "tombstone staging allocation failed");
        HIP_TRY(ctx, hipMemcpyAsync(staged, removed, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        d_rem = staged;
    }
    uint32_t mx = 0;
    if (n) {
        uint32_t *d_mx = (uint32_t *)ctx->d_mail;
        HIP_TRY(ctx, hipMemsetAsync(d_mx, 0, sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, launch_max_u32(d_rem, n, d_mx, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&mx, d_mx, sizeof mx, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ii2_tomb *t = new (std::nothrow) ii2_tomb();
    if (!t) return II2_ENOMEM;
    t->device = ctx->device;
    t->n_words = n ? (uint64_t)(mx >> 5) + 1 : 0;
    // bitmap, then its summary (1 bit per (1 << TOMB_SUM_SHIFT) docs) in the same allocation
    const uint64_t words_padded = (t->n_words + 4 + 3) & ~3ull;
    const uint64_t n_sum = (t->n_words >> TOMB_SUM_SHIFT) + 2;
    if (ii2::dm_malloc_retry((void **)&t->d_words, (words_padded + n_sum) * sizeof(uint32_t)) != hipSuccess) {
        delete t;
        return fail(ctx, II2_ENOMEM, "tombstone bitmap allocation failed");
    }
    t->d_summary = t->d_words + words_padded;
    hipError_t e = hipMemsetAsync(t->d_words, 0, (words_padded + n_sum) * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) e = launch_tomb_build(d_rem, n, t->d_words, t->n_words, t->d_summary, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(t->d_words);
        delete t;
        ctx->err = std::string("tombstone build: ") + hipGetErrorString(e);
        return II2_EHIP;
    }
    *out = t;
    return II2_OK;
}

void ii2_tomb_free(ii2_tomb *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->d_words) (void)hipFree(t->d_words);
    delete t;
}

}  // extern "C"

// ---- intersect ------------------------------------------------------------------------------
static int make_list_view(ii2_ctx *ctx, const ii2_seg *seg, uint64_t idx, ListView *v) {
    if (!seg || idx >= seg->n_lists) return fail(ctx, II2_EINVAL, "list index out of range");
    if (seg->device != ctx->device) return fail(ctx, II2_EINVAL, "segment lives on another device");
    if (int rc = ii2_seg_host_blk_off(ctx, seg)) return rc;
    const uint32_t b0 = seg->h_blk_off[idx], b1 = seg->h_blk_off[idx + 1];
    v->skip = seg->d_skip + b0;
    v->payload = seg->d_payload;
    v->last_doc = seg->d_last_doc + idx;
    v->nblk = b1 - b0;
    v->pad = 0;
    return II2_OK;
}

// device-side address of a word of the pinned host mailbox (hipHostMalloc memory is mapped), or null if the runtime
// does not give one
uint64_t *ii2_mapped_mail(ii2_ctx *ctx, uint32_t word) {
    if (!ctx->d_mail_mapped) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ctx->h_mail, 0) != hipSuccess || !dp) { (void)hipGetLastError(); return nullptr; }
        ctx->d_mail_mapped = (uint64_t *)dp;
    }
    return ctx->d_mail_mapped + word;
}

// AND / OR of lists that hold <= SMALL_SET_BLOCKS blocks together: one single-workgroup kernel (setop_small.hip).
// *taken = false when the query is too large (or the path is switched off).
static int ii2_setop_small_unlocked(ii2_ctx *ctx, bool is_union, uint32_t n, const ListView *views, const ii2_seg *const *segs,
                             const uint64_t *list_idx, const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *d_count, bool *taken) {
    *taken = false;
    if (!ctx->opt_small_setop || n == 0 || n > MAX_LISTS) return II2_OK;
    SmallSetParams sp;
    std::memset(&sp, 0, sizeof sp);
    uint32_t m = 0, nb = 0;
    for (uint32_t i = 0; i < n; i++) {             // by blocks first: no size is fetched for a query that is too large anyway
        if (views[i].nblk == 0 && !is_union) return II2_OK;      // (an AND with an empty list is answered by the caller)
        if (views[i].nblk > SMALL_SET_BLOCKS - nb) return II2_OK;
        nb += views[i].nblk;
    }
    nb = 0;
    uint32_t np = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (views[i].nblk == 0) continue;
        if (int rc = ii2_seg_host_cnt(ctx, segs[i])) return rc;
        const uint32_t c = segs[i]->h_cnt[list_idx ? list_idx[i] : 0];
        // an AND pays off below ~2k postings (the general path is four launches, ~14-20 us whatever the size); an OR up
        // to the kernel's capacity (the merge passes are ~40 launches)
        if (c > (is_union ? SMALL_SET_POSTINGS : SMALL_SET_POSTINGS / 4u) - np) return II2_OK;
        sp.lists[m] = views[i];
        sp.blk_base[m] = nb;
        sp.lpre[m] = np;
        nb += views[i].nblk;
        np += c;
        m++;
    }
    if (m == 0) return II2_OK;
    sp.blk_base[m] = nb;
    sp.lpre[m] = np;
    sp.n_lists = m;
    sp.n_blocks = nb;
    sp.is_union = is_union ? 1u : 0u;
    sp.tomb = tomb ? tomb->d_words : nullptr;
    sp.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
    sp.out = d_out;
    sp.out_cap = cap;
    sp.d_count = d_count;
    if (!ctx->d_small) {
        const size_t bytes = ((size_t)SMALL_SET_POSTINGS + 16) * sizeof(uint32_t);
        if (ii2::dm_malloc_retry((void **)&ctx->d_small, bytes) != hipSuccess) return fail(ctx, II2_ENOMEM, "small set-operation scratch allocation failed");
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_small, 0, bytes, ctx->stream));
    }
    sp.sorted = ctx->d_small;
    sp.ticket = ctx->d_small + (size_t)SMALL_SET_POSTINGS;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ii2_profile_pair(ctx, &e0, &e1);
    HIP_TRY(ctx, launch_setop_small(sp, ctx->stream, e0, e1));
    *taken = true;
    return II2_OK;
}

// Look-back records for a launch of n_wg workgroups (lookback.h): a buffer of the context's own that only those kernels write,
// every word tagged with its launch's number - nothing to clear between launches (cleared when it grows or the numbers wrap).
// Kernels whose workgroups wait for lower-numbered workgroups of the SAME launch (lookback.h) must not share the GPU with
// another such launch: workgroups go to the eight XCDs round robin and every XCD starts its share in order, but the XCDs do not
// wait for each other - with two of these kernels from two contexts side by side, the workgroups of one can fill an XCD, spinning
// for a predecessor that sits in another XCD's queue behind the spinning workgroups of the other: each waits for a slot the
// other holds (seen: three contexts encoding merged segments at once under a tracer - every encoder ran out its seconds of
// bounded waits and handed over to the two-pass encoder).  The merge's tile kernel with direct placement waits too, but only for
// its scanner, workgroup 0 - the head of its XCD's queue: several of THEM side by side are safe (a freed slot goes to a
// scanner before it goes to any other workgroup of that kernel), one of them beside a look-back kernel is not.
// So, per device: a look-back kernel (exclusive = true) waits - on the GPU, not on the host - for every earlier kernel of
// either kind and every later one waits for it; tile kernels (exclusive = false) only wait for the last look-back kernel.
// Kernels without waits between workgroups run beside all of them as before.
// The order costs nothing while one context works alone: stream order already is an order, so nothing is recorded or waited for
// until a DIFFERENT stream comes along - which then records an event on the earlier stream itself (an event recorded late only
// covers more) and waits for that.  (Recording after every launch put a marker between back-to-back queries of one context: 41 ->
// 46 us per two-list AND.)
namespace {
struct LbChain {
    hipEvent_t ev_x = nullptr;              // the last exclusive kernel: recorded here (x_recorded), or still only in its stream's order (x_stream)
    bool x_recorded = false;
    hipStream_t x_stream = nullptr;         // stream of the last exclusive kernel while the event is not recorded yet
    hipStream_t x_owner = nullptr;          // ... and once it is (that stream needs no wait)
    hipEvent_t ev_s[16] = {};               // the shared kernels since then: one stream each
    hipStream_t s_stream[16] = {};
    unsigned n_s = 0;
};
std::mutex g_lb_mu;
LbChain g_lb[64];
}  // namespace
int ii2_lookback_launch(ii2_ctx *ctx, bool exclusive, const std::function<hipError_t()> &launch) {
    const int dev = ctx->device;
    if (dev < 0 || dev >= 64) return fail(ctx, II2_EINVAL, "device number out of range");
    if (ctx->opt_debug_no_chain) { HIP_TRY(ctx, launch()); return II2_OK; }      // (experiments: what happens without the order)
    std::lock_guard<std::mutex> g(g_lb_mu);
    LbChain &c = g_lb[dev];
    hipStream_t me = ctx->stream;
    if (!c.ev_x) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&c.ev_x, hipEventDisableTiming));
        for (int i = 0; i < 16; i++) HIP_TRY(ctx, hipEventCreateWithFlags(&c.ev_s[i], hipEventDisableTiming));
    }
    // behind the last exclusive kernel
    if (c.x_stream && c.x_stream != me) {
        HIP_TRY(ctx, hipEventRecord(c.ev_x, c.x_stream));
        c.x_recorded = true; c.x_owner = c.x_stream; c.x_stream = nullptr;
    }
    if (c.x_recorded && c.x_owner != me) HIP_TRY(ctx, hipStreamWaitEvent(me, c.ev_x, 0));
    if (exclusive) {
        // ... and behind every shared kernel since then
        for (unsigned i = 0; i < c.n_s; i++) {
            if (c.s_stream[i] == me) continue;
            HIP_TRY(ctx, hipEventRecord(c.ev_s[i], c.s_stream[i]));
            HIP_TRY(ctx, hipStreamWaitEvent(me, c.ev_s[i], 0));
        }
        c.n_s = 0;
        HIP_TRY(ctx, launch());
        c.x_stream = me; c.x_recorded = false; c.x_owner = nullptr;
    } else {
        bool known = false;
        for (unsigned i = 0; i < c.n_s; i++) known |= c.s_stream[i] == me;
        if (!known && c.n_s == 16u) {          // (more than 16 streams with a shared kernel in flight: this one queues behind the first)
            HIP_TRY(ctx, hipEventRecord(c.ev_s[0], c.s_stream[0]));
            HIP_TRY(ctx, hipStreamWaitEvent(me, c.ev_s[0], 0));
            c.s_stream[0] = me;
            known = true;
        }
        HIP_TRY(ctx, launch());
        if (!known) c.s_stream[c.n_s++] = me;
    }
    return II2_OK;
}
// a context goes away (its stream has been waited for): nothing of it is left to wait for
void ii2_lookback_forget(ii2_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= 64) return;
    std::lock_guard<std::mutex> g(g_lb_mu);
    LbChain &c = g_lb[ctx->device];
    if (c.x_stream == ctx->stream) c.x_stream = nullptr;
    if (c.x_owner == ctx->stream) c.x_owner = nullptr;          // (the recorded event stays valid: it is the chain's own)
    unsigned k = 0;
    for (unsigned i = 0; i < c.n_s; i++)
        if (c.s_stream[i] != ctx->stream) c.s_stream[k++] = c.s_stream[i];
    c.n_s = k;
}

int ii2_lookback_prepare(ii2_ctx *ctx, size_t n_wg, ii2::LookBack *lb) {
    hipStream_t st = ctx->stream;
    if (n_wg > ctx->lb_cap || ctx->lb_epoch >= (1u << 24) - 1u) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (n_wg > ctx->lb_cap) {
            if (ctx->d_lb) (void)hipFree(ctx->d_lb);
            ctx->d_lb = nullptr;
            ctx->lb_cap = 0;
            const size_t cap_wg = std::max<size_t>(4096, n_wg + n_wg / 4);
            if (ii2::dm_malloc_retry((void **)&ctx->d_lb, (8 + cap_wg + 2 * ((cap_wg + 63) / 64)) * sizeof(unsigned long long)) != hipSuccess)
                return fail(ctx, II2_ENOMEM, "look-back records allocation failed");
            ctx->lb_cap = cap_wg;
        }
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_lb, 0, (8 + ctx->lb_cap + 2 * ((ctx->lb_cap + 63) / 64)) * sizeof(unsigned long long), st));
        ctx->lb_epoch = 0;
        ctx->lb_pending = 0;
    }
    lb->err = ctx->d_lb;
    lb->agg = ctx->d_lb + 8;
    lb->grp = ctx->d_lb + 8 + ctx->lb_cap;
    lb->epoch = ++ctx->lb_epoch;
    lb->spin = 0;
    if (!ctx->lb_pending) ctx->lb_pending = lb->epoch;
    return II2_OK;
}

// first doc, first doc of the last block and last doc of a non-empty list: fetched once per (segment, list), then cached
static int list_span(ii2_ctx *ctx, const ii2_seg *seg, uint64_t idx, const ListView &v, ii2_seg::ListSpan *out) {
    if (seg->h_spans.size() == 3 * seg->n_lists && idx < seg->n_lists) {      // mirrored when the segment was created: no fetch, no sync
        *out = ii2_seg::ListSpan{seg->h_spans[3 * idx], seg->h_spans[3 * idx + 1], seg->h_spans[3 * idx + 2]};
        return II2_OK;
    }
    {
        std::lock_guard<std::mutex> sg(seg->span_mu);
        auto hit = seg->span_cache.find(idx);
        if (hit != seg->span_cache.end()) { *out = hit->second; return II2_OK; }
    }
    ii2_skip e[2];
    uint32_t last = 0;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(&e[0], v.skip, sizeof(ii2_skip), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(&e[1], v.skip + (v.nblk - 1), sizeof(ii2_skip), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(&last, v.last_doc, sizeof last, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *out = ii2_seg::ListSpan{e[0].first_doc, e[1].first_doc, last};
    std::lock_guard<std::mutex> sg(seg->span_mu);
    seg->span_cache[idx] = *out;
    return II2_OK;
}

static int intersect_unlocked(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx,
                              const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *d_count) {
    if (n == 0 || n > MAX_LISTS || !segs || !d_count) return fail(ctx, II2_EINVAL, "ii2_intersect: bad argument");
    hipStream_t st = ctx->stream;
    IntersectParams p;
    std::memset(&p, 0, sizeof p);
    std::vector<ListView> views(n);
    bool any_empty = false;
    for (uint32_t i = 0; i < n; i++) {
        int rc = make_list_view(ctx, segs[i], list_idx ? list_idx[i] : 0, &views[i]);
        if (rc) return rc;
        any_empty |= views[i].nblk == 0;
    }
    if (any_empty) {
        HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint64_t), st));
        return II2_OK;
    }
    if (!d_out) return fail(ctx, II2_EINVAL, "ii2_intersect: output buffer is NULL");
    if (ctx->opt_intersect_g <= 0) {
        bool taken = false;
        if (int rc = ii2_setop_small_unlocked(ctx, false, n, views.data(), segs, list_idx, tomb, d_out, cap, d_count, &taken)) return rc;
        if (taken) return II2_OK;
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return views[a].nblk < views[b].nblk; });
    std::vector<const ii2_seg *> vseg(n);
    std::vector<uint64_t> vidx(n);
    {
        std::vector<ListView> sorted(n);
        for (uint32_t i = 0; i < n; i++) { sorted[i] = views[order[i]]; vseg[i] = segs[order[i]]; vidx[i] = list_idx ? list_idx[order[i]] : 0; }
        views.swap(sorted);
    }
    for (uint32_t i = 0; i < n; i++) p.lists[i] = views[i];
    p.n_lists = n;
    const uint32_t nblk0 = views[0].nblk;
    // tile height: aim the tile's doc span at the LDS byte map; keep >= ~8 tiles per CU
    uint32_t G = 1;
    double per_block_span = 0;                   // docs per driver block (tile-height heuristics)
    uint32_t dense_first_doc = 0, dense_last_doc = 0;
    if (ctx->opt_intersect_g > 0) G = (uint32_t)std::min<int64_t>(ctx->opt_intersect_g, ISECT_GMAX);
    else if (nblk0 > 1) {
        ii2_seg::ListSpan ends;
        if (int rc = list_span(ctx, vseg[0], vidx[0], views[0], &ends)) return rc;
        const double per_block = (double)(ends.last_block_first_doc - ends.first_doc) / (double)(nblk0 - 1);
        per_block_span = per_block;
        dense_first_doc = ends.first_doc;
        dense_last_doc = ends.last_doc;
        const double g = per_block > 0 ? 0.85 * ISECT_SMAX / per_block : ISECT_GMAX;
        G = g >= ISECT_GMAX ? ISECT_GMAX : g < 1 ? 1u : (uint32_t)g;
        while (G > 1 && nblk0 / G < 8u * (uint32_t)ctx->cu_count) G >>= 1;
    }
    // Lists that are dense together (the headline 2-term query): every wave streams through its own run of driver
    // blocks, no partition pass, no workgroup barriers (intersect_dense.hip).  The driver must be dense enough for
    // the 1-bit-per-doc result bitmap to stay small next to the payload.
    if (ctx->opt_intersect_dense && ctx->opt_intersect_g <= 0 && n >= 2 && n <= DENSE_MAXL && nblk0 >= 1024 &&
        per_block_span > 0 && per_block_span <= 1100.0) {
        DenseParams dp;
        std::memset(&dp, 0, sizeof dp);
        for (uint32_t i = 0; i < n; i++) {
            dp.lists[i] = views[i];
            ii2_seg::ListSpan sp;
            if (int rc = list_span(ctx, vseg[i], vidx[i], views[i], &sp)) return rc;
            dp.first_doc[i] = sp.first_doc;
            dp.last_doc[i] = sp.last_doc;
        }
        dp.n_lists = n;
        // a wave's passes take 16 driver blocks each; one round (16 blocks) per wave by default: more, shorter waves balance
        // better than fewer, longer ones (measured on Zipf rank pairs 1/2 ... 2/3/5), and waves never wait for each other
        uint32_t bpw = ctx->opt_dense_bpw > 0 ? (uint32_t)ctx->opt_dense_bpw : 16u;
        bpw = std::min<uint32_t>(std::max<uint32_t>((bpw + 15u) & ~15u, 16u), 1024u);
        dp.bpw = bpw;
        dp.n_waves = (nblk0 + bpw - 1) / bpw;
        const uint32_t grid = (dp.n_waves + 3u) / 4u;
        dp.n_meta = grid * 4u;
        dp.base32 = dense_first_doc & ~31u;
        // two lists: the longer one is marked, the shorter one's postings are tested where they sit (intersect_and2.hip): the
        // hand-over to the second kernel is one bit per posting of the shorter list instead of a result bitmap
        const bool and2 = n == 2 && ctx->opt_intersect_and2 && bpw == 16u;
        const uint64_t bm_words = and2 ? (uint64_t)dp.n_meta * 128u : (((uint64_t)dense_last_doc - dp.base32) >> 5) + 1 + dp.n_meta + 8;
        size_t need = align_up(bm_words * sizeof(uint32_t)) + align_up((size_t)dp.n_meta * sizeof(uint4)) + align_up((size_t)grid * sizeof(uint32_t)) + 4096;
        int rc = ii2_ws_reserve(ctx, need);
        if (rc) return rc;
        dp.bitmap = ws_take<uint32_t>(ctx, bm_words);
        dp.hmask = and2 ? reinterpret_cast<uint2 *>(dp.bitmap) : nullptr;
        if (and2 && ctx->opt_intersect_and2 == 1) {
            // one launch (k_and2_fused): the look-back records live in a buffer of their own (only these kernels write it, every
            // word tagged with its launch's number: nothing to clear between launches)
            if (int rcl = ii2_lookback_prepare(ctx, grid, &dp.lb)) return rcl;
            if (ctx->opt_and2_spin) dp.lb.spin = ctx->opt_and2_spin > 0 ? (uint32_t)std::min<int64_t>(ctx->opt_and2_spin, 0x7FFFFFFF) : 0xFFFFFFFFu;
            const double spanA = (double)dp.last_doc[1] - (double)dp.first_doc[1] + 1.0;
            dp.a_scale = (float)((double)views[1].nblk / spanA);
            dp.b_dpb = (float)per_block_span;
        }
        dp.meta = ws_take<uint4>(ctx, dp.n_meta);
        dp.wg_sum = ws_take<uint32_t>(ctx, grid);
        dp.tomb = tomb ? tomb->d_words : nullptr;
        dp.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
        dp.out = d_out;
        dp.out_cap = cap;
        dp.d_count = d_count;
        dp.debug = nullptr;
        if (ctx->opt_debug_stamps) {
            if (!ctx->d_debug && ii2::dm_malloc_retry((void **)&ctx->d_debug, (size_t)2048 * 8 * sizeof(unsigned long long)) != hipSuccess)
                return fail(ctx, II2_ENOMEM, "debug buffer allocation failed");
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_debug, 0, (size_t)2048 * 8 * sizeof(unsigned long long), st));
            dp.debug = ctx->d_debug;
            dp.debug_expand = ctx->opt_debug_stamps == 2 ? 1u : 0u;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ii2_profile_pair(ctx, &e0, &e1);
        if (and2 && dp.lb.agg) {           // (the one-launch form waits between workgroups: one such kernel per device at a time)
            if (int rcq = ii2_lookback_launch(ctx, true, [&] { return launch_intersect_and2(dp, st, e0, e1); })) return rcq;
        } else {
            HIP_TRY(ctx, and2 ? launch_intersect_and2(dp, st, e0, e1) : launch_intersect_dense(dp, st, e0, e1));
        }
        return II2_OK;
    }
    // a tiny sparse driver (a rare term against long lists) would keep only a handful of workgroups busy, each decoding
    // one block of the long list per candidate, one after the other: split its blocks over several tiles
    uint32_t sub = 1;
    if (n >= 2 && G == 1 && (per_block_span >= 8192.0 || (nblk0 == 1 && views[n - 1].nblk >= 64)) && ctx->opt_intersect_g <= 0) {
        const uint32_t want_tiles = (ctx->opt_intersect_subtiles > 0 ? (uint32_t)ctx->opt_intersect_subtiles : 4u) * (uint32_t)ctx->cu_count;
        if (nblk0 < want_tiles) sub = std::min<uint32_t>(ctx->opt_intersect_submax > 0 ? (uint32_t)ctx->opt_intersect_submax : 16u, (want_tiles + nblk0 - 1u) / nblk0);
    }
    p.G = G;
    // the pipelined gallop pays when a driver block faces many blocks of a long list (candidates then hit distinct blocks)
    p.sparse_driver = ((per_block_span >= 8192.0 && views[n - 1].nblk / 16u >= nblk0) || sub > 1) ? 1u : 0u;
    p.sub = sub;
    p.n_tiles = ((nblk0 + G - 1) / G) * sub;
    const size_t dstride = 2 + 4 * (size_t)n;
    p.desc_words = (uint32_t)dstride;
    uint32_t slot_words = (ISECT_SMAX + 32u) / 32u;
    if (slot_words < G * 256u) slot_words = G * 256u;
    slot_words = (slot_words + 3u) & ~3u;
    p.slot_words = slot_words;
    size_t need = align_up((size_t)p.n_tiles * dstride * sizeof(uint32_t)) + align_up((size_t)p.n_tiles * slot_words * sizeof(uint32_t)) +
                  2 * align_up(((size_t)p.n_tiles + 2) * sizeof(uint32_t)) + align_up(((size_t)p.n_tiles / 64 + p.n_tiles / 4096 + 4) * sizeof(uint32_t)) + 4096;
    int rc = ii2_ws_reserve(ctx, need);
    if (rc) return rc;
    p.ranges = ws_take<uint32_t>(ctx, (size_t)p.n_tiles * dstride);
    p.tmp = ws_take<uint32_t>(ctx, (size_t)p.n_tiles * slot_words);
    p.tile_count = ws_take<uint32_t>(ctx, (size_t)p.n_tiles + 1);
    p.n_sums1 = p.n_tiles / 64 + 1;
    p.n_sums = p.n_sums1;
    p.sums = ws_take<uint32_t>(ctx, p.n_sums);
    uint64_t *d_tile_off = nullptr;
    const uint32_t wgs_default = 5u;    // LDS per workgroup: ~29 KB
    p.max_grid = (uint32_t)ctx->cu_count * (ctx->opt_intersect_wgs > 0 ? (uint32_t)ctx->opt_intersect_wgs : wgs_default);
    p.bitmap_mode = ctx->opt_intersect_bitmap ? 1u : 0u;
    // measured on 100M-doc Zipf pairs: the gallop path wins from ~32 docs per driver posting on (ranks 30/60: 80 -> 64 us),
    // the map tiles below that (ranks 10/20: 78 vs 100 us)
    p.map_docs_per_block = ctx->opt_intersect_map_docs > 0 ? (uint32_t)ctx->opt_intersect_map_docs : 8192u;
    p.tomb = tomb ? tomb->d_words : nullptr;
    p.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
    p.out = d_out;
    p.out_cap = cap;
    p.d_count = d_count;
    p.debug = nullptr;
    if (ctx->opt_debug_stamps) {
        if (!ctx->d_debug && ii2::dm_malloc_retry((void **)&ctx->d_debug, (size_t)2048 * 8 * sizeof(unsigned long long)) != hipSuccess)
            return fail(ctx, II2_ENOMEM, "debug buffer allocation failed");
        p.debug = ctx->d_debug;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ii2_profile_pair(ctx, &e0, &e1);
    HIP_TRY(ctx, launch_intersect(p, d_tile_off, st, e0, e1));
    return II2_OK;
}

// Union of lists that are dense TOGETHER (>= 1 posting per 16 docs of their common range): the byte-map tiles of
// the intersection with OR semantics over fixed doc ranges — no decode-to-raw, no fold, ~20x the merge path's rate.
int ii2_union_dense_unlocked(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx, const ii2_tomb *tomb,
                             uint32_t *d_out, uint64_t cap, uint64_t *d_count, bool *taken) {
    *taken = false;
    if (!ctx->opt_union_dense || n == 0 || n > MAX_LISTS) return II2_OK;
    hipStream_t st = ctx->stream;
    IntersectParams p;
    std::memset(&p, 0, sizeof p);
    uint32_t m = 0;
    uint64_t total_blocks = 0;
    const ii2_seg *nz_seg[MAX_LISTS];
    uint64_t nz_idx[MAX_LISTS];
    for (uint32_t i = 0; i < n; i++) {
        ListView v;
        int rc = make_list_view(ctx, segs[i], list_idx ? list_idx[i] : 0, &v);
        if (rc) return rc;
        if (v.nblk == 0) continue;
        nz_seg[m] = segs[i];
        nz_idx[m] = list_idx ? list_idx[i] : 0;
        p.lists[m++] = v;
        total_blocks += v.nblk;
    }
    if (m == 0) return II2_OK;
    if (total_blocks <= SMALL_SET_BLOCKS) {
        if (int rc = ii2_setop_small_unlocked(ctx, true, m, p.lists, nz_seg, nz_idx, tomb, d_out, cap, d_count, taken)) return rc;
        if (*taken) return II2_OK;
    }
    // a few lists of medium size: decode, rank every id by bisection in the other lists, filter, write (union_rank.hip)
    if (ctx->opt_union_rank && m <= UNION_RANK_MAXL && total_blocks <= UNION_RANK_MAX_POSTINGS / II2_DV1_BLOCK + m) {
        UnionRankParams up;
        std::memset(&up, 0, sizeof up);
        uint64_t np = 0;
        uint32_t nb = 0;
        for (uint32_t i = 0; i < m; i++) {
            if (int rc = ii2_seg_host_cnt(ctx, nz_seg[i])) return rc;
            up.lists[i] = p.lists[i];
            up.blk_base[i] = nb;
            up.lpre[i] = (uint32_t)np;
            nb += p.lists[i].nblk;
            np += nz_seg[i]->h_cnt[nz_idx[i]];
        }
        if (np <= UNION_RANK_MAX_POSTINGS) {
            up.blk_base[m] = nb;
            up.lpre[m] = (uint32_t)np;
            up.n_lists = m;
            up.n_blocks = nb;
            const uint32_t nwg = (uint32_t)((np + 2047) / 2048);
            const size_t need = 2 * align_up(np * sizeof(uint32_t)) + align_up(((size_t)nwg + 1) * sizeof(uint32_t)) + 4096;
            if (int rc = ii2_ws_reserve(ctx, need)) return rc;
            up.raw = ws_take<uint32_t>(ctx, np);
            up.sorted = ws_take<uint32_t>(ctx, np);
            up.wg_cnt = ws_take<uint32_t>(ctx, (size_t)nwg + 1);
            up.tomb = tomb ? tomb->d_words : nullptr;
            up.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
            up.out = d_out;
            up.out_cap = cap;
            up.d_count = d_count;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            ii2_profile_pair(ctx, &e0, &e1);
            HIP_TRY(ctx, launch_union_rank(up, st, e0, e1));
            *taken = true;
            return II2_OK;
        }
    }
    if (total_blocks < 64) return II2_OK;
    p.n_lists = m;
    // Few long lists, the longest one dense: the streaming kernel of the dense intersection with OR semantics — every
    // wave walks its own run of blocks of the longest list (the pacer), the other lists mark into the same bitmap
    // (intersect_dense.hip).  The lists' ends are cached per (segment, list): no host sync after the first use.
    if (ctx->opt_union_stream && m >= 2 && m <= DENSE_MAXL) {
        uint32_t pace = 0;
        for (uint32_t i = 1; i < m; i++) if (p.lists[i].nblk > p.lists[pace].nblk) pace = i;
        if (p.lists[pace].nblk >= 1024) {
            DenseParams dp;
            std::memset(&dp, 0, sizeof dp);
            uint32_t u_lo = 0xFFFFFFFFu, u_hi = 0;
            ii2_seg::ListSpan psp{};
            for (uint32_t i = 0, o = 1; i < m; i++) {
                ii2_seg::ListSpan sp;
                if (int rc = list_span(ctx, nz_seg[i], nz_idx[i], p.lists[i], &sp)) return rc;
                const uint32_t at = i == pace ? 0u : o++;
                dp.lists[at] = p.lists[i];
                dp.first_doc[at] = sp.first_doc;
                dp.last_doc[at] = sp.last_doc;
                if (i == pace) psp = sp;
                u_lo = std::min(u_lo, sp.first_doc);
                u_hi = std::max(u_hi, sp.last_doc);
            }
            const uint32_t nblk0 = dp.lists[0].nblk;
            const double per_block = (double)(psp.last_block_first_doc - psp.first_doc) / (double)(nblk0 - 1);
            // the stretches before the pacer's first and after its last doc are one wave's work each: keep them short
            const uint64_t own = (uint64_t)psp.last_doc - psp.first_doc + 1, all = (uint64_t)u_hi - u_lo + 1;
            if (per_block > 0 && per_block <= 1100.0 && all <= own + own / 4 + 65536) {
                dp.n_lists = m;
                dp.is_union = 1u;
                dp.u_lo = u_lo;
                dp.u_hi = u_hi;
                dp.bpw = 16u;
                dp.n_waves = (nblk0 + dp.bpw - 1) / dp.bpw;
                const uint32_t grid = (dp.n_waves + 3u) / 4u;
                dp.n_meta = grid * 4u;
                dp.base32 = u_lo & ~31u;
                const uint64_t bm_words = (((uint64_t)u_hi - dp.base32) >> 5) + 1 + dp.n_meta + 8;
                size_t need = align_up(bm_words * sizeof(uint32_t)) + align_up((size_t)dp.n_meta * sizeof(uint4)) + align_up((size_t)grid * sizeof(uint32_t)) + 4096;
                if (int rc = ii2_ws_reserve(ctx, need)) return rc;
                dp.bitmap = ws_take<uint32_t>(ctx, bm_words);
                dp.meta = ws_take<uint4>(ctx, dp.n_meta);
                dp.wg_sum = ws_take<uint32_t>(ctx, grid);
                dp.tomb = tomb ? tomb->d_words : nullptr;
                dp.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
                dp.out = d_out;
                dp.out_cap = cap;
                dp.d_count = d_count;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                ii2_profile_pair(ctx, &e0, &e1);
                HIP_TRY(ctx, launch_intersect_dense(dp, st, e0, e1));
                *taken = true;
                return II2_OK;
            }
        }
    }
    // the lists' common doc range from their cached ends (one round trip per list the first time it is used, none after)
    uint32_t mm[2] = {0xFFFFFFFFu, 0u};
    for (uint32_t i = 0; i < m; i++) {
        ii2_seg::ListSpan sp;
        if (int rc = list_span(ctx, nz_seg[i], nz_idx[i], p.lists[i], &sp)) return rc;
        mm[0] = std::min(mm[0], sp.first_doc);
        mm[1] = std::max(mm[1], sp.last_doc);
    }
    if (mm[1] < mm[0]) return II2_OK;
    constexpr uint32_t S = ISECT_SMAX - 64u;               // tile span: a multiple of 32 below the byte-map size
    const uint32_t base = mm[0] & ~31u;
    const uint64_t span = (uint64_t)mm[1] - base + 1;
    if (total_blocks * II2_DV1_BLOCK * (uint64_t)ctx->opt_union_sparsity < span) return II2_OK;       // too sparse (the tile count grows with the span): the merge passes do better
    const uint64_t n_tiles = (span + S - 1) / S;
    if (n_tiles >= (1ull << 24)) return II2_OK;
    p.op_union = 1u;
    p.sub = 1u;
    p.u_base = base;
    p.u_span = S;
    p.u_max = mm[1];
    p.n_tiles = (uint32_t)n_tiles;
    const size_t dstride = 2 + 4 * (size_t)m;
    p.desc_words = (uint32_t)dstride;
    p.slot_words = ((ISECT_SMAX + 32u) / 32u + 3u) & ~3u;
    size_t need = align_up((size_t)p.n_tiles * dstride * sizeof(uint32_t)) + align_up((size_t)p.n_tiles * p.slot_words * sizeof(uint32_t)) +
                  2 * align_up(((size_t)p.n_tiles + 2) * sizeof(uint32_t)) + align_up(((size_t)p.n_tiles / 64 + 4) * sizeof(uint32_t)) + 4096;
    int rc = ii2_ws_reserve(ctx, need);
    if (rc) return rc;
    p.ranges = ws_take<uint32_t>(ctx, (size_t)p.n_tiles * dstride);
    p.tmp = ws_take<uint32_t>(ctx, (size_t)p.n_tiles * p.slot_words);
    p.tile_count = ws_take<uint32_t>(ctx, (size_t)p.n_tiles + 1);
    p.n_sums1 = p.n_tiles / 64 + 1;
    p.n_sums = p.n_sums1;
    p.sums = ws_take<uint32_t>(ctx, p.n_sums);
    p.max_grid = (uint32_t)ctx->cu_count * (ctx->opt_intersect_wgs > 0 ? (uint32_t)ctx->opt_intersect_wgs : 5u);
    p.bitmap_mode = ctx->opt_intersect_bitmap ? 1u : 0u;
    p.tomb = tomb ? tomb->d_words : nullptr;
    p.tomb_nwords = tomb ? (uint32_t)std::min<uint64_t>(tomb->n_words, 0xFFFFFFFFull) : 0;
    p.out = d_out;
    p.out_cap = cap;
    p.d_count = d_count;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ii2_profile_pair(ctx, &e0, &e1);
    HIP_TRY(ctx, launch_intersect(p, nullptr, st, e0, e1));
    *taken = true;
    return II2_OK;
}

extern "C" {

int ii2_intersect_async(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx,
                        const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *d_count) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return intersect_unlocked(ctx, n, segs, list_idx, tomb, d_out, cap, d_count);
}

int ii2_intersect(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx,
                  const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *count) {
    if (!ctx || !count) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the count lands in the pinned host mailbox directly (the kernels write it once, at their end): one stream
    // synchronisation, no copy behind it
    uint64_t *d_cnt = ii2_mapped_mail(ctx, II2_MAIL_COUNT);
    int rc = intersect_unlocked(ctx, n, segs, list_idx, tomb, d_out, cap, d_cnt ? d_cnt : ctx->d_mail);
    if (rc) return rc;
    if (!d_cnt) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mail + II2_MAIL_COUNT, ctx->d_mail, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = ctx->h_mail[II2_MAIL_COUNT];
    if (*count == ~0ull && ctx->lb_pending) {
        // a bounded wait of the one-launch two-list AND ran out (its workgroups did not start in index order): nothing is
        // wrong with the inputs — the same query again through the two-kernel form, which has no inter-workgroup waits
        ctx->lb_pending = 0;
        ctx->lb_fallbacks++;
        const int64_t keep = ctx->opt_intersect_and2;
        ctx->opt_intersect_and2 = 2;
        rc = intersect_unlocked(ctx, n, segs, list_idx, tomb, d_out, cap, d_cnt ? d_cnt : ctx->d_mail);
        ctx->opt_intersect_and2 = keep;
        if (rc) return rc;
        if (!d_cnt) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mail + II2_MAIL_COUNT, ctx->d_mail, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        *count = ctx->h_mail[II2_MAIL_COUNT];
    }
    ctx->lb_pending = 0;
    if (*count > cap) return fail(ctx, II2_ECAPACITY, "ii2_intersect: result does not fit the output buffer (content unspecified)");
    return II2_OK;
}

int ii2_selftest(ii2_ctx *ctx) {
    if (!ctx) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ii2_ws_reserve(ctx, SELFTEST_SCRATCH + 4096);
    if (rc) return rc;
    uint8_t *scratch = ws_take<uint8_t>(ctx, SELFTEST_SCRATCH);
    uint32_t *d_fail = (uint32_t *)ctx->d_mail;
    HIP_TRY(ctx, hipMemsetAsync(d_fail, 0, sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, launch_selftest(d_fail, scratch, ctx->stream));
    uint32_t f = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&f, d_fail, sizeof f, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (f) {
        char msg[96];
        std::snprintf(msg, sizeof msg, "device self-test failed, mask 0x%x (1=scan 2=decode 4=count)", f);
        ctx->err = msg;
        return (int)f;
    }
    return II2_OK;
}

int ii2_ctx_counters(ii2_ctx *ctx, uint64_t *out, uint32_t n) {
    if (!ctx || !out) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (n > 0) out[0] = ctx->merge_fallbacks;
    if (n > 1) out[1] = ctx->lb_fallbacks;
    if (n > 2) out[2] = ctx->comm_syncs;
    return II2_OK;
}

int ii2_debug_read(ii2_ctx *ctx, uint64_t *out, uint64_t n_words) {
    if (!ctx || !out || !ctx->d_debug) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    const uint64_t have = (uint64_t)2048 * 8;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_debug, std::min(n_words, have) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return II2_OK;
}

int ii2_set_option(ii2_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    const std::string k(name);
    if (k == "intersect.g") ctx->opt_intersect_g = value;
    else if (k == "intersect.wgs") ctx->opt_intersect_wgs = value;
    else if (k == "merge.large_tile") ctx->opt_merge_large_tile = value;
    else if (k == "merge.bitmap_tiles") ctx->opt_merge_bitmap = value;
    else if (k == "debug.merge_skip") ctx->opt_merge_skip = value;
    else if (k == "merge.direct") ctx->opt_merge_direct = value;
    else if (k == "merge.spin") ctx->opt_merge_spin = value;
    else if (k == "debug.stamps") ctx->opt_debug_stamps = value;
    else if (k == "merge.alone") ctx->opt_merge_alone = value;
    else if (k == "debug.no_chain") ctx->opt_debug_no_chain = value;
    else if (k == "profile.events") ctx->opt_profile_events = value;
    else if (k == "intersect.bitmap") ctx->opt_intersect_bitmap = value;
    else if (k == "union.dense") ctx->opt_union_dense = value;
    else if (k == "union.stream") ctx->opt_union_stream = value;
    else if (k == "setop.small") ctx->opt_small_setop = value;
    else if (k == "union.rank") ctx->opt_union_rank = value;
    else if (k == "union.sparsity") ctx->opt_union_sparsity = value > 0 ? value : 2048;
    else if (k == "intersect.map_docs") ctx->opt_intersect_map_docs = value;
    else if (k == "intersect.dense") ctx->opt_intersect_dense = value;
    else if (k == "intersect.dense_bpw") ctx->opt_dense_bpw = value;
    else if (k == "intersect.subtiles") ctx->opt_intersect_subtiles = value;
    else if (k == "intersect.submax") ctx->opt_intersect_submax = value;
    else if (k == "intersect.and2") ctx->opt_intersect_and2 = value;      // 0: n-list kernel, 1: one launch (look-back), 2: two kernels
    else if (k == "intersect.and2_spin") ctx->opt_and2_spin = value;
    else if (k == "encode.stream") ctx->opt_encode_stream = value;        // 1: one-pass encoder behind a merge (default), 0: two-pass, -1: forced fallback (tests)
    else return fail(ctx, II2_EINVAL, "unknown option");
    return II2_OK;
}

}  // extern "C"
