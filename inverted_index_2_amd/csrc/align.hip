// align.hip — term alignment on the device.
//
// Replaces the k-way walk over the segments' term dictionaries that feeds the merging iterator (reference
// shard.go:253-278 / file/reader.go:33-71, ordered by file.CompareTermValues = bytes.Compare, file/types.go:24-26):
// given k sorted, duplicate-free dictionaries it produces the union dictionary and, per segment, which of its lists
// holds each union term — what the host used to compute with a std::sort over strings plus one pass per segment.
//
// Method: an LSD radix sort over fixed-width chunks.  A term is cut into 8-byte chunks, zero padded; comparing
// (chunk_0, chunk_1, ..., chunk_{m-1}, length) lexicographically with big-endian chunk values IS bytes.Compare
// (a shorter term that is a prefix of a longer one has equal chunks up to the padding and the smaller length).  So:
// stable sort by length (only when the lengths differ), then by the last chunk, ..., then by the first.  Equal terms
// (one per segment at most) end up adjacent; heads of runs number the union terms.  The sorts are hipcub radix sorts
// (plumbing, like the offset scans); keys are extracted by a kernel per pass.  8-byte big-endian term ids — the
// synthetic configs — take exactly one sort.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <new>

#include "internal.h"

namespace ii2 {

__device__ __forceinline__ uint32_t seg_of_term(const uint64_t *__restrict__ seg_first, uint32_t k, uint64_t g) {
    uint32_t lo = 0, hi = k;            // seg_first[lo] <= g < seg_first[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg_first[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_align_iota(uint32_t *__restrict__ perm, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)i;
}

// key of term perm[i] for one pass: its length (chunk < 0) or its big-endian 8-byte chunk `chunk`, zero padded
__global__ void k_align_keys(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, const uint32_t *__restrict__ perm, uint64_t n,
                             int chunk, uint64_t *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = perm[i];
    const uint64_t b = off[g], len = off[g + 1] - b;
    if (chunk < 0) { keys[i] = len; return; }
    const uint64_t c0 = 8ull * (uint64_t)chunk;
    uint64_t key = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        key <<= 8;
        if (c0 + (uint64_t)j < len) key |= bytes[b + c0 + (uint64_t)j];
    }
    keys[i] = key;
}

// head[i] = 1 when sorted term i differs from sorted term i - 1
__global__ void k_align_heads(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, const uint32_t *__restrict__ perm, uint64_t n,
                              uint32_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) { head[0] = 1u; return; }
    const uint32_t a = perm[i], b = perm[i - 1];
    const uint64_t oa = off[a], la = off[a + 1] - oa, ob = off[b], lb = off[b + 1] - ob;
    bool same = la == lb;
    for (uint64_t j = 0; same && j < la; j++) same = bytes[oa + j] == bytes[ob + j];
    head[i] = same ? 0u : 1u;
}

// uidx[g] = union index of input term g; rep[u] = one input term equal to union term u; sel[s][u] = local list of segment s
__global__ void k_align_scatter(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hpre, uint64_t n,
                                const uint64_t *__restrict__ seg_first, uint32_t k, uint64_t n_union, uint32_t *__restrict__ uidx,
                                uint32_t *__restrict__ rep, int32_t *__restrict__ sel) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = perm[i];
    const uint32_t u = hpre[i] + head[i] - 1u;          // hpre = exclusive scan of head
    uidx[g] = u;
    if (head[i]) rep[u] = g;
    const uint32_t s = seg_of_term(seg_first, k, g);
    sel[(uint64_t)s * n_union + u] = (int32_t)(g - seg_first[s]);
}

// ---- the aligned view of one segment, built from the alignment on the device ----
// flags[u] = 1 when segment s holds union term u
__global__ void k_align_flags(const int32_t *__restrict__ sel_row, uint64_t n_union, uint32_t *__restrict__ flags) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u <= n_union) flags[u] = (u < n_union && sel_row[u] >= 0) ? 1u : 0u;
}
// slot u of the view: the list first_list + sel[u] of the source, or an empty slot positioned before the next selected list
__global__ void k_align_view(const int32_t *__restrict__ sel_row, const uint32_t *__restrict__ before, uint64_t n_union, uint64_t first_list,
                             const uint32_t *__restrict__ src_blk_off, const uint32_t *__restrict__ src_cnt, const uint32_t *__restrict__ src_last,
                             uint32_t *__restrict__ blk_off, uint32_t *__restrict__ cnt, uint32_t *__restrict__ last_doc) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u > n_union) return;
    blk_off[u] = src_blk_off[first_list + before[u]];       // before[u] = selected slots before u: the lists of the dictionary are consecutive
    if (u == n_union) return;
    const int32_t j = sel_row[u];
    cnt[u] = j >= 0 ? src_cnt[first_list + (uint64_t)j] : 0u;
    last_doc[u] = j >= 0 ? src_last[first_list + (uint64_t)j] : 0u;
}
// blk_list of the view: the slot that owns each block of the store (0xFFFFFFFF: none of this view's lists)
__global__ void k_align_blk_list(const uint32_t *__restrict__ src_blk_list, uint64_t n_blocks, uint64_t first_list, uint64_t n_dict,
                                 const uint32_t *__restrict__ uidx_seg, uint32_t *__restrict__ blk_list) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t li = src_blk_list[b];
    uint32_t v = 0xFFFFFFFFu;
    if (li != 0xFFFFFFFFu && li >= first_list && (uint64_t)li - first_list < n_dict) v = uidx_seg[(uint64_t)li - first_list];
    blk_list[b] = v;
}

static unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace ii2

using namespace ii2;

struct ii2_align {
    int device = 0;
    uint32_t k = 0;
    uint64_t n_all = 0, n_union = 0;
    std::vector<uint64_t> seg_first;      // host copy [k + 1]
    uint32_t *d_uidx = nullptr;           // [n_all] union index of every input term
    uint32_t *d_rep = nullptr;            // [n_union]
    int32_t *d_sel = nullptr;             // [k][n_union]
    ~ii2_align() {
        if (d_uidx) (void)hipFree(d_uidx);
        if (d_rep) (void)hipFree(d_rep);
        if (d_sel) (void)hipFree(d_sel);
    }
};

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
int ii2_ws_reserve(ii2_ctx *ctx, size_t bytes);          // api.cpp
int ii2_seg_adopt_view(ii2_ctx *ctx, const ii2_seg *src, uint64_t n_out, uint32_t *d_blk_off, uint32_t *d_cnt, uint32_t *d_last_doc,
                       uint32_t *d_blk_list, ii2_seg **out);   // api.cpp

extern "C" {

int ii2_align_terms(ii2_ctx *ctx, uint32_t k, const uint8_t *term_bytes, const uint64_t *term_off, const uint64_t *seg_first, ii2_align **out) {
    if (!ctx || !out || !term_off || !seg_first || k == 0 || k > II2_MAX_LISTS) return fail(ctx, II2_EINVAL, "ii2_align_terms: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    hipStream_t st = ctx->stream;
    const uint64_t n = seg_first[k];
    if (seg_first[0] != 0) return fail(ctx, II2_EINVAL, "ii2_align_terms: seg_first[0] must be 0");
    for (uint32_t s = 0; s < k; s++)
        if (seg_first[s + 1] < seg_first[s]) return fail(ctx, II2_EINVAL, "ii2_align_terms: seg_first must be non-decreasing");
    if (n >= (1ull << 31)) return fail(ctx, II2_ERANGE, "ii2_align_terms: too many terms");
    uint64_t maxlen = 0, minlen = ~0ull;
    for (uint64_t i = 0; i < n; i++) {
        if (term_off[i + 1] < term_off[i]) return fail(ctx, II2_EINVAL, "ii2_align_terms: term_off must be non-decreasing");
        const uint64_t l = term_off[i + 1] - term_off[i];
        maxlen = std::max(maxlen, l);
        minlen = std::min(minlen, l);
    }
    const uint64_t nbytes = n ? term_off[n] : 0;
    if (nbytes && !term_bytes) return fail(ctx, II2_EINVAL, "ii2_align_terms: term_bytes is NULL");
    std::unique_ptr<ii2_align> a(new (std::nothrow) ii2_align());
    if (!a) return II2_ENOMEM;
    a->device = ctx->device;
    a->k = k;
    a->n_all = n;
    a->seg_first.assign(seg_first, seg_first + k + 1);
    if (n == 0) { *out = a.release(); return II2_OK; }

    size_t sort_b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                       (int)n, 0, 64, (hipStream_t)0);
    const size_t scan_b = scan_temp_bytes((size_t)n + 1);
    const size_t tmp_b = align_up(std::max(sort_b, scan_b));
    const size_t need = align_up(nbytes + 16) + align_up((n + 1) * sizeof(uint64_t)) + align_up((k + 1) * sizeof(uint64_t)) + 2 * align_up(n * sizeof(uint64_t)) +
                        2 * align_up(n * sizeof(uint32_t)) + 2 * align_up((n + 1) * sizeof(uint32_t)) + tmp_b + 4096;
    int rc = ii2_ws_reserve(ctx, need);
    if (rc) return rc;
    uint8_t *cur = ctx->ws;
    auto carve = [&](size_t bytes) { uint8_t *q = cur; cur += align_up(bytes); return q; };
    uint8_t *d_bytes = carve(nbytes + 16);
    uint64_t *d_off = (uint64_t *)carve((n + 1) * sizeof(uint64_t));
    uint64_t *d_first = (uint64_t *)carve((k + 1) * sizeof(uint64_t));
    uint64_t *d_key[2] = {(uint64_t *)carve(n * sizeof(uint64_t)), (uint64_t *)carve(n * sizeof(uint64_t))};
    uint32_t *d_perm[2] = {(uint32_t *)carve(n * sizeof(uint32_t)), (uint32_t *)carve(n * sizeof(uint32_t))};
    uint32_t *d_head = (uint32_t *)carve((n + 1) * sizeof(uint32_t));
    uint32_t *d_hpre = (uint32_t *)carve((n + 1) * sizeof(uint32_t));
    void *d_tmp = carve(tmp_b);
    if (nbytes) HIP_TRY(ctx, hipMemcpyAsync(d_bytes, term_bytes, nbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_off, term_off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_first, seg_first, (k + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_align_iota, dim3(grid_for(n)), dim3(256), 0, st, d_perm[0], n);
    int cb = 0;                                        // current buffer of perm
    auto sort_pass = [&](int chunk) -> hipError_t {
        hipLaunchKernelGGL(k_align_keys, dim3(grid_for(n)), dim3(256), 0, st, (const uint8_t *)d_bytes, (const uint64_t *)d_off, (const uint32_t *)d_perm[cb], n, chunk, d_key[0]);
        size_t tb = tmp_b;
        const int end_bit = chunk < 0 ? 32 : 64;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, (const uint64_t *)d_key[0], d_key[1], (const uint32_t *)d_perm[cb], d_perm[cb ^ 1], (int)n, 0, end_bit, st);
        cb ^= 1;
        return e;
    };
    const int chunks = (int)((maxlen + 7) / 8);
    if (minlen != maxlen) HIP_TRY(ctx, sort_pass(-1));          // least significant: the length
    for (int c = chunks - 1; c >= 0; c--) HIP_TRY(ctx, sort_pass(c));
    // (all terms empty: chunks == 0 and the order is the input order — they are all equal)
    const uint32_t *perm = d_perm[cb];
    hipLaunchKernelGGL(k_align_heads, dim3(grid_for(n)), dim3(256), 0, st, (const uint8_t *)d_bytes, (const uint64_t *)d_off, perm, n, d_head);
    HIP_TRY(ctx, hipMemsetAsync(d_head + n, 0, sizeof(uint32_t), st));
    HIP_TRY(ctx, scan_excl_u32(d_tmp, tmp_b, d_head, d_hpre, (size_t)n + 1, st));
    uint32_t n_union = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_union, d_hpre + n, sizeof n_union, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));                       // the one round trip: the union's size decides the result arrays
    a->n_union = n_union;
    if (hipMalloc((void **)&a->d_uidx, n * sizeof(uint32_t)) != hipSuccess || hipMalloc((void **)&a->d_rep, (size_t)n_union * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&a->d_sel, (size_t)k * n_union * sizeof(int32_t)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "ii2_align_terms: result allocation failed");
    HIP_TRY(ctx, hipMemsetAsync(a->d_sel, 0xFF, (size_t)k * n_union * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_align_scatter, dim3(grid_for(n)), dim3(256), 0, st, perm, (const uint32_t *)d_head, (const uint32_t *)d_hpre, n, (const uint64_t *)d_first, k,
                       (uint64_t)n_union, a->d_uidx, a->d_rep, a->d_sel);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));                       // the workspace may be reused by the next call
    *out = a.release();
    return II2_OK;
}

int ii2_align_info(const ii2_align *a, uint64_t *n_union, uint32_t *k) {
    if (!a) return II2_EINVAL;
    if (n_union) *n_union = a->n_union;
    if (k) *k = a->k;
    return II2_OK;
}

int ii2_align_export(ii2_ctx *ctx, const ii2_align *a, uint64_t *rep, int64_t *src_list) {
    if (!ctx || !a || a->device != ctx->device) return fail(ctx, II2_EINVAL, "ii2_align_export: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (a->n_union == 0) return II2_OK;
    if (rep) {
        std::vector<uint32_t> h(a->n_union);
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), a->d_rep, a->n_union * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint64_t u = 0; u < a->n_union; u++) rep[u] = h[u];
    }
    if (src_list) {
        std::vector<int32_t> h((size_t)a->k * a->n_union);
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), a->d_sel, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < h.size(); i++) src_list[i] = h[i];
    }
    return II2_OK;
}

int ii2_seg_select_aligned(ii2_ctx *ctx, const ii2_seg *src, const ii2_align *a, uint32_t s, uint64_t first_list, ii2_seg **out) {
    if (!ctx || !src || !a || !out || s >= a->k || src->device != ctx->device || a->device != ctx->device)
        return fail(ctx, II2_EINVAL, "ii2_seg_select_aligned: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    const uint64_t n_dict = a->seg_first[s + 1] - a->seg_first[s];
    if (first_list + n_dict > src->n_lists) return fail(ctx, II2_EINVAL, "ii2_seg_select_aligned: the dictionary does not fit the segment's lists");
    hipStream_t st = ctx->stream;
    const uint64_t nu = a->n_union;
    const size_t scan_b = scan_temp_bytes((size_t)nu + 1);
    int rc = ii2_ws_reserve(ctx, 2 * align_up((nu + 1) * sizeof(uint32_t)) + scan_b + 4096);
    if (rc) return rc;
    uint8_t *cur = ctx->ws;
    uint32_t *d_flags = (uint32_t *)cur; cur += align_up((nu + 1) * sizeof(uint32_t));
    uint32_t *d_before = (uint32_t *)cur; cur += align_up((nu + 1) * sizeof(uint32_t));
    void *d_tmp = cur;
    // one allocation for the view's four arrays
    uint32_t *d_blk_off = nullptr, *d_cnt = nullptr, *d_last = nullptr, *d_blk_list = nullptr;
    if (hipMalloc((void **)&d_blk_off, (nu + 1) * sizeof(uint32_t)) != hipSuccess || hipMalloc((void **)&d_cnt, (nu + 1) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&d_last, (nu + 1) * sizeof(uint32_t)) != hipSuccess || hipMalloc((void **)&d_blk_list, (src->n_blocks + 1) * sizeof(uint32_t)) != hipSuccess) {
        if (d_blk_off) (void)hipFree(d_blk_off);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_last) (void)hipFree(d_last);
        if (d_blk_list) (void)hipFree(d_blk_list);
        return fail(ctx, II2_ENOMEM, "segment allocation failed");
    }
    const int32_t *sel_row = a->d_sel + (size_t)s * nu;
    hipLaunchKernelGGL(k_align_flags, dim3(grid_for(nu + 1)), dim3(256), 0, st, sel_row, nu, d_flags);
    hipError_t e = scan_excl_u32(d_tmp, scan_b, d_flags, d_before, (size_t)nu + 1, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_align_view, dim3(grid_for(nu + 1)), dim3(256), 0, st, sel_row, (const uint32_t *)d_before, nu, first_list,
                           (const uint32_t *)src->d_blk_off, (const uint32_t *)src->d_cnt, (const uint32_t *)src->d_last_doc, d_blk_off, d_cnt, d_last);
        if (src->n_blocks)
            hipLaunchKernelGGL(k_align_blk_list, dim3(grid_for(src->n_blocks)), dim3(256), 0, st, (const uint32_t *)src->d_blk_list, src->n_blocks, first_list, n_dict,
                               (const uint32_t *)(a->d_uidx + a->seg_first[s]), d_blk_list);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(d_blk_off); (void)hipFree(d_cnt); (void)hipFree(d_last); (void)hipFree(d_blk_list);
        ctx->err = std::string("ii2_seg_select_aligned: ") + hipGetErrorString(e);
        return II2_EHIP;
    }
    return ii2_seg_adopt_view(ctx, src, nu, d_blk_off, d_cnt, d_last, d_blk_list, out);
}

void ii2_align_free(ii2_align *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    delete a;
}

}  // extern "C"
