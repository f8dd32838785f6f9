// align.hip — term alignment on the device, hand-written, on dictionaries that live in HBM.
//
// Replaces the k-way walk over the segments' term dictionaries that feeds the merging iterator (reference
// shard.go:253-278 / file/reader.go:33-71, ordered by file.CompareTermValues = bytes.Compare, file/types.go:24-26):
// given k sorted, duplicate-free dictionaries it produces the union dictionary and, per segment, which of its lists
// holds each union term.
//
// Method (round 3: no sort, no library, no per-call upload, no host pass over the terms): RANKING.  The k dictionaries
// are sorted already, so the place of term x of dictionary s in the k-way merge is
//     its index in s  +  sum over s' < s of |{y in s' : y <= x}|  +  sum over s' > s of |{y in s' : y < x}|
// (equal terms in dictionary order: a permutation of 0 .. n-1).  A workgroup takes 1024 consecutive terms of one
// dictionary (four per thread); where that stretch begins and ends in every other dictionary is found once per workgroup (2 k bisections,
// side by side in 2 k threads), and every term then bisects only inside those brackets — a few steps in cached data.  x
// opens a run of equal terms unless an earlier dictionary holds it; the run heads, scattered to their places and
// summed by prefix, number the union terms.  Comparison = bytes.Compare: a dictionary keeps, next to its bytes, every
// term's first 8 bytes as a big-endian integer (zero padded) — most comparisons end there; ties go on through the bytes
// and end with the lengths.
#include <algorithm>
#include <cstring>
#include <new>

#include "dv1_device.h"
#include "internal.h"

// a term dictionary resident in HBM: sorted, duplicate-free terms of one segment
struct ii2_dict {
    int device = 0;
    uint64_t n = 0, n_bytes = 0;
    uint8_t *d_bytes = nullptr;           // [n_bytes + 16]
    uint64_t *d_off = nullptr;            // [n + 1] byte offsets
    uint64_t *d_key = nullptr;            // [n] first 8 bytes, big-endian, zero padded
    uint32_t long_terms = 0;              // 1: some term is longer than 8 bytes (a tie of the keys does not decide)
    ~ii2_dict() {
        if (d_bytes) (void)hipFree(d_bytes);
        if (d_off) (void)hipFree(d_off);
        if (d_key) (void)hipFree(d_key);
    }
};

namespace ii2 {

constexpr uint32_t AL_THREADS = 256;      // threads per workgroup of the ranking kernel
constexpr uint32_t AL_T = 1024;           // terms per workgroup (the brackets in the other dictionaries are searched once per workgroup)

struct AlignDicts {
    const uint64_t *key[MAX_LISTS];
    const uint64_t *off[MAX_LISTS];
    const uint8_t *bytes[MAX_LISTS];
    uint32_t n[MAX_LISTS];
    uint32_t first[MAX_LISTS + 1];        // global number of each dictionary's first term
    uint32_t wg_first[MAX_LISTS + 1];     // first workgroup of each dictionary
    uint32_t k;
    uint32_t long_terms;                  // some dictionary has terms longer than 8 bytes
};

// keys of a dictionary + checks: offsets non-decreasing, terms strictly ascending (bad |= 1 / 2), any term longer than 8 bytes (bad |= 4)
__global__ void k_dict_keys(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ off, uint64_t n, uint64_t n_bytes, uint64_t *__restrict__ key,
                            uint32_t *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = off[i], e = off[i + 1];
    if (e < b || e > n_bytes) { atomicOr(bad, 1u); key[i] = 0; return; }
    const uint64_t len = e - b;
    uint64_t kk = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        kk <<= 8;
        if ((uint64_t)j < len) kk |= bytes[b + (uint64_t)j];
    }
    key[i] = kk;
    if (len > 8) atomicOr(bad, 4u);
}

// bytes.Compare of term (ka, bytes a[0 .. la)) with term (kb, b[0 .. lb)): -1 / 0 / 1.  The keys decide unless they tie.
__device__ __forceinline__ int term_cmp(uint64_t ka, const uint8_t *__restrict__ a, uint64_t la, uint64_t kb, const uint8_t *__restrict__ b, uint64_t lb, bool long_terms) {
    if (ka != kb) return ka < kb ? -1 : 1;
    if (long_terms) {
        const uint64_t m = la < lb ? la : lb;
        for (uint64_t j = 8; j < m; j++) {
            const uint8_t x = a[j], y = b[j];
            if (x != y) return x < y ? -1 : 1;
        }
    }
    return la == lb ? 0 : (la < lb ? -1 : 1);      // (equal keys, one a prefix of the other — also decides the zero padding)
}

struct TermRef { uint64_t key; const uint8_t *p; uint64_t len; };
__device__ __forceinline__ TermRef term_of(const AlignDicts &d, uint32_t s, uint32_t i) {
    const uint64_t b = d.off[s][i];
    return TermRef{d.key[s][i], d.bytes[s] + b, d.off[s][i + 1] - b};
}
// first index in [lo, hi) of dictionary s whose term is > x (upper = true) or >= x (upper = false)
__device__ __forceinline__ uint32_t bound_in(const AlignDicts &d, uint32_t s, uint32_t lo, uint32_t hi, const TermRef &x, bool upper) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const TermRef y = term_of(d, s, mid);
        const int c = term_cmp(y.key, y.p, y.len, x.key, x.p, x.len, d.long_terms != 0u);
        if (c < 0 || (upper && c == 0)) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// pos[g] = place of input term g in the k-way merge; head[place] = 1 when the term opens a run of equal terms.
// A thread owns AL_PER consecutive terms.  For every other dictionary the workgroup's bracket of it (keys and lengths)
// is staged in LDS with coalesced loads — bisecting in global memory would cost one cache line per lane and step — and
// a thread bisects there once, for its first term, then walks on for the following ones (they ascend).
constexpr uint32_t AL_PER = AL_T / AL_THREADS;        // 4
constexpr uint32_t AL_LCAP = 2048;                    // bracket entries staged per dictionary (larger brackets: bisection in global memory)
__global__ __launch_bounds__(AL_THREADS) void k_align_rank(AlignDicts d, uint32_t *__restrict__ pos, uint32_t *__restrict__ head) {
    __shared__ uint32_t br[2][MAX_LISTS];       // where the workgroup's stretch of terms begins / ends in every dictionary
    __shared__ uint64_t SK[AL_LCAP];
    __shared__ uint32_t SL[AL_LCAP];
    uint32_t s = 0;
    {
        uint32_t lo = 0, hi = d.k;              // wg_first[lo] <= blockIdx.x < wg_first[hi]
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (d.wg_first[mid] <= blockIdx.x) lo = mid; else hi = mid; }
        s = lo;
    }
    const uint32_t i0 = (blockIdx.x - d.wg_first[s]) * AL_T;
    const uint32_t i1 = i0 + AL_T < d.n[s] ? i0 + AL_T : d.n[s];
    const bool lt = d.long_terms != 0u;
    if (threadIdx.x < 2u * d.k) {
        const uint32_t sp = threadIdx.x >> 1, which = threadIdx.x & 1u;
        // lower bracket: terms of sp before the stretch's first term; upper bracket: up to and including its last term
        const TermRef x = term_of(d, s, which ? i1 - 1u : i0);
        br[which][sp] = bound_in(d, sp, 0u, d.n[sp], x, which != 0u);
    }
    // my terms
    TermRef x[AL_PER];
    uint32_t place[AL_PER];
    bool is_head[AL_PER], valid[AL_PER];
    const uint32_t ib = i0 + AL_PER * threadIdx.x;
#pragma unroll
    for (uint32_t j = 0; j < AL_PER; j++) {
        valid[j] = ib + j < i1;
        x[j] = valid[j] ? term_of(d, s, ib + j) : TermRef{0ull, nullptr, 0ull};
        place[j] = ib + j;
        is_head[j] = true;
    }
    for (uint32_t sp = 0; sp < d.k; sp++) {
        if (sp == s) continue;
        __syncthreads();                        // (first round: the brackets are written; later: the staged bracket is free again)
        const uint32_t lo = br[0][sp], hi = br[1][sp], nb = hi - lo;
        const bool upper = sp < s;              // sp < s: count the terms <= x; sp > s: the terms < x
        if (nb <= AL_LCAP) {
            for (uint32_t t = threadIdx.x; t < nb; t += AL_THREADS) {
                SK[t] = d.key[sp][lo + t];
                SL[t] = (uint32_t)(d.off[sp][lo + t + 1u] - d.off[sp][lo + t]);
            }
            __syncthreads();
            // y_t before x ?  (y < x, or y <= x when upper)
            auto before = [&](uint32_t t, const TermRef &xx) -> bool {
                const uint64_t ky = SK[t];
                if (ky != xx.key) return ky < xx.key;
                int c;
                if (lt) { const TermRef y = term_of(d, sp, lo + t); c = term_cmp(y.key, y.p, y.len, xx.key, xx.p, xx.len, true); }
                else c = SL[t] == (uint32_t)xx.len ? 0 : (SL[t] < (uint32_t)xx.len ? -1 : 1);
                return c < 0 || (upper && c == 0);
            };
            uint32_t t = 0;
#pragma unroll
            for (uint32_t j = 0; j < AL_PER; j++) {
                if (!valid[j]) continue;
                if (j == 0u || (t + 8u < nb && before(t + 8u, x[j]))) {       // far ahead (or the first term): bisection in [t, nb)
                    uint32_t a = t, e = nb;
                    while (a < e) { const uint32_t m = a + ((e - a) >> 1); if (before(m, x[j])) a = m + 1u; else e = m; }
                    t = a;
                } else {
                    while (t < nb && before(t, x[j])) t++;
                }
                place[j] += lo + t;
                if (upper && lo + t > 0u) {                                    // does the earlier dictionary hold x itself?  (the entry before the bound)
                    bool same;
                    if (t > 0u && !lt) same = SK[t - 1u] == x[j].key && SL[t - 1u] == (uint32_t)x[j].len;
                    else { const TermRef y = term_of(d, sp, lo + t - 1u); same = term_cmp(y.key, y.p, y.len, x[j].key, x[j].p, x[j].len, lt) == 0; }
                    if (same) is_head[j] = false;
                }
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < AL_PER; j++) {
                if (!valid[j]) continue;
                const uint32_t bd = bound_in(d, sp, lo, hi, x[j], upper);
                place[j] += bd;
                if (upper && bd > 0u) {
                    const TermRef y = term_of(d, sp, bd - 1u);
                    if (term_cmp(y.key, y.p, y.len, x[j].key, x[j].p, x[j].len, lt) == 0) is_head[j] = false;
                }
            }
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < AL_PER; j++) {
        if (!valid[j]) continue;
        pos[d.first[s] + ib + j] = place[j];
        head[place[j]] = is_head[j] ? 1u : 0u;
    }
}

// uidx[g] = union index of input term g; rep[u] = one input term equal to union term u; sel[s][u] = local list of segment s
__global__ void k_align_number(AlignDicts d, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ head, const uint32_t *__restrict__ hpre,
                               uint64_t n, uint64_t n_union, uint32_t *__restrict__ uidx, uint32_t *__restrict__ rep, int32_t *__restrict__ sel) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t pl = pos[g];
    const uint32_t u = hpre[pl] + head[pl] - 1u;        // hpre = exclusive scan of head: heads up to and including my run's
    uidx[g] = u;
    if (head[pl]) rep[u] = (uint32_t)g;
    uint32_t lo = 0, hi = d.k;                          // first[lo] <= g < first[hi]
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (d.first[mid] <= g) lo = mid; else hi = mid; }
    sel[(uint64_t)lo * n_union + u] = (int32_t)(g - d.first[lo]);
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

static int dict_create_unlocked(ii2_ctx *ctx, const uint8_t *term_bytes, const uint64_t *term_off, uint64_t n, int where, ii2_dict **out) {
    hipStream_t st = ctx->stream;
    if (n >= (1ull << 31)) return fail(ctx, II2_ERANGE, "ii2_dict_create: 2^31 or more terms");
    std::unique_ptr<ii2_dict> d(new (std::nothrow) ii2_dict());
    if (!d) return II2_ENOMEM;
    d->device = ctx->device;
    d->n = n;
    const hipMemcpyKind kind = where == II2_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (ii2::dm_malloc_retry((void **)&d->d_off, (n + 1) * sizeof(uint64_t)) != hipSuccess || ii2::dm_malloc_retry((void **)&d->d_key, (n + 1) * sizeof(uint64_t)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "ii2_dict_create: allocation failed");
    uint64_t ends[2] = {0, 0};            // term_off[0], term_off[n]
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(d->d_off, term_off, (n + 1) * sizeof(uint64_t), kind, st));
        if (where == II2_HOST) { ends[0] = term_off[0]; ends[1] = term_off[n]; }
        else {
            HIP_TRY(ctx, hipMemcpyAsync(&ends[0], term_off, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipMemcpyAsync(&ends[1], term_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
    } else HIP_TRY(ctx, hipMemsetAsync(d->d_off, 0, sizeof(uint64_t), st));
    if (ends[0] != 0) return fail(ctx, II2_EINVAL, "ii2_dict_create: term_off[0] must be 0");
    d->n_bytes = ends[1];
    if (d->n_bytes && !term_bytes) return fail(ctx, II2_EINVAL, "ii2_dict_create: term_bytes is NULL");
    if (ii2::dm_malloc_retry((void **)&d->d_bytes, d->n_bytes + 16) != hipSuccess) return fail(ctx, II2_ENOMEM, "ii2_dict_create: allocation failed");
    if (d->n_bytes) HIP_TRY(ctx, hipMemcpyAsync(d->d_bytes, term_bytes, d->n_bytes, kind, st));
    uint32_t *d_bad = (uint32_t *)ctx->d_mail;
    uint32_t bad = 0;
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), st));
    if (n) hipLaunchKernelGGL(k_dict_keys, dim3(grid_for(n)), dim3(256), 0, st, (const uint8_t *)d->d_bytes, (const uint64_t *)d->d_off, n, d->n_bytes, d->d_key, d_bad);
    HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (bad & 1u) return fail(ctx, II2_EINVAL, "ii2_dict_create: term_off must be non-decreasing and end at the bytes' length");
    d->long_terms = (bad & 4u) ? 1u : 0u;
    *out = d.release();
    return II2_OK;
}

static int align_dicts_unlocked(ii2_ctx *ctx, uint32_t k, const ii2_dict *const *dicts, ii2_align **out) {
    hipStream_t st = ctx->stream;
    AlignDicts ad;
    std::memset(&ad, 0, sizeof ad);
    ad.k = k;
    uint64_t n = 0, wgs = 0;
    std::unique_ptr<ii2_align> a(new (std::nothrow) ii2_align());
    if (!a) return II2_ENOMEM;
    a->device = ctx->device;
    a->k = k;
    a->seg_first.assign(k + 1, 0);
    for (uint32_t s = 0; s < k; s++) {
        if (!dicts[s] || dicts[s]->device != ctx->device) return fail(ctx, II2_EINVAL, "ii2_align_dicts: a dictionary is NULL or lives on another device");
        ad.key[s] = dicts[s]->d_key; ad.off[s] = dicts[s]->d_off; ad.bytes[s] = dicts[s]->d_bytes;
        ad.n[s] = (uint32_t)dicts[s]->n;
        ad.first[s] = (uint32_t)n;
        ad.wg_first[s] = (uint32_t)wgs;
        ad.long_terms |= dicts[s]->long_terms;
        a->seg_first[s] = n;
        n += dicts[s]->n;
        wgs += (dicts[s]->n + AL_T - 1) / AL_T;
    }
    if (n >= (1ull << 31) || wgs >= (1ull << 31)) return fail(ctx, II2_ERANGE, "ii2_align_dicts: 2^31 or more terms");
    ad.first[k] = (uint32_t)n;
    ad.wg_first[k] = (uint32_t)wgs;
    a->seg_first[k] = n;
    a->n_all = n;
    if (n == 0) { *out = a.release(); return II2_OK; }
    const size_t scan_b = scan_temp_bytes((size_t)n + 1);
    int rc = ii2_ws_reserve(ctx, 3 * align_up((n + 1) * sizeof(uint32_t)) + scan_b + 4096);
    if (rc) return rc;
    uint8_t *cur = ctx->ws;
    auto carve = [&](size_t bytes) { uint8_t *q = cur; cur += align_up(bytes); return q; };
    uint32_t *d_pos = (uint32_t *)carve((n + 1) * sizeof(uint32_t));
    uint32_t *d_head = (uint32_t *)carve((n + 1) * sizeof(uint32_t));
    uint32_t *d_hpre = (uint32_t *)carve((n + 1) * sizeof(uint32_t));
    void *d_tmp = carve(scan_b);
    HIP_TRY(ctx, hipMemsetAsync(d_head, 0, (n + 1) * sizeof(uint32_t), st));      // (dictionaries that break the contract leave places unwritten)
    hipLaunchKernelGGL(k_align_rank, dim3((unsigned)wgs), dim3(AL_THREADS), 0, st, ad, d_pos, d_head);
    HIP_TRY(ctx, scan_excl_u32(d_tmp, scan_b, d_head, d_hpre, (size_t)n + 1, st));
    uint32_t n_union = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_union, d_hpre + n, sizeof n_union, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));                       // the one round trip: the union's size decides the result arrays
    a->n_union = n_union;
    if (ii2::dm_malloc_retry((void **)&a->d_uidx, n * sizeof(uint32_t)) != hipSuccess || ii2::dm_malloc_retry((void **)&a->d_rep, ((size_t)n_union + 1) * sizeof(uint32_t)) != hipSuccess ||
        ii2::dm_malloc_retry((void **)&a->d_sel, ((size_t)k * n_union + 1) * sizeof(int32_t)) != hipSuccess)
        return fail(ctx, II2_ENOMEM, "ii2_align_dicts: result allocation failed");
    HIP_TRY(ctx, hipMemsetAsync(a->d_sel, 0xFF, (size_t)k * n_union * sizeof(int32_t), st));
    if (n_union) hipLaunchKernelGGL(k_align_number, dim3(grid_for(n)), dim3(256), 0, st, ad, (const uint32_t *)d_pos, (const uint32_t *)d_head, (const uint32_t *)d_hpre, n,
                                    (uint64_t)n_union, a->d_uidx, a->d_rep, a->d_sel);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));                       // the workspace may be reused by the next call
    *out = a.release();
    return II2_OK;
}

int ii2_dict_create(ii2_ctx *ctx, const uint8_t *term_bytes, const uint64_t *term_off, uint64_t n_terms, int where, ii2_dict **out) {
    if (!ctx || !out || (n_terms && !term_off)) return fail(ctx, II2_EINVAL, "ii2_dict_create: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    return dict_create_unlocked(ctx, term_bytes, term_off, n_terms, where, out);
}

void ii2_dict_free(ii2_dict *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    delete d;
}

int ii2_align_dicts(ii2_ctx *ctx, uint32_t k, const ii2_dict *const *dicts, ii2_align **out) {
    if (!ctx || !out || !dicts || k == 0 || k > II2_MAX_LISTS) return fail(ctx, II2_EINVAL, "ii2_align_dicts: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    return align_dicts_unlocked(ctx, k, dicts, out);
}

// the same from flat host arrays (what a caller without resident dictionaries has): k temporary dictionaries, one alignment
int ii2_align_terms(ii2_ctx *ctx, uint32_t k, const uint8_t *term_bytes, const uint64_t *term_off, const uint64_t *seg_first, ii2_align **out) {
    if (!ctx || !out || !term_off || !seg_first || k == 0 || k > II2_MAX_LISTS) return fail(ctx, II2_EINVAL, "ii2_align_terms: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    if (seg_first[0] != 0) return fail(ctx, II2_EINVAL, "ii2_align_terms: seg_first[0] must be 0");
    for (uint32_t s = 0; s < k; s++)
        if (seg_first[s + 1] < seg_first[s]) return fail(ctx, II2_EINVAL, "ii2_align_terms: seg_first must be non-decreasing");
    if (seg_first[k] >= (1ull << 31)) return fail(ctx, II2_ERANGE, "ii2_align_terms: too many terms");
    std::vector<ii2_dict *> dicts(k, nullptr);
    std::vector<uint64_t> rel;
    int rc = II2_OK;
    for (uint32_t s = 0; s < k && !rc; s++) {
        const uint64_t a0 = seg_first[s], a1 = seg_first[s + 1];
        // the dictionary's offsets start at 0: a slice of the flat table, shifted (k small host passes over the offsets of
        // one dictionary each — the flat entry point is the convenience form; resident dictionaries skip it)
        rel.resize(a1 - a0 + 1);
        bool mono = true;
        for (uint64_t i = a0; i <= a1; i++) { rel[i - a0] = term_off[i] - term_off[a0]; mono &= i == a0 || term_off[i] >= term_off[i - 1]; }
        if (!mono) { rc = fail(ctx, II2_EINVAL, "ii2_align_terms: term_off must be non-decreasing"); break; }
        rc = dict_create_unlocked(ctx, term_bytes ? term_bytes + term_off[a0] : nullptr, rel.data(), a1 - a0, II2_HOST, &dicts[s]);
    }
    if (!rc) rc = align_dicts_unlocked(ctx, k, dicts.data(), out);
    for (ii2_dict *d : dicts) delete d;
    return rc;
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

// one aligned view (ctx->mu held); wait: the stream is synchronised before the view is handed out (a batch waits once, at its end:
// the views' kernels follow each other on the context's stream, so they may share the scratch)
static int select_aligned_one(ii2_ctx *ctx, const ii2_seg *src, const ii2_align *a, uint32_t s, uint64_t first_list, ii2_seg **out, bool wait) {
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
    if (dm_alloc((void **)&d_blk_off, (nu + 1) * sizeof(uint32_t)) != hipSuccess || dm_alloc((void **)&d_cnt, (nu + 1) * sizeof(uint32_t)) != hipSuccess ||
        dm_alloc((void **)&d_last, (nu + 1) * sizeof(uint32_t)) != hipSuccess || dm_alloc((void **)&d_blk_list, (src->n_blocks + 1) * sizeof(uint32_t)) != hipSuccess) {
        dm_free(d_blk_off);
        dm_free(d_cnt);
        dm_free(d_last);
        dm_free(d_blk_list);
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
    if (e == hipSuccess && wait) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);           // (earlier kernels of the batch may still be writing the arrays)
        dm_free(d_blk_off); dm_free(d_cnt); dm_free(d_last); dm_free(d_blk_list);
        ctx->err = std::string("ii2_seg_select_aligned: ") + hipGetErrorString(e);
        return II2_EHIP;
    }
    return ii2_seg_adopt_view(ctx, src, nu, d_blk_off, d_cnt, d_last, d_blk_list, out);
}

int ii2_seg_select_aligned(ii2_ctx *ctx, const ii2_seg *src, const ii2_align *a, uint32_t s, uint64_t first_list, ii2_seg **out) {
    if (!ctx || !src || !a || !out || s >= a->k || src->device != ctx->device || a->device != ctx->device)
        return fail(ctx, II2_EINVAL, "ii2_seg_select_aligned: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return select_aligned_one(ctx, src, a, s, first_list, out, true);
}

// all k views of an alignment in one call: the same kernels, one wait at the end instead of one per view
int ii2_seg_select_aligned_all(ii2_ctx *ctx, const ii2_seg *const *srcs, const ii2_align *a, const uint64_t *first_list, ii2_seg **outs) {
    if (!ctx || !srcs || !a || !outs || a->device != ctx->device) return fail(ctx, II2_EINVAL, "ii2_seg_select_aligned_all: bad argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (uint32_t s = 0; s < a->k; s++) outs[s] = nullptr;
    int rc = II2_OK;
    for (uint32_t s = 0; s < a->k && !rc; s++) {
        if (!srcs[s] || srcs[s]->device != ctx->device) rc = fail(ctx, II2_EINVAL, "ii2_seg_select_aligned_all: a segment is NULL or lives on another device");
        else rc = select_aligned_one(ctx, srcs[s], a, s, first_list ? first_list[s] : 0, &outs[s], false);
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, II2_EHIP, "ii2_seg_select_aligned_all: sync failed");
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);            // (nothing of a failed batch is still being written when its views go)
        for (uint32_t s = 0; s < a->k; s++) { if (outs[s]) ii2_seg_free(outs[s]); outs[s] = nullptr; }
    }
    return rc;
}

void ii2_align_free(ii2_align *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    delete a;
}

}  // extern "C"
