/*
 * ii2_oracle.c — CPU restatement of the reference's posting-list hot path (plain C).
 * TEST INFRASTRUCTURE ONLY — see ii2_oracle.h for the rules and the parity status.
 * Citations are file:line into /root/reference (lezhnev74/inverted_index_2 @ 2024-10-26).
 */
#include "ii2_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- slices.Sort / slices.Compact -------------------------------------------------
 * The reference calls Go's slices.Sort (pdqsort) on []uint32; any correct ascending
 * sort gives the identical result, so this is an inlined-compare introsort-lite:
 * median-of-3 quicksort, insertion sort below 24, heapsort when recursion degenerates. */
static void ins_sort(uint32_t *v, size_t n) {
    for (size_t i = 1; i < n; i++) {
        uint32_t x = v[i];
        size_t j = i;
        while (j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; j--; }
        v[j] = x;
    }
}
static void sift(uint32_t *v, size_t i, size_t n) {
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= n) return;
        if (c + 1 < n && v[c + 1] > v[c]) c++;
        if (v[i] >= v[c]) return;
        uint32_t t = v[i]; v[i] = v[c]; v[c] = t;
        i = c;
    }
}
static void heap_sort(uint32_t *v, size_t n) {
    for (size_t i = n / 2; i-- > 0;) sift(v, i, n);
    for (size_t e = n; e-- > 1;) { uint32_t t = v[0]; v[0] = v[e]; v[e] = t; sift(v, 0, e); }
}
static void qs(uint32_t *v, size_t n, int depth) {
    while (n > 24) {
        if (depth-- == 0) { heap_sort(v, n); return; }
        uint32_t a = v[0], b = v[n / 2], c = v[n - 1];
        uint32_t p = a < b ? (b < c ? b : (a < c ? c : a)) : (a < c ? a : (b < c ? c : b));
        size_t i = 0, j = n - 1;
        for (;;) {
            while (v[i] < p) i++;
            while (v[j] > p) j--;
            if (i >= j) break;
            uint32_t t = v[i]; v[i] = v[j]; v[j] = t;
            i++; j--;
        }
        size_t left = j + 1;
        if (left < n - left) { qs(v, left, depth); v += left; n -= left; }
        else { qs(v + left, n - left, depth); n = left; }
    }
    ins_sort(v, n);
}
void orc_sort_u32(uint32_t *v, size_t n) {
    int depth = 2;
    for (size_t m = n; m > 1; m >>= 1) depth += 2;
    qs(v, n, depth);
}
size_t orc_compact_u32(uint32_t *v, size_t n) {
    if (n < 2) return n;
    size_t w = 1;
    for (size_t i = 1; i < n; i++)
        if (v[i] != v[w - 1]) v[w++] = v[i];
    return w;
}

/* file/types.go:14-22 — uniqueValues := append(append([]uint32{}, a...), b...);
 * slices.Sort; slices.Compact. */
size_t orc_merge_term_values(const uint32_t *a, size_t na, const uint32_t *b, size_t nb, uint32_t *out) {
    if (na) memcpy(out, a, na * sizeof *a);
    if (nb) memcpy(out + na, b, nb * sizeof *b);
    orc_sort_u32(out, na + nb);
    return orc_compact_u32(out, na + nb);
}

/* file/types.go:24-26 — bytes.Compare(a.Term, b.Term). */
int orc_compare_terms(const uint8_t *a, size_t la, const uint8_t *b, size_t lb) {
    size_t m = la < lb ? la : lb;
    int c = m ? memcmp(a, b, m) : 0;
    if (c) return c < 0 ? -1 : 1;
    return la < lb ? -1 : (la > lb ? 1 : 0);
}

/* removed_list.go:44-54 — concat all batches, slices.Sort, no dedupe. */
size_t orc_removed_values(const uint32_t *values, const uint64_t *batch_off, size_t n_batches, uint32_t *out) {
    size_t n = 0;
    for (size_t b = 0; b < n_batches; b++) {
        size_t len = (size_t)(batch_off[b + 1] - batch_off[b]);
        if (len) memcpy(out + n, values + batch_off[b], len * sizeof *out);
        n += len;
    }
    orc_sort_u32(out, n);
    return n;
}

/* removed_list.go:57-71 — delete batches with t < slices.Min(timestamps). */
void orc_removed_sync(const int64_t *batch_ts, size_t n_batches, const int64_t *timestamps, size_t n_ts, uint8_t *keep) {
    for (size_t b = 0; b < n_batches; b++) keep[b] = 1;
    if (n_ts == 0) return;
    int64_t oldest = timestamps[0];
    for (size_t i = 1; i < n_ts; i++) if (timestamps[i] < oldest) oldest = timestamps[i];
    for (size_t b = 0; b < n_batches; b++) if (batch_ts[b] < oldest) keep[b] = 0;
}

/* slices.BinarySearch: found flag only. */
static int bsearch_u32(const uint32_t *s, size_t n, uint32_t x) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (s[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && s[lo] == x;
}

/* shard.go:181-190 — in-place stable compaction of survivors. */
size_t orc_filter_removed(uint32_t *values, size_t n, const uint32_t *removed_sorted, size_t r) {
    size_t i = 0;
    for (size_t j = 0; j < n; j++) {
        uint32_t v = values[j];
        if (r && bsearch_u32(removed_sorted, r, v)) continue;
        values[i++] = v;
    }
    return i;
}

/* One term of the shard.go:163-212 loop.  scratch holds >= 2 * (sum of the term's
 * source lengths) values.  Writes survivors to out, returns their count. */
static size_t merge_one_term(uint32_t k, uint64_t t, const uint64_t *const *seg_off,
                             const uint32_t *const *seg_values, const uint8_t *const *present,
                             const uint32_t *removed_sorted, size_t r,
                             uint32_t *scratch, size_t scratch_half, uint32_t *out) {
    /* k-way merging iterator (shard.go:267): equal terms are folded with MergeTermValues,
     * one source after another; a lone source is yielded verbatim. */
    uint32_t *acc = scratch, *tmp = scratch + scratch_half;
    size_t nacc = 0;
    uint32_t sources = 0;
    for (uint32_t s = 0; s < k; s++) {
        size_t len = (size_t)(seg_off[s][t + 1] - seg_off[s][t]);
        int here = present && present[s] ? present[s][t] != 0 : len != 0;
        if (!here) continue;
        const uint32_t *src = seg_values[s] + seg_off[s][t];
        if (sources == 0) {
            if (len) memcpy(acc, src, len * sizeof *acc);
            nacc = len;
        } else {
            size_t m = orc_merge_term_values(acc, nacc, src, len, tmp);
            uint32_t *sw = acc; acc = tmp; tmp = sw;
            nacc = m;
        }
        sources++;
    }
    if (sources == 0) return 0;
    nacc = orc_filter_removed(acc, nacc, removed_sorted, r);
    if (nacc) memcpy(out, acc, nacc * sizeof *out);
    return nacc;
}

static size_t term_input_len(uint32_t k, uint64_t t, const uint64_t *const *seg_off) {
    size_t n = 0;
    for (uint32_t s = 0; s < k; s++) n += (size_t)(seg_off[s][t + 1] - seg_off[s][t]);
    return n;
}

uint64_t orc_merge_segments(uint32_t k, uint64_t n_terms,
                            const uint64_t *const *seg_off, const uint32_t *const *seg_values,
                            const uint8_t *const *present,
                            const uint32_t *removed_sorted, size_t r,
                            uint64_t *out_off, uint32_t *out_values) {
    size_t cap = 0;
    for (uint64_t t = 0; t < n_terms; t++) {
        size_t n = term_input_len(k, t, seg_off);
        if (n > cap) cap = n;
    }
    uint32_t *scratch = (uint32_t *)malloc((2 * cap + 2) * sizeof *scratch);
    uint64_t w = 0, survivors = 0;
    for (uint64_t t = 0; t < n_terms; t++) {
        out_off[t] = w;
        size_t n = merge_one_term(k, t, seg_off, seg_values, present, removed_sorted, r,
                                  scratch, cap + 1, out_values + w);
        w += n;
        if (n) survivors++;          /* shard.go:192-194: empty terms are dropped */
    }
    out_off[n_terms] = w;
    free(scratch);
    return survivors;
}

/* ---- threaded variant: worker pool over contiguous term ranges -------------------- */
typedef struct {
    uint32_t k; uint64_t t0, t1;
    const uint64_t *const *seg_off; const uint32_t *const *seg_values;
    const uint32_t *removed; size_t r;
    const uint64_t *in_pre;    /* prefix of per-term input lengths: where to park output */
    uint32_t *tmp_values;      /* survivors parked at in_pre[t] */
    uint64_t *counts;
} mt_job;
typedef struct { mt_job *jobs; size_t n_jobs; size_t next; pthread_mutex_t mu; } mt_queue;

static void *mt_worker(void *arg) {
    mt_queue *q = (mt_queue *)arg;
    for (;;) {
        pthread_mutex_lock(&q->mu);
        size_t j = q->next < q->n_jobs ? q->next++ : (size_t)-1;
        pthread_mutex_unlock(&q->mu);
        if (j == (size_t)-1) return NULL;
        mt_job *jb = &q->jobs[j];
        size_t cap = 0;
        for (uint64_t t = jb->t0; t < jb->t1; t++) {
            size_t n = (size_t)(jb->in_pre[t + 1] - jb->in_pre[t]);
            if (n > cap) cap = n;
        }
        uint32_t *scratch = (uint32_t *)malloc((2 * cap + 2) * sizeof *scratch);
        for (uint64_t t = jb->t0; t < jb->t1; t++)
            jb->counts[t] = merge_one_term(jb->k, t, jb->seg_off, jb->seg_values, NULL, jb->removed, jb->r,
                                           scratch, cap + 1, jb->tmp_values + jb->in_pre[t]);
        free(scratch);
    }
}

uint64_t orc_merge_segments_mt(uint32_t k, uint64_t n_terms,
                               const uint64_t *const *seg_off, const uint32_t *const *seg_values,
                               const uint32_t *removed_sorted, size_t r,
                               uint64_t *out_off, uint32_t *out_values, int threads) {
    if (threads < 1) threads = 1;
    uint64_t *in_pre = (uint64_t *)malloc((n_terms + 1) * sizeof *in_pre);
    in_pre[0] = 0;
    for (uint64_t t = 0; t < n_terms; t++) in_pre[t + 1] = in_pre[t] + term_input_len(k, t, seg_off);
    uint32_t *tmp = (uint32_t *)malloc((in_pre[n_terms] + 1) * sizeof *tmp);
    uint64_t *counts = (uint64_t *)calloc(n_terms + 1, sizeof *counts);
    /* work items of roughly equal input size, a few per thread */
    size_t n_jobs = (size_t)threads * 8;
    if (n_jobs > n_terms) n_jobs = n_terms ? (size_t)n_terms : 1;
    mt_job *jobs = (mt_job *)calloc(n_jobs, sizeof *jobs);
    uint64_t t = 0;
    size_t made = 0;
    for (size_t j = 0; j < n_jobs && t < n_terms; j++) {
        uint64_t target = in_pre[n_terms] / n_jobs * (j + 1);
        uint64_t e = t + 1;
        while (e < n_terms && (j + 1 < n_jobs) && in_pre[e] < target) e++;
        if (j + 1 == n_jobs) e = n_terms;
        jobs[made++] = (mt_job){k, t, e, seg_off, seg_values, removed_sorted, r, in_pre, tmp, counts};
        t = e;
    }
    mt_queue q = {jobs, made, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof *th);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, mt_worker, &q);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    uint64_t w = 0, survivors = 0;
    for (uint64_t tt = 0; tt < n_terms; tt++) {
        out_off[tt] = w;
        if (counts[tt]) {
            memcpy(out_values + w, tmp + in_pre[tt], counts[tt] * sizeof *tmp);
            w += counts[tt];
            survivors++;
        }
    }
    out_off[n_terms] = w;
    free(th); free(jobs); free(counts); free(tmp); free(in_pre);
    return survivors;
}

/* inverted_index.go:274-292 — found[prefix] = append(found[prefix], tv.Values...) for every
 * matching term, then slices.Sort + slices.Compact. */
size_t orc_union(uint32_t n_lists, const uint32_t *const *lists, const size_t *lens, uint32_t *out) {
    size_t n = 0;
    for (uint32_t i = 0; i < n_lists; i++) {
        if (lens[i]) memcpy(out + n, lists[i], lens[i] * sizeof *out);
        n += lens[i];
    }
    orc_sort_u32(out, n);
    return orc_compact_u32(out, n);
}

/* Build-defined (no reference function): fold lists with a two-pointer walk. */
size_t orc_intersect(uint32_t n_lists, const uint32_t *const *lists, const size_t *lens,
                     const uint32_t *removed_sorted, size_t r, uint32_t *out) {
    if (n_lists == 0) return 0;
    size_t n = lens[0];
    if (n) memcpy(out, lists[0], n * sizeof *out);
    n = orc_compact_u32(out, n);
    for (uint32_t l = 1; l < n_lists; l++) {
        const uint32_t *b = lists[l];
        size_t nb = lens[l], i = 0, j = 0, w = 0;
        while (i < n && j < nb) {
            uint32_t x = out[i], y = b[j];
            if (x < y) i++;
            else if (x > y) j++;
            else { out[w++] = x; i++; j++; }
        }
        n = w;
    }
    if (r) n = orc_filter_removed(out, n, removed_sorted, r);
    return n;
}

/* shard.go:362-378. */
uint32_t orc_shard_key(const uint8_t *term, size_t len) {
    uint8_t t0 = 0, t1 = 0;
    if (len >= 2) { t0 = term[0]; t1 = term[1]; }
    uint16_t key = (uint16_t)((uint16_t)t0 << 8);
    key = (uint16_t)(key + t1);
    return (uint32_t)(key >> 6);
}

/* ---- DV1 codec (CPU side of the round-trip checks) -------------------------------- */
static unsigned varint_len(uint32_t v) { unsigned n = 1; while (v >= 0x80) { v >>= 7; n++; } return n; }

uint32_t orc_dv1_encode(uint64_t n_lists, const uint64_t *post_off, const uint32_t *values,
                        uint32_t *blk_off, orc_skip *skip, uint8_t *payload, uint64_t *n_bytes) {
    uint32_t nb = 0;
    uint64_t bytes = 0;
    uint32_t last_doc = 0;
    for (uint64_t l = 0; l < n_lists; l++) {
        if (blk_off) blk_off[l] = nb;
        uint64_t a = post_off[l], e = post_off[l + 1];
        for (uint64_t p = a; p < e; p += ORC_DV1_BLOCK) {
            uint64_t pe = p + ORC_DV1_BLOCK < e ? p + ORC_DV1_BLOCK : e;
            if (skip) { skip[nb].first_doc = values[p]; skip[nb].byte_off = (uint32_t)bytes; }
            for (uint64_t q = p + 1; q < pe; q++) {
                uint32_t d = values[q] - values[q - 1];
                if (payload) {
                    while (d >= 0x80) { payload[bytes++] = (uint8_t)(d | 0x80); d >>= 7; }
                    payload[bytes++] = (uint8_t)d;
                } else bytes += varint_len(d);
            }
            nb++;
        }
        if (e > a) last_doc = values[e - 1];
    }
    if (blk_off) blk_off[n_lists] = nb;
    if (skip) { skip[nb].first_doc = last_doc; skip[nb].byte_off = (uint32_t)bytes; }
    if (n_bytes) *n_bytes = bytes;
    return nb;
}

uint64_t orc_dv1_decode(uint64_t n_lists, const uint32_t *blk_off, const orc_skip *skip,
                        const uint8_t *payload, uint64_t *out_post_off, uint32_t *out_values) {
    uint64_t w = 0;
    for (uint64_t l = 0; l < n_lists; l++) {
        out_post_off[l] = w;
        for (uint32_t b = blk_off[l]; b < blk_off[l + 1]; b++) {
            uint32_t cur = skip[b].first_doc;
            out_values[w++] = cur;
            uint64_t q = skip[b].byte_off, qe = skip[b + 1].byte_off;
            while (q < qe) {
                uint32_t d = 0; unsigned sh = 0; uint8_t c;
                do { c = payload[q++]; d |= (uint32_t)(c & 0x7F) << sh; sh += 7; } while ((c & 0x80) && q < qe);
                cur += d;
                out_values[w++] = cur;
            }
        }
    }
    out_post_off[n_lists] = w;
    return w;
}
