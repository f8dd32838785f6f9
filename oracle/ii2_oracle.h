/*
 * ii2_oracle.h — CPU restatement of the reference's posting-list hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED at the decoded-value level against the known-answer tables
 * the reference's own tests hold (tests/golden/ref_kats.json, transcribed from
 * shard_test.go, inverted_index_test.go, removed_list_test.go, file/writer_test.go).
 * The Go reference cannot be compiled here (no Go toolchain) so there is no
 * oracle/_ref build.  The DV1 device codec has no reference counterpart (the
 * reference's codec is github.com/ronanh/intcomp v1.1.0, not in /root/reference):
 * its byte format is "parity unpinned" by construction and is checked by round-trip
 * identity only, exactly as file/writer_test.go:11-46 checks intcomp.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef II2_ORACLE_H
#define II2_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* slices.Sort on []uint32 (ascending). */
void orc_sort_u32(uint32_t *v, size_t n);
/* slices.Compact: drop adjacent equal values, returns new length. */
size_t orc_compact_u32(uint32_t *v, size_t n);

/* file.MergeTermValues (file/types.go:14-22): concat -> sort -> compact.
 * out must hold na+nb values; returns the merged length. */
size_t orc_merge_term_values(const uint32_t *a, size_t na, const uint32_t *b, size_t nb, uint32_t *out);

/* file.CompareTermValues (file/types.go:24-26) == bytes.Compare on the terms. */
int orc_compare_terms(const uint8_t *a, size_t la, const uint8_t *b, size_t lb);

/* RemovedLists.Values (removed_list.go:44-54): concatenate every live batch and
 * sort; duplicates are kept.  batches are given flat: values[] with batch_off[nb+1]. */
size_t orc_removed_values(const uint32_t *values, const uint64_t *batch_off, size_t n_batches, uint32_t *out);

/* RemovedLists.Sync (removed_list.go:57-71): keep[i] = 0 for batches whose timestamp
 * is older than min(timestamps); no-op when timestamps is empty. */
void orc_removed_sync(const int64_t *batch_ts, size_t n_batches, const int64_t *timestamps, size_t n_ts, uint8_t *keep);

/* Tombstone filter (shard.go:181-190): stable in-place compaction keeping v iff
 * slices.BinarySearch(removed_sorted, v) does not find it.  Returns new length. */
size_t orc_filter_removed(uint32_t *values, size_t n, const uint32_t *removed_sorted, size_t r);

/* Shard.Merge main loop over term-aligned segments (shard.go:163-212 with the k-way
 * merging iterator of shard.go:253-278).  Segment s holds, for aligned term id t,
 * the list seg_values[s][seg_off[s][t] .. seg_off[s][t+1]); present[s][t] != 0 says
 * the term exists in that segment (NULL = "present iff non-empty").  A term held by
 * >= 2 sources is folded pairwise with orc_merge_term_values; a term held by exactly
 * one source passes through verbatim (no sort / dedupe) — SURVEY §8 a1.  Then the
 * tombstone filter, then out_count[t] (0 => the caller drops the term, shard.go:192-194).
 * out_off has n_terms+1 entries; out_values must hold the sum of all input lengths.
 * Returns the number of surviving terms (0 => no segment is written, shard.go:219-225). */
uint64_t orc_merge_segments(uint32_t k, uint64_t n_terms,
                            const uint64_t *const *seg_off, const uint32_t *const *seg_values,
                            const uint8_t *const *present,
                            const uint32_t *removed_sorted, size_t r,
                            uint64_t *out_off, uint32_t *out_values);

/* Same loop spread over `threads` workers, one contiguous term range per work item —
 * the CPU-baseline analogue of InvertedIndex.Merge's goroutine pool over shards
 * (inverted_index.go:83-103).  Output identical to orc_merge_segments. */
uint64_t orc_merge_segments_mt(uint32_t k, uint64_t n_terms,
                               const uint64_t *const *seg_off, const uint32_t *const *seg_values,
                               const uint32_t *removed_sorted, size_t r,
                               uint64_t *out_off, uint32_t *out_values, int threads);

/* PrefixSearch union (inverted_index.go:274-292): append every list, sort, compact. */
size_t orc_union(uint32_t n_lists, const uint32_t *const *lists, const size_t *lens, uint32_t *out);

/* Intersection — ABSENT in the reference; build-defined as the
 * ascending sorted-unique ids present in every (sorted-unique) input list, optionally
 * minus the tombstones.  out must hold lens[0] (the fold starts there).  Two-pointer folds. */
size_t orc_intersect(uint32_t n_lists, const uint32_t *const *lists, const size_t *lens,
                     const uint32_t *removed_sorted, size_t r, uint32_t *out);

/* shardKey (shard.go:362-378): top 10 bits of the first two term bytes; terms shorter
 * than 2 bytes map to 0.  Returns the numeric key (the reference formats it "%04d"). */
uint32_t orc_shard_key(const uint8_t *term, size_t len);

/* ---- DV1 (the build's device posting format; no reference counterpart) ---------- */
typedef struct { uint32_t first_doc; uint32_t byte_off; } orc_skip;
#define ORC_DV1_BLOCK 256u
/* Encode n_lists lists (post_off[n_lists+1] into values; each sorted-unique ascending).
 * Blocks never cross a list; a block's first posting lives in its skip entry, the
 * remaining cnt-1 postings are LEB128 varints of the gaps.  skip has n_blocks+1
 * entries (sentinel: first_doc = last doc of the last list, byte_off = n_bytes).
 * Pass NULL outputs to size: returns n_blocks, *n_bytes. */
uint32_t orc_dv1_encode(uint64_t n_lists, const uint64_t *post_off, const uint32_t *values,
                        uint32_t *blk_off, orc_skip *skip, uint8_t *payload, uint64_t *n_bytes);
/* Decode back; out_post_off[n_lists+1], out_values sized by the caller.  Returns n. */
uint64_t orc_dv1_decode(uint64_t n_lists, const uint32_t *blk_off, const orc_skip *skip,
                        const uint8_t *payload, uint64_t *out_post_off, uint32_t *out_values);

#ifdef __cplusplus
}
#endif
#endif
