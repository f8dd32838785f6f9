/*
 * ii2.h — C ABI of the MI355X posting-list engine (libii2_hip.so).
 *
 * Drop-in boundary for the posting-list hot path of lezhnev74/inverted_index_2.  The
 * reference is a plain Go library with no FFI seam (SURVEY.md §8 b), so the boundary is
 * drawn *beneath* its exported Go API: the Go methods keep their signatures and bind
 * these entry points through cgo (stub in INTEGRATION.md).  Each entry point names the
 * reference code it replaces (file:line into the reference tree).
 *
 * Conventions
 *   - every call returns 0 on success or a negative II2_E* code; ii2_last_error(ctx)
 *     gives the message (the Go side wraps it with fmt.Errorf("…: %w"), cf. shard.go:160).
 *   - plain pointers and sizes only.  `where` says whether a caller buffer lives in host
 *     memory (II2_HOST — what cgo passes) or in this device's HBM (II2_DEVICE).
 *   - inputs are read-only and never retained after return; outputs are written only on
 *     success (all-or-nothing per call).  One exception, stated where it applies: an intersect /
 *     union whose result does not fit `cap` returns II2_ECAPACITY with the buffer's content
 *     unspecified (the merge entry points write nothing in that case).
 *   - per-call limits (II2_ERANGE beyond them): a merge takes < 2^32 input postings, < 2^31 input
 *     blocks and < 2^30 term slots; ii2_align_terms takes < 2^31 terms; one segment holds < 2^31
 *     lists, < 2^31 blocks and < 4 GiB of payload (split larger inputs into several segments / calls).
 *   - a ctx is bound to one GPU and one HIP stream; calls on one ctx are serialised by an
 *     internal mutex, so a ctx may be shared by goroutines / threads (InvertedIndex.Merge
 *     fans Shard.Merge over `concurrency` goroutines, inverted_index.go:83-103); use one
 *     ctx per worker to overlap.
 *   - segments and tombstone bitmaps belong to a DEVICE, not to the ctx that made them: they
 *     are read-only once created and any ctx of that device may use them, from any thread,
 *     concurrently (the reference's readers share segments, segments.go:32-46).  The caller
 *     must not free one while a call that reads it is still running.
 *   - doc ids ("values") are uint32 everywhere, as in the reference (file/types.go:11).
 *   - there is NO CPU fallback: without a usable gfx950 device ii2_ctx_create fails.
 */
#ifndef II2_H
#define II2_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define II2_ABI_VERSION 1

enum { II2_HOST = 0, II2_DEVICE = 1 };

enum {
    II2_OK = 0,
    II2_EINVAL = -1,     /* bad argument (NULL, unsorted sizes, too many lists …) */
    II2_ENOMEM = -2,     /* device or host allocation failed */
    II2_EHIP = -3,       /* a HIP runtime call failed (message has the HIP error string) */
    II2_ECAPACITY = -4,  /* caller's output buffer is too small (merge: nothing written; intersect / union: content unspecified) */
    II2_ERANGE = -5,     /* beyond a per-call or per-segment limit (see the conventions above) */
    II2_ECOMM = -6,      /* RCCL failure / communicator not initialised */
    II2_ENODEVICE = -7   /* no usable gfx950 GPU — the library has no CPU path */
};

#define II2_MAX_LISTS 64u   /* lists per intersect / union call, segments per merge */
#define II2_DV1_BLOCK 256u  /* postings per DV1 block */

typedef struct ii2_ctx ii2_ctx;
typedef struct ii2_seg ii2_seg;    /* device-resident DV1 segment: n_lists posting lists */
typedef struct ii2_tomb ii2_tomb;  /* device-resident tombstone bitmap */

/* One entry per DV1 block (+ one sentinel): first doc id of the block and the byte offset
 * of its payload (the varint gaps of the block's remaining postings). */
typedef struct { uint32_t first_doc; uint32_t byte_off; } ii2_skip;

typedef struct {
    uint64_t n_lists;      /* aligned term slots */
    uint64_t n_postings;
    uint64_t n_blocks;
    uint64_t n_bytes;      /* payload bytes */
} ii2_seg_info;

typedef struct {
    uint64_t n_in;         /* postings read (sum of input list lengths) */
    uint64_t n_out;        /* postings written */
    uint64_t n_terms_out;  /* terms with >= 1 surviving posting (0 => write no segment, shard.go:219-225) */
    uint64_t n_tiles;      /* workgroup tiles the merge was cut into */
} ii2_merge_stats;

/* ---- context -------------------------------------------------------------------------- */
int ii2_abi_version(void);
/* Binds a context to GPU `device` (hipSetDevice ordinal).  Fails with II2_ENODEVICE when
 * there is no gfx950 device. */
int ii2_ctx_create(int device, uint32_t flags, ii2_ctx **out);
void ii2_ctx_destroy(ii2_ctx *ctx);
const char *ii2_last_error(const ii2_ctx *ctx);     /* ctx may be NULL: last create error */
int ii2_ctx_sync(ii2_ctx *ctx);                     /* waits for the ctx stream */
void *ii2_ctx_stream(ii2_ctx *ctx);                 /* the hipStream_t, for event timing */
int ii2_ctx_device(const ii2_ctx *ctx);             /* the device ordinal the context is bound to (-1 for NULL) */

/* raw device buffers for II2_DEVICE arguments */
int ii2_dev_alloc(ii2_ctx *ctx, size_t bytes, void **dptr);
int ii2_dev_free(ii2_ctx *ctx, void *dptr);
int ii2_copy_h2d(ii2_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int ii2_copy_d2h(ii2_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- segments: the encode / decode steps ----------------------------------------------- */
/* Encode step.  Replaces Writer.Append -> intcomp.CompressUint32 (file/writer.go:32-59):
 * builds a device-resident DV1 segment from n_lists lists given CSR-style
 * (post_off[n_lists+1] into values).  Lists are stored verbatim (any u32 sequence
 * round-trips, cf. file/writer_test.go:14), but merge / intersect / union require each
 * list ascending and duplicate-free — what the index itself always produces. */
int ii2_seg_encode(ii2_ctx *ctx, uint64_t n_lists, const uint64_t *post_off, const uint32_t *values,
                   int where, ii2_seg **out);
/* Adopt an already DV1-encoded segment (blk_off[n_lists+1], skip[n_blocks+1], payload[n_bytes]).
 * The caller states the lengths of its arrays; blk_off[n_lists] must equal n_blocks and
 * skip[n_blocks].byte_off must equal n_bytes, else II2_EINVAL — the library never reads past the
 * stated lengths, whatever the arrays contain. */
int ii2_seg_import(ii2_ctx *ctx, uint64_t n_lists, uint64_t n_postings, uint64_t n_blocks, uint64_t n_bytes,
                   const uint32_t *blk_off, const ii2_skip *skip, const uint8_t *payload, int where, ii2_seg **out);
/* Decode step.  Replaces Reader.Next -> intcomp.UncompressUint32 (file/reader.go:79-100):
 * post_off[n_lists+1] and values[n_postings] are written to caller buffers. */
int ii2_seg_decode(ii2_ctx *ctx, const ii2_seg *seg, uint64_t *post_off, uint32_t *values, int where);
/* Copy the DV1 arrays out (any of the three may be NULL). */
int ii2_seg_export(ii2_ctx *ctx, const ii2_seg *seg, uint32_t *blk_off, ii2_skip *skip, uint8_t *payload);
/* A view of `src` with n_out term slots that shares src's skip table and payload: slot i holds list
 * src_list[i] of src, or is empty when src_list[i] < 0.  This is how the host aligns the term ids of
 * k segments before a merge (terms a segment lacks become empty slots) and drops emptied terms after
 * one.  Selected indices must ascend and may only skip empty lists between two selected ones. */
int ii2_seg_select(ii2_ctx *ctx, const ii2_seg *src, uint64_t n_out, const int64_t *src_list, ii2_seg **out);
/* ---- term alignment on the device -------------------------------------------------------- */
/* Replaces the k-way walk over the segments' term dictionaries that feeds the merging iterator
 * (shard.go:253-278, file/reader.go:33-71; order = file.CompareTermValues = bytes.Compare,
 * file/types.go:24-26).  Input, host memory: k sorted, duplicate-free dictionaries, flat —
 * term_bytes, term_off[n_all + 1] (byte offsets of the terms), seg_first[k + 1] (index in term_off
 * of each dictionary's first term; seg_first[0] = 0, seg_first[k] = n_all).  The result stays on
 * the device: the union dictionary (n_union terms, ascending) and, per dictionary, which of its
 * terms is which union term. */
typedef struct ii2_align ii2_align;
/* A segment's term dictionary resident in HBM (sorted, duplicate-free terms: term_off[n_terms + 1] byte offsets into
 * term_bytes, term_off[0] = 0; `where` says where the two arrays live).  Made once, when the segment is created or
 * loaded; alignments then read it in place. */
typedef struct ii2_dict ii2_dict;
int ii2_dict_create(ii2_ctx *ctx, const uint8_t *term_bytes, const uint64_t *term_off, uint64_t n_terms, int where, ii2_dict **out);
void ii2_dict_free(ii2_dict *dict);
/* The alignment of k resident dictionaries: no upload, no host pass over the terms, no sort — every term finds its place
 * in the k-way merge by bounded bisections in the other dictionaries (align.hip). */
int ii2_align_dicts(ii2_ctx *ctx, uint32_t k, const ii2_dict *const *dicts, ii2_align **out);
/* The same from flat host arrays (k temporary dictionaries are made and freed inside the call): */
int ii2_align_terms(ii2_ctx *ctx, uint32_t k, const uint8_t *term_bytes, const uint64_t *term_off,
                    const uint64_t *seg_first, ii2_align **out);
int ii2_align_info(const ii2_align *a, uint64_t *n_union, uint32_t *k);
/* rep[n_union]: index (into term_off) of one input term equal to union term u, in union order;
 * src_list[k * n_union]: row s = for every union term the term of dictionary s equal to it (index
 * inside that dictionary) or -1.  Either may be NULL. */
int ii2_align_export(ii2_ctx *ctx, const ii2_align *a, uint64_t *rep, int64_t *src_list);
/* The term-aligned view of `src` for dictionary s of the alignment, built on the device: n_union
 * slots, slot u = list first_list + (index of union term u in dictionary s), or empty.  The
 * dictionary must describe the consecutive lists [first_list, first_list + its size) of src.
 * Same result as ii2_seg_select with row s of src_list, without the mapping ever visiting the host. */
int ii2_seg_select_aligned(ii2_ctx *ctx, const ii2_seg *src, const ii2_align *a, uint32_t s,
                           uint64_t first_list, ii2_seg **out);
/* The views of ALL k dictionaries of the alignment in one call: srcs[s] / first_list[s] (NULL: 0 for every one) as above,
 * outs[k].  The same kernels with one wait at the end instead of one per view; all-or-nothing (outs are NULL on error). */
int ii2_seg_select_aligned_all(ii2_ctx *ctx, const ii2_seg *const *srcs, const ii2_align *a, const uint64_t *first_list,
                               ii2_seg **outs);
void ii2_align_free(ii2_align *a);

int ii2_seg_get_info(const ii2_seg *seg, ii2_seg_info *info);
/* Frees a segment.  As with any free, no call that reads the segment may still be running (the library's calls are
 * synchronous: that is "after they returned"; a query started with ii2_intersect_async is waited for first).  The
 * segment's device arrays go back to a size-class cache and are handed out again by the next segment that is made — a
 * stream of Shard.Merge calls allocates nothing and no free waits for the device.  The cache keeps at most
 * II2_DEVMEM_CACHE_MB megabytes (environment, default 16384; 0: every free goes to the driver) and is emptied when the
 * process's last context is destroyed. */
void ii2_seg_free(ii2_seg *seg);
/* Bytes of segment arrays handed out / waiting in the cache for reuse (either pointer may be NULL). */
void ii2_devmem_stats(uint64_t *live_bytes, uint64_t *idle_bytes);

/* ---- tombstones ------------------------------------------------------------------------ */
/* Replaces RemovedLists.Values() + slices.BinarySearch per value (removed_list.go:44-54,
 * shard.go:165,183): a dense bitmap over [0, max(removed)] built on the device; duplicates
 * in `removed` are harmless, order is irrelevant. */
int ii2_tomb_create(ii2_ctx *ctx, const uint32_t *removed, uint64_t n_removed, int where, ii2_tomb **out);
void ii2_tomb_free(ii2_tomb *tomb);

/* ---- the hot path ---------------------------------------------------------------------- */
/* Segment merge.  Replaces the body of Shard.Merge's loop (shard.go:163-212) together with
 * the k-way merging iterator it drains (shard.go:253-278, go-iterators NewMergingIterator
 * folding equal terms with file.MergeTermValues, file/types.go:14-22): for every aligned
 * term slot t the union of the k segments' lists, sorted, duplicate-free, minus the
 * tombstones.  All k segments must have the same n_lists (the host aligns term ids; the
 * bytes.Compare term ordering of file/types.go:24-26 stays on the host).
 * Output, device-resident: out_off[n_lists+1] (u64) and out_values (u32, capacity
 * out_cap >= sum of the inputs' n_postings is always enough).  A term whose
 * out_off[t+1]==out_off[t] has no survivors and is dropped by the caller (shard.go:192-194).
 * All-or-nothing: when the merged postings do not fit out_cap the call returns II2_ECAPACITY and
 * neither out_off nor out_values has been written (the fit is decided on the device before the
 * packing pass and the offset scan write anything). */
int ii2_merge_segments(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const ii2_tomb *tomb,
                       uint64_t *d_out_off, uint32_t *d_out_values, uint64_t out_cap,
                       ii2_merge_stats *stats);
/* Same, then the encode step on the merged lists: returns a new DV1 segment (what
 * w.Append writes, shard.go:207).  *out is NULL when no term survives. */
int ii2_merge_segments_to_seg(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const ii2_tomb *tomb,
                              ii2_seg **out, ii2_merge_stats *stats);

/* The common Shard.Merge in ONE launch.  Every Shard.Put writes a direct segment with one posting per term (shard.go:33-67)
 * and Shard.Merge picks the smallest segments first (shard.go:135-146): the usual merge is a handful of tiny segments.  This
 * entry point does, in one kernel and one host round trip, what otherwise takes ii2_align_terms + k x ii2_seg_select_aligned +
 * ii2_tomb_create + ii2_merge_segments_to_seg + the empty-term compaction: term alignment of the k dictionaries
 * (bytes.Compare order), same-term union, tombstone filter, empty-term drop (shard.go:192-194) and the encode step.
 * Input: k segments, segs[s] holding exactly one list per term of dictionary s, the dictionaries flat in host memory as for
 * ii2_align_terms, and RemovedLists.Values() (host, any order, duplicates allowed) or NULL.  Output: *out = the merged
 * segment, one list per SURVIVING term (NULL when none survives, shard.go:219-225); kept[j] (capacity seg_first[k]) = index
 * into term_off of an input term equal to the term of output list j; *n_kept = number of output lists.
 * Limits (II2_ERANGE beyond them: use the general entry points): */
#define II2_SMALL_MERGE_TERMS 512u       /* input terms in all */
#define II2_SMALL_MERGE_POSTINGS 8192u   /* input postings in all */
#define II2_SMALL_MERGE_REMOVED 4096u    /* removed ids */
int ii2_merge_small(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const uint8_t *term_bytes, const uint64_t *term_off,
                    const uint64_t *seg_first, const uint32_t *removed, uint64_t n_removed, ii2_seg **out, uint64_t *kept,
                    uint64_t *n_kept, ii2_merge_stats *stats);

/* The small Shard.Read in ONE launch (reference shard.go:72-75 -> makeIterator shard.go:253-278: the k-way merging iterator
 * without the tombstone filter): the merged lists of k small segments straight to host memory — what otherwise takes
 * ii2_align_terms + k x ii2_seg_select_aligned + ii2_merge_segments + two downloads (1.5 ms for a few hundred bytes).
 * Dictionary s (flat, as for ii2_merge_small) names the lists list_first[s] .. of segs[s] (a range-restricted read passes
 * the slice of each segment's terms that lies in [min, max]; list_first == NULL: every dictionary starts at list 0).
 * Output: *n_union merged terms in bytes.Compare order, EVERY one of them (a read drops nothing); rep[j] (capacity
 * seg_first[k]) = index into term_off of an input term equal to term j; post_off[j] .. post_off[j + 1] (capacity
 * seg_first[k] + 1) = its ids in values (capacity cap; II2_ECAPACITY and nothing written when they do not fit).
 * Limits: II2_SMALL_MERGE_TERMS / II2_SMALL_MERGE_POSTINGS (postings of the whole segments), II2_ERANGE beyond them. */
int ii2_read_small(ii2_ctx *ctx, uint32_t k, const ii2_seg *const *segs, const uint8_t *term_bytes, const uint64_t *term_off,
                   const uint64_t *seg_first, const uint64_t *list_first, uint64_t *rep, uint64_t *post_off, uint32_t *values,
                   uint64_t cap, uint64_t *n_union);

/* Multi-term intersection (build-defined operator, absent in the reference — SURVEY §0 D1):
 * ascending ids present in every list segs[i]/list_idx[i], minus the tombstones when
 * tomb != NULL (tomb == NULL is the reference's Read behaviour).
 * d_out: device buffer, capacity cap >= the shortest list is always enough. */
int ii2_intersect(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx,
                  const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *count);
/* Enqueue only: the count lands in the device word d_count; no host synchronisation.  (What the path choice needs
 * to know about a list — its first and last doc — is mirrored on the host when its segment is created, for segments
 * of up to 65536 lists; a list of a larger segment costs one small fetch + sync the first time it is queried.) */
int ii2_intersect_async(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx,
                        const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *d_count);

/* Multi-term union.  Replaces PrefixSearch's append + slices.Sort + slices.Compact
 * (inverted_index.go:274-292).  cap >= sum of the list lengths is always enough. */
int ii2_union(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx,
              const ii2_tomb *tomb, uint32_t *d_out, uint64_t cap, uint64_t *count);

/* ---- host-buffer convenience (what the cgo binding calls) ------------------------------- */
/* k term-aligned segments, flat: seg_off[k*(n_terms+1)] (per segment, offsets into that
 * segment's own slice), seg_base[k+1] (where each segment's slice starts in values).
 * removed: RemovedLists.Values() (any order, duplicates allowed) or NULL. */
int ii2_merge_host(ii2_ctx *ctx, uint32_t k, uint64_t n_terms, const uint64_t *seg_off,
                   const uint64_t *seg_base, const uint32_t *values,
                   const uint32_t *removed, uint64_t n_removed,
                   uint64_t *out_off, uint32_t *out_values, uint64_t out_cap, ii2_merge_stats *stats);
/* n lists, flat: list_off[n+1] into values. */
int ii2_intersect_host(ii2_ctx *ctx, uint32_t n, const uint64_t *list_off, const uint32_t *values,
                       const uint32_t *removed, uint64_t n_removed,
                       uint32_t *out, uint64_t cap, uint64_t *count);
int ii2_union_host(ii2_ctx *ctx, uint32_t n, const uint64_t *list_off, const uint32_t *values,
                   const uint32_t *removed, uint64_t n_removed,
                   uint32_t *out, uint64_t cap, uint64_t *count);

/* ---- multi-GPU: concatenate per-shard results in rank order ----------------------------- */
/* Replaces InvertedIndex.Read's shard-order concatenation (inverted_index.go:330-339) when
 * the term space (or the doc-id space) is sharded over the GPUs of one node.  One process
 * per GPU; rank 0 calls ii2_comm_unique_id and hands the 128 bytes to the others. */
#define II2_UNIQUE_ID_BYTES 128
#define II2_MAX_RANKS 64u
int ii2_comm_unique_id(void *id_out);
int ii2_comm_init(ii2_ctx *ctx, int world, int rank, const void *unique_id);
/* Every rank contributes d_local[n_local]; every rank receives all contributions in rank
 * order in d_out (capacity cap, in values) and the per-rank counts in counts_host[world].
 * Whether the concatenation fits is decided on the SMALLEST capacity of all ranks, identically
 * on every rank: either every rank exchanges, or every rank returns II2_ECAPACITY.  d_local may
 * be d_out + (its own offset) for an in-place gather, and must not overlap d_out otherwise. */
int ii2_allgatherv(ii2_ctx *ctx, const uint32_t *d_local, uint64_t n_local,
                   uint32_t *d_out, uint64_t cap, uint64_t *counts_host);

/* The same exchange in bytes (any device array): n_bytes / cap_bytes / counts_host are bytes. */
int ii2_allgatherv_bytes(ii2_ctx *ctx, const void *d_local, uint64_t n_bytes, void *d_out, uint64_t cap_bytes,
                         uint64_t *counts_host);

/* The exchange of MERGED SEGMENTS (what Shard.Merge writes, shard.go:207): every rank contributes the DV1 segment of its
 * term range and receives ONE segment holding all ranks' lists in rank order — the terms' global order when the ranks own
 * contiguous term ranges (shardKey ranges are contiguous, shard.go:362-378).  The postings travel encoded (about one byte
 * per posting for merged lists instead of four): after one all-gather of the ranks' shapes, ONE grouped exchange moves the
 * segment's arrays (list table, skip table, payload and the per-list counts / last docs / block owners) to every peer - a
 * send and a receive per peer and array, every peer on its own xGMI link, no host wait until the segment is complete; block
 * numbers, byte offsets and owners are shifted on arrival.  Every rank takes the same decision: II2_ERANGE on all ranks when
 * the concatenation exceeds one segment's limits.  With no communicator (one rank) *out is a copy of `local`.  `local` must be
 * a whole segment, not a view made by ii2_seg_select* (II2_EINVAL). */
int ii2_seg_allgather(ii2_ctx *ctx, const ii2_seg *local, ii2_seg **out);
/* The same concatenation on ONE device: the lists of segs[0], then those of segs[1], ... as one segment (what a rank holds after
 * the exchange, made from segments it already has - e.g. the term-range chunks of a merge; also the one-GPU check of the
 * exchange's arithmetic: same plan, same shifts).  Whole segments only; II2_ERANGE beyond one segment's limits. */
int ii2_seg_concat(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, ii2_seg **out);
/* Its arithmetic, host only: shape[3 r ..] = {n_lists, n_blocks, n_bytes} of rank r; list_off / block_off / byte_off
 * (world + 1 entries each) = where rank r's lists, blocks and payload bytes start in the concatenated segment.
 * II2_ERANGE when the totals exceed one segment's limits (2^31 lists or blocks, 4 GiB of payload). */
int ii2_seg_gather_plan(const uint64_t *shape, int world, uint64_t *list_off, uint64_t *block_off, uint64_t *byte_off);

/* The exchange's arithmetic, host only (no GPU needed): offsets[r] = where rank r's contribution
 * starts in the concatenation (offsets has world + 1 entries, offsets[world] = total).
 * II2_ECAPACITY when the total exceeds cap (offsets are still filled), II2_EINVAL for
 * world outside 1..II2_MAX_RANKS. */
int ii2_gatherv_offsets(const uint64_t *counts, int world, uint64_t cap, uint64_t *offsets);

/* ---- diagnostics ------------------------------------------------------------------------ */
/* Runs the device self-checks (wave scan, block decode against a scalar decode). 0 = pass. */
int ii2_selftest(ii2_ctx *ctx);
/* Tuning and diagnostic knobs, by name; unknown names are II2_EINVAL.  Path selection (1 = default on):
 *   intersect.dense, intersect.dense_bpw   the wave-streaming kernel for dense queries / its driver blocks per wave
 *   intersect.bitmap, intersect.g, intersect.wgs, intersect.map_docs   the general tile kernel's modes and sizes
 *   union.stream, union.dense, union.sparsity   unions through the streaming kernel / through the OR tiles (the latter up
 *                                           to `sparsity` docs of the lists' common range per posting, default 2048)
 *   setop.small, union.rank                 short-list ANDs / ORs in one launch; ORs of a few medium lists by ranking
 *   merge.bitmap_tiles, merge.large_tile    bitmap tiles for dense terms (1: terms with >= 1 posting per 80 docs; N > 1: per N docs; 0: off)
 *                                           / input postings a doc-range tile of a large term aims at
 *   intersect.and2                          dense 2-list ANDs: 1 one launch (look-back for the output offsets), 2 two kernels, 0 the n-list kernel
 *   encode.stream                           merged segments encoded in one pass over the ids (1) or by the two-pass encoder (0)
 *   merge.spin, intersect.and2_spin         polls a bounded inter-workgroup wait may take (tests shorten them; see ii2_ctx_counters)
 *   merge.alone                             Mi input postings above which a merge's tile kernel does not share the GPU with another
 *                                           context's (default 64: big ones only get in each other's way, small ones hide each other's tails)
 *   debug.no_chain                          experiments: the kernels that wait between workgroups (one-launch AND, one-pass encoder,
 *                                           direct placement) are NOT ordered per device across contexts (they may
 *                                           then hold each other's slots until their bounded waits run out)
 *   debug.stamps, profile.events            see ii2_debug_read / ii2_profile_read below
 * Every combination returns the same results; the tests run the kernels with the alternatives switched on and off. */
int ii2_set_option(ii2_ctx *ctx, const char *name, int64_t value);
/* With option "debug.stamps"=1 the intersect kernel sums, per workgroup, the shader cycles spent
 * in each part of its tile loop; this copies those counters (8 words per workgroup) out. */
int ii2_debug_read(ii2_ctx *ctx, uint64_t *out, uint64_t n_words);
/* With option "profile.events"=N (N > 0) every Nth call brackets its pass (intersect: every kernel of the
 * query; merge: every kernel of the call) with HIP events on the ctx stream — a timed event pair idles the stream
 * for ~10 us, so sample (N = 8) when the calls themselves are being timed; this waits for the stream and
 * returns the summed device time and the number of bracketed launches since the previous read. */
int ii2_profile_read(ii2_ctx *ctx, double *total_ms, uint64_t *launches);
/* One event pair around a whole run of calls: begin != 0 records the start event on the ctx stream, begin == 0 the
 * end event; ii2_profile_region_ms waits for the end event and returns the device time between the two — the time
 * the stream spent on everything enqueued in between, without a per-call event pair's idle time. */
/* How often this context took a second path: out[0] = merges repeated through the parking + packing pass, out[1] = two-list ANDs
 * repeated through the two-kernel form / segments encoded again by the two-pass encoder - in both cases because a BOUNDED WAIT
 * between workgroups of one launch ran out.  Those launches (the merge's direct placement, the one-launch AND, the one-pass
 * encoder) order their output by letting a workgroup wait for workgroups with smaller indices; that terminates because the
 * hardware starts a launch's workgroups in index order, which HIP does not promise - hence the bound, and the repeat on a path
 * without such waits (results identical).  out[2] = host waits (stream synchronisations) spent inside ii2_allgatherv* /
 * ii2_seg_allgather / ii2_seg_concat so far.  n = number of words `out` holds (3 are written). */
int ii2_ctx_counters(ii2_ctx *ctx, uint64_t *out, uint32_t n);
int ii2_profile_region(ii2_ctx *ctx, int begin);
int ii2_profile_region_ms(ii2_ctx *ctx, double *ms);

#ifdef __cplusplus
}
#endif
#endif
