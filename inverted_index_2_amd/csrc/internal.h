// internal.h — host-side structures and kernel launcher prototypes of libii2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ii2.h"

// mailbox layout (u64 words; h_mail and d_mail both hold II2_MAIL_WORDS)
constexpr size_t II2_MAIL_WORDS = 1024;
constexpr size_t II2_MAIL_COUNT = 200;     // word of h_mail that receives the result count of ii2_intersect / ii2_union
constexpr size_t II2_MAIL_COMM = 256;      // all-gatherv: {count, cap} of this rank, then of every rank (2 + 2 * II2_MAX_RANKS words)

struct ii2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    // grow-only device scratch, carved per call (no hipMalloc on the steady-state path)
    uint8_t *ws = nullptr;
    size_t ws_cap = 0;
    size_t ws_used = 0;
    // small pinned host mailbox for counts read back after a call
    uint64_t *h_mail = nullptr;
    uint64_t *d_mail = nullptr;
    // options
    int64_t opt_intersect_g = 0;        // 0 = auto
    int64_t opt_intersect_wgs = 0;      // tile-kernel workgroups per CU (0 = default)
    int64_t opt_merge_large_tile = 0;   // 0 = default (MERGE_CAP / 2)
    int64_t opt_merge_direct = 1;       // tiles place their survivors themselves when the output buffer surely fits (0: always park + pack)
    int64_t opt_merge_spin = 0;         // bounded waits of the direct placement: polls (0 = default; tests shorten it)
    uint64_t merge_fallbacks = 0;       // merges repeated through the parking + packing pass after a bounded wait ran out
    int64_t opt_merge_skip = 0;         // timing experiments: phases of the merge tile kernel left out (results wrong)
    int64_t opt_merge_bitmap = 1;       // single-term tiles whose doc range fits the LDS bitmap are merged by marking bits
    uint8_t *aux = nullptr;             // grow-only: merge per-tile arrays + parked survivors
    size_t aux_cap = 0;
    void *h_small_in = nullptr, *h_small_out = nullptr, *d_small_in = nullptr, *d_small_out = nullptr;   // ii2_merge_small: upload / download blocks
    void *h_segs = nullptr;             // pinned staging block of a merge call's segment views ...
    void *d_segs = nullptr;             // ... and its device copy (MergeSegs)
    // grow-only staging buffers of the encode / decode / merge-to-segment paths (no hipMalloc per call)
    uint8_t *pool[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pool_cap[4] = {0, 0, 0, 0};
    int64_t opt_merge_alone = 0;        // merge: Mi input postings above which the tile kernel does not share the GPU with another one (0: 64)
    int64_t opt_debug_no_chain = 0;     // experiments: kernels that wait between workgroups are NOT ordered per device
    int64_t opt_debug_stamps = 0;       // intersect: collect per-phase cycle counters
    int64_t opt_intersect_map_docs = 0; // 0 = default (8192 docs per driver block)
    int64_t opt_union_rank = 1;         // unions of <= 8 lists / 2^20 postings by ranking (union_rank.hip)
    int64_t opt_union_sparsity = 2048;   // OR tiles are used up to this many docs of the common range per posting
    int64_t opt_small_setop = 1;        // queries of <= 32 blocks in all run as one single-workgroup kernel
    int64_t opt_union_stream = 1;       // ... and, for 2-4 lists paced by a long dense one, through the streaming kernel
    int64_t opt_union_dense = 1;        // unions of lists that are dense together go through the byte-map tiles (OR)
    int64_t opt_intersect_bitmap = 1;   // per-list bitmaps for very dense tiles
    int64_t opt_intersect_subtiles = 0; // tiny sparse drivers: tiles per CU their blocks are split into (0 = default)
    int64_t opt_intersect_submax = 0;   // ... and the most slices one driver block is cut into (0 = default)
    int64_t opt_intersect_dense = 1;    // dense 2..4-list queries go to the wave-streaming kernels (intersect_dense.hip)
    int64_t opt_dense_bpw = 0;          // driver blocks per wave there (0 = default)
    int64_t opt_encode_stream = 1;      // merged segments are encoded in one pass over the ids (encode_stream.hip)
    int64_t opt_and2_spin = 0;          // bounded waits of its look-back: polls (0 = default; 1 forces the fallback in tests)
    unsigned long long *d_lb = nullptr; // look-back records of the fused two-list AND: [0] error word, then agg[], grp[]
    size_t lb_cap = 0;                  // workgroups the records hold
    uint32_t lb_epoch = 0;              // launches so far
    uint64_t lb_fallbacks = 0;          // calls repeated through the two-kernel form
    uint32_t lb_pending = 0;            // epoch of a fused launch whose error word has not been looked at yet (0: none)
    int64_t opt_intersect_and2 = 1;     // dense 2-list ANDs: the shorter list's postings are tested against the longer one's bitmap (intersect_and2.hip)
    int64_t opt_profile_events = 0;     // N > 0: bracket the dominant kernel of every Nth call with HIP events
    uint64_t prof_calls = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;   // recorded pairs since the last read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;     // reusable pairs
    hipEvent_t region_ev[2] = {nullptr, nullptr};                 // ii2_profile_region: one pair around a whole run of calls
    unsigned long long *d_debug = nullptr;
    uint64_t *d_mail_mapped = nullptr;  // h_mail as the device sees it (result counts of the synchronous calls go there directly)
    uint32_t *d_small = nullptr;        // small set operations: ascending ids [8192] + the workgroup ticket (setop_small.hip)
    void *comm = nullptr;               // ncclComm_t
    int world = 1, rank = 0;
    uint64_t comm_syncs = 0;            // host waits inside the exchange entry points (what a chunked exchange pays per chunk)
    int cu_count = 256;
};

// skip table + payload of one encoded segment; shared by the views made with ii2_seg_select
namespace ii2 {
// device memory of segments: a size-class cache in front of hipMalloc / hipFree (devmem.cpp)
hipError_t dm_alloc(void **p, size_t bytes);
void dm_free(void *p);
hipError_t dm_malloc_retry(void **p, size_t bytes);   // hipMalloc; on failure the cache's idle arrays go back to the driver and it tries once more
void dm_trim(size_t keep_bytes);
void dm_user(int delta);
void dm_stats(uint64_t *live_bytes, uint64_t *idle_bytes);
}  // namespace ii2

struct ii2_seg_store {
    ii2_skip *d_skip = nullptr;
    uint8_t *d_payload = nullptr;
    void *slab = nullptr;            // one allocation that holds ALL arrays of a small segment (ii2_merge_small): freed as a whole
    ~ii2_seg_store() {
        ii2::dm_free(d_skip);
        ii2::dm_free(d_payload);
        ii2::dm_free(slab);
    }
};

// A segment is read-only once created and belongs to a DEVICE, not to a context: any context of that device may
// read it, from any thread (the reference's readers share segments, segments.go:32-46).
struct ii2_seg {
    int device = 0;
    bool is_view = false;                    // made by ii2_seg_select*: blocks numbered inside another segment's store (not self-contained)
    bool in_slab = false;                    // every array below lives in store->slab (nothing to free one by one)
    hipStream_t born = nullptr;              // stream of the call that is building the segment (cleared when it is finished): an unfinished segment's arrays may
                                             // still be written by enqueued kernels, so releasing one waits for that stream first
    std::shared_ptr<ii2_seg_store> store;   // owns d_skip / d_payload
    uint64_t n_lists = 0, n_postings = 0, n_blocks = 0, n_bytes = 0;
    // what a merge of this segment can at most read: the segment's own totals, or - for a view of some lists of a store
    // (ii2_seg_select) - the blocks and payload bytes of its window of the store; 0 = the totals above.  (A view's n_blocks and
    // n_bytes stay its STORE's: its block numbers and byte offsets live there.)
    uint64_t win_blocks = 0, win_bytes = 0;
    uint64_t merge_blocks() const { return win_blocks ? win_blocks : n_blocks; }
    uint64_t merge_bytes() const { return win_bytes ? win_bytes : n_bytes; }
    uint32_t *d_blk_off = nullptr;   // [n_lists+1]
    ii2_skip *d_skip = nullptr;      // [n_blocks+1]
    uint8_t *d_payload = nullptr;    // [n_bytes + 16]
    uint32_t *d_last_doc = nullptr;  // [n_lists] last doc id of each list (0 for an empty list)
    uint32_t *d_cnt = nullptr;       // [n_lists] postings of each list
    uint32_t *d_blk_list = nullptr;  // [n_blocks] list owning each block (0xFFFFFFFF: none of this view's lists)
    std::vector<uint32_t> h_blk_off; // host mirror of d_blk_off
    std::vector<uint32_t> h_cnt;     // host mirror of d_cnt (fetched when a small query first needs it)
    // per list: first_doc of its first and of its last block and its last doc (tile-height heuristic), fetched once
    struct ListSpan { uint32_t first_doc, last_block_first_doc, last_doc; };
    mutable std::mutex span_mu;             // contexts on different threads share the cache
    mutable std::unordered_map<uint64_t, ListSpan> span_cache;
    std::vector<uint32_t> h_spans;          // {first doc, first doc of the last block, last doc} per list, mirrored at creation (segments of <= SPAN_MIRROR_MAX lists)
    static constexpr uint64_t SPAN_MIRROR_MAX = 1u << 16;
};

namespace ii2 { constexpr uint32_t TOMB_SUM_SHIFT = 4; }    // docs per bit of a tombstone bitmap's summary: 1 << TOMB_SUM_SHIFT

struct ii2_tomb {
    int device = 0;
    uint32_t *d_words = nullptr;
    uint32_t *d_summary = nullptr;   // bit g set <=> some doc of [g << TOMB_SUM_SHIFT, (g + 1) << TOMB_SUM_SHIFT) is removed (same allocation as d_words)
    uint64_t n_words = 0;   // bitmap covers doc ids [0, 32*n_words)
};

void ii2_comm_destroy_internal(ii2_ctx *ctx);
int ii2_seg_alloc_internal(ii2_ctx *ctx, uint64_t n_lists, uint64_t n_postings, uint64_t n_blocks, uint64_t n_bytes, ii2_seg **out, bool with_meta = false);   // ctx->mu held
int ii2_seg_rebase_internal(ii2_ctx *ctx, ii2_seg *seg, int world, const uint64_t *lo, const uint64_t *bo, const uint64_t *qo);            // ctx->mu held
void *ii2_pool_get(ii2_ctx *ctx, int slot, size_t bytes);      // grow-only ctx buffer `slot`, at least `bytes` (nullptr: out of memory); ctx->mu held
uint64_t *ii2_mapped_mail(ii2_ctx *ctx, uint32_t word);
int ii2_seg_host_cnt(ii2_ctx *ctx, const ii2_seg *seg);       // fills seg->h_cnt on first use (thread-safe)
int ii2_seg_host_blk_off(ii2_ctx *ctx, const ii2_seg *seg);   // fills seg->h_blk_off on first use (thread-safe)
bool ii2_profile_pair(ii2_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1);   // false when profiling is off
// union through the intersection tiles (OR); *taken = false when the lists are too sparse for it (caller merges instead)
int ii2_union_dense_unlocked(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, const uint64_t *list_idx, const ii2_tomb *tomb,
                             uint32_t *d_out, uint64_t cap, uint64_t *d_count, bool *taken);
int ii2_seg_encode_dev_unlocked(ii2_ctx *ctx, uint64_t n_lists, const uint64_t *d_post_off, const uint32_t *d_values,
                                uint64_t n_postings, ii2_seg **out);
int ii2_seg_decode_dev_unlocked(ii2_ctx *ctx, const ii2_seg *seg, uint64_t *d_post_off, uint32_t *d_values);
int ii2_seg_encode_stream_unlocked(ii2_ctx *ctx, uint64_t n_lists, const uint64_t *d_post_off, const uint32_t *d_values, uint64_t n_postings,
                                   uint64_t n_nonempty, uint64_t payload_bound, ii2_seg **out);

namespace ii2 {

// One posting list as the kernels see it: `skip` points at the list's first block entry;
// skip[nblk] is readable (next list's first block or the sentinel) and bounds the payload.
struct ListView {
    const ii2_skip *skip;
    const uint8_t *payload;
    const uint32_t *last_doc;    // the list's last doc id
    uint32_t nblk;
    uint32_t pad;
};

struct SegView {
    const uint32_t *blk_off;    // [n_terms+1] first block of each aligned term slot
    const ii2_skip *skip;
    const uint8_t *payload;
    const uint32_t *cnt;        // [n_terms] postings of each list (same indexing as blk_off)
    const uint32_t *blk_list;   // [n_blocks of the store] list that owns each block, in the segment's own numbering
    const uint32_t *last_doc;   // [n_terms] last doc id of each list (same indexing as blk_off)
    uint32_t list_base;         // blk_off / cnt point at this list of the segment (blk_list[b] - list_base = term slot)
    uint32_t pad;
};

constexpr uint32_t MAX_LISTS = II2_MAX_LISTS;

struct IntersectParams {
    ListView lists[MAX_LISTS];   // lists[0] is the driver (fewest blocks)
    uint32_t n_lists;
    uint32_t G;                  // driver blocks per tile
    uint32_t sparse_driver;      // host hint: most tiles will gallop (picks the kernel instantiation with the pipelined gallop)
    uint32_t sub;                // tiles per driver block (1; > 1 only with G == 1: tiny drivers are spread over more workgroups)
    uint32_t n_tiles;
    uint32_t tomb_nwords;
    const uint32_t *tomb;        // may be null
    uint32_t *ranges;            // [n_tiles][2 + 4*n_lists] tile doc range + per-list phase descriptors
    uint32_t *out;               // final ids
    uint64_t out_cap;
    uint32_t *tmp;               // per-tile slot of slot_words: result bitmap words or survivor id list
    uint32_t *tile_count;        // [n_tiles] survivors per tile (bit 31: the slot holds an id list)
    uint64_t *d_count;           // result count
    unsigned long long *debug;   // optional per-workgroup phase cycle counters [grid][8] (diagnostics)
    uint32_t *sums;              // [n_sums] partial sums of tile_count: per 64 tiles, then per 4096 tiles
    uint32_t n_sums, n_sums1;    // total entries, entries of the per-64 level
    uint32_t slot_words;
    uint32_t desc_words;         // words per tile in `ranges`
    uint32_t max_grid;           // workgroups of the tile kernel (each walks tiles w, w+grid, ...)
    uint32_t bitmap_mode;        // 1: very dense tiles use per-list bitmaps (option intersect.bitmap)
    uint32_t map_docs_per_block; // tiles with more docs per driver block than this take the gallop path (option intersect.map_docs)
    uint32_t op_union;           // 1: OR instead of AND — fixed doc-range tiles [u_base + t * u_span, ...], every list searched per tile
    uint32_t u_base, u_span, u_max;
};

// small AND / OR in one workgroup (setop_small.hip)
constexpr uint32_t SMALL_SET_POSTINGS = 8192;  // postings the lists may hold together ...
constexpr uint32_t SMALL_SET_BLOCKS = 128;     // ... in at most this many DV1 blocks
struct SmallSetParams {
    ListView lists[MAX_LISTS];   // non-empty lists, any order
    uint32_t blk_base[MAX_LISTS + 1];   // first block of each list in the concatenated block list
    uint32_t lpre[MAX_LISTS + 1];       // exclusive prefix of the lists' posting counts; lpre[n_lists] <= SMALL_SET_POSTINGS
    uint32_t n_lists;
    uint32_t n_blocks;           // <= SMALL_SET_BLOCKS
    uint32_t is_union;
    uint32_t tomb_nwords;
    const uint32_t *tomb;        // may be null
    uint32_t *out;
    uint64_t out_cap;
    uint64_t *d_count;
    uint32_t *sorted;            // [SMALL_SET_POSTINGS] scratch: every id at its rank
    uint32_t *ticket;            // zero between launches (the last workgroup resets it)
};
hipError_t launch_setop_small(const SmallSetParams &p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// OR of a few medium-size lists by ranking (union_rank.hip)
constexpr uint32_t UNION_RANK_MAXL = 8;
constexpr uint32_t UNION_RANK_MAX_POSTINGS = 1u << 20;
struct UnionRankParams {
    ListView lists[UNION_RANK_MAXL];
    uint32_t blk_base[UNION_RANK_MAXL + 1];   // first block of each list in the concatenated block list
    uint32_t lpre[UNION_RANK_MAXL + 1];       // exclusive prefix of the lists' posting counts
    uint32_t n_lists;
    uint32_t n_blocks;
    uint32_t tomb_nwords;
    const uint32_t *tomb;        // may be null
    uint32_t *raw;               // [n_total] the lists decoded back to back
    uint32_t *sorted;            // [n_total] every id at its rank
    uint32_t *wg_cnt;            // survivors per 2048 ids
    uint32_t *out;
    uint64_t out_cap;
    uint64_t *d_count;
};
hipError_t launch_union_rank(const UnionRankParams &p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// output offsets inside one launch (lookback.h): records {epoch : 24 | value : 40} in a buffer of the context's own
struct LookBack {
    unsigned long long *agg;     // [workgroups]
    unsigned long long *grp;     // [2 * ceil(workgroups / 64)]
    unsigned long long *err;     // = epoch when a bounded wait ran out
    uint32_t epoch;              // this launch's number, 1 .. 2^24 - 1
    uint32_t spin;               // polls a wait may take (0: default; 0xFFFFFFFF: tests - workgroup 1 gives up at once)
};

// dense streaming intersection (intersect_dense.hip)
constexpr uint32_t DENSE_MAXL = 4;             // lists it takes
constexpr uint32_t DENSE_CAPW = 16384;         // docs per LDS window of a wave
struct DenseParams {
    ListView lists[DENSE_MAXL];  // lists[0] is the driver (fewest blocks)
    uint32_t first_doc[DENSE_MAXL], last_doc[DENSE_MAXL];   // of every list (host copies: starting guess of the block search)
    uint32_t n_lists;
    uint32_t is_union;           // 1: OR of the lists (lists[0] = the list with the most blocks paces the rounds)
    uint32_t u_lo, u_hi;         // union: smallest first doc / largest last doc over all lists
    uint32_t bpw;                // driver blocks per wave (a multiple of 16)
    uint32_t n_waves;            // waves with work; the grid is ceil(n_waves / 4) workgroups
    uint32_t n_meta;             // entries of meta (4 per workgroup)
    uint32_t base32;             // first doc of the driver & ~31: origin of the result bitmap
    uint32_t tomb_nwords;
    const uint32_t *tomb;        // may be null
    uint32_t *bitmap;            // result bits; wave w's slot starts at word ((its first doc & ~31) - base32) / 32 + w
    uint4 *meta;                 // per wave {first doc & ~31, words, ids, -}
    uint32_t *wg_sum;            // ids per workgroup
    uint32_t *out;
    uint64_t out_cap;
    uint64_t *d_count;
    unsigned long long *debug;   // optional per-workgroup cycle counters [2048][8] (diagnostics)
    uint32_t debug_expand;       // the counters are the expand kernel's (option debug.stamps = 2), else the tile kernel's
    uint2 *hmask;                // two-list AND (intersect_and2.hip): one answer bit per posting of lists[0], 64 bits per lane of a wave
    LookBack lb;                 // ... in one launch (k_and2_fused): the output offsets come from a look-back; lb.agg == nullptr selects the two-kernel form
    float a_scale;               // blocks of lists[1] per doc and ...
    float b_dpb;                 // ... docs per block of lists[0]: where a wave's first probe of lists[1]'s skip table goes
};
hipError_t launch_intersect_dense(const DenseParams &p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// AND of exactly two lists: lists[1] is marked, the postings of lists[0] are tested against it (meta = {first doc, last doc, ids, flags})
hipError_t launch_intersect_and2(const DenseParams &p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

}  // namespace ii2
int ii2_lookback_prepare(ii2_ctx *ctx, size_t n_wg, ii2::LookBack *lb);      // api.cpp; ctx->mu held
void ii2_lookback_forget(ii2_ctx *ctx);      // api.cpp: the context's stream is about to be destroyed
int ii2_lookback_launch(ii2_ctx *ctx, bool exclusive, const std::function<hipError_t()> &launch);      // api.cpp; ctx->mu held: kernels that wait between workgroups, ordered per device
namespace ii2 {

constexpr size_t SELFTEST_SCRATCH = 64 * 4 * 1408;
hipError_t launch_selftest(uint32_t *d_fail, uint8_t *d_scratch, hipStream_t s);

// scans (scan.hip) — temp storage comes from the caller
size_t scan_temp_bytes(size_t n);
hipError_t scan_excl_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s);
hipError_t scan3_excl(void *tmp, size_t tmp_bytes, const uint32_t *in0, uint64_t *out0, const uint32_t *in1, uint64_t *out1, const uint32_t *in2,
                      uint32_t *out2, size_t n, hipStream_t s);      // three scans of one length in the launches of one; tmp: 3 x scan_temp_bytes(n)
hipError_t scan_excl_u32_to_u64(void *tmp, size_t tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t s);
hipError_t scan_excl_u32_to_u64_guarded(void *tmp, size_t tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, const uint64_t *guard,
                                        uint64_t guard_max, hipStream_t s);
hipError_t scan_excl_u64(void *tmp, size_t tmp_bytes, const uint64_t *in, uint64_t *out, size_t n, hipStream_t s);

// codec
hipError_t launch_enc_list_blocks(const uint64_t *post_off, uint64_t n_lists, uint32_t *nblk, hipStream_t s);
hipError_t launch_enc_block_sizes(const uint64_t *post_off, const uint32_t *blk_off, uint64_t n_lists,
                                  const uint32_t *values, uint64_t n_blocks, uint32_t *sizes, ii2_skip *skip, hipStream_t s);
hipError_t launch_enc_write(const uint64_t *post_off, const uint32_t *blk_off, uint64_t n_lists,
                            const uint32_t *values, uint64_t n_blocks, const uint64_t *byte_off64,
                            ii2_skip *skip, uint8_t *payload, uint64_t n_postings, uint32_t *blk_list, hipStream_t s);
hipError_t launch_enc_stream(const uint64_t *post_off, const uint32_t *values, const uint32_t *blk_off, uint64_t n_lists, uint64_t n,
                             ii2_skip *skip, uint8_t *payload, uint64_t payload_cap, uint32_t *blk_list, uint32_t *part, uint64_t *d_result,
                             const LookBack &lb, unsigned long long *debug, hipStream_t s);      // part: [4 * enc_stream_workgroups(n)] scratch
uint64_t enc_stream_workgroups(uint64_t n);
hipError_t launch_enc_list_meta(const uint64_t *post_off, const uint32_t *values, uint64_t n_lists, uint32_t *cnt, uint32_t *last_doc, hipStream_t s);
hipError_t launch_dec_block_counts(const ii2_skip *skip, const uint8_t *payload, uint64_t n_blocks, uint32_t *counts, hipStream_t s);
hipError_t launch_dec_write(const ii2_skip *skip, const uint8_t *payload, uint64_t n_blocks, const uint64_t *bpo,
                            uint32_t *values, hipStream_t s);
hipError_t launch_gather_post_off(const uint32_t *blk_off, const uint64_t *bpo, uint64_t n_lists, uint64_t *post_off, hipStream_t s);
hipError_t launch_tomb_build(const uint32_t *removed, uint64_t n, uint32_t *words, uint64_t n_words, uint32_t *summary, hipStream_t s);
hipError_t launch_list_last_doc(const uint32_t *blk_off, const ii2_skip *skip, const uint8_t *payload, uint64_t n_lists, uint32_t *cnt, uint32_t *blk_list,
                                uint32_t *last_doc, hipStream_t s);
hipError_t launch_validate_counts(const uint32_t *blk_off, const uint32_t *blk_list, const ii2_skip *skip, const uint8_t *payload,
                                  uint64_t n_blocks, uint32_t *bad, hipStream_t s);
hipError_t launch_validate_seg(const uint32_t *blk_off, uint64_t n_lists, const ii2_skip *skip, uint64_t n_blocks, uint64_t n_bytes,
                               uint32_t *bad, hipStream_t s);
hipError_t launch_list_spans(const uint32_t *blk_off, const ii2_skip *skip, const uint32_t *last_doc, uint64_t n_lists, uint32_t *spans, hipStream_t s);
hipError_t launch_seg_rebase(uint32_t *blk_off, uint64_t n_lists, uint32_t add_blocks, ii2_skip *skip, uint64_t n_blocks, uint32_t add_bytes,
                             uint32_t *blk_list, uint32_t add_lists, hipStream_t s);
hipError_t launch_seg_close(uint32_t *blk_off_end, uint32_t n_blocks, ii2_skip *skip_end, uint32_t n_bytes, uint8_t *payload_end, uint32_t *blk_list_end, hipStream_t s);
hipError_t launch_sum_u32(const uint32_t *v, uint64_t n, uint64_t *out, hipStream_t s);
hipError_t launch_max_u32(const uint32_t *v, uint64_t n, uint32_t *out, hipStream_t s);

// intersect
constexpr uint32_t ISECT_GMAX = 16;         // driver blocks per tile (max)
constexpr uint32_t ISECT_SMAX = 16384;      // doc span a tile's LDS byte map can cover
hipError_t launch_intersect(const IntersectParams &p, uint64_t *d_tile_off, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// merge / union (merge.hip)
constexpr uint32_t MERGE_THREADS = 256;     // threads per workgroup of the tile kernel
constexpr uint32_t MERGE_CAP = 3584;        // postings a tile sorts in LDS (14 per thread)
constexpr uint32_t MERGE_NT_MAX = 256;      // terms per batch tile (one thread per term)
constexpr uint32_t MERGE_BM_WORDS = 2 * MERGE_CAP;          // LDS bitmap of a bitmap tile: the sort arrays' 56 KB
constexpr uint32_t MERGE_BM_DOCS = MERGE_BM_WORDS * 32u;    // docs a bitmap tile covers (458752)
// the k term-aligned inputs, by value in the kernel arguments
struct MergeSegs {
    SegView segs[MAX_LISTS];
    uint32_t k;
    uint32_t pad0;
    uint64_t n_terms;
};
static_assert(sizeof(MergeSegs) <= 4096, "MergeSegs travels through one pinned 4 KB block");

// direct placement of the merged postings (merge.hip): tile tickets, an error flag for bounded waits
struct MergeSync { uint32_t ticket, error, pad0, pad1; };

struct MergeParams {
    uint32_t k;
    uint32_t n_tiles_ub;          // host-side upper bound of the tile count (sizes grids and per-tile arrays)
    uint64_t n_terms;
    const uint32_t *tomb;
    const uint32_t *tomb_summary;  // 1 bit per (1 << TOMB_SUM_SHIFT) docs: most bit tests stop at this small, L2-resident array
    uint32_t tomb_nwords;
    uint32_t small_max;           // terms with more input postings get their own tiles (doc ranges)
    uint32_t range_target;        // input postings a range tile of a large term aims at
    uint32_t wmin;                // minimum packing weight of a term (bounds terms per batch)
    uint32_t batch_q;             // batch id = weight prefix / batch_q
    uint32_t bitmap_tiles;        // 1: dense terms are cut into bitmap tiles
    uint32_t bitmap_sparsity;     // ... when they hold at least one posting per this many docs
    uint32_t pad0;                // option debug.merge_skip (experiments / tests): phases left out, bit 6: the scanner never runs
    uint32_t spin_limit;          // polls a wait of the direct placement may take (0: default)
    uint32_t direct;              // 1: tiles go to their final place from inside the tile kernel (ticket order + scanner); 0: parking + packing pass
    MergeSync *sync;              // (zeroed by the host)
    uint64_t *tile_off;           // [n_tiles_ub + 1] output offset of every tile: written by the scanner (direct) or by the scan before the packing pass
    // per term (plan)
    uint32_t *tn;                 // [T+1] input postings
    uint32_t *tmin, *tmax;        // [T+1] smallest / largest doc id over the k lists
    uint32_t *tinfo;              // [T+1] longest list (bits 0-7), bit 8: bitmap tiles, bit 9: splitters uniform in doc space
    uint32_t *weight;             // [T+1] packing weight of a small term (0 for large terms)
    uint32_t *ntl;                // [T+1] tiles of a large term (0 for small terms)
    const uint64_t *npre;         // [T+1] exclusive prefix of tn: where a term's parking region starts
    const uint32_t *term_tile;    // [T+1] first tile of each term; [T] = number of tiles
    const uint32_t *n_tiles_dev;  // = term_tile + T
    // per tile
    uint4 *desc;                  // {t0, t1 | flags, dlo, dhi}
    uint4 *cut0;                  // [k * n_tiles_ub] where list s's part of the tile begins: {block | inside, payload byte, doc before, -} (merge.hip: cut_for)
    uint2 *cut1;                  // [k * n_tiles_ub] ... and where it ends: {block | inside, payload byte}
    uint32_t cut_ss, cut_st;      // entry of (list s, tile): s * cut_ss + tile * cut_st ([tile][s] for few lists, [s][tile] for many: the plan kernel's lanes then run along the tiles)
    uint32_t *term_alloc;         // [T] bump allocator inside a large term's parking region (zeroed by the host)
    uint32_t *tmp;                // parked survivors
    uint32_t *tile_count;         // [n_tiles_ub+1] survivors per tile (zeroed by the host)
    unsigned long long *tile_slot;   // [n_tiles_ub] where in tmp the tile parked them
    uint32_t *out_counts;         // [T] survivors per term (zeroed by the host)
    uint32_t *out_values;
    uint64_t out_cap;
    uint64_t *d_total;            // total survivors
    unsigned long long *debug;    // optional diagnostics words
};
constexpr uint32_t MERGE_DESC_LARGE = 1u << 30;   // desc.y: the tile belongs to a large term (its count goes through tile_count)
constexpr uint32_t MERGE_DESC_BITMAP = 1u << 31;  // desc.y: bitmap tile

hipError_t launch_merge_plan_terms(const MergeSegs *ms, const MergeParams &p, hipStream_t s);
hipError_t launch_merge_init(uint64_t *mail, uint32_t n_mail_words64, uint32_t *cnt, uint64_t n_cnt, void *aux, uint64_t aux_bytes, uint64_t *tile_off,
                             uint64_t n_off, hipStream_t s);      // the call's clears in one launch (aux_bytes: a multiple of 4)
hipError_t launch_merge_heads(const MergeParams &p, const uint64_t *wpre, uint32_t *head, hipStream_t s);
hipError_t launch_merge_term_tile(const MergeParams &p, const uint32_t *head, const uint32_t *hpre, const uint32_t *lpre,
                                  uint32_t *term_tile, hipStream_t s);
hipError_t launch_merge_tile_desc(const MergeSegs *ms, const MergeParams &p, hipStream_t s);
hipError_t launch_merge_tile_runs(const MergeSegs *ms, const MergeParams &p, hipStream_t s);
hipError_t launch_merge_tiles(const MergeSegs *ms, const MergeParams &p, uint32_t grid, hipStream_t s);
hipError_t launch_merge_large_counts(const MergeParams &p, const uint64_t *tile_off, hipStream_t s);
hipError_t launch_merge_pack(const MergeParams &p, const uint64_t *tile_off, hipStream_t s);
hipError_t launch_count_nonzero(const uint32_t *v, uint64_t n, uint64_t *out, hipStream_t s, const uint32_t *err = nullptr, const uint32_t *n_tiles = nullptr,
                                uint64_t *mail = nullptr);      // mail: also mail[3] = *err (0 without), mail[4] = *n_tiles

}  // namespace ii2
