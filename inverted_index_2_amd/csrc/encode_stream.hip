// encode_stream.hip — DV1 encoding of a CSR (post_off, values) in ONE pass over the ids (gfx950, wave64).
// Role: Writer.Append -> intcomp.CompressUint32 of every merged term (reference shard.go:207, file/writer.go:32-59); the byte
// format is this build's own (DV1, include/ii2.h) and the bytes are identical to the two-pass encoder's (codec.hip).
//
// codec.hip encodes block by block: a wave walks eight blocks one after the other, and every block costs it three dependent
// round trips (owner list, its offsets, the ids) - and it does so twice, once for the blocks' sizes and, after a scan and a host
// round trip for the payload's size, once more for the bytes: 2 x 3.9 GB read at ~2 TB/s for the 0.98 G postings of BASELINE
// configs[2]'s merge, 5 ms behind an 11 ms merge.  Here the unit is a run of consecutive OUTPUT POSITIONS, whatever lists they
// belong to:
//   * a wave takes 1024 consecutive ids, 16 per lane (four 16-byte loads), a workgroup 4096;
//   * which lists they belong to: the list of the wave's first position comes from a small partition pass (k_enc_partition: a
//     bisection of post_off per wave, all waves at once), then the lists that start inside the wave's run are read 64 at a time and leave their number at their first position in a wave-private LDS array; a prefix
//     maximum over it tells every lane the list it starts in, one load tells it where that list began and which block it began
//     with - from there the lane walks its 16 ids alone: position in the list (a multiple of 256 = a block's first id, which
//     is not encoded), gap, varint length;
//   * the workgroup's byte count goes through the look-back of lookback.h (payload offsets are a prefix sum over all
//     workgroups); while it travels the wave writes its varints into LDS, then the bytes leave as aligned 16-byte stores and
//     the blocks' skip entries {first id, byte offset} and owners are written by the lanes that hold their first ids.
// One read of the ids, one write of the payload, no sizes array, no scan launch, no host round trip before the payload exists
// (the caller allocates it from an upper bound).
#include "dv1_device.h"
#include "internal.h"
#include "lookback.h"

namespace ii2 {

constexpr uint32_t ES_PER_LANE = 16;
constexpr uint32_t ES_WAVE = 64u * ES_PER_LANE;              // ids per wave (1024)
constexpr uint32_t ES_TILES = 2;                             // tiles (1024 consecutive ids) a wave encodes behind one wait
constexpr uint32_t ES_WG = 4u * ES_TILES * ES_WAVE;          // ids per workgroup
// A wave's bytes at worst.  Ids ascend inside a list, so a list has at most 15 gaps of five bytes (>= 2^28) and a block's first id
// takes none: the fullest 1024 ids are the 15-id tail of one list and 63 lists of 16 (4800 bytes), not 5 x 1024 - which is what
// lets EIGHT workgroups share a CU's LDS instead of seven.  (Ids that do not ascend can exceed it: the wave then reports the same
// failure as a payload that does not fit and the two-pass encoder, which has no stage, takes over.)
constexpr uint32_t ES_STAGE_BYTES = 4864u;
constexpr uint32_t ES_STAGE = ES_STAGE_BYTES + 32u;          // + read-ahead of the copy-out
constexpr uint32_t ES_WAVE_LDS = ((ES_STAGE > ES_WAVE * 4u ? ES_STAGE : ES_WAVE * 4u) + 15u) & ~15u;   // bytes: the stage, or (before it) one u32 per position
static_assert(4u * ES_WAVE_LDS + 64u <= 20480u, "eight workgroups per CU");

// inclusive prefix maximum over the 64 lanes of a wave (lanes without a source see 0)
__device__ __forceinline__ uint32_t es_wave_incl_max(uint32_t x) {
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x = y > x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); x = y > x ? y : x;
    return x;
}

// bytes of gap's LEB128 varint: ceil(bits / 7), bits >= 1
__device__ __forceinline__ uint32_t es_varint_len(uint32_t gap) {
    const uint32_t bits = 32u - (uint32_t)__clz((int)(gap | 1u));
    return ((bits + 6u) * 37u) >> 8;                        // = (bits + 6) / 7 for bits <= 32
}

// list that holds every wave's first output position: the last list that starts at or before it.  One thread per wave of
// k_enc_stream (a bisection of post_off each: a few dozen microseconds for a million waves, instead of a four-round search at the
// head of every wave's chain of dependent loads)
__global__ __launch_bounds__(256) void k_enc_partition(const uint64_t *__restrict__ post_off, uint64_t n_lists, uint64_t n, uint32_t *__restrict__ part) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t P0 = w * ES_WAVE;
    if (P0 >= n) return;
    uint64_t lo = 0, hi = n_lists + 1u;           // first index with post_off > P0 (post_off[n_lists] = n > P0)
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (post_off[mid] <= P0) lo = mid + 1u; else hi = mid; }
    part[w] = (uint32_t)(lo - 1u);
}

struct EncStreamParams {
    const uint64_t *post_off;    // [n_lists + 1]
    const uint32_t *values;      // [n]
    const uint32_t *blk_off;     // [n_lists + 1] first block of every list (exclusive scan of ceil(cnt / 256))
    uint64_t n_lists, n;
    ii2_skip *skip;              // [n_blocks + 1]
    uint8_t *payload;            // upper-bound allocation (+ 16)
    uint32_t *blk_list;          // [n_blocks + 1]
    uint64_t payload_cap;        // bytes the payload may take (the call fails, nothing useful written, when the bytes exceed it)
    const uint32_t *part;        // [waves] list that holds every wave's first position (k_enc_partition)
    uint64_t *d_result;          // [0] payload bytes (all ones: a bounded wait ran out or the payload did not fit), [1] blocks
    LookBack lb;
    unsigned long long *debug;   // optional per-workgroup cycle counters [2048][8] (option debug.stamps: diagnostics)
};

// sum of the eight 3-bit fields of x (each <= 5)
__device__ __forceinline__ uint32_t es_sum3(uint32_t x) {
    const uint32_t t = (x & 007070707u) + ((x >> 3) & 007070707u);      // four 6-bit fields, each <= 10
    return ((t * 0x41041u) >> 18) & 63u;                                // field 3 of t x (1 + 2^6 + 2^12 + 2^18) = their sum (<= 40)
}
// bytes of the varints before id j (0..16) of a lane: lo / hi = 3 bits per id, ids 0-7 / 8-15
__device__ __forceinline__ uint32_t es_bytes_before(uint32_t lo, uint32_t hi, uint32_t j) {
    const uint32_t mlo = j >= 8u ? lo : lo & ((1u << (3u * j)) - 1u);
    const uint32_t mhi = j >= 8u ? hi & ((1u << (3u * (j - 8u))) - 1u) : 0u;
    return es_sum3(mlo) + es_sum3(mhi);
}

// list that holds output position pos: the last one that starts at or before it and is not empty.  `guess` is not after it
// (post_off[guess] <= pos); exact unless empty lists lie between - then a doubling walk and a bisection.
__device__ __forceinline__ uint64_t es_owner(const uint64_t *__restrict__ post_off, uint64_t n_lists, uint64_t guess, uint64_t pos) {
    uint64_t lo = guess;
    if (post_off[lo + 1u] > pos) return lo;
    uint64_t step = 1u, hi = lo + 1u;                  // post_off[hi] <= pos so far
    for (;;) {
        lo = hi;
        hi = lo + step < n_lists ? lo + step : n_lists;      // (post_off[n_lists] = n > pos)
        if (post_off[hi] > pos) break;
        step <<= 1;
    }
    while (hi - lo > 1u) { const uint64_t mid = lo + ((hi - lo) >> 1); if (post_off[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}

// One tile = 1024 consecutive output positions, 16 per lane, as one wave holds it between its phases.
struct EsTile {
    uint32_t v[ES_PER_LANE];
    uint64_t P0;                 // the tile's first output position
    uint64_t l0;                 // list that holds it ...
    uint32_t blk_l0;             // ... and that list's first block
    uint32_t nloc;               // ids of the tile (0: past the end)
    uint32_t prev0;              // the id before my first one
    uint32_t lens_lo, lens_hi;   // 3 bits per id: bytes of its varint (0: a block's first id)
    uint32_t L;                  // bit j: a list starts at my id j
    uint32_t cont;               // bit j: my id j starts a block inside a list (at most one)
    uint32_t cont_first;         // ... that id
    uint32_t my_pos;             // position of my first id in its list
    uint32_t starts_before;      // lists that start in the tile before my first id
    uint32_t lane_bytes, lane_off, bytes;       // my bytes, the bytes of the lanes before me, the tile's
    bool any_list_start;         // (wave-uniform) a list starts inside the tile
};
struct EsListLoads { uint64_t s, s1, ls0; };

// my 16 ids (guarded at the array's end), the id before the tile, the tile's first list: all requested, nothing waited for
__device__ __forceinline__ void es_tile_begin(const EncStreamParams &p, EsTile &T, uint64_t P0, uint32_t *lm) {
    const uint32_t i0 = ES_PER_LANE * (uint32_t)lane_id();
    T.P0 = P0;
    T.nloc = P0 < p.n ? (uint32_t)(p.n - P0 < ES_WAVE ? p.n - P0 : ES_WAVE) : 0u;
    T.lens_lo = T.lens_hi = T.L = T.cont = T.cont_first = T.my_pos = T.starts_before = T.lane_bytes = 0u;
    T.any_list_start = false;
    T.l0 = 0; T.blk_l0 = 0; T.prev0 = 0;
#pragma unroll
    for (uint32_t j = 0; j < ES_PER_LANE; j++) T.v[j] = 0u;
    if (T.nloc == 0u) return;
    if (T.nloc == ES_WAVE) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.values + P0 + i0);
        const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
        T.v[0] = a.x; T.v[1] = a.y; T.v[2] = a.z; T.v[3] = a.w; T.v[4] = b.x; T.v[5] = b.y; T.v[6] = b.z; T.v[7] = b.w;
        T.v[8] = c.x; T.v[9] = c.y; T.v[10] = c.z; T.v[11] = c.w; T.v[12] = d.x; T.v[13] = d.y; T.v[14] = d.z; T.v[15] = d.w;
    } else {
#pragma unroll
        for (uint32_t j = 0; j < ES_PER_LANE; j++) T.v[j] = i0 + j < T.nloc ? p.values[P0 + i0 + j] : 0u;
    }
    T.prev0 = P0 ? p.values[P0 - 1u] : 0u;           // (lane 0's; the other lanes' comes from their neighbour)
    lm[lane_id()] = 0u;
    T.l0 = p.part[P0 / ES_WAVE];
}
// the first 64 lists behind the tile's first one (lane i: list l0 + 1 + i), and where that first one began
__device__ __forceinline__ EsListLoads es_tile_lists_ask(const EncStreamParams &p, EsTile &T) {
    EsListLoads r{~0ull, 0ull, 0ull};
    if (T.nloc == 0u) return r;
    const uint64_t li = T.l0 + 1ull + (uint64_t)lane_id();
    r.s = li <= p.n_lists ? p.post_off[li] : ~0ull;
    r.s1 = li < p.n_lists ? p.post_off[li + 1ull] : 0ull;
    r.ls0 = p.post_off[T.l0];
    T.blk_l0 = p.blk_off[T.l0];
    return r;
}
// every non-empty list that starts inside the tile leaves a bit at its first position; from the bits: where my first id stands
// in its list, the block starts among my ids, the varint lengths (walk A)
__device__ __forceinline__ void es_tile_lists(const EncStreamParams &p, EsTile &T, const EsListLoads &r, uint32_t *lm) {
    if (T.nloc == 0u) return;
    const int l = lane_id();
    const uint32_t i0 = ES_PER_LANE * (uint32_t)l;
    const uint64_t P0 = T.P0, P1 = P0 + T.nloc;
    {
        uint64_t s = r.s, s1 = r.s1;
        for (uint64_t i = (uint64_t)l;;) {
            const bool in = s < P1;
            if (in && s1 > s) {                                  // (several empty lists may share s: the non-empty one owns it)
                const uint32_t rel = (uint32_t)(s - P0);
                atomicOr(&lm[rel >> 4], 1u << (rel & 15u));
            }
            if (__ballot(!in) != 0ull) break;
            i += 64u;
            const uint64_t li = T.l0 + 1ull + i;
            s = li <= p.n_lists ? p.post_off[li] : ~0ull;
            s1 = li < p.n_lists ? p.post_off[li + 1ull] : 0ull;
        }
    }
    const uint32_t before_wave = T.prev0;
    T.prev0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T.v[ES_PER_LANE - 1u], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    if (l == 0) T.prev0 = before_wave;
    // ---- where my first id stands in its list: behind the last list start before it, or behind the tile's first list's
    const uint32_t L = lm[l];
    T.L = L;
    T.any_list_start = __ballot(L != 0u) != 0ull;
    const uint32_t nloc = T.nloc;
    const uint32_t valid = nloc == ES_WAVE ? 0xFFFFu : i0 >= nloc ? 0u : nloc - i0 >= 16u ? 0xFFFFu : (1u << (nloc - i0)) - 1u;
    const uint32_t mp = L ? i0 + 32u - (uint32_t)__clz((int)L) : 0u;            // 1 + tile position of my last list start
    uint32_t before = 0u;
    if (T.any_list_start) {
        const uint32_t incl = es_wave_incl_max(mp);
        before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (l == 0) before = 0u;
        const uint32_t nl = (uint32_t)__popc(L);
        T.starts_before = wave_incl_scan(nl) - nl;
    }
    T.my_pos = before ? i0 - (before - 1u) : (uint32_t)(P0 + i0 - r.ls0);       // (a list holds < 2^32 ids)
    // a block starts every 256 ids of a list: inside my 16 at most once, and only before my first list start
    const uint32_t jc = (0u - T.my_pos) & 255u;
    const uint32_t below_first = L ? (L & (0u - L)) - 1u : 0xFFFFu;
    T.cont = (jc < 16u ? 1u << jc : 0u) & below_first & valid;
    if (T.cont) T.cont_first = p.values[P0 + i0 + jc];         // (asked for now, needed after the wait)
    const uint32_t keep = valid & ~(L | T.cont);               // ids that are written as a gap
    // ---- walk A: gaps and varint lengths
    uint32_t pr = T.prev0;
#pragma unroll
    for (uint32_t j = 0; j < ES_PER_LANE; j++) {
        const uint32_t gap = T.v[j] - pr;
        pr = T.v[j];
        const uint32_t len = es_varint_len(gap) & (uint32_t)__builtin_amdgcn_sbfe((int)keep, j, 1u);
        if (j < 8u) T.lens_lo |= len << (3u * j); else T.lens_hi |= len << (3u * (j - 8u));
    }
    T.lane_bytes = es_sum3(T.lens_lo) + es_sum3(T.lens_hi);
}
// walk B: my varints into the wave's LDS stage
__device__ __forceinline__ void es_tile_stage(const EsTile &T, uint8_t *st) {
    if (T.bytes == 0u) return;
    const uint32_t x = T.lens_lo | T.lens_hi;
    uint32_t q = T.lane_off;
    uint32_t pr = T.prev0;
    if (__ballot((x & 066666666u) != 0u) == 0ull) {          // the usual tile: one byte per gap
#pragma unroll
        for (uint32_t j = 0; j < ES_PER_LANE; j++) {
            const uint32_t len = j < 8u ? (T.lens_lo >> (3u * j)) & 7u : (T.lens_hi >> (3u * (j - 8u))) & 7u;
            const uint32_t gap = T.v[j] - pr;
            pr = T.v[j];
            if (len) st[q] = (uint8_t)gap;
            q += len;
        }
    } else {
        const bool four = __ballot((x & 044444444u) != 0u) != 0ull;                      // a varint of 4 or 5 bytes somewhere in the tile
        const bool three = four || __ballot((x & (x >> 1) & 011111111u) != 0u) != 0ull;  // ... of 3
#pragma unroll
        for (uint32_t j = 0; j < ES_PER_LANE; j++) {
            const uint32_t len = j < 8u ? (T.lens_lo >> (3u * j)) & 7u : (T.lens_hi >> (3u * (j - 8u))) & 7u;
            const uint32_t gap = T.v[j] - pr;
            pr = T.v[j];
            if (len) st[q] = (uint8_t)((gap & 0x7Fu) | (len > 1u ? 0x80u : 0u));
            if (len > 1u) st[q + 1u] = (uint8_t)(((gap >> 7) & 0x7Fu) | (len > 2u ? 0x80u : 0u));
            if (three) {
                if (len > 2u) st[q + 2u] = (uint8_t)(((gap >> 14) & 0x7Fu) | (len > 3u ? 0x80u : 0u));
                if (four) {
                    if (len > 3u) st[q + 3u] = (uint8_t)(((gap >> 21) & 0x7Fu) | (len > 4u ? 0x80u : 0u));
                    if (len > 4u) st[q + 4u] = (uint8_t)(gap >> 28);
                }
            }
            q += len;
        }
    }
}
// the tile's bytes leave as aligned 16-byte stores (LDS reads at any byte offset: five words + alignbyte), ragged ends byte by
// byte; then the skip entries and owners of the blocks that start in the tile (walk C).  base = global byte offset of the tile
__device__ __forceinline__ void es_tile_flush(const EncStreamParams &p, const EsTile &T, const uint8_t *st, unsigned long long base) {
    if (T.nloc == 0u) return;
    const int l = lane_id();
    const uint32_t i0 = ES_PER_LANE * (uint32_t)l;
    if (T.bytes != 0u) {
        uint8_t *dst = p.payload + base;
        const uint32_t mis = (uint32_t)((uintptr_t)dst & 15u);
        const uint32_t head = mis ? (16u - mis < T.bytes ? 16u - mis : T.bytes) : 0u;      // bytes before the first aligned chunk
        if ((uint32_t)l < head) dst[l] = st[l];
        const uint32_t body = (T.bytes - head) & ~15u;
        const LdsBytes16 src{st};
        for (uint32_t o = 16u * (uint32_t)l; o < body; o += 1024u) {
            const uint4 w = src(head + o);
            *reinterpret_cast<uint4 *>(dst + head + o) = w;
        }
        const uint32_t tail0 = head + body;
        if (tail0 + (uint32_t)l < T.bytes) dst[tail0 + l] = st[tail0 + l];
    }
    // ---- blocks that start inside a list: the lane that holds the first id knows everything but the list's number - the
    //      tile's first list unless a list started before it in the tile (then: that many lists on, empty ones skipped)
    if (T.cont) {
        const uint32_t jc = (uint32_t)__builtin_ctz(T.cont);
        const uint64_t pos = T.P0 + i0 + jc;
        uint64_t lst = T.l0;
        uint32_t b0 = T.blk_l0;
        if (T.starts_before) {
            lst = es_owner(p.post_off, p.n_lists, T.l0 + T.starts_before, pos);
            b0 = p.blk_off[lst];
        }
        const uint32_t b = b0 + ((T.my_pos + jc) >> 8);
        ii2_skip e;
        e.first_doc = T.cont_first;
        e.byte_off = (uint32_t)(base + T.lane_off + es_bytes_before(T.lens_lo, T.lens_hi, jc));
        p.skip[b] = e;
        p.blk_list[b] = (uint32_t)lst;
    }
    // ---- blocks that start a list: lane i takes list l0 + 1 + i again (as in the marking loop) - its first block is
    //      blk_off[list], its first id values[post_off[list]], and the byte offset comes from the lane that holds that position
    if (T.any_list_start) {
        const uint64_t P1 = T.P0 + T.nloc;
        for (uint64_t i = (uint64_t)l;; i += 64u) {
            const uint64_t li = T.l0 + 1ull + i;
            const uint64_t s = li <= p.n_lists ? p.post_off[li] : ~0ull;
            const uint64_t s1 = li < p.n_lists ? p.post_off[li + 1ull] : 0ull;
            const bool in = s < P1;
            const bool mine = in && s1 > s;
            const uint32_t rel = mine ? (uint32_t)(s - T.P0) : 0u;
            const int src = (int)((rel >> 4) << 2);
            const uint32_t o_off = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)T.lane_off);      // (all lanes: the sources must be active)
            const uint32_t o_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)T.lens_lo);
            const uint32_t o_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)T.lens_hi);
            if (mine) {
                const uint32_t b = p.blk_off[li];
                ii2_skip e;
                e.first_doc = p.values[s];
                e.byte_off = (uint32_t)(base + o_off + es_bytes_before(o_lo, o_hi, rel & 15u));
                p.skip[b] = e;
                p.blk_list[b] = (uint32_t)li;
            }
            if (__ballot(!in) != 0ull) break;
        }
    }
}

// A wave encodes TWO consecutive tiles behind ONE wait: what a workgroup waits for is the slowest of the few hundred workgroups
// started just before it (their loads, not their arithmetic - ~4 polls of 1.5 us with one tile per wave), and the second tile's
// bytes need no second wait.  The stage is used by the tiles one after the other.
template <bool STAMPS> __global__ __launch_bounds__(256) void k_enc_stream(EncStreamParams p) {
    __shared__ __align__(16) uint8_t lds[4][ES_WAVE_LDS];
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long wg_off;
    __shared__ uint32_t wg_err;
    const int l = lane_id();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x;
    const uint64_t P0 = ((uint64_t)g * 4u + wv) * (ES_TILES * ES_WAVE);           // my wave's first output position
    uint32_t *lm = reinterpret_cast<uint32_t *>(lds[wv]);            // [2][64] bit j of word i: a list starts at the tile's position 16 i + j
    uint8_t *st = lds[wv];                                           // (afterwards: a tile's bytes)
    if (threadIdx.x == 0) wg_err = 0u;
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = STAMPS && p.debug != nullptr;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();

    EsTile A, B;
    es_tile_begin(p, A, P0, lm);
    es_tile_begin(p, B, P0 + ES_WAVE, lm + 64);
    II2_STAMP(0)              // ids, the ids before, the tiles' first lists
    const EsListLoads ra = es_tile_lists_ask(p, A);
    const EsListLoads rb = es_tile_lists_ask(p, B);
    es_tile_lists(p, A, ra, lm);
    es_tile_lists(p, B, rb, lm + 64);
    II2_STAMP(1)              // list starts, walk A
    {
        const uint32_t ia = wave_incl_scan(A.lane_bytes), ib = wave_incl_scan(B.lane_bytes);
        A.bytes = wave_bcast(ia, 63); B.bytes = wave_bcast(ib, 63);
        A.lane_off = ia - A.lane_bytes; B.lane_off = ib - B.lane_bytes;
    }
    const bool wave_big = A.bytes > ES_STAGE_BYTES || B.bytes > ES_STAGE_BYTES;           // (only ids that do not ascend)
    if (l == 0) wcnt[wv] = (A.bytes + B.bytes) | (wave_big ? 0x80000000u : 0u);
    lds_barrier();
    II2_STAMP(3)              // the other waves
    const uint32_t r0 = wcnt[0], r1 = wcnt[1], r2 = wcnt[2], r3 = wcnt[3];
    const uint32_t c0 = r0 & 0x7FFFFFFFu, c1 = r1 & 0x7FFFFFFFu, c2 = r2 & 0x7FFFFFFFu, c3 = r3 & 0x7FFFFFFFu;
    const uint32_t total = c0 + c1 + c2 + c3;
    const uint32_t before_w = wv == 0u ? 0u : wv == 1u ? c0 : wv == 2u ? c0 + c1 : c0 + c1 + c2;
    const bool big = ((r0 | r1 | r2 | r3) & 0x80000000u) != 0u;
    if (threadIdx.x == 0) {
        if (big) {                                  // the failure first, and at the memory before the amount is on its way:
            lb_fail(p.lb);                          // whoever sums my amount in also finds the error word set
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        lb_publish(p.lb, g, total);
    }
    if (wv == 1u && lb_is_leader(g, gridDim.x)) {
        if (!lb_group_publish(p.lb, g, total) && l == 0) { wg_err = 1u; lb_fail(p.lb); }
    }
    // Both tiles before the wait when they fit the stage together (the usual case: ~1900 bytes each) - twice the work between the
    // publish and the question; otherwise the second tile takes the stage after the first has left.
    const uint32_t off_b = (A.bytes + 15u) & ~15u;
    const bool both = off_b + B.bytes <= ES_STAGE_BYTES;          // (wave-uniform)
    if (!big) {
        es_tile_stage(A, st);
        if (both) es_tile_stage(B, st + off_b);
    }
    II2_STAMP(4)              // publish, walk B of the first tile
    // ---- where the workgroup's bytes begin
    if (wv == 0u) {
        unsigned long long pre = 0ull;
        uint32_t polls = 0;
        uint32_t pm = 0;
        const bool ok = STAMPS ? lb_prefix(p.lb, g, gridDim.x, total, &pre, &polls, &pm) : lb_prefix(p.lb, g, gridDim.x, total, &pre);
        tacc[7] = polls | ((unsigned long long)pm << 32);
        if (l == 0) {
            wg_off = ok ? pre : 0ull;
            if (!ok) { wg_err = 1u; lb_fail(p.lb); }
        }
    }
    lds_barrier();
    II2_STAMP(5)              // the offset (wave 0: look-back; the others: waiting for it)
    const bool err = wg_err != 0u;
    const unsigned long long base = wg_off + before_w;           // global byte offset of my wave's first byte
    const bool fits = wg_off + total <= p.payload_cap;
    if (g == gridDim.x - 1u && threadIdx.x == 0) {              // the last workgroup: totals, the closing skip entry, the padding
        const bool bad = err || big || lb_failed(p.lb) || !fits;
        const unsigned long long nbytes = wg_off + total;
        p.d_result[0] = bad ? ~0ull : nbytes;
        const uint64_t nblk = p.blk_off[p.n_lists];
        p.d_result[1] = nblk;
        if (!bad) {
            p.skip[nblk].first_doc = p.n ? p.values[p.n - 1u] : 0u;
            p.skip[nblk].byte_off = (uint32_t)nbytes;
            p.blk_list[nblk] = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < 16u; k++) p.payload[nbytes + k] = 0;
        }
    }
    if (A.nloc == 0u || err || !fits || big) return;
    es_tile_flush(p, A, st, base);
    II2_STAMP(6)              // first tile: flush, skip entries
    if (!both) es_tile_stage(B, st);
    es_tile_flush(p, B, both ? st + off_b : st, base + A.bytes);
    II2_STAMP(2)              // second tile: walk B, flush, skip entries
    if (stamps && l == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&p.debug[(uint64_t)(blockIdx.x % 2048u) * 8u + i], tacc[i]);
#undef II2_STAMP
}

hipError_t launch_enc_stream(const uint64_t *post_off, const uint32_t *values, const uint32_t *blk_off, uint64_t n_lists, uint64_t n,
                             ii2_skip *skip, uint8_t *payload, uint64_t payload_cap, uint32_t *blk_list, uint32_t *part, uint64_t *d_result,
                             const LookBack &lb, unsigned long long *debug, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t waves = (n + ES_WAVE - 1u) / ES_WAVE;
    hipLaunchKernelGGL(k_enc_partition, dim3((unsigned)((waves + 255u) / 256u)), dim3(256), 0, s, post_off, n_lists, n, part);
    EncStreamParams p;
    p.part = part;
    p.post_off = post_off; p.values = values; p.blk_off = blk_off; p.n_lists = n_lists; p.n = n;
    p.skip = skip; p.payload = payload; p.blk_list = blk_list; p.payload_cap = payload_cap; p.d_result = d_result; p.lb = lb; p.debug = debug;
    const uint64_t grid = (n + ES_WG - 1u) / ES_WG;
    if (debug) hipLaunchKernelGGL(k_enc_stream<true>, dim3((unsigned)grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_enc_stream<false>, dim3((unsigned)grid), dim3(256), 0, s, p);
    return hipGetLastError();
}
uint64_t enc_stream_workgroups(uint64_t n) { return (n + ES_WG - 1u) / ES_WG; }

}  // namespace ii2
