// intersect_and2.hip — AND of TWO lists that are dense together (BASELINE configs[1]: the 2-term query over a 100M-doc
// index), gfx950, wave64, no MFMA.  Semantics: SURVEY §8 a14 (ascending ids present in both lists, minus the
// tombstoned ones); reading without tombstones = /root/reference shard.go:72-75.
//
// intersect_dense.hip treats its 2..4 lists alike: every list is marked into a wave-private LDS bitmap, the bitmaps are
// ANDed, the result words go to HBM and a second kernel digs the set bits out of them one by one (a loop that runs as
// long as the fullest of 64 words has bits).  For two lists that is more work than the problem holds:
//   * only the LONGER list (A) is marked;
//   * the SHORTER list (B) paces the rounds — a wave owns 16 of its blocks, a row of four lanes per block, 64 payload
//     bytes per lane — and every B posting is TESTED against A's bitmap where it sits in the lane's registers: a
//     running add per gap byte, one LDS read, one bit-field extract.  The answers are one bit per B posting (H, 64 bits
//     per lane, bit j = posting 64 * row-lane + j of the row's block: the posting before the lane's bytes, then its
//     bytes 0..62);
//   * the result ids ARE B postings: the second kernel re-walks the lane's bytes with H as the write mask — no bitmap
//     round trip through HBM (4 MB of H instead of 2 x 13 MB of result words), no per-bit extraction loop, and every
//     lane has the same 64 steps (no fullest-word imbalance).
// Exact for any input the general kernel accepts: A's blocks go through the same marking as intersect_dense.hip (multi-
// byte gaps, wide groups, short blocks), rounds wider than the LDS window are split, B blocks with multi-byte gaps are
// decoded by the whole wave and answer by posting index through a small LDS scratch.
#include <algorithm>

#include "dv1_device.h"
#include "internal.h"
#include "lookback.h"

namespace ii2 {

constexpr uint32_t A2_ROWS = 16;                              // blocks per pass: one per row of four lanes, 64 payload bytes per lane
constexpr uint32_t A2_GU = 512;                               // guard bits below and above a window (a lane of narrow groups spans < 512 docs)
constexpr uint32_t A2_CAPW = DENSE_CAPW;                      // docs a window covers (multiple of 32)
constexpr uint32_t A2_NW = ((A2_CAPW + 2 * A2_GU) / 32 + 2 + 3) & ~3u;    // words of the LDS bitmap (a multiple of 4: 16-byte clears)
constexpr uint32_t A2_HS = A2_ROWS * 8;                       // hard rows of B: 256 answer bits per row
constexpr uint32_t A2_WAVE_LDS = A2_NW + A2_HS;
constexpr uint32_t A2_STAGE = A2_ROWS * 256 + 8;              // u16 ids a wave stages (every posting of 16 full blocks + the flush's read-ahead)

__device__ __forceinline__ uint32_t a2_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// first index i in [0, n) with skip[i].first_doc > x, searched 64 ways per round by the whole wave
__device__ __forceinline__ uint32_t a2_skip_upper_bound(const ii2_skip *__restrict__ skip, uint32_t n, uint32_t x) {
    const uint32_t l = (uint32_t)lane_id();
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t sp = hi - lo;
        const uint32_t st = (sp + 63u) >> 6;
        const uint32_t pos = lo + l * st;
        const bool in = pos < hi;
        const uint32_t f = in ? skip[pos].first_doc : 0u;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(in && f <= x));     // probes are ascending: the matches are a prefix
        const uint32_t nin = (uint32_t)__popcll(__ballot(in));
        if (st == 1u) return cnt < nin ? lo + cnt : hi;
        const uint32_t nlo = cnt ? lo + (cnt - 1u) * st + 1u : lo;
        hi = cnt < nin ? lo + cnt * st : hi;
        lo = nlo;
    }
    return lo;
}

struct A2Bytes { uint4 g[4]; };

// my row's skip entry and the next one, for the blocks [at, at + 16) of a list (indices clamped onto the readable entry nblk)
__device__ __forceinline__ uint4 a2_ent_load(const ListView &L, uint32_t at, uint32_t row) {
    const uint32_t i0 = at + row < L.nblk ? at + row : L.nblk;
    const uint32_t i1 = i0 < L.nblk ? i0 + 1u : L.nblk;
    const ii2_skip e0 = L.skip[i0], e1 = L.skip[i1];
    return make_uint4(e0.first_doc, e0.byte_off, e1.first_doc, e1.byte_off);
}

// 64 payload bytes of my row's block [q0, q1): lane rl of the row takes bytes [64 rl, 64 rl + 64) (zero past the block's end at 16-byte grain)
__device__ __forceinline__ A2Bytes a2_fetch(const uint8_t *payload, bool rv, uint32_t q0, uint32_t q1, uint32_t rl) {
    const uint32_t len = rv ? q1 - q0 : 0u;
    A2Bytes B;
    const uint8_t *src = payload + q0 + 64u * rl;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        B.g[k] = make_uint4(0, 0, 0, 0);
        if (len > 64u * rl + 16u * k) __builtin_memcpy(&B.g[k], src + 16u * k, 16);   // segments carry 16 bytes of padding
    }
    return B;
}

// The lane's 64 bytes as sixteen words.  A FULL block of one-byte gaps has 255 payload bytes: the row's last lane holds one byte
// that is not its own (zeroed here).  Every other block of a valid row — a list's short last block, a block with a multi-byte
// gap — is `hard`: it goes through the general wave decoder, which takes any block.  Returns the ballot of the hard rows' lanes.
__device__ __forceinline__ unsigned long long a2_prep(const A2Bytes &B, bool rv, uint32_t len, uint32_t rl, uint32_t (&ww)[16]) {
    ww[0] = B.g[0].x; ww[1] = B.g[0].y; ww[2] = B.g[0].z; ww[3] = B.g[0].w;
    ww[4] = B.g[1].x; ww[5] = B.g[1].y; ww[6] = B.g[1].z; ww[7] = B.g[1].w;
    ww[8] = B.g[2].x; ww[9] = B.g[2].y; ww[10] = B.g[2].z; ww[11] = B.g[2].w;
    ww[12] = B.g[3].x; ww[13] = B.g[3].y; ww[14] = B.g[3].z;
    ww[15] = rl == 3u ? (B.g[3].w & 0x00FFFFFFu) : B.g[3].w;
    uint32_t any = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; k++) any |= ww[k];
    const bool hard = rv && (len != 255u || (any & 0x80808080u) != 0u);
    return __ballot(hard);
}

// id of the posting right before my bytes: the block's first doc + the gap sums of the row's lanes before me (rows of
// one-byte gaps only; `live` = my row is one).  *wide_or = OR of my sixteen group sums (a bit above bit 4 = some group of four
// postings spans 32 docs or more).
__device__ __forceinline__ uint32_t a2_group_sum(uint32_t x) {      // sum of a word's four bytes; opaque, so that the compiler recomputes it where it
    uint32_t g;                                                       // is needed instead of keeping sixteen of them (or their prefix sums) in registers
    asm volatile("v_sad_u8 %0, %1, 0, 0" : "=v"(g) : "v"(x));
    return g;
}
__device__ __forceinline__ uint32_t a2_lane_base(const uint32_t (&ww)[16], bool live, uint32_t f, uint32_t rl, uint32_t *wide_or) {
    uint32_t acc = 0, wide = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; k += 2u) {                             // (two groups per step: three-input or / add)
        const uint32_t g0 = a2_group_sum(ww[k]), g1 = a2_group_sum(ww[k + 1u]);
        wide |= g0 | g1;
        acc += g0 + g1;
    }
    *wide_or = wide;
    const uint32_t s = live ? acc : 0u;
    const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x90 /* quad_perm [0,0,1,2] */, 0xf, 0xf, false);
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x40 /* quad_perm [0,0,0,1] */, 0xf, 0xf, false);
    const uint32_t s3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, false);
    const uint32_t excl = rl == 0u ? 0u : rl == 1u ? s1 : rl == 2u ? s1 + s2 : s1 + s2 + s3;
    return f + excl;
}

// ---- marking one pass (up to sixteen blocks of A, a row of four lanes each) into bm for the window [wlo, wlo + wspan] ----
// Same scheme as intersect_dense.hip: (M << gap) | 1 builds the mask of four postings plus the posting before them, one 64-bit
// shift positions it, two ds_or put it into the bitmap; groups wider than 31 docs and lanes that start outside the window are
// placed posting by posting; rows with multi-byte gaps go through the wave decoder.  Exact for any block.
__device__ __forceinline__ void a2_mark_rows(uint32_t *lds_all, uint32_t *bm, const uint8_t *payload, bool rv, uint32_t f, uint32_t q0,
                                             uint32_t q1, const A2Bytes &B, uint32_t wlo, uint32_t wspan, uint32_t rl, uint32_t row) {
    const uint32_t len = rv ? q1 - q0 : 0u;
    const uint32_t myoff = 64u * rl;
    const uint32_t nb = len > myoff ? (len - myoff < 64u ? len - myoff : 64u) : 0u;
    uint32_t ww[16];
    const unsigned long long hm = a2_prep(B, rv, len, rl, ww);
    auto setbit = [&](uint32_t id, bool valid) {                       // exact range test: any id, any gap
        const uint32_t d = id - wlo;
        if (valid && d <= wspan) atomicOr(&bm[(d + A2_GU) >> 5], 1u << ((d + A2_GU) & 31u));
    };
    if (hm != 0ull) {
#pragma unroll 1
        for (uint32_t r = 0; r < A2_ROWS; r++) {
            if (((hm >> (4u * r)) & 0xFull) == 0ull) continue;
            const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)(4u * r)),
                           q1r = (uint32_t)__builtin_amdgcn_readlane((int)q1, (int)(4u * r));
            decode_block_wave4(GlobalBytes{payload}, q0r, q1r, fq,
                               [&](uint32_t, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                   setbit(id0, mask & 1u); setbit(id1, mask & 2u); setbit(id2, mask & 4u); setbit(id3, mask & 8u);
                               });
        }
    }
    const bool rowhard = ((hm >> (4u * row)) & 0xFull) != 0ull;
    const bool live = rv && !rowhard;
    uint32_t wide;
    const uint32_t base = a2_lane_base(ww, live, f, rl, &wide);         // id of the posting right before my bytes
    const uint32_t u = base - wlo + A2_GU;                               // its (guard-shifted) position, mod 2^32
    // A lane whose groups of four postings all span < 32 docs spans < 512 docs in all: if it starts outside
    // [wlo - GU, wlo + wspan] it lies wholly outside the window and is skipped.  A lane with a wider group is kept
    // whatever its start; when it starts outside that range every one of its groups is placed posting by posting.
    const bool inrange = u <= wspan + A2_GU;
    const bool haswide = wide >= 32u;
    const bool act = live && (inrange || haswide);
    const bool exactlane = haswide && !inrange;
    const uint32_t lim = wspan + 2u * A2_GU - 64u;                       // a position in the upper guard: where out-of-window groups are parked
    const uint32_t bmbits = (uint32_t)(bm - lds_all) * 32u;              // the bitmap's place inside the workgroup's LDS array, folded into the position
    if (__ballot(act && haswide) == 0ull) {
        // the usual pass: every lane that marks starts inside [wlo - GU, wlo + wspan] and spans < 512 docs, so no group needs
        // the clamp, the wide-group test or the second loop (the bitmap's upper guard holds 512 + 32 bits past the window)
        if (act) {
            uint32_t q = u + bmbits;
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) {
                const uint32_t x = ww[k];
                uint32_t M = (1u << ((x >> 24) & 31u)) | 1u;
                M = (M << ((x >> 16) & 31u)) | 1u;
                M = (M << ((x >> 8) & 31u)) | 1u;
                M = (M << (x & 31u)) | 1u;                               // bit 0: the posting before the group
                uint32_t *dst = lds_all + (q >> 5);
                const unsigned long long m64 = (unsigned long long)M << (q & 31u);
                atomicOr(dst, (uint32_t)m64);
                atomicOr(dst + 1, (uint32_t)(m64 >> 32));
                q += a2_group_sum(x);
            }
        }
        return;
    }
    if (act) {
        uint32_t q = u + bmbits;
        const uint32_t limb = lim + bmbits;
#pragma unroll
        for (uint32_t k = 0; k < 16u; k++) {
            const uint32_t x = ww[k];
            const uint32_t gs = a2_group_sum(x);
            const bool isw = exactlane || gs >= 32u;                     // the four gaps do not fit one 32-bit mask: placed by the loop below
            uint32_t M = (1u << ((x >> 24) & 31u)) | 1u;
            M = (M << ((x >> 16) & 31u)) | 1u;
            M = (M << ((x >> 8) & 31u)) | 1u;
            M = (M << (x & 31u)) | 1u;                                   // bit 0: the posting before the group
            M = isw ? 0u : M;
            const uint32_t qc = q < limb ? q : limb;                     // groups beyond the window: harmless bits in the guard
            const uint32_t sh = qc & 31u;
            uint32_t *dst = lds_all + (qc >> 5);
            const unsigned long long m64 = (unsigned long long)M << sh;    // (one 64-bit shift: both words)
            atomicOr(dst, (uint32_t)m64);
            atomicOr(dst + 1, (uint32_t)(m64 >> 32));                    // the part that spills into the next word (0 for most)
            q += gs;
        }
    }
    if (__ballot(act && haswide) != 0ull) {
        uint32_t prev = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 16u; k++) {
            const uint32_t gs = a2_group_sum(ww[k]);
            const bool isw = act && haswide && (exactlane || gs >= 32u);
            if (__ballot(isw) != 0ull) {
                if (isw) {
                    const uint32_t x = ww[k];
                    uint32_t d = u + prev - A2_GU;                       // doc - wlo of the posting before the group (mod 2^32)
                    if (k == 0u && rl == 0u && d <= wspan) atomicOr(&bm[(d + A2_GU) >> 5], 1u << ((d + A2_GU) & 31u));   // the block's first doc
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) {
                        d += (x >> (8u * j)) & 0xFFu;
                        if (4u * k + j < nb && d <= wspan) atomicOr(&bm[(d + A2_GU) >> 5], 1u << ((d + A2_GU) & 31u));
                    }
                }
            }
            prev += gs;
        }
    }
}

// What both kernels derive, identically, from a wave's sixteen B blocks: the lane's bytes, which rows are hard, the id before
// the lane's bytes and which of the lane's 64 postings exist.
struct A2Lane {
    uint32_t ww[16];
    unsigned long long hm;       // ballot of the hard rows' lanes (wave-uniform)
    bool live;                   // my row has a block of one-byte gaps
    uint32_t base;               // id of the posting right before my bytes (live lanes)
    unsigned long long valid;    // bit j: posting 64 rl + j of my row's block exists (live lanes; 0 otherwise)
};
__device__ __forceinline__ void a2_lane_setup(A2Lane &L, const A2Bytes &B, bool rv, const uint4 &E, uint32_t rl, uint32_t row) {
    uint32_t q1 = E.w;
    asm volatile("" : "+v"(q1));            // (opaque: everything below stays where the call is — hoisted above the marking passes it would hold registers there)
    const uint32_t len = rv ? q1 - E.y : 0u;
    L.hm = a2_prep(B, rv, len, rl, L.ww);
    const bool rowhard = ((L.hm >> (4u * row)) & 0xFull) != 0ull;
    L.live = rv && !rowhard;
    uint32_t wide;
    L.base = a2_lane_base(L.ww, L.live, E.x, rl, &wide);
    // a live row's block is full (256 postings): mine are 64 rl ... 64 rl + 63 — the one before my bytes, then my bytes 0..62
    L.valid = L.live ? ~0ull : 0ull;
}

template <bool RANGE>
__device__ __forceinline__ unsigned long long a2_test_lane(const uint32_t *lds_all, uint32_t bmbits, const A2Lane &L, uint32_t wlo, uint32_t wspan) {
    // u = bit position, inside the workgroup's LDS array, of the posting before my bytes (guard shift and the bitmap's own offset
    // folded in once); lanes without a block probe the window's first bit.  Per posting: one add (the gap byte selected by SDWA),
    // shift + mask for the word's address, the LDS read, one bit-field extract, one shift-or into the answer word.
    const uint32_t org = A2_GU + bmbits;
    uint32_t u = (L.live ? L.base - wlo : 0u) + org;
    uint32_t hlo = 0, hhi = 0;
    auto probe = [&](uint32_t j) {
        uint32_t uu = u;
        bool in = true;
        if (RANGE) { in = u - org <= wspan; uu = in ? u : org; }
        const uint32_t wd = lds_all[uu >> 5];
        uint32_t hit = __builtin_amdgcn_ubfe(wd, uu, 1u);           // (the offset operand is taken mod 32)
        if (RANGE) hit = in ? hit : 0u;
        if (j < 32u) hlo |= hit << j; else hhi |= hit << (j - 32u);
    };
    probe(0u);
#pragma unroll
    for (uint32_t k = 0; k < 63u; k++) {
        u += (L.ww[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
        probe(k + 1u);
    }
    return (((unsigned long long)hhi << 32) | hlo) & L.valid;
}

// ---- kernel 1: mark A, test B, one answer bit per B posting -------------------------------------------------------
__device__ __forceinline__ void and2_tiles_body(const DenseParams &p) {
    __shared__ __align__(16) uint32_t lds[4][A2_WAVE_LDS];
    __shared__ uint32_t wcnt[4];
    const int l = lane_id();
    const uint32_t wv = a2_uni(threadIdx.x >> 6);
    const uint32_t rl = (uint32_t)l & 3u, row = (uint32_t)l >> 2;
    const uint32_t w = blockIdx.x * 4u + wv;                 // this wave's number in doc order
    uint32_t *lds_all = &lds[0][0];
    uint32_t *bmA = lds[wv], *hs = bmA + A2_NW;
    const uint32_t bmbits = (uint32_t)(bmA - lds_all) * 32u;
    const ListView LB = p.lists[0], LA = p.lists[1];
    const uint32_t b0 = w * A2_ROWS;
    const uint32_t b1 = b0 + A2_ROWS < LB.nblk ? b0 + A2_ROWS : LB.nblk;
    uint32_t count = 0, lo_w = 0, hi_w = 0, flags = 0;
    unsigned long long H = 0ull;
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr && !p.debug_expand;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();

    if (b0 < b1) {                                           // wave-uniform
        const uint32_t nvB = b1 - b0;
        const bool rvB = row < nvB;
        const uint4 EB = a2_ent_load(LB, b0, row);
        const uint32_t b_last = a2_uni(*LB.last_doc);
        const uint32_t lo = a2_uni(EB.x);                                       // (row 0 = block b0)
        const uint32_t nf = (uint32_t)__builtin_amdgcn_readlane((int)EB.z, (int)(4u * (nvB - 1u)));
        const uint32_t hi = b1 < LB.nblk ? nf - 1u : b_last;
        lo_w = lo; hi_w = hi;
        // where A enters this wave's doc range: the last block that starts at or before lo.  One probe of 64 entries around
        // a linear guess first (the lists of a dense query are close to uniform: it nearly always brackets the answer and
        // costs one memory round trip instead of three); the full search otherwise
        uint32_t a0;
        {
            const uint32_t fj = p.first_doc[1], lj = p.last_doc[1];
            uint32_t ub = 0xFFFFFFFFu;
            if (LA.nblk > 64u && lj > fj) {
                const uint64_t rel = lo > fj ? (uint64_t)(lo - fj) : 0ull;
                uint64_t gss = rel * LA.nblk / ((uint64_t)(lj - fj) + 1ull);
                if (gss > LA.nblk) gss = LA.nblk;
                uint32_t wb = gss > 32ull ? (uint32_t)gss - 32u : 0u;
                if (wb + 64u > LA.nblk) wb = LA.nblk - 64u;
                const uint32_t fdoc = LA.skip[wb + (uint32_t)l].first_doc;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(fdoc <= lo));   // first docs ascend: the matches are a prefix
                if ((cnt > 0u || wb == 0u) && (cnt < 64u || wb + 64u == LA.nblk)) ub = wb + cnt;
            }
            if (ub == 0xFFFFFFFFu) ub = a2_skip_upper_bound(LA.skip, LA.nblk, lo);
            ub = a2_uni(ub);
            a0 = ub ? ub - 1u : 0u;
        }
        II2_STAMP(0)          // prologue: B's entries, search

        uint32_t wlo = lo & ~31u;
        uint32_t a_cur = a0;
        bool first_window = true;
        unsigned long long hm_any = 0ull;
        for (;;) {
            const uint32_t wspan = hi - wlo < A2_CAPW ? hi - wlo : A2_CAPW - 1u;
            const uint32_t whi = wlo + wspan;
            const uint32_t nw = (wspan >> 5) + 1u;
            const uint32_t ncl = nw + 2u * (A2_GU / 32u) + 2u;       // <= A2_NW - 2; cleared in 4-word steps
            // ---- A's passes over this window.  The entries of the first two passes are requested together; pass k is marked
            // while the payload of pass k + 1 is in flight — and behind A's LAST pass it is B's payload (and the window's
            // tombstone words) that is in flight: B's bytes only take registers once A's are nearly done with theirs.
            uint32_t cur = a_cur;
            uint4 Ecur = a2_ent_load(LA, cur, row), Enext = a2_ent_load(LA, cur + A2_ROWS, row);
            for (uint32_t i = 4u * (uint32_t)l; i < ncl; i += 256u) *reinterpret_cast<uint4 *>(&bmA[i]) = make_uint4(0, 0, 0, 0);
            bool rv = cur + row < LA.nblk && Ecur.x <= whi;          // first docs ascend: the valid rows are a prefix
            uint32_t nv = (uint32_t)__popcll(__ballot(rv)) >> 2;
            A2Bytes Bcur = a2_fetch(LA.payload, rv, Ecur.y, Ecur.w, rl);
            A2Bytes X;
            uint32_t tw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            II2_STAMP(1)      // clear, A's entries, first fetch
            for (;;) {
                const bool more = nv == A2_ROWS;
                const bool rvn = more && cur + A2_ROWS + row < LA.nblk && Enext.x <= whi;
                const uint32_t nvn = (uint32_t)__popcll(__ballot(rvn)) >> 2;
                const bool last = nvn == 0u;
                X = a2_fetch(last ? LB.payload : LA.payload, last ? rvB : rvn, last ? EB.y : Enext.y, last ? EB.w : Enext.w, rl);
                if (last && p.tomb) {                                // (a window holds <= 512 words: eight per lane, requested together)
                    const uint32_t tw0 = wlo >> 5;
#pragma unroll
                    for (uint32_t k = 0; k < 8u; k++) {
                        const uint32_t i = 64u * k + (uint32_t)l;
                        tw[k] = (i < nw && tw0 + i < p.tomb_nwords) ? p.tomb[tw0 + i] : 0u;
                    }
                }
                if (nv != 0u) a2_mark_rows(lds_all, bmA, LA.payload, rv, Ecur.x, Ecur.y, Ecur.w, Bcur, wlo, wspan, rl, row);
                cur += nv;
                if (last) break;
                Ecur = Enext; Bcur = X; rv = rvn; nv = nvn;
                Enext = a2_ent_load(LA, cur + A2_ROWS, row);
            }
            if (cur > a_cur + 1u) a_cur = cur - 1u;                  // A's last block in the window may reach past it
            II2_STAMP(2)      // mark A
            if (p.tomb) {                                            // removed docs are absent from A: no B posting finds them
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    const uint32_t i = 64u * k + (uint32_t)l;
                    if (64u * k < nw && tw[k] != 0u) bmA[A2_GU / 32u + i] &= ~tw[k];
                }
            }
            II2_STAMP(3)      // tombstones
            // ---- B's postings against the bitmap
            A2Lane LN;
            a2_lane_setup(LN, X, rvB, EB, rl, row);
            hm_any = LN.hm;
            if (LN.hm != 0ull && first_window) {                     // answers of the hard rows are collected in LDS
                hs[2 * l] = 0u; hs[2 * l + 1] = 0u;
                flags |= 2u;
            }
            const bool single = first_window && hi - wlo < A2_CAPW;
            if (single) H |= a2_test_lane<false>(lds_all, bmbits, LN, wlo, wspan);
            else H |= a2_test_lane<true>(lds_all, bmbits, LN, wlo, wspan);
            if (LN.hm != 0ull) {
#pragma unroll 1
                for (uint32_t r = 0; r < A2_ROWS; r++) {
                    if (((LN.hm >> (4u * r)) & 0xFull) == 0ull) continue;
                    const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)EB.x, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)EB.y, (int)(4u * r)),
                                   q1r = (uint32_t)__builtin_amdgcn_readlane((int)EB.w, (int)(4u * r));
                    uint32_t *hr = hs + 8u * r;
                    auto probe = [&](uint32_t ix, uint32_t id) {
                        const uint32_t d = id - wlo;
                        if (d <= wspan && ix < 256u) {
                            const uint32_t u = d + A2_GU;
                            if ((bmA[u >> 5] >> (u & 31u)) & 1u) atomicOr(&hr[ix >> 5], 1u << (ix & 31u));
                        }
                    };
                    decode_block_wave4(GlobalBytes{LB.payload}, q0r, q1r, fq,
                                       [&](uint32_t ix, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                           if (mask & 1u) { probe(ix, id0); ix++; }
                                           if (mask & 2u) { probe(ix, id1); ix++; }
                                           if (mask & 4u) { probe(ix, id2); ix++; }
                                           if (mask & 8u) { probe(ix, id3); ix++; }
                                       });
                }
            }
            II2_STAMP(4)      // test B
            if (hi - wlo < A2_CAPW) break;
            wlo += A2_CAPW;
            first_window = false;
        }
        if (hm_any != 0ull) {
            const bool rowhard = ((hm_any >> (4u * row)) & 0xFull) != 0ull;
            if (rowhard) H = ((unsigned long long)hs[8u * row + 2u * rl + 1u] << 32) | hs[8u * row + 2u * rl];
        }
        if (hi - lo > 0xFFFFu) flags |= 1u;                   // ids do not fit 16-bit offsets from the round's first doc
        count = wave_sum((uint32_t)__popcll(H));
        p.hmask[(size_t)w * 64u + (uint32_t)l] = make_uint2((uint32_t)H, (uint32_t)(H >> 32));
        II2_STAMP(5)          // count + store
    }
    if (stamps && l == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&p.debug[(uint64_t)(blockIdx.x % 2048u) * 8u + i], tacc[i]);
#undef II2_STAMP
    if (l == 0) {
        wcnt[wv] = count;
        if (w < p.n_meta) p.meta[w] = make_uint4(lo_w, hi_w, count, flags);
    }
    __syncthreads();
    if (threadIdx.x == 0) p.wg_sum[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

__global__ __launch_bounds__(256) void k_and2_tiles(DenseParams p) { and2_tiles_body(p); }

// ---- kernel 2: the B postings whose answer bit is set -> the final ascending id array ------------------------------
// Workgroup g re-walks the B blocks of tile workgroup g (same decomposition).  Its output offset = the counts of the
// workgroups before it, summed here by all 256 threads (a few thousand words) — no separate scan launch.
__global__ __launch_bounds__(256) void k_and2_expand(DenseParams p) {
    __shared__ __align__(16) uint16_t stage[4][A2_STAGE];   // ids of a round as offsets from the round's first doc, in output order
    __shared__ uint32_t hsx[4][A2_HS];
    __shared__ unsigned long long wsum[4];
    const int tid = (int)threadIdx.x, l = tid & 63;
    const uint32_t wv = a2_uni((uint32_t)tid >> 6);
    const uint32_t rl = (uint32_t)l & 3u, row = (uint32_t)l >> 2;
    const uint32_t w0 = blockIdx.x * 4u;
    const uint32_t w = w0 + wv;
    const ListView LB = p.lists[0];
    unsigned long long tacc[4] = {0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr && p.debug_expand;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();
    // everything this wave needs from memory is requested before the offset arithmetic
    const uint32_t b0 = w * A2_ROWS;
    const bool has = b0 < LB.nblk;
    const uint32_t b1 = !has ? b0 : (b0 + A2_ROWS < LB.nblk ? b0 + A2_ROWS : LB.nblk);
    const uint32_t nvB = b1 - b0;
    const bool rvB = row < nvB;
    const uint4 m = w < p.n_meta ? p.meta[w] : make_uint4(0, 0, 0, 0);       // {first doc, last doc, ids, flags}
    uint4 EB = make_uint4(0, 0, 0, 0);
    uint2 Hw = make_uint2(0, 0);
    if (has) {
        EB = a2_ent_load(LB, b0, row);
        Hw = p.hmask[(size_t)w * 64u + (uint32_t)l];
    }
    unsigned long long mine = 0;
    for (uint32_t g = (uint32_t)tid; g < blockIdx.x; g += 256u) mine += p.wg_sum[g];
    for (int d = 32; d >= 1; d >>= 1) mine += (unsigned long long)__shfl_xor((long long)mine, d, 64);
    if (l == 0) wsum[wv] = mine;
    uint32_t before = 0;                        // ids of the waves of this workgroup before mine
    for (uint32_t v = 0; v < wv; v++) before += p.meta[w0 + v].z;
    A2Bytes BB;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) BB.g[k] = make_uint4(0, 0, 0, 0);
    if (has && m.z != 0u) BB = a2_fetch(LB.payload, rvB, EB.y, EB.w, rl);
    __syncthreads();
    const unsigned long long off = wsum[0] + wsum[1] + wsum[2] + wsum[3] + before;
    if (blockIdx.x == gridDim.x - 1u && wv == 3u && l == 0) *p.d_count = off + m.z;     // the last wave: the total
    if (!has || m.z == 0u) return;
    II2_STAMP(0)      // prologue: meta, entries, answers, payload, offsets
    const uint32_t tot = m.z;
    const uint32_t lo = m.x;
    const bool wide = (m.w & 1u) != 0u;
    A2Lane LN;
    a2_lane_setup(LN, BB, rvB, EB, rl, row);
    const unsigned long long H = ((unsigned long long)Hw.y << 32) | Hw.x;
    const uint32_t pc = (uint32_t)__popcll(H);
    const uint32_t incl = wave_incl_scan(pc);
    const uint32_t q0 = incl - pc;              // my first id's place among the wave's ids
    const bool fits = off + tot <= p.out_cap;
    II2_STAMP(1)      // lane sums + scan
    uint16_t *st = stage[wv];
    // the easy rows: 64 steps, the same for every lane
    const unsigned long long He = LN.live ? H : 0ull;
    if (!wide) {
        uint32_t q = q0;
        uint32_t v = LN.base - lo;
        if (He & 1ull) { st[q] = (uint16_t)v; q++; }
#pragma unroll
        for (uint32_t k = 0; k < 63u; k++) {
            v += (LN.ww[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
            if ((He >> (k + 1u)) & 1ull) { st[q] = (uint16_t)v; q++; }
        }
    } else {                                    // a round wider than 64k docs (a sparse stretch): ids go out one by one
        uint32_t q = q0;
        uint32_t v = LN.base;
        if ((He & 1ull) && off + q < p.out_cap) p.out[off + q] = v;
        q += (uint32_t)(He & 1ull);
#pragma unroll
        for (uint32_t k = 0; k < 63u; k++) {
            v += (LN.ww[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
            const uint32_t hit = (uint32_t)((He >> (k + 1u)) & 1ull);
            if (hit && off + q < p.out_cap) p.out[off + q] = v;
            q += hit;
        }
    }
    // rows with multi-byte gaps: the answers are by posting index; the wave decodes the block again and every hit ranks itself
    if (LN.hm != 0ull) {
        uint32_t *hx = hsx[wv];
        hx[2 * l] = Hw.x; hx[2 * l + 1] = Hw.y;               // (lane 4 r + rl holds words 2 rl, 2 rl + 1 of row r: index 8 r + 2 rl = 2 l)
#pragma unroll 1
        for (uint32_t r = 0; r < A2_ROWS; r++) {
            if (((LN.hm >> (4u * r)) & 0xFull) == 0ull) continue;
            const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)EB.x, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)EB.y, (int)(4u * r)),
                           q1r = (uint32_t)__builtin_amdgcn_readlane((int)EB.w, (int)(4u * r));
            const uint32_t qrow = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)(4u * r));
            const uint32_t *hr = hx + 8u * r;
            auto place = [&](uint32_t ix, uint32_t id) {
                if (ix >= 256u) return;
                const uint32_t wd = hr[ix >> 5];
                if (!((wd >> (ix & 31u)) & 1u)) return;
                uint32_t rk = (uint32_t)__popc(wd & ((1u << (ix & 31u)) - 1u));
                for (uint32_t j = 0; j < (ix >> 5); j++) rk += (uint32_t)__popc(hr[j]);
                if (!wide) st[qrow + rk] = (uint16_t)(id - lo);
                else if (off + qrow + rk < p.out_cap) p.out[off + qrow + rk] = id;
            };
            decode_block_wave4(GlobalBytes{LB.payload}, q0r, q1r, fq,
                               [&](uint32_t ix, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                   if (mask & 1u) { place(ix, id0); ix++; }
                                   if (mask & 2u) { place(ix, id1); ix++; }
                                   if (mask & 4u) { place(ix, id2); ix++; }
                                   if (mask & 8u) { place(ix, id3); ix++; }
                               });
        }
    }
    II2_STAMP(2)      // stage
    if (!wide) {
        // written out 16 bytes per lane (wave-private LDS: program order is enough)
        for (uint32_t c = 4u * (uint32_t)l; c < tot; c += 256u) {
            const uint2 pk = *reinterpret_cast<const uint2 *>(&st[c]);
            const uint4 ids = make_uint4(lo + (pk.x & 0xFFFFu), lo + (pk.x >> 16), lo + (pk.y & 0xFFFFu), lo + (pk.y >> 16));
            if (c + 4u <= tot && fits) {
                uint32_t *dst = p.out + off + c;
                __builtin_memcpy(dst, &ids, 16);                    // one 16-byte store, any 4-byte alignment
            } else {
                const uint32_t vv[4] = {ids.x, ids.y, ids.z, ids.w};
                for (uint32_t k = 0; k < 4u; k++)
                    if (c + k < tot && off + c + k < p.out_cap) p.out[off + c + k] = vv[k];
            }
        }
    }
    II2_STAMP(3)      // flush
    if (stamps && l == 0)
        for (int i = 0; i < 4; i++) atomicAdd(&p.debug[(uint64_t)(blockIdx.x % 2048u) * 8u + i], tacc[i]);
#undef II2_STAMP
}


// ---- one launch: mark A, test B, place the ids ---------------------------------------------------------------------
// The two kernels above hand the answers over through HBM and the second one reads B again (entries, payload, answer bits:
// two dependent round trips before its first useful instruction).  Here the wave that found the answers still holds B's
// bytes and H in registers: it stages its ids (16-bit offsets, wave-private LDS, over the bitmap that is dead by then) and
// writes them out once it knows where — the number of ids of all workgroups before its own.
//
// That prefix comes from the look-back of lookback.h (two hops behind the slowest workgroup it depends on; every wait bounded:
// a workgroup that runs out of patience leaves without writing ids, the last workgroup then poisons the count - all ones - and
// the host repeats a failed call through the two kernels above).
constexpr uint32_t A2F_STAGE_W = (A2_STAGE + 1u) / 2u;            // stage in words (aliases the bitmap, which is dead by then)
constexpr uint32_t A2F_WAVE_LDS = (A2F_STAGE_W > A2_NW ? A2F_STAGE_W : A2_NW) + A2_HS;
static_assert(A2F_STAGE_W % 4u == 0u, "16-byte alignment of the per-wave LDS regions");

__global__ __launch_bounds__(256, 3) void k_and2_fused(DenseParams p) {
    __shared__ __align__(16) uint32_t lds[4][A2F_WAVE_LDS];
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long wg_off;
    __shared__ uint32_t wg_err;
    const int l = lane_id();
    const uint32_t wv = a2_uni(threadIdx.x >> 6);
    const uint32_t rl = (uint32_t)l & 3u, row = (uint32_t)l >> 2;
    const uint32_t g = blockIdx.x;
    const uint32_t w = g * 4u + wv;                          // this wave's number in doc order
    uint32_t *lds_all = &lds[0][0];
    uint32_t *bmA = lds[wv], *hs = bmA + (A2F_WAVE_LDS - A2_HS);
    uint16_t *st = reinterpret_cast<uint16_t *>(bmA);
    const uint32_t bmbits = (uint32_t)(bmA - lds_all) * 32u;
    const ListView LB = p.lists[0], LA = p.lists[1];
    const uint32_t b0 = w * A2_ROWS;
    const uint32_t b1 = b0 + A2_ROWS < LB.nblk ? b0 + A2_ROWS : LB.nblk;
    uint32_t count = 0, lo = 0, hi = 0;
    unsigned long long H = 0ull, hm_any = 0ull;
    A2Lane LN;
    uint4 EB = make_uint4(0, 0, 0, 0);
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) wg_err = 0u;
    const bool work = b0 < b1;                               // wave-uniform
    const uint32_t nvB = work ? b1 - b0 : 0u;
    const bool rvB = row < nvB;

    if (work) {
        // ---- first round trip: B's entries and, beside them, a 128-entry window of A's skip table around the place where a
        // uniform list would have this wave's first doc (B's docs-per-block and A's blocks-per-doc from the host: a guess that
        // the entries themselves confirm or refute)
        EB = a2_ent_load(LB, b0, row);
        const uint32_t fj = p.first_doc[1];
        uint32_t wb = 0;
        uint2 P0 = make_uint2(0, 0), P1 = make_uint2(0, 0);
        const bool probing = LA.nblk >= 128u;
        if (probing) {
            const float lo_guess = (float)p.first_doc[0] + (float)b0 * p.b_dpb;
            const float rel = lo_guess - (float)fj;
            float gs = rel > 0.f ? rel * p.a_scale : 0.f;
            if (gs > (float)LA.nblk) gs = (float)LA.nblk;
            const uint32_t gi = (uint32_t)gs;
            wb = gi > 40u ? gi - 40u : 0u;
            if (wb + 128u > LA.nblk) wb = LA.nblk - 128u;
            wb = a2_uni(wb);
            const ii2_skip e0 = LA.skip[wb + (uint32_t)l], e1 = LA.skip[wb + 64u + (uint32_t)l];
            P0 = make_uint2(e0.first_doc, e0.byte_off);
            P1 = make_uint2(e1.first_doc, e1.byte_off);
        }
        const uint32_t b_last = a2_uni(*LB.last_doc);
        lo = a2_uni(EB.x);                                                      // (row 0 = block b0)
        const uint32_t nf = (uint32_t)__builtin_amdgcn_readlane((int)EB.z, (int)(4u * (nvB - 1u)));
        hi = b1 < LB.nblk ? nf - 1u : b_last;
        // where A enters this wave's doc range: the last block that starts at or before lo
        uint32_t a0;
        bool from_probe = false;
        {
            uint32_t ub = 0xFFFFFFFFu;
            if (probing) {
                const uint32_t c0 = (uint32_t)__popcll(__ballot(P0.x <= lo)), c1 = (uint32_t)__popcll(__ballot(P1.x <= lo));   // first docs ascend: the matches are a prefix
                const uint32_t cnt = c0 + c1;
                if ((cnt > 0u || wb == 0u) && (cnt < 128u || wb + 128u == LA.nblk)) { ub = wb + cnt; from_probe = true; }
            }
            if (ub == 0xFFFFFFFFu) ub = a2_skip_upper_bound(LA.skip, LA.nblk, lo);
            ub = a2_uni(ub);
            a0 = ub ? ub - 1u : 0u;
        }
        II2_STAMP(0)          // prologue: B's entries, A's window, search

        uint32_t wlo = lo & ~31u;
        uint32_t a_cur = a0;
        bool first_window = true;
        for (;;) {
            const uint32_t wspan = hi - wlo < A2_CAPW ? hi - wlo : A2_CAPW - 1u;
            const uint32_t whi = wlo + wspan;
            const uint32_t nw = (wspan >> 5) + 1u;
            const uint32_t ncl = nw + 2u * (A2_GU / 32u) + 2u;       // <= A2_NW - 2; cleared in 4-word steps
            uint32_t cur = a_cur;
            uint4 Ecur, Enext;
            // the entries of A's first two passes: out of the window's registers when it holds them (no second round trip)
            if (first_window && from_probe && cur + 2u * A2_ROWS + 1u <= wb + 127u && cur + 2u * A2_ROWS + 1u <= LA.nblk) {
                auto pick = [&](uint32_t j, uint32_t &fd, uint32_t &bo) {      // entry wb + j, j < 128 (per lane)
                    const int src = (int)((j & 63u) << 2);
                    const uint32_t f0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)P0.x), f1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)P1.x);
                    const uint32_t o0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)P0.y), o1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)P1.y);
                    fd = j < 64u ? f0 : f1;
                    bo = j < 64u ? o0 : o1;
                };
                const uint32_t j0 = cur - wb + row;
                pick(j0, Ecur.x, Ecur.y); pick(j0 + 1u, Ecur.z, Ecur.w);
                pick(j0 + A2_ROWS, Enext.x, Enext.y); pick(j0 + A2_ROWS + 1u, Enext.z, Enext.w);
            } else {
                Ecur = a2_ent_load(LA, cur, row);
                Enext = a2_ent_load(LA, cur + A2_ROWS, row);
            }
            for (uint32_t i = 4u * (uint32_t)l; i < ncl; i += 256u) *reinterpret_cast<uint4 *>(&bmA[i]) = make_uint4(0, 0, 0, 0);
            bool rv = cur + row < LA.nblk && Ecur.x <= whi;          // first docs ascend: the valid rows are a prefix
            uint32_t nv = (uint32_t)__popcll(__ballot(rv)) >> 2;
            A2Bytes Bcur = a2_fetch(LA.payload, rv, Ecur.y, Ecur.w, rl);
            A2Bytes X;
            uint32_t tw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            II2_STAMP(1)      // clear, A's entries, first fetch
            for (;;) {
                const bool more = nv == A2_ROWS;
                const bool rvn = more && cur + A2_ROWS + row < LA.nblk && Enext.x <= whi;
                const uint32_t nvn = (uint32_t)__popcll(__ballot(rvn)) >> 2;
                const bool last = nvn == 0u;
                X = a2_fetch(last ? LB.payload : LA.payload, last ? rvB : rvn, last ? EB.y : Enext.y, last ? EB.w : Enext.w, rl);
                if (last && p.tomb) {                                // (a window holds <= 512 words: eight per lane, requested together)
                    const uint32_t tw0 = wlo >> 5;
#pragma unroll
                    for (uint32_t k = 0; k < 8u; k++) {
                        const uint32_t i = 64u * k + (uint32_t)l;
                        tw[k] = (i < nw && tw0 + i < p.tomb_nwords) ? p.tomb[tw0 + i] : 0u;
                    }
                }
                if (nv != 0u) a2_mark_rows(lds_all, bmA, LA.payload, rv, Ecur.x, Ecur.y, Ecur.w, Bcur, wlo, wspan, rl, row);
                cur += nv;
                if (last) break;
                Ecur = Enext; Bcur = X; rv = rvn; nv = nvn;
                Enext = a2_ent_load(LA, cur + A2_ROWS, row);
            }
            if (cur > a_cur + 1u) a_cur = cur - 1u;                  // A's last block in the window may reach past it
            II2_STAMP(2)      // mark A
            if (p.tomb) {                                            // removed docs are absent from A: no B posting finds them
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    const uint32_t i = 64u * k + (uint32_t)l;
                    if (64u * k < nw && tw[k] != 0u) bmA[A2_GU / 32u + i] &= ~tw[k];
                }
            }
            // ---- B's postings against the bitmap
            a2_lane_setup(LN, X, rvB, EB, rl, row);
            hm_any = LN.hm;
            if (LN.hm != 0ull && first_window) { hs[2 * l] = 0u; hs[2 * l + 1] = 0u; }      // answers of the hard rows are collected in LDS
            const bool single = first_window && hi - wlo < A2_CAPW;
            if (single) H |= a2_test_lane<false>(lds_all, bmbits, LN, wlo, wspan);
            else H |= a2_test_lane<true>(lds_all, bmbits, LN, wlo, wspan);
            if (LN.hm != 0ull) {
#pragma unroll 1
                for (uint32_t r = 0; r < A2_ROWS; r++) {
                    if (((LN.hm >> (4u * r)) & 0xFull) == 0ull) continue;
                    const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)EB.x, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)EB.y, (int)(4u * r)),
                                   q1r = (uint32_t)__builtin_amdgcn_readlane((int)EB.w, (int)(4u * r));
                    uint32_t *hr = hs + 8u * r;
                    auto probe = [&](uint32_t ix, uint32_t id) {
                        const uint32_t d = id - wlo;
                        if (d <= wspan && ix < 256u) {
                            const uint32_t u = d + A2_GU;
                            if ((bmA[u >> 5] >> (u & 31u)) & 1u) atomicOr(&hr[ix >> 5], 1u << (ix & 31u));
                        }
                    };
                    decode_block_wave4(GlobalBytes{LB.payload}, q0r, q1r, fq,
                                       [&](uint32_t ix, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                           if (mask & 1u) { probe(ix, id0); ix++; }
                                           if (mask & 2u) { probe(ix, id1); ix++; }
                                           if (mask & 4u) { probe(ix, id2); ix++; }
                                           if (mask & 8u) { probe(ix, id3); ix++; }
                                       });
                }
            }
            II2_STAMP(3)      // tombstones, test B
            if (hi - wlo < A2_CAPW) break;
            wlo += A2_CAPW;
            first_window = false;
        }
        if (hm_any != 0ull) {
            const bool rowhard = ((hm_any >> (4u * row)) & 0xFull) != 0ull;
            if (rowhard) H = ((unsigned long long)hs[8u * row + 2u * rl + 1u] << 32) | hs[8u * row + 2u * rl];
        }
    }
    const uint32_t pc = (uint32_t)__popcll(H);
    const uint32_t incl = wave_incl_scan(pc);
    count = wave_bcast(incl, 63);
    const uint32_t q0 = incl - pc;              // my first id's place among the wave's ids
    if (l == 0) wcnt[wv] = count;
    lds_barrier();
    const uint32_t c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
    const uint32_t total = c0 + c1 + c2 + c3;
    const uint32_t before = wv == 0u ? 0u : wv == 1u ? c0 : wv == 2u ? c0 + c1 : c0 + c1 + c2;
    if (threadIdx.x == 0) lb_publish(p.lb, g, total);
    II2_STAMP(4)              // count, barrier, publish
    if (wv == 1u && lb_is_leader(g, gridDim.x)) {            // the group's ids, as soon as its other members have published theirs
        if (!lb_group_publish(p.lb, g, total) && l == 0) { wg_err = 1u; lb_fail(p.lb); }
    }
    // ---- stage the ids of my postings whose answer bit is set: 16-bit offsets from the round's first doc, in output order
    const bool wide = work && hi - lo > 0xFFFFu;             // (a sparse stretch: ids go out one by one, below)
    if (work && !wide && count != 0u) {
        const unsigned long long He = LN.live ? H : 0ull;
        // 64 steps, the same for every lane: the running id advances by the gap byte (SDWA picks it), the answer word is
        // shifted left by adding it to itself — the bit that falls out is the step's write mask — and the lanes whose bit
        // was set store the id's 16-bit offset at their place and move on: 3 vector + 1 LDS + 2 scalar instructions a posting
        // (s_and_saveexec writes SCC: the clobber list says so — the compiler keeps branch conditions there across statements).
        // (The compiler's form of `if (bit) st[q++] = v` is 7 + 1 + 2.)
        uint32_t qa = (uint32_t)(uintptr_t)st + 2u * q0;        // (low 32 bits of an LDS pointer = its offset)
        uint32_t v = LN.base - lo;
        uint32_t h = __builtin_bitreverse32((uint32_t)He);
        unsigned long long sx;
        asm volatile("v_add_co_u32 %1, vcc, %1, %1\n\t"
                     "s_and_saveexec_b64 %3, vcc\n\t"
                     "ds_write_b16 %2, %0\n\t"
                     "v_add_u32 %2, 2, %2\n\t"
                     "s_mov_b64 exec, %3"
                     : "+v"(v), "+v"(h), "+v"(qa), "=&s"(sx) : : "vcc", "scc", "memory");
#define A2_STEP_TXT(SEL)                                                                                               \
        "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL "\n\t"           \
        "v_add_co_u32 %1, vcc, %1, %1\n\t"                                                                            \
        "s_and_saveexec_b64 %3, vcc\n\t"                                                                              \
        "ds_write_b16 %2, %0\n\t"                                                                                     \
        "v_add_u32 %2, 2, %2\n\t"                                                                                     \
        "s_mov_b64 exec, %3\n\t"
#define A2_STAGE_STEP(W, SEL) asm volatile(A2_STEP_TXT(SEL) : "+v"(v), "+v"(h), "+v"(qa), "=&s"(sx) : "v"(W) : "vcc", "scc", "memory");
#define A2_STAGE_3(W) asm volatile(A2_STEP_TXT("BYTE_0") A2_STEP_TXT("BYTE_1") A2_STEP_TXT("BYTE_2") : "+v"(v), "+v"(h), "+v"(qa), "=&s"(sx) : "v"(W) : "vcc", "scc", "memory");
#define A2_STAGE_WORD(W) asm volatile(A2_STEP_TXT("BYTE_0") A2_STEP_TXT("BYTE_1") A2_STEP_TXT("BYTE_2") A2_STEP_TXT("BYTE_3") : "+v"(v), "+v"(h), "+v"(qa), "=&s"(sx) : "v"(W) : "vcc", "scc", "memory");
        A2_STAGE_WORD(LN.ww[0]) A2_STAGE_WORD(LN.ww[1]) A2_STAGE_WORD(LN.ww[2]) A2_STAGE_WORD(LN.ww[3])
        A2_STAGE_WORD(LN.ww[4]) A2_STAGE_WORD(LN.ww[5]) A2_STAGE_WORD(LN.ww[6])
        A2_STAGE_3(LN.ww[7])                                                                                        // postings 1..31
        h = __builtin_bitreverse32((uint32_t)(He >> 32));
        A2_STAGE_STEP(LN.ww[7], "BYTE_3")
        A2_STAGE_WORD(LN.ww[8]) A2_STAGE_WORD(LN.ww[9]) A2_STAGE_WORD(LN.ww[10]) A2_STAGE_WORD(LN.ww[11])
        A2_STAGE_WORD(LN.ww[12]) A2_STAGE_WORD(LN.ww[13]) A2_STAGE_WORD(LN.ww[14])
        A2_STAGE_3(LN.ww[15])                                                                                       // postings 33..63
#undef A2_STAGE_WORD
#undef A2_STAGE_3
#undef A2_STAGE_STEP
#undef A2_STEP_TXT
        if (hm_any != 0ull) {                   // rows with multi-byte gaps: decoded again, every hit ranks itself among its row's answers
#pragma unroll 1
            for (uint32_t r = 0; r < A2_ROWS; r++) {
                if (((hm_any >> (4u * r)) & 0xFull) == 0ull) continue;
                const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)EB.x, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)EB.y, (int)(4u * r)),
                               q1r = (uint32_t)__builtin_amdgcn_readlane((int)EB.w, (int)(4u * r));
                const uint32_t qrow = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)(4u * r));
                const uint32_t *hr = hs + 8u * r;
                auto place = [&](uint32_t ix, uint32_t id) {
                    if (ix >= 256u) return;
                    const uint32_t wd = hr[ix >> 5];
                    if (!((wd >> (ix & 31u)) & 1u)) return;
                    uint32_t rk = (uint32_t)__popc(wd & ((1u << (ix & 31u)) - 1u));
                    for (uint32_t j = 0; j < (ix >> 5); j++) rk += (uint32_t)__popc(hr[j]);
                    st[qrow + rk] = (uint16_t)(id - lo);
                };
                decode_block_wave4(GlobalBytes{LB.payload}, q0r, q1r, fq,
                                   [&](uint32_t ix, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                       if (mask & 1u) { place(ix, id0); ix++; }
                                       if (mask & 2u) { place(ix, id1); ix++; }
                                       if (mask & 4u) { place(ix, id2); ix++; }
                                       if (mask & 8u) { place(ix, id3); ix++; }
                                   });
            }
        }
    }
    II2_STAMP(5)              // stage
    // ---- the ids of all workgroups before mine (wave 0)
    if (wv == 0u) {
        unsigned long long pre = 0ull;
        const bool ok = lb_prefix(p.lb, g, gridDim.x, total, &pre);
        if (l == 0) {
            wg_off = ok ? pre : 0ull;
            if (!ok) { wg_err = 1u; lb_fail(p.lb); }
        }
    }
    II2_STAMP(6)              // look-back
    lds_barrier();
    const bool err = wg_err != 0u;
    const unsigned long long off = wg_off + before;
    if (g == gridDim.x - 1u && threadIdx.x == 0) {          // the last workgroup: the total, or all ones when some workgroup gave up
        const bool anyerr = err || lb_failed(p.lb);
        *p.d_count = anyerr ? ~0ull : wg_off + total;
    }
    if (work && !err && count != 0u) {
        if (!wide) {
            const bool fits = off + count <= p.out_cap;
            for (uint32_t c = 4u * (uint32_t)l; c < count; c += 256u) {        // 16 bytes per lane (wave-private LDS: program order is enough)
                const uint2 pk = *reinterpret_cast<const uint2 *>(&st[c]);
                const uint4 ids = make_uint4(lo + (pk.x & 0xFFFFu), lo + (pk.x >> 16), lo + (pk.y & 0xFFFFu), lo + (pk.y >> 16));
                if (c + 4u <= count && fits) {
                    uint32_t *dst = p.out + off + c;
                    __builtin_memcpy(dst, &ids, 16);                // one 16-byte store, any 4-byte alignment
                } else {
                    const uint32_t vv[4] = {ids.x, ids.y, ids.z, ids.w};
                    for (uint32_t k = 0; k < 4u; k++)
                        if (c + k < count && off + c + k < p.out_cap) p.out[off + c + k] = vv[k];
                }
            }
        } else {
            const unsigned long long He = LN.live ? H : 0ull;
            uint32_t q = q0;
            uint32_t v = LN.base;
            if ((He & 1ull) && off + q < p.out_cap) p.out[off + q] = v;
            q += (uint32_t)(He & 1ull);
#pragma unroll
            for (uint32_t k = 0; k < 63u; k++) {
                v += (LN.ww[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
                const uint32_t hit = (uint32_t)((He >> (k + 1u)) & 1ull);
                if (hit && off + q < p.out_cap) p.out[off + q] = v;
                q += hit;
            }
            if (hm_any != 0ull) {
#pragma unroll 1
                for (uint32_t r = 0; r < A2_ROWS; r++) {
                    if (((hm_any >> (4u * r)) & 0xFull) == 0ull) continue;
                    const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)EB.x, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)EB.y, (int)(4u * r)),
                                   q1r = (uint32_t)__builtin_amdgcn_readlane((int)EB.w, (int)(4u * r));
                    const uint32_t qrow = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)(4u * r));
                    const uint32_t *hr = hs + 8u * r;
                    auto place = [&](uint32_t ix, uint32_t id) {
                        if (ix >= 256u) return;
                        const uint32_t wd = hr[ix >> 5];
                        if (!((wd >> (ix & 31u)) & 1u)) return;
                        uint32_t rk = (uint32_t)__popc(wd & ((1u << (ix & 31u)) - 1u));
                        for (uint32_t j = 0; j < (ix >> 5); j++) rk += (uint32_t)__popc(hr[j]);
                        if (off + qrow + rk < p.out_cap) p.out[off + qrow + rk] = id;
                    };
                    decode_block_wave4(GlobalBytes{LB.payload}, q0r, q1r, fq,
                                       [&](uint32_t ix, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                           if (mask & 1u) { place(ix, id0); ix++; }
                                           if (mask & 2u) { place(ix, id1); ix++; }
                                           if (mask & 4u) { place(ix, id2); ix++; }
                                           if (mask & 8u) { place(ix, id3); ix++; }
                                       });
                }
            }
        }
    }
    II2_STAMP(7)              // flush
    if (stamps && l == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&p.debug[(uint64_t)(blockIdx.x % 2048u) * 8u + i], tacc[i]);
#undef II2_STAMP
}

hipError_t launch_intersect_and2(const DenseParams &p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, s);
    const uint32_t grid = (p.n_waves + 3u) / 4u;
    if (p.lb.agg != nullptr) {
        hipLaunchKernelGGL(k_and2_fused, dim3(grid), dim3(256), 0, s, p);
        if (ev1) (void)hipEventRecord(ev1, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_and2_tiles, dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_and2_expand, dim3(grid), dim3(256), 0, s, p);
    if (ev1) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}

}  // namespace ii2
