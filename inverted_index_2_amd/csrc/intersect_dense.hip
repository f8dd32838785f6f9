// intersect_dense.hip — intersection of lists that are DENSE TOGETHER (the 2-term query over a 100M-doc index of
// BASELINE configs[1] and its relatives), gfx950, wave64, no MFMA.
//
// The general tile kernel (intersect.hip) spends ~40 % of its time at workgroup barriers and needs a partition
// pre-pass and a per-64-tile sum pass around it.  Here every WAVE is on its own:
//   * a wave owns a run of consecutive driver blocks and streams through it four blocks (one 16-lane DPP row per
//     block, 16 gap bytes per lane, loaded straight from HBM in decode layout) at a time — a ROUND;
//   * where the other lists enter the wave's doc range is found once (64-ary search of their skip tables); after
//     that the wave keeps 64 skip entries of every list in registers and advances through them two-pointer style:
//     no descriptors, no partition kernel, no search per tile;
//   * a round marks the postings of every list in wave-private LDS bitmaps ((M << gap) | 1 builds the mask of four
//     postings in registers, one 64-bit shift positions it, at most two ds_or per four postings), ANDs them, clears
//     the tombstoned bits and stores the result words into the wave's slot of a result bitmap (1 bit per doc) in HBM;
//   * waves never wait for each other: no workgroup barrier inside the loop, no inter-workgroup hand-off.
// A second kernel turns the result bitmap into the ascending id array: the offset of a wave's ids is the sum of the
// workgroup counts before it plus the wave counts inside its workgroup — computed in that kernel's prologue (no
// separate sum pass).
//
// Exact for any input the general kernel accepts (multi-byte gaps, short blocks, sparse stretches take slower
// paths inside the same loop); the host only picks this kernel when the driver is dense enough for the result
// bitmap to be small (api.cpp).
#include <algorithm>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t DN_ROWS = 16;                              // blocks per pass: one per row of four lanes, 64 payload bytes per lane
constexpr uint32_t DN_GU = 512;                               // guard bits below and above a window (a lane of narrow groups spans < 512 docs)
constexpr uint32_t DN_CAPW = DENSE_CAPW;                      // docs a window covers (multiple of 32)
constexpr uint32_t DN_NW = ((DN_CAPW + 2 * DN_GU) / 32 + 2 + 3) & ~3u;    // words of one LDS bitmap (a multiple of 4: 16-byte clears)
constexpr uint32_t DN_WAVE_LDS = 2 * DN_NW + 4;               // two bitmaps + the carry word (+ pad)

// first index i in [0, n) with skip[i].first_doc > x, searched 64 ways per round by the whole wave
__device__ __forceinline__ uint32_t wave_skip_upper_bound(const ii2_skip *__restrict__ skip, uint32_t n, uint32_t x) {
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

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// One PASS = up to sixteen blocks of one list against one window: a row of four lanes per block, 64 payload bytes
// per lane.  The wave's work is a sequence of passes; a pass needs its blocks' skip entries (one 16-byte load per
// lane: my row's entry and the next one) and then its payload (four 16-byte loads per lane).  Both are requested one
// pass ahead: while pass k is marked, the payload of pass k + 1 and the entries of pass k + 2 are in flight.
struct Pass {
    const uint8_t *payload;      // wave-uniform
    uint32_t wlo, wspan;         // the window the pass marks into (wave-uniform)
    uint32_t lowbits;            // union: bits of the window's first word that belong to the round before (wave-uniform)
    uint32_t flags;              // PF_* (wave-uniform)
    uint32_t nvalid;             // rows with a block (wave-uniform)
    bool rv;                     // per lane: my row has a block
    uint32_t f, q0, q1;          // per lane: my row's first doc and payload byte range
};
enum : uint32_t { PF_VALID = 1u, PF_FIRST = 2u, PF_LAST = 4u, PF_TOB = 8u, PF_FOLD = 16u };
// first / last pass of a window, marks into B, fold B into A first

// UNION: every list marks into the one bitmap A and the window's result is A itself (OR instead of AND).  The rounds
// are still paced by list 0 (the host passes the list with the most blocks); the first round of wave 0 starts at the
// smallest first doc of all lists and the last round ends at the largest last doc, so postings outside the pacing
// list's own doc range are covered too (long stretches split into windows like any other round).
template <uint32_t NL, bool UNION>
__global__ __launch_bounds__(256) void k_dense_tiles(DenseParams p) {
    __shared__ __align__(16) uint32_t lds[4][DN_WAVE_LDS];
    __shared__ uint32_t wcnt[4];
    const int l = lane_id();
    const uint32_t wv = uni(threadIdx.x >> 6);
    const uint32_t rl = (uint32_t)l & 3u, row = (uint32_t)l >> 2;
    const uint32_t w = blockIdx.x * 4u + wv;                 // this wave's number in doc order
    uint32_t *bmA = lds[wv], *bmB = bmA + DN_NW, *carry = bmB + DN_NW;
    const ListView drv = p.lists[0];
    const uint32_t b0 = w * p.bpw;
    const uint32_t b1 = b0 + p.bpw < drv.nblk ? b0 + p.bpw : drv.nblk;
    uint32_t count = 0, mlo_w = 0, nwords_w = 0;
    // diagnostics only (option debug.stamps): cycles of this wave per part of the loop
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr && !p.debug_expand;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();

    if (b0 < b1) {                                           // wave-uniform
        // my row's skip entry and the next one, for the blocks [at, at + 16) of a list (indices clamped onto the readable entry nblk)
        auto ent_load = [&](uint32_t j, uint32_t at) -> uint4 {
            const ListView L = p.lists[j];
            const uint32_t i0 = at + row < L.nblk ? at + row : L.nblk;
            const uint32_t i1 = i0 < L.nblk ? i0 + 1u : L.nblk;
            const ii2_skip e0 = L.skip[i0], e1 = L.skip[i1];
            return make_uint4(e0.first_doc, e0.byte_off, e1.first_doc, e1.byte_off);
        };
        uint4 E = ent_load(0u, b0);                          // entries the next gen() call consumes
        const uint32_t drv_last = uni(*drv.last_doc);
        const uint32_t end_doc = UNION ? p.u_hi : drv_last;
        const uint32_t lo_w = (UNION && w == 0u) ? p.u_lo : uni(E.x);          // (row 0 = block b0)
        const uint32_t hi_w = b1 < drv.nblk ? uni(drv.skip[b1].first_doc) - 1u : end_doc;
        mlo_w = lo_w & ~31u;
        nwords_w = ((hi_w - mlo_w) >> 5) + 1u;
        uint32_t *slot = p.bitmap + (size_t)((mlo_w - p.base32) >> 5) + w;      // slots of neighbouring waves never overlap (+ w)
        if (l == 0) { carry[0] = 0u; carry[1] = 0xFFFFFFFFu; }

        // where the other lists enter this wave's doc range: the last block that starts at or before lo_w
        uint32_t a[NL > 1 ? NL - 1 : 1];
#pragma unroll
        for (uint32_t j = 1; j < NL; j++) {
            // one probe of 64 entries around a linear guess first (the lists of a dense query are close to uniform: it nearly
            // always brackets the answer and costs one memory round trip instead of three); the full search otherwise
            const ListView L = p.lists[j];
            const uint32_t fj = p.first_doc[j], lj = p.last_doc[j];
            uint32_t ub = 0xFFFFFFFFu;
            if (L.nblk > 64u && lj > fj) {
                const uint64_t rel = lo_w > fj ? (uint64_t)(lo_w - fj) : 0ull;
                uint64_t gss = rel * L.nblk / ((uint64_t)(lj - fj) + 1ull);
                if (gss > L.nblk) gss = L.nblk;
                uint32_t wb = gss > 32ull ? (uint32_t)gss - 32u : 0u;
                if (wb + 64u > L.nblk) wb = L.nblk - 64u;
                const uint32_t fdoc = L.skip[wb + (uint32_t)l].first_doc;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(fdoc <= lo_w));   // first docs ascend: the matches are a prefix
                if ((cnt > 0u || wb == 0u) && (cnt < 64u || wb + 64u == L.nblk)) ub = wb + cnt;
            }
            if (ub == 0xFFFFFFFFu) ub = wave_skip_upper_bound(L.skip, L.nblk, lo_w);
            ub = uni(ub);
            a[j - 1] = ub ? ub - 1u : 0u;
        }

        // ---- the pass generator ----
        uint32_t gb = b0;                     // driver block of the round being generated
        uint32_t g_hi = 0, g_wlo = 0;         // its last doc; the window being generated
        uint32_t g_low = 0;                   // union: the round's first doc - g_wlo while the window is the round's first
        uint32_t g_stage = 0;                 // 0: driver pass next; j >= 1: passes of list j
        uint32_t g_cur = 0;                   // next block of list g_stage
        bool g_new = true;                    // E holds the driver entries of a round whose range is not set up yet
        bool g_first = false;                 // next pass of list g_stage is its first in this window
        bool g_done = false;
        auto gen = [&]() -> Pass {
            Pass P;
            P.payload = drv.payload; P.flags = 0u; P.nvalid = 0u; P.rv = false; P.f = E.x; P.q0 = E.y; P.q1 = E.w;
            P.wlo = 0u; P.wspan = 0u; P.lowbits = 0u;
            if (g_done) return P;
            const uint32_t j = g_stage;
            uint32_t nvalid;
            if (j == 0u) {
                nvalid = b1 - gb < DN_ROWS ? b1 - gb : DN_ROWS;
                if (g_new) {      // the round's doc range: first doc of its first block ... one before the first doc of the block after its last
                    const uint32_t nf = (uint32_t)__builtin_amdgcn_readlane((int)E.z, (int)(4u * (nvalid - 1u)));
                    g_hi = gb + nvalid < drv.nblk ? nf - 1u : end_doc;
                    const uint32_t lo = (UNION && gb == 0u) ? p.u_lo : uni(E.x);
                    g_wlo = lo & ~31u;
                    g_low = lo & 31u;
                    g_new = false;
                }
                P.flags = PF_VALID | PF_FIRST;
            }
            P.wlo = g_wlo;
            if (UNION) P.lowbits = g_low;
            P.wspan = g_hi - g_wlo < DN_CAPW ? g_hi - g_wlo : DN_CAPW - 1u;
            if (j != 0u) {
                const uint32_t whi = g_wlo + P.wspan;
                P.payload = p.lists[j].payload;
                const bool rvj = g_cur + row < p.lists[j].nblk && E.x <= whi;   // first docs ascend: the valid rows are a prefix
                nvalid = (uint32_t)__popcll(__ballot(rvj)) >> 2;
                P.flags = UNION ? PF_VALID : (PF_VALID | PF_TOB | ((g_first && j > 1u) ? PF_FOLD : 0u));
                g_first = false;
            }
            P.nvalid = nvalid;
            P.rv = row < nvalid;
            // advance; the entries of the next pass are requested now and used one pass later
            uint32_t nj = j, ncur = g_cur + nvalid;
            if (j == 0u) {
                if (NL == 1u) nj = NL;             // (single list: the window is complete)
                else { nj = 1u; ncur = a[0]; g_first = true; }
            } else if (nvalid < DN_ROWS) {         // list j is done for this window; its last block in the window may reach past it
#pragma unroll
                for (uint32_t jj = 1; jj < NL; jj++)
                    if (jj == j && ncur > a[jj - 1] + 1u) a[jj - 1] = ncur - 1u;
                nj = j + 1u;
                if (nj < NL) {
#pragma unroll
                    for (uint32_t jj = 1; jj < NL; jj++) if (jj == nj) ncur = a[jj - 1];
                    g_first = true;
                }
            }
            if (nj >= NL) {                        // window complete
                P.flags |= PF_LAST;
                nj = 0u;
                if (g_hi - g_wlo < DN_CAPW) {      // round complete
                    gb += DN_ROWS;
                    g_new = true;
                    if (gb >= b1) g_done = true;
                } else { g_wlo += DN_CAPW; g_low = 0u; }
                ncur = gb;
            }
            g_stage = nj;
            g_cur = ncur;
            if (!g_done) E = ent_load(nj, ncur);
            return P;
        };
        struct Bytes { uint4 g[4]; };
        auto fetch = [&](const Pass &P) -> Bytes {
            const uint32_t len = P.rv ? P.q1 - P.q0 : 0u;
            Bytes B;
            const uint8_t *src = P.payload + P.q0 + 64u * rl;
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                B.g[k] = make_uint4(0, 0, 0, 0);
                if (len > 64u * rl + 16u * k) __builtin_memcpy(&B.g[k], src + 16u * k, 16);   // segments carry 16 bytes of padding
            }
            return B;
        };

        // ---- marking one pass into bm for the window [wlo, wlo + wspan]; exact for any block ----
        auto mark_rows = [&](const Pass &P, const Bytes &B, uint32_t *bm) {
            const uint32_t wlo = P.wlo, wspan = P.wspan;
            const bool rv = P.rv;
            const uint32_t f = P.f;
            const uint32_t len = rv ? P.q1 - P.q0 : 0u;
            const uint32_t myoff = 64u * rl;
            const uint32_t nb = len > myoff ? (len - myoff < 64u ? len - myoff : 64u) : 0u;   // my bytes that belong to the block
            uint32_t ww[16] = {B.g[0].x, B.g[0].y, B.g[0].z, B.g[0].w, B.g[1].x, B.g[1].y, B.g[1].z, B.g[1].w,
                               B.g[2].x, B.g[2].y, B.g[2].z, B.g[2].w, B.g[3].x, B.g[3].y, B.g[3].z, B.g[3].w};
            if (__ballot(rv && len != 255u) == 0ull) {      // full blocks: only the row's last lane holds a byte that is not its own
                if (rl == 3u) ww[15] &= 0x00FFFFFFu;
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) {
                    const uint32_t n = nb > 4u * k ? (nb - 4u * k < 4u ? nb - 4u * k : 4u) : 0u;
                    ww[k] &= n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u);
                }
            }
            auto setbit = [&](uint32_t id, bool valid) {                       // exact range test: any id, any gap
                const uint32_t d = id - wlo;
                if (valid && d <= wspan) atomicOr(&bm[(d + DN_GU) >> 5], 1u << ((d + DN_GU) & 31u));
            };
            // rows whose block has multi-byte gaps (or more than 256 payload bytes): the general one-wave-per-block decoder
            uint32_t any = 0;
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) any |= ww[k];
            const bool hard = rv && (len > 256u || (any & 0x80808080u) != 0u);
            const unsigned long long hm = __ballot(hard);
            if (hm != 0ull) {
#pragma unroll 1
                for (uint32_t r = 0; r < DN_ROWS; r++) {
                    if (((hm >> (4u * r)) & 0xFull) == 0ull) continue;
                    const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)(4u * r)), q0r = (uint32_t)__builtin_amdgcn_readlane((int)P.q0, (int)(4u * r)),
                                   q1r = (uint32_t)__builtin_amdgcn_readlane((int)P.q1, (int)(4u * r));
                    decode_block_wave4(GlobalBytes{P.payload}, q0r, q1r, fq,
                                       [&](uint32_t, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                           setbit(id0, mask & 1u); setbit(id1, mask & 2u); setbit(id2, mask & 4u); setbit(id3, mask & 8u);
                                       });
                }
            }
            const bool rowhard = ((hm >> (4u * row)) & 0xFull) != 0ull;
            const bool live = rv && !rowhard;
            // sum of my 64 gaps, and whether any of my sixteen groups of four postings spans 32 docs or more (the per-group
            // sums are recomputed in the loops below: one v_sad_u8 each, cheaper than sixteen live registers)
            uint32_t acc = 0, wide = 0;
#pragma unroll
            for (uint32_t k = 0; k < 16u; k += 2u) {                             // (two groups per step: three-input or / add)
                const uint32_t g0 = __builtin_amdgcn_sad_u8(ww[k], 0u, 0u), g1 = __builtin_amdgcn_sad_u8(ww[k + 1u], 0u, 0u);
                wide |= g0 | g1;                                                 // a group sum >= 32 sets a bit above bit 4
                acc += g0 + g1;
            }
            // exclusive scan of the lane sums inside the row's four lanes (quad permutes)
            const uint32_t s = live ? acc : 0u;
            const uint32_t s1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x90 /* quad_perm [0,0,1,2] */, 0xf, 0xf, false);
            const uint32_t s2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x40 /* quad_perm [0,0,0,1] */, 0xf, 0xf, false);
            const uint32_t s3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, false);
            const uint32_t excl = rl == 0u ? 0u : rl == 1u ? s1 : rl == 2u ? s1 + s2 : s1 + s2 + s3;
            const uint32_t base = f + excl;                                      // id of the posting right before my bytes
            const uint32_t u = base - wlo + DN_GU;                               // its (guard-shifted) position, mod 2^32
            // A lane whose groups of four postings all span < 32 docs spans < 512 docs in all: if it starts outside
            // [wlo - GU, wlo + wspan] it lies wholly outside the window and is skipped.  A lane with a wider group is kept
            // whatever its start; when it starts outside that range every one of its groups is placed posting by posting.
            const bool inrange = u <= wspan + DN_GU;
            const bool haswide = wide >= 32u;
            const bool act = live && (inrange || haswide);
            const bool exactlane = haswide && !inrange;
            const uint32_t lim = wspan + 2u * DN_GU - 64u;                       // a position in the upper guard: where out-of-window groups are parked
            // the bitmap's place inside the workgroup's LDS array is folded into the position (the array's own address is a
            // link-time constant that ends up in the ds instruction's offset field: no address add per group)
            uint32_t *lds_all = &lds[0][0];
            const uint32_t bmbits = (uint32_t)(bm - lds_all) * 32u;
            if (act) {
                // a group's mask starts AT the posting before it (bit 0 = position q: the block's first doc for the row's
                // first lane, else a posting an earlier group already set — setting it again is harmless)
                // (q carries the bitmap's bit offset inside the workgroup's LDS array - a multiple of 32, so q & 31 is unchanged -:
                // one add per lane instead of one per group)
                uint32_t q = u + bmbits;
                const uint32_t limb = lim + bmbits;
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) {
                    const uint32_t x = ww[k];
                    const uint32_t gs = __builtin_amdgcn_sad_u8(x, 0u, 0u);
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
            // groups of four postings wider than 31 docs (a few per thousand in lists this dense) and every group of a lane that
            // starts outside the window but has such a group: posting by posting, exact range test.  Skipped when no lane has one.
            if (__ballot(act && haswide) != 0ull) {
                uint32_t prev = 0u;
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) {
                    const uint32_t gs = __builtin_amdgcn_sad_u8(ww[k], 0u, 0u);
                    const bool isw = act && haswide && (exactlane || gs >= 32u);
                    if (__ballot(isw) != 0ull) {
                        if (isw) {
                            const uint32_t x = ww[k];
                            uint32_t d = u + prev - DN_GU;                       // doc - wlo of the posting before the group (mod 2^32)
                            if (k == 0u && rl == 0u && d <= wspan) atomicOr(&bm[(d + DN_GU) >> 5], 1u << ((d + DN_GU) & 31u));   // the block's first doc
#pragma unroll
                            for (uint32_t j = 0; j < 4u; j++) {
                                d += (x >> (8u * j)) & 0xFFu;
                                if (4u * k + j < nb && d <= wspan) atomicOr(&bm[(d + DN_GU) >> 5], 1u << ((d + DN_GU) & 31u));
                            }
                        }
                    }
                    prev += gs;
                }
            }
        };

        auto finalise = [&](uint32_t wlo, uint32_t wspan, uint32_t lowbits) {
            // AND, tombstones, count, store to the wave's slot (wave-private LDS: program order is enough); two words per lane
            const uint32_t nw = (wspan >> 5) + 1u;
            const uint32_t sbase = (wlo - mlo_w) >> 5;
            for (uint32_t i = 2u * (uint32_t)l; i < nw; i += 128u) {
                const uint2 ra = *reinterpret_cast<const uint2 *>(&bmA[DN_GU / 32u + i]);
                uint32_t r0 = ra.x, r1 = ra.y;
                if (NL > 1u && !UNION) {
                    const uint2 rb = *reinterpret_cast<const uint2 *>(&bmB[DN_GU / 32u + i]);
                    r0 &= rb.x; r1 &= rb.y;
                }
                const bool has1 = i + 1u < nw;
                if (!has1) r1 = 0u;
                // union: docs of the first word below the round's first doc were output by the round before (an AND never
                // sees them: the driver has no posting there)
                if (UNION && i == 0u) r0 &= ~((1u << lowbits) - 1u);
                if ((wspan & 31u) != 31u) {
                    const uint32_t tm = (2u << (wspan & 31u)) - 1u;
                    if (i == nw - 1u) r0 &= tm;
                    if (i + 1u == nw - 1u) r1 &= tm;
                }
                if (p.tomb) {
                    const uint32_t tw = (wlo >> 5) + i;
                    if (tw < p.tomb_nwords) r0 &= ~p.tomb[tw];
                    if (has1 && tw + 1u < p.tomb_nwords) r1 &= ~p.tomb[tw + 1u];
                }
                count += (uint32_t)__popc(r0) + (uint32_t)__popc(r1);
                // the word that holds this round's first doc may also hold the previous round's last docs
                if (i == 0u && carry[1] == sbase) r0 |= carry[0];
                slot[sbase + i] = r0;
                if (has1) slot[sbase + i + 1u] = r1;
                if (i == nw - 1u) { carry[0] = r0; carry[1] = sbase + i; }
                if (has1 && i + 1u == nw - 1u) { carry[0] = r1; carry[1] = sbase + i + 1u; }
            }
        };
        auto clear = [&](uint32_t *bm, uint32_t ncl) {       // 16 bytes per lane
            for (uint32_t i = 4u * (uint32_t)l; i < ncl; i += 256u) *reinterpret_cast<uint4 *>(&bm[i]) = make_uint4(0, 0, 0, 0);
        };

        // one pass: clear / fold, mark, and the window's result when it was the window's last pass
        auto run_pass = [&](const Pass &P, const Bytes &B) {
            const uint32_t nw = (P.wspan >> 5) + 1u;
            const uint32_t ncl = nw + 2u * (DN_GU / 32u) + 2u;   // <= DN_NW - 2; cleared in 4-word steps
            if (P.flags & PF_FIRST) { clear(bmA, ncl); if (!UNION) clear(bmB, ncl); }
            if (P.flags & PF_FOLD) {
                for (uint32_t i = (uint32_t)l; i < nw; i += 64u) bmA[DN_GU / 32u + i] &= bmB[DN_GU / 32u + i];
                clear(bmB, ncl);
            }
            II2_STAMP(3)      // clear / fold
            if (P.nvalid != 0u) mark_rows(P, B, (P.flags & PF_TOB) ? bmB : bmA);
            II2_STAMP(4)      // mark
            if (P.flags & PF_LAST) finalise(P.wlo, P.wspan, P.lowbits);
            II2_STAMP(5)      // finalise
        };
        // Two passes per trip, their roles swapped: while P is marked, Q (entries + payload) is in flight, and the other way
        // round - the pass that was fetched ahead is marked where it sits (moving it into "the current pass" was 30 register
        // copies per pass, in a kernel that is VALU-issue-bound).
        Pass P = gen();
        Bytes B = fetch(P);
        II2_STAMP(0)          // prologue: searches, first entries, first fetch issued
        while (P.flags & PF_VALID) {
            if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            II2_STAMP(2)      // waiting for the prefetched payload
            Pass Q = gen();
            Bytes Bq = fetch(Q);                              // in flight while P is marked
            II2_STAMP(1)      // generator + fetch issue
            run_pass(P, B);
            if (!(Q.flags & PF_VALID)) break;
            if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            II2_STAMP(2)
            P = gen();
            B = fetch(P);                                     // in flight while Q is marked
            II2_STAMP(1)
            run_pass(Q, Bq);
        }
        count = wave_sum(count);
    }
    if (stamps && l == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&p.debug[(uint64_t)(blockIdx.x % 2048u) * 8u + i], tacc[i]);
#undef II2_STAMP
    if (l == 0) {
        wcnt[wv] = count;
        if (w < p.n_meta) p.meta[w] = make_uint4(mlo_w, nwords_w, count, 0u);
    }
    __syncthreads();
    if (threadIdx.x == 0) p.wg_sum[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// ---- expand: the waves' result bitmaps -> the final ascending id array --------------------------------------
// Workgroup g expands the slots of tile workgroup g (same decomposition).  Its output offset = the counts of the
// workgroups before it, summed here by all 256 threads (a few thousand words) — no separate scan launch.
constexpr uint32_t DX_PRE = 8;              // slot words a lane fetches up front (covers 512 words per wave; longer slots loop)
__global__ __launch_bounds__(256) void k_dense_expand(DenseParams p) {
    __shared__ __align__(16) uint16_t stage[4][2048];       // ids of a round as offsets from the round's first doc
    __shared__ unsigned long long wsum[4];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t w0 = blockIdx.x * 4u;
    const uint32_t w = w0 + (uint32_t)wv;
    unsigned long long tacc[4] = {0, 0, 0, 0};
    unsigned long long tprev = 0;
    const bool stamps = p.debug != nullptr && p.debug_expand;
#define II2_STAMP(i)                                                    \
    if (stamps) {                                                       \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();    \
        tacc[i] += tn_ - tprev;                                         \
        tprev = tn_;                                                    \
    }
    if (stamps) tprev = __builtin_amdgcn_s_memtime();
    // this wave's slot words first: they are independent of the offset arithmetic below
    const uint4 m = w < p.n_meta ? p.meta[w] : make_uint4(0, 0, 0, 0);
    const uint32_t *slot = p.bitmap + (size_t)((m.x - p.base32) >> 5) + w;
    uint32_t pre[DX_PRE];
#pragma unroll
    for (uint32_t k = 0; k < DX_PRE; k++) {
        const uint32_t i = 64u * k + (uint32_t)l;
        pre[k] = (m.z != 0u && i < m.y) ? slot[i] : 0u;
    }
    unsigned long long mine = 0;
    for (uint32_t g = (uint32_t)tid; g < blockIdx.x; g += 256u) mine += p.wg_sum[g];
    for (int d = 32; d >= 1; d >>= 1) mine += (unsigned long long)__shfl_xor((long long)mine, d, 64);
    if (l == 0) wsum[wv] = mine;
    uint32_t before = 0;                        // ids of the waves of this workgroup before mine
    for (uint32_t v = 0; v < (uint32_t)wv; v++) before += p.meta[w0 + v].z;
    __syncthreads();
    unsigned long long off = wsum[0] + wsum[1] + wsum[2] + wsum[3] + before;
    if (blockIdx.x == gridDim.x - 1u && wv == 3 && l == 0) *p.d_count = off + m.z;     // the last wave: the total
    if (m.y == 0u || m.z == 0u) return;
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    II2_STAMP(0)      // prologue: meta, slot words, offsets
    uint16_t *st = stage[wv];
    // 64 words per round: at most 2048 ids staged (16-bit offsets), then written out 16 bytes per lane
    auto emit = [&](uint32_t bits, uint32_t i0) {
        const uint32_t pc = (uint32_t)__popc(bits);
        const uint32_t incl = wave_incl_scan(pc);
        const uint32_t tot = wave_bcast(incl, 63);
        II2_STAMP(1)  // scan
        if (tot == 0u) return;
        uint32_t q = incl - pc;
        const uint32_t lanebase = 32u * (uint32_t)l;
        while (bits) {
            st[q++] = (uint16_t)(lanebase + (uint32_t)__ffs((int)bits) - 1u);
            bits &= bits - 1u;
        }
        II2_STAMP(2)  // stage
        const uint32_t rbase = m.x + 32u * i0;                      // first doc of the round
        for (uint32_t c = 4u * (uint32_t)l; c < tot; c += 256u) {
            const uint2 pk = *reinterpret_cast<const uint2 *>(&st[c]);
            const uint4 ids = make_uint4(rbase + (pk.x & 0xFFFFu), rbase + (pk.x >> 16), rbase + (pk.y & 0xFFFFu), rbase + (pk.y >> 16));
            if (c + 4u <= tot && off + c + 4u <= p.out_cap) {
                uint32_t *dst = p.out + off + c;
                __builtin_memcpy(dst, &ids, 16);                    // one 16-byte store, any 4-byte alignment
            } else {
                const uint32_t v[4] = {ids.x, ids.y, ids.z, ids.w};
                for (uint32_t k = 0; k < 4u; k++)
                    if (c + k < tot && off + c + k < p.out_cap) p.out[off + c + k] = v[k];
            }
        }
        off += tot;
        II2_STAMP(3)  // flush
    };
#pragma unroll
    for (uint32_t k = 0; k < DX_PRE; k++)
        if (64u * k < m.y) emit(pre[k], 64u * k);
    for (uint32_t i0 = 64u * DX_PRE; i0 < m.y; i0 += 64u) {
        const uint32_t i = i0 + (uint32_t)l;
        emit(i < m.y ? slot[i] : 0u, i0);
    }
    if (stamps && l == 0)
        for (int i = 0; i < 4; i++) atomicAdd(&p.debug[(uint64_t)(blockIdx.x % 2048u) * 8u + i], tacc[i]);
#undef II2_STAMP
}

hipError_t launch_intersect_dense(const DenseParams &p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, s);
    const uint32_t grid = (p.n_waves + 3u) / 4u;
    if (p.is_union) {
        if (p.n_lists == 2u) hipLaunchKernelGGL((k_dense_tiles<2u, true>), dim3(grid), dim3(256), 0, s, p);
        else if (p.n_lists == 3u) hipLaunchKernelGGL((k_dense_tiles<3u, true>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_dense_tiles<4u, true>), dim3(grid), dim3(256), 0, s, p);
    } else if (p.n_lists == 2u) hipLaunchKernelGGL((k_dense_tiles<2u, false>), dim3(grid), dim3(256), 0, s, p);
    else if (p.n_lists == 3u) hipLaunchKernelGGL((k_dense_tiles<3u, false>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_dense_tiles<4u, false>), dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_dense_expand, dim3(grid), dim3(256), 0, s, p);
    if (ev1) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}

}  // namespace ii2
