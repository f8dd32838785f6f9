// intersect_bm.hip — the bitmap tile kernel: intersection of up to 4 very dense DV1 lists (gfx950, wave64).
//
// Same tiling, descriptors, result slots and expand pass as intersect.hip (the host picks this kernel when
// the driver list is dense, <= ~3 docs per posting); what differs is the inside of a tile:
//   * one LDS BITMAP per list instead of the byte map: a lane turns four one-byte gaps into a 32-bit mask in
//     registers (M = (M << gap) | 1, one v_lshl_or_b32 per posting), shifts it to its doc position and ORs
//     it into its list's bitmap with two LDS atomics per four postings; the result is the AND of the bitmaps;
//   * ALL phases of a tile (its slice of every list) are staged together: fetched into registers during the
//     previous tile — a whole tile of prefetch distance — and parked in LDS with one commit, so a tile has
//     two barriers (after the commit, after the decode) instead of 2n + 2; the bitmaps are double-buffered
//     by tile parity, so finalising one tile needs no barrier against clearing for the next;
//   * the rows of a decode round take blocks from the flattened block list of all phases: no idle waves
//     when a phase has 6 groups for 4 waves;
//   * LDS is sized per query (dynamic): ~22 KB for two lists, so 6-7 workgroups share a CU.
// A tile this path cannot take (a phase with multi-byte gaps or a short block, too sparse, too wide, too
// many bytes for the staging buffer) is appended to a list that k_isect_tiles (intersect.hip) works off
// afterwards — correctness never depends on the eligibility test.
//
// STATUS (round 1): correct (same tests as the general kernel), but NOT faster — option "intersect.bm2", default 0.
// Measured on C2 (100M docs, 83M postings): 86 us vs 72 us for k_isect_tiles, plus ~9 us for the clean-up launch.
// The ablation (scripts/dbg history in DESIGN.md §5): LDS atomics are charged per ACTIVE LANE (two ds_or per four
// postings cost 25 us more than plain stores would; predicating the empty halves won 11 us), and four global
// atomics per tile on the per-64-tile sums cost 80 us until they were folded into one.  What this kernel needs
// next is an atomic-free way to own bitmap words (merge the masks of neighbouring lanes with DPP before writing).
#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t BM2_RAW = 12288;           // staged payload bytes per tile: 3 x uint4 per thread
constexpr uint32_t BM2_CHUNKS = BM2_RAW / 16u;
constexpr uint32_t BM2_SKIP = 256;            // staged skip entries per tile: 1 per thread
constexpr uint32_t BM2_GU = 128;              // guard bits below and above a tile's doc range
constexpr uint32_t BM2_DW = 2u + 4u * ISECTB_MAXL;

__host__ __device__ constexpr uint32_t bm2_words(uint32_t smax) { return smax / 32u + 1u + 9u; }   // per list bitmap
size_t bm2_lds_bytes(uint32_t n) {
    return (size_t)BM2_RAW + 32u + (BM2_SKIP + 8u) * sizeof(ii2_skip) + ((3u * BM2_DW + 3u) & ~3u) * 4u + 16u + 2u * (size_t)n * bm2_words(ISECT_SMAX) * 4u + 64u;
}

struct Bm2Tile {                      // wave-uniform description of one tile's staging (named members only: arrays
    bool ok;                          // indexed by a per-lane phase number would be put in scratch memory)
    uint32_t nchunks, nent, nblk_total;
    uint4 cstart;                     // first 16-byte chunk of each phase in the staging buffer
    uint4 estart;                     // first skip entry of each phase
    uint4 bstart;                     // first block of each phase in the flattened block list
    uint4 base16, bl;
};

// The asm pins the four (wave-uniform) candidates in scalar registers: left alone, LLVM folds a select of struct
// members into a load with a selected ADDRESS, which keeps the whole struct in scratch memory.
__device__ __forceinline__ uint32_t sel4(const uint4 &a, uint32_t j) {
    uint32_t x = a.x, y = a.y, z = a.z, w = a.w;
    asm volatile("" : "+s"(x), "+s"(y), "+s"(z), "+s"(w));
    uint32_t v = x;
    v = j == 1u ? y : v;
    v = j == 2u ? z : v;
    v = j == 3u ? w : v;
    return v;
}

__device__ __forceinline__ Bm2Tile bm2_describe(const uint32_t *D, uint32_t n) {
    Bm2Tile t;
    const uint32_t span = D[1] - D[0];
    bool ok = span < ISECT_SMAX;
    uint32_t c = 0, e = 0, b = 0;
    uint32_t cs[4], es[4], bs[4], b16[4], bls[4];
#pragma unroll
    for (uint32_t j = 0; j < ISECTB_MAXL; j++) {
        cs[j] = c; es[j] = e; bs[j] = b;
        b16[j] = 0; bls[j] = 0;
        if (j < n) {
            const uint32_t bl = D[2 + 4 * j], bh = D[3 + 4 * j], qlo = D[4 + 4 * j], qhi = D[5 + 4 * j];
            const uint32_t nb = bh - bl, bytes = qhi - qlo;
            ok = ok && nb > 0u && bytes == 255u * nb && (uint64_t)bytes * 13u >= (uint64_t)span * 4u;
            b16[j] = qlo & ~15u;
            bls[j] = bl;
            c += (qhi - (qlo & ~15u) + 15u) >> 4;
            e += nb + 1u;
            b += nb;
        }
    }
    t.cstart = make_uint4(cs[0], cs[1], cs[2], cs[3]);
    t.estart = make_uint4(es[0], es[1], es[2], es[3]);
    t.bstart = make_uint4(bs[0], bs[1], bs[2], bs[3]);
    t.base16 = make_uint4(b16[0], b16[1], b16[2], b16[3]);
    t.bl = make_uint4(bls[0], bls[1], bls[2], bls[3]);
    t.nchunks = c; t.nent = e; t.nblk_total = b;
    t.ok = ok && c <= BM2_CHUNKS && e <= BM2_SKIP;
    return t;
}

struct Bm2Prefetch { uint4 r[3]; ii2_skip sk; };
struct Bm2Lists { const uint8_t *pay0, *pay1, *pay2, *pay3; const ii2_skip *skp0, *skp1, *skp2, *skp3; };
// pointers that went through the asm above have lost their address space: name it, or the loads become flat loads
// (which count against lgkmcnt too — every LDS wait would then wait for the prefetch)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const uint8_t __attribute__((address_space(1))) *gbytes_t;
typedef const ii2_skip __attribute__((address_space(1))) *gskip_t;
__device__ __forceinline__ gbytes_t sel_pay(const Bm2Lists &L, uint32_t j) {
    const uint8_t *a = L.pay0, *b = L.pay1, *c = L.pay2, *d = L.pay3;
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
    const uint8_t *v = a;
    v = j == 1u ? b : v;
    v = j == 2u ? c : v;
    v = j == 3u ? d : v;
    return (gbytes_t)v;
}
__device__ __forceinline__ gskip_t sel_skp(const Bm2Lists &L, uint32_t j) {
    const ii2_skip *a = L.skp0, *b = L.skp1, *c = L.skp2, *d = L.skp3;
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
    const ii2_skip *v = a;
    v = j == 1u ? b : v;
    v = j == 2u ? c : v;
    v = j == 3u ? d : v;
    return (gskip_t)v;
}

// which phase a flattened index belongs to (starts ascending, n <= 4)
__device__ __forceinline__ uint32_t bm2_phase_of(const uint4 &starts, uint32_t n, uint32_t x) {
    return ((n > 1u && x >= starts.y) ? 1u : 0u) + ((n > 2u && x >= starts.z) ? 1u : 0u) + ((n > 3u && x >= starts.w) ? 1u : 0u);
}

__device__ __forceinline__ void bm2_issue(Bm2Prefetch &pf, const Bm2Tile &t, const Bm2Lists &L, uint32_t n, int tid) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t c = (uint32_t)tid + 256u * (uint32_t)k;
        pf.r[k] = make_uint4(0, 0, 0, 0);
        if (c < t.nchunks) {
            const uint32_t j = bm2_phase_of(t.cstart, n, c);
            const u32x4 v = *(const u32x4 __attribute__((address_space(1))) *)(sel_pay(L, j) + sel4(t.base16, j) + 16u * (c - sel4(t.cstart, j)));
            pf.r[k] = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
    pf.sk.first_doc = 0; pf.sk.byte_off = 0;
    if ((uint32_t)tid < t.nent) {
        const uint32_t j = bm2_phase_of(t.estart, n, (uint32_t)tid);
        const u32x2 v = *(const u32x2 __attribute__((address_space(1))) *)(sel_skp(L, j) + sel4(t.bl, j) + ((uint32_t)tid - sel4(t.estart, j)));
        pf.sk.first_doc = v.x;
        pf.sk.byte_off = v.y;
    }
}

__global__ __launch_bounds__(256) void k_isect_bm(IntersectParams p) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint8_t *raw = lds;                                                        // [BM2_RAW + 32]
    ii2_skip *skipbuf = reinterpret_cast<ii2_skip *>(lds + BM2_RAW + 32u);    // [BM2_SKIP + 8]
    uint32_t *desc = reinterpret_cast<uint32_t *>(skipbuf + BM2_SKIP + 8u);   // [3][BM2_DW]
    uint32_t *tcnt = desc + ((3u * BM2_DW + 3u) & ~3u);                                       // [4] survivors of the tile being finalised, by tile parity (entries 0, 1)
    uint32_t *bmall = tcnt + 4u;                                               // [2][n][BMW]
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t n = p.n_lists;
    const uint32_t stride = 2u + 4u * n;
    constexpr uint32_t BMW = bm2_words(ISECT_SMAX);
    const uint32_t rl = (uint32_t)l & 15u, rowid = (uint32_t)wv * 4u + ((uint32_t)l >> 4);

    uint32_t tile = blockIdx.x;
    if (tile >= p.n_tiles) return;
    // prime: descriptors of the first two tiles, both bitmap buffers cleared, first tile's bytes requested
    if ((uint32_t)tid < stride) desc[tid] = p.ranges[(uint64_t)tile * stride + tid];
    for (uint32_t i = (uint32_t)tid; i < 2u * n * BMW; i += 256u) bmall[i] = 0u;
    if (tid < 4) tcnt[tid] = 0u;
    __syncthreads();
    const Bm2Lists L = {p.lists[0].payload, p.lists[n > 1u ? 1 : 0].payload, p.lists[n > 2u ? 2 : 0].payload, p.lists[n > 3u ? 3 : 0].payload,
                        p.lists[0].skip, p.lists[n > 1u ? 1 : 0].skip, p.lists[n > 2u ? 2 : 0].skip, p.lists[n > 3u ? 3 : 0].skip};
    Bm2Prefetch pf;
    {
        const Bm2Tile first = bm2_describe(desc, n);
        if (first.ok) bm2_issue(pf, first, L, n, tid);
    }

    // descriptors of the next tile: requested now, handed over at the top of the loop (where the fetched
    // bytes are consumed anyway — no other point of the loop waits for memory)
    uint32_t dreg = 0;
    if (tile + gridDim.x < p.n_tiles && (uint32_t)tid < stride) dreg = p.ranges[(uint64_t)(tile + gridDim.x) * stride + tid];

    bool prev_ok = false;
    uint32_t prev_tile = 0, it = 0;
    for (; tile < p.n_tiles; it++, tile += gridDim.x) {
        const uint32_t r0 = it % 3u, r1 = (it + 1u) % 3u;
        const uint32_t *D = desc + r0 * BM2_DW;
        const Bm2Tile cur = bm2_describe(D, n);             // recomputed from LDS: a loop-carried struct would live in scratch
        const uint32_t next_tile = tile + gridDim.x, next2_tile = next_tile + gridDim.x;
        const bool has_next = next_tile < p.n_tiles && next_tile > tile;
        const bool has_next2 = has_next && next2_tile < p.n_tiles && next2_tile > next_tile;
        if (has_next && (uint32_t)tid < stride) desc[r1 * BM2_DW + tid] = dreg;     // ring of 3: tile it-1's entry may still be read
        uint32_t *bmcur = bmall + (it & 1u) * n * BMW;
        uint32_t *bmnext = bmall + ((it + 1u) & 1u) * n * BMW;

        if (cur.ok) {
            // park the prefetched bytes and skip entries
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t c = (uint32_t)tid + 256u * (uint32_t)k;
                if (c < cur.nchunks) *reinterpret_cast<uint4 *>(raw + 16u * c) = pf.r[k];
            }
            if ((uint32_t)tid < cur.nent) skipbuf[tid] = pf.sk;
        } else if (tid == 0) {
            p.defer[atomicAdd(p.n_defer, 1u)] = tile;          // left to k_isect_tiles
        }
        lds_barrier();                                          // B1: staging complete; finalise(it-1) is over everywhere
        // publish the previous tile's count: the waves added theirs in LDS, so one global atomic per tile
        // (an atomic per wave on the per-64-tile sums cost more than the whole decode)
        if (tid == 0 && prev_ok) {
            const uint32_t c = tcnt[(it + 1u) & 1u];
            tcnt[(it + 1u) & 1u] = 0u;
            p.tile_count[prev_tile] = c;
        }
        if (has_next2 && (uint32_t)tid < stride) dreg = p.ranges[(uint64_t)next2_tile * stride + tid];
        // request the next tile's bytes — they have this whole tile to arrive
        if (has_next) {
            const Bm2Tile nxt = bm2_describe(desc + r1 * BM2_DW, n);
            if (nxt.ok) bm2_issue(pf, nxt, L, n, tid);
        }
        // the other bitmap buffer was last read by finalise(it-1): clear it for tile it+1
        for (uint32_t i = (uint32_t)tid; 4u * i < n * BMW; i += 256u) reinterpret_cast<uint4 *>(bmnext)[i] = make_uint4(0, 0, 0, 0);

        const uint32_t lo = D[0], hi = D[1];
        const uint32_t mlo = lo & ~31u, mspan = hi - mlo;
        const uint32_t nwords = (mspan >> 5) + 1u;
        const uint32_t lim = mspan + 252u;                      // highest guard-shifted position ever written
        if (cur.ok) {
            // ---- decode: the rows of a round take consecutive blocks of the flattened block list ----
            for (uint32_t u0 = 0; u0 < cur.nblk_total; u0 += 16u) {
                const uint32_t u = u0 + rowid;
                const bool rv = u < cur.nblk_total;
                uint32_t j = 0, q0 = 0, q1 = 0, first = 0;
                if (rv) {
                    j = bm2_phase_of(cur.bstart, n, u);
                    const uint32_t ei = sel4(cur.estart, j) + (u - sel4(cur.bstart, j));
                    const ii2_skip e0 = skipbuf[ei], e1 = skipbuf[ei + 1u];
                    const uint32_t roff = 16u * sel4(cur.cstart, j) - sel4(cur.base16, j);     // staging offset of payload byte 0 of list j (mod 2^32)
                    q0 = e0.byte_off + roff;
                    q1 = e1.byte_off + roff;
                    first = e0.first_doc;
                }
                uint32_t *bmj = bmcur + j * BMW;
                auto setbit = [&](uint32_t *bm, uint32_t id, bool valid) {
                    const uint32_t pos = id - mlo + BM2_GU;
                    if (valid && pos <= lim) atomicOr(&bm[pos >> 5], 1u << (pos & 31u));
                };
                uint32_t base;
                uint4 w;
                if (decode_rows16(LdsBytes16{raw}, q0, q1, first, rv, base, w)) {
                    const uint32_t uu = base - mlo + BM2_GU;             // guard-shifted position of the posting before my bytes
                    const bool ok = rv && uu <= mspan + BM2_GU;          // else the lane lies wholly outside the tile's range
                    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
                    uint32_t ss[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) ss[g] = __builtin_amdgcn_sad_u8(ww[g], 0u, 0u);
                    const uint32_t smax = max(max(ss[0], ss[1]), max(ss[2], ss[3]));
                    const bool fast = ok && smax <= 31u;                 // four postings fit one 32-bit mask
                    const uint32_t seed = fast ? 1u : 0u;
                    uint32_t q = uu;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const uint32_t x = ww[g];
                        uint32_t M = (seed << ((x >> 24) & 31u)) | seed;
                        M = (M << ((x >> 16) & 31u)) | seed;
                        M = (M << ((x >> 8) & 31u)) | seed;
                        uint32_t P = q + (x & 0xFFu);
                        P = P < lim ? P : lim;
                        q += ss[g];
                        const unsigned long long MM = (unsigned long long)M << (P & 31u);
                        uint32_t *dst = bmj + (P >> 5);
                        if ((uint32_t)MM) atomicOr(dst, (uint32_t)MM);                    // LDS atomics cost per active lane:
                        if ((uint32_t)(MM >> 32)) atomicOr(dst + 1, (uint32_t)(MM >> 32)); // skip the empty halves
                    }
                    // lanes with a wide group walk their postings one by one if they can reach the tile's range — also a lane
                    // that STARTS more than the guard below the range (not ok) and jumps into it; a lane of narrow groups
                    // spans < 128 docs, so one that is not ok lies wholly outside
                    if (__ballot(rv && smax > 31u) != 0ull) {          // rare: the test for reaching the range is only made here
                        const bool slow = rv && smax > 31u && (ok || (uu + (ss[0] + ss[1] + ss[2] + ss[3])) < uu);   // second case: the sum wrapped past 2^32
                        uint32_t pp = uu;
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            pp += (ww[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                            if (slow && pp <= lim) atomicOr(&bmj[pp >> 5], 1u << (pp & 31u));
                        }
                    }
                    setbit(bmj, first, rv && rl == 0u);
                } else {
                    // some block of this round has multi-byte gaps or is short: one block at a time, wave-wide
                    for (int r = 0; r < 4; r++) {
                        const int src = 16 * r;
                        if (!__shfl((int)rv, src, 64)) continue;
                        const uint32_t jq0 = (uint32_t)__shfl((int)q0, src, 64), jq1 = (uint32_t)__shfl((int)q1, src, 64);
                        const uint32_t jf = (uint32_t)__shfl((int)first, src, 64), jj = (uint32_t)__shfl((int)j, src, 64);
                        uint32_t *bm = bmcur + jj * BMW;
                        decode_block_wave4(LdsBytes{raw}, jq0, jq1, jf,
                                           [&](uint32_t, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                               setbit(bm, id0, mask & 1u); setbit(bm, id1, mask & 2u);
                                               setbit(bm, id2, mask & 4u); setbit(bm, id3, mask & 8u);
                                           });
                    }
                }
            }
        }
        lds_barrier();                                          // B2: every bit of this tile is set
        if (cur.ok) {
            // ---- finalise: AND of the list bitmaps, tombstones cleared, survivors counted ----
            uint32_t *slot = p.tmp + (uint64_t)tile * p.slot_words;
            uint32_t mine = 0;
            for (uint32_t wi = (uint32_t)tid; wi < nwords; wi += 256u) {
                uint32_t word = bmcur[BM2_GU / 32u + wi];
                for (uint32_t j = 1; j < n; j++) word &= bmcur[j * BMW + BM2_GU / 32u + wi];
                if (wi == nwords - 1u && (mspan & 31u) != 31u) word &= (2u << (mspan & 31u)) - 1u;
                if (p.tomb) {
                    const uint32_t tw = (mlo >> 5) + wi;
                    if (tw < p.tomb_nwords) word &= ~p.tomb[tw];
                }
                slot[wi] = word;
                mine += (uint32_t)__popc(word);
            }
            mine = wave_sum(mine);
            if (l == 0 && mine) atomicAdd(&tcnt[it & 1u], mine);
        }
        prev_ok = cur.ok;
        prev_tile = tile;
    }
    lds_barrier();
    if (tid == 0 && prev_ok) {
        const uint32_t c = tcnt[(it + 1u) & 1u];
        p.tile_count[prev_tile] = c;
    }
}

hipError_t launch_intersect_bm(const IntersectParams &p, uint32_t grid, hipStream_t s) {
    const size_t lds = bm2_lds_bytes(p.n_lists);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_isect_bm), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_isect_bm, dim3(grid < p.n_tiles ? grid : p.n_tiles), dim3(256), lds, s, p);
    return hipGetLastError();
}

}  // namespace ii2
