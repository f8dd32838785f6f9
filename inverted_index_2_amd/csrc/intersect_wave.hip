// intersect_wave.hip — wave-level tile kernel of the multi-term intersection (2..4 lists).
//
// Same algorithm as k_isect_tiles (intersect.hip: byte map per doc range, bitmap result,
// gallop for sparse ranges) but the unit of work is one WAVE, not a workgroup:
//   * a mini-tile = up to 4 driver blocks = one row-decode round (one 16-lane row per block);
//     each wave owns an 8 KiB LDS byte map and walks mini-tiles w, w+W, w+2W, ...;
//   * no barriers at all — a wave only ever waits for its own loads;
//   * the pre-pass writes, per mini-tile, the (payload range, first_doc) of every block the
//     wave will decode, so the wave's dependent-load chain is two links (descriptor ->
//     payload) and both are issued a whole mini-tile ahead;
//   * payload bytes go from HBM straight into the registers of the lane that decodes them
//     (16 bytes per lane, the row-decode layout) — no LDS staging, no commit.
#include <algorithm>

#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

constexpr uint32_t WABLK = ISECTW_ABLK;           // prefetched blocks per other list and mini-tile (3 rounds)
constexpr uint32_t WMAP = ISECTW_SMAX + 32u;      // byte map bytes per wave
constexpr uint32_t WDESC_MAX = 16u + 40u * (ISECTW_MAXL - 1u);
constexpr uint32_t LIST_FLAG = 0x80000000u;
constexpr uint32_t NONE = 0xFFFFFFFFu;

// descriptor layout (words): [0] lo [1] hi [2] driver rows [3] -
//   [4 + 3r ..]  driver row r: q0, q1, first_doc
//   list j >= 1 at base = 16 + 40 (j-1): [base] blocks in range, [base+1] first block, [base+4+3i ..] block i: q0, q1, first_doc
__host__ __device__ constexpr uint32_t wdesc_words(uint32_t n) { return 16u + 40u * (n - 1u); }

// two upper bounds on one skip table at once (see intersect.hip)
__device__ __forceinline__ void wave_ub2(const ii2_skip *__restrict__ skip, uint32_t n, uint32_t xa, uint32_t xb, uint32_t &ra, uint32_t &rb) {
    const uint32_t l = (uint32_t)lane_id();
    uint32_t loa = 0, hia = n, lob = 0, hib = n;
    bool donea = false, doneb = false;
    while (!(donea && doneb)) {
        const uint32_t spa = hia - loa, spb = hib - lob;
        const uint32_t sta = (spa + 63u) >> 6, stb = (spb + 63u) >> 6;
        const uint32_t pa = loa + l * sta, pb = lob + l * stb;
        const bool ina = !donea && pa < hia, inb = !doneb && pb < hib;
        const uint32_t fa = ina ? skip[pa].first_doc : 0u;
        const uint32_t fb = inb ? skip[pb].first_doc : 0u;
        if (!donea) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(ina && fa <= xa));
            const uint32_t nin = (uint32_t)__popcll(__ballot(ina));
            if (spa == 0u) donea = true;
            else if (sta == 1u) { loa = cnt < nin ? loa + cnt : hia; hia = loa; donea = true; }
            else { const uint32_t nlo = cnt ? loa + (cnt - 1u) * sta + 1u : loa; hia = cnt < nin ? loa + cnt * sta : hia; loa = nlo; }
        }
        if (!doneb) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(inb && fb <= xb));
            const uint32_t nin = (uint32_t)__popcll(__ballot(inb));
            if (spb == 0u) doneb = true;
            else if (stb == 1u) { lob = cnt < nin ? lob + cnt : hib; hib = lob; doneb = true; }
            else { const uint32_t nlo = cnt ? lob + (cnt - 1u) * stb + 1u : lob; hib = cnt < nin ? lob + cnt * stb : hib; lob = nlo; }
        }
    }
    ra = loa;
    rb = lob;
}

// ---- pre-pass: one wave per (mini-tile, list) ------------------------------------------------
__global__ __launch_bounds__(256) void k_isectw_partition(IntersectParams p) {
    const uint64_t gw = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n = p.n_lists;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < p.n_sums) p.sums[gid] = 0;
    if (gw >= (uint64_t)p.n_tiles * n) return;
    const uint32_t m = (uint32_t)(gw / n), j = (uint32_t)(gw % n);
    const uint32_t l = (uint32_t)lane_id();
    const ListView d = p.lists[0];
    const uint32_t b0 = m * p.G;
    const uint32_t b1 = b0 + p.G < d.nblk ? b0 + p.G : d.nblk;
    const uint32_t lo = d.skip[b0].first_doc;
    const uint32_t hi = b1 < d.nblk ? d.skip[b1].first_doc - 1u : *d.last_doc;
    uint32_t *desc = p.ranges + (uint64_t)m * wdesc_words(n);
    if (j == 0) {
        if (l == 0) { desc[0] = lo; desc[1] = hi; desc[2] = b1 - b0; desc[3] = 0; }
        if (l < 4u) {
            uint32_t q0 = 0, q1 = 0, f = 0;
            if (b0 + l < b1) { const ii2_skip e0 = d.skip[b0 + l], e1 = d.skip[b0 + l + 1u]; q0 = e0.byte_off; q1 = e1.byte_off; f = e0.first_doc; }
            desc[4 + 3 * l] = q0; desc[5 + 3 * l] = q1; desc[6 + 3 * l] = f;
        }
        return;
    }
    const ListView L = p.lists[j];
    uint32_t ub, bh;
    wave_ub2(L.skip, L.nblk, lo, hi, ub, bh);
    const uint32_t bl = ub ? ub - 1u : 0u;
    if (bh < bl) bh = bl;
    const uint32_t nblk = bh - bl;
    const uint32_t base = 16u + 40u * (j - 1u);
    if (l == 0) { desc[base] = nblk; desc[base + 1] = bl; desc[base + 2] = 0; desc[base + 3] = 0; }
    if (l < WABLK) {
        uint32_t q0 = 0, q1 = 0, f = 0;
        if (l < nblk) { const ii2_skip e0 = L.skip[bl + l], e1 = L.skip[bl + l + 1u]; q0 = e0.byte_off; q1 = e1.byte_off; f = e0.first_doc; }
        desc[base + 4 + 3 * l] = q0; desc[base + 5 + 3 * l] = q1; desc[base + 6 + 3 * l] = f;
    }
}

// ---- the wave kernel ---------------------------------------------------------------------------
struct __align__(16) WaveSmem {
    uint8_t map[WMAP];                  // byte map | gallop: cand[1024] u32, hit[1024] u8, decoded block[256] u32
    uint32_t desc[2][WDESC_MAX];
};

__device__ __forceinline__ uint32_t bytes_eq_mask_w(uint32_t w, uint32_t n4) {
    const uint32_t x = w ^ n4;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    const uint32_t z = ~(t | x | 0x7F7F7F7Fu);
    return (((z >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// the 16 payload bytes this lane decodes in a row round: block q0..q1 of its row
__device__ __forceinline__ uint4 fetch16(const uint8_t *__restrict__ payload, uint32_t q0, uint32_t q1, bool rv) {
    const uint32_t rl = (uint32_t)lane_id() & 15u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rv && q1 - q0 > 16u * rl) __builtin_memcpy(&v, payload + q0 + 16u * rl, 16);
    return v;
}

template <int NL>
struct WavePayload {
    uint4 d;                     // driver round
    uint4 a[NL - 1][3];          // three rounds per other list
};

template <int NL>
__device__ __forceinline__ void fetch_payload(WavePayload<NL> &P, const uint32_t *D, const IntersectParams &p) {
    const uint32_t row = (uint32_t)lane_id() >> 4;
    P.d = fetch16(p.lists[0].payload, D[4 + 3 * row], D[5 + 3 * row], row < D[2]);
#pragma unroll
    for (int j = 1; j < NL; j++) {
        const uint32_t base = 16u + 40u * (uint32_t)(j - 1);
        const uint32_t nb = D[base] < WABLK ? D[base] : WABLK;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint32_t bi = 4u * (uint32_t)i + row;
            P.a[j - 1][i] = fetch16(p.lists[j].payload, D[base + 4 + 3 * bi], D[base + 5 + 3 * bi], bi < nb);
        }
    }
}

template <int NL>
__global__ __launch_bounds__(256) void k_isectw_tiles(IntersectParams p) {
    __shared__ WaveSmem smw[4];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const uint32_t row = (uint32_t)l >> 4, rl = (uint32_t)l & 15u;
    WaveSmem &sm = smw[wv];
    constexpr uint32_t n = NL;
    constexpr uint32_t DW = wdesc_words(NL);
    const uint32_t nwaves = gridDim.x * 4u;
    uint32_t m = blockIdx.x * 4u + (uint32_t)wv;
    if (m >= p.n_tiles) return;                      // no barrier anywhere below: waves are independent

    // prologue: this wave's first descriptor, its payload, and the next descriptor (in flight)
    for (uint32_t w = (uint32_t)l; w < DW; w += 64u) sm.desc[0][w] = p.ranges[(uint64_t)m * DW + w];
    WavePayload<NL> P;
    fetch_payload<NL>(P, sm.desc[0], p);
    uint32_t m1 = m + nwaves;
    uint32_t dreg[(DW + 63u) / 64u];
#pragma unroll
    for (uint32_t c = 0; c < (DW + 63u) / 64u; c++) {
        const uint32_t w = (uint32_t)l + 64u * c;
        dreg[c] = (m1 < p.n_tiles && w < DW) ? p.ranges[(uint64_t)m1 * DW + w] : 0u;
    }

    for (uint32_t it = 0;; it++) {
        const uint32_t *D = sm.desc[it & 1u];
        uint32_t *DNX = sm.desc[(it + 1u) & 1u];
        const bool has1 = m1 < p.n_tiles;
        WavePayload<NL> PN;
        const uint32_t m2 = m1 + nwaves;
        if (has1) {
            // park the next descriptor in LDS, start its payload loads, and fetch the one after
#pragma unroll
            for (uint32_t c = 0; c < (DW + 63u) / 64u; c++) {
                const uint32_t w = (uint32_t)l + 64u * c;
                if (w < DW) DNX[w] = dreg[c];
            }
            fetch_payload<NL>(PN, DNX, p);
#pragma unroll
            for (uint32_t c = 0; c < (DW + 63u) / 64u; c++) {
                const uint32_t w = (uint32_t)l + 64u * c;
                dreg[c] = (m2 < p.n_tiles && w < DW) ? p.ranges[(uint64_t)m2 * DW + w] : 0u;
            }
        }

        // ================= this mini-tile =================
        const uint32_t lo = D[0], hi = D[1];
        const uint32_t mlo = lo & ~31u;
        const uint32_t mspan = hi - mlo;
        uint32_t *slot = p.tmp + (uint64_t)m * p.slot_words;
        const uint32_t nrows = D[2];
        if (mspan < ISECTW_SMAX) {
            // ---------- byte-map path ----------
            const uint32_t nwords = (mspan >> 5) + 1u;
            for (uint32_t i = (uint32_t)l * 16u; i < nwords * 32u; i += 64u * 16u)
                *reinterpret_cast<uint4 *>(&sm.map[i]) = make_uint4(0, 0, 0, 0);
            const uint32_t dummy = mspan + 1u;
            uint8_t want = 0;
            auto mark4 = [&](uint32_t, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                const uint32_t o0 = id0 - mlo, o1 = id1 - mlo, o2 = id2 - mlo, o3 = id3 - mlo;
                const bool v0 = (mask & 1u) && o0 <= mspan, v1 = (mask & 2u) && o1 <= mspan;
                const bool v2 = (mask & 4u) && o2 <= mspan, v3 = (mask & 8u) && o3 <= mspan;
                if (want == 0) {
                    if (v0) sm.map[o0] = 1;
                    if (v1) sm.map[o1] = 1;
                    if (v2) sm.map[o2] = 1;
                    if (v3) sm.map[o3] = 1;
                } else {
                    const uint8_t m0 = v0 ? sm.map[o0] : (uint8_t)0xFE, m1_ = v1 ? sm.map[o1] : (uint8_t)0xFE;
                    const uint8_t m2_ = v2 ? sm.map[o2] : (uint8_t)0xFE, m3 = v3 ? sm.map[o3] : (uint8_t)0xFE;
                    const uint8_t nx = (uint8_t)(want + 1u);
                    if (m0 == want) sm.map[o0] = nx;
                    if (m1_ == want) sm.map[o1] = nx;
                    if (m2_ == want) sm.map[o2] = nx;
                    if (m3 == want) sm.map[o3] = nx;
                }
            };
            auto mark16 = [&](uint32_t base, const uint4 &w, bool rv, uint32_t first_doc) {
                uint32_t acc = rv ? base - mlo : 0xFFFFFFFFu;
                uint32_t o[16];
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    acc += (ww[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    o[k] = acc < dummy ? acc : dummy;
                }
                const uint32_t f = first_doc - mlo;
                const uint32_t of = (rv && rl == 0u && f < dummy) ? f : dummy;
                if (want == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k++) sm.map[o[k]] = 1;
                    sm.map[of] = 1;
                } else {
                    uint8_t mm[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) mm[k] = sm.map[o[k]];
                    const uint8_t mf = sm.map[of];
                    const uint8_t nx = (uint8_t)(want + 1u);
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (mm[k] == want) sm.map[o[k]] = nx;
                    if (mf == want) sm.map[of] = nx;
                }
            };
            // driver round
            {
                const uint32_t q0 = D[4 + 3 * row], q1 = D[5 + 3 * row], f = D[6 + 3 * row];
                const bool rv = row < nrows;
                uint32_t base;
                uint4 w = P.d;
                if (rows16_finish(q0, q1, f, rv, base, w)) mark16(base, w, rv, f);
                else
                    for (uint32_t r = 0; r < nrows; r++)
                        decode_block_wave4(GlobalBytes{p.lists[0].payload}, D[4 + 3 * r], D[5 + 3 * r], D[6 + 3 * r], mark4);
            }
            // the other lists
#pragma unroll
            for (int j = 1; j < NL; j++) {
                want = (uint8_t)j;
                const uint32_t base_d = 16u + 40u * (uint32_t)(j - 1);
                const uint32_t nblk = D[base_d], bl = D[base_d + 1];
                const uint32_t npre = nblk < WABLK ? nblk : WABLK;
                const ListView L = p.lists[j];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    if (4u * (uint32_t)i < npre) {
                        const uint32_t bi = 4u * (uint32_t)i + row;
                        const bool rv = bi < npre;
                        const uint32_t q0 = D[base_d + 4 + 3 * bi], q1 = D[base_d + 5 + 3 * bi], f = D[base_d + 6 + 3 * bi];
                        uint32_t base;
                        uint4 w = P.a[j - 1][i];
                        if (rows16_finish(q0, q1, f, rv, base, w)) mark16(base, w, rv, f);
                        else
                            for (uint32_t b = 4u * (uint32_t)i; b < 4u * (uint32_t)i + 4u && b < npre; b++)
                                decode_block_wave4(GlobalBytes{L.payload}, D[base_d + 4 + 3 * b], D[base_d + 5 + 3 * b],
                                                   D[base_d + 6 + 3 * b], mark4);
                    }
                }
                for (uint32_t b = bl + WABLK; b < bl + nblk; b++)        // blocks beyond the prefetched dozen
                    decode_block_wave4(GlobalBytes{L.payload}, L.skip[b].byte_off, L.skip[b + 1].byte_off, L.skip[b].first_doc, mark4);
            }
            // finalise: byte map -> bitmap words, tombstones cleared, survivors counted
            const uint32_t n4 = n * 0x01010101u;
            uint32_t mine = 0;
            for (uint32_t wi = (uint32_t)l; wi < nwords; wi += 64u) {
                const uint4 a = *reinterpret_cast<const uint4 *>(&sm.map[32u * wi]);
                const uint4 b = *reinterpret_cast<const uint4 *>(&sm.map[32u * wi + 16u]);
                uint32_t word = bytes_eq_mask_w(a.x, n4) | (bytes_eq_mask_w(a.y, n4) << 4) | (bytes_eq_mask_w(a.z, n4) << 8) |
                                (bytes_eq_mask_w(a.w, n4) << 12) | (bytes_eq_mask_w(b.x, n4) << 16) | (bytes_eq_mask_w(b.y, n4) << 20) |
                                (bytes_eq_mask_w(b.z, n4) << 24) | (bytes_eq_mask_w(b.w, n4) << 28);
                if (wi == nwords - 1u && (mspan & 31u) != 31u) word &= (2u << (mspan & 31u)) - 1u;
                if (p.tomb) {
                    const uint32_t tw = (mlo >> 5) + wi;
                    if (tw < p.tomb_nwords) word &= ~p.tomb[tw];
                }
                slot[wi] = word;
                mine += (uint32_t)__popc(word);
            }
            mine = wave_sum(mine);
            if (l == 0) {
                p.tile_count[m] = mine;
                if (mine) atomicAdd(&p.sums[m >> 6], mine);
            }
        } else {
            // ---------- gallop path (sparse / skewed range), wave-local ----------
            uint32_t *cand = reinterpret_cast<uint32_t *>(sm.map);
            uint8_t *hit = sm.map + 4096;
            uint32_t *wbuf = reinterpret_cast<uint32_t *>(sm.map + 5120);
            uint32_t ncand = 0;
            const ListView drv = p.lists[0];
            for (uint32_t r = 0; r < nrows; r++) {
                const uint32_t pb = r * 256u;
                const uint32_t c = decode_block_wave(GlobalBytes{drv.payload}, D[4 + 3 * r], D[5 + 3 * r], D[6 + 3 * r],
                                                     [&](uint32_t ix, uint32_t id) { cand[pb + ix] = id; hit[pb + ix] = 1; });
                ncand = pb + c;
            }
            __threadfence_block();
            for (uint32_t j = 1; j < n; j++) {
                const ListView L = p.lists[j];
                const uint32_t base_d = 16u + 40u * (j - 1u);
                const uint32_t bl = D[base_d + 1], bh = bl + D[base_d];
                for (uint32_t basei = 0; basei < ncand; basei += 64u) {
                    const uint32_t pi = basei + (uint32_t)l;
                    const bool alive = pi < ncand && hit[pi] == (uint8_t)j;
                    const uint32_t c = alive ? cand[pi] : 0u;
                    uint32_t blk = NONE;
                    if (alive && bl < bh) {
                        const uint32_t ub = skip_upper_bound(L.skip, bl, bh, c);
                        if (ub > bl) blk = ub - 1u;
                    }
                    unsigned long long pending = __ballot(blk != NONE);
                    while (pending) {
                        const int leader = __ffsll((long long)pending) - 1;
                        const uint32_t cur = wave_bcast(blk, leader);
                        const uint32_t cnt = decode_block_wave(GlobalBytes{L.payload}, L.skip[cur].byte_off, L.skip[cur + 1].byte_off,
                                                               L.skip[cur].first_doc, [&](uint32_t ix, uint32_t id) { wbuf[ix] = id; });
                        __threadfence_block();
                        if (blk == cur) {
                            uint32_t a = 0, e = cnt;
                            while (a < e) {
                                const uint32_t mid = (a + e) >> 1;
                                if (wbuf[mid] < c) a = mid + 1u; else e = mid;
                            }
                            if (a < cnt && wbuf[a] == c) hit[pi] = (uint8_t)(j + 1u);
                        }
                        __threadfence_block();
                        pending &= ~__ballot(blk == cur);
                    }
                }
            }
            uint32_t total = 0;
            for (uint32_t basei = 0; basei < ncand; basei += 64u) {
                const uint32_t pi = basei + (uint32_t)l;
                bool keep = pi < ncand && hit[pi] == (uint8_t)n;
                if (keep && p.tomb) {
                    const uint32_t v = cand[pi], w = v >> 5;
                    if (w < p.tomb_nwords && ((p.tomb[w] >> (v & 31u)) & 1u)) keep = false;
                }
                if (keep && pi > 0 && cand[pi - 1] == cand[pi]) keep = false;
                const uint32_t incl = wave_incl_scan(keep ? 1u : 0u);
                if (keep) slot[total + incl - 1u] = cand[pi];
                total += wave_bcast(incl, 63);
            }
            if (l == 0) {
                p.tile_count[m] = total | LIST_FLAG;
                if (total) atomicAdd(&p.sums[m >> 6], total);
            }
        }
        if (!has1) break;
        P = PN;
        m = m1;
        m1 = m2;
    }
}

hipError_t launch_intersect_wave(const IntersectParams &p, hipStream_t s) {
    const uint64_t nthr = (uint64_t)p.n_tiles * p.n_lists * 64u;
    const uint64_t pthr = std::max<uint64_t>(nthr, p.n_sums);
    hipLaunchKernelGGL(k_isectw_partition, dim3((unsigned)((pthr + 255) / 256)), dim3(256), 0, s, p);
    uint32_t grid = (p.n_tiles + 3u) / 4u;
    if (grid > p.max_grid) grid = p.max_grid;
    switch (p.n_lists) {
        case 2: hipLaunchKernelGGL(k_isectw_tiles<2>, dim3(grid), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(k_isectw_tiles<3>, dim3(grid), dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL(k_isectw_tiles<4>, dim3(grid), dim3(256), 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ii2
