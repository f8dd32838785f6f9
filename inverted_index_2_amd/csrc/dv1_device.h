// dv1_device.h — wave64 device helpers for the DV1 block-Δ-varint posting format (gfx950).
//
// DV1: a list is cut into blocks of 256 postings.  Block b has a skip entry
// {first_doc, byte_off}; its payload holds the LEB128 varint gaps of postings 1..cnt-1
// (posting 0 is first_doc), at payload[skip[b].byte_off .. skip[b+1].byte_off).
// Only the last block of a list may be short; its count is 1 + (#terminator bytes).
//
// Decode is additive: a byte q contributes (b & 0x7f) << 7*k(q), k = number of
// continuation bytes directly before it, and a posting's id is first_doc plus the
// inclusive prefix sum of the contributions up to its terminator byte.  So one wave
// decodes a 256-byte chunk with 4 bytes per lane and one DPP prefix sum — no per-varint
// serial dependency and no divergence on varint length.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ii2.h"

namespace ii2 {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for the
// wave's outstanding global loads / stores / atomics (vmcnt), so prefetches and result stores keep
// draining behind it.  Use only where the waves exchange data through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes of a wave (all lanes must be active).
// row_shr 1/2/4/8 inside each row of 16, then row_bcast:15 / row_bcast:31 carry the row
// totals across rows (gfx9 DPP).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return x;
}

// Reference form of the same scan (shuffles); used by the self-test to check the DPP form.
__device__ __forceinline__ uint32_t wave_incl_scan_shfl(uint32_t x) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
        if (l >= d) x += y;
    }
    return x;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(x), 63);
}

__device__ __forceinline__ uint32_t wave_bcast(uint32_t x, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)x, lane);
}

// value of lane-1 (lane 0 receives `first`)
__device__ __forceinline__ uint32_t wave_shift_up1(uint32_t x, uint32_t first) {
    uint32_t y = (uint32_t)__shfl_up((int)x, 1, 64);
    return lane_id() == 0 ? first : y;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);     // gfx950: one global_load_dword, any byte alignment
    return w;
}

// Number of terminator bytes (bit7 clear) among the low `nb` bytes of w.
__device__ __forceinline__ uint32_t count_terminators(uint32_t w, uint32_t nb) {
    uint32_t valid = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    uint32_t term = ~w & 0x80808080u & valid;
    return (uint32_t)__popc(term);
}

// Byte sources for the block decoder: 4 payload bytes at byte offset q (any alignment).
struct GlobalBytes {
    const uint8_t *__restrict__ p;
    __device__ __forceinline__ uint32_t operator()(uint32_t q) const { return load_u32_unaligned(p + q); }
};
// LDS-staged payload: two aligned dword reads + v_alignbyte (LDS reads must be dword aligned)
struct LdsBytes {
    const uint8_t *p;    // LDS, 4-byte aligned base
    __device__ __forceinline__ uint32_t operator()(uint32_t q) const {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(p + (q & ~3u));
        const uint32_t lo = w[0], hi = w[1];
        return __builtin_amdgcn_alignbyte(hi, lo, q & 3u);
    }
};

// Decode one DV1 block with one wave, four postings at a time.
// emit4(ix, id0, id1, id2, id3, mask) is called once per 256-byte chunk by every lane: bit j of
// mask says byte j of the lane's dword terminates a posting whose id is id_j; the lane's valid
// postings have consecutive indices starting at ix.  Posting 0 (first_doc) is delivered first,
// by lane 0, as emit4(0, first_doc, 0, 0, 0, 1) (mask 0 on the other lanes).  Handing the four
// candidates over together lets the caller issue its LDS reads back to back before any write.
// Returns the number of postings in the block (wave-uniform).  All 64 lanes must call with
// wave-uniform q0, q1, first_doc.
template <class Load, class Emit4>
__device__ __forceinline__ uint32_t decode_block_wave4(Load load, uint32_t q0, uint32_t q1, uint32_t first_doc, Emit4 emit4) {
    const int l = lane_id();
    emit4(0u, first_doc, 0u, 0u, 0u, l == 0 ? 1u : 0u);
    uint32_t carry_id = first_doc;   // running id after the last byte of the previous chunk
    uint32_t carry_cnt = 1;          // postings emitted so far
    uint32_t carry_run = 0;          // continuation bytes pending at the chunk boundary
    for (uint32_t q = q0; q < q1; q += 256) {
        const uint32_t myq = q + 4u * (uint32_t)l;
        uint32_t nb = myq < q1 ? (q1 - myq < 4u ? q1 - myq : 4u) : 0u;
        uint32_t w = nb ? load(myq) : 0u;
        if (nb < 4) w &= nb ? ((1u << (8 * nb)) - 1u) : 0u;
        const uint32_t cont = w & 0x80808080u;
        const bool plain = (__ballot(cont != 0) == 0ull) && carry_run == 0;   // wave-uniform
        uint32_t c0, c1, c2, c3;         // per-byte contributions
        uint32_t mask;                   // per-byte terminator flags (valid bytes only)
        if (plain) {
            c0 = w & 0xFFu; c1 = (w >> 8) & 0xFFu; c2 = (w >> 16) & 0xFFu; c3 = w >> 24;
            mask = (1u << nb) - 1u;
        } else {
            const uint32_t k0b = cont & 0x80u, k1b = cont & 0x8000u, k2b = cont & 0x800000u, k3b = cont & 0x80000000u;
            // trailing continuation run of this lane's valid bytes
            uint32_t trail = 0;
            if (nb == 4) trail = k3b ? (k2b ? (k1b ? (k0b ? 4u : 3u) : 2u) : 1u) : 0u;
            else if (nb == 3) trail = k2b ? (k1b ? (k0b ? 3u : 2u) : 1u) : 0u;
            else if (nb == 2) trail = k1b ? (k0b ? 2u : 1u) : 0u;
            else if (nb == 1) trail = k0b ? 1u : 0u;
            // run pending before byte 0 of this lane: lane-1's trail; a lane of four
            // continuation bytes extends the run of the lane before it (varints are <= 5 bytes)
            uint32_t prev = wave_shift_up1(trail, carry_run);
            uint32_t prev2 = wave_shift_up1(prev, 0u);
            uint32_t prev_nb4 = wave_shift_up1(trail == 4u ? 1u : 0u, 0u);
            uint32_t r0 = prev + (prev_nb4 ? prev2 : 0u);
            if (r0 > 4u) r0 = 4u;                 // garbage guard: keeps the shift < 32
            uint32_t r1 = k0b ? r0 + 1u : 0u;
            uint32_t r2 = k1b ? r1 + 1u : 0u;
            uint32_t r3 = k2b ? r2 + 1u : 0u;
            if (r1 > 4u) r1 = 4u;
            if (r2 > 4u) r2 = 4u;
            if (r3 > 4u) r3 = 4u;
            c0 = (w & 0x7Fu) << (7u * r0);
            c1 = ((w >> 8) & 0x7Fu) << (7u * r1);
            c2 = ((w >> 16) & 0x7Fu) << (7u * r2);
            c3 = ((w >> 24) & 0x7Fu) << (7u * r3);
            mask = ((nb > 0 && !k0b) ? 1u : 0u) | ((nb > 1 && !k1b) ? 2u : 0u) | ((nb > 2 && !k2b) ? 4u : 0u) |
                   ((nb > 3 && !k3b) ? 8u : 0u);
            carry_run = wave_bcast(trail == 4u ? (r0 + 4u > 4u ? 4u : r0 + 4u) : trail, 63);
        }
        const uint32_t s = c0 + c1 + c2 + c3;
        const uint32_t tc = (uint32_t)__popc(mask);
        const uint32_t si = wave_incl_scan(s);
        const uint32_t ti = wave_incl_scan(tc);
        const uint32_t id0 = carry_id + (si - s) + c0;
        const uint32_t id1 = id0 + c1, id2 = id1 + c2, id3 = id2 + c3;
        emit4(carry_cnt + (ti - tc), id0, id1, id2, id3, mask);
        carry_id += wave_bcast(si, 63);
        carry_cnt += wave_bcast(ti, 63);
    }
    return carry_cnt;
}

// Per-posting form: emit(idx_in_block, doc_id) once per posting, by the lane that owns the
// posting's terminator byte (lane 0 also owns posting 0).
template <class Load, class Emit>
__device__ __forceinline__ uint32_t decode_block_wave(Load load, uint32_t q0, uint32_t q1, uint32_t first_doc, Emit emit) {
    return decode_block_wave4(load, q0, q1, first_doc,
                              [&](uint32_t ix, uint32_t id0, uint32_t id1, uint32_t id2, uint32_t id3, uint32_t mask) {
                                  if (mask & 1u) { emit(ix, id0); ix++; }
                                  if (mask & 2u) { emit(ix, id1); ix++; }
                                  if (mask & 4u) { emit(ix, id2); ix++; }
                                  if (mask & 8u) { emit(ix, id3); ix++; }
                              });
}

template <class Emit>
__device__ __forceinline__ uint32_t decode_block_wave(const uint8_t *__restrict__ payload, uint32_t q0, uint32_t q1,
                                                      uint32_t first_doc, Emit emit) {
    return decode_block_wave(GlobalBytes{payload}, q0, q1, first_doc, emit);
}

// ---- four blocks per wave: one 16-lane DPP row per block, 16 payload bytes per lane ---------
// Inclusive prefix sum inside each row of 16 lanes (no cross-row steps).
__device__ __forceinline__ uint32_t row_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    return x;
}

// 16 payload bytes at byte offset q (any alignment) as four dwords
struct GlobalBytes16 {
    const uint8_t *__restrict__ p;
    __device__ __forceinline__ uint4 operator()(uint32_t q) const {
        uint4 v;
        __builtin_memcpy(&v, p + q, 16);
        return v;
    }
};
struct LdsBytes16 {
    const uint8_t *p;    // LDS, 4-byte aligned base; 20 bytes are read
    __device__ __forceinline__ uint4 operator()(uint32_t q) const {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(p + (q & ~3u));
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
        const uint32_t sh = q & 3u;
        return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                          __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
    }
};

// Row decode of one block per 16-lane row (fast path: payload <= 256 bytes, one byte per gap).
// Every lane passes ITS row's block: payload byte range [q0, q1) and first_doc; rows without a
// block pass q0 == q1 and row_valid = false.  On success each lane gets base = id of the posting
// just before its 16 bytes and w = its 16 gap bytes (bytes past the block's end are 0, so the
// running id simply stops advancing there) and true is returned.  Returns false — for the whole
// wave, nothing decoded — when any row's block needs the general decoder.
// Second half of the row decode, on bytes that are already in registers (`w` = the 16 bytes at
// q0 + 16*rl, undefined when that is past q1): masks the tail, checks the fast-path conditions,
// and turns w into gap bytes + base id.  See decode_rows16.
__device__ __forceinline__ bool rows16_finish(uint32_t q0, uint32_t q1, uint32_t first_doc, bool row_valid, uint32_t &base, uint4 &w) {
    const uint32_t rl = (uint32_t)lane_id() & 15u;
    const uint32_t len = row_valid ? q1 - q0 : 0u;
    const uint32_t myoff = 16u * rl;
    const uint32_t nb = len > myoff ? (len - myoff < 16u ? len - myoff : 16u) : 0u;
    if (nb == 0u) w = make_uint4(0, 0, 0, 0);
    else if (nb < 16u) {     // zero the bytes past the end of the block
        const uint32_t n0 = nb < 4u ? nb : 4u, n1 = nb < 4u ? 0u : (nb < 8u ? nb - 4u : 4u);
        const uint32_t n2 = nb < 8u ? 0u : (nb < 12u ? nb - 8u : 4u), n3 = nb < 12u ? 0u : nb - 12u;
        w.x &= n0 >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n0)) - 1u);
        w.y &= n1 >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n1)) - 1u);
        w.z &= n2 >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n2)) - 1u);
        w.w &= n3 >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n3)) - 1u);
    }
    const bool hard = len > 256u || (((w.x | w.y | w.z | w.w) & 0x80808080u) != 0u);
    if (__ballot(hard) != 0ull) return false;
    uint32_t s = __builtin_amdgcn_sad_u8(w.x, 0u, 0u);
    s = __builtin_amdgcn_sad_u8(w.y, 0u, s);
    s = __builtin_amdgcn_sad_u8(w.z, 0u, s);
    s = __builtin_amdgcn_sad_u8(w.w, 0u, s);
    const uint32_t incl = row_incl_scan(s);
    base = first_doc + incl - s;
    return true;
}

template <class Load16>
__device__ __forceinline__ bool decode_rows16(Load16 load16, uint32_t q0, uint32_t q1, uint32_t first_doc, bool row_valid,
                                              uint32_t &base, uint4 &w) {
    const uint32_t rl = (uint32_t)lane_id() & 15u;
    const uint32_t len = row_valid ? q1 - q0 : 0u;
    w = len > 16u * rl ? load16(q0 + 16u * rl) : make_uint4(0, 0, 0, 0);
    return rows16_finish(q0, q1, first_doc, row_valid, base, w);
}

// General row decoder: one block per 16-lane row, 16 payload bytes per lane and pass (256 per row
// and pass), any gap widths.  Each lane walks its 16 bytes serially (its position inside a varint
// comes from the four bytes before them), the row scans the lanes' sums and posting counts, then
// every lane emits the postings that end inside its bytes: emit(idx_in_block, doc_id), posting 0
// by the row's lane 0.  All 64 lanes must call; rows without a block pass row_valid = false.
// Reads up to 16 bytes past q1 (segments carry that padding).
// Returns the number of postings of the lane's row (1 for an invalid row's lanes is meaningless: check row_valid).
template <class Emit>
__device__ __forceinline__ uint32_t decode_rows16_any(const uint8_t *__restrict__ payload, uint32_t q0, uint32_t q1, uint32_t first_doc,
                                                      bool row_valid, Emit emit) {
    const uint32_t rl = (uint32_t)lane_id() & 15u;
    const int row_last = lane_id() | 15;
    const uint32_t len = row_valid ? q1 - q0 : 0u;
    if (row_valid && rl == 0u) emit(0u, first_doc);
    uint32_t run_id = first_doc, run_ix = 1u;
    for (uint32_t off = 0; __ballot(off < len) != 0ull; off += 256u) {
        const uint32_t my = off + 16u * rl;
        const uint32_t nb = my < len ? (len - my < 16u ? len - my : 16u) : 0u;
        uint4 w4 = make_uint4(0, 0, 0, 0);
        uint32_t prev = 0;
        if (nb) {
            __builtin_memcpy(&w4, payload + q0 + my, 16);
            if (my) prev = load_u32_unaligned(payload + q0 + my - 4u);
        }
        uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
        if (nb < 16u) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t n = nb > 4u * j ? (nb - 4u * j < 4u ? nb - 4u * j : 4u) : 0u;
                w[j] &= n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u);
            }
        }
        // continuation bytes pending right before my first byte (varints are <= 5 bytes)
        uint32_t sh = 7u * ((uint32_t)__clz((int)~(prev | 0x7F7F7F7Fu)) >> 3);
        uint32_t val[16];
        uint32_t sum = 0, tmask = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            sum += (c & 0x7Fu) << sh;
            if (c & 0x80u) sh = sh < 28u ? sh + 7u : 28u;
            else { sh = 0u; tmask |= 1u << i; }
            val[i] = sum;
        }
        tmask &= (1u << nb) - 1u;           // nb <= 16
        const uint32_t cnt = (uint32_t)__popc(tmask);
        const uint32_t isum = row_incl_scan(sum), icnt = row_incl_scan(cnt);
        const uint32_t base = run_id + isum - sum;
        uint32_t ix = run_ix + icnt - cnt;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if ((tmask >> i) & 1u) { emit(ix, base + val[i]); ix++; }
        run_id += (uint32_t)__shfl((int)isum, row_last, 64);
        run_ix += (uint32_t)__shfl((int)icnt, row_last, 64);
    }
    return run_ix;
}

// Posting count of a block without decoding ids (1 + terminators).  Wave-uniform result.
__device__ __forceinline__ uint32_t count_block_wave(const uint8_t *__restrict__ payload, uint32_t q0, uint32_t q1) {
    const int l = lane_id();
    uint32_t n = 0;
    for (uint32_t q = q0; q < q1; q += 256) {
        const uint32_t myq = q + 4u * (uint32_t)l;
        uint32_t nb = myq < q1 ? (q1 - myq < 4u ? q1 - myq : 4u) : 0u;
        uint32_t w = nb ? load_u32_unaligned(payload + myq) : 0xFFFFFFFFu;
        n += count_terminators(w, nb);
    }
    return 1u + wave_sum(n);
}

// first index i in [lo, hi) with skip[i].first_doc > x   (upper bound on first_doc)
__device__ __forceinline__ uint32_t skip_upper_bound(const ii2_skip *__restrict__ skip, uint32_t lo, uint32_t hi, uint32_t x) {
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (skip[mid].first_doc <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// first index i in [lo, hi) with get(i) > x, for an ascending sequence that is close to uniform between vlo (a lower bound of
// get(lo)) and vhi (an upper bound of get(hi - 1)): a linear guess, a doubling walk away from it until x is bracketed, then
// bisection inside the bracket — a handful of dependent loads instead of log2(hi - lo).  Exact for any ascending input.
template <class Get>
__device__ __forceinline__ uint32_t upper_bound_guess(Get get, uint32_t lo, uint32_t hi, uint32_t x, uint32_t vlo, uint32_t vhi) {
    if (lo >= hi) return lo;
    if (hi - lo > 4u && vhi > vlo) {
        const uint64_t rel = x > vlo ? (uint64_t)(x - vlo) : 0ull;
        uint64_t g64 = (uint64_t)lo + rel * (uint64_t)(hi - lo) / ((uint64_t)(vhi - vlo) + 1ull);
        uint32_t g = g64 >= hi ? hi - 1u : (uint32_t)g64;
        if (get(g) <= x) {                                  // answer in (g, hi]: walk up
            uint32_t a = g + 1u, step = 1u;
            while (a < hi) {
                const uint32_t pr = a + step - 1u < hi ? a + step - 1u : hi - 1u;
                if (get(pr) <= x) { a = pr + 1u; step <<= 1; } else { hi = pr; break; }
            }
            lo = a;                                         // get(i) <= x for i < lo; get(hi) > x or hi is the end
        } else {                                            // answer in [lo, g]: walk down
            uint32_t b = g, step = 1u;
            while (b > lo) {
                const uint32_t pr = b - lo > step ? b - step : lo;
                if (get(pr) > x) { b = pr; step <<= 1; } else { lo = pr + 1u; break; }
            }
            hi = b;
        }
    }
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (get(mid) <= x) lo = mid + 1u; else hi = mid; }
    return lo;
}

__device__ __forceinline__ unsigned varint_len(uint32_t v) {
    return v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u;
}

}  // namespace ii2
