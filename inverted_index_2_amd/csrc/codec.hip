// codec.hip — DV1 encode / decode kernels, tombstone bitmap build, device self-test.
// Encode replaces Writer.Append -> intcomp.CompressUint32 (reference file/writer.go:32-59),
// decode replaces Reader.Next -> intcomp.UncompressUint32 (file/reader.go:79-100) — as
// roles; the byte format is this build's own.
#include "dv1_device.h"
#include "internal.h"

namespace ii2 {

// ---- encode ---------------------------------------------------------------------------
__global__ void k_enc_list_blocks(const uint64_t *__restrict__ post_off, uint64_t n_lists, uint32_t *__restrict__ nblk) {
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l > n_lists) return;
    if (l == n_lists) { nblk[l] = 0; return; }
    uint64_t n = post_off[l + 1] - post_off[l];
    nblk[l] = (uint32_t)((n + II2_DV1_BLOCK - 1) / II2_DV1_BLOCK);
}

// list that owns block b: last l with blk_off[l] <= b
__device__ __forceinline__ uint64_t owner_list(const uint32_t *__restrict__ blk_off, uint64_t n_lists, uint32_t b) {
    uint64_t lo = 0, hi = n_lists;      // invariant: blk_off[lo] <= b < blk_off[hi]
    while (hi - lo > 1) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (blk_off[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

// A wave encodes ENC_BPW consecutive blocks, one after the other: lane i handles postings 4i..4i+3 of a block.  The list that
// owns the wave's first block is found by bisection (some twenty dependent loads in a table of a million lists); the owners
// of the following blocks by walking on from there (blocks and lists ascend together), so the search is paid once per
// ENC_BPW blocks.  WRITE: the block's bytes are put together in wave-private LDS and leave as 16-byte stores.
constexpr uint32_t ENC_BPW = 8;
constexpr uint32_t ENC_STAGE = 1280 + 32;      // a block's payload (255 gaps of <= 5 bytes) + room to start it at its address mod 16
template <bool WRITE>
__global__ __launch_bounds__(256) void k_enc_blocks(const uint64_t *__restrict__ post_off, const uint32_t *__restrict__ blk_off,
                                                    uint64_t n_lists, const uint32_t *__restrict__ values, uint64_t n_blocks,
                                                    uint32_t *__restrict__ sizes, const uint64_t *__restrict__ byte_off64,
                                                    ii2_skip *__restrict__ skip, uint8_t *__restrict__ payload, uint64_t n_postings,
                                                    uint32_t *__restrict__ blk_list) {
    __shared__ __align__(16) uint8_t stage[WRITE ? 4 : 1][WRITE ? ENC_STAGE : 16];
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int l = lane_id();
    const uint64_t b_begin = wave * ENC_BPW;
    if (b_begin > n_blocks) return;
    uint64_t li = b_begin < n_blocks ? owner_list(blk_off, n_lists, (uint32_t)b_begin) : 0;
    uint32_t li_end = b_begin < n_blocks ? blk_off[li + 1] : 0u;        // first block past list li
    for (uint64_t b = b_begin; b < b_begin + ENC_BPW && b <= n_blocks; b++) {
        if (b == n_blocks) {          // sentinel
            if (l == 0) {
                if (!WRITE) sizes[b] = 0;
                else {
                    skip[b].first_doc = n_postings ? values[n_postings - 1] : 0u;
                    skip[b].byte_off = (uint32_t)byte_off64[b];
                }
            }
            break;
        }
        while ((uint32_t)b >= li_end) { li++; li_end = blk_off[li + 1]; }   // (empty lists in between own no block)
        const uint64_t p0 = post_off[li] + (uint64_t)((uint32_t)b - blk_off[li]) * II2_DV1_BLOCK;
        const uint64_t pe = post_off[li + 1];
        const uint32_t cnt = (uint32_t)(pe - p0 < II2_DV1_BLOCK ? pe - p0 : II2_DV1_BLOCK);
        uint32_t d[4];
        uint32_t len = 0;
        {   // my four postings and the one before them
            const uint32_t i0 = 4u * (uint32_t)l;
            uint32_t v[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint32_t i = i0 + (uint32_t)j;
                v[j] = (i >= 1u && i <= cnt) ? values[p0 + i - 1u] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t i = i0 + (uint32_t)j;
                d[j] = 0;
                if (i >= 1u && i < cnt) {
                    d[j] = v[j + 1] - v[j];
                    len += varint_len(d[j]);
                }
            }
        }
        const uint32_t incl = wave_incl_scan(len);
        if (!WRITE) {
            if (l == 63) sizes[b] = incl;
            continue;
        }
        const uint64_t base = byte_off64[b];
        const uint32_t total = wave_bcast(incl, 63);
        if (l == 0) {
            skip[b].first_doc = values[p0];
            skip[b].byte_off = (uint32_t)base;
            if (blk_list) blk_list[b] = (uint32_t)li;         // the block's owner (what k_list_last_doc derives for imported segments)
        }
        // the block's bytes in LDS, shifted so that LDS offset and global address agree mod 16
        uint8_t *st = stage[threadIdx.x >> 6];
        const uint32_t mis = (uint32_t)((uintptr_t)(payload + base) & 15u);
        uint32_t q = mis + (incl - len);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i = 4u * (uint32_t)l + (uint32_t)j;
            if (i >= 1 && i < cnt) {
                uint32_t v = d[j];
                while (v >= 0x80u) { st[q++] = (uint8_t)(v | 0x80u); v >>= 7; }
                st[q++] = (uint8_t)v;
            }
        }
        // (wave-private LDS: program order is enough) whole 16-byte pieces leave as one store each, the ragged ends byte by byte
        uint8_t *dst = payload + base - mis;                                  // 16-byte aligned
        const uint32_t end = mis + total;
        for (uint32_t o = 16u * (uint32_t)l; o < end; o += 1024u) {
            if (o >= mis && o + 16u <= end) {
                *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(st + o);
            } else {
                const uint32_t a0 = o > mis ? o : mis, a1 = o + 16u < end ? o + 16u : end;
                for (uint32_t x = a0; x < a1; x++) dst[x] = st[x];
            }
        }
    }
}

hipError_t launch_enc_list_blocks(const uint64_t *post_off, uint64_t n_lists, uint32_t *nblk, hipStream_t s) {
    uint64_t n = n_lists + 1;
    hipLaunchKernelGGL(k_enc_list_blocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, post_off, n_lists, nblk);
    return hipGetLastError();
}

hipError_t launch_enc_block_sizes(const uint64_t *post_off, const uint32_t *blk_off, uint64_t n_lists,
                                  const uint32_t *values, uint64_t n_blocks, uint32_t *sizes, ii2_skip *skip, hipStream_t s) {
    uint64_t waves = (n_blocks + 1 + ENC_BPW - 1) / ENC_BPW;
    hipLaunchKernelGGL(k_enc_blocks<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, post_off, blk_off, n_lists,
                       values, n_blocks, sizes, (const uint64_t *)nullptr, skip, (uint8_t *)nullptr, (uint64_t)0, (uint32_t *)nullptr);
    return hipGetLastError();
}

hipError_t launch_enc_write(const uint64_t *post_off, const uint32_t *blk_off, uint64_t n_lists,
                            const uint32_t *values, uint64_t n_blocks, const uint64_t *byte_off64,
                            ii2_skip *skip, uint8_t *payload, uint64_t n_postings, uint32_t *blk_list, hipStream_t s) {
    uint64_t waves = (n_blocks + 1 + ENC_BPW - 1) / ENC_BPW;
    hipLaunchKernelGGL(k_enc_blocks<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, post_off, blk_off, n_lists,
                       values, n_blocks, (uint32_t *)nullptr, byte_off64, skip, payload, n_postings, blk_list);
    return hipGetLastError();
}

// per list: posting count and last doc id, straight from the CSR that is being encoded (an imported segment has to decode every
// list's last block for them: k_list_last_doc)
__global__ void k_enc_list_meta(const uint64_t *__restrict__ post_off, const uint32_t *__restrict__ values, uint64_t n_lists,
                                uint32_t *__restrict__ cnt, uint32_t *__restrict__ last_doc) {
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lists) return;
    const uint64_t p0 = post_off[l], p1 = post_off[l + 1];
    cnt[l] = (uint32_t)(p1 - p0);
    last_doc[l] = p1 > p0 ? values[p1 - 1] : 0u;
}
hipError_t launch_enc_list_meta(const uint64_t *post_off, const uint32_t *values, uint64_t n_lists, uint32_t *cnt, uint32_t *last_doc, hipStream_t s) {
    if (n_lists == 0) return hipSuccess;
    hipLaunchKernelGGL(k_enc_list_meta, dim3((unsigned)((n_lists + 255) / 256)), dim3(256), 0, s, post_off, values, n_lists, cnt, last_doc);
    return hipGetLastError();
}

// ---- decode ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dec_block_counts(const ii2_skip *__restrict__ skip, const uint8_t *__restrict__ payload,
                                                          uint64_t n_blocks, uint32_t *__restrict__ counts) {
    const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b > n_blocks) return;
    if (b == n_blocks) { if (lane_id() == 0) counts[b] = 0; return; }
    const uint32_t q0 = skip[b].byte_off, q1 = skip[b + 1].byte_off;
    const uint32_t c = count_block_wave(payload, q0, q1);
    if (lane_id() == 0) counts[b] = c;
}

__global__ __launch_bounds__(256) void k_dec_write(const ii2_skip *__restrict__ skip, const uint8_t *__restrict__ payload,
                                                   uint64_t n_blocks, const uint64_t *__restrict__ bpo, uint32_t *__restrict__ values) {
    const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_blocks) return;
    const uint32_t q0 = skip[b].byte_off, q1 = skip[b + 1].byte_off;
    uint32_t *out = values + bpo[b];
    decode_block_wave(payload, q0, q1, skip[b].first_doc, [&](uint32_t ix, uint32_t id) { out[ix] = id; });
}

__global__ void k_gather_post_off(const uint32_t *__restrict__ blk_off, const uint64_t *__restrict__ bpo, uint64_t n_lists,
                                  uint64_t *__restrict__ post_off) {
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l <= n_lists) post_off[l] = bpo[blk_off[l]];
}

hipError_t launch_dec_block_counts(const ii2_skip *skip, const uint8_t *payload, uint64_t n_blocks, uint32_t *counts, hipStream_t s) {
    uint64_t waves = n_blocks + 1;
    hipLaunchKernelGGL(k_dec_block_counts, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, skip, payload, n_blocks, counts);
    return hipGetLastError();
}

hipError_t launch_dec_write(const ii2_skip *skip, const uint8_t *payload, uint64_t n_blocks, const uint64_t *bpo,
                            uint32_t *values, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dec_write, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, s, skip, payload, n_blocks, bpo, values);
    return hipGetLastError();
}

hipError_t launch_gather_post_off(const uint32_t *blk_off, const uint64_t *bpo, uint64_t n_lists, uint64_t *post_off, hipStream_t s) {
    uint64_t n = n_lists + 1;
    hipLaunchKernelGGL(k_gather_post_off, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blk_off, bpo, n_lists, post_off);
    return hipGetLastError();
}

// per list: last doc id, posting count, and the owner entry of each of its blocks — one wave
// decodes the list's last block (all other blocks hold exactly II2_DV1_BLOCK postings)
__global__ __launch_bounds__(256) void k_list_last_doc(const uint32_t *__restrict__ blk_off, const ii2_skip *__restrict__ skip,
                                                       const uint8_t *__restrict__ payload, uint64_t n_lists,
                                                       uint32_t *__restrict__ cnt, uint32_t *__restrict__ blk_list,
                                                       uint32_t *__restrict__ last_doc) {
    const uint64_t li = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (li >= n_lists) return;
    const uint32_t b0 = blk_off[li], b1 = blk_off[li + 1];
    if (b1 == b0) { if (lane_id() == 0) { last_doc[li] = 0; cnt[li] = 0; } return; }
    for (uint32_t b = b0 + (uint32_t)lane_id(); b < b1; b += 64u) blk_list[b] = (uint32_t)li;
    const uint32_t b = b1 - 1u;
    uint32_t mx_ix = 0, mx_id = skip[b].first_doc;
    const uint32_t c = decode_block_wave(payload, skip[b].byte_off, skip[b + 1].byte_off, skip[b].first_doc,
                                         [&](uint32_t ix, uint32_t id) { if (ix >= mx_ix) { mx_ix = ix; mx_id = id; } });
    // the lane holding the highest posting index has the last id
    uint32_t best_ix = mx_ix, best_id = mx_id;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t oi = (uint32_t)__shfl_xor((int)best_ix, d, 64), od = (uint32_t)__shfl_xor((int)best_id, d, 64);
        if (oi > best_ix) { best_ix = oi; best_id = od; }
    }
    if (lane_id() == 0) { last_doc[li] = best_id; cnt[li] = (b1 - b0 - 1u) * II2_DV1_BLOCK + c; }
}

hipError_t launch_list_last_doc(const uint32_t *blk_off, const ii2_skip *skip, const uint8_t *payload, uint64_t n_lists, uint32_t *cnt,
                                uint32_t *blk_list, uint32_t *last_doc, hipStream_t s) {
    if (n_lists == 0) return hipSuccess;
    hipLaunchKernelGGL(k_list_last_doc, dim3((unsigned)((n_lists + 3) / 4)), dim3(256), 0, s, blk_off, skip, payload, n_lists, cnt, blk_list,
                       last_doc);
    return hipGetLastError();
}

// per list {first doc, first doc of its last block, last doc}: mirrored on the host when a segment is created, so that a
// query never has to fetch them (ii2_intersect_async stays enqueue-only)
__global__ void k_list_spans(const uint32_t *__restrict__ blk_off, const ii2_skip *__restrict__ skip, const uint32_t *__restrict__ last_doc,
                             uint64_t n_lists, uint32_t *__restrict__ spans) {
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_lists) return;
    const uint32_t b0 = blk_off[l], b1 = blk_off[l + 1];
    const bool any = b1 > b0;
    spans[3 * l] = any ? skip[b0].first_doc : 0u;
    spans[3 * l + 1] = any ? skip[b1 - 1u].first_doc : 0u;
    spans[3 * l + 2] = any ? last_doc[l] : 0u;
}

hipError_t launch_list_spans(const uint32_t *blk_off, const ii2_skip *skip, const uint32_t *last_doc, uint64_t n_lists, uint32_t *spans, hipStream_t s) {
    if (n_lists == 0) return hipSuccess;
    hipLaunchKernelGGL(k_list_spans, dim3((unsigned)((n_lists + 255) / 256)), dim3(256), 0, s, blk_off, skip, last_doc, n_lists, spans);
    return hipGetLastError();
}

// a gathered segment's part: block numbers of its lists, byte offsets of its blocks and the blocks' owners move up by what lies
// before it
__global__ void k_seg_rebase(uint32_t *__restrict__ blk_off, uint64_t n_lists, uint32_t add_blocks, ii2_skip *__restrict__ skip, uint64_t n_blocks,
                             uint32_t add_bytes, uint32_t *__restrict__ blk_list, uint32_t add_lists) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lists) blk_off[i] += add_blocks;
    if (i < n_blocks) {
        skip[i].byte_off += add_bytes;
        if (blk_list) blk_list[i] += add_lists;
    }
}

hipError_t launch_seg_rebase(uint32_t *blk_off, uint64_t n_lists, uint32_t add_blocks, ii2_skip *skip, uint64_t n_blocks, uint32_t add_bytes,
                             uint32_t *blk_list, uint32_t add_lists, hipStream_t s) {
    const uint64_t n = n_lists > n_blocks ? n_lists : n_blocks;
    if (n == 0 || (add_blocks == 0 && add_bytes == 0 && add_lists == 0)) return hipSuccess;
    hipLaunchKernelGGL(k_seg_rebase, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blk_off, n_lists, add_blocks, skip, n_blocks, add_bytes, blk_list, add_lists);
    return hipGetLastError();
}

// the closing entries of a concatenated segment (written on the device: no host source has to outlive the call)
__global__ void k_seg_close(uint32_t *blk_off_end, uint32_t n_blocks, ii2_skip *skip_end, uint32_t n_bytes, uint8_t *payload_end, uint32_t *blk_list_end) {
    const uint32_t t = threadIdx.x;
    if (t == 0) { *blk_off_end = n_blocks; skip_end->first_doc = 0u; skip_end->byte_off = n_bytes; if (blk_list_end) *blk_list_end = 0xFFFFFFFFu; }
    if (t < 16u) payload_end[t] = 0;
}
hipError_t launch_seg_close(uint32_t *blk_off_end, uint32_t n_blocks, ii2_skip *skip_end, uint32_t n_bytes, uint8_t *payload_end, uint32_t *blk_list_end, hipStream_t s) {
    hipLaunchKernelGGL(k_seg_close, dim3(1), dim3(64), 0, s, blk_off_end, n_blocks, skip_end, n_bytes, payload_end, blk_list_end);
    return hipGetLastError();
}

// structural check of an imported segment: offsets monotone and in range, no block longer than
// 255 five-byte varints — so that no kernel can be steered outside the payload
__global__ void k_validate_seg(const uint32_t *__restrict__ blk_off, uint64_t n_lists, const ii2_skip *__restrict__ skip,
                               uint64_t n_blocks, uint64_t n_bytes, uint32_t *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lists && blk_off[i] > blk_off[i + 1]) atomicOr(bad, 1u);
    if (i == 0 && (blk_off[0] != 0 || blk_off[n_lists] != n_blocks)) atomicOr(bad, 2u);
    if (i < n_blocks) {
        const uint32_t q0 = skip[i].byte_off, q1 = skip[i + 1].byte_off;
        if (q1 < q0 || q1 - q0 > 255u * 5u || q1 > n_bytes) atomicOr(bad, 4u);
    }
}

hipError_t launch_validate_seg(const uint32_t *blk_off, uint64_t n_lists, const ii2_skip *skip, uint64_t n_blocks, uint64_t n_bytes,
                               uint32_t *bad, hipStream_t s) {
    const uint64_t n = (n_lists > n_blocks ? n_lists : n_blocks) + 1;
    hipLaunchKernelGGL(k_validate_seg, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blk_off, n_lists, skip, n_blocks, n_bytes, bad);
    return hipGetLastError();
}

// every block but a list's last must hold exactly II2_DV1_BLOCK postings, the last one 1..II2_DV1_BLOCK:
// the merge places decoded blocks by that rule.  One wave per block (import is a one-time cost).
__global__ __launch_bounds__(256) void k_validate_counts(const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ blk_list,
                                                         const ii2_skip *__restrict__ skip, const uint8_t *__restrict__ payload,
                                                         uint64_t n_blocks, uint32_t *__restrict__ bad) {
    const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_blocks) return;
    const uint32_t li = blk_list[b];
    const uint32_t c = count_block_wave(payload, skip[b].byte_off, skip[b + 1].byte_off);
    const bool last = (uint32_t)b + 1u == blk_off[li + 1];
    if (lane_id() == 0 && (c > II2_DV1_BLOCK || (!last && c != II2_DV1_BLOCK))) atomicOr(bad, 8u);
}

hipError_t launch_validate_counts(const uint32_t *blk_off, const uint32_t *blk_list, const ii2_skip *skip, const uint8_t *payload,
                                  uint64_t n_blocks, uint32_t *bad, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_validate_counts, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, s, blk_off, blk_list, skip, payload, n_blocks, bad);
    return hipGetLastError();
}

// ---- tombstones -----------------------------------------------------------------------
// RemovedLists.Values() (removed_list.go:44-54) as a dense bitmap: bit v set <=> v removed.
// plus a summary with one bit per (1 << TOMB_SUM_SHIFT) docs (random bit tests of a 12 MB bitmap miss L2; the summary mostly does not)
__global__ void k_tomb_build(const uint32_t *__restrict__ removed, uint64_t n, uint32_t *__restrict__ words, uint64_t n_words,
                             uint32_t *__restrict__ summary) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = removed[i];
        uint64_t w = v >> 5;
        if (w < n_words) {
            atomicOr(&words[w], 1u << (v & 31u));
            atomicOr(&summary[v >> (5u + TOMB_SUM_SHIFT)], 1u << ((v >> TOMB_SUM_SHIFT) & 31u));
        }
    }
}

__global__ void k_max_u32(const uint32_t *__restrict__ v, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        m = v[i] > m ? v[i] : m;
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)m, d, 64);
        m = o > m ? o : m;
    }
    if (lane_id() == 0) atomicMax(out, m);
}

// sum of n u32 values into a u64 (one atomic per wave, few workgroups)
__global__ void k_sum_u32(const uint32_t *__restrict__ v, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m += v[i];
    for (int d = 32; d >= 1; d >>= 1) m += (unsigned long long)__shfl_xor((long long)m, d, 64);
    if (lane_id() == 0 && m) atomicAdd(out, m);
}

hipError_t launch_sum_u32(const uint32_t *v, uint64_t n, uint64_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 256) g = 256;
    hipLaunchKernelGGL(k_sum_u32, dim3(g), dim3(256), 0, s, v, n, (unsigned long long *)out);
    return hipGetLastError();
}

hipError_t launch_tomb_build(const uint32_t *removed, uint64_t n, uint32_t *words, uint64_t n_words, uint32_t *summary, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_tomb_build, dim3(g), dim3(256), 0, s, removed, n, words, n_words, summary);
    return hipGetLastError();
}

hipError_t launch_max_u32(const uint32_t *v, uint64_t n, uint32_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_max_u32, dim3(g), dim3(256), 0, s, v, n, out);
    return hipGetLastError();
}

// ---- self-test ------------------------------------------------------------------------
// fail bit 0: DPP scan != shuffle scan; bit 1: wave block decode != scalar decode.
__global__ void k_selftest(uint32_t *fail, uint8_t *gbuf) {
    const int l = lane_id();
    uint32_t seed = 0x9E3779B9u * (uint32_t)(threadIdx.x + 1) + blockIdx.x * 7919u;
    for (int it = 0; it < 8; it++) {
        seed = seed * 1664525u + 1013904223u;
        uint32_t x = seed >> (it * 3);
        if (wave_incl_scan(x) != wave_incl_scan_shfl(x)) atomicOr(fail, 1u);
    }
    // a block with mixed varint lengths, encoded by lane 0 into global scratch
    __shared__ uint32_t ids[4][256];
    __shared__ uint32_t got[4][256];
    const int wv = threadIdx.x >> 6;
    uint8_t *buf = gbuf + (size_t)(blockIdx.x * 4 + wv) * 1408;   // global scratch, 1408 B per wave
    uint32_t nbytes = 0;
    if (l == 0) {
        uint32_t s2 = 12345u + 977u * (uint32_t)(blockIdx.x * 4 + wv);
        uint32_t cur = 1000u + blockIdx.x;
        ids[wv][0] = cur;
        const uint32_t cnt = 256u - (uint32_t)((blockIdx.x * 4 + wv) % 5) * 50u;   // 256,206,156,106,56
        for (uint32_t i = 1; i < 256; i++) {
            s2 = s2 * 1664525u + 1013904223u;
            uint32_t sel = (s2 >> 28) & 7u;
            uint32_t gap = sel < 4 ? 1u + ((s2 >> 8) & 0x3Fu) : sel < 6 ? 128u + ((s2 >> 8) & 0x3FFFu)
                         : sel == 6 ? (1u << 14) + ((s2 >> 4) & 0xFFFFFu) : (1u << 21) + ((s2 >> 3) & 0xFFFFFFu);
            if (blockIdx.x == 0) gap = 1u + (s2 >> 30);            // pure 1-byte block
            if (blockIdx.x == 1 && i == 200) gap = 0xF0000000u;    // 5-byte varint (wraps; still additive)
            cur += gap;
            ids[wv][i] = cur;
            if (i < cnt) {
                uint32_t v = gap;
                while (v >= 0x80u) { buf[nbytes++] = (uint8_t)(v | 0x80u); v >>= 7; }
                buf[nbytes++] = (uint8_t)v;
            }
        }
        for (int z = 0; z < 8; z++) buf[nbytes + z] = 0xFF;
        got[wv][0] = cnt;   // stash
    }
    __threadfence();
    __syncthreads();
    const uint32_t cnt = got[wv][0];
    nbytes = 0;
    // recompute nbytes uniformly (lane 0 knows it; broadcast through LDS-free readlane)
    {
        uint32_t nb0 = 0;
        if (l == 0) {
            for (uint32_t i = 1; i < cnt; i++) nb0 += varint_len(ids[wv][i] - ids[wv][i - 1]);
        }
        nbytes = wave_bcast(nb0, 0);
    }
    __syncthreads();
    const uint32_t first = ids[wv][0];
    for (int i = l; i < 256; i += 64) got[wv][i] = 0xDEADBEEFu;
    __syncthreads();
    uint32_t n = decode_block_wave((const uint8_t *)buf, 0u, nbytes, first, [&](uint32_t ix, uint32_t id) {
        if (ix < 256) got[wv][ix] = id;
    });
    __syncthreads();
    if (n != cnt) atomicOr(fail, 2u);
    for (uint32_t i = (uint32_t)l; i < cnt; i += 64)
        if (got[wv][i] != ids[wv][i]) atomicOr(fail, 2u);
    if (count_block_wave((const uint8_t *)buf, 0u, nbytes) != cnt) atomicOr(fail, 4u);
}

hipError_t launch_selftest(uint32_t *d_fail, uint8_t *d_scratch, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest, dim3(64), dim3(256), 0, s, d_fail, d_scratch);
    return hipGetLastError();
}

}  // namespace ii2
