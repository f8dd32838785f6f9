// comm.cpp — rank-order concatenation of per-GPU result postings over RCCL / xGMI.
//
// Replaces InvertedIndex.Read's shard-order concatenation (reference inverted_index.go:330-339)
// when terms (merge) or doc ranges (one conjunctive query) are sharded over the GPUs of a
// node.  RCCL has no all-gatherv: counts travel with one ncclAllGather of u64, the payload
// with one grouped ncclSend/ncclRecv per peer — the 8 GPUs are fully connected, so every
// peer's copy rides its own xGMI link instead of a ring that one link would bound.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "internal.h"

#define NCCL_TRY(ctx, expr)                                                        \
    do {                                                                           \
        ncclResult_t r_ = (expr);                                                  \
        if (r_ != ncclSuccess) {                                                   \
            (ctx)->err = std::string(#expr) + ": " + ncclGetErrorString(r_);       \
            return II2_ECOMM;                                                      \
        }                                                                          \
    } while (0)

void ii2_comm_destroy_internal(ii2_ctx *ctx) {
    if (ctx && ctx->comm) {
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
}

extern "C" {

int ii2_comm_unique_id(void *id_out) {
    if (!id_out) return II2_EINVAL;
    static_assert(sizeof(ncclUniqueId) <= II2_UNIQUE_ID_BYTES, "unique id does not fit");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return II2_ECOMM;
    std::memset(id_out, 0, II2_UNIQUE_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return II2_OK;
}

int ii2_comm_init(ii2_ctx *ctx, int world, int rank, const void *unique_id) {
    if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    ii2_comm_destroy_internal(ctx);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm;
    NCCL_TRY(ctx, ncclCommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->world = world;
    ctx->rank = rank;
    return II2_OK;
}

// Pure host arithmetic of the exchange (unit-tested on the CPU, tests/test_comm_plan.py): where every rank's
// contribution lands in the concatenation and whether it fits.  offsets has world + 1 entries.
int ii2_gatherv_offsets(const uint64_t *counts, int world, uint64_t cap, uint64_t *offsets) {
    if (!counts || !offsets || world < 1 || world > (int)II2_MAX_RANKS) return II2_EINVAL;
    offsets[0] = 0;
    bool overflow = false;
    for (int r = 0; r < world; r++) {
        offsets[r + 1] = offsets[r] + counts[r];
        overflow |= offsets[r + 1] < offsets[r];
    }
    return (overflow || offsets[world] > cap) ? II2_ECAPACITY : II2_OK;
}

// The exchange itself, in elements of `esz` bytes (4: doc ids, 1: the arrays of an encoded segment).  ctx->mu is held.
static int gatherv_core(ii2_ctx *ctx, const void *d_local_v, uint64_t n_local, size_t esz, void *d_out_v, uint64_t cap,
                        uint64_t *counts_host, const char *who) {
    const uint8_t *d_local = (const uint8_t *)d_local_v;
    uint8_t *d_out = (uint8_t *)d_out_v;
    const int world = ctx->comm ? ctx->world : 1, rank = ctx->comm ? ctx->rank : 0;
    if (world > (int)II2_MAX_RANKS) { ctx->err = "world size above II2_MAX_RANKS"; return II2_EINVAL; }
    hipStream_t st = ctx->stream;
    auto say = [&](const char *m) { ctx->err = std::string(who) + ": " + m; };
    if (world == 1) {
        counts_host[0] = n_local;
        if (n_local > cap) { say("output capacity too small"); return II2_ECAPACITY; }
        if (n_local && !d_out) { say("output buffer is NULL"); return II2_EINVAL; }
        if (n_local && d_out != d_local &&
            hipMemcpyAsync(d_out, d_local, n_local * esz, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            say("copy failed");
            return II2_EHIP;
        }
        if (hipStreamSynchronize(st) != hipSuccess) { say("sync failed"); return II2_EHIP; }
        return II2_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    // 1. every rank learns every rank's {count, capacity}: the fit decision below is then the SAME on all ranks, so
    // no rank can bail out while its peers wait for its data.  A NULL output buffer counts as capacity 0.
    uint64_t *h_x = ctx->h_mail + II2_MAIL_COMM, *d_x = ctx->d_mail + II2_MAIL_COMM;      // [0..1] mine, [2 .. 2 + 2 world) all
    h_x[0] = n_local;
    h_x[1] = d_out ? cap : 0;
    if (hipMemcpyAsync(d_x, h_x, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) {
        say("count upload failed");
        return II2_EHIP;
    }
    NCCL_TRY(ctx, ncclAllGather(d_x, d_x + 2, 2, ncclUint64, comm, st));
    if (hipMemcpyAsync(h_x + 2, d_x + 2, (size_t)world * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        say("count download failed");
        return II2_EHIP;
    }
    std::vector<uint64_t> off(world + 1, 0);
    uint64_t min_cap = ~0ull;
    for (int r = 0; r < world; r++) {
        counts_host[r] = h_x[2 + 2 * r];
        min_cap = std::min(min_cap, h_x[2 + 2 * r + 1]);
    }
    if (ii2_gatherv_offsets(counts_host, world, min_cap, off.data()) != II2_OK) {
        say("the concatenation does not fit the smallest output buffer of the ranks (no rank exchanged anything)");
        return II2_ECAPACITY;
    }
    // the local contribution may sit in d_out only at its own slot (an in-place gather); anywhere else inside d_out it
    // would be overwritten by incoming data while it is being sent
    if (n_local && d_local != d_out + off[rank] * esz && d_local + n_local * esz > d_out && d_local < d_out + off[world] * esz) {
        say("d_local overlaps d_out outside its own slot");
        return II2_EINVAL;      // (local decision: the caller's bug on this rank; peers time out in RCCL — documented)
    }
    // 2. payload: one send + one recv per peer, grouped so they all progress together; the group is always closed.
    // Bytes travel as ncclUint8 (counts are bytes then); doc ids as ncclUint32.
    const ncclDataType_t dt = esz == 4 ? ncclUint32 : ncclUint8;
    const uint64_t unit = esz == 4 ? 1 : esz;           // elements -> items of dt
    ncclResult_t gr = ncclGroupStart();
    for (int r = 0; r < world && gr == ncclSuccess; r++) {
        if (r == rank) continue;
        if (n_local) gr = ncclSend(d_local, n_local * unit, dt, r, comm, st);
        if (gr == ncclSuccess && counts_host[r]) gr = ncclRecv(d_out + off[r] * esz, counts_host[r] * unit, dt, r, comm, st);
    }
    const ncclResult_t ge = ncclGroupEnd();
    if (gr != ncclSuccess || ge != ncclSuccess) {
        ctx->err = std::string(who) + ": grouped send/recv: " + ncclGetErrorString(gr != ncclSuccess ? gr : ge);
        return II2_ECOMM;
    }
    if (n_local && d_out + off[rank] * esz != d_local &&
        hipMemcpyAsync(d_out + off[rank] * esz, d_local, n_local * esz, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        say("local copy failed");
        return II2_EHIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) { say("sync failed"); return II2_EHIP; }
    ctx->comm_syncs++;
    return II2_OK;
}

int ii2_allgatherv(ii2_ctx *ctx, const uint32_t *d_local, uint64_t n_local, uint32_t *d_out, uint64_t cap,
                   uint64_t *counts_host) {
    if (!ctx || !counts_host || (n_local && !d_local)) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    return gatherv_core(ctx, d_local, n_local, sizeof(uint32_t), d_out, cap, counts_host, "ii2_allgatherv");
}

int ii2_allgatherv_bytes(ii2_ctx *ctx, const void *d_local, uint64_t n_bytes, void *d_out, uint64_t cap_bytes,
                         uint64_t *counts_host) {
    if (!ctx || !counts_host || (n_bytes && !d_local)) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    return gatherv_core(ctx, d_local, n_bytes, 1, d_out, cap_bytes, counts_host, "ii2_allgatherv_bytes");
}

// Host arithmetic of the segment exchange (no GPU needed): where every rank's lists, blocks and payload bytes land in
// the concatenated segment.  shape[3 r .. 3 r + 2] = {n_lists, n_blocks, n_bytes} of rank r; the three outputs have
// world + 1 entries each.
int ii2_seg_gather_plan(const uint64_t *shape, int world, uint64_t *list_off, uint64_t *block_off, uint64_t *byte_off) {
    if (!shape || !list_off || !block_off || !byte_off || world < 1 || world > (int)II2_MAX_RANKS) return II2_EINVAL;
    list_off[0] = block_off[0] = byte_off[0] = 0;
    for (int r = 0; r < world; r++) {
        list_off[r + 1] = list_off[r] + shape[3 * r];
        block_off[r + 1] = block_off[r] + shape[3 * r + 1];
        byte_off[r + 1] = byte_off[r] + shape[3 * r + 2];
    }
    // the DV1 limits of one segment (include/ii2.h): block numbers and byte offsets are 32 bits
    if (list_off[world] >= (1ull << 31) || block_off[world] >= (1ull << 31) || byte_off[world] >= 0xFFFFFFF0ull) return II2_ERANGE;
    return II2_OK;
}

// Several arrays, every rank's byte counts already known on every rank (from the shapes): ONE grouped exchange - a send and a
// receive per (peer, array), all peers together, each peer's copies on its own xGMI link - and NO host wait: the copies are
// ordered on the context's stream like any kernel.  off[a] has world + 1 byte offsets into out[a].
struct GatherPart { const uint8_t *local; uint8_t *out; const uint64_t *off; };
static int gatherv_known(ii2_ctx *ctx, int n_parts, const GatherPart *parts, const char *who) {
    const int world = ctx->comm ? ctx->world : 1, rank = ctx->comm ? ctx->rank : 0;
    hipStream_t st = ctx->stream;
    if (world > 1) {
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        ncclResult_t gr = ncclGroupStart();
        for (int r = 0; r < world && gr == ncclSuccess; r++) {
            if (r == rank) continue;
            for (int a = 0; a < n_parts && gr == ncclSuccess; a++) {
                const uint64_t mine = parts[a].off[rank + 1] - parts[a].off[rank], theirs = parts[a].off[r + 1] - parts[a].off[r];
                if (mine) gr = ncclSend(parts[a].local, mine, ncclUint8, r, comm, st);
                if (gr == ncclSuccess && theirs) gr = ncclRecv(parts[a].out + parts[a].off[r], theirs, ncclUint8, r, comm, st);
            }
        }
        const ncclResult_t ge = ncclGroupEnd();             // (the group is always closed)
        if (gr != ncclSuccess || ge != ncclSuccess) {
            ctx->err = std::string(who) + ": grouped send/recv: " + ncclGetErrorString(gr != ncclSuccess ? gr : ge);
            return II2_ECOMM;
        }
    }
    for (int a = 0; a < n_parts; a++) {
        const uint64_t mine = parts[a].off[rank + 1] - parts[a].off[rank];
        if (mine && hipMemcpyAsync(parts[a].out + parts[a].off[rank], parts[a].local, mine, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            ctx->err = std::string(who) + ": local copy failed";
            return II2_EHIP;
        }
    }
    return II2_OK;
}

int ii2_seg_allgather(ii2_ctx *ctx, const ii2_seg *local, ii2_seg **out) {
    if (!ctx || !local || !out || local->device != ctx->device) return II2_EINVAL;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    // a view made by ii2_seg_select* numbers its blocks inside its source's store (blk_off[0] != 0, the store's n_blocks and
    // n_bytes): it is not a self-contained segment and cannot travel as one
    if (local->is_view) { ctx->err = "ii2_seg_allgather: `local` is a view of another segment (ii2_seg_select*); only whole segments travel"; return II2_EINVAL; }
    const int world = ctx->comm ? ctx->world : 1;
    hipStream_t st = ctx->stream;
    // 1. shapes: {lists, blocks, payload bytes, postings} of every rank - the only host round trip before the data moves
    uint64_t *h_x = ctx->h_mail + II2_MAIL_COMM, *d_x = ctx->d_mail + II2_MAIL_COMM;      // [0..3] mine, [4 .. 4 + 4 world) all
    static_assert(II2_MAIL_COMM + 4 + 4 * II2_MAX_RANKS <= II2_MAIL_WORDS, "mailbox too small for the segment shapes");
    h_x[0] = local->n_lists; h_x[1] = local->n_blocks; h_x[2] = local->n_bytes; h_x[3] = local->n_postings;
    if (world == 1) std::memcpy(h_x + 4, h_x, 4 * sizeof(uint64_t));
    else {
        if (hipMemcpyAsync(d_x, h_x, 4 * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) { ctx->err = "ii2_seg_allgather: shape upload failed"; return II2_EHIP; }
        NCCL_TRY(ctx, ncclAllGather(d_x, d_x + 4, 4, ncclUint64, (ncclComm_t)ctx->comm, st));
        if (hipMemcpyAsync(h_x + 4, d_x + 4, (size_t)world * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { ctx->err = "ii2_seg_allgather: shape download failed"; return II2_EHIP; }
        ctx->comm_syncs++;
    }
    std::vector<uint64_t> shape(3 * world), lo(world + 1), bo(world + 1), qo(world + 1);
    uint64_t n_post = 0;
    for (int r = 0; r < world; r++) {
        shape[3 * r] = h_x[4 + 4 * r]; shape[3 * r + 1] = h_x[4 + 4 * r + 1]; shape[3 * r + 2] = h_x[4 + 4 * r + 2];
        n_post += h_x[4 + 4 * r + 3];
    }
    // (identical on every rank: either all ranks go on or all of them stop here)
    if (int rc = ii2_seg_gather_plan(shape.data(), world, lo.data(), bo.data(), qo.data())) {
        ctx->err = "ii2_seg_allgather: the concatenated segment exceeds the DV1 limits (2^31 lists / blocks, 4 GiB of payload)";
        return rc;
    }
    // 2. the concatenated segment's arrays - the DV1 arrays and the three derived ones (posting counts, last docs, block owners:
    // they travel too, so that the receiver does not decode every list's last block to get them back) - in ONE grouped
    // exchange: six sends and six receives per peer, no host wait in between.  Every part keeps its own numbering until step 3.
    ii2_seg *seg = nullptr;
    if (int rc = ii2_seg_alloc_internal(ctx, lo[world], n_post, bo[world], qo[world], &seg, true)) return rc;
    std::vector<uint64_t> lo4(world + 1), bo8(world + 1), bo4(world + 1);
    for (int r = 0; r <= world; r++) { lo4[r] = lo[r] * sizeof(uint32_t); bo8[r] = bo[r] * sizeof(ii2_skip); bo4[r] = bo[r] * sizeof(uint32_t); }
    const GatherPart parts[6] = {
        {(const uint8_t *)local->d_blk_off, (uint8_t *)seg->d_blk_off, lo4.data()},
        {(const uint8_t *)local->d_skip, (uint8_t *)seg->d_skip, bo8.data()},
        {(const uint8_t *)local->d_payload, (uint8_t *)seg->d_payload, qo.data()},
        {(const uint8_t *)local->d_cnt, (uint8_t *)seg->d_cnt, lo4.data()},
        {(const uint8_t *)local->d_last_doc, (uint8_t *)seg->d_last_doc, lo4.data()},
        {(const uint8_t *)local->d_blk_list, (uint8_t *)seg->d_blk_list, bo4.data()},
    };
    int rc = gatherv_known(ctx, 6, parts, "ii2_seg_allgather");
    // 3. block numbers, byte offsets and block owners of rank r's part move up by what the ranks before it hold; closing entries
    if (!rc) rc = ii2_seg_rebase_internal(ctx, seg, world, lo.data(), bo.data(), qo.data());
    if (rc) { (void)hipStreamSynchronize(st); ii2_seg_free(seg); return rc; }
    ctx->comm_syncs++;
    *out = seg;
    return II2_OK;
}

int ii2_seg_concat(ii2_ctx *ctx, uint32_t n, const ii2_seg *const *segs, ii2_seg **out) {
    if (!ctx || !segs || !out || n == 0 || n > II2_MAX_RANKS) return II2_EINVAL;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    std::vector<uint64_t> shape(3 * (size_t)n), lo(n + 1), bo(n + 1), qo(n + 1);
    uint64_t n_post = 0;
    for (uint32_t r = 0; r < n; r++) {
        if (!segs[r] || segs[r]->device != ctx->device) { ctx->err = "ii2_seg_concat: segment is NULL or lives on another device"; return II2_EINVAL; }
        if (segs[r]->is_view) { ctx->err = "ii2_seg_concat: a view of another segment (ii2_seg_select*) is not a whole segment"; return II2_EINVAL; }
        shape[3 * r] = segs[r]->n_lists; shape[3 * r + 1] = segs[r]->n_blocks; shape[3 * r + 2] = segs[r]->n_bytes;
        n_post += segs[r]->n_postings;
    }
    if (int rc = ii2_seg_gather_plan(shape.data(), (int)n, lo.data(), bo.data(), qo.data())) {
        ctx->err = "ii2_seg_concat: the concatenated segment exceeds the DV1 limits (2^31 lists / blocks, 4 GiB of payload)";
        return rc;
    }
    ii2_seg *seg = nullptr;
    if (int rc = ii2_seg_alloc_internal(ctx, lo[n], n_post, bo[n], qo[n], &seg, true)) return rc;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    auto put = [&](void *dst, const void *src, uint64_t bytes) { if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st); };
    for (uint32_t r = 0; r < n; r++) {
        const ii2_seg *s = segs[r];
        put(seg->d_blk_off + lo[r], s->d_blk_off, s->n_lists * sizeof(uint32_t));
        put(seg->d_skip + bo[r], s->d_skip, s->n_blocks * sizeof(ii2_skip));
        put(seg->d_payload + qo[r], s->d_payload, s->n_bytes);
        put(seg->d_cnt + lo[r], s->d_cnt, s->n_lists * sizeof(uint32_t));
        put(seg->d_last_doc + lo[r], s->d_last_doc, s->n_lists * sizeof(uint32_t));
        put(seg->d_blk_list + bo[r], s->d_blk_list, s->n_blocks * sizeof(uint32_t));
    }
    int rc = II2_OK;
    if (e != hipSuccess) { ctx->err = std::string("ii2_seg_concat: copy failed: ") + hipGetErrorString(e); rc = II2_EHIP; }
    if (!rc) rc = ii2_seg_rebase_internal(ctx, seg, (int)n, lo.data(), bo.data(), qo.data());
    if (rc) { (void)hipStreamSynchronize(st); ii2_seg_free(seg); return rc; }
    ctx->comm_syncs++;
    *out = seg;
    return II2_OK;
}

}  // extern "C"
