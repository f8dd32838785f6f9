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

int ii2_allgatherv(ii2_ctx *ctx, const uint32_t *d_local, uint64_t n_local, uint32_t *d_out, uint64_t cap,
                   uint64_t *counts_host) {
    if (!ctx || !counts_host || (n_local && !d_local)) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    const int world = ctx->comm ? ctx->world : 1, rank = ctx->comm ? ctx->rank : 0;
    if (world > (int)II2_MAX_RANKS) { ctx->err = "world size above II2_MAX_RANKS"; return II2_EINVAL; }
    hipStream_t st = ctx->stream;
    if (world == 1) {
        counts_host[0] = n_local;
        if (n_local > cap) { ctx->err = "ii2_allgatherv: output capacity too small"; return II2_ECAPACITY; }
        if (n_local && !d_out) { ctx->err = "ii2_allgatherv: output buffer is NULL"; return II2_EINVAL; }
        if (n_local && d_out != d_local &&
            hipMemcpyAsync(d_out, d_local, n_local * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            ctx->err = "ii2_allgatherv: copy failed";
            return II2_EHIP;
        }
        if (hipStreamSynchronize(st) != hipSuccess) { ctx->err = "ii2_allgatherv: sync failed"; return II2_EHIP; }
        return II2_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    // 1. every rank learns every rank's {count, capacity}: the fit decision below is then the SAME on all ranks, so
    // no rank can bail out while its peers wait for its data.  A NULL output buffer counts as capacity 0.
    uint64_t *h_x = ctx->h_mail + II2_MAIL_COMM, *d_x = ctx->d_mail + II2_MAIL_COMM;      // [0..1] mine, [2 .. 2 + 2 world) all
    h_x[0] = n_local;
    h_x[1] = d_out ? cap : 0;
    if (hipMemcpyAsync(d_x, h_x, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) {
        ctx->err = "ii2_allgatherv: count upload failed";
        return II2_EHIP;
    }
    NCCL_TRY(ctx, ncclAllGather(d_x, d_x + 2, 2, ncclUint64, comm, st));
    if (hipMemcpyAsync(h_x + 2, d_x + 2, (size_t)world * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        ctx->err = "ii2_allgatherv: count download failed";
        return II2_EHIP;
    }
    std::vector<uint64_t> off(world + 1, 0);
    uint64_t min_cap = ~0ull;
    for (int r = 0; r < world; r++) {
        counts_host[r] = h_x[2 + 2 * r];
        min_cap = std::min(min_cap, h_x[2 + 2 * r + 1]);
    }
    if (ii2_gatherv_offsets(counts_host, world, min_cap, off.data()) != II2_OK) {
        ctx->err = "ii2_allgatherv: the concatenation does not fit the smallest output buffer of the ranks (no rank exchanged anything)";
        return II2_ECAPACITY;
    }
    // the local contribution may sit in d_out only at its own slot (an in-place gather); anywhere else inside d_out it
    // would be overwritten by incoming data while it is being sent
    if (n_local && d_local != d_out + off[rank] && d_local + n_local > d_out && d_local < d_out + off[world]) {
        ctx->err = "ii2_allgatherv: d_local overlaps d_out outside its own slot";
        return II2_EINVAL;      // (local decision: the caller's bug on this rank; peers time out in RCCL — documented)
    }
    // 2. payload: one send + one recv per peer, grouped so they all progress together; the group is always closed
    ncclResult_t gr = ncclGroupStart();
    for (int r = 0; r < world && gr == ncclSuccess; r++) {
        if (r == rank) continue;
        if (n_local) gr = ncclSend(d_local, n_local, ncclUint32, r, comm, st);
        if (gr == ncclSuccess && counts_host[r]) gr = ncclRecv(d_out + off[r], counts_host[r], ncclUint32, r, comm, st);
    }
    const ncclResult_t ge = ncclGroupEnd();
    if (gr != ncclSuccess || ge != ncclSuccess) {
        ctx->err = std::string("ii2_allgatherv: grouped send/recv: ") + ncclGetErrorString(gr != ncclSuccess ? gr : ge);
        return II2_ECOMM;
    }
    if (n_local && d_out + off[rank] != d_local &&
        hipMemcpyAsync(d_out + off[rank], d_local, n_local * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        ctx->err = "ii2_allgatherv: local copy failed";
        return II2_EHIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) { ctx->err = "ii2_allgatherv: sync failed"; return II2_EHIP; }
    return II2_OK;
}

}  // extern "C"
