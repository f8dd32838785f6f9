// comm.cpp — rank-order concatenation of per-GPU result postings over RCCL / xGMI.
//
// Replaces InvertedIndex.Read's shard-order concatenation (reference inverted_index.go:330-339)
// when terms (merge) or doc ranges (one conjunctive query) are sharded over the GPUs of a
// node.  RCCL has no all-gatherv: counts travel with one ncclAllGather of u64, the payload
// with one grouped ncclSend/ncclRecv per peer — the 8 GPUs are fully connected, so every
// peer's copy rides its own xGMI link instead of a ring that one link would bound.
#include <rccl/rccl.h>

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

int ii2_allgatherv(ii2_ctx *ctx, const uint32_t *d_local, uint64_t n_local, uint32_t *d_out, uint64_t cap,
                   uint64_t *counts_host) {
    if (!ctx || !counts_host || (n_local && !d_local)) return II2_EINVAL;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return II2_EHIP; }
    const int world = ctx->comm ? ctx->world : 1, rank = ctx->comm ? ctx->rank : 0;
    if (world > 64) { ctx->err = "world size above 64"; return II2_EINVAL; }
    hipStream_t st = ctx->stream;
    if (world == 1) {
        counts_host[0] = n_local;
        if (n_local > cap) { ctx->err = "ii2_allgatherv: output capacity too small"; return II2_ECAPACITY; }
        if (n_local && d_out != d_local &&
            hipMemcpyAsync(d_out, d_local, n_local * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            ctx->err = "ii2_allgatherv: copy failed";
            return II2_EHIP;
        }
        if (hipStreamSynchronize(st) != hipSuccess) { ctx->err = "ii2_allgatherv: sync failed"; return II2_EHIP; }
        return II2_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    // 1. counts: d_mail[0] = mine, d_mail[1..world] = everyone's
    ctx->h_mail[0] = n_local;
    if (hipMemcpyAsync(ctx->d_mail, ctx->h_mail, sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) {
        ctx->err = "ii2_allgatherv: count upload failed";
        return II2_EHIP;
    }
    NCCL_TRY(ctx, ncclAllGather(ctx->d_mail, ctx->d_mail + 1, 1, ncclUint64, comm, st));
    if (hipMemcpyAsync(ctx->h_mail + 1, ctx->d_mail + 1, (size_t)world * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        ctx->err = "ii2_allgatherv: count download failed";
        return II2_EHIP;
    }
    std::vector<uint64_t> off(world + 1, 0);
    for (int r = 0; r < world; r++) {
        counts_host[r] = ctx->h_mail[1 + r];
        off[r + 1] = off[r] + counts_host[r];
    }
    if (off[world] > cap) { ctx->err = "ii2_allgatherv: output capacity too small"; return II2_ECAPACITY; }
    // 2. payload: one send + one recv per peer, grouped so they all progress together
    NCCL_TRY(ctx, ncclGroupStart());
    for (int r = 0; r < world; r++) {
        if (r == rank) continue;
        if (n_local) NCCL_TRY(ctx, ncclSend(d_local, n_local, ncclUint32, r, comm, st));
        if (counts_host[r]) NCCL_TRY(ctx, ncclRecv(d_out + off[r], counts_host[r], ncclUint32, r, comm, st));
    }
    NCCL_TRY(ctx, ncclGroupEnd());
    if (n_local && d_out + off[rank] != d_local &&
        hipMemcpyAsync(d_out + off[rank], d_local, n_local * sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        ctx->err = "ii2_allgatherv: local copy failed";
        return II2_EHIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) { ctx->err = "ii2_allgatherv: sync failed"; return II2_EHIP; }
    return II2_OK;
}

}  // extern "C"
