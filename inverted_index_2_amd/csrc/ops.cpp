// TEMPORARY stubs (replaced by the merge/union implementation).
#include "internal.h"
extern "C" {
int ii2_merge_segments(ii2_ctx *ctx, uint32_t, const ii2_seg *const *, const ii2_tomb *, uint64_t *, uint32_t *, uint64_t, ii2_merge_stats *) { if (ctx) ctx->err = "not implemented"; return II2_EINVAL; }
int ii2_merge_segments_to_seg(ii2_ctx *ctx, uint32_t, const ii2_seg *const *, const ii2_tomb *, ii2_seg **, ii2_merge_stats *) { if (ctx) ctx->err = "not implemented"; return II2_EINVAL; }
int ii2_union(ii2_ctx *ctx, uint32_t, const ii2_seg *const *, const uint64_t *, const ii2_tomb *, uint32_t *, uint64_t, uint64_t *) { if (ctx) ctx->err = "not implemented"; return II2_EINVAL; }
int ii2_merge_host(ii2_ctx *ctx, uint32_t, uint64_t, const uint64_t *, const uint64_t *, const uint32_t *, const uint32_t *, uint64_t, uint64_t *, uint32_t *, uint64_t, ii2_merge_stats *) { if (ctx) ctx->err = "not implemented"; return II2_EINVAL; }
int ii2_intersect_host(ii2_ctx *ctx, uint32_t, const uint64_t *, const uint32_t *, const uint32_t *, uint64_t, uint32_t *, uint64_t, uint64_t *) { if (ctx) ctx->err = "not implemented"; return II2_EINVAL; }
int ii2_union_host(ii2_ctx *ctx, uint32_t, const uint64_t *, const uint32_t *, const uint32_t *, uint64_t, uint32_t *, uint64_t, uint64_t *) { if (ctx) ctx->err = "not implemented"; return II2_EINVAL; }
}
