// scan.hip — device-wide exclusive prefix sums used by the planners (offset tables only;
// the posting kernels are hand-written).  hipcub::DeviceScan with caller-provided temp.
#include <hipcub/hipcub.hpp>

#include "internal.h"

namespace ii2 {

struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};

size_t scan_temp_bytes(size_t n) {
    size_t a = 0, b = 0, c = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, (hipStream_t)0);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)n, (hipStream_t)0);
    hipcub::TransformInputIterator<uint64_t, U32ToU64, const uint32_t *> it((const uint32_t *)nullptr, U32ToU64());
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, it, (uint64_t *)nullptr, (int)n, (hipStream_t)0);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return (m + 255) & ~(size_t)255;
}

hipError_t scan_excl_u32(void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, (int)n, s);
}

hipError_t scan_excl_u32_to_u64(void *tmp, size_t tmp_bytes, const uint32_t *in, uint64_t *out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipcub::TransformInputIterator<uint64_t, U32ToU64, const uint32_t *> it(in, U32ToU64());
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, it, out, (int)n, s);
}

hipError_t scan_excl_u64(void *tmp, size_t tmp_bytes, const uint64_t *in, uint64_t *out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, (int)n, s);
}

}  // namespace ii2
