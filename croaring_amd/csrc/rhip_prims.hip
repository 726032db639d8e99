#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rhip_prims.h"

namespace {
struct widen {
    __host__ __device__ unsigned long long operator()(uint32_t v) const { return (unsigned long long)v; }
};
}  // namespace

hipError_t prim_exscan_u32_u64(void* tmp, size_t& tmp_bytes, const uint32_t* in, unsigned long long* out, size_t n,
                               hipStream_t s) {
    auto it = rocprim::make_transform_iterator(in, widen());
    return rocprim::exclusive_scan(tmp, tmp_bytes, it, out, (unsigned long long)0, n + 1, rocprim::plus<unsigned long long>(), s);
}

hipError_t prim_sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                                   const uint32_t* vin, uint32_t* vout, size_t n, int end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, s);
}
hipError_t prim_sort_pairs_u64_u64(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                                   const unsigned long long* vin, unsigned long long* vout, size_t n, int end_bit,
                                   hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, s);
}
