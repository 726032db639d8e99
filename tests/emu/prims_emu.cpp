// Host stand-ins for the two rocPRIM-backed directory primitives (croaring_amd/csrc/rhip_prims.h)
// used when the kernels run under hipemu.  Test infrastructure only.
#include <algorithm>
#include <numeric>
#include <vector>

#include "rhip_prims.h"

hipError_t prim_exscan_u32_u64(void* tmp, size_t& tmp_bytes, const uint32_t* in, unsigned long long* out, size_t n,
                               hipStream_t) {
    if (!tmp) {
        tmp_bytes = 256;
        return hipSuccess;
    }
    unsigned long long acc = 0;
    for (size_t i = 0; i <= n; ++i) {
        const unsigned long long v = (i < n) ? in[i] : 0;  // in[n] is readable but ignored
        out[i] = acc;
        acc += v;
    }
    return hipSuccess;
}

hipError_t prim_sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                                   const uint32_t* vin, uint32_t* vout, size_t n, int end_bit, hipStream_t) {
    if (!tmp) {
        tmp_bytes = 256;
        return hipSuccess;
    }
    const unsigned long long mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (kin[a] & mask) < (kin[b] & mask); });
    for (size_t i = 0; i < n; ++i) {
        kout[i] = kin[idx[i]];
        vout[i] = vin[idx[i]];
    }
    return hipSuccess;
}

hipError_t prim_sort_pairs_u64_u64(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                                   const unsigned long long* vin, unsigned long long* vout, size_t n, int end_bit,
                                   hipStream_t) {
    if (!tmp) {
        tmp_bytes = 256;
        return hipSuccess;
    }
    const unsigned long long mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (kin[a] & mask) < (kin[b] & mask); });
    for (size_t i = 0; i < n; ++i) {
        kout[i] = kin[idx[i]];
        vout[i] = vin[idx[i]];
    }
    return hipSuccess;
}
