// Device-wide metadata primitives (prefix sums and key sorts over the container DIRECTORY,
// never over payload).  Implemented with rocPRIM in rhip_prims.hip so the hot-path
// translation unit stays free of the rocPRIM headers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// out[i] = sum_{j<i} in[j] for i in [0, n]; `in` must hold n+1 readable elements (in[n] is
// ignored), out must hold n+1, so out[n] is the grand total.
hipError_t prim_exscan_u32_u64(void* tmp, size_t& tmp_bytes, const uint32_t* in, unsigned long long* out, size_t n,
                               hipStream_t s);
// stable LSD radix sort of (key, value) pairs on key bits [0, end_bit)
hipError_t prim_sort_pairs_u64_u32(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                                   const uint32_t* vin, uint32_t* vout, size_t n, int end_bit, hipStream_t s);
hipError_t prim_sort_pairs_u64_u64(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                                   const unsigned long long* vin, unsigned long long* vout, size_t n, int end_bit,
                                   hipStream_t s);
