// Microbenchmark 10 (round 6): do large translation fragments help RANDOM reads?  k_many_l1 gathers ~512-byte members at
// random offsets of a 1.6-16 GB arena (67 % of its wave cycles wait).  Here: every half-wave reads one 512-byte record
// (16 bytes per lane) at a hashed offset of a buffer of S GiB, 64 records per half-wave in a row, 4 independent loads in
// flight per lane; the buffer is (a) hipMalloc'ed, (b) 1 GiB chunks mapped at a 1 GiB-aligned address (the driver may use
// 1 GiB pages), (c) the same chunks mapped 2 MiB past a GiB boundary (2 MiB pages at most).  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ buf, u64 n_rec, u64 per_half, uint32_t* sink, u64 salt) {
    const u64 half = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const uint32_t l = threadIdx.x & 31u;
    u32x4 acc = {0, 0, 0, 0};
    for (u64 r = 0; r < per_half; r += 4) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 rec = mix((half * per_half + r + j) * 0x9E3779B97F4A7C15ull + salt) % n_rec;
            v[j] = __builtin_nontemporal_load((const u32x4*)(buf + rec * 512ull) + l);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const u64 G = 1ull << 30;
    uint32_t* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto run = [&](const uint8_t* buf, u64 bytes) {
        const u64 n_rec = bytes / 512, per_half = 64, halves = 256ull * 8 * 2 * 64;  // 4 Gi bytes read
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather, dim3((unsigned)(halves * 32 / 256)), dim3(256), 0, 0, buf, n_rec, per_half, sink, (u64)r * 977);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return (double)halves * per_half * 512.0 / best / 1e6;
    };
    for (u64 S : {2ull, 16ull, 64ull}) {
        const u64 bytes = S * G;
        uint8_t* m; CK(hipMalloc(&m, bytes)); CK(hipMemset(m, 1, bytes));
        const double a = run(m, bytes);
        CK(hipFree(m));
        std::vector<hipMemGenericAllocationHandle_t> h((size_t)S);
        for (u64 k = 0; k < S; ++k) CK(hipMemCreate(&h[(size_t)k], G, &prop, 0));
        double bc[2];
        for (int w = 0; w < 2; ++w) {  // fresh range per way: an address is mapped once
            void* R = nullptr;
            CK(hipMemAddressReserve(&R, bytes + 2 * G, 0, nullptr, 0));
            uint8_t* at = (uint8_t*)(((uintptr_t)R + G - 1) / G * G) + (w ? (2ull << 20) : 0);
            for (u64 k = 0; k < S; ++k) CK(hipMemMap(at + k * G, G, 0, h[(size_t)k], 0));
            CK(hipMemSetAccess(at, bytes, &acc, 1));
            CK(hipMemset(at, 1, bytes));
            bc[w] = run(at, bytes);
            for (u64 k = 0; k < S; ++k) CK(hipMemUnmap(at + k * G, G));
        }
        for (u64 k = 0; k < S; ++k) CK(hipMemRelease(h[(size_t)k]));
        printf("%llu GiB: hipMalloc %.0f GB/s | 1 GiB chunks at a GiB-aligned address %.0f | 2 MiB past it %.0f\n", (unsigned long long)S, a, bc[0], bc[1]);
    }
    return 0;
}
