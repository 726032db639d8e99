// Microbenchmark 12 (round 6): WHERE to look for a fast address for a composed arena.  Eight 1 GiB chunks (as created), the
// sampled probe of the whole 8 GiB against a hipMalloc'ed pool at (a) 16 positions 10 GiB apart inside ONE 170 GiB range,
// (b) the first position of 12 SEPARATE reservations of 10 GiB (all kept alive), (c) positions 0 / 1 / 2 / 3 GiB of a third
// range.  Every address is mapped once.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../croaring_amd/csrc/rhip_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_fill(u64* p, u64 n) { for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull; }
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const u64 G = 1ull << 30, need = 8 * G, poolb = 8 * G;
    uint8_t* A; CK(hipMalloc(&A, poolb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, poolb / 8);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h[8];
    for (int k = 0; k < 8; ++k) CK(hipMemCreate(&h[k], G, &prop, 0));
    auto at_rate = [&](uint8_t* at) {
        for (int k = 0; k < 8; ++k) CK(hipMemMap(at + (u64)k * G, G, 0, h[k], 0));
        CK(hipMemSetAccess(at, need, &acc, 1));
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, 0, A, poolb / 8192ull, at, need / 8192ull, 8ull);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        for (int k = 0; k < 8; ++k) CK(hipMemUnmap(at + (u64)k * G, G));
        return (double)(need / 8192ull / 8) * 24576.0 / best / 1e6;
    };
    auto up = [&](void* R) { return (uint8_t*)(((uintptr_t)R + G - 1) / G * G) + (2ull << 20); };
    void* R1; CK(hipMemAddressReserve(&R1, 172 * G, 0, nullptr, 0));
    printf("pool %p\n(a) one range %p, positions 10 GiB apart:", (void*)A, R1);
    for (int p = 0; p < 16; ++p) printf(" %.0f", at_rate(up(R1) + (u64)p * 10 * G));
    printf("\n(b) first position of separate 10 GiB reservations:");
    for (int p = 0; p < 12; ++p) { void* R; CK(hipMemAddressReserve(&R, 10 * G, 0, nullptr, 0)); printf(" %.0f", at_rate(up(R))); }
    void* R3; CK(hipMemAddressReserve(&R3, 14 * G, 0, nullptr, 0));
    printf("\n(c) range %p, positions 0 / 1 / 2 / 3 GiB: ", R3);
    // (overlapping positions would map an address twice: four ranges instead)
    for (int p = 0; p < 4; ++p) { void* R; CK(hipMemAddressReserve(&R, 14 * G, 0, nullptr, 0)); printf(" %.0f", at_rate(up(R) + (u64)p * G)); }
    printf("\n");
    return 0;
}
