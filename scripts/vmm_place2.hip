// Microbenchmark 7 (round 6): is the placement mode of a C2 result arena a function of its VIRTUAL address or of its
// PHYSICAL pages?  hipMemMap maps the same physical chunks at any address:
//   sweep V: the SAME eight 1 GiB chunks mapped at R + v GiB, v = 0, 1, 2 ... (same pages, other addresses);
//   sweep P: eight OTHER chunks each time at the SAME address R (same addresses, other pages).
// The probe is k_place_probe (k_bb's access pattern) over the whole 8 GiB against one hipMalloc'ed operand pool.
// argv: number of chunks (default 40), VA range in GiB (default 96), step of sweep V in MiB (default 1024).  Not product code.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../croaring_amd/csrc/rhip_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_fill(u64* p, u64 n, u64 salt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = (i + salt) * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 40;
    const u64 RANGE = (argc > 2 ? (u64)atoll(argv[2]) : 96ull) << 30;
    const u64 STEP = (argc > 3 ? (u64)atoll(argv[3]) : 1024ull) << 20;
    const u64 CH = 1ull << 30, need = 8ull << 30, poolb = 8ull << 30;
    uint8_t* A;
    CK(hipMalloc(&A, poolb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, poolb / 8, 0ull);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto rate = [&](uint8_t* out) {
        const u64 n_slots = need / 8192ull;
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, 0, A, poolb / 8192ull, out, n_slots, 1ull);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return (double)n_slots * 24576.0 / best / 1e6;
    };
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h((size_t)N);
    for (int k = 0; k < N; ++k) CK(hipMemCreate(&h[(size_t)k], CH, &prop, 0));
    void* R = nullptr;
    CK(hipMemAddressReserve(&R, RANGE, 1ull << 30, nullptr, 0));
    printf("pool at %p (%.3f GiB), range at %p (%.3f GiB): range - pool = %.3f GiB\n", (void*)A, (double)(uintptr_t)A / (double)(1ull << 30), R,
           (double)(uintptr_t)R / (double)(1ull << 30), ((double)(uintptr_t)R - (double)(uintptr_t)A) / (double)(1ull << 30));
    auto with = [&](int p0, u64 voff) {
        uint8_t* at = (uint8_t*)R + voff;
        for (int j = 0; j < 8; ++j) CK(hipMemMap(at + (u64)j * CH, CH, 0, h[(size_t)(p0 + j)], 0));
        CK(hipMemSetAccess(at, need, &acc, 1));
        const double g = rate(at);
        CK(hipMemUnmap(at, need));
        return g;
    };
    printf("sweep V (chunks 0-7 at range + v):");
    for (u64 v = 0; v + need <= RANGE; v += STEP) printf(" %.2f:%.0f", (double)v / (double)(1ull << 30), with(0, v));
    printf("\nsweep V again, chunks 16-23:");
    for (u64 v = 0; v + need <= RANGE; v += STEP) printf(" %.2f:%.0f", (double)v / (double)(1ull << 30), with(16, v));
    printf("\nsweep P (chunks p..p+7 at range + 0):");
    for (int p = 0; p + 8 <= N; p += 2) printf(" %d:%.0f", p, with(p, 0));
    printf("\nsweep P (chunks p..p+7 at range + 20 GiB):");
    for (int p = 0; p + 8 <= N; p += 2) printf(" %d:%.0f", p, with(p, 20ull << 30));
    printf("\n");
    return 0;
}
