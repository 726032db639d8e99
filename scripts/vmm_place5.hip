// Microbenchmark 11 (round 6): is the placement mode a property of the PAIR (pool region j, arena chunk q) that meet in
// lockstep?  k_bb reads pool item i and writes result slot i: arena position j (GiB j of the arena) is written while pool GiB j
// is read.  Here the pool is hipMalloc'ed (as in the product), NQ physical chunks of 1 GiB are created, and every chunk q is
// probed AT every arena position j -- mapped (at never-used addresses) so that slots of position j fall into it -- against
// the pool: M[j][q].  Then arenas are COMPOSED: for every position the best chunk still unused (greedy), and the worst, and
// the full 8 GiB pass is timed on both and on a plain hipMalloc arena.  argv: NQ (default 16).  Not product code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../croaring_amd/csrc/rhip_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_fill(u64* p, u64 n, u64 salt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = (i + salt) * 0x9E3779B97F4A7C15ull;
}
// k_place_probe's access pattern over the slots [s0, s1) only
__global__ __launch_bounds__(256) void k_probe_range(const uint8_t* __restrict__ arenaA, u64 a_items, uint8_t* __restrict__ out, u64 s0, u64 s1) {
    const uint32_t lane = threadIdx.x & 63u;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 i = s0 + (((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6); i < s1; i += nwaves) {
        const u32x4* __restrict__ pa = (const u32x4*)(arenaA + (i % a_items) * 8192ull);
        const u32x4* __restrict__ pb = (const u32x4*)(arenaA + ((i * 97ull + 4096ull * 33ull) % a_items) * 8192ull);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = __builtin_nontemporal_load(pa + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 8; ++k) vb[k] = __builtin_nontemporal_load(pb + k * 64 + lane);
        u32x4* __restrict__ po = (u32x4*)(out + i * 8192ull);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(va[k] | vb[k], po + k * 64 + lane);
    }
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int NQ = argc > 1 ? atoi(argv[1]) : 16;
    const u64 G = 1ull << 30, need = 8 * G, poolb = 8 * G, SL = G / 8192;  // slots per GiB
    uint8_t* A;
    CK(hipMalloc(&A, poolb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, poolb / 8, 0ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h((size_t)NQ);
    for (int q = 0; q < NQ; ++q) CK(hipMemCreate(&h[(size_t)q], G, &prop, 0));
    // address space: every (j, q) probe gets a GiB of its own, plus three composed arenas
    const u64 va_len = ((u64)8 * NQ + 32) * G + 2 * G;
    void* R = nullptr;
    CK(hipMemAddressReserve(&R, va_len, 0, nullptr, 0));
    uint8_t* base0 = (uint8_t*)(((uintptr_t)R + G - 1) / G * G) + (2ull << 20);
    printf("pool %p, range %p, %d chunks\n", (void*)A, R, NQ);
    auto timed = [&](uint8_t* out, u64 s0, u64 s1) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe_range, dim3(8192), dim3(256), 0, 0, A, poolb / 8192ull, out, s0, s1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return (double)(s1 - s0) * 24576.0 / best / 1e6;
    };
    std::vector<std::vector<double>> M(8, std::vector<double>((size_t)NQ));
    u64 slot = 0;  // next free GiB of the range
    for (int j = 0; j < 8; ++j) {
        printf("position %d:", j);
        for (int q = 0; q < NQ; ++q) {
            uint8_t* at = base0 + (slot++) * G;
            CK(hipMemMap(at, G, 0, h[(size_t)q], 0));
            CK(hipMemSetAccess(at, G, &acc, 1));
            M[(size_t)j][(size_t)q] = timed(at - (u64)j * G, (u64)j * SL, (u64)(j + 1) * SL);
            CK(hipMemUnmap(at, G));
            printf(" %.0f", M[(size_t)j][(size_t)q]);
        }
        printf("\n");
    }
    auto compose = [&](bool best) {
        std::vector<int> used((size_t)NQ, 0), pick(8);
        double hm = 0;
        for (int j = 0; j < 8; ++j) {
            int b = -1;
            for (int q = 0; q < NQ; ++q)
                if (!used[(size_t)q] && (b < 0 || (best ? M[(size_t)j][(size_t)q] > M[(size_t)j][(size_t)b] : M[(size_t)j][(size_t)q] < M[(size_t)j][(size_t)b]))) b = q;
            used[(size_t)b] = 1; pick[(size_t)j] = b; hm += 1.0 / M[(size_t)j][(size_t)b];
        }
        uint8_t* at = base0 + slot * G;
        slot += 9;
        for (int j = 0; j < 8; ++j) CK(hipMemMap(at + (u64)j * G, G, 0, h[(size_t)pick[(size_t)j]], 0));
        CK(hipMemSetAccess(at, need, &acc, 1));
        const double full = timed(at, 0, 8 * SL);
        printf("%s composition: chunks", best ? "BEST" : "WORST");
        for (int j = 0; j < 8; ++j) printf(" %d", pick[(size_t)j]);
        printf(" -> full pass %.0f GB/s (harmonic mean of its pair rates %.0f)\n", full, 8.0 / hm);
        for (int j = 0; j < 8; ++j) CK(hipMemUnmap(at + (u64)j * G, G));
    };
    compose(true);
    compose(false);
    compose(true);
    uint8_t* m; CK(hipMalloc(&m, need));
    printf("hipMalloc arena: full pass %.0f GB/s; per position:", timed(m, 0, 8 * SL));
    for (int j = 0; j < 8; ++j) printf(" %.0f", timed(m, (u64)j * SL, (u64)(j + 1) * SL));
    printf("\n");
    return 0;
}
