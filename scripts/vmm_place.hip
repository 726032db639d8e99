// Microbenchmark 6 (round 6): is the placement mode of a C2 result arena a property of its 1 GiB PHYSICAL chunks, and
// can an arena be COMPOSED of the good ones?  The virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap)
// hands out physical chunks one by one and maps them where the caller likes:
//   1. N chunks of CH GiB are created and mapped back to back; every chunk is probed ALONE against the operand pool
//      (k_place_probe, the bitset kernel's access pattern: 2 reads of the pool per slot written) -> one rate per chunk,
//      and against a SECOND pool elsewhere (is the mode a property of the pair or of the chunk?);
//   2. the sliding window of 8 GiB over the mapped range (what a slab offers) -> one rate per offset;
//   3. the 8 best chunks are mapped into a fresh range, and the 8 worst into another: full-pass rates;
//   4. what the calls cost: create / map / unmap / release, and a hipMalloc of 8 GiB after the release (the scrub).
// argv: N (default 64), CH in MiB (default 1024).  Not product code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#include "../croaring_amd/csrc/rhip_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fill(u64* p, u64 n, u64 salt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = (i + salt) * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64;
    const u64 CH = (argc > 2 ? (u64)atoll(argv[2]) : 1024ull) << 20;
    const u64 need = 8ull << 30;
    const int per = (int)(need / CH);  // chunks per arena
    const u64 poolb = 8ull << 30;
    uint8_t *A, *A2;
    CK(hipMalloc(&A, poolb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, poolb / 8, 0ull);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto rate = [&](const uint8_t* pool, uint8_t* out, u64 bytes, u64 stride) {  // GB/s of the probe over `bytes` of `out`
        const u64 n_slots = bytes / 8192ull, n_items = (n_slots + stride - 1) / stride;
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, 0, pool, poolb / 8192ull, out, n_slots, stride);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return (double)n_items * 24576.0 / best / 1e6;
    };
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu, N %d chunks of %llu MiB, pool at %p\n", gran, N, (unsigned long long)(CH >> 20), (void*)A);
    std::vector<hipMemGenericAllocationHandle_t> h((size_t)N);
    double t = now_ms();
    for (int k = 0; k < N; ++k) CK(hipMemCreate(&h[(size_t)k], CH, &prop, 0));
    const double t_create = now_ms() - t;
    void* va = nullptr;
    t = now_ms();
    CK(hipMemAddressReserve(&va, (u64)N * CH, 1ull << 30, nullptr, 0));
    for (int k = 0; k < N; ++k) CK(hipMemMap((uint8_t*)va + (u64)k * CH, CH, 0, h[(size_t)k], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, (u64)N * CH, &acc, 1));
    const double t_map = now_ms() - t;
    printf("create %.1f ms, reserve+map+access %.1f ms, range at %p\n", t_create, t_map, va);
    // the second pool, allocated AFTER the chunks (somewhere else physically)
    CK(hipMalloc(&A2, poolb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A2, poolb / 8, 77ull);
    CK(hipDeviceSynchronize());
    std::vector<double> r1((size_t)N), r2((size_t)N);
    t = now_ms();
    for (int k = 0; k < N; ++k) r1[(size_t)k] = rate(A, (uint8_t*)va + (u64)k * CH, CH, 1);
    printf("per-chunk probes: %.1f ms for %d chunks\n", now_ms() - t, N);
    for (int k = 0; k < N; ++k) r2[(size_t)k] = rate(A2, (uint8_t*)va + (u64)k * CH, CH, 1);
    printf("chunk vs pool 1:"); for (int k = 0; k < N; ++k) printf(" %.0f", r1[(size_t)k]); printf("\n");
    printf("chunk vs pool 2:"); for (int k = 0; k < N; ++k) printf(" %.0f", r2[(size_t)k]); printf("\n");
    printf("window of 8 GiB at chunk k, vs pool 1 (full pass) | mean of its chunks' rates:\n");
    for (int k = 0; k + per <= N; k += std::max(1, per / 8)) {
        const double w = rate(A, (uint8_t*)va + (u64)k * CH, need, 1);
        double hm = 0;
        for (int j = 0; j < per; ++j) hm += 1.0 / r1[(size_t)(k + j)];
        printf(" %d:%.0f|%.0f", k, w, per / hm);
    }
    printf("\n");
    // compose: best `per` and worst `per` chunks
    std::vector<int> ord((size_t)N);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return r1[(size_t)a] > r1[(size_t)b]; });
    t = now_ms();
    CK(hipMemUnmap(va, (u64)N * CH));
    const double t_unmap = now_ms() - t;
    void *vb = nullptr, *vw = nullptr;
    CK(hipMemAddressReserve(&vb, need, 1ull << 30, nullptr, 0));
    CK(hipMemAddressReserve(&vw, need, 1ull << 30, nullptr, 0));
    t = now_ms();
    for (int j = 0; j < per; ++j) CK(hipMemMap((uint8_t*)vb + (u64)j * CH, CH, 0, h[(size_t)ord[(size_t)j]], 0));
    CK(hipMemSetAccess(vb, need, &acc, 1));
    const double t_map8 = now_ms() - t;
    for (int j = 0; j < per; ++j) CK(hipMemMap((uint8_t*)vw + (u64)j * CH, CH, 0, h[(size_t)ord[(size_t)(N - 1 - j)]], 0));
    CK(hipMemSetAccess(vw, need, &acc, 1));
    const double rb = rate(A, (uint8_t*)vb, need, 1), rw = rate(A, (uint8_t*)vw, need, 1);
    const double rb2 = rate(A2, (uint8_t*)vb, need, 1), rw2 = rate(A2, (uint8_t*)vw, need, 1);
    // the best chunks in REVERSED order (does the order inside the arena matter?)
    void* vr = nullptr;
    CK(hipMemUnmap(vb, need));
    CK(hipMemAddressReserve(&vr, need, 1ull << 30, nullptr, 0));
    for (int j = 0; j < per; ++j) CK(hipMemMap((uint8_t*)vr + (u64)j * CH, CH, 0, h[(size_t)ord[(size_t)(per - 1 - j)]], 0));
    CK(hipMemSetAccess(vr, need, &acc, 1));
    const double rr = rate(A, (uint8_t*)vr, need, 1);
    printf("composed of the %d best chunks: %.0f GB/s (vs pool 2: %.0f), reversed order %.0f; of the %d worst: %.0f (vs pool 2: %.0f); unmap all %.1f ms, map 8 GiB %.1f ms\n",
           per, rb, rb2, rr, per, rw, rw2, t_unmap, t_map8);
    printf("best chunks:"); for (int j = 0; j < per; ++j) printf(" %d", ord[(size_t)j]); printf("\n");
    // release everything that is not in the best arena; then what the next hipMalloc pays
    t = now_ms();
    CK(hipMemUnmap(vw, need));
    for (int j = per; j < N; ++j) CK(hipMemRelease(h[(size_t)ord[(size_t)j]]));
    const double t_rel = now_ms() - t;
    t = now_ms();
    void* m = nullptr;
    CK(hipMalloc(&m, need));
    const double t_malloc = now_ms() - t;
    const double rm = rate(A, (uint8_t*)m, need, 1);
    printf("release %d chunks %.1f ms; hipMalloc(8 GiB) afterwards %.1f ms, its rate %.0f; composed arena again %.0f\n", N - per, t_rel, t_malloc, rm,
           rate(A, (uint8_t*)vr, need, 1));
    // a plain hipMalloc arena, chunk by chunk
    printf("hipMalloc arena per 1 GiB:");
    for (int j = 0; j < 8; ++j) printf(" %.0f", rate(A, (uint8_t*)m + ((u64)j << 30), 1ull << 30, 1));
    printf("\n");
    return 0;
}
