// Microbenchmark 8 (round 6): the placement mode of a C2 result arena is a function of VIRTUAL addresses (vmm_place2:
// the same physical chunks stream at 6.4 or 5.9 TB/s depending on where they are mapped; other chunks at the same address
// give the same rate).  Here BOTH the operand pool and the arena are hipMemCreate'd chunks mapped inside one reserved
// range, so the whole (pool address, arena address) plane can be walked in one process:
//   E1  pool at B (B = the range rounded up to 2 GiB), arena at B + 16 GiB + v, v = 0 .. 96 GiB
//   E1b the same with everything shifted by 0x30400000 (what hipMalloc-like alignment looks like)
//   E2  pool at B + p, arena at pool + d: is the rate a function of d alone?
//   E3  the read-only pass (no stores) against the pool's own address
// Not product code.
#include <hip/hip_runtime.h>
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
__global__ __launch_bounds__(256) void k_read_probe(const uint8_t* __restrict__ arenaA, u64 a_items, u64 n_slots, uint32_t* sink) {
    const uint32_t lane = lane_id();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (u64 i = wave_uniform((uint32_t)(((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6)); i < n_slots; i += nwaves) {
        const u32x4* __restrict__ pa = (const u32x4*)(arenaA + (i % a_items) * 8192ull);
        const u32x4* __restrict__ pb = (const u32x4*)(arenaA + ((i * 97ull + 4096ull * 33ull) % a_items) * 8192ull);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = __builtin_nontemporal_load(pa + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 8; ++k) vb[k] = __builtin_nontemporal_load(pb + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 8; ++k) { u32x4 o = va[k] | vb[k]; acc += __popc(o.x) + __popc(o.y) + __popc(o.z) + __popc(o.w); }
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const u64 G = 1ull << 30, CH = G, need = 8 * G, poolb = 8 * G;
    const u64 RANGE = 400 * G;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hp(8), ha(8);
    for (int k = 0; k < 8; ++k) CK(hipMemCreate(&hp[(size_t)k], CH, &prop, 0));
    for (int k = 0; k < 8; ++k) CK(hipMemCreate(&ha[(size_t)k], CH, &prop, 0));
    void* R = nullptr;
    CK(hipMemAddressReserve(&R, RANGE + 2 * G, 0, nullptr, 0));
    uint8_t* B = (uint8_t*)((((uintptr_t)R + 2 * G - 1) / (2 * G)) * (2 * G));
    printf("range at %p, B = %p (%.3f GiB)\n", R, (void*)B, (double)(uintptr_t)B / (double)G);
    // does a reservation honour an address hint?
    {
        void* H = nullptr;
        hipError_t e = hipMemAddressReserve(&H, 16 * G, 0, (void*)0x600000000000ull, 0);
        printf("reserve with hint 0x600000000000: %s -> %p\n", hipGetErrorString(e), H);
        if (e == hipSuccess) (void)hipMemAddressFree(H, 16 * G);
        (void)hipGetLastError();
    }
    uint32_t* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint8_t *poolAt = nullptr, *arenaAt = nullptr;
    auto map8 = [&](std::vector<hipMemGenericAllocationHandle_t>& h, uint8_t* at) {
        for (int j = 0; j < 8; ++j) CK(hipMemMap(at + (u64)j * CH, CH, 0, h[(size_t)j], 0));
        CK(hipMemSetAccess(at, need, &acc, 1));
    };
    auto set_pool = [&](uint8_t* at) { if (poolAt == at) return; if (poolAt) { CK(hipDeviceSynchronize()); CK(hipMemUnmap(poolAt, poolb)); } map8(hp, at); poolAt = at; };
    auto set_arena = [&](uint8_t* at) { if (arenaAt == at) return; if (arenaAt) { CK(hipDeviceSynchronize()); CK(hipMemUnmap(arenaAt, need)); } map8(ha, at); arenaAt = at; };
    set_pool(B);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)B, poolb / 8, 0ull);
    CK(hipDeviceSynchronize());
    auto rate = [&](bool stores) {
        const u64 n_slots = need / 8192ull;
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            if (stores) hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, 0, poolAt, poolb / 8192ull, arenaAt, n_slots, 1ull);
            else hipLaunchKernelGGL(k_read_probe, dim3(8192), dim3(256), 0, 0, poolAt, poolb / 8192ull, n_slots, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return (double)n_slots * (stores ? 24576.0 : 16384.0) / best / 1e6;
    };
    for (int shifted = 0; shifted < 2; ++shifted) {
        const u64 sh = shifted ? 0x30400000ull : 0ull;
        set_pool(B + sh);
        printf("E1%s pool at B%s, arena at B + 16 GiB + v:", shifted ? "b" : "", shifted ? " + 0x30400000" : "");
        for (u64 v = 0; v <= 96; ++v) { set_arena(B + sh + 16 * G + v * G); printf(" %llu:%.0f", (unsigned long long)v, rate(true)); }
        printf("\n");
    }
    const u64 ps[] = {0, 1, 2, 4, 8, 16, 32, 64, 128, 192};
    for (u64 p : ps) {
        printf("E2 pool at B + %llu GiB, arena at pool + d:", (unsigned long long)p);
        for (u64 d = 8; d <= 120; d += 2) {
            set_arena(B + 390 * G);  // out of the way while the pool moves
            set_pool(B + p * G);
            set_arena(B + (p + d) * G);
            printf(" %llu:%.0f", (unsigned long long)d, rate(true));
        }
        printf("\n");
    }
    set_arena(B + 390 * G);
    printf("E3 read-only, pool at B + p:");
    for (u64 p = 0; p <= 40; ++p) { set_pool(B + p * G); printf(" %llu:%.0f", (unsigned long long)p, rate(false)); }
    printf("\nE3b read-only, pool at B + 0x30400000 + p:");
    for (u64 p = 0; p <= 40; p += 4) { set_pool(B + 0x30400000ull + p * G); printf(" %llu:%.0f", (unsigned long long)p, rate(false)); }
    printf("\n");
    return 0;
}
