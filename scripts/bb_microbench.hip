// Microbenchmark for the bitset x bitset streaming kernel: variants of the access pattern,
// timed with HIP events on an 8 GiB pool (past the 256 MiB Infinity Cache).  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

struct Item { uint32_t a, b, out; };

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t vpopc(u32x4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

// V0: current product kernel shape: persistent waves, nt loads + nt stores
template <int NTL, int NTS, int PERSIST>
__global__ __launch_bounds__(256) void k_v0(const uint8_t* __restrict__ A, uint8_t* __restrict__ O, const Item* __restrict__ q,
                                            uint32_t n, uint32_t* __restrict__ cards) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += (PERSIST ? nwaves : 0xFFFFFFFFu - w)) {
        Item t = q[w];
        const u32x4* pa = (const u32x4*)(A + (u64)t.a * 8192);
        const u32x4* pb = (const u32x4*)(A + (u64)t.b * 8192);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = NTL ? __builtin_nontemporal_load(pa + i * 64 + lane) : pa[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = NTL ? __builtin_nontemporal_load(pb + i * 64 + lane) : pb[i * 64 + lane];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { va[i] = va[i] & vb[i]; cnt += vpopc(va[i]); }
        cnt = wave_sum(cnt);
        u32x4* po = (u32x4*)(O + (u64)t.out * 8192);
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (NTS) __builtin_nontemporal_store(va[i], po + i * 64 + lane); else po[i * 64 + lane] = va[i]; }
        if (lane == 0) cards[t.out] = cnt;
        if (!PERSIST) break;
    }
}

// V1: one 256-thread block per container pair (2 x 16 B per thread per operand), card via LDS
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_v1(const uint8_t* __restrict__ A, uint8_t* __restrict__ O, const Item* __restrict__ q,
                                            uint32_t n, uint32_t* __restrict__ cards) {
    __shared__ uint32_t ws[4];
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        Item t = q[w];
        const u32x4* pa = (const u32x4*)(A + (u64)t.a * 8192);
        const u32x4* pb = (const u32x4*)(A + (u64)t.b * 8192);
        u32x4 a0 = NTL ? __builtin_nontemporal_load(pa + threadIdx.x) : pa[threadIdx.x];
        u32x4 a1 = NTL ? __builtin_nontemporal_load(pa + 256 + threadIdx.x) : pa[256 + threadIdx.x];
        u32x4 b0 = NTL ? __builtin_nontemporal_load(pb + threadIdx.x) : pb[threadIdx.x];
        u32x4 b1 = NTL ? __builtin_nontemporal_load(pb + 256 + threadIdx.x) : pb[256 + threadIdx.x];
        a0 &= b0; a1 &= b1;
        uint32_t cnt = wave_sum(vpopc(a0) + vpopc(a1));
        u32x4* po = (u32x4*)(O + (u64)t.out * 8192);
        if (NTS) { __builtin_nontemporal_store(a0, po + threadIdx.x); __builtin_nontemporal_store(a1, po + 256 + threadIdx.x); }
        else { po[threadIdx.x] = a0; po[256 + threadIdx.x] = a1; }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) cards[t.out] = ws[0] + ws[1] + ws[2] + ws[3];
    }
}

// V2: flat streaming, no item indirection: thread i handles uint4 i of a contiguous 2-in/1-out stream
// (upper bound for this access mix: "a[i] & b[i] -> o[i]" over huge contiguous arrays)
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_v2(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32x4 x = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        u32x4 y = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        x &= y;
        if (NTS) __builtin_nontemporal_store(x, o + i); else o[i] = x;
    }
}
// V3: flat with 4 x unroll per thread (more bytes in flight per lane)
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_v3(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        u32x4 x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = NTL ? __builtin_nontemporal_load(a + i + k * stride) : a[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = NTL ? __builtin_nontemporal_load(b + i + k * stride) : b[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] &= y[k]; if (NTS) __builtin_nontemporal_store(x[k], o + i + k * stride); else o[i + k * stride] = x[k]; }
    }
}
// V4: pure copy (1 read : 1 write) for the achievable-bandwidth reference
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ a, u32x4* __restrict__ o, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) o[i] = a[i];
}
// V5: read-only (popcount reduce) -- and_cardinality shape
__global__ __launch_bounds__(256) void k_read2(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u64 n, uint32_t* out) {
    uint32_t c = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) c += vpopc(a[i] & b[i]);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0 && c == 0xFFFFFFFF) out[0] = c;
}

__global__ void k_fill(u64* p, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
template <class F>
float timeit(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const uint32_t NBM = 256, NC = 4096, PAIRS = 250;
    const u64 ncont = (u64)NBM * NC;
    uint8_t *A, *O; uint32_t* cards; Item* q;
    CK(hipMalloc(&A, ncont * 8192));
    const u64 nitems = (u64)PAIRS * NC;
    CK(hipMalloc(&O, nitems * 8192));
    CK(hipMalloc(&cards, nitems * 4));
    CK(hipMalloc(&q, nitems * sizeof(Item)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, ncont * 1024);
    CK(hipDeviceSynchronize());
    std::vector<Item> h(nitems);
    for (uint32_t p = 0; p < PAIRS; ++p)
        for (uint32_t c = 0; c < NC; ++c) {
            uint32_t ia = p % NBM, ib = (p * 97 + 1) % NBM;
            h[(u64)p * NC + c] = Item{ia * NC + c, ib * NC + c, (uint32_t)((u64)p * NC + c)};
        }
    CK(hipMemcpy(q, h.data(), nitems * sizeof(Item), hipMemcpyHostToDevice));
    const double bytes = (double)nitems * 24576.0;
    auto rep = [&](const char* name, float ms, double by) { printf("%-46s %8.3f ms  %8.1f GB/s\n", name, ms, by / ms / 1e6); fflush(stdout); };
    uint32_t n = (uint32_t)nitems;
    for (int blocks : {512, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, 96, "v0 persist ntl+nts grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v0<1, 1, 1>), dim3(blocks), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    }
    rep("v0 persist plain/plain grid=2048", timeit([&] { hipLaunchKernelGGL((k_v0<0, 0, 1>), dim3(2048), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    rep("v0 persist ntl/plain grid=2048", timeit([&] { hipLaunchKernelGGL((k_v0<1, 0, 1>), dim3(2048), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    rep("v0 persist plain/nts grid=2048", timeit([&] { hipLaunchKernelGGL((k_v0<0, 1, 1>), dim3(2048), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    rep("v0 one-wave-per-item ntl+nts", timeit([&] { hipLaunchKernelGGL((k_v0<1, 1, 0>), dim3((n + 3) / 4), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    rep("v0 one-wave-per-item plain", timeit([&] { hipLaunchKernelGGL((k_v0<0, 0, 0>), dim3((n + 3) / 4), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    for (int blocks : {2048, 8192}) {
        char nm[96];
        snprintf(nm, 96, "v1 block-per-item ntl+nts grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v1<1, 1>), dim3(blocks), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
        snprintf(nm, 96, "v1 block-per-item plain grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v1<0, 0>), dim3(blocks), dim3(256), 0, 0, A, O, q, n, cards); }), bytes);
    }
    // flat streams: 2 x 4 GiB in, 4 GiB out
    const u64 nvec = (u64)4 << 30 >> 4;
    const u32x4* fa = (const u32x4*)A; const u32x4* fb = (const u32x4*)(A + ((u64)4 << 30)); u32x4* fo = (u32x4*)O;
    const double fbytes = 3.0 * (double)((u64)4 << 30);
    for (int blocks : {2048, 8192}) {
        char nm[96];
        snprintf(nm, 96, "v2 flat plain grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v2<0, 0>), dim3(blocks), dim3(256), 0, 0, fa, fb, fo, nvec); }), fbytes);
        snprintf(nm, 96, "v2 flat ntl+nts grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v2<1, 1>), dim3(blocks), dim3(256), 0, 0, fa, fb, fo, nvec); }), fbytes);
        snprintf(nm, 96, "v3 flat x4 plain grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v3<0, 0>), dim3(blocks), dim3(256), 0, 0, fa, fb, fo, nvec); }), fbytes);
        snprintf(nm, 96, "v3 flat x4 ntl+nts grid=%d", blocks);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_v3<1, 1>), dim3(blocks), dim3(256), 0, 0, fa, fb, fo, nvec); }), fbytes);
    }
    rep("copy 4GiB->4GiB grid=8192", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, fa, fo, nvec); }), 2.0 * (double)((u64)4 << 30));
    rep("read2 popcount 2x4GiB grid=8192", timeit([&] { hipLaunchKernelGGL(k_read2, dim3(8192), dim3(256), 0, 0, fa, fb, nvec, cards); }), 2.0 * (double)((u64)4 << 30));
    return 0;
}
