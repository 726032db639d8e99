// Microbenchmark 3: does the power-of-two alignment of the C2 pool (every bitmap exactly 32 MiB, so operands a and
// b of a pair differ by a multiple of 32 MiB) cost HBM bandwidth through channel/bank aliasing?  Same product
// kernel, bitmaps separated by a configurable pad.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../croaring_amd/csrc/rhip_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_fill(u64* p, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
int main() {
    const uint32_t NBM = 256, NC = 4096, PAIRS = 250;
    const u64 nitems = (u64)PAIRS * NC;
    const u64 maxpad = 2ull << 20;
    uint8_t *A, *Oa; u64 *meta, *off, *qr; BBItem* q; GenItem* rq; uint32_t* rc; u64* acc;
    CK(hipMalloc(&A, (u64)NBM * (NC * 8192ull + maxpad))); CK(hipMalloc(&Oa, (u64)PAIRS * (NC * 8192ull + maxpad)));
    CK(hipMalloc(&meta, nitems * 8)); CK(hipMalloc(&off, nitems * 8)); CK(hipMalloc(&q, nitems * sizeof(BBItem)));
    CK(hipMalloc(&rq, nitems * sizeof(GenItem))); CK(hipMalloc(&rc, 64)); CK(hipMalloc(&qr, 64)); CK(hipMalloc(&acc, 8192));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, (u64)NBM * (NC * 8192ull + maxpad) / 8);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<BBItem> h(nitems); std::vector<u64> ho(nitems);
    const u64 pads[] = {0, 8192, 3 * 8192, 17 * 8192 + 256, 1024 * 1024 + 8192, 4096, 256, 8192 * 5 + 1024};
    for (int round = 0; round < 2; ++round)
    for (u64 pad : pads) {
        const u64 stride = NC * 8192ull + pad;
        for (u64 k = 0; k < nitems; ++k) {
            uint32_t p = (uint32_t)(k / NC), c = (uint32_t)(k % NC);
            uint32_t ia = p % NBM, ib = (p * 97 + 1) % NBM;
            BBItem it; it.offa = ia * stride + c * 8192ull; it.offb = ib * stride + c * 8192ull;
            it.a = 0; it.b = 0; it.out = (uint32_t)k; it.pad = 0;
            h[k] = it; ho[k] = p * stride + c * 8192ull;
        }
        CK(hipMemcpy(q, h.data(), nitems * sizeof(BBItem), hipMemcpyHostToDevice));
        CK(hipMemcpy(off, ho.data(), nitems * 8, hipMemcpyHostToDevice));
        u64 hr[2] = {0, nitems}; CK(hipMemcpy(qr, hr, 16, hipMemcpyHostToDevice)); CK(hipMemset(rc, 0, 64));
        OutView O; O.key = nullptr; O.meta = meta; O.off = off; O.arena = Oa; O.slot = nullptr;
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_bb<OP_AND>), dim3(8192), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("pad %8llu B between bitmaps: %7.3f ms  %7.1f GB/s\n", pad, best, (double)nitems * 24576.0 / best / 1e6); fflush(stdout);
    }
    return 0;
}
