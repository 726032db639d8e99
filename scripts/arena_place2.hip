// Microbenchmark 5 (round 4): WHERE in the buffers does the k_bb time of a C2 batch go -- by 1 GiB chunk.
// One operand pool (256 x 4096 bitset containers, 8 GiB, hipMalloc), result arenas allocated in different ways.  For every
// arena: the full 250-pair launch (the number bench.py's roofline is taken from), then
//   * read-only  : k_bb in cardinality mode over A-chunk i (bitmaps 32 i .. 32 i + 31, both operands from it)   -> 8 numbers
//   * write-only : a plain fill of R-chunk j (1 GiB of the result arena)                                        -> 8 numbers
//   * the matrix : k_bb<or> reading A-chunk i, writing R-chunk j (131 072 container pairs, 3.2 GB)              -> 8 x 8
// If the slow mode is a property of the result arena's physical pages, columns of the matrix differ; if of the pool's,
// rows; if of their relative placement, a diagonal pattern.  argv[1] = list of ways (malloc,contig,...), each tried once
// per occurrence.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
__global__ void k_fill16(uint4* p, u64 n16) {  // write-only stream, 16 bytes per lane
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
constexpr uint32_t NBM = 256, NC = 4096, CH = 32;  // bitmaps per 1 GiB chunk
__global__ void k_make_queue(BBItem* q, uint32_t ai, uint32_t rj, int cardmode) {
    const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (u64)CH * NC) return;
    const uint32_t m = (uint32_t)(k / NC), c = (uint32_t)(k % NC);
    BBItem it;
    it.offa = ((u64)(CH * ai + m) * NC + c) * 8192ull;
    it.offb = ((u64)(CH * ai + (5 * m + 1) % CH) * NC + c) * 8192ull;
    it.offo = ((u64)rj * CH * NC + k) * 8192ull;
    it.out = cardmode ? m : (uint32_t)((u64)rj * CH * NC + k);
    it.slot = 8192u;
    q[k] = it;
}
int main(int argc, char** argv) {
    std::string ways = argc > 1 ? argv[1] : "malloc,malloc,malloc,contig";
    const u64 poolb = (u64)NBM * NC * 8192ull;
    uint8_t* A;
    CK(hipMalloc(&A, poolb));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, poolb / 8, 0ull);
    u64 *meta, *qr, *qr1, *acc; BBItem *q, *qs; GenItem* rq; uint32_t* rc;
    const u64 nfull = 250ull * NC, nsub = (u64)CH * NC;
    CK(hipMalloc(&meta, (u64)NBM * NC * 8)); CK(hipMalloc(&q, nfull * sizeof(BBItem))); CK(hipMalloc(&qs, nsub * sizeof(BBItem)));
    CK(hipMalloc(&rq, 1024 * sizeof(GenItem))); CK(hipMalloc(&rc, 64)); CK(hipMalloc(&qr, 64)); CK(hipMalloc(&qr1, 64)); CK(hipMalloc(&acc, 8192));
    {
        std::vector<BBItem> h(nfull);
        for (u64 k = 0; k < nfull; ++k) {
            uint32_t p = (uint32_t)(k / NC), c = (uint32_t)(k % NC);
            uint32_t ia = p % NBM, ib = (p * 97 + 1) % NBM;
            BBItem it; it.offa = (u64)ia * NC * 8192ull + c * 8192ull; it.offb = (u64)ib * NC * 8192ull + c * 8192ull;
            it.offo = k * 8192ull; it.out = (uint32_t)k; it.slot = 8192u;
            h[k] = it;
        }
        CK(hipMemcpy(q, h.data(), nfull * sizeof(BBItem), hipMemcpyHostToDevice));
        u64 hr[2] = {0, nfull}; CK(hipMemcpy(qr, hr, 16, hipMemcpyHostToDevice));
        u64 hs[2] = {0, nsub}; CK(hipMemcpy(qr1, hs, 16, hipMemcpyHostToDevice));
        CK(hipMemset(rc, 0, 64)); CK(hipMemset(acc, 0, 8192));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto&& launch) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return best;
    };
    // read-only, per A chunk (independent of the arena): once
    printf("pool at %p\nread-only (k_bb cardinality mode, 2 GiB read per chunk), ms per A-chunk:", (void*)A);
    for (uint32_t i = 0; i < 8; ++i) {
        hipLaunchKernelGGL(k_make_queue, dim3((unsigned)((nsub + 255) / 256)), dim3(256), 0, 0, qs, i, 0u, 1);
        OutView O; O.key = nullptr; O.meta = meta; O.off = nullptr; O.arena = nullptr; O.slot = nullptr;
        const float ms = timed([&] { hipLaunchKernelGGL((k_bb<OP_OR>), dim3(8192), dim3(256), 0, 0, A, A, O, qs, qr1, 1, acc, rq, rc); });
        printf(" %.3f", ms);
    }
    printf("\n");
    std::vector<void*> kept;
    for (size_t pos = 0; pos < ways.size();) {
        size_t e = ways.find(',', pos);
        if (e == std::string::npos) e = ways.size();
        const std::string way = ways.substr(pos, e - pos);
        pos = e + 1;
        void* Rp = nullptr;
        const u64 resb = (u64)NBM * NC * 8192ull;  // 8 GiB: room for 8 R-chunks (the full launch writes 250 / 256 of it)
        if (way == "contig") CK(hipExtMallocWithFlags(&Rp, resb, hipDeviceMallocContiguous));
        else CK(hipMalloc(&Rp, resb));
        kept.push_back(Rp);
        OutView O; O.key = nullptr; O.meta = meta; O.off = nullptr; O.arena = (uint8_t*)Rp; O.slot = nullptr;
        const float full = timed([&] { hipLaunchKernelGGL((k_bb<OP_OR>), dim3(8192), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc); });
        printf("== %s arena %p: full launch %.3f ms = %.0f GB/s\n", way.c_str(), Rp, full, (double)nfull * 24576.0 / full / 1e6);
        printf("   write-only (1 GiB fill), ms per R-chunk:");
        for (uint32_t j = 0; j < 8; ++j) {
            const float ms = timed([&] { hipLaunchKernelGGL(k_fill16, dim3(8192), dim3(256), 0, 0, (uint4*)((uint8_t*)Rp + ((u64)j << 30)), (1ull << 30) / 16); });
            printf(" %.3f", ms);
        }
        printf("\n   k_bb<or> reading A-chunk i (row), writing R-chunk j (column), ms:\n");
        for (uint32_t i = 0; i < 8; ++i) {
            printf("   ");
            for (uint32_t j = 0; j < 8; ++j) {
                hipLaunchKernelGGL(k_make_queue, dim3((unsigned)((nsub + 255) / 256)), dim3(256), 0, 0, qs, i, j, 0);
                const float ms = timed([&] { hipLaunchKernelGGL((k_bb<OP_OR>), dim3(8192), dim3(256), 0, 0, A, A, O, qs, qr1, 0, acc, rq, rc); });
                printf(" %.3f", ms);
            }
            printf("\n");
        }
        fflush(stdout);
    }
    for (void* p : kept) CK(hipFree(p));
    return 0;
}
