// Microbenchmark 2: the PRODUCT k_bb kernel (included from the engine sources) against ablations,
// same process, interleaved rounds.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
#include "../croaring_amd/csrc/rhip_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_fill(u64* p, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = i * 0x9E3779B97F4A7C15ull + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}

// ablation kernel: VAR bit0: out offset computed (no O.off load); bit1: no meta store; bit2: plain loads; bit3: 2 items in flight
template <int VAR>
__global__ __launch_bounds__(256) void k_abl(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                             OutView O, const BBItem* __restrict__ q, uint32_t n) {
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
        const BBItem t = q[w];
        const u32x4* __restrict__ pa = (const u32x4*)(arenaA + t.offa);
        const u32x4* __restrict__ pb = (const u32x4*)(arenaB + t.offb);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = (VAR & 4) ? pa[i * 64 + lane] : __builtin_nontemporal_load(pa + i * 64 + lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = (VAR & 4) ? pb[i * 64 + lane] : __builtin_nontemporal_load(pb + i * 64 + lane);
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { va[i] = va[i] & vb[i]; cnt += vpopc(va[i]); }
        const uint32_t card = wave_sum(cnt);
        const u64 ooff = (VAR & 1) ? (u64)t.out * 8192ull : O.off[t.out];
        u32x4* __restrict__ po = (u32x4*)(O.arena + ooff);
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(va[i], po + i * 64 + lane);
        if (!(VAR & 2) && lane == 0) O.meta[t.out] = pack_meta(T_BITSET, card, 0);
        if ((VAR & 2) && card == 0xFFFFFFFFu && lane == 0) O.meta[t.out] = 1;
    }
}

template <class F>
float timeit(F f, int reps = 4) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t NBM = 256, NC = 4096, PAIRS = 250;
    const u64 ncont = (u64)NBM * NC, nitems = (u64)PAIRS * NC;
    uint8_t *A, *Oa; u64 *meta, *off, *qr; BBItem* q; GenItem* rq; uint32_t* rc; u64* acc;
    CK(hipMalloc(&A, ncont * 8192)); CK(hipMalloc(&Oa, nitems * 8192));
    CK(hipMalloc(&meta, nitems * 8)); CK(hipMalloc(&off, nitems * 8)); CK(hipMalloc(&q, nitems * sizeof(BBItem)));
    CK(hipMalloc(&rq, nitems * sizeof(GenItem))); CK(hipMalloc(&rc, 64)); CK(hipMalloc(&qr, 64)); CK(hipMalloc(&acc, 8 * 1024));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, ncont * 1024);
    std::vector<BBItem> h(nitems); std::vector<u64> ho(nitems);
    for (int order = 0; order < 2; ++order) {
        // order 0: items in pair-major order; order 1: 64-item tiles of different pairs interleaved (old atomic-queue order)
        for (u64 k = 0; k < nitems; ++k) {
            u64 src = k;
            if (order == 1) { u64 tile = k / 64, l = k % 64; u64 ntile = nitems / 64; u64 st = (tile * 7919) % ntile; src = st * 64 + l; }
            uint32_t p = (uint32_t)(src / NC), c = (uint32_t)(src % NC);
            uint32_t ia = p % NBM, ib = (p * 97 + 1) % NBM;
            BBItem it; it.offa = ((u64)ia * NC + c) * 8192; it.offb = ((u64)ib * NC + c) * 8192;
            it.a = ia * NC + c; it.b = ib * NC + c; it.out = (uint32_t)src; it.pad = 0;
            h[k] = it; ho[k] = k * 8192;
        }
        CK(hipMemcpy(q, h.data(), nitems * sizeof(BBItem), hipMemcpyHostToDevice));
        CK(hipMemcpy(off, ho.data(), nitems * 8, hipMemcpyHostToDevice));
        u64 hr[2] = {0, nitems}; CK(hipMemcpy(qr, hr, 16, hipMemcpyHostToDevice)); CK(hipMemset(rc, 0, 64));
        OutView O; O.key = nullptr; O.meta = meta; O.off = off; O.arena = Oa; O.slot = nullptr;
        const double bytes = (double)nitems * 24576.0;
        uint32_t n = (uint32_t)nitems;
        auto rep = [&](const char* name, float ms) { printf("[order %d] %-44s %8.3f ms  %8.1f GB/s\n", order, name, ms, bytes / ms / 1e6); fflush(stdout); };
        for (int round = 0; round < 2; ++round) {
            for (int g : {2048, 4096, 8192}) {
                char nm[64]; snprintf(nm, 64, "product k_bb<AND> grid=%d", g);
                rep(nm, timeit([&] { hipLaunchKernelGGL((k_bb<OP_AND>), dim3(g), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc); }));
            }
            rep("product k_bb<OR> grid=4096", timeit([&] { hipLaunchKernelGGL((k_bb<OP_OR>), dim3(4096), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc); }));
            rep("abl0 (same structure) grid=4096", timeit([&] { hipLaunchKernelGGL((k_abl<0>), dim3(4096), dim3(256), 0, 0, A, A, O, q, n); }));
            rep("abl1 computed out offset", timeit([&] { hipLaunchKernelGGL((k_abl<1>), dim3(4096), dim3(256), 0, 0, A, A, O, q, n); }));
            rep("abl2 no meta store", timeit([&] { hipLaunchKernelGGL((k_abl<2>), dim3(4096), dim3(256), 0, 0, A, A, O, q, n); }));
            rep("abl3 computed off + no meta", timeit([&] { hipLaunchKernelGGL((k_abl<3>), dim3(4096), dim3(256), 0, 0, A, A, O, q, n); }));
            rep("abl4 plain loads", timeit([&] { hipLaunchKernelGGL((k_abl<4>), dim3(4096), dim3(256), 0, 0, A, A, O, q, n); }));
        }
    }
    {
        // DVFS / idle-gap hypothesis: same kernel, timed individually, with different things in front of it
        OutView O; O.key = nullptr; O.meta = meta; O.off = off; O.arena = Oa; O.slot = nullptr;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        auto one = [&](hipStream_t s) { float ms; CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL((k_bb<OP_AND>), dim3(4096), dim3(256), 0, s, A, A, O, q, qr, 0, acc, rq, rc);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
        for (int mode = 0; mode < 5; ++mode) {
            float sum = 0; int n = 0;
            for (int r = 0; r < 10; ++r) {
                hipStream_t s = (mode == 4) ? st : 0;
                if (mode == 1) { CK(hipDeviceSynchronize()); usleep(500); }
                if (mode == 2) { CK(hipDeviceSynchronize()); usleep(5000); }
                if (mode == 3) { hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, s, (u64*)Oa, (u64)200000); CK(hipDeviceSynchronize()); }
                float ms = one(s);
                if (r >= 2) { sum += ms; ++n; }
            }
            const char* nm[] = {"back-to-back", "sync + 0.5 ms idle before", "sync + 5 ms idle before", "1-wave 64-lane kernel + sync before", "non-blocking stream back-to-back"};
            printf("gap test: %-40s %8.3f ms  %8.1f GB/s\n", nm[mode], sum / n, (double)nitems * 24576.0 / (sum / n) / 1e6); fflush(stdout);
        }
    }
    return 0;
}
