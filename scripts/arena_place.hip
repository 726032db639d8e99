// Microbenchmark 4 (round 4): WHY does k_bb on C2 take 4.40 or 4.68 ms depending on where the result arena lands?
// One operand pool (256 x 4096 bitset containers, 8 GiB), the product kernel k_bb<OP_OR> over 250 pairs, the RESULT arena
// allocated in different ways, several fresh allocations per way (all kept alive until the way is done, so that every
// trial gets different physical pages):
//   malloc      hipMalloc
//   vmm_min     hipMemCreate (ONE handle) + hipMemMap, size / alignment rounded to the MINIMUM granularity
//   vmm_rec     the same at the RECOMMENDED granularity
//   vmm_1g      eight ... handles of 1 GiB mapped back to back
// argv[1] = trials per way (default 6); argv[2] = list of ways, e.g. "malloc,vmm_rec".
// Under `rocprofv3 --pmc ...` the per-dispatch counters line up with the printed rows (3 launches per trial).
// Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
struct Arena {
    void* p = nullptr;
    size_t size = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    bool vmm = false;
};
static size_t g_min = 0, g_rec = 0;
static Arena alloc_arena(const std::string& way, size_t bytes) {
    Arena a;
    if (way == "malloc") {
        CK(hipMalloc(&a.p, bytes));
        a.size = bytes;
        return a;
    }
    if (way == "contig") {  // physically contiguous VRAM (KFD contiguous flag)
        CK(hipExtMallocWithFlags(&a.p, bytes, hipDeviceMallocContiguous));
        a.size = bytes;
        return a;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    // vmm_a2m / vmm_a1g / vmm_a32g: ONE handle, the VIRTUAL range aligned to 2 MiB / 1 GiB / 32 GiB (a page-table fragment
    // needs virtual and physical address aligned alike)
    const size_t valign = way == "vmm_a2m" ? (2ull << 20) : way == "vmm_a1g" ? (1ull << 30) : way == "vmm_a32g" ? (32ull << 30) : 0;
    const size_t gran = valign ? valign : (way == "vmm_min" ? g_min : g_rec);
    const size_t chunk = way == "vmm_1g" ? (1ull << 30) : 0;
    const size_t size = (bytes + gran - 1) / gran * gran;
    a.vmm = true;
    a.size = chunk ? (size + chunk - 1) / chunk * chunk : size;
    CK(hipMemAddressReserve(&a.p, a.size, chunk ? chunk : gran, nullptr, 0));
    if (chunk) {
        for (size_t o = 0; o < a.size; o += chunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((char*)a.p + o, chunk, 0, h, 0));
            a.hs.push_back(h);
        }
    } else {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, a.size, &prop, 0));
        CK(hipMemMap(a.p, a.size, 0, h, 0));
        a.hs.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(a.p, a.size, &acc, 1));
    return a;
}
static void free_arena(Arena& a) {
    if (!a.vmm) { CK(hipFree(a.p)); return; }
    CK(hipMemUnmap(a.p, a.size));
    for (auto h : a.hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(a.p, a.size));
}
int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 6;
    std::string ways = argc > 2 ? argv[2] : "malloc,vmm_min,vmm_rec,vmm_1g";
    const uint32_t NBM = 256, NC = 4096, PAIRS = 250;
    const u64 nitems = (u64)PAIRS * NC;
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        CK(hipMemGetAllocationGranularity(&g_min, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&g_rec, &prop, hipMemAllocationGranularityRecommended));
        printf("granularity: minimum %zu B, recommended %zu B\n", g_min, g_rec);
    }
    uint8_t* A; u64 *meta, *qr; BBItem* q; GenItem* rq; uint32_t* rc; u64* acc;
    if (ways.rfind("slab", 0) == 0) {
        // ONE allocation ("slab:malloc" / "slab:contig") holds the pool AND the result arena: pool at 0, arena at
        // 8 GiB + skew.  If the k_bb time is a function of the RELATIVE placement of the three streams, it is a
        // deterministic function of skew here (at least with physically contiguous memory); `trials` fresh slabs.
        const std::string sway = ways.size() > 5 ? ways.substr(5) : "contig";
        const u64 poolb = (u64)NBM * NC * 8192ull, resb = nitems * 8192ull;
        std::vector<u64> skews = {0, 4096, 65536, 1ull << 20, 2ull << 20, 3ull << 20, 16ull << 20, 32ull << 20, 48ull << 20, 128ull << 20,
                                  256ull << 20, 384ull << 20, 512ull << 20, 768ull << 20, 1024ull << 20, 1536ull << 20, 2048ull << 20};
        // SLAB_GIB=n: the slab is n GiB (a power of two: one buddy block if the driver has one) and the arena is tried at
        // whole-GiB distances behind the pool, as far as the slab reaches
        u64 slab_bytes = poolb + resb + (2048ull << 20) + 4096;
        if (getenv("SLAB_GIB")) {
            slab_bytes = strtoull(getenv("SLAB_GIB"), nullptr, 0) << 30;
            skews.clear();
            // SLAB_STEP_MIB: distance step (default 1 GiB); SLAB_FROM_GIB / SLAB_TO_GIB: range of distances behind the pool
            const u64 step = (getenv("SLAB_STEP_MIB") ? strtoull(getenv("SLAB_STEP_MIB"), nullptr, 0) : 1024ull) << 20;
            const u64 from = (getenv("SLAB_FROM_GIB") ? strtoull(getenv("SLAB_FROM_GIB"), nullptr, 0) : 0ull) << 30;
            const u64 to = getenv("SLAB_TO_GIB") ? strtoull(getenv("SLAB_TO_GIB"), nullptr, 0) << 30 : ~0ull;
            for (u64 d = from; poolb + d + resb <= slab_bytes && d <= to; d += step) skews.push_back(d);
        }
        CK(hipMalloc(&meta, nitems * 8)); CK(hipMalloc(&q, nitems * sizeof(BBItem)));
        CK(hipMalloc(&rq, 1024 * sizeof(GenItem))); CK(hipMalloc(&rc, 64)); CK(hipMalloc(&qr, 64)); CK(hipMalloc(&acc, 8192));
        std::vector<BBItem> h(nitems);
        for (u64 k = 0; k < nitems; ++k) {
            uint32_t p = (uint32_t)(k / NC), c = (uint32_t)(k % NC);
            uint32_t ia = p % NBM, ib = (p * 97 + 1) % NBM;
            BBItem it; it.offa = (u64)ia * NC * 8192ull + c * 8192ull; it.offb = (u64)ib * NC * 8192ull + c * 8192ull;
            it.offo = k * 8192ull; it.out = (uint32_t)k; it.slot = 8192u;
            h[k] = it;
        }
        CK(hipMemcpy(q, h.data(), nitems * sizeof(BBItem), hipMemcpyHostToDevice));
        u64 hr[2] = {0, nitems}; CK(hipMemcpy(qr, hr, 16, hipMemcpyHostToDevice)); CK(hipMemset(rc, 0, 64));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<Arena> slabs;
        for (int t = 0; t < trials; ++t) {
            Arena sl = alloc_arena(sway, slab_bytes);
            slabs.push_back(sl);
            A = (uint8_t*)sl.p;
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, poolb / 8);
            CK(hipDeviceSynchronize());
            printf("slab %d (%s) at %p:", t, sway.c_str(), sl.p);
            for (u64 skew : skews) {
                OutView O; O.key = nullptr; O.meta = meta; O.off = nullptr; O.arena = A + poolb + skew; O.slot = nullptr;
                float best = 1e30f;
                for (int r = 0; r < 3; ++r) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL((k_bb<OP_OR>), dim3(8192), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r && ms < best) best = ms;
                }
                if (getenv("SLAB_GIB")) printf("  %.3fG:%.3f", (double)skew / (double)(1ull << 30), best);
                else printf("  %llu:%.3f", (unsigned long long)(skew >> 12), best);
                fflush(stdout);
            }
            printf("   (skew in 4 KiB pages : ms)\n");
            if (getenv("VARIANTS")) {
                // the same slab, arena at 8 GiB + 0: result slots ROTATED inside the arena (slot of item k = (k + rot) mod n),
                // the queue visited in a PERMUTED chunk order (chunk c' = c * 7919 mod nchunks), B = A
                auto run_q = [&](const char* name, u64 par) {
                    CK(hipMemcpy(q, h.data(), nitems * sizeof(BBItem), hipMemcpyHostToDevice));
                    OutView O; O.key = nullptr; O.meta = meta; O.off = nullptr; O.arena = A + poolb; O.slot = nullptr;
                    float best = 1e30f;
                    for (int r = 0; r < 3; ++r) {
                        CK(hipEventRecord(e0));
                        hipLaunchKernelGGL((k_bb<OP_OR>), dim3(8192), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r && ms < best) best = ms;
                    }
                    printf("  %s %llu: %.3f\n", name, (unsigned long long)par, best); fflush(stdout);
                };
                const std::vector<BBItem> h0 = h;
                for (u64 g = 1; g <= 7; ++g) {
                    const u64 rot = g * (1ull << 30) / 8192;
                    for (u64 k = 0; k < nitems; ++k) { h[k] = h0[k]; h[k].offo = ((k + rot) % nitems) * 8192ull; }
                    run_q("rot_GiB", g);
                }
                for (u64 cs : {32ull, 128ull, 1024ull, 4096ull}) {
                    const u64 nch = (nitems + cs - 1) / cs;
                    for (u64 k = 0; k < nitems; ++k) {
                        const u64 c = k / cs, cp = (c * 7919ull) % nch, src = cp * cs + k % cs;
                        h[k] = h0[src < nitems ? src : k];
                    }
                    run_q("perm_chunk", cs);
                }
                for (u64 k = 0; k < nitems; ++k) { h[k] = h0[k]; h[k].offb = h[k].offa; }
                run_q("b_equals_a", 0);
                for (u64 k = 0; k < nitems; ++k) { h[k] = h0[k]; h[k].offa = h0[nitems - 1 - k].offa; }
                run_q("a_reversed", 0);
                for (u64 k = 0; k < nitems; ++k) { h[k] = h0[k]; h[k].offo = (nitems - 1 - k) * 8192ull; }
                run_q("r_reversed", 0);
                h = h0;
            }
        }
        for (auto& a : slabs) free_arena(a);
        return 0;
    }
    const std::string pool_way = getenv("POOL_WAY") ? getenv("POOL_WAY") : "malloc";
    // SIZE_MODE=bench: pool and arenas sized as the engine's DBuf::ensure sizes them (n + n / 8 + 256) instead of exactly
    const bool bench_size = getenv("SIZE_MODE") && !strcmp(getenv("SIZE_MODE"), "bench");
    auto sized = [&](u64 n) { return bench_size ? n + n / 8 + 256 : n; };
    // POOL_BYTES / ARENA_BYTES: allocation sizes in bytes (at least what is needed), overriding SIZE_MODE
    const u64 pool_bytes = getenv("POOL_BYTES") ? strtoull(getenv("POOL_BYTES"), nullptr, 0) : sized((u64)NBM * NC * 8192ull + 64);
    const u64 arena_bytes = getenv("ARENA_BYTES") ? strtoull(getenv("ARENA_BYTES"), nullptr, 0) : sized(nitems * 8192ull + 4096);
    if (pool_bytes < (u64)NBM * NC * 8192ull || arena_bytes < nitems * 8192ull) { printf("POOL_BYTES / ARENA_BYTES too small\n"); return 1; }
    printf("pool %llu bytes, arenas %llu bytes\n", (unsigned long long)pool_bytes, (unsigned long long)arena_bytes);
    Arena poolA = alloc_arena(pool_way, pool_bytes);
    A = (uint8_t*)poolA.p;
    CK(hipMalloc(&meta, nitems * 8)); CK(hipMalloc(&q, nitems * sizeof(BBItem)));
    CK(hipMalloc(&rq, 1024 * sizeof(GenItem))); CK(hipMalloc(&rc, 64)); CK(hipMalloc(&qr, 64)); CK(hipMalloc(&acc, 8192));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64*)A, (u64)NBM * NC * 1024ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<BBItem> h(nitems);
    for (u64 k = 0; k < nitems; ++k) {
        uint32_t p = (uint32_t)(k / NC), c = (uint32_t)(k % NC);
        uint32_t ia = p % NBM, ib = (p * 97 + 1) % NBM;
        BBItem it; it.offa = (u64)ia * NC * 8192ull + c * 8192ull; it.offb = (u64)ib * NC * 8192ull + c * 8192ull;
        it.offo = k * 8192ull; it.out = (uint32_t)k; it.slot = 8192u;
        h[k] = it;
    }
    CK(hipMemcpy(q, h.data(), nitems * sizeof(BBItem), hipMemcpyHostToDevice));
    u64 hr[2] = {0, nitems}; CK(hipMemcpy(qr, hr, 16, hipMemcpyHostToDevice)); CK(hipMemset(rc, 0, 64));
    printf("pool: %s at %p\n", pool_way.c_str(), (void*)A);
    std::vector<std::string> wl;
    for (size_t pos = 0; pos < ways.size();) {
        size_t e = ways.find(',', pos);
        if (e == std::string::npos) e = ways.size();
        wl.push_back(ways.substr(pos, e - pos));
        pos = e + 1;
    }
    // trial t of every way before trial t + 1 of any (every way sees the allocator in comparable states); all arenas
    // stay alive until the end
    std::vector<Arena> kept;
    for (int t = 0; t < trials; ++t) {
        for (const std::string& way : wl) {
            Arena a = alloc_arena(way, arena_bytes);
            OutView O; O.key = nullptr; O.meta = meta; O.off = nullptr; O.arena = (uint8_t*)a.p; O.slot = nullptr;
            float best = 1e30f, first = 0;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_bb<OP_OR>), dim3(8192), dim3(256), 0, 0, A, A, O, q, qr, 0, acc, rq, rc);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r == 0) first = ms;
                if (r && ms < best) best = ms;
            }
            printf("%-8s trial %d: arena %p  first %7.3f ms  best %7.3f ms  %7.1f GB/s\n", way.c_str(), t, a.p, first, best,
                   (double)nitems * 24576.0 / best / 1e6);
            fflush(stdout);
            kept.push_back(a);
        }
    }
    for (auto& a : kept) free_arena(a);
    return 0;
}
