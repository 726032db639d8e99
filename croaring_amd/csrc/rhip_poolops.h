// rhip_poolops.h -- kernels that reshape device pools without leaving HBM: bitmap selection across pools
// (the building block of the in-place entry points) and run_optimize /
// remove_run_compression as container conversions.
#pragma once
#include "rhip_common.h"
#include "rhip_runs.h"

// ------------------------------------------------------------------ pool_select
// Output bitmap i is a copy of bitmap src_bitmap[i] of pool src_pool[i].  The host sends, per output
// bitmap, the source pool and the index of its first source container (src_c0) and the output
// bm_start; one thread per output container copies the directory entry, sizes the slot
// (align16(payload)) and records where the payload lives.
struct SelSrc {
    const uint8_t* p;   // payload address in the source arena
    u64 n16;            // 16-byte pieces to copy (the slot may be wider: line-aligned pools)
};
__global__ __launch_bounds__(256) void k_select_dir(const PoolView* __restrict__ srcs,
                                                    const uint32_t* __restrict__ src_pool,
                                                    const u64* __restrict__ src_c0,
                                                    const u64* __restrict__ out_bm_start, uint32_t n_bitmaps, u64 n_out,
                                                    u64* __restrict__ okey, uint8_t* __restrict__ otype,
                                                    uint32_t* __restrict__ ocard, uint32_t* __restrict__ onruns,
                                                    uint32_t* __restrict__ slot, SelSrc* __restrict__ from,
                                                    uint32_t amask) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    // bitmap of output container j: last i with out_bm_start[i] <= j
    u64 lo = 0, hi = n_bitmaps;
    while (lo + 1 < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (out_bm_start[mid] <= j) lo = mid;
        else hi = mid;
    }
    const PoolView V = srcs[src_pool[lo]];
    const u64 c = src_c0[lo] + (j - out_bm_start[lo]);
    const uint8_t t = V.type[c];
    const uint32_t cd = V.card[c], nr = V.nruns[c];
    okey[j] = V.key[c];
    otype[j] = t;
    ocard[j] = cd;
    onruns[j] = nr;
    const uint32_t pb = align16(payload_bytes(t, cd, nr));
    slot[j] = (pb + amask) & ~amask;
    from[j].p = V.arena + V.off[c];
    from[j].n16 = pb >> 4;
}

// Directory-only form for the in-place entry points: output container j takes its directory entry from its source
// and its payload STAYS where it is -- the offset is the source's offset plus off_add[source pool] (0 for the pool
// being updated, the append position for the freshly computed results that were copied behind its arena).
__global__ __launch_bounds__(256) void k_splice_dir(const PoolView* __restrict__ srcs, const u64* __restrict__ off_add,
                                                    const uint32_t* __restrict__ src_pool,
                                                    const u64* __restrict__ src_c0,
                                                    const u64* __restrict__ out_bm_start, uint32_t n_bitmaps, u64 n_out,
                                                    u64* __restrict__ okey, uint8_t* __restrict__ otype,
                                                    uint32_t* __restrict__ ocard, uint32_t* __restrict__ onruns,
                                                    u64* __restrict__ ooff) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    u64 lo = 0, hi = n_bitmaps;
    while (lo + 1 < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (out_bm_start[mid] <= j) lo = mid;
        else hi = mid;
    }
    const uint32_t sp = src_pool[lo];
    const PoolView V = srcs[sp];
    const u64 c = src_c0[lo] + (j - out_bm_start[lo]);
    okey[j] = V.key[c];
    otype[j] = V.type[c];
    ocard[j] = V.card[c];
    onruns[j] = V.nruns[c];
    ooff[j] = V.off[c] + off_add[sp];
}

// One wave per container: the payload in 16-byte pieces, 64 lanes wide.
__global__ __launch_bounds__(256) void k_select_copy(const SelSrc* __restrict__ from,
                                                     const u64* __restrict__ off, uint8_t* __restrict__ arena,
                                                     u64 n_out) {
    const uint32_t lane = lane_id();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 w = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_out; w += nwaves) {
        const uint4* __restrict__ s = (const uint4*)from[w].p;
        uint4* __restrict__ d = (uint4*)(arena + off[w]);
        const uint32_t n16 = (uint32_t)from[w].n16;
        for (uint32_t i = lane; i < n16; i += 64) d[i] = s[i];
    }
}

// ------------------------------------------------------------------ container conversions
enum { CONV_RUN_OPTIMIZE = 0, CONV_REMOVE_RUNS = 1 };

// Slot upper bounds from metadata only.  run_optimize never grows a payload (every conversion of
// convert_run_optimize / convert_run_to_efficient_container, convert.c:154-321, is taken only when the
// new serialized size is smaller); remove_run_compression turns a run container into an array
// (card <= 4096) or a bitset (convert_to_bitset_or_array_container, convert.c:118-147).
__global__ __launch_bounds__(256) void k_convert_slots(PoolView P, u64 n_cont, int mode, uint32_t* __restrict__ slot) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cont) return;
    const uint8_t t = P.type[j];
    const uint32_t cd = P.card[j];
    uint32_t b = payload_bytes(t, cd, P.nruns[j]);
    if (mode == CONV_REMOVE_RUNS && t == T_RUN) b = cd <= 4096u ? 2u * cd : 8192u;
    slot[j] = align16(b);
}

// One WAVE per container.  The container is rasterised into the wave's LDS image, pulled into
// registers (lane l: logical words [32 l, 32 l + 32)), its canonical run count is taken as in k_genw
// (bitset_container_number_of_runs, bitset.c:1046-1062; array_container_number_of_runs, array.h), the
// target type follows the reference and the shared emit fragment writes it:
//   run_optimize (roaring_bitmap_run_optimize, roaring.c:1530-1546 -> convert_run_optimize, convert.c:217-321):
//     run    -> convert_run_to_efficient_container (type_eff)
//     array  -> run iff 2 + 4 n_runs <  2 card
//     bitset -> run iff 2 + 4 n_runs <  8192
//   remove_run_compression (roaring.c:1564-1592): run -> array / bitset by cardinality, others unchanged.
__global__ __launch_bounds__(256) void k_convert(PoolView P, u64 n_cont, int mode, const u64* __restrict__ off,
                                                 uint8_t* __restrict__ arena, uint8_t* __restrict__ otype,
                                                 uint32_t* __restrict__ ocard, uint32_t* __restrict__ onruns) {
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* img = img_all[threadIdx.x >> 6];
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 wi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wi < n_cont; wi += nwaves) {
        const uint32_t ta = P.type[wi], ca = P.card[wi], nra = P.nruns[wi];
        wimg_build(img, P.arena + P.off[wi], ta, ca, nra);
        __builtin_amdgcn_wave_barrier();
        uint32_t r[32];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            r[k] = img[wown(lane, k)];
            cnt += __popc(r[k]);
        }
        const uint32_t rc = wave_sum(cnt);  // == ca for a valid pool
        uint32_t prev_msb = __shfl_up(r[31] >> 31, 1);
        if (lane == 0) prev_msb = 0;
        uint32_t next_lsb = __shfl_down(r[0] & 1u, 1);
        if (lane == 63) next_lsb = 0;
        uint32_t ns = 0;
        {
            uint32_t pm = prev_msb;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                ns += __popc(r[k] & ~((r[k] << 1) | pm));
                pm = r[k] >> 31;
            }
        }
        const uint32_t rn = wave_sum(ns);
        int ty = (int)ta;
        if (mode == CONV_RUN_OPTIMIZE) {
            if (ta == T_RUN) ty = type_eff(rc, rn);
            else if (ta == T_ARRAY) ty = (2u + 4u * rn < 2u * rc) ? T_RUN : T_ARRAY;
            else ty = (2u + 4u * rn < 8192u) ? T_RUN : T_BITSET;
        } else if (ta == T_RUN) {
            ty = type_ba(rc);
        }
        uint8_t* outp = arena + off[wi];
#include "rhip_wemit.inc"
        if (lane == 0) {
            otype[wi] = (uint8_t)ty;
            ocard[wi] = rc;
            onruns[wi] = (ty == T_RUN) ? rn : 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
