// rhip_flip.h -- roaring_bitmap_flip (roaring.h:986, src/roaring.c:2289-2342) for every bitmap of a pool:
// bitmap i is negated on [start_i, end_i).  Result candidates of a bitmap, in key order: its containers below the
// range (copied), one candidate per key of the range (the negated source container -- container_not_range /
// container_not, containers.h:2009-2073 -- or, where the source has none, container_range_of_ones,
// containers.h:300-312), its containers above the range (copied).  Empty results are dropped by the same
// compaction as the pairwise results.
#pragma once
#include "rhip_common.h"
#include "rhip_runs.h"

struct FlipBm {          // per bitmap, computed by the host from the directory mirror
    u64 c0, lo, hi, c1;  // source containers [c0, c1); [lo, hi) are those whose keys lie in [ks, ke]
    u64 ks;              // first key of the range (16-bit keys for 32-bit pools, 48-bit keys for 64-bit pools)
    uint32_t nk;         // number of keys in it (0: nothing to flip, pure copy)
    uint32_t s_low, e_low;  // low 16 bits of the first / last flipped value (closed range)
    uint32_t pad;
};
struct FlipWork {
    uint32_t src;    // source container (NONE32: the source has no container under this key)
    uint32_t range;  // flo | fhi << 16 (closed, flo <= fhi, inside the container); FLIP_COPY: copy the source unchanged
};
#define FLIP_COPY 0x00000001u  // flo = 1 > fhi = 0: never a real range

// One thread per candidate: key, slot upper bound, work descriptor.
__global__ __launch_bounds__(256) void k_flip_plan(PoolView P, const FlipBm* __restrict__ fb,
                                                   const u64* __restrict__ cand_start, uint32_t n_bitmaps, u64 n_cand,
                                                   u64* __restrict__ okey, uint32_t* __restrict__ oslot,
                                                   FlipWork* __restrict__ work) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cand) return;
    u64 lo = 0, hi = n_bitmaps;  // last bitmap whose candidates start at or before j
    while (lo + 1 < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (cand_start[mid] <= j) lo = mid;
        else hi = mid;
    }
    const FlipBm F = fb[lo];
    const u64 local = j - cand_start[lo];
    const u64 n_before = F.lo - F.c0;
    FlipWork w;
    u64 key;
    uint32_t slot;
    if (local < n_before || local >= n_before + F.nk) {
        const u64 src = local < n_before ? F.c0 + local : F.hi + (local - n_before - F.nk);
        key = P.key[src];
        slot = align16(payload_bytes(P.type[src], P.card[src], P.nruns[src]));
        w.src = (uint32_t)src;
        w.range = FLIP_COPY;
    } else {
        const uint32_t t = (uint32_t)(local - n_before);
        key = F.ks + t;
        const u64 c = lower_bound(P.key, F.lo, F.hi, key);
        const bool present = c < F.hi && P.key[c] == key;
        const uint32_t flo = t == 0 ? F.s_low : 0u, fhi = t == F.nk - 1 ? F.e_low : 65535u;
        w.src = present ? (uint32_t)c : NONE32;
        w.range = flo | (fhi << 16);
        slot = present ? 8192u : 16u;
    }
    okey[j] = key;
    oslot[j] = slot;
    work[j] = w;
}

// One WAVE per candidate.  Negated containers are typed as the reference types them: bitset and array sources
// -> array if the result has <= 4096 values, else bitset (mixed_negation.c:97-196); run sources ->
// convert_run_to_efficient_container (mixed_negation.c:235-266); no source -> one value: array, more: one run.
__global__ __launch_bounds__(256) void k_flip(PoolView P, OutView O, const FlipWork* __restrict__ work, u64 n_cand) {
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* img = img_all[threadIdx.x >> 6];
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 j = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; j < n_cand; j += nwaves) {
        const FlipWork w = work[j];
        uint8_t* outp = O.arena + O.off[j];
        if (w.range == FLIP_COPY) {
            const uint32_t t = P.type[w.src], cd = P.card[w.src], nr = P.nruns[w.src];
            const uint32_t n16 = align16(payload_bytes((uint8_t)t, cd, nr)) >> 4;
            const uint4* __restrict__ s = (const uint4*)(P.arena + P.off[w.src]);
            for (uint32_t i = lane; i < n16; i += 64) ((uint4*)outp)[i] = s[i];
            if (lane == 0) O.meta[j] = pack_meta(t, cd, nr);
            continue;
        }
        const uint32_t flo = w.range & 0xFFFFu, fhi = w.range >> 16;
        if (w.src == NONE32) {
            const uint32_t cd = fhi - flo + 1u;
            if (lane == 0) {
                if (cd == 1u) {
                    *(uint16_t*)outp = (uint16_t)flo;
                    O.meta[j] = pack_meta(T_ARRAY, 1u, 0u);
                } else {
                    *(uint32_t*)outp = flo | ((cd - 1u) << 16);
                    O.meta[j] = pack_meta(T_RUN, cd, 1u);
                }
            }
            continue;
        }
        const uint32_t ta = P.type[w.src];
        wimg_build(img, P.arena + P.off[w.src], ta, P.card[w.src], P.nruns[w.src]);
        __builtin_amdgcn_wave_barrier();
        uint32_t r[32];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const uint32_t b0 = (32u * lane + k) * 32u;  // first value of this word
            uint32_t m = 0;
            if (fhi >= b0 && flo < b0 + 32u) {
                const uint32_t lb = flo > b0 ? flo - b0 : 0u;
                const uint32_t nb = (fhi + 1u < b0 + 32u ? fhi + 1u : b0 + 32u) - b0;  // 1 .. 32
                m = (nb == 32u ? 0xFFFFFFFFu : (1u << nb) - 1u) & ~((1u << lb) - 1u);
            }
            r[k] = img[wown(lane, k)] ^ m;
            cnt += __popc(r[k]);
        }
        const uint32_t rc = wave_sum(cnt);
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
        const int ty = ta == T_RUN ? type_eff(rc, rn) : type_ba(rc);
#include "rhip_wemit.inc"
        if (lane == 0) O.meta[j] = pack_meta(rc ? (uint32_t)ty : (uint32_t)T_ARRAY, rc, (rc && ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}
