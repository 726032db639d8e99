// rhip_values.h -- the integer-list side of the path: pools built on the device from sorted value lists
// (roaring_bitmap_of_ptr, roaring.h:88 / src/roaring.c:195-199 -> roaring_bitmap_add_many :134-193; 64-bit:
// roaring64_bitmap_of_ptr, roaring64.h:92) and pools decoded back into value lists
// (roaring_bitmap_to_uint32_array, roaring.h:571 / src/roaring.c:1510-1512 -> ra_to_uint32_array;
// roaring64_bitmap_to_uint64_array, roaring64.h:768).
#pragma once
#include "rhip_common.h"
#include "rhip_runs.h"

// ------------------------------------------------------------------ values -> pool
// Bitmap i = vals[offs[i], offs[i+1]), strictly increasing.  A container starts where the bitmap starts or the
// high bits (v >> 16) change.  One thread per value.
template <class V>
__global__ __launch_bounds__(256) void k_val_flags(const V* __restrict__ vals, const u64* __restrict__ offs,
                                                   uint32_t n_bitmaps, u64 total, uint32_t* __restrict__ flag,
                                                   uint32_t* status) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    u64 lo = 0, hi = n_bitmaps;  // last bitmap whose range starts at or before idx (skips empty bitmaps)
    while (lo + 1 < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (offs[mid] <= idx) lo = mid;
        else hi = mid;
    }
    const V v = vals[idx];
    bool start = true;
    if (idx != offs[lo]) {
        const V p = vals[idx - 1];
        if (v <= p) atomicMin(status, (uint32_t)lo);  // not strictly increasing
        start = (v >> 16) != (p >> 16);
    }
    flag[idx] = start ? 1u : 0u;
}
template <class V>
__global__ __launch_bounds__(256) void k_val_starts(const V* __restrict__ vals, u64 total, const uint32_t* __restrict__ flag,
                                                    const u64* __restrict__ cidx, u64* __restrict__ cstart,
                                                    u64* __restrict__ key) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx > total) return;
    if (idx == total) {
        cstart[cidx[total]] = total;  // sentinel: cstart[n_cont]
        return;
    }
    if (flag[idx]) {
        const u64 c = cidx[idx];
        cstart[c] = idx;
        key[c] = (u64)(vals[idx] >> 16);
    }
}
__global__ __launch_bounds__(256) void k_val_bm(const u64* __restrict__ offs, const u64* __restrict__ cidx,
                                                uint32_t n_bitmaps, u64* __restrict__ bm_start) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_bitmaps) bm_start[i] = cidx[offs[i]];
}
// array if cardinality <= 4096 else bitset (what add_many leaves behind; runs only appear through run_optimize)
__global__ __launch_bounds__(256) void k_val_meta(const u64* __restrict__ cstart, u64 n_cont, uint8_t* __restrict__ type,
                                                  uint32_t* __restrict__ card, uint32_t* __restrict__ nruns,
                                                  uint32_t* __restrict__ slot) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cont) return;
    const uint32_t cd = (uint32_t)(cstart[c + 1] - cstart[c]);
    const bool arr = cd <= 4096u;
    type[c] = arr ? T_ARRAY : T_BITSET;
    card[c] = cd;
    nruns[c] = 0;
    slot[c] = arr ? align16(2u * cd) : 8192u;
}
// One WAVE per container: arrays are the low halves packed two per dword; bitsets are scattered into the
// wave's LDS image and streamed out.
template <class V>
__global__ __launch_bounds__(256) void k_val_payload(const V* __restrict__ vals, const u64* __restrict__ cstart,
                                                     const uint32_t* __restrict__ card, const u64* __restrict__ off,
                                                     uint8_t* __restrict__ arena, u64 n_cont) {
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* img = img_all[threadIdx.x >> 6];
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 c = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_cont; c += nwaves) {
        const V* __restrict__ src = vals + cstart[c];
        const uint32_t cd = card[c];
        uint8_t* dst = arena + off[c];
        if (cd <= 4096u) {
            uint32_t* __restrict__ d32 = (uint32_t*)dst;
            for (uint32_t i = lane; 2u * i < cd; i += 64) {
                const uint32_t lo = (uint32_t)src[2u * i] & 0xFFFFu;
                const uint32_t hi = (2u * i + 1u < cd) ? ((uint32_t)src[2u * i + 1u] & 0xFFFFu) : 0u;
                d32[i] = lo | (hi << 16);
            }
        } else {
            const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < cd; i += 64) {
                const uint32_t v = (uint32_t)src[i] & 0xFFFFu;
                atomicOr(&img[v >> 5], 1u << (v & 31));
            }
            __builtin_amdgcn_wave_barrier();
            uint4* __restrict__ po = (uint4*)dst;
#pragma unroll
            for (int i = 0; i < 8; ++i) po[i * 64 + lane] = ((const uint4*)img)[i * 64 + lane];
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------ pool -> values
// cpre = exclusive prefix of the container cardinalities in directory order = position of each container's first
// value in the decoded stream of the whole pool (bitmaps back to back).  One WAVE per container; arrays are
// widened directly, bitsets and runs are rasterised (wimg_build) and their bits enumerated, lane l owning the 32
// consecutive words [32 l, 32 l + 32).
template <class V>
__global__ __launch_bounds__(256) void k_to_values(PoolView P, u64 n_cont, const u64* __restrict__ cpre,
                                                   V* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* img = img_all[threadIdx.x >> 6];
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 c = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_cont; c += nwaves) {
        const uint32_t t = P.type[c], cd = P.card[c];
        const V hi = (V)(P.key[c] << 16);
        V* __restrict__ o = out + cpre[c];
        const uint8_t* __restrict__ p = P.arena + P.off[c];
        if (t == T_ARRAY) {
            const uint16_t* __restrict__ a = (const uint16_t*)p;
            for (uint32_t i = lane; i < cd; i += 64) o[i] = hi | (V)a[i];
            continue;
        }
        wimg_build(img, p, t, cd, P.nruns[c]);
        __builtin_amdgcn_wave_barrier();
        uint32_t r[32];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            r[k] = img[wown(lane, k)];
            cnt += __popc(r[k]);
        }
        uint32_t pos = wave_incl_scan(cnt) - cnt;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            uint32_t x = r[k];
            const uint32_t vbase = (32u * lane + k) * 32u;
            while (x) {
                o[pos++] = hi | (V)(vbase + (__ffs((int)x) - 1));
                x &= x - 1;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
