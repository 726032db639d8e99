// rhip_block.h -- 256-thread-workgroup LDS image machinery used by the many-way aggregation kernels
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ LDS bitset machinery
// A 65536-bit container image in LDS is uint32_t[2048]; thread t of a 256-thread workgroup
// owns words [8t, 8t+8) (two ds_read_b128 / ds_write_b128).
struct BlockScratch {
    uint32_t wsum[8];   // per-wave partials
    uint32_t wsum2[8];
};

__device__ __forceinline__ void lds_zero(uint32_t* dst) {
    uint4 z = make_uint4(0, 0, 0, 0);
    ((uint4*)dst)[2 * threadIdx.x] = z;
    ((uint4*)dst)[2 * threadIdx.x + 1] = z;
}

// exclusive prefix sum over the 256 threads of the block; *total gets the block total
__device__ __forceinline__ uint32_t blk_exscan(uint32_t v, uint32_t* wsum, uint32_t* total) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        uint32_t s = wsum[w];
        if (w < wave) off += s;
        tot += s;
    }
    *total = tot;
    return off + inc - v;
}
__device__ __forceinline__ uint32_t blk_sum(uint32_t v, uint32_t* wsum) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    return wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Emit the LDS image `img` (result words also in r[8]) as a container of type ty into the
// candidate slot.  stage is an 8 KiB LDS buffer for coalesced output of arrays / runs
// (K5: bitset -> sorted u16 list by per-thread popcount + block prefix sum).
__device__ void lds_emit(const uint32_t* img, const uint32_t r[8], int ty, uint32_t rc, uint32_t rn,
                         uint16_t* stage, uint8_t* out, BlockScratch* sc) {
    const uint32_t tid = threadIdx.x;
    if (ty == T_BITSET) {
        uint4* __restrict__ po = (uint4*)out;
        po[2 * tid] = make_uint4(r[0], r[1], r[2], r[3]);
        po[2 * tid + 1] = make_uint4(r[4], r[5], r[6], r[7]);
        return;
    }
    uint32_t nbytes;
    if (ty == T_ARRAY) {
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) cnt += __popc(r[k]);
        uint32_t tot;
        uint32_t base = blk_exscan(cnt, sc->wsum, &tot);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t x = r[k];
            const uint32_t vbase = (8u * tid + k) * 32u;
            while (x) {
                stage[base++] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                x &= x - 1;
            }
        }
        nbytes = 2u * rc;
    } else {
        // run extraction: starts = set bits whose predecessor is clear, ends = set bits whose
        // successor is clear; the k-th start pairs with the k-th end.
        const uint32_t prev_msb = tid ? (img[8 * tid - 1] >> 31) : 0u;
        const uint32_t next_lsb = tid < 255 ? (img[8 * tid + 8] & 1u) : 0u;
        uint32_t S[8], E[8], ns = 0, ne = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t pm = k ? (r[k - 1] >> 31) : prev_msb;
            uint32_t nl = k < 7 ? (r[k + 1] & 1u) : next_lsb;
            S[k] = r[k] & ~((r[k] << 1) | pm);
            E[k] = r[k] & ~((r[k] >> 1) | (nl << 31));
            ns += __popc(S[k]);
            ne += __popc(E[k]);
        }
        uint32_t tot;
        uint32_t bs = blk_exscan(ns, sc->wsum, &tot);
        uint32_t be = blk_exscan(ne, sc->wsum2, &tot);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t x = S[k];
            const uint32_t vbase = (8u * tid + k) * 32u;
            while (x) {
                stage[2 * bs] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                bs++;
                x &= x - 1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t x = E[k];
            const uint32_t vbase = (8u * tid + k) * 32u;
            while (x) {
                uint32_t e = vbase + (__ffs((int)x) - 1);
                stage[2 * be + 1] = (uint16_t)(e - stage[2 * be]);
                be++;
                x &= x - 1;
            }
        }
        nbytes = 4u * rn;
    }
    __syncthreads();
    const uint32_t n16 = (nbytes + 15u) >> 4;
    uint4* __restrict__ po = (uint4*)out;
    for (uint32_t i = tid; i < n16; i += 256) po[i] = ((const uint4*)stage)[i];
}
