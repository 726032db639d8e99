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

// Rasterise container c of pool V into the LDS image dst (K6 / K7 of SURVEY §2.2):
//   bitset: straight 16-byte copy;
//   array : zero + ds_or_b32 scatter (bitset_set_list, bitset_util.c:978-1141);
//   run   : zero + toggle bits at every run start and end+1, then an inclusive prefix-XOR over
//           the 65536 bits (in-word shifts + a cross-word parity carry obtained from one
//           ballot per wave) -- O(1) work per word regardless of run lengths
//           (replaces the serial bitset_set_lenrange loop, bitset_util.h:41-161).
__device__ void lds_load(uint32_t* dst, const PoolView& V, uint32_t c, BlockScratch* sc) {
    const uint32_t tid = threadIdx.x;
    const uint8_t ty = V.type[c];
    const uint8_t* p = V.arena + V.off[c];
    if (ty == T_BITSET) {
        const uint4* __restrict__ g = (const uint4*)p;
        uint4 x0 = g[2 * tid], x1 = g[2 * tid + 1];
        ((uint4*)dst)[2 * tid] = x0;
        ((uint4*)dst)[2 * tid + 1] = x1;
        __syncthreads();
        return;
    }
    lds_zero(dst);
    __syncthreads();
    if (ty == T_ARRAY) {
        const uint32_t n = V.card[c];
        const uint32_t* __restrict__ a2 = (const uint32_t*)p;  // two values per dword, slot is 16-byte padded
        for (uint32_t i = tid; 2 * i < n; i += 256) {
            uint32_t v2 = a2[i];
            uint32_t v = v2 & 0xFFFFu;
            atomicOr(&dst[v >> 5], 1u << (v & 31));
            if (2 * i + 1 < n) {
                v = v2 >> 16;
                atomicOr(&dst[v >> 5], 1u << (v & 31));
            }
        }
        __syncthreads();
        return;
    }
    {
        const uint32_t n = V.nruns[c];
        const uint32_t* __restrict__ r = (const uint32_t*)p;  // {u16 value, u16 length} little-endian
        for (uint32_t i = tid; i < n; i += 256) {
            uint32_t rl = r[i];
            uint32_t s = rl & 0xFFFFu, e1 = s + (rl >> 16) + 1u;
            atomicXor(&dst[s >> 5], 1u << (s & 31));
            if (e1 < 65536u) atomicXor(&dst[e1 >> 5], 1u << (e1 & 31));
        }
        __syncthreads();
        uint4 x0 = ((uint4*)dst)[2 * tid], x1 = ((uint4*)dst)[2 * tid + 1];
        uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        uint32_t par = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) par ^= __popc(w[k]) & 1u;
        const u64 m = __ballot(par != 0);
        uint32_t carry = mbcnt(m) & 1u;
        if (lane_id() == 0) sc->wsum[tid >> 6] = (uint32_t)__popcll(m) & 1u;
        __syncthreads();
        for (uint32_t wv = 0; wv < (tid >> 6); ++wv) carry ^= sc->wsum[wv];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t x = w[k], y = x;
            y ^= y << 1; y ^= y << 2; y ^= y << 4; y ^= y << 8; y ^= y << 16;
            w[k] = carry ? ~y : y;
            carry ^= __popc(x) & 1u;
        }
        ((uint4*)dst)[2 * tid] = make_uint4(w[0], w[1], w[2], w[3]);
        ((uint4*)dst)[2 * tid + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        __syncthreads();
    }
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
