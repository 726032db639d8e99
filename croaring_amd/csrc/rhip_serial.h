// rhip_serial.h -- device-side bulk serialization (output half): the portable images
// (RoaringFormatSpec, ra_portable_serialize src/roaring_array.c:469-531) of many bitmaps of a 32-bit pool
// are assembled in HBM back to back and leave the device in ONE copy, instead of one payload download and
// one host-side assembly per bitmap.
#pragma once
#include "rhip_common.h"

// portable header size (ra_portable_header_size, roaring_array.c:445-456)
__device__ __forceinline__ uint32_t ser_header_bytes(uint32_t n, bool hasrun) {
    if (hasrun) return n < 4u ? 4u + ((n + 7u) >> 3) + 4u * n : 4u + ((n + 7u) >> 3) + 8u * n;
    return 8u + 8u * n;
}
__device__ __forceinline__ void st_le16(uint8_t* p, uint32_t v) {  // destinations have no alignment guarantee
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}
__device__ __forceinline__ void st_le32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

// A SEGMENT is the container range [seg_c0[i], seg_c1[i]) that becomes one 32-bit portable image: a whole bitmap of
// a 32-bit pool, or one high-32 bucket of a 64-bit bitmap (roaring64 images are a bucket count followed by
// (u32 high, 32-bit image) pairs, roaring64.c:2262-2393).
// One WAVE per segment: serialized size and whether the run cookie is needed
// (ra_portable_size_in_bytes, roaring_array.c:458-466).
__global__ __launch_bounds__(256) void k_ser_size(PoolView P, const u64* __restrict__ seg_c0, const u64* __restrict__ seg_c1,
                                                  uint32_t n_sel, uint32_t* __restrict__ size,
                                                  uint8_t* __restrict__ hasrun) {
    const uint32_t lane = lane_id();
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_sel) return;  // wave-uniform
    const u64 c0 = seg_c0[i], c1 = seg_c1[i];
    u64 bytes = 0;
    uint32_t anyrun = 0;
    for (u64 c = c0 + lane; c < c1; c += 64) {
        const uint8_t t = P.type[c];
        bytes += payload_bytes(t, P.card[c], P.nruns[c]) + (t == T_RUN ? 2u : 0u);
        anyrun |= t == T_RUN;
    }
    bytes = wave_sum64(bytes);
    const bool hr = __ballot(anyrun) != 0;
    if (lane == 0) {
        size[i] = ser_header_bytes((uint32_t)(c1 - c0), hr) + (uint32_t)bytes;
        hasrun[i] = hr ? 1 : 0;
    }
}

// One WAVE per segment: cookie, run flags, descriptive header, offset header; the destination of every container
// payload (absolute byte offset in the blob) goes to dst[] for the copy kernel.  The image of segment i starts at
// boff[i] (+ add[i], the bytes of the 64-bit framing in front of it, when add != NULL).
__global__ __launch_bounds__(256) void k_ser_header(PoolView P, const u64* __restrict__ seg_c0,
                                                    const u64* __restrict__ seg_c1, uint32_t n_sel,
                                                    const u64* __restrict__ boff, const u64* __restrict__ add,
                                                    const uint8_t* __restrict__ hasrun, uint8_t* __restrict__ blob,
                                                    u64* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_sel) return;  // wave-uniform
    const u64 c0 = seg_c0[i], c1 = seg_c1[i];
    const uint32_t n = (uint32_t)(c1 - c0);
    const bool hr = hasrun[i] != 0;
    const u64 image0 = boff[i] + (add ? add[i] : 0);
    uint8_t* base = blob + image0;
    uint8_t* p = base;
    if (hr) {
        if (lane == 0) st_le32(p, 12347u | ((n - 1u) << 16));
        p += 4;
        const uint32_t nb = (n + 7u) >> 3;
        for (uint32_t k = lane; k < nb; k += 64) {  // byte k = run flags of containers 8k .. 8k+7
            uint32_t bits = 0;
            for (uint32_t h = 0; h < 8; ++h)
                if (8u * k + h < n && P.type[c0 + 8u * k + h] == T_RUN) bits |= 1u << h;
            p[k] = (uint8_t)bits;
        }
        p += nb;
    } else {
        if (lane == 0) {
            st_le32(p, 12346u);
            st_le32(p + 4, n);
        }
        p += 8;
    }
    uint8_t* desc = p;
    p += 4u * n;
    const bool with_offsets = !hr || n >= 4u;
    uint8_t* offh = p;
    if (with_offsets) p += 4u * n;
    u64 run = (u64)(p - base);  // offset of the first payload inside this bitmap's image
    for (uint32_t j0 = 0; j0 < n; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t sz = 0;
        uint8_t t = 0;
        if (j < n) {
            const u64 c = c0 + j;
            t = P.type[c];
            const uint32_t cd = P.card[c];
            sz = payload_bytes(t, cd, P.nruns[c]) + (t == T_RUN ? 2u : 0u);
            st_le16(desc + 4u * j, (uint32_t)(P.key[c] & 0xFFFFu));
            st_le16(desc + 4u * j + 2, cd - 1u);
        }
        const uint32_t inc = wave_incl_scan(sz);
        if (j < n) {
            const u64 o = run + inc - sz;
            if (with_offsets) st_le32(offh + 4u * j, (uint32_t)o);
            u64 d = image0 + o;
            if (t == T_RUN) {
                st_le16(blob + d, P.nruns[c0 + j]);
                d += 2;
            }
            dst[c0 + j] = d;
        }
        run += __shfl(inc, 63);
    }
}

// Copy B payload bytes from a 16-byte aligned source to an arbitrarily aligned destination with aligned
// dword stores: destination dword j holds payload bytes [4j - d, 4j - d + 4), assembled from two source dwords.
__device__ __forceinline__ void wave_copy_unaligned(uint8_t* dst, const uint8_t* __restrict__ src, uint32_t B,
                                                    uint32_t lane) {
    const uint32_t d = (uint32_t)((uintptr_t)dst & 3u);
    const uint32_t* __restrict__ S = (const uint32_t*)src;
    if (d == 0) {
        uint32_t* D = (uint32_t*)dst;
        const uint32_t nd = B >> 2;
        for (uint32_t k = lane; k < nd; k += 64) D[k] = S[k];
        for (uint32_t k = (nd << 2) + lane; k < B; k += 64) dst[k] = src[k];
        return;
    }
    const uint32_t head = (4u - d) < B ? (4u - d) : B;
    if (lane < head) dst[lane] = src[lane];
    const uint32_t nfull = (B + d >= 4u) ? ((B + d) >> 2) - 1u : 0u;  // aligned dwords 1 .. nfull are complete
    uint32_t* D = (uint32_t*)(dst - d);
    for (uint32_t j = 1u + lane; j <= nfull; j += 64) {
        const u64 two = ((u64)S[j] << 32) | (u64)S[j - 1];
        D[j] = (uint32_t)(two >> (8u * (4u - d)));
    }
    for (uint32_t k = 4u * (nfull + 1u) - d + lane; k < B; k += 64) dst[k] = src[k];
}

// One WORKGROUP per segment, its waves striding over the segment's containers.
__global__ __launch_bounds__(256) void k_ser_copy(PoolView P, const u64* __restrict__ seg_c0, const u64* __restrict__ seg_c1,
                                                  uint32_t n_sel, const u64* __restrict__ dst,
                                                  uint8_t* __restrict__ blob) {
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (uint32_t i = blockIdx.x; i < n_sel; i += gridDim.x) {
        const u64 c0 = seg_c0[i], c1 = seg_c1[i];
        for (u64 c = c0 + wave; c < c1; c += nw)
            wave_copy_unaligned(blob + dst[c], P.arena + P.off[c], payload_bytes(P.type[c], P.card[c], P.nruns[c]), lane);
    }
}
