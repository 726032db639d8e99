// rhip_grouped.h -- the image kernels of an X-GROUPED batch (round 4): k_filter_g (and / andnot / cardinality with an
// array operand), k_union_g (or / xor of two arrays through the image, bitset (op) array)
//
// The queue these kernels read is counting-sorted by the item's X container (rhip_common.h, XGroupView): a wave walks a
// chunk of consecutive items and rebuilds its image of X only when X changes.  In an all-pairs batch a container meets
// ~85 partners (weather_sept_85), so a chunk of 8-16 items shares one or two X: the part of k_filter / k_wave that
// rasterised the membership side for every item (55 % / 62 % of them) is paid once per run, and the operand
// bytes a pair re-reads from L2 drop from |X| + |Y| to ~|Y|.  The container algebra is the reference's, unchanged:
//   k_filter_g  intersect_vector16 / difference_uint16 (array_util.c:385-459), array_bitset_container_intersection /
//               _andnot (mixed_intersection.c:19-58, mixed_andnot.c:24-39); gate array.c:293-310
//   k_union_g   union_vector16 / xor_vector16 (array_util.c:1662-1740 and below) for two arrays that need the image,
//               array_bitset_container_union / _xor, bitset_array_container_andnot (mixed_union.c:22-31,
//               mixed_xor.c:23-39, mixed_andnot.c:54-72); result typing: decide_type()
#pragma once
#include "rhip_array.h"

// the streamed operand of an item (all by value: a reference to a work item would put it on the stack)
struct YSide { u64 off; uint32_t n; bool in_a; };
__device__ __forceinline__ YSide filt_y_side(int op, uint32_t types, uint32_t ca, uint32_t cb, u64 offa, u64 offb) {
    const bool ya = filt_y_is_a(op, types & 0xFF, types >> 8, ca, cb);
    return YSide{ya ? offa : offb, ya ? ca : cb, ya};
}
__device__ __forceinline__ YSide union_y_side(int op, uint32_t types, uint32_t ca, uint32_t cb, u64 offa, u64 offb) {
    const bool xa = union_x_is_a(op, types & 0xFF, types >> 8, ca, cb);
    return YSide{xa ? offb : offa, xa ? cb : ca, !xa};
}

// ------------------------------------------------------------------ grouped filter
// As filter_body (rhip_array.h), but the 8 KiB LDS image of X survives from item to item.
template <bool STAGED>
__device__ __forceinline__ void filter_g_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA,
                                              const uint8_t* __restrict__ arenaB, OutView O, const FatItem* __restrict__ q,
                                              const u64* __restrict__ xr /* group boundaries: [0] begin, [1] end of the filter group */,
                                              int kop, int cardmode, u64* pair_acc, uint32_t cmin) {
    const uint32_t lane = lane_id();
    uint32_t* img = lds + (threadIdx.x >> 6) * 2048u;
    uint16_t* ST = STAGED ? (uint16_t*)(lds + 4u * 2048u + (threadIdx.x >> 6) * FILTER_ST_WORDS) : nullptr;  // output window: 7 + 512 values
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(xr[1] - xr[0]);
    const uint32_t chunk = xg_chunk(n, nwaves, cmin);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6) * chunk;
    const uint32_t wend = w + chunk < n ? w + chunk : n;
    if (w >= wend) return;
    const uint8_t* cur_x = nullptr;  // the container whose image the wave holds
    // Two items ahead: the work item after next is fetched (scalar loads: w is wave-uniform) while the FIRST 512 values of
    // the next item's Y are already in flight -- item -> pointer -> payload is a chain of two dependent round trips that
    // a wave walking eight items in a row would otherwise wait out eight times.
    FatItem tnext = q[w], tnext2 = q[w + 1 < wend ? w + 1 : wend - 1];  // (clamped index: no conditional copy of an item)
    uint4 ycur = make_uint4(0, 0, 0, 0);
    {
        const YSide y = filt_y_side(item_op(kop, tnext.types), tnext.types, tnext.ca, tnext.cb, tnext.offa, tnext.offb);
        if (8 * lane < y.n) ycur = ((const uint4*)((y.in_a ? arenaA : arenaB) + y.off))[lane];
    }
    for (; w < wend; ++w) {
        const FatItem t = tnext;
        tnext = tnext2;
        tnext2 = q[w + 2 < wend ? w + 2 : wend - 1];
        uint4 ynext = make_uint4(0, 0, 0, 0);
        if (w + 1 < wend) {
            const YSide y = filt_y_side(item_op(kop, tnext.types), tnext.types, tnext.ca, tnext.cb, tnext.offa, tnext.offb);
            if (8 * lane < y.n) ynext = ((const uint4*)((y.in_a ? arenaA : arenaB) + y.off))[lane];
        }
        const int op = item_op(kop, t.types);
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        const uint32_t ca = t.ca, cb = t.cb;
        const bool y_is_a = filt_y_is_a(op, ta, tb, ca, cb);
        const uint8_t* yp = y_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint8_t* xp = y_is_a ? arenaB + t.offb : arenaA + t.offa;
        const uint32_t ny = y_is_a ? ca : cb, nx = y_is_a ? cb : ca;
        const bool x_bitset = (y_is_a ? tb : ta) == T_BITSET;
        const bool keep_present = op == OP_AND;
        const uint4* __restrict__ y4 = (const uint4*)yp;
        const uint4* __restrict__ x4 = (const uint4*)xp;
        const uint4 yfirst = ycur;  // first 512 values of Y: loaded while the previous item was worked on
        ycur = ynext;
        if (xp != cur_x) {
            cur_x = xp;
            if (x_bitset) {
                uint4 xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = x4[i * 64 + lane];
#pragma unroll
                for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = xv[i];
            } else {
                const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
                __builtin_amdgcn_wave_barrier();  // every lane's slice is zero before any lane scatters into it
                wave_scatter_or(img, x4, nx, lane);
            }
            __builtin_amdgcn_wave_barrier();
        }
        uint4* __restrict__ po4 = cardmode ? nullptr : (uint4*)(O.arena + t.offo);
        uint32_t run = 0, carry = 0;  // run: values already in global memory (a multiple of 8) -- cardinality mode: all
        uint4 q4 = yfirst;
        for (uint32_t base = 0; base < ny; base += 512) {
            const uint32_t i0 = base + 8 * lane;
            uint4 nxt = make_uint4(0, 0, 0, 0);  // the next step's 16 bytes: in flight while this step is tested
            if (i0 + 512 < ny) nxt = y4[(base >> 3) + 64 + lane];
            const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
            q4 = nxt;
            const uint32_t nval = i0 < ny ? (ny - i0 < 8u ? ny - i0 : 8u) : 0u;  // values this lane holds
            uint32_t vals[8];
            uint32_t keepmask = 0;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                vals[h] = v;
                const uint32_t present = (img[v >> 5] >> (v & 31)) & 1u;
                keepmask |= (present == (uint32_t)keep_present ? 1u : 0u) << h;
            }
            keepmask &= (1u << nval) - 1u;
            const uint32_t cnt = __popc(keepmask);
            const uint32_t inc = wave_incl_scan(cnt);
            const uint32_t tot = wave_lane<63>(inc);
            if (cardmode) {
                run += tot;
                continue;
            }
            if (!STAGED || keep_present) {  // and: few values survive as a rule -- the window's bookkeeping costs more than their stores
                uint16_t* __restrict__ out = (uint16_t*)po4;
                uint32_t pos = run + inc - cnt;
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    if ((keepmask >> h) & 1u) out[pos++] = (uint16_t)vals[h];
                run += tot;
                continue;
            }
            uint32_t pos = carry + inc - cnt;
#pragma unroll
            for (int h = 0; h < 8; ++h)
                if ((keepmask >> h) & 1u) ST[pos++] = (uint16_t)vals[h];
            __builtin_amdgcn_wave_barrier();
            const uint32_t filled = carry + tot, nfull = filled >> 3;
            for (uint32_t j = lane; j < nfull; j += 64) out_store16(&po4[(run >> 3) + j], ((const uint4*)ST)[j]);
            carry = filled & 7u;
            uint16_t keep = 0;
            if (lane < carry) keep = ST[8u * nfull + lane];
            __builtin_amdgcn_wave_barrier();  // the window has been read
            if (lane < carry) ST[lane] = keep;
            run += 8u * nfull;
        }
        if (!cardmode) {
            __builtin_amdgcn_wave_barrier();
            if (carry && lane == 0) out_store16(&po4[run >> 3], ((const uint4*)ST)[0]);  // (the slot is padded to 16 bytes)
            run += carry;
        }
        if (cardmode) {
            if (lane == 0 && run) atomicAdd(&pair_acc[t.out], (u64)run);
        } else if (lane == 0) {
            O.meta[t.out] = pack_meta(T_ARRAY, run, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
template <bool STAGED>
__global__ __launch_bounds__(256) void k_filter_g(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                  OutView O, const FatItem* __restrict__ q, const u64* __restrict__ xr,
                                                  int kop, int cardmode, u64* pair_acc, uint32_t cmin) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[STAGED ? FILTER_LDS_WORDS : 8192];
    filter_g_body<STAGED>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, xr, kop, cardmode, pair_acc, cmin);
}

// ------------------------------------------------------------------ grouped union
// or / xor / bitset \ array with X held in REGISTERS as a 65536-bit image -- 32 words per lane in k_bb's layout (lane l
// holds the 16-byte groups i * 64 + l) -- for as long as consecutive items share it: a bitset by 8 coalesced loads, an
// array by zero + ds_or scatter into the wave's LDS image + 8 conflict-free 16-byte reads back.  Per item only Y (always
// an array) is rasterised: into the zeroed LDS image with plain, non-returning ds_or; every lane reads ITS groups back,
// combines them with X, popcounts (the cardinality: no returning atomics).  A bitset result leaves from registers; an
// array result (rc <= 4096: decide_type) is extracted from the registers (wave_extract_groups).  Against k_wave this
// drops the X scatter per item, the returning atomics and the count table of the extraction; against k_ba the 8 KiB
// bitset load per item.
// Sorted array out of a 65536-bit image held in REGISTERS in k_bb's layout (lane l: the 16-byte groups i * 64 + l, i.e.
// words (i * 64 + l) * 4 + j), rc <= 4096 set bits.  Word order = (i, lane, j), so the output position of a lane's group
// i is the total of the groups i' < i plus a wave prefix over the lanes: eight prefix sums, two per packed wave scan --
// no count table in LDS, no second read of the image.  `img` (8 KiB, free) is only the staging buffer the sorted values
// are compacted into before they leave with coalesced 16-byte stores.
__device__ __forceinline__ void wave_extract_groups(const u32x4 (&r)[8], const uint32_t (&cnt)[8] /* set bits of r[i] */,
                                                    uint32_t* __restrict__ img, uint32_t lane, uint32_t rc,
                                                    uint8_t* __restrict__ outp) {
    uint32_t base[8];
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t p = cnt[2 * k] | (cnt[2 * k + 1] << 16);  // each <= 128; sums over the wave <= 8192
        const uint32_t inc = wave_incl_scan(p);
        const uint32_t tot = wave_lane<63>(inc), ex = inc - p;
        base[2 * k] = run + (ex & 0xFFFFu);
        run += tot & 0xFFFFu;
        base[2 * k + 1] = run + (ex >> 16);
        run += tot >> 16;
    }
    uint16_t* st16 = (uint16_t*)img;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t pos = base[i];
        const uint32_t wd[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x = wd[j];
            const uint32_t vbase = ((64u * i + lane) * 4u + j) * 32u;
            while (x) {
                st16[pos++] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                x &= x - 1;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t n16 = (2u * rc + 15u) >> 4;
    uint4* __restrict__ po = (uint4*)outp;
    for (uint32_t i = lane; i < n16; i += 64) out_store16(&po[i], ((const uint4*)img)[i]);
}

#ifndef RHIP_ABL_EXTRACT
#define RHIP_ABL_EXTRACT 0
#endif
template <int OP>
__device__ __forceinline__ void union_g_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA,
                                             const uint8_t* __restrict__ arenaB, OutView O, const FatItem* __restrict__ q,
                                             const u64* __restrict__ xr /* [0] begin of the filter group, [1] .. [2] the union group */,
                                             uint32_t cmin) {
    const uint32_t lane = lane_id();
    uint32_t* img = lds + (threadIdx.x >> 6) * 2048u;
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(xr[2] - xr[1]);
    q += (size_t)(xr[1] - xr[0]);
    const uint32_t chunk = xg_chunk(n, nwaves, cmin);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6) * chunk;
    const uint32_t wend = w + chunk < n ? w + chunk : n;
    if (w >= wend) return;
    const uint8_t* cur_x = nullptr;
    u32x4 vx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vx[i] = (u32x4)(0u);
    // two items ahead, as in filter_g_body: the next item's first 512 values of Y are in flight while this one is worked on
    FatItem tnext = q[w], tnext2 = q[w + 1 < wend ? w + 1 : wend - 1];  // (clamped index: no conditional copy of an item)
    uint4 ycur = make_uint4(0, 0, 0, 0);
    {
        const YSide y = union_y_side(item_op(OP, tnext.types), tnext.types, tnext.ca, tnext.cb, tnext.offa, tnext.offb);
        if (8 * lane < y.n) ycur = ((const uint4*)((y.in_a ? arenaA : arenaB) + y.off))[lane];
    }
    for (; w < wend; ++w) {
        const FatItem t = tnext;
        tnext = tnext2;
        tnext2 = q[w + 2 < wend ? w + 2 : wend - 1];
        uint4 ynext = make_uint4(0, 0, 0, 0);
        if (w + 1 < wend) {
            const YSide y = union_y_side(item_op(OP, tnext.types), tnext.types, tnext.ca, tnext.cb, tnext.offa, tnext.offb);
            if (8 * lane < y.n) ynext = ((const uint4*)((y.in_a ? arenaA : arenaB) + y.off))[lane];
        }
        const int op = item_op(OP, t.types);
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        const uint32_t ca = t.ca, cb = t.cb;
        const bool x_is_a = union_x_is_a(op, ta, tb, ca, cb);
        const uint8_t tx = x_is_a ? ta : tb;
        const uint32_t cx = x_is_a ? ca : cb, cy = x_is_a ? cb : ca;
        const uint8_t* xp = x_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint4* __restrict__ y4 = (const uint4*)(x_is_a ? arenaB + t.offb : arenaA + t.offa);
        const uint4 yfirst = ycur;
        ycur = ynext;
        const uint4 z = make_uint4(0, 0, 0, 0);
        if (xp != cur_x) {
            cur_x = xp;
            if (tx == T_BITSET) {
                const u32x4* __restrict__ px = (const u32x4*)xp;
#pragma unroll
                for (int i = 0; i < 8; ++i) vx[i] = px[i * 64 + lane];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
                __builtin_amdgcn_wave_barrier();
                wave_scatter_or(img, (const uint4*)xp, cx, lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint4 m4 = ((const uint4*)img)[i * 64 + lane];
                    vx[i].x = m4.x; vx[i].y = m4.y; vx[i].z = m4.z; vx[i].w = m4.w;
                }
                __builtin_amdgcn_wave_barrier();  // X has been read: the image is Y's from here on
            }
        }
        // ---- Y into the zeroed image
#pragma unroll
        for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
        __builtin_amdgcn_wave_barrier();  // every lane's slice is zero before any lane scatters into it
        {
            uint4 q4 = yfirst;
            for (uint32_t i = lane; 8 * i < cy; i += 64) {
                uint4 nxt = make_uint4(0, 0, 0, 0);
                if (8 * (i + 64) < cy) nxt = y4[i + 64];
                const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
                const uint32_t nval = cy - 8 * i < 8u ? cy - 8 * i : 8u;
#pragma unroll
                for (int h = 0; h < 8; ++h) {  // (a lane's values past the end OR a zero in: no branch per value)
                    const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                    atomicOr(&img[v >> 5], (uint32_t)h < nval ? 1u << (v & 31) : 0u);
                }
                q4 = nxt;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- combine, popcount
        u32x4 r[8];
        uint32_t cnt[8], tot = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 m4 = ((const uint4*)img)[i * 64 + lane];
            u32x4 m;
            m.x = m4.x; m.y = m4.y; m.z = m4.z; m.w = m4.w;
            r[i] = vop_any<OP>(op, vx[i], m);  // (andnot: X = a always; or / xor are symmetric)
            // (counted as two 64-bit halves: with 32 per-word popcounts here the compiler recognises the extraction's
            // `while (x) x &= x - 1` loops as counted loops, re-uses these as their trip counts and keeps all 32 alive)
            cnt[i] = (uint32_t)__popcll((u64)r[i].x | ((u64)r[i].y << 32)) + (uint32_t)__popcll((u64)r[i].z | ((u64)r[i].w << 32));
            tot += cnt[i];
        }
        const uint32_t rc = wave_sum(tot);
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, ta, tb, ca, cb, false, false, rc, 0);
        uint8_t* outp = O.arena + t.offo;
        if (rc && ty == T_BITSET) {
            u32x4* __restrict__ po = (u32x4*)outp;
#pragma unroll
            for (int i = 0; i < 8; ++i) out_store16(&po[i * 64 + lane], r[i]);
        } else if (rc) {
            __builtin_amdgcn_wave_barrier();  // every lane has read its groups of Y's image: it is the staging buffer now
#if RHIP_ABL_EXTRACT  /* ablation builds only (WRONG results): what the bit extraction of array results costs */
            if (rc == 0xFFFFFFFFu) wave_extract_groups(r, cnt, img, lane, rc, outp);
            else {
                const uint32_t n16 = (2u * rc + 15u) >> 4;
                uint4* __restrict__ po = (uint4*)outp;
                for (uint32_t i = lane; i < n16; i += 64) out_store16(&po[i], ((const uint4*)img)[i]);
            }
#else
            wave_extract_groups(r, cnt, img, lane, rc, outp);
#endif
        }
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, 0);
        __builtin_amdgcn_wave_barrier();  // the image is zeroed again by the next item
    }
}
// (-DRHIP_UNION_WAVES=5: variant build, five waves per SIMD if the compiler fits the kernel into 96 VGPRs)
#ifndef RHIP_UNION_WAVES
#define RHIP_UNION_WAVES 4
#endif
template <int OP>
__global__ __launch_bounds__(256, RHIP_UNION_WAVES) void k_union_g(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                 OutView O, const FatItem* __restrict__ q, const u64* __restrict__ xr,
                                                 uint32_t cmin) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
    union_g_body<OP>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, xr, cmin);
}
