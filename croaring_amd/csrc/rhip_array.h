// rhip_array.h -- wave-per-pair kernels for pairs with an array operand: k_filter (and / andnot), k_wave (or / xor / bitset \\ array)
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ short array probes (K9, K12 for short arrays)
// and / andnot / and_cardinality where the streamed array Y has at most PROBE_MAX values (four per lane): membership
// of every value is decided straight from global / L2 -- one gathered dword when X is a bitset, a two-level search
// when X is a sorted array: lane k first holds the pivot X[k * step] (step = ceil(nx / 64)), each lane locates its
// pivot interval with six register shuffles, then at most log2(step) <= 6 dependent probes inside the interval.
// No LDS, few registers: full wave occupancy, which is what hides the probe latency.  This is the GPU form of the
// reference's galloping intersection for skewed cardinalities (intersect_skewed_uint16 / advanceUntil,
// src/array_util.c:801-906, gate src/containers/array.c:293-310) and of array_bitset_container_intersection /
// array_bitset_container_andnot (mixed_intersection.c:19-46, mixed_andnot.c:24-39) for short arrays.
// The result is always an array (containers.h:741-746, 1799-1803).
__device__ __forceinline__ bool probe_sorted(const uint16_t* __restrict__ x16, uint32_t nx, uint32_t step, uint32_t piv,
                                             uint32_t v) {
    // pos = number of pivots <= v (pivot k sits in lane k; lanes past the last pivot hold 0xFFFFFFFF)
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t b = 32; b >= 1; b >>= 1) {
        const uint32_t pv = __shfl(piv, (int)(pos + b - 1));
        if (pv <= v) pos += b;
    }
    {
        const uint32_t pv = __shfl(piv, 63);
        if (pos == 63u && pv <= v) pos = 64u;
    }
    const uint32_t k = pos ? pos - 1u : 0u;
    uint32_t last = __shfl(piv, (int)k);      // X[idx], idx = largest index known to hold a value <= v
    uint32_t idx = k * step;
    const uint32_t end = (idx + step < nx) ? idx + step : nx;
    for (uint32_t b = 32; b >= 1; b >>= 1) {   // uniform trip count; intervals hold at most 64 values
        if (b < step) {
            const uint32_t c = idx + b;
            if (pos && c < end) {
                const uint32_t xv = x16[c];
                if (xv <= v) { idx = c; last = xv; }
            }
        }
    }
    return pos != 0u && last == v;
}
__device__ __forceinline__ void probe_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                               OutView O, const FatItem* __restrict__ q,
                                               const u64* __restrict__ qrange, int kop, int cardmode, u64* pair_acc) {
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
    FatItem tnext;
    if (w < n) tnext = q[w];
    for (; w < n; w += nwaves) {
        const FatItem t = tnext;
        if (w + nwaves < n) tnext = q[w + nwaves];
        const int op = item_op(kop, t.types);
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        bool y_is_a = true;
        if (op == OP_AND) y_is_a = (ta == T_ARRAY) && (tb != T_ARRAY || t.ca <= t.cb);
        const uint8_t* yp = y_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint8_t* xp = y_is_a ? arenaB + t.offb : arenaA + t.offa;
        const uint32_t ny = y_is_a ? t.ca : t.cb, nx = y_is_a ? t.cb : t.ca;
        const bool x_bitset = (y_is_a ? tb : ta) == T_BITSET;
        const bool keep_present = op == OP_AND;
        const uint16_t* __restrict__ y16 = (const uint16_t*)yp;
        constexpr uint32_t R = PROBE_MAX / 64u;  // values of Y per lane
        uint32_t v[R];
        bool pr[R];
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) v[r] = 64u * r + lane < ny ? y16[64u * r + lane] : 0u;
        if (x_bitset) {
            const uint32_t* __restrict__ xw = (const uint32_t*)xp;
            uint32_t wd[R];
#pragma unroll
            for (uint32_t r = 0; r < R; ++r) wd[r] = xw[v[r] >> 5];
#pragma unroll
            for (uint32_t r = 0; r < R; ++r) pr[r] = (wd[r] >> (v[r] & 31)) & 1u;
        } else {
            const uint16_t* __restrict__ x16 = (const uint16_t*)xp;
            const uint32_t step = (nx + 63u) >> 6;
            const uint32_t pi = lane * step;
            const uint32_t piv = pi < nx ? (uint32_t)x16[pi] : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t r = 0; r < R; ++r)  // (ny is wave-uniform: whole rounds are skipped)
                pr[r] = ny > 64u * r ? probe_sorted(x16, nx, step, piv, v[r]) : false;
        }
        bool kp[R];
        u64 m[R];
        uint32_t run = 0, before[R];
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) {
            kp[r] = 64u * r + lane < ny && pr[r] == keep_present;
            m[r] = __ballot(kp[r]);
            before[r] = run;
            run += (uint32_t)__popcll(m[r]);
        }
        if (cardmode) {
            if (lane == 0 && run) atomicAdd(&pair_acc[t.out], (u64)run);
        } else {
            uint16_t* __restrict__ out = (uint16_t*)(O.arena + t.offo);
#pragma unroll
            for (uint32_t r = 0; r < R; ++r)
                if (kp[r]) out[before[r] + mbcnt(m[r])] = (uint16_t)v[r];
            if (lane == 0) O.meta[t.out] = pack_meta(T_ARRAY, run, 0);
        }
    }
}
__global__ __launch_bounds__(256) void k_probe(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                               OutView O, const FatItem* __restrict__ q,
                                               const u64* __restrict__ qrange, int kop, int cardmode, u64* pair_acc) {
    uint32_t* lds = nullptr;
    probe_body(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, kop, cardmode, pair_acc);
}


__device__ __forceinline__ void wave_scatter_or(uint32_t* img, const uint4* __restrict__ x4, uint32_t nx, uint32_t lane) {
    const uint32_t nfull = nx >> 3;   // whole 8-value groups: no per-value bound checks
    for (uint32_t i = lane; i < nfull; i += 64) {
        const uint4 q4 = x4[i];
        const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
            atomicOr(&img[v >> 5], 1u << (v & 31));
        }
    }
    if ((nx & 7u) && lane == (nfull & 63u)) {  // the ragged tail group, one lane
        const uint4 q4 = x4[nfull];
        const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            if (h < (int)(nx & 7u)) {
                const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                atomicOr(&img[v >> 5], 1u << (v & 31));
            }
        }
    }
}

// ------------------------------------------------------------------ short array merged into a long one (K10, K11)
// or / xor of two arrays when the smaller one (Y, <= USMALL_MAX values: four per lane; the per-rank byte counters below
// hold up to 255) is short and |X| + |Y| <= 4096,
// so that the result is an array by the reference's rule before anything is computed (mixed_union.c:162-191,
// mixed_xor.c:196-219; half of the array pairs of weather_sept_85).  The result is X with a few values inserted
// (and, under xor, a few removed), so it is built BY RANK instead of through a 65536-bit image:
//   * every y finds its rank r(y) in X (number of X values below it) and whether X holds it: the two-level pivot
//     search of k_probe, straight from global / L2;
//   * new values (y not in X) bump a per-rank counter in LDS, values in both (xor) set a "deleted" bit per X index;
//   * X is streamed in 16-byte groups: X[i] goes to  i - (deleted before i) + (new values with rank <= i), the two
//     counts coming from one packed wave prefix sum over the groups plus a running count inside the group;
//   * a new y goes to  r(y) - (deleted before r(y)) + (new values before it).
// ~350 wave instructions for the median pair (26 x 1268 values) against ~1000 through the image kernel, whose cost
// is the 2048-word image however few values there are.  union_vector16 / xor_vector16 (array_util.c) on the CPU.
__device__ __forceinline__ uint32_t byte_sum(uint32_t c) {
#ifdef RHIP_EMU
    return (c & 0xFFu) + ((c >> 8) & 0xFFu) + ((c >> 16) & 0xFFu) + (c >> 24);
#else
    return __builtin_amdgcn_sad_u8(c, 0u, 0u);
#endif
}
// rank of v in the sorted array x16[0..nx) (number of values < v) and whether it is present; pivots as in probe_sorted
__device__ __forceinline__ uint32_t probe_rank(const uint16_t* __restrict__ x16, uint32_t nx, uint32_t step, uint32_t piv,
                                               uint32_t v, bool* present) {
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t b = 32; b >= 1; b >>= 1) {
        const uint32_t pv = __shfl(piv, (int)(pos + b - 1));
        if (pv <= v) pos += b;
    }
    {
        const uint32_t pv = __shfl(piv, 63);
        if (pos == 63u && pv <= v) pos = 64u;
    }
    const uint32_t k = pos ? pos - 1u : 0u;
    uint32_t last = __shfl(piv, (int)k);
    uint32_t idx = k * step;
    const uint32_t end = (idx + step < nx) ? idx + step : nx;
    for (uint32_t b = 32; b >= 1; b >>= 1) {
        if (b < step) {
            const uint32_t c = idx + b;
            if (pos && c < end) {
                const uint32_t xv = x16[c];
                if (xv <= v) { idx = c; last = xv; }
            }
        }
    }
    *present = pos != 0u && last == v;
    return pos == 0u ? 0u : (last == v ? idx : idx + 1u);
}
// k_usmall's LDS per wave, words.  GP holds a count of at most USMALL_MAX = 255 per 8-index group.  As 16-bit counts
// (GP8 = false, the default) the workgroup takes 29 184 bytes, five per CU; as BYTES (GP8 = true, RHIP_USMALL_GP8=1) 27 136
// and a sixth fits.  Measured (round 5, same box, alternating, three passes): C5 `or` / `xor` the same either way, weather
// `or` 0.555 -> 0.57-0.60 ms with bytes -- more resident k_usmall workgroups take LDS and issue slots from k_union_g, that
// batch's critical kernel (the stream-priority experiment said the same).  Kept as a switch, not as the default.
static_assert(USMALL_MAX <= 255u, "k_usmall: deleted-before counts fit a byte");
constexpr uint32_t D_WORDS = 1032, DEL_WORDS = 136, ST_WORDS = 392;
template <bool GP8> struct UsmallLds {
    typedef typename std::conditional<GP8, uint8_t, uint16_t>::type gp_t;
    static constexpr uint32_t GP_WORDS = GP8 ? 136u : 264u;
    static constexpr uint32_t WAVE_WORDS = D_WORDS + DEL_WORDS + GP_WORDS + ST_WORDS;
    static constexpr uint32_t WORDS = 4 * WAVE_WORDS;
};
constexpr uint32_t USMALL_LDS_WORDS = UsmallLds<false>::WORDS;
// OPC: the op known at compile time (a single-op batch: OP_OR drops the deleted-value arithmetic of xor from the X stream --
// three of its twelve instructions per value), or -1: read per item (multi-op batches, the merged launch of a light batch)
template <bool GP8, int OPC = -1>
__device__ __forceinline__ void usmall_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                OutView O, const FatItem* __restrict__ q,
                                                const u64* __restrict__ qrange, int kop) {
    // per wave: D = one byte counter per X index 0..nx (new values by rank), DEL = one bit per X index (xor: value
    // in both), GP = deleted-before count of every 8-index group
    // ST = output window of one step: <= 7 carried + 512 of X + 255 new values
    const uint32_t lane = lane_id();
    typedef typename UsmallLds<GP8>::gp_t usmall_gp_t;
    constexpr uint32_t GP_WORDS = UsmallLds<GP8>::GP_WORDS;
    uint32_t* D32 = lds + (threadIdx.x >> 6) * UsmallLds<GP8>::WAVE_WORDS;
    uint32_t* DEL = D32 + D_WORDS;
    usmall_gp_t* GP = (usmall_gp_t*)(DEL + DEL_WORDS);
    uint16_t* ST = (uint16_t*)(DEL + DEL_WORDS + GP_WORDS);
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
    FatItem tnext;
    if (w < n) tnext = q[w];
    PH_BEGIN();
    for (; w < n; w += nwaves) {
        const FatItem t = tnext;
        if (w + nwaves < n) tnext = q[w + nwaves];
        const int op = OPC >= 0 ? OPC : item_op(kop, t.types);
        const bool x_is_a = t.ca >= t.cb;
        const uint8_t* xp = x_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint8_t* yp = x_is_a ? arenaB + t.offb : arenaA + t.offa;
        const uint32_t nx = x_is_a ? t.ca : t.cb, ny = x_is_a ? t.cb : t.ca;
        const uint16_t* __restrict__ x16 = (const uint16_t*)xp;
        const uint16_t* __restrict__ y16 = (const uint16_t*)yp;
        const uint4* __restrict__ x4 = (const uint4*)xp;
        // loads first: the y values of this lane, its pivot of X
        constexpr uint32_t R = (USMALL_MAX + 63u) / 64u;
        bool ok[R];
        uint32_t v[R];
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) {
            ok[r] = 64u * r + lane < ny;
            v[r] = ok[r] ? y16[64u * r + lane] : 0u;
        }
        const uint32_t step = (nx + 63u) >> 6;
        const uint32_t pi = lane * step;
        const uint32_t piv = pi < nx ? (uint32_t)x16[pi] : 0xFFFFFFFFu;
        const uint32_t nd = 2u * ((nx + 8u) >> 3) + 2u, ndl = (nx + 32u) >> 5;  // whole 8-index groups, index nx included
        for (uint32_t i = lane; i < nd; i += 64) D32[i] = 0u;
        if (op == OP_XOR)
            for (uint32_t i = lane; i < ndl; i += 64) DEL[i] = 0u;
        __builtin_amdgcn_wave_barrier();
        PH(0);
        bool isnew[R];
        uint32_t rk[R], before[R];
        u64 m[R];
        uint32_t nnew = 0, ndel = 0;
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) {  // (ny is wave-uniform: whole rounds are skipped)
            bool pr = false;
            rk[r] = ny > 64u * r ? probe_rank(x16, nx, step, piv, v[r], &pr) : 0u;
            isnew[r] = ok[r] && !pr;
            const bool del = op == OP_XOR && ok[r] && pr;
            if (isnew[r]) atomicAdd(&D32[rk[r] >> 2], 1u << (8u * (rk[r] & 3u)));
            if (del) atomicOr(&DEL[rk[r] >> 5], 1u << (rk[r] & 31u));
            m[r] = __ballot(isnew[r]);
            before[r] = nnew;
            nnew += (uint32_t)__popcll(m[r]);
            ndel += (uint32_t)__popcll(__ballot(del));
        }
        __builtin_amdgcn_wave_barrier();
        PH(1);
        uint4* __restrict__ po4 = (uint4*)(O.arena + t.offo);
        // ---- X, 16 bytes (8 values) per lane and step.  The outputs of a step -- its X values and the new values of Y
        // whose rank falls into it -- are one contiguous stretch of the result: they are placed in the LDS window ST
        // (ST[0] = result index out_base, a multiple of 8) and leave as whole 16-byte groups, the incomplete last
        // group staying behind for the next step.  (Written straight to global they were 2-byte stores 16 bytes
        // apart, eight instructions per step for the same kilobyte: 40 % of the kernel.)
        const uint32_t ngroups = (nx + 7u) >> 3;
        auto del_below = [&](uint32_t r) -> uint32_t {
            if (op != OP_XOR) return 0u;
            const uint32_t g = r >> 3;
            if (g >= ngroups) return ndel;
            const uint32_t byte = (DEL[g >> 2] >> (8u * (g & 3u))) & 0xFFu;
            return (uint32_t)GP[g] + (uint32_t)__popc(byte & ((1u << (r & 7u)) - 1u));
        };
        uint32_t run_new = 0, run_del = 0, out_base = 0, carry = 0;
        for (uint32_t g0 = 0; g0 < ngroups; g0 += 64) {
            const uint32_t g = g0 + lane;
            const bool act = g < ngroups;
            uint4 xq = make_uint4(0, 0, 0, 0);
            uint32_t c0 = 0, c1 = 0, delb = 0;
            if (act) {
                xq = x4[g];
                c0 = D32[2u * g];
                c1 = D32[2u * g + 1u];
                if (op == OP_XOR) delb = (DEL[g >> 2] >> (8u * (g & 3u))) & 0xFFu;
            }
            const uint32_t packed = (byte_sum(c0) + byte_sum(c1)) | ((uint32_t)__popc(delb) << 16);
            const uint32_t inc = wave_incl_scan(packed);
            const uint32_t excl = inc - packed;
            uint32_t newc = run_new + (excl & 0xFFFFu), delc = run_del + (excl >> 16);
            if (op == OP_XOR && act) GP[g] = (usmall_gp_t)delc;
            const uint32_t tot = wave_lane<63>(inc);
            run_new += tot & 0xFFFFu;
            run_del += tot >> 16;
            const uint32_t nval = act ? (nx - 8u * g < 8u ? nx - 8u * g : 8u) : 0u;
            const uint32_t d[4] = {xq.x, xq.y, xq.z, xq.w};
            if (8u * (g0 + 64u) <= nx) {  // every lane holds a whole group (wave-uniform): no test per value
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    newc += ((h < 4 ? c0 : c1) >> (8 * (h & 3))) & 0xFFu;
                    const uint32_t dl = (delb >> h) & 1u;
                    if (!dl) ST[8u * g + h + newc - delc - out_base] = (uint16_t)((d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu);
                    delc += dl;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if ((uint32_t)h < nval) {
                        newc += ((h < 4 ? c0 : c1) >> (8 * (h & 3))) & 0xFFu;
                        const uint32_t dl = (delb >> h) & 1u;
                        if (!dl) ST[8u * g + h + newc - delc - out_base] = (uint16_t)((d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu);
                        delc += dl;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // GP of this step's groups is complete
            // the new values of Y whose rank is an X index of this step (the last step also takes rank == nx)
            const uint32_t rlo = 8u * g0, rhi = g0 + 64u < ngroups ? 8u * (g0 + 64u) : 0xFFFFFFFFu;
            uint32_t ynew = 0;
#pragma unroll
            for (uint32_t r = 0; r < R; ++r) {
                const bool here = isnew[r] && rk[r] >= rlo && rk[r] < rhi;
                if (here) ST[rk[r] - del_below(rk[r]) + before[r] + mbcnt(m[r]) - out_base] = (uint16_t)v[r];
                ynew += (uint32_t)__popcll(__ballot(here));
            }
            const uint32_t nxs = nx - 8u * g0 < 512u ? nx - 8u * g0 : 512u;  // X values of this step
            const uint32_t cnt = carry + nxs - (tot >> 16) + ynew;         // values in the window now
            __builtin_amdgcn_wave_barrier();
            const uint32_t nfull = cnt >> 3;
            for (uint32_t j = lane; j < nfull; j += 64) out_store16(&po4[(out_base >> 3) + j], ((const uint4*)ST)[j]);
            carry = cnt & 7u;
            uint16_t keep = 0;
            if (lane < carry) keep = ST[8u * nfull + lane];
            __builtin_amdgcn_wave_barrier();  // the window has been read
            if (lane < carry) ST[lane] = keep;
            out_base += 8u * nfull;
        }
        __builtin_amdgcn_wave_barrier();
        if (carry && lane == 0) out_store16(&po4[out_base >> 3], ((const uint4*)ST)[0]);  // (the slot is padded to 16 bytes)
        PH(2);
        if (lane == 0) O.meta[t.out] = pack_meta(T_ARRAY, nx + nnew - ndel, 0);
        __builtin_amdgcn_wave_barrier();  // LDS is reused by the next item
        PH(3);
    }
    PH_FLUSH(16);
}
template <bool GP8, int OPC = -1>
__global__ __launch_bounds__(256) void k_usmall(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                OutView O, const FatItem* __restrict__ q,
                                                const u64* __restrict__ qrange, int kop) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[UsmallLds<GP8>::WORDS];
    usmall_body<GP8, OPC>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, kop);
}


// ------------------------------------------------------------------ array filter (K8, K12)
// One WAVE per container pair, no workgroup barriers: the membership side X is brought into a wave-private 8 KiB LDS
// bitset -- a bitset container by 8 coalesced 16-byte loads per lane, an array by zero + ds_or_b32 scatter -- then the
// array operand Y is streamed 512 values per step (16 B/lane), each lane tests its values against the image and
// survivors are compacted with a wave prefix sum.  Replaces intersect_vector16 / difference_uint16
// (array_util.c:385-459) and array_bitset_container_intersection / _andnot (mixed_intersection.c:19-46,
// mixed_andnot.c:24-39); short streamed arrays take k_probe instead.  The result is always an array
// (containers.h:741-746, 1799-1803).
constexpr uint32_t FILTER_ST_WORDS = 264;                              // 528 u16 >= 7 + 512, a multiple of 16 bytes
constexpr uint32_t FILTER_LDS_WORDS = 4 * 2048 + 4 * FILTER_ST_WORDS;   // four waves: image + output window
// STAGED: with the output window (andnot and multi-op batches).  An `and` batch runs the instantiation without it: its
// 32 KiB blocks leave room for k_genw's 16 KiB waves on the same CU, and few of its values survive anyway.
template <bool STAGED>
__device__ __forceinline__ void filter_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                OutView O, const FatItem* __restrict__ q,
                                                const u64* __restrict__ qrange, int kop, int cardmode,
                                                u64* pair_acc) {
    const uint32_t lane = lane_id();
    uint32_t* img = lds + (threadIdx.x >> 6) * 2048u;
    uint16_t* ST = STAGED ? (uint16_t*)(lds + 4u * 2048u + (threadIdx.x >> 6) * FILTER_ST_WORDS) : nullptr;  // output window: 7 + 512 values
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
    FatItem tnext;
    if (w < n) tnext = q[w];
    PH_BEGIN();
    for (; w < n; w += nwaves) {
        const FatItem t = tnext;
        if (w + nwaves < n) tnext = q[w + nwaves];  // next work item in flight while this one is processed
        const int op = item_op(kop, t.types);
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        const uint32_t ca = t.ca, cb = t.cb;
        PH(0);
        // Y = the streamed array, X = the membership side
        bool y_is_a = true;
        if (op == OP_AND) y_is_a = (ta == T_ARRAY) && (tb != T_ARRAY || ca <= cb);
        const uint8_t* yp = y_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint8_t* xp = y_is_a ? arenaB + t.offb : arenaA + t.offa;
        const uint32_t ny = y_is_a ? ca : cb, nx = y_is_a ? cb : ca;
        const bool x_bitset = (y_is_a ? tb : ta) == T_BITSET;
        const bool keep_present = op == OP_AND;
        const uint4* __restrict__ y4 = (const uint4*)yp;
        const uint4* __restrict__ x4 = (const uint4*)xp;  // 8 values per lane per step (slots are 16-byte padded)
        uint4 yfirst = make_uint4(0, 0, 0, 0);
        if (8 * lane < ny) yfirst = y4[lane];  // first 512 values of Y: in flight while X is staged
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
        PH(1);
        // The kept values of a step are compacted into the LDS window ST (behind the <= 7 values the previous step left
        // over) and leave as whole 16-byte groups: written straight to global they were eight conditional 2-byte stores
        // per lane and step, the larger part of an andnot batch (most values survive).
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
        PH(2);
        if (cardmode) {
            if (lane == 0 && run) atomicAdd(&pair_acc[t.out], (u64)run);
        } else if (lane == 0) {
            O.meta[t.out] = pack_meta(T_ARRAY, run, 0);
        }
        __builtin_amdgcn_wave_barrier();
        PH(3);
    }
    PH_FLUSH(0);
}
template <bool STAGED>
__global__ __launch_bounds__(256) void k_filter(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                OutView O, const FatItem* __restrict__ q,
                                                const u64* __restrict__ qrange, int kop, int cardmode,
                                                u64* pair_acc) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[STAGED ? FILTER_LDS_WORDS : 8192];
    filter_body<STAGED>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, kop, cardmode, pair_acc);
}


// ------------------------------------------------------------------ bitset (op) array -> (mostly) bitset (K6)
// or / xor of a bitset with an array (either order) and bitset \ array: bitset_set_list_withcard /
// bitset_flip_list_withcard / bitset_clear_list (bitset_util.c:978-1141) behind array_bitset_container_union,
// array_bitset_container_xor and bitset_array_container_andnot (mixed_union.c:22-31, mixed_xor.c:23-39,
// mixed_andnot.c:54-72).  The BITSET never goes through LDS: its 32 words per lane stay in registers, laid out as in
// k_bb (lane l holds the 16-byte groups i * 64 + l).  Only the array is rasterised -- into a wave-private image,
// with plain (non-returning) ds_or while the bitset's loads are still in flight -- then every lane reads ITS groups of
// the image back (conflict-free 16-byte reads), combines, popcounts, and the result leaves from registers.  The
// cardinality comes from the popcount, so no returning atomics; the result is a bitset for or (containers.h:
// 1030-1045) and when xor / andnot leave more than 4096 values (mixed_xor.c:32-37, mixed_andnot.c:64-70), else the
// (rare) array result is re-queued for k_genw's extraction path, as k_bb does.
template <int OP>
__device__ __forceinline__ void ba_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                            OutView O, const FatItem* __restrict__ q, const u64* __restrict__ qrange,
                                            GenItem* retry_q, uint32_t* retry_count) {
    const uint32_t lane = lane_id();
    uint32_t* img = lds + (threadIdx.x >> 6) * 2048u;
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
    FatItem tnext;
    if (w < n) tnext = q[w];
    for (; w < n; w += nwaves) {
        const FatItem t = tnext;
        if (w + nwaves < n) tnext = q[w + nwaves];  // next work item in flight while this one is processed
        const int op = item_op(OP, t.types);
        const bool x_is_a = (uint8_t)(t.types & 0xFF) == T_BITSET;  // (andnot: always)
        const u32x4* __restrict__ px = (const u32x4*)(x_is_a ? arenaA + t.offa : arenaB + t.offb);
        const uint4* __restrict__ y4 = (const uint4*)(x_is_a ? arenaB + t.offb : arenaA + t.offa);
        const uint32_t cy = x_is_a ? t.cb : t.ca;
        u32x4 va[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = px[i * 64 + lane];
        uint4 yfirst = make_uint4(0, 0, 0, 0);
        if (8 * lane < cy) yfirst = y4[lane];
        {
            const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
        }
        __builtin_amdgcn_wave_barrier();  // every lane's slice is zero before any lane scatters into it
        for (uint32_t i = lane; 8 * i < cy; i += 64) {
            const uint4 q4 = i == lane ? yfirst : y4[i];
            const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
            const uint32_t nval = cy - 8 * i < 8u ? cy - 8 * i : 8u;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                if ((uint32_t)h < nval) {
                    const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                    atomicOr(&img[v >> 5], 1u << (v & 31));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 m4 = ((const uint4*)img)[i * 64 + lane];
            u32x4 m;
            m.x = m4.x; m.y = m4.y; m.z = m4.z; m.w = m4.w;
            va[i] = x_is_a ? vop_any<OP>(op, va[i], m) : vop_any<OP>(op, m, va[i]);
            tot += vpopc(va[i]);
        }
        const uint32_t card = wave_sum(tot);
        uint8_t* outp = O.arena + t.offo;
        if ((OP == OP_ITEM ? op == OP_OR : OP == OP_OR) || card > 4096u) {
            u32x4* __restrict__ po = (u32x4*)outp;
#pragma unroll
            for (int i = 0; i < 8; ++i) out_store16(&po[i * 64 + lane], va[i]);
            if (lane == 0) O.meta[t.out] = pack_meta(T_BITSET, card, 0);
            __builtin_amdgcn_wave_barrier();  // the image is zeroed again by the next item
            continue;
        }
        if (card == 0) {
            if (lane == 0) O.meta[t.out] = pack_meta(T_ARRAY, 0, 0);
        } else if (lane == 0) {
            // rare (the bitset held little more than the array took away): the result is an array -> re-queued for the
            // extraction path of k_genw, like k_bb's array results
            GenItem g;
            g.offa = t.offa; g.offb = t.offb; g.out = t.out; g.ca = t.ca; g.cb = t.cb; g.types = t.types;
            g.nra = 0; g.nrb = 0; g.offo = t.offo;
            retry_q[atomicAdd(retry_count, 1u)] = g;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
template <int OP>
__global__ __launch_bounds__(256) void k_ba(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                            OutView O, const FatItem* __restrict__ q, const u64* __restrict__ qrange,
                                            GenItem* retry_q, uint32_t* retry_count) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
    ba_body<OP>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, retry_q, retry_count);
}


// Sorted array out of a wave-private 8 KiB image holding rc <= 4096 set bits (the image is destroyed: it becomes the staging
// buffer), written to outp with coalesced 16-byte stores.  Shared by k_wave and the grouped union kernel (rhip_grouped.h).
__device__ __forceinline__ void wave_extract_array(uint32_t* __restrict__ img, uint32_t lane, uint32_t rc, uint8_t* __restrict__ outp) {
    // Balanced extraction.  Words are owned strided (lane l: words 64 r + l), so clustered values
    // spread over all lanes; the output position of each word comes from a two-level prefix:
    // per-word popcounts -> LDS, each lane prefix-sums 32 CONSECUTIVE counts, one wave scan of the
    // lane totals, word bases back to LDS and from there into registers.  The image is dead once the
    // words are in registers: first its lower half holds the u16 count/base table, then all of it is the
    // staging buffer the sorted values are compacted into (rc <= 4096 values = 8 KiB).
    uint32_t wv[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) wv[r] = img[64 * r + lane];
    __builtin_amdgcn_wave_barrier();
    uint16_t* tab = (uint16_t*)img;
#pragma unroll
    for (int r = 0; r < 32; ++r) tab[64 * r + lane] = (uint16_t)__popc(wv[r]);
    __builtin_amdgcn_wave_barrier();
    {
        uint4 c4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c4[i] = ((const uint4*)tab)[4 * lane + i];
        const uint32_t cw[16] = {c4[0].x, c4[0].y, c4[0].z, c4[0].w, c4[1].x, c4[1].y, c4[1].z, c4[1].w,
                                 c4[2].x, c4[2].y, c4[2].z, c4[2].w, c4[3].x, c4[3].y, c4[3].z, c4[3].w};
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += (cw[i] & 0xFFFFu) + (cw[i] >> 16);
        uint32_t base = wave_incl_scan(tot) - tot;
        uint32_t ow[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t lo = base;
            base += cw[i] & 0xFFFFu;
            const uint32_t hi = base;
            base += cw[i] >> 16;
            ow[i] = lo | (hi << 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ((uint4*)tab)[4 * lane + i] = make_uint4(ow[4 * i], ow[4 * i + 1], ow[4 * i + 2], ow[4 * i + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    uint16_t pos16[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) pos16[r] = tab[64 * r + lane];
    __builtin_amdgcn_wave_barrier();  // every base is in registers: the table may be overwritten
    uint16_t* st16 = (uint16_t*)img;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        uint32_t x = wv[r];
        uint32_t pos = pos16[r];
        const uint32_t vbase = (64u * r + lane) * 32u;
        while (x) {
            st16[pos++] = (uint16_t)(vbase + (__ffs((int)x) - 1));
            x &= x - 1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t n16 = (2u * rc + 15u) >> 4;
    uint4* __restrict__ po = (uint4*)outp;
    for (uint32_t i = lane; i < n16; i += 64) out_store16(&po[i], ((const uint4*)img)[i]);
}

// ------------------------------------------------------------------ wave-private LDS image kernel (K6, K10, K11)
// One WAVE per container pair for {array,bitset} x {array,bitset} pairs with at least one array under
// or / xor, and bitset \ array.  The wave owns an 8 KiB LDS image: X is loaded into it (bitset: 8
// coalesced 16-byte loads per lane; array: zero + ds_or scatter), then Y's values are applied with
// returning LDS atomics (ds_or_rtn / ds_xor_rtn / ds_and_rtn) whose old values give the cardinality
// delta -- bitset_set_list_withcard / bitset_flip_list_withcard / bitset_clear_list
// (bitset_util.c:978-1141) without their serial dependence.  The result is typed by the reference's
// rules and either streamed out as a bitset or extracted as a sorted array: words are owned strided
// (balanced under clustering), a two-level popcount prefix gives every word its output position, values are
// compacted into the (by then dead) image and leave with coalesced 16-byte stores.  No workgroup barrier anywhere.
__device__ __forceinline__ void wave_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const FatItem* __restrict__ q,
                                              const u64* __restrict__ qrange, int kop) {
    const uint32_t lane = lane_id();
    uint32_t* img = lds + (threadIdx.x >> 6) * 2048u;
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
    FatItem tnext;
    if (w < n) tnext = q[w];
    PH_BEGIN();
    for (; w < n; w += nwaves) {
        const FatItem t = tnext;
        if (w + nwaves < n) tnext = q[w + nwaves];  // next work item in flight while this one is processed
        const int op = item_op(kop, t.types);
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        const uint32_t ca = t.ca, cb = t.cb;
        PH(0);
        // X = image side, Y = applied array.  andnot: X = a (bitset), Y = b.  or/xor are symmetric:
        // take the bitset (or the larger array) as X.
        bool x_is_a = true;
        if (op != OP_ANDNOT) x_is_a = (ta == T_BITSET) || (tb != T_BITSET && ca >= cb);
        const uint8_t tx = x_is_a ? ta : tb;
        const uint32_t cx = x_is_a ? ca : cb, cy = x_is_a ? cb : ca;
        const uint8_t* xp = x_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint4* __restrict__ y4 = (const uint4*)(x_is_a ? arenaB + t.offb : arenaA + t.offa);
        uint4 yfirst = make_uint4(0, 0, 0, 0);
        if (8 * lane < cy) yfirst = y4[lane];  // first 512 values of Y: in flight while X is staged
        if (tx == T_BITSET) {
            const uint4* __restrict__ g = (const uint4*)xp;
            uint4 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = g[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = xv[i];
        } else {
            const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
            __builtin_amdgcn_wave_barrier();
            wave_scatter_or(img, (const uint4*)xp, cx, lane);
        }
        __builtin_amdgcn_wave_barrier();  // X is complete in the image before Y is applied (xor / clear are order-sensitive)
        PH(1);
        // (one copy of the loop per op, the next 16 bytes of Y loaded while the current ones are applied: written as
        // `i == lane ? yfirst : y4[i]` with the op tested per value the compiler produced four branch-guarded dword
        // loads with a full wait each, and three-way branches around every atomic)
        int delta = 0;
        auto apply = [&](auto opc) {
            constexpr int OPK = decltype(opc)::value;
            uint4 q4 = yfirst;
            for (uint32_t i = lane; 8 * i < cy; i += 64) {
                uint4 nxt = make_uint4(0, 0, 0, 0);
                if (8 * (i + 64) < cy) nxt = y4[i + 64];
                const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
                const uint32_t nval = cy - 8 * i < 8u ? cy - 8 * i : 8u;
                uint32_t old[8], bit[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                    bit[h] = ((uint32_t)h < nval) ? (1u << (v & 31)) : 0u;
                    if (OPK == OP_OR) old[h] = atomicOr(&img[v >> 5], bit[h]);
                    else if (OPK == OP_XOR) old[h] = atomicXor(&img[v >> 5], bit[h]);
                    else old[h] = atomicAnd(&img[v >> 5], ~bit[h]);
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const int was = (old[h] & bit[h]) != 0 ? 1 : 0, has = bit[h] != 0 ? 1 : 0;
                    if (OPK == OP_OR) delta += has - was;           // (was implies has)
                    else if (OPK == OP_XOR) delta += has - 2 * was;
                    else delta -= was;
                }
                q4 = nxt;
            }
        };
        if (op == OP_OR) apply(std::integral_constant<int, OP_OR>{});
        else if (op == OP_XOR) apply(std::integral_constant<int, OP_XOR>{});
        else apply(std::integral_constant<int, OP_ANDNOT>{});
        PH(2);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) delta += __shfl_xor(delta, o);
        const uint32_t rc = (uint32_t)((int)cx + delta);
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, ta, tb, ca, cb, false, false, rc, 0);
        uint8_t* outp = O.arena + t.offo;
        __builtin_amdgcn_wave_barrier();
        PH(3);
        if (rc && ty == T_BITSET) {
            uint4* __restrict__ po = (uint4*)outp;
            uint4 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = ((const uint4*)img)[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 8; ++i) out_store16(&po[i * 64 + lane], xv[i]);
        } else if (rc) {
            wave_extract_array(img, lane, rc, outp);
        }
        PH(5);
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, 0);
        __builtin_amdgcn_wave_barrier();
        PH(6);
    }
    PH_FLUSH(8);
}
__global__ __launch_bounds__(256) void k_wave(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const FatItem* __restrict__ q,
                                              const u64* __restrict__ qrange, int kop) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
    wave_body(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, kop);
}

