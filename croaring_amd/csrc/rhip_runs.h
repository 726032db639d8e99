// rhip_runs.h -- wave-per-pair kernels for pairs with a run operand: k_runs (interval algebra), k_genw (LDS image)
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ interval kernel (K13, K14, K16)
// run x run, array x run, run x array for all four ops, in O((nA + nB) log(nA + nB)) instead of
// rasterising 65536 bits: one WAVE per pair, no workgroup barrier.  Replaces the sequential interval
// merges run_container_{union,intersection,xor,andnot} (src/containers/run.c:231-283, 387-463, 348-383,
// 575-633), array_run_container_{intersection,union,andnot,lazy_xor}, run_array_container_andnot
// (mixed_intersection.c:73-111, mixed_union.c:66-108, mixed_andnot.c:277-412, mixed_xor.c:140-173).
//
// Each operand is read as a sorted BOUNDARY list b(0) <= b(1) <= ... <= b(2n-1) = s0, e0+1, s1, e1+1, ...
// (arrays: e = s).  Membership is a parity: x is in the operand iff |{j : b(j) <= x}| is odd.  The result
// can only change at a boundary p of either operand; with lb/ub = lower/upper bound of p in a list,
//   f(p-1) = op(lbA & 1, lbB & 1),   f(p) = op(ubA & 1, ubB & 1),
// so p starts a result run iff f(p) & !f(p-1) and ends one (at p-1) iff !f(p) & f(p-1).  Lanes evaluate
// boundaries in parallel (one binary search into the other list each); result starts and ends are ranked
// by ballot prefix counts per list plus a prefix lookup in the other list, and the k-th start pairs with
// the k-th end.  The run list is then typed by the reference's rules (convert_run_to_efficient_container
// etc.) and written as runs or expanded into an array; the rare bitset result is re-queued for k_genw.
struct IvList {
    const uint8_t* p;
    uint32_t n2;     // number of boundaries (2 x intervals)
    bool is_run;
    __device__ __forceinline__ uint32_t at(uint32_t j) const {
        if (is_run) {
            const uint32_t w = ((const uint32_t*)p)[j >> 1];
            const uint32_t s = w & 0xFFFFu;
            return (j & 1u) ? s + (w >> 16) + 1u : s;
        }
        const uint32_t v = ((const uint16_t*)p)[j >> 1];
        return v + (j & 1u);
    }
    __device__ __forceinline__ uint32_t lower(uint32_t x) const {  // first j with at(j) >= x
        uint32_t lo = 0, hi = n2;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (at(mid) < x) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
};
__device__ __forceinline__ bool bop(int op, uint32_t a, uint32_t b) {
    a &= 1u; b &= 1u;
    return op == OP_AND ? (a & b) : op == OP_OR ? (a | b) : op == OP_XOR ? (a ^ b) : (a & ~b & 1u);
}

__global__ __launch_bounds__(256) void k_runs(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const GenItem* __restrict__ q,
                                              const u64* __restrict__ qrange, int op, int cardmode, u64* pair_acc,
                                              GenItem* retry_q, uint32_t* retry_count) {
    constexpr uint32_t NB = 2 * RUNS_MAX_INTERVALS;  // max boundaries per list
    // per wave (~8 KiB): both operand lists staged in LDS (every binary-search probe is an LDS read), the
    // start/end prefix tables of both lists (bit 15 = flag), result starts / ends
    __shared__ __attribute__((aligned(16))) uint8_t lists_all[4][2][1024];  // 4 * RUNS_MAX_INTERVALS, padded to 16 bytes
    __shared__ uint16_t lds_all[4][4 * (NB + 1) + 2 * NB];
    const uint32_t lane = lane_id();
    uint16_t* base = lds_all[threadIdx.x >> 6];
    uint8_t* lsA = lists_all[threadIdx.x >> 6][0];
    uint8_t* lsB = lists_all[threadIdx.x >> 6][1];
    uint16_t* PS[2] = {base, base + (NB + 1)};                   // start-prefix of list A / B
    uint16_t* PE[2] = {base + 2 * (NB + 1), base + 3 * (NB + 1)};  // end-prefix of list A / B
    uint16_t* RS = base + 4 * (NB + 1);                           // result run starts
    uint16_t* RE = RS + NB;                                       // result run ends (inclusive)
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    GenItem tnext;
    if (wi < n) tnext = q[wi];
    for (; wi < n; wi += nwaves) {
        const GenItem t = tnext;
        if (wi + nwaves < n) tnext = q[wi + nwaves];  // next work item in flight while this one is processed
        const uint32_t ta = t.types & 0xFFu, tb = t.types >> 8;
        IvList L[2];
        L[0].p = lsA; L[0].is_run = ta == T_RUN; L[0].n2 = 2u * (ta == T_RUN ? t.nra : t.ca);
        L[1].p = lsB; L[1].is_run = tb == T_RUN; L[1].n2 = 2u * (tb == T_RUN ? t.nrb : t.cb);
        {   // stage both payloads (<= 1 KiB each, 16-byte padded slots): one 16-byte load per lane
            const uint32_t na16 = ((L[0].is_run ? 2u : 1u) * L[0].n2 + 15u) >> 4;
            const uint32_t nb16 = ((L[1].is_run ? 2u : 1u) * L[1].n2 + 15u) >> 4;
            if (lane < na16) ((uint4*)lsA)[lane] = ((const uint4*)(arenaA + t.offa))[lane];
            if (lane < nb16) ((uint4*)lsB)[lane] = ((const uint4*)(arenaB + t.offb))[lane];
            __builtin_amdgcn_wave_barrier();
        }
        // ---- pass 1: start / end flags of every boundary, exclusive prefix counts per list
        uint32_t tot_s[2], tot_e[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const IvList& own = L[x];
            const IvList& oth = L[1 - x];
            uint32_t run_s = 0, run_e = 0;
            for (uint32_t j0 = 0; j0 < own.n2; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool is_s = false, is_e = false;
                if (j < own.n2) {
                    const uint32_t p = own.at(j);
                    const bool dup_own = j > 0 && own.at(j - 1) == p;
                    const uint32_t lbo = oth.lower(p);
                    const bool in_oth = lbo < oth.n2 && oth.at(lbo) == p;
                    // a boundary present in both lists is handled once, by list A
                    if (!dup_own && !(x == 1 && in_oth)) {
                        const uint32_t ub_own = j + 1u + ((j + 1u < own.n2 && own.at(j + 1u) == p) ? 1u : 0u);
                        uint32_t ub_oth = lbo;
                        if (in_oth) ub_oth = lbo + 1u + ((lbo + 1u < oth.n2 && oth.at(lbo + 1u) == p) ? 1u : 0u);
                        const uint32_t lbA = x == 0 ? j : lbo, ubA = x == 0 ? ub_own : ub_oth;
                        const uint32_t lbB = x == 0 ? lbo : j, ubB = x == 0 ? ub_oth : ub_own;
                        const bool fb = bop(op, lbA, lbB), fa = bop(op, ubA, ubB);
                        is_s = fa && !fb;
                        is_e = !fa && fb;
                    }
                }
                const u64 ms = __ballot(is_s), me = __ballot(is_e);
                if (j < own.n2) {
                    PS[x][j] = (uint16_t)((run_s + mbcnt(ms)) | (is_s ? 0x8000u : 0u));
                    PE[x][j] = (uint16_t)((run_e + mbcnt(me)) | (is_e ? 0x8000u : 0u));
                }
                run_s += (uint32_t)__popcll(ms);
                run_e += (uint32_t)__popcll(me);
            }
            if (lane == 0) { PS[x][own.n2] = (uint16_t)run_s; PE[x][own.n2] = (uint16_t)run_e; }
            tot_s[x] = run_s; tot_e[x] = run_e;
        }
        const uint32_t rn = tot_s[0] + tot_s[1];  // == tot_e[0] + tot_e[1]
        __builtin_amdgcn_wave_barrier();
        // ---- pass 2: rank flagged boundaries over both lists, scatter into RS / RE
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const IvList& own = L[x];
            const IvList& oth = L[1 - x];
            for (uint32_t j0 = 0; j0 < own.n2; j0 += 64) {
                const uint32_t j = j0 + lane;
                if (j < own.n2) {
                    const uint32_t fs = PS[x][j], fe = PE[x][j];
                    if ((fs | fe) & 0x8000u) {
                        const uint32_t p = own.at(j);
                        const uint32_t lbo = oth.lower(p);
                        if (fs & 0x8000u) RS[(fs & 0x7FFFu) + (PS[1 - x][lbo] & 0x7FFFu)] = (uint16_t)p;
                        if (fe & 0x8000u) RE[(fe & 0x7FFFu) + (PE[1 - x][lbo] & 0x7FFFu)] = (uint16_t)(p - 1u);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- cardinality, typing
        uint32_t cnt = 0;
        for (uint32_t k = lane; k < rn; k += 64) cnt += (uint32_t)RE[k] - (uint32_t)RS[k] + 1u;
        const uint32_t rc = wave_sum(cnt);
        if (cardmode) {
            if (lane == 0 && rc) atomicAdd(&pair_acc[t.out], (u64)rc);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        const bool fulla = ta == T_RUN && t.ca == 65536u, fullb = tb == T_RUN && t.cb == 65536u;
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, (int)ta, (int)tb, t.ca, t.cb, fulla, fullb, rc, rn);
        if (rc && ty == T_BITSET) {
            // rare for this class: let the image kernel redo the pair
            if (lane == 0) retry_q[atomicAdd(retry_count, 1u)] = t;
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        uint8_t* outp = O.arena + t.offo;
        if (rc && ty == T_RUN) {
            uint32_t* __restrict__ o32 = (uint32_t*)outp;
            for (uint32_t k = lane; k < rn; k += 64)
                o32[k] = (uint32_t)RS[k] | (((uint32_t)RE[k] - (uint32_t)RS[k]) << 16);
        } else if (rc) {
            // expand runs into a sorted array: exclusive prefix of run lengths (reuses PS[0]), then one
            // binary search per output value
            uint16_t* PL = PS[0];
            uint32_t runbase = 0;
            for (uint32_t k0 = 0; k0 < rn; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t len = k < rn ? (uint32_t)RE[k] - (uint32_t)RS[k] + 1u : 0u;
                const uint32_t inc = wave_incl_scan(len);
                if (k < rn) PL[k] = (uint16_t)(runbase + inc - len);
                runbase += __shfl(inc, 63);
            }
            __builtin_amdgcn_wave_barrier();
            uint16_t* __restrict__ o16 = (uint16_t*)outp;
            for (uint32_t i = lane; i < rc; i += 64) {
                uint32_t lo = 0, hi = rn;  // last k with PL[k] <= i
                while (lo + 1 < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (PL[mid] <= i) lo = mid;
                    else hi = mid;
                }
                o16[i] = (uint16_t)(RS[lo] + (i - PL[lo]));
            }
        }
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ wave-level general pair kernel (K5, K7, K13-K16)
// Every type pair the specialised kernels do not take (all pairs with a run container, plus
// bitset x bitset results that must become arrays): ONE WAVE per container pair, two wave-private
// 8 KiB LDS images, no workgroup barrier.  Lane l owns the 32 consecutive logical words
// [32 l, 32 l + 32) -- the ownership that prefix-XOR run rasterisation and run counting need -- and a
// skewed transposed physical layout keeps both the per-lane accesses (k-th word of every lane) and
// the coalesced global<->LDS copies conflict-free:
__device__ __forceinline__ uint32_t wphys(uint32_t w) { return ((w & 31u) << 6) | (((w >> 5) + (w & 31u)) & 63u); }
__device__ __forceinline__ uint32_t wown(uint32_t lane, uint32_t k) { return (k << 6) | ((lane + k) & 63u); }

// Rasterise one container into a wave-private image (K6: array scatter; K7: runs as toggle bits at
// start / end+1 followed by a 65536-bit inclusive prefix-XOR -- 5 shift-xors per word, a serial carry
// over the lane's 32 words and ONE ballot for the carry across lanes).
__device__ void wimg_build(uint32_t* img, const uint8_t* __restrict__ p, uint32_t type, uint32_t card,
                           uint32_t nruns) {
    const uint32_t lane = lane_id();
    if (type == T_BITSET) {
        const uint4* __restrict__ g = (const uint4*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 x = g[i * 64 + lane];
            const uint32_t w0 = 4u * (i * 64 + lane);
            img[wphys(w0)] = x.x; img[wphys(w0 + 1)] = x.y; img[wphys(w0 + 2)] = x.z; img[wphys(w0 + 3)] = x.w;
        }
        return;
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
    __builtin_amdgcn_wave_barrier();  // the whole image is zero before any lane scatters into another lane's words
    const uint4* __restrict__ q4p = (const uint4*)p;
    if (type == T_ARRAY) {
        for (uint32_t i = lane; 8 * i < card; i += 64) {
            const uint4 q4 = q4p[i];
            const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                if (8 * i + h < card) {
                    const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                    atomicOr(&img[wphys(v >> 5)], 1u << (v & 31));
                }
            }
        }
        return;
    }
    for (uint32_t i = lane; 4 * i < nruns; i += 64) {  // 4 runs {u16 value, u16 length} per 16-byte load
        const uint4 q4 = q4p[i];
        const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (4 * i + h < nruns) {
                const uint32_t s0 = d[h] & 0xFFFFu, e1 = s0 + (d[h] >> 16) + 1u;
                atomicXor(&img[wphys(s0 >> 5)], 1u << (s0 & 31));
                if (e1 < 65536u) atomicXor(&img[wphys(e1 >> 5)], 1u << (e1 & 31));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t w[32];
    uint32_t par = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        w[k] = img[wown(lane, k)];
        par ^= __popc(w[k]) & 1u;
    }
    uint32_t carry = mbcnt(__ballot(par != 0)) & 1u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const uint32_t x = w[k];
        uint32_t y = x;
        y ^= y << 1; y ^= y << 2; y ^= y << 4; y ^= y << 8; y ^= y << 16;
        img[wown(lane, k)] = carry ? ~y : y;
        carry ^= __popc(x) & 1u;
    }
}

__global__ __launch_bounds__(256) void k_genw(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const GenItem* __restrict__ q,
                                              const u64* __restrict__ qrange, const uint32_t* __restrict__ qcount,
                                              int op, int cardmode, u64* pair_acc) {
    // ONE 8 KiB image per wave: operand A is rasterised, pulled into registers, then the same image is
    // reused for operand B and finally as the output staging buffer (16 waves per CU instead of 8)
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* ia = img_all[threadIdx.x >> 6];
    uint32_t* ib = ia;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = qrange ? (uint32_t)(qrange[1] - qrange[0]) : *qcount;
    for (uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wi < n; wi += nwaves) {
        const GenItem t = q[wi];
        const uint32_t ta = t.types & 0xFFu, tb = t.types >> 8;
        wimg_build(ia, arenaA + t.offa, ta, t.ca, t.nra);
        __builtin_amdgcn_wave_barrier();
        uint32_t r[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) r[k] = ia[wown(lane, k)];
        __builtin_amdgcn_wave_barrier();
        wimg_build(ib, arenaB + t.offb, tb, t.cb, t.nrb);
        __builtin_amdgcn_wave_barrier();
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const uint32_t a = r[k], b = ib[wown(lane, k)];
            r[k] = op == OP_AND ? (a & b) : op == OP_OR ? (a | b) : op == OP_XOR ? (a ^ b) : (a & ~b);
            cnt += __popc(r[k]);
        }
        const uint32_t rc = wave_sum(cnt);
        if (cardmode) {
            if (lane == 0 && rc) atomicAdd(&pair_acc[t.out], (u64)rc);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // canonical run count: set bits whose predecessor is clear (bitset_container_number_of_runs, bitset.c:1046-1062)
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
        const bool fulla = ta == T_RUN && t.ca == 65536u, fullb = tb == T_RUN && t.cb == 65536u;
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, (int)ta, (int)tb, t.ca, t.cb, fulla, fullb, rc, rn);
        uint8_t* outp = O.arena + t.offo;
        uint32_t* img = ia;
#include "rhip_wemit.inc"
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}
