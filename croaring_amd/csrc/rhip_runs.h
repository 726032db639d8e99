// rhip_runs.h -- wave-per-pair kernels for pairs with a run operand: k_runs (interval algebra), k_genw (LDS image)
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ interval kernel (K13, K14, K16)
// run x run, array x run, run x array for all four ops, in O(nA + nB) work instead of rasterising 65536 bits: one
// WAVE per pair, no workgroup barrier.  Replaces the sequential interval merges
// run_container_{union,intersection,xor,andnot} (src/containers/run.c:231-283, 387-463, 348-383, 575-633),
// array_run_container_{intersection,union,andnot,lazy_xor}, run_array_container_andnot
// (mixed_intersection.c:73-111, mixed_union.c:66-108, mixed_andnot.c:277-412, mixed_xor.c:140-173).
//
// Each operand is read as a sorted BOUNDARY list b(0) <= b(1) <= ... <= b(2n-1) = s0, e0+1, s1, e1+1, ...
// (arrays: e = s).  Membership is a parity: x is in the operand iff an odd number of its boundaries are <= x.  The
// two lists (staged in LDS) are merged by MERGE PATH: lane l owns merged positions [l * per, (l+1) * per) and finds
// its split (ia, ib) with one binary search along its diagonal; the state before its chunk is just (ia & 1, ib & 1).
// It then walks its <= 16 events sequentially (two LDS reads per event), toggling inA / inB.  Only the LAST event at
// a position is "effective" (an array's v+1 / next-v pair, or a boundary shared by both lists, toggles twice at
// one position); a result run starts at an effective event where op(inA, inB) turns 1 and ends before one where it
// turns 0, "turns" being relative to the previous effective event -- across lanes that is one ballot pair and a
// count-leading-zeros.  Three walks: (1) last effective state per lane, (2) count starts / ends, (3) write them at
// wave-scanned positions.  The run list is then typed by the reference's rules (convert_run_to_efficient_container
// etc.) and written as runs or expanded into an array; the rare bitset result is re-queued for k_genw.
// (Round 1 evaluated every boundary with a binary search into the other list, twice: 7 ns per pair of 100-run
// containers; the merge walk does the same in a fraction of the LDS round trips.)
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
    constexpr uint32_t NB = 2 * RUNS_MAX_INTERVALS;  // max boundaries per list = max result runs
    // per wave (~4 KiB): both operand lists staged in LDS, result run starts / ends
    __shared__ __attribute__((aligned(16))) uint8_t lists_all[4][2][1024];  // 4 * RUNS_MAX_INTERVALS, padded to 16 bytes
    __shared__ uint16_t lds_all[4][2 * NB + 2];
    const uint32_t lane = lane_id();
    uint16_t* base = lds_all[threadIdx.x >> 6];
    uint8_t* lsA = lists_all[threadIdx.x >> 6][0];
    uint8_t* lsB = lists_all[threadIdx.x >> 6][1];
    uint16_t* RS = base;            // result run starts
    uint16_t* RE = base + NB;       // result run ends (inclusive)
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    GenItem tnext;
    if (wi < n) tnext = q[wi];
    for (; wi < n; wi += nwaves) {
        const GenItem t = tnext;
        if (wi + nwaves < n) tnext = q[wi + nwaves];  // next work item in flight while this one is processed
        const uint32_t ta = t.types & 0xFFu, tb = t.types >> 8;
        IvList LA, LB;
        LA.p = lsA; LA.is_run = ta == T_RUN; LA.n2 = 2u * (ta == T_RUN ? t.nra : t.ca);
        LB.p = lsB; LB.is_run = tb == T_RUN; LB.n2 = 2u * (tb == T_RUN ? t.nrb : t.cb);
        {   // stage both payloads (<= 1 KiB each, 16-byte padded slots): one 16-byte load per lane
            const uint32_t na16 = ((LA.is_run ? 2u : 1u) * LA.n2 + 15u) >> 4;
            const uint32_t nb16 = ((LB.is_run ? 2u : 1u) * LB.n2 + 15u) >> 4;
            if (lane < na16) ((uint4*)lsA)[lane] = ((const uint4*)(arenaA + t.offa))[lane];
            if (lane < nb16) ((uint4*)lsB)[lane] = ((const uint4*)(arenaB + t.offb))[lane];
            __builtin_amdgcn_wave_barrier();
        }
        // ---- merge path: this lane's chunk of the merged boundary sequence
        constexpr uint32_t SENT = 0x20000u;  // past every boundary (the largest is 65536)
        const uint32_t nA = LA.n2, nB2 = LB.n2, E = nA + nB2;
        const uint32_t per = (E + 63u) >> 6;
        const uint32_t d0 = lane * per < E ? lane * per : E;
        const uint32_t d1 = d0 + per < E ? d0 + per : E;
        uint32_t ia0;
        {
            uint32_t lo = d0 > nB2 ? d0 - nB2 : 0u, hi = d0 < nA ? d0 : nA;
            while (lo < hi) {  // A goes first on ties: A[mid] <= B[d0 - mid - 1] means more of A lies before the diagonal
                const uint32_t mid = (lo + hi) >> 1;
                if (LA.at(mid) <= LB.at(d0 - mid - 1u)) lo = mid + 1u;
                else hi = mid;
            }
            ia0 = lo;
        }
        const uint32_t ib0 = d0 - ia0, steps = d1 - d0;
        // walk the chunk; fn(p, g) is called for every EFFECTIVE event (last event at position p), g = op state after it
        auto walk = [&](auto&& fn) {
            uint32_t ia = ia0, ib = ib0, inA = ia0 & 1u, inB = ib0 & 1u;
            uint32_t pa = ia < nA ? LA.at(ia) : SENT, pb = ib < nB2 ? LB.at(ib) : SENT;
            for (uint32_t sidx = 0; sidx < steps; ++sidx) {
                uint32_t pcur;
                if (pa <= pb) {
                    pcur = pa; inA ^= 1u; ++ia;
                    pa = ia < nA ? LA.at(ia) : SENT;
                } else {
                    pcur = pb; inB ^= 1u; ++ib;
                    pb = ib < nB2 ? LB.at(ib) : SENT;
                }
                const uint32_t pnext = pa < pb ? pa : pb;
                if (pnext != pcur) fn(pcur, bop(op, inA, inB));
            }
        };
        // walk 1: does the chunk hold an effective event, and the state after its last one
        bool has_eff = false, g_last = false;
        walk([&](uint32_t, bool g) { has_eff = true; g_last = g; });
        bool gprev = false;  // state after the last effective event BEFORE this chunk
        {
            const u64 mh = __ballot(has_eff), mg = __ballot(g_last);
            const u64 below = mh & ((1ull << lane) - 1ull);
            if (below) gprev = (mg >> (63 - __clzll((long long)below))) & 1ull;
        }
        // walk 2: result run starts / ends in this chunk
        uint32_t ns = 0, ne = 0;
        {
            bool gp = gprev;
            walk([&](uint32_t, bool g) { ns += (g && !gp) ? 1u : 0u; ne += (!g && gp) ? 1u : 0u; gp = g; });
        }
        const uint32_t incs = wave_incl_scan(ns), ince = wave_incl_scan(ne);
        const uint32_t rn = __shfl(incs, 63);  // == total ends: every run that starts also ends (both states end at 0)
        // walk 3: write them
        {
            uint32_t ks = incs - ns, ke = ince - ne;
            bool gp = gprev;
            walk([&](uint32_t pcur, bool g) {
                if (g && !gp) RS[ks++] = (uint16_t)pcur;
                if (!g && gp) RE[ke++] = (uint16_t)(pcur - 1u);
                gp = g;
            });
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
            // expand runs into a sorted array.  An array-typed result has short runs (2 rc <= 4 rn + 2), so: one lane
            // per run writes its first 8 values at the scanned position; the few longer runs are finished by the whole
            // wave, one after the other.  (A binary search per output value -- 8 dependent LDS reads for each of up to
            // 4096 values -- cost 20 us per pair on run-compressed data.)
            uint16_t* __restrict__ o16 = (uint16_t*)outp;
            uint32_t runbase = 0;
            for (uint32_t k0 = 0; k0 < rn; k0 += 64) {
                const uint32_t k = k0 + lane;
                const uint32_t st = k < rn ? (uint32_t)RS[k] : 0u;
                const uint32_t len = k < rn ? (uint32_t)RE[k] - st + 1u : 0u;
                const uint32_t inc = wave_incl_scan(len);
                const uint32_t pos = runbase + inc - len;
                runbase += __shfl(inc, 63);
                const uint32_t lim = len < 8u ? len : 8u;
                for (uint32_t j = 0; j < lim; ++j) o16[pos + j] = (uint16_t)(st + j);
                u64 longm = __ballot(len > 8u);
                while (longm) {
                    const int src = __ffsll((long long)longm) - 1;
                    longm &= longm - 1;
                    const uint32_t ls = __shfl(st, src), ll = __shfl(len, src), lp = __shfl(pos, src);
                    for (uint32_t j = 8u + lane; j < ll; j += 64) o16[lp + j] = (uint16_t)(ls + j);
                }
            }
        }
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ short interval lists: four pairs per wave
// The same interval algebra as k_runs for pairs with <= R16_MAX_IV intervals per operand and <= R16_MAX_CARD values in
// all: sparse run-compressed data (wikileaks-noquotes: three quarters of the matched pairs) is made of such pairs,
// and a wave that spends its ~2 000 instructions on one of them leaves 50-60 lanes idle throughout.  Here every
// 16-lane group of the wave owns one pair: merge path over 16 lanes, group-wide scans and sums (shuffles that never
// leave the group), group-private LDS.  Control flow around the collectives stays wave-uniform: the four groups walk
// their items in lockstep and the only loops with collectives inside have constant trip counts.
__device__ __forceinline__ uint32_t grp16_incl_scan(uint32_t v, uint32_t gl) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const uint32_t t = __shfl_up(v, o);
        if (gl >= (uint32_t)o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t grp16_sum(uint32_t v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void k_runs16(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                OutView O, const GenItem* __restrict__ q,
                                                const u64* __restrict__ qrange, int op, int cardmode, u64* pair_acc,
                                                GenItem* retry_q, uint32_t* retry_count) {
    constexpr uint32_t NB = 64;  // >= result runs of a pair: at most (boundaries of both lists) / 2 = 2 * R16_MAX_IV
    __shared__ __attribute__((aligned(16))) uint8_t lists_all[16][2][128];  // per group: both payloads, 16-byte padded
    __shared__ uint16_t lds_all[16][2 * NB];
    const uint32_t lane = lane_id(), grp = lane >> 4, gl = lane & 15u;
    const uint32_t gslot = (threadIdx.x >> 4);  // group index inside the block
    uint8_t* lsA = lists_all[gslot][0];
    uint8_t* lsB = lists_all[gslot][1];
    uint16_t* RS = lds_all[gslot];
    uint16_t* RE = RS + NB;
    const uint32_t glast = (lane & 48u) | 15u;  // last lane of this group
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    GenItem tnext = {};
    if (4 * wi + grp < n) tnext = q[4 * wi + grp];
    for (; 4 * wi < n; wi += nwaves) {
        const bool have = 4 * wi + grp < n;
        const GenItem t = tnext;
        if (4 * (wi + nwaves) + grp < n) tnext = q[4 * (wi + nwaves) + grp];
        const uint32_t ta = t.types & 0xFFu, tb = t.types >> 8;
        IvList LA, LB;
        LA.p = lsA; LA.is_run = ta == T_RUN; LA.n2 = have ? 2u * (ta == T_RUN ? t.nra : t.ca) : 0u;
        LB.p = lsB; LB.is_run = tb == T_RUN; LB.n2 = have ? 2u * (tb == T_RUN ? t.nrb : t.cb) : 0u;
        {   // stage both payloads (<= 124 bytes each): lanes 0-7 of the group load A, lanes 8-15 load B
            const uint32_t na16 = ((LA.is_run ? 2u : 1u) * LA.n2 + 15u) >> 4;
            const uint32_t nb16 = ((LB.is_run ? 2u : 1u) * LB.n2 + 15u) >> 4;
            if (gl < 8u) {
                if (gl < na16) ((uint4*)lsA)[gl] = ((const uint4*)(arenaA + t.offa))[gl];
            } else if (gl - 8u < nb16) {
                ((uint4*)lsB)[gl - 8u] = ((const uint4*)(arenaB + t.offb))[gl - 8u];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- merge path over the 16 lanes of the group (see k_runs)
        constexpr uint32_t SENT = 0x20000u;
        const uint32_t nA = LA.n2, nB2 = LB.n2, E = nA + nB2;
        const uint32_t per = (E + 15u) >> 4;
        const uint32_t d0 = gl * per < E ? gl * per : E;
        const uint32_t d1 = d0 + per < E ? d0 + per : E;
        uint32_t ia0;
        {
            uint32_t lo = d0 > nB2 ? d0 - nB2 : 0u, hi = d0 < nA ? d0 : nA;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (LA.at(mid) <= LB.at(d0 - mid - 1u)) lo = mid + 1u;
                else hi = mid;
            }
            ia0 = lo;
        }
        const uint32_t ib0 = d0 - ia0, steps = d1 - d0;
        auto walk = [&](auto&& fn) {
            uint32_t ia = ia0, ib = ib0, inA = ia0 & 1u, inB = ib0 & 1u;
            uint32_t pa = ia < nA ? LA.at(ia) : SENT, pb = ib < nB2 ? LB.at(ib) : SENT;
            for (uint32_t sidx = 0; sidx < steps; ++sidx) {
                uint32_t pcur;
                if (pa <= pb) {
                    pcur = pa; inA ^= 1u; ++ia;
                    pa = ia < nA ? LA.at(ia) : SENT;
                } else {
                    pcur = pb; inB ^= 1u; ++ib;
                    pb = ib < nB2 ? LB.at(ib) : SENT;
                }
                const uint32_t pnext = pa < pb ? pa : pb;
                if (pnext != pcur) fn(pcur, bop(op, inA, inB));
            }
        };
        bool has_eff = false, g_last = false;
        walk([&](uint32_t, bool g) { has_eff = true; g_last = g; });
        bool gprev = false;
        {
            const uint32_t mh = (uint32_t)(__ballot(has_eff) >> (16u * grp)) & 0xFFFFu;
            const uint32_t mg = (uint32_t)(__ballot(g_last) >> (16u * grp)) & 0xFFFFu;
            const uint32_t below = mh & ((1u << gl) - 1u);
            if (below) gprev = (mg >> (31 - __clz((int)below))) & 1u;
        }
        uint32_t ns = 0, ne = 0;
        {
            bool gp = gprev;
            walk([&](uint32_t, bool g) { ns += (g && !gp) ? 1u : 0u; ne += (!g && gp) ? 1u : 0u; gp = g; });
        }
        const uint32_t incs = grp16_incl_scan(ns, gl), ince = grp16_incl_scan(ne, gl);
        const uint32_t rn = __shfl(incs, glast);
        {
            uint32_t ks = incs - ns, ke = ince - ne;
            bool gp = gprev;
            walk([&](uint32_t pcur, bool g) {
                if (g && !gp) RS[ks++] = (uint16_t)pcur;
                if (!g && gp) RE[ke++] = (uint16_t)(pcur - 1u);
                gp = g;
            });
        }
        __builtin_amdgcn_wave_barrier();
        // ---- cardinality, typing
        uint32_t cnt = 0;
        for (uint32_t k = gl; k < rn; k += 16) cnt += (uint32_t)RE[k] - (uint32_t)RS[k] + 1u;
        const uint32_t rc = grp16_sum(cnt);
        if (cardmode) {  // (wave-uniform)
            if (have && gl == 0 && rc) atomicAdd(&pair_acc[t.out], (u64)rc);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        const bool fulla = ta == T_RUN && t.ca == 65536u, fullb = tb == T_RUN && t.cb == 65536u;
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, (int)ta, (int)tb, t.ca, t.cb, fulla, fullb, rc, rn);
        const bool redo = have && rc && ty == T_BITSET;  // cannot happen below 4097 values; kept for safety
        if (redo && gl == 0) retry_q[atomicAdd(retry_count, 1u)] = t;
        const bool wr = have && rc && !redo;
        uint8_t* outp = O.arena + t.offo;
        if (wr && ty == T_RUN) {
            uint32_t* __restrict__ o32 = (uint32_t*)outp;
            for (uint32_t k = gl; k < rn; k += 16)
                o32[k] = (uint32_t)RS[k] | (((uint32_t)RE[k] - (uint32_t)RS[k]) << 16);
        }
        const bool arr = wr && ty != T_RUN;
        {   // runs -> sorted array as in k_runs: a lane per run for its first 8 values, the group for the rest of a
            // longer one.  Constant trip count (rn < NB) and a wave-uniform long-run loop keep the collectives legal.
            uint16_t* __restrict__ o16 = (uint16_t*)outp;
            uint32_t runbase = 0;
#pragma unroll
            for (uint32_t k0 = 0; k0 < NB; k0 += 16) {
                const uint32_t k = k0 + gl;
                const bool v = arr && k < rn;
                const uint32_t st = v ? (uint32_t)RS[k] : 0u;
                const uint32_t len = v ? (uint32_t)RE[k] - st + 1u : 0u;
                const uint32_t inc = grp16_incl_scan(len, gl);
                const uint32_t pos = runbase + inc - len;
                runbase += __shfl(inc, glast);
                const uint32_t lim = len < 8u ? len : 8u;
                for (uint32_t j = 0; j < lim; ++j) o16[pos + j] = (uint16_t)(st + j);
                uint32_t lm = (uint32_t)(__ballot(len > 8u) >> (16u * grp)) & 0xFFFFu;
                while (__ballot(lm != 0u)) {
                    const bool act = lm != 0u;
                    const uint32_t src = (lane & 48u) | (act ? (uint32_t)__ffs((int)lm) - 1u : 0u);
                    lm &= lm - 1u;
                    const uint32_t ls = __shfl(st, src), ll = __shfl(len, src), lp = __shfl(pos, src);
                    if (act)
                        for (uint32_t j = 8u + gl; j < ll; j += 16) o16[lp + j] = (uint16_t)(ls + j);
                }
            }
        }
        if (have && !redo && gl == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
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
