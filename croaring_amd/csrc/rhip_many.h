// rhip_many.h -- many-way OR / XOR aggregation kernels (roaring_bitmap_or_many /
// roaring_bitmap_xor_many, src/roaring.c:775-809).
//
// The reference folds bitmap after bitmap into a growing answer ("for every key, OR all
// containers with that key into one 8 KiB accumulator, then canonicalise",
// roaring.c:2600-2682 + 2845-2856).  Here the same computation is a group-by-key:
//   1. gather (key, container) members of the selected bitmaps, stable radix sort by key;
//   2. split each key group into units of <= CH members;
//   3. k_many_l1: one workgroup per unit accumulates its members into an LDS bitset
//      (arrays: ds_or/ds_xor scatter by all four waves; bitsets: owner-thread word OR;
//      runs: toggle + prefix-xor rasterisation into a second LDS image, then word OR);
//   4. groups with one unit are canonicalised straight from LDS; groups with several units
//      write 8 KiB partial chunks that k_many_l2 combines (same shape as the multi-GPU
//      exchange: partial chunks -> owner -> combine -> canonicalise).
#pragma once
#include "rhip_kernels.h"

struct ManyView {
    const u64* skey;        // [M] sorted member keys
    const uint32_t* sval;   // [M] member container index (into the pool directory)
    const u64* gstart;      // [G+1] first member of each group
    const u64* ustart;      // [G+1] first unit of each group
    const uint32_t* n_groups;  // device scalar
    uint32_t ch;            // members per unit
};

// wave per selected bitmap: append its (key, container index) members
__global__ __launch_bounds__(256) void k_many_gather(PoolView P, const uint32_t* __restrict__ ids,
                                                     const u64* __restrict__ sel_start, uint32_t nsel,
                                                     u64* __restrict__ mkey, uint32_t* __restrict__ mval) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= nsel) return;
    const uint32_t b = ids[s];
    const u64 c0 = P.bm_start[b], c1 = P.bm_start[b + 1], d0 = sel_start[s];
    for (u64 i = c0 + lane_id(); i < c1; i += 64) {
        mkey[d0 + (i - c0)] = P.key[i];
        mval[d0 + (i - c0)] = (uint32_t)i;
    }
}

__global__ void k_many_heads(const u64* __restrict__ skey, u64 M, uint32_t* __restrict__ flag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) flag[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1u : 0u;
    if (i == M) flag[i] = 0;
}
// gid = exclusive scan of flag (so head i belongs to group gid[i]); gid[M] = number of groups
__global__ void k_many_gstart(const uint32_t* __restrict__ flag, const u64* __restrict__ gid, u64 M,
                              u64* __restrict__ gstart, uint32_t* __restrict__ n_groups) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M && flag[i]) gstart[gid[i]] = i;
    if (i == M) {
        gstart[gid[M]] = M;
        *n_groups = (uint32_t)gid[M];
    }
}
// per group: number of units; also the group's key
__global__ void k_many_units(const u64* __restrict__ gstart, const u64* __restrict__ skey,
                             const uint32_t* __restrict__ n_groups, uint32_t ch, uint32_t* __restrict__ nunits,
                             u64* __restrict__ gkey, u64 maxg) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 G = *n_groups;
    if (g < G) {
        u64 cnt = gstart[g + 1] - gstart[g];
        nunits[g] = (uint32_t)((cnt + ch - 1) / ch);
        gkey[g] = skey[gstart[g]];
    } else if (g <= maxg) {
        nunits[g] = 0;
    }
}

// unit -> group lookup: largest g with ustart[g] <= u
__device__ __forceinline__ uint32_t unit_group(const u64* __restrict__ ustart, uint32_t G, u64 u) {
    u64 lo = 0, hi = G;  // ustart[G] = total units
    while (lo + 1 < hi) {
        u64 mid = (lo + hi) >> 1;
        if (ustart[mid] <= u) lo = mid;
        else hi = mid;
    }
    return (uint32_t)lo;
}

// wave per unit: sum of member cardinalities -> gcard[group]; single-member groups also
// record their payload size (pass-through slot)
// ... and, for the replay of roaring_bitmap_or_many's full-union typing (full_union_is_run), where the group's LAST
// full-run member and LAST bitset member sit (position relative to the group start, +1; 0 = none): glast[2g],
// glast[2g+1].  A key present in every bitmap of a 100 000-bitmap set has 100 000 members (BASELINE config C4's key
// 0): finding these two positions inside the finalising workgroup cost it 0.3 ms of dependent loads.
__global__ __launch_bounds__(256) void k_many_cardsum(PoolView P, ManyView V, const u64* __restrict__ n_units,
                                                      u64* __restrict__ gcard, uint32_t* __restrict__ glast) {
    const u64 u = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= *n_units) return;
    const uint32_t G = *V.n_groups;
    const uint32_t g = unit_group(V.ustart, G, u);
    const u64 m0 = V.gstart[g] + (u - V.ustart[g]) * V.ch;
    const u64 m1 = (m0 + V.ch < V.gstart[g + 1]) ? m0 + V.ch : V.gstart[g + 1];
    u64 s = 0;
    uint32_t lrf = 0, lb = 0;
    const u64 gs = V.gstart[g];
    for (u64 m = m0 + lane_id(); m < m1; m += 64) {
        const uint32_t c = V.sval[m];
        const uint32_t cd = P.card[c];
        const uint8_t ty = P.type[c];
        s += cd;
        if (ty == T_RUN && cd == 65536u) lrf = (uint32_t)(m - gs) + 1u;
        if (ty == T_BITSET) lb = (uint32_t)(m - gs) + 1u;
    }
    s = wave_sum64(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(lrf, o), b = __shfl_xor(lb, o);
        lrf = a > lrf ? a : lrf;
        lb = b > lb ? b : lb;
    }
    if (lane_id() == 0) {
        atomicAdd(&gcard[g], s);
        if (lrf) atomicMax(&glast[2 * (u64)g], lrf);
        if (lb) atomicMax(&glast[2 * (u64)g + 1], lb);
    }
}

// slot size of every group (upper bound on the canonical result payload)
__global__ void k_many_slots(PoolView P, ManyView V, const u64* __restrict__ gcard, int force_typed,
                             uint32_t* __restrict__ slot, u64 maxg) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 G = *V.n_groups;
    if (g < G) {
        u64 cnt = V.gstart[g + 1] - V.gstart[g];
        uint32_t sz;
        if (cnt == 1 && !force_typed) {
            uint32_t c = V.sval[V.gstart[g]];
            sz = align16(payload_bytes(P.type[c], P.card[c], P.nruns[c]));
        } else {
            u64 ub = gcard[g] > 65536ull ? 65536ull : gcard[g];
            sz = align16((uint32_t)(2 * ub > 8192 ? 8192 : 2 * ub));
        }
        slot[g] = sz < 16u ? 16u : sz;
    } else if (g <= maxg) {
        slot[g] = 0;
    }
}

// Accumulate the members [m0, m1) into the LDS image acc (zeroed by the caller).
// Member lists built during phase A (relative member indices; a unit has at most 1024 members)
struct ManyLists {
    uint32_t n_bitset, n_run;
    uint16_t bitset[1024], run[1024];
};
// During the array scatter the image is addressed through an XOR swizzle of the low 5 word-index bits:
// arrays whose values are spaced by a multiple of 1024 (any regular stride, e.g. the stratified C4 data)
// would otherwise put every lane of a ds_or on the same LDS bank.  (On C4 itself the kernel is bound by
// the random 512-byte member gathers -- ~2 TB/s incl. directory sectors -- so this is insurance, not a
// measured win.)  The swizzle is an involution inside each aligned 32-word block: one pass converts
// either way.
__device__ __forceinline__ uint32_t mswz(uint32_t w) { return w ^ ((w >> 5) & 31u); }
__device__ __forceinline__ void many_swizzle_pass(uint32_t* acc) {
    const uint32_t tid = threadIdx.x;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = acc[mswz(8u * tid + k)];
    __syncthreads();
    ((uint4*)acc)[2 * tid] = make_uint4(w[0], w[1], w[2], w[3]);
    ((uint4*)acc)[2 * tid + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    __syncthreads();
}

__device__ void many_accumulate_chunk(uint32_t* acc, uint32_t* tmp, const PoolView& P, const ManyView& V, u64 m0,
                                      u64 m1, int op, BlockScratch* sc, ManyLists* ml) {
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    __syncthreads();
    if (tid == 0) { ml->n_bitset = 0; ml->n_run = 0; }
    many_swizzle_pass(acc);  // linear -> swizzled (ends with a barrier)
    // phase A: array members, LDS atomics (commutative: no ordering needed).  Each wave takes 64
    // members at a time: their directory entries are fetched lane-parallel (one member per lane) and
    // broadcast with shuffles, and members are consumed four at a time so that four independent
    // 16-byte payload loads per lane are in flight before the first LDS atomic needs one of them
    // (a member is a few hundred bytes at a random arena offset: this loop is latency-, not
    // bandwidth-limited unless loads overlap).
    for (u64 mb = m0 + 64ull * wave; mb < m1; mb += 256) {
        const u64 mi = mb + lane;
        uint32_t cd = 0, of_lo = 0, of_hi = 0;
        if (mi < m1) {
            const uint32_t c = V.sval[mi];
            const uint8_t ty = P.type[c];
            if (ty == T_ARRAY) {
                cd = P.card[c];
                const u64 of = P.off[c];
                of_lo = (uint32_t)of; of_hi = (uint32_t)(of >> 32);
            } else if (ty == T_BITSET) {
                ml->bitset[atomicAdd(&ml->n_bitset, 1u)] = (uint16_t)(mi - m0);
            } else {
                ml->run[atomicAdd(&ml->n_run, 1u)] = (uint16_t)(mi - m0);
            }
        }
        const uint32_t cnt = (uint32_t)((m1 - mb) < 64 ? (m1 - mb) : 64);
        for (uint32_t k = 0; k < cnt; k += 4) {
            uint4 q4[4];
            uint32_t cdk[4];
            const uint4* pk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t src = (k + u < cnt) ? k + u : k;
                cdk[u] = (k + u < cnt) ? __shfl(cd, src) : 0u;
                const u64 of = (u64)__shfl(of_lo, src) | ((u64)__shfl(of_hi, src) << 32);
                pk[u] = (const uint4*)(P.arena + of);
                q4[u] = (8 * lane < cdk[u]) ? pk[u][lane] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                for (uint32_t i = lane; 8 * i < cdk[u]; i += 64) {
                    const uint4 x = (i == lane) ? q4[u] : pk[u][i];
                    const uint32_t d[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        if (8 * i + h < cdk[u]) {
                            const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                            if (op == OP_OR) atomicOr(&acc[mswz(v >> 5)], 1u << (v & 31));
                            else atomicXor(&acc[mswz(v >> 5)], 1u << (v & 31));
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    many_swizzle_pass(acc);  // swizzled -> linear
    __syncthreads();
    // phase B: bitset members (listed by phase A), thread-owned words.  XOR/OR are commutative, so the
    // arbitrary list order is fine.
    {
        const uint32_t nb = ml->n_bitset;
        if (nb) {
            uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
            for (uint32_t k = 0; k < nb; ++k) {
                const uint32_t c = V.sval[m0 + ml->bitset[k]];
                const uint4* __restrict__ g = (const uint4*)(P.arena + P.off[c]);
                r0 = op4(op, r0, g[2 * tid]);
                r1 = op4(op, r1, g[2 * tid + 1]);
            }
            ((uint4*)acc)[2 * tid] = r0;
            ((uint4*)acc)[2 * tid + 1] = r1;
        }
    }
    __syncthreads();
    // phase C: run members, rasterised one at a time into tmp
    {
        const uint32_t nr = ml->n_run;
        for (uint32_t k = 0; k < nr; ++k) {
            const uint32_t c = V.sval[m0 + ml->run[k]];
            lds_load(tmp, P, c, sc);
            uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
            uint4 x0 = ((uint4*)tmp)[2 * tid], x1 = ((uint4*)tmp)[2 * tid + 1];
            ((uint4*)acc)[2 * tid] = op4(op, r0, x0);
            ((uint4*)acc)[2 * tid + 1] = op4(op, r1, x1);
            __syncthreads();
        }
    }
}

// Accumulate the members [m0, m1) into the LDS image acc (zeroed by the caller), 1024 members at a time.
__device__ void many_accumulate(uint32_t* acc, uint32_t* tmp, const PoolView& P, const ManyView& V, u64 m0, u64 m1,
                                int op, BlockScratch* sc, ManyLists* ml) {
    for (u64 c0 = m0; c0 < m1; c0 += 1024) many_accumulate_chunk(acc, tmp, P, V, c0, (c0 + 1024 < m1) ? c0 + 1024 : m1, op, sc, ml);
}

struct ManyOut {
    OutView O;          // candidate directory indexed by group
    u64* partial;       // [n_units][1024] scratch chunks (multi-unit groups)
    u64* chunk_out;     // partial_mode: [G][1024] final uncompressed chunks
    int partial_mode;   // 1: write uncompressed chunks instead of canonical containers
    int force_typed;    // 1: single-member groups are typed by cardinality too
    int exact_or_many;  // 1: reproduce roaring_bitmap_or_many's run-vs-bitset choice for FULL containers
    const uint32_t* glast;  // [2 G] last full-run / last bitset member of every group (k_many_cardsum)
    u64 first_lo, first_hi, second_lo, second_hi;  // container index ranges of ids[0] and ids[1]
};

// A union that fills the whole chunk is the one place where roaring_bitmap_or_many's result type
// depends on the fold order (SURVEY G11 / Appendix A "or_many"): the accumulator is a lazy bitset
// whose cardinality is only computed by bitset x bitset steps (container_lazy_ior, containers.h:
// 1343-1352, which turn a full result into a full RUN), a full run operand replaces it by a full run
// (containers.h:1407-1412), and a known-full accumulator short-circuits later steps (roaring.c:2621).
// Given that the final union IS full, the outcome follows from member metadata plus ONE question:
// "is the union already full after the last bitset member?" -- answered by re-accumulating that prefix.
// Returns true for a full run, false for a (full) bitset.
__device__ bool full_union_is_run(uint32_t* acc2, uint32_t* tmp, const PoolView& P, const ManyView& V,
                                  const ManyOut& MO, uint32_t g, u64 gs, u64 ge, BlockScratch* sc, ManyLists* ml) {
    const uint32_t c0 = V.sval[gs], c1 = V.sval[gs + 1];
    const bool first = c0 >= MO.first_lo && c0 < MO.first_hi && c1 >= MO.second_lo && c1 < MO.second_hi;
    auto isB = [&](uint32_t c) { return P.type[c] == T_BITSET; };
    auto isRF = [&](uint32_t c) { return P.type[c] == T_RUN && P.card[c] == 65536u; };
    u64 start;  // first member handled by the generic lazy_or_inplace step
    if (first) {  // roaring_bitmap_lazy_or(x0, x1), roaring.c:2529-2548
        if (isB(c0) || isB(c1)) { if (isRF(c0) || isRF(c1)) return true; }
        else if (isRF(c1)) return true;
        start = gs + 2;
    } else {
        if (isRF(c0)) return true;                          // container_is_full -> every step skipped
        if (isB(c0) && P.card[c0] == 65536u) return false;  // known-full bitset: skipped, repair keeps a bitset
        start = gs + 1;
    }
    // Any full run among the members [start, ge) wins (it is never skipped: the accumulator's cardinality is unknown
    // or below 65536 when it arrives); otherwise the LAST bitset member decides.  Both positions were recorded by
    // k_many_cardsum (relative to gs, +1).
    const uint32_t lrf = MO.glast[2 * (u64)g], lb_all = MO.glast[2 * (u64)g + 1];
    if (lrf && gs + lrf - 1u >= start) return true;
    const uint32_t lb = (lb_all && gs + lb_all - 1u >= start) ? lb_all : 0u;
    if (!lb) return false;
    const u64 last_b = gs + lb - 1u;
    // union of members [gs, last_b] full?
    __syncthreads();
    lds_zero(acc2);
    __syncthreads();
    many_accumulate(acc2, tmp, P, V, gs, last_b + 1, OP_OR, sc, ml);
    uint4 r0 = ((uint4*)acc2)[2 * threadIdx.x], r1 = ((uint4*)acc2)[2 * threadIdx.x + 1];
    return blk_sum(popc4(r0) + popc4(r1), sc->wsum) == 65536u;
}

// canonicalise the LDS image of a finished group: card <= 4096 -> array, else bitset
// (container_repair_after_lazy, containers.h:344-371); empty -> dropped by compaction
__device__ void many_finalize(uint32_t* acc, uint16_t* stage, const ManyOut& MO, uint32_t g, BlockScratch* sc,
                              uint32_t* acc2, uint32_t* tmp, const PoolView& P, const ManyView& V, u64 gs, u64 ge,
                              ManyLists* ml) {
    const uint32_t tid = threadIdx.x;
    uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
    if (MO.partial_mode) {
        uint4* po = (uint4*)(MO.chunk_out + (u64)g * 1024ull);
        po[2 * tid] = r0;
        po[2 * tid + 1] = r1;
        return;
    }
    uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const uint32_t rc = blk_sum(popc4(r0) + popc4(r1), sc->wsum);
    int ty = T_ARRAY;
    if (rc == 65536u && MO.exact_or_many && ge - gs >= 2 && full_union_is_run(acc2, tmp, P, V, MO, g, gs, ge, sc, ml)) {
        if (tid == 0) {
            *(uint32_t*)(MO.O.arena + MO.O.off[g]) = 0xFFFF0000u;  // one run {value 0, length 0xFFFF}
            MO.O.meta[g] = pack_meta(T_RUN, 65536u, 1u);
        }
        return;
    }
    if (rc) {
        ty = type_ba(rc);
        lds_emit(acc, r, ty, rc, 0, stage, MO.O.arena + MO.O.off[g], sc);
    }
    if (tid == 0) MO.O.meta[g] = pack_meta(ty, rc, 0);
}

__global__ __launch_bounds__(256) void k_many_l1(PoolView P, ManyView V, ManyOut MO, const u64* __restrict__ n_units,
                                                 int op) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint32_t tmp[2048];
    __shared__ __attribute__((aligned(16))) uint16_t stage[4096 + 8];  // doubles as the replay image (8 KiB)
    __shared__ BlockScratch sc;
    __shared__ ManyLists ml;
    const uint32_t G = *V.n_groups;
    const u64 U = *n_units;
    for (u64 u = blockIdx.x; u < U; u += gridDim.x) {
        const uint32_t g = unit_group(V.ustart, G, u);
        const u64 gs = V.gstart[g], ge = V.gstart[g + 1];
        const u64 nu = V.ustart[g + 1] - V.ustart[g];
        if (nu == 1 && (ge - gs) == 1 && !MO.force_typed && !MO.partial_mode) continue;  // pass-through copy path
        const u64 m0 = gs + (u - V.ustart[g]) * V.ch;
        const u64 m1 = (m0 + V.ch < ge) ? m0 + V.ch : ge;
        __syncthreads();
        lds_zero(acc);
        __syncthreads();
        many_accumulate(acc, tmp, P, V, m0, m1, op, &sc, &ml);
        if (nu == 1) {
            many_finalize(acc, stage, MO, g, &sc, (uint32_t*)stage, tmp, P, V, gs, ge, &ml);
        } else {
            uint4* po = (uint4*)(MO.partial + u * 1024ull);
            po[2 * threadIdx.x] = ((uint4*)acc)[2 * threadIdx.x];
            po[2 * threadIdx.x + 1] = ((uint4*)acc)[2 * threadIdx.x + 1];
        }
    }
}

// combine the partial chunks of multi-unit groups
__global__ __launch_bounds__(256) void k_many_l2(PoolView P, ManyView V, ManyOut MO, int op) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint32_t tmp[2048];
    __shared__ __attribute__((aligned(16))) uint16_t stage[4096 + 8];
    __shared__ BlockScratch sc;
    __shared__ ManyLists ml;
    const uint32_t G = *V.n_groups;
    const uint32_t tid = threadIdx.x;
    for (uint32_t g = blockIdx.x; g < G; g += gridDim.x) {
        const u64 u0 = V.ustart[g], u1 = V.ustart[g + 1];
        if (u1 - u0 < 2) continue;
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
        for (u64 u = u0; u < u1; ++u) {
            const uint4* __restrict__ p = (const uint4*)(MO.partial + u * 1024ull);
            r0 = op4(op, r0, p[2 * tid]);
            r1 = op4(op, r1, p[2 * tid + 1]);
        }
        __syncthreads();
        ((uint4*)acc)[2 * tid] = r0;
        ((uint4*)acc)[2 * tid + 1] = r1;
        __syncthreads();
        many_finalize(acc, stage, MO, g, &sc, (uint32_t*)stage, tmp, P, V, V.gstart[g], V.gstart[g + 1], &ml);
    }
}

// single-member groups keep their container unchanged (type included): roaring.c:2660-2676
__global__ __launch_bounds__(256) void k_many_copy(PoolView P, ManyView V, ManyOut MO) {
    const uint32_t G = *V.n_groups;
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < G; g += nwaves) {
        if (V.gstart[g + 1] - V.gstart[g] != 1) continue;
        const uint32_t c = V.sval[V.gstart[g]];
        const uint8_t ty = P.type[c];
        const uint32_t card = P.card[c], nr = P.nruns[c];
        const uint32_t n16 = (payload_bytes(ty, card, nr) + 15u) >> 4;
        const uint4* __restrict__ ps = (const uint4*)(P.arena + P.off[c]);
        uint4* __restrict__ po = (uint4*)(MO.O.arena + MO.O.off[g]);
        for (uint32_t i = lane; i < n16; i += 64) po[i] = ps[i];
        if (lane == 0) MO.O.meta[g] = pack_meta(ty, card, nr);
    }
}

__global__ void k_iota64(u64* p, u64 n, u64 mul) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * mul;
}
__global__ void k_fill8(uint8_t* p, u64 n, uint8_t v) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_fill32(uint32_t* p, u64 n, uint32_t v) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
