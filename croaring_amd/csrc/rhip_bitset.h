// rhip_bitset.h -- bitset x bitset streaming kernel (the HBM-roofline kernel), pass-through copy, synthetic C2 pool
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ bitset x bitset (K1-K3)
// One wave per container pair, persistent waves striding over the queue.  Each lane issues
// 16 independent 16-byte loads (8 per operand) before the first use: 16 KiB in flight per
// wave.  Result words stay in registers; popcount is fused; the typed result is written once.
// Replaces bitset_container_{and,or,xor,andnot}{,_nocard,_justcard} (src/containers/bitset.c:
// 343-942) and the two-pass justcard->nocard structure of mixed_intersection.c:305-325.
__device__ __forceinline__ uint4 op4(int op, uint4 a, uint4 b) {
    uint4 r;
    switch (op) {
        case OP_AND: r.x = a.x & b.x; r.y = a.y & b.y; r.z = a.z & b.z; r.w = a.w & b.w; break;
        case OP_OR: r.x = a.x | b.x; r.y = a.y | b.y; r.z = a.z | b.z; r.w = a.w | b.w; break;
        case OP_XOR: r.x = a.x ^ b.x; r.y = a.y ^ b.y; r.z = a.z ^ b.z; r.w = a.w ^ b.w; break;
        default: r.x = a.x & ~b.x; r.y = a.y & ~b.y; r.z = a.z & ~b.z; r.w = a.w & ~b.w; break;
    }
    return r;
}
__device__ __forceinline__ uint32_t popc4(uint4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

// native 128-bit vector for the streaming kernel (the nontemporal builtins need a native vector type)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int OP>
__device__ __forceinline__ u32x4 vop(u32x4 a, u32x4 b) {
    if (OP == OP_AND) return a & b;
    if (OP == OP_OR) return a | b;
    if (OP == OP_XOR) return a ^ b;
    return a & ~b;
}
__device__ __forceinline__ u32x4 vop_rt(int op, u32x4 a, u32x4 b) {  // op known per item only (multi-op batches)
    return op == OP_AND ? (a & b) : op == OP_OR ? (a | b) : op == OP_XOR ? (a ^ b) : (a & ~b);
}
template <int OP>
__device__ __forceinline__ u32x4 vop_any(int op, u32x4 a, u32x4 b) { return OP == OP_ITEM ? vop_rt(op, a, b) : vop<OP>(a, b); }
__device__ __forceinline__ uint32_t vpopc(u32x4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

template <int OP>
__device__ __forceinline__ void bb_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                            OutView O, const BBItem* __restrict__ q, const u64* __restrict__ qrange,
                                            int cardmode, u64* pair_acc, GenItem* retry_q, uint32_t* retry_count) {
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    if (cardmode) {
        // Cardinality only (bitset_container_and_justcard, bitset.c:421-430, under roaring_bitmap_and_cardinality,
        // roaring.c:3048-3076): nothing is written but the pair's sum.  The queue is in pair order -- the 4 096
        // container pairs of one bitmap pair are neighbours -- so a wave takes a CONTIGUOUS stretch of it and adds in a
        // register, with one atomic per pair it meets instead of one per container pair (4 096 same-address 64-bit
        // atomics per bitmap pair, 250 hot addresses per launch: the read-only form was slower per byte than `and`
        // with its 8 GiB of stores).
        const uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
        const uint32_t per = (n + nwaves - 1u) / nwaves;
        const uint32_t i0 = w * per < n ? w * per : n, i1 = i0 + per < n ? i0 + per : n;
        u64 acc = 0;
        for (uint32_t i = i0; i < i1; ++i) {
            const BBItem t = q[i];
            const int op = item_op(OP, t.slot);
            const u32x4* __restrict__ pa = (const u32x4*)(arenaA + t.offa);
            const u32x4* __restrict__ pb = (const u32x4*)(arenaB + t.offb);
            u32x4 va[8], vb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) va[k] = __builtin_nontemporal_load(pa + k * 64 + lane);
#pragma unroll
            for (int k = 0; k < 8; ++k) vb[k] = __builtin_nontemporal_load(pb + k * 64 + lane);
            uint32_t cnt = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) cnt += vpopc(vop_any<OP>(op, va[k], vb[k]));
            acc += cnt;  // per-lane partial sums: ONE wave reduction per pair, when its last item of this stretch is in
            if (i + 1 == i1 || q[i + 1].out != t.out) {  // (wave-uniform: the items are)
                const u64 s = wave_sum64(acc);
                if (s && lane == 0) atomicAdd(&pair_acc[t.out], s);
                acc = 0;
            }
        }
        return;
    }
    for (uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6); w < n; w += nwaves) {
        const BBItem t = q[w];
        const int op = item_op(OP, t.slot);            // (OP_ITEM: the item's own op, wave-uniform)
        const uint32_t slot = t.slot & 0xFFFFu;
        const bool is_or = OP == OP_ITEM ? op == OP_OR : OP == OP_OR;
        const u32x4* __restrict__ pa = (const u32x4*)(arenaA + t.offa);
        const u32x4* __restrict__ pb = (const u32x4*)(arenaB + t.offb);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = __builtin_nontemporal_load(pa + i * 64 + lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = __builtin_nontemporal_load(pb + i * 64 + lane);
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            va[i] = vop_any<OP>(op, va[i], vb[i]);
            cnt += vpopc(va[i]);
        }
        // The result words leave as soon as they exist: whenever the slot can hold a bitset (8192 bytes: always for
        // OR; for and / xor / andnot unless the operands are too sparse for a result above 4096 values) the eight
        // stores are issued BEFORE the cardinality reduction resolves, so they overlap it.  If the result then turns
        // out to be an array (card <= 4096) the retry pass rewrites the slot; a smaller slot can never need a bitset.
        if (is_or || slot >= 8192u) {
            u32x4* __restrict__ po = (u32x4*)(O.arena + t.offo);
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(va[i], po + i * 64 + lane);
        }
        const uint32_t card = wave_sum(cnt);
        // result typing: OR is always a bitset (containers.h:1015-1020); and/xor/andnot are a
        // bitset iff card > 4096 (mixed_intersection.c:305-325, mixed_xor.c:260-273,
        // mixed_andnot.c:482-497)
        if (is_or || card > 4096u) {
            if (lane == 0) O.meta[t.out] = pack_meta(T_BITSET, card, 0);
        } else if (card == 0) {
            if (lane == 0) O.meta[t.out] = pack_meta(T_ARRAY, 0, 0);
        } else {
            // rare: result becomes an array -> re-queue for the LDS extraction kernel
            if (lane == 0) {
                GenItem g;
                g.offa = t.offa; g.offb = t.offb; g.out = t.out; g.ca = 65536u; g.cb = 65536u;
                g.types = (uint32_t)T_BITSET | ((uint32_t)T_BITSET << 8) | ((uint32_t)op << ITEM_OP_SHIFT);
                g.nra = 0; g.nrb = 0; g.offo = t.offo;
                retry_q[atomicAdd(retry_count, 1u)] = g;
            }
        }
    }
}
template <int OP>
__global__ __launch_bounds__(256) void k_bb(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                            OutView O, const BBItem* __restrict__ q, const u64* __restrict__ qrange,
                                            int cardmode, u64* pair_acc, GenItem* retry_q, uint32_t* retry_count) {
    uint32_t* lds = nullptr;
    bb_body<OP>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, cardmode, pair_acc, retry_q, retry_count);
}


// ------------------------------------------------------------------ bitset x bitset -> (mostly) array
// and / andnot of two bitsets whose result is expected to hold at most 4096 values (three quarters of the bitset pairs
// of weather_sept_85 under and): one wave per pair reads both operands once, and writes the result in whichever form
// the reference's rule picks (card <= 4096 -> array via bitset extraction, mixed_intersection.c:305-325,
// mixed_andnot.c:482-497; else the bitset, straight from registers) -- no second pass over the operands.
// Lane l holds the 16-byte groups i * 64 + l (i = 0..7): segment i covers values [8192 i, 8192 i + 8192), so the
// sorted output is segment by segment, inside a segment lane by lane (wave prefix of popcounts), inside a lane bit by
// bit; values are compacted into a wave-private 8 KiB LDS buffer and leave with coalesced 16-byte stores.
template <int OP>
__device__ __forceinline__ void bba_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                             OutView O, const BBItem* __restrict__ q, const u64* __restrict__ qrange) {
    const uint32_t lane = lane_id();
    uint16_t* st16 = (uint16_t*)(lds + (threadIdx.x >> 6) * 2048u);
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    for (uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6); w < n; w += nwaves) {
        const BBItem t = q[w];
        const int op = item_op(OP, t.slot);
        const u32x4* __restrict__ pa = (const u32x4*)(arenaA + t.offa);
        const u32x4* __restrict__ pb = (const u32x4*)(arenaB + t.offb);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = pa[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = pb[i * 64 + lane];
        uint32_t cnt[8], tot = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            va[i] = vop_any<OP>(op, va[i], vb[i]);
            cnt[i] = vpopc(va[i]);
            tot += cnt[i];
        }
        const uint32_t card = wave_sum(tot);
        uint8_t* outp = O.arena + t.offo;
        if (card > 4096u) {
            u32x4* __restrict__ po = (u32x4*)outp;
#pragma unroll
            for (int i = 0; i < 8; ++i) out_store16(&po[i * 64 + lane], va[i]);
            if (lane == 0) O.meta[t.out] = pack_meta(T_BITSET, card, 0);
            continue;
        }
        if (card) {
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t inc = wave_incl_scan(cnt[i]);
                uint32_t pos = run + inc - cnt[i];
                run += wave_lane<63>(inc);
                const uint32_t wd[4] = {va[i].x, va[i].y, va[i].z, va[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t x = wd[j];
                    const uint32_t vbase = (4u * (i * 64 + lane) + j) * 32u;
                    while (x) {
                        st16[pos++] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                        x &= x - 1;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n16 = (2u * card + 15u) >> 4;
            uint4* __restrict__ po = (uint4*)outp;
            for (uint32_t i = lane; i < n16; i += 64) out_store16(&po[i], ((const uint4*)st16)[i]);
        }
        if (lane == 0) O.meta[t.out] = pack_meta(T_ARRAY, card, 0);
        __builtin_amdgcn_wave_barrier();  // the staging buffer is reused by the next item
    }
}
template <int OP>
__global__ __launch_bounds__(256) void k_bba(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                             OutView O, const BBItem* __restrict__ q, const u64* __restrict__ qrange) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
    bba_body<OP>(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange);
}


// ------------------------------------------------------------------ placement probe (rhip_engine.hip, place_arena)
// The access pattern of k_bb without a queue: wave-item i reads two 8 KiB containers of the operand arena -- at item
// order and at a permuted position, as the left and right operands of a batch do -- and writes their OR to slot i of a
// candidate result arena, all non-temporal.  Its time tells how the candidate's PHYSICAL pages sit relative to the
// operand's (the same kernel streams at 5.4, 5.8 or 6.3 TB/s depending on that distance).
__global__ __launch_bounds__(256) void k_place_probe(const uint8_t* __restrict__ arenaA, u64 a_items, uint8_t* __restrict__ out, u64 n_slots,
                                                     u64 stride) {
    // every stride-th slot of the WHOLE candidate (an allocation is composed of several physical blocks, each at its own
    // distance from the operand: probing its first gigabyte says nothing about the rest)
    const uint32_t lane = lane_id();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 i = wave_uniform((uint32_t)(((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6)) * stride; i < n_slots; i += nwaves * stride) {
        const u32x4* __restrict__ pa = (const u32x4*)(arenaA + (i % a_items) * 8192ull);
        const u32x4* __restrict__ pb = (const u32x4*)(arenaA + ((i * 97ull + 4096ull * 33ull) % a_items) * 8192ull);
        u32x4 va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = __builtin_nontemporal_load(pa + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 8; ++k) vb[k] = __builtin_nontemporal_load(pb + k * 64 + lane);
        u32x4* __restrict__ po = (u32x4*)(out + i * 8192ull);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(va[k] | vb[k], po + k * 64 + lane);
    }
}

// ------------------------------------------------------------------ pass-through copy
// The item says where from, where to, how much -- no directory loads.  A wave takes `per_wave` items at a time: FOUR
// (a quarter-wave each, 16 bytes per lane, while none of them exceeds 256 bytes; else the whole wave copies them one
// after the other), or SIXTEEN when the host expects tiny containers (the operand pools average <= 96 payload bytes per
// container: C5, wikileaks -- round 4: at four per wave k_copy was 226 us of a C5 `or` batch, three lanes in four
// idle and one dependent load chain per four items): four lanes copy one item each when all sixteen are <= 64 bytes,
// otherwise the sixteen go through the four-item form in four rounds.
__device__ __forceinline__ void copy_body(uint32_t* __restrict__ lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const CopyItem* __restrict__ q,
                                              const u64* __restrict__ qrange, uint32_t per_wave) {
    const uint32_t lane = lane_id(), grp = lane >> 4, gl = lane & 15u;
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = wave_uniform((bid * blockDim.x + threadIdx.x) >> 6);
    const bool wide = per_wave == 16u;                       // (wave-uniform: a kernel argument)
    const uint32_t lsh = wide ? 2u : 4u;                     // lanes per item = 1 << lsh
    const uint32_t mine = lane >> lsh, li = lane & ((1u << lsh) - 1u);
    CopyItem tn;
    tn.src = 0; tn.offo = 0; tn.meta = 0; tn.n16 = 0; tn.out = 0;
    if (per_wave * w + mine < n) tn = q[per_wave * w + mine];
    for (; per_wave * w < n; w += nwaves) {
        const uint32_t i0 = per_wave * w;
        const bool have = i0 + mine < n;
        const CopyItem t = tn;
        if (per_wave * (w + nwaves) + mine < n) tn = q[per_wave * (w + nwaves) + mine];  // the next items: in flight during the copy
        if (__ballot(have && t.n16 > (1u << lsh)) == 0) {  // every item fits its lanes
            if (have && li < t.n16) {
                const uint8_t* base = (t.src & COPY_FROM_B) ? arenaB : arenaA;
                out_store16(&((uint4*)(O.arena + t.offo))[li], ((const uint4*)(base + (t.src & ~COPY_FROM_B)))[li]);
            }
        } else {
            const u64 huge = __ballot(have && t.n16 > 16u);
            const uint32_t rounds = per_wave >> 2;
            for (uint32_t r = 0; r < rounds; ++r) {  // items 4 r .. 4 r + 3 (held by lanes 16 r .. 16 r + 15 in either form)
                if (i0 + 4 * r >= n) break;
                const u64 hr = wide ? ((huge >> (16 * r)) & 0xFFFFull) : huge;
                if (hr == 0) {
                    const uint32_t j = 4 * r + grp;  // this quarter-wave's item
                    const u64 src = __shfl(t.src, (int)(j << lsh)), offo = __shfl(t.offo, (int)(j << lsh));
                    const uint32_t n16 = __shfl(t.n16, (int)(j << lsh));
                    if (i0 + j < n && gl < n16) {
                        const uint8_t* base = (src & COPY_FROM_B) ? arenaB : arenaA;
                        out_store16(&((uint4*)(O.arena + offo))[gl], ((const uint4*)(base + (src & ~COPY_FROM_B)))[gl]);
                    }
                } else {
                    for (uint32_t g = 0; g < 4; ++g) {
                        const uint32_t j = 4 * r + g;
                        if (i0 + j >= n) break;
                        const u64 src = __shfl(t.src, (int)(j << lsh)), offo = __shfl(t.offo, (int)(j << lsh));
                        const uint32_t n16 = __shfl(t.n16, (int)(j << lsh));
                        const uint8_t* base = (src & COPY_FROM_B) ? arenaB : arenaA;
                        const uint4* __restrict__ ps = (const uint4*)(base + (src & ~COPY_FROM_B));
                        uint4* __restrict__ po = (uint4*)(O.arena + offo);
                        for (uint32_t i = lane; i < n16; i += 64) out_store16(&po[i], ps[i]);
                    }
                }
            }
        }
        if (have && li == 0) O.meta[t.out] = t.meta;
    }
}
__global__ __launch_bounds__(256) void k_copy(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const CopyItem* __restrict__ q,
                                              const u64* __restrict__ qrange, uint32_t per_wave) {
    uint32_t* lds = nullptr;
    copy_body(lds, blockIdx.x, gridDim.x, arenaA, arenaB, O, q, qrange, per_wave);
}



// ------------------------------------------------------------------ synthetic C2 pool
__device__ __forceinline__ u64 splitmix64_at(u64 seed, u64 idx) {  // idx-th output (1-based) of splitmix64(seed)
    u64 z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_synth_fill(u64* words, uint32_t n_bitmaps, uint32_t n_containers, u64 seed) {
    const u64 per_bm = (u64)n_containers * 1024ull;
    const u64 total = per_bm * n_bitmaps;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        u64 b = i / per_bm, w = i % per_bm;
        words[i] = splitmix64_at(seed + b, w + 1);
    }
}
__global__ __launch_bounds__(256) void k_synth_dir(const u64* words, uint32_t n_bitmaps, uint32_t n_containers,
                                                   u64* bm_start, u64* key, uint8_t* type, uint32_t* card,
                                                   uint32_t* nruns, u64* off) {
    // one wave per container: popcount its 1024 words
    u64 c = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 total = (u64)n_bitmaps * n_containers;
    if (c >= total) return;
    const uint4* p = (const uint4*)(words + c * 1024ull);
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) cnt += popc4(p[i * 64 + lane_id()]);
    cnt = wave_sum(cnt);
    if (lane_id() == 0) {
        key[c] = c % n_containers;
        type[c] = T_BITSET;
        card[c] = cnt;
        nruns[c] = 0;
        off[c] = c * 8192ull;
        if (c % n_containers == 0) bm_start[c / n_containers] = c;
        if (c == total - 1) bm_start[n_bitmaps] = total;
    }
}
