// rhip_many.h -- many-way OR / XOR aggregation kernels (roaring_bitmap_or_many /
// roaring_bitmap_xor_many, src/roaring.c:775-809).
//
// The reference folds bitmap after bitmap into a growing answer ("for every key, OR all
// containers with that key into one 8 KiB accumulator, then canonicalise",
// roaring.c:2600-2682 + 2845-2856; every step inserts into the answer's sorted directory,
// roaring_array.c:348-367).  Here the same computation is a group-by-key with ONE host wait at the end.
//
// Grouping, 16-bit keys (32-bit bitmaps) -- a COUNTING SORT over the key space, three launches, no library call:
//   k_many_hist     every member container of the selected bitmaps is counted into an LDS histogram over its key
//                   (count | result-slot weight), blocks flush their non-empty bins to the global histogram; also clears
//                   the scan / totals scratch of the call
//   k_many_keyscan  ONE workgroup walks the key space: non-empty bins become groups (start, key, partial-chunk slots,
//                   result-slot offset) and per-key scatter cursors; the histogram is returned to zero for the next call
//   k_many_scatter  the same member sets again: LDS count per key -> ONE global reservation per (block, key) -> every
//                   member's 64-bit DESCRIPTOR (payload offset, type, size: everything the accumulation needs) lands in
//                   its group.  The order inside a group is whatever the atomics give, so every member also carries its
//                   TAG (position in the gathered order): the one place where the reference's result depends on the
//                   fold order (the type of a FULL union) is replayed from the tags (full_union_decide).
// Grouping, 48-bit keys (roaring64): k_many_gather -> stable radix sort by key (rocPRIM) -> k_many_groups (one look-back
// pass: group heads, starts, keys, slot weights) -> k_many_units (look-back: partial-chunk slots, result-slot offsets);
// tags are the positions themselves.
//
// Accumulation, both:
//   k_many_l1       the grouped member list is cut into PIECES of `per` consecutive members (the host picks per = M /
//                   resident workgroups: every workgroup gets the same number of members whatever the group sizes, a
//                   100 000-member group is just 32 pieces); a workgroup walks the groups of its piece.  The array
//                   members are ONE flattened stream of 16-byte payload groups (every lane holds eight values whatever
//                   the cardinalities), four loads per lane in flight while the previous four feed the LDS atomics;
//                   bitset members: owner-thread word OR; run members: toggle + prefix-xor rasterisation.  A group
//                   inside one piece is canonicalised straight from LDS, a group cut by piece boundaries writes 8 KiB
//                   partial chunks
//   k_many_l2       combines the partial chunks of cut groups (same shape as the multi-GPU exchange)
//                   (a single-member group keeps its container unchanged: copied by the workgroup that meets it)
//   k_many_tail     ONE look-back pass: drops empty (xor) results, writes the result directory, totals and the
//                   completion word into pinned host memory
// Every buffer is sized on the host from upper bounds (members, distinct keys of the pool, per-bitmap payload bounds);
// every kernel takes its counts from device memory.
#pragma once
#include "rhip_kernels.h"
#ifndef RHIP_ABL_SC   // ablation builds (scripts/gpu_exp.sh c4-variants; profiles/r05_many_l1_notes.md): 0 in the product
#define RHIP_ABL_SC 0
#endif
#ifndef RHIP_ABL_L1
#define RHIP_ABL_L1 0
#endif

// ------------------------------------------------------------------ member descriptors
// bits 0..34  payload offset / 16 (arenas up to 512 GiB)
// bits 35..36 container type
// bit  37     cardinality == 65536
// bits 38..53 n: array = cardinality (1..4096), bitset = cardinality - 1, run = number of runs (1..32768)
// bits 54..62 run only: ceil(cardinality / 256) (1..256) -- an upper bound is all the slot sizing needs
__device__ __forceinline__ u64 md_pack(u64 off, uint32_t ty, uint32_t card, uint32_t nruns) {
    const uint32_t n = ty == T_ARRAY ? card : (ty == T_BITSET ? card - 1u : nruns);
    const uint32_t rq = ty == T_RUN ? (card + 255u) >> 8 : 0u;
    return (off >> 4) | ((u64)ty << 35) | ((u64)(card == 65536u ? 1u : 0u) << 37) | ((u64)n << 38) | ((u64)rq << 54);
}
__device__ __forceinline__ u64 md_off(u64 d) { return (d & ((1ull << 35) - 1ull)) << 4; }
__device__ __forceinline__ uint32_t md_type(u64 d) { return (uint32_t)(d >> 35) & 3u; }
__device__ __forceinline__ bool md_full(u64 d) { return ((d >> 37) & 1ull) != 0; }
__device__ __forceinline__ uint32_t md_n(u64 d) { return (uint32_t)(d >> 38) & 0xFFFFu; }
__device__ __forceinline__ uint32_t md_card_ub(u64 d) {  // >= the cardinality, exact for arrays and bitsets
    const uint32_t ty = md_type(d);
    return ty == T_ARRAY ? md_n(d) : (ty == T_BITSET ? md_n(d) + 1u : ((uint32_t)(d >> 54) & 0x1FFu) << 8);
}
__device__ __forceinline__ uint32_t md_payload(u64 d) {
    const uint32_t ty = md_type(d);
    return ty == T_BITSET ? 8192u : (ty == T_ARRAY ? 2u * md_n(d) : 4u * md_n(d));
}

// device-side totals of one call (inside the zeroed scratch words); the last kernel copies them to pinned memory
struct ManyTotals {
    u64 n_groups, n_partials, slot_bytes, kept, bytes_in, bytes_out, n_type[3], max_key, err;
};
#define MANY_ERR_KEYSPACE 1ull

struct ManyView {
    const u64* sdesc;       // [M] member descriptors, grouped by key
    const uint32_t* sord;   // [M] tag (position in the gathered order) of every member; null: the members of a group ARE
                            //     in gathered order and a member's tag is its position (or MO.arrays_only: tags are never asked for)
    u64* rdesc;             // [M] scratch of the full-union replay (a group only ever uses its own range)
    const u64* gstart;      // [G+1] first member of each group; gstart[G] = M
    const u64* pstart;      // [G+1] first partial-chunk slot of each group (groups cut by piece boundaries)
    const ManyTotals* tot;  // n_groups lives here
    u64 per;                // members per piece
};
struct ManyZero {   // scratch the first kernel of a call clears for the later ones
    u64* words; u64 n_words;
    uint32_t* glast; u64 n_glast;
    u64* table; u64 n_table;   // dense stage-1 table of the sharded form (u64 words), may be null
};
// result-slot weight of a member, in 16-byte units: the slot of a group is the sum of its members' weights capped at a
// bitset (512: a run list longer than that never survives, many_pass_through).  The same figure as the host's per-bitmap bound (k_bitmap_bounds' wmany:
// runs weighed by their cardinality rounded up to a multiple of 256) -- the arena is sized from the sum of those.
__device__ __forceinline__ uint32_t many_weight16(uint32_t ty, uint32_t card, uint32_t nruns) {
    return slot_bound((uint8_t)ty, ty == T_RUN ? ((card + 255u) & ~255u) : card, nruns) >> 4;
}
// pieces a group [gs, ge) is cut into by the piece boundaries (multiples of per)
__device__ __forceinline__ u64 many_group_pieces(u64 gs, u64 ge, u64 per) { return (ge - 1) / per - gs / per + 1; }

// members of the selected bitmaps -> (key, descriptor).  ids == null: every bitmap of the pool, one thread per container.
__global__ __launch_bounds__(256) void k_many_gather(PoolView P, const uint32_t* __restrict__ ids,
                                                     const u64* __restrict__ sel_start, uint32_t nsel, u64 M,
                                                     u64* __restrict__ mkey, u64* __restrict__ mdesc, ManyZero Z) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x, nth = (u64)gridDim.x * blockDim.x;
    for (u64 i = gid; i < Z.n_words; i += nth) Z.words[i] = 0;
    for (u64 i = gid; i < Z.n_glast; i += nth) Z.glast[i] = 0;
    for (u64 i = gid; i < Z.n_table; i += nth) Z.table[i] = 0;
    if (!ids) {
        for (u64 i = gid; i < M; i += nth) {
            mkey[i] = P.key[i];
            mdesc[i] = md_pack(P.off[i], P.type[i], P.card[i], P.nruns[i]);
        }
        return;
    }
    const u64 nw = nth >> 6;
    for (u64 s = gid >> 6; s < nsel; s += nw) {
        const uint32_t b = ids[s];
        const u64 c0 = P.bm_start[b], c1 = P.bm_start[b + 1], d0 = sel_start[s];
        for (u64 i = c0 + lane_id(); i < c1; i += 64) {
            mkey[d0 + (i - c0)] = P.key[i];
            mdesc[d0 + (i - c0)] = md_pack(P.off[i], P.type[i], P.card[i], P.nruns[i]);
        }
    }
}

// ------------------------------------------------------------------ the key dictionary of a roaring64 pool
// The counting sort below works over a dense key space.  A 32-bit pool has one: its 16-bit container keys.  A roaring64
// pool's 48-bit keys (high 32 bits << 16 | key16: no ART on the device) get one -- ONCE per pool, like the other host
// mirrors: the distinct keys are collected in an open-addressing table (k_kd_insert), compacted (k_kd_compact), sorted by
// one workgroup (k_kd_sort: a pool holds a few thousand distinct keys, 65 536 at most here) and every container's key is
// replaced by its rank (k_kd_rank).  Until round 6 every many-way call over a roaring64 pool gathered (key, descriptor)
// pairs and ran a library radix sort over them (rocPRIM: four kernels + five fills) -- BASELINE configs[4]'s aggregation.
constexpr u64 KD_EMPTY = ~0ull;
__device__ __forceinline__ u64 kd_hash(u64 k) {
    k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; k ^= k >> 33;
    return k;
}
__global__ __launch_bounds__(256) void k_kd_insert(const u64* __restrict__ key, u64 n, u64* __restrict__ tab, u64 mask,
                                                   u64* __restrict__ n_distinct) {
    uint32_t mine = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = key[i];
        for (u64 h = kd_hash(k) & mask;; h = (h + 1) & mask) {
            const u64 cur = __atomic_load_n(&tab[h], __ATOMIC_RELAXED);
            if (cur == k) break;
            if (cur == KD_EMPTY) {
                const u64 old = atomicCAS((unsigned long long*)&tab[h], (unsigned long long)KD_EMPTY, (unsigned long long)k);
                if (old == KD_EMPTY) { ++mine; break; }
                if (old == k) break;
            }
        }
    }
    mine = wave_sum(mine);
    if (lane_id() == 0 && mine) atomicAdd((unsigned long long*)n_distinct, (unsigned long long)mine);
}
__global__ __launch_bounds__(256) void k_kd_compact(const u64* __restrict__ tab, u64 slots, u64* __restrict__ out, u64 cap,
                                                    u64* __restrict__ cursor) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = tab[i];
        if (k != KD_EMPTY) {
            const u64 p = atomicAdd((unsigned long long*)cursor, 1ull);
            if (p < cap) out[p] = k;
        }
    }
}
// ascending bitonic sort of v[0, n) in place, ONE workgroup; v holds n2 >= n slots (n2 = next power of two), the slots
// behind n are filled with ~0
__global__ __launch_bounds__(1024) void k_kd_sort(u64* __restrict__ v, const u64* __restrict__ n_ptr, uint32_t n2) {
    const uint32_t n = *n_ptr < (u64)n2 ? (uint32_t)*n_ptr : n2;  // (the number of distinct keys lives on the device)
    for (uint32_t i = n + threadIdx.x; i < n2; i += 1024) v[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n2; i += 1024) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const u64 a = v[i], b = v[x];
                    if ((a > b) == ((i & k) == 0u)) { v[i] = b; v[x] = a; }
                }
            }
            __syncthreads();
        }
}
__global__ __launch_bounds__(256) void k_kd_rank(const u64* __restrict__ key, u64 n, const u64* __restrict__ dict,
                                                 const u64* __restrict__ n_ptr, uint32_t n2, uint32_t* __restrict__ kid) {
    const u64 K = *n_ptr < (u64)n2 ? *n_ptr : (u64)n2;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        kid[i] = (uint32_t)lower_bound(dict, 0, K, key[i]);
}

// ------------------------------------------------------------------ grouping by counting sort (16-bit keys)
// The members of a call in GATHERED order: member t (its tag) is container `t` of the pool (ids == null), or container
// bm_start[ids[s]] + (t - sel_start[s]) of the s-th selected bitmap.  A workgroup's share: `per_block` consecutive members
// (ids == null) or `per_block` consecutive selected bitmaps, a wave per bitmap.  k_many_hist and k_many_scatter iterate
// the same way: row b of the count matrix is workgroup b's.
struct ManySel {
    const uint32_t* ids;
    const u64* sel_start;  // [nsel + 1]
    uint32_t nsel;
    u64 M;
    u64 per_block;
    uint32_t n_blocks;     // workgroups that hold members = rows of the count matrix
    const uint32_t* kid;   // 48-bit keys (roaring64 pools): the dense id of every container's key -- its rank among the pool's
                           // distinct keys (rhip_key_dict, computed once per pool) -- which the counting sort takes for the key;
                           // null: the 16-bit key itself
};
constexpr uint32_t MC_THREADS = 1024;
constexpr uint32_t MC_W1 = 8192;    // keys per LDS window of k_many_hist (u64 bins: 64 KiB)
constexpr uint32_t MC_W3 = 16384;   // keys per LDS window of k_many_scatter (u32 bins: 64 KiB)
// count matrix entry: members of (workgroup, key) in bits 0..16, their slot weight (capped) in bits 17..31
constexpr uint32_t MC_CNT_BITS = 17, MC_CNT_MASK = (1u << MC_CNT_BITS) - 1u, MC_W_CAP = (1u << (32 - MC_CNT_BITS)) - 1u;
constexpr uint32_t MC_MAX_PER_BLOCK = MC_CNT_MASK;  // members of one workgroup

// Row of the count matrix (= share of the members) workgroup b takes.  Workgroup b runs on XCD b % 8 (observed, not
// promised: a speed matter only): the workgroups of one XCD take CONSECUTIVE rows, so that what they scatter into a
// group -- a few members per (row, key), adjacent for adjacent rows -- meets in ONE L2 and leaves as whole lines
// (with rows dealt round-robin every 8-byte descriptor left its L2 as a partial line: k_many_scatter 142 us on C4).
__device__ __forceinline__ uint32_t many_row(uint32_t b, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
// One member's directory entry, loaded whole: the grouping kernels are chains of dependent loads unless a thread has
// many of them in flight (k_many_scatter: 99 -> us on C4 with eight members' loads issued before the first is used).
struct ManyRec { uint32_t key, ty, cd, nr; u64 off; };
template <bool WITH_OFF>
__device__ __forceinline__ ManyRec many_load(const PoolView& P, const uint32_t* __restrict__ kid, u64 c) {
    // (non-temporal: the directory streams through once; what should stay in the L2 are the partly written lines of
    // k_many_scatter's output, which only leave as whole lines if they survive until their neighbours arrive)
    ManyRec r;
#if RHIP_ABL_SC == 3
    r.key = (uint32_t)P.key[c] & 0xFFFFu; r.ty = P.type[c]; r.cd = P.card[c]; r.nr = P.nruns[c];
    r.off = WITH_OFF ? P.off[c] : 0ull;
#else
    r.key = kid ? __builtin_nontemporal_load(&kid[c]) : (uint32_t)__builtin_nontemporal_load(&P.key[c]) & 0xFFFFu;
    r.ty = __builtin_nontemporal_load(&P.type[c]);
    r.cd = __builtin_nontemporal_load(&P.card[c]);
    r.nr = __builtin_nontemporal_load(&P.nruns[c]);
    r.off = WITH_OFF ? __builtin_nontemporal_load(&P.off[c]) : 0ull;
#endif
    return r;
}
// f(record, tag) for every member of row `row`; eight members per thread in flight
template <bool WITH_OFF, class F>
__device__ __forceinline__ void many_for_members(const PoolView& P, const ManySel& S, uint32_t row, F f) {
    if (!S.ids) {
        const u64 lo = (u64)row * S.per_block, hi = lo + S.per_block < S.M ? lo + S.per_block : S.M;
        for (u64 base = lo + threadIdx.x; base < hi; base += 8ull * MC_THREADS) {
            ManyRec r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u64 t = base + (u64)j * MC_THREADS;
                r[j] = many_load<WITH_OFF>(P, S.kid, t < hi ? t : hi - 1);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u64 t = base + (u64)j * MC_THREADS;
                if (t < hi) f(r[j], t);
            }
        }
        return;
    }
    const u64 s0 = (u64)row * S.per_block, s1 = s0 + S.per_block < S.nsel ? s0 + S.per_block : S.nsel;
    for (u64 s = s0 + (threadIdx.x >> 6); s < s1; s += MC_THREADS / 64) {
        const uint32_t b = S.ids[s];
        const u64 c0 = P.bm_start[b], c1 = P.bm_start[b + 1], d0 = S.sel_start[s];
        for (u64 i = c0 + lane_id(); i < c1; i += 64) f(many_load<WITH_OFF>(P, S.kid, i), d0 + (i - c0));
    }
}

// Row blockIdx of the count matrix mat[n_blocks][KS]: this workgroup's members per key (count | capped slot weight).
// Every entry of the row is written (zeros included): the matrix needs no clearing.  Also clears the scan / totals
// scratch of the call.
__global__ __launch_bounds__(MC_THREADS) void k_many_hist(PoolView P, ManySel S, uint32_t KS, uint32_t* __restrict__ mat,
                                                         ManyZero Z) {
    __shared__ u64 h[MC_W1];
    const u64 gid = (u64)blockIdx.x * MC_THREADS + threadIdx.x, nth = (u64)gridDim.x * MC_THREADS;
    for (u64 i = gid; i < Z.n_words; i += nth) Z.words[i] = 0;
    for (u64 i = gid; i < Z.n_glast; i += nth) Z.glast[i] = 0;
    for (u64 i = gid; i < Z.n_table; i += nth) Z.table[i] = 0;
    if (blockIdx.x >= S.n_blocks) return;  // (blocks beyond the member blocks only help clearing)
    const uint32_t myrow = many_row(blockIdx.x, S.n_blocks);
    for (uint32_t w0 = 0; w0 < KS; w0 += MC_W1) {
        for (uint32_t i = threadIdx.x; i < MC_W1; i += MC_THREADS) h[i] = 0;
        __syncthreads();
        many_for_members<false>(P, S, myrow, [&](const ManyRec& r, u64) {
            const uint32_t k = r.key - w0;
            if (k < MC_W1) atomicAdd(&h[k], 1ull | ((u64)many_weight16(r.ty, r.cd, r.nr) << 32));
        });
        __syncthreads();
        uint32_t* __restrict__ row = mat + (u64)myrow * KS + w0;
        for (uint32_t i = threadIdx.x; i < MC_W1 && w0 + i < KS; i += MC_THREADS) {
            const u64 v = h[i];
            const uint32_t w = (v >> 32) > MC_W_CAP ? MC_W_CAP : (uint32_t)(v >> 32);
            row[i] = (uint32_t)v | (w << MC_CNT_BITS);
        }
        __syncthreads();
    }
}

// The key space in tiles of 64 keys; a workgroup = 16 waves = 16 SEGMENTS of the matrix's rows over the tile's keys.
// (1) column sums: every wave adds up its rows (loads only), the segment totals meet in LDS; (2) wave 0: non-empty keys
// become groups in key order -- two look-back chains over the key tiles: (groups, members), then, with a group's first
// member known, (partial-chunk slots, result-slot units) -- and writes gstart / gkey / pstart / off as the accumulation
// kernels read them, kstart[k] = first member position of key k, kgrp[k] = its group, and the totals; (3) every wave
// walks its rows again: rel[b][k] = members of key k in workgroups < b.
struct ManyKeyLb { u64* status_a; u64* status_b; uint32_t* ticket; };
constexpr uint32_t MC_KT = 64, MC_SEG = 16;
__global__ __launch_bounds__(MC_KT * MC_SEG) void k_many_keyscan(const uint32_t* __restrict__ mat, uint32_t* __restrict__ rel,
                                                                uint32_t n_blocks, uint32_t KS, u64 per, int force_typed,
                                                                ManyKeyLb lb, u64* __restrict__ gstart, u64* __restrict__ gkey,
                                                                u64* __restrict__ pstart, u64* __restrict__ off,
                                                                uint32_t* __restrict__ kstart, uint32_t* __restrict__ kgrp,
                                                                ManyTotals* __restrict__ tot, const u64* __restrict__ dict) {
    // (dict != null: the key space is the dense ids of a roaring64 pool's distinct keys; dict[id] = the 48-bit key)
    __shared__ uint32_t s_cnt[MC_SEG][MC_KT], s_w[MC_SEG][MC_KT];
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, n_tiles = (KS + MC_KT - 1u) / MC_KT;
    if (tile >= n_tiles) return;
    const uint32_t seg = threadIdx.x >> 6, lane = lane_id(), k = tile * MC_KT + lane;
    const uint32_t rows = (n_blocks + MC_SEG - 1u) / MC_SEG;
    const uint32_t r0 = seg * rows < n_blocks ? seg * rows : n_blocks, r1 = r0 + rows < n_blocks ? r0 + rows : n_blocks;
    const bool kin = k < KS;
    const uint32_t* __restrict__ col = mat + (kin ? k : 0u);
    uint32_t cnt = 0, wsum = 0;
    {
        uint32_t b = r0;
        for (; b + 8 <= r1; b += 8) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = col[(u64)(b + j) * KS];
#pragma unroll
            for (int j = 0; j < 8; ++j) { cnt += v[j] & MC_CNT_MASK; wsum += v[j] >> MC_CNT_BITS; }
        }
        for (; b < r1; ++b) {
            const uint32_t v = col[(u64)b * KS];
            cnt += v & MC_CNT_MASK; wsum += v >> MC_CNT_BITS;
        }
        if (!kin) { cnt = 0; wsum = 0; }
    }
    s_cnt[seg][lane] = cnt;
    s_w[seg][lane] = wsum;
    __syncthreads();
    uint32_t before = 0;  // members of the key in earlier segments
    for (uint32_t q = 0; q < seg; ++q) before += s_cnt[q][lane];
    if (seg == 0) {
        uint32_t tc = 0, tw = 0;
#pragma unroll
        for (uint32_t q = 0; q < MC_SEG; ++q) { tc += s_cnt[q][lane]; tw += s_w[q][lane]; }
        // ---- chain a: groups | members in front of this key
        const uint32_t isg = tc ? 1u : 0u;
        const uint32_t ig = wave_incl_scan(isg), im = wave_incl_scan(tc);
        const uint32_t tg = wave_lane<63>(ig), tm = wave_lane<63>(im);
        const u64 pa = lb_exclusive_prefix(lb.status_a, tile, ((u64)tg << 32) | tm);
        const u64 g = (pa >> 32) + ig - isg, gs = (pa & 0xFFFFFFFFull) + im - tc, ge = gs + tc;
        const u64 Gt = (pa >> 32) + tg, Mt = (pa & 0xFFFFFFFFull) + tm;  // groups / members up to and including this tile
        // ---- chain b: partial-chunk slots | result-slot units in front of this group
        uint32_t np = 0, sl = 0;
        if (tc) {
            const u64 nu = many_group_pieces(gs, ge, per);
            np = nu > 1 ? (uint32_t)nu : 0u;
            sl = tw > 512u ? 512u : tw;  // (a single member's weight IS its slot; no result -- a pass-through run included -- exceeds a bitset)
        }
        const uint32_t ip = wave_incl_scan(np), is = wave_incl_scan(sl);
        const uint32_t tp = wave_lane<63>(ip), ts = wave_lane<63>(is);
        const u64 pb = lb_exclusive_prefix(lb.status_b, tile, ((u64)tp << 32) | ts);  // (slot units < 2^32: 65 536 groups x 8 192)
        const u64 pp = (pb >> 32) + ip - np, ss = (pb & 0xFFFFFFFFull) + is - sl;
        if (kin) { kstart[k] = (uint32_t)gs; kgrp[k] = (uint32_t)g; }
        if (tc) {
            const u64 kk = dict ? dict[k] : (u64)k;
            gstart[g] = gs; gkey[g] = kk; pstart[g] = pp; off[g] = 16ull * ss;
            if (g + 1 == Gt) atomicMax(&tot->max_key, kk);  // (one per non-empty tile: the last group so far)
        }
        if (tile == n_tiles - 1 && lane == 0) {  // the last tile closes the lists
            const u64 NP = (pb >> 32) + tp, NS = (pb & 0xFFFFFFFFull) + ts;
            gstart[Gt] = Mt; pstart[Gt] = NP; off[Gt] = 16ull * NS;
            tot->n_groups = Gt; tot->n_partials = NP; tot->slot_bytes = 16ull * NS;
        }
    }
    if (kin) {
        uint32_t* __restrict__ out = rel + k;
        uint32_t run = before, b = r0;
        for (; b + 8 <= r1; b += 8) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = col[(u64)(b + j) * KS];
#pragma unroll
            for (int j = 0; j < 8; ++j) { out[(u64)(b + j) * KS] = run; run += v[j] & MC_CNT_MASK; }
        }
        for (; b < r1; ++b) {
            out[(u64)b * KS] = run;
            run += col[(u64)b * KS] & MC_CNT_MASK;
        }
    }
}

// every member's descriptor (and tag) into its group: position = kstart[key] + rel[workgroup][key] (members of the key in
// earlier workgroups) + rank inside the workgroup (an LDS cursor) -- no global atomics.  glast[2g], glast[2g+1] = tag + 1
// of the group's LAST (in gathered order) full-run member (all_runs: run member of any length) and LAST bitset member
// (0 = none), for the replay of roaring_bitmap_or_many's full-union typing / roaring_bitmap_xor_many's fold.  reverse != 0 (tests): a workgroup fills its ranges from the top -- the
// order inside a group must not matter.
__global__ __launch_bounds__(MC_THREADS) void k_many_scatter(PoolView P, ManySel S, uint32_t KS, const uint32_t* __restrict__ rel,
                                                            const uint32_t* __restrict__ kstart, const uint32_t* __restrict__ kgrp,
                                                            u64* __restrict__ sdesc, uint32_t* __restrict__ sord,
                                                            uint32_t* __restrict__ glast, ManyTotals* __restrict__ tot, int reverse,
                                                            int all_runs) {
    __shared__ uint32_t h[MC_W3];
    __shared__ u64 s_bytes[MC_THREADS / 64];
    u64 bytes = 0;
    const uint32_t myrow = many_row(blockIdx.x, S.n_blocks);
    for (uint32_t w0 = 0; w0 < KS; w0 += MC_W3) {
        const uint32_t* __restrict__ row = rel + (u64)myrow * KS + w0;
        const uint32_t* __restrict__ nxt = myrow + 1 < S.n_blocks ? row + KS : nullptr;  // (reverse: the range's end)
        for (uint32_t i = threadIdx.x; i < MC_W3 && w0 + i < KS; i += MC_THREADS) {
            uint32_t base = kstart[w0 + i] + row[i];
            if (reverse)  // end of this workgroup's range of the key = start of the next workgroup's (or of the next key)
                base = nxt ? kstart[w0 + i] + nxt[i] : (w0 + i + 1 < KS ? kstart[w0 + i + 1] : (uint32_t)S.M);
            h[i] = base;
        }
        __syncthreads();
        many_for_members<true>(P, S, myrow, [&](const ManyRec& r, u64 t) {
            const uint32_t k = r.key - w0;
            if (k < MC_W3) {
                const uint32_t pos = reverse ? atomicSub(&h[k], 1u) - 1u : atomicAdd(&h[k], 1u);
#if RHIP_ABL_SC == 1   /* ablation builds only: the kernel without its stores */
                if (pos == 0xFFFFFFFFu) { sdesc[0] = md_pack(r.off, r.ty, r.cd, r.nr); sord[0] = (uint32_t)t; }
#elif RHIP_ABL_SC == 2  /* non-temporal stores */
                __builtin_nontemporal_store(md_pack(r.off, r.ty, r.cd, r.nr), &sdesc[pos]);
                __builtin_nontemporal_store((uint32_t)t, &sord[pos]);
#else
                sdesc[pos] = md_pack(r.off, r.ty, r.cd, r.nr);
                if (sord) sord[pos] = (uint32_t)t;
#endif
                bytes += payload_bytes((uint8_t)r.ty, r.cd, r.nr);
                // (all_runs -- xor_many: ANY run member makes the group replay the reference's fold, many_xor_replay)
                if (r.ty == T_BITSET || (r.ty == T_RUN && (all_runs || r.cd == 65536u)))
                    atomicMax(&glast[2 * (u64)kgrp[r.key] + (r.ty == T_BITSET ? 1u : 0u)], (uint32_t)t + 1u);
            }
        });
        __syncthreads();
    }
    bytes = wave_sum64(bytes);
    if (lane_id() == 0) s_bytes[threadIdx.x >> 6] = bytes;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 b = 0;
        for (uint32_t w = 0; w < MC_THREADS / 64; ++w) b += s_bytes[w];
        if (b) atomicAdd(&tot->bytes_in, b);
    }
}

constexpr uint32_t MANY_TILE = 2048;  // elements per block of the look-back passes: 256 threads x 8, lane-contiguous
struct ManyLb {
    u64* status_a;      // first look-back chain
    u64* status_b;      // second chain (same tile numbering)
    uint32_t* ticket;
};
// exclusive prefixes of the block's 32 (k, wave) partial sums in sa / sb (LDS) and the tile's two global prefixes:
// wave 0 walks chain a, wave 1 chain b.  On return sa / sb hold the exclusive prefixes inside the tile, *pa / *pb
// (LDS) the prefixes in front of the tile; returns the tile totals through ta / tb (LDS).
__device__ __forceinline__ void many_two_scans(uint32_t* sa, uint32_t* sb, const ManyLb& lb, uint32_t tile, u64* pa,
                                               u64* pb, u64* ta, u64* tb) {
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    __syncthreads();
    if (wv < 2) {
        uint32_t* s = wv ? sb : sa;
        const uint32_t v = lane < 32 ? s[lane] : 0u;
        const uint32_t inc = wave_incl_scan(v);
        if (lane < 32) s[lane] = inc - v;
        const u64 tot = __shfl(inc, 63);
        const u64 pfx = lb_exclusive_prefix(wv ? lb.status_b : lb.status_a, tile, tot);
        if (lane == 0) { *(wv ? pb : pa) = pfx; *(wv ? tb : ta) = tot; }
    }
    __syncthreads();
}

// group heads -> group ids; gstart / gkey / cardinality-bound prefix at the heads; glast[2g], glast[2g+1] = member
// index + 1 of the group's LAST full-run member and LAST bitset member (0 = none) for the replay of
// roaring_bitmap_or_many's full-union typing (full_union_is_run below).  A key present in every bitmap of a
// 100 000-bitmap set has 100 000 members (BASELINE config C4's key 0): nothing here is per-group serial.
__global__ __launch_bounds__(256) void k_many_groups(const u64* __restrict__ skey, const u64* __restrict__ sdesc, u64 M,
                                                     ManyLb lb, u64* __restrict__ gstart, u64* __restrict__ gcs,
                                                     u64* __restrict__ gkey, uint32_t* __restrict__ glast,
                                                     ManyTotals* __restrict__ tot, int all_runs) {
    __shared__ uint32_t s_h[32], s_c[32];
    __shared__ uint32_t s_tile;
    __shared__ u64 s_ph, s_pc, s_th, s_tc;
    __shared__ u64 s_bytes[4];
    if (threadIdx.x == 0) s_tile = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    if ((u64)tile * MANY_TILE >= M) return;
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    const u64 tbase = (u64)tile * MANY_TILE + threadIdx.x;
    u64 key[8], d[8];
    bool head[8];
    uint32_t rankh[8], cex[8], cu[8];
    u64 bytes = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 i = tbase + 256ull * k;
        const bool in = i < M;
        key[k] = in ? skey[i] : 0;
        const u64 prev = (in && i) ? skey[i - 1] : 0;
        d[k] = in ? sdesc[i] : 0;
        head[k] = in && (i == 0 || key[k] != prev);
        cu[k] = in ? md_card_ub(d[k]) : 0u;
        bytes += in ? md_payload(d[k]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 bal = __ballot(head[k]);
        rankh[k] = mbcnt(bal);
        const uint32_t inc = wave_incl_scan(cu[k]);
        cex[k] = inc - cu[k];
        const uint32_t wtot = __shfl(inc, 63);
        if (lane == 0) { s_h[4 * k + wv] = (uint32_t)__popcll(bal); s_c[4 * k + wv] = wtot; }
    }
    many_two_scans(s_h, s_c, lb, tile, &s_ph, &s_pc, &s_th, &s_tc);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 i = tbase + 256ull * k;
        const bool in = i < M;
        const u64 hex = s_ph + s_h[4 * k + wv] + rankh[k];  // heads strictly before i
        const u64 g = hex + (head[k] ? 1u : 0u) - 1u;       // (i in => at least one head at or before i)
        const u64 cpre = s_pc + s_c[4 * k + wv] + cex[k];
        if (head[k]) { gstart[g] = i; gkey[g] = key[k]; gcs[g] = cpre; }
        if (in && i == M - 1) {
            gstart[g + 1] = M; gcs[g + 1] = cpre + cu[k];
            tot->n_groups = g + 1; tot->max_key = key[k];
        }
        // last full-run / last bitset member of the group.  At a fixed k the wave's members are consecutive, so when
        // the flagged ones share a group only the highest lane needs to publish.
        const uint32_t ty = md_type(d[k]);
        const bool frf = in && ty == T_RUN && (all_runs || md_full(d[k])), fb = in && ty == T_BITSET;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bool f = x ? fb : frf;
            const u64 m = __ballot(f);
            if (m) {
                const int top = 63 - __clzll((long long)m);
                const u64 gtop = (u64)__shfl((uint32_t)g, top) | ((u64)__shfl((uint32_t)(g >> 32), top) << 32);
                const bool same = __ballot(f && g != gtop) == 0;
                if (same ? (int)lane == top : f) atomicMax(&glast[2 * g + x], (uint32_t)i + 1u);
            }
        }
    }
    bytes = wave_sum64(bytes);
    if (lane == 0) s_bytes[wv] = bytes;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&tot->bytes_in, s_bytes[0] + s_bytes[1] + s_bytes[2] + s_bytes[3]);
}

// per group (48-bit-key path): partial-chunk slots (the pieces of a group that piece boundaries cut) and the result-slot
// size (upper bound on the canonical result payload); look-back sums give the first partial slot of the group and the
// byte offset of its result slot.  Element G (one past the last group) carries the totals.
__global__ __launch_bounds__(256) void k_many_units(const u64* __restrict__ sdesc, const u64* __restrict__ gstart,
                                                    const u64* __restrict__ gcs, u64 per, int force_typed, ManyLb lb,
                                                    u64* __restrict__ pstart, u64* __restrict__ off,
                                                    ManyTotals* __restrict__ tot) {
    __shared__ uint32_t s_s[32], s_p[32];
    __shared__ uint32_t s_tile;
    __shared__ u64 s_pp, s_ps, s_tp, s_ts;
    if (threadIdx.x == 0) s_tile = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const u64 G = tot->n_groups;
    if ((u64)tile * MANY_TILE > G) return;  // (element G is the sentinel)
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    const u64 tbase = (u64)tile * MANY_TILE + threadIdx.x;
    uint32_t np[8], sl[8], sex[8], pex[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 g = tbase + 256ull * k;
        np[k] = 0; sl[k] = 0;
        if (g < G) {
            const u64 gs = gstart[g], ge = gstart[g + 1], cnt = ge - gs;
            const u64 nu = many_group_pieces(gs, ge, per);
            np[k] = nu > 1 ? (uint32_t)nu : 0u;
            uint32_t sz;
            if (cnt == 1 && !force_typed) {
                sz = align16(md_payload(sdesc[gs]));
                const uint32_t cc = align16(2u * md_card_ub(sdesc[gs]));  // (an inefficient run is re-typed: many_pass_through)
                sz = sz > (cc > 8192u ? 8192u : cc) ? sz : (cc > 8192u ? 8192u : cc);
                sz = sz > 8192u ? 8192u : sz;
            } else {
                const u64 c = gcs[g + 1] - gcs[g];
                sz = c >= 4096ull ? 8192u : align16(2u * (uint32_t)c);
            }
            sl[k] = (sz < 16u ? 16u : sz) >> 4;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t is = wave_incl_scan(sl[k]), ip = wave_incl_scan(np[k]);
        sex[k] = is - sl[k]; pex[k] = ip - np[k];
        const uint32_t ts = __shfl(is, 63), tp = __shfl(ip, 63);
        if (lane == 0) { s_s[4 * k + wv] = ts; s_p[4 * k + wv] = tp; }
    }
    many_two_scans(s_p, s_s, lb, tile, &s_pp, &s_ps, &s_tp, &s_ts);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 g = tbase + 256ull * k;
        if (g <= G) {
            const u64 pp = s_pp + s_p[4 * k + wv] + pex[k], ps = 16ull * (s_ps + s_s[4 * k + wv] + sex[k]);
            pstart[g] = pp;
            off[g] = ps;
            if (g == G) { tot->n_partials = pp; tot->slot_bytes = ps; }
        }
    }
}

// group of member position m: largest g with gstart[g] <= m (gstart[G] = M > m)
__device__ __forceinline__ uint32_t many_group_of(const u64* __restrict__ gstart, uint32_t G, u64 m) {
    u64 lo = 0, hi = G;
    while (lo + 1 < hi) {
        u64 mid = (lo + hi) >> 1;
        if (gstart[mid] <= m) lo = mid;
        else hi = mid;
    }
    return (uint32_t)lo;
}

// ------------------------------------------------------------------ accumulation of one unit into an LDS image
constexpr uint32_t MANY_CHUNK = 1024;  // members staged at a time (four per thread)
// One ARRAY member of the staged chunk as the stream of phase A sees it: its slots [start, next) of the chunk's stream and
// its descriptor -- 16 bytes, one ds_read_b128 when a lane enters the member.
struct __attribute__((aligned(16))) ManyArr { uint32_t start, next; u64 d; };
constexpr uint32_t MANY_TMP_WORDS = 4 * MANY_CHUNK + 8;  // ManyArr[MANY_CHUNK]; later one rasterised run member (2048 words) or many_emit's window
struct ManyLists {  // bitset / run members of the chunk (relative member indices), listed by the staging pass
    uint32_t n_bitset, n_run, n_g16, pad;
    uint16_t bitset[MANY_CHUNK], run[MANY_CHUNK];
};
// During the array scatter the image is addressed through an XOR swizzle of the low 5 word-index bits:
// arrays whose values are spaced by a multiple of 1024 (any regular stride, e.g. stratified data)
// would otherwise put every lane of a ds_or on the same LDS bank.  The swizzle is an involution inside each
// aligned 32-word block: one pass converts either way.
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

// Rasterise a run container (toggle bits at every run start and end + 1, then an inclusive prefix-XOR over the 65536
// bits, see lds_load in rhip_block.h) given by its descriptor into dst.
__device__ void many_raster_runs(uint32_t* dst, const uint8_t* __restrict__ arena, u64 d, BlockScratch* sc) {
    const uint32_t tid = threadIdx.x;
    lds_zero(dst);
    __syncthreads();
    const uint32_t n = md_n(d);
    const uint32_t* __restrict__ r = (const uint32_t*)(arena + md_off(d));
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t rl = r[i];
        const uint32_t s = rl & 0xFFFFu, e1 = s + (rl >> 16) + 1u;
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

// Accumulate the members [m0, m1) (at most MANY_CHUNK of them) into the LDS image acc.
// `tmp` (16 KiB) first holds the records of the chunk's array members (ManyArr: slot range of the stream + descriptor),
// later the rasterised run members.
// One 16-byte payload group (eight sorted values, the first nv of them valid) into the swizzled image.  No branch per
// value: an invalid value contributes the mask 0 (x | 0 = x ^ 0 = x), and a wave none of whose groups is ragged (only a
// member's last group can be) skips the validity selects altogether.  The swizzle of both halves
// of a dword is ONE xor: (d >> 5) & 0x03E003E0 is ((v >> 10) & 31) << 5 for either half.  (Round 5: the loop used to
// test the op and the validity with three scalar branches and an exec-mask save / restore per value.)
// An array member in the stream of phase A: its payload seen as whole 128-byte LINES -- `lead` 16-byte slots in front of
// its first group (the payload is 16-byte aligned, not line aligned), its ceil(card / 8) groups, padding to the end of the
// last line.  A step of an octet of lanes is then exactly one line of one member: one memory request, no straddling
// (with groups packed back to back a step touched two lines and a third of the second touches missed the L2: 20.7 M L2
// requests and 2.07 GB from memory for 1.64 GB of payload on C4, `profiles/r05_many_l1_tcc.md`).
__device__ __forceinline__ uint32_t many_lead(u64 d) { return (uint32_t)(d & 7ull); }  // (offset / 16) mod 8
__device__ __forceinline__ uint32_t many_slots(u64 d) { return (many_lead(d) + ((md_n(d) + 7u) >> 3) + 7u) & ~7u; }

template <int OP, bool FULL>
__device__ __forceinline__ void many_scatter8(uint32_t* acc, const uint4& x, uint32_t nv) {
#if RHIP_ABL_L1 == 1 || RHIP_ABL_L1 == 7  /* ablation builds only: the member stream without its LDS atomics */
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u && nv == 77u) atomicOr(&acc[0], 1u);
    return;
#endif
    const uint32_t dd[4] = {x.x, x.y, x.z, x.w};
    const uint32_t keep = FULL ? 0xFFu : (1u << nv) - 1u;  // bit h: value h of the group is there
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t d = dd[q] ^ ((dd[q] >> 5) & 0x03E003E0u);
        const uint32_t w0 = (d >> 5) & 2047u, w1 = d >> 21;
        // (a value that is not there shifts a ZERO into place: one bit-field extract per value where a compare and a select
        // were two)
        const uint32_t b0 = (FULL ? 1u : (keep >> (2 * q)) & 1u) << (d & 31u);
        const uint32_t b1 = (FULL ? 1u : (keep >> (2 * q + 1)) & 1u) << ((d >> 16) & 31u);
#if RHIP_ABL_L1 == 3 || RHIP_ABL_L1 == 4  /* ablation builds only (wrong results): plain stores instead of atomics */
        ((volatile uint32_t*)acc)[w0] = b0; ((volatile uint32_t*)acc)[w1] = b1;
#elif RHIP_ABL_L1 == 5 || RHIP_ABL_L1 == 6  /* ablation: byte stores */
        ((volatile uint8_t*)acc)[w0 * 4 + (d & 3u)] = (uint8_t)b0; ((volatile uint8_t*)acc)[w1 * 4 + ((d >> 16) & 3u)] = (uint8_t)b1;
#else
        if (OP == OP_OR) { atomicOr(&acc[w0], b0); atomicOr(&acc[w1], b1); }
        else { atomicXor(&acc[w0], b0); atomicXor(&acc[w1], b1); }
#endif
    }
}

template <int PF, int OP>
__device__ __forceinline__ void many_accumulate_chunk(uint32_t* acc, uint32_t* tmp, const uint8_t* __restrict__ arena,
                                      const u64* __restrict__ sdesc, u64 m0, u64 m1, BlockScratch* sc,
                                      ManyLists* ml) {
    constexpr int op = OP;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const uint32_t nm = (uint32_t)(m1 - m0);
    ManyArr* rec = (ManyArr*)tmp;                 // [MANY_CHUNK]: the chunk's ARRAY members, compacted, in member order
    __syncthreads();
    if (tid == 0) { ml->n_bitset = 0; ml->n_run = 0; }
    __syncthreads();
    // ---- staging: the array members' records (slot prefix + descriptor), lists of the bitset / run members.
    // A thread takes four CONSECUTIVE members: their loads go out together and ONE block scan -- slots in the low 20
    // bits (a member is at most 66 lines = 528 slots), array members counted above them -- serves the chunk.
    uint32_t carry;
    {
        u64 d[4];
        uint32_t ng[4], mine = 0;
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t j = 4u * tid + r;
            d[r] = j < nm ? sdesc[m0 + j] : 0ull;
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t j = 4u * tid + r;
            ng[r] = 0;
            if (j < nm) {
                const uint32_t ty = md_type(d[r]);
                if (ty == T_ARRAY) ng[r] = many_slots(d[r]) | (1u << 20);
                else if (ty == T_BITSET) ml->bitset[atomicAdd(&ml->n_bitset, 1u)] = (uint16_t)j;
                else ml->run[atomicAdd(&ml->n_run, 1u)] = (uint16_t)j;
            }
            mine += ng[r];
        }
        uint32_t ex = blk_exscan(mine, sc->wsum, &carry);
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            if (ng[r]) {
                const uint32_t st = ex & 0xFFFFFu;
                rec[ex >> 20] = ManyArr{st, st + (ng[r] & 0xFFFFFu), d[r]};
            }
            ex += ng[r];
        }
    }
    const uint32_t T = carry & 0xFFFFFu, na = carry >> 20;  // slots of the stream, array members (block-uniform)
    if (tid == 0) ml->n_g16 = T;
    __syncthreads();
    // ---- phase A: the array members as one flattened stream of 16-byte groups, LDS atomics (commutative: no ordering
    // needed).  The stream is cut into 32 equal ranges of whole lines, one per OCTET of lanes (8 per wave): an octet walks
    // its range one 128-byte line per step, every lane eight values whatever the members' cardinalities.
    // The walk is BRANCH-FREE: every lane loads at every step -- a lane with nothing to fetch (lead / padding slot, past
    // its range) re-reads its member's first group and contributes no value -- and enters the next member with a select
    // plus one 16-byte LDS read.  Until round 5 the fetch was nested conditionals around the load; with loads inside
    // divergent branches the compiler cannot count the loads in flight and waited for ALL of them (s_waitcnt vmcnt(0))
    // before every scatter -- including the one it had just issued: no prefetch depth ever overlapped anything, each
    // step paid a full memory latency, and only occupancy hid it (profiles/r05_many_l1_notes.md: payloads forced into
    // the L2 were no faster, PF 2 / 4 / 8 made no difference).  Straight-line, the PF loads of the ring stay in flight
    // behind the scatter of the oldest.
    if (na) {
        const uint32_t S = (((T + 31u) >> 5) + 7u) & ~7u;  // slots per octet range
        const uint32_t oct = wave * 8u + (lane >> 3), sub = lane & 7u;
        uint32_t qb = oct * S;                               // first slot of the octet's current line
        const uint32_t qe = (oct + 1u) * S < T ? (oct + 1u) * S : T;
        uint32_t m = 0;
        if (qb < qe) {  // largest member with start <= qb
            uint32_t lo = 0, hi = na;
            while (lo + 1u < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (rec[mid].start <= qb) lo = mid;
                else hi = mid;
            }
            m = lo;
        }
        ManyArr R = rec[m];
        constexpr int NB = PF + 1;  // buffers of the ring
        uint4 buf[NB];
        uint32_t bn[NB];
        auto fetch = [&](uint4& x, uint32_t& nv) {  // this lane's slot of the octet's next line
            const bool active = qb < qe;
            const bool adv = active && qb >= R.next;  // (a line belongs to one member and every member has a line: one step)
            m += adv ? 1u : 0u;
            const ManyArr Rn = rec[m];
            R.start = adv ? Rn.start : R.start; R.next = adv ? Rn.next : R.next; R.d = adv ? Rn.d : R.d;
            const uint32_t card = md_n(R.d);
            const int j = (int)(qb + sub) - (int)(R.start + many_lead(R.d));  // group of the member (negative: lead slot)
            const bool valid = active && j >= 0 && 8u * (uint32_t)j < card;
            const uint32_t jj = valid ? (uint32_t)j : 0u;
            nv = valid ? (card - 8u * jj < 8u ? card - 8u * jj : 8u) : 0u;
#if RHIP_ABL_L1 == 2 || RHIP_ABL_L1 == 4 || RHIP_ABL_L1 == 6 || RHIP_ABL_L1 == 7 /* ablation builds only: the LDS atomics without the member loads */
            x = make_uint4(qb * 2654435761u, qb * 40503u + lane, (qb + lane) * 2246822519u, qb ^ (lane * 3266489917u));
#elif RHIP_ABL_L1 == 8  /* ablation: every member inside one MiB (cache-resident payloads) */
            x = ((const uint4*)(arena + (md_off(R.d) & 0xFFC00ull)))[jj];
#else
            x = ((const uint4*)(arena + md_off(R.d)))[jj];
#endif
            qb += 8u;
        };
        const uint32_t nsteps = S >> 3;
#pragma unroll
        for (int p = 0; p < PF; ++p) fetch(buf[p], bn[p]);
        // A ring of PF + 1 buffers, unrolled over one turn: step i issues the load of step i + PF into the buffer step i - 1
        // emptied, then scatters its own -- PF loads in flight behind every scatter, and no buffer is ever copied (a
        // rotation through "current" / "next" variables costs a register move per loaded dword at the loop's back edge,
        // and a move needs its load to have LANDED: the compiler put s_waitcnt vmcnt(0) there).
        for (uint32_t c = 0; c < nsteps; c += NB) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                fetch(buf[(i + PF) % NB], bn[(i + PF) % NB]);
                const uint32_t nv = bn[i];
                // (wave-uniform: only a member's last group can be ragged; lead / padding slots and lanes past the range sit out)
                if (__ballot(nv != 0u && nv != 8u) == 0ull) {
                    if (nv) many_scatter8<OP, true>(acc, buf[i], nv);
                } else if (nv) {
                    many_scatter8<OP, false>(acc, buf[i], nv);
                }
            }
        }
    }
    __syncthreads();
    // ---- phase B: bitset members, thread-owned words (through the swizzle: the image stays swizzled until the whole
    // group is in), two members' loads in flight.  XOR / OR are commutative, so the arbitrary list order is fine.
    {
        const uint32_t nb = ml->n_bitset;
        if (nb) {
            uint32_t r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = acc[mswz(8u * tid + k)];
            uint4 r0 = make_uint4(r[0], r[1], r[2], r[3]), r1 = make_uint4(r[4], r[5], r[6], r[7]);
            for (uint32_t k = 0; k < nb; k += 2) {  // (two members' loads in flight: four put the kernel's register peak here)
                uint4 a[2], b[2];
#pragma unroll
                for (uint32_t u = 0; u < 2; ++u) {
                    a[u] = make_uint4(0, 0, 0, 0); b[u] = a[u];
                    if (k + u < nb) {
                        const uint4* __restrict__ g = (const uint4*)(arena + md_off(sdesc[m0 + ml->bitset[k + u]]));  // (the records hold array members only)
                        a[u] = g[2 * tid];
                        b[u] = g[2 * tid + 1];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 2; ++u) { r0 = op4(op, r0, a[u]); r1 = op4(op, r1, b[u]); }  // (x op 0 = x for or / xor)
            }
            const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[mswz(8u * tid + k)] = w[k];
        }
    }
    __syncthreads();
    // ---- phase C: run members, rasterised one at a time into tmp (the staging tables are dead by now: the descriptor
    // comes from global memory again)
    {
        const uint32_t nr = ml->n_run;
        for (uint32_t k = 0; k < nr; ++k) {
            many_raster_runs(tmp, arena, sdesc[m0 + ml->run[k]], sc);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t x = tmp[8u * tid + q], i = mswz(8u * tid + q);
                acc[i] = op == OP_OR ? (acc[i] | x) : (acc[i] ^ x);
            }
            __syncthreads();
        }
    }
}

// Accumulate the members [m0, m1) into the LDS image acc (ZEROED by the caller), MANY_CHUNK members at a time.
template <int PF, int OP>
__device__ __forceinline__ void many_accumulate(uint32_t* acc, uint32_t* tmp, const uint8_t* __restrict__ arena,
                                const u64* __restrict__ sdesc, u64 m0, u64 m1, BlockScratch* sc, ManyLists* ml) {
    // (the caller's image is all zero, which reads the same through the swizzle; it leaves in linear order)
    for (u64 c0 = m0; c0 < m1; c0 += MANY_CHUNK)
        many_accumulate_chunk<PF, OP>(acc, tmp, arena, sdesc, c0, (c0 + MANY_CHUNK < m1) ? c0 + MANY_CHUNK : m1, sc, ml);
    __syncthreads();
    many_swizzle_pass(acc);  // swizzled -> linear (ends with a barrier)
}

struct ManyOut {
    OutView O;          // candidate directory indexed by group (O.key = group keys)
    u64* partial;       // [n_units][1024] scratch chunks (multi-unit groups)
    u64* chunk_out;     // partial modes: the uncompressed chunks
    int partial_mode;   // 0: canonical containers; 1: chunk of group g at row g; 2: dense table of the sharded form,
                        //    chunk of key k at row (k % world) * dense_b + k / world
    int force_typed;    // 1: single-member groups are typed by cardinality too
    int exact_or_many;  // 1: reproduce roaring_bitmap_or_many's run-vs-bitset choice for FULL containers
    int arrays_only;    // 1: the pool holds array containers only (type census): a full union is a bitset, member tags are not kept
    int exact_xor_many; // 1: reproduce roaring_bitmap_xor_many's container types: a group with a run member replays the fold (many_xor_replay)
    int single_copy;    // 1: ONE bitmap selected -- the reference returns roaring_bitmap_copy (roaring.c:779-781, 799-801): no repair pass
    uint32_t world, dense_b;
    u64 key_space;
    const uint32_t* glast;  // [2 G] tag + 1 of the last full-run / last bitset member of every group (0 = none)
    u64 first_lo, first_hi, second_lo, second_hi;  // container index ranges of ids[0] and ids[1]
                                                   // (their members' tags: 0 + i and (first_hi - first_lo) + i)
    ManyTotals* tot;
};
// row of group g's chunk in a partial mode; ~0 = the key does not fit the dense table (error recorded)
__device__ __forceinline__ u64 many_chunk_row(const ManyOut& MO, uint32_t g) {
    if (MO.partial_mode == 1) return g;
    const u64 k = MO.O.key[g];
    if (k >= MO.key_space) {
        if (threadIdx.x == 0) atomicOr(&MO.tot->err, MANY_ERR_KEYSPACE);
        return ~0ull;
    }
    return (k % MO.world) * (u64)MO.dense_b + k / MO.world;
}

__device__ __forceinline__ bool many_key_in(const PoolView& P, u64 lo, u64 hi, u64 key) {
    const u64 j = lower_bound(P.key, lo, hi, key);
    return j < hi && P.key[j] == key;
}
// A union that fills the whole chunk is the one place where roaring_bitmap_or_many's result type
// depends on the fold order (SURVEY G11 / Appendix A "or_many"): the accumulator is a lazy bitset
// whose cardinality is only computed by bitset x bitset steps (container_lazy_ior, containers.h:
// 1343-1352, which turn a full result into a full RUN), a full run operand replaces it by a full run
// (containers.h:1407-1412), and a known-full accumulator short-circuits later steps (roaring.c:2621).
// Given that the final union IS full, the outcome follows from member metadata plus ONE question:
// "is the union already full after the last bitset member?" -- answered by re-accumulating that prefix (the caller
// does that, through the same accumulation code as everything else: the answer "2" below).
// "First", "last", "after" mean the GATHERED order (the fold order of the reference): a member's tag.  In the sorted
// (48-bit) path a member's tag is its position; in the counting path the positions inside a group are arbitrary and
// the tags are read from V.sord.
// Returns 1 for a full run, 0 for a (full) bitset, 2: full run iff the union of the members with tag <= *replay_tag is
// full.  Block-uniform (metadata only); `red` is LDS scratch of 4 u64.
__device__ int full_union_decide(const PoolView& P, const ManyView& V, const ManyOut& MO, uint32_t g, u64 gs, u64 ge,
                                 u64* red, u64* replay_tag) {
    if (MO.arrays_only) return 0;  // no bitset, no run among the members: the lazy accumulator stays a bitset (and no tags were kept)
    const u64 key = MO.O.key[g];
    // the first two members come from ids[0] and ids[1] iff both of those bitmaps hold the key
    const u64 j0 = lower_bound(P.key, MO.first_lo, MO.first_hi, key), j1 = lower_bound(P.key, MO.second_lo, MO.second_hi, key);
    const bool first = j0 < MO.first_hi && P.key[j0] == key && j1 < MO.second_hi && P.key[j1] == key;
    u64 d0, d1 = 0, tprev;  // tprev: tag of the last member the reference's FIRST step consumes
    if (!V.sord) {
        d0 = V.sdesc[gs]; d1 = V.sdesc[gs + 1];
        tprev = first ? gs + 1 : gs;
    } else if (first) {
        d0 = md_pack(P.off[j0], P.type[j0], P.card[j0], P.nruns[j0]);
        d1 = md_pack(P.off[j1], P.type[j1], P.card[j1], P.nruns[j1]);
        tprev = (MO.first_hi - MO.first_lo) + (j1 - MO.second_lo);
    } else {  // the group's first member in gathered order: smallest tag
        u64 best = ~0ull;
        for (u64 i = gs + threadIdx.x; i < ge; i += blockDim.x) {
            const u64 c = ((u64)V.sord[i] << 32) | (i - gs);
            best = c < best ? c : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 x = __shfl_xor(best, o);
            best = x < best ? x : best;
        }
        __syncthreads();
        if (lane_id() == 0) red[threadIdx.x >> 6] = best;
        __syncthreads();
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) best = red[w] < best ? red[w] : best;
        __syncthreads();
        d0 = V.sdesc[gs + (best & 0xFFFFFFFFull)];
        tprev = best >> 32;
    }
    auto isB = [&](u64 d) { return md_type(d) == T_BITSET; };
    auto isRF = [&](u64 d) { return md_type(d) == T_RUN && md_full(d); };
    if (first) {  // roaring_bitmap_lazy_or(x0, x1), roaring.c:2529-2548
        if (isB(d0) || isB(d1)) { if (isRF(d0) || isRF(d1)) return 1; }
        else if (isRF(d1)) return 1;
    } else {
        if (isRF(d0)) return 1;                       // container_is_full -> every step skipped
        if (isB(d0) && md_full(d0)) return 0;         // known-full bitset: skipped, repair keeps a bitset
    }
    // Any full run among the later members wins (it is never skipped: the accumulator's cardinality is unknown
    // or below 65536 when it arrives); otherwise the LAST bitset member decides.  Both tags (+ 1) were recorded by the
    // grouping kernels.
    const uint32_t lrf = MO.glast[2 * (u64)g], lb = MO.glast[2 * (u64)g + 1];
    if (lrf && (u64)lrf - 1u > tprev) return 1;
    if (!lb || (u64)lb - 1u <= tprev) return 0;
    *replay_tag = (u64)lb - 1u;  // union of the members up to the last bitset member full?
    return 2;
}
// the members of group [gs, ge) with tag <= tag as a descriptor range: in place (sorted path), or compacted into the
// group's range of V.rdesc (any order: the question is only whether their union is full)
__device__ const u64* many_replay_members(const ManyView& V, u64 gs, u64 ge, u64 tag, uint32_t* cnt /* LDS */, u64* m0, u64* m1) {
    if (!V.sord) { *m0 = gs; *m1 = tag + 1; return V.sdesc; }
    __syncthreads();
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();
    for (u64 i = gs + threadIdx.x; i < ge; i += blockDim.x)
        if ((u64)V.sord[i] <= tag) V.rdesc[gs + atomicAdd(cnt, 1u)] = V.sdesc[i];
    __syncthreads();
    *m0 = gs; *m1 = gs + *cnt;
    return V.rdesc;
}
// the result of a group whose union is full: one run {0, 0xFFFF}, or the all-ones bitset
__device__ __forceinline__ void many_emit_full(const ManyOut& MO, uint32_t g, bool as_run) {
    const uint32_t tid = threadIdx.x;
    uint8_t* out = MO.O.arena + MO.O.off[g];
    if (as_run) {
        if (tid == 0) {
            *(uint32_t*)out = 0xFFFF0000u;  // {value 0, length 0xFFFF}
            MO.O.meta[g] = pack_meta(T_RUN, 65536u, 1u);
        }
        return;
    }
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
    ((uint4*)out)[2 * tid] = ones;
    ((uint4*)out)[2 * tid + 1] = ones;
    if (tid == 0) MO.O.meta[g] = pack_meta(T_BITSET, 65536u, 0u);
}

// Emit the image words r[8] of this thread (words [8 tid, 8 tid + 8)) as a bitset or as a sorted array: the many-way
// results are never runs (container_repair_after_lazy), so this is lds_emit without its run extraction.
__device__ __forceinline__ void many_emit(const uint32_t r[8], int ty, uint32_t rc, uint16_t* stage, uint8_t* out,
                                          BlockScratch* sc) {
    const uint32_t tid = threadIdx.x;
    if (ty == T_BITSET) {
        uint4* __restrict__ po = (uint4*)out;
        po[2 * tid] = make_uint4(r[0], r[1], r[2], r[3]);
        po[2 * tid + 1] = make_uint4(r[4], r[5], r[6], r[7]);
        return;
    }
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
    __syncthreads();
    const uint32_t n16 = (2u * rc + 15u) >> 4;
    uint4* __restrict__ po = (uint4*)out;
    for (uint32_t i = tid; i < n16; i += 256) po[i] = ((const uint4*)stage)[i];
}

// What to do with the finished LDS image of group g (members [gs, ge)): partial modes store the chunk; otherwise it is
// canonicalised -- card <= 4096 -> array, else bitset (container_repair_after_lazy, containers.h:344-371); empty ->
// dropped by the tail.  Returns true when the caller must first answer full_union_decide's question by accumulating
// the members with tag <= *replay_tag (many_replay_members) and calling many_emit_full(.., union is full).
__device__ __forceinline__ bool many_finalize(uint32_t* acc, uint16_t* stage, const ManyOut& MO, uint32_t g,
                                              BlockScratch* sc, const PoolView& P, const ManyView& V, u64 gs, u64 ge,
                                              u64* red, u64* replay_tag) {
    const uint32_t tid = threadIdx.x;
    uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
    if (MO.partial_mode) {
        const u64 row = many_chunk_row(MO, g);
        if (row == ~0ull) return false;
        uint4* po = (uint4*)(MO.chunk_out + row * 1024ull);
        po[2 * tid] = r0;
        po[2 * tid + 1] = r1;
        return false;
    }
    uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const uint32_t rc = blk_sum(popc4(r0) + popc4(r1), sc->wsum);
    if (rc == 65536u && MO.exact_or_many && ge - gs >= 2) {
        const int dec = full_union_decide(P, V, MO, g, gs, ge, red, replay_tag);  // (block-uniform: metadata only)
        if (dec == 2) return true;
        many_emit_full(MO, g, dec == 1);
        return false;
    }
    int ty = T_ARRAY;
    if (rc) {
        ty = type_ba(rc);
        many_emit(r, ty, rc, stage, MO.O.arena + MO.O.off[g], sc);
    }
    if (tid == 0) MO.O.meta[g] = pack_meta(ty, rc, 0);
    return false;
}
__device__ __forceinline__ bool many_image_full(const uint32_t* acc, BlockScratch* sc) {
    const uint4 r0 = ((const uint4*)acc)[2 * threadIdx.x], r1 = ((const uint4*)acc)[2 * threadIdx.x + 1];
    return blk_sum(popc4(r0) + popc4(r1), sc->wsum) == 65536u;
}

// A single-member group keeps its container (roaring.c:2660-2676) -- but container_repair_after_lazy still visits it
// (containers.h:344-371): an array or a bitset stays as it is, a RUN container goes through
// convert_run_to_efficient_container (convert.c:154-200) and only stays a run if that is its smallest form.  Every
// container run_optimize produced is; a run list somebody serialized by hand (3 000 runs = 12 KB) is not, and comes out
// as an array or a bitset: such a member is NOT handled here (returns false) and takes the accumulation path, whose
// typing by cardinality is exactly that conversion.  The whole workgroup copies.
__device__ __forceinline__ bool many_pass_through(u64 d, const uint8_t* __restrict__ arena, const ManyOut& MO, uint32_t g,
                                                  BlockScratch* sc) {
    const uint32_t ty = md_type(d), n = md_n(d);
    const uint32_t n16 = (md_payload(d) + 15u) >> 4;
    const uint4* __restrict__ ps = (const uint4*)(arena + md_off(d));
    uint32_t card = ty == T_ARRAY ? n : n + 1u;
    if (ty == T_RUN) {  // the descriptor has no exact cardinality, the runs do (sum of length + 1)
        uint32_t c = 0;
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) {
            const uint4 x = ps[i];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (4u * i + k < n) c += (w[k] >> 16) + 1u;
        }
        card = blk_sum(c, sc->wsum);
        if (!MO.single_copy && type_eff(card, n) != T_RUN) return false;  // (block-uniform)
    }
    uint4* __restrict__ po = (uint4*)(MO.O.arena + MO.O.off[g]);
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) po[i] = ps[i];
    if (threadIdx.x == 0) MO.O.meta[g] = pack_meta(ty, card, ty == T_RUN ? n : 0u);
    return true;
}

// ------------------------------------------------------------------ roaring_bitmap_xor_many: the fold, replayed
// roaring_bitmap_xor_many (roaring.c:795-809) is a FIXED left fold -- lazy_xor(x0, x1), lazy_xor_inplace(.., x2) ...,
// repair_after_lazy -- so the container TYPES of its result are reproducible, and they are not always the canonical
// ones: a RUN accumulator survives R ^ R (run_run_container_xor, mixed_xor.c:179-186: typed by size) and R ^ A with
// fewer than 32 array values (array_run_container_xor, mixed_xor.c:104-138), the first step keeps A ^ R as a raw run
// (containers.h:1636-1651), and whatever is a run at the end goes through convert_run_to_efficient_container
// (containers.h:344-371) -- the result can be a run container.  Without run members every path ends typed by
// cardinality (many_finalize).  A group WITH a run member takes this replay: its members one by one in gathered order
// (their tags), the accumulator's type tracked by the reference's rules from the accumulator's cardinality and run
// count after every step.  State: T = 0 (no accumulator: none yet, or REMOVED when it became empty, roaring.c:2812-2820),
// T_BITSET (lazy or not: the two behave alike), T_ARRAY, T_RUN.
constexpr uint32_t XR_CAP = 1024;  // members sorted at a time (tmp: 2048 words of run raster + XR_CAP u64 sort entries)
static_assert(MANY_TMP_WORDS >= 2048 + 2 * XR_CAP, "the replay's raster buffer and sort window share tmp");

__device__ __forceinline__ void blk_sum2(uint32_t a, uint32_t b, BlockScratch* sc, uint32_t* sa, uint32_t* sb) {
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if (lane_id() == 0) { sc->wsum[threadIdx.x >> 6] = a; sc->wsum2[threadIdx.x >> 6] = b; }
    __syncthreads();
    *sa = sc->wsum[0] + sc->wsum[1] + sc->wsum[2] + sc->wsum[3];
    *sb = sc->wsum2[0] + sc->wsum2[1] + sc->wsum2[2] + sc->wsum2[3];
}
// cardinality and canonical run count of the (linear) LDS image
__device__ __forceinline__ void many_image_stats(const uint32_t* acc, BlockScratch* sc, uint32_t* card, uint32_t* nruns) {
    const uint32_t tid = threadIdx.x;
    const uint4 r0 = ((const uint4*)acc)[2 * tid], r1 = ((const uint4*)acc)[2 * tid + 1];
    const uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    uint32_t pm = tid ? (acc[8 * tid - 1] >> 31) : 0u, c = 0, ns = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c += __popc(r[k]);
        ns += __popc(r[k] & ~((r[k] << 1) | pm));  // run starts: set bits whose predecessor is clear
        pm = r[k] >> 31;
    }
    blk_sum2(c, ns, sc, card, nruns);
}
// one member XORed into the linear image (tmp[0, 2048): raster scratch of a run member)
__device__ __forceinline__ void many_xor_member(uint32_t* acc, uint32_t* tmp, const uint8_t* __restrict__ arena, u64 d,
                                                BlockScratch* sc) {
    const uint32_t tid = threadIdx.x, ty = md_type(d);
    if (ty == T_ARRAY) {
        const uint32_t n = md_n(d);
        const uint16_t* __restrict__ a = (const uint16_t*)(arena + md_off(d));
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t v = a[i];
            atomicXor(&acc[v >> 5], 1u << (v & 31u));
        }
    } else if (ty == T_BITSET) {
        const uint4* __restrict__ g = (const uint4*)(arena + md_off(d));
        const uint4 a = g[2 * tid], b = g[2 * tid + 1];
        uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
        r0 = op4(OP_XOR, r0, a); r1 = op4(OP_XOR, r1, b);
        ((uint4*)acc)[2 * tid] = r0; ((uint4*)acc)[2 * tid + 1] = r1;
    } else {
        many_raster_runs(tmp, arena, d, sc);  // (ends with a barrier)
        uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
        r0 = op4(OP_XOR, r0, ((const uint4*)tmp)[2 * tid]); r1 = op4(OP_XOR, r1, ((const uint4*)tmp)[2 * tid + 1]);
        ((uint4*)acc)[2 * tid] = r0; ((uint4*)acc)[2 * tid + 1] = r1;
    }
    __syncthreads();
}
__device__ __forceinline__ void xr_sort(u64* L, uint32_t n) {  // bitonic, ascending; n <= XR_CAP
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (uint32_t i = n + threadIdx.x; i < n2; i += 256) L[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n2; i += 256) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const u64 a = L[i], b = L[x];
                    if ((a > b) == ((i & k) == 0u)) { L[i] = b; L[x] = a; }
                }
            }
            __syncthreads();
        }
}
// The whole workgroup; acc / tmp are the caller's LDS image and scratch, s_cnt one LDS word.  Writes the group's result
// (payload + meta) like many_finalize does.
__device__ void many_xor_replay(uint32_t* acc, uint32_t* tmp, const PoolView& P, const ManyView& V, const ManyOut& MO,
                                uint32_t g, u64 gs, u64 ge, BlockScratch* sc, uint32_t* s_cnt) {
    const uint32_t tid = threadIdx.x;
    const u64 key = MO.O.key[g];
    // the first step is container_lazy_xor iff ids[0] AND ids[1] hold the key (their members have the two smallest tags)
    const bool first = many_key_in(P, MO.first_lo, MO.first_hi, key) && many_key_in(P, MO.second_lo, MO.second_hi, key);
    u64* L = (u64*)(tmp + 2048);
    __syncthreads();
    lds_zero(acc);
    __syncthreads();
    uint32_t T = 0, card = 0, nruns = 0, step = 0;
    const u64 m = ge - gs, Mtot = V.gstart[V.tot->n_groups];
    u64 done = 0, lo = 0, width = Mtot + 1;  // tags in [lo, lo + width) next
    while (done < m) {
        uint32_t cnt;
        if (!V.sord) {  // sorted path: positions ARE the tags
            cnt = (uint32_t)(m - done < XR_CAP ? m - done : XR_CAP);
            for (uint32_t j = tid; j < cnt; j += 256) L[j] = done + j;
            __syncthreads();
        } else {
            // the members with lo <= tag < lo + width into the window; too many: halve the width (tags are unique, so
            // a width of XR_CAP always fits); none: move on
            for (;;) {
                __syncthreads();
                if (tid == 0) *s_cnt = 0;
                __syncthreads();
                const u64 hi = lo + width;
                for (u64 i = gs + tid; i < ge; i += 256) {
                    const u64 t = V.sord[i];
                    if (t >= lo && t < hi) {
                        const uint32_t p = atomicAdd(s_cnt, 1u);
                        if (p < XR_CAP) L[p] = (t << 32) | (i - gs);
                    }
                }
                __syncthreads();
                cnt = *s_cnt;
                if (cnt > XR_CAP) { width = width / 2 > XR_CAP ? width / 2 : XR_CAP; continue; }
                lo = hi;
                if (cnt) break;
            }
            xr_sort(L, cnt);
        }
        for (uint32_t j = 0; j < cnt; ++j) {
            const u64 d = V.sdesc[gs + (L[j] & 0xFFFFFFFFull)];
            const uint32_t ty = md_type(d), mc = md_n(d);  // (mc: an array member's cardinality)
            const uint32_t pc = card;                      // the accumulator's cardinality before this step
            many_xor_member(acc, tmp, P.arena, d, sc);
            many_image_stats(acc, sc, &card, &nruns);
            ++step;
            const uint32_t by_card = (uint32_t)type_ba(card), by_size = (uint32_t)type_eff(card, nruns);
            if (T == 0u) T = ty;                                   // cloned (roaring.c:2718-2731, 2829-2841)
            else if (T == T_BITSET && ty == T_BITSET) T = T_BITSET;  // bitset_container_xor_nocard
            else if (first && step == 2u) {                        // container_lazy_xor, containers.h:1570-1653
                if (T == T_ARRAY && ty == T_ARRAY) T = pc + mc <= 1024u ? T_ARRAY : T_BITSET;  // mixed_xor.c:221-253
                else if (T == T_RUN && ty == T_RUN) T = by_size;
                else if (T == T_BITSET || ty == T_BITSET) T = T_BITSET;
                else T = T_RUN;                                    // array_run_container_lazy_xor: a raw run
            } else {                                               // container_lazy_ixor -> container_ixor
                if (T == T_RUN && ty == T_RUN) T = by_size;
                else if (T == T_ARRAY && ty == T_RUN) T = pc < 32u ? by_size : by_card;
                else if (T == T_RUN && ty == T_ARRAY) T = mc < 32u ? by_size : by_card;
                else T = by_card;
            }
            if (card == 0u) T = 0u;                                // container_nonzero_cardinality fails: key removed
        }
        done += cnt;
    }
    // repair_after_lazy: a run goes through convert_run_to_efficient_container, a bitset is typed by cardinality
    int ty = T_ARRAY;
    if (card) {
        ty = T == T_RUN ? type_eff(card, nruns) : type_ba(card);
        const uint4 r0 = ((const uint4*)acc)[2 * tid], r1 = ((const uint4*)acc)[2 * tid + 1];
        const uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        __syncthreads();
        if (ty == T_RUN) lds_emit(acc, r, T_RUN, card, nruns, (uint16_t*)tmp, MO.O.arena + MO.O.off[g], sc);
        else many_emit(r, ty, card, (uint16_t*)tmp, MO.O.arena + MO.O.off[g], sc);
    }
    if (tid == 0) MO.O.meta[g] = pack_meta(ty, card, ty == T_RUN ? nruns : 0u);
    __syncthreads();
}
// does group g (members [gs, ge)) take the replay?
__device__ __forceinline__ bool many_xor_replays(const ManyOut& MO, uint32_t g, u64 gs, u64 ge) {
    return MO.exact_xor_many && ge - gs >= 2 && MO.glast[2 * (u64)g] != 0u;
}

// PF = loads of the ring in flight behind every scatter (RHIP_MANY_PF selects 1 / 2 / 3 / 4; default 1)
// RHIP_MANY_WAVES = waves per SIMD the kernel is compiled for = workgroups per CU: five (96 VGPRs; the few spills -- 32
// bytes per lane -- are outside the member stream's loop).  Five only pays when five workgroups really are resident:
// 24.7 KB of LDS each (the array window shares the staging tables) and MANY_RESIDENT pieces per call, one per resident
// workgroup (round 5 measured "five waves: no change" with 32.9 KB of LDS and 1 024 pieces -- neither let a fifth
// workgroup in; with both fixed C4 goes 0.573 -> 0.549 ms on one box, profiles/r05_many_l1_notes.md).
#ifndef RHIP_MANY_WAVES
#define RHIP_MANY_WAVES 5
#endif
constexpr unsigned long long MANY_RESIDENT = 256ull * RHIP_MANY_WAVES;  // gfx950: 256 CUs
template <int PF, int OP>
__global__ __launch_bounds__(256, RHIP_MANY_WAVES) void k_many_l1(PoolView P, ManyView V, ManyOut MO) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint32_t tmp[MANY_TMP_WORDS];
    // The array-extraction window of many_emit lives in tmp: the staging tables are dead once a group's image is complete.
    // (As an array of its own it put the workgroup at 32 936 bytes of LDS -- 168 bytes more than a fifth of the CU's 160 KiB,
    // so a fifth workgroup per CU could never be resident whatever the register count: the "five waves per SIMD, no
    // change" of profiles/r05_many_l1_notes.md was this.)
    static_assert(MANY_TMP_WORDS * 4 >= (4096 + 8) * 2, "many_emit's window fits the staging tables");
    uint16_t* stage = (uint16_t*)tmp;
    __shared__ BlockScratch sc;
    __shared__ ManyLists ml;
    __shared__ u64 red[4];
    __shared__ uint32_t s_cnt;
    const uint32_t G = (uint32_t)V.tot->n_groups;
    if (!G) return;
    const u64 M = V.gstart[G], per = V.per;
    for (u64 piece = blockIdx.x; piece * per < M; piece += gridDim.x) {
        const u64 lo = piece * per, hi = lo + per < M ? lo + per : M;
        for (uint32_t g = many_group_of(V.gstart, G, lo); g < G; ++g) {
            const u64 gs = V.gstart[g], ge = V.gstart[g + 1];
            if (gs >= hi) break;
            const u64 q0 = gs / per, nu = (ge - 1) / per - q0 + 1;  // pieces the group is cut into
            if (nu == 1 && (ge - gs) == 1 && !MO.force_typed && !MO.partial_mode &&
                many_pass_through(V.sdesc[gs], P.arena, MO, g, &sc))  // a single member keeps its container (unless it is an inefficient run)
                continue;
            if (many_xor_replays(MO, g, gs, ge)) {  // xor_many, a run member: the reference's fold decides the type
                if (nu == 1) many_xor_replay(acc, tmp, P, V, MO, g, gs, ge, &sc, &s_cnt);
                continue;                           // (a cut group: k_many_l2 replays it)
            }
            const u64* desc = V.sdesc;
            u64 a0 = gs > lo ? gs : lo, a1 = ge < hi ? ge : hi;
            // One accumulation site for the piece's members and -- rarely -- for the replay of a full union's prefix
            for (bool replay = false;;) {
                __syncthreads();
                lds_zero(acc);
                __syncthreads();
                many_accumulate<PF, OP>(acc, tmp, P.arena, desc, a0, a1, &sc, &ml);  // (a replay only happens under OP_OR)
                if (replay) {
                    many_emit_full(MO, g, many_image_full(acc, &sc));
                    break;
                }
                if (nu != 1) {
                    uint4* po = (uint4*)(MO.partial + (V.pstart[g] + (piece - q0)) * 1024ull);
                    po[2 * threadIdx.x] = ((uint4*)acc)[2 * threadIdx.x];
                    po[2 * threadIdx.x + 1] = ((uint4*)acc)[2 * threadIdx.x + 1];
                    break;
                }
                u64 rtag = 0;
                if (!many_finalize(acc, stage, MO, g, &sc, P, V, gs, ge, red, &rtag)) break;
                replay = true;
                desc = many_replay_members(V, gs, ge, rtag, &s_cnt, &a0, &a1);
            }
        }
    }
}

// combine the partial chunks of the groups that piece boundaries cut
__global__ __launch_bounds__(256) void k_many_l2(PoolView P, ManyView V, ManyOut MO, int op) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint32_t tmp[MANY_TMP_WORDS];
    __shared__ __attribute__((aligned(16))) uint16_t stage[4096 + 8];
    __shared__ BlockScratch sc;
    __shared__ ManyLists ml;
    __shared__ u64 red[4];
    __shared__ uint32_t s_cnt;
    const uint32_t G = (uint32_t)V.tot->n_groups;
    const uint32_t tid = threadIdx.x;
    if (V.tot->n_partials == 0) return;  // no group was cut
    for (uint32_t g = blockIdx.x; g < G; g += gridDim.x) {
        const u64 p0 = V.pstart[g], np = V.pstart[g + 1] - p0;
        if (np < 2) continue;
        if (many_xor_replays(MO, g, V.gstart[g], V.gstart[g + 1])) {
            many_xor_replay(acc, tmp, P, V, MO, g, V.gstart[g], V.gstart[g + 1], &sc, &s_cnt);
            continue;
        }
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
        for (u64 u = 0; u < np; u += 8) {  // eight chunks' loads in flight (one at a time: 15 GB/s for the one block)
            uint4 a[8], b[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                a[k] = make_uint4(0, 0, 0, 0); b[k] = a[k];
                if (u + k < np) {
                    const uint4* __restrict__ p = (const uint4*)(MO.partial + (p0 + u + k) * 1024ull);
                    a[k] = p[2 * tid];
                    b[k] = p[2 * tid + 1];
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) { r0 = op4(op, r0, a[k]); r1 = op4(op, r1, b[k]); }  // (x op 0 = x for or / xor)
        }
        __syncthreads();
        ((uint4*)acc)[2 * tid] = r0;
        ((uint4*)acc)[2 * tid + 1] = r1;
        __syncthreads();
        const u64 gs = V.gstart[g], ge = V.gstart[g + 1];
        u64 rtag = 0;
        if (many_finalize(acc, stage, MO, g, &sc, P, V, gs, ge, red, &rtag)) {
            u64 m0, m1;
            const u64* desc = many_replay_members(V, gs, ge, rtag, &s_cnt, &m0, &m1);
            __syncthreads();
            lds_zero(acc);
            __syncthreads();
            many_accumulate<2, OP_OR>(acc, tmp, P.arena, desc, m0, m1, &sc, &ml);
            many_emit_full(MO, g, many_image_full(acc, &sc));
        }
    }
}

// Stage 3 of the sharded form, dense exchange: the table holds `world` rows per owned key, row s * B + j = source
// rank s's chunk (all zero if s never saw the key) of key rank + world * j.  One workgroup per key combines the rows
// and canonicalises (card <= 4096 -> array, else bitset); slots are fixed (8 KiB per key), empty keys are dropped by
// the tail.  No gather, no sort: the table's shape IS the grouping.
__global__ __launch_bounds__(256) void k_many_dense_finalize(const u64* __restrict__ table, uint32_t world, uint32_t B,
                                                             uint32_t rank, int op, OutView O, u64* __restrict__ zwords,
                                                             u64 n_zwords) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint16_t stage[4096 + 8];
    __shared__ BlockScratch sc;
    const uint32_t tid = threadIdx.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + tid; i < n_zwords; i += (u64)gridDim.x * blockDim.x) zwords[i] = 0;  // the tail's scratch
    for (uint32_t j = blockIdx.x; j < B; j += gridDim.x) {
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
        for (uint32_t s = 0; s < world; s += 4) {
            uint4 a[4], b[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                a[u] = make_uint4(0, 0, 0, 0); b[u] = a[u];
                if (s + u < world) {
                    const uint4* __restrict__ p = (const uint4*)(table + ((u64)(s + u) * B + j) * 1024ull);
                    a[u] = p[2 * tid];
                    b[u] = p[2 * tid + 1];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) { r0 = op4(op, r0, a[u]); r1 = op4(op, r1, b[u]); }
        }
        const uint32_t r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        const uint32_t rc = blk_sum(popc4(r0) + popc4(r1), sc.wsum);
        int ty = T_ARRAY;
        if (rc) {
            ty = type_ba(rc);
            __syncthreads();
            ((uint4*)acc)[2 * tid] = r0;
            ((uint4*)acc)[2 * tid + 1] = r1;
            __syncthreads();
            many_emit(r, ty, rc, stage, O.arena + (u64)j * 8192ull, &sc);
        }
        if (tid == 0) {
            O.meta[j] = pack_meta(ty, rc, 0);
            O.key[j] = (u64)rank + (u64)world * j;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ tail: compaction + directory + totals
// One look-back pass over the n candidate groups: keep = non-empty, prefix -> position in the result directory of the
// ONE result bitmap.  The block of the last tile adds up the per-tile sums and hands the totals (and the completion
// word) to the host in pinned memory.  n comes from device memory (tot->n_groups) unless n_fixed is given.
__global__ __launch_bounds__(256) void k_many_tail(const u64* __restrict__ key, const u64* __restrict__ meta,
                                                   const u64* __restrict__ off, u64 off_stride, DirOut R, u64 n_fixed,
                                                   int use_fixed, LbState lb, u64* __restrict__ part,
                                                   ManyTotals* __restrict__ tot, ManyTotals* __restrict__ host_tot,
                                                   u64* host_flag, u64 seq, u64* host_err, u64 err_tag) {
    __shared__ uint32_t s_cnt[32];
    __shared__ uint32_t s_tile;
    __shared__ u64 s_prefix, s_total;
    __shared__ u64 s_bytes[4];
    __shared__ uint32_t s_types[4][3];
    __shared__ u64 s_tot[4][4];
    const u64 n = use_fixed ? n_fixed : tot->n_groups;
    if (threadIdx.x == 0) s_tile = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const u64 n_tiles = n ? (n + MANY_TILE - 1) / MANY_TILE : 1;
    if (tile >= n_tiles) return;
    const u64 tbase = (u64)tile * MANY_TILE + threadIdx.x;
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    u64 m[8];
    uint32_t rank[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 i = tbase + 256ull * k;
        m[k] = i < n ? meta[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 bal = __ballot(meta_card(m[k]) != 0);
        rank[k] = mbcnt(bal);
        if (lane == 0) s_cnt[4 * k + wv] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t v = lane < 32 ? s_cnt[lane] : 0u;
        const uint32_t inc = wave_incl_scan(v);
        if (lane < 32) s_cnt[lane] = inc - v;
        const u64 t = __shfl(inc, 63);
        const u64 pfx = lb_exclusive_prefix(lb.status, tile, t);
        if (lane == 0) { s_prefix = pfx; s_total = t; }
    }
    __syncthreads();
    u64 bytes = 0;
    uint32_t nty[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 i = tbase + 256ull * k;
        if (i < n && meta_card(m[k])) {
            const u64 ex = s_prefix + s_cnt[4 * k + wv] + rank[k];
            const uint32_t ty = meta_type(m[k]);
            R.key[ex] = key[i];
            R.type[ex] = (uint8_t)ty;
            R.card[ex] = meta_card(m[k]);
            R.nruns[ex] = meta_nruns(m[k]);
            R.off[ex] = off ? off[i] : i * off_stride;
            bytes += payload_bytes((uint8_t)ty, meta_card(m[k]), meta_nruns(m[k]));
            nty[ty - 1]++;
        }
    }
    bytes = wave_sum64(bytes);
#pragma unroll
    for (int t = 0; t < 3; ++t) nty[t] = wave_sum(nty[t]);
    if (lane == 0) {
        s_bytes[wv] = bytes;
#pragma unroll
        for (int t = 0; t < 3; ++t) s_types[wv][t] = nty[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 b = 0;
        uint32_t ty[3] = {0, 0, 0};
        for (int w = 0; w < 4; ++w) {
            b += s_bytes[w];
            for (int t = 0; t < 3; ++t) ty[t] += s_types[w][t];
        }
        lb_store(&part[2 * (size_t)tile], TAIL_READY | b);
        lb_store(&part[2 * (size_t)tile + 1], TAIL_READY | (u64)ty[0] | ((u64)ty[1] << 16) | ((u64)ty[2] << 32));
    }
    if (tile != n_tiles - 1) return;
    __syncthreads();
    const u64 kept = s_prefix + s_total;
    u64 tb = 0, t0 = 0, t1 = 0, t2 = 0;
    for (u64 t = threadIdx.x; t < n_tiles; t += 256) {
        u64 pb, pk;
        while (!((pb = lb_load(&part[2 * t])) & TAIL_READY)) {}
        while (!((pk = lb_load(&part[2 * t + 1])) & TAIL_READY)) {}
        tb += pb & ~TAIL_READY;
        t0 += pk & 0xFFFFu; t1 += (pk >> 16) & 0xFFFFu; t2 += (pk >> 32) & 0xFFFFu;
    }
    tb = wave_sum64(tb); t0 = wave_sum64(t0); t1 = wave_sum64(t1); t2 = wave_sum64(t2);
    if (lane == 0) { s_tot[wv][0] = tb; s_tot[wv][1] = t0; s_tot[wv][2] = t1; s_tot[wv][3] = t2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ManyTotals T = *tot;
        if (use_fixed) T.n_groups = n;
        T.kept = kept;
        T.bytes_out = 0; T.n_type[0] = T.n_type[1] = T.n_type[2] = 0;
        for (int w = 0; w < 4; ++w) {
            T.bytes_out += s_tot[w][0];
            for (int t = 0; t < 3; ++t) T.n_type[t] += s_tot[w][t + 1];
        }
        R.bm_start[0] = 0;
        R.bm_start[1] = kept;
        const u64* lv = (const u64*)&T;
        u64* hv = (u64*)host_tot;
        for (uint32_t k = 0; k < sizeof(ManyTotals) / 8; ++k) hv[k] = lv[k];
        if (T.err) __atomic_store_n(host_err, (err_tag << 8) | T.err, __ATOMIC_RELAXED);  // (see k_many_publish)
        __threadfence_system();
        __atomic_store_n(host_flag, seq, __ATOMIC_RELEASE);
    }
}
// partial modes have no tail: the totals (number of groups, largest key, error word) alone
// The error word carries the TAG of the pipeline that raised it (err_tag << 8 | bits): the call that ends a pipeline
// only believes a word with its own tag, so an abandoned pipeline (stage 1 ran, its finalize never did) cannot fail
// the next one, and nothing ever has to clear the word.  Everything runs on the context's one stream, in order.
__global__ void k_many_publish(const ManyTotals* __restrict__ tot, ManyTotals* __restrict__ host_tot, u64* host_flag, u64 seq,
                               u64* host_err, u64 err_tag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const u64* lv = (const u64*)tot;
        u64* hv = (u64*)host_tot;
        for (uint32_t k = 0; k < sizeof(ManyTotals) / 8; ++k) hv[k] = lv[k];
        if (tot->err) __atomic_store_n(host_err, (err_tag << 8) | tot->err, __ATOMIC_RELAXED);
        __threadfence_system();
        __atomic_store_n(host_flag, seq, __ATOMIC_RELEASE);
    }
}

// directory of rhip_many_finalize's pseudo pool: chunk i is a "bitset container" at byte offset 8192 i
__global__ void k_pseudo_dir(u64* __restrict__ off, uint32_t* __restrict__ card, uint32_t* __restrict__ nruns,
                             uint8_t* __restrict__ type, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { off[i] = 8192ull * i; card[i] = 65536u; nruns[i] = 0u; type[i] = (uint8_t)T_BITSET; }
}
__global__ void k_iota64(u64* p, u64 n, u64 mul) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * mul;
}
