// rhip_common.h -- types shared by every kernel: pool / work-item views, wave64 helpers, the reference's result-typing rules
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef unsigned long long u64;

enum { T_BITSET = 1, T_ARRAY = 2, T_RUN = 3 };
enum { OP_AND = 0, OP_OR = 1, OP_XOR = 2, OP_ANDNOT = 3,
       OP_ITEM = 4 };  // kernel argument of a multi-op batch (rhip_pairwise_multi): every work item carries its own op
// the op of an item: bits 16..17 of FatItem / GenItem `types`, of BBItem `slot` (written by k_emit)
#define ITEM_OP_SHIFT 16
__device__ __forceinline__ int item_op(int kop, uint32_t field) { return kop < OP_ITEM ? kop : (int)((field >> ITEM_OP_SHIFT) & 3u); }
enum { CLS_BB = 0, CLS_GEN = 1, CLS_COPY = 2, CLS_RETRY = 3, CLS_FILT = 4, CLS_WAVE = 5, CLS_RUNS = 6, CLS_PROBE = 7, CLS_BBA = 8, CLS_USMALL = 9, CLS_RUNS16 = 10, CLS_RUNS16W = 11, CLS_BA = 12, N_CLS = 13 };
// interval pairs that run eight / four to a wave (k_ivl<8, .>, k_ivl<16, .>): at most that many intervals per operand and
// values in both operands together; two size classes, because the pairs of a wave advance in lockstep.  (63 / 8 lanes:
// round 3 -- the walk costs the same wave instructions per pair whatever the group width, the per-wave overhead is shared)
#define R16_MAX_IV 63u
#define R16_G 8u  // lanes per pair of that class (eight pairs per wave)
#define R16_MAX_CARD 1024u
#define R16W_MAX_IV 127u
#define R16W_MAX_CARD 4096u
#define USMALL_MAX 255u  // smaller array of a k_usmall item (or / xor of two arrays): at most four values per lane
#define PROBE_MAX 256u  // streamed array of a k_probe item: at most four values per lane
#define RUNS_G 32u  // lanes per pair of the long class
#define RUNS_MAX_INTERVALS 255u  // per operand, for the interval kernel (k_runs); 255 keeps its LDS at 4 x 8 KiB - 64 B
// k_genw's long-list interval path: intervals of both operands together (LDS: 8 bytes per interval and list, as much again
// for the run table: 2 x 8 KiB)
#define RUNSL_MAX_SUM 2032u
#define NONE32 0xFFFFFFFFu

struct PoolView {
    const u64* bm_start;  // [n_bitmaps+1] first container of each bitmap
    const u64* key;       // [n_cont] 16-bit (or 48-bit) container key
    const uint8_t* type;  // [n_cont]
    const uint32_t* card; // [n_cont] cardinality
    const uint32_t* nruns;// [n_cont] run count (runs only)
    const u64* off;       // [n_cont] byte offset of the payload in arena
    const uint8_t* arena;
};

struct OutView {  // candidate (pre-compaction) result directory + the result arena
    u64* key;
    u64* meta;       // card | nruns << 32 | type << 56 : one 8-byte store per result container
    const u64* off;  // exclusive scan of slot[]
    uint8_t* arena;
    uint32_t* slot;  // upper-bound payload bytes of each candidate (multiple of 16)
};
__device__ __forceinline__ u64 pack_meta(uint32_t type, uint32_t card, uint32_t nruns) {
    return (u64)card | ((u64)nruns << 32) | ((u64)type << 56);
}
__device__ __forceinline__ uint32_t meta_card(u64 m) { return (uint32_t)m; }
__device__ __forceinline__ uint32_t meta_nruns(u64 m) { return (uint32_t)(m >> 32) & 0xFFFFFFu; }
__device__ __forceinline__ uint32_t meta_type(u64 m) { return (uint32_t)(m >> 56); }

// Work items carry everything a class kernel needs, resolved at plan time: operand payload offsets, the byte offset
// of the result slot in the result arena (offo), the candidate index whose meta word the kernel writes (out; in
// cardinality mode: the pair index whose accumulator it adds to), cardinalities / types / run counts.
struct __attribute__((aligned(16))) CopyItem {  // pass-through container
    u64 src;       // payload offset in arena A, or in arena B with COPY_FROM_B set
    u64 offo;
    u64 meta;      // pack_meta(type, card, nruns) of the container, stored unchanged
    uint32_t out;
    uint32_t n16;  // payload length in 16-byte units
};
#define COPY_FROM_B (1ull << 63)
struct __attribute__((aligned(16))) FatItem {  // array/bitset pair item
    u64 offa, offb, offo;
    uint32_t out;
    uint32_t ca, cb;     // cardinalities
    uint32_t types;      // ta | tb << 8
    uint32_t pad0, pad1;
};
struct __attribute__((aligned(16))) GenItem {  // general pair item (any type pair, runs included)
    u64 offa, offb;
    uint32_t out, ca, cb, types;   // types = ta | tb << 8
    uint32_t nra, nrb;             // run counts
    u64 offo;
};
struct __attribute__((aligned(16))) BBItem {  // bitset x bitset work item
    u64 offa, offb, offo;
    uint32_t out;
    uint32_t slot;  // bytes of the result slot: 8192 whenever the result can be a bitset at all
};

struct Stats {  // device-side counters of one call, see rhip_stats_t
    u64 matched_pairs, passthrough, bytes_in, bytes_out, n_bb, result_containers;
    u64 n_cand;      // candidates before empty results were dropped
    u64 slot_bytes;  // result arena bytes handed out (slots are upper bounds, 16-byte granular)
    u64 n_type[3];   // result containers by type: bitset, array, run
};


// ------------------------------------------------------------------ phase clocks (diagnostic builds only)
// -DRHIP_PHASES: the wave-per-pair kernels accumulate s_memrealtime ticks (100 MHz) per phase into g_phase[],
// read back with rhip_debug_phases().  Compiled out of the product build (the macros expand to nothing).
#ifdef RHIP_PHASES
__device__ u64 g_phase[32];
#define PH_BEGIN() u64 ph_t_ = wall_clock64(); u64 ph_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PH(n) do { const u64 t__ = wall_clock64(); ph_acc_[n] += t__ - ph_t_; ph_t_ = t__; } while (0)
#define PH_FLUSH(base) do { if (lane_id() == 0) for (int k__ = 0; k__ < 8; ++k__) atomicAdd(&g_phase[(base) + k__], ph_acc_[k__]); } while (0)
#else
#define PH_BEGIN() do { } while (0)
#define PH(n) do { } while (0)
#define PH_FLUSH(base) do { } while (0)
#endif

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t mbcnt(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// Cross-lane moves by DPP (data-parallel primitives: the operand of a VALU instruction is read from another lane of the
// same 16-lane row, or broadcast from the last lane of the previous row) instead of ds_bpermute: no LDS-crossbar round
// trip (~100 cycles, an address computation and a full wait per step of a scan), one instruction per step.  Round 4:
// every step loop of the array kernels ends in a wave prefix sum, and those kernels are bound by instruction issue and
// by exactly such dependent chains.
//   0x110 + n  row_shr:n    lane i of a row reads lane i - n of the same row (invalid below the row: 0 with bound_ctrl)
//   0x142      row_bcast:15 lane 15 of a row to every lane of the next row   (row_mask 0xA: rows 1 and 3 take it)
//   0x143      row_bcast:31 lane 31 to every lane of rows 2 and 3            (row_mask 0xC)
// Lanes whose row is masked out, or whose source is invalid without bound_ctrl, get `old` = 0.
template <int CTRL, int ROW_MASK, bool BOUND_CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, BOUND_CTRL);
}
// value of lane `l` (a constant) for the whole wave: one v_readlane, the result is wave-uniform (a scalar register)
template <int L>
__device__ __forceinline__ uint32_t wave_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, L); }
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v);
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return wave_lane<63>(wave_incl_scan(v)); }
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp0<0x111, 0xF, true>(v);   // within rows of 16: Hillis-Steele, zeros shifted in
    v += dpp0<0x112, 0xF, true>(v);
    v += dpp0<0x114, 0xF, true>(v);
    v += dpp0<0x118, 0xF, true>(v);
    v += dpp0<0x142, 0xA, false>(v);  // rows 1, 3 += total of rows 0, 2
    v += dpp0<0x143, 0xC, false>(v);  // rows 2, 3 += total of rows 0 + 1
    return v;
}

__device__ __forceinline__ uint32_t payload_bytes(uint8_t type, uint32_t card, uint32_t nruns) {
    return type == T_BITSET ? 8192u : (type == T_ARRAY ? 2u * card : 4u * nruns);
}
__device__ __forceinline__ uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

// first index in [lo,hi) with key[idx] >= k
__device__ __forceinline__ u64 lower_bound(const u64* __restrict__ key, u64 lo, u64 hi, u64 k) {
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (key[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// upper bound on the result cardinality of op over a matched pair
__device__ __forceinline__ uint32_t ub_card(int op, uint32_t ca, uint32_t cb) {
    if (op == OP_AND) return ca < cb ? ca : cb;
    if (op == OP_ANDNOT) return ca;
    uint32_t s = ca + cb;
    return s > 65536u ? 65536u : s;
}
// Upper bound on the result payload: whatever type the reference's rules pick, the payload is
// <= min(8192, 2*ub_card) (bitset 8192 needs card > 4096; array = 2*card; a run survives
// convert_run_to_efficient_container only if 2+4*n_runs <= min(8192, 2*card), convert.c:154-170).
__device__ __forceinline__ uint32_t matched_slot(int op, uint32_t ca, uint32_t cb) {
    uint32_t ub = 2u * ub_card(op, ca, cb);
    if (ub > 8192u) ub = 8192u;
    ub = align16(ub);
    return ub < 16u ? 16u : ub;
}


// ------------------------------------------------------------------ result typing
__device__ __forceinline__ int type_eff(uint32_t rc, uint32_t rn) {
    // convert_run_to_efficient_container, convert.c:154-200
    uint32_t size_run = 2u + 4u * rn, size_arr = 2u * rc;
    uint32_t mn = size_arr < 8192u ? size_arr : 8192u;
    if (size_run <= mn) return T_RUN;
    return rc <= 4096u ? T_ARRAY : T_BITSET;
}
__device__ __forceinline__ int type_ba(uint32_t rc) { return rc <= 4096u ? T_ARRAY : T_BITSET; }

// The reference's result-type rules, as a pure function of the operand
// types/cardinalities and the result's cardinality / canonical run count.
__device__ int decide_type(int op, int ta, int tb, uint32_t ca, uint32_t cb, bool fulla, bool fullb, uint32_t rc,
                           uint32_t rn) {
    const bool aA = ta == T_ARRAY, aB = ta == T_BITSET, aR = ta == T_RUN;
    const bool bA = tb == T_ARRAY, bB = tb == T_BITSET, bR = tb == T_RUN;
    switch (op) {
        case OP_AND:  // containers.h:726-806
            if (aA || bA) return T_ARRAY;
            if (aB && bB) return type_ba(rc);
            if (aR && bR) return type_eff(rc, rn);
            {   // bitset x run, mixed_intersection.c:117-202
                const bool full = aR ? fulla : fullb;
                const uint32_t crun = aR ? ca : cb;
                if (full) return T_BITSET;
                if (crun <= 4096u) return T_ARRAY;
                return type_ba(rc);
            }
        case OP_OR:  // containers.h:1008-1103
            if (aB && bB) return T_BITSET;
            if (aA && bA) return (ca + cb <= 4096u) ? T_ARRAY : type_ba(rc);  // mixed_union.c:162-191
            if (aR && bR) return type_eff(rc, rn);
            if ((aB && bA) || (aA && bB)) return T_BITSET;
            if (aB || bB) return (aR ? fulla : fullb) ? T_RUN : T_BITSET;
            return type_eff(rc, rn);  // array x run, mixed_union.c:66-108
        case OP_XOR:  // containers.h:1449-1524
            if (aA && bA) return (ca + cb <= 4096u) ? T_ARRAY : type_ba(rc);  // mixed_xor.c:196-219
            if (aR && bR) return type_eff(rc, rn);
            if (aB || bB) return type_ba(rc);
            {   // array x run, mixed_xor.c:104-138
                const uint32_t carr = aA ? ca : cb, crun = aA ? cb : ca;
                if (carr < 32u) return type_eff(rc, rn);
                if (crun <= 4096u) return (carr + crun <= 4096u) ? T_ARRAY : type_ba(rc);
                return type_ba(rc);
            }
        default:  // OP_ANDNOT, containers.h:1783-1876
            if (aA) return T_ARRAY;
            if (aB) return type_ba(rc);
            // a is a run
            if (bR) return type_eff(rc, rn);                       // mixed_andnot.c:430-438
            if (bB) return ca <= 4096u ? T_ARRAY : type_ba(rc);    // mixed_andnot.c:104-150
            if (ca <= 32u) return type_eff(rc, rn);                // mixed_andnot.c:277-361
            return ca <= 4096u ? T_ARRAY : type_ba(rc);
    }
}

// ------------------------------------------------------------------ result stores
// A result payload is written once and not read again by the batch that writes it, while the OPERANDS of a realdata
// batch are re-read by every pair they appear in (weather_sept_85: ~85 times, a 7.6 MB working set against 4 MB of L2
// per XCD): results leave with non-temporal stores so that a gigabyte of them does not push the operands out.
// (-DRHIP_NT_OUT=0: plain stores, for A/B measurements.)
#ifndef RHIP_NT_OUT
#define RHIP_NT_OUT 1
#endif
typedef unsigned int rhip_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void out_store16(uint4* __restrict__ p, const uint4 v) {
#if RHIP_NT_OUT
    rhip_v4u t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, (rhip_v4u*)p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void out_store16(rhip_v4u* __restrict__ p, const rhip_v4u v) {
#if RHIP_NT_OUT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ------------------------------------------------------------------ which operand is the MEMBERSHIP / image side
// The image kernels treat their operands asymmetrically: X is rasterised (8 KiB image: LDS for the filter, registers for
// the union kernels), Y is streamed against it.  One definition, used by the planning kernels (X-grouped queues, below)
// and by every kernel that builds an image, so that they always agree.
//   filter (and / cardinality): Y = the array, the SMALLER one of two arrays; andnot: Y = a (the array), X = b
//   union  (or / xor): X = the bitset, else the larger array; bitset \ array: X = a
__device__ __forceinline__ bool filt_y_is_a(int op, uint32_t ta, uint32_t tb, uint32_t ca, uint32_t cb) {
    if (op != OP_AND) return true;
    return (ta == T_ARRAY) && (tb != T_ARRAY || ca <= cb);
}
__device__ __forceinline__ bool union_x_is_a(int op, uint32_t ta, uint32_t tb, uint32_t ca, uint32_t cb) {
    if (op == OP_ANDNOT) return true;
    return (ta == T_BITSET) || (tb != T_BITSET && ca >= cb);
}
// X-grouped queues (round 4).  In a batch where a container meets MANY partners (all pairs of a pool: ~85 per container
// on weather_sept_85) the image kernels used to rasterise the same X once per item -- 55-62 % of their time.  When the
// host expects that much reuse, the items of the filter class and of the union classes (two-array image pairs, bitset
// (op) array) are not queued in pair order but COUNTING-SORTED by their X container: k_count adds every such item to a
// histogram over (group, X container), the batch's one prefix scan runs over the histogram as well, k_emit claims a
// slot inside the item's bucket with an atomic decrement (which also returns the histogram to zero for the next
// batch), and the grouped kernels (rhip_grouped.h) walk runs of items with the same X, building its image once.
// Results do not depend on the order inside a bucket: every item carries its own result slot.
struct XGroupView {
    uint32_t* hist;       // [2 * nx + 1] item counts per (group, X container); zero outside k_count .. k_emit
    const u64* hstart;    // exclusive scan of hist (same indexing)
    uint32_t nx;          // X containers per group: n_cont(A) + n_cont(B), or n_cont(A) when both operands are ONE pool
    uint32_t na;          // n_cont(A): index of B's first container inside a group (0 when both are one pool)
    uint32_t on;
};
enum { XG_FILT = 0, XG_UNION = 1 };
__device__ __forceinline__ uint32_t xg_index(const XGroupView& X, int grp, bool x_in_a, u64 ia, u64 ib) {
    return (uint32_t)grp * X.nx + (x_in_a ? (uint32_t)ia : X.na + (uint32_t)ib);
}
// items of a grouped queue one wave walks in a row (the reuse it can see), given the queue length and the launch
__device__ __forceinline__ uint32_t xg_chunk(uint32_t n, uint32_t nwaves, uint32_t cmin) {
    const uint32_t c = (n + nwaves - 1) / nwaves;
    return c < cmin ? cmin : c;
}

// A value every lane of the wave holds alike, TOLD to the compiler (v_readfirstlane): the index of the wave in its launch,
// above all.  The item a wave works on is then fetched by scalar loads, lives in scalar registers, and every branch on its
// op / types / sizes is a scalar branch -- without it the compiler treated them as divergent (exec-mask bookkeeping
// around each one: k_wave was 982 vector + 1229 scalar instructions).
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ------------------------------------------------------------------ sub-wave groups
// G consecutive lanes (G = 16, 32 or 64) that work on one item: the group's view of the wave collectives.  The wave
// must reach each of them in uniform control flow; only what a lane reads is restricted to its own group.
template <uint32_t G>
struct Grp {
    uint32_t lane, grp, gl, glast;
    __device__ __forceinline__ Grp() {
        lane = lane_id(); grp = lane / G; gl = lane % G; glast = (lane & ~(G - 1u)) | (G - 1u);
    }
    __device__ __forceinline__ u64 ballot(bool p) const {  // the group's slice of the wave ballot
        const u64 b = __ballot(p);
        if (G == 64) return b;
        return (b >> (G * grp)) & ((1ull << (G & 63u)) - 1ull);
    }
    __device__ __forceinline__ uint32_t rank(u64 m) const {  // group ballot bits below this lane
        return (uint32_t)__popcll(m & ((1ull << gl) - 1ull));
    }
    __device__ __forceinline__ uint32_t incl_scan(uint32_t v) const {
        if (G == 64) return wave_incl_scan(v);
        // (DPP rows are 16 lanes: a 16-lane group is a row, a 32-lane group two rows + one broadcast; an 8-lane group
        // shares its row with another group, so its steps are masked by the lane's position in the group)
        { const uint32_t t = dpp0<0x111, 0xF, true>(v); v += (G >= 16 || gl >= 1u) ? t : 0u; }
        { const uint32_t t = dpp0<0x112, 0xF, true>(v); v += (G >= 16 || gl >= 2u) ? t : 0u; }
        { const uint32_t t = dpp0<0x114, 0xF, true>(v); v += (G >= 16 || gl >= 4u) ? t : 0u; }
        if (G >= 16) v += dpp0<0x118, 0xF, true>(v);
        if (G >= 32) v += dpp0<0x142, 0xA, false>(v);
        return v;
    }
    __device__ __forceinline__ uint32_t sum(uint32_t v) const {
        if (G == 64) return wave_sum(v);
#pragma unroll
        for (uint32_t o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    }
    __device__ __forceinline__ uint32_t wave_max(uint32_t v) const {  // v is group-uniform: max over the groups
#pragma unroll
        for (uint32_t o = G; o < 64; o <<= 1) {
            const uint32_t t = __shfl_xor(v, o);
            v = t > v ? t : v;
        }
        return v;
    }
};

