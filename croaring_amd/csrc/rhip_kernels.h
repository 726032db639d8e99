// rhip_kernels.h -- CDNA4 (gfx950, wave64) kernels of the Roaring set-operation engine.
//
// Data layout in HBM: a pool is a structure-of-arrays container DIRECTORY
// (key, type, card, nruns, byte offset; containers sorted by (bitmap, key)) plus one payload
// ARENA.  Every payload starts 16-byte aligned and is padded to a multiple of 16 bytes, so
// every kernel moves payload as 16 B/lane (1 KiB per wave instruction, fully coalesced).
// A bitset container is 1024 contiguous u64 words = 8 wave-wide 16-byte loads.
//
// Kernel inventory (one per SURVEY §2.2 row it replaces):
//   k_count / k_emit   two-pointer key merge of roaring.c:742-768 as lane-per-key binary
//                      search + wave ballot/mbcnt ranking, one wave per 256-entry directory
//                      tile -> work items in 6 class queues at deterministic positions
//   k_bb               K1-K3: bitset (x) bitset {and,or,xor,andnot} fused with popcount,
//                      one wave per container pair, 16 x 16-byte loads in flight per lane
//   k_copy             pass-through containers (roaring.c:914-941 clone paths)
//   k_filter           K8/K9/K12: array filtered by membership (and / andnot), wave per pair
//   k_wave             K6/K10/K11: or / xor / bitset \ array with an array operand, wave-private
//                      LDS image + returning LDS atomics, wave per pair
//   k_runs             K13/K14/K16: run x run, run x short array as interval algebra in
//                      O(n log n): boundary parity membership, ballot-ranked result runs
//   k_genw             K5/K7/K15 + the rest: run x bitset, long run/array pairs, bitset x bitset
//                      results that become arrays: wave-private LDS image (runs: toggle bits +
//                      prefix-xor), op + popcount + run counting, result re-typed by the
//                      reference's rules (Appendix A) and extracted with prefix sums
//   k_many_*           group-by-key OR/XOR accumulation for or_many / xor_many
//   k_compact          drops empty results, builds the result directory
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

enum { T_BITSET = 1, T_ARRAY = 2, T_RUN = 3 };
enum { OP_AND = 0, OP_OR = 1, OP_XOR = 2, OP_ANDNOT = 3 };
enum { CLS_BB = 0, CLS_GEN = 1, CLS_COPY = 2, CLS_RETRY = 3, CLS_FILT = 4, CLS_WAVE = 5, CLS_RUNS = 6, N_CLS = 7 };
#define RUNS_MAX_INTERVALS 256u  // per operand, for the interval kernel (k_runs)
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

struct Item {
    uint32_t a;    // container index in pool A (NONE32: pass-through from B)
    uint32_t b;    // container index in pool B (NONE32: pass-through from A)
    uint32_t out;  // candidate index (cardinality mode: pair index)
};
struct __attribute__((aligned(16))) FatItem {  // array/bitset pair item: everything the kernel needs, resolved at plan time
    u64 offa, offb;      // payload offsets in arena A / arena B
    uint32_t out;        // candidate index (cardinality mode: pair index)
    uint32_t ca, cb;     // cardinalities
    uint32_t types;      // ta | tb << 8
};
struct __attribute__((aligned(16))) GenItem {  // general pair item (any type pair, runs included)
    u64 offa, offb;
    uint32_t out, ca, cb, types;   // types = ta | tb << 8
    uint32_t nra, nrb, pad0, pad1; // run counts
};
struct __attribute__((aligned(16))) BBItem {  // bitset x bitset work item: payload offsets resolved at plan time
    u64 offa, offb;
    uint32_t a, b, out, pad;
};

struct Stats {  // device-side counters, see rhip_stats_t
    u64 matched_pairs, passthrough, bytes_in, bytes_out, n_bb, result_containers;
};

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t mbcnt(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane_id() >= (uint32_t)o) v += t;
    }
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

// ------------------------------------------------------------------ planning
// Four lower_bound searches per lane issued together (independent dependent-load chains).
__device__ __forceinline__ void lower_bound4(const u64* __restrict__ key, u64 lo0, u64 hi0, const u64 k[4],
                                             const bool act[4], u64 out[4]) {
    u64 lo[4], hi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { lo[t] = lo0; hi[t] = act[t] ? hi0 : lo0; }
    bool more = true;
    while (more) {
        more = false;
        u64 mid[4], kv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { mid[t] = (lo[t] + hi[t]) >> 1; kv[t] = (lo[t] < hi[t]) ? key[mid[t]] : 0; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (lo[t] < hi[t]) {
                if (kv[t] < k[t]) lo[t] = mid[t] + 1;
                else hi[t] = mid[t];
                more |= lo[t] < hi[t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = lo[t];
}

// Planning works on UNITS: one unit = one tile of up to 256 consecutive directory entries of the
// left bitmap of a pair ("A-tile"), or -- for OR/XOR, whose result also carries the right bitmap's
// unmatched containers -- of the right bitmap ("B-tile").  One wave per unit, so a batch of 250 pairs
// of 4096-container bitmaps plans on 4000 waves instead of 250.
// Count arrays (and their exclusive scan) have 5 sections of n_units+1 entries:
enum { SEC_CAND = 0, SEC_M = 1, SEC_BB = 2, SEC_GEN = 3, SEC_COPY = 4, SEC_FILT = 5, SEC_WAVE = 6, SEC_RUNS = 7, N_SEC = 8 };
// work class of a matched container pair
// ia / ib = number of intervals of the operand when it is read as an interval list (runs: n_runs, arrays: card)
__device__ __forceinline__ int classify(int op, int cardmode, uint8_t ta, uint8_t tb, uint32_t ia, uint32_t ib) {
    if (ta == T_BITSET && tb == T_BITSET) return CLS_BB;
    // interval algebra in O(n log n) when a run container meets a run / a short array
    if ((ta == T_RUN || tb == T_RUN) && ta != T_BITSET && tb != T_BITSET && ia <= RUNS_MAX_INTERVALS &&
        ib <= RUNS_MAX_INTERVALS)
        return CLS_RUNS;
    // array filtered by membership in an array / bitset: and (either order), array \ x
    if (cardmode || op == OP_AND) {
        if ((ta == T_ARRAY && tb != T_RUN) || (tb == T_ARRAY && ta != T_RUN)) return CLS_FILT;
    } else if (op == OP_ANDNOT) {
        if (ta == T_ARRAY && tb != T_RUN) return CLS_FILT;
        if (ta == T_BITSET && tb == T_ARRAY) return CLS_WAVE;  // bitset \ array: clear-list in LDS
    } else {
        if (ta != T_RUN && tb != T_RUN) return CLS_WAVE;       // or / xor with an array operand
    }
    return CLS_GEN;
}
#define UNIT_B 0x80000000u

struct UnitView {
    const uint32_t* pair;   // [U] pair index of the unit
    const uint32_t* tile;   // [U] tile index inside its side; UNIT_B flag marks a B-tile
    const u64* pair_unit0;  // [npairs+1] first unit of each pair
    uint32_t n_units;
};

// One wave per unit: contributions of the tile to every section.
__global__ __launch_bounds__(256) void k_count(PoolView A, PoolView B, const uint32_t* __restrict__ lhs,
                                               const uint32_t* __restrict__ rhs, UnitView U, int op, int cardmode,
                                               uint32_t* __restrict__ counts) {
    const uint32_t u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= U.n_units) return;
    const uint32_t lane = lane_id();
    const uint32_t p = U.pair[u];
    const bool bside = (U.tile[u] & UNIT_B) != 0;
    const u64 tile = U.tile[u] & ~UNIT_B;
    const u64 a0 = A.bm_start[lhs[p]], a1 = A.bm_start[lhs[p] + 1];
    const u64 b0 = B.bm_start[rhs[p]], b1 = B.bm_start[rhs[p] + 1];
    // s* = the side this tile walks, l* = the side it searches
    const PoolView& SV = bside ? B : A;
    const PoolView& LV = bside ? A : B;
    const u64 s0 = (bside ? b0 : a0) + tile * 256, sEnd = bside ? b1 : a1;
    const u64 s1 = s0 + 256 < sEnd ? s0 + 256 : sEnd;
    const u64 l0 = bside ? a0 : b0, l1 = bside ? a1 : b1;
    u64 k[4], j[4];
    bool act[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        act[t] = s0 + 64 * t + lane < s1;
        k[t] = act[t] ? SV.key[s0 + 64 * t + lane] : 0;
    }
    lower_bound4(LV.key, l0, l1, k, act, j);
    uint32_t matched = 0, nbb = 0, nfilt = 0, nwave = 0, nruns_cls = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool found = act[t] && j[t] < l1 && LV.key[j[t]] == k[t];
        int cls = -1;
        if (found && !bside) {
            const u64 ai = s0 + 64 * t + lane;
            const uint8_t ta = SV.type[ai], tb = LV.type[j[t]];
            cls = classify(op, cardmode, ta, tb, ta == T_RUN ? SV.nruns[ai] : SV.card[ai],
                           tb == T_RUN ? LV.nruns[j[t]] : LV.card[j[t]]);
        }
        matched += (uint32_t)__popcll(__ballot(found));
        nbb += (uint32_t)__popcll(__ballot(cls == CLS_BB));
        nfilt += (uint32_t)__popcll(__ballot(cls == CLS_FILT));
        nwave += (uint32_t)__popcll(__ballot(cls == CLS_WAVE));
        nruns_cls += (uint32_t)__popcll(__ballot(cls == CLS_RUNS));
    }
    if (lane == 0) {
        const uint32_t n = (uint32_t)(s1 - s0);
        const size_t S = (size_t)U.n_units + 1;
        uint32_t ncopy;
        if (bside) ncopy = n - matched;                              // OR/XOR only
        else ncopy = (cardmode || op == OP_AND) ? 0u : n - matched;  // A-only containers pass through
        counts[SEC_CAND * S + u] = bside ? ncopy : matched + ncopy;
        counts[SEC_M * S + u] = matched;
        counts[SEC_BB * S + u] = nbb;
        counts[SEC_GEN * S + u] = bside ? 0u : matched - nbb - nfilt - nwave - nruns_cls;
        counts[SEC_RUNS * S + u] = nruns_cls;
        counts[SEC_FILT * S + u] = nfilt;
        counts[SEC_WAVE * S + u] = nwave;
        counts[SEC_COPY * S + u] = ncopy;
    }
}

// One wave per unit: emit candidates in merged key order (roaring.c:742-768, 895-951) and the work
// items of each class at deterministic queue positions (no atomics).  The position of a candidate
// inside its result bitmap is computed by ranking, not by a serial merge:
//   matched / A-only element i (key k):  i + |{B keys < k}| - |{matched keys < k}|
//   B-only element j (key k)          :  j + |{A keys < k}| - |{matched keys < k}|
// with |{matched keys < k}| = (matched count of the pair's earlier tiles, from the scan) + a ballot rank.
struct EmitQueues {
    BBItem* bb;   // section SEC_BB
    GenItem* gen; // section SEC_GEN
    Item* copy;   // section SEC_COPY
    FatItem* filt;  // section SEC_FILT
    FatItem* wave;  // section SEC_WAVE
    GenItem* runs;  // section SEC_RUNS
};
__global__ __launch_bounds__(256) void k_emit(PoolView A, PoolView B, const uint32_t* __restrict__ lhs,
                                              const uint32_t* __restrict__ rhs, UnitView U, int op, int cardmode,
                                              const u64* __restrict__ starts, OutView O, EmitQueues Q,
                                              u64* __restrict__ unit_bytes) {
    const uint32_t u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= U.n_units) return;
    const uint32_t lane = lane_id();
    const size_t S = (size_t)U.n_units + 1;
    const uint32_t p = U.pair[u];
    const bool bside = (U.tile[u] & UNIT_B) != 0;
    const u64 tile = U.tile[u] & ~UNIT_B;
    const u64 a0 = A.bm_start[lhs[p]], a1 = A.bm_start[lhs[p] + 1];
    const u64 b0 = B.bm_start[rhs[p]], b1 = B.bm_start[rhs[p] + 1];
    const u64 u0 = U.pair_unit0[p];
    const u64 base = starts[SEC_CAND * S + u0];
    u64 qbb = starts[SEC_BB * S + u] - starts[SEC_BB * S];
    u64 qgen = starts[SEC_GEN * S + u] - starts[SEC_GEN * S];
    u64 qcopy = starts[SEC_COPY * S + u] - starts[SEC_COPY * S];
    u64 qfilt = starts[SEC_FILT * S + u] - starts[SEC_FILT * S];
    u64 qwave = starts[SEC_WAVE * S + u] - starts[SEC_WAVE * S];
    u64 qruns = starts[SEC_RUNS * S + u] - starts[SEC_RUNS * S];
    u64 bytes_in = 0;
    u64 k[4], j[4];
    bool act[4];
    if (!bside) {
        const u64 s0 = a0 + tile * 256;
        uint32_t mbefore = (uint32_t)(starts[SEC_M * S + u] - starts[SEC_M * S + u0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            act[t] = s0 + 64 * t + lane < a1 && 64 * t + lane < 256;
            k[t] = act[t] ? A.key[s0 + 64 * t + lane] : 0;
        }
        lower_bound4(B.key, b0, b1, k, act, j);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u64 ai = s0 + 64 * t + lane;
            const bool found = act[t] && j[t] < b1 && B.key[j[t]] == k[t];
            const u64 fm = __ballot(found);
            const uint32_t mb = mbefore + mbcnt(fm);
            mbefore += (uint32_t)__popcll(fm);
            const bool emit = act[t] && (found || (!cardmode && op != OP_AND));
            uint8_t ta = 0, tb = 0;
            uint32_t ca = 0, cb = 0, pa = 0, pos = 0, nra = 0, nrb = 0;
            if (emit) {
                const uint32_t ilocal = (uint32_t)(ai - a0), lbcount = (uint32_t)(j[t] - b0);
                if (op == OP_AND || cardmode) pos = mb;
                else if (op == OP_ANDNOT) pos = ilocal;
                else pos = ilocal + lbcount - mb;
                ta = A.type[ai];
                ca = A.card[ai];
                nra = A.nruns[ai];
                pa = payload_bytes(ta, ca, nra);
                bytes_in += pa;
                if (found) {
                    tb = B.type[j[t]];
                    cb = B.card[j[t]];
                    nrb = B.nruns[j[t]];
                    bytes_in += payload_bytes(tb, cb, nrb);
                }
                if (!cardmode) {
                    O.key[base + pos] = k[t];
                    uint32_t sl = found ? matched_slot(op, ca, cb) : align16(pa);
                    O.slot[base + pos] = sl < 16u ? 16u : sl;
                }
            }
            const uint32_t outidx = cardmode ? p : (uint32_t)(base + pos);
            const int cls = (emit && found) ? classify(op, cardmode, ta, tb, ta == T_RUN ? nra : ca, tb == T_RUN ? nrb : cb) : -1;
            const bool isbb = cls == CLS_BB;
            const bool isgen = cls == CLS_GEN;
            const bool isfilt = cls == CLS_FILT;
            const bool iswave = cls == CLS_WAVE;
            const bool isruns = cls == CLS_RUNS;
            const bool iscopy = emit && !found;
            const u64 mbb = __ballot(isbb), mgen = __ballot(isgen), mcp = __ballot(iscopy), mfl = __ballot(isfilt);
            const u64 mwv = __ballot(iswave), mrn = __ballot(isruns);
            if (isbb) {
                BBItem it;
                it.offa = A.off[ai]; it.offb = B.off[j[t]];
                it.a = (uint32_t)ai; it.b = (uint32_t)j[t]; it.out = outidx; it.pad = 0;
                Q.bb[qbb + mbcnt(mbb)] = it;
            }
            if (isgen || isruns) {
                GenItem it;
                it.offa = A.off[ai]; it.offb = B.off[j[t]];
                it.out = outidx; it.ca = ca; it.cb = cb; it.types = (uint32_t)ta | ((uint32_t)tb << 8);
                it.nra = nra; it.nrb = nrb; it.pad0 = 0; it.pad1 = 0;
                if (isgen) Q.gen[qgen + mbcnt(mgen)] = it;
                else Q.runs[qruns + mbcnt(mrn)] = it;
            }
            if (isfilt || iswave) {
                FatItem it;
                it.offa = A.off[ai]; it.offb = B.off[j[t]];
                it.out = outidx; it.ca = ca; it.cb = cb; it.types = (uint32_t)ta | ((uint32_t)tb << 8);
                if (isfilt) Q.filt[qfilt + mbcnt(mfl)] = it;
                else Q.wave[qwave + mbcnt(mwv)] = it;
            }
            if (iscopy) Q.copy[qcopy + mbcnt(mcp)] = Item{(uint32_t)ai, NONE32, outidx};
            qbb += __popcll(mbb); qgen += __popcll(mgen); qcopy += __popcll(mcp); qfilt += __popcll(mfl); qwave += __popcll(mwv); qruns += __popcll(mrn);
        }
    } else {
        const u64 nAt = (a1 - a0 + 255) / 256;
        const u64 s0 = b0 + tile * 256;
        uint32_t mbefore = (uint32_t)(starts[SEC_M * S + u] - starts[SEC_M * S + u0 + nAt]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            act[t] = s0 + 64 * t + lane < b1;
            k[t] = act[t] ? B.key[s0 + 64 * t + lane] : 0;
        }
        lower_bound4(A.key, a0, a1, k, act, j);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u64 bi = s0 + 64 * t + lane;
            const bool found = act[t] && j[t] < a1 && A.key[j[t]] == k[t];
            const u64 fm = __ballot(found);
            const uint32_t mb = mbefore + mbcnt(fm);
            mbefore += (uint32_t)__popcll(fm);
            const bool emit = act[t] && !found;
            const u64 mcp = __ballot(emit);
            if (emit) {
                const uint32_t pos = (uint32_t)(bi - b0) + (uint32_t)(j[t] - a0) - mb;
                const uint32_t pb = payload_bytes(B.type[bi], B.card[bi], B.nruns[bi]);
                O.key[base + pos] = k[t];
                O.slot[base + pos] = align16(pb) < 16u ? 16u : align16(pb);
                bytes_in += pb;
                Q.copy[qcopy + mbcnt(mcp)] = Item{NONE32, (uint32_t)bi, (uint32_t)(base + pos)};
            }
            qcopy += __popcll(mcp);
        }
    }
    bytes_in = wave_sum64(bytes_in);
    if (lane == 0) unit_bytes[u] = bytes_in;  // summed by k_sum_u64 (no contended atomics)
}

__global__ __launch_bounds__(1024) void k_sum_u64(const u64* __restrict__ v, u64 n, u64* __restrict__ out) {
    __shared__ u64 sb[16];
    u64 s = 0;
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = wave_sum64(s);
    if (lane_id() == 0) sb[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) t += sb[w];
        *out = t;
    }
}

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
__device__ __forceinline__ uint32_t vpopc(u32x4 v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

template <int OP>
__global__ __launch_bounds__(256) void k_bb(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                            OutView O, const BBItem* __restrict__ q, const u64* __restrict__ qrange,
                                            int cardmode, u64* pair_acc, GenItem* retry_q, uint32_t* retry_count) {
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
        const BBItem t = q[w];
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
            va[i] = vop<OP>(va[i], vb[i]);
            cnt += vpopc(va[i]);
        }
        const uint32_t card = wave_sum(cnt);
        if (cardmode) {
            if (lane == 0 && card) atomicAdd(&pair_acc[t.out], (u64)card);
            continue;
        }
        // result typing: OR is always a bitset (containers.h:1015-1020); and/xor/andnot are a
        // bitset iff card > 4096 (mixed_intersection.c:305-325, mixed_xor.c:260-273,
        // mixed_andnot.c:482-497)
        if (OP == OP_OR || card > 4096u) {
            u32x4* __restrict__ po = (u32x4*)(O.arena + O.off[t.out]);
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(va[i], po + i * 64 + lane);
            if (lane == 0) O.meta[t.out] = pack_meta(T_BITSET, card, 0);
        } else if (card == 0) {
            if (lane == 0) O.meta[t.out] = pack_meta(T_ARRAY, 0, 0);
        } else {
            // rare: result becomes an array -> re-queue for the LDS extraction kernel
            if (lane == 0) {
                GenItem g;
                g.offa = t.offa; g.offb = t.offb; g.out = t.out; g.ca = 65536u; g.cb = 65536u;
                g.types = (uint32_t)T_BITSET | ((uint32_t)T_BITSET << 8);
                g.nra = 0; g.nrb = 0; g.pad0 = 0; g.pad1 = 0;
                retry_q[atomicAdd(retry_count, 1u)] = g;
            }
        }
    }
}

// ------------------------------------------------------------------ pass-through copy
__global__ __launch_bounds__(256) void k_copy(PoolView A, PoolView B, OutView O, const Item* __restrict__ q,
                                              const u64* __restrict__ qrange) {
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
        Item t = q[w];
        const PoolView& S = (t.b == NONE32) ? A : B;
        const uint32_t c = (t.b == NONE32) ? t.a : t.b;
        const uint8_t ty = S.type[c];
        const uint32_t card = S.card[c], nr = S.nruns[c];
        const uint32_t n16 = (payload_bytes(ty, card, nr) + 15u) >> 4;
        const uint4* __restrict__ ps = (const uint4*)(S.arena + S.off[c]);
        uint4* __restrict__ po = (uint4*)(O.arena + O.off[t.out]);
        for (uint32_t i = lane; i < n16; i += 64) po[i] = ps[i];
        if (lane == 0) O.meta[t.out] = pack_meta(ty, card, nr);
    }
}

__device__ int decide_type(int op, int ta, int tb, uint32_t ca, uint32_t cb, bool fulla, bool fullb, uint32_t rc,
                           uint32_t rn);

// ------------------------------------------------------------------ array filter (K8, K9, K12)
// One WAVE per container pair, no workgroup barriers: the array operand Y is streamed 64 values at
// a time, each lane tests its value for membership in X and survivors are compacted with a ballot +
// mbcnt prefix.  X = bitset: one gathered dword test per value (array_bitset_container_intersection
// / _andnot, mixed_intersection.c:19-46, mixed_andnot.c:24-39).  X = array: X is first scattered into
// a wave-private 8 KiB LDS bitset (ds_or_b32), replacing the SIMD merge / galloping intersections of
// array_util.c:385-459, 801-906 (intersect_vector16, intersect_skewed_uint16) and difference_uint16.
// The result is always an array (containers.h:741-746, 1799-1803).
__global__ __launch_bounds__(256) void k_filter(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                OutView O, const FatItem* __restrict__ q,
                                                const u64* __restrict__ qrange, int op, int cardmode,
                                                u64* pair_acc) {
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* img = img_all[threadIdx.x >> 6];
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    FatItem tnext;
    if (w < n) tnext = q[w];
    for (; w < n; w += nwaves) {
        const FatItem t = tnext;
        if (w + nwaves < n) tnext = q[w + nwaves];  // next work item in flight while this one is processed
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        const uint32_t ca = t.ca, cb = t.cb;
        // Y = the streamed array, X = the membership side
        bool y_is_a = true;
        if (op == OP_AND) y_is_a = (ta == T_ARRAY) && (tb != T_ARRAY || ca <= cb);
        const uint8_t* yp = y_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint8_t* xp = y_is_a ? arenaB + t.offb : arenaA + t.offa;
        const uint32_t ny = y_is_a ? ca : cb, nx = y_is_a ? cb : ca;
        const bool x_bitset = (y_is_a ? tb : ta) == T_BITSET;
        const bool keep_present = op == OP_AND;
        const uint32_t* __restrict__ xw = (const uint32_t*)xp;
        const uint4* __restrict__ y4 = (const uint4*)yp;
        uint4 yfirst = make_uint4(0, 0, 0, 0);
        if (8 * lane < ny) yfirst = y4[lane];  // first 512 values of Y: in flight during the X scatter
        if (!x_bitset) {
            const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
            const uint4* __restrict__ x4 = (const uint4*)xp;  // 8 values per lane per step (slots are 16-byte padded)
            for (uint32_t i = lane; 8 * i < nx; i += 64) {
                const uint4 q4 = x4[i];
                const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (8 * i + h < nx) {
                        const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                        atomicOr(&img[v >> 5], 1u << (v & 31));
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        uint16_t* __restrict__ out = cardmode ? nullptr : (uint16_t*)(O.arena + O.off[t.out]);
        uint32_t run = 0;
        for (uint32_t base = 0; base < ny; base += 512) {
            const uint32_t i0 = base + 8 * lane;
            uint4 q4 = yfirst;
            if (base) q4 = (i0 < ny) ? y4[(base >> 3) + lane] : make_uint4(0, 0, 0, 0);
            const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
            uint32_t vals[8];
            uint32_t keepmask = 0;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                vals[h] = v;
                const uint32_t word = x_bitset ? xw[v >> 5] : img[v >> 5];
                const bool present = (word >> (v & 31)) & 1u;
                if (i0 + h < ny && present == keep_present) keepmask |= 1u << h;
            }
            const uint32_t cnt = __popc(keepmask);
            const uint32_t inc = wave_incl_scan(cnt);
            if (!cardmode) {
                uint32_t pos = run + inc - cnt;
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    if ((keepmask >> h) & 1u) out[pos++] = (uint16_t)vals[h];
            }
            run += __shfl(inc, 63);
        }
        if (cardmode) {
            if (lane == 0 && run) atomicAdd(&pair_acc[t.out], (u64)run);
        } else if (lane == 0) {
            O.meta[t.out] = pack_meta(T_ARRAY, run, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------ wave-private LDS image kernel (K6, K10, K11)
// One WAVE per container pair for {array,bitset} x {array,bitset} pairs with at least one array under
// or / xor, and bitset \ array.  The wave owns an 8 KiB LDS image: X is loaded into it (bitset: 8
// coalesced 16-byte loads per lane; array: zero + ds_or scatter), then Y's values are applied with
// returning LDS atomics (ds_or_rtn / ds_xor_rtn / ds_and_rtn) whose old values give the cardinality
// delta -- bitset_set_list_withcard / bitset_flip_list_withcard / bitset_clear_list
// (bitset_util.c:978-1141) without their serial dependence.  The result is typed by the reference's
// rules and either streamed out as a bitset or extracted as a sorted array (lane owns 32 consecutive
// words; wave prefix sum of popcounts).  No workgroup barrier anywhere.
__global__ __launch_bounds__(256) void k_wave(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                              OutView O, const FatItem* __restrict__ q,
                                              const u64* __restrict__ qrange, int op) {
    __shared__ __attribute__((aligned(16))) uint32_t img_all[4][2048];
    const uint32_t lane = lane_id();
    uint32_t* img = img_all[threadIdx.x >> 6];
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
        const FatItem t = q[w];
        const uint8_t ta = (uint8_t)(t.types & 0xFF), tb = (uint8_t)(t.types >> 8);
        const uint32_t ca = t.ca, cb = t.cb;
        // X = image side, Y = applied array.  andnot: X = a (bitset), Y = b.  or/xor are symmetric:
        // take the bitset (or the larger array) as X.
        bool x_is_a = true;
        if (op != OP_ANDNOT) x_is_a = (ta == T_BITSET) || (tb != T_BITSET && ca >= cb);
        const uint8_t tx = x_is_a ? ta : tb;
        const uint32_t cx = x_is_a ? ca : cb, cy = x_is_a ? cb : ca;
        const uint8_t* xp = x_is_a ? arenaA + t.offa : arenaB + t.offb;
        const uint32_t* __restrict__ y2 = (const uint32_t*)(x_is_a ? arenaB + t.offb : arenaA + t.offa);
        if (tx == T_BITSET) {
            const uint4* __restrict__ g = (const uint4*)xp;
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = g[i * 64 + lane];
        } else {
            const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) ((uint4*)img)[i * 64 + lane] = z;
            const uint4* __restrict__ x4 = (const uint4*)xp;
            for (uint32_t i = lane; 8 * i < cx; i += 64) {
                const uint4 q4 = x4[i];
                const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (8 * i + h < cx) {
                        const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                        atomicOr(&img[v >> 5], 1u << (v & 31));
                    }
                }
            }
        }
        int delta = 0;
        {
            const uint4* __restrict__ y4 = (const uint4*)y2;
            for (uint32_t i = lane; 8 * i < cy; i += 64) {
                const uint4 q4 = y4[i];
                const uint32_t d[4] = {q4.x, q4.y, q4.z, q4.w};
                uint32_t old[8], bit[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                    bit[h] = (8 * i + h < cy) ? (1u << (v & 31)) : 0u;
                    if (op == OP_OR) old[h] = atomicOr(&img[v >> 5], bit[h]);
                    else if (op == OP_XOR) old[h] = atomicXor(&img[v >> 5], bit[h]);
                    else old[h] = atomicAnd(&img[v >> 5], ~bit[h]);
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (bit[h]) {
                        const bool was = (old[h] & bit[h]) != 0;
                        if (op == OP_OR) delta += was ? 0 : 1;
                        else if (op == OP_XOR) delta += was ? -1 : 1;
                        else delta -= was ? 1 : 0;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) delta += __shfl_xor(delta, o);
        const uint32_t rc = (uint32_t)((int)cx + delta);
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, ta, tb, ca, cb, false, false, rc, 0);
        uint8_t* outp = O.arena + O.off[t.out];
        __builtin_amdgcn_wave_barrier();
        if (rc && ty == T_BITSET) {
            uint4* __restrict__ po = (uint4*)outp;
#pragma unroll
            for (int i = 0; i < 8; ++i) po[i * 64 + lane] = ((const uint4*)img)[i * 64 + lane];
        } else if (rc) {
            // Balanced extraction.  Words are owned strided (lane l: words 64 r + l), so clustered values
            // spread over all lanes; the output position of each word comes from a two-level prefix:
            // per-word popcounts -> LDS, each lane prefix-sums 32 CONSECUTIVE counts, one wave scan of the
            // lane totals, word bases back to LDS.  The image is dead once the words are in registers,
            // so its first 4 KiB hold the u16 count/base table.
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
            uint16_t* __restrict__ o16 = (uint16_t*)outp;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                uint32_t x = wv[r];
                uint32_t pos = tab[64 * r + lane];
                const uint32_t vbase = (64u * r + lane) * 32u;
                while (x) {
                    o16[pos++] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                    x &= x - 1;
                }
            }
        }
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, 0);
        __builtin_amdgcn_wave_barrier();
    }
}

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
    __shared__ __attribute__((aligned(16))) uint8_t lists_all[4][2][4 * RUNS_MAX_INTERVALS];
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
        uint8_t* outp = O.arena + O.off[t.out];
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
        uint8_t* outp = O.arena + O.off[t.out];
        __builtin_amdgcn_wave_barrier();
        if (rc && ty == T_BITSET) {
#pragma unroll
            for (int k = 0; k < 32; ++k) ia[wown(lane, k)] = r[k];
            __builtin_amdgcn_wave_barrier();
            uint4* __restrict__ po = (uint4*)outp;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t w0 = 4u * (i * 64 + lane);
                po[i * 64 + lane] = make_uint4(ia[wphys(w0)], ia[wphys(w0 + 1)], ia[wphys(w0 + 2)], ia[wphys(w0 + 3)]);
            }
        } else if (rc && ty == T_ARRAY) {
            uint16_t* st16 = (uint16_t*)ib;  // both images are dead: ib becomes the u16 staging buffer
            uint32_t pos = wave_incl_scan(cnt) - cnt;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                uint32_t x = r[k];
                const uint32_t vbase = (32u * lane + k) * 32u;
                while (x) {
                    st16[pos++] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                    x &= x - 1;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n16 = (2u * rc + 15u) >> 4;
            uint4* __restrict__ po = (uint4*)outp;
            for (uint32_t i = lane; i < n16; i += 64) po[i] = ((const uint4*)ib)[i];
        } else if (rc) {
            // runs: k-th start pairs with k-th end (run_container layout {value, length}, run.h:48-73)
            uint16_t* st16 = (uint16_t*)ib;
            uint32_t ne = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t nl = k < 31 ? (r[k + 1] & 1u) : next_lsb;
                ne += __popc(r[k] & ~((r[k] >> 1) | (nl << 31)));
            }
            uint32_t bs = wave_incl_scan(ns) - ns;
            uint32_t be = wave_incl_scan(ne) - ne;
            {
                uint32_t pm = prev_msb;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    uint32_t x = r[k] & ~((r[k] << 1) | pm);
                    pm = r[k] >> 31;
                    const uint32_t vbase = (32u * lane + k) * 32u;
                    while (x) {
                        st16[2 * bs] = (uint16_t)(vbase + (__ffs((int)x) - 1));
                        ++bs;
                        x &= x - 1;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t nl = k < 31 ? (r[k + 1] & 1u) : next_lsb;
                uint32_t x = r[k] & ~((r[k] >> 1) | (nl << 31));
                const uint32_t vbase = (32u * lane + k) * 32u;
                while (x) {
                    const uint32_t e = vbase + (__ffs((int)x) - 1);
                    st16[2 * be + 1] = (uint16_t)(e - st16[2 * be]);
                    ++be;
                    x &= x - 1;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n16 = (4u * rn + 15u) >> 4;
            uint4* __restrict__ po = (uint4*)outp;
            for (uint32_t i = lane; i < n16; i += 64) po[i] = ((const uint4*)ib)[i];
        }
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
    }
}

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

// ------------------------------------------------------------------ directory compaction
__global__ void k_flags(const u64* __restrict__ meta, u64 n, uint32_t* __restrict__ flag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = meta_card(meta[i]) ? 1u : 0u;
}
struct DirOut {
    u64* bm_start;
    u64* key;
    uint8_t* type;
    uint32_t* card;
    uint32_t* nruns;
    u64* off;
};
// grid-stride, 1024 threads per block: one pair of atomics per block for the statistics
__global__ __launch_bounds__(1024) void k_compact(OutView O, u64 n, const u64* __restrict__ newidx, DirOut R,
                                                  Stats* stats) {
    __shared__ u64 sb[16];
    __shared__ uint32_t sk[16];
    u64 bytes = 0;
    uint32_t keep = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 m = O.meta[i];
        if (meta_card(m)) {
            const u64 d = newidx[i];
            const uint32_t ty = meta_type(m);
            R.key[d] = O.key[i];
            R.type[d] = (uint8_t)ty;
            R.card[d] = meta_card(m);
            R.nruns[d] = meta_nruns(m);
            R.off[d] = O.off[i];
            bytes += payload_bytes((uint8_t)ty, meta_card(m), meta_nruns(m));
            keep++;
        }
    }
    bytes = wave_sum64(bytes);
    keep = wave_sum(keep);
    if (lane_id() == 0) { sb[threadIdx.x >> 6] = bytes; sk[threadIdx.x >> 6] = keep; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 b = 0; uint32_t k = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) { b += sb[w]; k += sk[w]; }
        if (k) { atomicAdd(&stats->bytes_out, b); atomicAdd(&stats->result_containers, (u64)k); }
    }
}
__global__ void k_bm_start(const u64* __restrict__ cand_start, const u64* __restrict__ pair_unit0, uint32_t npairs,
                           const u64* __restrict__ newidx, u64* __restrict__ bm_start) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= npairs) bm_start[p] = newidx[cand_start[pair_unit0[p]]];
}

// per-bitmap cardinality = sum of container cardinalities (roaring.c:1436-1443); wave per bitmap
__global__ __launch_bounds__(256) void k_bitmap_cards(PoolView P, uint32_t nbm, u64* __restrict__ out) {
    uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= nbm) return;
    u64 s = 0;
    for (u64 i = P.bm_start[b] + lane_id(); i < P.bm_start[b + 1]; i += 64) s += P.card[i];
    s = wave_sum64(s);
    if (lane_id() == 0) out[b] = s;
}
__global__ __launch_bounds__(256) void k_payload_stats(const uint8_t* type, const uint32_t* card,
                                                       const uint32_t* nruns, u64 n, u64* out /*[4]*/) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 bytes = 0;
    uint32_t nb = 0, na = 0, nr = 0;
    if (i < n) {
        uint8_t t = type[i];
        bytes = payload_bytes(t, card[i], nruns[i]);
        nb = t == T_BITSET; na = t == T_ARRAY; nr = t == T_RUN;
    }
    bytes = wave_sum64(bytes); nb = wave_sum(nb); na = wave_sum(na); nr = wave_sum(nr);
    if (lane_id() == 0) {
        if (bytes) atomicAdd(&out[0], bytes);
        if (nb) atomicAdd(&out[1], (u64)nb);
        if (na) atomicAdd(&out[2], (u64)na);
        if (nr) atomicAdd(&out[3], (u64)nr);
    }
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
