// rhip_plan.h -- planning (key merge -> typed work items), the single-pass prefix scan that places them, and the
// fused result-directory compaction of the batched pairwise pipeline
//
// One rhip_pairwise call is FIVE dependent launches and one host synchronisation at the very end:
//   k_count   per unit (a tile of <= 256 directory entries of one side of one pair): matched / per-class counts,
//             result-slot bytes, algorithmic input bytes; remembers every key's match position; zeroes the scan /
//             statistics scratch of the call
//   k_scan    ONE decoupled look-back exclusive scan over the 10 concatenated count sections
//   k_emit    candidates in merged key order, slot offsets, work items at deterministic queue positions
//   (class kernels: k_bb / k_filter / k_wave / k_ivl / k_genw / k_copy ..., concurrently on auxiliary streams)
//   k_tail    drop empty results + build the result directory + per-bitmap starts + statistics, one look-back pass
// Nothing is read back in between: every buffer is sized from host-side upper bounds (container counts and
// per-bitmap payload bounds mirrored on the host), every kernel takes its item count from device memory.
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ decoupled look-back (single-pass scan)
// status word of a tile: bits 63..62 = state (0 not ready, 1 tile aggregate, 2 inclusive prefix), bits 61..0 = value
#define LB_AGG (1ull << 62)
#define LB_PREFIX (2ull << 62)
#define LB_VAL(x) ((x) & ((1ull << 62) - 1))
struct LbState {
    u64* status;        // [n_tiles], zero before the kernel starts
    uint32_t* ticket;   // dynamic tile numbering: tiles are numbered in the order their blocks START, so a tile only
                        // ever waits for tiles whose blocks are already running (no dependence on dispatch order)
};
__device__ __forceinline__ u64 lb_load(const u64* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void lb_store(u64* p, u64 v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
// Called by the 64 lanes of ONE wave of the block with the block's aggregate; returns (to every lane) the exclusive
// prefix of the tile.  The look-back is wave-parallel: the lanes read the status of the 64 preceding tiles at once,
// the nearest tile that already published an inclusive prefix ends the walk, the aggregates in front of it are summed
// with one wave reduction.  (A one-thread walk cost one global round trip per preceding tile: 0.3 ms for the 1 400
// tiles of a 2.9 M-candidate batch.)
__device__ __forceinline__ u64 lb_exclusive_prefix(u64* status, uint32_t tile, u64 aggregate) {
    const uint32_t lane = lane_id();
    if (tile == 0) {
        if (lane == 0) lb_store(&status[0], LB_PREFIX | aggregate);
        return 0;
    }
    if (lane == 0) lb_store(&status[tile], LB_AGG | aggregate);
    u64 run = 0;
    long long base = (long long)tile - 1;  // lane l looks at tile base - l
    for (;;) {
        const long long idx = base - (long long)lane;
        const u64 s = idx >= 0 ? lb_load(&status[idx]) : LB_PREFIX;  // in front of tile 0: prefix 0
        const u64 ready = __ballot((s >> 62) != 0), pref = __ballot((s >> 62) == 2);
        if (pref) {
            const uint32_t first = (uint32_t)__ffsll((long long)pref) - 1u;  // nearest published prefix
            const u64 need = first == 63u ? ~0ull : ((1ull << (first + 1u)) - 1ull);
            if ((ready & need) != need) continue;  // an aggregate in front of it is not there yet: look again
            run += wave_sum64(lane <= first ? LB_VAL(s) : 0ull);
            break;
        }
        if (ready != ~0ull) continue;
        run += wave_sum64(LB_VAL(s));
        base -= 64;
    }
    if (lane == 0) lb_store(&status[tile], LB_PREFIX | (run + aggregate));
    return run;
}
__device__ __forceinline__ u64 wave_incl_scan64(u64 v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)v, o), hi = __shfl_up((uint32_t)(v >> 32), o);
        if (lane_id() >= (uint32_t)o) v += (u64)lo | ((u64)hi << 32);
    }
    return v;
}
// exclusive prefix of v over the 256 threads of the block (u64), *total = block sum; sm: 4 u64 of shared memory
__device__ __forceinline__ u64 blk_exscan64(u64 v, u64* sm, u64* total) {
    const u64 inc = wave_incl_scan64(v);
    __syncthreads();
    if (lane_id() == 63) sm[threadIdx.x >> 6] = inc;
    __syncthreads();
    u64 off = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        const u64 s = sm[w];
        if (w < (threadIdx.x >> 6)) off += s;
        tot += s;
    }
    *total = tot;
    return off + inc - v;
}

constexpr uint32_t SCAN_TILE = 2048;  // elements per block: 256 threads x 8
// out[i] = sum_{j<i} in[j], i in [0, n).  `ranges` (may be null): for every section k of `sec_len` elements,
// ranges[2k] = out[k * sec_len] and ranges[2k+1] = out[k * sec_len + sec_len - 1] (begin / end of the section when its
// last element is a zero sentinel).
// Elements behind the n_sec sections (the X-group histogram of a grouped batch, rhip_common.h): every xlen-th of them
// marks a group boundary, ranges[2 n_sec + k] = out[n_sec * sec_len + k * xlen].
__global__ __launch_bounds__(256) void k_scan(const uint32_t* __restrict__ in, u64* __restrict__ out, u64 n, LbState lb,
                                              u64* __restrict__ ranges, u64 sec_len, u64 n_sec, u64 xlen) {
    __shared__ u64 sm[4];
    __shared__ uint32_t s_tile;
    __shared__ u64 s_prefix;
    if (threadIdx.x == 0) s_tile = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const u64 base = (u64)tile * SCAN_TILE + 8ull * threadIdx.x;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base + k < n ? in[base + k] : 0u;
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) mine += v[k];
    u64 total;
    u64 ex = blk_exscan64(mine, sm, &total);
    if (threadIdx.x < 64) {
        const u64 pfx = lb_exclusive_prefix(lb.status, tile, total);
        if (threadIdx.x == 0) s_prefix = pfx;
    }
    __syncthreads();
    ex += s_prefix;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 i = base + k;
        if (i < n) {
            out[i] = ex;
            if (ranges) {
                const u64 sec = i / sec_len, r = i - sec * sec_len;
                if (sec < n_sec) {
                    if (r == 0) ranges[2 * sec] = ex;
                    if (r == sec_len - 1) ranges[2 * sec + 1] = ex;
                } else if (xlen) {
                    const u64 x = i - n_sec * sec_len;
                    if (x % xlen == 0) ranges[2 * n_sec + x / xlen] = ex;
                }
            }
        }
        ex += v[k];
    }
}

// ------------------------------------------------------------------ batch description, host -> device
// A small batch description (two index lists) is pulled from pinned host memory by a kernel on the compute queue:
// fully coalesced 16-byte reads over PCIe, no dependent chain.  A copy command would run on the DMA engine, with one
// cross-engine dependency behind the previous batch's last kernel and one in front of k_count.
__global__ __launch_bounds__(256) void k_stage_in(const uint4* __restrict__ host_src, uint4* __restrict__ dst, uint32_t n16) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = host_src[i];
}
// A batch over a prepared pair list whose plan is cached (rhip_engine.hip, PlanCache) starts here instead of at k_count:
// the call's scratch words are cleared as k_count clears them -- except the section ranges, which come back from the
// cache entry -- and, in cardinality mode, the per-pair accumulators.
__global__ __launch_bounds__(256) void k_plan_restore(u64* __restrict__ words, uint32_t n_words, uint32_t w_ranges,
                                                      const u64* __restrict__ saved, uint32_t n_ranges,
                                                      u64* __restrict__ pair_acc, uint32_t n_pairs) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (uint32_t i = gid; i < n_words; i += nth) {
        const uint32_t r = i - w_ranges;  // (unsigned: below the ranges it wraps around)
        words[i] = r < n_ranges ? saved[r] : 0ull;
    }
    if (pair_acc)
        for (uint32_t i = gid; i < n_pairs; i += nth) pair_acc[i] = 0ull;
}
// ... and the first batch of that plan leaves its section ranges with the cache entry (behind k_emit on the plan's stream)
__global__ void k_ranges_save(const u64* __restrict__ ranges, u64* __restrict__ saved, uint32_t n) {
    if (threadIdx.x < n) saved[threadIdx.x] = ranges[threadIdx.x];
}

// ------------------------------------------------------------------ planning
// Planning works on UNITS: one unit = one tile of up to 256 consecutive directory entries of the
// left bitmap of a pair ("A-tile"), or -- for OR/XOR, whose result also carries the right bitmap's
// unmatched containers -- of the right bitmap ("B-tile").  One wave per unit, so a batch of 250 pairs
// of 4096-container bitmaps plans on 4000 waves instead of 250.
// Count arrays (and their exclusive scan) have N_SEC sections of n_units+1 entries (the last one a zero sentinel):
enum { SEC_CAND = 0, SEC_M = 1, SEC_BB = 2, SEC_GEN = 3, SEC_COPY = 4, SEC_FILT = 5, SEC_WAVE = 6, SEC_RUNS = 7,
       SEC_SLOT = 8,   // result-slot size of the unit's candidates, in 16-byte units
       SEC_BYTES = 9,  // algorithmic input bytes of the unit (payload of matched operands and pass-through containers)
       SEC_PROBE = 10,
       SEC_BBA = 11,
       SEC_USMALL = 12,
       SEC_RUNS16 = 13,
       SEC_RUNS16W = 14,
       SEC_BA = 15,
       N_SEC = 16 };
// work class of a matched container pair
// ia / ib = number of intervals of the operand when it is read as an interval list (runs: n_runs, arrays: card)
// ca / cb = cardinalities
__device__ __forceinline__ int classify(int op, int cardmode, uint8_t ta, uint8_t tb, uint32_t ia, uint32_t ib,
                                        uint32_t ca, uint32_t cb) {
    if (ta == T_BITSET && tb == T_BITSET) {
        // Two bitsets whose and / andnot is EXPECTED (cardinalities, independence) to fall to <= 4096 values take
        // the kernel that can emit an array straight away; everything else streams through k_bb, which re-queues
        // the odd array result.  Either kernel is correct for either outcome: this only picks the cheaper one.
        if (!cardmode && op == OP_AND && (u64)ia * ib <= 4096ull * 65536ull) return CLS_BBA;
        if (!cardmode && op == OP_ANDNOT && (u64)ia * (65536u - ib) <= 4096ull * 65536ull) return CLS_BBA;
        return CLS_BB;
    }
    // interval algebra in O(n log n) when a run container meets a run / a short array
    if ((ta == T_RUN || tb == T_RUN) && ta != T_BITSET && tb != T_BITSET && ia <= RUNS_MAX_INTERVALS &&
        ib <= RUNS_MAX_INTERVALS) {
        // short lists with few values (the containers of sparse, run-compressed data): four pairs per wave
        if (ia <= R16_MAX_IV && ib <= R16_MAX_IV && ca + cb <= R16_MAX_CARD) return CLS_RUNS16;
        if (ia <= R16W_MAX_IV && ib <= R16W_MAX_IV && ca + cb <= R16W_MAX_CARD) return CLS_RUNS16W;
        return CLS_RUNS;
    }
    // array filtered by membership in an array / bitset: and (either order), array \ x; a short streamed array
    // (the smaller one when both are arrays) probes global memory directly, a long one goes through the LDS image
    if (cardmode || op == OP_AND) {
        if ((ta == T_ARRAY && tb != T_RUN) || (tb == T_ARRAY && ta != T_RUN)) {
            const uint32_t ny = (ta == T_ARRAY && tb == T_ARRAY) ? (ia < ib ? ia : ib) : (ta == T_ARRAY ? ia : ib);
            return ny <= PROBE_MAX ? CLS_PROBE : CLS_FILT;
        }
    } else if (op == OP_ANDNOT) {
        if (ta == T_ARRAY && tb != T_RUN) return ia <= PROBE_MAX ? CLS_PROBE : CLS_FILT;
        if (ta == T_BITSET && tb == T_ARRAY) return CLS_BA;  // bitset \\ array: the bitset stays in registers
    } else {
        // or / xor of two arrays, one of them short and the sum small enough that the result is an array whatever its
        // cardinality (mixed_union.c:162-191, mixed_xor.c:196-219): the short one is merged INTO the long one by rank
        if (ta == T_ARRAY && tb == T_ARRAY && (ia < ib ? ia : ib) <= USMALL_MAX && ia + ib <= 4096u) return CLS_USMALL;
        if ((ta == T_BITSET && tb == T_ARRAY) || (ta == T_ARRAY && tb == T_BITSET)) return CLS_BA;  // bitset | ^ array
        if (ta != T_RUN && tb != T_RUN) return CLS_WAVE;       // or / xor of two arrays through the image
    }
    return CLS_GEN;
}
#define UNIT_B 0x80000000u

struct UnitView {
    const uint32_t* pair;   // [U] pair index of the unit
    const uint32_t* tile;   // [U] tile index inside its side; UNIT_B flag marks a B-tile
    const u64* pair_unit0;  // [npairs+1] first unit of each pair
    uint32_t n_units;
    uint32_t n_pairs;
    uint32_t implicit;      // 0: the arrays above; 1 / 2: every pair has exactly that many one-tile units (A, then B)
                            //    and the arrays are not there (no bitmap of the batch has more than 256 containers)
    // Multi-op batches (rhip_pairwise_multi): n_pairs = n_ops x n_real VIRTUAL pairs, virtual pair v = op index
    // v / n_real over real pair v % n_real (the index into lhs / rhs); ops = the ops, two bits each.  A plain batch has
    // n_real = n_pairs and its one op in bits 0..1.
    uint32_t n_real;
    uint32_t ops;
};
__device__ __forceinline__ int unit_op(const UnitView& U, uint32_t vpair) { return (int)((U.ops >> (2u * (vpair / U.n_real))) & 3u); }
struct UnitId { uint32_t pair; bool bside; u64 tile; u64 unit0; };
__device__ __forceinline__ UnitId unit_id(const UnitView& U, uint32_t u) {
    UnitId r;
    if (U.implicit) {
        r.pair = U.implicit == 2 ? u >> 1 : u;
        r.bside = U.implicit == 2 && (u & 1u);
        r.tile = 0;
        r.unit0 = (u64)r.pair * U.implicit;
    } else {
        r.pair = U.pair[u];
        r.bside = (U.tile[u] & UNIT_B) != 0;
        r.tile = U.tile[u] & ~UNIT_B;
        r.unit0 = U.pair_unit0[r.pair];
    }
    return r;
}
struct PlanZero {   // scratch the first kernel of a call clears for the later ones
    u64* words;     // scan states, tickets, retry counter, statistics: one contiguous region
    uint32_t n_words;
    u64* pair_acc;  // cardinality mode: per-pair accumulators (n_pairs entries), else null
};

// first index in [lo,hi) with key[idx] >= k, four searches per lane issued together (independent load chains)
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

// the same search over a key list staged in LDS (indices relative to the list's start)
__device__ __forceinline__ void lower_bound4_lds(const u64* lk, uint32_t n, const u64 k[4], const bool act[4], u64 out[4]) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { lo[t] = 0; hi[t] = act[t] ? n : 0u; }
    bool more = true;
    while (more) {
        more = false;
        uint32_t mid[4];
        u64 kv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { mid[t] = (lo[t] + hi[t]) >> 1; kv[t] = (lo[t] < hi[t]) ? lk[mid[t]] : 0; }
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

// Unit of group `grp` of planning wave `w`.  With two units per pair (A side, B side) the groups of a wave take the SAME
// side of consecutive pairs, so that the side-dependent code of k_emit stays wave-uniform around its collectives.
// The launch needs ceil(n_units / (64 / G)) + 1 waves.
template <uint32_t G>
__device__ __forceinline__ uint32_t unit_of_group(const UnitView& U, uint32_t w, uint32_t grp) {
    constexpr uint32_t NG = 64 / G;
    if (G == 64) return w;
    if (U.implicit == 2) return 2u * ((w >> 1) * NG + grp) + (w & 1u);
    return w * NG + grp;
}

// match[] entry of a directory element: position of its key in the other side's range (lower bound, relative to the
// range start) and whether the key is present there
#define MATCH_FOUND 0x80000000u

// One group of G lanes per unit (G = 64: tiles of up to 256 directory entries, four rounds of 64; G = 16: four units
// per wave when no bitmap of the batch has more than 64 containers -- the realdata sets with 10-50 containers per
// bitmap otherwise leave three quarters of every planning wave idle): contributions of the unit to every section; the
// match positions are kept for k_emit.
constexpr u64 TAIL_EARLY = 1ull << 63;  // bit of a batch's completion word: "published by a tail that may have run too early"
template <uint32_t G>
__global__ __launch_bounds__(256) void k_count(PoolView A, PoolView B, const uint32_t* __restrict__ lhs,
                                               const uint32_t* __restrict__ rhs, UnitView U, int cardmode,
                                               uint32_t* __restrict__ counts, uint32_t* __restrict__ match, PlanZero Z,
                                               XGroupView X) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    const size_t S = (size_t)U.n_units + 1;
    for (u64 i = gid; i < Z.n_words; i += nthreads) Z.words[i] = 0;
    if (Z.pair_acc)
        for (u64 i = gid; i < U.n_pairs; i += nthreads) Z.pair_acc[i] = 0;
    if (gid < N_SEC) counts[gid * S + U.n_units] = 0;  // section sentinels
    const Grp<G> gr;
    // (one unit per wave: its index, pair, ranges and op are wave-uniform -- scalar loads, scalar branches)
    const uint32_t u = G == 64 ? wave_uniform(unit_of_group<G>(U, (uint32_t)(gid >> 6), gr.grp)) : unit_of_group<G>(U, (uint32_t)(gid >> 6), gr.grp);
    if (u >= U.n_units) return;
    const uint32_t lane = gr.gl;
    const UnitId uid = unit_id(U, u);
    const uint32_t p = uid.pair % U.n_real;
    const int op = unit_op(U, uid.pair);
    const bool bside = uid.bside;
    const u64 tile = uid.tile;
    const u64 a0 = A.bm_start[lhs[p]], a1 = A.bm_start[lhs[p] + 1];
    const u64 b0 = B.bm_start[rhs[p]], b1 = B.bm_start[rhs[p] + 1];
    // s* = the side this tile walks, l* = the side it searches
    const PoolView& SV = bside ? B : A;
    const PoolView& LV = bside ? A : B;
    // (a multi-op batch gives every virtual pair a B-side unit; under and / andnot it has nothing to contribute)
    const bool dead = bside && !(op == OP_OR || op == OP_XOR);
    const u64 s0 = (bside ? b0 : a0) + tile * (4 * G), sEnd = dead ? s0 : (bside ? b1 : a1);
    const u64 s1 = s0 + 4 * G < sEnd ? s0 + 4 * G : sEnd;
    const u64 l0 = bside ? a0 : b0, l1 = bside ? a1 : b1;
    u64 k[4], j[4];
    bool act[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        act[t] = s0 + G * t + lane < s1;
        k[t] = act[t] ? SV.key[s0 + G * t + lane] : 0;
    }
    // The partner's key list (the side that is searched) is staged in LDS when it fits the group's 4 G entries -- always
    // with implicit units -- and searched there: the binary search was a chain of log2(n) dependent trips to the L2 (seven
    // of the ~12 of this kernel on C5's 95-container bitmaps: k_count<64> 40 us for 39 800 units), now one coalesced load.
    __shared__ u64 s_keys[4][256];
    u64* lk = s_keys[threadIdx.x >> 6] + gr.grp * (4 * G);
    const uint32_t ln = (uint32_t)(l1 - l0);
    const bool staged = __ballot(l1 - l0 > (u64)(4 * G)) == 0ull;  // (wave-uniform)
    if (staged) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (G * t + lane < ln) lk[G * t + lane] = LV.key[l0 + G * t + lane];
        __builtin_amdgcn_wave_barrier();
        lower_bound4_lds(lk, ln, k, act, j);
#pragma unroll
        for (int t = 0; t < 4; ++t) j[t] += l0;
    } else {
        lower_bound4(LV.key, l0, l1, k, act, j);
    }
    uint32_t matched = 0, nbb = 0, nfilt = 0, nwave = 0, nruns_cls = 0, nprobe = 0, nbba = 0, nusm = 0, nr16 = 0, nr16w = 0, nba = 0, slot16 = 0, bytes = 0;
    const bool keep_unmatched = bside || !(cardmode || op == OP_AND);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        // (a bitmap with fewer than G t containers leaves the later rounds empty: skipped when that holds for every
        // unit of the wave -- the planning kernels are bound by instruction issue, not by memory)
        if (__ballot(act[t]) == 0ull) continue;
        const bool found = act[t] && j[t] < l1 && (staged ? lk[j[t] - l0] : LV.key[j[t]]) == k[t];
        if (act[t]) match[(size_t)u * (4 * G) + G * t + lane] = (uint32_t)(j[t] - l0) | (found ? MATCH_FOUND : 0u);
        int cls = -1;
        if (act[t] && (bside ? !found : (found || keep_unmatched))) {
            const u64 si = s0 + G * t + lane;
            const uint8_t ts = SV.type[si];
            const uint32_t cs = SV.card[si], ns = SV.nruns[si];
            const uint32_t ps = payload_bytes(ts, cs, ns);
            bytes += ps;
            if (found) {  // A-tile, matched
                const uint8_t tl = LV.type[j[t]];
                const uint32_t cl = LV.card[j[t]], nl = LV.nruns[j[t]];
                bytes += payload_bytes(tl, cl, nl);
                cls = classify(op, cardmode, ts, tl, ts == T_RUN ? ns : cs, tl == T_RUN ? nl : cl, cs, cl);
                if (!cardmode) slot16 += matched_slot(op, cs, cl) >> 4;
                if (X.on && (cls == CLS_FILT || cls == CLS_WAVE || cls == CLS_BA)) {  // grouped queues: one more item for its X
                    const bool x_in_a = cls == CLS_FILT ? !filt_y_is_a(op, ts, tl, cs, cl) : union_x_is_a(op, ts, tl, cs, cl);
                    atomicAdd(&X.hist[xg_index(X, cls == CLS_FILT ? XG_FILT : XG_UNION, x_in_a, si, j[t])], 1u);
                }
            } else if (!cardmode) {
                const uint32_t sl = align16(ps) >> 4;
                slot16 += sl ? sl : 1u;
            }
        }
        matched += (uint32_t)__popcll(gr.ballot(found));
        nbb += (uint32_t)__popcll(gr.ballot(cls == CLS_BB));
        nfilt += (uint32_t)__popcll(gr.ballot(cls == CLS_FILT));
        nwave += (uint32_t)__popcll(gr.ballot(cls == CLS_WAVE));
        nruns_cls += (uint32_t)__popcll(gr.ballot(cls == CLS_RUNS));
        nprobe += (uint32_t)__popcll(gr.ballot(cls == CLS_PROBE));
        nbba += (uint32_t)__popcll(gr.ballot(cls == CLS_BBA));
        nusm += (uint32_t)__popcll(gr.ballot(cls == CLS_USMALL));
        nr16 += (uint32_t)__popcll(gr.ballot(cls == CLS_RUNS16));
        nr16w += (uint32_t)__popcll(gr.ballot(cls == CLS_RUNS16W));
        nba += (uint32_t)__popcll(gr.ballot(cls == CLS_BA));
    }
    slot16 = gr.sum(slot16);
    bytes = gr.sum(bytes);
    if (lane == 0) {
        const uint32_t n = (uint32_t)(s1 - s0);
        uint32_t ncopy;
        if (bside) ncopy = n - matched;                              // OR/XOR only
        else ncopy = (cardmode || op == OP_AND) ? 0u : n - matched;  // A-only containers pass through
        counts[SEC_CAND * S + u] = bside ? ncopy : matched + ncopy;
        counts[SEC_M * S + u] = matched;
        counts[SEC_BB * S + u] = nbb;
        counts[SEC_GEN * S + u] = bside ? 0u : matched - nbb - nfilt - nwave - nruns_cls - nprobe - nbba - nusm - nr16 - nr16w - nba;
        counts[SEC_PROBE * S + u] = nprobe;
        counts[SEC_BBA * S + u] = nbba;
        counts[SEC_USMALL * S + u] = nusm;
        counts[SEC_RUNS16 * S + u] = nr16;
        counts[SEC_RUNS16W * S + u] = nr16w;
        counts[SEC_RUNS * S + u] = nruns_cls;
        counts[SEC_FILT * S + u] = nfilt;
        counts[SEC_WAVE * S + u] = nwave;
        counts[SEC_BA * S + u] = nba;
        counts[SEC_COPY * S + u] = ncopy;
        counts[SEC_SLOT * S + u] = slot16;
        counts[SEC_BYTES * S + u] = bytes;
    }
}

// One wave per unit: emit candidates in merged key order (roaring.c:742-768, 895-951) and the work
// items of each class at deterministic queue positions (no atomics).  The position of a candidate
// inside its result bitmap is computed by ranking, not by a serial merge:
//   matched / A-only element i (key k):  i + |{B keys < k}| - |{matched keys < k}|
//   B-only element j (key k)          :  j + |{A keys < k}| - |{matched keys < k}|
// with |{matched keys < k}| = (matched count of the pair's earlier tiles, from the scan) + a ballot rank.
// Result slots are laid out in unit order (A-tiles of a pair, then its B-tiles) and, inside a unit, in lane order:
// offset = 16 x (scanned slot units before this unit + wave prefix inside it).  A bitmap's slots are contiguous.
struct EmitQueues {
    BBItem* bb;     // section SEC_BB
    GenItem* gen;   // section SEC_GEN
    CopyItem* copy; // section SEC_COPY
    FatItem* filt;  // section SEC_FILT
    FatItem* wave;  // section SEC_WAVE
    GenItem* runs;  // section SEC_RUNS
    FatItem* probe; // section SEC_PROBE
    BBItem* bba;    // section SEC_BBA
    FatItem* usmall; // section SEC_USMALL
    GenItem* runs16; // section SEC_RUNS16
    GenItem* runs16w; // section SEC_RUNS16W
    FatItem* ba;      // section SEC_BA
};
struct CandOut {     // candidate (pre-compaction) result directory
    u64* key;        // [cand]
    u64* off;        // [cand] byte offset of the slot in the result arena
    uint32_t* pair;  // [cand] result bitmap (pair index) of the candidate
};
template <uint32_t G>
__global__ __launch_bounds__(256) void k_emit(PoolView A, PoolView B, const uint32_t* __restrict__ lhs,
                                              const uint32_t* __restrict__ rhs, UnitView U, int cardmode,
                                              const u64* __restrict__ starts, const uint32_t* __restrict__ match,
                                              CandOut O, EmitQueues Q, XGroupView X) {
    const Grp<G> gr;
    const uint32_t u = G == 64 ? wave_uniform(unit_of_group<G>(U, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, gr.grp))
                               : unit_of_group<G>(U, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, gr.grp);
    if (u >= U.n_units) return;
    const uint32_t lane = gr.gl;
    const size_t S = (size_t)U.n_units + 1;
    const UnitId uid = unit_id(U, u);
    const uint32_t p = uid.pair;               // result bitmap (virtual pair)
    const uint32_t pr = uid.pair % U.n_real;   // index into the pair list
    const int op = unit_op(U, uid.pair);
    const uint32_t opbits = (uint32_t)op << ITEM_OP_SHIFT;
    const bool bside = uid.bside;
    const u64 tile = uid.tile;
    const u64 a0 = A.bm_start[lhs[pr]], a1 = A.bm_start[lhs[pr] + 1];
    const u64 b0 = B.bm_start[rhs[pr]], b1 = B.bm_start[rhs[pr] + 1];
    if (bside && !(op == OP_OR || op == OP_XOR)) return;  // (multi-op batches: the B-side unit of an and / andnot pair)
    const u64 u0 = uid.unit0;
    const u64 base = starts[SEC_CAND * S + u0];
    u64 qbb = starts[SEC_BB * S + u] - starts[SEC_BB * S];
    u64 qgen = starts[SEC_GEN * S + u] - starts[SEC_GEN * S];
    u64 qcopy = starts[SEC_COPY * S + u] - starts[SEC_COPY * S];
    u64 qfilt = starts[SEC_FILT * S + u] - starts[SEC_FILT * S];
    u64 qwave = starts[SEC_WAVE * S + u] - starts[SEC_WAVE * S];
    u64 qruns = starts[SEC_RUNS * S + u] - starts[SEC_RUNS * S];
    u64 qprobe = starts[SEC_PROBE * S + u] - starts[SEC_PROBE * S];
    u64 qbba = starts[SEC_BBA * S + u] - starts[SEC_BBA * S];
    u64 qusm = starts[SEC_USMALL * S + u] - starts[SEC_USMALL * S];
    u64 qr16 = starts[SEC_RUNS16 * S + u] - starts[SEC_RUNS16 * S];
    u64 qr16w = starts[SEC_RUNS16W * S + u] - starts[SEC_RUNS16W * S];
    u64 qba = starts[SEC_BA * S + u] - starts[SEC_BA * S];
    u64 slot_run = 16ull * (starts[SEC_SLOT * S + u] - starts[SEC_SLOT * S]);  // arena offset of the unit's first slot
    if (!bside) {
        const u64 s0 = a0 + tile * (4 * G);
        uint32_t mbefore = (uint32_t)(starts[SEC_M * S + u] - starts[SEC_M * S + u0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u64 ai = s0 + G * t + lane;
            const bool act = ai < a1;
            if (__ballot(act) == 0ull) break;  // (rounds past the end of every unit of the wave)
            const uint32_t mt = act ? match[(size_t)u * (4 * G) + G * t + lane] : 0u;
            const bool found = (mt & MATCH_FOUND) != 0;
            const uint32_t lbcount = mt & ~MATCH_FOUND;
            const u64 bj = b0 + lbcount;
            const u64 fm = gr.ballot(found);
            const uint32_t mb = mbefore + gr.rank(fm);
            mbefore += (uint32_t)__popcll(fm);
            const bool emit = act && (found || (!cardmode && op != OP_AND));
            uint8_t ta = 0, tb = 0;
            uint32_t ca = 0, cb = 0, pa = 0, pos = 0, nra = 0, nrb = 0, sl = 0;
            u64 key = 0;
            if (emit) {
                const uint32_t ilocal = (uint32_t)(ai - a0);
                if (op == OP_AND || cardmode) pos = mb;
                else if (op == OP_ANDNOT) pos = ilocal;
                else pos = ilocal + lbcount - mb;
                key = A.key[ai];
                ta = A.type[ai];
                ca = A.card[ai];
                nra = A.nruns[ai];
                pa = payload_bytes(ta, ca, nra);
                if (found) {
                    tb = B.type[bj];
                    cb = B.card[bj];
                    nrb = B.nruns[bj];
                }
                if (!cardmode) {
                    sl = found ? matched_slot(op, ca, cb) : align16(pa);
                    sl = sl < 16u ? 16u : sl;
                }
            }
            const uint32_t inc = gr.incl_scan(sl);
            const u64 offo = slot_run + inc - sl;
            slot_run += __shfl(inc, gr.glast);
            if (emit && !cardmode) {
                O.key[base + pos] = key;
                O.off[base + pos] = offo;
                O.pair[base + pos] = p;
            }
            const uint32_t outidx = cardmode ? p : (uint32_t)(base + pos);
            const int cls = (emit && found) ? classify(op, cardmode, ta, tb, ta == T_RUN ? nra : ca, tb == T_RUN ? nrb : cb, ca, cb) : -1;
            const bool isbb = cls == CLS_BB;
            const bool isbba = cls == CLS_BBA;
            const bool isusm = cls == CLS_USMALL;
            const bool isgen = cls == CLS_GEN;
            const bool isfilt = cls == CLS_FILT;
            const bool iswave = cls == CLS_WAVE;
            const bool isruns = cls == CLS_RUNS;
            const bool isr16 = cls == CLS_RUNS16;
            const bool isr16w = cls == CLS_RUNS16W;
            const bool isprobe = cls == CLS_PROBE;
            const bool isba = cls == CLS_BA;
            const bool iscopy = emit && !found;
            const u64 mbb = gr.ballot(isbb), mgen = gr.ballot(isgen), mcp = gr.ballot(iscopy), mfl = gr.ballot(isfilt);
            const u64 mwv = gr.ballot(iswave), mrn = gr.ballot(isruns), mpr = gr.ballot(isprobe), mba = gr.ballot(isbba);
            const u64 mus = gr.ballot(isusm), mr16 = gr.ballot(isr16), mr16w = gr.ballot(isr16w), mbar = gr.ballot(isba);
            if (isbb || isbba) {
                BBItem it;
                it.offa = A.off[ai]; it.offb = B.off[bj]; it.offo = offo; it.out = outidx; it.slot = sl | opbits;
                if (isbb) Q.bb[qbb + gr.rank(mbb)] = it;
                else Q.bba[qbba + gr.rank(mba)] = it;
            }
            if (isgen || isruns || isr16 || isr16w) {
                GenItem it;
                it.offa = A.off[ai]; it.offb = B.off[bj];
                it.out = outidx; it.ca = ca; it.cb = cb; it.types = (uint32_t)ta | ((uint32_t)tb << 8) | opbits;
                it.nra = nra; it.nrb = nrb; it.offo = offo;
                if (isgen) Q.gen[qgen + gr.rank(mgen)] = it;
                else if (isruns) Q.runs[qruns + gr.rank(mrn)] = it;
                else if (isr16) Q.runs16[qr16 + gr.rank(mr16)] = it;
                else Q.runs16w[qr16w + gr.rank(mr16w)] = it;
            }
            if (isfilt || iswave || isprobe || isusm || isba) {
                FatItem it;
                it.offa = A.off[ai]; it.offb = B.off[bj]; it.offo = offo;
                it.out = outidx; it.ca = ca; it.cb = cb; it.types = (uint32_t)ta | ((uint32_t)tb << 8) | opbits;
                it.pad0 = 0; it.pad1 = 0;
                if (X.on && (isfilt || iswave || isba)) {
                    // grouped queue (Q.filt's buffer): a slot inside the bucket of the item's X container; the decrement
                    // hands out count-1 .. 0 and leaves the histogram zero for the next batch
                    const bool x_in_a = isfilt ? !filt_y_is_a(op, ta, tb, ca, cb) : union_x_is_a(op, ta, tb, ca, cb);
                    const uint32_t xi = xg_index(X, isfilt ? XG_FILT : XG_UNION, x_in_a, ai, bj);
                    const uint32_t k = atomicSub(&X.hist[xi], 1u) - 1u;
                    Q.filt[X.hstart[xi] - X.hstart[0] + k] = it;
                } else if (isfilt) Q.filt[qfilt + gr.rank(mfl)] = it;
                else if (iswave) Q.wave[qwave + gr.rank(mwv)] = it;
                else if (isprobe) Q.probe[qprobe + gr.rank(mpr)] = it;
                else if (isba) Q.ba[qba + gr.rank(mbar)] = it;
                else Q.usmall[qusm + gr.rank(mus)] = it;
            }
            if (iscopy) {
                CopyItem it;
                it.src = A.off[ai]; it.offo = offo; it.meta = pack_meta(ta, ca, nra);
                it.out = outidx; it.n16 = (pa + 15u) >> 4;
                Q.copy[qcopy + gr.rank(mcp)] = it;
            }
            qbb += __popcll(mbb); qgen += __popcll(mgen); qcopy += __popcll(mcp); qfilt += __popcll(mfl); qwave += __popcll(mwv); qruns += __popcll(mrn); qprobe += __popcll(mpr); qbba += __popcll(mba); qusm += __popcll(mus); qr16 += __popcll(mr16); qr16w += __popcll(mr16w); qba += __popcll(mbar);
        }
    } else {
        const u64 nAt = U.implicit ? 1 : (a1 - a0 + 255) / 256;  // A-tiles of the pair in front of its B-tiles
        const u64 s0 = b0 + tile * (4 * G);
        uint32_t mbefore = (uint32_t)(starts[SEC_M * S + u] - starts[SEC_M * S + u0 + nAt]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u64 bi = s0 + G * t + lane;
            const bool act = bi < b1;
            if (__ballot(act) == 0ull) break;
            const uint32_t mt = act ? match[(size_t)u * (4 * G) + G * t + lane] : 0u;
            const bool found = (mt & MATCH_FOUND) != 0;
            const u64 fm = gr.ballot(found);
            const uint32_t mb = mbefore + gr.rank(fm);
            mbefore += (uint32_t)__popcll(fm);
            const bool emit = act && !found;
            const u64 mcp = gr.ballot(emit);
            uint32_t sl = 0, pb = 0, cb = 0, nrb = 0;
            uint8_t tb = 0;
            if (emit) {
                tb = B.type[bi]; cb = B.card[bi]; nrb = B.nruns[bi];
                pb = payload_bytes(tb, cb, nrb);
                sl = align16(pb) < 16u ? 16u : align16(pb);
            }
            const uint32_t inc = gr.incl_scan(sl);
            const u64 offo = slot_run + inc - sl;
            slot_run += __shfl(inc, gr.glast);
            if (emit) {
                const uint32_t pos = (uint32_t)(bi - b0) + (mt & ~MATCH_FOUND) - mb;
                O.key[base + pos] = B.key[bi];
                O.off[base + pos] = offo;
                O.pair[base + pos] = p;
                CopyItem it;
                it.src = B.off[bi] | COPY_FROM_B; it.offo = offo; it.meta = pack_meta(tb, cb, nrb);
                it.out = (uint32_t)(base + pos); it.n16 = (pb + 15u) >> 4;
                Q.copy[qcopy + gr.rank(mcp)] = it;
            }
            qcopy += __popcll(mcp);
        }
    }
}

// matched container pairs of a batch = the items of the class queues (the SEC_M section also counts the matched keys
// of B-tiles, which k_emit needs for ranking: under or / xor it is twice this)
__device__ __forceinline__ u64 matched_total(const u64* __restrict__ ranges) {
    const int secs[] = {SEC_BB, SEC_GEN, SEC_FILT, SEC_WAVE, SEC_RUNS, SEC_PROBE, SEC_BBA, SEC_USMALL, SEC_RUNS16, SEC_RUNS16W, SEC_BA};
    u64 m = 0;
    for (int k = 0; k < 11; ++k) m += ranges[2 * secs[k] + 1] - ranges[2 * secs[k]];
    return m;
}

// ------------------------------------------------------------------ fused tail: compaction + directory + statistics
struct DirOut {
    u64* bm_start;
    u64* key;
    uint8_t* type;
    uint32_t* card;
    uint32_t* nruns;
    u64* off;
};
// One look-back pass over the candidates: keep = non-empty result (class kernels wrote meta), prefix -> position in
// the result directory.  A thread that sees the first candidate of a result bitmap (or a gap of bitmaps with no
// candidates) writes the bitmap starts; the block holding the last candidate finishes them and the totals.
// n_cand is read from `ranges` (device), the launch is sized by the host's upper bound.
#define TAIL_READY (1ull << 63)
constexpr uint32_t TAIL_PER_THREAD = 8;
constexpr uint32_t TAIL_TILE = 256 * TAIL_PER_THREAD;  // candidates per block
// Join without events (round 4).  The class kernels of a forked batch run on up to three auxiliary streams; joining them
// into the main stream with events costs three barrier packets in front of the tail -- 20 us between the end of the
// last class kernel and the start of k_tail, on every forked batch (profiles/r04_timelines.txt).  Instead every
// auxiliary stream ends with k_join_signal and the main stream runs k_join_wait in front of k_tail: ONE wave that waits
// for the flags of the streams in `mask` (the kernel boundary behind it is the acquire the tail needs: the class
// kernels' results were released at the end of their kernels, possibly on other XCDs).  One wave holds one slot, so
// the kernels it waits for always find theirs.  (Waiting inside k_tail itself saves the 5 us of that boundary but keeps
// every block of the tail resident while it waits: a four-op batch on weather_sept_85 went 1.6 -> 2.4 ms.)
__global__ void k_join_signal(u64* flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) lb_store(flag, 1ull);
}
// (Bounded: max_spins x ~3.4 us of s_sleep, 0.2 s by default.  HIP does not promise that kernels of different streams run
// side by side, so the gate may give up: it then leaves a mark, the tail behind it runs too early, and
// rhip_pairwise_end -- which sees the mark -- waits for the auxiliary streams the ordinary way and runs the tail again:
// the batch is finished correctly either way, a time-out only costs time.  force_fail (tests) reports a time-out
// whatever the flags say.)
__global__ void k_join_wait(const u64* flags, uint32_t mask, u64* timed_out, u64 max_spins, int force_fail) {
    if (blockIdx.x == 0 && threadIdx.x < 8u && ((mask >> threadIdx.x) & 1u)) {
        u64 spins = 0;
        while (lb_load(&flags[threadIdx.x]) == 0ull) {
            __builtin_amdgcn_s_sleep(127);
            if (++spins > max_spins) { lb_store(timed_out, 1ull); break; }
        }
        if (force_fail) lb_store(timed_out, 1ull);
    }
}
// Self-test of a context (rhip_ctx_create): do kernels of two streams run SIDE BY SIDE here?  This kernel waits (bounded:
// ~5 ms) for a flag that a kernel launched later, on another stream, sets.  Under a tool that serialises kernel
// execution (rocprofv3 --pmc does) it runs alone, times out, and the context joins its streams with events instead.
__global__ void k_conc_probe(const u64* flag, u64* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u64 seen = 2;
        for (uint32_t i = 0; i < 20000u; ++i) {
            if (lb_load(flag) != 0ull) { seen = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        lb_store(out, seen);
    }
}
__global__ __launch_bounds__(256) void k_tail(const u64* __restrict__ ranges, CandOut C, const u64* __restrict__ meta,
                                              DirOut R, uint32_t n_pairs, LbState lb, u64* __restrict__ part,
                                              Stats* __restrict__ host_stats, u64* host_flag, u64 seq,
                                              const u64* join_timed_out = nullptr) {
    __shared__ u64 sm[4];
    __shared__ uint32_t s_tile;
    __shared__ u64 s_prefix;
    __shared__ u64 s_bytes[4];
    __shared__ uint32_t s_types[4][3];
    __shared__ u64 s_tot[4][3];
    __shared__ uint32_t s_cnt[4 * TAIL_PER_THREAD];
    const u64 n = ranges[2 * SEC_CAND + 1] - ranges[2 * SEC_CAND];
    if (threadIdx.x == 0) s_tile = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const u64 n_tiles = n ? (n + TAIL_TILE - 1) / TAIL_TILE : 1;
    if (tile >= n_tiles) return;
    // candidate (k, thread) of the tile = tile base + 256 k + thread: every load and store below is lane-contiguous,
    // and all the loads are issued before the first wait
    const u64 tbase = (u64)tile * TAIL_TILE + threadIdx.x;
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    u64 m[TAIL_PER_THREAD], ckey[TAIL_PER_THREAD], coff[TAIL_PER_THREAD];
    uint32_t cp[TAIL_PER_THREAD], cprev[TAIL_PER_THREAD], rank[TAIL_PER_THREAD];
#pragma unroll
    for (int k = 0; k < (int)TAIL_PER_THREAD; ++k) {
        const u64 i = tbase + 256ull * k;
        const bool in = i < n;
        m[k] = in ? meta[i] : 0;
        cp[k] = in ? C.pair[i] : 0u;
        cprev[k] = in && i ? C.pair[i - 1] + 1u : 0u;
        ckey[k] = in ? C.key[i] : 0;
        coff[k] = in ? C.off[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < (int)TAIL_PER_THREAD; ++k) {
        const u64 bal = __ballot(meta_card(m[k]) != 0);
        rank[k] = mbcnt(bal);
        if (lane == 0) s_cnt[4 * k + wv] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t v = lane < 4 * TAIL_PER_THREAD ? s_cnt[lane] : 0u;
        const uint32_t inc = wave_incl_scan(v);
        if (lane < 4 * TAIL_PER_THREAD) s_cnt[lane] = inc - v;
        const u64 tot = __shfl(inc, 63);
        const u64 pfx = lb_exclusive_prefix(lb.status, tile, tot);
        if (lane == 0) { s_prefix = pfx; sm[0] = tot; }
    }
    __syncthreads();
    const u64 total = sm[0];
    u64 bytes = 0;
    uint32_t nty[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < (int)TAIL_PER_THREAD; ++k) {
        const u64 i = tbase + 256ull * k;
        if (i < n) {
            const u64 ex = s_prefix + s_cnt[4 * k + wv] + rank[k];
            // result bitmaps starting at candidate i: every pair in (pair of candidate i-1, pair of candidate i]
            for (uint32_t q = cprev[k]; q <= cp[k]; ++q) R.bm_start[q] = ex;
            if (meta_card(m[k])) {
                const uint32_t ty = meta_type(m[k]);
                R.key[ex] = ckey[k];
                R.type[ex] = (uint8_t)ty;
                R.card[ex] = meta_card(m[k]);
                R.nruns[ex] = meta_nruns(m[k]);
                R.off[ex] = coff[k];
                bytes += payload_bytes((uint8_t)ty, meta_card(m[k]), meta_nruns(m[k]));
                nty[ty - 1]++;
            }
        }
    }
    bytes = wave_sum64(bytes);
#pragma unroll
    for (int t = 0; t < 3; ++t) nty[t] = wave_sum(nty[t]);
    if (lane_id() == 0) {
        s_bytes[threadIdx.x >> 6] = bytes;
#pragma unroll
        for (int t = 0; t < 3; ++t) s_types[threadIdx.x >> 6][t] = nty[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 b = 0;
        uint32_t ty[3] = {0, 0, 0};
        for (int w = 0; w < 4; ++w) {
            b += s_bytes[w];
            for (int t = 0; t < 3; ++t) ty[t] += s_types[w][t];
        }
        // Per-tile partial sums, published like the look-back states: the value and a "ready" bit in ONE word written
        // with a device-coherent store.  No fence and no counter: an agent-scope release here writes back the XCD's
        // whole L2 -- once per tile, in a kernel whose L2 is full of freshly stored directory lines (it doubled the
        // kernel), and same-address atomics for the sums cost 0.2 us per tile.  Each count is <= TAIL_TILE, so three
        // fit one word.
        lb_store(&part[2 * (size_t)tile], TAIL_READY | b);
        lb_store(&part[2 * (size_t)tile + 1], TAIL_READY | (u64)ty[0] | ((u64)ty[1] << 16) | ((u64)ty[2] << 32));
    }
    if (tile != n_tiles - 1) return;
    __syncthreads();  // this block's own partial sums are published before any of its threads waits for them below
    // ---- the block of the last tile closes the call.  Every other tile has started (tickets are handed out in start
    // order), so its partial sums arrive; this block adds them up and hands the totals to the host in pinned memory:
    // no copy kernel after the tail.  (The directory stores of other blocks may still be in flight then -- whatever
    // reads them is stream-ordered behind this kernel; the host itself only reads the totals.)
    const u64 kept = s_prefix + total;
    {   // bitmaps after the last candidate are empty
        const uint32_t plast = n ? C.pair[n - 1] + 1u : 0u;
        for (u64 q = (u64)plast + threadIdx.x; q <= n_pairs; q += 256) R.bm_start[q] = kept;
    }
    u64 tb = 0, t0 = 0, t1 = 0, t2 = 0;
    for (u64 t = threadIdx.x; t < n_tiles; t += 256) {
        u64 pb, pk;
        while (!((pb = lb_load(&part[2 * t])) & TAIL_READY)) {}
        while (!((pk = lb_load(&part[2 * t + 1])) & TAIL_READY)) {}
        tb += pb & ~TAIL_READY;
        t0 += pk & 0xFFFFu; t1 += (pk >> 16) & 0xFFFFu; t2 += (pk >> 32) & 0xFFFFu;
    }
    tb = wave_sum64(tb); t0 = wave_sum64(t0); t1 = wave_sum64(t1); t2 = wave_sum64(t2);
    __syncthreads();  // s_bytes of this block's own tile was consumed above
    if (lane_id() == 0) {
        s_bytes[threadIdx.x >> 6] = tb;
        s_tot[threadIdx.x >> 6][0] = t0; s_tot[threadIdx.x >> 6][1] = t1; s_tot[threadIdx.x >> 6][2] = t2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Stats st = {};
        st.result_containers = kept;
        st.n_cand = n;
        st.matched_pairs = matched_total(ranges);
        st.passthrough = ranges[2 * SEC_COPY + 1] - ranges[2 * SEC_COPY];
        st.n_bb = ranges[2 * SEC_BB + 1] - ranges[2 * SEC_BB] + ranges[2 * SEC_BBA + 1] - ranges[2 * SEC_BBA];
        st.bytes_in = ranges[2 * SEC_BYTES + 1] - ranges[2 * SEC_BYTES];
        st.slot_bytes = 16ull * (ranges[2 * SEC_SLOT + 1] - ranges[2 * SEC_SLOT]);
        for (int w = 0; w < 4; ++w) {
            st.bytes_out += s_bytes[w];
            for (int t = 0; t < 3; ++t) st.n_type[t] += s_tot[w][t];
        }
        const u64* lv = (const u64*)&st;
        u64* hv = (u64*)host_stats;
        for (uint32_t k = 0; k < sizeof(Stats) / 8; ++k) hv[k] = lv[k];
        // "the totals are there": the host polls this word instead of waiting for the stream's completion signal
        __threadfence_system();
        // (a flag join in front of this tail that gave up: the class kernels may still be writing -- the completion word
        // says so, TAIL_EARLY, and rhip_pairwise_end runs the tail again; nobody who polls the word takes this for "done")
        const u64 early = (join_timed_out && lb_load(join_timed_out) != 0ull) ? TAIL_EARLY : 0ull;
        __atomic_store_n(host_flag, seq | early, __ATOMIC_RELEASE);
    }
}
// cardinality mode has no tail: the same statistics from the section totals
__global__ void k_card_stats(const u64* __restrict__ ranges, Stats* __restrict__ host_stats) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        Stats st = {};
        st.matched_pairs = matched_total(ranges);
        st.n_bb = ranges[2 * SEC_BB + 1] - ranges[2 * SEC_BB];
        st.bytes_in = ranges[2 * SEC_BYTES + 1] - ranges[2 * SEC_BYTES];
        *host_stats = st;
    }
}

// ------------------------------------------------------------------ per-class statistics (opt-in, rhip_ctx_set_class_stats)
// One block per work class walks that class's queue after the batch has run: items, payload bytes of the operands
// (from the item) and of the results (from the candidate meta words the class kernels wrote) -- SURVEY §8d's algorithmic
// bytes, split by the kernel that moved them.  out[3 * cls + {0, 1, 2}] = items, bytes in, bytes out.
struct ClassQueues {
    const BBItem* bb; const BBItem* bba;
    const FatItem* fat[5];   // filt, wave, probe, usmall, ba
    const GenItem* gen[4];   // gen, runs, runs16, runs16w
    const CopyItem* copy;
};
__device__ __forceinline__ uint32_t meta_payload(u64 m) {
    return meta_card(m) ? payload_bytes((uint8_t)meta_type(m), meta_card(m), meta_nruns(m)) : 0u;
}
// grouped != 0: the filter class's buffer holds the X-grouped queue -- the filter items first, then the union items
// (two-array image pairs = k_wave's class, bitset (op) array = k_ba's), told apart by their types.
__global__ __launch_bounds__(256) void k_class_stats(const u64* __restrict__ ranges, ClassQueues Q,
                                                     const u64* __restrict__ meta, u64* __restrict__ out, int grouped) {
    __shared__ u64 sb[4][3];
    const int secs[N_CLS] = {SEC_BB, SEC_GEN, SEC_COPY, -1, SEC_FILT, SEC_WAVE, SEC_RUNS, SEC_PROBE, SEC_BBA, SEC_USMALL,
                             SEC_RUNS16, SEC_RUNS16W, SEC_BA};
    const int cls = blockIdx.x;
    const int sec = secs[cls];
    u64 n = 0, bin = 0, bout = 0, nx = 0;
    const bool xcls = grouped && (cls == CLS_FILT || cls == CLS_WAVE || cls == CLS_BA);
    if (xcls) {
        const u64* xr = ranges + 2 * N_SEC;
        const u64 lo = cls == CLS_FILT ? 0 : xr[1] - xr[0], hi = (cls == CLS_FILT ? xr[1] : xr[2]) - xr[0];
        for (u64 i = lo + threadIdx.x; i < hi; i += 256) {
            const FatItem& t = Q.fat[0][i];
            const bool a_bitset = (t.types & 0xFF) == T_BITSET || ((t.types >> 8) & 0xFF) == T_BITSET;
            if (cls == CLS_WAVE && a_bitset) continue;
            if (cls == CLS_BA && !a_bitset) continue;
            ++nx;
            bin += payload_bytes((uint8_t)(t.types & 0xFF), t.ca, 0) + payload_bytes((uint8_t)(t.types >> 8), t.cb, 0);
            bout += meta_payload(meta[t.out]);
        }
    } else if (sec >= 0) {
        n = ranges[2 * sec + 1] - ranges[2 * sec];
        for (u64 i = threadIdx.x; i < n; i += 256) {
            uint32_t o = 0;
            if (cls == CLS_BB || cls == CLS_BBA) {
                const BBItem& t = (cls == CLS_BB ? Q.bb : Q.bba)[i];
                bin += 16384u; o = t.out;
            } else if (cls == CLS_COPY) {
                const CopyItem& t = Q.copy[i];
                bin += meta_payload(t.meta); o = t.out;
            } else if (cls == CLS_GEN || cls == CLS_RUNS || cls == CLS_RUNS16 || cls == CLS_RUNS16W) {
                const GenItem& t = Q.gen[cls == CLS_GEN ? 0 : cls == CLS_RUNS ? 1 : cls == CLS_RUNS16 ? 2 : 3][i];
                bin += payload_bytes((uint8_t)(t.types & 0xFF), t.ca, t.nra) + payload_bytes((uint8_t)(t.types >> 8), t.cb, t.nrb);
                o = t.out;
            } else {
                const FatItem& t = Q.fat[cls == CLS_FILT ? 0 : cls == CLS_WAVE ? 1 : cls == CLS_PROBE ? 2 : cls == CLS_USMALL ? 3 : 4][i];
                bin += payload_bytes((uint8_t)(t.types & 0xFF), t.ca, 0) + payload_bytes((uint8_t)(t.types >> 8), t.cb, 0);
                o = t.out;
            }
            bout += meta_payload(meta[o]);
        }
    }
    bin = wave_sum64(bin); bout = wave_sum64(bout); nx = wave_sum64(nx);
    if (lane_id() == 0) { sb[threadIdx.x >> 6][0] = bin; sb[threadIdx.x >> 6][1] = bout; sb[threadIdx.x >> 6][2] = nx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (xcls) n = sb[0][2] + sb[1][2] + sb[2][2] + sb[3][2];
        out[3 * cls] = n;
        out[3 * cls + 1] = sb[0][0] + sb[1][0] + sb[2][0] + sb[3][0];
        out[3 * cls + 2] = sb[0][1] + sb[1][1] + sb[2][1] + sb[3][1];
    }
}

// ------------------------------------------------------------------ directory compaction (flip / many-way paths)
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
__global__ void k_flags(const u64* __restrict__ meta, u64 n, uint32_t* __restrict__ flag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = meta_card(meta[i]) ? 1u : 0u;
}
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

// ------------------------------------------------------------------ per-bitmap host mirrors
// wave per bitmap: cardinality (roaring.c:1436-1443), the payload bound W used to size result arenas without a
// device round trip, and which container types occur
//   W = sum over containers of max(align16(payload), min(8192, align16(2 * card)))
// (an AND / ANDNOT result slot of a matched pair is <= W(a-side container); an OR / XOR slot <= the sum of both).
__global__ __launch_bounds__(256) void k_bitmap_cards(PoolView P, uint32_t nbm, u64* __restrict__ out) {
    uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= nbm) return;
    u64 s = 0;
    for (u64 i = P.bm_start[b] + lane_id(); i < P.bm_start[b + 1]; i += 64) s += P.card[i];
    s = wave_sum64(s);
    if (lane_id() == 0) out[b] = s;
}
__device__ __forceinline__ uint32_t slot_bound(uint8_t type, uint32_t card, uint32_t nruns) {
    const uint32_t p = align16(payload_bytes(type, card, nruns));
    uint32_t c = align16(2u * card);
    c = c > 8192u ? 8192u : c;
    const uint32_t w = p > c ? p : c;
    return w < 16u ? 16u : w;
}
// wmany: the same bound for the many-way path, whose member descriptors (rhip_many.h) carry a run container's
// cardinality rounded up to a multiple of 256 -- in bits 0 .. WM_RUNS_SHIFT - 1 (a bitmap's bound is below 2^34); the
// bits above count the bitmap's RUN containers (the host's "interval work dominates this pool" census)
constexpr int WM_RUNS_SHIFT = 44;
__global__ __launch_bounds__(256) void k_bitmap_bounds(PoolView P, uint32_t nbm, u64* __restrict__ wout,
                                                       u64* __restrict__ wmany,
                                                       uint32_t* __restrict__ census /* [4] bitset, array, run, some payload above 8192 bytes */,
                                                       u64* __restrict__ maxkey) {
    uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= nbm) return;
    u64 s = 0, sm = 0;
    uint32_t seen = 0;
    for (u64 i = P.bm_start[b] + lane_id(); i < P.bm_start[b + 1]; i += 64) {
        const uint8_t t = P.type[i];
        const uint32_t cd = P.card[i], nr = P.nruns[i];
        s += slot_bound(t, cd, nr);
        sm += slot_bound(t, t == T_RUN ? ((cd + 255u) & ~255u) : cd, nr);
        sm += t == T_RUN ? (1ull << WM_RUNS_SHIFT) : 0ull;  // (the bitmap's run containers, counted above the bound's bits)
        seen |= 1u << (t - 1);
        if (payload_bytes(t, cd, nr) > 8192u) seen |= 8u;  // a run list longer than a bitset (valid; run_optimize never leaves one)
    }
    s = wave_sum64(s);
    sm = wave_sum64(sm);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) seen |= __shfl_xor(seen, o);
    if (lane_id() == 0) {
        wout[b] = s;
        wmany[b] = sm;
        if (P.bm_start[b + 1] > P.bm_start[b]) {  // keys ascend inside a bitmap: its last key is its largest
            const u64 k = P.key[P.bm_start[b + 1] - 1];
            if (k > __atomic_load_n(maxkey, __ATOMIC_RELAXED)) atomicMax(maxkey, k);  // (100 000 same-address atomics cost 4 ms)
        }
        for (int t = 0; t < 4; ++t)
            if ((seen >> t) & 1u) census[t] = 1u;  // benign race: every writer stores the same value
    }
}
// distinct 16-bit keys of a 32-bit pool: mark (bits = 2048 zeroed words), then count
__global__ void k_key_mark(const u64* __restrict__ key, u64 n, uint32_t* __restrict__ bits) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)key[i] & 0xFFFFu, bit = 1u << (k & 31u);
        if (!(bits[k >> 5] & bit)) atomicOr(&bits[k >> 5], bit);
    }
}
__global__ __launch_bounds__(256) void k_key_count(const uint32_t* __restrict__ bits, u64* __restrict__ out) {
    __shared__ uint32_t sw[4];
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < 2048u; i += 256u) s += (uint32_t)__popc(bits[i]);
    s = wave_sum(s);
    if (lane_id() == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (u64)sw[0] + sw[1] + sw[2] + sw[3];
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
