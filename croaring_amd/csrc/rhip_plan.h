// rhip_plan.h -- planning (key merge -> typed work items) and result-directory compaction kernels
#pragma once
#include "rhip_common.h"

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
// Count arrays (and their exclusive scan) have N_SEC sections of n_units+1 entries:
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
