// rhip_heap.h -- roaring_bitmap_or_many_heap (src/roaring_priority_queue.c:200-247) with the reference's container TYPES.
//
// The reference's second many-way union is a tournament: a binary min-heap keyed by roaring_bitmap_portable_size_in_bytes,
// the two smallest elements are lazily OR-ed and the result pushed back, until one bitmap is left; one repair pass.  Its
// lazy unions never convert to bitsets early (bitsetconversion = false): two arrays stay an array up to 1024 values in all,
// run | array stays a raw run, run | run is typed by size at every step, a bitset absorbs whatever it meets, full runs
// short-circuit.  So the TYPE of every result container depends on which partial unions met in which order -- and the
// order depends on the serialized size of every intermediate.  The values are roaring_bitmap_or_many's (rhip_or_many is
// the fast path for them); what this file adds is the tournament itself, for callers that need the heap's bytes:
//   * the heap lives on the host, restated move for move (ties are broken by heap position);
//   * an element is an input bitmap of the pool, or a TEMPORARY: one 8-byte record per key of the key space (type as the
//     reference holds it at that moment -- possibly a lazy bitset of unknown cardinality, a raw run --, cardinality, run
//     count, where the values are: a container of the pool, or an 8 KiB image in the step scratch);
//   * one step = k_heap_keys (a thread per key: unmatched containers move, matched keys are queued) + k_heap_merge (a
//     workgroup per matched key: OR into an LDS image, cardinality and run count of the union, the reference's typing rule
//     for this kind of step) + k_heap_publish (the new element's serialized size to the host, which needs it for the
//     next poll): n - 1 dependent steps of ~three launches.  Exact, not fast: ~40 us per step plus the unions themselves.
#pragma once
#include "rhip_many.h"

// ------------------------------------------------------------------ element records
// bits 0..1 type (0: the element has no container under this key; T_BITSET / T_ARRAY / T_RUN), bit 2 cardinality known
// (a lazy bitset's is not), bit 3 the values are container `ref` of the pool (else image `ref`), bits 4..20 cardinality,
// bits 21..37 run count, bits 38..63 ref
__device__ __forceinline__ u64 hr_pack(uint32_t ty, bool known, bool inref, uint32_t card, uint32_t nruns, u64 ref) {
    return (u64)ty | ((u64)(known ? 1u : 0u) << 2) | ((u64)(inref ? 1u : 0u) << 3) | ((u64)card << 4) | ((u64)nruns << 21) | (ref << 38);
}
__device__ __forceinline__ uint32_t hr_type(u64 r) { return (uint32_t)r & 3u; }
__device__ __forceinline__ bool hr_known(u64 r) { return ((r >> 2) & 1ull) != 0; }
__device__ __forceinline__ bool hr_inref(u64 r) { return ((r >> 3) & 1ull) != 0; }
__device__ __forceinline__ uint32_t hr_card(u64 r) { return (uint32_t)(r >> 4) & 0x1FFFFu; }
__device__ __forceinline__ uint32_t hr_nruns(u64 r) { return (uint32_t)(r >> 21) & 0x1FFFFu; }
__device__ __forceinline__ u64 hr_ref(u64 r) { return r >> 38; }
constexpr u64 HR_MAX_REF = (1ull << 26) - 1ull;
// container_is_full, containers.h:262-277 (a lazy bitset of unknown cardinality is not "full")
__device__ __forceinline__ bool hr_full(u64 r) {
    const uint32_t ty = hr_type(r);
    return ty == T_RUN ? (hr_nruns(r) == 1u && hr_card(r) == 65536u) : (hr_known(r) && hr_card(r) == 65536u);
}
// container_size_in_bytes (bitset.h / array.h / run.h serialized sizes): what roaring_bitmap_portable_size_in_bytes adds up
__device__ __forceinline__ uint32_t hr_size(u64 r) {
    const uint32_t ty = hr_type(r);
    return ty == T_BITSET ? 8192u : (ty == T_ARRAY ? 2u * hr_card(r) : (ty == T_RUN ? 2u + 4u * hr_nruns(r) : 0u));
}

struct HeapElem {     // one operand of a step
    const u64* col;   // a temporary's records [KS], or null: bitmap [lo, hi) of the pool
    u64 lo, hi;
};
struct HeapTotals { u64 size, count, has_run, n_work, img_next, err; };

__device__ __forceinline__ u64 heap_rec(const PoolView& P, const HeapElem& E, uint32_t k) {
    if (E.col) return E.col[k];
    const u64 j = lower_bound(P.key, E.lo, E.hi, (u64)k);
    if (j >= E.hi || P.key[j] != (u64)k) return 0ull;
    return hr_pack(P.type[j], true, true, P.card[j], P.nruns[j], j);
}

// a thread per key: records of the two operands; one present -> it moves into the new element; both -> queued for k_heap_merge
__global__ __launch_bounds__(256) void k_heap_keys(PoolView P, HeapElem X1, HeapElem X2, u64* __restrict__ dst, uint32_t KS,
                                                   uint32_t* __restrict__ wk, u64* __restrict__ w1, u64* __restrict__ w2,
                                                   HeapTotals* __restrict__ tot) {
    __shared__ u64 s_sz[4];
    __shared__ uint32_t s_cnt[4], s_run[4];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    u64 sz = 0;
    uint32_t cnt = 0, hasrun = 0;
    if (k < KS) {
        const u64 r1 = heap_rec(P, X1, k), r2 = heap_rec(P, X2, k);
        if (r1 && r2) {
            const uint32_t i = (uint32_t)atomicAdd((unsigned long long*)&tot->n_work, 1ull);
            wk[i] = k; w1[i] = r1; w2[i] = r2;
        } else {
            const u64 r = r1 ? r1 : r2;
            dst[k] = r;
            if (r) { sz = hr_size(r); cnt = 1; hasrun = hr_type(r) == T_RUN ? 1u : 0u; }
        }
    }
    sz = wave_sum64(sz); cnt = wave_sum(cnt); hasrun = wave_sum(hasrun);
    if (lane_id() == 0) { s_sz[threadIdx.x >> 6] = sz; s_cnt[threadIdx.x >> 6] = cnt; s_run[threadIdx.x >> 6] = hasrun; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 a = s_sz[0] + s_sz[1] + s_sz[2] + s_sz[3];
        const uint32_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], h = s_run[0] + s_run[1] + s_run[2] + s_run[3];
        if (c) { atomicAdd((unsigned long long*)&tot->size, (unsigned long long)a); atomicAdd((unsigned long long*)&tot->count, (unsigned long long)c); }
        if (h) atomicOr((unsigned long long*)&tot->has_run, 1ull);
    }
}

// the values of one operand record into / onto the LDS image `acc` (first: the image is overwritten; else OR-ed)
__device__ __forceinline__ void heap_load(uint32_t* acc, uint32_t* tmp, const PoolView& P, const u64* __restrict__ img, u64 r,
                                          bool first, BlockScratch* sc) {
    const uint32_t tid = threadIdx.x;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    bool words = false;
    if (!hr_inref(r)) {
        const uint4* __restrict__ g = (const uint4*)(img + hr_ref(r) * 1024ull);
        a = g[2 * tid]; b = g[2 * tid + 1];
        words = true;
    } else {
        const u64 j = hr_ref(r);
        const uint32_t ty = P.type[j];
        if (ty == T_BITSET) {
            const uint4* __restrict__ g = (const uint4*)(P.arena + P.off[j]);
            a = g[2 * tid]; b = g[2 * tid + 1];
            words = true;
        } else if (ty == T_RUN) {
            many_raster_runs(tmp, P.arena, md_pack(P.off[j], T_RUN, P.card[j], P.nruns[j]), sc);  // (ends with a barrier)
            a = ((const uint4*)tmp)[2 * tid]; b = ((const uint4*)tmp)[2 * tid + 1];
            words = true;
        }
    }
    if (first) {
        ((uint4*)acc)[2 * tid] = a; ((uint4*)acc)[2 * tid + 1] = b;  // (an array operand: zeros, its values follow)
    } else if (words) {
        uint4 r0 = ((uint4*)acc)[2 * tid], r1 = ((uint4*)acc)[2 * tid + 1];
        r0 = op4(OP_OR, r0, a); r1 = op4(OP_OR, r1, b);
        ((uint4*)acc)[2 * tid] = r0; ((uint4*)acc)[2 * tid + 1] = r1;
    }
    __syncthreads();
    if (hr_inref(r) && P.type[hr_ref(r)] == T_ARRAY) {
        const u64 j = hr_ref(r);
        const uint32_t n = P.card[j];
        const uint16_t* __restrict__ v = (const uint16_t*)(P.arena + P.off[j]);
        for (uint32_t i = tid; i < n; i += 256) atomicOr(&acc[v[i] >> 5], 1u << (v[i] & 31u));
        __syncthreads();
    }
}

// convert_run_to_efficient_container's outcome as a record (a bitset made from a run knows its cardinality)
__device__ __forceinline__ u64 hr_by_size(uint32_t card, uint32_t nruns, u64 ref) {
    const int ty = type_eff(card, nruns);
    return hr_pack((uint32_t)ty, true, false, card, ty == T_RUN ? nruns : 0u, ref);
}
// The type the reference leaves under a key both operands hold.  `inplace`: container_lazy_ior (containers.h:1333-1442),
// else container_lazy_or (:1113-1215) -- they differ for two bitsets only (lazy_ior counts, LAZY_OR_BITSET_CONVERSION_TO_FULL,
// and a full result becomes a full run).  card / nruns: of the union.
__device__ __forceinline__ u64 heap_rule(u64 a, u64 b, bool inplace, uint32_t card, uint32_t nruns, u64 ref) {
    const uint32_t ta = hr_type(a), tb = hr_type(b);
    const u64 lazy_bitset = hr_pack(T_BITSET, false, false, 0u, 0u, ref);
    if (ta == T_BITSET && tb == T_BITSET) {
        if (!inplace) return lazy_bitset;
        return card == 65536u ? hr_pack(T_RUN, true, false, 65536u, 1u, ref) : hr_pack(T_BITSET, true, false, card, 0u, ref);
    }
    if (ta == T_ARRAY && tb == T_ARRAY)  // array_array_container_lazy(_inplace)_union, mixed_union.c:247-365
        return hr_card(a) + hr_card(b) <= 1024u ? hr_pack(T_ARRAY, true, false, card, 0u, ref) : lazy_bitset;
    if (ta == T_RUN && tb == T_RUN) return hr_by_size(card, nruns, ref);
    if (ta == T_BITSET || tb == T_BITSET) {  // bitset with array: lazy; with a run: the FULL run wins, else lazy
        const u64 o = ta == T_BITSET ? b : a;
        if (hr_type(o) == T_RUN && hr_full(o)) return hr_pack(T_RUN, true, false, 65536u, 1u, ref);
        return lazy_bitset;
    }
    return hr_pack(T_RUN, true, false, card, nruns, ref);  // array | run: array_run_container_union, a raw run ("we are lazy")
}

// a workgroup per matched key.  mode 0: roaring_bitmap_lazy_or(x1, x2, false); 1: roaring_bitmap_lazy_or_inplace(x1 = the
// temporary, x2) -- a full accumulator container is left alone (roaring.c:2621); 3: lazy_or_from_lazy_inputs(x1, x2) -- a
// bitset second operand goes first (roaring_priority_queue.c:135-147)
__global__ __launch_bounds__(256) void k_heap_merge(PoolView P, int mode, const uint32_t* __restrict__ wk, const u64* __restrict__ w1,
                                                    const u64* __restrict__ w2, u64* __restrict__ img, u64 img_cap,
                                                    u64* __restrict__ dst, HeapTotals* __restrict__ tot) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint32_t tmp[2048];
    __shared__ BlockScratch sc;
    __shared__ u64 s_slot;
    const uint32_t nw = (uint32_t)tot->n_work, tid = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < nw; i += gridDim.x) {
        u64 a = w1[i], b = w2[i];
        const uint32_t k = wk[i];
        u64 res;
        if (mode == 1 && hr_full(a)) {
            res = a;  // (the accumulator's container stays as it is, values and type)
        } else {
            if (mode == 3 && hr_type(b) == T_BITSET && hr_type(a) != T_BITSET) { const u64 t = a; a = b; b = t; }
            // the union's image goes where an operand's image already is, else into a new one
            __syncthreads();
            if (tid == 0) {
                u64 slot = !hr_inref(a) ? hr_ref(a) : (!hr_inref(b) ? hr_ref(b) : (u64)atomicAdd((unsigned long long*)&tot->img_next, 1ull));
                if (slot >= img_cap) { atomicOr((unsigned long long*)&tot->err, 1ull); slot = 0; }
                s_slot = slot;
            }
            __syncthreads();
            const u64 slot = s_slot;
            heap_load(acc, tmp, P, img, a, true, &sc);
            heap_load(acc, tmp, P, img, b, false, &sc);
            uint32_t card, nruns;
            many_image_stats(acc, &sc, &card, &nruns);
            uint4* __restrict__ po = (uint4*)(img + slot * 1024ull);
            po[2 * tid] = ((const uint4*)acc)[2 * tid];
            po[2 * tid + 1] = ((const uint4*)acc)[2 * tid + 1];
            res = heap_rule(a, b, mode != 0, card, nruns, slot);
        }
        if (tid == 0) {
            dst[k] = res;
            atomicAdd((unsigned long long*)&tot->size, (unsigned long long)hr_size(res));
            atomicAdd((unsigned long long*)&tot->count, 1ull);
            if (hr_type(res) == T_RUN) atomicOr((unsigned long long*)&tot->has_run, 1ull);
        }
    }
}
// the new element's size figures to the host (pinned), the step's counters back to zero (img_next and err stay)
__global__ void k_heap_publish(HeapTotals* __restrict__ tot, u64* __restrict__ host, u64* host_flag, u64 seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        host[0] = tot->size; host[1] = tot->count; host[2] = tot->has_run; host[3] = tot->err;
        tot->size = 0; tot->count = 0; tot->has_run = 0; tot->n_work = 0;
        __threadfence_system();
        __atomic_store_n(host_flag, seq, __ATOMIC_RELEASE);
    }
}

// roaring_bitmap_repair_after_lazy (roaring.c:2845-2856, containers.h:344-371) and the result: a bitset is counted and typed
// by cardinality, an array stays, a run goes through convert_run_to_efficient_container.  Slot k x 8192 of the arena.
__global__ __launch_bounds__(256) void k_heap_emit(PoolView P, const u64* __restrict__ col, uint32_t KS, const u64* __restrict__ img,
                                                   OutView O) {
    __shared__ __attribute__((aligned(16))) uint32_t acc[2048];
    __shared__ __attribute__((aligned(16))) uint32_t tmp[2048 + 8];
    __shared__ BlockScratch sc;
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = blockIdx.x; k < KS; k += gridDim.x) {
        const u64 r = col[k];
        if (tid == 0) O.key[k] = k;
        if (!r) {
            if (tid == 0) O.meta[k] = pack_meta(T_ARRAY, 0u, 0u);
            continue;
        }
        __syncthreads();
        heap_load(acc, tmp, P, img, r, true, &sc);
        uint32_t card, nruns;
        many_image_stats(acc, &sc, &card, &nruns);
        const uint32_t sty = hr_type(r);
        const int ty = sty == T_RUN ? type_eff(card, nruns) : (sty == T_ARRAY ? T_ARRAY : type_ba(card));
        const uint4 r0 = ((const uint4*)acc)[2 * tid], r1 = ((const uint4*)acc)[2 * tid + 1];
        const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        __syncthreads();
        lds_emit(acc, w, ty, card, nruns, (uint16_t*)tmp, O.arena + (u64)k * 8192ull, &sc);
        if (tid == 0) O.meta[k] = pack_meta((uint32_t)ty, card, ty == T_RUN ? nruns : 0u);
        __syncthreads();
    }
}
