// rhip_deser.h -- device-side portable deserialization (input half): n portable images packed
// in ONE blob are uploaded with one copy; headers are parsed, every container is validated and its payload is
// moved into the 16-byte aligned arena by kernels.  Accepts exactly what the host loader in rhip_engine.hip
// (parse_portable32) accepts, i.e. ra_portable_deserialize (src/roaring_array.c:633-813) followed by
// roaring_bitmap_internal_validate (src/roaring.c:454-523); 64-bit images follow
// roaring64_bitmap_portable_deserialize_safe (src/roaring64.c:2442-2535): u64 bucket count, then per bucket a
// u32 high key and a 32-bit image.
#pragma once
#include "rhip_common.h"

__device__ __forceinline__ uint32_t ld_le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ u64 ld_le64(const uint8_t* p) { return (u64)ld_le32(p) | ((u64)ld_le32(p + 4) << 32); }

struct DesOut {   // directory under construction (fill pass only)
    u64* key;
    uint8_t* type;
    uint32_t* card;
    uint32_t* nruns;
    u64* src;        // absolute blob offset of the payload (after the n_runs prefix of run containers)
    uint32_t* slot;  // payload bytes rounded up to the pool's slot granule (amask + 1 = 16 or 128)
    uint32_t amask;
};

// Walk one 32-bit image at blob[p, end): returns its size in bytes, 0 if malformed.  Executed by a whole wave
// (all arguments wave-uniform).  FILL = false only counts (ncont += n); FILL = true also writes the directory
// entries [cbase, cbase + n).  The position of container j is the prefix sum of the sizes before it, and the
// size of a run container is only known once its n_runs prefix has been read at that position: the offset
// header, when the format has one, is used as a HINT that lets all 64 lanes read their n_runs at once, and is
// then verified against the computed positions (the reference ignores the offset header entirely, so a wrong
// hint must not change the outcome: the chunk is then redone one run container at a time).
template <bool FILL>
__device__ u64 des_walk32(const uint8_t* __restrict__ blob, u64 p, u64 end, u64 key_hi, u64 cbase, DesOut D,
                          uint32_t& ncont) {
    const uint32_t lane = lane_id();
    if (end - p < 4) return 0;
    const uint32_t cookie = ld_le32(blob + p);
    u64 q = p + 4;
    uint32_t n;
    bool hasrun = false;
    u64 flags = 0;
    if ((cookie & 0xFFFFu) == 12347u) {
        hasrun = true;
        n = (cookie >> 16) + 1u;
        const u64 nb = (n + 7u) >> 3;
        if (end - q < nb) return 0;
        flags = q;
        q += nb;
    } else if (cookie == 12346u) {
        if (end - q < 4) return 0;
        n = ld_le32(blob + q);
        q += 4;
    } else {
        return 0;
    }
    if (n > 65536u) return 0;
    if (end - q < 4ull * n) return 0;
    const u64 desc = q;
    q += 4ull * n;
    const bool with_offsets = !hasrun || n >= 4u;
    u64 offh = 0;
    if (with_offsets) {
        if (end - q < 4ull * n) return 0;
        offh = q;
        q += 4ull * n;
    }
    u64 run_base = q;
    uint32_t prev_key = 0;  // key of the last container of the previous chunk (valid when j0 > 0)
    for (uint32_t j0 = 0; j0 < n; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool live = j < n;
        uint32_t k16 = 0, card = 0, sz = 0, nr = 0;
        bool isrun = false;
        if (live) {
            k16 = ld_le16(blob + desc + 4ull * j);
            card = ld_le16(blob + desc + 4ull * j + 2) + 1u;
            isrun = hasrun && ((blob[flags + (j >> 3)] >> (j & 7u)) & 1u);
            if (!isrun) sz = card > 4096u ? 8192u : 2u * card;
        }
        // keys strictly increasing (ra_portable_deserialize does not check; internal_validate does)
        uint32_t pk = __shfl_up(k16, 1);
        if (lane == 0) pk = prev_key;
        const bool key_bad = live && (j > 0) && k16 <= pk;
        if (__ballot(key_bad)) return 0;
        prev_key = __shfl(k16, 63);
        const u64 runmask = __ballot(isrun);
        bool hinted = false;
        if (runmask && with_offsets) {
            // parallel attempt: n_runs read at the hinted positions
            u64 hint = 0;
            bool hint_ok = true;
            if (isrun) {
                hint = p + ld_le32(blob + offh + 4ull * j);
                hint_ok = hint >= q && hint <= end && end - hint >= 2;
                if (hint_ok) {
                    nr = ld_le16(blob + hint);
                    sz = 2u + 4u * nr;
                }
            }
            const uint32_t inc = wave_incl_scan(sz);
            const u64 pos = run_base + inc - sz;
            hinted = __ballot(isrun && (!hint_ok || pos != hint)) == 0;
        }
        if (runmask && !hinted) {
            // sequential: resolve the run containers of this chunk in order
            if (isrun) sz = 0;
            u64 m = runmask;
            while (m) {
                const uint32_t r = (uint32_t)__ffsll((long long)m) - 1u;
                m &= m - 1;
                const uint32_t inc = wave_incl_scan(sz);
                const uint32_t before = __shfl(inc - sz, (int)r);
                const u64 pos = run_base + before;
                if (pos > end || end - pos < 2) return 0;
                const uint32_t v = ld_le16(blob + pos);
                if (lane == r) {
                    nr = v;
                    sz = 2u + 4u * v;
                }
            }
        }
        if (__ballot(isrun && nr == 0u)) return 0;  // a run container holds at least one run
        const uint32_t inc = wave_incl_scan(sz);
        const u64 pos = run_base + inc - sz;
        const bool oob = live && (pos > end || end - pos < sz);
        if (__ballot(oob)) return 0;
        if (FILL && live) {
            const u64 c = cbase + j;
            D.key[c] = (key_hi << 16) | k16;
            D.type[c] = isrun ? T_RUN : (card > 4096u ? T_BITSET : T_ARRAY);
            D.card[c] = isrun ? 0u : card;  // run cardinalities come from the payload pass
            D.nruns[c] = nr;
            D.src[c] = isrun ? pos + 2 : pos;
            D.slot[c] = ((isrun ? 4u * nr : sz) + D.amask) & ~D.amask;
        }
        run_base += __shfl(inc, 63);
    }
    ncont += n;
    return run_base - p;
}

// One WAVE per bitmap.  FILL = false: ncont[i] = number of containers, status = first malformed bitmap.
// FILL = true: directory entries from bm_start[i] on.
template <bool FILL>
__global__ __launch_bounds__(256) void k_des_walk(const uint8_t* __restrict__ blob, const u64* __restrict__ offs,
                                                  const u64* __restrict__ lens, uint32_t n_bitmaps, int is64,
                                                  const u64* __restrict__ bm_start, DesOut D,
                                                  uint32_t* __restrict__ ncont, uint32_t* status) {
    const uint32_t lane = lane_id();
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_bitmaps) return;  // wave-uniform
    u64 p = offs[i];
    const u64 end = p + lens[i];
    uint32_t cnt = 0;
    bool ok = true;
    const u64 cbase = FILL ? bm_start[i] : 0;
    if (!is64) {
        ok = des_walk32<FILL>(blob, p, end, 0, cbase, D, cnt) != 0;
    } else if (end - p < 8) {
        ok = false;
    } else {
        const u64 nb = ld_le64(blob + p);
        p += 8;
        u64 prev = 0;
        for (u64 b = 0; b < nb && ok; ++b) {
            if (end - p < 4) { ok = false; break; }
            const u64 high = ld_le32(blob + p);
            p += 4;
            if (b > 0 && high <= prev) { ok = false; break; }  // buckets strictly ascending
            prev = high;
            const u64 used = des_walk32<FILL>(blob, p, end, high, cbase + cnt, D, cnt);
            if (!used) { ok = false; break; }
            p += used;
        }
    }
    if (lane == 0) {
        if (!FILL) ncont[i] = ok ? cnt : 0u;
        if (!ok) atomicMin(status, i);
    }
}

// One WAVE per container: the payload moves from its (arbitrarily aligned) place in the blob to its 16-byte
// aligned slot, one dword per lane per step through registers, and is validated on the way as
// roaring_bitmap_internal_validate would (bitset.c:1023-1044, array.c:456-493, run.c:669-716):
//   bitset: popcount == cardinality of the descriptive header (and > 4096 by construction)
//   array : strictly increasing
//   run   : value + length <= 65535, runs sorted, disjoint and NOT adjacent; cardinality = sum(length + 1)
__global__ __launch_bounds__(256) void k_des_payload(const uint8_t* __restrict__ blob, DesOut D,
                                                     const u64* __restrict__ off, uint8_t* __restrict__ arena,
                                                     u64 n_cont, const u64* __restrict__ bm_start,
                                                     uint32_t n_bitmaps, uint32_t* status) {
    const uint32_t lane = lane_id();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 c = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < n_cont; c += nwaves) {
        const uint32_t t = D.type[c], card = D.card[c], nr = D.nruns[c];
        const uint32_t B = t == T_BITSET ? 8192u : (t == T_ARRAY ? 2u * card : 4u * nr);
        const u64 sp = D.src[c];
        const uint32_t d = (uint32_t)(sp & 3u);
        const uint32_t* __restrict__ S = (const uint32_t*)(blob + (sp - d));  // aligned dwords covering the payload
        uint32_t* __restrict__ O = (uint32_t*)(arena + off[c]);
        const uint32_t nd = (B + 3u) >> 2;  // output dwords (the last may be half used: odd array cardinality)
        uint32_t acc = 0;       // bitset: popcount; run: cardinality
        uint32_t carry = 0;     // last value (array) / last run end (run) of the previous step
        bool bad = false;
        for (uint32_t k0 = 0; k0 < nd; k0 += 64) {
            const uint32_t k = k0 + lane;
            uint32_t w = 0;
            if (k < nd) {
                const u64 two = ((u64)S[k + 1] << 32) | (u64)S[k];
                w = (uint32_t)(two >> (8u * d));
                if (4u * k + 4u > B) w &= 0xFFFFu;  // the upper half lies beyond the payload
                O[k] = w;
            }
            if (t == T_BITSET) {
                acc += __popc(w);
            } else if (t == T_ARRAY) {
                const uint32_t v0 = w & 0xFFFFu, v1 = w >> 16;
                const bool has0 = 2u * k < card, has1 = 2u * k + 1u < card;
                uint32_t pv = __shfl_up(v1, 1);
                if (lane == 0) pv = carry;
                if (has0 && k > 0 && v0 <= pv) bad = true;
                if (has1 && v1 <= v0) bad = true;
                carry = __shfl(v1, 63);
            } else {
                const uint32_t s = w & 0xFFFFu, l = w >> 16, e = s + l;
                const bool has = k < nr;
                uint32_t pe = __shfl_up(e, 1);
                if (lane == 0) pe = carry;
                if (has && e > 65535u) bad = true;
                if (has && k > 0 && s <= pe + 1u) bad = true;
                if (has) acc += l + 1u;
                carry = __shfl(e, 63);
            }
        }
        const uint32_t total = wave_sum(acc);
        if (t == T_BITSET && total != card) bad = true;
        const bool anybad = __ballot(bad) != 0;
        if (lane == 0) {
            if (t == T_RUN) D.card[c] = total;
            if (anybad) {
                // bitmap of container c: last i with bm_start[i] <= c
                u64 lo = 0, hi = n_bitmaps;
                while (lo + 1 < hi) {
                    const u64 mid = (lo + hi) >> 1;
                    if (bm_start[mid] <= c) lo = mid;
                    else hi = mid;
                }
                atomicMin(status, (uint32_t)lo);
            }
        }
    }
}
