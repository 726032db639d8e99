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

// One WAVE per container: the payload moves from its (arbitrarily aligned) place in the blob to its 16-byte aligned slot,
// SIXTEEN bytes per lane and step through registers, and is validated on the way as roaring_bitmap_internal_validate would
// (bitset.c:1023-1044, array.c:456-493, run.c:669-716):
//   bitset: popcount == cardinality of the descriptive header (and > 4096 by construction)
//   array : strictly increasing
//   run   : value + length <= 65535, runs sorted, disjoint and NOT adjacent; cardinality = sum(length + 1)
// Round 6: until then a lane moved ONE dword per step behind ~80 wave-level instructions (clamps, a 64-bit funnel shift, the
// order test with two shuffles) -- 1.5 ms for C4's 1.64 GB, and the ablations (no stores 1.25 ms, no validation 1.20,
// directory only 0.17; more loads in flight: nothing) said instruction issue, not memory: a wave64 instruction holds its
// SIMD for four cycles.  Now a lane loads the two aligned 16-byte groups that cover its output group, v_alignbyte funnels
// them by the container's (wave-uniform) misalignment, and the tests run on eight values / four runs at a time: a third of
// the instructions per byte.  Bytes of the last group past the payload are written as ZERO (the slot is padded to 16).
__device__ __forceinline__ uint32_t des_align(uint32_t hi, uint32_t lo, uint32_t bytes) {
#ifdef RHIP_EMU
    return bytes ? (lo >> (8u * bytes)) | (hi << (32u - 8u * bytes)) : lo;
#else
    return __builtin_amdgcn_alignbyte(hi, lo, bytes);
#endif
}
// (The directory arrays come as `const __restrict__` kernel arguments of their own -- not inside DesOut -- so that the
// compiler may read them through the scalar cache after the kernel's first store: with possibly aliasing pointers every
// directory word was a VECTOR load followed by s_waitcnt vmcnt(0), i.e. a wait for every payload load in flight.  card_out
// is the same array as card_in: a wave rewrites only the cardinality of the run container it has just read.)
__global__ __launch_bounds__(256) void k_des_payload(const uint8_t* __restrict__ blob, const uint8_t* __restrict__ type_in,
                                                     const uint32_t* __restrict__ card_in, const uint32_t* __restrict__ nruns_in,
                                                     const u64* __restrict__ src_in, uint32_t* card_out,
                                                     const u64* __restrict__ off, uint8_t* __restrict__ arena,
                                                     u64 n_cont, const u64* __restrict__ bm_start,
                                                     uint32_t n_bitmaps, uint32_t* __restrict__ status) {
    const uint32_t lane = lane_id();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    // A wave's turn is two dependent round trips -- the container's directory words, then its payload -- and at one
    // container per turn that chain, not bandwidth, set the pace (3.2 M containers over 8 192 waves: 390 turns of ~3 us).
    // So the turns are pipelined two deep: while container c is validated and stored, the first step of the payload of
    // c + nwaves is in flight and the directory words of c + 2 nwaves are on their way.
    struct Raw { uint32_t tw, card, nr; u64 sp, offv; uint32_t sh; bool in; };  // directory words as loaded (nothing computed: no wait)
    struct Dir { uint32_t t, card, nr, B, d16; const uint4* S; uint4* O; };
    auto load_raw = [&](u64 c) {
        const u64 cc = c < n_cont ? c : n_cont - 1;  // (a clamped index: the loads are unconditional)
        Raw r;
        // (the type byte through the SCALAR cache, as a dword: a vector byte load here would be counted with the payload
        // loads in flight, and waiting for it would mean waiting for all of them)
        r.tw = ((const uint32_t*)type_in)[cc >> 2];
        r.sh = 8u * (uint32_t)(cc & 3u);
        r.card = card_in[cc]; r.nr = nruns_in[cc];
        r.sp = src_in[cc];
        r.offv = off[cc];
        r.in = c < n_cont;
        return r;
    };
    auto derive = [&](const Raw& r) {
        Dir d;
        d.t = (r.tw >> r.sh) & 0xFFu;
        d.card = r.card; d.nr = r.nr;
        d.B = r.in ? (d.t == T_BITSET ? 8192u : (d.t == T_ARRAY ? 2u * d.card : 4u * d.nr)) : 0u;
        d.d16 = (uint32_t)(r.sp & 15u);
        d.S = (const uint4*)(blob + (r.sp - d.d16));  // aligned 16-byte groups covering the payload
        d.O = (uint4*)(arena + r.offv);
        return d;
    };
    u64 c = wave_uniform((uint32_t)(((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (c >= n_cont) return;
    Dir cur = derive(load_raw(c)), nxt = derive(load_raw(c + nwaves));
    uint4 a0 = cur.S[16u * lane < cur.B ? lane : 0u], b0 = cur.S[16u * lane < cur.B ? lane + 1u : 0u];
    for (; c < n_cont; c += nwaves) {
        const uint4 a1 = nxt.S[16u * lane < nxt.B ? lane : 0u], b1 = nxt.S[16u * lane < nxt.B ? lane + 1u : 0u];  // (nxt.B = 0 past the end)
        const Raw nn = load_raw(c + 2 * nwaves);  // (used at the end of the turn)
        const uint32_t t = cur.t, card = cur.card, B = cur.B, d16 = cur.d16, dsw = d16 >> 2, dby = d16 & 3u;
        const uint4* __restrict__ S = cur.S;
        uint4* __restrict__ O = cur.O;
        const uint32_t n16 = (B + 15u) >> 4;
        uint32_t acc = 0;       // bitset: popcount; run: cardinality
        uint32_t carry = 0;     // last value (array) / last run end (run) of the previous step
        bool bad = false;
        for (uint32_t k0 = 0; k0 < n16; k0 += 64) {
            const uint32_t k = k0 + lane;
            const bool act = k < n16;
            uint4 a = a0, b = b0;  // (the first step arrived while the previous container was worked on)
            if (k0) {
                a = S[act ? k : 0u];
                b = S[act ? k + 1u : 0u];
            }
            const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t w[4];
            switch (dsw) {  // (wave-uniform: register indices are static in every arm)
                case 0: for (int j = 0; j < 4; ++j) w[j] = des_align(x[j + 1], x[j], dby); break;
                case 1: for (int j = 0; j < 4; ++j) w[j] = des_align(x[j + 2], x[j + 1], dby); break;
                case 2: for (int j = 0; j < 4; ++j) w[j] = des_align(x[j + 3], x[j + 2], dby); break;
                default: for (int j = 0; j < 4; ++j) w[j] = des_align(x[j + 4], x[j + 3], dby); break;
            }
            const uint32_t rem = act ? (B - 16u * k < 16u ? B - 16u * k : 16u) : 0u;  // payload bytes of this group
            if (rem < 16u) {  // (the container's last group, and the idle lanes: bytes past the payload are zero)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t vb = rem > 4u * j ? rem - 4u * j : 0u;
                    w[j] = vb >= 4u ? w[j] : (vb ? w[j] & ((1u << (8u * vb)) - 1u) : 0u);
                }
            }
            if (act) O[k] = make_uint4(w[0], w[1], w[2], w[3]);
            if (t == T_BITSET) {
                acc += __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
            } else if (t == T_ARRAY) {
                const uint32_t nv = rem >> 1;  // values of this lane: 8, less in the last group, 0 in idle lanes
                uint32_t v[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) v[h] = (w[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
#pragma unroll
                for (int h = 1; h < 8; ++h) bad |= (uint32_t)h < nv && v[h] <= v[h - 1];
                uint32_t pv = __shfl_up(v[7], 1);  // (the lane below an active lane is full)
                if (lane == 0) pv = carry;
                if (nv && k > 0 && v[0] <= pv) bad = true;
                carry = __shfl(v[7], 63);
            } else {
                const uint32_t nq = rem >> 2;  // runs of this lane
                uint32_t pe = __shfl_up((w[3] & 0xFFFFu) + (w[3] >> 16), 1);
                if (lane == 0) pe = carry;
                uint32_t e3 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t st = w[j] & 0xFFFFu, ln = w[j] >> 16, en = st + ln;
                    const bool has = (uint32_t)j < nq;
                    if (has && en > 65535u) bad = true;
                    if (has && (j > 0 || k > 0) && st <= pe + 1u) bad = true;
                    if (has) acc += ln + 1u;
                    pe = en;
                    e3 = en;
                }
                carry = __shfl(e3, 63);
            }
        }
        const uint32_t total = wave_sum(acc);
        if (t == T_BITSET && total != card) bad = true;
        const bool anybad = __ballot(bad) != 0;
        if (lane == 0) {
            if (t == T_RUN) card_out[c] = total;
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
        cur = nxt; nxt = derive(nn); a0 = a1; b0 = b1;
    }
}
