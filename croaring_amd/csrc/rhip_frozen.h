// rhip_frozen.h -- the FROZEN serialization format on the device ("frozen-style zero-copy load"),
// both directions, for 32-bit pools (roaring64 has no frozen format).
//
// Layout of one image (src/roaring.c:3176-3205), read from its END:
//   <bitset zone><run zone><array zone><keys u16 x n><counts u16 x n><typecodes u8 x n><header u32>
// zones = the payloads of all containers of one type in container order; counts[i] = cardinality - 1 (bitset, array)
// or n_runs (run); header = (n << 15) | FROZEN_COOKIE.  The reference's reader (roaring_bitmap_frozen_view,
// roaring.c:3330-3457) builds pointers INTO a 32-byte aligned buffer; the device has its own arena, so the loader is a
// copy and the alignment requirement falls away -- the zones are already what a GPU wants: three flat, type-pure
// streams, one coalesced copy each.  Writing (roaring_bitmap_frozen_size_in_bytes / _serialize, roaring.c:3207-3328)
// is the same walk in the other direction.
#pragma once
#include "rhip_common.h"
#include "rhip_deser.h"
#include "rhip_serial.h"

constexpr uint32_t FROZEN_COOKIE = 13766u;  // roaring_array.h:38

// ------------------------------------------------------------------ writing
// One WAVE per selected bitmap (container range [seg_c0[i], seg_c1[i])): zone sizes and the image size.
__global__ __launch_bounds__(256) void k_frz_size(PoolView P, const u64* __restrict__ seg_c0, const u64* __restrict__ seg_c1,
                                                  uint32_t n_sel, uint32_t* __restrict__ size,
                                                  uint32_t* __restrict__ zones /* [n_sel][2]: bitset, run bytes */) {
    const uint32_t lane = lane_id();
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_sel) return;  // wave-uniform
    const u64 c0 = seg_c0[i], c1 = seg_c1[i];
    u64 zb = 0, zr = 0, za = 0;
    for (u64 c = c0 + lane; c < c1; c += 64) {
        const uint8_t t = P.type[c];
        const uint32_t b = payload_bytes(t, P.card[c], P.nruns[c]);
        if (t == T_BITSET) zb += b;
        else if (t == T_RUN) zr += b;
        else za += b;
    }
    zb = wave_sum64(zb); zr = wave_sum64(zr); za = wave_sum64(za);
    if (lane == 0) {
        size[i] = (uint32_t)(zb + zr + za + 5ull * (c1 - c0) + 4ull);
        zones[2 * i] = (uint32_t)zb;
        zones[2 * i + 1] = (uint32_t)zr;
    }
}

// One WAVE per selected bitmap: keys / counts / typecodes / header, and the destination (absolute blob offset) of every
// container payload for k_ser_copy.  The image of bitmap i starts at boff[i] (a multiple of 32: what a zero-copy
// reader of the blob needs) and is size[i] bytes long.
__global__ __launch_bounds__(256) void k_frz_layout(PoolView P, const u64* __restrict__ seg_c0, const u64* __restrict__ seg_c1,
                                                    uint32_t n_sel, const u64* __restrict__ boff,
                                                    const uint32_t* __restrict__ size, const uint32_t* __restrict__ zones,
                                                    uint8_t* __restrict__ blob, u64* __restrict__ dst) {
    const uint32_t lane = lane_id();
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_sel) return;  // wave-uniform
    const u64 c0 = seg_c0[i], c1 = seg_c1[i];
    const uint32_t n = (uint32_t)(c1 - c0);
    const u64 image0 = boff[i];
    uint8_t* tail = blob + image0 + size[i] - 4u - 5ull * n;
    uint8_t *keys = tail, *counts = tail + 2ull * n, *types = tail + 4ull * n;
    if (lane == 0) st_le32(types + n, (n << 15) | FROZEN_COOKIE);
    u64 at_b = image0, at_r = image0 + zones[2 * i], at_a = at_r + zones[2 * i + 1];
    for (uint32_t j0 = 0; j0 < n; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t sb = 0, sr = 0, sa = 0;
        uint8_t t = 0;
        if (j < n) {
            const u64 c = c0 + j;
            t = P.type[c];
            const uint32_t cd = P.card[c], nr = P.nruns[c];
            const uint32_t b = payload_bytes(t, cd, nr);
            if (t == T_BITSET) sb = b;
            else if (t == T_RUN) sr = b;
            else sa = b;
            st_le16(keys + 2u * j, (uint32_t)(P.key[c] & 0xFFFFu));
            st_le16(counts + 2u * j, t == T_RUN ? nr : cd - 1u);
            types[j] = t;
        }
        const uint32_t ib = wave_incl_scan(sb), ir = wave_incl_scan(sr), ia = wave_incl_scan(sa);
        if (j < n) dst[c0 + j] = t == T_BITSET ? at_b + ib - sb : t == T_RUN ? at_r + ir - sr : at_a + ia - sa;
        at_b += __shfl(ib, 63); at_r += __shfl(ir, 63); at_a += __shfl(ia, 63);
    }
}

// ------------------------------------------------------------------ reading
// One WAVE per image blob[offs[i], offs[i] + lens[i]).  FILL = false: ncont[i] = number of containers, status = first
// rejected image.  FILL = true: directory entries from bm_start[i] on (DesOut of rhip_deser.h: the payload pass that
// follows -- k_des_payload -- is the portable loader's, validation included).
// Accepted = what roaring_bitmap_frozen_view accepts (cookie, typecodes 1..3, length EXACTLY zones + 5 n + 4) AND
// roaring_bitmap_internal_validate would pass (roaring.c:454-523): keys strictly increasing, array cardinality <= 4096,
// bitset cardinality > 4096, at least one run; the payload checks happen in k_des_payload.  The view itself trusts its
// input ("the bitmap is not validated", roaring.h:855-860); a pool is an operand of kernels that rely on these
// invariants, so the loader is as strict as the portable one.
template <bool FILL>
__global__ __launch_bounds__(256) void k_frz_walk(const uint8_t* __restrict__ blob, const u64* __restrict__ offs,
                                                  const u64* __restrict__ lens, uint32_t n_bitmaps,
                                                  const u64* __restrict__ bm_start, DesOut D,
                                                  uint32_t* __restrict__ ncont, uint32_t* status) {
    const uint32_t lane = lane_id();
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n_bitmaps) return;  // wave-uniform
    const u64 p = offs[i], len = lens[i];
    bool ok = len >= 4;
    uint32_t n = 0;
    if (ok) {
        const uint32_t h = ld_le32(blob + p + len - 4);
        ok = (h & 0x7FFFu) == FROZEN_COOKIE;
        n = h >> 15;
        ok = ok && n <= 65536u && len >= 4ull + 5ull * n;
    }
    if (ok) {
        const uint8_t* keys = blob + p + len - 4 - 5ull * n;
        const uint8_t* counts = keys + 2ull * n;
        const uint8_t* types = keys + 4ull * n;
        // pass 1: zone sizes, typecodes, per-type count rules, key order
        u64 zb = 0, zr = 0, za = 0;
        uint32_t prev_key = 0;
        bool bad = false;
        for (uint32_t j0 = 0; j0 < n; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool live = j < n;
            uint32_t t = 0, cnt = 0, k16 = 0;
            if (live) {
                t = types[j];
                cnt = ld_le16(counts + 2ull * j);
                k16 = ld_le16(keys + 2ull * j);
                if (t == T_BITSET) { zb += 8192u; bad |= cnt + 1u <= 4096u; }
                else if (t == T_RUN) { zr += 4ull * cnt; bad |= cnt == 0u; }
                else if (t == T_ARRAY) { za += 2ull * (cnt + 1u); bad |= cnt + 1u > 4096u; }
                else bad = true;
            }
            uint32_t pk = __shfl_up(k16, 1);
            if (lane == 0) pk = prev_key;
            bad |= live && j > 0 && k16 <= pk;
            prev_key = __shfl(k16, 63);
        }
        zb = wave_sum64(zb); zr = wave_sum64(zr); za = wave_sum64(za);
        ok = __ballot(bad) == 0 && len == zb + zr + za + 5ull * n + 4ull;
        if (FILL && ok) {
            const u64 cbase = bm_start[i];
            u64 at_b = p, at_r = p + zb, at_a = p + zb + zr;
            for (uint32_t j0 = 0; j0 < n; j0 += 64) {
                const uint32_t j = j0 + lane;
                uint32_t sb = 0, sr = 0, sa = 0, t = 0, cnt = 0;
                if (j < n) {
                    t = types[j];
                    cnt = ld_le16(counts + 2ull * j);
                    if (t == T_BITSET) sb = 8192u;
                    else if (t == T_RUN) sr = 4u * cnt;
                    else sa = 2u * (cnt + 1u);
                }
                const uint32_t ib = wave_incl_scan(sb), ir = wave_incl_scan(sr), ia = wave_incl_scan(sa);
                if (j < n) {
                    const u64 c = cbase + j;
                    D.key[c] = ld_le16(keys + 2ull * j);
                    D.type[c] = (uint8_t)t;
                    D.card[c] = t == T_RUN ? 0u : cnt + 1u;  // run cardinalities come from the payload pass
                    D.nruns[c] = t == T_RUN ? cnt : 0u;
                    D.src[c] = t == T_BITSET ? at_b + ib - sb : t == T_RUN ? at_r + ir - sr : at_a + ia - sa;
                    D.slot[c] = ((sb + sr + sa) + D.amask) & ~D.amask;
                }
                at_b += __shfl(ib, 63); at_r += __shfl(ir, 63); at_a += __shfl(ia, 63);
            }
        }
    }
    if (lane == 0) {
        if (!FILL) ncont[i] = ok ? n : 0u;
        if (!ok) atomicMin(status, i);
    }
}
