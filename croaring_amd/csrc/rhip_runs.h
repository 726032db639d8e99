// rhip_runs.h -- wave-per-pair kernels for pairs with a run operand: k_runs (interval algebra), k_genw (LDS image)
#pragma once
#include "rhip_common.h"

// ------------------------------------------------------------------ interval kernel (K13, K14, K16)
// run x run, array x run, run x array for all four ops, in O(nA + nB) work instead of rasterising 65536 bits: one
// WAVE per pair, no workgroup barrier.  Replaces the sequential interval merges
// run_container_{union,intersection,xor,andnot} (src/containers/run.c:231-283, 387-463, 348-383, 575-633),
// array_run_container_{intersection,union,andnot,lazy_xor}, run_array_container_andnot
// (mixed_intersection.c:73-111, mixed_union.c:66-108, mixed_andnot.c:277-412, mixed_xor.c:140-173).
//
// Each operand is read as a sorted BOUNDARY list b(0) <= b(1) <= ... <= b(2n-1) = s0, e0+1, s1, e1+1, ...
// (arrays: e = s).  Membership is a parity: x is in the operand iff an odd number of its boundaries are <= x.  The
// two lists (staged in LDS) are merged by MERGE PATH: lane l owns merged positions [l * per, (l+1) * per) and finds
// its split (ia, ib) with one binary search along its diagonal; the state before its chunk is just (ia & 1, ib & 1).
// It then walks its <= 16 events sequentially (one LDS read per event -- issued one event ahead -- and no branch), toggling
// inA / inB.  Only the LAST event at a position is "effective" (an array's v+1 / next-v pair, or a boundary shared by
// both lists, toggles twice at one position); a result run starts at an effective event where op(inA, inB) turns 1 and
// ends before one where it turns 0, "turns" being relative to the previous effective event -- across lanes that is one
// ballot pair and a count-leading-zeros.  ONE walk: a lane's transitions go to a private slice of the run table and are
// compacted to their scanned positions afterwards.  The run list is then typed by the reference's rules
// (convert_run_to_efficient_container etc.) and written as runs or expanded into an array; the rare bitset result is
// re-queued for k_genw.  The kernel is bound by VALU issue, not by memory: ~43 instructions per event (round 1
// evaluated every boundary with a binary search into the other list, twice: 7 ns per pair of 100-run containers;
// round 2 walked twice with boolean state the compiler kept in scalar masks: 68 instructions per event and walk).
// G lanes per pair (64 / G pairs per wave), at most MAXIV intervals per operand.  The walk of a lane's chunk is a chain
// of dependent LDS reads, so a wave is latency-bound whatever its width: sparse run-compressed data (wikileaks-noquotes:
// nine tenths of the matched pairs have <= 127 intervals a side) runs EIGHT (<= 31 intervals) or FOUR pairs per wave on 8- / 16-lane groups -- group-wide
// scans and sums by shuffles that never leave the group, group-private LDS -- and only long lists take the whole wave.
// Control flow around the collectives stays wave-uniform: the groups of a wave walk their items in lockstep.
template <uint32_t G, uint32_t MAXIV>
struct IvlShape {
    static constexpr uint32_t NG = 64 / G;                        // pairs per wave
    static constexpr uint32_t NB = 2 * MAXIV + 2;                 // >= result runs: (boundaries of both lists) / 2
    static constexpr uint32_t LBYTES = (4 * MAXIV + 15) & ~15u;   // staged payload of one operand, 16-byte padded
    static constexpr uint32_t LIST_BYTES = 4 * NG * 2 * LBYTES;   // per block
    static constexpr uint32_t RSE_BYTES = 4 * NG * 2 * NB * 2;
    static constexpr uint32_t LDS_BYTES = LIST_BYTES + RSE_BYTES;
};
// One item of a group: both lists staged at lsA (list A, then list B HALF u16 slots later), run table at RSE.
// HALF_FIXED != 0: the fixed two-slot layout of the queue kernels.  HALF_FIXED == 0 (PACKED, one pair per wave): list B
// follows list A directly, so the LDS need is set by the SUM of the two lengths -- a 100-run container against a
// 1 800-value array fits where two fixed 2 047-interval slots would not (k_genw's long-list path).
// Returns true when the result must be a bitset and retry_q is null (the caller's image path redoes the pair); with a
// retry queue the item is re-queued there.
template <uint32_t G, uint32_t HALF_FIXED>
__device__ __forceinline__ bool ivl_item(const Grp<G>& gr, uint8_t* lsA, uint16_t* RSE, bool have, const GenItem& t,
                                         const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                         const OutView& O, int kop, int cardmode, u64* pair_acc, GenItem* retry_q,
                                         uint32_t* retry_count) {
    const uint32_t lane = gr.lane, gl = gr.gl;
    {
        const uint32_t ta = t.types & 0xFFu, tb = (t.types >> 8) & 0xFFu;
        const int op = item_op(kop, t.types);  // (multi-op batches: the groups of a wave may hold different ops)
        // boundary counts (2 x intervals); an absent item is a pair of empty lists
        const uint32_t nA = have ? 2u * (ta == T_RUN ? t.nra : t.ca) : 0u;
        const uint32_t nB2 = have ? 2u * (tb == T_RUN ? t.nrb : t.cb) : 0u;
        // list B's first slot (u16 units; staging writes whole 16-slot granules)
        const uint32_t HALF = HALF_FIXED ? HALF_FIXED : ((nA + 15u) & ~15u);
        {   // stage both payloads in BOUNDARY form, u16 x[j] with boundary j = x[j] + (j & 1): a run (s, len) becomes
            // (s, s + len), an array value v becomes (v, v) -- reading a boundary is then one LDS load and one add
            // whatever the container type, and the walk below has no branch on it.  16 payload bytes per lane and round.
            auto stage = [&](uint8_t* ls, const uint8_t* __restrict__ src, bool is_run, uint32_t n2) {
                if (is_run) {
                    const uint32_t n16 = (2u * n2 + 15u) >> 4;
                    for (uint32_t i = gl; i < n16; i += G) {
                        uint4 x = ((const uint4*)src)[i];
                        x.x += x.x << 16; x.y += x.y << 16; x.z += x.z << 16; x.w += x.w << 16;
                        ((uint4*)ls)[i] = x;
                    }
                } else {
                    const uint32_t n16 = (n2 + 15u) >> 4;
                    for (uint32_t i = gl; i < n16; i += G) {
                        const uint4 x = ((const uint4*)src)[i];
                        ((uint4*)ls)[2 * i] = make_uint4((x.x & 0xFFFFu) * 0x10001u, (x.x >> 16) * 0x10001u,
                                                         (x.y & 0xFFFFu) * 0x10001u, (x.y >> 16) * 0x10001u);
                        ((uint4*)ls)[2 * i + 1] = make_uint4((x.z & 0xFFFFu) * 0x10001u, (x.z >> 16) * 0x10001u,
                                                             (x.w & 0xFFFFu) * 0x10001u, (x.w >> 16) * 0x10001u);
                    }
                }
            };
            stage(lsA, arenaA + t.offa, ta == T_RUN, nA);
            stage(lsA + 2 * HALF, arenaB + t.offb, tb == T_RUN, nB2);
            __builtin_amdgcn_wave_barrier();
        }
        const uint16_t* __restrict__ LL = (const uint16_t*)lsA;  // A's boundaries at [0, HALF), B's at [HALF, 2 HALF)
        auto bnd = [&](uint32_t base, uint32_t j) -> uint32_t { return (uint32_t)LL[base + j] + (j & 1u); };
        // ---- merge path: this lane's chunk of the merged boundary sequence
        constexpr uint32_t SENT = 0x20000u;  // past every boundary (the largest is 65536)
        const uint32_t E = nA + nB2;
        const uint32_t per = (E + G - 1u) / G;
        const uint32_t d0 = gl * per < E ? gl * per : E;
        const uint32_t d1 = d0 + per < E ? d0 + per : E;
        uint32_t ia0;
        {
            uint32_t lo = d0 > nB2 ? d0 - nB2 : 0u, hi = d0 < nA ? d0 : nA;
            while (lo < hi) {  // A goes first on ties: A[mid] <= B[d0 - mid - 1] means more of A lies before the diagonal
                const uint32_t mid = (lo + hi) >> 1;
                const bool more = bnd(0, mid) <= bnd(HALF, d0 - mid - 1u);
                lo = more ? mid + 1u : lo;
                hi = more ? hi : mid;
            }
            ia0 = lo;
        }
        const uint32_t ib0 = d0 - ia0, steps = d1 - d0;
        // op as a truth table over (inA, inB): bit (inA + 2 inB)
        const uint32_t tt = op == OP_AND ? 0x8u : op == OP_OR ? 0xEu : op == OP_XOR ? 0x6u : 0x2u;
        // ONE branch-free walk over the chunk's events.  An event at position p is EFFECTIVE when it is the last one at
        // that position (the only kind that counts); g = op state after it.  The lane's transitions (start / end values,
        // alternating) go to a lane-private slice of the run table as if the state before the chunk were 0; the true
        // state (a ballot pair and a count-leading-zeros) only changes the FIRST effective event: its start is dropped
        // (the run began in an earlier chunk) or an end is put in front of it.  The slices are then compacted to their
        // scanned positions -- independent LDS copies, into the staging area of the two lists, which is dead by now.
        // (A second walk to write them cost as much as the first: a chain of dependent LDS reads.)
        // The next boundary of each list is read ONE EVENT AHEAD (pa1 / pb1): the step that consumes a boundary only
        // selects registers, and the LDS read it issues is for the event after next -- off the dependent chain unless
        // the same list advances twice in a row.  All state is 0 / 1 integers in vector registers: as bools the compiler
        // kept it in scalar lane masks, three scalar instructions per update (the loop was 68 instructions a step).
        uint32_t heff = 0, g_first = 0, gp = 0, p_first = 0, nrec = 0;
        uint16_t* TMP = RSE + gl * per;
        {
            uint32_t ia = ia0, ib = ib0;
            uint32_t x = (ia0 & 1u) | ((ib0 & 1u) << 1);  // inA + 2 inB
            uint32_t pa = ia < nA ? bnd(0, ia) : SENT, pb = ib < nB2 ? bnd(HALF, ib) : SENT;
            uint32_t pa1 = ia + 1u < nA ? bnd(0, ia + 1u) : SENT, pb1 = ib + 1u < nB2 ? bnd(HALF, ib + 1u) : SENT;
            for (uint32_t sidx = 0; sidx < steps; ++sidx) {
                const bool tA = pa <= pb;
                const uint32_t pcur = tA ? pa : pb;
                x ^= tA ? 1u : 2u;
                ia += tA ? 1u : 0u; ib += tA ? 0u : 1u;
                const uint32_t idx = (tA ? ia : ib) + 1u, lim = tA ? nA : nB2;
                const uint32_t raw = bnd(tA ? 0u : HALF, idx);  // (two slots past the list are inside the staging area)
                const uint32_t val = idx < lim ? raw : SENT;
                pa = tA ? pa1 : pa; pb = tA ? pb : pb1;
                pa1 = tA ? val : pa1; pb1 = tA ? pb1 : val;
                const uint32_t pnext = pa < pb ? pa : pb;
                const uint32_t eff = pnext != pcur ? 1u : 0u;
                const uint32_t g = (tt >> x) & 1u;
                const uint32_t first = eff & ~heff;
                g_first = first ? g : g_first;
                p_first = first ? pcur : p_first;
                heff |= eff;
                TMP[nrec] = (uint16_t)(pcur + g - 1u);  // start: p, end: p - 1 (kept only when this is a transition)
                nrec += eff & (g ^ gp);
                gp = eff ? g : gp;
            }
        }
        const bool has_eff = heff != 0u, g_last = gp != 0u;
        // starts and ends alternate, beginning with a start
        uint32_t ns = (nrec + 1u) >> 1, ne = nrec >> 1;
        bool gprev = false;  // state after the last effective event BEFORE this chunk
        {
            const u64 mh = gr.ballot(has_eff), mg = gr.ballot(g_last);
            const u64 below = mh & ((1ull << gl) - 1ull);
            if (below) gprev = (mg >> (63 - __clzll((long long)below))) & 1ull;
        }
        const bool drop_first = gprev && has_eff && g_first;   // no start: the state was 1 already
        const bool add_end = gprev && has_eff && !g_first;     // the first event ends a run begun in an earlier chunk
        ns -= drop_first ? 1u : 0u;
        ne += add_end ? 1u : 0u;
        // (one scan for both: a lane has at most 2 x 64 transitions, the group at most 2 x 4096 of each kind)
        const uint32_t inc2 = gr.incl_scan(ns | (ne << 16));
        const uint32_t rn = __shfl(inc2, gr.glast) & 0xFFFFu;  // == total ends: every run that starts also ends (both states end at 0)
        uint16_t* OUT = (uint16_t*)lsA;
        __builtin_amdgcn_wave_barrier();  // every lane is done reading the lists
        {
            uint32_t kt = ((inc2 & 0xFFFFu) - ns) + ((inc2 >> 16) - ne);
            if (add_end) OUT[kt++] = (uint16_t)(p_first - 1u);
            const uint32_t skip = drop_first ? 1u : 0u;
            for (uint32_t j = skip; j < nrec; ++j) OUT[kt + j - skip] = TMP[j];
        }
        const uint32_t* RUN = (const uint32_t*)OUT;  // run k = RUN[k]: start | end << 16
        __builtin_amdgcn_wave_barrier();
        if (__ballot(rn != 0u) == 0ull) {  // every pair of the wave came out empty (most of a sparse `and` batch)
            if (have && gl == 0 && !cardmode) O.meta[t.out] = pack_meta(T_ARRAY, 0u, 0u);
            __builtin_amdgcn_wave_barrier();
            return false;
        }
        // ---- cardinality, typing
        uint32_t cnt = 0;
        for (uint32_t k = gl; k < rn; k += G) {
            const uint32_t w = RUN[k];
            cnt += (w >> 16) - (w & 0xFFFFu) + 1u;
        }
        const uint32_t rc = gr.sum(cnt);
        if (cardmode) {  // (wave-uniform)
            if (have && gl == 0 && rc) atomicAdd(&pair_acc[t.out], (u64)rc);
            __builtin_amdgcn_wave_barrier();
            return false;
        }
        const bool fulla = ta == T_RUN && t.ca == 65536u, fullb = tb == T_RUN && t.cb == 65536u;
        int ty = T_ARRAY;
        if (rc) ty = decide_type(op, (int)ta, (int)tb, t.ca, t.cb, fulla, fullb, rc, rn);
        // a bitset result (rare for this class): let the image kernel redo the pair
        const bool redo = have && rc && ty == T_BITSET;
        if (redo && gl == 0 && retry_q) retry_q[atomicAdd(retry_count, 1u)] = t;
        const bool wr = have && rc && !redo;
        uint8_t* outp = O.arena + t.offo;
        if (wr && ty == T_RUN) {
            uint32_t* __restrict__ o32 = (uint32_t*)outp;
            for (uint32_t k = gl; k < rn; k += G) {
                const uint32_t w = RUN[k];
                o32[k] = (w & 0xFFFFu) | (((w >> 16) - (w & 0xFFFFu)) << 16);
            }
        }
        {   // runs -> sorted array.  An array-typed result has short runs (2 rc <= 4 rn + 2), so: one lane per run
            // writes its first 8 values at the scanned position; the few longer runs are finished by the whole group,
            // one after the other.  (A binary search per output value -- 8 dependent LDS reads for each of up to 4096
            // values -- was the slowest part of the kernel on run-compressed data.)
            const bool arr = wr && ty != T_RUN;
            uint16_t* __restrict__ o16 = (uint16_t*)outp;
            const uint32_t rnm = gr.wave_max(arr ? rn : 0u);
            uint32_t runbase = 0;
            for (uint32_t k0 = 0; k0 < rnm; k0 += G) {
                const uint32_t k = k0 + gl;
                const bool v = arr && k < rn;
                const uint32_t w = v ? RUN[k] : 0u;
                const uint32_t st = w & 0xFFFFu;
                const uint32_t len = v ? (w >> 16) - st + 1u : 0u;
                const uint32_t inc = gr.incl_scan(len);
                const uint32_t pos = runbase + inc - len;
                runbase += __shfl(inc, gr.glast);
                const uint32_t lim = len < 8u ? len : 8u;
                for (uint32_t j = 0; j < lim; ++j) o16[pos + j] = (uint16_t)(st + j);
                u64 lm = gr.ballot(len > 8u);
                while (__ballot(lm != 0ull)) {
                    const bool act = lm != 0ull;
                    const uint32_t src = (lane & ~(G - 1u)) | (act ? (uint32_t)__ffsll((long long)lm) - 1u : 0u);
                    lm &= lm - 1ull;
                    const uint32_t ls = __shfl(st, src), ll = __shfl(len, src), lp = __shfl(pos, src);
                    if (act)
                        for (uint32_t j = 8u + gl; j < ll; j += G) o16[lp + j] = (uint16_t)(ls + j);
                }
            }
        }
        if (have && !redo && gl == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
        return redo && !retry_q;
    }
}

// bid / nblk: this block's index and the number of blocks of ITS size class (k_ivl_all runs the three classes in one
// launch); lds: the block's LDS, IvlShape<G, MAXIV>::LDS_BYTES of it
template <uint32_t G, uint32_t MAXIV>
__device__ __forceinline__ void ivl_body(uint8_t* lds, uint32_t bid, uint32_t nblk, const uint8_t* __restrict__ arenaA,
                                         const uint8_t* __restrict__ arenaB, const OutView& O,
                                         const GenItem* __restrict__ q, const u64* __restrict__ qrange, int kop,
                                         int cardmode, u64* pair_acc, GenItem* retry_q, uint32_t* retry_count) {
    using SH = IvlShape<G, MAXIV>;
    constexpr uint32_t NG = SH::NG, NB = SH::NB, LBYTES = SH::LBYTES;
    const Grp<G> gr;
    const uint32_t gslot = threadIdx.x / G;
    uint8_t* lsA = lds + (size_t)gslot * 2 * LBYTES;   // this group's two lists, then (after all lists) its run table
    uint16_t* RSE = (uint16_t*)(lds + SH::LIST_BYTES) + (size_t)gslot * 2 * NB;
    const uint32_t nwaves = (nblk * blockDim.x) >> 6;
    const uint32_t n = (uint32_t)(qrange[1] - qrange[0]);
    uint32_t wi = (bid * blockDim.x + threadIdx.x) >> 6;
    GenItem tnext = {};
    if (NG * wi + gr.grp < n) tnext = q[NG * wi + gr.grp];
    for (; NG * wi < n; wi += nwaves) {
        const bool have = NG * wi + gr.grp < n;
        const GenItem t = tnext;
        if (NG * (wi + nwaves) + gr.grp < n) tnext = q[NG * (wi + nwaves) + gr.grp];  // next item in flight meanwhile
        ivl_item<G, LBYTES / 2>(gr, lsA, RSE, have, t, arenaA, arenaB, O, kop, cardmode, pair_acc, retry_q, retry_count);
    }
}

// ------------------------------------------------------------------ long interval lists (items of the GENERAL queue)
// A run container against a run / an array, too long for the three size classes above but RUNSL_MAX_SUM intervals or fewer
// together: interval algebra on a whole wave with the two lists packed back to back, O(intervals) instead of k_genw's two
// 65 536-bit images (census-income, wikileaks-noquotes: most of the general queue).  The items stay in the general
// queue; k_genw<true> takes this path for them.
__device__ __forceinline__ bool gen_item_is_interval(const GenItem& t, int op, int cardmode) {
    const uint32_t ta = t.types & 0xFFu, tb = (t.types >> 8) & 0xFFu;
    if (!((ta == T_RUN || tb == T_RUN) && ta != T_BITSET && tb != T_BITSET &&
          (ta == T_RUN ? t.nra : t.ca) + (tb == T_RUN ? t.nrb : t.cb) <= RUNSL_MAX_SUM))
        return false;
    if (cardmode) return true;
    // ... unless the reference's typing rule makes the result a bitset whatever its run count (a 30 000-value run
    // container minus / xor an array: mixed_andnot.c:277-412, mixed_xor.c:104-138): with rn = 1 every rule that looks at
    // the run count answers "run", so T_BITSET here means certain -- the true cardinality is >= the bound
    const uint32_t lb = op == OP_AND ? 0u : op == OP_OR ? (t.ca > t.cb ? t.ca : t.cb)
                      : op == OP_XOR ? (t.ca > t.cb ? t.ca - t.cb : t.cb - t.ca) : (t.ca > t.cb ? t.ca - t.cb : 0u);
    return decide_type(op, (int)ta, (int)tb, t.ca, t.cb, ta == T_RUN && t.ca == 65536u, tb == T_RUN && t.cb == 65536u, lb, 1u) != T_BITSET;
}
constexpr uint32_t IVLL_LIST_BYTES = 4u * RUNSL_MAX_SUM + 64u;         // both lists + over-read slack
constexpr uint32_t IVLL_BYTES = IVLL_LIST_BYTES + 4u * RUNSL_MAX_SUM;  // + run table: 16 320 bytes a wave
// The three size classes in ONE launch (a small batch is a chain of dependent launches: each one less is ~7 us):
// blocks [0, g1) take the short lists, [g1, g1 + g2) the wide ones, the rest one pair per wave.  The block's LDS is
// the largest of the three shapes (32 KiB); at ~100 VGPRs four blocks per CU fit either way.
struct IvlQueues {
    const GenItem* q[3];   // CLS_RUNS16, CLS_RUNS16W, CLS_RUNS
    const u64* range[3];
};
// Five waves per SIMD: the kernel fits 96 VGPRs without a spill (105 left to itself = four waves) and five workgroups'
// 32 KiB are exactly a CU's LDS.  Round 5, same box, alternating: C5 `or` 0.716-0.734 -> 0.659-0.680 ms, `xor` 0.709-0.724 ->
// 0.683-0.687, everything else within noise (C5 `and`: unchanged -- its 847 000 short pairs are issue-bound).
#ifndef RHIP_IVL_WAVES
#define RHIP_IVL_WAVES 5
#endif
__global__ __launch_bounds__(256, RHIP_IVL_WAVES) void k_ivl_all(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                                 OutView O, IvlQueues Q, uint32_t g1, uint32_t g2, int op, int cardmode,
                                                 u64* pair_acc, GenItem* retry_q, uint32_t* retry_count) {
    constexpr uint32_t LDS_A = IvlShape<R16_G, R16_MAX_IV>::LDS_BYTES, LDS_B = IvlShape<16, R16W_MAX_IV>::LDS_BYTES,
                       LDS_C = IvlShape<RUNS_G, RUNS_MAX_INTERVALS>::LDS_BYTES;
    constexpr uint32_t LDS_MAX = LDS_A > LDS_B ? (LDS_A > LDS_C ? LDS_A : LDS_C) : (LDS_B > LDS_C ? LDS_B : LDS_C);
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_MAX];
    const uint32_t b = blockIdx.x;
    if (b < g1)
        ivl_body<R16_G, R16_MAX_IV>(lds, b, g1, arenaA, arenaB, O, Q.q[0], Q.range[0], op, cardmode, pair_acc, retry_q, retry_count);
    else if (b < g1 + g2)
        ivl_body<16, R16W_MAX_IV>(lds, b - g1, g2, arenaA, arenaB, O, Q.q[1], Q.range[1], op, cardmode, pair_acc, retry_q,
                                  retry_count);
    else
        ivl_body<RUNS_G, RUNS_MAX_INTERVALS>(lds, b - g1 - g2, gridDim.x - g1 - g2, arenaA, arenaB, O, Q.q[2], Q.range[2], op,
                                         cardmode, pair_acc, retry_q, retry_count);
}

// ------------------------------------------------------------------ wave-level general pair kernel (K5, K7, K13-K16)
// Every type pair the specialised kernels do not take (all pairs with a run container, plus
// bitset x bitset results that must become arrays): ONE WAVE per container pair, two wave-private
// 8 KiB LDS images (operand A, operand B; A's doubles as output staging), no workgroup barrier.  Lane l owns the 32 consecutive logical words
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
    __builtin_amdgcn_wave_barrier();  // the whole image is zero before any lane scatters into another lane's words
    const uint4* __restrict__ q4p = (const uint4*)p;
    // (four 16-byte loads of a lane in flight per round: a 3 000-value array is two global round trips, not six)
    if (type == T_ARRAY) {
        const uint32_t n16 = (card + 7u) >> 3;
        for (uint32_t i0 = 0; i0 < n16; i0 += 256) {
            uint4 q4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q4[u] = i0 + 64u * u + lane < n16 ? q4p[i0 + 64u * u + lane] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * u + lane;
                const uint32_t d[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (8 * i + h < card) {
                        const uint32_t v = (d[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                        atomicOr(&img[wphys(v >> 5)], 1u << (v & 31));
                    }
                }
            }
        }
        return;
    }
    {
        const uint32_t n16 = (nruns + 3u) >> 2;  // 4 runs {u16 value, u16 length} per 16-byte load
        for (uint32_t i0 = 0; i0 < n16; i0 += 256) {
            uint4 q4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q4[u] = i0 + 64u * u + lane < n16 ? q4p[i0 + 64u * u + lane] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * u + lane;
                const uint32_t d[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if (4 * i + h < nruns) {
                        const uint32_t s0 = d[h] & 0xFFFFu, e1 = s0 + (d[h] >> 16) + 1u;
                        atomicXor(&img[wphys(s0 >> 5)], 1u << (s0 & 31));
                        if (e1 < 65536u) atomicXor(&img[wphys(e1 >> 5)], 1u << (e1 & 31));
                    }
                }
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

// ONE-WAVE workgroups (launch with 64 threads per block): at 248 VGPRs a wave needs half a SIMD's register file, and a
// four-wave workgroup needs that on all four SIMDs of one CU at once -- next to a machine-filling k_filter / k_wave
// (whose freed slots go to whoever fits first) such a workgroup waited for the big kernel's grid to drain: 216 us for
// the 1 173 run pairs of a weather_sept_85 batch that take 47 us alone.  A single wave fits wherever two slots free up.
// INLINE_IVL (the instantiation that serves the general queue): its long interval lists (gen_item_is_interval) take the
// interval path, here, next to the image items.  Every such item is a 10-20 us chain of dependent steps whatever the
// machine does around it, so what counts is that the chains run beside other work: a kernel of their own for them
// (one-wave blocks, 71 VGPRs) was measured and dropped -- behind k_ivl_all on its stream it made census1881 or / andnot
// 10-15 % slower, as a block range of k_ivl_all / k_classes its two working waves sat on two of the four SIMDs.  The
// price here -- 256 VGPRs and a few spilled registers where the image path alone needs 246 -- is not paid by the
// retry pass (k_genw<false>).
template <bool INLINE_IVL>
__global__ __launch_bounds__(64, 2) void k_genw(const uint8_t* __restrict__ arenaA, const uint8_t* __restrict__ arenaB,
                                             OutView O, const GenItem* __restrict__ q,
                                             const u64* __restrict__ qrange, const uint32_t* __restrict__ qcount,
                                             int kop, int cardmode, u64* pair_acc,
                                             const GenItem* __restrict__ q2, const uint32_t* __restrict__ q2count) {
    // TWO 8 KiB images per wave (round 3; one image, operand A parked in 32 registers while B was built, until then):
    // both payloads are fetched together, A and B are rasterised side by side, the result is formed in registers and
    // image A becomes the output staging buffer.  The long-list interval path uses the same 16 KiB.
    __shared__ __attribute__((aligned(16))) uint32_t img_all[1][16384 / 4];
    const uint32_t lane = lane_id();
    uint32_t* ia = img_all[0];
    uint32_t* ib = ia + 2048;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    // items: queue q (its length from the section range, or from a counter), then -- when given -- the re-queued
    // results in q2 (single-stream batches run both in one launch, after every kernel that re-queues)
    const uint32_t n1 = qrange ? (uint32_t)(qrange[1] - qrange[0]) : *qcount;
    const uint32_t n = n1 + (q2 ? *q2count : 0u);
    PH_BEGIN();
    for (uint32_t wi = wave_uniform((blockIdx.x * blockDim.x + threadIdx.x) >> 6); wi < n; wi += nwaves) {
        const GenItem t = wi < n1 ? q[wi] : q2[wi - n1];
        const uint32_t ta = t.types & 0xFFu, tb = (t.types >> 8) & 0xFFu;
        const int op = item_op(kop, t.types);
        // the long interval lists of the general queue: the interval path; a result that has to be a bitset comes back
        // through the image path right away
        if (qrange && wi < n1 && gen_item_is_interval(t, op, cardmode)) {
            if (!INLINE_IVL) continue;  // (not reached: the retry pass has no section range)
            const Grp<64> gr;
            if (!ivl_item<64, 0>(gr, (uint8_t*)ia, (uint16_t*)((uint8_t*)ia + IVLL_LIST_BYTES), true, t, arenaA, arenaB, O, kop, cardmode,
                                 pair_acc, nullptr, nullptr))
                continue;
        }
        PH(0);
        wimg_build(ia, arenaA + t.offa, ta, t.ca, t.nra);
        PH(2);
        wimg_build(ib, arenaB + t.offb, tb, t.cb, t.nrb);
        __builtin_amdgcn_wave_barrier();
        PH(3);
        uint32_t r[32];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const uint32_t a = ia[wown(lane, k)], b = ib[wown(lane, k)];
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
        uint8_t* outp = O.arena + t.offo;
        uint32_t* img = ia;
        PH(4);
#include "rhip_wemit.inc"
        if (lane == 0) O.meta[t.out] = pack_meta(ty, rc, (ty == T_RUN) ? rn : 0u);
        __builtin_amdgcn_wave_barrier();
        PH(5);
    }
    PH_FLUSH(24);
}
