// rhip_classes.h -- the class kernels of a SMALL batch in one launch.
//
// A batch below the fork threshold runs its class kernels on one stream: census1881 `and` is k_ivl_all 14 us -> k_filter
// 9 us -> k_probe 7 us -> ..., each a handful of items whose duration is one item's latency chain plus a kernel
// boundary.  Here they are ONE launch: block ranges map to the class bodies (the shape k_ivl_all already has for the
// three interval size classes), so the chains run side by side and the batch pays the longest of them once.  Every
// body is the same code as its stand-alone kernel (x_body in rhip_array.h / rhip_bitset.h / rhip_runs.h); the
// templated ones run as their OP_ITEM instantiation -- k_emit writes the op into every item.  Register and LDS
// footprint is the largest of the bodies (~120 VGPRs, 32 KiB), which is why big batches keep their own kernels.
// k_genw (248 VGPRs, consumer of the re-queued results) stays a launch of its own behind this one.
#pragma once
#include "rhip_array.h"
#include "rhip_bitset.h"
#include "rhip_runs.h"

enum { CSEG_IVL16 = 0, CSEG_IVL16W, CSEG_IVL64, CSEG_FILT, CSEG_PROBE, CSEG_USMALL, CSEG_WAVE, CSEG_BA, CSEG_BBA, CSEG_BB,
       CSEG_COPY, N_CSEG };
struct ClassLaunch {
    const uint8_t* arenaA;
    const uint8_t* arenaB;
    OutView O;
    const u64* ranges;        // section ranges of the batch (device)
    const BBItem *q_bb, *q_bba;
    const FatItem *q_filt, *q_probe, *q_usmall, *q_wave, *q_ba;
    const CopyItem* q_copy;
    const GenItem *q_r16, *q_r16w, *q_r64;
    GenItem* retry_q;
    uint32_t* retry_count;
    u64* pair_acc;
    int kop, cardmode;
    uint32_t copy_per_wave;   // 4 or 16 (copy_body)
    uint32_t nb[N_CSEG];      // blocks of each segment (0: the class cannot occur)
};
constexpr uint32_t CLASSES_LDS_WORDS = FILTER_LDS_WORDS > 8192 ? FILTER_LDS_WORDS : 8192;
static_assert(USMALL_LDS_WORDS <= CLASSES_LDS_WORDS, "k_classes: LDS of the largest body");

__global__ __launch_bounds__(256) void k_classes(ClassLaunch L) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[CLASSES_LDS_WORDS];
    uint32_t b = blockIdx.x, seg = 0;
    while (seg < N_CSEG && b >= L.nb[seg]) { b -= L.nb[seg]; ++seg; }
    const uint32_t nblk = seg < N_CSEG ? L.nb[seg] : 0u;
    const u64* R = L.ranges;
    switch (seg) {
        case CSEG_IVL16:
            ivl_body<R16_G, R16_MAX_IV>((uint8_t*)lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_r16, R + 2 * SEC_RUNS16, L.kop, L.cardmode,
                                     L.pair_acc, L.retry_q, L.retry_count);
            break;
        case CSEG_IVL16W:
            ivl_body<16, R16W_MAX_IV>((uint8_t*)lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_r16w, R + 2 * SEC_RUNS16W, L.kop, L.cardmode,
                                      L.pair_acc, L.retry_q, L.retry_count);
            break;
        case CSEG_IVL64:
            ivl_body<RUNS_G, RUNS_MAX_INTERVALS>((uint8_t*)lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_r64, R + 2 * SEC_RUNS, L.kop,
                                             L.cardmode, L.pair_acc, L.retry_q, L.retry_count);
            break;
        case CSEG_FILT:
            filter_body<true>(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_filt, R + 2 * SEC_FILT, L.kop, L.cardmode, L.pair_acc);
            break;
        case CSEG_PROBE:
            probe_body(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_probe, R + 2 * SEC_PROBE, L.kop, L.cardmode, L.pair_acc);
            break;
        case CSEG_USMALL:
            usmall_body<false>(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_usmall, R + 2 * SEC_USMALL, L.kop);
            break;
        case CSEG_WAVE:
            wave_body(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_wave, R + 2 * SEC_WAVE, L.kop);
            break;
        case CSEG_BA:
            ba_body<OP_ITEM>(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_ba, R + 2 * SEC_BA, L.retry_q, L.retry_count);
            break;
        case CSEG_BBA:
            bba_body<OP_ITEM>(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_bba, R + 2 * SEC_BBA);
            break;
        case CSEG_BB:
            bb_body<OP_ITEM>(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_bb, R + 2 * SEC_BB, L.cardmode, L.pair_acc, L.retry_q,
                             L.retry_count);
            break;
        case CSEG_COPY:
            copy_body(lds, b, nblk, L.arenaA, L.arenaB, L.O, L.q_copy, R + 2 * SEC_COPY, L.copy_per_wave);
            break;
        default: break;
    }
}
