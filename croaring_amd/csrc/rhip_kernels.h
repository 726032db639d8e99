// rhip_kernels.h -- CDNA4 (gfx950, wave64) kernels of the Roaring set-operation engine.
//
// Data layout in HBM: a pool is a structure-of-arrays container DIRECTORY
// (key, type, card, nruns, byte offset; containers sorted by (bitmap, key)) plus one payload
// ARENA.  Every payload starts 16-byte aligned and is padded to a multiple of 16 bytes, so
// every kernel moves payload as 16 B/lane (1 KiB per wave instruction, fully coalesced); a loaded
// pool of larger containers pads to whole 128-byte lines instead (rhip_pool_payload_align).
// A bitset container is 1024 contiguous u64 words = 8 wave-wide 16-byte loads.
//
// Kernel inventory (one per SURVEY §2.2 row it replaces; DESIGN.md §2 / §4 have the pipeline and the table):
//   k_stage_in         small batch descriptions pulled from pinned host memory on the compute queue
//   k_count / k_emit   two-pointer key merge of roaring.c:742-768 as lane-per-key binary search + ballot ranking; a
//                      group of 64 / 32 / 16 lanes per planning unit (tile of <= 256 / 128 / 64 directory entries)
//                      -> work items in the class queues at deterministic positions; k_scan between them
//   k_bb               K1-K3: bitset (x) bitset {and,or,xor,andnot} fused with popcount, one wave per container
//                      pair, 16 x 16-byte loads in flight per lane
//   k_bba              bitset (x) bitset whose and / andnot is expected to be an array: one pass, array written
//   k_copy             pass-through containers (roaring.c:914-941 clone paths), four short ones per wave -- sixteen
//                      when the pools hold tiny containers
//   k_probe            K9/K12 for a streamed array of <= 256 values: pivot search straight from global / L2, no LDS
//   k_filter           K8/K12: array filtered by membership (and / andnot), wave-private LDS image, wave per pair
//   k_usmall           K10/K11 for a short operand (<= 255 values): rank merge into the long array
//   k_wave             K6/K10/K11: or / xor / bitset \ array with an array operand, wave-private LDS image +
//                      returning LDS atomics, wave per pair
//   k_filter_g / k_union_g   the same algebra for an X-GROUPED batch (rhip_grouped.h): the queue is counting-sorted by the
//                      image-side container, a wave rebuilds its image only when that container changes
//   k_ivl<G, MAXIV>    K13/K14/K16: interval algebra by merge path on boundary lists; four pairs per wave (G = 16) up
//                      to 31 / 127 intervals a side, one pair per wave (G = 64) up to 255
//   k_genw             K5/K7/K15 + the rest: run x bitset, long run/array pairs, bitset x bitset results that become
//                      arrays: wave-private LDS image (runs: toggle bits + prefix-xor), op + popcount + run counting,
//                      result re-typed by the reference's rules (Appendix A) and extracted with prefix sums
//   k_tail             drops empty results, builds the result directory, totals + completion word to pinned memory
//   k_join_signal / k_join_wait   join of a forked batch's auxiliary streams by flags (one-wave gate in front of k_tail);
//                      k_conc_probe: the context's self-test that kernels of two streams really overlap
//   k_place_probe      k_bb's access pattern without a queue: times a candidate result arena against the operand pool
//                      (measured placement of large result arenas)
//   k_many_*           group-by-key OR/XOR accumulation for or_many / xor_many
//   k_compact          directory compaction of the flip / many-way paths
//   k_des_* / k_ser_* / k_frz_*   portable and frozen images parsed / assembled on the device (rhip_deser.h, rhip_serial.h, rhip_frozen.h)
#pragma once
#include "rhip_common.h"
#include "rhip_plan.h"
#include "rhip_bitset.h"
#include "rhip_array.h"
#include "rhip_grouped.h"
#include "rhip_runs.h"
#include "rhip_classes.h"
#include "rhip_block.h"
