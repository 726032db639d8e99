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
#include "rhip_common.h"
#include "rhip_plan.h"
#include "rhip_bitset.h"
#include "rhip_array.h"
#include "rhip_runs.h"
#include "rhip_block.h"
