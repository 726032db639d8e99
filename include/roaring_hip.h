/*
 * roaring_hip.h -- C ABI of libroaring_hip.so, the MI355X (gfx950) Roaring
 * set-operation engine.
 *
 * Boundary: CRoaring has no plug-in registry; its boundary is
 * the C ABI of libroaring.  Per-call offload of a sub-microsecond CPU op can
 * never win, so the engine's native unit of work is a BATCH of
 * bitmap pairs over DEVICE-RESIDENT bitmaps.  Each entry point below names the
 * reference entry point (file:line under the reference tree) whose semantics --
 * result set, result container types, empty-container dropping, error
 * convention -- it reproduces for every element of the batch.  The
 * CRoaring-named per-call drop-ins (roaring_bitmap_and, ...) built on top of
 * this API are declared in roaring_hip_compat.h.
 *
 * Conventions
 *  - plain C types only; device buffers are opaque inside rhip_pool_t.
 *  - pointer-returning functions return NULL on error (CRoaring convention,
 *    include/roaring/roaring.h:216-226); int-returning functions return 0 on
 *    success, a negative rhip_status otherwise.  rhip_last_error() gives text.
 *  - there is NO CPU fallback: every entry point fails if no HIP device is
 *    usable.
 *  - wire format = portable RoaringFormatSpec (src/roaring_array.c:469-813);
 *    64-bit pools use the portable 64-bit format (src/roaring64.c:2323-2393).
 *  - threading: a CONTEXT IS SINGLE-THREADED.  Every call that takes a context
 *    (or a pool / batch of it) shares its stream, its pinned staging areas and
 *    its scratch buffers; callers serialise them (one context per thread is the
 *    intended use -- contexts are independent of each other, also on one
 *    device).  The exceptions, safe from any thread at any time: rhip_pool_free
 *    (a finaliser may run anywhere; a pool that is an operand of batches in
 *    flight is released by the last of them), rhip_last_error (thread-local)
 *    and rhip_version.  The reference's functions are re-entrant on distinct
 *    bitmaps (include/roaring/roaring.h:102-113); the CRoaring-named drop-ins of
 *    roaring_hip_compat.h keep that property (they lock around the device).
 */
#ifndef ROARING_HIP_H
#define ROARING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rhip_ctx_s rhip_ctx_t;   /* one per process+device: stream, scratch */
typedef struct rhip_pool_s rhip_pool_t; /* a device-resident, immutable set of bitmaps */

typedef enum { RHIP_AND = 0, RHIP_OR = 1, RHIP_XOR = 2, RHIP_ANDNOT = 3 } rhip_op;
typedef enum {
    RHIP_PRED_INTERSECT = 0,        /* roaring_bitmap_intersect        (roaring.h:237) */
    RHIP_PRED_IS_SUBSET = 1,        /* roaring_bitmap_is_subset        (roaring.h:905) */
    RHIP_PRED_IS_STRICT_SUBSET = 2, /* roaring_bitmap_is_strict_subset (roaring.h:912) */
    RHIP_PRED_EQUALS = 3            /* roaring_bitmap_equals           (roaring.h:899) */
} rhip_pred;
typedef enum {
    RHIP_OK = 0,
    RHIP_ERR_DEVICE = -1,   /* no device / HIP runtime error */
    RHIP_ERR_ALLOC = -2,    /* device or host allocation failed */
    RHIP_ERR_FORMAT = -3,   /* malformed portable buffer */
    RHIP_ERR_ARG = -4       /* bad argument (index out of range, ...) */
} rhip_status;

/* container type codes, identical to include/roaring/containers/containers.h:48-53 */
enum { RHIP_BITSET = 1, RHIP_ARRAY = 2, RHIP_RUN = 3 };

/* ---- context ---------------------------------------------------------- */
/* device < 0 selects the current HIP device.  NULL if no device. */
rhip_ctx_t *rhip_ctx_create(int device);
void rhip_ctx_destroy(rhip_ctx_t *ctx);
/* the hipStream_t every kernel of this context is launched on */
void *rhip_ctx_stream(rhip_ctx_t *ctx);
int rhip_ctx_synchronize(rhip_ctx_t *ctx);
const char *rhip_last_error(void);
const char *rhip_version(void);

/* ---- pools: loading / storing ------------------------------------------ */
/* roaring_bitmap_portable_deserialize_safe (include/roaring/roaring.h:746,
 * src/roaring.c:1631-1651) for n buffers at once: headers are parsed on the
 * host, payloads land in one HBM arena (bitset words 16-byte aligned and
 * contiguous).  Container invariants are those of
 * roaring_bitmap_internal_validate (src/roaring.c:454-523); buffers violating
 * them are rejected with RHIP_ERR_FORMAT. */
rhip_pool_t *rhip_pool_from_portable(rhip_ctx_t *ctx, size_t n, const char *const *bufs, const size_t *lens);
/* same for the 64-bit portable format: roaring64_bitmap_portable_deserialize_safe
 * (include/roaring/roaring64.h:652, src/roaring64.c:2442-2535).  Keys are the
 * high 48 bits. */
rhip_pool_t *rhip_pool_from_portable64(rhip_ctx_t *ctx, size_t n, const char *const *bufs, const size_t *lens);
/* Same loaders for n images packed in ONE host blob (image i = blob[offsets[i], offsets[i] + lens[i]), e.g. the
 * output of rhip_pool_portable_serialize_many): a single host-to-device copy, then headers are parsed and every
 * container is validated and moved into place by kernels.  Accepts and rejects exactly what the loaders above
 * do.  is64 != 0: the images are roaring64 portable images. */
rhip_pool_t *rhip_pool_from_blob(rhip_ctx_t *ctx, const char *blob, size_t blob_bytes, size_t n,
                                 const uint64_t *offsets, const uint64_t *lens, int is64);
/* The FROZEN format (include/roaring/roaring.h:815-860, layout src/roaring.c:3176-3205): n images in one host blob,
 * parsed and moved into place by kernels like the loader above.  roaring_bitmap_frozen_view (src/roaring.c:3330-3457)
 * points INTO a 32-byte aligned buffer; a device pool is a copy, so neither the blob nor the offsets need any alignment.
 * Accepted: what frozen_view accepts (cookie 13766, typecodes 1..3, length exactly zones + 5 n + 4) and
 * roaring_bitmap_internal_validate passes -- the view does not validate its input, a pool must hold valid containers:
 * unsorted keys, an array above 4096 values, a bitset of 4096 or fewer, zero runs, or a bad payload give
 * RHIP_ERR_FORMAT.  32-bit pools only. */
rhip_pool_t *rhip_pool_from_frozen(rhip_ctx_t *ctx, const char *blob, size_t blob_bytes, size_t n,
                                   const uint64_t *offsets, const uint64_t *lens);
/* roaring_bitmap_of_ptr (roaring.h:88, src/roaring.c:195-199) / roaring64_bitmap_of_ptr (roaring64.h:92) for n
 * bitmaps at once, built on the device: bitmap i = values[offsets[i] .. offsets[i+1]) (offsets[0] = 0), STRICTLY
 * INCREASING inside a bitmap (anything else is rejected -- sort and deduplicate first).  Containers come out as the
 * reference leaves them after add_many: arrays up to 4096 values, bitsets above; apply rhip_pool_run_optimize for
 * the benchmark pipeline of_ptr -> run_optimize (benchmarks/benchmark.cpp:1938-1942). */
rhip_pool_t *rhip_pool_from_sorted_u32(rhip_ctx_t *ctx, size_t n, const uint32_t *values, const uint64_t *offsets);
rhip_pool_t *rhip_pool_from_sorted_u64(rhip_ctx_t *ctx, size_t n, const uint64_t *values, const uint64_t *offsets);
/* SURVEY §8d C2 generator: n_bitmaps bitmaps with keys 0..n_containers-1, all
 * bitset containers, word w of bitmap b = splitmix64 stream seeded
 * seed + b (generated on the device). */
rhip_pool_t *rhip_pool_synth_bitset(rhip_ctx_t *ctx, uint32_t n_bitmaps, uint32_t n_containers, uint64_t seed);
/* SURVEY §8d C4 generator (BASELINE config[3]: or_many over 100 000 sparse array-dominant bitmaps), HOST side, no
 * device needed: portable images of the sparse bitmaps first, first + stride, ... (count of them).
 * Bitmap b draws from its own pcg32 stream (benchmarks/random.h:18-31; state = b, inc = 2 b + 1): 32 distinct keys
 * k = pcg32() & 4095 (duplicates rejected), then per key in ascending order card = 1 + (pcg32() & 511) and `card`
 * distinct values v = pcg32() & 0xFFFF (duplicates rejected) -- all array containers.
 * rhip_synth_sparse_sizes fills offsets[0..count] (offsets[k] = start of image k in a back-to-back packing,
 * offsets[count] = total bytes); rhip_synth_sparse_fill writes the images into buf at those offsets.  Feed the
 * blob to rhip_pool_from_blob (and the same bytes to the CPU reference). */
int rhip_synth_sparse_sizes(uint64_t first, uint64_t stride, size_t count, uint64_t *offsets);
int rhip_synth_sparse_fill(uint64_t first, uint64_t stride, size_t count, const uint64_t *offsets, char *buf);
void rhip_pool_free(rhip_pool_t *pool); /* roaring_bitmap_free, roaring.h:365 */

uint32_t rhip_pool_size(const rhip_pool_t *pool);          /* number of bitmaps */
uint64_t rhip_pool_containers(const rhip_pool_t *pool);    /* total containers */
int rhip_pool_is64(const rhip_pool_t *pool);
/* largest container key of the pool (high 16 bits of a value; high 48 for 64-bit pools), 0 for an empty pool;
 * computed once per pool and cached */
int rhip_pool_max_key(rhip_pool_t *pool, uint64_t *out);
/* payload bytes (bitset 8192, array 2*card, run 4*n_runs) of all containers */
uint64_t rhip_pool_payload_bytes(rhip_pool_t *pool);
/* bytes of HBM the pool's payload arena occupies (padding included), and the slot granule its loader chose: 16, or 128
 * (whole cache lines) when the images average 256 bytes per container or more -- RHIP_POOL_ALIGN=16|128 pins it.  The
 * many-way gather reads ~512-byte members at random offsets: line-aligned slots cost ~12 % more arena and save the
 * 7/8 of a line an unaligned member straddles. */
uint64_t rhip_pool_arena_bytes(const rhip_pool_t *pool);
uint32_t rhip_pool_payload_align(const rhip_pool_t *pool);
/* per-type container counts out[0]=bitset out[1]=array out[2]=run */
int rhip_pool_type_counts(rhip_pool_t *pool, uint64_t out[3]);

/* roaring_bitmap_portable_size_in_bytes (roaring.h:791) of bitmap i (64-bit
 * pools: roaring64_bitmap_portable_size_in_bytes, roaring64.h:595). 0 on error. */
size_t rhip_pool_portable_size(rhip_pool_t *pool, uint32_t i);
/* roaring_bitmap_portable_serialize (roaring.h:807): writes bitmap i, returns bytes written */
size_t rhip_pool_portable_serialize(rhip_pool_t *pool, uint32_t i, char *buf);
/* roaring_bitmap_get_cardinality (roaring.h:537, src/roaring.c:1436-1443) of every bitmap */
int rhip_pool_cardinalities(rhip_pool_t *pool, uint64_t *out /* [rhip_pool_size] */);
/* Bulk form of the two calls above: the portable images of pool[ids[0..n)] (ids == NULL: every bitmap, in order;
 * ids must not repeat; 64-bit pools: ids must be NULL) are assembled on the device back to back and downloaded in
 * one copy.
 * rhip_pool_portable_sizes fills offsets[0..n_sel] (offsets[k] = start of image k, offsets[n_sel] = total bytes);
 * rhip_pool_portable_serialize_many writes the images into buf (capacity cap), optionally the same offsets, and
 * returns the bytes written (0 on error).  Each image is byte-identical to rhip_pool_portable_serialize's. */
int rhip_pool_portable_sizes(rhip_pool_t *pool, size_t n, const uint32_t *ids, uint64_t *offsets);
size_t rhip_pool_portable_serialize_many(rhip_pool_t *pool, size_t n, const uint32_t *ids, char *buf, size_t cap,
                                         uint64_t *offsets);
/* roaring_bitmap_frozen_size_in_bytes / roaring_bitmap_frozen_serialize (roaring.h:829-846, src/roaring.c:3207-3328) for
 * pool[ids[0..n)] (ids == NULL: every bitmap; 32-bit pools only), assembled on the device and downloaded in one copy.
 * The images are packed at 32-BYTE ALIGNED offsets -- what roaring_bitmap_frozen_view requires of its buffer, so a host
 * that places the blob at a 32-byte aligned address can view every image in place; the gaps are zero.
 * offsets[k] = start of image k, offsets[n_sel] = bytes of the packed blob; lens[k] (may be NULL) = exact length of
 * image k = roaring_bitmap_frozen_size_in_bytes.  Each image is byte-identical to the reference's. */
int rhip_pool_frozen_sizes(rhip_pool_t *pool, size_t n, const uint32_t *ids, uint64_t *offsets, uint64_t *lens);
size_t rhip_pool_frozen_serialize_many(rhip_pool_t *pool, size_t n, const uint32_t *ids, char *buf, size_t cap,
                                       uint64_t *offsets, uint64_t *lens);
/* roaring_bitmap_to_uint32_array (roaring.h:571, src/roaring.c:1510-1512) / roaring64_bitmap_to_uint64_array
 * (roaring64.h:768) for the whole pool: the sorted values of bitmap 0, then bitmap 1, ...; offsets (n+1 entries,
 * may be NULL) receives where each bitmap starts.  out == NULL: offsets only.  capacity is in values. */
int rhip_pool_to_u32(rhip_pool_t *pool, uint32_t *out, size_t capacity, uint64_t *offsets);
int rhip_pool_to_u64(rhip_pool_t *pool, uint64_t *out, size_t capacity, uint64_t *offsets);

/* ---- pairwise set operations ------------------------------------------- */
/* For k in [0,npairs): result k = op(A[lhs[k]], B[rhs[k]]) with the exact
 * semantics of roaring_bitmap_and (roaring.h:225, src/roaring.c:731-770),
 * _or (:288, :877-953), _xor (:320, :1121-1196), _andnot (:342, :1275-1338),
 * including the per-container result types of container_and/or/xor/andnot
 * (include/roaring/containers/containers.h:726-806, 1008-1103, 1449-1524,
 * 1783-1876).  A and B may be the same pool.  `reuse` (may be NULL) is a result
 * pool from an earlier call whose device buffers are recycled (it is consumed:
 * the returned pool replaces it). */
rhip_pool_t *rhip_pairwise(rhip_ctx_t *ctx, rhip_op op, rhip_pool_t *A, rhip_pool_t *B, size_t npairs,
                           const uint32_t *lhs, const uint32_t *rhs, rhip_pool_t *reuse);
/* The same call in two halves, so that the host side of batch i+1 (pair-list pass, staging, launches: the part of a
 * small batch that the device waits for) runs while the kernels of batch i execute.  rhip_pairwise_begin enqueues
 * everything and returns without waiting for the device; `lhs` / `rhs` are copied and may be freed at once.
 * rhip_pairwise_end waits for that batch and returns its result pool (NULL on error; the handle is consumed either
 * way).  Up to RHIP_MAX_BATCHES_IN_FLIGHT batches of one context may be in flight, ended in any order; they execute in
 * begin order on the context's stream.  Until a batch has ended, its operand pools cannot be updated in place or
 * recycled (as `reuse` they are refused, RHIP_ERR_ARG); rhip_pool_free of an operand is accepted and DEFERRED -- the last
 * batch that reads the pool releases it when it ends; the batch's result cannot be an operand or the `reuse` of another
 * batch (RHIP_ERR_ARG); and any other call on the context simply waits for the batches in flight.  rhip_pairwise == begin followed by end.
 *
 * Placement of a NEW large result arena.  When a call has to allocate a result arena of 2 GiB or more beside an operand
 * pool of 64 MiB or more, and no batch of the context is in flight, the library does not take the first allocation: where
 * the arena lies -- its physical pages on some machines, its VIRTUAL address on others -- moves the bitset
 * kernel by up to 17 %, so it measures.  Stage 0: the arena is COMPOSED -- up to three times as many 1 GiB chunks as it
 * needs are created (hipMemCreate), each timed alone with the kernel's access pattern against the operand pool (the rate is
 * a property of the chunk: DESIGN.md 3), the best mapped side by side; transient footprint 3 x the arena, the chunks not
 * taken stay with the context (with their rates, for the next arena beside the same operand) until it is steady.  Only
 * when the composition stays below the bar -- stage 1: up to RHIP_ARENA_TRIES (10) candidate allocations, each timed with the
 * kernel's access pattern against the operand pool, the first that streams at 6.25 TB/s taken; they stay allocated until
 * the choice, and the search ends when they reach half of the device memory that was free when it began.  Stage 2, when no
 * candidate reaches the bar: one more allocation (hipMemCreate) is mapped at one position of a reserved address range
 * after the other (512 GiB of address space, no memory; ~2 ms a position) until one does.  This is the one place where
 * rhip_pairwise_begin WAITS for the device -- 20-60 ms, once per NEW result pool (longer while the driver is still clearing
 * memory another process has just freed).  The losers stay with the context as spares (the next search probes them first)
 * until it has run 16 batches without placing, rhip_ctx_trim, rhip_ctx_destroy, or any allocation of this library fails;
 * RHIP_ARENA_SPARES=0 releases them at once.  A pool handed back through `reuse` keeps its arena and its placement, and
 * the arena of a pool that is freed is parked with its context and taken back, as placed, by the next result pool for the
 * same operand: steady-state callers never meet the search.  RHIP_ARENA_VMM=0 leaves stages 0 and 2 out; RHIP_ARENA_TRIES=0
 * (environment, read by rhip_ctx_create) turns placement off. */
#define RHIP_MAX_BATCHES_IN_FLIGHT 4
typedef struct rhip_batch_s rhip_batch_t;
rhip_batch_t *rhip_pairwise_begin(rhip_ctx_t *ctx, rhip_op op, rhip_pool_t *A, rhip_pool_t *B, size_t npairs,
                                  const uint32_t *lhs, const uint32_t *rhs, rhip_pool_t *reuse);
rhip_pool_t *rhip_pairwise_end(rhip_batch_t *batch);
/* Several ops over ONE pair list in one batch -- the reference's benchmark loop issues exactly that, and / or / xor /
 * andnot of every pair one after the other (benchmarks/benchmark.cpp:2035-2091).  The batch is planned once: one
 * staging copy of the pair list, one key-merge launch, one scan, one emit, class kernels whose queues hold the items
 * of all the ops (every item carries its op), one tail.  Result: a pool of n_ops x npairs bitmaps, bitmap
 * o * npairs + k = ops[o](A[lhs[k]], B[rhs[k]]), each byte-identical to what rhip_pairwise gives for that op.
 * 1 <= n_ops <= 4; ops may repeat.  _begin / rhip_pairwise_end as for rhip_pairwise. */
rhip_pool_t *rhip_pairwise_multi(rhip_ctx_t *ctx, size_t n_ops, const rhip_op *ops, rhip_pool_t *A, rhip_pool_t *B,
                                 size_t npairs, const uint32_t *lhs, const uint32_t *rhs, rhip_pool_t *reuse);
rhip_batch_t *rhip_pairwise_multi_begin(rhip_ctx_t *ctx, size_t n_ops, const rhip_op *ops, rhip_pool_t *A,
                                        rhip_pool_t *B, size_t npairs, const uint32_t *lhs, const uint32_t *rhs,
                                        rhip_pool_t *reuse);
/* ---- prepared pair lists --------------------------------------------------------------------------------------------
 * The reference has no pair list: its benchmark walks the bitmaps -- every op of every unordered pair, or of
 * successive bitmaps (benchmarks/benchmark.cpp:2035-2091, "successive_*" / "*_all_pairs" loops) -- and a query engine
 * evaluates the same pairs again and again with different ops.  A pair list that is used more than once is prepared
 * ONCE: validated, its indices resident on the device, the sums that size a batch (container counts, result-slot
 * bounds) taken.  Batches over it then skip the host's pass over the pairs and the staging copy -- 25-40 us of a
 * 20 000-pair call -- and are otherwise exactly rhip_pairwise / _multi / _cardinality: same kernels, same results.
 * The list PINS its operand pools: rhip_pool_free of an operand is accepted and deferred until the last list over it
 * (and the last batch in flight) is gone.  If an operand is updated in place (or recycled) the list re-validates itself
 * at its next use.  A list is used with the context it was made on (anything else: RHIP_ERR_ARG).
 * rhip_pairlist_free while batches over the list are in flight is deferred to the last of them.
 * THE PLAN IS KEPT WITH THE LIST (round 6).  Which containers of a pair meet, the class every matched container pair
 * goes to, its result slot -- the output of the key merge (src/roaring.c:742-768), here the three planning kernels -- is a
 * pure function of the operand pools' directories, the pairs and the ops.  The first batch of a given (ops, form) over
 * a list leaves its class queues, candidate directory and section ranges with the list; every later one starts at its
 * class kernels (one small kernel restores the call's scratch).  Up to six plans per list, least recently used
 * replaced, never one a batch in flight reads; a pool that is updated in place or reloaded invalidates them (they are
 * keyed by the pools' generation).  Cost: device memory for the queues, sized by the same upper bounds as a batch's own
 * scratch, for the life of the list.  RHIP_PLAN_CACHE=0 in the environment switches it off. */
typedef struct rhip_pairlist_s rhip_pairlist_t;
rhip_pairlist_t *rhip_pairlist_create(rhip_ctx_t *ctx, rhip_pool_t *A, rhip_pool_t *B, size_t npairs,
                                      const uint32_t *lhs, const uint32_t *rhs);
/* all unordered pairs (i, j), i < j, of one pool, row by row: n (n - 1) / 2 pairs */
rhip_pairlist_t *rhip_pairlist_all_pairs(rhip_ctx_t *ctx, rhip_pool_t *A);
/* successive bitmaps (i, i + 1): n - 1 pairs */
rhip_pairlist_t *rhip_pairlist_successive(rhip_ctx_t *ctx, rhip_pool_t *A);
size_t rhip_pairlist_size(const rhip_pairlist_t *list);
/* the pairs themselves (lhs / rhs: rhip_pairlist_size entries each; either may be NULL) */
int rhip_pairlist_pairs(const rhip_pairlist_t *list, uint32_t *lhs, uint32_t *rhs);
void rhip_pairlist_free(rhip_pairlist_t *list);
/* Forgets the plans kept with the list (those no batch in flight reads) and releases their device memory; returns how
 * many.  The next batch of each (ops, form) plans afresh -- what bench.py's `ms_first_call` measures. */
int rhip_pairlist_drop_plans(rhip_pairlist_t *list);
/* rhip_pairwise_multi_begin / rhip_pairwise_multi / rhip_pairwise_cardinality over a prepared list (n_ops = 1: rhip_pairwise) */
rhip_batch_t *rhip_pairwise_list_begin(rhip_ctx_t *ctx, size_t n_ops, const rhip_op *ops, rhip_pairlist_t *list,
                                       rhip_pool_t *reuse);
rhip_pool_t *rhip_pairwise_list(rhip_ctx_t *ctx, size_t n_ops, const rhip_op *ops, rhip_pairlist_t *list,
                                rhip_pool_t *reuse);
int rhip_pairwise_list_cardinality(rhip_ctx_t *ctx, rhip_op op, rhip_pairlist_t *list, uint64_t *out);
/* roaring_bitmap_{and,or,xor,andnot}_cardinality (roaring.h:231,258,270,264;
 * src/roaring.c:3048-3107): nothing is materialised. */
int rhip_pairwise_cardinality(rhip_ctx_t *ctx, rhip_op op, rhip_pool_t *A, rhip_pool_t *B, size_t npairs,
                              const uint32_t *lhs, const uint32_t *rhs, uint64_t *out);
/* out[k] = pred(A[lhs[k]], B[rhs[k]]) as 0 / 1: roaring_bitmap_intersect (src/roaring.c:2998-3027), _is_subset
 * (:2151-2183, "A[lhs] is a subset of B[rhs]"), _is_strict_subset (:3172-3177), _equals (:2128-2149).  Evaluated
 * by the cardinality-only kernels (a batch has no use for the reference's early exit); nothing is materialised. */
int rhip_pairwise_predicate(rhip_ctx_t *ctx, rhip_pred pred, rhip_pool_t *A, rhip_pool_t *B, size_t npairs,
                            const uint32_t *lhs, const uint32_t *rhs, uint8_t *out);
/* roaring_bitmap_{and,or,xor,andnot}_inplace (roaring.h:280,295,326,348; src/roaring.c:812-875, 1063-1119,
 * 1200-1273, 1342-1434) for a batch, on device-resident handles: A[lhs[k]] <- op(A[lhs[k]], B[rhs[k]]), every
 * other bitmap of A unchanged; the handle A stays valid and now names the updated pool.  lhs must not repeat.
 * B may be A.  The results are spliced into A: their payload is appended to A's arena and only A's directory is
 * rebuilt -- bitmaps that are not updated are not copied (cost = the op + one copy of the RESULTS).  The replaced
 * bitmaps' old slots stay behind as garbage; once the arena has doubled since the last compaction the update goes
 * through rhip_pool_select instead, which rewrites the pool compactly. */
int rhip_pairwise_inplace(rhip_ctx_t *ctx, rhip_op op, rhip_pool_t *A, rhip_pool_t *B, size_t npairs,
                          const uint32_t *lhs, const uint32_t *rhs);

/* ---- reshaping pools on the device ---------------------------------------- */
/* A new pool of n bitmaps: bitmap i is a copy of srcs[src_pool[i]] bitmap src_bitmap[i].  Gathers operands and
 * results of earlier calls into one pool so that chained expressions ((a & b) | c ...) never leave HBM; all
 * sources must be of the same key width. */
rhip_pool_t *rhip_pool_select(rhip_ctx_t *ctx, size_t n_src, rhip_pool_t *const *srcs, size_t n,
                              const uint32_t *src_pool, const uint32_t *src_bitmap);
/* roaring_bitmap_run_optimize (roaring.h:621, src/roaring.c:1530-1546 -> convert_run_optimize,
 * src/containers/convert.c:217-321) applied to every bitmap: a new pool whose containers have the type the
 * reference would leave them in (byte-identical portable serialization). */
rhip_pool_t *rhip_pool_run_optimize(rhip_ctx_t *ctx, rhip_pool_t *pool);
/* roaring_bitmap_remove_run_compression (roaring.h:612, src/roaring.c:1564-1592): run containers become arrays
 * (cardinality <= 4096) or bitsets. */
rhip_pool_t *rhip_pool_remove_run_compression(rhip_ctx_t *ctx, rhip_pool_t *pool);
/* roaring_bitmap_flip (roaring.h:986, src/roaring.c:2289-2342) for every bitmap of a 32-bit pool: bitmap i is
 * negated on [starts[i], ends[i]) with the reference's own argument handling (start >= end, or a start beyond
 * 2^32: plain copy; both ends are truncated to 32 bits as roaring_bitmap_flip does) and its container typing
 * (container_not_range / container_not, containers.h:2009-2073; container_range_of_ones where the source has no
 * container under a key).  64-bit pools: roaring64_bitmap_flip (roaring64.h:525, src/roaring64.c:2007-2074) --
 * [min, max) over the 64-bit universe, a copy when min >= max; every 48-bit key of the range gets a container
 * (full ones where the bitmap had none), so a range may span at most 2^32 - 16 containers (RHIP_ERR_ARG beyond).
 * Returns a new pool; byte-identical portable serialization. */
rhip_pool_t *rhip_pool_flip(rhip_ctx_t *ctx, rhip_pool_t *pool, const uint64_t *starts, const uint64_t *ends);

/* ---- many-way aggregation ------------------------------------------------ */
/* roaring_bitmap_or_many (roaring.h:304, src/roaring.c:775-790) /
 * roaring_bitmap_xor_many (roaring.h:334, src/roaring.c:795-809) over
 * pool[ids[0..n)] (ids == NULL: the whole pool, in order).  Returns a pool
 * holding ONE bitmap.  n == 0 gives an empty bitmap, n == 1 a copy.
 * Parity level: BYTE-IDENTICAL to the reference, container types included.  Both functions are fixed left folds whose
 * order-dependent typing is replayed on the device: the run-vs-bitset choice of a FULL union under or_many; under
 * xor_many the whole fold of every key that has a run member (a run accumulator survives R ^ R and R ^ small array, an
 * accumulator that empties is removed and re-cloned: src/roaring.c:2684-2843, containers/mixed_xor.c).  Keys without run
 * members -- and the partial / sharded forms below -- are typed by cardinality, which is what the fold gives there. */
rhip_pool_t *rhip_or_many(rhip_ctx_t *ctx, rhip_pool_t *pool, size_t n, const uint32_t *ids);
rhip_pool_t *rhip_xor_many(rhip_ctx_t *ctx, rhip_pool_t *pool, size_t n, const uint32_t *ids);
/* roaring_bitmap_or_many_heap (roaring.h:312, src/roaring_priority_queue.c:200-247) over pool[ids[0..n)], BYTE-IDENTICAL to the
 * reference: the same set as rhip_or_many, with the container types the reference's size-ordered tournament leaves (lazy
 * unions without early bitset conversion: arrays stay arrays to 1024 values, run | array stays a run, run | run is typed
 * by size at every step; ties between equal sizes are broken by heap position, as the reference's heap breaks them).
 * The tournament is n - 1 sequentially dependent merges, each needing the serialized size of the previous result: the
 * heap runs on the host (restated move for move), every merge is three small launches and one wait -- ~40 us per step
 * plus the unions.  EXACT, NOT FAST: rhip_or_many is the throughput path (one wait in all); use this one where the heap's
 * bytes matter.  32-bit pools (roaring64 has no heap union).  Scratch: one 8-byte record per key of the key space and live
 * temporary, one 8 KiB image per merged container (RHIP_ERR_ALLOC above half of the free device memory). */
rhip_pool_t *rhip_or_many_heap(rhip_ctx_t *ctx, rhip_pool_t *pool, size_t n, const uint32_t *ids);

/* Multi-GPU or_many / xor_many building blocks.  Stage 1 on each
 * rank: reduce the local shard to one UNCOMPRESSED 1024-word chunk per distinct
 * key (no cardinality, no typing).  The caller exchanges chunks between ranks
 * (RCCL all-to-all to the key-range owner) and calls stage 2 on the owner. */
typedef struct rhip_partials_s {
    uint64_t n_keys;       /* distinct keys in this shard */
    uint64_t *d_keys;      /* device: [n_keys] ascending */
    uint64_t *d_words;     /* device: [n_keys][1024] */
    uint64_t max_key;      /* d_keys[n_keys - 1] (0 when n_keys == 0): lets the caller pick a fixed-shape exchange */
    uint64_t capacity;     /* internal: chunks the buffers can hold (they are recycled by rhip_partials_free) */
} rhip_partials_t;
int rhip_many_partials(rhip_ctx_t *ctx, rhip_op op /* RHIP_OR | RHIP_XOR */, rhip_pool_t *pool, size_t n,
                       const uint32_t *ids, rhip_partials_t *out);
/* Returns the chunk buffers to the context, which keeps the largest pair for its next rhip_many_partials instead of
 * freeing it.  Reuse is protected by STREAM ORDER only: whatever read d_keys / d_words must have been enqueued on the
 * context's stream (rhip_ctx_stream) -- or have completed -- before this call; a consumer on another stream must be
 * synchronised first. */
void rhip_partials_free(rhip_ctx_t *ctx, rhip_partials_t *p);
/* Stage 2: n_chunks (key, 1024-word chunk) records in device memory, any order,
 * duplicates allowed: chunks with equal keys are combined with op, then every
 * key is canonicalised (card <= 4096 -> array, else bitset; empty dropped) into
 * a pool holding ONE bitmap. */
rhip_pool_t *rhip_many_finalize(rhip_ctx_t *ctx, rhip_op op, int is64, uint64_t n_chunks, const uint64_t *d_keys,
                                const uint64_t *d_words);

/* The same two stages for the DENSE exchange (every rank sees most keys and the key space is small, e.g. the 4096
 * keys of BASELINE config C4), with NO host wait between the start of stage 1 and the end of stage 3:
 *  - rhip_many_partials_dense enqueues stage 1 and returns at once.  d_table (caller-owned device memory,
 *    world * B * 1024 words, B = ceil(key_space / world)) is zero-filled and the chunk of key k is written to row
 *    (k % world) * B + k / world: block d of the table (rows [d B, (d + 1) B)) is what rank d must receive, so ONE
 *    fixed-shape all-to-all (enqueued on the context's stream) is the whole exchange.  A key >= key_space is an
 *    error that the owner's rhip_many_finalize_dense on this context reports (RHIP_ERR_ARG).
 *  - rhip_many_finalize_dense: d_table now holds world blocks of keys_per_rank rows, block s = what source rank s
 *    sent (all-zero rows where it never saw the key); row s * keys_per_rank + j belongs to key rank + world * j.  The
 *    rows of a key are combined with op and canonicalised (card <= 4096 -> array, else bitset; empty dropped) -- no
 *    gather, no sort: the table's shape is the grouping.  The ONE host wait of the pipeline is at the end of this call. */
int rhip_many_partials_dense(rhip_ctx_t *ctx, rhip_op op, rhip_pool_t *pool, size_t n, const uint32_t *ids,
                             uint64_t key_space, uint32_t world, uint64_t *d_table);
rhip_pool_t *rhip_many_finalize_dense(rhip_ctx_t *ctx, rhip_op op, int is64, uint32_t world, uint32_t rank,
                                      uint64_t keys_per_rank, const uint64_t *d_table);

/* The whole sharded aggregation in one call, the exchange over RCCL included -- roaring_bitmap_or_many (roaring.h:304,
 * src/roaring.c:775-790) / Roaring64Map::fastunion (cpp/roaring/roaring64map.hh:1549-1670) over bitmaps spread across
 * the GPUs of a node, one process (and one rhip context) per GPU.  Every rank of the communicator calls it with ITS
 * bitmaps (`local`, or the selection ids[0..n) of it; ids == NULL: all of them), the same op and the same key_space.
 *   nccl_comm   an ncclComm_t of the caller (rccl.h:36) whose device is the context's; world and rank are read from it.
 *               librccl is opened with dlopen at the first call (RHIP_RCCL_LIB overrides the name): this library has
 *               no link-time dependency on it, and every other entry point works without it.
 *   key_space   > 0: an exclusive upper bound of the container keys on EVERY rank (e.g. 4096): the dense exchange --
 *               stage 1 writes the [world x ceil(key_space / world)] chunk table, ONE ncclAllToAll on the context's
 *               stream moves it, stage 3 combines the rows; one host wait, at the very end.  A larger key anywhere
 *               fails the call (RHIP_ERR_ARG).  0: the sparse exchange, any key width (roaring64 pools): every
 *               rank counts and packs its chunks by owner on the device, the world x world count matrix is all-gathered
 *               (one small collective, one host wait: send / receive counts are host arguments) and the chunks leave
 *               in one group of ncclSend / ncclRecv.  At most 64 ranks.
 *   owned       out: this rank's share of the result, a one-bitmap pool holding the container keys with
 *               key % world == rank (the shares are disjoint: their serialized forms concatenate by key).
 * Everything is enqueued on the context's stream (rhip_ctx_stream); the call returns when the share is complete.
 * RCCL has no OR / XOR reduction and a ring all-reduce of 8 KiB chunks would be bound by one xGMI link: the
 * personalised all-to-all keeps all point-to-point links busy.  Returns RHIP_OK or an error code.
 * Failure of ONE rank: a rank whose local stage fails (a bad id, a key >= key_space ...) still takes part in every
 * collective of the call -- contributing an empty share -- and returns its error afterwards; its peers complete with
 * what they received and are not left waiting.  Null arguments, a missing librccl and a failure to allocate the exchange
 * buffers are reported BEFORE the first collective: the group then has to be resolved by the caller, as in any RCCL program. */
int rhip_many_sharded(rhip_ctx_t *ctx, void *nccl_comm, rhip_op op, rhip_pool_t *local, size_t n, const uint32_t *ids,
                      uint64_t key_space, rhip_pool_t **owned);

/* ---- measurement hooks (bench.py) --------------------------------------- */
/* algorithmic bytes and matched container pairs of the last
 * rhip_pairwise / rhip_pairwise_cardinality / rhip_or_many call on ctx */
typedef struct rhip_stats_s {
    uint64_t matched_pairs;     /* container pairs sent to a set-op kernel */
    uint64_t passthrough;       /* containers copied unchanged */
    uint64_t bytes_in;          /* payload bytes read (matched + pass-through) */
    uint64_t bytes_out;         /* payload bytes of result containers */
    uint64_t n_bitset_pairs;    /* matched pairs handled by the bitset x bitset kernel */
    uint64_t result_containers; /* non-empty result containers */
    float ms_bitset_kernel;     /* HIP-event time of the bitset x bitset kernel launch, ms */
    float ms_total;             /* HIP-event time of the whole call on the stream, ms */
} rhip_stats_t;
int rhip_last_stats(rhip_ctx_t *ctx, rhip_stats_t *out);
/* enable/disable HIP-event timing of kernel launches (adds two event records per call) */
void rhip_ctx_set_timing(rhip_ctx_t *ctx, int enabled);
/* Per-kernel split of a pairwise batch's algorithmic bytes (the vocabulary of roaring_bitmap_statistics,
 * src/roaring.c:381-419, applied to the work instead of to a bitmap): for every class kernel, the container pairs
 * (pass-through containers for k_copy) it processed, the payload bytes of their operands and of their results.
 * Opt-in (rhip_ctx_set_class_stats): costs one more kernel and one more wait per rhip_pairwise / rhip_pairwise_end.
 * rhip_last_class_stats fills out[0 .. min(capacity, n)) for the last batch ended and returns n, the number of
 * classes. */
typedef struct rhip_class_stats_s {
    const char *kernel;  /* static string, e.g. "k_filter" */
    uint64_t items, bytes_in, bytes_out;
} rhip_class_stats_t;
void rhip_ctx_set_class_stats(rhip_ctx_t *ctx, int enabled);
int rhip_last_class_stats(rhip_ctx_t *ctx, rhip_class_stats_t *out, int capacity);
/* Diagnostics: host time of rhip_pairwise by phase, microseconds accumulated since the last reset:
 * [0] passes over the pair list, [1] scratch sizing + host-to-device copy of the batch description,
 * [2] planning launches, [3] class + tail launches, [4] wait for completion, [5] result bookkeeping.
 * The wait polls a completion word the last kernel writes into pinned memory; RHIP_SPIN_WAIT=0 in the environment
 * (read at rhip_ctx_create) makes it block on the stream instead. */
int rhip_debug_host_clock(rhip_ctx_t *ctx, double out_us[8], int reset);
/* Large result arenas are placed by measurement (see rhip_pairwise_begin).  This returns the probe rates (GB/s) of the
 * context's last placement -- single chunks and compositions of stage 0, the candidate allocations in the order they were
 * made, the positions of the address range in the order they were visited -- and how many there were. */
int rhip_debug_last_placement(rhip_ctx_t *ctx, float *out_gbps, int capacity);
/* Batches of this context whose flag join (the one-wave gate in front of the tail kernel of a forked batch) gave up --
 * HIP does not promise that kernels of different streams run side by side -- and that were then finished through the
 * fallback: the auxiliary streams waited for the ordinary way, the tail run again.  Such a batch returns the same result
 * as any other; the reference's functions cannot fail for scheduling reasons (roaring.h:102-113) and neither do these. */
unsigned long long rhip_debug_join_recovered(rhip_ctx_t *ctx);
/* Tests: after `skip` more device allocations of the library, the next `count` fail as if the device were out of memory. */
void rhip_debug_fail_allocs(int skip, int count);
/* 1 if the last batch begun on the context took its plan from its pair list's cache (no planning kernels ran), else 0 */
int rhip_debug_plan_cached(rhip_ctx_t *ctx);
/* Releases the spare result arenas the context keeps from its placement searches (see rhip_pairwise_begin); returns the
 * bytes released.  Pools and batches are untouched. */
unsigned long long rhip_ctx_trim(rhip_ctx_t *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ROARING_HIP_H */
