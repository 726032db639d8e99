/*
 * roaring_hip_compat.h -- the CRoaring-named, per-call drop-in entry points of libroaring_hip.so.
 *
 * These are the exact C symbols (names, signatures, ownership and error conventions) of the
 * reference's hot path, include/roaring/roaring.h in CRoaring 5.1.0:
 *
 *   roaring_bitmap_and            roaring.h:225    roaring_bitmap_and_inplace       roaring.h:280
 *   roaring_bitmap_or             roaring.h:288    roaring_bitmap_or_inplace        roaring.h:295
 *   roaring_bitmap_xor            roaring.h:320    roaring_bitmap_xor_inplace       roaring.h:326
 *   roaring_bitmap_andnot         roaring.h:342    roaring_bitmap_andnot_inplace    roaring.h:348
 *   roaring_bitmap_and_cardinality roaring.h:231   roaring_bitmap_or_cardinality    roaring.h:258
 *   roaring_bitmap_andnot_cardinality roaring.h:264  roaring_bitmap_xor_cardinality roaring.h:270
 *   roaring_bitmap_or_many        roaring.h:304    roaring_bitmap_or_many_heap      roaring.h:312
 *   roaring_bitmap_xor_many       roaring.h:334
 *   roaring_bitmap_lazy_or(_inplace) roaring.h:932,944   roaring_bitmap_lazy_xor(_inplace) roaring.h:968,976
 *   roaring_bitmap_repair_after_lazy roaring.h:954
 *
 * They operate on the reference's own host structs (roaring_bitmap_t / roaring_array_t and the three
 * container structs are part of CRoaring's ABI, include/roaring/roaring_types.h:61-68,
 * containers/bitset.h:40-48, array.h:46-50, run.h:69-73): operands are read in place, every result
 * is a fresh caller-owned roaring_bitmap_t allocated with malloc / posix_memalign exactly as the
 * reference allocates, so the reference's roaring_bitmap_free (and every other reference function)
 * works on it.  Results carry COW = cow(x1) || cow(x2) (src/roaring.c:738); shared containers in the
 * operands are read through (containers.h:71-152) and never mutated.
 *
 * Each call uploads its operands, runs the batched device pipeline with a batch of one, and
 * downloads the result: correct, but latency-bound.  Throughput users bind the batched
 * API in roaring_hip.h instead.  A process-wide context on the current HIP device is created on
 * first use; without a device pointer-returning functions return NULL (roaring.h:216-226), the
 * *_cardinality functions return UINT64_MAX.  The void in-place / repair functions have no error channel: when a
 * device allocation fails inside one, the drop-ins give back what they hold themselves (the recycled operand / result
 * pools of every idle lane, the spare result arenas of their contexts) and the call is made ONCE MORE; only a failure
 * that persists prints rhip_last_error() to stderr and abort()s -- rather than silently leaving x1 unchanged.
 * Parity: EVERY symbol below returns the reference's bytes (container types included).  The three many-way functions
 * have order-dependent typing, which is replayed: roaring_bitmap_or_many's full-union choice and roaring_bitmap_xor_many's
 * fold on the device, roaring_bitmap_or_many_heap's size-ordered tournament (src/roaring_priority_queue.c:200-247) step by
 * step -- its heap on the host, every merge on the device (rhip_or_many_heap: exact, n - 1 dependent steps of ~40 us; should
 * its scratch exceed half of the free device memory the drop-in says so on stderr, once, and returns the same SET with
 * roaring_bitmap_or_many's types).
 *
 * How a program uses it: keep including <roaring/roaring.h>, link libroaring_hip.so BEFORE libroaring
 * (or build libroaring with these symbols renamed): these symbols then resolve here,
 * everything else (add, contains, iterators, serialization, free ...) stays the reference's.
 */
#ifndef ROARING_HIP_COMPAT_H
#define ROARING_HIP_COMPAT_H

#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
#define _Bool bool
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* opaque here; the layout is the reference's (a program normally sees it through roaring.h) */
typedef struct roaring_bitmap_s roaring_bitmap_t;

roaring_bitmap_t *roaring_bitmap_and(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
roaring_bitmap_t *roaring_bitmap_or(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
roaring_bitmap_t *roaring_bitmap_xor(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
roaring_bitmap_t *roaring_bitmap_andnot(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);

void roaring_bitmap_and_inplace(roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
void roaring_bitmap_or_inplace(roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
void roaring_bitmap_xor_inplace(roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
void roaring_bitmap_andnot_inplace(roaring_bitmap_t *r1, const roaring_bitmap_t *r2);

uint64_t roaring_bitmap_and_cardinality(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
uint64_t roaring_bitmap_or_cardinality(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
uint64_t roaring_bitmap_andnot_cardinality(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
uint64_t roaring_bitmap_xor_cardinality(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);

roaring_bitmap_t *roaring_bitmap_or_many(size_t number, const roaring_bitmap_t **rs);
roaring_bitmap_t *roaring_bitmap_or_many_heap(uint32_t number, const roaring_bitmap_t **rs);
roaring_bitmap_t *roaring_bitmap_xor_many(size_t number, const roaring_bitmap_t **rs);

/* lazy family, roaring.h:932-978: eager on the device (results are always canonical);
 * repair_after_lazy re-canonicalises any bitmap (also one left unrepaired by the reference's lazy ops) */
roaring_bitmap_t *roaring_bitmap_lazy_or(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2, const _Bool bitsetconversion);
void roaring_bitmap_lazy_or_inplace(roaring_bitmap_t *r1, const roaring_bitmap_t *r2, const _Bool bitsetconversion);
roaring_bitmap_t *roaring_bitmap_lazy_xor(const roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
void roaring_bitmap_lazy_xor_inplace(roaring_bitmap_t *r1, const roaring_bitmap_t *r2);
void roaring_bitmap_repair_after_lazy(roaring_bitmap_t *r1);

/* ---- 64-bit (include/roaring/roaring64.h:423-531) -----------------------------------------------------------
 * roaring64_bitmap_t is an ART with a private node layout, so these go through the portable format: operands are
 * serialized with the REFERENCE's roaring64_bitmap_portable_serialize (resolved with dlsym(RTLD_DEFAULT) from the
 * program this library is loaded into -- link the program with -rdynamic if libroaring is linked statically),
 * results are built by its roaring64_bitmap_portable_deserialize_safe, in-place forms use roaring64_bitmap_overwrite.
 * Results are byte-identical (portable serialization) to the reference's. */
typedef struct roaring64_bitmap_s roaring64_bitmap_t;
roaring64_bitmap_t *roaring64_bitmap_and(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);        /* roaring64.h:423 */
roaring64_bitmap_t *roaring64_bitmap_or(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);         /* :468 */
roaring64_bitmap_t *roaring64_bitmap_xor(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);        /* :488 */
roaring64_bitmap_t *roaring64_bitmap_andnot(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);     /* :509 */
void roaring64_bitmap_and_inplace(roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);                      /* :439 */
void roaring64_bitmap_or_inplace(roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);                       /* :480 */
void roaring64_bitmap_xor_inplace(roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);                      /* :501 */
void roaring64_bitmap_andnot_inplace(roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);                   /* :522 */
uint64_t roaring64_bitmap_and_cardinality(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);        /* :429 */
uint64_t roaring64_bitmap_or_cardinality(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);         /* :474 */
uint64_t roaring64_bitmap_xor_cardinality(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);        /* :494 */
uint64_t roaring64_bitmap_andnot_cardinality(const roaring64_bitmap_t *r1, const roaring64_bitmap_t *r2);     /* :515 */
roaring64_bitmap_t *roaring64_bitmap_flip(const roaring64_bitmap_t *r, uint64_t min, uint64_t max);           /* :531 */

/* ---- the reference's allocator hook (include/roaring/memory.h:29-38, src/memory.c:44-46) --------------------------
 * The drop-ins above allocate every result through the reference's own roaring_malloc / roaring_aligned_malloc and
 * release containers through roaring_free / roaring_aligned_free (resolved from the host program), so a memory hook the
 * program installed with roaring_init_memory_hook is honoured in both directions.
 * rhip_install_pinned_allocator installs a hook of this library's: a page-locked, device-mapped arena of `arena_bytes`
 * (hipHostMalloc; 0 = 256 MiB) with size-class free lists -- every container the reference allocates afterwards is a
 * valid DMA source / target; requests above 24 KiB, and all of them once the arena is full, fall through to the C
 * library.  Call it before the first bitmap is created (blocks allocated earlier are still freed correctly: the hook
 * tells its own blocks from the C library's by address).  0 on success; RHIP_ERR_ARG when the reference's
 * roaring_init_memory_hook is not visible in the process, RHIP_ERR_ALLOC without a HIP device.
 * rhip_pinned_allocator_stats: {arena bytes, bytes in use, blocks served from the arena, requests passed to the C library}. */
int rhip_install_pinned_allocator(size_t arena_bytes);
int rhip_pinned_allocator_stats(unsigned long long out[4]);

/* Deliberately NOT drop-in symbols (they stay the reference's): roaring_bitmap_get_cardinality (roaring.h:537),
 * roaring_bitmap_portable_deserialize_safe / _serialize / _size_in_bytes (roaring.h:746, 791, 807), point operations,
 * iterators: host-trivial or latency-bound per call.  Their batched device forms are rhip_pool_cardinalities,
 * rhip_pool_from_portable / _from_blob, rhip_pool_portable_serialize(_many) in roaring_hip.h. */

#ifdef __cplusplus
}
#endif
#endif /* ROARING_HIP_COMPAT_H */
