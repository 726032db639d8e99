/*
 * roaring_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, scalar, single-threaded restatement of the CRoaring 5.1.0 hot
 * path (pairwise and/or/xor/andnot, and_cardinality, get_cardinality,
 * or_many / xor_many, portable (de)serialization, 64-bit variants).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check the HIP engine bit-for-bit.  Nothing under
 * croaring_amd/ may include, link or call it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against (1) the reference's own golden vectors (Java-produced
 * bitmapwithruns.bin / bitmapwithoutruns.bin, the closed-form cardinalities
 * of toplevel_unit.c / mixed_container_unit.c), (2) fixtures produced by the
 * real reference library built from /root/reference (oracle/_ref, recipe in
 * oracle/Makefile) and committed under tests/golden/, and (3) when
 * oracle/_ref/libcroaring_ref.so is present, the reference itself on
 * randomised inputs (byte-identical portable serialization for every
 * pairwise op).
 *
 * Citations are relative to /root/reference.
 */
#ifndef ROARING_ORACLE_H
#define ROARING_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { OC_BITSET = 1, OC_ARRAY = 2, OC_RUN = 3 }; /* containers.h:48-53 typecodes */
enum { OC_AND = 0, OC_OR = 1, OC_XOR = 2, OC_ANDNOT = 3 };

/* One container. `card` is the cardinality (for bitsets -1 means "unknown",
 * bitset.h:42); for runs `nruns` is the number of {value,length} pairs. */
typedef struct oc_container_s {
    uint8_t type;
    int32_t card;
    int32_t nruns;
    void *data; /* bitset: uint64_t[1024]; array: uint16_t[card]; run: uint16_t[2*nruns] */
} oc_container_t;

/* roaring_array_t restated (roaring_types.h:61-68): sorted keys + containers */
typedef struct oc_bitmap_s {
    int32_t n;
    int32_t cap;
    uint16_t *keys;
    oc_container_t *c;
} oc_bitmap_t;

/* 64-bit bitmap: sorted high-32 buckets each holding a 32-bit bitmap
 * (the shape of the portable 64-bit format, roaring64.c:2323-2393). */
typedef struct oc_bitmap64_s {
    int64_t n;
    int64_t cap;
    uint32_t *high;
    oc_bitmap_t **bm;
} oc_bitmap64_t;

/* ---- lifecycle / construction ---- */
oc_bitmap_t *oc_create(void);
void oc_free(oc_bitmap_t *b);
oc_bitmap_t *oc_copy(const oc_bitmap_t *b);
/* sorted, strictly increasing values (roaring_bitmap_of_ptr on sorted input) */
oc_bitmap_t *oc_from_sorted(const uint32_t *vals, size_t n);
/* roaring_bitmap_run_optimize, roaring.c:1530-1546 */
int oc_run_optimize(oc_bitmap_t *b);
int oc_remove_run_compression(oc_bitmap_t *b);
oc_bitmap_t *oc_flip(const oc_bitmap_t *x, uint64_t range_start, uint64_t range_end);
int oc_intersect(const oc_bitmap_t *a, const oc_bitmap_t *b);
int oc_is_subset(const oc_bitmap_t *a, const oc_bitmap_t *b);
int oc_is_strict_subset(const oc_bitmap_t *a, const oc_bitmap_t *b);

/* ---- portable format, roaring_array.c:445-531, 633-813 ---- */
oc_bitmap_t *oc_deserialize(const char *buf, size_t maxbytes);
size_t oc_size_in_bytes(const oc_bitmap_t *b);
size_t oc_serialize(const oc_bitmap_t *b, char *buf);

/* ---- frozen format, roaring.c:3176-3457 (frozen_view restated as a copy: no alignment requirement) ---- */
size_t oc_frozen_size_in_bytes(const oc_bitmap_t *b);
size_t oc_frozen_serialize(const oc_bitmap_t *b, char *buf);
oc_bitmap_t *oc_frozen_deserialize(const char *buf, size_t length);

/* ---- hot path ---- */
oc_bitmap_t *oc_op(int op, const oc_bitmap_t *a, const oc_bitmap_t *b);
oc_bitmap_t *oc_and(const oc_bitmap_t *a, const oc_bitmap_t *b);
oc_bitmap_t *oc_or(const oc_bitmap_t *a, const oc_bitmap_t *b);
oc_bitmap_t *oc_xor(const oc_bitmap_t *a, const oc_bitmap_t *b);
oc_bitmap_t *oc_andnot(const oc_bitmap_t *a, const oc_bitmap_t *b);
uint64_t oc_get_cardinality(const oc_bitmap_t *b);
uint64_t oc_and_cardinality(const oc_bitmap_t *a, const oc_bitmap_t *b);
uint64_t oc_op_cardinality(int op, const oc_bitmap_t *a, const oc_bitmap_t *b);
oc_bitmap_t *oc_or_many(size_t n, const oc_bitmap_t **x);
/* roaring_bitmap_or_many_heap (roaring_priority_queue.c:200-247): byte-identical container types */
oc_bitmap_t *oc_or_many_heap(size_t n, const oc_bitmap_t **x);
oc_bitmap_t *oc_xor_many(size_t n, const oc_bitmap_t **x);

/* ---- checks / decoding ---- */
/* roaring_bitmap_internal_validate restated (roaring.c:454-523); 1 = valid */
int oc_validate(const oc_bitmap_t *b);
/* decode to sorted uint32 (out must hold oc_get_cardinality values) */
void oc_to_uint32(const oc_bitmap_t *b, uint32_t *out);
int oc_equals(const oc_bitmap_t *a, const oc_bitmap_t *b); /* set equality */
/* per-type container counts: out[0]=bitset out[1]=array out[2]=run */
void oc_type_counts(const oc_bitmap_t *b, int64_t out[3]);

/* ---- 64-bit (roaring64.c:1332-1373, 1541-1593, 1663-1720, 1809-1861) ---- */
oc_bitmap64_t *oc64_deserialize(const char *buf, size_t maxbytes);
size_t oc64_size_in_bytes(const oc_bitmap64_t *b);
size_t oc64_serialize(const oc_bitmap64_t *b, char *buf);
oc_bitmap64_t *oc64_from_sorted(const uint64_t *vals, size_t n);
int oc64_run_optimize(oc_bitmap64_t *b);
oc_bitmap64_t *oc64_op(int op, const oc_bitmap64_t *a, const oc_bitmap64_t *b);
uint64_t oc64_get_cardinality(const oc_bitmap64_t *b);
oc_bitmap64_t *oc64_or_many(size_t n, const oc_bitmap64_t **x);
oc_bitmap64_t *oc64_flip(const oc_bitmap64_t *x, uint64_t min, uint64_t max); /* roaring64_bitmap_flip */
void oc64_free(oc_bitmap64_t *b);

#ifdef __cplusplus
}
#endif
#endif
