/* TEST INFRASTRUCTURE ONLY -- force-included (-include) when compiling the REFERENCE library for the
 * drop-in acceptance harness: the 19 hot-path public symbols are renamed to ref_*, so the
 * public names can be provided by libroaring_hip.so while ref_* stays callable side by side. */
#define roaring_bitmap_and ref_roaring_bitmap_and
#define roaring_bitmap_or ref_roaring_bitmap_or
#define roaring_bitmap_xor ref_roaring_bitmap_xor
#define roaring_bitmap_andnot ref_roaring_bitmap_andnot
#define roaring_bitmap_and_inplace ref_roaring_bitmap_and_inplace
#define roaring_bitmap_or_inplace ref_roaring_bitmap_or_inplace
#define roaring_bitmap_xor_inplace ref_roaring_bitmap_xor_inplace
#define roaring_bitmap_andnot_inplace ref_roaring_bitmap_andnot_inplace
#define roaring_bitmap_and_cardinality ref_roaring_bitmap_and_cardinality
#define roaring_bitmap_or_cardinality ref_roaring_bitmap_or_cardinality
#define roaring_bitmap_xor_cardinality ref_roaring_bitmap_xor_cardinality
#define roaring_bitmap_andnot_cardinality ref_roaring_bitmap_andnot_cardinality
#define roaring_bitmap_or_many ref_roaring_bitmap_or_many
#define roaring_bitmap_or_many_heap ref_roaring_bitmap_or_many_heap
#define roaring_bitmap_xor_many ref_roaring_bitmap_xor_many
