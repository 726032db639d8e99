/*
 * roaring_oracle.c -- TEST INFRASTRUCTURE ONLY (see roaring_oracle.h).
 *
 * Scalar restatement of the CRoaring 5.1.0 set-operation hot path.  Each
 * function cites the reference function (file:line under /root/reference)
 * whose behaviour -- in particular whose RESULT CONTAINER TYPE -- it restates.
 * The code is written from the algorithm descriptions, not copied: it uses one
 * uniform container record, scalar loops only, and no SIMD dispatch.
 */
#include "roaring_oracle.h"

#include <stdlib.h>
#include <string.h>

#define WORDS 1024
#define MAXARR 4096 /* DEFAULT_MAX_SIZE, containers/array.h:38 */

/* ------------------------------------------------------------------ utils */
static uint64_t *words_new(void) { return (uint64_t *)calloc(WORDS, 8); }

static int popcnt_words(const uint64_t *w) {
    int c = 0;
    for (int i = 0; i < WORDS; i++) c += __builtin_popcountll(w[i]);
    return c;
}
static inline int w_get(const uint64_t *w, uint32_t v) { return (int)((w[v >> 6] >> (v & 63)) & 1); }
static inline void w_set(uint64_t *w, uint32_t v) { w[v >> 6] |= (uint64_t)1 << (v & 63); }

/* set/clear/flip [lo, hi] inclusive -- bitset_util.h:41-161 range helpers */
static void w_range(uint64_t *w, uint32_t lo, uint32_t hi, int mode) {
    for (uint32_t wi = lo >> 6; wi <= (hi >> 6); wi++) {
        uint32_t a = (wi == (lo >> 6)) ? (lo & 63) : 0;
        uint32_t b = (wi == (hi >> 6)) ? (hi & 63) : 63;
        uint64_t m = (b == 63 ? ~(uint64_t)0 : (((uint64_t)1 << (b + 1)) - 1)) & ~(((uint64_t)1 << a) - 1);
        if (mode == 0) w[wi] |= m;
        else if (mode == 1) w[wi] &= ~m;
        else w[wi] ^= m;
    }
}

static oc_container_t mk_bitset(uint64_t *w, int card) {
    oc_container_t c = {OC_BITSET, card, 0, w};
    return c;
}
static oc_container_t mk_array(uint16_t *a, int card) {
    oc_container_t c = {OC_ARRAY, card, 0, a};
    return c;
}
static oc_container_t mk_run(uint16_t *r, int nruns) {
    oc_container_t c = {OC_RUN, 0, nruns, r};
    int card = nruns;
    for (int i = 0; i < nruns; i++) card += r[2 * i + 1];
    c.card = card; /* run_container_cardinality, run.c:1077-1090 */
    return c;
}
static oc_container_t mk_empty(void) { return mk_array((uint16_t *)malloc(2), 0); }

static void c_free(oc_container_t *c) {
    free(c->data);
    c->data = NULL;
}

static size_t c_payload(const oc_container_t *c) {
    /* bitset.h:431-433 / array.h:173-175 / run.h:456 serialized sizes minus the run count prefix */
    if (c->type == OC_BITSET) return 8192;
    if (c->type == OC_ARRAY) return 2 * (size_t)c->card;
    return 4 * (size_t)c->nruns;
}

static oc_container_t c_clone(const oc_container_t *c) {
    oc_container_t r = *c;
    size_t n = c_payload(c);
    r.data = malloc(n ? n : 2);
    memcpy(r.data, c->data, n);
    return r;
}

static int run_is_full(const oc_container_t *c) { /* run.h:394-397 */
    const uint16_t *r = (const uint16_t *)c->data;
    return c->nruns == 1 && r[0] == 0 && r[1] == 0xFFFF;
}

/* --------------------------------------------------------- conversions */
/* array_container_from_bitset, convert.c:92-110 */
static oc_container_t array_from_words(const uint64_t *w, int card) {
    uint16_t *a = (uint16_t *)malloc(2 * (size_t)(card ? card : 1));
    int k = 0;
    for (int i = 0; i < WORDS; i++) {
        uint64_t x = w[i];
        while (x) {
            a[k++] = (uint16_t)(i * 64 + __builtin_ctzll(x));
            x &= x - 1;
        }
    }
    return mk_array(a, k);
}

/* bitset_container_from_array / _from_run, convert.c:22-60 */
static uint64_t *words_from(const oc_container_t *c) {
    uint64_t *w = words_new();
    if (c->type == OC_BITSET) {
        memcpy(w, c->data, 8192);
    } else if (c->type == OC_ARRAY) {
        const uint16_t *a = (const uint16_t *)c->data;
        for (int i = 0; i < c->card; i++) w_set(w, a[i]);
    } else {
        const uint16_t *r = (const uint16_t *)c->data;
        for (int i = 0; i < c->nruns; i++) w_range(w, r[2 * i], (uint32_t)r[2 * i] + r[2 * i + 1], 0);
    }
    return w;
}

/* array_container_from_run, convert.c:62-90 */
static oc_container_t array_from_run(const oc_container_t *c) {
    const uint16_t *r = (const uint16_t *)c->data;
    uint16_t *a = (uint16_t *)malloc(2 * (size_t)(c->card ? c->card : 1));
    int k = 0;
    for (int i = 0; i < c->nruns; i++)
        for (uint32_t v = r[2 * i]; v <= (uint32_t)r[2 * i] + r[2 * i + 1]; v++) a[k++] = (uint16_t)v;
    return mk_array(a, k);
}

/* bitset result typing shared by the "≤4096 ? array : bitset" rules
 * (mixed_xor.c:23-39, mixed_andnot.c:54-72, ...).  Takes ownership of w. */
static oc_container_t bitset_or_array(uint64_t *w) {
    int card = popcnt_words(w);
    if (card <= MAXARR) {
        oc_container_t a = array_from_words(w, card);
        free(w);
        return a;
    }
    return mk_bitset(w, card);
}

/* convert_run_to_efficient_container, convert.c:154-200.  Takes ownership. */
static oc_container_t run_to_efficient(oc_container_t run) {
    int32_t size_run = 2 + 4 * run.nruns; /* run.h:456 */
    int32_t card = run.card;
    int32_t size_arr = 2 * card; /* array.h:173 */
    int32_t min_non_run = 8192 < size_arr ? 8192 : size_arr;
    if (size_run <= min_non_run) return run;
    oc_container_t out;
    if (card <= MAXARR) {
        out = array_from_run(&run);
    } else {
        out = mk_bitset(words_from(&run), card);
    }
    c_free(&run);
    return out;
}

/* ------------------------------------------------------- array kernels */
/* array_container_intersection, array.c:288-325 (scalar: array_util.c intersect_uint16) */
static oc_container_t aa_and(const oc_container_t *x, const oc_container_t *y) {
    const uint16_t *a = (const uint16_t *)x->data, *b = (const uint16_t *)y->data;
    int na = x->card, nb = y->card, i = 0, j = 0, k = 0;
    uint16_t *o = (uint16_t *)malloc(2 * (size_t)((na < nb ? na : nb) + 1));
    while (i < na && j < nb) {
        if (a[i] < b[j]) i++;
        else if (a[i] > b[j]) j++;
        else { o[k++] = a[i]; i++; j++; }
    }
    return mk_array(o, k);
}
static int aa_and_card(const oc_container_t *x, const oc_container_t *y) {
    const uint16_t *a = (const uint16_t *)x->data, *b = (const uint16_t *)y->data;
    int na = x->card, nb = y->card, i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) i++;
        else if (a[i] > b[j]) j++;
        else { k++; i++; j++; }
    }
    return k;
}
/* union_uint16, array_util.c:1104-1151 */
static int merge_or(const uint16_t *a, int na, const uint16_t *b, int nb, uint16_t *o) {
    int i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) o[k++] = a[i++];
        else if (a[i] > b[j]) o[k++] = b[j++];
        else { o[k++] = a[i]; i++; j++; }
    }
    while (i < na) o[k++] = a[i++];
    while (j < nb) o[k++] = b[j++];
    return k;
}
/* xor_uint16, array_util.c:1198-1227 */
static int merge_xor(const uint16_t *a, int na, const uint16_t *b, int nb, uint16_t *o) {
    int i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) o[k++] = a[i++];
        else if (a[i] > b[j]) o[k++] = b[j++];
        else { i++; j++; }
    }
    while (i < na) o[k++] = a[i++];
    while (j < nb) o[k++] = b[j++];
    return k;
}
/* difference_uint16, array_util.c:1153-1196 */
static int merge_andnot(const uint16_t *a, int na, const uint16_t *b, int nb, uint16_t *o) {
    int i = 0, j = 0, k = 0;
    while (i < na) {
        while (j < nb && b[j] < a[i]) j++;
        if (j < nb && b[j] == a[i]) { i++; j++; }
        else o[k++] = a[i++];
    }
    return k;
}

/* array_array_container_union, mixed_union.c:162-191 */
static oc_container_t aa_or(const oc_container_t *x, const oc_container_t *y) {
    int total = x->card + y->card;
    if (total <= MAXARR) {
        uint16_t *o = (uint16_t *)malloc(2 * (size_t)(total + 1));
        int k = merge_or((const uint16_t *)x->data, x->card, (const uint16_t *)y->data, y->card, o);
        return mk_array(o, k);
    }
    uint64_t *w = words_from(x);
    const uint16_t *b = (const uint16_t *)y->data;
    for (int i = 0; i < y->card; i++) w_set(w, b[i]);
    return bitset_or_array(w);
}
/* array_array_container_xor, mixed_xor.c:196-219 */
static oc_container_t aa_xor(const oc_container_t *x, const oc_container_t *y) {
    int total = x->card + y->card;
    if (total <= MAXARR) {
        uint16_t *o = (uint16_t *)malloc(2 * (size_t)(total + 1));
        int k = merge_xor((const uint16_t *)x->data, x->card, (const uint16_t *)y->data, y->card, o);
        return mk_array(o, k);
    }
    uint64_t *w = words_from(x);
    const uint16_t *b = (const uint16_t *)y->data;
    for (int i = 0; i < y->card; i++) w[b[i] >> 6] ^= (uint64_t)1 << (b[i] & 63);
    return bitset_or_array(w);
}
/* array_array_container_andnot, mixed_andnot.c:463-468 */
static oc_container_t aa_andnot(const oc_container_t *x, const oc_container_t *y) {
    uint16_t *o = (uint16_t *)malloc(2 * (size_t)(x->card + 1));
    int k = merge_andnot((const uint16_t *)x->data, x->card, (const uint16_t *)y->data, y->card, o);
    return mk_array(o, k);
}

/* --------------------------------------------------------- run kernels */
typedef struct { uint16_t *r; int n; } runbuf;

/* run_container_append, run.h:331-345 (merging append) */
static void rb_append(runbuf *d, uint32_t v, uint32_t len) {
    if (d->n == 0) { d->r[0] = (uint16_t)v; d->r[1] = (uint16_t)len; d->n = 1; return; }
    uint32_t pv = d->r[2 * (d->n - 1)], pl = d->r[2 * (d->n - 1) + 1];
    uint32_t prevend = pv + pl;
    if (v > prevend + 1) {
        d->r[2 * d->n] = (uint16_t)v; d->r[2 * d->n + 1] = (uint16_t)len; d->n++;
    } else {
        uint32_t newend = v + len + 1;
        if (newend > prevend) d->r[2 * (d->n - 1) + 1] = (uint16_t)(newend - 1 - pv);
    }
}

/* run_container_smart_append_exclusive, run.c:824-867 */
static void rb_append_xor(runbuf *d, uint32_t start, uint32_t length) {
    uint16_t *last = d->n ? d->r + 2 * (d->n - 1) : NULL;
    int old_end = 0;
    if (!d->n || (int)start > (old_end = (int)last[0] + last[1] + 1)) {
        d->r[2 * d->n] = (uint16_t)start; d->r[2 * d->n + 1] = (uint16_t)length; d->n++;
        return;
    }
    if (old_end == (int)start) { last[1] = (uint16_t)(last[1] + length + 1); return; }
    int new_end = (int)start + (int)length + 1;
    if (start == last[0]) {
        if (new_end < old_end) { last[0] = (uint16_t)new_end; last[1] = (uint16_t)(old_end - new_end - 1); }
        else if (new_end > old_end) { last[0] = (uint16_t)old_end; last[1] = (uint16_t)(new_end - old_end - 1); }
        else d->n--;
        return;
    }
    last[1] = (uint16_t)(start - last[0] - 1);
    if (new_end < old_end) {
        d->r[2 * d->n] = (uint16_t)new_end; d->r[2 * d->n + 1] = (uint16_t)(old_end - new_end - 1); d->n++;
    } else if (new_end > old_end) {
        d->r[2 * d->n] = (uint16_t)old_end; d->r[2 * d->n + 1] = (uint16_t)(new_end - old_end - 1); d->n++;
    }
}

static runbuf rb_new(int cap) {
    runbuf d;
    d.r = (uint16_t *)malloc(4 * (size_t)(cap + 2));
    d.n = 0;
    return d;
}

/* run_container_union, run.c:231-283 */
static oc_container_t rr_union_raw(const oc_container_t *x, const oc_container_t *y) {
    if (run_is_full(x)) return c_clone(x);
    if (run_is_full(y)) return c_clone(y);
    const uint16_t *a = (const uint16_t *)x->data, *b = (const uint16_t *)y->data;
    runbuf d = rb_new(x->nruns + y->nruns);
    int i = 0, j = 0;
    while (i < x->nruns && j < y->nruns) {
        if (a[2 * i] <= b[2 * j]) { rb_append(&d, a[2 * i], a[2 * i + 1]); i++; }
        else { rb_append(&d, b[2 * j], b[2 * j + 1]); j++; }
    }
    while (j < y->nruns) { rb_append(&d, b[2 * j], b[2 * j + 1]); j++; }
    while (i < x->nruns) { rb_append(&d, a[2 * i], a[2 * i + 1]); i++; }
    return mk_run(d.r, d.n);
}

/* run_container_intersection, run.c:387-463 */
static oc_container_t rr_and_raw(const oc_container_t *x, const oc_container_t *y) {
    if (run_is_full(x)) return c_clone(y);
    if (run_is_full(y)) return c_clone(x);
    const uint16_t *a = (const uint16_t *)x->data, *b = (const uint16_t *)y->data;
    runbuf d = rb_new(x->nruns + y->nruns);
    int i = 0, j = 0;
    while (i < x->nruns && j < y->nruns) {
        int32_t s = a[2 * i], e = s + a[2 * i + 1] + 1;
        int32_t xs = b[2 * j], xe = xs + b[2 * j + 1] + 1;
        if (e <= xs) i++;
        else if (xe <= s) j++;
        else {
            int32_t ls = s > xs ? s : xs, ee;
            if (e == xe) { ee = e; i++; j++; }
            else if (e < xe) { ee = e; i++; }
            else { ee = xe; j++; }
            d.r[2 * d.n] = (uint16_t)ls; d.r[2 * d.n + 1] = (uint16_t)(ee - ls - 1); d.n++;
        }
    }
    return mk_run(d.r, d.n);
}

/* run_container_xor, run.c:348-383 */
static oc_container_t rr_xor_raw(const oc_container_t *x, const oc_container_t *y) {
    const uint16_t *a = (const uint16_t *)x->data, *b = (const uint16_t *)y->data;
    runbuf d = rb_new(x->nruns + y->nruns);
    int i = 0, j = 0;
    while (i < x->nruns && j < y->nruns) {
        if (a[2 * i] <= b[2 * j]) { rb_append_xor(&d, a[2 * i], a[2 * i + 1]); i++; }
        else { rb_append_xor(&d, b[2 * j], b[2 * j + 1]); j++; }
    }
    while (i < x->nruns) { rb_append_xor(&d, a[2 * i], a[2 * i + 1]); i++; }
    while (j < y->nruns) { rb_append_xor(&d, b[2 * j], b[2 * j + 1]); j++; }
    return mk_run(d.r, d.n);
}

/* run_container_andnot, run.c:575-633 */
static oc_container_t rr_andnot_raw(const oc_container_t *x, const oc_container_t *y) {
    const uint16_t *a = (const uint16_t *)x->data, *b = (const uint16_t *)y->data;
    runbuf d = rb_new(x->nruns + y->nruns);
    int i = 0, j = 0;
    int32_t s = a[0], e = s + a[1] + 1;
    int32_t s2 = y->nruns ? b[0] : 0, e2 = y->nruns ? s2 + b[1] + 1 : 0;
    while (i < x->nruns && j < y->nruns) {
        if (e <= s2) {
            d.r[2 * d.n] = (uint16_t)s; d.r[2 * d.n + 1] = (uint16_t)(e - s - 1); d.n++;
            i++;
            if (i < x->nruns) { s = a[2 * i]; e = s + a[2 * i + 1] + 1; }
        } else if (e2 <= s) {
            j++;
            if (j < y->nruns) { s2 = b[2 * j]; e2 = s2 + b[2 * j + 1] + 1; }
        } else {
            if (s < s2) { d.r[2 * d.n] = (uint16_t)s; d.r[2 * d.n + 1] = (uint16_t)(s2 - s - 1); d.n++; }
            if (e2 < e) s = e2;
            else {
                i++;
                if (i < x->nruns) { s = a[2 * i]; e = s + a[2 * i + 1] + 1; }
            }
        }
    }
    if (i < x->nruns) {
        d.r[2 * d.n] = (uint16_t)s; d.r[2 * d.n + 1] = (uint16_t)(e - s - 1); d.n++;
        i++;
        for (; i < x->nruns; i++) { d.r[2 * d.n] = a[2 * i]; d.r[2 * d.n + 1] = a[2 * i + 1]; d.n++; }
    }
    return mk_run(d.r, d.n);
}

/* ------------------------------------------------------- mixed kernels */
/* array_bitset_container_intersection / _andnot, mixed_intersection.c:19-46, mixed_andnot.c:24-39 */
static oc_container_t ab_filter(const oc_container_t *arr, const oc_container_t *bs, int keep_if_present) {
    const uint16_t *a = (const uint16_t *)arr->data;
    const uint64_t *w = (const uint64_t *)bs->data;
    uint16_t *o = (uint16_t *)malloc(2 * (size_t)(arr->card + 1));
    int k = 0;
    for (int i = 0; i < arr->card; i++)
        if (w_get(w, a[i]) == keep_if_present) o[k++] = a[i];
    return mk_array(o, k);
}

/* membership of v in a run list (binary search on starts) */
static int run_contains(const oc_container_t *run, uint32_t v) {
    const uint16_t *r = (const uint16_t *)run->data;
    int lo = 0, hi = run->nruns - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (r[2 * mid] <= v) lo = mid + 1;
        else hi = mid - 1;
    }
    if (hi < 0) return 0;
    return v <= (uint32_t)r[2 * hi] + r[2 * hi + 1];
}

/* array_run_container_intersection, mixed_intersection.c:73-111;
 * array_run_container_andnot, mixed_andnot.c:374-412 */
static oc_container_t ar_filter(const oc_container_t *arr, const oc_container_t *run, int keep_if_present) {
    const uint16_t *a = (const uint16_t *)arr->data;
    uint16_t *o = (uint16_t *)malloc(2 * (size_t)(arr->card + 1));
    int k = 0;
    for (int i = 0; i < arr->card; i++)
        if (run_contains(run, a[i]) == keep_if_present) o[k++] = a[i];
    return mk_array(o, k);
}

/* run_bitset_container_intersection, mixed_intersection.c:117-202 */
static oc_container_t rb_and(const oc_container_t *run, const oc_container_t *bs) {
    if (run_is_full(run)) return c_clone(bs);
    const uint16_t *r = (const uint16_t *)run->data;
    const uint64_t *w = (const uint64_t *)bs->data;
    if (run->card <= MAXARR) {
        uint16_t *o = (uint16_t *)malloc(2 * (size_t)(run->card + 1));
        int k = 0;
        for (int i = 0; i < run->nruns; i++)
            for (uint32_t v = r[2 * i]; v <= (uint32_t)r[2 * i] + r[2 * i + 1]; v++)
                if (w_get(w, v)) o[k++] = (uint16_t)v;
        return mk_array(o, k);
    }
    uint64_t *ow = words_new();
    memcpy(ow, w, 8192);
    uint32_t start = 0;
    for (int i = 0; i < run->nruns; i++) {
        uint32_t end = r[2 * i];
        if (end > start) w_range(ow, start, end - 1, 1);
        start = end + r[2 * i + 1] + 1;
    }
    if (start < 65536) w_range(ow, start, 65535, 1);
    return bitset_or_array(ow);
}

/* array_run_container_union, mixed_union.c:66-108 (result is a raw run) */
static oc_container_t ar_union_raw(const oc_container_t *arr, const oc_container_t *run) {
    if (run_is_full(run)) return c_clone(run);
    const uint16_t *a = (const uint16_t *)arr->data, *r = (const uint16_t *)run->data;
    runbuf d = rb_new(arr->card + run->nruns);
    int i = 0, j = 0;
    while (j < run->nruns && i < arr->card) {
        if (r[2 * j] <= a[i]) { rb_append(&d, r[2 * j], r[2 * j + 1]); j++; }
        else { rb_append(&d, a[i], 0); i++; }
    }
    while (i < arr->card) { rb_append(&d, a[i], 0); i++; }
    while (j < run->nruns) { rb_append(&d, r[2 * j], r[2 * j + 1]); j++; }
    return mk_run(d.r, d.n);
}

/* array_run_container_lazy_xor, mixed_xor.c:140-173 (raw run) */
static oc_container_t ar_xor_raw(const oc_container_t *arr, const oc_container_t *run) {
    const uint16_t *a = (const uint16_t *)arr->data, *r = (const uint16_t *)run->data;
    runbuf d = rb_new(arr->card + run->nruns);
    int i = 0, j = 0;
    while (j < run->nruns && i < arr->card) {
        if (r[2 * j] <= a[i]) { rb_append_xor(&d, r[2 * j], r[2 * j + 1]); j++; }
        else { rb_append_xor(&d, a[i], 0); i++; }
    }
    while (i < arr->card) { rb_append_xor(&d, a[i], 0); i++; }
    while (j < run->nruns) { rb_append_xor(&d, r[2 * j], r[2 * j + 1]); j++; }
    return mk_run(d.r, d.n);
}

/* array_run_container_xor, mixed_xor.c:104-138 */
static oc_container_t ar_xor(const oc_container_t *arr, const oc_container_t *run) {
    if (arr->card < 32) return run_to_efficient(ar_xor_raw(arr, run));
    if (run->card <= MAXARR) {
        oc_container_t tmp = array_from_run(run);
        oc_container_t out = aa_xor(&tmp, arr);
        c_free(&tmp);
        return out;
    }
    uint64_t *w = words_from(run);
    const uint16_t *a = (const uint16_t *)arr->data;
    for (int i = 0; i < arr->card; i++) w[a[i] >> 6] ^= (uint64_t)1 << (a[i] & 63);
    return bitset_or_array(w);
}

/* run_array_container_andnot, mixed_andnot.c:277-361 */
static oc_container_t ra_andnot(const oc_container_t *run, const oc_container_t *arr) {
    const uint16_t *r = (const uint16_t *)run->data, *a = (const uint16_t *)arr->data;
    if (run->card <= 32) {
        if (arr->card == 0) return c_clone(run);
        /* interval list minus points, kept as runs, then smallest-serialization typing */
        runbuf d = rb_new(run->card + arr->card);
        int j = 0;
        for (int i = 0; i < run->nruns; i++) {
            int32_t s = r[2 * i], e = s + r[2 * i + 1]; /* inclusive */
            while (j < arr->card && a[j] < s) j++;
            int32_t cur = s;
            while (j < arr->card && a[j] <= e) {
                if (a[j] > cur) { d.r[2 * d.n] = (uint16_t)cur; d.r[2 * d.n + 1] = (uint16_t)(a[j] - cur - 1); d.n++; }
                cur = a[j] + 1;
                j++;
            }
            if (cur <= e) { d.r[2 * d.n] = (uint16_t)cur; d.r[2 * d.n + 1] = (uint16_t)(e - cur); d.n++; }
        }
        return run_to_efficient(mk_run(d.r, d.n));
    }
    if (run->card <= MAXARR) {
        oc_container_t tmp = array_from_run(run);
        oc_container_t out = aa_andnot(&tmp, arr);
        c_free(&tmp);
        return out;
    }
    uint64_t *w = words_from(run);
    for (int i = 0; i < arr->card; i++) w[a[i] >> 6] &= ~((uint64_t)1 << (a[i] & 63));
    return bitset_or_array(w);
}

/* run_bitset_container_andnot, mixed_andnot.c:104-150 */
static oc_container_t rb_andnot(const oc_container_t *run, const oc_container_t *bs) {
    const uint16_t *r = (const uint16_t *)run->data;
    const uint64_t *w = (const uint64_t *)bs->data;
    if (run->card <= MAXARR) {
        uint16_t *o = (uint16_t *)malloc(2 * (size_t)(run->card + 1));
        int k = 0;
        for (int i = 0; i < run->nruns; i++)
            for (uint32_t v = r[2 * i]; v <= (uint32_t)r[2 * i] + r[2 * i + 1]; v++)
                if (!w_get(w, v)) o[k++] = (uint16_t)v;
        return mk_array(o, k);
    }
    uint64_t *rw = words_from(run);
    for (int i = 0; i < WORDS; i++) rw[i] &= ~w[i];
    return bitset_or_array(rw);
}

/* ------------------------------------------------ container dispatchers */
static void words_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    /* bitset.c:343-942 scalar forms; andnot = a & ~b (bitset.c:369) */
    for (int i = 0; i < WORDS; i++) {
        switch (op) {
            case OC_AND: o[i] = a[i] & b[i]; break;
            case OC_OR: o[i] = a[i] | b[i]; break;
            case OC_XOR: o[i] = a[i] ^ b[i]; break;
            default: o[i] = a[i] & ~b[i]; break;
        }
    }
}

/* container_and, containers.h:726-806 */
static oc_container_t c_and(const oc_container_t *x, const oc_container_t *y) {
    int t1 = x->type, t2 = y->type;
    if (t1 == OC_BITSET && t2 == OC_BITSET) { /* mixed_intersection.c:305-325 */
        uint64_t *w = words_new();
        words_op(OC_AND, (const uint64_t *)x->data, (const uint64_t *)y->data, w);
        return bitset_or_array(w);
    }
    if (t1 == OC_ARRAY && t2 == OC_ARRAY) return aa_and(x, y);
    if (t1 == OC_RUN && t2 == OC_RUN) return run_to_efficient(rr_and_raw(x, y));
    if (t1 == OC_BITSET && t2 == OC_ARRAY) return ab_filter(y, x, 1);
    if (t1 == OC_ARRAY && t2 == OC_BITSET) return ab_filter(x, y, 1);
    if (t1 == OC_BITSET && t2 == OC_RUN) return rb_and(y, x);
    if (t1 == OC_RUN && t2 == OC_BITSET) return rb_and(x, y);
    if (t1 == OC_ARRAY && t2 == OC_RUN) return run_is_full(y) ? c_clone(x) : ar_filter(x, y, 1);
    return run_is_full(x) ? c_clone(y) : ar_filter(y, x, 1);
}

/* run∪bitset, mixed_union.c:41-63 */
static oc_container_t rb_or(const oc_container_t *run, const oc_container_t *bs) {
    if (run_is_full(run)) return c_clone(run); /* containers.h:1056-1062 */
    uint64_t *w = words_new();
    memcpy(w, bs->data, 8192);
    const uint16_t *r = (const uint16_t *)run->data;
    for (int i = 0; i < run->nruns; i++) w_range(w, r[2 * i], (uint32_t)r[2 * i] + r[2 * i + 1], 0);
    return mk_bitset(w, popcnt_words(w));
}

/* container_or, containers.h:1008-1103 */
static oc_container_t c_or(const oc_container_t *x, const oc_container_t *y) {
    int t1 = x->type, t2 = y->type;
    if (t1 == OC_BITSET && t2 == OC_BITSET) { /* always a bitset, containers.h:1015-1020 */
        uint64_t *w = words_new();
        words_op(OC_OR, (const uint64_t *)x->data, (const uint64_t *)y->data, w);
        return mk_bitset(w, popcnt_words(w));
    }
    if (t1 == OC_ARRAY && t2 == OC_ARRAY) return aa_or(x, y);
    if (t1 == OC_RUN && t2 == OC_RUN) return run_to_efficient(rr_union_raw(x, y));
    if ((t1 == OC_BITSET && t2 == OC_ARRAY) || (t1 == OC_ARRAY && t2 == OC_BITSET)) {
        const oc_container_t *bs = t1 == OC_BITSET ? x : y, *ar = t1 == OC_BITSET ? y : x;
        uint64_t *w = words_new(); /* mixed_union.c:22-31 */
        memcpy(w, bs->data, 8192);
        const uint16_t *a = (const uint16_t *)ar->data;
        for (int i = 0; i < ar->card; i++) w_set(w, a[i]);
        return mk_bitset(w, popcnt_words(w));
    }
    if (t1 == OC_BITSET && t2 == OC_RUN) return rb_or(y, x);
    if (t1 == OC_RUN && t2 == OC_BITSET) return rb_or(x, y);
    if (t1 == OC_ARRAY && t2 == OC_RUN) return run_to_efficient(ar_union_raw(x, y));
    return run_to_efficient(ar_union_raw(y, x));
}

/* container_xor, containers.h:1449-1524 */
static oc_container_t c_xor(const oc_container_t *x, const oc_container_t *y) {
    int t1 = x->type, t2 = y->type;
    if (t1 == OC_ARRAY && t2 == OC_ARRAY) return aa_xor(x, y);
    if (t1 == OC_RUN && t2 == OC_RUN) return run_to_efficient(rr_xor_raw(x, y)); /* mixed_xor.c:179-186 */
    if (t1 == OC_ARRAY && t2 == OC_RUN) return ar_xor(x, y);
    if (t1 == OC_RUN && t2 == OC_ARRAY) return ar_xor(y, x);
    /* everything touching a bitset goes through the bitset and is re-typed by card:
     * mixed_xor.c:23-39 (A,B), :61-81 (R,B), :260-273 (B,B) */
    uint64_t *wa = words_from(x), *wb = words_from(y);
    words_op(OC_XOR, wa, wb, wa);
    free(wb);
    return bitset_or_array(wa);
}

/* container_andnot, containers.h:1783-1876 */
static oc_container_t c_andnot(const oc_container_t *x, const oc_container_t *y) {
    int t1 = x->type, t2 = y->type;
    if (t1 == OC_ARRAY && t2 == OC_ARRAY) return aa_andnot(x, y);
    if (t1 == OC_RUN && t2 == OC_RUN) {
        if (run_is_full(y)) return mk_empty();
        return run_to_efficient(rr_andnot_raw(x, y)); /* mixed_andnot.c:430-438 */
    }
    if (t1 == OC_ARRAY && t2 == OC_BITSET) return ab_filter(x, y, 0);
    if (t1 == OC_ARRAY && t2 == OC_RUN) return run_is_full(y) ? mk_empty() : ar_filter(x, y, 0);
    if (t1 == OC_RUN && t2 == OC_ARRAY) return ra_andnot(x, y);
    if (t1 == OC_RUN && t2 == OC_BITSET) return rb_andnot(x, y);
    if (t1 == OC_BITSET && t2 == OC_RUN && run_is_full(y)) return mk_empty();
    /* B\B (mixed_andnot.c:482-497), B\A (:54-72), B\R (:175-196) */
    uint64_t *wa = words_from(x), *wb = words_from(y);
    words_op(OC_ANDNOT, wa, wb, wa);
    free(wb);
    return bitset_or_array(wa);
}

static oc_container_t c_op(int op, const oc_container_t *x, const oc_container_t *y) {
    switch (op) {
        case OC_AND: return c_and(x, y);
        case OC_OR: return c_or(x, y);
        case OC_XOR: return c_xor(x, y);
        default: return c_andnot(x, y);
    }
}

/* container_and_cardinality, containers.h:811-859 */
static int c_and_card(const oc_container_t *x, const oc_container_t *y) {
    if (x->type == OC_ARRAY && y->type == OC_ARRAY) return aa_and_card(x, y);
    oc_container_t r = c_and(x, y);
    int card = r.card;
    c_free(&r);
    return card;
}

/* ------------------------------------------------------- bitmap level */
oc_bitmap_t *oc_create(void) {
    oc_bitmap_t *b = (oc_bitmap_t *)calloc(1, sizeof(*b));
    return b;
}
static void bm_push(oc_bitmap_t *b, uint16_t key, oc_container_t c) { /* ra_append, roaring_array.c:196-205 */
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 4;
        b->keys = (uint16_t *)realloc(b->keys, 2 * (size_t)b->cap);
        b->c = (oc_container_t *)realloc(b->c, sizeof(oc_container_t) * (size_t)b->cap);
    }
    b->keys[b->n] = key;
    b->c[b->n] = c;
    b->n++;
}
void oc_free(oc_bitmap_t *b) {
    if (!b) return;
    for (int i = 0; i < b->n; i++) c_free(&b->c[i]);
    free(b->keys);
    free(b->c);
    free(b);
}
oc_bitmap_t *oc_copy(const oc_bitmap_t *b) {
    oc_bitmap_t *r = oc_create();
    for (int i = 0; i < b->n; i++) bm_push(r, b->keys[i], c_clone(&b->c[i]));
    return r;
}

oc_bitmap_t *oc_from_sorted(const uint32_t *vals, size_t n) {
    oc_bitmap_t *r = oc_create();
    size_t i = 0;
    while (i < n) {
        uint32_t hi = vals[i] >> 16;
        size_t j = i;
        while (j < n && (vals[j] >> 16) == hi) j++;
        size_t card = j - i;
        if (card <= MAXARR) { /* arrays hold up to DEFAULT_MAX_SIZE values, array.h:38 */
            uint16_t *a = (uint16_t *)malloc(2 * card);
            for (size_t k = 0; k < card; k++) a[k] = (uint16_t)vals[i + k];
            bm_push(r, (uint16_t)hi, mk_array(a, (int)card));
        } else {
            uint64_t *w = words_new();
            for (size_t k = 0; k < card; k++) w_set(w, vals[i + k] & 0xFFFF);
            bm_push(r, (uint16_t)hi, mk_bitset(w, (int)card));
        }
        i = j;
    }
    return r;
}

static int words_nruns(const uint64_t *w) { /* bitset_container_number_of_runs, bitset.c:1046-1062 */
    int n = 0;
    uint64_t carry_next;
    for (int i = 0; i < WORDS; i++) {
        uint64_t x = w[i];
        carry_next = (i + 1 < WORDS) ? (w[i + 1] & 1) : 0;
        /* a run ends at bit k when bit k is set and bit k+1 is clear */
        uint64_t next = (x >> 1) | (carry_next << 63);
        n += __builtin_popcountll(x & ~next);
    }
    return n;
}

static oc_container_t run_from_words(const uint64_t *w, int nruns) {
    uint16_t *r = (uint16_t *)malloc(4 * (size_t)(nruns + 1));
    int k = 0, in = 0;
    uint32_t start = 0;
    for (uint32_t v = 0; v < 65536; v++) {
        int bit = w_get(w, v);
        if (bit && !in) { start = v; in = 1; }
        if (!bit && in) { r[2 * k] = (uint16_t)start; r[2 * k + 1] = (uint16_t)(v - 1 - start); k++; in = 0; }
    }
    if (in) { r[2 * k] = (uint16_t)start; r[2 * k + 1] = (uint16_t)(65535 - start); k++; }
    return mk_run(r, k);
}

/* convert_run_optimize, convert.c:217-321 */
int oc_run_optimize(oc_bitmap_t *b) {
    int any = 0;
    for (int i = 0; i < b->n; i++) {
        oc_container_t *c = &b->c[i];
        if (c->type == OC_RUN) {
            *c = run_to_efficient(*c);
        } else if (c->type == OC_ARRAY) {
            const uint16_t *a = (const uint16_t *)c->data;
            int nr = 0;
            for (int k = 0; k < c->card; k++)
                if (k == 0 || a[k] != a[k - 1] + 1) nr++;
            if (2 + 4 * nr >= 2 * c->card) continue;
            uint64_t *w = words_from(c);
            oc_container_t r = run_from_words(w, nr);
            free(w);
            c_free(c);
            *c = r;
        } else {
            int nr = words_nruns((const uint64_t *)c->data);
            if (8192 <= 2 + 4 * nr) continue;
            oc_container_t r = run_from_words((const uint64_t *)c->data, nr);
            c_free(c);
            *c = r;
        }
        if (b->c[i].type == OC_RUN) any = 1;
    }
    return any;
}

/* roaring_bitmap_remove_run_compression, roaring.c:1564-1592 -> convert_to_bitset_or_array_container,
 * convert.c:118-147: every run container becomes an array (card <= 4096) or a bitset. */
int oc_remove_run_compression(oc_bitmap_t *b) {
    int any = 0;
    for (int i = 0; i < b->n; i++) {
        oc_container_t *c = &b->c[i];
        if (c->type != OC_RUN) continue;
        any = 1;
        oc_container_t r = c->card <= MAXARR ? array_from_run(c) : mk_bitset(words_from(c), c->card);
        c_free(c);
        *c = r;
    }
    return any;
}

/* roaring_bitmap_intersect (roaring.c:2998-3027), _is_subset (:2151-2183), _is_strict_subset (:3172-3177):
 * restated through the cardinality identities; the reference's container walks decide the same sets. */
int oc_intersect(const oc_bitmap_t *a, const oc_bitmap_t *b) { return oc_op_cardinality(OC_AND, a, b) != 0; }
int oc_is_subset(const oc_bitmap_t *a, const oc_bitmap_t *b) { return oc_op_cardinality(OC_ANDNOT, a, b) == 0; }
int oc_is_strict_subset(const oc_bitmap_t *a, const oc_bitmap_t *b) {
    return oc_get_cardinality(b) > oc_get_cardinality(a) && oc_is_subset(a, b);
}

/* roaring_bitmap_flip (roaring.c:2289-2342) on [range_start, range_end): containers outside the key range are
 * copied; inside it every key gets container_not_range / container_not of the source container
 * (containers.h:2009-2073: bitset and array sources -> bitset or array by cardinality, mixed_negation.c:97-196;
 * run sources -> convert_run_to_efficient_container, mixed_negation.c:235-266) or, where the source has no
 * container, container_range_of_ones (containers.h:300-312: one value -> array, else one run); empty results are
 * dropped (roaring.c:2185-2211). */
static oc_container_t flip_container(const oc_container_t *src, uint32_t lo, uint32_t hi_excl) {
    if (!src) {
        /* container_range_of_ones is called with an EXCLUSIVE end but sizes the range as end - start + 1
         * (containers.h:304-305), so only a single value becomes an array; two or more become a run */
        uint32_t card = hi_excl - lo;
        if (card + 1 <= 2) {
            uint16_t *a = (uint16_t *)malloc(2 * card + 2);
            for (uint32_t k = 0; k < card; k++) a[k] = (uint16_t)(lo + k);
            return mk_array(a, (int)card);
        }
        uint16_t *r = (uint16_t *)malloc(4);
        r[0] = (uint16_t)lo;
        r[1] = (uint16_t)(card - 1);
        oc_container_t out = mk_run(r, 1);
        out.card = (int)card;
        return out;
    }
    uint64_t *w = words_from(src);
    for (uint32_t v = lo; v < hi_excl; v++) w[v >> 6] ^= (uint64_t)1 << (v & 63);
    if (src->type == OC_RUN) {
        int nr = words_nruns(w);
        if (nr == 0) {
            free(w);
            return mk_empty();
        }
        oc_container_t r = run_from_words(w, nr);
        free(w);
        return run_to_efficient(r);
    }
    return bitset_or_array(w);
}
oc_bitmap_t *oc_flip(const oc_bitmap_t *x, uint64_t range_start, uint64_t range_end) {
    if (range_start >= range_end || range_start > (uint64_t)0xFFFFFFFFu + 1) return oc_copy(x);
    /* roaring_bitmap_flip truncates both ends to 32 bits before roaring_bitmap_flip_closed (roaring.c:2295-2296) */
    uint32_t s = (uint32_t)range_start, e = (uint32_t)(range_end - 1); /* closed */
    if (s > e) return oc_copy(x); /* roaring_bitmap_flip_closed, roaring.c:2302-2304 */
    uint32_t ks = s >> 16, ke = e >> 16;
    oc_bitmap_t *r = oc_create();
    int i = 0;
    for (; i < x->n && x->keys[i] < ks; i++) bm_push(r, x->keys[i], c_clone(&x->c[i]));
    for (uint32_t k = ks; k <= ke; k++) {
        uint32_t lo = k == ks ? (s & 0xFFFF) : 0, hi = k == ke ? (e & 0xFFFF) + 1 : 65536;
        const oc_container_t *src = NULL;
        if (i < x->n && x->keys[i] == k) src = &x->c[i++];
        oc_container_t f = flip_container(src, lo, hi);
        if (f.card > 0) bm_push(r, (uint16_t)k, f);
        else c_free(&f);
    }
    for (; i < x->n; i++) bm_push(r, x->keys[i], c_clone(&x->c[i]));
    return r;
}

/* ------------------------------------------------------ portable format */
static int bm_has_run(const oc_bitmap_t *b) {
    for (int i = 0; i < b->n; i++)
        if (b->c[i].type == OC_RUN) return 1;
    return 0;
}
static size_t header_size(const oc_bitmap_t *b) { /* ra_portable_header_size, roaring_array.c:445-456 */
    size_t n = (size_t)b->n;
    if (bm_has_run(b)) return n < 4 ? 4 + (n + 7) / 8 + 4 * n : 4 + (n + 7) / 8 + 8 * n;
    return 8 + 8 * n;
}
size_t oc_size_in_bytes(const oc_bitmap_t *b) { /* roaring_array.c:458-466 */
    size_t s = header_size(b);
    for (int i = 0; i < b->n; i++) s += c_payload(&b->c[i]) + (b->c[i].type == OC_RUN ? 2 : 0);
    return s;
}
static void put16(char **p, uint16_t v) { memcpy(*p, &v, 2); *p += 2; }
static void put32(char **p, uint32_t v) { memcpy(*p, &v, 4); *p += 4; }

size_t oc_serialize(const oc_bitmap_t *b, char *buf) { /* ra_portable_serialize, roaring_array.c:469-531 */
    char *p = buf;
    int hasrun = bm_has_run(b);
    uint32_t n = (uint32_t)b->n;
    if (hasrun) {
        put32(&p, 12347u | ((n - 1) << 16));
        size_t s = (n + 7) / 8;
        memset(p, 0, s);
        for (uint32_t i = 0; i < n; i++)
            if (b->c[i].type == OC_RUN) p[i / 8] |= (char)(1 << (i % 8));
        p += s;
    } else {
        put32(&p, 12346u);
        put32(&p, n);
    }
    for (uint32_t i = 0; i < n; i++) {
        put16(&p, b->keys[i]);
        put16(&p, (uint16_t)(b->c[i].card - 1));
    }
    if (!hasrun || n >= 4) {
        uint32_t off = (uint32_t)header_size(b);
        for (uint32_t i = 0; i < n; i++) {
            put32(&p, off);
            off += (uint32_t)(c_payload(&b->c[i]) + (b->c[i].type == OC_RUN ? 2 : 0));
        }
    }
    for (uint32_t i = 0; i < n; i++) {
        const oc_container_t *c = &b->c[i];
        if (c->type == OC_RUN) put16(&p, (uint16_t)c->nruns);
        memcpy(p, c->data, c_payload(c));
        p += c_payload(c);
    }
    return (size_t)(p - buf);
}

oc_bitmap_t *oc_deserialize(const char *buf, size_t maxbytes) { /* ra_portable_deserialize, roaring_array.c:633-813 */
    const char *p = buf, *end = buf + maxbytes;
    uint32_t cookie, n;
    if (maxbytes < 4) return NULL;
    memcpy(&cookie, p, 4);
    p += 4;
    int hasrun = 0;
    const uint8_t *runflags = NULL;
    if ((cookie & 0xFFFF) == 12347u) {
        hasrun = 1;
        n = (cookie >> 16) + 1;
        runflags = (const uint8_t *)p;
        p += (n + 7) / 8;
    } else if (cookie == 12346u) {
        if (p + 4 > end) return NULL;
        memcpy(&n, p, 4);
        p += 4;
    } else {
        return NULL;
    }
    if (n > 65536 || p + 4 * (size_t)n > end) return NULL;
    const char *desc = p;
    p += 4 * (size_t)n;
    if (!hasrun || n >= 4) p += 4 * (size_t)n; /* offsets are redundant for a sequential reader */
    if (p > end) return NULL;
    oc_bitmap_t *b = oc_create();
    for (uint32_t i = 0; i < n; i++) {
        uint16_t key, cm1;
        memcpy(&key, desc + 4 * i, 2);
        memcpy(&cm1, desc + 4 * i + 2, 2);
        int card = (int)cm1 + 1;
        int isrun = hasrun && ((runflags[i / 8] >> (i % 8)) & 1);
        if (isrun) {
            uint16_t nr;
            if (p + 2 > end) { oc_free(b); return NULL; }
            memcpy(&nr, p, 2);
            p += 2;
            if (p + 4 * (size_t)nr > end) { oc_free(b); return NULL; }
            uint16_t *r = (uint16_t *)malloc(4 * (size_t)(nr + 1));
            memcpy(r, p, 4 * (size_t)nr);
            p += 4 * (size_t)nr;
            bm_push(b, key, mk_run(r, nr));
        } else if (card > MAXARR) { /* type inferred from cardinality, roaring_array.c:723-731 */
            if (p + 8192 > end) { oc_free(b); return NULL; }
            uint64_t *w = words_new();
            memcpy(w, p, 8192);
            p += 8192;
            bm_push(b, key, mk_bitset(w, card));
        } else {
            if (p + 2 * (size_t)card > end) { oc_free(b); return NULL; }
            uint16_t *a = (uint16_t *)malloc(2 * (size_t)card);
            memcpy(a, p, 2 * (size_t)card);
            p += 2 * (size_t)card;
            bm_push(b, key, mk_array(a, card));
        }
    }
    return b;
}

/* ----------------------------------------------------------- frozen format */
/* The "frozen" serialization (roaring.c:3176-3205 describes the layout):
 *   <bitset zone><run zone><array zone><keys u16 x n><counts u16 x n><typecodes u8 x n><header u32>
 * zones = the payloads of all containers of one type, in container order; counts[i] = cardinality - 1 for
 * bitset / array containers and n_runs for run containers; header = (n << 15) | FROZEN_COOKIE (13766,
 * roaring.h / roaring.c FROZEN_COOKIE).  The image is read from its END (the header is the last word). */
#define OC_FROZEN_COOKIE 13766u

size_t oc_frozen_size_in_bytes(const oc_bitmap_t *b) { /* roaring_bitmap_frozen_size_in_bytes, roaring.c:3207-3234 */
    size_t s = 0;
    for (int i = 0; i < b->n; i++) s += c_payload(&b->c[i]);
    return s + 5 * (size_t)b->n + 4;
}

size_t oc_frozen_serialize(const oc_bitmap_t *b, char *buf) { /* roaring_bitmap_frozen_serialize, roaring.c:3242-3328 */
    size_t zone[4] = {0, 0, 0, 0}; /* bytes per typecode */
    for (int i = 0; i < b->n; i++) zone[b->c[i].type] += c_payload(&b->c[i]);
    char *at[4];
    at[OC_BITSET] = buf;
    at[OC_RUN] = buf + zone[OC_BITSET];
    at[OC_ARRAY] = at[OC_RUN] + zone[OC_RUN];
    char *keys = at[OC_ARRAY] + zone[OC_ARRAY];
    char *counts = keys + 2 * (size_t)b->n;
    char *types = counts + 2 * (size_t)b->n;
    char *header = types + (size_t)b->n;
    for (int i = 0; i < b->n; i++) {
        const oc_container_t *c = &b->c[i];
        size_t nb = c_payload(c);
        memcpy(at[c->type], c->data, nb);
        at[c->type] += nb;
        uint16_t count;
        if (c->type == OC_RUN) count = (uint16_t)c->nruns;
        else if (c->type == OC_BITSET && c->card < 0) count = (uint16_t)(popcnt_words((const uint64_t *)c->data) - 1);
        else count = (uint16_t)(c->card - 1);
        memcpy(keys + 2 * (size_t)i, &b->keys[i], 2);
        memcpy(counts + 2 * (size_t)i, &count, 2);
        types[i] = (char)c->type;
    }
    uint32_t h = ((uint32_t)b->n << 15) | OC_FROZEN_COOKIE;
    memcpy(header, &h, 4);
    return (size_t)(header + 4 - buf);
}

/* roaring_bitmap_frozen_view, roaring.c:3330-3457, as a COPY: same acceptance -- cookie, typecodes in {1, 2, 3}, and
 * length EXACTLY the sum of the zones, the three per-container arrays and the header -- except the 32-byte alignment
 * of the buffer, which only the zero-copy view needs.  Like the view it does not look inside the payloads. */
oc_bitmap_t *oc_frozen_deserialize(const char *buf, size_t length) {
    if (length < 4) return NULL;
    uint32_t h;
    memcpy(&h, buf + length - 4, 4);
    if ((h & 0x7FFF) != OC_FROZEN_COOKIE) return NULL;
    size_t n = h >> 15;
    if (length < 4 + 5 * n) return NULL;
    const char *keys = buf + length - 4 - 5 * n, *counts = buf + length - 4 - 3 * n, *types = buf + length - 4 - n;
    size_t zone[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        uint16_t cnt;
        memcpy(&cnt, counts + 2 * i, 2);
        switch ((uint8_t)types[i]) {
            case OC_BITSET: zone[OC_BITSET] += 8192; break;
            case OC_RUN: zone[OC_RUN] += 4 * (size_t)cnt; break;
            case OC_ARRAY: zone[OC_ARRAY] += 2 * ((size_t)cnt + 1); break;
            default: return NULL;
        }
    }
    if (length != zone[OC_BITSET] + zone[OC_RUN] + zone[OC_ARRAY] + 5 * n + 4) return NULL;
    const char *at[4];
    at[OC_BITSET] = buf;
    at[OC_RUN] = buf + zone[OC_BITSET];
    at[OC_ARRAY] = at[OC_RUN] + zone[OC_RUN];
    oc_bitmap_t *b = oc_create();
    for (size_t i = 0; i < n; i++) {
        uint16_t cnt, key;
        memcpy(&cnt, counts + 2 * i, 2);
        memcpy(&key, keys + 2 * i, 2);
        uint8_t t = (uint8_t)types[i];
        if (t == OC_BITSET) {
            uint64_t *w = words_new();
            memcpy(w, at[t], 8192);
            at[t] += 8192;
            bm_push(b, key, mk_bitset(w, (int)cnt + 1));
        } else if (t == OC_RUN) {
            uint16_t *r = (uint16_t *)malloc(4 * ((size_t)cnt + 1));
            memcpy(r, at[t], 4 * (size_t)cnt);
            at[t] += 4 * (size_t)cnt;
            bm_push(b, key, mk_run(r, cnt));
        } else {
            uint16_t *a = (uint16_t *)malloc(2 * ((size_t)cnt + 1));
            memcpy(a, at[t], 2 * ((size_t)cnt + 1));
            at[t] += 2 * ((size_t)cnt + 1);
            bm_push(b, key, mk_array(a, (int)cnt + 1));
        }
    }
    return b;
}

/* ----------------------------------------------------------- pairwise */
/* roaring_bitmap_and (roaring.c:731-770), _or (:877-953), _xor (:1121-1196),
 * _andnot (:1275-1338): two-pointer merge over the key arrays; matched keys go
 * through the container dispatcher, empties are dropped (never for OR),
 * unmatched containers are copied with their type unchanged. */
oc_bitmap_t *oc_op(int op, const oc_bitmap_t *a, const oc_bitmap_t *b) {
    oc_bitmap_t *r = oc_create();
    int i = 0, j = 0;
    while (i < a->n && j < b->n) {
        uint16_t ka = a->keys[i], kb = b->keys[j];
        if (ka == kb) {
            oc_container_t c = c_op(op, &a->c[i], &b->c[j]);
            if (c.card > 0) bm_push(r, ka, c); /* container_nonzero_cardinality, roaring.c:756-760 */
            else c_free(&c);
            i++;
            j++;
        } else if (ka < kb) {
            if (op != OC_AND) bm_push(r, ka, c_clone(&a->c[i]));
            i++;
        } else {
            if (op == OC_OR || op == OC_XOR) bm_push(r, kb, c_clone(&b->c[j]));
            j++;
        }
    }
    if (op != OC_AND)
        for (; i < a->n; i++) bm_push(r, a->keys[i], c_clone(&a->c[i]));
    if (op == OC_OR || op == OC_XOR)
        for (; j < b->n; j++) bm_push(r, b->keys[j], c_clone(&b->c[j]));
    return r;
}
oc_bitmap_t *oc_and(const oc_bitmap_t *a, const oc_bitmap_t *b) { return oc_op(OC_AND, a, b); }
oc_bitmap_t *oc_or(const oc_bitmap_t *a, const oc_bitmap_t *b) { return oc_op(OC_OR, a, b); }
oc_bitmap_t *oc_xor(const oc_bitmap_t *a, const oc_bitmap_t *b) { return oc_op(OC_XOR, a, b); }
oc_bitmap_t *oc_andnot(const oc_bitmap_t *a, const oc_bitmap_t *b) { return oc_op(OC_ANDNOT, a, b); }

uint64_t oc_get_cardinality(const oc_bitmap_t *b) { /* roaring.c:1436-1443 */
    uint64_t s = 0;
    for (int i = 0; i < b->n; i++) s += (uint64_t)b->c[i].card;
    return s;
}
uint64_t oc_and_cardinality(const oc_bitmap_t *a, const oc_bitmap_t *b) { /* roaring.c:3048-3076 */
    uint64_t s = 0;
    int i = 0, j = 0;
    while (i < a->n && j < b->n) {
        if (a->keys[i] == b->keys[j]) { s += (uint64_t)c_and_card(&a->c[i], &b->c[j]); i++; j++; }
        else if (a->keys[i] < b->keys[j]) i++;
        else j++;
    }
    return s;
}
uint64_t oc_op_cardinality(int op, const oc_bitmap_t *a, const oc_bitmap_t *b) { /* roaring.c:3086-3107 */
    uint64_t c1 = oc_get_cardinality(a), c2 = oc_get_cardinality(b), in = oc_and_cardinality(a, b);
    switch (op) {
        case OC_AND: return in;
        case OC_OR: return c1 + c2 - in;
        case OC_XOR: return c1 + c2 - 2 * in;
        default: return c1 - in;
    }
}

/* -------------------------------------------------------------- *_many */
/* One accumulation step of roaring_bitmap_or_many on a matched key.
 * State = the accumulated container `acc` (may be a lazy bitset, card -1).
 * first_step selects roaring_bitmap_lazy_or's matched-key branch
 * (roaring.c:2529-2548) instead of lazy_or_inplace's (roaring.c:2619-2647). */
static void lazy_or_step(oc_container_t *acc, const oc_container_t *c2, int first_step) {
    if (!first_step) {
        /* container_is_full(c1): full run, or bitset with KNOWN card 65536 (containers.h:262-277) */
        if ((acc->type == OC_RUN && run_is_full(acc)) || (acc->type != OC_RUN && acc->card == 65536)) return;
    }
    if (first_step && (acc->type == OC_BITSET || c2->type == OC_BITSET)) {
        /* container_lazy_or, containers.h:1113-1215: B,B -> or_nocard; B,A -> lazy; B,R -> full run or lazy */
        const oc_container_t *run = acc->type == OC_RUN ? acc : (c2->type == OC_RUN ? c2 : NULL);
        if (run && run_is_full(run)) {
            oc_container_t r = c_clone(run);
            c_free(acc);
            *acc = r;
            return;
        }
        uint64_t *w = words_from(acc), *w2 = words_from(c2);
        for (int i = 0; i < WORDS; i++) w[i] |= w2[i];
        free(w2);
        c_free(acc);
        *acc = mk_bitset(w, -1);
        return;
    }
    /* acc is (converted to) a bitset, then container_lazy_ior(B, c2), containers.h:1333-1442 */
    if (acc->type != OC_BITSET) {
        uint64_t *w = words_from(acc);
        int card = acc->card; /* bitset_container_from_array/run keep the cardinality */
        c_free(acc);
        *acc = mk_bitset(w, card);
    }
    uint64_t *w = (uint64_t *)acc->data;
    if (c2->type == OC_BITSET) {
        const uint64_t *w2 = (const uint64_t *)c2->data;
        for (int i = 0; i < WORDS; i++) w[i] |= w2[i];
        acc->card = popcnt_words(w); /* LAZY_OR_BITSET_CONVERSION_TO_FULL: counted */
        if (acc->card == 65536) {    /* becomes a full run, containers.h:1343-1352 */
            uint16_t *r = (uint16_t *)malloc(4);
            r[0] = 0;
            r[1] = 0xFFFF;
            c_free(acc);
            *acc = mk_run(r, 1);
        }
    } else if (c2->type == OC_ARRAY) {
        const uint16_t *a = (const uint16_t *)c2->data;
        for (int i = 0; i < c2->card; i++) w_set(w, a[i]);
        acc->card = -1;
    } else {
        if (run_is_full(c2)) {
            oc_container_t r = c_clone(c2);
            c_free(acc);
            *acc = r;
            return;
        }
        const uint16_t *r = (const uint16_t *)c2->data;
        for (int i = 0; i < c2->nruns; i++) w_range(w, r[2 * i], (uint32_t)r[2 * i] + r[2 * i + 1], 0);
        acc->card = -1;
    }
}

/* container_repair_after_lazy, containers.h:344-371 */
static void repair(oc_container_t *c) {
    if (c->type == OC_BITSET) {
        c->card = popcnt_words((const uint64_t *)c->data);
        if (c->card <= MAXARR) {
            oc_container_t a = array_from_words((const uint64_t *)c->data, c->card);
            c_free(c);
            *c = a;
        }
    } else if (c->type == OC_RUN) {
        *c = run_to_efficient(*c);
    }
}

static int bm_find(const oc_bitmap_t *b, uint16_t key) {
    int lo = 0, hi = b->n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (b->keys[mid] < key) lo = mid + 1;
        else if (b->keys[mid] > key) hi = mid - 1;
        else return mid;
    }
    return -(lo + 1);
}
static void bm_insert(oc_bitmap_t *b, int pos, uint16_t key, oc_container_t c) { /* ra_insert_new_key_value_at */
    bm_push(b, key, c);
    for (int i = b->n - 1; i > pos; i--) {
        b->keys[i] = b->keys[i - 1];
        b->c[i] = b->c[i - 1];
    }
    b->keys[pos] = key;
    b->c[pos] = c;
}

/* roaring_bitmap_or_many, roaring.c:775-790 */
oc_bitmap_t *oc_or_many(size_t n, const oc_bitmap_t **x) {
    if (n == 0) return oc_create();
    if (n == 1) return oc_copy(x[0]);
    oc_bitmap_t *ans;
    /* lazy_or(x0, x1): empty operands short-circuit to a copy (roaring.c:2515-2520) */
    if (x[0]->n == 0) ans = oc_copy(x[1]);
    else if (x[1]->n == 0) ans = oc_copy(x[0]);
    else {
        ans = oc_copy(x[0]);
        for (int j = 0; j < x[1]->n; j++) {
            int pos = bm_find(ans, x[1]->keys[j]);
            if (pos >= 0) lazy_or_step(&ans->c[pos], &x[1]->c[j], 1);
            else bm_insert(ans, -pos - 1, x[1]->keys[j], c_clone(&x[1]->c[j]));
        }
    }
    for (size_t k = 2; k < n; k++) {
        /* lazy_or_inplace (roaring.c:2600-2682); an empty answer is overwritten by a copy */
        for (int j = 0; j < x[k]->n; j++) {
            int pos = bm_find(ans, x[k]->keys[j]);
            if (pos >= 0) lazy_or_step(&ans->c[pos], &x[k]->c[j], 0);
            else bm_insert(ans, -pos - 1, x[k]->keys[j], c_clone(&x[k]->c[j]));
        }
    }
    for (int i = 0; i < ans->n; i++) repair(&ans->c[i]); /* roaring.c:2845-2856 */
    return ans;
}

/* One matched-key step of roaring_bitmap_xor_many's fold.  first_step: container_lazy_xor
 * (containers.h:1570-1653, called by roaring_bitmap_lazy_xor, roaring.c:2684-2770, for a key both x[0] and x[1]
 * hold); otherwise container_lazy_ixor (containers.h:1747-1773, called by roaring_bitmap_lazy_xor_inplace,
 * roaring.c:2772-2843): only B,B stays lazy, every other pair is the EAGER container_ixor -- whose typing is
 * container_xor's (mixed_xor.c:221-380: the _ixor forms forward to the _xor forms). */
static void lazy_xor_step(oc_container_t *acc, const oc_container_t *c2, int first_step) {
    int t1 = acc->type, t2 = c2->type;
    oc_container_t out;
    if (t1 == OC_BITSET && t2 == OC_BITSET) { /* bitset_container_xor_nocard: cardinality unknown */
        uint64_t *w = words_from(acc);
        const uint64_t *w2 = (const uint64_t *)c2->data;
        for (int i = 0; i < WORDS; i++) w[i] ^= w2[i];
        out = mk_bitset(w, -1);
    } else if (first_step && t1 == OC_ARRAY && t2 == OC_ARRAY) {
        /* array_array_container_lazy_xor, mixed_xor.c:221-253: an array up to ARRAY_LAZY_LOWERBOUND = 1024
         * values in total, a lazy bitset (never converted back here) beyond */
        if (acc->card + c2->card <= 1024) {
            uint16_t *o = (uint16_t *)malloc(2 * (size_t)(acc->card + c2->card + 1));
            int k = merge_xor((const uint16_t *)acc->data, acc->card, (const uint16_t *)c2->data, c2->card, o);
            out = mk_array(o, k);
        } else {
            uint64_t *w = words_from(acc), *w2 = words_from(c2);
            for (int i = 0; i < WORDS; i++) w[i] ^= w2[i];
            free(w2);
            out = mk_bitset(w, -1);
        }
    } else if (first_step && (t1 == OC_BITSET || t2 == OC_BITSET)) {
        /* array_bitset_container_lazy_xor / run_bitset_container_lazy_xor, mixed_xor.c:41-59, 83-98 */
        uint64_t *w = words_from(acc), *w2 = words_from(c2);
        for (int i = 0; i < WORDS; i++) w[i] ^= w2[i];
        free(w2);
        out = mk_bitset(w, -1);
    } else if (first_step && t1 != t2) { /* A,R / R,A: array_run_container_lazy_xor keeps the raw run list */
        out = t1 == OC_ARRAY ? ar_xor_raw(acc, c2) : ar_xor_raw(c2, acc);
    } else {
        /* first step R,R (run_run_container_xor: converted at once), and every eager step.  A lazy bitset gets its
         * cardinality first (containers.h:1763-1769); c_xor never reads a bitset's card field, so nothing to do */
        out = c_xor(acc, c2);
    }
    c_free(acc);
    *acc = out;
}
/* container_nonzero_cardinality, containers.h:436-451 (a lazy bitset is scanned: bitset.h bitset_container_empty) */
static int c_nonzero(const oc_container_t *c) {
    if (c->type == OC_ARRAY) return c->card > 0;
    if (c->type == OC_RUN) return c->nruns > 0;
    if (c->card >= 0) return c->card > 0;
    const uint64_t *w = (const uint64_t *)c->data;
    for (int i = 0; i < WORDS; i++)
        if (w[i]) return 1;
    return 0;
}
static void bm_remove(oc_bitmap_t *b, int pos) { /* ra_remove_at_index, roaring_array.c:381-389 */
    c_free(&b->c[pos]);
    for (int i = pos; i + 1 < b->n; i++) {
        b->keys[i] = b->keys[i + 1];
        b->c[i] = b->c[i + 1];
    }
    b->n--;
}

/* roaring_bitmap_xor_many, roaring.c:795-809: lazy_xor(x0, x1), then lazy_xor_inplace with x2 ..., then
 * repair_after_lazy.  A FIXED left fold, so the container types of the result are reproducible (unlike the heap's):
 * the answer's container under a key is a clone of the first member, an accumulator that becomes empty is REMOVED
 * (roaring.c:2713-2717, 2812-2820) -- the next member under that key is cloned afresh. */
oc_bitmap_t *oc_xor_many(size_t n, const oc_bitmap_t **x) {
    if (n == 0) return oc_create();
    if (n == 1) return oc_copy(x[0]);
    oc_bitmap_t *ans = oc_copy(x[0]);
    for (size_t k = 1; k < n; k++) {
        for (int j = 0; j < x[k]->n; j++) {
            int pos = bm_find(ans, x[k]->keys[j]);
            if (pos < 0) {
                bm_insert(ans, -pos - 1, x[k]->keys[j], c_clone(&x[k]->c[j]));
                continue;
            }
            /* the first step's matched keys are exactly the keys x[0] and x[1] share: ans == copy of x[0] at k == 1 */
            lazy_xor_step(&ans->c[pos], &x[k]->c[j], k == 1);
            if (!c_nonzero(&ans->c[pos])) bm_remove(ans, pos);
        }
    }
    for (int i = 0; i < ans->n; i++) repair(&ans->c[i]);
    return ans;
}

/* ------------------------------------------------------- or_many_heap */
/* container_lazy_or (containers.h:1113-1215) and container_lazy_ior (:1333-1442) WITHOUT the early bitset conversion
 * (bitsetconversion = false, as roaring_priority_queue.c calls them).  Functional restatement: a new container is
 * returned whatever the reference does in place; `inplace` selects the one typing difference between the two --
 * B,B: lazy_or leaves the cardinality unknown, lazy_ior counts (LAZY_OR_BITSET_CONVERSION_TO_FULL, perfparameters.h:45)
 * and turns a full result into a full run. */
static oc_container_t lazy_or_nc(const oc_container_t *c1, const oc_container_t *c2, int inplace) {
    int t1 = c1->type, t2 = c2->type;
    if (t1 == OC_BITSET && t2 == OC_BITSET) {
        uint64_t *w = words_from(c1);
        const uint64_t *w2 = (const uint64_t *)c2->data;
        for (int i = 0; i < WORDS; i++) w[i] |= w2[i];
        if (!inplace) return mk_bitset(w, -1);
        int card = popcnt_words(w);
        if (card == 65536) {
            free(w);
            uint16_t *r = (uint16_t *)malloc(4);
            r[0] = 0; r[1] = 0xFFFF;
            return mk_run(r, 1);
        }
        return mk_bitset(w, card);
    }
    if (t1 == OC_ARRAY && t2 == OC_ARRAY) { /* array_array_container_lazy(_inplace)_union, mixed_union.c:247-365 */
        if (c1->card + c2->card <= 1024) {
            uint16_t *o = (uint16_t *)malloc(2 * (size_t)(c1->card + c2->card + 1));
            int k = merge_or((const uint16_t *)c1->data, c1->card, (const uint16_t *)c2->data, c2->card, o);
            return mk_array(o, k);
        }
        uint64_t *w = words_from(c1), *w2 = words_from(c2);
        for (int i = 0; i < WORDS; i++) w[i] |= w2[i];
        free(w2);
        return mk_bitset(w, -1);
    }
    if (t1 == OC_RUN && t2 == OC_RUN) return run_to_efficient(rr_union_raw(c1, c2)); /* converted at once in both forms */
    if (t1 == OC_BITSET || t2 == OC_BITSET) {
        const oc_container_t *other = t1 == OC_BITSET ? c2 : c1;
        if (other->type == OC_RUN && run_is_full(other)) return c_clone(other); /* containers.h:1169-1196, 1395-1416 */
        uint64_t *w = words_from(c1), *w2 = words_from(c2);
        for (int i = 0; i < WORDS; i++) w[i] |= w2[i];
        free(w2);
        return mk_bitset(w, -1);
    }
    /* A,R / R,A: array_run_container_union -- the raw run list, no conversion ("we are lazy") */
    return t1 == OC_ARRAY ? ar_union_raw(c1, c2) : ar_union_raw(c2, c1);
}
static int c_is_full(const oc_container_t *c) { /* container_is_full, containers.h:262-277 */
    return c->type == OC_RUN ? run_is_full(c) : c->card == 65536;
}
/* what one step of the tournament builds out of two elements.  mode 0: roaring_bitmap_lazy_or(x1, x2, false)
 * (roaring.c:2509-2598); 1: roaring_bitmap_lazy_or_inplace(acc = x1, x2, false) (roaring.c:2600-2682: a full accumulator
 * container is skipped); 3: lazy_or_from_lazy_inputs (roaring_priority_queue.c:98-192: a bitset second operand goes first). */
static oc_bitmap_t *heap_merge(const oc_bitmap_t *x1, const oc_bitmap_t *x2, int mode) {
    oc_bitmap_t *ans = oc_create();
    int i = 0, j = 0;
    while (i < x1->n || j < x2->n) {
        if (j >= x2->n || (i < x1->n && x1->keys[i] < x2->keys[j])) { bm_push(ans, x1->keys[i], c_clone(&x1->c[i])); i++; }
        else if (i >= x1->n || x2->keys[j] < x1->keys[i]) { bm_push(ans, x2->keys[j], c_clone(&x2->c[j])); j++; }
        else {
            const oc_container_t *a = &x1->c[i], *b = &x2->c[j];
            oc_container_t r;
            if (mode == 0) r = lazy_or_nc(a, b, 0);
            else if (mode == 1) r = c_is_full(a) ? c_clone(a) : lazy_or_nc(a, b, 1);
            else r = (b->type == OC_BITSET && a->type != OC_BITSET) ? lazy_or_nc(b, a, 1) : lazy_or_nc(a, b, 1);
            bm_push(ans, x1->keys[i], r);
            i++; j++;
        }
    }
    return ans;
}
/* roaring_bitmap_portable_size_in_bytes of a (possibly lazy) bitmap: a lazy bitset is 8192 bytes like any other */
static uint64_t heap_size(const oc_bitmap_t *b) { return (uint64_t)oc_size_in_bytes(b); }

typedef struct { uint64_t size; int temp; oc_bitmap_t *bm; } pq_el;
typedef struct { pq_el *e; uint64_t n; } pq_t;
/* the reference's binary heap, move for move (roaring_priority_queue.c:27-96): ties are broken by position */
static void pq_down(pq_t *pq, uint32_t i) {
    uint32_t size = (uint32_t)pq->n, hsize = size >> 1;
    pq_el ai = pq->e[i];
    while (i < hsize) {
        uint32_t l = (i << 1) + 1, r = l + 1;
        pq_el best = pq->e[l];
        if (r < size && pq->e[r].size < best.size) { l = r; best = pq->e[r]; }
        if (!(best.size < ai.size)) break;
        pq->e[i] = best;
        i = l;
    }
    pq->e[i] = ai;
}
static void pq_push(pq_t *pq, pq_el t) {
    uint64_t i = pq->n;
    pq->e[pq->n++] = t;
    while (i > 0) {
        uint64_t p = (i - 1) >> 1;
        pq_el ap = pq->e[p];
        if (!(t.size < ap.size)) break;
        pq->e[i] = ap;
        i = p;
    }
    pq->e[i] = t;
}
static pq_el pq_pop(pq_t *pq) {
    pq_el ans = pq->e[0];
    if (pq->n > 1) {
        pq->e[0] = pq->e[--pq->n];
        pq_down(pq, 0);
    } else
        --pq->n;
    return ans;
}
/* roaring_bitmap_or_many_heap, roaring_priority_queue.c:200-247: a tournament ordered by serialized size, lazy unions
 * without early bitset conversion, one repair pass at the end. */
oc_bitmap_t *oc_or_many_heap(size_t n, const oc_bitmap_t **x) {
    if (n == 0) return oc_create();
    if (n == 1) return oc_copy(x[0]);
    pq_t pq;
    pq.e = (pq_el *)malloc(sizeof(pq_el) * n);
    pq.n = n;
    for (size_t i = 0; i < n; i++) { pq.e[i].bm = (oc_bitmap_t *)x[i]; pq.e[i].temp = 0; pq.e[i].size = heap_size(x[i]); }
    for (int32_t i = (int32_t)(n >> 1); i >= 0; i--) pq_down(&pq, (uint32_t)i);
    while (pq.n > 1) {
        pq_el x1 = pq_pop(&pq), x2 = pq_pop(&pq);
        pq_el ne;
        ne.temp = 1;
        if (x1.temp && x2.temp) {
            /* (an empty operand: the other one is handed back as it is -- the same containers) */
            ne.bm = heap_merge(x1.bm, x2.bm, 3);
        } else if (x2.temp) {
            ne.bm = heap_merge(x2.bm, x1.bm, 1);
        } else if (x1.temp) {
            ne.bm = heap_merge(x1.bm, x2.bm, 1);
        } else {
            ne.bm = heap_merge(x1.bm, x2.bm, 0);
        }
        if (x1.temp) oc_free(x1.bm);
        if (x2.temp) oc_free(x2.bm);
        ne.size = heap_size(ne.bm);
        pq_push(&pq, ne);
    }
    pq_el X = pq_pop(&pq);
    free(pq.e);
    oc_bitmap_t *ans = X.bm;
    for (int i = 0; i < ans->n; i++) repair(&ans->c[i]);
    return ans;
}

/* -------------------------------------------------------------- checks */
int oc_validate(const oc_bitmap_t *b) { /* roaring.c:454-523 + bitset.c:1023-1044, array.c:456-493, run.c:669-716 */
    for (int i = 0; i < b->n; i++) {
        if (i && b->keys[i] <= b->keys[i - 1]) return 0;
        const oc_container_t *c = &b->c[i];
        if (c->type == OC_BITSET) {
            if (c->card <= MAXARR || c->card != popcnt_words((const uint64_t *)c->data)) return 0;
        } else if (c->type == OC_ARRAY) {
            if (c->card < 1 || c->card > MAXARR) return 0;
            const uint16_t *a = (const uint16_t *)c->data;
            for (int k = 1; k < c->card; k++)
                if (a[k] <= a[k - 1]) return 0;
        } else if (c->type == OC_RUN) {
            if (c->nruns < 1) return 0;
            const uint16_t *r = (const uint16_t *)c->data;
            int32_t last_end = -2;
            for (int k = 0; k < c->nruns; k++) {
                int32_t s = r[2 * k], e = s + r[2 * k + 1];
                if (e > 65535 || s <= last_end + 1) return 0; /* non-overlapping AND non-adjacent */
                last_end = e;
            }
        } else {
            return 0;
        }
    }
    return 1;
}

void oc_to_uint32(const oc_bitmap_t *b, uint32_t *out) {
    size_t k = 0;
    for (int i = 0; i < b->n; i++) {
        const oc_container_t *c = &b->c[i];
        uint32_t hi = (uint32_t)b->keys[i] << 16;
        if (c->type == OC_ARRAY) {
            const uint16_t *a = (const uint16_t *)c->data;
            for (int j = 0; j < c->card; j++) out[k++] = hi | a[j];
        } else if (c->type == OC_RUN) {
            const uint16_t *r = (const uint16_t *)c->data;
            for (int j = 0; j < c->nruns; j++)
                for (uint32_t v = r[2 * j]; v <= (uint32_t)r[2 * j] + r[2 * j + 1]; v++) out[k++] = hi | v;
        } else {
            const uint64_t *w = (const uint64_t *)c->data;
            for (int wi = 0; wi < WORDS; wi++) {
                uint64_t x = w[wi];
                while (x) {
                    out[k++] = hi | (uint32_t)(wi * 64 + __builtin_ctzll(x));
                    x &= x - 1;
                }
            }
        }
    }
}

int oc_equals(const oc_bitmap_t *a, const oc_bitmap_t *b) {
    uint64_t ca = oc_get_cardinality(a), cb = oc_get_cardinality(b);
    if (ca != cb) return 0;
    uint32_t *va = (uint32_t *)malloc(4 * (size_t)(ca + 1)), *vb = (uint32_t *)malloc(4 * (size_t)(cb + 1));
    oc_to_uint32(a, va);
    oc_to_uint32(b, vb);
    int eq = memcmp(va, vb, 4 * (size_t)ca) == 0;
    free(va);
    free(vb);
    return eq;
}

void oc_type_counts(const oc_bitmap_t *b, int64_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    for (int i = 0; i < b->n; i++) out[b->c[i].type - 1]++;
}

/* -------------------------------------------------------------- 64-bit */
static oc_bitmap64_t *oc64_create(void) { return (oc_bitmap64_t *)calloc(1, sizeof(oc_bitmap64_t)); }
static void b64_push(oc_bitmap64_t *b, uint32_t high, oc_bitmap_t *bm) {
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 4;
        b->high = (uint32_t *)realloc(b->high, 4 * (size_t)b->cap);
        b->bm = (oc_bitmap_t **)realloc(b->bm, sizeof(void *) * (size_t)b->cap);
    }
    b->high[b->n] = high;
    b->bm[b->n] = bm;
    b->n++;
}
void oc64_free(oc_bitmap64_t *b) {
    if (!b) return;
    for (int64_t i = 0; i < b->n; i++) oc_free(b->bm[i]);
    free(b->high);
    free(b->bm);
    free(b);
}
/* roaring64_bitmap_portable_deserialize_safe, roaring64.c:2442-2535 */
oc_bitmap64_t *oc64_deserialize(const char *buf, size_t maxbytes) {
    if (maxbytes < 8) return NULL;
    uint64_t nb;
    memcpy(&nb, buf, 8);
    const char *p = buf + 8, *end = buf + maxbytes;
    oc_bitmap64_t *r = oc64_create();
    for (uint64_t i = 0; i < nb; i++) {
        if (p + 4 > end) { oc64_free(r); return NULL; }
        uint32_t high;
        memcpy(&high, p, 4);
        p += 4;
        oc_bitmap_t *bm = oc_deserialize(p, (size_t)(end - p));
        if (!bm) { oc64_free(r); return NULL; }
        p += oc_size_in_bytes(bm);
        b64_push(r, high, bm);
    }
    return r;
}
size_t oc64_size_in_bytes(const oc_bitmap64_t *b) { /* roaring64.c:2262-2321 */
    size_t s = 8;
    for (int64_t i = 0; i < b->n; i++) s += 4 + oc_size_in_bytes(b->bm[i]);
    return s;
}
size_t oc64_serialize(const oc_bitmap64_t *b, char *buf) { /* roaring64.c:2323-2393 */
    char *p = buf;
    uint64_t nb = (uint64_t)b->n;
    memcpy(p, &nb, 8);
    p += 8;
    for (int64_t i = 0; i < b->n; i++) {
        memcpy(p, &b->high[i], 4);
        p += 4;
        p += oc_serialize(b->bm[i], p);
    }
    return (size_t)(p - buf);
}
oc_bitmap64_t *oc64_from_sorted(const uint64_t *vals, size_t n) {
    oc_bitmap64_t *r = oc64_create();
    size_t i = 0;
    while (i < n) {
        uint32_t high = (uint32_t)(vals[i] >> 32);
        size_t j = i;
        while (j < n && (uint32_t)(vals[j] >> 32) == high) j++;
        uint32_t *lo = (uint32_t *)malloc(4 * (j - i));
        for (size_t k = i; k < j; k++) lo[k - i] = (uint32_t)vals[k];
        b64_push(r, high, oc_from_sorted(lo, j - i));
        free(lo);
        i = j;
    }
    return r;
}
int oc64_run_optimize(oc_bitmap64_t *b) {
    int any = 0;
    for (int64_t i = 0; i < b->n; i++) any |= oc_run_optimize(b->bm[i]);
    return any;
}
/* roaring64_bitmap_and/or/xor/andnot: same per-key container ops under 48-bit keys;
 * grouping the keys by their high 32 bits gives the same result set and types. */
oc_bitmap64_t *oc64_op(int op, const oc_bitmap64_t *a, const oc_bitmap64_t *b) {
    oc_bitmap64_t *r = oc64_create();
    int64_t i = 0, j = 0;
    while (i < a->n && j < b->n) {
        if (a->high[i] == b->high[j]) {
            oc_bitmap_t *t = oc_op(op, a->bm[i], b->bm[j]);
            if (t->n) b64_push(r, a->high[i], t);
            else oc_free(t);
            i++;
            j++;
        } else if (a->high[i] < b->high[j]) {
            if (op != OC_AND) b64_push(r, a->high[i], oc_copy(a->bm[i]));
            i++;
        } else {
            if (op == OC_OR || op == OC_XOR) b64_push(r, b->high[j], oc_copy(b->bm[j]));
            j++;
        }
    }
    if (op != OC_AND)
        for (; i < a->n; i++) b64_push(r, a->high[i], oc_copy(a->bm[i]));
    if (op == OC_OR || op == OC_XOR)
        for (; j < b->n; j++) b64_push(r, b->high[j], oc_copy(b->bm[j]));
    return r;
}
uint64_t oc64_get_cardinality(const oc_bitmap64_t *b) {
    uint64_t s = 0;
    for (int64_t i = 0; i < b->n; i++) s += oc_get_cardinality(b->bm[i]);
    return s;
}
/* roaring64_bitmap_flip (roaring64.c:2007-2074): [min, max) over the 64-bit universe.  Per 48-bit key of the range the
 * reference applies container_not / container_not_range (or container_range_of_ones where the key is absent) -- the
 * same container-level functions roaring_bitmap_flip_closed applies per 16-bit key -- so, bucket by bucket (high 32
 * bits), the result is what oc_flip gives on the bucket's own 32-bit sub-range; buckets the bitmap lacks start empty.
 * (oc_flip's 32-bit truncation quirk does not come into play: every sub-range lies inside [0, 2^32].) */
oc_bitmap64_t *oc64_flip(const oc_bitmap64_t *x, uint64_t min, uint64_t max) {
    oc_bitmap64_t *r = oc64_create();
    if (min >= max) {
        for (int64_t i = 0; i < x->n; i++) b64_push(r, x->high[i], oc_copy(x->bm[i]));
        return r;
    }
    const uint64_t last = max - 1; /* closed */
    const uint64_t hb0 = min >> 32, hb1 = last >> 32;
    int64_t i = 0;
    for (; i < x->n && x->high[i] < hb0; i++) b64_push(r, x->high[i], oc_copy(x->bm[i]));
    for (uint64_t hb = hb0; hb <= hb1; hb++) {
        const uint64_t lo = hb == hb0 ? (min & 0xFFFFFFFFull) : 0;
        const uint64_t hi = hb == hb1 ? (last & 0xFFFFFFFFull) + 1 : 0x100000000ull; /* exclusive */
        oc_bitmap_t *src = NULL, *empty = NULL;
        if (i < x->n && x->high[i] == hb) src = x->bm[i++];
        else src = empty = oc_create();
        oc_bitmap_t *f = oc_flip(src, lo, hi);
        if (empty) oc_free(empty);
        if (f->n) b64_push(r, (uint32_t)hb, f);
        else oc_free(f);
    }
    for (; i < x->n; i++) b64_push(r, x->high[i], oc_copy(x->bm[i]));
    return r;
}
/* No C many-way API exists for 64-bit; reference = left fold of or. */
oc_bitmap64_t *oc64_or_many(size_t n, const oc_bitmap64_t **x) {
    oc_bitmap64_t *ans = oc64_create();
    for (size_t k = 0; k < n; k++) {
        oc_bitmap64_t *t = oc64_op(OC_OR, ans, x[k]);
        oc64_free(ans);
        ans = t;
    }
    return ans;
}
