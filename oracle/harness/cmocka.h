/* TEST INFRASTRUCTURE ONLY -- a minimal stand-in for the parts of cmocka 2.x that the reference's
 * tests/toplevel_unit.c uses (cmocka itself is fetched from the network by the reference's build and
 * is not available here).  Written from the documented cmocka API; failures abort the
 * current test via longjmp and are counted. */
#ifndef DROPIN_CMOCKA_SHIM_H
#define DROPIN_CMOCKA_SHIM_H
#include <setjmp.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void (*CMUnitTestFunction)(void **state);
struct CMUnitTest {
    const char *name;
    CMUnitTestFunction test_func;
};
#define cmocka_unit_test(f) { #f, f }

static jmp_buf shim_jmp;
static int shim_failed_here;
static void shim_fail(const char *file, int line, const char *what) {
    fprintf(stderr, "[  ERROR   ] %s:%d: %s\n", file, line, what);
    shim_failed_here = 1;
    longjmp(shim_jmp, 1);
}
#define fail_msg(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); shim_fail(__FILE__, __LINE__, "fail_msg"); } while (0)
#define fail() shim_fail(__FILE__, __LINE__, "fail()")
#define assert_true(c) do { if (!(c)) shim_fail(__FILE__, __LINE__, "assert_true(" #c ")"); } while (0)
#define assert_false(c) do { if (c) shim_fail(__FILE__, __LINE__, "assert_false(" #c ")"); } while (0)
#define assert_non_null(p) do { if ((p) == NULL) shim_fail(__FILE__, __LINE__, "assert_non_null(" #p ")"); } while (0)
#define assert_null(p) do { if ((p) != NULL) shim_fail(__FILE__, __LINE__, "assert_null(" #p ")"); } while (0)
#define assert_ptr_equal(a, b) do { if ((const void *)(a) != (const void *)(b)) shim_fail(__FILE__, __LINE__, "assert_ptr_equal(" #a ", " #b ")"); } while (0)
#define assert_ptr_not_equal(a, b) do { if ((const void *)(a) == (const void *)(b)) shim_fail(__FILE__, __LINE__, "assert_ptr_not_equal(" #a ", " #b ")"); } while (0)
#define assert_int_equal(a, b) do { long long a_ = (long long)(a), b_ = (long long)(b); if (a_ != b_) { \
    fprintf(stderr, "  %lld != %lld\n", a_, b_); shim_fail(__FILE__, __LINE__, "assert_int_equal(" #a ", " #b ")"); } } while (0)
#define assert_int_not_equal(a, b) do { if ((long long)(a) == (long long)(b)) shim_fail(__FILE__, __LINE__, "assert_int_not_equal"); } while (0)
#define assert_uint_in_range(v, lo, hi) do { unsigned long long v_ = (unsigned long long)(v); \
    if (v_ < (unsigned long long)(lo) || v_ > (unsigned long long)(hi)) shim_fail(__FILE__, __LINE__, "assert_uint_in_range(" #v ")"); } while (0)
#define assert_in_range(v, lo, hi) assert_uint_in_range(v, lo, hi)
#define assert_memory_equal(a, b, n) do { if (memcmp((a), (b), (n)) != 0) shim_fail(__FILE__, __LINE__, "assert_memory_equal"); } while (0)
#define assert_string_equal(a, b) do { if (strcmp((a), (b)) != 0) shim_fail(__FILE__, __LINE__, "assert_string_equal"); } while (0)

static int shim_run(const char *group, const struct CMUnitTest *tests, size_t n) {
    int failed = 0;
    const char *only = getenv("SHIM_ONLY");
    fprintf(stderr, "[==========] %s: running %zu test(s)\n", group, n);
    for (size_t i = 0; i < n; ++i) {
        if (only && !strstr(tests[i].name, only)) continue;
        void *state = NULL;
        shim_failed_here = 0;
        fprintf(stderr, "[ RUN      ] %s\n", tests[i].name);
        if (setjmp(shim_jmp) == 0) tests[i].test_func(&state);
        if (shim_failed_here) { ++failed; fprintf(stderr, "[  FAILED  ] %s\n", tests[i].name); }
        else fprintf(stderr, "[       OK ] %s\n", tests[i].name);
    }
    fprintf(stderr, "[==========] %s: %zu test(s) run, %d failed\n", group, n, failed);
    printf("%s: %zu tests, %d failed\n", group, n, failed);
    return failed;
}
#define cmocka_run_group_tests(tests, setup, teardown) shim_run(#tests, tests, sizeof(tests) / sizeof((tests)[0]))
#endif
