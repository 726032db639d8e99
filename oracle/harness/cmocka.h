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

#include <time.h>
static double shim_now(void) {
    struct timespec ts;
    timespec_get(&ts, TIME_UTC); /* (C11 / C++17: the harness is compiled with -std=c11, which hides clock_gettime) */
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* Selection (environment): SHIM_ONLY=substring -- tests whose name contains it; SHIM_EVERY=N [SHIM_PHASE=k] -- every N-th
 * test, starting at k; SHIM_MAX_SECONDS=T -- no test is STARTED after T seconds (a slice of a long harness that still
 * ends with a verdict: what ran must pass, what did not is listed as such).  Every test's wall time is printed. */
static int shim_run(const char *group, const struct CMUnitTest *tests, size_t n) {
    int failed = 0;
    size_t ran = 0, not_run = 0;
    const char *only = getenv("SHIM_ONLY");
    const long every = getenv("SHIM_EVERY") ? atol(getenv("SHIM_EVERY")) : 1, phase = getenv("SHIM_PHASE") ? atol(getenv("SHIM_PHASE")) : 0;
    const double budget = getenv("SHIM_MAX_SECONDS") ? atof(getenv("SHIM_MAX_SECONDS")) : 0.0, t_start = shim_now();
    fprintf(stderr, "[==========] %s: running %zu test(s)\n", group, n);
    for (size_t i = 0; i < n; ++i) {
        if (only && !strstr(tests[i].name, only)) continue;
        if (every > 1 && (long)(i % (size_t)every) != phase % every) continue;
        if (budget > 0.0 && shim_now() - t_start > budget) {
            ++not_run;
            fprintf(stderr, "[ NOT RUN  ] %s (time budget of %.0f s spent)\n", tests[i].name, budget);
            continue;
        }
        void *state = NULL;
        shim_failed_here = 0;
        fprintf(stderr, "[ RUN      ] %s\n", tests[i].name);
        const double t0 = shim_now();
        if (setjmp(shim_jmp) == 0) tests[i].test_func(&state);
        ++ran;
        if (shim_failed_here) { ++failed; fprintf(stderr, "[  FAILED  ] %s\n", tests[i].name); }
        else fprintf(stderr, "[       OK ] %s (%.0f ms)\n", tests[i].name, 1e3 * (shim_now() - t0));
    }
    fprintf(stderr, "[==========] %s: %zu test(s) run, %d failed\n", group, ran, failed);
    printf("%s: %zu tests, %d failed (%zu run, %zu not started within the time budget)\n", group, n, failed, ran, not_run);
    return failed;
}
#define cmocka_run_group_tests(tests, setup, teardown) shim_run(#tests, tests, sizeof(tests) / sizeof((tests)[0]))
#endif
