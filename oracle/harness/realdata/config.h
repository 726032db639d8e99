/* TEST INFRASTRUCTURE ONLY -- stands in for the reference's generated tests/config.h when the reference's
 * tests/realdata_unit.c is built for the emulator harness: that test walks the text datasets where they lie in the
 * reference tree (build container only; the binary is skipped wherever /root/reference is absent). */
#define BENCHMARK_DATA_DIR "/root/reference/benchmarks/realdata/"
#define TEST_DATA_DIR "/root/reference/tests/testdata/"
