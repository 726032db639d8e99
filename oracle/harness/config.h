/* TEST INFRASTRUCTURE ONLY -- stands in for the reference's generated tests/config.h (tests/config.h.in).
 * The data files are the verbatim fixtures committed under tests/golden/. */
#define BENCHMARK_DATA_DIR "/root/repo/tests/golden/"
#define TEST_DATA_DIR "/root/repo/tests/golden/"
