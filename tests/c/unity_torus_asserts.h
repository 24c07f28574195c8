/*
 * Torus-distance versions of the two assertion macros test/tests.c of the reference defines on top of Unity (test/tests.c:30-31):
 *     #define TEST_ASSERT_TORUS_WITHIN_MESSAGE        TEST_ASSERT_HEX64_WITHIN_MESSAGE
 *     #define TEST_ASSERT_TORUS_ARRAY_WITHIN_MESSAGE  TEST_ASSERT_INT64_ARRAY_WITHIN_MESSAGE
 * Unity compares the two values as (un)signed 64-bit INTEGERS, so a phase just below 0 against an expected 0, or two values either side of 2^63, are
 * reported as "not within delta" although their distance on the torus is tiny; which of the reference's tests trips over this changes from run to run
 * (at the lvl2 set: test_FDFB_CLOT21 / _CLOT21_2, test_circuit_bootstrap, with the reference's own library as with this one,
 * profiles/r03_reference_tests_lvl2_*.txt).  oracle/ref/Makefile's `lvl2w` builds of the suite swap these two definitions in ON THE FLY (sed | gcc -x c -)
 * and pre-include this header; the test functions themselves stay the reference's.
 */
#ifndef MOSFHET_UNITY_TORUS_ASSERTS_H
#define MOSFHET_UNITY_TORUS_ASSERTS_H
#include <stdint.h>

static inline uint64_t mosfhet_torus_dist(uint64_t a, uint64_t b) {
  const uint64_t d = a - b;
  return d > (uint64_t)0 - d ? (uint64_t)0 - d : d;
}
#define MOSFHET_TORUS_WITHIN_MESSAGE(delta, expected, actual, message) \
  TEST_ASSERT_MESSAGE(mosfhet_torus_dist((uint64_t)(expected), (uint64_t)(actual)) <= (uint64_t)(delta), message)
#define MOSFHET_TORUS_ARRAY_WITHIN_MESSAGE(delta, expected, actual, num_elements, message)                                                   \
  do {                                                                                                                                       \
    const uint64_t *mte_ = (const uint64_t *)(expected), *mta_ = (const uint64_t *)(actual);                                                 \
    for (size_t mti_ = 0; mti_ < (size_t)(num_elements); mti_++)                                                                            \
      TEST_ASSERT_MESSAGE(mosfhet_torus_dist(mte_[mti_], mta_[mti_]) <= (uint64_t)(delta), message);                                         \
  } while (0)
#endif
