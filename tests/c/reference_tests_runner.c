/*
 * Runs ONE test function of the reference's own Unity test-suite (test/tests.c, compiled unchanged with -Dmain=mosfhet_reference_tests_main and linked
 * to libmosfhet_hip.so by oracle/ref/Makefile, target `app`): reference_tests_hip <test name>.  One process per test, so that a test which aborts (the
 * reference's failure mode is assert / exit) does not take the others with it; tests/test_gpu_parity.py::test_reference_test_suite_on_the_gpu walks
 * the list the reference's main() runs.  The reference's source is not modified and not copied: this file only looks its test functions up by name.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "unity.h"

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <test function of test/tests.c>\n", argv[0]);
    return 2;
  }
  void (*fn)(void) = (void (*)(void))dlsym(RTLD_DEFAULT, argv[1]);
  if (!fn) {
    fprintf(stderr, "no such test: %s\n", argv[1]);
    return 2;
  }
  UnityBegin("test/tests.c");
  UnityDefaultTestRun(fn, argv[1], 0);
  return UnityEnd();
}
