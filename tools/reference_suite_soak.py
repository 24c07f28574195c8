"""Failure counts of the reference's own test-suite (test/tests.c, compiled unchanged by oracle/ref/Makefile) over many runs, one process per test and run.

    python tools/reference_suite_soak.py <binary under oracle/_ref> <passes> [first test ... ]

reference_tests_ref / reference_tests_lvl2w_ref: the reference's own library on the CPU (build container); reference_tests_hip / reference_tests_lvl2w_hip: the same
program linked to this repository's library (GPU box).  Every test draws fresh keys, messages and noise per run, and several of its bounds are a few sigma wide: this
is how the named exceptions of tests/test_gpu_parity.py::test_reference_test_suite_on_the_gpu were measured (profiles/r05_reference_suite_soak_*.txt)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", sys.argv[1])
    passes = int(sys.argv[2])
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    names = sys.argv[3:] or re.search(r'REFERENCE_SUITE = """(.*?)"""\.split\(\)', src, re.S).group(1).split()
    if sys.argv[1].endswith("_ref") and not sys.argv[3:]:
        names = [n for n in names if n != "test_trlwe_full_packing_ks"]      # the reference's own library segfaults there
    fails, first = collections.Counter(), {}
    for _ in range(passes):
        for n in names:
            p = subprocess.run([exe, n], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
            if p.returncode != 0:
                fails[n] += 1
                first.setdefault(n, ([ln for ln in p.stdout.splitlines() if "FAIL" in ln] or ["rc %d" % p.returncode])[0][:260])
    print("%s: %d passes of %d tests" % (sys.argv[1], passes, len(names)))
    for n in names:
        if fails[n]:
            print("  %-44s %4d / %d   %s" % (n, fails[n], passes, first[n]))
    print("  every other test: 0 / %d" % passes)


if __name__ == "__main__":
    main()
