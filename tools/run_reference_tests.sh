#!/bin/bash
# tools/run_reference_tests.sh [timeout seconds per test]: the 41 tests the reference's test/tests.c main() runs, one process each (oracle/_ref/reference_tests_hip)
T=${1:-300}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TESTS="test_functional_mv_bootstrap_UBR test_trgsw_trlwe_mul test_normal_generator test_tlwe test_tlwe_ks test_trlwe_ks test_blind_rotate test_functional_bootstrap_ga
test_functional_bootstrap_ga_bounded_key test_functional_bootstrap test_poly_DFT test_poly_DFT_mul test_tlwe_pack1_ks_CDKS21 test_trlwe test_trgsw test_trgsw_sub
test_trgsw_mul_by_xai test_trgsw_dft test_trgsw_mul test_functional_mv_bootstrap test_programmable_bootstrap test_FDFB_KS21 test_FDFB_new test_FDFB_CLOT21
test_FDFB_CLOT21_2 test_tlwe_mul test_trlwe_mul test_trlwe_poly_mul test_public_mux test_compressed_trlwe test_trlwe_full_packing_ks test_multivalue_bootstrap_CLOT21
test_trlwe_packing_ks test_circuit_bootstrap test_functional_bootstrap_with_encrypted_LUT test_tlwe_pack_key_priv_ks test_tlwe_pack1_ks test_trgsw_reg_sub
test_functional_bootstrap_trgsw test_functional_bootstrap_unfolded test_trlwe_pack_key_priv_ks"
pass=0; fail=0
for t in $TESTS; do
  s=$(date +%s%N)
  out=$(timeout $T ${EXE:-$ROOT/oracle/_ref/reference_tests_hip} $t 2>&1); rc=$?
  e=$(date +%s%N)
  if [ $rc -eq 0 ]; then pass=$((pass+1)); st=PASS; else fail=$((fail+1)); st="FAIL(rc=$rc)"; fi
  printf "%-48s %-12s %6d ms\n" $t "$st" $(( (e - s) / 1000000 ))
  if [ $rc -ne 0 ]; then echo "$out" | grep -v "^$" | grep -v "^---\|Tests .* Failures\|^FAIL$" | tail -2 | cut -c1-300 | sed 's/^/      /'; fi
done
echo "passed $pass of $((pass+fail))"
