/* Host-only helpers of the drop-in layer (torus polynomials, rotations, digits, TRGSW constructors, secret-key distributions, exact products) exercised once each:
 * built by tests/test_host_and_abi.py::test_host_helpers_are_clean_under_sanitizers from the host layer's SOURCES with AddressSanitizer, UndefinedBehaviorSanitizer
 * and the leak checker (no GPU involved: nothing here launches a kernel). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mosfhet.h>
int main(void) {
  const int N = 256, l = 3, Bg = 7;
  mosfhet_seed(7);
  TorusPolynomial a = polynomial_new_torus_polynomial(N), b = polynomial_new_torus_polynomial(N), c = polynomial_new_torus_polynomial(N);
  generate_random_bytes(sizeof(Torus) * N, (uint8_t *)a->coeffs);
  generate_random_bytes(sizeof(Torus) * N, (uint8_t *)b->coeffs);
  for (int r = -3; r < 2 * N + 3; r += 7) { torus_polynomial_mul_by_xai(c, a, r); torus_polynomial_mul_by_xai_addto(c, b, r); torus_polynomial_mul_by_xai_minus_1(c, a, r); }
  polynomial_naive_mul_torus(c, a, b); polynomial_naive_mul_addto_torus(c, a, b); polynomial_full_mul_with_scale(c, a, b, 64, 60);
  TorusPolynomial *dec = polynomial_new_array_of_torus_polynomials(N, 2 * l);
  polynomial_decompose(dec, a, Bg, l);
  for (int i = 0; i < l; i++) polynomial_decompose_i(dec[i], a, Bg, l, i);
  TRLWE_Key key = trlwe_new_bounded_key(N, 1, 16, 1e-9), tk = trlwe_new_ternary_key(N, 1, 32, 1e-9), sk = trlwe_new_sparse_generic_key(N, 1, 16, 8, 1e-9), gk = trlwe_new_gaussian_key(N, 1, 3.0, 1e-9);
  TRLWE_Key sg = trlwe_new_sparse_gaussian_key(N, 1, 16, 3.0, 1e-9), sb = trlwe_new_sparse_binary_key(N, 1, 16, 1e-9), st = trlwe_new_sparse_ternary_key(N, 1, 16, 1e-9);
  TLWE_Key lk = tlwe_new_bounded_key(64, 4, 1e-6);
  TRGSW_Key gkey = trgsw_new_key(key, l, Bg);
  TRLWE s1 = trlwe_new_sample(a, key), s2 = trlwe_new_compressed_sample(b, key), s3 = trlwe_alloc_new_sample(1, N);
  trlwe_compressed_subto(s1, s2); trlwe_mul_by_xai_addto(s3, s1, 5); trlwe_mul_by_xai_minus_1(s3, s1, N + 5); trlwe_scale(s3, s1, 3); trlwe_decompose(dec, s1, Bg, l);
  uint64_t lut[4] = {1, 2, 3, 0}; trlwe_LUT_packing(s3, lut, 2, 3);
  TRGSW g1 = trgsw_new_sample(1, gkey), g2 = trgsw_new_exp_sample(5, gkey), g3 = trgsw_new_noiseless_trivial_sample(1, l, Bg, 1, N), g4 = trgsw_new_monomial_sample(-1, N + 3, gkey);
  trgsw_add(g3, g1, g2); trgsw_sub(g3, g1, g2); trgsw_addto(g3, g4); trgsw_copy(g3, g1); trgsw_mul_by_xai(g3, g1, 9); trgsw_mul_by_xai_addto(g3, g2, 2 * N - 1); trgsw_mul_by_xai_minus_1(g3, g4, 1);
  trgsw_naive_mul_trlwe(s3, s1, g2); trgsw_naive_mul(g3, g1, g2);
  printf("exponent of TRGSW(X^5): %lu\n", (unsigned long)_debug_trgsw_decrypt_exp_sample(g2, gkey));
  BinaryPolynomial bp = polynomial_new_binary_polynomial(N), bq = polynomial_new_binary_polynomial(N), br = polynomial_new_binary_polynomial(N);
  for (int i = 0; i < N; i++) { bp->coeffs[i] = i & 1; bq->coeffs[i] = (i >> 1) & 1; }
  polynomial_naive_mul_binary(br, bp, bq); polynomial_naive_mul_addto_torus_binary(c, a, bp);
  polynomial_torus_scale(c, a, 5); polynomial_torus_scale2(c, a, 5); polynomial_negate_torus_polynomial(c, a); polynomial_copy_torus_polynomial(c, a); polynomial_zero_torus_polynomial(c);
  polynomial_add_torus_polynomials(c, a, b); polynomial_sub_torus_polynomials(c, a, b); polynomial_addto_torus_polynomial(c, a); polynomial_subto_torus_polynomial(c, a);
  free(bp->coeffs); free(bp); free(bq->coeffs); free(bq); free(br->coeffs); free(br);
  free_trgsw(g1); free_trgsw(g2); free_trgsw(g3); free_trgsw(g4); free_trlwe(s1); free_trlwe(s2); free_trlwe(s3);
  free_trgsw_key(gkey); free_tlwe_key(lk);
  free_trlwe_key(key); free_trlwe_key(tk); free_trlwe_key(sk); free_trlwe_key(gk); free_trlwe_key(sg); free_trlwe_key(sb); free_trlwe_key(st);
  free_array_of_polynomials(dec, 2 * l); free_polynomial(a); free_polynomial(b); free_polynomial(c);
  printf("host helpers ok\n");
  return 0;
}
