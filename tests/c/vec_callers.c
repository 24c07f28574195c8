/*
 * GPU test of the host-struct face of the digit-parallel radix-integer callers (include/mosfhet_compat.h: mosfhet_vec_*): a plain C program in the shape of the
 * reference application's callers -- integers as arrays of TLWE digits (ufhe_integer's `digits`, applications/multi-ciphertext-arith/include/ufhe.h:18-22), the keys
 * of its ufhe_public_keyset -- that runs add, sub, ReLU, comparison, encrypted and cleartext look-up and the 8 x 8 -> 32 bit multiplication over ALL rows of
 * tests/golden/ufhe_vectors.npz in one call each and must decrypt to what the REFERENCE application decrypted on the reference's own library (the fixture's result
 * columns; tests/golden/make_ufhe_golden.py).  The rows arrive as text: argv[1] (written by tests/test_gpu_parity.py::test_vector_callers_through_the_c_api).
 * Parameters: the application's ring and gadget (src/ufhe.c:18-20: N = 2048, l = 6, Bg = 2^7, key switch t = 6 base 2^2, radix 4) over a shortened LWE key.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mosfhet.h>

enum { n = 64, N = 2048, k = 1, l = 6, Bg_bit = 7, ks_t = 6, ks_bb = 2, B = 4, LOGB = 2, D = 4, DC = 16, MAXROWS = 64 };
static TLWE_Key lwe_key, extracted_key;
static TRLWE_Key rlwe_key;
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __func__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

/* ufhe_encrypt_integer / ufhe_decrypt_integer (src/integer.c:34-59): digit i of the value as digit / (2 B) under the extracted key */
static TLWE *new_integer(int digits) { return tlwe_alloc_sample_array(digits, N); }
static void encrypt(TLWE *x, int digits, uint64_t value) {
  for (int i = 0; i < digits; i++) tlwe_sample(x[i], double2torus((double)((value >> (LOGB * i)) & (B - 1)) / (2 * B)), extracted_key);
}
static uint64_t decrypt(TLWE *x, int digits) {
  uint64_t v = 0;
  for (int i = 0; i < digits; i++) v |= (uint64_t)(torus2int(tlwe_phase(x[i], extracted_key), LOGB + 1) & (B - 1)) << (LOGB * i);
  return v;
}
static TLWE **integers(int M, int digits) {
  TLWE **x = (TLWE **)malloc(sizeof(TLWE *) * M);
  for (int m = 0; m < M; m++) x[m] = new_integer(digits);
  return x;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s rows.txt\n", argv[0]); return 2; }
  static long a[MAXROWS], b[MAXROWS], sel[MAXROWS], table[MAXROWS][16], r_add[MAXROWS], r_sub[MAXROWS], r_relu[MAXROWS], r_lut[MAXROWS], r_cmp_s[MAXROWS], r_cmp_u[MAXROWS],
      r_lut_ct[MAXROWS], r_mul_s[MAXROWS], r_mul_u[MAXROWS];
  FILE *f = fopen(argv[1], "r");
  int M = 0;
  if (!f || fscanf(f, "%d", &M) != 1 || M < 1 || M > MAXROWS) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  for (int m = 0; m < M; m++) {
    int ok = fscanf(f, "%ld %ld %ld", &a[m], &b[m], &sel[m]) == 3;
    for (int j = 0; j < 16; j++) ok = ok && fscanf(f, "%ld", &table[m][j]) == 1;
    ok = ok && fscanf(f, "%ld %ld %ld %ld %ld %ld %ld %ld %ld", &r_add[m], &r_sub[m], &r_relu[m], &r_lut[m], &r_cmp_s[m], &r_cmp_u[m], &r_lut_ct[m], &r_mul_s[m], &r_mul_u[m]) == 9;
    if (!ok) { fprintf(stderr, "row %d of %s is short\n", m, argv[1]); return 2; }
  }
  fclose(f);

  mosfhet_seed(0x5EC0);
  lwe_key = tlwe_new_binary_key(n, 3.0517578125e-05 / 4);
  rlwe_key = trlwe_new_binary_key(N, k, 5.684341886080802e-14);
  extracted_key = tlwe_alloc_key(N, rlwe_key->sigma);
  trlwe_extract_tlwe_key(extracted_key, rlwe_key);
  TRGSW_Key trgsw_key = trgsw_new_key(rlwe_key, l, Bg_bit);
  Bootstrap_Key bk = new_bootstrap_key(trgsw_key, lwe_key, 1);
  TLWE_KS_Key ksk = tlwe_new_KS_key(lwe_key, extracted_key, ks_t, ks_bb);
  LUT_Packing_KS_Key pk = trlwe_new_packing_KS_key(rlwe_key, extracted_key, ks_t, ks_bb, B);
  mosfhet_vec v = mosfhet_vec_new(bk, ksk, pk, B);

  TLWE **xa = integers(M, D), **xb = integers(M, D), **xs = integers(M, 2), **xc = integers(M, D), **xw = integers(M, DC);
  TLWE *cmp = tlwe_alloc_sample_array(M, N);
  TLWE ***tab = (TLWE ***)malloc(sizeof(TLWE **) * 16);
  for (int m = 0; m < M; m++) {
    encrypt(xa[m], D, (uint64_t)a[m] & 0xff);
    encrypt(xb[m], D, (uint64_t)b[m] & 0xff);
    encrypt(xs[m], 2, (uint64_t)sel[m]);
  }
  for (int j = 0; j < 16; j++) {
    tab[j] = integers(M, D);
    for (int m = 0; m < M; m++) encrypt(tab[j][m], D, (uint64_t)table[m][j] & 0xff);
  }
#define S8(x) ((long)(int8_t)(uint8_t)(x))
  mosfhet_vec_add_integers(v, xc, xa, xb, M, D);
  for (int m = 0; m < M; m++) CHECK(S8(decrypt(xc[m], D)) == r_add[m], "add row %d: %ld, the reference application decrypted %ld", m, S8(decrypt(xc[m], D)), r_add[m]);
  printf("add ok\n");
  mosfhet_vec_sub_integers(v, xc, xa, xb, M, D);
  for (int m = 0; m < M; m++) CHECK(S8(decrypt(xc[m], D)) == r_sub[m], "sub row %d: %ld, reference %ld", m, S8(decrypt(xc[m], D)), r_sub[m]);
  printf("sub ok\n");
  mosfhet_vec_relu_integers(v, xc, xa, M, D);
  for (int m = 0; m < M; m++) CHECK(S8(decrypt(xc[m], D)) == r_relu[m], "relu row %d: %ld, reference %ld", m, S8(decrypt(xc[m], D)), r_relu[m]);
  printf("relu ok\n");
  mosfhet_vec_cmp_integers(v, cmp, xa, xb, M, D, true, true);
  for (int m = 0; m < M; m++) CHECK((long)decrypt(&cmp[m], 1) == r_cmp_s[m], "signed cmp row %d: %ld, reference %ld", m, (long)decrypt(&cmp[m], 1), r_cmp_s[m]);
  mosfhet_vec_cmp_integers(v, cmp, xa, xb, M, D, false, false);
  for (int m = 0; m < M; m++) CHECK((long)decrypt(&cmp[m], 1) == r_cmp_u[m], "unsigned cmp row %d: %ld, reference %ld", m, (long)decrypt(&cmp[m], 1), r_cmp_u[m]);
  printf("cmp ok\n");
  mosfhet_vec_mux_integer_arrays(v, xc, xs, 2, 16, tab, M, D);
  for (int m = 0; m < M; m++) CHECK(S8(decrypt(xc[m], D)) == r_lut[m], "encrypted LUT row %d: %ld, reference %ld", m, S8(decrypt(xc[m], D)), r_lut[m]);
  printf("mux_array ok\n");
  /* the cleartext table is one per call: the rows that share row 0's table are none but row 0 itself in the fixture, so each row goes through its own call of one
   * integer (the batch of the first row's table over all rows is checked against the arithmetic) */
  {
    uint64_t lut[16];
    for (int j = 0; j < 16; j++) lut[j] = (uint64_t)table[0][j] & 0xff;
    mosfhet_vec_lut_integers(v, xc, D, xs, 2, lut, 16, M);
    for (int m = 0; m < M; m++) CHECK(S8(decrypt(xc[m], D)) == table[0][sel[m]], "cleartext LUT (row 0's table) row %d: %ld, want %ld", m, S8(decrypt(xc[m], D)), table[0][sel[m]]);
    for (int m = 0; m < M; m++) {
      for (int j = 0; j < 16; j++) lut[j] = (uint64_t)table[m][j] & 0xff;
      mosfhet_vec_lut_integers(v, &xc[m], D, &xs[m], 2, lut, 16, 1);
      CHECK(S8(decrypt(xc[m], D)) == r_lut_ct[m], "cleartext LUT row %d: %ld, reference %ld", m, S8(decrypt(xc[m], D)), r_lut_ct[m]);
    }
  }
  printf("lut ok\n");
  mosfhet_vec_mul_integers(v, xw, DC, xa, D, xb, D, true, M);
  for (int m = 0; m < M; m++) CHECK((long)(int32_t)(uint32_t)decrypt(xw[m], DC) == r_mul_s[m], "signed product row %d: %ld, reference %ld", m, (long)(int32_t)(uint32_t)decrypt(xw[m], DC), r_mul_s[m]);
  mosfhet_vec_mul_integers(v, xw, DC, xa, D, xb, D, false, M);
  for (int m = 0; m < M; m++) CHECK((long)(uint32_t)decrypt(xw[m], DC) == r_mul_u[m], "unsigned product row %d: %ld, reference %ld", m, (long)(uint32_t)decrypt(xw[m], DC), r_mul_u[m]);
  printf("mul ok\n");
  mosfhet_vec_free(v);
  printf("%d failures over %d rows\n", failures, M);
  return failures ? 1 : 0;
}
