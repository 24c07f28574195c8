/*
 * mosfhet_compat_extra.c -- what the reference's own test-suite (test/tests.c) needs to LINK beyond the path of SURVEY.md section 8: the TRGSW registers,
 * the debug decryptors, the exact schoolbook TRGSW products, bounded secret keys and the compressed-sample entry points.  Thin compositions of calls
 * this library already has (TRGSW products run on the device through trgsw_mul_DFT2; everything torus-domain is exact host arithmetic); no new kernels.
 * They exist so that the reference's tests of the PATH functions can run unchanged against the product library (tests/test_gpu_parity.py::
 * test_reference_test_suite_on_the_gpu); nothing here is tuned.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "compat_internal.h"

#define W 64

/* ------------------------------------------------------------------ secret keys with coefficients in (-bound/2, bound/2]  (src/tlwe.c:70-78, src/trlwe.c:119-130) */
static void bounded_words(Torus *s, int count, uint64_t bound) {
  mc_rnd_bytes(s, sizeof(Torus) * (size_t)count);
  for (int i = 0; i < count; i++) s[i] = (s[i] & (bound - 1)) - ((bound >> 1) - 1);
}
TLWE_Key tlwe_new_bounded_key(int n, uint64_t bound, double sigma) {
  TLWE_Key key = tlwe_alloc_key(n, sigma);
  bounded_words(key->s, n, bound);
  return key;
}
TRLWE_Key trlwe_new_bounded_key(int N, int k, uint64_t bound, double sigma) {
  TRLWE_Key key = trlwe_alloc_key(N, k, sigma);
  for (int i = 0; i < k; i++) bounded_words(key->s[i]->coeffs, N, bound);
  return key;
}

/* ------------------------------------------------------------------ "compressed" samples (src/trlwe_compressed*.c): the reference stores the mask as a
 * PRNG seed and expands it when the sample is used -- a storage format of ITS table keys.  Through this API a compressed sample is an ordinary sample. */
TRLWE trlwe_new_compressed_sample(TorusPolynomial m, TRLWE_Key key) { return trlwe_new_sample(m, key); }
void trlwe_compressed_subto(TRLWE out, TRLWE in) { trlwe_subto(out, in); }

/* ------------------------------------------------------------------ exact TRGSW products (src/trgsw.c:456-480): digits times rows by schoolbook products */
void trgsw_naive_mul_trlwe(TRLWE out, TRLWE in1, TRGSW in2) {
  const int N = in1->b->N, l = in2->l, k = in1->k, rows = (k + 1) * l;
  TorusPolynomial *dec = polynomial_new_array_of_torus_polynomials(N, rows);
  trlwe_decompose(dec, in1, in2->Bg_bit, l);
  for (int p = 0; p <= k; p++) {
    TorusPolynomial dst = p < k ? out->a[p] : out->b;
    polynomial_zero_torus_polynomial(dst);
    for (int r = 0; r < rows; r++) polynomial_naive_mul_addto_torus(dst, dec[r], p < k ? in2->samples[r]->a[p] : in2->samples[r]->b);
  }
  free_array_of_polynomials(dec, rows);
}
void trgsw_naive_mul(TRGSW out, TRGSW in1, TRGSW in2) {
  const int rows = (in1->samples[0]->k + 1) * in2->l;
  for (int r = 0; r < rows; r++) trgsw_naive_mul_trlwe(out->samples[r], in1->samples[r], in2);
}

/* ------------------------------------------------------------------ debug decryptors (src/trgsw.c:190-268): the exponent e of a TRGSW(X^e) sample */
static uint64_t exponent_of(TorusPolynomial phase, int Bg_bit, int complain) {
  const Torus delta = (Torus)1 << (W - 1 - Bg_bit);
  int found = -1;
  for (int j = 0; j < phase->N; j++) {
    const Torus c = phase->coeffs[j];
    if (c > delta && c < (Torus)0 - delta) {   /* |c| > delta: a coefficient carrying the gadget value */
      if (found != -1) {
        if (complain) printf("[TRGSW Exp Decryption error] Current: %lf*x^%d - Previous: %lf*x^%d\n", torus2double(c), j, torus2double(phase->coeffs[found]), found);
        return (uint64_t)-1;
      }
      found = j;
    }
  }
  return (uint64_t)(int64_t)found;
}
uint64_t _debug_trgsw_decrypt_exp_sample(TRGSW c, TRGSW_Key key) {
  const int N = key->trlwe_key->s[0]->N;
  TorusPolynomial ph = polynomial_new_torus_polynomial(N);
  trlwe_phase(ph, c->samples[key->l], key->trlwe_key);      /* row k l: the b-component row of level 0, message X^e 2^(W - Bg_bit) */
  const uint64_t r = exponent_of(ph, key->Bg_bit, 0);
  free_polynomial(ph);
  return r;
}
uint64_t _debug_trgsw_decrypt_exp_DFT_sample(TRGSW_DFT c, TRGSW_Key key) {
  const int N = key->trlwe_key->s[0]->N, k = key->trlwe_key->k;
  TRLWE probe = trlwe_new_noiseless_trivial_sample(NULL, k, N);
  probe->b->coeffs[0] = (Torus)1 << (W - key->Bg_bit);
  TRLWE_DFT res = trlwe_alloc_new_DFT_sample(k, N);
  trgsw_mul_trlwe_DFT(res, probe, c);
  trlwe_from_DFT(probe, res);
  TorusPolynomial ph = polynomial_new_torus_polynomial(N);
  trlwe_phase(ph, probe, key->trlwe_key);
  const uint64_t r = exponent_of(ph, key->Bg_bit, 1);
  free_polynomial(ph);
  free_trlwe(res);
  free_trlwe(probe);
  return r;
}

/* ------------------------------------------------------------------ TRGSW registers (src/register.c): an exponent kept as TRGSW_DFT(X^m) and TRGSW_DFT(X^-m), so
 * that subtraction is a product too */
TRGSW_REG trgsw_reg_alloc(int l, int Bg_bit, int k, int N) {
  TRGSW_REG r = (TRGSW_REG)mc_xmalloc(sizeof(*r));
  r->positive = trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N);
  r->negative = trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N);
  return r;
}
TRGSW_REG *trgsw_reg_alloc_array(int count, int l, int Bg_bit, int k, int N) {
  TRGSW_REG *r = (TRGSW_REG *)mc_xmalloc(sizeof(TRGSW_REG) * (size_t)(count > 0 ? count : 1));
  for (int i = 0; i < count; i++) r[i] = trgsw_reg_alloc(l, Bg_bit, k, N);
  return r;
}
void free_trgsw_reg(TRGSW_REG p) {
  if (!p) return;
  free_trgsw(p->negative);
  free_trgsw(p->positive);
  free(p);
}
void free_trgsw_reg_array(TRGSW_REG *p, int count) {
  if (!p) return;
  for (int i = 0; i < count; i++) free_trgsw_reg(p[i]);
  free(p);
}
void trgsw_reg_sample(TRGSW_REG out, Torus m, TRGSW_Key key) {
  const int N = key->trlwe_key->s[0]->N;
  trgsw_monomial_DFT_sample(out->positive, 1, (int)m, key);
  trgsw_monomial_DFT_sample(out->negative, 1, N - (int)m, key);
}
void trgsw_reg_copy(TRGSW_REG out, TRGSW_REG in) {
  trgsw_DFT_copy(out->positive, in->positive);
  trgsw_DFT_copy(out->negative, in->negative);
}
void trgsw_reg_negate(TRGSW_REG reg) {
  TRGSW_DFT t = reg->positive;
  reg->positive = reg->negative;
  reg->negative = t;
}
void trgsw_reg_add(TRGSW_REG out, TRGSW_REG in1, TRGSW_REG in2) {
  trgsw_mul_DFT2(out->positive, in1->positive, in2->positive);
  trgsw_mul_DFT2(out->negative, in1->negative, in2->negative);
}
void trgsw_reg_addto(TRGSW_REG out, TRGSW_REG in) { trgsw_reg_add(out, out, in); }
void trgsw_reg_sub(TRGSW_REG out, TRGSW_REG in1, TRGSW_REG in2) {
  trgsw_mul_DFT2(out->positive, in1->positive, in2->negative);
  trgsw_mul_DFT2(out->negative, in1->negative, in2->positive);
}
void trgsw_reg_subto(TRGSW_REG out, TRGSW_REG in) { trgsw_reg_sub(out, out, in); }

/* ------------------------------------------------------------------ more secret-key distributions (src/trlwe.c:137-228): sparse / ternary / Gaussian coefficients.
 * The device key generators take any small-integer TRLWE key (keygen_kernels.h). */
static void sparse_fill(Torus *out, int size, int h, int ternary, int gaussian, double key_sigma) {
  memset(out, 0, sizeof(Torus) * (size_t)size);
  if (h > size) {   /* more nonzero coefficients than positions: the loop below would never end */
    fprintf(stderr, "mosfhet_amd: key with Hamming weight %d over %d coefficients\n", h, size);
    abort();
  }
  Torus val = 1;
  for (int hw = 0; hw < h;) {
    /* unbiased index for ANY size (k * N need not be a power of two): rejection below the largest multiple of size */
    const uint64_t limit = UINT64_MAX - UINT64_MAX % (uint64_t)size, r = mc_rnd64();
    if (r >= limit) continue;
    const int idx = (int)(r % (uint64_t)size);
    if (out[idx]) continue;
    if (gaussian) {
      val = (Torus)(int64_t)mc_rnd_normal(key_sigma);
      if (!val) continue;                 /* a drawn 0 is not a nonzero coefficient: draw again */
    }
    out[idx] = val;
    if (ternary) val = (Torus)0 - val;    /* +1, -1, +1, ... */
    hw++;
  }
}
TRLWE_Key trlwe_new_ternary_key(int N, int k, int h, double sigma) {
  TRLWE_Key key = trlwe_alloc_key(N, k, sigma);
  for (int i = 0; i < k; i++) sparse_fill(key->s[i]->coeffs, N, h, 1, 0, 0);
  return key;
}
TRLWE_Key trlwe_new_sparse_ternary_key(int N, int k, int h, double sigma) {   /* h nonzero coefficients over all k polynomials together */
  TRLWE_Key key = trlwe_alloc_key(N, k, sigma);
  Torus *all = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)k * N);
  sparse_fill(all, k * N, h, 1, 0, 0);
  for (int i = 0; i < k; i++) memcpy(key->s[i]->coeffs, all + (size_t)i * N, sizeof(Torus) * (size_t)N);
  free(all);
  return key;
}
TRLWE_Key trlwe_new_sparse_binary_key(int N, int k, int h, double sigma) {
  TRLWE_Key key = trlwe_alloc_key(N, k, sigma);
  for (int i = 0; i < k; i++) sparse_fill(key->s[i]->coeffs, N, h, 0, 0, 0);
  return key;
}
TRLWE_Key trlwe_new_sparse_gaussian_key(int N, int k, int h, double key_sigma, double noise_sigma) {
  TRLWE_Key key = trlwe_new_sparse_binary_key(N, k, h, noise_sigma);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < N; j++)
      if (key->s[i]->coeffs[j] == 1) {
        key->s[i]->coeffs[j] = (Torus)(int64_t)mc_rnd_normal(key_sigma);
        if (!key->s[i]->coeffs[j]) key->s[i]->coeffs[j] = 1;
      }
  return key;
}
TRLWE_Key trlwe_new_sparse_generic_key(int N, int k, int h, uint64_t key_bound, double noise_sigma) {
  TRLWE_Key key = trlwe_new_sparse_binary_key(N, k, h, noise_sigma);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < N; j++)
      if (key->s[i]->coeffs[j] == 1) {
        key->s[i]->coeffs[j] = (mc_rnd64() & (key_bound - 1)) - ((key_bound >> 1) - 1);
        if (!key->s[i]->coeffs[j]) key->s[i]->coeffs[j] = 1;
      }
  return key;
}
TRLWE_Key trlwe_new_gaussian_key(int N, int k, double key_sigma, double noise_sigma) {
  TRLWE_Key key = trlwe_alloc_key(N, k, noise_sigma);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < N; j++) key->s[i]->coeffs[j] = (Torus)(int64_t)mc_rnd_normal(key_sigma);
  return key;
}

/* ------------------------------------------------------------------ small conveniences (src/trlwe.c:333-370, src/polynomial.c:29-38,237-250,305-317) */
void print_trlwe_msg(TRLWE in, uint64_t prec, TRLWE_Key key) {
  const int N = in->b->N;
  TorusPolynomial ph = polynomial_new_torus_polynomial(N);
  trlwe_phase(ph, in, key);
  for (int i = 0; i < N; i++) printf(i + 1 < N ? "%lu, " : "%lu\n", (unsigned long)torus2int(ph->coeffs[i], (int)prec));
  free_polynomial(ph);
}
uint64_t _debug_trlwe_decrypt_exp_sample(TRLWE c, uint64_t prec, TRLWE_Key key) {   /* index of the one coefficient that carries a message of `prec` bits */
  const int N = key->s[0]->N;
  const Torus delta = (Torus)1 << (W - prec - 1);
  TorusPolynomial ph = polynomial_new_torus_polynomial(N);
  trlwe_phase(ph, c, key);
  int found = -1;
  for (int j = 0; j < N; j++)
    if (ph->coeffs[j] > delta && ph->coeffs[j] < (Torus)0 - delta) {
      if (found != -1) { found = -1; break; }
      found = j;
    }
  if (found == -1) {
    printf("\nTRGSW error: no value\n");
    exit(0);     /* the reference prints the phase and exits, src/trlwe.c:359-367 */
  }
  free_polynomial(ph);
  return (uint64_t)found;
}

BinaryPolynomial polynomial_new_binary_polynomial(int N) {
  BinaryPolynomial p = (BinaryPolynomial)mc_xmalloc(sizeof(*p));
  p->coeffs = (Binary *)mc_xmalloc(sizeof(Binary) * (size_t)N);
  p->N = N;
  return p;
}
void polynomial_naive_mul_binary(BinaryPolynomial out, BinaryPolynomial in1, BinaryPolynomial in2) {
  const int N = in2->N;
  for (int j = 0; j < N; j++) out->coeffs[j] = 0;
  for (int i = 0; i < N; i++) {
    if (!in2->coeffs[i]) continue;
    for (int j = 0; j < N - i; j++) out->coeffs[i + j] += in1->coeffs[j] * in2->coeffs[i];
    for (int j = N - i; j < N; j++) out->coeffs[i + j - N] -= in1->coeffs[j] * in2->coeffs[i];
  }
}
void polynomial_naive_mul_addto_torus_binary(TorusPolynomial out, TorusPolynomial in1, BinaryPolynomial in2) {
  const int N = in2->N;
  for (int i = 0; i < N; i++) {
    if (!in2->coeffs[i]) continue;
    for (int j = 0; j < N - i; j++) out->coeffs[i + j] += in1->coeffs[j];
    for (int j = N - i; j < N; j++) out->coeffs[i + j - N] -= in1->coeffs[j];
  }
}

/* ------------------------------------------------------------------ the rest of the "compressed" entry points: ordinary samples here (see above); their files hold
 * whole samples, so they are this library's own, like every DFT-domain file */
void trlwe_save_compressed_sample(FILE *fd, TRLWE c) { trlwe_save_sample(fd, c); }
void trlwe_load_compressed_sample(FILE *fd, TRLWE c) { trlwe_load_sample(fd, c); }
TRLWE trlwe_load_new_compressed_sample(FILE *fd, int k, int N) { return trlwe_load_new_sample(fd, k, N); }
void trlwe_compressed_DFT_sample(TRLWE_DFT out, TorusPolynomial m, TRLWE_Key key) {
  TRLWE c = trlwe_new_sample(m, key);
  trlwe_to_DFT(out, c);
  free_trlwe(c);
}
TRLWE_DFT trlwe_new_compressed_DFT_sample(TorusPolynomial m, TRLWE_Key key) {
  TRLWE_DFT out = trlwe_alloc_new_DFT_sample(key->k, key->s[0]->N);
  trlwe_compressed_DFT_sample(out, m, key);
  return out;
}
void trlwe_compressed_DFT_mul_addto(TRLWE_DFT out, DFT_Polynomial in1, TRLWE_DFT in2) { trlwe_DFT_mul_addto_by_polynomial(out, in2, in1); }

/* ------------------------------------------------------------------ key switch without precomputed multiples (src/tlwe.c:214-230,305-320): one sample per (input
 * word, digit position); the digit multiplies it on the device (mosfhet_hip_tlwe_keyswitch_no_precomp_batch).  The rows live in device memory: `s` is NULL. */
TLWE_KS_Key_m tlwe_new_KS_key_no_precomp(TLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  const int n_in = in_key->n, n_out = out_key->n;
  TLWE_KS_Key_m key = (TLWE_KS_Key_m)mc_xmalloc(sizeof(*key));
  key->s = NULL; key->base_bit = base_bit; key->t = t; key->n = n_in; key->n_out = n_out;
  const size_t row = (size_t)n_out + 1, words = (size_t)n_in * t * row;
  Torus *h = (Torus *)mc_xmalloc(sizeof(Torus) * words);
  TLWE c = tlwe_alloc_sample(n_out);
  for (int i = 0; i < n_in; i++)
    for (int j = 0; j < t; j++) {
      tlwe_sample(c, in_key->s[i] * ((Torus)1 << (W - (j + 1) * base_bit)), out_key);
      memcpy(h + ((size_t)i * t + j) * row, c->a, sizeof(Torus) * (size_t)n_out);
      h[((size_t)i * t + j) * row + n_out] = c->b;
    }
  free_tlwe(c);
  (void)mosfhet_engine_ctx();
  key->device = mc_dev_alloc(sizeof(Torus) * words);
  mc_dev_copy(key->device, h, sizeof(Torus) * words, HIP_H2D);
  free(h);
  return key;
}

void tlwe_keyswitch_no_precomp(TLWE out, TLWE in, TLWE_KS_Key_m key) {
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int n_in = key->n, n_out = key->n_out;
  const size_t in_w = (size_t)n_in + 1, out_w = (size_t)n_out + 1;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + out_w)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + out_w));
  memcpy(h, in->a, sizeof(Torus) * (size_t)n_in);
  h[n_in] = in->b;
  mc_dev_copy(d, h, sizeof(Torus) * in_w, HIP_H2D);
  if (mosfhet_hip_tlwe_keyswitch_no_precomp_batch(ctx, (const uint64_t *)key->device, d + in_w, d, 1, n_in, n_out, key->t, key->base_bit, NULL) || mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("tlwe_keyswitch_no_precomp");
  mc_dev_copy(h + in_w, d + in_w, sizeof(Torus) * out_w, HIP_D2H);
  memcpy(out->a, h + in_w, sizeof(Torus) * (size_t)n_out);
  out->b = h[in_w + n_out];
  mc_hstage_free(h);
}

/* ------------------------------------------------------------------ exact 128-bit products (src/polynomial.c:428-437, src/fft/karatsuba.c:92-102): the full product of the
 * two coefficient vectors in arithmetic mod 2^128, every coefficient shifted right by bit_scale, THEN folded negacyclically.  The reference multiplies by
 * Karatsuba; Karatsuba's identities hold in Z / 2^128, so the schoolbook product below gives the same words. */
void polynomial_full_mul_with_scale(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2, int bit_size, int bit_scale) {
  (void)bit_size;
  const int N = in1->N;
  unsigned __int128 *prod = (unsigned __int128 *)mc_xmalloc(sizeof(unsigned __int128) * (size_t)2 * N);
  memset(prod, 0, sizeof(unsigned __int128) * (size_t)2 * N);
  for (int i = 0; i < N; i++) {
    const unsigned __int128 x = in1->coeffs[i];
    if (!x) continue;
    for (int j = 0; j < N; j++) prod[i + j] += x * in2->coeffs[j];
  }
  for (int i = 0; i < N; i++) out->coeffs[i] = (Torus)(prod[i] >> bit_scale) - (Torus)(prod[N + i] >> bit_scale);
  free(prod);
}

/* src/trlwe.c:692-713: tensor product of two samples with those exact products, relinearised by one FFT key switch */
void trlwe_tensor_prod(TRLWE out, TRLWE in1, TRLWE in2, int precision, TRLWE_KS_Key rl_key) {
  const int N = in1->b->N, scale = W - precision;
  TorusPolynomial tmp = polynomial_new_torus_polynomial(N);
  TRLWE t = trlwe_alloc_new_sample(1, N);
  polynomial_full_mul_with_scale(t->a[0], in1->a[0], in2->a[0], W, scale);
  memset(t->b->coeffs, 0, sizeof(Torus) * (size_t)N);
  polynomial_full_mul_with_scale(out->a[0], in1->a[0], in2->b, W, scale);
  polynomial_full_mul_with_scale(tmp, in1->b, in2->a[0], W, scale);
  polynomial_addto_torus_polynomial(out->a[0], tmp);
  polynomial_full_mul_with_scale(out->b, in1->b, in2->b, W, scale);
  trlwe_keyswitch(t, t, rl_key);
  trlwe_subto(out, t);
  free_polynomial(tmp);
  free_trlwe(t);
}
