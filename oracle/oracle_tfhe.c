/*
 * oracle_tfhe.c -- TRGSW external product, blind rotation, functional / programmable
 * bootstrap, plus deterministic key and sample generation for the tests.
 * TEST INFRASTRUCTURE ONLY (see mosfhet_oracle.h).
 */
#include "mosfhet_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define W 64

/* src/trgsw.c:345-349  trgsw_to_DFT: every polynomial of every row */
void orc_trgsw_to_dft(const orc_fft_plan *p, double *out, const Torus *in, int k, int l) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  const size_t polys = (size_t)(k + 1) * l * (k + 1);
  for (size_t q = 0; q < polys; q++) orc_torus_to_dft(p, out + q * N, in + q * N);
}

/* src/trgsw.c:385-423  trgsw_mul_trlwe_DFT.
 * out_DFT[c] = sum_{p<=k} sum_{j<l} DFT(digit_j(in[p])) (.) row[p*l+j][c], rows of a[0] first, b last,
 * levels in increasing j, accumulated in that order (first product, then mul-add). */
static int g_product_order = 0;
/* 0 (default): the reference's order -- one mul-add chain over all (k+1) l rows, as described above.
 * 1: "per input component" -- the l rows of each input component q are chained from zero into a partial sum of their own and the partial sums are
 *    added in component order ((P_0 + P_1) + ...).  The order of a bootstrap that is split over one workgroup per accumulator component
 *    (mosfhet_amd/csrc/bootstrap_kernels.h: pbs_split_kernel), each of which can only see its own component's rows; the reference's result
 *    differs from it by FFT-level rounding only (tests/test_oracle_vs_reference.py holds this order to the reference within the same tolerance).
 * Every composition of this oracle (bootstraps, circuit bootstraps, ...) goes through orc_trgsw_mul_trlwe_dft and follows the switch. */
void orc_set_product_order(int order) { g_product_order = order; }
int orc_get_product_order(void) { return g_product_order; }

static void trgsw_mul_trlwe_dft_by_component(const orc_fft_plan *p, double *out_dft, const Torus *in, const double *trgsw_dft, int k, int l,
                                             int Bg_bit, int N) {
  Torus *dec = (Torus *)malloc(sizeof(Torus) * (size_t)N);
  double *dec_dft = (double *)malloc(sizeof(double) * (size_t)N);
  double *part = (double *)malloc(sizeof(double) * (size_t)(k + 1) * N);
  for (int q = 0; q <= k; q++) {
    memset(part, 0, sizeof(double) * (size_t)(k + 1) * N);
    for (int j = 0; j < l; j++) {
      orc_poly_decompose_i(dec, in + (size_t)q * N, N, Bg_bit, l, j);
      orc_int_to_dft(p, dec_dft, (const int64_t *)dec);
      const double *row = trgsw_dft + (size_t)(q * l + j) * (k + 1) * N;
      for (int c = 0; c <= k; c++) orc_dft_mul_addto(part + (size_t)c * N, dec_dft, row + (size_t)c * N, N);
    }
    for (size_t x = 0; x < (size_t)(k + 1) * N; x++) out_dft[x] = q == 0 ? part[x] : out_dft[x] + part[x];
  }
  free(dec);
  free(dec_dft);
  free(part);
}

void orc_trgsw_mul_trlwe_dft(const orc_fft_plan *p, double *out_dft, const Torus *in,
                             const double *trgsw_dft, int k, int l, int Bg_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  if (g_product_order == 1) {
    trgsw_mul_trlwe_dft_by_component(p, out_dft, in, trgsw_dft, k, l, Bg_bit, N);
    return;
  }
  Torus *dec = (Torus *)malloc(sizeof(Torus) * (size_t)N);
  double *dec_dft = (double *)malloc(sizeof(double) * (size_t)N);
  memset(out_dft, 0, sizeof(double) * (size_t)(k + 1) * N);
  for (int q = 0; q <= k; q++) {
    for (int j = 0; j < l; j++) {
      orc_poly_decompose_i(dec, in + (size_t)q * N, N, Bg_bit, l, j);
      orc_int_to_dft(p, dec_dft, (const int64_t *)dec);
      const double *row = trgsw_dft + (size_t)(q * l + j) * (k + 1) * N;
      for (int c = 0; c <= k; c++) orc_dft_mul_addto(out_dft + (size_t)c * N, dec_dft, row + (size_t)c * N, N);
    }
  }
  free(dec);
  free(dec_dft);
}

/* trgsw_mul_trlwe_DFT followed by trlwe_from_DFT (src/trlwe.c:629-634) */
void orc_external_product(const orc_fft_plan *p, Torus *out, const Torus *in,
                          const double *trgsw_dft, int k, int l, int Bg_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  double *tmp = (double *)malloc(sizeof(double) * (size_t)(k + 1) * N);
  orc_trgsw_mul_trlwe_dft(p, tmp, in, trgsw_dft, k, l, Bg_bit);
  for (int c = 0; c <= k; c++) orc_dft_to_torus(p, out + (size_t)c * N, tmp + (size_t)c * N);
  free(tmp);
}

static int log2_int(int x) {
  int r = 0;
  while ((1 << r) < x) r++;
  return r;
}

/* src/bootstrap.c:107-122  blind_rotate: acc += BK_i (.) (acc * (X^a_i - 1)), skipping a_i == 0 */
void orc_blind_rotate(const orc_fft_plan *p, Torus *acc, const Torus *a, const double *bk_dft,
                      int n, int k, int l, int Bg_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1), log_2N = log2_int(2 * N);
  const size_t trgsw_sz = (size_t)(k + 1) * l * (k + 1) * N;
  Torus *rot = (Torus *)malloc(sizeof(Torus) * (size_t)(k + 1) * N);
  Torus *prod = (Torus *)malloc(sizeof(Torus) * (size_t)(k + 1) * N);
  for (int i = 0; i < n; i++) {
    const int ai = (int)orc_torus2int(a[i], log_2N);
    if (!ai) continue;
    for (int c = 0; c <= k; c++) orc_poly_mul_by_xai_minus_1(rot + (size_t)c * N, acc + (size_t)c * N, N, ai);
    orc_external_product(p, prod, rot, bk_dft + (size_t)i * trgsw_sz, k, l, Bg_bit);
    for (size_t c = 0; c < (size_t)(k + 1) * N; c++) acc[c] += prod[c];
  }
  free(rot);
  free(prod);
}

/* src/bootstrap.c:192-198 */
void orc_functional_bootstrap_wo_extract(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in,
                                         const double *bk_dft, int n, int k, int l, int Bg_bit, int torus_base) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1), log_2N = log2_int(2 * N);
  const Torus prec_offset = orc_double2torus(1. / (4 * torus_base));
  const int rot = 2 * N - (int)orc_torus2int(in[n] + prec_offset, log_2N);
  for (int c = 0; c <= k; c++) orc_poly_mul_by_xai(out + (size_t)c * N, tv + (size_t)c * N, N, rot);
  orc_blind_rotate(p, out, in, bk_dft, n, k, l, Bg_bit);
}

/* src/bootstrap.c:200-206 */
void orc_functional_bootstrap(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in,
                              const double *bk_dft, int n, int k, int l, int Bg_bit, int torus_base) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *acc = (Torus *)malloc(sizeof(Torus) * (size_t)(k + 1) * N);
  orc_functional_bootstrap_wo_extract(p, acc, tv, in, bk_dft, n, k, l, Bg_bit, torus_base);
  orc_trlwe_extract_tlwe(out, acc, k, N, 0);
  free(acc);
}

/* src/bootstrap.c:208-220 */
void orc_programmable_bootstrap(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in,
                                const double *bk_dft, int n, int k, int l, int Bg_bit,
                                int precision, int kappa, int theta) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *tmp = (Torus *)malloc(sizeof(Torus) * (size_t)(n + 1));
  orc_pbs_preprocess(tmp, in, n, N, kappa, theta);
  orc_functional_bootstrap(p, out, tv, tmp, bk_dft, n, k, l, Bg_bit, 1 << (precision - 1));
  free(tmp);
}

/* src/bootstrap.c:519-538  full-domain functional bootstrap ("this work"): a first bootstrap with the constant
 * test vector `sign` extracts the sign of the phase; ct_sign.b -= sign; key switch back to dimension n; add the
 * input; second bootstrap with the user's test vector at doubled torus_base. */
void orc_full_domain_functional_bootstrap(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in,
                                          const double *bk_dft, const Torus *ksk, int n, int k, int l, int Bg_bit,
                                          int t, int base_bit, int precision) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *tv_sign = (Torus *)malloc(sizeof(Torus) * (size_t)(k + 1) * N);
  Torus *ct_sign = (Torus *)malloc(sizeof(Torus) * (size_t)(k * N + 1));
  Torus *in2 = (Torus *)malloc(sizeof(Torus) * (size_t)(n + 1));
  const Torus sign = ((Torus)1 << (W - 2)) - ((Torus)1 << (W - precision - 2));
  orc_trlwe_torus_packing(tv_sign, &sign, k, N, 1);
  orc_functional_bootstrap(p, ct_sign, tv_sign, in, bk_dft, n, k, l, Bg_bit, 1 << (precision - 1));
  ct_sign[(size_t)k * N] -= sign;
  orc_tlwe_keyswitch(in2, ct_sign, ksk, k * N, n, t, base_bit);
  for (int i = 0; i <= n; i++) in2[i] += in[i];
  orc_functional_bootstrap(p, out, tv, in2, bk_dft, n, k, l, Bg_bit, 1 << precision);
  free(tv_sign);
  free(ct_sign);
  free(in2);
}

/* src/bootstrap.c:222-230  one blind rotation at torus_base * n_luts, then n_luts extractions slot_size apart */
void orc_multivalue_bootstrap_CLOT21(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in,
                                     const double *bk_dft, int n, int k, int l, int Bg_bit, int torus_base, int n_luts) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1), slot_size = N / (n_luts * torus_base);
  Torus *acc = (Torus *)malloc(sizeof(Torus) * (size_t)(k + 1) * N);
  orc_functional_bootstrap_wo_extract(p, acc, tv, in, bk_dft, n, k, l, Bg_bit, torus_base * n_luts);
  for (int i = 0; i < n_luts; i++) orc_trlwe_extract_tlwe(out + (size_t)i * (k * N + 1), acc, k, N, i * slot_size);
  free(acc);
}

/* src/trlwe.c:677-687  interleaved packing of n_luts LUTs of lut_size slots */
void orc_trlwe_torus_packing_many_LUT(Torus *out, const Torus *lut, int k, int N, int lut_size, int n_luts) {
  memset(out, 0, sizeof(Torus) * (size_t)(k + 1) * N);
  const int span = N / (lut_size * n_luts);
  for (int i = 0; i < lut_size; i++)
    for (int j = 0; j < n_luts; j++)
      for (int r = 0; r < span; r++) out[(size_t)k * N + (size_t)(i * n_luts + j) * span + r] = lut[j * lut_size + i];
}

/* ------------------------------------------------------------------------------------------
 * FFT-based TRLWE key switch, Galois automorphisms and the GA blind rotation (k = 1).
 * ------------------------------------------------------------------------------------------ */
/* src/keyswitch.c:162-193  out = (0, in.b) - IDFT( sum_j DFT(digit_j(in.a)) (.) KS[j] ), rounded digits of
 * polynomial_decompose_i with (base_bit, t).  In-place use (out == in) is relied upon by trlwe.c:780. */
void orc_trlwe_keyswitch(const orc_fft_plan *p, Torus *out, const Torus *in, const double *ks_dft, int t, int base_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *dec = (Torus *)malloc(sizeof(Torus) * (size_t)N);
  Torus *as = (Torus *)malloc(sizeof(Torus) * (size_t)2 * N);
  double *dec_dft = (double *)malloc(sizeof(double) * (size_t)N);
  double *acc = (double *)calloc((size_t)2 * N, sizeof(double));
  for (int j = 0; j < t; j++) {
    orc_poly_decompose_i(dec, in, N, base_bit, t, j);
    orc_int_to_dft(p, dec_dft, (const int64_t *)dec);
    for (int c = 0; c < 2; c++) orc_dft_mul_addto(acc + (size_t)c * N, dec_dft, ks_dft + ((size_t)j * 2 + c) * N, N);
  }
  for (int c = 0; c < 2; c++) orc_dft_to_torus(p, as + (size_t)c * N, acc + (size_t)c * N);
  for (int i = 0; i < N; i++) {
    const Torus b = in[N + i];
    out[i] = (Torus)0 - as[i];
    out[N + i] = b - as[N + i];
  }
  free(dec);
  free(as);
  free(dec_dft);
  free(acc);
}

/* src/trlwe.c:775-781  X -> X^gen on both components, then key switch back from s(X^gen) to s(X) */
void orc_trlwe_eval_automorphism(const orc_fft_plan *p, Torus *out, const Torus *in, uint64_t gen, const double *ks_dft,
                                 int t, int base_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *tmp = (Torus *)malloc(sizeof(Torus) * (size_t)2 * N);
  orc_poly_permute(tmp, in, N, gen);
  orc_poly_permute(tmp + N, in + N, N, gen);
  orc_trlwe_keyswitch(p, out, tmp, ks_dft, t, base_bit);
  free(tmp);
}

/* src/misc.c:142-159 tabulates the inverses of odd x modulo 2N; computed here by Newton iteration mod 2^k */
uint32_t orc_inverse_mod_2N(uint32_t x, int N) {
  const uint32_t mask = 2u * (uint32_t)N - 1;
  uint32_t inv = x;  /* x * x = 1 mod 8 */
  for (int i = 0; i < 4; i++) inv = (inv * (2u - x * inv)) & mask;
  return inv & mask;
}

/* src/bootstrap_ga.c:39-60  blind rotation with automorphisms; every a_i is forced odd */
void orc_blind_rotate_ga(const orc_fft_plan *p, Torus *acc, const Torus *a, const double *bk_dft, const double *ak_dft,
                         int n, int l, int Bg_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1), log_2N = log2_int(2 * N);
  const uint64_t mod_mask = 2 * (uint64_t)N - 1;
  const size_t trgsw_sz = (size_t)2 * l * 2 * N, ak_sz = (size_t)l * 2 * N;
  Torus *rot = (Torus *)malloc(sizeof(Torus) * (size_t)2 * N);
  const uint64_t w0 = orc_inverse_mod_2N((uint32_t)(orc_torus2int(a[0], log_2N) | 1), N);
  orc_trlwe_eval_automorphism(p, rot, acc, w0, ak_dft + ((w0 - 1) >> 1) * ak_sz, l, Bg_bit);
  for (int i = 0; i < n - 1; i++) {
    const uint64_t ai = orc_torus2int(a[i], log_2N) | 1;
    const uint64_t w1 = orc_inverse_mod_2N((uint32_t)(orc_torus2int(a[i + 1], log_2N) | 1), N);
    const uint64_t gen = (ai * w1) & mod_mask;
    orc_external_product(p, acc, rot, bk_dft + (size_t)i * trgsw_sz, 1, l, Bg_bit);
    orc_trlwe_eval_automorphism(p, rot, acc, gen, ak_dft + ((gen - 1) >> 1) * ak_sz, l, Bg_bit);
  }
  const uint64_t an = orc_torus2int(a[n - 1], log_2N) | 1;
  orc_external_product(p, rot, rot, bk_dft + (size_t)(n - 1) * trgsw_sz, 1, l, Bg_bit);
  orc_trlwe_eval_automorphism(p, acc, rot, an, ak_dft + ((an - 1) >> 1) * ak_sz, l, Bg_bit);
  free(rot);
}

/* src/bootstrap_ga.c:62-68 */
void orc_functional_bootstrap_wo_extract_ga(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                                            const double *ak_dft, int n, int l, int Bg_bit, int torus_base) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1), log_2N = log2_int(2 * N);
  const Torus prec_offset = orc_double2torus(1. / (4 * torus_base));
  const int rot = 2 * N - (int)orc_torus2int(in[n] + prec_offset, log_2N);
  for (int c = 0; c < 2; c++) orc_poly_mul_by_xai(out + (size_t)c * N, tv + (size_t)c * N, N, rot);
  orc_blind_rotate_ga(p, out, in, bk_dft, ak_dft, n, l, Bg_bit);
}

/* src/bootstrap_ga.c:70-76 */
void orc_functional_bootstrap_ga(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                                 const double *ak_dft, int n, int l, int Bg_bit, int torus_base) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *acc = (Torus *)malloc(sizeof(Torus) * (size_t)2 * N);
  orc_functional_bootstrap_wo_extract_ga(p, acc, tv, in, bk_dft, ak_dft, n, l, Bg_bit, torus_base);
  orc_trlwe_extract_tlwe(out, acc, 1, N, 0);
  free(acc);
}

/* src/keyswitch.c:12-37  KS[j] = TRLWE_out( s_in(X) * 2^(W-(j+1)bb) ), kept in the torus domain */
void orc_gen_trlwe_ks_key(orc_rng *r, Torus *ks, const Torus *s_in, const Torus *s_out, int N, int t, int base_bit, double sigma) {
  Torus *msg = (Torus *)malloc(sizeof(Torus) * (size_t)N);
  for (int j = 0; j < t; j++) {
    for (int i = 0; i < N; i++) msg[i] = s_in[i] * ((Torus)1 << (W - (j + 1) * base_bit));
    orc_trlwe_sample(r, ks + (size_t)j * 2 * N, msg, s_out, 1, N, sigma);
  }
  free(msg);
}

/* src/keyswitch.c:500-511 with skip_even: entry j switches from s(X^(2j+1)) back to s(X) */
void orc_gen_automorphism_keyset(orc_rng *r, Torus *ak, const Torus *s, int N, int t, int base_bit, double sigma) {
  Torus *s2 = (Torus *)malloc(sizeof(Torus) * (size_t)N);
  for (int j = 0; j < N; j++) {
    orc_poly_permute(s2, s, N, (uint64_t)(2 * j + 1));
    orc_gen_trlwe_ks_key(r, ak + (size_t)j * t * 2 * N, s2, s, N, t, base_bit, sigma);
  }
  free(s2);
}

/* src/bootstrap_ga.c:17-20  BK_i = TRGSW(X^{s_i}) */
void orc_gen_bootstrap_key_ga(orc_rng *r, Torus *bk, const Torus *lwe_s, int n, const Torus *rlwe_s, int N, int l, int Bg_bit,
                              double sigma) {
  const size_t sz = (size_t)2 * l * 2 * N;
  for (int i = 0; i < n; i++) orc_trgsw_monomial_sample(r, bk + (size_t)i * sz, 1, (int)lwe_s[i], rlwe_s, 1, N, l, Bg_bit, sigma);
}

/* ------------------------------------------------------------------------------------------
 * Circuit bootstrap (k = 1): LWE -> TRGSW.
 * ------------------------------------------------------------------------------------------ */
/* src/keyswitch.c:458-475  TLWE(m) -> TRLWE(m X^0) by table lookup; rows are TRLWE samples (2N words); same digit
 * rule as tlwe_keyswitch; in.b goes to coefficient 0 of the b polynomial. */
void orc_trlwe_packing1_keyswitch(Torus *out, const Torus *in, const Torus *ksk, int n, int N, int t, int base_bit) {
  const Torus round_off = (Torus)1 << (W - 1 - base_bit * t);
  const Torus mask = ((Torus)1 << base_bit) - 1;
  const size_t row = (size_t)2 * N, per_j = ((size_t)1 << base_bit) - 1;
  memset(out, 0, sizeof(Torus) * row);
  out[N] = in[n];
  for (int i = 0; i < n; i++) {
    const Torus ai = in[i] + round_off;
    for (int j = 0; j < t; j++) {
      const Torus v = (ai >> (W - (j + 1) * base_bit)) & mask;
      if (!v) continue;
      const Torus *r = ksk + (((size_t)i * t + j) * per_j + (v - 1)) * row;
      for (size_t c = 0; c < row; c++) out[c] -= r[c];
    }
  }
}

/* src/keyswitch.c:52-63  TRLWE_s(m) -> TRLWE_s(-s m) with two FFT key switches: ks1 on (-in.b, 0), ks0 on (in.a, 0) */
void orc_trlwe_priv_keyswitch_2(const orc_fft_plan *p, Torus *out, const Torus *in, const double *ks0_dft, const double *ks1_dft,
                                int t, int base_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1);
  Torus *tmp = (Torus *)calloc((size_t)2 * N, sizeof(Torus)), *tmp2 = (Torus *)calloc((size_t)2 * N, sizeof(Torus));
  for (int i = 0; i < N; i++) tmp[i] = (Torus)0 - in[N + i];
  orc_trlwe_keyswitch(p, tmp, tmp, ks1_dft, t, base_bit);
  memcpy(tmp2, in, sizeof(Torus) * (size_t)N);
  orc_trlwe_keyswitch(p, tmp2, tmp2, ks0_dft, t, base_bit);
  for (int i = 0; i < 2 * N; i++) out[i] = tmp2[i] + tmp[i];
  free(tmp);
  free(tmp2);
}

/* src/bootstrap.c:346-366  circuit_bootstrap_3: one blind rotation with the 2l-slot LUT (0,...,0, 2^(W-Bg), ..., 2^(W-l Bg)),
 * then per level: extract, packing key switch (row l+i) and private key switch of that row (row i). */
void orc_circuit_bootstrap_3(const orc_fft_plan *p, Torus *out, const Torus *in, const double *bk_dft, const double *kska0_dft,
                             const double *kska1_dft, int ta, int bba, const Torus *kskb, int tb, int bbb, int n, int l, int Bg_bit) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  const int N = 2 * (cnt + 1), slot_size = N / (2 * l);
  Torus *lut = (Torus *)calloc((size_t)2 * l + 1, sizeof(Torus));
  Torus *tv = (Torus *)malloc(sizeof(Torus) * (size_t)2 * N), *acc = (Torus *)malloc(sizeof(Torus) * (size_t)2 * N);
  Torus *ext = (Torus *)malloc(sizeof(Torus) * (size_t)(N + 1));
  for (int i = 0; i < l; i++) lut[l + i] = (Torus)1 << (W - (i + 1) * Bg_bit);
  orc_trlwe_torus_packing(tv, lut, 1, N, 2 * l);
  orc_functional_bootstrap_wo_extract(p, acc, tv, in, bk_dft, n, 1, l, Bg_bit, 2 * l);
  for (int i = 0; i < l; i++) {
    orc_trlwe_extract_tlwe(ext, acc, 1, N, i * slot_size);
    orc_trlwe_packing1_keyswitch(out + (size_t)(l + i) * 2 * N, ext, kskb, N, N, tb, bbb);
    orc_trlwe_priv_keyswitch_2(p, out + (size_t)i * 2 * N, out + (size_t)(l + i) * 2 * N, kska0_dft, kska1_dft, ta, bba);
  }
  free(lut);
  free(tv);
  free(acc);
  free(ext);
}

/* src/keyswitch.c:368-390  KS[i][j][v-1] = TRLWE_out(0) with s_in[i] v 2^(W-(j+1)bb) added to coefficient 0 of b */
void orc_gen_packing1_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in, int n, const Torus *s_out, int N, int t, int base_bit, double sigma) {
  const int base = 1 << base_bit;
  const size_t row = (size_t)2 * N;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < t; j++)
      for (int v = 1; v < base; v++) {
        Torus *dst = ksk + (((size_t)i * t + j) * (base - 1) + (v - 1)) * row;
        orc_trlwe_sample(r, dst, NULL, s_out, 1, N, sigma);
        dst[N] += s_in[i] * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit));
      }
}

/* src/keyswitch.c:39-50  ks0 switches from (-s_out * s_in), ks1 from (-s_out), both to s_out */
void orc_gen_priv_ks_key(orc_rng *r, Torus *ks0, Torus *ks1, const Torus *s_out, const Torus *s_in, int N, int t, int base_bit, double sigma) {
  Torus *neg = (Torus *)malloc(sizeof(Torus) * (size_t)N), *prod = (Torus *)malloc(sizeof(Torus) * (size_t)N);
  for (int i = 0; i < N; i++) neg[i] = (Torus)0 - s_out[i];
  orc_poly_naive_mul(prod, neg, s_in, N);
  orc_gen_trlwe_ks_key(r, ks0, prod, s_out, N, t, base_bit, sigma);
  orc_gen_trlwe_ks_key(r, ks1, neg, s_out, N, t, base_bit, sigma);
  free(neg);
  free(prod);
}

/* ------------------------------------------------------------------------------------------
 * Deterministic inputs.  The reference seeds from RDRAND / urandom (src/misc.c:34-49) and is not
 * reproducible, so tests draw keys and samples from splitmix64 with the same distributions.
 * ------------------------------------------------------------------------------------------ */
uint64_t orc_rng_next(orc_rng *r) {
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* src/misc.c:87-91  Box-Muller on two uniform torus values (shifted off zero) */
double orc_rng_normal(orc_rng *r, double sigma) {
  const double u1 = ((double)(orc_rng_next(r) >> 11) + 0.5) * 0x1p-53;
  const double u2 = ((double)(orc_rng_next(r) >> 11) + 0.5) * 0x1p-53;
  return cos(2. * M_PI * u1) * sqrt(-2. * log(u2)) * sigma;
}

/* src/tlwe.c:70-82 with bound = 2: s = r & 1 */
void orc_gen_binary_key(orc_rng *r, Torus *s, int n) {
  for (int i = 0; i < n; i++) s[i] = orc_rng_next(r) & 1;
}

/* src/tlwe.c:106-115  b = <a,s> + m + e */
void orc_tlwe_sample(orc_rng *r, Torus *out, Torus m, const Torus *s, int n, double sigma) {
  Torus b = m;
  for (int i = 0; i < n; i++) {
    out[i] = orc_rng_next(r);
    b += s[i] * out[i];
  }
  out[n] = b + orc_double2torus(orc_rng_normal(r, sigma));
}

/* src/trlwe.c:296-316  b = sum_p a_p * s_p + e (+ m); exact negacyclic product */
void orc_trlwe_sample(orc_rng *r, Torus *out, const Torus *m, const Torus *s, int k, int N, double sigma) {
  Torus *b = out + (size_t)k * N;
  for (int p = 0; p < k; p++)
    for (int i = 0; i < N; i++) out[(size_t)p * N + i] = orc_rng_next(r);
  for (int i = 0; i < N; i++) b[i] = orc_double2torus(orc_rng_normal(r, sigma));
  for (int p = 0; p < k; p++) orc_poly_naive_mul_addto(b, out + (size_t)p * N, s + (size_t)p * N, N);
  if (m)
    for (int i = 0; i < N; i++) b[i] += m[i];
}

/* src/trgsw.c:152-168  TRGSW(m X^e): (k+1)l fresh TRLWE(0) rows, row p*l+j gets m*2^(W-(j+1)Bg) on
 * coefficient e of component p */
void orc_trgsw_monomial_sample(orc_rng *r, Torus *out, int64_t m, int e, const Torus *s,
                               int k, int N, int l, int Bg_bit, double sigma) {
  if (e & N) m = -m;
  e &= N - 1;
  const size_t row = (size_t)(k + 1) * N;
  for (int q = 0; q < (k + 1) * l; q++) orc_trlwe_sample(r, out + q * row, NULL, s, k, N, sigma);
  for (int j = 0; j < l; j++) {
    const Torus h = (Torus)1 << (W - (j + 1) * Bg_bit);
    for (int p = 0; p <= k; p++) out[(size_t)(p * l + j) * row + (size_t)p * N + e] += (Torus)m * h;
  }
}

/* src/bootstrap.c:14-18  BK_i = TRGSW(s_i) (message on X^0), kept in the torus domain here */
void orc_gen_bootstrap_key(orc_rng *r, Torus *bk, const Torus *lwe_s, int n, const Torus *rlwe_s,
                           int k, int N, int l, int Bg_bit, double sigma) {
  const size_t sz = (size_t)(k + 1) * l * (k + 1) * N;
  for (int i = 0; i < n; i++)
    orc_trgsw_monomial_sample(r, bk + (size_t)i * sz, (int64_t)lwe_s[i], 0, rlwe_s, k, N, l, Bg_bit, sigma);
}

/* src/tlwe.c:193-212  KS[i][j][v-1] = TLWE_out(s_in[i] * v * 2^(W-(j+1)bb)) */
void orc_gen_tlwe_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in, int n_in, const Torus *s_out, int n_out,
                         int t, int base_bit, double sigma) {
  const int base = 1 << base_bit;
  const size_t row = (size_t)n_out + 1;
  for (int i = 0; i < n_in; i++)
    for (int j = 0; j < t; j++)
      for (int v = 1; v < base; v++) {
        const Torus msg = s_in[i] * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit));
        orc_tlwe_sample(r, ksk + (((size_t)i * t + j) * (base - 1) + (v - 1)) * row, msg, s_out, n_out, sigma);
      }
}
