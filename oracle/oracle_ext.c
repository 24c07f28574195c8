/*
 * oracle_ext.c -- the callers either side of the bootstrap:
 * public_mux and the KS21 / CLOT21 full-domain bootstraps, multi-value phase 1/2, the table-lookup private key
 * switch and circuit_bootstrap / _2, the TRGSW-accumulator bootstrap, the FFT tensor product and tlwe_mul.
 * TEST INFRASTRUCTURE ONLY (see mosfhet_oracle.h).  k = 1 throughout.
 */
#include "mosfhet_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define W 64

static int plan_N(const orc_fft_plan *p) {
  int cnt = 0;
  (void)orc_fft_twiddles(p, &cnt);
  return 2 * (cnt + 1);
}

static Torus *talloc(size_t words) { return (Torus *)calloc(words ? words : 1, sizeof(Torus)); }

/* src/bootstrap.c:369-389  public_mux: out = (0, p0) + sum_i selector[i] (.) DFT(dec_i(p1 - p0)), UN-rounded decomposition
 * (polynomial_decompose, src/polynomial.c:55-72); accumulation order i = 0 .. l-1, a then b. */
void orc_public_mux(const orc_fft_plan *p, Torus *out, const Torus *p0, const Torus *p1, const double *sel_dft, int l, int Bg_bit) {
  const int N = plan_N(p);
  Torus *d = talloc(N), *dec = talloc((size_t)l * N);
  double *dd = (double *)malloc(sizeof(double) * (size_t)N), *acc = (double *)calloc((size_t)2 * N, sizeof(double));
  for (int c = 0; c < N; c++) d[c] = p1[c] - p0[c];
  orc_poly_decompose(dec, d, N, Bg_bit, l);
  for (int i = 0; i < l; i++) {
    orc_torus_to_dft(p, dd, dec + (size_t)i * N);
    for (int c = 0; c < 2; c++) orc_dft_mul_addto(acc + (size_t)c * N, sel_dft + ((size_t)i * 2 + c) * N, dd, N);
  }
  for (int c = 0; c < 2; c++) orc_dft_to_torus(p, out + (size_t)c * N, acc + (size_t)c * N);
  for (int c = 0; c < N; c++) out[N + c] += p0[c];
  free(d); free(dec); free(dd); free(acc);
}

/* src/bootstrap.c:391-432 (variant 0, _KS21) and :434-463 (variant 1, _KS21_2).  tv has 2N coefficients. */
void orc_full_domain_functional_bootstrap_KS21(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                                               const Torus *ksk, int n, int l, int Bg_bit, int t, int base_bit, int torus_base, int variant) {
  const int N = plan_N(p);
  Torus *tmp_trlwe = talloc((size_t)2 * N), *tmp_trlwe2 = talloc((size_t)2 * N), *tmp = talloc((size_t)N + 1);
  double *sign_dec = (double *)malloc(sizeof(double) * (size_t)l * 2 * N);
  if (variant == 0) {
    const int slot_size = N / (l * torus_base / 2);
    Torus *lut = talloc((size_t)l * torus_base / 2);
    for (int i = 0; i < l; i++)
      for (int j = 0; j < torus_base / 2; j++) lut[i * torus_base / 2 + j] = (Torus)-1 << (W - (i + 1) * Bg_bit - 1);
    orc_trlwe_torus_packing_many_LUT(tmp_trlwe, lut, 1, N, torus_base / 2, l);
    orc_functional_bootstrap_wo_extract(p, tmp_trlwe2, tmp_trlwe, in, bk_dft, n, 1, l, Bg_bit, l * torus_base / 2);
    for (int i = 0; i < l; i++) {
      const Torus sign = (Torus)-1 << (W - (i + 1) * Bg_bit - 1);
      orc_trlwe_extract_tlwe(tmp, tmp_trlwe2, 1, N, i * slot_size);
      tmp[N] -= sign;
      orc_trlwe_packing1_keyswitch(tmp_trlwe, tmp, ksk, N, N, t, base_bit);
      for (int c = 0; c < 2; c++) orc_torus_to_dft(p, sign_dec + ((size_t)i * 2 + c) * N, tmp_trlwe + (size_t)c * N);
    }
    free(lut);
  } else {
    for (int i = 0; i < l; i++) {
      const Torus sign = (Torus)-1 << (W - (i + 1) * Bg_bit - 1);
      orc_trlwe_torus_packing(tmp_trlwe, &sign, 1, N, 1);
      orc_functional_bootstrap(p, tmp, tmp_trlwe, in, bk_dft, n, 1, l, Bg_bit, torus_base / 2);
      tmp[N] -= sign;
      orc_trlwe_packing1_keyswitch(tmp_trlwe, tmp, ksk, N, N, t, base_bit);
      for (int c = 0; c < 2; c++) orc_torus_to_dft(p, sign_dec + ((size_t)i * 2 + c) * N, tmp_trlwe + (size_t)c * N);
    }
  }
  Torus *p0 = talloc(N), *p1 = talloc(N);
  for (int i = 0; i < N; i++) {
    p0[i] = tv[i];
    p1[i] = (Torus)0 - tv[i + N];
  }
  orc_public_mux(p, tmp_trlwe, p0, p1, sign_dec, l, Bg_bit);
  orc_functional_bootstrap(p, out, tmp_trlwe, in, bk_dft, n, 1, l, Bg_bit, torus_base / 2);
  free(tmp_trlwe); free(tmp_trlwe2); free(tmp); free(sign_dec); free(p0); free(p1);
}

/* src/bootstrap.c:232-243  multivalue_bootstrap_phase1: out[0] = BR of the constant vector 1/(4 torus_base);
 * out[i] = out[0] X^(i N / torus_base); out[torus_base] = out[0] X^torus_base + out[0]. */
void orc_multivalue_bootstrap_phase1(const orc_fft_plan *p, Torus *out, const Torus *in, const double *bk_dft, int n, int l, int Bg_bit,
                                     int torus_base) {
  const int N = plan_N(p);
  const size_t sz = (size_t)2 * N;
  Torus *tv = talloc(sz);
  for (int i = 0; i < N; i++) tv[N + i] = orc_double2torus(1. / (4 * torus_base));
  orc_functional_bootstrap_wo_extract(p, out, tv, in, bk_dft, n, 1, l, Bg_bit, torus_base);
  for (int i = 1; i < torus_base; i++)
    for (int c = 0; c < 2; c++) orc_poly_mul_by_xai(out + i * sz + (size_t)c * N, out + (size_t)c * N, N, i * N / torus_base);
  for (int c = 0; c < 2; c++) {
    Torus *dst = out + torus_base * sz + (size_t)c * N;
    orc_poly_mul_by_xai(dst, out + (size_t)c * N, N, torus_base);
    for (int i = 0; i < N; i++) dst[i] += out[(size_t)c * N + i];
  }
  free(tv);
}

/* src/trlwe.c:554-578  extract with += / -= */
static void extract_acc(Torus *out, const Torus *in, int N, int idx, int sign) {
  for (int j = 0; j <= idx; j++) out[j] += sign > 0 ? in[idx - j] : (Torus)0 - in[idx - j];
  for (int j = idx + 1; j < N; j++) out[j] += sign > 0 ? (Torus)0 - in[N + idx - j] : in[N + idx - j];
  out[N] += sign > 0 ? in[N + idx] : (Torus)0 - in[N + idx];
}

/* src/trlwe.c:603-611  trlwe_mv_extract_tlwe_scaling_addto */
void orc_trlwe_mv_extract_tlwe_scaling_addto(Torus *out, const Torus *in, int N, int scale) {
  const int amount = scale;
  for (int i = amount / 2; i < amount; i++) extract_acc(out, in, N, N - 1 - (i - amount / 2), -1);
  for (int i = 0; i < amount / 2; i++) extract_acc(out, in, N, i, +1);
}

/* src/bootstrap.c:245-265  multivalue_bootstrap_phase2: cleartext LUT `lut_in` (torus_base small integers) against the rotated
 * accumulators of phase 1; bit j of the LUT selects signed sums of them, extracted with weight 2^j. */
void orc_multivalue_bootstrap_phase2(Torus *out, const int *lut_in, const Torus *rotated_tv, int N, int torus_base, int log_torus_base) {
  const size_t sz = (size_t)2 * N;
  Torus *tmp = talloc(sz);
  memset(out, 0, sizeof(Torus) * ((size_t)N + 1));
  for (int j = 0; j < log_torus_base; j++) {
    const int in_over_tv_0 = ((lut_in[0] >> j) & 1) + ((lut_in[torus_base - 1] >> j) & 1);
    if (in_over_tv_0 == 2) memcpy(tmp, rotated_tv + torus_base * sz, sizeof(Torus) * sz);
    else if (in_over_tv_0 == 1) memcpy(tmp, rotated_tv, sizeof(Torus) * sz);
    else memset(tmp, 0, sizeof(Torus) * sz);
    for (int i = 1; i < torus_base; i++) {
      const int d = ((lut_in[i] >> j) & 1) - ((lut_in[i - 1] >> j) & 1);
      if (d == 1) for (size_t c = 0; c < sz; c++) tmp[c] += rotated_tv[i * sz + c];
      else if (d == -1) for (size_t c = 0; c < sz; c++) tmp[c] -= rotated_tv[i * sz + c];
    }
    orc_trlwe_mv_extract_tlwe_scaling_addto(out, tmp, N, 1 << j);
  }
  free(tmp);
}

/* src/keyswitch.c:611-637  trlwe_new_priv_SK_KS_key_N2: entries i <= n (i = n stands for the b word, key -1),
 * s[i][j][v-1] = TRLWE( -s_out * s_i v 2^(64-(j+1)bb) ), rows uncompressed. */
void orc_gen_priv_sk_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in, int n, const Torus *s_out, int N, int t, int base_bit, double sigma) {
  const int base = 1 << base_bit;
  const size_t row = (size_t)2 * N;
  for (int i = 0; i <= n; i++) {
    const Torus s_i = i < n ? s_in[i] : (Torus)-1;
    for (int j = 0; j < t; j++)
      for (int v = 1; v < base; v++) {
        Torus *dst = ksk + (((size_t)i * t + j) * (base - 1) + (v - 1)) * row;
        const Torus dec_key = s_i * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit));
        orc_trlwe_sample(r, dst, NULL, s_out, 1, N, sigma);
        for (int e = 0; e < N; e++) dst[N + e] += ((Torus)0 - s_out[e]) * dec_key;
      }
  }
}

/* src/keyswitch.c:639-656  trlwe_priv_keyswitch: LWE(m) under s_in -> TRLWE(-s_out m); the b word is digit-decomposed too. */
void orc_trlwe_priv_keyswitch(Torus *out, const Torus *in, const Torus *ksk, int n, int N, int t, int base_bit) {
  const Torus prec_offset = (Torus)1 << (W - (1 + base_bit * t));
  const Torus mask = ((Torus)1 << base_bit) - 1;
  const int base = 1 << base_bit;
  const size_t row = (size_t)2 * N;
  memset(out, 0, sizeof(Torus) * row);
  for (int i = 0; i <= n; i++) {
    const Torus aibar = in[i] + prec_offset;   /* in[n] is the b word */
    for (int j = 0; j < t; j++) {
      const Torus aij = (aibar >> (W - (j + 1) * base_bit)) & mask;
      if (aij != 0) {
        const Torus *src = ksk + (((size_t)i * t + j) * (base - 1) + (aij - 1)) * row;
        for (size_t c = 0; c < row; c++) out[c] -= src[c];
      }
    }
  }
}

/* src/keyswitch.c:346-366  trlwe_packing_keyswitch: torus_base LWE samples (in: [torus_base][n + 1]) into the torus_base slots of one TRLWE sample.
 * out = (0, b_e on the N / torus_base coefficients of slot e), then for every mask word i, sample e and digit position j the row KS[i][e][j][digit - 1] is
 * subtracted; ksk rows in the order (i, e, j, v), 2N words each; rounding offset 2^(W - 1 - t base_bit). */
void orc_trlwe_lut_packing_keyswitch(Torus *out, const Torus *in, const Torus *ksk, int n, int N, int t, int base_bit, int torus_base) {
  const Torus prec_offset = (Torus)1 << (W - (1 + base_bit * t));
  const Torus mask = ((Torus)1 << base_bit) - 1;
  const int base = 1 << base_bit, span = N / torus_base;
  const size_t row = (size_t)2 * N;
  for (int q = 0; q < N; q++) {
    out[q] = 0;
    out[N + q] = in[(size_t)(q / span) * (n + 1) + n];
  }
  for (int i = 0; i < n; i++)
    for (int e = 0; e < torus_base; e++) {
      const Torus aibar = in[(size_t)e * (n + 1) + i] + prec_offset;
      for (int j = 0; j < t; j++) {
        const Torus aij = (aibar >> (W - (j + 1) * base_bit)) & mask;
        if (aij != 0) {
          const Torus *src = ksk + ((((size_t)i * torus_base + e) * t + j) * (base - 1) + (aij - 1)) * row;
          for (size_t c = 0; c < row; c++) out[c] -= src[c];
        }
      }
    }
}

/* src/keyswitch.c:214-241  trlwe_new_packing_KS_key: KS[i][e][j][v-1] = TRLWE(0) + s_i v 2^(W - (j+1) base_bit) on the b coefficients of slot e */
void orc_gen_lut_packing_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in, int n, const Torus *s_out, int N, int t, int base_bit, int torus_base, double sigma) {
  const int base = 1 << base_bit, span = N / torus_base;
  const size_t row = (size_t)2 * N;
  for (int i = 0; i < n; i++)
    for (int e = 0; e < torus_base; e++)
      for (int j = 0; j < t; j++)
        for (int v = 1; v < base; v++) {
          Torus *dst = ksk + ((((size_t)i * torus_base + e) * t + j) * (base - 1) + (v - 1)) * row;
          orc_trlwe_sample(r, dst, NULL, s_out, 1, N, sigma);
          const Torus dec = s_in[i] * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit));
          for (int q = e * span; q < (e + 1) * span; q++) dst[N + q] += dec;
        }
}

/* src/bootstrap.c:309-322 (variant 0, circuit_bootstrap) and :324-344 (variant 1, circuit_bootstrap_2) */
void orc_circuit_bootstrap(const orc_fft_plan *p, Torus *out, const Torus *in, const double *bk_dft, const Torus *kska, int ta, int bba,
                           const Torus *kskb, int tb, int bbb, int n, int l, int Bg_bit, int variant) {
  const int N = plan_N(p);
  const size_t row = (size_t)2 * N;
  Torus *tv = talloc(row), *tmp = talloc(row), *tmp_out = talloc((size_t)N + 1);
  if (variant == 0) {
    for (int i = 0; i < l; i++) {
      const Torus lut[2] = {0, (Torus)1 << (W - (i + 1) * Bg_bit)};
      orc_trlwe_torus_packing(tv, lut, 1, N, 2);
      orc_functional_bootstrap(p, tmp_out, tv, in, bk_dft, n, 1, l, Bg_bit, 2);
      orc_trlwe_priv_keyswitch(out + (size_t)i * row, tmp_out, kska, N, N, ta, bba);
      orc_trlwe_packing1_keyswitch(out + (size_t)(l + i) * row, tmp_out, kskb, N, N, tb, bbb);
    }
  } else {
    const int slot_size = N / (2 * l);
    Torus *lut = talloc((size_t)2 * l + 1);   /* + 1: trlwe_torus_packing indexes lut[2l] for the last coefficients when 2l does not divide N */
    for (int i = 0; i < l; i++) lut[l + i] = (Torus)1 << (W - (i + 1) * Bg_bit);
    orc_trlwe_torus_packing(tv, lut, 1, N, 2 * l);
    orc_functional_bootstrap_wo_extract(p, tmp, tv, in, bk_dft, n, 1, l, Bg_bit, 2 * l);
    for (int i = 0; i < l; i++) {
      orc_trlwe_extract_tlwe(tmp_out, tmp, 1, N, i * slot_size);
      orc_trlwe_priv_keyswitch(out + (size_t)i * row, tmp_out, kska, N, N, ta, bba);
      orc_trlwe_packing1_keyswitch(out + (size_t)(l + i) * row, tmp_out, kskb, N, N, tb, bbb);
    }
    free(lut);
  }
  free(tv); free(tmp); free(tmp_out);
}

/* src/bootstrap.c:267-295  functional_bootstrap_trgsw_phase1: blind rotation with a TRGSW accumulator.  trgsw_mul_DFT is the
 * row-wise external product (src/trgsw.c:425-431), so every one of the 2l rows of the trivial TRGSW(1) is rotated independently. */
void orc_functional_bootstrap_trgsw_phase1(const orc_fft_plan *p, double *out_dft, const Torus *in, const double *bk_dft, int n, int l,
                                           int Bg_bit, int torus_base) {
  const int N = plan_N(p);
  int log_N2 = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  const size_t row = (size_t)2 * N;
  Torus *tv = talloc(row), *acc = talloc((size_t)2 * l * row);
  const int rot = 2 * N - (int)orc_torus2int(in[n] + orc_double2torus(1. / (4 * torus_base)), log_N2);
  for (int q = 0; q < 2 * l; q++) {
    memset(tv, 0, sizeof(Torus) * row);
    tv[(size_t)(q / l) * N] = (Torus)1 << (W - (q % l + 1) * Bg_bit);   /* rows < l: gadget on a; rows >= l: on b (src/trgsw.c:130-142) */
    for (int c = 0; c < 2; c++) orc_poly_mul_by_xai(acc + q * row + (size_t)c * N, tv + (size_t)c * N, N, rot);
    orc_blind_rotate(p, acc + q * row, in, bk_dft, n, 1, l, Bg_bit);
  }
  orc_trgsw_to_dft(p, out_dft, acc, 1, l);
  free(tv); free(acc);
}

/* src/bootstrap.c:297-306  phase 2: tv (.) TRGSW_DFT(X^-phase), sample extract at 0 */
void orc_functional_bootstrap_trgsw_phase2(const orc_fft_plan *p, Torus *out, const double *in_dft, const Torus *tv, int l, int Bg_bit) {
  const int N = plan_N(p);
  Torus *tmp = talloc((size_t)2 * N);
  orc_external_product(p, tmp, tv, in_dft, 1, l, Bg_bit);
  orc_trlwe_extract_tlwe(out, tmp, 1, N, 0);
  free(tmp);
}

/* src/keyswitch.c:3-10  trlwe_new_RL_key: switches from s^2 to s */
void orc_gen_rl_key(orc_rng *r, Torus *ks, const Torus *s, int N, int t, int base_bit, double sigma) {
  Torus *s2 = talloc(N);
  orc_poly_naive_mul(s2, s, s, N);
  orc_gen_trlwe_ks_key(r, ks, s2, s, N, t, base_bit, sigma);
  free(s2);
}

/* src/trlwe.c:727-771  trlwe_tensor_prod_FFT: operands are rescaled to (64 - precision)/2-ish bits (polynomial_torus_scale =
 * torus2int, src/polynomial.c:322-326), multiplied in the DFT domain, the s^2 term is relinearised with rl_key. */
void orc_trlwe_tensor_prod_fft(const orc_fft_plan *p, Torus *out, const Torus *in1, const Torus *in2, int precision, const double *rl_dft,
                               int t, int base_bit) {
  const int N = plan_N(p);
  const int half_prec1 = W - (W - precision) / 2, half_prec2 = W - (W - precision + 1) / 2;
  Torus *tmp = talloc(N), *t2 = talloc((size_t)2 * N), *res = talloc((size_t)2 * N);
  double *A1 = (double *)malloc(sizeof(double) * (size_t)6 * N), *A2 = A1 + N, *B1 = A2 + N, *B2 = B1 + N, *tmp_dft = B2 + N, *ta = tmp_dft + N;
  for (int i = 0; i < N; i++) tmp[i] = orc_torus2int(in1[i], half_prec1);
  orc_torus_to_dft(p, A1, tmp);
  for (int i = 0; i < N; i++) tmp[i] = orc_torus2int(in2[i], half_prec2);
  orc_torus_to_dft(p, A2, tmp);
  orc_dft_mul(ta, A1, A2, N);
  for (int i = 0; i < N; i++) tmp[i] = orc_torus2int(in1[N + i], half_prec1);
  orc_torus_to_dft(p, B1, tmp);
  for (int i = 0; i < N; i++) tmp[i] = orc_torus2int(in2[N + i], half_prec2);
  orc_torus_to_dft(p, B2, tmp);
  orc_dft_mul(tmp_dft, A1, B2, N);
  orc_dft_mul_addto(tmp_dft, B1, A2, N);
  orc_dft_to_torus(p, res, tmp_dft);
  orc_dft_mul(tmp_dft, B1, B2, N);
  orc_dft_to_torus(p, res + N, tmp_dft);
  orc_dft_to_torus(p, t2, ta);           /* t = (A1 A2, 0) */
  memset(t2 + N, 0, sizeof(Torus) * (size_t)N);
  orc_trlwe_keyswitch(p, t2, t2, rl_dft, t, base_bit);
  for (int i = 0; i < 2 * N; i++) out[i] = res[i] - t2[i];
  free(tmp); free(t2); free(res); free(A1);
}

/* src/tlwe.c:322-332  tlwe_mul: pack both operands, tensor product, extract coefficient 0 */
void orc_tlwe_mul(const orc_fft_plan *p, Torus *out, const Torus *in1, const Torus *in2, int precision, const Torus *ksk, int tk, int bbk,
                  const double *rl_dft, int tr, int bbr) {
  const int N = plan_N(p);
  Torus *tmp1 = talloc((size_t)2 * N), *tmp2 = talloc((size_t)2 * N);
  orc_trlwe_packing1_keyswitch(tmp1, in1, ksk, N, N, tk, bbk);
  orc_trlwe_packing1_keyswitch(tmp2, in2, ksk, N, N, tk, bbk);
  orc_trlwe_tensor_prod_fft(p, tmp1, tmp1, tmp2, precision, rl_dft, tr, bbr);
  orc_trlwe_extract_tlwe(out, tmp1, 1, N, 0);
  free(tmp1); free(tmp2);
}

/* src/bootstrap.c:465-491 (variant 0, _CLOT21: tv = two test vectors [2][2][N]) and :493-517 (variant 1, _CLOT21_2: tv =
 * 2 torus_base cleartext LUT words, torus_base = 2^(precision-2)) */
void orc_full_domain_functional_bootstrap_CLOT21(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                                                 const Torus *ksk, int tk, int bbk, const double *rl_dft, int tr, int bbr, int n, int l,
                                                 int Bg_bit, int precision, int variant) {
  const int N = plan_N(p);
  const size_t row = (size_t)2 * N;
  Torus *tmp_trlwe = talloc(row), *tmp_trlwe2 = talloc(row);
  Torus *ct_sign = talloc((size_t)N + 1), *ct_f0 = talloc((size_t)N + 1), *ct_f1 = talloc((size_t)N + 1);
  const Torus sign = (Torus)1 << (W - precision - 1);
  if (variant == 0) {
    orc_trlwe_torus_packing(tmp_trlwe, &sign, 1, N, 1);
    orc_functional_bootstrap(p, ct_f0, tv, in, bk_dft, n, 1, l, Bg_bit, 1 << (precision - 1));
    orc_functional_bootstrap(p, ct_f1, tv + row, in, bk_dft, n, 1, l, Bg_bit, 1 << (precision - 1));
    orc_functional_bootstrap(p, ct_sign, tmp_trlwe, in, bk_dft, n, 1, l, Bg_bit, 1 << (precision - 1));
  } else {
    const int torus_base = 1 << (precision - 2), slot_size = N / (4 * torus_base);
    Torus *lut = talloc((size_t)4 * torus_base);
    memcpy(lut, tv, sizeof(Torus) * 2 * (size_t)torus_base);
    for (int i = 2 * torus_base; i < 3 * torus_base; i++) lut[i] = sign;
    orc_trlwe_torus_packing_many_LUT(tmp_trlwe, lut, 1, N, torus_base, 4);
    orc_functional_bootstrap_wo_extract(p, tmp_trlwe2, tmp_trlwe, in, bk_dft, n, 1, l, Bg_bit, 4 * torus_base);
    orc_trlwe_extract_tlwe(ct_f0, tmp_trlwe2, 1, N, 0);
    orc_trlwe_extract_tlwe(ct_f1, tmp_trlwe2, 1, N, slot_size);
    orc_trlwe_extract_tlwe(ct_sign, tmp_trlwe2, 1, N, 2 * slot_size);
    free(lut);
  }
  ct_sign[N] -= sign;
  orc_tlwe_mul(p, ct_f1, ct_f1, ct_sign, precision, ksk, tk, bbk, rl_dft, tr, bbr);
  ct_sign[N] += 2 * sign;
  orc_tlwe_mul(p, ct_f0, ct_f0, ct_sign, precision, ksk, tk, bbk, rl_dft, tr, bbr);
  for (int i = 0; i <= N; i++) out[i] = ct_f0[i] + ct_f1[i];
  free(tmp_trlwe); free(tmp_trlwe2); free(ct_sign); free(ct_f0); free(ct_f1);
}

/* src/bootstrap.c:23-48  new_bootstrap_key with unfolding u > 1: for every group of u key bits, 2^u torus-domain TRGSW samples
 * su[i 2^u/u + j] = TRGSW( prod_{b<u} (bit b of j ? s_{i+b} : 1 - s_{i+b}) ), i.e. the indicator of the group's bit pattern. */
void orc_gen_bootstrap_key_unfolded(orc_rng *r, Torus *su, const Torus *lwe_s, int n, const Torus *rlwe_s, int N, int l, int Bg_bit, double sigma,
                                    int unfolding) {
  const int key_exp = 1 << unfolding, final_exp = key_exp / unfolding;
  const size_t sz = (size_t)2 * l * 2 * N;
  for (int i = 0; i < n; i += unfolding)
    for (int j = 0; j < key_exp; j++) {
      Torus key = 1;
      for (int u = 0, j_ = j; u < unfolding; u++, j_ >>= 1) key *= (j_ & 1) ? lwe_s[i + u] : 1 - lwe_s[i + u];
      orc_trgsw_monomial_sample(r, su + ((size_t)i * final_exp + j) * sz, (int64_t)key, 0, rlwe_s, 1, N, l, Bg_bit, sigma);
    }
}

/* src/bootstrap.c:124-149  blind_rotate_unfolded: per group, xai = sum_j X^(modswitch(sum of the group's mask words selected by j)) su_j
 * assembled in the torus domain, transformed (trgsw_to_DFT), and acc <- xai (.) acc (the product REPLACES the accumulator). */
void orc_blind_rotate_unfolded(const orc_fft_plan *p, Torus *acc, const Torus *a, const Torus *su, int n, int l, int Bg_bit, int unfolding) {
  const int N = plan_N(p);
  int log_N2 = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  const int key_exp = 1 << unfolding, final_exp = key_exp / unfolding;
  const size_t sz = (size_t)2 * l * 2 * N;
  Torus *xai = talloc(sz), *out = talloc((size_t)2 * N);
  double *xai_dft = (double *)malloc(sizeof(double) * sz);
  for (int i = 0; i < n; i += unfolding) {
    memcpy(xai, su + (size_t)i * final_exp * sz, sizeof(Torus) * sz);
    for (int j = 1; j < key_exp; j++) {
      Torus a_i = 0;
      for (int u = 0, j_ = j; u < unfolding; u++, j_ >>= 1)
        if (j_ & 1) a_i += a[i + u];
      const int rot = (int)orc_torus2int(a_i, log_N2);
      const Torus *src = su + ((size_t)i * final_exp + j) * sz;
      for (int q = 0; q < 2 * l * 2; q++) orc_poly_mul_by_xai_addto(xai + (size_t)q * N, src + (size_t)q * N, N, rot);
    }
    orc_trgsw_to_dft(p, xai_dft, xai, 1, l);
    orc_external_product(p, out, acc, xai_dft, 1, l, Bg_bit);
    memcpy(acc, out, sizeof(Torus) * (size_t)2 * N);
  }
  free(xai); free(out); free(xai_dft);
}

/* ------------------------------------------------------------------------------------------------------------------------------------------
 * Unfolding 2 with the per-group TRGSW assembled in the DFT DOMAIN (the order the GPU's unfolding-2 kernels compute in; mosfhet_amd/csrc/
 * unfold_kernels.h).  The reference assembles  xai = su_0 + sum_{j=1..3} X^(e_j) su_j  in the torus domain and transforms it
 * (src/bootstrap.c:134-143); multiplication by X^e is pointwise multiplication by y^e at every root y the transform evaluates at, so with the
 * samples transformed once (su_dft) the same TRGSW_DFT is  S = su_dft_0 + sum_j y^(e_j) (.) su_dft_j  up to floating-point rounding: no transform
 * of key material per step at all.  Slot sigma (natural order) holds the value at y = psi^(4 bitrev(sigma) + 1), psi = exp(i pi / N)
 * (oracle_fft.c); with sigma = 8 t + m:
 *     4 bitrev(sigma) + 1 = (4 bitrev(t) + 1) + bitrev3(m) N / 4,   so   y^e = W[(4 bitrev(t) + 1) e mod 2N] * W[(bitrev3(m) e mod 8) N / 4],
 * W[x] = exp(i pi x / N) -- one table lookup per lane and a wave-uniform eighth root per register on the GPU.
 * FIXED ORDER (mirrored by the kernels): per slot and j = 1, 2, 3 in order
 *     c = (m == 0) ? base : base * kappa      base = W[(4 bitrev(t) + 1) e_j mod 2N], kappa = W[(bitrev3(m) e_j mod 8) N / 4];
 *                                             cr = fma(-bi, ki, br * kr),  ci = fma(bi, kr, br * ki)
 *     S = K_0;  S.re = fma(-ci, Kj.im, fma(cr, Kj.re, S.re)),  S.im = fma(ci, Kj.re, fma(cr, Kj.im, S.im))
 * then the external product of src/trgsw.c:270-286 with S exactly as orc_external_product does it; the product REPLACES the accumulator. */
void orc_monomial_table(double *out /*[2N][2]*/, int N) {
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int x = 0; x < 2 * N; x++) {
    out[2 * x] = (double)cosl(pi * (long double)x / (long double)N);
    out[2 * x + 1] = (double)sinl(pi * (long double)x / (long double)N);
  }
}

static unsigned bitrev_u(unsigned x, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

/* the selector of one group: S[2l][2][N] from the group's four transformed samples K[4][2l][2][N] and the two mask words */
void orc_unfold2_selector_dft(double *S, const double *K, Torus a0, Torus a1, int N, int l) {
  const int M = N / 2;
  int log_N2 = 0, logM = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  while ((1 << logM) < M) logM++;
  double *Wt = (double *)malloc(sizeof(double) * 4 * (size_t)N);
  orc_monomial_table(Wt, N);
  const size_t sz = (size_t)2 * l * 2 * N;
  const unsigned e[4] = {0, (unsigned)orc_torus2int(a0, log_N2), (unsigned)orc_torus2int(a1, log_N2), (unsigned)orc_torus2int(a0 + a1, log_N2)};
  for (int rc = 0; rc < 2 * l * 2; rc++)
    for (int sg = 0; sg < M; sg++) {
      const int t = sg >> 3, m = sg & 7;
      const unsigned g0 = 4u * bitrev_u((unsigned)t, logM - 3) + 1u;
      const size_t at = (size_t)rc * N + 2 * (size_t)sg;
      double sr = K[at], si = K[at + 1];
      for (int j = 1; j < 4; j++) {
        const unsigned xb = (g0 * e[j]) & (unsigned)(2 * N - 1);
        double cr = Wt[2 * xb], ci = Wt[2 * xb + 1];
        if (m) {
          const unsigned q8 = (bitrev_u((unsigned)m, 3) * e[j]) & 7u;
          const double kr = Wt[2 * (size_t)(q8 * (unsigned)(N / 4))], ki = Wt[2 * (size_t)(q8 * (unsigned)(N / 4)) + 1];
          const double xr = fma(-ci, ki, cr * kr), xi = fma(ci, kr, cr * ki);
          cr = xr; ci = xi;
        }
        const double Kr = K[(size_t)j * sz + at], Ki = K[(size_t)j * sz + at + 1];
        sr = fma(-ci, Ki, fma(cr, Kr, sr));
        si = fma(ci, Kr, fma(cr, Ki, si));
      }
      S[at] = sr;
      S[at + 1] = si;
    }
  free(Wt);
}

void orc_blind_rotate_unfolded2_dft(const orc_fft_plan *p, Torus *acc, const Torus *a, const double *su_dft /*[n/2 * 4][2l][2][N]*/, int n, int l, int Bg_bit) {
  const int N = plan_N(p);
  const size_t sz = (size_t)2 * l * 2 * N;
  double *S = (double *)malloc(sizeof(double) * sz);
  Torus *out = talloc((size_t)2 * N);
  for (int i = 0; i < n; i += 2) {
    orc_unfold2_selector_dft(S, su_dft + (size_t)(i / 2) * 4 * sz, a[i], a[i + 1], N, l);
    orc_external_product(p, out, acc, S, 1, l, Bg_bit);
    memcpy(acc, out, sizeof(Torus) * (size_t)2 * N);
  }
  free(S); free(out);
}

/* functional_bootstrap(_wo_extract) with an unfolding-2 key in that order */
void orc_functional_bootstrap_unfolded2_dft(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *su_dft, int n, int l, int Bg_bit,
                                            int torus_base, int extract) {
  const int N = plan_N(p);
  int log_N2 = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  Torus *acc = talloc((size_t)2 * N);
  const int rot = 2 * N - (int)orc_torus2int(in[n] + orc_double2torus(1. / (4 * torus_base)), log_N2);
  for (int c = 0; c < 2; c++) orc_poly_mul_by_xai(acc + (size_t)c * N, tv + (size_t)c * N, N, rot);
  orc_blind_rotate_unfolded2_dft(p, acc, in, su_dft, n, l, Bg_bit);
  if (extract) orc_trlwe_extract_tlwe(out, acc, 1, N, 0);
  else memcpy(out, acc, sizeof(Torus) * (size_t)2 * N);
  free(acc);
}

/* src/bootstrap.c:192-206 with key->unfolding > 1 */
void orc_functional_bootstrap_unfolded(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const Torus *su, int n, int l, int Bg_bit,
                                       int torus_base, int unfolding, int extract) {
  const int N = plan_N(p);
  int log_N2 = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  Torus *acc = talloc((size_t)2 * N);
  const int rot = 2 * N - (int)orc_torus2int(in[n] + orc_double2torus(1. / (4 * torus_base)), log_N2);
  for (int c = 0; c < 2; c++) orc_poly_mul_by_xai(acc + (size_t)c * N, tv + (size_t)c * N, N, rot);
  orc_blind_rotate_unfolded(p, acc, in, su, n, l, Bg_bit, unfolding);
  if (extract) orc_trlwe_extract_tlwe(out, acc, 1, N, 0);
  else memcpy(out, acc, sizeof(Torus) * (size_t)2 * N);
  free(acc);
}

/* src/bootstrap.c:151-175  multivalue_bootstrap_UBR_phase1: the per-group TRGSW of blind_rotate_unfolded, transformed, one per group
 * (out_dft: [n / u][2l][2][N] doubles).  Depends on the input's mask only, so one phase 1 serves any number of test vectors. */
void orc_multivalue_bootstrap_UBR_phase1(const orc_fft_plan *p, double *out_dft, const Torus *in, const Torus *su, int n, int l, int Bg_bit, int unfolding) {
  const int N = plan_N(p);
  int log_N2 = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  (void)Bg_bit;
  const int key_exp = 1 << unfolding, final_exp = key_exp / unfolding;
  const size_t sz = (size_t)2 * l * 2 * N;
  Torus *xai = talloc(sz);
  for (int i = 0, g = 0; i < n; i += unfolding, g++) {
    memcpy(xai, su + (size_t)i * final_exp * sz, sizeof(Torus) * sz);
    for (int j = 1; j < key_exp; j++) {
      Torus a_i = 0;
      for (int u = 0, j_ = j; u < unfolding; u++, j_ >>= 1)
        if (j_ & 1) a_i += in[i + u];
      const int rot = (int)orc_torus2int(a_i, log_N2);
      const Torus *src = su + ((size_t)i * final_exp + j) * sz;
      for (int q = 0; q < 2 * l * 2; q++) orc_poly_mul_by_xai_addto(xai + (size_t)q * N, src + (size_t)q * N, N, rot);
    }
    orc_trgsw_to_dft(p, out_dft + (size_t)g * sz, xai, 1, l);
  }
  free(xai);
}

/* src/bootstrap.c:177-190  multivalue_bootstrap_UBR_phase2: rotate tv by the body, chain of external products, extract */
void orc_multivalue_bootstrap_UBR_phase2(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *sa_dft, int n, int l,
                                         int Bg_bit, int unfolding, int torus_base) {
  const int N = plan_N(p);
  int log_N2 = 0;
  while ((1 << log_N2) < 2 * N) log_N2++;
  const size_t sz = (size_t)2 * l * 2 * N;
  Torus *acc = talloc((size_t)2 * N), *tmp = talloc((size_t)2 * N);
  const int rot = 2 * N - (int)orc_torus2int(in[n] + orc_double2torus(1. / (4 * torus_base)), log_N2);
  for (int c = 0; c < 2; c++) orc_poly_mul_by_xai(acc + (size_t)c * N, tv + (size_t)c * N, N, rot);
  for (int g = 0; g < n / unfolding; g++) {
    orc_external_product(p, tmp, acc, sa_dft + (size_t)g * sz, 1, l, Bg_bit);
    memcpy(acc, tmp, sizeof(Torus) * (size_t)2 * N);
  }
  orc_trlwe_extract_tlwe(out, acc, 1, N, 0);
  free(acc); free(tmp);
}

/* src/trlwe.c:580-622  the multi-value extraction helpers.  mode 0: trlwe_mv_extract_tlwe (`amount` outputs), 1: _scaling (=),
 * 2: _scaling_addto (+=), 3: _scaling_subto (-=).  `out` holds `amount` TLWEs in mode 0, one otherwise (read in modes 2, 3). */
void orc_trlwe_mv_extract(Torus *out, const Torus *in, int N, int mode, int amount) {
  const size_t w = (size_t)N + 1;
  if (mode == 0) {
    for (int i = 0; i < amount / 2; i++) orc_trlwe_extract_tlwe(out + i * w, in, 1, N, i);
    for (int i = amount / 2; i < amount; i++) {
      orc_trlwe_extract_tlwe(out + i * w, in, 1, N, N - 1 - (i - amount / 2));
      for (size_t c = 0; c < w; c++) out[i * w + c] = (Torus)0 - out[i * w + c];
    }
  } else if (mode == 1) {
    orc_trlwe_extract_tlwe(out, in, 1, N, amount / 2);
    for (int i = amount / 2 + 1; i < amount; i++) extract_acc(out, in, N, N - 1 - (i - amount / 2), -1);
    for (int i = 0; i < amount / 2; i++) extract_acc(out, in, N, i, +1);
  } else {
    const int s = mode == 2 ? +1 : -1;
    for (int i = amount / 2; i < amount; i++) extract_acc(out, in, N, N - 1 - (i - amount / 2), -s);
    for (int i = 0; i < amount / 2; i++) extract_acc(out, in, N, i, s);
  }
}
