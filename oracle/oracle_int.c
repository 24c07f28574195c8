/*
 * oracle_int.c -- exact (mod 2^64) integer pieces of the bootstrap path.
 * TEST INFRASTRUCTURE ONLY (see mosfhet_oracle.h).  Each function restates the
 * reference function cited above it; parity with the reference build in
 * oracle/_ref is bit-for-bit (tests/test_oracle_vs_reference.py).
 */
#include "mosfhet_oracle.h"
#include <string.h>
#include <stdlib.h>

#define W 64

/* src/misc.c:13-15  double2torus: (Torus)(int64_t)(2^64 * x) */
Torus orc_double2torus(double x) { return (Torus)((int64_t)(18446744073709551616.0 * x)); }

/* src/misc.c:18-22  torus2int: round(x * 2^log_scale / 2^64) */
uint64_t orc_torus2int(Torus x, int log_scale) {
  const Torus half = (Torus)1 << (W - log_scale - 1);
  return (x + half) >> (W - log_scale);
}

/* src/misc.c:25-28 */
Torus orc_int2torus(uint64_t x, int log_scale) { return x << (W - log_scale); }

/* shared by decompose_i / decompose: sum_{i<l} 2^(W-1-i*Bg) */
static Torus digit_offset(int Bg_bit, int l) {
  Torus off = 0;
  for (int i = 0; i < l; i++) off += (Torus)1 << (W - 1 - i * Bg_bit);
  return off;
}

/* src/polynomial.c:74-89  polynomial_decompose_i: rounded signed digit i of every coefficient.
 * offset = 2^(W-1-l*Bg) (rounding) + sum_{j<l} 2^(W-1-j*Bg) (recentring); digit in [-Bg/2, Bg/2)
 * stored two's complement. */
void orc_poly_decompose_i(Torus *out, const Torus *in, int N, int Bg_bit, int l, int i) {
  const Torus off = digit_offset(Bg_bit, l) + ((Torus)1 << (W - 1 - l * Bg_bit));
  const Torus mask = ((Torus)1 << Bg_bit) - 1, half = (Torus)1 << (Bg_bit - 1);
  const int shift = W - (i + 1) * Bg_bit;
  for (int c = 0; c < N; c++) out[c] = (((in[c] + off) >> shift) & mask) - half;
}

/* src/polynomial.c:55-72  polynomial_decompose: all l digits, WITHOUT the rounding term. */
void orc_poly_decompose(Torus *out, const Torus *in, int N, int Bg_bit, int l) {
  const Torus off = digit_offset(Bg_bit, l);
  const Torus mask = ((Torus)1 << Bg_bit) - 1, half = (Torus)1 << (Bg_bit - 1);
  for (int i = 0; i < l; i++) {
    const int shift = W - (i + 1) * Bg_bit;
    for (int c = 0; c < N; c++) out[(size_t)i * N + c] = (((in[c] + off) >> shift) & mask) - half;
  }
}

/* coefficient i of in * X^a, a already reduced to [0, 2N): sign flips once per wrap. */
static inline Torus rot_coeff(const Torus *in, int N, int a, int i) {
  int src = i - a, neg = 0;
  while (src < 0) { src += N; neg ^= 1; }
  return neg ? (Torus)0 - in[src] : in[src];
}

/* src/polynomial.c:184-199 */
void orc_poly_mul_by_xai(Torus *out, const Torus *in, int N, int a) {
  a &= 2 * N - 1;
  for (int i = 0; i < N; i++) out[i] = rot_coeff(in, N, a, i);
}

/* src/polynomial.c:202-217 */
void orc_poly_mul_by_xai_addto(Torus *out, const Torus *in, int N, int a) {
  a &= 2 * N - 1;
  for (int i = 0; i < N; i++) out[i] += rot_coeff(in, N, a, i);
}

/* src/polynomial.c:220-235  (a == 0 gives the zero polynomial) */
void orc_poly_mul_by_xai_minus_1(Torus *out, const Torus *in, int N, int a) {
  a &= 2 * N - 1;
  for (int i = 0; i < N; i++) out[i] = rot_coeff(in, N, a, i) - in[i];
}

/* src/polynomial.c:264-274  exact negacyclic out += in1 * in2 */
void orc_poly_naive_mul_addto(Torus *out, const Torus *in1, const Torus *in2, int N) {
  for (int i = 0; i < N; i++) {
    const Torus m = in2[i];
    if (!m) continue;
    for (int j = i; j < N; j++) out[j] += in1[j - i] * m;
    for (int j = 0; j < i; j++) out[j] -= in1[N + j - i] * m;
  }
}

/* src/polynomial.c:290-303 */
void orc_poly_naive_mul(Torus *out, const Torus *in1, const Torus *in2, int N) {
  memset(out, 0, sizeof(Torus) * (size_t)N);
  orc_poly_naive_mul_addto(out, in1, in2, N);
}

/* src/polynomial.c:442-450  X -> X^gen (gen odd) */
void orc_poly_permute(Torus *out, const Torus *in, int N, uint64_t gen) {
  const uint64_t mask = (uint64_t)N - 1;
  for (uint64_t i = 0; i < (uint64_t)N; i++) {
    const uint64_t idx = i * gen;
    out[idx & mask] = (idx & (uint64_t)N) ? (Torus)0 - in[i] : in[i];
  }
}

/* src/trlwe.c:540-552  sample extract at coefficient idx */
void orc_trlwe_extract_tlwe(Torus *out, const Torus *in, int k, int N, int idx) {
  for (int p = 0; p < k; p++) {
    const Torus *ap = in + (size_t)p * N;
    for (int j = 0; j <= idx; j++) out[p * N + j] = ap[idx - j];
    for (int j = idx + 1; j < N; j++) out[p * N + j] = (Torus)0 - ap[N + idx - j];
  }
  out[(size_t)k * N] = in[(size_t)k * N + idx];
}

/* src/trlwe.c:662-667  trivial TRLWE whose b holds `size` LUT slots of N/size coefficients */
void orc_trlwe_torus_packing(Torus *out, const Torus *lut, int k, int N, int size) {
  memset(out, 0, sizeof(Torus) * (size_t)k * N);
  for (int i = 0; i < N; i++) out[(size_t)k * N + i] = lut[i / (N / size)];
}

/* src/tlwe.c:289-303  LWE -> LWE key switch by table lookup.
 * ksk[i][j][v-1] = TLWE_out(s_in[i] * v * 2^(W-(j+1)bb)),  rows of n_out+1 words. */
void orc_tlwe_keyswitch(Torus *out, const Torus *in, const Torus *ksk,
                        int n_in, int n_out, int t, int base_bit) {
  const Torus round_off = (Torus)1 << (W - 1 - base_bit * t);
  const Torus mask = ((Torus)1 << base_bit) - 1;
  const size_t row = (size_t)n_out + 1, per_j = ((size_t)1 << base_bit) - 1;
  memset(out, 0, sizeof(Torus) * n_out);
  out[n_out] = in[n_in];
  for (int i = 0; i < n_in; i++) {
    const Torus ai = in[i] + round_off;
    for (int j = 0; j < t; j++) {
      const Torus v = (ai >> (W - (j + 1) * base_bit)) & mask;
      if (!v) continue;
      const Torus *r = ksk + (((size_t)i * t + j) * per_j + (v - 1)) * row;
      for (size_t c = 0; c < row; c++) out[c] -= r[c];
    }
  }
}

/* src/tlwe.c:135-141 */
Torus orc_tlwe_phase(const Torus *c, const Torus *s, int n) {
  Torus sa = 0;
  for (int i = 0; i < n; i++) sa += s[i] * c[i];
  return c[n] - sa;
}

/* src/trlwe.c:324-331, with the exact product so the oracle's phase carries no FFT error */
void orc_trlwe_phase(Torus *out, const Torus *c, const Torus *s, int k, int N) {
  memset(out, 0, sizeof(Torus) * (size_t)N);
  for (int p = 0; p < k; p++) orc_poly_naive_mul_addto(out, c + (size_t)p * N, s + (size_t)p * N, N);
  for (int i = 0; i < N; i++) out[i] = c[(size_t)k * N + i] - out[i];
}

/* src/bootstrap.c:208-217  programmable_bootstrap's scaling/rounding of the input sample */
void orc_pbs_preprocess(Torus *out, const Torus *in, int n, int N, int kappa, int theta) {
  int log_2N = 1;
  while ((1 << log_2N) < 2 * N) log_2N++;
  const Torus rnd = (Torus)1 << (W - log_2N + theta - 1);
  const Torus msk = ~(((Torus)1 << (W - log_2N + theta)) - 1);
  for (int i = 0; i <= n; i++) out[i] = ((in[i] << kappa) + rnd) & msk;
}
