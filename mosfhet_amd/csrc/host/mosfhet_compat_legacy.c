/*
 * mosfhet_compat_legacy.c -- the single-object helpers of the reference's API that sit either side of the bootstrap path (SURVEY.md section 8 rows
 * a9, a11, a13, a19, a24, a27, a28 name them): what the reference's own tests and applications call between two bootstraps.
 *
 *   torus domain (host structs, exact integer arithmetic mod 2^64; done where the data lives, like trlwe_extract_tlwe and tlwe_addto):
 *       polynomial_{zero,copy,negate,add,addto,sub,subto}_torus_polynomial(s), polynomial_torus_scale[2], polynomial_decompose[_i], trlwe_decompose,
 *       torus_polynomial_mul_by_xai[_addto,_minus_1], trlwe_mul_by_xai_{addto,minus_1}, trgsw_mul_by_xai[_addto,_minus_1], trgsw_{add,addto,sub,copy},
 *       the exact O(N^2) products polynomial_naive_mul_*, trgsw_[new_]noiseless_trivial_sample, trgsw_new_{sample,monomial_sample,exp_sample},
 *       trlwe_LUT_packing, trlwe_scale
 *   DFT domain (device-resident objects of mosfhet_compat_dft.c; every operation is a kernel launch, nothing is computed on the host):
 *       polynomial_{add,sub,scale_and_add}_DFT_polynomials, trlwe_DFT_{add,addto,sub,copy,mul_by_polynomial,mul_addto_by_polynomial,phase},
 *       trlwe_[new_]noiseless_trivial_DFT_sample, trgsw_DFT_{add,sub,copy,mul_addto_by_polynomial}, trgsw_from_DFT, trgsw_mul_DFT[2],
 *       trgsw_monomial_DFT_sample, trgsw_mul_trlwe_DFT_prefetch, polynomial_mul[_addto]_torus
 *   unfolded blind rotation with caller-held key material: blind_rotate_unfolded, multivalue_bootstrap_UBR_phase1 / _phase2
 * k = 1 wherever a device object is involved (every parameter set of the reference, test/tests.c:37-62).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "compat_internal.h"

#define W 64

static mosfhet_hip_ctx_t ectx(void) { return (mosfhet_hip_ctx_t)mosfhet_engine_ctx(); }
static void check_rc(int rc, const char *what) {
  if (rc || mosfhet_hip_ctx_sync(ectx(), NULL)) mc_die(what);
}
static void need(int cond, const char *what) {
  if (cond) return;
  fprintf(stderr, "mosfhet_amd: %s\n", what);
  abort();
}

/* ================================================================== torus polynomials (host) */
void polynomial_zero_torus_polynomial(TorusPolynomial p) { memset(p->coeffs, 0, sizeof(Torus) * (size_t)p->N); }
void polynomial_copy_torus_polynomial(TorusPolynomial out, TorusPolynomial in) { memmove(out->coeffs, in->coeffs, sizeof(Torus) * (size_t)in->N); }
void polynomial_negate_torus_polynomial(TorusPolynomial out, TorusPolynomial in) {
  for (int i = 0; i < in->N; i++) out->coeffs[i] = (Torus)0 - in->coeffs[i];
}
void polynomial_add_torus_polynomials(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2) {
  for (int i = 0; i < in2->N; i++) out->coeffs[i] = in1->coeffs[i] + in2->coeffs[i];
}
void polynomial_addto_torus_polynomial(TorusPolynomial out, TorusPolynomial in) { polynomial_add_torus_polynomials(out, out, in); }
void polynomial_sub_torus_polynomials(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2) {
  for (int i = 0; i < in2->N; i++) out->coeffs[i] = in1->coeffs[i] - in2->coeffs[i];
}
void polynomial_subto_torus_polynomial(TorusPolynomial out, TorusPolynomial in) { polynomial_sub_torus_polynomials(out, out, in); }

TorusPolynomial *polynomial_new_array_of_torus_polynomials(int N, int size) {
  TorusPolynomial *r = (TorusPolynomial *)mc_xmalloc(sizeof(TorusPolynomial) * (size_t)(size > 0 ? size : 1));
  for (int i = 0; i < size; i++) r[i] = polynomial_new_torus_polynomial(N);
  return r;
}

/* src/polynomial.c:319-325: out[i] = torus2int(in[i], log_scale), the rounded top log_scale bits */
void polynomial_torus_scale(TorusPolynomial out, TorusPolynomial in, int log_scale) {
  for (int i = 0; i < in->N; i++) out->coeffs[i] = torus2int(in->coeffs[i], log_scale);
}
void polynomial_torus_scale2(TorusPolynomial out, TorusPolynomial in, uint64_t scale) {
  for (int i = 0; i < in->N; i++) out->coeffs[i] = in->coeffs[i] * scale;
}

/* Gadget digits.  Digit i of x is bits [W - (i+1) Bg_bit, W - i Bg_bit) of x + offset, recentred to [-Bg/2, Bg/2) and kept as a two's complement word.
 * polynomial_decompose (src/polynomial.c:55-72, public_mux's variant) uses offset = sum_{j<l} 2^(W-1 - j Bg_bit); polynomial_decompose_i (:74-89, the
 * external product's) adds the rounding term 2^(W-1 - l Bg_bit) on top. */
static Torus gadget_offset(int Bg_bit, int l, int rounding) {
  Torus off = rounding ? (Torus)1 << (W - 1 - l * Bg_bit) : 0;
  for (int j = 0; j < l; j++) off += (Torus)1 << (W - 1 - j * Bg_bit);
  return off;
}
static void digits_of(Torus *out, const Torus *in, int N, int Bg_bit, int i, Torus off) {
  const int shift = W - (i + 1) * Bg_bit;
  const Torus mask = ((Torus)1 << Bg_bit) - 1, half = (Torus)1 << (Bg_bit - 1);
  for (int c = 0; c < N; c++) out[c] = (((in[c] + off) >> shift) & mask) - half;
}
void polynomial_decompose_i(TorusPolynomial out, TorusPolynomial in, int Bg_bit, int l, int i) {
  digits_of(out->coeffs, in->coeffs, in->N, Bg_bit, i, gadget_offset(Bg_bit, l, 1));
}
void polynomial_decompose(TorusPolynomial *out, TorusPolynomial in, int Bg_bit, int l) {
  const Torus off = gadget_offset(Bg_bit, l, 0);
  for (int i = 0; i < l; i++) digits_of(out[i]->coeffs, in->coeffs, in->N, Bg_bit, i, off);
}
/* src/trlwe.c:636-641: the (k+1) l digit polynomials of a sample, mask components first (polynomial_decompose's digits) */
void trlwe_decompose(TorusPolynomial *out, TRLWE in, int Bg_bit, int l) {
  for (int p = 0; p < in->k; p++) polynomial_decompose(out + (size_t)p * l, in->a[p], Bg_bit, l);
  polynomial_decompose(out + (size_t)in->k * l, in->b, Bg_bit, l);
}

/* X^a-rotations, a taken mod 2N (src/polynomial.c:184-235).  Coefficient i of in * X^a is +-in[(i - a) mod N], negative when i - a wraps an odd number
 * of times.  mode 0: out = in X^a; 1: out += in X^a; 2: out = in (X^a - 1).  out != in, as in the reference (asserted there). */
static void rotate_into(Torus *out, const Torus *in, int N, int a, int mode) {
  a &= 2 * N - 1;
  need(out != in, "torus_polynomial_mul_by_xai*: out and in must differ (src/polynomial.c:185)");
  const int flip = a >= N;            /* X^N = -1 */
  const int r = flip ? a - N : a;     /* rotation by r < N, then the sign */
  for (int i = 0; i < N; i++) {
    const int wrapped = i < r;
    Torus v = in[wrapped ? i - r + N : i - r];
    if (wrapped != flip) v = (Torus)0 - v;
    if (mode == 0) out[i] = v;
    else if (mode == 1) out[i] += v;
    else out[i] = v - in[i];
  }
}
void torus_polynomial_mul_by_xai(TorusPolynomial out, TorusPolynomial in, int a) { rotate_into(out->coeffs, in->coeffs, out->N, a, 0); }
void torus_polynomial_mul_by_xai_addto(TorusPolynomial out, TorusPolynomial in, int a) { rotate_into(out->coeffs, in->coeffs, out->N, a, 1); }
void torus_polynomial_mul_by_xai_minus_1(TorusPolynomial out, TorusPolynomial in, int a) { rotate_into(out->coeffs, in->coeffs, out->N, a, 2); }

static void trlwe_rotate(TRLWE out, TRLWE in, int a, int mode) {
  for (int p = 0; p < in->k; p++) rotate_into(out->a[p]->coeffs, in->a[p]->coeffs, in->b->N, a, mode);
  rotate_into(out->b->coeffs, in->b->coeffs, in->b->N, a, mode);
}
void trlwe_mul_by_xai_addto(TRLWE out, TRLWE in, int a) { trlwe_rotate(out, in, a, 1); }       /* src/trlwe.c:464-469 */
void trlwe_mul_by_xai_minus_1(TRLWE out, TRLWE in, int a) { trlwe_rotate(out, in, a, 2); }     /* src/trlwe.c:471-476 */

/* exact negacyclic products (src/polynomial.c:237-317): schoolbook, mod X^N + 1 and mod 2^64; what the reference's tests measure the FFT products against */
static void schoolbook(Torus *out, const Torus *x, const Torus *y, int N, int accumulate) {
  if (!accumulate) memset(out, 0, sizeof(Torus) * (size_t)N);
  for (int i = 0; i < N; i++) {
    const Torus yi = y[i];
    if (!yi) continue;
    for (int j = 0; j < N - i; j++) out[i + j] += x[j] * yi;
    for (int j = N - i; j < N; j++) out[i + j - N] -= x[j] * yi;
  }
}
void polynomial_naive_mul_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2) {
  need(out != in1 && out != in2, "polynomial_naive_mul_torus: out must not alias an input");
  schoolbook(out->coeffs, in1->coeffs, in2->coeffs, in2->N, 0);
}
void polynomial_naive_mul_addto_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2) {
  need(out != in1 && out != in2, "polynomial_naive_mul_addto_torus: out must not alias an input");
  schoolbook(out->coeffs, in1->coeffs, in2->coeffs, in2->N, 1);
}

/* ================================================================== TRLWE / TRGSW in the torus domain (host) */
void trlwe_scale(TRLWE out, TRLWE in, uint64_t scale) {   /* src/trlwe.c:269-274 */
  for (int p = 0; p < in->k; p++) polynomial_torus_scale2(out->a[p], in->a[p], scale);
  polynomial_torus_scale2(out->b, in->b, scale);
}

/* src/trlwe.c:669-675: 2^in_prec table entries of out_prec bits each, spread over the N coefficients of a trivial sample */
void trlwe_LUT_packing(TRLWE out, uint64_t *in, uint64_t in_prec, uint64_t out_prec) {
  const int N = out->b->N;
  const int slot = N >> in_prec;
  need(slot >= 1, "trlwe_LUT_packing: more table entries than coefficients");
  trlwe_noiseless_trivial_sample(out, NULL);
  for (int i = 0; i < N; i++) out->b->coeffs[i] = int2torus(in[i / slot], (int)out_prec);
}

static int trgsw_rows(TRGSW g) { return (g->samples[0]->k + 1) * g->l; }
void trgsw_add(TRGSW out, TRGSW in1, TRGSW in2) {
  for (int r = 0; r < trgsw_rows(in1); r++) trlwe_add(out->samples[r], in1->samples[r], in2->samples[r]);
}
void trgsw_addto(TRGSW out, TRGSW in) { trgsw_add(out, out, in); }
void trgsw_sub(TRGSW out, TRGSW in1, TRGSW in2) {
  for (int r = 0; r < trgsw_rows(in1); r++) trlwe_sub(out->samples[r], in1->samples[r], in2->samples[r]);
}
void trgsw_copy(TRGSW out, TRGSW in) {
  for (int r = 0; r < trgsw_rows(in); r++) trlwe_copy(out->samples[r], in->samples[r]);
}
void trgsw_mul_by_xai(TRGSW out, TRGSW in, int a) {
  for (int r = 0; r < trgsw_rows(in); r++) trlwe_rotate(out->samples[r], in->samples[r], a, 0);
}
void trgsw_mul_by_xai_addto(TRGSW out, TRGSW in, int a) {
  for (int r = 0; r < trgsw_rows(in); r++) trlwe_rotate(out->samples[r], in->samples[r], a, 1);
}
void trgsw_mul_by_xai_minus_1(TRGSW out, TRGSW in, int a) {
  for (int r = 0; r < trgsw_rows(in); r++) trlwe_rotate(out->samples[r], in->samples[r], a, 2);
}

/* src/trgsw.c:128-150: the gadget matrix times m with no mask and no noise: row p l + i carries m 2^(W - (i+1) Bg_bit) on component p, coefficient 0 */
void trgsw_noiseless_trivial_sample(TRGSW out, Torus m, int l, int Bg_bit, int k, int N) {
  (void)N;
  for (int r = 0; r < (k + 1) * l; r++) trlwe_noiseless_trivial_sample(out->samples[r], NULL);
  for (int i = 0; i < l; i++) {
    const Torus h = m << (W - (i + 1) * Bg_bit);
    for (int p = 0; p < k; p++) out->samples[p * l + i]->a[p]->coeffs[0] = h;
    out->samples[k * l + i]->b->coeffs[0] = h;
  }
}
TRGSW trgsw_new_noiseless_trivial_sample(Torus m, int l, int Bg_bit, int k, int N) {
  TRGSW g = trgsw_alloc_new_sample(l, Bg_bit, k, N);
  trgsw_noiseless_trivial_sample(g, m, l, Bg_bit, k, N);
  return g;
}
TRGSW trgsw_new_monomial_sample(int64_t m, int e, TRGSW_Key key) {   /* src/trgsw.c:178-184: TRGSW(m X^e) */
  TRGSW g = trgsw_alloc_new_sample(key->l, key->Bg_bit, key->trlwe_key->k, key->trlwe_key->s[0]->N);
  trgsw_monomial_sample(g, m, e, key);
  return g;
}
TRGSW trgsw_new_sample(Torus m, TRGSW_Key key) { return trgsw_new_monomial_sample((int64_t)m, 0, key); }   /* src/trgsw.c:186-188 */
TRGSW trgsw_new_exp_sample(int e, TRGSW_Key key) { return trgsw_new_monomial_sample(1, e, key); }          /* src/trgsw.c:271-273 */

/* ================================================================== DFT domain (device) */
static double *poly_dev(DFT_Polynomial p, const char *who) {
  need(p && (mc_poly_kind(p) == MC_POLY_DFT_OWNER || mc_poly_kind(p) == MC_POLY_DFT_VIEW || mc_poly_kind(p) == MC_POLY_DFT_SHARED), who);
  return p->coeffs;
}
static void lincomb(double *out, const double *a, const double *b, double cb, size_t n, const char *who) {
  check_rc(mosfhet_hip_dft_lincomb_batch(ectx(), out, a, b, cb, n, NULL), who);
}
static const char *kNotOurs = "DFT-domain argument was not made by this library's allocators";

void polynomial_add_DFT_polynomials(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2) {
  lincomb(poly_dev(out, kNotOurs), poly_dev(in1, kNotOurs), poly_dev(in2, kNotOurs), 1.0, (size_t)in2->N, "polynomial_add_DFT_polynomials");
}
void polynomial_sub_DFT_polynomials(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2) {
  lincomb(poly_dev(out, kNotOurs), poly_dev(in1, kNotOurs), poly_dev(in2, kNotOurs), -1.0, (size_t)in2->N, "polynomial_sub_DFT_polynomials");
}
void polynomial_scale_and_add_DFT_polynomials(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2, uint64_t scale) {   /* out = in1 + scale in2 */
  lincomb(poly_dev(out, kNotOurs), poly_dev(in1, kNotOurs), poly_dev(in2, kNotOurs), (double)scale, (size_t)in2->N, "polynomial_scale_and_add_DFT_polynomials");
}

/* a TRLWE_DFT made by trlwe_alloc_new_DFT_sample[_array] is one block [2][N/2] complex; one made by hand from two polynomials is handled per component */
static int trlwe_dft_is_block(TRLWE_DFT c) { return c->k == 1 && c->b->coeffs == c->a[0]->coeffs + c->b->N; }
static void trlwe_dft_lincomb(TRLWE_DFT out, TRLWE_DFT a, TRLWE_DFT b, double cb, const char *who) {
  need(out->k == 1 && b->k == 1 && (!a || a->k == 1), "TRLWE_DFT: k = 1 only");
  const size_t N = (size_t)b->b->N;
  if (trlwe_dft_is_block(out) && trlwe_dft_is_block(b) && (!a || trlwe_dft_is_block(a))) {
    lincomb(poly_dev(out->a[0], kNotOurs), a ? poly_dev(a->a[0], kNotOurs) : NULL, poly_dev(b->a[0], kNotOurs), cb, 2 * N, who);
    return;
  }
  lincomb(poly_dev(out->a[0], kNotOurs), a ? poly_dev(a->a[0], kNotOurs) : NULL, poly_dev(b->a[0], kNotOurs), cb, N, who);
  lincomb(poly_dev(out->b, kNotOurs), a ? poly_dev(a->b, kNotOurs) : NULL, poly_dev(b->b, kNotOurs), cb, N, who);
}
void trlwe_DFT_add(TRLWE_DFT out, TRLWE_DFT in1, TRLWE_DFT in2) { trlwe_dft_lincomb(out, in1, in2, 1.0, "trlwe_DFT_add"); }
void trlwe_DFT_addto(TRLWE_DFT out, TRLWE_DFT in) { trlwe_dft_lincomb(out, out, in, 1.0, "trlwe_DFT_addto"); }
void trlwe_DFT_sub(TRLWE_DFT out, TRLWE_DFT in1, TRLWE_DFT in2) { trlwe_dft_lincomb(out, in1, in2, -1.0, "trlwe_DFT_sub"); }
void trlwe_DFT_copy(TRLWE_DFT out, TRLWE_DFT in) { trlwe_dft_lincomb(out, NULL, in, 1.0, "trlwe_DFT_copy"); }

/* src/trlwe.c:491-505: every component of the sample times one polynomial (pointwise in the DFT domain) */
static void trlwe_dft_mul_poly(TRLWE_DFT out, TRLWE_DFT in, DFT_Polynomial poly, int addto, const char *who) {
  need(out->k == 1 && in->k == 1, "TRLWE_DFT: k = 1 only");
  const int N = in->b->N;
  const double *p = poly_dev(poly, kNotOurs);
  check_rc(mosfhet_hip_dft_mul_batch(ectx(), poly_dev(out->a[0], kNotOurs), poly_dev(in->a[0], kNotOurs), p, N, 1, addto, NULL), who);
  check_rc(mosfhet_hip_dft_mul_batch(ectx(), poly_dev(out->b, kNotOurs), poly_dev(in->b, kNotOurs), p, N, 1, addto, NULL), who);
}
void trlwe_DFT_mul_by_polynomial(TRLWE_DFT out, TRLWE_DFT in, DFT_Polynomial in2) { trlwe_dft_mul_poly(out, in, in2, 0, "trlwe_DFT_mul_by_polynomial"); }
void trlwe_DFT_mul_addto_by_polynomial(TRLWE_DFT out, TRLWE_DFT in, DFT_Polynomial in2) { trlwe_dft_mul_poly(out, in, in2, 1, "trlwe_DFT_mul_addto_by_polynomial"); }

void trlwe_noiseless_trivial_DFT_sample(TRLWE_DFT out, DFT_Polynomial m) {   /* src/trlwe.c:282-289: (0, m) */
  need(out->k == 1, "TRLWE_DFT: k = 1 only");
  const size_t bytes = sizeof(double) * (size_t)out->b->N;
  mc_use_device();
  if (hipMemset(poly_dev(out->a[0], kNotOurs), 0, bytes)) mc_die("trlwe_noiseless_trivial_DFT_sample");
  if (m) mc_dev_copy(poly_dev(out->b, kNotOurs), poly_dev(m, kNotOurs), bytes, HIP_D2D);
  else if (hipMemset(poly_dev(out->b, kNotOurs), 0, bytes)) mc_die("trlwe_noiseless_trivial_DFT_sample");
}
TRLWE_DFT trlwe_new_noiseless_trivial_DFT_sample(DFT_Polynomial m, int k, int N) {
  TRLWE_DFT c = trlwe_alloc_new_DFT_sample(k, N);
  trlwe_noiseless_trivial_DFT_sample(c, m);
  return c;
}

/* src/trlwe.c:372-382: phase of a DFT-domain sample, b - sum a_i s_i computed in the DFT domain, then one inverse transform */
void trlwe_DFT_phase(TorusPolynomial out, TRLWE_DFT in, TRLWE_Key key) {
  need(in->k == 1 && key->k == 1, "trlwe_DFT_phase: k = 1 only");
  const int N = out->N;
  DFT_Polynomial s = polynomial_new_DFT_polynomial(N), acc = polynomial_new_DFT_polynomial(N);
  polynomial_torus_to_DFT(s, key->s[0]);
  polynomial_mul_DFT(acc, in->a[0], s);
  polynomial_sub_DFT_polynomials(acc, in->b, acc);
  polynomial_DFT_to_torus(out, acc);
  free_polynomial(s);
  free_polynomial(acc);
}

/* FFT products of torus polynomials (src/polynomial.c:281-303): both factors transformed, multiplied pointwise, transformed back */
void polynomial_mul_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2) {
  const int N = in2->N;
  DFT_Polynomial *t = polynomial_new_array_of_polynomials_DFT(N, 2);
  polynomial_torus_to_DFT(t[0], in1);
  polynomial_torus_to_DFT(t[1], in2);
  polynomial_mul_DFT(t[0], t[0], t[1]);
  polynomial_DFT_to_torus(out, t[0]);
  free_array_of_polynomials(t, 2);
}
void polynomial_mul_addto_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2) {
  TorusPolynomial prod = polynomial_new_torus_polynomial(in2->N);
  polynomial_mul_torus(prod, in1, in2);
  polynomial_addto_torus_polynomial(out, prod);
  free_polynomial(prod);
}

/* ---- TRGSW_DFT: one block [(k+1) l][k+1][N/2] complex when made by trgsw_alloc_new_DFT_sample[_array]; else row by row */
static int trgsw_dft_rows(TRGSW_DFT g) { return 2 * g->l; }
static int trgsw_dft_is_block(TRGSW_DFT g) {
  const size_t row = (size_t)2 * g->samples[0]->b->N;
  for (int r = 0; r < trgsw_dft_rows(g); r++)
    if (!trlwe_dft_is_block(g->samples[r]) || g->samples[r]->a[0]->coeffs != g->samples[0]->a[0]->coeffs + (size_t)r * row) return 0;
  return 1;
}
static void trgsw_dft_lincomb(TRGSW_DFT out, TRGSW_DFT a, TRGSW_DFT b, double cb, const char *who) {
  need(out->l == b->l && (!a || a->l == b->l), "TRGSW_DFT: gadget sizes differ");
  if (trgsw_dft_is_block(out) && trgsw_dft_is_block(b) && (!a || trgsw_dft_is_block(a))) {
    const size_t n = (size_t)trgsw_dft_rows(b) * 2 * (size_t)b->samples[0]->b->N;
    lincomb(poly_dev(out->samples[0]->a[0], kNotOurs), a ? poly_dev(a->samples[0]->a[0], kNotOurs) : NULL, poly_dev(b->samples[0]->a[0], kNotOurs), cb, n, who);
    return;
  }
  for (int r = 0; r < trgsw_dft_rows(b); r++) trlwe_dft_lincomb(out->samples[r], a ? a->samples[r] : NULL, b->samples[r], cb, who);
}
void trgsw_DFT_add(TRGSW_DFT out, TRGSW_DFT in1, TRGSW_DFT in2) { trgsw_dft_lincomb(out, in1, in2, 1.0, "trgsw_DFT_add"); }
void trgsw_DFT_sub(TRGSW_DFT out, TRGSW_DFT in1, TRGSW_DFT in2) { trgsw_dft_lincomb(out, in1, in2, -1.0, "trgsw_DFT_sub"); }
void trgsw_DFT_copy(TRGSW_DFT out, TRGSW_DFT in) {
  trgsw_dft_lincomb(out, NULL, in, 1.0, "trgsw_DFT_copy");
  out->Bg_bit = in->Bg_bit;
}
void trgsw_DFT_mul_addto_by_polynomial(TRGSW_DFT out, TRGSW_DFT in1, DFT_Polynomial in2) {   /* src/trgsw.c:449-454 */
  for (int r = 0; r < trgsw_dft_rows(in1); r++) trlwe_dft_mul_poly(out->samples[r], in1->samples[r], in2, 1, "trgsw_DFT_mul_addto_by_polynomial");
}

void trgsw_from_DFT(TRGSW out, TRGSW_DFT in) {   /* src/trgsw.c:351-357 */
  for (int r = 0; r < trgsw_dft_rows(in); r++) trlwe_from_DFT(out->samples[r], in->samples[r]);
}

/* src/trgsw.c:425-431: row i of the result = in2 (.) row i of in1 -- one launch over the (k+1) l rows of in1 against the one selector in2 */
void trgsw_mul_DFT(TRGSW_DFT out, TRGSW in1, TRGSW_DFT in2) {
  need(out != in2, "trgsw_mul_DFT: out must differ from in2 (src/trgsw.c:426)");
  const int rows = trgsw_rows(in1), N = in1->samples[0]->b->N;
  need(in1->samples[0]->k == 1 && rows == trgsw_dft_rows(out) && trgsw_dft_is_block(out) && trgsw_dft_is_block(in2),
       "trgsw_mul_DFT: k = 1 and DFT samples made by trgsw_alloc_new_DFT_sample");
  const size_t words = (size_t)rows * 2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * words), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * words);
  for (int r = 0; r < rows; r++) mc_trlwe_to_flat(h + (size_t)r * 2 * N, in1->samples[r]);
  mc_dev_copy(d, h, sizeof(Torus) * words, HIP_H2D);
  check_rc(mosfhet_hip_external_product_dft_batch(ectx(), in2->samples[0]->a[0]->coeffs, 0, out->samples[0]->a[0]->coeffs, d, N, in2->l, in2->Bg_bit, rows, NULL),
           "trgsw_mul_DFT");
  out->Bg_bit = in2->Bg_bit;
  mc_hstage_free(h);
}
/* src/trgsw.c:433-447: the same with in1 in the DFT domain: its rows go back to the torus domain first (on the device), then as above */
void trgsw_mul_DFT2(TRGSW_DFT out, TRGSW_DFT in1, TRGSW_DFT in2) {
  need(out != in2, "trgsw_mul_DFT2: out must differ from in2 (src/trgsw.c:434)");
  const int rows = trgsw_dft_rows(in1), N = in1->samples[0]->b->N;
  need(rows == trgsw_dft_rows(out) && trgsw_dft_is_block(out) && trgsw_dft_is_block(in1) && trgsw_dft_is_block(in2),
       "trgsw_mul_DFT2: DFT samples made by trgsw_alloc_new_DFT_sample");
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (size_t)rows * 2 * N);
  check_rc(mosfhet_hip_dft_to_torus_batch(ectx(), d, in1->samples[0]->a[0]->coeffs, N, rows * 2, NULL), "trgsw_mul_DFT2");
  check_rc(mosfhet_hip_external_product_dft_batch(ectx(), in2->samples[0]->a[0]->coeffs, 0, out->samples[0]->a[0]->coeffs, d, N, in2->l, in2->Bg_bit, rows, NULL),
           "trgsw_mul_DFT2");
  out->Bg_bit = in2->Bg_bit;
}
void trgsw_mul_trlwe_DFT_prefetch(TRLWE_DFT out, TRLWE in1, TRGSW_DFT in2) { trgsw_mul_trlwe_DFT(out, in1, in2); }   /* src/trgsw.c: a cache-hinted twin */

void trgsw_monomial_DFT_sample(TRGSW_DFT out, int64_t m, int e, TRGSW_Key key) {   /* src/trgsw.c:170-175 */
  TRGSW g = trgsw_new_monomial_sample(m, e, key);
  trgsw_to_DFT(out, g);
  free_trgsw(g);
}

/* ================================================================== unfolded blind rotation on caller-held key material (src/bootstrap.c:124-190) */
/* blind_rotate_unfolded(tv, a, s, size, unfolding): s = the 2^u / u * size torus-domain TRGSW samples of new_bootstrap_key(.., unfolding)'s `su`
 * layout.  The samples go to the device as a temporary unfolded key; the rotation itself is the kernel behind functional_bootstrap with such a key. */
void blind_rotate_unfolded(TRLWE tv, Torus *a, TRGSW *s, int size, int unfolding) {
  need(unfolding == 2 || unfolding == 4 || unfolding == 8, "blind_rotate_unfolded: unfolding must be 2, 4 or 8");
  need(size % unfolding == 0, "blind_rotate_unfolded: size must be a multiple of the unfolding");
  const int l = s[0]->l, Bg_bit = s[0]->Bg_bit, N = tv->b->N, rows = 2 * l;
  need(tv->k == 1 && s[0]->samples[0]->k == 1, "blind_rotate_unfolded: k = 1 only");
  const size_t entries = (size_t)size * ((size_t)1 << unfolding) / unfolding, esz = (size_t)rows * 2 * N;
  Torus *flat = (Torus *)mc_xmalloc(sizeof(Torus) * entries * esz);
  for (size_t e = 0; e < entries; e++)
    for (int r = 0; r < rows; r++) mc_trlwe_to_flat(flat + e * esz + (size_t)r * 2 * N, s[e]->samples[r]);
  mosfhet_hip_bsk_t key = NULL;
  if (mosfhet_hip_bsk_unfolded_create(ectx(), &key, flat, size, N, l, Bg_bit, unfolding)) mc_die("blind_rotate_unfolded");
  free(flat);
  /* the unfolded kernel starts from tv X^(-bbar), bbar = torus2int(b + 2^64 / (4 torus_base)): with torus_base = 1 and b = -2^62 that rotation is the
   * identity, which leaves the blind rotation alone (src/bootstrap.c:124-149) */
  const size_t acc_w = (size_t)2 * N, words = 2 * acc_w + (size_t)size + 1;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * words), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * words);
  mc_trlwe_to_flat(h, tv);
  memcpy(h + acc_w, a, sizeof(Torus) * (size_t)size);
  h[acc_w + size] = (Torus)0 - ((Torus)1 << 62);
  mc_dev_copy(d, h, sizeof(Torus) * (acc_w + (size_t)size + 1), HIP_H2D);
  Torus *d_out = d + acc_w + size + 1;
  check_rc(mosfhet_hip_functional_bootstrap_wo_extract_batch(ectx(), key, d_out, d, 1, d + acc_w, 1, 1, NULL), "blind_rotate_unfolded");
  mc_dev_copy(h, d_out, sizeof(Torus) * acc_w, HIP_D2H);
  mc_trlwe_from_flat(tv, h);
  mc_hstage_free(h);
  mosfhet_hip_bsk_destroy(key);
}

/* the n / u selectors of one input, out[g] = DFT(sum_j X^(sum of the group's mask words selected by j) su[g][j]) */
void multivalue_bootstrap_UBR_phase1(TRGSW_DFT *out, TLWE in, Bootstrap_Key key) {
  need(key->unfolding > 1, "multivalue_bootstrap_UBR_phase1: needs a key made with unfolding > 1 (src/bootstrap.c:156)");
  const int n = key->n, groups = n / key->unfolding, N = key->N;
  const size_t esz = (size_t)2 * key->l * 2 * N;
  double *base = out[0]->samples[0]->a[0]->coeffs;
  int contiguous = 1;
  for (int g = 0; g < groups; g++) {
    need(out[g]->l == key->l && trgsw_dft_is_block(out[g]), "multivalue_bootstrap_UBR_phase1: `out` must hold n / unfolding samples of trgsw_alloc_new_DFT_sample[_array]");
    if (out[g]->samples[0]->a[0]->coeffs != base + (size_t)g * esz) contiguous = 0;
  }
  double *dst = contiguous ? base : (double *)mc_dev_alloc(sizeof(double) * esz * (size_t)groups);
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * ((size_t)n + 1)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * ((size_t)n + 1));
  memcpy(h, in->a, sizeof(Torus) * (size_t)n);
  h[n] = in->b;
  mc_dev_copy(d, h, sizeof(Torus) * ((size_t)n + 1), HIP_H2D);
  check_rc(mosfhet_hip_multivalue_bootstrap_UBR_phase1_batch(ectx(), (mosfhet_hip_bsk_t)key->device, dst, d, 1, NULL), "multivalue_bootstrap_UBR_phase1");
  mc_hstage_free(h);
  for (int g = 0; g < groups; g++) {
    if (!contiguous) mc_dev_copy(out[g]->samples[0]->a[0]->coeffs, dst + (size_t)g * esz, sizeof(double) * esz, HIP_D2D);
    out[g]->Bg_bit = key->Bg_bit;
  }
  if (!contiguous) hipFree(dst);
}

/* out = extract(tv X^(-b) rotated through the n / u selectors of phase 1) */
void multivalue_bootstrap_UBR_phase2(TLWE out, TRLWE tv, TLWE in, TRGSW_DFT *sa, Bootstrap_Key key, int torus_base) {
  need(key->unfolding > 1, "multivalue_bootstrap_UBR_phase2: needs a key made with unfolding > 1");
  const int n = key->n, groups = n / key->unfolding, N = key->N;
  const size_t esz = (size_t)2 * key->l * 2 * N, row = (size_t)2 * N;
  double *base = sa[0]->samples[0]->a[0]->coeffs;
  int contiguous = 1;
  for (int g = 0; g < groups; g++) {
    need(sa[g]->l == key->l && trgsw_dft_is_block(sa[g]), "multivalue_bootstrap_UBR_phase2: `sa` must hold the samples phase 1 filled");
    if (sa[g]->samples[0]->a[0]->coeffs != base + (size_t)g * esz) contiguous = 0;
  }
  double *src = base;
  if (!contiguous) {
    src = (double *)mc_dev_alloc(sizeof(double) * esz * (size_t)groups);
    for (int g = 0; g < groups; g++) mc_dev_copy(src + (size_t)g * esz, sa[g]->samples[0]->a[0]->coeffs, sizeof(double) * esz, HIP_D2D);
  }
  const size_t words = row + (size_t)n + 1 + (size_t)N + 1;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * words), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * words);
  mc_trlwe_to_flat(h, tv);
  memcpy(h + row, in->a, sizeof(Torus) * (size_t)n);
  h[row + n] = in->b;
  mc_dev_copy(d, h, sizeof(Torus) * (row + (size_t)n + 1), HIP_H2D);
  Torus *d_out = d + row + n + 1;
  check_rc(mosfhet_hip_multivalue_bootstrap_UBR_phase2_batch(ectx(), (mosfhet_hip_bsk_t)key->device, d_out, d, 1, d + row, src, 1, torus_base, NULL),
           "multivalue_bootstrap_UBR_phase2");
  mc_dev_copy(h, d_out, sizeof(Torus) * ((size_t)N + 1), HIP_D2H);
  memcpy(out->a, h, sizeof(Torus) * (size_t)N);
  out->b = h[N];
  mc_hstage_free(h);
  if (!contiguous) hipFree(src);
}

/* ================================================================== DFT-domain samples on disk (src/trlwe.c:66-85, src/trgsw.c:80-98) */
/* The reference writes the (k+1) N doubles of a sample as they sit in memory, no header -- element order is whatever its FFT back-end uses, so such a
 * file never moved between its own builds either.  Same shape here: the engine's slot order, read back by this library only. */
static void dft_poly_io(FILE *fd, DFT_Polynomial p, int write, const char *who) {
  const size_t n = (size_t)p->N;
  double *h = (double *)mc_xmalloc(sizeof(double) * n);
  mc_use_device();
  if (write) {
    mc_dev_copy(h, poly_dev(p, kNotOurs), sizeof(double) * n, HIP_D2H);
    if (fwrite(h, sizeof(double), n, fd) != n) { fprintf(stderr, "mosfhet_amd: %s: write failed\n", who); abort(); }
  } else {
    if (fread(h, sizeof(double), n, fd) != n) { fprintf(stderr, "mosfhet_amd: %s: short read\n", who); abort(); }
    mc_dev_copy(poly_dev(p, kNotOurs), h, sizeof(double) * n, HIP_H2D);
  }
  free(h);
}
void trlwe_save_DFT_sample(FILE *fd, TRLWE_DFT c) {
  for (int i = 0; i < c->k; i++) dft_poly_io(fd, c->a[i], 1, "trlwe_save_DFT_sample");
  dft_poly_io(fd, c->b, 1, "trlwe_save_DFT_sample");
}
void trlwe_load_DFT_sample(FILE *fd, TRLWE_DFT c) {
  for (int i = 0; i < c->k; i++) dft_poly_io(fd, c->a[i], 0, "trlwe_load_DFT_sample");
  dft_poly_io(fd, c->b, 0, "trlwe_load_DFT_sample");
}
TRLWE_DFT trlwe_load_new_DFT_sample(FILE *fd, int k, int N) {
  TRLWE_DFT c = trlwe_alloc_new_DFT_sample(k, N);
  trlwe_load_DFT_sample(fd, c);
  return c;
}
void trgsw_save_DFT_sample(FILE *fd, TRGSW_DFT c) {
  for (int r = 0; r < trgsw_dft_rows(c); r++) trlwe_save_DFT_sample(fd, c->samples[r]);
}
void trgsw_load_DFT_sample(FILE *fd, TRGSW_DFT out) {
  for (int r = 0; r < trgsw_dft_rows(out); r++) trlwe_load_DFT_sample(fd, out->samples[r]);
}
TRGSW_DFT trgsw_load_new_DFT_sample(FILE *fd, int l, int Bg_bit, int k, int N) {
  TRGSW_DFT g = trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N);
  trgsw_load_DFT_sample(fd, g);
  return g;
}
