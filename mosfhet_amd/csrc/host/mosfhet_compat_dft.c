/*
 * mosfhet_compat_dft.c -- the reference's DFT-level API (SURVEY.md 8(b) "must-keep" signatures) on device-resident objects.
 *
 * DFT_Polynomial / TRLWE_DFT / TRGSW_DFT keep the reference's struct shapes (include/mosfhet.h:37-40,78-81,111-114 of the reference), but every
 * `coeffs` of a DFT-domain polynomial points to DEVICE memory in the engine's slot order.  One object = one device block:
 *     TRLWE_DFT      [k+1][N/2] complex            a[0] owns the block, b is a view at + N doubles
 *     TRGSW_DFT      [(k+1) l][k+1][N/2] complex   samples[0]->a[0] owns it; the layout of one bootstrap-key entry
 *     arrays of them one block for the whole array (every element holds a reference: elements may be freed one by one, in any order, or through
 *                    the *_array functions), so an array of TRGSW_DFT is directly a key for blind_rotate
 * Every function stages its torus-domain arguments through the calling thread's staging buffers and waits for its result; nothing here computes
 * on the host.  k = 1 (every parameter set of the reference, test/tests.c:37-62).
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>

#include "compat_internal.h"

static mosfhet_hip_ctx_t ectx(void) { return (mosfhet_hip_ctx_t)mosfhet_engine_ctx(); }
static void check_rc(int rc, const char *what) {
  if (rc || mosfhet_hip_ctx_sync(ectx(), NULL)) mc_die(what);
}
static void need(int cond, const char *what) {
  if (cond) return;
  fprintf(stderr, "mosfhet_amd: %s\n", what);
  abort();
}

void init_fft(int N) { /* src/polynomial.c:341-356 builds the per-thread FFT plans; here: start the engine (twiddle tables of every ring live in the context) */
  need(N == 1024 || N == 2048 || N == 4096, "init_fft: ring degree must be 1024, 2048 or 4096");
  (void)mosfhet_engine_ctx();
}

uint16_t inverse_mod_2N(uint16_t x, uint16_t N) { /* src/misc.c:142-159: inverse of odd x modulo 2N (a power of two), here by Newton iteration */
  const uint32_t mask = 2u * N - 1;
  uint32_t inv = x;
  for (int i = 0; i < 4; i++) inv = (inv * (2u - x * inv)) & mask;
  return (uint16_t)inv;
}

/* ------------------------------------------------------------------ DFT polynomials */
DFT_Polynomial polynomial_new_DFT_polynomial(int N) {
  return (DFT_Polynomial)mc_poly_shell(MC_POLY_DFT_OWNER, mc_dev_alloc(sizeof(double) * (size_t)N), N);
}

DFT_Polynomial *polynomial_new_array_of_polynomials_DFT(int N, int size) {
  DFT_Polynomial *r = (DFT_Polynomial *)mc_xmalloc(sizeof(DFT_Polynomial) * (size_t)(size > 0 ? size : 1));
  double *block = (double *)mc_dev_alloc(sizeof(double) * (size_t)N * (size_t)(size > 0 ? size : 1));
  McShare *share = mc_share_new(block, size);
  for (int i = 0; i < size; i++) r[i] = (DFT_Polynomial)mc_poly_shell_shared(share, block + (size_t)i * N, N);
  if (size <= 0) free(share);
  return r;
}

void free_array_of_polynomials(void *p, int size) {
  if (!p) return;
  for (int i = 0; i < size; i++) free_polynomial(((void **)p)[i]);
  free(p);
}

void polynomial_torus_to_DFT(DFT_Polynomial out, TorusPolynomial in) {
  const int N = in->N;
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (size_t)N);
  mc_dev_copy(d, in->coeffs, sizeof(Torus) * (size_t)N, HIP_H2D);
  check_rc(mosfhet_hip_torus_to_dft_batch(ectx(), out->coeffs, d, N, 1, NULL), "polynomial_torus_to_DFT");
}

void polynomial_DFT_to_torus(TorusPolynomial out, const DFT_Polynomial in) {
  const int N = in->N;
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (size_t)N);
  check_rc(mosfhet_hip_dft_to_torus_batch(ectx(), d, in->coeffs, N, 1, NULL), "polynomial_DFT_to_torus");
  mc_dev_copy(out->coeffs, d, sizeof(Torus) * (size_t)N, HIP_D2H);
}

void polynomial_mul_DFT(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2) {
  check_rc(mosfhet_hip_dft_mul_batch(ectx(), out->coeffs, in1->coeffs, in2->coeffs, in1->N, 1, 0, NULL), "polynomial_mul_DFT");
}

void polynomial_mul_addto_DFT(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2) {
  check_rc(mosfhet_hip_dft_mul_batch(ectx(), out->coeffs, in1->coeffs, in2->coeffs, in1->N, 1, 1, NULL), "polynomial_mul_addto_DFT");
}

void polynomial_copy_DFT_polynomial(DFT_Polynomial out, DFT_Polynomial in) {
  mc_use_device();
  mc_dev_copy(out->coeffs, in->coeffs, sizeof(double) * (size_t)in->N, HIP_D2D);
}

/* ------------------------------------------------------------------ TRLWE_DFT */
/* owner: 1 = the object owns its block, 0 = a view; share != NULL: the object holds one reference to an array's block */
static TRLWE_DFT trlwe_dft_shell(double *base, int N, int owner, McShare *share) {
  TRLWE_DFT c = (TRLWE_DFT)mc_xmalloc(sizeof(*c));
  c->a = (DFT_Polynomial *)mc_xmalloc(sizeof(DFT_Polynomial));
  c->a[0] = share ? (DFT_Polynomial)mc_poly_shell_shared(share, base, N) : (DFT_Polynomial)mc_poly_shell(owner ? MC_POLY_DFT_OWNER : MC_POLY_DFT_VIEW, base, N);
  c->b = (DFT_Polynomial)mc_poly_shell(MC_POLY_DFT_VIEW, base + N, N);
  c->k = 1;
  return c;
}

/* device block of a TRLWE_DFT made here ([2][N/2] complex); aborts on anything else */
static double *trlwe_dft_base(TRLWE_DFT c, const char *who) {
  need(c && c->k == 1 && c->b->coeffs == c->a[0]->coeffs + c->b->N, who);
  return c->a[0]->coeffs;
}

TRLWE_DFT trlwe_alloc_new_DFT_sample(int k, int N) {
  need(k == 1, "trlwe_alloc_new_DFT_sample: k = 1 only");
  return trlwe_dft_shell((double *)mc_dev_alloc(sizeof(double) * 2 * (size_t)N), N, 1, NULL);
}

TRLWE_DFT *trlwe_alloc_new_DFT_sample_array(int count, int k, int N) {
  need(k == 1, "trlwe_alloc_new_DFT_sample_array: k = 1 only");
  TRLWE_DFT *r = (TRLWE_DFT *)mc_xmalloc(sizeof(TRLWE_DFT) * (size_t)(count > 0 ? count : 1));
  double *block = (double *)mc_dev_alloc(sizeof(double) * 2 * (size_t)N * (size_t)(count > 0 ? count : 1));
  McShare *share = mc_share_new(block, count);
  for (int i = 0; i < count; i++) r[i] = trlwe_dft_shell(block + (size_t)i * 2 * N, N, 0, share);
  if (count <= 0) free(share);
  return r;
}

void trlwe_to_DFT(TRLWE_DFT out, TRLWE in) {
  const int N = in->b->N;
  double *dst = trlwe_dft_base(out, "trlwe_to_DFT: `out` was not made by trlwe_alloc_new_DFT_sample");
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * 2 * (size_t)N), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * 2 * (size_t)N);
  mc_trlwe_to_flat(h, in);
  mc_dev_copy(d, h, sizeof(Torus) * 2 * (size_t)N, HIP_H2D);
  check_rc(mosfhet_hip_torus_to_dft_batch(ectx(), dst, d, N, 2, NULL), "trlwe_to_DFT");
  mc_hstage_free(h);
}

void trlwe_from_DFT(TRLWE out, TRLWE_DFT in) {
  const int N = out->b->N;
  const double *src = trlwe_dft_base(in, "trlwe_from_DFT: `in` was not made by trlwe_alloc_new_DFT_sample");
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * 2 * (size_t)N), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * 2 * (size_t)N);
  check_rc(mosfhet_hip_dft_to_torus_batch(ectx(), d, src, N, 2, NULL), "trlwe_from_DFT");
  mc_dev_copy(h, d, sizeof(Torus) * 2 * (size_t)N, HIP_D2H);
  mc_trlwe_from_flat(out, h);
  mc_hstage_free(h);
}

/* ------------------------------------------------------------------ TRGSW_DFT */
static TRGSW_DFT trgsw_dft_shell(double *base, int l, int Bg_bit, int N, int owner, McShare *share) {
  TRGSW_DFT g = (TRGSW_DFT)mc_xmalloc(sizeof(*g));
  g->samples = (TRLWE_DFT *)mc_xmalloc(sizeof(TRLWE_DFT) * (size_t)2 * l);
  for (int r = 0; r < 2 * l; r++) g->samples[r] = trlwe_dft_shell(base + (size_t)r * 2 * N, N, owner && r == 0, r == 0 ? share : NULL);
  g->l = l;
  g->Bg_bit = Bg_bit;
  return g;
}

static size_t trgsw_dft_doubles(int l, int N) { return (size_t)2 * l * 2 * N; }

static double *trgsw_dft_base(TRGSW_DFT g, const char *who) {
  need(g && g->samples && g->l >= 1, who);
  double *base = trlwe_dft_base(g->samples[0], who);
  const int N = g->samples[0]->b->N;
  for (int r = 1; r < 2 * g->l; r++) need(trlwe_dft_base(g->samples[r], who) == base + (size_t)r * 2 * N, who);
  return base;
}

TRGSW_DFT *mc_trgsw_dft_views(double *base, int n, int l, int Bg_bit, int N) {
  TRGSW_DFT *r = (TRGSW_DFT *)mc_xmalloc(sizeof(TRGSW_DFT) * (size_t)n);
  for (int i = 0; i < n; i++) r[i] = trgsw_dft_shell(base + (size_t)i * trgsw_dft_doubles(l, N), l, Bg_bit, N, 0, NULL);
  return r;
}

void mc_trgsw_dft_views_free(TRGSW_DFT *views, int n) {
  for (int i = 0; i < n; i++) free_trgsw(views[i]);
  free(views);
}

TRGSW_DFT trgsw_alloc_new_DFT_sample(int l, int Bg_bit, int k, int N) {
  need(k == 1, "trgsw_alloc_new_DFT_sample: k = 1 only");
  return trgsw_dft_shell((double *)mc_dev_alloc(sizeof(double) * trgsw_dft_doubles(l, N)), l, Bg_bit, N, 1, NULL);
}

TRGSW_DFT *trgsw_alloc_new_DFT_sample_array(int count, int l, int Bg_bit, int k, int N) {
  need(k == 1, "trgsw_alloc_new_DFT_sample_array: k = 1 only");
  TRGSW_DFT *r = (TRGSW_DFT *)mc_xmalloc(sizeof(TRGSW_DFT) * (size_t)(count > 0 ? count : 1));
  double *block = (double *)mc_dev_alloc(sizeof(double) * trgsw_dft_doubles(l, N) * (size_t)(count > 0 ? count : 1));
  McShare *share = mc_share_new(block, count);
  for (int i = 0; i < count; i++) r[i] = trgsw_dft_shell(block + (size_t)i * trgsw_dft_doubles(l, N), l, Bg_bit, N, 0, share);
  if (count <= 0) free(share);
  return r;
}

void trgsw_to_DFT(TRGSW_DFT out, TRGSW in) {
  const int l = in->l, N = in->samples[0]->b->N, rows = 2 * l;
  need(out->l == l, "trgsw_to_DFT: gadget sizes differ");
  double *dst = trgsw_dft_base(out, "trgsw_to_DFT: `out` was not made by trgsw_alloc_new_DFT_sample");
  const size_t words = (size_t)rows * 2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * words), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * words);
  for (int r = 0; r < rows; r++) mc_trlwe_to_flat(h + (size_t)r * 2 * N, in->samples[r]);
  mc_dev_copy(d, h, sizeof(Torus) * words, HIP_H2D);
  check_rc(mosfhet_hip_torus_to_dft_batch(ectx(), dst, d, N, rows * 2, NULL), "trgsw_to_DFT");
  out->Bg_bit = in->Bg_bit;
  mc_hstage_free(h);
}

void trgsw_mul_trlwe_DFT(TRLWE_DFT out, TRLWE in1, TRGSW_DFT in2) {
  const int N = in1->b->N;
  const double *g = trgsw_dft_base(in2, "trgsw_mul_trlwe_DFT: `in2` was not made by this library");
  double *dst = trlwe_dft_base(out, "trgsw_mul_trlwe_DFT: `out` was not made by trlwe_alloc_new_DFT_sample");
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * 2 * (size_t)N), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * 2 * (size_t)N);
  mc_trlwe_to_flat(h, in1);
  mc_dev_copy(d, h, sizeof(Torus) * 2 * (size_t)N, HIP_H2D);
  check_rc(mosfhet_hip_external_product_dft_batch(ectx(), g, 0, dst, d, N, in2->l, in2->Bg_bit, 1, NULL), "trgsw_mul_trlwe_DFT");
  mc_hstage_free(h);
}

/* An array of TRGSW_DFT as one device key: the array's own block when its entries are consecutive (Bootstrap_Key.s, trgsw_alloc_new_DFT_sample_array),
 * else a gathered copy (*owned is set and the caller frees it). */
static double *key_block(TRGSW_DFT *s, int size, int *l, int *Bg_bit, int *N, int *owned, const char *who) {
  need(s && size >= 1, who);
  double *base = trgsw_dft_base(s[0], who);
  *l = s[0]->l; *Bg_bit = s[0]->Bg_bit; *N = s[0]->samples[0]->b->N; *owned = 0;
  const size_t esz = trgsw_dft_doubles(*l, *N);
  int contiguous = 1;
  for (int i = 1; i < size; i++) {
    need(s[i]->l == *l && s[i]->Bg_bit == *Bg_bit && s[i]->samples[0]->b->N == *N, who);
    if (trgsw_dft_base(s[i], who) != base + (size_t)i * esz) contiguous = 0;
  }
  if (contiguous) return base;
  double *blk = (double *)mc_dev_alloc(sizeof(double) * esz * (size_t)size);
  for (int i = 0; i < size; i++) mc_dev_copy(blk + (size_t)i * esz, trgsw_dft_base(s[i], who), sizeof(double) * esz, HIP_D2D);
  *owned = 1;
  return blk;
}

/* src/bootstrap.c:107-122: tv <- tv * X^{sum a_i s_i} by `size` CMUX steps with the selectors s[0..size) */
void blind_rotate(TRLWE tv, Torus *a, TRGSW_DFT *s, int size) {
  int l, Bg_bit, N, owned;
  double *blk = key_block(s, size, &l, &Bg_bit, &N, &owned, "blind_rotate: `s` must be TRGSW_DFT samples made by this library (one ring, one gadget)");
  need(tv->b->N == N, "blind_rotate: ring degrees of tv and s differ");
  mosfhet_hip_bsk_t view = NULL;
  if (mosfhet_hip_bsk_view_create(ectx(), &view, blk, size, 1, N, l, Bg_bit)) mc_die("blind_rotate");
  const size_t acc_w = (size_t)2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (acc_w + size + 1)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (acc_w + size + 1));
  mc_trlwe_to_flat(h, tv);
  memcpy(h + acc_w, a, sizeof(Torus) * (size_t)size);
  h[acc_w + size] = 0;
  mc_dev_copy(d, h, sizeof(Torus) * (acc_w + size + 1), HIP_H2D);
  check_rc(mosfhet_hip_blind_rotate_batch(ectx(), view, d, d + acc_w, 1, NULL), "blind_rotate");
  mc_dev_copy(h, d, sizeof(Torus) * acc_w, HIP_D2H);
  mc_trlwe_from_flat(tv, h);
  mc_hstage_free(h);
  mosfhet_hip_bsk_destroy(view);
  if (owned) hipFree(blk);
}

/* src/bootstrap_ga.c:35-60.  `ak` must be the automorphism key set of a Bootstrap_GA_Key (entry j <-> generator 2j + 1, one device key set). */
void blind_rotate_ga(TRLWE tv, Torus *a, TRGSW_DFT *s, TRLWE_KS_Key *ak, int size) {
  int l, Bg_bit, N, owned;
  double *blk = key_block(s, size, &l, &Bg_bit, &N, &owned, "blind_rotate_ga: `s` must be TRGSW_DFT samples made by this library (one ring, one gadget)");
  need(ak && ak[0] && ak[0]->device && ak[0]->entry == 0, "blind_rotate_ga: `ak` must be the .ak of a Bootstrap_GA_Key");
  mosfhet_hip_bsk_t view = NULL;
  if (mosfhet_hip_bsk_view_create(ectx(), &view, blk, size, 1, N, l, Bg_bit)) mc_die("blind_rotate_ga");
  const size_t acc_w = (size_t)2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (acc_w + size + 1)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (acc_w + size + 1));
  mc_trlwe_to_flat(h, tv);
  memcpy(h + acc_w, a, sizeof(Torus) * (size_t)size);
  h[acc_w + size] = 0;
  mc_dev_copy(d, h, sizeof(Torus) * (acc_w + size + 1), HIP_H2D);
  check_rc(mosfhet_hip_blind_rotate_ga_batch(ectx(), view, (mosfhet_hip_gak_t)ak[0]->device, d, d + acc_w, 1, NULL), "blind_rotate_ga");
  mc_dev_copy(h, d, sizeof(Torus) * acc_w, HIP_D2H);
  mc_trlwe_from_flat(tv, h);
  mc_hstage_free(h);
  mosfhet_hip_bsk_destroy(view);
  if (owned) hipFree(blk);
}

/* src/trlwe.c:775-781: out = KeySwitch_{ks_key}(in(X^gen)); ks_key switches from key(X^gen) back to key (any entry of a key set) */
void trlwe_eval_automorphism(TRLWE out, TRLWE in, uint64_t gen, TRLWE_KS_Key ks_key) {
  const int N = in->b->N;
  const size_t row = (size_t)2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * 2 * row), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * 2 * row);
  mc_trlwe_to_flat(h, in);
  mc_dev_copy(d, h, sizeof(Torus) * row, HIP_H2D);
  check_rc(mosfhet_hip_trlwe_eval_automorphism_entry_batch(ectx(), (mosfhet_hip_gak_t)ks_key->device, ks_key->entry, d + row, d, (int)(gen & (uint64_t)(2 * N - 1)), 1, NULL),
           "trlwe_eval_automorphism");
  mc_dev_copy(h + row, d + row, sizeof(Torus) * row, HIP_D2H);
  mc_trlwe_from_flat(out, h + row);
  mc_hstage_free(h);
}

/* src/bootstrap.c:369-389: out = (0, p0) + sum_i selector[i] (.) digit_i(p1 - p0); the l selector rows are TRLWE_DFT (trlwe_to_DFT of the packed samples) */
void public_mux(TRLWE out, TorusPolynomial p0, TorusPolynomial p1, TRLWE_DFT *selector, int l, int Bg_bit) {
  const int N = out->b->N;
  const size_t row = (size_t)2 * N;
  const char *who = "public_mux: `selector` must be TRLWE_DFT samples made by this library";
  double *sel = trlwe_dft_base(selector[0], who);
  int owned = 0;
  for (int i = 1; i < l; i++)
    if (trlwe_dft_base(selector[i], who) != sel + (size_t)i * row) owned = 1;
  if (owned) {   /* selector rows allocated one by one: gather them */
    sel = (double *)mc_dev_alloc(sizeof(double) * row * (size_t)l);
    for (int i = 0; i < l; i++) mc_dev_copy(sel + (size_t)i * row, trlwe_dft_base(selector[i], who), sizeof(double) * row, HIP_D2D);
  }
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (2 * (size_t)N + row)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (2 * (size_t)N + row));
  memcpy(h, p0->coeffs, sizeof(Torus) * (size_t)N);
  memcpy(h + N, p1->coeffs, sizeof(Torus) * (size_t)N);
  mc_dev_copy(d, h, sizeof(Torus) * 2 * (size_t)N, HIP_H2D);
  check_rc(mosfhet_hip_public_mux_dft_batch(ectx(), d + 2 * N, d, d + N, sel, N, l, Bg_bit, 1, NULL), "public_mux");
  mc_dev_copy(h + 2 * N, d + 2 * N, sizeof(Torus) * row, HIP_D2H);
  mc_trlwe_from_flat(out, h + 2 * N);
  mc_hstage_free(h);
  if (owned) hipFree(sel);
}

/* ------------------------------------------------------------------ TRGSW-accumulator bootstrap (src/bootstrap.c:267-306) */
void functional_bootstrap_trgsw_phase1(TRGSW_DFT out, TLWE in, Bootstrap_Key key, int torus_base) {
  const int n = key->n;
  double *dst = trgsw_dft_base(out, "functional_bootstrap_trgsw_phase1: `out` was not made by trgsw_alloc_new_DFT_sample");
  need(out->l == key->l && out->samples[0]->b->N == key->N, "functional_bootstrap_trgsw_phase1: `out` and the key differ in ring or gadget");
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * ((size_t)n + 1)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * ((size_t)n + 1));
  memcpy(h, in->a, sizeof(Torus) * (size_t)n);
  h[n] = in->b;
  mc_dev_copy(d, h, sizeof(Torus) * ((size_t)n + 1), HIP_H2D);
  check_rc(mosfhet_hip_functional_bootstrap_trgsw_phase1_batch(ectx(), (mosfhet_hip_bsk_t)key->device, dst, d, 1, torus_base, NULL), "functional_bootstrap_trgsw_phase1");
  out->Bg_bit = key->Bg_bit;
  mc_hstage_free(h);
}

void functional_bootstrap_trgsw_phase2(TLWE out, TRGSW_DFT in, TRLWE tv) {
  const int N = tv->b->N;
  const size_t row = (size_t)2 * N;
  const double *g = trgsw_dft_base(in, "functional_bootstrap_trgsw_phase2: `in` was not made by this library");
  mosfhet_hip_bsk_t view = NULL;   /* the selector is its own one-entry key: ring and gadget come from the sample, as in the reference (src/bootstrap.c:297-306) */
  if (mosfhet_hip_bsk_view_create(ectx(), &view, g, 1, 1, N, in->l, in->Bg_bit)) mc_die("functional_bootstrap_trgsw_phase2");
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (row + N + 1)), *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (row + N + 1));
  mc_trlwe_to_flat(h, tv);
  mc_dev_copy(d, h, sizeof(Torus) * row, HIP_H2D);
  check_rc(mosfhet_hip_functional_bootstrap_trgsw_phase2_batch(ectx(), view, d + row, g, d, 1, 1, NULL), "functional_bootstrap_trgsw_phase2");
  mc_dev_copy(h + row, d + row, sizeof(Torus) * ((size_t)N + 1), HIP_D2H);
  memcpy(out->a, h + row, sizeof(Torus) * (size_t)N);
  out->b = h[row + N];
  mc_hstage_free(h);
  mosfhet_hip_bsk_destroy(view);
}

/* ------------------------------------------------------------------ arrays and convenience constructors (src/trlwe.c:15-21,96-102,318-322, src/trgsw.c:82-88,137-143) */
TRLWE *trlwe_alloc_new_sample_array(int count, int k, int N) {
  TRLWE *r = (TRLWE *)mc_xmalloc(sizeof(TRLWE) * (size_t)(count > 0 ? count : 1));
  for (int i = 0; i < count; i++) r[i] = trlwe_alloc_new_sample(k, N);
  return r;
}

void free_trlwe_array(void *p, int count) {
  if (!p) return;
  for (int i = 0; i < count; i++) free_trlwe(((void **)p)[i]);
  free(p);
}

TRLWE trlwe_new_sample(TorusPolynomial m, TRLWE_Key key) {
  TRLWE c = trlwe_alloc_new_sample(key->k, key->s[0]->N);
  trlwe_sample(c, m, key);
  return c;
}

TRGSW *trgsw_alloc_new_sample_array(int count, int l, int Bg_bit, int k, int N) {
  TRGSW *r = (TRGSW *)mc_xmalloc(sizeof(TRGSW) * (size_t)(count > 0 ? count : 1));
  for (int i = 0; i < count; i++) r[i] = trgsw_alloc_new_sample(l, Bg_bit, k, N);
  return r;
}

void free_trgsw_array(void *p, int count) {
  if (!p) return;
  for (int i = 0; i < count; i++) free_trgsw(((void **)p)[i]);
  free(p);
}
