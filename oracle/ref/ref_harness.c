/*
 * ref_harness.c -- flat-buffer shim over the REAL reference library (antoniocgj/MOSFHET).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is our own code; it is compiled TOGETHER WITH the
 * reference's sources where they lie under /root/reference (oracle/ref/Makefile) into
 * oracle/_ref/libmosfhet_ref_{avx512,ffnt}.so.  Nothing from the reference is copied into the
 * repository.  It only marshals flat arrays (layouts of oracle/mosfhet_oracle.h) into the
 * reference's pointer-based structs (include/mosfhet.h:32-133) and calls its public functions,
 * so tests can pin the oracle against the reference and bench.py can time the reference's own
 * bootstrap as the CPU baseline (cpu_baseline.kind = "reference").
 */
#include <mosfhet.h>
#include <sys/time.h>

/* ---------- marshalling helpers ---------- */
static TRLWE trlwe_from_flat(const Torus *flat, int k, int N) {
  TRLWE c = trlwe_alloc_new_sample(k, N);
  for (int p = 0; p < k; p++) memcpy(c->a[p]->coeffs, flat + (size_t)p * N, sizeof(Torus) * N);
  memcpy(c->b->coeffs, flat + (size_t)k * N, sizeof(Torus) * N);
  return c;
}

static void trlwe_to_flat(Torus *flat, TRLWE c, int N) {
  for (int p = 0; p < c->k; p++) memcpy(flat + (size_t)p * N, c->a[p]->coeffs, sizeof(Torus) * N);
  memcpy(flat + (size_t)c->k * N, c->b->coeffs, sizeof(Torus) * N);
}

static TLWE tlwe_from_flat(const Torus *flat, int n) {
  TLWE c = tlwe_alloc_sample(n);
  memcpy(c->a, flat, sizeof(Torus) * n);
  c->b = flat[n];
  return c;
}

static void tlwe_to_flat(Torus *flat, TLWE c) {
  memcpy(flat, c->a, sizeof(Torus) * c->n);
  flat[c->n] = c->b;
}

static TorusPolynomial poly_from_flat(const Torus *flat, int N) {
  TorusPolynomial p = polynomial_new_torus_polynomial(N);
  memcpy(p->coeffs, flat, sizeof(Torus) * N);
  return p;
}

/* ---------- scalars ---------- */
uint64_t ref_torus2int(uint64_t x, int log_scale) { return torus2int(x, log_scale); }
uint64_t ref_double2torus(double x) { return double2torus(x); }
const char *ref_backend(void) {
#if defined(USE_SPQLIOS) && defined(AVX512_OPT)
  return "spqlios_avx512";
#elif defined(USE_SPQLIOS)
  return "spqlios_fma";
#else
  return "ffnt";
#endif
}
void ref_init(int N) { init_fft(N); }

/* ---------- integer polynomial ops ---------- */
void ref_poly_decompose_i(Torus *out, const Torus *in, int N, int Bg_bit, int l, int i) {
  TorusPolynomial pi = poly_from_flat(in, N), po = polynomial_new_torus_polynomial(N);
  polynomial_decompose_i(po, pi, Bg_bit, l, i);
  memcpy(out, po->coeffs, sizeof(Torus) * N);
  free_polynomial(pi);
  free_polynomial(po);
}

void ref_poly_decompose(Torus *out, const Torus *in, int N, int Bg_bit, int l) {
  TorusPolynomial pi = poly_from_flat(in, N);
  TorusPolynomial *po = polynomial_new_array_of_torus_polynomials(N, l);
  polynomial_decompose(po, pi, Bg_bit, l);
  for (int i = 0; i < l; i++) memcpy(out + (size_t)i * N, po[i]->coeffs, sizeof(Torus) * N);
  free_polynomial(pi);
  free_array_of_polynomials((void *)po, l);
}

/* which: 0 = mul_by_xai, 1 = mul_by_xai_addto (out preloaded), 2 = mul_by_xai_minus_1 */
void ref_poly_mul_by_xai(Torus *out, const Torus *in, int N, int a, int which) {
  TorusPolynomial pi = poly_from_flat(in, N), po = poly_from_flat(out, N);
  if (which == 0) torus_polynomial_mul_by_xai(po, pi, a);
  else if (which == 1) torus_polynomial_mul_by_xai_addto(po, pi, a);
  else torus_polynomial_mul_by_xai_minus_1(po, pi, a);
  memcpy(out, po->coeffs, sizeof(Torus) * N);
  free_polynomial(pi);
  free_polynomial(po);
}

void ref_poly_permute(Torus *out, const Torus *in, int N, uint64_t gen) {
  TorusPolynomial pi = poly_from_flat(in, N), po = polynomial_new_torus_polynomial(N);
  polynomial_permute(po, pi, gen);
  memcpy(out, po->coeffs, sizeof(Torus) * N);
  free_polynomial(pi);
  free_polynomial(po);
}

void ref_poly_naive_mul(Torus *out, const Torus *a, const Torus *b, int N) {
  TorusPolynomial pa = poly_from_flat(a, N), pb = poly_from_flat(b, N), po = polynomial_new_torus_polynomial(N);
  polynomial_naive_mul_torus(po, pa, pb);
  memcpy(out, po->coeffs, sizeof(Torus) * N);
  free_polynomial(pa);
  free_polynomial(pb);
  free_polynomial(po);
}

/* FFT product: polynomial_mul_torus (src/polynomial.c:276) */
void ref_poly_mul_fft(Torus *out, const Torus *a, const Torus *b, int N) {
  TorusPolynomial pa = poly_from_flat(a, N), pb = poly_from_flat(b, N), po = polynomial_new_torus_polynomial(N);
  polynomial_mul_torus(po, pa, pb);
  memcpy(out, po->coeffs, sizeof(Torus) * N);
  free_polynomial(pa);
  free_polynomial(pb);
  free_polynomial(po);
}

/* Torus -> DFT -> Torus round trip (test_poly_DFT, test/tests.c:231-242) */
void ref_poly_dft_roundtrip(Torus *out, const Torus *in, int N) {
  TorusPolynomial pi = poly_from_flat(in, N), po = polynomial_new_torus_polynomial(N);
  DFT_Polynomial d = polynomial_new_DFT_polynomial(N);
  polynomial_torus_to_DFT(d, pi);
  polynomial_DFT_to_torus(po, d);
  memcpy(out, po->coeffs, sizeof(Torus) * N);
  free_polynomial(pi);
  free_polynomial(po);
  free_DFT_polynomial(d);
}

void ref_trlwe_extract_tlwe(Torus *out, const Torus *in, int k, int N, int idx) {
  TRLWE c = trlwe_from_flat(in, k, N);
  TLWE o = tlwe_alloc_sample(k * N);
  trlwe_extract_tlwe(o, c, idx);
  tlwe_to_flat(out, o);
  free_trlwe(c);
  free_tlwe(o);
}

void ref_trlwe_torus_packing(Torus *out, Torus *lut, int k, int N, int size) {
  TRLWE c = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(c, lut, size);
  trlwe_to_flat(out, c, N);
  free_trlwe(c);
}

Torus ref_tlwe_phase(const Torus *c, Torus *s, int n) {
  TLWE ct = tlwe_from_flat(c, n);
  struct _TLWE_Key key = {.s = s, .n = n, .sigma = 0};
  Torus r = tlwe_phase(ct, &key);
  free_tlwe(ct);
  return r;
}

/* ---------- TRGSW / bootstrap key ---------- */
static TRGSW_DFT trgsw_dft_from_flat(const Torus *flat, int k, int N, int l, int Bg_bit) {
  TRGSW g = trgsw_alloc_new_sample(l, Bg_bit, k, N);
  const size_t row = (size_t)(k + 1) * N;
  for (int q = 0; q < (k + 1) * l; q++) {
    for (int p = 0; p < k; p++) memcpy(g->samples[q]->a[p]->coeffs, flat + q * row + (size_t)p * N, sizeof(Torus) * N);
    memcpy(g->samples[q]->b->coeffs, flat + q * row + (size_t)k * N, sizeof(Torus) * N);
  }
  TRGSW_DFT gd = trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N);
  trgsw_to_DFT(gd, g);
  free_trgsw(g);
  return gd;
}

/* out = TRGSW (.) in, back in the torus domain: trgsw_mul_trlwe_DFT + trlwe_from_DFT */
void ref_external_product(Torus *out, const Torus *in, const Torus *trgsw_flat, int k, int N, int l, int Bg_bit) {
  TRGSW_DFT gd = trgsw_dft_from_flat(trgsw_flat, k, N, l, Bg_bit);
  TRLWE ci = trlwe_from_flat(in, k, N), co = trlwe_alloc_new_sample(k, N);
  TRLWE_DFT tmp = trlwe_alloc_new_DFT_sample(k, N);
  trgsw_mul_trlwe_DFT(tmp, ci, gd);
  trlwe_from_DFT(co, tmp);
  trlwe_to_flat(out, co, N);
  free_trlwe(ci);
  free_trlwe(co);
  free_trlwe(tmp);
  free_trgsw(gd);
}

/* Bootstrap_Key from torus-domain rows u64[n][(k+1)l][k+1][N] (what new_bootstrap_key builds at
 * src/bootstrap.c:14-18, minus the non-reproducible encryption) */
void *ref_bk_new(const Torus *bk_flat, int n, int k, int N, int l, int Bg_bit) {
  Bootstrap_Key res = (Bootstrap_Key)safe_malloc(sizeof(*res));
  res->s = (TRGSW_DFT *)safe_malloc(sizeof(TRGSW_DFT) * n);
  res->su = NULL;
  res->n = n;
  res->k = k;
  res->l = l;
  res->N = N;
  res->Bg_bit = Bg_bit;
  res->unfolding = 1;
  const size_t sz = (size_t)(k + 1) * l * (k + 1) * N;
  for (int i = 0; i < n; i++) res->s[i] = trgsw_dft_from_flat(bk_flat + i * sz, k, N, l, Bg_bit);
  return res;
}

void ref_bk_free(void *h) { free_bootstrap_key((Bootstrap_Key)h); }

void ref_blind_rotate(Torus *acc, const Torus *a, void *h) {
  Bootstrap_Key bk = (Bootstrap_Key)h;
  TRLWE c = trlwe_from_flat(acc, bk->k, bk->N);
  blind_rotate(c, (Torus *)a, bk->s, bk->n);
  trlwe_to_flat(acc, c, bk->N);
  free_trlwe(c);
}

void ref_functional_bootstrap_wo_extract(Torus *out, const Torus *tv, const Torus *in, void *h, int torus_base) {
  Bootstrap_Key bk = (Bootstrap_Key)h;
  TRLWE t = trlwe_from_flat(tv, bk->k, bk->N), o = trlwe_alloc_new_sample(bk->k, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n);
  functional_bootstrap_wo_extract(o, t, c, bk, torus_base);
  trlwe_to_flat(out, o, bk->N);
  free_trlwe(t);
  free_trlwe(o);
  free_tlwe(c);
}

void ref_functional_bootstrap(Torus *out, const Torus *tv, const Torus *in, void *h, int torus_base) {
  Bootstrap_Key bk = (Bootstrap_Key)h;
  TRLWE t = trlwe_from_flat(tv, bk->k, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->k * bk->N);
  functional_bootstrap(o, t, c, bk, torus_base);
  tlwe_to_flat(out, o);
  free_trlwe(t);
  free_tlwe(c);
  free_tlwe(o);
}

void ref_programmable_bootstrap(Torus *out, const Torus *tv, const Torus *in, void *h, int precision, int kappa, int theta) {
  Bootstrap_Key bk = (Bootstrap_Key)h;
  TRLWE t = trlwe_from_flat(tv, bk->k, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->k * bk->N);
  programmable_bootstrap(o, t, c, bk, precision, kappa, theta);
  tlwe_to_flat(out, o);
  free_trlwe(t);
  free_tlwe(c);
  free_tlwe(o);
}

/* ---------- LWE key switch ---------- */
void *ref_ksk_new(const Torus *flat, int n_in, int n_out, int t, int base_bit) {
  const int base = 1 << base_bit;
  TLWE_KS_Key res = (TLWE_KS_Key)safe_malloc(sizeof(*res));
  res->base_bit = base_bit;
  res->t = t;
  res->n = n_in;
  res->s = (TLWE ***)safe_malloc(sizeof(TLWE **) * n_in);
  const size_t row = (size_t)n_out + 1;
  for (int i = 0; i < n_in; i++) {
    res->s[i] = (TLWE **)safe_malloc(sizeof(TLWE *) * t);
    for (int j = 0; j < t; j++) {
      res->s[i][j] = (TLWE *)safe_malloc(sizeof(TLWE) * (base - 1));
      for (int v = 0; v < base - 1; v++)
        res->s[i][j][v] = tlwe_from_flat(flat + (((size_t)i * t + j) * (base - 1) + v) * row, n_out);
    }
  }
  return res;
}

void ref_ksk_free(void *h) { free_tlwe_ks_key((TLWE_KS_Key)h); }

void ref_tlwe_keyswitch(Torus *out, const Torus *in, void *h, int n_out) {
  TLWE_KS_Key ks = (TLWE_KS_Key)h;
  TLWE c = tlwe_from_flat(in, ks->n), o = tlwe_alloc_sample(n_out);
  tlwe_keyswitch(o, c, ks);
  tlwe_to_flat(out, o);
  free_tlwe(c);
  free_tlwe(o);
}

/* ---------- FDFB / multi-value (src/bootstrap.c:519-538, 222-230) ---------- */
void ref_full_domain_functional_bootstrap(Torus *out, const Torus *tv, const Torus *in, void *bkh, void *kskh, int precision) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TRLWE t = trlwe_from_flat(tv, bk->k, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->k * bk->N);
  full_domain_functional_bootstrap(o, t, c, bk, (TLWE_KS_Key)kskh, precision);
  tlwe_to_flat(out, o);
  free_trlwe(t);
  free_tlwe(c);
  free_tlwe(o);
}

void ref_multivalue_bootstrap_CLOT21(Torus *out, const Torus *tv, const Torus *in, void *bkh, int torus_base, int n_luts) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TRLWE t = trlwe_from_flat(tv, bk->k, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n);
  TLWE *o = tlwe_alloc_sample_array(n_luts, bk->k * bk->N);
  multivalue_bootstrap_CLOT21(o, t, c, bk, torus_base, n_luts);
  for (int i = 0; i < n_luts; i++) tlwe_to_flat(out + (size_t)i * (bk->k * bk->N + 1), o[i]);
  free_trlwe(t);
  free_tlwe(c);
  free_tlwe_array(o, n_luts);
}

void ref_trlwe_torus_packing_many_LUT(Torus *out, Torus *lut, int k, int N, int lut_size, int n_luts) {
  TRLWE c = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing_many_LUT(c, lut, lut_size, n_luts);
  trlwe_to_flat(out, c, N);
  free_trlwe(c);
}

/* ---------- FFT-based TRLWE key switch, automorphisms, GA bootstrap (src/keyswitch.c:162-193, src/trlwe.c:775-781,
 * src/bootstrap_ga.c) ---------- */
static TRLWE_KS_Key trlwe_ks_from_flat(const Torus *flat /*[t][k+1][N]*/, int k, int N, int t, int base_bit) {
  TRLWE_KS_Key res = (TRLWE_KS_Key)safe_malloc(sizeof(*res));
  res->base_bit = base_bit;
  res->k = k;
  res->t = t;
  res->s = (TRLWE_DFT **)safe_malloc(sizeof(TRLWE_DFT *) * k);
  res->s[0] = (TRLWE_DFT *)safe_malloc(sizeof(TRLWE_DFT) * t);
  for (int j = 0; j < t; j++) {
    TRLWE tmp = trlwe_from_flat(flat + (size_t)j * (k + 1) * N, k, N);
    res->s[0][j] = trlwe_alloc_new_DFT_sample(k, N);
    trlwe_to_DFT(res->s[0][j], tmp);
    free_trlwe(tmp);
  }
  return res;
}

void ref_trlwe_keyswitch(Torus *out, const Torus *in, const Torus *ks_flat, int N, int t, int base_bit) {
  TRLWE_KS_Key ks = trlwe_ks_from_flat(ks_flat, 1, N, t, base_bit);
  TRLWE c = trlwe_from_flat(in, 1, N), o = trlwe_alloc_new_sample(1, N);
  trlwe_keyswitch(o, c, ks);
  trlwe_to_flat(out, o, N);
  free_trlwe(c);
  free_trlwe(o);
  free_trlwe_ks_key(ks);
}

void ref_trlwe_eval_automorphism(Torus *out, const Torus *in, uint64_t gen, const Torus *ks_flat, int N, int t, int base_bit) {
  TRLWE_KS_Key ks = trlwe_ks_from_flat(ks_flat, 1, N, t, base_bit);
  TRLWE c = trlwe_from_flat(in, 1, N), o = trlwe_alloc_new_sample(1, N);
  trlwe_eval_automorphism(o, c, gen, ks);
  trlwe_to_flat(out, o, N);
  free_trlwe(c);
  free_trlwe(o);
  free_trlwe_ks_key(ks);
}

uint32_t ref_inverse_mod_2N(uint32_t x, int N) { return inverse_mod_2N((uint16_t)x, (uint16_t)N); }

/* Bootstrap_GA_Key from torus-domain TRGSW(X^{s_i}) rows and the automorphism key set ak_flat[N][l][2][N]
 * (what new_bootstrap_key_ga builds, src/bootstrap_ga.c:5-24, minus the non-reproducible encryption) */
void *ref_bk_ga_new(const Torus *bk_flat, const Torus *ak_flat, int n, int N, int l, int Bg_bit) {
  Bootstrap_GA_Key res = (Bootstrap_GA_Key)safe_malloc(sizeof(*res));
  res->s = (TRGSW_DFT *)safe_malloc(sizeof(TRGSW_DFT) * n);
  res->su = NULL;
  res->n = n; res->k = 1; res->l = l; res->N = N; res->Bg_bit = Bg_bit; res->unfolding = 1;
  const size_t sz = (size_t)2 * l * 2 * N;
  for (int i = 0; i < n; i++) res->s[i] = trgsw_dft_from_flat(bk_flat + i * sz, 1, N, l, Bg_bit);
  res->ak = (TRLWE_KS_Key *)safe_malloc(sizeof(TRLWE_KS_Key) * N);
  for (int j = 0; j < N; j++) res->ak[j] = trlwe_ks_from_flat(ak_flat + (size_t)j * l * 2 * N, 1, N, l, Bg_bit);
  return res;
}

void ref_bk_ga_free(void *h) { free_bootstrap_key_ga((Bootstrap_GA_Key)h); }

void ref_functional_bootstrap_ga(Torus *out, const Torus *tv, const Torus *in, void *h, int torus_base, int extract) {
  Bootstrap_GA_Key bk = (Bootstrap_GA_Key)h;
  TRLWE t = trlwe_from_flat(tv, 1, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n);
  if (extract) {
    TLWE o = tlwe_alloc_sample(bk->N);
    functional_bootstrap_ga(o, t, c, bk, torus_base);
    tlwe_to_flat(out, o);
    free_tlwe(o);
  } else {
    TRLWE o = trlwe_alloc_new_sample(1, bk->N);
    functional_bootstrap_wo_extract_ga(o, t, c, bk, torus_base);
    trlwe_to_flat(out, o, bk->N);
    free_trlwe(o);
  }
  free_trlwe(t);
  free_tlwe(c);
}

/* ---------- circuit bootstrap (src/bootstrap.c:346-366, src/keyswitch.c:52-63,458-475) ---------- */
/* Generic_KS_Key from flat rows [n][t][2^bb-1][2][N]; built with plain (uncompressed) TRLWE rows -- the library is
 * compiled with USE_COMPRESSED_TRLWE, whose _MACRO_trlwe_subto expects seed-compressed rows, so the table-lookup
 * loop of trlwe_packing1_keyswitch (keyswitch.c:458-475) is restated here on plain rows with the library's trlwe_subto. */
void ref_trlwe_packing1_keyswitch(Torus *out, const Torus *in, const Torus *ksk, int n, int N, int t, int base_bit) {
  const int bit_size = sizeof(Torus) * 8;
  const Torus prec_offset = 1UL << (bit_size - (1 + base_bit * t));
  const Torus mask = (1UL << base_bit) - 1;
  const size_t row = (size_t)2 * N, per_j = (1UL << base_bit) - 1;
  TRLWE o = trlwe_alloc_new_sample(1, N), r = trlwe_alloc_new_sample(1, N);
  trlwe_noiseless_trivial_sample(o, NULL);
  o->b->coeffs[0] = in[n];
  for (int i = 0; i < n; i++) {
    const Torus aibar = in[i] + prec_offset;
    for (int j = 0; j < t; j++) {
      const Torus aij = (aibar >> (bit_size - (j + 1) * base_bit)) & mask;
      if (aij != 0) {
        const Torus *src = ksk + (((size_t)i * t + j) * per_j + (aij - 1)) * row;
        memcpy(r->a[0]->coeffs, src, sizeof(Torus) * N);
        memcpy(r->b->coeffs, src + N, sizeof(Torus) * N);
        trlwe_subto(o, r);
      }
    }
  }
  trlwe_to_flat(out, o, N);
  free_trlwe(o);
  free_trlwe(r);
}

void ref_trlwe_priv_keyswitch_2(Torus *out, const Torus *in, const Torus *ks0_flat, const Torus *ks1_flat, int N, int t, int base_bit) {
  TRLWE_KS_Key ks[2] = {trlwe_ks_from_flat(ks0_flat, 1, N, t, base_bit), trlwe_ks_from_flat(ks1_flat, 1, N, t, base_bit)};
  TRLWE c = trlwe_from_flat(in, 1, N), o = trlwe_alloc_new_sample(1, N);
  trlwe_priv_keyswitch_2(o, c, ks);
  trlwe_to_flat(out, o, N);
  free_trlwe(c);
  free_trlwe(o);
  free_trlwe_ks_key(ks[0]);
  free_trlwe_ks_key(ks[1]);
}

/* ---------- callers either side of the bootstrap ----------
 * Table-lookup TRLWE keys are made by the REFERENCE's own key generation from secret words we pass in, and exported to flat
 * rows for the oracle: this library is built with USE_COMPRESSED_TRLWE, so a row is a seed + b; trlwe_compressed_subto
 * (src/trlwe_compressed*.c) regenerates the mask, and 0 - (0 - row) is the row. */
static TRLWE_Key trlwe_key_from_words(const Torus *s, int N, double sigma) {
  TRLWE_Key key = trlwe_alloc_key(N, 1, sigma);
  memcpy(key->s[0]->coeffs, s, sizeof(Torus) * N);
  polynomial_torus_to_DFT(key->s_dft[0], key->s[0]);
  return key;
}

static TLWE_Key tlwe_key_from_words(const Torus *s, int n, double sigma) {
  TLWE_Key key = tlwe_alloc_key(n, sigma);
  memcpy(key->s, s, sizeof(Torus) * n);
  return key;
}

/* kind 0: trlwe_new_packing1_KS_key (keyswitch.c:368-390); kind 1: trlwe_new_priv_SK_KS_key_N2 (keyswitch.c:611-637) */
void *ref_generic_key_new(int kind, const Torus *s_out, int N, const Torus *s_in, int n, int t, int base_bit, double sigma) {
  TRLWE_Key ko = trlwe_key_from_words(s_out, N, sigma);
  TLWE_Key ki = tlwe_key_from_words(s_in, n, sigma);
  Generic_KS_Key res = kind == 0 ? trlwe_new_packing1_KS_key(ko, ki, t, base_bit) : trlwe_new_priv_SK_KS_key_N2(ko, ki, t, base_bit);
  free_trlwe_key(ko);
  free_tlwe_key(ki);
  return res;
}

void ref_generic_key_free(void *h) { free_trlwe_generic_ks_key((Generic_KS_Key)h); }

void ref_generic_key_export(void *h, Torus *flat, int N) {
  Generic_KS_Key key = (Generic_KS_Key)h;
  const int per_j = (1 << key->base_bit) - 1, entries = key->n + (key->include_b ? 1 : 0);
  TRLWE tmp = trlwe_alloc_new_sample(1, N);
  for (int i = 0; i < entries; i++)
    for (int j = 0; j < key->t; j++)
      for (int v = 0; v < per_j; v++) {
        trlwe_noiseless_trivial_sample(tmp, NULL);
        trlwe_compressed_subto(tmp, key->s[i][j][v]);
        Torus *dst = flat + (((size_t)i * key->t + j) * per_j + v) * 2 * N;
        for (int c = 0; c < N; c++) {
          dst[c] = (Torus)0 - tmp->a[0]->coeffs[c];
          dst[N + c] = (Torus)0 - tmp->b->coeffs[c];
        }
      }
  free_trlwe(tmp);
}

/* kind 0: trlwe_packing1_keyswitch (keyswitch.c:458-475); kind 1: trlwe_priv_keyswitch (keyswitch.c:639-656) -- the library's own loops */
void ref_generic_keyswitch(int kind, Torus *out, const Torus *in, void *h, int N) {
  Generic_KS_Key key = (Generic_KS_Key)h;
  TLWE c = tlwe_from_flat(in, key->n);
  TRLWE o = trlwe_alloc_new_sample(1, N);
  if (kind == 0) trlwe_packing1_keyswitch(o, c, key);
  else trlwe_priv_keyswitch(o, c, key);
  trlwe_to_flat(out, o, N);
  free_tlwe(c);
  free_trlwe(o);
}

void ref_public_mux(Torus *out, const Torus *p0, const Torus *p1, const Torus *sel_flat /*[l][2][N]*/, int N, int l, int Bg_bit) {
  TRLWE_DFT *sel = trlwe_alloc_new_DFT_sample_array(l, 1, N);
  for (int i = 0; i < l; i++) {
    TRLWE tmp = trlwe_from_flat(sel_flat + (size_t)i * 2 * N, 1, N);
    trlwe_to_DFT(sel[i], tmp);
    free_trlwe(tmp);
  }
  TorusPolynomial q0 = poly_from_flat(p0, N), q1 = poly_from_flat(p1, N);
  TRLWE o = trlwe_alloc_new_sample(1, N);
  public_mux(o, q0, q1, sel, l, Bg_bit);
  trlwe_to_flat(out, o, N);
  free_trlwe(o);
  free_polynomial(q0);
  free_polynomial(q1);
  free_trlwe_array(sel, l);
}

void ref_full_domain_functional_bootstrap_KS21(Torus *out, const Torus *tv /*[2N]*/, const Torus *in, void *bkh, void *kskh, int torus_base, int variant) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TorusPolynomial t = poly_from_flat(tv, 2 * bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->N);
  if (variant == 0) full_domain_functional_bootstrap_KS21(o, t, c, bk, (Generic_KS_Key)kskh, torus_base);
  else full_domain_functional_bootstrap_KS21_2(o, t, c, bk, (Generic_KS_Key)kskh, torus_base);
  tlwe_to_flat(out, o);
  free_polynomial(t);
  free_tlwe(c);
  free_tlwe(o);
}

void ref_multivalue_bootstrap_phase1(Torus *out /*[tb+1][2][N]*/, const Torus *in, void *bkh, int torus_base) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TRLWE *o = trlwe_alloc_new_sample_array(torus_base + 1, 1, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n);
  multivalue_bootstrap_phase1(o, c, bk, torus_base);
  for (int i = 0; i <= torus_base; i++) trlwe_to_flat(out + (size_t)i * 2 * bk->N, o[i], bk->N);
  free_tlwe(c);
  free_trlwe_array(o, torus_base + 1);
}

void ref_multivalue_bootstrap_phase2(Torus *out, int *lut_in, const Torus *rotated /*[tb+1][2][N]*/, int N, int torus_base, int log_torus_base) {
  TRLWE *r = (TRLWE *)safe_malloc(sizeof(TRLWE) * (torus_base + 1));
  for (int i = 0; i <= torus_base; i++) r[i] = trlwe_from_flat(rotated + (size_t)i * 2 * N, 1, N);
  TLWE o = tlwe_alloc_sample(N);
  multivalue_bootstrap_phase2(o, lut_in, r, torus_base, log_torus_base);
  tlwe_to_flat(out, o);
  free_tlwe(o);
  for (int i = 0; i <= torus_base; i++) free_trlwe(r[i]);
  free(r);
}

/* variant 0: circuit_bootstrap, 1: circuit_bootstrap_2 (kska = priv SK key), 3: circuit_bootstrap_3 (kska_flat = [2][ta][2][N] rows) */
void ref_circuit_bootstrap(Torus *out /*[2l][2][N]*/, const Torus *in, void *bkh, void *kskah, const Torus *kska_flat, int ta, int bba, void *kskbh,
                           int variant) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TRGSW o = trgsw_alloc_new_sample(bk->l, bk->Bg_bit, 1, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n);
  if (variant == 0) circuit_bootstrap(o, c, bk, (Generic_KS_Key)kskah, (Generic_KS_Key)kskbh);
  else if (variant == 1) circuit_bootstrap_2(o, c, bk, (Generic_KS_Key)kskah, (Generic_KS_Key)kskbh);
  else {
    TRLWE_KS_Key ks[2] = {trlwe_ks_from_flat(kska_flat, 1, bk->N, ta, bba), trlwe_ks_from_flat(kska_flat + (size_t)ta * 2 * bk->N, 1, bk->N, ta, bba)};
    circuit_bootstrap_3(o, c, bk, ks, (Generic_KS_Key)kskbh);
    free_trlwe_ks_key(ks[0]);
    free_trlwe_ks_key(ks[1]);
  }
  for (int q = 0; q < 2 * bk->l; q++) trlwe_to_flat(out + (size_t)q * 2 * bk->N, o->samples[q], bk->N);
  free_tlwe(c);
  free_trgsw(o);
}

/* functional_bootstrap_trgsw_phase1 (+ trgsw_from_DFT so the accumulator can be compared in the torus domain) and phase2 */
void ref_functional_bootstrap_trgsw(Torus *acc_out /*[2l][2][N] or NULL*/, Torus *out /*[N+1]*/, const Torus *tv, const Torus *in, void *bkh,
                                    int torus_base) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TRGSW_DFT g = trgsw_alloc_new_DFT_sample(bk->l, bk->Bg_bit, 1, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->N);
  functional_bootstrap_trgsw_phase1(g, c, bk, torus_base);
  if (acc_out) {
    TRGSW gt = trgsw_alloc_new_sample(bk->l, bk->Bg_bit, 1, bk->N);
    trgsw_from_DFT(gt, g);
    for (int q = 0; q < 2 * bk->l; q++) trlwe_to_flat(acc_out + (size_t)q * 2 * bk->N, gt->samples[q], bk->N);
    free_trgsw(gt);
  }
  TRLWE t = trlwe_from_flat(tv, 1, bk->N);
  functional_bootstrap_trgsw_phase2(o, g, t);
  tlwe_to_flat(out, o);
  free_trlwe(t);
  free_tlwe(c);
  free_tlwe(o);
  free_trgsw(g);
}

void ref_trlwe_tensor_prod_FFT(Torus *out, const Torus *in1, const Torus *in2, int precision, const Torus *rl_flat, int N, int t, int base_bit) {
  TRLWE_KS_Key rl = trlwe_ks_from_flat(rl_flat, 1, N, t, base_bit);
  TRLWE a = trlwe_from_flat(in1, 1, N), b = trlwe_from_flat(in2, 1, N), o = trlwe_alloc_new_sample(1, N);
  trlwe_tensor_prod_FFT(o, a, b, precision, rl);
  trlwe_to_flat(out, o, N);
  free_trlwe(a);
  free_trlwe(b);
  free_trlwe(o);
  free_trlwe_ks_key(rl);
}

void ref_tlwe_mul(Torus *out, const Torus *in1, const Torus *in2, int precision, void *kskh, const Torus *rl_flat, int N, int t, int base_bit) {
  TRLWE_KS_Key rl = trlwe_ks_from_flat(rl_flat, 1, N, t, base_bit);
  TLWE a = tlwe_from_flat(in1, N), b = tlwe_from_flat(in2, N), o = tlwe_alloc_sample(N);
  tlwe_mul(o, a, b, precision, (Generic_KS_Key)kskh, rl);
  tlwe_to_flat(out, o);
  free_tlwe(a);
  free_tlwe(b);
  free_tlwe(o);
  free_trlwe_ks_key(rl);
}

/* variant 0: full_domain_functional_bootstrap_CLOT21 (tv = two TRLWE test vectors [2][2][N]); 1: _CLOT21_2 (tv = 2^(precision-1) LUT words) */
void ref_full_domain_functional_bootstrap_CLOT21(Torus *out, const Torus *tv, const Torus *in, void *bkh, void *kskh, const Torus *rl_flat, int t,
                                                 int base_bit, int precision, int variant) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  TRLWE_KS_Key rl = trlwe_ks_from_flat(rl_flat, 1, bk->N, t, base_bit);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->N);
  if (variant == 0) {
    TRLWE tvs[2] = {trlwe_from_flat(tv, 1, bk->N), trlwe_from_flat(tv + (size_t)2 * bk->N, 1, bk->N)};
    full_domain_functional_bootstrap_CLOT21(o, tvs, c, bk, (Generic_KS_Key)kskh, rl, precision);
    free_trlwe(tvs[0]);
    free_trlwe(tvs[1]);
  } else {
    full_domain_functional_bootstrap_CLOT21_2(o, (Torus *)tv, c, bk, (Generic_KS_Key)kskh, rl, precision);
  }
  tlwe_to_flat(out, o);
  free_tlwe(c);
  free_tlwe(o);
  free_trlwe_ks_key(rl);
}

/* Bootstrap_Key with unfolding > 1 from torus-domain rows su[n 2^u/u][2l][2][N] (what new_bootstrap_key builds at src/bootstrap.c:23-48, minus the
 * non-reproducible encryption); ref_functional_bootstrap* then take the blind_rotate_unfolded branch (src/bootstrap.c:196-197). */
void *ref_bk_unfolded_new(const Torus *su_flat, int n, int N, int l, int Bg_bit, int unfolding) {
  Bootstrap_Key res = (Bootstrap_Key)safe_malloc(sizeof(*res));
  const int count = n * (1 << unfolding) / unfolding;
  res->s = NULL;
  res->n = n; res->k = 1; res->l = l; res->N = N; res->Bg_bit = Bg_bit; res->unfolding = unfolding;
  res->su = trgsw_alloc_new_sample_array(count, l, Bg_bit, 1, N);
  const size_t row = (size_t)2 * N, sz = (size_t)2 * l * row;
  for (int i = 0; i < count; i++)
    for (int q = 0; q < 2 * l; q++) {
      memcpy(res->su[i]->samples[q]->a[0]->coeffs, su_flat + i * sz + q * row, sizeof(Torus) * N);
      memcpy(res->su[i]->samples[q]->b->coeffs, su_flat + i * sz + q * row + N, sizeof(Torus) * N);
    }
  return res;
}

/* multivalue_bootstrap_UBR_phase1 + phase2 (src/bootstrap.c:151-190) with an unfolded key: one phase 1, then phase 2 per test vector */
void ref_multivalue_bootstrap_UBR(Torus *out /*[n_tv][N+1]*/, const Torus *tvs /*[n_tv][2][N]*/, int n_tv, const Torus *in, void *bkh, int torus_base) {
  Bootstrap_Key bk = (Bootstrap_Key)bkh;
  const int groups = bk->n / bk->unfolding;
  TRGSW_DFT *sa = (TRGSW_DFT *)safe_malloc(sizeof(TRGSW_DFT) * groups);
  for (int g = 0; g < groups; g++) sa[g] = trgsw_alloc_new_DFT_sample(bk->l, bk->Bg_bit, 1, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->N);
  multivalue_bootstrap_UBR_phase1(sa, c, bk);
  for (int i = 0; i < n_tv; i++) {
    TRLWE t = trlwe_from_flat(tvs + (size_t)i * 2 * bk->N, 1, bk->N);
    multivalue_bootstrap_UBR_phase2(o, t, c, sa, bk, torus_base);
    tlwe_to_flat(out + (size_t)i * (bk->N + 1), o);
    free_trlwe(t);
  }
  for (int g = 0; g < groups; g++) free_trgsw(sa[g]);
  free(sa);
  free_tlwe(c);
  free_tlwe(o);
}

/* trlwe_mv_extract_tlwe / _scaling / _scaling_addto / _scaling_subto (src/trlwe.c:580-622); modes as in the oracle */
void ref_trlwe_mv_extract(Torus *out, const Torus *in, int N, int mode, int amount) {
  TRLWE c = trlwe_from_flat(in, 1, N);
  if (mode == 0) {
    TLWE *o = tlwe_alloc_sample_array(amount, N);
    trlwe_mv_extract_tlwe(o, c, amount);
    for (int i = 0; i < amount; i++) tlwe_to_flat(out + (size_t)i * (N + 1), o[i]);
    free_tlwe_array(o, amount);
  } else {
    TLWE o = tlwe_from_flat(out, N);
    if (mode == 1) trlwe_mv_extract_tlwe_scaling(o, c, amount);
    else if (mode == 2) trlwe_mv_extract_tlwe_scaling_addto(o, c, amount);
    else trlwe_mv_extract_tlwe_scaling_subto(o, c, amount);
    tlwe_to_flat(out, o);
    free_tlwe(o);
  }
  free_trlwe(c);
}

/* ---------- CPU baseline: time `reps` reference programmable bootstraps on the calling thread.
 * Re-entrant across threads once ref_init(N) has run on the main thread (FFT processors are
 * __thread, src/polynomial.c:338-349). Returns elapsed seconds. ---------- */
/* ---------- on-disk formats: the reference's own writers / readers (src/tlwe.c:43-99,247-287, src/trlwe.c:24-43,230-251, src/trgsw.c:29-42) ---------- */
int ref_save_host_objects(const char *path, const Torus *lwe_s, int n, double lwe_sigma, const Torus *rlwe_s, int k, int N, double rlwe_sigma, int l,
                          int Bg_bit, const Torus *tlwe_ct, const Torus *trlwe_ct) {
  FILE *fd = fopen(path, "wb");
  if (!fd) return 1;
  TLWE_Key lk = tlwe_alloc_key(n, lwe_sigma);
  memcpy(lk->s, lwe_s, sizeof(Torus) * n);
  TRLWE_Key rk = trlwe_alloc_key(N, k, rlwe_sigma);
  for (int i = 0; i < k; i++) memcpy(rk->s[i]->coeffs, rlwe_s + (size_t)i * N, sizeof(Torus) * N);
  TRGSW_Key gk = trgsw_new_key(rk, l, Bg_bit);
  TLWE c = tlwe_from_flat(tlwe_ct, n);
  TRLWE rc = trlwe_from_flat(trlwe_ct, k, N);
  tlwe_save_key(fd, lk);
  trlwe_save_key(fd, rk);
  trgsw_save_key(fd, gk);
  tlwe_save_sample(fd, c);
  trlwe_save_sample(fd, rc);
  fclose(fd);
  free_tlwe(c);
  free_trlwe(rc);
  free_trgsw_key(gk);
  free_trlwe_key(rk);
  free_tlwe_key(lk);
  return 0;
}

int ref_ksk_save(const char *path, void *h) {
  FILE *fd = fopen(path, "wb");
  if (!fd) return 1;
  tlwe_save_KS_key(fd, (TLWE_KS_Key)h);
  fclose(fd);
  return 0;
}

void *ref_ksk_load(const char *path) {
  FILE *fd = fopen(path, "rb");
  if (!fd) return NULL;
  TLWE_KS_Key k = tlwe_load_new_KS_key(fd);
  fclose(fd);
  return k;
}

double ref_bench_programmable_bootstrap(const Torus *tv, const Torus *in, void *h, int precision, int reps) {
  Bootstrap_Key bk = (Bootstrap_Key)h;
  TRLWE t = trlwe_from_flat(tv, bk->k, bk->N);
  TLWE c = tlwe_from_flat(in, bk->n), o = tlwe_alloc_sample(bk->k * bk->N);
  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  for (int r = 0; r < reps; r++) programmable_bootstrap(o, t, c, bk, precision, 0, 0);
  gettimeofday(&t1, NULL);
  free_trlwe(t);
  free_tlwe(c);
  free_tlwe(o);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_usec - t0.tv_usec);
}
