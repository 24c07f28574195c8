/*
 * mosfhet_compat_vec.c -- the digit-parallel radix-integer callers (mosfhet_hip_vec_*, csrc/capi_vec.inc) behind host structs: M independent integers, each an
 * array of TLWE digits -- the `digits` member of the reference application's ufhe_integer (applications/multi-ciphertext-arith/include/ufhe.h:18-22) -- under
 * the keys of its ufhe_public_keyset (ufhe.h:12-16).  What the application computes one integer at a time (src/integer.c:62-264, src/lut.c:6-64, src/ml.c:4-20),
 * a MOSFHET program can ask for M integers at once; the results decrypt to what the application's loops give (tests/c/vec_callers.c on the rows of
 * tests/golden/ufhe_vectors.npz).  Marshalling only: structs <-> the digit-major device layout [digit][M][N + 1].
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "compat_internal.h"

struct _mosfhet_vec { mosfhet_hip_vec_t h; int N, torus_base, log_torus_base; };

mosfhet_vec mosfhet_vec_new(Bootstrap_Key bk, TLWE_KS_Key ks_key, LUT_Packing_KS_Key packing_key, int torus_base) {
  if (!bk || !ks_key || !packing_key || torus_base < 2 || (torus_base & (torus_base - 1))) {
    fprintf(stderr, "mosfhet_vec_new: keys must be given and torus_base a power of two\n");
    abort();
  }
  mosfhet_vec v = (mosfhet_vec)mc_xmalloc(sizeof(*v));
  v->N = bk->N;
  v->torus_base = torus_base;
  v->log_torus_base = 0;
  while ((1 << v->log_torus_base) < torus_base) v->log_torus_base++;
  mc_use_device();
  if (mosfhet_hip_vec_create((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &v->h, (mosfhet_hip_bsk_t)bk->device, (mosfhet_hip_ksk_t)ks_key->device,
                             (mosfhet_hip_ksk_t)packing_key->device, torus_base))
    mc_die("mosfhet_vec_new");
  return v;
}

void mosfhet_vec_free(mosfhet_vec v) {
  if (!v) return;
  mosfhet_hip_vec_destroy(v->h);
  free(v);
}

/* x[m][i] (integer m, digit i) -> device [d][M][N + 1] */
static Torus *upload(TLWE **x, int M, int d, int N) {
  const size_t w = (size_t)N + 1, words = (size_t)d * M * w;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * words);
  for (int i = 0; i < d; i++)
    for (int m = 0; m < M; m++) {
      const TLWE c = x[m][i];
      if (c->n != N) { fprintf(stderr, "mosfhet_vec: digit samples must have dimension N = %d (got %d)\n", N, c->n); abort(); }
      Torus *dst = h + ((size_t)i * M + m) * w;
      memcpy(dst, c->a, sizeof(Torus) * N);
      dst[N] = c->b;
    }
  Torus *dev = (Torus *)mc_dev_alloc(sizeof(Torus) * words);
  mc_dev_copy(dev, h, sizeof(Torus) * words, HIP_H2D);
  mc_hstage_free(h);
  return dev;
}

static void download(TLWE **x, const Torus *dev, int M, int d, int N) {
  const size_t w = (size_t)N + 1, words = (size_t)d * M * w;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * words);
  mc_dev_copy(h, dev, sizeof(Torus) * words, HIP_D2H);
  for (int i = 0; i < d; i++)
    for (int m = 0; m < M; m++) {
      const Torus *src = h + ((size_t)i * M + m) * w;
      memcpy(x[m][i]->a, src, sizeof(Torus) * N);
      x[m][i]->b = src[N];
    }
  mc_hstage_free(h);
}

static Torus *dev_words(int M, int d, int N) { return (Torus *)mc_dev_alloc(sizeof(Torus) * (size_t)d * M * ((size_t)N + 1)); }
static void done(mosfhet_vec v, int rc, const char *who) {
  if (rc || mosfhet_hip_ctx_sync((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), NULL)) mc_die(who);
}

static void addsub(mosfhet_vec v, TLWE **c, TLWE **a, TLWE **b, int M, int d, int subtract) {
  mc_use_device();
  Torus *da = upload(a, M, d, v->N), *db = upload(b, M, d, v->N), *dc = dev_words(M, d, v->N);
  done(v, mosfhet_hip_vec_addsub(v->h, dc, da, db, M, d, subtract, NULL), "mosfhet_vec_add / _sub");
  download(c, dc, M, d, v->N);
  hipFree(da); hipFree(db); hipFree(dc);
}
void mosfhet_vec_add_integers(mosfhet_vec v, TLWE **c, TLWE **a, TLWE **b, int M, int d) { addsub(v, c, a, b, M, d, 0); }
void mosfhet_vec_sub_integers(mosfhet_vec v, TLWE **c, TLWE **a, TLWE **b, int M, int d) { addsub(v, c, a, b, M, d, 1); }

void mosfhet_vec_relu_integers(mosfhet_vec v, TLWE **out, TLWE **in, int M, int d) {
  mc_use_device();
  Torus *di = upload(in, M, d, v->N), *dout = dev_words(M, d, v->N);
  done(v, mosfhet_hip_vec_relu(v->h, dout, di, M, d, NULL), "mosfhet_vec_relu_integers");
  download(out, dout, M, d, v->N);
  hipFree(di); hipFree(dout);
}

void mosfhet_vec_cmp_integers(mosfhet_vec v, TLWE *c, TLWE **a, TLWE **b, int M, int d, bool a_signed, bool b_signed) {
  mc_use_device();
  Torus *da = upload(a, M, d, v->N), *db = upload(b, M, d, v->N), *dc = dev_words(M, 1, v->N);
  done(v, mosfhet_hip_vec_cmp(v->h, dc, da, db, M, d, a_signed, b_signed, NULL), "mosfhet_vec_cmp_integers");
  TLWE **rows = (TLWE **)mc_xmalloc(sizeof(TLWE *) * M);   /* one digit per integer: c[m] */
  for (int m = 0; m < M; m++) rows[m] = &c[m];
  download(rows, dc, M, 1, v->N);
  free(rows);
  hipFree(da); hipFree(db); hipFree(dc);
}

void mosfhet_vec_mux_integer_arrays(mosfhet_vec v, TLWE **out, TLWE **selector, int d_sel, int size, TLWE ***vec, int M, int d) {
  mc_use_device();
  const size_t per = (size_t)d * M * ((size_t)v->N + 1);
  Torus *dt = (Torus *)mc_dev_alloc(sizeof(Torus) * per * size);   /* [size][d][M][N + 1], consumed by the tree */
  for (int e = 0; e < size; e++) {
    Torus *one = upload(vec[e], M, d, v->N);
    mc_dev_copy(dt + per * e, one, sizeof(Torus) * per, HIP_D2D);
    hipFree(one);
  }
  Torus *ds = upload(selector, M, d_sel, v->N), *dout = dev_words(M, d, v->N);
  done(v, mosfhet_hip_vec_mux_array(v->h, dout, dt, ds, size, d, M, NULL), "mosfhet_vec_mux_integer_arrays");
  download(out, dout, M, d, v->N);
  hipFree(dt); hipFree(ds); hipFree(dout);
}

void mosfhet_vec_lut_integers(mosfhet_vec v, TLWE **out, int d_out, TLWE **selector, int d_sel, uint64_t *lut, int size, int M) {
  mc_use_device();
  if (d_out * v->log_torus_base > 64) { fprintf(stderr, "mosfhet_vec_lut_integers: %d digits of %d bits do not fit a 64-bit table entry\n", d_out, v->log_torus_base); abort(); }
  (void)d_sel;
  int ds_digits = 0;
  while ((1 << (ds_digits * v->log_torus_base)) < size) ds_digits++;
  Torus *ds = upload(selector, M, ds_digits, v->N), *dout = dev_words(M, d_out, v->N);
  done(v, mosfhet_hip_vec_lut_cleartext(v->h, dout, ds, lut, size, d_out, M, NULL), "mosfhet_vec_lut_integers");
  download(out, dout, M, d_out, v->N);
  hipFree(ds); hipFree(dout);
}

void mosfhet_vec_mul_integers(mosfhet_vec v, TLWE **c, int dc, TLWE **a, int da, TLWE **b, int db, bool is_signed, int M) {
  mc_use_device();
  Torus *pa = upload(a, M, da, v->N), *pb = upload(b, M, db, v->N), *pc = dev_words(M, dc, v->N);
  done(v, mosfhet_hip_vec_mul(v->h, pc, dc, pa, da, pb, db, is_signed, M, NULL), "mosfhet_vec_mul_integers");
  download(c, pc, M, dc, v->N);
  hipFree(pa); hipFree(pb); hipFree(pc);
}

void mosfhet_vec_sl_add_integers(mosfhet_vec v, TLWE **c, int dc, TLWE **a, int da, int g, TLWE **b, int db, int h, bool is_signed, int M) {
  mc_use_device();
  Torus *pa = upload(a, M, da, v->N), *pb = upload(b, M, db, v->N), *pc = dev_words(M, dc, v->N);
  done(v, mosfhet_hip_vec_sl_add(v->h, pc, dc, pa, da, g, pb, db, h, is_signed, M, NULL), "mosfhet_vec_sl_add_integers");
  download(c, pc, M, dc, v->N);
  hipFree(pa); hipFree(pb); hipFree(pc);
}
