/*
 * mosfhet_compat.c -- host side (plain C) of the MOSFHET-compatible API, include/mosfhet_compat.h.
 *
 * Host-only work lives here: allocation, key and sample generation, phases, LUT packing, marshalling of the
 * reference's pointer-based structs into the flat batches the device layer takes.  Everything on the hot path
 * (bootstraps, key switch) is forwarded to the C ABI of include/mosfhet_hip.h -- there is no CPU implementation
 * of those in this library.  Reference file:line citations are in the header next to each prototype.
 */
#define _GNU_SOURCE
#include "compat_internal.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define W 64

void mc_die(const char *what) {
  fprintf(stderr, "mosfhet_amd: %s: %s\n", what, mosfhet_hip_last_error());
  abort();
}

void *mc_xmalloc(size_t sz) {
  void *p = NULL;
  if (posix_memalign(&p, 64, sz ? sz : 64)) {
    perror("mosfhet_amd: allocation failed");
    exit(EXIT_FAILURE);
  }
  return p;
}
void *safe_malloc(size_t size) { return mc_xmalloc(size); }           /* src/misc.c:104-113 */
void *safe_aligned_malloc(size_t size) { return mc_xmalloc(size); }   /* src/misc.c:115-128 */

/* ------------------------------------------------------------------ engine: one context per device in use (mosfhet_compat_multi.c) */
static mosfhet_hip_ctx_t g_ctxs[MC_MAX_DEVICES];
static int g_started = 0;

int mc_engine_started(void) { return __atomic_load_n(&g_started, __ATOMIC_ACQUIRE); }

void mosfhet_set_device(int device) {
  if (mc_engine_started() && device != g_mc_devs[0]) {
    fprintf(stderr, "mosfhet_amd: mosfhet_set_device after the engine was created\n");
    abort();
  }
  g_mc_ndev = 1;
  g_mc_devs[0] = device;
}

/* The layer is re-entrant like the reference (thread-local FFT state there, src/polynomial.c:269-352): the engine is created once under a lock,
 * keys are read-only after creation and may be shared by any number of host threads (device temporaries belong to the calling thread,
 * csrc/capi.hip), every thread has its own staging buffer and its own random stream (csprng.c). */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void engine_start(void) {
  pthread_mutex_lock(&g_lock);
  if (!g_started) {
    if (g_mc_devs[0] < 0) {
      mc_devices_from_env();
      if (g_mc_devs[0] < 0) {
        const char *e = getenv("MOSFHET_HIP_DEVICE");
        g_mc_devs[0] = e ? atoi(e) : 0;
      }
    }
    for (int d = 0; d < g_mc_ndev; d++)
      if (mosfhet_hip_ctx_create(&g_ctxs[d], g_mc_devs[d])) mc_die("engine start-up");
    /* the device generators' noise key comes from this layer's generator: the operating system's entropy, or the mosfhet_seed stream in test runs */
    uint8_t secret[32];
    mc_rnd_bytes(secret, sizeof(secret));
    if (mosfhet_hip_set_keygen_secret(secret)) mc_die("engine start-up (key-generation secret)");
    __atomic_store_n(&g_started, 1, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&g_lock);
}

mosfhet_hip_ctx_t mc_ctx_of(int d) {
  if (!mc_engine_started()) engine_start();
  return g_ctxs[d];
}

void *mosfhet_engine_ctx(void) { return mc_ctx_of(t_mc_dev); }

/* HIP's current device is per host thread and starts at 0: every thread that allocates staging memory or creates streams for the engine first makes
 * ITS device current (the C ABI does the same at the top of every call). */
void mc_use_device(void) {
  (void)mosfhet_engine_ctx();
  if (hipSetDevice(g_mc_devs[t_mc_dev])) {
    fprintf(stderr, "mosfhet_amd: hipSetDevice(%d) failed\n", g_mc_devs[t_mc_dev]);
    abort();
  }
}

/* ------------------------------------------------------------------ torus scalars */
double torus2double(Torus x) { return (double)x / 18446744073709551616.0; }
Torus double2torus(double x) { return (Torus)((int64_t)(18446744073709551616.0 * x)); }
uint64_t torus2int(Torus x, int log_scale) { return (x + ((Torus)1 << (W - log_scale - 1))) >> (W - log_scale); }
Torus int2torus(uint64_t x, int log_scale) { return x << (W - log_scale); }

/* ------------------------------------------------------------------ polynomials */
/* Every polynomial shell this library hands out -- torus domain (host coefficients) or DFT domain (device coefficients) -- sits behind a hidden tag:
 * the reference frees both kinds through the same void * functions (free_polynomial, free_trlwe, free_trgsw: src/polynomial.c:40-53, src/trlwe.c:86-94). */
typedef struct { uint64_t kind, pad; struct _TorusPolynomial p; } PolyBox;
#define POLY_BOX(ptr) ((PolyBox *)((char *)(ptr) - offsetof(PolyBox, p)))

void *mc_poly_shell(int kind, void *coeffs, int N) {
  PolyBox *b = (PolyBox *)mc_xmalloc(sizeof(*b));
  b->kind = (uint64_t)kind;
  b->pad = 0;
  b->p.coeffs = (Torus *)coeffs;
  b->p.N = N;
  return &b->p;
}

McShare *mc_share_new(void *dev, long refs) {
  McShare *sh = (McShare *)mc_xmalloc(sizeof(*sh));
  sh->dev = dev;
  sh->refs = refs;
  if (refs <= 0) {   /* an empty array: nothing will ever release the block */
    mc_use_device();
    hipFree(dev);
    sh->dev = NULL;
  }
  return sh;
}
void *mc_poly_shell_shared(McShare *share, void *coeffs, int N) {
  void *p = mc_poly_shell(MC_POLY_DFT_SHARED, coeffs, N);
  POLY_BOX(p)->pad = (uint64_t)(uintptr_t)share;
  return p;
}

int mc_poly_kind(const void *poly) { return (int)POLY_BOX(poly)->kind; }

TorusPolynomial polynomial_new_torus_polynomial(int N) {
  return (TorusPolynomial)mc_poly_shell(MC_POLY_TORUS, mc_xmalloc(sizeof(Torus) * (size_t)N), N);
}

void free_polynomial(void *p) {
  if (!p) return;
  PolyBox *b = POLY_BOX(p);
  switch ((int)b->kind) {
    case MC_POLY_TORUS: free(b->p.coeffs); break;
    case MC_POLY_DFT_OWNER: mc_use_device(); hipFree(b->p.coeffs); break;   /* the device block of a DFT-domain object hangs off its first polynomial */
    case MC_POLY_DFT_VIEW: break;
    case MC_POLY_DFT_SHARED: {   /* one element of an array that shares a device block: the last reference releases it */
      McShare *sh = (McShare *)(uintptr_t)b->pad;
      if (__atomic_sub_fetch(&sh->refs, 1, __ATOMIC_ACQ_REL) == 0) {
        mc_use_device();
        hipFree(sh->dev);
        free(sh);
      }
      break;
    }
    default:
      fprintf(stderr, "mosfhet_amd: free_polynomial: %p was not allocated by this library\n", p);
      abort();
  }
  b->kind = 0;
  free(b);
}
void free_DFT_polynomial(DFT_Polynomial p) { free_polynomial(p); }   /* src/polynomial.c:47-53 */

/* exact negacyclic out += a * s; fast path for 0/1 coefficients of s */
static void negacyclic_mul_addto(Torus *out, const Torus *a, const Torus *s, int N) {
  for (int i = 0; i < N; i++) {
    const Torus m = s[i];
    if (!m) continue;
    if (m == 1) {
      for (int j = i; j < N; j++) out[j] += a[j - i];
      for (int j = 0; j < i; j++) out[j] -= a[N + j - i];
    } else {
      for (int j = i; j < N; j++) out[j] += a[j - i] * m;
      for (int j = 0; j < i; j++) out[j] -= a[N + j - i] * m;
    }
  }
}

/* ------------------------------------------------------------------ TLWE */
TLWE_Key tlwe_alloc_key(int n, double sigma) {
  TLWE_Key k = (TLWE_Key)mc_xmalloc(sizeof(*k));
  k->n = n;
  k->sigma = sigma;
  k->s = (Integer *)mc_xmalloc(sizeof(Integer) * (size_t)n);
  return k;
}

TLWE_Key tlwe_new_binary_key(int n, double sigma) {
  TLWE_Key k = tlwe_alloc_key(n, sigma);
  for (int i = 0; i < n; i++) k->s[i] = mc_rnd64() & 1;
  return k;
}

void free_tlwe_key(TLWE_Key key) {
  if (!key) return;
  free(key->s);
  free(key);
}

TLWE tlwe_alloc_sample(int n) {
  TLWE c = (TLWE)mc_xmalloc(sizeof(*c));
  c->a = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)n);
  c->n = n;
  c->b = 0;
  return c;
}

TLWE *tlwe_alloc_sample_array(int count, int n) {
  TLWE *r = (TLWE *)mc_xmalloc(sizeof(TLWE) * (size_t)count);
  for (int i = 0; i < count; i++) r[i] = tlwe_alloc_sample(n);
  return r;
}

void free_tlwe(TLWE p) {
  if (!p) return;
  free(p->a);
  free(p);
}

void free_tlwe_array(TLWE *p, int count) {
  for (int i = 0; i < count; i++) free_tlwe(p[i]);
  free(p);
}

void tlwe_noiseless_trivial_sample(TLWE out, Torus m) {
  memset(out->a, 0, sizeof(Torus) * (size_t)out->n);
  out->b = m;
}

void mosfhet_tlwe_sample_flat(Torus *out, Torus m, TLWE_Key key) {
  Torus b = m;
  for (int i = 0; i < key->n; i++) {
    out[i] = mc_rnd64();
    b += key->s[i] * out[i];
  }
  out[key->n] = b + double2torus(mc_rnd_normal(key->sigma));
}

void tlwe_sample(TLWE out, Torus m, TLWE_Key key) {
  Torus b = m;
  for (int i = 0; i < key->n; i++) {
    out->a[i] = mc_rnd64();
    b += key->s[i] * out->a[i];
  }
  out->b = b + double2torus(mc_rnd_normal(key->sigma));
}

TLWE tlwe_new_sample(Torus m, TLWE_Key key) {
  TLWE c = tlwe_alloc_sample(key->n);
  tlwe_sample(c, m, key);
  return c;
}

Torus tlwe_phase(TLWE c, TLWE_Key key) {
  Torus acc = 0;
  for (int i = 0; i < key->n; i++) acc += key->s[i] * c->a[i];
  return c->b - acc;
}

void tlwe_copy(TLWE out, TLWE in) {
  memcpy(out->a, in->a, sizeof(Torus) * (size_t)in->n);
  out->b = in->b;
}

TLWE tlwe_new_noiseless_trivial_sample(Torus m, int n) {
  TLWE c = tlwe_alloc_sample(n);
  tlwe_noiseless_trivial_sample(c, m);
  return c;
}
#define TLWE_BINOP(name, expr_a, expr_b)                                  \
  void name(TLWE out, TLWE in1, TLWE in2) {                               \
    for (int i = 0; i < in1->n; i++) out->a[i] = expr_a;                  \
    out->b = expr_b;                                                      \
  }
TLWE_BINOP(tlwe_add, in1->a[i] + in2->a[i], in1->b + in2->b)
TLWE_BINOP(tlwe_sub, in1->a[i] - in2->a[i], in1->b - in2->b)
void tlwe_addto(TLWE out, TLWE in) { tlwe_add(out, out, in); }
void tlwe_subto(TLWE out, TLWE in) { tlwe_sub(out, out, in); }
void tlwe_negate(TLWE out, TLWE in) {
  for (int i = 0; i < in->n; i++) out->a[i] = (Torus)0 - in->a[i];
  out->b = (Torus)0 - in->b;
}
void tlwe_scale(TLWE out, TLWE in1, Torus in2) {
  for (int i = 0; i < in1->n; i++) out->a[i] = in1->a[i] * in2;
  out->b = in1->b * in2;
}
void tlwe_scale_addto(TLWE out, TLWE in1, Torus in2) {
  for (int i = 0; i < in1->n; i++) out->a[i] += in1->a[i] * in2;
  out->b += in1->b * in2;
}
void tlwe_scale_subto(TLWE out, TLWE in1, Torus in2) {
  for (int i = 0; i < in1->n; i++) out->a[i] -= in1->a[i] * in2;
  out->b -= in1->b * in2;
}

/* ------------------------------------------------------------------ TRLWE */
TRLWE_Key trlwe_alloc_key(int N, int k, double sigma) {
  TRLWE_Key key = (TRLWE_Key)mc_xmalloc(sizeof(*key));
  key->k = k;
  key->sigma = sigma;
  key->s = (IntPolynomial *)mc_xmalloc(sizeof(IntPolynomial) * (size_t)k);
  key->s_dft = NULL; /* DFT-domain data is device resident in this engine */
  for (int i = 0; i < k; i++) key->s[i] = polynomial_new_torus_polynomial(N);
  return key;
}

TRLWE_Key trlwe_new_binary_key(int N, int k, double sigma) {
  TRLWE_Key key = trlwe_alloc_key(N, k, sigma);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < N; j++) key->s[i]->coeffs[j] = mc_rnd64() & 1;
  return key;
}

void free_trlwe_key(TRLWE_Key key) {
  if (!key) return;
  for (int i = 0; i < key->k; i++) free_polynomial(key->s[i]);
  free(key->s);
  free(key);
}

TRLWE trlwe_alloc_new_sample(int k, int N) {
  TRLWE c = (TRLWE)mc_xmalloc(sizeof(*c));
  c->a = (TorusPolynomial *)mc_xmalloc(sizeof(TorusPolynomial) * (size_t)k);
  for (int i = 0; i < k; i++) c->a[i] = polynomial_new_torus_polynomial(N);
  c->b = polynomial_new_torus_polynomial(N);
  c->k = k;
  return c;
}

void free_trlwe(void *pv) {
  TRLWE p = (TRLWE)pv;
  if (!p) return;
  for (int i = 0; i < p->k; i++) free_polynomial(p->a[i]);
  free_polynomial(p->b);
  free(p->a);
  free(p);
}

void trlwe_noiseless_trivial_sample(TRLWE out, TorusPolynomial m) {
  const int N = out->b->N;
  for (int i = 0; i < out->k; i++) memset(out->a[i]->coeffs, 0, sizeof(Torus) * (size_t)N);
  if (m) memcpy(out->b->coeffs, m->coeffs, sizeof(Torus) * (size_t)N);
  else memset(out->b->coeffs, 0, sizeof(Torus) * (size_t)N);
}

TRLWE trlwe_new_noiseless_trivial_sample(TorusPolynomial m, int k, int N) {
  TRLWE c = trlwe_alloc_new_sample(k, N);
  trlwe_noiseless_trivial_sample(c, m);
  return c;
}

void trlwe_sample(TRLWE out, TorusPolynomial m, TRLWE_Key key) {
  const int N = key->s[0]->N;
  for (int i = 0; i < key->k; i++)
    for (int j = 0; j < N; j++) out->a[i]->coeffs[j] = mc_rnd64();
  for (int j = 0; j < N; j++) out->b->coeffs[j] = double2torus(mc_rnd_normal(key->sigma));
  for (int i = 0; i < key->k; i++) negacyclic_mul_addto(out->b->coeffs, out->a[i]->coeffs, key->s[i]->coeffs, N);
  if (m)
    for (int j = 0; j < N; j++) out->b->coeffs[j] += m->coeffs[j];
}

void trlwe_phase(TorusPolynomial out, TRLWE in, TRLWE_Key key) {
  const int N = key->s[0]->N;
  memset(out->coeffs, 0, sizeof(Torus) * (size_t)N);
  for (int i = 0; i < in->k; i++) negacyclic_mul_addto(out->coeffs, in->a[i]->coeffs, key->s[i]->coeffs, N);
  for (int j = 0; j < N; j++) out->coeffs[j] = in->b->coeffs[j] - out->coeffs[j];
}

static void poly_op(TorusPolynomial out, TorusPolynomial a, TorusPolynomial b, int sign) {
  for (int j = 0; j < out->N; j++) out->coeffs[j] = sign > 0 ? a->coeffs[j] + b->coeffs[j] : a->coeffs[j] - b->coeffs[j];
}
void trlwe_add(TRLWE out, TRLWE in1, TRLWE in2) {
  for (int p = 0; p < out->k; p++) poly_op(out->a[p], in1->a[p], in2->a[p], +1);
  poly_op(out->b, in1->b, in2->b, +1);
}
void trlwe_sub(TRLWE out, TRLWE in1, TRLWE in2) {
  for (int p = 0; p < out->k; p++) poly_op(out->a[p], in1->a[p], in2->a[p], -1);
  poly_op(out->b, in1->b, in2->b, -1);
}
void trlwe_addto(TRLWE out, TRLWE in) { trlwe_add(out, out, in); }
void trlwe_subto(TRLWE out, TRLWE in) { trlwe_sub(out, out, in); }
void trlwe_negate(TRLWE out, TRLWE in) {
  for (int p = 0; p <= out->k; p++) {
    TorusPolynomial o = p < out->k ? out->a[p] : out->b, i = p < in->k ? in->a[p] : in->b;
    for (int j = 0; j < o->N; j++) o->coeffs[j] = (Torus)0 - i->coeffs[j];
  }
}
void trlwe_copy(TRLWE out, TRLWE in) {
  for (int p = 0; p < out->k; p++) memcpy(out->a[p]->coeffs, in->a[p]->coeffs, sizeof(Torus) * (size_t)in->b->N);
  memcpy(out->b->coeffs, in->b->coeffs, sizeof(Torus) * (size_t)in->b->N);
}
/* out = in * X^a, a in [0, 2N)  (src/polynomial.c:184-199) */
void trlwe_mul_by_xai(TRLWE out, TRLWE in, int a) {
  const int N = in->b->N;
  a &= 2 * N - 1;
  for (int p = 0; p <= out->k; p++) {
    const Torus *src = (p < in->k ? in->a[p] : in->b)->coeffs;
    Torus *dst = (p < out->k ? out->a[p] : out->b)->coeffs;
    for (int j = 0; j < N; j++) {
      int s = j - a, neg = 0;
      while (s < 0) { s += N; neg ^= 1; }
      dst[j] = neg ? (Torus)0 - src[s] : src[s];
    }
  }
}

void trlwe_torus_packing(TRLWE out, Torus *in, int size) {
  const int N = out->b->N;
  trlwe_noiseless_trivial_sample(out, NULL);
  for (int i = 0; i < N; i++) out->b->coeffs[i] = in[i / (N / size)];
}

void trlwe_extract_tlwe_key(TLWE_Key out, TRLWE_Key in) {
  const int N = in->s[0]->N;
  for (int i = 0; i < in->k; i++) memcpy(out->s + (size_t)i * N, in->s[i]->coeffs, sizeof(Torus) * (size_t)N);
}

void trlwe_extract_tlwe(TLWE out, TRLWE in, int idx) {
  const int N = in->b->N;
  for (int p = 0; p < in->k; p++) {
    const Torus *ap = in->a[p]->coeffs;
    for (int j = 0; j <= idx; j++) out->a[p * N + j] = ap[idx - j];
    for (int j = idx + 1; j < N; j++) out->a[p * N + j] = (Torus)0 - ap[N + idx - j];
  }
  out->b = in->b->coeffs[idx];
}

static void extract_acc(TLWE out, TRLWE in, int idx, int sign) {
  const int N = in->b->N;
  for (int p = 0; p < in->k; p++) {
    const Torus *c = in->a[p]->coeffs;
    Torus *o = out->a + (size_t)p * N;
    for (int j = 0; j <= idx; j++) o[j] += sign > 0 ? c[idx - j] : (Torus)0 - c[idx - j];
    for (int j = idx + 1; j < N; j++) o[j] += sign > 0 ? (Torus)0 - c[N + idx - j] : c[N + idx - j];
  }
  out->b += sign > 0 ? in->b->coeffs[idx] : (Torus)0 - in->b->coeffs[idx];
}
void trlwe_extract_tlwe_addto(TLWE out, TRLWE in, int idx) { extract_acc(out, in, idx, +1); }
void trlwe_extract_tlwe_subto(TLWE out, TRLWE in, int idx) { extract_acc(out, in, idx, -1); }
void trlwe_mv_extract_tlwe(TLWE *out, TRLWE in, int amount) {
  const int N = in->b->N;
  for (int i = 0; i < amount / 2; i++) trlwe_extract_tlwe(out[i], in, i);
  for (int i = amount / 2; i < amount; i++) {
    trlwe_extract_tlwe(out[i], in, N - 1 - (i - amount / 2));
    tlwe_negate(out[i], out[i]);
  }
}
void trlwe_mv_extract_tlwe_scaling(TLWE out, TRLWE in, int scale) {
  const int N = in->b->N, amount = scale;
  trlwe_extract_tlwe(out, in, amount / 2);
  for (int i = amount / 2 + 1; i < amount; i++) extract_acc(out, in, N - 1 - (i - amount / 2), -1);
  for (int i = 0; i < amount / 2; i++) extract_acc(out, in, i, +1);
}
void trlwe_mv_extract_tlwe_scaling_addto(TLWE out, TRLWE in, int scale) {
  const int N = in->b->N, amount = scale;
  for (int i = amount / 2; i < amount; i++) extract_acc(out, in, N - 1 - (i - amount / 2), -1);
  for (int i = 0; i < amount / 2; i++) extract_acc(out, in, i, +1);
}
void trlwe_mv_extract_tlwe_scaling_subto(TLWE out, TRLWE in, int scale) {
  const int N = in->b->N, amount = scale;
  for (int i = amount / 2; i < amount; i++) extract_acc(out, in, N - 1 - (i - amount / 2), +1);
  for (int i = 0; i < amount / 2; i++) extract_acc(out, in, i, -1);
}

/* ------------------------------------------------------------------ TRGSW */
TRGSW_Key trgsw_new_key(TRLWE_Key trlwe_key, int l, int Bg_bit) {
  TRGSW_Key k = (TRGSW_Key)mc_xmalloc(sizeof(*k));
  k->trlwe_key = trlwe_key;
  k->l = l;
  k->Bg_bit = Bg_bit;
  return k;
}

void free_trgsw_key(TRGSW_Key key) { free(key); }

TRGSW trgsw_alloc_new_sample(int l, int Bg_bit, int k, int N) {
  TRGSW g = (TRGSW)mc_xmalloc(sizeof(*g));
  g->samples = (TRLWE *)mc_xmalloc(sizeof(TRLWE) * (size_t)l * (k + 1));
  for (int i = 0; i < l * (k + 1); i++) g->samples[i] = trlwe_alloc_new_sample(k, N);
  g->l = l;
  g->Bg_bit = Bg_bit;
  return g;
}

void free_trgsw(void *pv) {
  TRGSW p = (TRGSW)pv;
  if (!p) return;
  const int rows = p->l * (p->samples[0]->k + 1);
  for (int i = 0; i < rows; i++) free_trlwe(p->samples[i]);
  free(p->samples);
  free(p);
}

void trgsw_monomial_sample(TRGSW out, int64_t m, int e, TRGSW_Key key) {
  const int l = key->l, k = key->trlwe_key->k, N = key->trlwe_key->s[0]->N;
  if (e & N) m = -m;
  e &= N - 1;
  for (int q = 0; q < l * (k + 1); q++) trlwe_sample(out->samples[q], NULL, key->trlwe_key);
  for (int j = 0; j < l; j++) {
    const Torus h = (Torus)1 << (W - (j + 1) * key->Bg_bit);
    for (int p = 0; p < k; p++) out->samples[p * l + j]->a[p]->coeffs[e] += (Torus)m * h;
    out->samples[k * l + j]->b->coeffs[e] += (Torus)m * h;
  }
}

/* ------------------------------------------------------------------ flat marshalling */
void mc_trlwe_to_flat(Torus *flat, TRLWE c) {
  const int N = c->b->N;
  for (int p = 0; p < c->k; p++) memcpy(flat + (size_t)p * N, c->a[p]->coeffs, sizeof(Torus) * (size_t)N);
  memcpy(flat + (size_t)c->k * N, c->b->coeffs, sizeof(Torus) * (size_t)N);
}

void mc_trlwe_from_flat(TRLWE c, const Torus *flat) {
  const int N = c->b->N;
  for (int p = 0; p < c->k; p++) memcpy(c->a[p]->coeffs, flat + (size_t)p * N, sizeof(Torus) * (size_t)N);
  memcpy(c->b->coeffs, flat + (size_t)c->k * N, sizeof(Torus) * (size_t)N);
}

/* ------------------------------------------------------------------ marshalling pool
 * The reference's callers hold batches as arrays of separately allocated TLWE structs (applications/multi-ciphertext-arith/src/lut.c:12-17); turning
 * 4096 of them into the flat device layout and back is 53 MB of pointer-chasing copies -- 2.5 ms on one core next to a 15.5 ms kernel.  A small pool
 * of helper threads (plain memcpy workers: they never touch HIP) splits those loops; the calling thread takes part.  One job at a time: a second
 * caller thread that finds the pool busy simply copies on its own.  MOSFHET_HIP_MARSHAL_THREADS = helpers (default 1 .. 7 by core count, 0 disables). */
typedef void (*mc_range_fn)(void *arg, int lo, int hi);
static struct {
  pthread_mutex_t user, lock;
  pthread_cond_t go, done;
  int helpers, started, generation, count, next, grain, working;
  mc_range_fn fn;
  void *arg;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, -1, 0, 0, 0, 0, 0, 0, NULL, NULL};

static int pool_take(int *lo, int *hi) {   /* g_pool.lock held */
  if (g_pool.next >= g_pool.count) return 0;
  *lo = g_pool.next;
  *hi = *lo + g_pool.grain < g_pool.count ? *lo + g_pool.grain : g_pool.count;
  g_pool.next = *hi;
  return 1;
}
static int g_pool_stop = 0;                       /* set (under g_pool.lock) when the library is unloaded: workers leave */
static pthread_t g_pool_threads[16];
static pthread_once_t g_pool_once = PTHREAD_ONCE_INIT;
static void *pool_worker(void *unused) {
  (void)unused;
  int seen = 0;
  pthread_mutex_lock(&g_pool.lock);
  for (;;) {
    while (g_pool.generation == seen && !g_pool_stop) pthread_cond_wait(&g_pool.go, &g_pool.lock);
    if (g_pool_stop) break;
    seen = g_pool.generation;
    g_pool.working++;
    int lo, hi;
    while (pool_take(&lo, &hi)) {
      pthread_mutex_unlock(&g_pool.lock);
      g_pool.fn(g_pool.arg, lo, hi);
      pthread_mutex_lock(&g_pool.lock);
    }
    if (--g_pool.working == 0) pthread_cond_broadcast(&g_pool.done);
  }
  pthread_mutex_unlock(&g_pool.lock);
  return NULL;
}
/* fork(): the child has none of the workers and may have inherited the locks in any state -- it starts over with an empty pool */
static void pool_atfork_child(void) {
  pthread_mutex_init(&g_pool.user, NULL);
  pthread_mutex_init(&g_pool.lock, NULL);
  pthread_cond_init(&g_pool.go, NULL);
  pthread_cond_init(&g_pool.done, NULL);
  g_pool.started = 0; g_pool.working = 0; g_pool.count = 0; g_pool.next = 0;
}
static void pool_init_once(void) {
  const char *e = getenv("MOSFHET_HIP_MARSHAL_THREADS");
  /* default: half of the cores this process may use, less the caller, between 1 and 7 helpers (3 on an 8-core host; a GPU node has far more cores than
   * the copies can use: 7 helpers measured 251.8 against 246 - 250 k bootstraps/s with 3) */
  int cores = 8;
  cpu_set_t set;
  if (!sched_getaffinity(0, sizeof(set), &set)) cores = CPU_COUNT(&set);
  int want = e ? atoi(e) : (cores / 2 - 1 < 1 ? 1 : (cores / 2 - 1 > 7 ? 7 : cores / 2 - 1));
  if (want < 0) want = 0;
  if (want > 15) want = 15;
  __atomic_store_n(&g_pool.helpers, want, __ATOMIC_RELEASE);
  pthread_atfork(NULL, NULL, pool_atfork_child);
}
/* the library is going away (exit, or dlclose of a ctypes handle): the workers run code of this very object, so they are stopped and joined first */
__attribute__((destructor)) static void pool_shutdown(void) {
  pthread_mutex_lock(&g_pool.lock);
  const int n = g_pool.started;
  g_pool_stop = 1;
  pthread_cond_broadcast(&g_pool.go);
  pthread_mutex_unlock(&g_pool.lock);
  for (int i = 0; i < n; i++) pthread_join(g_pool_threads[i], NULL);
  pthread_mutex_lock(&g_pool.lock);
  g_pool.started = 0;
  pthread_mutex_unlock(&g_pool.lock);
}
static void mc_parallel_for(mc_range_fn fn, void *arg, int count, int grain) {
  pthread_once(&g_pool_once, pool_init_once);
  if (__atomic_load_n(&g_pool.helpers, __ATOMIC_ACQUIRE) == 0 || count < 4 * grain || pthread_mutex_trylock(&g_pool.user)) {
    fn(arg, 0, count);
    return;
  }
  pthread_mutex_lock(&g_pool.lock);
  while (!g_pool_stop && g_pool.started < __atomic_load_n(&g_pool.helpers, __ATOMIC_ACQUIRE)) {
    if (pthread_create(&g_pool_threads[g_pool.started], NULL, pool_worker, NULL)) { __atomic_store_n(&g_pool.helpers, g_pool.started, __ATOMIC_RELEASE); break; }
    g_pool.started++;
  }
  g_pool.fn = fn; g_pool.arg = arg; g_pool.count = count; g_pool.next = 0; g_pool.grain = grain;
  g_pool.generation++;
  g_pool.working++;                    /* the caller works too */
  pthread_cond_broadcast(&g_pool.go);
  int lo, hi;
  while (pool_take(&lo, &hi)) {
    pthread_mutex_unlock(&g_pool.lock);
    fn(arg, lo, hi);
    pthread_mutex_lock(&g_pool.lock);
  }
  g_pool.working--;
  while (g_pool.working > 0) pthread_cond_wait(&g_pool.done, &g_pool.lock);
  pthread_mutex_unlock(&g_pool.lock);
  pthread_mutex_unlock(&g_pool.user);
}

typedef struct { Torus *flat; TLWE *c; int n; } TlweSpan;
static void tlwe_to_flat_range(void *pv, int lo, int hi) {
  const TlweSpan *a = (const TlweSpan *)pv;
  for (int i = lo; i < hi; i++) {
    memcpy(a->flat + (size_t)i * (a->n + 1), a->c[i]->a, sizeof(Torus) * (size_t)a->n);
    a->flat[(size_t)i * (a->n + 1) + a->n] = a->c[i]->b;
  }
}
static void tlwe_from_flat_range(void *pv, int lo, int hi) {
  const TlweSpan *a = (const TlweSpan *)pv;
  for (int i = lo; i < hi; i++) {
    memcpy(a->c[i]->a, a->flat + (size_t)i * (a->n + 1), sizeof(Torus) * (size_t)a->n);
    a->c[i]->b = a->flat[(size_t)i * (a->n + 1) + a->n];
  }
}
static void tlwe_array_to_flat(Torus *flat, TLWE *c, int count, int n) {
  if (n < 0) return;
  TlweSpan a = {flat, c, n};
  mc_parallel_for(tlwe_to_flat_range, &a, count, 64);
}

static void tlwe_array_from_flat(TLWE *c, const Torus *flat, int count, int n) {
  if (n < 0) return;
  TlweSpan a = {(Torus *)flat, c, n};
  mc_parallel_for(tlwe_from_flat_range, &a, count, 64);
}

void *mc_dev_alloc(size_t bytes) {
  void *p = NULL;
  mc_use_device();
  if (hipMalloc(&p, bytes ? bytes : 8)) {
    fprintf(stderr, "mosfhet_amd: hipMalloc(%zu) failed\n", bytes);
    abort();
  }
  return p;
}

/* Device staging buffer of the single-call wrappers: every wrapper is synchronous (it waits for its result before returning), so one
 * growable buffer per host thread is reused instead of a hipMalloc / hipFree pair per call (those cost more than the copies). */
static __thread void *g_stage = NULL;
static __thread size_t g_stage_bytes = 0;
static pthread_key_t g_stage_key;                      /* its destructor releases a thread's buffer when the thread exits */
static pthread_once_t g_stage_once = PTHREAD_ONCE_INIT;
static void stage_release(void *p) { if (p) hipFree(p); }
static void stage_key_init(void) { pthread_key_create(&g_stage_key, stage_release); }
void *mc_stage_alloc(size_t bytes) {
  if (bytes > g_stage_bytes) {
    if (g_stage) hipFree(g_stage);
    g_stage_bytes = bytes < 65536 ? 65536 : bytes + bytes / 2;
    g_stage = mc_dev_alloc(g_stage_bytes);
    pthread_once(&g_stage_once, stage_key_init);
    pthread_setspecific(g_stage_key, g_stage);
  }
  return g_stage;
}
static void stage_free(void *p) { (void)p; }

/* Host staging of the wrappers (the flat batches copied to / from the device) is PINNED memory, a few cached buffers per thread: copies of
 * page-locked memory run at PCIe speed, pageable ones at a fraction of it (the batch entry points move 13 KB per bootstrap).  The wrappers
 * nest at most two deep; anything beyond the cached slots falls back to ordinary memory. */
#define HSTAGE_SLOTS 4
static __thread struct { void *p; size_t bytes; int used; } g_hstage[HSTAGE_SLOTS];
static pthread_key_t g_hstage_key;
static pthread_once_t g_hstage_once = PTHREAD_ONCE_INIT;
static void hstage_release(void *unused) {
  (void)unused;
  for (int i = 0; i < HSTAGE_SLOTS; i++)
    if (g_hstage[i].p) { hipHostFree(g_hstage[i].p); g_hstage[i].p = NULL; }
}
static void hstage_key_init(void) { pthread_key_create(&g_hstage_key, hstage_release); }
void *mc_hstage_alloc(size_t bytes) {
  if (!bytes) bytes = 64;
  mc_use_device();
  for (int i = 0; i < HSTAGE_SLOTS; i++) {
    if (g_hstage[i].used) continue;
    if (g_hstage[i].bytes < bytes) {
      if (g_hstage[i].p) hipHostFree(g_hstage[i].p);
      g_hstage[i].p = NULL;
      g_hstage[i].bytes = 0;
      const size_t want = bytes < 65536 ? 65536 : bytes + bytes / 2;
      if (hipHostMalloc(&g_hstage[i].p, want, 0) || !g_hstage[i].p) { g_hstage[i].p = NULL; break; }   /* cannot pin: ordinary memory below */
      g_hstage[i].bytes = want;
      pthread_once(&g_hstage_once, hstage_key_init);
      pthread_setspecific(g_hstage_key, g_hstage);      /* non-NULL: the destructor runs at thread exit */
    }
    g_hstage[i].used = 1;
    return g_hstage[i].p;
  }
  return mc_xmalloc(bytes);
}
void mc_hstage_free(void *p) {
  for (int i = 0; i < HSTAGE_SLOTS; i++)
    if (g_hstage[i].p == p && g_hstage[i].used) { g_hstage[i].used = 0; return; }
  free(p);
}

void mc_dev_copy(void *dst, const void *src, size_t bytes, int kind) {
  if (bytes && hipMemcpy(dst, src, bytes, kind)) {
    fprintf(stderr, "mosfhet_amd: hipMemcpy failed\n");
    abort();
  }
}

/* ------------------------------------------------------------------ bootstrap key */
void mosfhet_gen_bootstrap_key_flat(Torus *out, TRGSW_Key out_key, TLWE_Key in_key) {
  const int l = out_key->l, k = out_key->trlwe_key->k, N = out_key->trlwe_key->s[0]->N;
  const size_t row = (size_t)(k + 1) * N, sz = (size_t)(k + 1) * l * row;
  TRGSW tmp = trgsw_alloc_new_sample(l, out_key->Bg_bit, k, N);
  for (int i = 0; i < in_key->n; i++) {
    trgsw_monomial_sample(tmp, (int64_t)in_key->s[i], 0, out_key);
    for (int q = 0; q < (k + 1) * l; q++) mc_trlwe_to_flat(out + (size_t)i * sz + q * row, tmp->samples[q]);
  }
  free_trgsw(tmp);
}

/* Keys alive in this process, for the one reference signature that names no key (functional_bootstrap_trgsw_phase2, src/bootstrap.c:297). */
#define MAX_KEYS 64
static Bootstrap_Key g_keys[MAX_KEYS];

static void remember_key(Bootstrap_Key key) {
  pthread_mutex_lock(&g_lock);
  int seen = 0;
  for (int i = 0; i < MAX_KEYS; i++)
    if (g_keys[i] == key) seen = 1;
  for (int i = 0; i < MAX_KEYS && !seen; i++)
    if (!g_keys[i]) { g_keys[i] = key; seen = 1; }
  pthread_mutex_unlock(&g_lock);
}

static void forget_key(Bootstrap_Key key) {
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < MAX_KEYS; i++)
    if (g_keys[i] == key) g_keys[i] = NULL;
  pthread_mutex_unlock(&g_lock);
}

void *mosfhet_bootstrap_key_device(Bootstrap_Key key) { return key ? key->device : NULL; }

/* Bootstrap_Key.s: the reference's array of n TRGSW_DFT (src/bootstrap.c:7-19), here n views of the device-resident key */
static TRGSW_DFT *key_views(void *dev, int n, int l, int Bg_bit, int N) {
  const double *base = mosfhet_hip_bsk_device_dft((mosfhet_hip_bsk_t)dev);
  return base ? mc_trgsw_dft_views((double *)base, n, l, Bg_bit, N) : NULL;
}

/* src/bootstrap.c:23-48: 2^u torus-domain TRGSW samples per group of u key bits, su[i 2^u/u + j] = TRGSW(indicator of bit pattern j) */
void mosfhet_gen_bootstrap_key_unfolded_flat(Torus *out, TRGSW_Key out_key, TLWE_Key in_key, int unfolding) {
  const int l = out_key->l, N = out_key->trlwe_key->s[0]->N, key_exp = 1 << unfolding, final_exp = key_exp / unfolding;
  const size_t row = (size_t)2 * N, sz = (size_t)2 * l * row;
  TRGSW tmp = trgsw_alloc_new_sample(l, out_key->Bg_bit, 1, N);
  for (int i = 0; i < in_key->n; i += unfolding)
    for (int j = 0; j < key_exp; j++) {
      Torus key = 1;
      for (int u = 0, j_ = j; u < unfolding; u++, j_ >>= 1) key *= (j_ & 1) ? in_key->s[i + u] : 1 - in_key->s[i + u];
      trgsw_monomial_sample(tmp, (int64_t)key, 0, out_key);
      for (int q = 0; q < 2 * l; q++) mc_trlwe_to_flat(out + ((size_t)i * final_exp + j) * sz + q * row, tmp->samples[q]);
    }
  free_trgsw(tmp);
}

Bootstrap_Key new_bootstrap_key(TRGSW_Key out_key, TLWE_Key in_key, int unfolding) {
  const int l = out_key->l, k = out_key->trlwe_key->k, N = out_key->trlwe_key->s[0]->N, n = in_key->n;
  if ((unfolding != 1 && unfolding != 2 && unfolding != 4 && unfolding != 8) || n % unfolding || (unfolding > 1 && k != 1)) {
    fprintf(stderr, "mosfhet_amd: new_bootstrap_key: unfolding = %d must be 1, 2, 4 or 8 with n divisible by it (test/tests.c:34) and k = 1; the "
                    "reference's key layout (src/bootstrap.c:34-45: group stride 2^u / u as an integer) is only consistent for powers of two\n", unfolding);
    abort();
  }
  Bootstrap_Key res = (Bootstrap_Key)mc_xmalloc(sizeof(*res));
  res->n = n; res->k = k; res->l = l; res->N = N; res->Bg_bit = out_key->Bg_bit; res->unfolding = unfolding;
  res->su = NULL;   /* device resident as well (reference: host TRGSW array, src/bootstrap.c:35) */
  mosfhet_hip_bsk_t dev = NULL;
  if (k != 1 || (N != 1024 && N != 2048 && N != 4096)) {
    /* k > 1 or a ring without a tuned kernel (the reference is generic in both, src/trgsw.c:385-423): the engine's general path (csrc/general_kernels.h)
     * serves functional / programmable bootstraps (+ wo_extract), the full-domain bootstrap, multi-value bootstraps and key switch + bootstrap with such a
     * key.  Bootstrap_Key.s (the array legacy callers hand to blind_rotate) is not provided. */
    if (unfolding != 1 || k < 1 || k > 3 || N < 256 || N > 16384 || (N & (N - 1))) {
      fprintf(stderr, "mosfhet_amd: new_bootstrap_key: k = %d, N = %d, unfolding = %d has no kernel (k <= 3, N a power of two in 256 .. 16384, unfolding 1 "
                      "outside k = 1, N in 1024 / 2048 / 4096)\n", k, N, unfolding);
      abort();
    }
    if (N <= 8192) {
      /* encrypted on the device like the tuned rings' keys (mosfhet_hip_bsk_generate_k: exact a * s over the k key polynomials, ChaCha20 noise) */
      Torus *s_all = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)k * N);
      for (int m = 0; m < k; m++) memcpy(s_all + (size_t)m * N, out_key->trlwe_key->s[m]->coeffs, sizeof(Torus) * (size_t)N);
      if (mosfhet_hip_bsk_generate_k((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, s_all, k, N, in_key->s, n, l, out_key->Bg_bit, out_key->trlwe_key->sigma, mc_rnd64()))
        mc_die("new_bootstrap_key (general ring)");
      free(s_all);
    } else {   /* N = 16384: encrypted on the host (trgsw_monomial_sample), transformed on the device */
      const size_t words = (size_t)n * (k + 1) * l * (k + 1) * N;
      Torus *flat = (Torus *)mc_xmalloc(sizeof(Torus) * words);
      mosfhet_gen_bootstrap_key_flat(flat, out_key, in_key);
      if (mosfhet_hip_bsk_create((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, flat, n, k, N, l, out_key->Bg_bit)) mc_die("new_bootstrap_key (general ring)");
      free(flat);
    }
    res->device = dev;
    res->s = NULL;
    remember_key(res);
    return res;
  }
  if (unfolding == 1) {
    /* encrypted on the device (counter-based generator seeded from this thread's stream), transformed there: no host copy of the key */
    if (mosfhet_hip_bsk_generate((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, out_key->trlwe_key->s[0]->coeffs, N, in_key->s, n, l, out_key->Bg_bit,
                                 out_key->trlwe_key->sigma, mc_rnd64(), 0))
      mc_die("new_bootstrap_key");
  } else {
    /* the 2^u / u * n torus-domain samples are encrypted on the device too (u = 8 at lvl2: 5 GB of them) */
    if (mosfhet_hip_bsk_unfolded_generate((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, out_key->trlwe_key->s[0]->coeffs, N, in_key->s, n, l, out_key->Bg_bit,
                                          out_key->trlwe_key->sigma, mc_rnd64(), unfolding))
      mc_die("new_bootstrap_key (unfolded)");
  }
  res->device = dev;
  res->s = key_views(dev, n, l, out_key->Bg_bit, N);
  remember_key(res);
  return res;
}

void free_bootstrap_key(Bootstrap_Key key) {
  if (!key) return;
  forget_key(key);
  if (key->s) mc_trgsw_dft_views_free(key->s, key->n);
  mc_replicas_free(key->device);
  mosfhet_hip_bsk_destroy((mosfhet_hip_bsk_t)key->device);
  free(key);
}

/* ------------------------------------------------------------------ bootstraps (GPU) */
enum { MODE_FUNCTIONAL, MODE_PROGRAMMABLE, MODE_WO_EXTRACT };
#define MC_SHOULD_SHARD(count) (g_mc_ndev > 1 && !t_mc_in_shard && (count) >= 2 * g_mc_ndev)   /* several devices, not already inside a slice */

/* Large batches are pipelined in chunks over two streams of the calling thread: while the GPU bootstraps chunk c, the host packs chunk c + 1 into
 * the pinned staging buffer and unpacks the results of chunk c - 1 (the pointer-chasing TLWE arrays cost about as much host time as the PCIe
 * copies).  Chunks of 1024 (MOSFHET_HIP_PIPE_CHUNK): two of them on the two streams are one residency round of the chip. */
#define PIPE_CHUNK pipe_chunk()
static int pipe_chunk(void) {   /* MOSFHET_HIP_PIPE_CHUNK overrides (tools/compat_latency.c sweeps it) */
  static int v = 0;
  if (!v) {
    const char *e = getenv("MOSFHET_HIP_PIPE_CHUNK");
    int x = e ? atoi(e) : 1024;
    v = x >= 64 ? x : 1024;
  }
  return v;
}
/* Two streams of the calling thread, chunks alternate; a chunk's upload, kernel and download are queued on ITS stream.  That keeps the residency rounds
 * aligned: the kernel of chunk c + 2 sits behind the download of chunk c, which ends about when chunk c + 1's kernel does, so the two kernels of a round
 * start together (round 4 measured the alternative -- dedicated copy streams, every chunk uploaded as early as the host packs it, kernels waiting for
 * nothing but their own upload: 19.5 instead of 16.8 ms per 4096; queued back to back, the blocks of the next round trickle in as the blocks of this one
 * finish and the workgroups no longer walk the key together: experiments/README.md).  A chunk comes back in pieces of 512 ciphertexts, an event behind
 * each, so the host unpacks one piece while the next lands -- what is left exposed at the end is one piece, not a chunk. */
static __thread void *g_pipe_streams[2];
static int pipe_piece(void) {   /* MOSFHET_HIP_PIPE_PIECE overrides (ciphertexts per download piece) */
  static int v = 0;
  if (!v) {
    const char *e = getenv("MOSFHET_HIP_PIPE_PIECE");
    v = e ? atoi(e) : 512;
    if (v < 32) v = 32;
  }
  return v;
}
static void bootstrap_pipelined(int mode, TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, int a0, int kappa, int theta) {
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  mosfhet_hip_bsk_t bsk = (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK);
  const int n = key->n, N = key->N, k = key->k;
  const size_t in_row = (size_t)n + 1, out_row = (size_t)k * N + 1, tv_w = (size_t)(k + 1) * N;
  const size_t in_w = (size_t)count * in_row, out_w = (size_t)count * out_row;
  mc_use_device();
  for (int i = 0; i < 2; i++)
    if (!g_pipe_streams[i] && hipStreamCreate(&g_pipe_streams[i])) mc_die("bootstrap (stream)");
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  Torus *h_tv = h + in_w, *h_out = h + in_w + tv_w, *d_tv = d + in_w, *d_out = d + in_w + tv_w;
  const int piece = pipe_piece(), n_pieces = (count + piece - 1) / piece + (count + PIPE_CHUNK - 1) / PIPE_CHUNK;
  void **ev = (void **)mc_xmalloc(sizeof(void *) * (size_t)n_pieces);
  int made = 0, taken = 0;   /* events recorded / waited for; pieces are unpacked in the order they were queued */
  mc_trlwe_to_flat(h_tv, tv);
  mc_dev_copy(d_tv, h_tv, sizeof(Torus) * tv_w, HIP_H2D);
  int prev_lo = -1, prev_cnt = 0;
  for (int lo = 0, c = 0;; lo += PIPE_CHUNK, c++) {   /* one more turn than there are chunks: the last one only unpacks */
    const int cnt = lo < count ? (count - lo < PIPE_CHUNK ? count - lo : PIPE_CHUNK) : 0;
    if (cnt) {
      void *st = g_pipe_streams[c & 1];
      tlwe_array_to_flat(h + (size_t)lo * in_row, in + lo, cnt, n);
      if (hipMemcpyAsync(d + (size_t)lo * in_row, h + (size_t)lo * in_row, sizeof(Torus) * (size_t)cnt * in_row, HIP_H2D, st)) mc_die("bootstrap (copy in)");
      const int rc = mode == MODE_PROGRAMMABLE
                         ? mosfhet_hip_programmable_bootstrap_batch(ctx, bsk, d_out + (size_t)lo * out_row, d_tv, 1, d + (size_t)lo * in_row, cnt, a0, kappa, theta, st)
                         : mosfhet_hip_functional_bootstrap_batch(ctx, bsk, d_out + (size_t)lo * out_row, d_tv, 1, d + (size_t)lo * in_row, cnt, a0, st);
      if (rc) mc_die("bootstrap");
      for (int p = lo; p < lo + cnt; p += piece) {
        const int pc = lo + cnt - p < piece ? lo + cnt - p : piece;
        if (hipMemcpyAsync(h_out + (size_t)p * out_row, d_out + (size_t)p * out_row, sizeof(Torus) * (size_t)pc * out_row, HIP_D2H, st)) mc_die("bootstrap (copy out)");
        if (hipEventCreateWithFlags(&ev[made], HIP_EVENT_DISABLE_TIMING) || hipEventRecord(ev[made], st)) mc_die("bootstrap (event)");
        made++;
      }
    }
    if (prev_lo >= 0)   /* the previous chunk's results, piece by piece, while this chunk runs */
      for (int p = prev_lo; p < prev_lo + prev_cnt; p += piece, taken++) {
        const int pc = prev_lo + prev_cnt - p < piece ? prev_lo + prev_cnt - p : piece;
        if (hipEventSynchronize(ev[taken])) mc_die("bootstrap");
        tlwe_array_from_flat(out + p, h_out + (size_t)p * out_row, pc, k * N);
      }
    prev_lo = cnt ? lo : -1;
    prev_cnt = cnt;
    if (!cnt) break;
  }
  for (int i = 0; i < made; i++) hipEventDestroy(ev[i]);
  free(ev);
  stage_free(d);
  mc_hstage_free(h);
}

typedef struct { int mode; TLWE *out; TRLWE tv; TLWE *in; Bootstrap_Key key; int a0, kappa, theta; } BootstrapSlice;
static void bootstrap_many(int mode, TLWE *out, TRLWE out_trlwe, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, int a0, int kappa, int theta);
static void bootstrap_slice(void *pv, int lo, int hi) {
  BootstrapSlice *a = (BootstrapSlice *)pv;
  if (hi > lo) bootstrap_many(a->mode, a->out + lo, NULL, a->tv, a->in + lo, hi - lo, a->key, a->a0, a->kappa, a->theta);
}

static void bootstrap_many(int mode, TLWE *out, TRLWE out_trlwe, TRLWE tv, TLWE *in, int count, Bootstrap_Key key,
                           int a0, int kappa, int theta) {
  if (g_mc_ndev > 1 && !t_mc_in_shard && mode != MODE_WO_EXTRACT && count >= 2 * g_mc_ndev) {
    /* several devices: contiguous slices, one per device, each through this same function on its device's host thread (mosfhet_compat_multi.c) */
    BootstrapSlice a = {mode, out, tv, in, key, a0, kappa, theta};
    void *keys[1] = {key->device};
    const int kinds[1] = {MC_KEY_BSK};
    mc_run_sharded(bootstrap_slice, &a, count, keys, kinds, 1);
    return;
  }
  /* unfolded keys keep the single-stream path: their accumulators live in the calling thread's device pool, one set per thread, so chunks on two
   * streams would share it (mosfhet_hip.h: compositions on several streams must be ordered by the caller) */
  if (mode != MODE_WO_EXTRACT && count >= 2 * PIPE_CHUNK && key->unfolding == 1) {
    bootstrap_pipelined(mode, out, tv, in, count, key, a0, kappa, theta);
    return;
  }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  mosfhet_hip_bsk_t bsk = (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK);
  const int n = key->n, N = key->N, k = key->k;
  const size_t in_w = (size_t)count * (n + 1), tv_w = (size_t)(k + 1) * N;
  const size_t out_w = (mode == MODE_WO_EXTRACT) ? (size_t)count * tv_w : (size_t)count * (k * N + 1);
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  tlwe_array_to_flat(h, in, count, n);
  mc_trlwe_to_flat(h + in_w, tv);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * (in_w + tv_w), HIP_H2D);
  int rc;
  if (mode == MODE_PROGRAMMABLE)
    rc = mosfhet_hip_programmable_bootstrap_batch(ctx, bsk, d + in_w + tv_w, d + in_w, 1, d, count, a0, kappa, theta, NULL);
  else if (mode == MODE_FUNCTIONAL)
    rc = mosfhet_hip_functional_bootstrap_batch(ctx, bsk, d + in_w + tv_w, d + in_w, 1, d, count, a0, NULL);
  else
    rc = mosfhet_hip_functional_bootstrap_wo_extract_batch(ctx, bsk, d + in_w + tv_w, d + in_w, 1, d, count, a0, NULL);
  if (rc || mosfhet_hip_ctx_sync(ctx, NULL)) mc_die("bootstrap");
  mc_dev_copy(h + in_w + tv_w, d + in_w + tv_w, sizeof(Torus) * out_w, HIP_D2H);
  if (mode == MODE_WO_EXTRACT) mc_trlwe_from_flat(out_trlwe, h + in_w + tv_w);
  else tlwe_array_from_flat(out, h + in_w + tv_w, count, k * N);
  stage_free(d);
  mc_hstage_free(h);
}

void functional_bootstrap_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, int torus_base) {
  bootstrap_many(MODE_FUNCTIONAL, out, NULL, tv, in, count, key, torus_base, 0, 0);
}

void programmable_bootstrap_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, int precision, int kappa, int theta) {
  bootstrap_many(MODE_PROGRAMMABLE, out, NULL, tv, in, count, key, precision, kappa, theta);
}

void functional_bootstrap(TLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, int torus_base) {
  bootstrap_many(MODE_FUNCTIONAL, &out, NULL, tv, &in, 1, key, torus_base, 0, 0);
}

void programmable_bootstrap(TLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, int precision, int kappa, int theta) {
  bootstrap_many(MODE_PROGRAMMABLE, &out, NULL, tv, &in, 1, key, precision, kappa, theta);
}

void functional_bootstrap_wo_extract(TRLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, int torus_base) {
  bootstrap_many(MODE_WO_EXTRACT, NULL, out, tv, &in, 1, key, torus_base, 0, 0);
}

void trlwe_torus_packing_many_LUT(TRLWE out, Torus *in, int lut_size, int n_luts) {
  const int N = out->b->N, span = N / (lut_size * n_luts);
  trlwe_noiseless_trivial_sample(out, NULL);
  for (int i = 0; i < lut_size; i++)
    for (int j = 0; j < n_luts; j++)
      for (int r = 0; r < span; r++) out->b->coeffs[(i * n_luts + j) * span + r] = in[j * lut_size + i];
}

typedef struct { TLWE *out; TRLWE tv; TLWE *in; Bootstrap_Key key; TLWE_KS_Key ksk; int precision; } FdfbSlice;
static void fdfb_slice(void *pv, int lo, int hi) {
  FdfbSlice *a = (FdfbSlice *)pv;
  if (hi > lo) full_domain_functional_bootstrap_batch(a->out + lo, a->tv, a->in + lo, hi - lo, a->key, a->ksk, a->precision);
}

void full_domain_functional_bootstrap_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, TLWE_KS_Key ksk, int precision) {
  if (MC_SHOULD_SHARD(count)) {
    FdfbSlice a = {out, tv, in, key, ksk, precision};
    void *keys[2] = {key->device, ksk->device};
    const int kinds[2] = {MC_KEY_BSK, MC_KEY_KSK};
    mc_run_sharded(fdfb_slice, &a, count, keys, kinds, 2);
    return;
  }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int n = key->n, N = key->N, k = key->k;
  const size_t in_w = (size_t)count * (n + 1), tv_w = (size_t)(k + 1) * N, out_w = (size_t)count * (k * N + 1);
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  tlwe_array_to_flat(h, in, count, n);
  mc_trlwe_to_flat(h + in_w, tv);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * (in_w + tv_w), HIP_H2D);
  if (mosfhet_hip_full_domain_functional_bootstrap_batch(ctx, (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK),
                                                         (mosfhet_hip_ksk_t)mc_key_here(ksk->device, MC_KEY_KSK), d + in_w + tv_w, d + in_w, 1, d, count, precision, NULL) ||
      mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("full_domain_functional_bootstrap");
  mc_dev_copy(h + in_w + tv_w, d + in_w + tv_w, sizeof(Torus) * out_w, HIP_D2H);
  tlwe_array_from_flat(out, h + in_w + tv_w, count, k * N);
  stage_free(d);
  mc_hstage_free(h);
}

void full_domain_functional_bootstrap(TLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, TLWE_KS_Key ksk, int precision) {
  full_domain_functional_bootstrap_batch(&out, tv, &in, 1, key, ksk, precision);
}

/* The copy-out stream of the batch wrappers, one per host thread: NON-BLOCKING -- the launches sit on the default stream, and an ordinary stream's copies
 * queue up behind ALL of them (measured on circuit bootstraps: the first piece landed when the last kernel had finished); this one is ordered by events alone. */
static void *mc_copy_stream(void) {
  static __thread void *st;
  mc_use_device();
  if (!st && hipStreamCreateWithFlags(&st, HIP_STREAM_NON_BLOCKING)) mc_die("copy stream");
  return st;
}

/* A large batch of TLWE results (the kernels have finished: the caller synchronised): back in pieces with an event behind each, a piece unpacked into the
 * structs while the next one lands (8192 outputs of 16 KB for 1024 multi-value bootstraps at lvl2: 134 MB). */
static void tlwe_array_download(TLWE *out, Torus *h_flat, const Torus *d_flat, int count, int n_out) {
  static int piece_env = -1;   /* MOSFHET_COMPAT_TLWE_PIECE: outputs per piece (0 = one copy, then unpack); read once, by whichever thread comes first */
  int PIECE = __atomic_load_n(&piece_env, __ATOMIC_RELAXED);
  if (PIECE < 0) {
    const char *e = getenv("MOSFHET_COMPAT_TLWE_PIECE");
    PIECE = e ? atoi(e) : 1024;
    if (PIECE < 0) PIECE = 0;
    __atomic_store_n(&piece_env, PIECE, __ATOMIC_RELAXED);
  }
  const size_t item = (size_t)n_out + 1;
  if (PIECE == 0 || count <= PIECE) {
    mc_dev_copy(h_flat, d_flat, sizeof(Torus) * (size_t)count * item, HIP_D2H);
    tlwe_array_from_flat(out, h_flat, count, n_out);
    return;
  }
  const int n_pieces = (count + PIECE - 1) / PIECE;
  void *cs = mc_copy_stream();
  void **ev = (void **)mc_xmalloc(sizeof(void *) * (size_t)n_pieces);
  for (int p = 0; p < n_pieces; p++) {
    const int lo = p * PIECE, cnt = count - lo < PIECE ? count - lo : PIECE;
    if (hipMemcpyAsync(h_flat + (size_t)lo * item, d_flat + (size_t)lo * item, sizeof(Torus) * (size_t)cnt * item, HIP_D2H, cs) ||
        hipEventCreateWithFlags(&ev[p], HIP_EVENT_DISABLE_TIMING) || hipEventRecord(ev[p], cs))
      mc_die("copy out");
  }
  for (int p = 0; p < n_pieces; p++) {
    const int lo = p * PIECE, cnt = count - lo < PIECE ? count - lo : PIECE;
    if (hipEventSynchronize(ev[p])) mc_die("copy out");
    tlwe_array_from_flat(out + lo, h_flat + (size_t)lo * item, cnt, n_out);
    hipEventDestroy(ev[p]);
  }
  free(ev);
}

/* Every sharded entry point below follows bootstrap_many: with several devices in use (and not already inside a slice) the batch is cut by
 * mc_run_sharded into one contiguous slice per device, each slice re-enters the same function on its device's host thread, and every key handle is
 * looked up with mc_key_here (the replica on the calling thread's device).  SURVEY 8(e); src/bootstrap.c:222-230,346-366,391-517, src/bootstrap_ga.c:62-76
 * are the single-sample callers these batches stand for. */
typedef struct { TLWE *out; TRLWE tv; TLWE *in; Bootstrap_Key key; int torus_base, n_luts; } MvSlice;
static void mv_slice(void *pv, int lo, int hi) {
  MvSlice *a = (MvSlice *)pv;
  if (hi > lo) multivalue_bootstrap_CLOT21_batch(a->out + (size_t)lo * a->n_luts, a->tv, a->in + lo, hi - lo, a->key, a->torus_base, a->n_luts);
}

/* out[b * n_luts + j] = LUT j of input b (src/bootstrap.c:222-230 once per input: one blind rotation, n_luts extractions) */
void multivalue_bootstrap_CLOT21_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, int torus_base, int n_luts) {
  if (MC_SHOULD_SHARD(count)) {
    MvSlice a = {out, tv, in, key, torus_base, n_luts};
    void *keys[1] = {key->device};
    const int kinds[1] = {MC_KEY_BSK};
    mc_run_sharded(mv_slice, &a, count, keys, kinds, 1);
    return;
  }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int n = key->n, N = key->N, k = key->k;
  const size_t in_w = (size_t)count * (n + 1), tv_w = (size_t)(k + 1) * N, out_w = (size_t)count * n_luts * (k * N + 1);
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  tlwe_array_to_flat(h, in, count, n);
  mc_trlwe_to_flat(h + in_w, tv);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * (in_w + tv_w), HIP_H2D);
  if (mosfhet_hip_multivalue_bootstrap_CLOT21_batch(ctx, (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK), d + in_w + tv_w, d + in_w, 1, d,
                                                    count, torus_base, n_luts, NULL) ||
      mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("multivalue_bootstrap_CLOT21");
  tlwe_array_download(out, h + in_w + tv_w, d + in_w + tv_w, count * n_luts, k * N);
  stage_free(d);
  mc_hstage_free(h);
}

void multivalue_bootstrap_CLOT21(TLWE *out, TRLWE tv, TLWE in, Bootstrap_Key key, int torus_base, int n_luts) {
  multivalue_bootstrap_CLOT21_batch(out, tv, &in, 1, key, torus_base, n_luts);
}

/* ------------------------------------------------------------------ Galois-automorphism bootstrap */
void polynomial_permute(TorusPolynomial out, TorusPolynomial in, uint64_t gen) {
  const uint64_t N = (uint64_t)in->N, mask = N - 1;
  for (uint64_t i = 0; i < N; i++) {
    const uint64_t idx = i * gen;
    out->coeffs[idx & mask] = (idx & N) ? (Torus)0 - in->coeffs[i] : in->coeffs[i];
  }
}

void mosfhet_gen_bootstrap_key_ga_flat(Torus *out, TRGSW_Key out_key, TLWE_Key in_key) {
  const int l = out_key->l, k = out_key->trlwe_key->k, N = out_key->trlwe_key->s[0]->N;
  const size_t row = (size_t)(k + 1) * N, sz = (size_t)(k + 1) * l * row;
  TRGSW tmp = trgsw_alloc_new_sample(l, out_key->Bg_bit, k, N);
  for (int i = 0; i < in_key->n; i++) {
    trgsw_monomial_sample(tmp, 1, (int)in_key->s[i], out_key);   /* src/bootstrap_ga.c:19 */
    for (int q = 0; q < (k + 1) * l; q++) mc_trlwe_to_flat(out + (size_t)i * sz + q * row, tmp->samples[q]);
  }
  free_trgsw(tmp);
}

/* entry j (generator 2j+1): rows r < t = TRLWE_key( key(X^(2j+1)) * 2^(64-(r+1) base_bit) )  [src/keyswitch.c:12-37,500-511] */
void mosfhet_gen_automorphism_keyset_flat(Torus *out, TRLWE_Key key, int t, int base_bit) {
  const int N = key->s[0]->N;
  TorusPolynomial s2 = polynomial_new_torus_polynomial(N), msg = polynomial_new_torus_polynomial(N);
  TRLWE tmp = trlwe_alloc_new_sample(1, N);
  for (int j = 0; j < N; j++) {
    polynomial_permute(s2, key->s[0], (uint64_t)(2 * j + 1));
    for (int r = 0; r < t; r++) {
      for (int i = 0; i < N; i++) msg->coeffs[i] = s2->coeffs[i] * ((Torus)1 << (W - (r + 1) * base_bit));
      trlwe_sample(tmp, msg, key);
      mc_trlwe_to_flat(out + (((size_t)j * t + r) * 2) * N, tmp);
    }
  }
  free_polynomial(s2);
  free_polynomial(msg);
  free_trlwe(tmp);
}

static mosfhet_hip_gak_t fft_ks_keys_new(TRLWE_Key out_key, const Torus *msgs, int entries, int t, int base_bit, const char *who);
static TRLWE_KS_Key trlwe_ks_header(void *dev, int entry, int owner, int t, int base_bit);

Bootstrap_GA_Key new_bootstrap_key_ga(TRGSW_Key out_key, TLWE_Key in_key) {
  const int l = out_key->l, k = out_key->trlwe_key->k, N = out_key->trlwe_key->s[0]->N, n = in_key->n;
  if (k != 1) { fprintf(stderr, "mosfhet_amd: new_bootstrap_key_ga: k = 1 only\n"); abort(); }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  Bootstrap_GA_Key res = (Bootstrap_GA_Key)mc_xmalloc(sizeof(*res));
  res->n = n; res->k = k; res->l = l; res->N = N; res->Bg_bit = out_key->Bg_bit; res->unfolding = 1;
  res->su = NULL;
  /* the TRGSW(X^{s_i}) samples are encrypted on the device (mosfhet_hip_bsk_generate, ga = 1) */
  mosfhet_hip_bsk_t dev = NULL;
  if (mosfhet_hip_bsk_generate(ctx, &dev, out_key->trlwe_key->s[0]->coeffs, N, in_key->s, n, l, out_key->Bg_bit, out_key->trlwe_key->sigma, mc_rnd64(), 1))
    mc_die("new_bootstrap_key_ga");
  /* automorphism key set (src/bootstrap_ga.c:10: t = l, base_bit = Bg_bit): entry j switches from s(X^(2j+1)), encrypted on the device too */
  Torus *msgs = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)N * N);
  TorusPolynomial perm = polynomial_new_torus_polynomial(N);
  for (int j = 0; j < N; j++) {
    polynomial_permute(perm, out_key->trlwe_key->s[0], (uint64_t)(2 * j + 1));
    memcpy(msgs + (size_t)j * N, perm->coeffs, sizeof(Torus) * (size_t)N);
  }
  free_polynomial(perm);
  mosfhet_hip_gak_t gak = fft_ks_keys_new(out_key->trlwe_key, msgs, N, l, out_key->Bg_bit, "new_bootstrap_key_ga (automorphism keys)");
  free(msgs);
  res->device = dev;
  res->ak_device = gak;
  res->s = key_views(dev, n, l, out_key->Bg_bit, N);
  res->ak = (TRLWE_KS_Key *)mc_xmalloc(sizeof(TRLWE_KS_Key) * (size_t)N);   /* src/bootstrap_ga.c:10: entry j <-> generator 2j + 1 */
  for (int j = 0; j < N; j++) res->ak[j] = trlwe_ks_header(gak, j, 0, l, out_key->Bg_bit);
  return res;
}

void free_bootstrap_key_ga(Bootstrap_GA_Key key) {
  if (!key) return;
  if (key->s) mc_trgsw_dft_views_free(key->s, key->n);
  for (int j = 0; j < key->N; j++) free(key->ak[j]);
  free(key->ak);
  mc_replicas_free(key->device);
  mc_replicas_free(key->ak_device);
  mosfhet_hip_bsk_destroy((mosfhet_hip_bsk_t)key->device);
  mosfhet_hip_gak_destroy((mosfhet_hip_gak_t)key->ak_device);
  free(key);
}

typedef struct { TLWE *out; TRLWE tv; TLWE *in; Bootstrap_GA_Key key; int torus_base; } GaSlice;
static void bootstrap_ga_many(TLWE *out, TRLWE out_trlwe, TRLWE tv, TLWE *in, int count, Bootstrap_GA_Key key, int torus_base);
static void ga_slice(void *pv, int lo, int hi) {
  GaSlice *a = (GaSlice *)pv;
  if (hi > lo) bootstrap_ga_many(a->out + lo, NULL, a->tv, a->in + lo, hi - lo, a->key, a->torus_base);
}

static void bootstrap_ga_many(TLWE *out, TRLWE out_trlwe, TRLWE tv, TLWE *in, int count, Bootstrap_GA_Key key, int torus_base) {
  if (out && MC_SHOULD_SHARD(count)) {   /* bootstrap key + automorphism key set replicated (src/bootstrap_ga.c:62-76 per sample) */
    GaSlice a = {out, tv, in, key, torus_base};
    void *keys[2] = {key->device, key->ak_device};
    const int kinds[2] = {MC_KEY_BSK, MC_KEY_GAK};
    mc_run_sharded(ga_slice, &a, count, keys, kinds, 2);
    return;
  }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int n = key->n, N = key->N, extract = out != NULL;
  const size_t in_w = (size_t)count * (n + 1), tv_w = (size_t)2 * N;
  const size_t out_w = extract ? (size_t)count * (N + 1) : (size_t)count * tv_w;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  tlwe_array_to_flat(h, in, count, n);
  mc_trlwe_to_flat(h + in_w, tv);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + tv_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * (in_w + tv_w), HIP_H2D);
  if (mosfhet_hip_functional_bootstrap_ga_batch(ctx, (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK), (mosfhet_hip_gak_t)mc_key_here(key->ak_device, MC_KEY_GAK),
                                                d + in_w + tv_w, d + in_w, 1, d,
                                                count, torus_base, extract, NULL) ||
      mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("functional_bootstrap_ga");
  mc_dev_copy(h + in_w + tv_w, d + in_w + tv_w, sizeof(Torus) * out_w, HIP_D2H);
  if (extract) tlwe_array_from_flat(out, h + in_w + tv_w, count, N);
  else mc_trlwe_from_flat(out_trlwe, h + in_w + tv_w);
  stage_free(d);
  mc_hstage_free(h);
}

void functional_bootstrap_ga_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_GA_Key key, int torus_base) {
  bootstrap_ga_many(out, NULL, tv, in, count, key, torus_base);
}

void functional_bootstrap_ga(TLWE out, TRLWE tv, TLWE in, Bootstrap_GA_Key key, int torus_base) {
  bootstrap_ga_many(&out, NULL, tv, &in, 1, key, torus_base);
}

void functional_bootstrap_wo_extract_ga(TRLWE out, TRLWE tv, TLWE in, Bootstrap_GA_Key key, int torus_base) {
  bootstrap_ga_many(NULL, out, tv, &in, 1, key, torus_base);
}

/* ------------------------------------------------------------------ LWE key switch */
void mosfhet_gen_tlwe_ks_key_flat(Torus *out, TLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  const int base = 1 << base_bit;
  const size_t row = (size_t)out_key->n + 1;
  for (int i = 0; i < in_key->n; i++)
    for (int j = 0; j < t; j++)
      for (int v = 1; v < base; v++)
        mosfhet_tlwe_sample_flat(out + (((size_t)i * t + j) * (base - 1) + (v - 1)) * row,
                                 in_key->s[i] * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit)), out_key);
}

/* host view with the reference's shape s[i][j][v] (mosfhet.h:62-65): TLWE headers aliasing the flat table, which the key owns; + device copy */
static TLWE_KS_Key tlwe_ks_wrap_dev(Torus *flat, int n_in, int n_out, int t, int base_bit, const char *who, mosfhet_hip_ksk_t have);
static TLWE_KS_Key tlwe_ks_wrap(Torus *flat, int n_in, int n_out, int t, int base_bit, const char *who) { return tlwe_ks_wrap_dev(flat, n_in, n_out, t, base_bit, who, NULL); }
/* have != NULL: the device table already exists (made there: tlwe_new_KS_key) and `flat` is its exported image; otherwise `flat` is uploaded */
static TLWE_KS_Key tlwe_ks_wrap_dev(Torus *flat, int n_in, int n_out, int t, int base_bit, const char *who, mosfhet_hip_ksk_t have) {
  const int base = 1 << base_bit;
  const size_t row = (size_t)n_out + 1;
  TLWE_KS_Key res = (TLWE_KS_Key)mc_xmalloc(sizeof(*res));
  res->base_bit = base_bit;
  res->t = t;
  res->n = n_in;
  res->s = (TLWE ***)mc_xmalloc(sizeof(TLWE **) * (size_t)n_in);
  for (int i = 0; i < n_in; i++) {
    res->s[i] = (TLWE **)mc_xmalloc(sizeof(TLWE *) * (size_t)t);
    for (int j = 0; j < t; j++) {
      res->s[i][j] = (TLWE *)mc_xmalloc(sizeof(TLWE) * (size_t)(base - 1));
      for (int v = 0; v < base - 1; v++) {
        Torus *r = flat + (((size_t)i * t + j) * (base - 1) + v) * row;
        TLWE c = (TLWE)mc_xmalloc(sizeof(*c));
        c->a = r;
        c->b = r[n_out];
        c->n = n_out;
        res->s[i][j][v] = c;
      }
    }
  }
  mosfhet_hip_ksk_t dev = have;
  if (!dev && mosfhet_hip_ksk_create((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, flat, n_in, n_out, t, base_bit)) mc_die(who);
  res->device = dev;
  return res;
}

/* tlwe_new_KS_key (src/tlwe.c:193-212).  The table is encrypted ON THE DEVICE (mosfhet_hip_tlwe_ksk_generate: exact a * s, Gaussian noise of the output key's sigma,
 * masks from a seed drawn from the host generator; lvl2's 1.2 GB in milliseconds instead of seconds of host encryption) and read back once, because its samples
 * key->s[i][j][v] are caller-visible in the reference's struct; the device copy that made them is the one the key switches use.  Parameters the device generator
 * does not take are encrypted on the host and uploaded, as before. */
TLWE_KS_Key tlwe_new_KS_key(TLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  const size_t rows = (size_t)in_key->n * t * ((1 << base_bit) - 1), words = rows * ((size_t)out_key->n + 1);
  Torus *flat = (Torus *)mc_xmalloc(sizeof(Torus) * words);
  mosfhet_hip_ksk_t dev = NULL;
  /* (noise: the device generators' ChaCha key is drawn from this layer's seedable generator at engine start-up and again by mosfhet_seed -- csprng.c -- so the table
   * is reproducible under mosfhet_seed like everything else) */
  const int rc = mosfhet_hip_tlwe_ksk_generate((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, out_key->s, out_key->n, in_key->s, in_key->n, t, base_bit, out_key->sigma, mc_rnd64(), 0);
  if (rc == 0) {
    if (mosfhet_hip_ksk_export_rows(dev, 0, rows, flat)) mc_die("tlwe_new_KS_key (read back)");
    return tlwe_ks_wrap_dev(flat, in_key->n, out_key->n, t, base_bit, "tlwe_new_KS_key", dev);
  }
  if (rc != MOSFHET_HIP_EINVAL) mc_die("tlwe_new_KS_key (device generation)");   /* only parameters the device generator does not take fall back to host encryption */
  mosfhet_gen_tlwe_ks_key_flat(flat, out_key, in_key, t, base_bit);
  return tlwe_ks_wrap(flat, in_key->n, out_key->n, t, base_bit, "tlwe_new_KS_key");
}

void free_tlwe_ks_key(TLWE_KS_Key key) {
  if (!key) return;
  const int base = 1 << key->base_bit;
  Torus *flat = key->s[0][0][0]->a;
  for (int i = 0; i < key->n; i++) {
    for (int j = 0; j < key->t; j++) {
      for (int v = 0; v < base - 1; v++) free(key->s[i][j][v]);
      free(key->s[i][j]);
    }
    free(key->s[i]);
  }
  free(key->s);
  free(flat);
  mc_replicas_free(key->device);
  mosfhet_hip_ksk_destroy((mosfhet_hip_ksk_t)key->device);
  free(key);
}

typedef struct { TLWE *out, *in; TLWE_KS_Key ks; } KsSlice;
static void ks_slice(void *pv, int lo, int hi) {
  KsSlice *a = (KsSlice *)pv;
  if (hi > lo) tlwe_keyswitch_batch(a->out + lo, a->in + lo, hi - lo, a->ks);
}

void tlwe_keyswitch_batch(TLWE *out, TLWE *in, int count, TLWE_KS_Key ks) {
  if (MC_SHOULD_SHARD(count)) {
    KsSlice a = {out, in, ks};
    void *keys[1] = {ks->device};
    const int kinds[1] = {MC_KEY_KSK};
    mc_run_sharded(ks_slice, &a, count, keys, kinds, 1);
    return;
  }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int n_in = ks->n, n_out = out[0]->n;
  const size_t in_w = (size_t)count * (n_in + 1), out_w = (size_t)count * (n_out + 1);
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + out_w));
  tlwe_array_to_flat(h, in, count, n_in);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * in_w, HIP_H2D);
  if (mosfhet_hip_tlwe_keyswitch_batch(ctx, (mosfhet_hip_ksk_t)mc_key_here(ks->device, MC_KEY_KSK), d + in_w, d, count, NULL) || mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("tlwe_keyswitch");
  mc_dev_copy(h + in_w, d + in_w, sizeof(Torus) * out_w, HIP_D2H);
  tlwe_array_from_flat(out, h + in_w, count, n_out);
  stage_free(d);
  mc_hstage_free(h);
}

void tlwe_keyswitch(TLWE out, TLWE in, TLWE_KS_Key ks_key) { tlwe_keyswitch_batch(&out, &in, 1, ks_key); }

/* ------------------------------------------------------------------ TRLWE key switches, circuit bootstrap */
/* rows r < t = TRLWE_out( s_in * 2^(64 - (r+1) base_bit) )   [src/keyswitch.c:12-37] */
void mosfhet_gen_trlwe_ks_key_flat(Torus *out, const Torus *s_in, TRLWE_Key out_key, int t, int base_bit) {
  const int N = out_key->s[0]->N;
  TorusPolynomial msg = polynomial_new_torus_polynomial(N);
  TRLWE tmp = trlwe_alloc_new_sample(1, N);
  for (int r = 0; r < t; r++) {
    for (int i = 0; i < N; i++) msg->coeffs[i] = s_in[i] * ((Torus)1 << (W - (r + 1) * base_bit));
    trlwe_sample(tmp, msg, out_key);
    mc_trlwe_to_flat(out + (size_t)r * 2 * N, tmp);
  }
  free_polynomial(msg);
  free_trlwe(tmp);
}

/* entry 0 switches from -s_out * s_in, entry 1 from -s_out   [src/keyswitch.c:39-50] */
void mosfhet_gen_priv_ks_key_flat(Torus *out, TRLWE_Key out_key, TRLWE_Key in_key, int t, int base_bit) {
  const int N = out_key->s[0]->N;
  Torus *neg = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)N), *prod = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)N);
  for (int i = 0; i < N; i++) neg[i] = (Torus)0 - out_key->s[0]->coeffs[i];
  memset(prod, 0, sizeof(Torus) * (size_t)N);
  negacyclic_mul_addto(prod, neg, in_key->s[0]->coeffs, N);
  mosfhet_gen_trlwe_ks_key_flat(out, prod, out_key, t, base_bit);
  mosfhet_gen_trlwe_ks_key_flat(out + (size_t)t * 2 * N, neg, out_key, t, base_bit);
  free(neg);
  free(prod);
}

/* s[i][j][v-1] = TRLWE_out( constant s_in[i] * v * 2^(64 - (j+1) base_bit) )   [src/keyswitch.c:368-390] */
void mosfhet_gen_packing1_ks_key_flat(Torus *out, TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  const int N = out_key->s[0]->N, base = 1 << base_bit;
  TRLWE tmp = trlwe_alloc_new_sample(1, N);
  for (int i = 0; i < in_key->n; i++)
    for (int j = 0; j < t; j++)
      for (int v = 1; v < base; v++) {
        trlwe_sample(tmp, NULL, out_key);
        tmp->b->coeffs[0] += in_key->s[i] * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit));
        mc_trlwe_to_flat(out + ((((size_t)i * t + j) * (base - 1)) + (v - 1)) * 2 * N, tmp);
      }
  free_trlwe(tmp);
}

static TRLWE_KS_Key trlwe_ks_header(void *dev, int entry, int owner, int t, int base_bit) {
  TRLWE_KS_Key res = (TRLWE_KS_Key)mc_xmalloc(sizeof(*res));
  res->s = NULL; res->base_bit = base_bit; res->t = t; res->k = 1;
  res->device = dev; res->entry = entry; res->owner = owner;
  return res;
}

/* FFT key-switch keys are encrypted on the device (mosfhet_hip_trlwe_ksk_generate): msgs = the polynomials being switched from, [entries][N] */
static mosfhet_hip_gak_t fft_ks_keys_new(TRLWE_Key out_key, const Torus *msgs, int entries, int t, int base_bit, const char *who) {
  mosfhet_hip_gak_t dev = NULL;
  if (mosfhet_hip_trlwe_ksk_generate((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, out_key->s[0]->coeffs, out_key->s[0]->N, msgs, entries, t, base_bit,
                                     out_key->sigma, mc_rnd64()))
    mc_die(who);
  return dev;
}

TRLWE_KS_Key trlwe_new_KS_key(TRLWE_Key out_key, TRLWE_Key in_key, int t, int base_bit) {
  if (out_key->k != 1 || in_key->k != 1) { fprintf(stderr, "mosfhet_amd: trlwe_new_KS_key: k = 1 only\n"); abort(); }
  return trlwe_ks_header(fft_ks_keys_new(out_key, in_key->s[0]->coeffs, 1, t, base_bit, "trlwe_new_KS_key"), 0, 1, t, base_bit);
}

TRLWE_KS_Key *trlwe_new_priv_KS_key(TRLWE_Key out_key, TRLWE_Key in_key, int t, int base_bit) {
  const int N = out_key->s[0]->N;
  if (out_key->k != 1 || in_key->k != 1) { fprintf(stderr, "mosfhet_amd: trlwe_new_priv_KS_key: k = 1 only\n"); abort(); }
  /* entry 0 switches from -s_out * s_in, entry 1 from -s_out (src/keyswitch.c:39-50) */
  Torus *msgs = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)2 * N);
  memset(msgs, 0, sizeof(Torus) * (size_t)N);
  for (int i = 0; i < N; i++) msgs[N + i] = (Torus)0 - out_key->s[0]->coeffs[i];
  negacyclic_mul_addto(msgs, msgs + N, in_key->s[0]->coeffs, N);
  mosfhet_hip_gak_t dev = fft_ks_keys_new(out_key, msgs, 2, t, base_bit, "trlwe_new_priv_KS_key");
  free(msgs);
  TRLWE_KS_Key *res = (TRLWE_KS_Key *)mc_xmalloc(sizeof(TRLWE_KS_Key) * 2);
  res[0] = trlwe_ks_header(dev, 0, 1, t, base_bit);   /* entry 0 owns the shared device key set */
  res[1] = trlwe_ks_header(dev, 1, 0, t, base_bit);
  return res;
}

/* src/keyswitch.c:500-524: key sets for the automorphisms X -> X^gen: entry j switches from key(X^gens[j]) back to key.  One device key set behind
 * `count` headers (header 0 owns it: free the headers with free_trlwe_ks_key, header 0 last or first -- the set goes with it). */
static TRLWE_KS_Key *automorphism_keyset(TRLWE_Key key, const uint64_t *gens, int count, int t, int base_bit, const char *who) {
  const int N = key->s[0]->N;
  if (key->k != 1) { fprintf(stderr, "mosfhet_amd: %s: k = 1 only\n", who); abort(); }
  Torus *msgs = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)count * N);
  TorusPolynomial perm = polynomial_new_torus_polynomial(N);
  for (int j = 0; j < count; j++) {
    if (gens[j] & 1) polynomial_permute(perm, key->s[0], gens[j]);
    else memset(perm->coeffs, 0, sizeof(Torus) * (size_t)N);   /* even "generators" are no automorphisms of the ring: a key from 0, never usable */
    memcpy(msgs + (size_t)j * N, perm->coeffs, sizeof(Torus) * (size_t)N);
  }
  free_polynomial(perm);
  mosfhet_hip_gak_t dev = fft_ks_keys_new(key, msgs, count, t, base_bit, who);
  free(msgs);
  TRLWE_KS_Key *res = (TRLWE_KS_Key *)mc_xmalloc(sizeof(TRLWE_KS_Key) * (size_t)count);
  for (int j = 0; j < count; j++) res[j] = trlwe_ks_header(dev, j, j == 0, t, base_bit);
  return res;
}

TRLWE_KS_Key *trlwe_new_automorphism_KS_keyset(TRLWE_Key key, bool skip_even, int t, int base_bit) {
  const int N = key->s[0]->N, count = skip_even ? N : 2 * N;
  uint64_t *gens = (uint64_t *)mc_xmalloc(sizeof(uint64_t) * (size_t)count);
  for (int j = 0; j < count; j++) gens[j] = skip_even ? (uint64_t)(2 * j + 1) : (uint64_t)j;
  TRLWE_KS_Key *res = automorphism_keyset(key, gens, count, t, base_bit, "trlwe_new_automorphism_KS_keyset");
  free(gens);
  return res;
}

TRLWE_KS_Key *trlwe_new_automorphism_KS_keyset_2(TRLWE_Key key, uint64_t *gens, uint64_t size, int t, int base_bit) {
  return automorphism_keyset(key, gens, (int)size, t, base_bit, "trlwe_new_automorphism_KS_keyset_2");
}

void free_trlwe_ks_key(TRLWE_KS_Key key) {
  if (!key) return;
  if (key->owner) {
    mc_replicas_free(key->device);
    mosfhet_hip_gak_destroy((mosfhet_hip_gak_t)key->device);
  }
  free(key);
}

static void trlwe_ks_run(int mode, TRLWE out, TRLWE in, TRLWE_KS_Key key) {
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int N = in->b->N;
  const size_t w = (size_t)2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * w);
  mc_trlwe_to_flat(h, in);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * 2 * w);
  mc_dev_copy(d, h, sizeof(Torus) * w, HIP_H2D);
  int rc = mode ? mosfhet_hip_trlwe_priv_keyswitch_2_batch(ctx, (mosfhet_hip_gak_t)key->device, d + w, d, 1, NULL)
                : mosfhet_hip_trlwe_keyswitch_batch(ctx, (mosfhet_hip_gak_t)key->device, key->entry, d + w, d, 1, NULL);
  if (rc || mosfhet_hip_ctx_sync(ctx, NULL)) mc_die(mode ? "trlwe_priv_keyswitch_2" : "trlwe_keyswitch");
  mc_dev_copy(h, d + w, sizeof(Torus) * w, HIP_D2H);
  mc_trlwe_from_flat(out, h);
  stage_free(d);
  mc_hstage_free(h);
}

void trlwe_keyswitch(TRLWE out, TRLWE in, TRLWE_KS_Key ks_key) { trlwe_ks_run(0, out, in, ks_key); }
void trlwe_priv_keyswitch_2(TRLWE out, TRLWE in, TRLWE_KS_Key *ks_key) { trlwe_ks_run(1, out, in, ks_key[0]); }

/* The rows are generated ON THE DEVICE (mosfhet_hip_trlwe_table_ksk_generate: exact a * s, Gaussian noise, counter-based generator
 * seeded from the host generator): the 6 GB packing key of test_circuit_bootstrap (test/tests.c:974) takes milliseconds. */
static Generic_KS_Key table_key_new(int kind, TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit, const char *who) {
  const int N = out_key->s[0]->N;
  if (out_key->k != 1) { fprintf(stderr, "mosfhet_amd: %s: k = 1 only\n", who); abort(); }
  Generic_KS_Key res = (Generic_KS_Key)mc_xmalloc(sizeof(*res));
  res->s = NULL; res->base_bit = base_bit; res->t = t; res->n = in_key->n; res->include_b = kind;
  mosfhet_hip_ksk_t dev = NULL;
  /* full rows in HBM by default (6 GB for the lvl2 packing key: nothing on a 288 GB part, and the word-lane key switch streams stored rows 25 % faster than it
   * regenerates masks); MOSFHET_HIP_FULL_TABLE_KEYS=0 keeps the rows seed-compressed like the reference's default build (USE_COMPRESSED_TRLWE,
   * src/keyswitch.c:231-241): half the bytes, the key switches regenerate the masks (same results either way) */
  const char *full = getenv("MOSFHET_HIP_FULL_TABLE_KEYS");
  int rc = !(full && full[0] == '0')
               ? mosfhet_hip_trlwe_table_ksk_generate((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, kind, out_key->s[0]->coeffs, N, in_key->s, in_key->n, t,
                                                      base_bit, out_key->sigma, mc_rnd64())
               : mosfhet_hip_trlwe_table_ksk_generate_compressed((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, kind, out_key->s[0]->coeffs, N, in_key->s,
                                                                 in_key->n, t, base_bit, out_key->sigma, mc_rnd64());
  if (rc) mc_die(who);
  res->device = dev;
  return res;
}

Generic_KS_Key trlwe_new_packing1_KS_key(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  return table_key_new(0, out_key, in_key, t, base_bit, "trlwe_new_packing1_KS_key");
}

void free_trlwe_generic_ks_key(Generic_KS_Key key) {
  if (!key) return;
  mc_replicas_free(key->device);
  mosfhet_hip_ksk_destroy((mosfhet_hip_ksk_t)key->device);
  free(key);
}

void trlwe_packing1_keyswitch(TRLWE out, TLWE in, Generic_KS_Key ks) {
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int N = out->b->N, n = ks->n;
  const size_t in_w = (size_t)n + 1, out_w = (size_t)2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + out_w));
  tlwe_array_to_flat(h, &in, 1, n);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * in_w, HIP_H2D);
  if (mosfhet_hip_trlwe_packing1_keyswitch_batch(ctx, (mosfhet_hip_ksk_t)ks->device, d + in_w, d, 1, NULL) || mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("trlwe_packing1_keyswitch");
  mc_dev_copy(h + in_w, d + in_w, sizeof(Torus) * out_w, HIP_D2H);
  mc_trlwe_from_flat(out, h + in_w);
  stage_free(d);
  mc_hstage_free(h);
}

/* TRGSW outputs of a circuit-bootstrap batch: [count][2l][2][N] flat -> the structs' 2l (k + 1) separately allocated polynomials, over the helper threads
 * (256 KiB per output at lvl2: 268 MB per 1024 -- serial, that copy took as long as the kernels), level by level while the kernels still run.  A circuit bootstrap finishes gadget level i -- rows i and l + i of EVERY output -- long before level i + 1
 * (a packing key switch of the whole batch lies between them), and the C ABI records an event per level (mosfhet_hip_circuit_bootstrap_*_batch_ev).  The copy
 * stream waits for it and brings the two row sets back as strided copies (width one row, pitch one TRGSW), in pieces of PIECE outputs with an event each; the
 * host unpacks a piece while the next one lands.  Only the last level's copy is left behind the kernels (64 of 268 MB per 1024 outputs at lvl2). */
typedef struct { TRGSW *out; const Torus *flat; int rows, q0, q1; size_t row; } TrgswRows;
static void trgsw_rows_from_flat_range(void *pv, int lo, int hi) {
  const TrgswRows *a = (const TrgswRows *)pv;
  for (int b = lo; b < hi; b++) {
    mc_trlwe_from_flat(a->out[b]->samples[a->q0], a->flat + ((size_t)b * a->rows + a->q0) * a->row);
    mc_trlwe_from_flat(a->out[b]->samples[a->q1], a->flat + ((size_t)b * a->rows + a->q1) * a->row);
  }
}
typedef struct { void **level, **piece; int l, n_pieces; } LevelEvents;
enum { LEVEL_PIECE = 256 };
static LevelEvents level_events_new(int l, int count) {
  LevelEvents e;
  e.l = l;
  e.n_pieces = (count + LEVEL_PIECE - 1) / LEVEL_PIECE;
  mc_use_device();
  e.level = (void **)mc_xmalloc(sizeof(void *) * (size_t)l);
  e.piece = (void **)mc_xmalloc(sizeof(void *) * (size_t)l * (size_t)(e.n_pieces ? e.n_pieces : 1));
  for (int i = 0; i < l; i++)
    if (hipEventCreateWithFlags(&e.level[i], HIP_EVENT_DISABLE_TIMING)) mc_die("circuit bootstrap (event)");
  for (int i = 0; i < l * e.n_pieces; i++)
    if (hipEventCreateWithFlags(&e.piece[i], HIP_EVENT_DISABLE_TIMING)) mc_die("circuit bootstrap (event)");
  return e;
}
static void trgsw_levels_download(TRGSW *out, Torus *h_flat, const Torus *d_flat, int count, int l, size_t row, LevelEvents *e) {
  const int rows = 2 * l;
  const size_t item = (size_t)rows * row, pitch = sizeof(Torus) * item, width = sizeof(Torus) * row;
  void *cs = mc_copy_stream();
  for (int i = 0; i < l; i++) {
    if (hipStreamWaitEvent(cs, e->level[i], 0)) mc_die("circuit bootstrap (copy out)");
    for (int p = 0; p < e->n_pieces; p++) {
      const int lo = p * LEVEL_PIECE, cnt = count - lo < LEVEL_PIECE ? count - lo : LEVEL_PIECE;
      const size_t at = (size_t)lo * item;
      if (hipMemcpy2DAsync(h_flat + at + (size_t)i * row, pitch, d_flat + at + (size_t)i * row, pitch, width, (size_t)cnt, HIP_D2H, cs) ||
          hipMemcpy2DAsync(h_flat + at + (size_t)(l + i) * row, pitch, d_flat + at + (size_t)(l + i) * row, pitch, width, (size_t)cnt, HIP_D2H, cs) ||
          hipEventRecord(e->piece[i * e->n_pieces + p], cs))
        mc_die("circuit bootstrap (copy out)");
    }
  }
  for (int i = 0; i < l; i++)
    for (int p = 0; p < e->n_pieces; p++) {
      const int lo = p * LEVEL_PIECE, cnt = count - lo < LEVEL_PIECE ? count - lo : LEVEL_PIECE;
      if (hipEventSynchronize(e->piece[i * e->n_pieces + p])) mc_die("circuit bootstrap (copy out)");
      TrgswRows a = {out + lo, h_flat + (size_t)lo * item, rows, i, l + i, row};
      mc_parallel_for(trgsw_rows_from_flat_range, &a, cnt, 1);
    }
  for (int i = 0; i < l; i++) hipEventDestroy(e->level[i]);
  for (int i = 0; i < l * e->n_pieces; i++) hipEventDestroy(e->piece[i]);
  free(e->level);
  free(e->piece);
}

typedef struct { TRGSW *out; TLWE *in; Bootstrap_Key key; TRLWE_KS_Key *kska; Generic_KS_Key kskb; } Cb3Slice;
static void cb3_slice(void *pv, int lo, int hi) {
  Cb3Slice *a = (Cb3Slice *)pv;
  if (hi > lo) circuit_bootstrap_3_batch(a->out + lo, a->in + lo, hi - lo, a->key, a->kska, a->kskb);
}

void circuit_bootstrap_3_batch(TRGSW *out, TLWE *in, int count, Bootstrap_Key key, TRLWE_KS_Key *kska, Generic_KS_Key kskb) {
  if (MC_SHOULD_SHARD(count)) {   /* BASELINE configs[3]: bootstrap key, private FFT key-switch pair and the packing table replicated (src/bootstrap.c:346-366 per sample) */
    Cb3Slice a = {out, in, key, kska, kskb};
    void *keys[3] = {key->device, kska[0]->device, kskb->device};
    const int kinds[3] = {MC_KEY_BSK, MC_KEY_GAK, MC_KEY_KSK};
    mc_run_sharded(cb3_slice, &a, count, keys, kinds, 3);
    return;
  }
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int n = key->n, N = key->N, l = key->l;
  const size_t in_w = (size_t)count * (n + 1), row = (size_t)2 * N, out_w = (size_t)count * 2 * l * row;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + out_w));
  tlwe_array_to_flat(h, in, count, n);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * in_w, HIP_H2D);
  LevelEvents ev = level_events_new(l, count);
  if (mosfhet_hip_circuit_bootstrap_3_batch_ev(ctx, (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK), (mosfhet_hip_gak_t)mc_key_here(kska[0]->device, MC_KEY_GAK),
                                               (mosfhet_hip_ksk_t)mc_key_here(kskb->device, MC_KEY_KSK), d + in_w, d, count, NULL, ev.level))
    mc_die("circuit_bootstrap_3");
  trgsw_levels_download(out, h + in_w, d + in_w, count, l, row, &ev);
  if (mosfhet_hip_ctx_sync(ctx, NULL)) mc_die("circuit_bootstrap_3");
  stage_free(d);
  mc_hstage_free(h);
}

void circuit_bootstrap_3(TRGSW out, TLWE in, Bootstrap_Key key, TRLWE_KS_Key *kska, Generic_KS_Key kskb) {
  circuit_bootstrap_3_batch(&out, &in, 1, key, kska, kskb);
}

/* ------------------------------------------------------------------ callers either side of the bootstrap (GPU compositions) */
typedef struct { Torus *h, *d; size_t words; } Buf;
static Buf buf_new(size_t words) {
  Buf b;
  b.words = words;
  b.h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (words ? words : 1));
  b.d = (Torus *)mc_stage_alloc(sizeof(Torus) * (words ? words : 1));
  return b;
}
static void buf_up(Buf *b, size_t off, size_t words) { mc_dev_copy(b->d + off, b->h + off, sizeof(Torus) * words, HIP_H2D); }
static void buf_down(Buf *b, size_t off, size_t words) { mc_dev_copy(b->h + off, b->d + off, sizeof(Torus) * words, HIP_D2H); }
static void buf_free(Buf *b) { stage_free(b->d); mc_hstage_free(b->h); }
static mosfhet_hip_ctx_t ectx(void) { return (mosfhet_hip_ctx_t)mosfhet_engine_ctx(); }
static void check_rc(int rc, const char *what) { if (rc || mosfhet_hip_ctx_sync(ectx(), NULL)) mc_die(what); }

/* s[i][j][v-1] = TRLWE_out( -s_out * s_i v 2^(64-(j+1)bb) ), i <= n, s_n = -1 for the b word   [src/keyswitch.c:611-637] */
void mosfhet_gen_priv_sk_ks_key_flat(Torus *out, TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  const int N = out_key->s[0]->N, base = 1 << base_bit;
  TRLWE tmp = trlwe_alloc_new_sample(1, N);
  for (int i = 0; i <= in_key->n; i++) {
    const Torus s_i = i < in_key->n ? in_key->s[i] : (Torus)-1;
    for (int j = 0; j < t; j++)
      for (int v = 1; v < base; v++) {
        const Torus dec_key = s_i * (Torus)v * ((Torus)1 << (W - (j + 1) * base_bit));
        trlwe_sample(tmp, NULL, out_key);
        for (int e = 0; e < N; e++) tmp->b->coeffs[e] += ((Torus)0 - out_key->s[0]->coeffs[e]) * dec_key;
        mc_trlwe_to_flat(out + ((((size_t)i * t + j) * (base - 1)) + (v - 1)) * 2 * N, tmp);
      }
  }
  free_trlwe(tmp);
}

Generic_KS_Key trlwe_new_priv_SK_KS_key_N2(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  return table_key_new(1, out_key, in_key, t, base_bit, "trlwe_new_priv_SK_KS_key_N2");
}

void trlwe_priv_keyswitch(TRLWE out, TLWE in, Generic_KS_Key ks) {
  const int N = out->b->N, n = ks->n;
  Buf b = buf_new((size_t)n + 1 + 2 * N);
  tlwe_array_to_flat(b.h, &in, 1, n);
  buf_up(&b, 0, (size_t)n + 1);
  check_rc(mosfhet_hip_trlwe_priv_keyswitch_batch(ectx(), (mosfhet_hip_ksk_t)ks->device, b.d + n + 1, b.d, 1, NULL), "trlwe_priv_keyswitch");
  buf_down(&b, (size_t)n + 1, (size_t)2 * N);
  mc_trlwe_from_flat(out, b.h + n + 1);
  buf_free(&b);
}

typedef struct { TRGSW *out; TLWE *in; Bootstrap_Key key; Generic_KS_Key kska, kskb; int variant; } CbSlice;
static void circuit_bootstrap_many(TRGSW *out, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb, int variant);
static void cb_slice(void *pv, int lo, int hi) {
  CbSlice *a = (CbSlice *)pv;
  if (hi > lo) circuit_bootstrap_many(a->out + lo, a->in + lo, hi - lo, a->key, a->kska, a->kskb, a->variant);
}

static void circuit_bootstrap_many(TRGSW *out, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb, int variant) {
  if (MC_SHOULD_SHARD(count)) {   /* src/bootstrap.c:309-344 per sample: bootstrap key and both table keys replicated */
    CbSlice a = {out, in, key, kska, kskb, variant};
    void *keys[3] = {key->device, kska->device, kskb->device};
    const int kinds[3] = {MC_KEY_BSK, MC_KEY_KSK, MC_KEY_KSK};
    mc_run_sharded(cb_slice, &a, count, keys, kinds, 3);
    return;
  }
  const int n = key->n, N = key->N, l = key->l;
  const size_t in_w = (size_t)count * (n + 1), row = (size_t)2 * N, out_w = (size_t)count * 2 * l * row;
  Buf b = buf_new(in_w + out_w);
  tlwe_array_to_flat(b.h, in, count, n);
  buf_up(&b, 0, in_w);
  LevelEvents ev = level_events_new(l, count);
  if (mosfhet_hip_circuit_bootstrap_batch_ev(ectx(), (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK), (mosfhet_hip_ksk_t)mc_key_here(kska->device, MC_KEY_KSK),
                                             (mosfhet_hip_ksk_t)mc_key_here(kskb->device, MC_KEY_KSK), b.d + in_w, b.d, count, variant, NULL, ev.level))
    mc_die("circuit_bootstrap");
  trgsw_levels_download(out, b.h + in_w, b.d + in_w, count, l, row, &ev);
  check_rc(0, "circuit_bootstrap");
  buf_free(&b);
}

void circuit_bootstrap(TRGSW out, TLWE in, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb) { circuit_bootstrap_many(&out, &in, 1, key, kska, kskb, 0); }
void circuit_bootstrap_2(TRGSW out, TLWE in, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb) { circuit_bootstrap_many(&out, &in, 1, key, kska, kskb, 1); }
void circuit_bootstrap_2_batch(TRGSW *out, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb) {
  circuit_bootstrap_many(out, in, count, key, kska, kskb, 1);
}

typedef struct { TLWE *out; TorusPolynomial tv; TLWE *in; Bootstrap_Key key; Generic_KS_Key ksk; int torus_base, variant; } Ks21Slice;
static void fdfb_ks21_many(TLWE *out, TorusPolynomial tv, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base, int variant);
static void ks21_slice(void *pv, int lo, int hi) {
  Ks21Slice *a = (Ks21Slice *)pv;
  if (hi > lo) fdfb_ks21_many(a->out + lo, a->tv, a->in + lo, hi - lo, a->key, a->ksk, a->torus_base, a->variant);
}

static void fdfb_ks21_many(TLWE *out, TorusPolynomial tv, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base, int variant) {
  if (MC_SHOULD_SHARD(count)) {   /* src/bootstrap.c:391-463 per sample */
    Ks21Slice a = {out, tv, in, key, ksk, torus_base, variant};
    void *keys[2] = {key->device, ksk->device};
    const int kinds[2] = {MC_KEY_BSK, MC_KEY_KSK};
    mc_run_sharded(ks21_slice, &a, count, keys, kinds, 2);
    return;
  }
  const int n = key->n, N = key->N;
  const size_t in_w = (size_t)count * (n + 1), tv_w = (size_t)2 * N, out_w = (size_t)count * (N + 1);
  if (tv->N != 2 * N) { fprintf(stderr, "mosfhet_amd: full_domain_functional_bootstrap_KS21: tv must have 2N coefficients\n"); abort(); }
  Buf b = buf_new(in_w + tv_w + out_w);
  tlwe_array_to_flat(b.h, in, count, n);
  memcpy(b.h + in_w, tv->coeffs, sizeof(Torus) * tv_w);
  buf_up(&b, 0, in_w + tv_w);
  check_rc(mosfhet_hip_full_domain_functional_bootstrap_KS21_batch(ectx(), (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK), (mosfhet_hip_ksk_t)mc_key_here(ksk->device, MC_KEY_KSK),
                                                                   b.d + in_w + tv_w, b.d + in_w, b.d, count, torus_base, variant, NULL), "full_domain_functional_bootstrap_KS21");
  buf_down(&b, in_w + tv_w, out_w);
  tlwe_array_from_flat(out, b.h + in_w + tv_w, count, N);
  buf_free(&b);
}

void full_domain_functional_bootstrap_KS21(TLWE out, TorusPolynomial tv, TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base) {
  fdfb_ks21_many(&out, tv, &in, 1, key, ksk, torus_base, 0);
}
void full_domain_functional_bootstrap_KS21_2(TLWE out, TorusPolynomial tv, TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base) {
  fdfb_ks21_many(&out, tv, &in, 1, key, ksk, torus_base, 1);
}
void full_domain_functional_bootstrap_KS21_batch(TLWE *out, TorusPolynomial tv, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base) {
  fdfb_ks21_many(out, tv, in, count, key, ksk, torus_base, 0);
}

TRLWE_KS_Key trlwe_new_RL_key(TRLWE_Key key, int t, int base_bit) {
  const int N = key->s[0]->N;
  if (key->k != 1) { fprintf(stderr, "mosfhet_amd: trlwe_new_RL_key: k = 1 only\n"); abort(); }
  Torus *s2 = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)N);
  memset(s2, 0, sizeof(Torus) * (size_t)N);
  negacyclic_mul_addto(s2, key->s[0]->coeffs, key->s[0]->coeffs, N);   /* s^2 (src/keyswitch.c:6) */
  mosfhet_hip_gak_t dev = fft_ks_keys_new(key, s2, 1, t, base_bit, "trlwe_new_RL_key");
  free(s2);
  return trlwe_ks_header(dev, 0, 1, t, base_bit);
}

void trlwe_tensor_prod_FFT(TRLWE out, TRLWE in1, TRLWE in2, int precision, TRLWE_KS_Key rl_key) {
  const int N = in1->b->N;
  const size_t row = (size_t)2 * N;
  Buf b = buf_new(3 * row);
  mc_trlwe_to_flat(b.h, in1);
  mc_trlwe_to_flat(b.h + row, in2);
  buf_up(&b, 0, 2 * row);
  check_rc(mosfhet_hip_trlwe_tensor_prod_FFT_batch(ectx(), (mosfhet_hip_gak_t)rl_key->device, b.d + 2 * row, b.d, b.d + row, precision, 1, NULL), "trlwe_tensor_prod_FFT");
  buf_down(&b, 2 * row, row);
  mc_trlwe_from_flat(out, b.h + 2 * row);
  buf_free(&b);
}

void tlwe_mul(TLWE out, TLWE in1, TLWE in2, int precision, Generic_KS_Key ksk, TRLWE_KS_Key rlk) {
  const int N = in1->n;
  const size_t w = (size_t)N + 1;
  Buf b = buf_new(3 * w);
  tlwe_array_to_flat(b.h, &in1, 1, N);
  tlwe_array_to_flat(b.h + w, &in2, 1, N);
  buf_up(&b, 0, 2 * w);
  check_rc(mosfhet_hip_tlwe_mul_batch(ectx(), (mosfhet_hip_ksk_t)ksk->device, (mosfhet_hip_gak_t)rlk->device, b.d + 2 * w, b.d, b.d + w, precision, 1, NULL), "tlwe_mul");
  buf_down(&b, 2 * w, w);
  tlwe_array_from_flat(&out, b.h + 2 * w, 1, N);
  buf_free(&b);
}

typedef struct { TLWE *out; const Torus *tv_flat; size_t tv_w; TLWE *in; Bootstrap_Key key; Generic_KS_Key ksk; TRLWE_KS_Key rlk; int precision, variant; } Clot21Slice;
static void fdfb_clot21_many(TLWE *out, const Torus *tv_flat, size_t tv_w, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision,
                             int variant);
static void clot21_slice(void *pv, int lo, int hi) {
  Clot21Slice *a = (Clot21Slice *)pv;
  if (hi > lo) fdfb_clot21_many(a->out + lo, a->tv_flat, a->tv_w, a->in + lo, hi - lo, a->key, a->ksk, a->rlk, a->precision, a->variant);
}

static void fdfb_clot21_many(TLWE *out, const Torus *tv_flat, size_t tv_w, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision,
                             int variant) {
  if (MC_SHOULD_SHARD(count)) {   /* src/bootstrap.c:465-517 per sample: bootstrap key, packing table and relinearisation key replicated */
    Clot21Slice a = {out, tv_flat, tv_w, in, key, ksk, rlk, precision, variant};
    void *keys[3] = {key->device, ksk->device, rlk->device};
    const int kinds[3] = {MC_KEY_BSK, MC_KEY_KSK, MC_KEY_GAK};
    mc_run_sharded(clot21_slice, &a, count, keys, kinds, 3);
    return;
  }
  const int n = key->n, N = key->N;
  const size_t in_w = (size_t)count * (n + 1), out_w = (size_t)count * (N + 1);
  Buf b = buf_new(in_w + tv_w + out_w);
  tlwe_array_to_flat(b.h, in, count, n);
  memcpy(b.h + in_w, tv_flat, sizeof(Torus) * tv_w);
  buf_up(&b, 0, in_w + tv_w);
  check_rc(mosfhet_hip_full_domain_functional_bootstrap_CLOT21_batch(ectx(), (mosfhet_hip_bsk_t)mc_key_here(key->device, MC_KEY_BSK), (mosfhet_hip_ksk_t)mc_key_here(ksk->device, MC_KEY_KSK),
                                                                     (mosfhet_hip_gak_t)mc_key_here(rlk->device, MC_KEY_GAK), b.d + in_w + tv_w, b.d + in_w, b.d, count, precision, variant, NULL),
           "full_domain_functional_bootstrap_CLOT21");
  buf_down(&b, in_w + tv_w, out_w);
  tlwe_array_from_flat(out, b.h + in_w + tv_w, count, N);
  buf_free(&b);
}

void full_domain_functional_bootstrap_CLOT21(TLWE out, TRLWE tv[2], TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision) {
  const size_t row = (size_t)2 * key->N;
  Torus *flat = (Torus *)mc_xmalloc(sizeof(Torus) * 2 * row);
  mc_trlwe_to_flat(flat, tv[0]);
  mc_trlwe_to_flat(flat + row, tv[1]);
  fdfb_clot21_many(&out, flat, 2 * row, &in, 1, key, ksk, rlk, precision, 0);
  free(flat);
}

void full_domain_functional_bootstrap_CLOT21_2(TLWE out, Torus *tv, TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision) {
  fdfb_clot21_many(&out, tv, (size_t)1 << (precision - 1), &in, 1, key, ksk, rlk, precision, 1);
}

void full_domain_functional_bootstrap_CLOT21_2_batch(TLWE *out, Torus *tv, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision) {
  fdfb_clot21_many(out, tv, (size_t)1 << (precision - 1), in, count, key, ksk, rlk, precision, 1);
}

void multivalue_bootstrap_phase1(TRLWE *out, TLWE in, Bootstrap_Key key, int torus_base) {
  const int n = key->n, N = key->N;
  const size_t row = (size_t)2 * N, out_w = (size_t)(torus_base + 1) * row;
  Buf b = buf_new((size_t)n + 1 + out_w);
  tlwe_array_to_flat(b.h, &in, 1, n);
  buf_up(&b, 0, (size_t)n + 1);
  check_rc(mosfhet_hip_multivalue_bootstrap_phase1_batch(ectx(), (mosfhet_hip_bsk_t)key->device, b.d + n + 1, b.d, 1, torus_base, NULL), "multivalue_bootstrap_phase1");
  buf_down(&b, (size_t)n + 1, out_w);
  for (int i = 0; i <= torus_base; i++) mc_trlwe_from_flat(out[i], b.h + n + 1 + i * row);
  buf_free(&b);
}

void multivalue_bootstrap_phase2(TLWE out, int *in, TRLWE *rotated_tv, int torus_base, int log_torus_base) {
  const int N = rotated_tv[0]->b->N;
  const size_t row = (size_t)2 * N, in_w = (size_t)(torus_base + 1) * row;
  Buf b = buf_new(in_w + N + 1);
  for (int i = 0; i <= torus_base; i++) mc_trlwe_to_flat(b.h + i * row, rotated_tv[i]);
  buf_up(&b, 0, in_w);
  check_rc(mosfhet_hip_multivalue_bootstrap_phase2_batch(ectx(), b.d + in_w, in, b.d, N, torus_base, log_torus_base, 1, NULL), "multivalue_bootstrap_phase2");
  buf_down(&b, in_w, (size_t)N + 1);
  tlwe_array_from_flat(&out, b.h + in_w, 1, N);
  buf_free(&b);
}

/* ------------------------------------------------------------------ on-disk formats
 * Torus-domain objects (secret keys, samples, the LWE key-switch table, the table-lookup TRLWE keys) are written byte for byte as the reference
 * writes them (uncompressed rows: the reference's PORTABLE / A_PRNG=none shape of _MACRO_trlwe_save_sample, src/keyswitch.c:236-240).  DFT-domain
 * keys keep the reference's integer header and then hold the engine's own image, preceded by its layout tag -- the reference's DFT contents are
 * backend-defined too (src/polynomial.c:336-357), so such files were never portable between builds.  A short read aborts (the reference ignores
 * fread's result and goes on with garbage). */
static void xwrite(const void *p, size_t size, size_t count, FILE *fd) {
  if (fwrite(p, size, count, fd) != count) { perror("mosfhet_amd: write failed"); abort(); }
}
static void xread(void *p, size_t size, size_t count, FILE *fd) {
  if (fread(p, size, count, fd) != count) { fprintf(stderr, "mosfhet_amd: key / sample file is truncated\n"); abort(); }
}
static void read_ints(FILE *fd, int *v, int count) { xread(v, sizeof(int), (size_t)count, fd); }
static void check_layout_tag(FILE *fd, const char *who) {
  unsigned tag = 0;
  xread(&tag, sizeof(tag), 1, fd);
  if (tag != mosfhet_hip_dft_layout_id()) {
    fprintf(stderr, "mosfhet_amd: %s: file holds DFT layout %08x, this engine uses %08x (files with DFT-domain keys are not portable between FFT back-ends)\n",
            who, tag, mosfhet_hip_dft_layout_id());
    abort();
  }
}

void tlwe_save_sample(FILE *fd, TLWE c) { xwrite(c->a, sizeof(Torus), (size_t)c->n, fd); xwrite(&c->b, sizeof(Torus), 1, fd); }
void tlwe_load_sample(FILE *fd, TLWE c) { xread(c->a, sizeof(Torus), (size_t)c->n, fd); xread(&c->b, sizeof(Torus), 1, fd); }
TLWE tlwe_load_new_sample(FILE *fd, int n) { TLWE c = tlwe_alloc_sample(n); tlwe_load_sample(fd, c); return c; }

void tlwe_save_key(FILE *fd, TLWE_Key key) {
  xwrite(&key->n, sizeof(int), 1, fd);
  xwrite(&key->sigma, sizeof(double), 1, fd);
  xwrite(key->s, sizeof(Torus), (size_t)key->n, fd);
}
TLWE_Key tlwe_load_new_key(FILE *fd) {
  int n; double sigma;
  read_ints(fd, &n, 1);
  xread(&sigma, sizeof(double), 1, fd);
  TLWE_Key key = tlwe_alloc_key(n, sigma);
  xread(key->s, sizeof(Torus), (size_t)n, fd);
  return key;
}

void trlwe_save_sample(FILE *fd, TRLWE c) {
  for (int i = 0; i < c->k; i++) xwrite(c->a[i]->coeffs, sizeof(Torus), (size_t)c->b->N, fd);
  xwrite(c->b->coeffs, sizeof(Torus), (size_t)c->b->N, fd);
}
void trlwe_load_sample(FILE *fd, TRLWE c) {
  for (int i = 0; i < c->k; i++) xread(c->a[i]->coeffs, sizeof(Torus), (size_t)c->b->N, fd);
  xread(c->b->coeffs, sizeof(Torus), (size_t)c->b->N, fd);
}
TRLWE trlwe_load_new_sample(FILE *fd, int k, int N) { TRLWE c = trlwe_alloc_new_sample(k, N); trlwe_load_sample(fd, c); return c; }

void trlwe_save_key(FILE *fd, TRLWE_Key key) {
  xwrite(&key->k, sizeof(int), 1, fd);
  xwrite(&key->s[0]->N, sizeof(int), 1, fd);
  xwrite(&key->sigma, sizeof(double), 1, fd);
  for (int i = 0; i < key->k; i++) xwrite(key->s[i]->coeffs, sizeof(Torus), (size_t)key->s[0]->N, fd);
}
TRLWE_Key trlwe_load_new_key(FILE *fd) {
  int kN[2]; double sigma;
  read_ints(fd, kN, 2);
  xread(&sigma, sizeof(double), 1, fd);
  TRLWE_Key key = trlwe_alloc_key(kN[1], kN[0], sigma);
  for (int i = 0; i < key->k; i++) xread(key->s[i]->coeffs, sizeof(Torus), (size_t)kN[1], fd);
  return key;
}

void trgsw_save_key(FILE *fd, TRGSW_Key key) {
  xwrite(&key->l, sizeof(int), 1, fd);
  xwrite(&key->Bg_bit, sizeof(int), 1, fd);
  trlwe_save_key(fd, key->trlwe_key);
}
TRGSW_Key trgsw_load_new_key(FILE *fd) {
  int v[2];
  read_ints(fd, v, 2);
  return trgsw_new_key(trlwe_load_new_key(fd), v[0], v[1]);
}

void trgsw_save_sample(FILE *fd, TRGSW c) {
  for (int i = 0; i < c->l * (c->samples[0]->k + 1); i++) trlwe_save_sample(fd, c->samples[i]);
}
void trgsw_load_sample(FILE *fd, TRGSW c) {
  for (int i = 0; i < c->l * (c->samples[0]->k + 1); i++) trlwe_load_sample(fd, c->samples[i]);
}
TRGSW trgsw_load_new_sample(FILE *fd, int l, int Bg_bit, int k, int N) {
  TRGSW c = trgsw_alloc_new_sample(l, Bg_bit, k, N);
  trgsw_load_sample(fd, c);
  return c;
}

/* LWE key-switch key: header n, t, base_bit, n_out, then every row a[n_out], b -- the flat table as it is (src/tlwe.c:247-287) */
void tlwe_save_KS_key(FILE *fd, TLWE_KS_Key key) {
  const int n_out = key->s[0][0][0]->n;
  xwrite(&key->n, sizeof(int), 1, fd);
  xwrite(&key->t, sizeof(int), 1, fd);
  xwrite(&key->base_bit, sizeof(int), 1, fd);
  xwrite(&n_out, sizeof(int), 1, fd);
  xwrite(key->s[0][0][0]->a, sizeof(Torus), (size_t)key->n * key->t * ((1 << key->base_bit) - 1) * ((size_t)n_out + 1), fd);
}
TLWE_KS_Key tlwe_load_new_KS_key(FILE *fd) {
  int v[4];  /* n, t, base_bit, n_out */
  read_ints(fd, v, 4);
  if (v[0] < 1 || v[1] < 1 || v[2] < 1 || v[2] > 8 || v[3] < 1) { fprintf(stderr, "mosfhet_amd: tlwe_load_new_KS_key: bad header\n"); abort(); }
  const size_t words = (size_t)v[0] * v[1] * ((1 << v[2]) - 1) * ((size_t)v[3] + 1);
  Torus *flat = (Torus *)mc_xmalloc(sizeof(Torus) * words);
  xread(flat, sizeof(Torus), words, fd);
  return tlwe_ks_wrap(flat, v[0], v[3], v[1], v[2], "tlwe_load_new_KS_key");
}

/* table-lookup TRLWE keys (packing / private): header base_bit, t, n, k, N, include_b, then (n + include_b) t (2^bb - 1) rows a[N], b[N]
 * (src/keyswitch.c:409-455, uncompressed rows), streamed through a bounded host buffer: the config-4 packing key is 6 GB */
#define KS_IO_CHUNK_BYTES ((size_t)64 << 20)
void trlwe_save_generic_ks_key(FILE *fd, Generic_KS_Key key) {
  int info[6];
  if (mosfhet_hip_ksk_info((mosfhet_hip_ksk_t)key->device, info)) mc_die("trlwe_save_generic_ks_key");
  const int N = info[1] / 2, k = 1;
  xwrite(&key->base_bit, sizeof(int), 1, fd);
  xwrite(&key->t, sizeof(int), 1, fd);
  xwrite(&key->n, sizeof(int), 1, fd);
  xwrite(&k, sizeof(int), 1, fd);
  xwrite(&N, sizeof(int), 1, fd);
  xwrite(&key->include_b, sizeof(int), 1, fd);
  const size_t rows = (size_t)info[0] * key->t * ((1 << key->base_bit) - 1), row_bytes = (size_t)info[1] * sizeof(Torus);
  size_t chunk = KS_IO_CHUNK_BYTES / row_bytes;
  if (chunk < 1) chunk = 1;
  Torus *buf = (Torus *)mc_xmalloc(chunk * row_bytes);
  for (size_t r = 0; r < rows; r += chunk) {
    const size_t c = rows - r < chunk ? rows - r : chunk;
    if (mosfhet_hip_ksk_export_rows((mosfhet_hip_ksk_t)key->device, r, c, buf)) mc_die("trlwe_save_generic_ks_key");
    xwrite(buf, row_bytes, c, fd);
  }
  free(buf);
}
Generic_KS_Key trlwe_load_new_generic_ks_key(FILE *fd) {
  int v[6];  /* base_bit, t, n, k, N, include_b */
  read_ints(fd, v, 6);
  if (v[3] != 1 || v[5] < 0 || v[5] > 1) { fprintf(stderr, "mosfhet_amd: trlwe_load_new_generic_ks_key: k = 1 keys only\n"); abort(); }
  Generic_KS_Key res = (Generic_KS_Key)mc_xmalloc(sizeof(*res));
  res->s = NULL; res->base_bit = v[0]; res->t = v[1]; res->n = v[2]; res->include_b = v[5];
  mosfhet_hip_ksk_t dev = NULL;
  if (mosfhet_hip_ksk_alloc((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, v[5] ? 2 : 1, v[2], v[4], v[1], v[0])) mc_die("trlwe_load_new_generic_ks_key");
  const size_t rows = (size_t)(v[2] + v[5]) * v[1] * ((1 << v[0]) - 1), row_bytes = (size_t)2 * v[4] * sizeof(Torus);
  size_t chunk = KS_IO_CHUNK_BYTES / row_bytes;
  if (chunk < 1) chunk = 1;
  Torus *buf = (Torus *)mc_xmalloc(chunk * row_bytes);
  for (size_t r = 0; r < rows; r += chunk) {
    const size_t c = rows - r < chunk ? rows - r : chunk;
    xread(buf, row_bytes, c, fd);
    if (mosfhet_hip_ksk_import_rows(dev, r, c, buf)) mc_die("trlwe_load_new_generic_ks_key");
  }
  free(buf);
  res->device = dev;
  return res;
}

/* FFT-based TRLWE key-switch key: header base_bit, t, k_in, k, N (src/keyswitch.c:122-160), layout tag, image of this key's t rows */
void trlwe_save_KS_key(FILE *fd, TRLWE_KS_Key key) {
  int info[4];
  if (mosfhet_hip_trlwe_ksk_info((mosfhet_hip_gak_t)key->device, info)) mc_die("trlwe_save_KS_key");
  const int one = 1, N = info[1];
  const unsigned tag = mosfhet_hip_dft_layout_id();
  xwrite(&key->base_bit, sizeof(int), 1, fd);
  xwrite(&key->t, sizeof(int), 1, fd);
  xwrite(&one, sizeof(int), 1, fd);
  xwrite(&one, sizeof(int), 1, fd);
  xwrite(&N, sizeof(int), 1, fd);
  xwrite(&tag, sizeof(tag), 1, fd);
  const size_t bytes = mosfhet_hip_trlwe_ksk_bytes((mosfhet_hip_gak_t)key->device), entry_bytes = bytes / (size_t)info[0];
  char *img = (char *)mc_xmalloc(bytes);
  if (mosfhet_hip_trlwe_ksk_export((mosfhet_hip_gak_t)key->device, img)) mc_die("trlwe_save_KS_key");
  xwrite(img + (size_t)key->entry * entry_bytes, 1, entry_bytes, fd);
  free(img);
}
TRLWE_KS_Key trlwe_load_new_KS_key(FILE *fd) {
  int v[5];  /* base_bit, t, k_in, k, N */
  read_ints(fd, v, 5);
  if (v[2] != 1 || v[3] != 1) { fprintf(stderr, "mosfhet_amd: trlwe_load_new_KS_key: k = 1 keys only\n"); abort(); }
  check_layout_tag(fd, "trlwe_load_new_KS_key");
  const size_t bytes = (size_t)v[1] * 2 * v[4] * sizeof(double);
  char *img = (char *)mc_xmalloc(bytes);
  xread(img, 1, bytes, fd);
  mosfhet_hip_gak_t dev = NULL;
  if (mosfhet_hip_trlwe_ksk_import((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, img, 1, v[4], v[1], v[0])) mc_die("trlwe_load_new_KS_key");
  free(img);
  return trlwe_ks_header(dev, 0, 1, v[1], v[0]);
}

/* bootstrap key: header n, l, k, N, Bg_bit, unfolding (src/bootstrap.c:63-104), layout tag, image (DFT rows; torus-domain samples when unfolded) */
void save_bootstrap_key(FILE *fd, Bootstrap_Key key) {
  mosfhet_hip_bsk_t dev = (mosfhet_hip_bsk_t)mosfhet_bootstrap_key_device(key);
  const unsigned tag = mosfhet_hip_dft_layout_id();
  xwrite(&key->n, sizeof(int), 1, fd);
  xwrite(&key->l, sizeof(int), 1, fd);
  xwrite(&key->k, sizeof(int), 1, fd);
  xwrite(&key->N, sizeof(int), 1, fd);
  xwrite(&key->Bg_bit, sizeof(int), 1, fd);
  xwrite(&key->unfolding, sizeof(int), 1, fd);
  xwrite(&tag, sizeof(tag), 1, fd);
  const size_t bytes = mosfhet_hip_bsk_bytes(dev);
  char *img = (char *)mc_xmalloc(bytes);
  if (mosfhet_hip_bsk_export(dev, img)) mc_die("save_bootstrap_key");
  xwrite(img, 1, bytes, fd);
  free(img);
}
Bootstrap_Key load_new_bootstrap_key(FILE *fd) {
  int v[6];  /* n, l, k, N, Bg_bit, unfolding */
  read_ints(fd, v, 6);
  check_layout_tag(fd, "load_new_bootstrap_key");
  if (v[0] < 1 || v[1] < 1 || v[2] < 1 || v[3] < 2 || v[5] < 1 || v[5] > 8) { fprintf(stderr, "mosfhet_amd: load_new_bootstrap_key: bad header\n"); abort(); }
  const size_t bytes = v[5] == 1 ? (size_t)v[0] * (v[2] + 1) * v[1] * (v[2] + 1) * v[3] * sizeof(double)
                                 : (size_t)v[0] * ((size_t)1 << v[5]) / v[5] * 2 * v[1] * 2 * v[3] * sizeof(Torus);
  char *img = (char *)mc_xmalloc(bytes);
  xread(img, 1, bytes, fd);
  Bootstrap_Key res = (Bootstrap_Key)mc_xmalloc(sizeof(*res));
  res->n = v[0]; res->l = v[1]; res->k = v[2]; res->N = v[3]; res->Bg_bit = v[4]; res->unfolding = v[5];
  res->su = NULL;
  mosfhet_hip_bsk_t dev = NULL;
  if (mosfhet_hip_bsk_import((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, img, v[0], v[2], v[3], v[1], v[4], v[5])) mc_die("load_new_bootstrap_key");
  free(img);
  res->device = dev;
  res->s = key_views(dev, res->n, res->l, res->Bg_bit, res->N);
  remember_key(res);
  return res;
}

/* ------------------------------------------------------------------ LUT-packing key switch (src/keyswitch.c:214-366): torus_base LWE samples into the slots of
 * one TRLWE sample -- what the reference's radix-integer application builds its encrypted lookup tables with (applications/multi-ciphertext-arith/src/
 * lut.c).  Device-resident table key like the others: `s` is NULL. */
LUT_Packing_KS_Key trlwe_new_packing_KS_key(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit, int torus_base) {
  if (out_key->k != 1) { fprintf(stderr, "mosfhet_amd: trlwe_new_packing_KS_key: k = 1 only\n"); abort(); }
  LUT_Packing_KS_Key res = (LUT_Packing_KS_Key)mc_xmalloc(sizeof(*res));
  res->s = NULL; res->base_bit = base_bit; res->t = t; res->torus_base = torus_base; res->n = in_key->n;
  mosfhet_hip_ksk_t dev = NULL;
  if (mosfhet_hip_trlwe_lut_packing_ksk_generate((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, out_key->s[0]->coeffs, out_key->s[0]->N, in_key->s, in_key->n, t, base_bit,
                                                 torus_base, out_key->sigma, mc_rnd64()))
    mc_die("trlwe_new_packing_KS_key");
  res->device = dev;
  return res;
}

void free_trlwe_packing_ks_key(LUT_Packing_KS_Key key) {
  if (!key) return;
  mosfhet_hip_ksk_destroy((mosfhet_hip_ksk_t)key->device);
  free(key);
}

void trlwe_packing_keyswitch(TRLWE out, TLWE *in, LUT_Packing_KS_Key ks) {
  mosfhet_hip_ctx_t ctx = (mosfhet_hip_ctx_t)mosfhet_engine_ctx();
  const int N = out->b->N, n = ks->n, tb = ks->torus_base;
  const size_t in_w = (size_t)tb * (n + 1), out_w = (size_t)2 * N;
  Torus *h = (Torus *)mc_hstage_alloc(sizeof(Torus) * (in_w + out_w));
  tlwe_array_to_flat(h, in, tb, n);
  Torus *d = (Torus *)mc_stage_alloc(sizeof(Torus) * (in_w + out_w));
  mc_dev_copy(d, h, sizeof(Torus) * in_w, HIP_H2D);
  if (mosfhet_hip_trlwe_lut_packing_keyswitch_batch(ctx, (mosfhet_hip_ksk_t)ks->device, tb, d + in_w, d, 1, NULL) || mosfhet_hip_ctx_sync(ctx, NULL))
    mc_die("trlwe_packing_keyswitch");
  mc_dev_copy(h + in_w, d + in_w, sizeof(Torus) * out_w, HIP_D2H);
  mc_trlwe_from_flat(out, h + in_w);
  stage_free(d);
  mc_hstage_free(h);
}

/* file = the reference's: base_bit, t, torus_base, n, k, N, then the n * torus_base * t * (2^base_bit - 1) TRLWE samples in key order (uncompressed rows) */
void trlwe_save_packing_KS_key(FILE *fd, LUT_Packing_KS_Key key) {
  int info[6];
  if (mosfhet_hip_ksk_info((mosfhet_hip_ksk_t)key->device, info)) mc_die("trlwe_save_packing_KS_key");
  const int N = info[1] / 2, k = 1;
  xwrite(&key->base_bit, sizeof(int), 1, fd);
  xwrite(&key->t, sizeof(int), 1, fd);
  xwrite(&key->torus_base, sizeof(int), 1, fd);
  xwrite(&key->n, sizeof(int), 1, fd);
  xwrite(&k, sizeof(int), 1, fd);
  xwrite(&N, sizeof(int), 1, fd);
  const size_t rows = (size_t)info[0] * key->t * ((1 << key->base_bit) - 1), row_bytes = (size_t)info[1] * sizeof(Torus);
  size_t chunk = KS_IO_CHUNK_BYTES / row_bytes;
  if (chunk < 1) chunk = 1;
  Torus *buf = (Torus *)mc_xmalloc(chunk * row_bytes);
  for (size_t r = 0; r < rows; r += chunk) {
    const size_t c = rows - r < chunk ? rows - r : chunk;
    if (mosfhet_hip_ksk_export_rows((mosfhet_hip_ksk_t)key->device, r, c, buf)) mc_die("trlwe_save_packing_KS_key");
    xwrite(buf, row_bytes, c, fd);
  }
  free(buf);
}

LUT_Packing_KS_Key trlwe_load_new_packing_KS_key(FILE *fd) {
  int v[6];  /* base_bit, t, torus_base, n, k, N */
  read_ints(fd, v, 6);
  if (v[4] != 1 || v[2] < 1 || v[3] < 1) { fprintf(stderr, "mosfhet_amd: trlwe_load_new_packing_KS_key: k = 1 keys only\n"); abort(); }
  LUT_Packing_KS_Key res = (LUT_Packing_KS_Key)mc_xmalloc(sizeof(*res));
  res->s = NULL; res->base_bit = v[0]; res->t = v[1]; res->torus_base = v[2]; res->n = v[3];
  mosfhet_hip_ksk_t dev = NULL;   /* a table with n * torus_base digit sources and no b word: ksk kind 2 counts one source more than its `n` argument */
  if (mosfhet_hip_ksk_alloc((mosfhet_hip_ctx_t)mosfhet_engine_ctx(), &dev, 2, v[3] * v[2] - 1, v[5], v[1], v[0])) mc_die("trlwe_load_new_packing_KS_key");
  const size_t rows = (size_t)v[3] * v[2] * v[1] * ((1 << v[0]) - 1), row_bytes = (size_t)2 * v[5] * sizeof(Torus);
  size_t chunk = KS_IO_CHUNK_BYTES / row_bytes;
  if (chunk < 1) chunk = 1;
  Torus *buf = (Torus *)mc_xmalloc(chunk * row_bytes);
  for (size_t r = 0; r < rows; r += chunk) {
    const size_t c = rows - r < chunk ? rows - r : chunk;
    xread(buf, row_bytes, c, fd);
    if (mosfhet_hip_ksk_import_rows(dev, r, c, buf)) mc_die("trlwe_load_new_packing_KS_key");
  }
  free(buf);
  res->device = dev;
  return res;
}

/* ------------------------------------------------------------------ packing key switches built on the FFT key switch (src/keyswitch.c:98-106,195-227,476-546): needed
 * by the reference's test-suite only; compositions of trlwe_keyswitch calls, one device key set behind the reference's key shapes */
TRLWE_KS_Key trlwe_new_full_packing_KS_key(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  /* entry i switches the constant polynomial s_i: KS[i][j] = TRLWE(s_i 2^(64 - (j+1) base_bit)) */
  const int N = out_key->s[0]->N, n = in_key->n;
  if (out_key->k != 1) { fprintf(stderr, "mosfhet_amd: trlwe_new_full_packing_KS_key: k = 1 only\n"); abort(); }
  Torus *msgs = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)n * N);
  memset(msgs, 0, sizeof(Torus) * (size_t)n * N);
  for (int i = 0; i < n; i++) msgs[(size_t)i * N] = in_key->s[i];
  mosfhet_hip_gak_t dev = fft_ks_keys_new(out_key, msgs, n, t, base_bit, "trlwe_new_full_packing_KS_key");
  free(msgs);
  TRLWE_KS_Key res = trlwe_ks_header(dev, 0, 1, t, base_bit);
  res->k = n;   /* the reference's key has k = n input "polynomials" (mosfhet.h:90-93) */
  return res;
}

/* out = (0, sum_j in[j].b X^j) - sum_i KeySwitch_i(a_i(X)),  a_i(X) = sum_{j < size} in[j].a[i] X^j */
void trlwe_full_packing_keyswitch(TRLWE out, TLWE *in, uint64_t size, TRLWE_KS_Key ks_key) {
  const int N = out->b->N, n = ks_key->k;
  TRLWE ai = trlwe_alloc_new_sample(1, N), part = trlwe_alloc_new_sample(1, N);
  TRLWE_KS_Key entry = trlwe_ks_header(ks_key->device, 0, 0, ks_key->t, ks_key->base_bit);
  trlwe_noiseless_trivial_sample(out, NULL);
  for (int i = 0; i < n; i++) {
    trlwe_noiseless_trivial_sample(ai, NULL);
    for (uint64_t j = 0; j < size; j++) ai->a[0]->coeffs[j] = in[j]->a[i];
    entry->entry = i;
    trlwe_keyswitch(part, ai, entry);          /* (0, 0) - sum_j DFT(dec_j(a_i)) (.) KS[i][j] */
    trlwe_addto(out, part);
  }
  for (uint64_t j = 0; j < size; j++) out->b->coeffs[j] += in[j]->b;
  free(entry);
  free_trlwe(ai);
  free_trlwe(part);
}

TRLWE_KS_Key *trlwe_new_packing1_KS_key_CDKS21(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit) {
  const int N = out_key->s[0]->N;
  int log_N = 0;
  while ((1 << log_N) < N) log_N++;
  TorusPolynomial padded = polynomial_new_torus_polynomial(N), perm = polynomial_new_torus_polynomial(N);
  memset(padded->coeffs, 0, sizeof(Torus) * (size_t)N);
  memcpy(padded->coeffs, in_key->s, sizeof(Torus) * (size_t)(in_key->n < N ? in_key->n : N));
  Torus *msgs = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)log_N * N);
  for (int j = 0; j < log_N; j++) {
    polynomial_permute(perm, padded, ((uint64_t)1 << (log_N - j)) + 1);
    memcpy(msgs + (size_t)j * N, perm->coeffs, sizeof(Torus) * (size_t)N);
  }
  mosfhet_hip_gak_t dev = fft_ks_keys_new(out_key, msgs, log_N, t, base_bit, "trlwe_new_packing1_KS_key_CDKS21");
  free(msgs);
  free_polynomial(padded);
  free_polynomial(perm);
  TRLWE_KS_Key *res = (TRLWE_KS_Key *)mc_xmalloc(sizeof(TRLWE_KS_Key) * (size_t)log_N);
  for (int j = 0; j < log_N; j++) res[j] = trlwe_ks_header(dev, j, j == 0, t, base_bit);
  return res;
}

void trlwe_packing1_keyswitch_CDKS21(TRLWE out, TLWE in, TRLWE_KS_Key *ks_key) {
  const int N = out->b->N;
  TRLWE tmp = trlwe_alloc_new_sample(1, N);
  trlwe_noiseless_trivial_sample(out, NULL);
  for (int i = 1; i < N; i++) out->a[0]->coeffs[N - i] = (Torus)0 - in->a[i];   /* the LWE mask as the polynomial whose constant term pairs with the key */
  out->a[0]->coeffs[0] = in->a[0];
  out->b->coeffs[0] = in->b;
  for (int i = 1, j = 0; i < N; i <<= 1, j++) {   /* trace: sum over the automorphisms X -> X^(N / 2^j + 1) */
    const uint64_t gen = (uint64_t)(N >> j) + 1;
    polynomial_permute(tmp->a[0], out->a[0], gen);
    polynomial_permute(tmp->b, out->b, gen);
    trlwe_keyswitch(tmp, tmp, ks_key[j]);
    trlwe_addto(out, tmp);
  }
  free_trlwe(tmp);
}

/* TRLWE(m) -> TRLWE(m v) for a fixed polynomial v (src/keyswitch.c:64-96,574-608): entry 0 switches from s_in v, entry 1 from v; then
 * out = sum_j dec_j(in.b) KS1_j - sum_j dec_j(in.a) KS0_j, each sum one FFT key switch of a sample whose mask is the polynomial being decomposed */
TRLWE_KS_Key trlwe_new_RLWE_priv_KS_key(TRLWE_Key out_key, TRLWE_Key in_key, TorusPolynomial v, int t, int base_bit) {
  const int N = out_key->s[0]->N;
  if (out_key->k != 1 || in_key->k != 1 || in_key->s[0]->N != N) { fprintf(stderr, "mosfhet_amd: trlwe_new_RLWE_priv_KS_key: k = 1 and one ring only\n"); abort(); }
  Torus *msgs = (Torus *)mc_xmalloc(sizeof(Torus) * (size_t)2 * N);
  memset(msgs, 0, sizeof(Torus) * (size_t)N);
  negacyclic_mul_addto(msgs, v->coeffs, in_key->s[0]->coeffs, N);
  memcpy(msgs + N, v->coeffs, sizeof(Torus) * (size_t)N);
  mosfhet_hip_gak_t dev = fft_ks_keys_new(out_key, msgs, 2, t, base_bit, "trlwe_new_RLWE_priv_KS_key");
  free(msgs);
  return trlwe_ks_header(dev, 0, 1, t, base_bit);
}

void trlwe_RLWE_priv_keyswitch(TRLWE out, TRLWE in, TRLWE_KS_Key ks_key) {
  const int N = out->b->N;
  TRLWE src = trlwe_alloc_new_sample(1, N), as = trlwe_alloc_new_sample(1, N);
  TRLWE_KS_Key entry = trlwe_ks_header(ks_key->device, 0, 0, ks_key->t, ks_key->base_bit);
  trlwe_noiseless_trivial_sample(src, NULL);
  memcpy(src->a[0]->coeffs, in->a[0]->coeffs, sizeof(Torus) * (size_t)N);
  trlwe_keyswitch(as, src, entry);                 /* - sum_j dec_j(in.a) KS0_j */
  memcpy(src->a[0]->coeffs, in->b->coeffs, sizeof(Torus) * (size_t)N);
  entry->entry = 1;
  trlwe_keyswitch(out, src, entry);                /* - sum_j dec_j(in.b) KS1_j */
  trlwe_negate(out, out);
  trlwe_addto(out, as);
  free(entry);
  free_trlwe(src);
  free_trlwe(as);
}

TRLWE_KS_Key *trlwe_new_gadget_to_RGSW_KS(TRLWE_Key key, int t, int base_bit) {   /* one key per mask component: v = -s_i */
  TRLWE_KS_Key *res = (TRLWE_KS_Key *)mc_xmalloc(sizeof(TRLWE_KS_Key) * (size_t)key->k);
  TorusPolynomial neg = polynomial_new_torus_polynomial(key->s[0]->N);
  for (int i = 0; i < key->k; i++) {
    for (int j = 0; j < neg->N; j++) neg->coeffs[j] = (Torus)0 - key->s[i]->coeffs[j];
    res[i] = trlwe_new_RLWE_priv_KS_key(key, key, neg, t, base_bit);
  }
  free_polynomial(neg);
  return res;
}

/* the l gadget rows TRLWE(m 2^(64 - (i+1) Bg)) become a TRGSW_DFT(m): rows of component j through the key of -s_j, rows of b as they are */
void trgsw_from_gadget(TRGSW_DFT out, TRLWE *gadget, TRLWE_KS_Key *ksk) {
  const int k = out->samples[0]->k, N = out->samples[0]->b->N;
  TRLWE tmp = trlwe_alloc_new_sample(k, N);
  for (int j = 0; j < k; j++)
    for (int i = 0; i < out->l; i++) {
      trlwe_RLWE_priv_keyswitch(tmp, gadget[i], ksk[j]);
      trlwe_to_DFT(out->samples[i + j * out->l], tmp);
    }
  for (int i = 0; i < out->l; i++) trlwe_to_DFT(out->samples[i + k * out->l], gadget[i]);
  free_trlwe(tmp);
}
