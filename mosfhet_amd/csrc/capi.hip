// capi.hip -- C ABI of the gfx950 bootstrap engine (declared in include/mosfhet_hip.h).
// Host-side glue only: argument checks, device buffers, kernel launches.  No CPU fallback: every compute
// entry point fails with MOSFHET_HIP_ENODEV / MOSFHET_HIP_EHIP when the device path is unavailable.
#include "../../include/mosfhet_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <type_traits>
#include <vector>

#include <atomic>
#include <map>
#include <mutex>

#include "bootstrap_kernels.h"
#include "general_kernels.h"
#include "keyswitch_kernels.h"
#include "ext_kernels.h"
#include "unfold_kernels.h"
#include "keygen_kernels.h"

using namespace mosfhet;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(MOSFHET_HIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Handles own their device memory: the destructors release it (on the handle's device), so a creation that fails half-way gives everything back.
struct mosfhet_hip_ctx {
  int device = 0;
  d2 *tw1024 = nullptr, *tw2048 = nullptr, *tw4096 = nullptr;  // device twiddle tables
  std::map<int, d2 *> tw_general;                               // ... of the other rings, made on first use (general_kernels.h)
  d2 *wtab[3] = {nullptr, nullptr, nullptr};                    // W[x] = exp(i pi x / N), x < 2N, for N = 1024, 2048, 4096 (unfold_kernels.h)
  std::mutex tw_lock;
  ~mosfhet_hip_ctx() {
    (void)hipSetDevice(device);
    if (tw1024) (void)hipFree(tw1024);
    if (tw2048) (void)hipFree(tw2048);
    if (tw4096) (void)hipFree(tw4096);
    for (auto &e : tw_general) (void)hipFree(e.second);
    for (d2 *w : wtab)
      if (w) (void)hipFree(w);
  }
};

// scoped device buffer for the temporaries of the creation functions
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

// Temporaries of the compositions (FDFB, multi-value, circuit bootstraps, tlwe_mul, ...) and the transposed batches of the table key
// switches live in a pool that belongs to the CALLING HOST THREAD (one set of growable device buffers per device), never in a context or key
// handle: handles are read-only after creation, so any number of host threads may share them, as the reference's callers share its keys
// (re-entrant through thread-local scratch, src/polynomial.c:269-352).  A thread's launches are ordered by the stream it passes; buffers are
// released at thread exit (hipFree waits for work in flight).
enum { POOL_BSK = 0, POOL_EXT0 = 1, POOL_EXT1 = 2, POOL_CTX0 = 3, POOL_UNFOLD = 6, POOL_PACK = 7, POOL_VEC = 8, POOL_VEC2 = 9, POOL_VEC3 = 10, POOL_VEC4 = 11, POOL_SLOTS = 12 };
struct ThreadPool {
  struct Dev {
    int device = -1;
    KsWorkspace ws;
    uint64_t *buf[POOL_SLOTS] = {};
    size_t words[POOL_SLOTS] = {};
  };
  std::vector<Dev> devs;
  Dev &get(int device) {
    for (Dev &d : devs)
      if (d.device == device) return d;
    devs.emplace_back();
    devs.back().device = device;
    return devs.back();
  }
  ~ThreadPool() {
    for (Dev &d : devs) {
      if (hipSetDevice(d.device) != hipSuccess) continue;
      if (d.ws.inT) (void)hipFree(d.ws.inT);
      if (d.ws.outT) (void)hipFree(d.ws.outT);
      for (int i = 0; i < POOL_SLOTS; i++)
        if (d.buf[i]) (void)hipFree(d.buf[i]);
    }
  }
};
static thread_local ThreadPool t_pool;

static KsWorkspace &tl_ws(int device) { return t_pool.get(device).ws; }

static int pool_get(int device, int slot, size_t words, uint64_t **out) {
  ThreadPool::Dev &d = t_pool.get(device);
  if (d.words[slot] < words) {
    if (d.buf[slot]) (void)hipFree(d.buf[slot]);
    d.buf[slot] = nullptr;
    d.words[slot] = 0;
    HIP_TRY(hipMalloc((void **)&d.buf[slot], words * sizeof(uint64_t)));
    d.words[slot] = words;
  }
  *out = d.buf[slot];
  return MOSFHET_HIP_OK;
}

struct mosfhet_hip_bsk {
  mosfhet_hip_ctx_t ctx = nullptr;
  d2 *d_bk = nullptr;  // [n][(k+1)l][k+1][P][lanes]
  int n, k, N, l, Bg_bit;
  int unfolding = 1;          // > 1: d_bk is null and d_su holds the torus-domain samples of new_bootstrap_key (src/bootstrap.c:23-48)
  uint64_t *d_su = nullptr;   // [n 2^u / u][2l][2][N]
  d2 *d_su_dft = nullptr;     // unfolding 2: the same samples transformed, [n / 2][4][2l][2][8][T] (unfold_kernels.h); the rotation reads these, UBR phase 1 reads d_su
  std::mutex su_dft_lock;     // ... made at creation / cloning (unfold2_ready); a key made under MOSFHET_HIP_UNFOLD2_DFT=0 (torus-domain path) does not pay
  std::atomic<int> su_dft_ready{0};   // for the second copy unless the DFT path is switched on later
  size_t bytes = 0;
  bool general = false;       // k > 1 or a ring without a tuned kernel: natural slot order, general_kernels.h (bootstraps and external products only)
  bool owns = true;           // false: d_bk belongs to the caller (mosfhet_hip_bsk_view_create)
  ~mosfhet_hip_bsk() {
    if (ctx) (void)hipSetDevice(ctx->device);
    if (d_bk && owns) (void)hipFree(d_bk);
    if (d_su) (void)hipFree(d_su);
    if (d_su_dft) (void)hipFree(d_su_dft);
  }
};

struct mosfhet_hip_ksk {
  mosfhet_hip_ctx_t ctx = nullptr;
  uint64_t *d_ksk = nullptr;
  int n_in, n_out, t, base_bit;
  int row, b_word;  // output row words and the word that receives in.b (LWE: n_out + 1, n_out; packing -> TRLWE: 2N, N)
  bool compressed = false;  // TRLWE table keys only: d_ksk holds the b halves [rows][N], the masks are keygen_mix(seed, row, word)
  uint64_t seed = 0;
  size_t bytes = 0;
  ~mosfhet_hip_ksk() {
    if (ctx) (void)hipSetDevice(ctx->device);
    if (d_ksk) (void)hipFree(d_ksk);
  }
};

struct mosfhet_hip_gak {
  mosfhet_hip_ctx_t ctx = nullptr;
  d2 *d_ak = nullptr;  // [entries][t][2][8][T]
  int N, t, base_bit, entries;
  size_t bytes = 0;
  ~mosfhet_hip_gak() {
    if (ctx) (void)hipSetDevice(ctx->device);
    if (d_ak) (void)hipFree(d_ak);
  }
};

extern "C" const char *mosfhet_hip_last_error(void) { return g_err; }
extern "C" const char *mosfhet_hip_version(void) { return "mosfhet_amd 0.1 (gfx950)"; }

extern "C" int mosfhet_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- twiddle table (same definition as oracle_fft.c:orc_fft_make_twiddles; own code) ----
static unsigned bit_reverse(unsigned x, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

static void make_twiddles(int N, std::vector<double> &tw) {
  const int M = N / 2;
  int logM = 0;
  while ((1 << logM) < M) logM++;
  tw.assign(2 * (size_t)(M - 1), 0.0);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int lev = 0; lev < logM; lev++)
    for (unsigned nu = 0; nu < (1u << lev); nu++) {
      double *e = tw.data() + 2 * ((size_t)(1u << lev) - 1 + nu);
      if (nu & 1) {  // i * sibling, exact
        e[0] = -e[-1];
        e[1] = e[-2];
      } else {
        const long double frac = (long double)(4 * bit_reverse(nu, lev) + 1) / (long double)(1ull << (lev + 3));
        e[0] = (double)cosl(two_pi * frac);
        e[1] = (double)sinl(two_pi * frac);
      }
    }
}

extern "C" int mosfhet_hip_twiddles(int N, double *h_out) {
  if (N < 8 || (N & (N - 1)) || !h_out) return fail(MOSFHET_HIP_EINVAL, "twiddles: bad N=%d", N);
  std::vector<double> tw;
  make_twiddles(N, tw);
  memcpy(h_out, tw.data(), tw.size() * sizeof(double));
  return MOSFHET_HIP_OK;
}

static unsigned int *pace_ring(int dev, bool may_allocate);   // counters of the N = 2048 rendezvous (launch_pbs)

extern "C" int mosfhet_hip_ctx_create(mosfhet_hip_ctx_t *out, int device) {
  if (!out) return fail(MOSFHET_HIP_EINVAL, "ctx_create: null out");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(MOSFHET_HIP_ENODEV, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(MOSFHET_HIP_EINVAL, "ctx_create: device %d of %d", device, ndev);
  {
    // the library carries gfx950 code objects only: say so instead of failing at the first launch
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(MOSFHET_HIP_ENODEV, "device %d is %s: this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  }
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<mosfhet_hip_ctx> c(new mosfhet_hip_ctx());
  c->device = device;
  for (int N : {1024, 2048, 4096}) {
    std::vector<double> tw;
    make_twiddles(N, tw);
    d2 *&dst = (N == 1024) ? c->tw1024 : (N == 2048 ? c->tw2048 : c->tw4096);
    HIP_TRY(hipMalloc((void **)&dst, tw.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(dst, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    // same definition as oracle_ext.c:orc_monomial_table
    std::vector<double> w(4 * (size_t)N);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int x = 0; x < 2 * N; x++) {
      w[2 * (size_t)x] = (double)cosl(pi * (long double)x / (long double)N);
      w[2 * (size_t)x + 1] = (double)sinl(pi * (long double)x / (long double)N);
    }
    d2 *&wd = c->wtab[N == 1024 ? 0 : (N == 2048 ? 1 : 2)];
    HIP_TRY(hipMalloc((void **)&wd, w.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(wd, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  (void)pace_ring(device, true);
  *out = c.release();
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_ctx_destroy(mosfhet_hip_ctx_t ctx) {
  if (!ctx) return MOSFHET_HIP_OK;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  delete ctx;
  return MOSFHET_HIP_OK;
}

// NULL means HIP's default (null) stream -- the same convention as every HIP API, and what torch hands out for
// its default stream -- never a private stream, so work stays ordered with the caller's copies.
static hipStream_t pick(mosfhet_hip_ctx_t, void *stream) { return (hipStream_t)stream; }

// Ring dispatch: BODY runs with `F` = the transform type of ring degree n_ (Fft1024 / Fft2048 / Fft4096) and `TW` = its twiddle table.
static bool ring_ok(int N) { return N == 1024 || N == 2048 || N == 4096; }
#define RING_DISPATCH(ctx_, n_, ...)                                                                   \
  do {                                                                                                 \
    if ((n_) == 1024) { using F = Fft1024; const d2 *TW = (ctx_)->tw1024; (void)TW; __VA_ARGS__; }     \
    else if ((n_) == 2048) { using F = Fft2048; const d2 *TW = (ctx_)->tw2048; (void)TW; __VA_ARGS__; } \
    else { using F = Fft4096; const d2 *TW = (ctx_)->tw4096; (void)TW; __VA_ARGS__; }                  \
  } while (0)

extern "C" int mosfhet_hip_ctx_sync(mosfhet_hip_ctx_t ctx, void *stream) {
  if (!ctx) return fail(MOSFHET_HIP_EINVAL, "ctx_sync: null ctx");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(pick(ctx, stream)));
  return MOSFHET_HIP_OK;
}

// Compile-time gadgets (2 x 2^8, 4 x 2^9) exist where a parameter set of the reference or of BASELINE.json runs them: N = 1024 and N = 2048.  At N = 4096 (the reference's
// sets there have l = 1) they take the run-time-gadget instantiations -- same bits, fewer kernels (round 6 pruning: tools/kernel_table.py).
template <class F> constexpr bool kCompileTimeGadgets = F::N != 4096;

// k = 1 and N in {1024, 2048, 4096}: the tuned kernels.  Any other power-of-two ring up to 16384 and k <= 3: the general path (general_kernels.h).
static bool general_ring(int k, int N) { return !(k == 1 && ring_ok(N)); }
static int check_params(const char *who, int k, int N, int l, int Bg_bit, bool allow_general = false) {
  if (general_ring(k, N)) {
    if (!allow_general) {
      if (k != 1) return fail(MOSFHET_HIP_EINVAL, "%s: only k = 1 is supported here (got %d)", who, k);
      return fail(MOSFHET_HIP_EINVAL, "%s: ring degree N = %d not supported here (1024, 2048, 4096)", who, N);
    }
    if (k < 1 || k > 3) return fail(MOSFHET_HIP_EINVAL, "%s: k = %d not supported (1 .. 3)", who, k);
    if (N < 256 || N > 16384 || (N & (N - 1))) return fail(MOSFHET_HIP_EINVAL, "%s: ring degree N = %d not supported (a power of two in 256 .. 16384)", who, N);
  }
  if (l < 1 || Bg_bit < 1 || Bg_bit > 31 || l * Bg_bit >= 64) return fail(MOSFHET_HIP_EINVAL, "%s: bad gadget l=%d Bg_bit=%d (Bg_bit <= 31, l*Bg_bit < 64)", who, l, Bg_bit);
  if (l > 6) return fail(MOSFHET_HIP_EINVAL, "%s: l = %d not instantiated (1..6: the reference's own programs go up to l = 6, applications/multi-ciphertext-arith/src/ufhe.c:19)", who, l);
  return MOSFHET_HIP_OK;
}

static int ilog2(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

// twiddle table of a general ring on the context's device, made on first use
static int general_twiddles(mosfhet_hip_ctx_t ctx, int N, const d2 **out) {
  if (N == 1024) { *out = ctx->tw1024; return MOSFHET_HIP_OK; }
  if (N == 2048) { *out = ctx->tw2048; return MOSFHET_HIP_OK; }
  if (N == 4096) { *out = ctx->tw4096; return MOSFHET_HIP_OK; }
  std::lock_guard<std::mutex> hold(ctx->tw_lock);
  auto it = ctx->tw_general.find(N);
  if (it == ctx->tw_general.end()) {
    std::vector<double> tw;
    make_twiddles(N, tw);
    d2 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, tw.size() * sizeof(double) + 16));
    HIP_TRY(hipMemcpy(d, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->tw_general.emplace(N, d).first;
  }
  *out = it->second;
  return MOSFHET_HIP_OK;
}

// general kernels keep the whole transform in LDS: (N / 2) complex = 8 N bytes, above 64 KiB only with the attribute set
template <class K>
static int general_lds(K kernel, int N) {
  const int bytes = 8 * N;
  if (bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return MOSFHET_HIP_OK;
}

#define TUNED_ONLY(bsk_, who_)                                                                                                                       \
  do {                                                                                                                                               \
    if ((bsk_) && (bsk_)->general)                                                                                                                   \
      return fail(MOSFHET_HIP_EINVAL, "%s: keys of the general-ring path (k > 1 or N outside 1024 / 2048 / 4096) serve bootstraps, the full-domain " \
                                      "bootstrap, multi-value bootstraps (CLOT21), key-switch + bootstrap, external products and CMUX only", who_);                                         \
  } while (0)

// ---- bootstrap key ----
extern "C" int mosfhet_hip_bsk_create_from_device(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *d_bk,
                                                  int n, int k, int N, int l, int Bg_bit, void *stream) {
  if (!ctx || !out || !d_bk || n < 1) return fail(MOSFHET_HIP_EINVAL, "bsk_create: bad argument");
  int rc = check_params("bsk_create", k, N, l, Bg_bit, true);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  std::unique_ptr<mosfhet_hip_bsk> b_owner(new mosfhet_hip_bsk());
  mosfhet_hip_bsk *b = b_owner.get();
  b->ctx = ctx; b->n = n; b->k = k; b->N = N; b->l = l; b->Bg_bit = Bg_bit;
  b->general = general_ring(k, N);
  const size_t polys = (size_t)n * (k + 1) * l * (k + 1);
  b->bytes = polys * N * sizeof(double);
  HIP_TRY(hipMalloc((void **)&b->d_bk, b->bytes));
  if (b->general) {
    const d2 *tw = nullptr;
    if ((rc = general_twiddles(ctx, N, &tw)) || (rc = general_lds(torus_to_dft_general_kernel, N))) return rc;
    hipLaunchKernelGGL(torus_to_dft_general_kernel, dim3((unsigned)polys), dim3(GEN_THREADS), (size_t)8 * N, pick(ctx, stream), d_bk, b->d_bk, tw, N, ilog2(N / 2));
  } else
  RING_DISPATCH(ctx, N, hipLaunchKernelGGL(torus_to_dft_kernel<F>, dim3((unsigned)polys), dim3(F::THREADS), 0, pick(ctx, stream), d_bk, b->d_bk, TW));
  HIP_TRY(hipGetLastError());
  *out = b_owner.release();
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_bsk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *h_bk,
                                      int n, int k, int N, int l, int Bg_bit) {
  if (!ctx || !out || !h_bk || n < 1) return fail(MOSFHET_HIP_EINVAL, "bsk_create: bad argument");
  int rc = check_params("bsk_create", k, N, l, Bg_bit, true);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)n * (k + 1) * l * (k + 1) * N * sizeof(uint64_t);
  uint64_t *d_tmp = nullptr;
  HIP_TRY(hipMalloc((void **)&d_tmp, bytes));
  hipError_t e = hipMemcpy(d_tmp, h_bk, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(d_tmp); return fail(MOSFHET_HIP_EHIP, "bsk upload: %s", hipGetErrorString(e)); }
  rc = mosfhet_hip_bsk_create_from_device(ctx, out, d_tmp, n, k, N, l, Bg_bit, nullptr);
  hipStreamSynchronize(nullptr);
  hipFree(d_tmp);
  return rc;
}

extern "C" int mosfhet_hip_bsk_destroy(mosfhet_hip_bsk_t bsk) {
  if (!bsk) return MOSFHET_HIP_OK;
  delete bsk;
  return MOSFHET_HIP_OK;
}

extern "C" size_t mosfhet_hip_bsk_bytes(mosfhet_hip_bsk_t bsk) { return bsk ? bsk->bytes : 0; }

extern "C" int mosfhet_hip_bsk_export_dft(mosfhet_hip_bsk_t bsk, double *h_out) {
  if (!bsk || !h_out) return fail(MOSFHET_HIP_EINVAL, "bsk_export: bad argument");
  if (bsk->unfolding > 1) return fail(MOSFHET_HIP_EINVAL, "bsk_export: an unfolded key has no DFT form");
  HIP_TRY(hipSetDevice(bsk->ctx->device));
  std::vector<double> tmp(bsk->bytes / sizeof(double));
  HIP_TRY(hipMemcpy(tmp.data(), bsk->d_bk, bsk->bytes, hipMemcpyDeviceToHost));
  const int M = bsk->N / 2;
  const size_t polys = bsk->bytes / sizeof(double) / bsk->N;
  if (bsk->general) {   // natural slot order already
    memcpy(h_out, tmp.data(), bsk->bytes);
    return MOSFHET_HIP_OK;
  }
  for (size_t q = 0; q < polys; q++)
    for (int j = 0; j < M; j++) {  // oracle index j = thread*8 + m  <->  device index m*T + thread, T = N/16
      const int dev = (j & 7) * (bsk->N / 16) + (j >> 3);
      h_out[q * bsk->N + 2 * j] = tmp[q * bsk->N + 2 * dev];
      h_out[q * bsk->N + 2 * j + 1] = tmp[q * bsk->N + 2 * dev + 1];
    }
  return MOSFHET_HIP_OK;
}

// ---- external-product launches ----
// Persistent teams (external_product_kernel): the grid is the chip's resident capacity -- 2 wavefronts per SIMD, i.e. CUs x 8 / (wavefronts per team)
// teams -- capped by the batch; gadgets of the reference's parameter sets get the compile-time instantiation.
static int device_cus() {   // CUs of the current device (cached per device id)
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int c = cache[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) return 0;
    cache[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}
static int resident_teams(int threads) {
  const int cus = device_cus();
  return (cus > 0 ? cus : 256) * 8 / (threads / 64);
}

// Scratch bytes per lane of a kernel of this library (hipFuncAttributes::localSizeBytes = the code object's private_segment_fixed_size), asked once per
// instantiation: the code object is the same on every device.  -1 when the runtime cannot say.
template <class K>
static int kernel_scratch_bytes(K kernel) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void *>(kernel)) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return (int)a.localSizeBytes;
}

// Which unit loop a multi-wavefront external-product instantiation takes (external_product_kernel: FORM).  The software-pipelined loop is taken where it is faster
// (ep_pipelined_by_default: lvl2's <Fft2048L, 4, *> without the CMUX operand).  Its wrong-unit failure of rounds 3 - 5 was a compiler reordering of LDS reads
// across a workgroup barrier, fixed at the root in round 5 (negacyclic_fft.h: workgroup_sync; bootstrap_kernels.h has the story).  The launcher still asks the
// runtime for that instantiation's scratch size at its first launch and takes the plain loop (same bits) when it is not zero: not for correctness any more, but
// because the pipelined form was measured (+11 %) and soaked (tools/soak.py) as a scratch-free build only -- a build that spills in-flight ciphertext words is a
// different kernel and most likely the slower one.  MOSFHET_HIP_EP_PAIRS=0 / mosfhet_hip_set_ep_plain_loop force the plain loop.
struct EpKernelInfo { const char *name; int pipelined_by_default, scratch_bytes, takes_pipelined; };
static std::mutex g_ep_info_lock;
static std::vector<EpKernelInfo> g_ep_info;   // every instantiation that has been asked about (mosfhet_hip_ep_kernel_info)
template <class F, int L, int BG, bool CMUX>
static bool ep_multiwave_pipelined_ok(const char *name) {
  static std::atomic<int> verdict{-1};
  int v = verdict.load(std::memory_order_acquire);
  if (v < 0) {
    const char *e = getenv("MOSFHET_HIP_EP_PAIRS");
    const int scratch = kernel_scratch_bytes(external_product_kernel<F, L, BG, CMUX, 0>);
    v = (scratch == 0 && !(e && e[0] == '0')) ? 1 : 0;
    std::lock_guard<std::mutex> hold(g_ep_info_lock);
    if (verdict.load(std::memory_order_relaxed) < 0) g_ep_info.push_back(EpKernelInfo{name, 1, scratch, v});
    verdict.store(v, std::memory_order_release);
  }
  return v == 1;
}

static std::atomic<int> g_ep_plain_loop{0};   // mosfhet_hip_set_ep_plain_loop: every multi-wavefront instantiation takes the plain loop (tests run both forms in one process)
extern "C" int mosfhet_hip_set_ep_plain_loop(int on) {
  g_ep_plain_loop.store(on ? 1 : 0, std::memory_order_relaxed);
  return MOSFHET_HIP_OK;
}

template <class FF, int LL, int BB, bool CM>
static void ep_go(const char *name, dim3 grid, dim3 block, hipStream_t s, const d2 *row, const d2 *tw, const uint64_t *d_in, uint64_t *d_out, int Bg_bit, int count, size_t key_stride,
                  size_t in_stride, const uint64_t *d_in0, d2 *d_out_dft) {
  if constexpr (FF::THREADS > 64 && ep_pipelined_by_default<FF, LL, CM>()) {
    if (!ep_multiwave_pipelined_ok<FF, LL, BB, CM>(name) || g_ep_plain_loop.load(std::memory_order_relaxed)) {
      hipLaunchKernelGGL((external_product_kernel<FF, LL, BB, CM, 1>), grid, block, 0, s, row, tw, d_in, d_out, Bg_bit, count, key_stride, in_stride, d_in0, d_out_dft);
      return;
    }
  }
  hipLaunchKernelGGL((external_product_kernel<FF, LL, BB, CM, 0>), grid, block, 0, s, row, tw, d_in, d_out, Bg_bit, count, key_stride, in_stride, d_in0, d_out_dft);
}

template <class F>
static void launch_external_product(int l, int Bg_bit, hipStream_t s, const d2 *row, const d2 *tw, const uint64_t *d_in, uint64_t *d_out, int count, size_t key_stride,
                                    size_t in_stride, const uint64_t *d_in0, d2 *d_out_dft, bool bounded) {
  // bounded: every key coefficient is the transform of torus words (|x| <= 2^63: keys this library transformed itself).  Only then may the compile-time
  // 2 x 2^8 gadget skip the reduction mod 1 in its rounding (pbs_kernel: kReduce).  Caller-supplied DFT objects -- non-owning key views, TRGSW_DFT
  // selectors, sums made with trgsw_DFT_add / _mul_addto -- carry no such bound and take the run-time-gadget instantiation, which reduces like
  // the reference (fft_processor_spqlios.c:155-165 reduces any double).
  if (!bounded && l == 2 && Bg_bit == 8 && std::is_same<F, Fft1024>::value) Bg_bit = -8;   // (negative: routed to the <2, 0> instantiation below)
  const int cap = resident_teams(F::THREADS);
  if constexpr (std::is_same<F, Fft1024>::value) {
    // one key entry for the whole batch at N = 1024, l <= 2: the entry fits LDS next to eight teams' transpose buffers (external_product_ldskey_kernel)
    if (key_stride == 0 && l <= 2 && count >= 64) {
      const int wgs = (count + 7) / 8, cus = cap / 8;
      const dim3 grid((unsigned)(wgs < cus ? wgs : cus)), block(512);
#define EPL_GO(LL, BB)                                                                                                                                    \
  do {                                                                                                                                                    \
    if (d_in0) hipLaunchKernelGGL((external_product_ldskey_kernel<LL, BB, true>), grid, block, 0, s, row, tw, d_in, d_out, Bg_bit, count, in_stride, d_in0, d_out_dft);   \
    else hipLaunchKernelGGL((external_product_ldskey_kernel<LL, BB, false>), grid, block, 0, s, row, tw, d_in, d_out, Bg_bit, count, in_stride, d_in0, d_out_dft);      \
  } while (0)
      if (l == 2 && Bg_bit == 8) EPL_GO(2, 8);
      else if (l == 1 && Bg_bit == 23) EPL_GO(1, 23);
      else if (l == 1) EPL_GO(1, 0);
      else { if (Bg_bit < 0) Bg_bit = -Bg_bit; EPL_GO(2, 0); }
#undef EPL_GO
      return;
    }
  }
  const bool unbounded_2x8 = Bg_bit < 0;
  if (unbounded_2x8) Bg_bit = -Bg_bit;
  const dim3 grid((unsigned)(count < cap ? count : cap)), block(F::THREADS);
#define EP_GO_F(FF, LL, BB)                                                                                                                                        \
  do {                                                                                                                                                             \
    if (d_in0) ep_go<FF, LL, BB, true>("external_product_kernel<" #FF ", " #LL ", " #BB ", cmux>", grid, block, s, row, tw, d_in, d_out, Bg_bit, count, key_stride, in_stride, d_in0, d_out_dft); \
    else ep_go<FF, LL, BB, false>("external_product_kernel<" #FF ", " #LL ", " #BB ">", grid, block, s, row, tw, d_in, d_out, Bg_bit, count, key_stride, in_stride, d_in0, d_out_dft);        \
  } while (0)
#define EP_GO(LL, BB) EP_GO_F(F, LL, BB)
  if constexpr (std::is_same<F, Fft2048>::value) {
    // N = 2048, l = 4 (lvl2): rows two at a time with the pass twiddles in LDS and, outside the CMUX form, the pipelined unit loop
    // (external_product_kernel: kPairs / kPipe; bit-identical; lvl2: 0.437 against 0.492 ms per 16,384 units in a same-box A/B), guarded by ep_go.
    if (l == 4 && Bg_bit == 9) { EP_GO_F(Fft2048L, 4, 9); return; }
    if (l == 4) { EP_GO_F(Fft2048L, 4, 0); return; }   // (other even lengths: not measured)
  }
  if (kCompileTimeGadgets<F> && l == 2 && Bg_bit == 8 && !unbounded_2x8) { if constexpr (kCompileTimeGadgets<F>) EP_GO(2, 8); }
  else if (kCompileTimeGadgets<F> && l == 4 && Bg_bit == 9) { if constexpr (kCompileTimeGadgets<F> && !std::is_same<F, Fft2048>::value) EP_GO(4, 9); }   // (N = 2048, l = 4 left above: no instantiation here)
  else if (l == 1 && Bg_bit == 23) EP_GO(1, 23);
  else if (l == 1) EP_GO(1, 0);
  else if (l == 2) EP_GO(2, 0);
  else if (l == 3) EP_GO(3, 0);
  else if (l == 4) { if constexpr (!std::is_same<F, Fft2048>::value) EP_GO(4, 0); }
  else if (l == 5) EP_GO(5, 0);
  else EP_GO(6, 0);
#undef EP_GO
#undef EP_GO_F
}

// What the launcher decided for the multi-wavefront instantiations that take the pipelined unit loop by default, for tests and tools: entry i of those asked about so
// far (an instantiation is asked about at its first launch).  Returns MOSFHET_HIP_EINVAL past the end.
extern "C" int mosfhet_hip_ep_kernel_info(int i, const char **name, int *scratch_bytes, int *takes_pipelined) {
  std::lock_guard<std::mutex> hold(g_ep_info_lock);
  if (i < 0 || (size_t)i >= g_ep_info.size()) return MOSFHET_HIP_EINVAL;
  if (name) *name = g_ep_info[i].name;
  if (scratch_bytes) *scratch_bytes = g_ep_info[i].scratch_bytes;
  if (takes_pipelined) *takes_pipelined = g_ep_info[i].takes_pipelined;
  return MOSFHET_HIP_OK;
}

// ---- bootstrap launches ----
// One launch per residency round when the key does not fit the L2s (N >= 2048).  All teams walk the key rows in the same order and
// share each row through their XCD's L2 while they stay close together; in one big launch the teams of later rounds start as earlier
// ones finish, the phases smear out and the sharing collapses (4096 bootstraps at lvl2 in one launch: L2 hit rate 52 %, 318 GB of
// fabric reads; one round alone: 96 %, 7 GB).  Kernel boundaries re-align the teams: 79 -> 74 ms.  A round = CUs x resident teams
// per CU (LDS-limited: 4 at N = 2048, 2 at N = 4096).  MOSFHET_HIP_ROUND_CHUNK overrides (0 = single launch).
static int round_chunk(int threads) {
  static std::atomic<int> env{-2};
  int e = env.load(std::memory_order_relaxed);
  if (e == -2) { const char *v = getenv("MOSFHET_HIP_ROUND_CHUNK"); e = v ? atoi(v) : -1; if (e < -1) e = -1; env.store(e, std::memory_order_relaxed); }
  if (e >= 0) return e;
  const int cus = device_cus();   // of the current device (several devices in one process: mosfhet_compat_multi.c)
  return threads == 128 ? 4 * (cus > 0 ? cus : 256) : 0;   // N = 4096 (2 teams per CU) measured slower in rounds (tail idling): single launch
}

// Teams of one residency round re-align every K CMUX steps (pace_teams, bootstrap_kernels.h): kernel boundaries alone leave them drifting apart inside
// a launch, and how far depends on what ran before (lvl2, 1024 per launch: 17.6 ms back to back but 19.5-20.3 ms after any other full-chip kernel, with
// twice the fabric reads -- the state every composition runs in).  Re-aligned every 32 steps: 17.3 ms in both cases (tools/gpu_perf_modes3.py;
// experiments/README.md "Round 4", pacing).  MOSFHET_HIP_PACE=K overrides (0 = off), MOSFHET_HIP_PACE_LIMIT the bounded wait in 10 ns ticks.
static int pace_every() {
  static std::atomic<int> v{-1};
  int r = v.load(std::memory_order_relaxed);
  if (r < 0) { const char *e = getenv("MOSFHET_HIP_PACE"); r = e ? atoi(e) : 32; if (r < 0) r = 0; v.store(r, std::memory_order_relaxed); }
  return r;
}
static int pace_limit() {
  static std::atomic<int> v{-1};
  int r = v.load(std::memory_order_relaxed);
  if (r < 0) { const char *e = getenv("MOSFHET_HIP_PACE_LIMIT"); r = e ? atoi(e) : 100000; if (r < 1) r = 1; v.store(r, std::memory_order_relaxed); }
  return r;
}
constexpr int PACE_WORDS = 288;   // 8 per-XCD counters on 128-byte lines of their own + the give-up flag (word 256)
// a ring of 256 counter blocks per device (allocated when the first context on the device is made -- not lazily at a launch, which could sit inside a stream
// capture -- and kept for the life of the process), handed out round-robin to all host threads and prepared (pace_prepare_kernel) on the launch stream in front of the launch that uses it.
// A block comes round again 256 paced launches later; should the earlier launch still be running then, the two share a block and the rendezvous misfires -- a
// timing matter only (results never depend on it), ended by the bounded wait.
constexpr int PACE_MAX_DEV = 64, PACE_SLOTS = 256;
static std::atomic<unsigned int *> g_pace_ring[PACE_MAX_DEV];
static std::atomic<unsigned> g_pace_next[PACE_MAX_DEV];
static unsigned int *pace_ring(int dev, bool may_allocate) {
  static std::mutex mu;
  if (dev < 0 || dev >= PACE_MAX_DEV) return nullptr;
  unsigned int *mem = g_pace_ring[dev].load(std::memory_order_acquire);
  if (mem || !may_allocate) return mem;
  std::lock_guard<std::mutex> g(mu);
  mem = g_pace_ring[dev].load(std::memory_order_relaxed);
  if (!mem) {
    if (hipMalloc((void **)&mem, (size_t)PACE_SLOTS * PACE_WORDS * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    // launches that skip the rendezvous after one gave up (pace_teams): MOSFHET_HIP_PACE_SKIP, default 16; 0 = every launch tries (rounds 4's behaviour)
    const char *e = getenv("MOSFHET_HIP_PACE_SKIP");
    const unsigned int skip = e && atoi(e) >= 0 ? (unsigned int)atoi(e) : 16u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(pace_skip_after_giveup), &skip, sizeof(skip)) != hipSuccess) (void)hipGetLastError();
    g_pace_ring[dev].store(mem, std::memory_order_release);
  }
  return mem;
}
static unsigned int *pace_slot(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  unsigned int *mem = pace_ring(dev, false);   // no ring (allocation failed at context creation): the launch runs unpaced
  if (!mem) return nullptr;
  unsigned int *slot = mem + (size_t)PACE_WORDS * (g_pace_next[dev].fetch_add(1u, std::memory_order_relaxed) % PACE_SLOTS);
  hipLaunchKernelGGL(pace_prepare_kernel, dim3(1), dim3(PACE_WORDS), 0, s, slot);
  if (hipGetLastError() != hipSuccess) return nullptr;
  return slot;
}

// Launches of `ctx`'s device that will still skip the rendezvous because an earlier one gave up (pace_teams); synchronises the device.  For tests and tools.
extern "C" int mosfhet_hip_pace_skip_credit(mosfhet_hip_ctx_t ctx, int *credit) {
  if (!ctx || !credit) return fail(MOSFHET_HIP_EINVAL, "pace_skip_credit: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned int v = 0;
  HIP_TRY(hipMemcpyFromSymbol(&v, HIP_SYMBOL(pace_skip_credit), sizeof(v)));
  *credit = (int)v;
  return MOSFHET_HIP_OK;
}

template <class F, int L, int BG>
static void launch_pbs(const PbsParams &p_in, int count, hipStream_t s) {
  PbsParams p = p_in;
  const int chunk = F::THREADS > 64 && p.count < 0 ? round_chunk(F::THREADS) : 0;   // p.count < 0: key larger than the L2s
  const bool pace = F::THREADS > 64 && pace_every() > 0 && chunk > 0 && count >= 64;   // (all teams of a launch of <= chunk are resident)
  p.pace_every = pace ? pace_every() : 0;
  p.pace_limit = pace_limit();
  if (chunk > 0 && count > chunk) {
    // row mode (TRGSW accumulators, per-level test vectors: block b takes input b / rows and test vector b % rows of ONE shared set): rounds of whole inputs
    const int rows = p.rows > 1 ? p.rows : 1, step = chunk < rows ? rows : chunk - chunk % rows;   // (a chunk below one input's rows: one input per launch)
    const size_t out_row = p.extract ? (size_t)F::N + 1 : (size_t)2 * F::N;
    for (int lo = 0; lo < count; lo += step) {
      PbsParams q = p;
      q.pace = pace ? pace_slot(s) : nullptr;
      q.in = p.in + (size_t)(lo / rows) * (p.n + 1);
      q.out = p.out + (size_t)lo * out_row;
      q.tv = rows > 1 ? p.tv : (p.tv ? p.tv + (size_t)lo * p.tv_stride : p.tv);
      const int c = count - lo < step ? count - lo : step;
      hipLaunchKernelGGL((pbs_kernel<F, L, BG>), dim3((unsigned)c), dim3(F::THREADS), 0, s, q);
    }
    return;
  }
  p.pace = pace && count <= chunk ? pace_slot(s) : nullptr;
  hipLaunchKernelGGL((pbs_kernel<F, L, BG>), dim3((unsigned)count), dim3(F::THREADS), 0, s, p);
}

template <class F>
static int launch_pbs_f(int l, int Bg_bit, const PbsParams &p, int count, hipStream_t s, bool bounded) {
  // Gadget bases of the reference's parameter sets get a compile-time instantiation (test/benchmark.c:53-75,
  // test/tests.c:37-62,967); anything else runs the run-time-Bg variant.  bounded: see launch_external_product -- a key view over caller-held
  // TRGSW_DFT objects (blind_rotate(tv, a, TRGSW_DFT *s, size)) takes the reducing run-time-gadget kernel.
  if (kCompileTimeGadgets<F> && l == 2 && Bg_bit == 8 && (bounded || F::N != 1024)) { if constexpr (kCompileTimeGadgets<F>) launch_pbs<F, 2, 8>(p, count, s); }
  else if (kCompileTimeGadgets<F> && l == 4 && Bg_bit == 9) { if constexpr (kCompileTimeGadgets<F>) launch_pbs<F, 4, 9>(p, count, s); }
  // 6 x 2^7 at N = 2048: the one parameter set of the reference's radix-integer application (applications/multi-ciphertext-arith/src/ufhe.c:18-20); the compile-time
  // gadget is worth 20 - 35 % on this kernel (l = 4: 17.1 against 21.7 ms per 1024 with the gadget at run time)
  else if (l == 6 && Bg_bit == 7 && F::N == 2048) { if constexpr (F::N == 2048) launch_pbs<F, 6, 7>(p, count, s); }
  // l = 1: the transform grouping with a full last pass (negacyclic_fft.h, Fft2048T: same results, same key layout; +2 % at SET_2, +6 % at SET_3)
  else if (l == 1 && Bg_bit == 23) launch_pbs<typename WideTail<F>::type, 1, 23>(p, count, s);
  else if (l == 1) launch_pbs<typename WideTail<F>::type, 1, 0>(p, count, s);
  else if (l == 2) launch_pbs<F, 2, 0>(p, count, s);
  else if (l == 3) launch_pbs<F, 3, 0>(p, count, s);
  else if (l == 4) launch_pbs<F, 4, 0>(p, count, s);
  else if (l == 5) launch_pbs<F, 5, 0>(p, count, s);
  else if (l == 6) launch_pbs<F, 6, 0>(p, count, s);
  else return fail(MOSFHET_HIP_EINVAL, "l = %d not instantiated", l);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

// Batches up to this size take pbs_team_kernel (N = 1024): below ~1.5 workgroups per CU the one-wavefront-per-ciphertext kernel leaves
// most of the chip idle and a bootstrap's latency is what counts.  MOSFHET_HIP_TEAM_MAX overrides (0 disables).
static std::atomic<int> g_team_max{-1};
static int team_max_batch() {
  if (g_team_max < 0) {
    const char *e = getenv("MOSFHET_HIP_TEAM_MAX");
    g_team_max = e ? atoi(e) : 512;
  }
  return g_team_max;
}
extern "C" int mosfhet_hip_set_team_max_batch(int max_batch) {
  g_team_max = max_batch < 0 ? 0 : max_batch;
  return MOSFHET_HIP_OK;
}

// The same switch for N = 2048 (pbs_wide_team_kernel: one workgroup of two transform teams per ciphertext).
// MOSFHET_HIP_WIDE_TEAM_MAX overrides (0 disables).
static std::atomic<int> g_wide_team_max{-1};
static int wide_team_max_batch() {
  if (g_wide_team_max < 0) {
    const char *e = getenv("MOSFHET_HIP_WIDE_TEAM_MAX");
    g_wide_team_max = e ? atoi(e) : 512;
  }
  return g_wide_team_max;
}
extern "C" int mosfhet_hip_set_wide_team_max_batch(int max_batch) {
  g_wide_team_max = max_batch < 0 ? 0 : max_batch;
  return MOSFHET_HIP_OK;
}

template <class F, int LL, int BB>
static int launch_wide_team(const PbsParams &p, int count, hipStream_t s) {
  constexpr size_t lds = sizeof(d2) * (size_t)2 * F::XCH_SLOTS + sizeof(uint64_t) * 2 * F::N;
  // per launch, like general_lds: the attribute belongs to the current device's copy of the kernel (several devices in one process: mosfhet_compat_multi.c)
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_wide_team_kernel<F, LL, BB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((pbs_wide_team_kernel<F, LL, BB>), dim3((unsigned)count), dim3(2 * F::THREADS), lds, s, p);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}
// bounded: see launch_external_product -- at N = 1024 the compile-time 2 x 2^8 instantiation rounds without the reduction mod 1, which only keys this
// library transformed itself allow; key views over caller-held TRGSW_DFT sums take the reducing run-time-gadget instantiation
// N = 2048, even gadget lengths, at most one workgroup per CU: the teams take their rows two at a time (pbs_wide_pair_kernel).  MOSFHET_HIP_WIDE_PAIRS=0 / 1.
static int wide_pairs_enabled() {
  static std::atomic<int> v{-1};
  int r = v.load(std::memory_order_relaxed);
  if (r < 0) { const char *e = getenv("MOSFHET_HIP_WIDE_PAIRS"); r = e ? (atoi(e) != 0) : 1; v.store(r, std::memory_order_relaxed); }
  return r;
}
template <int LL, int BB>
static int launch_wide_pair(const PbsParams &p, int count, hipStream_t s) {
  using F = Fft2048L;
  constexpr size_t lds = sizeof(d2) * ((size_t)2 * F::XCH_SLOTS + (size_t)4 * F::M) + sizeof(uint64_t) * 2 * F::N;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_wide_pair_kernel<F, LL, BB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((pbs_wide_pair_kernel<F, LL, BB>), dim3((unsigned)count), dim3(2 * F::THREADS), lds, s, p);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

// ---- one bootstrap on two workgroups (pbs_split_kernel): N = 2048, l = 2, 4, 6, batches of at most half the CUs ----
// MOSFHET_HIP_SPLIT_MAX: largest batch that takes it (-1 = CUs / 2, the default; 0 = never).  The ONE switch of this library that changes bits: the split kernel sums
// the external product per accumulator component (see the kernel), within FFT rounding of every other kernel's order.
static std::atomic<int> g_split_max{-2};
static int split_max_batch() {
  int v = g_split_max.load(std::memory_order_relaxed);
  if (v == -2) {
    const char *e = getenv("MOSFHET_HIP_SPLIT_MAX");
    v = e ? atoi(e) : -1;
    if (v < -1) v = -1;
    g_split_max.store(v, std::memory_order_relaxed);
  }
  return v < 0 ? device_cus() / 2 : v;
}
extern "C" int mosfhet_hip_set_split_max_batch(int max_batch) {
  g_split_max = max_batch < -1 ? -1 : max_batch;
  return MOSFHET_HIP_OK;
}
// MOSFHET_HIP_KS_WORDS: from how many ciphertexts a table key switch with 2 - 4 digit bits takes the word-lane kernel (keyswitch_words_kernels.h); 0 = never
extern "C" int mosfhet_hip_set_ks_words(int min_count) {
  g_ks_words_min = min_count < 0 ? 17 : min_count;
  return MOSFHET_HIP_OK;
}
// The launch plan of the word-lane key switch for a shape (no device needed): plan[8] = applies (0 / 1 at the current MOSFHET_HIP_KS_WORDS), positions per stage, stages per
// input word, LDS-DMA requests per wavefront and stage, LDS bytes, workgroups, ciphertext groups, input-word splits.  For the CPU test that sweeps the shapes.
extern "C" int mosfhet_hip_ks_words_plan(int count, int n_in, int row, int t, int base_bit, int compressed_lwe, long long plan[8]) {
  if (!plan || count < 1 || n_in < 1 || row < 1 || t < 1 || base_bit < 1 || base_bit > 8) return fail(MOSFHET_HIP_EINVAL, "ks_words_plan: bad shape");
  plan[0] = ks_words_applies(count, n_in, row, t, base_bit, compressed_lwe != 0, compressed_lwe ? row - 1 : 0) ? 1 : 0;
  for (int k = 1; k < 8; k++) plan[k] = 0;
  if (base_bit < 2 || base_bit > 4) return MOSFHET_HIP_OK;
  const KsWordsPlan p = ks_words_plan(count > 8192 ? 8192 : count, n_in, row, t, base_bit);
  plan[1] = p.JB; plan[2] = p.chunks; plan[3] = p.pf; plan[4] = (long long)p.lds;
  plan[5] = (long long)((p.wblocks * p.splits + 7) / 8) * 8 * p.groups; plan[6] = p.groups; plan[7] = p.splits;
  return MOSFHET_HIP_OK;
}
// wavefronts of table_ks_words_kernel whose bounded wait on the workgroup's LDS counters ran out since the library was loaded (synchronises the device): 0 unless
// something is broken -- the waits involve the eight resident wavefronts of one workgroup only.  For tests and the soak.
extern "C" int mosfhet_hip_ks_words_gave_up(mosfhet_hip_ctx_t ctx, unsigned int *count) {
  if (!ctx || !count) return fail(MOSFHET_HIP_EINVAL, "ks_words_gave_up: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(count, HIP_SYMBOL(ksw_gave_up_count), sizeof(unsigned int)));
  return MOSFHET_HIP_OK;
}
// MOSFHET_HIP_SPLIT_LIMIT: how long the first workgroup of a pair waits for the second before it takes the bootstrap alone, in 10 ns ticks (default 2 ms);
// 0 = every bootstrap alone (test switch: same bits)
static std::atomic<int> g_split_limit{-1};
static int split_wait_limit() {
  int v = g_split_limit.load(std::memory_order_relaxed);
  if (v < 0) {
    const char *e = getenv("MOSFHET_HIP_SPLIT_LIMIT");
    v = e ? atoi(e) : 200000;
    if (v < 0) v = 0;
    g_split_limit.store(v, std::memory_order_relaxed);
  }
  return v;
}
extern "C" int mosfhet_hip_set_split_wait_limit(int ticks) {
  g_split_limit = ticks < 0 ? 0 : ticks;
  return MOSFHET_HIP_OK;
}
// Exchange slots + pairing words of the split launches of this host thread: one set per (device, stream) -- launches on one stream are ordered, launches on
// different streams may overlap and must not share slots.  SPLIT_SETS sets per thread; a further stream takes over the least recently used one (after a device synchronisation).
constexpr int SPLIT_SETS = 8;
struct SplitSet { int device = -1; hipStream_t stream = nullptr; d2 *xbuf = nullptr; unsigned int *state = nullptr; int cap = 0, last_count = 0; unsigned long long used = 0; };
struct SplitSets {
  SplitSet set[SPLIT_SETS];
  ~SplitSets() {
    for (SplitSet &x : set)
      if (x.xbuf && hipSetDevice(x.device) == hipSuccess) { (void)hipFree(x.xbuf); (void)hipFree(x.state); }
  }
};
static thread_local SplitSets t_split;
static thread_local SplitSet *t_split_last = nullptr;
static SplitSet *split_set(hipStream_t s, int count) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  static thread_local unsigned long long tick = 0;
  SplitSet *free_slot = nullptr, *oldest = nullptr;
  for (SplitSet &x : t_split.set) {
    if (x.xbuf && x.device == dev && x.stream == s && x.cap >= count) { x.used = ++tick; return &x; }
    if (!x.xbuf && !free_slot) free_slot = &x;
    if (x.xbuf && x.device == dev && x.cap >= count && (!oldest || x.used < oldest->used)) oldest = &x;
  }
  if (!free_slot) {
    // more streams than sets: the least recently used set of this device changes hands once everything queued on the device has finished (rare, and better than
    // a kernel choice -- and with it the low bits -- that depends on how many streams a thread has used)
    if (!oldest || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    oldest->stream = s;
    oldest->used = ++tick;
    return oldest;
  }
  free_slot->used = ++tick;
  const int cap = device_cus() / 2 > count ? device_cus() / 2 : count;
  // per bootstrap: [2 receivers][2 step parities][M] slots of the external product's exchange + [2 parities][M] of the Galois bootstrap's key-switch exchange
  if (hipMalloc((void **)&free_slot->xbuf, (size_t)cap * 6 * 1024 * sizeof(d2)) != hipSuccess) { (void)hipGetLastError(); free_slot->xbuf = nullptr; return nullptr; }
  if (hipMalloc((void **)&free_slot->state, (size_t)cap * sizeof(unsigned int)) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(free_slot->xbuf);
    free_slot->xbuf = nullptr;
    return nullptr;
  }
  free_slot->device = dev;
  free_slot->stream = s;
  free_slot->cap = cap;
  return free_slot;
}
template <int LL, int BB>
static int launch_split(const PbsParams &p, int count, hipStream_t s, bool *taken) {
  using F = Fft2048L;
  *taken = false;
  SplitSet *set = split_set(s, count);
  if (!set) return MOSFHET_HIP_OK;   // no slots for this stream: the caller goes on to the one-CU kernel
  constexpr size_t lds = sizeof(d2) * ((size_t)2 * F::XCH_SLOTS + (size_t)4 * F::M) + sizeof(uint64_t) * 2 * F::N;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_split_kernel<F, LL, BB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t words = (size_t)count * 2 * 2 * F::M * 2;
  hipLaunchKernelGGL(split_prepare_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint64_t *>(set->xbuf), words, set->state, count);
  SplitParams sp;
  sp.xbuf = set->xbuf;
  sp.state = set->state;
  sp.count = count;
  sp.limit = split_wait_limit();
  hipLaunchKernelGGL((pbs_split_kernel<F, LL, BB>), dim3((unsigned)(16 * ((count + 7) / 8))), dim3(2 * F::THREADS), lds, s, p, sp);
  HIP_TRY(hipGetLastError());
  set->last_count = count;
  t_split_last = set;
  *taken = true;
  return MOSFHET_HIP_OK;
}
template <int LL, int BB>
static int launch_ga_split(const GaParams &g, int count, hipStream_t s, bool *taken) {
  using F = Fft2048L;
  *taken = false;
  SplitSet *set = split_set(s, count);
  if (!set) return MOSFHET_HIP_OK;
  constexpr size_t lds = sizeof(d2) * ((size_t)2 * F::XCH_SLOTS + (size_t)4 * F::M) + sizeof(uint64_t) * 2 * F::N;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_ga_split_kernel<F, LL, BB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t words = (size_t)count * 6 * F::M * 2;
  hipLaunchKernelGGL(split_prepare_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint64_t *>(set->xbuf), words, set->state, count);
  SplitParams sp;
  sp.xbuf = set->xbuf;
  sp.state = set->state;
  sp.count = count;
  sp.limit = split_wait_limit();
  hipLaunchKernelGGL((pbs_ga_split_kernel<F, LL, BB>), dim3((unsigned)(16 * ((count + 7) / 8))), dim3(2 * F::THREADS), lds, s, g, sp);
  HIP_TRY(hipGetLastError());
  set->last_count = count;
  t_split_last = set;
  *taken = true;
  return MOSFHET_HIP_OK;
}

// How the bootstraps of this host thread's LAST split launch were taken (synchronises its stream): by a pair of workgroups, or alone by the first to arrive.
// For tests and bench.py.  EINVAL when the thread has not made one.
extern "C" int mosfhet_hip_split_last_launch(int *count, int *paired, int *alone) {
  SplitSet *set = t_split_last;
  if (!set || !count || !paired || !alone) return fail(MOSFHET_HIP_EINVAL, "split_last_launch: no split launch on this thread yet");
  HIP_TRY(hipSetDevice(set->device));
  HIP_TRY(hipStreamSynchronize(set->stream));
  std::vector<unsigned int> st((size_t)set->last_count);
  HIP_TRY(hipMemcpy(st.data(), set->state, st.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
  *count = set->last_count;
  *paired = *alone = 0;
  for (unsigned int v : st) {
    if (v == 2u) ++*paired;
    else ++*alone;   // 3: the wait ran out; 0: the wait limit was 0 (every bootstrap alone)
  }
  return MOSFHET_HIP_OK;
}

template <class F>
static int launch_wide_team_f(int l, int Bg, const PbsParams &p, int count, hipStream_t s, bool bounded) {
  if constexpr (F::N == 2048) {
    // at most half the CUs' worth of ciphertexts, even gadget lengths up to 6: two CUs per bootstrap (pbs_split_kernel)
    if ((l == 2 || l == 4 || l == 6) && count <= split_max_batch()) {
      bool taken = false;
      int rc;
      if (l == 4) rc = Bg == 9 ? launch_split<4, 9>(p, count, s, &taken) : launch_split<4, 0>(p, count, s, &taken);
      else if (l == 6) rc = Bg == 7 ? launch_split<6, 7>(p, count, s, &taken) : launch_split<6, 0>(p, count, s, &taken);   // 6 x 2^7: the radix-integer application's set
      else rc = launch_split<2, 0>(p, count, s, &taken);
      if (rc != MOSFHET_HIP_OK || taken) return rc;
    }
    // one workgroup per CU (137 KiB of LDS each): up to as many ciphertexts as the device has CUs; beyond that pbs_wide_team_kernel runs two workgroups per CU
    if (wide_pairs_enabled() && l % 2 == 0 && count <= device_cus()) {
      if (l == 4 && Bg == 9) return launch_wide_pair<4, 9>(p, count, s);
      if (l == 2) return launch_wide_pair<2, 0>(p, count, s);
      if (l == 4) return launch_wide_pair<4, 0>(p, count, s);
      if (l == 6) return launch_wide_pair<6, 0>(p, count, s);
    }
  }
  if constexpr (kCompileTimeGadgets<F>) {
    if (l == 4 && Bg == 9) return launch_wide_team<F, 4, 9>(p, count, s);
    if (l == 2 && Bg == 8 && (bounded || F::N != 1024)) return launch_wide_team<F, 2, 8>(p, count, s);
  }
  if (l == 1 && Bg == 23) return launch_wide_team<F, 1, 23>(p, count, s);
  if (l == 1) return launch_wide_team<F, 1, 0>(p, count, s);
  if (l == 2) return launch_wide_team<F, 2, 0>(p, count, s);
  if (l == 3) return launch_wide_team<F, 3, 0>(p, count, s);
  if (l == 4) return launch_wide_team<F, 4, 0>(p, count, s);
  if (l == 5) return launch_wide_team<F, 5, 0>(p, count, s);
  if (l == 6) return launch_wide_team<F, 6, 0>(p, count, s);
  return fail(MOSFHET_HIP_EINVAL, "l = %d not instantiated", l);
}

static int bootstrap_unfolded(const char *who, mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out, const uint64_t *d_tv, int tv_count,
                              const uint64_t *d_in, int count, int pre, int kappa, int theta, int torus_base, int extract, int skip_init, void *stream, int rows);

static int bootstrap_common(const char *who, mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                            const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count, int pre, int kappa,
                            int theta, int torus_base, int extract, int skip_init, void *stream, int rows = 1) {
  if (!ctx || !bsk || count < 0) return fail(MOSFHET_HIP_EINVAL, "%s: bad argument", who);
  if (count == 0) return MOSFHET_HIP_OK;   // an empty batch is a no-op (its buffers may be null)
  if (!d_out || !d_in) return fail(MOSFHET_HIP_EINVAL, "%s: null buffer", who);
  if (!skip_init && rows == 1 && (!d_tv || (tv_count != 1 && tv_count != count)))
    return fail(MOSFHET_HIP_EINVAL, "%s: tv_count must be 1 or count (got %d, count %d)", who, tv_count, count);
  if (!skip_init && torus_base < 1) return fail(MOSFHET_HIP_EINVAL, "%s: torus_base %d", who, torus_base);
  if (pre && (kappa < 0 || kappa > 63 || theta < 0 || theta > 52)) return fail(MOSFHET_HIP_EINVAL, "%s: kappa/theta out of range", who);
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (bsk->unfolding > 1) return bootstrap_unfolded(who, ctx, bsk, d_out, d_tv, tv_count, d_in, count, pre, kappa, theta, torus_base, extract, skip_init, stream, rows);
  if (bsk->general) {
    if (rows != 1) return fail(MOSFHET_HIP_EINVAL, "%s: TRGSW accumulators need a tuned ring (k = 1, N in 1024 / 2048 / 4096)", who);
    GeneralParams g;
    int rc = general_twiddles(ctx, bsk->N, &g.tw);
    if (rc || (rc = general_lds(pbs_general_kernel, bsk->N))) return rc;
    const int k = bsk->k, N = bsk->N;
    // accumulators ([count][k+1][N] words; the output buffer itself when the rotated TRLWE is what is asked for) and products ([count][k+1][N/2] complex)
    uint64_t *scratch = nullptr;
    const size_t acc_words = extract ? (size_t)count * (k + 1) * N : 0, prod_words = (size_t)count * (k + 1) * N;
    if ((rc = pool_get(ctx->device, POOL_BSK, acc_words + prod_words, &scratch))) return rc;
    g.bk = bsk->d_bk; g.in = d_in; g.tv = d_tv; g.out = d_out;
    g.acc = extract ? scratch : d_out;
    g.prod = reinterpret_cast<d2 *>(scratch + acc_words);
    g.tv_stride = (tv_count == 1) ? 0 : (long long)(k + 1) * N;
    g.n = bsk->n; g.k = k; g.N = N; g.logM = ilog2(N / 2); g.l = bsk->l; g.Bg_bit = bsk->Bg_bit;
    g.pre = pre; g.kappa = kappa; g.theta = theta;
    g.prec_offset = skip_init ? 0 : (uint64_t)((int64_t)(18446744073709551616.0 * (1. / (4 * (double)torus_base))));
    g.extract = extract; g.skip_init = skip_init;
    hipLaunchKernelGGL(pbs_general_kernel, dim3((unsigned)count), dim3(GEN_THREADS), (size_t)8 * N, pick(ctx, stream), g);
    HIP_TRY(hipGetLastError());
    return MOSFHET_HIP_OK;
  }
  PbsParams p;
  p.bk = bsk->d_bk;
  p.tw = bsk->N == 1024 ? ctx->tw1024 : (bsk->N == 2048 ? ctx->tw2048 : ctx->tw4096);
  p.in = d_in;
  p.tv = d_tv;
  p.out = d_out;
  p.tv_stride = (tv_count == 1) ? 0 : (long long)(bsk->k + 1) * bsk->N;
  p.n = bsk->n;
  p.Bg_bit = bsk->Bg_bit;
  p.pre = pre;
  p.kappa = kappa;
  p.theta = theta;
  // src/misc.c:13-15 double2torus(1 / (4 torus_base))
  p.prec_offset = skip_init ? 0 : (uint64_t)((int64_t)(18446744073709551616.0 * (1. / (4 * (double)torus_base))));
  p.extract = extract;
  p.skip_init = skip_init;
  p.count = bsk->bytes > ((size_t)96 << 20) ? -1 : 0;   // launch hint: split into residency rounds (launch_pbs)
  p.rows = rows;
  // small batches: the latency-oriented team kernel (one workgroup of 2l wavefronts per ciphertext), N = 1024
  if (bsk->N == 1024 && rows == 1 && count <= team_max_batch() && bsk->l <= 4) {
    hipStream_t s = pick(ctx, stream);
    const int l = bsk->l, Bg = bsk->Bg_bit;
#define TEAM_LAUNCH(LL, BB) hipLaunchKernelGGL((pbs_team_kernel<LL, BB>), dim3((unsigned)count), dim3(64 * 2 * LL), 0, s, p)
    if (l == 2 && Bg == 8 && bsk->owns) TEAM_LAUNCH(2, 8);
    else if (l == 4 && Bg == 9) TEAM_LAUNCH(4, 9);
    else if (l == 1) TEAM_LAUNCH(1, 0);
    else if (l == 2) TEAM_LAUNCH(2, 0);
    else if (l == 3) TEAM_LAUNCH(3, 0);
    else TEAM_LAUNCH(4, 0);
#undef TEAM_LAUNCH
    HIP_TRY(hipGetLastError());
    return MOSFHET_HIP_OK;
  }
  // (count = workgroups: ciphertexts x accumulator rows; N = 4096: 136 KiB of LDS, one workgroup per CU -- half the batch)
  // N = 1024: what pbs_team_kernel (above) does not take -- gadgets longer than 4, TRGSW accumulator rows
  if (bsk->N == 1024 && (bsk->l > 4 || rows > 1) && count <= team_max_batch()) return launch_wide_team_f<Fft1024>(bsk->l, bsk->Bg_bit, p, count, pick(ctx, stream), bsk->owns);
  if (bsk->N == 2048 && count <= wide_team_max_batch()) return launch_wide_team_f<Fft2048>(bsk->l, bsk->Bg_bit, p, count, pick(ctx, stream), bsk->owns);
  if (bsk->N == 4096 && count <= wide_team_max_batch() / 2) return launch_wide_team_f<Fft4096>(bsk->l, bsk->Bg_bit, p, count, pick(ctx, stream), bsk->owns);
  int rc_pbs = MOSFHET_HIP_OK;
  RING_DISPATCH(ctx, bsk->N, rc_pbs = launch_pbs_f<F>(bsk->l, bsk->Bg_bit, p, count, pick(ctx, stream), bsk->owns));
  return rc_pbs;
}

extern "C" int mosfhet_hip_programmable_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                                        const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count,
                                                        int precision, int kappa, int theta, void *stream) {
  if (precision < 1 || precision > 30) return fail(MOSFHET_HIP_EINVAL, "programmable_bootstrap: precision %d", precision);
  return bootstrap_common("programmable_bootstrap", ctx, bsk, d_out, d_tv, tv_count, d_in, count, 1, kappa, theta,
                          1 << (precision - 1), 1, 0, stream);
}

extern "C" int mosfhet_hip_functional_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                                      const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count,
                                                      int torus_base, void *stream) {
  return bootstrap_common("functional_bootstrap", ctx, bsk, d_out, d_tv, tv_count, d_in, count, 0, 0, 0, torus_base, 1, 0, stream);
}

extern "C" int mosfhet_hip_functional_bootstrap_wo_extract_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                                                 const uint64_t *d_tv, int tv_count, const uint64_t *d_in,
                                                                 int count, int torus_base, void *stream) {
  return bootstrap_common("functional_bootstrap_wo_extract", ctx, bsk, d_out, d_tv, tv_count, d_in, count, 0, 0, 0, torus_base, 0, 0, stream);
}

extern "C" int mosfhet_hip_blind_rotate_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_acc,
                                              const uint64_t *d_in, int count, void *stream) {
  return bootstrap_common("blind_rotate", ctx, bsk, d_acc, nullptr, 0, d_in, count, 0, 0, 0, 1, 0, 1, stream);
}

// external product / CMUX of the general-ring path (general_kernels.h): one workgroup per unit, products in the calling thread's pool
static int external_product_general(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, int key_index, uint64_t *d_out, const uint64_t *d_in, const uint64_t *d_in0,
                                    int count, void *stream) {
  const int k = bsk->k, N = bsk->N, M = N / 2;
  const d2 *tw = nullptr;
  int rc = general_twiddles(ctx, N, &tw);
  if (rc || (rc = general_lds(external_product_general_kernel, N))) return rc;
  uint64_t *prod = nullptr;
  if ((rc = pool_get(ctx->device, POOL_BSK, (size_t)count * (k + 1) * N, &prod))) return rc;
  const d2 *g = bsk->d_bk + (size_t)key_index * (k + 1) * bsk->l * (k + 1) * M;
  hipLaunchKernelGGL(external_product_general_kernel, dim3((unsigned)count), dim3(GEN_THREADS), (size_t)8 * N, pick(ctx, stream), g, (size_t)0, tw, d_in, d_out,
                     reinterpret_cast<d2 *>(prod), k, N, ilog2(M), bsk->l, bsk->Bg_bit, d_in0);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_external_product_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, int key_index, uint64_t *d_out,
                                                  const uint64_t *d_in, int count, void *stream) {
  if (!ctx || !bsk || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0 || key_index < 0 || key_index >= bsk->n)
    return fail(MOSFHET_HIP_EINVAL, "external_product: bad argument");
  if (bsk->unfolding > 1) return fail(MOSFHET_HIP_EINVAL, "external_product: an unfolded key has no DFT entries");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (bsk->general) return external_product_general(ctx, bsk, key_index, d_out, d_in, nullptr, count, stream);
  const d2 *row = bsk->d_bk + (size_t)key_index * (2 * bsk->l * 2 * (bsk->N / 2));
  hipStream_t s = pick(ctx, stream);
  RING_DISPATCH(ctx, bsk->N, launch_external_product<F>(bsk->l, bsk->Bg_bit, s, row, TW, d_in, d_out, count, (size_t)0, (size_t)2 * F::N, nullptr, nullptr, bsk->owns));
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

// ---- polynomial-level entry points ----
extern "C" int mosfhet_hip_torus_to_dft_batch(mosfhet_hip_ctx_t ctx, double *d_out, const uint64_t *d_in, int N, int count, void *stream) {
  if (!ctx || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0) return fail(MOSFHET_HIP_EINVAL, "torus_to_dft: bad argument");
  if (!ring_ok(N) && (N < 256 || N > 16384 || (N & (N - 1)))) return fail(MOSFHET_HIP_EINVAL, "torus_to_dft: N = %d not supported (a power of two in 256 .. 16384)", N);
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ring_ok(N)) {   // general ring: natural slot order (general_kernels.h)
    const d2 *tw = nullptr;
    int rc = general_twiddles(ctx, N, &tw);
    if (rc || (rc = general_lds(torus_to_dft_general_kernel, N))) return rc;
    hipLaunchKernelGGL(torus_to_dft_general_kernel, dim3((unsigned)count), dim3(GEN_THREADS), (size_t)8 * N, pick(ctx, stream), d_in, (d2 *)d_out, tw, N, ilog2(N / 2));
    HIP_TRY(hipGetLastError());
    return MOSFHET_HIP_OK;
  }
  RING_DISPATCH(ctx, N, hipLaunchKernelGGL(torus_to_dft_kernel<F>, dim3(count), dim3(F::THREADS), 0, pick(ctx, stream), d_in, (d2 *)d_out, TW));
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_dft_to_torus_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const double *d_in, int N, int count, void *stream) {
  if (!ctx || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0) return fail(MOSFHET_HIP_EINVAL, "dft_to_torus: bad argument");
  if (!ring_ok(N) && (N < 256 || N > 16384 || (N & (N - 1)))) return fail(MOSFHET_HIP_EINVAL, "dft_to_torus: N = %d not supported (a power of two in 256 .. 16384)", N);
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ring_ok(N)) {
    const d2 *tw = nullptr;
    int rc = general_twiddles(ctx, N, &tw);
    if (rc || (rc = general_lds(dft_to_torus_general_kernel, N))) return rc;
    hipLaunchKernelGGL(dft_to_torus_general_kernel, dim3((unsigned)count), dim3(GEN_THREADS), (size_t)8 * N, pick(ctx, stream), (const d2 *)d_in, d_out, tw, N, ilog2(N / 2));
    HIP_TRY(hipGetLastError());
    return MOSFHET_HIP_OK;
  }
  RING_DISPATCH(ctx, N, hipLaunchKernelGGL(dft_to_torus_kernel<F>, dim3(count), dim3(F::THREADS), 0, pick(ctx, stream), (const d2 *)d_in, d_out, TW));
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_dft_mul_batch(mosfhet_hip_ctx_t ctx, double *d_out, const double *d_a, const double *d_b, int N,
                                         int count, int addto, void *stream) {
  if (!ctx || (count > 0 && !d_out) || (count > 0 && !d_a) || (count > 0 && !d_b) || count < 0 || N < 2) return fail(MOSFHET_HIP_EINVAL, "dft_mul: bad argument");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t total = (size_t)count * (N / 2);
  hipLaunchKernelGGL(dft_mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pick(ctx, stream), (d2 *)d_out,
                     (const d2 *)d_a, (const d2 *)d_b, total, addto);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

// ---- LWE key switch ----
extern "C" int mosfhet_hip_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_ksk, int n_in, int n_out,
                                      int t, int base_bit) {
  if (!ctx || !out || !h_ksk || n_in < 1 || n_out < 1 || t < 1 || base_bit < 1 || base_bit > 8 || t * base_bit >= 64)
    return fail(MOSFHET_HIP_EINVAL, "ksk_create: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  std::unique_ptr<mosfhet_hip_ksk> k_owner(new mosfhet_hip_ksk());
  mosfhet_hip_ksk *k = k_owner.get();
  k->ctx = ctx; k->n_in = n_in; k->n_out = n_out; k->t = t; k->base_bit = base_bit;
  k->row = n_out + 1; k->b_word = n_out;
  k->bytes = (size_t)n_in * t * ((1u << base_bit) - 1) * (n_out + 1) * sizeof(uint64_t);
  HIP_TRY(hipMalloc((void **)&k->d_ksk, k->bytes + ksw_slack_bytes(k->row)));   // (+ slack: the word-lane key switch reads past the last row, keyswitch_words_kernels.h)
  HIP_TRY(hipMemcpy(k->d_ksk, h_ksk, k->bytes, hipMemcpyHostToDevice));
  *out = k_owner.release();
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_ksk_destroy(mosfhet_hip_ksk_t ksk) {
  if (!ksk) return MOSFHET_HIP_OK;
  delete ksk;
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_tlwe_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, uint64_t *d_out, const uint64_t *d_in,
                                                int count, void *stream) {
  if (!ctx || !ksk || count < 0) return fail(MOSFHET_HIP_EINVAL, "tlwe_keyswitch: bad argument");
  if (count == 0) return MOSFHET_HIP_OK;
  if (!d_out || !d_in) return fail(MOSFHET_HIP_EINVAL, "tlwe_keyswitch: null buffer");
  HIP_TRY(hipSetDevice(ctx->device));
  if (ksk->b_word != ksk->n_out) return fail(MOSFHET_HIP_EINVAL, "tlwe_keyswitch: this key is a packing (LWE -> TRLWE) key");
  HIP_TRY(launch_tlwe_keyswitch(ksk->d_ksk, d_out, (size_t)ksk->row, d_in, (size_t)ksk->n_in + 1, count, ksk->n_in, ksk->row, ksk->b_word, ksk->t,
                                ksk->base_bit, tl_ws(ctx->device), pick(ctx, stream), ksk->compressed, ksk->seed));
  return MOSFHET_HIP_OK;
}

// ---- glue entry points ----
extern "C" int mosfhet_hip_trlwe_extract_tlwe_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const uint64_t *d_in, int N, int idx, int count,
                                                    void *stream) {
  if (!ctx || (count > 0 && !d_out) || (count > 0 && !d_in) || N < 1 || idx < 0 || idx >= N || count < 0) return fail(MOSFHET_HIP_EINVAL, "trlwe_extract: bad argument");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(trlwe_extract_kernel, dim3((N + 255) / 256, count), dim3(256), 0, pick(ctx, stream), d_out, (size_t)N + 1, d_in,
                     (size_t)2 * N, N, idx);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_trlwe_extract_tlwe_k_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const uint64_t *d_in, int k, int N, int idx, int count, void *stream) {
  if (!ctx || (count > 0 && !d_out) || (count > 0 && !d_in) || k < 1 || N < 1 || idx < 0 || idx >= N || count < 0) return fail(MOSFHET_HIP_EINVAL, "trlwe_extract: bad argument");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(trlwe_extract_k_kernel, dim3((unsigned)(((size_t)k * N + 255) / 256), count), dim3(256), 0, pick(ctx, stream), d_out, (size_t)k * N + 1, d_in,
                     (size_t)(k + 1) * N, N, k, idx);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_tlwe_addto_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const uint64_t *d_in, int n, int count, void *stream) {
  if (!ctx || (count > 0 && !d_out) || (count > 0 && !d_in) || n < 0 || count < 0) return fail(MOSFHET_HIP_EINVAL, "tlwe_addto: bad argument");
  const size_t words = (size_t)count * (n + 1);
  if (!words) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(words_addto_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, pick(ctx, stream), d_out, d_in, words);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

static int bsk_scratch(mosfhet_hip_bsk_t bsk, size_t words, uint64_t **out) { return pool_get(bsk->ctx->device, POOL_BSK, words, out); }

extern "C" int mosfhet_hip_full_domain_functional_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t ksk,
                                                                  uint64_t *d_out, const uint64_t *d_tv, int tv_count,
                                                                  const uint64_t *d_in, int count, int precision, void *stream) {
  if (!ctx || !bsk || !ksk || (count > 0 && !d_out) || (count > 0 && !d_tv) || (count > 0 && !d_in) || count < 0) return fail(MOSFHET_HIP_EINVAL, "fdfb: bad argument");
  if (precision < 1 || precision > 30) return fail(MOSFHET_HIP_EINVAL, "fdfb: precision %d", precision);
  if (ksk->n_in != bsk->k * bsk->N || ksk->n_out != bsk->n)
    return fail(MOSFHET_HIP_EINVAL, "fdfb: key-switch key is %d -> %d, expected %d -> %d", ksk->n_in, ksk->n_out, bsk->k * bsk->N, bsk->n);
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const int N = bsk->N, n = bsk->n, k = bsk->k;
  const size_t w_tv = (size_t)(k + 1) * N, w_sign = (size_t)count * (k * N + 1), w_in2 = (size_t)count * (n + 1);
  // (general keys keep their accumulators in the POOL_BSK slot: this composition's temporaries take another one)
  uint64_t *tv_sign = nullptr;
  int rc = bsk->general ? pool_get(bsk->ctx->device, POOL_EXT0, w_tv + w_sign + w_in2, &tv_sign) : bsk_scratch(bsk, w_tv + w_sign + w_in2, &tv_sign);
  if (rc) return rc;
  uint64_t *ct_sign = tv_sign + w_tv, *in2 = ct_sign + w_sign;
  hipStream_t s = pick(ctx, stream);
  // src/bootstrap.c:525-527: sign = 2^62 - 2^(62 - precision), constant test vector (mask components zero, body constant)
  const uint64_t sign = (1ull << 62) - (1ull << (62 - precision));
  if (k > 1) HIP_TRY(hipMemsetAsync(tv_sign, 0, (size_t)(k - 1) * N * sizeof(uint64_t), s));
  hipLaunchKernelGGL(trlwe_constant_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tv_sign + (size_t)(k - 1) * N, N, sign);
  if ((rc = mosfhet_hip_functional_bootstrap_batch(ctx, bsk, ct_sign, tv_sign, 1, d_in, count, 1 << (precision - 1), stream))) return rc;
  hipLaunchKernelGGL(tlwe_add_to_b_kernel, dim3((count + 255) / 256), dim3(256), 0, s, ct_sign, count, (size_t)k * N + 1, (uint64_t)0 - sign);
  if ((rc = mosfhet_hip_tlwe_keyswitch_batch(ctx, ksk, in2, ct_sign, count, stream))) return rc;
  if ((rc = mosfhet_hip_tlwe_addto_batch(ctx, in2, d_in, n, count, stream))) return rc;
  return mosfhet_hip_functional_bootstrap_batch(ctx, bsk, d_out, d_tv, tv_count, in2, count, 1 << precision, stream);
}

extern "C" int mosfhet_hip_multivalue_bootstrap_CLOT21_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                                             const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count,
                                                             int torus_base, int n_luts, void *stream) {
  if (!ctx || !bsk || (count > 0 && !d_out) || (count > 0 && !d_tv) || (count > 0 && !d_in) || count < 0 || torus_base < 1 || n_luts < 1)
    return fail(MOSFHET_HIP_EINVAL, "multivalue_CLOT21: bad argument");
  const int N = bsk->N, k = bsk->k;
  if (N % (n_luts * torus_base)) return fail(MOSFHET_HIP_EINVAL, "multivalue_CLOT21: N not divisible by n_luts * torus_base");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  uint64_t *rotated = nullptr;
  const size_t trlwe = (size_t)(k + 1) * N, tlwe = (size_t)k * N + 1;
  // (general keys keep their accumulators in the POOL_BSK slot: this composition's temporaries take another one)
  int rc = bsk->general ? pool_get(ctx->device, POOL_EXT0, (size_t)count * trlwe, &rotated) : bsk_scratch(bsk, (size_t)count * trlwe, &rotated);
  if (rc) return rc;
  if ((rc = mosfhet_hip_functional_bootstrap_wo_extract_batch(ctx, bsk, rotated, d_tv, tv_count, d_in, count, torus_base * n_luts, stream)))
    return rc;
  const int slot = N / (n_luts * torus_base);
  for (int i = 0; i < n_luts; i++) {
    if (k == 1)
      hipLaunchKernelGGL(trlwe_extract_kernel, dim3((N + 255) / 256, count), dim3(256), 0, pick(ctx, stream), d_out + (size_t)i * tlwe, (size_t)n_luts * tlwe, rotated, trlwe, N,
                         i * slot);
    else   // src/trlwe.c:540-552 over the k mask polynomials
      hipLaunchKernelGGL(trlwe_extract_k_kernel, dim3((k * N + 255) / 256, count), dim3(256), 0, pick(ctx, stream), d_out + (size_t)i * tlwe, (size_t)n_luts * tlwe, rotated,
                         trlwe, N, k, i * slot);
  }
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

// ---- Galois automorphisms ----
extern "C" int mosfhet_hip_trlwe_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t *out, const uint64_t *h_rows, int entries, int N, int t,
                                            int base_bit) {
  if (!ctx || !out || !h_rows || entries < 1) return fail(MOSFHET_HIP_EINVAL, "trlwe_ksk_create: bad argument");
  if (!ring_ok(N)) return fail(MOSFHET_HIP_EINVAL, "trlwe_ksk_create: N = %d not supported (1024, 2048, 4096)", N);
  if (t < 1 || base_bit < 1 || t * base_bit >= 64) return fail(MOSFHET_HIP_EINVAL, "trlwe_ksk_create: bad t = %d base_bit = %d", t, base_bit);
  HIP_TRY(hipSetDevice(ctx->device));
  std::unique_ptr<mosfhet_hip_gak> g_owner(new mosfhet_hip_gak());
  mosfhet_hip_gak *g = g_owner.get();
  g->ctx = ctx; g->N = N; g->t = t; g->base_bit = base_bit; g->entries = entries;
  const size_t polys = (size_t)entries * t * 2;
  g->bytes = polys * N * sizeof(double);
  DevBuf tmp;                                      // torus-domain rows, released on every way out
  HIP_TRY(tmp.alloc(g->bytes));
  uint64_t *d_tmp = tmp.as<uint64_t>();
  HIP_TRY(hipMalloc((void **)&g->d_ak, g->bytes));
  HIP_TRY(hipMemcpy(d_tmp, h_rows, g->bytes, hipMemcpyHostToDevice));
  RING_DISPATCH(ctx, N, hipLaunchKernelGGL(torus_to_dft_kernel<F>, dim3((unsigned)polys), dim3(F::THREADS), 0, nullptr, d_tmp, g->d_ak, TW));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(nullptr));
  *out = g_owner.release();
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_gak_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t *out, const uint64_t *h_ak, int N, int t, int base_bit) {
  return mosfhet_hip_trlwe_ksk_create(ctx, out, h_ak, N, N, t, base_bit);
}

extern "C" int mosfhet_hip_gak_destroy(mosfhet_hip_gak_t gak) {
  if (!gak) return MOSFHET_HIP_OK;
  delete gak;
  return MOSFHET_HIP_OK;
}

template <class F, int L, int BG>
static void launch_ga(const GaParams &g_in, int count, hipStream_t s) {
  GaParams g = g_in;
  // Galois bootstraps at N = 2048 (the bootstrap key does not fit the L2s): one launch per residency round, its teams re-aligned on the bootstrap-key walk, like launch_pbs
  const int round = F::THREADS > 64 && F::N == 2048 && g.mode == 0 ? round_chunk(F::THREADS) : 0;
  const bool pace = round > 0 && count >= 64 && pace_every() > 0;
  if (pace) {
    g.p.pace_every = pace_every();
    g.p.pace_limit = pace_limit();
  }
  if (round > 0 && count > round) {
    const size_t out_row = g.p.extract ? (size_t)F::N + 1 : (size_t)2 * F::N;
    for (int lo = 0; lo < count; lo += round) {
      GaParams q = g;
      q.p.pace = pace ? pace_slot(s) : nullptr;
      q.p.in = g.p.in + (size_t)lo * (g.p.n + 1);
      q.p.out = g.p.out + (size_t)lo * out_row;
      q.p.tv = g.p.tv ? g.p.tv + (size_t)lo * g.p.tv_stride : g.p.tv;
      const int c = count - lo < round ? count - lo : round;
      hipLaunchKernelGGL((pbs_ga_kernel<F, L, BG>), dim3((unsigned)c), dim3(F::THREADS), 0, s, q);
    }
    return;
  }
  if (pace) g.p.pace = pace_slot(s);
  hipLaunchKernelGGL((pbs_ga_kernel<F, L, BG>), dim3((unsigned)count), dim3(F::THREADS), 0, s, g);
}

template <class F>
static int launch_ga_f(int l, int Bg_bit, const GaParams &g, int count, hipStream_t s) {
  // at most half the CUs' worth of ciphertexts at N = 2048, l = 4: two CUs per bootstrap (pbs_ga_split_kernel; the external products in pbs_split_kernel's summation order)
  if constexpr (F::N == 2048) {
    if (g.mode == 0 && l == 4 && count <= split_max_batch()) {
      GaParams gs = g;
      gs.p.Bg_bit = Bg_bit;
      bool taken = false;
      const int rc = Bg_bit == 9 ? launch_ga_split<4, 9>(gs, count, s, &taken) : launch_ga_split<4, 0>(gs, count, s, &taken);
      if (rc != MOSFHET_HIP_OK || taken) return rc;
    }
  }
  // few ciphertexts: two transform teams per ciphertext (pbs_ga_wide_kernel, bit-identical; the switch-overs of the plain bootstrap's latency kernels)
  if constexpr (F::N <= 2048) {
    if (g.mode == 0 && count <= (F::N == 1024 ? team_max_batch() : wide_team_max_batch())) {
      constexpr size_t lds = sizeof(d2) * (size_t)2 * F::XCH_SLOTS + sizeof(uint64_t) * 2 * F::N;
      if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pbs_ga_wide_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GaParams gw = g;
      gw.p.Bg_bit = Bg_bit;
      hipLaunchKernelGGL(pbs_ga_wide_kernel<F>, dim3((unsigned)count), dim3(2 * F::THREADS), lds, s, gw, l);
      HIP_TRY(hipGetLastError());
      return MOSFHET_HIP_OK;
    }
  }
  if constexpr (std::is_same<F, Fft2048>::value) {
    // lvl2: the pass twiddles in LDS (Fft2048L) take the kernel out of scratch (68 bytes -> 0, 250 registers; same speed: experiments/README.md round 4).
    // The run-time-gadget instantiations spill with either transform and stay where they were.
    if (l == 4 && Bg_bit == 9) { launch_ga<Fft2048L, 4, 9>(g, count, s); HIP_TRY(hipGetLastError()); return MOSFHET_HIP_OK; }
  }
  if (kCompileTimeGadgets<F> && l == 2 && Bg_bit == 8) { if constexpr (kCompileTimeGadgets<F>) launch_ga<F, 2, 8>(g, count, s); }
  else if (kCompileTimeGadgets<F> && !std::is_same<F, Fft2048>::value && l == 4 && Bg_bit == 9) {   // (N = 2048: left above with the LDS-twiddle transform)
    if constexpr (kCompileTimeGadgets<F> && !std::is_same<F, Fft2048>::value) launch_ga<F, 4, 9>(g, count, s);
  }
  else if (l == 1) launch_ga<F, 1, 0>(g, count, s);
  else if (l == 2) launch_ga<F, 2, 0>(g, count, s);
  else if (l == 3) launch_ga<F, 3, 0>(g, count, s);
  else if (l == 4) launch_ga<F, 4, 0>(g, count, s);
  else if (l == 5) launch_ga<F, 5, 0>(g, count, s);
  else if (l == 6) launch_ga<F, 6, 0>(g, count, s);
  else return fail(MOSFHET_HIP_EINVAL, "l = %d not instantiated", l);
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_trlwe_eval_automorphism_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t gak, uint64_t *d_out,
                                                         const uint64_t *d_in, int gen, int count, void *stream) {
  if (!ctx || !gak || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0) return fail(MOSFHET_HIP_EINVAL, "eval_automorphism: bad argument");
  if (gen < 1 || gen >= 2 * gak->N || !(gen & 1)) return fail(MOSFHET_HIP_EINVAL, "eval_automorphism: generator %d must be odd and < 2N", gen);
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  GaParams g;
  memset(&g, 0, sizeof(g));
  g.p.tw = gak->N == 1024 ? ctx->tw1024 : (gak->N == 2048 ? ctx->tw2048 : ctx->tw4096);
  g.p.in = d_in;
  g.p.out = d_out;
  g.p.Bg_bit = gak->base_bit;
  g.ak = gak->d_ak;
  g.mode = 1;
  g.gen = gen;
  g.entry = -1;
  int rc_ga = MOSFHET_HIP_OK;
  RING_DISPATCH(ctx, gak->N, rc_ga = launch_ga_f<F>(gak->t, gak->base_bit, g, count, pick(ctx, stream)));
  return rc_ga;
}

extern "C" int mosfhet_hip_functional_bootstrap_ga_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t gak,
                                                         uint64_t *d_out, const uint64_t *d_tv, int tv_count, const uint64_t *d_in,
                                                         int count, int torus_base, int extract, void *stream) {
  TUNED_ONLY(bsk, "functional_bootstrap_ga");
  if (!ctx || !bsk || !gak || (count > 0 && !d_out) || (count > 0 && !d_tv) || (count > 0 && !d_in) || count < 0 || torus_base < 1)
    return fail(MOSFHET_HIP_EINVAL, "functional_bootstrap_ga: bad argument");
  if (tv_count != 1 && tv_count != count) return fail(MOSFHET_HIP_EINVAL, "functional_bootstrap_ga: tv_count must be 1 or count");
  if (bsk->unfolding > 1) return fail(MOSFHET_HIP_EINVAL, "functional_bootstrap_ga: needs a key without unfolding");
  if (gak->N != bsk->N || gak->t != bsk->l || gak->base_bit != bsk->Bg_bit)
    return fail(MOSFHET_HIP_EINVAL, "functional_bootstrap_ga: automorphism keys must use the bootstrap key's N, l, Bg_bit");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  GaParams g;
  memset(&g, 0, sizeof(g));
  g.p.bk = bsk->d_bk;
  g.p.tw = bsk->N == 1024 ? ctx->tw1024 : (bsk->N == 2048 ? ctx->tw2048 : ctx->tw4096);
  g.p.in = d_in;
  g.p.tv = d_tv;
  g.p.out = d_out;
  g.p.tv_stride = (tv_count == 1) ? 0 : (long long)2 * bsk->N;
  g.p.n = bsk->n;
  g.p.Bg_bit = bsk->Bg_bit;
  g.p.prec_offset = (uint64_t)((int64_t)(18446744073709551616.0 * (1. / (4 * (double)torus_base))));
  g.p.extract = extract ? 1 : 0;
  g.ak = gak->d_ak;
  g.mode = 0;
  int rc_ga = MOSFHET_HIP_OK;
  RING_DISPATCH(ctx, bsk->N, rc_ga = launch_ga_f<F>(bsk->l, bsk->Bg_bit, g, count, pick(ctx, stream)));
  return rc_ga;
}

// ---- FFT TRLWE key switches with run-time parameters, packing key switch, circuit bootstrap ----
static int launch_fft_ks(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t tks, const d2 *ks0, const d2 *ks1, uint64_t *d_out, size_t out_stride,
                         const uint64_t *d_in, size_t in_stride, int count, int mode, hipStream_t s) {
  RING_DISPATCH(ctx, tks->N, hipLaunchKernelGGL(trlwe_fft_keyswitch_kernel<F>, dim3(count), dim3(F::THREADS), 0, s, ks0, ks1, TW, d_in, in_stride, d_out,
                                                out_stride, tks->t, tks->base_bit, mode));
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_trlwe_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t tks, int entry, uint64_t *d_out, const uint64_t *d_in,
                                                 int count, void *stream) {
  if (!ctx || !tks || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0 || entry < 0 || entry >= tks->entries) return fail(MOSFHET_HIP_EINVAL, "trlwe_keyswitch: bad argument");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const d2 *k = tks->d_ak + (size_t)entry * tks->t * 2 * (tks->N / 2);
  return launch_fft_ks(ctx, tks, k, k, d_out, (size_t)2 * tks->N, d_in, (size_t)2 * tks->N, count, 0, pick(ctx, stream));
}

extern "C" int mosfhet_hip_trlwe_priv_keyswitch_2_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t tks, uint64_t *d_out, const uint64_t *d_in,
                                                        int count, void *stream) {
  if (!ctx || !tks || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0 || tks->entries != 2) return fail(MOSFHET_HIP_EINVAL, "priv_keyswitch_2: needs a 2-entry key set");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t esz = (size_t)tks->t * 2 * (tks->N / 2);
  return launch_fft_ks(ctx, tks, tks->d_ak, tks->d_ak + esz, d_out, (size_t)2 * tks->N, d_in, (size_t)2 * tks->N, count, 1, pick(ctx, stream));
}

extern "C" int mosfhet_hip_packing1_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_rows, int n, int N, int t,
                                               int base_bit) {
  // same table layout as the LWE key: rows of 2N words instead of n_out + 1, in.b lands on word N (b[0])
  int rc = mosfhet_hip_ksk_create(ctx, out, h_rows, n, 2 * N - 1, t, base_bit);
  if (rc) return rc;
  (*out)->b_word = N;
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_trlwe_packing1_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, uint64_t *d_out, const uint64_t *d_in,
                                                          int count, void *stream) {
  if (!ctx || !ksk || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0) return fail(MOSFHET_HIP_EINVAL, "packing1_keyswitch: bad argument");
  if (ksk->b_word == ksk->n_out) return fail(MOSFHET_HIP_EINVAL, "packing1_keyswitch: this key is an LWE -> LWE key");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(launch_tlwe_keyswitch(ksk->d_ksk, d_out, (size_t)ksk->row, d_in, (size_t)ksk->n_in + 1, count, ksk->n_in, ksk->row, ksk->b_word, ksk->t,
                                ksk->base_bit, tl_ws(ctx->device), pick(ctx, stream), ksk->compressed, ksk->seed));
  return MOSFHET_HIP_OK;
}

// circuit bootstraps: one packing switch for all l levels when that saves table sweeps (a tile of the switch is 512 ciphertexts wide); MOSFHET_HIP_CB_TOGETHER=0 / 1 forces
static bool cb_levels_together(int count, int l) {
  static std::atomic<int> v{-2};
  int r = v.load(std::memory_order_relaxed);
  if (r == -2) { const char *e = getenv("MOSFHET_HIP_CB_TOGETHER"); r = e ? atoi(e) : -1; v.store(r, std::memory_order_relaxed); }
  if (r >= 0) return r != 0 && l > 1;
  return l > 1 && ((size_t)l * count + 511) / 512 < (size_t)l * (((size_t)count + 511) / 512);
}
// circuit_bootstrap / FDFB KS21_2 (one bootstrap per level): the l bootstraps of an input side by side in row mode -- one launch, or residency rounds of whole inputs
// (launch_pbs) -- instead of l launches of fewer than a round's worth of ciphertexts (capi_ext.inc)
// Row mode belongs to the folded bootstrap kernels: an unfolded key (bootstrap_unfolded) keeps one launch per level, as the reference's
// functional_bootstrap dispatches on key->unfolding underneath circuit_bootstrap and the full-domain bootstraps (src/bootstrap.c:196-197,313-317,476-478).
static bool cb_bootstraps_together(mosfhet_hip_bsk_t bsk, int count, int l) {
  const char *e = getenv("MOSFHET_HIP_CB_TOGETHER");
  return bsk->unfolding == 1 && l > 1 && count < 1024 && !(e && atoi(e) == 0);   // (from 1024 on, l launches of count ARE l count / 1024 residency rounds)
}

// test vector of circuit_bootstrap_3 (src/bootstrap.c:350-355): 2l slots, slot l + i = 2^(64 - (i+1) Bg), slots < l zero
__global__ void circuit_bootstrap_lut_kernel(uint64_t *__restrict__ tv, int N, int l, int Bg_bit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int slot = i / (N / (2 * l));
  tv[i] = 0;
  tv[N + i] = slot >= l ? (1ull << (64 - (slot - l + 1) * Bg_bit)) : 0;
}

extern "C" int mosfhet_hip_circuit_bootstrap_3_batch_ev(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t kska, mosfhet_hip_ksk_t kskb,
                                                        uint64_t *d_out, const uint64_t *d_in, int count, void *stream, void *const *level_done) {
  TUNED_ONLY(bsk, "circuit_bootstrap_3");
  if (!ctx || !bsk || !kska || !kskb || (count > 0 && !d_out) || (count > 0 && !d_in) || count < 0) return fail(MOSFHET_HIP_EINVAL, "circuit_bootstrap_3: bad argument");
  const int N = bsk->N, l = bsk->l;
  if (kska->entries != 2 || kska->N != N) return fail(MOSFHET_HIP_EINVAL, "circuit_bootstrap_3: kska must be the 2-entry private key-switch set for N");
  if (kskb->row != 2 * N || kskb->b_word != N || kskb->n_in != N) return fail(MOSFHET_HIP_EINVAL, "circuit_bootstrap_3: kskb must be a packing key N -> TRLWE(N)");
  if (N % (2 * l)) return fail(MOSFHET_HIP_EINVAL, "circuit_bootstrap_3: N not divisible by 2l");
  if (count == 0) return MOSFHET_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  // Few ciphertexts (fewer than one 512-wide tile of the packing switch): every level's switch would sweep the whole table (3 - 6 GB) for a fraction of a tile, so the
  // l levels' samples are extracted first and switched TOGETHER -- ceil(l count / 512) sweeps instead of l -- into a staging block, from where the private switch reads
  // them and a strided copy puts them into their TRGSW rows.  Large batches keep one switch per level: same number of sweeps, and the levels finish one after the
  // other (what the level events are for).  Same operations on the same words either way: same bits.
  const bool together = cb_levels_together(count, l);
  const size_t w_tv = (size_t)2 * N, w_acc = (size_t)count * 2 * N, w_ext = ((size_t)count * (N + 1) * (together ? l : 1) + 1) & ~(size_t)1 /* keeps the staging block 16-byte aligned */,
               w_stage = together ? (size_t)l * count * 2 * N : 0;
  uint64_t *tv = nullptr;
  int rc = bsk_scratch(bsk, w_tv + w_acc + w_ext + w_stage, &tv);
  if (rc) return rc;
  uint64_t *acc = tv + w_tv, *ext = acc + w_acc, *stage = ext + w_ext;
  hipStream_t s = pick(ctx, stream);
  hipLaunchKernelGGL(circuit_bootstrap_lut_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tv, N, l, bsk->Bg_bit);
  if ((rc = mosfhet_hip_functional_bootstrap_wo_extract_batch(ctx, bsk, acc, tv, 1, d_in, count, 2 * l, stream))) return rc;
  const int slot = N / (2 * l);
  const size_t trgsw = (size_t)2 * l * 2 * N, esz = (size_t)kska->t * 2 * (N / 2);
  if (together) {
    for (int i = 0; i < l; i++)
      hipLaunchKernelGGL(trlwe_extract_kernel, dim3((N + 255) / 256, count), dim3(256), 0, s, ext + (size_t)i * count * (N + 1), (size_t)N + 1, acc, (size_t)2 * N, N, i * slot);
    HIP_TRY(launch_tlwe_keyswitch(kskb->d_ksk, stage, (size_t)2 * N, ext, (size_t)N + 1, l * count, N, 2 * N, N, kskb->t, kskb->base_bit, tl_ws(ctx->device), s, kskb->compressed,
                                  kskb->seed));
    for (int i = 0; i < l; i++) {
      const uint64_t *sw = stage + (size_t)i * count * 2 * N;
      uint64_t *row_b = d_out + (size_t)(l + i) * 2 * N, *row_a = d_out + (size_t)i * 2 * N;
      HIP_TRY(hipMemcpy2DAsync(row_b, trgsw * sizeof(uint64_t), sw, (size_t)2 * N * sizeof(uint64_t), (size_t)2 * N * sizeof(uint64_t), (size_t)count, hipMemcpyDeviceToDevice, s));
      if ((rc = launch_fft_ks(ctx, kska, kska->d_ak, kska->d_ak + esz, row_a, trgsw, sw, (size_t)2 * N, count, 1, s))) return rc;
      if (level_done && level_done[i]) HIP_TRY(hipEventRecord((hipEvent_t)level_done[i], s));
    }
    HIP_TRY(hipGetLastError());
    return MOSFHET_HIP_OK;
  }
  for (int i = 0; i < l; i++) {
    hipLaunchKernelGGL(trlwe_extract_kernel, dim3((N + 255) / 256, count), dim3(256), 0, s, ext, (size_t)N + 1, acc, (size_t)2 * N, N, i * slot);
    uint64_t *row_b = d_out + (size_t)(l + i) * 2 * N, *row_a = d_out + (size_t)i * 2 * N;
    HIP_TRY(launch_tlwe_keyswitch(kskb->d_ksk, row_b, trgsw, ext, (size_t)N + 1, count, N, 2 * N, N, kskb->t, kskb->base_bit, tl_ws(ctx->device), s, kskb->compressed, kskb->seed));
    if ((rc = launch_fft_ks(ctx, kska, kska->d_ak, kska->d_ak + esz, row_a, trgsw, row_b, trgsw, count, 1, s))) return rc;
    if (level_done && level_done[i]) HIP_TRY(hipEventRecord((hipEvent_t)level_done[i], s));   // rows i and l + i of every output are final
  }
  HIP_TRY(hipGetLastError());
  return MOSFHET_HIP_OK;
}

extern "C" int mosfhet_hip_circuit_bootstrap_3_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t kska, mosfhet_hip_ksk_t kskb,
                                                     uint64_t *d_out, const uint64_t *d_in, int count, void *stream) {
  return mosfhet_hip_circuit_bootstrap_3_batch_ev(ctx, bsk, kska, kskb, d_out, d_in, count, stream, nullptr);
}

// ---- timing hook ----
extern "C" int mosfhet_hip_time_programmable_bootstrap(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                                       const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count,
                                                       int precision, int reps, void *stream, float *ms_per_launch) {
  if (!ms_per_launch || reps < 1) return fail(MOSFHET_HIP_EINVAL, "time_pbs: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = pick(ctx, stream);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, s));
  for (int r = 0; r < reps; r++) {
    int rc = mosfhet_hip_programmable_bootstrap_batch(ctx, bsk, d_out, d_tv, tv_count, d_in, count, precision, 0, 0, s);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_per_launch = ms / (float)reps;
  return MOSFHET_HIP_OK;
}

#include "capi_ext.inc"
#include "capi_dft.inc"
#include "capi_vec.inc"
