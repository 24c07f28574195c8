// general_kernels.h -- the bootstrap path for ANY power-of-two ring degree and k >= 1 (gfx950).
//
// The reference is generic in both (src/trgsw.c:385-423 loops over k, src/fft/ffnt/ffnt.c serves any power of two); the tuned kernels of
// bootstrap_kernels.h exist for k = 1 and N in {1024, 2048, 4096}, the rings of every parameter set the reference ships (test/tests.c:37-62,
// test/benchmark.c:53-75).  Everything else runs here: one workgroup per ciphertext (or polynomial), the transform in LDS in NATURAL recursion order
// -- level by level, the loop structure of oracle/oracle_fft.c with one workgroup barrier per level --, accumulator and products in global memory.
// Correctness path, not a tuned one: per element the butterflies, the multiply-accumulate chain over the TRGSW rows (row q*l + j, q = accumulator
// component, levels in increasing j: src/trgsw.c:393-419) and the rounding are the oracle's, so results are bit-identical to it.
// DFT-domain data of this path is in natural slot order (slot j = the oracle's index j), [polynomial][N/2] complex; a key made here is used here only.
//   blind_rotate                  [src/bootstrap.c:107-122]        pbs_general_kernel
//   functional / programmable bootstrap (+ wo_extract)  [:192-220]  pbs_general_kernel (pre-processing, test-vector rotation, sample extract fused)
//   trgsw_mul_trlwe_DFT + trlwe_from_DFT [src/trgsw.c:385-423, src/trlwe.c:629-634]   external_product_general_kernel
//   polynomial_torus_to_DFT / _DFT_to_torus [src/polynomial.c:359-375]                torus_to_dft_general_kernel / dft_to_torus_general_kernel
#pragma once
#include "negacyclic_fft.h"

namespace mosfhet {

constexpr int GEN_THREADS = 256;

// in-place forward transform of z[M] (LDS), natural order: oracle_fft.c:fwd_inplace.  Ends with a barrier.
__device__ __forceinline__ void general_forward(d2 *z, const d2 *__restrict__ tw, int logM) {
  const int M = 1 << logM;
  for (int lev = 0; lev < logM; lev++) {
    const int half = M >> (lev + 1);
    for (int b = threadIdx.x; b < M / 2; b += GEN_THREADS) {
      const int nu = b / half, pos = b - nu * half;
      const d2 s = tw[(1 << lev) - 1 + nu];
      d2 *lo = z + (size_t)nu * 2 * half + pos, *hi = lo + half;
      const d2 a = *lo, c = *hi;
      double ar = a.x, ai = a.y, br = c.x, bi = c.y;
      bf_fwd(ar, ai, br, bi, s.x, s.y);
      *lo = d2{ar, ai};
      *hi = d2{br, bi};
    }
    workgroup_sync();
  }
}

// in-place inverse, UNSCALED: oracle_fft.c:inv_inplace
__device__ __forceinline__ void general_inverse(d2 *z, const d2 *__restrict__ tw, int logM) {
  const int M = 1 << logM;
  for (int lev = logM - 1; lev >= 0; lev--) {
    const int half = M >> (lev + 1);
    for (int b = threadIdx.x; b < M / 2; b += GEN_THREADS) {
      const int nu = b / half, pos = b - nu * half;
      const d2 s = tw[(1 << lev) - 1 + nu];
      d2 *lo = z + (size_t)nu * 2 * half + pos, *hi = lo + half;
      const d2 u = *lo, v = *hi;
      double ur = u.x, ui = u.y, vr = v.x, vi = v.y;
      bf_inv(ur, ui, vr, vi, s.x, s.y);
      *lo = d2{ur, ui};
      *hi = d2{vr, vi};
    }
    workgroup_sync();
  }
}

// coefficient i of poly * X^a for run-time N (a in [0, 2N)); src/polynomial.c:184-199
__device__ __forceinline__ uint64_t rot_coeff_rt(const uint64_t *poly, int i, int a_lo, bool flip, int N) {
  const int src = i - a_lo;
  const bool neg = (src < 0) != flip;
  const uint64_t v = poly[src & (N - 1)];
  return neg ? (0 - v) : v;
}

__device__ __forceinline__ double digit_rt(uint64_t dd, int lv, int Bg_bit) {
  const uint32_t u = (uint32_t)(dd >> (64 - (lv + 1) * Bg_bit)) & ((1u << Bg_bit) - 1);
  return (double)((int)u - (1 << (Bg_bit - 1)));
}

struct GeneralParams {
  const d2 *__restrict__ bk;         // [n][(k+1)l][k+1][M] complex, natural slot order
  const d2 *tw;                      // twiddle table of the ring, M - 1 entries
  const uint64_t *__restrict__ in;   // [B][n+1]
  const uint64_t *__restrict__ tv;   // [tv_count][k+1][N]
  uint64_t *out;                     // extract: [B][kN+1]; else [B][k+1][N]
  uint64_t *acc;        // [B][k+1][N] accumulators (== out when !extract)
  d2 *__restrict__ prod;             // [B][k+1][M] products of the current step
  long long tv_stride;
  int n, k, N, logM, l, Bg_bit;
  int pre, kappa, theta;
  uint64_t prec_offset;
  int extract, skip_init;
};

__global__ __launch_bounds__(GEN_THREADS) void pbs_general_kernel(GeneralParams p) {
  extern __shared__ __attribute__((aligned(16))) d2 z[];   // [M]
  const int N = p.N, M = N / 2, k = p.k, l = p.l, Bg = p.Bg_bit, log2N2 = p.logM + 2;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  uint64_t *acc = p.acc + b * (size_t)(k + 1) * N;
  d2 *prod = p.prod + b * (size_t)(k + 1) * M;
  auto pre = [&](uint64_t x) -> uint64_t {          // src/bootstrap.c:213-217
    if (!p.pre) return x;
    const uint64_t rnd = 1ull << (64 - log2N2 + p.theta - 1);
    const uint64_t msk = ~((1ull << (64 - log2N2 + p.theta)) - 1);
    return ((x << p.kappa) + rnd) & msk;
  };
  auto modsw = [&](uint64_t x) -> int { return (int)((x + (1ull << (63 - log2N2))) >> (64 - log2N2)); };   // src/misc.c:18-22
  if (!p.skip_init) {
    // src/bootstrap.c:194-195: acc = tv * X^(2N - bbar)
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const int rot = (2 * N - modsw(pre(ct[p.n]) + p.prec_offset)) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = threadIdx.x; x < (k + 1) * N; x += GEN_THREADS) {
      const int c = x / N, i = x - c * N;
      acc[x] = rot_coeff_rt(tv + (size_t)c * N, i, a_lo, flip, N);
    }
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - l * Bg);
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg);
  const RoundCtx scale(p.logM);
  const size_t trgsw_sz = (size_t)(k + 1) * l * (k + 1) * M;
  for (int i = 0; i < p.n; i++) {
    const int abar = modsw(pre(ct[i]));
    if (abar == 0) continue;   // src/bootstrap.c:114
    const d2 *__restrict__ bkrow = p.bk + (size_t)i * trgsw_sz;
    const int a_lo = abar & (N - 1);
    const bool flip = (abar & N) != 0;
    for (int q = 0; q <= k; q++) {
      const uint64_t *src = acc + (size_t)q * N;
      for (int j = 0; j < l; j++) {
        // digit j of (X^abar - 1) acc[q]  (src/polynomial.c:220-235, 74-89), folded: z_x = d_x + i d_(x+M)
        for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
          const uint64_t lo = rot_coeff_rt(src, x, a_lo, flip, N) - src[x] + off, hi = rot_coeff_rt(src, x + M, a_lo, flip, N) - src[x + M] + off;
          z[x] = d2{digit_rt(lo, j, Bg), digit_rt(hi, j, Bg)};
        }
        workgroup_sync();
        general_forward(z, p.tw, p.logM);
        const d2 *__restrict__ row = bkrow + (size_t)(q * l + j) * (k + 1) * M;
        const bool first = (q == 0 && j == 0);
        for (int c = 0; c <= k; c++)
          for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
            const d2 d = z[x], kk = row[(size_t)c * M + x];
            d2 o = first ? d2{0.0, 0.0} : prod[(size_t)c * M + x];
            o.x = __builtin_fma(-d.y, kk.y, __builtin_fma(d.x, kk.x, o.x));
            o.y = __builtin_fma(d.y, kk.x, __builtin_fma(d.x, kk.y, o.y));
            prod[(size_t)c * M + x] = o;     // same thread reads and writes slot x of every row: no cross-thread hazard in global memory
          }
        workgroup_sync();
      }
    }
    for (int c = 0; c <= k; c++) {
      for (int x = threadIdx.x; x < M; x += GEN_THREADS) z[x] = prod[(size_t)c * M + x];
      workgroup_sync();
      general_inverse(z, p.tw, p.logM);
      uint64_t *dst = acc + (size_t)c * N;
      for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
        const d2 v = z[x];
        dst[x] += round_mod_2_64(v.x, scale);           // src/trlwe.c:629-634 + :437-439
        dst[x + M] += round_mod_2_64(v.y, scale);
      }
      workgroup_sync();
    }
  }
  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0: a[c N + j] = acc_c[0] (j = 0), -acc_c[N - j] otherwise; b = acc_k[0]
    uint64_t *dst = p.out + b * (size_t)(k * N + 1);
    for (int x = threadIdx.x; x < k * N; x += GEN_THREADS) {
      const int c = x / N, j = x - c * N;
      const uint64_t *a = acc + (size_t)c * N;
      dst[x] = j == 0 ? a[0] : (0 - a[N - j]);
    }
    if (threadIdx.x == 0) dst[(size_t)k * N] = acc[(size_t)k * N];
  }
}

// out[b] = TRGSW (.) in[b] for a batch against one TRGSW_DFT (key_stride = 0) or one per unit; in0 != nullptr: the CMUX
// out[b] = in0[b] + TRGSW (.) (in[b] - in0[b]) (trlwe_sub + trgsw_mul_trlwe_DFT + trlwe_from_DFT + trlwe_add, applications/leveled_lut/vertical_packing.c:24-33;
// out may be in0: a block reads its in0 words for the digits before it writes anything, and the last read of a word is the add that overwrites it)
__global__ __launch_bounds__(GEN_THREADS) void external_product_general_kernel(const d2 *__restrict__ trgsw, size_t key_stride, const d2 *__restrict__ tw,
                                                                            const uint64_t *in, uint64_t *out, d2 *__restrict__ prod_all,
                                                                            int k, int N, int logM, int l, int Bg, const uint64_t *in0) {
  extern __shared__ __attribute__((aligned(16))) d2 z[];
  const int M = N / 2;
  const size_t b = blockIdx.x;
  const uint64_t *src_all = in + b * (size_t)(k + 1) * N;
  const uint64_t *base_all = in0 ? in0 + b * (size_t)(k + 1) * N : nullptr;
  d2 *prod = prod_all + b * (size_t)(k + 1) * M;
  const d2 *__restrict__ g = trgsw + b * key_stride;
  uint64_t off = 1ull << (63 - l * Bg);
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg);
  const RoundCtx scale(logM);
  for (int q = 0; q <= k; q++) {
    const uint64_t *src = src_all + (size_t)q * N;
    const uint64_t *sub = base_all ? base_all + (size_t)q * N : nullptr;
    for (int j = 0; j < l; j++) {
      for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
        const uint64_t lo = sub ? src[x] - sub[x] : src[x], hi = sub ? src[x + M] - sub[x + M] : src[x + M];
        z[x] = d2{digit_rt(lo + off, j, Bg), digit_rt(hi + off, j, Bg)};
      }
      workgroup_sync();
      general_forward(z, tw, logM);
      const d2 *__restrict__ row = g + (size_t)(q * l + j) * (k + 1) * M;
      const bool first = (q == 0 && j == 0);
      for (int c = 0; c <= k; c++)
        for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
          const d2 d = z[x], kk = row[(size_t)c * M + x];
          d2 o = first ? d2{0.0, 0.0} : prod[(size_t)c * M + x];
          o.x = __builtin_fma(-d.y, kk.y, __builtin_fma(d.x, kk.x, o.x));
          o.y = __builtin_fma(d.y, kk.x, __builtin_fma(d.x, kk.y, o.y));
          prod[(size_t)c * M + x] = o;
        }
      workgroup_sync();
    }
  }
  for (int c = 0; c <= k; c++) {
    for (int x = threadIdx.x; x < M; x += GEN_THREADS) z[x] = prod[(size_t)c * M + x];
    workgroup_sync();
    general_inverse(z, tw, logM);
    uint64_t *dst = out + (b * (size_t)(k + 1) + c) * N;
    const uint64_t *add = base_all ? base_all + (size_t)c * N : nullptr;
    for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
      const d2 v = z[x];
      const uint64_t r0 = round_mod_2_64(v.x, scale), r1 = round_mod_2_64(v.y, scale);
      dst[x] = add ? add[x] + r0 : r0;
      dst[x + M] = add ? add[x + M] + r1 : r1;
    }
    workgroup_sync();
  }
}

// polynomial_torus_to_DFT for a flat array, natural slot order (also trgsw_to_DFT of the key upload)
__global__ __launch_bounds__(GEN_THREADS) void torus_to_dft_general_kernel(const uint64_t *__restrict__ in, d2 *__restrict__ out, const d2 *__restrict__ tw, int N, int logM) {
  extern __shared__ __attribute__((aligned(16))) d2 z[];
  const int M = N / 2;
  const uint64_t *src = in + (size_t)blockIdx.x * N;
  for (int x = threadIdx.x; x < M; x += GEN_THREADS) z[x] = d2{torus_to_double(src[x]), torus_to_double(src[x + M])};
  workgroup_sync();
  general_forward(z, tw, logM);
  d2 *dst = out + (size_t)blockIdx.x * M;
  for (int x = threadIdx.x; x < M; x += GEN_THREADS) dst[x] = z[x];
}

__global__ __launch_bounds__(GEN_THREADS) void dft_to_torus_general_kernel(const d2 *__restrict__ in, uint64_t *__restrict__ out, const d2 *__restrict__ tw, int N, int logM) {
  extern __shared__ __attribute__((aligned(16))) d2 z[];
  const int M = N / 2;
  const d2 *src = in + (size_t)blockIdx.x * M;
  for (int x = threadIdx.x; x < M; x += GEN_THREADS) z[x] = src[x];
  workgroup_sync();
  general_inverse(z, tw, logM);
  const RoundCtx scale(logM);
  uint64_t *dst = out + (size_t)blockIdx.x * N;
  for (int x = threadIdx.x; x < M; x += GEN_THREADS) {
    const d2 v = z[x];
    dst[x] = round_mod_2_64(v.x, scale);
    dst[x + M] = round_mod_2_64(v.y, scale);
  }
}

}  // namespace mosfhet
