// bootstrap_kernels.h -- device code of the programmable-bootstrap hot path (gfx950).
//
// Replaces, fused in ONE persistent kernel per batch (reference file:line in brackets):
//   programmable_bootstrap pre-processing            [src/bootstrap.c:208-217]
//   functional_bootstrap_wo_extract: acc = tv X^-b   [src/bootstrap.c:192-195, src/polynomial.c:184-199]
//   blind_rotate: n CMUX steps                        [src/bootstrap.c:107-122]
//     (X^a - 1) acc                                   [src/polynomial.c:220-235]
//     trgsw_mul_trlwe_DFT = digit extraction + (k+1)l forward transforms + complex MACs against BK_i
//                                                     [src/trgsw.c:385-423, src/polynomial.c:74-89,379-426]
//     trlwe_from_DFT + trlwe_addto                    [src/trlwe.c:629-634,437-439]
//   trlwe_extract_tlwe(acc, 0)                        [src/trlwe.c:540-552]
//
// Work decomposition (N = 1024, k = 1): one 64-lane wavefront (= one workgroup) per ciphertext.  The
// accumulator (2 x 1024 Torus64 = 16 KiB) lives in LDS for the whole bootstrap; a second 8 KiB LDS region is
// the transpose buffer of the transform.  Per CMUX step and decomposed polynomial r = (component p, level j):
//   * lanes read acc[p] and its rotation straight from LDS in the transform's input layout
//     (lane t owns coefficients t + 64 m and t + 64 m + 512), subtract, slice digit j, convert to double;
//   * forward transform in registers (negacyclic_fft.h);
//   * every lane multiplies ITS 8 frequency slots by the matching slots of bootstrap-key row r for both
//     output components and accumulates in registers -- slot order is the same for the key and the data,
//     so the MAC is lane-local and the key is read with fully coalesced 16-byte loads
//     (bk[i][r][c][m][lane], 1 KiB per wave instruction);
//   after all (k+1) l rows: two inverse transforms, round to Torus64 mod 2^64, acc += result (in LDS).
// All workgroups walk the key rows in the same order, so a 64 KiB row is fetched from HBM about once per
// XCD and then served from that XCD's L2 / the Infinity Cache.
#pragma once
#include "negacyclic_fft.h"

namespace mosfhet {

struct PbsParams {
  const d2 *__restrict__ bk;         // [n][(k+1)l][k+1][8][64] complex, slot order
  const d2 *__restrict__ tw;         // twiddle table, M - 1 entries
  const uint64_t *__restrict__ in;   // [B][n+1] input TLWE samples (a..., b)
  const uint64_t *__restrict__ tv;   // [tv_count][k+1][N] test vectors
  uint64_t *__restrict__ out;        // [B][kN+1] (extract) or [B][k+1][N] (wo_extract)
  long long tv_stride;               // in Torus words between consecutive ciphertexts' test vectors (0 = shared)
  int n, Bg_bit;
  int pre;                           // 1: apply programmable_bootstrap's ((x << kappa) + rnd) & mask
  int kappa, theta;
  uint64_t prec_offset;              // double2torus(1 / (4 torus_base))
  int extract;                       // 1: write TLWE (sample extract at 0); 0: write the rotated TRLWE
  int skip_init;                     // 1: blind_rotate only -- acc is loaded from `out` as is
};

// src/misc.c:18-22 with log_scale = log2(2N)
template <int LOG2N>
__device__ __forceinline__ uint32_t modswitch(uint64_t x) {
  return (uint32_t)((x + (1ull << (63 - LOG2N))) >> (64 - LOG2N));
}

// src/bootstrap.c:213-217
__device__ __forceinline__ uint64_t pbs_pre(uint64_t x, const PbsParams &p, int log2N2) {
  if (!p.pre) return x;
  const uint64_t rnd = 1ull << (64 - log2N2 + p.theta - 1);
  const uint64_t msk = ~((1ull << (64 - log2N2 + p.theta)) - 1);
  return ((x << p.kappa) + rnd) & msk;
}

// coefficient i of poly * X^a  (a in [0, 2N)), poly in LDS/global; src/polynomial.c:184-199
template <int N>
__device__ __forceinline__ uint64_t rot_coeff(const uint64_t *poly, int i, int a_lo, bool flip) {
  const int src = i - a_lo;
  const bool neg = (src < 0) != flip;
  const uint64_t v = poly[src & (N - 1)];
  return neg ? (0 - v) : v;
}

// One CMUX step on the LDS-resident accumulator: acc += BK_i (.) ((X^abar - 1) acc).
// L = gadget levels.  bkrow = bk + i * (2 L * 2 * 512).
template <int L>
__device__ __forceinline__ void cmux_step_1024(uint64_t (*acc)[1024], d2 *xch, const Fft1024 &fft,
                                               const d2 *__restrict__ bkrow, int abar, int Bg_bit, int lane) {
  constexpr int N = 1024, M = 512;
  const int a_lo = abar & (N - 1);
  const bool flip = (abar & N) != 0;
  // src/polynomial.c:74-89: offset = 2^(63 - L Bg) + sum_{i<L} 2^(63 - i Bg)
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const uint64_t mask = (1ull << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);

  double o_re[2][8], o_im[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }

  // one decomposed polynomial r = p * L + lv at a time (kept as a real loop: bounded register pressure)
#pragma unroll 1
  for (int r = 0; r < 2 * L; r++) {
    const int p = r / L, lv = r - p * L;
    const d2 *__restrict__ row = bkrow + (size_t)r * (2 * M);
    // issue this row's key loads first: their latency hides behind the transform
    d2 kv[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) kv[c][m] = row[c * M + m * 64 + lane];

    const int shift = 64 - (lv + 1) * Bg_bit;
    const uint64_t *ap = acc[p];
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * 64 + lane;
      const uint64_t d_lo = rot_coeff<N>(ap, j, a_lo, flip) - ap[j];
      const uint64_t d_hi = rot_coeff<N>(ap, j + M, a_lo, flip) - ap[j + M];
      re[m] = (double)((int)(((d_lo + off) >> shift) & mask) - half);
      im[m] = (double)((int)(((d_hi + off) >> shift) & mask) - half);
    }
    fft.forward(re, im, xch, lane);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) {
        o_re[c][m] = __builtin_fma(-im[m], kv[c][m].y, __builtin_fma(re[m], kv[c][m].x, o_re[c][m]));
        o_im[c][m] = __builtin_fma(im[m], kv[c][m].x, __builtin_fma(re[m], kv[c][m].y, o_im[c][m]));
      }
  }
  // all reads of acc are done (same wave, program order); now the two inverse transforms update it
  const double scale = 0x1p-64 / (double)M;
#pragma unroll
  for (int c = 0; c < 2; c++) {
    fft.inverse(o_re[c], o_im[c], xch, lane);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * 64 + lane;
      acc[c][j] += round_mod_2_64(o_re[c][m], scale);
      acc[c][j + M] += round_mod_2_64(o_im[c][m], scale);
    }
  }
  wave_lds_sync();
}

template <int L>
__global__ __launch_bounds__(64, 2) void pbs_kernel_1024(PbsParams p) {
  constexpr int N = 1024, LOG2N2 = 11;
  __shared__ __attribute__((aligned(16))) uint64_t acc[2][N];
  __shared__ __attribute__((aligned(16))) d2 xch[512];
  const int lane = threadIdx.x;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);

  Fft1024 fft;
  fft.init(p.tw, lane);

  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
#pragma unroll
    for (int c = 0; c < 2; c++)
      for (int i = lane; i < N; i += 64) acc[c][i] = src[c * N + i];
  } else {
    // src/bootstrap.c:194-195: acc = tv * X^(2N - bbar)
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
#pragma unroll
    for (int c = 0; c < 2; c++)
      for (int i = lane; i < N; i += 64) acc[c][i] = rot_coeff<N>(tv + c * N, i, a_lo, flip);
  }
  wave_lds_sync();

  const size_t row_sz = (size_t)2 * L * 2 * 512;
  for (int i = 0; i < p.n; i++) {
    const int abar = (int)modswitch<LOG2N2>(pbs_pre(ct[i], p, LOG2N2));
    if (abar == 0) continue;  // src/bootstrap.c:114
    cmux_step_1024<L>(acc, xch, fft, p.bk + (size_t)i * row_sz, abar, p.Bg_bit, lane);
  }

  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0: a[0] = acc_a[0], a[j] = -acc_a[N - j]; b = acc_b[0]
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = lane; j < N; j += 64) dst[j] = (j == 0) ? acc[0][0] : (0 - acc[0][N - j]);
    if (lane == 0) dst[N] = acc[1][0];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
#pragma unroll
    for (int c = 0; c < 2; c++)
      for (int i = lane; i < N; i += 64) dst[c * N + i] = acc[c][i];
  }
}

// trgsw_to_DFT / polynomial_torus_to_DFT for a flat array of polynomials [src/trgsw.c:345-349,
// src/polynomial.c:368-375]: one wavefront per polynomial, output in slot order [m][lane].
__global__ __launch_bounds__(64) void torus_to_dft_kernel_1024(const uint64_t *__restrict__ in, d2 *__restrict__ out,
                                                              const d2 *__restrict__ tw) {
  constexpr int N = 1024, M = 512;
  __shared__ __attribute__((aligned(16))) d2 xch[512];
  const int lane = threadIdx.x;
  const uint64_t *src = in + (size_t)blockIdx.x * N;
  Fft1024 fft;
  fft.init(tw, lane);
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    re[m] = torus_to_double(src[m * 64 + lane]);
    im[m] = torus_to_double(src[m * 64 + lane + M]);
  }
  fft.forward(re, im, xch, lane);
  d2 *dst = out + (size_t)blockIdx.x * M;
#pragma unroll
  for (int m = 0; m < 8; m++) dst[m * 64 + lane] = d2{re[m], im[m]};
}

// polynomial_DFT_to_torus for a flat array [src/polynomial.c:359-366]
__global__ __launch_bounds__(64) void dft_to_torus_kernel_1024(const d2 *__restrict__ in, uint64_t *__restrict__ out,
                                                              const d2 *__restrict__ tw) {
  constexpr int N = 1024, M = 512;
  __shared__ __attribute__((aligned(16))) d2 xch[512];
  const int lane = threadIdx.x;
  const d2 *src = in + (size_t)blockIdx.x * M;
  Fft1024 fft;
  fft.init(tw, lane);
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; m++) { const d2 v = src[m * 64 + lane]; re[m] = v.x; im[m] = v.y; }
  fft.inverse(re, im, xch, lane);
  uint64_t *dst = out + (size_t)blockIdx.x * N;
  const double scale = 0x1p-64 / (double)M;
#pragma unroll
  for (int m = 0; m < 8; m++) {
    dst[m * 64 + lane] = round_mod_2_64(re[m], scale);
    dst[m * 64 + lane + M] = round_mod_2_64(im[m], scale);
  }
}

// polynomial_mul_DFT / polynomial_mul_addto_DFT on slot-ordered arrays [src/polynomial.c:379-426]
__global__ void dft_mul_kernel(d2 *__restrict__ out, const d2 *__restrict__ a, const d2 *__restrict__ b, size_t count,
                               int addto) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const d2 x = a[i], y = b[i];
  d2 o = addto ? out[i] : d2{0.0, 0.0};
  o.x = __builtin_fma(-x.y, y.y, __builtin_fma(x.x, y.x, o.x));
  o.y = __builtin_fma(x.y, y.x, __builtin_fma(x.x, y.y, o.y));
  out[i] = o;
}

}  // namespace mosfhet
