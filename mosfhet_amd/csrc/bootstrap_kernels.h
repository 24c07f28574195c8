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
// Work decomposition (N = 1024, k = 1): one 64-lane wavefront (= one workgroup) per ciphertext, resident for the
// whole bootstrap.  Accumulator component a lives in VGPRs (transform input layout: lane t owns coefficients
// t + 64 m and t + 64 m + 512), component b in LDS (8 KiB); a 9 KiB LDS buffer serves the transform's transposes
// and stages component a for its rotation.  Per CMUX step:
//   * per component p: (X^abar - 1) acc[p] + gadget offset once, keeping only the top L*Bg bits per coefficient
//     (one 32-bit word when L*Bg <= 32) -- the digits of every level are bit-fields of that word;
//   * per level: digits -> double (v_cvt_f64_i32), forward transform in registers (negacyclic_fft.h), then every
//     lane multiplies ITS 8 frequency slots by the matching slots of bootstrap-key row r for both output
//     components and accumulates in registers -- slot order is the same for the key and the data, so the MAC is
//     lane-local and the key is read with fully coalesced 16-byte loads (bk[i][r][c][m][lane], 1 KiB per wave
//     instruction), issued under the last transform pass;
//   * after all (k+1) l rows: two inverse transforms, round to Torus64 mod 2^64, acc += result.
// All workgroups walk the key rows in the same order, so a 64 KiB row is fetched from HBM about once and then
// served from the XCDs' L2s / the Infinity Cache.
#pragma once
#include <type_traits>

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

// Gadget decomposition state of one coefficient: the top L*Bg bits of (d + offset), from which digit j is a
// bit-field (src/polynomial.c:74-89: digit_j = ((d + off) >> (64 - (j+1) Bg)) & (2^Bg - 1)) - 2^(Bg-1)).
// With Bg known at compile time and L*Bg <= 32 it is one 32-bit register; otherwise the full 64-bit word is kept.
template <int L, int BG>
struct Digits {
  static constexpr bool kPacked = (BG > 0) && (L * BG <= 32);
  using word_t = typename std::conditional<kPacked, uint32_t, uint64_t>::type;
  static __device__ __forceinline__ word_t pack(uint64_t dd, int Bg_bit) {
    if constexpr (kPacked) return (uint32_t)(dd >> (64 - L * BG));
    else return dd;
  }
  // signed digit `lv` as a double (exact)
  static __device__ __forceinline__ double digit(word_t w, int lv, int Bg_bit) {
    if constexpr (kPacked) {
      const uint32_t u = (w >> ((L - 1 - lv) * BG)) & ((1u << BG) - 1);
      return (double)((int)u - (1 << (BG - 1)));
    } else {
      const int bg = BG > 0 ? BG : Bg_bit;
      const uint32_t u = (uint32_t)(w >> (64 - (lv + 1) * bg)) & ((1u << bg) - 1);
      return (double)((int)u - (1 << (bg - 1)));
    }
  }
};

// Digit words of one accumulator component.  `home` != nullptr: the component lives in LDS (al / ah unused);
// otherwise it lives in registers (al, ah) and is staged through the transpose buffer for the rotation.
template <int L, int BG>
__device__ __forceinline__ void cmux_digits_r(typename Digits<L, BG>::word_t (&w_lo)[8], typename Digits<L, BG>::word_t (&w_hi)[8],
                                              const uint64_t (&al)[8], const uint64_t (&ah)[8], const uint64_t *home, d2 *xch,
                                              int a_lo, bool flip, uint64_t off, int Bg_bit, int lane) {
  constexpr int N = 1024, M = 512;
  using D = Digits<L, BG>;
  // keep the rotated LDS addresses from being hoisted out of the component loop (they would be spilled there)
  asm volatile("" : "+s"(a_lo));
  if (home) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * 64 + lane;
      w_lo[m] = D::pack(rot_coeff<N>(home, j, a_lo, flip) - home[j] + off, Bg_bit);
      w_hi[m] = D::pack(rot_coeff<N>(home, j + M, a_lo, flip) - home[j + M] + off, Bg_bit);
    }
  } else {
    // stage the component (8 KiB) and read it back rotated by abar
    uint64_t *st = reinterpret_cast<uint64_t *>(xch);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      st[m * 64 + lane] = al[m];
      st[M + m * 64 + lane] = ah[m];
    }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * 64 + lane;
      w_lo[m] = D::pack(rot_coeff<N>(st, j, a_lo, flip) - al[m] + off, Bg_bit);
      w_hi[m] = D::pack(rot_coeff<N>(st, j + M, a_lo, flip) - ah[m] + off, Bg_bit);
    }
    wave_lds_sync();
  }
}

// The L rows of component p: digits -> forward transform -> MAC against key rows p*L .. p*L+L-1.
template <int L, int BG, bool KVHALF = false>
__device__ __forceinline__ void cmux_rows(const typename Digits<L, BG>::word_t (&w_lo)[8], const typename Digits<L, BG>::word_t (&w_hi)[8],
                                          int p, double (&o_re)[2][8], double (&o_im)[2][8], d2 *xch, const Fft1024 &fft,
                                          const d2 *__restrict__ bkrow, int Bg_bit, int lane) {
  constexpr int M = 512;
  using D = Digits<L, BG>;
#pragma unroll 1
  for (int lv = 0; lv < L; lv++) {
    const d2 *__restrict__ row = bkrow + (size_t)(p * L + lv) * (2 * M);
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      re[m] = D::digit(w_lo[m], lv, Bg_bit);
      im[m] = D::digit(w_hi[m], lv, Bg_bit);
    }
    fft.forward_ab(re, im, xch, lane);
    if constexpr (!KVHALF) {
      d2 kv[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) kv[c][m] = row[c * M + m * 64 + lane];
      fft.forward_c(re, im);
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) {
          o_re[c][m] = __builtin_fma(-im[m], kv[c][m].y, __builtin_fma(re[m], kv[c][m].x, o_re[c][m]));
          o_im[c][m] = __builtin_fma(im[m], kv[c][m].x, __builtin_fma(re[m], kv[c][m].y, o_im[c][m]));
        }
    } else {
      // register-lean form: component 0 of the key row is loaded under pass C, component 1 under the MAC of component 0
      d2 k0[8], k1[8];
#pragma unroll
      for (int m = 0; m < 8; m++) k0[m] = row[m * 64 + lane];
      fft.forward_c(re, im);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        k1[m] = row[M + m * 64 + lane];
        o_re[0][m] = __builtin_fma(-im[m], k0[m].y, __builtin_fma(re[m], k0[m].x, o_re[0][m]));
        o_im[0][m] = __builtin_fma(im[m], k0[m].x, __builtin_fma(re[m], k0[m].y, o_im[0][m]));
      }
#pragma unroll
      for (int m = 0; m < 8; m++) {
        o_re[1][m] = __builtin_fma(-im[m], k1[m].y, __builtin_fma(re[m], k1[m].x, o_re[1][m]));
        o_im[1][m] = __builtin_fma(im[m], k1[m].x, __builtin_fma(re[m], k1[m].y, o_im[1][m]));
      }
    }
  }
}

// The fused bootstrap kernel.  Accumulator placement: component a (acc[0]) in VGPRs in the transform's input
// layout (lane owns coefficients m*64+lane and m*64+lane+512: 32 VGPRs), component b (acc[1]) resident in LDS
// (8 KiB); with the 9 KiB transpose buffer that is 17 KiB of LDS and <= 256 VGPRs per wavefront, i.e. two
// wavefronts per SIMD / eight ciphertexts per CU.  (Both components in LDS cost 25 KiB -> 6 per CU and measured
// 22 % slower; both in registers spills.)
template <int L, int BG>
__global__ __launch_bounds__(64, 2) void pbs_kernel_1024(PbsParams p) {
  constexpr int N = 1024, M = 512, LOG2N2 = 11;
  __shared__ __attribute__((aligned(16))) d2 xch[Fft1024::XCH_SLOTS];
  __shared__ __attribute__((aligned(16))) uint64_t acc1[N];
  const int lane = threadIdx.x;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;

  Fft1024 fft;
  fft.init(p.tw, lane);

  uint64_t al[8], ah[8];
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] = src[m * 64 + lane];
      ah[m] = src[M + m * 64 + lane];
      acc1[m * 64 + lane] = src[N + m * 64 + lane];
      acc1[M + m * 64 + lane] = src[N + M + m * 64 + lane];
    }
  } else {
    // src/bootstrap.c:194-195: acc = tv * X^(2N - bbar), gathered straight from global memory
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] = rot_coeff<N>(tv, m * 64 + lane, a_lo, flip);
      ah[m] = rot_coeff<N>(tv, M + m * 64 + lane, a_lo, flip);
      acc1[m * 64 + lane] = rot_coeff<N>(tv + N, m * 64 + lane, a_lo, flip);
      acc1[M + m * 64 + lane] = rot_coeff<N>(tv + N, M + m * 64 + lane, a_lo, flip);
    }
  }
  wave_lds_sync();

  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const double scale = 0x1p-64 / (double)M;
  const size_t row_sz = (size_t)2 * L * 2 * 512;

  for (int i = 0; i < p.n; i++) {
    const int abar = (int)modswitch<LOG2N2>(pbs_pre(ct[i], p, LOG2N2));
    if (abar == 0) continue;  // src/bootstrap.c:114
    const d2 *__restrict__ bkrow = p.bk + (size_t)i * row_sz;
    const int a_lo = abar & (N - 1);
    const bool flip = (abar & N) != 0;
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
      typename Digits<L, BG>::word_t w_lo[8], w_hi[8];
      cmux_digits_r<L, BG>(w_lo, w_hi, al, ah, p ? acc1 : nullptr, xch, a_lo, flip, off, Bg_bit, lane);
      cmux_rows<L, BG, true>(w_lo, w_hi, p, o_re, o_im, xch, fft, bkrow, Bg_bit, lane);
    }
    fft.inverse(o_re[0], o_im[0], xch, lane);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] += round_mod_2_64(o_re[0][m], scale);
      ah[m] += round_mod_2_64(o_im[0][m], scale);
    }
    fft.inverse(o_re[1], o_im[1], xch, lane);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      acc1[m * 64 + lane] += round_mod_2_64(o_re[1][m], scale);
      acc1[M + m * 64 + lane] += round_mod_2_64(o_im[1][m], scale);
    }
    wave_lds_sync();
  }

  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0: a[0] = acc_a[0], a[j] = -acc_a[N - j]; b = acc_b[0]
    uint64_t *st = reinterpret_cast<uint64_t *>(xch);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      st[m * 64 + lane] = al[m];
      st[M + m * 64 + lane] = ah[m];
    }
    wave_lds_sync();
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = lane; j < N; j += 64) dst[j] = (j == 0) ? st[0] : (0 - st[N - j]);
    if (lane == 0) dst[N] = acc1[0];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      dst[m * 64 + lane] = al[m];
      dst[M + m * 64 + lane] = ah[m];
      dst[N + m * 64 + lane] = acc1[m * 64 + lane];
      dst[N + M + m * 64 + lane] = acc1[M + m * 64 + lane];
    }
  }
}

// trgsw_to_DFT / polynomial_torus_to_DFT for a flat array of polynomials [src/trgsw.c:345-349,
// src/polynomial.c:368-375]: one wavefront per polynomial, output in slot order [m][lane].
__global__ __launch_bounds__(64) void torus_to_dft_kernel_1024(const uint64_t *__restrict__ in, d2 *__restrict__ out,
                                                              const d2 *__restrict__ tw) {
  constexpr int N = 1024, M = 512;
  __shared__ __attribute__((aligned(16))) d2 xch[Fft1024::XCH_SLOTS];
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
  __shared__ __attribute__((aligned(16))) d2 xch[Fft1024::XCH_SLOTS];
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
