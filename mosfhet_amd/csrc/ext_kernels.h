// ext_kernels.h -- kernels of the callers either side of the bootstrap: public_mux,
// multi-value phase 1 / 2, the FFT tensor product.  Same conventions as bootstrap_kernels.h: one team (F::THREADS lanes) per
// ciphertext, folded polynomials (re = coefficient j, im = coefficient j + M), thread t owns slots m * T + t.
#pragma once
#include "bootstrap_kernels.h"

namespace mosfhet {

// digits of p1 - p0 with the UN-rounded rule of polynomial_decompose [src/polynomial.c:55-72]; dec[i][c] two's complement
__global__ void public_mux_digits_kernel(const uint64_t *__restrict__ p0, const uint64_t *__restrict__ p1, uint64_t *__restrict__ dec, int N, int l,
                                         int Bg_bit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  uint64_t off = 0;
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg_bit);
  const uint64_t d = p1[c] - p0[c] + off, mask = (1ull << Bg_bit) - 1, half = 1ull << (Bg_bit - 1);
  for (int i = 0; i < l; i++) dec[(size_t)i * N + c] = ((d >> (64 - (i + 1) * Bg_bit)) & mask) - half;
}

// public_mux [src/bootstrap.c:369-389]: out[b] = (0, p0) + sum_i DFT(sel[b][i]) (.) pdec[i]; selector rows arrive in the torus
// domain (fresh from the packing key switch) and are transformed here (trlwe_to_DFT fused); pdec = DFT of the digit polynomials.
template <class F>
__global__ __launch_bounds__(F::THREADS, 2) void public_mux_kernel(const uint64_t *__restrict__ sel, size_t sel_stride, const d2 *__restrict__ pdec,
                                                                 const uint64_t *__restrict__ p0, const d2 *__restrict__ tw,
                                                                 uint64_t *__restrict__ out, size_t out_stride, int l,
                                                                 const d2 *__restrict__ sel_dft = nullptr) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int tid = threadIdx.x;
  const uint64_t *s = sel + (size_t)blockIdx.x * sel_stride;
  uint64_t *o = out + (size_t)blockIdx.x * out_stride;
  F fft;
  fft.init(tw, tid);
  double o_re[2][8], o_im[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
  for (int i = 0; i < l; i++) {
    d2 k[8];
#pragma unroll
    for (int m = 0; m < 8; m++) k[m] = pdec[(size_t)i * M + m * T + tid];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      double re[8], im[8];
      if (sel_dft) {
        // the selector arrives as TRLWE_DFT rows (the reference's signature, include/mosfhet.h public_mux): [b][l][2][M] complex, slot order
        const d2 *src = sel_dft + ((size_t)blockIdx.x * l * 2 + (size_t)i * 2 + c) * M;
#pragma unroll
        for (int m = 0; m < 8; m++) { const d2 v = src[m * T + tid]; re[m] = v.x; im[m] = v.y; }
      } else {
        const uint64_t *src = s + ((size_t)i * 2 + c) * N;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          re[m] = torus_to_double(src[m * T + tid]);
          im[m] = torus_to_double(src[M + m * T + tid]);
        }
        fft.forward(re, im, xch, tid);
      }
#pragma unroll
      for (int m = 0; m < 8; m++) {
        o_re[c][m] = __builtin_fma(-im[m], k[m].y, __builtin_fma(re[m], k[m].x, o_re[c][m]));
        o_im[c][m] = __builtin_fma(im[m], k[m].x, __builtin_fma(re[m], k[m].y, o_im[c][m]));
      }
    }
  }
  const RoundCtx scale(0x1p-64 / (double)M);
#pragma unroll
  for (int c = 0; c < 2; c++) {
    fft.inverse(o_re[c], o_im[c], xch, tid);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      uint64_t lo = round_mod_2_64(o_re[c][m], scale), hi = round_mod_2_64(o_im[c][m], scale);
      if (c == 1) {
        lo += p0[m * T + tid];
        hi += p0[M + m * T + tid];
      }
      o[(size_t)c * N + m * T + tid] = lo;
      o[(size_t)c * N + M + m * T + tid] = hi;
    }
  }
}

// test vector of full_domain_functional_bootstrap_KS21 [src/bootstrap.c:399-404]: l interleaved LUTs of torus_base/2 slots,
// LUT j constant -2^(63 - (j+1) Bg)  (trlwe_torus_packing_many_LUT, src/trlwe.c:677-687)
__global__ void ks21_sign_lut_kernel(uint64_t *__restrict__ tv, int N, int l, int Bg_bit, int half_base) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= N) return;
  const int span = N / (half_base * l), idx = x / span, j = idx % l;   // integer division as the reference: a tail stays zero
  tv[x] = 0;
  tv[N + x] = idx < half_base * l ? ~0ull << (64 - (j + 1) * Bg_bit - 1) : 0;
}

// p0[i] = tv[i], p1[i] = -tv[i + N]  [src/bootstrap.c:417-421]
__global__ void ks21_split_tv_kernel(const uint64_t *__restrict__ tv, uint64_t *__restrict__ p0, uint64_t *__restrict__ p1, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { p0[i] = tv[i]; p1[i] = 0 - tv[N + i]; }
}

// trivial TRGSW(1) [src/trgsw.c:130-142]: row q < l carries 2^(64-(q+1)Bg) on a[0], row l + q on b[0]; grid = (2N / 256, 2l)
__global__ void trgsw_trivial_one_kernel(uint64_t *__restrict__ g, int N, int l, int Bg_bit) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (pos >= 2 * N) return;
  const int hot = (q / l) * N;
  g[(size_t)q * 2 * N + pos] = pos == hot ? 1ull << (64 - (q % l + 1) * Bg_bit) : 0;
}

// two-slot test vector {0, h} of circuit_bootstrap [src/bootstrap.c:314-315]
__global__ void circuit_bootstrap_lut2_kernel(uint64_t *__restrict__ tv, int N, uint64_t h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { tv[i] = 0; tv[N + i] = i >= N / 2 ? h : 0; }
}

// multivalue_bootstrap_phase1 rotations [src/bootstrap.c:236-241]: out[b][i] = acc[b] X^(i N / torus_base) for i < torus_base
// (i = 0: copy) and out[b][torus_base] = acc[b] X^torus_base + acc[b].  grid = (2N / 256, torus_base + 1, count)
__global__ void mv_phase1_rotate_kernel(const uint64_t *__restrict__ acc, uint64_t *__restrict__ out, int N, int torus_base) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (pos >= 2 * N) return;
  const uint64_t *c = acc + (size_t)blockIdx.z * 2 * N + (pos >= N ? N : 0);
  const int j = pos & (N - 1), a = i < torus_base ? i * N / torus_base : torus_base;
  uint64_t v = j >= a ? c[j - a] : (uint64_t)0 - c[N + j - a];   // torus_polynomial_mul_by_xai, a < N (src/polynomial.c:184-199)
  if (i == torus_base) v += c[j];
  out[((size_t)blockIdx.z * (torus_base + 1) + i) * 2 * N + pos] = v;
}

// multivalue_bootstrap_phase2 [src/bootstrap.c:245-265] for a batch sharing one cleartext LUT: coef.c[j][i] in {-1, 0, 1} is the
// multiplier of rotated_tv[i] (i <= torus_base) in the signed sum of bit j (computed on the host from the LUT), then
// trlwe_mv_extract_tlwe_scaling_addto with weight 2^j [src/trlwe.c:554-578,603-611].  One workgroup of 256 per ciphertext.
struct MvCoef { signed char c[6][65]; };
__global__ __launch_bounds__(256) void mv_phase2_kernel(const uint64_t *__restrict__ rot, uint64_t *__restrict__ out, MvCoef coef, int N, int torus_base,
                                                      int log_torus_base) {
  extern __shared__ uint64_t tmp[];  // [2][N]
  const int tid = threadIdx.x;
  const uint64_t *r = rot + (size_t)blockIdx.x * (torus_base + 1) * 2 * N;
  uint64_t acc[16], acc_b = 0;       // a words x = tid + 256 k (N <= 4096)
#pragma unroll
  for (int k = 0; k < 16; k++) acc[k] = 0;
  for (int j = 0; j < log_torus_base; j++) {
    for (int pos = tid; pos < 2 * N; pos += 256) {
      uint64_t v = 0;
      for (int i = 0; i <= torus_base; i++) {
        const int cf = coef.c[j][i];
        if (cf == 1) v += r[(size_t)i * 2 * N + pos];
        else if (cf == -1) v -= r[(size_t)i * 2 * N + pos];
      }
      tmp[pos] = v;
    }
    workgroup_sync();
    const int amount = 1 << j;
    // subtracted extractions: idx = N - 1 - (i - amount/2), i in [amount/2, amount); added: idx = i < amount/2
    for (int e = 0; e < amount; e++) {
      const bool add = e < amount / 2;
      const int idx = add ? e : N - 1 - (e - amount / 2);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int x = tid + 256 * k;
        if (x < N) {
          const uint64_t v = x <= idx ? tmp[idx - x] : (uint64_t)0 - tmp[N + idx - x];
          acc[k] += add ? v : (uint64_t)0 - v;
        }
      }
      if (tid == 0) acc_b += add ? tmp[N + idx] : (uint64_t)0 - tmp[N + idx];
    }
    workgroup_sync();
  }
  uint64_t *o = out + (size_t)blockIdx.x * (N + 1);
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (tid + 256 * k < N) o[tid + 256 * k] = acc[k];
  if (tid == 0) o[N] = acc_b;
}

// trlwe_tensor_prod_FFT before relinearisation [src/trlwe.c:727-757]: operands rescaled with torus2int (polynomial_torus_scale,
// src/polynomial.c:322-326) to half_prec1 / half_prec2 bits, res = (A1 B2 + B1 A2, B1 B2), t2 = (A1 A2, 0); products in the order
// of the reference's calls (mul, then mul_addto).  The relinearisation (trlwe_keyswitch of t2, res - that) is the FFT key-switch
// kernel in mode 2.  One team per pair.
template <class F>
__global__ __launch_bounds__(F::THREADS, 2) void trlwe_tensor_prod_kernel(const uint64_t *__restrict__ in1, size_t in1_stride,
                                                                        const uint64_t *__restrict__ in2, size_t in2_stride, const d2 *__restrict__ tw,
                                                                        uint64_t *__restrict__ res, uint64_t *__restrict__ t2, int precision) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int tid = threadIdx.x;
  const uint64_t *c1 = in1 + (size_t)blockIdx.x * in1_stride, *c2 = in2 + (size_t)blockIdx.x * in2_stride;
  uint64_t *r = res + (size_t)blockIdx.x * 2 * N, *tt = t2 + (size_t)blockIdx.x * 2 * N;
  F fft;
  fft.init(tw, tid);
  const int hp1 = 64 - (64 - precision) / 2, hp2 = 64 - (64 - precision + 1) / 2;
  const RoundCtx scale(0x1p-64 / (double)M);
  auto load = [&](const uint64_t *src, int hp, double (&re)[8], double (&im)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {   // torus2int(x, hp) = (x + 2^(63-hp)) >> (64-hp)   [src/misc.c:18-22]
      re[m] = torus_to_double((src[m * T + tid] + (1ull << (63 - hp))) >> (64 - hp));
      im[m] = torus_to_double((src[M + m * T + tid] + (1ull << (63 - hp))) >> (64 - hp));
    }
    fft.forward(re, im, xch, tid);
  };
  auto store = [&](uint64_t *dst, double (&re)[8], double (&im)[8]) {
    fft.inverse(re, im, xch, tid);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      dst[m * T + tid] = round_mod_2_64(re[m], scale);
      dst[M + m * T + tid] = round_mod_2_64(im[m], scale);
    }
  };
  double a1r[8], a1i[8], a2r[8], a2i[8], xr[8], xi[8];
  load(c1, hp1, a1r, a1i);
  load(c2, hp2, a2r, a2i);
#pragma unroll
  for (int m = 0; m < 8; m++) {      // T = A1 A2
    xr[m] = __builtin_fma(-a1i[m], a2i[m], __builtin_fma(a1r[m], a2r[m], 0.0));
    xi[m] = __builtin_fma(a1i[m], a2r[m], __builtin_fma(a1r[m], a2i[m], 0.0));
  }
  store(tt, xr, xi);
#pragma unroll
  for (int m = 0; m < 8; m++) { tt[N + m * T + tid] = 0; tt[N + M + m * T + tid] = 0; }
  double b1r[8], b1i[8], b2r[8], b2i[8];
  load(c1 + N, hp1, b1r, b1i);
  load(c2 + N, hp2, b2r, b2i);
#pragma unroll
  for (int m = 0; m < 8; m++) {      // A = A1 B2 + B1 A2
    xr[m] = __builtin_fma(-a1i[m], b2i[m], __builtin_fma(a1r[m], b2r[m], 0.0));
    xi[m] = __builtin_fma(a1i[m], b2r[m], __builtin_fma(a1r[m], b2i[m], 0.0));
    xr[m] = __builtin_fma(-b1i[m], a2i[m], __builtin_fma(b1r[m], a2r[m], xr[m]));
    xi[m] = __builtin_fma(b1i[m], a2r[m], __builtin_fma(b1r[m], a2i[m], xi[m]));
  }
  store(r, xr, xi);
#pragma unroll
  for (int m = 0; m < 8; m++) {      // B = B1 B2
    xr[m] = __builtin_fma(-b1i[m], b2i[m], __builtin_fma(b1r[m], b2r[m], 0.0));
    xi[m] = __builtin_fma(b1i[m], b2r[m], __builtin_fma(b1r[m], b2i[m], 0.0));
  }
  store(r + N, xr, xi);
}

// test vector of full_domain_functional_bootstrap_CLOT21_2 [src/bootstrap.c:503-508]: 4 interleaved LUTs of torus_base slots:
// the two halves of the user's LUT, the constant `sign`, zero  (trlwe_torus_packing_many_LUT, src/trlwe.c:677-687)
__global__ void clot21_lut_kernel(uint64_t *__restrict__ tv, const uint64_t *__restrict__ lut, uint64_t sign, int N, int torus_base) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= N) return;
  const int span = N / (4 * torus_base), idx = x / span, j = idx % 4, i = idx / 4;
  tv[x] = 0;
  tv[N + x] = i >= torus_base ? 0 : (j < 2 ? lut[j * torus_base + i] : (j == 2 ? sign : 0));
}

// Blind rotation with unfolding u > 1 [src/bootstrap.c:124-149]: per group of u mask words, the TRGSW  xai = sum_j X^(rot_j) su_j  is
// assembled row by row in the torus domain straight from HBM (rot_j = mod-switched SUM of the group's mask words selected by j),
// transformed, and multiplied with the digits of the accumulator; the product replaces the accumulator.  A straightforward
// kernel (run-time l, Bg, u; both accumulator components in registers): 3 forward transforms per TRGSW row instead of 1.
struct UnfoldParams {
  const uint64_t *__restrict__ su;   // [n 2^u / u][2l][2][N] torus domain
  const d2 *__restrict__ tw;
  const uint64_t *__restrict__ in;   // [B][n+1]
  const uint64_t *__restrict__ tv;   // [tv_count][2][N]
  uint64_t *__restrict__ out;        // [B][2][N]
  long long tv_stride;
  int n, l, Bg_bit, unfolding;
  uint64_t prec_offset;
};

template <class F>
__global__ __launch_bounds__(F::THREADS) void pbs_unfolded_kernel(UnfoldParams p) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int t = threadIdx.x;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  F fft;
  fft.init(p.tw, t);
  uint64_t acc[2][2][8];   // [component][half][m]
  {
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(ct[p.n] + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) {
        acc[c][0][m] = rot_coeff<N>(tv + c * N, m * T + t, a_lo, flip);
        acc[c][1][m] = rot_coeff<N>(tv + c * N, M + m * T + t, a_lo, flip);
      }
  }
  const int l = p.l, Bg = p.Bg_bit, u = p.unfolding, key_exp = 1 << u, final_exp = key_exp / u;
  uint64_t off = 1ull << (63 - l * Bg);
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg);
  const uint32_t mask = (1u << Bg) - 1;
  const int half = 1 << (Bg - 1);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t trgsw_sz = (size_t)2 * l * 2 * N;
#pragma unroll 1
  for (int i = 0; i < p.n; i += u) {
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
    for (int q = 0; q < 2 * l; q++) {
      const int comp = q / l, shift = 64 - (q % l + 1) * Bg;
      double dr[8], di[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const uint64_t lo = comp ? acc[1][0][m] : acc[0][0][m], hi = comp ? acc[1][1][m] : acc[0][1][m];
        dr[m] = (double)((int)((uint32_t)((lo + off) >> shift) & mask) - half);
        di[m] = (double)((int)((uint32_t)((hi + off) >> shift) & mask) - half);
      }
      fft.forward(dr, di, xch, t);
#pragma unroll 1
      for (int c = 0; c < 2; c++) {
        uint64_t xl[8], xh[8];
#pragma unroll
        for (int m = 0; m < 8; m++) { xl[m] = 0; xh[m] = 0; }
#pragma unroll 1
        for (int j = 0; j < key_exp; j++) {
          uint64_t a_i = 0;
          for (int bb = 0; bb < u; bb++)
            if ((j >> bb) & 1) a_i += ct[i + bb];
          const int rot = j ? (int)modswitch<LOG2N2>(a_i) : 0;
          const int a_lo = rot & (N - 1);
          const bool flip = (rot & N) != 0;
          const uint64_t *__restrict__ src = p.su + ((size_t)i * final_exp + j) * trgsw_sz + ((size_t)q * 2 + c) * N;
#pragma unroll
          for (int m = 0; m < 8; m++) {
            xl[m] += rot_coeff<N>(src, m * T + t, a_lo, flip);
            xh[m] += rot_coeff<N>(src, M + m * T + t, a_lo, flip);
          }
        }
        double kr[8], ki[8];
#pragma unroll
        for (int m = 0; m < 8; m++) { kr[m] = torus_to_double(xl[m]); ki[m] = torus_to_double(xh[m]); }
        fft.forward(kr, ki, xch, t);
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const double re = __builtin_fma(-di[m], ki[m], __builtin_fma(dr[m], kr[m], c ? o_re[1][m] : o_re[0][m]));
          const double im = __builtin_fma(di[m], kr[m], __builtin_fma(dr[m], ki[m], c ? o_im[1][m] : o_im[0][m]));
          if (c) { o_re[1][m] = re; o_im[1][m] = im; } else { o_re[0][m] = re; o_im[0][m] = im; }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      fft.inverse(o_re[c], o_im[c], xch, t);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        acc[c][0][m] = round_mod_2_64(o_re[c][m], scale);
        acc[c][1][m] = round_mod_2_64(o_im[c][m], scale);
      }
    }
  }
  uint64_t *o = p.out + b * (size_t)(2 * N);
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int m = 0; m < 8; m++) {
      o[c * N + m * T + t] = acc[c][0][m];
      o[c * N + M + m * T + t] = acc[c][1][m];
    }
}

// trlwe_mv_extract_tlwe / _scaling / _scaling_addto / _scaling_subto [src/trlwe.c:580-622].  mode 0: `amount` outputs per ciphertext,
// out[b][i] = +-extract(in[b], idx_i) (grid.y = amount); modes 1-3: one output, out (=, +=, -=) sum_i sign_i extract(in[b], idx_i).
// grid = (N / 256, mode 0 ? amount : 1, count)
__global__ void mv_extract_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int N, int mode, int amount) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= N) return;
  const uint64_t *c = in + (size_t)blockIdx.z * 2 * N;
  auto ext = [&](int idx) { return x <= idx ? c[idx - x] : (uint64_t)0 - c[N + idx - x]; };
  if (mode == 0) {
    const int i = blockIdx.y;
    uint64_t *o = out + ((size_t)blockIdx.z * amount + i) * (N + 1);
    const bool neg = i >= amount / 2;
    const int idx = neg ? N - 1 - (i - amount / 2) : i;
    o[x] = neg ? (uint64_t)0 - ext(idx) : ext(idx);
    if (x == 0) o[N] = neg ? (uint64_t)0 - c[N + idx] : c[N + idx];
    return;
  }
  uint64_t *o = out + (size_t)blockIdx.z * (N + 1);
  uint64_t acc = 0, acc_b = 0;
  const int first_sub = mode == 1 ? amount / 2 + 1 : amount / 2;
  for (int i = first_sub; i < amount; i++) { const int idx = N - 1 - (i - amount / 2); acc -= ext(idx); if (x == 0) acc_b -= c[N + idx]; }
  for (int i = 0; i < amount / 2; i++) { acc += ext(i); if (x == 0) acc_b += c[N + i]; }
  if (mode == 1) { acc += ext(amount / 2); if (x == 0) acc_b += c[N + amount / 2]; }
  if (mode == 1) { o[x] = acc; if (x == 0) o[N] = acc_b; }
  else if (mode == 2) { o[x] += acc; if (x == 0) o[N] += acc_b; }
  else { o[x] -= acc; if (x == 0) o[N] -= acc_b; }
}

// multivalue_bootstrap_UBR_phase1 [src/bootstrap.c:151-175]: the per-group TRGSW of the unfolded rotation, transformed and written
// out: out[b][g] = DFT( sum_j X^(rot_j) su[g 2^u + j] ), [2l][2][M] complex in slot order.  grid = (2l * 2, n / u, count): one team per
// polynomial.
template <class F>
__global__ __launch_bounds__(F::THREADS) void ubr_phase1_kernel(const uint64_t *__restrict__ su, const d2 *__restrict__ tw, const uint64_t *__restrict__ in,
                                                              d2 *__restrict__ out, int n, int l, int unfolding) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int t = threadIdx.x, qc = blockIdx.x, g = blockIdx.y;
  const uint64_t *__restrict__ ct = in + (size_t)blockIdx.z * (n + 1) + (size_t)g * unfolding;
  F fft;
  fft.init(tw, t);
  const int key_exp = 1 << unfolding;
  const size_t trgsw_sz = (size_t)2 * l * 2 * N;
  uint64_t xl[8], xh[8];
#pragma unroll
  for (int m = 0; m < 8; m++) { xl[m] = 0; xh[m] = 0; }
#pragma unroll 1
  for (int j = 0; j < key_exp; j++) {
    uint64_t a_i = 0;
    for (int bb = 0; bb < unfolding; bb++)
      if ((j >> bb) & 1) a_i += ct[bb];
    const int rot = j ? (int)modswitch<LOG2N2>(a_i) : 0;
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    const uint64_t *__restrict__ src = su + ((size_t)g * key_exp + j) * trgsw_sz + (size_t)qc * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      xl[m] += rot_coeff<N>(src, m * T + t, a_lo, flip);
      xh[m] += rot_coeff<N>(src, M + m * T + t, a_lo, flip);
    }
  }
  double kr[8], ki[8];
#pragma unroll
  for (int m = 0; m < 8; m++) { kr[m] = torus_to_double(xl[m]); ki[m] = torus_to_double(xh[m]); }
  fft.forward(kr, ki, xch, t);
  d2 *dst = out + (((size_t)blockIdx.z * gridDim.y + g) * gridDim.x + qc) * M;
#pragma unroll
  for (int m = 0; m < 8; m++) dst[m * T + t] = d2{kr[m], ki[m]};
}

// multivalue_bootstrap_UBR_phase2 [src/bootstrap.c:177-190]: acc = tv X^(2N - bbar); for every group acc <- sa[b][g] (.) acc.
// One team per (ciphertext, test vector): blockIdx.x = b * tv_count + v; writes the rotated TRLWE (the caller extracts).  tv_stride_b != 0 (with
// tv_count = 1): every ciphertext has its own test vector.
template <class F>
__global__ __launch_bounds__(F::THREADS) void ubr_phase2_kernel(const d2 *__restrict__ sa, const d2 *__restrict__ tw, const uint64_t *__restrict__ in,
                                                              const uint64_t *__restrict__ tvs, uint64_t *__restrict__ out, int n, int l, int Bg_bit,
                                                              int groups, int tv_count, uint64_t prec_offset, long long tv_stride_b) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int t = threadIdx.x;
  const size_t b = blockIdx.x / tv_count, v = blockIdx.x % tv_count;
  const uint64_t *__restrict__ ct = in + b * (size_t)(n + 1);
  const uint64_t *__restrict__ tv = tvs + v * (size_t)(2 * N) + b * (size_t)tv_stride_b;   // tv_stride_b: words between the test vectors of consecutive ciphertexts (0: shared)
  F fft;
  fft.init(tw, t);
  uint64_t acc[2][2][8];
  {
    const uint32_t bbar = modswitch<LOG2N2>(ct[n] + prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) {
        acc[c][0][m] = rot_coeff<N>(tv + c * N, m * T + t, a_lo, flip);
        acc[c][1][m] = rot_coeff<N>(tv + c * N, M + m * T + t, a_lo, flip);
      }
  }
  uint64_t off = 1ull << (63 - l * Bg_bit);
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg_bit);
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t key_sz = (size_t)2 * l * 2 * M;
#pragma unroll 1
  for (int g = 0; g < groups; g++) {
    const d2 *__restrict__ key = sa + (b * groups + g) * key_sz;
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
    for (int q = 0; q < 2 * l; q++) {
      const int comp = q / l, shift = 64 - (q % l + 1) * Bg_bit;
      double dr[8], di[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const uint64_t lo = comp ? acc[1][0][m] : acc[0][0][m], hi = comp ? acc[1][1][m] : acc[0][1][m];
        dr[m] = (double)((int)((uint32_t)((lo + off) >> shift) & mask) - half);
        di[m] = (double)((int)((uint32_t)((hi + off) >> shift) & mask) - half);
      }
      fft.forward(dr, di, xch, t);
      const d2 *__restrict__ row = key + (size_t)q * 2 * M;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const d2 k0 = row[m * T + t], k1 = row[M + m * T + t];
        o_re[0][m] = __builtin_fma(-di[m], k0.y, __builtin_fma(dr[m], k0.x, o_re[0][m]));
        o_im[0][m] = __builtin_fma(di[m], k0.x, __builtin_fma(dr[m], k0.y, o_im[0][m]));
        o_re[1][m] = __builtin_fma(-di[m], k1.y, __builtin_fma(dr[m], k1.x, o_re[1][m]));
        o_im[1][m] = __builtin_fma(di[m], k1.x, __builtin_fma(dr[m], k1.y, o_im[1][m]));
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      fft.inverse(o_re[c], o_im[c], xch, t);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        acc[c][0][m] = round_mod_2_64(o_re[c][m], scale);
        acc[c][1][m] = round_mod_2_64(o_im[c][m], scale);
      }
    }
  }
  uint64_t *o = out + (size_t)blockIdx.x * (2 * N);
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int m = 0; m < 8; m++) {
      o[c * N + m * T + t] = acc[c][0][m];
      o[c * N + M + m * T + t] = acc[c][1][m];
    }
}

// The same on two transform teams per workgroup (N = 2048; cf. bootstrap_kernels.h: pbs_wide_team_kernel): team w transforms the digits of row
// 2 ph + w, both teams multiply-accumulate the phase's two rows -- team c the output component c, rows in order: the fma chain of ubr_phase2_kernel,
// bit-identical results -- and after the last phase run one inverse transform each.  l forward + 1 inverse transform on the critical path of every
// group instead of 2l + 2: what a single unfolded bootstrap waits for.  Dynamic LDS: 2 exchange buffers + the accumulator.
template <class F>
__global__ __launch_bounds__(2 * F::THREADS) void ubr_phase2_wide_kernel(const d2 *__restrict__ sa, const d2 *__restrict__ tw, const uint64_t *__restrict__ in,
                                                                        const uint64_t *__restrict__ tvs, uint64_t *__restrict__ out, int n, int l, int Bg_bit,
                                                                        int groups, int tv_count, uint64_t prec_offset, long long tv_stride_b) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, WG = 2 * T;
  extern __shared__ __attribute__((aligned(16))) unsigned char ubr_lds[];
  d2 *xch_all = reinterpret_cast<d2 *>(ubr_lds);                                                  // [2][F::XCH_SLOTS]
  uint64_t *acc = reinterpret_cast<uint64_t *>(ubr_lds + sizeof(d2) * (size_t)2 * F::XCH_SLOTS);   // [2][N]
  const int tid = threadIdx.x, team = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  d2 *xch = xch_all + (size_t)team * F::XCH_SLOTS;
  const size_t b = blockIdx.x / tv_count, v = blockIdx.x % tv_count;
  const uint64_t *__restrict__ ct = in + b * (size_t)(n + 1);
  const uint64_t *__restrict__ tv = tvs + v * (size_t)(2 * N) + b * (size_t)tv_stride_b;   // tv_stride_b: words between the test vectors of consecutive ciphertexts (0: shared)
  F fft;
  fft.init(tw, t);
  {
    const uint32_t bbar = modswitch<LOG2N2>(ct[n] + prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += WG) acc[x] = rot_coeff<N>(tv + (x / N) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - l * Bg_bit);
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg_bit);
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t key_sz = (size_t)2 * l * 2 * M;
#pragma unroll 1
  for (int g = 0; g < groups; g++) {
    const d2 *__restrict__ key = sa + (b * groups + g) * key_sz + (size_t)team * M;   // this team's output component of every row
    double o_re[8], o_im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
#pragma unroll 1
    for (int ph = 0; ph < l; ph++) {
      d2 kk[2][8];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int m = 0; m < 8; m++) kk[r][m] = key[(size_t)(2 * ph + r) * (2 * M) + m * T + t];
      const int row = 2 * ph + team, q = row / l, shift = 64 - (row % l + 1) * Bg_bit;
      const uint64_t *accq = acc + (size_t)q * N;
      double re[8], im[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        re[m] = (double)((int)((uint32_t)((accq[m * T + t] + off) >> shift) & mask) - half);
        im[m] = (double)((int)((uint32_t)((accq[M + m * T + t] + off) >> shift) & mask) - half);
      }
      fft.forward(re, im, xch, t);
#pragma unroll
      for (int m = 0; m < 8; m++) xch[m * T + t] = d2{re[m], im[m]};
      workgroup_sync();
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const d2 *__restrict__ dr = xch_all + (size_t)r * F::XCH_SLOTS;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const d2 d = dr[m * T + t], k = kk[r][m];
          o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
          o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
        }
      }
      workgroup_sync();
    }
    fft.inverse(o_re, o_im, xch, t);
    uint64_t *accw = acc + (size_t)team * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {      // the product REPLACES the accumulator
      accw[m * T + t] = round_mod_2_64(o_re[m], scale);
      accw[M + m * T + t] = round_mod_2_64(o_im[m], scale);
    }
    workgroup_sync();
  }
  uint64_t *o = out + (size_t)blockIdx.x * (2 * N);
  for (int x = tid; x < 2 * N; x += WG) o[x] = acc[x];
}

}  // namespace mosfhet
