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
                                                                 uint64_t *__restrict__ out, size_t out_stride, int l) {
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
      const uint64_t *src = s + ((size_t)i * 2 + c) * N;
      double re[8], im[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        re[m] = torus_to_double(src[m * T + tid]);
        im[m] = torus_to_double(src[M + m * T + tid]);
      }
      fft.forward(re, im, xch, tid);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        o_re[c][m] = __builtin_fma(-im[m], k[m].y, __builtin_fma(re[m], k[m].x, o_re[c][m]));
        o_im[c][m] = __builtin_fma(im[m], k[m].x, __builtin_fma(re[m], k[m].y, o_im[c][m]));
      }
    }
  }
  const double scale = 0x1p-64 / (double)M;
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
  const int span = N / (half_base * l), j = (x / span) % l;
  tv[x] = 0;
  tv[N + x] = ~0ull << (64 - (j + 1) * Bg_bit - 1);
}

// p0[i] = tv[i], p1[i] = -tv[i + N]  [src/bootstrap.c:417-421]
__global__ void ks21_split_tv_kernel(const uint64_t *__restrict__ tv, uint64_t *__restrict__ p0, uint64_t *__restrict__ p1, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { p0[i] = tv[i]; p1[i] = 0 - tv[N + i]; }
}

}  // namespace mosfhet
