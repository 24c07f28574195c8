// unfold_kernels.h -- blind rotation with unfolding 2 [src/bootstrap.c:23-48,124-149], the per-group TRGSW assembled in the DFT domain (gfx950).
//
// The reference's unfolded rotation handles u mask words per step: it builds  xai = su_0 + sum_{j=1..2^u-1} X^(e_j) su_j  from 2^u torus-domain TRGSW
// samples of the key (su_j = TRGSW(indicator of the group's bit pattern j), e_j = mod-switched SUM of the mask words j selects), transforms it and
// replaces the accumulator by xai (.) acc: n / u external products instead of n, for one TRGSW assembly + trgsw_to_DFT per step.  On the GPU that
// assembly (rotate and add 2^u samples, then (k+1)^2 l forward transforms per ciphertext and step) is what makes the straightforward kernel
// (ext_kernels.h: pbs_unfolded_kernel) several times SLOWER than the plain rotation for batches.  Here the samples are transformed ONCE, when the key is
// made, and the assembly happens where it needs no transform: multiplication by X^e is pointwise multiplication by y^e at every root y the transform
// evaluates at, so   S = su_dft_0 + sum_j y^(e_j) (.) su_dft_j   is the same TRGSW_DFT up to rounding, and the step is ONE external product with it.
// Per step and ciphertext: the transforms of one plain CMUX step -- for two mask words --, no rotation of the accumulator, three complex multiply-adds
// per key word.  (u = 4 would read 16 key samples per step.)
// Slot (register m, lane t) holds the value at y = psi^(4 bitrev(8 t + m) + 1), psi = exp(i pi / N), and
//     4 bitrev(8 t + m) + 1 = (4 bitrev(t) + 1) + bitrev3(m) N / 4,  so  y^e = W[(4 bitrev(t) + 1) e mod 2N] * W[(bitrev3(m) e mod 8) N / 4],  W[x] = exp(i pi x / N):
// one table gather per lane and a wave-uniform eighth root of unity per register.
// Two forms with the same order of operations, hence the same bits (oracle/oracle_ext.c:orc_blind_rotate_unfolded2_dft mirrors it; against the
// reference's torus-domain assembly the results differ by FFT rounding, 2^27 - 2^30 in phase after a full rotation -- tests/test_oracle_golden.py,
// tests/test_oracle_vs_reference.py):
//   pbs_unfold2_kernel    one team per ciphertext, S formed in registers row by row                     (batches: throughput)
//   unfold2_select_kernel S of ALL groups of a ciphertext side by side, then ext_kernels.h:ubr_phase2[_wide]_kernel walks the n / 2 products  (few ciphertexts: latency)
#pragma once
#include "bootstrap_kernels.h"

namespace mosfhet {

// The factor y^(e_j) at slot (m, t): base_j times a wave-uniform eighth root of unity (none for m = 0)
template <class F>
__device__ __forceinline__ d2 unfold2_factor(const d2 &base, unsigned e, int m, const d2 *__restrict__ wtab) {
  if (m == 0) return base;
  const unsigned r3 = ((m & 1) << 2) | (m & 2) | ((m >> 2) & 1);     // bitrev3(m), a constant once unrolled
  const d2 kp = wtab[((r3 * e) & 7u) * (unsigned)(F::N / 4)];
  return d2{__builtin_fma(-base.y, kp.y, base.x * kp.x), __builtin_fma(base.y, kp.x, base.x * kp.y)};
}

// Selector polynomials, registers m = M0 .. M0 + CNT - 1 of lane t, for COMPS consecutive polynomials (the two components of one TRGSW row, or one):
// sel[c][i] = K_0 + sum_{j=1..3} y^(e_j) K_j at slot (M0 + i, t); `row` = the first polynomial inside sample 0, the other samples `sample_sz` apart;
// base[j-1] = W[(4 bitrev(t) + 1) e_j mod 2N], e[j-1] = e_j.  (The bootstrap kernel asks for a quarter of a row at a time: selector + key words of all
// eight registers would not fit beside the accumulator, the products and the digits; both components together share the factors.)
template <class F, int M0, int CNT, int COMPS>
__device__ __forceinline__ void unfold2_select(d2 (&sel)[COMPS][CNT], const d2 *__restrict__ row, size_t sample_sz, const d2 *__restrict__ wtab, const d2 (&base)[3],
                                               const unsigned (&e)[3], int t) {
  constexpr int M = F::M, T = F::THREADS;
#pragma unroll
  for (int c = 0; c < COMPS; c++)
#pragma unroll
    for (int i = 0; i < CNT; i++) sel[c][i] = row[c * M + (M0 + i) * T + t];
#pragma unroll
  for (int j = 1; j < 4; j++) {
    const d2 *__restrict__ kj = row + (size_t)j * sample_sz;
    d2 kk[COMPS][CNT];
#pragma unroll
    for (int c = 0; c < COMPS; c++)
#pragma unroll
      for (int i = 0; i < CNT; i++) kk[c][i] = kj[c * M + (M0 + i) * T + t];
#pragma unroll
    for (int i = 0; i < CNT; i++) {
      const d2 f = unfold2_factor<F>(base[j - 1], e[j - 1], M0 + i, wtab);
#pragma unroll
      for (int c = 0; c < COMPS; c++) {
        sel[c][i].x = __builtin_fma(-f.y, kk[c][i].y, __builtin_fma(f.x, kk[c][i].x, sel[c][i].x));
        sel[c][i].y = __builtin_fma(f.y, kk[c][i].x, __builtin_fma(f.x, kk[c][i].y, sel[c][i].y));
      }
    }
  }
}

// exponents and per-lane factors of one group: e_1 = a_i, e_2 = a_(i+1), e_3 = their SUM, each mod-switched afterwards (src/bootstrap.c:136-141)
template <class F>
__device__ __forceinline__ void unfold2_group(unsigned (&e)[3], d2 (&base)[3], uint64_t a1, uint64_t a2, const d2 *__restrict__ wtab, int t) {
  constexpr int LOG2N2 = F::LOGM + 2, LOGT = F::LOGM - 3;
  const unsigned g0 = 4u * (__builtin_bitreverse32((unsigned)t) >> (32 - LOGT)) + 1u;   // 4 bitrev(t) + 1
  e[0] = modswitch<LOG2N2>(a1);
  e[1] = modswitch<LOG2N2>(a2);
  e[2] = modswitch<LOG2N2>(a1 + a2);
#pragma unroll
  for (int j = 0; j < 3; j++) base[j] = wtab[(g0 * e[j]) & (unsigned)(2 * F::N - 1)];
}

// Selectors of all groups, written out: out[b][g] = S of group g for ciphertext b, [2l][2][8][T] complex -- the layout of ubr_phase1_kernel's output, so
// ubr_phase2_kernel / ubr_phase2_wide_kernel finish the bootstrap.  grid = (2l * 2, n / 2, count): one team per polynomial; `in` is the input as the
// rotation sees it (programmable_bootstrap's scaling is applied by the caller, capi_ext.inc: pbs_pre_kernel).
template <class F>
__global__ __launch_bounds__(F::THREADS) void unfold2_select_kernel(const d2 *__restrict__ su_dft, const d2 *__restrict__ wtab, const uint64_t *__restrict__ in,
                                                                  d2 *__restrict__ out, int n, int l) {
  constexpr int M = F::M, T = F::THREADS;
  const int t = threadIdx.x, qc = blockIdx.x, g = blockIdx.y;
  const uint64_t *__restrict__ ct = in + (size_t)blockIdx.z * (n + 1) + (size_t)g * 2;
  unsigned e[3];
  d2 base[3];
  unfold2_group<F>(e, base, ct[0], ct[1], wtab, t);
  const size_t sample_sz = (size_t)2 * l * 2 * M;
  d2 sel[1][8];
  unfold2_select<F, 0, 8, 1>(sel, su_dft + (size_t)g * 4 * sample_sz + (size_t)qc * M, sample_sz, wtab, base, e, t);
  d2 *dst = out + (((size_t)blockIdx.z * gridDim.y + g) * gridDim.x + qc) * M;
#pragma unroll
  for (int m = 0; m < 8; m++) dst[m * T + t] = sel[0][m];
}

// registers M0, M0 + 1 of one TRGSW row: selector words of both components, then the multiply-accumulate of the external product (src/trgsw.c:270-286)
template <class F, int M0>
__device__ __forceinline__ void unfold2_mac(double (&o_re)[2][8], double (&o_im)[2][8], const double (&re)[8], const double (&im)[8], const d2 *__restrict__ row,
                                            size_t sample_sz, const d2 *__restrict__ wtab, const d2 (&base)[3], const unsigned (&e)[3], int t) {
  d2 sel[2][2];
  unfold2_select<F, M0, 2, 2>(sel, row, sample_sz, wtab, base, e, t);
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = M0 + i;
      o_re[c][m] = __builtin_fma(-im[m], sel[c][i].y, __builtin_fma(re[m], sel[c][i].x, o_re[c][m]));
      o_im[c][m] = __builtin_fma(im[m], sel[c][i].x, __builtin_fma(re[m], sel[c][i].y, o_im[c][m]));
    }
}

struct Unfold2Params {
  PbsParams p;                       // p.bk: transformed samples [n / 2][4][2l][2][8][T] complex (slot order); p.n: LWE dimension (even)
  const d2 *__restrict__ wtab;       // W[x] = exp(i pi x / N), x < 2N
};

template <class F, int L, int BG>
__global__ __launch_bounds__(F::THREADS, 2) void pbs_unfold2_kernel(Unfold2Params u) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, LOGT = F::LOGM - 3;
  const PbsParams &p = u.p;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  __shared__ __attribute__((aligned(16))) uint64_t acc1[N];
  const int t = threadIdx.x;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft.init(p.tw, t);
  uint64_t al[8], ah[8];
  {
    // src/bootstrap.c:194-195: acc = tv * X^(2N - bbar)
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] = rot_coeff<N>(tv, m * T + t, a_lo, flip);
      ah[m] = rot_coeff<N>(tv, M + m * T + t, a_lo, flip);
      acc1[m * T + t] = rot_coeff<N>(tv + N, m * T + t, a_lo, flip);
      acc1[M + m * T + t] = rot_coeff<N>(tv + N, M + m * T + t, a_lo, flip);
    }
  }
  F::sync();
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t sample_sz = (size_t)2 * L * 2 * M;                                     // one TRGSW sample in complex slots
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);

  for (int i = 0; i < p.n; i += 2) {
    const uint64_t a1 = pbs_pre(ct[i], p, LOG2N2), a2 = pbs_pre(ct[i + 1], p, LOG2N2);
    unsigned e[3];
    d2 base[3];
    unfold2_group<F>(e, base, a1, a2, u.wtab, t);
    const d2 *__restrict__ group = p.bk + (size_t)(i / 2) * 4 * sample_sz;
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
    for (int q = 0; q < 2; q++) {
#pragma unroll 1
      for (int lv = 0; lv < L; lv++) {
        // gadget digit lv of accumulator component q (src/polynomial.c:594-613: offset once, then shift and mask), straight from the accumulator: nothing
        // is rotated here, so there is no intermediate polynomial worth packing into registers
        const int shift = 64 - (lv + 1) * Bg_bit;
        double re[8], im[8];
        if (q == 0) {
#pragma unroll
          for (int m = 0; m < 8; m++) {
            re[m] = (double)((int)((uint32_t)((al[m] + off) >> shift) & mask) - half);
            im[m] = (double)((int)((uint32_t)((ah[m] + off) >> shift) & mask) - half);
          }
        } else {
#pragma unroll
          for (int m = 0; m < 8; m++) {
            re[m] = (double)((int)((uint32_t)((acc1[m * T + t] + off) >> shift) & mask) - half);
            im[m] = (double)((int)((uint32_t)((acc1[M + m * T + t] + off) >> shift) & mask) - half);
          }
        }
        fft.forward(re, im, xch, t);
        const d2 *__restrict__ row0 = group + (size_t)(q * L + lv) * (2 * M);
        unfold2_mac<F, 0>(o_re, o_im, re, im, row0, sample_sz, u.wtab, base, e, t);
        unfold2_mac<F, 2>(o_re, o_im, re, im, row0, sample_sz, u.wtab, base, e, t);
        unfold2_mac<F, 4>(o_re, o_im, re, im, row0, sample_sz, u.wtab, base, e, t);
        unfold2_mac<F, 6>(o_re, o_im, re, im, row0, sample_sz, u.wtab, base, e, t);
      }
    }
    fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
#pragma unroll
    for (int m = 0; m < 8; m++) {      // the product REPLACES the accumulator (src/bootstrap.c:144-145)
      al[m] = round_mod_2_64(o_re[0][m], scale);
      ah[m] = round_mod_2_64(o_im[0][m], scale);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      acc1[m * T + t] = round_mod_2_64(o_re[1][m], scale);
      acc1[M + m * T + t] = round_mod_2_64(o_im[1][m], scale);
    }
    F::sync();
  }

  if (p.extract) {
    uint64_t *st = reinterpret_cast<uint64_t *>(xch);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      st[m * T + t] = al[m];
      st[M + m * T + t] = ah[m];
    }
    F::sync();
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = t; j < N; j += T) dst[j] = (j == 0) ? st[0] : (0 - st[N - j]);
    if (t == 0) dst[N] = acc1[0];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      dst[m * T + t] = al[m];
      dst[M + m * T + t] = ah[m];
      dst[N + m * T + t] = acc1[m * T + t];
      dst[N + M + m * T + t] = acc1[M + m * T + t];
    }
  }
}

}  // namespace mosfhet
