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
  const d2 *__restrict__ bk;         // [n][(k+1)l][k+1][8][T] complex, slot order (T = 64 or 128 threads)
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
  int count = 0;                     // host-side launch hint only (capi.hip: launch_pbs), not read by the kernels
  int rows = 1;                      // > 1: TRGSW accumulator (blind_rotate_trgsw, src/bootstrap.c:267-282): groups of `rows` consecutive
                                     // blocks share input ciphertext b / rows and start from test vector b % rows of one shared set
  unsigned int *pace = nullptr;      // != nullptr: the teams of the launch re-align every `pace_every` CMUX steps (pace_teams(): a bounded wait on a device-wide
  int pace_every = 0, pace_limit = 4000;   // counter, {arrivals, give-up flag}, zeroed by the launcher) so that they keep walking the bootstrap key TOGETHER when one
                                     // step's rows are a sizeable part of an L2 (N >= 2048); purely a matter of timing -- results do not depend on it
};

// Re-alignment of the teams of one launch (all of them resident: the launcher sends one residency round per launch).  What has to stay together is
// the set of teams that share an L2, so the rendezvous is per XCD: every team adds itself to the counter of the XCD it runs on (HW_REG_XCC_ID; one
// 128-byte line per XCD, so that the eight counters do not queue behind each other at the memory side) and waits until all the teams dealt to that XCD
// have arrived at rendezvous number `round` -- but never longer than `limit` timer ticks: a launch whose teams are NOT all resident (another stream's
// kernel holds some CUs), or one that is not dealt round-robin over the XCDs (`mine` assumes blocks b and b + 8 share one), must not hang on it.  The
// first team that times out raises the give-up flag and nobody waits again in this launch -- and leaves a credit in `pace_skip_credit` (one word per
// device): the launcher's pace_prepare_kernel starts that many following paced launches with the flag already raised, so a caller that keeps the chip shared
// (several streams or processes with full-chip launches) pays the bounded wait once in PACE_SKIP + 1 launches instead of in every one.  Memory order:
// relaxed device-scope atomics, nothing is communicated but time.  Layout of `pace`: 8 counters at 32-word spacing, then the flag at word 256.
__device__ unsigned int pace_skip_credit = 0;
__device__ unsigned int pace_skip_after_giveup = 16;   // (set by the launcher from MOSFHET_HIP_PACE_SKIP when the ring is made)

// In front of every paced launch, on its stream: counters to zero; the flag raised while a credit is left.
__global__ void pace_prepare_kernel(unsigned int *slot) {
  const unsigned int t = threadIdx.x;
  if (t < 256) slot[t] = 0u;
  if (t == 256) {
    const unsigned int credit = __hip_atomic_load(&pace_skip_credit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (credit > 0u) __hip_atomic_store(&pace_skip_credit, credit - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (racing launchers may count one credit twice: timing only)
    slot[256] = credit > 0u ? 1u : 0u;
  }
  if (t > 256) slot[t] = 0u;
}

__device__ __forceinline__ void pace_teams(unsigned int *pace, unsigned int round, int t, int limit) {
  workgroup_sync();
  if (t == 0 && __hip_atomic_load(pace + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    unsigned int xcd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcd));
    unsigned int *cnt = pace + 32 * (xcd & 7u);
    const unsigned int mine = (gridDim.x - (blockIdx.x & 7u) + 7u) / 8u;   // teams whose index is congruent to this one's mod 8
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int want = round * mine;
    const long long t0 = wall_clock64();                      // 100 MHz constant-rate timer
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      if (wall_clock64() - t0 > limit) {
        __hip_atomic_store(pace + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pace_skip_credit, pace_skip_after_giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  workgroup_sync();
}

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

// Gadget decomposition state of a lane's coefficient pair (j, j + M): the top L*Bg bits of (d + offset), from
// which digit lv is a bit-field (src/polynomial.c:74-89: ((d + off) >> (64 - (lv+1) Bg)) & (2^Bg - 1)) - 2^(Bg-1)).
//   mode PACKED (Bg compile-time, L*Bg <= 32): one 32-bit word per coefficient holding exactly those bits;
//   mode SPLIT  (Bg compile-time, L*Bg  > 32, at most 16 bits of digits reach below bit 32): the high dword per
//               coefficient plus one shared 32-bit word with the low-level digits of both coefficients;
//   mode WIDE   (run-time Bg or anything else): the full 64-bit words.
template <int X>
struct kCeilLog2 { static constexpr int value = X <= 1 ? 0 : 1 + kCeilLog2<(X + 1) / 2>::value; };
template <>
struct kCeilLog2<1> { static constexpr int value = 0; };

template <int L, int BG>
struct Digits {
  static constexpr int lo_levels() {
    int n = 0;
    for (int lv = 0; lv < L; lv++) n += (BG > 0 && 64 - (lv + 1) * BG < 32) ? 1 : 0;
    return n;
  }
  static constexpr bool kPacked = (BG > 0) && (L * BG <= 32);
  static constexpr bool kSplit = (BG > 0) && !kPacked && (lo_levels() * BG <= 16);
  static constexpr int kLo = lo_levels(), kHi = L - lo_levels();
  using word_t = typename std::conditional<kPacked || kSplit, uint32_t, uint64_t>::type;

  // u - 2^(BG-1) for a BG-bit field u is the field with its top bit flipped, read as a signed number: the packed words keep
  // the fields top-bit-flipped (one XOR per word) and digit() is a single signed bit-field extract.
  static constexpr uint32_t packed_flip() {
    uint32_t m = 0;
    for (int lv = 0; lv < L; lv++) m |= 1u << ((L - 1 - lv) * BG + BG - 1);
    return m;
  }
  static constexpr uint32_t split_hi_flip() {
    uint32_t m = 0;
    for (int lv = 0; lv < kHi; lv++) m |= 1u << (32 - (lv + 1) * BG + BG - 1);
    return m;
  }
  static constexpr uint32_t split_ext_flip() {
    uint32_t m = 0;
    for (int lv = kHi; lv < L; lv++) m |= (1u << ((L - 1 - lv) * BG + BG - 1)) | (1u << (16 + (L - 1 - lv) * BG + BG - 1));
    return m;
  }
  static __device__ __forceinline__ void pack(word_t &w_lo, word_t &w_hi, uint32_t &ext, uint64_t dd_lo, uint64_t dd_hi) {
    if constexpr (kPacked) {
      w_lo = (uint32_t)(dd_lo >> (64 - L * BG)) ^ packed_flip();
      w_hi = (uint32_t)(dd_hi >> (64 - L * BG)) ^ packed_flip();
      ext = 0;
    } else if constexpr (kSplit) {
      w_lo = (uint32_t)(dd_lo >> 32) ^ split_hi_flip();
      w_hi = (uint32_t)(dd_hi >> 32) ^ split_hi_flip();
      // bits [64 - L*BG, 64 - kHi*BG) of each word: the kLo low-level digits, kLo*BG <= 16 bits
      constexpr uint32_t m = (1u << (kLo * BG)) - 1;
      ext = (((uint32_t)(dd_lo >> (64 - L * BG)) & m) | (((uint32_t)(dd_hi >> (64 - L * BG)) & m) << 16)) ^ split_ext_flip();
    } else {
      w_lo = dd_lo;
      w_hi = dd_hi;
      ext = 0;
    }
  }
  // signed digit `lv` of the coefficient (half = 0: j, 1: j + M) as a double (exact)
  static __device__ __forceinline__ double digit(word_t w, uint32_t ext, int half, int lv, int Bg_bit) {
    if constexpr (kPacked) {
      return (double)(int)__builtin_amdgcn_sbfe(w, (unsigned)((L - 1 - lv) * BG), (unsigned)BG);   // the builtin is typed unsigned
    } else if constexpr (kSplit) {
      if (lv < kHi) return (double)(int)__builtin_amdgcn_sbfe(w, (unsigned)(32 - (lv + 1) * BG), (unsigned)BG);
      return (double)(int)__builtin_amdgcn_sbfe(ext, (unsigned)(16 * half + (L - 1 - lv) * BG), (unsigned)BG);
    } else {
      const int bg = BG > 0 ? BG : Bg_bit;
      const uint32_t u = (uint32_t)(w >> (64 - (lv + 1) * bg)) & ((1u << bg) - 1);
      return (double)((int)u - (1 << (bg - 1)));
    }
  }
};

// Digit words of one accumulator component for this thread's 8 coefficient pairs.  `home` != nullptr: the component
// lives in LDS (al / ah unused); otherwise it lives in registers (al, ah) and is staged through the transpose
// buffer for the rotation.  F = Fft1024 (one wavefront) or Fft2048 (two wavefronts, workgroup barriers).
// x + off for the run-time-gadget builds of the fused bootstrap kernel (cmux_digits; the external-product and Galois kernels gain nothing from it): the (wave-uniform) offset word is read from its SGPR pair by ONE 64-bit add.  Written as `x + off` the compiler splits it into
// add / add-with-carry, whose second half reads VCC and therefore needs the offset's high word in a VGPR; at 256 registers that copy tips the allocator into 78 - 130
// spills per lane (N = 2048) with 64 scratch operations per CMUX step -- with this form 14 - 96 and 10 - 24 (experiments/README.md round 5).  Compile-time gadgets
// keep the plain expression (their offset is an immediate).
template <int BG>
__device__ __forceinline__ uint64_t add_offset(uint64_t x, uint64_t off) {
  if constexpr (BG == 0) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(off));
    return r;
  } else {
    return x + off;
  }
}
template <class F, int L, int BG>
__device__ __forceinline__ void cmux_digits(typename Digits<L, BG>::word_t (&w_lo)[8], typename Digits<L, BG>::word_t (&w_hi)[8],
                                            uint32_t (&ext)[8], const uint64_t (&al)[8], const uint64_t (&ah)[8],
                                            const uint64_t *home, d2 *xch, int a_lo, bool flip, uint64_t off, int t) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  using D = Digits<L, BG>;
  // keep the rotated LDS addresses from being hoisted out of the component loop (they would be spilled there)
  asm volatile("" : "+s"(a_lo));
  if (home) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * T + t;
      D::pack(w_lo[m], w_hi[m], ext[m], add_offset<BG>(rot_coeff<N>(home, j, a_lo, flip) - home[j], off),
              add_offset<BG>(rot_coeff<N>(home, j + M, a_lo, flip) - home[j + M], off));
    }
  } else {
    // stage the component and read it back rotated by abar
    uint64_t *st = reinterpret_cast<uint64_t *>(xch);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      st[m * T + t] = al[m];
      st[M + m * T + t] = ah[m];
    }
    F::sync();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * T + t;
      D::pack(w_lo[m], w_hi[m], ext[m], add_offset<BG>(rot_coeff<N>(st, j, a_lo, flip) - al[m], off),
              add_offset<BG>(rot_coeff<N>(st, j + M, a_lo, flip) - ah[m], off));
    }
    F::sync();
  }
}


// The L rows of component p: digits -> forward transform -> MAC against key rows p*L .. p*L+L-1.
// The key row's component 0 is loaded under the last transform pass, component 1 under the MAC of component 0.
// KEY_LDS: the key rows are in LDS (external_product_ldskey_kernel): no early fetch into registers, each component's slots are read where they are used
template <class F, int L, int BG, bool KEY_LDS = false>
__device__ __forceinline__ void cmux_rows(const typename Digits<L, BG>::word_t (&w_lo)[8], const typename Digits<L, BG>::word_t (&w_hi)[8],
                                          const uint32_t (&ext)[8], int p, double (&o_re)[2][8], double (&o_im)[2][8], d2 *xch,
                                          const F &fft, const d2 *__restrict__ bkrow, int Bg_bit, int t) {
  constexpr int M = F::M, T = F::THREADS;
  using D = Digits<L, BG>;
  // two levels are unrolled (the next row's digit conversion and first pass overlap the MAC tail; no scratch at l = 2), more stay rolled
  constexpr int kUnroll = (L <= 2 && !KEY_LDS) ? L : 1;   // KEY_LDS: rolled -- unrolled, the LDS reads of both rows are hoisted and spill
#pragma unroll kUnroll
  for (int lv = 0; lv < L; lv++) {
    const d2 *__restrict__ row = bkrow + (size_t)(p * L + lv) * (2 * M);
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      re[m] = D::digit(w_lo[m], ext[m], 0, lv, Bg_bit);
      im[m] = D::digit(w_hi[m], ext[m], 1, lv, Bg_bit);
    }
    fft.forward_head(re, im, xch, t);
    if constexpr (KEY_LDS) {
      fft.forward_tail(re, im);
#pragma unroll
      for (int c = 0; c < 2; c++) {
        d2 kk[8];
#pragma unroll
        for (int m = 0; m < 8; m++) kk[m] = row[c * M + m * T + t];
#pragma unroll
        for (int m = 0; m < 8; m++) {
          o_re[c][m] = __builtin_fma(-im[m], kk[m].y, __builtin_fma(re[m], kk[m].x, o_re[c][m]));
          o_im[c][m] = __builtin_fma(im[m], kk[m].x, __builtin_fma(re[m], kk[m].y, o_im[c][m]));
        }
      }
      continue;
    }
    d2 k0[8], k1[8];
#pragma unroll
    for (int m = 0; m < 8; m++) k0[m] = row[m * T + t];
    fft.forward_tail(re, im);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      k1[m] = row[M + m * T + t];
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

// The same rows TWO AT A TIME (L even; transforms with forward2_head: N = 2048): the forward transforms of rows (lv, lv + 1) are software-pipelined
// through the one exchange buffer (Fft2048T::forward2_head) and the four key half-rows of the pair move through two 32-register buffers, each requested
// behind the register pass or the products next to it (the pass's temporaries and a landing buffer never live together):
//     kA <- row x, component 0      when y is parked in LDS (its registers are free): under x's last pass
//     kB <- row x, component 1      behind x's last pass: under x's products of component 0
//     kA <- row y, component 0      behind those products: under x's products of component 1, y's read-back and last pass
//     kB <- row y, component 1      behind y's last pass: under y's products of component 0
// Per output component the products are accumulated in row order with the fma chain of cmux_rows: bit-identical.
// Where it pays (same-box A/B, experiments/README.md round 4): the lvl2 external product with the twiddles of passes B and C in LDS (Fft2048L) and the
// pipelined unit loop, -11 %; the fused bootstrap kernel gains nothing (its second wavefront per SIMD already fills the exchanges) and keeps cmux_rows.
template <class F, int L, int BG>
__device__ __forceinline__ void cmux_rows2(const typename Digits<L, BG>::word_t (&w_lo)[8], const typename Digits<L, BG>::word_t (&w_hi)[8],
                                           const uint32_t (&ext)[8], int p, double (&o_re)[2][8], double (&o_im)[2][8], d2 *xch,
                                           const F &fft, const d2 *__restrict__ bkrow, int Bg_bit, int t) {
  static_assert(L % 2 == 0 && F::kForward2, "rows are taken two at a time");
  constexpr int M = F::M, T = F::THREADS;
  using D = Digits<L, BG>;
#pragma unroll 1
  for (int lv = 0; lv < L; lv += 2) {
    const d2 *__restrict__ row_x = bkrow + (size_t)(p * L + lv) * (2 * M), *__restrict__ row_y = row_x + 2 * M;
    double xr[8], xi[8], yr[8], yi[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      xr[m] = D::digit(w_lo[m], ext[m], 0, lv, Bg_bit);
      xi[m] = D::digit(w_hi[m], ext[m], 1, lv, Bg_bit);
      yr[m] = D::digit(w_lo[m], ext[m], 0, lv + 1, Bg_bit);
      yi[m] = D::digit(w_hi[m], ext[m], 1, lv + 1, Bg_bit);
    }
    fft.forward2_head(xr, xi, yr, yi, xch, t);
    d2 kA[8], kB[8];
#pragma unroll
    for (int m = 0; m < 8; m++) kA[m] = row_x[m * T + t];
    fft.pass_d_fwd(xr, xi);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < 8; m++) kB[m] = row_x[M + m * T + t];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      o_re[0][m] = __builtin_fma(-xi[m], kA[m].y, __builtin_fma(xr[m], kA[m].x, o_re[0][m]));
      o_im[0][m] = __builtin_fma(xi[m], kA[m].x, __builtin_fma(xr[m], kA[m].y, o_im[0][m]));
    }
    asm volatile("" ::: "memory");   // the next request reuses the registers the products above have consumed
#pragma unroll
    for (int m = 0; m < 8; m++) kA[m] = row_y[m * T + t];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      o_re[1][m] = __builtin_fma(-xi[m], kB[m].y, __builtin_fma(xr[m], kB[m].x, o_re[1][m]));
      o_im[1][m] = __builtin_fma(xi[m], kB[m].x, __builtin_fma(xr[m], kB[m].y, o_im[1][m]));
    }
    asm volatile("" ::: "memory");
    fft.forward2_fetch(yr, yi, xch, t);
    fft.pass_d_fwd(yr, yi);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < 8; m++) kB[m] = row_y[M + m * T + t];
    F::forward2_done();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      o_re[0][m] = __builtin_fma(-yi[m], kA[m].y, __builtin_fma(yr[m], kA[m].x, o_re[0][m]));
      o_im[0][m] = __builtin_fma(yi[m], kA[m].x, __builtin_fma(yr[m], kA[m].y, o_im[0][m]));
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      o_re[1][m] = __builtin_fma(-yi[m], kB[m].y, __builtin_fma(yr[m], kB[m].x, o_re[1][m]));
      o_im[1][m] = __builtin_fma(yi[m], kB[m].x, __builtin_fma(yr[m], kB[m].y, o_im[1][m]));
    }
  }
}

// The fused bootstrap kernel.  Accumulator placement: component a (acc[0]) in VGPRs in the transform's input
// layout (thread t owns coefficients m*T+t and m*T+t+M: 32 VGPRs), component b (acc[1]) resident in LDS; with
// the transpose buffer that is 17 KiB of LDS per wavefront and <= 256 VGPRs, i.e. two wavefronts per SIMD /
// eight per CU.  (Both components in LDS: 25 KiB -> 6 per CU, measured 22 % slower; both in registers spills.)
// F = Fft1024: one wavefront per ciphertext; F = Fft2048: two wavefronts (128 threads) per ciphertext.
template <class F, int L, int BG>
__global__ __launch_bounds__(F::THREADS, 2) void pbs_kernel(PbsParams p) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2;
  // Every output of the external product is a sum of 2L * N products digit * key coefficient with |digit| <= 2^(BG-1) and |key| <= 2^63
  // (the key is (double)(int64_t) of torus words): |sum| <= 2^(ceil log2(2L) + log2 N + BG - 1 + 63).  Below 2^83 the rounding needs no
  // reduction mod 1 in front (add_rounded); that holds for SET_1's 2 x 2^8 gadget at N = 1024 (2^82) and is decided at compile time.
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  __shared__ __attribute__((aligned(16))) uint64_t acc1[N];
  const int t = threadIdx.x;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + (p.rows > 1 ? b / (size_t)p.rows : b) * (size_t)(p.n + 1);
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;

  F fft;
  fft_setup(fft, p.tw, t);

  uint64_t al[8], ah[8];
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] = src[m * T + t];
      ah[m] = src[M + m * T + t];
      acc1[m * T + t] = src[N + m * T + t];
      acc1[M + m * T + t] = src[N + M + m * T + t];
    }
  } else {
    // src/bootstrap.c:194-195: acc = tv * X^(2N - bbar), gathered straight from global memory
    const uint64_t *__restrict__ tv = p.rows > 1 ? p.tv + (b % (size_t)p.rows) * (size_t)(2 * N) : p.tv + b * (size_t)p.tv_stride;
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
  const size_t row_sz = (size_t)2 * L * 2 * M;

  for (int i = 0; i < p.n; i++) {
    if (T > 64 && p.pace && i > 0 && i % p.pace_every == 0) pace_teams(p.pace, (unsigned)(i / p.pace_every), t, p.pace_limit);   // (before the skip: every team counts every step)
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
    {
      constexpr int kUnrollQ = L == 1 ? 2 : 1;
#pragma unroll kUnrollQ
      for (int q = 0; q < 2; q++) {
        typename Digits<L, BG>::word_t w_lo[8], w_hi[8];
        uint32_t ext[8];
        cmux_digits<F, L, BG>(w_lo, w_hi, ext, al, ah, q ? acc1 : nullptr, xch, a_lo, flip, off, t);
        // rows two at a time where the transform keeps its pass twiddles in LDS (tools/ab/pbs_ab.hip -DAB_LTW instantiates that) -- at two wavefronts per SIMD
        // the pairs gain nothing here (experiments/README.md round 4): production instantiates pbs_kernel on the register-twiddle transforms
        if constexpr (F::kLtw && F::kForward2 && L % 2 == 0) cmux_rows2<F, L, BG>(w_lo, w_hi, ext, q, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
        else cmux_rows<F, L, BG>(w_lo, w_hi, ext, q, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
      }
    }
    fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] = add_rounded<kReduce>(al[m], o_re[0][m], scale);
      ah[m] = add_rounded<kReduce>(ah[m], o_im[0][m], scale);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      acc1[m * T + t] = add_rounded<kReduce>(acc1[m * T + t], o_re[1][m], scale);
      acc1[M + m * T + t] = add_rounded<kReduce>(acc1[M + m * T + t], o_im[1][m], scale);
    }
    F::sync();
  }

  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0: a[0] = acc_a[0], a[j] = -acc_a[N - j]; b = acc_b[0]
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

// ------------------------------------------------------------------------------------------------------------
// Latency-oriented variant for SMALL batches (fewer ciphertexts than the chip has CUs x 8): one workgroup of 2L wavefronts per
// ciphertext, wavefront w owns TRGSW row w of every CMUX (its digits, its forward transform), then wavefronts 0 and 1 run the
// multiply-accumulate over all rows -- in row order, the same fma chain as pbs_kernel and the oracle, so results are bit-identical --
// and the inverse transform of one output component each.  Per CMUX the critical path is 1 forward + 1 inverse transform instead of
// 2L + 2: a single bootstrap takes ~1/3 of the time, at ~1/3 of the throughput kernel's per-CU rate (capi.hip picks by batch size).
// Accumulator (both components) in LDS; per-wavefront transpose buffers; the transformed digits are handed over through LDS.
// N = 1024 only (Fft1024: wave-level transposes, workgroup barriers only at the two hand-overs).
// ------------------------------------------------------------------------------------------------------------
template <int L, int BG>
__global__ __launch_bounds__(64 * 2 * L) void pbs_team_kernel(PbsParams p) {
  using F = Fft1024;
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, R = 2 * L, TEAM = T * R;
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);   // see pbs_kernel
  __shared__ __attribute__((aligned(16))) d2 xch[R][F::XCH_SLOTS];     // transposes; after the forward transform: D_w (slots m T + t)
  __shared__ __attribute__((aligned(16))) d2 xinv[2][F::XCH_SLOTS];    // transposes of the two inverse transforms
  __shared__ __attribute__((aligned(16))) uint64_t acc[2][N];
  const int tid = threadIdx.x, w = tid >> 6, t = tid & 63;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft.init(p.tw, t);
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += TEAM) acc[x >> 10][x & (N - 1)] = src[x];
  } else {
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += TEAM) acc[x >> 10][x & (N - 1)] = rot_coeff<N>(tv + (x >> 10) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * L * 2 * M;
  const int q = w / L, shift = 64 - (w % L + 1) * Bg_bit;   // this wavefront's row: component q, level w % L
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  for (int i = 0; i < p.n; i++) {
    const int abar = (int)modswitch<LOG2N2>(pbs_pre(ct[i], p, LOG2N2));
    if (!abar) continue;   // src/bootstrap.c:114 (uniform over the workgroup)
    const d2 *__restrict__ bkrow = p.bk + (size_t)i * row_sz;
    const int a_lo = abar & (N - 1);
    const bool flip = (abar & N) != 0;
    // wavefronts 0 and 1 fetch their component of all R key rows now, under the digit extraction and the forward transform
    d2 kk[R][8];
    if (w < 2) {
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int m = 0; m < 8; m++) kk[r][m] = bkrow[(size_t)r * (2 * M) + (size_t)w * M + m * T + t];
    }
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int j = m * T + t;
      const uint64_t d_lo = rot_coeff<N>(acc[q], j, a_lo, flip) - acc[q][j] + off;
      const uint64_t d_hi = rot_coeff<N>(acc[q], j + M, a_lo, flip) - acc[q][j + M] + off;
      re[m] = (double)((int)((uint32_t)(d_lo >> shift) & mask) - half);
      im[m] = (double)((int)((uint32_t)(d_hi >> shift) & mask) - half);
    }
    fft.forward(re, im, xch[w], t);
#pragma unroll
    for (int m = 0; m < 8; m++) xch[w][m * T + t] = d2{re[m], im[m]};
    workgroup_sync();
    if (w < 2) {   // output component c = w: fma chain over the rows in order, inverse, round, accumulate
      double o_re[8], o_im[8];
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
#pragma unroll
      for (int r = 0; r < R; r++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const d2 d = xch[r][m * T + t], k = kk[r][m];
          o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
          o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
        }
      }
      fft.inverse(o_re, o_im, xinv[w], t);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        acc[w][m * T + t] = add_rounded<kReduce>(acc[w][m * T + t], o_re[m], scale);
        acc[w][M + m * T + t] = add_rounded<kReduce>(acc[w][M + m * T + t], o_im[m], scale);
      }
    }
    workgroup_sync();
  }
  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = tid; j < N; j += TEAM) dst[j] = (j == 0) ? acc[0][0] : (0 - acc[0][N - j]);
    if (tid == 0) dst[N] = acc[1][0];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += TEAM) dst[x] = acc[x >> 10][x & (N - 1)];
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same idea on a ring of two wavefronts (N = 2048: the lvl2 set and the reference's applications, where ONE bootstrap takes 11.8 ms at l = 4 and
// 16 ms at l = 6 on pbs_kernel and gate-by-gate programs wait for exactly that): TEAMS transform teams of F::THREADS threads per ciphertext, team w
// owns TRGSW rows w, w + TEAMS, ... of every CMUX (digits, forward transform, the transformed digits left in its own exchange buffer), teams 0 and 1
// run the multiply-accumulate over the rows of each phase -- in row order, the fma chain of pbs_kernel and the oracle: bit-identical results -- and
// the inverse transform of one output component each.  Critical path per CMUX: 2L / TEAMS forward + 1 inverse transform instead of 2L + 2.
// F's cross-wavefront exchanges use workgroup barriers, so every team walks the same sequence of barriers (with more than two teams, those without an
// output component or, in a ragged last phase, without a row execute the barriers of the transform and nothing else).
// TEAMS = 2 (four wavefronts, one per SIMD: every transform has an FP64 pipe to itself, and each phase's key rows -- 64 KiB per CU at ~50 GB/s -- arrive
// under the phase's own transform) measured 8.8 ms per lvl2 bootstrap against 9.3 ms with four teams, which share the pipes two to one and in lock step
// cannot hide each other's LDS exchanges; requesting the rows a whole phase ahead into a second register buffer changed nothing (9.4 ms).
// Dynamic LDS: TEAMS exchange buffers + the accumulator (68 KiB at two teams).
// ------------------------------------------------------------------------------------------------------------
template <int L> struct WideTeams { static constexpr int value = 2; };   // teams per ciphertext: four wavefronts, one per SIMD (measured against 4 teams below); rows in l phases of two

template <class F, int L, int BG>
__global__ __launch_bounds__(F::THREADS * WideTeams<L>::value) void pbs_wide_team_kernel(PbsParams p) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, R = 2 * L, TEAMS = WideTeams<L>::value, PHASES = (R + TEAMS - 1) / TEAMS, WG = T * TEAMS;
  constexpr int KPRE = TEAMS == 4 ? 3 : TEAMS;   // key rows requested ahead of each phase's forward transform (what the registers hold without spilling)
  static_assert(TEAMS >= 2, "teams 0 and 1 own the two output components");
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);   // see pbs_kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
  d2 *xch_all = reinterpret_cast<d2 *>(wide_lds);                                                     // [TEAMS][F::XCH_SLOTS]
  uint64_t *acc = reinterpret_cast<uint64_t *>(wide_lds + sizeof(d2) * (size_t)TEAMS * F::XCH_SLOTS);  // [2][N]
  const int tid = threadIdx.x, team = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  d2 *xch = xch_all + (size_t)team * F::XCH_SLOTS;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + (p.rows > 1 ? b / (size_t)p.rows : b) * (size_t)(p.n + 1);   // rows > 1: TRGSW accumulators, see PbsParams
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft.init(p.tw, t);
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) acc[x] = src[x];
  } else {
    const uint64_t *__restrict__ tv = p.rows > 1 ? p.tv + (b % (size_t)p.rows) * (size_t)(2 * N) : p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += WG) acc[x] = rot_coeff<N>(tv + (x / N) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * L * 2 * M;
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  const bool mac = team < 2;    // this team owns output component `team`
  for (int i = 0; i < p.n; i++) {
    const int abar = (int)modswitch<LOG2N2>(pbs_pre(ct[i], p, LOG2N2));
    if (!abar) continue;   // src/bootstrap.c:114 (uniform over the workgroup)
    const d2 *__restrict__ bkrow = p.bk + (size_t)i * row_sz;
    const int a_lo = abar & (N - 1);
    const bool flip = (abar & N) != 0;
    double o_re[8], o_im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
#pragma unroll
    for (int ph = 0; ph < PHASES; ph++) {
      // teams 0 and 1 request their component of the phase's first key rows now, under the digit extraction and the forward transform
      const int rows = (ph + 1) * TEAMS <= R ? TEAMS : R - ph * TEAMS;   // rows of this phase (a constant once the phases are unrolled)
      d2 kk[KPRE][8];
      if (mac) {
#pragma unroll
        for (int r = 0; r < KPRE; r++)
          if (r < rows)
#pragma unroll
            for (int m = 0; m < 8; m++) kk[r][m] = bkrow[(size_t)(ph * TEAMS + r) * (2 * M) + (size_t)team * M + m * T + t];
      }
      const int row = ph * TEAMS + team, q = row / L, shift = 64 - (row % L + 1) * Bg_bit;   // this team's row: component q, level row % L
      double re[8], im[8];
      if (row >= R) {   // ragged last phase: no row for this team -- keep the barrier sequence, leave the FP64 pipe to the others
        F::transform_barriers_only();
        workgroup_sync();
        workgroup_sync();
        continue;
      }
      {
        const uint64_t *accq = acc + (size_t)q * N;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int j = m * T + t;
          const uint64_t d_lo = rot_coeff<N>(accq, j, a_lo, flip) - accq[j] + off;
          const uint64_t d_hi = rot_coeff<N>(accq, j + M, a_lo, flip) - accq[j + M] + off;
          re[m] = (double)((int)((uint32_t)(d_lo >> shift) & mask) - half);
          im[m] = (double)((int)((uint32_t)(d_hi >> shift) & mask) - half);
        }
      }
      fft.forward(re, im, xch, t);
#pragma unroll
      for (int m = 0; m < 8; m++) xch[m * T + t] = d2{re[m], im[m]};
      workgroup_sync();
      if (mac) {   // fma chain over the phase's rows in order
#pragma unroll
        for (int r = 0; r < TEAMS; r++) {
          if (r >= rows) continue;
          const d2 *__restrict__ dr = xch_all + (size_t)r * F::XCH_SLOTS;
#pragma unroll
          for (int m = 0; m < 8; m++) {
            const d2 d = dr[m * T + t];
            d2 k;
            if (r < KPRE) k = kk[r < KPRE ? r : 0][m];
            else k = bkrow[(size_t)(ph * TEAMS + r) * (2 * M) + (size_t)team * M + m * T + t];
            o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
            o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
          }
        }
      }
      workgroup_sync();   // the transformed digits are consumed: the buffers are free for the next phase's exchanges / the inverse
    }
    if (mac) {   // (workgroup barriers inside: the other teams walk the same barriers)
      fft.inverse(o_re, o_im, xch, t);
      uint64_t *accw = acc + (size_t)team * N;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        accw[m * T + t] = add_rounded<kReduce>(accw[m * T + t], o_re[m], scale);
        accw[M + m * T + t] = add_rounded<kReduce>(accw[M + m * T + t], o_im[m], scale);
      }
    } else {
      F::transform_barriers_only();
    }
    workgroup_sync();
  }
  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = tid; j < N; j += WG) dst[j] = (j == 0) ? acc[0] : (0 - acc[N - j]);
    if (tid == 0) dst[N] = acc[N];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) dst[x] = acc[x];
  }
}

// ------------------------------------------------------------------------------------------------------------
// pbs_wide_team_kernel with each team taking its rows TWO AT A TIME (N = 2048, even gadget length, the transform with its pass twiddles in LDS): the two
// forward transforms of a team's pair are software-pipelined through the team's one exchange buffer (Fft2048T::forward2_head: every register pass of one runs
// while the other's exchange is in flight -- at ONE wavefront per SIMD, which is what this kernel runs at, that is where the idle time was), both teams hand
// their pair over through LDS and multiply-accumulate the four rows of the double phase in row order: the same butterflies, digits and fma chain as
// pbs_wide_team_kernel, pbs_kernel and the oracle -- bit-identical.  A double phase is 4 rows: team w owns rows 4 d + w (x) and 4 d + 2 + w (y) (one accumulator
// component, two levels: the rotated accumulator words are read once for both), so that both x rows -- rows 0 and 1 of the chain -- are handed over first and are
// multiplied while the y rows land.  One workgroup per CU (137 KiB of LDS: two exchange buffers, four
// hand-over buffers, the accumulator, the twiddle table) and up to 512 registers per lane: all four key rows of a double phase are requested before its transforms.
// ------------------------------------------------------------------------------------------------------------
template <class F, int L, int BG>
__global__ __launch_bounds__(2 * F::THREADS) void pbs_wide_pair_kernel(PbsParams p) {
  static_assert(F::kForward2 && F::kLtw && L % 2 == 0, "row pairs need the pipelined forward pair and an even gadget length");
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, R = 2 * L, DPH = R / 4, WG = 2 * T;
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);   // see pbs_kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
  d2 *xch_all = reinterpret_cast<d2 *>(wide_lds);                                   // [2][F::XCH_SLOTS]
  d2 *hand = xch_all + (size_t)2 * F::XCH_SLOTS;                                    // [4][M]: row r of the double phase
  uint64_t *acc = reinterpret_cast<uint64_t *>(hand + (size_t)4 * M);               // [2][N]
  const int tid = threadIdx.x, team = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  d2 *xch = xch_all + (size_t)team * F::XCH_SLOTS;
  const size_t b = blockIdx.x;
  const uint64_t *__restrict__ ct = p.in + (p.rows > 1 ? b / (size_t)p.rows : b) * (size_t)(p.n + 1);
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft_setup(fft, p.tw, t);
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) acc[x] = src[x];
  } else {
    const uint64_t *__restrict__ tv = p.rows > 1 ? p.tv + (b % (size_t)p.rows) * (size_t)(2 * N) : p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += WG) acc[x] = rot_coeff<N>(tv + (x / N) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * L * 2 * M;
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  for (int i = 0; i < p.n; i++) {
    const int abar = (int)modswitch<LOG2N2>(pbs_pre(ct[i], p, LOG2N2));
    if (!abar) continue;   // src/bootstrap.c:114 (uniform over the workgroup)
    const d2 *__restrict__ bkrow = p.bk + (size_t)i * row_sz;
    const int a_lo = abar & (N - 1);
    const bool flip = (abar & N) != 0;
    double o_re[8], o_im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
#pragma unroll
    for (int dp = 0; dp < DPH; dp++) {
      d2 kk[4][8];   // this team's output component of the double phase's four key rows
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int m = 0; m < 8; m++) kk[r][m] = bkrow[(size_t)(4 * dp + r) * (2 * M) + (size_t)team * M + m * T + t];
      // rows rx (x) and rx + 2 (y): L is even, so both teams' x rows lie in one accumulator component and both y rows in one (the same one unless L % 4 != 0)
      const int rx = 4 * dp + team, ry = rx + 2, qx = (4 * dp) / L, qy = (4 * dp + 2) / L, sx = 64 - (rx % L + 1) * Bg_bit, sy = 64 - (ry % L + 1) * Bg_bit;
      double xr[8], xi[8], yr[8], yi[8];
      {
        const uint64_t *accx = acc + (size_t)qx * N, *accy = acc + (size_t)qy * N;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int j = m * T + t;
          const uint64_t d_lo = rot_coeff<N>(accx, j, a_lo, flip) - accx[j] + off;
          const uint64_t d_hi = rot_coeff<N>(accx, j + M, a_lo, flip) - accx[j + M] + off;
          xr[m] = (double)((int)((uint32_t)(d_lo >> sx) & mask) - half);
          xi[m] = (double)((int)((uint32_t)(d_hi >> sx) & mask) - half);
          const uint64_t e_lo = qy == qx ? d_lo : rot_coeff<N>(accy, j, a_lo, flip) - accy[j] + off;
          const uint64_t e_hi = qy == qx ? d_hi : rot_coeff<N>(accy, j + M, a_lo, flip) - accy[j + M] + off;
          yr[m] = (double)((int)((uint32_t)(e_lo >> sy) & mask) - half);
          yi[m] = (double)((int)((uint32_t)(e_hi >> sy) & mask) - half);
        }
      }
      fft.forward2_head(xr, xi, yr, yi, xch, t);
      fft.pass_d_fwd(xr, xi);
      d2 *hx = hand + (size_t)team * M, *hy = hand + (size_t)(2 + team) * M;   // hand-over buffer r holds row 4 dp + r
#pragma unroll
      for (int m = 0; m < 8; m++) hx[m * T + t] = d2{xr[m], xi[m]};
      fft.forward2_fetch(yr, yi, xch, t);
      fft.pass_d_fwd(yr, yi);
      F::forward2_done();   // (a workgroup barrier: both teams' x rows are handed over)
#pragma unroll
      for (int m = 0; m < 8; m++) hy[m * T + t] = d2{yr[m], yi[m]};
#pragma unroll
      for (int r = 0; r < 2; r++) {   // fma chain over the double phase's rows in order: the x rows (0, 1) while the y rows land
        const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const d2 d = dr[m * T + t], k = kk[r][m];
          o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
          o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
        }
      }
      workgroup_sync();
#pragma unroll
      for (int r = 2; r < 4; r++) {
        const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const d2 d = dr[m * T + t], k = kk[r][m];
          o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
          o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
        }
      }
      workgroup_sync();   // the hand-over buffers are consumed
    }
    fft.inverse(o_re, o_im, xch, t);
    uint64_t *accw = acc + (size_t)team * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      accw[m * T + t] = add_rounded<kReduce>(accw[m * T + t], o_re[m], scale);
      accw[M + m * T + t] = add_rounded<kReduce>(accw[M + m * T + t], o_im[m], scale);
    }
    workgroup_sync();
  }
  if (p.extract) {
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = tid; j < N; j += WG) dst[j] = (j == 0) ? acc[0] : (0 - acc[N - j]);
    if (tid == 0) dst[N] = acc[N];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) dst[x] = acc[x];
  }
}

// ------------------------------------------------------------------------------------------------------------
// pbs_split_kernel: ONE bootstrap on TWO workgroups (two CUs), split by accumulator component -- for batches that leave half the chip idle (N = 2048, l = 2 / 4 / 6:
// up to CUs / 2 ciphertexts, the share of configs[3] / [4] one GPU of eight gets; l = 4: one double phase, l = 6: a double phase and one row per team, l = 2: one row per team).  Workgroup h of a pair keeps accumulator component h.  The rows h l ..
// h l + l - 1 of a TRGSW sample are exactly the rows that decompose component h (src/trgsw.c:393-419), so per CMUX step workgroup h
//   * rotates and decomposes ITS component (both teams read it), runs the l forward transforms (pbs_wide_pair_kernel's double phase: team w takes rows
//     h l + w and h l + 2 + w, pipelined, handed over through LDS) and multiplies-accumulates them against the key rows' two components, team w the output
//     component w: the partial sums S_h[0], S_h[1] -- half the transforms, half the products and half the key rows (128 KiB) of a step;
//   * team 1 - h SENDS S_h[1 - h] (16 KiB) to the partner; team h RECEIVES S_(1-h)[h], adds it to S_h[h], runs the ONE inverse transform, rounds and adds
//     to its component -- all it needs for its next decomposition.
// Summation order: out[c] = (chain over rows 0 .. l-1 from zero) + (chain over rows l .. 2l-1 from zero) -- NOT the reference's one chain over all 2 l rows
// (an fma chain cannot be cut without changing roundings).  The difference is FFT-level rounding; the oracle restates this order
// (oracle_tfhe.c: orc_set_product_order(1), held to the reference within the tolerance of the plain order) and the kernel is bit-identical to THAT.
// Exchange (tools/ubench/xchg.hip: 1.5 us per step against 3.75 us for data + flag): no flag.  Receive slots hold a sentinel -- a NaN pattern that fma
// arithmetic on finite numbers never produces; the sender's lanes store their 16-byte items with device-scope stores, the receiver's lanes poll THEIR OWN
// items with device-scope loads until both halves of every item differ from the sentinel, then put the sentinel back (complete before the workgroup's next send
// can be observed: s_waitcnt + the step's barrier).  Two slot sets alternate by step, so a sender never meets a slot its partner has not reset.
// Pairing: blocks B and B + 8 (the same XCD when workgroups are dealt round-robin; nothing depends on it) claim their ciphertext through one state word:
// the first to arrive waits a bounded time for the second; if that runs out (the partner is not resident: the chip is shared) it takes the whole
// bootstrap ALONE -- both components, the same summation order, the same bits -- and the late partner leaves.  Nothing can hang and no result depends on timing.
// ------------------------------------------------------------------------------------------------------------
constexpr uint64_t kSplitSentinel = 0xFFF7A5C3DEADBEEFull;
struct SplitParams {
  d2 *xbuf;              // [count][2 receivers][2 step parities][M] receive slots, all sentinel between launches
  unsigned int *state;   // [count] 0 nobody, 1 one workgroup waiting, 2 paired, 3 taken alone; zero before the launch
  int count;             // ciphertexts (x accumulator rows)
  int limit;             // bound of the pairing wait in 10 ns ticks; <= 0: every bootstrap is taken alone by its first workgroup (test switch)
};
__global__ void split_prepare_kernel(uint64_t *xbuf_words, size_t n_words, unsigned int *state, int count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) xbuf_words[i] = kSplitSentinel;
  if (i < (size_t)count) state[i] = 0u;
}
// the lane's eight 16-byte items of a 16 KiB set (item m at [m * 128 + t]), device scope
__device__ __forceinline__ void split_store8(d2 *p, const double (&re)[8], const double (&im)[8]) {
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const d2 v = d2{re[m], im[m]};
    // (s_nop 1: a store of more than 8 bytes reads its data registers late -- the VALU instruction behind it must not overwrite them for two wait states, and the
    // compiler's hazard recognizer does not look into inline assembly; tools/ubench/xchg.hip sent half-computed real parts from lanes 12 - 15 of every 16 without it)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + m * 128), "v"(v) : "memory");
  }
}
__device__ __forceinline__ void split_load8(d2 (&v)[8], const d2 *p) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %2, %9, off sc1\n\t"
      "global_load_dwordx4 %3, %9, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %4, %10, off sc1\n\t"
      "global_load_dwordx4 %5, %10, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %6, %11, off sc1\n\t"
      "global_load_dwordx4 %7, %11, off offset:2048 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p), "v"(p + 256), "v"(p + 512), "v"(p + 768)
      : "memory");
}

template <class F, int L, int BG>
__global__ __launch_bounds__(2 * F::THREADS) void pbs_split_kernel(PbsParams p, SplitParams sp) {
  static_assert(F::kForward2 && F::kLtw && L % 2 == 0 && L >= 2 && L <= 6 && F::THREADS == 128,
                "the l rows of one accumulator component: l / 4 double phases of pbs_wide_pair_kernel (two pipelined rows per team) and, for l = 2 and 6, one row per team");
  constexpr int QUADS = L / 4, TAIL = L % 4;   // TAIL = 2: the last two levels go one row per team
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, WG = 2 * T;
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);   // see pbs_kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
  d2 *xch_all = reinterpret_cast<d2 *>(wide_lds);                                   // [2][F::XCH_SLOTS]
  d2 *hand = xch_all + (size_t)2 * F::XCH_SLOTS;                                    // [4][M]: row r of the double phase
  uint64_t *acc = reinterpret_cast<uint64_t *>(hand + (size_t)4 * M);               // [2][N] (paired: only component h is kept up to date)
  __shared__ int mode_s;
  const int tid = threadIdx.x, team = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  d2 *xch = xch_all + (size_t)team * F::XCH_SLOTS;
  const int h = (int)((blockIdx.x >> 3) & 1u);
  const size_t b = (size_t)(blockIdx.x >> 4) * 8 + (blockIdx.x & 7u);
  if (b >= (size_t)sp.count) return;
  // ---- pairing ----
  if (tid == 0) {
    unsigned int *st = sp.state + b;
    int mode;   // 0 paired, 1 alone, 2 leave
    if (sp.limit <= 0) {
      mode = h == 0 ? 1 : 2;
    } else {
      unsigned int seen = 0u;
      if (__hip_atomic_compare_exchange_strong(st, &seen, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        const long long t0 = wall_clock64();   // first here: wait for the partner, but not for ever
        mode = -1;
        while (mode < 0) {
          if (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) mode = 0;
          else if (wall_clock64() - t0 > sp.limit) {
            unsigned int one = 1u;
            mode = __hip_atomic_compare_exchange_strong(st, &one, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
          } else __builtin_amdgcn_s_sleep(8);
        }
      } else if (seen == 1u) {
        unsigned int one = 1u;
        mode = __hip_atomic_compare_exchange_strong(st, &one, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0 : 2;
      } else {
        mode = 2;
      }
    }
    mode_s = mode;
  }
  workgroup_sync();
  const int mode = mode_s;
  if (mode == 2) return;
  const bool alone = mode == 1;
  const int dp_lo = alone ? 0 : h, dp_hi = alone ? 2 : h + 1;
  d2 *recv = sp.xbuf + ((size_t)b * 2 + (size_t)h) * 2 * M, *send = sp.xbuf + ((size_t)b * 2 + (size_t)(1 - h)) * 2 * M;

  const uint64_t *__restrict__ ct = p.in + (p.rows > 1 ? b / (size_t)p.rows : b) * (size_t)(p.n + 1);
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft_setup(fft, p.tw, t);
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) acc[x] = src[x];
  } else {
    const uint64_t *__restrict__ tv = p.rows > 1 ? p.tv + (b % (size_t)p.rows) * (size_t)(2 * N) : p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(pbs_pre(ct[p.n], p, LOG2N2) + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += WG) acc[x] = rot_coeff<N>(tv + (x / N) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * L * 2 * M;
  const uint32_t mask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  int par = 0;
  for (int i = 0; i < p.n; i++) {
    const int abar = (int)modswitch<LOG2N2>(pbs_pre(ct[i], p, LOG2N2));
    if (!abar) continue;   // src/bootstrap.c:114 (uniform over both workgroups of a pair)
    const d2 *__restrict__ bkrow = p.bk + (size_t)i * row_sz;
    const int a_lo = abar & (N - 1);
    const bool flip = (abar & N) != 0;
    double s_re[8], s_im[8];   // this team's output component: sum of the components' partial sums, component 0's first
#pragma unroll 1
    for (int dp = dp_lo; dp < dp_hi; dp++) {
      const uint64_t *accx = acc + (size_t)dp * N;
      const d2 *__restrict__ rows = bkrow + (size_t)(dp * L) * (2 * M) + (size_t)team * M + t;   // rows dp L .. dp L + L - 1, this team's output component
      double o_re[8], o_im[8];
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
#pragma unroll
      for (int q = 0; q < QUADS; q++) {   // levels 4 q .. 4 q + 3: team w transforms levels 4 q + w (x) and 4 q + 2 + w (y), pipelined
        d2 kk[4][8];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int m = 0; m < 8; m++) kk[r][m] = rows[(size_t)(4 * q + r) * (2 * M) + m * T];
        const int sx = 64 - (4 * q + team + 1) * Bg_bit, sy = 64 - (4 * q + team + 3) * Bg_bit;
        double xr[8], xi[8], yr[8], yi[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int j = m * T + t;
          const uint64_t d_lo = rot_coeff<N>(accx, j, a_lo, flip) - accx[j] + off;
          const uint64_t d_hi = rot_coeff<N>(accx, j + M, a_lo, flip) - accx[j + M] + off;
          xr[m] = (double)((int)((uint32_t)(d_lo >> sx) & mask) - half);
          xi[m] = (double)((int)((uint32_t)(d_hi >> sx) & mask) - half);
          yr[m] = (double)((int)((uint32_t)(d_lo >> sy) & mask) - half);
          yi[m] = (double)((int)((uint32_t)(d_hi >> sy) & mask) - half);
        }
        fft.forward2_head(xr, xi, yr, yi, xch, t);
        fft.pass_d_fwd(xr, xi);
        d2 *hx = hand + (size_t)team * M, *hy = hand + (size_t)(2 + team) * M;   // hand-over buffer r holds level 4 q + r
#pragma unroll
        for (int m = 0; m < 8; m++) hx[m * T + t] = d2{xr[m], xi[m]};
        fft.forward2_fetch(yr, yi, xch, t);
        fft.pass_d_fwd(yr, yi);
        F::forward2_done();   // (a workgroup barrier: both teams' x rows are handed over)
#pragma unroll
        for (int m = 0; m < 8; m++) hy[m * T + t] = d2{yr[m], yi[m]};
#pragma unroll
        for (int r = 0; r < 2; r++) {   // fma chain over the component's rows in order: the x rows while the y rows land
          const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
          for (int m = 0; m < 8; m++) {
            const d2 d = dr[m * T + t], k = kk[r][m];
            o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
            o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
          }
        }
        workgroup_sync();
#pragma unroll
        for (int r = 2; r < 4; r++) {
          const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
          for (int m = 0; m < 8; m++) {
            const d2 d = dr[m * T + t], k = kk[r][m];
            o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
            o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
          }
        }
        if (q + 1 < QUADS || TAIL) workgroup_sync();   // the hand-over buffers are consumed (the last phase's barrier stands behind the sum below)
      }
      if constexpr (TAIL == 2) {   // levels L - 2, L - 1: one row per team
        d2 kk[2][8];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int m = 0; m < 8; m++) kk[r][m] = rows[(size_t)(L - 2 + r) * (2 * M) + m * T];
        const int sz = 64 - (L - 2 + team + 1) * Bg_bit;
        double zr[8], zi[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const int j = m * T + t;
          const uint64_t d_lo = rot_coeff<N>(accx, j, a_lo, flip) - accx[j] + off;
          const uint64_t d_hi = rot_coeff<N>(accx, j + M, a_lo, flip) - accx[j + M] + off;
          zr[m] = (double)((int)((uint32_t)(d_lo >> sz) & mask) - half);
          zi[m] = (double)((int)((uint32_t)(d_hi >> sz) & mask) - half);
        }
        fft.forward(zr, zi, xch, t);
        d2 *hz = hand + (size_t)team * M;
#pragma unroll
        for (int m = 0; m < 8; m++) hz[m * T + t] = d2{zr[m], zi[m]};
        workgroup_sync();
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
          for (int m = 0; m < 8; m++) {
            const d2 d = dr[m * T + t], k = kk[r][m];
            o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
            o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
          }
        }
      }
      if (dp == dp_lo) {
#pragma unroll
        for (int m = 0; m < 8; m++) { s_re[m] = o_re[m]; s_im[m] = o_im[m]; }
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) { s_re[m] = s_re[m] + o_re[m]; s_im[m] = s_im[m] + o_im[m]; }
      }
      workgroup_sync();   // the hand-over buffers are consumed
    }
    bool inv = true;
    if (!alone) {
      if (team != h) {   // S_h[1 - h] goes to the workgroup that keeps component 1 - h
        split_store8(send + (size_t)par * M + t, s_re, s_im);
        inv = false;
      } else {
        d2 *slot = recv + (size_t)par * M + t;
        d2 v[8];
        bool all;
        do {
          split_load8(v, slot);
          all = true;
#pragma unroll
          for (int m = 0; m < 8; m++) {   // (element copies first: __builtin_bit_cast of a vector ELEMENT reads element 0 whichever is named -- clang 22)
            const double vx = v[m].x, vy = v[m].y;
            all = all && __builtin_bit_cast(uint64_t, vx) != kSplitSentinel && __builtin_bit_cast(uint64_t, vy) != kSplitSentinel;
          }
        } while (!all);
        const double sent = __builtin_bit_cast(double, kSplitSentinel);
#pragma unroll
        for (int m = 0; m < 8; m++) {   // (IEEE addition commutes: own + partner's is S_0 + S_1 for both workgroups)
          s_re[m] = s_re[m] + v[m].x;
          s_im[m] = s_im[m] + v[m].y;
          const d2 sv = d2{sent, sent};
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slot + m * 128), "v"(sv) : "memory");
        }
      }
      par ^= 1;
    }
    if (inv) {   // (workgroup barriers inside: a team that sits it out walks the same barriers)
      fft.inverse(s_re, s_im, xch, t);
      uint64_t *accw = acc + (size_t)team * N;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        accw[m * T + t] = add_rounded<kReduce>(accw[m * T + t], s_re[m], scale);
        accw[M + m * T + t] = add_rounded<kReduce>(accw[M + m * T + t], s_im[m], scale);
      }
    } else {
      F::transform_barriers_only();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slots put back to the sentinel are at their coherence point before the barrier that precedes the next send
    workgroup_sync();
  }
  const bool mine0 = alone || h == 0, mine1 = alone || h == 1;   // which components this workgroup holds
  if (p.extract) {
    // src/trlwe.c:540-552 at idx = 0: the mask comes from component 0, the body from component 1
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    if (mine0) for (int j = tid; j < N; j += WG) dst[j] = (j == 0) ? acc[0] : (0 - acc[N - j]);
    if (mine1 && tid == 0) dst[N] = acc[N];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG)
      if (x < N ? mine0 : mine1) dst[x] = acc[x];
  }
}

// ------------------------------------------------------------------------------------------------------------
// Galois-automorphism bootstrap [src/bootstrap_ga.c:39-76] and its building blocks:
//   trlwe_keyswitch          [src/keyswitch.c:162-193]  out = (0, b) - IDFT(sum_j DFT(digit_j(a)) (.) KS[j])
//   trlwe_eval_automorphism  [src/trlwe.c:775-781, src/polynomial.c:442-450]  X -> X^gen on both components, then
//                            key switch with ak[(gen-1)/2]
//   blind_rotate_ga          acc <- Auto_{w0}(acc); per i: acc <- Auto_{gen_i}(BK_i (.) acc), BK_i = TRGSW(X^{s_i})
// Same accumulator placement as pbs_kernel: component a in VGPRs (al, ah), component b in LDS (acc1).
// ak entry layout = L key rows of a bootstrap-key entry: [L][2][8][T] complex, slot order.
// ------------------------------------------------------------------------------------------------------------
// acc <- TRGSW (.) acc (external product, result REPLACES the accumulator)
template <class F, int L, int BG>
__device__ __forceinline__ void ga_external_product(uint64_t (&al)[8], uint64_t (&ah)[8], uint64_t *acc1, d2 *xch, const F &fft,
                                                    const d2 *__restrict__ bkrow, uint64_t off, int Bg_bit, const RoundCtx &scale, int t) {
  constexpr int M = F::M, T = F::THREADS;
  using D = Digits<L, BG>;
  double o_re[2][8], o_im[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
  for (int q = 0; q < 2; q++) {
    typename D::word_t w_lo[8], w_hi[8];
    uint32_t ext[8];
    if (q == 0) {
#pragma unroll
      for (int m = 0; m < 8; m++) D::pack(w_lo[m], w_hi[m], ext[m], al[m] + off, ah[m] + off);
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) D::pack(w_lo[m], w_hi[m], ext[m], acc1[m * T + t] + off, acc1[M + m * T + t] + off);
    }
    cmux_rows<F, L, BG>(w_lo, w_hi, ext, q, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
  }
  fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
#pragma unroll
  for (int m = 0; m < 8; m++) {
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

// acc <- Auto_gen(acc): permute both components (out[(i gen) mod N] = +-in[i]) and key switch with `entry`
template <class F, int L, int BG>
__device__ __forceinline__ void ga_eval_automorphism(uint64_t (&al)[8], uint64_t (&ah)[8], uint64_t *acc1, d2 *xch, const F &fft,
                                                     const d2 *__restrict__ entry, int gen, uint64_t off, int Bg_bit, const RoundCtx &scale,
                                                     int t) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  using D = Digits<L, BG>;
  typename D::word_t w_lo[8], w_hi[8];
  uint32_t ext[8];
  {
    // component a: scatter through the staging buffer, read back in layout A, keep only the digit words
    uint64_t *st = reinterpret_cast<uint64_t *>(xch);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int i0 = (m * T + t) * gen, i1 = (M + m * T + t) * gen;
      st[i0 & (N - 1)] = (i0 & N) ? (0 - al[m]) : al[m];
      st[i1 & (N - 1)] = (i1 & N) ? (0 - ah[m]) : ah[m];
    }
    F::sync();
#pragma unroll
    for (int m = 0; m < 8; m++) D::pack(w_lo[m], w_hi[m], ext[m], st[m * T + t] + off, st[M + m * T + t] + off);
    F::sync();
  }
  {
    // component b: permute in place in LDS (read everything, barrier, scatter)
    uint64_t v_lo[8], v_hi[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { v_lo[m] = acc1[m * T + t]; v_hi[m] = acc1[M + m * T + t]; }
    F::sync();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int i0 = (m * T + t) * gen, i1 = (M + m * T + t) * gen;
      acc1[i0 & (N - 1)] = (i0 & N) ? (0 - v_lo[m]) : v_lo[m];
      acc1[i1 & (N - 1)] = (i1 & N) ? (0 - v_hi[m]) : v_hi[m];
    }
    F::sync();
  }
  double o_re[2][8], o_im[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
  cmux_rows<F, L, BG>(w_lo, w_hi, ext, 0, o_re, o_im, xch, fft, entry, Bg_bit, t);
  fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
#pragma unroll
  for (int m = 0; m < 8; m++) {
    al[m] = 0 - round_mod_2_64(o_re[0][m], scale);
    ah[m] = 0 - round_mod_2_64(o_im[0][m], scale);
  }
#pragma unroll
  for (int m = 0; m < 8; m++) {
    acc1[m * T + t] -= round_mod_2_64(o_re[1][m], scale);
    acc1[M + m * T + t] -= round_mod_2_64(o_im[1][m], scale);
  }
  F::sync();
}

// inverse of odd x modulo 2N (a power of two) by Newton iteration [src/misc.c:142-159 tabulates it]
__device__ __forceinline__ uint32_t inverse_mod_2n(uint32_t x, uint32_t mask) {
  uint32_t inv = x;
#pragma unroll
  for (int i = 0; i < 4; i++) inv = (inv * (2u - x * inv)) & mask;
  return inv;
}

struct GaParams {
  PbsParams p;
  const d2 *__restrict__ ak;  // [N][L][2][8][T] complex: automorphism key-switch keys, entry (gen - 1) / 2
  int mode;                   // 0: functional_bootstrap(_wo_extract)_ga (p.skip_init: blind_rotate_ga on the accumulators in p.out);
                              // 1: one trlwe_eval_automorphism with generator `gen`
  int gen;
  int entry;                  // mode 1: key-set entry to switch with; < 0: (gen - 1) / 2 (trlwe_new_automorphism_KS_keyset's order)
};

template <class F, int L, int BG>
__global__ __launch_bounds__(F::THREADS, 2) void pbs_ga_kernel(GaParams g) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  __shared__ __attribute__((aligned(16))) uint64_t acc1[N];
  const PbsParams &p = g.p;
  const int t = threadIdx.x;
  const size_t b = blockIdx.x;
  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft_setup(fft, p.tw, t);
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * L * 2 * M, ak_sz = (size_t)L * 2 * M;
  uint64_t al[8], ah[8];

  if (g.mode == 1) {
    // trlwe_eval_automorphism on a batch of TRLWE samples: in = p.in [B][2][N], out = p.out
    const uint64_t *src = p.in + b * (size_t)(2 * N);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      al[m] = src[m * T + t];
      ah[m] = src[M + m * T + t];
      acc1[m * T + t] = src[N + m * T + t];
      acc1[M + m * T + t] = src[N + M + m * T + t];
    }
    F::sync();
    ga_eval_automorphism<F, L, BG>(al, ah, acc1, xch, fft, g.ak + (size_t)(g.entry >= 0 ? g.entry : (g.gen - 1) >> 1) * ak_sz, g.gen, off, Bg_bit, scale, t);
  } else {
    const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
    if (p.skip_init) {
      // blind_rotate_ga (src/bootstrap_ga.c:35-60) on a caller-supplied accumulator
      const uint64_t *src = p.out + b * (size_t)(2 * N);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        al[m] = src[m * T + t];
        ah[m] = src[M + m * T + t];
        acc1[m * T + t] = src[N + m * T + t];
        acc1[M + m * T + t] = src[N + M + m * T + t];
      }
    } else {
      // src/bootstrap_ga.c:64-65: acc = tv * X^(2N - bbar)
      const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
      const uint32_t bbar = modswitch<LOG2N2>(ct[p.n] + p.prec_offset);
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
    const uint32_t mask = 2 * N - 1;
    // src/bootstrap_ga.c:44-45: w0' = (a_0 | 1)^-1 ; acc = Auto_{w0'}(acc)
    uint32_t a_cur = modswitch<LOG2N2>(ct[0]) | 1u;
    {
      const int gen = (int)inverse_mod_2n(a_cur, mask);
      ga_eval_automorphism<F, L, BG>(al, ah, acc1, xch, fft, g.ak + (size_t)((gen - 1) >> 1) * ak_sz, gen, off, Bg_bit, scale, t);
    }
    for (int i = 0; i < p.n; i++) {
      if (T > 64 && p.pace && i > 0 && i % p.pace_every == 0) pace_teams(p.pace, (unsigned)(i / p.pace_every), t, p.pace_limit);
      // :46-58: acc = BK_i (.) acc ; gen = a_i * (a_{i+1})^-1 (last step: a_{n-1}) ; acc = Auto_gen(acc)
      int gen;
      if (i + 1 < p.n) {
        const uint32_t a_next = modswitch<LOG2N2>(ct[i + 1]) | 1u;
        gen = (int)((a_cur * inverse_mod_2n(a_next, mask)) & mask);
        a_cur = a_next;
      } else {
        gen = (int)a_cur;
      }
      ga_external_product<F, L, BG>(al, ah, acc1, xch, fft, p.bk + (size_t)i * row_sz, off, Bg_bit, scale, t);
      ga_eval_automorphism<F, L, BG>(al, ah, acc1, xch, fft, g.ak + (size_t)((gen - 1) >> 1) * ak_sz, gen, off, Bg_bit, scale, t);
    }
  }

  if (g.mode == 0 && p.extract) {
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

// ------------------------------------------------------------------------------------------------------------
// Galois-automorphism bootstrap for FEW ciphertexts: two transform teams per ciphertext (cf. pbs_wide_team_kernel), run-time gadget.  Both accumulator
// components live in LDS.  Every product -- the external product with BK_i (2l rows) and the key switch of an automorphism (l rows, component a only) --
// runs its rows two at a time: team w transforms the digits of row 2 ph + w, both teams multiply-accumulate the phase's rows in row order (team c the
// output component c: the fma chain of cmux_rows), one inverse transform each.  Same digits, same chains, same rounding as pbs_ga_kernel: bit-identical.
// A team without a row in a ragged last phase (odd l in the key switch) executes the transform's barriers only.
// ------------------------------------------------------------------------------------------------------------
template <class F>
struct GaWide {
  static constexpr int N = F::N, M = F::M, T = F::THREADS, WG = 2 * T;
  const F &fft;
  d2 *xch_all, *xch;
  uint64_t *acc;   // [2][N]
  int team, t, tid, l, Bg_bit;
  uint64_t off;
  uint32_t mask;
  int half;

  // o = sum over `rows` rows: DFT(digit(level r % l of component (comp_of_row0 + r / l))) (.) key[r][team]   (rows in order)
  __device__ __forceinline__ void product(double (&o_re)[8], double (&o_im)[8], const d2 *__restrict__ key, int rows) const {
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
#pragma unroll 1
    for (int r0 = 0; r0 < rows; r0 += 2) {
      const int in_phase = rows - r0 < 2 ? rows - r0 : 2, row = r0 + team;
      d2 kk[2][8];
#pragma unroll
      for (int r = 0; r < 2; r++)
        if (r < in_phase)
#pragma unroll
          for (int m = 0; m < 8; m++) kk[r][m] = key[(size_t)(r0 + r) * (2 * M) + (size_t)team * M + m * T + t];
      if (row < rows) {
        const uint64_t *accq = acc + (size_t)(row / l) * N;
        const int shift = 64 - (row % l + 1) * Bg_bit;
        double re[8], im[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
          re[m] = (double)((int)((uint32_t)((accq[m * T + t] + off) >> shift) & mask) - half);
          im[m] = (double)((int)((uint32_t)((accq[M + m * T + t] + off) >> shift) & mask) - half);
        }
        fft.forward(re, im, xch, t);
#pragma unroll
        for (int m = 0; m < 8; m++) xch[m * T + t] = d2{re[m], im[m]};
      } else {
        F::transform_barriers_only();
      }
      workgroup_sync();
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (r >= in_phase) continue;
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
  }

  // acc <- BK (.) acc   (src/bootstrap_ga.c:50: the product replaces the accumulator)
  __device__ __forceinline__ void external_product(const d2 *__restrict__ bkrow, const RoundCtx &scale) const {
    double o_re[8], o_im[8];
    product(o_re, o_im, bkrow, 2 * l);
    uint64_t *accw = acc + (size_t)team * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      accw[m * T + t] = round_mod_2_64(o_re[m], scale);
      accw[M + m * T + t] = round_mod_2_64(o_im[m], scale);
    }
    workgroup_sync();
  }

  // acc <- Auto_gen(acc): permute both components in place (out[(i gen) mod N] = +-in[i]), then key switch component a with `entry`:
  // a = -as(a), b = b - as(a) [src/trlwe.c:775-781, src/keyswitch.c:162-193]
  __device__ __forceinline__ void eval_automorphism(const d2 *__restrict__ entry, int gen, const RoundCtx &scale) const {
    constexpr int PER = 2 * N / WG;
    uint64_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) v[j] = acc[j * WG + tid];
    workgroup_sync();
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int x = j * WG + tid, c = x / N, i = x & (N - 1), ig = i * gen;
      acc[c * N + (ig & (N - 1))] = (ig & N) ? (0 - v[j]) : v[j];
    }
    workgroup_sync();
    double o_re[8], o_im[8];
    product(o_re, o_im, entry, l);
    uint64_t *accw = acc + (size_t)team * N;
    if (team == 0) {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        accw[m * T + t] = 0 - round_mod_2_64(o_re[m], scale);
        accw[M + m * T + t] = 0 - round_mod_2_64(o_im[m], scale);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        accw[m * T + t] -= round_mod_2_64(o_re[m], scale);
        accw[M + m * T + t] -= round_mod_2_64(o_im[m], scale);
      }
    }
    workgroup_sync();
  }
};

template <class F>
__global__ __launch_bounds__(2 * F::THREADS) void pbs_ga_wide_kernel(GaParams g, int l) {
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, WG = 2 * T;
  extern __shared__ __attribute__((aligned(16))) unsigned char ga_lds[];
  d2 *xch_all = reinterpret_cast<d2 *>(ga_lds);
  uint64_t *acc = reinterpret_cast<uint64_t *>(ga_lds + sizeof(d2) * (size_t)2 * F::XCH_SLOTS);
  const PbsParams &p = g.p;
  const int tid = threadIdx.x, team = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  const size_t b = blockIdx.x;
  const int Bg_bit = p.Bg_bit;
  F fft;
  fft.init(p.tw, t);
  uint64_t off = 1ull << (63 - l * Bg_bit);
  for (int i = 0; i < l; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * l * 2 * M, ak_sz = (size_t)l * 2 * M;
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) acc[x] = src[x];
  } else {
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(ct[p.n] + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += WG) acc[x] = rot_coeff<N>(tv + (x / N) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  const GaWide<F> w{fft, xch_all, xch_all + (size_t)team * F::XCH_SLOTS, acc, team, t, tid, l, Bg_bit, off, (1u << Bg_bit) - 1, 1 << (Bg_bit - 1)};
  const uint32_t mask = 2 * N - 1;
  uint32_t a_cur = modswitch<LOG2N2>(ct[0]) | 1u;
  {
    const int gen = (int)inverse_mod_2n(a_cur, mask);
    w.eval_automorphism(g.ak + (size_t)((gen - 1) >> 1) * ak_sz, gen, scale);
  }
  for (int i = 0; i < p.n; i++) {
    int gen;
    if (i + 1 < p.n) {
      const uint32_t a_next = modswitch<LOG2N2>(ct[i + 1]) | 1u;
      gen = (int)((a_cur * inverse_mod_2n(a_next, mask)) & mask);
      a_cur = a_next;
    } else {
      gen = (int)a_cur;
    }
    w.external_product(p.bk + (size_t)i * row_sz, scale);
    w.eval_automorphism(g.ak + (size_t)((gen - 1) >> 1) * ak_sz, gen, scale);
  }
  if (p.extract) {
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    for (int j = tid; j < N; j += WG) dst[j] = (j == 0) ? acc[0] : (0 - acc[N - j]);
    if (tid == 0) dst[N] = acc[N];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) dst[x] = acc[x];
  }
}

// ------------------------------------------------------------------------------------------------------------
// pbs_ga_split_kernel: the Galois-automorphism bootstrap on TWO workgroups, like pbs_split_kernel (which see: pairing, sentinel slots, the take-it-alone fall-back).
// Workgroup 0 keeps accumulator component a, workgroup 1 component b.  Per step (src/bootstrap_ga.c:48-52):
//   * acc <- BK_i (.) acc: each workgroup decomposes ITS component (l = 4 rows, the double phase of pbs_wide_pair_kernel), multiplies into both output components,
//     sends the partial sum it does not keep, adds its partner's, inverse-transforms and REPLACES its component -- the per-component summation order of
//     pbs_split_kernel (oracle: orc_set_product_order(1); orc_blind_rotate_ga goes through the same external product);
//   * acc <- Auto_gen(acc): each workgroup permutes its own component; the key switch of the permuted a (src/keyswitch.c:162-193: l digit polynomials against the
//     key's l rows, ONE chain -- the reference's order, nothing to split) runs on workgroup 0, which has a: team 0 gets -as(a) for its own component, team 1's sum
//     goes to workgroup 1 (a second slot set), whose team 1 inverse-transforms it and subtracts it from the permuted b.
// Two exchanges per step; a step is two forward pairs and two inverse transforms deep instead of the three forward phases and two inverse pairs of one CU.
// ------------------------------------------------------------------------------------------------------------
template <class F, int L, int BG>
__global__ __launch_bounds__(2 * F::THREADS) void pbs_ga_split_kernel(GaParams g, SplitParams sp) {
  static_assert(F::kForward2 && F::kLtw && L == 4 && F::THREADS == 128, "one double phase = the l = 4 rows of a component / of an automorphism key");
  constexpr int N = F::N, M = F::M, T = F::THREADS, LOG2N2 = F::LOGM + 2, WG = 2 * T;
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
  d2 *xch_all = reinterpret_cast<d2 *>(wide_lds);                                   // [2][F::XCH_SLOTS]
  d2 *hand = xch_all + (size_t)2 * F::XCH_SLOTS;                                    // [4][M]
  uint64_t *acc = reinterpret_cast<uint64_t *>(hand + (size_t)4 * M);               // [2][N] (paired: only component h is kept up to date)
  __shared__ int mode_s;
  const PbsParams &p = g.p;
  const int tid = threadIdx.x, team = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  d2 *xch = xch_all + (size_t)team * F::XCH_SLOTS;
  const int h = (int)((blockIdx.x >> 3) & 1u);
  const size_t b = (size_t)(blockIdx.x >> 4) * 8 + (blockIdx.x & 7u);
  if (b >= (size_t)sp.count) return;
  if (tid == 0) {   // pairing: see pbs_split_kernel
    unsigned int *st = sp.state + b;
    int mode;
    if (sp.limit <= 0) {
      mode = h == 0 ? 1 : 2;
    } else {
      unsigned int seen = 0u;
      if (__hip_atomic_compare_exchange_strong(st, &seen, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        const long long t0 = wall_clock64();
        mode = -1;
        while (mode < 0) {
          if (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) mode = 0;
          else if (wall_clock64() - t0 > sp.limit) {
            unsigned int one = 1u;
            mode = __hip_atomic_compare_exchange_strong(st, &one, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
          } else __builtin_amdgcn_s_sleep(8);
        }
      } else if (seen == 1u) {
        unsigned int one = 1u;
        mode = __hip_atomic_compare_exchange_strong(st, &one, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0 : 2;
      } else {
        mode = 2;
      }
    }
    mode_s = mode;
  }
  workgroup_sync();
  const int mode = mode_s;
  if (mode == 2) return;
  const bool alone = mode == 1;
  const int dp_lo = alone ? 0 : h, dp_hi = alone ? 2 : h + 1;
  d2 *recv_ep = sp.xbuf + ((size_t)b * 2 + (size_t)h) * 2 * M, *send_ep = sp.xbuf + ((size_t)b * 2 + (size_t)(1 - h)) * 2 * M;
  d2 *slot_ks = sp.xbuf + (size_t)sp.count * 4 * M + (size_t)b * 2 * M;   // [2 parities][M]: workgroup 0's team 1 -> workgroup 1's team 1

  const int Bg_bit = BG > 0 ? BG : p.Bg_bit;
  F fft;
  fft_setup(fft, p.tw, t);
  const uint64_t *__restrict__ ct = p.in + b * (size_t)(p.n + 1);
  if (p.skip_init) {
    const uint64_t *src = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG) acc[x] = src[x];
  } else {
    const uint64_t *__restrict__ tv = p.tv + b * (size_t)p.tv_stride;
    const uint32_t bbar = modswitch<LOG2N2>(ct[p.n] + p.prec_offset);
    const int rot = (2 * N - (int)bbar) & (2 * N - 1);
    const int a_lo = rot & (N - 1);
    const bool flip = (rot & N) != 0;
    for (int x = tid; x < 2 * N; x += WG) acc[x] = rot_coeff<N>(tv + (x / N) * N, x & (N - 1), a_lo, flip);
  }
  workgroup_sync();
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  const size_t row_sz = (size_t)2 * L * 2 * M, ak_sz = (size_t)L * 2 * M;
  const uint32_t dmask = (1u << Bg_bit) - 1;
  const int half = 1 << (Bg_bit - 1);
  const int sx = 64 - (team + 1) * Bg_bit, sy = 64 - (team + 3) * Bg_bit;   // this team's levels: team (x) and team + 2 (y)
  const double sent = __builtin_bit_cast(double, kSplitSentinel);

  // o = the chain over the four rows `rows` (this team's output component, lane offset included) of DFT(digit_level(accx)): the double phase of pbs_wide_pair_kernel
  auto quad = [&](const uint64_t *accx, const d2 *__restrict__ rows, double (&o_re)[8], double (&o_im)[8]) {
    d2 kk[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int m = 0; m < 8; m++) kk[r][m] = rows[(size_t)r * (2 * M) + m * T];
#pragma unroll
    for (int m = 0; m < 8; m++) { o_re[m] = 0.0; o_im[m] = 0.0; }
    double xr[8], xi[8], yr[8], yi[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const uint64_t d_lo = accx[m * T + t] + off, d_hi = accx[M + m * T + t] + off;
      xr[m] = (double)((int)((uint32_t)(d_lo >> sx) & dmask) - half);
      xi[m] = (double)((int)((uint32_t)(d_hi >> sx) & dmask) - half);
      yr[m] = (double)((int)((uint32_t)(d_lo >> sy) & dmask) - half);
      yi[m] = (double)((int)((uint32_t)(d_hi >> sy) & dmask) - half);
    }
    fft.forward2_head(xr, xi, yr, yi, xch, t);
    fft.pass_d_fwd(xr, xi);
    d2 *hx = hand + (size_t)team * M, *hy = hand + (size_t)(2 + team) * M;
#pragma unroll
    for (int m = 0; m < 8; m++) hx[m * T + t] = d2{xr[m], xi[m]};
    fft.forward2_fetch(yr, yi, xch, t);
    fft.pass_d_fwd(yr, yi);
    F::forward2_done();
#pragma unroll
    for (int m = 0; m < 8; m++) hy[m * T + t] = d2{yr[m], yi[m]};
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const d2 d = dr[m * T + t], k = kk[r][m];
        o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
        o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
      }
    }
    workgroup_sync();
#pragma unroll
    for (int r = 2; r < 4; r++) {
      const d2 *__restrict__ dr = hand + (size_t)r * M;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const d2 d = dr[m * T + t], k = kk[r][m];
        o_re[m] = __builtin_fma(-d.y, k.y, __builtin_fma(d.x, k.x, o_re[m]));
        o_im[m] = __builtin_fma(d.y, k.x, __builtin_fma(d.x, k.y, o_im[m]));
      }
    }
    workgroup_sync();   // the hand-over buffers are consumed
  };
  // the lane's eight items of a slot set: wait for them, put the sentinel back
  auto receive = [&](d2 *slot, d2 (&v)[8]) {
    bool all;
    do {
      split_load8(v, slot);
      all = true;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const double vx = v[m].x, vy = v[m].y;
        all = all && __builtin_bit_cast(uint64_t, vx) != kSplitSentinel && __builtin_bit_cast(uint64_t, vy) != kSplitSentinel;
      }
    } while (!all);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const d2 sv = d2{sent, sent};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slot + m * 128), "v"(sv) : "memory");
    }
  };
  int par_ep = 0, par_ks = 0;

  // acc <- BK (.) acc (the product REPLACES the accumulator: src/bootstrap_ga.c:50)
  auto external_product = [&](const d2 *__restrict__ bkrow) {
    double s_re[8], s_im[8];
#pragma unroll 1
    for (int dp = dp_lo; dp < dp_hi; dp++) {
      double o_re[8], o_im[8];
      quad(acc + (size_t)dp * N, bkrow + (size_t)(dp * L) * (2 * M) + (size_t)team * M + t, o_re, o_im);
      if (dp == dp_lo) {
#pragma unroll
        for (int m = 0; m < 8; m++) { s_re[m] = o_re[m]; s_im[m] = o_im[m]; }
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) { s_re[m] = s_re[m] + o_re[m]; s_im[m] = s_im[m] + o_im[m]; }
      }
    }
    bool inv = true;
    if (!alone) {
      if (team != h) {
        split_store8(send_ep + (size_t)par_ep * M + t, s_re, s_im);
        inv = false;
      } else {
        d2 v[8];
        receive(recv_ep + (size_t)par_ep * M + t, v);
#pragma unroll
        for (int m = 0; m < 8; m++) { s_re[m] = s_re[m] + v[m].x; s_im[m] = s_im[m] + v[m].y; }
      }
      par_ep ^= 1;
    }
    if (inv) {
      fft.inverse(s_re, s_im, xch, t);
      uint64_t *accw = acc + (size_t)team * N;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        accw[m * T + t] = round_mod_2_64(s_re[m], scale);
        accw[M + m * T + t] = round_mod_2_64(s_im[m], scale);
      }
    } else {
      F::transform_barriers_only();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the slots put back are at their coherence point before the barrier that precedes the next send)
    workgroup_sync();
  };

  // acc <- Auto_gen(acc) [src/trlwe.c:775-781, src/keyswitch.c:162-193]: permute (out[(i gen) mod N] = +-in[i]), then a = -as(a), b = b - as(a)
  auto eval_automorphism = [&](const d2 *__restrict__ entry, int gen) {
    constexpr int PER = N / WG;
    for (int c = dp_lo; c < dp_hi; c++) {   // the components this workgroup keeps
      uint64_t *ac = acc + (size_t)c * N;
      uint64_t v[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) v[j] = ac[j * WG + tid];
      workgroup_sync();
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int i = j * WG + tid, ig = i * gen;
        ac[ig & (N - 1)] = (ig & N) ? (0 - v[j]) : v[j];
      }
      workgroup_sync();
    }
    double o_re[8], o_im[8];
    if (alone || h == 0) quad(acc, entry + (size_t)team * M + t, o_re, o_im);   // the key switch of the permuted a: team w's output component w
    bool inv = alone || team == h;
    if (!alone) {
      if (h == 0 && team == 1) split_store8(slot_ks + (size_t)par_ks * M + t, o_re, o_im);
      if (h == 1 && team == 1) {
        d2 v[8];
        receive(slot_ks + (size_t)par_ks * M + t, v);
#pragma unroll
        for (int m = 0; m < 8; m++) { o_re[m] = v[m].x; o_im[m] = v[m].y; }
      }
      par_ks ^= 1;
    }
    if (inv) {
      fft.inverse(o_re, o_im, xch, t);
      uint64_t *accw = acc + (size_t)team * N;
      if (team == 0) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
          accw[m * T + t] = 0 - round_mod_2_64(o_re[m], scale);
          accw[M + m * T + t] = 0 - round_mod_2_64(o_im[m], scale);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) {
          accw[m * T + t] -= round_mod_2_64(o_re[m], scale);
          accw[M + m * T + t] -= round_mod_2_64(o_im[m], scale);
        }
      }
    } else {
      F::transform_barriers_only();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    workgroup_sync();
  };

  const uint32_t mask2n = 2 * N - 1;
  uint32_t a_cur = modswitch<LOG2N2>(ct[0]) | 1u;
  {
    const int gen = (int)inverse_mod_2n(a_cur, mask2n);
    eval_automorphism(g.ak + (size_t)((gen - 1) >> 1) * ak_sz, gen);
  }
  for (int i = 0; i < p.n; i++) {
    int gen;
    if (i + 1 < p.n) {
      const uint32_t a_next = modswitch<LOG2N2>(ct[i + 1]) | 1u;
      gen = (int)((a_cur * inverse_mod_2n(a_next, mask2n)) & mask2n);
      a_cur = a_next;
    } else {
      gen = (int)a_cur;
    }
    external_product(p.bk + (size_t)i * row_sz);
    eval_automorphism(g.ak + (size_t)((gen - 1) >> 1) * ak_sz, gen);
  }
  const bool mine0 = alone || h == 0, mine1 = alone || h == 1;
  if (p.extract) {
    uint64_t *dst = p.out + b * (size_t)(N + 1);
    if (mine0) for (int j = tid; j < N; j += WG) dst[j] = (j == 0) ? acc[0] : (0 - acc[N - j]);
    if (mine1 && tid == 0) dst[N] = acc[N];
  } else {
    uint64_t *dst = p.out + b * (size_t)(2 * N);
    for (int x = tid; x < 2 * N; x += WG)
      if (x < N ? mine0 : mine1) dst[x] = acc[x];
  }
}

// ------------------------------------------------------------------------------------------------------------
// FFT-based TRLWE key switch with run-time (t, base_bit) [src/keyswitch.c:162-193] and trlwe_priv_keyswitch_2
// [src/keyswitch.c:52-63], used by circuit_bootstrap_3 (kska: t = 20, base_bit = 2 in the reference's test).
// One team per TRLWE sample; everything stays in registers (thread owns coefficients m*T+t and m*T+t+M).
// Key entry layout: [t][2][8][T] complex (slot order), as produced by torus_to_dft_kernel.
// ------------------------------------------------------------------------------------------------------------
// o += sum_{j<t} DFT(digit_j(a)) (.) entry[j][c]   (digits of a_lo/a_hi + offset, rounded rule of polynomial_decompose_i)
template <class F>
__device__ __forceinline__ void ks_rows_rt(const uint64_t (&dd_lo)[8], const uint64_t (&dd_hi)[8], double (&o_re)[2][8], double (&o_im)[2][8],
                                           d2 *xch, const F &fft, const d2 *__restrict__ entry, int t, int base_bit, int tid) {
  constexpr int M = F::M, T = F::THREADS;
  const uint32_t mask = (1u << base_bit) - 1;
  const int half = 1 << (base_bit - 1);
#pragma unroll 1
  for (int j = 0; j < t; j++) {
    const d2 *__restrict__ row = entry + (size_t)j * (2 * M);
    const int shift = 64 - (j + 1) * base_bit;
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      re[m] = (double)((int)((uint32_t)(dd_lo[m] >> shift) & mask) - half);
      im[m] = (double)((int)((uint32_t)(dd_hi[m] >> shift) & mask) - half);
    }
    fft.forward_head(re, im, xch, tid);
    d2 k0[8], k1[8];
#pragma unroll
    for (int m = 0; m < 8; m++) k0[m] = row[m * T + tid];
    fft.forward_tail(re, im);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      k1[m] = row[M + m * T + tid];
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

// mode 0: out = trlwe_keyswitch(in, ks0)                       = (0, in.b) - as(in.a; ks0)
// mode 1: out = trlwe_priv_keyswitch_2(in, {ks0, ks1})         = -as(in.a; ks0) - as(-in.b; ks1)
// mode 2: out = base - trlwe_keyswitch(in, ks0)                = base - (0, in.b) + as(in.a; ks0)   (relinearisation step of
//         trlwe_tensor_prod_FFT, src/trlwe.c:758-761)
// One wavefront per SIMD (launch bound 1): the two result pairs, the digit words and the product accumulators are ~330 live registers; with room for
// the overflow in the accumulation registers there is no scratch (two per SIMD: 388 bytes of it).  The kernel is 0.16 ms of a 38 ms circuit bootstrap.
template <class F>
__global__ __launch_bounds__(F::THREADS, 1) void trlwe_fft_keyswitch_kernel(const d2 *__restrict__ ks0, const d2 *__restrict__ ks1,
                                                                          const d2 *__restrict__ tw, const uint64_t *in,
                                                                          size_t in_stride, uint64_t *out, size_t out_stride,   // in place is relied upon (src/trlwe.c:780): no __restrict__
                                                                          int t, int base_bit, int mode,
                                                                          const uint64_t *__restrict__ base = nullptr, size_t base_stride = 0) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int tid = threadIdx.x;
  const uint64_t *c = in + (size_t)blockIdx.x * in_stride;
  uint64_t *o = out + (size_t)blockIdx.x * out_stride;
  F fft;
  fft.init(tw, tid);
  uint64_t off = 1ull << (63 - t * base_bit);
  for (int i = 0; i < t; i++) off += 1ull << (63 - i * base_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  uint64_t res_a_lo[8], res_a_hi[8], res_b_lo[8], res_b_hi[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    res_a_lo[m] = 0;
    res_a_hi[m] = 0;
    res_b_lo[m] = mode == 0 ? c[N + m * T + tid] : 0;
    res_b_hi[m] = mode == 0 ? c[N + M + m * T + tid] : 0;
    if (mode == 2) {  // kept negated: res = -(base - (0, in.b)), the common "-=" below then yields -(out), negated at the store
      const uint64_t *bs = base + (size_t)blockIdx.x * base_stride;
      res_a_lo[m] = 0 - bs[m * T + tid];
      res_a_hi[m] = 0 - bs[M + m * T + tid];
      res_b_lo[m] = c[N + m * T + tid] - bs[N + m * T + tid];
      res_b_hi[m] = c[N + M + m * T + tid] - bs[N + M + m * T + tid];
    }
  }
  const int passes = mode == 1 ? 2 : 1;
#pragma unroll 1
  for (int pass = 0; pass < passes; pass++) {
    // pass 0 of mode 1: a' = -in.b with ks1 (src/keyswitch.c:55-57); otherwise a' = in.a with ks0
    const bool neg_b = (mode == 1 && pass == 0);
    const d2 *__restrict__ entry = neg_b ? ks1 : ks0;
    uint64_t dd_lo[8], dd_hi[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const uint64_t x_lo = neg_b ? (0 - c[N + m * T + tid]) : c[m * T + tid];
      const uint64_t x_hi = neg_b ? (0 - c[N + M + m * T + tid]) : c[M + m * T + tid];
      dd_lo[m] = x_lo + off;
      dd_hi[m] = x_hi + off;
    }
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[cc][m] = 0.0; o_im[cc][m] = 0.0; }
    ks_rows_rt<F>(dd_lo, dd_hi, o_re, o_im, xch, fft, entry, t, base_bit, tid);
    fft.inverse(o_re[0], o_im[0], xch, tid);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      res_a_lo[m] -= round_mod_2_64(o_re[0][m], scale);
      res_a_hi[m] -= round_mod_2_64(o_im[0][m], scale);
    }
    fft.inverse(o_re[1], o_im[1], xch, tid);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      res_b_lo[m] -= round_mod_2_64(o_re[1][m], scale);
      res_b_hi[m] -= round_mod_2_64(o_im[1][m], scale);
    }
  }
#pragma unroll
  for (int m = 0; m < 8; m++) {
    o[m * T + tid] = mode == 2 ? 0 - res_a_lo[m] : res_a_lo[m];
    o[M + m * T + tid] = mode == 2 ? 0 - res_a_hi[m] : res_a_hi[m];
    o[N + m * T + tid] = mode == 2 ? 0 - res_b_lo[m] : res_b_lo[m];
    o[N + M + m * T + tid] = mode == 2 ? 0 - res_b_hi[m] : res_b_hi[m];
  }
}

// trgsw_to_DFT / polynomial_torus_to_DFT for a flat array of polynomials [src/trgsw.c:345-349,
// src/polynomial.c:368-375]: one team per polynomial, output in slot order [m][thread].
template <class F>
__global__ __launch_bounds__(F::THREADS) void torus_to_dft_kernel(const uint64_t *__restrict__ in, d2 *__restrict__ out,
                                                                const d2 *__restrict__ tw) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int t = threadIdx.x;
  const uint64_t *src = in + (size_t)blockIdx.x * N;
  F fft;
  fft.init(tw, t);
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    re[m] = torus_to_double(src[m * T + t]);
    im[m] = torus_to_double(src[m * T + t + M]);
  }
  fft.forward(re, im, xch, t);
  d2 *dst = out + (size_t)blockIdx.x * M;
#pragma unroll
  for (int m = 0; m < 8; m++) dst[m * T + t] = d2{re[m], im[m]};
}

// polynomial_DFT_to_torus for a flat array [src/polynomial.c:359-366]
template <class F>
__global__ __launch_bounds__(F::THREADS) void dft_to_torus_kernel(const d2 *__restrict__ in, uint64_t *__restrict__ out,
                                                                const d2 *__restrict__ tw) {
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int t = threadIdx.x;
  const d2 *src = in + (size_t)blockIdx.x * M;
  F fft;
  fft.init(tw, t);
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; m++) { const d2 v = src[m * T + t]; re[m] = v.x; im[m] = v.y; }
  fft.inverse(re, im, xch, t);
  uint64_t *dst = out + (size_t)blockIdx.x * N;
  const RoundCtx scale(0x1p-64 / (double)M);
#pragma unroll
  for (int m = 0; m < 8; m++) {
    dst[m * T + t] = round_mod_2_64(re[m], scale);
    dst[m * T + t + M] = round_mod_2_64(im[m], scale);
  }
}

// trgsw_mul_trlwe_DFT + trlwe_from_DFT for a batch [src/trgsw.c:385-423, src/trlwe.c:629-634]: out[b] = TRGSW (.) in[b], back in the torus domain.
// The HBM-bound kernel of the path: per unit one TRLWE in and one out (32 KiB at N = 1024) against a key entry that stays in the caches.
//   * persistent teams: the grid is the chip's resident capacity and every team walks units b, b + grid, ... -- twiddles, rounding constants and the
//     launch cost are paid once per team, and half a ciphertext is always in flight from HBM (software pipeline below);
//   * BG > 0: gadget base known at compile time (packed digit words, as in pbs_kernel); BG = 0: run-time Bg_bit;
//   * the two inverse transforms are pipelined through the one transpose buffer (inverse2), the first key row initialises the accumulators.
// key_stride (in d2): 0 = one TRGSW for the whole batch, else TRGSW b starts at bkrow + b * key_stride (per-ciphertext selectors,
// functional_bootstrap_trgsw_phase2); in_stride (words): 0 = one shared TRLWE input.
// in0 != nullptr: CMUX (applications/leveled_lut/vertical_packing.c:24-33): out[b] = in0[b] + TRGSW (.) (in[b] - in0[b])  (out may alias in0)
// out_dft != nullptr: trgsw_mul_trlwe_DFT as the reference declares it (include/mosfhet.h:344): the result stays in the DFT domain,
// out_dft[b][c][slot] in slot order; trlwe_from_DFT (dft_to_torus_kernel) finishes it with the same inverse transform and rounding.
#define EP_NT_LOAD(p) __builtin_nontemporal_load(p)
#define EP_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
// The unit loop comes in two forms: PLAIN (each component requested where it is used) and PIPELINED (the next component always in flight under the rows of the
// current one).  FORM = 0 takes the form ep_pipelined_by_default() names, 1 forces the plain loop (the launcher's fall-back, capi.hip: ep_form), 2 forces the
// pipelined one (experiments only: tools/ab/ep_ab.hip, tools/spill_hazard).
template <class F, int L, bool CMUX>
constexpr bool ep_pipelined_by_default() {
  // one-wavefront teams always; two-wavefront teams with the rows taken in pairs (pass twiddles in LDS) at l = 4 without the CMUX operand -- the one multi-wavefront
  // instantiation that is FASTER pipelined (-11 % at lvl2); the others are slower that way (see the note in the kernel body)
  return F::THREADS == 64 || (F::kForward2 && F::kLtw && L % 2 == 0 && !CMUX && L == 4);
}
template <class F, int L, int BG, bool CMUX, int FORM = 0>
__global__ __launch_bounds__(F::THREADS, 2) void external_product_kernel(const d2 *__restrict__ bkrow0, const d2 *__restrict__ tw,
                                                                       const uint64_t *__restrict__ in, uint64_t *out, int Bg_bit_rt, int count,
                                                                       size_t key_stride = 0, size_t in_stride = 2 * F::N,
                                                                       const uint64_t *in0 = nullptr, d2 *__restrict__ out_dft = nullptr) {   // out may alias in0 (CMUX in place): neither is __restrict__
  constexpr int N = F::N, M = F::M, T = F::THREADS;
  using D = Digits<L, BG>;
  // rounding without the reduction mod 1 where the gadget bounds the sums (see pbs_kernel)
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);
  __shared__ __attribute__((aligned(16))) d2 xch[F::XCH_SLOTS];
  const int t = threadIdx.x;
  const int Bg_bit = BG > 0 ? BG : Bg_bit_rt;
  F fft;
  fft_setup(fft, tw, t);
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  // rows two at a time + the software-pipelined unit loop: transforms with the pass twiddles in LDS (Fft2048L: the registers that makes free are what
  // both need); the launcher picks that type for the gadgets where the build has no scratch
  constexpr bool kPairs = F::kForward2 && F::kLtw && L % 2 == 0;
  constexpr bool kPipe = FORM == 2 || (FORM == 0 && ep_pipelined_by_default<F, L, CMUX>());

  // Software pipeline over the team's units: the 8 KiB of a component are requested one phase before they are needed and wait in registers as raw
  // words (32 VGPRs) only until they arrive, then as packed digit words (16) -- so a team always has half a ciphertext in flight from HBM.
  uint64_t raw_lo[8], raw_hi[8];
  auto request = [&](size_t u, int q) {
    // ciphertexts stream through once: non-temporal loads and stores (EP_NT) keep them from displacing the key entry and the twiddles in the caches
    const uint64_t *ct = in + u * in_stride + (size_t)q * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      raw_lo[m] = EP_NT_LOAD(&ct[m * T + t]);
      raw_hi[m] = EP_NT_LOAD(&ct[M + m * T + t]);
    }
    if constexpr (CMUX) {
      const uint64_t *c0 = in0 + u * 2 * N + (size_t)q * N;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        raw_lo[m] -= c0[m * T + t];
        raw_hi[m] -= c0[M + m * T + t];
      }
    }
  };
  typename D::word_t w_lo[8], w_hi[8];
  uint32_t ext[8];
  auto pack = [&]() {
#pragma unroll
    for (int m = 0; m < 8; m++) D::pack(w_lo[m], w_hi[m], ext[m], raw_lo[m] + off, raw_hi[m] + off);
  };
  if constexpr (!kPipe) {
    // PLAIN loop: component loop rolled, each component requested where it is used.  What every ring of two or four wavefronts per team takes, with one
    // exception (ep_pipelined_by_default): the pipelined form below is not faster on these rings in general (SET_2: 0.249-0.292 vs 0.237-0.243 ms: the teams are
    // issue-bound, and the first key-row wait of a unit waits for the older ciphertext loads too).
    // History of a bug that lived here (rounds 3 - 5; experiments/README.md "Round 5"): with the pipelined loop, the l = 1 builds of N = 2048 / 4096 gave 1 - 2 % wrong
    // units per launch.  Cause, found in round 5 by patching one failing listing at assembly level and then reading the DFT-domain results slot by slot: the
    // COMPILER had sunk the last LDS reads of forward_head (layout D) and the pass behind them from in front of forward_tail's workgroup barrier into the block
    // behind the `if (next unit) request(...)` branch that follows it -- LLVM's machine sinking does not count S_BARRIER or the fences around it as a store -- so the
    // other wavefront of the team, released by the barrier, could write the next transform's first exchange over data this one had not read yet.  It showed only
    // where a conditional branch follows a barrier closely (this loop form), only on teams with a real barrier (two or four wavefronts), only when the two wavefronts
    // drift apart by more than a row's products (memory contention: several workgroups per CU) -- and never had anything to do with the register spills round 3
    // blamed.  Fixed at the root: every workgroup barrier of this library goes through workgroup_sync() (negacyclic_fft.h), which pins memory accesses to their
    // side of the barrier; tests/test_host_and_abi.py checks the generated code for the pattern, tests/test_gpu_parity.py runs the once-failing builds.
    for (size_t u = blockIdx.x; u < (size_t)count; u += gridDim.x) {
      const d2 *__restrict__ bkrow = bkrow0 + u * key_stride;
      double o_re[2][8], o_im[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
#pragma unroll 1
      for (int q = 0; q < 2; q++) {
        request(u, q);
        pack();
        if constexpr (kPairs) cmux_rows2<F, L, BG>(w_lo, w_hi, ext, q, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
        else cmux_rows<F, L, BG>(w_lo, w_hi, ext, q, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
      }
      if (out_dft) {
        d2 *dd = out_dft + u * 2 * M;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int m = 0; m < 8; m++) dd[c * M + m * T + t] = d2{o_re[c][m], o_im[c][m]};
        continue;
      }
      fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
      uint64_t *dst = out + u * 2 * N;
      const uint64_t *c0 = CMUX ? in0 + u * 2 * N : nullptr;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const uint64_t s_lo = CMUX ? c0[c * N + m * T + t] : 0, s_hi = CMUX ? c0[c * N + m * T + t + M] : 0;
          EP_NT_STORE(add_rounded<kReduce>(s_lo, o_re[c][m], scale), &dst[c * N + m * T + t]);
          EP_NT_STORE(add_rounded<kReduce>(s_hi, o_im[c][m], scale), &dst[c * N + m * T + t + M]);
        }
    }
    return;
  }
  if ((size_t)blockIdx.x < (size_t)count) request(blockIdx.x, 0);
  for (size_t u = blockIdx.x; u < (size_t)count; u += gridDim.x) {
    const d2 *__restrict__ bkrow = bkrow0 + u * key_stride;
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
    pack();              // component a (requested during the previous unit)
    request(u, 1);       // component b: in flight under the rows of component a
    if constexpr (kPairs) {
      cmux_rows2<F, L, BG>(w_lo, w_hi, ext, 0, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
    } else {
      cmux_rows<F, L, BG>(w_lo, w_hi, ext, 0, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
    }
    pack();
    if (u + gridDim.x < (size_t)count) request(u + gridDim.x, 0);   // the next unit's component a: under the rows of b, the inverse pair and the stores
    if constexpr (kPairs) cmux_rows2<F, L, BG>(w_lo, w_hi, ext, 1, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
    else cmux_rows<F, L, BG>(w_lo, w_hi, ext, 1, o_re, o_im, xch, fft, bkrow, Bg_bit, t);
    if (out_dft) {
      d2 *dd = out_dft + u * 2 * M;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) dd[c * M + m * T + t] = d2{o_re[c][m], o_im[c][m]};
      continue;
    }
    fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
    uint64_t *dst = out + u * 2 * N;
    const uint64_t *c0 = CMUX ? in0 + u * 2 * N : nullptr;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) {   // (out may alias in0: each lane reads its words of in0 before it writes them)
        const uint64_t s_lo = CMUX ? c0[c * N + m * T + t] : 0, s_hi = CMUX ? c0[c * N + m * T + t + M] : 0;
        EP_NT_STORE(add_rounded<kReduce>(s_lo, o_re[c][m], scale), &dst[c * N + m * T + t]);
        EP_NT_STORE(add_rounded<kReduce>(s_hi, o_im[c][m], scale), &dst[c * N + m * T + t + M]);
      }
  }
}

// The same at N = 1024 when the whole batch shares ONE key entry (key_stride = 0): eight teams (wavefronts) per workgroup, one workgroup per CU, and
// the 2 l rows of the entry (64 KiB at l = 2) staged ONCE into LDS next to the teams' transpose buffers (8 x 9 KiB) -- the key rows then cost
// ds_read_b128 instead of 64 KiB of L2 traffic per unit behind the HBM loads in the wavefront's in-order memory queue.  l <= 2.
// (More teams per CU, half-size transpose buffers and other request schedules were measured and dropped in round 6: experiments/README.md, r06_ldskey_teams_halfbuffer_ahead.patch.)
template <int L, int BG, bool CMUX>
__global__ __launch_bounds__(512, 2) void external_product_ldskey_kernel(const d2 *__restrict__ bkrow0, const d2 *__restrict__ tw, const uint64_t *__restrict__ in,
                                                                       uint64_t *out, int Bg_bit_rt, int count, size_t in_stride,
                                                                       const uint64_t *in0, d2 *__restrict__ out_dft) {   // out may alias in0: no __restrict__
  using F = Fft1024;
  constexpr int N = F::N, M = F::M, T = 64, TEAMS = 8;
  using D = Digits<L, BG>;
  constexpr bool kReduce = !(BG > 0 && kCeilLog2<2 * L>::value + (F::LOGM + 1) + BG - 1 + 63 < 83);
  __shared__ __attribute__((aligned(16))) d2 key[2 * L * 2 * M];
  __shared__ __attribute__((aligned(16))) d2 xch_all[TEAMS][F::XCH_SLOTS];
  const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
  d2 *xch = xch_all[w];
  for (int i = threadIdx.x; i < 2 * L * 2 * M; i += 64 * TEAMS) key[i] = bkrow0[i];
  const int Bg_bit = BG > 0 ? BG : Bg_bit_rt;
  F fft;
  fft.init(tw, t);
  uint64_t off = 1ull << (63 - L * Bg_bit);
#pragma unroll
  for (int i = 0; i < L; i++) off += 1ull << (63 - i * Bg_bit);
  const RoundCtx scale(0x1p-64 / (double)M);
  workgroup_sync();
  const size_t first = (size_t)blockIdx.x * TEAMS + w, stride = (size_t)gridDim.x * TEAMS;

  uint64_t raw_lo[8], raw_hi[8];
  auto request = [&](size_t u, int q) {
    const uint64_t *ct = in + u * in_stride + (size_t)q * N;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      raw_lo[m] = EP_NT_LOAD(&ct[m * T + t]);
      raw_hi[m] = EP_NT_LOAD(&ct[M + m * T + t]);
    }
    if constexpr (CMUX) {
      const uint64_t *c0 = in0 + u * 2 * N + (size_t)q * N;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        raw_lo[m] -= c0[m * T + t];
        raw_hi[m] -= c0[M + m * T + t];
      }
    }
  };
  typename D::word_t w_lo[8], w_hi[8];
  uint32_t ext[8];
  auto pack = [&]() {
#pragma unroll
    for (int m = 0; m < 8; m++) D::pack(w_lo[m], w_hi[m], ext[m], raw_lo[m] + off, raw_hi[m] + off);
  };
  // Compile-time gadgets whose rounding bit 2^(63 - L BG) lies in the HIGH dword (L BG <= 31), no CMUX operand: the digits of a word are a function of its high
  // dword alone -- off has no bits below 2^32, so no carry crosses -- and only the high dwords are requested: the same HBM lines (every line still moves once:
  // 8(d)'s bytes are unchanged), half the registers per component in flight.  They are spent on distance: BOTH components of the team's next unit are
  // requested while this one is computed (component a over the whole unit, component b over two thirds of it), where the 64-bit form has one third / two thirds.
  constexpr bool kHi = !CMUX && D::kPacked && L * BG <= 31;
  if constexpr (kHi) {
    const uint32_t off_hi = (uint32_t)(off >> 32);
    uint32_t h_lo[2][8], h_hi[2][8];
    auto request_hi = [&](size_t u, int q) {
      const uint32_t *ct = reinterpret_cast<const uint32_t *>(in + u * in_stride + (size_t)q * N) + 1;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        h_lo[q][m] = EP_NT_LOAD(&ct[2 * (m * T + t)]);
        h_hi[q][m] = EP_NT_LOAD(&ct[2 * (M + m * T + t)]);
      }
    };
    auto pack_hi = [&](int q) {
#pragma unroll
      for (int m = 0; m < 8; m++) D::pack(w_lo[m], w_hi[m], ext[m], (uint64_t)(h_lo[q][m] + off_hi) << 32, (uint64_t)(h_hi[q][m] + off_hi) << 32);
    };
    if (first < (size_t)count) {
      request_hi(first, 0);
      request_hi(first, 1);
    }
    for (size_t u = first; u < (size_t)count; u += stride) {
      double o_re[2][8], o_im[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
      const bool more = u + stride < (size_t)count;
      pack_hi(0);
      if (more) request_hi(u + stride, 0);
      cmux_rows<F, L, BG, true>(w_lo, w_hi, ext, 0, o_re, o_im, xch, fft, key, Bg_bit, t);
      pack_hi(1);
      if (more) request_hi(u + stride, 1);
      cmux_rows<F, L, BG, true>(w_lo, w_hi, ext, 1, o_re, o_im, xch, fft, key, Bg_bit, t);
      if (out_dft) {
        d2 *dd = out_dft + u * 2 * M;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int m = 0; m < 8; m++) dd[c * M + m * T + t] = d2{o_re[c][m], o_im[c][m]};
        continue;
      }
      fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
      uint64_t *dst = out + u * 2 * N;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) {
          EP_NT_STORE(add_rounded<kReduce>(0, o_re[c][m], scale), &dst[c * N + m * T + t]);
          EP_NT_STORE(add_rounded<kReduce>(0, o_im[c][m], scale), &dst[c * N + m * T + t + M]);
        }
    }
    return;
  }
  if (first < (size_t)count) request(first, 0);
  for (size_t u = first; u < (size_t)count; u += stride) {
    double o_re[2][8], o_im[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) { o_re[c][m] = 0.0; o_im[c][m] = 0.0; }
    pack();
    request(u, 1);
    cmux_rows<F, L, BG, true>(w_lo, w_hi, ext, 0, o_re, o_im, xch, fft, key, Bg_bit, t);
    pack();
    if (u + stride < (size_t)count) request(u + stride, 0);
    cmux_rows<F, L, BG, true>(w_lo, w_hi, ext, 1, o_re, o_im, xch, fft, key, Bg_bit, t);
    if (out_dft) {
      d2 *dd = out_dft + u * 2 * M;
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 8; m++) dd[c * M + m * T + t] = d2{o_re[c][m], o_im[c][m]};
      continue;
    }
    fft.inverse2(o_re[0], o_im[0], o_re[1], o_im[1], xch, t);
    uint64_t *dst = out + u * 2 * N;
    const uint64_t *c0 = CMUX ? in0 + u * 2 * N : nullptr;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const uint64_t s_lo = CMUX ? c0[c * N + m * T + t] : 0, s_hi = CMUX ? c0[c * N + m * T + t + M] : 0;
        EP_NT_STORE(add_rounded<kReduce>(s_lo, o_re[c][m], scale), &dst[c * N + m * T + t]);
        EP_NT_STORE(add_rounded<kReduce>(s_hi, o_im[c][m], scale), &dst[c * N + m * T + t + M]);
      }
  }
}

// polynomial_mul_DFT / polynomial_mul_addto_DFT on slot-ordered arrays [src/polynomial.c:379-426]
__global__ void dft_mul_kernel(d2 *out, const d2 *a, const d2 *b, size_t count,   // out may be a or b
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
