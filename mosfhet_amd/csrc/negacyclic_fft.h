// negacyclic_fft.h -- wave-resident negacyclic double-precision transform for gfx950 (CDNA4).
//
// Replaces the reference's FFT back-ends (src/fft/spqlios/*, src/fft/ffnt/*) behind
// polynomial_torus_to_DFT / polynomial_DFT_to_torus (src/polynomial.c:359-375).
//
// One real polynomial of degree < N is folded to M = N/2 complex points z_j = p_j + i p_{j+M} and
// evaluated at the M roots of y^M = i.  The twist of the textbook "twist + FFT" formulation is folded
// into the butterflies: level `lev` splits  z mod (y^L - c)  into  z mod (y^(L/2) -+ s), s = sqrt(c),
// with butterfly (a, b) -> (a + s b, a - s b).  A node's twiddle depends only on the node, the second
// child's twiddle is i times the first's (free), and the first three levels have lane-independent
// twiddles (scalar registers).  Output order is whatever the recursion leaves ("slot order"); the
// bootstrap key is transformed by the same code, so no bit reversal is ever done.
//
// Mapping for N = 1024 (M = 512): one 64-lane wavefront owns one transform, 8 points per lane
// (16 VGPR pairs).  Index j has 9 bits; three register-resident passes of three radix-2 levels each:
//     pass A: register = j[8:6], lane = j[5:0]                 levels 0-2, twiddles in SGPRs
//     pass B: register = j[5:3], lane = (j[8:6], j[2:0])       levels 3-5, 4 twiddles per lane
//     pass C: register = j[2:0], lane = j[8:3]                 levels 6-8, 4 twiddles per lane
// with two LDS transposes (ds_write_b128 / ds_read_b128, XOR-swizzled so both sides are
// bank-conflict free under the gfx950 lane-group rules; tools/lds_conflicts.py checks them).
// The inverse runs the passes backwards with (u, v) -> (u + v, (u - v) conj(s)).
//
// The floating-point operation order is FIXED and identical to oracle/oracle_fft.c, so results are
// bit-identical to the CPU oracle.  Build with -ffp-contract=off: every fma is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mosfhet {

typedef double __attribute__((ext_vector_type(2))) d2;

// ---- wave-level LDS ordering: one wavefront owns its LDS region, DS ops of a wave execute in order,
// so only the compiler has to be kept from reordering across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- butterflies (operation order = oracle_fft.c) ----
// forward, twiddle s = (sr, si):  a' = a + s b ; b' = 2a - a'
__device__ __forceinline__ void bf_fwd(double &ar, double &ai, double &br, double &bi, double sr, double si) {
  const double xr = __builtin_fma(-si, bi, __builtin_fma(sr, br, ar));
  const double xi = __builtin_fma(si, br, __builtin_fma(sr, bi, ai));
  br = __builtin_fma(2.0, ar, -xr);
  bi = __builtin_fma(2.0, ai, -xi);
  ar = xr;
  ai = xi;
}
// forward with twiddle i*w (second child of a parent): s = (-wi, wr)
__device__ __forceinline__ void bf_fwd_i(double &ar, double &ai, double &br, double &bi, double wr, double wi) {
  bf_fwd(ar, ai, br, bi, -wi, wr);
}
// inverse: a' = u + v ; b' = (u - v) conj(s)
__device__ __forceinline__ void bf_inv(double &ur, double &ui, double &vr, double &vi, double sr, double si) {
  const double dr = ur - vr, di = ui - vi;
  ur = ur + vr;
  ui = ui + vi;
  vr = __builtin_fma(sr, dr, si * di);
  vi = __builtin_fma(sr, di, -(si * dr));
}
__device__ __forceinline__ void bf_inv_i(double &ur, double &ui, double &vr, double &vi, double wr, double wi) {
  bf_inv(ur, ui, vr, vi, -wi, wr);
}

// Twiddles of one three-level pass for this lane: w0 = level a (node nu), w1 = level a+1 (node 2nu),
// w2a / w2b = level a+2 (nodes 4nu, 4nu+2); the odd nodes are i times their even sibling.
struct PassTw {
  double w0r, w0i, w1r, w1i, w2ar, w2ai, w2br, w2bi;
};

// three radix-2 levels on 8 register-resident points; register index bit 2 is the highest index bit.
__device__ __forceinline__ void pass_fwd(double (&re)[8], double (&im)[8], const PassTw &w) {
#pragma unroll
  for (int m = 0; m < 4; m++) bf_fwd(re[m], im[m], re[m + 4], im[m + 4], w.w0r, w.w0i);
  bf_fwd(re[0], im[0], re[2], im[2], w.w1r, w.w1i);
  bf_fwd(re[1], im[1], re[3], im[3], w.w1r, w.w1i);
  bf_fwd_i(re[4], im[4], re[6], im[6], w.w1r, w.w1i);
  bf_fwd_i(re[5], im[5], re[7], im[7], w.w1r, w.w1i);
  bf_fwd(re[0], im[0], re[1], im[1], w.w2ar, w.w2ai);
  bf_fwd_i(re[2], im[2], re[3], im[3], w.w2ar, w.w2ai);
  bf_fwd(re[4], im[4], re[5], im[5], w.w2br, w.w2bi);
  bf_fwd_i(re[6], im[6], re[7], im[7], w.w2br, w.w2bi);
}

__device__ __forceinline__ void pass_inv(double (&re)[8], double (&im)[8], const PassTw &w) {
  bf_inv(re[0], im[0], re[1], im[1], w.w2ar, w.w2ai);
  bf_inv_i(re[2], im[2], re[3], im[3], w.w2ar, w.w2ai);
  bf_inv(re[4], im[4], re[5], im[5], w.w2br, w.w2bi);
  bf_inv_i(re[6], im[6], re[7], im[7], w.w2br, w.w2bi);
  bf_inv(re[0], im[0], re[2], im[2], w.w1r, w.w1i);
  bf_inv(re[1], im[1], re[3], im[3], w.w1r, w.w1i);
  bf_inv_i(re[4], im[4], re[6], im[6], w.w1r, w.w1i);
  bf_inv_i(re[5], im[5], re[7], im[7], w.w1r, w.w1i);
#pragma unroll
  for (int m = 0; m < 4; m++) bf_inv(re[m], im[m], re[m + 4], im[m + 4], w.w0r, w.w0i);
}

// Twiddle table: (re, im) of node (2^lev - 1 + nu), M - 1 entries (host: make_twiddles()).
__device__ __forceinline__ PassTw load_pass_tw(const d2 *__restrict__ tw, int lev, int nu) {
  const d2 a = tw[(1 << lev) - 1 + nu];
  const d2 b = tw[(2 << lev) - 1 + 2 * nu];
  const d2 c = tw[(4 << lev) - 1 + 4 * nu];
  const d2 d = tw[(4 << lev) - 1 + 4 * nu + 2];
  PassTw w;
  w.w0r = a.x; w.w0i = a.y; w.w1r = b.x; w.w1i = b.y;
  w.w2ar = c.x; w.w2ai = c.y; w.w2br = d.x; w.w2bi = d.y;
  return w;
}

// ------------------------------------------------------------------------------------------------
// N = 1024: 64 lanes x 8 points.
// ------------------------------------------------------------------------------------------------
struct Fft1024 {
  static constexpr int N = 1024, M = 512, LOGM = 9, LANES = 64, P = 8;
  PassTw wa, wb, wc;  // wa is lane-uniform (lives in SGPRs), wb / wc are per lane

  __device__ __forceinline__ void init(const d2 *__restrict__ tw, int lane) {
    wa = load_pass_tw(tw, 0, 0);
    wb = load_pass_tw(tw, 3, lane >> 3);
    wc = load_pass_tw(tw, 6, lane);
  }

  // physical 16-byte slot of element j for each exchange (see tools/lds_conflicts.py)
  static __device__ __forceinline__ int slot_ab(int j) { return j ^ (((j >> 6) & 7) << 3); }
  static __device__ __forceinline__ int slot_bc(int j) { return j ^ ((j >> 4) & 7); }
  static __device__ __forceinline__ int slot_cb(int j) {
    return (j & 0x100) | (((j >> 6) & 1) << 7) | ((j & 7) << 4) | (((j >> 7) & 1) << 3) | (((j >> 3) & 7) ^ (j & 7));
  }
  // index of register m of this lane in each layout
  static __device__ __forceinline__ int idx_a(int lane, int m) { return (m << 6) | lane; }
  static __device__ __forceinline__ int idx_b(int lane, int m) { return ((lane >> 3) << 6) | (m << 3) | (lane & 7); }
  static __device__ __forceinline__ int idx_c(int lane, int m) { return (lane << 3) | m; }

  // forward: input in layout A, output in layout C ("slot order": slot = lane*8 + m)
  __device__ __forceinline__ void forward(double (&re)[8], double (&im)[8], d2 *xch, int lane) const {
    pass_fwd(re, im, wa);
#pragma unroll
    for (int m = 0; m < 8; m++) xch[slot_ab(idx_a(lane, m))] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = xch[slot_ab(idx_b(lane, m))]; re[m] = v.x; im[m] = v.y; }
    pass_fwd(re, im, wb);
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) xch[slot_bc(idx_b(lane, m))] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = xch[slot_bc(idx_c(lane, m))]; re[m] = v.x; im[m] = v.y; }
    pass_fwd(re, im, wc);
    wave_lds_sync();
  }

  // inverse: input in layout C, output in layout A, UNSCALED (caller multiplies by 1/M)
  __device__ __forceinline__ void inverse(double (&re)[8], double (&im)[8], d2 *xch, int lane) const {
    pass_inv(re, im, wc);
#pragma unroll
    for (int m = 0; m < 8; m++) xch[slot_cb(idx_c(lane, m))] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = xch[slot_cb(idx_b(lane, m))]; re[m] = v.x; im[m] = v.y; }
    pass_inv(re, im, wb);
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) xch[idx_b(lane, m)] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = xch[idx_a(lane, m)]; re[m] = v.x; im[m] = v.y; }
    pass_inv(re, im, wa);
    wave_lds_sync();
  }
};

// double -> Torus64, round to nearest, mod 2^64 (values reach ~2^84).  `scale` = 2^-64 / M.
// Same arithmetic as oracle_fft.c:round_mod_2_64 (the two exact power-of-two scalings are merged);
// semantics of the reference's AVX-512 path, fft_processor_spqlios.c:155-165.
__device__ __forceinline__ uint64_t round_mod_2_64(double v, double scale) {
  double f = v * scale;
  f = f - __builtin_rint(f);
  const double g = __builtin_rint(f * 0x1p64);
  double hi = __builtin_floor(g * 0x1p-32);
  const double lo = __builtin_fma(-hi, 0x1p32, g);
  if (hi < 0.0) hi += 0x1p32;
  return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

// (double)(int64_t)x, correctly rounded (matches the C cast used by the reference and the oracle)
__device__ __forceinline__ double torus_to_double(uint64_t x) { return (double)(int64_t)x; }

}  // namespace mosfhet
