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
// with two LDS transposes (ds_write_b128 / ds_read_b128 through padded layouts that are bank-conflict free on
// both sides under the gfx950 lane-group rules and addressable as lane base + immediate; tools/lds_conflicts.py).
// The inverse runs the passes backwards with (u, v) -> (u + v, (u - v) conj(s)).
//
// The floating-point operation order is FIXED and identical to oracle/oracle_fft.c, so results are
// bit-identical to the CPU oracle.  Build with -ffp-contract=off: every fma is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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

// the first one / two levels of a three-level pass alone (N = 2048 / 4096 with the short pass in front of the last full one, see Fft2048T)
__device__ __forceinline__ void pass_fwd_top1(double (&re)[8], double (&im)[8], const PassTw &w) {
#pragma unroll
  for (int m = 0; m < 4; m++) bf_fwd(re[m], im[m], re[m + 4], im[m + 4], w.w0r, w.w0i);
}
__device__ __forceinline__ void pass_inv_top1(double (&re)[8], double (&im)[8], const PassTw &w) {
#pragma unroll
  for (int m = 0; m < 4; m++) bf_inv(re[m], im[m], re[m + 4], im[m + 4], w.w0r, w.w0i);
}
__device__ __forceinline__ void pass_fwd_top2(double (&re)[8], double (&im)[8], const PassTw &w) {
  pass_fwd_top1(re, im, w);
  bf_fwd(re[0], im[0], re[2], im[2], w.w1r, w.w1i);
  bf_fwd(re[1], im[1], re[3], im[3], w.w1r, w.w1i);
  bf_fwd_i(re[4], im[4], re[6], im[6], w.w1r, w.w1i);
  bf_fwd_i(re[5], im[5], re[7], im[7], w.w1r, w.w1i);
}
__device__ __forceinline__ void pass_inv_top2(double (&re)[8], double (&im)[8], const PassTw &w) {
  bf_inv(re[0], im[0], re[2], im[2], w.w1r, w.w1i);
  bf_inv(re[1], im[1], re[3], im[3], w.w1r, w.w1i);
  bf_inv_i(re[4], im[4], re[6], im[6], w.w1r, w.w1i);
  bf_inv_i(re[5], im[5], re[7], im[7], w.w1r, w.w1i);
  pass_inv_top1(re, im, w);
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
  static constexpr int N = 1024, M = 512, LOGM = 9, THREADS = 64, P = 8;
  static constexpr bool kLtw = false, kForward2 = false;
  static __device__ __forceinline__ void sync() { wave_lds_sync(); }
  PassTw wa, wb, wc;  // wa is lane-uniform (lives in SGPRs), wb / wc are per lane

  __device__ __forceinline__ void init(const d2 *__restrict__ tw, int lane) {
    wa = load_pass_tw(tw, 0, 0);
    wb = load_pass_tw(tw, 3, lane >> 3);
    wc = load_pass_tw(tw, 6, lane);
  }

  // LDS transposes.  Physical 16-byte slot of element j:  A<->B exchanges use f1(j) = j + 8 (j >> 6),
  // B<->C exchanges use f2(j) = j + (j >> 3)  (both 576 slots = 9 KiB).  Both are bank-conflict free for the
  // ds_write_b128 and the ds_read_b128 side under gfx950's lane groups AND additive in the register index, so
  // every access is `lane base + immediate offset` (tools/lds_conflicts.py verifies both properties):
  //   layout A (reg m = j[8:6], lane = j[5:0]):            f1 = lane + 72 m
  //   layout B (reg m = j[5:3], lane = (h = j[8:6], lo)):  f1 = 72 h + lo + 8 m ;  f2 = 72 h + lo + 9 m
  //   layout C (reg m = j[2:0], lane = j[8:3]):            f2 = 9 lane + m
  static constexpr int XCH_SLOTS = 576;
  static __device__ __forceinline__ void transform_barriers_only() {}   // the transforms of one wavefront have no workgroup barriers

  // forward: input in layout A, output in layout C ("slot order": slot = lane*8 + m).
  // Split in two (head = passes A, B and both transposes; tail = pass C) so the caller can issue the bootstrap-key
  // loads in between.
  __device__ __forceinline__ void forward_head(double (&re)[8], double (&im)[8], d2 *xch, int lane) const {
    d2 *pa = xch + lane, *pb = xch + 72 * (lane >> 3) + (lane & 7), *pc = xch + 9 * lane;
    pass_fwd(re, im, wa);
#pragma unroll
    for (int m = 0; m < 8; m++) pa[72 * m] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb[8 * m]; re[m] = v.x; im[m] = v.y; }
    pass_fwd(re, im, wb);
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb[9 * m] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc[m]; re[m] = v.x; im[m] = v.y; }
  }
  __device__ __forceinline__ void forward_tail(double (&re)[8], double (&im)[8]) const {
    pass_fwd(re, im, wc);
    wave_lds_sync();
  }
  __device__ __forceinline__ void forward(double (&re)[8], double (&im)[8], d2 *xch, int lane) const {
    forward_head(re, im, xch, lane);
    forward_tail(re, im);
  }

  // inverse: input in layout C, output in layout A, UNSCALED (caller multiplies by 1/M)
  __device__ __forceinline__ void inverse(double (&re)[8], double (&im)[8], d2 *xch, int lane) const {
    d2 *pa = xch + lane, *pb = xch + 72 * (lane >> 3) + (lane & 7), *pc = xch + 9 * lane;
    pass_inv(re, im, wc);
#pragma unroll
    for (int m = 0; m < 8; m++) pc[m] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb[9 * m]; re[m] = v.x; im[m] = v.y; }
    pass_inv(re, im, wb);
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb[8 * m] = d2{re[m], im[m]};
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[72 * m]; re[m] = v.x; im[m] = v.y; }
    pass_inv(re, im, wa);
    wave_lds_sync();
  }

  // Two independent inverse transforms (the two components of an external product) software-pipelined through the ONE
  // transpose buffer: DS operations of a wavefront execute in order, so Y's writes may be queued right behind X's reads of
  // the same slots, and each transform's register pass runs while the other's transpose is in flight.
  __device__ __forceinline__ void inverse2(double (&xr)[8], double (&xi)[8], double (&yr)[8], double (&yi)[8], d2 *xch, int lane) const {
    d2 *pa = xch + lane, *pb = xch + 72 * (lane >> 3) + (lane & 7), *pc = xch + 9 * lane;
    pass_inv(xr, xi, wc);
#pragma unroll
    for (int m = 0; m < 8; m++) pc[m] = d2{xr[m], xi[m]};
    wave_lds_sync();
    pass_inv(yr, yi, wc);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb[9 * m]; xr[m] = v.x; xi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pc[m] = d2{yr[m], yi[m]};
    wave_lds_sync();
    pass_inv(xr, xi, wb);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb[9 * m]; yr[m] = v.x; yi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb[8 * m] = d2{xr[m], xi[m]};
    wave_lds_sync();
    pass_inv(yr, yi, wb);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[72 * m]; xr[m] = v.x; xi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb[8 * m] = d2{yr[m], yi[m]};
    wave_lds_sync();
    pass_inv(xr, xi, wa);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[72 * m]; yr[m] = v.x; yi[m] = v.y; }
    pass_inv(yr, yi, wa);
    wave_lds_sync();
  }
};

// ------------------------------------------------------------------------------------------------
// N = 2048 (M = 1024): two wavefronts (128 threads) x 8 points share one transform.
// Index j has 10 bits; passes of 3, 3, 3 and 1 radix-2 levels:
//     pass A: register = j[9:7], thread = j[6:0]                       levels 0-2, twiddles in SGPRs
//     pass B: register = j[6:4], thread = (h = j[9:7], j[3:0])         levels 3-5
//     pass C: register = j[3:1], thread = (u = j[9:4], e = j[0]) = 2u+e levels 6-8
//     pass D: register = j[2:0], thread = j[9:3]                       level 9 (pairs m, m+1)
// Three LDS transposes per transform, each through its own padded layout (tools/lds_layout_search.py: conflict
// free for ds_write_b128 and ds_read_b128 in both directions, every access = thread base + immediate):
//     A<->B: slot = j                      B<->C: slot = j + 2 (j >> 4)         C<->D: slot = j + (j >> 3)
// Only the A<->B exchange crosses the two wavefronts (workgroup barriers); B<->C and C<->D are wave-local (wave-level ordering).
// Slot order of the DFT domain = layout D: device index m * 128 + thread.
// ------------------------------------------------------------------------------------------------
// Workgroup barrier of this library.  NEVER call __syncthreads() directly: LLVM's machine-level sinking (ROCm 7.2.0, clang 22) moves LDS loads that stand IN
// FRONT of a barrier into a successor block BEHIND it when their only uses are there -- neither S_BARRIER nor the fences around it count as a store for that pass --
// so another wavefront, released by the barrier, can overwrite what this one has not read yet.  That is the root cause of the "wrong units" of the pipelined
// external-product loop on two-wavefront teams (rounds 3 - 5; experiments/README.md "Round 5": the D-layout reads of forward_head sunk behind forward_tail's barrier
// into the block behind the conditional next-unit request).  An empty asm with a memory clobber on both sides reads and writes memory as far as every pass is
// concerned, which pins loads and stores to their side of the barrier; it emits nothing.
__device__ __forceinline__ void workgroup_sync() {
  asm volatile("" ::: "memory");
  __syncthreads();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void team_sync() { workgroup_sync(); }

// TAIL3 = false: levels grouped 3 + 3 + 3 + 1 over the four layouts (pass C = levels 6-8, pass D = level 9).
// TAIL3 = true:  3 + 3 + 1 + 3 (pass C = level 6, the top register bit of layout C; pass D = levels 7-9): the pass the caller overlaps with its key-row
// loads (forward_tail) is then a full three-level one, as at N = 1024.  Same layouts, same exchanges, same butterflies with the same twiddles in the
// same order per element -- bit-identical results (oracle_fft.c), same slot order.  Which one is faster depends on the registers the caller has left
// under the last pass: l = 1 bootstraps gain 2 % (SET_2) to 6 % (SET_3), the l = 4 kernel of lvl2 loses 5 % (experiments/README.md, round 2).
// LTW = true: the per-lane twiddles of passes B and C (32 VGPRs) live in a 4.5 KiB LDS table of the workgroup instead -- 8 sets of pass B (one per
// j[9:7]) and 64 of pass C (one per j[9:4]), 4 complex each -- and are read where a pass needs them (4 ds_read_b128).  For kernels that are short of
// registers and have the LDS to spare (same-box A/B, experiments/README.md round 4: the lvl2 external product -6 %; the fused bootstrap, whose LDS
// pipe is the second-busiest unit, +1 %).  Same values, same butterflies: bit-identical.  Kernels set such a transform up with fft_setup().
struct Fft2048TwRegs {
  static constexpr bool kLtw = false;
  PassTw wb, wc;
  __device__ __forceinline__ const PassTw &WB() const { return wb; }
  __device__ __forceinline__ const PassTw &WC() const { return wc; }
};
struct Fft2048TwLds {
  static constexpr bool kLtw = true;
  static constexpr int LTW_SLOTS = 4 * (8 + 64);
  // table layout: [4 twiddles][8 sets] of pass B, then [4][64] of pass C -- the sets of one twiddle side by side, so that the 16 lanes the LDS serves
  // together read 16-byte entries of DIFFERENT sets from different banks (set-major, [set][4], the 8 sets a service group touches collide two ways:
  // SQ_LDS_BANK_CONFLICT 5 % of the LDS cycles of the first build)
  const d2 *lb, *lc;
  template <int SETS>
  static __device__ __forceinline__ PassTw lds_tw(const d2 *q) {
    const d2 a = q[0], b = q[SETS], c = q[2 * SETS], d = q[3 * SETS];
    PassTw w;
    w.w0r = a.x; w.w0i = a.y; w.w1r = b.x; w.w1i = b.y; w.w2ar = c.x; w.w2ai = c.y; w.w2br = d.x; w.w2bi = d.y;
    return w;
  }
  __device__ __forceinline__ PassTw WB() const { return lds_tw<8>(lb); }
  __device__ __forceinline__ PassTw WC() const { return lds_tw<64>(lc); }
  // fills the table (threads 0..71 of the workgroup, one set each) and points this thread at its sets; the caller synchronises the workgroup
  __device__ __forceinline__ void init_ltw(const d2 *__restrict__ tw, int t, d2 *tab) {
    if (t < 72) {
      const int lev = t < 8 ? 3 : 6, nu = t < 8 ? t : t - 8, sets = t < 8 ? 8 : 64;
      d2 *dst = (t < 8 ? tab : tab + 32) + nu;
      dst[0] = tw[(1 << lev) - 1 + nu];
      dst[sets] = tw[(2 << lev) - 1 + 2 * nu];
      dst[2 * sets] = tw[(4 << lev) - 1 + 4 * nu];
      dst[3 * sets] = tw[(4 << lev) - 1 + 4 * nu + 2];
    }
    lb = tab + (t >> 4);
    lc = tab + 32 + (t >> 1);
  }
};

template <bool TAIL3, bool LTW = false>
struct Fft2048T : std::conditional<LTW, Fft2048TwLds, Fft2048TwRegs>::type {
  static_assert(!(TAIL3 && LTW), "the LDS table holds full three-level sets");
  static constexpr int N = 2048, M = 1024, LOGM = 10, THREADS = 128, P = 8;
  static constexpr int XCH_SLOTS = 1152;
  static constexpr bool kForward2 = true;
  static __device__ __forceinline__ void sync() { team_sync(); }
  PassTw wa;
  PassTw wd;   // TAIL3: levels 7-9 of layout D (and only w0 of wc is used); else only w2a / w2b: level 9, nodes 4t and 4t+2 (4t+1, 4t+3 are i times those)

  __device__ __forceinline__ void init(const d2 *__restrict__ tw, int t) {
    wa = load_pass_tw(tw, 0, 0);
    if constexpr (!LTW) this->wb = load_pass_tw(tw, 3, t >> 4);
    if constexpr (TAIL3) {
      const d2 c = tw[(1 << 6) - 1 + (t >> 1)];
      this->wc.w0r = c.x; this->wc.w0i = c.y;
      this->wc.w1r = this->wc.w1i = this->wc.w2ar = this->wc.w2ai = this->wc.w2br = this->wc.w2bi = 0.0;
      wd = load_pass_tw(tw, 7, t);
    } else {
      if constexpr (!LTW) this->wc = load_pass_tw(tw, 6, t >> 1);
      const d2 a = tw[(1 << 9) - 1 + 4 * t], b = tw[(1 << 9) - 1 + 4 * t + 2];
      wd.w0r = wd.w0i = wd.w1r = wd.w1i = 0.0;
      wd.w2ar = a.x; wd.w2ai = a.y; wd.w2br = b.x; wd.w2bi = b.y;
    }
  }

  // thread bases (in 16-byte slots) of the four layouts under the three slot maps
  static __device__ __forceinline__ int base_a(int t) { return t; }                                  // + 128 m
  static __device__ __forceinline__ int base_b1(int t) { return 128 * (t >> 4) + (t & 15); }         // + 16 m      (slot = j)
  static __device__ __forceinline__ int base_b2(int t) { return 144 * (t >> 4) + (t & 15); }         // + 18 m      (j + 2 (j>>4))
  static __device__ __forceinline__ int base_c2(int t) { return 18 * (t >> 1) + (t & 1); }           // + 2 m       (j + 2 (j>>4))
  static __device__ __forceinline__ int base_c3(int t) { return 18 * (t >> 1) + (t & 1); }           // + off_c3[m] (j + (j>>3))
  static __device__ __forceinline__ int base_d3(int t) { return 9 * t; }                             // + m         (j + (j>>3))
  // layout C under slot = j + (j >> 3): j = 16u + 2m + e  ->  18u + e + 2m + (m >> 2)
  static __device__ __forceinline__ constexpr int off_c3(int m) { return 2 * m + (m >> 2); }

  __device__ __forceinline__ void pass_c_fwd(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) pass_fwd_top1(re, im, this->WC());
    else pass_fwd(re, im, this->WC());
  }
  __device__ __forceinline__ void pass_c_inv(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) pass_inv_top1(re, im, this->WC());
    else pass_inv(re, im, this->WC());
  }
  __device__ __forceinline__ void pass_d_fwd(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) {
      pass_fwd(re, im, wd);
    } else {
      bf_fwd(re[0], im[0], re[1], im[1], wd.w2ar, wd.w2ai);
      bf_fwd_i(re[2], im[2], re[3], im[3], wd.w2ar, wd.w2ai);
      bf_fwd(re[4], im[4], re[5], im[5], wd.w2br, wd.w2bi);
      bf_fwd_i(re[6], im[6], re[7], im[7], wd.w2br, wd.w2bi);
    }
  }
  __device__ __forceinline__ void pass_d_inv(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) {
      pass_inv(re, im, wd);
    } else {
      bf_inv(re[0], im[0], re[1], im[1], wd.w2ar, wd.w2ai);
      bf_inv_i(re[2], im[2], re[3], im[3], wd.w2ar, wd.w2ai);
      bf_inv(re[4], im[4], re[5], im[5], wd.w2br, wd.w2bi);
      bf_inv_i(re[6], im[6], re[7], im[7], wd.w2br, wd.w2bi);
    }
  }

  // forward: input in layout A, output in layout D.  forward_head leaves the data ready for pass D so the caller
  // can issue the key loads before it.  Entry requires that nobody still reads xch (barrier at the previous exit).
  __device__ __forceinline__ void forward_head(double (&re)[8], double (&im)[8], d2 *xch, int t) const {
    pass_fwd(re, im, wa);
    {
      d2 *w = xch + base_a(t), *r = xch + base_b1(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[128 * m] = d2{re[m], im[m]};
      team_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[16 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_fwd(re, im, this->WB());
    team_sync();
    // From here on the wavefront index is j[9] in every layout (B: t = 16 j[9:7] + j[3:0], C: t = 2 j[9:4] + j[0], D: t = j[9:3]) and
    // both slot maps send j < 512 to slots < 576 and j >= 512 to slots >= 576: the B<->C and C<->D exchanges stay inside a wavefront
    // and its own half of the buffer, so wave-level ordering is enough (3 workgroup barriers per transform instead of 6).
    {
      d2 *w = xch + base_b2(t), *r = xch + base_c2(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[18 * m] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[2 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_c_fwd(re, im);
    wave_lds_sync();
    {
      d2 *w = xch + base_c3(t), *r = xch + base_d3(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[off_c3(m)] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[m]; re[m] = v.x; im[m] = v.y; }
    }
  }
  __device__ __forceinline__ void forward_tail(double (&re)[8], double (&im)[8]) const {
    pass_d_fwd(re, im);
    team_sync();
  }
  __device__ __forceinline__ void forward(double (&re)[8], double (&im)[8], d2 *xch, int t) const {
    forward_head(re, im, xch, t);
    forward_tail(re, im);
  }

  // Two forward transforms (two consecutive TRGSW rows' digit polynomials) software-pipelined through the ONE buffer, the mirror image of inverse2:
  // every register pass of one transform runs while the other's exchange is in flight, and the cross-wavefront A -> B exchanges cost 4 workgroup
  // barriers for the pair (+ 1 in forward2_done) instead of 6.  Same butterflies on the same elements in the same order as forward(): bit-identical.
  // On return x is in layout D in front of its last pass (as after forward_head) and y is PARKED in the buffer in layout C (slot map j + (j >> 3)):
  // its registers are free until forward2_fetch reads it back in layout D.  The scheduling barriers keep the compiler from interleaving the stages
  // any further (it then holds both transforms' temporaries at once and spills).
  __device__ __forceinline__ void forward2_head(double (&xr)[8], double (&xi)[8], double (&yr)[8], double (&yi)[8], d2 *xch, int t) const {
    d2 *pa = xch + base_a(t), *pb1 = xch + base_b1(t), *pb2 = xch + base_b2(t), *pc2 = xch + base_c2(t), *pc3 = xch + base_c3(t), *pd = xch + base_d3(t);
    pass_fwd(xr, xi, wa);
#pragma unroll
    for (int m = 0; m < 8; m++) pa[128 * m] = d2{xr[m], xi[m]};
    team_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb1[16 * m]; xr[m] = v.x; xi[m] = v.y; }
    pass_fwd(yr, yi, wa);
    team_sync(); __builtin_amdgcn_sched_barrier(0);   // both wavefronts have read x
#pragma unroll
    for (int m = 0; m < 8; m++) pa[128 * m] = d2{yr[m], yi[m]};
    team_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb1[16 * m]; yr[m] = v.x; yi[m] = v.y; }
    pass_fwd(xr, xi, this->WB());
    team_sync(); __builtin_amdgcn_sched_barrier(0);   // both wavefronts have read y; from here on every exchange stays inside a wavefront and its half of the buffer (see forward_head)
#pragma unroll
    for (int m = 0; m < 8; m++) pb2[18 * m] = d2{xr[m], xi[m]};
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc2[2 * m]; xr[m] = v.x; xi[m] = v.y; }
    pass_fwd(yr, yi, this->WB());
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) pb2[18 * m] = d2{yr[m], yi[m]};
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc2[2 * m]; yr[m] = v.x; yi[m] = v.y; }
    pass_c_fwd(xr, xi);
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) pc3[off_c3(m)] = d2{xr[m], xi[m]};
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pd[m]; xr[m] = v.x; xi[m] = v.y; }
    pass_c_fwd(yr, yi);
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; m++) pc3[off_c3(m)] = d2{yr[m], yi[m]};
    wave_lds_sync(); __builtin_amdgcn_sched_barrier(0);
  }
  // y back from the buffer, in layout D in front of its last pass.  The caller runs pass_d_fwd on it and then forward2_done().
  __device__ __forceinline__ void forward2_fetch(double (&yr)[8], double (&yi)[8], d2 *xch, int t) const {
    const d2 *pd = xch + base_d3(t);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pd[m]; yr[m] = v.x; yi[m] = v.y; }
  }
  static __device__ __forceinline__ void forward2_done() { team_sync(); }   // nobody reads the buffer any more (entry condition of the next transform)

  // inverse: input in layout D, output in layout A, UNSCALED
  __device__ __forceinline__ void inverse(double (&re)[8], double (&im)[8], d2 *xch, int t) const {
    pass_d_inv(re, im);
    {  // D -> C and C -> B stay inside a wavefront (see forward_head): wave-level ordering
      d2 *w = xch + base_d3(t), *r = xch + base_c3(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[m] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[off_c3(m)]; re[m] = v.x; im[m] = v.y; }
    }
    pass_c_inv(re, im);
    wave_lds_sync();
    {
      d2 *w = xch + base_c2(t), *r = xch + base_b2(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[2 * m] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[18 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_inv(re, im, this->WB());
    team_sync();
    {
      d2 *w = xch + base_b1(t), *r = xch + base_a(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[16 * m] = d2{re[m], im[m]};
      team_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[128 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_inv(re, im, wa);
    team_sync();
  }

  // the workgroup barriers of one forward / inverse transform and nothing else: for teams of a multi-team workgroup that sit a transform out
  // (bootstrap_kernels.h: pbs_wide_team_kernel)
  static __device__ __forceinline__ void transform_barriers_only() {
    team_sync();
    team_sync();
    team_sync();
  }

  // two inverse transforms pipelined through the one buffer (cf. Fft1024::inverse2): wave-local exchanges rely on in-order DS
  // execution, the cross-wavefront B -> A exchange keeps its workgroup barriers (5 in all instead of 6) and every register pass
  // runs while the other transform's exchange is in flight.
  __device__ __forceinline__ void inverse2(double (&xr)[8], double (&xi)[8], double (&yr)[8], double (&yi)[8], d2 *xch, int t) const {
    d2 *pd = xch + base_d3(t), *pc3 = xch + base_c3(t), *pc2 = xch + base_c2(t), *pb2 = xch + base_b2(t), *pb1 = xch + base_b1(t), *pa = xch + base_a(t);
    pass_d_inv(xr, xi);
#pragma unroll
    for (int m = 0; m < 8; m++) pd[m] = d2{xr[m], xi[m]};
    wave_lds_sync();
    pass_d_inv(yr, yi);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc3[off_c3(m)]; xr[m] = v.x; xi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pd[m] = d2{yr[m], yi[m]};
    wave_lds_sync();
    pass_c_inv(xr, xi);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc3[off_c3(m)]; yr[m] = v.x; yi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pc2[2 * m] = d2{xr[m], xi[m]};
    wave_lds_sync();
    pass_c_inv(yr, yi);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb2[18 * m]; xr[m] = v.x; xi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pc2[2 * m] = d2{yr[m], yi[m]};
    wave_lds_sync();
    pass_inv(xr, xi, this->WB());
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb2[18 * m]; yr[m] = v.x; yi[m] = v.y; }
    team_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb1[16 * m] = d2{xr[m], xi[m]};
    team_sync();
    pass_inv(yr, yi, this->WB());
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[128 * m]; xr[m] = v.x; xi[m] = v.y; }
    team_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb1[16 * m] = d2{yr[m], yi[m]};
    team_sync();
    pass_inv(xr, xi, wa);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[128 * m]; yr[m] = v.x; yi[m] = v.y; }
    pass_inv(yr, yi, wa);
    team_sync();
  }
};
using Fft2048 = Fft2048T<false>;
using Fft2048W = Fft2048T<true>;
using Fft2048L = Fft2048T<false, true>;   // twiddles of passes B and C in LDS

// transform set-up of a kernel: twiddles into registers and, for a transform that keeps some in LDS, the workgroup's table
template <class F>
__device__ __forceinline__ void fft_setup(F &fft, const d2 *__restrict__ tw, int t) {
  fft.init(tw, t);
  if constexpr (F::kLtw) {
    __shared__ __attribute__((aligned(16))) d2 ltw_tab[F::LTW_SLOTS];
    fft.init_ltw(tw, t, ltw_tab);
    workgroup_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// N = 4096 (M = 2048; the reference's SET_3, test/tests.c:48): four wavefronts (256 threads) x 8 points.
// Index j has 11 bits; passes of 3, 3, 3 and 2 radix-2 levels:
//     pass A: register = j[10:8], thread = j[7:0]                         levels 0-2, twiddles in SGPRs
//     pass B: register = j[7:5],  thread = (j[10:8], j[4:0])              levels 3-5
//     pass C: register = j[4:2],  thread = (j[10:5], j[1:0])              levels 6-8
//     pass D: register = j[2:0],  thread = j[10:3]                        levels 9-10 (the last two stages of a pass)
// Slot maps (tools/lds_layout_search.py, conflict free for ds_write_b128 / ds_read_b128 in both directions, every access =
// thread base + immediate):   A<->B: slot = j      B<->C: slot = j + 4 (j >> 5)      C<->D: slot = j + (j >> 3)     (2304 slots)
// In layouts B, C, D the wavefront index is (j[10], j[9]) and both padded maps send each block of 512 indices to its own 576
// slots, so B<->C and C<->D stay inside a wavefront (wave-level ordering); only A<->B crosses wavefronts.
// Slot order of the DFT domain = layout D: device index m * 256 + thread.
// ------------------------------------------------------------------------------------------------
// TAIL3: levels grouped 3 + 3 + 2 + 3 instead of 3 + 3 + 3 + 2 (cf. Fft2048T)
template <bool TAIL3>
struct Fft4096T {
  static constexpr int N = 4096, M = 2048, LOGM = 11, THREADS = 256, P = 8;
  static constexpr bool kLtw = false, kForward2 = false;
  static constexpr int XCH_SLOTS = 2304;
  static __device__ __forceinline__ void sync() { team_sync(); }
  PassTw wa, wb, wc, wd;  // TAIL3: wc = levels 6-7 (w0, w1), wd = levels 8-10; else wc = levels 6-8, wd: only w1 (level 9, node 2t) and w2a / w2b (level 10)

  __device__ __forceinline__ void init(const d2 *__restrict__ tw, int t) {
    wa = load_pass_tw(tw, 0, 0);
    wb = load_pass_tw(tw, 3, t >> 5);
    if constexpr (TAIL3) {
      const d2 c0 = tw[(1 << 6) - 1 + (t >> 2)], c1 = tw[(1 << 7) - 1 + 2 * (t >> 2)];
      wc.w0r = c0.x; wc.w0i = c0.y; wc.w1r = c1.x; wc.w1i = c1.y;
      wc.w2ar = wc.w2ai = wc.w2br = wc.w2bi = 0.0;
      wd = load_pass_tw(tw, 8, t);
    } else {
      wc = load_pass_tw(tw, 6, t >> 2);
      const d2 a = tw[(1 << 9) - 1 + 2 * t], b = tw[(1 << 10) - 1 + 4 * t], c = tw[(1 << 10) - 1 + 4 * t + 2];
      wd.w0r = 0.0; wd.w0i = 0.0;
      wd.w1r = a.x; wd.w1i = a.y; wd.w2ar = b.x; wd.w2ai = b.y; wd.w2br = c.x; wd.w2bi = c.y;
    }
  }

  static __device__ __forceinline__ int base_a(int t) { return t; }                                  // + 256 m   (slot = j)
  static __device__ __forceinline__ int base_b1(int t) { return 256 * (t >> 5) + (t & 31); }         // + 32 m    (slot = j)
  static __device__ __forceinline__ int base_b2(int t) { return 288 * (t >> 5) + (t & 31); }         // + 36 m    (j + 4 (j>>5))
  static __device__ __forceinline__ int base_c(int t) { return 36 * (t >> 2) + (t & 3); }            // + 4 m (j + 4 (j>>5));  + off_c3(m) (j + (j>>3))
  static __device__ __forceinline__ int base_d3(int t) { return 9 * t; }                             // + m       (j + (j>>3))
  static __device__ __forceinline__ constexpr int off_c3(int m) { return 4 * m + (m >> 1); }

  // the last two stages of a three-level pass (register pairs (m, m+2), then (m, m+1))
  __device__ __forceinline__ void pass_c_fwd(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) pass_fwd_top2(re, im, wc);
    else pass_fwd(re, im, wc);
  }
  __device__ __forceinline__ void pass_c_inv(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) pass_inv_top2(re, im, wc);
    else pass_inv(re, im, wc);
  }
  __device__ __forceinline__ void pass_d_fwd(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) {
      pass_fwd(re, im, wd);
    } else {
      bf_fwd(re[0], im[0], re[2], im[2], wd.w1r, wd.w1i);
      bf_fwd(re[1], im[1], re[3], im[3], wd.w1r, wd.w1i);
      bf_fwd_i(re[4], im[4], re[6], im[6], wd.w1r, wd.w1i);
      bf_fwd_i(re[5], im[5], re[7], im[7], wd.w1r, wd.w1i);
      bf_fwd(re[0], im[0], re[1], im[1], wd.w2ar, wd.w2ai);
      bf_fwd_i(re[2], im[2], re[3], im[3], wd.w2ar, wd.w2ai);
      bf_fwd(re[4], im[4], re[5], im[5], wd.w2br, wd.w2bi);
      bf_fwd_i(re[6], im[6], re[7], im[7], wd.w2br, wd.w2bi);
    }
  }
  __device__ __forceinline__ void pass_d_inv(double (&re)[8], double (&im)[8]) const {
    if constexpr (TAIL3) {
      pass_inv(re, im, wd);
    } else {
      bf_inv(re[0], im[0], re[1], im[1], wd.w2ar, wd.w2ai);
      bf_inv_i(re[2], im[2], re[3], im[3], wd.w2ar, wd.w2ai);
      bf_inv(re[4], im[4], re[5], im[5], wd.w2br, wd.w2bi);
      bf_inv_i(re[6], im[6], re[7], im[7], wd.w2br, wd.w2bi);
      bf_inv(re[0], im[0], re[2], im[2], wd.w1r, wd.w1i);
      bf_inv(re[1], im[1], re[3], im[3], wd.w1r, wd.w1i);
      bf_inv_i(re[4], im[4], re[6], im[6], wd.w1r, wd.w1i);
      bf_inv_i(re[5], im[5], re[7], im[7], wd.w1r, wd.w1i);
    }
  }

  __device__ __forceinline__ void forward_head(double (&re)[8], double (&im)[8], d2 *xch, int t) const {
    pass_fwd(re, im, wa);
    {
      d2 *w = xch + base_a(t), *r = xch + base_b1(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[256 * m] = d2{re[m], im[m]};
      team_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[32 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_fwd(re, im, wb);
    team_sync();
    {
      d2 *w = xch + base_b2(t), *r = xch + base_c(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[36 * m] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[4 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_c_fwd(re, im);
    wave_lds_sync();
    {
      d2 *w = xch + base_c(t), *r = xch + base_d3(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[off_c3(m)] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[m]; re[m] = v.x; im[m] = v.y; }
    }
  }
  __device__ __forceinline__ void forward_tail(double (&re)[8], double (&im)[8]) const {
    pass_d_fwd(re, im);
    team_sync();
  }
  __device__ __forceinline__ void forward(double (&re)[8], double (&im)[8], d2 *xch, int t) const {
    forward_head(re, im, xch, t);
    forward_tail(re, im);
  }
  // inverse: input in layout D, output in layout A, UNSCALED
  __device__ __forceinline__ void inverse(double (&re)[8], double (&im)[8], d2 *xch, int t) const {
    pass_d_inv(re, im);
    {
      d2 *w = xch + base_d3(t), *r = xch + base_c(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[m] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[off_c3(m)]; re[m] = v.x; im[m] = v.y; }
    }
    pass_c_inv(re, im);
    wave_lds_sync();
    {
      d2 *w = xch + base_c(t), *r = xch + base_b2(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[4 * m] = d2{re[m], im[m]};
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[36 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_inv(re, im, wb);
    team_sync();
    {
      d2 *w = xch + base_b1(t), *r = xch + base_a(t);
#pragma unroll
      for (int m = 0; m < 8; m++) w[32 * m] = d2{re[m], im[m]};
      team_sync();
#pragma unroll
      for (int m = 0; m < 8; m++) { const d2 v = r[256 * m]; re[m] = v.x; im[m] = v.y; }
    }
    pass_inv(re, im, wa);
    team_sync();
  }

  // the workgroup barriers of one forward / inverse transform and nothing else (cf. Fft2048T)
  static __device__ __forceinline__ void transform_barriers_only() {
    team_sync();
    team_sync();
    team_sync();
  }
  __device__ __forceinline__ void inverse2(double (&xr)[8], double (&xi)[8], double (&yr)[8], double (&yi)[8], d2 *xch, int t) const {
    d2 *pd = xch + base_d3(t), *pc = xch + base_c(t), *pb2 = xch + base_b2(t), *pb1 = xch + base_b1(t), *pa = xch + base_a(t);
    pass_d_inv(xr, xi);
#pragma unroll
    for (int m = 0; m < 8; m++) pd[m] = d2{xr[m], xi[m]};
    wave_lds_sync();
    pass_d_inv(yr, yi);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc[off_c3(m)]; xr[m] = v.x; xi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pd[m] = d2{yr[m], yi[m]};
    wave_lds_sync();
    pass_c_inv(xr, xi);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pc[off_c3(m)]; yr[m] = v.x; yi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pc[4 * m] = d2{xr[m], xi[m]};
    wave_lds_sync();
    pass_c_inv(yr, yi);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb2[36 * m]; xr[m] = v.x; xi[m] = v.y; }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pc[4 * m] = d2{yr[m], yi[m]};
    wave_lds_sync();
    pass_inv(xr, xi, wb);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pb2[36 * m]; yr[m] = v.x; yi[m] = v.y; }
    team_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb1[32 * m] = d2{xr[m], xi[m]};
    team_sync();
    pass_inv(yr, yi, wb);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[256 * m]; xr[m] = v.x; xi[m] = v.y; }
    team_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) pb1[32 * m] = d2{yr[m], yi[m]};
    team_sync();
    pass_inv(xr, xi, wa);
#pragma unroll
    for (int m = 0; m < 8; m++) { const d2 v = pa[256 * m]; yr[m] = v.x; yi[m] = v.y; }
    pass_inv(yr, yi, wa);
    team_sync();
  }
};
using Fft4096 = Fft4096T<false>;
using Fft4096W = Fft4096T<true>;

// the wide-tail sibling of a transform (itself where there is none)
template <class F> struct WideTail { using type = F; };
template <> struct WideTail<Fft2048> { using type = Fft2048W; };
template <> struct WideTail<Fft4096> { using type = Fft4096W; };

// double -> Torus64, round to nearest, mod 2^64 (values reach ~2^84).  `scale` = 2^-64 / M.
// Same result as oracle_fft.c:round_mod_2_64 for every input (semantics of the reference's AVX-512 path,
// fft_processor_spqlios.c:155-165: vcvtpd2qq of the fractional part scaled by 2^64, which wraps 2^63 to -2^63):
//   f = frac part of v*scale in [-1/2, 1/2];  out = rint(f 2^64) mod 2^64.
// There is no 64-bit float->int conversion on the VALU, so the integer is assembled from two 32-bit halves with the
// magic-number trick (adding 1.5 * 2^52 leaves rint(x) mod 2^32 in the low dword of the sum for |x| <= 2^31):
//   h = rint(f 2^32), q = f 2^32 - h (exact, |q| <= 1/2), x = rint(q 2^32);  rint(f 2^64) = h 2^32 + x  because h 2^32 is an
//   even integer (ties-to-even is preserved);  bit 32 of the second sum's mantissa is set iff x < 0 (the borrow).
// Constant placement: v_fma_f64 (VOP3) on gfx9 takes no literal and only one scalar source, and a VOP2 v_fmac would need 1.5 * 2^52
// copied into its accumulator in front of every use (96-128 extra moves per CMUX when left to the compiler), so 2^32 is pinned
// in a scalar register pair and the magic number in a vector register pair, once per kernel (RoundCtx).
struct RoundCtx {
  double scale, magic, two32, scale32;
  __device__ __forceinline__ explicit RoundCtx(double s) : scale(s), magic(0x1.8p52), two32(0x1p32), scale32(s * 0x1p32) {
    asm volatile("" : "+v"(magic));
    asm volatile("" : "+s"(two32));
    asm volatile("" : "+s"(scale32));
  }
  // run-time ring (general_kernels.h): s = 2^-(64 + log2 M) and s 2^32, both assembled from their exponents by the scalar unit (there is no scalar
  // floating point to multiply them with)
  __device__ __forceinline__ explicit RoundCtx(int logM)
      : scale(__longlong_as_double((long long)(1023 - 64 - logM) << 52)), magic(0x1.8p52), two32(0x1p32), scale32(__longlong_as_double((long long)(1023 - 32 - logM) << 52)) {
    asm volatile("" : "+v"(magic));
    asm volatile("" : "+s"(two32));
    asm volatile("" : "+s"(scale32));
  }
};
__device__ __forceinline__ uint64_t round_mod_2_64(double v, const RoundCtx &rc) {
  double f = v * rc.scale;
  f = f - __builtin_rint(f);
  const double A = __builtin_fma(f, rc.two32, rc.magic);
  const double B = A - rc.magic;
  const double q = __builtin_fma(f, rc.two32, -B);
  const double C = __builtin_fma(q, rc.two32, rc.magic);
  const uint64_t a = (uint64_t)__double_as_longlong(A), c = (uint64_t)__double_as_longlong(C);
  const uint32_t hi = (uint32_t)a - ((uint32_t)(c >> 32) & 1u);
  return ((uint64_t)hi << 32) | (uint64_t)(uint32_t)c;
}

// acc + round_mod_2_64(v) with the 64-bit sum spelled as two 32-bit halves and an explicit carry: the rounded value exists as two
// unrelated 32-bit registers (low words of A and C), and a 64-bit add would first have to move them into a register pair
// REDUCE = false skips the reduction mod 1 in front: the magic-number add rounds f 2^32 to an integer exactly as long as |f 2^32| < 2^51,
// and the low 32 bits of that integer do not care about whole multiples of 2^32, so for |f| < 2^19 the two results are the same bits.  The
// caller may only ask for it when the bound holds for EVERY input (see pbs_kernel: |sum| <= rows * N * Bg/2 * 2^63 by construction).
template <bool REDUCE = true>
__device__ __forceinline__ uint64_t add_rounded(uint64_t acc, double v, const RoundCtx &rc) {
  // scale is a power of two, so v * scale is exact and fma(v, scale 2^32, c) == fma(v scale, 2^32, c) bit for bit: without the reduction the scaling
  // multiply folds into the two fmas
  double A, B, q;
  if constexpr (REDUCE) {
    double f = v * rc.scale;
    f = f - __builtin_rint(f);
    A = __builtin_fma(f, rc.two32, rc.magic);
    B = A - rc.magic;
    q = __builtin_fma(f, rc.two32, -B);
  } else {
    A = __builtin_fma(v, rc.scale32, rc.magic);
    B = A - rc.magic;
    q = __builtin_fma(v, rc.scale32, -B);
  }
  const double C = __builtin_fma(q, rc.two32, rc.magic);
  const uint64_t a = (uint64_t)__double_as_longlong(A), c = (uint64_t)__double_as_longlong(C);
  const uint32_t c_lo = (uint32_t)c;
  const uint32_t lo = (uint32_t)acc + c_lo;
  const uint32_t carry = lo < c_lo ? 1u : 0u;
  uint32_t borrow;   // 0 or 0xffffffff: -(bit 32 of C).  Spelled in assembly: the compiler turns the bit-field builtin into v_alignbit_b32 + v_ashrrev_i32
  asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(borrow) : "v"((uint32_t)(c >> 32)));
  const uint32_t hi = (uint32_t)(acc >> 32) + (uint32_t)a + borrow + carry;
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// (double)(int64_t)x, correctly rounded (matches the C cast used by the reference and the oracle)
__device__ __forceinline__ double torus_to_double(uint64_t x) { return (double)(int64_t)x; }

}  // namespace mosfhet
