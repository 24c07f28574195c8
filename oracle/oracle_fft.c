/*
 * oracle_fft.c -- negacyclic double-precision transform of the oracle.
 * TEST INFRASTRUCTURE ONLY (see mosfhet_oracle.h).
 *
 * Mathematical map (identical to the reference's "Torus -> DFT", ffnt.c:235-271,820-831,
 * and spqlios' ifft): a real polynomial p of degree < N is folded to M = N/2 complex
 * coefficients z_j = p_j + i p_{j+M} and evaluated at the M roots y of y^M = i, i.e. at
 * y = psi^(4k+1), psi = exp(i pi / N); these and their conjugates are all roots of X^N + 1, so
 * pointwise products are negacyclic products.  The reference obtains the same values as
 * "twist by psi^j, then an M-point FFT"; here the twist is folded into the butterflies:
 *
 *   level lev = 0..log2(M)-1 splits every residue  z mod (y^L - c)  into
 *   z mod (y^(L/2) - s)  and  z mod (y^(L/2) + s),  s = sqrt(c):   (a, b) -> (a + s b, a - s b)
 *
 * so each level-lev node nu has ONE twiddle  s = exp(2 pi i (4 bitrev_lev(nu) + 1) / 2^(lev+3)),
 * the second child of a parent has exactly i times the twiddle of the first, and the output is
 * left in the order the recursion produces (never un-permuted, as in the reference; DFT-domain
 * data never leaves the engine).  The inverse runs the levels backwards with
 * (u, v) -> (u + v, (u - v) conj(s)) and a final exact scale by 1/M.
 *
 * FIXED OPERATION ORDER (the HIP kernels reproduce it exactly -> bit-identical results):
 *   forward butterfly   xr = fma(-si, bi, fma(sr, br, ar));  xi = fma(si, br, fma(sr, bi, ai));
 *                       yr = fma(2, ar, -xr);                yi = fma(2, ai, -xi);
 *   inverse butterfly   pr = ur + vr; pi = ui + vi; dr = ur - vr; di = ui - vi;
 *                       qr = fma(sr, dr, si * di);           qi = fma(sr, di, -(si * dr));
 *   complex MAC         or = fma(-di, bi, fma(dr, br, or));  oi = fma(di, br, fma(dr, bi, oi));
 *   rounding            w = v * 2^-log2(M); f = w * 2^-64; f -= rint(f); g = rint(f * 2^64);
 *                       out = g mod 2^64                             (round to nearest, as the
 *                       reference's default AVX-512 path fft_processor_spqlios.c:155-165; the kernels
 *                       assemble the same integer from two 32-bit halves, see negacyclic_fft.h)
 * Compile with -ffp-contract=off: every fma above is explicit, nothing else may be fused.
 */
#include "mosfhet_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_fft_plan {
  int N, M, logM;
  double *tw; /* (re,im) for node index (2^lev - 1 + nu), M-1 entries */
};

static unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

void orc_fft_make_twiddles(double *out, int N) {
  const int M = N / 2;
  int logM = 0;
  while ((1 << logM) < M) logM++;
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int lev = 0; lev < logM; lev++) {
    for (unsigned nu = 0; nu < (1u << lev); nu++) {
      double *e = out + 2 * ((size_t)(1u << lev) - 1 + nu);
      if (nu & 1) { /* i times the sibling: exact, so kernels may derive it by swap/negate */
        e[0] = -e[-1];
        e[1] = e[-2];
      } else {
        const long double frac = (long double)(4 * bitrev(nu, lev) + 1) / (long double)(1ull << (lev + 3));
        e[0] = (double)cosl(two_pi * frac);
        e[1] = (double)sinl(two_pi * frac);
      }
    }
  }
}

orc_fft_plan *orc_fft_plan_new(int N) {
  orc_fft_plan *p = (orc_fft_plan *)malloc(sizeof(*p));
  p->N = N;
  p->M = N / 2;
  p->logM = 0;
  while ((1 << p->logM) < p->M) p->logM++;
  p->tw = (double *)malloc(sizeof(double) * 2 * (size_t)(p->M - 1 + 1));
  orc_fft_make_twiddles(p->tw, N);
  return p;
}

void orc_fft_plan_free(orc_fft_plan *p) {
  if (!p) return;
  free(p->tw);
  free(p);
}

const double *orc_fft_twiddles(const orc_fft_plan *p, int *count_complex) {
  if (count_complex) *count_complex = p->M - 1;
  return p->tw;
}

static void fwd_inplace(const orc_fft_plan *p, double *z) {
  const int M = p->M;
  for (int lev = 0; lev < p->logM; lev++) {
    const int half = M >> (lev + 1);
    for (int nu = 0; nu < (1 << lev); nu++) {
      const double sr = p->tw[2 * ((1 << lev) - 1 + nu)], si = p->tw[2 * ((1 << lev) - 1 + nu) + 1];
      double *lo = z + 2 * (size_t)(nu * 2 * half), *hi = lo + 2 * (size_t)half;
      for (int j = 0; j < half; j++) {
        const double ar = lo[2 * j], ai = lo[2 * j + 1], br = hi[2 * j], bi = hi[2 * j + 1];
        const double xr = fma(-si, bi, fma(sr, br, ar));
        const double xi = fma(si, br, fma(sr, bi, ai));
        lo[2 * j] = xr;
        lo[2 * j + 1] = xi;
        hi[2 * j] = fma(2.0, ar, -xr);
        hi[2 * j + 1] = fma(2.0, ai, -xi);
      }
    }
  }
}

static void inv_inplace(const orc_fft_plan *p, double *z) {
  const int M = p->M;
  for (int lev = p->logM - 1; lev >= 0; lev--) {
    const int half = M >> (lev + 1);
    for (int nu = 0; nu < (1 << lev); nu++) {
      const double sr = p->tw[2 * ((1 << lev) - 1 + nu)], si = p->tw[2 * ((1 << lev) - 1 + nu) + 1];
      double *lo = z + 2 * (size_t)(nu * 2 * half), *hi = lo + 2 * (size_t)half;
      for (int j = 0; j < half; j++) {
        const double ur = lo[2 * j], ui = lo[2 * j + 1], vr = hi[2 * j], vi = hi[2 * j + 1];
        const double dr = ur - vr, di = ui - vi;
        lo[2 * j] = ur + vr;
        lo[2 * j + 1] = ui + vi;
        hi[2 * j] = fma(sr, dr, si * di);
        hi[2 * j + 1] = fma(sr, di, -(si * dr));
      }
    }
  }
}

/* src/polynomial.c:368-375 -> execute_reverse_torus64 (fft_processor_spqlios.c:81-97, ffnt.c:820-831):
 * input conversion is (double)(int64_t)c. */
void orc_torus_to_dft(const orc_fft_plan *p, double *out, const Torus *in) {
  const int M = p->M;
  for (int j = 0; j < M; j++) {
    out[2 * j] = (double)(int64_t)in[j];
    out[2 * j + 1] = (double)(int64_t)in[j + M];
  }
  fwd_inplace(p, out);
}

void orc_int_to_dft(const orc_fft_plan *p, double *out, const int64_t *in) {
  const int M = p->M;
  for (int j = 0; j < M; j++) {
    out[2 * j] = (double)in[j];
    out[2 * j + 1] = (double)in[j + M];
  }
  fwd_inplace(p, out);
}

static inline Torus round_mod_2_64(double v, double inv_m) {
  const double w = v * inv_m;
  double f = w * 0x1p-64;
  f = f - rint(f);
  const double g = rint(f * 0x1p64);                 /* in [-2^63, 2^63] */
  if (g >= 0x1p63) return (Torus)1 << 63;            /* vcvtpd2qq wraps the exact tie +2^63 to -2^63 = 2^63 mod 2^64 */
  return (Torus)(int64_t)g;
}

/* src/polynomial.c:359-366 -> execute_direct_torus64 (fft_processor_spqlios.c:128-165) */
void orc_dft_to_torus(const orc_fft_plan *p, Torus *out, const double *in) {
  const int M = p->M;
  double *z = (double *)malloc(sizeof(double) * (size_t)p->N);
  memcpy(z, in, sizeof(double) * (size_t)p->N);
  inv_inplace(p, z);
  const double inv_m = 1.0 / (double)M;
  for (int j = 0; j < M; j++) {
    out[j] = round_mod_2_64(z[2 * j], inv_m);
    out[j + M] = round_mod_2_64(z[2 * j + 1], inv_m);
  }
  free(z);
}

/* src/polynomial.c:379-401 */
void orc_dft_mul(double *out, const double *a, const double *b, int N) {
  for (int j = 0; j < N / 2; j++) {
    const double ar = a[2 * j], ai = a[2 * j + 1], br = b[2 * j], bi = b[2 * j + 1];
    out[2 * j] = fma(-ai, bi, fma(ar, br, 0.0));
    out[2 * j + 1] = fma(ai, br, fma(ar, bi, 0.0));
  }
}

/* src/polynomial.c:406-426 */
void orc_dft_mul_addto(double *out, const double *a, const double *b, int N) {
  for (int j = 0; j < N / 2; j++) {
    const double ar = a[2 * j], ai = a[2 * j + 1], br = b[2 * j], bi = b[2 * j + 1];
    out[2 * j] = fma(-ai, bi, fma(ar, br, out[2 * j]));
    out[2 * j + 1] = fma(ai, br, fma(ar, bi, out[2 * j + 1]));
  }
}

/* src/polynomial.c:276-288  polynomial_mul_torus */
void orc_poly_mul_fft(const orc_fft_plan *p, Torus *out, const Torus *a, const Torus *b) {
  const int N = p->N;
  double *fa = (double *)malloc(sizeof(double) * 3 * (size_t)N), *fb = fa + N, *fc = fb + N;
  orc_torus_to_dft(p, fa, a);
  orc_torus_to_dft(p, fb, b);
  orc_dft_mul(fc, fa, fb, N);
  orc_dft_to_torus(p, out, fc);
  free(fa);
}
