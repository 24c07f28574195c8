// pbs_ab.hip -- one-kernel build of the fused bootstrap for same-box A/B experiments (experiments/README.md): the production kernel source
// (mosfhet_amd/csrc/bootstrap_kernels.h) compiled alone, with the experiment's -D switches, behind a two-function C ABI.  tools/ab/run_ab.py builds
// one .so per variant, runs them back to back on one GPU and checks every variant's output against the production library bit for bit.
#include "../../mosfhet_amd/csrc/bootstrap_kernels.h"

using namespace mosfhet;

#ifndef AB_WIDE
#define AB_WIDE false
#endif
#ifndef AB_N
#define AB_N 1024
#endif
#if AB_N == 1024
using ABF = Fft1024;
#elif AB_N == 2048
#ifdef AB_LTW
using ABF = Fft2048L;          // pass twiddles in LDS; rows two at a time (cmux_rows2)
#else
using ABF = Fft2048T<AB_WIDE>;
#endif
#else
using ABF = Fft4096T<AB_WIDE>;
#endif
#ifndef AB_L
#define AB_L 2
#endif
#ifndef AB_BG
#define AB_BG 8
#endif

extern "C" int ab_pbs(const double *d_bk, const double *d_tw, const uint64_t *d_in, const uint64_t *d_tv, uint64_t *d_out, int n, int count, int precision, int reps,
                      float *ms_per_launch) {
  PbsParams p;
  p.bk = (const d2 *)d_bk;
  p.tw = (const d2 *)d_tw;
  p.in = d_in;
  p.tv = d_tv;
  p.out = d_out;
  p.tv_stride = 0;
  p.n = n;
  p.Bg_bit = AB_BG;
  p.pre = 1;
  p.kappa = 0;
  p.theta = 0;
  p.prec_offset = (uint64_t)((int64_t)(18446744073709551616.0 * (1. / (4 * (double)(1 << (precision - 1))))));
  p.extract = 1;
  p.skip_init = 0;
#ifdef AB_PACE
  static unsigned int *d_pace = nullptr;
  if (!d_pace && hipMalloc((void **)&d_pace, 288 * 4) != hipSuccess) return -4;   // pace_teams' block: 8 per-XCD counters + flag
  p.pace = d_pace;
  p.pace_every = AB_PACE;
#endif
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
  hipEventRecord(e0, nullptr);
#ifdef AB_DISTURB
  static void *d_big = nullptr;   // a 256 MB fill in front of every launch: what a composition's other kernels do to the caches (experiments/README.md round 4)
  if (!d_big && hipMalloc(&d_big, (size_t)256 << 20) != hipSuccess) return -5;
#endif
  for (int r = 0; r < reps; r++) {
#ifdef AB_DISTURB
    (void)hipMemsetAsync(d_big, r, (size_t)256 << 20, nullptr);
#endif
#ifdef AB_PACE
    (void)hipMemsetAsync(d_pace, 0, 288 * 4, nullptr);
#endif
    hipLaunchKernelGGL((pbs_kernel<ABF, AB_L, AB_BG>), dim3((unsigned)count), dim3(ABF::THREADS), 0, nullptr, p);
  }
  hipEventRecord(e1, nullptr);
  if (hipEventSynchronize(e1) != hipSuccess) return -2;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_per_launch = ms / (float)reps;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
