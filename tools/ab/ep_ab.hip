// ep_ab.hip -- one-kernel build of the external-product kernel for same-box A/B experiments (cf. pbs_ab.hip).
#include "../../mosfhet_amd/csrc/bootstrap_kernels.h"
using namespace mosfhet;
#ifndef AB_N
#define AB_N 1024
#endif
#ifndef AB_L
#define AB_L 2
#endif
#ifndef AB_BG
#define AB_BG 8
#endif
#ifndef AB_CMUX
#define AB_CMUX false   // true: the CMUX form (out = in0 + TRGSW (.) (in - in0)); ab_ep_in0 must then point at a batch of the same shape
#endif
#ifndef AB_FORM
#define AB_FORM 0   // 1: plain unit loop, 2: software-pipelined unit loop on any ring (external_product_kernel: FORM)
#endif
#if AB_N == 1024
using ABF = Fft1024;
#elif AB_N == 2048
#ifdef AB_LTW
using ABF = Fft2048L;   // pass twiddles in LDS, rows two at a time: lvl2's production transform
#else
using ABF = Fft2048;
#endif
#else
using ABF = Fft4096;
#endif
extern "C" int ab_ep_bg_rt = 0;   // run-time gadget base of AB_BG = 0 builds
extern "C" double *ab_ep_out_dft = nullptr;      // != nullptr: the result stays in the DFT domain (no inverse transforms, no rounding): which half of a unit goes wrong?
extern "C" const uint64_t *ab_ep_in0 = nullptr;   // passed as `in0` (unused without CMUX): the debug buffer of tools/spill_hazard
extern "C" int ab_ep(const double *d_row, const double *d_tw, const uint64_t *d_in, uint64_t *d_out, int count, int grid, int reps, float *ms_per_launch) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
  hipEventRecord(e0, nullptr);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((external_product_kernel<ABF, AB_L, AB_BG, AB_CMUX, AB_FORM>), dim3((unsigned)grid), dim3(ABF::THREADS), 0, nullptr, (const d2 *)d_row, (const d2 *)d_tw, d_in, d_out,
                       AB_BG ? AB_BG : ab_ep_bg_rt, count, (size_t)0, (size_t)(2 * ABF::N), ab_ep_in0, (d2 *)ab_ep_out_dft);
  hipEventRecord(e1, nullptr);
  if (hipEventSynchronize(e1) != hipSuccess) return -2;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_per_launch = ms / (float)reps;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

#if AB_N == 1024
// LDS-key variant: one workgroup of 8 teams per CU
extern "C" int ab_epl(const double *d_row, const double *d_tw, const uint64_t *d_in, uint64_t *d_out, int count, int grid, int reps, float *ms_per_launch) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
  hipEventRecord(e0, nullptr);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((external_product_ldskey_kernel<2, 8, false>), dim3((unsigned)grid), dim3(512), 0, nullptr, (const d2 *)d_row, (const d2 *)d_tw, d_in, d_out, 8, count,
                       (size_t)2048, (const uint64_t *)nullptr, (d2 *)nullptr);
  hipEventRecord(e1, nullptr);
  if (hipEventSynchronize(e1) != hipSuccess) return -2;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_per_launch = ms / (float)reps;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
#endif
