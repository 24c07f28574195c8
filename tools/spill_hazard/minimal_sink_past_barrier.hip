// minimal_sink_past_barrier.hip -- minimal reproducer, fit for an upstream report, of the compiler behaviour behind the "wrong units" of rounds 3 - 5
// (experiments/README.md, Round 5): LDS loads that stand IN FRONT of __syncthreads() in the source are emitted BEHIND the s_barrier, sunk into the successor block
// of a conditional branch where their first use is.  ROCm 7.2.0 (AMD clang 22.0.0git roc-7.2.0), gfx950 (the same with gfx942):
//
//     hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o min.s tools/spill_hazard/minimal_sink_past_barrier.hip
//     grep -n "ds_read\|ds_write\|s_barrier\|wave barrier\|s_cbranch\|^.LBB" min.s
//
// k_raw, loop header:     ds_write_b128 x4 ; "; wave barrier" ; s_barrier ; s_cbranch_vccz .LBB0_2 ;   .LBB0_2:  ds_read_b128 x4   <-- the reads of r[], behind the barrier
// k_fixed:                ds_write_b128 x4 ; "; wave barrier" ; ds_read_b128 x4 ; s_barrier ; s_cbranch ...
//
// Why it is wrong: after the barrier the OTHER wavefront of the workgroup writes the same LDS words (exchange 2); with the reads behind the barrier it can do so before
// this wavefront has read them.  Machine sinking treats the loads as movable because nothing between them and the end of their block "may store": S_BARRIER and the
// ATOMIC_FENCE pseudos of the workgroup fences are side-effect-only to it.  An empty asm with a memory clobber next to the barrier (k_fixed; workgroup_sync() in
// mosfhet_amd/csrc/negacyclic_fft.h) is treated as a load + store and pins the reads; it emits no instruction.
// tools/check_lds_barriers.py finds the pattern in a listing; tests/test_host_and_abi.py compiles this file and reports whether the toolchain still does it.
#include <hip/hip_runtime.h>

template <bool FIXED>
__device__ __forceinline__ void barrier() {
  if (FIXED) asm volatile("" ::: "memory");
  __syncthreads();
  if (FIXED) asm volatile("" ::: "memory");
}

template <bool FIXED>
__device__ __forceinline__ void body(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ extra, int n, int have_extra) {
  __shared__ double buf[1024];
  const int t = threadIdx.x;
  double acc = 0.0, e = 0.0;
  for (int u = 0; u < n; u++) {
    // exchange 1 stays inside a wavefront: every thread writes 8 values and reads the 8 its neighbour wrote (wave-level ordering is enough: DS instructions of a wave execute in order)
    for (int m = 0; m < 8; m++) buf[8 * t + m] = in[(u * 128 + t) * 8 + m];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double r[8];
    for (int m = 0; m < 8; m++) r[m] = buf[8 * (t ^ 1) + m];
    barrier<FIXED>();                          // nobody reads buf any more: the OTHER wavefront may overwrite it from here on
    if (have_extra) e = extra[u * 128 + t];    // a conditional block behind the barrier ...
    for (int m = 0; m < 8; m++) acc += r[m] * (e + m);   // ... and the first use of r[] behind that
    // exchange 2 crosses the wavefronts: thread t writes where thread t ^ 64 has just read
    for (int m = 0; m < 8; m++) buf[8 * (t ^ 64) + m] = acc + m;
    barrier<FIXED>();
    acc += buf[8 * t];
    barrier<FIXED>();
  }
  out[blockIdx.x * 128 + t] = acc;
}

extern "C" __global__ __launch_bounds__(128) void k_raw(const double *in, double *out, const double *extra, int n, int have_extra) { body<false>(in, out, extra, n, have_extra); }
extern "C" __global__ __launch_bounds__(128) void k_fixed(const double *in, double *out, const double *extra, int n, int have_extra) { body<true>(in, out, extra, n, have_extra); }
