// ubench.hip -- gfx950 micro-benchmarks behind the design decisions of the bootstrap kernel (experiments/README.md).
// Build:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ubench tools/ubench/ubench.hip      Run on the GPU box: tools/ubench/ubench
// Every test runs `waves` wavefronts per workgroup (64 threads each; a workgroup's waves spread over the CU's 4 SIMDs) on `blocks`
// workgroups, times a loop of ITER iterations with s_memtime (shader cycles) and prints cycles per instruction per wavefront.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;

// ---- 1. FP64 fma: CH independent dependency chains, 16 fmas per chain per iteration
template <int CH>
__global__ void k_fma64(double *out, long long *cyc, double a, double b) {
  double x[CH];
  for (int c = 0; c < CH; c++) x[c] = (double)(threadIdx.x + c);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) x[c] = __builtin_fma(x[c], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < CH; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- 2. 32-bit integer VALU (v_add_u32 chains), same shape
template <int CH>
__global__ void k_add32(uint32_t *out, long long *cyc, uint32_t a) {
  uint32_t x[CH];
  for (int c = 0; c < CH; c++) x[c] = threadIdx.x + c;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) { x[c] = x[c] * 3u + a; asm volatile("" : "+v"(x[c])); }   // v_mad_u32_u24 / v_mul_lo: see asm; kept opaque
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  for (int c = 0; c < CH; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// plain v_xor / v_add (one-pass integer ops)
template <int CH>
__global__ void k_xor32(uint32_t *out, long long *cyc, uint32_t a) {
  uint32_t x[CH];
  for (int c = 0; c < CH; c++) x[c] = threadIdx.x + c;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  for (int c = 0; c < CH; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- 3. mix: per iteration 16 x (4 independent fma64 + MIX v_add_u32)
template <int MIX>
__global__ void k_mix(double *out, long long *cyc, double a, double b, uint32_t ia) {
  double x[4];
  uint32_t y[4];
  for (int c = 0; c < 4; c++) { x[c] = (double)(threadIdx.x + c); y[c] = threadIdx.x + c; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        if (c < MIX) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[c]) : "v"(ia));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < 4; c++) s += x[c] + (double)y[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- 4. v_mfma_f64_16x16x4_f64: NACC independent accumulators, optional FMA64 vector fmas interleaved per MFMA
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int VF>
__global__ void k_mfma64(double *out, long long *cyc, double a, double b) {
  d4 acc[NACC];
  for (int c = 0; c < NACC; c++) acc[c] = d4{0, 0, 0, 0};
  double va = a + threadIdx.x, vb = b - threadIdx.x;
  double x[4] = {1.0, 2.0, 3.0, 4.0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER / 4; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < NACC; c++) {
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc[c], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VF; v++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[v & 3]) : "v"(a), "v"(b));
      }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = x[0] + x[1] + x[2] + x[3];
  for (int c = 0; c < NACC; c++) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- 5. numerics of v_mfma_f64_16x16x4_f64: one MFMA on random data, compared on the host with fma chains in both k orders
__global__ void k_mfma64_num(const double *A, const double *B, const double *C, double *D) {
  // A: 16x4 (row = lane & 15, k = lane >> 4), B: 4x16 (k = lane >> 4, col = lane & 15), C/D: col = lane & 15, row = (lane >> 4) + 4 reg
  const int lane = threadIdx.x;
  const double a = A[(lane & 15) * 4 + (lane >> 4)], b = B[(lane >> 4) * 16 + (lane & 15)];
  d4 c;
  for (int r = 0; r < 4; r++) c[r] = C[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = c[r];
}

// ---- 6. LDS transposes: 8 x ds_write_b128 + wave barrier + 8 x ds_read_b128 per iteration (the transform's exchange), per wave region
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool SPLIT64>
__global__ void k_lds(double *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) d2 lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  d2 *xch = lds + w * 576;
  d2 v[8];
  for (int m = 0; m < 8; m++) v[m] = d2{(double)lane, (double)m};
  d2 *pa = xch + lane, *pb = xch + 72 * (lane >> 3) + (lane & 7);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
    if (SPLIT64) {
      double *qa = (double *)xch + lane, *qb = (double *)xch + 72 * (lane >> 3) + (lane & 7);
#pragma unroll
      for (int m = 0; m < 8; m++) qa[72 * m] = v[m].x;
#pragma unroll
      for (int m = 0; m < 8; m++) qa[576 + 72 * m] = v[m].y;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int m = 0; m < 8; m++) v[m].x = qb[8 * m];
#pragma unroll
      for (int m = 0; m < 8; m++) v[m].y = qb[576 + 8 * m];
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) pa[72 * m] = v[m];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int m = 0; m < 8; m++) v[m] = pb[8 * m];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int m = 0; m < 8; m++) s += v[m].x + v[m].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
}

// ---- 7. v_permlane32_swap / v_permlane16_swap: 16 swaps per iteration (one 3-bit exchange of 8 doubles would need 3 x 16)
template <int WHICH>
__global__ void k_permlane(uint32_t *out, long long *cyc) {
  uint32_t x[16];
  for (int c = 0; c < 16; c++) x[c] = threadIdx.x * 16 + c;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      if (WHICH == 32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[c]), "+v"(x[c + 1]));
      else asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[c]), "+v"(x[c + 1]));
    }
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      if (WHICH == 32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[c]), "+v"(x[(c + 3) & 15]));
      else asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[c]), "+v"(x[(c + 3) & 15]));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  for (int c = 0; c < 16; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double median(std::vector<long long> &v) {
  std::sort(v.begin(), v.end());
  return (double)v[v.size() / 2];
}

template <class Launch>
static void run(const char *name, int blocks, int waves, double instr_per_wave, Launch launch) {
  const int nw = blocks * waves;
  long long *d_cyc;
  void *d_out;
  CHECK(hipMalloc(&d_cyc, nw * sizeof(long long)));
  CHECK(hipMalloc(&d_out, (size_t)blocks * waves * 64 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch(d_out, d_cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch(d_out, d_cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> cyc(nw);
  CHECK(hipMemcpy(cyc.data(), d_cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
  const double med = median(cyc);
  printf("%-44s blocks=%4d waves/wg=%2d  cycles/instr/wave=%7.2f   (median wave %9.0f cycles, launch %.3f ms)\n", name, blocks, waves,
         med / instr_per_wave, med, ms);
  (void)hipFree(d_cyc);
  (void)hipFree(d_out);
}

#include <algorithm>

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  const int CU = prop.multiProcessorCount;
  const double F = 16.0 * ITER;
  for (int blocks : {1, CU}) {
    for (int waves : {1, 4, 8, 16}) {
      run("fma_f64 1 chain", blocks, waves, F * 1, [&](void *o, long long *c) { hipLaunchKernelGGL(k_fma64<1>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("fma_f64 2 chains", blocks, waves, F * 2, [&](void *o, long long *c) { hipLaunchKernelGGL(k_fma64<2>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("fma_f64 4 chains", blocks, waves, F * 4, [&](void *o, long long *c) { hipLaunchKernelGGL(k_fma64<4>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("fma_f64 8 chains", blocks, waves, F * 8, [&](void *o, long long *c) { hipLaunchKernelGGL(k_fma64<8>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("v_add_u32 1 chain", blocks, waves, F * 1, [&](void *o, long long *c) { hipLaunchKernelGGL(k_xor32<1>, dim3(blocks), dim3(64 * waves), 0, 0, (uint32_t *)o, c, 7u); });
      run("v_add_u32 4 chains", blocks, waves, F * 4, [&](void *o, long long *c) { hipLaunchKernelGGL(k_xor32<4>, dim3(blocks), dim3(64 * waves), 0, 0, (uint32_t *)o, c, 7u); });
      run("mul/mad u32 4 chains", blocks, waves, F * 4, [&](void *o, long long *c) { hipLaunchKernelGGL(k_add32<4>, dim3(blocks), dim3(64 * waves), 0, 0, (uint32_t *)o, c, 7u); });
      run("mix: 4 fma_f64 + 0 add_u32 (per fma)", blocks, waves, F * 4, [&](void *o, long long *c) { hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9, 7u); });
      run("mix: 4 fma_f64 + 2 add_u32 (per fma)", blocks, waves, F * 4, [&](void *o, long long *c) { hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9, 7u); });
      run("mix: 4 fma_f64 + 4 add_u32 (per fma)", blocks, waves, F * 4, [&](void *o, long long *c) { hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9, 7u); });
    }
    for (int waves : {4, 8}) {
      const double Fm = 16.0 * (ITER / 4);
      run("mfma_f64_16x16x4 1 acc (per mfma)", blocks, waves, Fm * 1, [&](void *o, long long *c) { hipLaunchKernelGGL((k_mfma64<1, 0>), dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("mfma_f64_16x16x4 4 acc (per mfma)", blocks, waves, Fm * 4, [&](void *o, long long *c) { hipLaunchKernelGGL((k_mfma64<4, 0>), dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("mfma_f64 4 acc + 4 fma_f64 each (per mfma)", blocks, waves, Fm * 4, [&](void *o, long long *c) { hipLaunchKernelGGL((k_mfma64<4, 4>), dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("mfma_f64 4 acc + 8 fma_f64 each (per mfma)", blocks, waves, Fm * 4, [&](void *o, long long *c) { hipLaunchKernelGGL((k_mfma64<4, 8>), dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
      run("mfma_f64 4 acc + 16 fma_f64 each (per mfma)", blocks, waves, Fm * 4, [&](void *o, long long *c) { hipLaunchKernelGGL((k_mfma64<4, 16>), dim3(blocks), dim3(64 * waves), 0, 0, (double *)o, c, 1.0000001, 1e-9); });
    }
    for (int waves : {1, 4, 8}) {
      run("LDS exchange 8xb128 w + 8xb128 r (per xchg)", blocks, waves, (double)ITER, [&](void *o, long long *c) { hipLaunchKernelGGL(k_lds<false>, dim3(blocks), dim3(64 * waves), waves * 576 * 16, 0, (double *)o, c); });
      run("LDS exchange 16xb64 w + 16xb64 r (per xchg)", blocks, waves, (double)ITER, [&](void *o, long long *c) { hipLaunchKernelGGL(k_lds<true>, dim3(blocks), dim3(64 * waves), waves * 576 * 16, 0, (double *)o, c); });
      run("v_permlane32_swap (per swap)", blocks, waves, 16.0 * ITER, [&](void *o, long long *c) { hipLaunchKernelGGL(k_permlane<32>, dim3(blocks), dim3(64 * waves), 0, 0, (uint32_t *)o, c); });
      run("v_permlane16_swap (per swap)", blocks, waves, 16.0 * ITER, [&](void *o, long long *c) { hipLaunchKernelGGL(k_permlane<16>, dim3(blocks), dim3(64 * waves), 0, 0, (uint32_t *)o, c); });
    }
  }
  // numerics of the FP64 MFMA
  {
    std::vector<double> A(64), B(64), C(256), D(256);
    srand(12345);
    auto rnd = []() { return ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20); };
    for (auto &x : A) x = rnd();
    for (auto &x : B) x = rnd();
    for (auto &x : C) x = rnd();
    double *dA, *dB, *dC, *dD;
    CHECK(hipMalloc(&dA, 512)); CHECK(hipMalloc(&dB, 512)); CHECK(hipMalloc(&dC, 2048)); CHECK(hipMalloc(&dD, 2048));
    CHECK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma64_num, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    CHECK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    int fwd = 0, rev = 0, sep = 0;
    for (int i = 0; i < 16; i++)
      for (int j = 0; j < 16; j++) {
        double f = C[i * 16 + j], r = C[i * 16 + j], s = 0;
        for (int k = 0; k < 4; k++) f = fma(A[i * 4 + k], B[k * 16 + j], f);
        for (int k = 3; k >= 0; k--) r = fma(A[i * 4 + k], B[k * 16 + j], r);
        for (int k = 0; k < 4; k++) s = fma(A[i * 4 + k], B[k * 16 + j], s);
        s += C[i * 16 + j];
        fwd += D[i * 16 + j] == f; rev += D[i * 16 + j] == r; sep += D[i * 16 + j] == s;
      }
    printf("mfma_f64_16x16x4 numerics: of 256 outputs, %d equal the fma chain k=0..3 from C, %d the chain k=3..0, %d the dot product added to C last\n", fwd, rev, sep);
  }
  return 0;
}
