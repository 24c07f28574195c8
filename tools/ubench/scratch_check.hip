// scratch_check.hip -- does private (scratch) memory keep a wave's data while other workgroups of the same kernel run on the CU?
// Round-3 diagnosis of the pipelined external-product kernel on two-wavefront rings (experiments/README.md): whole units came out wrong, never from
// the first workgroup of a CU, only in builds whose register spills went to scratch.  Every thread keeps a tag in a private array (forced to scratch by
// a run-time index), works for a while (LDS traffic, global loads in flight, barriers) and checks the tag again.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/scratch_check.hip -o tools/ubench/scratch_check && tools/ubench/scratch_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int THREADS>
__global__ __launch_bounds__(THREADS, 2) void scratch_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, unsigned *errors, int iters, int idx_rt, int words) {
  __shared__ uint64_t lds[2304];   // 18 KiB: four workgroups of 128 threads per CU
  volatile uint64_t priv[4];
  const uint64_t tag = ((uint64_t)blockIdx.x << 32) | (uint64_t)threadIdx.x * 0x9E3779B9u;
  priv[idx_rt & 3] = tag;
  priv[(idx_rt + 1) & 3] = ~tag;
  uint64_t acc = 0;
  unsigned bad = 0;
  for (int it = 0; it < iters; it++) {
    const uint64_t *p = in + ((size_t)(blockIdx.x * 131 + it * 977) % (size_t)words) * THREADS;
    uint64_t v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = __builtin_nontemporal_load(&p[threadIdx.x + (size_t)m * THREADS * 17 % (size_t)words]);
    lds[threadIdx.x] = acc;
    __syncthreads();
    acc += lds[(threadIdx.x * 7 + it) % THREADS];
    __syncthreads();
    if (priv[idx_rt & 3] != tag) bad++;
    if (priv[(idx_rt + 1) & 3] != ~tag) bad++;
#pragma unroll
    for (int m = 0; m < 8; m++) acc += v[m];
  }
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
  if (bad) atomicAdd(errors, bad);
}

int main() {
  const int words = 1 << 20;
  uint64_t *in, *out;
  unsigned *err;
  hipMalloc(&in, (size_t)words * 256 * 8 + (1 << 20));
  hipMalloc(&out, (size_t)8192 * 256 * 8);
  hipMalloc(&err, 4);
  hipMemset(in, 1, (size_t)words * 256 * 8);
  for (int threads : {64, 128, 256})
    for (int grid : {256, 512, 1024, 2048, 4096}) {
      hipMemset(err, 0, 4);
      for (int rep = 0; rep < 5; rep++) {
        if (threads == 64) hipLaunchKernelGGL(scratch_kernel<64>, dim3(grid), dim3(64), 0, 0, in, out, err, 200, rep, words);
        else if (threads == 128) hipLaunchKernelGGL(scratch_kernel<128>, dim3(grid), dim3(128), 0, 0, in, out, err, 200, rep, words);
        else hipLaunchKernelGGL(scratch_kernel<256>, dim3(grid), dim3(256), 0, 0, in, out, err, 200, rep, words);
      }
      unsigned h = 0;
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
      printf("threads %3d grid %4d: %u scratch mismatches (%s)\n", threads, grid, h, hipGetErrorString(e));
    }
  return 0;
}
