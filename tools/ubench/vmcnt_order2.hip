// vmcnt_order2.hip -- are SCRATCH loads and GLOBAL loads of one wave completed in issue order with respect to each other (one vmcnt counter)?
// Round-3 diagnosis (experiments/README.md): the pipelined external-product kernel fails only in builds with spill code, where a scratch reload of
// spilled ciphertext words is followed by 16 global loads and consumed behind `s_waitcnt vmcnt(N)` with N = the younger global loads.
//   [32 nt global stores] -> scratch_load (older) -> 16 nt global loads (younger, HBM misses) -> s_waitcnt vmcnt(16) -> copy the scratch-loaded register
//   -> s_waitcnt vmcnt(0) -> compare the copy with the value that was stored.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(5))) uint64_t *priv_ptr;

template <bool STORES>
__global__ __launch_bounds__(128, 2) void order2_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, unsigned *errors, int units, int iters) {
  __shared__ uint64_t lds[2304];
  volatile uint64_t priv[8];
  unsigned bad = 0;
  uint64_t keep = 0;
  for (int it = 0; it < iters; it++) {
    const uint64_t tag = ((uint64_t)blockIdx.x << 40) ^ ((uint64_t)threadIdx.x << 20) ^ (uint64_t)it * 0x9E3779B97F4A7C15ull;
    priv[it & 7] = tag;                                           // compiler-generated scratch store
    // thrash the L1 in between (the failing kernel reloads its spill a whole unit later)
    const uint64_t *q = in + ((size_t)((blockIdx.x * 40503u + it * 2654435761u) % (unsigned)units)) * 2048 + threadIdx.x;
    uint64_t acc = 0;
#pragma unroll
    for (int m = 0; m < 16; m++) acc += q[m * 128];
    lds[threadIdx.x] = acc;
    __syncthreads();
    keep += lds[(threadIdx.x + 3) & 127];
    __syncthreads();
    const uint64_t *p = in + ((size_t)((blockIdx.x * 2654435761u + it * 40503u + 7u) % (unsigned)units)) * 2048 + threadIdx.x;
    uint64_t *o = out + ((size_t)blockIdx.x * 128 + threadIdx.x) * 32;
    priv_ptr sp = (priv_ptr)&priv[it & 7];
    uint64_t r[16], s, c;
    if constexpr (STORES) {
#pragma unroll
      for (int m = 0; m < 32; m++) __builtin_nontemporal_store(keep + m, &o[m]);
    }
    asm volatile(
        "scratch_load_dwordx2 %16, %18, off\n\t"
        "global_load_dwordx2 %0, %19, off nt\n\tglobal_load_dwordx2 %1, %19, off offset:1024 nt\n\tglobal_load_dwordx2 %2, %19, off offset:2048 nt\n\t"
        "global_load_dwordx2 %3, %19, off offset:3072 nt\n\tglobal_load_dwordx2 %4, %20, off nt\n\tglobal_load_dwordx2 %5, %20, off offset:1024 nt\n\t"
        "global_load_dwordx2 %6, %20, off offset:2048 nt\n\tglobal_load_dwordx2 %7, %20, off offset:3072 nt\n\tglobal_load_dwordx2 %8, %21, off nt\n\t"
        "global_load_dwordx2 %9, %21, off offset:1024 nt\n\tglobal_load_dwordx2 %10, %21, off offset:2048 nt\n\tglobal_load_dwordx2 %11, %21, off offset:3072 nt\n\t"
        "global_load_dwordx2 %12, %22, off nt\n\tglobal_load_dwordx2 %13, %22, off offset:1024 nt\n\tglobal_load_dwordx2 %14, %22, off offset:2048 nt\n\t"
        "global_load_dwordx2 %15, %22, off offset:3072 nt\n\t"
        "s_waitcnt vmcnt(16)\n\t"
        "v_mov_b64 %17, %16\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]),
          "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]), "=&v"(s), "=&v"(c)
        : "v"(sp), "v"(p), "v"(p + 512), "v"(p + 1024), "v"(p + 1536)
        : "memory");
    if (c != tag) bad++;             // the copy taken behind the partial wait
    if (s != tag) bad += 1000;       // the register after the full wait (must always hold)
#pragma unroll
    for (int m = 0; m < 16; m++) keep += r[m];
  }
  out[(size_t)blockIdx.x * 128 + threadIdx.x] = keep;
  if (bad) atomicAdd(errors, bad);
}

int main() {
  const int units = 32768;
  uint64_t *in, *out;
  unsigned *err;
  if (hipMalloc(&in, (size_t)units * 2048 * 8) || hipMalloc(&out, (size_t)4096 * 128 * 32 * 8) || hipMalloc(&err, 4)) return 1;
  uint64_t *h = (uint64_t *)malloc((size_t)units * 2048 * 8);
  uint64_t z = 88172645463325252ull;
  for (size_t i = 0; i < (size_t)units * 2048; i++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = z; }
  (void)hipMemcpy(in, h, (size_t)units * 2048 * 8, hipMemcpyHostToDevice);
  for (int st = 0; st < 2; st++)
    for (int grid : {256, 1024, 4096}) {
      (void)hipMemset(err, 0, 4);
      for (int rep = 0; rep < 3; rep++) {
        if (st) hipLaunchKernelGGL(order2_kernel<true>, dim3(grid), dim3(128), 0, 0, in, out, err, units, 300);
        else hipLaunchKernelGGL(order2_kernel<false>, dim3(grid), dim3(128), 0, 0, in, out, err, units, 300);
      }
      unsigned e = 0;
      hipError_t rc = hipDeviceSynchronize();
      (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("scratch load, then 16 nt global loads, vmcnt(16); older nt stores=%d, grid %4d: %u stale copies (>= 1000: wrong after the full wait) (%s)\n", st, grid, e, hipGetErrorString(rc));
    }
  return 0;
}
