// vmcnt_order.hip -- do a wave's global loads land in issue order, so that `s_waitcnt vmcnt(N)` after a burst of K loads guarantees the K - N oldest?
// (Round-3 diagnosis, experiments/README.md.)  16 non-temporal 8-byte loads of one 16 KiB unit, partial wait, copy the registers the wait covers, full
// wait, compare.  Variants: a scratch store of the covered registers right behind the partial wait (what the compiler's spill code did).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KEEP, bool SCRATCH>
__global__ __launch_bounds__(128, 2) void order_kernel(const uint64_t *__restrict__ in, unsigned *errors, int units, int iters) {
  __shared__ uint64_t lds[2304];
  unsigned bad = 0;
  volatile uint64_t priv[4];
  for (int it = 0; it < iters; it++) {
    const uint64_t *p = in + ((size_t)((blockIdx.x * 2654435761u + it * 40503u) % (unsigned)units)) * 2048 + threadIdx.x;
    uint64_t r[16], c[3];
    asm volatile(
        "global_load_dwordx2 %0, %19, off nt\n\tglobal_load_dwordx2 %1, %19, off offset:1024 nt\n\tglobal_load_dwordx2 %2, %19, off offset:2048 nt\n\t"
        "global_load_dwordx2 %3, %19, off offset:3072 nt\n\tglobal_load_dwordx2 %4, %20, off nt\n\tglobal_load_dwordx2 %5, %20, off offset:1024 nt\n\t"
        "global_load_dwordx2 %6, %20, off offset:2048 nt\n\tglobal_load_dwordx2 %7, %20, off offset:3072 nt\n\tglobal_load_dwordx2 %8, %21, off nt\n\t"
        "global_load_dwordx2 %9, %21, off offset:1024 nt\n\tglobal_load_dwordx2 %10, %21, off offset:2048 nt\n\tglobal_load_dwordx2 %11, %21, off offset:3072 nt\n\t"
        "global_load_dwordx2 %12, %22, off nt\n\tglobal_load_dwordx2 %13, %22, off offset:1024 nt\n\tglobal_load_dwordx2 %14, %22, off offset:2048 nt\n\t"
        "global_load_dwordx2 %15, %22, off offset:3072 nt\n\t"
        "s_waitcnt vmcnt(%23)\n\t"
        "v_mov_b64 %16, %0\n\tv_mov_b64 %17, %1\n\tv_mov_b64 %18, %2\n\t"     // low dwords of the three oldest loads, copied right behind the partial wait
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]),
          "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2])
        : "v"(p), "v"(p + 512), "v"(p + 1024), "v"(p + 1536), "i"(KEEP)
        : "memory");
    if constexpr (SCRATCH) {
      priv[it & 3] = c[1];
      priv[(it + 1) & 3] = c[2];
    }
    lds[threadIdx.x] = r[5];
    __syncthreads();
    if (c[0] != r[0] || c[1] != r[1] || c[2] != r[2]) bad++;
    if constexpr (SCRATCH) {
      if (priv[it & 3] != r[1] || priv[(it + 1) & 3] != r[2]) bad++;
    }
    uint64_t s = lds[(threadIdx.x + 1) & 127];
#pragma unroll
    for (int m = 3; m < 16; m++) s += r[m];
    __syncthreads();
    if (s == 0x123456789ull) bad++;   // keep everything live
  }
  if (bad) atomicAdd(errors, bad);
}

int main() {
  const int units = 32768;   // 512 MiB: beyond the Infinity Cache
  uint64_t *in;
  unsigned *err;
  hipMalloc(&in, (size_t)units * 2048 * 8);
  hipMalloc(&err, 4);
  uint64_t *h = (uint64_t *)malloc((size_t)units * 2048 * 8);
  uint64_t z = 88172645463325252ull;
  for (size_t i = 0; i < (size_t)units * 2048; i++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = z; }
  hipMemcpy(in, h, (size_t)units * 2048 * 8, hipMemcpyHostToDevice);
  for (int sc = 0; sc < 2; sc++)
    for (int grid : {256, 1024, 4096}) {
      hipMemset(err, 0, 4);
      for (int rep = 0; rep < 3; rep++) {
        if (sc) hipLaunchKernelGGL((order_kernel<13, true>), dim3(grid), dim3(128), 0, 0, in, err, units, 400);
        else hipLaunchKernelGGL((order_kernel<13, false>), dim3(grid), dim3(128), 0, 0, in, err, units, 400);
      }
      unsigned e = 0;
      hipError_t rc = hipDeviceSynchronize();
      hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("vmcnt(13) after 16 nt loads, scratch=%d, grid %4d: %u registers read before they landed (%s)\n", sc, grid, e, hipGetErrorString(rc));
    }
  return 0;
}
