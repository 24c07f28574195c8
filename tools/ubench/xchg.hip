// xchg.hip -- the per-step exchange of a bootstrap split over TWO workgroups (pbs_split_kernel): every step each workgroup sends 16 KiB of partial DFT sums to
// its partner and receives 16 KiB, with NO flag: the receive slots hold a sentinel (a NaN pattern no fma chain on finite numbers produces), the sender's lanes
// store their 16-byte items with device-scope stores (sc1), the receiver's lanes poll their own items with device-scope loads until both halves differ from the
// sentinel, then put the sentinel back.  Two slot sets alternate by step parity.  Measures the cost per step (both directions at once, as the kernel does it),
// same XCD and different XCDs, and checks every received word.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/xchg tools/ubench/xchg.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double __attribute__((ext_vector_type(2))) d2;
constexpr uint64_t SENTINEL = 0xFFF7A5C3DEADBEEFull;
constexpr int T = 128, M = 1024;

// (s_nop 1: a store of more than 8 bytes reads its data registers late; the VALU instruction behind it must not overwrite them for two wait states, and the
// compiler's hazard recognizer does not look into inline assembly -- without it lanes 12-15 of every 16 sent the NEXT item's half-computed real part)
#ifndef ST_BITS
#define ST_BITS "sc1"
#endif
#ifndef LD_BITS
#define LD_BITS "sc1"
#endif
__device__ __forceinline__ void store_dev(d2 *p, d2 v) { asm volatile("global_store_dwordx4 %0, %1, off " ST_BITS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ d2 load_dev(const d2 *p) {
  d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// a finite double with all 52 mantissa bits in use (the first version sent small integers: their low dwords are zero and a torn item went unnoticed)
__device__ __forceinline__ double payload(int r, int m, int t, int h, int c) {
  uint64_t x = (uint64_t)r * 0x9E3779B97F4A7C15ull + (uint64_t)(m * 131 + t * 8 + h * 2 + c + 1) * 0xC2B2AE3D27D4EB4Full;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  return __builtin_bit_cast(double, (x & 0x800FFFFFFFFFFFFFull) | 0x4330000000000000ull);
}

// the lane's eight items (stride T) requested together, one wait
__device__ __forceinline__ void load8_dev(d2 (&v)[8], const d2 *p) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off " LD_BITS "\n\t"
      "global_load_dwordx4 %1, %8, off offset:2048 " LD_BITS "\n\t"
      "global_load_dwordx4 %2, %9, off " LD_BITS "\n\t"
      "global_load_dwordx4 %3, %9, off offset:2048 " LD_BITS "\n\t"
      "global_load_dwordx4 %4, %10, off " LD_BITS "\n\t"
      "global_load_dwordx4 %5, %10, off offset:2048 " LD_BITS "\n\t"
      "global_load_dwordx4 %6, %11, off " LD_BITS "\n\t"
      "global_load_dwordx4 %7, %11, off offset:2048 " LD_BITS "\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p), "v"(p + 2 * T), "v"(p + 4 * T), "v"(p + 6 * T)
      : "memory");
}

// buf[2 sides][2 parities][M]: side s = the receive slots of workgroup s
__global__ __launch_bounds__(256) void k_xchg(d2 *buf, int a, int b, int rounds, int work, unsigned long long *errors, double *sink) {
  const int h = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (h < 0) return;
  const int team = threadIdx.x / T, t = threadIdx.x % T;
  const d2 sent = d2{__builtin_bit_cast(double, SENTINEL), __builtin_bit_cast(double, SENTINEL)};
  unsigned long long bad = 0;
  double acc = 1.0;
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < work; i++) acc = __builtin_fma(acc, 1.0000001, 1e-9);   // stands for the step's transforms
    const int par = r & 1;
    if (team != h) {
      d2 *dst = buf + ((size_t)(1 - h) * 2 + par) * M;
#pragma unroll
      for (int m = 0; m < 8; m++) store_dev(dst + m * T + t, d2{payload(r, m, t, h, 0), payload(r, m, t, h, 1)});
    } else {
      d2 *src = buf + ((size_t)h * 2 + par) * M;
      d2 v[8];
      bool all;
      int polls = 0;
      do {
        if (++polls > 200000) { bad += 1ull << 40; break; }   // (a variant whose loads never see the partner's stores: give up, counted in the last field)
        all = true;
        load8_dev(v, src + t);
#pragma unroll
        for (int m = 0; m < 8; m++) {   // (element copies first: __builtin_bit_cast of a vector ELEMENT reads element 0 whichever is named -- clang 22)
          const double vx = v[m].x, vy = v[m].y;
          all = all && __builtin_bit_cast(uint64_t, vx) != SENTINEL && __builtin_bit_cast(uint64_t, vy) != SENTINEL;
        }
      } while (!all);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const double vx = v[m].x, vy = v[m].y;
        const uint64_t gx = __builtin_bit_cast(uint64_t, vx), gy = __builtin_bit_cast(uint64_t, vy);
        const uint64_t wx = __builtin_bit_cast(uint64_t, payload(r, m, t, 1 - h, 0)), wy = __builtin_bit_cast(uint64_t, payload(r, m, t, 1 - h, 1));
        if (gx != wx || gy != wy) {
          bad++;
          if (r < 2) {
            const unsigned long long slot = atomicAdd(errors + 1, 1ull);
            if (slot < 16) { errors[2 + 4 * slot] = ((unsigned long long)r << 32) | (h << 16) | (m << 8) | t; errors[3 + 4 * slot] = gx; errors[4 + 4 * slot] = wx; errors[5 + 4 * slot] = gy ^ wy; }
          }
          if ((uint32_t)gx == (uint32_t)SENTINEL || (uint32_t)gy == (uint32_t)SENTINEL) bad += 1ull << 20;          // a low dword still holds the sentinel's
          if ((gx >> 32) == (SENTINEL >> 32) || (gy >> 32) == (SENTINEL >> 32)) bad += 1ull << 40;                  // a high dword does
        }
        store_dev(src + m * T + t, sent);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the reset is at its coherence point before this workgroup's next send can be observed
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void k_fill(uint64_t *p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = SENTINEL;
}

int main() {
  d2 *buf;
  double *sink;
  unsigned long long *errors, h_err = 0;
  const size_t words = (size_t)2 * 2 * M * 2;
  CHECK(hipMalloc(&buf, words * 8));
  CHECK(hipMalloc(&sink, 64 * 256 * 8));
  CHECK(hipMalloc(&errors, 8 * 80));
  CHECK(hipMemset(errors, 0, 8 * 80));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int rounds = 20000;
  for (int work : {0, 2000})
    for (int b : {8, 1, 4}) {
      hipLaunchKernelGGL(k_fill, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, nullptr, (uint64_t *)buf, words);
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_xchg, dim3(16), dim3(256), 0, nullptr, buf, 0, b, rounds, work, errors, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(&h_err, errors, 8, hipMemcpyDeviceToHost));
      printf("workgroups 0 and %d (%s), %d fma of stand-in work per step: %.2f us per step (16 KiB each way, sentinel slots), wrong items so far %llu (of them with a sentinel low dword %llu, high dword %llu)\n", b,
             b % 8 == 0 ? "same XCD if round-robin" : "different XCDs", work, ms * 1e3 / rounds, h_err & 0xFFFFF, (h_err >> 20) & 0xFFFFF, h_err >> 40);
    }
  unsigned long long rec[80];
  CHECK(hipMemcpy(rec, errors, sizeof(rec), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16 && i < (int)rec[1]; i++)
    printf("  mismatch: round %llu side %llu item %llu lane %llu: got x %016llx want %016llx, y xor %016llx\n", rec[2 + 4 * i] >> 32, (rec[2 + 4 * i] >> 16) & 0xFFFF, (rec[2 + 4 * i] >> 8) & 0xFF,
           rec[2 + 4 * i] & 0xFF, rec[3 + 4 * i], rec[4 + 4 * i], rec[5 + 4 * i]);
  return 0;
}
