// keyswitch_kernels.h -- batched LWE -> LWE key switch by table lookup (gfx950).
//
// Replaces tlwe_keyswitch [src/tlwe.c:289-303] for a batch:
//   out = (0, ..., 0, in.b);  for i < n_in, j < t:  v = ((in.a[i] + 2^(63 - t bb)) >> (64 - (j+1) bb)) & (2^bb - 1);
//   if v != 0:  out -= KS[i][j][v-1]            (rows of n_out + 1 Torus words, exact mod 2^64)
//
// Pure integer, HBM/L2-bound gather: a ciphertext touches up to n_in * t rows (24 MB at SET_1) of a
// 72 MB table.  One workgroup switches G ciphertexts together: thread c owns output word(s) c of all G
// accumulators (registers), the digit of every (ciphertext, i, j) is wave-uniform (scalar unit), and each
// selected row is read with one fully coalesced sweep.  Loads are issued unconditionally (digit 0 reads row 0
// and is masked afterwards) so that G * t independent row reads are in flight per i; ciphertexts of a tile
// that pick the same row hit in the CU's L1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mosfhet {

constexpr int KS_THREADS = 256;
constexpr int KS_MAXC = 4;  // output words per thread -> n_out + 1 <= 1024

template <int G>
__global__ __launch_bounds__(KS_THREADS) void tlwe_keyswitch_kernel(const uint64_t *__restrict__ ksk, uint64_t *__restrict__ out,
                                                                    const uint64_t *__restrict__ in, int count, int n_in,
                                                                    int n_out, int t, int base_bit) {
  const int row = n_out + 1;
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * G;
  const int per_j = (1 << base_bit) - 1;
  const uint64_t round_off = 1ull << (63 - base_bit * t);
  const uint64_t mask = (1ull << base_bit) - 1;

  uint64_t acc[G][KS_MAXC];
#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int q = 0; q < KS_MAXC; q++) acc[g][q] = 0;

  for (int i = 0; i < n_in; i++) {
    uint64_t ai[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int b = b0 + g < count ? b0 + g : count - 1;  // tail tile: recompute the last ciphertext, never stored
      ai[g] = in[(size_t)b * (n_in + 1) + i] + round_off;
    }
    const uint64_t *__restrict__ ki = ksk + (size_t)i * t * per_j * row;
    for (int j = 0; j < t; j++) {
      const int shift = 64 - (j + 1) * base_bit;
#pragma unroll
      for (int g = 0; g < G; g++) {
        const uint32_t v = (uint32_t)((ai[g] >> shift) & mask);
        const uint64_t *__restrict__ r = ki + ((size_t)j * per_j + (v ? v - 1 : 0)) * row;
        const uint64_t keep = v ? ~0ull : 0ull;
#pragma unroll
        for (int q = 0; q < KS_MAXC; q++) {
          const int c = tid + q * KS_THREADS;
          if (c < row) acc[g][q] -= r[c] & keep;
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; g++) {
    if (b0 + g >= count) break;
    uint64_t *dst = out + (size_t)(b0 + g) * row;
#pragma unroll
    for (int q = 0; q < KS_MAXC; q++) {
      const int c = tid + q * KS_THREADS;
      if (c < n_out) dst[c] = acc[g][q];
      else if (c == n_out) dst[c] = acc[g][q] + in[(size_t)(b0 + g) * (n_in + 1) + n_in];
    }
  }
}

inline void launch_tlwe_keyswitch(const uint64_t *ksk, uint64_t *out, const uint64_t *in, int count, int n_in, int n_out, int t,
                                  int base_bit, hipStream_t s) {
  constexpr int G = 8;
  const int blocks = (count + G - 1) / G;
  hipLaunchKernelGGL((tlwe_keyswitch_kernel<G>), dim3(blocks), dim3(KS_THREADS), 0, s, ksk, out, in, count, n_in, n_out, t, base_bit);
}

}  // namespace mosfhet
