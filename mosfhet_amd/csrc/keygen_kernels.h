// keygen_kernels.h -- on-device generation of the multi-gigabyte table-lookup TRLWE key-switch keys: the
// packing key of trlwe_new_packing1_KS_key (src/keyswitch.c:368-390) and the private key of trlwe_new_priv_SK_KS_key_N2
// (src/keyswitch.c:611-637).  Every row is a fresh TRLWE encryption under the binary key s_out:
//     a uniform,  b = a * s_out + e + message,  e ~ N(0, sigma) on the torus  (src/trlwe.c:296-316)
// Randomness (the reference: AES-CTR / SHAKE256 streams seeded from RDSEED, src/misc.c:34-91, and for seed-compressed rows a public per-row seed,
// src/trlwe_compressed_vaes.c:139-160):
//   * MASKS are public values and come from a counter-based generator on a PUBLIC 64-bit seed (keygen_mix: splitmix64 of (seed, row, word)), so that a
//     seed-compressed key can regenerate them inside the key-switch kernel for a few integer operations per word;
//   * NOISE is secret: it comes from ChaCha20 under a 256-bit key that is independent of the mask seed (NoiseKey: drawn from the operating system by the
//     C ABI, or handed over by the host layer from its own ChaCha20 stream), counter = (row, coefficient), nonce = a value that is UNIQUE PER GENERATE
//     CALL under that key (the library's call counter and a generator-kind tag: capi_ext.inc, noise_key_for_call) -- never the caller's mask seed.
//     Knowing every mask word and the mask seed says nothing about the noise terms, and two keys generated with the same public seed (two kinds of
//     table key, a plain and a Galois bootstrap key, the same key for two input keys ...) share their masks but not their noise, so the difference
//     of two such rows is still an encryption, not the bare key-dependent message.
// A 6 GB key takes milliseconds instead of minutes of host time.  The product a * s_out is exact (integer adds over the set bits of the key).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "negacyclic_fft.h"   // workgroup_sync()

namespace mosfhet {

__device__ __forceinline__ uint64_t keygen_mix(uint64_t seed, uint64_t row, uint64_t idx, uint64_t stream) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (row * 0x100000001B3ull + idx * 4 + stream + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct NoiseKey { uint32_t k[8]; uint64_t nonce; };   // nonce: unique per generate call under this key

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }
#define MOSFHET_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7)

// first four words of the ChaCha20 block (RFC 8439 core, 64-bit block counter + 64-bit nonce): two 64-bit uniforms
__device__ __forceinline__ void chacha20_uniforms(const NoiseKey &key, uint64_t counter, uint64_t nonce, uint64_t &u1, uint64_t &u2) {
  const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                          (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = s[i];
#pragma unroll 2
  for (int r = 0; r < 10; r++) {
    MOSFHET_QR(x[0], x[4], x[8], x[12]); MOSFHET_QR(x[1], x[5], x[9], x[13]); MOSFHET_QR(x[2], x[6], x[10], x[14]); MOSFHET_QR(x[3], x[7], x[11], x[15]);
    MOSFHET_QR(x[0], x[5], x[10], x[15]); MOSFHET_QR(x[1], x[6], x[11], x[12]); MOSFHET_QR(x[2], x[7], x[8], x[13]); MOSFHET_QR(x[3], x[4], x[9], x[14]);
  }
  u1 = (uint64_t)(x[0] + s[0]) | ((uint64_t)(x[1] + s[1]) << 32);
  u2 = (uint64_t)(x[2] + s[2]) | ((uint64_t)(x[3] + s[3]) << 32);
}

// Gaussian noise term of coefficient x of row r on the torus: Box-Muller on two uniforms (src/misc.c:87-91), double2torus of sigma z (src/misc.c:13-15)
__device__ __forceinline__ uint64_t keygen_noise(const NoiseKey &key, uint64_t row, uint64_t x, double sigma) {
  uint64_t a, b;
  chacha20_uniforms(key, (row << 16) | x, key.nonce, a, b);
  const double u1 = ((double)(a >> 11) + 0.5) * 0x1p-53, u2 = ((double)(b >> 11) + 0.5) * 0x1p-53;
  const double z = cos(6.283185307179586 * u1) * sqrt(-2.0 * log(u2)) * sigma;
  return (uint64_t)(int64_t)(18446744073709551616.0 * z);
}

// kind 0 (packing): rows (i < n, j < t, v in 1..2^bb-1), message = s_in[i] v 2^(64-(j+1)bb) on X^0
// kind 1 (private): rows (i <= n, ...), message polynomial = -s_out * (s_i v 2^(64-(j+1)bb)), s_n = -1
// kind 2 (LUT packing, trlwe_new_packing_KS_key, src/keyswitch.c:214-241): rows ((i, e) with i < n, e < slots; j; v), message = s_in[i] v 2^(64-(j+1)bb) on
//         the N / slots coefficients of slot e
// One workgroup of 256 threads per row; the mask lives in LDS while the key's set bits are walked.
__global__ __launch_bounds__(256) void trlwe_table_keygen_kernel(uint64_t *__restrict__ rows, const uint64_t *__restrict__ s_out,
                                                               const uint64_t *__restrict__ s_in, int n, int N, int t, int base_bit, double sigma,
                                                               uint64_t seed, int kind, size_t first_row, int compressed, NoiseKey nkey, int slots = 1) {
  extern __shared__ uint64_t sh[];   // a[N], then the indices of the set key bits (uint16) packed behind it
  uint64_t *a = sh;
  uint16_t *ones = reinterpret_cast<uint16_t *>(sh + N);
  int16_t *vals = reinterpret_cast<int16_t *>(ones + N);   // the key's nonzero coefficients (binary keys: all 1; bounded keys: small integers)
  __shared__ int n_ones, key_is_binary;
  const int tid = threadIdx.x;
  const size_t r = first_row + blockIdx.x;
  const int cands = (1 << base_bit) - 1;
  const int v = (int)(r % cands) + 1, j = (int)((r / cands) % t), i = (int)(r / ((size_t)cands * t));
  // compressed: only the b halves are stored ([rows][N]); the mask of row r is keygen_mix(seed, r, x, 0) again whenever it is needed
  uint64_t *dst = compressed ? rows + r * (size_t)N : rows + r * 2 * (size_t)N;
  uint64_t *dst_b = compressed ? dst : dst + N;
  if (tid == 0) {
    int c = 0, binary = 1;
    for (int x = 0; x < N; x++)
      if (s_out[x]) { vals[c] = (int16_t)(int64_t)s_out[x]; binary &= s_out[x] == 1; ones[c++] = (uint16_t)x; }
    n_ones = c;
    key_is_binary = binary;
  }
  for (int x = tid; x < N; x += 256) {
    const uint64_t ax = keygen_mix(seed, r, x, 0);
    a[x] = ax;
    if (!compressed) dst[x] = ax;
  }
  workgroup_sync();
  const int slot = kind == 2 ? i % slots : 0, span = N / slots;
  const uint64_t s_i = kind == 2 ? s_in[i / slots] : (i < n ? s_in[i] : ~0ull);
  const uint64_t dec = s_i * (uint64_t)v * (1ull << (64 - (j + 1) * base_bit));
  const int cnt = n_ones;
  for (int x = tid; x < N; x += 256) {
    uint64_t acc = 0;
    if (key_is_binary) {
      for (int q = 0; q < cnt; q++) {       // (a * s)[x] = sum over set bits p of +-a[x - p]  (negacyclic wrap)
        const int p = ones[q], src = x - p;
        const uint64_t w = a[src & (N - 1)];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
    } else {
      for (int q = 0; q < cnt; q++) {       // general small coefficients (tlwe / trlwe_new_bounded_key)
        const int p = ones[q], src = x - p;
        const uint64_t w = a[src & (N - 1)] * (uint64_t)(int64_t)vals[q];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
    }
    acc += keygen_noise(nkey, r, (uint64_t)x, sigma);
    if (kind == 0) { if (x == 0) acc += dec; }
    else if (kind == 2) { if (x / span == slot) acc += dec; }
    else acc += ((uint64_t)0 - s_out[x]) * dec;
    dst_b[x] = acc;
  }
}

// Bootstrap key in the torus domain, BK_i = TRGSW(s_i) (new_bootstrap_key, src/bootstrap.c:14-18) or, ga != 0, TRGSW(X^{s_i}) (new_bootstrap_key_ga,
// src/bootstrap_ga.c:17-20): row r = i * 2l + q, q = c * l + j, is a fresh TRLWE(0) with m * 2^(64 - (j+1) Bg) added to coefficient e of component c
// (trgsw_monomial_sample, src/trgsw.c:152-168; k = 1).  One workgroup per row, same generator and exact a * s as the table keys above.
__global__ __launch_bounds__(256) void trgsw_bk_keygen_kernel(uint64_t *__restrict__ rows, const uint64_t *__restrict__ s_out, const uint64_t *__restrict__ s_in,
                                                            int N, int l, int Bg_bit, double sigma, uint64_t seed, int ga, NoiseKey nkey) {
  extern __shared__ uint64_t sh[];
  uint64_t *a = sh;
  uint16_t *ones = reinterpret_cast<uint16_t *>(sh + N);
  int16_t *vals = reinterpret_cast<int16_t *>(ones + N);   // the key's nonzero coefficients (binary keys: all 1; bounded keys: small integers)
  __shared__ int n_ones, key_is_binary;
  const int tid = threadIdx.x;
  const size_t r = blockIdx.x;
  const int q = (int)(r % (2 * l)), c = q / l, j = q % l;
  const size_t i = r / (2 * l);
  uint64_t *dst = rows + r * 2 * (size_t)N;
  if (tid == 0) {
    int cnt = 0, binary = 1;
    for (int x = 0; x < N; x++)
      if (s_out[x]) { vals[cnt] = (int16_t)(int64_t)s_out[x]; binary &= s_out[x] == 1; ones[cnt++] = (uint16_t)x; }
    n_ones = cnt;
    key_is_binary = binary;
  }
  for (int x = tid; x < N; x += 256) a[x] = keygen_mix(seed, r, x, 0);
  workgroup_sync();
  const uint64_t h = 1ull << (64 - (j + 1) * Bg_bit);
  // ga: TRGSW(X^{s_i}) for ANY integer key coefficient (bounded keys, src/bootstrap_ga.c:17-20 with tlwe_new_bounded_key): exponent mod 2N, X^N = -1
  const int e_full = ga ? (int)(s_in[i] & (uint64_t)(2 * N - 1)) : 0;
  const int e = e_full & (N - 1);
  const uint64_t val = ga ? ((e_full & N) ? (uint64_t)0 - h : h) : s_in[i] * h;
  const int cnt = n_ones;
  for (int x = tid; x < N; x += 256) {
    uint64_t acc = 0;
    if (key_is_binary) {
      for (int k = 0; k < cnt; k++) {
        const int p = ones[k], src = x - p;
        const uint64_t w = a[src & (N - 1)];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
    } else {
      for (int k = 0; k < cnt; k++) {
        const int p = ones[k], src = x - p;
        const uint64_t w = a[src & (N - 1)] * (uint64_t)(int64_t)vals[k];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
    }
    acc += keygen_noise(nkey, r, (uint64_t)x, sigma);
    dst[x] = a[x] + ((c == 0 && x == e) ? val : 0);     // the gadget goes on the mask AFTER b = a * s + e was formed from the plain mask
    dst[N + x] = acc + ((c == 1 && x == e) ? val : 0);
  }
}

// The same for ANY k >= 1 and power-of-two N (keys of the general-ring path): row r = i (k+1) l + c l + j is a fresh TRLWE_k(0) -- masks a_0 .. a_{k-1} uniform (mask m of a
// row: generator stream m), b = sum_m a_m * s_m + e, every product exact -- with s_lwe[i] 2^(64 - (j+1) Bg) added to coefficient 0 of component c <= k
// (trgsw_monomial_sample, src/trgsw.c:152-168 loops over the k + 1 components; new_bootstrap_key, src/bootstrap.c:14-18).  One workgroup per row, one mask at a time in LDS.
__global__ __launch_bounds__(256) void trgsw_bk_keygen_k_kernel(uint64_t *__restrict__ rows, const uint64_t *__restrict__ s_out /*[k][N]*/, const uint64_t *__restrict__ s_in,
                                                              int k, int N, int l, int Bg_bit, double sigma, uint64_t seed, NoiseKey nkey) {
  extern __shared__ uint64_t sh[];
  uint64_t *a = sh;
  uint16_t *ones = reinterpret_cast<uint16_t *>(sh + N);
  int16_t *vals = reinterpret_cast<int16_t *>(ones + N);
  __shared__ int n_ones;
  const int tid = threadIdx.x;
  const size_t r = blockIdx.x;
  const int per = (k + 1) * l, q = (int)(r % per), c = q / l, j = q % l;
  const size_t i = r / per;
  uint64_t *dst = rows + r * (size_t)(k + 1) * N, *dst_b = dst + (size_t)k * N;
  const uint64_t val = s_in[i] * (1ull << (64 - (j + 1) * Bg_bit));
  for (int m = 0; m < k; m++) {
    workgroup_sync();   // the previous mask's LDS image is consumed
    const uint64_t *s_m = s_out + (size_t)m * N;
    if (tid == 0) {
      int cnt = 0;
      for (int x = 0; x < N; x++)
        if (s_m[x]) { vals[cnt] = (int16_t)(int64_t)s_m[x]; ones[cnt++] = (uint16_t)x; }
      n_ones = cnt;
    }
    for (int x = tid; x < N; x += 256) a[x] = keygen_mix(seed, r, x, (uint64_t)m);
    workgroup_sync();
    const int cnt = n_ones;
    for (int x = tid; x < N; x += 256) {   // (a thread owns the same words x in every pass: the read-modify-write of dst_b needs no barrier)
      uint64_t acc = m ? dst_b[x] : 0;
      for (int u = 0; u < cnt; u++) {
        const int p = ones[u], src = x - p;
        const uint64_t w = a[src & (N - 1)] * (uint64_t)(int64_t)vals[u];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
      dst[(size_t)m * N + x] = a[x] + ((c == m && x == 0) ? val : 0);     // the gadget goes on the mask AFTER b was formed from the plain mask
      dst_b[x] = acc;
    }
  }
  for (int x = tid; x < N; x += 256) dst_b[x] += keygen_noise(nkey, r, (uint64_t)x, sigma) + ((c == k && x == 0) ? val : 0);
}

// FFT-based TRLWE key-switch keys in the torus domain (trlwe_new_KS_key, src/keyswitch.c:12-37; the automorphism key set, :500-511; the private
// pair, :39-50; the relinearisation key, :3-10 -- they differ only in the polynomial being switched from): entry e, row r < t is
// TRLWE_{s_out}(msg_e(X) * 2^(64 - (r+1) bb)).  One workgroup per row; the caller transforms the rows afterwards.
__global__ __launch_bounds__(256) void trlwe_poly_keygen_kernel(uint64_t *__restrict__ rows, const uint64_t *__restrict__ s_out, const uint64_t *__restrict__ msgs,
                                                              int N, int t, int base_bit, double sigma, uint64_t seed, NoiseKey nkey) {
  extern __shared__ uint64_t sh[];
  uint64_t *a = sh;
  uint16_t *ones = reinterpret_cast<uint16_t *>(sh + N);
  int16_t *vals = reinterpret_cast<int16_t *>(ones + N);   // the key's nonzero coefficients (binary keys: all 1; bounded keys: small integers)
  __shared__ int n_ones, key_is_binary;
  const int tid = threadIdx.x;
  const size_t r = blockIdx.x;
  const int j = (int)(r % t);
  const size_t e = r / t;
  uint64_t *dst = rows + r * 2 * (size_t)N;
  if (tid == 0) {
    int cnt = 0, binary = 1;
    for (int x = 0; x < N; x++)
      if (s_out[x]) { vals[cnt] = (int16_t)(int64_t)s_out[x]; binary &= s_out[x] == 1; ones[cnt++] = (uint16_t)x; }
    n_ones = cnt;
    key_is_binary = binary;
  }
  for (int x = tid; x < N; x += 256) {
    const uint64_t ax = keygen_mix(seed, r, x, 0);
    a[x] = ax;
    dst[x] = ax;
  }
  workgroup_sync();
  const int shift = 64 - (j + 1) * base_bit, cnt = n_ones;
  for (int x = tid; x < N; x += 256) {
    uint64_t acc = 0;
    if (key_is_binary) {
      for (int k = 0; k < cnt; k++) {
        const int p = ones[k], src = x - p;
        const uint64_t w = a[src & (N - 1)];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
    } else {
      for (int k = 0; k < cnt; k++) {
        const int p = ones[k], src = x - p;
        const uint64_t w = a[src & (N - 1)] * (uint64_t)(int64_t)vals[k];
        acc += src < 0 ? (uint64_t)0 - w : w;
      }
    }
    acc += keygen_noise(nkey, r, (uint64_t)x, sigma);
    dst[N + x] = acc + (msgs[e * (size_t)N + x] << shift);
  }
}

// rows [first_row, first_row + gridDim.x) of a seed-compressed table key in full (`row` words each): the first mask_words regenerated, the
// rest copied from the stored b part ([rows][row - mask_words])
__global__ __launch_bounds__(256) void table_expand_kernel(uint64_t *__restrict__ out, const uint64_t *__restrict__ b_rows, int row, int mask_words,
                                                         uint64_t seed, size_t first_row) {
  const size_t r = first_row + blockIdx.x;
  uint64_t *dst = out + (size_t)blockIdx.x * row;
  for (int x = threadIdx.x; x < row; x += 256)
    dst[x] = x < mask_words ? keygen_mix(seed, r, x, 0) : b_rows[r * (size_t)(row - mask_words) + (x - mask_words)];
}

// LWE -> LWE key-switch table (tlwe_new_KS_key, src/tlwe.c:193-212): row (i, j, v) = TLWE_{s_out}(s_in[i] v 2^(64 - (j+1) bb)), a uniform from the
// counter-based generator, b = <a, s_out> + e + message.  One wavefront per row.  compressed: only b is stored ([rows] words).
__global__ __launch_bounds__(64) void tlwe_ksk_keygen_kernel(uint64_t *__restrict__ rows, const uint64_t *__restrict__ s_out, const uint64_t *__restrict__ s_in,
                                                            int n_out, int t, int base_bit, double sigma, uint64_t seed, size_t first_row, int compressed, NoiseKey nkey) {
  const size_t r = first_row + blockIdx.x;
  const int cands = (1 << base_bit) - 1, lane = threadIdx.x;
  const int v = (int)(r % cands) + 1, j = (int)((r / cands) % t);
  const size_t i = r / ((size_t)cands * t);
  uint64_t *dst = compressed ? rows + r : rows + r * (size_t)(n_out + 1);
  uint64_t acc = 0;
  for (int x = lane; x < n_out; x += 64) {
    const uint64_t ax = keygen_mix(seed, r, x, 0);
    if (!compressed) dst[x] = ax;
    acc += ax * s_out[x];
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) {
    acc += keygen_noise(nkey, r, 0, sigma);
    acc += s_in[i] * (uint64_t)v * (1ull << (64 - (j + 1) * base_bit));
    dst[compressed ? 0 : n_out] = acc;
  }
}

}  // namespace mosfhet
