// ksw_ab.hip -- same-box A/B of the two table key-switch forms (keyswitch_kernels.h: ciphertexts on the lanes, LDS gather; keyswitch_words_kernels.h: output
// words on the lanes, wave-uniform digits), every output word compared, both timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mosfhet_amd/csrc -o tools/ab/_build/ksw_ab tools/ab/ksw_ab.hip
//   tools/ab/_build/ksw_ab [count n_in row b_word t base_bit compressed]...      (no arguments: the shapes of the BASELINE configs and a few odd ones)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

#include "keyswitch_kernels.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace mosfhet;

__global__ void fill_kernel(uint64_t *p, size_t n, uint64_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = keygen_mix(seed, i >> 20, i & 0xfffff, 3);
}

__global__ void count_diff_kernel(const uint64_t *a, const uint64_t *b, size_t n, unsigned long long *diff) {
  unsigned long long d = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d += a[i] != b[i];
  if (d) atomicAdd(diff, d);
}

static int g_soak = 0;   // ksw_ab soak <launches> <case...>: the word-lane kernel repeated, every launch's output compared on the device with the tiles' output

static float time_ms(hipStream_t s, int reps, const std::function<void()> &f) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f();
  CHECK(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int g = 0; g < 3; g++) {
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) f();
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / reps < best) best = ms / reps;
  }
  return best;
}

static int run_case(int count, int n_in, int row, int b_word, int t, int bb, int compressed) {
  const int cands = (1 << bb) - 1;
  const int mask_words = !compressed ? 0 : (b_word == row - 1 ? row - 1 : row / 2);
  const size_t rows = (size_t)n_in * t * cands, key_words = rows * (size_t)(row - mask_words);
  const int in_words = n_in + (b_word >= 0 ? 1 : 0);
  uint64_t *ksk, *in, *out_a, *out_b;
  CHECK(hipMalloc(&ksk, key_words * 8 + ksw_slack_bytes(row)));   // (the word-lane form reads past the last row: keyswitch_words_kernels.h)
  CHECK(hipMalloc(&in, (size_t)count * in_words * 8));
  CHECK(hipMalloc(&out_a, (size_t)count * row * 8));
  CHECK(hipMalloc(&out_b, (size_t)count * row * 8));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, ksk, key_words, 11ull);
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s, in, (size_t)count * in_words, 12ull);
  CHECK(hipMemsetAsync(out_a, 0xAB, (size_t)count * row * 8, s));
  CHECK(hipMemsetAsync(out_b, 0xCD, (size_t)count * row * 8, s));
  KsWorkspace wa, wb;
  const uint64_t seed = 0x5EED;
  auto tiles = [&]() {
    if (bb >= 3 && (count > 256 || bb > 4))
      CHECK(launch_tlwe_keyswitch_nw<8>(ksk, out_a, row, in, in_words, count, n_in, row, b_word, t, bb, wa, s, compressed, seed, mask_words));
    else
      CHECK(launch_tlwe_keyswitch_nw<KS_NW>(ksk, out_a, row, in, in_words, count, n_in, row, b_word, t, bb, wa, s, compressed, seed, mask_words));
  };
  auto direct = [&]() { CHECK(launch_tlwe_keyswitch_small(ksk, out_a, row, in, in_words, count, n_in, row, b_word, t, bb, wa, s, compressed, seed, mask_words)); };
  auto words = [&]() { CHECK(launch_tlwe_keyswitch_words(ksk, out_b, row, in, in_words, count, n_in, row, b_word, t, bb, wb, s, compressed, seed, mask_words)); };
  const int reps = (size_t)count * n_in * t * row > (1ull << 34) ? 5 : 20;
  const bool use_direct = getenv("KSW_AB_DIRECT") != nullptr;      // compare with the direct (row-gather) kernels of <= 16 ciphertexts instead of the tiles
  const float ms_a = use_direct ? time_ms(s, reps, direct) : time_ms(s, reps, tiles), ms_b = time_ms(s, reps, words);
  std::vector<uint64_t> a((size_t)count * row), b((size_t)count * row);
  CHECK(hipMemcpy(a.data(), out_a, a.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(b.data(), out_b, b.size() * 8, hipMemcpyDeviceToHost));
  size_t bad = 0, first = (size_t)-1;
  for (size_t i = 0; i < a.size(); i++)
    if (a[i] != b[i]) { if (!bad) first = i; bad++; }
  if (g_soak) {
    unsigned long long *d_diff, h_diff = 0;
    CHECK(hipMalloc(&d_diff, 8));
    CHECK(hipMemsetAsync(d_diff, 0, 8, s));
    for (int r = 0; r < g_soak; r++) {
      CHECK(hipMemsetAsync(out_b, 0x5A, (size_t)count * row * 8, s));
      words();
      hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, s, out_a, out_b, (size_t)count * row, d_diff);
    }
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(&h_diff, d_diff, 8, hipMemcpyDeviceToHost));
    unsigned int gave_up = 0;
    CHECK(hipMemcpyFromSymbol(&gave_up, HIP_SYMBOL(ksw_gave_up_count), sizeof(gave_up)));
    printf("soak: %d launches of the word-lane kernel, %llu differing words in total, %u wavefronts gave up a wait\n", g_soak, h_diff, gave_up);
    (void)hipFree(d_diff);
  }
  const KsWordsPlan p = ks_words_plan(count, n_in, row, t, bb);
  printf("count=%5d n_in=%5d row=%5d b_word=%5d t=%2d bb=%d %s  tiles %8.3f ms  words %8.3f ms  (x%.2f; JB=%d groups=%d wblocks=%d splits=%d lds=%zu)  differing words: %zu%s\n",
         count, n_in, row, b_word, t, bb, compressed ? "compressed" : "plain     ", ms_a, ms_b, ms_a / ms_b, p.JB, p.groups, p.wblocks, p.splits, p.lds, bad,
         bad ? " <-- MISMATCH" : "");
  if (bad) printf("   first at ciphertext %zu word %zu: tiles %016llx words %016llx\n", first / row, first % row, (unsigned long long)a[first], (unsigned long long)b[first]);
  fflush(stdout);
  (void)hipFree(ksk); (void)hipFree(in); (void)hipFree(out_a); (void)hipFree(out_b);
  if (wa.inT) (void)hipFree(wa.inT);
  if (wa.outT) (void)hipFree(wa.outT);
  if (wb.inT) (void)hipFree(wb.inT);
  if (wb.outT) (void)hipFree(wb.outT);
  (void)hipStreamDestroy(s);
  return bad != 0;
}

int main(int argc, char **argv) {
  int fails = 0;
  if (argc >= 3 && !strcmp(argv[1], "soak")) {
    g_soak = atoi(argv[2]);
    argc -= 2;
    argv += 2;
  }
  if (argc >= 8) {
    for (int k = 1; k + 6 < argc; k += 7)
      fails += run_case(atoi(argv[k]), atoi(argv[k + 1]), atoi(argv[k + 2]), atoi(argv[k + 3]), atoi(argv[k + 4]), atoi(argv[k + 5]), atoi(argv[k + 6]));
    return fails != 0;
  }
  // small and odd shapes first (a wrong kernel shows here in milliseconds)
  fails += run_case(17, 16, 17, 16, 2, 2, 0);
  fails += run_case(70, 40, 100, 99, 3, 3, 0);
  fails += run_case(129, 64, 130, -1, 4, 4, 0);
  fails += run_case(100, 64, 128, 64, 5, 2, 1);
  fails += run_case(600, 100, 586, 585, 5, 2, 1);
  fails += run_case(64, 32, 256, 128, 7, 4, 1);
  fails += run_case(300, 50, 200, 199, 20, 2, 0);
  if (fails) return 1;
  // SET_1 LWE switch (gate), lvl2 LWE switch (FDFB), packing switch (config 3): 128 (one GPU's share), 1024, 4096
  fails += run_case(4096, 1024, 586, 585, 5, 2, 0);
  fails += run_case(128, 2048, 633, 632, 8, 4, 0);
  fails += run_case(1024, 2048, 633, 632, 8, 4, 0);
  fails += run_case(4096, 2048, 633, 632, 8, 4, 0);
  fails += run_case(1024, 2048, 633, 632, 8, 4, 1);
  fails += run_case(128, 2048, 4096, 2048, 6, 4, 0);
  fails += run_case(512, 2048, 4096, 2048, 6, 4, 0);
  fails += run_case(1024, 2048, 4096, 2048, 6, 4, 0);
  fails += run_case(1024, 2048, 4096, 2048, 6, 4, 1);
  fails += run_case(512, 2048, 4096, 2048, 6, 4, 1);
  fails += run_case(4096, 2048, 4096, 2048, 6, 4, 1);
  return fails != 0;
}
