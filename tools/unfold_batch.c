/* tools/unfold_batch.c -- throughput of functional_bootstrap_batch with unfolded keys at TFHEpp-lvl2 parameters (drop-in API, host structs):
 *   gcc -O2 -Iinclude tools/unfold_batch.c -Lmosfhet_amd -lmosfhet_hip -lm ;  ./a.out [count]   (MOSFHET_HIP_UNFOLD_SPLIT_MAX / _BUDGET_GIB select the form) */
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>
#include <mosfhet.h>
static double now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + 1e-6 * tv.tv_usec; }
int main(int argc, char **argv) {
  const int n = 632, N = 2048, k = 1, l = 4, Bg_bit = 9, count = argc > 1 ? atoi(argv[1]) : 1024;
  TLWE_Key lk = tlwe_new_binary_key(n, 3.0517578125e-05);
  TRLWE_Key rk = trlwe_new_binary_key(N, k, 5.684341886080802e-14);
  TRGSW_Key gk = trgsw_new_key(rk, l, Bg_bit);
  TLWE_Key ok = tlwe_alloc_key(N, rk->sigma);
  trlwe_extract_tlwe_key(ok, rk);
  Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE *in = tlwe_alloc_sample_array(count, n), *out = tlwe_alloc_sample_array(count, N);
  for (int i = 0; i < count; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lk);
  for (int u = 1; u <= 8; u *= 2) {
    Bootstrap_Key bk = new_bootstrap_key(gk, lk, u);
    functional_bootstrap_batch(out, tv, in, count, bk, 4);
    double t0 = now();
    functional_bootstrap_batch(out, tv, in, count, bk, 4);
    double dt = now() - t0;
    int bad = 0;
    for (int i = 0; i < count; i++) { int64_t d = (int64_t)(tlwe_phase(out[i], ok) - lut[i % 4]); bad += (d < 0 ? -d : d) >= (1LL << 58); }
    printf("unfolding %d: %d bootstraps in %.1f ms = %.1f k/s, %d wrong\n", u, count, dt * 1e3, count / dt / 1e3, bad);
    free_bootstrap_key(bk);
  }
  return 0;
}
