/* single-call latency through the compat API: gcc -O2 -Iinclude tools/compat_latency.c -Lmosfhet_amd -lmosfhet_hip -Wl,-rpath,$PWD/mosfhet_amd */
#include <stdio.h>
#include <time.h>
#include <mosfhet.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
int main(void) {
  mosfhet_seed(1);
  TLWE_Key lk = tlwe_new_binary_key(585, 9.1418e-5);
  TRLWE_Key rk = trlwe_new_binary_key(1024, 1, 2.989e-8);
  TRGSW_Key gk = trgsw_new_key(rk, 2, 8);
  Bootstrap_Key bk = new_bootstrap_key(gk, lk, 1);
  Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
  TRLWE tv = trlwe_alloc_new_sample(1, 1024);
  trlwe_torus_packing(tv, lut, 4);
  TLWE in = tlwe_new_sample(double2torus(1. / 8), lk), out = tlwe_alloc_sample(1024);
  programmable_bootstrap(out, tv, in, bk, 3, 0, 0);
  const double t0 = now();
  for (int i = 0; i < 50; i++) programmable_bootstrap(out, tv, in, bk, 3, 0, 0);
  printf("programmable_bootstrap(out, tv, in, bk, 3, 0, 0): %.3f ms per call (single sample, host structs in and out)\n", (now() - t0) / 50);
  /* the canonical gate of the reference's callers (applications/multi-ciphertext-arith/src/integer.c:94-95): key switch N -> n, then bootstrap */
  {
    TLWE_Key xk = tlwe_alloc_key(1024, 2.989e-8);
    trlwe_extract_tlwe_key(xk, rk);
    TLWE_KS_Key ks = tlwe_new_KS_key(lk, xk, 5, 2);
    TLWE big = tlwe_new_sample(double2torus(1. / 8), xk), small = tlwe_alloc_sample(585);
    tlwe_keyswitch(small, big, ks);
    programmable_bootstrap(out, tv, small, bk, 3, 0, 0);
    const double g0 = now();
    for (int i = 0; i < 50; i++) {
      tlwe_keyswitch(small, big, ks);
      programmable_bootstrap(out, tv, small, bk, 3, 0, 0);
    }
    printf("tlwe_keyswitch + programmable_bootstrap (one gate): %.3f ms per gate (two calls, host structs in and out)\n", (now() - g0) / 50);
    const double k0 = now();
    for (int i = 0; i < 50; i++) tlwe_keyswitch(small, big, ks);
    printf("tlwe_keyswitch alone: %.3f ms per call\n", (now() - k0) / 50);
  }
  /* the batch entry with host structs: marshalling + copies + kernel */
  enum { B = 4096 };
  TLWE *ins = tlwe_alloc_sample_array(B, 585), *outs = tlwe_alloc_sample_array(B, 1024);
  for (int i = 0; i < B; i++) tlwe_sample(ins[i], double2torus((i % 4) / 8.), lk);
  programmable_bootstrap_batch(outs, tv, ins, B, bk, 3, 0, 0);
  const double t1 = now();
  for (int i = 0; i < 5; i++) programmable_bootstrap_batch(outs, tv, ins, B, bk, 3, 0, 0);
  const double ms = (now() - t1) / 5;
  printf("programmable_bootstrap_batch, %d samples: %.2f ms per call = %.1f k bootstraps/s (host structs in and out)\n", B, ms, B / ms);
  return 0;
}
