/*
 * tools/in_api_devices.c -- BASELINE.json configs[3] and [4] through the DROP-IN API with several devices behind it (mosfhet_set_devices): one process,
 * host structs in and out, every *_batch call cut into one slice per device by the library (csrc/host/mosfhet_compat_multi.c), keys replicated device
 * to device on first use.  What a maintainer of a MOSFHET program gets without touching their code beyond the *_batch calls.
 *
 *   gcc -O2 -Iinclude tools/in_api_devices.c -o tools/in_api_devices_bin -Lmosfhet_amd -lmosfhet_hip -lm -Wl,-rpath,$PWD/mosfhet_amd
 *   tools/in_api_devices_bin 0,1,2,3,4,5,6,7 [steps]        (a device may be listed more than once: "0,0" = two contexts on one GPU)
 *
 * One JSON line per workload (TFHEpp lvl2 parameters: n = 632, N = 2048, l = 4, Bg = 2^9): the first call (with key replication) and the steady state.
 */
#include <mosfhet.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static uint64_t tdist(Torus a, Torus b) { int64_t d = (int64_t)(a - b); return (uint64_t)(d < 0 ? -d : d); }

static int n_dev = 1;
static void report(const char *name, const char *workload, int units, double first_ms, double ms, int ok) {
  printf("{\"metric\": \"%s\", \"value\": %.1f, \"unit\": \"units/s\", \"n_gpus\": %d, \"ms_per_step\": %.3f, \"first_call_ms\": %.3f, \"scaling\": \"strong\", "
         "\"dtype\": \"f64\", \"data\": \"synthetic\", \"config\": {\"workload\": \"%s\", \"entry\": \"drop-in API, host structs, mosfhet_set_devices\"}, \"decrypts\": %s}\n",
         name, units / (ms * 1e-3), n_dev, ms, first_ms, workload, ok ? "true" : "false");
  fflush(stdout);
}

int main(int argc, char **argv) {
  enum { n = 632, N = 2048, k = 1, l = 4, Bg_bit = 9, B = 1024 };
  int devs[16];
  n_dev = 0;
  {
    char *list = strdup(argc > 1 ? argv[1] : "0"), *save = NULL;
    for (char *tok = strtok_r(list, ",", &save); tok && n_dev < 16; tok = strtok_r(NULL, ",", &save)) devs[n_dev++] = atoi(tok);
    free(list);
  }
  const int steps = argc > 2 ? atoi(argv[2]) : 3;
  mosfhet_set_devices(n_dev, devs);
  mosfhet_seed(0x4D4F5346);
  const double sigma_rlwe = 5.684341886080802e-14, sigma_lwe = 3.0517578125e-05;   /* 2^-44, 2^-15: TFHEpp lvl2 */
  TLWE_Key lwe_key = tlwe_new_binary_key(n, sigma_lwe);
  TRLWE_Key rkey = trlwe_new_binary_key(N, k, sigma_rlwe);
  TLWE_Key ext = tlwe_alloc_key(N, sigma_rlwe);
  trlwe_extract_tlwe_key(ext, rkey);
  TRGSW_Key gkey = trgsw_new_key(rkey, l, Bg_bit);
  double t0 = now();
  Bootstrap_Key bk = new_bootstrap_key(gkey, lwe_key, 1);
  TRLWE_KS_Key *kska = trlwe_new_priv_KS_key(rkey, rkey, 20, 2);
  Generic_KS_Key kskb = trlwe_new_packing1_KS_key(rkey, ext, 6, 4);     /* test/tests.c:967-976: 184,320 rows, 3 GB seed-compressed */
  fprintf(stderr, "keys of configs[3] on the primary device: %.0f ms\n", now() - t0);
  TLWE *in = tlwe_alloc_sample_array(B, n), *out = tlwe_alloc_sample_array(8 * B, N);

  /* ---- configs[3]: circuit_bootstrap_3 (src/bootstrap.c:346-366), batch of 1024 ---- */
  {
    TRGSW *sel = (TRGSW *)malloc(sizeof(TRGSW) * B);
    for (int i = 0; i < B; i++) { sel[i] = trgsw_alloc_new_sample(l, Bg_bit, k, N); tlwe_sample(in[i], double2torus((i & 1) / 4.), lwe_key); }
    t0 = now();
    circuit_bootstrap_3_batch(sel, in, B, bk, kska, kskb);
    const double first = now() - t0;
    t0 = now();
    for (int s = 0; s < steps; s++) circuit_bootstrap_3_batch(sel, in, B, bk, kska, kskb);
    const double ms = (now() - t0) / steps;
    TorusPolynomial ph = polynomial_new_torus_polynomial(N);
    int ok = 1;
    for (int i = 0; i < B; i += 37) {
      trlwe_phase(ph, sel[i]->samples[l], rkey);
      ok &= tdist((Torus)(i & 1) << (64 - Bg_bit), ph->coeffs[0]) < (1ULL << 52);
    }
    report("circuit bootstraps/sec (circuit_bootstrap_3_batch), N=2048 l=4", "circuit_bootstrap_3_batch, 1024 host TLWE in, 1024 host TRGSW out, packing key t=6 bb=4, private key t=20 bb=2 (BASELINE.json configs[3])", B, first, ms, ok);
    free_polynomial(ph);
    for (int i = 0; i < B; i++) free_trgsw(sel[i]);
    free(sel);
  }
  free_trlwe_generic_ks_key(kskb); free_trlwe_ks_key(kska[0]); free_trlwe_ks_key(kska[1]); free(kska);

  /* ---- configs[4]: multi-value bootstrap, 8 LUTs of 2 slots per blind rotation (src/bootstrap.c:222-230) ---- */
  {
    enum { SLOTS = 2, LUTS = 8 };
    Torus lut[SLOTS * LUTS];
    for (int i = 0; i < SLOTS * LUTS; i++) lut[i] = int2torus((uint64_t)((5 * i + 3) & 15), 4);
    TRLWE tv = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing_many_LUT(tv, lut, SLOTS, LUTS);
    for (int i = 0; i < B; i++) tlwe_sample(in[i], double2torus((i % SLOTS) / (2. * SLOTS)), lwe_key);
    t0 = now();
    multivalue_bootstrap_CLOT21_batch(out, tv, in, B, bk, SLOTS, LUTS);
    const double first = now() - t0;
    t0 = now();
    for (int s = 0; s < steps; s++) multivalue_bootstrap_CLOT21_batch(out, tv, in, B, bk, SLOTS, LUTS);
    const double ms = (now() - t0) / steps;
    int ok = 1;
    for (int i = 0; i < B; i += 13)
      for (int j = 0; j < LUTS; j++) ok &= tdist(lut[j * SLOTS + i % SLOTS], tlwe_phase(out[i * LUTS + j], ext)) < (1ULL << 58);
    report("multi-value bootstraps/sec (multivalue_bootstrap_CLOT21_batch, 8 LUTs), N=2048", "multivalue_bootstrap_CLOT21_batch, torus_base 2, 8 LUTs, 1024 inputs (BASELINE.json configs[4])", B, first, ms, ok);
    free_trlwe(tv);
  }
  /* ---- configs[4]: full_domain_functional_bootstrap, precision 3 (src/bootstrap.c:519-538) ---- */
  {
    t0 = now();
    TLWE_KS_Key ks = tlwe_new_KS_key(lwe_key, ext, 8, 4);
    fprintf(stderr, "LWE key-switch key (1.2 GB, host generated): %.0f ms\n", now() - t0);
    Torus lut8[8];
    for (int i = 0; i < 8; i++) lut8[i] = int2torus((uint64_t)((3 * i + 1) & 7), 3);
    TRLWE tv8 = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing_many_LUT(tv8, lut8, 4, 2);
    for (int i = 0; i < B; i++) tlwe_sample(in[i], int2torus((uint64_t)(i % 8), 3), lwe_key);
    t0 = now();
    full_domain_functional_bootstrap_batch(out, tv8, in, B, bk, ks, 3);
    const double first = now() - t0;
    t0 = now();
    for (int s = 0; s < steps; s++) full_domain_functional_bootstrap_batch(out, tv8, in, B, bk, ks, 3);
    const double ms = (now() - t0) / steps;
    int ok = 1;
    for (int i = 0; i < B; i += 7) ok &= tdist(lut8[i % 8], tlwe_phase(out[i], ext)) < (1ULL << 58);
    report("full-domain functional bootstraps/sec (full_domain_functional_bootstrap_batch), N=2048", "full_domain_functional_bootstrap_batch, precision 3, 1024 inputs (BASELINE.json configs[4])", B, first, ms, ok);
    free_trlwe(tv8); free_tlwe_ks_key(ks);
  }
  /* ---- configs[4]: Galois-automorphism bootstrap (src/bootstrap_ga.c:62-76) ---- */
  {
    t0 = now();
    Bootstrap_GA_Key gk = new_bootstrap_key_ga(gkey, lwe_key);
    fprintf(stderr, "Galois bootstrap key + 2048 automorphism keys (256 MiB): %.0f ms\n", now() - t0);
    Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
    TRLWE tv = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing(tv, lut, 4);
    for (int i = 0; i < B; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lwe_key);
    t0 = now();
    functional_bootstrap_ga_batch(out, tv, in, B, gk, 4);
    const double first = now() - t0;
    t0 = now();
    for (int s = 0; s < steps; s++) functional_bootstrap_ga_batch(out, tv, in, B, gk, 4);
    const double ms = (now() - t0) / steps;
    int ok = 1;
    for (int i = 0; i < B; i += 7) ok &= tdist(lut[i % 4], tlwe_phase(out[i], ext)) < (1ULL << 58);
    report("Galois-automorphism functional bootstraps/sec (functional_bootstrap_ga_batch), N=2048", "functional_bootstrap_ga_batch, 1024 inputs, n=632 (BASELINE.json configs[4])", B, first, ms, ok);
    free_trlwe(tv); free_bootstrap_key_ga(gk);
  }
  {
    static const char *route[4] = {"same device", "peer to peer", "device to device (no peer access)", "host bounce buffer"};
    unsigned long long bytes[4];
    double seconds[4];
    int keys[4];
    mosfhet_replication_stats(bytes, seconds, keys);
    for (int r = 0; r < 4; r++)
      if (keys[r])
        printf("{\"key_replication\": \"%s\", \"keys\": %d, \"megabytes\": %.1f, \"ms\": %.1f, \"gb_per_s\": %.1f, \"n_gpus\": %d}\n", route[r], keys[r], bytes[r] / 1e6,
               seconds[r] * 1e3, bytes[r] / 1e9 / (seconds[r] > 0 ? seconds[r] : 1), n_dev);
  }
  free_tlwe_array(in, B); free_tlwe_array(out, 8 * B);
  free_bootstrap_key(bk); free_trgsw_key(gkey); free_tlwe_key(ext); free_trlwe_key(rkey); free_tlwe_key(lwe_key);
  return 0;
}
