/*
 * GPU test of the multi-device path of the MOSFHET-compatible API (mosfhet_set_devices; csrc/host/mosfhet_compat_multi.c): the device list names
 * GPU 0 TWICE, i.e. two contexts, two host threads, two sets of staging buffers and a replicated key set on one physical GPU -- everything the
 * 8-GPU case does except the second board.  Every sharded batch must reproduce, bit for bit, what the same samples give one at a time on the primary
 * context.  Run by tests/test_gpu_parity.py::test_multi_device_compat; exit status = number of failed checks.
 */
#include <mosfhet.h>

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __func__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static int same_tlwe(TLWE a, TLWE b) { return a->b == b->b && !memcmp(a->a, b->a, sizeof(Torus) * (size_t)a->n); }
static uint64_t tdist(Torus a, Torus b) { int64_t d = (int64_t)(a - b); return (uint64_t)(d < 0 ? -d : d); }

static int same_trgsw(TRGSW a, TRGSW b, int rows, int N) {
  for (int q = 0; q < rows; q++)
    if (memcmp(a->samples[q]->b->coeffs, b->samples[q]->b->coeffs, sizeof(Torus) * (size_t)N) || memcmp(a->samples[q]->a[0]->coeffs, b->samples[q]->a[0]->coeffs, sizeof(Torus) * (size_t)N))
      return 0;
  return 1;
}

/* BASELINE configs[3] and [4] inside the drop-in API: circuit bootstraps (both variants), the Galois bootstrap, the multi-value bootstrap and the KS21 /
 * CLOT21 full-domain bootstraps -- every sharded batch against its single calls on the primary context, bit for bit.  Keys replicated: bootstrap keys
 * (plain and Galois), the automorphism key set, the private FFT key-switch pair, the relinearisation key, the packing and private table keys
 * (seed-compressed: they travel as their b halves). */
static void wider_callers(TLWE_Key lwe_key, int n_dev) {
  enum { N = 1024, k = 1, wl = 4, wBg = 9, C = 19 };   /* >= 2 x 8 devices: every batch below is sharded even over eight contexts */
  (void)n_dev;
  const int n = lwe_key->n;
  TRLWE_Key wkey = trlwe_new_binary_key(N, k, 5.684341886080802e-14);
  TLWE_Key wex = tlwe_alloc_key(N, wkey->sigma);
  trlwe_extract_tlwe_key(wex, wkey);
  TRGSW_Key wgkey = trgsw_new_key(wkey, wl, wBg);
  Bootstrap_Key wbk = new_bootstrap_key(wgkey, lwe_key, 1);
  Generic_KS_Key wpack = trlwe_new_packing1_KS_key(wkey, wex, 12, 2), wpriv = trlwe_new_priv_SK_KS_key_N2(wkey, wex, 6, 3);
  TRLWE_KS_Key *kska = trlwe_new_priv_KS_key(wkey, wkey, 20, 2), rlk = trlwe_new_RL_key(wkey, 2, 20);
  TLWE *in = tlwe_alloc_sample_array(C, n), *out = tlwe_alloc_sample_array(8 * C, N);
  TLWE one = tlwe_alloc_sample(N);
  for (int i = 0; i < C; i++) tlwe_sample(in[i], int2torus((uint64_t)(i % 8), 3), lwe_key);

  /* circuit_bootstrap_3 (src/bootstrap.c:346-366) and circuit_bootstrap_2 (:324-344) */
  {
    TRGSW *sel = (TRGSW *)malloc(sizeof(TRGSW) * C), single = trgsw_alloc_new_sample(wl, wBg, k, N);
    TLWE *bits = tlwe_alloc_sample_array(C, n);
    for (int i = 0; i < C; i++) { sel[i] = trgsw_alloc_new_sample(wl, wBg, k, N); tlwe_sample(bits[i], double2torus((i & 1) / 4.), lwe_key); }
    circuit_bootstrap_3_batch(sel, bits, C, wbk, kska, wpack);
    for (int i = 0; i < C; i++) {
      circuit_bootstrap_3(single, bits[i], wbk, kska, wpack);
      CHECK(same_trgsw(single, sel[i], 2 * wl, N), "circuit_bootstrap_3 %d of the sharded batch differs from its single call", i);
    }
    TorusPolynomial ph = polynomial_new_torus_polynomial(N);
    for (int i = 0; i < C; i++) {   /* b row of level 0 decrypts to m * 2^(64 - Bg) on X^0 */
      trlwe_phase(ph, sel[i]->samples[wl], wkey);
      CHECK(tdist((Torus)(i & 1) << (64 - wBg), ph->coeffs[0]) < (1ULL << 48), "circuit_bootstrap_3 %d does not decrypt", i);
    }
    free_polynomial(ph);
    circuit_bootstrap_2_batch(sel, bits, C, wbk, wpriv, wpack);
    for (int i = 0; i < C; i++) {
      circuit_bootstrap_2(single, bits[i], wbk, wpriv, wpack);
      CHECK(same_trgsw(single, sel[i], 2 * wl, N), "circuit_bootstrap_2 %d of the sharded batch differs from its single call", i);
    }
    for (int i = 0; i < C; i++) free_trgsw(sel[i]);
    free(sel); free_trgsw(single); free_tlwe_array(bits, C);
  }
  /* functional_bootstrap_ga (src/bootstrap_ga.c:62-76): short LWE key as in the reference's test (its odd-forcing of the mask shifts the phase) */
  {
    enum { gn = 40 };
    TLWE_Key skey = tlwe_new_binary_key(gn, 9.1418e-5 / 4);
    Bootstrap_GA_Key gk = new_bootstrap_key_ga(wgkey, skey);
    Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
    TRLWE tv = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing(tv, lut, 4);
    TLWE *gin = tlwe_alloc_sample_array(C, gn);
    for (int i = 0; i < C; i++) tlwe_sample(gin[i], double2torus((i % 4) / 8.), skey);
    functional_bootstrap_ga_batch(out, tv, gin, C, gk, 4);
    for (int i = 0; i < C; i++) {
      functional_bootstrap_ga(one, tv, gin[i], gk, 4);
      CHECK(same_tlwe(one, out[i]), "functional_bootstrap_ga %d of the sharded batch differs from its single call", i);
      CHECK(tdist(lut[i % 4], tlwe_phase(out[i], wex)) < (1ULL << 58), "functional_bootstrap_ga %d does not decrypt", i);
    }
    free_tlwe_array(gin, C); free_trlwe(tv); free_bootstrap_key_ga(gk); free_tlwe_key(skey);
  }
  /* multivalue_bootstrap_CLOT21 (src/bootstrap.c:222-230): 4 LUTs of 4 slots, one blind rotation each */
  {
    enum { SLOTS = 4, LUTS = 4 };
    Torus lut[SLOTS * LUTS];
    for (int i = 0; i < SLOTS * LUTS; i++) lut[i] = int2torus((uint64_t)((7 * i + 2) & 15), 4);
    TRLWE tv = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing_many_LUT(tv, lut, SLOTS, LUTS);
    TLWE *min = tlwe_alloc_sample_array(C, n), mone[LUTS];
    for (int j = 0; j < LUTS; j++) mone[j] = tlwe_alloc_sample(N);
    for (int i = 0; i < C; i++) tlwe_sample(min[i], double2torus((i % SLOTS) / (2. * SLOTS)), lwe_key);
    multivalue_bootstrap_CLOT21_batch(out, tv, min, C, wbk, SLOTS, LUTS);
    for (int i = 0; i < C; i++) {
      multivalue_bootstrap_CLOT21(mone, tv, min[i], wbk, SLOTS, LUTS);
      for (int j = 0; j < LUTS; j++) {
        CHECK(same_tlwe(mone[j], out[i * LUTS + j]), "multivalue_bootstrap_CLOT21 %d, LUT %d of the sharded batch differs from its single call", i, j);
        CHECK(tdist(lut[j * SLOTS + i % SLOTS], tlwe_phase(out[i * LUTS + j], wex)) < (1ULL << 58), "multivalue_bootstrap_CLOT21 %d, LUT %d does not decrypt", i, j);
      }
    }
    for (int j = 0; j < LUTS; j++) free_tlwe(mone[j]);
    free_tlwe_array(min, C); free_trlwe(tv);
  }
  /* full_domain_functional_bootstrap_KS21 (src/bootstrap.c:391-426) and _CLOT21_2 (:491-517) */
  {
    Torus in8[8];
    for (int i = 0; i < 8; i++) in8[i] = int2torus((uint64_t)((5 * i + 3) & 15), 4);
    TorusPolynomial poly = polynomial_new_torus_polynomial(2 * N);
    for (int i = 0; i < 2 * N; i++) poly->coeffs[i] = in8[i / (N / 4)];
    full_domain_functional_bootstrap_KS21_batch(out, poly, in, C, wbk, wpack, 8);
    for (int i = 0; i < C; i++) {
      full_domain_functional_bootstrap_KS21(one, poly, in[i], wbk, wpack, 8);
      CHECK(same_tlwe(one, out[i]), "full_domain_functional_bootstrap_KS21 %d of the sharded batch differs from its single call", i);
      CHECK(tdist(in8[i % 8], tlwe_phase(out[i], wex)) < (1ULL << 58), "full_domain_functional_bootstrap_KS21 %d does not decrypt", i);
    }
    full_domain_functional_bootstrap_CLOT21_2_batch(out, in8, in, C, wbk, wpack, rlk, 4);
    for (int i = 0; i < C; i++) {
      full_domain_functional_bootstrap_CLOT21_2(one, in8, in[i], wbk, wpack, rlk, 4);
      CHECK(same_tlwe(one, out[i]), "full_domain_functional_bootstrap_CLOT21_2 %d of the sharded batch differs from its single call", i);
      CHECK(tdist(in8[i % 8], tlwe_phase(out[i], wex)) < (1ULL << 59), "full_domain_functional_bootstrap_CLOT21_2 %d does not decrypt", i);
    }
    free_polynomial(poly);
  }
  free_tlwe(one); free_tlwe_array(in, C); free_tlwe_array(out, 8 * C);
  free_trlwe_ks_key(rlk); free_trlwe_ks_key(kska[0]); free_trlwe_ks_key(kska[1]); free(kska);
  free_trlwe_generic_ks_key(wpack); free_trlwe_generic_ks_key(wpriv); free_bootstrap_key(wbk); free_trgsw_key(wgkey); free_tlwe_key(wex); free_trlwe_key(wkey);
}

int main(int argc, char **argv) {
  enum { n = 64, N = 1024, k = 1, l = 2, Bg_bit = 8, COUNT = 301, BIG = 2 * 4096 + 11 };
  int devs[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* up to eight contexts on GPU 0: the slice arithmetic, worker threads and replicas of an 8-GPU node */
  const int n_dev = argc > 1 ? atoi(argv[1]) : 2;
  /* "distinct": context i on board i (the box must have n_dev of them) -- the first run on more than one physical GPU takes this form; the key replicas must then
   * travel peer to peer (or, with MOSFHET_HIP_NO_PEER set, by the route that forces) and every sharded batch must still equal the primary board's single calls */
  const int distinct = argc > 2 && !strcmp(argv[2], "distinct");
  if (n_dev < 1 || n_dev > 8) { printf("usage: multi_device [1 .. 8] [distinct]\n"); return 255; }
  setvbuf(stdout, NULL, _IOLBF, 0);
  if (distinct) {
    for (int i = 0; i < n_dev; i++) devs[i] = i;
    printf("multi_device: %d contexts on %d distinct boards\n", n_dev, n_dev);
  }
  mosfhet_set_devices(n_dev, devs);
  mosfhet_seed(0x4D4F5346);
  CHECK(mosfhet_device_count() == n_dev, "device count");
  TLWE_Key lwe_key = tlwe_new_binary_key(n, 9.1418e-5 / 4);
  TRLWE_Key rlwe_key = trlwe_new_binary_key(N, k, 2.989e-8);
  TRGSW_Key trgsw_key = trgsw_new_key(rlwe_key, l, Bg_bit);
  TLWE_Key extracted = tlwe_alloc_key(N, 2.989e-8);
  trlwe_extract_tlwe_key(extracted, rlwe_key);
  Bootstrap_Key bk = new_bootstrap_key(trgsw_key, lwe_key, 1);
  TLWE_KS_Key ks = tlwe_new_KS_key(lwe_key, extracted, 5, 2);
  Torus lut[4] = {int2torus(3, 4), int2torus(7, 4), int2torus(11, 4), int2torus(15, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE *in = tlwe_alloc_sample_array(BIG, n), *out = tlwe_alloc_sample_array(BIG, N), *back = tlwe_alloc_sample_array(COUNT, n);
  TLWE one = tlwe_alloc_sample(N), one_n = tlwe_alloc_sample(n);
  for (int i = 0; i < BIG; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lwe_key);

  /* programmable bootstraps: a ragged batch (slices of 151 and 150), then one whose slices are pipelined over two streams each */
  programmable_bootstrap_batch(out, tv, in, COUNT, bk, 3, 0, 0);
  int bad = 0;
  for (int i = 0; i < COUNT; i++) bad += tdist(lut[i % 4], tlwe_phase(out[i], extracted)) >= (1ULL << 58);
  CHECK(bad == 0, "%d of %d sharded bootstraps do not decrypt", bad, COUNT);
  int probe[6] = {0, 149, 150, 151, 152, COUNT - 1};
  if (n_dev > 2) {   /* both sides of the first and of the last slice boundary (mosfhet_amd/shard.py: the first COUNT % n slices get one more) */
    const int base = COUNT / n_dev, extra = COUNT % n_dev, first_end = base + (extra > 0), last_begin = COUNT - base;
    probe[1] = first_end - 1; probe[2] = first_end; probe[3] = last_begin - 1; probe[4] = last_begin;
  }
  for (int q = 0; q < 6; q++) {
    programmable_bootstrap(one, tv, in[probe[q]], bk, 3, 0, 0);
    CHECK(same_tlwe(one, out[probe[q]]), "sample %d of the sharded batch differs from its single call on the primary device", probe[q]);
  }
  /* LWE key switch back, sharded, against single calls */
  tlwe_keyswitch_batch(back, out, COUNT, ks);
  for (int q = 0; q < 6; q++) {
    tlwe_keyswitch(one_n, out[probe[q]], ks);
    CHECK(same_tlwe(one_n, back[probe[q]]), "key switch %d of the sharded batch differs from its single call", probe[q]);
  }
  /* full-domain functional bootstrap (bootstrap key + key-switch key replicated) */
  {
    Torus lut8[8];
    for (int i = 0; i < 8; i++) lut8[i] = int2torus((uint64_t)((3 * i + 1) & 7), 3);
    TRLWE tv8 = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing_many_LUT(tv8, lut8, 4, 2);   /* test_FDFB_new (test/tests.c:1095-1127): 2 interleaved tables of 4 */
    TLWE *fin = tlwe_alloc_sample_array(16, n), *fout = tlwe_alloc_sample_array(16, N);
    for (int i = 0; i < 16; i++) tlwe_sample(fin[i], int2torus((uint64_t)(i % 8), 3), lwe_key);
    full_domain_functional_bootstrap_batch(fout, tv8, fin, 16, bk, ks, 3);
    for (int i = 0; i < 16; i++) {
      full_domain_functional_bootstrap(one, tv8, fin[i], bk, ks, 3);
      CHECK(same_tlwe(one, fout[i]), "full-domain bootstrap %d of the sharded batch differs from its single call", i);
      CHECK(tdist(lut8[i % 8], tlwe_phase(fout[i], extracted)) < (1ULL << 59), "full-domain bootstrap %d does not decrypt", i);
    }
    free_tlwe_array(fin, 16); free_tlwe_array(fout, 16); free_trlwe(tv8);
  }
  /* a batch large enough that every device's slice takes the two-stream pipelined path */
  functional_bootstrap_batch(out, tv, in, BIG, bk, 4);
  bad = 0;
  for (int i = 0; i < BIG; i++) bad += tdist(lut[i % 4], tlwe_phase(out[i], extracted)) >= (1ULL << 58);
  CHECK(bad == 0, "%d of %d bootstraps of the large sharded batch do not decrypt", bad, BIG);
  const int probe2[5] = {0, 4100, 4101, 4102, BIG - 1};
  for (int q = 0; q < 5; q++) {
    functional_bootstrap(one, tv, in[probe2[q]], bk, 4);
    CHECK(same_tlwe(one, out[probe2[q]]), "sample %d of the large sharded batch differs from its single call", probe2[q]);
  }
  wider_callers(lwe_key, n_dev);
  /* a key of the general-ring path (k = 2, N = 512: csrc/general_kernels.h) replicates and shards like the tuned ones */
  {
    enum { gN = 512, gk = 2, gl = 2, gBg = 10, GC = 18 };
    TRLWE_Key grk = trlwe_new_binary_key(gN, gk, 2.989e-11);
    TRGSW_Key ggk = trgsw_new_key(grk, gl, gBg);
    TLWE_Key gex = tlwe_alloc_key(gk * gN, 2.989e-11);
    trlwe_extract_tlwe_key(gex, grk);
    Bootstrap_Key gbk = new_bootstrap_key(ggk, lwe_key, 1);
    TRLWE gtv = trlwe_alloc_new_sample(gk, gN);
    trlwe_torus_packing(gtv, lut, 4);
    TLWE *gout = tlwe_alloc_sample_array(GC, gk * gN);
    TLWE gone = tlwe_alloc_sample(gk * gN);
    functional_bootstrap_batch(gout, gtv, in, GC, gbk, 4);
    for (int i = 0; i < GC; i++) {
      functional_bootstrap(gone, gtv, in[i], gbk, 4);
      CHECK(same_tlwe(gone, gout[i]), "general-ring bootstrap %d of the sharded batch differs from its single call", i);
      CHECK(tdist(lut[i % 4], tlwe_phase(gout[i], gex)) < (1ULL << 58), "general-ring bootstrap %d does not decrypt", i);
    }
    free_tlwe(gone); free_tlwe_array(gout, GC); free_trlwe(gtv); free_bootstrap_key(gbk); free_tlwe_key(gex); free_trgsw_key(ggk); free_trlwe_key(grk);
  }
  {
    static const char *route[4] = {"same device", "peer to peer", "device to device (no peer access)", "host bounce buffer"};
    unsigned long long bytes[4];
    double seconds[4];
    int keys[4];
    mosfhet_replication_stats(bytes, seconds, keys);
    for (int r = 0; r < 4; r++)
      if (keys[r])
        printf("key replication, %s: %d keys, %.1f MB in %.1f ms (%.1f GB/s)\n", route[r], keys[r], bytes[r] / 1e6, seconds[r] * 1e3, bytes[r] / 1e9 / (seconds[r] > 0 ? seconds[r] : 1));
    if (n_dev > 1) CHECK(keys[0] + keys[1] + keys[2] + keys[3] >= 8, "fewer replicated keys than the sharded calls use (%d)", keys[0] + keys[1] + keys[2] + keys[3]);
    if (distinct && n_dev > 1) {
      CHECK(keys[0] == 0, "%d key replicas took the same-device route between distinct boards", keys[0]);
      if (!getenv("MOSFHET_HIP_NO_PEER")) CHECK(keys[1] > 0 && keys[2] + keys[3] == 0, "replicas between distinct boards did not all travel peer to peer (%d / %d / %d)", keys[1], keys[2], keys[3]);
    }
  }
  printf("multi_device (%d contexts): %s\n", n_dev, failures ? "FAILED" : "ok");
  free_tlwe(one); free_tlwe(one_n); free_tlwe_array(in, BIG); free_tlwe_array(out, BIG); free_tlwe_array(back, COUNT); free_trlwe(tv);
  free_tlwe_ks_key(ks); free_bootstrap_key(bk); free_trgsw_key(trgsw_key); free_trlwe_key(rlwe_key); free_tlwe_key(lwe_key); free_tlwe_key(extracted);
  return failures > 255 ? 255 : failures;
}
