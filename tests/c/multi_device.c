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

int main(int argc, char **argv) {
  enum { n = 64, N = 1024, k = 1, l = 2, Bg_bit = 8, COUNT = 301, BIG = 2 * 4096 + 11 };
  const int devs[2] = {0, 0};
  const int n_dev = argc > 1 ? atoi(argv[1]) : 2;
  setvbuf(stdout, NULL, _IOLBF, 0);
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
  const int probe[6] = {0, 149, 150, 151, 152, COUNT - 1};
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
  printf("multi_device (%d contexts): %s\n", n_dev, failures ? "FAILED" : "ok");
  free_tlwe(one); free_tlwe(one_n); free_tlwe_array(in, BIG); free_tlwe_array(out, BIG); free_tlwe_array(back, COUNT); free_trlwe(tv);
  free_tlwe_ks_key(ks); free_bootstrap_key(bk); free_trgsw_key(trgsw_key); free_trlwe_key(rlwe_key); free_tlwe_key(lwe_key); free_tlwe_key(extracted);
  return failures > 255 ? 255 : failures;
}
