/*
 * GPU test of the MOSFHET-compatible C API: a plain C program written against <mosfhet.h> (include/mosfhet.h forwards to
 * include/mosfhet_compat.h) the way a mosfhet.h caller is (cf. the reference's test/tests.c cases named per function below), linked against
 * libmosfhet_hip.so only.  Every case decrypts with the secret keys and applies the reference test's own tolerance.
 * Run by tests/test_gpu_parity.py::test_compat_c_api_suite; exit status = number of failed cases.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mosfhet.h>

static int failures = 0;
static uint64_t tdist(Torus a, Torus b) {
  int64_t d = (int64_t)(a - b);
  return (uint64_t)(d < 0 ? -d : d);
}
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __func__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)
#define WITHIN(tol, want, got, what) CHECK(tdist((want), (got)) < (tol), "%s: want %016llx got %016llx", what, (unsigned long long)(want), (unsigned long long)(got))

/* SET_1 of test/benchmark.c:53-54 with a shortened LWE key so that the suite runs in seconds */
enum { n = 96, N = 1024, k = 1, l = 2, Bg_bit = 8, ks_t = 5, ks_bb = 2 };
static const double lwe_sigma = 9.1418e-5 / 4, rlwe_sigma = 2.989e-8;

static TLWE_Key lwe_key, extracted_key;
static TRLWE_Key rlwe_key;
static TRGSW_Key trgsw_key;
static Bootstrap_Key bk;

/* test_functional_bootstrap (test/tests.c:1446-1480): m = j/8 on a 4-slot LUT, tolerance 2^58 */
static void case_functional_bootstrap(void) {
  Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE out = tlwe_alloc_sample(N);
  for (int j = 0; j < 4; j++) {
    TLWE in = tlwe_new_sample(double2torus(j / 8.), lwe_key);
    functional_bootstrap(out, tv, in, bk, 4);
    WITHIN(1ULL << 58, lut[j], tlwe_phase(out, extracted_key), "functional_bootstrap");
    free_tlwe(in);
  }
  /* wo_extract + host sample extract is the same ciphertext */
  TLWE in = tlwe_new_sample(double2torus(2 / 8.), lwe_key);
  TRLWE acc = trlwe_alloc_new_sample(k, N);
  TLWE out2 = tlwe_alloc_sample(N);
  functional_bootstrap(out, tv, in, bk, 4);
  functional_bootstrap_wo_extract(acc, tv, in, bk, 4);
  trlwe_extract_tlwe(out2, acc, 0);
  CHECK(out->b == out2->b && !memcmp(out->a, out2->a, sizeof(Torus) * N), "wo_extract + extract differs from functional_bootstrap");
  free_tlwe(in); free_tlwe(out); free_tlwe(out2); free_trlwe(acc); free_trlwe(tv);
}

/* test_programmable_bootstrap (test/tests.c:1483-1520): precision 3, messages on the half torus; batch entry point too */
static void case_programmable_bootstrap(void) {
  enum { COUNT = 70 };
  Torus lut[4] = {int2torus(3, 4), int2torus(7, 4), int2torus(11, 4), int2torus(15, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE *in = tlwe_alloc_sample_array(COUNT, n), *out = tlwe_alloc_sample_array(COUNT, N);
  for (int i = 0; i < COUNT; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lwe_key);
  programmable_bootstrap_batch(out, tv, in, COUNT, bk, 3, 0, 0);
  for (int i = 0; i < COUNT; i++) WITHIN(1ULL << 58, lut[i % 4], tlwe_phase(out[i], extracted_key), "programmable_bootstrap_batch");
  TLWE one = tlwe_alloc_sample(N);
  programmable_bootstrap(one, tv, in[5], bk, 3, 0, 0);
  CHECK(one->b == out[5]->b && !memcmp(one->a, out[5]->a, sizeof(Torus) * N), "single call differs from the batch entry");
  free_tlwe(one); free_tlwe_array(in, COUNT); free_tlwe_array(out, COUNT); free_trlwe(tv);
}

/* blind_rotate with the reference's signature (src/bootstrap.c:107-122): the caller rotates the test vector by the body
 * itself (as functional_bootstrap_wo_extract does, :192-198), then blind_rotate + sample extract == functional_bootstrap */
static void case_blind_rotate(void) {
  Torus lut[4] = {int2torus(2, 4), int2torus(6, 4), int2torus(10, 4), int2torus(14, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N), acc = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE in = tlwe_new_sample(double2torus(1 / 8.), lwe_key);
  /* acc = tv * X^(2N - b~) on the host, then blind_rotate, then extract == functional_bootstrap */
  const int log2N = 11;
  const uint64_t bt = torus2int(in->b + double2torus(1. / 16), log2N), rot = (2 * N - bt) & (2 * N - 1);
  for (int j = 0; j < N; j++) {
    const uint64_t idx = j + rot;
    acc->a[0]->coeffs[idx & (N - 1)] = 0;
    acc->b->coeffs[idx & (N - 1)] = (idx & N) ? (Torus)0 - tv->b->coeffs[j] : tv->b->coeffs[j];
  }
  blind_rotate(acc, in->a, bk->s, n);   /* takes the raw mask: the mod switch is inside (src/bootstrap.c:113) */
  TLWE got = tlwe_alloc_sample(N), want = tlwe_alloc_sample(N);
  trlwe_extract_tlwe(got, acc, 0);
  functional_bootstrap(want, tv, in, bk, 4);
  CHECK(got->b == want->b && !memcmp(got->a, want->a, sizeof(Torus) * N), "blind_rotate + extract differs from functional_bootstrap");
  WITHIN(1ULL << 58, lut[1], tlwe_phase(got, extracted_key), "blind_rotate");
  free_tlwe(in); free_tlwe(got); free_tlwe(want); free_trlwe(tv); free_trlwe(acc);
}

/* test_tlwe_ks (test/tests.c:751-790): N -> n key switch; SET_1's t * base_bit = 10 bits leaves ~2^57 of rounding noise
 * (the set "should fail most tests", test/tests.c:39), hence 2^60 here */
static void case_tlwe_keyswitch(void) {
  enum { COUNT = 33 };
  TLWE_KS_Key ksk = tlwe_new_KS_key(lwe_key, extracted_key, ks_t, ks_bb);
  TLWE *in = tlwe_alloc_sample_array(COUNT, N), *out = tlwe_alloc_sample_array(COUNT, n);
  for (int i = 0; i < COUNT; i++) tlwe_sample(in[i], double2torus((i % 8) / 8.), extracted_key);
  tlwe_keyswitch_batch(out, in, COUNT, ksk);
  for (int i = 0; i < COUNT; i++) WITHIN(1ULL << 60, double2torus((i % 8) / 8.), tlwe_phase(out[i], lwe_key), "tlwe_keyswitch_batch");
  TLWE one = tlwe_alloc_sample(n);
  tlwe_keyswitch(one, in[3], ksk);
  CHECK(one->b == out[3]->b && !memcmp(one->a, out[3]->a, sizeof(Torus) * n), "single key switch differs from the batch entry");
  /* test_FDFB_new (test/tests.c:1095-1127): messages i/8 on the WHOLE torus, LUT packed as 2 interleaved tables of 4 */
  Torus lut[8];
  for (int i = 0; i < 8; i++) lut[i] = int2torus((uint64_t)((3 * i + 1) & 7), 3);
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing_many_LUT(tv, lut, 4, 2);
  TLWE fout = tlwe_alloc_sample(N);
  for (int m = 0; m < 8; m++) {
    TLWE c = tlwe_new_sample(int2torus((uint64_t)m, 3), lwe_key);
    full_domain_functional_bootstrap(fout, tv, c, bk, ksk, 3);
    WITHIN(1ULL << 59, lut[m], tlwe_phase(fout, extracted_key), "full_domain_functional_bootstrap");
    free_tlwe(c);
  }
  free_tlwe(fout); free_trlwe(tv); free_tlwe(one);
  free_tlwe_array(in, COUNT); free_tlwe_array(out, COUNT); free_tlwe_ks_key(ksk);
}

/* test_multivalue_bootstrap_CLOT21 (test/tests.c:931-963): several LUTs evaluated with ONE blind rotation */
static void case_multivalue(void) {
  enum { LUTS = 4, SLOTS = 4 };
  Torus lut[LUTS * SLOTS];
  for (int j = 0; j < LUTS; j++)
    for (int i = 0; i < SLOTS; i++) lut[j * SLOTS + i] = int2torus((uint64_t)((5 * j + 3 * i + 1) & 15), 4);
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing_many_LUT(tv, lut, SLOTS, LUTS);
  TLWE *out = tlwe_alloc_sample_array(LUTS, N);
  for (int m = 0; m < SLOTS; m++) {
    TLWE in = tlwe_new_sample(double2torus(m / 8.), lwe_key);
    multivalue_bootstrap_CLOT21(out, tv, in, bk, SLOTS, LUTS);
    for (int j = 0; j < LUTS; j++) WITHIN(1ULL << 58, lut[j * SLOTS + m], tlwe_phase(out[j], extracted_key), "multivalue_bootstrap_CLOT21");
    free_tlwe(in);
  }
  free_tlwe_array(out, LUTS); free_trlwe(tv);
}

/* test_functional_bootstrap_ga (test/tests.c:1615-1650) */
static void case_bootstrap_ga(void) {
  enum { n_ga = 32 };  /* short key: the forced-odd mask drift of blind_rotate_ga grows with n */
  TLWE_Key key = tlwe_new_binary_key(n_ga, lwe_sigma);
  Bootstrap_GA_Key gk = new_bootstrap_key_ga(trgsw_key, key);
  Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE out = tlwe_alloc_sample(N);
  for (int j = 0; j < 4; j++) {
    TLWE in = tlwe_new_sample(double2torus(j / 8.), key);
    functional_bootstrap_ga(out, tv, in, gk, 4);
    WITHIN(1ULL << 58, lut[j], tlwe_phase(out, extracted_key), "functional_bootstrap_ga");
    free_tlwe(in);
  }
  free_tlwe(out); free_trlwe(tv); free_bootstrap_key_ga(gk); free_tlwe_key(key);
}

/* test_circuit_bootstrap (test/tests.c:965-1022) + test_trlwe_pack_key_priv_ks (:893-925), gadget l=4 Bg=2^9 */
static void case_circuit_bootstrap(void) {
  enum { cl = 4, cBg = 9 };
  /* the l = 4, Bg = 2^9 gadget belongs to the 2^-44 noise level of the reference's lvl2 set (test/tests.c:24-33) */
  TRLWE_Key ckey = trlwe_new_binary_key(N, k, 5.684341886080802e-14);
  TLWE_Key ckey_extracted = tlwe_alloc_key(N, ckey->sigma);
  trlwe_extract_tlwe_key(ckey_extracted, ckey);
  TRGSW_Key gkey = trgsw_new_key(ckey, cl, cBg);
  Bootstrap_Key cbk = new_bootstrap_key(gkey, lwe_key, 1);
  TRLWE_KS_Key *kska = trlwe_new_priv_KS_key(ckey, ckey, 20, 2);
  Generic_KS_Key kskb = trlwe_new_packing1_KS_key(ckey, ckey_extracted, 12, 2);
  TorusPolynomial ph = polynomial_new_torus_polynomial(N), msg = polynomial_new_torus_polynomial(N);
  /* private key switch: TRLWE(m) -> TRLWE(-s m), tolerance 2^52 */
  TRLWE c = trlwe_alloc_new_sample(k, N), c2 = trlwe_alloc_new_sample(k, N);
  memset(msg->coeffs, 0, sizeof(Torus) * N);
  msg->coeffs[0] = double2torus(0.125);
  trlwe_sample(c, msg, ckey);
  trlwe_priv_keyswitch_2(c2, c, kska);
  trlwe_phase(ph, c2, ckey);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 52, (Torus)0 - ckey->s[0]->coeffs[i] * msg->coeffs[0], ph->coeffs[i], "trlwe_priv_keyswitch_2");
  /* packing key switch: LWE(1/8) under the extracted key -> TRLWE with the message in coefficient 0 */
  TLWE lw = tlwe_new_sample(double2torus(0.125), ckey_extracted);
  trlwe_packing1_keyswitch(c2, lw, kskb);
  trlwe_phase(ph, c2, ckey);
  WITHIN(1ULL << 58, double2torus(0.125), ph->coeffs[0], "trlwe_packing1_keyswitch coefficient 0");
  for (int i = 1; i < N; i++) WITHIN(1ULL << 58, (Torus)0, ph->coeffs[i], "trlwe_packing1_keyswitch");
  /* LWE(1/4) -> TRGSW(1), LWE(0) -> TRGSW(0): every row decrypts to m * gadget on its component (first two levels;
   * the 24-bit packing key switch leaves ~2^44 of rounding noise, the private one multiplies it by the key) */
  TRGSW out[2] = {trgsw_alloc_new_sample(cl, cBg, k, N), trgsw_alloc_new_sample(cl, cBg, k, N)};
  TLWE in[2] = {tlwe_new_sample(double2torus(0.25), lwe_key), tlwe_new_sample(0, lwe_key)};
  circuit_bootstrap_3_batch(out, in, 2, cbk, kska, kskb);
  for (int b = 0; b < 2; b++) {
    const Torus m = b == 0 ? 1 : 0;
    for (int i = 0; i < 2; i++) {
      const Torus h = m << (64 - (i + 1) * cBg);
      trlwe_phase(ph, out[b]->samples[cl + i], ckey);                /* b rows: m h on X^0 */
      WITHIN(1ULL << 48, h, ph->coeffs[0], "circuit_bootstrap_3 b row");
      for (int j = 1; j < N; j++) WITHIN(1ULL << 48, (Torus)0, ph->coeffs[j], "circuit_bootstrap_3 b row tail");
      trlwe_phase(ph, out[b]->samples[i], ckey);                     /* a rows: -s m h */
      for (int j = 0; j < N; j++) WITHIN(1ULL << 52, (Torus)0 - ckey->s[0]->coeffs[j] * h, ph->coeffs[j], "circuit_bootstrap_3 a row");
    }
  }
  TRGSW single = trgsw_alloc_new_sample(cl, cBg, k, N);
  circuit_bootstrap_3(single, in[0], cbk, kska, kskb);
  for (int q = 0; q < 2 * cl; q++)
    CHECK(!memcmp(single->samples[q]->b->coeffs, out[0]->samples[q]->b->coeffs, sizeof(Torus) * N) &&
          !memcmp(single->samples[q]->a[0]->coeffs, out[0]->samples[q]->a[0]->coeffs, sizeof(Torus) * N), "single circuit bootstrap differs from the batch entry, row %d", q);
  free_trgsw(single); free_trgsw(out[0]); free_trgsw(out[1]); free_tlwe(in[0]); free_tlwe(in[1]); free_tlwe(lw);
  free_trlwe(c); free_trlwe(c2); free_polynomial(ph); free_polynomial(msg);
  free_trlwe_generic_ks_key(kskb); free_trlwe_ks_key(kska[0]); free_trlwe_ks_key(kska[1]); free(kska);
  free_bootstrap_key(cbk); free_trgsw_key(gkey); free_tlwe_key(ckey_extracted); free_trlwe_key(ckey);
}

/* ---- the wider callers: one low-noise key set (2^-44, gadget l = 4, Bg = 2^9) shared by the cases below ---- */
enum { wl = 4, wBg = 9 };
static TRLWE_Key wkey;
static TLWE_Key wkey_extracted;
static TRGSW_Key wgkey;
static Bootstrap_Key wbk;
static Generic_KS_Key wpack, wpriv;
static void wide_setup(void) {
  if (wkey) return;
  wkey = trlwe_new_binary_key(N, k, 5.684341886080802e-14);
  wkey_extracted = tlwe_alloc_key(N, wkey->sigma);
  trlwe_extract_tlwe_key(wkey_extracted, wkey);
  wgkey = trgsw_new_key(wkey, wl, wBg);
  wbk = new_bootstrap_key(wgkey, lwe_key, 1);
  wpack = trlwe_new_packing1_KS_key(wkey, wkey_extracted, 12, 2);
  wpriv = trlwe_new_priv_SK_KS_key_N2(wkey, wkey_extracted, 6, 3);
}

/* test_functional_bootstrap_unfolded (test/tests.c:1486-1530): same calls, key made with unfolding = 2 and 4 */
static void case_unfolded(void) {
  wide_setup();
  Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE out = tlwe_alloc_sample(N);
  for (int u = 2; u <= 4; u += 2) {
    Bootstrap_Key ubk = new_bootstrap_key(wgkey, lwe_key, u);
    for (int j = 0; j < 4; j++) {
      TLWE in = tlwe_new_sample(double2torus(j / 8.), lwe_key);
      functional_bootstrap(out, tv, in, ubk, 4);
      WITHIN(1ULL << 58, lut[j], tlwe_phase(out, wkey_extracted), "functional_bootstrap (unfolded key)");
      /* programmable_bootstrap scales its input and calls functional_bootstrap, which dispatches on key->unfolding (src/bootstrap.c:208-219,196-197) */
      programmable_bootstrap(out, tv, in, ubk, 3, 0, 0);
      WITHIN(1ULL << 58, lut[j], tlwe_phase(out, wkey_extracted), "programmable_bootstrap (unfolded key)");
      free_tlwe(in);
    }
    if (u == 2) {
      /* a batch past two pipeline chunks with an unfolded key: must equal the single calls (the layer keeps such batches on one stream) */
      enum { COUNT = 2 * 2048 + 5 };
      TLWE *in = tlwe_alloc_sample_array(COUNT, n), *bo = tlwe_alloc_sample_array(COUNT, N);
      for (int i = 0; i < COUNT; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lwe_key);
      functional_bootstrap_batch(bo, tv, in, COUNT, ubk, 4);
      int bad = 0;
      for (int i = 0; i < COUNT; i++) bad += tdist(lut[i % 4], tlwe_phase(bo[i], wkey_extracted)) >= (1ULL << 58);
      CHECK(bad == 0, "%d of %d outputs of the unfolded batch do not decrypt", bad, COUNT);
      const int probe[5] = {0, 2047, 2048, 4096, COUNT - 1};
      for (int q = 0; q < 5; q++) {
        functional_bootstrap(out, tv, in[probe[q]], ubk, 4);
        CHECK(out->b == bo[probe[q]]->b && !memcmp(out->a, bo[probe[q]]->a, sizeof(Torus) * N), "sample %d of the unfolded batch differs from its single call", probe[q]);
      }
      programmable_bootstrap_batch(bo, tv, in, COUNT, ubk, 3, 0, 0);
      programmable_bootstrap(out, tv, in[2049], ubk, 3, 0, 0);
      CHECK(out->b == bo[2049]->b && !memcmp(out->a, bo[2049]->a, sizeof(Torus) * N), "programmable_bootstrap_batch (unfolded key) differs from the single call");
      free_tlwe_array(in, COUNT); free_tlwe_array(bo, COUNT);
    }
    free_bootstrap_key(ubk);
  }
  free_tlwe(out); free_trlwe(tv);
}

/* test_FDFB_KS21 (test/tests.c:1058-1092) and test_FDFB_CLOT21_2 (:1179-1218) */
static void case_fdfb_variants(void) {
  wide_setup();
  Torus in8[8];
  for (int i = 0; i < 8; i++) in8[i] = int2torus((uint64_t)((5 * i + 3) & 15), 4);
  TorusPolynomial poly = polynomial_new_torus_polynomial(2 * N);
  for (int i = 0; i < 2 * N; i++) poly->coeffs[i] = in8[i / (N / 4)];
  TRLWE_KS_Key rlk = trlwe_new_RL_key(wkey, 2, 20);
  TLWE out = tlwe_alloc_sample(N);
  for (int i = 0; i < 8; i++) {
    TLWE c = tlwe_new_sample(int2torus((uint64_t)i, 3), lwe_key);
    full_domain_functional_bootstrap_KS21(out, poly, c, wbk, wpack, 8);
    WITHIN(1ULL << 58, in8[i], tlwe_phase(out, wkey_extracted), "full_domain_functional_bootstrap_KS21");
    full_domain_functional_bootstrap_KS21_2(out, poly, c, wbk, wpack, 8);
    WITHIN(1ULL << 58, in8[i], tlwe_phase(out, wkey_extracted), "full_domain_functional_bootstrap_KS21_2");
    full_domain_functional_bootstrap_CLOT21_2(out, in8, c, wbk, wpack, rlk, 4);
    WITHIN(1ULL << (64 - 4 - 1), in8[i], tlwe_phase(out, wkey_extracted), "full_domain_functional_bootstrap_CLOT21_2");
    free_tlwe(c);
  }
  /* tlwe_mul (src/tlwe.c:322-332): 3/16 * 2/16 * 2^4 = 6/16 */
  TLWE a = tlwe_new_sample(int2torus(3, 4), wkey_extracted), b = tlwe_new_sample(int2torus(2, 4), wkey_extracted);
  tlwe_mul(out, a, b, 4, wpack, rlk);
  WITHIN(1ULL << 58, int2torus(6, 4), tlwe_phase(out, wkey_extracted), "tlwe_mul");
  free_tlwe(a); free_tlwe(b); free_tlwe(out); free_polynomial(poly); free_trlwe_ks_key(rlk);
}

/* test_multivalue_bootstrap (test/tests.c:899-929 pattern): phase 1 once per input, phase 2 once per cleartext LUT */
static void case_multivalue_phases(void) {
  wide_setup();
  enum { tb = 4, log_tb = 2 };
  TRLWE rotated[tb + 1];
  for (int i = 0; i <= tb; i++) rotated[i] = trlwe_alloc_new_sample(k, N);
  TLWE out = tlwe_alloc_sample(N);
  int luts[3][tb] = {{0, 1, 2, 3}, {3, 1, 0, 2}, {2, 2, 1, 0}};
  for (int m = 0; m < tb; m++) {
    TLWE in = tlwe_new_sample(double2torus(m / 8.), lwe_key);
    multivalue_bootstrap_phase1(rotated, in, wbk, tb);
    for (int f = 0; f < 3; f++) {
      multivalue_bootstrap_phase2(out, luts[f], rotated, tb, log_tb);
      WITHIN(1ULL << 58, int2torus((uint64_t)luts[f][m], log_tb + 1), tlwe_phase(out, wkey_extracted), "multivalue_bootstrap_phase2");
    }
    free_tlwe(in);
  }
  for (int i = 0; i <= tb; i++) free_trlwe(rotated[i]);
  free_tlwe(out);
}

/* circuit_bootstrap_2 with the table-lookup private key switch (src/bootstrap.c:324-344, src/keyswitch.c:639-656), public_mux with
 * the fresh TRGSW rows as selector (src/bootstrap.c:369-389), and the TRGSW-accumulator bootstrap (:267-306) */
static void case_circuit_2_mux_trgsw(void) {
  wide_setup();
  TorusPolynomial ph = polynomial_new_torus_polynomial(N);
  TRLWE c2 = trlwe_alloc_new_sample(k, N);
  TLWE lw = tlwe_new_sample(double2torus(0.125), wkey_extracted);
  trlwe_priv_keyswitch(c2, lw, wpriv);
  trlwe_phase(ph, c2, wkey);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 56, (Torus)0 - wkey->s[0]->coeffs[i] * double2torus(0.125), ph->coeffs[i], "trlwe_priv_keyswitch");
  TRGSW sel[2] = {trgsw_alloc_new_sample(wl, wBg, k, N), trgsw_alloc_new_sample(wl, wBg, k, N)};
  TLWE in[2] = {tlwe_new_sample(double2torus(0.25), lwe_key), tlwe_new_sample(0, lwe_key)};
  circuit_bootstrap_2_batch(sel, in, 2, wbk, wpriv, wpack);
  TRGSW one = trgsw_alloc_new_sample(wl, wBg, k, N);
  circuit_bootstrap(one, in[0], wbk, wpriv, wpack);
  TorusPolynomial p0 = polynomial_new_torus_polynomial(N), p1 = polynomial_new_torus_polynomial(N);
  for (int i = 0; i < N; i++) { p0->coeffs[i] = int2torus((uint64_t)(i & 7), 3); p1->coeffs[i] = int2torus((uint64_t)((i >> 3) & 7), 3); }
  TRLWE_DFT *sel_dft = trlwe_alloc_new_DFT_sample_array(wl, k, N);
  for (int b = 0; b < 3; b++) {
    TRGSW g = b < 2 ? sel[b] : one;
    for (int i = 0; i < wl; i++) trlwe_to_DFT(sel_dft[i], g->samples[wl + i]);   /* the b rows of TRGSW(m) are the gadget encryption of m (src/bootstrap.c:417-420) */
    public_mux(c2, p0, p1, sel_dft, wl, wBg);
    trlwe_phase(ph, c2, wkey);
    const TorusPolynomial want = (b == 1) ? p0 : p1;       /* TRGSW(1) picks p1, TRGSW(0) picks p0 */
    for (int i = 0; i < N; i++) WITHIN(1ULL << 58, want->coeffs[i], ph->coeffs[i], "public_mux with circuit-bootstrapped selector");
  }
  /* full TRGSW bootstrap: LWE -> TRGSW_DFT(X^-phase), then any test vector by one external product */
  TRGSW_DFT acc = trgsw_alloc_new_DFT_sample(wl, wBg, k, N);
  Torus lut[4] = {int2torus(2, 4), int2torus(6, 4), int2torus(10, 4), int2torus(14, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE out = tlwe_alloc_sample(N);
  for (int j = 0; j < 4; j++) {
    TLWE c = tlwe_new_sample(double2torus(j / 8.), lwe_key);
    functional_bootstrap_trgsw_phase1(acc, c, wbk, 4);
    functional_bootstrap_trgsw_phase2(out, acc, tv);
    WITHIN(1ULL << 58, lut[j], tlwe_phase(out, wkey_extracted), "functional_bootstrap_trgsw");
    free_tlwe(c);
  }
  free_trgsw(acc); free_trlwe_array(sel_dft, wl); free_trlwe(tv); free_tlwe(out); free_polynomial(p0); free_polynomial(p1); free_trgsw(one);
  free_trgsw(sel[0]); free_trgsw(sel[1]); free_tlwe(in[0]); free_tlwe(in[1]); free_tlwe(lw); free_trlwe(c2); free_polynomial(ph);
}

/* The radix-integer addition of the reference's application layer (applications/multi-ciphertext-arith/src/integer.c:78-107,
 * ufhe_sl_add_integer): digits v / (2 torus_base) under the extracted key; per digit: add, key switch N -> n, bootstrap with the
 * constant ADDSUB test vector (ufhe.c:57-60), reduce the digit with trlwe_mv_extract_tlwe_scaling_subto and propagate the carry with
 * _scaling_addto -- exactly the call sequence of the reference, through the compat API. */
static void case_radix_integer_add(void) {
  wide_setup();
  enum { tb = 4, digits = 3 };
  TLWE_KS_Key ksk = tlwe_new_KS_key(lwe_key, wkey_extracted, 8, 2);
  TRLWE addsub = trlwe_alloc_new_sample(k, N), tmp2 = trlwe_alloc_new_sample(k, N);
  Torus c = double2torus(-1. / (4 * tb));
  trlwe_torus_packing(addsub, &c, 1);
  TLWE tmp = tlwe_alloc_sample(n);
  const int av[digits] = {3, 2, 0}, bv[digits] = {2, 1, 0};        /* 0b1011 + 0b0110 = 0b10001 */
  const int want[digits] = {1, 0, 1};
  TLWE a[digits], b[digits], cd[digits];
  for (int i = 0; i < digits; i++) {
    a[i] = tlwe_new_sample(double2torus((double)av[i] / (2 * tb)), wkey_extracted);
    b[i] = tlwe_new_sample(double2torus((double)bv[i] / (2 * tb)), wkey_extracted);
    cd[i] = tlwe_new_noiseless_trivial_sample(0, N);
  }
  for (int i = 0; i < digits; i++) {
    tlwe_addto(cd[i], a[i]);
    tlwe_addto(cd[i], b[i]);
    tlwe_keyswitch(tmp, cd[i], ksk);
    functional_bootstrap_wo_extract(tmp2, addsub, tmp, wbk, tb);
    trlwe_mv_extract_tlwe_scaling_subto(cd[i], tmp2, tb);
    cd[i]->b -= double2torus(1. / 4);
    if (i != digits - 1) {
      tlwe_noiseless_trivial_sample(cd[i + 1], double2torus(1. / (tb * 4)));
      trlwe_mv_extract_tlwe_scaling_addto(cd[i + 1], tmp2, 1);
    }
  }
  for (int i = 0; i < digits; i++)
    WITHIN(1ULL << 58, double2torus((double)want[i] / (2 * tb)), tlwe_phase(cd[i], wkey_extracted), "radix-integer addition digit");
  for (int i = 0; i < digits; i++) { free_tlwe(a[i]); free_tlwe(b[i]); free_tlwe(cd[i]); }
  free_tlwe(tmp); free_trlwe(addsub); free_trlwe(tmp2); free_tlwe_ks_key(ksk);
}

/* on-disk formats (src/bootstrap.c:63-104, src/tlwe.c:43-99,247-287, src/trlwe.c:24-43,230-251, src/keyswitch.c:122-160,409-455): every key
 * written to a file, read back, and used beside the original -- the results must be IDENTICAL, the operations are deterministic in the key */
static int same_tlwe(TLWE a, TLWE b) { return a->n == b->n && a->b == b->b && !memcmp(a->a, b->a, sizeof(Torus) * a->n); }
static int same_trlwe(TRLWE a, TRLWE b) {
  return !memcmp(a->a[0]->coeffs, b->a[0]->coeffs, sizeof(Torus) * N) && !memcmp(a->b->coeffs, b->b->coeffs, sizeof(Torus) * N);
}
static void case_key_files(void) {
  wide_setup();
  FILE *fd = tmpfile();
  CHECK(fd != NULL, "tmpfile");
  if (!fd) return;
  TLWE_KS_Key ksk = tlwe_new_KS_key(lwe_key, extracted_key, ks_t, ks_bb);
  TRLWE_KS_Key *priv2 = trlwe_new_priv_KS_key(wkey, wkey, 20, 2);
  Bootstrap_Key ubk = new_bootstrap_key(wgkey, lwe_key, 2);
  TLWE ct = tlwe_new_sample(double2torus(0.125), lwe_key);
  TorusPolynomial msg = polynomial_new_torus_polynomial(N);
  for (int i = 0; i < N; i++) msg->coeffs[i] = double2torus((i % 8) / 8.);
  TRLWE rct = trlwe_alloc_new_sample(k, N);
  trlwe_sample(rct, msg, wkey);
  TRGSW g = trgsw_alloc_new_sample(wl, wBg, k, N);
  trgsw_monomial_sample(g, 1, 3, wgkey);
  /* write everything into one file ... */
  tlwe_save_key(fd, lwe_key); trlwe_save_key(fd, wkey); trgsw_save_key(fd, wgkey);
  tlwe_save_sample(fd, ct); trlwe_save_sample(fd, rct); trgsw_save_sample(fd, g);
  save_bootstrap_key(fd, bk); save_bootstrap_key(fd, ubk);
  tlwe_save_KS_key(fd, ksk);
  trlwe_save_KS_key(fd, priv2[0]); trlwe_save_KS_key(fd, priv2[1]);
  trlwe_save_generic_ks_key(fd, wpack); trlwe_save_generic_ks_key(fd, wpriv);
  const long bytes = ftell(fd);
  rewind(fd);
  /* ... and read it back */
  TLWE_Key lwe2 = tlwe_load_new_key(fd);
  TRLWE_Key wkey2 = trlwe_load_new_key(fd);
  TRGSW_Key wgkey2 = trgsw_load_new_key(fd);
  TLWE ct2 = tlwe_load_new_sample(fd, n);
  TRLWE rct2 = trlwe_load_new_sample(fd, k, N);
  TRGSW g2 = trgsw_load_new_sample(fd, wl, wBg, k, N);
  Bootstrap_Key bk2 = load_new_bootstrap_key(fd), ubk2 = load_new_bootstrap_key(fd);
  TLWE_KS_Key ksk2 = tlwe_load_new_KS_key(fd);
  TRLWE_KS_Key priv2b[2];
  priv2b[0] = trlwe_load_new_KS_key(fd); priv2b[1] = trlwe_load_new_KS_key(fd);
  Generic_KS_Key wpack2 = trlwe_load_new_generic_ks_key(fd), wpriv2 = trlwe_load_new_generic_ks_key(fd);
  CHECK(ftell(fd) == bytes && fgetc(fd) == EOF, "the readers consumed %ld of %ld bytes", ftell(fd), bytes);
  fclose(fd);
  /* host objects */
  CHECK(lwe2->n == n && lwe2->sigma == lwe_key->sigma && !memcmp(lwe2->s, lwe_key->s, sizeof(Torus) * n), "tlwe key");
  CHECK(wkey2->k == k && wkey2->sigma == wkey->sigma && !memcmp(wkey2->s[0]->coeffs, wkey->s[0]->coeffs, sizeof(Torus) * N), "trlwe key");
  CHECK(wgkey2->l == wl && wgkey2->Bg_bit == wBg && !memcmp(wgkey2->trlwe_key->s[0]->coeffs, wkey->s[0]->coeffs, sizeof(Torus) * N), "trgsw key");
  CHECK(same_tlwe(ct, ct2), "tlwe sample");
  CHECK(same_trlwe(rct, rct2), "trlwe sample");
  for (int q = 0; q < 2 * wl; q++) CHECK(same_trlwe(g->samples[q], g2->samples[q]), "trgsw sample row %d", q);
  /* bootstrap keys: plain and unfolded */
  Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE o1 = tlwe_alloc_sample(N), o2 = tlwe_alloc_sample(N);
  CHECK(bk2->n == bk->n && bk2->l == bk->l && bk2->k == bk->k && bk2->N == bk->N && bk2->Bg_bit == bk->Bg_bit && bk2->unfolding == 1, "bootstrap key header");
  functional_bootstrap(o1, tv, ct, bk, 4); functional_bootstrap(o2, tv, ct, bk2, 4);
  CHECK(same_tlwe(o1, o2), "bootstrap with the reloaded key differs");
  WITHIN(1ULL << 58, lut[1], tlwe_phase(o2, extracted_key), "bootstrap with the reloaded key");
  CHECK(ubk2->unfolding == 2, "unfolded key header");
  functional_bootstrap(o1, tv, ct, ubk, 4); functional_bootstrap(o2, tv, ct, ubk2, 4);
  CHECK(same_tlwe(o1, o2), "bootstrap with the reloaded unfolded key differs");
  /* LWE key switch (the loaded key also has the host table view) */
  TLWE big = tlwe_new_sample(double2torus(0.375), extracted_key), s1 = tlwe_alloc_sample(n), s2 = tlwe_alloc_sample(n);
  tlwe_keyswitch(s1, big, ksk); tlwe_keyswitch(s2, big, ksk2);
  CHECK(same_tlwe(s1, s2), "key switch with the reloaded key differs");
  CHECK(ksk2->n == ksk->n && ksk2->t == ks_t && ksk2->base_bit == ks_bb && same_tlwe(ksk->s[N - 1][ks_t - 1][2], ksk2->s[N - 1][ks_t - 1][2]), "key-switch key host view");
  /* FFT-based private key switch (two entries of one device key set, saved one by one) */
  TRLWE r1 = trlwe_alloc_new_sample(k, N), r2 = trlwe_alloc_new_sample(k, N);
  trlwe_priv_keyswitch_2(r1, rct, priv2);
  trlwe_keyswitch(r2, rct, priv2b[0]); trlwe_keyswitch(r1, rct, priv2[0]);
  CHECK(same_trlwe(r1, r2), "trlwe_keyswitch with reloaded key 0 differs");
  trlwe_keyswitch(r2, rct, priv2b[1]); trlwe_keyswitch(r1, rct, priv2[1]);
  CHECK(same_trlwe(r1, r2), "trlwe_keyswitch with reloaded key 1 differs");
  /* table-lookup TRLWE keys */
  TLWE lw = tlwe_new_sample(double2torus(0.125), wkey_extracted);
  trlwe_packing1_keyswitch(r1, lw, wpack); trlwe_packing1_keyswitch(r2, lw, wpack2);
  CHECK(same_trlwe(r1, r2), "packing key switch with the reloaded key differs");
  trlwe_priv_keyswitch(r1, lw, wpriv); trlwe_priv_keyswitch(r2, lw, wpriv2);
  CHECK(same_trlwe(r1, r2), "private key switch with the reloaded key differs");
  CHECK(wpack2->n == wpack->n && wpack2->t == wpack->t && wpack2->base_bit == wpack->base_bit && wpack2->include_b == 0 && wpriv2->include_b == 1, "generic key header");
  free_tlwe(lw); free_trlwe(r1); free_trlwe(r2); free_tlwe(big); free_tlwe(s1); free_tlwe(s2); free_tlwe(o1); free_tlwe(o2); free_trlwe(tv);
  free_trlwe_generic_ks_key(wpack2); free_trlwe_generic_ks_key(wpriv2); free_trlwe_ks_key(priv2b[0]); free_trlwe_ks_key(priv2b[1]);
  free_tlwe_ks_key(ksk2); free_bootstrap_key(bk2); free_bootstrap_key(ubk2); free_trgsw(g2); free_trlwe(rct2); free_tlwe(ct2);
  free_trlwe_key(wgkey2->trlwe_key); free_trgsw_key(wgkey2); free_trlwe_key(wkey2); free_tlwe_key(lwe2);
  free_trgsw(g); free_trlwe(rct); free_polynomial(msg); free_tlwe(ct); free_bootstrap_key(ubk);
  free_trlwe_ks_key(priv2[0]); free_trlwe_ks_key(priv2[1]); free(priv2); free_tlwe_ks_key(ksk);
}

/* Re-entrancy (SURVEY 8(b) threading; the reference is re-entrant through thread-local FFT state, src/polynomial.c:269-352): several host threads
 * run bootstraps, key switches and the compositions that need device temporaries AT THE SAME TIME on the SAME keys.  Every operation is
 * deterministic in (key, input), so each thread must reproduce, bit for bit, what the main thread computed alone beforehand. */
enum { TH = 4, TH_CT = 6, TH_LUTS = 2, TH_ROUNDS = 3 };
typedef struct {
  int id, bad;
  TLWE *in, *big;                                  /* TH_CT inputs under lwe_key / extracted_key */
  TLWE *want_pbs, *want_ks, *want_fdfb, *want_mv;  /* results of the single-threaded pass */
  TRGSW want_cb;
  TLWE_KS_Key ksk;
  TRLWE tv, tv_fdfb, tv_mv;
} ThreadJob;
static int tlwe_arrays_differ(TLWE *a, TLWE *b, int count) {
  int bad = 0;
  for (int i = 0; i < count; i++) bad += !same_tlwe(a[i], b[i]);
  return bad;
}
static void thread_ops(ThreadJob *j, TLWE *pbs, TLWE *ks, TLWE *fdfb, TLWE *mv, TRGSW cb) {
  programmable_bootstrap_batch(pbs, j->tv, j->in, TH_CT, bk, 3, 0, 0);
  tlwe_keyswitch_batch(ks, j->big, TH_CT, j->ksk);
  full_domain_functional_bootstrap_batch(fdfb, j->tv_fdfb, j->in, TH_CT, bk, j->ksk, 3);
  multivalue_bootstrap_CLOT21(mv, j->tv_mv, j->in[j->id % TH_CT], bk, 4, TH_LUTS);
  circuit_bootstrap_2_batch(&cb, &j->in[(j->id + 1) % TH_CT], 1, wbk, wpriv, wpack);
}
static void *thread_main(void *arg) {
  ThreadJob *j = (ThreadJob *)arg;
  TLWE *pbs = tlwe_alloc_sample_array(TH_CT, N), *ks = tlwe_alloc_sample_array(TH_CT, n), *fdfb = tlwe_alloc_sample_array(TH_CT, N);
  TLWE *mv = tlwe_alloc_sample_array(TH_LUTS, N);
  TRGSW cb = trgsw_alloc_new_sample(wl, wBg, k, N);
  for (int r = 0; r < TH_ROUNDS; r++) {
    thread_ops(j, pbs, ks, fdfb, mv, cb);
    j->bad += tlwe_arrays_differ(pbs, j->want_pbs, TH_CT) + tlwe_arrays_differ(ks, j->want_ks, TH_CT) + tlwe_arrays_differ(fdfb, j->want_fdfb, TH_CT) +
              tlwe_arrays_differ(mv, j->want_mv, TH_LUTS);
    for (int q = 0; q < 2 * wl; q++) j->bad += !same_trlwe(cb->samples[q], j->want_cb->samples[q]);
    /* each thread draws from its own random stream: fresh samples must decrypt */
    TLWE fresh = tlwe_new_sample(double2torus(0.125), lwe_key);
    j->bad += tdist(tlwe_phase(fresh, lwe_key), double2torus(0.125)) >= (1ULL << 54);
    free_tlwe(fresh);
  }
  free_trgsw(cb); free_tlwe_array(mv, TH_LUTS); free_tlwe_array(fdfb, TH_CT); free_tlwe_array(ks, TH_CT); free_tlwe_array(pbs, TH_CT);
  return NULL;
}
static void case_threads(void) {
  wide_setup();
  TLWE_KS_Key ksk = tlwe_new_KS_key(lwe_key, extracted_key, ks_t, ks_bb);
  Torus lut[4] = {int2torus(3, 4), int2torus(7, 4), int2torus(11, 4), int2torus(15, 4)}, lut8[8];
  for (int i = 0; i < 8; i++) lut8[i] = int2torus((uint64_t)((3 * i + 1) & 7), 3);
  TRLWE tv = trlwe_alloc_new_sample(k, N), tv_fdfb = trlwe_alloc_new_sample(k, N), tv_mv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  trlwe_torus_packing_many_LUT(tv_fdfb, lut8, 4, 2);
  trlwe_torus_packing_many_LUT(tv_mv, lut8, 4, TH_LUTS);
  ThreadJob jobs[TH];
  pthread_t th[TH];
  for (int t = 0; t < TH; t++) {
    ThreadJob *j = &jobs[t];
    j->id = t; j->bad = 0; j->ksk = ksk; j->tv = tv; j->tv_fdfb = tv_fdfb; j->tv_mv = tv_mv;
    j->in = tlwe_alloc_sample_array(TH_CT, n); j->big = tlwe_alloc_sample_array(TH_CT, N);
    for (int i = 0; i < TH_CT; i++) {
      tlwe_sample(j->in[i], double2torus(((i + t) % 4) / 8.), lwe_key);
      tlwe_sample(j->big[i], double2torus(((i + 2 * t) % 8) / 8.), extracted_key);
    }
    j->want_pbs = tlwe_alloc_sample_array(TH_CT, N); j->want_ks = tlwe_alloc_sample_array(TH_CT, n);
    j->want_fdfb = tlwe_alloc_sample_array(TH_CT, N); j->want_mv = tlwe_alloc_sample_array(TH_LUTS, N);
    j->want_cb = trgsw_alloc_new_sample(wl, wBg, k, N);
    thread_ops(j, j->want_pbs, j->want_ks, j->want_fdfb, j->want_mv, j->want_cb);
    for (int i = 0; i < TH_CT; i++) WITHIN(1ULL << 58, lut[(i + t) % 4], tlwe_phase(j->want_pbs[i], extracted_key), "single-threaded pass");
  }
  for (int t = 0; t < TH; t++) CHECK(!pthread_create(&th[t], NULL, thread_main, &jobs[t]), "pthread_create");
  for (int t = 0; t < TH; t++) {
    pthread_join(th[t], NULL);
    CHECK(jobs[t].bad == 0, "thread %d: %d results differ from the single-threaded pass", t, jobs[t].bad);
    ThreadJob *j = &jobs[t];
    free_trgsw(j->want_cb); free_tlwe_array(j->want_mv, TH_LUTS); free_tlwe_array(j->want_fdfb, TH_CT); free_tlwe_array(j->want_ks, TH_CT);
    free_tlwe_array(j->want_pbs, TH_CT); free_tlwe_array(j->big, TH_CT); free_tlwe_array(j->in, TH_CT);
  }
  free_trlwe(tv); free_trlwe(tv_fdfb); free_trlwe(tv_mv); free_tlwe_ks_key(ksk);
}

/* the reference's default parameter set SET_2 (test/tests.c:43-45: N = 2048, l = 1, Bg = 2^23) and its largest, SET_3 (:47-49: N = 4096, l = 1,
 * Bg = 2^22), with a shortened LWE key: bootstraps, the batch entry, the key switch back and the full-domain bootstrap on those rings */
static void case_other_rings(void) {
  /* the reference's larger rings (tuned kernels), then what only the general path serves (csrc/general_kernels.h): k = 2 and rings outside 1024 .. 4096 */
  static const struct { int N, k, l, Bg_bit, n; double sigma; } sets[5] = {{2048, 1, 1, 23, 64, 2.2148688116005568e-16}, {4096, 1, 1, 22, 64, 2.2148688116005568e-16},
                                                                           {512, 2, 2, 8, 24, 2.2148688116005568e-16}, {8192, 1, 1, 23, 12, 2.2148688116005568e-16},
                                                                           {1024, 2, 2, 10, 16, 2.2148688116005568e-16}};
  for (int q = 0; q < 5; q++) {
    const int N2 = sets[q].k * sets[q].N, l2 = sets[q].l, Bg2 = sets[q].Bg_bit, n2 = sets[q].n;   /* N2: dimension of the extracted LWE key */
    TLWE_Key lk = tlwe_new_binary_key(n2, 1.0e-7);
    TRLWE_Key rk = trlwe_new_binary_key(sets[q].N, sets[q].k, sets[q].sigma);
    TLWE_Key xk = tlwe_alloc_key(N2, rk->sigma);
    trlwe_extract_tlwe_key(xk, rk);
    TRGSW_Key gk = trgsw_new_key(rk, l2, Bg2);
    Bootstrap_Key b2 = new_bootstrap_key(gk, lk, 1);
    TLWE_KS_Key ks2 = tlwe_new_KS_key(lk, xk, 5, 3);
    Torus lut[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)}, lut8[8];
    for (int i = 0; i < 8; i++) lut8[i] = int2torus((uint64_t)((3 * i + 1) & 7), 3);
    TRLWE tv = trlwe_alloc_new_sample(sets[q].k, sets[q].N), tv8 = trlwe_alloc_new_sample(sets[q].k, sets[q].N);
    trlwe_torus_packing(tv, lut, 4);
    trlwe_torus_packing_many_LUT(tv8, lut8, 4, 2);
    enum { COUNT = 9 };
    TLWE *in = tlwe_alloc_sample_array(COUNT, n2), *out = tlwe_alloc_sample_array(COUNT, N2), *back = tlwe_alloc_sample_array(COUNT, n2);
    for (int i = 0; i < COUNT; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lk);
    programmable_bootstrap_batch(out, tv, in, COUNT, b2, 3, 0, 0);
    for (int i = 0; i < COUNT; i++) WITHIN(1ULL << 58, lut[i % 4], tlwe_phase(out[i], xk), "programmable_bootstrap_batch on the larger rings");
    TLWE one = tlwe_alloc_sample(N2);
    functional_bootstrap(one, tv, in[3], b2, 4);
    CHECK(same_tlwe(one, out[3]), "N = %d, k = %d: functional_bootstrap differs from the programmable batch entry", sets[q].N, sets[q].k);
    tlwe_keyswitch_batch(back, out, COUNT, ks2);
    for (int i = 0; i < COUNT; i++) WITHIN(1ULL << 59, lut[i % 4], tlwe_phase(back[i], lk), "key switch back from the larger rings");
    for (int m = 0; m < 8; m++) {
      TLWE c = tlwe_new_sample(int2torus((uint64_t)m, 3), lk);
      full_domain_functional_bootstrap(one, tv8, c, b2, ks2, 3);
      WITHIN(1ULL << 58, lut8[m], tlwe_phase(one, xk), "full_domain_functional_bootstrap on the larger rings");
      free_tlwe(c);
    }
    free_tlwe(one); free_tlwe_array(in, COUNT); free_tlwe_array(out, COUNT); free_tlwe_array(back, COUNT); free_trlwe(tv); free_trlwe(tv8);
    free_tlwe_ks_key(ks2); free_bootstrap_key(b2); free_trgsw_key(gk); free_tlwe_key(xk); free_trlwe_key(rk); free_tlwe_key(lk);
  }
}

/* batches past two pipeline chunks (the layer overlaps host packing / unpacking and the copies with the kernels, csrc/host/mosfhet_compat.c:
 * bootstrap_pipelined): every output decrypts, and the outputs equal those of the same samples bootstrapped one by one */
static void case_big_batch(void) {
  enum { COUNT = 2 * 2048 + 37 };
  Torus lut[4] = {int2torus(3, 4), int2torus(7, 4), int2torus(11, 4), int2torus(15, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  TLWE *in = tlwe_alloc_sample_array(COUNT, n), *out = tlwe_alloc_sample_array(COUNT, N);
  for (int i = 0; i < COUNT; i++) tlwe_sample(in[i], double2torus((i % 4) / 8.), lwe_key);
  programmable_bootstrap_batch(out, tv, in, COUNT, bk, 3, 0, 0);
  int bad = 0;
  for (int i = 0; i < COUNT; i++) bad += tdist(lut[i % 4], tlwe_phase(out[i], extracted_key)) >= (1ULL << 58);
  CHECK(bad == 0, "%d of %d outputs of the pipelined batch do not decrypt", bad, COUNT);
  const int probe[6] = {0, 2047, 2048, 4095, 4096, COUNT - 1};
  TLWE one = tlwe_alloc_sample(N);
  for (int q = 0; q < 6; q++) {
    programmable_bootstrap(one, tv, in[probe[q]], bk, 3, 0, 0);
    CHECK(same_tlwe(one, out[probe[q]]), "sample %d of the pipelined batch differs from its single call", probe[q]);
  }
  functional_bootstrap_batch(out, tv, in, COUNT, bk, 4);
  functional_bootstrap(one, tv, in[2049], bk, 4);
  CHECK(same_tlwe(one, out[2049]), "functional_bootstrap_batch (pipelined) differs from the single call");
  free_tlwe(one); free_tlwe_array(in, COUNT); free_tlwe_array(out, COUNT); free_trlwe(tv);
}

/* The reference's DFT-level API (SURVEY 8(b) must-keep signatures), used the way applications/leveled_lut/vertical_packing.c:9-52 uses it:
 * selectors encrypted bit by bit as TRGSW samples and moved to the DFT domain (trgsw_to_DFT), a CMUX tree made of trlwe_sub + trgsw_mul_trlwe_DFT +
 * trlwe_from_DFT + trlwe_add, then blind_rotate over the remaining selectors and a sample extract.  With MOSFHET_SUITE_DUMP set, inputs and the final
 * accumulator are written to that file so that tests/test_gpu_parity.py can recompute them with the oracle (bit for bit). */
static void vp_cmux(TRLWE out, TRLWE in1, TRLWE in2, TRGSW_DFT selector) {
  TRLWE_DFT tmp = trlwe_alloc_new_DFT_sample(out->k, out->b->N);
  TRLWE tmp2 = trlwe_alloc_new_sample(out->k, out->b->N);
  trlwe_sub(tmp2, in2, in1);
  trgsw_mul_trlwe_DFT(tmp, tmp2, selector);
  trlwe_from_DFT(tmp2, tmp);
  trlwe_add(out, tmp2, in1);
  free_trlwe(tmp);
  free_trlwe(tmp2);
}

static void dump_trlwe(FILE *f, TRLWE c) {
  for (int p = 0; p < c->k; p++) fwrite(c->a[p]->coeffs, sizeof(Torus), (size_t)c->b->N, f);
  fwrite(c->b->coeffs, sizeof(Torus), (size_t)c->b->N, f);
}

static void case_dft_level_api(void) {
  enum { BITS = 13, OUT_PREC = 4, LUTS = (1 << BITS) / N, LOG_N = 10 };
  init_fft(N);
  const char *dump_path = getenv("MOSFHET_SUITE_DUMP");
  FILE *dump = dump_path ? fopen(dump_path, "wb") : NULL;
  /* --- polynomial level: DFT(a) * DFT(b) against the exact negacyclic product (test/tests.c:231-276: 2^40) */
  TorusPolynomial pa = polynomial_new_torus_polynomial(N), pb = polynomial_new_torus_polynomial(N), pc = polynomial_new_torus_polynomial(N);
  DFT_Polynomial *fd = polynomial_new_array_of_polynomials_DFT(N, 3);
  generate_random_bytes(sizeof(Torus) * N, (uint8_t *)pa->coeffs);
  for (int i = 0; i < N; i++) pb->coeffs[i] = (Torus)((int64_t)(i % 17) - 8);
  polynomial_torus_to_DFT(fd[0], pa);
  polynomial_torus_to_DFT(fd[1], pb);
  polynomial_mul_DFT(fd[2], fd[0], fd[1]);
  polynomial_mul_addto_DFT(fd[2], fd[0], fd[1]);          /* 2 a b */
  polynomial_DFT_to_torus(pc, fd[2]);
  for (int i = 0; i < N; i++) {
    Torus want = 0;
    for (int j = 0; j <= i; j++) want += pa->coeffs[j] * pb->coeffs[i - j];
    for (int j = i + 1; j < N; j++) want -= pa->coeffs[j] * pb->coeffs[N + i - j];
    WITHIN(1ULL << 40, 2 * want, pc->coeffs[i], "polynomial_mul_DFT + polynomial_mul_addto_DFT");
  }
  polynomial_DFT_to_torus(pc, fd[0]);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 14, pa->coeffs[i], pc->coeffs[i], "polynomial_torus_to_DFT / polynomial_DFT_to_torus round trip");
  free_array_of_polynomials(fd, 3);
  /* --- TRLWE level: trlwe_to_DFT / trlwe_from_DFT round trip */
  TRLWE ct = trlwe_new_sample(pa, rlwe_key), back = trlwe_alloc_new_sample(k, N);
  TRLWE_DFT ctd = trlwe_alloc_new_DFT_sample(k, N);
  trlwe_to_DFT(ctd, ct);
  trlwe_from_DFT(back, ctd);
  for (int i = 0; i < N; i++) {
    WITHIN(1ULL << 14, ct->a[0]->coeffs[i], back->a[0]->coeffs[i], "trlwe_to_DFT / trlwe_from_DFT (a)");
    WITHIN(1ULL << 14, ct->b->coeffs[i], back->b->coeffs[i], "trlwe_to_DFT / trlwe_from_DFT (b)");
  }
  free_trlwe(ctd); free_trlwe(back); free_trlwe(ct);
  /* --- vertical packing (encrypt_bits + eval_LUT of the reference application) */
  uint32_t input;
  generate_random_bytes(4, (uint8_t *)&input);
  input &= (1u << BITS) - 1;
  TRGSW bit = trgsw_alloc_new_sample(l, Bg_bit, k, N);
  TRGSW_DFT *sel = trgsw_alloc_new_DFT_sample_array(BITS, l, Bg_bit, k, N);
  if (dump) { int hdr[6] = {N, l, Bg_bit, BITS, LUTS, (int)input}; fwrite(hdr, sizeof(int), 6, dump); }
  for (int i = 0; i < BITS; i++) {
    trgsw_monomial_sample(bit, (input >> i) & 1, 0, trgsw_key);
    trgsw_to_DFT(sel[i], bit);
    if (dump) for (int r = 0; r < 2 * l; r++) dump_trlwe(dump, bit->samples[r]);
  }
  uint32_t *lut = (uint32_t *)safe_malloc(sizeof(uint32_t) << BITS);
  generate_random_bytes(sizeof(uint32_t) << BITS, (uint8_t *)lut);
  TRLWE *tab = trlwe_alloc_new_sample_array(LUTS, k, N);
  for (int i = 0; i < LUTS; i++) {
    trlwe_sample(tab[i], NULL, rlwe_key);
    for (int j = 0; j < N; j++) tab[i]->b->coeffs[j] += int2torus((lut[i * N + j] &= (1u << OUT_PREC) - 1), OUT_PREC);
    if (dump) dump_trlwe(dump, tab[i]);
  }
  for (int i = 0; i < BITS - LOG_N; i++) {
    const int half = 1 << (BITS - LOG_N - i - 1);
    for (int j = 0; j < half; j++) vp_cmux(tab[j], tab[j], tab[j + half], sel[BITS - i - 1]);
  }
  Torus a[32];
  for (int i = 0; i < LOG_N; i++) a[i] = int2torus(2 * N - (1 << i), LOG_N + 1);
  blind_rotate(tab[0], a, sel, LOG_N);
  if (dump) { dump_trlwe(dump, tab[0]); fclose(dump); }
  TLWE res = tlwe_alloc_sample(N);
  trlwe_extract_tlwe(res, tab[0], 0);
  const uint32_t got = (uint32_t)torus2int(tlwe_phase(res, extracted_key), OUT_PREC) & ((1u << OUT_PREC) - 1);
  CHECK(got == lut[input], "vertical packing: LUT[%u] = %u, evaluated %u", input, lut[input], got);
  /* the same blind rotation with selectors that are NOT one device block (gathered inside blind_rotate) */
  {
    TRGSW_DFT lone[2] = {trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N), trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N)};
    TRLWE t1 = trlwe_new_sample(pa, rlwe_key), t2 = trlwe_alloc_new_sample(k, N);
    trlwe_copy(t2, t1);
    for (int i = 0; i < 2; i++) { trgsw_monomial_sample(bit, i, 0, trgsw_key); trgsw_to_DFT(lone[i], bit); trgsw_to_DFT(sel[i], bit); }
    blind_rotate(t1, a, lone, 2);
    blind_rotate(t2, a, sel, 2);
    CHECK(!memcmp(t1->a[0]->coeffs, t2->a[0]->coeffs, sizeof(Torus) * N) && !memcmp(t1->b->coeffs, t2->b->coeffs, sizeof(Torus) * N),
          "blind_rotate over scattered selectors differs from the contiguous array");
    free_trgsw(lone[0]); free_trgsw(lone[1]); free_trlwe(t1); free_trlwe(t2);
  }
  /* elements of a DFT-domain array are freed ONE BY ONE, element 0 first, as callers of the reference may (each element is its own allocation there,
   * src/trgsw.c:82-88): the array's device block has to stay until its last element goes */
  {
    TRGSW_DFT *arr = trgsw_alloc_new_DFT_sample_array(3, l, Bg_bit, k, N);
    TRGSW_DFT single = trgsw_alloc_new_DFT_sample(l, Bg_bit, k, N);
    TRLWE t0 = trlwe_new_sample(pa, rlwe_key), t1 = trlwe_alloc_new_sample(k, N), t2 = trlwe_alloc_new_sample(k, N);
    TRLWE_DFT td = trlwe_alloc_new_DFT_sample(k, N);
    free_trgsw(arr[0]);
    trgsw_monomial_sample(bit, 1, 3, trgsw_key);
    trgsw_to_DFT(arr[2], bit);
    trgsw_to_DFT(single, bit);
    free_trgsw(arr[1]);
    trgsw_mul_trlwe_DFT(td, t0, arr[2]); trlwe_from_DFT(t1, td);
    trgsw_mul_trlwe_DFT(td, t0, single); trlwe_from_DFT(t2, td);
    CHECK(!memcmp(t1->a[0]->coeffs, t2->a[0]->coeffs, sizeof(Torus) * N) && !memcmp(t1->b->coeffs, t2->b->coeffs, sizeof(Torus) * N),
          "an array element used after element 0 was freed differs from a lone sample");
    free_trgsw(arr[2]); free(arr); free_trgsw(single); free_trlwe(td); free_trlwe(t0); free_trlwe(t1); free_trlwe(t2);
    DFT_Polynomial *pd = polynomial_new_array_of_polynomials_DFT(N, 2);
    free_polynomial(pd[0]);
    polynomial_torus_to_DFT(pd[1], pa);
    polynomial_DFT_to_torus(pc, pd[1]);
    for (int i = 0; i < N; i++) WITHIN(1ULL << 14, pa->coeffs[i], pc->coeffs[i], "array element 1 after element 0 was freed");
    free_polynomial(pd[1]); free(pd);
  }
  /* blind_rotate over the bootstrap key's own entries = functional_bootstrap_wo_extract without the first rotation */
  {
    TLWE in = tlwe_new_sample(double2torus(3 / 8.), lwe_key);
    Torus l4[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
    TRLWE tv = trlwe_alloc_new_sample(k, N), r1 = trlwe_alloc_new_sample(k, N), r2 = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing(tv, l4, 4);
    functional_bootstrap_wo_extract(r1, tv, in, bk, 4);
    trlwe_mul_by_xai(r2, tv, 2 * N - (int)torus2int(in->b + double2torus(1. / 16), LOG_N + 1));
    blind_rotate(r2, in->a, bk->s, bk->n);
    CHECK(!memcmp(r1->a[0]->coeffs, r2->a[0]->coeffs, sizeof(Torus) * N) && !memcmp(r1->b->coeffs, r2->b->coeffs, sizeof(Torus) * N),
          "blind_rotate(tv, a, key->s, n) differs from functional_bootstrap_wo_extract");
    free_tlwe(in); free_trlwe(tv); free_trlwe(r1); free_trlwe(r2);
  }
  /* Galois side: trlwe_eval_automorphism with a key of the set, blind_rotate_ga = functional_bootstrap_wo_extract_ga without the first rotation */
  {
    enum { NG = 24 };
    TLWE_Key short_key = tlwe_new_binary_key(NG, lwe_sigma);
    Bootstrap_GA_Key gk = new_bootstrap_key_ga(trgsw_key, short_key);
    const uint64_t gen = 5;
    TorusPolynomial msg = polynomial_new_torus_polynomial(N), want = polynomial_new_torus_polynomial(N), ph = polynomial_new_torus_polynomial(N);
    for (int i = 0; i < N; i++) msg->coeffs[i] = int2torus((uint64_t)(i & 3), 3);
    TRLWE c1 = trlwe_new_sample(msg, rlwe_key), c2 = trlwe_alloc_new_sample(k, N);
    trlwe_eval_automorphism(c2, c1, gen, gk->ak[(gen - 1) >> 1]);
    polynomial_permute(want, msg, gen);
    trlwe_phase(ph, c2, rlwe_key);
    for (int i = 0; i < N; i++) WITHIN(1ULL << 56, want->coeffs[i], ph->coeffs[i], "trlwe_eval_automorphism");   /* key-switch noise of a 2 x 2^8 gadget: ~2^52 */
    CHECK(inverse_mod_2N(5, N) * 5 % (2 * N) == 1, "inverse_mod_2N");
    TLWE in = tlwe_new_sample(double2torus(1 / 8.), short_key);
    Torus l4[4] = {int2torus(1, 4), int2torus(5, 4), int2torus(9, 4), int2torus(13, 4)};
    TRLWE tv = trlwe_alloc_new_sample(k, N), r1 = trlwe_alloc_new_sample(k, N), r2 = trlwe_alloc_new_sample(k, N);
    trlwe_torus_packing(tv, l4, 4);
    functional_bootstrap_wo_extract_ga(r1, tv, in, gk, 4);
    trlwe_mul_by_xai(r2, tv, 2 * N - (int)torus2int(in->b + double2torus(1. / 16), LOG_N + 1));
    blind_rotate_ga(r2, in->a, gk->s, gk->ak, gk->n);
    CHECK(!memcmp(r1->a[0]->coeffs, r2->a[0]->coeffs, sizeof(Torus) * N) && !memcmp(r1->b->coeffs, r2->b->coeffs, sizeof(Torus) * N),
          "blind_rotate_ga differs from functional_bootstrap_wo_extract_ga");
    free_tlwe(in); free_trlwe(tv); free_trlwe(r1); free_trlwe(r2); free_trlwe(c1); free_trlwe(c2);
    free_polynomial(msg); free_polynomial(want); free_polynomial(ph); free_bootstrap_key_ga(gk); free_tlwe_key(short_key);
  }
  free_tlwe(res); free_trlwe_array(tab, LUTS); free(lut); free_trgsw_array(sel, BITS); free_trgsw(bit);
  free_polynomial(pa); free_polynomial(pb); free_polynomial(pc);
}

/* The single-object helpers around the path (csrc/host/mosfhet_compat_legacy.c): DFT-domain arithmetic on device objects, TRGSW products, the
 * automorphism key sets and the unfolded blind rotation on caller-held samples, each checked by decryption or against the exact integer result. */
static void poly_phase_check_key(TRLWE c, TorusPolynomial want, uint64_t tol, const char *what, TRLWE_Key key) {
  TorusPolynomial ph = polynomial_new_torus_polynomial(N);
  trlwe_phase(ph, c, key);
  uint64_t worst = 0;
  for (int i = 0; i < N; i++) { const uint64_t d = tdist(want->coeffs[i], ph->coeffs[i]); if (d > worst) worst = d; }
  CHECK(worst < tol, "%s: worst phase error %016llx", what, (unsigned long long)worst);
  free_polynomial(ph);
}
static void poly_phase_check(TRLWE c, TorusPolynomial want, uint64_t tol, const char *what) { poly_phase_check_key(c, want, tol, what, rlwe_key); }
static void case_legacy_helpers(void) {
  TorusPolynomial m1 = polynomial_new_torus_polynomial(N), m2 = polynomial_new_torus_polynomial(N), sm = polynomial_new_torus_polynomial(N);
  TorusPolynomial want = polynomial_new_torus_polynomial(N), got = polynomial_new_torus_polynomial(N);
  generate_random_bytes(sizeof(Torus) * N, (uint8_t *)m1->coeffs);
  generate_random_bytes(sizeof(Torus) * N, (uint8_t *)m2->coeffs);
  for (int i = 0; i < N; i++) sm->coeffs[i] = (Torus)((int64_t)((i * 7) % 13) - 6);
  /* --- products through the transform against the exact schoolbook product (test_poly_DFT_mul, test/tests.c:244-276: 2^40) */
  polynomial_naive_mul_torus(want, m1, sm);
  polynomial_mul_torus(got, m1, sm);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 40, want->coeffs[i], got->coeffs[i], "polynomial_mul_torus");
  polynomial_copy_torus_polynomial(got, m2);
  polynomial_mul_addto_torus(got, m1, sm);
  polynomial_addto_torus_polynomial(want, m2);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 40, want->coeffs[i], got->coeffs[i], "polynomial_mul_addto_torus");
  /* --- DFT polynomials: a + b, a - b, a + 3 b */
  DFT_Polynomial *f = polynomial_new_array_of_polynomials_DFT(N, 3);
  polynomial_torus_to_DFT(f[0], m1);
  polynomial_torus_to_DFT(f[1], m2);
  polynomial_add_DFT_polynomials(f[2], f[0], f[1]);
  polynomial_DFT_to_torus(got, f[2]);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 16, m1->coeffs[i] + m2->coeffs[i], got->coeffs[i], "polynomial_add_DFT_polynomials");
  polynomial_sub_DFT_polynomials(f[2], f[0], f[1]);
  polynomial_DFT_to_torus(got, f[2]);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 16, m1->coeffs[i] - m2->coeffs[i], got->coeffs[i], "polynomial_sub_DFT_polynomials");
  polynomial_scale_and_add_DFT_polynomials(f[2], f[0], f[1], 3);
  polynomial_DFT_to_torus(got, f[2]);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 16, m1->coeffs[i] + 3 * m2->coeffs[i], got->coeffs[i], "polynomial_scale_and_add_DFT_polynomials");
  /* --- TRLWE_DFT arithmetic, by decryption */
  TRLWE c1 = trlwe_new_sample(m1, rlwe_key), c2 = trlwe_new_sample(m2, rlwe_key), r = trlwe_alloc_new_sample(k, N);
  TRLWE_DFT *d = trlwe_alloc_new_DFT_sample_array(3, k, N);
  TRLWE_DFT lone = trlwe_alloc_new_DFT_sample(k, N);
  trlwe_to_DFT(d[0], c1);
  trlwe_to_DFT(d[1], c2);
  trlwe_DFT_add(d[2], d[0], d[1]);
  trlwe_from_DFT(r, d[2]);
  polynomial_add_torus_polynomials(want, m1, m2);
  poly_phase_check(r, want, 1ULL << 45, "trlwe_DFT_add");
  trlwe_DFT_sub(lone, d[0], d[1]);
  trlwe_from_DFT(r, lone);
  polynomial_sub_torus_polynomials(want, m1, m2);
  poly_phase_check(r, want, 1ULL << 45, "trlwe_DFT_sub");
  trlwe_DFT_copy(lone, d[0]);
  trlwe_DFT_addto(lone, d[0]);
  trlwe_from_DFT(r, lone);
  polynomial_torus_scale2(want, m1, 2);
  poly_phase_check(r, want, 1ULL << 45, "trlwe_DFT_copy + trlwe_DFT_addto");
  trlwe_DFT_phase(got, d[1], rlwe_key);
  for (int i = 0; i < N; i++) WITHIN(1ULL << 45, m2->coeffs[i], got->coeffs[i], "trlwe_DFT_phase");
  polynomial_torus_to_DFT(f[2], sm);
  trlwe_DFT_mul_by_polynomial(d[2], d[0], f[2]);
  trlwe_DFT_mul_addto_by_polynomial(d[2], d[1], f[2]);
  trlwe_from_DFT(r, d[2]);
  polynomial_add_torus_polynomials(got, m1, m2);
  polynomial_naive_mul_torus(want, got, sm);
  poly_phase_check(r, want, 1ULL << 50, "trlwe_DFT_mul_by_polynomial + _mul_addto_by_polynomial");
  TRLWE_DFT triv = trlwe_new_noiseless_trivial_DFT_sample(f[0], k, N);
  trlwe_from_DFT(r, triv);
  int mask_zero = 1;
  for (int i = 0; i < N; i++) mask_zero &= r->a[0]->coeffs[i] == 0;
  CHECK(mask_zero, "trlwe_new_noiseless_trivial_DFT_sample: mask is not zero");
  for (int i = 0; i < N; i++) WITHIN(1ULL << 16, m1->coeffs[i], r->b->coeffs[i], "trlwe_new_noiseless_trivial_DFT_sample");
  free_trlwe(triv);
  /* --- TRGSW: constructors, DFT round trip, sums and products; checked through external products on a sample under the low-noise key of
   * wide_setup (l = 4, Bg = 2^9: a product of two encrypted TRGSW samples needs that headroom, as in the reference's own TRGSW tests) */
  wide_setup();
  TRLWE cw = trlwe_new_sample(m1, wkey);
#define WCHECK(tol, what) poly_phase_check_key(r, want, tol, what, wkey)
  TRGSW_DFT *g = trgsw_alloc_new_DFT_sample_array(4, wl, wBg, k, N);
  TRGSW e3 = trgsw_new_exp_sample(3, wgkey), e5 = trgsw_new_exp_sample(5, wgkey), one = trgsw_new_sample(1, wgkey);
  TRGSW tr = trgsw_new_noiseless_trivial_sample(1, wl, wBg, k, N), gt = trgsw_alloc_new_sample(wl, wBg, k, N);
  trgsw_to_DFT(g[0], e3);
  trgsw_from_DFT(gt, g[0]);
  for (int row = 0; row < 2 * wl; row++)
    for (int i = 0; i < N; i += 37) WITHIN(1ULL << 14, e3->samples[row]->b->coeffs[i], gt->samples[row]->b->coeffs[i], "trgsw_to_DFT / trgsw_from_DFT");
  trgsw_to_DFT(g[1], e5);
  trgsw_mul_DFT(g[2], e3, g[1]);                       /* TRGSW(X^3) x TRGSW_DFT(X^5) = TRGSW_DFT(X^8) */
  trgsw_mul_trlwe_DFT(d[2], cw, g[2]);
  trlwe_from_DFT(r, d[2]);
  torus_polynomial_mul_by_xai(want, m1, 8);
  WCHECK(1ULL << 50, "trgsw_mul_DFT");
  trgsw_mul_DFT2(g[3], g[0], g[1]);
  trgsw_mul_trlwe_DFT_prefetch(d[2], cw, g[3]);
  trlwe_from_DFT(r, d[2]);
  WCHECK(1ULL << 50, "trgsw_mul_DFT2");
  trgsw_DFT_add(g[3], g[0], g[1]);                      /* TRGSW_DFT(X^3 + X^5) */
  trgsw_mul_trlwe_DFT(d[2], cw, g[3]);
  trlwe_from_DFT(r, d[2]);
  torus_polynomial_mul_by_xai(want, m1, 3);
  torus_polynomial_mul_by_xai_addto(want, m1, 5);
  WCHECK(1ULL << 50, "trgsw_DFT_add");
  trgsw_DFT_sub(g[3], g[3], g[1]);                      /* back to X^3 */
  {   /* ... through a file and back (trgsw_save_DFT_sample / trgsw_load_new_DFT_sample, trlwe_*_DFT_sample inside) */
    FILE *fd = tmpfile();
    trgsw_save_DFT_sample(fd, g[3]);
    rewind(fd);
    TRGSW_DFT back = trgsw_load_new_DFT_sample(fd, wl, wBg, k, N);
    fclose(fd);
    trgsw_DFT_copy(g[3], back);
    free_trgsw(back);
  }
  trgsw_DFT_copy(g[2], g[3]);
  trgsw_mul_trlwe_DFT(d[2], cw, g[2]);
  trlwe_from_DFT(r, d[2]);
  torus_polynomial_mul_by_xai(want, m1, 3);
  WCHECK(1ULL << 50, "trgsw_DFT_sub + trgsw_DFT_copy");
  trgsw_monomial_DFT_sample(g[2], 1, N + 7, wgkey); /* X^(N+7) = -X^7 */
  trgsw_mul_trlwe_DFT(d[2], cw, g[2]);
  trlwe_from_DFT(r, d[2]);
  torus_polynomial_mul_by_xai(want, m1, N + 7);
  WCHECK(1ULL << 50, "trgsw_monomial_DFT_sample");
  trgsw_to_DFT(g[2], tr);                               /* the noise-free gadget of 1: an identity up to the digits' rounding */
  trgsw_mul_trlwe_DFT(d[2], cw, g[2]);
  trlwe_from_DFT(r, d[2]);
  poly_phase_check_key(r, m1, 1ULL << 40, "trgsw_new_noiseless_trivial_sample", wkey);
  trgsw_to_DFT(g[2], one);
  polynomial_torus_to_DFT(f[2], sm);
  trgsw_DFT_copy(g[3], g[2]);
  trgsw_DFT_mul_addto_by_polynomial(g[3], g[2], f[2]);  /* TRGSW_DFT(1 + sm) */
  trgsw_mul_trlwe_DFT(d[2], cw, g[3]);
  trlwe_from_DFT(r, d[2]);
  polynomial_naive_mul_torus(want, m1, sm);
  polynomial_addto_torus_polynomial(want, m1);
  WCHECK(1ULL << 50, "trgsw_DFT_mul_addto_by_polynomial");
  free_trgsw(e3); free_trgsw(e5); free_trgsw(one); free_trgsw(tr); free_trgsw(gt); free_trgsw_array(g, 4);
  /* --- automorphism key set: X -> X^5 through entry 2 of the odd-generator set (test_trlwe_ks tolerance, test/tests.c:794-830) */
  TRLWE_KS_Key *aks = trlwe_new_automorphism_KS_keyset(wkey, true, 3, 12);
  trlwe_eval_automorphism(r, cw, 5, aks[2]);
  polynomial_permute(want, m1, 5);
  WCHECK(1ULL << 50, "trlwe_new_automorphism_KS_keyset + trlwe_eval_automorphism");
  uint64_t gens[2] = {2 * N - 1, 9};
  TRLWE_KS_Key *aks2 = trlwe_new_automorphism_KS_keyset_2(wkey, gens, 2, 3, 12);
  trlwe_eval_automorphism(r, cw, 2 * N - 1, aks2[0]);
  polynomial_permute(want, m1, 2 * N - 1);
  WCHECK(1ULL << 50, "trlwe_new_automorphism_KS_keyset_2");
  for (int j = N - 1; j >= 0; j--) free_trlwe_ks_key(aks[j]);
  free(aks);
  free_trlwe_ks_key(aks2[1]); free_trlwe_ks_key(aks2[0]); free(aks2);
  /* --- the exact tensor product next to the FFT one (test_trlwe_mul, test/tests.c:1334-1372: 4-bit messages on the constant terms, product mod 16) */
  {
    TRLWE_KS_Key rlk = trlwe_new_RL_key(wkey, 2, 20);
    TRLWE p1 = trlwe_new_sample(NULL, wkey), p2 = trlwe_new_sample(NULL, wkey), po = trlwe_alloc_new_sample(k, N);
    TLWE px = tlwe_alloc_sample(N);
    for (int a = 3; a <= 13; a += 5) {
      const int b = 7;
      trlwe_sample(p1, NULL, wkey); trlwe_sample(p2, NULL, wkey);
      p1->b->coeffs[0] += int2torus(a, 4);
      p2->b->coeffs[0] += int2torus(b, 4);
      trlwe_tensor_prod(po, p1, p2, 4, rlk);
      trlwe_extract_tlwe(px, po, 0);
      CHECK((int)torus2int(tlwe_phase(px, wkey_extracted), 4) == ((a * b) & 15), "trlwe_tensor_prod: %d * %d", a, b);
      trlwe_tensor_prod_FFT(po, p1, p2, 4, rlk);
      trlwe_extract_tlwe(px, po, 0);
      CHECK((int)torus2int(tlwe_phase(px, wkey_extracted), 4) == ((a * b) & 15), "trlwe_tensor_prod_FFT: %d * %d", a, b);
    }
    free_trlwe(p1); free_trlwe(p2); free_trlwe(po); free_tlwe(px); free_trlwe_ks_key(rlk);
  }
  /* --- key switch without precomputed multiples (test_tlwe_ks's shape, test/tests.c:751-790, through tlwe_new_KS_key_no_precomp) */
  {
    TLWE_KS_Key_m mk = tlwe_new_KS_key_no_precomp(lwe_key, wkey_extracted, 8, 2);
    TLWE big = tlwe_alloc_sample(N), small = tlwe_alloc_sample(n);
    for (int j = 0; j < 4; j++) {
      tlwe_sample(big, double2torus(j / 8.), wkey_extracted);
      tlwe_keyswitch_no_precomp(small, big, mk);
      WITHIN(1ULL << 58, double2torus(j / 8.), tlwe_phase(small, lwe_key), "tlwe_keyswitch_no_precomp");
    }
    free_tlwe(big); free_tlwe(small);
  }
  /* --- unfolded blind rotation: the legacy entry points on caller-held objects */
  Torus lut[4] = {int2torus(3, 4), int2torus(7, 4), int2torus(11, 4), int2torus(15, 4)};
  TRLWE tv = trlwe_alloc_new_sample(k, N), acc = trlwe_alloc_new_sample(k, N);
  trlwe_torus_packing(tv, lut, 4);
  Bootstrap_Key ubk = new_bootstrap_key(wgkey, lwe_key, 2);
  TLWE in = tlwe_new_sample(double2torus(2 / 8.), lwe_key), out = tlwe_alloc_sample(N), ref = tlwe_alloc_sample(N);
  TRGSW_DFT *sa = trgsw_alloc_new_DFT_sample_array(n / 2, wl, wBg, k, N);
  multivalue_bootstrap_UBR_phase1(sa, in, ubk);
  multivalue_bootstrap_UBR_phase2(out, tv, in, sa, ubk, 4);
  WITHIN(1ULL << 58, lut[2], tlwe_phase(out, wkey_extracted), "multivalue_bootstrap_UBR_phase1 + _phase2");
  functional_bootstrap(ref, tv, in, ubk, 4);
  WITHIN(1ULL << 58, lut[2], tlwe_phase(ref, wkey_extracted), "functional_bootstrap (unfolded key)");
  TRGSW_DFT *sb = (TRGSW_DFT *)safe_malloc(sizeof(TRGSW_DFT) * (n / 2));   /* the same through samples allocated one by one */
  for (int i = 0; i < n / 2; i++) sb[i] = trgsw_alloc_new_DFT_sample(wl, wBg, k, N);
  multivalue_bootstrap_UBR_phase1(sb, in, ubk);
  multivalue_bootstrap_UBR_phase2(ref, tv, in, sb, ubk, 4);
  CHECK(ref->b == out->b && !memcmp(ref->a, out->a, sizeof(Torus) * N), "UBR phases through separately allocated samples differ from the array form");
  for (int i = 0; i < n / 2; i++) free_trgsw(sb[i]);
  free(sb);
  free_trgsw_array(sa, n / 2);
  /* blind_rotate_unfolded with the torus-domain samples held by the caller (new_bootstrap_key's su layout) */
  const int entries = n * 4 / 2;
  Torus *flat = (Torus *)safe_malloc(sizeof(Torus) * (size_t)entries * 2 * wl * 2 * N);
  mosfhet_gen_bootstrap_key_unfolded_flat(flat, wgkey, lwe_key, 2);
  TRGSW *su = trgsw_alloc_new_sample_array(entries, wl, wBg, k, N);
  for (int e = 0; e < entries; e++)
    for (int row = 0; row < 2 * wl; row++) {
      const Torus *src = flat + ((size_t)e * 2 * wl + row) * 2 * N;
      memcpy(su[e]->samples[row]->a[0]->coeffs, src, sizeof(Torus) * N);
      memcpy(su[e]->samples[row]->b->coeffs, src + N, sizeof(Torus) * N);
    }
  free(flat);
  const uint64_t bt = torus2int(in->b + double2torus(1. / 16), 11);
  trlwe_mul_by_xai(acc, tv, (int)((2 * N - bt) & (2 * N - 1)));
  blind_rotate_unfolded(acc, in->a, su, n, 2);
  trlwe_extract_tlwe(out, acc, 0);
  WITHIN(1ULL << 58, lut[2], tlwe_phase(out, wkey_extracted), "blind_rotate_unfolded");
  free_trgsw_array(su, entries);
  free_bootstrap_key(ubk);
  free_tlwe(in); free_tlwe(out); free_tlwe(ref); free_trlwe(tv); free_trlwe(acc);
  free_trlwe(c1); free_trlwe(c2); free_trlwe(r); free_trlwe(lone); free_trlwe_array(d, 3);
  free_array_of_polynomials(f, 3);
  free_polynomial(m1); free_polynomial(m2); free_polynomial(sm); free_polynomial(want); free_polynomial(got);
}

int main(int argc, char **argv) {
  setvbuf(stdout, NULL, _IOLBF, 0);
  mosfhet_seed(0x4D4F5346);
  lwe_key = tlwe_new_binary_key(n, lwe_sigma);
  rlwe_key = trlwe_new_binary_key(N, k, rlwe_sigma);
  trgsw_key = trgsw_new_key(rlwe_key, l, Bg_bit);
  extracted_key = tlwe_alloc_key(N, rlwe_sigma);
  trlwe_extract_tlwe_key(extracted_key, rlwe_key);
  bk = new_bootstrap_key(trgsw_key, lwe_key, 1);
  struct { const char *name; void (*fn)(void); } cases[] = {
    {"functional_bootstrap", case_functional_bootstrap}, {"programmable_bootstrap", case_programmable_bootstrap},
    {"blind_rotate", case_blind_rotate},                 {"tlwe_keyswitch+fdfb", case_tlwe_keyswitch},
    {"multivalue", case_multivalue},                     {"bootstrap_ga", case_bootstrap_ga},
    {"circuit_bootstrap", case_circuit_bootstrap},       {"unfolded", case_unfolded},
    {"fdfb_variants", case_fdfb_variants},               {"multivalue_phases", case_multivalue_phases},
    {"circuit_2+mux+trgsw", case_circuit_2_mux_trgsw},   {"radix_integer_add", case_radix_integer_add},
    {"key_files", case_key_files},                       {"threads", case_threads},
    {"other_rings", case_other_rings},                   {"big_batch", case_big_batch},
    {"dft_level_api", case_dft_level_api},                {"legacy_helpers", case_legacy_helpers},
  };
  for (unsigned i = 0; i < sizeof(cases) / sizeof(cases[0]); i++) {
    if (argc > 1 && strcmp(argv[1], cases[i].name)) continue;
    const int before = failures;
    cases[i].fn();
    printf("%-24s %s\n", cases[i].name, failures == before ? "ok" : "FAILED");
  }
  if (wkey) {
    free_trlwe_generic_ks_key(wpack); free_trlwe_generic_ks_key(wpriv); free_bootstrap_key(wbk); free_trgsw_key(wgkey);
    free_tlwe_key(wkey_extracted); free_trlwe_key(wkey);
  }
  free_bootstrap_key(bk); free_trgsw_key(trgsw_key); free_trlwe_key(rlwe_key); free_tlwe_key(lwe_key); free_tlwe_key(extracted_key);
  return failures > 255 ? 255 : failures;
}
