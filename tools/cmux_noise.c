/* tools/cmux_noise.c -- phase error of a CMUX tree + blind rotation (the flow of the reference's applications/leveled_lut/vertical_packing.c, same parameters)
 * measured instead of rounded: prints log2 |phase - message| for REPS independent runs.  Links to this library or to the reference's (same API):
 *   gcc -O2 -Iinclude tools/cmux_noise.c -Lmosfhet_amd -lmosfhet_hip -lm        |  gcc -O2 -I/root/reference/include tools/cmux_noise.c oracle/_ref/libmosfhet_ref_avx512.so -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <mosfhet.h>

static void cmux(TRLWE out, TRLWE in1, TRLWE in2, TRGSW_DFT sel, TRLWE_DFT tmp) {
  trlwe_sub(out, in2, in1);
  trgsw_mul_trlwe_DFT(tmp, out, sel);
  trlwe_from_DFT(out, tmp);
  trlwe_addto(out, in1);
}

int main(int argc, char **argv) {
  const int N = 2048, k = 1, Bg_bit = 23, l = 1, BITS = 20, LOG_N = 11, OUT_PREC = 16, reps = argc > 1 ? atoi(argv[1]) : 20, in_bits = argc > 2 ? atoi(argv[2]) : 20;   /* in_bits: random low bits of the input (the application: 11) */
  const double sigma = 2.2148688116005568e-16;
  TRLWE_Key key = trlwe_new_binary_key(N, k, sigma);
  TRGSW_Key gkey = trgsw_new_key(key, l, Bg_bit);
  TLWE_Key okey = tlwe_alloc_key(N, sigma);
  trlwe_extract_tlwe_key(okey, key);
  const int n_luts = (1 << BITS) / N;
  TRLWE *tab = trlwe_alloc_new_sample_array(n_luts, k, N), *work = trlwe_alloc_new_sample_array(n_luts, k, N);
  TRGSW bit = trgsw_alloc_new_sample(l, Bg_bit, k, N);
  TRGSW_DFT *sel = trgsw_alloc_new_DFT_sample_array(BITS, l, Bg_bit, k, N);
  TRLWE_DFT tmp = trlwe_alloc_new_DFT_sample(k, N);
  TLWE res = tlwe_alloc_sample(N);
  uint32_t *lut = (uint32_t *)safe_aligned_malloc(sizeof(uint32_t) << BITS), *rnd = (uint32_t *)safe_aligned_malloc(64);   /* the reference's generator stores whole vectors */
  double sum2 = 0, worst = 0;
  int wrong = 0;
  for (int r = 0; r < reps; r++) {
    generate_random_bytes(64, (uint8_t *)rnd);
    uint32_t input = rnd[0];
    input &= (1u << in_bits) - 1;
    generate_random_bytes(sizeof(uint32_t) << BITS, (uint8_t *)lut);
    for (int i = 0; i < (1 << BITS); i++) lut[i] &= (1u << OUT_PREC) - 1;
    for (int i = 0; i < BITS; i++) {
      trgsw_monomial_sample(bit, (input >> i) & 1, 0, gkey);
      trgsw_to_DFT(sel[i], bit);
    }
    for (int i = 0; i < n_luts; i++) {
      trlwe_sample(tab[i], NULL, key);
      for (int j = 0; j < N; j++) tab[i]->b->coeffs[j] += int2torus(lut[(size_t)i * N + j], OUT_PREC);
    }
    for (int i = 0; i < BITS - LOG_N; i++) {          /* the tree over the top bits, in place: sample j keeps (bit ? sample j + half : sample j) */
      const int half = 1 << (BITS - LOG_N - i - 1);
      for (int j = 0; j < half; j++) {
        cmux(work[0], tab[j], tab[j + half], sel[BITS - i - 1], tmp);
        trlwe_copy(tab[j], work[0]);
      }
    }
    Torus *a = (Torus *)safe_aligned_malloc(sizeof(Torus) * 64);                                    /* the low bits rotate the surviving sample by X^(-2^i) per set bit: blind_rotate */
    for (int i = 0; i < LOG_N; i++) a[i] = int2torus(2 * N - (1 << i), LOG_N + 1);
    blind_rotate(tab[0], a, sel, LOG_N);
    trlwe_extract_tlwe(res, tab[0], 0);
    const Torus want = int2torus(lut[input], OUT_PREC), got = tlwe_phase(res, okey);
    const double err = fabs((double)(int64_t)(got - want));
    sum2 += err * err;
    if (err > worst) worst = err;
    wrong += torus2int(got, OUT_PREC) != lut[input];
  }
  printf("runs %d  rms phase error 2^%.2f  worst 2^%.2f  wrong at %d bits: %d\n", reps, log2(sqrt(sum2 / reps)), log2(worst), OUT_PREC, wrong);
  return 0;
}
