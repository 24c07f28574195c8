/*
 * CPU test helper for the on-disk formats of the host objects (include/mosfhet_compat.h): reads a file holding, in this order, the output of
 * tlwe_save_key, trlwe_save_key, trgsw_save_key, tlwe_save_sample, trlwe_save_sample -- written by the REFERENCE's own writers in
 * tests/test_host_and_abi.py -- with this library's readers, prints what it found, and writes the objects out again with this library's writers.
 * The test compares the two files byte for byte.  No GPU work: none of these objects lives on the device.
 * Mode "ks" does the same for a tlwe_save_KS_key file (src/tlwe.c:247-287); loading that key uploads its table, so it runs under -m gpu.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mosfhet_compat.h"

int main(int argc, char **argv) {
  if (argc != 4) return 2;
  FILE *in = fopen(argv[2], "rb"), *out = fopen(argv[3], "wb");
  if (!in || !out) return 3;
  if (!strcmp(argv[1], "ks")) {
    TLWE_KS_Key ks = tlwe_load_new_KS_key(in);
    if (fgetc(in) != EOF) return 4;
    printf("n=%d t=%d base_bit=%d n_out=%d\n", ks->n, ks->t, ks->base_bit, ks->s[0][0][0]->n);
    tlwe_save_KS_key(out, ks);
    fclose(in);
    fclose(out);
    free_tlwe_ks_key(ks);
    return 0;
  }
  TLWE_Key lk = tlwe_load_new_key(in);
  TRLWE_Key rk = trlwe_load_new_key(in);
  TRGSW_Key gk = trgsw_load_new_key(in);
  const int N = rk->s[0]->N;
  TLWE c = tlwe_load_new_sample(in, lk->n);
  TRLWE rc = trlwe_load_new_sample(in, rk->k, N);
  if (fgetc(in) != EOF) return 4;
  printf("n=%d lwe_sigma=%.17g k=%d N=%d rlwe_sigma=%.17g l=%d Bg_bit=%d gk.k=%d s0=%llu c.b=%llu rc.b0=%llu phase=%llu\n", lk->n, lk->sigma, rk->k, N, rk->sigma,
         gk->l, gk->Bg_bit, gk->trlwe_key->k, (unsigned long long)lk->s[0], (unsigned long long)c->b, (unsigned long long)rc->b->coeffs[0],
         (unsigned long long)tlwe_phase(c, lk));
  tlwe_save_key(out, lk);
  trlwe_save_key(out, rk);
  trgsw_save_key(out, gk);
  tlwe_save_sample(out, c);
  trlwe_save_sample(out, rc);
  fclose(in);
  fclose(out);
  free_tlwe(c); free_trlwe(rc); free_trlwe_key(gk->trlwe_key); free_trgsw_key(gk); free_trlwe_key(rk); free_tlwe_key(lk);
  return 0;
}
