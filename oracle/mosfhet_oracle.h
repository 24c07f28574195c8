/*
 * mosfhet_oracle.h -- CPU ORACLE for the TFHE programmable-bootstrap hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The shipped
 * library (mosfhet_amd/csrc) never links, imports or calls anything in oracle/.
 *
 * What it is: a plain-C restatement of the algorithms of antoniocgj/MOSFHET on the
 * path  programmable_bootstrap -> functional_bootstrap -> blind_rotate ->
 * trgsw_mul_trlwe_DFT -> sample extract -> tlwe_keyswitch, over flat buffers.
 * Every function cites the reference file:line it follows.
 *
 * Parity status: PINNED.  The integer functions are checked bit-for-bit and the
 * floating-point ones within the reference's own tolerances against
 *   (a) the reference itself, compiled from /root/reference by oracle/ref/Makefile
 *       into oracle/_ref/ (FFNT pure-C and AVX-512 SPQLIOS builds), and
 *   (b) the golden vectors in tests/golden/ that were produced by that build
 *       (tests/golden/make_golden.py),
 * see tests/test_oracle_vs_reference.py and tests/test_oracle_golden.py.
 *
 * Floating point: the reference has three interchangeable FFT back-ends whose
 * results differ in the low ~26-30 bits.  The oracle's
 * negacyclic transform computes the same mathematical map (evaluation of the
 * folded polynomial at the roots of X^N = -1, "Torus -> DFT" of
 * src/fft/ffnt/ffnt.c:235-271,820-831) with one fixed, documented operation order
 * (oracle_fft.c).  The HIP kernels use exactly that order, so GPU-vs-oracle
 * comparisons are bit-exact even across a full n-step blind rotation, while
 * oracle-vs-reference comparisons use the reference's tolerances.
 *
 * Flat layouts (W = 64, Torus = uint64_t, all arithmetic mod 2^64):
 *   TLWE(n)            u64[n+1]                 a[0..n-1], b           (mosfhet.h:51-54)
 *   TRLWE(k,N)         u64[k+1][N]              a[0..k-1], b           (mosfhet.h:73-76)
 *   TRGSW(k,N,l)       u64[(k+1)l][k+1][N]      row p*l+j              (mosfhet.h:106-109, trgsw.c:152-168)
 *   bootstrap key      u64[n][(k+1)l][k+1][N]   torus domain           (bootstrap.c:14-18 before trgsw_to_DFT)
 *   LWE KS key         u64[n_in][t][2^bb-1][n_out+1]                   (tlwe.c:193-212)
 *   DFT polynomial     double[N]  = N/2 complex, interleaved (re,im), oracle slot order
 */
#ifndef MOSFHET_ORACLE_H
#define MOSFHET_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t Torus;
typedef struct { uint64_t s; } orc_rng;   /* splitmix64 state of the deterministic test-input generator */

/* ---- scalars (src/misc.c:13-28) ---- */
Torus    orc_double2torus(double x);
uint64_t orc_torus2int(Torus x, int log_scale);
Torus    orc_int2torus(uint64_t x, int log_scale);

/* ---- integer polynomial ops ---- */
void orc_poly_decompose_i(Torus *out, const Torus *in, int N, int Bg_bit, int l, int i);       /* polynomial.c:74-89 */
void orc_poly_decompose(Torus *out /*[l][N]*/, const Torus *in, int N, int Bg_bit, int l);     /* polynomial.c:55-72 */
void orc_poly_mul_by_xai(Torus *out, const Torus *in, int N, int a);                           /* polynomial.c:184-199 */
void orc_poly_mul_by_xai_addto(Torus *out, const Torus *in, int N, int a);                     /* polynomial.c:202-217 */
void orc_poly_mul_by_xai_minus_1(Torus *out, const Torus *in, int N, int a);                   /* polynomial.c:220-235 */
void orc_poly_naive_mul(Torus *out, const Torus *in1, const Torus *in2, int N);                /* polynomial.c:290-303 */
void orc_poly_naive_mul_addto(Torus *out, const Torus *in1, const Torus *in2, int N);          /* polynomial.c:264-274 */
void orc_poly_permute(Torus *out, const Torus *in, int N, uint64_t gen);                       /* polynomial.c:442-450 */
void orc_trlwe_extract_tlwe(Torus *out, const Torus *in, int k, int N, int idx);               /* trlwe.c:540-552 */
void orc_trlwe_torus_packing(Torus *out, const Torus *lut, int k, int N, int size);            /* trlwe.c:662-667 */
void orc_tlwe_keyswitch(Torus *out, const Torus *in, const Torus *ksk,
                        int n_in, int n_out, int t, int base_bit);                             /* tlwe.c:289-303 */
Torus orc_tlwe_phase(const Torus *c, const Torus *s, int n);                                   /* tlwe.c:135-141 */
void orc_trlwe_phase(Torus *out, const Torus *c, const Torus *s /*[k][N]*/, int k, int N);     /* trlwe.c:324-331 (exact, naive product) */
void orc_pbs_preprocess(Torus *out, const Torus *in, int n, int N, int kappa, int theta);      /* bootstrap.c:208-217 */

/* ---- negacyclic transform (oracle_fft.c) ---- */
typedef struct orc_fft_plan orc_fft_plan;        /* twiddles for one N */
orc_fft_plan *orc_fft_plan_new(int N);
void orc_fft_plan_free(orc_fft_plan *p);
const double *orc_fft_twiddles(const orc_fft_plan *p, int *count_complex); /* (re,im) pairs, level-major */
/* the table generator itself, so the product's table can be compared with it */
void orc_fft_make_twiddles(double *out /*[2*(N/2-1)]*/, int N);

void orc_torus_to_dft(const orc_fft_plan *p, double *out /*[N]*/, const Torus *in);   /* polynomial.c:368-375 -> execute_reverse_torus64 */
void orc_int_to_dft(const orc_fft_plan *p, double *out, const int64_t *in);          /* same, input already signed small ints */
void orc_dft_to_torus(const orc_fft_plan *p, Torus *out, const double *in);           /* polynomial.c:359-366 -> execute_direct_torus64 (AVX-512 rounding, fft_processor_spqlios.c:155-165) */
void orc_dft_mul(double *out, const double *a, const double *b, int N);               /* polynomial.c:379-401 */
void orc_dft_mul_addto(double *out, const double *a, const double *b, int N);         /* polynomial.c:406-426 */
void orc_poly_mul_fft(const orc_fft_plan *p, Torus *out, const Torus *a, const Torus *b); /* polynomial.c:276-288 */

/* ---- TRGSW / bootstrap ---- */
void orc_trgsw_to_dft(const orc_fft_plan *p, double *out /*[(k+1)l][k+1][N]*/,
                      const Torus *in, int k, int l);                                 /* trgsw.c:345-349 */
void orc_set_product_order(int order);   /* 0: the reference's one chain over all rows; 1: per input component partial sums, added (oracle_tfhe.c) */
int orc_get_product_order(void);
void orc_trgsw_mul_trlwe_dft(const orc_fft_plan *p, double *out_dft /*[k+1][N]*/, const Torus *in /*[k+1][N]*/,
                             const double *trgsw_dft, int k, int l, int Bg_bit);       /* trgsw.c:385-423 */
void orc_external_product(const orc_fft_plan *p, Torus *out, const Torus *in,
                          const double *trgsw_dft, int k, int l, int Bg_bit);          /* trgsw.c:385 + trlwe.c:629 */
void orc_blind_rotate(const orc_fft_plan *p, Torus *acc /*[k+1][N] in place*/, const Torus *a,
                      const double *bk_dft /*[n][(k+1)l][k+1][N]*/, int n, int k, int l, int Bg_bit); /* bootstrap.c:107-122 */
void orc_functional_bootstrap_wo_extract(const orc_fft_plan *p, Torus *out /*[k+1][N]*/, const Torus *tv,
                      const Torus *in /*[n+1]*/, const double *bk_dft, int n, int k, int l, int Bg_bit,
                      int torus_base);                                                 /* bootstrap.c:192-198 */
void orc_functional_bootstrap(const orc_fft_plan *p, Torus *out /*[kN+1]*/, const Torus *tv, const Torus *in,
                      const double *bk_dft, int n, int k, int l, int Bg_bit, int torus_base); /* bootstrap.c:200-206 */
void orc_programmable_bootstrap(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in,
                      const double *bk_dft, int n, int k, int l, int Bg_bit,
                      int precision, int kappa, int theta);                            /* bootstrap.c:208-220 */

void orc_full_domain_functional_bootstrap(const orc_fft_plan *p, Torus *out /*[kN+1]*/, const Torus *tv, const Torus *in /*[n+1]*/,
                      const double *bk_dft, const Torus *ksk /*[kN][t][2^bb-1][n+1]*/, int n, int k, int l, int Bg_bit,
                      int t, int base_bit, int precision);                             /* bootstrap.c:519-538 */
void orc_multivalue_bootstrap_CLOT21(const orc_fft_plan *p, Torus *out /*[n_luts][kN+1]*/, const Torus *tv, const Torus *in,
                      const double *bk_dft, int n, int k, int l, int Bg_bit, int torus_base, int n_luts); /* bootstrap.c:222-230 */
void orc_trlwe_torus_packing_many_LUT(Torus *out, const Torus *lut, int k, int N, int lut_size, int n_luts); /* trlwe.c:677-687 */

/* ---- FFT-based TRLWE key switch, Galois automorphisms, GA bootstrap (k = 1) ---- */
void orc_trlwe_keyswitch(const orc_fft_plan *p, Torus *out /*[2][N], may alias in*/, const Torus *in, const double *ks_dft /*[t][2][N]*/,
                         int t, int base_bit);                                          /* keyswitch.c:162-193 */
void orc_trlwe_eval_automorphism(const orc_fft_plan *p, Torus *out, const Torus *in, uint64_t gen, const double *ks_dft,
                                 int t, int base_bit);                                  /* trlwe.c:775-781 */
uint32_t orc_inverse_mod_2N(uint32_t x, int N);                                         /* misc.c:142-159 (odd x) */
void orc_blind_rotate_ga(const orc_fft_plan *p, Torus *acc, const Torus *a, const double *bk_dft, const double *ak_dft /*[N][t][2][N]*/,
                         int n, int l, int Bg_bit);                                     /* bootstrap_ga.c:39-60 (t = l, base_bit = Bg_bit) */
void orc_functional_bootstrap_wo_extract_ga(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                         const double *ak_dft, int n, int l, int Bg_bit, int torus_base); /* bootstrap_ga.c:62-68 */
void orc_functional_bootstrap_ga(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                         const double *ak_dft, int n, int l, int Bg_bit, int torus_base); /* bootstrap_ga.c:70-76 */
void orc_gen_trlwe_ks_key(orc_rng *r, Torus *ks /*[t][2][N]*/, const Torus *s_in /*[N]*/, const Torus *s_out /*[N]*/, int N, int t,
                          int base_bit, double sigma);                                  /* keyswitch.c:12-37 */
void orc_gen_automorphism_keyset(orc_rng *r, Torus *ak /*[N][t][2][N]*/, const Torus *s, int N, int t, int base_bit, double sigma); /* keyswitch.c:500-511, skip_even */
void orc_gen_bootstrap_key_ga(orc_rng *r, Torus *bk, const Torus *lwe_s, int n, const Torus *rlwe_s, int N, int l, int Bg_bit,
                              double sigma);                                            /* bootstrap_ga.c:17-20: BK_i = TRGSW(X^{s_i}) */

/* ---- circuit bootstrap (k = 1) ---- */
void orc_trlwe_packing1_keyswitch(Torus *out /*[2][N]*/, const Torus *in /*[n+1]*/, const Torus *ksk /*[n][t][2^bb-1][2][N]*/,
                                  int n, int N, int t, int base_bit);                    /* keyswitch.c:458-475 */
void orc_trlwe_priv_keyswitch_2(const orc_fft_plan *p, Torus *out, const Torus *in, const double *ks0_dft, const double *ks1_dft,
                                int t, int base_bit);                                    /* keyswitch.c:52-63 */
void orc_circuit_bootstrap_3(const orc_fft_plan *p, Torus *out /*[2l][2][N]*/, const Torus *in, const double *bk_dft, const double *kska0_dft,
                             const double *kska1_dft, int ta, int bba, const Torus *kskb, int tb, int bbb, int n, int l, int Bg_bit); /* bootstrap.c:346-366 */
void orc_gen_packing1_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in /*[n]*/, int n, const Torus *s_out /*[N]*/, int N, int t, int base_bit,
                             double sigma);                                              /* keyswitch.c:368-390 */
void orc_gen_priv_ks_key(orc_rng *r, Torus *ks0 /*[t][2][N]*/, Torus *ks1, const Torus *s_out, const Torus *s_in, int N, int t, int base_bit,
                         double sigma);                                                  /* keyswitch.c:39-50 */

/* ---- callers either side of the bootstrap (oracle_ext.c; k = 1) ---- */
void orc_public_mux(const orc_fft_plan *p, Torus *out /*[2][N]*/, const Torus *p0, const Torus *p1, const double *sel_dft /*[l][2][N]*/, int l,
                    int Bg_bit);                                                          /* bootstrap.c:369-389 */
void orc_full_domain_functional_bootstrap_KS21(const orc_fft_plan *p, Torus *out /*[N+1]*/, const Torus *tv /*[2N]*/, const Torus *in,
                    const double *bk_dft, const Torus *ksk /*[N][t][2^bb-1][2][N]*/, int n, int l, int Bg_bit, int t, int base_bit, int torus_base,
                    int variant /*0: _KS21, 1: _KS21_2*/);                                /* bootstrap.c:391-463 */
void orc_multivalue_bootstrap_phase1(const orc_fft_plan *p, Torus *out /*[torus_base+1][2][N]*/, const Torus *in, const double *bk_dft, int n, int l,
                    int Bg_bit, int torus_base);                                          /* bootstrap.c:232-243 */
void orc_trlwe_mv_extract_tlwe_scaling_addto(Torus *out /*[N+1]*/, const Torus *in /*[2][N]*/, int N, int scale); /* trlwe.c:603-611 */
void orc_multivalue_bootstrap_phase2(Torus *out /*[N+1]*/, const int *lut_in /*[torus_base]*/, const Torus *rotated_tv, int N, int torus_base,
                    int log_torus_base);                                                  /* bootstrap.c:245-265 */
void orc_gen_priv_sk_ks_key(orc_rng *r, Torus *ksk /*[n+1][t][2^bb-1][2][N]*/, const Torus *s_in, int n, const Torus *s_out, int N, int t,
                    int base_bit, double sigma);                                          /* keyswitch.c:611-637 */
void orc_trlwe_priv_keyswitch(Torus *out /*[2][N]*/, const Torus *in /*[n+1]*/, const Torus *ksk, int n, int N, int t, int base_bit); /* keyswitch.c:639-656 */
void orc_circuit_bootstrap(const orc_fft_plan *p, Torus *out /*[2l][2][N]*/, const Torus *in, const double *bk_dft, const Torus *kska, int ta, int bba,
                    const Torus *kskb, int tb, int bbb, int n, int l, int Bg_bit, int variant /*0: circuit_bootstrap, 1: _2*/); /* bootstrap.c:309-344 */
void orc_functional_bootstrap_trgsw_phase1(const orc_fft_plan *p, double *out_dft /*[2l][2][N]*/, const Torus *in, const double *bk_dft, int n, int l,
                    int Bg_bit, int torus_base);                                          /* bootstrap.c:267-295 */
void orc_functional_bootstrap_trgsw_phase2(const orc_fft_plan *p, Torus *out /*[N+1]*/, const double *in_dft, const Torus *tv, int l, int Bg_bit); /* bootstrap.c:297-306 */
void orc_gen_rl_key(orc_rng *r, Torus *ks /*[t][2][N]*/, const Torus *s, int N, int t, int base_bit, double sigma);   /* keyswitch.c:3-10 */
void orc_trlwe_tensor_prod_fft(const orc_fft_plan *p, Torus *out, const Torus *in1, const Torus *in2, int precision, const double *rl_dft, int t,
                    int base_bit);                                                        /* trlwe.c:727-771 */
void orc_tlwe_mul(const orc_fft_plan *p, Torus *out /*[N+1]*/, const Torus *in1, const Torus *in2, int precision, const Torus *ksk, int tk, int bbk,
                    const double *rl_dft, int tr, int bbr);                               /* tlwe.c:322-332 */
void orc_full_domain_functional_bootstrap_CLOT21(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *bk_dft,
                    const Torus *ksk, int tk, int bbk, const double *rl_dft, int tr, int bbr, int n, int l, int Bg_bit, int precision,
                    int variant /*0: _CLOT21 (tv = [2][2][N]), 1: _CLOT21_2 (tv = 2^(precision-1) LUT words)*/); /* bootstrap.c:465-517 */

/* ---- blind-rotate unfolding (oracle_ext.c) ---- */
void orc_gen_bootstrap_key_unfolded(orc_rng *r, Torus *su /*[n 2^u/u][2l][2][N]*/, const Torus *lwe_s, int n, const Torus *rlwe_s, int N, int l,
                    int Bg_bit, double sigma, int unfolding);                             /* bootstrap.c:23-48 */
void orc_blind_rotate_unfolded(const orc_fft_plan *p, Torus *acc, const Torus *a, const Torus *su, int n, int l, int Bg_bit, int unfolding); /* bootstrap.c:124-149 */
/* unfolding 2 with the per-group TRGSW assembled in the DFT domain (the GPU kernel's order; oracle_ext.c) */
void orc_monomial_table(double *out /*[2N][2]*/, int N);
void orc_unfold2_selector_dft(double *S, const double *K, Torus a0, Torus a1, int N, int l);
void orc_blind_rotate_unfolded2_dft(const orc_fft_plan *p, Torus *acc, const Torus *a, const double *su_dft, int n, int l, int Bg_bit);
void orc_functional_bootstrap_unfolded2_dft(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const double *su_dft, int n, int l, int Bg_bit,
                                            int torus_base, int extract);
void orc_functional_bootstrap_unfolded(const orc_fft_plan *p, Torus *out, const Torus *tv, const Torus *in, const Torus *su, int n, int l, int Bg_bit,
                    int torus_base, int unfolding, int extract);                          /* bootstrap.c:192-206 with key->unfolding > 1 */

void orc_multivalue_bootstrap_UBR_phase1(const orc_fft_plan *p, double *out_dft /*[n/u][2l][2][N]*/, const Torus *in, const Torus *su, int n, int l,
                    int Bg_bit, int unfolding);                                           /* bootstrap.c:151-175 */
void orc_multivalue_bootstrap_UBR_phase2(const orc_fft_plan *p, Torus *out /*[N+1]*/, const Torus *tv, const Torus *in, const double *sa_dft, int n,
                    int l, int Bg_bit, int unfolding, int torus_base);                    /* bootstrap.c:177-190 */

void orc_trlwe_mv_extract(Torus *out, const Torus *in /*[2][N]*/, int N, int mode /*0 mv_extract, 1 scaling, 2 scaling_addto, 3 scaling_subto*/,
                    int amount);                                                          /* trlwe.c:580-622 */

/* ---- deterministic test-input generation (own code; the reference's RNG is RDRAND-seeded
 *      and not reproducible, src/misc.c:34-49) ---- */
uint64_t orc_rng_next(orc_rng *r);                       /* splitmix64 */
double   orc_rng_normal(orc_rng *r, double sigma);       /* Box-Muller as misc.c:87-91 */
void orc_gen_binary_key(orc_rng *r, Torus *s, int n);    /* tlwe.c:70-82 with bound 2 */
void orc_tlwe_sample(orc_rng *r, Torus *out, Torus m, const Torus *s, int n, double sigma);            /* tlwe.c:106-115 */
void orc_trlwe_sample(orc_rng *r, Torus *out, const Torus *m, const Torus *s, int k, int N, double sigma); /* trlwe.c:296-316 (binary key) */
void orc_trgsw_monomial_sample(orc_rng *r, Torus *out, int64_t m, int e, const Torus *s,
                               int k, int N, int l, int Bg_bit, double sigma);         /* trgsw.c:152-168 */
void orc_gen_bootstrap_key(orc_rng *r, Torus *bk /*[n][(k+1)l][k+1][N]*/, const Torus *lwe_s, int n,
                           const Torus *rlwe_s, int k, int N, int l, int Bg_bit, double sigma); /* bootstrap.c:14-18 */
void orc_gen_tlwe_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in, int n_in, const Torus *s_out, int n_out,
                         int t, int base_bit, double sigma);                           /* tlwe.c:193-212 */

#ifdef __cplusplus
}
#endif
/* LUT-packing key switch (src/keyswitch.c:214-241,346-366) */
void orc_trlwe_lut_packing_keyswitch(Torus *out, const Torus *in, const Torus *ksk, int n, int N, int t, int base_bit, int torus_base);
void orc_gen_lut_packing_ks_key(orc_rng *r, Torus *ksk, const Torus *s_in, int n, const Torus *s_out, int N, int t, int base_bit, int torus_base, double sigma);
#endif
