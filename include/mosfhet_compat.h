/*
 * mosfhet_compat.h -- MOSFHET-compatible host API of the MI355X bootstrap engine (hot-path subset).
 *
 * Plain C.  The types below have the SAME field order and meaning as the torus-domain types of the
 * reference's include/mosfhet.h (lines cited per type), because reference callers poke the fields directly
 * (e.g. `lut->b->coeffs[i]`, `ct->b -= x`, test/tests.c:1453, src/bootstrap.c:409).  The functions keep the
 * reference's names, argument order and error behaviour (void; abort with a message on failure, as the
 * reference's assert / safe_malloc do, src/misc.c:104-128), so a program written against mosfhet.h for this
 * path re-links against libmosfhet_hip.so unchanged.  What differs, by design:
 *   - DFT-domain objects (DFT_Polynomial, TRLWE_DFT, TRGSW_DFT, the entries of Bootstrap_Key.s) keep the reference's struct shapes, but
 *     their `coeffs` point to DEVICE memory in the engine's own element order -- as in the reference, where DFT contents are private to the FFT
 *     back-end (src/polynomial.c:336-357), host code must not read them; TLWE_KS_Key additionally owns a device copy of its table;
 *   - every bootstrap / key switch runs on the GPU through include/mosfhet_hip.h; there is no CPU path;
 *   - new *_batch entry points take arrays of samples (the reference has no batching API; its callers loop,
 *     e.g. applications/multi-ciphertext-arith/src/lut.c:12-17);
 *   - randomness comes from a seedable generator (mosfhet_seed); the reference seeds from RDRAND and is not
 *     reproducible (src/misc.c:34-49).
 * Functions of mosfhet.h outside the bootstrap path and its callers are not provided (scope).
 */
#ifndef MOSFHET_COMPAT_H
#define MOSFHET_COMPAT_H
#include <stddef.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t Torus;            /* mosfhet.h:27 */
typedef Torus Integer;             /* mosfhet.h:48 */

typedef struct _TorusPolynomial { Torus *coeffs; int N; } *TorusPolynomial;          /* mosfhet.h:32-35 */
typedef TorusPolynomial IntPolynomial;                                               /* mosfhet.h:47 */
typedef struct _DFT_Polynomial { double *coeffs; int N; } *DFT_Polynomial;           /* mosfhet.h:37-40 */

typedef struct _TLWE { Torus *a, b; int n; } *TLWE;                                  /* mosfhet.h:51-54 */
typedef struct _TLWE_Key { Integer *s; int n; double sigma; } *TLWE_Key;             /* mosfhet.h:56-60 */
typedef struct _TLWE_KS_Key { TLWE ***s; int base_bit, t, n;                         /* mosfhet.h:62-65 */
                              void *device; } *TLWE_KS_Key;                          /* + engine handle (appended) */

typedef struct _TRLWE { TorusPolynomial *a, b; int k; } *TRLWE;                      /* mosfhet.h:73-76 */
typedef struct _TRLWE_Key { IntPolynomial *s; DFT_Polynomial *s_dft; int k; double sigma; } *TRLWE_Key; /* mosfhet.h:83-88 */

typedef struct _TRLWE_DFT { DFT_Polynomial *a, b; int k; } *TRLWE_DFT;               /* mosfhet.h:78-81; coefficients in device memory */
typedef struct _TRGSW { TRLWE *samples; int l, Bg_bit; } *TRGSW;                     /* mosfhet.h:106-109 */
typedef struct _TRGSW_DFT { TRLWE_DFT *samples; int l, Bg_bit; } *TRGSW_DFT;         /* mosfhet.h:111-114; one device block per sample (or array) */
typedef struct _TRGSW_Key { TRLWE_Key trlwe_key; int l, Bg_bit; } *TRGSW_Key;        /* mosfhet.h:116-119 */

/* mosfhet.h:129-133.  s[i] (unfolding 1) are views of the device-resident key, usable wherever the reference takes a TRGSW_DFT; su is NULL (an
 * unfolded key's samples live on the device as well).  `device` (appended) is the engine's key handle. */
typedef struct _Bootstrap_Key { TRGSW_DFT *s; TRGSW *su; int n, k, N, Bg_bit, l, unfolding; void *device; } *Bootstrap_Key;

/* ---- engine control (new) ---- */
void mosfhet_seed(uint64_t seed);              /* reseed the host generator used by all *_sample / *_key functions */
void mosfhet_set_device(int device);           /* GPU used by this process (default: env MOSFHET_HIP_DEVICE or 0) */
void mosfhet_set_devices(int n, const int *ids);   /* several GPUs (default: env MOSFHET_HIP_DEVICES="0,1,..."): ids[0] is the primary device (keys are made
                                                    * there, single-sample calls run there); EVERY *_batch entry point -- bootstraps, Galois bootstraps, LWE key
                                                    * switches, full-domain bootstraps (plain, KS21, CLOT21), multi-value and circuit bootstraps -- cuts its
                                                    * batch into n contiguous slices, one per GPU, each on its own host thread, stream and staging, with the
                                                    * keys replicated per GPU on first use, device to device.  SURVEY 8(e); no collective.  Before the first call. */
int mosfhet_device_count(void);                /* GPUs in use */
void mosfhet_replication_stats(unsigned long long bytes[4], double seconds[4], int keys[4]);   /* key replication so far, per route: 0 same device, 1 peer to
                                                    * peer (xGMI), 2 device to device without peer access, 3 host bounce buffer (mosfhet_hip_last_clone_route) */
void *mosfhet_engine_ctx(void);                /* the mosfhet_hip_ctx_t behind the compat layer */
void *mosfhet_bootstrap_key_device(Bootstrap_Key key);   /* the mosfhet_hip_bsk_t behind a Bootstrap_Key */

/* ---- torus scalars (src/misc.c:13-28) ---- */
double torus2double(Torus x);
Torus double2torus(double x);
uint64_t torus2int(Torus x, int log_scale);
Torus int2torus(uint64_t x, int log_scale);

/* ---- polynomials (src/polynomial.c:3-53) ---- */
TorusPolynomial polynomial_new_torus_polynomial(int N);
void free_polynomial(void *p);                                     /* torus- and DFT-domain polynomials alike, as in the reference */
void *safe_malloc(size_t size);                                    /* src/misc.c:104-113 */
void *safe_aligned_malloc(size_t size);                            /* src/misc.c:115-128 */
void generate_random_bytes(uint64_t amount, uint8_t *pointer);     /* src/misc.c:79-82 (here: the host layer's ChaCha20 stream) */
double generate_normal_random(double sigma);                       /* src/misc.c:87-91 */
void generate_torus_normal_random_array(Torus *out, double sigma, int N);   /* src/misc.c:93-97 */
void generate_rnd_seed(uint64_t *p);                               /* src/misc.c:34-49 */

/* ---- DFT-level functions (src/polynomial.c:336-426, src/trlwe.c:622-634, src/trgsw.c:345-357,385-423)  -> GPU.
 * The legacy signatures of SURVEY 8(b): every object named *_DFT lives in device memory; each call copies its torus-domain arguments in or out
 * and waits for its result, like the single-sample bootstrap entry points.  Batched work belongs on include/mosfhet_hip.h. ---- */
void init_fft(int N);                                                              /* polynomial.c:341-356: starts the engine; N in {1024, 2048, 4096} */
DFT_Polynomial polynomial_new_DFT_polynomial(int N);                               /* polynomial.c:12-19 */
DFT_Polynomial *polynomial_new_array_of_polynomials_DFT(int N, int size);          /* polynomial.c:21-27 */
void free_DFT_polynomial(DFT_Polynomial p);                                        /* polynomial.c:47-53 */
void free_array_of_polynomials(void *p, int size);                                 /* polynomial.c:27-33 */
void polynomial_torus_to_DFT(DFT_Polynomial out, TorusPolynomial in);              /* polynomial.c:368-375 */
void polynomial_DFT_to_torus(TorusPolynomial out, const DFT_Polynomial in);        /* polynomial.c:359-366 */
void polynomial_mul_DFT(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2);        /* polynomial.c:379-400 */
void polynomial_mul_addto_DFT(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2);  /* polynomial.c:404-426 */
void polynomial_copy_DFT_polynomial(DFT_Polynomial out, DFT_Polynomial in);
TRLWE_DFT trlwe_alloc_new_DFT_sample(int k, int N);                                /* trlwe.c:54-64 */
TRLWE_DFT *trlwe_alloc_new_DFT_sample_array(int count, int k, int N);              /* trlwe.c:66-72 */
void trlwe_to_DFT(TRLWE_DFT out, TRLWE in);                                        /* trlwe.c:622-627 */
void trlwe_from_DFT(TRLWE out, TRLWE_DFT in);                                      /* trlwe.c:629-634 */
TRGSW_DFT trgsw_alloc_new_DFT_sample(int l, int Bg_bit, int k, int N);             /* trgsw.c:61-72 */
TRGSW_DFT *trgsw_alloc_new_DFT_sample_array(int count, int l, int Bg_bit, int k, int N);   /* trgsw.c:74-80: one contiguous device block */
void trgsw_to_DFT(TRGSW_DFT out, TRGSW in);                                        /* trgsw.c:345-349 */
void trgsw_mul_trlwe_DFT(TRLWE_DFT out, TRLWE in1, TRGSW_DFT in2);                 /* trgsw.c:385-423: the external product, result in the DFT domain */

/* ---- TLWE (src/tlwe.c) ---- */
TLWE_Key tlwe_alloc_key(int n, double sigma);                     /* :60-67 */
TLWE_Key tlwe_new_binary_key(int n, double sigma);                /* :81-83 */
void free_tlwe_key(TLWE_Key key);                                 /* :102-105 */
TLWE tlwe_alloc_sample(int n);                                    /* :3-10 */
TLWE *tlwe_alloc_sample_array(int count, int n);                  /* :12-19 */
void free_tlwe(TLWE p);
void free_tlwe_array(TLWE *p, int count);
void tlwe_noiseless_trivial_sample(TLWE out, Torus m);            /* :32-35 */
void tlwe_sample(TLWE out, Torus m, TLWE_Key key);                /* :106-115 */
TLWE tlwe_new_sample(Torus m, TLWE_Key key);                      /* :122-133 */
Torus tlwe_phase(TLWE c, TLWE_Key key);                           /* :135-141 */
void tlwe_copy(TLWE out, TLWE in);                                /* :117-120 */
TLWE tlwe_new_noiseless_trivial_sample(Torus m, int n);           /* :26-30 */
void tlwe_add(TLWE out, TLWE in1, TLWE in2);                      /* :143-149  (host; the glue between bootstraps in callers) */
void tlwe_addto(TLWE out, TLWE in);                               /* :171-173 */
void tlwe_sub(TLWE out, TLWE in1, TLWE in2);                      /* :175-180 */
void tlwe_subto(TLWE out, TLWE in);
void tlwe_negate(TLWE out, TLWE in);                              /* :182-187 */
void tlwe_scale(TLWE out, TLWE in1, Torus in2);                   /* :151-156 */
void tlwe_scale_addto(TLWE out, TLWE in1, Torus in2);
void tlwe_scale_subto(TLWE out, TLWE in1, Torus in2);             /* :163-169 */
TLWE_KS_Key tlwe_new_KS_key(TLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);   /* :193-212 */
void free_tlwe_ks_key(TLWE_KS_Key key);                           /* :232-245 */
void tlwe_keyswitch(TLWE out, TLWE in, TLWE_KS_Key ks_key);       /* :289-303  -> GPU */

/* ---- TRLWE (src/trlwe.c) ---- */
TRLWE_Key trlwe_alloc_key(int N, int k, double sigma);            /* :104-116 */
TRLWE_Key trlwe_new_binary_key(int N, int k, double sigma);       /* :132-134 */
void free_trlwe_key(TRLWE_Key key);
TRLWE trlwe_alloc_new_sample(int k, int N);                       /* :3-13 */
TRLWE *trlwe_alloc_new_sample_array(int count, int k, int N);     /* :15-21 */
void free_trlwe(void *p);                                         /* :86-94 (TRLWE and TRLWE_DFT) */
void free_trlwe_array(void *p, int count);                        /* :96-102 */
TRLWE trlwe_new_sample(TorusPolynomial m, TRLWE_Key key);         /* :318-322 */
void trlwe_noiseless_trivial_sample(TRLWE out, TorusPolynomial m);/* :273-283 */
TRLWE trlwe_new_noiseless_trivial_sample(TorusPolynomial m, int k, int N);
void trlwe_sample(TRLWE out, TorusPolynomial m, TRLWE_Key key);   /* :296-316 */
void trlwe_phase(TorusPolynomial out, TRLWE in, TRLWE_Key key);   /* :324-331 (exact product here) */
void trlwe_torus_packing(TRLWE out, Torus *in, int size);         /* :662-667 */
void trlwe_add(TRLWE out, TRLWE in1, TRLWE in2);                  /* :394-411  (host) */
void trlwe_addto(TRLWE out, TRLWE in);                            /* :437-439 */
void trlwe_sub(TRLWE out, TRLWE in1, TRLWE in2);
void trlwe_subto(TRLWE out, TRLWE in);
void trlwe_negate(TRLWE out, TRLWE in);
void trlwe_copy(TRLWE out, TRLWE in);
void trlwe_mul_by_xai(TRLWE out, TRLWE in, int a);                /* :441-447 with polynomial.c:184-199; out != in */
void trlwe_extract_tlwe_key(TLWE_Key out, TRLWE_Key in);          /* :531-538 */
void trlwe_extract_tlwe(TLWE out, TRLWE in, int idx);             /* :540-552 (host) */
void trlwe_extract_tlwe_addto(TLWE out, TRLWE in, int idx);       /* :554-565 (host) */
void trlwe_extract_tlwe_subto(TLWE out, TRLWE in, int idx);       /* :567-578 (host) */
void trlwe_mv_extract_tlwe(TLWE *out, TRLWE in, int amount);      /* :580-589 (host; batches: mosfhet_hip_trlwe_mv_extract_batch) */
void trlwe_mv_extract_tlwe_scaling(TLWE out, TRLWE in, int scale);        /* :591-601 */
void trlwe_mv_extract_tlwe_scaling_addto(TLWE out, TRLWE in, int scale);  /* :603-611 */
void trlwe_mv_extract_tlwe_scaling_subto(TLWE out, TRLWE in, int scale);  /* :613-622 */

/* ---- TRGSW (src/trgsw.c) ---- */
TRGSW_Key trgsw_new_key(TRLWE_Key trlwe_key, int l, int Bg_bit);  /* :20-27 */
void free_trgsw_key(TRGSW_Key key);
TRGSW trgsw_alloc_new_sample(int l, int Bg_bit, int k, int N);    /* :48-59 */
void free_trgsw(void *p);                                         /* TRGSW and TRGSW_DFT */
void free_trgsw_array(void *p, int count);                        /* :137-143 */
TRGSW *trgsw_alloc_new_sample_array(int count, int l, int Bg_bit, int k, int N);   /* :82-88 */
void trgsw_monomial_sample(TRGSW out, int64_t m, int e, TRGSW_Key key);   /* :152-168 */

/* ---- bootstrap (src/bootstrap.c)  -> GPU ---- */
Bootstrap_Key new_bootstrap_key(TRGSW_Key out_key, TLWE_Key in_key, int unfolding);   /* :3-48 (unfolding 1, 2, 4, 8; n divisible by it) */
void free_bootstrap_key(Bootstrap_Key key);                                            /* :51-61 */
void blind_rotate(TRLWE tv, Torus *a, TRGSW_DFT *s, int size);                         /* :107-122 */
void functional_bootstrap_wo_extract(TRLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, int torus_base);  /* :192-198 */
void functional_bootstrap(TLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, int torus_base);              /* :200-206 */
void programmable_bootstrap(TLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, int precision, int kappa, int theta); /* :208-220 */

void full_domain_functional_bootstrap(TLWE out, TRLWE tv, TLWE in, Bootstrap_Key key, TLWE_KS_Key tlwe_ksk, int precision); /* :519-538 */
void multivalue_bootstrap_CLOT21(TLWE *out, TRLWE tv, TLWE in, Bootstrap_Key key, int torus_base, int n_luts);              /* :222-230 */
void multivalue_bootstrap_CLOT21_batch(TLWE *out /* [count * n_luts]: LUT j of input b at b * n_luts + j */, TRLWE tv, TLWE *in, int count, Bootstrap_Key key,
                                       int torus_base, int n_luts);                                                          /* new */
void trlwe_torus_packing_many_LUT(TRLWE out, Torus *in, int lut_size, int n_luts);   /* src/trlwe.c:677-687 (host) */

/* ---- bootstrap with Galois automorphisms (src/bootstrap_ga.c)  -> GPU ---- */
typedef struct _TRLWE_KS_Key { void **s; int base_bit, t, k;                          /* mosfhet.h:90-93 */
                               void *device; int entry, owner; } *TRLWE_KS_Key;        /* + engine handle (appended) */
/* mosfhet.h:135-140: s[i] views of the device key, ak[j] = key of generator 2j + 1 (N headers sharing one device key set) */
typedef struct _Bootstrap_GA_Key { TRGSW_DFT *s; TRGSW *su; TRLWE_KS_Key *ak; int n, k, N, Bg_bit, l, unfolding; void *device, *ak_device; } *Bootstrap_GA_Key;
Bootstrap_GA_Key new_bootstrap_key_ga(TRGSW_Key out_key, TLWE_Key in_key);                                   /* :5-24 */
void free_bootstrap_key_ga(Bootstrap_GA_Key key);                                                            /* :26-33 */
void functional_bootstrap_wo_extract_ga(TRLWE out, TRLWE tv, TLWE in, Bootstrap_GA_Key key, int torus_base); /* :62-68 */
void functional_bootstrap_ga(TLWE out, TRLWE tv, TLWE in, Bootstrap_GA_Key key, int torus_base);             /* :70-76 */
void blind_rotate_ga(TRLWE tv, Torus *a, TRGSW_DFT *s, TRLWE_KS_Key *ak, int size);                          /* :35-60; s and ak from one Bootstrap_GA_Key */
void trlwe_eval_automorphism(TRLWE out, TRLWE in, uint64_t gen, TRLWE_KS_Key ks_key);                        /* src/trlwe.c:775-781 */
uint16_t inverse_mod_2N(uint16_t x, uint16_t N);                                                             /* src/misc.c:142-159 (host) */
void functional_bootstrap_ga_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_GA_Key key, int torus_base);  /* new */
void polynomial_permute(TorusPolynomial out, TorusPolynomial in, uint64_t gen);                               /* src/polynomial.c:442-450 (host) */
void mosfhet_gen_bootstrap_key_ga_flat(Torus *out /*[n][(k+1)l][k+1][N]*/, TRGSW_Key out_key, TLWE_Key in_key);       /* BK_i = TRGSW(X^{s_i}) */
void mosfhet_gen_automorphism_keyset_flat(Torus *out /*[N][t][2][N]*/, TRLWE_Key key, int t, int base_bit);           /* src/keyswitch.c:500-511 */

/* ---- TRLWE key switches and circuit bootstrap (src/keyswitch.c, src/bootstrap.c:346-366)  -> GPU ----
 * Both key types are device resident: `s` is NULL (reference: host arrays of TRLWE_DFT / TRLWE, mosfhet.h:90-103) and
 * `device` holds the engine handle.  A TRLWE_KS_Key array returned by trlwe_new_priv_KS_key shares ONE device key set
 * (entry = index in the set), as the automorphism keys do. */
typedef struct _Generic_KS_Key { TRLWE ***s; int base_bit, t, n, include_b;            /* mosfhet.h:100-103 */
                                 void *device; } *Generic_KS_Key;                      /* + engine handle (appended) */
TRLWE_KS_Key trlwe_new_KS_key(TRLWE_Key out_key, TRLWE_Key in_key, int t, int base_bit);         /* keyswitch.c:12-37 */
void free_trlwe_ks_key(TRLWE_KS_Key key);                                                         /* keyswitch.c:195-203 */
void trlwe_keyswitch(TRLWE out, TRLWE in, TRLWE_KS_Key ks_key);                                   /* keyswitch.c:162-193; out == in allowed */
TRLWE_KS_Key *trlwe_new_priv_KS_key(TRLWE_Key out_key, TRLWE_Key in_key, int t, int base_bit);    /* keyswitch.c:39-50 (array of 2) */
void trlwe_priv_keyswitch_2(TRLWE out, TRLWE in, TRLWE_KS_Key *ks_key);                           /* keyswitch.c:52-63 */
Generic_KS_Key trlwe_new_packing1_KS_key(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);/* keyswitch.c:368-390 */
void free_trlwe_generic_ks_key(Generic_KS_Key key);                                               /* keyswitch.c:392-407 */
void trlwe_packing1_keyswitch(TRLWE out, TLWE in, Generic_KS_Key ks_key);                         /* keyswitch.c:458-475 */
void circuit_bootstrap_3(TRGSW out, TLWE in, Bootstrap_Key key, TRLWE_KS_Key *kska, Generic_KS_Key kskb);   /* bootstrap.c:346-366 */
void circuit_bootstrap_3_batch(TRGSW *out, TLWE *in, int count, Bootstrap_Key key, TRLWE_KS_Key *kska, Generic_KS_Key kskb); /* new */
void mosfhet_gen_trlwe_ks_key_flat(Torus *out /*[t][2][N]*/, const Torus *s_in /*[N]*/, TRLWE_Key out_key, int t, int base_bit);
void mosfhet_gen_priv_ks_key_flat(Torus *out /*[2][t][2][N]*/, TRLWE_Key out_key, TRLWE_Key in_key, int t, int base_bit);
void mosfhet_gen_packing1_ks_key_flat(Torus *out /*[n][t][2^bb-1][2][N]*/, TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);

/* ---- callers either side of the bootstrap  -> GPU compositions ---- */
Generic_KS_Key trlwe_new_priv_SK_KS_key_N2(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);       /* keyswitch.c:611-637 */
void trlwe_priv_keyswitch(TRLWE out, TLWE in, Generic_KS_Key ks_key);                                      /* keyswitch.c:639-656 */
void circuit_bootstrap(TRGSW out, TLWE in, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb);   /* bootstrap.c:309-322 */
void circuit_bootstrap_2(TRGSW out, TLWE in, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb); /* bootstrap.c:324-344 */
void circuit_bootstrap_2_batch(TRGSW *out, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key kska, Generic_KS_Key kskb); /* new */
void public_mux(TRLWE out, TorusPolynomial p0, TorusPolynomial p1, TRLWE_DFT *selector, int l, int Bg_bit);   /* bootstrap.c:369-389 */
void full_domain_functional_bootstrap_KS21(TLWE out, TorusPolynomial tv, TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base);   /* :391-432 */
void full_domain_functional_bootstrap_KS21_2(TLWE out, TorusPolynomial tv, TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base); /* :434-463 */
void full_domain_functional_bootstrap_KS21_batch(TLWE *out, TorusPolynomial tv, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, int torus_base); /* new */
TRLWE_KS_Key trlwe_new_RL_key(TRLWE_Key key, int t, int base_bit);                                         /* keyswitch.c:3-10 */
void trlwe_tensor_prod_FFT(TRLWE out, TRLWE in1, TRLWE in2, int precision, TRLWE_KS_Key rl_key);           /* trlwe.c:727-771 */
void tlwe_mul(TLWE out, TLWE in1, TLWE in2, int precision, Generic_KS_Key ksk, TRLWE_KS_Key rlk);          /* tlwe.c:322-332 */
void full_domain_functional_bootstrap_CLOT21(TLWE out, TRLWE tv[2], TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision);   /* bootstrap.c:465-491 */
void full_domain_functional_bootstrap_CLOT21_2(TLWE out, Torus *tv, TLWE in, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk, int precision);   /* :493-517 */
void full_domain_functional_bootstrap_CLOT21_2_batch(TLWE *out, Torus *tv, TLWE *in, int count, Bootstrap_Key key, Generic_KS_Key ksk, TRLWE_KS_Key rlk,
                                                     int precision);                                       /* new */
void multivalue_bootstrap_phase1(TRLWE *out, TLWE in, Bootstrap_Key key, int torus_base);                  /* bootstrap.c:232-243 */
void multivalue_bootstrap_phase2(TLWE out, int *in, TRLWE *rotated_tv, int torus_base, int log_torus_base);/* bootstrap.c:245-265 */
void functional_bootstrap_trgsw_phase1(TRGSW_DFT out, TLWE in, Bootstrap_Key key, int torus_base);         /* bootstrap.c:284-295 */
void functional_bootstrap_trgsw_phase2(TLWE out, TRGSW_DFT in, TRLWE tv);                                  /* bootstrap.c:297-306 */
void mosfhet_gen_priv_sk_ks_key_flat(Torus *out /*[n+1][t][2^bb-1][2][N]*/, TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);
void mosfhet_gen_bootstrap_key_unfolded_flat(Torus *out /*[n 2^u/u][2l][2][N]*/, TRGSW_Key out_key, TLWE_Key in_key, int unfolding);

/* ---- on-disk formats.  Torus-domain objects are written byte for byte as the reference writes them (table keys: its uncompressed
 * row form); DFT-domain keys keep the reference's integer header, then a layout tag and the engine's own image (DFT contents are backend-defined in
 * the reference as well, src/polynomial.c:336-357).  Short reads abort. ---- */
void tlwe_save_sample(FILE *fd, TLWE c);                         /* tlwe.c:43-46 */
void tlwe_load_sample(FILE *fd, TLWE c);                         /* tlwe.c:55-58 */
TLWE tlwe_load_new_sample(FILE *fd, int n);                      /* tlwe.c:48-53 */
void tlwe_save_key(FILE *fd, TLWE_Key key);                      /* tlwe.c:85-89 */
TLWE_Key tlwe_load_new_key(FILE *fd);                            /* tlwe.c:91-99 */
void trlwe_save_sample(FILE *fd, TRLWE c);                       /* trlwe.c:24-29 */
void trlwe_load_sample(FILE *fd, TRLWE c);                       /* trlwe.c:31-37 */
TRLWE trlwe_load_new_sample(FILE *fd, int k, int N);             /* trlwe.c:39-43 */
void trlwe_save_key(FILE *fd, TRLWE_Key key);                    /* trlwe.c:230-237 */
TRLWE_Key trlwe_load_new_key(FILE *fd);                          /* trlwe.c:239-251 */
void trgsw_save_key(FILE *fd, TRGSW_Key key);                    /* trgsw.c:38-42 */
TRGSW_Key trgsw_load_new_key(FILE *fd);                          /* trgsw.c:29-36 */
void trgsw_save_sample(FILE *fd, TRGSW c);                       /* trgsw.c:60-64 */
void trgsw_load_sample(FILE *fd, TRGSW c);
TRGSW trgsw_load_new_sample(FILE *fd, int l, int Bg_bit, int k, int N);   /* trgsw.c:66-72 */
void tlwe_save_KS_key(FILE *fd, TLWE_KS_Key key);                /* tlwe.c:275-287 */
TLWE_KS_Key tlwe_load_new_KS_key(FILE *fd);                      /* tlwe.c:247-273 */
void trlwe_save_generic_ks_key(FILE *fd, Generic_KS_Key key);    /* keyswitch.c:409-424 */
Generic_KS_Key trlwe_load_new_generic_ks_key(FILE *fd);          /* keyswitch.c:426-455 */
void trlwe_save_KS_key(FILE *fd, TRLWE_KS_Key key);              /* keyswitch.c:122-134 */
TRLWE_KS_Key trlwe_load_new_KS_key(FILE *fd);                    /* keyswitch.c:136-160 */
void save_bootstrap_key(FILE *fd, Bootstrap_Key key);            /* bootstrap.c:63-81 */
Bootstrap_Key load_new_bootstrap_key(FILE *fd);                  /* bootstrap.c:83-104 */

/* ---- batch extensions (new): arrays of `count` samples, one shared test vector ---- */
void functional_bootstrap_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, int torus_base);
void programmable_bootstrap_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key,
                                  int precision, int kappa, int theta);
void tlwe_keyswitch_batch(TLWE *out, TLWE *in, int count, TLWE_KS_Key ks_key);
void full_domain_functional_bootstrap_batch(TLWE *out, TRLWE tv, TLWE *in, int count, Bootstrap_Key key, TLWE_KS_Key tlwe_ksk,
                                            int precision);

/* ---- flat helpers used by the Python binding and bench.py (new) ----
 * Generate a whole bootstrap / key-switch key in the flat torus-domain layouts of mosfhet_hip.h. */
void mosfhet_gen_bootstrap_key_flat(Torus *out /*[n][(k+1)l][k+1][N]*/, TRGSW_Key out_key, TLWE_Key in_key);
void mosfhet_gen_tlwe_ks_key_flat(Torus *out /*[n_in][t][2^bb-1][n_out+1]*/, TLWE_Key out_key, TLWE_Key in_key,
                                  int t, int base_bit);
void mosfhet_tlwe_sample_flat(Torus *out /*[n+1]*/, Torus m, TLWE_Key key);

/* ---- single-object helpers either side of the path (csrc/host/mosfhet_compat_legacy.c) ----
 * torus domain: host structs, exact arithmetic mod 2^64 (file:line = the reference function each one stands for) */
void polynomial_zero_torus_polynomial(TorusPolynomial p);                                                   /* polynomial.c:139-141 */
void polynomial_copy_torus_polynomial(TorusPolynomial out, TorusPolynomial in);                             /* :135-137 */
void polynomial_negate_torus_polynomial(TorusPolynomial out, TorusPolynomial in);                           /* :144-148 */
void polynomial_add_torus_polynomials(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2);       /* :95-99 */
void polynomial_addto_torus_polynomial(TorusPolynomial out, TorusPolynomial in);                            /* :156-160 */
void polynomial_sub_torus_polynomials(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2);       /* :163-177 */
void polynomial_subto_torus_polynomial(TorusPolynomial out, TorusPolynomial in);                            /* :180-182 */
TorusPolynomial *polynomial_new_array_of_torus_polynomials(int N, int size);                                /* :20-26; free_array_of_polynomials */
void polynomial_torus_scale(TorusPolynomial out, TorusPolynomial in, int log_scale);                        /* :319-325 */
void polynomial_torus_scale2(TorusPolynomial out, TorusPolynomial in, uint64_t scale);                      /* :327-332 */
void polynomial_decompose(TorusPolynomial *out, TorusPolynomial in, int Bg_bit, int l);                     /* :55-72 (public_mux's digits) */
void polynomial_decompose_i(TorusPolynomial out, TorusPolynomial in, int Bg_bit, int l, int i);             /* :74-89 (the external product's digits) */
void trlwe_decompose(TorusPolynomial *out, TRLWE in, int Bg_bit, int l);                                    /* trlwe.c:636-660 */
void torus_polynomial_mul_by_xai(TorusPolynomial out, TorusPolynomial in, int a);                           /* polynomial.c:184-199; out != in */
void torus_polynomial_mul_by_xai_addto(TorusPolynomial out, TorusPolynomial in, int a);                     /* :202-217 */
void torus_polynomial_mul_by_xai_minus_1(TorusPolynomial out, TorusPolynomial in, int a);                   /* :220-235 */
void polynomial_naive_mul_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2);             /* :290-303: exact, O(N^2) */
void polynomial_naive_mul_addto_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2);       /* :266-277 */
void trlwe_mul_by_xai_addto(TRLWE out, TRLWE in, int a);                                                    /* trlwe.c:464-469 */
void trlwe_mul_by_xai_minus_1(TRLWE out, TRLWE in, int a);                                                  /* trlwe.c:471-476 */
void trlwe_scale(TRLWE out, TRLWE in, uint64_t scale);                                                      /* trlwe.c:269-274 */
void trlwe_LUT_packing(TRLWE out, uint64_t *in, uint64_t in_prec, uint64_t out_prec);                       /* trlwe.c:669-675 */
void trgsw_add(TRGSW out, TRGSW in1, TRGSW in2);                                                            /* trgsw.c:312-317 */
void trgsw_addto(TRGSW out, TRGSW in);                                                                      /* :319-321 */
void trgsw_sub(TRGSW out, TRGSW in1, TRGSW in2);                                                            /* :275-280 */
void trgsw_copy(TRGSW out, TRGSW in);                                                                       /* :296-301 */
void trgsw_mul_by_xai(TRGSW out, TRGSW in, int a);                                                          /* :324-329 */
void trgsw_mul_by_xai_addto(TRGSW out, TRGSW in, int a);                                                    /* :331-336 */
void trgsw_mul_by_xai_minus_1(TRGSW out, TRGSW in, int a);                                                  /* :338-343 */
void trgsw_noiseless_trivial_sample(TRGSW out, Torus m, int l, int Bg_bit, int k, int N);                   /* :128-143 */
TRGSW trgsw_new_noiseless_trivial_sample(Torus m, int l, int Bg_bit, int k, int N);                         /* :145-149 */
TRGSW trgsw_new_monomial_sample(int64_t m, int e, TRGSW_Key key);                                           /* :178-184 */
TRGSW trgsw_new_sample(Torus m, TRGSW_Key key);                                                             /* :186-188 */
TRGSW trgsw_new_exp_sample(int e, TRGSW_Key key);                                                           /* :271-273 */
/* DFT domain: device-resident objects (see trlwe_alloc_new_DFT_sample above); every call is a kernel launch */
void polynomial_add_DFT_polynomials(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2);            /* polynomial.c:102-106 */
void polynomial_sub_DFT_polynomials(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2);            /* :129-133 */
void polynomial_scale_and_add_DFT_polynomials(DFT_Polynomial out, DFT_Polynomial in1, DFT_Polynomial in2, uint64_t scale);   /* :108-121 */
void polynomial_mul_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2);                   /* :281-292: product through the transform */
void polynomial_mul_addto_torus(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2);             /* :294-303 */
void trlwe_DFT_add(TRLWE_DFT out, TRLWE_DFT in1, TRLWE_DFT in2);                                            /* trlwe.c:443-448 */
void trlwe_DFT_addto(TRLWE_DFT out, TRLWE_DFT in);                                                          /* :450-452 */
void trlwe_DFT_sub(TRLWE_DFT out, TRLWE_DFT in1, TRLWE_DFT in2);                                            /* :454-459 */
void trlwe_DFT_copy(TRLWE_DFT out, TRLWE_DFT in);                                                           /* :478-483 */
void trlwe_DFT_mul_by_polynomial(TRLWE_DFT out, TRLWE_DFT in, DFT_Polynomial in2);                          /* :491-496 */
void trlwe_DFT_mul_addto_by_polynomial(TRLWE_DFT out, TRLWE_DFT in, DFT_Polynomial in2);                    /* :498-505 */
void trlwe_noiseless_trivial_DFT_sample(TRLWE_DFT out, DFT_Polynomial m);                                   /* :282-289 */
TRLWE_DFT trlwe_new_noiseless_trivial_DFT_sample(DFT_Polynomial m, int k, int N);                           /* :291-295 */
void trlwe_DFT_phase(TorusPolynomial out, TRLWE_DFT in, TRLWE_Key key);                                     /* :372-382 */
void trgsw_DFT_add(TRGSW_DFT out, TRGSW_DFT in1, TRGSW_DFT in2);                                            /* trgsw.c:289-294 */
void trgsw_DFT_sub(TRGSW_DFT out, TRGSW_DFT in1, TRGSW_DFT in2);                                            /* :282-287 */
void trgsw_DFT_copy(TRGSW_DFT out, TRGSW_DFT in);                                                           /* :303-308 */
void trgsw_DFT_mul_addto_by_polynomial(TRGSW_DFT out, TRGSW_DFT in1, DFT_Polynomial in2);                   /* :449-454 */
void trgsw_from_DFT(TRGSW out, TRGSW_DFT in);                                                               /* :351-357 */
void trgsw_mul_DFT(TRGSW_DFT out, TRGSW in1, TRGSW_DFT in2);                                                /* :425-431; out != in2 */
void trgsw_mul_DFT2(TRGSW_DFT out, TRGSW_DFT in1, TRGSW_DFT in2);                                           /* :433-447 */
void trgsw_mul_trlwe_DFT_prefetch(TRLWE_DFT out, TRLWE in1, TRGSW_DFT in2);                                 /* = trgsw_mul_trlwe_DFT */
void trgsw_monomial_DFT_sample(TRGSW_DFT out, int64_t m, int e, TRGSW_Key key);                             /* :170-175 */
void trlwe_save_DFT_sample(FILE *fd, TRLWE_DFT c);                                                          /* trlwe.c:66-71: (k+1) N doubles, this engine's slot order */
void trlwe_load_DFT_sample(FILE *fd, TRLWE_DFT c);                                                          /* :79-85 */
TRLWE_DFT trlwe_load_new_DFT_sample(FILE *fd, int k, int N);                                                /* :73-77 */
void trgsw_save_DFT_sample(FILE *fd, TRGSW_DFT c);                                                          /* trgsw.c:80-84 */
void trgsw_load_DFT_sample(FILE *fd, TRGSW_DFT out);                                                        /* :94-98 */
TRGSW_DFT trgsw_load_new_DFT_sample(FILE *fd, int l, int Bg_bit, int k, int N);                             /* :86-92 */
/* LUT-packing key switch (src/keyswitch.c:214-366; the encrypted lookup tables of applications/multi-ciphertext-arith): device-resident table key */
typedef struct _LUT_Packing_KS_Key { TRLWE ****s; int base_bit, t, torus_base, n;          /* mosfhet.h:95-98 */
                                     void *device; } *LUT_Packing_KS_Key;                  /* + engine handle (appended); s is NULL */
LUT_Packing_KS_Key trlwe_new_packing_KS_key(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit, int torus_base);   /* keyswitch.c:214-241 */
void trlwe_packing_keyswitch(TRLWE out, TLWE *in, LUT_Packing_KS_Key ks_key);                                /* :346-366; in: torus_base samples */
void trlwe_save_packing_KS_key(FILE *fd, LUT_Packing_KS_Key key);                                            /* :243-262 (uncompressed rows) */
LUT_Packing_KS_Key trlwe_load_new_packing_KS_key(FILE *fd);                                                  /* :264-296 */
void free_trlwe_packing_ks_key(LUT_Packing_KS_Key key);                                                      /* :298-316 */
/* ---- digit-parallel radix-integer callers on host structs (new; csrc/host/mosfhet_compat_vec.c over mosfhet_hip_vec_*, csrc/capi_vec.inc) ----
 * M independent integers per call, each an array of TLWE digits under the extracted TRLWE key: x[m] is the `digits` member of the reference application's
 * ufhe_integer m (applications/multi-ciphertext-arith/include/ufhe.h:18-22), the keys are the members of its ufhe_public_keyset (:12-16), torus_base its radix.
 * Every carry step / tree level is ONE key switch + ONE packing switch + ONE bootstrap launch over all M; results decrypt to what the application's own loops give
 * (src/integer.c:62-264, src/lut.c:6-64, src/ml.c:4-20; tests/c/vec_callers.c on tests/golden/ufhe_vectors.npz).  M <= 65535.  Synchronous, abort on error like the rest. */
typedef struct _mosfhet_vec *mosfhet_vec;
mosfhet_vec mosfhet_vec_new(Bootstrap_Key bootstrap_key, TLWE_KS_Key ks_key, LUT_Packing_KS_Key packing_key, int torus_base);
void mosfhet_vec_free(mosfhet_vec v);
void mosfhet_vec_add_integers(mosfhet_vec v, TLWE **c, TLWE **a, TLWE **b, int M, int d);                       /* c[m] = a[m] + b[m]      (ufhe_add_integer, integer.c:109-113) */
void mosfhet_vec_sub_integers(mosfhet_vec v, TLWE **c, TLWE **a, TLWE **b, int M, int d);                       /* c[m] = a[m] - b[m]      (ufhe_sub_integer, :136-158) */
void mosfhet_vec_sl_add_integers(mosfhet_vec v, TLWE **c, int dc, TLWE **a, int da, int g, TLWE **b, int db, int h, bool is_signed, int M);   /* a B^g + b B^h (:79-107) */
void mosfhet_vec_mul_integers(mosfhet_vec v, TLWE **c, int dc, TLWE **a, int da, TLWE **b, int db, bool is_signed, int M);                    /* a b (:166-203) */
void mosfhet_vec_cmp_integers(mosfhet_vec v, TLWE *c, TLWE **a, TLWE **b, int M, int d, bool a_signed, bool b_signed);   /* c[m]: one digit, 0 / 1 / 2 for < / = / > (:217-264) */
void mosfhet_vec_relu_integers(mosfhet_vec v, TLWE **out, TLWE **in, int M, int d);                             /* max(in, 0)              (ufhe_relu_integer, ml.c:4-20) */
void mosfhet_vec_mux_integer_arrays(mosfhet_vec v, TLWE **out, TLWE **selector, int d_sel, int size, TLWE ***vec, int M, int d);   /* out[m] = vec[selector[m]][m] (lut.c:49-64); vec[e][m]: digits */
void mosfhet_vec_lut_integers(mosfhet_vec v, TLWE **out, int d_out, TLWE **selector, int d_sel, uint64_t *lut, int size, int M);   /* out[m] = lut[selector[m]], cleartext table (lut.c:23-47) */

/* ---- beyond the path: what the reference's own test-suite needs to link (csrc/host/mosfhet_compat_extra.c; compositions of the calls above) ---- */
typedef struct _TRGSW_REG { TRGSW_DFT positive, negative; } *TRGSW_REG;                                      /* mosfhet.h:123-125 */
TRGSW_REG trgsw_reg_alloc(int l, int Bg_bit, int k, int N);                                                  /* register.c:18-24 */
TRGSW_REG *trgsw_reg_alloc_array(int count, int l, int Bg_bit, int k, int N);
void trgsw_reg_sample(TRGSW_REG out, Torus m, TRGSW_Key key);
void trgsw_reg_copy(TRGSW_REG out, TRGSW_REG in);
void trgsw_reg_add(TRGSW_REG out, TRGSW_REG in1, TRGSW_REG in2);
void trgsw_reg_negate(TRGSW_REG reg);
void trgsw_reg_sub(TRGSW_REG out, TRGSW_REG in1, TRGSW_REG in2);
void trgsw_reg_subto(TRGSW_REG out, TRGSW_REG in);
void trgsw_reg_addto(TRGSW_REG out, TRGSW_REG in1);
void free_trgsw_reg(TRGSW_REG p);
void free_trgsw_reg_array(TRGSW_REG *p, int count);
uint64_t _debug_trgsw_decrypt_exp_sample(TRGSW c, TRGSW_Key key);                                            /* trgsw.c:190-221 */
uint64_t _debug_trgsw_decrypt_exp_DFT_sample(TRGSW_DFT c, TRGSW_Key key);                                    /* trgsw.c:243-268 */
void trgsw_naive_mul_trlwe(TRLWE out, TRLWE in1, TRGSW in2);                                                 /* trgsw.c:456-473: exact */
void trgsw_naive_mul(TRGSW out, TRGSW in1, TRGSW in2);                                                       /* trgsw.c:475-480 */
TLWE_Key tlwe_new_bounded_key(int n, uint64_t bound, double sigma);                                          /* tlwe.c:70-78 */
TRLWE_Key trlwe_new_bounded_key(int N, int k, uint64_t bound, double sigma);                                 /* trlwe.c:119-130 */
TRLWE trlwe_new_compressed_sample(TorusPolynomial m, TRLWE_Key key);                                         /* here: an ordinary sample */
void trlwe_compressed_subto(TRLWE out, TRLWE in);
TRLWE_KS_Key trlwe_new_full_packing_KS_key(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);         /* keyswitch.c:98-106 */
void trlwe_full_packing_keyswitch(TRLWE out, TLWE *in, uint64_t size, TRLWE_KS_Key ks_key);                  /* keyswitch.c:195-227 */
TRLWE_KS_Key *trlwe_new_packing1_KS_key_CDKS21(TRLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);     /* keyswitch.c:476-497 */
void trlwe_packing1_keyswitch_CDKS21(TRLWE out, TLWE in, TRLWE_KS_Key *ks_key);                              /* keyswitch.c:526-546 */

typedef int16_t Binary;                                                                                      /* mosfhet.h:29 */
typedef struct _BinaryPolynomial { Binary *coeffs; int N; } *BinaryPolynomial;                               /* mosfhet.h:42-45 */
BinaryPolynomial polynomial_new_binary_polynomial(int N);
void polynomial_naive_mul_binary(BinaryPolynomial out, BinaryPolynomial in1, BinaryPolynomial in2);
void polynomial_naive_mul_addto_torus_binary(TorusPolynomial out, TorusPolynomial in1, BinaryPolynomial in2);
TRLWE_Key trlwe_new_ternary_key(int N, int k, int h, double sigma);                                          /* trlwe.c:158-165 */
TRLWE_Key trlwe_new_sparse_ternary_key(int N, int k, int h, double sigma);
TRLWE_Key trlwe_new_sparse_binary_key(int N, int k, int h, double sigma);
TRLWE_Key trlwe_new_gaussian_key(int N, int k, double key_sigma, double noise_sigma);
TRLWE_Key trlwe_new_sparse_gaussian_key(int N, int k, int h, double key_sigma, double noise_sigma);
TRLWE_Key trlwe_new_sparse_generic_key(int N, int k, int h, uint64_t key_bound, double noise_sigma);
TRLWE trlwe_load_new_compressed_sample(FILE *fd, int k, int N);
void trlwe_load_compressed_sample(FILE *fd, TRLWE c);
void trlwe_save_compressed_sample(FILE *fd, TRLWE c);
void trlwe_compressed_DFT_sample(TRLWE_DFT out, TorusPolynomial m, TRLWE_Key key);
TRLWE_DFT trlwe_new_compressed_DFT_sample(TorusPolynomial m, TRLWE_Key key);
void trlwe_compressed_DFT_mul_addto(TRLWE_DFT out, DFT_Polynomial in1, TRLWE_DFT in2);
void print_trlwe_msg(TRLWE in, uint64_t prec, TRLWE_Key key);                                                /* trlwe.c:333-342 */
uint64_t _debug_trlwe_decrypt_exp_sample(TRLWE c, uint64_t prec, TRLWE_Key key);                             /* trlwe.c:344-370 */
TRLWE_KS_Key trlwe_new_RLWE_priv_KS_key(TRLWE_Key out_key, TRLWE_Key in_key, TorusPolynomial v, int t, int base_bit);   /* keyswitch.c:574-608 */
void trlwe_RLWE_priv_keyswitch(TRLWE out, TRLWE in, TRLWE_KS_Key ks_key);                                    /* keyswitch.c:64-96 */
TRLWE_KS_Key *trlwe_new_gadget_to_RGSW_KS(TRLWE_Key key, int t, int base_bit);                               /* keyswitch.c:548-557 */
void trgsw_from_gadget(TRGSW_DFT out, TRLWE *gadget, TRLWE_KS_Key *ksk);                                     /* keyswitch.c:559-571 */

typedef struct _TLWE_KS_Key_m { TLWE **s; int base_bit, t, n;                                                /* mosfhet.h:67-70 */
                                void *device; int n_out; } *TLWE_KS_Key_m;                                    /* + device rows (appended); s is NULL */
TLWE_KS_Key_m tlwe_new_KS_key_no_precomp(TLWE_Key out_key, TLWE_Key in_key, int t, int base_bit);            /* tlwe.c:214-230 */
void tlwe_keyswitch_no_precomp(TLWE out, TLWE in, TLWE_KS_Key_m ks_key);                                     /* tlwe.c:305-320 */
void polynomial_full_mul_with_scale(TorusPolynomial out, TorusPolynomial in1, TorusPolynomial in2, int bit_size, int bit_scale);   /* polynomial.c:428-437: exact */
void trlwe_tensor_prod(TRLWE out, TRLWE in1, TRLWE in2, int precision, TRLWE_KS_Key rl_key);                 /* trlwe.c:692-713 */

/* unfolded blind rotation on caller-held key material, and the automorphism key sets */
void blind_rotate_unfolded(TRLWE tv, Torus *a, TRGSW *s, int size, int unfolding);                          /* bootstrap.c:124-149; s in new_bootstrap_key's su layout */
void multivalue_bootstrap_UBR_phase1(TRGSW_DFT *out, TLWE in, Bootstrap_Key key);                           /* bootstrap.c:151-175; out: n / unfolding samples */
void multivalue_bootstrap_UBR_phase2(TLWE out, TRLWE tv, TLWE in, TRGSW_DFT *sa, Bootstrap_Key key, int torus_base);   /* bootstrap.c:177-190 */
TRLWE_KS_Key *trlwe_new_automorphism_KS_keyset(TRLWE_Key key, bool skip_even, int t, int base_bit);         /* keyswitch.c:500-511; N (or 2N) headers over one device key set */
TRLWE_KS_Key *trlwe_new_automorphism_KS_keyset_2(TRLWE_Key key, uint64_t *gens, uint64_t size, int t, int base_bit);   /* keyswitch.c:513-524 */

#ifdef __cplusplus
}
#endif
#endif
