/*
 * mosfhet_hip.h -- C ABI of the MI355X (gfx950) TFHE bootstrap engine.
 *
 * This is the thin device layer that a MOSFHET build links instead of its CPU hot path
 * (src/bootstrap.c, src/trgsw.c, src/polynomial.c with the src/fft back-ends, src/tlwe.c key switch).  Plain C: opaque
 * handles, raw pointers and sizes, no C++ / torch types.  Each entry point names the reference
 * function(s) of antoniocgj/MOSFHET it replaces (file:line in that repository); INTEGRATION.md shows
 * the reference-side binding.  The MOSFHET-compatible struct API (TLWE, TRLWE, Bootstrap_Key,
 * functional_bootstrap(), ...) built on top of this layer is declared in mosfhet_compat.h.
 *
 * Conventions
 *   Torus            uint64_t, arithmetic mod 2^64                         (include/mosfhet.h:27)
 *   TLWE(n)          Torus[n+1]:  a[0..n-1], b                             (mosfhet.h:51-54, flattened)
 *   TRLWE(k,N)       Torus[k+1][N]: a[0..k-1], b                           (mosfhet.h:73-76, flattened)
 *   TRGSW(k,N,l)     Torus[(k+1)l][k+1][N], row p*l+j                      (mosfhet.h:106-109, src/trgsw.c:152-168)
 *   batches          arrays of the above, densely packed, batch index outermost
 *   d_* parameters   DEVICE pointers (hipMalloc / torch CUDA storage); h_* parameters host pointers
 *   stream           a hipStream_t passed as void*; NULL = HIP's default (null) stream, as in every HIP API
 *                    (torch's default stream is that NULL handle).  Calls are asynchronous on the stream
 *                    and never synchronise the device unless documented
 *   return value     0 on success, a negative MOSFHET_HIP_E* code otherwise; mosfhet_hip_last_error()
 *                    returns a thread-local message.  The legacy void API of mosfhet_compat.h aborts on
 *                    error, like the reference's assert/exit behaviour (src/misc.c:104-128).
 *   supported        tuned kernels: k = 1; N = 1024, 2048, 4096 (all ring degrees of the reference's parameter sets, test/tests.c:37-62);
 *                    l <= 6 with l*Bg_bit < 64 (compile-time specialisations for 2x8, 4x9, 1x23); any n.  Every entry point.
 *                    general path (csrc/general_kernels.h; the reference is generic in k and N, src/trgsw.c:385-423): k <= 3 and any power-of-two
 *                    N in 256 .. 16384 -- a correctness path, one workgroup per ciphertext, bit-identical to the same oracle.  A bootstrap key with such
 *                    parameters (mosfhet_hip_bsk_create[_from_device]) serves programmable / functional bootstraps (+ wo_extract), blind_rotate, the
 *                    full-domain bootstrap, key switch + bootstrap, external products, CMUX and multivalue_bootstrap_CLOT21; the other entry points
 *                    return MOSFHET_HIP_EINVAL for it.
 *                    mosfhet_hip_torus_to_dft_batch / _dft_to_torus_batch take any such N (natural slot order outside 1024 / 2048 / 4096).
 *   threading        re-entrant, like the reference (thread-local scratch there, src/polynomial.c:269-352): a context and its key
 *                    handles are read-only after creation and may be shared by any number of host threads; device temporaries of
 *                    the compositions and of the table key switches belong to the CALLING THREAD (grown on demand, released at
 *                    thread exit).  A thread that spreads compositions over several streams must order them itself (its
 *                    temporaries are shared between its own launches); create / destroy of a handle must not race with its use.
 * DFT-domain data (bootstrap key) is device-resident in the engine's own slot order and never leaves it,
 * as in the reference where the element order is private to the FFT back-end (src/polynomial.c:336-357).
 */
#ifndef MOSFHET_HIP_H
#define MOSFHET_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MOSFHET_HIP_OK 0
#define MOSFHET_HIP_EINVAL (-1)      /* bad argument / unsupported parameter set */
#define MOSFHET_HIP_EHIP (-2)        /* a HIP runtime call failed (message has the hipError string) */
#define MOSFHET_HIP_ENODEV (-3)      /* no usable gfx950 device */

typedef struct mosfhet_hip_ctx *mosfhet_hip_ctx_t;     /* one per (process, device) */
typedef struct mosfhet_hip_bsk *mosfhet_hip_bsk_t;     /* device-resident DFT bootstrap key */
typedef struct mosfhet_hip_ksk *mosfhet_hip_ksk_t;     /* device-resident LWE key-switch key */
typedef struct mosfhet_hip_gak *mosfhet_hip_gak_t;     /* device-resident automorphism (TRLWE key-switch) key set */

const char *mosfhet_hip_last_error(void);
const char *mosfhet_hip_version(void);

/* Context: selects `device` and creates the twiddle tables (replaces the lazy per-thread
 * FFT-processor set-up of init_fft, src/polynomial.c:336-357). */
int mosfhet_hip_ctx_create(mosfhet_hip_ctx_t *out, int device);
int mosfhet_hip_ctx_destroy(mosfhet_hip_ctx_t ctx);
int mosfhet_hip_ctx_sync(mosfhet_hip_ctx_t ctx, void *stream);          /* hipStreamSynchronize */
int mosfhet_hip_device_count(void);

/* The twiddle table the engine uses for ring degree N: (re, im) pairs, N/2 - 1 of them (host copy;
 * lets tests check it is bit-identical to the oracle's). */
int mosfhet_hip_twiddles(int N, double *h_out);

/* Bootstrap key.  Takes the key in the TORUS domain, h_bk = Torus[n][(k+1)l][k+1][N] with
 * BK_i = TRGSW(s_i) -- what new_bootstrap_key holds in `tmp` before trgsw_to_DFT (src/bootstrap.c:14-18)
 * -- uploads it and runs the engine's own forward transform over every polynomial
 * (replaces trgsw_to_DFT, src/trgsw.c:345-349).  Synchronous. */
int mosfhet_hip_bsk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *h_bk,
                           int n, int k, int N, int l, int Bg_bit);
/* Same from a device-resident torus-domain key (e.g. generated on the GPU or received over xGMI). */
int mosfhet_hip_bsk_create_from_device(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *d_bk,
                                       int n, int k, int N, int l, int Bg_bit, void *stream);
int mosfhet_hip_bsk_destroy(mosfhet_hip_bsk_t bsk);
size_t mosfhet_hip_bsk_bytes(mosfhet_hip_bsk_t bsk);                     /* device bytes of the DFT key */
/* Copy the DFT-domain key back in ORACLE slot order (natural recursion index, interleaved re/im),
 * double[n][(k+1)l][k+1][N]; test hook only. */
int mosfhet_hip_bsk_export_dft(mosfhet_hip_bsk_t bsk, double *h_out);

/* programmable_bootstrap over a batch (src/bootstrap.c:208-220): for every b < count
 *   d_out[b] = TLWE(kN) = SampleExtract_0( BlindRotate( tv_b * X^{-round(2N b'/2^64)}, a', BK ) )
 * with (a', b') = ((x << kappa) + 2^(63-log2(2N)+theta)) & ~(2^(64-log2(2N)+theta) - 1) and
 * torus_base = 2^(precision-1).  d_tv holds tv_count test vectors; tv_count == 1 shares one test
 * vector across the batch (the reference's callers pass one LUT per call), tv_count == count gives
 * each ciphertext its own. */
int mosfhet_hip_programmable_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk,
                                             uint64_t *d_out /*[count][kN+1]*/, const uint64_t *d_tv /*[tv_count][k+1][N]*/,
                                             int tv_count, const uint64_t *d_in /*[count][n+1]*/, int count,
                                             int precision, int kappa, int theta, void *stream);

/* functional_bootstrap over a batch (src/bootstrap.c:200-206). */
int mosfhet_hip_functional_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk,
                                           uint64_t *d_out /*[count][kN+1]*/, const uint64_t *d_tv, int tv_count,
                                           const uint64_t *d_in /*[count][n+1]*/, int count, int torus_base, void *stream);

/* functional_bootstrap_wo_extract over a batch (src/bootstrap.c:192-198): writes the rotated TRLWE. */
int mosfhet_hip_functional_bootstrap_wo_extract_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk,
                                                      uint64_t *d_out /*[count][k+1][N]*/, const uint64_t *d_tv, int tv_count,
                                                      const uint64_t *d_in, int count, int torus_base, void *stream);

/* blind_rotate over a batch (src/bootstrap.c:107-122): d_acc[b] is rotated IN PLACE by the n mask
 * words of d_in[b] (the b word of d_in is ignored). */
int mosfhet_hip_blind_rotate_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_acc /*[count][k+1][N]*/,
                                   const uint64_t *d_in /*[count][n+1]*/, int count, void *stream);

/* trgsw_mul_trlwe_DFT + trlwe_from_DFT over a batch (src/trgsw.c:385-423, src/trlwe.c:629-634):
 * d_out[b] = BK[key_index] (.) d_in[b], result back in the torus domain. */
int mosfhet_hip_external_product_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, int key_index,
                                       uint64_t *d_out /*[count][k+1][N]*/, const uint64_t *d_in /*[count][k+1][N]*/,
                                       int count, void *stream);

/* polynomial_torus_to_DFT / polynomial_DFT_to_torus / polynomial_mul_DFT / polynomial_mul_addto_DFT over
 * flat arrays of polynomials (src/polynomial.c:359-426).  DFT polynomials are N doubles (N/2 complex,
 * interleaved, engine slot order). */
int mosfhet_hip_torus_to_dft_batch(mosfhet_hip_ctx_t ctx, double *d_out, const uint64_t *d_in, int N, int count, void *stream);
int mosfhet_hip_dft_to_torus_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const double *d_in, int N, int count, void *stream);
int mosfhet_hip_dft_mul_batch(mosfhet_hip_ctx_t ctx, double *d_out, const double *d_a, const double *d_b,
                              int N, int count, int addto, void *stream);

/* LWE key switch.  Key layout Torus[n_in][t][2^base_bit - 1][n_out + 1], entry [i][j][v-1] =
 * TLWE_out(s_in[i] * v * 2^(64-(j+1)base_bit)) -- tlwe_new_KS_key's table (src/tlwe.c:193-212) flattened. */
int mosfhet_hip_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_ksk,
                           int n_in, int n_out, int t, int base_bit);
int mosfhet_hip_ksk_destroy(mosfhet_hip_ksk_t ksk);
/* tlwe_keyswitch over a batch (src/tlwe.c:289-303), bit-exact. */
int mosfhet_hip_tlwe_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, uint64_t *d_out /*[count][n_out+1]*/,
                                     const uint64_t *d_in /*[count][n_in+1]*/, int count, void *stream);

/* trlwe_extract_tlwe at coefficient idx over a batch (src/trlwe.c:540-552), k = 1. */
int mosfhet_hip_trlwe_extract_tlwe_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out /*[count][N+1]*/, const uint64_t *d_in /*[count][2][N]*/,
                                         int N, int idx, int count, void *stream);
/* ... for any k >= 1 (the loop over the k mask polynomials of src/trlwe.c:540-552) */
int mosfhet_hip_trlwe_extract_tlwe_k_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out /*[count][kN+1]*/, const uint64_t *d_in /*[count][k+1][N]*/,
                                           int k, int N, int idx, int count, void *stream);
/* tlwe_addto over a batch (src/tlwe.c:170-173): d_out[b] += d_in[b], samples of n+1 words. */
int mosfhet_hip_tlwe_addto_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const uint64_t *d_in, int n, int count, void *stream);

/* full_domain_functional_bootstrap over a batch (src/bootstrap.c:519-538): sign bootstrap with the constant test
 * vector 2^62 - 2^(62-precision) at torus_base 2^(precision-1), b -= sign, LWE key switch (ksk: kN -> n), += input,
 * second bootstrap with d_tv at torus_base 2^precision.  Scratch buffers live in the bsk handle. */
int mosfhet_hip_full_domain_functional_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t ksk,
                                                       uint64_t *d_out /*[count][kN+1]*/, const uint64_t *d_tv, int tv_count,
                                                       const uint64_t *d_in /*[count][n+1]*/, int count, int precision, void *stream);

/* multivalue_bootstrap_CLOT21 over a batch (src/bootstrap.c:222-230): one blind rotation at torus_base * n_luts,
 * then n_luts sample extractions N / (n_luts * torus_base) coefficients apart. */
int mosfhet_hip_multivalue_bootstrap_CLOT21_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out /*[count][n_luts][kN+1]*/,
                                                  const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count,
                                                  int torus_base, int n_luts, void *stream);

/* Automorphism key set (Bootstrap_GA_Key.ak, src/bootstrap_ga.c:10, src/keyswitch.c:500-511 with skip_even): N FFT-based
 * TRLWE key-switch keys, entry j switching from s(X^(2j+1)) back to s(X).  h_ak = Torus[N][t][k+1][N] in the TORUS
 * domain (rows KS[j] = TRLWE(s_in(X) 2^(64-(j+1) base_bit)), src/keyswitch.c:12-37); transformed on upload. k = 1. */
int mosfhet_hip_gak_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t *out, const uint64_t *h_ak, int N, int t, int base_bit);
int mosfhet_hip_gak_destroy(mosfhet_hip_gak_t gak);
/* trlwe_eval_automorphism over a batch (src/trlwe.c:775-781 = polynomial_permute, src/polynomial.c:442-450, followed by the
 * FFT-based trlwe_keyswitch, src/keyswitch.c:162-193): d_out[b] = KeySwitch_{ak[(gen-1)/2]}(d_in[b](X^gen)); gen odd, < 2N.
 * With gen = 1 this is trlwe_keyswitch itself (entry 0 switches from s to s). */
int mosfhet_hip_trlwe_eval_automorphism_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t gak, uint64_t *d_out /*[count][2][N]*/,
                                              const uint64_t *d_in /*[count][2][N]*/, int gen, int count, void *stream);
/* functional_bootstrap_ga / functional_bootstrap_wo_extract_ga over a batch (src/bootstrap_ga.c:62-76).  bsk must hold
 * BK_i = TRGSW(X^{s_i}) (new_bootstrap_key_ga, :17-20); gak must have t = l and base_bit = Bg_bit (:10).
 * extract != 0: d_out = TLWE [count][N+1]; else the rotated TRLWE [count][2][N]. */
int mosfhet_hip_functional_bootstrap_ga_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t gak, uint64_t *d_out,
                                              const uint64_t *d_tv, int tv_count, const uint64_t *d_in /*[count][n+1]*/, int count,
                                              int torus_base, int extract, void *stream);

/* Generic set of FFT-based TRLWE key-switch keys (TRLWE_KS_Key, src/keyswitch.c:12-37): `entries` keys of t rows each,
 * h_rows = Torus[entries][t][2][N] in the torus domain.  (mosfhet_hip_gak_create is this with entries = N.) */
int mosfhet_hip_trlwe_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t *out, const uint64_t *h_rows, int entries, int N, int t,
                                 int base_bit);
/* trlwe_keyswitch over a batch (src/keyswitch.c:162-193) with key `entry` of the set; any t. */
int mosfhet_hip_trlwe_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t tks, int entry, uint64_t *d_out /*[count][2][N]*/,
                                      const uint64_t *d_in, int count, void *stream);
/* trlwe_priv_keyswitch_2 over a batch (src/keyswitch.c:52-63); tks = the two keys of trlwe_new_priv_KS_key (:39-50). */
int mosfhet_hip_trlwe_priv_keyswitch_2_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t tks, uint64_t *d_out, const uint64_t *d_in,
                                             int count, void *stream);
/* LWE -> TRLWE packing key (Generic_KS_Key of trlwe_new_packing1_KS_key, src/keyswitch.c:368-390), rows UNcompressed:
 * h_rows = Torus[n][t][2^base_bit - 1][2][N]; and trlwe_packing1_keyswitch over a batch (src/keyswitch.c:458-475). */
int mosfhet_hip_packing1_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_rows, int n, int N, int t,
                                    int base_bit);
int mosfhet_hip_trlwe_packing1_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, uint64_t *d_out /*[count][2][N]*/,
                                               const uint64_t *d_in /*[count][n+1]*/, int count, void *stream);
/* circuit_bootstrap_3 over a batch (src/bootstrap.c:346-366): d_out[b] = TRGSW = Torus[2l][2][N], rows i < l from the private
 * key switch, rows l + i from the packing key switch.  kska = 2-entry key set, kskb = packing key (n = N). */
int mosfhet_hip_circuit_bootstrap_3_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t kska, mosfhet_hip_ksk_t kskb,
                                          uint64_t *d_out /*[count][2l][2][N]*/, const uint64_t *d_in /*[count][n+1]*/, int count,
                                          void *stream);
/* The same with progress events: level_done = l caller-created hipEvent_t (as void *, entries may be NULL) or NULL; level_done[i] is recorded on the
 * launch stream when gadget level i is finished, i.e. rows i and l + i of EVERY output are final -- a host that wants the TRGSWs back can copy a level
 * out (hipMemcpy2DAsync on a stream that waits for the event) while the next level's key switches run: 268 MB per 1024 outputs at lvl2. */
int mosfhet_hip_circuit_bootstrap_3_batch_ev(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t kska, mosfhet_hip_ksk_t kskb,
                                             uint64_t *d_out, const uint64_t *d_in, int count, void *stream, void *const *level_done);

/* ---- callers either side of the bootstrap ---- */
/* public_mux over a batch (src/bootstrap.c:369-389): d_out[b] = (0, p0) + sum_i sel[b][i] * dec_i(p1 - p0); selector rows are
 * torus-domain TRLWEs [count][l][2][N] (the reference takes TRLWE_DFT; the transform is fused); p0, p1 shared by the batch. */
int mosfhet_hip_public_mux_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out /*[count][2][N]*/, const uint64_t *d_p0 /*[N]*/, const uint64_t *d_p1,
                                 const uint64_t *d_sel, int N, int l, int Bg_bit, int count, void *stream);
/* full_domain_functional_bootstrap_KS21 (variant 0, src/bootstrap.c:391-432) / _KS21_2 (variant 1, :434-463) over a batch;
 * d_tv = the 2N-coefficient cleartext test polynomial, pksk = packing key N -> TRLWE(N). */
int mosfhet_hip_full_domain_functional_bootstrap_KS21_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t pksk,
                                                            uint64_t *d_out /*[count][N+1]*/, const uint64_t *d_tv /*[2N]*/,
                                                            const uint64_t *d_in, int count, int torus_base, int variant, void *stream);

/* multivalue_bootstrap_phase1 (src/bootstrap.c:232-243): d_out[b] = torus_base + 1 rotated accumulators [2][N]. */
int mosfhet_hip_multivalue_bootstrap_phase1_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out /*[count][torus_base+1][2][N]*/,
                                                  const uint64_t *d_in, int count, int torus_base, void *stream);
/* multivalue_bootstrap_phase2 (src/bootstrap.c:245-265) for a batch sharing one cleartext LUT h_lut[torus_base] (HOST ints). */
int mosfhet_hip_multivalue_bootstrap_phase2_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out /*[count][N+1]*/, const int *h_lut,
                                                  const uint64_t *d_rotated, int N, int torus_base, int log_torus_base, int count, void *stream);
/* Table-lookup private key switch LWE(m) -> TRLWE(-s m): key of trlwe_new_priv_SK_KS_key_N2 (src/keyswitch.c:611-637), rows
 * uncompressed, h_rows = Torus[n+1][t][2^base_bit-1][2][N] (entry n belongs to the b word); trlwe_priv_keyswitch (:639-656). */
int mosfhet_hip_priv_ksk_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_rows, int n, int N, int t, int base_bit);
int mosfhet_hip_trlwe_priv_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, uint64_t *d_out /*[count][2][N]*/,
                                           const uint64_t *d_in /*[count][n+1]*/, int count, void *stream);
/* circuit_bootstrap (variant 0, src/bootstrap.c:309-322) / circuit_bootstrap_2 (variant 1, :324-344): kska = private key-switch
 * key (priv_ksk_create, n = N), kskb = packing key. */
int mosfhet_hip_circuit_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t kska, mosfhet_hip_ksk_t kskb,
                                        uint64_t *d_out /*[count][2l][2][N]*/, const uint64_t *d_in, int count, int variant, void *stream);
int mosfhet_hip_circuit_bootstrap_batch_ev(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t kska, mosfhet_hip_ksk_t kskb,
                                           uint64_t *d_out, const uint64_t *d_in, int count, int variant, void *stream,
                                           void *const *level_done /* as in mosfhet_hip_circuit_bootstrap_3_batch_ev */);

/* functional_bootstrap_trgsw_phase1 (src/bootstrap.c:284-295): blind rotation with a TRGSW accumulator; d_out_dft[b] = TRGSW_DFT(X^-phase)
 * as [2l][2][N/2] complex in the engine's slot order (the layout of one bootstrap-key entry).  phase2 (:297-306): d_out[b] =
 * SampleExtract_0(tv (.) d_in_dft[b]); tv_count = 1 (shared) or count.  bsk supplies N, l, Bg_bit in phase 2. */
int mosfhet_hip_functional_bootstrap_trgsw_phase1_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, double *d_out_dft, const uint64_t *d_in,
                                                        int count, int torus_base, void *stream);
int mosfhet_hip_functional_bootstrap_trgsw_phase2_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out /*[count][N+1]*/,
                                                        const double *d_in_dft, const uint64_t *d_tv, int tv_count, int count, void *stream);

/* trlwe_tensor_prod_FFT (src/trlwe.c:727-771) over a batch of pairs; rlk = 1-entry key set (mosfhet_hip_trlwe_ksk_create) holding the rows of
 * trlwe_new_RL_key (src/keyswitch.c:3-10).  tlwe_mul (src/tlwe.c:322-332): LWE x LWE under the extracted key, pksk = packing key. */
int mosfhet_hip_trlwe_tensor_prod_FFT_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t rlk, uint64_t *d_out /*[count][2][N]*/, const uint64_t *d_in1,
                                            const uint64_t *d_in2, int precision, int count, void *stream);
int mosfhet_hip_tlwe_mul_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t pksk, mosfhet_hip_gak_t rlk, uint64_t *d_out /*[count][N+1]*/, const uint64_t *d_in1,
                               const uint64_t *d_in2, int precision, int count, void *stream);
/* full_domain_functional_bootstrap_CLOT21 (variant 0, src/bootstrap.c:465-491: d_tv = two TRLWE test vectors [2][2][N]) and _CLOT21_2
 * (variant 1, :493-517: d_tv = 2^(precision-1) cleartext LUT words, on the device). */
int mosfhet_hip_full_domain_functional_bootstrap_CLOT21_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t pksk, mosfhet_hip_gak_t rlk,
                                                              uint64_t *d_out /*[count][N+1]*/, const uint64_t *d_tv, const uint64_t *d_in, int count,
                                                              int precision, int variant, void *stream);

/* Bootstrap key with blind-rotate unfolding u = 2, 4 or 8 (new_bootstrap_key(.., unfolding), src/bootstrap.c:23-48; the reference's group
 * stride 2^u / u is an integer quotient, :34-45, so other factors make overlapping groups there and are rejected here): h_su =
 * Torus[n 2^u / u][2l][2][N], torus domain; n divisible by u.  The handle works with functional_bootstrap[_wo_extract]_batch and the
 * compositions built on them (they take the blind_rotate_unfolded branch, src/bootstrap.c:124-149,196-197). */
int mosfhet_hip_bsk_unfolded_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *h_su, int n, int N, int l, int Bg_bit,
                                    int unfolding);

/* Bootstraps of few ciphertexts with an unfolded key assemble the selectors of all key groups side by side first (two launches per chunk of `max_batch`
 * ciphertexts, up to 8 chunks; same results bit for bit); larger batches run one fused kernel.  -1: default (64, capped by 2 GiB of selectors), 0: never. */
int mosfhet_hip_set_unfold_split_max(int max_batch);
/* Unfolding 2 rotates with the per-group TRGSW assembled in the DFT domain (the key's samples are transformed once, when the handle is made; per step
 * and ciphertext the transforms of ONE CMUX for two mask words: mosfhet_amd/csrc/unfold_kernels.h).  Same mathematics as src/bootstrap.c:124-149, another
 * order of floating-point operations (oracle/oracle_ext.c:orc_blind_rotate_unfolded2_dft mirrors it bit for bit; it agrees with the reference's own
 * results to FFT rounding).  on = 0 (or MOSFHET_HIP_UNFOLD2_DFT=0) selects the torus-domain assembly, as for u = 4 and 8. */
int mosfhet_hip_set_unfold2_dft(int on);
/* the same key encrypted on the device (torus-domain samples, generator and secrets as mosfhet_hip_bsk_generate) */
int mosfhet_hip_bsk_unfolded_generate(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *h_s_rlwe /*[N]*/, int N, const uint64_t *h_s_lwe /*[n]*/, int n,
                                      int l, int Bg_bit, double sigma, uint64_t seed, int unfolding);

/* multivalue_bootstrap_UBR_phase1 / phase2 (src/bootstrap.c:151-190) with an unfolded key: d_sa = [count][n/u][2l][2][N/2] complex
 * (the per-group TRGSW_DFT of each input); phase 2 evaluates tv_count shared test vectors per ciphertext: d_out = [count][tv_count][N+1]. */
int mosfhet_hip_multivalue_bootstrap_UBR_phase1_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, double *d_sa, const uint64_t *d_in, int count,
                                                      void *stream);
int mosfhet_hip_multivalue_bootstrap_UBR_phase2_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out, const uint64_t *d_tvs,
                                                      int tv_count, const uint64_t *d_in, const double *d_sa, int count, int torus_base,
                                                      void *stream);

/* trlwe_mv_extract_tlwe (mode 0: d_out = [count][amount][N+1]), _scaling (1), _scaling_addto (2), _scaling_subto (3: d_out = [count][N+1])
 * (src/trlwe.c:580-622); `amount` is the reference's amount / scale argument. */
int mosfhet_hip_trlwe_mv_extract_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out, const uint64_t *d_in /*[count][2][N]*/, int N, int mode, int amount,
                                       int count, void *stream);

/* On-device generation of a table-lookup TRLWE key (the 6 GB packing key of config 4 in milliseconds): kind 0 = packing key
 * (trlwe_new_packing1_KS_key, src/keyswitch.c:368-390), kind 1 = private key (trlwe_new_priv_SK_KS_key_N2, :611-637); rows are fresh
 * TRLWE encryptions under the binary key h_s_out[N] (exact a * s, Gaussian noise sigma, counter-based generator seeded by `seed`).
 * ksk_export_rows copies rows [first_row, first_row + count) back for inspection. */
int mosfhet_hip_trlwe_table_ksk_generate(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, int kind, const uint64_t *h_s_out, int N,
                                         const uint64_t *h_s_in, int n, int t, int base_bit, double sigma, uint64_t seed);
int mosfhet_hip_ksk_export_rows(mosfhet_hip_ksk_t ksk, size_t first_row, size_t count, uint64_t *h_out);
/* Seed-compressed form (the reference's USE_COMPRESSED_TRLWE key rows, src/keyswitch.c:231-241, regenerated by
 * trlwe_compressed_subto, src/trlwe_compressed_vaes.c:139-160): the same rows as mosfhet_hip_trlwe_table_ksk_generate makes for this seed, but
 * only the b halves stay in HBM (mosfhet_hip_ksk_bytes: half); every key switch that takes the handle regenerates the mask words inside the
 * kernel from (seed, row, word).  Results are bit-identical to the uncompressed key's.  ksk_export_rows expands the rows. */
int mosfhet_hip_trlwe_table_ksk_generate_compressed(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, int kind, const uint64_t *h_s_out, int N,
                                                    const uint64_t *h_s_in, int n, int t, int base_bit, double sigma, uint64_t seed);
/* tlwe_keyswitch_no_precomp (src/tlwe.c:305-320): key = plain device rows [n_in][t][n_out + 1], one sample per (input word, digit position), multiplied by the
 * digit in the kernel (the reference's double rounding offset included) */
int mosfhet_hip_tlwe_keyswitch_no_precomp_batch(mosfhet_hip_ctx_t ctx, const uint64_t *d_rows, uint64_t *d_out, const uint64_t *d_in, int count, int n_in, int n_out,
                                                int t, int base_bit, void *stream);
/* LUT-packing key switch (trlwe_new_packing_KS_key / trlwe_packing_keyswitch, src/keyswitch.c:214-241,346-366): `torus_base` LWE samples into the `torus_base`
 * slots of one TRLWE sample.  The key is a table key with n * torus_base digit sources (ksk kind 2: export / import / alloc as such);
 * d_in [count][torus_base][n + 1], d_out [count][2][N]. */
int mosfhet_hip_trlwe_lut_packing_ksk_generate(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_s_out /*[N]*/, int N, const uint64_t *h_s_in /*[n]*/,
                                               int n, int t, int base_bit, int torus_base, double sigma, uint64_t seed);
int mosfhet_hip_trlwe_lut_packing_keyswitch_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, int torus_base, uint64_t *d_out, const uint64_t *d_in, int count,
                                                  void *stream);
size_t mosfhet_hip_ksk_bytes(mosfhet_hip_ksk_t ksk);                     /* device bytes of a key-switch table */
/* On-device generation of the bootstrap key (new_bootstrap_key without unfolding, src/bootstrap.c:3-21: BK_i = TRGSW(s_i); ga != 0: the
 * TRGSW(X^{s_i}) samples of new_bootstrap_key_ga, src/bootstrap_ga.c:5-24) from the binary TRLWE key h_s_rlwe[N] and LWE key h_s_lwe[n]:
 * encrypted in the torus domain by the counter-based generator (exact a * s, Gaussian noise sigma), then transformed -- no host key, no upload. */
int mosfhet_hip_bsk_generate(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *h_s_rlwe, int N, const uint64_t *h_s_lwe, int n, int l, int Bg_bit,
                             double sigma, uint64_t seed, int ga);
/* ... for any k <= 3 and any ring the engine serves up to N = 8192 (keys of the general-ring path): h_s_rlwe = the k key polynomials [k][N]; the k + 1 components of a
 * row as trgsw_monomial_sample makes them (src/trgsw.c:152-168).  k = 1 on a tuned ring is mosfhet_hip_bsk_generate(.., ga = 0). */
int mosfhet_hip_bsk_generate_k(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const uint64_t *h_s_rlwe /*[k][N]*/, int k, int N, const uint64_t *h_s_lwe, int n, int l,
                               int Bg_bit, double sigma, uint64_t seed);
/* On-device generation of FFT-based TRLWE key-switch keys: entry e, row r < t = TRLWE_{s_out}(h_msgs[e](X) * 2^(64 - (r+1) base_bit)), then the
 * engine's forward transform.  h_msgs = Torus[entries][N] holds the polynomial each entry switches FROM: the other key (trlwe_new_KS_key,
 * src/keyswitch.c:12-37), its Galois images s(X^(2e+1)) for e < N (trlwe_new_automorphism_KS_keyset, :500-511), (-s * s_in, -s) (trlwe_new_priv_KS_key,
 * :39-50) or s^2 (trlwe_new_RL_key, :3-10). */
int mosfhet_hip_trlwe_ksk_generate(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t *out, const uint64_t *h_s_out, int N, const uint64_t *h_msgs, int entries, int t,
                                   int base_bit, double sigma, uint64_t seed);
/* On-device generation of the LWE -> LWE key-switch table of tlwe_new_KS_key (src/tlwe.c:193-212) from the two binary keys: rows
 * TLWE_{s_out}(s_in[i] v 2^(64 - (j+1) base_bit)), masks from the counter-based generator, Gaussian noise sigma.  compressed != 0 stores one word
 * per row (b) and tlwe_keyswitch regenerates the masks inside the kernel: lvl2's 1.2 GB table becomes 2 MB, results are bit-identical to the
 * uncompressed key of the same seed.  The handle works with every entry point that takes an LWE key-switch key. */
int mosfhet_hip_tlwe_ksk_generate(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, const uint64_t *h_s_out, int n_out, const uint64_t *h_s_in, int n_in, int t,
                                  int base_bit, double sigma, uint64_t seed, int compressed);

/* Kernel selection for the N = 1024 bootstraps: batches of at most `max_batch` ciphertexts run the latency-oriented kernel (one workgroup
 * of 2l wavefronts per ciphertext, ~1/3 of the latency), larger ones the throughput kernel (one wavefront per ciphertext).  Results are
 * bit-identical.  Default 512 (or env MOSFHET_HIP_TEAM_MAX); 0 disables the latency kernel. */
int mosfhet_hip_set_team_max_batch(int max_batch);
/* the same switch-over for N = 2048 and, at half the value, N = 4096 (one workgroup of two transform teams per ciphertext; default 512, MOSFHET_HIP_WIDE_TEAM_MAX; 0 disables) */
int mosfhet_hip_set_wide_team_max_batch(int max_batch);
/* N = 2048, l = 2, 4 or 6 (l = 4: the TFHEpp lvl2 set of BASELINE.json configs[2..4]; l = 6: the radix-integer application's): batches of at most `max_batch` bootstraps take TWO CUs each (pbs_split_kernel: one workgroup per
 * accumulator component, one 16 KiB exchange per CMUX step) -- the share one GPU gets when a config's batch is sharded over eight leaves half its CUs idle otherwise.
 * -1 = half the device's CUs (default, MOSFHET_HIP_SPLIT_MAX), 0 = never.  The ONE selection that changes bits: that kernel adds the external product's rows per
 * accumulator component and then the two partial sums (src/trgsw.c:393-419 is one chain over all rows); the results differ from every other kernel's by FFT-level
 * rounding -- the tolerance the reference's own tests accept between its two FFT back-ends -- and are bit-identical to the oracle's restatement of that order
 * (oracle/oracle_tfhe.c: orc_set_product_order). */
int mosfhet_hip_set_split_max_batch(int max_batch);
/* Table key switches (tlwe_keyswitch, trlwe_packing1_keyswitch, trlwe_priv_keyswitch: src/tlwe.c:289-303, src/keyswitch.c:458-475,639-656) with 2 - 4 digit bits:
 * from `min_count` ciphertexts on they run with OUTPUT WORDS on the lanes (wave-uniform digits pick the candidate by register-relative addressing: one scalar move and
 * one 64-bit add per ciphertext, input word, digit position and output word) instead of ciphertexts on the lanes (a per-lane LDS gather).  Integer sums: the same
 * bits either way.  Default 17 = every batch the direct kernels of up to 16 ciphertexts leave (MOSFHET_HIP_KS_WORDS); 0 = never. */
int mosfhet_hip_set_ks_words(int min_count);
/* wavefronts of that kernel whose bounded wait on their workgroup's counters ran out since the library was loaded (synchronises the device): 0 unless something is broken */
int mosfhet_hip_ks_words_gave_up(mosfhet_hip_ctx_t ctx, unsigned int *count);
/* its launch plan for a shape, without a device (tests): plan = { taken at the current setting, digit positions per stage, stages per input word, LDS-DMA requests per wavefront
 * and stage, LDS bytes per workgroup, workgroups, ciphertext groups of 512, input-word splits }; compressed_lwe: seed-compressed LWE rows (they stay with the other form) */
int mosfhet_hip_ks_words_plan(int count, int n_in, int row, int t, int base_bit, int compressed_lwe, long long plan[8]);
/* how long (10 ns ticks; default 200000 = 2 ms, MOSFHET_HIP_SPLIT_LIMIT) the first workgroup of such a pair waits for its partner before it takes the whole bootstrap
 * alone (same summation order, same bits); 0 = always alone (test switch) */
int mosfhet_hip_set_split_wait_limit(int ticks);
/* of the calling host thread's last split launch (synchronises its stream): bootstraps taken by a pair of workgroups / alone */
int mosfhet_hip_split_last_launch(int *count, int *paired, int *alone);

/* Unit-loop form of the external-product kernel on rings of two wavefronts per team (N = 2048, l = 4; trgsw_mul_trlwe_DFT, src/trgsw.c:385-423).  The
 * software-pipelined loop is taken only by the instantiation that has been soaked clean and only while its build has no scratch (capi.hip: ep_go); the plain
 * loop gives the same bits 11 % slower.  set_ep_plain_loop(1) forces the plain loop for every multi-wavefront instantiation (tests run both in one
 * process; MOSFHET_HIP_EP_PAIRS=0 does the same for a whole process).  ep_kernel_info(i, ...) reports, for the i-th such instantiation launched so far,
 * its name, the scratch bytes per lane of its pipelined build and whether the launcher takes that build; MOSFHET_HIP_EINVAL past the end. */
int mosfhet_hip_set_ep_plain_loop(int on);
int mosfhet_hip_ep_kernel_info(int i, const char **name, int *scratch_bytes, int *takes_pipelined);
/* N >= 2048 bootstraps re-align their teams per XCD every few CMUX steps (a bounded wait: timing only).  A launch whose wait ran out (the chip was shared
 * with another launch) makes the next MOSFHET_HIP_PACE_SKIP (16) paced launches of the device skip the rendezvous; this returns how many are still to skip. */
int mosfhet_hip_pace_skip_credit(mosfhet_hip_ctx_t ctx, int *credit);

/* Digit-parallel forms of the radix-integer callers of applications/multi-ciphertext-arith for M INDEPENDENT integers: the gate sequences of
 * ufhe_sl_add_integer / ufhe_sub_integer (src/integer.c:79-107,136-156), ufhe_relu_integer (src/ml.c:4-20) and ufhe_encrypted_tlwe_lut (src/lut.c:6-20), one
 * key-switch / packing-switch / bootstrap LAUNCH per carry step or tree level instead of one gate per digit and integer.  Integers are digit-major on the device:
 * [d][M][N+1] torus words, digit / (2 torus_base) per sample under the extracted TRLWE key (ufhe_encrypt_integer).  The handle borrows the keys (bootstrap key
 * n -> N, LWE key switch N -> n, LUT-packing key of torus_base slots N -> TRLWE(N); pksk may be NULL for add / sub only) and is read-only after creation.
 *   addsub: c = a + b (subtract = 0) or a - b (1) over all d digits, the carry out of the top digit dropped (equal-width operands); c must not alias a, b
 *   relu: out = in > 0 ? in : 0 for signed integers; out may alias in
 *   encrypted_lut: table[0][m] = table[selector_m][m]; d_table [size][M][N+1] is consumed, d_sel [log_B size][M][N+1] = selector digits, least significant first
 *   cmp: c[m] = 0 / 1 / 2 for a[m] < / = / > b[m] (ufhe_cmp_integer, src/integer.c:205-264); d_c [M][N+1] is the one result digit
 *   lut_cleartext: out[m] = h_lut[selector_m] for a CLEARTEXT table (ufhe_lut_integer, src/lut.c:23-47): one multi-value rotation per selector, then the tree */
typedef struct mosfhet_hip_vec *mosfhet_hip_vec_t;
int mosfhet_hip_vec_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_vec_t *out, mosfhet_hip_bsk_t bsk, mosfhet_hip_ksk_t ksk, mosfhet_hip_ksk_t pksk, int torus_base);
int mosfhet_hip_vec_destroy(mosfhet_hip_vec_t vec);
int mosfhet_hip_vec_addsub(mosfhet_hip_vec_t vec, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, int M, int d, int subtract, void *stream);
int mosfhet_hip_vec_relu(mosfhet_hip_vec_t vec, uint64_t *d_out, const uint64_t *d_in, int M, int d, void *stream);
int mosfhet_hip_vec_encrypted_lut(mosfhet_hip_vec_t vec, uint64_t *d_table, const uint64_t *d_sel, int size, int M, void *stream);
int mosfhet_hip_vec_cmp(mosfhet_hip_vec_t vec, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, int M, int d, int a_signed, int b_signed, void *stream);
/*   mul: c = a * b (ufhe_mul_integer, src/integer.c:166-203): schoolbook over a's digits -- per digit one multi-value rotation, two packed product tables per integer,
 *        two bootstrap launches over all digits of b, a shifted addition and a shifted accumulation; a [da][M][N+1], b [db][M][N+1], c [dc][M][N+1], no aliasing */
/*   mux_array: out[m] = vec[selector_m][m] over arrays of `size` integers of d digits (ufhe_mux_integer_array, src/lut.c:49-64): all d trees of all M instances as ONE
 *        tree over d M instances; d_tables [size][d][M][N+1] is consumed, d_sel [log_B size][M][N+1], d_out [d][M][N+1] */
int mosfhet_hip_vec_mux_array(mosfhet_hip_vec_t vec, uint64_t *d_out, uint64_t *d_tables, const uint64_t *d_sel, int size, int d, int M, void *stream);
int mosfhet_hip_vec_sl_add(mosfhet_hip_vec_t vec, uint64_t *d_c, int dc, const uint64_t *d_a, int da, int g, const uint64_t *d_b, int db, int h, int is_signed, int M,
                           void *stream);   /* c = a B^g + b B^h (ufhe_sl_add_integer, src/integer.c:79-107) */
int mosfhet_hip_vec_extend(mosfhet_hip_vec_t vec, uint64_t *d_c, int dc, int d_ini, int is_signed, int M, void *stream);   /* digits [d_ini, dc) <- zero or sign (ufhe_extend_integer, :62-77) */
int mosfhet_hip_vec_mul(mosfhet_hip_vec_t vec, uint64_t *d_c, int dc, const uint64_t *d_a, int da, const uint64_t *d_b, int db, int is_signed, int M, void *stream);
int mosfhet_hip_vec_lut_cleartext(mosfhet_hip_vec_t vec, uint64_t *d_out, const uint64_t *d_sel, const uint64_t *h_lut, int size, int d_out_digits, int M, void *stream);

/* The canonical caller pattern in one call (applications/multi-ciphertext-arith/src/integer.c:94-95): tlwe_keyswitch kN -> n, then
 * functional_bootstrap (extract = 1: d_out [count][kN+1]) or functional_bootstrap_wo_extract (extract = 0: d_out [count][k+1][N]);
 * same stream, no host synchronisation in between. */
int mosfhet_hip_keyswitch_functional_bootstrap_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t ksk, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                                     const uint64_t *d_tv, int tv_count, const uint64_t *d_in /*[count][kN+1]*/, int count,
                                                     int torus_base, int extract, void *stream);

/* CMUX over a batch with one shared selector = entry `key_index` of a key handle (e.g. circuit-bootstrap outputs turned into a handle by
 * mosfhet_hip_bsk_create_from_device): d_out[b] = d_in0[b] + key (.) (d_in1[b] - d_in0[b]); d_out may alias d_in0.  The leveled caller of
 * the path (applications/leveled_lut/vertical_packing.c:24-52: CMUX tree, then blind_rotate with the selectors as key). */
int mosfhet_hip_cmux_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, int key_index, uint64_t *d_out /*[count][2][N]*/, const uint64_t *d_in0,
                           const uint64_t *d_in1, int count, void *stream);

/* Key images for the on-disk formats (save_bootstrap_key / load_new_bootstrap_key src/bootstrap.c:63-104, trlwe_save_KS_key /
 * trlwe_load_new_KS_key src/keyswitch.c:122-160, tlwe_save_KS_key / tlwe_load_new_KS_key src/tlwe.c:247-287, trlwe_save_generic_ks_key /
 * trlwe_load_new_generic_ks_key src/keyswitch.c:409-455).  DFT-domain contents are backend-defined in the reference too (src/polynomial.c:336-357):
 * an image is the engine's own layout (tag mosfhet_hip_dft_layout_id), mosfhet_hip_bsk_bytes / mosfhet_hip_trlwe_ksk_bytes long, and exact -- an
 * exported and re-imported key gives bit-identical results.  An unfolded bootstrap key's image is its torus-domain samples.  Table keys are the
 * reference's uncompressed row order and stream row-wise (ksk_export_rows / ksk_alloc + ksk_import_rows), 2N or n_out + 1 words per row.
 * info arrays: bsk {n, k, N, l, Bg_bit, unfolding}; trlwe_ksk {entries, N, t, base_bit}; ksk {n_in, row words, b_word, t, base_bit, kind}
 * with kind 0 = LWE -> LWE, 1 = packing, 2 = private (n_in counts the b entry). */
unsigned mosfhet_hip_dft_layout_id(void);
int mosfhet_hip_bsk_info(mosfhet_hip_bsk_t bsk, int *out6);
int mosfhet_hip_bsk_export(mosfhet_hip_bsk_t bsk, void *h_out);
int mosfhet_hip_bsk_import(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const void *h_image, int n, int k, int N, int l, int Bg_bit, int unfolding);
int mosfhet_hip_trlwe_ksk_info(mosfhet_hip_gak_t tks, int *out4);
size_t mosfhet_hip_trlwe_ksk_bytes(mosfhet_hip_gak_t tks);
int mosfhet_hip_trlwe_ksk_export(mosfhet_hip_gak_t tks, void *h_out);
int mosfhet_hip_trlwe_ksk_import(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t *out, const void *h_image, int entries, int N, int t, int base_bit);
int mosfhet_hip_ksk_info(mosfhet_hip_ksk_t ksk, int *out6);
int mosfhet_hip_ksk_alloc(mosfhet_hip_ctx_t ctx, mosfhet_hip_ksk_t *out, int kind, int n, int n_out_or_N, int t, int base_bit);
int mosfhet_hip_ksk_import_rows(mosfhet_hip_ksk_t ksk, size_t first_row, size_t count, const uint64_t *h_in);

/* Key replication for several GPUs (keys replicated per GPU, optionally by hipMemcpyPeer over xGMI instead of host copies): a copy of
 * `src` on the device of `dst_ctx`, DEVICE TO DEVICE -- hipMemcpyPeer with peer access enabled where the two devices allow it (xGMI on an MI355X node),
 * HIP's own staged copy otherwise, and a pinned 64 MiB host bounce buffer when even that is refused.  Every field travels as it is: a seed-compressed
 * table key stays compressed (3 GB instead of the 6 GB its exported rows take at BASELINE configs[3]), an unfolded key keeps its torus-domain samples.
 * Same bits on both devices, so results do not depend on which replica serves a slice.  The source key's device need not be the caller's current one.
 * mosfhet_hip_last_clone_route(): how the calling thread's last clone travelled -- 0 same device, 1 peer to peer, 2 device to device without peer
 * access (HIP stages it), 3 host bounce buffer. */
int mosfhet_hip_bsk_clone(mosfhet_hip_ctx_t dst_ctx, mosfhet_hip_bsk_t *out, mosfhet_hip_bsk_t src);
int mosfhet_hip_ksk_clone(mosfhet_hip_ctx_t dst_ctx, mosfhet_hip_ksk_t *out, mosfhet_hip_ksk_t src);
int mosfhet_hip_gak_clone(mosfhet_hip_ctx_t dst_ctx, mosfhet_hip_gak_t *out, mosfhet_hip_gak_t src);
int mosfhet_hip_last_clone_route(void);
size_t mosfhet_hip_gak_bytes(mosfhet_hip_gak_t gak);                     /* device bytes of an FFT key-switch key set (= mosfhet_hip_trlwe_ksk_bytes) */

/* Timing hook for bench.py: runs `reps` launches of the programmable-bootstrap kernel on `stream`
 * bracketed by hipEvents ON THAT STREAM and returns the average kernel time in milliseconds
 * (synchronises the stream). */
int mosfhet_hip_time_programmable_bootstrap(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, uint64_t *d_out,
                                            const uint64_t *d_tv, int tv_count, const uint64_t *d_in, int count,
                                            int precision, int reps, void *stream, float *ms_per_launch);

/* Secret of the on-device key generators (bsk_generate, *_ksk_generate): the 256-bit ChaCha20 key their NOISE terms are drawn under.  It is independent
 * of the public 64-bit mask seed those calls take (masks are public and regenerable from that seed; noise is not).  Drawn from the operating system at
 * first use unless set here (32 bytes); process-wide.
 * Every generate call draws its noise under a ChaCha20 nonce no other call under this secret gets (a call counter and the generator kind), so the
 * `seed` arguments need NOT be unique: two keys made with the same seed share their masks, never their noise.  Setting the secret restarts that
 * sequence -- the same secret followed by the same generate calls reproduces the same keys (reproducible test runs; the host layer does this under
 * mosfhet_seed).  Never install one secret twice for keys that go to different parties. */
int mosfhet_hip_set_keygen_secret(const void *key32);

/* ---- DFT-level entry points behind the reference's legacy signatures (mosfhet.h:179-182,263-264,342-344,454,296 of the reference; csrc/capi_dft.inc).
 * DFT-domain objects are device arrays of doubles in the engine's slot order ([polynomial][N/2] complex). ---- */
/* non-owning key handle over n consecutive TRGSW_DFT entries ([(k+1)l][k+1][N/2] complex each) already on the device; destroy with mosfhet_hip_bsk_destroy */
int mosfhet_hip_bsk_view_create(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t *out, const double *d_dft, int n, int k, int N, int l, int Bg_bit);
const double *mosfhet_hip_bsk_device_dft(mosfhet_hip_bsk_t bsk);          /* device address of entry 0 (NULL: unfolded key) */
/* trgsw_mul_trlwe_DFT (src/trgsw.c:385-423), result left in the DFT domain: d_out_dft [count][k+1][N/2] complex; key_stride in doubles, 0 = shared TRGSW */
int mosfhet_hip_external_product_dft_batch(mosfhet_hip_ctx_t ctx, const double *d_trgsw_dft, size_t key_stride, double *d_out_dft,
                                           const uint64_t *d_in /*[count][k+1][N]*/, int N, int l, int Bg_bit, int count, void *stream);
/* public_mux (src/bootstrap.c:369-389) with the selector rows in the DFT domain: d_sel_dft [count][l][2][N/2] complex */
int mosfhet_hip_public_mux_dft_batch(mosfhet_hip_ctx_t ctx, uint64_t *d_out /*[count][2][N]*/, const uint64_t *d_p0 /*[N]*/, const uint64_t *d_p1,
                                     const double *d_sel_dft, int N, int l, int Bg_bit, int count, void *stream);
/* blind_rotate_ga (src/bootstrap_ga.c:35-60) in place on d_acc [count][2][N]; d_in [count][n+1] (mask words) */
int mosfhet_hip_blind_rotate_ga_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_bsk_t bsk, mosfhet_hip_gak_t gak, uint64_t *d_acc, const uint64_t *d_in, int count,
                                      void *stream);
/* polynomial_add_DFT_polynomials / _sub_ / _scale_and_add_ (src/polynomial.c:102-128) and the TRLWE_DFT / TRGSW_DFT sums built on them, over flat
 * device arrays: d_out[i] = (d_a ? d_a[i] : 0) + cb * d_b[i] (d_out may alias d_a or d_b) */
int mosfhet_hip_dft_lincomb_batch(mosfhet_hip_ctx_t ctx, double *d_out, const double *d_a, const double *d_b, double cb, size_t n_doubles, void *stream);
/* trlwe_eval_automorphism (src/trlwe.c:775-781) with key-set entry `entry` (the batch entry point above uses entry (gen - 1) / 2) */
int mosfhet_hip_trlwe_eval_automorphism_entry_batch(mosfhet_hip_ctx_t ctx, mosfhet_hip_gak_t gak, int entry, uint64_t *d_out, const uint64_t *d_in, int gen,
                                                    int count, void *stream);

#ifdef __cplusplus
}
#endif
#endif
