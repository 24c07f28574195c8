/* compat_internal.h -- helpers shared by the files of the host layer (not part of any public header; hidden visibility). */
#ifndef MOSFHET_COMPAT_INTERNAL_H
#define MOSFHET_COMPAT_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

#include "mosfhet_compat.h"
#include "mosfhet_hip.h"

#define MC_HIDDEN __attribute__((visibility("hidden")))

/* HIP runtime entry points used for staging buffers (declared here to keep the host layer plain C) */
extern int hipMalloc(void **ptr, size_t size);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind);
extern int hipMemset(void *dst, int value, size_t size);
extern int hipHostMalloc(void **ptr, size_t size, unsigned int flags);
extern int hipStreamCreate(void **stream);
extern int hipStreamCreateWithFlags(void **stream, unsigned int flags);
#define HIP_STREAM_NON_BLOCKING 1u
extern int hipStreamSynchronize(void *stream);
extern int hipMemcpyAsync(void *dst, const void *src, size_t size, int kind, void *stream);
extern int hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, int kind, void *stream);
extern int hipHostFree(void *ptr);
extern int hipEventCreateWithFlags(void **event, unsigned flags);
extern int hipEventRecord(void *event, void *stream);
extern int hipEventSynchronize(void *event);
extern int hipEventDestroy(void *event);
extern int hipStreamWaitEvent(void *stream, void *event, unsigned flags);
#define HIP_EVENT_DISABLE_TIMING 2u
extern int hipSetDevice(int device);
extern int hipGetDevice(int *device);
#define HIP_H2D 1
#define HIP_D2H 2
#define HIP_D2D 3

/* mosfhet_compat.c */
MC_HIDDEN void mc_die(const char *what);                       /* prints the C ABI's last error and aborts (the reference's assert / exit behaviour) */
MC_HIDDEN void *mc_xmalloc(size_t sz);                         /* 64-byte aligned, exits on failure (src/misc.c:115-128) */
MC_HIDDEN void *mc_dev_alloc(size_t bytes);                    /* hipMalloc on the engine's device, aborts on failure */
MC_HIDDEN void mc_dev_copy(void *dst, const void *src, size_t bytes, int kind);
MC_HIDDEN void *mc_stage_alloc(size_t bytes);                  /* per-thread growable device staging buffer */
MC_HIDDEN void *mc_hstage_alloc(size_t bytes);                 /* per-thread pinned host staging (a few cached slots) */
MC_HIDDEN void mc_hstage_free(void *p);
MC_HIDDEN void mc_use_device(void);                            /* make the engine's device current on the calling thread */
MC_HIDDEN void mc_trlwe_to_flat(Torus *flat, TRLWE c);
MC_HIDDEN void mc_trlwe_from_flat(TRLWE c, const Torus *flat);

/* mosfhet_compat_multi.c: several devices behind the API.  Device INDEX d = position in the list given to mosfhet_set_devices; index 0 is the primary */
#define MC_MAX_DEVICES 16
enum { MC_KEY_BSK = 0, MC_KEY_KSK = 1, MC_KEY_GAK = 2 };   /* bootstrap key, table key (LWE / packing / private), FFT key-switch key set */
extern MC_HIDDEN int g_mc_ndev;                    /* devices in use */
extern MC_HIDDEN int g_mc_devs[MC_MAX_DEVICES];    /* their HIP ordinals */
extern MC_HIDDEN __thread int t_mc_in_shard;      /* inside one slice of a sharded batch (no nested sharding) */
extern MC_HIDDEN __thread int t_mc_dev;            /* device index of the calling thread (workers of a sharded batch: their slice's device) */
MC_HIDDEN int mc_engine_started(void);
MC_HIDDEN mosfhet_hip_ctx_t mc_ctx_of(int d);      /* context of device index d */
MC_HIDDEN void mc_devices_from_env(void);
MC_HIDDEN void *mc_key_here(void *primary, int kind);   /* the key's handle on the calling thread's device */
MC_HIDDEN void mc_replicas_free(void *primary);
typedef void (*mc_slice_fn)(void *args, int lo, int hi);
MC_HIDDEN void mc_run_sharded(mc_slice_fn fn, void *args, int count, void *const *keys, const int *kinds, int n_keys);

/* csprng.c: ChaCha20 generator, one stream per host thread */
MC_HIDDEN uint64_t mc_rnd64(void);
MC_HIDDEN void mc_rnd_bytes(void *out, size_t bytes);
MC_HIDDEN double mc_rnd_normal(double sigma);
MC_HIDDEN void mc_chacha20_block(uint32_t out[16], const uint32_t key[8], uint64_t counter, uint64_t nonce);

/* polynomial shells carry a hidden tag in front of the struct, so that free_polynomial / free_trlwe / free_trgsw -- which the reference
 * declares on void * and uses for torus-domain and DFT-domain objects alike -- know what they hold */
enum { MC_POLY_TORUS = 0x544f5255, MC_POLY_DFT_OWNER = 0x4446544f, MC_POLY_DFT_VIEW = 0x44465456, MC_POLY_DFT_SHARED = 0x44465453 };
MC_HIDDEN void *mc_poly_shell(int kind, void *coeffs, int N);  /* allocates {tag | struct {coeffs, N}} and returns the struct */
/* One device block behind an ARRAY of DFT-domain objects (polynomial_new_array_of_polynomials_DFT, trlwe_ / trgsw_alloc_new_DFT_sample_array): every
 * element holds a reference, the block is released with the last one -- so elements may be freed one by one in any order, as in the reference, where
 * each element is its own allocation (src/trgsw.c:82-88). */
typedef struct { void *dev; long refs; } McShare;
MC_HIDDEN McShare *mc_share_new(void *dev, long refs);
MC_HIDDEN void *mc_poly_shell_shared(McShare *share, void *coeffs, int N);
MC_HIDDEN int mc_poly_kind(const void *poly);

/* mosfhet_compat_dft.c */
MC_HIDDEN TRGSW_DFT *mc_trgsw_dft_views(double *base, int n, int l, int Bg_bit, int N);   /* n non-owning TRGSW_DFT over consecutive key entries */
MC_HIDDEN void mc_trgsw_dft_views_free(TRGSW_DFT *views, int n);

#endif
