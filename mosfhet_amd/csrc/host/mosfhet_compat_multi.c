/*
 * mosfhet_compat_multi.c -- several GPUs behind the MOSFHET-compatible API (every bootstrap / key switch is an independent unit,
 * batches shard with no exchange step, keys are replicated per GPU).
 *
 *   mosfhet_set_devices(n, ids)   before the first call: the process uses those GPUs (ids[0] is the primary device, where single-sample calls run
 *                                 and keys are created); env MOSFHET_HIP_DEVICES="0,1,..." does the same.
 * A *_batch entry point then cuts its batch into contiguous slices (mosfhet_amd/shard.py: the first count % n slices get one unit more) and runs
 * slice d on device d from its own host thread, with that thread's stream, pinned staging and device staging buffers -- the same single-device code
 * path, once per device.  A key is replicated on demand, device to device (mosfhet_hip_bsk_clone / _ksk_clone / _gak_clone: hipMemcpyPeer over xGMI
 * between peers; bit-identical key, so results do not depend on the device count).  No collective anywhere.
 * A device may be listed more than once (two contexts on one GPU): that is how the CPU-box-sized tests exercise this file on a one-GPU machine.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "compat_internal.h"

int g_mc_ndev = 1;
int g_mc_devs[MC_MAX_DEVICES] = {-1};
__thread int t_mc_dev = 0;

static pthread_mutex_t g_rep_lock = PTHREAD_MUTEX_INITIALIZER;
typedef struct { void *primary; void *rep[MC_MAX_DEVICES]; int kind; } Replicas;
#define MAX_REPLICATED 256
static Replicas g_reps[MAX_REPLICATED];

void mosfhet_set_devices(int n, const int *ids) {
  if (mc_engine_started()) {
    fprintf(stderr, "mosfhet_amd: mosfhet_set_devices after the engine was created\n");
    abort();
  }
  if (n < 1 || n > MC_MAX_DEVICES || !ids) {
    fprintf(stderr, "mosfhet_amd: mosfhet_set_devices: 1 .. %d devices\n", MC_MAX_DEVICES);
    abort();
  }
  g_mc_ndev = n;
  for (int i = 0; i < n; i++) g_mc_devs[i] = ids[i];
}

int mosfhet_device_count(void) { return g_mc_ndev; }

void mc_devices_from_env(void) {
  const char *e = getenv("MOSFHET_HIP_DEVICES");
  if (!e || !*e) return;
  int n = 0;
  char *copy = strdup(e), *save = NULL;
  for (char *tok = strtok_r(copy, ",", &save); tok && n < MC_MAX_DEVICES; tok = strtok_r(NULL, ",", &save)) g_mc_devs[n++] = atoi(tok);
  free(copy);
  if (n) g_mc_ndev = n;
}

/* Replication statistics (tests/c/multi_device.c prints them): bytes and seconds per route of mosfhet_hip_last_clone_route */
static double g_rep_seconds[4];
static unsigned long long g_rep_bytes[4];
static int g_rep_keys[4];
void mosfhet_replication_stats(unsigned long long bytes[4], double seconds[4], int keys[4]) {
  pthread_mutex_lock(&g_rep_lock);
  for (int r = 0; r < 4; r++) { bytes[r] = g_rep_bytes[r]; seconds[r] = g_rep_seconds[r]; keys[r] = g_rep_keys[r]; }
  pthread_mutex_unlock(&g_rep_lock);
}

/* the handle of `primary` (kind MC_KEY_*) on device index d, cloned device to device the first time it is asked for (mosfhet_hip_*_clone: hipMemcpyPeer
 * over xGMI where the devices are peers; a seed-compressed table key travels compressed) */
static void *replica_on(void *primary, int kind, int d) {
  if (d == 0 || !primary) return primary;
  pthread_mutex_lock(&g_rep_lock);
  Replicas *slot = NULL;
  for (int i = 0; i < MAX_REPLICATED && !slot; i++)
    if (g_reps[i].primary == primary) slot = &g_reps[i];
  for (int i = 0; i < MAX_REPLICATED && !slot; i++)
    if (!g_reps[i].primary) { slot = &g_reps[i]; memset(slot, 0, sizeof(*slot)); slot->primary = primary; slot->kind = kind; }
  if (!slot) { fprintf(stderr, "mosfhet_amd: too many replicated keys\n"); abort(); }
  if (!slot->rep[d]) {
    mosfhet_hip_ctx_t ctx = mc_ctx_of(d);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t bytes = 0;
    if (kind == MC_KEY_BSK) {
      mosfhet_hip_bsk_t dst = NULL;
      if (mosfhet_hip_bsk_clone(ctx, &dst, (mosfhet_hip_bsk_t)primary)) mc_die("key replication (bootstrap key)");
      bytes = mosfhet_hip_bsk_bytes(dst);
      slot->rep[d] = dst;
    } else if (kind == MC_KEY_KSK) {
      mosfhet_hip_ksk_t dst = NULL;
      if (mosfhet_hip_ksk_clone(ctx, &dst, (mosfhet_hip_ksk_t)primary)) mc_die("key replication (table key)");
      bytes = mosfhet_hip_ksk_bytes(dst);
      slot->rep[d] = dst;
    } else {
      mosfhet_hip_gak_t dst = NULL;
      if (mosfhet_hip_gak_clone(ctx, &dst, (mosfhet_hip_gak_t)primary)) mc_die("key replication (FFT key-switch keys)");
      bytes = mosfhet_hip_gak_bytes(dst);
      slot->rep[d] = dst;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const int route = mosfhet_hip_last_clone_route() & 3;
    g_rep_bytes[route] += bytes;
    g_rep_seconds[route] += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    g_rep_keys[route]++;
  }
  void *r = slot->rep[d];
  pthread_mutex_unlock(&g_rep_lock);
  return r;
}

void *mc_key_here(void *primary, int kind) { return replica_on(primary, kind, t_mc_dev); }

void mc_replicas_free(void *primary) {
  if (!primary) return;
  pthread_mutex_lock(&g_rep_lock);
  for (int i = 0; i < MAX_REPLICATED; i++)
    if (g_reps[i].primary == primary) {
      for (int d = 1; d < MC_MAX_DEVICES; d++)
        if (g_reps[i].rep[d]) {
          if (g_reps[i].kind == MC_KEY_BSK) mosfhet_hip_bsk_destroy((mosfhet_hip_bsk_t)g_reps[i].rep[d]);
          else if (g_reps[i].kind == MC_KEY_KSK) mosfhet_hip_ksk_destroy((mosfhet_hip_ksk_t)g_reps[i].rep[d]);
          else mosfhet_hip_gak_destroy((mosfhet_hip_gak_t)g_reps[i].rep[d]);
        }
      memset(&g_reps[i], 0, sizeof(g_reps[i]));
    }
  pthread_mutex_unlock(&g_rep_lock);
}

typedef struct { mc_slice_fn fn; void *args; int lo, hi; } Job;

/* one persistent host thread per extra device (its staging buffers, streams and device pool live as long as the process): a sharded call hands
 * every worker its slice and runs slice 0 itself */
typedef struct {
  pthread_t th;
  pthread_mutex_t m;
  pthread_cond_t cv;
  Job job;
  int dev, has_job, started;
} Worker;
static Worker g_workers[MC_MAX_DEVICES];
static pthread_mutex_t g_shard_lock = PTHREAD_MUTEX_INITIALIZER;   /* one sharded call at a time; other caller threads queue here */
__thread int t_mc_in_shard = 0;

static void *worker_main(void *pv) {
  Worker *w = (Worker *)pv;
  t_mc_dev = w->dev;
  t_mc_in_shard = 1;
  pthread_mutex_lock(&w->m);
  for (;;) {
    while (!w->has_job) pthread_cond_wait(&w->cv, &w->m);
    Job j = w->job;
    pthread_mutex_unlock(&w->m);
    j.fn(j.args, j.lo, j.hi);
    pthread_mutex_lock(&w->m);
    w->has_job = 0;
    pthread_cond_broadcast(&w->cv);
  }
  return NULL;
}

static void worker_submit(int d, Job j) {
  Worker *w = &g_workers[d];
  if (!w->started) {
    pthread_mutex_init(&w->m, NULL);
    pthread_cond_init(&w->cv, NULL);
    w->dev = d;
    w->has_job = 0;
    w->started = 1;
    if (pthread_create(&w->th, NULL, worker_main, w)) { perror("mosfhet_amd: pthread_create"); abort(); }
    pthread_detach(w->th);
  }
  pthread_mutex_lock(&w->m);
  w->job = j;
  w->has_job = 1;
  pthread_cond_broadcast(&w->cv);
  pthread_mutex_unlock(&w->m);
}

static void worker_wait(int d) {
  Worker *w = &g_workers[d];
  pthread_mutex_lock(&w->m);
  while (w->has_job) pthread_cond_wait(&w->cv, &w->m);
  pthread_mutex_unlock(&w->m);
}

/* run fn(args, lo, hi) over [0, count): one contiguous slice per device (mosfhet_amd/shard.py: shard_bounds), slice 0 on the calling thread (primary
 * device), the others on the devices' worker threads.  keys[] = primary handles, replicated up front from the calling thread. */
void mc_run_sharded(mc_slice_fn fn, void *args, int count, void *const *keys, const int *kinds, int n_keys) {
  const int G = (g_mc_ndev < count) ? g_mc_ndev : (count > 0 ? count : 1);
  if (G <= 1 || t_mc_in_shard) {   /* single device, or already inside a slice */
    fn(args, 0, count);
    return;
  }
  pthread_mutex_lock(&g_shard_lock);
  for (int d = 1; d < G; d++)
    for (int q = 0; q < n_keys; q++) (void)replica_on(keys[q], kinds[q], d);
  mc_use_device();   /* replication switched this thread's current device around */
  const int base = count / G, extra = count % G;
  Job jobs[MC_MAX_DEVICES];
  for (int d = 0; d < G; d++) {
    const int lo = d * base + (d < extra ? d : extra);
    jobs[d].fn = fn; jobs[d].args = args; jobs[d].lo = lo; jobs[d].hi = lo + base + (d < extra ? 1 : 0);
  }
  for (int d = 1; d < G; d++) worker_submit(d, jobs[d]);
  t_mc_in_shard = 1;
  fn(args, jobs[0].lo, jobs[0].hi);
  t_mc_in_shard = 0;
  for (int d = 1; d < G; d++) worker_wait(d);
  pthread_mutex_unlock(&g_shard_lock);
}
