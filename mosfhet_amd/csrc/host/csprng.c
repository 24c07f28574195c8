/*
 * csprng.c -- randomness of the host layer: a ChaCha20 keystream per host thread.
 *
 * Replaces the reference's SHAKE256 / AES-CTR generators seeded from RDSEED (src/misc.c:34-95, src/rnd/, src/sha3/): secret keys, noise and masks drawn
 * by the host layer come from a cryptographic generator keyed from the operating system (getrandom).  mosfhet_seed(seed) replaces that key by one derived
 * from the 64-bit seed: REPRODUCIBLE RUNS FOR TESTS AND BENCHMARKS ONLY -- a 64-bit seed bounds the security of everything generated afterwards.
 * Threads never share generator state: stream t uses nonce t of the process key (the seeding thread keeps nonce 0).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>

#include "compat_internal.h"

#define ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d) \
  a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12); a += b; d ^= a; d = ROTL32(d, 8); c += d; b ^= c; b = ROTL32(b, 7)

/* RFC 8439 block function with a 64-bit block counter and a 64-bit nonce (the original ChaCha20 parameterisation) */
void mc_chacha20_block(uint32_t out[16], const uint32_t key[8], uint64_t counter, uint64_t nonce) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                    (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x[16];
  memcpy(x, s, sizeof(x));
  for (int r = 0; r < 10; r++) {
    QR(x[0], x[4], x[8], x[12]); QR(x[1], x[5], x[9], x[13]); QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
    QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]); QR(x[2], x[7], x[8], x[13]); QR(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

static pthread_mutex_t g_rng_lock = PTHREAD_MUTEX_INITIALIZER;
static uint32_t g_key[8];
static int g_key_set = 0;
static uint64_t g_next_nonce = 1;

static __thread uint32_t t_key[8];
static __thread uint64_t t_nonce, t_counter;
static __thread uint32_t t_block[16];
static __thread int t_pos = 16, t_ready = 0;

static void key_from_os(void) {
  size_t got = 0;
  while (got < sizeof(g_key)) {
    ssize_t r = getrandom((char *)g_key + got, sizeof(g_key) - got, 0);
    if (r <= 0) break;
    got += (size_t)r;
  }
  if (got < sizeof(g_key)) {
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(g_key, sizeof(g_key), 1, f) != 1) {
      fprintf(stderr, "mosfhet_amd: no entropy source (getrandom and /dev/urandom both failed)\n");
      abort();
    }
    fclose(f);
  }
}

static void key_from_seed(uint64_t seed) {
  /* the seed keys one ChaCha20 block whose first half becomes the process key */
  uint32_t k0[8] = {(uint32_t)seed, (uint32_t)(seed >> 32), 0x6d6f7366u, 0x68657421u, 0, 0, 0, 0}, blk[16];
  mc_chacha20_block(blk, k0, 0, 0);
  memcpy(g_key, blk, sizeof(g_key));
}

void mosfhet_seed(uint64_t seed) {
  pthread_mutex_lock(&g_rng_lock);
  key_from_seed(seed);
  g_key_set = 1;
  g_next_nonce = 1;
  memcpy(t_key, g_key, sizeof(t_key));
  pthread_mutex_unlock(&g_rng_lock);
  t_nonce = 0;
  t_counter = 0;
  t_pos = 16;
  t_ready = 1;
  /* the device generators' noise key follows the seed: an engine that is already running gets a fresh secret from the new stream (which also restarts
   * its per-call nonce sequence), exactly what engine start-up does for a seed set before the first use -- so a re-seeded process reproduces the keys
   * of its device generators too, noise included */
  if (mc_engine_started()) {
    uint8_t secret[32];
    mc_rnd_bytes(secret, sizeof(secret));
    if (mosfhet_hip_set_keygen_secret(secret)) mc_die("mosfhet_seed (key-generation secret)");
  }
}

static void thread_start(void) {
  pthread_mutex_lock(&g_rng_lock);
  if (!g_key_set) {
    key_from_os();
    g_key_set = 1;
  }
  memcpy(t_key, g_key, sizeof(t_key));
  t_nonce = g_next_nonce++;
  pthread_mutex_unlock(&g_rng_lock);
  t_counter = 0;
  t_pos = 16;
  t_ready = 1;
}

uint64_t mc_rnd64(void) {
  if (!t_ready) thread_start();
  if (t_pos >= 16) {
    mc_chacha20_block(t_block, t_key, t_counter++, t_nonce);
    t_pos = 0;
  }
  const uint64_t r = (uint64_t)t_block[t_pos] | ((uint64_t)t_block[t_pos + 1] << 32);
  t_pos += 2;
  return r;
}

void mc_rnd_bytes(void *out, size_t bytes) {
  unsigned char *o = (unsigned char *)out;
  while (bytes >= 8) {
    const uint64_t r = mc_rnd64();
    memcpy(o, &r, 8);
    o += 8;
    bytes -= 8;
  }
  if (bytes) {
    const uint64_t r = mc_rnd64();
    memcpy(o, &r, bytes);
  }
}

double mc_rnd_normal(double sigma) { /* Box-Muller, as src/misc.c:87-91 */
  const double u1 = ((double)(mc_rnd64() >> 11) + 0.5) * 0x1p-53, u2 = ((double)(mc_rnd64() >> 11) + 0.5) * 0x1p-53;
  return cos(2. * M_PI * u1) * sqrt(-2. * log(u2)) * sigma;
}

/* the reference's public randomness entry points (src/misc.c:79-95), used by its applications */
void generate_random_bytes(uint64_t amount, uint8_t *pointer) { mc_rnd_bytes(pointer, (size_t)amount); }
double generate_normal_random(double sigma) { return mc_rnd_normal(sigma); }
void generate_torus_normal_random_array(Torus *out, double sigma, int N) {
  for (int i = 0; i < N; i++) out[i] = double2torus(mc_rnd_normal(sigma));
}
void generate_rnd_seed(uint64_t *p) { mc_rnd_bytes(p, 32); }   /* src/misc.c:34-49: four words */
