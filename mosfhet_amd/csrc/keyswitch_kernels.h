// keyswitch_kernels.h -- batched LWE -> LWE key switch by table lookup (gfx950).
//
// Replaces tlwe_keyswitch [src/tlwe.c:289-303] for a batch:
//   out = (0, ..., 0, in.b);  for i < n_in, j < t:  v = ((in.a[i] + 2^(63 - t bb)) >> (64 - (j+1) bb)) & (2^bb - 1);
//   if v != 0:  out -= KS[i][j][v-1]            (rows of n_out + 1 Torus words, exact mod 2^64)
//
// Pure integer and gather-bound: un-batched, every ciphertext touches up to n_in*t rows (24 MB at SET_1, 83 MB at
// lvl2) of a 72 MB / 1.2 GB table.  The batch is processed so that the table is streamed ONCE per tile of
// ciphertexts instead of once per ciphertext:
//   * ciphertexts sit on the LANES (a wavefront owns 64 ciphertexts), output words in registers: a wavefront
//     accumulates a slice of W output words for its 64 ciphertexts (2 W VGPRs);
//   * a workgroup of NW wavefronts (NW*64 ciphertexts) shares one slice; for a block of key digits (i, j0..j0+JB) it
//     stages the 2^bb - 1 candidate rows of that slice into LDS with coalesced loads -- plus an all-zero row for
//     digit 0, so there is no branch -- and every lane then reads ITS row (ds_read_b128, row stride W+2 words:
//     distinct digits land on distinct bank groups, equal digits broadcast) and subtracts;
//   * the batch is transposed on the way in and out (transpose_u64_kernel) so that both the per-lane input words
//     and the result words are coalesced.
// Table traffic per launch: (ciphertexts / (64 NW)) * table size, served from L2 / Infinity Cache.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keygen_kernels.h"

namespace mosfhet {

// out[c][r] = sum_{s < parts} in[s * part_stride + r * ldin + c] for an R x C matrix of 64-bit words (leading
// dimensions ldin / ldout; parts = 1: plain transpose); out rows c >= C and columns r >= R are not touched.
// 32 x 32 tiles through LDS, block = (32, 8).
__global__ __launch_bounds__(256) void transpose_u64_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int R, int C,
                                                            size_t ldin, size_t ldout, int parts, size_t part_stride) {
  __shared__ uint64_t tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = threadIdx.y; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + threadIdx.x;
    uint64_t v = 0;
    if (r < R && c < C)
      for (int s = 0; s < parts; s++) v += in[(size_t)s * part_stride + (size_t)r * ldin + c];
    tile[k][threadIdx.x] = v;
  }
  workgroup_sync();
  for (int k = threadIdx.y; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + threadIdx.x;
    if (c < C && r < R) out[(size_t)c * ldout + r] = tile[threadIdx.x][k];
  }
}

typedef unsigned long long ul2 __attribute__((ext_vector_type(2)));

constexpr int KS_W = 32;         // output words per wavefront slice
constexpr int KS_NW = 4;         // wavefronts (x 64 ciphertexts) per workgroup
constexpr int KS_LDS_BYTES = 16384;

constexpr int KS_PF_MAX = 16;    // upper bound of staged row elements a thread prefetches per stage

// inT: [n_in + 1][Bp] (word i of ciphertext c at inT[i * Bp + c]; Bp a multiple of 64 * NW, padding zero-filled)
// outT: [row][Bp]; rows have `row` words and the input's b word is added into word `b_word`
// (LWE -> LWE: row = n_out + 1, b_word = n_out; LWE -> TRLWE packing: row = 2N, b_word = N, src/keyswitch.c:458-475)
// Stages (i, j0..j0+JB) are software-pipelined: while a stage is consumed from one LDS buffer, the rows of the next
// stage are already in flight into registers and are written to the other buffer afterwards -- one barrier per stage.
// PF = ceil(JB * cands / (64 NW / W)): candidate-row words each thread moves per stage (compile time: no predicated
// load chains).  Requires t % JB == 0 or handles the short last block of an i by re-staging valid rows only.
// MASKGEN (seed-compressed TRLWE table keys; the reference: trlwe_compressed_subto, src/trlwe_compressed_vaes.c:139-160): ksk holds
// only the b halves, [rows][row / 2]; workgroups whose slice lies in the mask half regenerate their candidate rows from (seed, row, word) with
// the generator that made the key (keygen_mix) instead of loading them -- half the key bytes in HBM, no loads at all for half of the grid.
template <int W, int NW, int PF, bool MASKGEN = false>
__global__ __launch_bounds__(64 * NW) void tlwe_keyswitch_kernel(const uint64_t *__restrict__ ksk, const uint64_t *__restrict__ inT,
                                                                uint64_t *__restrict__ outT, size_t Bp, int n_in, int row, int b_word, int t,
                                                                int base_bit, int JB, int i_per_split, uint64_t seed, int mask_words) {
  extern __shared__ __attribute__((aligned(16))) uint64_t rows[];  // [2][JB][cands + 1][W + 2]
  constexpr int RS = W + 2;                                        // row stride in words (16-byte aligned, bank-skewed)
  constexpr int SV_STEP = (64 * NW) / W;
  const int tid = threadIdx.x;
  const int cands = (1 << base_bit) - 1;
  const int w0 = blockIdx.x * W;                                   // first output word of this slice
  const size_t ct = (size_t)blockIdx.y * (64 * NW) + tid;          // this lane's ciphertext (column of inT / outT)
  const uint64_t round_off = 1ull << (63 - base_bit * t);
  const uint32_t mask = (1u << base_bit) - 1;
  const size_t buf_words = (size_t)JB * (cands + 1) * RS;

  uint64_t acc[W];
#pragma unroll
  for (int w = 0; w < W; w++) acc[w] = 0;
  // blockIdx.z splits the mask words i over several workgroups; partial sums go to outT[z] and are added up by the
  // transposing epilogue.  Only split 0 carries the b word.
  const int i_begin = blockIdx.z * i_per_split, i_end = (i_begin + i_per_split < n_in) ? i_begin + i_per_split : n_in;
  outT += (size_t)blockIdx.z * (size_t)row * Bp;
  if (blockIdx.z == 0 && b_word >= w0 && b_word < w0 + W) {
    const uint64_t b = inT[(size_t)n_in * Bp + ct];
#pragma unroll
    for (int w = 0; w < W; w++)
      if (w0 + w == b_word) acc[w] = b;
  }
  // zero rows (digit 0) of every j slot of both buffers, once
  for (int k = tid; k < 2 * JB * RS; k += 64 * NW) {
    const int bj = k / RS;  // buffer * JB + jj
    rows[(size_t)(bj / JB) * buf_words + (size_t)(bj % JB) * (cands + 1) * RS + (k % RS)] = 0;
  }

  // staging map: this thread moves word sw of candidate rows vv = sv0 + k * SV_STEP (k < PF) of a stage;
  // LDS slot of candidate row vv = jj * cands + v is (jj * (cands + 1) + v + 1) * RS.  Rows past the end of the
  // stage (vv >= jb * cands) are clamped to the last valid row for the load and simply not written.
  const int sw = (w0 + tid % W < row) ? tid % W : 0, sv0 = tid / W;
  int slot[PF];
  size_t goff[PF];
#pragma unroll
  for (int k = 0; k < PF; k++) {
    const int vv = sv0 + k * SV_STEP, jj = vv / cands, v = vv - jj * cands;
    slot[k] = (jj * (cands + 1) + v + 1) * RS + tid % W;
    goff[k] = (size_t)vv * row;
  }
  const bool in_row = (w0 + tid % W) < row;

  uint64_t pf[PF];
  const size_t i_stride = (size_t)t * cands * row;
  const uint64_t *__restrict__ kbase = ksk + w0 + sw;
  // MASKGEN: candidate row vv of stage (i_, j_) is row ((i_ t + j_) cands + vv) of the key; words below mask_words are regenerated, the others
  // come from the stored b part, [rows][row - mask_words] (TRLWE rows: N of 2N words each; LWE rows: the single b word)
  const int word = w0 + sw, b_stride = row - mask_words;
  auto fetch_compressed = [&](int i_, int j_, int k, int last) -> uint64_t {
    const int vv = sv0 + k * SV_STEP;
    const size_t r = ((size_t)i_ * t + j_) * cands + (vv <= last ? vv : last);
    if (word < mask_words) return keygen_mix(seed, r, (uint64_t)word, 0);
    return ksk[r * (size_t)b_stride + (size_t)(word - mask_words)];
  };
  // prefetch stage (i = i_begin, j0 = 0)
  {
    const int last = ((t < JB ? t : JB) * cands - 1);
#pragma unroll
    for (int k = 0; k < PF; k++) {
      if constexpr (MASKGEN) pf[k] = fetch_compressed(i_begin, 0, k, last);
      else pf[k] = kbase[(size_t)i_begin * i_stride + ((sv0 + k * SV_STEP <= last) ? goff[k] : (size_t)last * row)];
    }
  }
  uint64_t a_next = inT[(size_t)i_begin * Bp + ct] + round_off;
  int buf_sel = 0;
  for (int i = i_begin; i < i_end; i++) {
    const uint64_t a = a_next;
    if (i + 1 < i_end) a_next = inT[(size_t)(i + 1) * Bp + ct] + round_off;
    for (int j0 = 0; j0 < t; j0 += JB) {
      const int jb = (t - j0 < JB) ? (t - j0) : JB;
      uint64_t *buf = rows + (size_t)buf_sel * buf_words;
      buf_sel ^= 1;
#pragma unroll
      for (int k = 0; k < PF; k++)
        if (in_row && sv0 + k * SV_STEP < jb * cands) buf[slot[k]] = 0 - pf[k];   // rows are staged NEGATED: the inner op is a 64-bit add
      workgroup_sync();  // stage data visible; the other buffer is free (its readers passed the previous barrier)
      // next stage: (i, j0 + JB) or (i + 1, 0)
      {
        int ni = i, nj = j0 + JB;
        if (nj >= t) { ni = i + 1; nj = 0; }
        if (ni < i_end) {
          const int last = (((t - nj < JB) ? (t - nj) : JB) * cands - 1);
          const uint64_t *__restrict__ src = kbase + (size_t)ni * i_stride + (size_t)nj * cands * row;
#pragma unroll
          for (int k = 0; k < PF; k++) {
            if constexpr (MASKGEN) pf[k] = fetch_compressed(ni, nj, k, last);
            else pf[k] = src[(sv0 + k * SV_STEP <= last) ? goff[k] : (size_t)last * row];
          }
        }
      }
      for (int jj = 0; jj < jb; jj++) {
        const uint32_t v = (uint32_t)(a >> (64 - (j0 + jj + 1) * base_bit)) & mask;
        const ul2 *r = reinterpret_cast<const ul2 *>(buf + (size_t)(jj * (cands + 1) + v) * RS);  // 16-byte aligned rows
#pragma unroll
        for (int w = 0; w < W; w += 2) {
          const ul2 x = r[w / 2];
          acc[w] += x.x;       // one v_lshl_add_u64 instead of a borrow pair (out -= row of src/tlwe.c:297, src/keyswitch.c:472)
          acc[w + 1] += x.y;
        }
      }
    }
  }
#pragma unroll
  for (int w = 0; w < W; w++)
    if (w0 + w < row) outT[(size_t)(w0 + w) * Bp + ct] = acc[w];
}

// ---- small batched TLWE / TRLWE glue kernels (all exact mod 2^64) ----
// tlwe_addto over flat batches [src/tlwe.c:170-173]: out[i] += in[i]
__global__ void words_addto_kernel(uint64_t *__restrict__ out, const uint64_t *__restrict__ in, size_t words) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) out[i] += in[i];
}

// ct[b].b += delta for every sample of a batch (row = n + 1 words)  [src/bootstrap.c:530: ct_sign->b -= sign]
__global__ void tlwe_add_to_b_kernel(uint64_t *__restrict__ ct, int count, size_t row, uint64_t delta) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < count) ct[(size_t)b * row + row - 1] += delta;
}

// the same with the samples `stride` words apart and the b word at `word` (samples interleaved with others: capi_ext.inc, the KS21 level layout)
__global__ void tlwe_add_to_word_kernel(uint64_t *__restrict__ ct, int count, size_t stride, size_t word, uint64_t delta) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < count) ct[(size_t)b * stride + word] += delta;
}

// fill a trivial TRLWE whose b polynomial is the constant `value` (trlwe_torus_packing with one slot, src/trlwe.c:662-667)
__global__ void trlwe_constant_kernel(uint64_t *__restrict__ tv, int N, uint64_t value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { tv[i] = 0; tv[N + i] = value; }
}

// trlwe_extract_tlwe at coefficient idx for a batch (k = 1) [src/trlwe.c:540-552]; in stride / out stride in words
__global__ void trlwe_extract_kernel(uint64_t *__restrict__ out, size_t out_stride, const uint64_t *__restrict__ in, size_t in_stride,
                                     int N, int idx) {
  const uint64_t *c = in + (size_t)blockIdx.y * in_stride;
  uint64_t *o = out + (size_t)blockIdx.y * out_stride;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < N) o[j] = (j <= idx) ? c[idx - j] : (0 - c[N + idx - j]);
  if (j == 0) o[N] = c[N + idx];
}

// the same for k >= 1 mask polynomials [src/trlwe.c:540-552 loops over i < k]: in = [k+1][N] words, out = [kN+1]
__global__ void trlwe_extract_k_kernel(uint64_t *__restrict__ out, size_t out_stride, const uint64_t *__restrict__ in, size_t in_stride,
                                       int N, int k, int idx) {
  const uint64_t *c = in + (size_t)blockIdx.y * in_stride;
  uint64_t *o = out + (size_t)blockIdx.y * out_stride;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < k * N) {
    const int i = x / N, j = x - i * N;
    const uint64_t *a = c + (size_t)i * N;
    o[x] = (j <= idx) ? a[idx - j] : (0 - a[N + idx - j]);
  }
  if (x == 0) o[(size_t)k * N] = c[(size_t)k * N + idx];
}

struct KsWorkspace {
  uint64_t *inT = nullptr, *outT = nullptr;
  size_t words_in = 0, words_out = 0;
};

// Returns hipSuccess or the failing error.  ws is grown on demand (kept by the key handle between calls).
// in: rows of n_in + 1 words (n_in when b_word < 0) spaced in_stride words apart; out: rows of `row` words spaced out_stride apart.
template <int NW>
inline hipError_t launch_tlwe_keyswitch_nw(const uint64_t *ksk, uint64_t *out, size_t out_stride, const uint64_t *in, size_t in_stride, int count,
                                           int n_in, int row, int b_word, int t, int base_bit, KsWorkspace &ws, hipStream_t s, bool compressed,
                                           uint64_t seed, int mask_words) {
  constexpr int W = KS_W, TILE = 64 * NW;
  const size_t Bp = ((size_t)count + TILE - 1) / TILE * TILE;
  // split the mask words over blockIdx.z until the grid fills the chip (each workgroup walks its i-range serially)
  const int slices = (row + W - 1) / W, ct_blocks = (int)(Bp / TILE);
  // ~4 x the resident capacity (6 workgroups per CU): the workgroups are not equally long (tile padding, split remainders) and the
  // kernel is latency-bound (LDS ~42 %, VALU ~50 %, HBM ~34 % busy, profiles/r01d_ks_summary.txt), so a finer grid balances better:
  // packing switch 7.4 -> 6.3 ms, lvl2 LWE switch 6.1 -> 5.1 ms per batch.
  constexpr int target_wgs = 6144;
  int split = (target_wgs + slices * ct_blocks - 1) / (slices * ct_blocks);
  // one or two tiles of ciphertexts (one GPU's share of a sharded batch: 128 inputs) leave each workgroup a long serial walk -- 128 input words x 4 stages at 16 splits,
  // ~1 ms whatever the batch -- so few tiles are cut finer (64 splits: 128 lvl2 switches 1.02 -> 0.4 ms; the partial sums are integers mod 2^64: same bits)
  const int max_split = ct_blocks <= 2 ? 64 : 16;
  if (split > max_split) split = max_split;
  if (split > n_in / 8) split = n_in / 8;
  if (split < 1) split = 1;
  const int i_per_split = (n_in + split - 1) / split;
  split = (n_in + i_per_split - 1) / i_per_split;
  const int in_words = n_in + (b_word >= 0 ? 1 : 0);  // b_word < 0: every input word is a digit source, none is copied (private key switch)
  const size_t need_in = (size_t)in_words * Bp, need_out = (size_t)split * row * Bp;
  hipError_t e;
  if (ws.words_in < need_in) {
    if (ws.inT) (void)hipFree(ws.inT);
    if ((e = hipMalloc((void **)&ws.inT, need_in * sizeof(uint64_t))) != hipSuccess) return e;
    ws.words_in = need_in;
  }
  if (ws.words_out < need_out) {
    if (ws.outT) (void)hipFree(ws.outT);
    if ((e = hipMalloc((void **)&ws.outT, need_out * sizeof(uint64_t))) != hipSuccess) return e;
    ws.words_out = need_out;
  }
  if (Bp != (size_t)count && (e = hipMemsetAsync(ws.inT, 0, need_in * sizeof(uint64_t), s)) != hipSuccess) return e;
  // in[count][n_in + 1] -> inT[n_in + 1][Bp]
  hipLaunchKernelGGL(transpose_u64_kernel, dim3((in_words + 31) / 32, (count + 31) / 32), dim3(32, 8), 0, s, in, ws.inT, count, in_words,
                     in_stride, Bp, 1, (size_t)0);
  const int cands = (1 << base_bit) - 1;
  // LDS per buffer = digit positions per stage (one barrier per stage).  Small digit sets (base_bit 2) take all of an input word's positions in
  // one stage; base_bit 3 and 4 take TWO positions per stage (same-box sweep with the 512-ciphertext tile: packing switch 5.00 -> 4.40 ms per 1024
  // and 18.9 -> 16.9 ms per 4096, lvl2 LWE switch 4.06 -> 3.54 ms; three positions are slower again); wider digit sets one.  A table whose
  // every word is regenerated (seed-compressed LWE key) keeps one position: its stage is generator-bound, 4.02 vs 4.11 ms.
  const int two_positions = 2 * (cands + 1) * (W + 2) * 8;
  const bool all_generated = compressed && mask_words == row - 1;
  const int lds_budget = (cands < 7 ? KS_LDS_BYTES : ((cands <= 15 && !all_generated) ? two_positions : 4352));
  int JB = lds_budget / ((cands + 1) * (W + 2) * 8);          // per LDS buffer
  // rows a stage can prefetch through registers: 8 per thread, 16 for the widest digit sets (base_bit 8 with a 256-ciphertext tile, where
  // one digit position alone has more candidate rows than 8 per thread cover)
  const int pf_regs = cands > 8 * (TILE / W) ? KS_PF_MAX : 8;
  const int pf_cap = pf_regs * (TILE / W) / cands;
  if (JB > pf_cap) JB = pf_cap;
  if (JB < 1) JB = 1;
  if (JB > t) JB = t;
  if (cands > KS_PF_MAX * (TILE / W)) return hipErrorInvalidValue;  // base_bit too large for the staging registers (cannot happen for base_bit <= 8)
  const size_t lds = 2 * (size_t)JB * (cands + 1) * (W + 2) * 8;
  const int pf = (JB * cands + TILE / W - 1) / (TILE / W);
  const dim3 grid(slices, ct_blocks, split);
  // stages of base_bit 7 / 8 need 70 / 139 KiB of the CU's 160 KiB: beyond the 64 KiB a kernel gets without asking
#define KS_LAUNCH_MG(PF, MG)                                                                                                                   \
  do {                                                                                                                                         \
    if (lds > 65536 && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tlwe_keyswitch_kernel<W, NW, PF, MG>),                        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)                          \
      return e;                                                                                                                                \
    hipLaunchKernelGGL((tlwe_keyswitch_kernel<W, NW, PF, MG>), grid, dim3(TILE), lds, s, ksk, ws.inT, ws.outT, Bp, n_in, row, b_word, t, base_bit, \
                       JB, i_per_split, seed, mask_words);                                                                                     \
  } while (0)
#define KS_LAUNCH(PF)                 \
  do {                                \
    if (compressed) KS_LAUNCH_MG(PF, true); \
    else KS_LAUNCH_MG(PF, false);     \
  } while (0)
  switch (pf) {
    case 1: KS_LAUNCH(1); break;
    case 2: KS_LAUNCH(2); break;
    case 3: KS_LAUNCH(3); break;
    case 4: KS_LAUNCH(4); break;
    case 5: KS_LAUNCH(5); break;
    case 6: KS_LAUNCH(6); break;
    case 7: KS_LAUNCH(7); break;
    case 8: KS_LAUNCH(8); break;
    case 9: case 10: KS_LAUNCH(10); break;
    case 11: case 12: KS_LAUNCH(12); break;
    case 13: case 14: KS_LAUNCH(14); break;
    default: KS_LAUNCH(16); break;
  }
#undef KS_LAUNCH
#undef KS_LAUNCH_MG
  // outT[row][Bp] -> out[count][row]
  hipLaunchKernelGGL(transpose_u64_kernel, dim3((count + 31) / 32, (row + 31) / 32), dim3(32, 8), 0, s, ws.outT, out, row, count, Bp, out_stride,
                     split, (size_t)row * Bp);
  return hipGetLastError();
}

// ---- few ciphertexts (single calls of the legacy API, small batches): the tiled kernel above would walk the whole table for a tile with one live lane
// (1.7 ms for one lvl2-sized switch).  Here a ciphertext's switch is spread over the chip instead: workgroup (slice, split, ct) owns 256 output words and
// a range of the input words, reads ONLY the rows the digits select (coalesced, 2 KiB per row and workgroup) and leaves a partial sum; a second small
// kernel adds the partial sums and the b word.  62 MB of rows per lvl2-sized switch instead of the 186 MB table, on ~1000 workgroups instead of one lane.
template <bool MASKGEN>
__global__ __launch_bounds__(256) void tlwe_keyswitch_small_kernel(const uint64_t *__restrict__ ksk, const uint64_t *__restrict__ in, size_t in_stride,
                                                                  uint64_t *__restrict__ part, int count, int n_in, int row, int t, int base_bit, int i_per_split,
                                                                  uint64_t seed, int mask_words) {
  const int w = blockIdx.x * 256 + threadIdx.x, split = blockIdx.y, ct = blockIdx.z;
  const int cands = (1 << base_bit) - 1;
  const uint32_t mask = (1u << base_bit) - 1;
  const uint64_t round_off = 1ull << (63 - base_bit * t);
  const int i_begin = split * i_per_split, i_end = (i_begin + i_per_split < n_in) ? i_begin + i_per_split : n_in;
  const uint64_t *__restrict__ a_words = in + (size_t)ct * in_stride;
  const int b_stride = row - mask_words;
  uint64_t acc = 0;
  if (w < row)
    for (int i = i_begin; i < i_end; i++) {
      const uint64_t a = a_words[i] + round_off;
      for (int j = 0; j < t; j++) {
        const uint32_t v = (uint32_t)(a >> (64 - (j + 1) * base_bit)) & mask;
        if (!v) continue;
        const size_t r = ((size_t)i * t + j) * cands + (v - 1);
        if constexpr (MASKGEN) acc -= w < mask_words ? keygen_mix(seed, r, (uint64_t)w, 0) : ksk[r * (size_t)b_stride + (size_t)(w - mask_words)];
        else acc -= ksk[r * (size_t)row + w];
      }
    }
  if (w < row) part[((size_t)split * count + ct) * row + w] = acc;
}

__global__ __launch_bounds__(256) void tlwe_keyswitch_small_reduce_kernel(uint64_t *__restrict__ out, size_t out_stride, const uint64_t *__restrict__ part,
                                                                         const uint64_t *__restrict__ in, size_t in_stride, int count, int n_in, int row, int b_word,
                                                                         int splits) {
  const int w = blockIdx.x * 256 + threadIdx.x, ct = blockIdx.y;
  if (w >= row) return;
  uint64_t acc = (w == b_word) ? in[(size_t)ct * in_stride + n_in] : 0;
  for (int sp = 0; sp < splits; sp++) acc += part[((size_t)sp * count + ct) * row + w];
  out[(size_t)ct * out_stride + w] = acc;
}

inline hipError_t launch_tlwe_keyswitch_small(const uint64_t *ksk, uint64_t *out, size_t out_stride, const uint64_t *in, size_t in_stride, int count, int n_in, int row,
                                              int b_word, int t, int base_bit, KsWorkspace &ws, hipStream_t s, bool compressed, uint64_t seed, int mask_words) {
  const int slices = (row + 255) / 256;
  int splits = (1024 + slices * count - 1) / (slices * count);
  if (splits > n_in / 4) splits = n_in / 4;
  if (splits < 1) splits = 1;
  const int i_per_split = (n_in + splits - 1) / splits;
  splits = (n_in + i_per_split - 1) / i_per_split;
  const size_t need_out = (size_t)splits * count * row;
  hipError_t e;
  if (ws.words_out < need_out) {
    if (ws.outT) (void)hipFree(ws.outT);
    if ((e = hipMalloc((void **)&ws.outT, need_out * sizeof(uint64_t))) != hipSuccess) return e;
    ws.words_out = need_out;
  }
  const dim3 grid(slices, splits, count);
  if (compressed)
    hipLaunchKernelGGL(tlwe_keyswitch_small_kernel<true>, grid, dim3(256), 0, s, ksk, in, in_stride, ws.outT, count, n_in, row, t, base_bit, i_per_split, seed, mask_words);
  else
    hipLaunchKernelGGL(tlwe_keyswitch_small_kernel<false>, grid, dim3(256), 0, s, ksk, in, in_stride, ws.outT, count, n_in, row, t, base_bit, i_per_split, seed, mask_words);
  hipLaunchKernelGGL(tlwe_keyswitch_small_reduce_kernel, dim3(slices, count), dim3(256), 0, s, out, out_stride, ws.outT, in, in_stride, count, n_in, row, b_word, splits);
  return hipGetLastError();
}

// tlwe_keyswitch_no_precomp (src/tlwe.c:305-320): the key holds ONE sample per (input word, digit position), TLWE(s_i 2^(64 - (j+1) bb)), and the digit
// multiplies it:  out = (0, b) - sum_{i,j} digit_ij KS[i][j].  The reference adds the rounding offset 2^(63 - t bb) twice before it cuts the digits (:314,316);
// so does this.  Same work split as tlwe_keyswitch_small_kernel; partial sums finished by tlwe_keyswitch_small_reduce_kernel.
__global__ __launch_bounds__(256) void tlwe_keyswitch_scaled_kernel(const uint64_t *__restrict__ ksk, const uint64_t *__restrict__ in, size_t in_stride,
                                                                   uint64_t *__restrict__ part, int count, int n_in, int row, int t, int base_bit, int i_per_split) {
  const int w = blockIdx.x * 256 + threadIdx.x, split = blockIdx.y, ct = blockIdx.z;
  const uint32_t mask = (1u << base_bit) - 1;
  const uint64_t round_off = 2 * (1ull << (63 - base_bit * t));
  const int i_begin = split * i_per_split, i_end = (i_begin + i_per_split < n_in) ? i_begin + i_per_split : n_in;
  const uint64_t *__restrict__ a_words = in + (size_t)ct * in_stride;
  uint64_t acc = 0;
  if (w < row)
    for (int i = i_begin; i < i_end; i++) {
      const uint64_t a = a_words[i] + round_off;
      for (int j = 0; j < t; j++) {
        const uint64_t v = (uint32_t)(a >> (64 - (j + 1) * base_bit)) & mask;
        acc -= v * ksk[((size_t)i * t + j) * (size_t)row + w];
      }
    }
  if (w < row) part[((size_t)split * count + ct) * row + w] = acc;
}

inline hipError_t launch_tlwe_keyswitch_scaled(const uint64_t *ksk, uint64_t *out, size_t out_stride, const uint64_t *in, size_t in_stride, int count, int n_in, int row,
                                               int t, int base_bit, KsWorkspace &ws, hipStream_t s) {
  const int slices = (row + 255) / 256;
  int splits = (1024 + slices * count - 1) / (slices * count);
  if (splits > n_in / 4) splits = n_in / 4;
  if (splits < 1) splits = 1;
  const int i_per_split = (n_in + splits - 1) / splits;
  splits = (n_in + i_per_split - 1) / i_per_split;
  const size_t need_out = (size_t)splits * count * row;
  hipError_t e;
  if (ws.words_out < need_out) {
    if (ws.outT) (void)hipFree(ws.outT);
    if ((e = hipMalloc((void **)&ws.outT, need_out * sizeof(uint64_t))) != hipSuccess) return e;
    ws.words_out = need_out;
  }
  hipLaunchKernelGGL(tlwe_keyswitch_scaled_kernel, dim3(slices, splits, count), dim3(256), 0, s, ksk, in, in_stride, ws.outT, count, n_in, row, t, base_bit, i_per_split);
  hipLaunchKernelGGL(tlwe_keyswitch_small_reduce_kernel, dim3(slices, count), dim3(256), 0, s, out, out_stride, ws.outT, in, in_stride, count, n_in, row, row - 1, splits);
  return hipGetLastError();
}

}  // namespace mosfhet
#include "keyswitch_words_kernels.h"   // the other orientation (output words on the lanes): launch_tlwe_keyswitch_words
namespace mosfhet {

// Tile of 256 ciphertexts per workgroup for small digit sets (the table is cache resident), 512 for base_bit >= 3, where the
// multi-gigabyte table is re-read once per tile (packing switch 5.3 -> 4.9 ms, lvl2 LWE switch 4.45 -> 4.16 ms; SET_1 prefers 256).
inline hipError_t launch_tlwe_keyswitch(const uint64_t *ksk, uint64_t *out, size_t out_stride, const uint64_t *in, size_t in_stride, int count,
                                        int n_in, int row, int b_word, int t, int base_bit, KsWorkspace &ws, hipStream_t s, bool compressed = false,
                                        uint64_t seed = 0) {
  // compressed keys: TRLWE rows (b_word = N or none) keep their b polynomial, LWE rows (b_word = row - 1) their one b word
  const int mask_words = !compressed ? 0 : (b_word == row - 1 ? row - 1 : row / 2);
  // up to this many ciphertexts take the direct form (MOSFHET_KS_SMALL_MAX overrides, 0 disables): beyond it the tiled kernel's one pass over the table wins
  static const int small_max = getenv("MOSFHET_KS_SMALL_MAX") ? atoi(getenv("MOSFHET_KS_SMALL_MAX")) : 16;
  if (count <= small_max)
    return launch_tlwe_keyswitch_small(ksk, out, out_stride, in, in_stride, count, n_in, row, b_word, t, base_bit, ws, s, compressed, seed, mask_words);
  // digit sets of at most 15 candidates: output words on the lanes, wave-uniform digits (keyswitch_words_kernels.h; MOSFHET_HIP_KS_WORDS / mosfhet_hip_set_ks_words: 0 = the LDS-gather tiles below)
  if (ks_words_applies(count, n_in, row, t, base_bit, compressed, mask_words))
    return launch_tlwe_keyswitch_words(ksk, out, out_stride, in, in_stride, count, n_in, row, b_word, t, base_bit, ws, s, compressed, seed, mask_words);
  // (up to 256 ciphertexts one 256-wide tile holds them all: the 512-wide tile's extra wavefronts would only stage rows -- +15 % on circuit bootstraps of
  // 64 - 256 ciphertexts, the shape of a batch of 1024 split over 8 GPUs)
  if (base_bit >= 3 && (count > 256 || base_bit > 4))   // (wider digits need the 512-thread tile's staging width)
    return launch_tlwe_keyswitch_nw<8>(ksk, out, out_stride, in, in_stride, count, n_in, row, b_word, t, base_bit, ws, s, compressed, seed, mask_words);
  return launch_tlwe_keyswitch_nw<KS_NW>(ksk, out, out_stride, in, in_stride, count, n_in, row, b_word, t, base_bit, ws, s, compressed, seed, mask_words);
}

}  // namespace mosfhet
