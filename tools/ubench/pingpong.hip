// pingpong.hip -- latency of a hand-off between two workgroups through global memory (flag + 16 KiB payload), same XCD and different XCDs:
// the cost a bootstrap spread over several workgroups would pay per CMUX step.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pingpong tools/ubench/pingpong.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// grid workgroups of 128 threads; workgroups A and B take turns: the writer stores `words` payload words, then releases a flag; the reader spins on the flag,
// then reads the payload.  rounds round trips.
__global__ void k_pingpong(uint64_t *payload, unsigned *flag, int a, int b, int rounds, int words, uint64_t *sink) {
  const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (me < 0) return;
  uint64_t acc = 0;
  for (int r = 0; r < rounds; r++) {
    const unsigned turn = 2u * r + 1;
    if (me == 0) {
      for (int i = threadIdx.x; i < words; i += blockDim.x) payload[i] = acc + i + r;
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flag, turn, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != turn + 1) {}
      __syncthreads();
      for (int i = threadIdx.x; i < words; i += blockDim.x) acc += __builtin_nontemporal_load(payload + words + i);
    } else {
      if (threadIdx.x == 0) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != turn) {}
      __syncthreads();
      for (int i = threadIdx.x; i < words; i += blockDim.x) acc += __builtin_nontemporal_load(payload + i);
      for (int i = threadIdx.x; i < words; i += blockDim.x) payload[words + i] = acc + i;
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flag, turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  uint64_t *payload, *sink;
  unsigned *flag;
  CHECK(hipMalloc(&payload, 2 * 4096 * 8));
  CHECK(hipMalloc(&sink, 64 * 128 * 8));
  CHECK(hipMalloc(&flag, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int rounds = 20000;
  for (int words : {0, 2048, 4096})
    for (int b : {8, 1, 4}) {
      CHECK(hipMemset(flag, 0, 4));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(128), 0, nullptr, payload, flag, 0, b, rounds, words, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("workgroups 0 and %d (%s), payload %5d bytes each way: %.2f us per round trip (%.2f us per hand-off)\n", b, b % 8 == 0 ? "same XCD if round-robin" : "different XCDs",
             words * 8, ms * 1e3 / rounds, ms * 1e3 / rounds / 2);
    }
  return 0;
}
