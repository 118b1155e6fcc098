// What does a grid-wide seam cost on this chip?  (a) back-to-back dependent launches of a tiny kernel on one stream
// (launch-to-launch period = what every one of the generator's ~230 layer seams pays today) against (b) an in-kernel
// grid barrier between the same phases (one agent-scope atomic counter, every block spins with s_sleep): the price
// a persistent fused kernel would pay instead.  Each phase touches `bytes` of memory per block so that the seam
// carries a real release/acquire of data across the eight XCDs' L2s.  Development aid for DESIGN.md section 9.
//   hipcc --offload-arch=gfx950 -O3 -o seam_cost tools/seam_cost.hip && ./seam_cost
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int words_per_block, int phase) {
  float* p = buf + (size_t)blockIdx.x * words_per_block;
  // read what the NEIGHBOUR block wrote in the previous phase, write our own slab
  const float* q = buf + (size_t)((blockIdx.x + 97) % gridDim.x) * words_per_block;
  for (int i = threadIdx.x; i < words_per_block; i += 256) p[i] = q[i] * 0.5f + (float)phase;
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                     // release our slab (L2 write-back at agent scope)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();                                     // acquire the others' slabs
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void fused_kernel(float* buf, int words_per_block, int phases, unsigned* counter) {
  float* p = buf + (size_t)blockIdx.x * words_per_block;
  const float* q = buf + (size_t)((blockIdx.x + 97) % gridDim.x) * words_per_block;
  for (int ph = 0; ph < phases; ++ph) {
    for (int i = threadIdx.x; i < words_per_block; i += 256) p[i] = __builtin_nontemporal_load(q + i) * 0.5f + (float)ph;
    grid_barrier(counter, (unsigned)(ph + 1) * gridDim.x);
  }
}

int main() {
  const int blocks = 256, phases = 200;
  for (int words : {64, 4096, 16384}) {                  // 256 B, 16 KiB, 64 KiB per block and phase
    float* d;
    unsigned* c;
    (void)hipMalloc(&d, (size_t)blocks * words * 4);
    (void)hipMalloc(&c, 4);
    (void)hipMemset(d, 0, (size_t)blocks * words * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms_l = 0.f, ms_f = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(blocks), dim3(256), 0, 0, d, words, ph);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms_l, e0, e1);
      (void)hipMemset(c, 0, 4);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(fused_kernel, dim3(blocks), dim3(256), 0, 0, d, words, phases, c);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms_f, e0, e1);
    }
    printf("%6d B per block and phase: %d dependent launches %.2f us each; one launch with %d grid barriers %.2f us per phase\n",
           words * 4, phases, ms_l * 1e3f / phases, phases, ms_f * 1e3f / phases);
    (void)hipFree(d);
    (void)hipFree(c);
  }
  return 0;
}
