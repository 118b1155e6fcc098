// Sustained fp32 MFMA rate of this chip: every wave issues independent v_mfma_f32_32x32x2_f32
// back to back from registers (no memory traffic).  Development aid: the reference point for
// "how busy is the matrix pipe at the clock the chip actually holds under this load".
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, int mode) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // operands: 16 + 16 per-lane pseudo-random values in registers (realistic bit toggling; constant
  // operands let the chip hold its boost clock and flatter the number)
  float a[16], b[16];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + (unsigned)mode;
  for (int u = 0; u < 16; ++u) {
    h = h * 1664525u + 1013904223u;
    a[u] = mode ? ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.5f;
    h = h * 1664525u + 1013904223u;
    b[u] = mode ? ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1e-3f : 0.25f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int mode = 0; mode <= 1; ++mode)
    for (int wpb = 1; wpb <= 2; ++wpb) {        // blocks per CU (4 waves each)
      const int blocks = 256 * wpb, iters = 40000;
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, 100, mode);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
      printf("%s operands, %d block(s)/CU: %.1f ms, %.1f TFLOP/s (fp32 MFMA 32x32x2, registers only)\n",
             mode ? "random  " : "constant", wpb, ms, flops / ms / 1e9);
    }
  return 0;
}
