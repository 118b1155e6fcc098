// Evaluation metrics of the reference's offline scripts, on the device:
//   inferrence.py:188-204  rescale a volume to 0..255 (ScaleIntensityRangePercentiles 0/100 ==
//                          min/max), round, then MAE against the ground truth;
//   metrics.py:213-223, psnr_ssim_metric.py:88-106  MSE and PSNR with data_range = 256.
// HBM-bound two-stage reductions (fixed order, no atomics).
#include "mpgan_common.h"

namespace mpgan {

constexpr int MET_BLOCKS = 1024;

__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, long n,
                                                             float* __restrict__ partials) {
  __shared__ float smin[4], smax[4];
  float lo = 3.4e38f, hi = -3.4e38f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { smin[w] = lo; smax[w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    partials[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const float* __restrict__ partials, int nb,
                                                          float* __restrict__ out2) {
  float lo = 3.4e38f, hi = -3.4e38f;
  for (int i = threadIdx.x; i < nb; i += 64) {
    lo = fminf(lo, partials[2 * i]);
    hi = fmaxf(hi, partials[2 * i + 1]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  if (threadIdx.x == 0) { out2[0] = lo; out2[1] = hi; }
}

// y = round((x - min) / (max - min) * (b_max - b_min) + b_min), clipped to [b_min, b_max]
__global__ __launch_bounds__(256) void rescale_round_kernel(const float* __restrict__ x, long n,
                                                            const float* __restrict__ mm, float b_min, float b_max,
                                                            int do_round, float* __restrict__ y) {
  const float lo = mm[0], hi = mm[1];
  const float span = hi - lo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = span != 0.f ? (x[i] - lo) / span : 0.f;
    v = v * (b_max - b_min) + b_min;
    v = fminf(fmaxf(v, b_min), b_max);
    y[i] = do_round ? rintf(v) : v;
  }
}

__global__ __launch_bounds__(256) void err_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          long n, float* __restrict__ partials) {
  __shared__ float s1[4], s2[4];
  float ae = 0.f, se = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    ae += fabsf(d);
    se += d * d;
  }
  ae = wave_sum(ae);
  se = wave_sum(se);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { s1[w] = ae; s2[w] = se; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];
    partials[2 * blockIdx.x + 1] = s2[0] + s2[1] + s2[2] + s2[3];
  }
}

// out3 = (MAE, MSE, PSNR = 10 log10(range^2 / MSE))
__global__ __launch_bounds__(64) void err_final_kernel(const float* __restrict__ partials, int nb, double inv_n,
                                                       float data_range, float* __restrict__ out3) {
  double ae = 0.0, se = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) {
    ae += (double)partials[2 * i];
    se += (double)partials[2 * i + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ae += __shfl_xor(ae, off, 64);
    se += __shfl_xor(se, off, 64);
  }
  if (threadIdx.x == 0) {
    const double mae = ae * inv_n, mse = se * inv_n;
    out3[0] = (float)mae;
    out3[1] = (float)mse;
    out3[2] = mse > 0.0 ? (float)(10.0 * log10((double)data_range * data_range / mse)) : INFINITY;
  }
}

}  // namespace mpgan

using namespace mpgan;

extern "C" int32_t mpgan_metric_partials(void) { return 2 * MET_BLOCKS; }

extern "C" int mpgan_rescale_minmax(const float* x, int64_t numel, float b_min, float b_max, int32_t do_round,
                                    float* partials, float* minmax2, float* y, void* stream) {
  MPGAN_CHECK_ARG(x && partials && minmax2 && y && numel > 0, "rescale_minmax: bad argument");
  long blocks = (numel + 255) / 256;
  if (blocks > MET_BLOCKS) blocks = MET_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long)numel, partials);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, partials, (int)blocks, minmax2);
  hipLaunchKernelGGL(rescale_round_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long)numel, minmax2, b_min,
                     b_max, do_round, y);
  return check_launch("rescale_minmax");
}

extern "C" int mpgan_image_errors(const float* a, const float* b, int64_t numel, float data_range, float* partials,
                                  float* out3, void* stream) {
  MPGAN_CHECK_ARG(a && b && partials && out3 && numel > 0, "image_errors: bad argument");
  long blocks = (numel + 255) / 256;
  if (blocks > MET_BLOCKS) blocks = MET_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(err_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, (long)numel, partials);
  hipLaunchKernelGGL(err_final_kernel, dim3(1), dim3(64), 0, st, partials, (int)blocks, 1.0 / (double)numel,
                     data_range, out3);
  return check_launch("image_errors");
}
