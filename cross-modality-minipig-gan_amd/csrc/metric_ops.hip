// Evaluation metrics of the reference's offline scripts, on the device:
//   inferrence.py:188-204  rescale a volume to 0..255 (ScaleIntensityRangePercentiles 0/100 ==
//                          min/max), round, then MAE against the ground truth;
//   metrics.py:213-223, psnr_ssim_metric.py:88-106  MSE, PSNR and SSIM with data_range = 256.
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

// ---- SSIM (skimage.metrics.structural_similarity as psnr_ssim_metric.py:91-92 calls it) -------
// 7-wide uniform window (7x7 for a slice, 7x7x7 for a volume), K1 = 0.01, K2 = 0.03, sample
// covariance (NP/(NP-1)), mean of S over the interior that excludes the 3 border samples per
// windowed axis.  A block stages the (TZ+WZ-1) x 14 x 38 input region of both images in LDS;
// a thread owns one (y, x) column of the 8 x 32 tile: it forms the 7x7 plane sums of
// (a, b, a^2, b^2, ab) for every staged z once (double precision: sums of 343 products of
// 0..255 values would cost fp32 its variance digits) and slides the z window over them.
constexpr int SS_W = 7, SS_TY = 8, SS_TX = 32, SS_RY = SS_TY + SS_W - 1, SS_RX = SS_TX + SS_W - 1;

template <int WZ, int TZ>
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           int D, int H, int W, int tiles_x, int tiles_y,
                                                           double c1, double c2, double* __restrict__ partials) {
  constexpr int RZ = TZ + WZ - 1;
  extern __shared__ float sm[];                 // [2][RZ][SS_RY][SS_RX]
  float* sa = sm;
  float* sb = sm + RZ * SS_RY * SS_RX;
  const int tid = threadIdx.x;
  const int bx = blockIdx.x % tiles_x;
  const int by = (blockIdx.x / tiles_x) % tiles_y;
  const int bz = blockIdx.x / (tiles_x * tiles_y);
  const int z0 = bz * TZ, y0 = by * SS_TY, x0 = bx * SS_TX;   // origin of the tile = first window corner
  for (int i = tid; i < RZ * SS_RY * SS_RX; i += 256) {
    const int rx = i % SS_RX, ry = (i / SS_RX) % SS_RY, rz = i / (SS_RX * SS_RY);
    const int z = z0 + rz, y = y0 + ry, x = x0 + rx;
    const bool ok = z < D && y < H && x < W;
    const long off = ((long)z * H + y) * W + x;
    sa[i] = ok ? a[off] : 0.f;
    sb[i] = ok ? b[off] : 0.f;
  }
  __syncthreads();
  const int tx = tid % SS_TX, ty = tid / SS_TX;
  const int OD = D - WZ + 1, OH = H - SS_W + 1, OW = W - SS_W + 1;   // outputs = window corners
  double pa[RZ], pb[RZ], paa[RZ], pbb[RZ], pab[RZ];
#pragma unroll
  for (int rz = 0; rz < RZ; ++rz) {
    double s_a = 0, s_b = 0, s_aa = 0, s_bb = 0, s_ab = 0;
    for (int dy = 0; dy < SS_W; ++dy) {
      const float* ra = sa + (rz * SS_RY + ty + dy) * SS_RX + tx;
      const float* rb = sb + (rz * SS_RY + ty + dy) * SS_RX + tx;
#pragma unroll
      for (int dx = 0; dx < SS_W; ++dx) {
        const double u = (double)ra[dx], v = (double)rb[dx];
        s_a += u; s_b += v;
        s_aa = fma(u, u, s_aa); s_bb = fma(v, v, s_bb); s_ab = fma(u, v, s_ab);
      }
    }
    pa[rz] = s_a; pb[rz] = s_b; paa[rz] = s_aa; pbb[rz] = s_bb; pab[rz] = s_ab;
  }
  constexpr double NP = (double)(WZ * SS_W * SS_W);
  constexpr double cov_norm = NP / (NP - 1.0);
  double local = 0.0;
#pragma unroll
  for (int zo = 0; zo < TZ; ++zo) {
    if (z0 + zo >= OD || y0 + ty >= OH || x0 + tx >= OW) continue;
    double s_a = 0, s_b = 0, s_aa = 0, s_bb = 0, s_ab = 0;
#pragma unroll
    for (int dz = 0; dz < WZ; ++dz) {
      s_a += pa[zo + dz]; s_b += pb[zo + dz]; s_aa += paa[zo + dz]; s_bb += pbb[zo + dz]; s_ab += pab[zo + dz];
    }
    const double ux = s_a / NP, uy = s_b / NP;
    const double vx = cov_norm * (s_aa / NP - ux * ux), vy = cov_norm * (s_bb / NP - uy * uy);
    const double vxy = cov_norm * (s_ab / NP - ux * uy);
    const double A1 = 2.0 * ux * uy + c1, A2 = 2.0 * vxy + c2;
    const double B1 = ux * ux + uy * uy + c1, B2 = vx + vy + c2;
    local += (A1 * A2) / (B1 * B2);
  }
  // block sum in a fixed order
  __shared__ double red[256];
  red[tid] = local;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (tid < s2) red[tid] += red[tid + s2];
    __syncthreads();
  }
  if (tid == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ssim_final_kernel(const double* __restrict__ partials, int nb, double inv_count,
                                                         float* __restrict__ out1) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) out1[0] = (float)(red[0] * inv_count);
}

// ---- np.percentile on the device (ScaleIntensityRangePercentilesd, GAN_final.py:386-394) --------
// Exact order statistics by a 3-pass radix select (11 + 11 + 10 bits) over the monotone uint32
// image of the floats; integer atomics only, so the result does not depend on execution order.
constexpr int PCT_MAXT = 4;                     // order statistics per call (2 percentiles x {lo, lo+1})
constexpr int PCT_BINS = 2048;
struct PctState {
  unsigned prefix[PCT_MAXT];                    // key bits fixed so far (left-aligned)
  unsigned long long k[PCT_MAXT];               // rank still to find inside the prefix class
};

__device__ __forceinline__ unsigned pct_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pct_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// pass 0: bits 31..21, one histogram shared by all targets; pass 1: bits 20..10 among the keys whose
// top 11 bits equal the target's prefix; pass 2: bits 9..0 among those whose top 22 bits do.
template <int PASS>
__global__ __launch_bounds__(256) void pct_hist_kernel(const float* __restrict__ x, long n, int nt,
                                                       const PctState* __restrict__ stt, unsigned* __restrict__ hist) {
  extern __shared__ unsigned lh[];              // [PASS == 0 ? 1 : nt][PCT_BINS]
  const int nh = PASS == 0 ? 1 : nt;
  for (int i = threadIdx.x; i < nh * PCT_BINS; i += 256) lh[i] = 0;
  unsigned pre[PCT_MAXT];
#pragma unroll
  for (int t = 0; t < PCT_MAXT; ++t) pre[t] = (PASS > 0 && t < nt) ? stt->prefix[t] : 0u;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const unsigned key = pct_key(x[i]);
    if (PASS == 0) {
      atomicAdd(&lh[key >> 21], 1u);
    } else {
#pragma unroll
      for (int t = 0; t < PCT_MAXT; ++t) {
        if (t >= nt) break;
        if (PASS == 1 && (key >> 21) == (pre[t] >> 21)) atomicAdd(&lh[t * PCT_BINS + ((key >> 10) & 2047u)], 1u);
        if (PASS == 2 && (key >> 10) == (pre[t] >> 10)) atomicAdd(&lh[t * PCT_BINS + (key & 1023u)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nh * PCT_BINS; i += 256)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// one thread per target walks its histogram to the bin that holds rank k; clears the histogram
template <int PASS>
__global__ __launch_bounds__(64) void pct_scan_kernel(int nt, PctState* __restrict__ stt, unsigned* __restrict__ hist) {
  const int t = threadIdx.x;
  if (t < nt) {
    const unsigned* h = hist + (PASS == 0 ? 0 : t * PCT_BINS);
    unsigned long long k = stt->k[t], cum = 0;
    const int bins = PASS == 2 ? 1024 : PCT_BINS;
    int bsel = bins - 1;
    for (int b2 = 0; b2 < bins; ++b2) {
      const unsigned c = h[b2];
      if (k < cum + c) { bsel = b2; break; }
      cum += c;
    }
    stt->k[t] = k - cum;
    const int shift = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    stt->prefix[t] |= (unsigned)bsel << shift;
  }
  __syncthreads();
  const int nh = PASS == 0 ? 1 : nt;
  for (int i = threadIdx.x; i < nh * PCT_BINS; i += 64) hist[i] = 0;
}

__global__ void pct_init_kernel(PctState* stt, unsigned* hist, int nt, unsigned long long k0, unsigned long long k1,
                                unsigned long long k2, unsigned long long k3) {
  const unsigned long long ks[4] = {k0, k1, k2, k3};
  if (threadIdx.x < PCT_MAXT) {
    stt->prefix[threadIdx.x] = 0;
    stt->k[threadIdx.x] = threadIdx.x < nt ? ks[threadIdx.x] : 0;
  }
  for (int i = threadIdx.x; i < PCT_MAXT * PCT_BINS; i += blockDim.x) hist[i] = 0;
}

// out[j] = s[lo_j] + frac_j * (s[lo_j + 1] - s[lo_j])   (numpy's linear interpolation, in double)
__global__ void pct_final_kernel(const PctState* __restrict__ stt, int nq, double f0, double f1,
                                 float* __restrict__ out) {
  const double fr[2] = {f0, f1};
  if ((int)threadIdx.x < nq) {
    const double a = (double)pct_unkey(stt->prefix[2 * threadIdx.x]);
    const double b = (double)pct_unkey(stt->prefix[2 * threadIdx.x + 1]);
    const double t = fr[threadIdx.x];
    out[threadIdx.x] = (float)(t >= 0.5 ? b - (b - a) * (1.0 - t) : a + (b - a) * t);
  }
}

// y = (x - a_min) / (a_max - a_min) * (b_max - b_min) + b_min, optionally clipped (MONAI ScaleIntensityRange)
__global__ __launch_bounds__(256) void scale_range_kernel(const float* __restrict__ x, long n,
                                                          const float* __restrict__ a_minmax, float b_min, float b_max,
                                                          int clip, float* __restrict__ y) {
  const float lo = a_minmax[0], hi = a_minmax[1];
  const float span = hi - lo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i] - lo;
    if (span != 0.f) {                           // MONAI: a_min == a_max -> warn and return x - a_min
      v = v / span * (b_max - b_min) + b_min;
      if (clip) v = fminf(fmaxf(v, b_min), b_max);
    }
    y[i] = v;
  }
}

constexpr int SS_TZ3 = 4;

static long ssim_blocks(const int32_t* dhw, bool vol) {
  const int OD = vol ? dhw[0] - SS_W + 1 : dhw[0], OH = dhw[1] - SS_W + 1, OW = dhw[2] - SS_W + 1;
  const long tz = vol ? (OD + SS_TZ3 - 1) / SS_TZ3 : OD;
  return tz * ((OH + SS_TY - 1) / SS_TY) * ((OW + SS_TX - 1) / SS_TX);
}

}  // namespace mpgan

using namespace mpgan;

extern "C" int64_t mpgan_ssim_workspace(const int32_t* dhw) {
  if (!dhw || dhw[1] < SS_W || dhw[2] < SS_W || dhw[0] < 1) return -1;
  const bool vol = dhw[0] >= SS_W;
  return ssim_blocks(dhw, vol) * (int64_t)sizeof(double);
}

extern "C" int mpgan_ssim(const float* a, const float* b, const int32_t* dhw, float data_range, void* workspace,
                          int64_t workspace_bytes, float* out1, void* stream) {
  MPGAN_CHECK_ARG(a && b && dhw && workspace && out1, "ssim: null pointer");
  MPGAN_CHECK_ARG(dhw[1] >= SS_W && dhw[2] >= SS_W && dhw[0] >= 1, "ssim: extents below the 7-wide window");
  MPGAN_UNSUPPORTED(dhw[0] > 1 && dhw[0] < SS_W, "ssim: depth %d is neither a slice (1) nor >= 7", dhw[0]);
  const bool vol = dhw[0] >= SS_W;
  const long blocks = ssim_blocks(dhw, vol);
  MPGAN_CHECK_ARG(workspace_bytes >= blocks * (int64_t)sizeof(double), "ssim: workspace too small");
  MPGAN_CHECK_ARG(blocks < (1L << 31), "ssim: too many tiles");
  const int OD = vol ? dhw[0] - SS_W + 1 : dhw[0], OH = dhw[1] - SS_W + 1, OW = dhw[2] - SS_W + 1;
  const int tiles_x = (OW + SS_TX - 1) / SS_TX, tiles_y = (OH + SS_TY - 1) / SS_TY;
  const double c1 = (0.01 * (double)data_range) * (0.01 * (double)data_range);
  const double c2 = (0.03 * (double)data_range) * (0.03 * (double)data_range);
  double* part = static_cast<double*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  if (vol) {
    const size_t smem = 2ul * (SS_TZ3 + SS_W - 1) * SS_RY * SS_RX * sizeof(float);
    hipLaunchKernelGGL((ssim_partial_kernel<SS_W, SS_TZ3>), dim3((unsigned)blocks), dim3(256), smem, st, a, b, dhw[0],
                       dhw[1], dhw[2], tiles_x, tiles_y, c1, c2, part);
  } else {
    const size_t smem = 2ul * SS_RY * SS_RX * sizeof(float);
    hipLaunchKernelGGL((ssim_partial_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), smem, st, a, b, dhw[0], dhw[1],
                       dhw[2], tiles_x, tiles_y, c1, c2, part);
  }
  hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(256), 0, st, part, (int)blocks,
                     1.0 / ((double)OD * OH * OW), out1);
  return check_launch("ssim");
}

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

extern "C" int64_t mpgan_percentile_workspace(void) {
  return (int64_t)sizeof(PctState) + (int64_t)PCT_MAXT * PCT_BINS * (int64_t)sizeof(unsigned) + 64;
}

extern "C" int mpgan_percentiles(const float* x, int64_t numel, const double* q_host, int32_t nq, void* workspace,
                                 int64_t workspace_bytes, float* out, void* stream) {
  MPGAN_CHECK_ARG(x && q_host && workspace && out && numel > 0, "percentiles: bad argument");
  MPGAN_UNSUPPORTED(nq < 1 || nq > 2, "percentiles: 1 or 2 percentiles per call (got %d)", nq);
  MPGAN_CHECK_ARG(workspace_bytes >= mpgan_percentile_workspace(), "percentiles: workspace too small");
  MPGAN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "percentiles: workspace must be 8-byte aligned");
  unsigned long long ks[4] = {0, 0, 0, 0};
  double fr[2] = {0, 0};
  for (int j = 0; j < nq; ++j) {
    MPGAN_CHECK_ARG(q_host[j] >= 0.0 && q_host[j] <= 100.0, "percentiles: q outside [0, 100]");
    const double r = q_host[j] / 100.0 * (double)(numel - 1);      // numpy: virtual index q * (n - 1)
    long lo = (long)r;
    if (lo > numel - 1) lo = numel - 1;
    const long hi = lo + 1 < numel ? lo + 1 : lo;
    ks[2 * j] = (unsigned long long)lo;
    ks[2 * j + 1] = (unsigned long long)hi;
    fr[j] = r - (double)lo;
  }
  PctState* stt = static_cast<PctState*>(workspace);
  unsigned* hist = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + ((sizeof(PctState) + 63) / 64) * 64);
  const int nt = 2 * nq;
  hipStream_t st = (hipStream_t)stream;
  long blocks = (numel + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pct_init_kernel, dim3(1), dim3(256), 0, st, stt, hist, nt, ks[0], ks[1], ks[2], ks[3]);
  hipLaunchKernelGGL((pct_hist_kernel<0>), dim3((unsigned)blocks), dim3(256), PCT_BINS * sizeof(unsigned), st, x,
                     (long)numel, nt, stt, hist);
  hipLaunchKernelGGL((pct_scan_kernel<0>), dim3(1), dim3(64), 0, st, nt, stt, hist);
  hipLaunchKernelGGL((pct_hist_kernel<1>), dim3((unsigned)blocks), dim3(256), nt * PCT_BINS * sizeof(unsigned), st, x,
                     (long)numel, nt, stt, hist);
  hipLaunchKernelGGL((pct_scan_kernel<1>), dim3(1), dim3(64), 0, st, nt, stt, hist);
  hipLaunchKernelGGL((pct_hist_kernel<2>), dim3((unsigned)blocks), dim3(256), nt * PCT_BINS * sizeof(unsigned), st, x,
                     (long)numel, nt, stt, hist);
  hipLaunchKernelGGL((pct_scan_kernel<2>), dim3(1), dim3(64), 0, st, nt, stt, hist);
  hipLaunchKernelGGL(pct_final_kernel, dim3(1), dim3(64), 0, st, stt, nq, fr[0], fr[1], out);
  return check_launch("percentiles");
}

extern "C" int mpgan_scale_intensity_range(const float* x, int64_t numel, const float* a_minmax, float b_min,
                                           float b_max, int32_t clip, float* y, void* stream) {
  MPGAN_CHECK_ARG(x && a_minmax && y && numel > 0, "scale_intensity_range: bad argument");
  long blocks = (numel + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_range_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)numel,
                     a_minmax, b_min, b_max, clip, y);
  return check_launch("scale_intensity_range");
}

// ---------------------------------------------------------------------------------------------------
// Resampling onto the 256 mm identity grid (next-row N3, code/GAN/transforms.py:79-213: ResampleT1T2d =
// itk.resample_image_filter(identity transform, LinearInterpolateImageFunction, reference image with
// origin = -size/2, spacing = 256/size, identity direction), default pixel 0).  ITK is not in this image; the
// kernel restates ITK 5's published algorithm:
//   p   = origin_out + i * spacing_out                         (output physical point, identity direction)
//   c   = (Direction_in * diag(spacing_in))^-1 (p - origin_in)  (continuous index into the input)
//   inside  <=>  -0.5 <= c_d < size_d - 0.5 for every d        (ImageFunction::IsInsideBuffer)
//   value   = trilinear interpolation at c, the base index clamped to [0, size-1] and a neighbour beyond the last
//             index dropped (LinearInterpolateImageFunction::EvaluateOptimized), else 0.
// One thread per output voxel: an HBM/L2-bound gather (8 reads, 1 write).
// ---------------------------------------------------------------------------------------------------
struct ResampleGeom {
  double m[9];        // inverse of Direction*diag(spacing), row-major, in ITK (x, y, z) order
  double origin_in[3];
  double origin_out[3], spacing_out[3];
  int in_size[3];     // (x, y, z)
  int out_size[3];
};

__global__ __launch_bounds__(256) void resample_linear_kernel(const float* __restrict__ in, ResampleGeom g,
                                                              float* __restrict__ out) {
  const long total = (long)g.out_size[0] * g.out_size[1] * g.out_size[2];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % g.out_size[0]);
    const long q = i / g.out_size[0];
    const int oy = (int)(q % g.out_size[1]), oz = (int)(q / g.out_size[1]);
    const double d0 = g.origin_out[0] + ox * g.spacing_out[0] - g.origin_in[0];
    const double d1 = g.origin_out[1] + oy * g.spacing_out[1] - g.origin_in[1];
    const double d2 = g.origin_out[2] + oz * g.spacing_out[2] - g.origin_in[2];
    double c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) c[r] = g.m[3 * r] * d0 + g.m[3 * r + 1] * d1 + g.m[3 * r + 2] * d2;
    bool inside = true;
    int b[3];
    double f[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      inside = inside && c[d] >= -0.5 && c[d] < (double)g.in_size[d] - 0.5;
      int bi = (int)floor(c[d]);
      if (bi < 0) bi = 0;
      b[d] = bi;
      double fr = c[d] - (double)bi;
      if (fr < 0.0) fr = 0.0;                       // c in [-0.5, 0): the base sample alone
      if (bi + 1 > g.in_size[d] - 1) fr = 0.0;      // no neighbour beyond the last index
      f[d] = fr;
    }
    float v = 0.f;
    if (inside) {
      const long sx = 1, sy = g.in_size[0], sz = (long)g.in_size[0] * g.in_size[1];
      const long base = b[0] * sx + b[1] * sy + b[2] * sz;
      const long ax = f[0] > 0.0 ? sx : 0, ay = f[1] > 0.0 ? sy : 0, az = f[2] > 0.0 ? sz : 0;
      const double v000 = in[base], v100 = in[base + ax], v010 = in[base + ay], v110 = in[base + ax + ay];
      const double v001 = in[base + az], v101 = in[base + ax + az], v011 = in[base + ay + az],
                   v111 = in[base + ax + ay + az];
      const double x00 = v000 + f[0] * (v100 - v000), x10 = v010 + f[0] * (v110 - v010);
      const double x01 = v001 + f[0] * (v101 - v001), x11 = v011 + f[0] * (v111 - v011);
      const double y0 = x00 + f[1] * (x10 - x00), y1 = x01 + f[1] * (x11 - x01);
      v = (float)(y0 + f[2] * (y1 - y0));
    }
    out[i] = v;
  }
}

extern "C" int mpgan_resample_to_identity_grid(const float* vol, const int32_t* in_dhw, const double* origin_xyz,
                                               const double* spacing_xyz, const double* direction_3x3,
                                               const int32_t* out_dhw, double extent_mm, float* out, void* stream) {
  MPGAN_CHECK_ARG(vol && in_dhw && origin_xyz && spacing_xyz && direction_3x3 && out_dhw && out && extent_mm > 0,
                  "resample: null argument");
  ResampleGeom g;
  double a[9];   // A = Direction * diag(spacing)
  for (int d = 0; d < 3; ++d) {
    MPGAN_CHECK_ARG(in_dhw[d] > 0 && out_dhw[d] > 0 && spacing_xyz[d] > 0, "resample: bad size / spacing in dim %d", d);
    g.in_size[d] = in_dhw[2 - d];               // arrays are (z, y, x); ITK indexes (x, y, z)
    g.out_size[d] = out_dhw[2 - d];
    g.origin_in[d] = origin_xyz[d];
    g.origin_out[d] = -0.5 * g.out_size[d];      // transforms.py:144: SetOrigin(-output_size / 2)
    g.spacing_out[d] = extent_mm / g.out_size[d];
    for (int r = 0; r < 3; ++r) a[3 * r + d] = direction_3x3[3 * r + d] * spacing_xyz[d];
  }
  const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  MPGAN_CHECK_ARG(det != 0.0, "resample: singular direction matrix");
  g.m[0] = (a[4] * a[8] - a[5] * a[7]) / det; g.m[1] = (a[2] * a[7] - a[1] * a[8]) / det; g.m[2] = (a[1] * a[5] - a[2] * a[4]) / det;
  g.m[3] = (a[5] * a[6] - a[3] * a[8]) / det; g.m[4] = (a[0] * a[8] - a[2] * a[6]) / det; g.m[5] = (a[2] * a[3] - a[0] * a[5]) / det;
  g.m[6] = (a[3] * a[7] - a[4] * a[6]) / det; g.m[7] = (a[1] * a[6] - a[0] * a[7]) / det; g.m[8] = (a[0] * a[4] - a[1] * a[3]) / det;
  const long total = (long)out_dhw[0] * out_dhw[1] * out_dhw[2];
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(resample_linear_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vol, g, out);
  return check_launch("resample_to_identity_grid");
}
