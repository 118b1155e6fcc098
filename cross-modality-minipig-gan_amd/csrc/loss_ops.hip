// Discriminator head (Flatten + Linear(F,1) + Sigmoid + BCE), L1 loss, Adam,
// weight repacking, pointwise helpers and the variant-B patch gather/scatter.
// All HBM-bound: 16-byte accesses, grid-stride loops, two-stage deterministic
// reductions (no atomics).
#include "mpgan_common.h"

namespace mpgan {

constexpr int LIN_CHUNKS = 64;
constexpr int L1_PARTIALS = 1024;

static inline int ew_blocks2(long total) {
  long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

// ---- linear head -----------------------------------------------------------
__global__ __launch_bounds__(256) void linear1_partial_kernel(const float* __restrict__ z, Pro p, long P, int C,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ partials) {
  __shared__ float sh[4];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const long F = P * C;
  const long per = ((F / 4 + LIN_CHUNKS - 1) / LIN_CHUNKS) * 4;
  const long beg = (long)chunk * per;
  const long end = beg + per < F ? beg + per : F;
  const float slope = pro_slope(p);
  const float* zn = z + (long)n * F;
  float acc = 0.f;
  for (long k = beg + (long)threadIdx.x * 4; k < end; k += 1024) {
    float4 zv = *reinterpret_cast<const float4*>(zn + k);
    float4 wv = *reinterpret_cast<const float4*>(w + k);
    float a[4] = {zv.x, zv.y, zv.z, zv.w};
    if (p.scale) {
      const int c = (int)(k % C);
      const int si = n * p.n_stride + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = act_apply(a[e] * p.scale[si + e] + p.shift[si + e], p.act, slope);
    }
    acc += a[0] * wv.x + a[1] * wv.y + a[2] * wv.z + a[3] * wv.w;
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partials[n * LIN_CHUNKS + chunk] = acc;
}

__global__ void linear1_final_kernel(const float* __restrict__ partials, const float* bias, int n,
                                     float* __restrict__ logit, float* __restrict__ prob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < LIN_CHUNKS; ++k) s += partials[i * LIN_CHUNKS + k];
  s += bias ? bias[0] : 0.f;
  logit[i] = s;
  if (prob) prob[i] = 1.f / (1.f + expf(-s));
}

__global__ __launch_bounds__(256) void bce_forward_kernel(const float* __restrict__ prob,
                                                          const float* __restrict__ target, int n,
                                                          float* __restrict__ loss) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float pr = prob[i], t = target[i];
    acc += -(t * fmaxf(logf(pr), -100.f) + (1.f - t) * fmaxf(logf(1.f - pr), -100.f));
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) *loss = acc / (float)n;
}

__global__ void bce_backward_kernel(const float* __restrict__ prob, const float* __restrict__ target, int n,
                                    const float* __restrict__ gout, float* __restrict__ dprob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float pr = prob[i];
  dprob[i] = (pr - target[i]) / fmaxf((1.f - pr) * pr, 1e-12f) * (gout[0] / (float)n);
}

__global__ void sigmoid_backward_kernel(const float* __restrict__ dprob, const float* __restrict__ prob, int n,
                                        float* __restrict__ dlogit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float pr = prob[i];
  dlogit[i] = dprob[i] * (1.f - pr) * pr;
}

__global__ void sigmoid_forward_kernel(const float* __restrict__ logit, int n, float* __restrict__ prob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) prob[i] = 1.f / (1.f + expf(-logit[i]));
}

__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ scalar, long numel,
                                                              float* __restrict__ y) {
  const float s = scalar[0];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) y[i] = x[i] * s;
}

// one thread per 4 consecutive k (channels-last order); loops over the batch
__global__ __launch_bounds__(256) void linear1_backward_kernel(const float* __restrict__ z, Pro p, int N, long P,
                                                               int C, const float* __restrict__ w,
                                                               const float* __restrict__ dlogit,
                                                               float* __restrict__ g_a, float* __restrict__ dw,
                                                               float beta) {
  const long F = P * C;
  const float slope = pro_slope(p);
  for (long k = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; k < F; k += (long)gridDim.x * blockDim.x * 4) {
    const int c = (int)(k % C);
    const long pix = k / C;
    const float4 wv = *reinterpret_cast<const float4*>(w + k);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
      const float d = dlogit[n];
      if (g_a) *reinterpret_cast<float4*>(g_a + (long)n * F + k) = make_float4(d * wv.x, d * wv.y, d * wv.z, d * wv.w);
      if (dw) {
        float4 zv = *reinterpret_cast<const float4*>(z + (long)n * F + k);
        float a[4] = {zv.x, zv.y, zv.z, zv.w};
        if (p.scale) {
          const int si = n * p.n_stride + c;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = act_apply(a[e] * p.scale[si + e] + p.shift[si + e], p.act, slope);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += d * a[e];
      }
    }
    if (dw) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long o = (long)(c + e) * P + pix;  // torch flatten order: c-major
        dw[o] = beta != 0.f ? beta * dw[o] + acc[e] : acc[e];
      }
    }
  }
}

__global__ void vec_sum_accum_kernel(const float* __restrict__ v, int n, float* out, float beta) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i];
    *out = beta != 0.f ? beta * (*out) + s : s;
  }
}

// out[0] = sum_i v[i] * w[i] in index order (a few dozen device scalars: the perceptual loss's per-tap terms)
__global__ void weighted_sum_kernel(const float* __restrict__ v, const float* __restrict__ w, int n, float* out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += v[i] * w[i];
    *out = s;
  }
}

// ---- sigmoid + BCE ----------------------------------------------------------
__global__ __launch_bounds__(256) void sigmoid_bce_kernel(const float* __restrict__ logit, int n, float target,
                                                          float loss_scale, float* __restrict__ prob,
                                                          float* __restrict__ loss, float* __restrict__ dlogit) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = logit[i];
    const float pr = 1.f / (1.f + expf(-x));
    if (prob) prob[i] = pr;
    const float lp = fmaxf(logf(pr), -100.f);
    const float lq = fmaxf(logf(1.f - pr), -100.f);
    acc += -(target * lp + (1.f - target) * lq);
    if (dlogit) {
      // autograd: binary_cross_entropy_backward then sigmoid_backward
      const float gp = (pr - target) / fmaxf((1.f - pr) * pr, 1e-12f) * (loss_scale / (float)n);
      dlogit[i] = gp * (1.f - pr) * pr;
    }
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0 && loss) *loss = acc / (float)n;
}

// ---- L1 ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         long numel, float gscale, float* __restrict__ partials,
                                                         float* __restrict__ grad) {
  __shared__ float sh[4];
  float acc = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const float d = a[i] - b[i];
    acc += fabsf(d);
    if (grad) grad[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ partials, int n, double inv_numel,
                                                       float* __restrict__ loss) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += sh[i];
    *loss = (float)(t * inv_numel);
  }
}

// ---- Adam ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long numel,
                                                   float w1, float b2, float one_minus_b2, float step_size,
                                                   float bc2_sqrt, float eps, float gscale) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const float gi = g[i] * gscale;
    float mi = m[i], vi = v[i];
    // torch: exp_avg.lerp_(grad, 1-beta1)
    const float diff = gi - mi;
    mi = w1 < 0.5f ? mi + w1 * diff : gi - diff * (1.f - w1);
    vi = vi * b2 + one_minus_b2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// ---- weight repack (table driven, one launch per network) -----------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           const int64_t* __restrict__ table) {
  const int64_t* e = table + (long)blockIdx.y * 8;
  const long so = e[0], dof = e[1];
  const int Cout = (int)e[2], Cin = (int)e[3], T = (int)e[4];
  const int transposed = (int)e[5], dgrad = (int)e[6];
  const long total = (long)Cout * Cin * T;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int co, ci, t;
    if (dgrad == 0) {  // dst [co][t][ci]
      ci = (int)(i % Cin);
      long q = i / Cin;
      t = (int)(q % T);
      co = (int)(q / T);
    } else if (dgrad == 1) {  // dst [ci][t][co]
      co = (int)(i % Cout);
      long q = i / Cout;
      t = (int)(q % T);
      ci = (int)(q / T);
    } else {  // dst [t][ci][co]: a Linear-as-conv's data gradient run as one (P x Cout) * (Cout x T*Cin) GEMM
      co = (int)(i % Cout);
      long q = i / Cout;
      ci = (int)(q % Cin);
      t = (int)(q / Cin);
    }
    const long s = transposed ? ((long)ci * Cout + co) * T + t : ((long)co * Cin + ci) * T + t;
    dst[dof + i] = src[so + s];
  }
}

// ---- pointwise -------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_tanh_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       long numel, int apply_tanh, float* __restrict__ y) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    float v = a[i] + (b ? b[i] : 0.f);
    y[i] = apply_tanh ? tanhf(v) : v;
  }
}

__global__ __launch_bounds__(256) void tanh_backward_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                            long numel, float* __restrict__ dx) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const float t = y[i];
    dx[i] = g[i] * (1.f - t * t);
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, float alpha,
                                                    const float* __restrict__ b, float beta, long numel,
                                                    float* __restrict__ y) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride)
    y[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// ---- patches -----------------------------------------------------------------------
struct PatchGeom {
  int B, D, H, W, S, rz, ry, rx;
};

__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ vol, PatchGeom g,
                                                           const int32_t* __restrict__ corners,
                                                           float* __restrict__ patches) {
  const long per = (long)g.rz * g.ry * g.rx;
  const long total = (long)g.B * g.S * per;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long pidx = i / per;
    long r = i - pidx * per;
    const int x = (int)(r % g.rx);
    r /= g.rx;
    const int y = (int)(r % g.ry);
    const int z = (int)(r / g.ry);
    const int b = (int)(pidx / g.S);
    const int32_t* c = corners + pidx * 3;
    patches[i] = vol[(((long)b * g.D + c[0] + z) * g.H + c[1] + y) * g.W + c[2] + x];
  }
}

// one thread per volume voxel; walks the samples of its volume in order
__global__ __launch_bounds__(256) void patch_scatter_kernel(const float* __restrict__ dp, PatchGeom g,
                                                            const int32_t* __restrict__ corners,
                                                            float* __restrict__ dvol) {
  const long per = (long)g.rz * g.ry * g.rx;
  const long vox = (long)g.D * g.H * g.W;
  const long total = (long)g.B * vox;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i / vox);
    long r = i - (long)b * vox;
    const int x = (int)(r % g.W);
    r /= g.W;
    const int y = (int)(r % g.H);
    const int z = (int)(r / g.H);
    float acc = 0.f;
    for (int s = 0; s < g.S; ++s) {
      const int32_t* c = corners + ((long)b * g.S + s) * 3;
      const int dz = z - c[0], dy = y - c[1], dx = x - c[2];
      if ((unsigned)dz < (unsigned)g.rz && (unsigned)dy < (unsigned)g.ry && (unsigned)dx < (unsigned)g.rx)
        acc += dp[((long)b * g.S + s) * per + ((long)dz * g.ry + dy) * g.rx + dx];
    }
    dvol[i] += acc;
  }
}

}  // namespace mpgan

using namespace mpgan;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int32_t mpgan_linear1_partials(int32_t n) { return n * LIN_CHUNKS; }

extern "C" int mpgan_linear1_forward(const float* z, const mpgan_prologue* p, int32_t n, int64_t P, int32_t c,
                                     const float* w_perm, const float* bias, float* partials, float* logit,
                                     float* prob, void* stream) {
  MPGAN_CHECK_ARG(z && w_perm && partials && logit && n > 0 && P > 0 && c > 0, "linear1_forward: bad argument");
  MPGAN_UNSUPPORTED(c % 4 != 0 || !al16(z) || !al16(w_perm), "linear1_forward: needs C %% 4 == 0 and 16-B alignment");
  hipLaunchKernelGGL(linear1_partial_kernel, dim3(LIN_CHUNKS, n), dim3(256), 0, (hipStream_t)stream, z, make_pro(p),
                     (long)P, c, w_perm, partials);
  hipLaunchKernelGGL(linear1_final_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, partials, bias, n,
                     logit, prob);
  return check_launch("linear1_forward");
}

extern "C" int mpgan_linear1_backward(const float* z, const mpgan_prologue* p, int32_t n, int64_t P, int32_t c,
                                      const float* w_perm, const float* dlogit, float* g_a, float* dw, float* dbias,
                                      float beta, void* stream) {
  MPGAN_CHECK_ARG(z && w_perm && dlogit && n > 0 && P > 0 && c > 0, "linear1_backward: bad argument");
  MPGAN_UNSUPPORTED(c % 4 != 0 || !al16(z) || !al16(w_perm) || (g_a && !al16(g_a)),
                    "linear1_backward: needs C %% 4 == 0 and 16-B alignment");
  const long F = (long)P * c;
  hipLaunchKernelGGL(linear1_backward_kernel, dim3(ew_blocks2(F / 4)), dim3(256), 0, (hipStream_t)stream, z,
                     make_pro(p), n, (long)P, c, w_perm, dlogit, g_a, dw, beta);
  if (dbias) hipLaunchKernelGGL(vec_sum_accum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dlogit, n, dbias, beta);
  return check_launch("linear1_backward");
}

extern "C" int mpgan_sigmoid_bce(const float* logit, int32_t n, float target, float loss_scale, float* prob,
                                 float* loss, float* dlogit, void* stream) {
  MPGAN_CHECK_ARG(logit && n > 0, "sigmoid_bce: bad argument");
  hipLaunchKernelGGL(sigmoid_bce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logit, n, target, loss_scale,
                     prob, loss, dlogit);
  return check_launch("sigmoid_bce");
}

extern "C" int mpgan_bce_forward(const float* prob, const float* target, int32_t n, float* loss, void* stream) {
  MPGAN_CHECK_ARG(prob && target && loss && n > 0, "bce_forward: bad argument");
  hipLaunchKernelGGL(bce_forward_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, prob, target, n, loss);
  return check_launch("bce_forward");
}

extern "C" int mpgan_bce_backward(const float* prob, const float* target, int32_t n, const float* gout, float* dprob,
                                  void* stream) {
  MPGAN_CHECK_ARG(prob && target && gout && dprob && n > 0, "bce_backward: bad argument");
  hipLaunchKernelGGL(bce_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob, target, n,
                     gout, dprob);
  return check_launch("bce_backward");
}

extern "C" int mpgan_sigmoid_backward(const float* dprob, const float* prob, int32_t n, float* dlogit, void* stream) {
  MPGAN_CHECK_ARG(dprob && prob && dlogit && n > 0, "sigmoid_backward: bad argument");
  hipLaunchKernelGGL(sigmoid_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dprob, prob, n,
                     dlogit);
  return check_launch("sigmoid_backward");
}

extern "C" int mpgan_sigmoid_forward(const float* logit, int32_t n, float* prob, void* stream) {
  MPGAN_CHECK_ARG(logit && prob && n > 0, "sigmoid_forward: bad argument");
  hipLaunchKernelGGL(sigmoid_forward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, logit, n, prob);
  return check_launch("sigmoid_forward");
}

extern "C" int mpgan_scale_by_device_scalar(const float* x, const float* scalar, int64_t numel, float* y,
                                            void* stream) {
  MPGAN_CHECK_ARG(x && scalar && y && numel > 0, "scale_by_device_scalar: bad argument");
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(ew_blocks2(numel)), dim3(256), 0, (hipStream_t)stream, x, scalar,
                     (long)numel, y);
  return check_launch("scale_by_device_scalar");
}

extern "C" int mpgan_weighted_sum(const float* v, const float* w, int32_t n, float* out, void* stream) {
  MPGAN_CHECK_ARG(v && w && out && n > 0 && n <= 4096, "weighted_sum: bad argument (1..4096 terms)");
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, v, w, n, out);
  return check_launch("weighted_sum");
}

extern "C" int32_t mpgan_l1_partials(void) { return L1_PARTIALS; }

extern "C" int mpgan_l1_loss(const float* a, const float* b, int64_t numel, float grad_scale, float* partials,
                             float* loss, float* grad_a, void* stream) {
  MPGAN_CHECK_ARG(a && b && partials && loss && numel > 0, "l1_loss: bad argument");
  long blocks = (numel + 255) / 256;
  if (blocks > L1_PARTIALS) blocks = L1_PARTIALS;
  hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, (long)numel,
                     grad_scale / (float)numel, partials, grad_a);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)blocks,
                     1.0 / (double)numel, loss);
  return check_launch("l1_loss");
}

extern "C" int mpgan_adam_step(float* p, const float* g, float* m, float* v, int64_t numel, double lr, double b1,
                               double b2, double eps, int32_t step, float grad_scale, void* stream) {
  MPGAN_CHECK_ARG(p && g && m && v && numel > 0 && step > 0, "adam_step: bad argument");
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks2(numel)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)numel,
                     (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), step_size, bc2_sqrt, (float)eps, grad_scale);
  return check_launch("adam_step");
}

extern "C" int mpgan_pack_weights(const float* flat_params, float* packed, const int64_t* table, int32_t n_entries,
                                  int64_t max_elems, void* stream) {
  MPGAN_CHECK_ARG(flat_params && packed && table && n_entries > 0 && max_elems > 0, "pack_weights: bad argument");
  long bx = (max_elems + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)bx, (unsigned)n_entries), dim3(256), 0, (hipStream_t)stream,
                     flat_params, packed, table);
  return check_launch("pack_weights");
}

extern "C" int mpgan_add_tanh(const float* a, const float* b, int64_t numel, int32_t apply_tanh, float* y,
                              void* stream) {
  MPGAN_CHECK_ARG(a && y && numel > 0, "add_tanh: bad argument");
  hipLaunchKernelGGL(add_tanh_kernel, dim3(ew_blocks2(numel)), dim3(256), 0, (hipStream_t)stream, a, b, (long)numel,
                     apply_tanh, y);
  return check_launch("add_tanh");
}

extern "C" int mpgan_tanh_backward(const float* g, const float* y, int64_t numel, float* dx, void* stream) {
  MPGAN_CHECK_ARG(g && y && dx && numel > 0, "tanh_backward: bad argument");
  hipLaunchKernelGGL(tanh_backward_kernel, dim3(ew_blocks2(numel)), dim3(256), 0, (hipStream_t)stream, g, y,
                     (long)numel, dx);
  return check_launch("tanh_backward");
}

extern "C" int mpgan_axpby(const float* a, float alpha, const float* b, float beta, int64_t numel, float* y,
                           void* stream) {
  MPGAN_CHECK_ARG(a && y && numel > 0, "axpby: bad argument");
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks2(numel)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, beta,
                     (long)numel, y);
  return check_launch("axpby");
}

static int patch_geom(PatchGeom& g, int32_t b, const int32_t dhw[3], int32_t samples, const int32_t roi[3]) {
  MPGAN_CHECK_ARG(b > 0 && samples > 0 && dhw && roi, "patch: bad argument");
  for (int d = 0; d < 3; ++d) MPGAN_CHECK_ARG(roi[d] > 0 && roi[d] <= dhw[d], "patch: roi exceeds volume in dim %d", d);
  g.B = b; g.D = dhw[0]; g.H = dhw[1]; g.W = dhw[2]; g.S = samples;
  g.rz = roi[0]; g.ry = roi[1]; g.rx = roi[2];
  return MPGAN_OK;
}

extern "C" int mpgan_patch_gather(const float* vol, int32_t b, const int32_t dhw[3], const int32_t* corners,
                                  int32_t samples, const int32_t roi[3], float* patches, void* stream) {
  MPGAN_CHECK_ARG(vol && corners && patches, "patch_gather: null pointer");
  PatchGeom g;
  int rc = patch_geom(g, b, dhw, samples, roi);
  if (rc) return rc;
  const long total = (long)b * samples * roi[0] * roi[1] * roi[2];
  hipLaunchKernelGGL(patch_gather_kernel, dim3(ew_blocks2(total)), dim3(256), 0, (hipStream_t)stream, vol, g, corners,
                     patches);
  return check_launch("patch_gather");
}

extern "C" int mpgan_patch_scatter_add(const float* dpatches, int32_t b, const int32_t dhw[3], const int32_t* corners,
                                       int32_t samples, const int32_t roi[3], float* dvol, void* stream) {
  MPGAN_CHECK_ARG(dpatches && corners && dvol, "patch_scatter_add: null pointer");
  PatchGeom g;
  int rc = patch_geom(g, b, dhw, samples, roi);
  if (rc) return rc;
  const long total = (long)b * dhw[0] * dhw[1] * dhw[2];
  hipLaunchKernelGGL(patch_scatter_kernel, dim3(ew_blocks2(total)), dim3(256), 0, (hipStream_t)stream, dpatches, g,
                     corners, dvol);
  return check_launch("patch_scatter_add");
}
