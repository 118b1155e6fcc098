// BatchNorm (training mode) / InstanceNorm statistics, fused norm+activation
// apply and their backward -- the HBM-bound half of the GAN step.
//
// All kernels stream channels-last rows (pitch ld) with 16-byte accesses when
// C % 4 == 0, reduce over pixels per thread, then across the block through LDS
// in a FIXED order, and leave one partial row per (sample, chunk); a tiny
// finalize kernel combines partials in double precision.  No atomics anywhere,
// so statistics (and therefore the whole forward) are bitwise reproducible.
#include "mpgan_common.h"
#include "norm_fold.h"

namespace mpgan {

static inline int stats_chunks_host(long pixels_per_sample, int C = 0) {
  // These passes are latency-bound per block: a chunk is FOUR passes of the block's R = 256 / (C / 4) rows, so every
  // thread issues four independent row loads and a 128-channel map at 32^2 still makes 512 blocks (256 rows per
  // chunk whatever C left it with 64 blocks walking 32 dependent passes each: 28 us for a 2 MB tensor).
  long rows = 256;
  if (C >= 4 && C % 4 == 0) {
    long R = 256 / (C / 4);
    if (R < 1) R = 1;
    rows = 4 * R;
    if (rows > 256) rows = 256;
    if (rows < 16) rows = 16;
  }
  long c = pixels_per_sample / rows;
  if (c < 1) c = 1;
  if (c > 256) c = 256;
  return (int)c;
}

template <int V>
struct Vec;
template <>
struct Vec<4> {
  using T = float4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct Vec<1> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};

// ---------------------------------------------------------------------------
// Generic per-channel partial reduction over pixel chunks.
// NQ quantities per element produced by functor F(n, c, pixel-row pointers...).
// grid = (chunks, n); block = 256 threads laid out as R rows x CG column groups.
// ---------------------------------------------------------------------------
struct ReduceGeom {
  int C, CG, R;           // channels, column groups (C/V), rows per pass
  long P;                 // pixels per sample
  int chunks;
};

// Fixed-order sum over the R row slots of red[R][W] (W = NQ * C values per row) into out[W].  With W <= 128 all
// 256 threads take part: G = 256 / W groups each fold rows gq, gq + G, ..., then the G group sums are added in
// order (a 16-channel layer had 48 threads walking 64 rows one after another: the tail of every block).
__device__ __forceinline__ void fold_rows(float* red, int R, int W, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const int G = 256 / W;
  if (G >= 2 && R >= 2 * G) {
    const int i = tid % W, gq = tid / W;
    float s = 0.f;
    if (gq < G)
      for (int rr = gq; rr < R; rr += G) s += red[rr * W + i];
    __syncthreads();                       // every row slot has been read
    if (gq < G) red[gq * W + i] = s;
    __syncthreads();
    if (tid < W) {
      float t = red[tid];
      for (int k = 1; k < G; ++k) t += red[k * W + tid];
      out[tid] = t;
      red[tid] = t;                        // (only this thread reads column tid: the result may stay in row slot 0)
    }
    return;
  }
  for (int i = tid; i < W; i += 256) {
    float s = 0.f;
    for (int rr = 0; rr < R; ++rr) s += red[rr * W + i];
    out[i] = s;
    red[i] = s;
  }
}

template <int V, int NQ, class F>
__device__ __forceinline__ void chunk_reduce(const ReduceGeom& g, float* partials, F f) {
  extern __shared__ float red[];  // [R][NQ][C]
  const int tid = threadIdx.x;
  const int q = tid % g.CG, r = tid / g.CG;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const long per = (g.P + g.chunks - 1) / g.chunks;
  const long beg = (long)chunk * per;
  const long end = beg + per < g.P ? beg + per : g.P;
  float acc[NQ][V];
#pragma unroll
  for (int a = 0; a < NQ; ++a)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[a][e] = 0.f;
  if (r < g.R) {
    for (long pix = beg + r; pix < end; pix += g.R) f(n, q * V, (long)n * g.P + pix, acc);
#pragma unroll
    for (int a = 0; a < NQ; ++a)
#pragma unroll
      for (int e = 0; e < V; ++e) red[(r * NQ + a) * g.C + q * V + e] = acc[a][e];
  }
  __syncthreads();
  fold_rows(red, g.R, NQ * g.C, partials + ((long)n * g.chunks + chunk) * NQ * g.C);
}

template <int V>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ z, int ldz, ReduceGeom g,
                                                            float* __restrict__ partials) {
  chunk_reduce<V, 2>(g, partials, [&](int, int c, long row, float (&acc)[2][V]) {
    float v[V];
    Vec<V>::load(z + row * ldz + c, v);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      acc[0][e] += v[e];
      acc[1][e] += v[e] * v[e];
    }
  });
}

// Perceptual-loss taps of a peer pass (test_runs/GAN.py:288-298: L1 between the two passes'
// activations) enter the backward as:  g_a = g - ca*sign(a_peer - a),
// gy = g_a*act'(y) - cy*sign(y_peer - y), and a dz term -cz*sign(z_peer - z) that bypasses the norm.
template <int V>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const float* __restrict__ gr, int ldg,
                                                              const float* __restrict__ z, int ldz, Pro p,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, Peer pr, ReduceGeom g,
                                                              float* __restrict__ partials) {
  extern __shared__ float red[];  // [R][3][C]
  const float slope = pro_slope(p);
  const bool leaky = p.act == MPGAN_ACT_LEAKY;
  float cz = 0.f, cy = 0.f, ca = 0.f;
  if (pr.coef) { cz = pr.coef[0]; cy = pr.coef[1]; ca = pr.coef[2]; }
  const int tid = threadIdx.x;
  const int q = tid % g.CG, r = tid / g.CG;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int c = q * V;
  const long per = (g.P + g.chunks - 1) / g.chunks;
  const long beg = (long)chunk * per;
  const long end = beg + per < g.P ? beg + per : g.P;
  float acc[3][V];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[a][e] = 0.f;
  if (r < g.R) {
    // per-channel parameters of this thread's columns: loaded once, not per pixel
    const int si = n * p.n_stride + c;
    float sc[V], sh[V], mu[V], is[V], psc[V], psh[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      sc[e] = p.scale[si + e]; sh[e] = p.shift[si + e]; mu[e] = mean[si + e]; is[e] = invstd[si + e];
      psc[e] = pr.coef ? pr.scale[c + e] : 0.f;
      psh[e] = pr.coef ? pr.shift[c + e] : 0.f;
    }
    // four rows per trip, their loads issued together (the sums still run in pixel order)
    for (long pix0 = beg + r; pix0 < end; pix0 += 4L * g.R) {
      float zv[4][V], gv[4][V], zp[4][V];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long pix = pix0 + (long)u * g.R;
        ok[u] = pix < end;
        const long row = (long)n * g.P + (ok[u] ? pix : pix0);
        Vec<V>::load(z + row * ldz + c, zv[u]);
        Vec<V>::load(gr + row * ldg + c, gv[u]);
        if (pr.coef) Vec<V>::load(pr.z + row * pr.ld + c, zp[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float y = zv[u][e] * sc[e] + sh[e];
          const float zh = (zv[u][e] - mu[e]) * is[e];
          float ga = gv[u][e], gy_extra = 0.f;
          if (pr.coef) {
            const float yp = zp[u][e] * psc[e] + psh[e];
            const float ap = (leaky && yp < 0.f) ? yp * slope : yp;
            const float a = (leaky && y < 0.f) ? y * slope : y;
            ga -= ca * sgn(ap - a);
            gy_extra = -cy * sgn(yp - y);
          }
          const bool neg = leaky && y < 0.f;
          const float gy = (neg ? ga * slope : ga) + gy_extra;
          acc[0][e] += gy;
          acc[1][e] += gy * zh;
          acc[2][e] += neg ? ga * y : 0.f;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int e = 0; e < V; ++e) red[(r * 3 + a) * g.C + c + e] = acc[a][e];
  }
  __syncthreads();
  const long rowi = (long)n * g.chunks + chunk;
  fold_rows(red, g.R, 3 * g.C, partials + rowi * 3 * g.C);
  // PReLU-slope gradient: this row's third sums added over the CHANNELS, one scalar per row behind all the rows --
  // the finalize then adds rows scalars instead of rows x C values in its one slope block (C = 128: 28 us -> 4)
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
    for (int j = tid; j < g.C; j += 64) t += red[2 * g.C + j];
    t = wave_sum(t);
    if (tid == 0) partials[(long)gridDim.x * gridDim.y * 3 * g.C + rowi] = t;
  }
  (void)cz;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One wave per statistic: lanes stride over the partial rows (fixed assignment),
// then a fixed butterfly -- deterministic, and no longer a serial 1000-row walk.
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ partials, int N, int chunks,
                                                            int C, int W, long P, int instance,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float momentum,
                                                            float* running_mean, float* running_var,
                                                            int64_t* nbt, float* __restrict__ scale,
                                                            float* __restrict__ shift, float* __restrict__ mean,
                                                            float* __restrict__ invstd) {
  const int total = instance ? N * C : C;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gt >> 6, lane = threadIdx.x & 63;
  if (gt == 0 && nbt) *nbt += 1;
  if (i >= total) return;
  const int c = instance ? i % C : i;
  const int nb = instance ? i / C : 0, ne = instance ? nb + 1 : N;
  const int rows = (ne - nb) * chunks;
  const float* base = partials + (long)nb * chunks * 2 * W;     // rows are [2][W], W >= C channels wide
  double s = 0.0, ss = 0.0;
  for (int r = lane; r < rows; r += 64) {
    const float* row = base + (long)r * 2 * W;
    s += (double)row[c];
    ss += (double)row[W + c];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (lane != 0) return;
  const double cnt = (double)P * (ne - nb);
  const double m = s / cnt;
  double var = ss / cnt - m * m;
  if (var < 0.0) var = 0.0;
  const float istd = (float)(1.0 / sqrt(var + (double)eps));
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float sc = ga * istd;
  scale[i] = sc;
  shift[i] = be - (float)m * sc;
  mean[i] = (float)m;
  invstd[i] = istd;
  if (running_mean && !instance) {
    const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// BatchNorm with >= 256 partial rows: ONE BLOCK per channel (lanes stride over the rows, fixed
// assignment; fixed-order tree in LDS) -- 8 dependent loads per lane for 2048 rows instead of 32.
__global__ __launch_bounds__(256) void norm_finalize_wide_kernel(const float* __restrict__ partials, int rows, int C,
                                                                 int W, double cnt, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps,
                                                                 float momentum, float* running_mean,
                                                                 float* running_var, int64_t* nbt,
                                                                 float* __restrict__ scale, float* __restrict__ shift,
                                                                 float* __restrict__ mean, float* __restrict__ invstd) {
  __shared__ double rs[256], rq[256];
  const int c = blockIdx.x, t = threadIdx.x;
  if (c == 0 && t == 0 && nbt) *nbt += 1;
  double s = 0.0, ss = 0.0;
  for (int r = t; r < rows; r += 1024) {          // four rows per trip: eight independent loads in flight
    float v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * 256;
      const float* row = partials + (long)(rr < rows ? rr : r) * 2 * W;
      const float m = rr < rows ? 1.f : 0.f;
      v[u][0] = row[c] * m;
      v[u][1] = row[W + c] * m;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (double)v[u][0];
      ss += (double)v[u][1];
    }
  }
  rs[t] = s;
  rq[t] = ss;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (t < h) { rs[t] += rs[t + h]; rq[t] += rq[t + h]; }
    __syncthreads();
  }
  if (t != 0) return;
  const double m = rs[0] / cnt;
  double var = rq[0] / cnt - m * m;
  if (var < 0.0) var = 0.0;
  const float istd = (float)(1.0 / sqrt(var + (double)eps));
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float sc = ga * istd;
  scale[c] = sc;
  shift[c] = be - (float)m * sc;
  mean[c] = (float)m;
  invstd[c] = istd;
  if (running_mean) {
    const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// Thousands of partial rows (a conv's fused statistics leave one per pixel tile) are first
// folded to NQ_COMPACT rows with coalesced reads: lane = channel, waves stride over rows.
constexpr int COMPACT_ROWS = 32;
__global__ __launch_bounds__(256) void partials_compact_kernel(const float* __restrict__ partials, int rows, int W,
                                                               float* __restrict__ compact) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);   // column of the [rows][W] matrix (W = NQ*C)
  const int w = threadIdx.x >> 6, g = blockIdx.y;
  float s = 0.f;
  if (c < W)
    for (int r = g * 4 + w; r < rows; r += COMPACT_ROWS * 4) s += partials[(long)r * W + c];
  sh[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < W) compact[(long)g * W + c] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// PReLU-slope gradient: the sum of the third partial over every row and channel, by ONE block.  norm_bwd_reduce
// leaves each row's sum over the channels as one scalar behind the rows (partials[rows * 3 * C + row]), so this
// is a sum of `rows` floats -- no per-channel hand-off, hence no second launch.
__device__ __forceinline__ void slope_grad_block(const float* __restrict__ partials, int rows, int C, float* dslope) {
  __shared__ double red[256];
  const float* __restrict__ sc = partials + (long)rows * 3 * C;
  double s = 0.0;
  const int t = (int)threadIdx.x;
  for (int i = t; i < rows; i += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = i + 256 * u;
      v[u] = e < rows ? sc[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  red[t] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (t < h) red[t] += red[t + h];
    __syncthreads();
  }
  if (t == 0) *dslope += (float)red[0];
}

// BatchNorm with >= 512 partial rows (the 128^2 / 256^2 levels of the generator: 1024-4096 rows; the rows a
// backward-data epilogue leaves: one per tile): ONE BLOCK per channel, thread t takes rows t, t + 256, ... with
// four independent row pairs in flight, fixed-order tree in LDS.  (One wave per channel walked 4096 rows in 16
// dependent trips: 16-75 us for per-channel scalar work.)  The grid's last block is the slope gradient.
__global__ __launch_bounds__(256) void norm_bwd_finalize_wide_kernel(const float* __restrict__ partials, int rows, int C,
                                                                     double cnt, float* dgamma, float* dbeta,
                                                                     float* dslope, float* __restrict__ c1,
                                                                     float* __restrict__ c2) {
  if (dslope && blockIdx.x == gridDim.x - 1) {
    slope_grad_block(partials, rows, C, dslope);
    return;
  }
  __shared__ double r1[256], r2[256];
  const int c = blockIdx.x, t = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = t; r < rows; r += 1024) {
    float v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * 256;
      const float* row = partials + (long)(rr < rows ? rr : r) * 3 * C;
      const float m = rr < rows ? 1.f : 0.f;
      v[u][0] = row[c] * m;
      v[u][1] = row[C + c] * m;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s1 += (double)v[u][0];
      s2 += (double)v[u][1];
    }
  }
  r1[t] = s1;
  r2[t] = s2;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (t < h) { r1[t] += r1[t + h]; r2[t] += r2[t + h]; }
    __syncthreads();
  }
  if (t != 0) return;
  c1[c] = (float)(r1[0] / cnt);
  c2[c] = (float)(r2[0] / cnt);
  if (dgamma) dgamma[c] += (float)r2[0];
  if (dbeta) dbeta[c] += (float)r1[0];
}

// One wave per channel (dgamma, dbeta, c1, c2).
// The grid's LAST block (when dslope != null) adds the PReLU-slope gradient instead (slope_grad_block).
__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(const float* __restrict__ partials, int N, int chunks,
                                                                int C, long P, int instance, float* dgamma,
                                                                float* dbeta, float* dslope,
                                                                float* __restrict__ c1, float* __restrict__ c2) {
  if (dslope && blockIdx.x == gridDim.x - 1) {
    slope_grad_block(partials, N * chunks, C, dslope);
    return;
  }
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gt >> 6, lane = threadIdx.x & 63;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  if (!instance) {
    double s1 = 0.0, s2 = 0.0;
    const int rows = N * chunks;
    for (int r = lane; r < rows; r += 256) {
      float v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * 64;
        const float* row = partials + (long)(rr < rows ? rr : r) * 3 * C;
        const float m = rr < rows ? 1.f : 0.f;
        v[u][0] = row[c] * m;
        v[u][1] = row[C + c] * m;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1 += (double)v[u][0];
        s2 += (double)v[u][1];
      }
    }
    t1 = wave_sum_d(s1);
    t2 = wave_sum_d(s2);
    if (lane == 0) {
      const double cnt = (double)P * N;
      c1[c] = (float)(t1 / cnt);
      c2[c] = (float)(t2 / cnt);
    }
  } else {
    for (int n = 0; n < N; ++n) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = lane; k < chunks; k += 64) {
        const float* row = partials + ((long)n * chunks + k) * 3 * C;
        s1 += (double)row[c];
        s2 += (double)row[C + c];
      }
      s1 = wave_sum_d(s1);
      s2 = wave_sum_d(s2);
      t1 += s1;
      t2 += s2;
      if (lane == 0) {
        c1[n * C + c] = (float)(s1 / (double)P);
        c2[n * C + c] = (float)(s2 / (double)P);
      }
    }
  }
  if (lane == 0) {
    if (dgamma) dgamma[c] += (float)t2;
    if (dbeta) dbeta[c] += (float)t1;
  }
}

// ---------------------------------------------------------------------------
// element-wise kernels: grid-stride over (pixel rows x column groups)
// ---------------------------------------------------------------------------
//   FOLD: the z-side scale / shift are folded from the producer's statistics accumulators at block start
//         (norm_fold.h) instead of being read from a finalize launch's output; block (0, 0) publishes them.
template <int V, bool FOLD = false>
__global__ __launch_bounds__(256) void norm_act_add_kernel(const float* __restrict__ z, int ldz, Pro pz,
                                                           const float* __restrict__ r, int ldr, Pro pr, long rows,
                                                           long P, int C, int tanh_out, float* __restrict__ out,
                                                           int ldo, NormFold fz = NormFold{}) {
  __shared__ long long fwords[FOLD ? 4 * 256 : 1];
  __shared__ float fsc[FOLD ? 256 : 1], fsh[FOLD ? 256 : 1];
  if constexpr (FOLD)
    fold_stats_block(fz, C, fwords, fsc, fsh, (int)threadIdx.x, 256, blockIdx.x == 0 && blockIdx.y == 0);
  const int CG = C / V, R = 256 / CG;
  const int q = threadIdx.x % CG, rr = threadIdx.x / CG;
  if (rr >= R) return;
  const int n = blockIdx.y, c = q * V;
  const float sz = pro_slope(pz), sr = pro_slope(pr);
  float zsc[V], zsh[V], rsc[V], rsh[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    if constexpr (FOLD) {
      zsc[e] = fsc[c + e];
      zsh[e] = fsh[c + e];
    } else {
      zsc[e] = pz.scale ? pz.scale[n * pz.n_stride + c + e] : 1.f;
      zsh[e] = pz.scale ? pz.shift[n * pz.n_stride + c + e] : 0.f;
    }
    rsc[e] = pr.scale ? pr.scale[n * pr.n_stride + c + e] : 1.f;
    rsh[e] = pr.scale ? pr.shift[n * pr.n_stride + c + e] : 0.f;
  }
  // four rows per trip with their loads issued together: these passes are latency-bound per thread
  const long stride = (long)gridDim.x * R;
  for (long pix0 = (long)blockIdx.x * R + rr; pix0 < P; pix0 += 4 * stride) {
    float v[4][V], rv[4][V];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = pix0 + u * stride;
      ok[u] = pix < P;
      const long row = (long)n * P + (ok[u] ? pix : pix0);
      Vec<V>::load(z + row * ldz + c, v[u]);
      if (r) Vec<V>::load(r + row * ldr + c, rv[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const long row = (long)n * P + pix0 + u * stride;
      float o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = pz.scale ? act_apply(v[u][e] * zsc[e] + zsh[e], pz.act, sz) : v[u][e];
      if (r) {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] += pr.scale ? act_apply(rv[u][e] * rsc[e] + rsh[e], pr.act, sr) : rv[u][e];
      }
      if (tanh_out) {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = tanhf(o[e]);
      }
      Vec<V>::store(out + row * ldo + c, o);
    }
  }
  (void)rows;
}

// gr and dz may alias (in-place): every element is read before it is written by the same thread.
// Thread = fixed V columns (per-channel parameters hoisted), rows strided; grid = (row blocks, samples).
template <int V>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* gr, int ldg,
                                                             const float* __restrict__ z, int ldz, Pro p,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ c1,
                                                             const float* __restrict__ c2, Peer pr, long rows, long P,
                                                             int C, float* dz, int lddz) {
  const int CG = C / V, R = 256 / CG;
  const int q = threadIdx.x % CG, r = threadIdx.x / CG;
  if (r >= R) return;
  const int n = blockIdx.y, c = q * V;
  const float slope = pro_slope(p);
  const bool leaky = p.act == MPGAN_ACT_LEAKY;
  float cz = 0.f, cy = 0.f, ca = 0.f;
  if (pr.coef) { cz = pr.coef[0]; cy = pr.coef[1]; ca = pr.coef[2]; }
  const int si = n * p.n_stride + c;
  float sc[V], sh[V], mu[V], is[V], k1[V], k2[V], psc[V], psh[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    sc[e] = p.scale[si + e]; sh[e] = p.shift[si + e]; mu[e] = mean[si + e]; is[e] = invstd[si + e];
    k1[e] = c1[si + e]; k2[e] = c2[si + e];
    psc[e] = pr.coef ? pr.scale[c + e] : 0.f;
    psh[e] = pr.coef ? pr.shift[c + e] : 0.f;
  }
  // four rows per trip, all their loads in front of the first store (gr and dz may alias: a row is read in full
  // before it is written, and no thread touches another thread's rows)
  const long stride = (long)gridDim.x * R;
  for (long pix0 = (long)blockIdx.x * R + r; pix0 < P; pix0 += 4 * stride) {
    float zv[4][V], gv[4][V], zp[4][V];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = pix0 + u * stride;
      ok[u] = pix < P;
      const long row = (long)n * P + (ok[u] ? pix : pix0);
      Vec<V>::load(z + row * ldz + c, zv[u]);
      Vec<V>::load(gr + row * ldg + c, gv[u]);
      if (pr.coef) Vec<V>::load(pr.z + row * pr.ld + c, zp[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const long row = (long)n * P + pix0 + u * stride;
      float o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float y = zv[u][e] * sc[e] + sh[e];
        const float zh = (zv[u][e] - mu[e]) * is[e];
        float ga = gv[u][e], gy_extra = 0.f, dz_extra = 0.f;
        if (pr.coef) {
          const float yp = zp[u][e] * psc[e] + psh[e];
          const float ap = (leaky && yp < 0.f) ? yp * slope : yp;
          const float a = (leaky && y < 0.f) ? y * slope : y;
          ga -= ca * sgn(ap - a);
          gy_extra = -cy * sgn(yp - y);
          dz_extra = -cz * sgn(zp[u][e] - zv[u][e]);
        }
        const float gy = ((leaky && y < 0.f) ? ga * slope : ga) + gy_extra;
        o[e] = sc[e] * (gy - k1[e] - zh * k2[e]) + dz_extra;
      }
      Vec<V>::store(dz + row * lddz + c, o);
    }
  }
  (void)rows;
}

// Perceptual-loss value of one conv+norm+act layer: sums of |z-z'|, |y-y'|, |a-a'| (block partials).
__global__ __launch_bounds__(256) void tap_l1_kernel(const float* __restrict__ za, int lda, Pro pa,
                                                     const float* __restrict__ zb, int ldb, Pro pb, long rows, int C,
                                                     float* __restrict__ partials) {
  __shared__ float sh[3][4];
  const float sa = pro_slope(pa), sb = pro_slope(pb);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float z1 = za[row * lda + c], z2 = zb[row * ldb + c];
    const float y1 = z1 * pa.scale[c] + pa.shift[c], y2 = z2 * pb.scale[c] + pb.shift[c];
    const float a1 = act_apply(y1, pa.act, sa), a2 = act_apply(y2, pb.act, sb);
    s0 += fabsf(z1 - z2);
    s1 += fabsf(y1 - y2);
    s2 += fabsf(a1 - a2);
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

__global__ __launch_bounds__(64) void tap_l1_final_kernel(const float* __restrict__ partials, int nblocks, double inv_numel,
                                                          float* __restrict__ out3) {
  const int q = threadIdx.x;
  if (q >= 3) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += (double)partials[b * 3 + q];
  out3[q] = (float)(s * inv_numel);
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int rows,
                                                              int row_stride, int C, float* __restrict__ out,
                                                              float beta) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gt >> 6, lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += (double)partials[(long)r * row_stride + c];
  s = wave_sum_d(s);
  if (lane == 0) out[c] = beta != 0.f ? beta * out[c] + (float)s : (float)s;
}

template <int V>
__global__ __launch_bounds__(256) void copy_slice_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst,
                                                         int ldd, long rows, int C, int accumulate) {
  const int CG = C / V;
  const long total = rows * CG;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / CG;
    const int c = (int)(i - row * CG) * V;
    float v[V];
    Vec<V>::load(src + row * lds_ + c, v);
    if (accumulate) {
      float d[V];
      Vec<V>::load(dst + row * ldd + c, d);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] += d[e];
    }
    Vec<V>::store(dst + row * ldd + c, v);
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int make_reduce_geom(ReduceGeom& g, int C, long P, bool vec, const char* what) {
  const int V = vec ? 4 : 1;
  g.C = C;
  g.CG = C / V;
  MPGAN_UNSUPPORTED(g.CG > 256, "%s: C=%d too wide", what, C);
  g.R = 256 / g.CG;
  g.P = P;
  g.chunks = stats_chunks_host(P, C);
  return MPGAN_OK;
}

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace mpgan

using namespace mpgan;

extern "C" int32_t mpgan_stats_chunks(int64_t pixels_per_sample, int32_t c) { return stats_chunks_host(pixels_per_sample, c); }

extern "C" int mpgan_channel_stats(const float* z, int32_t ldz, int32_t n, int64_t P, int32_t c, float* partials,
                                   void* stream) {
  MPGAN_CHECK_ARG(z && partials && n > 0 && P > 0 && c > 0 && ldz >= c, "channel_stats: bad argument");
  const bool vec = (c % 4 == 0) && (ldz % 4 == 0) && aligned16(z);
  ReduceGeom g;
  int rc = make_reduce_geom(g, c, P, vec, "channel_stats");
  if (rc) return rc;
  dim3 grid(g.chunks, n);
  const size_t smem = (size_t)g.R * 2 * c * sizeof(float);
  if (vec)
    hipLaunchKernelGGL(channel_stats_kernel<4>, grid, dim3(256), smem, (hipStream_t)stream, z, ldz, g, partials);
  else
    hipLaunchKernelGGL(channel_stats_kernel<1>, grid, dim3(256), smem, (hipStream_t)stream, z, ldz, g, partials);
  return check_launch("channel_stats");
}

extern "C" int mpgan_norm_finalize_strided(const float* partials, int32_t n, int32_t chunks, int32_t c, int32_t cstride,
                                           int64_t P, int32_t instance, const float* gamma, const float* beta,
                                           float eps, float momentum, float* running_mean, float* running_var,
                                           int64_t* nbt, float* scale, float* shift, float* mean, float* invstd,
                                           void* stream) {
  MPGAN_CHECK_ARG(partials && scale && shift && mean && invstd && n > 0 && c > 0 && chunks > 0 && cstride >= c,
                  "norm_finalize: bad argument");
  if (!instance && (long)n * chunks > 16384) {   // (the block-per-channel finalize below takes 16K rows in 16 trips)
    // fold the rows first; the compact rows live in the caller's buffer right after the input rows
    const int rows = n * chunks, W = 2 * cstride;
    float* compact = const_cast<float*>(partials) + (long)rows * W;
    hipLaunchKernelGGL(partials_compact_kernel, dim3((W + 63) / 64, COMPACT_ROWS), dim3(256), 0, (hipStream_t)stream,
                       partials, rows, W, compact);
    partials = compact;
    P = P * n;   // the element count per channel stays n*P
    n = 1;
    chunks = COMPACT_ROWS;
  }
  if (!instance && (long)n * chunks >= 256) {
    hipLaunchKernelGGL(norm_finalize_wide_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partials, n * chunks, c,
                       cstride, (double)P * n, gamma, beta, eps, momentum, running_mean, running_var, nbt, scale, shift,
                       mean, invstd);
    return check_launch("norm_finalize");
  }
  const int total = instance ? n * c : c;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials, n,
                     chunks, c, cstride, (long)P, instance, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                     scale, shift, mean, invstd);
  return check_launch("norm_finalize");
}

extern "C" int mpgan_norm_finalize(const float* partials, int32_t n, int32_t chunks, int32_t c, int64_t P,
                                   int32_t instance, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, int64_t* nbt,
                                   float* scale, float* shift, float* mean, float* invstd, void* stream) {
  return mpgan_norm_finalize_strided(partials, n, chunks, c, c, P, instance, gamma, beta, eps, momentum, running_mean,
                                     running_var, nbt, scale, shift, mean, invstd, stream);
}

extern "C" int mpgan_norm_act_add(const float* z, int32_t ldz, const mpgan_prologue* pz, const float* r, int32_t ldr,
                                  const mpgan_prologue* pr, int32_t n, int64_t P, int32_t c, int32_t tanh_out,
                                  float* out, int32_t ldo, void* stream) {
  MPGAN_CHECK_ARG(z && out && n > 0 && P > 0 && c > 0 && ldz >= c && ldo >= c && (!r || ldr >= c),
                  "norm_act_add: bad argument");
  const bool vec = (c % 4 == 0) && (ldz % 4 == 0) && (ldo % 4 == 0) && aligned16(z) && aligned16(out) &&
                   (!r || ((ldr % 4 == 0) && aligned16(r)));
  const long rows = (long)n * P;
  Pro a = make_pro(pz), b = make_pro(pr);
  const int CGa = vec ? c / 4 : c;
  MPGAN_UNSUPPORTED(CGa > 256, "norm_act_add: C=%d too wide", c);
  const int Ra = 256 / CGa;
  long gxa = (P + 4L * Ra - 1) / (4L * Ra);          // four rows per thread (the kernel batches their loads)
  const long capa = 4096 / n > 1 ? 4096 / n : 1;
  if (gxa > capa) gxa = capa;
  dim3 grida((unsigned)gxa, (unsigned)n);
  if (vec)
    hipLaunchKernelGGL((norm_act_add_kernel<4, false>), grida, dim3(256), 0, (hipStream_t)stream, z, ldz, a, r, ldr, b,
                       rows, (long)P, c, tanh_out, out, ldo, NormFold{});
  else
    hipLaunchKernelGGL((norm_act_add_kernel<1, false>), grida, dim3(256), 0, (hipStream_t)stream, z, ldz, a, r, ldr, b,
                       rows, (long)P, c, tanh_out, out, ldo, NormFold{});
  return check_launch("norm_act_add");
}

extern "C" int mpgan_norm_act_add_fold(const float* z, int32_t ldz, const mpgan_prologue* pz,
                                       const mpgan_norm_fold* fold_z, const float* r, int32_t ldr,
                                       const mpgan_prologue* pr, int32_t n, int64_t P, int32_t c, int32_t tanh_out,
                                       float* out, int32_t ldo, void* stream) {
  NormFold fz = make_fold(fold_z);
  if (!fz.acc) return mpgan_norm_act_add(z, ldz, pz, r, ldr, pr, n, P, c, tanh_out, out, ldo, stream);
  MPGAN_CHECK_ARG(z && out && pz && n > 0 && P > 0 && c > 0 && ldz >= c && ldo >= c && (!r || ldr >= c) &&
                      fz.rep > 0 && fz.cstride >= c && fz.cnt > 0 && fz.scale && fz.shift && fz.mean && fz.invstd,
                  "norm_act_add_fold: bad argument");
  MPGAN_UNSUPPORTED(c > 256, "norm_act_add_fold: C=%d too wide", c);
  const bool vec = (c % 4 == 0) && (ldz % 4 == 0) && (ldo % 4 == 0) && aligned16(z) && aligned16(out) &&
                   (!r || ((ldr % 4 == 0) && aligned16(r)));
  const long rows = (long)n * P;
  Pro a = make_pro(pz), b = make_pro(pr);
  a.scale = fz.scale;            // "has a prologue": the values come from the fold
  a.shift = fz.shift;
  a.n_stride = 0;
  const int CGa = vec ? c / 4 : c;
  const int Ra = 256 / CGa;
  long gxa = (P + 4L * Ra - 1) / (4L * Ra);          // four rows per thread (the kernel batches their loads)
  const long capa = 4096 / n > 1 ? 4096 / n : 1;
  if (gxa > capa) gxa = capa;
  dim3 grida((unsigned)gxa, (unsigned)n);
  if (vec)
    hipLaunchKernelGGL((norm_act_add_kernel<4, true>), grida, dim3(256), 0, (hipStream_t)stream, z, ldz, a, r, ldr, b,
                       rows, (long)P, c, tanh_out, out, ldo, fz);
  else
    hipLaunchKernelGGL((norm_act_add_kernel<1, true>), grida, dim3(256), 0, (hipStream_t)stream, z, ldz, a, r, ldr, b,
                       rows, (long)P, c, tanh_out, out, ldo, fz);
  return check_launch("norm_act_add_fold");
}

extern "C" int mpgan_norm_bwd_reduce(const float* g, int32_t ldg, const float* z, int32_t ldz,
                                     const mpgan_prologue* p, const float* mean, const float* invstd,
                                     const mpgan_peer_taps* peer, int32_t n, int64_t P, int32_t c, float* partials,
                                     void* stream) {
  MPGAN_CHECK_ARG(g && z && p && p->scale && mean && invstd && partials && n > 0 && P > 0 && c > 0 && ldg >= c &&
                      ldz >= c,
                  "norm_bwd_reduce: bad argument");
  const bool vec = (c % 4 == 0) && (ldz % 4 == 0) && (ldg % 4 == 0) && aligned16(z) && aligned16(g);
  ReduceGeom geo;
  int rc = make_reduce_geom(geo, c, P, vec, "norm_bwd_reduce");
  if (rc) return rc;
  dim3 grid(geo.chunks, n);
  const size_t smem = (size_t)geo.R * 3 * c * sizeof(float);
  Pro pp = make_pro(p);
  Peer pe = make_peer(peer);
  MPGAN_UNSUPPORTED(pe.coef && p->n_stride != 0, "norm_bwd_reduce: peer taps are defined for BatchNorm layers only");
  if (vec)
    hipLaunchKernelGGL(norm_bwd_reduce_kernel<4>, grid, dim3(256), smem, (hipStream_t)stream, g, ldg, z, ldz, pp, mean,
                       invstd, pe, geo, partials);
  else
    hipLaunchKernelGGL(norm_bwd_reduce_kernel<1>, grid, dim3(256), smem, (hipStream_t)stream, g, ldg, z, ldz, pp, mean,
                       invstd, pe, geo, partials);
  return check_launch("norm_bwd_reduce");
}

extern "C" int mpgan_norm_bwd_finalize(const float* partials, int32_t n, int32_t chunks, int32_t c, int64_t P,
                                       int32_t instance, float* dgamma, float* dbeta, float* dslope, float* c1,
                                       float* c2, void* stream) {
  MPGAN_CHECK_ARG(partials && c1 && c2 && n > 0 && c > 0 && chunks > 0, "norm_bwd_finalize: bad argument");
  // per-channel slope terms go to the tail of the partials buffer (>= c floats past the partial rows)
  // (Measured and dropped: one 1024-thread block finalizing every channel plus the slope sum: 23-31 us
  //  against 10 -- one CU's address unit serialises its 16 waves' one-line-per-lane row reads.)
  const long rows = (long)n * chunks;
  if (!instance && rows >= 512 && rows < (1L << 31)) {   // many rows: a block per channel (see the kernel)
    hipLaunchKernelGGL(norm_bwd_finalize_wide_kernel, dim3(c + (dslope ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                       partials, (int)rows, c, (double)P * n, dgamma, dbeta, dslope, c1, c2);
    return check_launch("norm_bwd_finalize_wide");
  }
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((c + 3) / 4 + (dslope ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                     partials, n, chunks, c, (long)P, instance, dgamma, dbeta, dslope, c1, c2);
  return check_launch("norm_bwd_finalize");
}

extern "C" int mpgan_norm_bwd_apply(const float* g, int32_t ldg, const float* z, int32_t ldz, const mpgan_prologue* p,
                                    const float* mean, const float* invstd, const float* c1, const float* c2,
                                    const mpgan_peer_taps* peer, int32_t n, int64_t P, int32_t c, float* dz,
                                    int32_t lddz, void* stream) {
  MPGAN_CHECK_ARG(g && z && p && p->scale && mean && invstd && c1 && c2 && dz && n > 0 && P > 0 && c > 0 &&
                      ldg >= c && ldz >= c && lddz >= c,
                  "norm_bwd_apply: bad argument");
  const bool vec = (c % 4 == 0) && (ldz % 4 == 0) && (ldg % 4 == 0) && (lddz % 4 == 0) && aligned16(z) &&
                   aligned16(g) && aligned16(dz);
  const long rows = (long)n * P;
  Pro pp = make_pro(p);
  Peer pe = make_peer(peer);
  const int CGa = vec ? c / 4 : c;
  MPGAN_UNSUPPORTED(CGa > 256, "norm_bwd_apply: C=%d too wide", c);
  const int Ra = 256 / CGa;
  long gxa = (P + 4L * Ra - 1) / (4L * Ra);          // four rows per thread (the kernel batches their loads)
  const long capa = 4096 / n > 1 ? 4096 / n : 1;
  if (gxa > capa) gxa = capa;
  dim3 grida((unsigned)gxa, (unsigned)n);
  if (vec)
    hipLaunchKernelGGL(norm_bwd_apply_kernel<4>, grida, dim3(256), 0, (hipStream_t)stream, g, ldg, z, ldz, pp, mean,
                       invstd, c1, c2, pe, rows, (long)P, c, dz, lddz);
  else
    hipLaunchKernelGGL(norm_bwd_apply_kernel<1>, grida, dim3(256), 0, (hipStream_t)stream, g, ldg, z, ldz, pp, mean,
                       invstd, c1, c2, pe, rows, (long)P, c, dz, lddz);
  return check_launch("norm_bwd_apply");
}

extern "C" int mpgan_reduce_partials(const float* partials, int32_t rows, int32_t row_stride, int32_t c, float* out,
                                     float beta, void* stream) {
  MPGAN_CHECK_ARG(partials && out && rows > 0 && c > 0 && row_stride >= c, "reduce_partials: bad argument");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials, rows,
                     row_stride, c, out, beta);
  return check_launch("reduce_partials");
}

extern "C" int mpgan_copy_slice(const float* src, int32_t lds_, float* dst, int32_t ldd, int64_t pixels, int32_t c,
                                int32_t accumulate, void* stream) {
  MPGAN_CHECK_ARG(src && dst && pixels > 0 && c > 0 && lds_ >= c && ldd >= c, "copy_slice: bad argument");
  const bool vec = (c % 4 == 0) && (lds_ % 4 == 0) && (ldd % 4 == 0) && aligned16(src) && aligned16(dst);
  if (vec)
    hipLaunchKernelGGL(copy_slice_kernel<4>, dim3(ew_blocks(pixels * (c / 4))), dim3(256), 0, (hipStream_t)stream, src,
                       lds_, dst, ldd, (long)pixels, c, accumulate);
  else
    hipLaunchKernelGGL(copy_slice_kernel<1>, dim3(ew_blocks(pixels * c)), dim3(256), 0, (hipStream_t)stream, src, lds_,
                       dst, ldd, (long)pixels, c, accumulate);
  return check_launch("copy_slice");
}

// eval-mode BatchNorm: scale/shift from the running statistics (inferrence.py:97-110 runs the
// generator under .eval()); mean/invstd are filled too so the vectors are complete.
__global__ void norm_from_running_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ rm, const float* __restrict__ rv, float eps, int c,
                                         float* __restrict__ scale, float* __restrict__ shift,
                                         float* __restrict__ mean, float* __restrict__ invstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float istd = 1.f / sqrtf(rv[i] + eps);
  const float sc = (gamma ? gamma[i] : 1.f) * istd;
  scale[i] = sc;
  shift[i] = (beta ? beta[i] : 0.f) - rm[i] * sc;
  mean[i] = rm[i];
  invstd[i] = istd;
}

extern "C" int mpgan_norm_from_running(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, int32_t c, float* scale, float* shift,
                                       float* mean, float* invstd, void* stream) {
  MPGAN_CHECK_ARG(running_mean && running_var && scale && shift && mean && invstd && c > 0,
                  "norm_from_running: bad argument");
  hipLaunchKernelGGL(norm_from_running_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, c, scale, shift, mean, invstd);
  return check_launch("norm_from_running");
}

// All eval-mode norm layers of a network in ONE launch: block b serves table row b =
// {gamma, beta, running_mean, running_var, scale, shift, mean, invstd (device addresses), C, eps bits}.
__global__ __launch_bounds__(256) void norm_from_running_multi_kernel(const long long* __restrict__ table) {
  const long long* e = table + 10 * (long)blockIdx.x;
  const float* gamma = reinterpret_cast<const float*>(e[0]);
  const float* beta = reinterpret_cast<const float*>(e[1]);
  const float* rm = reinterpret_cast<const float*>(e[2]);
  const float* rv = reinterpret_cast<const float*>(e[3]);
  float* scale = reinterpret_cast<float*>(e[4]);
  float* shift = reinterpret_cast<float*>(e[5]);
  float* mean = reinterpret_cast<float*>(e[6]);
  float* invstd = reinterpret_cast<float*>(e[7]);
  const int c = (int)e[8];
  const float eps = __uint_as_float((unsigned)e[9]);
  for (int i = threadIdx.x; i < c; i += 256) {
    const float istd = 1.f / sqrtf(rv[i] + eps);
    const float sc = (gamma ? gamma[i] : 1.f) * istd;
    scale[i] = sc;
    shift[i] = (beta ? beta[i] : 0.f) - rm[i] * sc;
    mean[i] = rm[i];
    invstd[i] = istd;
  }
}

extern "C" int mpgan_norm_from_running_multi(const int64_t* table, int32_t n_layers, void* stream) {
  MPGAN_CHECK_ARG(table && n_layers > 0, "norm_from_running_multi: bad argument");
  hipLaunchKernelGGL(norm_from_running_multi_kernel, dim3((unsigned)n_layers), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(table));
  return check_launch("norm_from_running_multi");
}

// Epilogue-activation vectors of every conv of an eval-mode plan in ONE launch (mpgan_conv_forward_act): block b serves
// table row b = {gamma, beta, running_mean, running_var, conv bias, PReLU weight (one element), scale, shift, slope
// (device addresses), c_norm, c_total, eps bits}.  Channels < c_norm carry the running-statistics BatchNorm + PReLU of
// the layer: scale = gamma / sqrt(var + eps), shift = beta - mean * scale + bias * scale, slope = the PReLU weight
// (1 when the layer has none); channels c_norm .. c_total - 1 (the residual half of a fused unit0 || residual conv, or a
// conv without a norm layer) stay linear: scale 1, shift = bias, slope 1.
__global__ __launch_bounds__(256) void epi_vectors_multi_kernel(const long long* __restrict__ table) {
  const long long* e = table + 12 * (long)blockIdx.x;
  const float* gamma = reinterpret_cast<const float*>(e[0]);
  const float* beta = reinterpret_cast<const float*>(e[1]);
  const float* rm = reinterpret_cast<const float*>(e[2]);
  const float* rv = reinterpret_cast<const float*>(e[3]);
  const float* bias = reinterpret_cast<const float*>(e[4]);
  const float* alpha = reinterpret_cast<const float*>(e[5]);
  float* scale = reinterpret_cast<float*>(e[6]);
  float* shift = reinterpret_cast<float*>(e[7]);
  float* slope = reinterpret_cast<float*>(e[8]);
  const int c_norm = (int)e[9], c_total = (int)e[10];
  const float eps = __uint_as_float((unsigned)e[11]);
  const float a = alpha ? alpha[0] : 1.f;
  for (int i = threadIdx.x; i < c_total; i += 256) {
    const float b = bias ? bias[i] : 0.f;
    if (i < c_norm) {
      const float sc = (gamma ? gamma[i] : 1.f) / sqrtf(rv[i] + eps);
      scale[i] = sc;
      shift[i] = fmaf(b - rm[i], sc, beta ? beta[i] : 0.f);
      slope[i] = a;
    } else {
      scale[i] = 1.f;
      shift[i] = b;
      slope[i] = 1.f;
    }
  }
}

extern "C" int mpgan_epi_vectors_multi(const int64_t* table, int32_t n_layers, void* stream) {
  MPGAN_CHECK_ARG(table && n_layers > 0, "epi_vectors_multi: bad argument");
  hipLaunchKernelGGL(epi_vectors_multi_kernel, dim3((unsigned)n_layers), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(table));
  return check_launch("epi_vectors_multi");
}

extern "C" int32_t mpgan_tap_l1_partials(void) { return 3 * 1024; }

extern "C" int mpgan_tap_l1(const float* za, int32_t lda, const mpgan_prologue* pa, const float* zb, int32_t ldb,
                            const mpgan_prologue* pb, int64_t rows, int32_t c, float* partials, float* out3,
                            void* stream) {
  MPGAN_CHECK_ARG(za && zb && pa && pb && pa->scale && pb->scale && partials && out3 && rows > 0 && c > 0,
                  "tap_l1: bad argument");
  MPGAN_UNSUPPORTED(pa->n_stride != 0 || pb->n_stride != 0, "tap_l1: BatchNorm layers only");
  long blocks = (rows * c + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(tap_l1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, za, lda, make_pro(pa), zb,
                     ldb, make_pro(pb), (long)rows, c, partials);
  hipLaunchKernelGGL(tap_l1_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, (int)blocks,
                     1.0 / ((double)rows * c), out3);
  return check_launch("tap_l1");
}
