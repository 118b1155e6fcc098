// Implicit-GEMM convolution on the fp32 matrix cores of gfx950.
//
// One kernel family serves every dense contraction of the GAN step:
//   * ConvNd forward                      (forward gather:   i = o*s - p + k)
//   * ConvTransposeNd forward             (phase-decomposed: i = (o + p - k)/s)
//   * backward-data of ConvNd             (= transposed gather over dy)
//   * backward-data of ConvTransposeNd    (= forward gather over dy)
// GEMM view: rows m = output pixels (of one phase), cols = output channels,
// K = (tap, input channel).  A rows are gathered straight from the
// channels-last activation (128 B per pixel and K-step, coalesced), with the
// producer's BatchNorm/InstanceNorm + PReLU/LeakyReLU applied on load;
// B rows come from the packed [Cout][tap][Cin] weights.
//
// Tile: 128 pixels x BN channels x 32 K per step, 256 threads = 4 waves, each
// wave owns TMxTN 32x32 accumulators of v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 cycles per issue per SIMD, so one wave per SIMD already paces the pipe).
// LDS rows are [row][32 K + 4 pad] floats: each lane fetches FOUR consecutive K
// of its row with one ds_read_b128 (conflict-free at pitch 36), and the two
// half-waves take different K quads so a quad feeds four MFMAs.
#include "mpgan_common.h"

namespace mpgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Phase {
  int Mz, My, Mx;     // extents of the m-grid of this phase
  int oz, oy, ox;     // output coordinate = m*ostride + o?
  int nz, ny, nx;     // taps per dimension
  int kz0, ky0, kx0;  // kernel index of tap j: k0 + kstep*j
  int dz0, dy0, dx0;  // input offset of tap j:  d0 + dstep*j
};

struct GatherConv {
  const float* in;
  const float* wp;
  float* out;
  const float* bias;
  const float* resid;
  Pro pro;
  int N, Di, Hi, Wi, Cin, ldi;
  int Do, Ho, Wo, Cout, ldo, ldr;
  int Kz, Ky, Kx;
  int ostride[3], istride[3], kstep[3], dstep[3];
  int nphase, tanh_out;
  Phase ph[8];
};

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int PITCH = BK + 4;

struct TapInfo {
  int tapflat, dz, dy, dx, ci;
  bool valid;
};

__device__ __forceinline__ TapInfo decode_k(const GatherConv& p, const Phase& ph, int kidx, int Kp) {
  TapInfo t;
  t.valid = kidx < Kp;
  int tap = kidx / p.Cin;
  t.ci = kidx - tap * p.Cin;
  int jx = tap % ph.nx;
  int tq = tap / ph.nx;
  int jy = tq % ph.ny;
  int jz = tq / ph.ny;
  int kz = ph.kz0 + p.kstep[0] * jz, ky = ph.ky0 + p.kstep[1] * jy, kx = ph.kx0 + p.kstep[2] * jx;
  t.tapflat = (kz * p.Ky + ky) * p.Kx + kx;
  t.dz = ph.dz0 + p.dstep[0] * jz;
  t.dy = ph.dy0 + p.dstep[1] * jy;
  t.dx = ph.dx0 + p.dstep[2] * jx;
  return t;
}

template <int BN, int TM, int TN, int WN, bool SCALAR>
__global__ __launch_bounds__(256) void gather_conv_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int STAGE = (BM + BN) * PITCH;
  constexpr int BROWS = BN / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const Phase& ph = p.ph[blockIdx.z];
  const long Mtot = (long)p.N * ph.Mz * ph.My * ph.Mx;
  const long m0 = (long)blockIdx.x * BM;
  if (m0 >= Mtot) return;
  const int n0 = blockIdx.y * BN;
  const int ntaps = ph.nz * ph.ny * ph.nx;
  const int Kp = ntaps * p.Cin;
  const int nk = (Kp + BK - 1) / BK;
  const long Ktot = (long)p.Kz * p.Ky * p.Kx * p.Cin;
  const float slope = pro_slope(p.pro);

  // ---- per-thread load assignment: K-chunk column cc, rows r0 + 32*i ----
  const int cc = tid & 7, r0 = tid >> 3;
  int rn[4], rz[4], ry[4], rx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long m = m0 + r0 + 32 * i;
    if (m < Mtot) {
      int mx = (int)(m % ph.Mx);
      long q = m / ph.Mx;
      int my = (int)(q % ph.My);
      q /= ph.My;
      int mz = (int)(q % ph.Mz);
      rn[i] = (int)(q / ph.Mz);
      rz[i] = mz * p.istride[0];
      ry[i] = my * p.istride[1];
      rx[i] = mx * p.istride[2];
    } else {
      rn[i] = 0;
      rz[i] = ry[i] = rx[i] = -(1 << 28);
    }
  }

  float4 ra[4], rb[BROWS];

  auto load_a_elem = [&](int i, const TapInfo& t) -> float {
    int iz = rz[i] + t.dz, iy = ry[i] + t.dy, ix = rx[i] + t.dx;
    bool ok = t.valid && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi &&
              (unsigned)ix < (unsigned)p.Wi;
    if (!ok) return 0.f;
    long pix = (((long)rn[i] * p.Di + iz) * p.Hi + iy) * p.Wi + ix;
    float v = p.in[pix * p.ldi + t.ci];
    if (p.pro.scale) {
      int si = rn[i] * p.pro.n_stride + t.ci;
      v = act_apply(v * p.pro.scale[si] + p.pro.shift[si], p.pro.act, slope);
    }
    return v;
  };

  auto global_load = [&](int kt) {
    const int kidx = kt * BK + cc * 4;
    if constexpr (!SCALAR) {
      const TapInfo t = decode_k(p, ph, kidx, Kp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iz = rz[i] + t.dz, iy = ry[i] + t.dy, ix = rx[i] + t.dx;
        bool ok = t.valid && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi &&
                  (unsigned)ix < (unsigned)p.Wi;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          long pix = (((long)rn[i] * p.Di + iz) * p.Hi + iy) * p.Wi + ix;
          v = *reinterpret_cast<const float4*>(p.in + pix * p.ldi + t.ci);
          if (p.pro.scale) {
            int si = rn[i] * p.pro.n_stride + t.ci;
            float4 sc = *reinterpret_cast<const float4*>(p.pro.scale + si);
            float4 sh = *reinterpret_cast<const float4*>(p.pro.shift + si);
            v.x = act_apply(v.x * sc.x + sh.x, p.pro.act, slope);
            v.y = act_apply(v.y * sc.y + sh.y, p.pro.act, slope);
            v.z = act_apply(v.z * sc.z + sh.z, p.pro.act, slope);
            v.w = act_apply(v.w * sc.w + sh.w, p.pro.act, slope);
          }
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        int co = n0 + r0 + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t.valid && co < p.Cout)
          v = *reinterpret_cast<const float4*>(p.wp + (long)co * Ktot + (long)t.tapflat * p.Cin + t.ci);
        rb[i] = v;
      }
    } else {
      TapInfo t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = decode_k(p, ph, kidx + e, Kp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i].x = load_a_elem(i, t[0]);
        ra[i].y = load_a_elem(i, t[1]);
        ra[i].z = load_a_elem(i, t[2]);
        ra[i].w = load_a_elem(i, t[3]);
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        int co = n0 + r0 + 32 * i;
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          w[e] = (t[e].valid && co < p.Cout)
                     ? p.wp[(long)co * Ktot + (long)t[e].tapflat * p.Cin + t[e].ci]
                     : 0.f;
        rb[i] = make_float4(w[0], w[1], w[2], w[3]);
      }
    }
  };

  auto lds_store = [&](int buf) {
    float* As = lds + buf * STAGE;
    float* Bs = As + BM * PITCH;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(As + (r0 + 32 * i) * PITCH + cc * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<float4*>(Bs + (r0 + 32 * i) * PITCH + cc * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (nk > 0) {
    global_load(0);
    lds_store(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) global_load(kt + 1);
    const float* As = lds + cur * STAGE + (wm * TM * 32 + li) * PITCH + 4 * lh;
    const float* Bs = lds + cur * STAGE + BM * PITCH + (wn * TN * 32 + li) * PITCH + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        a[tm] = *reinterpret_cast<const float4*>(As + tm * 32 * PITCH + 8 * g);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        b[tn] = *reinterpret_cast<const float4*>(Bs + tn * 32 * PITCH + 8 * g);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) lds_store(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: row -> output pixel map through LDS, then bias/resid/tanh ----
  int* rowpix = reinterpret_cast<int*>(lds);
  if (tid < BM) {
    long m = m0 + tid;
    int pix = -1;
    if (m < Mtot) {
      int mx = (int)(m % ph.Mx);
      long q = m / ph.Mx;
      int my = (int)(q % ph.My);
      q /= ph.My;
      int mz = (int)(q % ph.Mz);
      int n = (int)(q / ph.Mz);
      int oz = mz * p.ostride[0] + ph.oz, oy = my * p.ostride[1] + ph.oy, ox = mx * p.ostride[2] + ph.ox;
      if (oz < p.Do && oy < p.Ho && ox < p.Wo) pix = ((n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    }
    rowpix[tid] = pix;
  }
  __syncthreads();
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int co = n0 + (wn * TN + tn) * 32 + li;
      if (co >= p.Cout) continue;
      const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int pix = rowpix[row];
        if (pix < 0) continue;
        float v = acc[tm][tn][r] + bv;
        if (p.resid) v += p.resid[(long)pix * p.ldr + co];
        if (p.tanh_out) v = tanhf(v);
        p.out[(long)pix * p.ldo + co] = v;
      }
    }
}

template <int BN, int TM, int TN, int WN, bool SCALAR>
static int launch_variant(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_kernel<BN, TM, TN, WN, SCALAR>;
  constexpr int smem = 2 * (BM + BN) * PITCH * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_conv: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)((maxM + BM - 1) / BM), (unsigned)((p.Cout + BN - 1) / BN), (unsigned)p.nphase);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  return check_launch("gather_conv");
}

static int launch_gather(const GatherConv& p, hipStream_t st) {
  long maxM = 0;
  for (int i = 0; i < p.nphase; ++i) {
    long m = (long)p.N * p.ph[i].Mz * p.ph[i].My * p.ph[i].Mx;
    if (m > maxM) maxM = m;
  }
  if (maxM == 0) return MPGAN_OK;
  MPGAN_CHECK_ARG((long)p.N * p.Do * p.Ho * p.Wo < (1L << 31), "gather_conv: too many output pixels");
  const bool vec = (p.Cin % 4 == 0) && (p.ldi % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(p.wp) & 15) == 0) &&
                   (!p.pro.scale || (((reinterpret_cast<uintptr_t>(p.pro.scale) |
                                       reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0 &&
                                     p.pro.n_stride % 4 == 0));
  if (vec) {
    if (p.Cout > 64) return launch_variant<128, 2, 2, 2, false>(p, maxM, st);
    if (p.Cout > 32) return launch_variant<64, 1, 2, 1, false>(p, maxM, st);
    return launch_variant<32, 1, 1, 1, false>(p, maxM, st);
  }
  if (p.Cout > 64) return launch_variant<128, 2, 2, 2, true>(p, maxM, st);
  if (p.Cout > 32) return launch_variant<64, 1, 2, 1, true>(p, maxM, st);
  return launch_variant<32, 1, 1, 1, true>(p, maxM, st);
}

// ---- host-side geometry builders -------------------------------------------
static int check_geom(const mpgan_conv_geom* g) {
  MPGAN_CHECK_ARG(g != nullptr, "conv: null geometry");
  MPGAN_CHECK_ARG(g->n > 0 && g->cin > 0 && g->cout > 0, "conv: bad n/cin/cout");
  for (int d = 0; d < 3; ++d) {
    MPGAN_CHECK_ARG(g->in_dhw[d] > 0 && g->out_dhw[d] > 0 && g->k[d] > 0 && g->stride[d] > 0 && g->pad[d] >= 0,
                    "conv: bad spatial geometry in dim %d", d);
    MPGAN_UNSUPPORTED(g->stride[d] > 2, "conv: stride > 2 unsupported");
    if (!g->transposed) {
      int o = (g->in_dhw[d] + 2 * g->pad[d] - g->k[d]) / g->stride[d] + 1;
      MPGAN_CHECK_ARG(o == g->out_dhw[d], "conv: out_dhw[%d]=%d does not match geometry (%d)", d, g->out_dhw[d], o);
    } else {
      int lo = (g->in_dhw[d] - 1) * g->stride[d] - 2 * g->pad[d] + g->k[d];
      MPGAN_CHECK_ARG(g->out_dhw[d] >= lo && g->out_dhw[d] < lo + g->stride[d],
                      "convT: out_dhw[%d]=%d outside [%d,%d)", d, g->out_dhw[d], lo, lo + g->stride[d]);
    }
  }
  return MPGAN_OK;
}

// forward-type gather: produced[o] = sum_k gathered[o*s - p + k] * W[k]
static void build_forward(GatherConv& p, int n, const int32_t* gath_dhw, int cg, const int32_t* prod_dhw, int cp,
                          const int32_t* k, const int32_t* s, const int32_t* pad) {
  p.N = n; p.Di = gath_dhw[0]; p.Hi = gath_dhw[1]; p.Wi = gath_dhw[2]; p.Cin = cg;
  p.Do = prod_dhw[0]; p.Ho = prod_dhw[1]; p.Wo = prod_dhw[2]; p.Cout = cp;
  p.Kz = k[0]; p.Ky = k[1]; p.Kx = k[2];
  for (int d = 0; d < 3; ++d) { p.ostride[d] = 1; p.istride[d] = s[d]; p.kstep[d] = 1; p.dstep[d] = 1; }
  p.nphase = 1;
  Phase& ph = p.ph[0];
  ph.Mz = prod_dhw[0]; ph.My = prod_dhw[1]; ph.Mx = prod_dhw[2];
  ph.oz = ph.oy = ph.ox = 0;
  ph.nz = k[0]; ph.ny = k[1]; ph.nx = k[2];
  ph.kz0 = ph.ky0 = ph.kx0 = 0;
  ph.dz0 = -pad[0]; ph.dy0 = -pad[1]; ph.dx0 = -pad[2];
}

// transposed-type gather: produced[o] = sum_k gathered[(o + p - k)/s] * W[k]
static void build_transposed(GatherConv& p, int n, const int32_t* gath_dhw, int cg, const int32_t* prod_dhw, int cp,
                             const int32_t* k, const int32_t* s, const int32_t* pad) {
  p.N = n; p.Di = gath_dhw[0]; p.Hi = gath_dhw[1]; p.Wi = gath_dhw[2]; p.Cin = cg;
  p.Do = prod_dhw[0]; p.Ho = prod_dhw[1]; p.Wo = prod_dhw[2]; p.Cout = cp;
  p.Kz = k[0]; p.Ky = k[1]; p.Kx = k[2];
  for (int d = 0; d < 3; ++d) { p.ostride[d] = s[d]; p.istride[d] = 1; p.kstep[d] = s[d]; p.dstep[d] = -1; }
  int np = 0;
  for (int pz = 0; pz < s[0]; ++pz)
    for (int py = 0; py < s[1]; ++py)
      for (int px = 0; px < s[2]; ++px) {
        Phase& ph = p.ph[np++];
        const int phs[3] = {pz, py, px};
        int M[3], nj[3], k0[3], d0[3];
        for (int d = 0; d < 3; ++d) {
          int r = (phs[d] + pad[d]) % s[d];
          nj[d] = r < k[d] ? (k[d] - r + s[d] - 1) / s[d] : 0;
          k0[d] = r;
          d0[d] = (phs[d] + pad[d] - r) / s[d];
          M[d] = prod_dhw[d] > phs[d] ? (prod_dhw[d] - phs[d] + s[d] - 1) / s[d] : 0;
        }
        ph.Mz = M[0]; ph.My = M[1]; ph.Mx = M[2];
        ph.oz = pz; ph.oy = py; ph.ox = px;
        ph.nz = nj[0]; ph.ny = nj[1]; ph.nx = nj[2];
        ph.kz0 = k0[0]; ph.ky0 = k0[1]; ph.kx0 = k0[2];
        ph.dz0 = d0[0]; ph.dy0 = d0[1]; ph.dx0 = d0[2];
        if (nj[0] == 0 || nj[1] == 0 || nj[2] == 0) { ph.nz = 0; ph.ny = 1; ph.nx = 1; }  // no taps: bias only
      }
  p.nphase = np;
}

}  // namespace mpgan

using namespace mpgan;

extern "C" int mpgan_conv_forward(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                                  const float* bias, const mpgan_prologue* pro, const float* resid, int32_t ldr,
                                  int32_t tanh_out, float* y, int32_t ldy, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(x && w_packed && y, "conv_forward: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && ldy >= g->cout && (!resid || ldr >= g->cout), "conv_forward: bad pitch");
  GatherConv p{};
  p.in = x; p.wp = w_packed; p.out = y; p.bias = bias; p.resid = resid;
  p.pro = make_pro(pro);
  p.ldi = ldx; p.ldo = ldy; p.ldr = ldr; p.tanh_out = tanh_out;
  if (!g->transposed)
    build_forward(p, g->n, g->in_dhw, g->cin, g->out_dhw, g->cout, g->k, g->stride, g->pad);
  else
    build_transposed(p, g->n, g->in_dhw, g->cin, g->out_dhw, g->cout, g->k, g->stride, g->pad);
  return launch_gather(p, (hipStream_t)stream);
}

extern "C" int mpgan_conv_backward_data(const mpgan_conv_geom* g, const float* dy, int32_t lddy,
                                        const float* w_packed_bwd, const float* resid, int32_t ldr, float* dx,
                                        int32_t lddx, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(dy && w_packed_bwd && dx, "conv_backward_data: null pointer");
  MPGAN_CHECK_ARG(lddy >= g->cout && lddx >= g->cin && (!resid || ldr >= g->cin), "conv_backward_data: bad pitch");
  GatherConv p{};
  p.in = dy; p.wp = w_packed_bwd; p.out = dx; p.bias = nullptr; p.resid = resid;
  p.pro = make_pro(nullptr);
  p.ldi = lddy; p.ldo = lddx; p.ldr = ldr; p.tanh_out = 0;
  if (!g->transposed)
    build_transposed(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  else
    build_forward(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  return launch_gather(p, (hipStream_t)stream);
}
