// Weight gradient of ConvNd / ConvTransposeNd on the fp32 matrix cores.
//
// GEMM view: R[cd][t*Cg + cg] = sum_m dense[m][cd] * gathered[pix(m,t)][cg]
// with m over the COARSE pixel grid and pix(m,t) = m*stride - pad + k_t
// (forward-type mapping):
//   ConvNd:           dense = dy (Cout), gathered = x (Cin, optional norm+act prologue)
//   ConvTransposeNd:  dense = x  (Cin),  gathered = dy (Cout)
// so R is dW in torch layout [cd][cg][t] up to the (t,cg) order, which the
// split-K reducer fixes while summing the slabs in a fixed order
// (deterministic; no atomics).
//
// Both operands are "K-major" as stored (one channels-last pixel row per K), so
// LDS tiles are [pixel][channel] and the 32x32x2 MFMA operand of lane (i,h) is
// tile[2s+h][i]: consecutive lanes read consecutive floats -- conflict-free
// ds_read_b32 with no transposition.
#include "mpgan_common.h"
#include <stdlib.h>
#include <type_traits>

#include "wgrad_pipe.h"

namespace mpgan {

// KW = 1: the 4 waves tile the BD x BG block (WM x WN waves of TM x TN tiles).
// KW = 4: small block (<= 64x32): every wave owns the whole tile and a quarter
//         of each K-step; the 4 wave-partials go to 4 slabs of the workspace.
template <int BD, int BG, int TM, int TN, int WN, int KW, bool SCALAR_D, bool SCALAR_G>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int STAGE = WBK * (BD + BG);
  constexpr int DCH = BD / 4, GCH = BG / 4;          // float4 chunks per pixel row
  constexpr int DLOADS = (WBK * DCH) / 256;          // chunks per thread
  constexpr int GLOADS = (WBK * GCH) / 256 > 0 ? (WBK * GCH) / 256 : 1;
  constexpr int GTHREADS = WBK * GCH;                // threads that load the gathered tile (<=256)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = KW == 1 ? wid / WN : 0, wn = KW == 1 ? wid % WN : 0;
  const int T = p.Kz * p.Ky * p.Kx;
  const int NC = T * p.Cg;  // columns of R
  const WBlockId bid = wgrad_block_id(p);
  const int c0 = bid.tc * BG;
  const int d0 = bid.td * BD;
  const long M = (long)p.N * p.Mz * p.My * p.Mx;
  const long mbeg = (long)bid.split * p.chunk;
  const long mend = mbeg + p.chunk < M ? mbeg + p.chunk : M;
  const int nk = mbeg < mend ? (int)((mend - mbeg + WBK - 1) / WBK) : 0;
  const float slope = pro_slope(p.pro);

  // ---- gathered operand: this thread's column chunk is fixed for the block ----
  const int gcc = tid % GCH, grow0 = tid / GCH;      // rows grow0 + (256/GCH)*i
  int gtap[4], gci[4], gkz[4], gky[4], gkx[4];
  bool gvalid[4];
  {
    const int nel = SCALAR_G ? 4 : 1;
    for (int e = 0; e < 4; ++e) {
      gvalid[e] = false; gtap[e] = gci[e] = gkz[e] = gky[e] = gkx[e] = 0;
      if (e >= nel) continue;
      int col = c0 + gcc * 4 + e;
      if (col < NC) {
        int t = col / p.Cg;
        gci[e] = col - t * p.Cg;
        gtap[e] = t;
        gkx[e] = t % p.Kx;
        int q = t / p.Kx;
        gky[e] = q % p.Ky;
        gkz[e] = q / p.Ky;
        gvalid[e] = true;
      }
    }
  }
  // ---- dense operand chunk ----
  const int dcc = tid % DCH, drow0 = tid / DCH;      // rows drow0 + (256/DCH)*i

  float4 rd[DLOADS], rg[GLOADS];
  float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);   // fused bias gradient: column sums of the dense rows

  // pixel cursors of this thread's gathered rows: divisions once, then += 32 with carries
  constexpr int GROWSTEP = 256 / GCH;
  int cn[GLOADS], cz[GLOADS], cy[GLOADS], cx[GLOADS];
#pragma unroll
  for (int i = 0; i < GLOADS; ++i) {
    const unsigned um = (unsigned)(mbeg + grow0 + GROWSTEP * i), uMx = p.Mx, uMy = p.My, uMz = p.Mz;
    cx[i] = (int)(um % uMx);
    unsigned q = um / uMx;
    cy[i] = (int)(q % uMy);
    q /= uMy;
    cz[i] = (int)(q % uMz);
    cn[i] = (int)(q / uMz);
  }

  auto global_load = [&](int kt) {
    const long mb = mbeg + (long)kt * WBK;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) {
      const long m = mb + drow0 + (256 / DCH) * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = d0 + dcc * 4;
      if constexpr (!SCALAR_D) {
        const bool ok = m < mend && c < p.Cd;
        v = *reinterpret_cast<const float4*>(p.dense + (ok ? m * p.ldd + c : 0));
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (m < mend) {
        const float* src = p.dense + m * p.ldd;
        if (c + 0 < p.Cd) v.x = src[c + 0];
        if (c + 1 < p.Cd) v.y = src[c + 1];
        if (c + 2 < p.Cd) v.z = src[c + 2];
        if (c + 3 < p.Cd) v.w = src[c + 3];
      }
      rd[i] = v;
      bacc.x += v.x; bacc.y += v.y; bacc.z += v.z; bacc.w += v.w;
    }
#pragma unroll
    for (int i = 0; i < GLOADS; ++i) {
      const long m = mb + grow0 + GROWSTEP * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int n = cn[i];
      const int bz = cz[i] * p.sz - p.pz, by = cy[i] * p.sy - p.py, bx = cx[i] * p.sx - p.px;
      if constexpr (!SCALAR_G) {
        const int iz = bz + gkz[0], iy = by + gky[0], ix = bx + gkx[0];
        const bool ok = m < mend && gvalid[0] && (unsigned)iz < (unsigned)p.Gz && (unsigned)iy < (unsigned)p.Gy &&
                        (unsigned)ix < (unsigned)p.Gx;
        const long pix = (((long)n * p.Gz + iz) * p.Gy + iy) * p.Gx + ix;
        v = *reinterpret_cast<const float4*>(p.gath + (ok ? pix * p.ldg + gci[0] : 0));
        if (p.pro.scale) {
          const int si = ok ? n * p.pro.n_stride + gci[0] : 0;
          const float4 sc = *reinterpret_cast<const float4*>(p.pro.scale + si);
          const float4 sh = *reinterpret_cast<const float4*>(p.pro.shift + si);
          v.x = act_apply(v.x * sc.x + sh.x, p.pro.act, slope);
          v.y = act_apply(v.y * sc.y + sh.y, p.pro.act, slope);
          v.z = act_apply(v.z * sc.z + sh.z, p.pro.act, slope);
          v.w = act_apply(v.w * sc.w + sh.w, p.pro.act, slope);
        }
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int iz = bz + gkz[e], iy = by + gky[e], ix = bx + gkx[e];
          const bool ok = m < mend && gvalid[e] && (unsigned)iz < (unsigned)p.Gz &&
                          (unsigned)iy < (unsigned)p.Gy && (unsigned)ix < (unsigned)p.Gx;
          const long pix = (((long)n * p.Gz + iz) * p.Gy + iy) * p.Gx + ix;
          float x = p.gath[ok ? pix * p.ldg + gci[e] : 0];
          if (p.pro.scale) {
            const int si = ok ? n * p.pro.n_stride + gci[e] : 0;
            x = act_apply(x * p.pro.scale[si] + p.pro.shift[si], p.pro.act, slope);
          }
          w[e] = ok ? x : 0.f;
        }
        v = make_float4(w[0], w[1], w[2], w[3]);
      }
      rg[i] = v;
      // advance this row's cursor by one K-step (32 pixels)
      cx[i] += WBK;
      while (cx[i] >= p.Mx) {
        cx[i] -= p.Mx;
        if (++cy[i] == p.My) {
          cy[i] = 0;
          if (++cz[i] == p.Mz) { cz[i] = 0; ++cn[i]; }
        }
      }
    }
  };

  auto lds_store = [&](int buf) {
    float* Ds = lds + buf * STAGE;
    float* Gs = Ds + WBK * BD;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i)
      *reinterpret_cast<float4*>(Ds + (drow0 + (256 / DCH) * i) * BD + dcc * 4) = rd[i];
#pragma unroll
    for (int i = 0; i < GLOADS; ++i)
      *reinterpret_cast<float4*>(Gs + (grow0 + (256 / GCH) * i) * BG + gcc * 4) = rg[i];
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
    const float* Ds = lds + cur * STAGE + wm * TM * 32 + li;
    const float* Gs = lds + cur * STAGE + WBK * BD + wn * TN * 32 + li;
    // fragment reads run one group (4 MFMA K-steps) ahead of their MFMAs
    constexpr int NG = (16 / KW) / 4;              // groups of 4 steps: 4 (KW=1) or 1 (KW=4)
    float a[2][4][TM], b[2][4][TN];
    auto read_group = [&](int g, int slot) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = (KW == 1 ? 0 : (16 / KW) * wid) + 4 * g + q;
        const int kk = 2 * s + lh;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[slot][q][tm] = Ds[kk * BD + tm * 32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[slot][q][tn] = Gs[kk * BG + tn * 32];
      }
    };
    read_group(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int sl = g & 1;
      if (g + 1 < NG) read_group(g + 1, sl ^ 1);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][q][tm], b[sl][q][tn], acc[tm][tn], 0, 0, 0);
      if (g + 1 < NG) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nk) lds_store(cur ^ 1);
    __syncthreads();
  }

  if (p.bias_partial != nullptr && bid.tc == 0) {
    // block-reduce the per-thread column sums over the (256/DCH) row groups, fixed order
    float* red = lds;   // [256/DCH][BD]; the K-loop is done with LDS
    *reinterpret_cast<float4*>(red + drow0 * BD + dcc * 4) = bacc;
    __syncthreads();
    if (tid < BD && d0 + tid < p.Cd) {
      float t = 0.f;
      for (int r = 0; r < 256 / DCH; ++r) t += red[r * BD + tid];
      p.bias_partial[(long)bid.split * p.Cd + d0 + tid] = t;
    }
  }
  float* out = p.partial + ((long)bid.split * KW + (KW == 1 ? 0 : wid)) * p.Cd * NC;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = c0 + (wn * TN + tn) * 32 + li;
      if (col >= NC) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cd = d0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (cd < p.Cd) out[(long)cd * NC + col] = acc[tm][tn][r];
      }
    }
}

// dW[cd][cg][t] = beta*dW + sum_split partial[split][cd][t*Cg+cg]
// block = 32 consecutive elements x 8 split lanes; lanes sum their splits in
// order, then the 8 lane sums are added in order: deterministic.  The trailing
// `bias_blocks` blocks reduce the fused bias-gradient partials the same way
// (db[c] = beta*db[c] + sum_split bp[split][c]) -- one launch instead of two.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int nsplit, int Cd, int Cg, int T, float beta,
                                                           const float* __restrict__ bp, int bsplit,
                                                           float* __restrict__ db, int bias_blocks) {
  __shared__ float sh[8][33];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int main_blocks = (int)gridDim.x - bias_blocks;
  if ((int)blockIdx.x >= main_blocks) {
    const int c = ((int)blockIdx.x - main_blocks) * 32 + e;
    float s = 0.f;
    if (c < Cd)
      for (int k = sl; k < bsplit; k += 8) s += bp[(long)k * Cd + c];
    sh[sl][e] = s;
    __syncthreads();
    if (sl == 0 && c < Cd) {
      float t = sh[0][e];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += sh[k][e];
      db[c] = beta != 0.f ? beta * db[c] + t : t;
    }
    return;
  }
  const long total = (long)Cd * Cg * T;
  const long NC = (long)T * Cg;
  for (long base = (long)blockIdx.x * 32; base < total; base += (long)main_blocks * 32) {
    const long i = base + e;
    float s = 0.f;
    if (i < total)
      for (int k = sl; k < nsplit; k += 8) s += partial[(long)k * total + i];
    sh[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
      float t = sh[0][e];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += sh[k][e];
      const long cd = i / NC;
      const long rem = i - cd * NC;
      const int tt = (int)(rem / Cg);
      const int cg = (int)(rem - (long)tt * Cg);
      const long o = (cd * Cg + cg) * T + tt;
      dw[o] = beta != 0.f ? beta * dw[o] + t : t;
    }
    __syncthreads();
  }
}

// The same reduction for SMALL results (the generator's narrow layers: 16 x 144 ... 64 x 288 outputs, hundreds
// of slabs): with 32 elements per block only total/32 (72 for a 16 -> 16 layer) blocks exist and every thread
// walks 64 slabs.  Here a block takes 8 consecutive elements (one 32-byte sector per slab) x 32 split lanes:
// four times the blocks, a quarter of the loads per thread, all of them independent.  Lane sums, then the 32
// lanes, are added in a fixed order.  The trailing `bias_blocks` blocks do the fused bias gradient as above.
__global__ __launch_bounds__(256) void wgrad_reduce_narrow_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  int nsplit, int Cd, int Cg, int T, float beta,
                                                                  const float* __restrict__ bp, int bsplit,
                                                                  float* __restrict__ db, int bias_blocks) {
  __shared__ float sh[32][9];
  const int e = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int main_blocks = (int)gridDim.x - bias_blocks;
  if ((int)blockIdx.x >= main_blocks) {
    const int c = ((int)blockIdx.x - main_blocks) * 8 + e;
    float s = 0.f;
    if (c < Cd)
      for (int k = sl; k < bsplit; k += 32) s += bp[(long)k * Cd + c];
    sh[sl][e] = s;
    __syncthreads();
    if (sl == 0 && c < Cd) {
      float t = sh[0][e];
#pragma unroll
      for (int k = 1; k < 32; ++k) t += sh[k][e];
      db[c] = beta != 0.f ? beta * db[c] + t : t;
    }
    return;
  }
  const long total = (long)Cd * Cg * T;
  const long NC = (long)T * Cg;
  const long i = (long)blockIdx.x * 8 + e;
  float s = 0.f;
  if (i < total) {
    const float* src = partial + i;
    int k = sl;
    for (; k + 32 * 7 < nsplit; k += 32 * 8) {               // eight independent loads in flight
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + 32 * u) * total];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; k += 32) s += src[(long)k * total];
  }
  sh[sl][e] = s;
  __syncthreads();
  if (sl == 0 && i < total) {
    float t = sh[0][e];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sh[k][e];
    const long cd = i / NC;
    const long rem = i - cd * NC;
    const int tt = (int)(rem / Cg);
    const int cg = (int)(rem - (long)tt * Cg);
    const long o = (cd * Cg + cg) * T + tt;
    dw[o] = beta != 0.f ? beta * dw[o] + t : t;
  }
}

// One entry for both reducers: the narrow form when the wide one would leave most CUs idle.
static inline void launch_wgrad_reduce(hipStream_t st, const float* partial, float* dw, int nsplit, int Cd, int Cg, int T,
                                       float beta, const float* bp, int bsplit, float* db) {
  const long total = (long)Cd * Cg * T;
  if (total <= 32L * 512 && nsplit >= 64) {
    const int blocks = (int)((total + 7) / 8), bb = db ? (Cd + 7) / 8 : 0;
    hipLaunchKernelGGL(wgrad_reduce_narrow_kernel, dim3(blocks + bb), dim3(256), 0, st, partial, dw, nsplit, Cd, Cg, T, beta,
                       bp, bsplit, db, bb);
    return;
  }
  int blocks = (int)((total + 31) / 32);
  if (blocks > 8192) blocks = 8192;
  const int bb = db ? (Cd + 31) / 32 : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks + bb), dim3(256), 0, st, partial, dw, nsplit, Cd, Cg, T, beta, bp,
                     bsplit, db, bb);
}

// db[c] += sum_split bias_partial[split][c]  (one wave per channel, fixed order)
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* __restrict__ bp, int nsplit, int Cd,
                                                                float* __restrict__ db, float beta) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gt >> 6, lane = threadIdx.x & 63;
  if (c >= Cd) return;
  double s = 0.0;
  for (int k = lane; k < nsplit; k += 64) s += (double)bp[(long)k * Cd + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) db[c] = beta != 0.f ? beta * db[c] + (float)s : (float)s;
}

// ---------------------------------------------------------------------------
// Thin weight gradient: the gathered operand has ONE channel (Cin == 1 convs:
// G down0, D conv1, the 1->1 conv; and the 32->1 transposed conv, whose gathered
// operand is dy).  dW[c][t] = sum_m dense[m][c] * g[pix(m,t)] is an HBM-bound
// reduction with C*T <= a few hundred outputs -- an MFMA tile would be >95 %
// padding.  One thread = V channels x all taps, pixel lanes strided over the
// chunk; LDS reduction over the lanes in fixed order; per-block partial slabs go
// through the same deterministic slab reducer as the MFMA path.
// ---------------------------------------------------------------------------
template <int V, int T, bool DENSE_BF16 = false>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const WgradParams p) {
  extern __shared__ float red[];                 // [PL][Cd*T + Cd]
  const int CQ = (p.Cd + V - 1) / V;
  const int PL = 256 / CQ;                       // pixel lanes
  const int q = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int c = q * V;
  const long M = (long)p.N * p.Mz * p.My * p.Mx;
  const long mbeg = (long)blockIdx.x * p.chunk;
  const long mend = mbeg + p.chunk < M ? mbeg + p.chunk : M;
  float acc[T][V], bsum[V];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
#pragma unroll
  for (int e = 0; e < V; ++e) bsum[e] = 0.f;
  if (pl < PL) {
    for (long m = mbeg + pl; m < mend; m += PL) {
      unsigned r, umx, umy, umz;
      fdivmod((unsigned)m, p.fMx, r, umx);
      fdivmod(r, p.fMy, r, umy);
      fdivmod(r, p.fMz, r, umz);
      const int mx = (int)umx, my = (int)umy, mz = (int)umz, n = (int)r;
      float d[V];
      if constexpr (DENSE_BF16) {               // host admits V == 4, Cd % 4 == 0 only
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.dense) + m * p.ldd + c);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = (float)t4[e < 4 ? e : 0];
      } else if (V == 4 && c + 3 < p.Cd) {
        const float4 t4 = *reinterpret_cast<const float4*>(p.dense + m * p.ldd + c);
        d[0] = t4.x; d[V > 1 ? 1 : 0] = t4.y; d[V > 2 ? 2 : 0] = t4.z; d[V > 3 ? 3 : 0] = t4.w;
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = c + e < p.Cd ? p.dense[m * p.ldd + c + e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < V; ++e) bsum[e] += d[e];
      const int bz = mz * p.sz - p.pz, by = my * p.sy - p.py, bx = mx * p.sx - p.px;
      // ONE 64-bit address per pixel (its tap (0, 0, 0); it may lie in front of the tensor and is then never read) and a
      // wave-uniform 32-bit offset per (kz, ky) row of taps: the per-tap 64-bit pixel index cost more than the FMAs
      const float* __restrict__ g0 = p.gath + ((((long)n * p.Gz + bz) * p.Gy + by) * p.Gx + bx) * p.ldg;
      constexpr int KX = T == 1 ? 1 : 3, KY = KX, KZ = T / (KX * KY);       // host admits only 1, 3x3 and 3x3x3 kernels here
#pragma unroll
      for (int kz = 0; kz < KZ; ++kz) {
        const bool okz = (unsigned)(bz + kz) < (unsigned)p.Gz;
#pragma unroll
        for (int ky = 0; ky < KY; ++ky) {
          const bool oky = okz && (unsigned)(by + ky) < (unsigned)p.Gy;
          const int ro = (kz * p.Gy + ky) * p.Gx * p.ldg;
#pragma unroll
          for (int kx = 0; kx < KX; ++kx) {
            const int t = (kz * KY + ky) * KX + kx;
            const bool ok = oky && (unsigned)(bx + kx) < (unsigned)p.Gx;
            const float g = ok ? g0[ro + kx * p.ldg] : 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) acc[t][e] = fmaf(d[e], g, acc[t][e]);
          }
        }
      }
    }
  }
  const int W = p.Cd * T + p.Cd;                 // floats per lane row: [c][t] then bias[c]
  // The lane rows are folded in `fold_rounds` rounds over PL / rounds LDS rows (round 0 stores, the others add their
  // own element: one owner per element and round, fixed order), so that 27-tap layers with 16 / 32 dense channels
  // (3-D: 64 x 448 / 32 x 896 floats = 112 KiB of lane rows) stay within 64 KiB and two blocks per CU.
  const int NR = p.fold_rounds > 1 ? p.fold_rounds : 1;
  const int PLr = PL / NR;
  for (int r = 0; r < NR; ++r) {
    if (pl < PLr * NR && pl / PLr == r) {
      const int row = pl - r * PLr;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if (c + e >= p.Cd) break;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          float* dst = red + row * W + (c + e) * T + t;
          *dst = r == 0 ? acc[t][e] : *dst + acc[t][e];
        }
        float* db = red + row * W + p.Cd * T + c + e;
        *db = r == 0 ? bsum[e] : *db + bsum[e];
      }
    }
    __syncthreads();
  }
  const int PLf = PLr;                           // rows left to fold
  float* out = p.partial + (long)blockIdx.x * p.Cd * T;
  // fold the PL lane rows: narrow rows (few channels) are first folded by G row groups in parallel
  const int G = W < 128 ? 256 / W : 1;
  if (G > 1) {
    const int i = threadIdx.x % W, g = threadIdx.x / W;
    const int GR = G < PLf ? G : PLf;
    float s1 = 0.f;
    if (g < GR)
      for (int r = g; r < PLf; r += GR) s1 += red[r * W + i];
    __syncthreads();
    if (g < GR) red[g * W + i] = s1;
    __syncthreads();
  }
  const int rows = G > 1 ? (G < PLf ? G : PLf) : PLf;
  for (int i = threadIdx.x; i < W; i += 256) {
    float s2 = 0.f;
    for (int r = 0; r < rows; ++r) s2 += red[r * W + i];
    if (i < p.Cd * T) out[i] = s2;
    else if (p.bias_partial) p.bias_partial[(long)blockIdx.x * p.Cd + (i - p.Cd * T)] = s2;
  }
}

// ---------------------------------------------------------------------------
// Thin weight gradient, row-walking form, for the 1 -> 64 pad-free stride-1 3x3 / 3x3x3 conv (D.conv1; its dense
// operand dy is the largest tensor of the discriminator: 1 GB in bf16 at 128^3).  lane = output channel, one wave
// walks whole output rows four pixels at a time; the gathered samples of those four pixels (a 6-wide window per
// (kz, ky)) are WAVE-UNIFORM, so they come through the scalar cache into SGPRs and every FMA takes one as its
// scalar operand: no per-lane address arithmetic or bounds tests at all (the lane-strided kernel above spends
// ~4 vector instructions on addresses per FMA at 27 taps).  The last group of a row is re-anchored at Mx - 4 and
// its already-counted pixels masked, so nothing is read past the row.
// ---------------------------------------------------------------------------
template <int T, bool DENSE_BF16>
__global__ __launch_bounds__(256) void thin_wgrad_rows_kernel(const WgradParams p) {
  __shared__ float red[4][64 * T + 64];
  constexpr int KZ = T == 27 ? 3 : 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int rows = p.N * p.Mz * p.My;
  const int nwaves = (int)gridDim.x * 4;
  const int Mx = p.Mx, ngroups = (Mx + 3) >> 2;
  float acc[T], bsum = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  for (int r = (int)blockIdx.x * 4 + wave; r < rows; r += nwaves) {
    const int my = r % p.My, q = r / p.My;
    const int mz = q % p.Mz, n = q / p.Mz;
    const long drow = (long)r * Mx;                                        // first pixel of the dense row
    const float* __restrict__ g0 = p.gath + (((long)n * p.Gz + mz) * p.Gy + my) * p.Gx;
    for (int gi = 0; gi < ngroups; ++gi) {
      const int x0 = gi + 1 < ngroups ? 4 * gi : Mx - 4;
      const int first = 4 * gi - x0;                                       // pixels j < first were counted by the previous group
      float d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long e = (drow + x0 + j) * p.ldd + lane;
        float v;
        if constexpr (DENSE_BF16) v = (float)reinterpret_cast<const __bf16*>(p.dense)[e];
        else v = p.dense[e];
        d[j] = j >= first ? v : 0.f;
        bsum += d[j];
      }
#pragma unroll
      for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* __restrict__ w = g0 + ((long)kz * p.Gy + ky) * p.Gx + x0;   // wave-uniform: scalar loads
          float win[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) win[i] = w[i];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[(kz * 3 + ky) * 3 + kx] = fmaf(d[j], win[j + kx], acc[(kz * 3 + ky) * 3 + kx]);
        }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) red[wave][lane * T + t] = acc[t];
  red[wave][64 * T + lane] = bsum;
  __syncthreads();
  float* out = p.partial + (long)blockIdx.x * 64 * T;
  for (int i = threadIdx.x; i < 64 * T + 64; i += 256) {
    const float s = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (i < 64 * T) out[i] = s;
    else if (p.bias_partial) p.bias_partial[(long)blockIdx.x * 64 + (i - 64 * T)] = s;
  }
}

static bool thin_rows_ok(const WgradParams& p, int T) {
  static const bool off = dev_env("MPGAN_DBG_NO_THIN_ROWS") != nullptr;
  return !off && p.Cd == 64 && p.Cg == 1 && p.ldg == 1 && !p.pro.scale && (T == 9 || T == 27) && p.Kx == 3 && p.Ky == 3 &&
         p.Kz == (T == 27 ? 3 : 1) && p.sz == 1 && p.sy == 1 && p.sx == 1 && p.pz == 0 && p.py == 0 && p.px == 0 && p.Mx >= 4 &&
         p.Gx == p.Mx + 2 && p.Gy == p.My + 2 && p.Gz == p.Mz + (T == 27 ? 2 : 0);
}

template <bool DENSE_BF16>
static void launch_thin_rows(const WgradParams& p, int T, int blocks, hipStream_t st) {
  if (T == 9) hipLaunchKernelGGL((thin_wgrad_rows_kernel<9, DENSE_BF16>), dim3(blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((thin_wgrad_rows_kernel<27, DENSE_BF16>), dim3(blocks), dim3(256), 0, st, p);
}

// ---------------------------------------------------------------------------
// Weight gradient of the 16 -> 16 3x3x3 stride-1 conv (the 3-D U-Net's 64^3 level; conv_igemm.hip's
// gather_patch3d_c16_kernel serves its forward and backward-data): dW[co][tap][ci] = sum_px dy[px][co] * a[px + tap][ci].
// The MFMA pipeline above tiles this 16 x 432 result as 32 x 128 (a quarter of each tile is real) and re-stages
// every input pixel once per tap.  Here a persistent block walks 2 x 8 x 8 pixel tiles: dy's tile and the
// 4 x 10 x 10 input patch (producer's BatchNorm + PReLU applied on the way in, zero outside = the padding) are
// staged once, then v_mfma_f32_16x16x4_f32 contracts over PIXELS: A[co][px] from the dy tile, B[px][ci] from the
// patch shifted by the tap -- wave w owns taps w, w+4, ... (7 accumulators of 4 registers), the last wave also the
// bias gradient (B = ones).  Accumulators live across the block's tiles; one partial slab per block goes through
// the fixed-order reducer.
// ---------------------------------------------------------------------------
constexpr int WP3_TZ = 2, WP3_TY = 8, WP3_TX = 8, WP3_PY = 10, WP3_PX = 10, WP3_PROWS = 400, WP3_BLOCKS = 512;
struct WP3Grid { int tiles_z, tiles_y, tiles_x; };

//   MM16 (MPGAN_CONV_MM_BF16): v_mfma_f32_16x16x32_bf16 contracts 32 pixels per instruction; lane (channel ln, quarter kq)
//   reads the 8 pixels 32 jj + 8 kq + j of its channel (as many ds_read_b32 as the fp32 form's 4-pixel steps), rounds
//   them to bf16: 28 instead of 224 MFMAs per tile and wave.  The bias gradient (column sums of dy) is summed from the
//   UNROUNDED values on the vector ALUs (the contract rounds matrix operands only).
template <bool HAS_PRO, bool MM16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_patch3d_c16_kernel(const WgradParams p, const WP3Grid tg) {
  __shared__ __attribute__((aligned(16))) float patch[WP3_PROWS * 16];   // [patch pixel][ci]
  __shared__ __attribute__((aligned(16))) float dyt[128 * 16];           // [tile pixel][co]
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ln = lane & 15, g = lane >> 4;
  f32x4 acc[7], accb;
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = 0.f;
  const int c4 = tid & 3;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  float slope = 1.f;
  if constexpr (HAS_PRO) {
    sc = *reinterpret_cast<const float4*>(p.pro.scale + 4 * c4);
    sh = *reinterpret_cast<const float4*>(p.pro.shift + 4 * c4);
    slope = pro_slope(p.pro);
  }
  const int act = p.pro.act;
  int toff[7];                                                  // this wave's taps: patch offset of tap w + 4 k (floats)
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int tap = __builtin_amdgcn_readfirstlane(wid) + 4 * k;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    toff[k] = ((kz * WP3_PY + ky) * WP3_PX + kx) * 16;
  }
  const bool last_wave = __builtin_amdgcn_readfirstlane(wid) == 3;   // 6 taps + the bias column sums
  const unsigned ntiles = (unsigned)(p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x);
  const unsigned per = (ntiles + gridDim.x - 1) / gridDim.x;
  const unsigned w0 = xcd_remap(blockIdx.x, gridDim.x) * per;
  const unsigned wend = w0 + per < ntiles ? w0 + per : ntiles;
  // Staging is "issue every load of the tile, then store" (a bounds test around each load put each chunk behind its
  // own memory round trip), and the NEXT tile's operands are fetched into registers under the current contraction.
  constexpr int NPCH = (WP3_PROWS * 4 + 255) / 256;              // 7 patch chunks per thread
  int pzyx[NPCH];
#pragma unroll
  for (int i = 0; i < NPCH; ++i) {
    const int pr = (tid + 256 * i) >> 2;
    const int pz = pr / (WP3_PY * WP3_PX), rem = pr - pz * (WP3_PY * WP3_PX);
    const int py = rem / WP3_PX, px = rem - py * WP3_PX;
    pzyx[i] = pr < WP3_PROWS ? ((pz << 16) | (py << 8) | px) : -1;
  }
  float4 pv[NPCH], dv[2];
  unsigned pok = 0;
  auto load_tile = [&](unsigned t) {
    const int tx = t % tg.tiles_x; t /= tg.tiles_x;
    const int ty = t % tg.tiles_y; t /= tg.tiles_y;
    const int tz = t % tg.tiles_z;
    const int n = t / tg.tiles_z;
    const int oz0 = tz * WP3_TZ, oy0 = ty * WP3_TY, ox0 = tx * WP3_TX;
    pok = 0;
#pragma unroll
    for (int i = 0; i < NPCH; ++i) {
      const int iz = oz0 - p.pz + (pzyx[i] >> 16), iy = oy0 - p.py + ((pzyx[i] >> 8) & 255), ix = ox0 - p.px + (pzyx[i] & 255);
      const bool ok = pzyx[i] >= 0 && (unsigned)iz < (unsigned)p.Gz && (unsigned)iy < (unsigned)p.Gy && (unsigned)ix < (unsigned)p.Gx;
      const long off = ok ? ((((long)n * p.Gz + iz) * p.Gy + iy) * p.Gx + ix) * p.ldg : 0;
      pv[i] = *reinterpret_cast<const float4*>(p.gath + off + 4 * c4);
      pok |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = (tid + 256 * i) >> 2;                        // tile pixel 0..127
      const int oz = oz0 + (q >> 6), oy = oy0 + ((q >> 3) & 7), ox = ox0 + (q & 7);
      const bool ok = oz < p.Mz && oy < p.My && ox < p.Mx;
      const long off = ok ? ((((long)n * p.Mz + oz) * p.My + oy) * p.Mx + ox) * p.ldd : 0;
      const float4 v = *reinterpret_cast<const float4*>(p.dense + off + 4 * c4);
      dv[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NPCH; ++i) {
      float4 v = pv[i];
      if constexpr (HAS_PRO) {
        v.x = act_apply(fmaf(v.x, sc.x, sh.x), act, slope);
        v.y = act_apply(fmaf(v.y, sc.y, sh.y), act, slope);
        v.z = act_apply(fmaf(v.z, sc.z, sh.z), act, slope);
        v.w = act_apply(fmaf(v.w, sc.w, sh.w), act, slope);
      }
      const bool ok = (pok >> i) & 1u;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if (pzyx[i] >= 0) {
        const int pr = ((pzyx[i] >> 16) * WP3_PY + ((pzyx[i] >> 8) & 255)) * WP3_PX + (pzyx[i] & 255);
        *reinterpret_cast<float4*>(patch + pr * 16 + 4 * c4) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(dyt + ((tid + 256 * i) >> 2) * 16 + 4 * c4) = dv[i];
  };
  // contraction over the tile's 128 pixels, four at a time (k = pixel 4 j + g): branch-free bodies, one per wave kind
  auto contract = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    if constexpr (MM16) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        bf16x8 af;
        int pb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int q = 32 * jj + 8 * g + e;                     // k = pixel q of the tile
          const float dv_ = dyt[q * 16 + ln];
          if constexpr (BIAS) accb[0] += dv_;                    // (MM16: accb[0] is this lane's running column sum)
          af[e] = (__bf16)dv_;
          pb[e] = ((((q >> 6) * WP3_PY) + ((q >> 3) & 7)) * WP3_PX + (q & 7)) * 16 + ln;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          if (BIAS && k == 6) break;
          bf16x8 bf;
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[e] = (__bf16)patch[pb[e] + toff[k]];
          acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[k], 0, 0, 0);
        }
      }
      return;
    }
#pragma unroll 4
    for (int j = 0; j < 32; ++j) {
      const int q = 4 * j + g;
      const float a = dyt[q * 16 + ln];                          // A[row = co = ln][k]
      const int pb = (((q >> 6) * WP3_PY) + ((q >> 3) & 7)) * WP3_PX + (q & 7);
      const float* brow = patch + pb * 16 + ln;                  // B[k][col = ci = ln] at tap (0, 0, 0)
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[toff[k]], acc[k], 0, 0, 0);
      if constexpr (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, accb, 0, 0, 0);   // taps 3, 7, .. 23: six
      else acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[toff[6]], acc[6], 0, 0, 0);        // taps w + 24 <= 26
    }
  };
  if (w0 < wend) load_tile(w0);
  for (unsigned tt = w0; tt < wend; ++tt) {
    __syncthreads();                                            // the previous tile's reads are done
    store_tile();
    __syncthreads();
    if (tt + 1 < wend) load_tile(tt + 1);
    if (last_wave) contract(std::true_type{});
    else contract(std::false_type{});
  }
  // ---- D[row = co = 4 g + i][col = ci = ln] -> partial[block][co][tap * 16 + ci]; bias partial [block][co] ----
  float* out = p.partial + (long)blockIdx.x * 16 * 432;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int tap = wid + 4 * k;
    if (tap < 27)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(4 * g + i) * 432 + tap * 16 + ln] = acc[k][i];
  }
  if constexpr (MM16) {
    if (wid == 3 && p.bias_partial) {                          // lanes (co = ln, quarter g): fold the four quarters, fixed order
      float s = accb[0];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) p.bias_partial[(long)blockIdx.x * 16 + ln] = s;
    }
  } else {
    if (wid == 3 && p.bias_partial && ln == 0)
#pragma unroll
      for (int i = 0; i < 4; ++i) p.bias_partial[(long)blockIdx.x * 16 + 4 * g + i] = accb[i];
  }
}

static bool wgrad_p3_geom_ok(const mpgan_conv_geom* g) {
  static const bool off = dev_env("MPGAN_DBG_NO_PATCH3D") != nullptr;
  if (off || g->transposed || g->cin != 16 || g->cout != 16) return false;
  for (int d = 0; d < 3; ++d)
    if (g->k[d] != 3 || g->stride[d] != 1 || g->pad[d] < 0 || g->pad[d] > 1) return false;
  return g->out_dhw[0] >= 2 && g->out_dhw[1] >= 4 && g->out_dhw[2] >= 4;
}

static int wgrad_p3_blocks(const mpgan_conv_geom* g) {
  const long tiles = (long)g->n * ((g->out_dhw[0] + WP3_TZ - 1) / WP3_TZ) * ((g->out_dhw[1] + WP3_TY - 1) / WP3_TY) *
                     ((g->out_dhw[2] + WP3_TX - 1) / WP3_TX);
  return (int)(tiles < WP3_BLOCKS ? tiles : WP3_BLOCKS);
}

// ---------------------------------------------------------------------------
// 2-D patch form of the weight gradient for the generator's narrow layers (<= 64 dense and <= 32 gathered
// channels, 3x3 taps; round 3): the U-Net's 16 -> 16 and 32 -> 32 stride-1 units, its stride-2 down convs
// (16 -> 32, 32 -> 64; each twice: unit0 and the strided residual conv) and ConvTranspose2d(64 -> 16).
//   R[cd][t][cg] = sum over coarse pixels m of dense[m][cd] * gath[m * S - 1 + k_t][cg]      (S = 1 | 2)
// The K-stepped pipeline above tiles such a 16 x 144 result as 32 x 128, re-stages every gathered pixel once per
// tap and leaves one slab per K split (16 -> 16 at 128^2: 47 us + an 11 us reducer over 11 MB of slabs for
// 1.2 GFLOP).  Here persistent blocks walk 8 x TX tiles of coarse pixels: the dense tile and the gathered patch
// (producer's BatchNorm + PReLU applied on the way in, zero outside = the padding) are staged ONCE per tile --
// the next tile's loads are in flight under the current contraction -- and v_mfma_f32_16x16x4_f32 contracts over
// PIXELS (A[cd][px] from the dense tile, B[px][cg] from the patch shifted by the tap).  Work units (tap, 16-wide
// dense block) are dealt round-robin to the four waves; accumulators live across the block's tiles, so a block
// leaves ONE slab (and one bias row: column sums of the dense operand, accumulated by the loading threads) for
// the fixed-order reducer: <= 512 slabs whatever the image size.
// LDS pitches are chosen so that the four pixel groups of a wave read four different 16-bank quarters:
// S * pitch = 16 or 48 (mod 64) floats.
// ---------------------------------------------------------------------------
constexpr int WP2_TY = 8;
template <int CD, int CG, int S, int TX>
struct WP2 {
  static constexpr int TP = WP2_TY * TX;                       // coarse pixels per tile
  static constexpr int PY = (WP2_TY - 1) * S + 3, PX = (TX - 1) * S + 3;
  static constexpr int PROWS = PY * PX;
  static constexpr int PGP = S == 1 ? (CG == 16 ? 16 : (CG == 32 ? 48 : 80)) : (CG == 16 ? 24 : 40);
  static constexpr int PDP = CD == 16 ? 16 : (CD == 32 ? 48 : 80);
  static constexpr int CA = CD / 16, CB = CG / 16;
  static constexpr int NUNITS = 9 * CA, NU = (NUNITS + 3) / 4;
  static constexpr int GQ = CG / 4, DQ = CD / 4;               // float4 per pixel
  static constexpr int NPCH = (PROWS * GQ + 255) / 256, NDCH = (TP * DQ) / 256;
  static constexpr int SMEM = (PROWS * PGP + TP * PDP) * 4;
  static_assert((TP * DQ) % 256 == 0 && 256 % DQ == 0 && 256 % GQ == 0, "thread -> channel-quad mapping must be fixed");
};
struct WP2Grid { int tiles_y, tiles_x, ntiles; };

template <int CD, int CG, int S, int TX, bool HAS_PRO>
__global__ __launch_bounds__(256, 2) void wgrad_patch2d_kernel(const WgradParams p, const WP2Grid tg) {
  using K = WP2<CD, CG, S, TX>;
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  float* patch = lds2;                                          // [patch pixel][PGP]
  float* dyt = lds2 + K::PROWS * K::PGP;                        // [tile pixel][PDP]
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  f32x4 acc[K::NU][K::CB];
#pragma unroll
  for (int u = 0; u < K::NU; ++u)
#pragma unroll
    for (int b = 0; b < K::CB; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[u][b][i] = 0.f;
  // This wave's units: idx = wid + 4 u -> (tap = idx / CA, dense block ca = idx % CA).  4 % CA == 0, so ca = wid % CA
  // is the SAME for all of a wave's units: one A read per pixel step serves them all.  Per-lane LDS bases (floats):
  // lane (ln, g) reads A[row = cd = 16 ca + ln][k = pixel 4 j + g] and B[k][col = cg = 16 b + ln] at the tap's shift.
  static_assert(4 % K::CA == 0, "a wave's units must share their dense block");
  const int nvalid = (K::NUNITS - wid + 3) / 4;                 // units this wave really owns (NU or NU - 1)
  int a_base = g * K::PDP + ln + 16 * (wid % K::CA);
  int b_base[K::NU];
#pragma unroll
  for (int u = 0; u < K::NU; ++u) {
    const int idx = wid + 4 * u;
    const int tap = (idx < K::NUNITS ? idx : wid) / K::CA;      // a unit past the end re-reads a valid one (never run)
    const int ky = tap / 3, kx = tap - 3 * ky;
    b_base[u] = (ky * K::PX + kx) * K::PGP + g * S * K::PGP + ln;
  }
  // gathered patch: this thread's chunks (tile-independent decode); dense tile: its chunks
  const int gq = tid % K::GQ, dq = tid % K::DQ;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  float slope = 1.f;
  if constexpr (HAS_PRO) {
    sc = *reinterpret_cast<const float4*>(p.pro.scale + 4 * gq);
    sh = *reinterpret_cast<const float4*>(p.pro.shift + 4 * gq);
    slope = pro_slope(p.pro);
  }
  const int act = p.pro.act;
  int pyx[K::NPCH];
#pragma unroll
  for (int i = 0; i < K::NPCH; ++i) {
    const int pr = (tid + 256 * i) / K::GQ;
    pyx[i] = pr < K::PROWS ? (((pr / K::PX) << 16) | (pr % K::PX)) : -1;
  }
  float4 pv[K::NPCH], dv[K::NDCH];
  unsigned pok = 0;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);               // column sums of this thread's dense chunks (bias gradient)
  const int Gy = p.Gy, Gx = p.Gx, My = p.My, Mx = p.Mx, ldg = p.ldg, ldd = p.ldd;
  auto load_tile = [&](unsigned tt) {
    const unsigned tx = tt % (unsigned)tg.tiles_x;
    unsigned t = tt / (unsigned)tg.tiles_x;
    const unsigned ty = t % (unsigned)tg.tiles_y;
    const int n = (int)(t / (unsigned)tg.tiles_y);
    const int my0 = (int)ty * WP2_TY, mx0 = (int)tx * TX;
    const int y0 = my0 * S - 1, x0 = mx0 * S - 1;
    const float* __restrict__ gbase = p.gath + (long)n * Gy * Gx * ldg + 4 * gq;
    pok = 0;
#pragma unroll
    for (int i = 0; i < K::NPCH; ++i) {
      const int iy = y0 + (pyx[i] >> 16), ix = x0 + (pyx[i] & 0xffff);
      const bool ok = pyx[i] >= 0 && (unsigned)iy < (unsigned)Gy && (unsigned)ix < (unsigned)Gx;
      pv[i] = ok ? *reinterpret_cast<const float4*>(gbase + ((long)iy * Gx + ix) * ldg) : make_float4(0.f, 0.f, 0.f, 0.f);
      pok |= (ok ? 1u : 0u) << i;
    }
    const float* __restrict__ dbase = p.dense + (long)n * My * Mx * ldd + 4 * dq;
#pragma unroll
    for (int i = 0; i < K::NDCH; ++i) {
      const int q = (tid + 256 * i) / K::DQ;
      const int my = my0 + q / TX, mx = mx0 + q % TX;
      dv[i] = (my < My && mx < Mx) ? *reinterpret_cast<const float4*>(dbase + ((long)my * Mx + mx) * ldd)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < K::NPCH; ++i) {
      float4 v = pv[i];
      if constexpr (HAS_PRO) {
        if ((pok >> i) & 1u) {                                   // padding stays zero: act(0 * scale + shift) != 0
          v.x = act_apply(fmaf(v.x, sc.x, sh.x), act, slope);
          v.y = act_apply(fmaf(v.y, sc.y, sh.y), act, slope);
          v.z = act_apply(fmaf(v.z, sc.z, sh.z), act, slope);
          v.w = act_apply(fmaf(v.w, sc.w, sh.w), act, slope);
        }
      }
      if (pyx[i] >= 0)
        *reinterpret_cast<float4*>(patch + ((pyx[i] >> 16) * K::PX + (pyx[i] & 0xffff)) * K::PGP + 4 * gq) = v;
    }
#pragma unroll
    for (int i = 0; i < K::NDCH; ++i) {
      const int q = (tid + 256 * i) / K::DQ;
      *reinterpret_cast<float4*>(dyt + q * K::PDP + 4 * dq) = dv[i];
      bsum.x += dv[i].x; bsum.y += dv[i].y; bsum.z += dv[i].z; bsum.w += dv[i].w;
    }
  };

  const unsigned ntiles = (unsigned)tg.ntiles;
  const unsigned per = (ntiles + gridDim.x - 1) / gridDim.x;
  const unsigned w0 = xcd_remap(blockIdx.x, gridDim.x) * per;
  const unsigned wend = w0 + per < ntiles ? w0 + per : ntiles;
  if (w0 < wend) load_tile(w0);
  for (unsigned tt = w0; tt < wend; ++tt) {
    __syncthreads();                                            // the previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (tt + 1 < wend) load_tile(tt + 1);                       // in flight under this tile's contraction
    // ---- contraction over the tile's pixels, four at a time: k = pixel 4 j + g.  Fully unrolled and branch-free:
    //      pixel step j sits at row j / (TX / 4), columns 4 (j % (TX / 4)) + g of the tile, so every LDS address is
    //      "per-lane base + compile-time offset" and the reads of later steps are issued under the MFMAs of earlier
    //      ones (a wave-uniform test per unit inside this loop put every MFMA into a basic block of its own, behind
    //      its own two LDS round trips: 36 us for the 16 -> 16 layer instead of 15).  A wave owns NU or NU - 1 units.
    auto contract = [&](auto nv_tag) {
      constexpr int NV = decltype(nv_tag)::value;
#pragma unroll
      for (int j = 0; j < K::TP / 4; ++j) {
        constexpr int JR = TX / 4;
        const int poff = ((j / JR) * S * K::PX + (j % JR) * 4 * S) * K::PGP;      // compile-time after unrolling
        const float a = dyt[a_base + 4 * j * K::PDP];
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
          for (int b = 0; b < K::CB; ++b)
            acc[u][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, patch[b_base[u] + poff + 16 * b], acc[u][b], 0, 0, 0);
      }
    };
    if (nvalid == K::NU) contract(std::integral_constant<int, K::NU>{});
    else contract(std::integral_constant<int, (K::NU > 1 ? K::NU - 1 : 1)>{});
  }
  // ---- D[row = cd = 16 ca + 4 g + i][col = cg = 16 b + ln] -> partial[block][cd][tap * CG + cg] ----
  float* out = p.partial + (long)blockIdx.x * CD * 9 * CG;
#pragma unroll
  for (int u = 0; u < K::NU; ++u) {
    const int idx = wid + 4 * u;
    if (idx < K::NUNITS) {
      const int tap = idx / K::CA, ca = idx - tap * K::CA;
#pragma unroll
      for (int b = 0; b < K::CB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(16 * ca + 4 * g + i) * (9 * CG) + tap * CG + 16 * b + ln] = acc[u][b][i];
    }
  }
  if (p.bias_partial) {                                         // fold the threads' column sums in a fixed order
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(lds2);
    red[tid] = bsum;
    __syncthreads();
    if (tid < K::DQ) {
      float4 t = red[tid];
      for (int r = tid + K::DQ; r < 256; r += K::DQ) {
        const float4 v = red[r];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      *reinterpret_cast<float4*>(p.bias_partial + (long)blockIdx.x * CD + 4 * tid) = t;
    }
  }
}

// Which instance (if any) serves this conv: 0 none, else an index into the dispatch below.
struct WP2Plan { int inst, blocks, tiles_y, tiles_x, ntiles, Cd, Cg; };
static WP2Plan wgrad_p2_plan(const mpgan_conv_geom* g) {
  WP2Plan pl{};
  static const bool off = dev_env("MPGAN_DBG_NO_WPATCH2D") != nullptr;
  if (off || g->k[0] != 1 || g->k[1] != 3 || g->k[2] != 3 || g->in_dhw[0] != 1 || g->out_dhw[0] != 1) return pl;
  if (g->pad[1] != 1 || g->pad[2] != 1 || g->stride[1] != g->stride[2] || g->stride[1] < 1 || g->stride[1] > 2) return pl;
  const int S = g->stride[1];
  int Cd, Cg, My, Mx;
  if (!g->transposed) {
    Cd = g->cout; Cg = g->cin; My = g->out_dhw[1]; Mx = g->out_dhw[2];
    // forward-type mapping needs out = ceil(in / S) with pad 1, k 3 (true for every even extent)
    if (My != (g->in_dhw[1] + 2 - 3) / S + 1 || Mx != (g->in_dhw[2] + 2 - 3) / S + 1) return pl;
  } else {
    if (S != 2 || g->out_dhw[1] != 2 * g->in_dhw[1] || g->out_dhw[2] != 2 * g->in_dhw[2]) return pl;
    Cd = g->cin; Cg = g->cout; My = g->in_dhw[1]; Mx = g->in_dhw[2];
  }
  // Blocks are capped so that the slabs (Cd x 9 x Cg floats per block) stay within ~4.7 MB: the reducer's time is
  // their traffic, and every block must still walk several tiles for the prefetch to pay.  32 -> 64 channels
  // (72 KB slabs: 64 blocks) is left to the K-stepped kernel.
  int inst = 0, tx = 16, cap = 512;
  if (S == 1 && Cd == 16 && Cg == 16) { inst = 1; cap = 512; }
  else if (S == 1 && Cd == 32 && Cg == 32) { inst = 2; cap = 128; }
  else if (S == 2 && Cd == 32 && Cg == 16) { inst = 3; cap = 256; }
  else if (S == 2 && Cd == 64 && Cg == 16) { inst = 5; tx = 8; cap = 128; }
  static const int cap_env = dev_env("MPGAN_DBG_WPATCH2D_BLOCKS") ? atoi(dev_env("MPGAN_DBG_WPATCH2D_BLOCKS")) : 0;
  if (cap_env > 0) cap = cap_env;
  if (!inst || My < 4 || Mx < 4) return pl;
  pl.tiles_y = (My + WP2_TY - 1) / WP2_TY;
  pl.tiles_x = (Mx + tx - 1) / tx;
  const long nt = (long)g->n * pl.tiles_y * pl.tiles_x;
  if (nt >= (1L << 31)) return pl;
  pl.ntiles = (int)nt;
  pl.blocks = (int)(nt < cap ? nt : cap);                      // persistent blocks, each walking its range of tiles
  pl.inst = inst; pl.Cd = Cd; pl.Cg = Cg;
  return pl;
}

template <int CD, int CG, int S, int TX>
static int launch_wgrad_p2(const WgradParams& p, const WP2Plan& pl, hipStream_t st) {
  using K = WP2<CD, CG, S, TX>;
  const WP2Grid tg{pl.tiles_y, pl.tiles_x, pl.ntiles};
  auto k1 = wgrad_patch2d_kernel<CD, CG, S, TX, true>;
  auto k0 = wgrad_patch2d_kernel<CD, CG, S, TX, false>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM);
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM);
    if (e1 != hipSuccess || e0 != hipSuccess) {
      set_error("wgrad_patch2d: hipFuncSetAttribute: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e0));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  if (p.pro.scale) hipLaunchKernelGGL(k1, dim3(pl.blocks), dim3(256), K::SMEM, st, p, tg);
  else hipLaunchKernelGGL(k0, dim3(pl.blocks), dim3(256), K::SMEM, st, p, tg);
  return check_launch("wgrad_patch2d");
}

struct ThinWgradPlan { int blocks; long chunk; bool ok; int rounds; };
static ThinWgradPlan plan_thin_wgrad(int Cd, int Cg, int T, long M, bool has_pro, long lds_cap = 64 * 1024) {
  ThinWgradPlan t;
  t.ok = Cg == 1 && !has_pro && Cd <= 64 && (T == 1 || T == 9 || T == 27);   // kernel shape checked by the caller
  // latency-bound pixel walk: short chunks keep every CU busy, but each block pays a fixed LDS fold
  // and adds a slab to the reducer (measured: 256-pixel chunks are slower than 2048-pixel ones)
  const long per = Cd * T < 64 ? 2048 : 1024;    // a handful of outputs: fewer, longer walks
  long blocks = (M + per - 1) / per;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  t.chunk = (M + blocks - 1) / blocks;
  t.blocks = (int)((M + t.chunk - 1) / t.chunk);
  // LDS: PL * (Cd*T + Cd) floats must fit 64 KiB
  const int V = Cd % 4 == 0 ? 4 : 1;
  const int CQ = (Cd + V - 1) / V;
  const int PL = 256 / CQ;
  t.rounds = 1;                                  // fold rounds of the lane rows (thin_wgrad_kernel): halve the LDS rows until they fit
  while (t.rounds < 4 && (PL / t.rounds) % 2 == 0 && (long)(PL / t.rounds) * (Cd * T + Cd) * 4 > lds_cap) t.rounds *= 2;
  if ((long)(PL / t.rounds) * (Cd * T + Cd) * 4 > lds_cap) t.ok = false;
  // measured at C5 (128^3, bs 4): 1 -> 16 stride 2 (16 dense channels, two rounds) 136 -> 110 us against the generic
  // MFMA kernel, but ConvTranspose 32 -> 1 (32 dense channels: 108 accumulators over only 32 pixel lanes) 141 -> 187 us
  if (t.rounds > 1 && Cd > 16) t.ok = false;
  return t;
}

struct WgradPlan {
  int BD, BG, nsplit, kw;
  long chunk;
  int tiles_c, tiles_d;
};

static WgradPlan plan_wgrad(int Cd, int NC, long M) {
  WgradPlan pl;
  pl.BD = Cd > 64 ? 128 : (Cd > 32 ? 64 : 32);
  pl.BG = NC > 64 ? 128 : (NC > 32 ? 64 : 32);
  pl.kw = 1;
  if (pl.BD <= 64 && pl.BG == 32) pl.kw = 4;       // (32|64) x 32: wave-split K
  else if (pl.BD == 32 && pl.BG == 64) pl.kw = 4;
  pl.tiles_c = (NC + pl.BG - 1) / pl.BG;
  pl.tiles_d = (Cd + pl.BD - 1) / pl.BD;
  long tiles = (long)pl.tiles_c * pl.tiles_d;
  // at most 1024 blocks = two full rounds of the 512 resident ones (2 per CU): rounding UP here left
  // D.conv2 (5 column tiles x 205 splits = 1025 blocks) with one straggler block after the second round
  long want = 1024 / tiles;
  if (want < 1) want = 1;
  long maxsplit = M / 256;                         // at least 8 K-steps per split (small maps need the blocks)
  if (maxsplit < 1) maxsplit = 1;
  long ns = want < maxsplit ? want : maxsplit;
  if (ns < 1) ns = 1;
  if (ns > 256) ns = 256;
  // Short blocks (the generator's 32^2 level: 8 K-steps each) must not spill a few blocks into a second round of the
  // 512 resident ones: 9 column tiles x 64 splits = 576 blocks ran as one full round + 64 stragglers (128 -> 128
  // at 32^2: 82 us for a 31 us contraction).  Between one and two rounds, take one.
  if (tiles * ns > 512 && tiles * ns < 1024 && tiles <= 512) ns = 512 / tiles;
  long chunk = (M + ns - 1) / ns;
  chunk = (chunk + WBK - 1) / WBK * WBK;
  ns = (M + chunk - 1) / chunk;
  if (ns < 1) ns = 1;
  pl.nsplit = (int)ns;
  pl.chunk = chunk;
  return pl;
}

template <int BD, int BG, int TM, int TN, int WN, int KW, bool SD, bool SG>
static int launch_wgrad_variant(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
  auto kern = wgrad_kernel<BD, BG, TM, TN, WN, KW, SD, SG>;
  constexpr int smem = 2 * WBK * (BD + BG) * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)pl.tiles_c * pl.tiles_d * pl.nsplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  return check_launch("wgrad");
}

template <int BD, int BG, int TM, int TN, int WN, int PRO, bool PAD>
static int launch_wgrad_pipe_variant(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
  auto kern = wgrad_pipe_kernel<BD, BG, TM, TN, WN, PRO, PAD>;
  constexpr int smem = 2 * WBK * (BD + BG) * (int)sizeof(float) + 2 * 32 * 16;   // + the row tables
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("wgrad_pipe: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)pl.tiles_c * pl.tiles_d * pl.nsplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  return check_launch("wgrad_pipe");
}

// PRO: 0 none, 1 per-channel norm + activation, 3 per-channel norm + LeakyReLU with a host-known
// slope in [0, 1] (the discriminator's layers; 128 x 128 tiles only, as is the pad-free variant).
template <int PRO>
static int dispatch_wgrad_pipe(const WgradParams& p, const WgradPlan& pl, hipStream_t st, bool& handled) {
  handled = true;
  const bool nopad = (p.pz | p.py | p.px) == 0;
  constexpr int P1 = PRO == 3 ? 1 : PRO;
  if (pl.BD == 128 && pl.BG == 128) {
    if (nopad) return launch_wgrad_pipe_variant<128, 128, 2, 2, 2, PRO, false>(p, pl, st);
    return launch_wgrad_pipe_variant<128, 128, 2, 2, 2, P1, true>(p, pl, st);
  }
  if (pl.BD == 128 && pl.BG == 64) return launch_wgrad_pipe_variant<128, 64, 1, 2, 1, P1, true>(p, pl, st);
  if (pl.BD == 64 && pl.BG == 128) return launch_wgrad_pipe_variant<64, 128, 2, 1, 4, P1, true>(p, pl, st);
  if (pl.BD == 64 && pl.BG == 64) return launch_wgrad_pipe_variant<64, 64, 1, 1, 2, P1, true>(p, pl, st);
  if (pl.BD == 32 && pl.BG == 128) return launch_wgrad_pipe_variant<32, 128, 1, 1, 4, P1, true>(p, pl, st);
  if (pl.BD == 128 && pl.BG == 32) return launch_wgrad_pipe_variant<128, 32, 1, 1, 1, P1, true>(p, pl, st);
  handled = false;
  return MPGAN_OK;
}

template <bool SD, bool SG>
static int dispatch_wgrad(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
  if (pl.BD == 128 && pl.BG == 128) return launch_wgrad_variant<128, 128, 2, 2, 2, 1, SD, SG>(p, pl, st);
  if (pl.BD == 128 && pl.BG == 64) return launch_wgrad_variant<128, 64, 1, 2, 1, 1, SD, SG>(p, pl, st);
  if (pl.BD == 128 && pl.BG == 32) return launch_wgrad_variant<128, 32, 1, 1, 1, 1, SD, SG>(p, pl, st);
  if (pl.BD == 64 && pl.BG == 128) return launch_wgrad_variant<64, 128, 2, 1, 4, 1, SD, SG>(p, pl, st);
  if (pl.BD == 64 && pl.BG == 64) return launch_wgrad_variant<64, 64, 1, 1, 2, 1, SD, SG>(p, pl, st);
  if (pl.BD == 64 && pl.BG == 32) return launch_wgrad_variant<64, 32, 2, 1, 1, 4, SD, SG>(p, pl, st);
  if (pl.BD == 32 && pl.BG == 128) return launch_wgrad_variant<32, 128, 1, 1, 4, 1, SD, SG>(p, pl, st);
  if (pl.BD == 32 && pl.BG == 64) return launch_wgrad_variant<32, 64, 1, 2, 1, 4, SD, SG>(p, pl, st);
  return launch_wgrad_variant<32, 32, 1, 1, 1, 4, SD, SG>(p, pl, st);
}

}  // namespace mpgan

using namespace mpgan;

static void wgrad_dims(const mpgan_conv_geom* g, int& Cd, int& Cg, int& T, long& M) {
  T = g->k[0] * g->k[1] * g->k[2];
  if (!g->transposed) {
    Cd = g->cout; Cg = g->cin;
    M = (long)g->n * g->out_dhw[0] * g->out_dhw[1] * g->out_dhw[2];
  } else {
    Cd = g->cin; Cg = g->cout;
    M = (long)g->n * g->in_dhw[0] * g->in_dhw[1] * g->in_dhw[2];
  }
}

extern "C" int64_t mpgan_conv_wgrad_workspace(const mpgan_conv_geom* g) {
  if (!g) return -1;
  int Cd, Cg, T;
  long M;
  wgrad_dims(g, Cd, Cg, T, M);
  WgradPlan pl = plan_wgrad(Cd, T * Cg, M);
  int64_t need = ((int64_t)pl.nsplit * pl.kw * Cd * T * Cg + (int64_t)pl.nsplit * Cd) * (int64_t)sizeof(float);
  ThinWgradPlan tp = plan_thin_wgrad(Cd, Cg, T, M, false);
  if (tp.ok) {
    const int64_t tneed = ((int64_t)tp.blocks * Cd * T + (int64_t)tp.blocks * Cd) * (int64_t)sizeof(float);
    if (tneed > need) need = tneed;
  }
  if (wgrad_p3_geom_ok(g)) {                     // 3-D patch form: one [16][432] slab + 16 bias sums per persistent block
    const int64_t pneed = (int64_t)wgrad_p3_blocks(g) * (16 * 432 + 16) * (int64_t)sizeof(float);
    if (pneed > need) need = pneed;
  }
  {                                              // 2-D patch form: one [Cd][9 * Cg] slab + Cd bias sums per persistent block
    const WP2Plan p2 = wgrad_p2_plan(g);
    if (p2.inst) {
      const int64_t pneed = (int64_t)p2.blocks * (p2.Cd * 9 * p2.Cg + p2.Cd) * (int64_t)sizeof(float);
      if (pneed > need) need = pneed;
    }
  }
  return need;
}

extern "C" int mpgan_conv_backward_weight(const mpgan_conv_geom* g, const float* x, int32_t ldx,
                                          const mpgan_prologue* pro, const float* dy, int32_t lddy, float* dw,
                                          float* dbias, float beta, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
  MPGAN_CHECK_ARG(g && x && dy && dw && workspace, "conv_backward_weight: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && lddy >= g->cout, "conv_backward_weight: bad pitch");
  int Cd, Cg, T;
  long M;
  wgrad_dims(g, Cd, Cg, T, M);
  MPGAN_CHECK_ARG(M < (1L << 31) - 64 &&
                      (long)g->n * g->in_dhw[0] * g->in_dhw[1] * g->in_dhw[2] < (1L << 31) &&
                      (long)g->n * g->out_dhw[0] * g->out_dhw[1] * g->out_dhw[2] < (1L << 31),
                  "conv_backward_weight: more than 2^31 pixels");
  WgradPlan pl = plan_wgrad(Cd, T * Cg, M);
  const int64_t slab_floats = (int64_t)pl.nsplit * pl.kw * Cd * T * Cg;
  const int64_t need = (slab_floats + (int64_t)pl.nsplit * Cd) * (int64_t)sizeof(float);
  MPGAN_CHECK_ARG(workspace_bytes >= need, "conv_backward_weight: workspace %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
  WgradParams p{};
  p.partial = static_cast<float*>(workspace);
  MPGAN_UNSUPPORTED(dbias && g->transposed,
                    "conv_backward_weight: fused bias gradient is for ConvNd only (dy is the gathered operand of a "
                    "transposed conv)");
  p.bias_partial = dbias ? p.partial + slab_floats : nullptr;
  p.Kz = g->k[0]; p.Ky = g->k[1]; p.Kx = g->k[2];
  p.sz = g->stride[0]; p.sy = g->stride[1]; p.sx = g->stride[2];
  p.pz = g->pad[0]; p.py = g->pad[1]; p.px = g->pad[2];
  p.N = g->n;
  p.nsplit = pl.nsplit; p.chunk = pl.chunk;
  p.tiles_c = pl.tiles_c; p.tiles_d = pl.tiles_d;
  if (!g->transposed) {
    p.dense = dy; p.ldd = lddy; p.Cd = g->cout;
    p.gath = x; p.ldg = ldx; p.Cg = g->cin;
    p.pro = make_pro(pro);
    p.Mz = g->out_dhw[0]; p.My = g->out_dhw[1]; p.Mx = g->out_dhw[2];
    p.Gz = g->in_dhw[0]; p.Gy = g->in_dhw[1]; p.Gx = g->in_dhw[2];
  } else {
    MPGAN_UNSUPPORTED(pro && pro->scale, "conv_backward_weight: prologue on a transposed conv input");
    p.dense = x; p.ldd = ldx; p.Cd = g->cin;
    p.gath = dy; p.ldg = lddy; p.Cg = g->cout;
    p.pro = make_pro(nullptr);
    p.Mz = g->in_dhw[0]; p.My = g->in_dhw[1]; p.Mx = g->in_dhw[2];
    p.Gz = g->out_dhw[0]; p.Gy = g->out_dhw[1]; p.Gx = g->out_dhw[2];
  }
  p.fMx = make_fastdiv(p.Mx); p.fMy = make_fastdiv(p.My); p.fMz = make_fastdiv(p.Mz);
  hipStream_t st0 = (hipStream_t)stream;
  if (wgrad_p3_geom_ok(g) && p.ldd % 4 == 0 && p.ldg % 4 == 0 && p.pro.n_stride == 0 &&
      ((reinterpret_cast<uintptr_t>(p.dense) | reinterpret_cast<uintptr_t>(p.gath)) & 15) == 0 &&
      (!p.pro.scale || ((reinterpret_cast<uintptr_t>(p.pro.scale) | reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0)) {
    const int nb = wgrad_p3_blocks(g);
    const int64_t pslab = (int64_t)nb * 16 * 432;
    MPGAN_CHECK_ARG(workspace_bytes >= (pslab + (int64_t)nb * 16) * (int64_t)sizeof(float),
                    "conv_backward_weight: workspace too small for the 3-D patch form");
    p.bias_partial = dbias ? p.partial + pslab : nullptr;
    const WP3Grid tg{(p.Mz + WP3_TZ - 1) / WP3_TZ, (p.My + WP3_TY - 1) / WP3_TY, (p.Mx + WP3_TX - 1) / WP3_TX};
    static const bool no_mm16 = dev_env("MPGAN_DBG_NO_MM16") != nullptr;
    if ((g->flags & MPGAN_CONV_MM_BF16) && !no_mm16) {
      if (p.pro.scale) hipLaunchKernelGGL((wgrad_patch3d_c16_kernel<true, true>), dim3(nb), dim3(256), 0, st0, p, tg);
      else hipLaunchKernelGGL((wgrad_patch3d_c16_kernel<false, true>), dim3(nb), dim3(256), 0, st0, p, tg);
    } else if (p.pro.scale) hipLaunchKernelGGL(wgrad_patch3d_c16_kernel<true>, dim3(nb), dim3(256), 0, st0, p, tg);
    else hipLaunchKernelGGL(wgrad_patch3d_c16_kernel<false>, dim3(nb), dim3(256), 0, st0, p, tg);
    int rcp = check_launch("wgrad_patch3d_c16");
    if (rcp) return rcp;
    launch_wgrad_reduce(st0, p.partial, dw, nb, 16, 16, 27, beta, p.bias_partial, nb, dbias);
    return check_launch("wgrad_patch3d_reduce");
  }
  {
    const WP2Plan p2 = wgrad_p2_plan(g);
    if (p2.inst && p.ldd % 4 == 0 && p.ldg % 4 == 0 && p.pro.n_stride == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dense) | reinterpret_cast<uintptr_t>(p.gath)) & 15) == 0 &&
        (!p.pro.scale || ((reinterpret_cast<uintptr_t>(p.pro.scale) | reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0)) {
      const int64_t pslab = (int64_t)p2.blocks * Cd * 9 * Cg;
      MPGAN_CHECK_ARG(workspace_bytes >= (pslab + (int64_t)p2.blocks * Cd) * (int64_t)sizeof(float),
                      "conv_backward_weight: workspace too small for the 2-D patch form");
      p.bias_partial = dbias ? p.partial + pslab : nullptr;
      int rcp;
      switch (p2.inst) {
        case 1: rcp = launch_wgrad_p2<16, 16, 1, 16>(p, p2, st0); break;
        case 2: rcp = launch_wgrad_p2<32, 32, 1, 16>(p, p2, st0); break;
        case 3: rcp = launch_wgrad_p2<32, 16, 2, 16>(p, p2, st0); break;
        default: rcp = launch_wgrad_p2<64, 16, 2, 8>(p, p2, st0); break;
      }
      if (rcp) return rcp;
      launch_wgrad_reduce(st0, p.partial, dw, p2.blocks, Cd, Cg, 9, beta, p.bias_partial, p2.blocks, dbias);
      return check_launch("wgrad_patch2d_reduce");
    }
  }
  {
    ThinWgradPlan tp = plan_thin_wgrad(Cd, Cg, T, M, p.pro.scale != nullptr);
    const bool v4 = (p.Cd % 4 == 0) && (p.ldd % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.dense) & 15) == 0);
    static const bool no_thin = dev_env("MPGAN_DBG_NO_THIN") != nullptr;
    const bool kshape = (T == 1) || (T == 9 && p.Kz == 1 && p.Ky == 3 && p.Kx == 3) ||
                        (T == 27 && p.Kz == 3 && p.Ky == 3 && p.Kx == 3);
    if (tp.ok && kshape && !no_thin) {
      const int64_t tslab = (int64_t)tp.blocks * Cd * T;
      MPGAN_CHECK_ARG(workspace_bytes >= (tslab + (int64_t)tp.blocks * Cd) * (int64_t)sizeof(float),
                      "conv_backward_weight: workspace too small for the thin path");
      p.chunk = tp.chunk;
      p.bias_partial = dbias ? p.partial + tslab : nullptr;
      const int V = v4 ? 4 : 1;
      const int CQ = (Cd + V - 1) / V, PL = 256 / CQ;
      // (the plan assumed V = 4 where Cd % 4 == 0; an unaligned operand walks scalar lanes: more lane rows, same rule)
      int rounds = 1;
      while (rounds < 8 && (PL / rounds) % 2 == 0 && (size_t)(PL / rounds) * (Cd * T + Cd) * sizeof(float) > 64 * 1024) rounds *= 2;
      p.fold_rounds = rounds;
      const size_t smem = (size_t)(PL / rounds) * (Cd * T + Cd) * sizeof(float);
      MPGAN_UNSUPPORTED(smem > 64 * 1024, "conv_backward_weight: thin path lane rows exceed 64 KiB");
      dim3 grid(tp.blocks);
#define THIN_LAUNCH(VV, TT) hipLaunchKernelGGL((thin_wgrad_kernel<VV, TT>), grid, dim3(256), smem, st0, p)
      if (thin_rows_ok(p, T)) launch_thin_rows<false>(p, T, tp.blocks, st0);
      else if (v4) { if (T == 1) THIN_LAUNCH(4, 1); else if (T == 9) THIN_LAUNCH(4, 9); else THIN_LAUNCH(4, 27); }
      else    { if (T == 1) THIN_LAUNCH(1, 1); else if (T == 9) THIN_LAUNCH(1, 9); else THIN_LAUNCH(1, 27); }
#undef THIN_LAUNCH
      int rc0 = check_launch("thin_wgrad");
      if (rc0) return rc0;
      launch_wgrad_reduce(st0, p.partial, dw, tp.blocks, Cd, Cg, T, beta, p.bias_partial, tp.blocks, dbias);
      return check_launch("thin_wgrad_reduce");
    }
  }
  const bool vd = (p.Cd % 4 == 0) && (p.ldd % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.dense) & 15) == 0);
  const bool vg = (p.Cg % 4 == 0) && (p.ldg % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.gath) & 15) == 0) &&
                  (!p.pro.scale || (((reinterpret_cast<uintptr_t>(p.pro.scale) |
                                      reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0 &&
                                    p.pro.n_stride % 4 == 0));
  hipStream_t st = (hipStream_t)stream;
  int rc = MPGAN_OK;
  bool handled = false;
  static const bool no_pipe = dev_env("MPGAN_DBG_NO_PIPE") != nullptr;
  // the pipelined kernel addresses each operand as base + unsigned 32-bit byte offset
  const bool small = (long)M * p.ldd * 4 < (1L << 32) && (long)p.N * p.Gz * p.Gy * p.Gx * p.ldg * 4 < (1L << 32);
  if ((g->flags & MPGAN_CONV_MM_BF16) && vd && vg && pl.kw == 1 && small && p.pro.n_stride == 0)
    rc = launch_wgrad_mm16(p, pl.BD, pl.BG, st, handled);                // bf16 matrix operands (conv_mm16.hip)
  if (!handled && vd && vg && pl.kw == 1 && small && p.pro.n_stride == 0 && !no_pipe) {
    const bool fast_leaky = p.pro.scale && p.pro.act == MPGAN_ACT_LEAKY && !p.pro.slope_ptr && p.pro.slope >= 0.f &&
                            p.pro.slope <= 1.f;
    rc = !p.pro.scale ? dispatch_wgrad_pipe<0>(p, pl, st, handled)
         : fast_leaky ? dispatch_wgrad_pipe<3>(p, pl, st, handled)
                      : dispatch_wgrad_pipe<1>(p, pl, st, handled);
  }
  if (handled) { /* done */ }
  else if (vd && vg) rc = dispatch_wgrad<false, false>(p, pl, st);
  else if (vd) rc = dispatch_wgrad<false, true>(p, pl, st);
  else if (vg) rc = dispatch_wgrad<true, false>(p, pl, st);
  else rc = dispatch_wgrad<true, true>(p, pl, st);
  if (rc) return rc;
  launch_wgrad_reduce(st, p.partial, dw, pl.nsplit * pl.kw, Cd, Cg, T, beta, p.bias_partial, pl.nsplit, dbias);
  return check_launch("wgrad_reduce");
}

// Weight gradient of a 1 -> C ConvNd whose output gradient is stored as bf16 (D.conv1 in the bf16 path):
// the thin HBM-bound reduction, dy converted on load; dW / dbias and the slabs stay fp32.
extern "C" int mpgan_conv_backward_weight_bf16dy(const mpgan_conv_geom* g, const float* x, int32_t ldx, const void* dy,
                                                 int32_t lddy, float* dw, float* dbias, float beta, void* workspace,
                                                 int64_t workspace_bytes, void* stream) {
  MPGAN_CHECK_ARG(g && x && dy && dw && workspace, "conv_backward_weight_bf16dy: null pointer");
  MPGAN_UNSUPPORTED(g->transposed || g->cin != 1, "conv_backward_weight_bf16dy: ConvNd with one input channel only");
  int Cd, Cg, T;
  long M;
  wgrad_dims(g, Cd, Cg, T, M);
  MPGAN_CHECK_ARG(M < (1L << 31) - 64, "conv_backward_weight_bf16dy: more than 2^31 pixels");
  ThinWgradPlan tp = plan_thin_wgrad(Cd, Cg, T, M, false, 150 * 1024);   // 3-D 1 -> 64: 112 KiB of lane rows, one block per CU
  const bool kshape = (T == 1) || (T == 9 && g->k[0] == 1) || (T == 27 && g->k[0] == 3 && g->k[1] == 3 && g->k[2] == 3);
  MPGAN_UNSUPPORTED(!tp.ok || !kshape || Cd % 4 != 0 || lddy % 4 != 0 || (reinterpret_cast<uintptr_t>(dy) & 7),
                    "conv_backward_weight_bf16dy: thin path only (Cout %% 4 == 0, <= 64 channels, 1 / 3x3 / 3x3x3 kernel)");
  const int64_t tslab = (int64_t)tp.blocks * Cd * T;
  MPGAN_CHECK_ARG(workspace_bytes >= (tslab + (int64_t)tp.blocks * Cd) * (int64_t)sizeof(float),
                  "conv_backward_weight_bf16dy: workspace too small");
  WgradParams p{};
  p.partial = static_cast<float*>(workspace);
  p.bias_partial = dbias ? p.partial + tslab : nullptr;
  p.Kz = g->k[0]; p.Ky = g->k[1]; p.Kx = g->k[2];
  p.sz = g->stride[0]; p.sy = g->stride[1]; p.sx = g->stride[2];
  p.pz = g->pad[0]; p.py = g->pad[1]; p.px = g->pad[2];
  p.N = g->n;
  p.chunk = tp.chunk;
  p.dense = static_cast<const float*>(dy); p.ldd = lddy; p.Cd = Cd; p.dense_bf16 = 1;
  p.gath = x; p.ldg = ldx; p.Cg = 1;
  p.pro = make_pro(nullptr);
  p.Mz = g->out_dhw[0]; p.My = g->out_dhw[1]; p.Mx = g->out_dhw[2];
  p.Gz = g->in_dhw[0]; p.Gy = g->in_dhw[1]; p.Gx = g->in_dhw[2];
  p.fMx = make_fastdiv(p.Mx); p.fMy = make_fastdiv(p.My); p.fMz = make_fastdiv(p.Mz);
  hipStream_t st = (hipStream_t)stream;
  const int CQ = Cd / 4, PL = 256 / CQ;
  const size_t smem = (size_t)PL * (Cd * T + Cd) * sizeof(float);
  dim3 grid(tp.blocks);
  if (thin_rows_ok(p, T)) {
    launch_thin_rows<true>(p, T, tp.blocks, st);
  } else {
  if (smem > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(thin_wgrad_kernel<4, 27, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      if (e != hipSuccess) { set_error("thin_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MPGAN_ERR_HIP; }
      attr_set = true;
    }
    MPGAN_UNSUPPORTED(T != 27, "conv_backward_weight_bf16dy: more than 64 KiB of lane rows outside the 3x3x3 case");
  }
  if (T == 1) hipLaunchKernelGGL((thin_wgrad_kernel<4, 1, true>), grid, dim3(256), smem, st, p);
  else if (T == 9) hipLaunchKernelGGL((thin_wgrad_kernel<4, 9, true>), grid, dim3(256), smem, st, p);
  else hipLaunchKernelGGL((thin_wgrad_kernel<4, 27, true>), grid, dim3(256), smem, st, p);
  }
  int rc = check_launch("thin_wgrad_bf16dy");
  if (rc) return rc;
  const long total = (long)Cd * T;
  const int blocks = (int)((total + 31) / 32);
  const int bb = dbias ? (Cd + 31) / 32 : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks + bb), dim3(256), 0, st, p.partial, dw, tp.blocks, Cd, 1, T, beta,
                     p.bias_partial, tp.blocks, dbias, bb);
  return check_launch("thin_wgrad_bf16dy_reduce");
}

extern "C" int64_t mpgan_conv_wgrad_workspace_bf16dy(const mpgan_conv_geom* g) {
  if (!g) return -1;
  int Cd, Cg, T;
  long M;
  wgrad_dims(g, Cd, Cg, T, M);
  ThinWgradPlan tp = plan_thin_wgrad(Cd, Cg, T, M, false, 150 * 1024);
  if (!tp.ok) return -1;
  return ((int64_t)tp.blocks * Cd * T + (int64_t)tp.blocks * Cd) * (int64_t)sizeof(float);
}
