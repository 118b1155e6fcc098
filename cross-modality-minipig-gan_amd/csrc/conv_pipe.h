// The K-stepped software-pipelined gather kernel and the epilogue the K-stepped kernels share, as templates:
// conv_igemm.hip instantiates the fp32-MFMA forms, conv_mm16.hip the forms whose matrix operands are rounded to
// bf16 on their way into LDS (fp32 storage in HBM, fp32 accumulation; config C5's generator) -- two translation
// units so that they compile side by side.
#pragma once
#include "mpgan_common.h"
#include "conv_geom.h"
#include "lds_dma.h"

namespace mpgan {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int PITCH = BK + 4;

// Shared epilogue: row -> output pixel map through LDS, bias / residual / tanh,
// optional fused BatchNorm statistics.  Called after the K-loop's final barrier.
template <int BN, int TM, int TN, int WN>
__device__ __forceinline__ void conv_epilogue(const GatherConv& p, const Phase& ph, f32x16 (&acc)[TM][TN], float* lds,
                                              long m0, int n0, long Mtot, int stats_row, bool zero_rows = false,
                                              int tid = threadIdx.x, bool active = true, const float* bias_pre = nullptr) {
  // `tid` is the thread's index inside its 256-thread K group; only the group with `active` holds the tile's
  // sums and writes anything (the in-block split-K form of the pipelined kernel calls this from every group so
  // that the barriers below are reached by all waves).
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int Cout = p.Cout;
  const int ksplit_id = p.ksplit > 1 ? (int)(xcd_remap(blockIdx.x, gridDim.x) / (unsigned)(p.ntiles * p.mtiles * (p.packed ? 1 : p.nphase))) : 0;
  // ---- epilogue: row -> output pixel map through LDS, then bias/resid/tanh ----
  int* rowpix = reinterpret_cast<int*>(lds);
  if (active && tid < BM) {
    const unsigned m = (unsigned)m0 + tid;
    int pix = -1;
    if (m < (unsigned)Mtot) {
      unsigned q, umx, umy, umz;
      fdivmod(m, ph.fMx, q, umx);
      fdivmod(q, ph.fMy, q, umy);
      fdivmod(q, ph.fMz, q, umz);
      const int mx = (int)umx, my = (int)umy, mz = (int)umz;
      const int n = (int)q;
      int oz = mz * p.ostride[0] + ph.oz, oy = my * p.ostride[1] + ph.oy, ox = mx * p.ostride[2] + ph.ox;
      if (oz < p.Do && oy < p.Ho && ox < p.Wo) pix = ((n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    }
    rowpix[tid] = pix;
  }
  __syncthreads();
  MPGAN_STAMP(p, 4);    // epilogue: row -> pixel map ready
  if (p.ksplit > 1) {   // split-K: raw partial sums, reduced (with bias) by splitk_reduce_kernel
    float* part = p.kpartial + (long)ksplit_id * ((long)p.N * p.Do * p.Ho * p.Wo) * Cout;
    if (active)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int co = n0 + (wn * TN + tn) * 32 + li;
        if (co >= Cout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int pix = rowpix[row];
          if (pix >= 0) part[(long)pix * Cout + co] = acc[tm][tn][r];
        }
      }
    return;
  }
  const float* gres = p.resid;
  float* gout = p.out;
  const int ldo = p.ldo, ldr = p.ldr, tanh_out = p.tanh_out;
  float bv[TN];
  int cov[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    cov[tn] = n0 + (wn * TN + tn) * 32 + li;
    bv[tn] = bias_pre ? bias_pre[tn] : ((p.bias && cov[tn] < Cout) ? p.bias[cov[tn]] : 0.f);   // bias_pre: fetched before the K loop
  }
  // epilogue activation (eval-mode inference): this lane's columns' affine and slope
  const bool ea = p.epi.scale != nullptr;
  float esc[TN], esh[TN], esl[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const bool okc = ea && cov[tn] < Cout;
    esc[tn] = okc ? p.epi.scale[cov[tn]] : 1.f;
    esh[tn] = okc ? p.epi.shift[cov[tn]] : 0.f;
    esl[tn] = okc ? p.epi.slope[cov[tn]] : 1.f;
  }
  // fused norm-backward sums of the produced gradient (BwdStats): per-column vectors and running sums
  const bool bw = p.bwd.part != nullptr;
  // Output path A (the usual one): every 32 x 32 accumulator tile goes through a wave-private LDS transpose and leaves
  // as 16-byte stores -- a lane holds ONE channel of 16 rows, so direct stores are 4 bytes per lane and 16 store
  // instructions per tile; the phase stamps show 3-6.5 us of every K-stepped launch in issuing them (all blocks end
  // their K loops together and the scalar stores queue up).  Path B (below): the element-wise walk, kept for the
  // BwdStats form (its sums need each element beside its z) and for unaligned / odd-width outputs.
  const bool vec_out = !bw && (Cout % 4 == 0) && (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(gout) & 15) == 0) &&
                       (!gres || ((ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(gres) & 15) == 0));
  if (vec_out) {
    constexpr int TP = 36;                                 // pitch of the transpose tile: conflict-free both ways
    float* wt = lds + 2048 + wid * 32 * TP;                // clear of rowpix (ints 0..511) and the statistics area (1024..)
    if (active) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[tm][tn][r] + bv[tn];
            if (ea) v = epi_act1(v, esc[tn], esh[tn], esl[tn]);
            wt[((r & 3) + 8 * (r >> 2) + 4 * lh) * TP + li] = v;
          }
          const int c4 = lane & 7, co = n0 + (wn * TN + tn) * 32 + 4 * c4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = (lane >> 3) + 8 * k;
            const int pix = rowpix[(wm * TM + tm) * 32 + row];
            float4 v = *reinterpret_cast<const float4*>(wt + row * TP + 4 * c4);
            if (pix >= 0 && co < Cout) {
              if (gres) {
                const float4 rr = *reinterpret_cast<const float4*>(gres + (long)pix * ldr + co);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
              }
              if (tanh_out) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
              *reinterpret_cast<float4*>(gout + (long)pix * ldo + co) = v;
            }
          }
        }
    }
  }
  float bsc[TN], bsh[TN], bmu[TN], bis[TN], b1[TN], b2[TN], b3[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const bool okc = bw && cov[tn] < Cout;
    bsc[tn] = okc ? p.bwd.scale[cov[tn]] : 0.f;
    bsh[tn] = okc ? p.bwd.shift[cov[tn]] : 0.f;
    bmu[tn] = okc ? p.bwd.mean[cov[tn]] : 0.f;
    bis[tn] = okc ? p.bwd.invstd[cov[tn]] : 0.f;
    b1[tn] = b2[tn] = b3[tn] = 0.f;
  }
  const bool bleaky = p.bwd.leaky != 0;
  const float bslope = p.bwd.slope;
  // row-major walk: the 64-bit pixel offset is formed once per row, not once per element
  if (active && !vec_out)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    // BwdStats: the tile's z values are fetched in ONE batch in front of the stores (loads between the stores could
    // not be hoisted over them -- the compiler cannot rule out aliasing -- and would each wait out a full round trip)
    float zv[16][TN];
    if (bw) {
      const float* __restrict__ zb = p.bwd.z;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int pix = rowpix[row];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          zv[r][tn] = (pix >= 0 && cov[tn] < Cout) ? zb[(long)pix * p.bwd.ldz + cov[tn]] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int pix = rowpix[row];
      if (pix < 0) continue;
      float* orow = gout + (long)pix * ldo;
      const float* rrow = gres ? gres + (long)pix * ldr : nullptr;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if (cov[tn] >= Cout) continue;
        float v = acc[tm][tn][r] + bv[tn];
        if (ea) v = epi_act1(v, esc[tn], esh[tn], esl[tn]);
        if (rrow) v += rrow[cov[tn]];
        if (tanh_out) v = tanhf(v);
        orow[cov[tn]] = v;
        if (bw) {                            // as norm_bwd_reduce_kernel (norm_ops.hip), element by element
          const float zz = zv[r][tn];
          const float y = zz * bsc[tn] + bsh[tn];
          const float zh = (zz - bmu[tn]) * bis[tn];
          const bool neg = bleaky && y < 0.f;
          const float gy = neg ? v * bslope : v;
          b1[tn] += gy;
          b2[tn] += gy * zh;
          b3[tn] += neg ? v * y : 0.f;
        }
      }
    }
  }
  MPGAN_STAMP(p, 5);    // epilogue: output stores issued
  if (bw) {
    // the two half-waves (shuffle), then the WM waves sharing a column range (LDS, fixed order): no atomics
    constexpr int WMB = 4 / WN;
    float* stb = lds + 1024;                // [WMB][3][BN], clear of rowpix
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = (wn * TN + tn) * 32 + li;
      b1[tn] += __shfl_xor(b1[tn], 32, 64);
      b2[tn] += __shfl_xor(b2[tn], 32, 64);
      b3[tn] += __shfl_xor(b3[tn], 32, 64);
      if (active && lh == 0) {
        stb[(wm * 3 + 0) * BN + col] = b1[tn];
        stb[(wm * 3 + 1) * BN + col] = b2[tn];
        stb[(wm * 3 + 2) * BN + col] = b3[tn];
      }
    }
    __syncthreads();
    if (active && tid < BN && n0 + tid < Cout) {
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int w = 0; w < WMB; ++w) {
        t1 += stb[(w * 3 + 0) * BN + tid];
        t2 += stb[(w * 3 + 1) * BN + tid];
        t3 += stb[(w * 3 + 2) * BN + tid];
      }
      float* row = p.bwd.part + (long)stats_row * 3 * Cout;
      row[n0 + tid] = t1;
      row[Cout + n0 + tid] = t2;
      row[2 * Cout + n0 + tid] = t3;
    }
  }
  if (p.stats || p.stats_acc) {
    // Fused BatchNorm statistics of z = acc + bias over this tile's valid rows.  Rows without
    // an output pixel carry acc == 0 exactly (their A rows are zero-filled), so the raw column
    // sums S1 = sum(acc), S2 = sum(acc^2) need no row test; the bias enters in closed form,
    //   sum(z) = S1 + nv*b,  sum(z^2) = S2 + 2*b*S1 + nv*b^2   (nv = rows with a pixel).
    // Registers, then the two half-waves (shuffle), then the WM waves sharing a column range
    // (LDS, fixed order): deterministic, no atomics.
    constexpr int WM = 4 / WN;
    float* st = lds + 1024;                 // [WM][2][BN], clear of rowpix
    int* nvp = reinterpret_cast<int*>(lds) + 512;
    if (zero_rows && active) {              // FAST kernels gather clamped (non-zero) rows past the last pixel
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (rowpix[row] < 0) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = 0.f;
          }
        }
    }
    if (active && tid < BM) {
      const unsigned long long b = __ballot(rowpix[tid] >= 0);
      if (lane == 0) nvp[wid] = __popcll(b);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = (wn * TN + tn) * 32 + li;
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[tm][tn][r];
          sm += v;
          sq = fmaf(v, v, sq);
        }
      sm += __shfl_xor(sm, 32, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (active && lh == 0) {
        st[(wm * 2 + 0) * BN + col] = sm;
        st[(wm * 2 + 1) * BN + col] = sq;
      }
    }
    __syncthreads();
    if (active && tid < BN && n0 + tid < Cout) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        sm += st[(w * 2 + 0) * BN + tid];
        sq += st[(w * 2 + 1) * BN + tid];
      }
      const float nv = (float)(nvp[0] + nvp[1]);
      const float b = p.bias ? p.bias[n0 + tid] : 0.f;
      if (p.stats_acc) {
        long long* rep = p.stats_acc + (long)(blockIdx.x % (unsigned)p.acc_rep) * ACC_WORDS * Cout;
        acc_add(rep, Cout, 0, n0 + tid, sm + nv * b);
        acc_add(rep, Cout, 2, n0 + tid, sq + b * (2.f * sm + nv * b));
      } else {
      float* row = p.stats + (long)stats_row * 2 * Cout;
      row[n0 + tid] = sm + nv * b;
      row[Cout + n0 + tid] = sq + b * (2.f * sm + nv * b);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Software-pipelined main kernel (vector path, Cin % 32 == 0 or Cin in {4,8,16}).
// Same tiles and LDS image as gather_conv_kernel, but the K-step is ONE basic
// block: cursor advance is branch-free, the next tile's address math + global
// loads are interleaved with the MFMAs of fragment group 0 and its
// normalise/mask + LDS stores with the MFMAs of group 3 (sched_group_barrier),
// so a wave keeps the matrix pipe fed without relying on a partner wave.
//   WRAPS : taps crossed per K-step (1 for Cin % 32 == 0, 32/Cin below 32)
//   PRO   : 0 none, 1 per-channel scale/shift (BatchNorm), 2 per-(n,c) (InstanceNorm),
//           3 per-channel + LeakyReLU whose slope the host knows to lie in [0, 1]
// ---------------------------------------------------------------------------
//   FAST  : pad-free forward gather with Cout % BN == 0 (the discriminator's valid convs): every tap of
//           every pixel is in range, so the per-row range tests and the zero masks of both operands
//           (84 of the K-step's 146 vector instructions) are dropped; rows past the last pixel gather
//           the last pixel instead of zeros and are cleared before the fused statistics.
//   KS    : in-block split of the K axis over KS groups of 4 waves (own LDS stages each; sums folded through LDS
//           before the epilogue).  For layers whose output grid yields about one block per CU (the U-Net's
//           32 x 32 levels: 128 pixel tiles x 2 channel tiles) a single 4-wave block leaves each SIMD with ONE
//           wave, and every LDS / barrier / load stall of that wave is a stall of the matrix pipe; KS = 2 puts a
//           second, independent wave on each SIMD without adding a launch or HBM traffic.
//   MM16  : matrix operands rounded to bf16 on their way into LDS, contraction on v_mfma_f32_32x32x16_bf16 (16x the
//           fp32 matrix rate), fp32 accumulation; HBM tensors, prologue arithmetic, bias / residual / statistics
//           stay fp32.  For the generator's 3-D layers at config C5, which run at 0.45-0.62 of the fp32 matrix peak
//           (arithmetic-bound: profiles/r03_c5_g*_calls.txt).  LDS rows are then [row][32 bf16 + 8 pad] (80 bytes:
//           the ds_read_b128 of a lane's 8 consecutive K is conflict-free, chunk index 5 r mod 16), a K-step is two
//           MFMAs per 32 x 32 tile, and the loop is bound by its staging work, not by the matrix pipe.
typedef __bf16 mm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mm_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int MM16_PITCHB = 80;                 // bytes per LDS row of the MM16 form
constexpr int CONV_EPI_FLOATS = 2048 + 4 * 32 * 36 + 1024;   // conv_epilogue's LDS scratch (row map, statistics, transpose tiles)

template <int BN, int TM, int TN, int WN, int WRAPS, int PRO, bool FAST = false, int KS = 1, bool MM16 = false>
__global__ __launch_bounds__(256 * KS) void gather_conv_pipe_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  constexpr int STAGE = MM16 ? (BM + BN) * (MM16_PITCHB / 4) : (BM + BN) * PITCH;   // floats
  constexpr int BROWS = BN / 32;
  constexpr int NMF = 4 * TM * TN;          // MFMAs per fragment group

  const int tid = threadIdx.x & 255;        // index inside the K group
  const int kg = KS > 1 ? (int)(threadIdx.x >> 8) : 0;
  float* lds = lds_all + kg * 2 * STAGE;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  MPGAN_STAMP(p, 0);
  MPGAN_STAMP_VALUE(p, 6, 1);                 // kernel kind: K-stepped pipeline
  const BlockId bid = conv_block_id(p);
  const Phase ph = p.ph[bid.phase];
  const long Mtot = (long)p.N * ph.Mz * ph.My * ph.Mx;
  const long m0 = (long)bid.mt * BM;
  const int n0 = bid.nt * BN;
  const int stats_row = bid.row;
  if (m0 >= Mtot) {
    if (p.stats && kg == 0 && tid < BN && n0 + tid < p.Cout) {
      float* row = p.stats + (long)stats_row * 2 * p.Cout;
      row[n0 + tid] = 0.f;
      row[p.Cout + n0 + tid] = 0.f;
    }
    if (p.bwd.part && kg == 0 && tid < BN && n0 + tid < p.Cout) {
      float* row = p.bwd.part + (long)stats_row * 3 * p.Cout;
      row[n0 + tid] = 0.f;
      row[p.Cout + n0 + tid] = 0.f;
      row[2 * p.Cout + n0 + tid] = 0.f;
    }
    return;
  }
  const int Cin = p.Cin, Cout = p.Cout, Di = p.Di, Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  const int ntaps = ph.nz * ph.ny * ph.nx;
  const int Kp = ntaps * Cin;
  const int nk_all = (Kp + BK - 1) / BK;
  const int nk_per = (nk_all + p.ksplit * KS - 1) / (p.ksplit * KS);
  const int kt_begin = (bid.split * KS + kg) * nk_per;  // split-K: this K group's K-step range
  const int nk = nk_all - kt_begin < nk_per ? (nk_all - kt_begin > 0 ? nk_all - kt_begin : 0) : nk_per;
  const long Ktot = (long)p.Kz * p.Ky * p.Kx * Cin;
  const float* __restrict__ gin = p.in;
  const float* __restrict__ gw = p.wp;
  const float* __restrict__ gscale = p.pro.scale;
  const float* __restrict__ gshift = p.pro.shift;
  const int nstride = p.pro.n_stride, act = p.pro.act;
  const float slope = PRO ? pro_slope(p.pro) : 1.f;

  const int cc = tid & 7, r0 = tid >> 3;
  // Gathered rows of this thread.  All global addressing below is "uniform base + unsigned
  // 32-bit BYTE offset" (host guarantees < 4 GiB per operand): one add per address, no
  // 64-bit multiplies in the K-step.
  int rn[4], rz[4], ry[4], rx[4];
  unsigned rbB[4];                       // byte offset of (row pixel, channel 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned m = (unsigned)m0 + r0 + 32 * i;
    if constexpr (FAST) m = m < (unsigned)Mtot ? m : (unsigned)Mtot - 1u;
    if (m < (unsigned)Mtot) {
      unsigned q, umx, umy, umz;
      fdivmod(m, ph.fMx, q, umx);
      fdivmod(q, ph.fMy, q, umy);
      fdivmod(q, ph.fMz, q, umz);
      const int mx = (int)umx, my = (int)umy, mz = (int)umz;
      rn[i] = (int)q;
      rz[i] = mz * p.istride[0];
      ry[i] = my * p.istride[1];
      rx[i] = mx * p.istride[2];
    } else {
      rn[i] = 0;
      rz[i] = ry[i] = rx[i] = -(1 << 28);
    }
    rbB[i] = (unsigned)(((rn[i] * Di + rz[i]) * Hi + ry[i]) * Wi + rx[i]) * (unsigned)ldi * 4u;
  }
  // B rows of this thread: byte offsets of (co, k = 0), clamped for co >= Cout
  unsigned wrowB[BROWS];
  unsigned bvalid = 0;
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int co = n0 + r0 + 32 * i;
    const bool ok = co < Cout;
    wrowB[i] = ok ? (unsigned)co * (unsigned)Ktot * 4u : 0u;
    bvalid |= (ok ? 1u : 0u) << i;
  }
  const int ksz = p.kstep[0], ksy = p.kstep[1], ksx = p.kstep[2];
  const int dsz = p.dstep[0], dsy = p.dstep[1], dsx = p.dstep[2];
  const int Ky = p.Ky, Kx = p.Kx;
  const char* __restrict__ ginb = reinterpret_cast<const char*>(gin);
  const char* __restrict__ gwb = reinterpret_cast<const char*>(gw);
  const char* __restrict__ gscb = reinterpret_cast<const char*>(gscale);
  const char* __restrict__ gshb = reinterpret_cast<const char*>(gshift);

  // K cursor.  WRAPS == 1 (Cin % 32 == 0): a K-step lies inside ONE tap, so the tap walk
  // is wave-uniform and lives in scalar registers (SALU); only the channel differs per
  // thread (uci + 4*cc).  WRAPS > 1: taps differ between the threads of a K-step.
  constexpr bool UCUR = (WRAPS == 1);
  int ci, jz, jy, jx, dz, dy, dx, woff;
  int cmask;                              // -1 while the tap index is inside this phase's tap list, else 0
  unsigned deltaB;                        // byte offset of the tap relative to the row's base pixel
  auto place = [&]() {
    cmask = UCUR ? (ci < Cin ? -1 : 0) : (jz < ph.nz ? -1 : 0);
    const int kz = ph.kz0 + ksz * jz, ky = ph.ky0 + ksy * jy, kx = ph.kx0 + ksx * jx;
    woff = ((kz * Ky + ky) * Kx + kx) * Cin;
    dz = ph.dz0 + dsz * jz;
    dy = ph.dy0 + dsy * jy;
    dx = ph.dx0 + dsx * jx;
    deltaB = (unsigned)((dz * Hi + dy) * Wi + dx) * (unsigned)ldi * 4u;
  };
  // K order.  UCUR: channel-chunk major, taps inner -- the taps that re-read an input element
  // ((ky,kx) neighbours, and the rows shared with the tile above/below) are then a few K-steps
  // apart instead of Cin/32 times that, close enough for the 4 MiB per-XCD L2 to still hold them.
  // Otherwise tap major (a K-step spans several taps).
  {
    int tap;
    if constexpr (UCUR) {
      const int nt = ntaps > 0 ? ntaps : 1;
      const int chunk = kt_begin / nt;
      tap = kt_begin - chunk * nt;
      ci = chunk * BK;
    } else {
      const int kidx = kt_begin * BK + cc * 4;
      tap = kidx / Cin;
      ci = kidx - tap * Cin;
    }
    jx = tap % ph.nx;
    const int tq = tap / ph.nx;
    jy = tq % ph.ny;
    jz = tq / ph.ny;
    if constexpr (UCUR) {                 // computed from block-uniform values: pin them to SGPRs
      ci = __builtin_amdgcn_readfirstlane(ci);
      jx = __builtin_amdgcn_readfirstlane(jx);
      jy = __builtin_amdgcn_readfirstlane(jy);
      jz = __builtin_amdgcn_readfirstlane(jz);
    }
    place();
  }
  auto advance = [&]() {
    if constexpr (UCUR) {
      jx += 1;
      const int wx = jx == ph.nx ? 1 : 0;
      jx -= (-wx) & ph.nx;
      jy += wx;
      const int wy = jy == ph.ny ? 1 : 0;
      jy -= (-wy) & ph.ny;
      jz += wy;
      const int wz = jz == ph.nz ? 1 : 0;
      jz -= (-wz) & ph.nz;
      ci += (-wz) & BK;
      place();
      return;
    }
#pragma unroll
    for (int w = 0; w < WRAPS; ++w) {
      ci += BK / WRAPS;
      const int wc = ci >= Cin ? 1 : 0;
      ci -= (-wc) & Cin;
      jx += wc;
      const int wx = jx == ph.nx ? 1 : 0;
      jx -= (-wx) & ph.nx;
      jy += wx;
      const int wy = jy == ph.ny ? 1 : 0;
      jy -= (-wy) & ph.ny;
      jz += wy;
    }
    place();
  };

  // Two register stages: a tile is loaded during K-step kt (under the MFMAs of group 0),
  // written to LDS during K-step kt+1 (under group 3) and consumed in K-step kt+2, so a
  // global load has a whole K-step (~4000 cycles) to land before anything waits on it.
  struct Stage {
    float4 ra[4], rb[BROWS], rsc[PRO == 2 ? 4 : 1], rsh[PRO == 2 ? 4 : 1];
    unsigned amask;
    int kvalid;
  };
  Stage SX, SY;

  auto issue_loads = [&](Stage& S) {
    float4 (&ra)[4] = S.ra;
    float4 (&rb)[BROWS] = S.rb;
    auto& rsc = S.rsc;
    auto& rsh = S.rsh;
    unsigned amask = 0;
    S.kvalid = cmask;
    const int tci = UCUR ? ci + 4 * cc : ci;                 // this thread's channel
    const unsigned cisB = (unsigned)(tci & cmask) * 4u;      // channel 0 past the last tap: reads stay in range
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ok = cmask;                                                  // 0 / -1
      if constexpr (!FAST) {
        const int iz = rz[i] + dz, iy = ry[i] + dy, ix = rx[i] + dx;
        ok = ((unsigned)iz < (unsigned)Di ? cmask : 0) & ((unsigned)iy < (unsigned)Hi ? -1 : 0) &
             ((unsigned)ix < (unsigned)Wi ? -1 : 0);
      }
      const unsigned boff = (rbB[i] + deltaB + (unsigned)tci * 4u) & (unsigned)ok;
      ra[i] = *reinterpret_cast<const float4*>(ginb + boff);
      if constexpr (!FAST) amask |= ((unsigned)ok & 1u) << i;
      if constexpr (PRO == 2) {
        const unsigned sb = (unsigned)(rn[i] * nstride) * 4u + cisB;
        rsc[i] = *reinterpret_cast<const float4*>(gscb + sb);
        rsh[i] = *reinterpret_cast<const float4*>(gshb + sb);
      }
    }
    if constexpr (PRO == 1 || PRO == 3) {
      rsc[0] = *reinterpret_cast<const float4*>(gscb + cisB);
      rsh[0] = *reinterpret_cast<const float4*>(gshb + cisB);
    }
    const unsigned wkB = (unsigned)(woff + tci) * 4u;
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      rb[i] = *reinterpret_cast<const float4*>(gwb + ((wrowB[i] + wkB) & (unsigned)cmask));
    S.amask = amask;
    advance();
  };

  auto store_tile = [&](int buf, const Stage& S) {
    const float4 (&ra)[4] = S.ra;
    const float4 (&rb)[BROWS] = S.rb;
    const auto& rsc = S.rsc;
    const auto& rsh = S.rsh;
    const unsigned amask = S.amask;
    const int kvalid = S.kvalid;
    float* As = lds + buf * STAGE;
    float* Bs = As + (MM16 ? BM * (MM16_PITCHB / 4) : BM * PITCH);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if constexpr (PRO == 3) {           // LeakyReLU with a host-known slope in [0, 1]: max(y, slope*y), exact
        const float4 sc = rsc[0], sh = rsh[0];
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
        v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
      } else if constexpr (PRO != 0) {
        const float4 sc = rsc[PRO == 2 ? i : 0], sh = rsh[PRO == 2 ? i : 0];
        v.x = act_apply(v.x * sc.x + sh.x, act, slope);
        v.y = act_apply(v.y * sc.y + sh.y, act, slope);
        v.z = act_apply(v.z * sc.z + sh.z, act, slope);
        v.w = act_apply(v.w * sc.w + sh.w, act, slope);
      }
      if constexpr (!FAST) {
        const bool ok = (amask >> i) & 1u;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      }
      if constexpr (MM16) {
        mm_bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<mm_bf16x4*>(reinterpret_cast<char*>(As) + (r0 + 32 * i) * MM16_PITCHB + cc * 8) = o;
      } else {
        *reinterpret_cast<float4*>(As + (r0 + 32 * i) * PITCH + cc * 4) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      float4 v = rb[i];
      if constexpr (!FAST) {      // FAST: every channel row exists and tiles past the last K-step are never read
        const bool ok = (kvalid & (int)((bvalid >> i) & 1u)) != 0;   // kvalid is 0 / -1
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      }
      if constexpr (MM16) {
        mm_bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<mm_bf16x4*>(reinterpret_cast<char*>(Bs) + (r0 + 32 * i) * MM16_PITCHB + cc * 8) = o;
      } else {
        *reinterpret_cast<float4*>(Bs + (r0 + 32 * i) * PITCH + cc * 4) = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bias_pre[TN];       // the epilogue's bias values: in flight under the whole K loop
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int co = n0 + (wn * TN + tn) * 32 + li;
    bias_pre[tn] = (p.bias && co < Cout) ? p.bias[co] : 0.f;
  }

  if (nk > 0) {
    issue_loads(SY);        // tile 0
    issue_loads(SX);        // tile 1, issued beside it: the block's cold round trips overlap (the phase stamps put the
    store_tile(0, SY);      // prologue at 4-7 us of a 15-60 us launch); tile 1 stays in flight into the first K-step
  }
  __syncthreads();
  MPGAN_STAMP(p, 1);        // prologue done: first tile in LDS

  // One K-step: MFMAs on LDS buffer cb; under group 0 load tile kt+2 into `Sn`, under
  // group 3 write tile kt+1 (held by `Sp`) to the other buffer.  Loads past the last tile
  // hit one clamped address (cvalid = 0) and store zeros nobody reads.
  auto step = [&](int cb, Stage& Sn, const Stage& Sp) {
    if constexpr (MM16) {
      // lane (row li, half lh) holds K = 16 s + 8 lh .. + 7 of its row for k-sub s: 16 bytes at 32 s + 16 lh
      const char* Ab = reinterpret_cast<const char*>(lds + cb * STAGE) + (wm * TM * 32 + li) * MM16_PITCHB + 16 * lh;
      const char* Bb = reinterpret_cast<const char*>(lds + cb * STAGE) + (BM + wn * TN * 32 + li) * MM16_PITCHB + 16 * lh;
      mm_bf16x8 a[2][TM], b[2][TN];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[s][tm] = *reinterpret_cast<const mm_bf16x8*>(Ab + tm * 32 * MM16_PITCHB + 32 * s);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[s][tn] = *reinterpret_cast<const mm_bf16x8*>(Bb + tn * 32 * MM16_PITCHB + 32 * s);
      }
      issue_loads(Sn);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], b[0][tn], acc[tm][tn], 0, 0, 0);
      store_tile(cb ^ 1, Sp);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][tm], b[1][tn], acc[tm][tn], 0, 0, 0);
      return;
    }
    const float* As = lds + cb * STAGE + (wm * TM * 32 + li) * PITCH + 4 * lh;
    const float* Bs = lds + cb * STAGE + BM * PITCH + (wn * TN * 32 + li) * PITCH + 4 * lh;
    float4 a[2][TM], b[2][TN];
    auto read_group = [&](int g, int slot) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        a[slot][tm] = *reinterpret_cast<const float4*>(As + tm * 32 * PITCH + 8 * g);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        b[slot][tn] = *reinterpret_cast<const float4*>(Bs + tn * 32 * PITCH + 8 * g);
    };
    read_group(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sl = g & 1;
      if (g < 3) read_group(g + 1, sl ^ 1);
      if (g == 0) issue_loads(Sn);
      if (g == 3) store_tile(cb ^ 1, Sp);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].x, b[sl][tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].y, b[sl][tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].z, b[sl][tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].w, b[sl][tn].w, acc[tm][tn], 0, 0, 0);
        }
      // ---- issue order inside this group ----
      if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);       // next group's fragment reads
      if (g == 0) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x006, 13, 0);                    // address math (VALU|SALU)
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     // one global load
        }
      } else if (g == 3) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 11, 0);                    // normalise / mask
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                     // one LDS store
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (KS == 1) {
    for (int kt = 0; kt < nk; kt += 2) {
      step(0, SY, SX);
      __syncthreads();
      if (kt + 1 < nk) {
        step(1, SX, SY);
        __syncthreads();
      }
    }
    MPGAN_STAMP(p, 2);      // K loop done
    MPGAN_STAMP(p, 3);
    conv_epilogue<BN, TM, TN, WN>(p, ph, acc, lds, m0, n0, Mtot, stats_row, FAST && m0 + BM > Mtot, threadIdx.x, true, bias_pre);
    MPGAN_STAMP(p, 7);
  } else {
    // every group walks nk_per K-steps' worth of barriers; a group whose range is shorter idles at them
    for (int kt = 0; kt < nk_per; kt += 2) {
      if (kt < nk) step(0, SY, SX);
      __syncthreads();
      if (kt + 1 < nk_per) {
        if (kt + 1 < nk) step(1, SX, SY);
        __syncthreads();
      }
    }
    MPGAN_STAMP(p, 2);      // K loop done
    // fold the groups' sums into group 0 through LDS: [register][thread] floats, conflict-free
    float* red = lds_all;
    for (int g = 1; g < KS; ++g) {
      if (kg == g) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((a * TN + b) * 16 + r) * 256 + tid] = acc[a][b][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * TN + b) * 16 + r) * 256 + tid];
      }
      __syncthreads();
    }
    MPGAN_STAMP(p, 3);      // in-block split-K fold done
    conv_epilogue<BN, TM, TN, WN>(p, ph, acc, lds_all, m0, n0, Mtot, stats_row, FAST && m0 + BM > Mtot, tid, kg == 0, bias_pre);
    MPGAN_STAMP(p, 7);
  }
}

}  // namespace mpgan
