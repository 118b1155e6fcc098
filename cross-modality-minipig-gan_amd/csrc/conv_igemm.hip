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
#include "conv_geom.h"
#include "lds_dma.h"
#include "conv_pipe.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace mpgan {


// Per-thread cursor over the flattened K = (tap, ci) axis.  It is decoded with
// integer divisions ONCE; each K-step then advances it by 32 with compares only.
struct KCursor {
  int ci, jz, jy, jx;   // position
  int delta;            // pixel offset of the tap relative to the row's base pixel
  int dz, dy, dx;       // per-dimension input offsets of the tap
  int woff;             // tapflat * Cin
  bool valid;           // tap index still inside this phase's tap list
};

template <int BN, int TM, int TN, int WN, bool SCALAR>
__global__ __launch_bounds__(256) void gather_conv_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int STAGE = (BM + BN) * PITCH;
  constexpr int BROWS = BN / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const BlockId bid = conv_block_id(p);
  const Phase ph = p.ph[bid.phase];
  const long Mtot = (long)p.N * ph.Mz * ph.My * ph.Mx;
  const long m0 = (long)bid.mt * BM;
  const int n0 = bid.nt * BN;
  const int stats_row = bid.row;
  if (m0 >= Mtot) {
    if (p.stats && tid < BN && n0 + tid < p.Cout) {   // an empty tile of a short phase still owns a partial row
      float* row = p.stats + (long)stats_row * 2 * p.Cout;
      row[n0 + tid] = 0.f;
      row[p.Cout + n0 + tid] = 0.f;
    }
    if (p.bwd.part && tid < BN && n0 + tid < p.Cout) {
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
  const int nk = (Kp + BK - 1) / BK;
  const long Ktot = (long)p.Kz * p.Ky * p.Kx * Cin;
  const float* __restrict__ gin = p.in;
  const float* __restrict__ gw = p.wp;
  const float* __restrict__ gscale = p.pro.scale;
  const float* __restrict__ gshift = p.pro.shift;
  const int nstride = p.pro.n_stride, act = p.pro.act;
  const float slope = pro_slope(p.pro);

  // ---- per-thread load assignment: K-chunk column cc, rows r0 + 32*i ----
  const int cc = tid & 7, r0 = tid >> 3;
  int rn[4], rz[4], ry[4], rx[4], rbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned m = (unsigned)m0 + r0 + 32 * i;   // host guarantees Mtot < 2^31
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
    rbase[i] = ((rn[i] * Di + rz[i]) * Hi + ry[i]) * Wi + rx[i];
  }
  const int ksz = p.kstep[0], ksy = p.kstep[1], ksx = p.kstep[2];
  const int dsz = p.dstep[0], dsy = p.dstep[1], dsx = p.dstep[2];
  const int Ky = p.Ky, Kx = p.Kx;

  auto place = [&](KCursor& c) {
    c.valid = c.jz < ph.nz;
    const int kz = ph.kz0 + ksz * c.jz, ky = ph.ky0 + ksy * c.jy, kx = ph.kx0 + ksx * c.jx;
    c.woff = ((kz * Ky + ky) * Kx + kx) * Cin;
    c.dz = ph.dz0 + dsz * c.jz;
    c.dy = ph.dy0 + dsy * c.jy;
    c.dx = ph.dx0 + dsx * c.jx;
    c.delta = (c.dz * Hi + c.dy) * Wi + c.dx;
  };
  auto seek = [&](KCursor& c, int kidx) {   // divisions: once per thread
    int tap = kidx / Cin;
    c.ci = kidx - tap * Cin;
    c.jx = tap % ph.nx;
    int tq = tap / ph.nx;
    c.jy = tq % ph.ny;
    c.jz = tq / ph.ny;
    place(c);
  };
  auto advance = [&](KCursor& c, int by) {
    c.ci += by;
    if (c.ci >= Cin) {
      do {
        c.ci -= Cin;
        if (++c.jx == ph.nx) {
          c.jx = 0;
          if (++c.jy == ph.ny) { c.jy = 0; ++c.jz; }
        }
      } while (c.ci >= Cin);
      place(c);
    }
  };

  constexpr int NCUR = SCALAR ? 4 : 1;
  KCursor cur[NCUR];
  if (ntaps > 0) {
#pragma unroll
    for (int e = 0; e < NCUR; ++e) seek(cur[e], cc * 4 + e);
  } else {
#pragma unroll
    for (int e = 0; e < NCUR; ++e) { cur[e] = KCursor{}; cur[e].valid = false; }
  }

  float4 ra[4], rb[BROWS], rsc[4], rsh[4];
  unsigned amask = 0, bmask = 0;     // validity bits of the staged registers

  // Issue every global load of the next K-tile back to back, branch-free: invalid
  // (padding / tail) elements read a clamped in-range address and are zeroed when
  // the tile is written to LDS, after the MFMA phase has hidden the latency.
  auto global_load = [&]() {
    amask = 0; bmask = 0;
    if constexpr (!SCALAR) {
      const KCursor& c = cur[0];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iz = rz[i] + c.dz, iy = ry[i] + c.dy, ix = rx[i] + c.dx;
        const bool ok = c.valid && (unsigned)iz < (unsigned)Di && (unsigned)iy < (unsigned)Hi &&
                        (unsigned)ix < (unsigned)Wi;
        const long off = ok ? (long)(rbase[i] + c.delta) * ldi + c.ci : 0;
        ra[i] = *reinterpret_cast<const float4*>(gin + off);
        amask |= (ok ? 1u : 0u) << i;
        if (gscale && (nstride != 0 || i == 0)) {
          const int si = rn[i] * nstride + c.ci;
          rsc[i] = *reinterpret_cast<const float4*>(gscale + si);
          rsh[i] = *reinterpret_cast<const float4*>(gshift + si);
        }
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int co = n0 + r0 + 32 * i;
        const bool ok = c.valid && co < Cout;
        const long off = ok ? (long)co * Ktot + c.woff + c.ci : 0;
        rb[i] = *reinterpret_cast<const float4*>(gw + off);
        bmask |= (ok ? 1u : 0u) << i;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const KCursor& c = cur[e];
          const int iz = rz[i] + c.dz, iy = ry[i] + c.dy, ix = rx[i] + c.dx;
          const bool ok = c.valid && (unsigned)iz < (unsigned)Di && (unsigned)iy < (unsigned)Hi &&
                          (unsigned)ix < (unsigned)Wi;
          const long off = ok ? (long)(rbase[i] + c.delta) * ldi + c.ci : 0;
          float x = gin[off];
          if (gscale) {
            const int si = ok ? rn[i] * nstride + c.ci : 0;
            x = act_apply(x * gscale[si] + gshift[si], act, slope);
          }
          v[e] = ok ? x : 0.f;
        }
        ra[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
      amask = 0xF;
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int co = n0 + r0 + 32 * i;
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const KCursor& c = cur[e];
          const bool ok = c.valid && co < Cout;
          const float x = gw[ok ? (long)co * Ktot + c.woff + c.ci : 0];
          w[e] = ok ? x : 0.f;
        }
        rb[i] = make_float4(w[0], w[1], w[2], w[3]);
      }
      bmask = 0xFF;
    }
#pragma unroll
    for (int e = 0; e < NCUR; ++e) advance(cur[e], BK);
  };

  auto lds_store = [&](int buf) {
    float* As = lds + buf * STAGE;
    float* Bs = As + BM * PITCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if constexpr (!SCALAR) {
        if (gscale) {
          const float4 sc = nstride != 0 ? rsc[i] : rsc[0], sh = nstride != 0 ? rsh[i] : rsh[0];
          v.x = act_apply(v.x * sc.x + sh.x, act, slope);
          v.y = act_apply(v.y * sc.y + sh.y, act, slope);
          v.z = act_apply(v.z * sc.z + sh.z, act, slope);
          v.w = act_apply(v.w * sc.w + sh.w, act, slope);
        }
        if (!((amask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(As + (r0 + 32 * i) * PITCH + cc * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      float4 v = rb[i];
      if (!((bmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(Bs + (r0 + 32 * i) * PITCH + cc * 4) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (nk > 0) {
    global_load();
    lds_store(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cb = kt & 1;
    if (kt + 1 < nk) global_load();
    const float* As = lds + cb * STAGE + (wm * TM * 32 + li) * PITCH + 4 * lh;
    const float* Bs = lds + cb * STAGE + BM * PITCH + (wn * TN * 32 + li) * PITCH + 4 * lh;
    // Fragment reads run one K-group (8 K) ahead of the MFMAs that consume them;
    // sched_group_barrier pins "reads of group g+1, then MFMAs of group g" so the
    // LDS latency hides under 16 MFMAs instead of stalling every 8.
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
    __builtin_amdgcn_s_setprio(1);   // the MFMA stream must not lose issue slots to the partner wave's address math
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sl = g & 1;
      if (g < 3) read_group(g + 1, sl ^ 1);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].x, b[sl][tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].y, b[sl][tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].z, b[sl][tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][tm].w, b[sl][tn].w, acc[tm][tn], 0, 0, 0);
        }
      if (g < 3) {
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);      // DS reads of the next group first
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);  // then this group's MFMAs
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (kt + 1 < nk) lds_store(cb ^ 1);
    __syncthreads();
  }

  conv_epilogue<BN, TM, TN, WN>(p, ph, acc, lds, m0, n0, Mtot, stats_row);
}



// ---------------------------------------------------------------------------
// DMA-staged form of the K-stepped kernel (round 3), for gathers WITHOUT a normalise-on-load prologue: the
// discriminator's backward-data launches (15 ms of the C3 step).  The pipelined kernel above spends ~60 vector
// instructions per K-step and thread on its operands (address math, range masks, LDS stores) beside 64 MFMAs per
// wave, and two waves per SIMD share the vector issue port: dropping that work was worth +14 % in a what-if build
// (DESIGN.md 5.1).  Here nothing passes through registers: both operands go global -> LDS by LDS-DMA buffer loads
// (buffer_load_dwordx4 ... offen lds: a per-thread 32-bit voffset computed once, the tap / channel-chunk walk in the
// scalar soffset; a masked piece selects an out-of-range voffset and the bounds check delivers the zeros), exactly
// the staging of gather_conv_bf16_wide_kernel (conv_bf16.hip) with fp32 rows:
//   * LDS rows of 32 floats = one 128-byte line, unpadded (LDS-DMA writes lane-linear), 16-byte chunk c of row r at
//     c ^ ((r >> 1) & 7): the ds_read_b128 of fragment group g (chunk 2g + lh of row li) is conflict-free;
//   * two stages of (128 + BN) rows (32 KiB at BN = 128: two blocks per CU as before); tile kt+2 is issued into the
//     stage tile kt leaves, behind the barrier in front of kt's last fragment group, half there and half under the
//     next K-step's first group (an LDS-DMA instruction costs its wave 60-185 issue cycles);
//   * fragment reads as inline asm (hipcc otherwise drains the DMAs in front of every LDS read), ONE read site per
//     group position; MFMAs, accumulator layout and epilogue (conv_epilogue: bias, residual, fused norm-backward
//     sums, 16-byte stores) are the pipelined kernel's.
// Tile 128 x BN, 4 waves, K order channel-chunk major / taps inner, Cin % 32 == 0, at most 32 taps per phase.
// ---------------------------------------------------------------------------
//   NST: LDS stages; 2 everywhere.  For the 32-wide tile (the generator's strided / transposed 3-D layers without a
//   prologue: 2-16 K-steps per block, thousands of blocks) four stages of 20 KiB were tried on the theory that a K-step
//   there takes a load latency -- it LOST 8-19 % against the pipelined kernel (80 KiB of LDS: two blocks per CU
//   instead of three; those launches live on the number of blocks whose prologues and epilogues overlap), while two
//   stages (40 KiB, four blocks per CU) gain 8-14 % over it.
template <int BN, int TM, int TN, int WN, int NST = 2>
__global__ __launch_bounds__(256) void gather_conv_dma_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  char* ldsb = reinterpret_cast<char*>(lds_all);
  constexpr int ROWB = 128;
  constexpr bool SPLIT = NST == 2;                        // (two stages: a tile's pieces in two halves, see above)
  constexpr int STAGEB = (BM + BN) * ROWB;
  constexpr int PR = 32;                                  // rows one LDS-DMA instruction of every wave fills
  constexpr int AP = BM / PR, BP = BN / PR;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  MPGAN_STAMP(p, 0);
  MPGAN_STAMP_VALUE(p, 6, 1);
  const BlockId bid = conv_block_id(p);
  const Phase ph = p.ph[bid.phase];
  const long Mtot = (long)p.N * ph.Mz * ph.My * ph.Mx;
  const long m0 = (long)bid.mt * BM;
  const int n0 = bid.nt * BN;
  const int stats_row = bid.row;
  if (m0 >= Mtot) {
    if (p.stats && tid < BN && n0 + tid < p.Cout) {
      float* row = p.stats + (long)stats_row * 2 * p.Cout;
      row[n0 + tid] = 0.f;
      row[p.Cout + n0 + tid] = 0.f;
    }
    if (p.bwd.part && tid < BN && n0 + tid < p.Cout) {
      float* row = p.bwd.part + (long)stats_row * 3 * p.Cout;
      row[n0 + tid] = 0.f;
      row[p.Cout + n0 + tid] = 0.f;
      row[2 * p.Cout + n0 + tid] = 0.f;
    }
    return;
  }
  const int Cin = p.Cin, Cout = p.Cout, Di = p.Di, Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  const int ntaps = ph.nz * ph.ny * ph.nx;
  const int nk = ntaps * (Cin / BK);
  const unsigned Ktot4 = (unsigned)(p.Kz * p.Ky * p.Kx * Cin) * 4u;
  const unsigned bytesA = (unsigned)((((long)p.N * Di * Hi * Wi - 1) * ldi + Cin) * 4);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, (unsigned)Cout * Ktot4, 0x00020000);

  // this thread's pieces: rows r0 + 32 i, 16-byte chunk ck of the K-step (source-side swizzle)
  const int r0 = tid >> 3, cc = tid & 7;
  const int ck = cc ^ ((r0 >> 1) & 7);
  unsigned voffA[AP], tmask[AP], voffB[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    unsigned m = (unsigned)m0 + r0 + PR * i;
    const bool live = m < (unsigned)Mtot;
    m = live ? m : (unsigned)Mtot - 1u;
    unsigned q, umx, umy, umz;
    fdivmod(m, ph.fMx, q, umx);
    fdivmod(q, ph.fMy, q, umy);
    fdivmod(q, ph.fMz, q, umz);
    const int bz = (int)umz * p.istride[0], by = (int)umy * p.istride[1], bx = (int)umx * p.istride[2];
    voffA[i] = (unsigned)((((int)q * Di + bz) * Hi + by) * Wi + bx) * (unsigned)ldi * 4u + (unsigned)ck * 16u;
    unsigned mk = 0;
    int j = 0;
    for (int jz = 0; jz < ph.nz; ++jz) {
      const bool okz = (unsigned)(bz + ph.dz0 + p.dstep[0] * jz) < (unsigned)Di;
      for (int jy = 0; jy < ph.ny; ++jy) {
        const bool oky = okz && (unsigned)(by + ph.dy0 + p.dstep[1] * jy) < (unsigned)Hi;
        for (int jx = 0; jx < ph.nx; ++jx, ++j) {
          const bool ok = oky && (unsigned)(bx + ph.dx0 + p.dstep[2] * jx) < (unsigned)Wi;
          mk |= (ok ? 1u : 0u) << j;
        }
      }
    }
    tmask[i] = live ? mk : 0u;                           // rows past the last pixel stage zeros
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int co = n0 + r0 + PR * i;
    voffB[i] = co < Cout ? (unsigned)co * Ktot4 + (unsigned)ck * 16u : HW_OOB;   // rows past the last channel: zeros
  }
  // incremental, wave-uniform tap walk (scalar registers): see gather_conv_bf16_wide_kernel
  const int aX = p.dstep[2] * ldi * 4;
  const int aY = p.dstep[1] * Wi * ldi * 4 - (ph.nx - 1) * aX;
  const int aZ = p.dstep[0] * Hi * Wi * ldi * 4 - (ph.ny - 1) * p.dstep[1] * Wi * ldi * 4 - (ph.nx - 1) * aX;
  const int wX = p.kstep[2] * Cin * 4;
  const int wY = p.kstep[1] * p.Kx * Cin * 4 - (ph.nx - 1) * wX;
  const int wZ = p.kstep[0] * p.Ky * p.Kx * Cin * 4 - (ph.ny - 1) * p.kstep[1] * p.Kx * Cin * 4 - (ph.nx - 1) * wX;
  const int delta0 = ((ph.dz0 * Hi + ph.dy0) * Wi + ph.dx0) * ldi * 4;
  const int woff0 = ((ph.kz0 * p.Ky + ph.ky0) * p.Kx + ph.kx0) * Cin * 4;
  const int nx = ph.nx, ny = ph.ny;
  int itap = 0, jx = 0, jy = 0, deltaB = delta0, woffB = woff0, ciB = 0;
  const int wbase = __builtin_amdgcn_readfirstlane(8 * wid * ROWB);
  auto issue = [&](int stage, auto part) {               // part 0: the whole tile; 1 / 2: its first / second half
    constexpr int P = decltype(part)::value;
    constexpr int A0 = P == 2 ? AP / 2 : 0, A1 = P == 1 ? AP / 2 : AP;
    constexpr int B0 = P == 2 ? BP / 2 : 0, B1 = P == 1 ? BP / 2 : BP;
    char* As = ldsb + stage + wbase;
    char* Bs = As + BM * ROWB;
#pragma unroll
    for (int i = A0; i < A1; ++i) {
      const unsigned v = ((tmask[i] >> itap) & 1u) ? voffA[i] + (unsigned)deltaB : HW_OOB;
      BLDS16(rsA, As + PR * i * ROWB, v, ciB);
    }
#pragma unroll
    for (int i = B0; i < B1; ++i) BLDS16(rsB, Bs + PR * i * ROWB, voffB[i], woffB + ciB);
    if constexpr (P != 1) {                               // the cursor moves on behind a tile's last piece
      itap += 1;
      jx += 1;
      int incA = aX, incW = wX;
      if (jx == nx) {
        jx = 0;
        jy += 1;
        incA = aY; incW = wY;
        if (jy == ny) { jy = 0; incA = aZ; incW = wZ; }
      }
      deltaB += incA;
      woffB += incW;
      if (itap == ntaps) { itap = 0; jx = 0; jy = 0; deltaB = delta0; woffB = woff0; ciB += BK * 4; }
    }
  };
  constexpr std::integral_constant<int, 0> ALL{};
  constexpr std::integral_constant<int, 1> HALF0{};
  constexpr std::integral_constant<int, 2> HALF1{};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bias_pre[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int co = n0 + (wn * TN + tn) * 32 + li;
    bias_pre[tn] = (p.bias && co < Cout) ? p.bias[co] : 0.f;
  }

  const int sw = (li >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) foff[g] = ((2 * g + lh) ^ sw) * 16;
  const int arow = (wm * TM * 32 + li) * ROWB;
  const int brow = BM * ROWB + (wn * TN * 32 + li) * ROWB;
  const unsigned lds_base = lds_addr(ldsb);
  i32x4 fa[2][TM], fb[2][TN];
  auto read_group = [&](int stage, int g, int slot) {
    const unsigned As = lds_base + stage + arow + foff[g], Bs = lds_base + stage + brow + foff[g];
    lds_read_b128_n<TM, 32 * ROWB>(fa[slot], As);
    lds_read_b128_n<TN, 32 * ROWB>(fb[slot], Bs);
  };
  constexpr int NL = AP + BP;
  // wait until at most t tiles' worth of this wave's DMAs are outstanding (t < NST is block-uniform)
  auto wait_tiles = [&](int t) {
    if (t <= 0) asm volatile("s_waitcnt vmcnt(0) ; tail: the awaited tile is the only one in flight (two stages)" ::: "memory");   // (tools/check_isa.py)
    else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else if (t == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL) : "memory");
  };
  static_assert(NST >= 2 && NST <= 4 && 3 * NL < 64, "wait_tiles counts up to three tiles in flight behind the awaited one");
  if (nk > 0) {
    // tiles 0 .. NST-1 fill the stages (tile kt lives in stage kt % NST); tile kt+NST is issued when tile kt's stage is free
#pragma unroll
    for (int i = 0; i < NST; ++i)
      if (i < nk) issue(i * STAGEB, ALL);
    wait_tiles((nk < NST ? nk : NST) - 1);               // tile 0 landed
    asm volatile("s_barrier" ::: "memory");
    MPGAN_STAMP(p, 1);
    read_group(0, 0, 0);
    int cst = 0;
    bool pend = false;                                    // SPLIT: the second half of the tile issued behind the last barrier
    for (int kt = 0; kt < nk; ++kt) {
      const int nst = cst == (NST - 1) * STAGEB ? 0 : cst + STAGEB;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int sl = g & 1;
        lds_wait<TM, TN>(fa[sl], fb[sl]);
        if (g < 3) {
          read_group(cst, g + 1, sl ^ 1);
          if constexpr (SPLIT)
            if (g == 0 && pend) issue(nst, HALF1);        // (nst: the stage the previous K-step left)
        } else if (kt + 1 < nk) {
          // tile kt+1 must have landed; behind it in flight: tiles kt+2 .. min(kt+NST-1, nk-1)
          const int last = kt + NST - 1 < nk - 1 ? kt + NST - 1 : nk - 1;
          wait_tiles(last - (kt + 1));
          asm volatile("s_barrier" ::: "memory");
          pend = kt + NST < nk;                           // tile kt+NST takes the stage tile kt leaves
          if constexpr (SPLIT) {
            if (pend) issue(cst, HALF0);
          } else {
            if (pend) issue(cst, ALL);
          }
          read_group(nst, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const float4 a = __builtin_bit_cast(float4, fa[sl][tm]), b = __builtin_bit_cast(float4, fb[sl][tn]);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[tm][tn], 0, 0, 0);
          }
      }
      cst = nst;
    }
  }
  asm volatile("s_barrier" ::: "memory");                // every fragment read done: the LDS becomes the epilogue's
  MPGAN_STAMP(p, 2);
  MPGAN_STAMP(p, 3);
  conv_epilogue<BN, TM, TN, WN>(p, ph, acc, lds_all, m0, n0, Mtot, stats_row, false, tid, true, bias_pre);
}

// ---------------------------------------------------------------------------
// Thin layers (1 input channel, or 1 output channel): HBM-bound stencils.  An
// MFMA tile would be >= 97 % padding there, so these run on the vector ALUs with
// the (tiny) weight set staged in LDS.  Same GatherConv geometry / phases.
// ---------------------------------------------------------------------------
struct PixDecode {
  int n, bz, by, bx, opix;   // input base coords (m*istride) and output pixel (-1: none)
};

__device__ __forceinline__ PixDecode decode_pixel(const GatherConv& p, const Phase& ph, unsigned m) {
  PixDecode d;
  unsigned q, umx, umy, umz;
  fdivmod(m, ph.fMx, q, umx);
  fdivmod(q, ph.fMy, q, umy);
  fdivmod(q, ph.fMz, q, umz);
  const int mx = (int)umx, my = (int)umy, mz = (int)umz;
  d.n = (int)q;
  d.bz = mz * p.istride[0]; d.by = my * p.istride[1]; d.bx = mx * p.istride[2];
  const int oz = mz * p.ostride[0] + ph.oz, oy = my * p.ostride[1] + ph.oy, ox = mx * p.ostride[2] + ph.ox;
  d.opix = (oz < p.Do && oy < p.Ho && ox < p.Wo) ? ((d.n * p.Do + oz) * p.Ho + oy) * p.Wo + ox : -1;
  return d;
}

// Cin == 1: one thread = one output pixel x V consecutive output channels.
template <int V>
__global__ __launch_bounds__(256) void thin_cin1_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [T][CoutPad]
  const Phase ph = p.ph[blockIdx.z];
  const int T = p.Kz * p.Ky * p.Kx;
  const int CQ = (p.Cout + V - 1) / V, CP = CQ * V;
  for (int i = threadIdx.x; i < T * CP; i += 256) {
    const int t = i / CP, co = i - t * CP;
    wl[i] = co < p.Cout ? p.wp[(long)co * T + t] : 0.f;
  }
  __syncthreads();
  const unsigned Mtot = (unsigned)p.N * ph.Mz * ph.My * ph.Mx;
  const unsigned gi = blockIdx.x * 256u + threadIdx.x;
  const unsigned m = gi / (unsigned)CQ;
  if (m >= Mtot) return;
  const int co = (int)(gi - m * CQ) * V;
  const PixDecode d = decode_pixel(p, ph, m);
  if (d.opix < 0) return;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = (p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
  for (int jz = 0; jz < ph.nz; ++jz) {
    const int iz = d.bz + ph.dz0 + p.dstep[0] * jz, kz = ph.kz0 + p.kstep[0] * jz;
    if ((unsigned)iz >= (unsigned)p.Di) continue;
    for (int jy = 0; jy < ph.ny; ++jy) {
      const int iy = d.by + ph.dy0 + p.dstep[1] * jy, ky = ph.ky0 + p.kstep[1] * jy;
      if ((unsigned)iy >= (unsigned)p.Hi) continue;
      const long rowbase = (((long)d.n * p.Di + iz) * p.Hi + iy) * p.Wi;
      for (int jx = 0; jx < ph.nx; ++jx) {
        const int ix = d.bx + ph.dx0 + p.dstep[2] * jx, kx = ph.kx0 + p.kstep[2] * jx;
        if ((unsigned)ix >= (unsigned)p.Wi) continue;
        const float x = p.in[(rowbase + ix) * p.ldi];
        const float* w = wl + ((kz * p.Ky + ky) * p.Kx + kx) * CP + co;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = fmaf(x, w[e], acc[e]);
      }
    }
  }
  float* o = p.out + (long)d.opix * p.ldo + co;
  const float* r = p.resid ? p.resid + (long)d.opix * p.ldr + co : nullptr;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    if (co + e >= p.Cout) break;
    float v = acc[e];
    if (p.epi.scale) v = epi_act1(v, p.epi.scale[co + e], p.epi.shift[co + e], p.epi.slope[co + e]);
    if (r) v += r[e];
    if (p.tanh_out) v = tanhf(v);
    acc[e] = v;
  }
  if (V == 4 && co + 3 < p.Cout)
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  else
    for (int e = 0; e < V && co + e < p.Cout; ++e) o[e] = acc[e];
}

// Cin == 1, Cout = 4*CQ <= 64: one thread = one output pixel x ALL output channels.  The input
// sample of a tap is loaded once and meets the tap's whole weight row, read from LDS at a
// wave-uniform address (broadcast); a thread stores 16*CQ contiguous bytes, a wave a contiguous run.
// (A fully unrolled 27-tap instance was tried in round 4: hipcc hoists every tap's weight row out of LDS at once and
//  spills -- 96 -> 881 us at 128^3; the run-time tap loops stay.)
template <int CQ, bool OUT_BF16 = false>
__global__ __launch_bounds__(256) void thin_cin1_full_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [T][4*CQ]
  constexpr int CO = 4 * CQ;
  const Phase ph = p.ph[blockIdx.z];
  const int T = p.Kz * p.Ky * p.Kx;
  for (int i = threadIdx.x; i < T * CO; i += 256) {
    const int t = i / CO, co = i - t * CO;
    wl[i] = p.wp[(long)co * T + t];
  }
  __syncthreads();
  const unsigned Mtot = (unsigned)p.N * ph.Mz * ph.My * ph.Mx;
  const unsigned m = blockIdx.x * 256u + threadIdx.x;
  const PixDecode d = decode_pixel(p, ph, m < Mtot ? m : 0u);
  const bool valid = m < Mtot && d.opix >= 0;
  if (!valid && !p.stats && !p.stats_acc) return;   // with fused statistics every thread reaches the block reduction
  float4 acc[CQ];
#pragma unroll
  for (int q = 0; q < CQ; ++q)
    acc[q] = p.bias ? *reinterpret_cast<const float4*>(p.bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ gin = p.in;
  const int ldi = p.ldi;
  for (int jz = 0; jz < (valid ? ph.nz : 0); ++jz) {
    const int iz = d.bz + ph.dz0 + p.dstep[0] * jz, kz = ph.kz0 + p.kstep[0] * jz;
    if ((unsigned)iz >= (unsigned)p.Di) continue;
    for (int jy = 0; jy < ph.ny; ++jy) {
      const int iy = d.by + ph.dy0 + p.dstep[1] * jy, ky = ph.ky0 + p.kstep[1] * jy;
      if ((unsigned)iy >= (unsigned)p.Hi) continue;
      const int rowbase = ((d.n * p.Di + iz) * p.Hi + iy) * p.Wi;
      for (int jx = 0; jx < ph.nx; ++jx) {
        const int ix = d.bx + ph.dx0 + p.dstep[2] * jx, kx = ph.kx0 + p.kstep[2] * jx;
        if ((unsigned)ix >= (unsigned)p.Wi) continue;
        const float x = gin[(long)(rowbase + ix) * ldi];
        const float4* w = reinterpret_cast<const float4*>(wl + ((kz * p.Ky + ky) * p.Kx + kx) * CO);
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
          const float4 wv = w[q];
          acc[q].x = fmaf(x, wv.x, acc[q].x);
          acc[q].y = fmaf(x, wv.y, acc[q].y);
          acc[q].z = fmaf(x, wv.z, acc[q].z);
          acc[q].w = fmaf(x, wv.w, acc[q].w);
        }
      }
    }
  }
  if (valid) {
    if constexpr (OUT_BF16) {       // bf16 storage (config C5): the raw output rounded once; statistics below stay fp32
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      __bf16* o = reinterpret_cast<__bf16*>(p.out) + (long)d.opix * p.ldo;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        bf16x4 t;
        t[0] = (__bf16)acc[q].x; t[1] = (__bf16)acc[q].y; t[2] = (__bf16)acc[q].z; t[3] = (__bf16)acc[q].w;
        *reinterpret_cast<bf16x4*>(o + 4 * q) = t;
      }
    } else {
    float* o = p.out + (long)d.opix * p.ldo;
    const float* r = p.resid ? p.resid + (long)d.opix * p.ldr : nullptr;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      float4 v = acc[q];
      if (p.epi.scale) {                     // (wave-uniform addresses: scalar loads)
        const float4 sc = *reinterpret_cast<const float4*>(p.epi.scale + 4 * q), sh = *reinterpret_cast<const float4*>(p.epi.shift + 4 * q),
                     sl = *reinterpret_cast<const float4*>(p.epi.slope + 4 * q);
        v.x = epi_act1(v.x, sc.x, sh.x, sl.x); v.y = epi_act1(v.y, sc.y, sh.y, sl.y);
        v.z = epi_act1(v.z, sc.z, sh.z, sl.z); v.w = epi_act1(v.w, sc.w, sh.w, sl.w);
      }
      if (r) {
        const float4 rv = *reinterpret_cast<const float4*>(r + 4 * q);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (p.tanh_out) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
      *reinterpret_cast<float4*>(o + 4 * q) = v;
    }
    }
  }
  if (p.stats || p.stats_acc) {
    // Fused BatchNorm statistics (raw output incl. bias; the host forbids resid/tanh with them): the block's
    // 256 pixel rows go through LDS [256][CO+1], then 2*CO threads sum one column each in row order.
    float* sred = wl + ((T * CO + 3) & ~3);
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      float* dst = sred + threadIdx.x * (CO + 1) + 4 * q;
      dst[0] = valid ? acc[q].x : 0.f; dst[1] = valid ? acc[q].y : 0.f;
      dst[2] = valid ? acc[q].z : 0.f; dst[3] = valid ? acc[q].w : 0.f;
    }
    __syncthreads();
    // 2*CO (column, moment) pairs x RG row groups: each thread folds 256/RG rows, then the first 2*CO
    // threads fold the RG partials -- fixed order, 256/RG + RG serial steps instead of 256
    constexpr int RG = 256 / (2 * CO);
    const int pair = threadIdx.x % (2 * CO), g = threadIdx.x / (2 * CO);
    const int c = pair % CO, sq = pair / CO;
    float t = 0.f;
    for (int r2 = g; r2 < 256; r2 += RG) {
      const float v = sred[r2 * (CO + 1) + c];
      t += sq ? v * v : v;
    }
    __syncthreads();
    sred[g * (2 * CO) + pair] = t;
    __syncthreads();
    if (threadIdx.x < 2 * CO) {
      float tt = 0.f;
#pragma unroll
      for (int k = 0; k < RG; ++k) tt += sred[k * (2 * CO) + threadIdx.x];
      if (p.stats_acc)
        acc_add(p.stats_acc + (long)(blockIdx.x % (unsigned)p.acc_rep) * ACC_WORDS * CO, CO, sq ? 2 : 0, c, tt);
      else
        p.stats[((long)blockIdx.z * gridDim.x + blockIdx.x) * 2 * CO + sq * CO + c] = tt;
    }
  }
}

// Cout == 1: LANES = Cin/4 lanes share one output pixel (16-byte channel chunks,
// coalesced rows), then a shuffle reduction.
template <int LANES, bool IN_BF16 = false>
__global__ __launch_bounds__(256) void thin_cout1_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [T][Cin]
  const Phase ph = p.ph[blockIdx.z];
  const int T = p.Kz * p.Ky * p.Kx;
  for (int i = threadIdx.x; i < T * p.Cin; i += 256) wl[i] = p.wp[i];
  __syncthreads();
  const unsigned Mtot = (unsigned)p.N * ph.Mz * ph.My * ph.Mx;
  const unsigned gi = blockIdx.x * 256u + threadIdx.x;
  const unsigned m = gi / LANES;
  const int l = (int)(gi % LANES);
  const bool live = m < Mtot;
  PixDecode d = decode_pixel(p, ph, live ? m : 0u);
  float acc = 0.f;
  if (live && d.opix >= 0) {
    for (int jz = 0; jz < ph.nz; ++jz) {
      const int iz = d.bz + ph.dz0 + p.dstep[0] * jz, kz = ph.kz0 + p.kstep[0] * jz;
      if ((unsigned)iz >= (unsigned)p.Di) continue;
      for (int jy = 0; jy < ph.ny; ++jy) {
        const int iy = d.by + ph.dy0 + p.dstep[1] * jy, ky = ph.ky0 + p.kstep[1] * jy;
        if ((unsigned)iy >= (unsigned)p.Hi) continue;
        const long rowbase = (((long)d.n * p.Di + iz) * p.Hi + iy) * p.Wi;
        for (int jx = 0; jx < ph.nx; ++jx) {
          const int ix = d.bx + ph.dx0 + p.dstep[2] * jx, kx = ph.kx0 + p.kstep[2] * jx;
          if ((unsigned)ix >= (unsigned)p.Wi) continue;
          float4 v;
          if constexpr (IN_BF16) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.in) + (rowbase + ix) * p.ldi + 4 * l);
            v = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
          } else {
            v = *reinterpret_cast<const float4*>(p.in + (rowbase + ix) * p.ldi + 4 * l);
          }
          const float4 w = *reinterpret_cast<const float4*>(wl + ((kz * p.Ky + ky) * p.Kx + kx) * p.Cin + 4 * l);
          acc = fmaf(v.x, w.x, acc);
          acc = fmaf(v.y, w.y, acc);
          acc = fmaf(v.z, w.z, acc);
          acc = fmaf(v.w, w.w, acc);
        }
      }
    }
  }
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (live && d.opix >= 0 && l == 0) {
    float v = acc + (p.bias ? p.bias[0] : 0.f);
    if (p.epi.scale) v = epi_act1(v, p.epi.scale[0], p.epi.shift[0], p.epi.slope[0]);
    if (p.resid) v += p.resid[(long)d.opix * p.ldr];
    if (p.tanh_out) v = tanhf(v);
    p.out[(long)d.opix * p.ldo] = v;
  }
}

// ConvTranspose2d(Cin -> 1, k3 s2 p1, output_padding 1) forward -- the U-Net's last up-conv.
// LANES = Cin/4 lanes share one INPUT pixel (y, x): its 2x2 neighbourhood is loaded once
// (16-byte channel chunks, coalesced) and produces the 2x2 output quad (2y+a, 2x+b) = all
// four phases; the lane partials are folded with a transposing butterfly (4-5 shuffles).
template <int LANES>
__global__ __launch_bounds__(256) void convt_quad_cout1_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [9][Cin]
  const int Cin = p.Cin;
  for (int i = threadIdx.x; i < 9 * Cin; i += 256) wl[i] = p.wp[i];
  __syncthreads();
  const Phase& ph = p.ph[0];                      // phase (0,0): m-grid == input grid
  const unsigned Mtot = (unsigned)p.N * p.Hi * p.Wi;
  const unsigned gi = blockIdx.x * 256u + threadIdx.x;
  const unsigned m = gi / LANES;
  const int l = (int)(gi % LANES);
  const bool live = m < Mtot;
  unsigned q, ux, uy;
  fdivmod(live ? m : 0u, ph.fMx, q, ux);
  fdivmod(q, ph.fMy, q, uy);
  const int n = (int)q, y = (int)uy, x = (int)ux;
  const float* base = p.in + ((long)(n * p.Hi + y) * p.Wi + x) * p.ldi + 4 * l;
  const bool y1 = y + 1 < p.Hi, x1 = x + 1 < p.Wi;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v00 = live ? *reinterpret_cast<const float4*>(base) : z4;
  const float4 v01 = live && x1 ? *reinterpret_cast<const float4*>(base + p.ldi) : z4;
  const float4 v10 = live && y1 ? *reinterpret_cast<const float4*>(base + (long)p.Wi * p.ldi) : z4;
  const float4 v11 = live && y1 && x1 ? *reinterpret_cast<const float4*>(base + (long)(p.Wi + 1) * p.ldi) : z4;
  auto W = [&](int ky, int kx) { return *reinterpret_cast<const float4*>(wl + (ky * 3 + kx) * Cin + 4 * l); };
  auto dot = [](const float4& a, const float4& b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); return fmaf(a.w, b.w, acc);
  };
  // out[o] = sum_k in[(o + 1 - k) / 2] * w[k]:  even o <- (i=o/2, k=1);  odd o <- (i+1, k=0) + (i, k=2)
  float o00 = dot(v00, W(1, 1), 0.f);
  float o01 = dot(v01, W(1, 0), dot(v00, W(1, 2), 0.f));
  float o10 = dot(v10, W(0, 1), dot(v00, W(2, 1), 0.f));
  float o11 = dot(v11, W(0, 0), dot(v10, W(0, 2), dot(v01, W(2, 0), dot(v00, W(2, 2), 0.f))));
  // transposing butterfly: after it lane (l & 3) = 2*b + a holds output (2y+a, 2x+b) summed over all lanes
  const bool b0 = l & 1, b1 = l & 2;
  const float r0 = (b0 ? o10 : o00) + __shfl_xor(b0 ? o00 : o10, 1, 64);
  const float r1 = (b0 ? o11 : o01) + __shfl_xor(b0 ? o01 : o11, 1, 64);
  float t = (b1 ? r1 : r0) + __shfl_xor(b1 ? r0 : r1, 2, 64);
#pragma unroll
  for (int off = 4; off < LANES; off <<= 1) t += __shfl_xor(t, off, 64);
  float sv = 0.f;                               // raw output (with bias) of this lane, 0 if it owns none
  if (live && l < 4) {
    const int oy = 2 * y + (l & 1), ox = 2 * x + (l >> 1);
    if (oy < p.Ho && ox < p.Wo) {
      const long pix = ((long)n * p.Ho + oy) * p.Wo + ox;
      float v = t + (p.bias ? p.bias[0] : 0.f);
      sv = v;
      if (p.epi.scale) v = epi_act1(v, p.epi.scale[0], p.epi.shift[0], p.epi.slope[0]);
      if (p.resid) v += p.resid[pix * p.ldr];
      if (p.tanh_out) v = tanhf(v);
      p.out[pix * p.ldo] = v;
    }
  }
  if (p.stats || p.stats_acc) {                 // fused BatchNorm statistics of the single output channel
    __shared__ float ws1[4], ws2[4];
    const float s1 = wave_sum(sv), s2 = wave_sum(sv * sv);
    if ((threadIdx.x & 63) == 0) { ws1[threadIdx.x >> 6] = s1; ws2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t1 = ws1[0] + ws1[1] + ws1[2] + ws1[3], t2 = ws2[0] + ws2[1] + ws2[2] + ws2[3];
      if (p.stats_acc) {
        long long* rep = p.stats_acc + (long)(blockIdx.x % (unsigned)p.acc_rep) * ACC_WORDS;
        acc_add(rep, 1, 0, 0, t1);
        acc_add(rep, 1, 2, 0, t2);
      } else {
        p.stats[2 * (long)blockIdx.x] = t1;
        p.stats[2 * (long)blockIdx.x + 1] = t2;
      }
    }
  }
}

static bool convt_quad_ok(const GatherConv& p) {
  const int lanes = p.Cin / 4;
  if (!(p.Cout == 1 && !p.pro.scale && p.Cin % 4 == 0 && p.ldi % 4 == 0 && (lanes == 4 || lanes == 8 || lanes == 16) &&
        ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0)))
    return false;
  if (!(p.nphase == 4 && p.Di == 1 && p.Do == 1 && p.Kz == 1 && p.Ky == 3 && p.Kx == 3 && p.ostride[1] == 2 &&
        p.ostride[2] == 2 && p.istride[1] == 1 && p.istride[2] == 1 && p.dstep[1] == -1 && p.dstep[2] == -1))
    return false;
  const Phase& a = p.ph[0];
  const Phase& d = p.ph[3];
  // k3 s2 p1: phase (0,0) = tap k=1 at offset 0; phase (1,1) = taps k=0 (offset +1), k=2 (offset 0)
  return a.My == p.Hi && a.Mx == p.Wi && a.ny == 1 && a.nx == 1 && a.ky0 == 1 && a.kx0 == 1 && a.dy0 == 0 &&
         a.dx0 == 0 && d.ny == 2 && d.nx == 2 && d.ky0 == 0 && d.kx0 == 0 && d.dy0 == 1 && d.dx0 == 1 &&
         p.Ho <= 2 * p.Hi && p.Wo <= 2 * p.Wi;
}

// ConvTranspose3d(Cin -> 1, k3 s2 p1, output_padding 1) forward -- the 3-D U-Net's last up-conv (the quad kernel's
// 3-D sibling).  LANES = Cin/4 lanes share one INPUT voxel: its 2x2x2 neighbourhood is loaded once (16-byte channel
// chunks) and produces the 2x2x2 output octet (2z+a, 2y+b, 2x+c) = all eight phases (27 tap terms); the lane
// partials are folded with a transposing butterfly, after which lane (l & 7) = a + 2b + 4c owns one finished output.
// The phase-by-phase gather it replaces read every input voxel 27/8 times and wrote each output row in eight
// interleaved passes.
template <int LANES>
__global__ __launch_bounds__(256) void convt_oct_cout1_kernel(const GatherConv p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [27][Cin]
  const int Cin = p.Cin;
  for (int i = threadIdx.x; i < 27 * Cin; i += 256) wl[i] = p.wp[i];
  __syncthreads();
  const Phase& ph = p.ph[0];                      // phase (0,0,0): m-grid == input grid
  const unsigned Mtot = (unsigned)p.N * p.Di * p.Hi * p.Wi;
  const unsigned gi = blockIdx.x * 256u + threadIdx.x;
  const unsigned m = gi / LANES;
  const int l = (int)(gi % LANES);
  const bool live = m < Mtot;
  unsigned q, ux, uy, uz;
  fdivmod(live ? m : 0u, ph.fMx, q, ux);
  fdivmod(q, ph.fMy, q, uy);
  fdivmod(q, ph.fMz, q, uz);
  const int n = (int)q, z = (int)uz, y = (int)uy, x = (int)ux;
  const float* base = p.in + ((((long)n * p.Di + z) * p.Hi + y) * p.Wi + x) * p.ldi + 4 * l;
  const bool z1 = z + 1 < p.Di, y1 = y + 1 < p.Hi, x1 = x + 1 < p.Wi;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[2][2][2];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const bool ok = live && (dz == 0 || z1) && (dy == 0 || y1) && (dx == 0 || x1);
        v[dz][dy][dx] = ok ? *reinterpret_cast<const float4*>(base + (((long)dz * p.Hi + dy) * p.Wi + dx) * p.ldi) : z4;
      }
  auto W = [&](int kz, int ky, int kx) { return *reinterpret_cast<const float4*>(wl + ((kz * 3 + ky) * 3 + kx) * Cin + 4 * l); };
  auto dot = [](const float4& a, const float4& b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); return fmaf(a.w, b.w, acc);
  };
  // out[o] = sum_k in[(o + 1 - k) / 2] * w[k] per dimension:  even o <- (offset 0, k=1);  odd o <- (offset 1, k=0) + (offset 0, k=2)
  float o[8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float s = 0.f;
#pragma unroll
        for (int ta = 0; ta <= a; ++ta)
#pragma unroll
          for (int tb = 0; tb <= b; ++tb)
#pragma unroll
            for (int tc = 0; tc <= c; ++tc) {
              // parity 0: one term (offset 0, k 1); parity 1: term 0 = (offset 1, k 0), term 1 = (offset 0, k 2)
              const int dz = a ? 1 - ta : 0, kz = a ? 2 * ta : 1;
              const int dy = b ? 1 - tb : 0, ky = b ? 2 * tb : 1;
              const int dx = c ? 1 - tc : 0, kx = c ? 2 * tc : 1;
              s = dot(v[dz][dy][dx], W(kz, ky, kx), s);
            }
        o[a + 2 * b + 4 * c] = s;
      }
  // transposing butterfly over the lane bits 0..2: after it lane (l & 7) = i holds o[i] summed over those 8 lanes
  float r4[4], r2[2];
  const bool b0 = l & 1, b1 = l & 2, b2 = l & 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) r4[i] = (b0 ? o[2 * i + 1] : o[2 * i]) + __shfl_xor(b0 ? o[2 * i] : o[2 * i + 1], 1, 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) r2[i] = (b1 ? r4[2 * i + 1] : r4[2 * i]) + __shfl_xor(b1 ? r4[2 * i] : r4[2 * i + 1], 2, 64);
  if constexpr (LANES == 4) {
    // four lanes per voxel: the butterfly ends after two stages, lane l owns outputs (a, b) = (l & 1, l >> 1) for c = 0, 1
    if (live) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int oz = 2 * z + (l & 1), oy = 2 * y + ((l >> 1) & 1), ox = 2 * x + c;
        if (oz < p.Do && oy < p.Ho && ox < p.Wo) {
          const long pix = (((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
          float vv = r2[c] + (p.bias ? p.bias[0] : 0.f);
          if (p.epi.scale) vv = epi_act1(vv, p.epi.scale[0], p.epi.shift[0], p.epi.slope[0]);
          if (p.resid) vv += p.resid[pix * p.ldr];
          if (p.tanh_out) vv = tanhf(vv);
          p.out[pix * p.ldo] = vv;
        }
      }
    }
    return;
  }
  float t = (b2 ? r2[1] : r2[0]) + __shfl_xor(b2 ? r2[0] : r2[1], 4, 64);
#pragma unroll
  for (int off = 8; off < LANES; off <<= 1) t += __shfl_xor(t, off, 64);
  if (live && l < 8) {
    const int oz = 2 * z + (l & 1), oy = 2 * y + ((l >> 1) & 1), ox = 2 * x + (l >> 2);
    if (oz < p.Do && oy < p.Ho && ox < p.Wo) {
      const long pix = (((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
      float vv = t + (p.bias ? p.bias[0] : 0.f);
      if (p.epi.scale) vv = epi_act1(vv, p.epi.scale[0], p.epi.shift[0], p.epi.slope[0]);
      if (p.resid) vv += p.resid[pix * p.ldr];
      if (p.tanh_out) vv = tanhf(vv);
      p.out[pix * p.ldo] = vv;
    }
  }
}

static bool convt_oct_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_CONVT_OCT") != nullptr;
  const int lanes = p.Cin / 4;
  if (off || !(p.Cout == 1 && !p.pro.scale && !p.stats && !p.stats_acc && p.Cin % 4 == 0 && p.ldi % 4 == 0 &&
               (lanes == 4 || lanes == 8 || lanes == 16) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0)))
    return false;
  if (!(p.nphase == 8 && p.Kz == 3 && p.Ky == 3 && p.Kx == 3)) return false;
  for (int d = 0; d < 3; ++d)
    if (p.ostride[d] != 2 || p.istride[d] != 1 || p.dstep[d] != -1) return false;
  const Phase& a = p.ph[0];
  const Phase& h = p.ph[7];
  // k3 s2 p1: phase (0,0,0) = tap k=1 at offset 0; phase (1,1,1) = taps k=0 (offset +1), k=2 (offset 0) per dimension
  return a.Mz == p.Di && a.My == p.Hi && a.Mx == p.Wi && a.nz == 1 && a.ny == 1 && a.nx == 1 && a.kz0 == 1 && a.ky0 == 1 &&
         a.kx0 == 1 && a.dz0 == 0 && a.dy0 == 0 && a.dx0 == 0 && h.nz == 2 && h.ny == 2 && h.nx == 2 && h.kz0 == 0 &&
         h.ky0 == 0 && h.kx0 == 0 && h.dz0 == 1 && h.dy0 == 1 && h.dx0 == 1 && p.Do <= 2 * p.Di && p.Ho <= 2 * p.Hi &&
         p.Wo <= 2 * p.Wi;
}

// 1 -> 1 channel, 3x3x3 (or 1x3x3), stride 1, padding 1: the top-level ResidualUnit's conv of every U-Net and its
// backward-data (same footprint, taps reversed).  One thread = FOUR consecutive x outputs: each of the nine
// (z, y) input rows is read as one 6-wide window (an unaligned 16-byte + an 8-byte load) instead of 27 scalar
// loads per output with per-tap bounds tests; weights are wave-uniform (scalar loads).  Interior threads take the
// vector path, the two border threads of a row read element by element.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
// REVX: the x taps run backwards (backward-data: window position 2 - jx of tap jx; forward: jx) -- a template
// parameter, so that the window index of every product is a compile-time constant (as a runtime value it cost two
// selects per FMA: 216 beside the 108 FMAs of a thread).
template <bool REVX>
__global__ __launch_bounds__(256) void thin_c1c1_rows4_kernel(const GatherConv p) {
  const Phase& ph = p.ph[0];
  const int Mx = ph.Mx, groups = (Mx + 3) >> 2;
  const long total = (long)p.N * ph.Mz * ph.My * groups;
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= total) return;
  const int gx = (int)(gi % groups);
  long r = gi / groups;
  const int y = (int)(r % ph.My); r /= ph.My;
  const int z = (int)(r % ph.Mz);
  const int n = (int)(r / ph.Mz);
  const int x0 = 4 * gx;
  const bool interior = x0 >= 1 && x0 + 5 <= p.Wi;          // window x0-1 .. x0+4 inside the row
  float o[4];
  const float bv = p.bias ? p.bias[0] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = bv;
  for (int jz = 0; jz < ph.nz; ++jz) {
    const int iz = z + ph.dz0 + p.dstep[0] * jz, kz = ph.kz0 + p.kstep[0] * jz;
    if ((unsigned)iz >= (unsigned)p.Di) continue;
#pragma unroll
    for (int jy = 0; jy < 3; ++jy) {
      const int iy = y + ph.dy0 + p.dstep[1] * jy, ky = ph.ky0 + p.kstep[1] * jy;
      if ((unsigned)iy >= (unsigned)p.Hi) continue;
      const float* __restrict__ row = p.in + (((long)n * p.Di + iz) * p.Hi + iy) * p.Wi;
      float win[6];
      if (interior) {
        const f32x4u a = *reinterpret_cast<const f32x4u*>(row + x0 - 1);
        const f32x2u b = *reinterpret_cast<const f32x2u*>(row + x0 + 3);
        win[0] = a.x; win[1] = a.y; win[2] = a.z; win[3] = a.w; win[4] = b.x; win[5] = b.y;
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int xi = x0 - 1 + i;
          win[i] = (unsigned)xi < (unsigned)p.Wi ? row[xi] : 0.f;
        }
      }
      const float* __restrict__ wr = p.wp + (kz * 3 + ky) * 3;      // packed [1][tap][1]
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) {
        const int off = REVX ? 2 - jx : jx;                          // window position of output 0's sample (thin_c1c1_ok)
        const float w = wr[ph.kx0 + p.kstep[2] * jx];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(w, win[j + off], o[j]);
      }
    }
  }
  const long obase = (((long)n * p.Do + z) * p.Ho + y) * p.Wo + x0;
  if (x0 + 3 < Mx && (p.Wo & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.resid)) & 15) == 0) {
    float4 v = make_float4(o[0], o[1], o[2], o[3]);                // whole quad, 16-byte aligned rows: one load, one store
    if (p.resid) {
      const float4 rr = *reinterpret_cast<const float4*>(p.resid + obase);
      v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
    }
    if (p.tanh_out) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
    *reinterpret_cast<float4*>(p.out + obase) = v;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (x0 + j >= Mx) break;
    float v = o[j];
    if (p.resid) v += p.resid[obase + j];
    if (p.tanh_out) v = tanhf(v);
    p.out[obase + j] = v;
  }
}

static bool thin_c1c1_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_THIN_ROWS") != nullptr;
  if (off || !(p.Cin == 1 && p.Cout == 1 && p.ldi == 1 && p.ldo == 1 && (!p.resid || p.ldr == 1) && p.nphase == 1 &&
               !p.pro.scale && !p.stats && !p.stats_acc && !p.in_bf16 && !p.out_bf16 && !p.epi.scale && p.Ky == 3 && p.Kx == 3 &&
               (p.Kz == 3 || p.Kz == 1)))
    return false;
  const Phase& ph = p.ph[0];
  if (!(ph.ny == 3 && ph.nx == 3 && ph.nz == p.Kz && ph.oz == 0 && ph.oy == 0 && ph.ox == 0 && p.Do == ph.Mz &&
        p.Ho == ph.My && p.Wo == ph.Mx && p.Wi == ph.Mx && p.Hi == ph.My && p.Di == ph.Mz))
    return false;
  for (int d = 0; d < 3; ++d) {
    if (p.istride[d] != 1 || p.ostride[d] != 1) return false;
    if (d == 0 && p.Kz == 1) continue;
    const int d0 = d == 0 ? ph.dz0 : (d == 1 ? ph.dy0 : ph.dx0);
    if (!((p.dstep[d] == 1 && d0 == -1) || (p.dstep[d] == -1 && d0 == 1))) return false;   // offsets -1, 0, +1
  }
  return true;
}

static bool thin_cin1_ok(const GatherConv& p) {
  const int T = p.Kz * p.Ky * p.Kx;
  return p.Cin == 1 && !p.pro.scale && (long)T * ((p.Cout + 3) / 4 * 4) * 4 <= 48 * 1024;
}

static bool thin_cout1_ok(const GatherConv& p) {
  const int T = p.Kz * p.Ky * p.Kx;
  const int lanes = p.Cin / 4;
  return p.Cout == 1 && !p.pro.scale && p.Cin % 4 == 0 && p.ldi % 4 == 0 && lanes >= 1 && lanes <= 64 &&
         (lanes & (lanes - 1)) == 0 && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0) &&
         (long)T * p.Cin * 4 <= 48 * 1024;
}

// Cin == 1 -> 64 channels, stride 1, no padding, 3x3 / 3x3x3 (D.conv1), row-walking form: lane = output channel,
// one wave walks whole output rows four pixels at a time.  The tap weights of the lane's channel sit in registers;
// the input samples of the four pixels (a 6-wide window per (kz, ky)) are wave-uniform and arrive through the
// scalar cache, so an FMA takes them as its scalar operand -- no per-lane addresses, bounds tests or LDS reads --
// and a pixel's 64 channels leave as ONE contiguous 128 / 256-byte store.  The last group of a row is re-anchored
// at Mx - 4 (re-stores the same values; statistics count each pixel once).  Grid and statistics rows as
// thin_cin1_full_kernel's: one [sum | sum^2] row per block over whatever pixels its four waves walked.
template <int T, bool OUT_BF16>
__global__ __launch_bounds__(256) void thin_cin1_rows_kernel(const GatherConv p) {
  __shared__ float red[4][128];
  constexpr int KZ = T == 27 ? 3 : 1;
  const Phase& ph = p.ph[0];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int rows = p.N * ph.Mz * ph.My, Mx = ph.Mx;
  const int nwaves = (int)gridDim.x * 4, ngroups = (Mx + 3) >> 2;
  float w[T];
#pragma unroll
  for (int t = 0; t < T; ++t) w[t] = p.wp[(long)lane * T + t];               // packed [Cout][tap][Cin = 1]
  const float bv = p.bias ? p.bias[lane] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int r = (int)blockIdx.x * 4 + wave; r < rows; r += nwaves) {
    const int my = r % ph.My, q = r / ph.My;
    const int mz = q % ph.Mz, n = q / ph.Mz;
    const float* __restrict__ g0 = p.in + (((long)n * p.Di + mz) * p.Hi + my) * p.Wi;
    const long orow = (long)r * Mx;
    for (int gi = 0; gi < ngroups; ++gi) {
      const int x0 = gi + 1 < ngroups ? 4 * gi : Mx - 4;
      const int first = 4 * gi - x0;
      float o[4] = {bv, bv, bv, bv};
#pragma unroll
      for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* __restrict__ src = g0 + ((long)kz * p.Hi + ky) * p.Wi + x0;  // wave-uniform: scalar loads
          float win[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) win[i] = src[i];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(w[(kz * 3 + ky) * 3 + kx], win[j + kx], o[j]);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long e = (orow + x0 + j) * p.ldo + lane;
        if constexpr (OUT_BF16) reinterpret_cast<__bf16*>(p.out)[e] = (__bf16)o[j];
        else p.out[e] = o[j];
        const float v = j >= first ? o[j] : 0.f;
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
    }
  }
  if (p.stats) {
    red[wave][lane] = s1;
    red[wave][64 + lane] = s2;
    __syncthreads();
    if (threadIdx.x < 128)
      p.stats[(long)blockIdx.x * 128 + threadIdx.x] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

static bool thin_cin1_rows_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_THIN_ROWS") != nullptr;
  const int T = p.Kz * p.Ky * p.Kx;
  const Phase& ph = p.ph[0];
  return !off && p.Cin == 1 && p.Cout == 64 && p.ldi == 1 && p.nphase == 1 && !p.pro.scale && !p.resid && !p.tanh_out &&
         !p.stats_acc && !p.in_bf16 && !p.epi.scale && (T == 9 || T == 27) && p.Kx == 3 && p.Ky == 3 && p.Kz == (T == 27 ? 3 : 1) &&
         ph.nx == 3 && ph.ny == 3 && ph.nz == p.Kz && ph.dx0 == 0 && ph.dy0 == 0 && ph.dz0 == 0 && ph.kx0 == 0 &&
         ph.ky0 == 0 && ph.kz0 == 0 && p.dstep[0] == 1 && p.dstep[1] == 1 && p.dstep[2] == 1 && p.kstep[0] == 1 &&
         p.kstep[1] == 1 && p.kstep[2] == 1 && p.istride[0] == 1 && p.istride[1] == 1 && p.istride[2] == 1 &&
         p.ostride[0] == 1 && p.ostride[1] == 1 && p.ostride[2] == 1 && ph.oz == 0 && ph.oy == 0 && ph.ox == 0 &&
         ph.Mx >= 4 && p.Wi == ph.Mx + 2 && p.Hi == ph.My + 2 && p.Di == ph.Mz + (T == 27 ? 2 : 0) && p.Wo == ph.Mx &&
         p.Ho == ph.My && p.Do == ph.Mz;
}

static int launch_thin(const GatherConv& p, long maxM, hipStream_t st) {
  const int T = p.Kz * p.Ky * p.Kx;
  if (p.out_bf16) {     // 1 -> C conv writing bf16 (D.conv1 of the bf16 path): all-channel kernel only
    MPGAN_UNSUPPORTED(!(thin_cin1_ok(p) && (p.Cout == 16 || p.Cout == 32 || p.Cout == 64) && p.ldo % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(p.out) & 7) == 0 && !p.resid && !p.tanh_out && !p.in_bf16 &&
                        (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)),
                      "thin conv (bf16 out): needs Cin == 1, Cout in {16, 32, 64}, no resid/tanh, aligned output");
    MPGAN_UNSUPPORTED(p.stats && p.Cout > 64, "thin conv (bf16 out): fused statistics up to 64 channels");
    dim3 grid((unsigned)((maxM + 255) / 256), 1, (unsigned)p.nphase);
    if (thin_cin1_rows_ok(p)) {
      if (T == 9) hipLaunchKernelGGL((thin_cin1_rows_kernel<9, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((thin_cin1_rows_kernel<27, true>), grid, dim3(256), 0, st, p);
      return check_launch("thin_cin1_rows_bf16");
    }
    const size_t smem = ((size_t)((T * p.Cout + 3) & ~3) + (p.stats ? 256 * (size_t)(p.Cout + 1) : 0)) * sizeof(float);
    if (p.Cout == 16) hipLaunchKernelGGL((thin_cin1_full_kernel<4, true>), grid, dim3(256), smem, st, p);
    else if (p.Cout == 32) hipLaunchKernelGGL((thin_cin1_full_kernel<8, true>), grid, dim3(256), smem, st, p);
    else {
      static bool attr_set = false;
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(thin_cin1_full_kernel<16, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) { set_error("thin_cin1_full: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MPGAN_ERR_HIP; }
        attr_set = true;
      }
      hipLaunchKernelGGL((thin_cin1_full_kernel<16, true>), grid, dim3(256), smem, st, p);
    }
    return check_launch("thin_cin1_full_bf16");
  }
  if (p.in_bf16) {      // C -> 1 gather over bf16 data (backward-data of D.conv1 in the bf16 path)
    if (thin_cout1_mfma_bf16_ok(p)) return launch_thin_cout1_mfma_bf16(p, st);
    const int lanes_b = p.Cin / 4;
    MPGAN_UNSUPPORTED(!(p.Cout == 1 && !p.pro.scale && p.Cin % 4 == 0 && p.ldi % 4 == 0 &&
                        (lanes_b == 4 || lanes_b == 8 || lanes_b == 16) && (reinterpret_cast<uintptr_t>(p.in) & 7) == 0 &&
                        (long)T * p.Cin * 4 <= 48 * 1024 && !p.stats),
                      "thin conv (bf16 in): needs Cout == 1 and 16, 32 or 64 gathered channels");
    const long threads_b = maxM * lanes_b;
    dim3 grid((unsigned)((threads_b + 255) / 256), 1, (unsigned)p.nphase);
    const size_t smem = (size_t)T * p.Cin * sizeof(float);
    if (lanes_b == 4) hipLaunchKernelGGL((thin_cout1_kernel<4, true>), grid, dim3(256), smem, st, p);
    else if (lanes_b == 8) hipLaunchKernelGGL((thin_cout1_kernel<8, true>), grid, dim3(256), smem, st, p);
    else hipLaunchKernelGGL((thin_cout1_kernel<16, true>), grid, dim3(256), smem, st, p);
    return check_launch("thin_cout1_bf16");
  }
  if (thin_c1c1_ok(p)) {
    const Phase& ph0 = p.ph[0];
    const long threads4 = (long)p.N * ph0.Mz * ph0.My * ((ph0.Mx + 3) / 4);
    if (p.dstep[2] < 0) hipLaunchKernelGGL(thin_c1c1_rows4_kernel<true>, dim3((unsigned)((threads4 + 255) / 256)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(thin_c1c1_rows4_kernel<false>, dim3((unsigned)((threads4 + 255) / 256)), dim3(256), 0, st, p);
    return check_launch("thin_c1c1_rows4");
  }
  if (thin_cin1_ok(p)) {
    const bool v4 = (p.Cout % 4 == 0) && (p.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
    const bool full = v4 && (p.Cout == 16 || p.Cout == 32 || p.Cout == 64) &&
                      (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) &&
                      (!p.resid || ((p.ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0));
    MPGAN_UNSUPPORTED((p.stats || p.stats_acc) && !(full && p.Cout <= 32),
                      "thin conv: fused statistics need the all-channel kernel (Cout 16 or 32, 16-byte aligned output/bias)");
    if (full) {
      dim3 grid((unsigned)((maxM + 255) / 256), 1, (unsigned)p.nphase);
      if (!p.stats && thin_cin1_rows_ok(p)) {
        if (T == 9) hipLaunchKernelGGL((thin_cin1_rows_kernel<9, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((thin_cin1_rows_kernel<27, false>), grid, dim3(256), 0, st, p);
        return check_launch("thin_cin1_rows");
      }
      const size_t smem = ((size_t)((T * p.Cout + 3) & ~3) + ((p.stats || p.stats_acc) ? 256 * (size_t)(p.Cout + 1) : 0)) * sizeof(float);
      if (p.Cout == 16) hipLaunchKernelGGL(thin_cin1_full_kernel<4>, grid, dim3(256), smem, st, p);
      else if (p.Cout == 32) hipLaunchKernelGGL(thin_cin1_full_kernel<8>, grid, dim3(256), smem, st, p);
      else hipLaunchKernelGGL(thin_cin1_full_kernel<16>, grid, dim3(256), smem, st, p);
      return check_launch("thin_cin1_full");
    }
    const int V = v4 ? 4 : 1;
    const int CQ = (p.Cout + V - 1) / V;
    const long threads = maxM * CQ;
    dim3 grid((unsigned)((threads + 255) / 256), 1, (unsigned)p.nphase);
    const size_t smem = (size_t)T * CQ * V * sizeof(float);
    if (v4) hipLaunchKernelGGL(thin_cin1_kernel<4>, grid, dim3(256), smem, st, p);
    else hipLaunchKernelGGL(thin_cin1_kernel<1>, grid, dim3(256), smem, st, p);
    return check_launch("thin_cin1");
  }
  const int lanes = p.Cin / 4;
  MPGAN_UNSUPPORTED((p.stats || p.stats_acc) && !convt_quad_ok(p),
                    "thin conv: fused statistics of a 1-channel output exist for ConvTranspose2d(C -> 1, k3 s2) only");
  if (convt_quad_ok(p)) {
    const long qthreads = (long)p.N * p.Hi * p.Wi * lanes;
    dim3 qgrid((unsigned)((qthreads + 255) / 256));
    const size_t qsmem = (size_t)9 * p.Cin * sizeof(float);
    if (lanes == 4) hipLaunchKernelGGL(convt_quad_cout1_kernel<4>, qgrid, dim3(256), qsmem, st, p);
    else if (lanes == 8) hipLaunchKernelGGL(convt_quad_cout1_kernel<8>, qgrid, dim3(256), qsmem, st, p);
    else hipLaunchKernelGGL(convt_quad_cout1_kernel<16>, qgrid, dim3(256), qsmem, st, p);
    return check_launch("convt_quad_cout1");
  }
  if (convt_oct_ok(p)) {
    const long othreads = (long)p.N * p.Di * p.Hi * p.Wi * lanes;
    dim3 ogrid((unsigned)((othreads + 255) / 256));
    const size_t osmem = (size_t)27 * p.Cin * sizeof(float);
    if (lanes == 4) hipLaunchKernelGGL(convt_oct_cout1_kernel<4>, ogrid, dim3(256), osmem, st, p);
    else if (lanes == 8) hipLaunchKernelGGL(convt_oct_cout1_kernel<8>, ogrid, dim3(256), osmem, st, p);
    else hipLaunchKernelGGL(convt_oct_cout1_kernel<16>, ogrid, dim3(256), osmem, st, p);
    return check_launch("convt_oct_cout1");
  }
  const long threads = maxM * lanes;
  dim3 grid((unsigned)((threads + 255) / 256), 1, (unsigned)p.nphase);
  const size_t smem = (size_t)T * p.Cin * sizeof(float);
  switch (lanes) {
    case 1: hipLaunchKernelGGL(thin_cout1_kernel<1>, grid, dim3(256), smem, st, p); break;
    case 2: hipLaunchKernelGGL(thin_cout1_kernel<2>, grid, dim3(256), smem, st, p); break;
    case 4: hipLaunchKernelGGL(thin_cout1_kernel<4>, grid, dim3(256), smem, st, p); break;
    case 8: hipLaunchKernelGGL(thin_cout1_kernel<8>, grid, dim3(256), smem, st, p); break;
    case 16: hipLaunchKernelGGL(thin_cout1_kernel<16>, grid, dim3(256), smem, st, p); break;
    case 32: hipLaunchKernelGGL(thin_cout1_kernel<32>, grid, dim3(256), smem, st, p); break;
    default: hipLaunchKernelGGL(thin_cout1_kernel<64>, grid, dim3(256), smem, st, p); break;
  }
  return check_launch("thin_cout1");
}

// ---------------------------------------------------------------------------
// Patch kernel: 2-D layers with <= 32 output channels and 16/32/64 gathered channels
// (the U-Net's stride-1 units, its stride-2 down convs up to 32 channels, the transposed
// convs and every backward-data gather that lands on <= 32 channels).  Their K is short
// (<= 9 taps x 16..64 channels), so the K-stepped kernel above spends its time on the
// prologue, per-step barriers and re-fetching each input pixel once per tap.  Here a
// block stages the INPUT PATCH of an 8 x 16 tile of m-positions (halo included,
// normalise + activation applied once per element) and all weights of the phase in
// LDS with one round of global loads, then runs the whole contraction from LDS:
// one barrier, every input element fetched from HBM/L2 once per tile.
// Rows: wave w owns tile rows 2w, 2w+1; lane operand maps as in the kernels above.
// ---------------------------------------------------------------------------
constexpr int PT_H = 8, PT_W = 16;
__device__ float g_patch_zero_page[64];   // what the persistent patch kernel's padding chunks load

struct PatchLaunch {
  int tiles_x, tiles_y;
  int patch_floats;      // LDS floats reserved for the input patch (max over phases)
  FastDiv fCout;         // the staging loops are VALU-bound: no software divides in them
  FastDiv fPW[8], fNx[8];
  // merged form (one block per tile, all phases): union of the phases' tap offsets, LDS offset of
  // the statistics scratch (the patch stays live across phases)
  int merged, ylo, yhi, xlo, xhi, stats_off;
  FastDiv fPWm, fKx;
  int w_floats;          // persistent form: LDS floats of the staged weights (the patch follows them)
  int pca, pcw;          // persistent form: LDS pitch (floats) of a patch pixel / of a weight row
  const void* zero_page; // persistent form: >= 16 bytes of zeros in global memory (padding reads it)
  FastDiv fTx, fTy;      // persistent form: tile index -> (sample, tile row, tile column)
  int stagger;           // persistent form: sleep (x 64 clocks) per dispatch round before a block starts (see the kernel)
  int tw_off;            // persistent form: LDS float offset of the output transpose staging (4 waves x 16 px x pitch), or -1:
                         //   the epilogue then leaves through it as 16-byte stores (vector output path)
  int dbg;               // MPGAN_DBG_PATCH_SKIP bits (what-if timing builds): 1 no output stores, 2 no MFMAs, 4 no patch loads, 8 no LDS patch stores
};

//   MERGE : one block walks ALL phases of its tile (strided backward-data / transposed convs): the
//           phases read the same input pixels, so the patch (union of their tap offsets) and the
//           whole kernel's weights are staged once instead of once per phase -- a quarter of the
//           blocks, loads and round trips of the phase-per-block form.
template <int CIN, int PRO, bool NARROW, bool MERGE>
__global__ __launch_bounds__(256) void gather_patch_kernel(const GatherConv p, const PatchLaunch pl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int PC = CIN + 4;
  constexpr int CQ = CIN / 4;
  constexpr int LU = CIN == 16 ? 3 : (CIN == 32 ? 6 : 12);   // patch chunks per thread in flight
  constexpr int WU = CIN == 16 ? 3 : (CIN == 32 ? 9 : 8);    // weight chunks per thread in flight
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const unsigned wblk = xcd_remap(blockIdx.x, gridDim.x);
  // !MERGE: phase fastest -- the phases of one tile read the same input patch and run back to back on one XCD
  const int phase0 = MERGE ? 0 : (int)(wblk % (unsigned)p.nphase);
  const int nph = MERGE ? p.nphase : 1;
  unsigned q = MERGE ? wblk : wblk / (unsigned)p.nphase;
  const int tx = (int)(q % (unsigned)pl.tiles_x);
  q /= (unsigned)pl.tiles_x;
  const int ty = (int)(q % (unsigned)pl.tiles_y);
  const int n = (int)(q / (unsigned)pl.tiles_y);
  const int Cout = p.Cout;
  const int my0 = ty * PT_H, mx0 = tx * PT_W;
  if constexpr (!MERGE) {
    const Phase& ph = p.ph[phase0];
    const int My = ph.Mz > 0 ? ph.My : 0, Mx = ph.Mx;
    if (my0 >= My || mx0 >= Mx) {
      if (p.stats && tid < Cout) {
        float* row = p.stats + (long)wblk * 2 * Cout;
        row[tid] = 0.f;
        row[Cout + tid] = 0.f;
      }
      return;
    }
  }
  const int Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  const int isy = p.istride[1], isx = p.istride[2];
  const int dsy = p.dstep[1], dsx = p.dstep[2];
  int ylo, yhi, xlo, xhi, sntaps;             // staged patch: tap-offset range; staged weights: tap count
  if constexpr (MERGE) {
    ylo = pl.ylo; yhi = pl.yhi; xlo = pl.xlo; xhi = pl.xhi;
    sntaps = p.Ky * p.Kx;                     // every tap of the kernel, indexed by its flat position
  } else {
    const Phase& ph = p.ph[phase0];
    const int ny = ph.nz > 0 ? ph.ny : 0, nx = ph.nx;
    const int ye = ph.dy0 + dsy * (ny > 0 ? ny - 1 : 0), xe = ph.dx0 + dsx * (nx > 0 ? nx - 1 : 0);
    ylo = ph.dy0 < ye ? ph.dy0 : ye; yhi = ph.dy0 < ye ? ye : ph.dy0;
    xlo = ph.dx0 < xe ? ph.dx0 : xe; xhi = ph.dx0 < xe ? xe : ph.dx0;
    sntaps = ny * nx;
  }
  const int PH = (PT_H - 1) * isy + (yhi - ylo) + 1, PW = (PT_W - 1) * isx + (xhi - xlo) + 1;
  const int y0 = my0 * isy + ylo, x0 = mx0 * isx + xlo;
  float* patch = lds;
  float* wl = lds + pl.patch_floats;
  const float* __restrict__ gin = p.in + (long)n * Hi * Wi * ldi;
  const float* __restrict__ gw = p.wp;

  // ---- stage patch + weights: every global load of the first round is issued before any
  //      LDS store, so a block pays ONE memory round trip ----
  {
    const int cq = tid & (CQ - 1);                   // 256 % CQ == 0: a thread keeps its channel chunk
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float slope = 1.f;
    int act = 0;
    if constexpr (PRO != 0) {             // per channel, or per (sample, channel): a block serves one sample
      sc = *reinterpret_cast<const float4*>(p.pro.scale + (long)n * p.pro.n_stride + 4 * cq);
      sh = *reinterpret_cast<const float4*>(p.pro.shift + (long)n * p.pro.n_stride + 4 * cq);
      slope = pro_slope(p.pro);
      act = p.pro.act;
    }
    const int total = PH * PW * CQ;
    const FastDiv fPW = MERGE ? pl.fPWm : pl.fPW[phase0], fNx = MERGE ? pl.fKx : pl.fNx[phase0], fCout = pl.fCout;
    const Phase& ph0 = p.ph[phase0];
    const int wtotal = sntaps * Cout * CQ;
    const int Ktot = p.Kz * p.Ky * p.Kx * CIN;
    float4 pv[LU], wv[WU];
    unsigned pok;

    auto load_patch = [&](int base) {
      pok = 0;
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int idx = base + u * 256 + tid;
        const unsigned pix = (unsigned)idx / (unsigned)CQ;
        unsigned py, px;
        fdivmod(pix, fPW, py, px);
        const int iy = y0 + (int)py, ix = x0 + (int)px;
        const unsigned ok = (idx < total ? 1u : 0u) & ((unsigned)iy < (unsigned)Hi ? 1u : 0u) &
                            ((unsigned)ix < (unsigned)Wi ? 1u : 0u);
        const long off = ok ? ((long)iy * Wi + ix) * ldi + 4 * cq : 0;
        pv[u] = *reinterpret_cast<const float4*>(gin + off);
        pok |= ok << u;
      }
    };
    auto store_patch = [&](int base) {
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int idx = base + u * 256 + tid;
        float4 x = pv[u];
        if constexpr (PRO != 0) {
          x.x = act_apply(x.x * sc.x + sh.x, act, slope);
          x.y = act_apply(x.y * sc.y + sh.y, act, slope);
          x.z = act_apply(x.z * sc.z + sh.z, act, slope);
          x.w = act_apply(x.w * sc.w + sh.w, act, slope);
        }
        const bool ok = (pok >> u) & 1u;
        x.x = ok ? x.x : 0.f; x.y = ok ? x.y : 0.f; x.z = ok ? x.z : 0.f; x.w = ok ? x.w : 0.f;
        if (idx < total) *reinterpret_cast<float4*>(patch + (idx / CQ) * PC + 4 * cq) = x;
      }
    };
    // weights of this phase: LDS rows [tap][co] at pitch PC
    auto load_w = [&](int base) {
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        const int idx = base + u * 256 + tid;
        const unsigned row = idx < wtotal ? (unsigned)idx / (unsigned)CQ : 0u;
        unsigned t, co, jy, jx;
        fdivmod(row, fCout, t, co);
        fdivmod(t, fNx, jy, jx);
        const int ky = MERGE ? (int)jy : ph0.ky0 + p.kstep[1] * (int)jy;
        const int kx = MERGE ? (int)jx : ph0.kx0 + p.kstep[2] * (int)jx;
        const int tapflat = (ph0.kz0 * p.Ky + ky) * p.Kx + kx;
        wv[u] = *reinterpret_cast<const float4*>(gw + ((int)co * Ktot + tapflat * CIN + 4 * cq));
      }
    };
    auto store_w = [&](int base) {
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        const int idx = base + u * 256 + tid;
        if (idx < wtotal) *reinterpret_cast<float4*>(wl + (idx / CQ) * PC + 4 * cq) = wv[u];
      }
    };
    load_patch(0);
    if (wtotal > 0) load_w(0);
    store_patch(0);
    if (wtotal > 0) store_w(0);
    for (int base = 256 * LU; base < total; base += 256 * LU) {
      load_patch(base);
      store_patch(base);
    }
    for (int base = 256 * WU; base < wtotal; base += 256 * WU) {
      load_w(base);
      store_w(base);
    }
  }
  __syncthreads();

  const float* gres = p.resid;
  float* gout = p.out;
  const int ldo = p.ldo, ldr = p.ldr, tanh_out = p.tanh_out;
  const int osy = p.ostride[1], osx = p.ostride[2];
  for (int phase = phase0; phase < phase0 + nph; ++phase) {
  const Phase& ph = p.ph[phase];
  const int My = ph.Mz > 0 ? ph.My : 0, Mx = ph.Mx;
  const int ny = ph.nz > 0 ? ph.ny : 0, nx = ph.nx;
  const long srow = MERGE ? (long)wblk * p.nphase + phase : (long)wblk;
  if (MERGE && (my0 >= My || mx0 >= Mx)) {      // this phase has no pixel in the tile (block-uniform)
    if (p.stats && tid < Cout) {
      float* row = p.stats + srow * 2 * Cout;
      row[tid] = 0.f;
      row[Cout + tid] = 0.f;
    }
    continue;
  }
  float sm = 0.f, sq = 0.f;
  int scol;          // statistics column of this lane
  bool swrite;       // lane that publishes the wave's column sums

  if constexpr (!NARROW) {
    // ---- 32x32x2 tiles: wave = 2 tile rows x 16 columns of m-positions, 32 output channels ----
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const int tyl = 2 * wid + (li >> 4), txl = li & 15;
      const float* Arow = patch + ((tyl * isy - ylo) * PW + (txl * isx - xlo)) * PC + 4 * lh;
      const float* Brow = wl + (li < Cout ? li : 0) * PC + 4 * lh;
      for (int jy = 0; jy < ny; ++jy) {
        const int dy = ph.dy0 + dsy * jy;
        for (int jx = 0; jx < nx; ++jx) {
          const int dx = ph.dx0 + dsx * jx;
          const float* A = Arow + (dy * PW + dx) * PC;
          const float* B = Brow + (MERGE ? (ph.ky0 + p.kstep[1] * jy) * p.Kx + ph.kx0 + p.kstep[2] * jx : jy * nx + jx) * Cout * PC;
#pragma unroll
          for (int g = 0; g < CIN / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(A + 8 * g);
            const float4 b = *reinterpret_cast<const float4*>(B + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
          }
        }
      }
    }
    const int co = li;
    const float bv = (p.bias && co < Cout) ? p.bias[co] : 0.f;
    const bool ea = p.epi.scale != nullptr && co < Cout;      // epilogue activation (eval-mode inference)
    const float esc = ea ? p.epi.scale[co] : 1.f, esh = ea ? p.epi.shift[co] : 0.f, esl = ea ? p.epi.slope[co] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;      // row inside the wave's 32
      const int my = my0 + 2 * wid + (rl >> 4), mx = mx0 + (rl & 15);
      const int oy = my * osy + ph.oy, ox = mx * osx + ph.ox;
      const bool ok = my < My && mx < Mx && oy < p.Ho && ox < p.Wo && co < Cout;
      if (ok) {
        const long pix = ((long)n * p.Ho + oy) * p.Wo + ox;
        float v = acc[r] + bv;
        sm += v;
        sq += v * v;
        if (ea) v = epi_act1(v, esc, esh, esl);
        if (gres) v += gres[pix * ldr + co];
        if (tanh_out) v = tanhf(v);
        gout[pix * ldo + co] = v;
      }
    }
    sm += __shfl_xor(sm, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    scol = li;
    swrite = lh == 0;
  } else {
    // ---- <= 16 output channels: 16x16x4 tiles (no padded columns); wave = 2 tile rows, one
    //      accumulator per row; lane (c = lane & 15, kq = lane >> 4) feeds channels 4*kq..4*kq+3
    //      of each 16-channel group through the four MFMAs of a quad ----
    const int l16 = lane & 15, kq = lane >> 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* A0row = patch + (((2 * wid) * isy - ylo) * PW + (l16 * isx - xlo)) * PC + 4 * kq;
      const float* A1row = A0row + isy * PW * PC;
      const float* Brow = wl + (l16 < Cout ? l16 : 0) * PC + 4 * kq;
      for (int jy = 0; jy < ny; ++jy) {
        const int dy = ph.dy0 + dsy * jy;
        for (int jx = 0; jx < nx; ++jx) {
          const int dx = ph.dx0 + dsx * jx;
          const int aoff = (dy * PW + dx) * PC;
          const float* B = Brow + (MERGE ? (ph.ky0 + p.kstep[1] * jy) * p.Kx + ph.kx0 + p.kstep[2] * jx : jy * nx + jx) * Cout * PC;
#pragma unroll
          for (int g = 0; g < CIN / 16; ++g) {
            const float4 a0 = *reinterpret_cast<const float4*>(A0row + aoff + 16 * g);
            const float4 a1 = *reinterpret_cast<const float4*>(A1row + aoff + 16 * g);
            const float4 b = *reinterpret_cast<const float4*>(B + 16 * g);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
          }
        }
      }
    }
    const int co = l16;
    const float bv = (p.bias && co < Cout) ? p.bias[co] : 0.f;
    const bool ea = p.epi.scale != nullptr && co < Cout;      // epilogue activation (eval-mode inference)
    const float esc = ea ? p.epi.scale[co] : 1.f, esh = ea ? p.epi.shift[co] : 0.f, esl = ea ? p.epi.slope[co] : 1.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int my = my0 + 2 * wid + mt, mx = mx0 + 4 * kq + r;
        const int oy = my * osy + ph.oy, ox = mx * osx + ph.ox;
        const bool ok = my < My && mx < Mx && oy < p.Ho && ox < p.Wo && co < Cout;
        if (ok) {
          const long pix = ((long)n * p.Ho + oy) * p.Wo + ox;
          float v = (mt == 0 ? acc0[r] : acc1[r]) + bv;
          sm += v;
          sq += v * v;
          if (ea) v = epi_act1(v, esc, esh, esl);
          if (gres) v += gres[pix * ldr + co];
          if (tanh_out) v = tanhf(v);
          gout[pix * ldo + co] = v;
        }
      }
    sm += __shfl_xor(sm, 16, 64);
    sq += __shfl_xor(sq, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    scol = l16;
    swrite = kq == 0;
  }

  if (p.stats) {
    __syncthreads();                         // !MERGE: the patch is dead, reuse its head for the wave partials
    float* st = MERGE ? lds + pl.stats_off : lds;   // [4 waves][2][32]
    if (swrite) {
      st[(wid * 2 + 0) * 32 + scol] = sm;
      st[(wid * 2 + 1) * 32 + scol] = sq;
    }
    __syncthreads();
    if (tid < Cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += st[(w * 2 + 0) * 32 + tid];
        b += st[(w * 2 + 1) * 32 + tid];
      }
      float* row = p.stats + srow * 2 * Cout;
      row[tid] = a;
      row[Cout + tid] = b;
    }
  }
  }   // phases
}

// ---------------------------------------------------------------------------
// Persistent form of the patch kernel (one phase, or all phases merged): a block stages the weights ONCE and
// then walks several tiles; the input patch of tile t+1 is in flight (global loads into registers) while
// tile t is contracted from LDS and its results are stored, and goes to LDS behind one barrier.  Against the
// one-tile-per-block form above this removes the per-tile weight staging, takes the patch load's round trip off
// the critical path and lets a block's output stores overlap its next input loads -- the three things that kept
// those launches (all blocks resident at once, all of them load -> compute -> store in lockstep) at 25-45 % of
// either roof.  LDS: [weights][patch][statistics slots]; statistics rows: one per (tile, phase), as before.
// ---------------------------------------------------------------------------
template <int CIN, int PRO, bool NARROW, bool MERGE>
__global__ __launch_bounds__(256) void gather_patch_persist_kernel(const GatherConv p, const PatchLaunch pl,
                                                                   const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int CQ = CIN / 4;
  constexpr int LU = CIN == 16 ? 9 : 12;      // patch chunks per thread: the WHOLE patch of a tile in one round (host checks)
  // weight chunks per thread and trip: ALL of a 3x3 layer's weights in one round of loads (32 -> 32: 2304 chunks = 9
  // per thread; with 4 per trip the block spent three dependent round trips -- 8.6 us measured by the phase stamps,
  // 37 % of its life -- before its first MFMA)
  // -- and branch-free: slots past the weights re-read row 0 (an L1 hit) instead of testing (a test per slot puts
  // every load into a basic block of its own behind its own wait: 18 us).  3x3 taps, Cout <= 16 (NARROW) or <= 32.
  constexpr int WU = NARROW ? (CIN == 16 ? 3 : (CIN == 32 ? 5 : 9)) : (CIN == 16 ? 5 : 9);
  // LDS pitches (floats per pixel / per weight row), chosen by the host so that the fragments' ds_read_b128 are
  // conflict-free: CIN + 8 for the 16x16x4 form (CIN + 4 cost 61 % of the LDS cycles in bank conflicts there,
  // rocprofv3 SQ_LDS_BANK_CONFLICT), CIN + 4 for the 32x32x2 form.
  const int PCA = pl.pca, PCW = pl.pcw;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int Cout = p.Cout;
  const int Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  const int isy = p.istride[1], isx = p.istride[2];
  const int dsy = p.dstep[1], dsx = p.dstep[2];
  const int nph = MERGE ? p.nphase : 1;
  int ylo, yhi, xlo, xhi, sntaps;
  if constexpr (MERGE) {
    ylo = pl.ylo; yhi = pl.yhi; xlo = pl.xlo; xhi = pl.xhi;
    sntaps = p.Ky * p.Kx;
  } else {
    const Phase& ph = p.ph[0];
    const int ny = ph.nz > 0 ? ph.ny : 0, nx = ph.nx;
    const int ye = ph.dy0 + dsy * (ny > 0 ? ny - 1 : 0), xe = ph.dx0 + dsx * (nx > 0 ? nx - 1 : 0);
    ylo = ph.dy0 < ye ? ph.dy0 : ye; yhi = ph.dy0 < ye ? ye : ph.dy0;
    xlo = ph.dx0 < xe ? ph.dx0 : xe; xhi = ph.dx0 < xe ? xe : ph.dx0;
    sntaps = ny * nx;
  }
  const int PH = (PT_H - 1) * isy + (yhi - ylo) + 1, PW = (PT_W - 1) * isx + (xhi - xlo) + 1;
  float* wl = lds;
  float* patch = lds + pl.w_floats;
  float* stslots = patch + pl.patch_floats;          // [nph][4 waves][2][32]
  float* fold_sc = stslots + nph * 256;              // [CIN] scale, [CIN] shift, then 4*CIN long long of scratch (fold only)
  float* fold_sh = fold_sc + CIN;
  const bool folding = PRO != 0 && p.fold.acc != nullptr;
  const bool want_stats = p.stats != nullptr || p.stats_acc != nullptr;
  float tot_a = 0.f, tot_b = 0.f;                    // this block's sums over all its tiles (accumulator form)
  const float* __restrict__ gw = p.wp;
  const int cq = tid & (CQ - 1);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  float slope = 1.f;
  int act = 0;
  if constexpr (PRO != 0) {
    slope = pro_slope(p.pro);
    act = p.pro.act;
  }
  const int total = PH * PW * CQ;
  const FastDiv fPW = MERGE ? pl.fPWm : pl.fPW[0], fNx = MERGE ? pl.fKx : pl.fNx[0], fCout = pl.fCout;
  const Phase& ph0 = p.ph[0];
  MPGAN_STAMP(p, 0);
  MPGAN_STAMP_VALUE(p, 6, 2);                 // kernel kind: persistent patch
  unsigned long long st_contract = 0, st_tail = 0, st_tiles = 0, st_mfma = 0;
  (void)st_contract; (void)st_tail; (void)st_tiles; (void)st_mfma;

  // ---- weights: once per block.  Staged BEHIND the first patch's loads (stage_weights() is called once they
  //      are in flight): the block's two cold round trips -- weights, first patch -- overlap instead of adding up ----
  auto stage_weights = [&]() {
    const int wtotal = sntaps * Cout * CQ;
    const int Ktot = p.Kz * p.Ky * p.Kx * CIN;
    for (int base = 0; base < wtotal; base += 256 * WU) {
      float4 wv[WU];
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        const int idx = base + u * 256 + tid;
        const unsigned row = idx < wtotal ? (unsigned)idx / (unsigned)CQ : 0u;
        unsigned t, co, jy, jx;
        fdivmod(row, fCout, t, co);
        fdivmod(t, fNx, jy, jx);
        const int ky = MERGE ? (int)jy : ph0.ky0 + p.kstep[1] * (int)jy;
        const int kx = MERGE ? (int)jx : ph0.kx0 + p.kstep[2] * (int)jx;
        const int tapflat = (ph0.kz0 * p.Ky + ky) * p.Kx + kx;
        wv[u] = *reinterpret_cast<const float4*>(gw + ((int)co * Ktot + tapflat * CIN + 4 * cq));
      }
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        const int idx = base + u * 256 + tid;
        if (idx < wtotal) *reinterpret_cast<float4*>(wl + (idx / CQ) * PCW + 4 * cq) = wv[u];
      }
    }
  };
  if (folding) {   // BatchNorm of the producer, finalised here instead of by a launch of its own (norm_fold.h)
    fold_stats_block(p.fold, CIN, reinterpret_cast<long long*>(fold_sh + CIN), fold_sc, fold_sh, tid, 256, blockIdx.x == 0);
    if constexpr (PRO != 0) {
      sc = *reinterpret_cast<const float4*>(fold_sc + 4 * cq);
      sh = *reinterpret_cast<const float4*>(fold_sh + 4 * cq);
    }
  }

  // ---- this thread's patch chunks: position inside the patch, decoded ONCE (the same for every tile) ----
  // pyx: (py << 16) | px, or -1 for a slot past the patch; per tile a chunk then costs a range test, two
  // multiply-adds and a select instead of a software division (the staging was the bulk of the 9.5 vector
  // instructions per MFMA this kernel family showed in SQ_INSTS_VALU / SQ_INSTS_MFMA).
  int pyx[LU];
  const int nu = (total + 255) / 256;       // register slots in use: the rest of the unrolled loops is skipped (block-uniform)
#pragma unroll
  for (int u = 0; u < LU; ++u) {
    pyx[u] = -1;
    if (u < nu) {
      const int idx = u * 256 + tid;
      unsigned py, px;
      fdivmod((unsigned)idx / (unsigned)CQ, fPW, py, px);
      pyx[u] = idx < total ? (int)((py << 16) | px) : -1;
    }
  }
  const float* zero4 = reinterpret_cast<const float*>(pl.zero_page);   // 16 bytes of zeros in global memory
  float4 pv[LU];
  unsigned pok = 0;
  auto decode = [&](int t, int& n, int& my0, int& mx0) {
    unsigned q, tx, ty, nn;
    fdivmod((unsigned)t, pl.fTx, q, tx);
    fdivmod(q, pl.fTy, nn, ty);
    my0 = (int)ty * PT_H;
    mx0 = (int)tx * PT_W;
    n = (int)nn;
  };
  float4 scn = sc, shn = sh;                          // prologue vectors of the tile held in pv (loaded WITH its patch)
  auto load_patch = [&](int t) {
    int n, my0, mx0;
    decode(t, n, my0, mx0);
    const int y0 = my0 * isy + ylo, x0 = mx0 * isx + xlo;
    const float* __restrict__ gin = p.in + (long)n * Hi * Wi * ldi + 4 * cq;
    if constexpr (PRO != 0) {
      if (!folding) {
        scn = *reinterpret_cast<const float4*>(p.pro.scale + (long)n * p.pro.n_stride + 4 * cq);
        shn = *reinterpret_cast<const float4*>(p.pro.shift + (long)n * p.pro.n_stride + 4 * cq);
      }
    }
    pok = 0;
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      if (u >= nu) break;
      const int iy = y0 + (pyx[u] >> 16), ix = x0 + (pyx[u] & 0xffff);
      const bool ok = pyx[u] >= 0 && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
      const float* src = ok ? gin + (iy * Wi + ix) * ldi : zero4;      // padding reads the zero page
      if (pl.dbg & 4) src = zero4;
      pv[u] = *reinterpret_cast<const float4*>(src);
      if constexpr (PRO != 0) pok |= (ok ? 1u : 0u) << u;              // act(0*scale + shift) != 0: masked after the prologue
    }
  };
  auto store_patch = [&]() {
    if constexpr (PRO != 0) {
      if (!folding) { sc = scn; sh = shn; }
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      if (u >= nu) break;
      float4 x = pv[u];
      if constexpr (PRO != 0) {
        x.x = act_apply(x.x * sc.x + sh.x, act, slope);
        x.y = act_apply(x.y * sc.y + sh.y, act, slope);
        x.z = act_apply(x.z * sc.z + sh.z, act, slope);
        x.w = act_apply(x.w * sc.w + sh.w, act, slope);
        const bool ok = (pok >> u) & 1u;
        x.x = ok ? x.x : 0.f; x.y = ok ? x.y : 0.f; x.z = ok ? x.z : 0.f; x.w = ok ? x.w : 0.f;
      }
      if (pyx[u] >= 0 && !(pl.dbg & 8))
        *reinterpret_cast<float4*>(patch + ((pyx[u] >> 16) * PW + (pyx[u] & 0xffff)) * PCA + 4 * cq) = x;
    }
  };

  const float* gres = p.resid;
  float* gout = p.out;
  const int ldo = p.ldo, ldr = p.ldr, tanh_out = p.tanh_out;
  const int osy = p.ostride[1], osx = p.ostride[2];
  // this lane's output channel is the same for every tile and phase: its bias is fetched ONCE, up front (loaded
  // behind each tile's MFMA loops it was an exposed L2 round trip per tile)
  const int bias_co = NARROW ? (lane & 15) : (lane & 31);
  const float bias_v = (p.bias && bias_co < Cout) ? p.bias[bias_co] : 0.f;
  const bool ea = p.epi.scale != nullptr;                      // epilogue activation (eval-mode inference): this lane's channel
  const float esc = (ea && bias_co < Cout) ? p.epi.scale[bias_co] : 1.f, esh = (ea && bias_co < Cout) ? p.epi.shift[bias_co] : 0.f,
              esl = (ea && bias_co < Cout) ? p.epi.slope[bias_co] : 1.f;

  // De-phase the blocks that share a CU (dispatch deals the first 256 blocks one per CU, then the next 256, ...):
  // identical blocks started together stay in lockstep -- all staging, then all contracting -- and leave the matrix
  // pipe idle while they stage.  pl.stagger x 64 clocks of sleep per dispatch round (0: off).
  if (pl.stagger > 0) {
    const int rounds = (int)(blockIdx.x >> 8);
    for (int i = 0; i < rounds * pl.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  int t = (int)blockIdx.x;
  if (t < ntiles) load_patch(t);              // in flight while the weights are fetched and staged
  stage_weights();
  MPGAN_STAMP(p, 1);                          // weights staged (their stores issued)
  if (t < ntiles) store_patch();
  __syncthreads();
  MPGAN_STAMP(p, 2);                          // first patch in LDS
  for (; t < ntiles; t += (int)gridDim.x) {
    const unsigned long long st_a = MPGAN_STAMP_NOW();
    const int tn = t + (int)gridDim.x;
    int n, my0, mx0;
    decode(t, n, my0, mx0);
    if (tn < ntiles) load_patch(tn);                 // in flight under this tile's contraction and stores

    for (int phase = 0; phase < nph; ++phase) {
      const Phase& ph = p.ph[phase];
      const int My = ph.Mz > 0 ? ph.My : 0, Mx = ph.Mx;
      const int ny = (ph.nz > 0 && !(pl.dbg & 2)) ? ph.ny : 0, nx = ph.nx;
      float* st = stslots + phase * 256;
      const bool live = my0 < My && mx0 < Mx;          // block-uniform
      float sm = 0.f, sq = 0.f;
      int scol = 0;
      bool swrite = false;
      const unsigned long long st_ph = MPGAN_STAMP_NOW();
      (void)st_ph;
      if (live) {
        // (32-bit element offsets in the epilogues below: the host admits only outputs below 2^31 elements)
        if constexpr (!NARROW) {
          const int li = lane & 31, lh = lane >> 5;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          // residual operand of the epilogue (backward-data accumulating into a gradient): fetched BEFORE the
          // contraction -- loaded beside the stores, each store waited out its own round trip (+4-7 us per tile)
          float rres[16];
          if (gres) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
              const int my = my0 + 2 * wid + rh, oy = my * osy + ph.oy;
              const bool rowok = my < My && oy < p.Ho && li < Cout;
              const int rowbase = ((n * p.Ho + oy) * p.Wo + ph.ox);
#pragma unroll
              for (int rq = 0; rq < 8; ++rq) {
                const int mx = mx0 + 4 * lh + (rq & 3) + 8 * (rq >> 2);
                const int ox = mx * osx;
                const bool ok = rowok && mx < Mx && ox + ph.ox < p.Wo;
                rres[rh * 8 + rq] = ok ? gres[(rowbase + ox) * ldr + li] : 0.f;
              }
            }
          }
          {
            const int tyl = 2 * wid + (li >> 4), txl = li & 15;
            const float* Arow = patch + ((tyl * isy - ylo) * PW + (txl * isx - xlo)) * PCA + 4 * lh;
            const float* Brow = wl + (li < Cout ? li : 0) * PCW + 4 * lh;
            for (int jy = 0; jy < ny; ++jy) {
              const int dy = ph.dy0 + dsy * jy;
              for (int jx = 0; jx < nx; ++jx) {
                const int dx = ph.dx0 + dsx * jx;
                const float* A = Arow + (dy * PW + dx) * PCA;
                const float* B = Brow + (MERGE ? (ph.ky0 + p.kstep[1] * jy) * p.Kx + ph.kx0 + p.kstep[2] * jx : jy * nx + jx) * Cout * PCW;
#pragma unroll
                for (int g = 0; g < CIN / 8; ++g) {
                  const float4 a = *reinterpret_cast<const float4*>(A + 8 * g);
                  const float4 b = *reinterpret_cast<const float4*>(B + 8 * g);
                  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                }
              }
            }
          }
          st_mfma += MPGAN_STAMP_NOW() - st_ph;
          const int co = li;
          const float bv = bias_v;
          // register r of lane (li, lh) is m-row 2*wid + (r >> 3), m-column 4*lh + (r & 3) + 8*((r >> 2) & 1)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh) {
            const int my = my0 + 2 * wid + rh, oy = my * osy + ph.oy;
            const bool rowok = my < My && oy < p.Ho && co < Cout;
            const int rowbase = ((n * p.Ho + oy) * p.Wo + ph.ox) ;
            float* tw = lds + pl.tw_off + wid * (16 * 40);     // this wave's transpose staging: [16 px][40]
#pragma unroll
            for (int rq = 0; rq < 8; ++rq) {
              const int r = rh * 8 + rq;
              const int pxl = 4 * lh + (rq & 3) + 8 * (rq >> 2);
              const int mx = mx0 + pxl;
              const int ox = mx * osx;
              const bool ok = rowok && mx < Mx && ox + ph.ox < p.Wo;
              float v = acc[r] + bv;
              if (ok) {
                sm += v;
                sq += v * v;
              }
              if (ea) v = epi_act1(v, esc, esh, esl);
              if (gres) v += rres[r];
              if (tanh_out) v = tanhf(v);
              if (pl.tw_off >= 0) tw[pxl * 40 + li] = v;       // vector path: through LDS, stored below
              else if (ok && !(pl.dbg & 1)) gout[(rowbase + ox) * ldo + co] = v;
            }
            if (pl.tw_off >= 0) {
              // a lane held ONE channel of 16 pixels (16 four-byte stores per tile: 2.8 us of epilogue per tile against
              // 1.6 us of MFMA loops, measured by the phase stamps); now lane (px, c4) stores 16 bytes
              __builtin_amdgcn_wave_barrier();
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const int pxl = (lane >> 3) + 8 * k, c4 = lane & 7;
                const float4 o = *reinterpret_cast<const float4*>(tw + pxl * 40 + 4 * c4);
                const int mx = mx0 + pxl, ox = mx * osx;
                if (my < My && oy < p.Ho && mx < Mx && ox + ph.ox < p.Wo && 4 * c4 < Cout && !(pl.dbg & 1))
                  *reinterpret_cast<float4*>(gout + (rowbase + ox) * ldo + 4 * c4) = o;
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
          sm += __shfl_xor(sm, 32, 64);
          sq += __shfl_xor(sq, 32, 64);
          scol = li;
          swrite = lh == 0;
        } else {
          const int l16 = lane & 15, kq = lane >> 4;
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          float rres[8];                                  // residual operand, fetched before the contraction (see above)
          if (gres) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const int my = my0 + 2 * wid + mt, oy = my * osy + ph.oy;
              const bool rowok = my < My && oy < p.Ho && l16 < Cout;
              const int rowbase = (n * p.Ho + oy) * p.Wo + ph.ox;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int mx = mx0 + 4 * kq + r;
                const int ox = mx * osx;
                const bool ok = rowok && mx < Mx && ox + ph.ox < p.Wo;
                rres[mt * 4 + r] = ok ? gres[(rowbase + ox) * ldr + l16] : 0.f;
              }
            }
          }
          {
            const float* A0row = patch + (((2 * wid) * isy - ylo) * PW + (l16 * isx - xlo)) * PCA + 4 * kq;
            const float* A1row = A0row + isy * PW * PCA;
            const float* Brow = wl + (l16 < Cout ? l16 : 0) * PCW + 4 * kq;
            for (int jy = 0; jy < ny; ++jy) {
              const int dy = ph.dy0 + dsy * jy;
              for (int jx = 0; jx < nx; ++jx) {
                const int dx = ph.dx0 + dsx * jx;
                const int aoff = (dy * PW + dx) * PCA;
                const float* B = Brow + (MERGE ? (ph.ky0 + p.kstep[1] * jy) * p.Kx + ph.kx0 + p.kstep[2] * jx : jy * nx + jx) * Cout * PCW;
#pragma unroll
                for (int g = 0; g < CIN / 16; ++g) {
                  const float4 a0 = *reinterpret_cast<const float4*>(A0row + aoff + 16 * g);
                  const float4 a1 = *reinterpret_cast<const float4*>(A1row + aoff + 16 * g);
                  const float4 b = *reinterpret_cast<const float4*>(B + 16 * g);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
                }
              }
            }
          }
          st_mfma += MPGAN_STAMP_NOW() - st_ph;
          const int co = l16;
          const float bv = bias_v;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const int my = my0 + 2 * wid + mt, oy = my * osy + ph.oy;
            const bool rowok = my < My && oy < p.Ho && co < Cout;
            const int rowbase = (n * p.Ho + oy) * p.Wo + ph.ox;
            float* tw = lds + pl.tw_off + wid * (16 * 20);     // this wave's transpose staging: [16 px][20]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int pxl = 4 * kq + r;
              const int mx = mx0 + pxl;
              const int ox = mx * osx;
              const bool ok = rowok && mx < Mx && ox + ph.ox < p.Wo;
              float v = (mt == 0 ? acc0[r] : acc1[r]) + bv;
              if (ok) {
                sm += v;
                sq += v * v;
              }
              if (ea) v = epi_act1(v, esc, esh, esl);
              if (gres) v += rres[mt * 4 + r];
              if (tanh_out) v = tanhf(v);
              if (pl.tw_off >= 0) tw[pxl * 20 + l16] = v;
              else if (ok && !(pl.dbg & 1)) gout[(rowbase + ox) * ldo + co] = v;
            }
            if (pl.tw_off >= 0) {                              // lane (px = lane / 4, c4 = lane % 4) stores 16 bytes (see above)
              __builtin_amdgcn_wave_barrier();
              const int pxl = lane >> 2, c4 = lane & 3;
              const float4 o = *reinterpret_cast<const float4*>(tw + pxl * 20 + 4 * c4);
              const int mx = mx0 + pxl, ox = mx * osx;
              if (my < My && oy < p.Ho && mx < Mx && ox + ph.ox < p.Wo && 4 * c4 < Cout && !(pl.dbg & 1))
                *reinterpret_cast<float4*>(gout + (rowbase + ox) * ldo + 4 * c4) = o;
              __builtin_amdgcn_wave_barrier();
            }
          }
          sm += __shfl_xor(sm, 16, 64);
          sq += __shfl_xor(sq, 16, 64);
          sm += __shfl_xor(sm, 32, 64);
          sq += __shfl_xor(sq, 32, 64);
          scol = l16;
          swrite = kq == 0;
        }
      }
      if (want_stats && swrite) {                     // a dead phase leaves sm = sq = 0 in its slots
        st[(wid * 2 + 0) * 32 + scol] = sm;
        st[(wid * 2 + 1) * 32 + scol] = sq;
      } else if (want_stats && !live && lane < 32) {
        st[(wid * 2 + 0) * 32 + lane] = 0.f;
        st[(wid * 2 + 1) * 32 + lane] = 0.f;
      }
    }
    const unsigned long long st_b = MPGAN_STAMP_NOW();
    __syncthreads();                                  // every wave is done with the patch; statistics slots are complete
    if (want_stats && tid < Cout) {
      for (int phase = 0; phase < nph; ++phase) {
        const float* st = stslots + phase * 256;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += st[(w * 2 + 0) * 32 + tid];
          b += st[(w * 2 + 1) * 32 + tid];
        }
        if (p.stats_acc) {
          tot_a += a;
          tot_b += b;
        } else {
          float* row = p.stats + ((long)t * nph + phase) * 2 * Cout;
          row[tid] = a;
          row[Cout + tid] = b;
        }
      }
    }
    if (tn < ntiles) store_patch();
    __syncthreads();
    st_contract += st_b - st_a;
    st_tail += MPGAN_STAMP_NOW() - st_b;
    st_tiles += 1;
  }
  MPGAN_STAMP_VALUE(p, 3, st_contract);       // sum over this block's tiles: contraction + epilogue stores issued
  MPGAN_STAMP_VALUE(p, 4, st_tail);           // ... barrier, statistics rows, next patch's arrival + LDS stores, barrier
  MPGAN_STAMP_VALUE(p, 5, st_tiles);
  MPGAN_STAMP_VALUE(p, 8, st_mfma);           // ... of which: residual prefetch + fragment reads + MFMAs (thread 0's wave)
  MPGAN_STAMP(p, 7);
  if (p.stats_acc && tid < Cout) {                   // one set of atomics per block, whatever its number of tiles
    long long* rep = p.stats_acc + (long)(blockIdx.x % (unsigned)p.acc_rep) * ACC_WORDS * Cout;
    acc_add(rep, Cout, 0, tid, tot_a);
    acc_add(rep, Cout, 2, tid, tot_b);
  }
}

// Geometry test for the patch kernel (pointer alignment is checked at launch).
static bool patch_plan(const GatherConv& p, PatchLaunch* out, int* smem_bytes) {
  static const bool off = dev_env("MPGAN_DBG_NO_PATCH") != nullptr;
  if (off) return false;
  if (p.Di != 1 || p.Do != 1 || p.Kz != 1) return false;
  if (p.nphase > 8) return false;                             // (PatchLaunch's per-phase tables; border-class phases: conv_geom.h)
  if (!(p.Cin == 16 || p.Cin == 32 || p.Cin == 64)) return false;
  if (p.Cout < 2 || p.Cout > 32) return false;
  if (p.pro.scale && p.pro.n_stride % 4 != 0) return false;
  if (p.ksplit > 1) return false;
  int maxMy = 0, maxMx = 0, maxpix = 0, maxtaps = 0;
  for (int i = 0; i < p.nphase; ++i) {
    const Phase& ph = p.ph[i];
    if (ph.nz > 1) return false;
    const int ny = ph.nz > 0 ? ph.ny : 0, nx = ph.nx;
    if (ny * nx > 9) return false;
    const int ys = (ny > 0 ? ny - 1 : 0) * abs(p.dstep[1]), xs = (nx > 0 ? nx - 1 : 0) * abs(p.dstep[2]);
    const int PH = (PT_H - 1) * p.istride[1] + ys + 1, PW = (PT_W - 1) * p.istride[2] + xs + 1;
    if (PH * PW > maxpix) maxpix = PH * PW;
    if (ny * nx > maxtaps) maxtaps = ny * nx;
    if (out) {
      out->fPW[i] = make_fastdiv((unsigned)PW);
      out->fNx[i] = make_fastdiv((unsigned)(nx > 0 ? nx : 1));
    }
    if (ph.Mz > 0 && ph.My > maxMy) maxMy = ph.My;
    if (ph.Mx > maxMx) maxMx = ph.Mx;
  }
  if (maxMy == 0 || maxMx == 0) return false;
  const int PC = p.Cin + 4;
  int patch_floats = maxpix * PC;
  if (patch_floats < 256) patch_floats = 256;              // the statistics partials reuse the head
  const long bytes = ((long)patch_floats + (long)maxtaps * p.Cout * PC) * 4;
  if (bytes > 96 * 1024) return false;
  // merged form: one block per tile walks every phase (see MERGE on the kernel)
  static const bool no_merge = dev_env("MPGAN_DBG_NO_MERGE") != nullptr;
  bool merged = p.nphase > 1 && !p.pro.scale && !no_merge;
  int ylo = 1 << 20, yhi = -(1 << 20), xlo = 1 << 20, xhi = -(1 << 20);
  long mbytes = 0;
  int mpatch = 0;
  if (merged) {
    for (int i = 0; i < p.nphase; ++i) {
      const Phase& ph = p.ph[i];
      if (ph.nz == 0) continue;
      const int ye = ph.dy0 + p.dstep[1] * (ph.ny - 1), xe = ph.dx0 + p.dstep[2] * (ph.nx - 1);
      ylo = std::min(ylo, std::min(ph.dy0, ye)); yhi = std::max(yhi, std::max(ph.dy0, ye));
      xlo = std::min(xlo, std::min(ph.dx0, xe)); xhi = std::max(xhi, std::max(ph.dx0, xe));
    }
    if (ylo > yhi) merged = false;
  }
  if (merged) {
    const int PHu = (PT_H - 1) * p.istride[1] + (yhi - ylo) + 1, PWu = (PT_W - 1) * p.istride[2] + (xhi - xlo) + 1;
    mpatch = PHu * PWu * PC;
    mbytes = ((long)mpatch + (long)p.Ky * p.Kx * p.Cout * PC + 256) * 4;
    if (mbytes > 96 * 1024) merged = false;
    else if (out) {
      out->fPWm = make_fastdiv((unsigned)PWu);
      out->fKx = make_fastdiv((unsigned)p.Kx);
    }
  }
  if (out) {
    out->tiles_x = (maxMx + PT_W - 1) / PT_W;
    out->tiles_y = (maxMy + PT_H - 1) / PT_H;
    out->patch_floats = merged ? mpatch : patch_floats;
    out->fCout = make_fastdiv((unsigned)p.Cout);
    out->merged = merged ? 1 : 0;
    out->ylo = ylo; out->yhi = yhi; out->xlo = xlo; out->xhi = xhi;
    out->stats_off = mpatch + p.Ky * p.Kx * p.Cout * PC;
  }
  if (smem_bytes) *smem_bytes = (int)(merged ? mbytes : bytes);
  return true;
}

template <int CIN, int PRO, bool NARROW, bool MERGE>
static int launch_patch_variant(const GatherConv& p, const PatchLaunch& pl, int smem, hipStream_t st) {
  auto kern = gather_patch_kernel<CIN, PRO, NARROW, MERGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) {
      set_error("gather_patch: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)(pl.tiles_x * pl.tiles_y * p.N * (MERGE ? 1 : p.nphase)));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p, pl);
  return check_launch("gather_patch");
}

template <int CIN>
static int launch_patch_cin(const GatherConv& p, const PatchLaunch& pl, int smem, hipStream_t st) {
  const bool pro = p.pro.scale != nullptr;
  if (pl.merged) {       // never with a prologue (patch_plan)
    if (p.Cout <= 16) return launch_patch_variant<CIN, 0, true, true>(p, pl, smem, st);
    return launch_patch_variant<CIN, 0, false, true>(p, pl, smem, st);
  }
  if (p.Cout <= 16)
    return pro ? launch_patch_variant<CIN, 1, true, false>(p, pl, smem, st)
               : launch_patch_variant<CIN, 0, true, false>(p, pl, smem, st);
  return pro ? launch_patch_variant<CIN, 1, false, false>(p, pl, smem, st)
             : launch_patch_variant<CIN, 0, false, false>(p, pl, smem, st);
}

// ---- persistent form -------------------------------------------------------------------------
template <int CIN, int PRO, bool NARROW, bool MERGE>
static int launch_patch_persist_variant(const GatherConv& p, const PatchLaunch& pl, int smem, int ntiles, int grid,
                                        hipStream_t st) {
  auto kern = gather_patch_persist_kernel<CIN, PRO, NARROW, MERGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) {
      set_error("gather_patch_persist: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, st, p, pl, ntiles);
  return check_launch("gather_patch_persist");
}

template <int CIN>
static int launch_patch_persist_cin(const GatherConv& p, const PatchLaunch& pl, int smem, int ntiles, int grid,
                                    hipStream_t st) {
  const bool pro = p.pro.scale != nullptr;
  if (pl.merged) {
    if (p.Cout <= 16) return launch_patch_persist_variant<CIN, 0, true, true>(p, pl, smem, ntiles, grid, st);
    return launch_patch_persist_variant<CIN, 0, false, true>(p, pl, smem, ntiles, grid, st);
  }
  if (p.Cout <= 16)
    return pro ? launch_patch_persist_variant<CIN, 1, true, false>(p, pl, smem, ntiles, grid, st)
               : launch_patch_persist_variant<CIN, 0, true, false>(p, pl, smem, ntiles, grid, st);
  return pro ? launch_patch_persist_variant<CIN, 1, false, false>(p, pl, smem, ntiles, grid, st)
             : launch_patch_persist_variant<CIN, 0, false, false>(p, pl, smem, ntiles, grid, st);
}

// Plan of the persistent form; false: keep the one-tile-per-block kernel (several unmerged phases, a patch
// that does not fit one register round, LDS).
static bool patch_persist_plan(const GatherConv& p, const PatchLaunch& pl, PatchLaunch* out, int* smem, int* ntiles,
                               int* grid) {
  static const bool off = dev_env("MPGAN_DBG_NO_PERSIST") != nullptr;
  if (off) return false;
  if (p.nphase != 1 && !pl.merged) return false;
  const int PC = p.Cin + 4, CQ = p.Cin / 4;
  int PH, PW, wtaps;
  if (pl.merged) {
    PH = (PT_H - 1) * p.istride[1] + (pl.yhi - pl.ylo) + 1;
    PW = (PT_W - 1) * p.istride[2] + (pl.xhi - pl.xlo) + 1;
    wtaps = p.Ky * p.Kx;
  } else {
    const Phase& ph = p.ph[0];
    const int ny = ph.nz > 0 ? ph.ny : 0, nx = ph.nx;
    PH = (PT_H - 1) * p.istride[1] + (ny > 0 ? ny - 1 : 0) * abs(p.dstep[1]) + 1;
    PW = (PT_W - 1) * p.istride[2] + (nx > 0 ? nx - 1 : 0) * abs(p.dstep[2]) + 1;
    wtaps = ny * nx;
  }
  const int LU = p.Cin == 16 ? 9 : 12;
  if (PH * PW * CQ > LU * 256) return false;
  if (PH > 0xffff || PW > 0xffff) return false;
  // 32-bit element offsets in the kernel
  if ((long)p.N * p.Ho * p.Wo * (p.ldo > p.ldr ? p.ldo : p.ldr) >= (1L << 31) || (long)p.Hi * p.Wi * p.ldi >= (1L << 31)) return false;
  *out = pl;
  const bool narrow = p.Cout <= 16;
  out->pcw = narrow ? p.Cin + 8 : PC;                                   // see the kernel: conflict-free ds_read_b128
  out->pca = narrow ? (p.istride[2] == 1 ? p.Cin + 8 : p.Cin + 4) : PC;
  out->w_floats = (wtaps * p.Cout * out->pcw + 3) & ~3;
  int pf = PH * PW * out->pca;
  out->patch_floats = (pf + 3) & ~3;
  {
    out->fTx = make_fastdiv((unsigned)pl.tiles_x);
    out->fTy = make_fastdiv((unsigned)pl.tiles_y);
    static const int dbg_skip = dev_env("MPGAN_DBG_PATCH_SKIP") ? atoi(dev_env("MPGAN_DBG_PATCH_SKIP")) : 0;
    out->dbg = dbg_skip;
    static const int stag = dev_env("MPGAN_DBG_PATCH_STAGGER") ? atoi(dev_env("MPGAN_DBG_PATCH_STAGGER")) : 0;
    out->stagger = stag;
    static void* zp = nullptr;          // address of the device-side zero page, looked up once
    if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(g_patch_zero_page)) != hipSuccess) { zp = nullptr; return false; }
    out->zero_page = zp;
  }
  const int nph = pl.merged ? p.nphase : 1;
  long floats = (long)out->w_floats + out->patch_floats + nph * 256 + 2 * p.Cin + 8 * p.Cin + 4;   // + fold scratch
  floats = (floats + 3) & ~3L;
  // vector output path: one tile row of a wave (16 pixels x Cout) goes through LDS and leaves as 16-byte stores
  const bool vec_out = (p.Cout % 4 == 0) && (p.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
  out->tw_off = -1;
  if (vec_out) {
    out->tw_off = (int)floats;
    floats += 4 * 16 * (narrow ? 20 : 40);
  }
  const long bytes = floats * 4;
  if (bytes > 150 * 1024) return false;
  *smem = (int)bytes;
  *ntiles = pl.tiles_x * pl.tiles_y * p.N;
  static const int bpc_env = dev_env("MPGAN_DBG_PATCH_BPC") ? atoi(dev_env("MPGAN_DBG_PATCH_BPC")) : 0;
  int bpc = (int)((160L * 1024) / bytes);
  if (bpc > 3) bpc = 3;
  if (bpc < 1) bpc = 1;
  if (bpc_env > 0) bpc = bpc_env;
  int g = 256 * bpc;
  if (g > *ntiles) g = *ntiles;
  *grid = g;
  return true;
}

static int launch_patch(const GatherConv& p, const PatchLaunch& pl, int smem, hipStream_t st) {
  {
    PatchLaunch pp;
    int psmem = 0, ntiles = 0, grid = 0;
    const bool persist = patch_persist_plan(p, pl, &pp, &psmem, &ntiles, &grid);
    MPGAN_UNSUPPORTED(!persist && (p.fold.acc || p.stats_acc),
                      "gather_patch: accumulator statistics / fold-on-load need the persistent patch kernel "
                      "(mpgan_conv_acc_supported / mpgan_conv_fold_supported said otherwise?)");
    if (persist) {
      switch (p.Cin) {
        case 16: return launch_patch_persist_cin<16>(p, pp, psmem, ntiles, grid, st);
        case 32: return launch_patch_persist_cin<32>(p, pp, psmem, ntiles, grid, st);
        default: return launch_patch_persist_cin<64>(p, pp, psmem, ntiles, grid, st);
      }
    }
  }
  switch (p.Cin) {
    case 16: return launch_patch_cin<16>(p, pl, smem, st);
    case 32: return launch_patch_cin<32>(p, pl, smem, st);
    default: return launch_patch_cin<64>(p, pl, smem, st);
  }
}

template <int BN, int TM, int TN, int WN, bool SCALAR>
static int launch_variant(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_kernel<BN, TM, TN, WN, SCALAR>;
  static const int lds_pad = dev_env("MPGAN_DBG_LDS_PAD") ? atoi(dev_env("MPGAN_DBG_LDS_PAD")) : 0;
  const int smem = 2 * (BM + BN) * PITCH * (int)sizeof(float) + lds_pad;
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
  GatherConv q = p;
  const long pairs = set_tile_grid(q, BM);
  q.ntiles = (p.Cout + BN - 1) / BN;
  q.phase_outer = (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 4 > (3L << 20) ? 1 : 0;
  q.ksplit = 1;
  dim3 grid((unsigned)(pairs * q.ntiles));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q);
  return check_launch("gather_conv");
}

// Kernel selection, shared by the launcher and mpgan_conv_variant():
//   1 = thin_cin1, 2 = thin_cout1, 32/64/128 = gather_conv_kernel<BN>; mpgan_conv_variant() adds
//   16 = gather_patch_kernel (see patch_plan).

static int select_variant(const GatherConv& p, long maxM, bool thin1, bool thin2) {
  if (thin1) return 1;
  if (thin2) return 2;
  // Largest channel tile that still yields >= 2 blocks per CU; small-M layers (the
  // U-Net bottom: 16K pixels) take narrower tiles rather than leave CUs idle.
  // Widest channel tile that still gives every CU a block: the pipelined kernel keeps the
  // matrix pipe ~85 % fed from ONE resident block, while narrow tiles (16 MFMAs per K-step
  // and wave) cannot cover their own load/store/barrier overhead.
  const long mtiles = phase_tile_rows(p, BM);
  static const int force_bn = dev_env("MPGAN_DBG_BN") ? atoi(dev_env("MPGAN_DBG_BN")) : 0;   // experiments
  if (force_bn == 32 || force_bn == 64 || force_bn == 128) return force_bn;
  int bn = 32;
  if (p.Cout > 64 && mtiles * ((p.Cout + 127) / 128) >= 256) bn = 128;
  else if (p.Cout > 32 && mtiles * ((p.Cout + 63) / 64) >= 256) bn = 64;
  else if (p.Cout > 64 && mtiles * ((p.Cout + 63) / 64) >= 128) bn = 64;
  return bn;
}

template <int BN, int TM, int TN, int WN, int WRAPS, int PRO, bool FAST = false, int KS = 1>
static int launch_pipe_variant(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_pipe_kernel<BN, TM, TN, WN, WRAPS, PRO, FAST, KS>;
  static const int lds_pad = dev_env("MPGAN_DBG_LDS_PAD") ? atoi(dev_env("MPGAN_DBG_LDS_PAD")) : 0;
  const int smem = KS * 2 * (BM + BN) * PITCH * (int)sizeof(float) + lds_pad;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_conv_pipe: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  GatherConv q = p;
  const long pairs = set_tile_grid(q, BM);
  q.ntiles = (p.Cout + BN - 1) / BN;
  q.phase_outer = (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 4 > (3L << 20) ? 1 : 0;
  if (q.ksplit < 1) q.ksplit = 1;
  dim3 grid((unsigned)(pairs * q.ntiles * q.ksplit));
  hipLaunchKernelGGL(kern, grid, dim3(256 * KS), smem, st, q);
  return check_launch("gather_conv_pipe");
}

// every tap of every pixel in range and no partial channel tile: see FAST on gather_conv_pipe_kernel
static bool fast_geometry(const GatherConv& p, int bn) {
  static const bool off = dev_env("MPGAN_DBG_NO_FAST") != nullptr;
  if (off || p.nphase != 1 || p.Cout % bn != 0 || p.Cin % 32 != 0 || p.ksplit > 1) return false;
  const Phase& ph = p.ph[0];
  if (ph.nz * ph.ny * ph.nx == 0) return false;
  const int d0[3] = {ph.dz0, ph.dy0, ph.dx0}, nj[3] = {ph.nz, ph.ny, ph.nx}, M[3] = {ph.Mz, ph.My, ph.Mx};
  const int G[3] = {p.Di, p.Hi, p.Wi};
  for (int d = 0; d < 3; ++d) {
    const int lo = p.dstep[d] > 0 ? d0[d] : d0[d] + p.dstep[d] * (nj[d] - 1);
    const int hi = p.dstep[d] > 0 ? d0[d] + p.dstep[d] * (nj[d] - 1) : d0[d];
    if (lo < 0 || (M[d] - 1) * p.istride[d] + hi > G[d] - 1) return false;
  }
  return true;
}

// In-block split-K (KS = 2 of gather_conv_pipe_kernel): when the output grid gives at most ~1.5 blocks per CU and
// every K group still gets >= 4 K-steps.
static bool pipe_wants_ksplit2(const GatherConv& p, int bn, long maxM) {
  static const bool off = dev_env("MPGAN_DBG_NO_KS2") != nullptr;
  if (off || p.ksplit > 1 || p.Cin % 32 != 0) return false;
  const long blocks = phase_tile_rows(p, BM) * ((p.Cout + bn - 1) / bn);
  if (blocks > 384) return false;
  int min_nk = 1 << 30;
  for (int i = 0; i < p.nphase; ++i) {
    const int nk = p.ph[i].nz * p.ph[i].ny * p.ph[i].nx * p.Cin / BK;
    if (nk > 0 && nk < min_nk) min_nk = nk;
  }
  return min_nk >= 8 && min_nk < (1 << 30);
}

// The DMA-staged form (gather_conv_dma_kernel) serves prologue-free gathers with enough tiles to fill the chip
// several times over: the discriminator's backward-data launches.  MPGAN_DBG_NO_DMA=1 turns it off (A/B runs);
// the threshold is the geometry's own `min_blocks` (tests pass 1 to run small shapes through it).
static bool dma_form_ok(const GatherConv& p, int variant, long maxM) {
  static const bool off = dev_env("MPGAN_DBG_NO_DMA") != nullptr;
  if (off || p.pro.scale || p.ksplit > 1 || p.Cin % 32 != 0 || (variant != 128 && variant != 64 && variant != 32)) return false;
  if (p.stats_acc || p.fold.acc) return false;
  const long blocks = phase_tile_rows(p, BM) * ((p.Cout + variant - 1) / variant);
  if (blocks < (p.min_blocks > 0 ? p.min_blocks : FORM_MIN_BLOCKS_DEFAULT)) return false;
  for (int i = 0; i < p.nphase; ++i) {
    const int nt = p.ph[i].nz * p.ph[i].ny * p.ph[i].nx;
    if (nt < 1 || nt > 32) return false;                   // one validity bit per tap and row
  }
  const long bytes_a = (long)p.N * p.Di * p.Hi * p.Wi * p.ldi * 4, ktot4 = (long)p.Kz * p.Ky * p.Kx * p.Cin * 4;
  return bytes_a < (long)HW_OOB && (long)p.Cout * ktot4 < (long)HW_OOB && ktot4 < 0xFFFF;
}

template <int BN, int TM, int TN, int WN, int NST = 2>
static int launch_dma_variant(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_dma_kernel<BN, TM, TN, WN, NST>;
  constexpr int smem_loop = NST * (BM + BN) * 128, smem_epi = (2048 + 4 * 32 * 36 + 1024) * 4;   // (conv_epilogue's scratch)
  constexpr int smem = smem_loop > smem_epi ? smem_loop : smem_epi;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_conv_dma: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  GatherConv q = p;
  const long pairs = set_tile_grid(q, BM);
  q.ntiles = (p.Cout + BN - 1) / BN;
  q.phase_outer = (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 4 > (3L << 20) ? 1 : 0;
  q.ksplit = 1;
  dim3 grid((unsigned)(pairs * q.ntiles));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q);
  return check_launch("gather_conv_dma");
}

template <int WRAPS, int PRO>
static int launch_pipe_bn(const GatherConv& p, int variant, long maxM, hipStream_t st) {
  if constexpr (WRAPS == 1 && PRO == 3) {
    if (variant == 128 && fast_geometry(p, 128)) return launch_pipe_variant<128, 2, 2, 2, 1, 3, true>(p, maxM, st);
  }
  if constexpr (WRAPS == 1 && PRO != 3) {
    if (variant != 128 && pipe_wants_ksplit2(p, variant, maxM)) {
      if (variant == 64) return launch_pipe_variant<64, 1, 2, 1, 1, PRO, false, 2>(p, maxM, st);
      return launch_pipe_variant<32, 1, 1, 1, 1, PRO, false, 2>(p, maxM, st);
    }
  }
  if constexpr (WRAPS == 1 && PRO == 0) {
    if (dma_form_ok(p, variant, maxM)) {
      if (variant == 128) return launch_dma_variant<128, 2, 2, 2>(p, maxM, st);
      if (variant == 64) return launch_dma_variant<64, 1, 2, 1>(p, maxM, st);
      return launch_dma_variant<32, 1, 1, 1, 2>(p, maxM, st);
    }
  }
  if (variant == 128) return launch_pipe_variant<128, 2, 2, 2, WRAPS, PRO>(p, maxM, st);
  if (variant == 64) return launch_pipe_variant<64, 1, 2, 1, WRAPS, PRO>(p, maxM, st);
  return launch_pipe_variant<32, 1, 1, 1, WRAPS, PRO>(p, maxM, st);
}

template <int WRAPS>
static int launch_pipe_pro(const GatherConv& p, int variant, long maxM, hipStream_t st) {
  if (!p.pro.scale) return launch_pipe_bn<WRAPS, 0>(p, variant, maxM, st);
  if (p.pro.n_stride == 0 && p.pro.act == MPGAN_ACT_LEAKY && !p.pro.slope_ptr && p.pro.slope >= 0.f &&
      p.pro.slope <= 1.f)
    return launch_pipe_bn<WRAPS, 3>(p, variant, maxM, st);
  if (p.pro.n_stride == 0) return launch_pipe_bn<WRAPS, 1>(p, variant, maxM, st);
  return launch_pipe_bn<WRAPS, 2>(p, variant, maxM, st);
}

// ---------------------------------------------------------------------------
// 3-D patch kernel, 16 -> 16 channels, 3x3x3, stride 1 (the U-Net's 64^3 level at config C5: down0.unit1, the
// up-path ResidualUnit conv and their backward-data gathers -- the generator's most expensive layers there).
// The K-stepped kernel pads 16 output channels to a 32-wide MFMA tile and re-stages every input pixel once per
// tap; here a block owns 2 x 8 x 8 output pixels, stages their 4 x 10 x 10 input patch ONCE (producer's
// BatchNorm + PReLU applied per element on the way in; out-of-range = the conv's zero padding) together with all
// 27 taps' weights, and contracts from LDS with v_mfma_f32_16x16x4_f32 (no padded columns): per tap and wave two A
// reads, one B read, eight MFMAs.  73 KiB of LDS: two blocks per CU, one's staging under the other's MFMAs.
// Fused BatchNorm statistics: one [sum | sum^2] row per tile.
// ---------------------------------------------------------------------------
constexpr int P3_TZ = 2, P3_TY = 8, P3_TX = 8;
constexpr int P3_PY = P3_TY + 2, P3_PX = P3_TX + 2;
constexpr int P3_PROWS = (P3_TZ + 2) * P3_PY * P3_PX;          // 400 patch pixels
constexpr int P3_PA = 24, P3_PW = 20;                          // LDS pitches (floats) of a patch pixel / a weight row
constexpr int P3_SMEM = (P3_PROWS * P3_PA + 27 * 16 * P3_PW + 4 * 2 * 16) * 4;
constexpr int P3M_PXP = 16;                                    // MM16: pixels per patch row in LDS
constexpr int P3M_PATCHB = (P3_TZ + 2) * P3_PY * P3M_PXP * 32, P3M_WB = 27 * 16 * 32;
constexpr int P3M_SMEM = P3M_PATCHB + P3M_WB + 4 * 2 * 16 * 4;
struct P3Grid { int tiles_z, tiles_y, tiles_x; };

//   MM16 (MPGAN_CONV_MM_BF16): the contraction on v_mfma_f32_16x16x32_bf16 -- K = 32 is two taps x 16 channels; lane
//   (row ln, k-quarter kq) holds the 8 channels 8 (kq & 1) .. + 7 of tap 2 t + (kq >> 1): 14 MFMAs per 16-row block
//   replace 108 (the 28th tap slot multiplies zero weights).  Patch and weights are rounded to bf16 ONCE, on their way
//   into LDS (a first version kept them fp32 and converted at fragment-read time: 336 conversions per lane and tile
//   and 3-4-way bank conflicts of the per-lane tap offsets made it SLOWER than the fp32 kernel, 221 vs 165 us).
//   Images: a pixel / weight row is 32 unpadded bytes, a patch row 16 pixels (10 used), so the ds_read_b128 of a
//   16-lane group -- 8 lanes x channels 0-7 and 8 lanes x channels 8-15 over two patch rows -- covers 16 distinct
//   16-byte slots: chunk index 2 (x + 16 y) + (kq & 1) mod 16.
template <bool HAS_PRO, bool MM16 = false>
__global__ __launch_bounds__(256, 2) void gather_patch3d_c16_kernel(const GatherConv p, const P3Grid tg) {
  extern __shared__ __attribute__((aligned(16))) float sm3[];
  float* patch = sm3;
  float* wl = sm3 + P3_PROWS * P3_PA;
  float* st = MM16 ? sm3 + (P3M_PATCHB + P3M_WB) / 4 : wl + 27 * 16 * P3_PW;   // [4 waves][2][16]
  char* patchb = reinterpret_cast<char*>(sm3);                 // MM16 images (bf16)
  char* wlb = patchb + P3M_PATCHB;
  typedef __bf16 p3_bf16x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const Phase& ph = p.ph[0];
  // Persistent blocks: the 27 taps' weights are staged once, then the block walks tiles (XCD-contiguous ranges).
  // All staging is "issue every load of the round, then store": a bounds test around each load had put every chunk
  // into a basic block of its own behind its own memory round trip (seven per tile); and the NEXT tile's patch is
  // fetched into registers under the current tile's contraction.
  constexpr int NPCH = (P3_PROWS * 4 + 255) / 256;               // 7 patch chunks per thread
  constexpr int NWCH = (27 * 16 * 4 + 255) / 256;                // 7 weight chunks per thread
  const int c4 = tid & 3;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  float slope = 1.f;
  if constexpr (HAS_PRO) {
    sc = *reinterpret_cast<const float4*>(p.pro.scale + 4 * c4);
    sh = *reinterpret_cast<const float4*>(p.pro.shift + 4 * c4);
    slope = pro_slope(p.pro);
  }
  const int act = p.pro.act;
  int pzyx[NPCH];                                                // (pz << 16) | (py << 8) | px of this thread's chunks, -1 past the patch
#pragma unroll
  for (int i = 0; i < NPCH; ++i) {
    const int pr = (tid + 256 * i) >> 2;
    const int pz = pr / (P3_PY * P3_PX), rem = pr - pz * (P3_PY * P3_PX);
    const int py = rem / P3_PX, px = rem - py * P3_PX;
    pzyx[i] = pr < P3_PROWS ? ((pz << 16) | (py << 8) | px) : -1;
  }
  const int mnz = ph.dz0 + (p.dstep[0] < 0 ? 2 * p.dstep[0] : 0), mny = ph.dy0 + (p.dstep[1] < 0 ? 2 * p.dstep[1] : 0),
            mnx = ph.dx0 + (p.dstep[2] < 0 ? 2 * p.dstep[2] : 0);
  float4 pv[NPCH];
  unsigned pok = 0;
  auto decode = [&](unsigned t, int& n, int& oz0, int& oy0, int& ox0) {
    const int tx = t % tg.tiles_x; t /= tg.tiles_x;
    const int ty = t % tg.tiles_y; t /= tg.tiles_y;
    const int tz = t % tg.tiles_z;
    n = (int)(t / tg.tiles_z);
    oz0 = tz * P3_TZ; oy0 = ty * P3_TY; ox0 = tx * P3_TX;
  };
  auto load_patch = [&](unsigned t) {
    int n, oz0, oy0, ox0;
    decode(t, n, oz0, oy0, ox0);
    const int pz0 = oz0 + mnz, py0 = oy0 + mny, px0 = ox0 + mnx;
    const float* __restrict__ gin = p.in + 4 * c4;
    pok = 0;
#pragma unroll
    for (int i = 0; i < NPCH; ++i) {
      const int iz = pz0 + (pzyx[i] >> 16), iy = py0 + ((pzyx[i] >> 8) & 255), ix = px0 + (pzyx[i] & 255);
      const bool ok = pzyx[i] >= 0 && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi &&
                      (unsigned)ix < (unsigned)p.Wi;
      const long off = ok ? ((((long)n * p.Di + iz) * p.Hi + iy) * p.Wi + ix) * p.ldi : 0;   // out of range: a valid address, masked below
      pv[i] = *reinterpret_cast<const float4*>(gin + off);
      pok |= (ok ? 1u : 0u) << i;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPCH; ++i) {
      float4 v = pv[i];
      if constexpr (HAS_PRO) {
        v.x = act_apply(fmaf(v.x, sc.x, sh.x), act, slope);
        v.y = act_apply(fmaf(v.y, sc.y, sh.y), act, slope);
        v.z = act_apply(fmaf(v.z, sc.z, sh.z), act, slope);
        v.w = act_apply(fmaf(v.w, sc.w, sh.w), act, slope);
      }
      const bool ok = (pok >> i) & 1u;                           // the conv's zero padding is a zero of the ACTIVATED tensor
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      if (pzyx[i] >= 0) {
        if constexpr (MM16) {
          const int pr = ((pzyx[i] >> 16) * P3_PY + ((pzyx[i] >> 8) & 255)) * P3M_PXP + (pzyx[i] & 255);
          p3_bf16x4 o;
          o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
          *reinterpret_cast<p3_bf16x4*>(patchb + pr * 32 + 8 * c4) = o;
        } else {
        const int pr = ((pzyx[i] >> 16) * P3_PY + ((pzyx[i] >> 8) & 255)) * P3_PX + (pzyx[i] & 255);
        *reinterpret_cast<float4*>(patch + pr * P3_PA + 4 * c4) = v;
        }
      }
    }
  };
  const unsigned ntiles = (unsigned)(p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x);
  const unsigned per = (ntiles + gridDim.x - 1) / gridDim.x;
  const unsigned w0 = xcd_remap(blockIdx.x, gridDim.x) * per;
  const unsigned wend = w0 + per < ntiles ? w0 + per : ntiles;
  if (w0 < wend) load_patch(w0);                               // in flight while the weights are fetched
  {
    const float* __restrict__ gw = p.wp;                       // packed [Cout = 16][27][Cin = 16]
    float4 wv[NWCH];
#pragma unroll
    for (int i = 0; i < NWCH; ++i) {
      const int e = tid + 256 * i;
      wv[i] = *reinterpret_cast<const float4*>(gw + (long)(e < 27 * 16 * 4 ? e : 0) * 4);   // row * 16 + 4 * k4 == 4 * e
    }
#pragma unroll
    for (int i = 0; i < NWCH; ++i) {
      const int e = tid + 256 * i;
      if (e < 27 * 16 * 4) {
        const int k4 = e & 3, row = e >> 2;                    // row = co * 27 + tap
        const int co = row / 27, tap = row - co * 27;
        if constexpr (MM16) {
          p3_bf16x4 o;
          o[0] = (__bf16)wv[i].x; o[1] = (__bf16)wv[i].y; o[2] = (__bf16)wv[i].z; o[3] = (__bf16)wv[i].w;
          *reinterpret_cast<p3_bf16x4*>(wlb + (tap * 16 + co) * 32 + 8 * k4) = o;
        } else {
          *reinterpret_cast<float4*>(wl + (tap * 16 + co) * P3_PW + 4 * k4) = wv[i];
        }
      }
    }
  }
  for (unsigned tt = w0; tt < wend; ++tt) {
  const int stats_row = (int)tt;
  int n, oz0, oy0, ox0;
  decode(tt, n, oz0, oy0, ox0);
  store_patch();
  __syncthreads();
  if (tt + 1 < wend) load_patch(tt + 1);                       // under this tile's contraction and output stores
  // ---- contraction: wave = 32 pixels (two 16-row blocks) x 16 channels ----
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int ln = lane & 15, g = lane >> 4;
  int abase[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int q = wid * 32 + rb * 16 + ln;
    abase[rb] = (((q >> 6) * P3_PY) + ((q >> 3) & 7)) * P3_PX + (q & 7);
  }
  f32x4 acc[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[rb][i] = 0.f;
  const int fz = p.dstep[0] < 0 ? 2 : 0, fy = p.dstep[1] < 0 ? 2 : 0, fx = p.dstep[2] < 0 ? 2 : 0;
  if constexpr (MM16) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const int kb = (g & 1) * 16;                                 // byte offset of this lane's 8 channels in a 32-byte row
    const bool second = (g >> 1) != 0;                           // ... of the pair's second tap
    int ab[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int q = wid * 32 + rb * 16 + ln;
      ab[rb] = ((((q >> 6) * P3_PY) + ((q >> 3) & 7)) * P3M_PXP + (q & 7)) * 32 + kb;
    }
    const int wb = ln * 32 + kb;
#pragma unroll
    for (int tp = 0; tp < 14; ++tp) {
      int offs[2], taps[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int t = 2 * tp + e < 27 ? 2 * tp + e : 26;         // (slot 27: a valid address, its weights read as zeros)
        const int jz = t / 9, jy = (t / 3) % 3, jx = t % 3;
        offs[e] = (((fz + p.dstep[0] * jz) * P3_PY + (fy + p.dstep[1] * jy)) * P3M_PXP + (fx + p.dstep[2] * jx)) * 32;
        taps[e] = (((ph.kz0 + p.kstep[0] * jz) * 3 + (ph.ky0 + p.kstep[1] * jy)) * 3 + (ph.kx0 + p.kstep[2] * jx)) * 16 * 32;
      }
      const int off = second ? offs[1] : offs[0], tapb = second ? taps[1] : taps[0];
      bf16x8 bf = *reinterpret_cast<const bf16x8*>(wlb + tapb + wb);
      if (second && 2 * tp + 1 >= 27) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[e] = (__bf16)0.f;
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(patchb + ab[rb] + off);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[rb], 0, 0, 0);
      }
    }
  } else
#pragma unroll
  for (int jz = 0; jz < 3; ++jz)
#pragma unroll
    for (int jy = 0; jy < 3; ++jy)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) {
        const int off = ((fz + p.dstep[0] * jz) * P3_PY + (fy + p.dstep[1] * jy)) * P3_PX + (fx + p.dstep[2] * jx);
        const int tap = ((ph.kz0 + p.kstep[0] * jz) * 3 + (ph.ky0 + p.kstep[1] * jy)) * 3 + (ph.kx0 + p.kstep[2] * jx);
        const float4 b = *reinterpret_cast<const float4*>(wl + (tap * 16 + ln) * P3_PW + 4 * g);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const float4 a = *reinterpret_cast<const float4*>(patch + (abase[rb] + off) * P3_PA + 4 * g);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[rb], 0, 0, 0);
        }
      }
  // ---- epilogue: D[row = 4 g + i][col = ln] -> out[pixel][channel], bias / residual, statistics of z = acc + bias ----
  const float bv = p.bias ? p.bias[ln] : 0.f;
  const bool ea = p.epi.scale != nullptr;                      // epilogue activation (eval-mode inference)
  const float esc = ea ? p.epi.scale[ln] : 1.f, esh = ea ? p.epi.shift[ln] : 0.f, esl = ea ? p.epi.slope[ln] : 1.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wid * 32 + rb * 16 + 4 * g + i;
      const int oz = oz0 + (q >> 6), oy = oy0 + ((q >> 3) & 7), ox = ox0 + (q & 7);
      if (oz < ph.Mz && oy < ph.My && ox < ph.Mx) {
        const long pix = (((long)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
        const float z = acc[rb][i] + bv;
        s1 += z;
        s2 = fmaf(z, z, s2);
        float v = z;
        if (ea) v = epi_act1(v, esc, esh, esl);
        if (p.resid) v += p.resid[pix * p.ldr + ln];
        p.out[pix * p.ldo + ln] = v;
      }
    }
  if (p.stats) {
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (g == 0) { st[(wid * 2 + 0) * 16 + ln] = s1; st[(wid * 2 + 1) * 16 + ln] = s2; }
    __syncthreads();
    if (tid < 32) {
      const int sq = tid >> 4, c = tid & 15;
      p.stats[(long)stats_row * 32 + tid] = (st[(0 * 2 + sq) * 16 + c] + st[(1 * 2 + sq) * 16 + c]) +
                                            (st[(2 * 2 + sq) * 16 + c] + st[(3 * 2 + sq) * 16 + c]);
    }
  }
  __syncthreads();                                             // patch and st are rewritten by the next tile
  }
}

// Geometry / launch-mode half of the 3-D patch kernel's predicate: what mpgan_conv_variant and
// mpgan_conv_stats_rows can decide from the conv alone (the caller sizes its partial rows from it).
static bool patch3d_geom_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_PATCH3D") != nullptr;
  if (off || !(p.Cin == 16 && p.Cout == 16 && p.nphase == 1 && p.Kz == 3 && p.Ky == 3 && p.Kx == 3 &&
               !p.tanh_out && !p.fold.acc && !p.stats_acc && !p.bwd.part && !p.in_bf16 && !p.out_bf16 && p.ksplit <= 1 &&
               p.pro.n_stride == 0))
    return false;
  const Phase& ph = p.ph[0];
  if (!(ph.nz == 3 && ph.ny == 3 && ph.nx == 3 && ph.oz == 0 && ph.oy == 0 && ph.ox == 0 && p.Do == ph.Mz && p.Ho == ph.My &&
        p.Wo == ph.Mx && ph.Mz >= 2 && ph.My >= 4 && ph.Mx >= 4))
    return false;
  for (int d = 0; d < 3; ++d)
    if (p.istride[d] != 1 || p.ostride[d] != 1 || (p.dstep[d] != 1 && p.dstep[d] != -1) || (p.kstep[d] != 1)) return false;
  return (long)p.N * p.Di * p.Hi * p.Wi * p.Cin < (1L << 31);
}

// The operand half: pitches, alignment and the 32-bit offset range of the tensors actually passed.
static bool patch3d_operands_ok(const GatherConv& p) {
  return p.ldi % 4 == 0 && ((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.wp)) & 15) == 0 &&
         (!p.pro.scale || ((reinterpret_cast<uintptr_t>(p.pro.scale) | reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0) &&
         (long)p.N * p.Di * p.Hi * p.Wi * p.ldi < (1L << 31);
}

static P3Grid patch3d_grid(const GatherConv& p) {
  const Phase& ph = p.ph[0];
  return P3Grid{(ph.Mz + P3_TZ - 1) / P3_TZ, (ph.My + P3_TY - 1) / P3_TY, (ph.Mx + P3_TX - 1) / P3_TX};
}

static int launch_patch3d(const GatherConv& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_patch3d_c16_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_patch3d_c16_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_patch3d_c16_kernel<true, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P3M_SMEM);
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_patch3d_c16_kernel<false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, P3M_SMEM);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      set_error("gather_patch3d: hipFuncSetAttribute: %s",
                hipGetErrorString(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : e4))));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const P3Grid tg = patch3d_grid(p);
  const long ntiles = (long)p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512));          // two resident blocks per CU, each walking its range of tiles
  static const bool no_mm16 = dev_env("MPGAN_DBG_NO_MM16") != nullptr;
  if (p.mm16 && !no_mm16) {
    if (p.pro.scale) hipLaunchKernelGGL((gather_patch3d_c16_kernel<true, true>), grid, dim3(256), P3M_SMEM, st, p, tg);
    else hipLaunchKernelGGL((gather_patch3d_c16_kernel<false, true>), grid, dim3(256), P3M_SMEM, st, p, tg);
  } else if (p.pro.scale) hipLaunchKernelGGL(gather_patch3d_c16_kernel<true>, grid, dim3(256), P3_SMEM, st, p, tg);
  else hipLaunchKernelGGL(gather_patch3d_c16_kernel<false>, grid, dim3(256), P3_SMEM, st, p, tg);
  return check_launch("gather_patch3d_c16");
}

#ifdef MPGAN_STAMPS
static int launch_gather_impl(const GatherConv& p, hipStream_t st);
static int launch_gather(const GatherConv& p0, hipStream_t st) {
  GatherConv p = p0;
  StampCtx& c = stamp_ctx();
  p.stamps = nullptr;
  p.stamp_blocks = 0;
  if (c.base && c.next < c.launches) {
    p.stamps = c.base + c.next * c.blocks * MPGAN_STAMP_SLOTS;
    p.stamp_blocks = (int)c.blocks;
    c.next += 1;
  }
  return launch_gather_impl(p, st);
}
static int launch_gather_impl(const GatherConv& p, hipStream_t st) {
#else
static int launch_gather(const GatherConv& p, hipStream_t st) {
#endif
  const long maxM = max_phase_pixels(p);
  if (maxM == 0) return MPGAN_OK;
  MPGAN_CHECK_ARG((long)p.N * p.Do * p.Ho * p.Wo < (1L << 31) && (long)p.N * p.Di * p.Hi * p.Wi < (1L << 31) &&
                      maxM < (1L << 31) - 256,
                  "gather_conv: more than 2^31 pixels");
  MPGAN_CHECK_ARG((long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx < (1L << 31), "gather_conv: weight larger than 2^31");
  const int variant = select_variant(p, maxM, thin_cin1_ok(p), thin_cout1_ok(p));
  MPGAN_UNSUPPORTED(p.epi.scale && (p.stats || p.stats_acc || p.bwd.part || p.fold.acc || p.ksplit > 1 || p.in_bf16 || p.out_bf16),
                    "gather_conv: the epilogue activation goes with a plain forward (no fused statistics, split-K or bf16 storage)");
  MPGAN_UNSUPPORTED(p.fold.acc && (p.pro.n_stride != 0 || p.fold.cstride < p.Cin),
                    "gather_conv: fold-on-load is per channel (BatchNorm) over >= Cin accumulator columns");
  if (variant <= 2) {
    MPGAN_UNSUPPORTED(p.fold.acc != nullptr, "thin conv: no fold-on-load");
    MPGAN_UNSUPPORTED(p.bwd.part != nullptr, "thin conv: no fused norm-backward sums (mpgan_conv_bwd_stats_rows() == 0)");
    return launch_thin(p, maxM, st);
  }
  if (patch3d_geom_ok(p)) {
    if (patch3d_operands_ok(p)) return launch_patch3d(p, st);
    // the caller sized its partial rows for one row per 2x8x8 tile (mpgan_conv_stats_rows): the K-stepped fallback
    // would write a different number of them
    MPGAN_UNSUPPORTED(p.stats != nullptr, "gather_conv: fused statistics of a 3-D patch-kernel geometry need 16-byte "
                                          "aligned operands, a channel pitch % 4 == 0 and < 2^31 input elements");
  }
  {
    PatchLaunch pl;
    int smem = 0;
    if (patch_plan(p, &pl, &smem)) {
      MPGAN_UNSUPPORTED(p.bwd.part != nullptr, "patch kernel: no fused norm-backward sums (mpgan_conv_bwd_stats_rows() == 0)");
      const bool aligned = (p.ldi % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0) &&
                           ((reinterpret_cast<uintptr_t>(p.wp) & 15) == 0) &&
                           (!p.pro.scale || ((reinterpret_cast<uintptr_t>(p.pro.scale) |
                                              reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0);
      if (aligned) return launch_patch(p, pl, smem, st);
      MPGAN_UNSUPPORTED(p.stats != nullptr, "gather_conv: fused statistics of a patch-kernel geometry need 16-byte "
                                            "aligned operands (the partial-row count differs otherwise)");
    }
  }
  MPGAN_UNSUPPORTED(p.fold.acc != nullptr, "gather_conv: fold-on-load is served by the persistent patch kernel only");
  const bool vec = (p.Cin % 4 == 0) && (p.ldi % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(p.wp) & 15) == 0) &&
                   (!p.pro.scale || (((reinterpret_cast<uintptr_t>(p.pro.scale) |
                                       reinterpret_cast<uintptr_t>(p.pro.shift)) & 15) == 0 &&
                                     p.pro.n_stride % 4 == 0));
  static const bool no_pipe = dev_env("MPGAN_DBG_NO_PIPE") != nullptr;
  // the pipelined kernel addresses each operand as base + unsigned 32-bit byte offset
  const bool small = (long)p.N * p.Di * p.Hi * p.Wi * p.ldi * 4 < (1L << 32) &&
                     (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 4 < (1L << 32) &&
                     (long)p.N * (p.pro.n_stride > 0 ? p.pro.n_stride : 0) * 4 < (1L << 31);
  if (p.mm16 && vec && small && mm16_gather_ok(p))       // bf16 matrix operands (conv_mm16.hip): same tiles, same rows
    return launch_gather_mm16(p, variant, variant != 128 && pipe_wants_ksplit2(p, variant, maxM), maxM, st);
  if (vec && !no_pipe && small) {
    if (p.Cin % 32 == 0) return launch_pipe_pro<1>(p, variant, maxM, st);
    if (p.Cin == 16) return launch_pipe_pro<2>(p, variant, maxM, st);
  }
  if (vec) {
    if (variant == 128) return launch_variant<128, 2, 2, 2, false>(p, maxM, st);
    if (variant == 64) return launch_variant<64, 1, 2, 1, false>(p, maxM, st);
    return launch_variant<32, 1, 1, 1, false>(p, maxM, st);
  }
  if (variant == 128) return launch_variant<128, 2, 2, 2, true>(p, maxM, st);
  if (variant == 64) return launch_variant<64, 1, 2, 1, true>(p, maxM, st);
  return launch_variant<32, 1, 1, 1, true>(p, maxM, st);
}

}  // namespace mpgan

using namespace mpgan;

extern "C" int32_t mpgan_conv_variant(const mpgan_conv_geom* g, int32_t backward_data, int32_t has_prologue);
extern "C" int32_t mpgan_conv_acc_supported(const mpgan_conv_geom* g, int32_t has_prologue);

static void build_for_forward(GatherConv& p, const mpgan_conv_geom* g) {
  set_geom_flags(p, g);
  if (!g->transposed)
    build_forward(p, g->n, g->in_dhw, g->cin, g->out_dhw, g->cout, g->k, g->stride, g->pad);
  else
    build_transposed(p, g->n, g->in_dhw, g->cin, g->out_dhw, g->cout, g->k, g->stride, g->pad);
}

// out[pix][co] = bias[co] + sum_s partial[s][pix][co]   (fixed order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long rows,
                                                            int Cout, const float* __restrict__ bias,
                                                            float* __restrict__ out, int ldo) {
  const long total = rows * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / Cout;
    const int co = (int)(i - pix * Cout);
    float s = bias ? bias[co] : 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[(long)k * total + i];
    out[pix * ldo + co] = s;
  }
}

// Split-K plan for a forward conv whose output grid alone cannot fill the chip
// (the variant-B head Linear(512*8^3 -> 64): 7 pixel tiles, 8192 K-steps).
static int plan_ksplit(const GatherConv& p) {
  if (!(p.Cin % 32 == 0) || p.Cout == 1 || p.nphase != 1) return 1;
  if (patch_plan(p, nullptr, nullptr)) return 1;
  const long maxM = max_phase_pixels(p);
  const int variant = select_variant(p, maxM, false, false);
  const long blocks = (maxM + BM - 1) / BM * ((p.Cout + variant - 1) / variant) * p.nphase;
  if (blocks >= 128) return 1;
  const Phase& ph = p.ph[0];
  const long nk = ((long)ph.nz * ph.ny * ph.nx * p.Cin + BK - 1) / BK;
  long want = (512 + blocks - 1) / blocks;
  if (want > nk / 8) want = nk / 8;            // keep >= 8 K-steps per slice
  if (want > 256) want = 256;
  return want < 2 ? 1 : (int)want;
}

extern "C" int64_t mpgan_conv_splitk_workspace(const mpgan_conv_geom* g) {
  if (check_geom(g)) return -1;
  GatherConv p{};
  build_for_forward(p, g);
  const int ks = plan_ksplit(p);
  if (ks <= 1) return 0;
  return (int64_t)ks * g->n * g->out_dhw[0] * g->out_dhw[1] * g->out_dhw[2] * g->cout * (int64_t)sizeof(float);
}

extern "C" int32_t mpgan_conv_stats_rows(const mpgan_conv_geom* g, int32_t has_prologue) {
  const int v = mpgan_conv_variant(g, 0, has_prologue);
  if (v == 1 && (g->cout == 16 || g->cout == 32)) {   // all-channel 1 -> C stencil: one partial row per 256-pixel block
    GatherConv p1{};
    build_for_forward(p1, g);
    return (int32_t)((max_phase_pixels(p1) + 255) / 256) * p1.nphase;
  }
  if (v == 2) {           // quad kernel of ConvTranspose2d(C -> 1, k3 s2): one partial row per block
    static const float dummy16[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    GatherConv p2{};
    build_for_forward(p2, g);
    p2.in = dummy16;
    p2.ldi = g->cin;
    p2.pro = make_pro(nullptr);
    if (!convt_quad_ok(p2)) return 0;
    return (int32_t)(((long)p2.N * p2.Hi * p2.Wi * (p2.Cin / 4) + 255) / 256);
  }
  if (v < 16) return 0;   // other thin VALU kernels (or invalid geometry): no fused statistics
  GatherConv p{};
  build_for_forward(p, g);
  if (v == 16 || v == 17) {   // patch kernel: one partial row per (tile, phase)
    PatchLaunch pl;
    patch_plan(p, &pl, nullptr);
    return (int32_t)(pl.tiles_x * pl.tiles_y * p.N * p.nphase);
  }
  if (v == 18) {              // 3-D patch kernel: one partial row per 2 x 8 x 8 tile
    const P3Grid tg = patch3d_grid(p);
    return (int32_t)(p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x);
  }
  return (int32_t)phase_tile_rows(p, BM);
}

extern "C" int mpgan_conv_forward_fold(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                                       const float* bias, const mpgan_prologue* pro, const mpgan_norm_fold* fold,
                                       const float* resid, int32_t ldr, int32_t tanh_out, float* stats_partials,
                                       int64_t* stats_acc, int32_t acc_replicas, float* y, int32_t ldy, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(x && w_packed && y, "conv_forward: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && ldy >= g->cout && (!resid || ldr >= g->cout), "conv_forward: bad pitch");
  MPGAN_CHECK_ARG(!(stats_partials && stats_acc), "conv_forward: partial rows OR accumulators, not both");
  GatherConv p{};
  p.in = x; p.wp = w_packed; p.out = y; p.bias = bias; p.resid = resid;
  p.pro = make_pro(pro);
  p.fold = make_fold(fold);
  if (p.fold.acc) {
    MPGAN_CHECK_ARG(pro != nullptr && fold->scale && fold->shift && fold->mean && fold->invstd && fold->replicas > 0 &&
                        fold->count > 0,
                    "conv_forward: fold-on-load needs a prologue (activation) and output vectors");
    p.pro.scale = fold->scale;      // marks "has a per-channel prologue" for dispatch; the kernel folds instead of reading it
    p.pro.shift = fold->shift;
    p.pro.n_stride = 0;
  }
  p.ldi = ldx; p.ldo = ldy; p.ldr = ldr; p.tanh_out = tanh_out;
  build_for_forward(p, g);
  if (stats_partials || stats_acc) {
    MPGAN_CHECK_ARG(!resid && !tanh_out, "conv_forward: fused statistics describe the raw conv output (no resid/tanh)");
    const int code = p.pro.scale ? (p.pro.n_stride ? 2 : 1) : 0;
    if (stats_partials)
      MPGAN_UNSUPPORTED(mpgan_conv_stats_rows(g, code) == 0,
                        "conv_forward: this geometry runs on a thin kernel without fused statistics "
                        "(mpgan_conv_stats_rows() == 0): use mpgan_channel_stats");
    else
      MPGAN_UNSUPPORTED(acc_replicas <= 0 || !mpgan_conv_acc_supported(g, code),
                        "conv_forward: no accumulator statistics for this geometry (mpgan_conv_acc_supported() == 0)");
    p.stats = stats_partials;
    p.stats_acc = reinterpret_cast<long long*>(stats_acc);
    p.acc_rep = acc_replicas;
  }
  return launch_gather(p, (hipStream_t)stream);
}

// Forward conv of eval-mode inference: y = prelu(conv(x) * scale[c] + shift[c], slope[c]) (+ resid) (tanh) -- the layer's
// running-statistics BatchNorm, its PReLU and the conv's own bias (folded into `shift`) applied in the epilogue, so the
// ACTIVATED tensor is what reaches HBM and no norm_act_add launch follows (EpiAct, conv_geom.h).
extern "C" int mpgan_conv_forward_act(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                                      const float* scale, const float* shift, const float* slope, const float* resid,
                                      int32_t ldr, int32_t tanh_out, float* y, int32_t ldy, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(x && w_packed && y && scale && shift && slope, "conv_forward_act: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && ldy >= g->cout && (!resid || ldr >= g->cout), "conv_forward_act: bad pitch");
  MPGAN_CHECK_ARG(((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(slope)) & 15) == 0,
                  "conv_forward_act: scale / shift / slope must be 16-byte aligned");
  GatherConv p{};
  p.in = x; p.wp = w_packed; p.out = y; p.bias = nullptr; p.resid = resid;
  p.pro = make_pro(nullptr);
  p.fold = make_fold(nullptr);
  p.epi.scale = scale; p.epi.shift = shift; p.epi.slope = slope;
  p.ldi = ldx; p.ldo = ldy; p.ldr = ldr; p.tanh_out = tanh_out;
  build_for_forward(p, g);
  return launch_gather(p, (hipStream_t)stream);
}

extern "C" int mpgan_conv_forward(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                                  const float* bias, const mpgan_prologue* pro, const float* resid, int32_t ldr,
                                  int32_t tanh_out, float* stats_partials, float* y, int32_t ldy, void* stream) {
  return mpgan_conv_forward_fold(g, x, ldx, w_packed, bias, pro, nullptr, resid, ldr, tanh_out, stats_partials, nullptr,
                                 0, y, ldy, stream);
}

// Which kernel would serve this forward conv: shared by the two queries below (geometry only, aligned operands).
static int forward_kernel_class(const mpgan_conv_geom* g, int32_t has_prologue, bool* persist) {
  *persist = false;
  const int v = mpgan_conv_variant(g, 0, has_prologue);
  if (v == 16 || v == 17) {
    GatherConv p{};
    static const float dummy[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    p.in = dummy;
    p.pro = make_pro(nullptr);
    if (has_prologue) p.pro.scale = dummy;
    build_for_forward(p, g);
    p.ldi = g->cin;
    if (has_prologue == 2) p.pro.n_stride = g->cin;
    PatchLaunch pl, pp;
    int smem = 0, psmem = 0, ntiles = 0, grid = 0;
    if (patch_plan(p, &pl, &smem)) *persist = patch_persist_plan(p, pl, &pp, &psmem, &ntiles, &grid);
  }
  return v;
}

extern "C" int32_t mpgan_conv_acc_supported(const mpgan_conv_geom* g, int32_t has_prologue) {
  bool persist;
  const int v = forward_kernel_class(g, has_prologue, &persist);
  if (v < 0) return 0;
  if (v == 16 || v == 17) return persist ? 1 : 0;
  if (v == 1) return (g->cout == 16 || g->cout == 32) ? 1 : 0;
  if (v == 2) return mpgan_conv_stats_rows(g, has_prologue) > 0 ? 1 : 0;      // the quad kernel of ConvTranspose2d(C -> 1)
  return 1;                                                                  // K-stepped kernels share conv_epilogue
}

extern "C" int32_t mpgan_conv_fold_supported(const mpgan_conv_geom* g) {
  bool persist;
  const int v = forward_kernel_class(g, 1, &persist);
  return v == 16 && persist ? 1 : 0;
}

// y = conv(prologue(x)) + bias with the K axis split over blocks (small output grids).
extern "C" int mpgan_conv_forward_splitk(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                                         const float* bias, const mpgan_prologue* pro, void* workspace,
                                         int64_t workspace_bytes, float* y, int32_t ldy, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(x && w_packed && y, "conv_forward_splitk: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && ldy >= g->cout, "conv_forward_splitk: bad pitch");
  GatherConv p{};
  p.in = x; p.wp = w_packed; p.out = y; p.bias = bias;
  p.pro = make_pro(pro);
  p.ldi = ldx; p.ldo = ldy;
  build_for_forward(p, g);
  const int ks = plan_ksplit(p);
  if (ks <= 1) return launch_gather(p, (hipStream_t)stream);
  const long rows = (long)g->n * g->out_dhw[0] * g->out_dhw[1] * g->out_dhw[2];
  const int64_t need = (int64_t)ks * rows * g->cout * (int64_t)sizeof(float);
  MPGAN_CHECK_ARG(workspace && workspace_bytes >= need, "conv_forward_splitk: workspace %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
  MPGAN_UNSUPPORTED((p.ldi % 4) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w_packed) & 15),
                    "conv_forward_splitk: operands must be 16-byte vectorisable");
  p.ksplit = ks;
  p.kpartial = static_cast<float*>(workspace);
  p.bias = nullptr;
  rc = launch_gather(p, (hipStream_t)stream);
  if (rc) return rc;
  long blocks = (rows * g->cout + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p.kpartial, ks,
                     rows, g->cout, bias, y, ldy);
  return check_launch("splitk_reduce");
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
  set_geom_flags(p, g);
  if (!g->transposed)
    build_transposed(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  else
    build_forward(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  return launch_gather(p, (hipStream_t)stream);
}

// Partial rows mpgan_conv_backward_data_stats leaves for this geometry; 0 = that launch is served by a thin or
// patch kernel, which has no fused sums (run mpgan_norm_bwd_reduce instead).
extern "C" int32_t mpgan_conv_bwd_stats_rows(const mpgan_conv_geom* g) {
  const int v = mpgan_conv_variant(g, 1, 0);
  if (v < 32 || v == 1128) return 0;
  GatherConv p{};
  set_geom_flags(p, g);
  if (!g->transposed) build_transposed(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  else build_forward(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  return (int32_t)phase_tile_rows(p, BM);
}

// mpgan_conv_backward_data + the reduce pass of the norm layer in front of this conv's input, in one launch:
// dx is the gradient w.r.t. a = act(scale * z + shift); partials[rows][3][Cin] receive the sums mpgan_norm_bwd_reduce
// would form from dx and z (BatchNorm: scale / shift / mean / invstd hold Cin values; any number of rows feeds
// mpgan_norm_bwd_finalize with n = 1, chunks = rows).
extern "C" int mpgan_conv_backward_data_stats(const mpgan_conv_geom* g, const float* dy, int32_t lddy,
                                              const float* w_packed_bwd, float* dx, int32_t lddx, const float* z,
                                              int32_t ldz, const float* scale, const float* shift, const float* mean,
                                              const float* invstd, int32_t act, float slope, float* partials,
                                              void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(dy && w_packed_bwd && dx && z && scale && shift && mean && invstd && partials,
                  "conv_backward_data_stats: null pointer");
  MPGAN_CHECK_ARG(lddy >= g->cout && lddx >= g->cin && ldz >= g->cin, "conv_backward_data_stats: bad pitch");
  MPGAN_UNSUPPORTED(mpgan_conv_bwd_stats_rows(g) == 0, "conv_backward_data_stats: this geometry is served by a kernel "
                                                      "without fused sums (mpgan_conv_bwd_stats_rows() == 0)");
  GatherConv p{};
  p.in = dy; p.wp = w_packed_bwd; p.out = dx; p.bias = nullptr; p.resid = nullptr;
  p.pro = make_pro(nullptr);
  p.ldi = lddy; p.ldo = lddx; p.ldr = 0; p.tanh_out = 0;
  p.bwd.z = z; p.bwd.ldz = ldz; p.bwd.scale = scale; p.bwd.shift = shift; p.bwd.mean = mean; p.bwd.invstd = invstd;
  p.bwd.leaky = act == MPGAN_ACT_LEAKY; p.bwd.slope = slope; p.bwd.part = partials;
  set_geom_flags(p, g);
  if (!g->transposed)
    build_transposed(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  else
    build_forward(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  return launch_gather(p, (hipStream_t)stream);
}

extern "C" int32_t mpgan_conv_variant(const mpgan_conv_geom* g, int32_t backward_data, int32_t has_prologue) {
  if (check_geom(g)) return -1;
  GatherConv p{};
  static const float dummy[4] = {0, 0, 0, 0};
  p.in = dummy;  // 16-byte aligned stand-ins: only geometry drives the choice
  p.pro = make_pro(nullptr);
  if (has_prologue) p.pro.scale = dummy;
  set_geom_flags(p, g);
  const int cg = backward_data ? g->cout : g->cin, cp = backward_data ? g->cin : g->cout;
  const int32_t* gd = backward_data ? g->out_dhw : g->in_dhw;
  const int32_t* pd = backward_data ? g->in_dhw : g->out_dhw;
  if (backward_data) {
    if (!g->transposed) build_transposed(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad);
    else build_forward(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad);
  } else {
    if (!g->transposed) build_forward(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad);
    else build_transposed(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad);
  }
  p.ldi = cg;
  if (has_prologue == 2) p.pro.n_stride = cg;   // per-(sample, channel) scale/shift (InstanceNorm)
  const int T = p.Kz * p.Ky * p.Kx;
  const bool t1 = p.Cin == 1 && !p.pro.scale && (long)T * ((p.Cout + 3) / 4 * 4) * 4 <= 48 * 1024;
  const int lanes = p.Cin / 4;
  const bool t2 = p.Cout == 1 && !p.pro.scale && p.Cin % 4 == 0 && lanes >= 1 && lanes <= 64 &&
                  (lanes & (lanes - 1)) == 0 && (long)T * p.Cin * 4 <= 48 * 1024;
  const int v = select_variant(p, max_phase_pixels(p), t1, t2);
  if (v > 2 && patch3d_geom_ok(p)) return 18;                                     // 3-D patch kernel, 16 -> 16 channels
  if (v > 2) {
    PatchLaunch pl;
    if (patch_plan(p, &pl, nullptr)) return pl.merged ? 17 : 16;
  }
  if (v == 128 && has_prologue == 3 && fast_geometry(p, 128)) return 1128;
  if ((v == 32 || v == 64) && p.Cin % 32 == 0 && has_prologue != 3 && pipe_wants_ksplit2(p, v, max_phase_pixels(p)))
    return 2000 + v;
  if (!has_prologue && dma_form_ok(p, v, max_phase_pixels(p))) return 3000 + v;   // gather_conv_dma_kernel
  return v;
}


// ---- thin layers of the bf16 path (D.conv1: 1 -> 64): fp32 image in, bf16 activations out, and back ----
extern "C" int mpgan_conv_forward_f32_to_bf16(const mpgan_conv_geom* g, const float* x, int32_t ldx,
                                              const float* w_packed, const float* bias, float* stats_partials,
                                              void* y, int32_t ldy, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(x && w_packed && y, "conv_forward_f32_to_bf16: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && ldy >= g->cout, "conv_forward_f32_to_bf16: bad pitch");
  MPGAN_UNSUPPORTED(g->cin != 1 || g->transposed, "conv_forward_f32_to_bf16: ConvNd with one input channel only");
  GatherConv p{};
  p.in = x; p.wp = w_packed; p.out = static_cast<float*>(y); p.bias = bias;
  p.pro = make_pro(nullptr);
  p.ldi = ldx; p.ldo = ldy;
  p.out_bf16 = 1;
  p.stats = stats_partials;
  build_forward(p, g->n, g->in_dhw, g->cin, g->out_dhw, g->cout, g->k, g->stride, g->pad);
  const long maxM = max_phase_pixels(p);
  MPGAN_CHECK_ARG(maxM < (1L << 31) - 256, "conv_forward_f32_to_bf16: more than 2^31 pixels");
  return launch_thin(p, maxM, (hipStream_t)stream);
}

extern "C" int mpgan_conv_backward_data_bf16_to_f32(const mpgan_conv_geom* g, const void* dy, int32_t lddy,
                                                    const float* w_packed_bwd, float* dx, int32_t lddx, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(dy && w_packed_bwd && dx, "conv_backward_data_bf16_to_f32: null pointer");
  MPGAN_CHECK_ARG(lddy >= g->cout && lddx >= g->cin, "conv_backward_data_bf16_to_f32: bad pitch");
  MPGAN_UNSUPPORTED(g->cin != 1 || g->transposed, "conv_backward_data_bf16_to_f32: ConvNd with one input channel only");
  GatherConv p{};
  p.in = static_cast<const float*>(dy); p.wp = w_packed_bwd; p.out = dx;
  p.pro = make_pro(nullptr);
  p.ldi = lddy; p.ldo = lddx;
  p.in_bf16 = 1;
  build_transposed(p, g->n, g->out_dhw, g->cout, g->in_dhw, g->cin, g->k, g->stride, g->pad);
  const long maxM = max_phase_pixels(p);
  MPGAN_CHECK_ARG(maxM < (1L << 31) - 256, "conv_backward_data_bf16_to_f32: more than 2^31 pixels");
  return launch_thin(p, maxM, (hipStream_t)stream);
}
