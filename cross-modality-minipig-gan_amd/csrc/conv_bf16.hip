// bf16-storage implicit-GEMM convolution for gfx950 (config C5: the reference's 3-D graph, GAN_final.py:167-189).
//
// Activations and packed weights live in HBM as bf16; products accumulate in fp32 on
// v_mfma_f32_32x32x16_bf16; BatchNorm statistics are taken from the fp32 accumulators; master weights,
// gradients of weights and Adam stay fp32 (the fp32 kernels' buffers).
//
// At 16x the fp32 matrix rate a K-step has no room for a normalise-on-load prologue (32 elements per
// thread and K-step = ~150 vector instructions against 512 matrix cycles), so in bf16 mode the producer's
// BatchNorm + LeakyReLU is materialised once per layer by an HBM-bound pass (norm_act_bf16_kernel: 0.65 ms
// for D.conv2's 2 GB output against ~20 ms of matrix work that consumes it) and BOTH operands of the
// contraction go global -> LDS by LDS-DMA (global_load_lds_dwordx4), never through registers:
//   * LDS image: rows of 64 K-elements (128 B), lane-linear per wave instruction, swizzled on the SOURCE
//     address (chunk ^ ((row>>1)&7)) so that the fragments' ds_read_b128 are conflict-free;
//   * out-of-range taps of a backward-data / padded gather read a 256-byte zero page instead of being masked;
//   * three LDS stages, counted vmcnt, one raw s_barrier per K-step: a tile's loads have two K-steps to land.
// Tile: 256 pixels x BN channels x 64 K, 256 threads = 4 waves, one block per CU (144 KiB of LDS).
#include "mpgan_common.h"
#include "conv_geom.h"
#include "lds_dma.h"
#include <stdlib.h>
#include <type_traits>

namespace mpgan {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero_page[16];   // 256 B of zeros: what a masked LDS-DMA gather reads
__device__ int g_hb_dbg;            // MPGAN_DBG_HB what-if bits (development): 1 = gathers confined to a 64 KiB window,
                                    // 2 = no MFMAs, 4 = no fragment reads either, 8 = weights confined to 16 KiB,
                                    // 16 = no LDS-DMA (the contraction runs on whatever the LDS holds), 32 = no epilogue
// The what-if tests exist in `make DEV=1 WHATIF=1` builds only: in the product build (and in plain DEV=1 builds, which
// only add the routing switches) HB_DBG is the constant 0, the
// tests fold away and the K loops are single basic blocks again (a runtime test around the fragment reads and another
// around the MFMAs had split every k-sub into three blocks, which kept hipcc from placing the next k-sub's address
// arithmetic under the MFMAs).
#ifdef MPGAN_HB_WHATIF
#define HB_DBG g_hb_dbg
#else
#define HB_DBG 0
#endif

// ---------------------------------------------------------------------------
// Epilogue of the eight-wave kernels straight from the accumulators (round 3; the first version wrote the whole fp32
// tile to LDS, summed its columns there and read it a third time for the stores: 0.85 of D.conv2's 4.43 ms forward).
//   * statistics: a lane holds column li of rows (r&3) + 8(r>>2) + 4 lh of each 32-row block -- it sums its own
//     values over the rows that own an output pixel, the two half-waves are folded with one shuffle, the WMW waves
//     that share the columns through part[WMW][2][BNT];
//   * stores: each 32 x (32 TN) block of the wave goes through a wave-private slab (no block barrier) and leaves as
//     16-byte bf16 stores, 64 TN contiguous bytes per row.
// rowpix[row] = output pixel of tile row `row` or -1; rowbase = the wave's first tile row, colw = its first tile
// column, cbase = the produced channel of that column.  Block barriers inside: call from uniform control flow.
// ---------------------------------------------------------------------------
template <int TN> struct HbSlab {
  static constexpr int PITCH = TN * 32 + 8;               // floats; rows r and r+4 land half a bank sweep apart
  static constexpr int WAVE = 32 * PITCH * 4;
};
// bw (BwdStats with part != null; backward-data launches): the launch also leaves the norm-backward sums of the gradient
// it produces against z, the raw output of the layer in front (bf16, BatchNorm + LeakyReLU(slope)) -- per tile and
// column sum(gy) and sum(gy * zhat), gy = g * act'(y), from the gradient AS STORED (rounded to bf16) -- as one row
// [3][Cout] of bw.part (the third part, a PReLU slope's gradient, is zero here): what norm_bwd_reduce_bf16 would re-read
// g and z for.  z is fetched 16 bytes per lane beside the store of the same 8 channels.  pair_cols: the tile's columns
// c and c + BNT / 2 are the same produced channel (two phases of a pair share the tile).
template <int TM, int TN, int WMW, int BNT>
__device__ __forceinline__ void hb_epilogue(const f32x16 (&acc)[TM][TN], float* slab, const int* rowpix, float* part,
                                            int rowbase, int colw, int cbase, int wmw, const float* bias, float* stats_row,
                                            int n0, int Cout, char* goutb, int ldo, int tid, int lane,
                                            const BwdStats* bw = nullptr, float* bwd_row = nullptr, bool pair_cols = false) {
  constexpr int PITCH = HbSlab<TN>::PITCH;
  const int li = lane & 31, lh = lane >> 5;
  float bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = cbase + tn * 32 + li;
    bv[tn] = (bias && col < Cout) ? bias[col] : 0.f;
  }
  if (stats_row) {
    float sm[TN], sq[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) sm[tn] = sq[tn] = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 rp = *reinterpret_cast<const int4*>(rowpix + rowbase + tm * 32 + 8 * g + 4 * lh);
        const int ok[4] = {rp.x >= 0, rp.y >= 0, rp.z >= 0, rp.w >= 0};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const float v = ok[e] ? acc[tm][tn][4 * g + e] + bv[tn] : 0.f;
            sm[tn] += v;
            sq[tn] = fmaf(v, v, sq[tn]);
          }
      }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      sm[tn] += __shfl_xor(sm[tn], 32, 64);
      sq[tn] += __shfl_xor(sq[tn], 32, 64);
      if (lh == 0) {
        part[(wmw * 2 + 0) * BNT + colw + tn * 32 + li] = sm[tn];
        part[(wmw * 2 + 1) * BNT + colw + tn * 32 + li] = sq[tn];
      }
    }
    __syncthreads();
    if (tid < BNT && n0 + tid < Cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WMW; ++w) {
        a += part[(w * 2 + 0) * BNT + tid];
        b += part[(w * 2 + 1) * BNT + tid];
      }
      stats_row[n0 + tid] = a;
      stats_row[Cout + n0 + tid] = b;
    }
  }
  constexpr int CH = TN * 4;                              // 16-byte output chunks per row
  // norm-backward sums: this lane stores chunk lane % CH of every row it handles, i.e. always the same 8 channels
  const bool bws = bw != nullptr && bwd_row != nullptr;
  const int bch = cbase + (lane % CH) * 8;
  float bsc[8], bsh[8], bmu[8], bis[8], bs1[8], bs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool okc = bws && bch + e < Cout;
    bsc[e] = okc ? bw->scale[bch + e] : 0.f;
    bsh[e] = okc ? bw->shift[bch + e] : 0.f;
    bmu[e] = okc ? bw->mean[bch + e] : 0.f;
    bis[e] = okc ? bw->invstd[bch + e] : 0.f;
    bs1[e] = bs2[e] = 0.f;
  }
  const float bslope = bws ? bw->slope : 0.f;
  const __bf16* bz = bws ? reinterpret_cast<const __bf16*>(bw->z) : nullptr;
  const int bldz = bws ? bw->ldz : 0;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + tn * 32 + li] = acc[tm][tn][r] + bv[tn];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 32 * CH / 64; ++j) {
      const int idx = lane + 64 * j, row = idx / CH, ch = idx % CH;
      const int pix = rowpix[rowbase + tm * 32 + row];
      const float4 v0 = *reinterpret_cast<const float4*>(slab + row * PITCH + ch * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(slab + row * PITCH + ch * 8 + 4);
      if (pix < 0 || cbase + ch * 8 >= Cout) continue;
      bf16x8 o;
      o[0] = (__bf16)v0.x; o[1] = (__bf16)v0.y; o[2] = (__bf16)v0.z; o[3] = (__bf16)v0.w;
      o[4] = (__bf16)v1.x; o[5] = (__bf16)v1.y; o[6] = (__bf16)v1.z; o[7] = (__bf16)v1.w;
      *reinterpret_cast<bf16x8*>(goutb + ((long)pix * ldo + cbase + ch * 8) * 2) = o;
      if (bws) {
        const bf16x8 zz = *reinterpret_cast<const bf16x8*>(bz + (long)pix * bldz + cbase + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float zf = (float)zz[e], g = (float)o[e];
          const float y = zf * bsc[e] + bsh[e];
          const float gy = y < 0.f ? g * bslope : g;
          bs1[e] += gy;
          bs2[e] = fmaf(gy, (zf - bmu[e]) * bis[e], bs2[e]);
        }
      }
    }
  }
  if (bws) {
    // lanes that share a chunk (lane % CH), then the WMW waves that share the columns (fixed order: no atomics)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int off = CH; off < 64; off <<= 1) {
        bs1[e] += __shfl_xor(bs1[e], off, 64);
        bs2[e] += __shfl_xor(bs2[e], off, 64);
      }
    }
    __syncthreads();                                      // (`part` may still hold the forward statistics' partials)
    if (lane < CH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        part[(wmw * 2 + 0) * BNT + colw + lane * 8 + e] = bs1[e];
        part[(wmw * 2 + 1) * BNT + colw + lane * 8 + e] = bs2[e];
      }
    }
    __syncthreads();
    const int ncol = pair_cols ? BNT / 2 : BNT;
    if (tid < ncol && n0 + tid < Cout) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < WMW; ++w) {
        a1 += part[(w * 2 + 0) * BNT + tid];
        a2 += part[(w * 2 + 1) * BNT + tid];
        if (pair_cols) {
          a1 += part[(w * 2 + 0) * BNT + tid + BNT / 2];
          a2 += part[(w * 2 + 1) * BNT + tid + BNT / 2];
        }
      }
      bwd_row[n0 + tid] = a1;
      bwd_row[Cout + n0 + tid] = a2;
      bwd_row[2 * Cout + n0 + tid] = 0.f;
    }
  }
}

constexpr int HB_BM = 256;          // pixels per tile
constexpr int HB_BK = 64;           // K elements per step (one 128-byte LDS row)
constexpr int HB_ROWB = 128;

//   NW: waves per block.  4 = one wave per SIMD (128 x 64 or 64 x 64 per wave); 8 = two waves per SIMD (64 x 64 or
//   64 x 32 per wave): one wave's vmcnt / barrier / fragment waits sit under the other's MFMAs.
template <int BN, int NW = 4>
struct HbTile {
  static constexpr int NT = NW * 64;
  static constexpr int WM = NW == 8 ? 4 : (BN == 128 ? 2 : 4), WN = NW == 8 ? 2 : (BN == 128 ? 2 : 1);
  static constexpr int TM = HB_BM / WM / 32, TN = BN / WN / 32;
  static constexpr int PASS_ROWS = NT / 8;                            // rows one LDS-DMA instruction of every wave fills
  static constexpr int APIECES = HB_BM / PASS_ROWS, BPIECES = BN / PASS_ROWS;
  static constexpr int STAGE = (HB_BM + BN) * HB_ROWB;
  static constexpr int NSTAGE = 3;
  static constexpr int IMG_PITCH = BN + 4;                            // floats
  static constexpr int ROWPIX = HB_BM * IMG_PITCH * 4;                // int[256] behind the fp32 epilogue image
  static constexpr int SMEM_LOOP = NSTAGE * STAGE;
  static constexpr int SMEM_EPI = ROWPIX + HB_BM * 4 + 4096;                  // + the column-sum partials
  static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
};

//   MASK : taps can fall outside the gathered tensor (backward-data gathers, padded convs): per-row tap bitmasks,
//          invalid pieces read the zero page.  !MASK: every tap of every pixel is in range (valid convs).
template <int BN, bool MASK, int NW>
__global__ __launch_bounds__(NW * 64, 1) void gather_conv_bf16_kernel(const GatherConv p) {
  using T = HbTile<BN, NW>;
  constexpr int TM = T::TM, TN = T::TN, WN = T::WN, NT = T::NT, PR = T::PASS_ROWS;
  constexpr int NLOADS = T::APIECES + T::BPIECES;     // LDS-DMA instructions per thread and K-step
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const BlockId bid = conv_block_id(p);
  const Phase& ph = p.ph[bid.phase];
  const long Mtot = (long)p.N * ph.Mz * ph.My * ph.Mx;
  const long m0 = (long)bid.mt * HB_BM;
  const int n0 = bid.nt * BN;
  const int stats_row = bid.row;
  const int Cout = p.Cout;
  if (m0 >= Mtot) {                                   // empty tile of a short phase (block-uniform)
    if (p.stats && tid < BN && n0 + tid < Cout) {
      float* row = p.stats + (long)stats_row * 2 * Cout;
      row[n0 + tid] = 0.f;
      row[Cout + n0 + tid] = 0.f;
    }
    return;
  }
  const int Cin = p.Cin, Di = p.Di, Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  const int ntaps = ph.nz * ph.ny * ph.nx;
  const int nchunk = Cin / HB_BK;
  const int nk = ntaps * nchunk;
  const char* __restrict__ ginb = reinterpret_cast<const char*>(p.in);
  const char* __restrict__ gwb = reinterpret_cast<const char*>(p.wp);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // ---- this thread's pieces: rows r0 + 32*i, 16-byte chunk `ck` of the K-step (source-side swizzle) ----
  const int r0 = tid >> 3, cc = tid & 7;
  const int ck = cc ^ ((r0 >> 1) & 7);
  unsigned rbB[T::APIECES];            // byte offset of (row pixel, channel 0)
  unsigned tmask[MASK ? T::APIECES : 1];
#pragma unroll
  for (int i = 0; i < T::APIECES; ++i) {
    unsigned m = (unsigned)m0 + r0 + PR * i;
    const bool live = m < (unsigned)Mtot;
    m = live ? m : (unsigned)Mtot - 1u;               // clamped rows gather a real pixel; never stored
    unsigned q, umx, umy, umz;
    fdivmod(m, ph.fMx, q, umx);
    fdivmod(q, ph.fMy, q, umy);
    fdivmod(q, ph.fMz, q, umz);
    const int bz = (int)umz * p.istride[0], by = (int)umy * p.istride[1], bx = (int)umx * p.istride[2];
    rbB[i] = (unsigned)((((int)q * Di + bz) * Hi + by) * Wi + bx) * (unsigned)ldi * 2u;
    if (HB_DBG & 1) rbB[i] &= 0xFF80u;
    if constexpr (MASK) {
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
      tmask[i] = live ? mk : 0u;
    }
  }
  const unsigned Ktot2 = (unsigned)(p.Kz * p.Ky * p.Kx * Cin) * 2u;
  unsigned wrowB[T::BPIECES];
#pragma unroll
  for (int i = 0; i < T::BPIECES; ++i) {
    int co = n0 + r0 + PR * i;
    co = co < Cout ? co : Cout - 1;                   // clamped columns are computed and dropped
    wrowB[i] = (unsigned)co * Ktot2;
    if (HB_DBG & 8) wrowB[i] &= 0x3F80u;
  }
  const int dbg = HB_DBG;

  // K order: channel-chunk major, taps inner (the taps that re-read an input element are then adjacent K-steps).
  // The tap walk is wave-uniform and lives in scalar registers: the K loop makes NO LDS access besides the
  // fragment reads (hipcc drains every LDS-DMA in flight, vmcnt(0), in front of an LDS read it cannot tell
  // apart from the staging area -- a table in LDS would serialise the pipeline).
  int itap = 0, ichunk = 0, istage = 0;               // cursor of the NEXT tile to issue
  int jx = 0, jy = 0, jz = 0;
  const int dsz = p.dstep[0], dsy = p.dstep[1], dsx = p.dstep[2], ksz = p.kstep[0], ksy = p.kstep[1], ksx = p.kstep[2];
  auto issue = [&]() {
    const int dz = ph.dz0 + dsz * jz, dy = ph.dy0 + dsy * jy, dx = ph.dx0 + dsx * jx;
    const int kz = ph.kz0 + ksz * jz, ky = ph.ky0 + ksy * jy, kx = ph.kx0 + ksx * jx;
    unsigned deltaB = (unsigned)(((dz * Hi + dy) * Wi + dx) * ldi * 2);
    unsigned woffB = (unsigned)(((kz * p.Ky + ky) * p.Kx + kx) * Cin * 2);
    if (dbg & 1) deltaB = 0;
    if (dbg & 8) woffB = 0;
    const unsigned ciB = (unsigned)(ichunk * HB_BK + ck * 8) * 2u;
    char* As = lds + istage * T::STAGE + (8 * wid) * HB_ROWB;
    char* Bs = As + HB_BM * HB_ROWB;
    if (!(dbg & 16)) {
#pragma unroll
    for (int i = 0; i < T::APIECES; ++i) {
      const char* src = ginb + (rbB[i] + deltaB + ciB);
      if constexpr (MASK) src = ((tmask[i] >> itap) & 1u) ? src : zero;
      GLDS16(src, As + PR * i * HB_ROWB);
    }
#pragma unroll
    for (int i = 0; i < T::BPIECES; ++i) GLDS16(gwb + (wrowB[i] + woffB + ciB), Bs + PR * i * HB_ROWB);
    }
    itap += 1;
    jx += 1;
    if (jx == ph.nx) { jx = 0; jy += 1; }
    if (jy == ph.ny) { jy = 0; jz += 1; }
    if (itap == ntaps) { itap = 0; jz = 0; ichunk += 1; }
    istage = istage == T::NSTAGE - 1 ? 0 : istage + 1;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addresses: lane (li, lh) of k-sub s takes chunk 2s+lh of its row, stored at chunk ^ ((row>>1)&7)
  const int sw = (li >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = ((2 * s + lh) ^ sw) * 16;
  const int arow = (wm * (HB_BM / T::WM) + li) * HB_ROWB;
  const int brow = HB_BM * HB_ROWB + (wn * (BN / WN) + li) * HB_ROWB;

  const unsigned lds_base = lds_addr(lds);
  // Pipeline: tile kt is consumed from stage kt % 3 while tiles kt+1 and kt+2 are in flight.  The K-step's ONE
  // barrier sits in front of its LAST k-sub's MFMAs: by then this wave has read all of stage kt into registers,
  // so behind the barrier (tile kt+1 landed for every wave, stage kt free for every wave) the first fragments of
  // tile kt+1 are read and tile kt+3's DMAs are issued UNDER those MFMAs -- no bubble at the K-step boundary.
  i32x4 fa[2][TM], fb[2][TN];
  auto read_frags = [&](int stage, int s, int set) {
    const unsigned As = lds_base + stage * T::STAGE + arow;
    const unsigned Bs = lds_base + stage * T::STAGE + brow;
    lds_read_b128_n<TM, 32 * HB_ROWB>(fa[set], As + foff[s]);
    lds_read_b128_n<TN, 32 * HB_ROWB>(fb[set], Bs + foff[s]);
  };
  if (nk > 0) issue();
  if (nk > 1) issue();
  if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");             // tile 0 landed
  if (nk > 0 && !(dbg & 4)) read_frags(0, 0, 0);
  if (nk > 2) issue();
  int cstage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int nstage = cstage == T::NSTAGE - 1 ? 0 : cstage + 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int set = s & 1;
      lds_wait<TM, TN>(fa[set], fb[set]);
      if (s < 3) {
        if (!(dbg & 4)) read_frags(cstage, s + 1, set ^ 1);
      } else if (kt + 1 < nk) {
        // tile kt+1 has landed once all but the newest NLOADS DMAs of this wave are done (tile kt+2's)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) ; tail: the last tile" ::: "memory");      // (tools/check_isa.py)
        asm volatile("s_barrier" ::: "memory");
        if (!(dbg & 4)) read_frags(nstage, 0, 0);
        if (kt + 3 < nk) issue();
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 2))
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][tm]),
                                                                __builtin_bit_cast(bf16x8, fb[set][tn]), acc[tm][tn], 0, 0, 0);
    }
    cstage = nstage;
  }
  asm volatile("s_barrier" ::: "memory");             // all fragment reads done: LDS becomes the epilogue image
  if (dbg & 32) return;                               // (what-if: no epilogue)

  // ---- epilogue: fp32 image [pixel][channel] in LDS -> statistics, bf16 rows stored 16 bytes per lane ----
  float* img = reinterpret_cast<float*>(lds);
  int* rowpix = reinterpret_cast<int*>(lds + T::ROWPIX);
  if (tid < HB_BM) {
    const unsigned m = (unsigned)m0 + tid;            // one thread per row
    int pix = -1;
    if (m < (unsigned)Mtot) {
      unsigned q, umx, umy, umz;
      fdivmod(m, ph.fMx, q, umx);
      fdivmod(q, ph.fMy, q, umy);
      fdivmod(q, ph.fMz, q, umz);
      const int oz = (int)umz * p.ostride[0] + ph.oz, oy = (int)umy * p.ostride[1] + ph.oy,
                ox = (int)umx * p.ostride[2] + ph.ox;
      if (oz < p.Do && oy < p.Ho && ox < p.Wo) pix = (((int)q * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    }
    rowpix[tid] = pix;
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = (wn * TN + tn) * 32 + li;
    const float bv = (p.bias && n0 + col < Cout) ? p.bias[n0 + col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (HB_BM / T::WM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        img[row * T::IMG_PITCH + col] = acc[tm][tn][r] + bv;
      }
  }
  __syncthreads();
  if (p.stats) {
    // column sums of z = acc + bias over the rows that own an output pixel: thread (column, row half)
    float* part = reinterpret_cast<float*>(lds + T::ROWPIX + HB_BM * 4);   // [2 halves][2][BN]
    constexpr int HALVES = NT / BN;                   // row groups: 2 or 4 (four waves), 4 or 8 (eight)
    const int c = tid % BN, hf = tid / BN;
    float sm = 0.f, sq = 0.f;
    for (int row = hf * (HB_BM / HALVES); row < (hf + 1) * (HB_BM / HALVES); ++row) {
      const float v = rowpix[row] >= 0 ? img[row * T::IMG_PITCH + c] : 0.f;
      sm += v;
      sq = fmaf(v, v, sq);
    }
    if (hf > 0) {
      part[((hf - 1) * 2 + 0) * BN + c] = sm;
      part[((hf - 1) * 2 + 1) * BN + c] = sq;
    }
    __syncthreads();
    if (hf == 0 && n0 + c < Cout) {
#pragma unroll
      for (int h = 1; h < HALVES; ++h) {
        sm += part[((h - 1) * 2 + 0) * BN + c];
        sq += part[((h - 1) * 2 + 1) * BN + c];
      }
      float* row = p.stats + (long)stats_row * 2 * Cout;
      row[n0 + c] = sm;
      row[Cout + n0 + c] = sq;
    }
  }
  char* goutb = reinterpret_cast<char*>(p.out);
  constexpr int CHUNKS = BN / 8;                      // 16-byte chunks per row
  for (int q = tid; q < HB_BM * CHUNKS; q += NT) {
    const int row = q / CHUNKS, ch = q % CHUNKS;
    const int pix = rowpix[row];
    if (pix < 0 || n0 + ch * 8 >= Cout) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(img + row * T::IMG_PITCH + ch * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(img + row * T::IMG_PITCH + ch * 8 + 4);
    bf16x8 o;
    o[0] = (__bf16)v0.x; o[1] = (__bf16)v0.y; o[2] = (__bf16)v0.z; o[3] = (__bf16)v0.w;
    o[4] = (__bf16)v1.x; o[5] = (__bf16)v1.y; o[6] = (__bf16)v1.z; o[7] = (__bf16)v1.w;
    *reinterpret_cast<bf16x8*>(goutb + ((long)pix * p.ldo + n0 + ch * 8) * 2) = o;
  }
}

template <int BN, bool MASK, int NW>
static int hb_launch(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_bf16_kernel<BN, MASK, NW>;
  constexpr int smem = HbTile<BN, NW>::SMEM;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_conv_bf16: hipFuncSetAttribute(%d): %s", smem, hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  static int dbg_set = -1;
  if (dbg_set < 0) {
    const char* e = dev_env("MPGAN_DBG_HB");
    dbg_set = e ? atoi(e) : 0;
    if (dbg_set) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hb_dbg), &dbg_set, sizeof(int));
  }
  GatherConv q = p;
  const long pairs = set_tile_grid(q, HB_BM);
  q.ntiles = (p.Cout + BN - 1) / BN;
  q.phase_outer = (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 2 > (3L << 20) ? 1 : 0;
  q.ksplit = 1;
  dim3 grid((unsigned)(pairs * q.ntiles));
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, q);
  return check_launch("gather_conv_bf16");
}

// every tap of every pixel of every phase inside the gathered tensor?
static bool hb_all_in_range(const GatherConv& p) {
  const int G[3] = {p.Di, p.Hi, p.Wi};
  for (int i = 0; i < p.nphase; ++i) {
    const Phase& ph = p.ph[i];
    if (ph.nz * ph.ny * ph.nx == 0) return false;
    const int d0[3] = {ph.dz0, ph.dy0, ph.dx0}, nj[3] = {ph.nz, ph.ny, ph.nx}, M[3] = {ph.Mz, ph.My, ph.Mx};
    for (int d = 0; d < 3; ++d) {
      if (M[d] == 0) continue;
      const int e = d0[d] + p.dstep[d] * (nj[d] - 1);
      const int lo = d0[d] < e ? d0[d] : e, hi = d0[d] < e ? e : d0[d];
      if (lo < 0 || (M[d] - 1) * p.istride[d] + hi > G[d] - 1) return false;
    }
  }
  return true;
}

static int hb_check(const GatherConv& p, const char* what) {
  MPGAN_UNSUPPORTED(p.Cin % HB_BK != 0, "%s: bf16 path needs gathered channels %% 64 == 0 (got %d)", what, p.Cin);
  MPGAN_UNSUPPORTED(p.Cout % 8 != 0 || p.ldo % 8 != 0 || p.ldi % 8 != 0, "%s: bf16 path needs channels / pitches %% 8 == 0", what);
  MPGAN_UNSUPPORTED((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.wp) | reinterpret_cast<uintptr_t>(p.out)) & 15,
                    "%s: bf16 operands must be 16-byte aligned", what);
  MPGAN_CHECK_ARG((long)p.N * p.Do * p.Ho * p.Wo < (1L << 31) && (long)p.N * p.Di * p.Hi * p.Wi < (1L << 31), "%s: more than 2^31 pixels", what);
  MPGAN_UNSUPPORTED((long)p.N * p.Di * p.Hi * p.Wi * p.ldi * 2 >= (1L << 32) || (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 2 >= (1L << 32),
                    "%s: bf16 operand of 4 GiB or more (32-bit byte offsets)", what);
  if (!hb_all_in_range(p))           // masked gathers keep one validity bit per tap and row
    for (int i = 0; i < p.nphase; ++i)
      MPGAN_UNSUPPORTED(p.ph[i].nz * p.ph[i].ny * p.ph[i].nx > 32, "%s: more than 32 taps per phase of a masked gather", what);
  return MPGAN_OK;
}

// ---------------------------------------------------------------------------
// Wide form of the K-stepped kernel (round 3): 128 x 64 per wave instead of 64 x 64.
// What bounds the 256 x 128 tile above is the L2 -> LDS fill (66 GB/s per CU when the pieces are whole
// 128-byte lines served by the XCD's L2) at 87 FLOP per staged byte, plus 3.4-6.1 vector instructions per MFMA for the
// pieces' 64-bit source addresses and zero-page selects (profiles/r03_pmc_bf16_sq_counters.csv: matrix pipe busy
// 34-43 %).  Here:
//   * eight waves as WM x WN = 2 x 4 (256 x 256 tile, layers with >= 256 produced channels: 128 FLOP per staged byte)
//     or 4 x 2 (512 x 128: 102), each wave 128 x 64: six ds_read_b128 per eight MFMAs instead of four per four;
//   * K-steps of 64 like above (a piece must be a whole 128-byte line: a first version with 64-byte rows and three
//     stages of K = 32 filled at 39 GB/s per CU, every line fetched twice as two half-lines a K-step apart), so only
//     TWO stages fit (128 / 160 KiB): tile kt+2 is issued into the stage tile kt leaves, behind the barrier in front of
//     kt's last k-sub, and has one K-step (2,048 matrix cycles) to land;
//   * LDS-DMA through BUFFER loads (buffer_load_dwordx4 ... offen lds): the per-thread part of a piece's address is a
//     32-bit voffset computed once, the tap / channel-chunk walk is the scalar soffset -- no vector instruction per
//     piece in the pad-free form; a masked piece adds its tap offset and selects an out-of-range voffset (the
//     hardware's bounds check returns zeros: no zero page, no 64-bit select);
//   * epilogue without the fp32 tile image: statistics from the accumulators (lane = column), 32 x 64 blocks of a
//     wave transposed through a wave-private LDS slab into 16-byte bf16 stores (128 contiguous bytes per row).
// ---------------------------------------------------------------------------
constexpr int HW_BK = 64;
constexpr int HW_ROWB = 128;

template <int WM, int WN, bool RING_ = true>
struct HwTile {
  static constexpr int BM = WM * 128, BN = WN * 64, NT = 512;
  static constexpr int PR = NT / 8;                                   // rows one LDS-DMA instruction of every wave fills
  static constexpr int APIECES = BM / PR, BPIECES = BN / PR;
  static constexpr int STAGE = (BM + BN) * HW_ROWB;
  static constexpr bool RING = RING_ && BM == BN;                     // five-unit ring (see the kernel) or two stages
  static constexpr int UNIT = BM * HW_ROWB;
  static constexpr int SMEM_LOOP = RING ? 5 * UNIT : 2 * STAGE;
  static constexpr int EP_WAVE = HbSlab<2>::WAVE;
  static constexpr int EP_ROWPIX = 8 * EP_WAVE;
  static constexpr int EP_PART = EP_ROWPIX + 2 * BM * 4;              // (two row -> pixel tables in the pair form)
  static constexpr int SMEM_EPI = EP_PART + WM * 2 * BN * 4;
  static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
  static_assert(WM * WN == 8 && BM % PR == 0 && BN % PR == 0, "eight waves; whole DMA passes");
};


//   PAIR (256 x 256 only): the tile's columns are the 128 produced channels of TWO phases of a strided backward-data
//   gather whose phases read the same gathered pixels through different kernel taps (k = 4, stride 2: all eight).
//   The gathered tile is staged once for both -- 128 FLOP per staged byte instead of the 102 of a 512 x 128 tile per
//   phase -- and each half of the columns has its own weight-row base and its own output pixel per row.
template <int WM, int WN, bool MASK, bool RING = true, bool PAIR = false>
__global__ __launch_bounds__(512, 1) void gather_conv_bf16_wide_kernel(const GatherConv p) {
  static_assert(!PAIR || (WM == 2 && WN == 4), "phase pairs: 256 x 256 tiles");
  using T = HwTile<WM, WN, RING>;
  constexpr int BM = T::BM, BN = T::BN, PR = T::PR, TM = 4, TN = 2;
  constexpr int NLOADS = T::APIECES + T::BPIECES;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const BlockId bid = conv_block_id(p);               // (PAIR: p.nphase counts pairs, bid.phase is the pair)
  const Phase& ph = p.ph[PAIR ? 2 * bid.phase : bid.phase];
  const long Mtot = (long)p.N * ph.Mz * ph.My * ph.Mx;
  const long m0 = (long)bid.mt * BM;
  const int n0 = PAIR ? 0 : bid.nt * BN;
  const int stats_row = bid.row;
  const int Cout = p.Cout;
  if (m0 >= Mtot) {                                   // empty tile of a short phase (block-uniform)
    if (!PAIR && p.stats && tid < BN && n0 + tid < Cout) {
      float* row = p.stats + (long)stats_row * 2 * Cout;
      row[n0 + tid] = 0.f;
      row[Cout + n0 + tid] = 0.f;
    }
    return;
  }
  const int Cin = p.Cin, Di = p.Di, Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  const int ntaps = ph.nz * ph.ny * ph.nx;
  const int nchunk = Cin / HW_BK;
  const int nk = ntaps * nchunk;
  const unsigned bytesA = (unsigned)((((long)p.N * Di * Hi * Wi - 1) * ldi + Cin) * 2);
  const unsigned Ktot2 = (unsigned)(p.Kz * p.Ky * p.Kx * Cin) * 2u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, (unsigned)Cout * Ktot2, 0x00020000);
  auto tap0 = [&](const Phase& q) { return ((q.kz0 * p.Ky + q.ky0) * p.Kx + q.kx0) * Cin * 2; };

  // ---- this thread's pieces: rows r0 + 64*i, 16-byte chunk `ck` of the K-step (source-side swizzle) ----
  const int r0 = tid >> 3, cc = tid & 7;
  const int ck = cc ^ ((r0 >> 1) & 7);
  unsigned voffA[T::APIECES];           // byte offset of (row pixel, channel 8*ck)
  unsigned tmask[MASK ? T::APIECES : 1];
#pragma unroll
  for (int i = 0; i < T::APIECES; ++i) {
    unsigned m = (unsigned)m0 + r0 + PR * i;
    const bool live = m < (unsigned)Mtot;
    m = live ? m : (unsigned)Mtot - 1u;               // clamped rows gather a real pixel; never stored
    unsigned q, umx, umy, umz;
    fdivmod(m, ph.fMx, q, umx);
    fdivmod(q, ph.fMy, q, umy);
    fdivmod(q, ph.fMz, q, umz);
    const int bz = (int)umz * p.istride[0], by = (int)umy * p.istride[1], bx = (int)umx * p.istride[2];
    voffA[i] = (unsigned)((((int)q * Di + bz) * Hi + by) * Wi + bx) * (unsigned)ldi * 2u + (unsigned)ck * 16u;
    if constexpr (MASK) {
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
      tmask[i] = live ? mk : 0u;
    }
  }
  unsigned voffB[T::BPIECES];
#pragma unroll
  for (int i = 0; i < T::BPIECES; ++i) {
    if constexpr (PAIR) {                             // columns 0-127: phase 2q, 128-255: phase 2q+1 (Cout == 128)
      const int row = r0 + PR * i;
      voffB[i] = (unsigned)(row & 127) * Ktot2 + (unsigned)ck * 16u + (unsigned)tap0(p.ph[2 * bid.phase + (row >> 7)]);
    } else {
      int co = n0 + r0 + PR * i;
      co = co < Cout ? co : Cout - 1;                 // clamped columns are computed and dropped
      voffB[i] = (unsigned)co * Ktot2 + (unsigned)ck * 16u;
    }
  }
  const int dbg = HB_DBG;

  // K order: channel-chunk major, taps inner.  The walk is wave-uniform and INCREMENTAL: the gathered operand's byte
  // offset and the weight offset of the next tap are the previous ones plus one of three precomputed steps (x, x-wrap,
  // y-wrap) -- a handful of scalar registers instead of the phase's whole tap geometry (recomputing the offsets from
  // (jz, jy, jx) had the compiler re-load the geometry from the kernel arguments inside the loop, behind an
  // lgkmcnt(0) that also waits for the fragment reads just issued).
  const int aX = p.dstep[2] * ldi * 2;
  const int aY = p.dstep[1] * Wi * ldi * 2 - (ph.nx - 1) * aX;
  const int aZ = p.dstep[0] * Hi * Wi * ldi * 2 - (ph.ny - 1) * p.dstep[1] * Wi * ldi * 2 - (ph.nx - 1) * aX;
  const int wX = p.kstep[2] * Cin * 2;
  const int wY = p.kstep[1] * p.Kx * Cin * 2 - (ph.nx - 1) * wX;
  const int wZ = p.kstep[0] * p.Ky * p.Kx * Cin * 2 - (ph.ny - 1) * p.kstep[1] * p.Kx * Cin * 2 - (ph.nx - 1) * wX;
  const int delta0 = ((ph.dz0 * Hi + ph.dy0) * Wi + ph.dx0) * ldi * 2;   // >= 0 when every tap is in range (!MASK)
  const int woff0 = PAIR ? 0 : tap0(ph);              // (PAIR: each row's first tap is in its voffset)
  const int nx = ph.nx, ny = ph.ny;
  const int wbase = __builtin_amdgcn_readfirstlane(8 * wid * HW_ROWB);
  // two cursors: the gathered operand's tiles (A) run one K-step ahead of the weights' (B) in the ring form
  int itapA = 0, jxA = 0, jyA = 0, deltaB = delta0, ciA = 0;
  int itapB = 0, jxB = 0, jyB = 0, woffB = woff0, ciB = 0;
  auto issueA = [&](int unit, auto part) {             // part: 0 = all pieces, 1 / 2 = first / second half of them
    constexpr int P = decltype(part)::value;
    constexpr int I0 = P == 2 ? T::APIECES / 2 : 0, I1 = P == 1 ? T::APIECES / 2 : T::APIECES;
    char* As = lds + unit + wbase;
    if (!(dbg & 16)) {
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        if constexpr (MASK) {
          const unsigned v = ((tmask[i] >> itapA) & 1u) ? voffA[i] + (unsigned)deltaB : HW_OOB;
          BLDS16(rsA, As + PR * i * HW_ROWB, v, ciA);
        } else {
          BLDS16(rsA, As + PR * i * HW_ROWB, voffA[i], deltaB + ciA);
        }
      }
    }
    if constexpr (P != 1) {                             // the cursor moves on behind a tile's last piece
      itapA += 1;
      jxA += 1;
      int inc = aX;
      if (jxA == nx) {
        jxA = 0;
        jyA += 1;
        inc = aY;
        if (jyA == ny) { jyA = 0; inc = aZ; }
      }
      deltaB += inc;
      if (itapA == ntaps) { itapA = 0; jxA = 0; jyA = 0; deltaB = delta0; ciA += HW_BK * 2; }
    }
  };
  auto issueB = [&](int unit, auto part) {
    constexpr int P = decltype(part)::value;
    constexpr int I0 = P == 2 ? T::BPIECES / 2 : 0, I1 = P == 1 ? T::BPIECES / 2 : T::BPIECES;
    char* Bs = lds + unit + wbase;
    if (!(dbg & 16)) {
#pragma unroll
      for (int i = I0; i < I1; ++i) BLDS16(rsB, Bs + PR * i * HW_ROWB, voffB[i], woffB + ciB);
    }
    if constexpr (P != 1) {
      itapB += 1;
      jxB += 1;
      int inc = wX;
      if (jxB == nx) {
        jxB = 0;
        jyB += 1;
        inc = wY;
        if (jyB == ny) { jyB = 0; inc = wZ; }
      }
      woffB += inc;
      if (itapB == ntaps) { itapB = 0; jxB = 0; jyB = 0; woffB = woff0; ciB += HW_BK * 2; }
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

  // fragment addresses: lane (li, lh) of k-sub s takes chunk 2s+lh of its row, stored at chunk ^ ((row>>1)&7)
  const int sw = (li >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = ((2 * s + lh) ^ sw) * 16;
  const int arow = (wm * 128 + li) * HW_ROWB;
  const int brow = (wn * 64 + li) * HW_ROWB;
  const unsigned lds_base = lds_addr(lds);
  i32x4 fa[2][TM], fb[2][TN];
  auto read_frags = [&](int unitA, int unitB, int s, int set) {
    const unsigned As = lds_base + unitA + arow + foff[s];
    const unsigned Bs = lds_base + unitB + brow + foff[s];
    lds_read_b128_n<TM, 32 * HW_ROWB>(fa[set], As);
    lds_read_b128_n<TN, 32 * HW_ROWB>(fb[set], Bs);
  };
  auto mfmas = [&](int set) {
    if (!(dbg & 2))
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][tm]),
                                                                __builtin_bit_cast(bf16x8, fb[set][tn]), acc[tm][tn], 0, 0, 0);
  };
  if constexpr (T::RING) {
    // 256 x 256: five units of 32 KiB (the whole LDS) in a ring, A_0 B_0 A_1 B_1 A_2 | B_2 A_3 B_3 ... : unit u lives in
    // slot u % 5.  A K-step's barrier (in front of its last k-sub's MFMAs, as above) frees BOTH of its units; behind
    // it the weights of step kt+2 go into the slot A_kt leaves and the gathered tile of step kt+3 into B_kt's.  The
    // gathered operand -- the one that misses the L2 -- so has two K-steps to land and the L2-resident weights one,
    // 1.5 tiles are in flight instead of the single one two whole stages allow (which left the fill engine idle for
    // a load latency per K-step: fill alone 2.12 ms, contraction alone 2.41, together 3.61 on D.conv3's forward),
    // and every piece is still a whole 128-byte line.  vmcnt: in front of step kt+1 everything but the newest
    // gathered tile (A_kt+2) must have landed.
    constexpr int U = T::UNIT, AP = T::APIECES, BP = T::BPIECES;
    issueA(0, ALL);
    issueB(U, ALL);
    if (nk > 1) { issueA(2 * U, ALL); issueB(3 * U, ALL); }
    if (nk > 2) issueA(4 * U, ALL);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AP + BP) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");             // A_0, B_0 landed
    int ua = 0, ub = U;                                 // byte offsets of the units of step kt
    if (!(dbg & 4)) read_frags(ua, ub, 0, 0);
    // The pieces of a tile are not issued in one burst behind the barrier (an LDS-DMA instruction costs its wave
    // 60-185 issue cycles, and both waves of a SIMD pass the barrier together): two pieces per k-sub -- the weights of
    // step kt+2 behind the barrier and under the next k-sub, the gathered tile of step kt+3 under the two after that.
    bool pend_b = false, pend_a = false;
    int unit_b = 0, unit_a = 0;
    for (int kt = 0; kt < nk; ++kt) {
      int na = ua + 2 * U, nb = ub + 2 * U;
      na = na >= 5 * U ? na - 5 * U : na;
      nb = nb >= 5 * U ? nb - 5 * U : nb;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int set = s & 1;
        lds_wait<TM, TN>(fa[set], fb[set]);
        if (s < 3) {
          if (!(dbg & 4)) read_frags(ua, ub, s + 1, set ^ 1);
          if (s == 0 && pend_b) issueB(unit_b, HALF1);
          if (s == 1 && pend_a) issueA(unit_a, HALF0);
          if (s == 2 && pend_a) issueA(unit_a, HALF1);
        } else if (kt + 1 < nk) {
          if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) ; tail: the last tile" ::: "memory");      // (tools/check_isa.py)
          asm volatile("s_barrier" ::: "memory");
          pend_b = kt + 2 < nk;
          pend_a = kt + 3 < nk;
          unit_b = ua;
          unit_a = ub;
          if (pend_b) issueB(unit_b, HALF0);
          if (!(dbg & 4)) read_frags(na, nb, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set);
      }
      ua = na;
      ub = nb;
    }
  } else {
    // Two stages: tile kt is consumed from stage kt & 1 while tile kt+1 is in flight into the other.  The K-step's one
    // barrier sits in front of its LAST k-sub's MFMAs: this wave has then read all of stage kt & 1 into registers, so
    // behind the barrier (tile kt+1 landed for every wave, the stage free for every wave) tile kt+2's DMAs are issued
    // into it and the first fragments of tile kt+1 are read, all UNDER those MFMAs.
    constexpr int SA = BM * HW_ROWB;                    // a stage: BM rows of A, then BN rows of B
    issueA(0, ALL);
    issueB(SA, ALL);
    if (nk > 1) { issueA(T::STAGE, ALL); issueB(T::STAGE + SA, ALL); }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");             // tile 0 landed
    if (!(dbg & 4)) read_frags(0, SA, 0, 0);
    int cst = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int nst = cst ? 0 : T::STAGE;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int set = s & 1;
        lds_wait<TM, TN>(fa[set], fb[set]);
        if (s < 3) {
          if (!(dbg & 4)) read_frags(cst, cst + SA, s + 1, set ^ 1);
        } else if (kt + 1 < nk) {
          // the only DMAs of this wave still in flight are tile kt+1's
          asm volatile("s_waitcnt vmcnt(0) ; tail: two stages, the one tile in flight" ::: "memory");   // (tools/check_isa.py)
          asm volatile("s_barrier" ::: "memory");
          if (kt + 2 < nk) { issueA(cst, ALL); issueB(cst + SA, ALL); }
          if (!(dbg & 4)) read_frags(nst, nst + SA, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set);
      }
      cst = nst;
    }
  }
  asm volatile("s_barrier" ::: "memory");             // all fragment reads done: the LDS becomes the epilogue's
  if (dbg & 32) return;                               // (what-if: no epilogue)

  // ---- epilogue ----
  int* rowpix = reinterpret_cast<int*>(lds + T::EP_ROWPIX);          // [PAIR ? 2 : 1][BM]
  if (tid < (PAIR ? 2 : 1) * BM) {
    const int half = tid / BM, row = tid - half * BM;
    const Phase& po = p.ph[PAIR ? 2 * bid.phase + half : bid.phase];
    const unsigned m = (unsigned)m0 + row;            // one thread per row (and phase of the pair)
    int pix = -1;
    if (m < (unsigned)Mtot) {
      unsigned q, umx, umy, umz;
      fdivmod(m, po.fMx, q, umx);
      fdivmod(q, po.fMy, q, umy);
      fdivmod(q, po.fMz, q, umz);
      const int oz = (int)umz * p.ostride[0] + po.oz, oy = (int)umy * p.ostride[1] + po.oy,
                ox = (int)umx * p.ostride[2] + po.ox;
      if (oz < p.Do && oy < p.Ho && ox < p.Wo) pix = (((int)q * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    }
    rowpix[tid] = pix;
  }
  const int cbase = PAIR ? (wn & 1) * 64 : n0 + wn * 64;             // first produced channel of this wave's columns
  const BwdStats bwl = p.bwd;                                        // (a local copy: no pointer into the kernel arguments)
  __syncthreads();
  hb_epilogue<TM, TN, WM, BN>(acc, reinterpret_cast<float*>(lds + wid * T::EP_WAVE),
                              rowpix + (PAIR ? (wn >> 1) * BM : 0), reinterpret_cast<float*>(lds + T::EP_PART), wm * 128,
                              wn * 64, cbase, wm, p.bias, (!PAIR && p.stats) ? p.stats + (long)stats_row * 2 * Cout : nullptr,
                              n0, Cout, reinterpret_cast<char*>(p.out), p.ldo, tid, lane, &bwl,
                              bwl.part ? bwl.part + (long)stats_row * 3 * Cout : nullptr, PAIR);
}

// Which K-stepped form serves a gather: 0 = 256 x 128/64 tile (gather_conv_bf16_kernel), 1 = 256 x 256, 2 = 512 x 128,
// 3 = 256 x 256 over the phase pairs of a strided backward-data gather (gather_conv_bf16_wide_kernel).  The wide forms need enough tiles to fill the chip twice over; MPGAN_DBG_HB_WIDE=0
// turns them off (A/B runs).  mpgan_conv_stats_rows_bf16 follows the same choice (rows = phases x m-tiles).
// (the threshold is the geometry's own `min_blocks`, include/mpgan_hip.h: sizing queries and launches see the same value)
static bool hw_pairs_congruent(const GatherConv& p) {
  if (p.nphase < 2 || p.nphase % 2 || p.classes) return false;
  for (int i = 0; i < p.nphase; i += 2) {
    const Phase &a = p.ph[i], &b = p.ph[i + 1];
    if (a.Mz != b.Mz || a.My != b.My || a.Mx != b.Mx || a.nz != b.nz || a.ny != b.ny || a.nx != b.nx || a.dz0 != b.dz0 ||
        a.dy0 != b.dy0 || a.dx0 != b.dx0 || a.nz * a.ny * a.nx == 0)
      return false;
  }
  return true;
}
static int hw_choice(const GatherConv& p, bool with_stats) {
  static int forced = -2;
  if (forced == -2) {
    const char* e = dev_env("MPGAN_DBG_HB_WIDE");
    forced = e ? atoi(e) : -1;
  }
  if (forced == 0) return 0;
  if (p.Cin % HW_BK != 0 || p.Cout % 8 != 0) return 0;
  if ((long)p.N * p.Di * p.Hi * p.Wi * p.ldi * 2 >= (long)HW_OOB || (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 2 >= (long)HW_OOB)
    return 0;
  const long maxM = max_phase_pixels(p);
  const int g_hw_min_blocks = p.min_blocks > 0 ? p.min_blocks : FORM_MIN_BLOCKS_DEFAULT;
  if (p.Cout > 128) {
    const long blocks = phase_tile_rows(p, 256) * ((p.Cout + 255) / 256);
    return (blocks >= g_hw_min_blocks || forced == 1) ? 1 : 0;
  }
  if (p.Cout == 128 && !with_stats && !p.bias && hw_pairs_congruent(p) && forced != 2) {   // 3 = 256 x 256 over phase pairs
    const long blocks = ((maxM + 255) / 256) * (p.nphase / 2);
    if (blocks >= g_hw_min_blocks || forced == 3) return 3;
  }
  if (p.Cout > 64) {
    const long blocks = phase_tile_rows(p, 512);
    return (blocks >= g_hw_min_blocks || forced == 2) ? 2 : 0;
  }
  return 0;
}
static int hw_bm(int choice) { return choice == 2 ? 512 : 256; }

template <int WM, int WN, bool MASK, bool RING = true, bool PAIR = false>
static int hw_launch(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_bf16_wide_kernel<WM, WN, MASK, RING, PAIR>;
  using T = HwTile<WM, WN, RING>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM);
    if (e != hipSuccess) {
      set_error("gather_conv_bf16_wide: hipFuncSetAttribute(%d): %s", T::SMEM, hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  static int dbg_set = -1;
  if (dbg_set < 0) {
    const char* e = dev_env("MPGAN_DBG_HB");
    dbg_set = e ? atoi(e) : 0;
    if (dbg_set) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hb_dbg), &dbg_set, sizeof(int));
  }
  GatherConv q = p;
  long pairs;
  if constexpr (PAIR) {                                // (congruent phases, never border classes: hw_pairs_congruent)
    q.packed = 0;
    q.mtiles = (int)((maxM + T::BM - 1) / T::BM);
    q.nphase = p.nphase / 2;                           // (the kernel reads phases 2 i and 2 i + 1 of pair i)
    pairs = (long)q.mtiles * q.nphase;
  } else {
    pairs = set_tile_grid(q, T::BM);
  }
  q.ntiles = PAIR ? 1 : (p.Cout + T::BN - 1) / T::BN;
  q.phase_outer = (long)p.Cout * p.Cin * p.Kz * p.Ky * p.Kx * 2 > (3L << 20) ? 1 : 0;
  q.ksplit = 1;
  dim3 grid((unsigned)(pairs * q.ntiles));
  hipLaunchKernelGGL(kern, grid, dim3(512), T::SMEM, st, q);
  return check_launch("gather_conv_bf16_wide");
}

// ---------------------------------------------------------------------------
// Patch form for stride-1 3x3x3 gathers (D.conv2 forward and backward-data at config C5: 64 -> 128 and 128 -> 64
// channels, 27 taps).  The K-stepped kernel above re-stages every input pixel once per tap (27 x) and is bound by
// the L2 -> LDS fill rate.  Here a block owns a 4 x 8 x 8 block of output pixels: the 6 x 10 x 10
// input patch of one 64-channel chunk (75 KiB) is staged ONCE and all 27 taps read their A fragments from it at a
// wave-uniform row offset; only the weights stream through the three-stage ring (BN rows x 128 B per tap).  Staged
// bytes per FLOP drop 2.5x (forward) / 3.4x (backward-data).  Out-of-range patch rows (image borders, the
// backward-data gather's halo) read the zero page, so no tap masks exist.  Eight waves, 64 x 64 (BN = 128) or
// 64 x 32 (BN = 64) per wave; epilogue, statistics rows (one per tile) and swizzles as above.
// ---------------------------------------------------------------------------
constexpr int HP_TZ = 4, HP_TY = 8, HP_TX = 8;                 // output pixels per tile: 256
constexpr int HP_PY = HP_TY + 2, HP_PX = HP_TX + 2;
// LDS rows of the patch: row(z, y, x) = z * HP_PZ + y * HP_PX + x with the plane pitch padded from 100 to 104 rows
// (= 8 mod 16).  A ds_read_b128 is served in four groups of 16 lanes, conflict-free when the 16 rows of a group are
// distinct mod 16 (row parity = which half of the 64 banks, (row >> 1) & 7 = the XOR swizzle of the 16-byte chunk).
// With tile rows in q = (z, y, x) order a group held four x-quads of four different y lines, 10 rows apart: up to
// three lanes on one bank quarter (SQ_LDS_BANK_CONFLICT: 30 % of the conv2 backward-data launch's cycles, round 3).
// Now a wave's 32 rows are 2 z x 2 y x 8 x and a lane group = all 8 x of both z planes at one y: rows b + x + 8 z.
constexpr int HP_PZ = 104;
constexpr int HP_PROWS = (HP_TZ + 2) * HP_PZ;                  // 624 patch rows of 128 B (24 of them padding)
constexpr int HP_PATCH = HP_PROWS * HB_ROWB;                   // 79,872 B
constexpr int HP_PPIECES = (HP_PROWS * 8 + 511) / 512;         // LDS-DMA instructions per thread for one patch: 10

// Tile row (MFMA row li of wave sub-tile s = wm * TM + tm) -> tile pixel (z, y, x), see above.  The lane groups of
// ds_read_b128 are {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+32 for the upper half-wave).
__device__ __forceinline__ void hp_row_pixel(int s, int li, int& z, int& y, int& x) {
  const bool g1 = (li >= 4 && li <= 11) || (li >= 16 && li <= 19) || li >= 28;
  const int k = g1 ? (li < 12 ? li - 4 : (li < 20 ? li - 8 : li - 16)) : (li < 4 ? li : (li < 16 ? li - 8 : li - 12));
  x = k & 7;
  z = 2 * (s >> 2) + (k >> 3);
  y = 2 * (s & 3) + (g1 ? 1 : 0);
}

template <int BN>
struct HpTile {
  // 128 columns: 4 x 2 waves of 64 x 64; 64 columns: 8 x 1 waves of 32 x 64 (one A read per k-sub: the patch reads
  // carry the per-tap address arithmetic, the weight reads none)
  static constexpr int WM = BN == 64 ? 8 : 4, WN = 8 / WM, TM = HB_BM / WM / 32, TN = BN / WN / 32;
  static constexpr int BPIECES = BN / 64;
  static constexpr int TPS = BN == 64 ? 3 : 1;                        // taps per ring stage (and per block barrier): with 64
                                                                      // columns a tap is only 8 MFMAs per wave, the barrier
                                                                      // per tap cost a third of the loop (MFMA-only what-if)
  static constexpr int BTAP = BN * HB_ROWB;
  static constexpr int BSTAGE = TPS * BTAP;
  // ring stages: with 128 columns and one 16 KiB tap per stage, three stages keep 32 KiB of weights in flight -- the
  // ring then streams at 32 KiB per load latency = 28 GB/s per CU (fill-only what-if of D.conv2's forward: 2.3 ms of a
  // 3.9 ms loop); five stages (the patch + 80 KiB = all the LDS there is) keep four taps in flight
  static constexpr int NR = BN == 128 ? 5 : 3;
  static constexpr int SMEM_LOOP = HP_PATCH + NR * BSTAGE;
  static constexpr int EP_ROWPIX = 8 * HbSlab<TN>::WAVE;              // behind the eight waves' slabs
  static constexpr int EP_PART = EP_ROWPIX + HB_BM * 4;
  static constexpr int SMEM_EPI = EP_PART + WM * 2 * BN * 4;
  static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
};

struct HpGrid { int tiles_z, tiles_y, tiles_x; };

template <int BN>
__global__ __launch_bounds__(512, 1) void gather_patch_bf16_kernel(const GatherConv p, const HpGrid tg) {
  using T = HpTile<BN>;
  constexpr int TM = T::TM, TN = T::TN, WN = T::WN;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const BlockId bid = conv_block_id(p);
  const Phase& ph = p.ph[0];
  const int n0 = bid.nt * BN;
  const int stats_row = bid.mt;
  const int Cout = p.Cout, Cin = p.Cin, Di = p.Di, Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  // tile origin
  int t = bid.mt;
  const int tx = t % tg.tiles_x; t /= tg.tiles_x;
  const int ty = t % tg.tiles_y; t /= tg.tiles_y;
  const int tz = t % tg.tiles_z;
  const int n = t / tg.tiles_z;
  const int oz0 = tz * HP_TZ, oy0 = ty * HP_TY, ox0 = tx * HP_TX;
  // patch origin in the gathered tensor: the smallest input coordinate any tap of the tile's first pixel reads
  const int mnz = ph.dz0 + (p.dstep[0] < 0 ? 2 * p.dstep[0] : 0), mny = ph.dy0 + (p.dstep[1] < 0 ? 2 * p.dstep[1] : 0),
            mnx = ph.dx0 + (p.dstep[2] < 0 ? 2 * p.dstep[2] : 0);
  const int pz0 = oz0 + mnz, py0 = oy0 + mny, px0 = ox0 + mnx;
  const char* __restrict__ ginb = reinterpret_cast<const char*>(p.in);
  const char* __restrict__ gwb = reinterpret_cast<const char*>(p.wp);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // ---- this thread's patch pieces: piece q = tid + 512 i -> patch row q >> 3, LDS chunk position q & 7 ----
  unsigned ppB[HP_PPIECES];            // byte offset of (pixel, channel 0) or ~0u: out of range / beyond the patch
  const int pck = (tid & 7) ^ (((tid >> 3) >> 1) & 7);      // source chunk: (row >> 1) & 7 with row = (q >> 3); 64 i drops out
#pragma unroll
  for (int i = 0; i < HP_PPIECES; ++i) {
    const int pr = (tid >> 3) + 64 * i;                       // LDS row (planes padded to HP_PZ rows: rem >= 100 is padding)
    const int pz = pr / HP_PZ, rem = pr - pz * HP_PZ;
    const int py = rem / HP_PX, px = rem - py * HP_PX;
    const int iz = pz0 + pz, iy = py0 + py, ix = px0 + px;
    const bool ok = pr < HP_PROWS && rem < HP_PY * HP_PX && (unsigned)iz < (unsigned)Di && (unsigned)iy < (unsigned)Hi &&
                    (unsigned)ix < (unsigned)Wi;
    ppB[i] = ok ? (unsigned)((((n * Di + iz) * Hi + iy) * Wi + ix)) * (unsigned)ldi * 2u : 0xFFFFFFFFu;
  }
  const int r0 = tid >> 3, cc = tid & 7;
  const int ck = cc ^ ((r0 >> 1) & 7);
  const unsigned Ktot2 = (unsigned)(p.Kz * p.Ky * p.Kx * Cin) * 2u;
  unsigned wrowB[T::BPIECES];
#pragma unroll
  for (int i = 0; i < T::BPIECES; ++i) {
    int co = n0 + r0 + 64 * i;
    co = co < Cout ? co : Cout - 1;
    wrowB[i] = (unsigned)co * Ktot2;
  }
  const int nchunk = Cin / HB_BK;
  char* const patch = lds;
  char* const bring = lds + HP_PATCH;

  const int dbg = HB_DBG;                           // what-if bits as in the K-stepped kernels (2, 4, 16, 32)
  auto issue_patch = [&](int chunk) {
    const unsigned ciB = (unsigned)(chunk * HB_BK + pck * 8) * 2u;
    if (!(dbg & 16))
#pragma unroll
    for (int i = 0; i < HP_PPIECES; ++i) {
      const int pr = (tid >> 3) + 64 * i;
      if (pr < HP_PROWS) {
        const char* src = ppB[i] != 0xFFFFFFFFu ? ginb + (ppB[i] + ciB) : zero;
        GLDS16(src, patch + (8 * wid + 64 * i) * HB_ROWB);
      }
    }
  };
  // weights of tap (jz, jy, jx) of the current chunk -> ring stage
  int jx = 0, jy = 0, jz = 0, bstage = 0, btap = 0;
  auto issue_b = [&](int chunk) {
    const int kz = ph.kz0 + p.kstep[0] * jz, ky = ph.ky0 + p.kstep[1] * jy, kx = ph.kx0 + p.kstep[2] * jx;
    const unsigned woffB = (unsigned)(((kz * p.Ky + ky) * p.Kx + kx) * Cin * 2);
    const unsigned ciB = (unsigned)(chunk * HB_BK + ck * 8) * 2u;
    char* Bs = bring + bstage * T::BSTAGE + btap * T::BTAP + (8 * wid) * HB_ROWB;
    if (!(dbg & 16))
#pragma unroll
    for (int i = 0; i < T::BPIECES; ++i) GLDS16(gwb + (wrowB[i] + woffB + ciB), Bs + 64 * i * HB_ROWB);
    jx += 1;
    if (jx == 3) { jx = 0; jy += 1; }
    if (jy == 3) { jy = 0; jz += 1; }
    if (jz == 3) jz = 0;
    btap += 1;
    if (btap == T::TPS) { btap = 0; bstage = bstage == T::NR - 1 ? 0 : bstage + 1; }
  };
  auto issue_group = [&](int chunk) {                   // the TPS taps of one ring stage
#pragma unroll
    for (int t = 0; t < T::TPS; ++t) issue_b(chunk);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // A fragments: tile row of lane li in wave tile tm -> its patch row at tap offset (0, 0, 0)
  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    int z, y, x;
    hp_row_pixel(wm * TM + tm, li, z, y, x);
    abase[tm] = z * HP_PZ + y * HP_PX + x;
  }
  // B fragments as in the K-stepped kernel
  const int sw = (li >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = ((2 * s + lh) ^ sw) * 16;
  const int brow = (wn * (BN / WN) + li) * HB_ROWB;
  const unsigned lds_base = lds_addr(lds);
  const unsigned ring_base = lds_base + HP_PATCH;

  i32x4 fa[2][TM], fb[2][TN];
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    // every wave is done with the previous chunk's patch and ring: stage the new patch and the first two taps
    asm volatile("s_waitcnt vmcnt(0) ; tail: chunk boundary" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    issue_patch(chunk);
    // groups 0 .. NR-1 fill the ring (27 / TPS >= NR); group g+NR is issued when group g's stage is free
#pragma unroll
    for (int i = 0; i < T::NR; ++i) issue_group(chunk);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((T::NR - 1) * T::BPIECES * T::TPS) : "memory");   // patch + group 0 landed (this wave's part)
    asm volatile("s_barrier" ::: "memory");
    int cstage = 0, tin = 0;                                               // ring stage of the current tap, its slot in it
    int tzo = p.dstep[0] < 0 ? 2 : 0, tyo = p.dstep[1] < 0 ? 2 : 0, txo = p.dstep[2] < 0 ? 2 : 0;   // tap 0's patch offset
    int kx_ = 0, ky_ = 0;
    auto read_frags = [&](int delta, int stage, int slot, int s, int set) {
      if (dbg & 4) return;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int pr = abase[tm] + delta;
        fa[set][tm] = lds_read_b128(lds_base + pr * HB_ROWB + (((2 * s + lh) ^ ((pr >> 1) & 7)) << 4));
      }
      const unsigned Bs = ring_base + stage * T::BSTAGE + slot * T::BTAP + brow;
      lds_read_b128_n<TN, 32 * HB_ROWB>(fb[set], Bs + foff[s]);
    };
    int delta = tzo * HP_PZ + tyo * HP_PX + txo;
    read_frags(delta, 0, 0, 0, 0);
    for (int tap = 0; tap < 27; ++tap) {
      const int nstage = cstage == T::NR - 1 ? 0 : cstage + 1;
      const bool group_end = tin == T::TPS - 1;           // (27 = 9 x 3: groups never straddle the chunk's end)
      // next tap's patch offset (wave-uniform walk, x fastest)
      int ndelta = delta;
      {
        int nx = kx_ + 1, ny = ky_, carry_y = 0;
        if (nx == 3) { nx = 0; ny += 1; }
        if (ny == 3) { ny = 0; carry_y = 1; }
        const int sx = p.dstep[2] < 0 ? -1 : 1, sy = p.dstep[1] < 0 ? -1 : 1, sz = p.dstep[0] < 0 ? -1 : 1;
        ndelta += sx * (nx - kx_) + sy * (ny - ky_) * HP_PX + sz * carry_y * HP_PZ;
        kx_ = nx; ky_ = ny;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int set = s & 1;
        lds_wait<TM, TN>(fa[set], fb[set]);
        if (s < 3) {
          read_frags(delta, cstage, tin, s + 1, set ^ 1);
        } else if (tap + 1 < 27) {
          // ONE read site for both cases (the next tap of this stage: no barrier, nothing to wait for; or the first tap
          // of the next stage): two sites writing the same fragment registers from different branches made the
          // compiler merge them with register copies, which run before the unprotected asm reads have landed
          if (group_end) {
            // the next group must have landed; behind it in flight: the groups up to min(g + NR - 1, last)
            constexpr int GP = T::BPIECES * T::TPS, NG = 27 / T::TPS;
            const int g = tap / T::TPS;
            const int behind = (g + T::NR - 1 < NG - 1 ? g + T::NR - 1 : NG - 1) - (g + 1);
            if (behind >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * GP) : "memory");
            else if (behind == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GP) : "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) ; tail: the last group" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
          }
          read_frags(ndelta, group_end ? nstage : cstage, group_end ? 0 : tin + 1, 0, 0);
          if (group_end && tap + 1 + (T::NR - 1) * T::TPS < 27) issue_group(chunk);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 2))
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][tm]),
                                                                  __builtin_bit_cast(bf16x8, fb[set][tn]), acc[tm][tn], 0, 0, 0);
      }
      if (group_end) { cstage = nstage; tin = 0; } else tin += 1;
      delta = ndelta;
    }
  }
  asm volatile("s_barrier" ::: "memory");
  if (dbg & 32) return;                               // (what-if: no epilogue)

  // ---- epilogue from the accumulators (hb_epilogue): wave (wm, wn) holds tile rows 64 wm .. + 63 = sub-tiles
  //      2 wm, 2 wm + 1 (hp_row_pixel) and columns wn * BN/2 .. ----
  int* rowpix = reinterpret_cast<int*>(lds + T::EP_ROWPIX);
  if (tid < HB_BM) {                                  // tile row tid = sub-tile (tid >> 5), MFMA row (tid & 31)
    int z, y, x;
    hp_row_pixel(tid >> 5, tid & 31, z, y, x);
    const int oz = oz0 + z, oy = oy0 + y, ox = ox0 + x;
    rowpix[tid] = (oz < ph.Mz && oy < ph.My && ox < ph.Mx) ? ((n * p.Do + oz) * p.Ho + oy) * p.Wo + ox : -1;
  }
  const BwdStats bwl = p.bwd;                          // (a local copy: no pointer into the kernel arguments)
  __syncthreads();
  hb_epilogue<TM, TN, T::WM, BN>(acc, reinterpret_cast<float*>(lds + wid * HbSlab<TN>::WAVE), rowpix,
                                 reinterpret_cast<float*>(lds + T::EP_PART), wm * TM * 32, wn * (BN / WN), n0 + wn * (BN / WN), wm,
                                 p.bias, p.stats ? p.stats + (long)stats_row * 2 * Cout : nullptr, n0, Cout,
                                 reinterpret_cast<char*>(p.out), p.ldo, tid, lane, &bwl,
                                 bwl.part ? bwl.part + (long)stats_row * 3 * Cout : nullptr, false);
}

static bool hp_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_HB_PATCH") != nullptr;
  if (off || p.nphase != 1) return false;
  const Phase& ph = p.ph[0];
  if (!(ph.nz == 3 && ph.ny == 3 && ph.nx == 3 && p.Kz == 3 && p.Ky == 3 && p.Kx == 3)) return false;
  for (int d = 0; d < 3; ++d)
    if (p.istride[d] != 1 || p.ostride[d] != 1 || (p.dstep[d] != 1 && p.dstep[d] != -1)) return false;
  return ph.oz == 0 && ph.oy == 0 && ph.ox == 0 && p.Do == ph.Mz && p.Ho == ph.My && p.Wo == ph.Mx && ph.Mz >= 2 * HP_TZ &&
         ph.My >= 2 * HP_TY && ph.Mx >= 2 * HP_TX;
}

// Patch form or K-stepped form?  With 128 produced channels (D.conv2's forward at C5) the 512 x 128 wide K-stepped
// kernel is the faster one since round 3 (4.33 -> 3.74 ms at 126^3 bs 4, same box, gpurun_out/r4/conv2_whatif.txt: the
// patch kernel's 64 x 64 wave tiles read four fragments per four MFMAs and its one block per CU leaves every tile's
// patch fill and epilogue exposed); with 64 produced channels (the backward-data) only the narrow K-stepped kernel
// exists (6.5 ms) and a patch form stays (the big-patch kernel below where it applies).  Decided on the COMPACT geometry so that the statistics-row query
// and the launch agree (hb_dispatch refuses a pitched operand that changes the wide form's availability).
static bool hp8_use(const GatherConv& p, bool with_stats);
static bool hp_use(const GatherConv& p, bool with_stats) {
  if (p.Cin % HB_BK != 0 || !hp_ok(p)) return false;
  static const bool always = dev_env("MPGAN_DBG_HB_PATCH_ALWAYS") != nullptr;   // (A/B runs, make DEV=1)
  if (always) return true;
  GatherConv c = p;
  c.ldi = c.Cin;
  return !(p.Cout > 64 && hw_choice(c, with_stats) == 2);
}

static HpGrid hp_grid(const GatherConv& p) {
  const Phase& ph = p.ph[0];
  return HpGrid{(ph.Mz + HP_TZ - 1) / HP_TZ, (ph.My + HP_TY - 1) / HP_TY, (ph.Mx + HP_TX - 1) / HP_TX};
}

template <int BN>
static int hp_launch(const GatherConv& p, hipStream_t st) {
  auto kern = gather_patch_bf16_kernel<BN>;
  constexpr int smem = HpTile<BN>::SMEM;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_patch_bf16: hipFuncSetAttribute(%d): %s", smem, hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  static int dbg_set = -1;
  if (dbg_set < 0) {
    const char* e = dev_env("MPGAN_DBG_HB");
    dbg_set = e ? atoi(e) : 0;
    if (dbg_set) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hb_dbg), &dbg_set, sizeof(int));
  }
  const HpGrid tg = hp_grid(p);
  GatherConv q = p;
  q.mtiles = p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  q.ntiles = (p.Cout + BN - 1) / BN;
  q.phase_outer = 0;
  q.ksplit = 1;
  dim3 grid((unsigned)q.mtiles * q.ntiles);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, q, tg);
  return check_launch("gather_patch_bf16");
}

// ---------------------------------------------------------------------------
// Big-patch form (round 4): the same idea on an 8 x 8 x 8 output tile with 32-channel chunks.
// What the round-4 what-if runs of the 4 x 8 x 8 kernel above said (gpurun_out/r4/conv2_whatif.txt, D.conv2 at 126^3
// bs 4): its fragment reads + MFMAs alone, no DMA and no epilogue, take 3.18 ms of the 4.33 ms forward -- 54 % of the
// matrix rate with nothing to wait for -- because a 64 x 64 wave tile issues four ds_read_b128 per four MFMAs, and the
// 512 x 128 K-stepped kernel (128 x 64 per wave: six reads per eight MFMAs) already beats it on the forward (3.74 ms)
// although it stages 2.1 x the bytes per FLOP.  This kernel combines the two: 512 tile rows give eight waves of
// 128 x 64 (BN = 128) or 64 x 64 (BN = 64), the 10 x 10 x 10 patch of ONE 32-channel chunk is 1000 rows of 64 B
// (1040 with the plane pitch padded to 104 rows, 65 KiB), weights stream through a three-stage ring in groups of
// TPS taps (one barrier per 48 / 32 MFMAs of a wave).  LDS rows are 64 B = four 16-byte chunks; chunk c of row r sits
// at c ^ ((r >> 2) & 3), so the 16 rows of a ds_read_b128 lane group (distinct mod 16, as above) cover the 64 banks
// once.  Both operands come by buffer-load LDS-DMA: a per-thread voffset computed once (out-of-image patch rows and
// the pad rows carry HW_OOB and read zeros), the chunk / tap walk in the scalar soffset -- no vector instruction per
// piece.  A piece is a half line (64 B per pixel); at 400 FLOP per staged byte the fill this kernel needs (10 GB/s per
// CU at half the matrix rate) is a quarter of what half-line pieces were measured to deliver (39 GB/s, round 3).
// ---------------------------------------------------------------------------
constexpr int H8_T = 8, H8_P = H8_T + 2;
constexpr int H8_PZ = 104;                                     // plane pitch in rows: 100 -> 104 (= 8 mod 16)
constexpr int H8_PROWS = H8_P * H8_PZ;                         // 1040 rows
constexpr int H8_ROWB = 64, H8_BK = 32;
constexpr int H8_PATCH = H8_PROWS * H8_ROWB;                   // 66,560 B
constexpr int H8_PPIECES = (H8_PROWS + 127) / 128;             // rounds of 8 waves x 16 rows: 9 (the last: wave 0 only)
constexpr int H8_BM = H8_T * H8_T * H8_T;                      // 512

template <int BN>
struct H8Tile {
  static constexpr int WM = BN == 128 ? 4 : 8, WN = 8 / WM;
  static constexpr int TM = H8_BM / WM / 32, TN = BN / WN / 32;        // 4 x 2 or 2 x 2 MFMA tiles per wave
  static constexpr int TPS = BN == 128 ? 3 : 4;                        // taps per ring stage and barrier
  static constexpr int NG = (27 + TPS - 1) / TPS;                      // 9 / 7 groups (the last of BN = 64 holds 3 taps)
  static constexpr int BTAP = BN * H8_ROWB;
  static constexpr int BSTAGE = TPS * BTAP;                            // 24 / 16 KiB
  static constexpr int GP = BSTAGE / 8192;                             // LDS-DMA instructions per wave and group: 3 / 2
  static constexpr int NR = 3;
  static constexpr int SMEM_LOOP = H8_PATCH + NR * BSTAGE;             // 137 / 113 KiB
  static constexpr int EP_ROWPIX = 8 * HbSlab<TN>::WAVE;
  static constexpr int EP_PART = EP_ROWPIX + H8_BM * 4;
  static constexpr int SMEM_EPI = EP_PART + WM * 2 * BN * 4;
  static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
  static_assert(BSTAGE % 8192 == 0 && (BN == 128 || BN == 64), "whole DMA passes per group");
};

template <int BN>
__global__ __launch_bounds__(512, 1) void gather_patch8_bf16_kernel(const GatherConv p, const HpGrid tg) {
  using T = H8Tile<BN>;
  constexpr int TM = T::TM, TN = T::TN, WN = T::WN, TPS = T::TPS, NG = T::NG, GP = T::GP, NR = T::NR;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const BlockId bid = conv_block_id(p);
  const Phase& ph = p.ph[0];
  const int n0 = bid.nt * BN;
  const int stats_row = bid.mt;
  const int Cout = p.Cout, Cin = p.Cin, Di = p.Di, Hi = p.Hi, Wi = p.Wi, ldi = p.ldi;
  int t = bid.mt;
  const int tx = t % tg.tiles_x; t /= tg.tiles_x;
  const int ty = t % tg.tiles_y; t /= tg.tiles_y;
  const int tz = t % tg.tiles_z;
  const int n = t / tg.tiles_z;
  const int oz0 = tz * H8_T, oy0 = ty * H8_T, ox0 = tx * H8_T;
  // patch origin in the gathered tensor: the smallest coordinate any tap of the tile's first pixel reads
  const int mnz = ph.dz0 + (p.dstep[0] < 0 ? 2 * p.dstep[0] : 0), mny = ph.dy0 + (p.dstep[1] < 0 ? 2 * p.dstep[1] : 0),
            mnx = ph.dx0 + (p.dstep[2] < 0 ? 2 * p.dstep[2] : 0);
  const int pz0 = oz0 + mnz, py0 = oy0 + mny, px0 = ox0 + mnx;
  const unsigned Ktot2 = (unsigned)(p.Kz * p.Ky * p.Kx * Cin) * 2u;
  const unsigned bytesA = (unsigned)((((long)p.N * Di * Hi * Wi - 1) * ldi + Cin) * 2);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, (unsigned)Cout * Ktot2, 0x00020000);

  // ---- this thread's pieces: LDS row (tid >> 2) + 128 i, physical chunk tid & 3 = logical chunk ck ----
  const int r0 = tid >> 2;
  const int ck = (tid & 3) ^ ((r0 >> 2) & 3);                  // (128 i and the column-block offsets drop out of (row >> 2) & 3)
  unsigned ppB[H8_PPIECES];                                    // byte offset of (patch pixel, channel 8 ck) or HW_OOB
#pragma unroll
  for (int i = 0; i < H8_PPIECES; ++i) {
    const int pr = r0 + 128 * i;
    const int pz = pr / H8_PZ, rem = pr - pz * H8_PZ;
    const int py = rem / H8_P, px = rem - py * H8_P;
    const int iz = pz0 + pz, iy = py0 + py, ix = px0 + px;
    const bool ok = pr < H8_PROWS && rem < H8_P * H8_P && (unsigned)iz < (unsigned)Di && (unsigned)iy < (unsigned)Hi &&
                    (unsigned)ix < (unsigned)Wi;
    ppB[i] = ok ? (unsigned)(((n * Di + iz) * Hi + iy) * Wi + ix) * (unsigned)ldi * 2u + (unsigned)ck * 16u : HW_OOB;
  }
  // weights: flat row fr = r0 + 128 i of a group = (tap-in-group fr / BN, column fr % BN)
  unsigned wvoff;                                               // (the same column for every i: 128 % BN == 0)
  {
    int co = n0 + (r0 % BN);
    co = co < Cout ? co : Cout - 1;                             // clamped columns are computed and dropped
    wvoff = (unsigned)co * Ktot2 + (unsigned)ck * 16u;
  }
  const int tapsel0 = BN == 128 ? 0 : (wid >> 2);               // BN = 64: a pass covers two taps, waves 4-7 the second
  const int nchunk = Cin / H8_BK;
  char* const patch = lds;
  char* const bring = lds + H8_PATCH;
  const int wrow = __builtin_amdgcn_readfirstlane(wid * 16 * H8_ROWB);
  const int dbg = HB_DBG;

  auto issue_patch = [&](int chunk) {
    if (dbg & 16) return;
#pragma unroll
    for (int i = 0; i < H8_PPIECES; ++i)
      if (i < H8_PPIECES - 1 || wid == 0) BLDS16(rsA, patch + wrow + 128 * i * H8_ROWB, ppB[i], chunk * (H8_BK * 2));
  };
  auto tap_woff = [&](int T_) {                                 // byte offset of tap number T_ (walk order, x fastest) in a weight row
    const int Tc = T_ < 27 ? T_ : 26;                           // (dummy taps of the last group re-read the last one)
    const int jz = Tc / 9, r9 = Tc - 9 * jz, jy = r9 / 3, jx = r9 - 3 * jy;
    const int kz = ph.kz0 + p.kstep[0] * jz, ky = ph.ky0 + p.kstep[1] * jy, kx = ph.kx0 + p.kstep[2] * jx;
    return ((kz * p.Ky + ky) * p.Kx + kx) * Cin * 2;
  };
  int gi = 0, gstage = 0;                                       // next group to issue, its ring stage
  auto issue_group = [&](int chunk) {
    if (!(dbg & 16)) {
#pragma unroll
      for (int i = 0; i < GP; ++i) {
        const int tsel = BN == 128 ? i : 2 * i + tapsel0;
        BLDS16(rsB, bring + gstage * T::BSTAGE + wrow + 128 * i * H8_ROWB, wvoff, tap_woff(gi * TPS + tsel) + chunk * (H8_BK * 2));
      }
    }
    gi += 1;
    gstage = gstage == NR - 1 ? 0 : gstage + 1;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // A fragments: tile row of lane li in wave tile tm -> its patch row at tap offset (0, 0, 0)
  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    int z, y, x;
    hp_row_pixel(wm * TM + tm, li, z, y, x);
    abase[tm] = (z * H8_PZ + y * H8_P + x) * H8_ROWB;          // in bytes
  }
  const int bsw = (li >> 2) & 3;
  const int brow = (wn * (BN / WN) + li) * H8_ROWB;
  const unsigned lds_base = lds_addr(lds);
  const unsigned ring_base = lds_base + H8_PATCH;
  const int sx = p.dstep[2] < 0 ? -1 : 1, sy = p.dstep[1] < 0 ? -1 : 1, sz = p.dstep[0] < 0 ? -1 : 1;
  const int delta0 = (p.dstep[0] < 0 ? 2 : 0) * H8_PZ + (p.dstep[1] < 0 ? 2 : 0) * H8_P + (p.dstep[2] < 0 ? 2 : 0);

  i32x4 fa[2][TM], fb[2][TN];
  unsigned acur[TM];                                   // this tap's A fragment addresses of k-sub 0
  auto read_frags = [&](int delta, int stage, int slot, int s, int set) {
    if (dbg & 4) return;
    // k-sub 1's chunk is k-sub 0's with bit 1 flipped, and the swizzle is an XOR: its address is k-sub 0's ^ 32
    // (one vector instruction per fragment instead of the four that form k-sub 0's)
    if (s == 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const unsigned u = (unsigned)(abase[tm] + delta * H8_ROWB);        // byte offset of the patch row (delta: wave-uniform)
        acur[tm] = lds_base + u + ((lh ^ ((u >> 8) & 3)) << 4);            // (row >> 2) & 3 = bits 8-9 of 64 row
        fa[set][tm] = lds_read_b128(acur[tm]);
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[set][tm] = lds_read_b128(acur[tm] ^ 32u);
    }
    const unsigned Bs = ring_base + stage * T::BSTAGE + slot * T::BTAP + brow + (((2 * s + lh) ^ bsw) << 4);
    lds_read_b128_n<TN, 32 * H8_ROWB>(fb[set], Bs);
  };
  auto mfmas = [&](int set) {
    if (dbg & 2) return;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][tm]),
                                                              __builtin_bit_cast(bf16x8, fb[set][tn]), acc[tm][tn], 0, 0, 0);
  };

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    // every wave is done with the previous chunk's patch and ring
    asm volatile("s_waitcnt vmcnt(0) ; tail: chunk boundary" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    issue_patch(chunk);
    gi = 0; gstage = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) issue_group(chunk);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NR - 1) * GP) : "memory");      // the patch + group 0 landed (this wave's part)
    asm volatile("s_barrier" ::: "memory");
    int cstage = 0, tin = 0;                            // ring stage of the current tap, its slot in it
    int kx_ = 0, ky_ = 0;
    int delta = delta0;
    read_frags(delta, 0, 0, 0, 0);
    for (int tap = 0; tap < 27; ++tap) {
      const int nstage = cstage == NR - 1 ? 0 : cstage + 1;
      const bool group_end = tin == TPS - 1 || tap == 26;
      // next tap's patch offset (wave-uniform walk, x fastest)
      int ndelta = delta;
      {
        int nx = kx_ + 1, ny = ky_, carry_y = 0;
        if (nx == 3) { nx = 0; ny += 1; }
        if (ny == 3) { ny = 0; carry_y = 1; }
        ndelta += sx * (nx - kx_) + sy * (ny - ky_) * H8_P + sz * carry_y * H8_PZ;
        kx_ = nx; ky_ = ny;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int set = s;
        lds_wait<TM, TN>(fa[set], fb[set]);
        if (s == 0) {
          read_frags(delta, cstage, tin, 1, 1);
        } else if (tap < 26) {
          // ONE read site for both cases (the next tap of this stage, or the first tap of the next stage behind the
          // barrier): see gather_patch_bf16_kernel
          if (group_end) {
            // group g+1 must have landed; behind it in flight: the groups up to min(g + NR - 1, NG - 1)
            const int g = tap / TPS;
            const int behind = (g + NR - 1 < NG - 1 ? g + NR - 1 : NG - 1) - (g + 1);
            if (behind >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) ; tail: the last group" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
          }
          read_frags(ndelta, group_end ? nstage : cstage, group_end ? 0 : tin + 1, 0, 0);
          if (group_end && gi < NG) issue_group(chunk);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set);
      }
      if (group_end) { cstage = nstage; tin = 0; } else tin += 1;
      delta = ndelta;
    }
  }
  asm volatile("s_barrier" ::: "memory");
  if (dbg & 32) return;                               // (what-if: no epilogue)

  // ---- epilogue from the accumulators (hb_epilogue): wave (wm, wn) holds tile rows 32 TM wm .. = sub-tiles TM wm ..
  //      (hp_row_pixel) and columns wn * BN / WN .. ----
  int* rowpix = reinterpret_cast<int*>(lds + T::EP_ROWPIX);
  {
    int z, y, x;
    hp_row_pixel(tid >> 5, tid & 31, z, y, x);         // tile row tid = sub-tile (tid >> 5), MFMA row (tid & 31)
    const int oz = oz0 + z, oy = oy0 + y, ox = ox0 + x;
    rowpix[tid] = (oz < ph.Mz && oy < ph.My && ox < ph.Mx) ? ((n * p.Do + oz) * p.Ho + oy) * p.Wo + ox : -1;
  }
  const BwdStats bwl = p.bwd;                          // (a local copy: no pointer into the kernel arguments)
  __syncthreads();
  hb_epilogue<TM, TN, T::WM, BN>(acc, reinterpret_cast<float*>(lds + wid * HbSlab<TN>::WAVE), rowpix,
                                 reinterpret_cast<float*>(lds + T::EP_PART), wm * TM * 32, wn * (BN / WN), n0 + wn * (BN / WN), wm,
                                 p.bias, p.stats ? p.stats + (long)stats_row * 2 * Cout : nullptr, n0, Cout,
                                 reinterpret_cast<char*>(p.out), p.ldo, tid, lane, &bwl,
                                 bwl.part ? bwl.part + (long)stats_row * 3 * Cout : nullptr, false);
}

// 8 x 8 x 8 tiles pay on maps of at least 16 pixels per dimension whose extent wastes little in the last tile
static bool hp8_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_HB_PATCH8") != nullptr;
  if (off || !hp_ok(p) || p.Cin % H8_BK != 0) return false;
  const Phase& ph = p.ph[0];
  if (ph.Mz < 2 * H8_T || ph.My < 2 * H8_T || ph.Mx < 2 * H8_T) return false;
  if ((long)p.N * p.Di * p.Hi * p.Wi * p.ldi * 2 >= (long)HW_OOB || (long)p.Cout * p.Cin * 27 * 2 >= (long)HW_OOB) return false;
  // the last tile of a dimension may be partly empty: at most 10 % of the issued rows in all
  const double used = (double)ph.Mz * ph.My * ph.Mx /
                      ((double)((ph.Mz + 7) / 8 * 8) * ((ph.My + 7) / 8 * 8) * ((ph.Mx + 7) / 8 * 8));
  return used >= 0.9 || p.min_blocks == 1;                 // (min_blocks = 1: the tests' way of running small shapes through the big-tile forms)
}
static HpGrid hp8_grid(const GatherConv& p) {
  const Phase& ph = p.ph[0];
  return HpGrid{(ph.Mz + H8_T - 1) / H8_T, (ph.My + H8_T - 1) / H8_T, (ph.Mx + H8_T - 1) / H8_T};
}

template <int BN>
static int hp8_launch(const GatherConv& p, hipStream_t st) {
  auto kern = gather_patch8_bf16_kernel<BN>;
  constexpr int smem = H8Tile<BN>::SMEM;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_patch8_bf16: hipFuncSetAttribute(%d): %s", smem, hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  static int dbg_set = -1;
  if (dbg_set < 0) {
    const char* e = dev_env("MPGAN_DBG_HB");
    dbg_set = e ? atoi(e) : 0;
    if (dbg_set) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hb_dbg), &dbg_set, sizeof(int));
  }
  const HpGrid tg = hp8_grid(p);
  GatherConv q = p;
  q.packed = 0;
  q.mtiles = p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  q.ntiles = (p.Cout + BN - 1) / BN;
  q.phase_outer = 0;
  q.ksplit = 1;
  dim3 grid((unsigned)q.mtiles * q.ntiles);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, q, tg);
  return check_launch("gather_patch8_bf16");
}

// Routing between the forms is decided on the COMPACT geometry (pitch = channels), like hp_use
// Measured at D.conv2's size (126^3 bs 4, product code, same box, gpurun_out/r4/conv2_patch8_b.txt; ms): 128 produced
// channels (forward): 512 x 128 K-stepped 3.85, big patch 4.14, 4 x 8 x 8 patch 4.25; 64 produced channels
// (backward-data; no wide K-stepped form exists): big patch 4.08, 4 x 8 x 8 patch 4.57-4.63.
static bool hp8_use(const GatherConv& p, bool with_stats) {
  GatherConv c = p;
  c.ldi = c.Cin;
  if (!hp8_ok(c)) return false;
  static const bool always = dev_env("MPGAN_DBG_HB_PATCH_ALWAYS") != nullptr;   // (A/B runs, make DEV=1)
  return always || !(p.Cout > 64 && hw_choice(c, with_stats) == 2);
}

static int hb_dispatch(const GatherConv& p, hipStream_t st, const char* what) {
  int rc = hb_check(p, what);
  if (rc) return rc;
  const long maxM = max_phase_pixels(p);
  if (maxM == 0) return MPGAN_OK;
  if (hp8_use(p, p.stats != nullptr)) {
    MPGAN_UNSUPPORTED(!hp8_ok(p), "%s: the channel pitch of this operand takes it beyond the 32-bit offset range of the patch form "
                      "the statistics rows were sized for: pass a compact tensor", what);
    return p.Cout > 64 ? hp8_launch<128>(p, st) : hp8_launch<64>(p, st);
  }
  if (hp_use(p, p.stats != nullptr)) return p.Cout > 64 ? hp_launch<128>(p, st) : hp_launch<64>(p, st);
  const bool mask = !hb_all_in_range(p);
  const int wide = hw_choice(p, p.stats != nullptr);
  if (p.stats) {
    // mpgan_conv_stats_rows_bf16 sized the caller's partial rows from the compact geometry (pitch = Cin): a pitched
    // operand close to the 4 GiB offset range can fall back to the narrower tile, which would write more rows
    GatherConv c = p;
    c.ldi = c.Cin;
    MPGAN_UNSUPPORTED(hw_bm(hw_choice(c, true)) != hw_bm(wide),
                      "%s: the channel pitch of this operand changes the tile height the statistics rows were sized for "
                      "(operand beyond the 32-bit offset range of the wide form): pass a compact tensor", what);
  }
  if (wide == 3) return mask ? hw_launch<2, 4, true, true, true>(p, maxM, st) : hw_launch<2, 4, false, true, true>(p, maxM, st);
  static const bool no_ring = dev_env("MPGAN_DBG_HB_NO_RING") != nullptr;     // development: two whole stages instead
  if (wide == 1 && no_ring) return mask ? hw_launch<2, 4, true, false>(p, maxM, st) : hw_launch<2, 4, false, false>(p, maxM, st);
  if (wide == 1) return mask ? hw_launch<2, 4, true>(p, maxM, st) : hw_launch<2, 4, false>(p, maxM, st);
  if (wide == 2) return mask ? hw_launch<4, 2, true>(p, maxM, st) : hw_launch<4, 2, false>(p, maxM, st);
  MPGAN_UNSUPPORTED(p.bwd.part != nullptr, "%s: no fused norm-backward sums on the narrow K-stepped kernel "
                                            "(mpgan_conv_bwd_stats_rows_bf16() == 0)", what);
  static int nw = 0;
  if (!nw) {
    const char* e = dev_env("MPGAN_DBG_HB_NW");      // development: force four or eight waves per block
    nw = e ? atoi(e) : 8;
  }
  if (nw == 4) {
    if (p.Cout > 64) return mask ? hb_launch<128, true, 4>(p, maxM, st) : hb_launch<128, false, 4>(p, maxM, st);
    return mask ? hb_launch<64, true, 4>(p, maxM, st) : hb_launch<64, false, 4>(p, maxM, st);
  }
  if (p.Cout > 64) return mask ? hb_launch<128, true, 8>(p, maxM, st) : hb_launch<128, false, 8>(p, maxM, st);
  return mask ? hb_launch<64, true, 8>(p, maxM, st) : hb_launch<64, false, 8>(p, maxM, st);
}

// ---------------------------------------------------------------------------
// Weight gradient, bf16 operands: R[cd][t*Cg + cg] = sum_m dense[m][cd] * gathered[pix(m,t)][cg], K = pixels.
// Both operands are stored [pixel][channel] (K-strided), so the LDS tiles keep that shape and the MFMA
// fragments (8 consecutive pixels of one channel per lane) come from ds_read_b64_tr_b16, the hardware
// transposing read: lane 4q+p of a 16-lane group addresses row q (pixel), columns 4p..4p+3 of a 4 x 16 block
// and receives one column.  Image: 256-byte rows, chunk ^ (((row&3)<<2) | ((row>>2)&3)) (conflict-free for
// these reads), applied on the LDS-DMA source address.
// Tile: 128 dense channels x 256 columns x 64 pixels per K-step, 4 waves (2 x 2, 64 x 128 each), three stages.
// Pad-free convs only (every tap of every coarse pixel in range): the discriminator's.
// ---------------------------------------------------------------------------
struct WgradHb {
  const char* dense;    // bf16 [M][ldd]
  const char* gath;     // bf16 [pixels][ldg]
  float* partial;       // [split][Cd][NC]
  int ldd, Cd, ldg, Cg;
  int N, Mz, My, Mx, Gz, Gy, Gx, Kz, Ky, Kx, sz, sy, sx;
  int nsplit, tiles_c, tiles_d;
  long chunk;           // pixels per split (multiple of 64)
  FastDiv fMx, fMy, fMz;
};

constexpr int WH_BD = 128, WH_BG = 256, WH_BK = 64;
constexpr int WH_STAGE = WH_BK * (WH_BD + WH_BG) * 2;      // 48 KiB
constexpr int WH_SMEM = 3 * WH_STAGE;

//   NW = 4: 2 x 2 waves of 64 x 128;  NW = 8: 2 x 4 waves of 64 x 64, two per SIMD (see gather_conv_bf16_kernel).
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void wgrad_bf16_kernel(const WgradHb p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int WN = NW / 2;                             // waves along the 256 columns
  constexpr int TM = 2, TN = WH_BG / WN / 32;
  constexpr int PR = NW * 4;                             // rows one LDS-DMA instruction of every wave fills
  constexpr int NP = WH_BK / PR;                         // row groups per thread
  constexpr int NLOADS = 3 * NP;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  const int tc = (int)(w % (unsigned)p.tiles_c);
  const unsigned wq = w / (unsigned)p.tiles_c;
  const int td = (int)(wq % (unsigned)p.tiles_d), split = (int)(wq / (unsigned)p.tiles_d);
  const int T = p.Kz * p.Ky * p.Kx;
  const int NC = T * p.Cg;
  const int c0 = tc * WH_BG, d0 = td * WH_BD;
  const long M = (long)p.N * p.Mz * p.My * p.Mx;
  const long mbeg = (long)split * p.chunk;
  const long mend = mbeg + p.chunk < M ? mbeg + p.chunk : M;
  const int nk = mbeg < mend ? (int)((mend - mbeg + WH_BK - 1) / WH_BK) : 0;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // this thread's pieces: rows 16*i + (tid >> 4) of each sub-tile, LDS position tid & 15, source chunk pos ^ f(row)
  const int prow = tid >> 4, ppos = tid & 15;
  const int f = ((prow & 3) << 2) | (wid & 3);            // f(row) = ((row&3)<<2) | ((row>>2)&3), row = 16i + prow
  const int chk = ppos ^ f;                               // 16-byte chunk (8 channels / columns) this thread fetches
  // dense operand: channels d0 + 8*chk
  const bool dok = d0 + chk * 8 < p.Cd;
  const unsigned dcolB = (unsigned)(d0 + chk * 8) * 2u;
  // gathered operand, sub-tiles j = 0, 1: column c0 + 128 j + 8*chk -> (tap, channel)
  unsigned gtapB[2];
  bool gok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = c0 + 128 * j + chk * 8;
    gok[j] = col < NC;
    const int t = gok[j] ? col / p.Cg : 0;
    const int ci = gok[j] ? col - t * p.Cg : 0;
    const int kx = t % p.Kx, tq = t / p.Kx, ky = tq % p.Ky, kz = tq / p.Ky;
    gtapB[j] = (unsigned)(((kz * p.Gy + ky) * p.Gx + kx) * p.ldg + ci) * 2u;
  }

  // Pixel cursors of this thread's rows (PR i + prow of every 64-pixel K-step): decoded with divisions
  // once, then advanced by 64 pixels per K-step with carries.  (No row table in LDS: hipcc drains every
  // LDS-DMA in flight in front of an LDS access it cannot tell apart from the staging area.)
  int cn[NP], cz[NP], cy[NP], cx[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    unsigned q, ux, uy, uz;
    fdivmod((unsigned)(mbeg + PR * i + prow), p.fMx, q, ux);
    fdivmod(q, p.fMy, q, uy);
    fdivmod(q, p.fMz, q, uz);
    cn[i] = (int)q; cz[i] = (int)uz; cy[i] = (int)uy; cx[i] = (int)ux;
  }
  int istage = 0;
  long mtile = mbeg;                                      // first pixel of the next tile to issue
  auto issue = [&]() {
    char* Ds = lds + istage * WH_STAGE + (4 * wid) * 256;           // wave base: rows PR i + 4 wid .. + 3
    char* Gs = lds + istage * WH_STAGE + WH_BK * 256 + (4 * wid) * 256;
    const unsigned mrowB = (unsigned)(mtile * p.ldd) * 2u;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = PR * i + prow;
      const bool valid = mtile + row < mend;
      const unsigned gpixB =
          (unsigned)(((cn[i] * p.Gz + cz[i] * p.sz) * p.Gy + cy[i] * p.sy) * p.Gx + cx[i] * p.sx) * (unsigned)p.ldg * 2u;
      const char* src = p.dense + (mrowB + (unsigned)(row * p.ldd) * 2u + dcolB);
      GLDS16((valid && dok) ? src : zero, Ds + PR * i * 256);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* gs = p.gath + (gpixB + gtapB[j]);
        GLDS16((valid && gok[j]) ? gs : zero, Gs + j * (WH_BK * 256) + PR * i * 256);
      }
      cx[i] += WH_BK;
      while (cx[i] >= p.Mx) {
        cx[i] -= p.Mx;
        if (++cy[i] == p.My) {
          cy[i] = 0;
          if (++cz[i] == p.Mz) { cz[i] = 0; ++cn[i]; }
        }
      }
    }
    mtile += WH_BK;
    istage = istage == 2 ? 0 : istage + 1;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // transposing-read addresses: group g = lane >> 4, i = lane & 15 -> q = i >> 2 (row inside the 4-row block),
  // pp = i & 3 (column quad); the group's 16 columns start at 16*(g&1) of a 32-channel tile, its pixels at
  // 16 s + 8 (g>>1) + 4 half
  const int tg = lane >> 4, tq4 = (lane & 15) >> 2, tpp = lane & 3;
  int aoff[TM][2], boff[TN][2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = 8 * (tg >> 1) + 4 * half + tq4;                   // pixel row inside a 16-pixel k-sub
    const int fr = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int chunk = (wm * 64 + tm * 32 + 16 * (tg & 1)) / 8 + (tpp >> 1);
      aoff[tm][half] = 256 * r + 16 * (chunk ^ fr) + 8 * (tpp & 1);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int colb = wn * (WH_BG / WN) + tn * 32 + 16 * (tg & 1);   // 0..255: sub-tile colb >> 7
      const int chunk = (colb & 127) / 8 + (tpp >> 1);
      boff[tn][half] = WH_BK * 256 * (1 + (colb >> 7)) + 256 * r + 16 * (chunk ^ fr) + 8 * (tpp & 1);
    }
  }

  const unsigned lds_base = lds_addr(lds);
  // (pipeline as in gather_conv_bf16_kernel: the K-step's barrier sits in front of its last k-sub's MFMAs)
  i32x2 fa[2][2 * TM], fb[2][2 * TN];                     // [set][2*tile + half]
  // fragment addresses of the stage being read, formed once per K-step (the k-sub sits in the read's offset field)
  unsigned pa[2 * TM], pb[2 * TN];
  auto set_bases = [&](int stage) {
    const unsigned S = lds_base + stage * WH_STAGE;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) { pa[2 * tm] = S + aoff[tm][0]; pa[2 * tm + 1] = S + aoff[tm][1]; }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { pb[2 * tn] = S + boff[tn][0]; pb[2 * tn + 1] = S + boff[tn][1]; }
  };
  auto read_frags = [&](int s, int set) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) fa[set][i] = lds_read_tr16_ksub(pa[i], s);
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) fb[set][i] = lds_read_tr16_ksub(pb[i], s);
  };
  if (nk > 0) issue();
  if (nk > 1) issue();
  if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  set_bases(0);
  if (nk > 0) read_frags(0, 0);
  if (nk > 2) issue();
  int cstage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int nstage = cstage == 2 ? 0 : cstage + 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int set = s & 1;
      lds_wait<2 * TM, 2 * TN>(fa[set], fb[set]);
      if (s < 3) {
        read_frags(s + 1, set ^ 1);
      } else if (kt + 1 < nk) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) ; tail: the last tile" ::: "memory");      // (tools/check_isa.py)
        asm volatile("s_barrier" ::: "memory");
        set_bases(nstage);
        read_frags(0, 0);
        if (kt + 3 < nk) issue();
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const i32x4 av = __builtin_shufflevector(fa[set][2 * tm], fa[set][2 * tm + 1], 0, 1, 2, 3);
          const i32x4 bv = __builtin_shufflevector(fb[set][2 * tn], fb[set][2 * tn + 1], 0, 1, 2, 3);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                acc[tm][tn], 0, 0, 0);
        }
    }
    cstage = nstage;
  }

  float* out = p.partial + (long)split * p.Cd * NC;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = c0 + wn * (WH_BG / WN) + tn * 32 + li;
      if (col >= NC) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cd = d0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (cd < p.Cd) out[(long)cd * NC + col] = acc[tm][tn][r];
      }
    }
}

// ---------------------------------------------------------------------------
// Wide form of the weight gradient (round 3), for layers with >= 256 dense channels (D.conv3 / D.conv4): 256 dense
// channels x 256 columns per block, eight waves as 2 x 4 of 128 x 64 (the same move as gather_conv_bf16_wide_kernel:
// 128 FLOP per staged byte instead of 85, 12 transposing reads per 8 MFMAs instead of 8 per 4).  Each operand of a
// K-step (64 pixels x 256 channels = two 128-channel sub-tiles of 256-byte rows, 32 KiB) is one unit of the five-unit
// ring described there: the gathered tile of step kt+3 and the dense tile of step kt+2 are issued behind step kt's
// barrier, two LDS-DMA pieces per k-sub.  Swizzle, transposing reads and the pixel cursors are wgrad_bf16_kernel's.
// ---------------------------------------------------------------------------
constexpr int WW_B = 256;                                  // dense channels = columns per block
constexpr int WW_SUB = WH_BK * 256;                        // one 128-channel sub-tile: 64 rows of 256 B = 16 KiB
constexpr int WW_UNIT = 2 * WW_SUB;                        // 32 KiB
constexpr int WW_SMEM = 5 * WW_UNIT;

__global__ __launch_bounds__(512, 1) void wgrad_bf16_wide_kernel(const WgradHb p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int TM = 4, TN = 2, WN = 4;
  constexpr int PR = 32;                                 // rows one LDS-DMA instruction of every wave fills (8 waves x 4)
  constexpr int NP = WH_BK / PR;                         // row groups per thread: 2
  constexpr int GP = 2 * NP, DP = 2 * NP;                // pieces per thread and K-step: gathered, dense
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  const int tc = (int)(w % (unsigned)p.tiles_c);
  const unsigned wq = w / (unsigned)p.tiles_c;
  const int td = (int)(wq % (unsigned)p.tiles_d), split = (int)(wq / (unsigned)p.tiles_d);
  const int T = p.Kz * p.Ky * p.Kx;
  const int NC = T * p.Cg;
  const int c0 = tc * WW_B, d0 = td * WW_B;
  const long M = (long)p.N * p.Mz * p.My * p.Mx;
  const long mbeg = (long)split * p.chunk;
  const long mend = mbeg + p.chunk < M ? mbeg + p.chunk : M;
  const int nk = mbeg < mend ? (int)((mend - mbeg + WH_BK - 1) / WH_BK) : 0;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // this thread's pieces: rows 32*i + (tid >> 4) of each sub-tile, LDS position tid & 15, source chunk pos ^ f(row)
  const int prow = tid >> 4, ppos = tid & 15;
  const int f = ((prow & 3) << 2) | (wid & 3);            // f(row) = ((row&3)<<2) | ((row>>2)&3), row = 32i + prow
  const int chk = ppos ^ f;                               // 16-byte chunk (8 channels / columns) this thread fetches
  unsigned dcolB[2], gtapB[2];
  bool dok[2], gok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cd = d0 + 128 * j + chk * 8;
    dok[j] = cd < p.Cd;
    dcolB[j] = (unsigned)cd * 2u;
    const int col = c0 + 128 * j + chk * 8;
    gok[j] = col < NC;
    const int t = gok[j] ? col / p.Cg : 0;
    const int ci = gok[j] ? col - t * p.Cg : 0;
    const int kx = t % p.Kx, tq = t / p.Kx, ky = tq % p.Ky, kz = tq / p.Ky;
    gtapB[j] = (unsigned)(((kz * p.Gy + ky) * p.Gx + kx) * p.ldg + ci) * 2u;
  }
  int cn[NP], cz[NP], cy[NP], cx[NP];                     // pixel cursors of the gathered operand's rows
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    unsigned q, ux, uy, uz;
    fdivmod((unsigned)(mbeg + PR * i + prow), p.fMx, q, ux);
    fdivmod(q, p.fMy, q, uy);
    fdivmod(q, p.fMz, q, uz);
    cn[i] = (int)q; cz[i] = (int)uz; cy[i] = (int)uy; cx[i] = (int)ux;
  }
  const int wbase = __builtin_amdgcn_readfirstlane(4 * wid * 256);
  long mtileG = mbeg, mtileD = mbeg;                      // first pixel of the next gathered / dense tile to issue
  auto issueG = [&](int unit, auto part) {                // part: 0 = both row groups, 1 / 2 = the first / second
    constexpr int P = decltype(part)::value;
    constexpr int I0 = P == 2 ? NP / 2 : 0, I1 = P == 1 ? NP / 2 : NP;
    char* Gs = lds + unit + wbase;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const int row = PR * i + prow;
      const bool valid = mtileG + row < mend;
      const unsigned gpixB =
          (unsigned)(((cn[i] * p.Gz + cz[i] * p.sz) * p.Gy + cy[i] * p.sy) * p.Gx + cx[i] * p.sx) * (unsigned)p.ldg * 2u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* gs = p.gath + (gpixB + gtapB[j]);
        GLDS16((valid && gok[j]) ? gs : zero, Gs + j * WW_SUB + PR * i * 256);
      }
      cx[i] += WH_BK;
      while (cx[i] >= p.Mx) {
        cx[i] -= p.Mx;
        if (++cy[i] == p.My) {
          cy[i] = 0;
          if (++cz[i] == p.Mz) { cz[i] = 0; ++cn[i]; }
        }
      }
    }
    if constexpr (P != 1) mtileG += WH_BK;
  };
  auto issueD = [&](int unit, auto part) {
    constexpr int P = decltype(part)::value;
    constexpr int I0 = P == 2 ? NP / 2 : 0, I1 = P == 1 ? NP / 2 : NP;
    char* Ds = lds + unit + wbase;
    const unsigned mrowB = (unsigned)(mtileD * p.ldd) * 2u;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const int row = PR * i + prow;
      const bool valid = mtileD + row < mend;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* src = p.dense + (mrowB + (unsigned)(row * p.ldd) * 2u + dcolB[j]);
        GLDS16((valid && dok[j]) ? src : zero, Ds + j * WW_SUB + PR * i * 256);
      }
    }
    if constexpr (P != 1) mtileD += WH_BK;
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

  // transposing-read addresses (see wgrad_bf16_kernel): the wave's 128 dense channels are sub-tile wm, its 64 columns
  // the half (wn & 1) of sub-tile wn >> 1
  const int tg = lane >> 4, tq4 = (lane & 15) >> 2, tpp = lane & 3;
  int aoff[TM][2], boff[TN][2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r = 8 * (tg >> 1) + 4 * half + tq4;                   // pixel row inside a 16-pixel k-sub
    const int fr = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int chunk = (tm * 32 + 16 * (tg & 1)) / 8 + (tpp >> 1);
      aoff[tm][half] = wm * WW_SUB + 256 * r + 16 * (chunk ^ fr) + 8 * (tpp & 1);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int chunk = ((wn & 1) * 64 + tn * 32 + 16 * (tg & 1)) / 8 + (tpp >> 1);
      boff[tn][half] = (wn >> 1) * WW_SUB + 256 * r + 16 * (chunk ^ fr) + 8 * (tpp & 1);
    }
  }
  const unsigned lds_base = lds_addr(lds);
  i32x2 fa[2][2 * TM], fb[2][2 * TN];                     // [set][2*tile + half]
  // fragment addresses of the units being read, formed once per K-step (the k-sub sits in the read's offset field)
  unsigned pa[2 * TM], pb[2 * TN];
  auto set_bases = [&](int unitD, int unitG) {
    const unsigned D = lds_base + unitD, G = lds_base + unitG;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) { pa[2 * tm] = D + aoff[tm][0]; pa[2 * tm + 1] = D + aoff[tm][1]; }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { pb[2 * tn] = G + boff[tn][0]; pb[2 * tn + 1] = G + boff[tn][1]; }
  };
  auto read_frags = [&](int s, int set) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) fa[set][i] = lds_read_tr16_ksub(pa[i], s);
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) fb[set][i] = lds_read_tr16_ksub(pb[i], s);
  };
  // ring: units G_0 D_0 G_1 D_1 G_2 | D_2 G_3 D_3 ... in slots u % 5 (gather_conv_bf16_wide_kernel)
  constexpr int U = WW_UNIT;
  if (nk > 0) {
    issueG(0, ALL);
    issueD(U, ALL);
    if (nk > 1) { issueG(2 * U, ALL); issueD(3 * U, ALL); }
    if (nk > 2) issueG(4 * U, ALL);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GP + DP) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP + DP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    int ug = 0, ud = U;
    set_bases(ud, ug);
    read_frags(0, 0);
    bool pend_d = false, pend_g = false;
    int unit_d = 0, unit_g = 0;
    for (int kt = 0; kt < nk; ++kt) {
      int ng = ug + 2 * U, nd = ud + 2 * U;
      ng = ng >= 5 * U ? ng - 5 * U : ng;
      nd = nd >= 5 * U ? nd - 5 * U : nd;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int set = s & 1;
        lds_wait<2 * TM, 2 * TN>(fa[set], fb[set]);
        if (s < 3) {
          read_frags(s + 1, set ^ 1);
          if (s == 0 && pend_d) issueD(unit_d, HALF1);
          if (s == 1 && pend_g) issueG(unit_g, HALF0);
          if (s == 2 && pend_g) issueG(unit_g, HALF1);
        } else if (kt + 1 < nk) {
          if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GP) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) ; tail: the last tile" ::: "memory");      // (tools/check_isa.py)
          asm volatile("s_barrier" ::: "memory");
          pend_d = kt + 2 < nk;
          pend_g = kt + 3 < nk;
          unit_d = ug;                                   // the dense tile of step kt+2 takes the slot G_kt leaves
          unit_g = ud;
          if (pend_d) issueD(unit_d, HALF0);
          set_bases(nd, ng);
          read_frags(0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const i32x4 av = __builtin_shufflevector(fa[set][2 * tm], fa[set][2 * tm + 1], 0, 1, 2, 3);
            const i32x4 bv = __builtin_shufflevector(fb[set][2 * tn], fb[set][2 * tn + 1], 0, 1, 2, 3);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                  acc[tm][tn], 0, 0, 0);
          }
      }
      ug = ng;
      ud = nd;
    }
  }

  float* out = p.partial + (long)split * p.Cd * NC;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = c0 + wn * 64 + tn * 32 + li;
      if (col >= NC) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cd = d0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (cd < p.Cd) out[(long)cd * NC + col] = acc[tm][tn][r];
      }
    }
}

struct WgradHbPlan { int nsplit, tiles_c, tiles_d, wide; long chunk; };
static WgradHbPlan plan_wgrad_hb(int Cd, int NC, long M) {
  WgradHbPlan pl;
  static const bool no_wide = dev_env("MPGAN_DBG_HB_WIDE") && atoi(dev_env("MPGAN_DBG_HB_WIDE")) == 0;
  pl.wide = (Cd > WH_BD && !no_wide) ? 1 : 0;             // 256 x 256 tiles (wgrad_bf16_wide_kernel)
  pl.tiles_c = (NC + WH_BG - 1) / WH_BG;
  pl.tiles_d = (Cd + (pl.wide ? WW_B : WH_BD) - 1) / (pl.wide ? WW_B : WH_BD);
  const long tiles = (long)pl.tiles_c * pl.tiles_d;
  long ns = tiles <= 256 ? 256 / tiles : 1;               // one block per CU, ONE round: never a few blocks more than CUs
  const long maxsplit = M / (8 * WH_BK) > 1 ? M / (8 * WH_BK) : 1;
  if (ns > maxsplit) ns = maxsplit;
  if (ns > 64) ns = 64;
  if (ns < 1) ns = 1;
  long chunk = (M + ns - 1) / ns;
  chunk = (chunk + WH_BK - 1) / WH_BK * WH_BK;
  ns = (M + chunk - 1) / chunk;
  pl.nsplit = (int)(ns < 1 ? 1 : ns);
  pl.chunk = chunk;
  return pl;
}

// dW[cd][cg][t] = beta*dW + sum_split partial[split][cd][t*Cg+cg]  (conv_wgrad.hip)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nsplit, int Cd, int Cg,
                                    int T, float beta, const float* __restrict__ bp, int bsplit, float* __restrict__ db,
                                    int bias_blocks);

// ---------------------------------------------------------------------------
// Backward-data of the first layer in the bf16 path (D.conv1: 64 gathered channels -> the ONE image channel, pad-free
// 3 x 3 (x 3), stride 1; GAN_final.py:167): dx[o] = sum_t sum_c dz[o + d(t)][c] * w[t][c].
// thin_cout1_kernel<16, true> gave 16 lanes to an output pixel and fetched every dz element once per tap (27 x through
// L1: 2.8 ms at 128^3 for 1 GB of dz).  Here the contraction over the channels is a matrix product per GATHERED pixel,
//     P[q][t] = sum_c dz[q][c] * w[t][c]        ([pixels x 64] . [64 x 27 -> 32], v_mfma_f32_32x32x16_bf16)
// and dx is a fold of P over the taps (col2im) inside the block:
//   * a block owns 4 x 8 x 8 (2-D: 1 x 16 x 16) output pixels and walks the (4+2) x (8+2) x (8+2) gathered pixels they
//     read in groups of 32 = one MFMA row block per wave and trip; the A fragments come STRAIGHT from global memory
//     (lane = pixel, half-wave = which 8 of a k-sub's 16 channels: four 16-byte loads per lane cover its 128-byte row);
//     pixels outside dz load zeros (= the padding of the transposed gather);
//   * the weights stay fp32 in effect: w = h + m + l, three bf16 terms (24 bits), three MFMAs per k-sub (the matrix
//     work is nothing here: 0.05 ms of MFMA time at 128^3);
//   * P goes to LDS as [tap][gathered pixel] (consecutive threads = consecutive x: conflict-free), then thread = output
//     pixel adds its 27 entries and stores one float.
// dz is read 2.3 x (the tiles' halos, mostly L2 hits) instead of 27 x.
// ---------------------------------------------------------------------------
constexpr int TC_MAXQ = 608;                                // 6 x 10 x 10 gathered pixels, rounded up to 19 x 32
struct TcGrid { int tz, ty, tx, tiles_z, tiles_y, tiles_x; };

__global__ __launch_bounds__(256, 2) void thin_cout1_mfma_bf16_kernel(const GatherConv p, const TcGrid tg) {
  extern __shared__ __attribute__((aligned(16))) float Pl[];          // [T][TC_MAXQ]
  const Phase& ph = p.ph[0];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int T = ph.nz * ph.ny * ph.nx;
  int t = blockIdx.x;
  const int bx = t % tg.tiles_x; t /= tg.tiles_x;
  const int by = t % tg.tiles_y; t /= tg.tiles_y;
  const int bz = t % tg.tiles_z;
  const int n = t / tg.tiles_z;
  const int oz0 = bz * tg.tz, oy0 = by * tg.ty, ox0 = bx * tg.tx;
  // gathered coordinates of tap j of output o: o + d0 + dstep * j (stride 1); the patch starts at the smallest one
  const int mnz = ph.dz0 + (p.dstep[0] < 0 ? (ph.nz - 1) * p.dstep[0] : 0), mny = ph.dy0 + (p.dstep[1] < 0 ? (ph.ny - 1) * p.dstep[1] : 0),
            mnx = ph.dx0 + (p.dstep[2] < 0 ? (ph.nx - 1) * p.dstep[2] : 0);
  const int PZ = tg.tz + ph.nz - 1, PY = tg.ty + ph.ny - 1, PX = tg.tx + ph.nx - 1;
  const int NQ = PZ * PY * PX;

  // B fragments: column li = tap (jz, jy, jx) in walk order, k = channel; hi and lo halves of the fp32 weight
  bf16x8 whi[4], wmi[4], wlo[4];
  {
    const bool tv = li < T;
    const int jx = tv ? li % ph.nx : 0, tq = tv ? li / ph.nx : 0, jy = tq % ph.ny, jz = tq / ph.ny;
    const int kz = ph.kz0 + p.kstep[0] * jz, ky = ph.ky0 + p.kstep[1] * jy, kx = ph.kx0 + p.kstep[2] * jx;
    const float* wrow = p.wp + (long)((kz * p.Ky + ky) * p.Kx + kx) * p.Cin;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = tv ? wrow[16 * s + 8 * lh + e] : 0.f;
        const __bf16 h = (__bf16)w;
        const __bf16 m = (__bf16)(w - (float)h);
        whi[s][e] = h;
        wmi[s][e] = m;
        wlo[s][e] = (__bf16)(w - (float)h - (float)m);
      }
  }
  const char* ginb = reinterpret_cast<const char*>(p.in);
  for (int rb = wid; rb * 32 < NQ; rb += 4) {
    const int q = rb * 32 + li;
    const int pz = q / (PY * PX), rem = q - pz * (PY * PX), py = rem / PX, px = rem - py * PX;
    const int iz = oz0 + mnz + pz, iy = oy0 + mny + py, ix = ox0 + mnx + px;
    const bool ok = q < NQ && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    const char* src = ginb + ((((long)n * p.Di + iz) * p.Hi + iy) * p.Wi + ix) * p.ldi * 2 + 16 * lh;
    bf16x8 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (ok) a[s] = *reinterpret_cast<const bf16x8*>(src + 32 * s);
      else
#pragma unroll
        for (int e = 0; e < 8; ++e) a[s][e] = (__bf16)0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], wlo[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], wmi[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], whi[s], acc, 0, 0, 0);
    }
    // acc[r] of lane (li, lh): column = tap li, row = gathered pixel rb*32 + (r & 3) + 8 (r >> 2) + 4 lh
    if (li < T) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Pl + li * TC_MAXQ + rb * 32 + 8 * g + 4 * lh) =
            make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
  }
  __syncthreads();
  // fold: thread = output pixel of the tile
  const int lx = tid % tg.tx, lq = tid / tg.tx, ly = lq % tg.ty, lz = lq / tg.ty;
  const int oz = oz0 + lz, oy = oy0 + ly, ox = ox0 + lx;
  if (lz < tg.tz && oz < ph.Mz && oy < ph.My && ox < ph.Mx) {
    float sum = 0.f;
    int j = 0;
    for (int jz = 0; jz < ph.nz; ++jz) {
      const int qz = lz + ph.dz0 + p.dstep[0] * jz - mnz;
      for (int jy = 0; jy < ph.ny; ++jy) {
        const int qy = ly + ph.dy0 + p.dstep[1] * jy - mny;
        for (int jx = 0; jx < ph.nx; ++jx, ++j) {
          const int qx = lx + ph.dx0 + p.dstep[2] * jx - mnx;
          sum += Pl[j * TC_MAXQ + (qz * PY + qy) * PX + qx];
        }
      }
    }
    p.out[(long)(((n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.ldo] = sum;
  }
}

// Serves this launch?  (64 bf16 channels -> one fp32 channel, stride 1, one phase, at most 3 taps per dimension.)
bool thin_cout1_mfma_bf16_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_TC_MFMA") != nullptr;
  if (off || !p.in_bf16 || p.Cout != 1 || p.Cin != 64 || p.nphase != 1 || p.bias || p.resid || p.tanh_out || p.stats ||
      p.pro.scale || p.ldi % 8 != 0 || (reinterpret_cast<uintptr_t>(p.in) & 15))
    return false;
  const Phase& ph = p.ph[0];
  if (ph.nz < 1 || ph.ny < 1 || ph.nx < 1 || ph.nz > 3 || ph.ny > 3 || ph.nx > 3) return false;
  for (int d = 0; d < 3; ++d)
    if (p.istride[d] != 1 || p.ostride[d] != 1 || (p.dstep[d] != 1 && p.dstep[d] != -1)) return false;
  if (ph.oz || ph.oy || ph.ox || ph.Mz != p.Do || ph.My != p.Ho || ph.Mx != p.Wo) return false;
  return (long)p.N * p.Do * p.Ho * p.Wo * p.ldo < (1L << 31);
}

int launch_thin_cout1_mfma_bf16(const GatherConv& p, hipStream_t st) {
  const Phase& ph = p.ph[0];
  TcGrid tg;
  if (ph.nz == 1) { tg.tz = 1; tg.ty = 16; tg.tx = 16; }        // 2-D: 18 x 18 = 324 gathered pixels
  else { tg.tz = 4; tg.ty = 8; tg.tx = 8; }                      // 3-D: 6 x 10 x 10 = 600
  tg.tiles_z = (ph.Mz + tg.tz - 1) / tg.tz;
  tg.tiles_y = (ph.My + tg.ty - 1) / tg.ty;
  tg.tiles_x = (ph.Mx + tg.tx - 1) / tg.tx;
  const int T = ph.nz * ph.ny * ph.nx;
  const int smem = T * TC_MAXQ * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(thin_cout1_mfma_bf16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 27 * TC_MAXQ * (int)sizeof(float));
    if (e != hipSuccess) {
      set_error("thin_cout1_mfma_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const long blocks = (long)p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  MPGAN_CHECK_ARG(blocks < (1L << 31), "thin_cout1_mfma_bf16: too many tiles");
  hipLaunchKernelGGL(thin_cout1_mfma_bf16_kernel, dim3((unsigned)blocks), dim3(256), smem, st, p, tg);
  return check_launch("thin_cout1_mfma_bf16");
}

// ---------------------------------------------------------------------------
// HBM-bound helpers of the bf16 path
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ld8(const __bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(__bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[e];
  *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// a = LeakyReLU(z*scale + shift): the layer's activation, materialised once (thread = 8 channels of one pixel)
template <typename TO>
__global__ __launch_bounds__(256) void norm_act_bf16_kernel(const __bf16* __restrict__ z, int ldz,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float slope, long rows,
                                                            int C, TO* __restrict__ out, int ldo) {
  const int CG = C / 8;
  const long total = rows * CG;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / CG;
    const int c = (int)(i - row * CG) * 8;
    float v[8], sc[8], sh[8];
    ld8(z + row * ldz + c, v);
    ld8(scale + c, sc);
    ld8(shift + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float y = v[e] * sc[e] + sh[e];
      v[e] = y < 0.f ? y * slope : y;
    }
    st8(out + row * ldo + c, v);
  }
}

// BatchNorm + LeakyReLU backward on bf16 z:  gy = g*act'(y);  partial rows [chunks*n][3][C] of
// (sum gy, sum gy*zhat, 0) in the layout mpgan_norm_bwd_finalize consumes; apply: dz = scale*(gy - c1 - zhat*c2).
// block = R rows x C/8 column groups; chunk = blockIdx.x of gridDim.x over the rows.
template <typename TG>
__global__ __launch_bounds__(256) void norm_bwd_reduce_bf16_kernel(const TG* __restrict__ g, int ldg,
                                                                   const __bf16* __restrict__ z, int ldz,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float slope,
                                                                   long rows, int C, float* __restrict__ partials) {
  extern __shared__ float red[];   // [R][2][C]
  const int CG = C / 8, R = 256 / CG;
  const int q = threadIdx.x % CG, r = threadIdx.x / CG, c = q * 8;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long beg = (long)blockIdx.x * per, end = beg + per < rows ? beg + per : rows;
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
  if (r < R) {
    float sc[8], sh[8], mu[8], is[8];
    ld8(scale + c, sc); ld8(shift + c, sh); ld8(mean + c, mu); ld8(invstd + c, is);
    for (long row = beg + r; row < end; row += R) {
      float zv[8], gv[8];
      ld8(z + row * ldz + c, zv);
      ld8(g + row * ldg + c, gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = zv[e] * sc[e] + sh[e];
        const float gy = y < 0.f ? gv[e] * slope : gv[e];
        a0[e] += gy;
        a1[e] = fmaf(gy, (zv[e] - mu[e]) * is[e], a1[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(r * 2 + 0) * C + c + e] = a0[e];
      red[(r * 2 + 1) * C + c + e] = a1[e];
    }
  }
  __syncthreads();
  float* out = partials + (long)blockIdx.x * 3 * C;
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    float s = 0.f;
    if (i < 2 * C)
      for (int rr = 0; rr < R; ++rr) s += red[rr * 2 * C + i];
    out[i] = s;
  }
}

// dz (bf16) = scale*(gy - c1 - zhat*c2); optionally per-block column sums of the ROUNDED dz (the conv's bias
// gradient: dbias = colsum(dy)) as partial rows [gridDim.x][C].
template <typename TG>
__global__ __launch_bounds__(256) void norm_bwd_apply_bf16_kernel(const TG* __restrict__ g, int ldg,
                                                                  const __bf16* __restrict__ z, int ldz,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ c1,
                                                                  const float* __restrict__ c2, float slope, long rows,
                                                                  int C, __bf16* __restrict__ dz, int lddz,
                                                                  float* __restrict__ bias_partials) {
  extern __shared__ float red[];   // [R][C]
  const int CG = C / 8, R = 256 / CG;
  const int q = threadIdx.x % CG, r = threadIdx.x / CG, c = q * 8;
  float bs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bs[e] = 0.f;
  if (r < R) {
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
    ld8(scale + c, sc); ld8(shift + c, sh); ld8(mean + c, mu); ld8(invstd + c, is); ld8(c1 + c, k1); ld8(c2 + c, k2);
    for (long row = (long)blockIdx.x * R + r; row < rows; row += (long)gridDim.x * R) {
      float zv[8], gv[8], o[8];
      ld8(z + row * ldz + c, zv);
      ld8(g + row * ldg + c, gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = zv[e] * sc[e] + sh[e];
        const float gy = y < 0.f ? gv[e] * slope : gv[e];
        o[e] = sc[e] * (gy - k1[e] - (zv[e] - mu[e]) * is[e] * k2[e]);
        bs[e] += (float)(__bf16)o[e];
      }
      st8(dz + row * lddz + c, o);
    }
  }
  if (bias_partials) {
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[r * C + c + e] = bs[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      float s = 0.f;
      for (int rr = 0; rr < R; ++rr) s += red[rr * C + i];
      bias_partials[(long)blockIdx.x * C + i] = s;
    }
  }
}

// table-driven repack of fp32 master weights into bf16 [Cout][tap][Cin] / [Cin][tap][Cout] (pack_weights_kernel's twin)
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                const int64_t* __restrict__ table) {
  const int64_t* e = table + (long)blockIdx.y * 8;
  const long so = e[0], dof = e[1];
  const int Cout = (int)e[2], Cin = (int)e[3], T = (int)e[4];
  const int transposed = (int)e[5], layout = (int)e[6];
  const long total = (long)Cout * Cin * T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // i walks the DESTINATION: layout 0 = [co][t][ci], 1 = [ci][t][co]
    int co, ci, t;
    if (layout == 0) { ci = (int)(i % Cin); const long r = i / Cin; t = (int)(r % T); co = (int)(r / T); }
    else { co = (int)(i % Cout); const long r = i / Cout; t = (int)(r % T); ci = (int)(r / T); }
    const long s = transposed ? ((long)ci * Cout + co) * T + t : ((long)co * Cin + ci) * T + t;
    dst[dof + i] = (__bf16)src[so + s];
  }
}

static inline int hb_ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace mpgan

using namespace mpgan;

// Geometry of a bf16 gather.  A stride-1 transposed gather (backward-data of a conv) is first built as ONE phase: the
// patch kernels stage out-of-image rows as zeros once per tile and have no use for border classes; only when neither
// patch form serves it is it rebuilt with them (conv_geom.h) for the K-stepped kernels.
static void build_gather_bf16(GatherConv& p, const mpgan_conv_geom* g, bool backward_data) {
  const bool fwd_type = backward_data ? g->transposed != 0 : g->transposed == 0;
  const int32_t *gd = backward_data ? g->out_dhw : g->in_dhw, *pd = backward_data ? g->in_dhw : g->out_dhw;
  const int cg = backward_data ? g->cout : g->cin, cp = backward_data ? g->cin : g->cout;
  if (fwd_type) {
    build_forward(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad);
    return;
  }
  build_transposed(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad, false);
  GatherConv c = p;
  c.ldi = c.Cin;
  if (c.Cin % HB_BK == 0 && (hp8_ok(c) || hp_ok(c))) return;
  build_transposed(p, g->n, gd, cg, pd, cp, g->k, g->stride, g->pad, true);
}

extern "C" int32_t mpgan_conv_stats_rows_bf16(const mpgan_conv_geom* g) {
  if (check_geom(g)) return -1;
  GatherConv p{};
  set_geom_flags(p, g);
  build_gather_bf16(p, g, false);
  if (hp8_use(p, true)) {    // big-patch form: one row per 8 x 8 x 8 tile
    const HpGrid tg = hp8_grid(p);
    return p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  }
  if (hp_use(p, true)) {     // patch form: one row per 4 x 8 x 8 tile
    const HpGrid tg = hp_grid(p);
    return p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  }
  p.ldi = p.Cin;
  const int bm = hw_bm(g->cin % HB_BK == 0 ? hw_choice(p, true) : 0);
  return (int32_t)phase_tile_rows(p, bm);
}

// Which bf16 kernel serves this geometry (profiling labels): 0 = K-stepped gather_conv_bf16_kernel,
// 1 = gather_patch_bf16_kernel (stride-1 3x3x3 gathers), 2 / 3 / 4 = gather_conv_bf16_wide_kernel 256 x 256 / 512 x 128 /
// 256 x 256 over phase pairs, 5 = gather_patch8_bf16_kernel (stride-1 3x3x3 gathers on maps of >= 16 pixels per dimension).
extern "C" int32_t mpgan_conv_variant_bf16(const mpgan_conv_geom* g, int32_t backward_data) {
  if (check_geom(g)) return -1;
  GatherConv p{};
  set_geom_flags(p, g);
  if (!backward_data) {
    build_gather_bf16(p, g, false);
  } else {
    build_gather_bf16(p, g, true);
  }
  if (hp8_use(p, !backward_data)) return 5;
  if (hp_use(p, !backward_data)) return 1;
  p.ldi = p.Cin;
  const int wide = p.Cin % HB_BK == 0 ? hw_choice(p, !backward_data) : 0;
  return wide ? 1 + wide : 0;
}

extern "C" int mpgan_conv_forward_bf16(const mpgan_conv_geom* g, const void* x, int32_t ldx, const void* w_packed,
                                       const float* bias, float* stats_partials, void* y, int32_t ldy, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(x && w_packed && y, "conv_forward_bf16: null pointer");
  MPGAN_CHECK_ARG(ldx >= g->cin && ldy >= g->cout, "conv_forward_bf16: bad pitch");
  GatherConv p{};
  p.in = static_cast<const float*>(x); p.wp = static_cast<const float*>(w_packed); p.out = static_cast<float*>(y);
  p.bias = bias; p.stats = stats_partials;
  p.pro = make_pro(nullptr);
  p.ldi = ldx; p.ldo = ldy;
  set_geom_flags(p, g);
  build_gather_bf16(p, g, false);
  return hb_dispatch(p, (hipStream_t)stream, "conv_forward_bf16");
}

extern "C" int mpgan_conv_backward_data_bf16(const mpgan_conv_geom* g, const void* dy, int32_t lddy,
                                             const void* w_packed_bwd, void* dx, int32_t lddx, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(dy && w_packed_bwd && dx, "conv_backward_data_bf16: null pointer");
  MPGAN_CHECK_ARG(lddy >= g->cout && lddx >= g->cin, "conv_backward_data_bf16: bad pitch");
  GatherConv p{};
  p.in = static_cast<const float*>(dy); p.wp = static_cast<const float*>(w_packed_bwd); p.out = static_cast<float*>(dx);
  p.pro = make_pro(nullptr);
  p.ldi = lddy; p.ldo = lddx;
  set_geom_flags(p, g);
  build_gather_bf16(p, g, true);
  return hb_dispatch(p, (hipStream_t)stream, "conv_backward_data_bf16");
}

// Rows of norm-backward partial sums mpgan_conv_backward_data_stats_bf16 leaves for this geometry: one per (phase or
// phase pair, m-tile) of the patch / wide form that serves it; 0 = served by the narrow K-stepped kernel, which has no
// fused sums (run mpgan_norm_bwd_reduce_bf16).
extern "C" int32_t mpgan_conv_bwd_stats_rows_bf16(const mpgan_conv_geom* g) {
  if (check_geom(g)) return -1;
  GatherConv p{};
  set_geom_flags(p, g);
  build_gather_bf16(p, g, true);
  if (p.Cin % HB_BK != 0 || p.Cout % 8 != 0) return 0;
  p.ldi = p.Cin;
  if (hp8_use(p, false)) {
    const HpGrid tg = hp8_grid(p);
    return p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  }
  if (hp_use(p, false)) {
    const HpGrid tg = hp_grid(p);
    return p.N * tg.tiles_z * tg.tiles_y * tg.tiles_x;
  }
  const int wide = hw_choice(p, false);
  if (wide == 0) return 0;
  if (wide == 3) return (int32_t)((max_phase_pixels(p) + 255) / 256) * (p.nphase / 2);
  return (int32_t)phase_tile_rows(p, hw_bm(wide));
}

// mpgan_conv_backward_data_bf16 + the reduce pass of the BatchNorm + LeakyReLU(slope) in front of this conv's input, in
// one launch: dx (bf16) is the gradient w.r.t. a = act(scale * z + shift); partials[rows][3][cin] receive the sums
// mpgan_norm_bwd_reduce_bf16 would form from the stored dx and z (feed mpgan_norm_bwd_finalize with n = 1, chunks = rows).
extern "C" int mpgan_conv_backward_data_stats_bf16(const mpgan_conv_geom* g, const void* dy, int32_t lddy,
                                                   const void* w_packed_bwd, void* dx, int32_t lddx, const void* z,
                                                   int32_t ldz, const float* scale, const float* shift, const float* mean,
                                                   const float* invstd, float slope, float* partials, void* stream) {
  int rc = check_geom(g);
  if (rc) return rc;
  MPGAN_CHECK_ARG(dy && w_packed_bwd && dx && z && scale && shift && mean && invstd && partials,
                  "conv_backward_data_stats_bf16: null pointer");
  MPGAN_CHECK_ARG(lddy >= g->cout && lddx >= g->cin && ldz >= g->cin, "conv_backward_data_stats_bf16: bad pitch");
  MPGAN_UNSUPPORTED(ldz % 8 != 0 || (reinterpret_cast<uintptr_t>(z) & 15) != 0, "conv_backward_data_stats_bf16: z pitch %% 8, 16-byte aligned");
  MPGAN_UNSUPPORTED(mpgan_conv_bwd_stats_rows_bf16(g) <= 0, "conv_backward_data_stats_bf16: this geometry has no fused sums "
                                                            "(mpgan_conv_bwd_stats_rows_bf16() == 0)");
  GatherConv p{};
  p.in = static_cast<const float*>(dy); p.wp = static_cast<const float*>(w_packed_bwd); p.out = static_cast<float*>(dx);
  p.pro = make_pro(nullptr);
  p.ldi = lddy; p.ldo = lddx;
  p.bwd.z = static_cast<const float*>(z); p.bwd.ldz = ldz;
  p.bwd.scale = scale; p.bwd.shift = shift; p.bwd.mean = mean; p.bwd.invstd = invstd;
  p.bwd.part = partials; p.bwd.leaky = 1; p.bwd.slope = slope;
  set_geom_flags(p, g);
  build_gather_bf16(p, g, true);
  {   // the rows were sized on the compact geometry: a pitched operand must not change the form
    GatherConv c = p;
    c.ldi = c.Cin;
    const bool a8 = hp8_use(c, false), a4 = !a8 && hp_use(c, false);
    const bool b8 = hp8_use(p, false), b4 = !b8 && hp_use(p, false);
    MPGAN_UNSUPPORTED(a8 != b8 || a4 != b4 || (!a8 && !a4 && hw_choice(c, false) != hw_choice(p, false)),
                      "conv_backward_data_stats_bf16: the channel pitch of dy changes the kernel form the partial rows were "
                      "sized for: pass a compact tensor");
  }
  return hb_dispatch(p, (hipStream_t)stream, "conv_backward_data_stats_bf16");
}

static void wgrad_hb_dims(const mpgan_conv_geom* g, int& Cd, int& Cg, int& T, long& M) {
  T = g->k[0] * g->k[1] * g->k[2];
  Cd = g->cout; Cg = g->cin;
  M = (long)g->n * g->out_dhw[0] * g->out_dhw[1] * g->out_dhw[2];
}

// Which kernel serves this layer's bf16 weight gradient (profiling labels): 0 = wgrad_bf16_kernel (128 x 256 tiles),
// 1 = wgrad_bf16_wide_kernel (256 x 256).
extern "C" int32_t mpgan_conv_wgrad_variant_bf16(const mpgan_conv_geom* g) {
  if (!g) return -1;
  int Cd, Cg, T;
  long M;
  wgrad_hb_dims(g, Cd, Cg, T, M);
  return plan_wgrad_hb(Cd, T * Cg, M).wide;
}

extern "C" int64_t mpgan_conv_wgrad_workspace_bf16(const mpgan_conv_geom* g) {
  if (!g) return -1;
  int Cd, Cg, T;
  long M;
  wgrad_hb_dims(g, Cd, Cg, T, M);
  const WgradHbPlan pl = plan_wgrad_hb(Cd, T * Cg, M);
  return (int64_t)pl.nsplit * Cd * T * Cg * (int64_t)sizeof(float);
}

extern "C" int mpgan_conv_backward_weight_bf16(const mpgan_conv_geom* g, const void* x, int32_t ldx, const void* dy,
                                               int32_t lddy, float* dw, float beta, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
  MPGAN_CHECK_ARG(g && x && dy && dw && workspace, "conv_backward_weight_bf16: null pointer");
  MPGAN_UNSUPPORTED(g->transposed, "conv_backward_weight_bf16: ConvNd only");
  MPGAN_UNSUPPORTED(g->pad[0] | g->pad[1] | g->pad[2], "conv_backward_weight_bf16: pad-free convs only");
  MPGAN_CHECK_ARG(ldx >= g->cin && lddy >= g->cout, "conv_backward_weight_bf16: bad pitch");
  MPGAN_UNSUPPORTED(g->cin % 8 || g->cout % 8 || ldx % 8 || lddy % 8 ||
                        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15),
                    "conv_backward_weight_bf16: channels / pitches %% 8 and 16-byte aligned operands");
  int Cd, Cg, T;
  long M;
  wgrad_hb_dims(g, Cd, Cg, T, M);
  MPGAN_CHECK_ARG(M < (1L << 31) - 64, "conv_backward_weight_bf16: more than 2^31 pixels");
  MPGAN_UNSUPPORTED(M * lddy * 2 >= (1L << 32) ||
                        (long)g->n * g->in_dhw[0] * g->in_dhw[1] * g->in_dhw[2] * ldx * 2 >= (1L << 32),
                    "conv_backward_weight_bf16: operand of 4 GiB or more");
  const WgradHbPlan pl = plan_wgrad_hb(Cd, T * Cg, M);
  const int64_t need = (int64_t)pl.nsplit * Cd * T * Cg * (int64_t)sizeof(float);
  MPGAN_CHECK_ARG(workspace_bytes >= need, "conv_backward_weight_bf16: workspace %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
  WgradHb p{};
  p.dense = static_cast<const char*>(dy); p.ldd = lddy; p.Cd = Cd;
  p.gath = static_cast<const char*>(x); p.ldg = ldx; p.Cg = Cg;
  p.partial = static_cast<float*>(workspace);
  p.N = g->n;
  p.Mz = g->out_dhw[0]; p.My = g->out_dhw[1]; p.Mx = g->out_dhw[2];
  p.Gz = g->in_dhw[0]; p.Gy = g->in_dhw[1]; p.Gx = g->in_dhw[2];
  p.Kz = g->k[0]; p.Ky = g->k[1]; p.Kx = g->k[2];
  p.sz = g->stride[0]; p.sy = g->stride[1]; p.sx = g->stride[2];
  p.nsplit = pl.nsplit; p.chunk = pl.chunk; p.tiles_c = pl.tiles_c; p.tiles_d = pl.tiles_d;
  p.fMx = make_fastdiv(p.Mx); p.fMy = make_fastdiv(p.My); p.fMz = make_fastdiv(p.Mz);
  static int nw = 0;
  if (!nw) {
    const char* e = dev_env("MPGAN_DBG_HB_NW");      // development: force four or eight waves per block
    nw = (e && atoi(e) == 4) ? 4 : 8;
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, WH_SMEM);
    hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<8>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, WH_SMEM);
    if (e4 != hipSuccess || e8 != hipSuccess) {
      nw = 0;
      set_error("wgrad_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e4 != hipSuccess ? e4 : e8));
      return MPGAN_ERR_HIP;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)pl.tiles_c * pl.tiles_d * pl.nsplit);
  if (pl.wide) {
    static bool wide_attr = false;
    if (!wide_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_wide_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WW_SMEM);
      if (e != hipSuccess) {
        set_error("wgrad_bf16_wide: hipFuncSetAttribute(%d): %s", WW_SMEM, hipGetErrorString(e));
        return MPGAN_ERR_HIP;
      }
      wide_attr = true;
    }
    hipLaunchKernelGGL(wgrad_bf16_wide_kernel, grid, dim3(512), WW_SMEM, st, p);
  } else if (nw == 4) hipLaunchKernelGGL(wgrad_bf16_kernel<4>, grid, dim3(256), WH_SMEM, st, p);
  else hipLaunchKernelGGL(wgrad_bf16_kernel<8>, grid, dim3(512), WH_SMEM, st, p);
  int rc = check_launch("wgrad_bf16");
  if (rc) return rc;
  const long total = (long)Cd * Cg * T;
  int blocks = (int)((total + 31) / 32);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, pl.nsplit, Cd, Cg, T, beta,
                     (const float*)nullptr, 0, (float*)nullptr, 0);
  return check_launch("wgrad_bf16_reduce");
}

extern "C" int mpgan_pack_weights_bf16(const float* flat_params, void* packed, const int64_t* table, int32_t n_entries,
                                       int64_t max_elems, void* stream) {
  MPGAN_CHECK_ARG(flat_params && packed && table && n_entries > 0 && max_elems > 0, "pack_weights_bf16: bad argument");
  long bx = (max_elems + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)bx, (unsigned)n_entries), dim3(256), 0,
                     (hipStream_t)stream, flat_params, static_cast<__bf16*>(packed), table);
  return check_launch("pack_weights_bf16");
}

extern "C" int mpgan_norm_act_bf16(const void* z, int32_t ldz, const float* scale, const float* shift, float slope,
                                   int64_t rows, int32_t c, void* out, int32_t ldo, int32_t out_f32, void* stream) {
  MPGAN_CHECK_ARG(z && scale && shift && out && rows > 0 && c > 0 && ldz >= c && ldo >= c, "norm_act_bf16: bad argument");
  MPGAN_UNSUPPORTED(c % 8 || ldz % 8 || ldo % 8 || ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out)) & 15),
                    "norm_act_bf16: channels / pitches %% 8, 16-byte aligned tensors");
  const int blocks = hb_ew_blocks(rows * (c / 8));
  if (out_f32)
    hipLaunchKernelGGL(norm_act_bf16_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const __bf16*>(z), ldz, scale, shift, slope, (long)rows, c, static_cast<float*>(out), ldo);
  else
    hipLaunchKernelGGL(norm_act_bf16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const __bf16*>(z), ldz, scale, shift, slope, (long)rows, c, static_cast<__bf16*>(out), ldo);
  return check_launch("norm_act_bf16");
}

extern "C" int32_t mpgan_norm_bwd_rows_bf16(int64_t rows, int32_t c) {
  if (c <= 0 || c % 8 || c / 8 > 256) return -1;
  const int R = 256 / (c / 8);
  long b = (rows + (long)R * 32 - 1) / ((long)R * 32);     // >= 32 rows per thread row
  if (b > 2048) b = 2048;
  return (int32_t)(b < 1 ? 1 : b);
}

extern "C" int mpgan_norm_bwd_reduce_bf16(const void* g, int32_t g_f32, int32_t ldg, const void* z, int32_t ldz,
                                          const float* scale, const float* shift, const float* mean,
                                          const float* invstd, float slope, int64_t rows, int32_t c, float* partials,
                                          void* stream) {
  MPGAN_CHECK_ARG(g && z && scale && shift && mean && invstd && partials && rows > 0 && ldg >= c && ldz >= c,
                  "norm_bwd_reduce_bf16: bad argument");
  const int nb = mpgan_norm_bwd_rows_bf16(rows, c);
  MPGAN_UNSUPPORTED(nb < 0 || ldg % 8 || ldz % 8, "norm_bwd_reduce_bf16: C %% 8, C <= 2048, pitches %% 8");
  const int R = 256 / (c / 8);
  const size_t smem = (size_t)R * 2 * c * sizeof(float);
  if (g_f32)
    hipLaunchKernelGGL(norm_bwd_reduce_bf16_kernel<float>, dim3(nb), dim3(256), smem, (hipStream_t)stream,
                       static_cast<const float*>(g), ldg, static_cast<const __bf16*>(z), ldz, scale, shift, mean, invstd,
                       slope, (long)rows, c, partials);
  else
    hipLaunchKernelGGL(norm_bwd_reduce_bf16_kernel<__bf16>, dim3(nb), dim3(256), smem, (hipStream_t)stream,
                       static_cast<const __bf16*>(g), ldg, static_cast<const __bf16*>(z), ldz, scale, shift, mean,
                       invstd, slope, (long)rows, c, partials);
  return check_launch("norm_bwd_reduce_bf16");
}

extern "C" int mpgan_norm_bwd_apply_bf16(const void* g, int32_t g_f32, int32_t ldg, const void* z, int32_t ldz,
                                         const float* scale, const float* shift, const float* mean, const float* invstd,
                                         const float* c1, const float* c2, float slope, int64_t rows, int32_t c, void* dz,
                                         int32_t lddz, float* bias_partials, void* stream) {
  MPGAN_CHECK_ARG(g && z && scale && shift && mean && invstd && c1 && c2 && dz && rows > 0 && ldg >= c && ldz >= c &&
                      lddz >= c,
                  "norm_bwd_apply_bf16: bad argument");
  const int nb = mpgan_norm_bwd_rows_bf16(rows, c);
  MPGAN_UNSUPPORTED(nb < 0 || ldg % 8 || ldz % 8 || lddz % 8, "norm_bwd_apply_bf16: C %% 8, C <= 2048, pitches %% 8");
  const int R = 256 / (c / 8);
  const size_t smem = bias_partials ? (size_t)R * c * sizeof(float) : 0;
  if (g_f32)
    hipLaunchKernelGGL(norm_bwd_apply_bf16_kernel<float>, dim3(nb), dim3(256), smem, (hipStream_t)stream,
                       static_cast<const float*>(g), ldg, static_cast<const __bf16*>(z), ldz, scale, shift, mean, invstd,
                       c1, c2, slope, (long)rows, c, static_cast<__bf16*>(dz), lddz, bias_partials);
  else
    hipLaunchKernelGGL(norm_bwd_apply_bf16_kernel<__bf16>, dim3(nb), dim3(256), smem, (hipStream_t)stream,
                       static_cast<const __bf16*>(g), ldg, static_cast<const __bf16*>(z), ldz, scale, shift, mean,
                       invstd, c1, c2, slope, (long)rows, c, static_cast<__bf16*>(dz), lddz, bias_partials);
  return check_launch("norm_bwd_apply_bf16");
}
