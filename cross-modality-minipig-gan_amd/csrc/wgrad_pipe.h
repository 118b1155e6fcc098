// Weight-gradient parameter block and the software-pipelined weight-gradient kernel as a template: conv_wgrad.hip
// instantiates the fp32-MFMA forms, conv_mm16.hip the forms with bf16 matrix operands (MPGAN_CONV_MM_BF16).
#pragma once
#include "mpgan_common.h"
#include <type_traits>

namespace mpgan {


typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
  const float* dense;
  const float* gath;
  float* partial;  // [split][Cd][T*Cg]
  float* bias_partial;  // [split][Cd] column sums of the dense operand (fused bias gradient) or null
  Pro pro;         // prologue on the gathered operand
  int ldd, Cd, ldg, Cg;
  int N, Mz, My, Mx;  // coarse grid
  int Gz, Gy, Gx;     // gathered tensor spatial dims
  int Kz, Ky, Kx;
  int sz, sy, sx, pz, py, px;
  int nsplit;
  long chunk;  // pixels per split (multiple of 32)
  int dense_bf16;  // thin kernel only: `dense` points at bf16 data (dy of D.conv1 in the bf16 path)
  int fold_rounds; // thin kernel only: rounds of its lane-row fold (0 / 1: one)
  int tiles_c, tiles_d;  // 1-D launch of tiles_c*tiles_d*nsplit blocks, XCD-remapped, column tile fastest
  FastDiv fMx, fMy, fMz;
};

struct WBlockId { int tc, td, split; };
__device__ __forceinline__ WBlockId wgrad_block_id(const WgradParams& p) {
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  WBlockId b;
  b.tc = (int)(w % (unsigned)p.tiles_c);
  const unsigned q = w / (unsigned)p.tiles_c;
  b.td = (int)(q % (unsigned)p.tiles_d);
  b.split = (int)(q / (unsigned)p.tiles_d);
  return b;
}

constexpr int WBK = 32;

// conv_mm16.hip
int launch_wgrad_mm16(const WgradParams& p, int BD, int BG, hipStream_t st, bool& handled);

// ---------------------------------------------------------------------------
// Software-pipelined variant (both operands 16-byte vectorisable, coarse grid
// at least 32 wide, BatchNorm-or-no prologue): the K-step is one basic block;
// tile kt+2 is loaded under the MFMAs of group 0, tile kt+1 goes to LDS under
// group 3 (two register stages), exactly as gather_conv_pipe_kernel does.
// ---------------------------------------------------------------------------
//   MM16: matrix operands rounded to bf16 (MPGAN_CONV_MM_BF16): the LDS tiles stay fp32 [pixel][channel]; a lane reads
//         the 8 pixels of its channel that one v_mfma_f32_32x32x16_bf16 k-sub wants (8 conflict-free ds_read_b32, as
//         many per K-step as the fp32 form's 16 two-pixel steps), rounds them and issues 2 instead of 16 MFMAs per tile.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
template <int BD, int BG, int TM, int TN, int WN, int PRO, bool PAD, bool MM16 = false>
__global__ __launch_bounds__(256) void wgrad_pipe_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int STAGE = WBK * (BD + BG);
  constexpr int DCH = BD / 4, GCH = BG / 4;
  constexpr int DLOADS = (WBK * DCH) / 256, GLOADS = (WBK * GCH) / 256;
  constexpr int DROWSTEP = 256 / DCH, GROWSTEP = 256 / GCH;
  constexpr int NMF = 4 * TM * TN;
  // Row table (2 x 32 entries behind the two tile stages): for each of the 32 pixels of a
  // K-step, the byte offset of its tap-(0,0,0) gathered pixel and (PAD) its gathered
  // coordinates / (!PAD) its validity.  Filled once per row and K-step under the MFMAs of
  // group 1; the 256 loaders then need one LDS read, one add and one mask per address.
  int4* rtab = reinterpret_cast<int4*>(lds + 2 * STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int T = p.Kz * p.Ky * p.Kx;
  const int NC = T * p.Cg;
  const WBlockId bid = wgrad_block_id(p);
  const int c0 = bid.tc * BG;
  const int d0 = bid.td * BD;
  const long M = (long)p.N * p.Mz * p.My * p.Mx;
  const long mbeg = (long)bid.split * p.chunk;
  const long mend = mbeg + p.chunk < M ? mbeg + p.chunk : M;
  const int nk = mbeg < mend ? (int)((mend - mbeg + WBK - 1) / WBK) : 0;
  const float slope = PRO ? pro_slope(p.pro) : 1.f;
  const int act = p.pro.act;
  const char* __restrict__ gdb = reinterpret_cast<const char*>(p.dense);   // base + unsigned 32-bit byte offsets
  const char* __restrict__ ggb = reinterpret_cast<const char*>(p.gath);    // (host: operands < 4 GiB)
  const int ldd = p.ldd, ldg = p.ldg, Gz = p.Gz, Gy = p.Gy, Gx = p.Gx;

  // gathered operand: this thread's column chunk (tap, channel) is fixed for the block
  const int gcc = tid % GCH, grow0 = tid / GCH;
  int gci, gkz, gky, gkx, gokm;
  {
    const int col = c0 + gcc * 4;
    gokm = col < NC ? -1 : 0;
    const int t = gokm ? col / p.Cg : 0;
    gci = gokm ? col - t * p.Cg : 0;
    gkx = t % p.Kx;
    const int q = t / p.Kx;
    gky = q % p.Ky;
    gkz = q / p.Ky;
  }
  const unsigned tapB = (unsigned)(((gkz * Gy + gky) * Gx + gkx) * ldg + gci) * 4u;
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRO != 0) {          // per-channel scale/shift of this thread's channels: loaded once
    psc = *reinterpret_cast<const float4*>(p.pro.scale + gci);
    psh = *reinterpret_cast<const float4*>(p.pro.shift + gci);
  }
  const int dcc = tid % DCH, drow0 = tid / DCH;
  const int dcol = d0 + dcc * 4;
  const int dokm = dcol < p.Cd ? -1 : 0;
  const int imend = (int)mend;
  int mrow = (int)mbeg;              // first pixel of the tile being loaded (32-bit: M < 2^31)
  unsigned dB = (unsigned)(((int)mbeg + drow0) * ldd + dcol) * 4u;
  const unsigned dstepB = (unsigned)(WBK * ldd) * 4u, drowB = (unsigned)(DROWSTEP * ldd) * 4u;

  // The tables are filled for consecutive tiles (mbeg, mbeg + 32, ...), so the entry of row r is carried as a cursor
  // -- coarse coordinates + byte offset, decoded with divisions ONCE -- and advanced by 32 pixels per fill as a
  // mixed-radix addition: 32 = dx0 + Mx (dy0 + My (dz0 + Mz dn0)) with block-uniform digits, one carry compare per
  // dimension (digit + carry stays below twice the radix), wave-uniform offset increments: 20 vector instructions
  // per K-step instead of the 38 of three multiply-high divisions, for any grid extent.
  int tx, ty, tz;
  unsigned toff;
  {
    unsigned q, ux, uy, uz;
    fdivmod((unsigned)((int)mbeg + (tid & 31)), p.fMx, q, ux);
    fdivmod(q, p.fMy, q, uy);
    fdivmod(q, p.fMz, q, uz);
    tx = (int)ux; ty = (int)uy; tz = (int)uz;
    toff = (unsigned)((((int)q * Gz + tz * p.sz - p.pz) * Gy + ty * p.sy - p.py) * Gx + tx * p.sx - p.px) * (unsigned)ldg * 4u;
  }
  unsigned dq, udx0, udy0, udz0;
  fdivmod((unsigned)WBK, p.fMx, dq, udx0);
  fdivmod(dq, p.fMy, dq, udy0);
  fdivmod(dq, p.fMz, dq, udz0);
  const int dx0 = (int)udx0, dy0 = (int)udy0, dz0 = (int)udz0;     // (dq: whole samples per 32 pixels)
  const unsigned incX = (unsigned)((dx0 * p.sx + (dy0 * p.sy + (dz0 * p.sz + (int)dq * Gz) * Gy) * Gx) * ldg) * 4u;
  const unsigned wrapX = (unsigned)((p.sy * Gx - p.Mx * p.sx) * ldg) * 4u;
  const unsigned wrapY = (unsigned)((p.sz * Gy * Gx - p.My * p.sy * Gx) * ldg) * 4u;
  const unsigned wrapZ = (unsigned)((Gz * Gy * Gx - p.Mz * p.sz * Gy * Gx) * ldg) * 4u;
  auto fill_table = [&](int buf, int mt) {
    const int r = tid & 31;          // 8 threads write the same entry with the same value
    const int iz0 = tz * p.sz - p.pz, iy0 = ty * p.sy - p.py, ix0 = tx * p.sx - p.px;
    const bool valid = mt + r < imend;
    int4 e;
    e.x = (int)toff;
    {                                // advance the cursor to the next tile
      tx += dx0;
      const bool cx = tx >= p.Mx;
      tx -= cx ? p.Mx : 0;
      ty += dy0 + (cx ? 1 : 0);
      const bool cy = ty >= p.My;
      ty -= cy ? p.My : 0;
      tz += dz0 + (cy ? 1 : 0);
      const bool cz = tz >= p.Mz;
      tz -= cz ? p.Mz : 0;
      toff += incX + (cx ? wrapX : 0u) + (cy ? wrapY : 0u) + (cz ? wrapZ : 0u);
    }
    if constexpr (PAD) {
      e.y = valid ? iz0 : -(1 << 28);
      e.z = iy0;
      e.w = ix0;
    } else {
      e.y = valid ? -1 : 0;
      e.z = 0;
      e.w = 0;
    }
    rtab[buf * 32 + r] = e;
  };

  struct Stage {
    float4 rd[DLOADS], rg[GLOADS];
    unsigned gmask, dmask;
  };
  Stage SX, SY;
  float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);

  auto issue_loads = [&](Stage& S, int buf) {
    unsigned dm = 0;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) {
      const int ok = (mrow + drow0 + DROWSTEP * i) < imend ? dokm : 0;
      S.rd[i] = *reinterpret_cast<const float4*>(gdb + ((dB + (unsigned)i * drowB) & (unsigned)ok));
      dm |= ((unsigned)ok & 1u) << i;
    }
    unsigned gm = 0;
#pragma unroll
    for (int i = 0; i < GLOADS; ++i) {
      const int4 e = rtab[buf * 32 + grow0 + GROWSTEP * i];
      int ok;
      if constexpr (PAD) {
        const int iz = e.y + gkz, iy = e.z + gky, ix = e.w + gkx;
        ok = ((unsigned)iz < (unsigned)Gz ? gokm : 0) & ((unsigned)iy < (unsigned)Gy ? -1 : 0) &
             ((unsigned)ix < (unsigned)Gx ? -1 : 0);
      } else {
        ok = e.y & gokm;
      }
      S.rg[i] = *reinterpret_cast<const float4*>(ggb + (((unsigned)e.x + tapB) & (unsigned)ok));
      gm |= ((unsigned)ok & 1u) << i;
    }
    S.gmask = gm;
    S.dmask = dm;
    dB += dstepB;
    mrow += WBK;
  };

  // FULL: the tile is known to lie wholly inside the block's pixel range (every K-step but the last two of a chunk):
  // the zero selects of both operands (32 v_cndmask + the mask bits of a K-step's 217 vector instructions, ISA count
  // of round 3's <128,128,..,3,false>) are dropped.  What they guarded is harmless there: a channel / column past
  // the operand's extent was LOADED from offset 0 (finite data) and only feeds result rows / columns the epilogue
  // discards; pad-free convs have no padding zeros to form.  Padded convs (PAD) keep the gathered operand's select.
  auto store_tile = [&](int buf, const Stage& S, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    float* Ds = lds + buf * STAGE;
    float* Gs = Ds + WBK * BD;
#pragma unroll
    for (int i = 0; i < DLOADS; ++i) {
      float4 v = S.rd[i];
      if constexpr (!FULL) {
        const bool ok = (S.dmask >> i) & 1u;     // rows past the chunk / channels past Cd
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      }
      bacc.x += v.x; bacc.y += v.y; bacc.z += v.z; bacc.w += v.w;
      *reinterpret_cast<float4*>(Ds + (drow0 + DROWSTEP * i) * BD + dcc * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < GLOADS; ++i) {
      float4 v = S.rg[i];
      if constexpr (PRO == 3) {          // LeakyReLU, host-known slope in [0, 1]: max(y, slope*y), exact
        v.x = v.x * psc.x + psh.x; v.y = v.y * psc.y + psh.y; v.z = v.z * psc.z + psh.z; v.w = v.w * psc.w + psh.w;
        v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope);
        v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
      } else if constexpr (PRO == 1) {
        v.x = act_apply(v.x * psc.x + psh.x, act, slope);
        v.y = act_apply(v.y * psc.y + psh.y, act, slope);
        v.z = act_apply(v.z * psc.z + psh.z, act, slope);
        v.w = act_apply(v.w * psc.w + psh.w, act, slope);
      }
      if constexpr (!FULL || PAD) {
        const bool ok = (S.gmask >> i) & 1u;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      }
      *reinterpret_cast<float4*>(Gs + (grow0 + GROWSTEP * i) * BG + gcc * 4) = v;
    }
  };
  constexpr std::integral_constant<bool, true> FULL_TILE{};
  constexpr std::integral_constant<bool, false> EDGE_TILE{};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  int mtab = (int)mbeg + 3 * WBK;     // tile whose table entry the next K-step writes
  if (nk > 0) {
    fill_table(0, (int)mbeg);
    fill_table(1, (int)mbeg + WBK);
    __syncthreads();
    issue_loads(SX, 0);
    store_tile(0, SX, EDGE_TILE);
    issue_loads(SX, 1);
    __syncthreads();                  // table 0 has been read by everyone
    fill_table(0, (int)mbeg + 2 * WBK);
  }
  __syncthreads();

  // K-step kt (LDS buffer cb = kt & 1): loads tile kt+2 through table[cb], writes the table of
  // tile kt+3 into table[cb ^ 1] (last read one barrier ago).
  auto step = [&](int cb, Stage& Sn, const Stage& Sp, auto full_tag) {
    const float* Ds = lds + cb * STAGE + wm * TM * 32 + li;
    const float* Gs = lds + cb * STAGE + WBK * BD + wn * TN * 32 + li;
    if constexpr (MM16) {
      // lane (channel li, half lh) of k-sub s holds pixels 16 s + 8 lh + j, j = 0..7, of its channel
      wg_bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int j = 0; j < 8; ++j) fa[s][tm][j] = (__bf16)Ds[(16 * s + 8 * lh + j) * BD + tm * 32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int j = 0; j < 8; ++j) fb[s][tn][j] = (__bf16)Gs[(16 * s + 8 * lh + j) * BG + tn * 32];
      }
      issue_loads(Sn, cb);
      fill_table(cb ^ 1, mtab);
      mtab += WBK;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][tm], fb[0][tn], acc[tm][tn], 0, 0, 0);
      store_tile(cb ^ 1, Sp, full_tag);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][tm], fb[1][tn], acc[tm][tn], 0, 0, 0);
      return;
    }
    float a[2][4][TM], b[2][4][TN];
    auto read_group = [&](int g, int slot) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = 2 * (4 * g + q) + lh;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[slot][q][tm] = Ds[kk * BD + tm * 32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[slot][q][tn] = Gs[kk * BG + tn * 32];
      }
    };
    read_group(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sl = g & 1;
      if (g < 3) read_group(g + 1, sl ^ 1);
      if (g == 0) issue_loads(Sn, cb);
      if (g == 1) { fill_table(cb ^ 1, mtab); mtab += WBK; }
      if (g == 3) store_tile(cb ^ 1, Sp, full_tag);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][q][tm], b[sl][q][tn], acc[tm][tn], 0, 0, 0);
      if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
      if (g == 0) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      } else if (g == 1) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);                     // row-table arithmetic
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      } else if (g == 3) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 9, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // K-step kt stores tile kt + 1: tiles 1 .. nk - 2 lie wholly inside the chunk (only the last tile can be ragged,
  // and the one behind it is stored as zeros nobody reads), so steps 0 .. nk - 3 store without selects
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {
    step(0, SY, SX, FULL_TILE);
    __syncthreads();
    step(1, SX, SY, FULL_TILE);
    __syncthreads();
  }
  for (; kt < nk; kt += 2) {
    step(0, SY, SX, EDGE_TILE);
    __syncthreads();
    if (kt + 1 < nk) {
      step(1, SX, SY, EDGE_TILE);
      __syncthreads();
    }
  }
  // tiles past the chunk were stored (as zeros) once more than consumed: bacc saw only zeros there

  if (p.bias_partial != nullptr && bid.tc == 0) {
    float* red = lds;
    *reinterpret_cast<float4*>(red + drow0 * BD + dcc * 4) = bacc;
    __syncthreads();
    if (tid < BD && d0 + tid < p.Cd) {
      float t = 0.f;
      for (int r = 0; r < DROWSTEP; ++r) t += red[r * BD + tid];
      p.bias_partial[(long)bid.split * p.Cd + d0 + tid] = t;
    }
  }
  float* out = p.partial + (long)bid.split * p.Cd * NC;
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

}  // namespace mpgan
