// Geometry of the gather-type convolutions shared by the fp32 (conv_igemm.hip) and bf16 (conv_bf16.hip)
// implicit-GEMM kernels: phases of a (transposed) gather, block-id decoding, host-side builders.
#pragma once
#include "mpgan_common.h"
#include "norm_fold.h"

namespace mpgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Phases of a gather: the 2^d sub-grids of a strided transposed gather (<= 8), or the 3^d border classes of a
// stride-1 transposed gather on a small map (build_transposed).
constexpr int MAX_PHASES = 27;

struct Phase {
  FastDiv fMx, fMy, fMz;
  int Mz, My, Mx;     // extents of the m-grid of this phase
  int oz, oy, ox;     // output coordinate = m*ostride + o?
  int nz, ny, nx;     // taps per dimension
  int kz0, ky0, kx0;  // kernel index of tap j: k0 + kstep*j
  int dz0, dy0, dx0;  // input offset of tap j:  d0 + dstep*j
};

// Backward-data launches of the K-stepped kernels: norm-backward partial sums of the gradient they produce.
// The epilogue holds g (the gradient w.r.t. the activation a = act(bn(z))) in registers; with z and the layer's
// norm vectors it leaves, per tile, sum(gy), sum(gy * zhat), sum(g * min(y, 0)) -- what norm_bwd_reduce would
// re-read g and z for (one [3][C] row per (phase, m-tile), any row order: the finalize adds them all).
struct BwdStats {
  const float* z;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  float* part;
  int ldz, leaky;
  float slope;
};

// Epilogue activation (eval-mode inference): with running-statistics BatchNorm the layer's
// affine is known before the launch, so the conv's epilogue emits the ACTIVATED tensor
//   y = prelu(acc * scale[c] + shift[c], slope[c]) (+ resid) (tanh)
// (the conv's own bias is folded into shift; slope[c] = 1 leaves a channel linear -- the residual half of a fused
// unit0 || residual conv).  scale == nullptr: off.  Never combined with fused statistics or norm-backward sums.
struct EpiAct {
  const float* scale;
  const float* shift;
  const float* slope;
};
__device__ __forceinline__ float epi_act1(float v, float sc, float sh, float sl) {
  v = fmaf(v, sc, sh);
  return v > 0.f ? v : v * sl;
}

struct GatherConv {
  const float* in;
  const float* wp;
  float* out;
  const float* bias;
  const float* resid;
  float* stats;   // optional [nphase*gridDim.x][2][Cout] partial sums of the output (fused BatchNorm statistics)
  Pro pro;
  int N, Di, Hi, Wi, Cin, ldi;
  int Do, Ho, Wo, Cout, ldo, ldr;
  int Kz, Ky, Kx;
  int ostride[3], istride[3], kstep[3], dstep[3];
  int nphase, tanh_out;
  int mtiles, ntiles;   // 1-D launch of ksplit*nphase*mtiles*ntiles blocks, XCD-remapped, n-tile fastest
  int phase_outer;      // block order: 1 = phase slowest, 0 = phases of an m-tile adjacent (see conv_block_id)
  int ksplit;           // > 1: each block covers a K slice and leaves raw partial sums in kpartial
  float* kpartial;      // [ksplit][N*Do*Ho*Wo][Cout]
  long long* stats_acc;  // fixed-point statistics accumulators [acc_rep][4][Cout] instead of `stats` rows (norm_fold.h)
  int acc_rep;
  NormFold fold;         // consumer side: fold the producer's accumulators into the prologue's scale / shift
  BwdStats bwd;          // see above (part == null: off)
  int in_bf16, out_bf16; // thin (VALU) kernels of the bf16 path: `in` / `out` point at bf16 data (weights stay fp32)
  int mm16;              // MPGAN_CONV_MM_BF16: matrix operands rounded to bf16 into LDS, bf16 MFMA, fp32 accumulation
  int min_blocks;        // the geometry's big-tile threshold (0: default), see mpgan_conv_geom
  EpiAct epi;            // epilogue activation (eval-mode inference); epi.scale == nullptr: off
  int classes;           // the phases are border classes of unequal size (build_transposed): launchers pack the tile list
  int packed;            // set by set_tile_grid: the grid holds only real (phase, m-tile) pairs, tile_start[] delimits the phases
  MPGAN_STAMP_FIELD      // development builds only (mpgan_common.h)
  Phase ph[MAX_PHASES];
  int tile_start[MAX_PHASES + 1];   // packed: first (phase, m-tile) pair = statistics row of phase i; past nphase = INT_MAX
  int group_start[9];               // packed: first tile of work group k (one eighth of EVERY phase), see decode_block
};

// `row`: index of the block's (phase, m-tile) pair = its row of fused statistics / norm-backward partial sums.
struct BlockId { int mt, nt, phase, split, row; };
__host__ __device__ __forceinline__ BlockId decode_block(const GatherConv& p, const unsigned w) {   // w: position in the work order
  BlockId b;
  if (p.packed) {
    // Border-class phases differ in size by up to 27 x and in K length (taps) by up to 3.4 x.  A grid of nphase x
    // max-tiles blocks would leave most of them empty; a list of the real tiles phase after phase would put all
    // 27-tap interior tiles on the one or two XCDs whose contiguous share of the work order (xcd_remap hands out
    // eighths) they fall into -- measured on variant B's 256 -> 512 backward-data: every class launched alone sums to
    // 34.4 ms, the phase-major list took 44.9 (and the nphase x max-tiles grid 79).  The list therefore runs through
    // eight GROUPS, group k holding the k-th eighth of every phase: each XCD's share has the same mix of tap counts,
    // tiles of a phase stay contiguous inside a group (same weights, neighbouring pixels in one L2).
    b.nt = (int)(w % (unsigned)p.ntiles);
    unsigned q = w / (unsigned)p.ntiles;
    const unsigned tot = (unsigned)p.mtiles;            // (packed: mtiles = all phases' tiles)
    b.split = (int)(q / tot);
    q -= (unsigned)b.split * tot;
    int k = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) k += q >= (unsigned)p.group_start[i] ? 1 : 0;
    unsigned r = q - (unsigned)p.group_start[k];        // position inside group k
    int ph = 0, mt = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < MAX_PHASES; ++i) {
      const unsigned tp = i < p.nphase ? (unsigned)(p.tile_start[i + 1] - p.tile_start[i]) : 0u;   // tiles of phase i
      const unsigned lo = (tp * (unsigned)k) >> 3, sz = ((tp * (unsigned)(k + 1)) >> 3) - lo;      // its k-th eighth
      if (!found && r < sz) { ph = i; mt = (int)(lo + r); found = true; }
      if (!found) r -= sz;
    }
    b.phase = ph;
    b.mt = mt;
    b.row = p.tile_start[ph] + mt;
    return b;
  }
  // n-tile fastest, then phase, then m-tile: the phases of a strided backward-data gather read the
  // SAME dy pixels (different taps), so they sit next to each other in the work order -- same XCD,
  // same time, one fetch into its L2 instead of one per phase.
  // When the whole weight tensor would crowd a 4 MiB L2 (each phase only touches its own taps'
  // share of it) the phases run one after another instead (phase_outer).
  b.nt = (int)(w % (unsigned)p.ntiles);
  unsigned q = w / (unsigned)p.ntiles;
  if (p.phase_outer) {
    b.mt = (int)(q % (unsigned)p.mtiles);
    q /= (unsigned)p.mtiles;
    b.phase = (int)(q % (unsigned)p.nphase);
    b.split = (int)(q / (unsigned)p.nphase);
  } else {
    b.phase = (int)(q % (unsigned)p.nphase);
    q /= (unsigned)p.nphase;
    b.mt = (int)(q % (unsigned)p.mtiles);
    b.split = (int)(q / (unsigned)p.mtiles);
  }
  b.row = b.phase * p.mtiles + b.mt;
  return b;
}
__device__ __forceinline__ BlockId conv_block_id(const GatherConv& p) {
  return decode_block(p, xcd_remap(blockIdx.x, gridDim.x));
}

inline long max_phase_pixels(const GatherConv& p) {
  long maxM = 0;
  for (int i = 0; i < p.nphase; ++i) {
    long m = (long)p.N * p.ph[i].Mz * p.ph[i].My * p.ph[i].Mx;
    if (m > maxM) maxM = m;
  }
  return maxM;
}

// (phase, m-tile) pairs of a launch with m-tiles of `bm` pixels = its blocks per n-tile and K split = its rows of fused
// statistics: every phase's own tile count when the phases are border classes, nphase x the largest phase's otherwise.
inline long phase_tile_rows(const GatherConv& p, int bm) {
  if (!p.classes) return (max_phase_pixels(p) + bm - 1) / bm * p.nphase;
  long t = 0;
  for (int i = 0; i < p.nphase; ++i) t += ((long)p.N * p.ph[i].Mz * p.ph[i].My * p.ph[i].Mx + bm - 1) / bm;
  return t;
}
// Launchers of the K-stepped kernels: q.mtiles / q.packed / q.tile_start for m-tiles of `bm` pixels; returns the
// number of (phase, m-tile) pairs (grid = that x ntiles x ksplit).
inline long set_tile_grid(GatherConv& q, int bm) {
  q.packed = 0;
  if (!q.classes) {
    q.mtiles = (int)((max_phase_pixels(q) + bm - 1) / bm);
    return (long)q.mtiles * q.nphase;
  }
  const char* only_s = dev_env("MPGAN_DBG_ONLY_PHASE");          // (make DEV=1: time one class at a time; read per call)
  const int only = only_s ? atoi(only_s) : -1;
  long t = 0;
  for (int i = 0; i <= MAX_PHASES; ++i) {
    q.tile_start[i] = i <= q.nphase ? (int)t : 0x7FFFFFFF;
    if (i < q.nphase && (only < 0 || only == i)) t += ((long)q.N * q.ph[i].Mz * q.ph[i].My * q.ph[i].Mx + bm - 1) / bm;
  }
  for (int i = q.nphase + 1; i <= MAX_PHASES; ++i) q.tile_start[i] = 0x7FFFFFFF;
  long gs = 0;                                            // group k = the k-th eighth of every phase (decode_block)
  for (int k = 0; k <= 8; ++k) {
    q.group_start[k] = (int)gs;
    if (k < 8)
      for (int i = 0; i < q.nphase; ++i) {
        const long tp = q.tile_start[i + 1] - q.tile_start[i];
        gs += ((tp * (k + 1)) >> 3) - ((tp * k) >> 3);
      }
  }
  q.packed = 1;
  q.mtiles = (int)t;
  return t;
}

// conv_bf16.hip: MFMA form of the C -> 1 gather over bf16 data (D.conv1's backward-data in the bf16 path)
bool thin_cout1_mfma_bf16_ok(const GatherConv& p);
int launch_thin_cout1_mfma_bf16(const GatherConv& p, hipStream_t st);

// ---- host-side geometry builders -------------------------------------------
inline int check_geom(const mpgan_conv_geom* g) {
  MPGAN_CHECK_ARG(g != nullptr, "conv: null geometry");
  MPGAN_CHECK_ARG(g->n > 0 && g->cin > 0 && g->cout > 0, "conv: bad n/cin/cout");
  MPGAN_CHECK_ARG((g->flags & ~MPGAN_CONV_MM_BF16) == 0 && g->min_blocks >= 0, "conv: unknown flags / negative min_blocks");
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

// the per-call dispatch knobs that travel with the geometry (include/mpgan_hip.h)
constexpr int FORM_MIN_BLOCKS_DEFAULT = 1024;
inline void set_geom_flags(GatherConv& p, const mpgan_conv_geom* g) {
  p.mm16 = (g->flags & MPGAN_CONV_MM_BF16) ? 1 : 0;
  p.min_blocks = g->min_blocks > 0 ? g->min_blocks : FORM_MIN_BLOCKS_DEFAULT;
}

// conv_mm16.hip: the MM16 instances of the K-stepped kernel (conv_pipe.h)
bool mm16_gather_ok(const GatherConv& p);
int launch_gather_mm16(const GatherConv& p, int variant, bool ksplit2, long maxM, hipStream_t st);

// forward-type gather: produced[o] = sum_k gathered[o*s - p + k] * W[k]
inline void build_forward(GatherConv& p, int n, const int32_t* gath_dhw, int cg, const int32_t* prod_dhw, int cp,
                          const int32_t* k, const int32_t* s, const int32_t* pad) {
  p.N = n; p.Di = gath_dhw[0]; p.Hi = gath_dhw[1]; p.Wi = gath_dhw[2]; p.Cin = cg;
  p.Do = prod_dhw[0]; p.Ho = prod_dhw[1]; p.Wo = prod_dhw[2]; p.Cout = cp;
  p.classes = 0; p.packed = 0;
  p.Kz = k[0]; p.Ky = k[1]; p.Kx = k[2];
  for (int d = 0; d < 3; ++d) { p.ostride[d] = 1; p.istride[d] = s[d]; p.kstep[d] = 1; p.dstep[d] = 1; }
  p.nphase = 1;
  Phase& ph = p.ph[0];
  ph.Mz = prod_dhw[0]; ph.My = prod_dhw[1]; ph.Mx = prod_dhw[2];
  ph.fMx = make_fastdiv(ph.Mx); ph.fMy = make_fastdiv(ph.My); ph.fMz = make_fastdiv(ph.Mz);
  ph.oz = ph.oy = ph.ox = 0;
  ph.nz = k[0]; ph.ny = k[1]; ph.nx = k[2];
  ph.kz0 = ph.ky0 = ph.kx0 = 0;
  ph.dz0 = -pad[0]; ph.dy0 = -pad[1]; ph.dx0 = -pad[2];
}

// transposed-type gather: produced[o] = sum_k gathered[(o + p - k)/s] * W[k]
inline void build_transposed(GatherConv& p, int n, const int32_t* gath_dhw, int cg, const int32_t* prod_dhw, int cp,
                             const int32_t* k, const int32_t* s, const int32_t* pad, bool classes_ok = true) {
  p.N = n; p.Di = gath_dhw[0]; p.Hi = gath_dhw[1]; p.Wi = gath_dhw[2]; p.Cin = cg;
  p.Do = prod_dhw[0]; p.Ho = prod_dhw[1]; p.Wo = prod_dhw[2]; p.Cout = cp;
  p.Kz = k[0]; p.Ky = k[1]; p.Kx = k[2];
  for (int d = 0; d < 3; ++d) { p.ostride[d] = s[d]; p.istride[d] = 1; p.kstep[d] = s[d]; p.dstep[d] = -1; }
  // Stride 1 on a SMALL map (the backward-data of a valid k^3 conv: produced extent = gathered extent + k - 1): the
  // border rows of the produced grid have fewer real taps than k -- on the 10^3 map of variant B's 256 -> 512 conv
  // only (8/10)^3 = 51 % of the (tap, pixel) pairs are real, the others multiply masked zeros.  Split each
  // dimension into <= 3 classes of produced coordinates -- left border, interior, right border -- that share a tap
  // RANGE, and make each combination a phase (ostride = 1, the class start as the phase's output offset, its tap range
  // as (k0, n)): border classes issue 2 of 3 taps, 79 % of the issued pairs are real.  Taps that are still out of
  // range for single rows of a border class are masked exactly as before: the result is unchanged.
  // Only where the K-stepped kernels serve the gather (>= 64 channels on both sides): the patch kernels stage the
  // padding as zeros once per tile and need one phase.
  p.classes = 0;
  if (classes_ok && !dev_env("MPGAN_DBG_NO_CLASSES") && s[0] == 1 && s[1] == 1 && s[2] == 1 && cg >= 64 && cp >= 64) {
    int nc[3], lo[3][3], hi[3][3], k0c[3][3], njc[3][3];
    double real = 1.0, issued = 1.0, issued_cls = 1.0;
    for (int d = 0; d < 3; ++d) {
      const int G = gath_dhw[d], P = prod_dhw[d], K = k[d], pd = pad[d];
      // produced o reads gathered o + pd - kk: left border o < K - 1 - pd (tap K-1 out of range), right border
      // o > G - 1 + pd - ... i.e. o + pd > G - 1 (tap 0 out of range)
      int a = K - 1 - pd;            // first interior coordinate
      int b = G - pd;                // first right-border coordinate
      if (a < 0) a = 0;
      if (b > P) b = P;
      if (b < a) { a = 0; b = P; }   // the borders overlap (map smaller than the kernel): one class, all taps
      nc[d] = 0;
      const int bounds[4] = {0, a, b, P};
      long pairs_cls = 0, pairs_real = 0;
      for (int c = 0; c < 3; ++c) {
        const int l = bounds[c], h = bounds[c + 1];
        if (h <= l) continue;
        int kmin = l + pd - (G - 1) > 0 ? l + pd - (G - 1) : 0;          // smallest tap any row of the class can use
        int kmax = (h - 1) + pd < K - 1 ? (h - 1) + pd : K - 1;          // largest
        if (kmax < kmin) { kmin = 0; kmax = -1; }
        lo[d][nc[d]] = l; hi[d][nc[d]] = h; k0c[d][nc[d]] = kmin; njc[d][nc[d]] = kmax - kmin + 1;
        pairs_cls += (long)(h - l) * (kmax - kmin + 1);
        nc[d] += 1;
      }
      for (int o = 0; o < P; ++o)
        for (int kk = 0; kk < K; ++kk) pairs_real += (o + pd - kk >= 0 && o + pd - kk < G) ? 1 : 0;
      real *= (double)pairs_real; issued *= (double)P * K; issued_cls *= (double)pairs_cls;
    }
    // worth it when a fifth or more of the single-phase form's pairs are padding and the classes remove most of it
    if (real < 0.8 * issued && issued_cls < 0.9 * issued && nc[0] * nc[1] * nc[2] <= MAX_PHASES) {
      for (int d = 0; d < 3; ++d) p.ostride[d] = 1;
      int np = 0;
      for (int cz = 0; cz < nc[0]; ++cz)
        for (int cy = 0; cy < nc[1]; ++cy)
          for (int cx = 0; cx < nc[2]; ++cx) {
            Phase& ph = p.ph[np++];
            const int ci[3] = {cz, cy, cx};
            int M[3], nj[3], k0[3], d0[3], o0[3];
            for (int d = 0; d < 3; ++d) {
              M[d] = hi[d][ci[d]] - lo[d][ci[d]];
              nj[d] = njc[d][ci[d]];
              k0[d] = k0c[d][ci[d]];
              o0[d] = lo[d][ci[d]];
              d0[d] = o0[d] + pad[d] - k0[d];            // gathered coordinate of tap j: m + d0 - j
            }
            ph.Mz = M[0]; ph.My = M[1]; ph.Mx = M[2];
            ph.fMx = make_fastdiv(M[2]); ph.fMy = make_fastdiv(M[1]); ph.fMz = make_fastdiv(M[0]);
            ph.oz = o0[0]; ph.oy = o0[1]; ph.ox = o0[2];
            ph.nz = nj[0]; ph.ny = nj[1]; ph.nx = nj[2];
            ph.kz0 = k0[0]; ph.ky0 = k0[1]; ph.kx0 = k0[2];
            ph.dz0 = d0[0]; ph.dy0 = d0[1]; ph.dx0 = d0[2];
            if (nj[0] <= 0 || nj[1] <= 0 || nj[2] <= 0) { ph.nz = 0; ph.ny = 1; ph.nx = 1; }
          }
      p.nphase = np;
      p.classes = 1;
      return;
    }
  }
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
        ph.fMx = make_fastdiv(M[2] > 0 ? M[2] : 1); ph.fMy = make_fastdiv(M[1] > 0 ? M[1] : 1);
        ph.fMz = make_fastdiv(M[0] > 0 ? M[0] : 1);
        ph.oz = pz; ph.oy = py; ph.ox = px;
        ph.nz = nj[0]; ph.ny = nj[1]; ph.nx = nj[2];
        ph.kz0 = k0[0]; ph.ky0 = k0[1]; ph.kx0 = k0[2];
        ph.dz0 = d0[0]; ph.dy0 = d0[1]; ph.dx0 = d0[2];
        if (nj[0] == 0 || nj[1] == 0 || nj[2] == 0) { ph.nz = 0; ph.ny = 1; ph.nx = 1; }  // no taps: bias only
      }
  p.nphase = np;
}

}  // namespace mpgan
