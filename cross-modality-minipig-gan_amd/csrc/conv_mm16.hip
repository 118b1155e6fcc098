// The K-stepped kernels with their matrix operands rounded to bf16 (MPGAN_CONV_MM_BF16, include/mpgan_hip.h):
// instances of the templates in conv_pipe.h / wgrad_pipe.h with MM16 = true, in a translation unit of their own.
//
// Why: config C5 (BASELINE.json: "3D 128^3 volume patches bs=4 bf16") ran its generator in fp32 through round 3, and
// the generator's heavy 3-D launches turned out ARITHMETIC-bound on the fp32 matrix pipe -- 16->16 @64^3 at 76-88
// TFLOP/s, 32->32 @32^3 at 61-90, 128->128 @16^3 at 83-98 of the 157 peak (profiles/r03_c5_g*_calls.txt): 0.45-0.62
// of that pipe, i.e. at least that share of their time is matrix time, and the bf16 pipe is 16x faster.  Storage
// stays fp32 (those launches move 100-1200 GB/s: bytes are not their problem, and the BatchNorm + PReLU prologue,
// bias, residual and statistics arithmetic keeps its precision); only the two MFMA operands are rounded, once, on
// their way into LDS, and products accumulate in fp32.
#include "conv_pipe.h"
#include "wgrad_pipe.h"
#include <stdlib.h>

namespace mpgan {

template <int BN, int TM, int TN, int WN, int WRAPS, int PRO, int KS>
static int launch_mm16_variant(const GatherConv& p, long maxM, hipStream_t st) {
  auto kern = gather_conv_pipe_kernel<BN, TM, TN, WN, WRAPS, PRO, false, KS, true>;
  constexpr int stage_b = (BM + BN) * MM16_PITCHB;
  constexpr int loop_b = KS * 2 * stage_b;
  constexpr int fold_b = KS > 1 ? TM * TN * 16 * 256 * 4 : 0;          // in-block split-K: the groups' sums through LDS
  constexpr int epi_b = CONV_EPI_FLOATS * 4;
  constexpr int smem = loop_b > fold_b ? (loop_b > epi_b ? loop_b : epi_b) : (fold_b > epi_b ? fold_b : epi_b);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("gather_conv_pipe (bf16 operands): hipFuncSetAttribute: %s", hipGetErrorString(e));
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
  hipLaunchKernelGGL(kern, grid, dim3(256 * KS), smem, st, q);
  return check_launch("gather_conv_pipe (bf16 operands)");
}

template <int WRAPS, int PRO>
static int launch_mm16_bn(const GatherConv& p, int variant, bool ksplit2, long maxM, hipStream_t st) {
  if constexpr (WRAPS == 1) {
    if (ksplit2 && variant == 64) return launch_mm16_variant<64, 1, 2, 1, 1, PRO, 2>(p, maxM, st);
    if (ksplit2 && variant == 32) return launch_mm16_variant<32, 1, 1, 1, 1, PRO, 2>(p, maxM, st);
  }
  if (variant == 128) return launch_mm16_variant<128, 2, 2, 2, WRAPS, PRO, 1>(p, maxM, st);
  if (variant == 64) return launch_mm16_variant<64, 1, 2, 1, WRAPS, PRO, 1>(p, maxM, st);
  return launch_mm16_variant<32, 1, 1, 1, WRAPS, PRO, 1>(p, maxM, st);
}

// What the MM16 instances cover: 16-byte vector operands (the caller checked), Cin a multiple of 32 or exactly 16,
// no prologue or a per-channel one (BatchNorm; InstanceNorm's per-sample vectors stay on the fp32 kernels).
bool mm16_gather_ok(const GatherConv& p) {
  static const bool off = dev_env("MPGAN_DBG_NO_MM16") != nullptr;
  return !off && (p.Cin % 32 == 0 || p.Cin == 16) && (!p.pro.scale || p.pro.n_stride == 0) && !p.fold.acc &&
         !p.stats_acc && p.ksplit <= 1 && !p.in_bf16 && !p.out_bf16;
}

int launch_gather_mm16(const GatherConv& p, int variant, bool ksplit2, long maxM, hipStream_t st) {
  if (p.Cin % 32 == 0)
    return p.pro.scale ? launch_mm16_bn<1, 1>(p, variant, ksplit2, maxM, st) : launch_mm16_bn<1, 0>(p, variant, ksplit2, maxM, st);
  return p.pro.scale ? launch_mm16_bn<2, 1>(p, variant, false, maxM, st) : launch_mm16_bn<2, 0>(p, variant, false, maxM, st);
}

// ---- weight gradient -----------------------------------------------------------------------------------------------
template <int BD, int BG, int TM, int TN, int WN, int PRO, bool PAD>
static int launch_wgrad_mm16_variant(const WgradParams& p, hipStream_t st) {
  auto kern = wgrad_pipe_kernel<BD, BG, TM, TN, WN, PRO, PAD, true>;
  constexpr int smem = 2 * WBK * (BD + BG) * (int)sizeof(float) + 2 * 32 * 16;   // + the row tables
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("wgrad_pipe (bf16 operands): hipFuncSetAttribute: %s", hipGetErrorString(e));
      return MPGAN_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)p.tiles_c * p.tiles_d * p.nsplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  return check_launch("wgrad_pipe (bf16 operands)");
}

template <int PRO>
static int dispatch_wgrad_mm16(const WgradParams& p, int BD, int BG, hipStream_t st, bool& handled) {
  handled = true;
  const bool nopad = (p.pz | p.py | p.px) == 0;
  if (BD == 128 && BG == 128) {
    if (nopad) return launch_wgrad_mm16_variant<128, 128, 2, 2, 2, PRO, false>(p, st);
    return launch_wgrad_mm16_variant<128, 128, 2, 2, 2, PRO, true>(p, st);
  }
  if (BD == 128 && BG == 64) return launch_wgrad_mm16_variant<128, 64, 1, 2, 1, PRO, true>(p, st);
  if (BD == 64 && BG == 128) return launch_wgrad_mm16_variant<64, 128, 2, 1, 4, PRO, true>(p, st);
  if (BD == 64 && BG == 64) return launch_wgrad_mm16_variant<64, 64, 1, 1, 2, PRO, true>(p, st);
  if (BD == 32 && BG == 128) return launch_wgrad_mm16_variant<32, 128, 1, 1, 4, PRO, true>(p, st);
  if (BD == 128 && BG == 32) return launch_wgrad_mm16_variant<128, 32, 1, 1, 1, PRO, true>(p, st);
  handled = false;
  return MPGAN_OK;
}

// The pipelined weight gradient with bf16 matrix operands; `handled` = false: this tile shape has no such instance
// (the caller runs the fp32 kernel).  Same tiles, splits and slabs as the fp32 form: the reducer is unchanged.
int launch_wgrad_mm16(const WgradParams& p, int BD, int BG, hipStream_t st, bool& handled) {
  static const bool off = dev_env("MPGAN_DBG_NO_MM16") != nullptr;
  handled = false;
  if (off) return MPGAN_OK;
  return p.pro.scale ? dispatch_wgrad_mm16<1>(p, BD, BG, st, handled) : dispatch_wgrad_mm16<0>(p, BD, BG, st, handled);
}

}  // namespace mpgan
