// Shared host/device helpers for libmpgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "mpgan_hip.h"

// ---- host dry-run build (development only: `make DRYRUN=1` -> libmpgan_hip_dry.so) -----------------------------
// Host code compiled alone (--cuda-host-only) with AddressSanitizer; every kernel launch becomes a host-side
// check of its launch geometry with the arguments still marshalled (the parameter structs are copied exactly as
// a real launch copies them), and the few other runtime calls succeed without a device.  tools/asan_dryrun.py
// drives whole training steps at the C3 / C5 shapes through it on the CPU container: any heap / stack overrun in
// the argument marshalling or the geometry builders is an ASan report, any launch outside the hardware's limits
// an error status.  The product build has none of this.
#ifdef MPGAN_DRYRUN
namespace mpgan {
int dry_note_launch(const char* kernel, const char* where, dim3 grid, dim3 block, size_t lds);
template <typename... A>
inline void dry_launch(const char* kernel, const char* where, dim3 grid, dim3 block, size_t lds, hipStream_t, A... args) {
  // the by-value copies above ARE the marshalling under test; touch every byte so ASan sees a short struct
  const volatile unsigned char* bytes[] = {reinterpret_cast<const volatile unsigned char*>(&args)...};
  const size_t sizes[] = {sizeof(args)...};
  unsigned acc = 0;
  for (size_t i = 0; i < sizeof...(A); ++i)
    for (size_t b = 0; b < sizes[i]; ++b) acc += bytes[i][b];
  (void)acc;
  dry_note_launch(kernel, where, grid, block, lds);
}
hipError_t dry_last_error();
inline hipError_t dry_ok(...) { return hipSuccess; }
inline hipError_t dry_symbol_address(void** p) {
  static char page[4096];
  *p = page;
  return hipSuccess;
}
}  // namespace mpgan
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  ::mpgan::dry_launch(#kern, __PRETTY_FUNCTION__, grid, block, lds, stream, __VA_ARGS__)
#define hipFuncSetAttribute(...) ::mpgan::dry_ok(__VA_ARGS__)
#define hipMemcpyToSymbol(...) ::mpgan::dry_ok(__VA_ARGS__)
#define hipMemsetAsync(...) ::mpgan::dry_ok(__VA_ARGS__)
#define hipGetSymbolAddress(p, sym) ::mpgan::dry_symbol_address(p)
#define hipGetLastError() ::mpgan::dry_last_error()
#endif

namespace mpgan {

void set_error(const char* fmt, ...);

// Development switches (MPGAN_DBG_* and friends: A/B runs, what-if builds, forced kernel forms) exist only in the
// development build (`make DEV=1` -> libmpgan_hip_dev.so, loaded through MPGAN_LIB_PATH): the product library reads
// no environment variable at all -- its dispatch depends on the arguments of a call and on nothing else.
#ifdef MPGAN_DEV_SWITCHES
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

#define MPGAN_CHECK_ARG(cond, ...)                         \
  do {                                                     \
    if (!(cond)) {                                         \
      ::mpgan::set_error(__VA_ARGS__);                     \
      return MPGAN_ERR_INVALID;                            \
    }                                                      \
  } while (0)

#define MPGAN_UNSUPPORTED(cond, ...)                       \
  do {                                                     \
    if (cond) {                                            \
      ::mpgan::set_error(__VA_ARGS__);                     \
      return MPGAN_ERR_UNSUPPORTED;                        \
    }                                                      \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MPGAN_ERR_HIP;
  }
  return MPGAN_OK;
}

// Device view of the optional normalise+activate prologue.
struct Pro {
  const float* scale;
  const float* shift;
  const float* slope_ptr;
  float slope;
  int n_stride;
  int act;
};

// Device view of mpgan_peer_taps (perceptual-loss taps against a peer pass).
struct Peer {
  const float* z;
  const float* scale;
  const float* shift;
  const float* coef;   // device float[3]: (z, y, a) gradient coefficients; null => no peer
  int ld;
};

inline Peer make_peer(const mpgan_peer_taps* t) {
  Peer r;
  if (t == nullptr || t->coef == nullptr) { r.z = r.scale = r.shift = r.coef = nullptr; r.ld = 0; }
  else { r.z = t->z_peer; r.scale = t->scale_peer; r.shift = t->shift_peer; r.coef = t->coef; r.ld = t->ld_peer; }
  return r;
}

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

inline Pro make_pro(const mpgan_prologue* p) {
  Pro r;
  if (p == nullptr || p->scale == nullptr) {
    r.scale = nullptr; r.shift = nullptr; r.slope_ptr = nullptr;
    r.slope = 1.f; r.n_stride = 0; r.act = MPGAN_ACT_NONE;
  } else {
    r.scale = p->scale; r.shift = p->shift; r.slope_ptr = p->slope_ptr;
    r.slope = p->slope; r.n_stride = p->n_stride; r.act = p->act;
  }
  return r;
}

__device__ __forceinline__ float pro_slope(const Pro& p) {
  return p.slope_ptr ? __builtin_nontemporal_load(p.slope_ptr) : p.slope;
}

__device__ __forceinline__ float act_apply(float y, int act, float slope) {
  return (act == MPGAN_ACT_LEAKY && y < 0.f) ? y * slope : y;
}

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2).  Map the
// hardware block id to a WORK id so that each XCD walks a contiguous range of work
// items: neighbouring tiles (shared halo rows, shared dy panels) then hit the same L2.
// Bijective for any grid size.  Placement is a speed matter only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned orig, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Division of a 32-bit unsigned by a launch-time constant: multiply-high + shift
// (Granlund-Montgomery, branch-free form) instead of the ~40-instruction software divide.
struct FastDiv {
  unsigned d, mul, sh;   // q = (((n - t) >> 1) + t) >> sh, t = mulhi(mul, n);  d == 1: q = n
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.sh = 0; f.d = 1; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.sh = l - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  if (f.d == 1) return n;
  const unsigned t = __umulhi(f.mul, n);
  return (((n - t) >> 1) + t) >> f.sh;
}
// branch-free form (keeps a software-pipelined K-step one basic block)
__device__ __forceinline__ void fdivmod_nb(unsigned n, const FastDiv& f, unsigned& q, unsigned& r) {
  const unsigned t = __umulhi(f.mul, n);
  const unsigned qq = (((n - t) >> 1) + t) >> f.sh;
  q = f.d == 1 ? n : qq;
  r = n - q * f.d;
}
__device__ __forceinline__ void fdivmod(unsigned n, const FastDiv& f, unsigned& q, unsigned& r) {
  q = fdiv(n, f);
  r = n - q * f.d;
}

// ---- in-kernel phase stamps (development builds only: `make STAMPS=1` -> libmpgan_hip_stamps.so) ----------
// With -DMPGAN_STAMPS every gather-conv launch gets a slice [blocks][MPGAN_STAMP_SLOTS] of a caller-supplied
// device buffer (mpgan_debug_stamps) and thread 0 of each block records the 100 MHz constant clock
// (s_memrealtime: the same time base on every CU) at its phase boundaries: what the in-kernel fixed cost of the
// generator's short kernels consists of is then MEASURED (tools/kernel_phases.py), not inferred from what-if
// builds.  In the product build the macros expand to nothing and GatherConv has no such field.
constexpr int MPGAN_STAMP_SLOTS = 12;
#ifdef MPGAN_STAMPS
struct StampCtx { unsigned long long* base; long launches, blocks; long next; };
StampCtx& stamp_ctx();                      // host side (capi.cpp)
#define MPGAN_STAMP_FIELD unsigned long long* stamps; int stamp_blocks;
#define MPGAN_STAMP(p, slot)                                                                         \
  do {                                                                                               \
    if ((p).stamps && threadIdx.x == 0 && (int)blockIdx.x < (p).stamp_blocks)                        \
      (p).stamps[(long)blockIdx.x * MPGAN_STAMP_SLOTS + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define MPGAN_STAMP_VALUE(p, slot, v)                                                                \
  do {                                                                                               \
    if ((p).stamps && threadIdx.x == 0 && (int)blockIdx.x < (p).stamp_blocks)                        \
      (p).stamps[(long)blockIdx.x * MPGAN_STAMP_SLOTS + (slot)] = (unsigned long long)(v);          \
  } while (0)
#define MPGAN_STAMP_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define MPGAN_STAMP_FIELD
#define MPGAN_STAMP(p, slot) ((void)0)
#define MPGAN_STAMP_VALUE(p, slot, v) ((void)0)
#define MPGAN_STAMP_NOW() 0ull
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace mpgan
