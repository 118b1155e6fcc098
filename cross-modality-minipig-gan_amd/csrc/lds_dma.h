// LDS-DMA staging helpers shared by the bf16 kernels (conv_bf16.hip) and the fp32 DMA-staged kernel (conv_igemm.hip).
#pragma once
#include "mpgan_common.h"

namespace mpgan {

#define GLDS16(gptr, lptr)                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),             \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// Fragment reads as inline asm: hipcc's waitcnt pass drains every LDS-DMA in flight (vmcnt(0)) in front of an LDS
// read it cannot tell apart from the DMA's destination, which would undo the counted-vmcnt pipeline; an asm
// statement is invisible to that pass.  Its result register is NOT protected either: each set of reads is
// followed, before its first use, by lds_wait(...) -- `s_waitcnt lgkmcnt(0)` tied to the registers it covers.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ i32x4 lds_read_b128(unsigned addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// N reads at addr, addr + STRIDE, ...: the stride goes into the instruction's 16-bit offset field, so the N addresses
// cost one address register instead of N vector adds (the asm operand must be a register, and hipcc cannot fold an
// addition into an asm statement's offset field by itself).
template <int N, int STRIDE, int I = 0>
__device__ __forceinline__ void lds_read_b128_n(i32x4 (&dst)[N], unsigned addr) {
  static_assert((N - 1) * STRIDE < 65536, "ds_read_b128 offset field");
  if constexpr (I < N) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(addr), "n"(I * STRIDE));
    lds_read_b128_n<N, STRIDE, I + 1>(dst, addr);
  }
}
__device__ __forceinline__ i32x2 lds_read_tr16_b64(unsigned addr) {
  i32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// k-sub s (0..3) of a tile whose k-subs lie 16 rows of 256 B apart: the k-sub goes into the offset field (s is a
// constant once the k-sub loop is unrolled; the dead cases fold away), so a tile's fragment addresses are formed once
// per K-step instead of once per k-sub.
template <int OFF>
__device__ __forceinline__ i32x2 lds_read_tr16_b64_o(unsigned addr) {
  i32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ i32x2 lds_read_tr16_ksub(unsigned addr, int s) {
  switch (s) {
    case 0: return lds_read_tr16_b64_o<0>(addr);
    case 1: return lds_read_tr16_b64_o<16 * 256>(addr);
    case 2: return lds_read_tr16_b64_o<2 * 16 * 256>(addr);
    default: return lds_read_tr16_b64_o<3 * 16 * 256>(addr);
  }
}
template <int NA, int NB, typename V>
__device__ __forceinline__ void lds_wait(V (&a)[NA], V (&b)[NB]) {
  if constexpr (NA == 4 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
  else if constexpr (NA == 2 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
  else if constexpr (NA == 4 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else if constexpr (NA == 2 && NB == 1)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]));
  else if constexpr (NA == 4 && NB == 8)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  else if constexpr (NA == 1 && NB == 1)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]));
  else if constexpr (NA == 1 && NB == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]));
  else if constexpr (NA == 8 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  else
    static_assert(NA == 0, "lds_wait: unsupported fragment set");
}

constexpr unsigned HW_OOB = 0xFFFF0000u;          // voffset of a masked buffer-load piece: beyond every operand admitted
#define BLDS16(rsrc, lptr, voff, soff)                                                                              \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), \
                                           (int)(soff), 0, 0)

}  // namespace mpgan
