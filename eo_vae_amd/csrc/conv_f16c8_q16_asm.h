// Hand-placed instruction statements shared by the 16 x 16-MFMA kernels (conv_f16c8_q16.hip, conv_f16c8_q16up.hip): LDS
// fragment reads, the two MFMA forms and the counted waits, each ONE volatile asm statement so that program order is issue
// order.  What hipcc does not know about such statements (and the callers therefore do by hand) is listed in
// conv_f16c8_q16.hip's header: consumers of an asm load may be scheduled in front of the wait (landed_*()), VALU write -> MFMA
// read needs two wait states (s_nop 1), MFMA write -> accumulator read at a loop exit (TAIL: s_nop 15).
#pragma once
#include <type_traits>

#include "conv_f16c8_common.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
using lds_ptr = __attribute__((address_space(3))) void*;

#ifndef EOVAE_Q16_RES_AHEAD
#define EOVAE_Q16_RES_AHEAD 3   // conv_f16c8_q16.hip epilogue: residual loads requested this many half co tiles ahead
#endif
#ifndef EOVAE_Q16_ABL
#define EOVAE_Q16_ABL 0   // timing-only experiments (WRONG results): 1 no s_nop in front of the correction MFMAs, 2 no output stores,
                          // 4 every tile of a block stored to the same place (the output stays in the caches), 8 no lane permutation in
                          // the epilogue, 16 no GroupNorm sums
#endif
typedef int i32x3 __attribute__((ext_vector_type(3)));
template <int OFF>
__device__ __forceinline__ void rd96(i32x3& d, unsigned a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void wr128(unsigned a, const i32x4& d) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(d), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void rd128(i32x4& d, unsigned a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
__device__ __forceinline__ void mfma_f(f32x4& c, const i32x4& a, const i32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// TAIL: the last MFMAs of a chunk carry their own s_nop 15 -- behind the chunk loop the compiler reads / moves accumulators
// (epilogue), and it knows no MFMA hazard for registers an asm statement wrote
template <bool TAIL = false>
__device__ __forceinline__ void mfma_c(f32x4& c, const i32x6& a, const i32x6& b, int sa, int sb) {
  if constexpr (TAIL) {
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\ts_nop 15"
                 : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    return;
  }
  // s_nop 1: the operands' last two dwords reach their six-register tuples through v_mov copies, and an MFMA may not read a
  // register a VALU instruction wrote less than two wait states earlier -- a hazard hipcc covers for its own MFMAs only
  if (EOVAE_Q16_ABL & 1)
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                 : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
  else
  asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
               : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
#ifndef EOVAE_Q16_SAFE
#define EOVAE_Q16_SAFE 0   // bisection: 1 every counted vmcnt wait becomes vmcnt(0), 2 every counted lgkmcnt wait lgkmcnt(0)
#endif
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"((EOVAE_Q16_SAFE & 2) ? 0 : N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((EOVAE_Q16_SAFE & 1) ? 0 : (N < 63 ? N : 63)) : "memory");
}

}  // namespace
