// Diagnostics that never touch a product tensor.
//
// eovae_clock_probe: the shader clock the chip holds under a dense matrix load, measured in the run that quotes it
// (bench.py `ceiling` / `clock_state`).  Boxes of one pool run the same build 10 % apart depending on
// a clock / power state nobody controls: the probe makes the state visible in the bench line.
// Method (MI355X_MICROARCH.md, DVFS give-back (6)): every workgroup (8 waves, one per CU) runs `iters` rounds of
// independent f16 MFMAs on pseudo-random operands held in registers and stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz wall clock) around the loop: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup.
// The stamps go only to the caller's buffer; nothing is computed from them.
#include "eovae_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, float* sink) {
  const unsigned tid = threadIdx.x + blockIdx.x * 512u;
  h16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {   // operands of both signs and a spread of exponents: power draw like real data
    unsigned h = (tid * 8u + (unsigned)j) * 2654435761u;
    h ^= h >> 15;
    a[j] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 8192.0f));
    b[j] = (_Float16)(((int)((h >> 16) & 0xffff) - 32768) * (1.0f / 16384.0f));
  }
  f32x16 c0, c1, c2, c3;
#pragma unroll
  for (int q = 0; q < 16; ++q) { c0[q] = 0.f; c1[q] = 0.f; c2[q] = 0.f; c3[q] = 0.f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
  if (s == 123.456f) *sink = s;   // keeps the MFMAs alive
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}

}  // namespace

extern "C" {

// out: device buffer of 2 * blocks uint64 {shader cycles, 100-MHz ticks} per workgroup; sink: any device float.
// iters rounds of 16 MFMAs per wave (8 waves per workgroup): ~iters x 0.5 us at 2 GHz.
int eovae_clock_probe(unsigned long long* out, float* sink, int blocks, int iters, void* stream) {
  EOVAE_CHECK_ARG(out && sink && blocks > 0 && blocks <= 4096 && iters > 0, "clock_probe: bad argument");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters, sink);
  EOVAE_CHECK_LAUNCH("eovae_clock_probe");
  return EOVAE_OK;
}

}  // extern "C"
