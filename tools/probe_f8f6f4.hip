// Probe for the block-scaled MFMA used by the f16 + fp8-correction arithmetic:
//   (1) operand pairing: lane (r, h) supplies 32 bytes of A row r and 32 bytes of B column r; the
//       product must be  D[m][n] = sum_h sum_j A_(m,h)[j] * B_(n,h)[j]  (same-lane byte j pairs with
//       byte j) for any lane -> k map, which is all the K-concatenated correction product needs;
//   (2) the E8M0 scale operands (uniform power of two) and mixed e4m3 / e5m2 operands;
//   (3) issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 against v_mfma_f32_32x32x16_{bf16,f16}:
//       the 32-channel step of bf16x3 (6 bf16 MFMAs) vs f16c8 (2 f16 MFMAs + 1 scaled fp8 MFMA).
// hipcc --offload-arch=gfx950 -O3 tools/probe_f8f6f4.hip -o gpurun_out/probe_f8 && gpurun_out/probe_f8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pack 4 floats into one dword of fp8 (fmt 0 = e4m3) or bf8 (fmt 1 = e5m2), byte 0 = v0
__device__ __forceinline__ int pack4(float v0, float v1, float v2, float v3, int fmt) {
  int w = 0;
  if (fmt == 0) {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v2, v3, w, true);
  }
  return w;
}

// la[lane][32], lb[lane][32] floats -> D[32][32]
template <int FA, int FB>
__global__ void pairing_kernel(const float* la, const float* lb, float* d, int sa, int sb) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int q = 0; q < 8; ++q) {
    const float* pa = la + lane * 32 + q * 4;
    const float* pb = lb + lane * 32 + q * 4;
    a[q] = pack4(pa[0], pa[1], pa[2], pa[3], FA);
    b[q] = pack4(pb[0], pb[1], pb[2], pb[3], FB);
  }
  f32x16 c;
  for (int q = 0; q < 16; ++q) c[q] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FA, FB, 0, sa, 0, sb);
  const int col = lane & 31, hh = lane >> 5;
  for (int q = 0; q < 16; ++q) d[((q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + col] = c[q];
}

// PAT 0: 6 bf16   1: 2 f16 + 1 scaled fp8   2: 3 scaled fp8   3: 6 f16   4: 12 16x16x32 bf16 (same flops as 0... x2 tiles)
template <int PAT>
__global__ __launch_bounds__(512) void rate_kernel(const float* src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 xb, wb;
  f16x8 xh, wh;
  i32x8 x8, w8;
  for (int j = 0; j < 8; ++j) {
    const float v = src[(lane * 8 + j) & 1023], w = src[(lane * 8 + j + 512) & 1023];
    xb[j] = (__bf16)v; wb[j] = (__bf16)w; xh[j] = (_Float16)v; wh[j] = (_Float16)w;
    x8[j] = pack4(v, w, -v, w * 0.5f, 1); w8[j] = pack4(w, v, -w, v * 0.5f, 0);
  }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (PAT == 0) {
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, xb, acc[t], 0, 0, 0);
      } else if (PAT == 1) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, x8, acc[t], 0, 1, 0, 115, 0, 127);
      } else if (PAT == 2) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
          acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, x8, acc[t], 0, 1, 0, 115, 0, 127);
      } else {
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) s += acc[t][q];
  if (s == 12345.678f) out[threadIdx.x] = s;   // keep the accumulators live
}

static float frand(unsigned& st) { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f * 2.f - 1.f; }

int main() {
  // ---------------- (1)+(2) pairing / scale / mixed formats, exact small integers ----------------
  std::vector<float> la(64 * 32), lb(64 * 32), d(32 * 32);
  unsigned st = 12345;
  for (auto& v : la) v = (float)((int)(frand(st) * 4.49f));   // -4..4: exact in e4m3 and e5m2
  for (auto& v : lb) v = (float)((int)(frand(st) * 4.49f));
  float *dla, *dlb, *dd;
  CK(hipMalloc(&dla, la.size() * 4)); CK(hipMalloc(&dlb, lb.size() * 4)); CK(hipMalloc(&dd, d.size() * 4));
  CK(hipMemcpy(dla, la.data(), la.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlb, lb.data(), lb.size() * 4, hipMemcpyHostToDevice));
  auto check = [&](const char* name, float scale) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 32; ++j) ref += (double)la[(m + 32 * h) * 32 + j] * lb[(n + 32 * h) * 32 + j];
        if (fabs(d[m * 32 + n] - ref * scale) > 1e-6 * fabs(ref * scale) + 1e-9) ++bad;
      }
    printf("pairing %-34s : %s (%d / 1024 mismatches; d[3][5] = %g)\n", name, bad ? "MISMATCH" : "exact", bad, d[3 * 32 + 5]);
  };
  hipLaunchKernelGGL((pairing_kernel<0, 0>), dim3(1), dim3(64), 0, 0, dla, dlb, dd, 127, 127);
  check("e4m3 x e4m3, scales 2^0", 1.f);
  hipLaunchKernelGGL((pairing_kernel<0, 1>), dim3(1), dim3(64), 0, 0, dla, dlb, dd, 127, 127);
  check("e4m3 x e5m2, scales 2^0", 1.f);
  hipLaunchKernelGGL((pairing_kernel<0, 1>), dim3(1), dim3(64), 0, 0, dla, dlb, dd, 127 - 12, 127 + 3);
  check("e4m3 x e5m2, scales 2^-12 * 2^3", 1.f / 512.f);
  hipLaunchKernelGGL((pairing_kernel<1, 1>), dim3(1), dim3(64), 0, 0, dla, dlb, dd, 127 - 20, 127 - 20);
  check("e5m2 x e5m2, scales 2^-20 * 2^-20", ldexpf(1.f, -40));

  // ---------------- (3) issue rate ----------------
  std::vector<float> src(1024);
  for (auto& v : src) v = frand(st);
  float *dsrc, *dout;
  CK(hipMalloc(&dsrc, 4096)); CK(hipMalloc(&dout, 4096));
  CK(hipMemcpy(dsrc, src.data(), 4096, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  const char* names[4] = {"6 x bf16 32x32x16 (bf16x3, 32 ch)", "2 x f16 + 1 x scaled fp8 (f16c8, 32 ch)",
                          "3 x scaled fp8 32x32x64", "6 x f16 32x32x16"};
  for (int threads = 256; threads <= 512; threads += 256) {
    for (int rep = 0; rep < 2; ++rep)
      for (int pat = 0; pat < 4; ++pat) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 3; ++k) {
          if (pat == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(256), dim3(threads), 0, 0, dsrc, dout, iters);
          if (pat == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(threads), 0, 0, dsrc, dout, iters);
          if (pat == 2) hipLaunchKernelGGL((rate_kernel<2>), dim3(256), dim3(threads), 0, 0, dsrc, dout, iters);
          if (pat == 3) hipLaunchKernelGGL((rate_kernel<3>), dim3(256), dim3(threads), 0, 0, dsrc, dout, iters);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        // 32-channel steps of a 32x32 tile: iters * 4 per wave; algorithmic flops 2*32*32*32 each
        const double steps = (double)iters * 4 * (threads / 64) * 256;
        if (rep) printf("%d waves/CU  %-42s %8.3f ms  %7.1f algorithmic TFLOP/s  (%.1f cycles/step/SIMD at 2.4 GHz)\n",
                        threads / 64, names[pat], ms, steps * 65536 / (ms * 1e-3) / 1e12,
                        ms * 1e-3 * 2.4e9 / (iters * 4.0 * (threads / 256)));
      }
  }
  return 0;
}
