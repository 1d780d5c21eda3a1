// Probe for the 1.5-unit scheme of DESIGN.md s4 (round 4): the correction product on block-scaled FP6 operands.
//   (1) FP6 (e2m3) operand packing of v_mfma_scale_f32_32x32x64_f8f6f4: 32 values x 6 bits per lane in dwords 0..5, element j in
//       bits [6j, 6j + 6) -- checked by an exact integer product against the fp8 pairing rule (lane (r, h) supplies the 32
//       K values of row / column r in K block h);
//   (2) which lane's scale byte applies to which (row, K block): one lane at a time gets the scale 2^1, all others 2^0, with
//       all-ones operands -- the rows / columns of D that change, and by how much, give the map (f16c8 round 2 found that
//       lane-varying scales "did not behave as a per-(row, K-block) scale": this prints what they do);
//   (3) issue rate: fp6 x fp6, fp6 x fp8, fp8 x fp8, against f16 32x32x16.
// hipcc --offload-arch=gfx950 -O3 tools/probe_mxfp6.hip -o gpurun_out/probe_mxfp6 && gpurun_out/probe_mxfp6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// e2m3 code of a small integer in -7..7 that the format holds exactly (0, 1, 1.5 .. : here 0, +-1, +-2, +-3, +-4, +-6)
__host__ __device__ inline unsigned e2m3_code(int v) {
  const unsigned s = v < 0 ? 32u : 0u;
  const int a = v < 0 ? -v : v;
  unsigned c = 0;
  if (a == 1) c = 8; else if (a == 2) c = 16; else if (a == 3) c = 20; else if (a == 4) c = 24; else if (a == 6) c = 28;
  return s | c;
}
__device__ inline i32x8 pack_fp6(const int* v) {   // 32 small integers -> 6 dwords
  unsigned long long acc = 0;
  int bits = 0, w = 0;
  i32x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    acc |= (unsigned long long)e2m3_code(v[j]) << bits;
    bits += 6;
    while (bits >= 32) { r[w++] = (int)(unsigned)acc; acc >>= 32; bits -= 32; }
  }
  return r;
}
__device__ inline i32x8 pack_fp8(const int* v) {   // e4m3
  i32x8 r;
  for (int q = 0; q < 8; ++q) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[4 * q], (float)v[4 * q + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[4 * q + 2], (float)v[4 * q + 3], w, true);
    r[q] = w;
  }
  return r;
}

// la[lane][32], lb[lane][32] ints; sa[lane], sb[lane] scale dwords -> D[32][32]
template <int FA, int FB>
__global__ void mx_kernel(const int* la, const int* lb, const int* sa, const int* sb, float* d) {
  const int lane = threadIdx.x;
  const i32x8 a = FA == 2 ? pack_fp6(la + lane * 32) : pack_fp8(la + lane * 32);
  const i32x8 b = FB == 2 ? pack_fp6(lb + lane * 32) : pack_fp8(lb + lane * 32);
  f32x16 c;
  for (int q = 0; q < 16; ++q) c[q] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FA, FB, 0, sa[lane], 0, sb[lane]);
  const int col = lane & 31, hh = lane >> 5;
  for (int q = 0; q < 16; ++q) d[((q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + col] = c[q];
}

template <int PAT>   // 0: f16 32x32x16 x4   1: fp8 x fp8   2: fp6 x fp6   3: fp6 x fp8
__global__ __launch_bounds__(512) void rate_kernel(const int* src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  i32x8 a8, b8;
  f16x8 ah, bh;
  for (int j = 0; j < 8; ++j) { a8[j] = src[(lane * 8 + j) & 1023]; b8[j] = src[(lane * 8 + j + 512) & 1023]; ah[j] = (_Float16)(float)(src[j] & 7); bh[j] = (_Float16)(float)(src[j + 8] & 7); }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (PAT == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
      } else if (PAT == 1) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 127, 0, 127);
      else if (PAT == 2) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 2, 2, 0, 127, 0, 127);
      else acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 2, 0, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) s += acc[t][q];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static int irand(unsigned& st) { st = st * 1664525u + 1013904223u; static const int tab[9] = {0, 1, -1, 2, -2, 3, -3, 4, -4}; return tab[(st >> 10) % 9]; }

int main() {
  std::vector<int> la(64 * 32), lb(64 * 32), sa(64, 127), sb(64, 127);
  std::vector<float> d(32 * 32);
  unsigned st = 777;
  for (auto& v : la) v = irand(st);
  for (auto& v : lb) v = irand(st);
  int *dla, *dlb, *dsa, *dsb;
  float* dd;
  CK(hipMalloc(&dla, la.size() * 4)); CK(hipMalloc(&dlb, lb.size() * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, d.size() * 4));
  auto up = [&]() {
    CK(hipMemcpy(dla, la.data(), la.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlb, lb.data(), lb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
  };
  auto run = [&](int fa, int fb) {
    up();
    if (fa == 2 && fb == 2) hipLaunchKernelGGL((mx_kernel<2, 2>), dim3(1), dim3(64), 0, 0, dla, dlb, dsa, dsb, dd);
    else if (fa == 2) hipLaunchKernelGGL((mx_kernel<2, 0>), dim3(1), dim3(64), 0, 0, dla, dlb, dsa, dsb, dd);
    else hipLaunchKernelGGL((mx_kernel<0, 0>), dim3(1), dim3(64), 0, 0, dla, dlb, dsa, dsb, dd);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
  };
  auto check = [&](const char* name) {
    int bad = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 32; ++j) ref += (double)la[(m + 32 * h) * 32 + j] * lb[(n + 32 * h) * 32 + j];
        if (fabs(d[m * 32 + n] - ref) > 1e-6 * fabs(ref) + 1e-9) ++bad;
      }
    printf("pairing %-28s : %s (%d / 1024 mismatches; d[3][5] = %g)\n", name, bad ? "MISMATCH" : "exact", bad, d[3 * 32 + 5]);
  };
  run(0, 0); check("e4m3 x e4m3");
  run(2, 2); check("e2m3 x e2m3 (fp6)");
  run(2, 0); check("e2m3 x e4m3");

  // ---- (2) which lane's scale reaches which (row, K block)?  all-ones operands: D = 64 everywhere at scale 2^0
  for (auto& v : la) v = 1;
  for (auto& v : lb) v = 1;
  for (int fmt = 0; fmt <= 2; fmt += 2) {
    for (int side = 0; side < 2; ++side) {
      printf("-- scale of %s, operands %s: lane -> {rows|cols changed: count, value} (64 = untouched)\n", side ? "B" : "A", fmt ? "fp6 e2m3" : "fp8 e4m3");
      for (int L = 0; L < 64; L += (L < 4 || (L >= 30 && L < 34) ? 1 : 7)) {
        std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
        (side ? sb : sa)[L] = 128;             // 2^1 in byte 0
        run(fmt, fmt);
        int nrow = 0, ncol = 0; float val = 64.f; int r0 = -1, c0 = -1;
        for (int m = 0; m < 32; ++m) { bool ch = false; for (int n = 0; n < 32; ++n) if (d[m * 32 + n] != 64.f) { ch = true; val = d[m * 32 + n]; } if (ch) { ++nrow; if (r0 < 0) r0 = m; } }
        for (int n = 0; n < 32; ++n) { bool ch = false; for (int m = 0; m < 32; ++m) if (d[m * 32 + n] != 64.f) ch = true; if (ch) { ++ncol; if (c0 < 0) c0 = n; } }
        printf("   lane %2d: %2d rows (first %2d) x %2d cols (first %2d) changed, value %g\n", L, nrow, r0, ncol, c0, val);
      }
    }
  }
  // byte select: scale dword = 0x80 7f 7f 7f with op_sel? (builtin arg is the op_sel value) -- byte 0 only is tested above

  // ---- (3) issue rate
  std::vector<int> src(1024);
  for (auto& v : src) { st = st * 1664525u + 1013904223u; v = (int)(st & 0x1f1f1f1f); }
  int* dsrc; float* dout;
  CK(hipMalloc(&dsrc, 4096)); CK(hipMalloc(&dout, 4096));
  CK(hipMemcpy(dsrc, src.data(), 4096, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  const char* names[4] = {"4 x f16 32x32x16 (K = 64)", "fp8 x fp8 32x32x64", "fp6 x fp6 32x32x64", "fp6 x fp8 32x32x64"};
  for (int rep = 0; rep < 2; ++rep)
    for (int pat = 0; pat < 4; ++pat) {
      CK(hipEventRecord(e0));
      for (int k = 0; k < 3; ++k) {
        if (pat == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(256), dim3(256), 0, 0, dsrc, dout, iters);
        if (pat == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(256), 0, 0, dsrc, dout, iters);
        if (pat == 2) hipLaunchKernelGGL((rate_kernel<2>), dim3(256), dim3(256), 0, 0, dsrc, dout, iters);
        if (pat == 3) hipLaunchKernelGGL((rate_kernel<3>), dim3(256), dim3(256), 0, 0, dsrc, dout, iters);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 3;
      if (rep) printf("%-28s %8.3f ms for %d x 4 K=64 products per wave (1 wave / SIMD)  -> %.1f cycles per K=64 product at 2.4 GHz\n", names[pat], ms, iters,
                      ms * 1e-3 * 2.4e9 / (iters * 4.0));
    }
  return 0;
}
