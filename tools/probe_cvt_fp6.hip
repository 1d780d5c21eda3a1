// Probe for the MX-FP6 operand path: the gfx950 conversion instructions that produce the 32 x 6-bit
// operand of v_mfma_scale_f32_32x32x64_f8f6f4 (e2m3), before any kernel relies on them.
//   (1) element order: which input (set a / b, index) lands in which 6-bit field of the 6 result dwords, for
//       v_cvt_scalef32_2xpk16_fp6_f32 (two sets of 16 f32) and v_cvt_scalef32_pk32_fp6_f16 (32 packed f16);
//   (2) rounding and saturation (ties, 7.5 clamp), the meaning of the scale operand (divisor), negative zero;
//   (3) issue cost of one conversion next to 32 v_max3 (what a 32-channel block costs a loader);
//   (4) end to end: two random [32 x 64] operands quantised per (row, 32-K block) with the block scale rule of the kernels
//       (scale byte = biased exponent of the block amax - 2), multiplied by the MFMA with the scale byte riding in dword 6
//       of each lane's fragment, against a host emulation of the same quantisation in double.
// hipcc --offload-arch=gfx950 -O3 tools/probe_cvt_fp6.hip -o gpurun_out/probe_cvt_fp6 && gpurun_out/probe_cvt_fp6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ROCm 7.2's hipcc allocates the builtin's 6-register result ON TOP of its 16-register sources (v[0:5] <- v[2:17], ...: no
// early-clobber on the instruction), and the hardware writes result dwords before it has read every source: wrong fields
// from the 13th on (first run of this probe, profiles/r05_probe_cvt_fp6.log).  The kernels therefore issue the instruction
// through inline assembly with an early-clobber result.
__device__ __forceinline__ u32x6 cvt_2xpk16(f32x16 a, f32x16 b, float scale) {
  u32x6 r;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(scale));
  return r;
}
template <int ASM>
__global__ void cvt32_kernel(const float* x, unsigned* o, const float* s) {
  f32x16 a, b;
  for (int i = 0; i < 16; ++i) { a[i] = x[threadIdx.x * 32 + i]; b[i] = x[threadIdx.x * 32 + 16 + i]; }
  const u32x6 r = ASM ? cvt_2xpk16(a, b, s[threadIdx.x]) : __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, s[threadIdx.x]);
  for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
__global__ void cvt16_kernel(const float* x, unsigned* o, const float* s) {
  f16x32 a;
  for (int i = 0; i < 32; ++i) a[i] = (_Float16)x[threadIdx.x * 32 + i];
  const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(a, s[threadIdx.x]);
  for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}

// block scale rule of the kernels: byte = max(biased exponent of amax - 2, 1); divisor = 2^(byte - 127)
__device__ __host__ inline unsigned scale_byte(float amax) {
  const unsigned bits = __builtin_bit_cast(unsigned, amax);
  const int be = (int)((bits >> 23) & 0xff) - 2;
  return (unsigned)(be < 1 ? 1 : be);
}

// one wave: lane (r, h) holds the 32 K values of row r / column r in K block h
__global__ void mx_mfma_kernel(const float* xa, const float* xb, float* d) {
  const int lane = threadIdx.x;
  f32x16 a0, a1, b0, b1;
  float ma = 0.f, mb = 0.f;
  for (int i = 0; i < 16; ++i) {
    a0[i] = xa[lane * 32 + i]; a1[i] = xa[lane * 32 + 16 + i];
    b0[i] = xb[lane * 32 + i]; b1[i] = xb[lane * 32 + 16 + i];
    ma = fmaxf(ma, fmaxf(fabsf(a0[i]), fabsf(a1[i])));
    mb = fmaxf(mb, fmaxf(fabsf(b0[i]), fabsf(b1[i])));
  }
  const unsigned sa = scale_byte(ma), sb = scale_byte(mb);
  const u32x6 qa = cvt_2xpk16(a0, a1, __builtin_bit_cast(float, sa << 23));
  const u32x6 qb = cvt_2xpk16(b0, b1, __builtin_bit_cast(float, sb << 23));
  i32x8 av, bv;
  for (int i = 0; i < 6; ++i) { av[i] = (int)qa[i]; bv[i] = (int)qb[i]; }
  av[6] = (int)sa; av[7] = 0x55555555;   // dword 6 = scale byte (+ junk above it: only byte 0 may count), dword 7 junk
  bv[6] = (int)(sb | 0xabcd00u); bv[7] = -1;
  f32x16 c;
  for (int q = 0; q < 16; ++q) c[q] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 2, 2, 0, av[6], 0, bv[6]);
  const int col = lane & 31, hh = lane >> 5;
  for (int q = 0; q < 16; ++q) d[((q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + col] = c[q];
}

template <int WITH_CVT>
__global__ __launch_bounds__(256) void cost_kernel(const float* src, float* out, int iters) {
  f32x16 a, b;
  for (int i = 0; i < 16; ++i) { a[i] = src[(threadIdx.x + i) & 1023]; b[i] = src[(threadIdx.x + 16 + i) & 1023]; }
  unsigned acc = 0;
  float m = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(fmaxf(fabsf(a[i]), fabsf(b[i])), m);
    if (WITH_CVT) {
      const u32x6 r = cvt_2xpk16(a, b, __builtin_bit_cast(float, scale_byte(m) << 23));
      acc ^= r[0] ^ r[5];
      a[it & 15] += __builtin_bit_cast(float, (r[1] & 0xff) | 0x3f800000u) * 1e-3f;
    } else {
      a[it & 15] += m * 1e-3f;
    }
    asm volatile("" : "+v"(m));
  }
  if (acc == 0x12345u || m == 7.f) out[threadIdx.x] = m + (float)acc;
}

static double e2m3_quant(double v) {   // nearest e2m3 value (ties to even), saturating at 7.5
  static double grid[32];
  static bool init = false;
  if (!init) {
    int n = 0;
    for (int m = 0; m < 8; ++m) grid[n++] = m / 8.0;                       // subnormals (step 0.125)
    for (int e = 1; e < 4; ++e) for (int m = 0; m < 8; ++m) grid[n++] = (1 + m / 8.0) * pow(2.0, e - 1);
    init = true;
  }
  const double a = fabs(v);
  if (a >= 7.5) return v < 0 ? -7.5 : 7.5;
  int lo = 0;
  while (lo + 1 < 32 && grid[lo + 1] <= a) ++lo;
  double q = grid[lo];
  if (lo + 1 < 32) {
    const double up = grid[lo + 1], dl = a - grid[lo], du = up - a;
    if (du < dl || (du == dl && (lo & 1))) q = up;   // tie -> even code
  }
  return v < 0 ? -q : q;
}

int main() {
  std::vector<float> x(64 * 32, 0.f), s(64, 1.f);
  std::vector<unsigned> o(64 * 6);
  float *dx, *ds; unsigned* dout;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&ds, 256)); CK(hipMalloc(&dout, o.size() * 4));
  auto run = [&](int which) {
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, s.data(), 256, hipMemcpyHostToDevice));
    if (which == 0) hipLaunchKernelGGL(cvt32_kernel<1>, dim3(1), dim3(64), 0, 0, dx, dout, ds);
    else if (which == 2) hipLaunchKernelGGL(cvt32_kernel<0>, dim3(1), dim3(64), 0, 0, dx, dout, ds);
    else hipLaunchKernelGGL(cvt16_kernel, dim3(1), dim3(64), 0, 0, dx, dout, ds);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  };
  auto field = [&](int lane, int f) {   // 6-bit field f of lane's 192 bits
    const int bit = 6 * f;
    unsigned long long w = o[lane * 6 + bit / 32];
    if (bit / 32 + 1 < 6) w |= (unsigned long long)o[lane * 6 + bit / 32 + 1] << 32;
    return (unsigned)((w >> (bit % 32)) & 63);
  };
  // (1) element order: lane j has 1.0 at input j only
  int map32[32];   // input index -> field, 2xpk16 (asm form)
  for (int which = 0; which < 3; ++which) {
    std::fill(x.begin(), x.end(), 0.f);
    for (int j = 0; j < 32; ++j) x[j * 32 + j] = 1.0f;
    run(which);
    printf("(1) %s: input index -> 6-bit field holding code 8 (1.0):", which == 1 ? "pk32_fp6_f16" : (which ? "2xpk16_fp6_f32 (builtin)" : "2xpk16_fp6_f32 (asm, early-clobber)"));
    bool ident = true, inter = true;
    for (int j = 0; j < 32; ++j) {
      int at = -1, cnt = 0;
      for (int f = 0; f < 32; ++f) if (field(j, f) == 8) { at = f; ++cnt; }
      printf(" %d>%d%s", j, at, cnt == 1 ? "" : "?");
      ident = ident && at == j && cnt == 1;
      inter = inter && at == (j < 16 ? 2 * j : 2 * (j - 16) + 1) && cnt == 1;
      if (which == 0) map32[j] = at;
    }
    printf("  => %s\n", ident ? "IDENTITY" : (inter ? "INTERLEAVED (a[i] -> field 2i, b[i] -> field 2i + 1)" : "OTHER / BROKEN"));
  }
  // (2) rounding / saturation / scale
  {
    const float tv[] = {1.0625f, 1.1875f, 0.0625f, 0.1875f, 7.4f, 7.6f, 100.f, -3.3f, 0.06f, 0.07f, 3.75f, 3.7499f, 1e-30f, -0.f, 5.f, 6.25f};
    for (int which = 0; which < 2; ++which) {
      std::fill(x.begin(), x.end(), 0.f);
      for (int j = 0; j < 16; ++j) { x[j] = tv[j]; x[32 + j] = tv[j] * 4.f; x[64 + j] = tv[j] * 0.25f; }
      s[0] = 1.f; s[1] = 4.f; s[2] = 0.25f;
      run(which);
      int bad = 0;
      for (int l = 0; l < 3; ++l)
        for (int j = 0; j < 16; ++j) {
          const unsigned c = field(l, which == 0 ? map32[j] : j);
          const int e = (c >> 3) & 3, m = c & 7;
          const double val = (e ? (1 + m / 8.0) * pow(2.0, e - 1) : m / 8.0) * ((c & 32) ? -1 : 1);
          const double ref = e2m3_quant(tv[j]);
          if (val != ref) { ++bad; printf("    lane %d in %g (x scale %g): code %u = %g, host RNE %g\n", l, tv[j], s[l], c, val, ref); }
        }
      printf("(2) %s rounding / saturation / scale-as-divisor: %s\n", which ? "pk32_fp6_f16" : "2xpk16_fp6_f32", bad ? "DIFFERS (above)" : "matches host RNE + clamp 7.5");
      s[0] = s[1] = s[2] = 1.f;
    }
  }
  // (4) end to end
  {
    std::vector<float> xa(64 * 32), xb(64 * 32), d(1024);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int l = 0; l < 64; ++l) {
      const float ra = ldexpf(1.f, (int)(l % 7) - 20), rb = ldexpf(1.f, (int)(l % 5) + 3);   // rows of very different magnitude
      for (int j = 0; j < 32; ++j) { xa[l * 32 + j] = rnd() * ra * (j % 3 == 0 ? 0.1f : 1.f); xb[l * 32 + j] = rnd() * rb; }
    }
    float *da, *db, *dd;
    CK(hipMalloc(&da, xa.size() * 4)); CK(hipMalloc(&db, xb.size() * 4)); CK(hipMalloc(&dd, 4096));
    CK(hipMemcpy(da, xa.data(), xa.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, xb.data(), xb.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mx_mfma_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
    auto quant_row = [&](const float* v, double* q) {
      float am = 0.f;
      for (int j = 0; j < 32; ++j) am = fmaxf(am, fabsf(v[j]));
      const double sc = ldexp(1.0, (int)scale_byte(am) - 127);
      for (int j = 0; j < 32; ++j) q[j] = e2m3_quant(v[j] / sc) * sc;
    };
    double worst = 0, worst_exact = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double ref = 0, exact = 0, mag = 0;
        for (int h = 0; h < 2; ++h) {
          double qa[32], qb[32];
          quant_row(&xa[(m + 32 * h) * 32], qa);
          quant_row(&xb[(n + 32 * h) * 32], qb);
          for (int j = 0; j < 32; ++j) {
            ref += qa[j] * qb[j];
            exact += (double)xa[(m + 32 * h) * 32 + j] * xb[(n + 32 * h) * 32 + j];
            mag += fabs((double)xa[(m + 32 * h) * 32 + j] * xb[(n + 32 * h) * 32 + j]);
          }
        }
        worst = fmax(worst, fabs(d[m * 32 + n] - ref) / (mag + 1e-300));
        worst_exact = fmax(worst_exact, fabs(d[m * 32 + n] - exact) / (mag + 1e-300));
      }
    printf("(4) MFMA fp6 x fp6 with per-lane block scales (dword 6 of the fragment): vs host emulation %.3g, vs exact product %.3g"
           " (relative to sum |a b|; the first must be ~1e-7, the second ~2^-4)\n", worst, worst_exact);
  }
  // (3) issue cost
  {
    std::vector<float> src(1024);
    unsigned st = 99;
    for (auto& v : src) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 65536.0f; }
    float *dsrc, *dout2;
    CK(hipMalloc(&dsrc, 4096)); CK(hipMalloc(&dout2, 4096));
    CK(hipMemcpy(dsrc, src.data(), 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int w = 0; w < 2; ++w)
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (w) hipLaunchKernelGGL((cost_kernel<1>), dim3(256), dim3(256), 0, 0, dsrc, dout2, iters);
        else hipLaunchKernelGGL((cost_kernel<0>), dim3(256), dim3(256), 0, 0, dsrc, dout2, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("(3) %s: %.3f ms for %d iterations, one wave per SIMD -> %.1f cycles per iteration at 2.4 GHz\n",
                        w ? "16 v_max3 + scale + 2xpk16_fp6_f32" : "16 v_max3 only", ms, iters, ms * 1e-3 * 2.4e9 / iters);
      }
  }
  return 0;
}
