// v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands: (1) operand / result layout -- lane (r = lane % 16, g = lane / 16)
// supplies the 32 K values of K block g of row / column r; D[4 g + q][r] in result register q -- checked by an exact integer
// product; (2) which lane's scale byte reaches which (row, K block): one lane at a time gets 2^1, all-ones operands.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mx16.hip -o /tmp/probe_mx16 && /tmp/probe_mx16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__host__ __device__ inline unsigned e2m3_code(int v) {
  const unsigned s = v < 0 ? 32u : 0u;
  const int a = v < 0 ? -v : v;
  unsigned c = 0;
  if (a == 1) c = 8; else if (a == 2) c = 16; else if (a == 3) c = 20; else if (a == 4) c = 24; else if (a == 6) c = 28;
  return s | c;
}
__device__ inline i32x8 pack_fp6(const int* v) {
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
__global__ void k(const int* la, const int* lb, const int* sa, const int* sb, float* d) {
  const int lane = threadIdx.x;
  const i32x8 a = pack_fp6(la + lane * 32), b = pack_fp6(lb + lane * 32);
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa[lane], 0, sb[lane]);
  const int n = lane & 15, g = lane >> 4;
  for (int q = 0; q < 4; ++q) d[(4 * g + q) * 16 + n] = c[q];
}
static int irand(unsigned& st) { st = st * 1664525u + 1013904223u; static const int tab[9] = {0, 1, -1, 2, -2, 3, -3, 4, -4}; return tab[(st >> 10) % 9]; }
int main() {
  std::vector<int> la(64 * 32), lb(64 * 32), sa(64, 127), sb(64, 127);
  std::vector<float> d(256);
  unsigned st = 4242;
  for (auto& v : la) v = irand(st);
  for (auto& v : lb) v = irand(st);
  int *dla, *dlb, *dsa, *dsb; float* dd;
  CK(hipMalloc(&dla, la.size() * 4)); CK(hipMalloc(&dlb, lb.size() * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 1024));
  auto run = [&]() {
    CK(hipMemcpy(dla, la.data(), la.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlb, lb.data(), lb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dla, dlb, dsa, dsb, dd);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
  };
  run();
  int bad = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double ref = 0;
      for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 32; ++j) ref += (double)la[(m + 16 * g) * 32 + j] * lb[(n + 16 * g) * 32 + j];
      if (fabs(d[m * 16 + n] - ref) > 1e-6 * fabs(ref) + 1e-9) ++bad;
    }
  printf("layout lane (r, g) = K block g of row r, D[4 g + q][r]: %s (%d / 256 mismatches)\n", bad ? "MISMATCH" : "exact", bad);
  for (auto& v : la) v = 1;
  for (auto& v : lb) v = 1;
  for (int side = 0; side < 2; ++side) {
    printf("-- scale of %s: lane -> rows x cols of D changed, value (128 = untouched; +32 = one K block doubled)\n", side ? "B" : "A");
    for (int L = 0; L < 64; L += (L % 16 < 2 || L % 16 == 15) ? 1 : 13) {
      std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
      (side ? sb : sa)[L] = 128;
      run();
      int nrow = 0, ncol = 0, r0 = -1, c0 = -1; float val = 128.f;
      for (int m = 0; m < 16; ++m) { bool ch = false; for (int n = 0; n < 16; ++n) if (d[m * 16 + n] != 128.f) { ch = true; val = d[m * 16 + n]; } if (ch) { ++nrow; if (r0 < 0) r0 = m; } }
      for (int n = 0; n < 16; ++n) { bool ch = false; for (int m = 0; m < 16; ++m) if (d[m * 16 + n] != 128.f) ch = true; if (ch) { ++ncol; if (c0 < 0) c0 = n; } }
      printf("   lane %2d: %2d rows (first %2d) x %2d cols (first %2d) changed, value %g\n", L, nrow, r0, ncol, c0, val);
    }
  }
  // byte 1..3 of the scale dword (op_sel 0 reads byte 0): does any other byte matter?
  std::fill(sa.begin(), sa.end(), 127 | (130 << 8) | (131 << 16) | (132u << 24)); std::fill(sb.begin(), sb.end(), 127);
  run();
  printf("scale dword with other bytes set: D[0][0] = %g (128 = only byte 0 is read)\n", d[0]);
  return 0;
}
