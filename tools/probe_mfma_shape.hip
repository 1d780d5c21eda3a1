// Which MFMA shape holds the higher clock at the one-wave conv kernel's operating point?  (MI355X_MICROARCH.md 'DVFS
// give-back' item 7: bare bf16 loops ran 1.12-1.15 x the FLOP/s on 16x16x32 against 32x32x16 at equal cycles per FLOP.)
// The loop below is the conv3x3_f16c8_w4 phase mix without its global traffic: one wave per SIMD, a 128 x 128 output tile
// per wave in 256 accumulator registers, every operand re-read from LDS by ds_read_b128, per 32-channel chunk
//     SHAPE 0: 32 x v_mfma_f32_32x32x16_f16  (16 b128 reads)  + 16 x v_mfma_scale_f32_32x32x64_f8f6f4  e2m3 (16 b128 reads)
//     SHAPE 1: 64 x v_mfma_f32_16x16x32_f16  (16 b128 reads)  + 32 x v_mfma_scale_f32_16x16x128_f8f6f4 e2m3 (16 b128 reads;
//              one MFMA spans two chunks' [lo | hi] blocks)
// i.e. the same flops, the same LDS bytes and the same MFMA cycles per iteration (3,072 per two chunks).  Operands are random
// f16 values / random e2m3 codes with scale bytes around 2^0.  Reported per variant: wall time per launch over the last
// launches of a > 2 s back-to-back series, and the in-kernel clock (s_memtime / s_memrealtime around the loop).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_shape.hip -o /tmp/probe_mfma_shape && /tmp/probe_mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int IMG = 32 * 1024;   // f16 image, then the fp6 image

__device__ __forceinline__ i32x4 lds128(const unsigned char* base, int off) {
  return *reinterpret_cast<const i32x4*>(base + (off & (IMG - 1)));
}
__device__ __forceinline__ i32x8 join(i32x4 a, i32x4 b) {
  i32x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

template <int SHAPE, bool CORR>
__global__ __launch_bounds__(256, 1) void loop_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                                      unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * IMG];
  for (int i = threadIdx.x; i < 2 * IMG / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned char* f16img = smem;
  const unsigned char* fp6img = smem + IMG;
  const int lo = lane * 16;
  // accumulators: 16 sixteen-register tuples in both shapes (the 16 x 16 MFMAs work on quarters of them: as 64 separate
  // f32x4 values hipcc 7.2 shuffled them between VGPRs and AGPRs inside the loop, ~270 v_accvgpr_* per iteration)
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  // fragment of 1 KiB (f16: one b128 per lane) / 2 KiB (fp6: two b128 per lane, kept as an eight-register tuple)
  auto ld16 = [&](int frag, int rot) __attribute__((always_inline)) { return lds128(f16img, frag * 1024 + lo + rot); };
  auto ld6 = [&](int frag, int rot) __attribute__((always_inline)) {
    i32x8 v = join(lds128(fp6img, (frag * 2) * 1024 + lo + rot), lds128(fp6img, (frag * 2 + 1) * 1024 + lo + rot));
    return v;
  };
  // the 16 x 16 shape's 64 four-register accumulators: MFMAs as inline asm with the accumulator TIED ("+a").  Through the
  // builtins hipcc 7.2 gave most MFMAs a destination other than their source accumulator and repaired the loop's back edge
  // with ~270 v_accvgpr_* per iteration (64 three-address MFMAs on four-register tuples do not coalesce).
  f32x4 acc4[8][8];
  if (SHAPE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc4[i][j][q] = 0.f;
  }
  struct Frag6 { i32x6 v; int s; };   // fp6 fragment as the asm wants it: a six-register tuple + the scale dword
  auto ld6a = [&](int frag, int rot) __attribute__((always_inline)) {
    const i32x4 x = lds128(fp6img, (frag * 2) * 1024 + lo + rot);
    const unsigned char* p2 = fp6img + (((frag * 2 + 1) * 1024 + lo + rot) & (IMG - 1));
    const i32x2 y = *reinterpret_cast<const i32x2*>(p2);
    Frag6 f;
    f.v[0] = x[0]; f.v[1] = x[1]; f.v[2] = x[2]; f.v[3] = x[3]; f.v[4] = y[0]; f.v[5] = y[1];
    f.s = *reinterpret_cast<const int*>(p2 + 8);
    return f;
  };
#define SB() __builtin_amdgcn_sched_barrier(0)
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 0) {
    // phases per iteration: [f16 c0 s0] [f16 c0 s1] [fp6 c0] [f16 c1 s0] [f16 c1 s1] [fp6 c1]; every phase requests the NEXT
    // phase's fragments before its own MFMAs (the real kernel's one-phase-ahead reads)
    i32x4 Fa[8], Fb[8];
    i32x8 X[8];
    auto loadF = [&](i32x4 (&F)[8], int ph, int rot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) F[i] = ld16(ph * 8 + i, rot);
    };
    auto loadX = [&](int c, int rot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) X[i] = ld6(c * 8 + i, rot);
    };
    auto mmaF = [&](i32x4 (&F)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, F[i]), __builtin_bit_cast(f16x8, F[4 + j]),
                                                             acc[i][j], 0, 0, 0);
    };
    auto mmaX = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(X[i]));   // stays an eight-register tuple (no copies into six)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(X[i], X[4 + j], acc[i][j], 2, 2, 0, X[i][6], 0, X[4 + j][6]);
    };
    loadF(Fa, 0, 0);
    for (int it = 0; it < iters; ++it) {
      const int rot = (it & 7) * 4096, rotn = ((it + 1) & 7) * 4096;
      SB(); loadF(Fb, 1, rot); mmaF(Fa);
      if (CORR) { SB(); loadX(0, rot); mmaF(Fb); SB(); loadF(Fa, 2, rot); mmaX(); }
      else { SB(); loadF(Fa, 2, rot); mmaF(Fb); }
      SB(); loadF(Fb, 3, rot); mmaF(Fa);
      if (CORR) { SB(); loadX(1, rot); mmaF(Fb); SB(); loadF(Fa, 0, rotn); mmaX(); }
      else { SB(); loadF(Fa, 0, rotn); mmaF(Fb); }
    }
  } else {
    // phases: [f16 c0] [f16 c1] [fp6 pair, columns 0..3] [fp6 pair, columns 4..7].  The MFMAs are volatile asm in program
    // order; the next phase's LDS reads are volatile asm too, one in front of every fourth MFMA (hipcc would issue all of them
    // at the top of the phase: ~18 exposed cycles per read, 78 % MFMA density instead of the 32 x 32 loop's 93 %), and the
    // phase starts with an explicit s_waitcnt lgkmcnt(0) -- a phase later than the requests.
    typedef int i32x3 __attribute__((ext_vector_type(3)));
    i32x4 Ga[16], Gb[16];
    i32x4 XAq[8], XBq[4], XCq[4];   // fp6 fragment = dwords 0..3 (q) + dwords 4, 5 and the scale dword (t)
    i32x3 XAt[8], XBt[4], XCt[4];
    const unsigned sbase = (unsigned)(size_t)smem;   // LDS byte address of the image
    auto rd128 = [&](i32x4& d, int off) __attribute__((always_inline)) {
      const unsigned a = sbase + (unsigned)(off & (IMG - 1));
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a));
#endif
    };
    auto rd96 = [&](i32x3& d, int off) __attribute__((always_inline)) {
      const unsigned a = sbase + IMG + (unsigned)(off & (IMG - 1));
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("ds_read_b96 %0, %1" : "=v"(d) : "v"(a));
#endif
    };
    auto rdG = [&](i32x4 (&G)[16], int k, int c, int rot) __attribute__((always_inline)) { rd128(G[k], (c * 16 + k) * 1024 + lo + rot); };
    auto rdXq = [&](i32x4& q, int frag, int rot) __attribute__((always_inline)) { rd128(q, IMG + (frag * 2) * 1024 + lo + rot); };
    auto rdXt = [&](i32x3& t, int frag, int rot) __attribute__((always_inline)) { rd96(t, (frag * 2 + 1) * 1024 + lo + rot); };
    auto mfmaG = [&](i32x4 (&G)[16], int i, int j) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc4[i][j]) : "v"(G[i]), "v"(G[8 + j]));
#endif
    };
    auto mfmaX = [&](const i32x4& aq, const i32x3& at, const i32x4& bq, const i32x3& bt, int i, int j) __attribute__((always_inline)) {
      i32x6 a6, b6;
      a6[0] = aq[0]; a6[1] = aq[1]; a6[2] = aq[2]; a6[3] = aq[3]; a6[4] = at[0]; a6[5] = at[1];
      b6[0] = bq[0]; b6[1] = bq[1]; b6[2] = bq[2]; b6[3] = bq[3]; b6[4] = bt[0]; b6[5] = bt[1];
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                   : "+a"(acc4[i][j])
                   : "v"(a6), "v"(b6), "v"(at[2]), "v"(bt[2]));
#endif
    };
    auto wait0 = [&]() __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    };
#pragma unroll
    for (int k = 0; k < 16; ++k) rdG(Ga, k, 0, 0);
    for (int it = 0; it < iters; ++it) {
      const int rot = (it & 7) * 4096, rotn = ((it + 1) & 7) * 4096;
      // P0: f16 chunk 0 from Ga; requests Gb (16 reads)
      wait0();
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        if (m % 4 == 0) rdG(Gb, m / 4, 1, rot);
        mfmaG(Ga, m / 8, m % 8);
      }
      if (CORR) {
        // P1: f16 chunk 1 from Gb; requests XA (8 x 2 reads) and XB (4 x 2): 24 reads
        wait0();
#pragma unroll
        for (int m = 0; m < 64; ++m) {
          if (m % 2 == 0 && m / 2 < 24) {
            const int k = m / 2;
            if (k < 8) rdXq(XAq[k], k, rot);
            else if (k < 16) rdXt(XAt[k - 8], k - 8, rot);
            else if (k < 20) rdXq(XBq[k - 16], 8 + k - 16, rot);
            else rdXt(XBt[k - 20], 8 + k - 20, rot);
          }
          mfmaG(Gb, m / 8, m % 8);
        }
        // P2: fp6, columns 0..3 from XA, XB; requests XC (8 reads)
        wait0();
#pragma unroll
        for (int m = 0; m < 32; ++m) {
          if (m % 4 == 0) {
            const int k = m / 4;
            if (k < 4) rdXq(XCq[k], 12 + k, rot);
            else rdXt(XCt[k - 4], 12 + k - 4, rot);
          }
          mfmaX(XAq[m / 4], XAt[m / 4], XBq[m % 4], XBt[m % 4], m / 4, m % 4);
        }
        // P3: fp6, columns 4..7 from XA, XC; requests the next iteration's Ga (16 reads)
        wait0();
#pragma unroll
        for (int m = 0; m < 32; ++m) {
          if (m % 2 == 0) rdG(Ga, m / 2, 0, rotn);
          mfmaX(XAq[m / 4], XAt[m / 4], XCq[m % 4], XCt[m % 4], m / 4, 4 + m % 4);
        }
      } else {
        wait0();
#pragma unroll
        for (int m = 0; m < 64; ++m) {
          if (m % 4 == 0) rdG(Ga, m / 4, 0, rotn);
          mfmaG(Gb, m / 8, m % 8);
        }
      }
    }
    wait0();
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) t += acc[i][j][q];
  if (SHAPE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) t += acc4[i][j][q];
  }
  out[blockIdx.x * 256 + threadIdx.x] = t;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = c1 - c0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

static unsigned short f16_bits(float v) {   // round-to-nearest of a small normal value (no subnormals / overflow here)
  unsigned u;
  memcpy(&u, &v, 4);
  const unsigned s = (u >> 16) & 0x8000u;
  int e = (int)((u >> 23) & 0xff) - 127 + 15;
  unsigned m = u & 0x7fffffu;
  if (e <= 0) return (unsigned short)s;
  unsigned r = (m + 0xfffu + ((m >> 13) & 1u)) >> 13;
  if (r == 0x400u) { r = 0; ++e; }
  return (unsigned short)(s | ((unsigned)e << 10) | r);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  int ncu = 0;
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<unsigned char> h(2 * IMG);
  srand(12345);
  auto rnd = []() { return (rand() & 0xffff) / 65536.0f; };
  for (int i = 0; i < IMG / 2; ++i) {   // roughly normal, |v| < 3
    const float v = (rnd() + rnd() + rnd() + rnd() - 2.0f) * 1.7f;
    const unsigned short b = f16_bits(v);
    memcpy(&h[2 * i], &b, 2);
  }
  // fp6 image: [fragment][piece 0 / 1][lane][16 B]; a lane's 32 bytes = 24 code bytes, scale byte (dword 6), pad
  for (int f = 0; f < 16; ++f)
    for (int l = 0; l < 64; ++l) {
      unsigned char rec[32];
      for (int b = 0; b < 24; ++b) rec[b] = (unsigned char)(rand() & 0xff);
      memset(rec + 24, 0, 8);
      rec[24] = (unsigned char)(120 + (rand() & 7));
      memcpy(&h[IMG + (f * 2) * 1024 + l * 16], rec, 16);
      memcpy(&h[IMG + (f * 2 + 1) * 1024 + l * 16], rec + 16, 16);
    }
  uint4* src;
  float* out;
  unsigned long long* stamps;
  CK(hipMalloc(&src, 2 * IMG));
  CK(hipMalloc(&out, (size_t)ncu * 256 * 4));
  CK(hipMalloc(&stamps, (size_t)ncu * 16));
  CK(hipMemcpy(src, h.data(), 2 * IMG, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 12000;   // ~ 20 ms per launch
  const double flop_iter = 2.0 * 128 * 128 * 64 * 4 * ncu;   // f16 part: two chunks x 32 channels, 4 waves, all CUs
  struct Var { const char* name; void (*k)(const uint4*, float*, int, unsigned long long*); bool corr; };
  const Var vars[] = {
      {"32x32 f16+fp6", loop_kernel<0, true>, true},   {"16x16 f16+fp6", loop_kernel<1, true>, true},
      {"32x32 f16 only", loop_kernel<0, false>, false}, {"16x16 f16 only", loop_kernel<1, false>, false}};
  for (int rd = 0; rd < rounds; ++rd)
    for (const Var& v : vars) {
      const int nl = 110;
      std::vector<float> ms;
      for (int i = 0; i < nl; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(v.k, dim3(ncu), dim3(256), 0, 0, src, out, iters, stamps);
        CK(hipEventRecord(e1, 0));
        if (i >= nl - 30) {
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          ms.push_back(t);
        }
      }
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> st(ncu * 2);
      CK(hipMemcpy(st.data(), stamps, (size_t)ncu * 16, hipMemcpyDeviceToHost));
      std::vector<double> clk, cyc;
      for (int b = 0; b < ncu; ++b) {
        clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
        cyc.push_back((double)st[2 * b] / iters);
      }
      std::sort(clk.begin(), clk.end());
      std::sort(cyc.begin(), cyc.end());
      std::sort(ms.begin(), ms.end());
      const double med = ms[ms.size() / 2];
      const double units = v.corr ? 2.64 / 2.0 : 1.0;   // work in f16-MFMA units relative to the f16 part
      printf("round %d  %-15s  %7.3f ms/launch  f16-part %7.1f TFLOP/s (x%.2f units of matrix work)  cycles/iter %7.1f  clock %.3f GHz\n",
             rd, v.name, med, flop_iter * iters / med / 1e9, units, cyc[ncu / 2], clk[ncu / 2]);
      fflush(stdout);
    }
  return 0;
}
