// Strided-batched GEMM with split-bf16 operands on the bf16 matrix cores:
//     C[b](m,n) (+)= alpha * sum_k A[b](m,k) * B[b](k,n)  (+ bias[m]) (+ res[b](m,n))
// A and B are fp32 in global memory with arbitrary element strides (one unit stride each); they
// are split to hi/lo bf16 while being staged into LDS as [k-step][row][16 k] images (32-byte
// rows: an MFMA fragment is one aligned ds_read_b128, 32 rows = 1 KiB contiguous).
// Serves every 1x1 convolution of the path (nin_shortcut, attention q/k/v/proj_out, quant convs:
// layers.py:84-87,123-126; model.py:165,236) -- forward, data gradient and (atomic, batch-reduced)
// weight gradient -- and the five attention contractions (layers.py:133-139).
// 1x1 convolutions are L2/HBM-bandwidth bound at these channel counts (<= 64 flop/byte).
#include "eovae_common.h"
#include <type_traits>

namespace {

struct GemmBArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  int M, N, K;
  long long sam, sak, sbk, sbn, scm, scn;
  long long batA, batB, batC;
  float alpha;
  int tiles_m, tiles_n, ksplit, atomic;
  // K may be extended over `kb` equally-strided slabs (e.g. the images of a batch for a 1x1
  // weight gradient): k_total = kb * K, slab j of A/B starts at j*kbA / j*kbB
  int kb;
  long long kbA, kbB;
  int vec;  // every k-contiguous operand: base, row / batch / slab strides multiples of 16 bytes
};

__device__ __forceinline__ void split8g(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

#ifndef EOVAE_GEMM_MINB
#define EOVAE_GEMM_MINB 4
#endif
template <bool A_KC, bool B_KC, int NTERMS>
__global__ __launch_bounds__(512, EOVAE_GEMM_MINB) void gemm_bf16_kernel(GemmBArgs g) {
  constexpr int BM = 128, BN = 256, KS = 32;
  constexpr int A_IMG = 2 * BM * 32, B_IMG = 2 * BN * 32;  // bytes of one (hi or lo) stage image
  constexpr int OFF_ALO = A_IMG, OFF_B = 2 * A_IMG, OFF_BLO = 2 * A_IMG + B_IMG;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_IMG + 2 * B_IMG];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int tm = b % g.tiles_m;
  b /= g.tiles_m;
  const int tn = b % g.tiles_n;
  b /= g.tiles_n;
  const int ks_id = b % g.ksplit;
  const int bat = b / g.ksplit;
  const float* A = g.A + bat * g.batA;
  const float* B = g.B + bat * g.batB;
  float* C = g.C + bat * g.batC;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  // K range of this block (multiples of the 32-deep stage; stages never straddle a slab)
  const int stages_slab = (g.K + KS - 1) / KS;
  const int stages_total = stages_slab * g.kb;
  const int per = (stages_total + g.ksplit - 1) / g.ksplit;
  const int s_beg = ks_id * per;
  int s_end = s_beg + per;
  if (s_end > stages_total) s_end = stages_total;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  float ar[8], br[2][8];
  // task -> (row, octet) maps chosen so the global loads coalesce along the contiguous axis
  auto a_task = [&](int t, int& m, int& oct) {
    if (A_KC) { m = t >> 2; oct = t & 3; } else { m = t & (BM - 1); oct = t >> 7; }
  };
  auto b_task = [&](int t, int& n, int& oct) {
    if (B_KC) { n = t >> 2; oct = t & 3; } else { n = t & (BN - 1); oct = t >> 8; }
  };
  // Row validity is stage-invariant: rows beyond M / N load a clamped (valid) row and are zeroed
  // when the stage is converted.  With K a multiple of the 32-deep stage (every GEMM of the model)
  // all loads are then UNCONDITIONAL; predicated loads compile to an exec-mask branch per load with
  // vmcnt(0) drains in between (see conv_bf16x3.hip), exposing the memory latency over and over.
  int am, aoct, bn_[2], boct[2];
  a_task(tid, am, aoct);
  const bool okm = m0 + am < g.M;
  const long long arow = (long long)(okm ? m0 + am : g.M - 1) * g.sam;
  bool okn[2];
  long long brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    b_task(tid + i * 512, bn_[i], boct[i]);
    okn[i] = n0 + bn_[i] < g.N;
    brow[i] = (long long)(okn[i] ? n0 + bn_[i] : g.N - 1) * g.sbn;
  }
  // MODE 0: ragged K (predicated); 1: K % 32 == 0; 2: additionally every k-contiguous operand is
  // 16-byte aligned, so its 8 consecutive k values are two dwordx4 loads instead of eight dword loads
  // that each touch the same cache lines again
  // row-contiguous operands (unit row stride): 32-bit lane offset of the (clamped) row
  const unsigned arow32 = (unsigned)(okm ? m0 + am : g.M - 1);
  const unsigned brow32[2] = {(unsigned)(okn[0] ? n0 + bn_[0] : g.N - 1), (unsigned)(okn[1] ? n0 + bn_[1] : g.N - 1)};
  auto load_stage = [&](int s, auto mode_t) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_t)::value;
    constexpr bool FULLK = MODE >= 1, VEC = MODE == 2;
    const int slab = s / stages_slab;
    const int k0 = (s - slab * stages_slab) * KS;
    const float* As_ = A + slab * g.kbA;
    const float* Bs_ = B + slab * g.kbB;
    if constexpr (A_KC) {  // k contiguous: one per-lane address, 8 consecutive floats
      const float* p = As_ + arow + (k0 + aoct * 8);
      if (VEC) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ar[j] = v0[j]; ar[4 + j] = v1[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) ar[j] = (FULLK || k0 + aoct * 8 + j < g.K) ? p[j] : 0.f;
      }
    } else {  // rows contiguous: the octet is wave-uniform => scalar base per k + 32-bit lane offset
      const int octu = __builtin_amdgcn_readfirstlane(aoct);
      const float* pu = As_ + (long long)(k0 + octu * 8) * g.sak;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        ar[j] = (FULLK || k0 + octu * 8 + j < g.K) ? pu[(long long)j * g.sak + arow32] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (B_KC) {
        const float* p = Bs_ + brow[i] + (k0 + boct[i] * 8);
        if (VEC) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { br[i][j] = v0[j]; br[i][4 + j] = v1[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) br[i][j] = (FULLK || k0 + boct[i] * 8 + j < g.K) ? p[j] : 0.f;
        }
      } else {
        const int octu = __builtin_amdgcn_readfirstlane(boct[i]);
        const float* pu = Bs_ + (long long)(k0 + octu * 8) * g.sbk;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          br[i][j] = (FULLK || k0 + octu * 8 + j < g.K) ? pu[(long long)j * g.sbk + brow32[i]] : 0.f;
      }
    }
  };
  auto store_stage = [&]() {
    {
      int m, oct;
      a_task(tid, m, oct);
      bf16x8 hi, lo;
      if (!okm) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ar[j] = 0.f;
      }
      split8g(ar, hi, lo);
      const int off = (oct >> 1) * (BM * 32) + m * 32 + (oct & 1) * 16;
      *reinterpret_cast<bf16x8*>(smem + off) = hi;
      if (NTERMS == 3) *reinterpret_cast<bf16x8*>(smem + OFF_ALO + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n, oct;
      b_task(tid + i * 512, n, oct);
      bf16x8 hi, lo;
      if (!okn[i]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) br[i][j] = 0.f;
      }
      split8g(br[i], hi, lo);
      const int off = (oct >> 1) * (BN * 32) + n * 32 + (oct & 1) * 16;
      *reinterpret_cast<bf16x8*>(smem + OFF_B + off) = hi;
      if (NTERMS == 3) *reinterpret_cast<bf16x8*>(smem + OFF_BLO + off) = lo;
    }
  };

  // one instantiation of the stage loop per load path (no phi join between the loads and their use)
  auto run = [&](auto full_t) __attribute__((always_inline)) {
  if (s_beg < s_end) load_stage(s_beg, full_t);
  for (int s = s_beg; s < s_end; ++s) {
    __syncthreads();
    store_stage();
    __syncthreads();
    if (s + 1 < s_end) load_stage(s + 1, full_t);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // B fragments one column tile at a time: 8 fewer live VGPRs than holding both (the kernel is
      // capped at 128 VGPRs for two blocks per CU, and a spilled address is reloaded through the same
      // in-order vmcnt queue as the operand loads)
      bf16x8 ahi[2], alo[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = kk * (BM * 32) + (wm + i * 32 + r) * 32 + h * 16;
        ahi[i] = *reinterpret_cast<const bf16x8*>(smem + off);
        if (NTERMS == 3) alo[i] = *reinterpret_cast<const bf16x8*>(smem + OFF_ALO + off);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = kk * (BN * 32) + (wn + j * 32 + r) * 32 + h * 16;
        const bf16x8 bhi = *reinterpret_cast<const bf16x8*>(smem + OFF_B + off);
        bf16x8 blo;
        if (NTERMS == 3) blo = *reinterpret_cast<const bf16x8*>(smem + OFF_BLO + off);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (NTERMS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bhi, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], blo, acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bhi, acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  };
  if (g.K % KS != 0) run(std::integral_constant<int, 0>{});
  else if ((A_KC || B_KC) && g.vec) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 1>{});

  // Epilogue in batches of 4 rows x 2 column tiles: bias / residual by unconditional loads from clamped
  // indices, then the batch is stored (plain, atomic, or masked at the tensor edge).  The per-element
  // form (`bias ? bias[m] : 0`, `if (res) v += res[o]`) put an s_waitcnt vmcnt(0) in front of every store.
  const bool lead = ks_id == 0;
  const float* biasp = lead ? g.bias : nullptr;
  const float* resp = (lead && g.res) ? g.res + bat * g.batC : nullptr;
  auto epilogue = [&](auto full_t) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_t)::value;
    long long ncol[2];
    bool nok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 32 + r;
      nok[j] = FULL || n < g.N;
      ncol[j] = (long long)(nok[j] ? n : g.N - 1) * g.scn;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float bias4[4], rv[4][2], v[4][2];
        long long mrow[4];
        bool mok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = m0 + wm + i * 32 + k + 8 * qq + 4 * h;
          mok[k] = FULL || m < g.M;
          const int mc = mok[k] ? m : g.M - 1;
          mrow[k] = (long long)mc * g.scm;
          bias4[k] = biasp ? biasp[mc] : 0.f;
        }
        if (resp) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) rv[k][j] = resp[mrow[k] + ncol[j]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            v[k][j] = acc[i][j][4 * qq + k] * g.alpha + bias4[k];
            if (resp) v[k][j] += rv[k][j];
            if (!FULL) asm volatile("" : "+v"(v[k][j]));
          }
        if (g.atomic) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (FULL || (mok[k] && nok[j])) atomicAdd(&C[mrow[k] + ncol[j]], v[k][j]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (FULL || (mok[k] && nok[j])) C[mrow[k] + ncol[j]] = v[k][j];
        }
      }
  };
  if (m0 + BM <= g.M && n0 + BN <= g.N) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

template <int NT>
void launch(const GemmBArgs& g, dim3 grid, hipStream_t st, bool ak, bool bk) {
  if (ak && bk) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, NT>), grid, dim3(512), 0, st, g);
  else if (ak) hipLaunchKernelGGL((gemm_bf16_kernel<true, false, NT>), grid, dim3(512), 0, st, g);
  else if (bk) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, NT>), grid, dim3(512), 0, st, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, false, NT>), grid, dim3(512), 0, st, g);
}

}  // namespace

extern "C" {

// kb > 1 extends K over kb slabs (strides kbA / kbB elements).
// precision 3 = bf16x3, 1 = bf16.  atomic = 1: C += (caller pre-zeroes C; required when ksplit > 1
// or when several batches share C through batC = 0).
int eovae_gemm_bf16x3(const float* A, const float* B, float* C, const float* bias, const float* res,
                      int batch, int M, int N, int K, long long sam, long long sak, long long sbk,
                      long long sbn, long long scm, long long scn, long long batA, long long batB,
                      long long batC, float alpha, int ksplit, int atomic, int precision,
                      int kb, long long kbA, long long kbB, void* stream) {
  EOVAE_CHECK_ARG(A && B && C, "gemm_bf16x3: null pointer");
  EOVAE_CHECK_ARG(kb >= 1, "gemm_bf16x3: kb %d", kb);
  EOVAE_CHECK_ARG(batch > 0 && M > 0 && N > 0 && K > 0, "gemm_bf16x3: bad shape");
  EOVAE_CHECK_ARG(sam == 1 || sak == 1, "gemm_bf16x3: A needs a unit stride");
  EOVAE_CHECK_ARG(sbk == 1 || sbn == 1, "gemm_bf16x3: B needs a unit stride");
  EOVAE_CHECK_ARG(precision == 1 || precision == 3, "gemm_bf16x3: precision %d", precision);
  EOVAE_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || atomic), "gemm_bf16x3: ksplit > 1 needs atomic");
  EOVAE_CHECK_ARG(batC != 0 || batch == 1 || atomic, "gemm_bf16x3: shared C needs atomic");
  GemmBArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.res = res;
  g.M = M; g.N = N; g.K = K;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.scm = scm; g.scn = scn;
  g.batA = batA; g.batB = batB; g.batC = batC; g.alpha = alpha;
  g.tiles_m = cdiv(M, 128); g.tiles_n = cdiv(N, 256); g.ksplit = ksplit; g.atomic = atomic;
  g.kb = kb; g.kbA = kbA; g.kbB = kbB;
  dim3 grid((unsigned)(batch * ksplit * g.tiles_m * g.tiles_n));
  const bool ak = (sak == 1 && sam != 1), bk = (sbk == 1 && sbn != 1);
  auto al16 = [](const float* p, long long row, long long bat, long long slab) {
    return ((uintptr_t)p % 16 == 0) && row % 4 == 0 && bat % 4 == 0 && slab % 4 == 0;
  };
  g.vec = (!ak || al16(A, sam, batA, kbA)) && (!bk || al16(B, sbn, batB, kbB));
  if (precision == 3) launch<3>(g, grid, (hipStream_t)stream, ak, bk);
  else launch<1>(g, grid, (hipStream_t)stream, ak, bk);
  EOVAE_CHECK_LAUNCH("eovae_gemm_bf16x3");
  return EOVAE_OK;
}

}  // extern "C"
