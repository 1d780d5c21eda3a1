// Bottleneck self-attention (AttnBlock, eo_vae/models/modules/layers.py:128-142):
//   one head, d = C = 512, 1024 tokens at 256x256 input, softmax(Q^T K / sqrt(C)) V.
// q, k, v arrive as NCHW conv outputs [B][C][HW] (token index contiguous), so all five
// contractions of forward+backward are strided-batched GEMMs on the fp32 matrix cores:
//     S  = scale * Q^T K          P = softmax_rows(S)          O  = V P^T
//     dV = dO P      dP = dO^T V      dS = P o (dP - rowsum(dP o P))
//     dQ = scale * K dS^T             dK = scale * Q dS
// The score matrix is materialised (4 MB / image at 1024 tokens).
#include "eovae_common.h"

namespace {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long long sam, sak, sbk, sbn, scm, scn;  // element strides
  long long batA, batB, batC;
  float alpha;
  int tiles_m, tiles_n;
};

// C[b](m,n) = alpha * sum_k A[b](m,k) * B[b](k,n); 128x128 block tile, 4 waves x (64x64),
// K streamed in chunks of 16 through LDS images As[k][m], Bs[k][n] (conflict-free b32 reads).
template <bool A_KCONTIG, bool B_KCONTIG>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128, KC = 16, LD = 129;
  __shared__ float As[KC * LD];
  __shared__ float Bs[KC * LD];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  int b = blockIdx.x;
  const int tm = b % g.tiles_m;
  b /= g.tiles_m;
  const int tn = b % g.tiles_n;
  const int bat = b / g.tiles_n;
  const float* A = g.A + bat * g.batA;
  const float* B = g.B + bat * g.batB;
  float* Cp = g.C + bat * g.batC;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ar[8], br[8];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int e = tid + i * 256;
      int m, k;
      if (A_KCONTIG) { m = e / KC; k = e % KC; } else { k = e / BM; m = e % BM; }
      float v = 0.f;
      if (m0 + m < g.M && k0 + k < g.K) v = A[(long long)(m0 + m) * g.sam + (long long)(k0 + k) * g.sak];
      ar[i] = v;
      int n, kb;
      if (B_KCONTIG) { n = e / KC; kb = e % KC; } else { kb = e / BN; n = e % BN; }
      float w = 0.f;
      if (n0 + n < g.N && k0 + kb < g.K) w = B[(long long)(k0 + kb) * g.sbk + (long long)(n0 + n) * g.sbn];
      br[i] = w;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int e = tid + i * 256;
      int m, k;
      if (A_KCONTIG) { m = e / KC; k = e % KC; } else { k = e / BM; m = e % BM; }
      As[k * LD + m] = ar[i];
      int n, kb;
      if (B_KCONTIG) { n = e / KC; kb = e % KC; } else { kb = e / BN; n = e % BN; }
      Bs[kb * LD + n] = br[i];
    }
  };

  const int nchunks = (g.K + KC - 1) / KC;
  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      const int k = 2 * kk + hi;
      const float a0 = As[k * LD + wm + l31], a1 = As[k * LD + wm + 32 + l31];
      const float b0 = Bs[k * LD + wn + l31], b1 = Bs[k * LD + wn + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int n = n0 + wn + j * 32 + l31;
        if (m < g.M && n < g.N) Cp[(long long)m * g.scm + (long long)n * g.scn] = acc[i][j][r] * g.alpha;
      }
}

// in-place row softmax; one wave per row, wave64 shuffle reductions
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, long long rows,
                                                           int n) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* p = s + row * n;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, p[i]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) {
    float e = __expf(p[i] - mx);
    p[i] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int i = lane; i < n; i += 64) p[i] *= inv;
}

// dS = P o (dP - sum_j dP o P), written over dP
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p,
                                                               float* __restrict__ dp,
                                                               long long rows, int n) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* pr = p + row * n;
  float* dr = dp + row * n;
  float dot = 0.f;
  for (int i = lane; i < n; i += 64) dot += pr[i] * dr[i];
  dot = wave_sum(dot);
  for (int i = lane; i < n; i += 64) dr[i] = pr[i] * (dr[i] - dot);
}

}  // namespace

extern "C" {

// Strided-batched fp32 GEMM, C = alpha * op(A) op(B).  Each of A and B must have unit stride
// along one of its two dimensions (that dimension is what the loads coalesce along).
int eovae_gemm_f32(const float* A, const float* B, float* Cm, int batch, int M, int N, int K,
                   long long sam, long long sak, long long sbk, long long sbn, long long scm,
                   long long scn, long long batA, long long batB, long long batC, float alpha,
                   void* stream) {
  EOVAE_CHECK_ARG(A && B && Cm, "gemm_f32: null pointer");
  EOVAE_CHECK_ARG(batch > 0 && M > 0 && N > 0 && K > 0, "gemm_f32: bad shape");
  EOVAE_CHECK_ARG(sam == 1 || sak == 1, "gemm_f32: A needs a unit stride");
  EOVAE_CHECK_ARG(sbk == 1 || sbn == 1, "gemm_f32: B needs a unit stride");
  GemmArgs g;
  g.A = A; g.B = B; g.C = Cm; g.M = M; g.N = N; g.K = K;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.scm = scm; g.scn = scn;
  g.batA = batA; g.batB = batB; g.batC = batC; g.alpha = alpha;
  g.tiles_m = cdiv(M, 128); g.tiles_n = cdiv(N, 128);
  dim3 grid(batch * g.tiles_m * g.tiles_n);
  hipStream_t st = (hipStream_t)stream;
  const bool ak = (sak == 1 && sam != 1), bk = (sbk == 1 && sbn != 1);
  if (ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, g);
  else if (ak) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, g);
  else if (bk) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, g);
  EOVAE_CHECK_LAUNCH("eovae_gemm_f32");
  return EOVAE_OK;
}

int eovae_softmax_rows(float* s, long long rows, int n, void* stream) {
  EOVAE_CHECK_ARG(s && rows > 0 && n > 0, "softmax_rows: bad argument");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, s, rows, n);
  EOVAE_CHECK_LAUNCH("eovae_softmax_rows");
  return EOVAE_OK;
}

int eovae_softmax_bwd_rows(const float* p, float* dp, long long rows, int n, void* stream) {
  EOVAE_CHECK_ARG(p && dp && rows > 0 && n > 0, "softmax_bwd_rows: bad argument");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, p, dp, rows, n);
  EOVAE_CHECK_LAUNCH("eovae_softmax_bwd_rows");
  return EOVAE_OK;
}

int eovae_gemm_bf16x3(const float* A, const float* B, float* C, const float* bias, const float* res,
                      int batch, int M, int N, int K, long long sam, long long sak, long long sbk,
                      long long sbn, long long scm, long long scn, long long batA, long long batB,
                      long long batC, float alpha, int ksplit, int atomic, int precision,
                      int kb, long long kbA, long long kbB, void* stream);

// precision 0: exact fp32 MFMA GEMM; 1 / 3: bf16 / split-bf16 GEMM (csrc/gemm_bf16x3.hip)
static int gemm_any(int precision, const float* A, const float* B, float* Cm, int batch, int M, int N,
                    int K, long long sam, long long sak, long long sbk, long long sbn, long long scm,
                    long long scn, long long batA, long long batB, long long batC, float alpha,
                    void* stream) {
  if (precision == 0)
    return eovae_gemm_f32(A, B, Cm, batch, M, N, K, sam, sak, sbk, sbn, scm, scn, batA, batB, batC,
                          alpha, stream);
  return eovae_gemm_bf16x3(A, B, Cm, nullptr, nullptr, batch, M, N, K, sam, sak, sbk, sbn, scm, scn,
                           batA, batB, batC, alpha, 1, 0, precision, 1, 0, 0, stream);
}

// Fused-by-sequence attention forward: P (B x T x T, caller-allocated, kept for backward) and
// O[b][c][t].  q, k, v, o: [B][C][T] contiguous.
int eovae_attention_fwd(const float* q, const float* k, const float* v, float* p, float* o,
                        int B, int Cc, int T, float scale, int precision, void* stream) {
  long long ct = (long long)Cc * T, tt = (long long)T * T;
  int rc = gemm_any(precision, q, k, p, B, T, T, Cc, 1, T, T, 1, T, 1, ct, ct, tt, scale, stream);
  if (rc) return rc;
  rc = eovae_softmax_rows(p, (long long)B * T, T, stream);
  if (rc) return rc;
  // O(m=c, n=i) = sum_j V(c,j) P(i,j)
  return gemm_any(precision, v, p, o, B, Cc, T, T, T, 1, 1, T, T, 1, ct, tt, ct, 1.f, stream);
}

// Backward: dq, dk, dv from do; dp is B x T x T scratch (overwritten with dS).
int eovae_attention_bwd(const float* q, const float* k, const float* v, const float* p,
                        const float* d_o, float* dp, float* dq, float* dk, float* dv, int B, int Cc,
                        int T, float scale, int precision, void* stream) {
  long long ct = (long long)Cc * T, tt = (long long)T * T;
  // dV(m=c, n=j) = sum_i dO(c,i) P(i,j)
  int rc = gemm_any(precision, d_o, p, dv, B, Cc, T, T, T, 1, T, 1, T, 1, ct, tt, ct, 1.f, stream);
  if (rc) return rc;
  // dP(m=i, n=j) = sum_c dO(c,i) V(c,j)
  rc = gemm_any(precision, d_o, v, dp, B, T, T, Cc, 1, T, T, 1, T, 1, ct, ct, tt, 1.f, stream);
  if (rc) return rc;
  rc = eovae_softmax_bwd_rows(p, dp, (long long)B * T, T, stream);
  if (rc) return rc;
  // dQ(m=c, n=i) = scale * sum_j K(c,j) dS(i,j)
  rc = gemm_any(precision, k, dp, dq, B, Cc, T, T, T, 1, 1, T, T, 1, ct, tt, ct, scale, stream);
  if (rc) return rc;
  // dK(m=c, n=j) = scale * sum_i Q(c,i) dS(i,j)
  return gemm_any(precision, q, dp, dk, B, Cc, T, T, T, 1, T, 1, T, 1, ct, tt, ct, scale, stream);
}

}  // extern "C"
