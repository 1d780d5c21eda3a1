// Kernels of the wavelength-conditioned weight generators (reference: eo_vae/models/modules/dynamic_conv.py:62-183,
// 336-366: sin-cos embedding -> FCResLayer -> nn.TransformerEncoder (d 256, 4 heads, FFN 2048, GELU, post-norm) over
// [128 weight tokens | C wavelength tokens | 1 bias token] -> fc_weight / fc_bias).
//
// The generators are < 0.3 GFLOP and cost a training step ~1.5-2.2 ms all the same (profiles/r04q_generator_cost.log): on
// library GEMMs + ATen glue they are ~330 launches of 7-10 us each, and the encoder's generator is the head (forward)
// and the tail (backward) of every step, where nothing overlaps it.  Here the whole token matrix ([<= 160, 256]) is a
// handful of small exact-fp32 kernels per layer whose epilogues carry the glue (bias, GELU and its derivative, residual,
// split-K partial sums folded by the LayerNorm that follows), so a layer is 9 launches forward and ~22 backward.
// Measured: as fast as the library path in the step -- a dependent tiny launch costs >= 5 us of stream
// time whatever it computes -- so this file buys ownership of the arithmetic (no BLAS library in the step), not time.
//
//   eovae_gen_gemm      C[b](m, n) = epi(alpha * sum_k A[b](m, k) * B[b](n, k))   strided, batched, optional split-K
//   eovae_gen_ln_fwd    y = LayerNorm(sum of split-K partials + bias + residual); keeps x_hat and 1/sigma
//   eovae_gen_ln_bwd    dx of that LayerNorm (rows) and d_gamma / d_beta (columns)
//   eovae_gen_colsum    column sums of a [rows, n] matrix (bias gradients)
// fp32 FMA arithmetic throughout (no matrix cores: at 132 rows the problems are latency-, not throughput-bound, and the
// generated kernels feed the parity-critical first / last convolutions).
#include "eovae_common.h"

namespace {

constexpr int GBM = 32, GBN = 64, GBK = 128;
constexpr int APITCH = GBM + 2;   // even: the two rows of a thread are one ds_read_b64
constexpr int BPITCH = GBN + 4;   // multiple of 4: the four columns of a thread are one ds_read_b128

struct GenGemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;      // [N] or null (epilogues 0, 1)
  const float* aux;       // epilogue 2: pre-activation [M, N] (row stride = scm)
  float* aux_out;         // epilogue 1: receives the pre-activation
  int M, N, K;
  long long sam, sak, sbn, sbk, scm;
  long long batA, batB, batC;
  float alpha;
  int epi;                // 0 store (+ bias) | 1 bias, keep pre-activation, store GELU | 2 times GELU'(aux) | 3 C += value
  int ksplit;             // > 1: slice s of K goes to C + s * part (no epilogue)
  long long part;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// These problems are LATENCY-bound (a [132, 256] x [256, 256] product is 17 MFLOP): a first version that staged 16 k per
// iteration paid one ~1.5 us memory round trip per iteration (28-44 us per K = 256 product), a second one with 256 threads
// per 32 x 64 tile still spent 15-24 us on 1,024 FMAs + 48 address computations per thread at one wave per SIMD
// (profiles/r04v_generator.log).  Now: a block of 1,024 threads takes K in slices of 128; all loads of a slice (12 per
// thread) are in flight at once and the next slice is fetched into registers while the current one is multiplied out
// of LDS; the four 256-thread groups of the block each multiply a quarter of the slice (k-groups, 16 waves per CU hide
// the LDS latency) and fold their sums through LDS for a coalesced epilogue.  Products with few output tiles and a
// deep K are split over blocks (ksplit) and the consumer folds the partial sums.
constexpr int GTHREADS = 1024, GKG = GTHREADS / 256, GKQ = GBK / GKG;
constexpr int NA = GBM * GBK / GTHREADS, NB = GBN * GBK / GTHREADS;

__global__ __launch_bounds__(GTHREADS) void gen_gemm_kernel(GenGemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[GBK * APITCH];
  __shared__ __attribute__((aligned(16))) float Bs[GBK * BPITCH];   // >= GKG * GBM * GBN floats: reused for the fold
  static_assert(GBK * BPITCH >= GKG * GBM * GBN, "fold buffer");
  int b = blockIdx.x;
  const int tm = b % g.tiles_m;
  b /= g.tiles_m;
  const int tn = b % g.tiles_n;
  b /= g.tiles_n;
  const int ks = b % g.ksplit;
  const int bat = b / g.ksplit;
  const float* A = g.A + bat * g.batA;
  const float* B = g.B + bat * g.batB;
  float* C = g.C + bat * g.batC + (long long)ks * g.part;
  const int m0 = tm * GBM, n0 = tn * GBN;
  const int tid = threadIdx.x, kg = tid >> 8, ty = (tid >> 4) & 15, tx = tid & 15;
  const int kper = ((g.K + g.ksplit - 1) / g.ksplit + 15) / 16 * 16;
  const int k_beg = ks * kper, k_end = min(g.K, k_beg + kper);
  const bool a_kc = g.sak == 1, b_kc = g.sbk == 1;
  // element e of a slice <-> (row, k): the index that is contiguous in memory runs fastest across lanes
  long long aoff[NA], boff[NB];
  int ak[NA], bk[NB], al[NA], bl[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + i * GTHREADS;
    const int k = a_kc ? e % GBK : e / GBM, m = a_kc ? e / GBK : e % GBM;
    al[i] = k * APITCH + m;
    ak[i] = m0 + m < g.M ? k : (1 << 28);                                  // a row past M never passes the k test
    aoff[i] = (long long)(m0 + m) * g.sam + (long long)(k_beg + k) * g.sak;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = tid + i * GTHREADS;
    const int k = b_kc ? e % GBK : e / GBN, n = b_kc ? e / GBK : e % GBN;
    bl[i] = k * BPITCH + n;
    bk[i] = n0 + n < g.N ? k : (1 << 28);
    boff[i] = (long long)(n0 + n) * g.sbn + (long long)(k_beg + k) * g.sbk;
  }
  float ra[NA], rb[NB];
  auto fetch = [&](int k0) {
    const long long da = (long long)(k0 - k_beg) * g.sak, db = (long long)(k0 - k_beg) * g.sbk;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = k0 + ak[i] < k_end ? A[aoff[i] + da] : 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = k0 + bk[i] < k_end ? B[boff[i] + db] : 0.f;
  };
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (k_beg < k_end) fetch(k_beg);
  for (int k0 = k_beg; k0 < k_end; k0 += GBK) {
#pragma unroll
    for (int i = 0; i < NA; ++i) As[al[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) Bs[bl[i]] = rb[i];
    __syncthreads();
    if (k0 + GBK < k_end) fetch(k0 + GBK);
    const int klen = (min(GBK, k_end - k0) + 3) & ~3;     // rows past k_end hold zeros
    const int kb = kg * GKQ, ke = min(klen, kb + GKQ);
#pragma unroll 4
    for (int k = kb; k < ke; ++k) {
      const float2 a = *reinterpret_cast<const float2*>(&As[k * APITCH + ty * 2]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[k * BPITCH + tx * 4]);
      acc[0][0] = fmaf(a.x, bv.x, acc[0][0]);
      acc[0][1] = fmaf(a.x, bv.y, acc[0][1]);
      acc[0][2] = fmaf(a.x, bv.z, acc[0][2]);
      acc[0][3] = fmaf(a.x, bv.w, acc[0][3]);
      acc[1][0] = fmaf(a.y, bv.x, acc[1][0]);
      acc[1][1] = fmaf(a.y, bv.y, acc[1][1]);
      acc[1][2] = fmaf(a.y, bv.z, acc[1][2]);
      acc[1][3] = fmaf(a.y, bv.w, acc[1][3]);
    }
    __syncthreads();
  }
  // fold the k-groups: red[kg][m][n], then two neighbouring columns per thread
  float* red = Bs;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    *reinterpret_cast<float4*>(&red[(kg * GBM + ty * 2 + i) * GBN + tx * 4]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  __syncthreads();
  const int ml = (tid * 2) / GBN, nl = (tid * 2) % GBN;
  const int m = m0 + ml;
  if (m >= g.M) return;
  float2 sum = make_float2(0.f, 0.f);
#pragma unroll
  for (int q = 0; q < GKG; ++q) {
    const float2 t = *reinterpret_cast<const float2*>(&red[(q * GBM + ml) * GBN + nl]);
    sum.x += t.x;
    sum.y += t.y;
  }
  const float vv[2] = {sum.x, sum.y};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + nl + j;
    if (n >= g.N) continue;
    float v = vv[j] * g.alpha;
    const long long o = (long long)m * g.scm + n;
    if (g.ksplit > 1) { C[o] = v; continue; }
    if (g.epi == 0) {
      if (g.bias) v += g.bias[n];
      C[o] = v;
    } else if (g.epi == 1) {
      v += g.bias ? g.bias[n] : 0.f;
      g.aux_out[(long long)bat * g.batC + o] = v;
      C[o] = gelu_f(v);
    } else if (g.epi == 2) {
      C[o] = v * gelu_grad_f(g.aux[(long long)bat * g.batC + o]);
    } else {
      C[o] += v;
    }
  }
}

// one wave per row of d = 64 * per columns (per <= PER): v = sum_s parts[s] + bias + res; y = (v - mean) * rstd * gamma + beta.
// Fully unrolled over the columns of a lane and over the partial sums four at a time: every load of a row is in flight
// together (a runtime loop per partial sum waited one memory round trip per term: 34 us for 4 rows).
template <int PER>
__global__ __launch_bounds__(256) void gen_ln_fwd_kernel(const float* __restrict__ parts, int nparts, long long part,
                                                         const float* __restrict__ bias, const float* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, float* __restrict__ xhat,
                                                         float* __restrict__ rstd, int rows, int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int per = d >> 6;
  const long long ro = (long long)row * d;
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = lane + 64 * j;
    v[j] = 0.f;
    if (j < per) {
      if (bias) v[j] = bias[c];
      if (res) v[j] += res[ro + c];
    }
  }
  for (int p0 = 0; p0 < nparts; p0 += 4) {
    float t[PER][4];
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        t[j][i] = (j < per && p0 + i < nparts) ? parts[(long long)(p0 + i) * part + ro + lane + 64 * j] : 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[j] += t[j][i];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) s += v[j];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const float t = j < per ? v[j] - mean : 0.f;
    q += t * t;
  }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = lane + 64 * j;
    if (j < per) {
      const float xh = (v[j] - mean) * rs;
      xhat[ro + c] = xh;
      y[ro + c] = xh * gamma[c] + beta[c];
    }
  }
  if (lane == 0) rstd[row] = rs;
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy_total * gamma,  dy_total = dy + sum_s parts[s] (split-K partial
// sums of the GEMM that produced the rest of the gradient), written back to dysum for the column sums that follow
template <int PER>
__global__ __launch_bounds__(256) void gen_ln_bwd_kernel(const float* dy, const float* __restrict__ parts, int nparts,
                                                         long long part, float* dysum,
                                                         const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, float* __restrict__ dx, int rows,
                                                         int d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int per = d >> 6;
  const long long ro = (long long)row * d;
  float gv[PER], xh[PER], gm[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = lane + 64 * j;
    gv[j] = j < per ? dy[ro + c] : 0.f;
    xh[j] = j < per ? xhat[ro + c] : 0.f;
    gm[j] = j < per ? gamma[c] : 0.f;
  }
  for (int p0 = 0; p0 < nparts; p0 += 4) {
    float t[PER][4];
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        t[j][i] = (j < per && p0 + i < nparts) ? parts[(long long)(p0 + i) * part + ro + lane + 64 * j] : 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) gv[j] += t[j][i];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (dysum && j < per) dysum[ro + lane + 64 * j] = gv[j];
    gv[j] *= gm[j];
    s1 += gv[j];
    s2 += gv[j] * xh[j];
  }
  const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d, rs = rstd[row];
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (j < per) dx[ro + lane + 64 * j] = rs * (gv[j] - m1 - xh[j] * m2);
}

// column sums over `rows` rows: out0[c] = sum_r a[r][c] (* b[r][c] when b is given), out1[c] = sum_r a[r][c] (when asked).
// 32 columns x 8 row groups per block (a 132-row loop in one thread would be 132 dependent-latency loads deep)
__global__ __launch_bounds__(256) void gen_colsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out0, float* __restrict__ out1, int rows, int n,
                                                         long long ld) {
  __shared__ float red[2][8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < n) {
#pragma unroll 4
    for (int r = rg; r < rows; r += 8) {
      const float v = a[(long long)r * ld + c];
      s0 += b ? v * b[(long long)r * ld + c] : v;
      s1 += v;
    }
  }
  red[0][rg][cl] = s0;
  red[1][rg][cl] = s1;
  __syncthreads();
  if (rg == 0 && c < n) {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t0 += red[0][i][cl]; t1 += red[1][i][cl]; }
    out0[c] = t0;
    if (out1) out1[c] = t1;
  }
}

}  // namespace

extern "C" {

// C[b](m, n) = epi(alpha * sum_k A[b](m, k) B[b](n, k)); element strides (sam, sak), (sbn, sbk); C rows scm apart, columns
// contiguous.  epi 0: + bias[n] (nullable) | 1: + bias, pre-activation -> aux_out (layout of C), GELU -> C | 2: times
// GELU'(aux) | 3: C += value.  ksplit > 1: K slice s -> C + s * part_stride, no epilogue (fold with eovae_gen_ln_fwd).
int eovae_gen_gemm(const float* A, const float* B, float* C, const float* bias, const float* aux, float* aux_out, int batch,
                   int M, int N, int K, long long sam, long long sak, long long sbn, long long sbk, long long scm,
                   long long batA, long long batB, long long batC, float alpha, int epi, int ksplit, long long part_stride,
                   void* stream) {
  EOVAE_CHECK_ARG(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0, "gen_gemm: bad argument");
  EOVAE_CHECK_ARG(epi >= 0 && epi <= 3 && ksplit >= 1, "gen_gemm: epi %d ksplit %d", epi, ksplit);
  EOVAE_CHECK_ARG(ksplit == 1 || (!bias && epi == 0), "gen_gemm: split-K slices carry no bias / epilogue (ksplit %d, epi %d, "
                  "bias %s): fold them with eovae_gen_ln_fwd", ksplit, epi, bias ? "given" : "none");
  EOVAE_CHECK_ARG(epi != 1 || aux_out, "gen_gemm: epilogue 1 needs aux_out");
  EOVAE_CHECK_ARG(epi != 2 || aux, "gen_gemm: epilogue 2 needs aux");
  GenGemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux; g.aux_out = aux_out;
  g.M = M; g.N = N; g.K = K; g.sam = sam; g.sak = sak; g.sbn = sbn; g.sbk = sbk; g.scm = scm;
  g.batA = batA; g.batB = batB; g.batC = batC; g.alpha = alpha; g.epi = epi; g.ksplit = ksplit; g.part = part_stride;
  g.tiles_m = cdiv(M, GBM); g.tiles_n = cdiv(N, GBN);
  hipLaunchKernelGGL(gen_gemm_kernel, dim3((unsigned)(batch * ksplit * g.tiles_m * g.tiles_n)), dim3(GTHREADS), 0,
                     (hipStream_t)stream, g);
  EOVAE_CHECK_LAUNCH("eovae_gen_gemm");
  return EOVAE_OK;
}

int eovae_gen_ln_fwd(const float* parts, int nparts, long long part_stride, const float* bias, const float* res,
                     const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int rows, int d, float eps,
                     void* stream) {
  EOVAE_CHECK_ARG(parts && gamma && beta && y && xhat && rstd && rows > 0 && nparts >= 1, "gen_ln_fwd: bad argument");
  EOVAE_CHECK_ARG(d % 64 == 0 && d <= 1024, "gen_ln_fwd: d = %d (multiple of 64, <= 1024)", d);
  if (d <= 256)
    hipLaunchKernelGGL(gen_ln_fwd_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, parts, nparts, part_stride, bias,
                       res, gamma, beta, y, xhat, rstd, rows, d, eps);
  else
    hipLaunchKernelGGL(gen_ln_fwd_kernel<16>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, parts, nparts, part_stride, bias,
                       res, gamma, beta, y, xhat, rstd, rows, d, eps);
  EOVAE_CHECK_LAUNCH("eovae_gen_ln_fwd");
  return EOVAE_OK;
}

// dx (rows) and, when dgamma / dbeta are given, their column sums (second launch).  nparts > 0: the incoming gradient is
// dy + sum of `nparts` partial matrices `part_stride` apart; the total is written to dysum (required then, may alias dy).
int eovae_gen_ln_bwd(const float* dy, const float* parts, int nparts, long long part_stride, float* dysum, const float* xhat,
                     const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, int rows, int d,
                     void* stream) {
  EOVAE_CHECK_ARG(dy && xhat && rstd && gamma && dx && rows > 0 && nparts >= 0, "gen_ln_bwd: bad argument");
  EOVAE_CHECK_ARG(nparts == 0 || (parts && dysum), "gen_ln_bwd: partial sums need parts and dysum");
  EOVAE_CHECK_ARG(d % 64 == 0 && d <= 1024, "gen_ln_bwd: d = %d (multiple of 64, <= 1024)", d);
  if (d <= 256)
    hipLaunchKernelGGL(gen_ln_bwd_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, parts, nparts, part_stride,
                       nparts ? dysum : (float*)nullptr, xhat, rstd, gamma, dx, rows, d);
  else
    hipLaunchKernelGGL(gen_ln_bwd_kernel<16>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, parts, nparts, part_stride,
                       nparts ? dysum : (float*)nullptr, xhat, rstd, gamma, dx, rows, d);
  if (dgamma && dbeta)
    hipLaunchKernelGGL(gen_colsum_kernel, dim3(cdiv(d, 32)), dim3(256), 0, (hipStream_t)stream, nparts ? (const float*)dysum : dy,
                       xhat, dgamma, dbeta, rows, d, (long long)d);
  EOVAE_CHECK_LAUNCH("eovae_gen_ln_bwd");
  return EOVAE_OK;
}

int eovae_gen_colsum(const float* x, float* out, int rows, int n, long long ld, void* stream) {
  EOVAE_CHECK_ARG(x && out && rows > 0 && n > 0, "gen_colsum: bad argument");
  hipLaunchKernelGGL(gen_colsum_kernel, dim3(cdiv(n, 32)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, out,
                     (float*)nullptr, rows, n, ld);
  EOVAE_CHECK_LAUNCH("eovae_gen_colsum");
  return EOVAE_OK;
}

}  // extern "C"
