// Data gradient of "nearest x2 + 3x3 convolution" (adjoint of layers.py:47-50) in sub-pixel form on MX operand records of dY
// -- conv_f16c8_q16.hip's walk (one wave per SIMD, 16 x 16 MFMAs, every byte fed by LDS-DMA; tile = 128 input channels x 16
// low-res rows x 32 columns) with the GEMM K running over the FOUR parity planes of the high-res dY (round 6).
//
//   dX[ci][y][x] = sum over parities p = (py, px), output channels co and the parity's 2 x 2 taps of
//                  W'[p Cout + co][ci][ky][kx] * dY[co][2 (y - ky + 1) + py][2 (x - kx + 1) + px]
// W' = eovae_subpix_expand(w) (non-zero on ky in {py, py + 1}, kx in {px, px + 1}).  With the data-gradient pack of W'
// (flip 1: K = 4 Cout rows parity-major, M = Cin, taps rotated: tile index tf = 8 - (3 ky + kx)) this is a stride-1
// convolution over the parity plane  P_p[co][r][c] = dY[co][2 r + py][2 c + px]  with the taps kyf in {1 - py, 2 - py},
// kxf in {1 - px, 2 - px} at input pixel (y + kyf - 1, x + kxf - 1): 4/9 of the MFMAs of the layer as the reference executes it.
// The 8-wave kernel's mode for this (conv_f16c8.hip SUBPIX = 2) gathers and converts a full fp32 halo chunk per 32 K
// channels for 4 of 9 taps' MFMAs: ~360 TFLOP/s of actual MFMA work (3 launches, 3.25 ms per step).  Here dY is written ONCE
// as MX operand records (eovae_records_mx_from_f32, pre-scaled from its amax) and the parity planes reach the LDS by DMA: a
// lane's pixel offset is 2 x the low-res offset, the parity a constant added to the descriptor base.
//
// A chunk = 32 channels of one parity = 4 f16 phases (taps (a, b), a, b in {0, 1}: kyf = 1 - py + a, kxf = 1 - px + b) + 2
// correction phases (tap pairs ((a, 0), (a, 1)): neighbours along x, i.e. the second K half reads one pixel further), every
// phase 64 MFMAs.  Request schedule (requested in phase r = readable from phase r + 2):
//   weights of phase p + 3 in phase p (ring of four 16 KB slots; f16 phases use one 8 KB unit, correction phases two);
//   FP6 image of THIS chunk in phases 0, 1 (first read: phase 3 reads phase 4's fragments; last read of the previous chunk's: its phase 4);
//   f16 image of the NEXT chunk in phase 3, all ten requests (its last readers of this chunk are phase 3's fragments, read in
//   phase 2; first reader: phase 5, which reads the next phase 0's fragments) -- with six phases per chunk the single-buffered
//   image has exactly that one phase.
// Everything about asm statements that conv_f16c8_q16.hip's header lists applies here unchanged (conv_f16c8_q16_asm.h).
#include <stdlib.h>

#include "conv_f16c8_common.h"
#include "conv_f16c8_q16_asm.h"

namespace {

constexpr int GBM = 128, GXW = 34, GNPX = 18 * GXW, GNTASK = 10;
constexpr int GPPL = 640 * 16;                 // one octet's / piece's plane of the halo images: 640 pixels (612 + DMA padding)
constexpr int GIMG = 4 * GPPL;                 // 40,960
constexpr int GOFF_XF = 0, GOFF_X6 = GIMG, GOFF_W = 2 * GIMG;
constexpr int GWSLOT = 16384, GNWS = 4;
constexpr int GLDS = GOFF_W + GNWS * GWSLOT;   // 147,456
constexpr int GNPH = 6;                        // phases per chunk: 0..3 f16, 4..5 correction

__device__ __host__ constexpr int g_is_c(int p) { return ((p % GNPH) + GNPH) % GNPH >= 4; }
__device__ __host__ constexpr int g_nw(int p) { return g_is_c(p + 3) ? 4 : 2; }
__device__ __host__ constexpr int g_nh(int p) { return p == 0 || p == 1 ? 5 : (p == 3 ? 10 : 0); }

// a: x / x6 = F / S plane of the dY records [N][Cin_k = Cout_layer][Hin = 2 H][Win = 2 W]; Cin = 4 * creal (GEMM K), creal =
// the layer's output channels; Cout = the layer's input channels (GEMM M); Hout, Wout = H, W (low-res); nchunk = Cin / 32;
// ncob = Cout / 128; tiles_y = Hout / 16, tiles_x = Wout / 32
__global__ __launch_bounds__(256, 1) void conv3x3_f16c8_q16upg_kernel(ConvCArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[GLDS];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wrow = wave * 4;
  const size_t hi_plane = (size_t)a.Hin * a.Win;      // pixels of a high-res channel plane
  const size_t out_plane = (size_t)a.Hout * a.Wout;
  const int cpp = a.creal / 32;                       // chunks per parity

  struct Tile { int cob, n, oy0, ox0; };
  auto decode = [&](int b) {
    Tile t;
    if (a.xcd_remap) {  // the ncob blocks sharing one halo tile take consecutive slots of ONE XCD
      const int xcd = b & 7, q = b >> 3;
      t.cob = q % a.ncob;
      b = (q / a.ncob) * 8 + xcd;
    } else {
      t.cob = b % a.ncob;
      b /= a.ncob;
    }
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    t.n = b / a.tiles_y;
    t.oy0 = ty * 16;
    t.ox0 = tx * 32;
    return t;
  };

  auto lds0_ = [&]() __attribute__((always_inline)) { return (unsigned)(size_t)smem; };
  float xinv = 1.f;   // range pre-scale of the records (amax of dY to 2^13..2^14), undone with the result
  if (a.xamax) {
    const unsigned bits = *a.xamax;
    if (bits) xinv = pow2f(expo_of_bits(bits) - 13);
  }

  // ---------------------------------------------------------------- halo requests (wave = channel octet / FP6 piece)
  // task k of a wave: low-res pixels 64 k + lane of the 18 x 34 halo; the lane's byte offset of high-res pixel (2 vy, 2 vx) in
  // a plane, or out of range (zeros); the chunk's parity (py, px) is added to the descriptor's base
  const unsigned OOR = 0x80000000u;
  unsigned xoff[GNTASK];
  auto setup_x = [&](const Tile& t) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int k = 0; k < GNTASK; ++k) {
      const int pix = ln + 64 * k;
      const int rr = (pix * 241) >> 13, cc = pix - GXW * rr;   // pix / 34, exact for pix < 640
      const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
      const bool ok = (pix < GNPX) & ((unsigned)vy < (unsigned)a.Hout) & ((unsigned)vx < (unsigned)a.Wout);
      xoff[k] = ok ? (unsigned)(2 * vy * a.Win + 2 * vx) * 16u : OOR;
    }
  };
  const unsigned plane_bytes = (unsigned)(hi_plane * 16);
  // chunk c: parity c / cpp, channels 32 (c % cpp) .. of the layer's dY
  auto srd_xf = [&](const Tile& t, int chunk) __attribute__((always_inline)) {   // F plane [n][creal / 8][pixel][16 B], octet 4 cl + wave
    const int par = chunk / cpp, cl = chunk - par * cpp;
    const unsigned poff = (unsigned)((par >> 1) * a.Win + (par & 1)) * 16u;
    const char* p = reinterpret_cast<const char*>(a.x) + (((size_t)t.n * (a.creal / 8) + 4 * cl + wave) * hi_plane) * 16 + poff;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, plane_bytes - poff, 0x00020000);
  };
  auto srd_x6 = [&](const Tile& t, int chunk) __attribute__((always_inline)) {   // S plane [n][creal / 32][piece][pixel][16 B], piece = wave
    const int par = chunk / cpp, cl = chunk - par * cpp;
    const unsigned poff = (unsigned)((par >> 1) * a.Win + (par & 1)) * 16u;
    const char* p = reinterpret_cast<const char*>(a.x6) + ((((size_t)t.n * (a.creal / 32) + cl) * 4 + wave) * hi_plane) * 16 + poff;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, plane_bytes - poff, 0x00020000);
  };
  unsigned char* const xf_dst = smem + GOFF_XF + wave * GPPL;
  unsigned char* const x6_dst = smem + GOFF_X6 + wave * GPPL;
  auto dma_x = [&](const __amdgpu_buffer_rsrc_t& srd, unsigned char* dst, int k) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)(dst + k * 1024), 16, (int)xoff[k], 0, 0, 0);
  };

  // ---------------------------------------------------------------- weight requests: 8 KB units, two DMAs each
  const size_t wbytes = (size_t)9 * a.nchunk * a.ncob * C8_WTILE;
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wp), 0, (unsigned)wbytes, 0x00020000);
  const int wvoff = tid * 16;
  auto dma_unit = [&](unsigned soff, unsigned slotbyte) __attribute__((always_inline)) {
    unsigned char* dst = smem + GOFF_W + slotbyte + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr)dst, 16, wvoff, (int)soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr)(dst + 4096), 16, wvoff, (int)(soff + 4096u), 0, 0);
  };
  // units of phase q (0..5) of chunk `chunk` of co-block `cob` into ring slot `slot`
  auto request_w = [&](auto qc, int cob, int chunk, unsigned slot) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    const unsigned sb = slot * GWSLOT;
    const int par = chunk / cpp;
    const int t00 = (1 - (par >> 1)) * 3 + (1 - (par & 1));      // tile index of the parity's tap (a, b) = (0, 0)
    if constexpr (q < 4) {
      constexpr int a_ = q >> 1, b_ = q & 1;
      dma_unit((unsigned)(((size_t)((t00 + 3 * a_ + b_) * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE), sb);
    } else {
      constexpr int a_ = q - 4;
      dma_unit((unsigned)(((size_t)((t00 + 3 * a_) * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE + C8_WOFF6), sb);
      dma_unit((unsigned)(((size_t)((t00 + 3 * a_ + 1) * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE + C8_WOFF6), sb + 8192);
    }
  };

  // ---------------------------------------------------------------- fragments (as conv_f16c8_q16.hip)
  const unsigned lds0 = lds0_();
  const unsigned a_f = lds0 + (unsigned)(GOFF_W + g * 2048 + li * 16);
  const unsigned a_c = lds0 + (unsigned)(GOFF_W + (g >> 1) * 8192 + (g & 1) * 4096 + li * 16);
  const unsigned b_f0 = lds0 + (unsigned)(GOFF_XF + g * GPPL + (wrow * GXW + li) * 16);
  const unsigned b_c0 = lds0 + (unsigned)(GOFF_X6 + 2 * (g & 1) * GPPL + (wrow * GXW + li) * 16) + (unsigned)(g >> 1) * 16u;
  unsigned b_f = b_f0, b_c = b_c0;   // + the chunk parity's tap origin (1 - py, 1 - px)

  f32x4 acc[8][8];   // [ci tile of 16][pixel tile: row 0..3 x half 0..1]
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  };
  i32x4 Ar[2][2], Br[2][8];
  i32x3 Ah[2][2], Bh[2][8];

  auto read_a = [&](auto isc, auto c2c, unsigned sb) __attribute__((always_inline)) {
    constexpr bool ISC = decltype(isc)::value;
    constexpr int c2 = decltype(c2c)::value;
    const unsigned base = (ISC ? a_c : a_f) + sb;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (ISC) {
        if (t == 0) { rd128<(2 * c2) * 256>(Ar[c2 & 1][0], base); rd96<(2 * c2) * 256 + 2048>(Ah[c2 & 1][0], base); }
        else { rd128<(2 * c2 + 1) * 256>(Ar[c2 & 1][1], base); rd96<(2 * c2 + 1) * 256 + 2048>(Ah[c2 & 1][1], base); }
      } else {
        if (t == 0) rd128<(2 * c2) * 256>(Ar[c2 & 1][0], base);
        else rd128<(2 * c2 + 1) * 256>(Ar[c2 & 1][1], base);
      }
    }
  };
  // B fragment of pixel tile j (row j >> 1, half j & 1) of phase q into Br[q & 1][j]; bf / bc = the bases of the chunk the
  // phase belongs to
  auto read_b = [&](auto qc, auto jc, unsigned bf, unsigned bc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, j = decltype(jc)::value, jr = j >> 1, hx = j & 1;
    if constexpr (q < 4) {
      constexpr int a_ = q >> 1, b_ = q & 1;
      rd128<((a_ + jr) * GXW + b_ + 16 * hx) * 16>(Br[q & 1][j], bf);
    } else {
      constexpr int a_ = q - 4;
      rd128<((a_ + jr) * GXW + 16 * hx) * 16>(Br[q & 1][j], bc);
      rd96<((a_ + jr) * GXW + 16 * hx) * 16 + GPPL>(Bh[q & 1][j], bc);
    }
  };
  auto mma = [&](auto qc, auto ic, auto jc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = decltype(ic)::value, j = decltype(jc)::value;
    constexpr int pr = (i >> 1) & 1, t = i & 1;
    if constexpr (q < 4) {
      mfma_f(acc[i][j], Br[q & 1][j], Ar[pr][t]);
    } else {
      i32x6 a6, b6;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a6[e] = Ar[pr][t][e]; b6[e] = Br[q & 1][j][e]; }
      a6[4] = Ah[pr][t][0]; a6[5] = Ah[pr][t][1];
      b6[4] = Bh[q & 1][j][0]; b6[5] = Bh[q & 1][j][1];
      mfma_c<(q == GNPH - 1 && i >= 6 && j == 7)>(acc[i][j], b6, a6, Bh[q & 1][j][2], Ah[pr][t][2]);
    }
  };

  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto landed_a = [&](int pr) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      asm volatile("" : "+v"(Ar[pr][t]));
      asm volatile("" : "+v"(Ah[pr][t]));
    }
  };
  auto landed_b = [&](int par) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("" : "+v"(Br[par][j]));
      asm volatile("" : "+v"(Bh[par][j]));
    }
  };
  // LDS byte shift of a chunk's tap origin (1 - py, 1 - px)
  auto tap_origin = [&](int chunk) __attribute__((always_inline)) {
    const int par = chunk / cpp;
    return (unsigned)(((1 - (par >> 1)) * GXW + (1 - (par & 1))) * 16);
  };

  // ---------------------------------------------------------------- epilogue: dx = acc * xinv, 16-byte stores (conv_f16c8_q16.hip)
  auto epilogue_tile = [&](const Tile& t) __attribute__((always_inline)) {
    const int co_blk = t.cob * GBM;
    const unsigned span = (unsigned)(GBM * out_plane * 4);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
        a.y + ((size_t)t.n * a.Cout + co_blk) * out_plane, 0, span, 0x00020000);
    // (as conv_f16c8_q16.hip's epilogue: lane 4 c + p takes lane 16 p + c's registers, a lane quad then writes one 64-byte
    // segment; the addresses derive from fresh_lane() so that they are not hoisted out of the tile loop and spilled)
    const int le = fresh_lane();
    const int ec = le >> 2, ep = le & 3;   // channel, pixel group
    const int psrc = ((ep << 4) | ec) << 2;
    const int lot = (int)(((size_t)ec * out_plane + (size_t)(t.oy0 + wrow) * a.Wout + t.ox0 + 4 * ep) * 4);
    const int tile16 = (int)(16 * out_plane * 4), rowb = a.Wout * 4;
    static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = acc[i][j][k];   // (copied out: __builtin_bit_cast of a vector element expression, q16.hip)
          o[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, a0))) * xinv;
        }
        store128_guarded(o, ysrd, lot, i * tile16 + (j >> 1) * rowb + (j & 1) * 64);
      }
    });
  };

  // ---------------------------------------------------------------- prologue
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  Tile cur = decode(tile);
  unsigned P = 0;   // running phase number (mod 4 = ring slot)
  {
    setup_x(cur);
    const __amdgpu_buffer_rsrc_t s0 = srd_xf(cur, 0);
#pragma unroll
    for (int k = 0; k < GNTASK; ++k) dma_x(s0, xf_dst, k);
    request_w(std::integral_constant<int, 0>{}, cur.cob, 0, 0u);
    request_w(std::integral_constant<int, 1>{}, cur.cob, 0, 1u);
    request_w(std::integral_constant<int, 2>{}, cur.cob, 0, 2u);
  }
  zero_acc();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  {
    using Q0 = std::integral_constant<int, 0>;
    const unsigned org = tap_origin(0);
    b_f = b_f0 + org;
    b_c = b_c0 + org;
    read_a(std::false_type{}, Q0{}, 0u);
    static_for<0, 8>([&](auto jc) __attribute__((always_inline)) { read_b(Q0{}, jc, b_f, b_c); });
  }

  // ---------------------------------------------------------------- main loop
  for (; tile < a.ntiles; tile += gridDim.x) {
    const bool has_next_tile = tile + (int)gridDim.x < a.ntiles;
    Tile nxt = cur;
    if (has_next_tile) nxt = decode(tile + gridDim.x);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool last_chunk = chunk + 1 == a.nchunk;
      // the chunk whose f16 image and first weights are requested behind this one: the next chunk, the next tile's first, or
      // (end of work) this chunk again -- re-requesting it writes the bytes that are already there
      const bool wrap = last_chunk && has_next_tile;
      const Tile& xt = wrap ? nxt : cur;
      const int xchunk = last_chunk ? (has_next_tile ? 0 : chunk) : chunk + 1;
      const __amdgpu_buffer_rsrc_t s6 = srd_x6(cur, chunk);
      __amdgpu_buffer_rsrc_t sf = s6;   // the next chunk's F source (set in phase 3)
      const unsigned org_n = tap_origin(xchunk);
      const unsigned b_fn = b_f0 + org_n, b_cn = b_c0 + org_n;   // fragment bases of the next chunk
      static_for<0, GNPH>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int pn = (p + 1) % GNPH;
        constexpr bool ISC = p >= 4, NISC = pn >= 4;
        using PN = std::integral_constant<int, pn>;
        barrier();   // phase p's weights, the images it reads and its first fragments are there
        landed_a(0);
        landed_b(p & 1);
        const unsigned sb = ((P + p) & 3u) * GWSLOT, sbn = ((P + p + 1) & 3u) * GWSLOT;
        // the next phase's fragments come from this chunk's images, behind the last phase from the next chunk's
        const unsigned bfn = p == GNPH - 1 ? b_fn : b_f, bcn = p == GNPH - 1 ? b_cn : b_c;
        if constexpr (p == 3) {
          if (wrap) setup_x(nxt);   // uniform; this chunk's FP6 image was requested in phases 0, 1
          sf = srd_xf(xt, xchunk);
        }
        static_for<0, 4>([&](auto c2c) __attribute__((always_inline)) {
          constexpr int c2 = decltype(c2c)::value;
          static_for<0, 16>([&](auto mc) __attribute__((always_inline)) {
            constexpr int mm = decltype(mc)::value;
            constexpr int j = mm >> 1, i = 2 * c2 + (mm & 1);
            mma(pc, std::integral_constant<int, i>{}, std::integral_constant<int, j>{});
            if constexpr (mm == 1) {
              if constexpr (c2 == 0) {   // weights of phase p + 3
                if constexpr (p + 3 < GNPH) request_w(std::integral_constant<int, p + 3>{}, cur.cob, chunk, (P + p + 3) & 3u);
                else request_w(std::integral_constant<int, p + 3 - GNPH>{}, xt.cob, xchunk, (P + p + 3) & 3u);
              }
              if constexpr (c2 == 1) {   // halo requests of this phase
                if constexpr (p == 0) { dma_x(s6, x6_dst, 0); dma_x(s6, x6_dst, 1); dma_x(s6, x6_dst, 2); dma_x(s6, x6_dst, 3); dma_x(s6, x6_dst, 4); }
                if constexpr (p == 1) { dma_x(s6, x6_dst, 5); dma_x(s6, x6_dst, 6); dma_x(s6, x6_dst, 7); dma_x(s6, x6_dst, 8); dma_x(s6, x6_dst, 9); }
                if constexpr (p == 3) { dma_x(sf, xf_dst, 0); dma_x(sf, xf_dst, 1); dma_x(sf, xf_dst, 2); }
              }
              if constexpr (c2 == 2 && p == 3) { dma_x(sf, xf_dst, 3); dma_x(sf, xf_dst, 4); dma_x(sf, xf_dst, 5); dma_x(sf, xf_dst, 6); }
              if constexpr (c2 == 3 && p == 3) { dma_x(sf, xf_dst, 7); dma_x(sf, xf_dst, 8); dma_x(sf, xf_dst, 9); }
            }
            if constexpr (mm == 3) {     // A fragments of the next co-tile pair (of the next phase behind the last pair)
              if constexpr (c2 < 3) read_a(std::integral_constant<bool, ISC>{}, std::integral_constant<int, c2 + 1>{}, sb);
              else read_a(std::integral_constant<bool, NISC>{}, std::integral_constant<int, 0>{}, sbn);
            }
            if constexpr (mm == 7) read_b(PN{}, std::integral_constant<int, 2 * c2>{}, bfn, bcn);
            if constexpr (mm == 11) read_b(PN{}, std::integral_constant<int, 2 * c2 + 1>{}, bfn, bcn);
            if constexpr (mm == 15) {    // the next pair's A fragments have landed; the B reads behind them may be outstanding
              if constexpr (c2 < 3) {
                wait_lgkm<NISC ? 4 : 2>();
                landed_a((c2 + 1) & 1);
              }
            }
          });
        });
        constexpr int NREQ = g_nw(p) + g_nh(p);
        if constexpr (p == GNPH - 1) {
          if (last_chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wait_lgkm<0>();    // the next phase-0 fragments, before the compiler-scheduled epilogue may touch their registers
            landed_a(0);
            landed_b(0);
          } else wait_vm<NREQ>();
        } else if constexpr (p == 0) {
          if (chunk != 0 || (EOVAE_Q16_SAFE & 1)) wait_vm<NREQ>();
        } else {
          wait_vm<NREQ>();
        }
      });
      P += GNPH;
      b_f = b_fn;
      b_c = b_cn;
    }
    epilogue_tile(cur);
    zero_acc();
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests behind the last chunk
}

}  // namespace

static int upg_n_cu() {
  static int cached_cu = 0;
  if (cached_cu == 0) {
    int dev = 0, n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    n_cu -= n_cu % 8;
    cached_cu = n_cu < 8 ? 8 : n_cu;
  }
  return cached_cu;
}

extern "C" {

// 1 when eovae_conv3x3_f16c8_up_dgrad_mx takes the layer: dY [N][Cout][2 H][2 W] -> dX [N][Cin][H][W]
int eovae_conv3x3_f16c8_up_dgrad_mx_eligible(int N, int Cin, int H, int W, int Cout) {
  if (!(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0)) return 0;
  if (eovae_opt(EOVAE_OPT_C8_UP_Q16) == 0 || eovae_opt(EOVAE_OPT_C8_MX) == 0) return 0;
  if (Cout % 32 || Cin % 128 || H % 16 || W % 32) return 0;
  const size_t hi_plane = (size_t)4 * H * W, lo_plane = (size_t)H * W;
  if (hi_plane * 16 >= (1ull << 31) || (size_t)GBM * lo_plane * 4 >= (1ull << 31)) return 0;   // 32-bit byte offsets
  if ((size_t)9 * (4 * Cout / 32) * (Cin / 128) * C8_WTILE >= (1ull << 32)) return 0;
  // one tile per CU at least: below that the 8-wave kernel's 8-row tiles fill the chip better and the records pass over dY
  // is not paid back (profiles/r06f_up_layer_ab.log: 0.96 x at 512 -> 512 @ 32^2, B = 16 = 128 tiles; option value 2 = whenever the shape allows)
  const long long tiles = (long long)N * (H / 16) * (W / 32) * (Cin / 128);
  return tiles >= (eovae_opt(EOVAE_OPT_C8_UP_Q16) >= 2 ? 1 : upg_n_cu()) ? 1 : 0;
}

// dX [N][Cin][H][W] = adjoint of the x2-upsample layer (layers.py:47-50) w.r.t. its input.  dyrec = MX operand records of the
// high-res dY [N][Cout][2 H][2 W] (eovae_records_mx_from_f32 with the same dyamax); wp = the f16c8 DATA-GRADIENT pack
// (flip 1) of eovae_subpix_expand(w) (K = 4 Cout rows).  The products are those of conv_f16c8_q16.hip's forward: f16 x f16 +
// the MX-FP6 correction product.
int eovae_conv3x3_f16c8_up_dgrad_mx(const float* dyrec, const void* wp, float* dx, const unsigned* dyamax, int N, int Cin, int H,
                                    int W, int Cout, void* stream) {
  EOVAE_CHECK_ARG(dyrec && wp && dx, "conv3x3_f16c8_up_dgrad_mx: null pointer");
  EOVAE_CHECK_ARG(eovae_conv3x3_f16c8_up_dgrad_mx_eligible(N, Cin, H, W, Cout),
                  "conv3x3_f16c8_up_dgrad_mx: shape not supported (N %d Cin %d %dx%d Cout %d; needs Cout %% 32, Cin %% 128, H %% 16, W %% 32)",
                  N, Cin, H, W, Cout);
  ConvCArgs a;
  a.x = dyrec; a.x6 = reinterpret_cast<const unsigned char*>(dyrec) + (size_t)N * Cout * 4 * H * W * 2;   // S plane behind the F plane
  a.wp = (const unsigned char*)wp; a.bias = nullptr; a.res = nullptr; a.y = dx; a.xamax = dyamax;
  a.N = N; a.Cin = 4 * Cout; a.creal = Cout; a.Hin = 2 * H; a.Win = 2 * W; a.Cout = Cin; a.Hout = H; a.Wout = W;
  a.nchunk = a.Cin / 32; a.ncob = Cin / 128; a.cpb = 1;
  a.mode = MODE_NORMAL; a.subpix = 2; a.packed = 2; a.mx = 1; a.stagger = 0; a.skew = 0; a.out_amax = nullptr;
  a.gn_sums = nullptr; a.gn_groups = 0; a.gn_shift = 0;
  a.tiles_x = W / 32; a.tiles_y = H / 16;
  const long long sp_tiles = (long long)N * a.tiles_y * a.tiles_x;
  a.xcd_remap = (a.ncob > 1 && sp_tiles % 8 == 0) ? 1 : 0;
  EOVAE_CHECK_ARG(sp_tiles * a.ncob < (1LL << 30), "conv3x3_f16c8_up_dgrad_mx: too many tiles");
  a.ntiles = (int)(sp_tiles * a.ncob);
  const int cached_cu = upg_n_cu();
  const dim3 grid((unsigned)(a.ntiles < cached_cu ? a.ntiles : cached_cu)), blk(256);
  hipLaunchKernelGGL(conv3x3_f16c8_q16upg_kernel, grid, blk, 0, (hipStream_t)stream, a);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_up_dgrad_mx");
  return EOVAE_OK;
}

}  // extern "C"
