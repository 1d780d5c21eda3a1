// Probe (round 6): what does a CU's store path sustain for the epilogue patterns of the one-wave kernels?
//   A  conv_f16c8_q16.hip today: one 16-byte store per lane; lanes (li, g): channel plane li (stride = plane bytes), 4 lane groups
//      contiguous -> 16 segments of 64 B per instruction.
//   B  interleaved halves: two stores per lane, 16 B each, 32 B apart per lane group, the partner store fills the gaps ->
//      per (channel, row) a full 128-byte line from two instructions (what conv_f16c8_q16up.hip's parity interleave does).
//   C  one 16-byte store per lane, fully contiguous (64 lanes x 16 B = 1 KB): the best case.
// One workgroup of 256 threads per CU (as the kernels), 64 KB per wave and tile, tiles walked like the kernels do.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_store_pattern.hip -o tools/probe_store_pattern && tools/probe_store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void k(float* out, size_t plane, int rowb, int tiles) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  f32x4 v = {(float)tid, 1.f, 2.f, 3.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    // a tile = 128 channels x 16 rows x 32 px; wave = 4 rows; block b's tile sequence strided over the image like the kernels
    const size_t tile_base = ((size_t)(blockIdx.x + t * gridDim.x) % 2048) * 32;   // column offset inside a wide image row
    float* base = out + tile_base;
#pragma unroll
    for (int i = 0; i < 8; ++i) {          // co tile
#pragma unroll
      for (int j = 0; j < 8; ++j) {        // pixel tile: row j >> 1, half j & 1
        float* p;
        if (PAT == 0) p = base + (size_t)(16 * i + li) * plane + (size_t)(wave * 4 + (j >> 1)) * rowb + (j & 1) * 16 + 4 * g;
        else if (PAT == 1) p = base + (size_t)(16 * i + li) * plane + (size_t)(wave * 4 + (j >> 1)) * rowb + 8 * g + (j & 1) * 4;
        else if (PAT == 3 || PAT == 4) {
          // D: A's 16 x 64-byte segments, but a QUAD of consecutive lanes holds one segment (lane L: channel L >> 2, pixels 4 (L & 3))
          p = base + (size_t)(16 * i + (lane >> 2)) * plane + (size_t)(wave * 4 + (j >> 1)) * rowb + (j & 1) * 16 + 4 * (lane & 3);
          if (PAT == 4) {   // ... reached from the MFMA layout with four ds_bpermute_b32
            const int src = (((lane & 3) << 4) | (lane >> 2)) << 2;
            f32x4 w;
            w[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v[0])));
            w[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v[1])));
            w[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v[2])));
            w[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v[3])));
            asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(w) : "memory");
            continue;
          }
        }
        else p = base + (size_t)(16 * i) * plane + (size_t)(wave * 4 + (j >> 1)) * rowb + (j & 1) * 256 + lane * 4;
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
  if (tid == 0) reinterpret_cast<unsigned long long*>(out)[1 + blockIdx.x] = t1 - t0;
}

template <int PAT>
void run(const char* name, float* d, size_t plane, int rowb, int blocks) {
  const int tiles = 64;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, d, plane, rowb, tiles);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, d, plane, rowb, tiles);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long cyc[1024];
  hipMemcpy(cyc, reinterpret_cast<unsigned long long*>(d) + 1, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < blocks; ++i) mean += (double)cyc[i];
  mean /= blocks;
  const double bytes_per_block = (double)tiles * 4 * 64 * 1024;
  printf("%-44s %3d blocks: %.3f ms  %.1f B/clk/CU at 2.4 GHz  %.0f GB/s chip\n", name, blocks, ms,
         bytes_per_block / (ms * 2.4e6), bytes_per_block * blocks / ms / 1e6);
  (void)mean;
}

int main() {
  const size_t plane = 256 * 256;            // floats per channel plane
  float* d;
  hipMalloc(&d, plane * 4 * 128 * 2 + (1 << 20));
  for (int blocks : {256, 8}) {
    run<0>("A: 16 x 64-byte segments per store (q16)", d, plane, 256, blocks);
    run<1>("B: interleaved halves, 128-byte lines", d, plane, 256, blocks);
    run<2>("C: 1 KB contiguous per store", d, plane, 256, blocks);
    run<3>("D: A's segments, one per lane QUAD", d, plane, 256, blocks);
    run<4>("E: D behind 4 ds_bpermute_b32", d, plane, 256, blocks);
  }
  return 0;
}
