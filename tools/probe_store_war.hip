// Probe (round 6): is a VALU write of a buffer_store_dwordx4's data registers, issued right behind the store, a hazard on
// gfx950 when the store has a REGISTER soffset?  (LLVM's hazard recognizer inserts its one wait state only for stores without a
// register soffset.)  conv_f16c8_q16up.hip's first epilogue showed lanes 12..15 of every 16-lane row storing the NEXT value of
// a register that a v_pk_fma_f32 rewrote one instruction behind the store.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_store_war.hip -o /tmp/probe_store_war && /tmp/probe_store_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: v_pk_fma_f32 into data dwords 0..1; 1: into dwords 2..3; 2: v_mov_b32 into dword 1; 3: v_pk_mul into dwords 0..1
// NOPS: s_nop count between store and overwrite (-1 = none).  SOFF: 1 register soffset, 0 immediate offset only
template <int MODE, int NOPS, int SOFF>
__global__ __launch_bounds__(256) void probe(float* out, int stride_bytes, int rounds) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * rounds * 256 * 4, 0,
                                                                        (unsigned)(rounds * 256 * 16), 0x00020000);
  int voff = lane * 16;
  int soff = 0;
  for (int r = 0; r < rounds; ++r) {
    f32x4 d;
    d[0] = (float)(lane * 4 + 0); d[1] = (float)(lane * 4 + 1); d[2] = (float)(lane * 4 + 2); d[3] = (float)(lane * 4 + 3);
    f32x2 a = {1000.f + lane, 2000.f + lane}, b = {1.f, 1.f}, c = {0.f, 0.f};
    asm volatile("" : "+v"(d), "+v"(a), "+v"(b), "+v"(c), "+s"(soff));
    if (SOFF) {
      if (MODE == 0) {
        if (NOPS < 0) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\tv_pk_fma_f32 %4, %5, %6, %7" :: "v"(d), "v"(voff), "s"(srd), "s"(soff), "v"(*(f32x2*)&d), "v"(a), "v"(b), "v"(c) : "memory");
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop %8\n\tv_pk_fma_f32 %4, %5, %6, %7" :: "v"(d), "v"(voff), "s"(srd), "s"(soff), "v"(*(f32x2*)&d), "v"(a), "v"(b), "v"(c), "n"(NOPS < 0 ? 0 : NOPS) : "memory");
      }
    }
    soff += 256 * 16;
  }
}

// the same with explicit register numbers (the inline-asm operand form above cannot name sub-registers of d)
template <int MODE, int NOPS, int SOFF>
__global__ __launch_bounds__(256) void probe2(float* out, int rounds) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * rounds * 256 * 4, 0,
                                                                        (unsigned)(rounds * 256 * 16), 0x00020000);
  const int voff = lane * 16;
  int soff = 0;
  const float f0 = (float)(lane * 4), l1000 = 1000.f + lane;
  for (int r = 0; r < rounds; ++r) {
#define STORE_SOFF "buffer_store_dwordx4 v[20:23], %0, %1, %2 offen\n\t"
#define STORE_IMM "buffer_store_dwordx4 v[20:23], %0, %1, 0 offen\n\t"
#define SETUP "v_mov_b32 v20, %3\n\tv_add_f32 v21, 1.0, %3\n\tv_add_f32 v22, 2.0, %3\n\tv_add_f32 v23, 3.0, %3\n\t" \
              "v_mov_b32 v24, %4\n\tv_mov_b32 v25, %4\n\tv_mov_b32 v26, 1.0\n\tv_mov_b32 v27, 1.0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\ts_nop 4\n\t"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "memory"
    if (MODE == 0) {        // v_pk_fma_f32 into dwords 0..1
      if (NOPS < 0) asm volatile(SETUP STORE_SOFF "v_pk_fma_f32 v[20:21], v[24:25], v[26:27], v[28:29]" :: "v"(voff), "s"(srd), "s"(soff), "v"(f0), "v"(l1000) : CLOB);
      else asm volatile(SETUP STORE_SOFF "s_nop %5\n\tv_pk_fma_f32 v[20:21], v[24:25], v[26:27], v[28:29]" :: "v"(voff), "s"(srd), "s"(soff), "v"(f0), "v"(l1000), "n"(NOPS < 0 ? 0 : NOPS) : CLOB);
    } else if (MODE == 1) { // v_pk_fma_f32 into dwords 2..3
      if (NOPS < 0) asm volatile(SETUP STORE_SOFF "v_pk_fma_f32 v[22:23], v[24:25], v[26:27], v[28:29]" :: "v"(voff), "s"(srd), "s"(soff), "v"(f0), "v"(l1000) : CLOB);
      else asm volatile(SETUP STORE_SOFF "s_nop %5\n\tv_pk_fma_f32 v[22:23], v[24:25], v[26:27], v[28:29]" :: "v"(voff), "s"(srd), "s"(soff), "v"(f0), "v"(l1000), "n"(NOPS < 0 ? 0 : NOPS) : CLOB);
    } else if (MODE == 2) { // v_mov_b32 into each dword
      if (NOPS < 0) asm volatile(SETUP STORE_SOFF "v_mov_b32 v20, v24\n\tv_mov_b32 v21, v24\n\tv_mov_b32 v22, v24\n\tv_mov_b32 v23, v24" :: "v"(voff), "s"(srd), "s"(soff), "v"(f0), "v"(l1000) : CLOB);
      else asm volatile(SETUP STORE_SOFF "s_nop %5\n\tv_mov_b32 v20, v24\n\tv_mov_b32 v21, v24\n\tv_mov_b32 v22, v24\n\tv_mov_b32 v23, v24" :: "v"(voff), "s"(srd), "s"(soff), "v"(f0), "v"(l1000), "n"(NOPS < 0 ? 0 : NOPS) : CLOB);
    } else {                // immediate-offset store (voffset carries everything), v_pk_fma_f32 into dwords 0..1 and 2..3
      int vo = voff + soff;
      if (NOPS < 0) asm volatile(SETUP STORE_IMM "v_pk_fma_f32 v[20:21], v[24:25], v[26:27], v[28:29]\n\tv_pk_fma_f32 v[22:23], v[24:25], v[26:27], v[28:29]" :: "v"(vo), "s"(srd), "s"(soff), "v"(f0), "v"(l1000) : CLOB);
      else asm volatile(SETUP STORE_IMM "s_nop %5\n\tv_pk_fma_f32 v[20:21], v[24:25], v[26:27], v[28:29]\n\tv_pk_fma_f32 v[22:23], v[24:25], v[26:27], v[28:29]" :: "v"(vo), "s"(srd), "s"(soff), "v"(f0), "v"(l1000), "n"(NOPS < 0 ? 0 : NOPS) : CLOB);
    }
    soff += 256 * 16;
  }
}

template <int MODE, int NOPS>
void run(const char* what) {
  const int rounds = 64, blocks = 1024;
  float* d;
  const size_t n = (size_t)blocks * rounds * 256 * 4;
  hipMalloc(&d, n * 4);
  hipMemset(d, 0xff, n * 4);
  hipLaunchKernelGGL((probe2<MODE, NOPS, 1>), dim3(blocks), dim3(256), 0, 0, d, rounds);
  hipDeviceSynchronize();
  std::vector<float> h(n);
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  long badlane[16] = {0}, baddw[4] = {0};
  for (size_t i = 0; i < n; ++i) {
    const int lane = (i / 4) % 256, dw = i % 4;
    if (h[i] != (float)(lane * 4 + dw)) { ++bad; ++badlane[lane % 16]; ++baddw[dw]; }
  }
  printf("%-60s bad %ld of %zu  by dword [%ld %ld %ld %ld]  by lane%%16 [", what, bad, n, baddw[0], baddw[1], baddw[2], baddw[3]);
  for (int i = 0; i < 16; ++i) printf("%ld ", badlane[i]);
  printf("]\n");
  hipFree(d);
}

int main() {
  run<0, -1>("pk_fma -> dwords 0..1, no wait state");
  run<0, 0>("pk_fma -> dwords 0..1, s_nop 0");
  run<0, 1>("pk_fma -> dwords 0..1, s_nop 1");
  run<0, 3>("pk_fma -> dwords 0..1, s_nop 3");
  run<1, -1>("pk_fma -> dwords 2..3, no wait state");
  run<1, 0>("pk_fma -> dwords 2..3, s_nop 0");
  run<1, 1>("pk_fma -> dwords 2..3, s_nop 1");
  run<1, 3>("pk_fma -> dwords 2..3, s_nop 3");
  run<1, 7>("pk_fma -> dwords 2..3, s_nop 7");
  run<2, -1>("v_mov -> dwords 0,1,2,3, no wait state");
  run<2, 0>("v_mov -> dwords 0,1,2,3, s_nop 0");
  run<2, 1>("v_mov -> dwords 0,1,2,3, s_nop 1");
  run<3, -1>("immediate-offset store, pk_fma both halves, no wait state");
  run<3, 0>("immediate-offset store, pk_fma both halves, s_nop 0");
  return 0;
}
