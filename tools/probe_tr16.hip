// Probe: source (lane, element) -> destination map of ds_read_b64_tr_b16 (raw 16-bit payloads).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short s[256];
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = in[i];
  __syncthreads();
  int lane = threadIdx.x;
  auto* ptr = (__attribute__((address_space(3))) s16x4*)(s + lane * 4);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  unsigned long long raw = __builtin_bit_cast(unsigned long long, v);
  *(unsigned long long*)(out + lane * 4) = raw;
}
int main() {
  unsigned short h[256], o[256];
  unsigned short *d, *dout;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) h[l * 4 + e] = (unsigned short)(l * 16 + e);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, dout);
  (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("dest lane: 4 elements as (srcLane.srcElem)\n");
  for (int l = 0; l < 32; ++l) { printf("L%02d:", l); for (int j = 0; j < 4; ++j) printf(" %2d.%d", o[l * 4 + j] / 16, o[l * 4 + j] % 16); printf(l % 4 == 3 ? "\n" : "  |  "); }
  return 0;
}
