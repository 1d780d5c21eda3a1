// Probe: source (lane, byte) -> destination map of ds_read_b64_tr_b8 (raw byte payloads).
// Every lane passes the address of its own 8 bytes (lane * 8); source byte value = its LDS byte offset / .. id.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned short* in, unsigned short* out, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
  __syncthreads();
  int lane = threadIdx.x;
  // 16-bit ids per BYTE are not possible: use two passes (low / high id bits) driven by `in`
  auto* ptr = (__attribute__((address_space(3))) i32x2*)((unsigned char*)s + lane * stride);
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32(ptr);
  unsigned long long raw = __builtin_bit_cast(unsigned long long, v);
  *(unsigned long long*)((unsigned char*)out + lane * 8) = raw;
}
int main() {
  unsigned char h[8192], lo[512], hi[512];
  unsigned short *d, *dout;
  (void)hipMalloc(&d, 8192); (void)hipMalloc(&dout, 512);
  for (int stride = 8; stride <= 32; stride *= 2) {
    // pass 0: byte value = (byte offset / stride... ) we encode source lane in pass 0, source byte-in-lane in pass 1
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = 0; i < 8192; ++i) h[i] = 0xff;
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) h[l * stride + e] = pass == 0 ? (unsigned char)l : (unsigned char)e;
      (void)hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
      k<<<1, 64>>>(d, dout, stride);
      (void)hipMemcpy(pass == 0 ? lo : hi, dout, 512, hipMemcpyDeviceToHost);
    }
    printf("stride %d bytes between the lanes' addresses; dest lane: 8 bytes as (srcLane.srcByte)\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("L%02d:", l);
      for (int j = 0; j < 8; ++j) printf(" %2d.%d", lo[l * 8 + j], hi[l * 8 + j]);
      printf(l % 2 == 1 ? "\n" : "  |  ");
    }
  }
  return 0;
}
