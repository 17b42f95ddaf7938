#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short X[64 * 64];   // X[row][col] = row * 64 + col
  for (int i = threadIdx.x; i < 64 * 64; i += 64) X[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  // group g reads block rows 4g + q (q = 0..3), cols 4p .. 4p+3 (16 columns 0..15)
  const unsigned short* addr = &X[(4 * g + q) * 64 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) printf("lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[4*l]/64, h[4*l]%64, h[4*l+1]/64, h[4*l+1]%64, h[4*l+2]/64, h[4*l+2]%64, h[4*l+3]/64, h[4*l+3]%64);
  return 0;
}
