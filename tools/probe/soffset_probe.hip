// What the buffer range check of gfx950 looks at: voffset alone, or voffset + soffset?  (raw buffer, stride 0)
//   hipcc -O3 --offload-arch=gfx950 -o soffset_probe soffset_probe.hip && ./soffset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* buf, unsigned bytes, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const int soff = __builtin_amdgcn_readfirstlane(64);
  // case 0: voffset in range, soffset keeps it in range
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, 16u, soff, 0);
  // case 1: voffset = ~0u (this kernel's "row outside the matrix"), soffset 64
  u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, ~0u, soff, 0);
  // case 2: voffset in range, voffset + soffset behind the range (range 256 bytes: 224 + 64)
  u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(r, 224u, soff, 0);
  // case 3: voffset behind the range by itself
  u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, 256u, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = __uint_as_float(a.x); out[1] = __uint_as_float(b.x); out[2] = __uint_as_float(c.x); out[3] = __uint_as_float(d.x);
  }
}
int main() {
  float *buf, *out, h[1024], o[4];
  for (int i = 0; i < 1024; ++i) h[i] = 1000.f + i;
  hipMalloc(&buf, 4096); hipMalloc(&out, 16);
  hipMemcpy(buf, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, 256u, out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("fault\n"); return 1; }
  hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
  printf("range 256 B.  voffset 16 + soffset 64 -> %.0f (element %d expected 1020)\n", o[0], (16 + 64) / 4);
  printf("voffset ~0u + soffset 64 -> %.0f (0 = out of range; 1015 = wrapped into range)\n", o[1]);
  printf("voffset 224 + soffset 64 (= 288 >= 256) -> %.0f (0 = soffset is range-checked; 1072 = it is not)\n", o[2]);
  printf("voffset 256, soffset 0 -> %.0f (0 expected)\n", o[3]);
  return 0;
}
