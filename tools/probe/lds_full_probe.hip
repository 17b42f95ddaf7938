// Does a workgroup really own all 160 KB of a gfx950 CU's LDS?  Writes a pattern to every word of a 163840-byte static
// allocation, reads it back, counts mismatches per 4 KB page.  (hipcc --offload-arch=gfx950 lds_full_probe.hip -o lds_full_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ void __launch_bounds__(256, 1) probe(unsigned* bad) {
  __shared__ unsigned buf[BYTES / 4];
  for (int i = threadIdx.x; i < BYTES / 4; i += 256) buf[i] = 0x9E3779B1u * (unsigned)i + blockIdx.x;
  __syncthreads();
  for (int i = threadIdx.x; i < BYTES / 4; i += 256)
    if (buf[i] != 0x9E3779B1u * (unsigned)i + blockIdx.x) atomicAdd(&bad[i / 1024], 1u);
}
template <int BYTES>
void run() {
  unsigned* bad; hipMalloc(&bad, 64 * 4); hipMemset(bad, 0, 64 * 4);
  hipLaunchKernelGGL(probe<BYTES>, dim3(1024), dim3(256), 0, 0, bad);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[64]; hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
  unsigned tot = 0; for (int i = 0; i < 64; ++i) tot += h[i];
  printf("%d bytes: launch %s, mismatching words %u", BYTES, hipGetErrorString(e), tot);
  for (int i = 0; i < 64; ++i) if (h[i]) printf("  page %d: %u", i, h[i]);
  printf("\n");
}
int main() { run<152576>(); run<162816>(); run<163840>(); return 0; }
