// Semantics of __builtin_amdgcn_permlane32_swap(a, b, fi, bc) on gfx950: which halves move where.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r[0] = %3u  r[1] = %3u   (a = lane, b = 100 + lane)\n", l, h[l], h[64 + l]);
  return 0;
}
