// Semantics of the gfx950 lane-swap builtins the register epilogues rely on (fcp_conv_f16x3_big.hip):
//   r = __builtin_amdgcn_permlane32_swap(a, b, false, false):  r[0] = {a[0:31],  b[0:31]},   r[1] = {a[32:63], b[32:63]}
//   r = __builtin_amdgcn_permlane16_swap(a, b, false, false):  r[0] = rows {a.r0, b.r0, a.r2, b.r2},  r[1] = rows {a.r1, b.r1, a.r3, b.r3}
// (rows = 16 lanes).  hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_swap.hip -o /tmp/pls && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d;
  if (hipMalloc(&d, 256 * 4) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  unsigned h[256];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int l : {0, 15, 16, 31, 32, 47, 48, 63})
    printf("lane %2d: swap32 -> %3u %3u   swap16 -> %3u %3u   (a = lane, b = 100 + lane)\n", l, h[l], h[64 + l], h[128 + l], h[192 + l]);
  return 0;
}
