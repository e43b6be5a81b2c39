// Shader clock under MFMA load: ratio of clock64() (shader cycles) to wall_clock64() (100 MHz).
// hipcc --offload-arch=gfx950 -O3 tools/probes/clk_probe.hip -o tools/probes/clk_probe.bin; gpurun -- ./tools/probes/clk_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_loop(const unsigned* __restrict__ seed, int iters, float* out, long long* clk) {
  f16x8 a, b;
  unsigned s = seed[threadIdx.x & 63] * (threadIdx.x + 1);
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    a[i] = seed[64] ? (_Float16)((int)(s >> 20) - 2048) * (_Float16)0.001f : (_Float16)0.f;
    s = s * 1664525u + 1013904223u;
    b[i] = seed[64] ? (_Float16)((int)(s >> 20) - 2048) * (_Float16)0.001f : (_Float16)0.f;
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
  const int blocks = 256 * 2;
  unsigned h[65];
  for (int i = 0; i < 64; ++i) h[i] = 12345u + 977u * i;
  unsigned* ds; float* out; long long* clk;
  hipMalloc(&ds, sizeof(h)); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  for (int data = 0; data < 2; ++data) {
    h[64] = data;
    hipMemcpy(ds, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 6; ++rep) {
      const int iters = 200000;   // 800k MFMAs per wave ~ 25.6M cycles ~ 11 ms
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      mfma_loop<<<blocks, 256>>>(ds, iters, out, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> c(blocks * 2);
      hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < blocks; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
      const double flops = (double)blocks * 4 * iters * 4.0 * 32 * 32 * 16 * 2;
      printf("data=%d rep=%d  %.2f ms  %.0f TFLOP/s  clock64/wall = %.3f (x100MHz => %.0f MHz)  cycles/MFMA/wave = %.2f\n", data, rep, ms,
             flops / ms / 1e9, cyc / wall, cyc / wall * 100, cyc / blocks / (4.0 * iters));
    }
  }
  return 0;
}
