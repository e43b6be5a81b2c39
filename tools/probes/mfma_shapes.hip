// Which MFMA shape does the most FLOPs per joule?  Bare loops on register operands (random data), whole chip, long enough
// to sit at the socket power cap; TFLOP/s at the cap is the energy efficiency.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shapes.hip -o tools/probes/mfma_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ void __launch_bounds__(256) mfma_loop(const unsigned* __restrict__ seed, int iters, float* out, long long* clk) {
  f16x8 a[4], b[4];
  unsigned s = seed[threadIdx.x & 63] * (threadIdx.x + 1);
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u;
      a[q][i] = (_Float16)((int)(s >> 20) - 2048) * (_Float16)0.001f;
      s = s * 1664525u + 1013904223u;
      b[q][i] = (_Float16)((int)(s >> 20) - 2048) * (_Float16)0.001f;
    }
  long long t0 = clock64(), w0 = wall_clock64();
  float r = 0;
  if constexpr (SHAPE == 32) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[3], c3, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  } else {
    f32x4 c[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 3], b[(q + (q >> 2)) & 3], c[q], 0, 0, 0);
    }
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 4; ++i) r += c[q][i];
  }
  long long t1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
  const int blocks = 256 * 2;
  unsigned h[65];
  for (int i = 0; i < 64; ++i) h[i] = 12345u + 977u * i;
  unsigned* ds; float* out; long long* clk;
  hipMalloc(&ds, sizeof(h)); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipMemcpy(ds, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shape = 0; shape < 2; ++shape)
    for (int rep = 0; rep < 40; ++rep) {
      const int iters = 400000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      double flops;
      if (shape == 0) { mfma_loop<32><<<blocks, 256>>>(ds, iters, out, clk); flops = (double)blocks * 4 * iters * 4.0 * 32 * 32 * 16 * 2; }
      else { mfma_loop<16><<<blocks, 256>>>(ds, iters, out, clk); flops = (double)blocks * 4 * iters * 8.0 * 16 * 16 * 32 * 2; }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> c(blocks * 2);
      hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < blocks; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
      if (rep % 8 == 7) {
        printf("shape=%s rep=%d  %.2f ms  %.0f TFLOP/s  sclk ~ %.0f MHz\n", shape ? "16x16x32" : "32x32x16", rep, ms, flops / ms / 1e9, cyc / wall * 100);
        fflush(stdout);
      }
    }
  return 0;
}
