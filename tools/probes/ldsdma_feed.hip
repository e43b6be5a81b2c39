// Microbenchmark: how many bytes per clock can one CU pull into LDS with `buffer_load ... lds` (1 KiB per wave instruction)?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_feed.hip -o /tmp/ldsdma_feed && /tmp/ldsdma_feed
// One 512-thread workgroup per CU (8 waves), each iteration every wave issues PIECES instructions (PIECES KiB) and waits
// for them (vmcnt(0)) before the next round; MODE picks where the bytes come from:
//   0: a private 64 KiB region per workgroup, re-read every iteration            (L2 / L1 resident)
//   1: ONE 32 KiB region shared by all workgroups + a private 32 KiB             (the conv's filter + activation mix)
//   2: a private stream through a 1 GiB buffer, never re-read                    (HBM)
//   3: a private stream through a 160 MiB buffer, re-read on every launch        (Infinity Cache)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PIECES, int WAIT_EACH>
__global__ void __launch_bounds__(512, 1) feed(const float* src, unsigned bytes, int iters, int mode, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  const unsigned wg = blockIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int stage = it & 1;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      unsigned off;
      const unsigned piece = (unsigned)(wave * PIECES + p);          // 0 .. 8*PIECES-1 KiB pieces of this iteration
      if (mode == 0) off = wg * 65536u + (piece * 1024u) % 65536u;
      else if (mode == 1) off = (p < PIECES / 2 ? (piece * 1024u) % 32768u : 32768u + wg * 32768u + (piece * 1024u) % 32768u);
      else off = (unsigned)(((unsigned long long)wg * iters + it) * (8u * PIECES * 1024u) % (bytes - 8u * PIECES * 1024u)) / 1024u * 1024u + piece * 1024u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + stage * 65536 + (piece * 1024u) % 65536u), 16,
                                               (int)(off + lane * 16u), 0, 0, 0);
    }
    if (WAIT_EACH) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // one iteration in flight behind
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PIECES, int WAIT_EACH>
void run(const float* src, size_t bytes, int mode, int cus, unsigned long long* dcyc) {
  const int iters = 400;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&feed<PIECES, WAIT_EACH>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<PIECES, WAIT_EACH>), dim3(cus), dim3(512), 131072, 0, src, (unsigned)(bytes > 0xFFFFFF00ull ? 0xFFFFFF00ull : bytes), iters, mode, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(cus);
  hipMemcpy(c.data(), dcyc, cus * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : c) avg += (double)v;
  avg /= cus;
  const double per_cu = (double)iters * 8 * PIECES * 1024;
  printf("mode %d  %2d KiB/wave/iter wait_each=%d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (cycle counter)  %6.0f cycles/iter\n", mode, PIECES, WAIT_EACH,
         ms * 1e3, per_cu * cus / (ms * 1e-3) / 1e12, per_cu / avg, avg / iters);
}

int main() {
  int cus = 256;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  cus = prop.multiProcessorCount;
  float* src;
  const size_t bytes = 1ull << 30;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  unsigned long long* dcyc;
  hipMalloc(&dcyc, cus * 8);
  for (int mode = 0; mode < 4; ++mode) {
    const size_t b = mode == 3 ? (160ull << 20) : bytes;
    run<8, 1>(src, b, mode, cus, dcyc);
    run<8, 0>(src, b, mode, cus, dcyc);
    run<4, 1>(src, b, mode, cus, dcyc);
    run<4, 0>(src, b, mode, cus, dcyc);
    run<2, 0>(src, b, mode, cus, dcyc);
  }
  return 0;
}
