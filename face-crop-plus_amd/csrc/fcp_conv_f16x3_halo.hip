// fp16x3 3x3 convolution (stride 1, pad 1) for narrow outputs (cout <= 32): halo-tile implicit GEMM.
//
// The generic kernels fetch a 128-byte operand row per output pixel, filter tap and 32-channel slice:
// with N = 32 output channels that is 20 KiB of LDS-DMA per 6 MFMAs of a wave, and the RRDB dense
// blocks (rrdb.py / _layers.py:168-200: 276 of the 351 convs have 32 filters) run at the operand-feed
// limit.  Here a workgroup owns an 8 x 32 patch of output pixels and stages, per 32-channel slice,
// the (8+2) x (32+2) halo patch ONCE; the nine taps read it at shifted row indices.  Operand traffic
// per slice drops from 9 x 32 KiB to 43 KiB (+ 36 KiB of filter taps), a whole slice (108 MFMAs per
// wave) hides one DMA round trip, and there is one barrier per slice.
//
//  * 4 waves; wave w computes image rows 2w, 2w+1 of the patch (two 32x32 MFMA tiles) x 32 filters.
//  * LDS stage = halo rows [344][128 B] + filter rows [9 taps][32][128 B]; two stages (158 KiB).
//  * Same arithmetic as the other fp16x3 kernels (al*bh + ah*bl + ah*bh per k-half, channel slices
//    outer, taps inner) => bit-identical results.
//  * Fragment reads are inline-asm ds_read_b128 (see fcp_conv_f16x3_big.hip), prefetched one tap ahead.
#include "fcp_conv_common.h"

#include <type_traits>

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 32;            // output patch
constexpr int HW_ = TW + 2;               // halo patch width (34); height TH + 2 = 10
constexpr int HROWS = 344;                // 340 halo rows padded to a multiple of 8
constexpr int ROWB = 128;
constexpr int A_BYTES = HROWS * ROWB;     // 44032
constexpr int B_BYTES = 9 * 32 * ROWB;    // 36864
constexpr int STAGE_B = A_BYTES + B_BYTES;   // 80896
constexpr int A_LD = 11;                  // DMA instructions per thread: 2752 16-byte pieces / 256 (last: waves 0..2)

__device__ __forceinline__ f16x8 lds_read128v(unsigned addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }

__global__ void __launch_bounds__(256, 1) conv3x3_halo_f16x3(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);

  const int tiles_x = (p.out_w + TW - 1) / TW, tiles_y = (p.out_h + TH - 1) / TH;
  const int nb = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tx = logical % tiles_x;
  const int ty = (logical / tiles_x) % tiles_y;
  const int ni = logical / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = tid >> 3;                          // 0..31 (+32 i)
  const int csrc = (tid & 7) ^ swz(lrow);             // rows differ by multiples of 32: one swizzle per thread

  // ---- DMA sources.  Halo row hr = lrow + 32 i -> input pixel (y0 - 1 + hr / 34, x0 - 1 + hr % 34).
  unsigned abase[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int hr = lrow + 32 * i;
    const int hy = hr / HW_, hx = hr - hy * HW_;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hr < (TH + 2) * HW_ && (unsigned)y < (unsigned)p.in_h && (unsigned)x < (unsigned)p.in_w;
    abase[i] = ok ? ((unsigned)((ni * p.in_h + y) * p.in_w + x) * (unsigned)p.in_ld + (unsigned)(csrc * 4)) * 4u : 0xFFFFFFFFu;
  }
  // filter row n = lrow: its 9 taps of one channel slice are contiguous (9 x 128 B)
  const unsigned wbase = (unsigned)((lrow * p.wrow) * 4 + csrc * 16);
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  auto dma_slice = [&](int cs, int stage) {
    char* a = lds + stage * STAGE_B + wave_u * 8 * ROWB;
#pragma unroll
    for (int i = 0; i < A_LD - 1; ++i) {
      const unsigned ro = abase[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * i * ROWB), 16,
                                               (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(cs * 128)), 0, 0, 0);
    }
    if (wave_u < 3) {   // rows 320..343
      const unsigned ro = abase[A_LD - 1];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * (A_LD - 1) * ROWB), 16,
                                               (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(cs * 128)), 0, 0, 0);
    }
    char* b = lds + stage * STAGE_B + A_BYTES + wave_u * 8 * ROWB;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b + 32 * t * ROWB), 16,
                                               (int)(wbase + (unsigned)((cs * 9 + t) * 128)), 0, 0, 0);
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // ---- fragment addressing
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int xl = lane & 31, half = lane >> 5;
  const int rb0 = (2 * wave_u) * HW_ + xl;            // halo row of (image row 2w, column xl) at tap (0,0)
  const int bsw = swz(xl);
  unsigned boff[4];                                   // filter fragment offsets inside a tap: [hi s0, hi s1, lo s0, lo s1]
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    boff[s] = (unsigned)(xl * ROWB + (((2 * s + half) ^ bsw) << 4));
    boff[2 + s] = (unsigned)(xl * ROWB + (((4 + 2 * s + half) ^ bsw) << 4));
  }

  f16x8 fah[2][2][2], fal[2][2][2], fbh[2][2], fbl[2][2];   // [set][tile | k-half][k-half]
  auto read_frags = [&](auto set_c, int tap, unsigned stage_off) {
    constexpr int set = decltype(set_c)::value;
    const int kh = tap / 3, kw = tap - kh * 3;
    const unsigned sbase = lds0 + stage_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hr = rb0 + (i + kh) * HW_ + kw;
      const int sw = swz(hr);
      const unsigned ra = sbase + (unsigned)(hr * ROWB);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        fah[set][i][s] = lds_read128v(ra + (unsigned)(((2 * s + half) ^ sw) << 4));
        fal[set][i][s] = lds_read128v(ra + (unsigned)(((4 + 2 * s + half) ^ sw) << 4));
      }
    }
    const unsigned rbq = sbase + (unsigned)(A_BYTES + tap * 32 * ROWB);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      fbh[set][s] = lds_read128v(rbq + boff[s]);
      fbl[set][s] = lds_read128v(rbq + boff[2 + s]);
    }
  };
  auto mfmas = [&](auto set_c) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[set][i][s], fbh[set][s], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[set][i][s], fbl[set][s], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[set][i][s], fbh[set][s], acc[i], 0, 0, 0);
    }
  };
  constexpr std::integral_constant<int, 0> SET0{};
  constexpr std::integral_constant<int, 1> SET1{};

  // ---- prologue
  const int nslices = p.ctiles;
  dma_slice(0, 0);
  if (nslices > 1) {
    dma_slice(1, 1);
    // slice 1 is 19 (wave 3) or 20 (waves 0..2) operations: at most 19 outstanding => slice 0 has landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + 9 - 1) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(SET0, 0, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- main loop over (slice, tap), two steps per iteration so the fragment set is a compile-time index
  int cs = 0, tap = 0;
  unsigned soff = 0u;                                   // byte offset of the stage holding slice cs
  const int total = nslices * 9;
  auto step = [&](auto set_c, auto nset_c, int g) {
    // prefetch the fragments of step g+1 into the other set, then run step g's MFMAs
    const bool more = g + 1 < total;
    if (more) {
      if (tap == 8) {
        // next step starts slice cs+1: its stage must have landed for everyone; this slice's stage is
        // dead afterwards (tap 8's fragments are already in registers), so slice cs+2 may overwrite it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned noff = soff ? 0u : (unsigned)STAGE_B;
        read_frags(nset_c, 0, noff);
        if (cs + 2 < nslices) dma_slice(cs + 2, cs & 1);
      } else {
        read_frags(nset_c, tap + 1, soff);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(set_c);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (++tap == 9) { tap = 0; ++cs; soff = soff ? 0u : (unsigned)STAGE_B; }
  };
  for (int g = 0; g < total; g += 2) {
    step(SET0, SET1, g);
    if (g + 1 < total) step(SET1, SET0, g + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue through a [256 pixels][32 channels] fp32 LDS tile
  float* Cs = smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = (2 * wave_u + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      Cs[row * 32 + xl] = acc[i][rr];
    }
  __syncthreads();
  const int ccol = (tid & 3) * 8;
  if (ccol >= p.cout) return;
  float bias8[8], ws8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bias8[e] = p.bias != nullptr ? p.bias[ccol + e] : 0.f;
    ws8[e] = p.wscale[ccol + e];
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = (tid >> 2) + 64 * g;
    const int y = y0 + (row >> 5), x = x0 + (row & 31);
    if (y >= p.out_h || x >= p.out_w) continue;
    const long m = ((long)ni * p.out_h + y) * p.out_w + x;
    float v[8], r1[8], r2[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + row * 32 + ccol);
      const f32x4 b = *reinterpret_cast<const f32x4*>(Cs + row * 32 + ccol + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    if (p.res1 != nullptr) load8(p.res1, m, p.res1_ld, ccol, p.res1_fmt, r1);
    if (p.res2 != nullptr) load8(p.res2, m, p.res2_ld, ccol, p.res2_fmt, r2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[e] * ws8[e] + bias8[e];
      if (p.res1 != nullptr && p.res1_pre) t += r1[e];
      t = t >= 0.f ? t : t * p.act_slope;
      t = t * p.alpha;
      if (p.res1 != nullptr && !p.res1_pre) t += r1[e];
      if (p.res2 != nullptr) t = t * p.alpha2 + r2[e];
      v[e] = t;
    }
    if (p.out_fmt == 1) {
      u32x4_t hi, lo;
      split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
      char* ob = reinterpret_cast<char*>(p.out) + m * p.out_ld * 4 + split_chan_off(ccol);
      *reinterpret_cast<u32x4_t*>(ob) = hi;
      *reinterpret_cast<u32x4_t*>(ob + 64) = lo;
    } else {
      float* dst = p.out + m * p.out_ld + ccol;
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

}  // namespace

namespace fcp_conv {

int launch_f16x3_halo(const ConvK& k, hipStream_t s) {
  const size_t lds = 2 * (size_t)STAGE_B;
  FCP_LDS_OPT_IN(&conv3x3_halo_f16x3, lds);
  const long tiles = (long)k.n * ((k.out_h + TH - 1) / TH) * ((k.out_w + TW - 1) / TW);
  FCP_REQUIRE(tiles < (1L << 31), "conv(halo): too many tiles");
  hipLaunchKernelGGL(conv3x3_halo_f16x3, dim3((unsigned)tiles), dim3(256), lds, s, k);
  FCP_LAUNCH_OK();
  return 0;
}

}  // namespace fcp_conv
