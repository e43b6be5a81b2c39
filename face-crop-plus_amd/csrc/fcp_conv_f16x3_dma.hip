// fp16x3 convolution, both operands fed by LDS-DMA.
//
// When the activation tensor is in split32 format, a K slice of a pixel (one filter tap x 32
// channels) is byte-for-byte the 128-byte LDS row image [hi k0..31 | lo k0..31], exactly like a
// row of the offline-split filter.  Both tiles are then moved global -> LDS by
// `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction, out-of-range lanes write zeros = the
// conv's zero padding): no staging registers, no conversion, no ds_write.  The 16-byte chunks of a
// row are XOR-swizzled on the *source* side (lane l of the DMA writes LDS position l & 7, so it
// fetches chunk (l & 7) ^ sw(row)); the fragment reads apply the same involution.
//
// Pipeline (NST LDS stages, one barrier per slice): all fragment reads of slice kt are issued
// first, then the DMA of slice kt+NST-1 (so the compiler never has an LDS read behind a pending
// DMA), then the 24 MFMAs; a counted vmcnt + s_barrier closes the step.
#include "fcp_conv_common.h"

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// EP8: the output or a residual is split32 (cout % 8 == 0): the tile is accumulated transposed (filters x pixels — same bits) and
// the epilogue works from the registers (conv_epilogue_regs); else the pixel-major tile and the staged fp32 epilogue.
template <int BN, int NST, bool EP8>
__global__ void __launch_bounds__(256, (BN == 128 ? 2 : 3)) conv_igemm_f16x3_dma(const ConvK p) {
  constexpr int WAVES_N = (BN == 32) ? 1 : 2;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WTM = BM / WAVES_M;
  constexpr int WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_LD = BM / 32;
  constexpr int B_LD = BN / 32;
  constexpr int ROWB = 128;
  constexpr int STAGE = (BM + BN) * ROWB;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);

  const int nb = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = logical % p.grid_n;
  const int tile_m = logical / p.grid_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = tid >> 3;                                     // 0..31 (+32 i)
  const int csrc = (tid & 7) ^ (((lrow >> 1) & 7) ^ ((lrow & 1) << 2));   // source chunk of LDS position tid & 7

  unsigned pbase[A_LD];
  int hi0[A_LD], wi0[A_LD];
  const int hw = p.out_h * p.out_w;
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    if (m < p.M) {
      const int ni = m / hw;
      const int rem = m - ni * hw;
      const int ho = rem / p.out_w;
      const int wo = rem - ho * p.out_w;
      pbase[i] = (unsigned)(ni * p.ph * p.pw);
      hi0[i] = ho * p.stride - p.pad_h;
      wi0[i] = wo * p.stride - p.pad;
    } else {
      pbase[i] = 0;
      hi0[i] = -(1 << 28);
      wi0[i] = 0;
    }
  }
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  // optional second source (1x1 convs): channels >= csplit come from in2 sampled at stride2
  __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2 ? p.in2 : p.in), 0, p.in2_bytes, 0x00020000);
  unsigned base2[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    base2[i] = 0xFFFFFFFFu;
    if (p.in2 != nullptr && m < p.M) {
      const int ni = m / hw;
      const int rem = m - ni * hw;
      const int ho = rem / p.out_w;
      const int wo = rem - ho * p.out_w;
      base2[i] = (((unsigned)(ni * p.ph2 + ho * p.stride2) * (unsigned)p.pw2 + (unsigned)(wo * p.stride2)) * (unsigned)p.in2_ld +
                  (unsigned)(csrc * 4)) * 4u;
    }
  }
  unsigned woff[B_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) woff[i] = (unsigned)(((tile_n * BN + lrow + 32 * i) * p.wrow + csrc * 4) * 4);
  TapPiece tp[A_LD];
  unsigned rowoff[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) tp[i] = make_tap_piece<false>(p, pbase[i], hi0[i], wi0[i], (unsigned)(csrc * 4));
  auto set_tap = [&](int tap, int kh_i, int kw_i) {
    if (p.in_up2) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int hi = hi0[i] + kh_i;
        int wi = wi0[i] + kw_i;
        const bool ok = (unsigned)hi < (unsigned)p.in_h && (unsigned)wi < (unsigned)p.in_w;
        hi >>= 1; wi >>= 1;
        const unsigned pix = pbase[i] + (unsigned)(hi * p.pw + wi);
        rowoff[i] = ok ? (pix * (unsigned)p.in_ld + (unsigned)(csrc * 4)) * 4u : 0xFFFFFFFFu;
      }
    } else {
      const unsigned tapoff = (unsigned)((kh_i * p.pw + kw_i) * p.in_ld) * 4u;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) rowoff[i] = ((tp[i].mask >> tap) & 1u) ? tp[i].base + tapoff : 0xFFFFFFFFu;
    }
  };
  int tap = 0, kh_i = 0, kw_i = 0, c0 = 0;
  auto advance = [&]() {
    ++tap;
    if (++kw_i >= p.kw) {
      kw_i = 0;
      if (++kh_i >= p.kh) { kh_i = 0; tap = 0; c0 += BK; }
    }
    set_tap(tap, kh_i, kw_i);
  };
  // one slice: A_LD + B_LD DMA instructions per wave, each 64 lanes x 16 B = rows 8*wave + 32*i .. +7
  auto dma_slice = [&](int kt, int stage) {
    char* a = lds + stage * STAGE + wave_u * 8 * ROWB;
    char* b = a + BM * ROWB;
    if (c0 >= p.csplit) {                  // wave-uniform: this slice comes from the second source
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const unsigned ro = base2[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in2, (__attribute__((address_space(3))) void*)(a + 32 * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)((c0 - p.csplit) * 4)), 0, 0, FCP_AUX_A);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const unsigned ro = rowoff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(c0 * 4)), 0, 0, FCP_AUX_A);
      }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b + 32 * i * ROWB), 16,
                                               (int)(woff[i] + (unsigned)(kt * BK * 4)), 0, 0, FCP_AUX_B);
  };

  // residual tile first: its HBM round trip overlaps the whole main loop
  ResPrefetch<BN> rpre;
  rpre.valid = false;
  (void)rpre;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int aoff = (wm * WTM + (lane & 31)) * ROWB;
  const int boff = BM * ROWB + (wn * WTN + (lane & 31)) * ROWB;
  const int rsw = (((lane & 31) >> 1) & 7) ^ ((lane & 1) << 2);
  const int half = lane >> 5;
  int offH[2], offL[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    offH[s] = ((2 * s + half) ^ rsw) << 4;
    offL[s] = ((4 + 2 * s + half) ^ rsw) << 4;
  }

  set_tap(0, 0, 0);
  dma_slice(0, 0);
  if (NST == 3 && p.ktiles > 1) {
    advance();
    dma_slice(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  int stage = 0;
  const int kt_end = p.ktiles;
  for (int kt = 0; kt < kt_end; ++kt) {
    const char* Ab = lds + stage * STAGE + aoff;
    const char* Bb = lds + stage * STAGE + boff;
    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[s][i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * ROWB + offH[s]);
        al[s][i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * ROWB + offL[s]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[s][j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * ROWB + offH[s]);
        bl[s][j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * ROWB + offL[s]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // fragments are in registers
    __builtin_amdgcn_sched_barrier(0);
    const int ahead = kt + NST - 1;
    const bool issue = ahead < p.ktiles;
    if (issue) {
      advance();
      int st = stage + NST - 1;
      if (st >= NST) st -= NST;
      dma_slice(ahead, st);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (EP8) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s][j], al[s][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s][j], ah[s][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s][j], ah[s][i], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], acc[i][j], 0, 0, 0);
          }
        }
    __builtin_amdgcn_sched_barrier(0);   // keep the waits below behind the MFMAs ("memory" does not order MFMAs)
    // slice kt+1 must have landed before anyone reads it; with three stages the DMA issued in this
    // step (slice kt+2) may stay in flight across the barrier
    if (NST == 3 && issue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (++stage >= NST) stage = 0;
  }

  if constexpr (EP8) {
    const int m_end = (tile_m + 1) * BM < p.M ? (tile_m + 1) * BM : p.M;
    conv_epilogue_regs<TM, TN, WTM, WTN>(p, acc, tile_m * BM, m_end, TM, tile_n * BN, (tile_n + 1) * BN < p.cout ? (tile_n + 1) * BN : p.cout,
                                         wm, wn, lane, hw);
  } else {
    conv_epilogue<BN, TM, TN, WTM, WTN>(p, acc, smem, tile_m, tile_n, tid, lane, wm, wn, hw);
  }
}

template <int BN, int NST, bool EP8>
int launch_ep(const ConvK& k, hipStream_t s) {
  size_t lds = (size_t)NST * (BM + BN) * 128;
  const size_t epi = EP8 ? 0 : (size_t)BM * BN * 4;       // the staged epilogue's C tile
  if (lds < epi) lds = epi;
  FCP_LDS_OPT_IN((&conv_igemm_f16x3_dma<BN, NST, EP8>), lds);
  hipLaunchKernelGGL((conv_igemm_f16x3_dma<BN, NST, EP8>), dim3(k.grid_m * k.grid_n), dim3(256), lds, s, k);
  FCP_LAUNCH_OK();
  return 0;
}
template <int BN, int NST>
int launch(const ConvK& k, hipStream_t s) {
  const bool ep8 = (k.out_fmt | k.res1_fmt | k.res2_fmt) != 0;
  return ep8 ? launch_ep<BN, NST, true>(k, s) : launch_ep<BN, NST, false>(k, s);
}

}  // namespace

namespace fcp_conv {

int launch_f16x3_dma(const ConvK& k, int tile_n, int stages, hipStream_t s) {
  if (stages == 3) {
    switch (tile_n) {
      case 32: return launch<32, 3>(k, s);
      case 64: return launch<64, 3>(k, s);
      default: return launch<128, 3>(k, s);
    }
  }
  switch (tile_n) {
    case 32: return launch<32, 2>(k, s);
    case 64: return launch<64, 2>(k, s);
    default: return launch<128, 2>(k, s);
  }
}

}  // namespace fcp_conv
