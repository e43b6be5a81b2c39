// NHWC implicit-GEMM convolution with fp32-equivalent accuracy at the fp16 matrix rate.
//
// Every fp32 operand x is split into two binary16 numbers, x = hi + lo (hi = x truncated
// to 11 significant bits, lo = x - hi, exact, truncated again), and each product is
// expanded as  a*b ~= ah*bh + ah*bl + al*bh  (the dropped al*bl term is 2^-20 relative),
// accumulated in fp32 by v_mfma_f32_32x32x16_f16.  Three fp16 MFMAs (16 k-values each)
// replace eight fp32 MFMAs (2 k-values each): 5.3x the matrix throughput of
// v_mfma_f32_32x32x2_f32 at ~2^-20 per-product error, i.e. fp32-roundoff class.  Filters
// are split offline (per-output-channel power-of-two pre-scaling keeps their lo parts out
// of the fp16 subnormal range; the scale is undone exactly in the epilogue); activations
// stay fp32 in HBM and are split on the fly while being staged into LDS.
//
// Geometry is the fp32 kernel's: 128 x BN tile per 256-thread workgroup, K walked in
// slices of 32 (one filter tap x 32 channels), raw buffer loads (out-of-range offset ==
// zero padding), two register sets (prefetch distance of two MFMA phases), two LDS
// buffers, one barrier per slice, the shared fused epilogue.
//
// LDS row image (128 B per pixel / per filter row and slice):
//   [ hi k0..7 | hi k8..15 | hi k16..23 | hi k24..31 | lo k0..7 | ... | lo k24..31 ]
// eight 16-byte chunks, XOR-swizzled by (row >> 1) & 7 like the fp32 kernel.  The MFMA
// operand of lane l for k-step s is chunk 2s + (l >> 5) (hi) / 4 + 2s + (l >> 5) (lo) of
// row l & 31: one ds_read_b128 each.
#include "fcp_conv_common.h"

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef u32x4_t u32x4;

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}
__device__ __forceinline__ u32x4 buf_load16u(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}

// ASPLIT: the activation tensor is already in split32 format (hi/lo binary16 planes per 32 channels):
// a K slice of a pixel is then byte-for-byte the LDS row image and is staged as a plain copy, exactly
// like the filter; otherwise (fp32 input) it is split on the fly.
template <int BN, bool CIN4, bool ASPLIT>
__global__ void __launch_bounds__(256, (BN == 128 ? 2 : (BN == 64 ? 3 : 4))) conv_igemm_f16x3(const ConvK p) {
  static_assert(!(CIN4 && ASPLIT), "the 3-channel stems always read fp32");
  constexpr int WAVES_N = (BN == 32) ? 1 : 2;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WTM = BM / WAVES_M;
  constexpr int WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int B_LD = BN / 32;       // 16-byte filter chunks per thread per slice
  constexpr int ROWB = 128;           // LDS bytes per row

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);
  char* Bs = As + 2 * BM * ROWB;

  const int nb = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_n = logical % p.grid_n;
  const int tile_m = logical / p.grid_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // activation staging: thread -> (row arow + 64*i, 8-channel group ag): two 16-byte loads each
  const int arow = tid >> 2, ag = tid & 3;
  // filter staging: bytes are already in LDS image order: thread -> (row brow + 32*i, chunk bc)
  const int brow = tid >> 3, bc = tid & 7;

  unsigned pbase[2];
  int hi0[2], wi0[2];
  const int hw = p.out_h * p.out_w;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = tile_m * BM + arow + 64 * i;
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
  unsigned woff[B_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) woff[i] = (unsigned)(((tile_n * BN + brow + 32 * i) * p.wrow + bc * 4) * 4);

  // this thread's four 16-byte activation pieces: (row i, half h).  fp32 input: channels 8ag+4h..+3 of the
  // slice; split32 input: chunk ag of the hi plane (h = 0) / of the lo plane (h = 1), in 4-byte units
  auto achan = [&](int h) -> unsigned { return ASPLIT ? (unsigned)(h * 16 + ag * 4) : (unsigned)(ag * 8 + h * 4); };
  TapPiece tp[2][2];
  unsigned rowoff[2][2];   // byte offset at channel 0 of the current tap, 0xFFFFFFFF = zero padding
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      tp[i][h] = make_tap_piece<CIN4>(p, pbase[i], hi0[i], wi0[i] + (CIN4 ? 2 * ag + h : 0), CIN4 ? 0u : achan(h));
  auto set_tap = [&](int tap, int kh_i, int kw_i) {
    if (p.in_up2) {   // nearest-x2 operand fetch: physical offset is not linear in the tap
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int hi = hi0[i] + kh_i;
          int wi = wi0[i] + (CIN4 ? 2 * ag + h : kw_i);
          const bool ok = (unsigned)hi < (unsigned)p.in_h && (unsigned)wi < (unsigned)p.in_w;
          hi >>= 1; wi >>= 1;
          const unsigned pix = pbase[i] + (unsigned)(hi * p.pw + wi);
          rowoff[i][h] = ok ? (pix * (unsigned)p.in_ld + (CIN4 ? 0u : achan(h))) * 4u : 0xFFFFFFFFu;
        }
    } else {
      const unsigned tapoff = (unsigned)((kh_i * p.pw + kw_i) * p.in_ld) * 4u;   // wave-uniform
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          rowoff[i][h] = ((tp[i][h].mask >> tap) & 1u) ? tp[i][h].base + tapoff : 0xFFFFFFFFu;
    }
  };

  f32x4 ra0[4], ra1[4];
  u32x4 rb0[B_LD], rb1[B_LD];
  auto load_slice = [&](f32x4 (&ra)[4], u32x4 (&rb)[B_LD], int kt, int c0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned ro = rowoff[i][h];
        ra[2 * i + h] = buf_load16(rs_in, ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(c0 * 4));
      }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) rb[i] = buf_load16u(rs_w, woff[i] + (unsigned)(kt * BK * 4));
  };
  // chunk swizzle sw(row) = ((row >> 1) & 7) ^ ((row & 1) << 2): ds_read_b128 (64 banks) sees 16 distinct
  // slots per 16-lane group, and the activation stores below — an 8-lane ds_write_b128 group is rows r, r+1
  // x 4 chunks, banks (addr/4) % 32 so rows 128 B apart alias — land in opposite halves of the row.
  const int asw = ((arow >> 1) & 7) ^ ((arow & 1) << 2);   // identical for rows arow and arow + 64
  const int bsw = ((brow >> 1) & 7) ^ ((brow & 1) << 2);
  auto store_slice = [&](const f32x4 (&ra)[4], const u32x4 (&rb)[B_LD], int buf) {
    char* a = As + buf * BM * ROWB + arow * ROWB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 hi, lo;
      if (ASPLIT) { hi = __builtin_bit_cast(u32x4, ra[2 * i]); lo = __builtin_bit_cast(u32x4, ra[2 * i + 1]); }
      else split8(ra[2 * i], ra[2 * i + 1], hi, lo);
      *reinterpret_cast<u32x4*>(a + 64 * i * ROWB + ((ag ^ asw) << 4)) = hi;
      *reinterpret_cast<u32x4*>(a + 64 * i * ROWB + (((4 + ag) ^ asw) << 4)) = lo;
    }
    char* b = Bs + buf * BN * ROWB + brow * ROWB + ((bc ^ bsw) << 4);
#pragma unroll
    for (int i = 0; i < B_LD; ++i) *reinterpret_cast<u32x4*>(b + 32 * i * ROWB) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int tap = 0, kh_i = 0, kw_i = 0, c0 = 0;
  auto advance = [&]() {      // next slice: taps fastest, then the 32-channel slice
    ++tap;
    if (CIN4) {
      ++kh_i;
    } else if (++kw_i >= p.kw) {
      kw_i = 0;
      if (++kh_i >= p.kh) { kh_i = 0; tap = 0; c0 += BK; }
    }
    set_tap(tap, kh_i, kw_i);
  };

  const int aoff = (wm * WTM + (lane & 31)) * ROWB;
  const int boff = (wn * WTN + (lane & 31)) * ROWB;
  const int rsw = (((lane & 31) >> 1) & 7) ^ ((lane & 1) << 2);
  const int half = lane >> 5;
  int offH[2], offL[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    offH[s] = ((2 * s + half) ^ rsw) << 4;
    offL[s] = ((4 + 2 * s + half) ^ rsw) << 4;
  }

  // ---- one pipeline step, hand-interleaved -------------------------------------------------
  // The wave's instruction stream is in order, so matrix work and the split (VALU) + LDS store of
  // the next slice only overlap if they alternate in program order.  A step is therefore emitted
  // as NM = 2*TM*TN*3 MFMAs with, behind every MFMA, a few "pieces" of the conversion pipeline of
  // slice kt+1 (registers that landed an iteration ago): per float pair A: pack hi, B: unpack hi,
  // C: residual, D: pack lo — two pairs in flight so dependent pieces are an MFMA apart.
  // sched_barrier(0) pins the order (the scheduler otherwise clusters all MFMAs first).
  constexpr int MF = TM * TN * 3;   // MFMAs per k-step
  constexpr int NM = 2 * MF;        // MFMAs per slice
  constexpr int NP = 35;            // conversion pieces: 8 pairs x 4 stages + 2 activation stores + filter store
  auto step = [&](int kt, f32x4 (&ra_ld)[4], u32x4 (&rb_ld)[B_LD], const f32x4 (&ra_st)[4],
                  const u32x4 (&rb_st)[B_LD]) {
    if (kt + 2 < p.ktiles) {
      advance();
      load_slice(ra_ld, rb_ld, kt + 2, c0);
    }
    const int buf = kt & 1, nbuf = buf ^ 1;
    const char* Ab = As + buf * BM * ROWB + aoff;
    const char* Bb = Bs + buf * BN * ROWB + boff;
    char* a_st = As + nbuf * BM * ROWB + arow * ROWB;
    char* b_st = Bs + nbuf * BN * ROWB + brow * ROWB + ((bc ^ bsw) << 4);

    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    auto read_frags = [&](int st) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[st][i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * ROWB + offH[st]);
        al[st][i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * ROWB + offL[st]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[st][j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * ROWB + offH[st]);
        bl[st][j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * ROWB + offL[st]);
      }
    };
    // conversion state of the 8 float pairs of this thread (pairs 0-3: row arow, 4-7: row arow+64)
    decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)) h2[8], l2[8];
    float r0[8], r1[8];
    auto xval = [&](int pr, int e) -> float { return ra_st[pr >> 1][(pr & 1) * 2 + e]; };
    // (runs unconditionally: on the last slice it converts stale registers into the idle LDS stage,
    //  which nobody reads — a branch here would split the block and undo the interleave)
    auto piece = [&](int pc) {
      if (pc == 16 || pc == 33) {          // activation rows: hi and lo chunks
        const int row = pc == 16 ? 0 : 1;
        u32x4 hi, lo;
        if (ASPLIT) {
          hi = __builtin_bit_cast(u32x4, ra_st[2 * row]);
          lo = __builtin_bit_cast(u32x4, ra_st[2 * row + 1]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            hi[q] = __builtin_bit_cast(unsigned, h2[row * 4 + q]);
            lo[q] = __builtin_bit_cast(unsigned, l2[row * 4 + q]);
          }
        }
        *reinterpret_cast<u32x4*>(a_st + 64 * row * ROWB + ((ag ^ asw) << 4)) = hi;
        *reinterpret_cast<u32x4*>(a_st + 64 * row * ROWB + (((4 + ag) ^ asw) << 4)) = lo;
        return;
      }
      if (pc == 34) {                      // filter rows (already split offline): plain copy
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *reinterpret_cast<u32x4*>(b_st + 32 * i * ROWB) = rb_st[i];
        return;
      }
      if (ASPLIT) return;                              // nothing to convert: the copy is the store above
      const int g = pc < 16 ? pc : pc - 1;             // 0..31 over the four 8-piece groups
      const int pr = (g >> 3) * 2 + (g & 1);           // pair index 0..7
      const int stage = (g & 7) >> 1;
      if (stage == 0) h2[pr] = __builtin_amdgcn_cvt_pkrtz(xval(pr, 0), xval(pr, 1));
      else if (stage == 1) { r0[pr] = (float)h2[pr][0]; r1[pr] = (float)h2[pr][1]; }
      else if (stage == 2) { r0[pr] = xval(pr, 0) - r0[pr]; r1[pr] = xval(pr, 1) - r1[pr]; }
      else l2[pr] = __builtin_amdgcn_cvt_pkrtz(r0[pr], r1[pr]);
    };

    read_frags(0);
    read_frags(1);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int st = m / MF, idx = m % MF;
      const int ij = idx / 3, term = idx % 3;
      const int i = ij / TN, j = ij % TN;
      // smallest contributions first: al*bh, ah*bl, ah*bh
      const f16x8 fa = term == 0 ? al[st][i] : ah[st][i];
      const f16x8 fb = term == 1 ? bl[st][j] : bh[st][j];
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pc = m * NP / NM; pc < (m + 1) * NP / NM; ++pc) piece(pc);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };

  set_tap(0, 0, 0);
  load_slice(ra0, rb0, 0, c0);
  store_slice(ra0, rb0, 0);
  if (p.ktiles > 1) {
    advance();
    load_slice(ra1, rb1, 1, c0);
  }
  __syncthreads();
  for (int kt = 0; kt < p.ktiles; kt += 2) {
    step(kt, ra0, rb0, ra1, rb1);
    if (kt + 1 < p.ktiles) step(kt + 1, ra1, rb1, ra0, rb0);
  }

  if (p.out_fmt | p.res1_fmt | p.res2_fmt)
    conv_epilogue8<BN, TM, TN, WTM, WTN>(p, acc, smem, tile_m, tile_n, tid, lane, wm, wn, hw);
  else
    conv_epilogue<BN, TM, TN, WTM, WTN>(p, acc, smem, tile_m, tile_n, tid, lane, wm, wn, hw);
}

template <int BN, bool CIN4, bool ASPLIT>
int launch(const ConvK& k, hipStream_t s) {
  const size_t lds = (size_t)2 * (BM + BN) * 128;
  FCP_LDS_OPT_IN((&conv_igemm_f16x3<BN, CIN4, ASPLIT>), lds);
  hipLaunchKernelGGL((conv_igemm_f16x3<BN, CIN4, ASPLIT>), dim3(k.grid_m * k.grid_n), dim3(256), lds, s, k);
  FCP_LAUNCH_OK();
  return 0;
}

}  // namespace

namespace fcp_conv {

int launch_f16x3(const ConvK& k, int tile_n, bool cin4, hipStream_t s) {
  if (cin4) {
    switch (tile_n) {
      case 32: return launch<32, true, false>(k, s);
      case 64: return launch<64, true, false>(k, s);
      default: return launch<128, true, false>(k, s);
    }
  }
  if (k.in_fmt == 1) {
    switch (tile_n) {
      case 32: return launch<32, false, true>(k, s);
      case 64: return launch<64, false, true>(k, s);
      default: return launch<128, false, true>(k, s);
    }
  }
  switch (tile_n) {
    case 32: return launch<32, false, false>(k, s);
    case 64: return launch<64, false, false>(k, s);
    default: return launch<128, false, false>(k, s);
  }
}

}  // namespace fcp_conv
