// Shared pieces of the convolution kernels (fp32-exact and fp16x3-split variants):
// kernel parameter block and the fused epilogue.
#pragma once
#include <type_traits>
#include "fcp_common.h"
#include "fcp_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// Cache policy of the LDS-DMA operand loads (the `aux` immediate of raw.ptr.buffer.load.lds on gfx950: 1 = sc0, 2 = nt,
// 16 = sc1), activations (A) and filters (B): default policy for both — nt / sc1 on either operand measured equal or slower
// (profiles/r03_probes.md section 1).
constexpr int FCP_AUX_A = 0;
constexpr int FCP_AUX_B = 0;

namespace fcp_conv {

constexpr int BM = 128;   // output pixels per workgroup tile
constexpr int BK = 32;    // K slice: one filter tap x 32 input channels

struct ConvK {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  const float* res1;
  const float* res2;
  int n, in_h, in_w, ph, pw, cin, in_ld, in_up2;
  int cout, kh, kw, stride, pad, out_h, out_w, out_ld;
  int pad_h;      // rows of zero padding above input row 0 (= pad - band_top: fcp_conv_desc.band_top / band_bottom)
  int M, ktiles, ctiles, wrow;
  float act_slope, alpha, alpha2;
  int res1_pre, res1_ld, res1_h, res1_w, res1_resize, res2_ld;
  float res1_sh, res1_sw;
  int grid_m, grid_n, vec_ok;
  unsigned in_bytes, w_bytes;
  const float* wscale;  // per-cout power-of-two filter scale (fp16x3 path) or nullptr
  int in_fmt, out_fmt, res1_fmt, res2_fmt;   // 0 = fp32 NHWC, 1 = split32 (see fcp_hip.h)
  // second source of a 1x1 conv (channels >= csplit), LDS-DMA kernels only; in2 == nullptr: off
  const float* in2;
  unsigned in2_bytes;
  int csplit, in2_ld, ph2, pw2, stride2;
  int nt_store;   // 1: split32 outputs are written with non-temporal (streaming) stores
  // 256-row kernel only: M-tile schedule (see conv_igemm_f16x3_big).  `balance` / `cu_budget` come from the descriptor,
  // the launcher derives the rest.
  int balance, cu_budget;
  int mfull, tail_rows, round_size;
  int big_tiles;        // 256-row kernel (persistent): tiles of the launch = virtual block ids 0 .. big_tiles - 1
  int big_persist;      // 1: round_size workgroups walk the tiles; 0: one workgroup per tile
};


// K order: slice kt = (channel slice kt / taps, tap kt % taps): the kh*kw taps of one 32-channel
// slice are consecutive, so a 3x3 conv touches only 32 channels of its pixel neighbourhood for 9
// slices in a row — the per-XCD working set stays L2 resident instead of being re-fetched per tap.
//
// Operand addressing per 16-byte piece: `base` = byte offset of the piece at tap (0,0), channel 0
// (wrapping unsigned arithmetic; only used when valid) and `mask` bit t = tap t reads a real pixel
// (not the zero padding).  Per slice the offset is base + tap_off(uniform) + c0*4, or 0xFFFFFFFF.
struct TapPiece {
  unsigned base, mask;
};

template <bool CIN4>
__device__ __forceinline__ TapPiece make_tap_piece(const ConvK& p, unsigned pbase, int hi0, int wi0, unsigned chan_off) {
  TapPiece t;
  t.base = ((pbase + (unsigned)(hi0 * p.pw + wi0)) * (unsigned)p.in_ld + chan_off) * 4u;
  t.mask = 0u;
  const int taps = CIN4 ? p.kh : p.kh * p.kw;
  for (int q = 0; q < taps; ++q) {
    const int kh_i = CIN4 ? q : q / p.kw;
    const int kw_i = CIN4 ? 0 : q - kh_i * p.kw;
    const bool ok = (unsigned)(hi0 + kh_i) < (unsigned)p.in_h && (unsigned)(wi0 + kw_i) < (unsigned)p.in_w;
    t.mask |= ok ? (1u << q) : 0u;
  }
  return t;
}

// Fused epilogue, shared by both MFMA variants.  `acc` is in the 32x32 MFMA C/D layout.
template <int BN, int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x16 (&acc)[TM][TN], float* smem, int tile_m,
                                              int tile_n, int tid, int lane, int wm, int wn, int hw) {
  // ---- epilogue.  The accumulators leave the MFMA layout (col = lane & 31,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5)) through LDS, so that every lane owns
  // 16-byte row-major chunks: residual loads and output stores are then
  // float4-wide and a wave covers whole 512-byte..1-KiB runs of the NHWC row.
  // (The last main-loop iteration ended with a barrier: the A/B slices are dead.)
  constexpr int CPR = BN / 4;             // float4 chunks per tile row
  constexpr int RPP = 256 / CPR;          // rows per pass
  constexpr int PASSES = BM / RPP;
  float* Cs = smem;                       // [BM][BN]
  const int ccol = (tid % CPR) * 4;
  const int crow = tid / CPR;
  const int co = tile_n * BN + ccol;
  const bool vec = p.vec_ok && (co + 3 < p.cout);
  const long m0 = (long)tile_m * BM + crow;

  const bool pre1 = p.res1 != nullptr && !p.res1_resize && vec;
  const bool pre2 = p.res2 != nullptr && vec;
  {
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int row = wm * WTM + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
          Cs[row * BN + wn * WTN + j * 32 + (lane & 31)] = acc[i][j][rr];
        }
  }
  __syncthreads();

  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, ws4 = {1.f, 1.f, 1.f, 1.f};
  if (p.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (co + e < p.cout) bias4[e] = p.bias[co + e];
  }
  if (p.wscale != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (co + e < p.cout) ws4[e] = p.wscale[co + e];
  }
  constexpr int EPC = PASSES < 4 ? PASSES : 4;     // passes per group: bounds the residual registers
#pragma unroll 1
  for (int g = 0; g < PASSES; g += EPC) {
    f32x4 r1v[EPC], r2v[EPC];
    if (pre1) {
#pragma unroll
      for (int i = 0; i < EPC; ++i) {
        const long m = m0 + (long)(g + i) * RPP;
        r1v[i] = m < p.M ? *reinterpret_cast<const f32x4*>(p.res1 + m * p.res1_ld + co) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (pre2) {
#pragma unroll
      for (int i = 0; i < EPC; ++i) {
        const long m = m0 + (long)(g + i) * RPP;
        r2v[i] = m < p.M ? *reinterpret_cast<const f32x4*>(p.res2 + m * p.res2_ld + co) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      const int row = crow + (g + i) * RPP;
      const long m = m0 + (long)(g + i) * RPP;
      if (m >= p.M || co >= p.cout) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * BN + ccol);
      f32x4 r1 = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
      if (pre1) {
        r1 = r1v[i];
      } else if (p.res1 != nullptr) {
        long roff;
        if (p.res1_resize) {
          const int ni = (int)(m / hw);
          const int rem = (int)(m - (long)ni * hw);
          const int ho = rem / p.out_w;
          const int wo = rem - ho * p.out_w;
          int sh = (int)floorf(ho * p.res1_sh);
          int sw = (int)floorf(wo * p.res1_sw);
          sh = sh < p.res1_h - 1 ? sh : p.res1_h - 1;
          sw = sw < p.res1_w - 1 ? sw : p.res1_w - 1;
          roff = (((long)ni * p.res1_h + sh) * p.res1_w + sw) * p.res1_ld + co;
        } else {
          roff = m * p.res1_ld + co;
        }
        if (vec) {
          r1 = *reinterpret_cast<const f32x4*>(p.res1 + roff);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.cout) r1[e] = p.res1[roff + e];
        }
      }
      if (pre2) {
        r2 = r2v[i];
      } else if (p.res2 != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < p.cout) r2[e] = p.res2[m * p.res2_ld + co + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e] * ws4[e] + bias4[e];
        if (p.res1 != nullptr && p.res1_pre) x += r1[e];
        x = x >= 0.f ? x : x * p.act_slope;
        x = x * p.alpha;
        if (p.res1 != nullptr && !p.res1_pre) x += r1[e];
        if (p.res2 != nullptr) x = x * p.alpha2 + r2[e];
        v[e] = x;
      }
      float* dst = p.out + m * p.out_ld + co;
      if (vec) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < p.cout) dst[e] = v[e];
      }
    }
  }
}


typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// Mixed-precision FMA (v_fma_mix_f32: every source is an f32 register or one binary16 half of a register, one rounding):
//   fcp_mix_sum(h, l, HI)  = float(h.half) + float(l.half)      — exactly what v_cvt_f32_f16 x2 + v_add_f32 give (the conversions
//   fcp_mix_diff(x, h, HI) = x - float(h.half)                    are exact, so there is one rounding either way), in ONE instruction.
// The epilogues of every fp16x3 kernel decode / encode 8-16 values per lane and chunk with these; as separate converts and
// adds that was a third of the bottleneck-chain kernel's epilogue instructions.
template <int HI>
__device__ __forceinline__ float fcp_mix_sum(unsigned h, unsigned l) {
  float r;
  const float one = 1.0f;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "s"(one), "v"(l));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "s"(one), "v"(l));
  return r;
}
template <int HI>
__device__ __forceinline__ float fcp_mix_diff(float x, unsigned h) {
  float r;
  const float minus_one = -1.0f;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(minus_one), "v"(x));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(minus_one), "v"(x));
  return r;
}

// 8 fp32 -> 8 hi + 8 lo binary16 (round-toward-zero packs; lo = x - hi is exact in fp32).  Idempotent on
// values that already are a hi + lo sum, so elementwise kernels may decode / re-encode freely.
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4_t& hi, u32x4_t& lo) {
  const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2 * q], x[2 * q + 1]));
    const float r0 = fcp_mix_diff<0>(x[2 * q], hu);
    const float r1 = fcp_mix_diff<1>(x[2 * q + 1], hu);
    hi[q] = hu;
    lo[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
  }
}
// inverse: value = float(hi) + float(lo) (exact)
__device__ __forceinline__ void join8(const u32x4_t& hi, const u32x4_t& lo, float (&x)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned hu = hi[q], lu = lo[q];
    x[2 * q] = fcp_mix_sum<0>(hu, lu);
    x[2 * q + 1] = fcp_mix_sum<1>(hu, lu);
  }
}
// byte offset of channel c (multiple of 8) inside a split32 pixel: group (c/32)*128 B, hi at (c%32)*2, lo +64
__device__ __forceinline__ long split_chan_off(int c) { return (long)(c >> 5) * 128 + (c & 31) * 2; }

// 8 consecutive channels of one pixel from an activation tensor of either format
__device__ __forceinline__ void load8(const float* base, long pix, int ld, int c, int fmt, float (&x)[8]) {
  if (fmt == 1) {
    const char* pb = reinterpret_cast<const char*>(base) + pix * ld * 4 + split_chan_off(c);
    join8(*reinterpret_cast<const u32x4_t*>(pb), *reinterpret_cast<const u32x4_t*>(pb + 64), x);
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(base + pix * ld + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(base + pix * ld + c + 4);
    x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
  }
}

// Residual tile of a workgroup, fetched at kernel start so its HBM latency hides under the main loop
// (the 1x1 "c3" convs of the bottlenecks have 2-8 K slices only: their cost is the epilogue's traffic).
template <int BN>
struct ResPrefetch {
  static constexpr int CPR = BN / 8, RPP = 256 / CPR, PASSES = BM / RPP;
  u32x4_t a[PASSES], b[PASSES];   // split32: hi, lo chunks; fp32: channels c..c+3, c+4..c+7
  bool valid;
};

template <int BN>
__device__ __forceinline__ void prefetch_res1(const ConvK& p, int tile_m, int tile_n, int tid, ResPrefetch<BN>& r) {
  constexpr int CPR = BN / 8, RPP = 256 / CPR, PASSES = BM / RPP;
  const int co = tile_n * BN + (tid % CPR) * 8;
  const long m0 = (long)tile_m * BM + tid / CPR;
  r.valid = p.res1 != nullptr && !p.res1_resize && co < p.cout;
  if (!r.valid) return;
#pragma unroll
  for (int g = 0; g < PASSES; ++g) {
    const long m = m0 + (long)g * RPP;
    if (m < p.M) {
      if (p.res1_fmt == 1) {
        const char* pb = reinterpret_cast<const char*>(p.res1) + m * p.res1_ld * 4 + split_chan_off(co);
        r.a[g] = *reinterpret_cast<const u32x4_t*>(pb);
        r.b[g] = *reinterpret_cast<const u32x4_t*>(pb + 64);
      } else {
        r.a[g] = *reinterpret_cast<const u32x4_t*>(p.res1 + m * p.res1_ld + co);
        r.b[g] = *reinterpret_cast<const u32x4_t*>(p.res1 + m * p.res1_ld + co + 4);
      }
    }
  }
}

template <int I, int N, typename F>
__device__ __forceinline__ void fcp_static_for(F&& f) {          // f(std::integral_constant<int, I>{}) for I .. N - 1
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    fcp_static_for<I + 1, N>(f);
  }
}

// Epilogue straight from the accumulators of a TRANSPOSED tile (the filter fragment was the MFMA's row operand: filters x
// pixels — same products, same K order, same bits as the pixel-major tile): no fp32 tile in LDS, no barrier, no read-back.
// Lane l holds, for pixel (l & 31) of a 32 x 32 tile, filters 8 q + 4 (l >> 5) + 0..3 in accumulator quad q.  Three lane
// permutations per register (v_permlane32_swap on the quads, then v_permlane32_swap + v_permlane16_swap on the results)
// leave lane (p = l & 15, g = l >> 4) with channels 8 g .. 8 g + 7 of pixel p (set A) and of pixel 16 + p (set B): one
// 16-byte hi and one 16-byte lo store per set, four neighbouring lanes filling the 64 bytes of a pixel's hi (lo) half —
// the store pattern and the expressions of the staged epilogue (conv_epilogue8).  cout % 8 == 0, 16-byte aligned tensors.
// Round 3: the staged form took 36 k cycles per 256 x 256 tile of the 256-row kernel with its stores ablated
// (tools/big_epi_ablate.sh, profiles/r03_probes.md section 15).
//   m_start / m_end: rows of the workgroup tile; tm_act: row tiles of this wave that exist (<= TM); co_tile / co_end: channels
template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void conv_epilogue_regs(const ConvK& p, f32x16 (&acc)[TM][TN], int m_start, int m_end, int tm_act, int co_tile,
                                                   int co_end, int wm, int wn, int lane, int hw) {
    const int lp = lane & 15, lg = lane >> 4;
  auto swap32 = [](float& x, float& y) {                          // x.lanes[32:63] <-> y.lanes[0:31]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    x = __uint_as_float(r0); y = __uint_as_float(r1);
  };
  auto swap16 = [](float& x, float& y) {                          // odd 16-lane rows of x <-> even rows of y
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    x = __uint_as_float(r0); y = __uint_as_float(r1);
  };
  // Issue order of the vector memory (round 5).  The first version loaded a set's residual, waited, computed, stored, and
  // went on to the next set: the stores may alias the residual as far as the compiler can tell, so it never moved a load
  // above an earlier store, and vmcnt retires in order — every one of the 2 TM TN sets waited for its own HBM round trip AND
  // for the acknowledgement of the previous set's stores (16 serial round trips per 256 x 256 tile with a residual: the
  // 1x1 expand + identity convs and the FPN merges spent two thirds of their time there).  Now the sets are handled in GROUPS
  // of up to two row tiles of one column tile (<= 4 sets): the residual pieces of the whole group are requested first (8
  // registers per set), THEN the finished previous group is stored, then the group is computed into registers: one exposed
  // round trip per group, and no wait ever sits behind a store.
  constexpr int GI = TM < 2 ? TM : 2;                              // row tiles per group
  constexpr int NGI = (TM + GI - 1) / GI;                          // groups per column tile
  constexpr int NG = TN * NGI;
  u32x4_t pend_a[GI][2], pend_b[GI][2];                            // finished values of the previous group (hi | lo, or 2 x f32x4)
  auto row_of = [&](int i, int set) { return m_start + wm * WTM + i * 32 + 16 * set + lp; };
  auto chan_of = [&](int jj) { return co_tile + wn * WTN + jj * 32 + 8 * lg; };
  auto flush = [&](auto gc) {                                      // stores of group g
    constexpr int g = decltype(gc)::value, jj = g / NGI, i0 = (g % NGI) * GI;
    const int co = chan_of(jj);
    fcp_static_for<0, GI>([&](auto ic) {
      constexpr int i = i0 + decltype(ic)::value;
      if constexpr (i < TM) {
        fcp_static_for<0, 2>([&](auto sc) {
          constexpr int set = decltype(sc)::value;
          const int mi = row_of(i, set);
          if (i < tm_act && mi < m_end && co < co_end) {
            if (p.out_fmt == 1) {
              char* ob = reinterpret_cast<char*>(p.out) + (long)mi * p.out_ld * 4 + split_chan_off(co);
              *reinterpret_cast<u32x4_t*>(ob) = pend_a[i - i0][set];
              *reinterpret_cast<u32x4_t*>(ob + 64) = pend_b[i - i0][set];
            } else {
              float* dst = p.out + (long)mi * p.out_ld + co;
              *reinterpret_cast<u32x4_t*>(dst) = pend_a[i - i0][set];
              *reinterpret_cast<u32x4_t*>(dst + 4) = pend_b[i - i0][set];
            }
          }
        });
      }
    });
  };
  fcp_static_for<0, NG>([&](auto gc) {
    constexpr int g = decltype(gc)::value, j = g / NGI, i0 = (g % NGI) * GI;
    const int co = chan_of(j);                                      // this lane's eight channels in column tile j
    const bool cok = co < co_end;
    const int cc = cok ? co : co_tile;
    // ---- the group's residual pieces (raw 16-byte loads, decoded at use)
    u32x4_t r1a[GI][2], r1b[GI][2];
    if (p.res1 != nullptr) {
      fcp_static_for<0, GI>([&](auto ic) {
        constexpr int i = i0 + decltype(ic)::value;
        if constexpr (i < TM) {
          fcp_static_for<0, 2>([&](auto sc) {
            constexpr int set = decltype(sc)::value;
            const int mi = row_of(i, set);
            const long m = (i < tm_act && mi < m_end && cok) ? (long)mi : (long)m_start;
            long rpix = m;
            if (p.res1_resize) {
              const int ni = (int)(m / hw);
              const int rem = (int)(m - (long)ni * hw);
              const int ho = rem / p.out_w;
              const int wo = rem - ho * p.out_w;
              int sh = (int)floorf(ho * p.res1_sh);
              int sw = (int)floorf(wo * p.res1_sw);
              sh = sh < p.res1_h - 1 ? sh : p.res1_h - 1;
              sw = sw < p.res1_w - 1 ? sw : p.res1_w - 1;
              rpix = ((long)ni * p.res1_h + sh) * p.res1_w + sw;
            }
            if (p.res1_fmt == 1) {
              const char* pb = reinterpret_cast<const char*>(p.res1) + rpix * p.res1_ld * 4 + split_chan_off(cc);
              r1a[i - i0][set] = *reinterpret_cast<const u32x4_t*>(pb);
              r1b[i - i0][set] = *reinterpret_cast<const u32x4_t*>(pb + 64);
            } else {
              r1a[i - i0][set] = *reinterpret_cast<const u32x4_t*>(p.res1 + rpix * p.res1_ld + cc);
              r1b[i - i0][set] = *reinterpret_cast<const u32x4_t*>(p.res1 + rpix * p.res1_ld + cc + 4);
            }
          });
        }
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g > 0) flush(std::integral_constant<int, g - 1>{});   // the previous group's stores, behind this group's loads
    __builtin_amdgcn_sched_barrier(0);
    float bias8[8], ws8[8];
    {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.wscale + cc), w1 = *reinterpret_cast<const f32x4*>(p.wscale + cc + 4);
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (p.bias != nullptr) { b0 = *reinterpret_cast<const f32x4*>(p.bias + cc); b1 = *reinterpret_cast<const f32x4*>(p.bias + cc + 4); }
#pragma unroll
      for (int e = 0; e < 4; ++e) { ws8[e] = w0[e]; ws8[4 + e] = w1[e]; bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    fcp_static_for<0, GI>([&](auto ic) {
      constexpr int i = i0 + decltype(ic)::value;
      if constexpr (i < TM) {
        float va[8], vb[8];                                           // sets A / B after the permutations
        fcp_static_for<0, 4>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          // (named floats: __builtin_bit_cast applied directly to a vector-element expression read element 0)
          float q0 = acc[i][j][e], q1 = acc[i][j][4 + e], q2 = acc[i][j][8 + e], q3 = acc[i][j][12 + e];
          swap32(q0, q1);               // q0: ch e (lanes 0-31) | 8 + e (32-63);   q1: 4 + e | 12 + e           of pixel l & 31
          swap32(q2, q3);               // q2: 16 + e | 24 + e;                     q3: 20 + e | 28 + e
          swap32(q0, q2); swap16(q0, q2);   // q0 = set A: rows of 16 lanes hold ch e, 8 + e, 16 + e, 24 + e of pixels 0-15; q2 = set B (pixels 16-31)
          swap32(q1, q3); swap16(q1, q3);   // the same for ch 4 + e, 12 + e, 20 + e, 28 + e
          va[e] = q0; va[4 + e] = q1; vb[e] = q2; vb[4 + e] = q3;
        });
        fcp_static_for<0, 2>([&](auto sc) {
          constexpr int set = decltype(sc)::value;
          float (&v)[8] = set == 0 ? va : vb;
          float r1[8], r2[8];
          if (p.res1 != nullptr) {
            if (p.res1_fmt == 1) {
              join8(r1a[i - i0][set], r1b[i - i0][set], r1);
            } else {
              const f32x4 fa = __builtin_bit_cast(f32x4, r1a[i - i0][set]), fb = __builtin_bit_cast(f32x4, r1b[i - i0][set]);
              r1[0] = fa[0]; r1[1] = fa[1]; r1[2] = fa[2]; r1[3] = fa[3]; r1[4] = fb[0]; r1[5] = fb[1]; r1[6] = fb[2]; r1[7] = fb[3];
            }
          }
          if (p.res2 != nullptr) {
            const int mi = row_of(i, set);
            load8(p.res2, (i < tm_act && mi < m_end && cok) ? (long)mi : (long)m_start, p.res2_ld, cc, p.res2_fmt, r2);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e] * ws8[e] + bias8[e];
            if (p.res1 != nullptr && p.res1_pre) x += r1[e];
            x = x >= 0.f ? x : x * p.act_slope;
            x = x * p.alpha;
            if (p.res1 != nullptr && !p.res1_pre) x += r1[e];
            if (p.res2 != nullptr) x = x * p.alpha2 + r2[e];
            v[e] = x;
          }
          if (p.out_fmt == 1) {
            split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, pend_a[i - i0][set], pend_b[i - i0][set]);
          } else {
            pend_a[i - i0][set] = __builtin_bit_cast(u32x4_t, f32x4{v[0], v[1], v[2], v[3]});
            pend_b[i - i0][set] = __builtin_bit_cast(u32x4_t, f32x4{v[4], v[5], v[6], v[7]});
          }
        });
      }
    });
  });
  flush(std::integral_constant<int, NG - 1>{});
}

// Epilogue for 8-channel granularity: used whenever the output or a residual is in split32 format
// (cout % 8 == 0, all tensors 16-byte aligned).  Same arithmetic as conv_epilogue.
template <int BN, int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void conv_epilogue8(const ConvK& p, f32x16 (&acc)[TM][TN], float* smem, int tile_m,
                                               int tile_n, int tid, int lane, int wm, int wn, int hw,
                                               const ResPrefetch<BN>* pre = nullptr) {
  constexpr int CPR = BN / 8;
  constexpr int RPP = 256 / CPR;
  constexpr int PASSES = BM / RPP;
  float* Cs = smem;
  const int ccol = (tid % CPR) * 8;
  const int crow = tid / CPR;
  const int co = tile_n * BN + ccol;
  const long m0 = (long)tile_m * BM + crow;
  {
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int row = wm * WTM + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
          Cs[row * BN + wn * WTN + j * 32 + (lane & 31)] = acc[i][j][rr];
        }
  }
  __syncthreads();
  if (co >= p.cout) return;
  float bias8[8], ws8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bias8[e] = p.bias != nullptr ? p.bias[co + e] : 0.f;
    ws8[e] = p.wscale != nullptr ? p.wscale[co + e] : 1.f;
  }
  const bool use_pre = pre != nullptr && pre->valid;
#pragma unroll
  for (int g = 0; g < PASSES; ++g) {
    const int row = crow + g * RPP;
    const long m = m0 + (long)g * RPP;
    if (m >= p.M) continue;
    float v[8], r1[8], r2[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + row * BN + ccol);
      const f32x4 b = *reinterpret_cast<const f32x4*>(Cs + row * BN + ccol + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    if (use_pre) {
      if (p.res1_fmt == 1) {
        join8(pre->a[g], pre->b[g], r1);
      } else {
        const f32x4 fa = __builtin_bit_cast(f32x4, pre->a[g]), fb = __builtin_bit_cast(f32x4, pre->b[g]);
        r1[0] = fa[0]; r1[1] = fa[1]; r1[2] = fa[2]; r1[3] = fa[3]; r1[4] = fb[0]; r1[5] = fb[1]; r1[6] = fb[2]; r1[7] = fb[3];
      }
    } else if (p.res1 != nullptr) {
      long rpix = m;
      if (p.res1_resize) {
        const int ni = (int)(m / hw);
        const int rem = (int)(m - (long)ni * hw);
        const int ho = rem / p.out_w;
        const int wo = rem - ho * p.out_w;
        int sh = (int)floorf(ho * p.res1_sh);
        int sw = (int)floorf(wo * p.res1_sw);
        sh = sh < p.res1_h - 1 ? sh : p.res1_h - 1;
        sw = sw < p.res1_w - 1 ? sw : p.res1_w - 1;
        rpix = ((long)ni * p.res1_h + sh) * p.res1_w + sw;
      }
      load8(p.res1, rpix, p.res1_ld, co, p.res1_fmt, r1);
    }
    if (p.res2 != nullptr) load8(p.res2, m, p.res2_ld, co, p.res2_fmt, r2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] * ws8[e] + bias8[e];
      if (p.res1 != nullptr && p.res1_pre) x += r1[e];
      x = x >= 0.f ? x : x * p.act_slope;
      x = x * p.alpha;
      if (p.res1 != nullptr && !p.res1_pre) x += r1[e];
      if (p.res2 != nullptr) x = x * p.alpha2 + r2[e];
      v[e] = x;
    }
    if (p.out_fmt == 1) {
      u32x4_t hi, lo;
      split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
      char* ob = reinterpret_cast<char*>(p.out) + m * p.out_ld * 4 + split_chan_off(co);
      if (p.nt_store) {
        __builtin_nontemporal_store(hi, reinterpret_cast<u32x4_t*>(ob));
        __builtin_nontemporal_store(lo, reinterpret_cast<u32x4_t*>(ob + 64));
      } else {
        *reinterpret_cast<u32x4_t*>(ob) = hi;
        *reinterpret_cast<u32x4_t*>(ob + 64) = lo;
      }
    } else {
      float* dst = p.out + m * p.out_ld + co;
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

int launch_f16x3(const ConvK& k, int tile_n, bool cin4, hipStream_t s);
int launch_f16x3_dma(const ConvK& k, int tile_n, int stages, hipStream_t s);
int launch_f16x3_halo(const ConvK& k, hipStream_t s);           // 3x3 / stride 1 / pad 1, cout <= 64: 8x32 halo tiles, one pass per 32 filters
int launch_f16x3_halo_wide(const ConvK& k, hipStream_t s);      // same patch, cout <= 128, cin % 64 == 0: column tiles inner, tap ring
int launch_f16x3_big(const ConvK& k, int tile_n, hipStream_t s);   // 256-row tiles, tile_n 128 | 256

}  // namespace fcp_conv
