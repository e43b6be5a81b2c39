// NHWC fp32 implicit-GEMM convolution on the gfx950 matrix cores.
//
// GEMM view: M = n*out_h*out_w output pixels, N = cout, K = kh*kw*cin.  One
// workgroup (4 wave64) owns a 128 x BN output tile and walks K in 32-float
// slices; each slice is one filter tap x 32 input channels (NHWC makes that a
// contiguous 128-byte run per pixel), staged global -> VGPR -> LDS with the
// next slice's loads in flight under the current slice's MFMAs (two LDS
// buffers, one barrier per slice).  The matrix instruction is
// v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit an fmaf chain, so
// results differ from the reference's ATen conv only by summation order.
//
// LDS image: [row][36] floats (32 + 4 pad -> 144-byte rows).  Fragments are
// fetched with ds_read_b128: lane l reads 4 consecutive k of row (l & 31) at
// k-chunk 2p + (l >> 5); element e of that vector feeds MFMA #e of the group,
// i.e. the hardware's "k index = lane >> 5" is mapped to physical k = 8p+e /
// 8p+4+e identically for A and B (any bijection of k is a valid contraction
// order).  9*r mod 16 is a bijection on every 16-lane service group of
// ds_read_b128, so the reads are bank-conflict free.
#include "fcp_common.h"
#include "fcp_hip.h"

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = 36;  // floats per LDS row

struct ConvK {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  const float* res1;
  const float* res2;
  int n, in_h, in_w, ph, pw, cin, in_ld, in_up2;
  int cout, kh, kw, stride, pad, out_h, out_w, out_ld;
  int M, ktiles, ctiles, wrow;
  float act_slope, alpha, alpha2;
  int res1_pre, res1_ld, res1_h, res1_w, res1_resize, res2_ld;
  float res1_sh, res1_sw;
  int grid_m, grid_n, vec_ok;
  unsigned in_bytes, w_bytes;
  int ablate;  // profiling-only knob (env FCP_CONV_ABLATE), 0 in production
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}

// BUF: operands are fetched with raw buffer loads (32-bit byte offsets from a
// wave-uniform descriptor; an offset of 0xFFFFFFFF is out of range and returns
// zeros, which is exactly the conv's zero padding) — branch-free, so the
// compiler can interleave the fetches with the MFMA stream.  Tensors of 4 GiB or
// more fall back to flat 64-bit addressing (BUF = false).
template <int BN, bool CIN4, bool BUF>
__global__ void __launch_bounds__(256) conv_igemm_f32(const ConvK p) {
  constexpr int WAVES_N = (BN == 32) ? 1 : 2;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WTM = BM / WAVES_M;
  constexpr int WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_LD = BM / 32;  // float4 loads per thread per slice
  constexpr int B_LD = BN / 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDK;

  // XCD-aware tile order: hardware round-robins consecutive workgroup ids over
  // the 8 XCDs; give each XCD a contiguous run of logical tiles so the N-tiles
  // of one pixel block and neighbouring pixel blocks (3x3 halos) share an L2.
  const int nb = gridDim.x;
  const int bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tile_n = logical % p.grid_n;
  const int tile_m = logical / p.grid_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int chunk = tid & 7;
  const int lrow = tid >> 3;  // 0..31

  // per-thread im2col row bookkeeping
  long nbase[A_LD];
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
      nbase[i] = (long)ni * p.ph * p.pw;
      hi0[i] = ho * p.stride - p.pad;
      wi0[i] = wo * p.stride - p.pad;
    } else {
      nbase[i] = 0;
      hi0[i] = -(1 << 28);
      wi0[i] = 0;
    }
  }
  const float* wbase = p.w + (long)(tile_n * BN + lrow) * p.wrow + chunk * 4;

  __amdgpu_buffer_rsrc_t rs_in, rs_w;
  unsigned rowoff[A_LD];   // byte offset of this thread's chunk at channel 0 of the current tap
  unsigned woff[B_LD];
  if (BUF) {
    rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      woff[i] = (unsigned)(((tile_n * BN + lrow + 32 * i) * p.wrow + chunk * 4) * 4);
  }
  auto set_tap = [&](int kh_i, int kw_i) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      int hi = hi0[i] + kh_i;
      int wi = wi0[i] + (CIN4 ? chunk : kw_i);
      const bool ok = (unsigned)hi < (unsigned)p.in_h && (unsigned)wi < (unsigned)p.in_w;
      if (p.in_up2) { hi >>= 1; wi >>= 1; }
      const unsigned pix = (unsigned)nbase[i] + (unsigned)(hi * p.pw + wi);
      rowoff[i] = ok ? (pix * (unsigned)p.in_ld + (CIN4 ? 0u : (unsigned)(chunk * 4))) * 4u : 0xFFFFFFFFu;
    }
  };

  f32x4 ra[A_LD], rb[B_LD];

  auto load_slice = [&](int kt, int kh_i, int kw_i, int c0) {
    if (BUF) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const unsigned vo = rowoff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : rowoff[i] + (unsigned)(c0 * 4);
        ra[i] = buf_load16(rs_in, vo);
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) rb[i] = buf_load16(rs_w, woff[i] + (unsigned)(kt * BK * 4));
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int hi = hi0[i] + kh_i;
        int wi = wi0[i] + (CIN4 ? chunk : kw_i);
        const bool ok = (unsigned)hi < (unsigned)p.in_h && (unsigned)wi < (unsigned)p.in_w;
        if (p.in_up2) { hi >>= 1; wi >>= 1; }
        const long pix = nbase[i] + (long)hi * p.pw + wi;
        const float* src = CIN4 ? p.in + pix * p.in_ld : p.in + pix * p.in_ld + c0 + chunk * 4;
        ra[i] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        rb[i] = *reinterpret_cast<const f32x4*>(wbase + (long)(32 * i) * p.wrow + kt * BK);
    }
  };
  auto store_slice = [&](int buf) {
    float* a = As + buf * BM * LDK + lrow * LDK + chunk * 4;
    float* b = Bs + buf * BN * LDK + lrow * LDK + chunk * 4;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(a + 32 * i * LDK) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) *reinterpret_cast<f32x4*>(b + 32 * i * LDK) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int kh_i = 0, kw_i = 0, c0 = 0;
  auto advance = [&]() {
    if (CIN4) {
      ++kh_i;
      if (BUF) set_tap(kh_i, 0);
    } else {
      c0 += BK;
      if (c0 >= p.cin) {
        c0 = 0;
        if (++kw_i >= p.kw) { kw_i = 0; ++kh_i; }
        if (BUF) set_tap(kh_i, kw_i);
      }
    }
  };

  if (BUF) set_tap(0, 0);
  load_slice(0, kh_i, kw_i, c0);
  store_slice(0);
  __syncthreads();

  const int aoff = (wm * WTM + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int boff = (wn * WTN + (lane & 31)) * LDK + (lane >> 5) * 4;

  for (int kt = 0; kt < p.ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < p.ktiles;
    if (more && !(p.ablate & 1)) {
      advance();
      load_slice(kt + 1, kh_i, kw_i, c0);
    }
    const float* Ab = As + buf * BM * LDK + aoff;
    const float* Bb = Bs + buf * BN * LDK + boff;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + ((p.ablate & 4) ? 0 : i * 32 * LDK + pp * 8));
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + ((p.ablate & 4) ? 0 : j * 32 * LDK + pp * 8));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    if (more && !(p.ablate & 2)) store_slice(buf ^ 1);
    if (!(p.ablate & 8)) __syncthreads();
  }

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

  f32x4 r1v[PASSES], r2v[PASSES];
  const bool pre1 = p.res1 != nullptr && !p.res1_resize && vec;
  const bool pre2 = p.res2 != nullptr && vec;
  if (pre1) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const long m = m0 + (long)i * RPP;
      r1v[i] = m < p.M ? *reinterpret_cast<const f32x4*>(p.res1 + m * p.res1_ld + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (pre2) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const long m = m0 + (long)i * RPP;
      r2v[i] = m < p.M ? *reinterpret_cast<const f32x4*>(p.res2 + m * p.res2_ld + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
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

  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (co + e < p.cout) bias4[e] = p.bias[co + e];
  }
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    const int row = crow + i * RPP;
    const long m = m0 + (long)i * RPP;
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
      float x = v[e] + bias4[e];
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

template <int BN, bool CIN4, bool BUF>
int launch(const ConvK& k, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  if (!attr_set) {
    FCP_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BN, CIN4, BUF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int blocks = k.grid_m * k.grid_n;
  hipLaunchKernelGGL((conv_igemm_f32<BN, CIN4, BUF>), dim3(blocks), dim3(256), lds, s, k);
  FCP_LAUNCH_OK();
  return 0;
}

}  // namespace

extern "C" int fcp_conv2d_nhwc_f32(const fcp_conv_desc* d, fcp_stream_t stream) {
  FCP_REQUIRE(d != nullptr, "conv: null descriptor");
  FCP_REQUIRE(d->in && d->w && d->out, "conv: null tensor pointer");
  FCP_REQUIRE(d->n > 0 && d->in_h > 0 && d->in_w > 0 && d->cout > 0, "conv: bad sizes");
  FCP_REQUIRE(d->tile_n == 32 || d->tile_n == 64 || d->tile_n == 128, "conv: tile_n must be 32/64/128");
  FCP_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->pad >= 0, "conv: bad filter geometry");
  FCP_REQUIRE(d->in_ld % 4 == 0 && ((uintptr_t)d->in & 15) == 0, "conv: input must be 16-byte aligned, in_ld %% 4 == 0");
  FCP_REQUIRE(((uintptr_t)d->w & 15) == 0, "conv: filter must be 16-byte aligned");
  if (d->cin4) {
    FCP_REQUIRE(d->cin <= 4 && d->in_ld == 4 && d->kw <= 8, "conv: cin4 mode needs cin<=4, in_ld==4, kw<=8");
  } else {
    FCP_REQUIRE(d->cin % 32 == 0 && d->cin > 0, "conv: cin must be a multiple of 32 (got %d)", d->cin);
  }
  const int eh = (d->in_h + 2 * d->pad - d->kh) / d->stride + 1;
  const int ew = (d->in_w + 2 * d->pad - d->kw) / d->stride + 1;
  FCP_REQUIRE(eh == d->out_h && ew == d->out_w, "conv: output size %dx%d does not match geometry %dx%d",
              d->out_h, d->out_w, eh, ew);
  if (d->in_up2) FCP_REQUIRE(d->in_h % 2 == 0 && d->in_w % 2 == 0, "conv: in_up2 needs even logical size");
  const long M = (long)d->n * d->out_h * d->out_w;
  FCP_REQUIRE(M < (1L << 31), "conv: too many output pixels");

  ConvK k;
  k.in = d->in; k.w = d->w; k.bias = d->bias; k.out = d->out; k.res1 = d->res1; k.res2 = d->res2;
  k.n = d->n; k.in_h = d->in_h; k.in_w = d->in_w;
  k.ph = d->in_up2 ? d->in_h / 2 : d->in_h;
  k.pw = d->in_up2 ? d->in_w / 2 : d->in_w;
  k.cin = d->cin; k.in_ld = d->in_ld; k.in_up2 = d->in_up2;
  k.cout = d->cout; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
  k.out_h = d->out_h; k.out_w = d->out_w; k.out_ld = d->out_ld;
  k.M = (int)M;
  if (d->cin4) {
    k.ktiles = d->kh; k.ctiles = 1; k.wrow = d->kh * 32;
  } else {
    k.ctiles = d->cin / 32; k.ktiles = d->kh * d->kw * k.ctiles; k.wrow = d->kh * d->kw * d->cin;
  }
  k.act_slope = d->act_slope; k.alpha = d->alpha; k.alpha2 = d->alpha2;
  k.res1_pre = d->res1_pre; k.res1_ld = d->res1_ld; k.res2_ld = d->res2_ld;
  k.res1_h = d->res1_h; k.res1_w = d->res1_w;
  k.res1_resize = 0; k.res1_sh = 1.f; k.res1_sw = 1.f;
  if (d->res1) {
    FCP_REQUIRE(d->res1_h > 0 && d->res1_w > 0 && d->res1_ld > 0, "conv: res1 geometry missing");
    if (d->res1_h != d->out_h || d->res1_w != d->out_w) {
      k.res1_resize = 1;
      k.res1_sh = (float)d->res1_h / (float)d->out_h;
      k.res1_sw = (float)d->res1_w / (float)d->out_w;
    }
  }
  if (d->res2) FCP_REQUIRE(d->res2_ld > 0, "conv: res2_ld missing");
  // float4 epilogue needs 16-byte aligned rows in every tensor it touches
  k.vec_ok = (d->out_ld % 4 == 0) && (((uintptr_t)d->out & 15) == 0);
  if (d->res1) k.vec_ok = k.vec_ok && (d->res1_ld % 4 == 0) && (((uintptr_t)d->res1 & 15) == 0);
  if (d->res2) k.vec_ok = k.vec_ok && (d->res2_ld % 4 == 0) && (((uintptr_t)d->res2 & 15) == 0);
  static const int ablate_env = getenv("FCP_CONV_ABLATE") ? atoi(getenv("FCP_CONV_ABLATE")) : 0;
  k.ablate = ablate_env;
  k.grid_m = fcp_cdiv(M, BM);
  k.grid_n = fcp_cdiv(d->cout, d->tile_n);
  hipStream_t s = (hipStream_t)stream;
  // buffer-load path needs every operand addressable with 32-bit byte offsets
  const unsigned long in_bytes = (unsigned long)d->n * k.ph * k.pw * d->in_ld * 4ul;
  const unsigned long w_bytes = (unsigned long)(k.grid_n * d->tile_n) * k.wrow * 4ul;
  const bool buf = in_bytes < 0xFFFFFFF0ul && w_bytes < 0xFFFFFFF0ul;
  k.in_bytes = buf ? (unsigned)in_bytes : 0u;
  k.w_bytes = buf ? (unsigned)w_bytes : 0u;
#define FCP_DISPATCH(BN_)                                                             \
  do {                                                                                \
    if (d->cin4) return buf ? launch<BN_, true, true>(k, s) : launch<BN_, true, false>(k, s); \
    return buf ? launch<BN_, false, true>(k, s) : launch<BN_, false, false>(k, s);   \
  } while (0)
  switch (d->tile_n) {
    case 32: FCP_DISPATCH(32);
    case 64: FCP_DISPATCH(64);
    default: FCP_DISPATCH(128);
  }
#undef FCP_DISPATCH
}
