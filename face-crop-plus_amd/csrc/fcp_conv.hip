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
// LDS image: [row][32] floats, the eight 16-byte chunks of a row XOR-swizzled by
// (row >> 1) & 7.  Fragments are fetched with ds_read_b128: lane l reads 4
// consecutive k of row (l & 31) at k-chunk 2p + (l >> 5); element e of that
// vector feeds MFMA #e of the group, i.e. the hardware's "k index = lane >> 5"
// is mapped to physical k = 8p+e / 8p+4+e identically for A and B (any
// bijection of k is a valid contraction order).  With the swizzle every 16-lane
// service group of ds_read_b128 touches 16 distinct bank slots (conflict free)
// and, unpadded, three 128x64 workgroups fit one CU's 160 KiB of LDS.
#include "fcp_conv_common.h"

#include <cstdlib>

using namespace fcp_conv;

namespace {

constexpr int LDK = 32;  // floats per LDS row (XOR-swizzled 16-byte chunks, no padding)


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
__global__ void __launch_bounds__(256, (BN == 128 ? 2 : (BN == 64 ? 3 : 4))) conv_igemm_f32(const ConvK p) {
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
  long nbase[A_LD];        // flat path only
  unsigned pbase[A_LD];    // buffer path: first pixel of the row's image (32-bit)
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
      pbase[i] = (unsigned)(ni * p.ph * p.pw);
      hi0[i] = ho * p.stride - p.pad_h;
      wi0[i] = wo * p.stride - p.pad;
    } else {
      nbase[i] = 0;
      pbase[i] = 0;
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
  TapPiece tp[A_LD];
  if (BUF) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      tp[i] = make_tap_piece<CIN4>(p, pbase[i], hi0[i], wi0[i] + (CIN4 ? chunk : 0), CIN4 ? 0u : (unsigned)(chunk * 4));
  }
  auto set_tap = [&](int tap, int kh_i, int kw_i) {
    if (p.in_up2) {   // nearest-x2 operand fetch: physical offset is not linear in the tap
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int hi = hi0[i] + kh_i;
        int wi = wi0[i] + (CIN4 ? chunk : kw_i);
        const bool ok = (unsigned)hi < (unsigned)p.in_h && (unsigned)wi < (unsigned)p.in_w;
        hi >>= 1; wi >>= 1;
        const unsigned pix = pbase[i] + (unsigned)(hi * p.pw + wi);
        rowoff[i] = ok ? (pix * (unsigned)p.in_ld + (CIN4 ? 0u : (unsigned)(chunk * 4))) * 4u : 0xFFFFFFFFu;
      }
    } else {
      const unsigned tapoff = (unsigned)((kh_i * p.pw + kw_i) * p.in_ld) * 4u;   // wave-uniform
#pragma unroll
      for (int i = 0; i < A_LD; ++i) rowoff[i] = ((tp[i].mask >> tap) & 1u) ? tp[i].base + tapoff : 0xFFFFFFFFu;
    }
  };

  // two register sets: the fetch of slice k+2 is in flight while slice k is multiplied and
  // slice k+1 (fetched one full iteration earlier) is written to LDS — a prefetch distance of
  // two MFMA phases, enough to cover HBM / Infinity-Cache latency of activations that miss L2.
  f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD];

  auto load_slice = [&](f32x4 (&ra)[A_LD], f32x4 (&rb)[B_LD], int kt, int kh_i, int kw_i, int c0) {
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
  // chunk' = chunk ^ ((row >> 1) & 7): a 16-lane ds_read_b128 service group then touches 16
  // distinct 16-byte bank slots (rows of equal parity get distinct chunks), conflict-free
  // without padding; rows are 128 B so three 128x64 workgroups fit one CU's 160 KiB.
  const int schunk = chunk ^ ((lrow >> 1) & 7);
  auto store_slice = [&](const f32x4 (&ra)[A_LD], const f32x4 (&rb)[B_LD], int buf) {
    float* a = As + buf * BM * LDK + lrow * LDK + schunk * 4;
    float* b = Bs + buf * BN * LDK + lrow * LDK + schunk * 4;
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

  int tap = 0, kh_i = 0, kw_i = 0, c0 = 0;
  auto advance = [&]() {      // next slice: taps fastest, then the 32-channel slice
    ++tap;
    if (CIN4) {
      ++kh_i;
    } else if (++kw_i >= p.kw) {
      kw_i = 0;
      if (++kh_i >= p.kh) { kh_i = 0; tap = 0; c0 += BK; }
    }
    if (BUF) set_tap(tap, kh_i, kw_i);
  };

  const int aoff = (wm * WTM + (lane & 31)) * LDK;
  const int boff = (wn * WTN + (lane & 31)) * LDK;
  const int rsw = ((lane & 31) >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) koff[pp] = (((2 * pp + (lane >> 5)) ^ rsw)) * 4;

  auto compute = [&](int buf) {
    const float* Ab = As + buf * BM * LDK + aoff;
    const float* Bb = Bs + buf * BN * LDK + boff;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + (i * 32 * LDK + koff[pp]));
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + (j * 32 * LDK + koff[pp]));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  };
  // one pipeline step: fetch slice kt+2 into the set that held slice kt, multiply slice kt,
  // park slice kt+1 (other set) in the LDS buffer slice kt-1 just vacated
  auto step = [&](int kt, f32x4 (&ra_ld)[A_LD], f32x4 (&rb_ld)[B_LD], const f32x4 (&ra_st)[A_LD],
                  const f32x4 (&rb_st)[B_LD]) {
    if (kt + 2 < p.ktiles) {
      advance();
      load_slice(ra_ld, rb_ld, kt + 2, kh_i, kw_i, c0);
    }
    compute(kt & 1);
    if (kt + 1 < p.ktiles) store_slice(ra_st, rb_st, (kt + 1) & 1);
    __syncthreads();
  };

  if (BUF) set_tap(0, 0, 0);
  load_slice(ra0, rb0, 0, kh_i, kw_i, c0);
  store_slice(ra0, rb0, 0);
  if (p.ktiles > 1) {
    advance();
    load_slice(ra1, rb1, 1, kh_i, kw_i, c0);
  }
  __syncthreads();
  for (int kt = 0; kt < p.ktiles; kt += 2) {
    step(kt, ra0, rb0, ra1, rb1);
    if (kt + 1 < p.ktiles) step(kt + 1, ra1, rb1, ra0, rb0);
  }

  conv_epilogue<BN, TM, TN, WTM, WTN>(p, acc, smem, tile_m, tile_n, tid, lane, wm, wn, hw);
}

template <int BN, bool CIN4, bool BUF>
int launch(const ConvK& k, hipStream_t s) {
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  FCP_LDS_OPT_IN((&conv_igemm_f32<BN, CIN4, BUF>), lds);
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
  const bool big = d->tile_m == 256, halo = d->tile_m == 1;
  FCP_REQUIRE(d->tile_m == 0 || d->tile_m == 128 || big || halo, "conv: tile_m must be 0/128/256 (or 1: halo-tile 3x3)");
  const bool halo_wide = halo && d->tile_n != 32;      // tile_n 64 / 128: column tiles inner, filters through a tap ring
  if (halo)
    FCP_REQUIRE(d->precision == 1 && d->in_fmt == 1 && !d->cin4 && d->kh == 3 && d->kw == 3 && d->stride == 1 &&
                d->pad == 1 && d->cout <= 64 && d->cout % 8 == 0 && d->cin >= 64 && !d->in2 &&
                (!d->res1 || (d->res1_h == d->out_h && d->res1_w == d->out_w)) || halo_wide,
                "conv: the halo-tile kernel needs a 3x3 / stride 1 / pad 1 conv with cin >= 64, cout <= 64 (cout %% 8 == 0) on the "
                "fp16x3 path with a split32 input, no second source, no resized residual");
  if (halo_wide)
    FCP_REQUIRE(d->precision == 1 && d->in_fmt == 1 && !d->cin4 && d->kh == 3 && d->kw == 3 && d->stride == 1 &&
                d->pad == 1 && d->cout % 8 == 0 && d->cin >= 64 && d->cin % 64 == 0 && !d->in2 &&
                ((d->tile_n == 64 && d->cout <= 64 && (!d->res1 || (d->res1_h == d->out_h && d->res1_w == d->out_w))) ||
                 (d->tile_n == 128 && d->cout <= 128 && !d->res1 && !d->res2)),
                "conv: the wide halo-tile kernel (tile_m 1, tile_n 64 / 128) needs a 3x3 / stride 1 / pad 1 conv with cin %% 64 == 0, "
                "cout <= tile_n (cout %% 8 == 0; no residuals above 64 filters) on the fp16x3 path with a split32 input");
  FCP_REQUIRE(d->tile_n == 32 || d->tile_n == 64 || d->tile_n == 128 || (big && (d->tile_n == 256 || d->tile_n == 192)),
              "conv: tile_n must be 32/64/128 (or 192 / 256 with tile_m 256)");
  if (big)
    FCP_REQUIRE(d->precision == 1 && d->in_fmt == 1 && !d->cin4 && !d->in_up2 && d->cout % 8 == 0 && d->tile_n >= 128,
                "conv: 256-row tiles need the fp16x3 path on a split32 input (no cin4 / in_up2), cout %% 8 == 0, tile_n 128/256");
  FCP_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->pad >= 0, "conv: bad filter geometry");
  FCP_REQUIRE(d->kh * d->kw <= 32 || d->cin4, "conv: at most 32 filter taps (kh*kw) outside cin4 mode");
  FCP_REQUIRE(d->in_ld % 4 == 0 && ((uintptr_t)d->in & 15) == 0, "conv: input must be 16-byte aligned, in_ld %% 4 == 0");
  FCP_REQUIRE(((uintptr_t)d->w & 15) == 0, "conv: filter must be 16-byte aligned");
  if (d->cin4) {
    FCP_REQUIRE(d->cin <= 4 && d->in_ld == 4 && d->kw <= 8, "conv: cin4 mode needs cin<=4, in_ld==4, kw<=8");
  } else {
    FCP_REQUIRE(d->cin % 32 == 0 && d->cin > 0, "conv: cin must be a multiple of 32 (got %d)", d->cin);
  }
  FCP_REQUIRE(d->band_top >= 0 && d->band_bottom >= 0 && d->band_top <= d->pad && d->band_bottom <= d->pad &&
              ((d->band_top | d->band_bottom) == 0 || (d->stride == 1 && !d->in2)),
              "conv: band_top / band_bottom must lie in 0..pad (stride-1 convs without a second source only)");
  const int eh = (d->in_h - d->band_top - d->band_bottom + 2 * d->pad - d->kh) / d->stride + 1;
  const int ew = (d->in_w + 2 * d->pad - d->kw) / d->stride + 1;
  FCP_REQUIRE(eh == d->out_h && ew == d->out_w, "conv: output size %dx%d does not match geometry %dx%d",
              d->out_h, d->out_w, eh, ew);
  if (d->in_up2) FCP_REQUIRE(d->in_h % 2 == 0 && d->in_w % 2 == 0, "conv: in_up2 needs even logical size");
  const long M = (long)d->n * d->out_h * d->out_w;
  FCP_REQUIRE(M < (1L << 31), "conv: too many output pixels");

  ConvK k;
  k.in = d->in; k.w = reinterpret_cast<const float*>(d->w); k.bias = d->bias; k.wscale = d->wscale; k.out = d->out; k.res1 = d->res1; k.res2 = d->res2;
  k.n = d->n; k.in_h = d->in_h; k.in_w = d->in_w;
  k.ph = d->in_up2 ? d->in_h / 2 : d->in_h;
  k.pw = d->in_up2 ? d->in_w / 2 : d->in_w;
  k.cin = d->cin; k.in_ld = d->in_ld; k.in_up2 = d->in_up2;
  k.cout = d->cout; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad; k.pad_h = d->pad - d->band_top;
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
  k.in_fmt = d->in_fmt; k.out_fmt = d->out_fmt; k.res1_fmt = d->res1 ? d->res1_fmt : 0; k.res2_fmt = d->res2 ? d->res2_fmt : 0;
  const int any_split = k.in_fmt | k.out_fmt | k.res1_fmt | k.res2_fmt;
  FCP_REQUIRE((unsigned)any_split <= 1u, "conv: tensor formats must be 0 (fp32) or 1 (split32)");
  if (any_split) {
    FCP_REQUIRE(d->precision == 1, "conv: split32 tensors need precision 1 (fp16x3 kernel)");
    FCP_REQUIRE(d->cout % 8 == 0, "conv: split32 epilogue needs cout %% 8 == 0");
    FCP_REQUIRE(!k.in_fmt || (!d->cin4 && d->in_ld % 32 == 0 && ((uintptr_t)d->in & 127) == 0),
                "conv: split32 input must be a 32-channel-group aligned view (in_ld %% 32 == 0), not cin4");
    FCP_REQUIRE(!k.out_fmt || (d->out_ld % 32 == 0 && ((uintptr_t)d->out & 127) == 0 && d->cout % 32 == 0),
                "conv: split32 output must be a 32-channel-group aligned view with cout %% 32 == 0");
    FCP_REQUIRE(!k.res1_fmt || (d->res1_ld % 32 == 0 && ((uintptr_t)d->res1 & 127) == 0), "conv: misaligned split32 res1");
    FCP_REQUIRE(!k.res2_fmt || (d->res2_ld % 32 == 0 && ((uintptr_t)d->res2 & 127) == 0), "conv: misaligned split32 res2");
    FCP_REQUIRE(((uintptr_t)d->out & 15) == 0 && d->out_ld % 4 == 0 && (!d->res1 || (((uintptr_t)d->res1 & 15) == 0 && d->res1_ld % 4 == 0)) &&
                (!d->res2 || (((uintptr_t)d->res2 & 15) == 0 && d->res2_ld % 4 == 0)), "conv: 8-channel epilogue needs 16-byte aligned tensors");
  }
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
  static const int nt_env = getenv("FCP_NT_STORE") ? atoi(getenv("FCP_NT_STORE")) : 1;
  k.nt_store = nt_env;
  k.balance = (d->flags & FCP_CONV_BALANCE_TAIL) ? 1 : 0;
  k.cu_budget = d->cu_budget;
  k.mfull = 0; k.tail_rows = 0; k.round_size = 0;
  k.in2 = nullptr; k.in2_bytes = 0; k.csplit = d->cin; k.in2_ld = 0; k.ph2 = 0; k.pw2 = 0; k.stride2 = 1;
  if (d->in2) {
    FCP_REQUIRE(d->precision == 1 && d->in_fmt == 1 && !d->cin4 && !d->in_up2 && d->kh == 1 && d->kw == 1 && d->pad == 0,
                "conv: a second source needs a 1x1 / pad 0 conv on the fp16x3 path with split32 inputs");
    FCP_REQUIRE(d->cin2 > 0 && d->cin2 < d->cin && d->cin2 % 32 == 0 && d->in2_ld % 32 == 0 && d->in2_ld >= d->cin2 &&
                ((uintptr_t)d->in2 & 127) == 0, "conv: in2 must be a 32-channel-group aligned split32 view, 0 < cin2 < cin");
    FCP_REQUIRE(d->in2_stride >= 1 && d->in2_h > 0 && d->in2_w > 0 && (long)(d->out_h - 1) * d->in2_stride < d->in2_h &&
                (long)(d->out_w - 1) * d->in2_stride < d->in2_w, "conv: in2 geometry does not cover the output grid");
    const unsigned long in2_bytes = (unsigned long)d->n * d->in2_h * d->in2_w * d->in2_ld * 4ul;
    FCP_REQUIRE(in2_bytes < 0xFFFFFFF0ul, "conv: in2 must be below 4 GiB");
    k.in2 = d->in2; k.in2_bytes = (unsigned)in2_bytes; k.csplit = d->cin - d->cin2; k.in2_ld = d->in2_ld;
    k.ph2 = d->in2_h; k.pw2 = d->in2_w; k.stride2 = d->in2_stride;
  }
  k.grid_m = fcp_cdiv(M, BM);
  k.grid_n = fcp_cdiv(d->cout, d->tile_n);
  hipStream_t s = (hipStream_t)stream;
  // buffer-load path needs every operand addressable with 32-bit byte offsets
  const unsigned long in_bytes = (unsigned long)d->n * k.ph * k.pw * d->in_ld * 4ul;
  const unsigned long w_bytes = (unsigned long)(k.grid_n * d->tile_n) * k.wrow * 4ul;
  FCP_REQUIRE(!(d->flags & FCP_CONV_FLAT_ADDR) || d->precision == 0, "conv: FCP_CONV_FLAT_ADDR is a precision-0 (fp32 kernel) option");
  const bool buf = in_bytes < 0xFFFFFFF0ul && w_bytes < 0xFFFFFFF0ul && !(d->flags & FCP_CONV_FLAT_ADDR);
  FCP_REQUIRE(d->precision == 0 || d->precision == 1, "conv: precision must be 0 (fp32) or 1 (fp16x3)");
  if (d->precision == 1) {
    FCP_REQUIRE(buf, "conv: the fp16x3 path needs tensors below 4 GiB (pack this layer with precision 0)");
    FCP_REQUIRE(d->wscale != nullptr, "conv: precision 1 needs wscale");
    k.in_bytes = (unsigned)in_bytes;
    k.w_bytes = (unsigned)w_bytes;
    // split32 activations: both operands are pure byte copies -> LDS-DMA kernel
    // two LDS stages (three were measured slower: they cost an occupancy step); profiling builds may override
    const int dma_stages = 2;
    if (halo) {
      k.grid_n = 1;
      const unsigned long out_bytes = (unsigned long)M * d->out_ld * 4ul;
      FCP_REQUIRE(out_bytes < 0xFFFFFFF0ul, "conv(halo): the output view must span less than 4 GiB");
      FCP_REQUIRE(d->wscale != nullptr && ((uintptr_t)d->wscale & 15) == 0 && ((uintptr_t)d->bias & 15) == 0,
                  "conv(halo): wscale / bias must be 16-byte aligned");
      k.in2_bytes = (unsigned)out_bytes;                // halo launches take no second source: the field carries |out|
      k.w_bytes = (unsigned)((unsigned long)fcp_cdiv(d->cout, 128) * 128ul * k.wrow * 4ul);
      return halo_wide ? launch_f16x3_halo_wide(k, s) : launch_f16x3_halo(k, s);
    }
    if (big) {
      k.w_bytes = (unsigned)((unsigned long)fcp_cdiv(d->cout, 128) * 128ul * k.wrow * 4ul);   // filters are padded to 128 rows
      return launch_f16x3_big(k, d->tile_n, s);
    }
    if (k.in_fmt == 1 && !d->cin4) return launch_f16x3_dma(k, d->tile_n, dma_stages, s);
    return launch_f16x3(k, d->tile_n, d->cin4 != 0, s);
  }
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
