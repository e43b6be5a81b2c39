// HBM-bound glue kernels of the detector body: uint8 -> fp32 NHWC4 conversion
// (mean subtraction / scaling fused) and the stem max-pool.  One 16-byte store
// per lane, grid-stride, >= 2048 workgroups when the tensor is large enough.
#include "fcp_conv_common.h"

#include <cstdarg>
#include <cstring>

static thread_local char g_err[512] = "";

void fcp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fcp_last_error(void) { return g_err; }
extern "C" int fcp_abi_version(void) { return FCP_ABI_VERSION; }

namespace {

using fcp_conv::join8;
using fcp_conv::split8;
using fcp_conv::split_chan_off;
using fcp_conv::u32x4_t;

// 4 pixels (12 bytes in, 64 bytes out) per thread iteration.
__global__ void __launch_bounds__(256) u8_to_nhwc4_kernel(const uint8_t* __restrict__ in,
                                                          float* __restrict__ out, long npix,
                                                          float s0, float s1, float s2, float div,
                                                          int do_div) {
  const long nquad = npix >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + q * 12);
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    const uint8_t b[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24),
                           (uint8_t)w1, (uint8_t)(w1 >> 8), (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24),
                           (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
    f32x4* dst = reinterpret_cast<f32x4*>(out + q * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float r = (float)b[3 * k] - s0, g = (float)b[3 * k + 1] - s1, bl = (float)b[3 * k + 2] - s2;
      if (do_div) { r = r / div; g = g / div; bl = bl / div; }
      dst[k] = f32x4{r, g, bl, 0.f};
    }
  }
  // tail (npix % 4 pixels), handled by the first threads of block 0
  const long tail0 = nquad << 2;
  if (blockIdx.x == 0 && threadIdx.x < (npix - tail0)) {
    const long px = tail0 + threadIdx.x;
    float r = (float)in[px * 3] - s0, g = (float)in[px * 3 + 1] - s1, bl = (float)in[px * 3 + 2] - s2;
    if (do_div) { r = r / div; g = g / div; bl = bl / div; }
    *reinterpret_cast<f32x4*>(out + px * 4) = f32x4{r, g, bl, 0.f};
  }
}

// fp32 NCHW (3 planes) -> fp32 NHWC4, same affine as above (reference-API entry:
// RetinaFace.predict / RRDBNet.predict receive float NCHW tensors).
__global__ void __launch_bounds__(256) f32nchw_to_nhwc4_kernel(const float* __restrict__ in,
                                                               float* __restrict__ out, int n, long hw,
                                                               float s0, float s1, float s2, float div,
                                                               int do_div) {
  const long total = (long)n * hw;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long ni = i / hw, p = i - ni * hw;
    const float* src = in + ni * 3 * hw + p;
    float r = src[0] - s0, g = src[hw] - s1, b = src[2 * hw] - s2;
    if (do_div) { r = r / div; g = g / div; b = b / div; }
    *reinterpret_cast<f32x4*>(out + i * 4) = f32x4{r, g, b, 0.f};
  }
}

__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int n, int h,
                                                           int w, int c4, int out_ld4, int oh, int ow) {
  const long total = (long)n * oh * ow * c4;
  const long stride = (long)gridDim.x * blockDim.x;
  const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
  f32x4* out4 = reinterpret_cast<f32x4*>(out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cc = (int)(i % c4);
    long t = i / c4;
    const int x = (int)(t % ow);
    t /= ow;
    const int y = (int)(t % oh);
    const int ni = (int)(t / oh);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = 2 * y - 1 + dy;
      if ((unsigned)yy >= (unsigned)h) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * x - 1 + dx;
        if ((unsigned)xx >= (unsigned)w) continue;
        const f32x4 v = in4[(((long)ni * h + yy) * w + xx) * c4 + cc];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    out4[(i / c4) * out_ld4 + cc] = m;
  }
}

// split32 tensors: one thread per (pixel, 8-channel chunk)
__global__ void __launch_bounds__(256) maxpool3x3s2_split_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int n, int h, int w, int c, int out_ld, int oh, int ow) {
  const int c8 = c >> 3;
  const long total = (long)n * oh * ow * c8;
  const long stride = (long)gridDim.x * blockDim.x;
  const char* ib = reinterpret_cast<const char*>(in);
  char* ob = reinterpret_cast<char*>(out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % c8) * 8;
    long t = i / c8;
    const int x = (int)(t % ow);
    t /= ow;
    const int y = (int)(t % oh);
    const int ni = (int)(t / oh);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = 2 * y - 1 + dy;
      if ((unsigned)yy >= (unsigned)h) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * x - 1 + dx;
        if ((unsigned)xx >= (unsigned)w) continue;
        const char* pb = ib + (((long)ni * h + yy) * w + xx) * c * 4 + split_chan_off(ch);
        float v[8];
        join8(*reinterpret_cast<const u32x4_t*>(pb), *reinterpret_cast<const u32x4_t*>(pb + 64), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    u32x4_t hi, lo;
    split8(f32x4{m[0], m[1], m[2], m[3]}, f32x4{m[4], m[5], m[6], m[7]}, hi, lo);
    char* qb = ob + (((long)ni * oh + y) * ow + x) * out_ld * 4 + split_chan_off(ch);
    *reinterpret_cast<u32x4_t*>(qb) = hi;
    *reinterpret_cast<u32x4_t*>(qb + 64) = lo;
  }
}

template <bool TO_SPLIT>
__global__ void __launch_bounds__(256) split_convert_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long npix, int c) {
  const int c8 = c >> 3;
  const long total = npix * c8;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % c8) * 8;
    const long pix = i / c8;
    if (TO_SPLIT) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(in + pix * c + ch);
      const f32x4 b = *reinterpret_cast<const f32x4*>(in + pix * c + ch + 4);
      u32x4_t hi, lo;
      split8(a, b, hi, lo);
      char* qb = reinterpret_cast<char*>(out) + pix * c * 4 + split_chan_off(ch);
      *reinterpret_cast<u32x4_t*>(qb) = hi;
      *reinterpret_cast<u32x4_t*>(qb + 64) = lo;
    } else {
      const char* pb = reinterpret_cast<const char*>(in) + pix * c * 4 + split_chan_off(ch);
      float v[8];
      join8(*reinterpret_cast<const u32x4_t*>(pb), *reinterpret_cast<const u32x4_t*>(pb + 64), v);
      *reinterpret_cast<f32x4*>(out + pix * c + ch) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(out + pix * c + ch + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

inline int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// max |x| over a channel-slice view (range guard of the fp16x3 path): one lane per 8 channels of a pixel, wave
// reduction, one atomicMax per wave on the bit pattern (non-negative floats order like their bits); NaN counts as +inf.
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long npix, int c, int ld, int fmt,
                                                     unsigned* __restrict__ out) {
  const int c8 = c >> 3;
  const long total = npix * c8;
  const long stride = (long)gridDim.x * blockDim.x;
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long px = i / c8;
    const int ch = (int)(i - px * c8) << 3;
    float v[8];
    if (fmt == 1) {
      const char* base = reinterpret_cast<const char*>(x + px * ld) + split_chan_off(ch);
      join8(*reinterpret_cast<const u32x4_t*>(base), *reinterpret_cast<const u32x4_t*>(base + 64), v);
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + px * ld + ch), b = *reinterpret_cast<const f32x4*>(x + px * ld + ch + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float a = v[k] != v[k] ? __builtin_inff() : __builtin_fabsf(v[k]);
      m = a > m ? a : m;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __builtin_bit_cast(unsigned, m));
}

}  // namespace

extern "C" int fcp_absmax_nhwc(const float* x, int64_t npix, int c, int ld, int fmt, float* out_max, fcp_stream_t stream) {
  FCP_REQUIRE(x && out_max && npix > 0 && c > 0 && c % 8 == 0 && ld >= c && ld % 4 == 0 && ((uintptr_t)x & 15) == 0,
              "absmax: bad arguments (c %% 8 == 0, 16-byte aligned view)");
  FCP_REQUIRE(fmt == 0 || (fmt == 1 && c % 32 == 0 && ld % 32 == 0 && ((uintptr_t)x & 127) == 0),
              "absmax: split32 views are 32-channel aligned");
  hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(npix * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (long)npix,
                     c, ld, fmt, reinterpret_cast<unsigned*>(out_max));
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_u8_to_nhwc4_f32(const uint8_t* in, float* out, int64_t npix,
                                   const float* sub_host, float div, fcp_stream_t stream) {
  FCP_REQUIRE(in && out && sub_host, "u8_to_nhwc4: null pointer");
  FCP_REQUIRE(npix > 0, "u8_to_nhwc4: empty input");
  FCP_REQUIRE(((uintptr_t)in & 3) == 0 && ((uintptr_t)out & 15) == 0, "u8_to_nhwc4: misaligned buffers");
  const int do_div = div != 1.0f;
  hipLaunchKernelGGL(u8_to_nhwc4_kernel, dim3(grid_for((npix + 3) / 4, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, (long)npix, sub_host[0], sub_host[1], sub_host[2], div,
                     do_div);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_f32nchw_to_nhwc4_f32(const float* in, float* out, int n, int h, int w,
                                        const float* sub_host, float div, fcp_stream_t stream) {
  FCP_REQUIRE(in && out && sub_host, "f32nchw_to_nhwc4: null pointer");
  FCP_REQUIRE(n > 0 && h > 0 && w > 0, "f32nchw_to_nhwc4: empty input");
  FCP_REQUIRE(((uintptr_t)out & 15) == 0, "f32nchw_to_nhwc4: misaligned output");
  const long hw = (long)h * w;
  hipLaunchKernelGGL(f32nchw_to_nhwc4_kernel, dim3(grid_for(n * hw, 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, n, hw, sub_host[0], sub_host[1], sub_host[2], div, (int)(div != 1.0f));
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_maxpool3x3s2_nhwc_f32(const float* in, float* out, int n, int h, int w, int c, int out_ld,
                                         int out_h, int out_w, fcp_stream_t stream) {
  FCP_REQUIRE(in && out, "maxpool: null pointer");
  FCP_REQUIRE(c % 4 == 0 && out_ld % 4 == 0 && out_ld >= c && ((uintptr_t)out & 15) == 0,
              "maxpool: c and out_ld must be multiples of 4, out_ld >= c, out 16-byte aligned");
  FCP_REQUIRE(out_h == (h + 2 - 3) / 2 + 1 && out_w == (w + 2 - 3) / 2 + 1, "maxpool: bad output size");
  const long total = (long)n * out_h * out_w * (c / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, n, h, w, c / 4, out_ld / 4, out_h, out_w);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_maxpool3x3s2_split32(const float* in, float* out, int n, int h, int w, int c, int out_ld,
                                        int out_h, int out_w, fcp_stream_t stream) {
  FCP_REQUIRE(in && out, "maxpool: null pointer");
  FCP_REQUIRE(c % 32 == 0 && out_ld % 32 == 0 && out_ld >= c && ((uintptr_t)in & 127) == 0 && ((uintptr_t)out & 127) == 0,
              "maxpool(split32): c and out_ld must be multiples of 32 (out_ld >= c) and buffers 128-byte aligned");
  FCP_REQUIRE(out_h == (h + 2 - 3) / 2 + 1 && out_w == (w + 2 - 3) / 2 + 1, "maxpool: bad output size");
  const long total = (long)n * out_h * out_w * (c / 8);
  hipLaunchKernelGGL(maxpool3x3s2_split_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, n, h, w, c, out_ld, out_h, out_w);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_f32_to_split32(const float* in, float* out, int64_t npix, int c, fcp_stream_t stream) {
  FCP_REQUIRE(in && out && npix > 0 && c > 0 && c % 32 == 0, "f32_to_split32: bad arguments (c %% 32 == 0)");
  hipLaunchKernelGGL(split_convert_kernel<true>, dim3(grid_for(npix * (c / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, (long)npix, c);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_split32_to_f32(const float* in, float* out, int64_t npix, int c, fcp_stream_t stream) {
  FCP_REQUIRE(in && out && npix > 0 && c > 0 && c % 32 == 0, "split32_to_f32: bad arguments (c %% 32 == 0)");
  hipLaunchKernelGGL(split_convert_kernel<false>, dim3(grid_for(npix * (c / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, (long)npix, c);
  FCP_LAUNCH_OK();
  return 0;
}
