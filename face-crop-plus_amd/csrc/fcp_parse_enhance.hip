// HBM-bound glue kernels of the BiSeNet parser and the RRDB enhancer:
//   * face pre-processing: /255 -> bilinear 512x512 (align_corners=False) -> (x-mean)/std
//   * global average pool, tiny fully-connected (1x1 conv on 1x1 maps) with BN + act,
//     channel-attention scale-add
//   * parse tail: bilinear x8 (align_corners=True) + nearest resize to the crop size +
//     argmax fused (only the pixels the nearest resize keeps are ever evaluated),
//     class histogram, group masks
//   * enhancer tail: bicubic x0.25 (4-tap, A=-0.75) + clamp + *255 + round-half-even -> uint8
// Float expressions follow ATen's CPU kernels op by op (file built with -ffp-contract=off).
#include "fcp_common.h"
#include "fcp_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

inline int grid_for(long items, int block, long cap = 16384) {
  long g = (items + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ATen area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
__device__ __forceinline__ void src_index_linear(float scale, int dst, int in_size, int& i0, int& i1,
                                                 float& l0, float& l1) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  const int off = (i0 < in_size - 1) ? 1 : 0;
  i1 = i0 + off;
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256) bise_preprocess_kernel(const uint8_t* __restrict__ in, int f, int h,
                                                              int w, float* __restrict__ out, int oh, int ow,
                                                              float m0, float m1, float m2, float s0, float s1,
                                                              float s2) {
  const long total = (long)f * oh * ow;
  const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % ow);
    const long t = i / ow;
    const int y = (int)(t % oh);
    const int fi = (int)(t / oh);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index_linear(sh, y, h, y0, y1, hy0, hy1);
    src_index_linear(sw, x, w, x0, x1, wx0, wx1);
    const uint8_t* base = in + (long)fi * h * w * 3;
    const uint8_t* p00 = base + ((long)y0 * w + x0) * 3;
    const uint8_t* p01 = base + ((long)y0 * w + x1) * 3;
    const uint8_t* p10 = base + ((long)y1 * w + x0) * 3;
    const uint8_t* p11 = base + ((long)y1 * w + x1) * 3;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v00 = (float)p00[c] / 255.0f, v01 = (float)p01[c] / 255.0f;
      const float v10 = (float)p10[c] / 255.0f, v11 = (float)p11[c] / 255.0f;
      const float v = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
      o[c] = (v - mean[c]) / sd[c];
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

// one workgroup per (image, 64-channel group): mean over h*w of an NHWC tensor slice.  Wave `part` sums the pixels part,
// part + 4, ... in ascending order and the four partial sums are added in order: that order is kept (the labels it feeds are
// held to the reference fixture bit for bit), but 32 loads are in flight per lane before the first add — as a plain
// load-add loop the 1024-pixel chains of BiSeNet's 64 x 64 feature map took 317 us of exposed HBM latency for 134 MB.
__global__ void __launch_bounds__(256) avgpool_kernel(const float* __restrict__ in, int hw, int c, int ld,
                                                      float* __restrict__ out) {
  __shared__ float red[4][64];
  const int img = blockIdx.x, cg = blockIdx.y;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int ch = cg * 64 + lane;
  constexpr int U = 32;
  float acc = 0.f;
  if (ch < c) {
    const float* base = in + (long)img * hw * ld + ch;
    int p = part;
    for (; p + 4 * (U - 1) < hw; p += 4 * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = base[(long)(p + 4 * u) * ld];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; p < hw; p += 4) acc += base[(long)p * ld];
  }
  red[part][lane] = acc;
  __syncthreads();
  if (part == 0 && ch < c) out[(long)img * c + ch] = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)hw;
}

// out[n][co] = act(scale[co] * dot(w[co,:], in[n,:]) + shift[co]); one wave per output
__global__ void __launch_bounds__(256) fc_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                 int n, int cin, int cout, int act, float* __restrict__ out) {
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= n * cout) return;
  const int ni = gw / cout, co = gw - ni * cout;
  float acc = 0.f;
  for (int k = lane; k < cin; k += 64) acc += w[(long)co * cin + k] * in[(long)ni * cin + k];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) {
    float v = acc;
    if (scale != nullptr) v = v * scale[co];
    if (shift != nullptr) v = v + shift[co];
    if (act == 1) v = v > 0.f ? v : 0.f;
    else if (act == 2) v = 1.0f / (1.0f + expf(-v));
    out[gw] = v;
  }
}

// out = x * s[n,c] (+ addv[n,c]) (+ addt[n,h,w,c]); float4 over channels
__global__ void __launch_bounds__(256) scale_add_kernel(const float* __restrict__ x, int x_ld,
                                                        const float* __restrict__ s, const float* __restrict__ addv,
                                                        const float* __restrict__ addt, int addt_ld, long npix,
                                                        int hw, int c, float* __restrict__ out, int out_ld) {
  const int c4 = c >> 2;
  const long total = npix * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4) * 4;
    const long p = i / c4;
    const int ni = (int)(p / hw);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + p * x_ld + cc);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (long)ni * c + cc);
    v = v * sv;
    if (addv != nullptr) v = v + *reinterpret_cast<const f32x4*>(addv + (long)ni * c + cc);
    if (addt != nullptr) v = v + *reinterpret_cast<const f32x4*>(addt + p * addt_ld + cc);
    *reinterpret_cast<f32x4*>(out + p * out_ld + cc) = v;
  }
}

// logits (f, lh, lw, ld>=ncls) at 1/8 resolution -> labels (f, oh, ow) uint8.
// Semantics: F.interpolate(bilinear, align_corners=True) to (mid_h, mid_w), then
// F.interpolate(nearest) to (oh, ow), then argmax over classes (first maximum).
__global__ void __launch_bounds__(256) parse_tail_kernel(const float* __restrict__ logits, int f, int lh, int lw,
                                                         int ld, int ncls, int mid_h, int mid_w, int oh, int ow,
                                                         uint8_t* __restrict__ labels) {
  const long total = (long)f * oh * ow;
  const float nsh = (float)mid_h / (float)oh, nsw = (float)mid_w / (float)ow;       // nearest scales
  const float bsh = mid_h > 1 ? (float)(lh - 1) / (float)(mid_h - 1) : 0.f;         // align_corners=True
  const float bsw = mid_w > 1 ? (float)(lw - 1) / (float)(mid_w - 1) : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % ow);
    const long t = i / ow;
    const int y = (int)(t % oh);
    const int fi = (int)(t / oh);
    int my = (int)floorf(y * nsh), mx = (int)floorf(x * nsw);
    my = my < mid_h - 1 ? my : mid_h - 1;
    mx = mx < mid_w - 1 ? mx : mid_w - 1;
    const float sy = bsh * my, sx = bsw * mx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < lh - 1 ? 1 : 0), x1 = x0 + (x0 < lw - 1 ? 1 : 0);
    const float hy1 = sy - (float)y0, hy0 = 1.f - hy1, wx1 = sx - (float)x0, wx0 = 1.f - wx1;
    const float* b = logits + (long)fi * lh * lw * ld;
    const float* p00 = b + ((long)y0 * lw + x0) * ld;
    const float* p01 = b + ((long)y0 * lw + x1) * ld;
    const float* p10 = b + ((long)y1 * lw + x0) * ld;
    const float* p11 = b + ((long)y1 * lw + x1) * ld;
    float best = -INFINITY;
    int arg = 0;
    bool seen_nan = false;
    for (int c = 0; c < ncls; ++c) {
      const float v = hy0 * (wx0 * p00[c] + wx1 * p01[c]) + hy1 * (wx0 * p10[c] + wx1 * p11[c]);
      if (seen_nan) continue;
      if (v != v) { arg = c; seen_nan = true; }          // torch.argmax treats NaN as the maximum
      else if (v > best) { best = v; arg = c; }
    }
    labels[i] = (uint8_t)arg;
  }
}

// per-face class histogram.  Sixteen replicas of the 32-bin histogram in LDS (thread t uses replica t & 15): a face is
// mostly one or two classes, and 256 threads adding to ONE LDS word serialise (64 us per 256 x 256 face with a single
// histogram); integer counts, so the result does not depend on the order.
__global__ void __launch_bounds__(256) label_hist_kernel(const uint8_t* __restrict__ labels, int hw, int ncls,
                                                         int* __restrict__ counts) {
  __shared__ int hist[16][33];
  const int fi = blockIdx.x;
  for (int i = threadIdx.x; i < 16 * 33; i += blockDim.x) (&hist[0][0])[i] = 0;
  __syncthreads();
  const uint8_t* l = labels + (long)fi * hw;
  int* mine = hist[threadIdx.x & 15];
  int p = threadIdx.x * 4;
  if ((hw & 3) == 0 && (((uintptr_t)l) & 3) == 0) {
    for (; p < hw; p += blockDim.x * 4) {                       // four labels per load
      const unsigned v = *reinterpret_cast<const unsigned*>(l + p);
      atomicAdd(&mine[v & 31], 1);
      atomicAdd(&mine[(v >> 8) & 31], 1);
      atomicAdd(&mine[(v >> 16) & 31], 1);
      atomicAdd(&mine[(v >> 24) & 31], 1);
    }
  } else {
    for (int q = threadIdx.x; q < hw; q += blockDim.x) atomicAdd(&mine[l[q] & 31], 1);
  }
  __syncthreads();
  if (threadIdx.x < ncls) {
    int t = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += hist[r][threadIdx.x];
    counts[(long)fi * ncls + threadIdx.x] = t;
  }
}

__global__ void __launch_bounds__(256) label_mask_kernel(const uint8_t* __restrict__ labels, long total,
                                                         unsigned class_bits, uint8_t* __restrict__ mask) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    mask[i] = ((class_bits >> (labels[i] & 31)) & 1u) ? 255 : 0;
}

// x4 (1, 4h, 4w, ld) fp32 -> uint8 RGB (h, w, 3): bicubic x0.25 (align_corners=False, A=-0.75),
// clamp(0,1) * 255, round half to even (rrdb.py:143-144).
__global__ void __launch_bounds__(256) bicubic_down4_kernel(const float* __restrict__ in, int h, int w, int ld,
                                                            uint8_t* __restrict__ out) {
  const float c0 = -0.09375f, c1 = 0.59375f;  // cubic weights at t = 0.5: (-3, 19, 19, -3) / 32
  const float wt[4] = {c0, c1, c1, c0};
  const long total = (long)h * w;
  const int iw = 4 * w, ih = 4 * h;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)(i / w);
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int yy = 4 * y + r;
      yy = yy < 0 ? 0 : (yy > ih - 1 ? ih - 1 : yy);
      float row[3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int xx = 4 * x + q;
        xx = xx < 0 ? 0 : (xx > iw - 1 ? iw - 1 : xx);
        const float* p = in + ((long)yy * iw + xx) * ld;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] = q == 0 ? p[c] * wt[0] : row[c] + p[c] * wt[q];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = r == 0 ? row[c] * wt[0] : acc[c] + row[c] * wt[r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = acc[c];
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      out[i * 3 + c] = (uint8_t)rintf(v * 255.0f);
    }
  }
}

}  // namespace

extern "C" int fcp_bise_preprocess_u8(const uint8_t* faces, int f, int h, int w, float* out, int out_h,
                                      int out_w, const float* mean_host, const float* std_host,
                                      fcp_stream_t stream) {
  FCP_REQUIRE(faces && out && mean_host && std_host, "bise_preprocess: null pointer");
  FCP_REQUIRE(f > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, "bise_preprocess: bad sizes");
  hipLaunchKernelGGL(bise_preprocess_kernel, dim3(grid_for((long)f * out_h * out_w, 256)), dim3(256), 0,
                     (hipStream_t)stream, faces, f, h, w, out, out_h, out_w, mean_host[0], mean_host[1],
                     mean_host[2], std_host[0], std_host[1], std_host[2]);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_avgpool_nhwc_f32(const float* in, int n, int hw, int c, int ld, float* out,
                                    fcp_stream_t stream) {
  FCP_REQUIRE(in && out && n > 0 && hw > 0 && c > 0 && ld >= c, "avgpool: bad arguments");
  hipLaunchKernelGGL(avgpool_kernel, dim3(n, (c + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, hw, c, ld,
                     out);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_fc_f32(const float* in, const float* w, const float* scale, const float* shift, int n,
                          int cin, int cout, int act, float* out, fcp_stream_t stream) {
  FCP_REQUIRE(in && w && out && n > 0 && cin > 0 && cout > 0, "fc: bad arguments");
  FCP_REQUIRE(act >= 0 && act <= 2, "fc: act must be 0 (none), 1 (relu) or 2 (sigmoid)");
  hipLaunchKernelGGL(fc_kernel, dim3(fcp_cdiv((long)n * cout * 64, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     w, scale, shift, n, cin, cout, act, out);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_scale_add_nhwc_f32(const float* x, int x_ld, const float* scale_nc, const float* add_nc,
                                      const float* add_t, int add_t_ld, int n, int hw, int c, float* out,
                                      int out_ld, fcp_stream_t stream) {
  FCP_REQUIRE(x && scale_nc && out && n > 0 && hw > 0, "scale_add: bad arguments");
  FCP_REQUIRE(c % 4 == 0 && x_ld % 4 == 0 && out_ld % 4 == 0 && (add_t == nullptr || add_t_ld % 4 == 0),
              "scale_add: channel counts / strides must be multiples of 4");
  hipLaunchKernelGGL(scale_add_kernel, dim3(grid_for((long)n * hw * (c / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, x_ld, scale_nc, add_nc, add_t, add_t_ld, (long)n * hw, hw, c, out,
                     out_ld);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_parse_tail(const float* logits, int f, int lh, int lw, int ld, int ncls, int mid_h,
                              int mid_w, int out_h, int out_w, uint8_t* labels, int32_t* counts,
                              fcp_stream_t stream) {
  FCP_REQUIRE(logits && labels, "parse_tail: null pointer");
  FCP_REQUIRE(f > 0 && ncls > 0 && ncls <= 32 && ld >= ncls, "parse_tail: bad sizes");
  hipLaunchKernelGGL(parse_tail_kernel, dim3(grid_for((long)f * out_h * out_w, 256)), dim3(256), 0,
                     (hipStream_t)stream, logits, f, lh, lw, ld, ncls, mid_h, mid_w, out_h, out_w, labels);
  FCP_LAUNCH_OK();
  if (counts != nullptr) {
    hipLaunchKernelGGL(label_hist_kernel, dim3(f), dim3(256), 0, (hipStream_t)stream, labels, out_h * out_w,
                       ncls, counts);
    FCP_LAUNCH_OK();
  }
  return 0;
}

extern "C" int fcp_label_mask_u8(const uint8_t* labels, int64_t total, uint32_t class_bits, uint8_t* mask,
                                 fcp_stream_t stream) {
  FCP_REQUIRE(labels && mask && total > 0, "label_mask: bad arguments");
  hipLaunchKernelGGL(label_mask_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, labels,
                     (long)total, class_bits, mask);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_bicubic_down4_u8(const float* x4, int h, int w, int ld, uint8_t* out_rgb,
                                    fcp_stream_t stream) {
  FCP_REQUIRE(x4 && out_rgb && h > 0 && w > 0 && ld >= 3, "bicubic_down4: bad arguments");
  hipLaunchKernelGGL(bicubic_down4_kernel, dim3(grid_for((long)h * w, 256)), dim3(256), 0, (hipStream_t)stream,
                     x4, h, w, ld, out_rgb);
  FCP_LAUNCH_OK();
  return 0;
}
