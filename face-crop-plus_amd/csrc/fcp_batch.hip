// Batch builder: cv2.resize (INTER_AREA / INTER_CUBIC, uint8) + cv2.copyMakeBorder of
// a ragged list of images into one (n, H, W, 3) uint8 batch (reference utils.py:316-335),
// one launch for the whole batch.
//
// Restates the portable C++ path of cv::resize (imgproc/resize.cpp):
//  * same size: copy;
//  * INTER_CUBIC 8U: float32 Keys coefficients (A = -0.75) -> int16 (scale 2^11),
//    int32 horizontal pass with replicated edge columns, vertical pass over clipped rows,
//    (v + 2^21) >> 22 saturated;
//  * INTER_AREA: integral scales -> box sums ((s+2)>>2 for 2x2, else cvRound(sum * (1.f/area)));
//    otherwise computeResizeAreaTab's float32 alpha tables (evaluated in double, per thread)
//    with the row accumulation and the column accumulation in float32, table order, cvRound.
// One thread per destination pixel (3 channels); every source byte of a cell is read by one or
// two threads only, so the kernel is a single pass over the source blob (HBM-bound).
// Built with -ffp-contract=off: every float/double expression rounds like the scalar C++.
#include "fcp_common.h"
#include "fcp_hip.h"

#include <cfloat>
#include <vector>

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// interpolateCubic + saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE)
__device__ __forceinline__ void cubic_tab(int d, double scale, int& s, int ic[4]) {
  float f = (float)((d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  const float A = -0.75f;
  float c[4];
  const float x1 = f + 1.f;
  c[0] = ((A * x1 - 5 * A) * x1 + 8 * A) * x1 - 4 * A;
  c[1] = ((A + 2) * f - (A + 3)) * f * f + 1;
  const float y = 1.f - f;
  c[2] = ((A + 2) * y - (A + 3)) * y * y + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) ic[k] = clampi((int)rintf(c[k] * 2048.f), -32768, 32767);
}

// One destination index of computeResizeAreaTab: entries are
//   [sx1-1 : a_first] (if has_first), [sx1 .. sx2-1 : a_mid], [sx2 : a_last] (if has_last).
struct AreaCell {
  int sx1, sx2;
  float a_first, a_mid, a_last;
  bool has_first, has_last;
};

__device__ __forceinline__ AreaCell area_cell(int d, double scale, int ssize) {
  AreaCell c;
  const double fsx1 = d * scale;
  const double fsx2 = fsx1 + scale;
  const double cell = fmin(scale, ssize - fsx1);
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = min(sx2, ssize - 1);
  sx1 = min(sx1, sx2);
  c.sx1 = sx1;
  c.sx2 = sx2;
  c.has_first = sx1 - fsx1 > 1e-3;
  c.a_first = (float)((sx1 - fsx1) / cell);
  c.a_mid = (float)(1.0 / cell);
  c.has_last = fsx2 - sx2 > 1e-3;
  c.a_last = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell);
  return c;
}

__device__ __forceinline__ int cell_count(const AreaCell& c) {
  return (c.has_first ? 1 : 0) + (c.sx2 - c.sx1) + (c.has_last ? 1 : 0);
}
__device__ __forceinline__ void cell_entry(const AreaCell& c, int j, int& si, float& a) {
  if (c.has_first) {
    if (j == 0) { si = c.sx1 - 1; a = c.a_first; return; }
    --j;
  }
  if (j < c.sx2 - c.sx1) { si = c.sx1 + j; a = c.a_mid; return; }
  si = c.sx2; a = c.a_last;
}

__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)clampi(v, 0, 255); }

__global__ void __launch_bounds__(256) build_batch_kernel(const uint8_t* __restrict__ blob,
                                                          const fcp_batch_item* __restrict__ items, int H, int W,
                                                          uint8_t* __restrict__ out) {
  const int img = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const fcp_batch_item it = items[img];
  uint8_t* o = out + (((long)img * H + y) * W + x) * 3;
  const int dx = x - it.left, dy = y - it.top;
  if ((unsigned)dx >= (unsigned)it.dw || (unsigned)dy >= (unsigned)it.dh) {
    o[0] = 0; o[1] = 0; o[2] = 0;          // BORDER_CONSTANT (value 0); other modes: fill_border_kernel
    return;
  }
  const uint8_t* S = blob + it.src_off;
  const int sh = it.sh, sw = it.sw;
  if (sh == it.dh && sw == it.dw) {        // "Source and destination are of same size. Use simple copy."
    const uint8_t* p = S + ((long)dy * sw + dx) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    return;
  }
  const double scale_x = 1. / ((double)it.dw / sw), scale_y = 1. / ((double)it.dh / sh);
  if (it.interp == 0) {                    // INTER_CUBIC
    int sx, sy, ia[4], ib[4];
    cubic_tab(dx, scale_x, sx, ia);
    cubic_tab(dy, scale_y, sy, ib);
    int v[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint8_t* row = S + (long)clampi(sy - 1 + k, 0, sh - 1) * sw * 3;
      int hsum[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint8_t* p = row + clampi(sx - 1 + j, 0, sw - 1) * 3;
        hsum[0] += p[0] * ia[j]; hsum[1] += p[1] * ia[j]; hsum[2] += p[2] * ia[j];
      }
      v[0] += hsum[0] * ib[k]; v[1] += hsum[1] * ib[k]; v[2] += hsum[2] * ib[k];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = sat_u8((v[c] + (1 << 21)) >> 22);
    return;
  }
  // INTER_AREA (scale_x, scale_y >= 1: checked by the launcher)
  const int isx = (int)rint(scale_x), isy = (int)rint(scale_y);
  if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON) {
    int sum[3] = {0, 0, 0};
    for (int yy = 0; yy < isy; ++yy) {
      const uint8_t* p = S + ((long)(dy * isy + yy) * sw + (long)dx * isx) * 3;
      for (int xx = 0; xx < isx; ++xx, p += 3) { sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; }
    }
    if (isx == 2 && isy == 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((sum[c] + 2) >> 2);
    } else {
      const float inv_area = __fdiv_rn(1.f, (float)(isx * isy));
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = sat_u8((int)rintf((float)sum[c] * inv_area));
    }
    return;
  }
  const AreaCell cx = area_cell(dx, scale_x, sw), cy = area_cell(dy, scale_y, sh);
  const int nx = cell_count(cx), ny = cell_count(cy);
  float sum[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < ny; ++k) {
    int sy; float beta;
    cell_entry(cy, k, sy, beta);
    const uint8_t* row = S + (long)sy * sw * 3;
    float buf[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < nx; ++j) {
      int sx; float alpha;
      cell_entry(cx, j, sx, alpha);
      const uint8_t* p = row + sx * 3;
      buf[0] = buf[0] + (float)p[0] * alpha;
      buf[1] = buf[1] + (float)p[1] * alpha;
      buf[2] = buf[2] + (float)p[2] * alpha;
    }
    if (k == 0) {
      sum[0] = beta * buf[0]; sum[1] = beta * buf[1]; sum[2] = beta * buf[2];
    } else {
      sum[0] += beta * buf[0]; sum[1] += beta * buf[1]; sum[2] += beta * buf[2];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = sat_u8((int)rintf(sum[c]));
}

// cv::borderInterpolate (replicate / reflect / wrap / reflect_101)
__device__ __forceinline__ int border_index(int p, int n, int border) {
  if ((unsigned)p < (unsigned)n) return p;
  if (border == 1) return p < 0 ? 0 : n - 1;
  if (border == 3) {
    p %= n;
    return p < 0 ? p + n : p;
  }
  if (n == 1) return 0;
  const int delta = border == 4 ? 1 : 0;
  do {
    if (p < 0) p = -p - 1 + delta;
    else p = n - 1 - (p - n) - delta;
  } while ((unsigned)p >= (unsigned)n);
  return p;
}

// copyMakeBorder for the non-constant modes: border pixels copy their mapped interior pixel of
// the same batch slot (interior pixels are not touched, so there is no read/write overlap).
__global__ void __launch_bounds__(256) fill_border_kernel(const fcp_batch_item* __restrict__ items, int H, int W,
                                                          int border, uint8_t* __restrict__ out) {
  const int img = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const fcp_batch_item it = items[img];
  const int dx = x - it.left, dy = y - it.top;
  if ((unsigned)dx < (unsigned)it.dw && (unsigned)dy < (unsigned)it.dh) return;
  uint8_t* base = out + (long)img * H * W * 3;
  const int mx = border_index(dx, it.dw, border) + it.left, my = border_index(dy, it.dh, border) + it.top;
  const uint8_t* p = base + ((long)my * W + mx) * 3;
  uint8_t* o = base + ((long)y * W + x) * 3;
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

}  // namespace

extern "C" int fcp_build_batch_u8(const uint8_t* src_blob, int64_t blob_bytes, const fcp_batch_item* items_host,
                                  const fcp_batch_item* items_dev, int n, int out_h, int out_w, int border,
                                  uint8_t* out, fcp_stream_t stream) {
  FCP_REQUIRE(n >= 0 && out_h > 0 && out_w > 0, "fcp_build_batch_u8: bad batch geometry n=%d %dx%d", n, out_h, out_w);
  FCP_REQUIRE(border >= 0 && border <= 4, "fcp_build_batch_u8: unknown border mode %d", border);
  if (n == 0) return 0;
  FCP_REQUIRE(src_blob && items_host && items_dev && out, "fcp_build_batch_u8: null pointer");
  FCP_REQUIRE(n <= 65535, "fcp_build_batch_u8: n=%d exceeds the grid limit", n);
  for (int i = 0; i < n; ++i) {
    const fcp_batch_item& it = items_host[i];
    FCP_REQUIRE(it.sh > 0 && it.sw > 0 && it.dh > 0 && it.dw > 0, "fcp_build_batch_u8: item %d has an empty image", i);
    FCP_REQUIRE(it.src_off >= 0 && it.src_off + (int64_t)it.sh * it.sw * 3 <= blob_bytes,
                "fcp_build_batch_u8: item %d lies outside the source blob", i);
    FCP_REQUIRE(it.top >= 0 && it.left >= 0 && it.top + it.dh <= out_h && it.left + it.dw <= out_w,
                "fcp_build_batch_u8: item %d (%dx%d at %d,%d) does not fit %dx%d", i, it.dw, it.dh, it.left, it.top,
                out_w, out_h);
    FCP_REQUIRE(it.interp == 0 || it.interp == 1, "fcp_build_batch_u8: item %d: interp must be 0 (cubic) or 1 (area)", i);
    FCP_REQUIRE(it.interp == 0 || (it.dh <= it.sh && it.dw <= it.sw),
                "fcp_build_batch_u8: item %d: INTER_AREA is implemented for decimation only", i);
    FCP_REQUIRE((int64_t)it.sh * it.sw * 3 < ((int64_t)1 << 31), "fcp_build_batch_u8: item %d too large", i);
  }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(fcp_cdiv(out_w, 64), fcp_cdiv(out_h, 4), n);
  build_batch_kernel<<<grid, 256, 0, s>>>(src_blob, items_dev, out_h, out_w, out);
  FCP_LAUNCH_OK();
  if (border != 0) {
    fill_border_kernel<<<grid, 256, 0, s>>>(items_dev, out_h, out_w, border, out);
    FCP_LAUNCH_OK();
  }
  return 0;
}
