// 5-point transform estimation + OpenCV-exact bilinear warpAffine (uint8, 3ch).
//
// estimate: cv2.estimateAffinePartial2D / estimateAffine2D with
// ransacReprojThreshold=inf accept the first minimal sample as all-inlier and
// then refine on all points (10 LM iterations on a *linear* residual), i.e. they
// return the linear least-squares similarity / affine; computed here in closed
// form in float64, one lane per face.
//
// warp: restatement of cv::warpAffine's fixed-point path (imgwarp.cpp):
// inverse map in double, AB_BITS=10, round_delta=16, INTER_BITS=5, bilinear
// weights = (32-fy)(32-fx)*32 etc. (sum 32768), out = (sum + 16384) >> 15,
// borders through cv::borderInterpolate.  Built with -ffp-contract=off so the
// double expressions round exactly like the scalar C++ they restate.
#include "fcp_common.h"
#include "fcp_hip.h"

namespace {

__device__ __forceinline__ bool estimate_one(const float* __restrict__ src, const float* __restrict__ dst, int i, int k,
                                             int allow_skew, bool is_live, double* __restrict__ mat, int* __restrict__ ok);

__global__ void __launch_bounds__(64) estimate_transform_kernel(const float* __restrict__ src,
                                                                const float* __restrict__ dst, int f, int k,
                                                                int allow_skew, double* __restrict__ mat,
                                                                int* __restrict__ ok,
                                                                const int* __restrict__ face_count,
                                                                long long* __restrict__ valid_total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int live = face_count != nullptr ? min(*face_count, f) : f;   // rows >= live are padding of a fixed-size batch
  bool counted = false;
  if (i < f) counted = estimate_one(src, dst, i, k, allow_skew, i < live, mat, ok);
  if (valid_total != nullptr) {                                        // whole wave gets here: one atomic per wave
    const unsigned long long votes = __ballot(counted);
    if ((threadIdx.x & 63) == 0 && votes != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(valid_total), (unsigned long long)__popcll(votes));
  }
}

// One face: returns whether it is a valid (live, non-degenerate) face; writes its matrix and ok flag.
__device__ __forceinline__ bool estimate_one(const float* __restrict__ src, const float* __restrict__ dst, int i, int k,
                                             int allow_skew, bool is_live, double* __restrict__ mat, int* __restrict__ ok) {
  const float* s = src + (long)i * k * 2;
  bool finite = is_live;
  double mx = 0, my = 0, MX = 0, MY = 0;
  for (int p = 0; p < k; ++p) {
    const double x = s[2 * p], y = s[2 * p + 1];
    finite = finite && isfinite(x) && isfinite(y);
    mx += x; my += y; MX += dst[2 * p]; MY += dst[2 * p + 1];
  }
  mx /= k; my /= k; MX /= k; MY /= k;
  double m[6] = {0, 0, 0, 0, 0, 0};
  bool good = finite;
  if (!allow_skew) {
    // minimise sum |a x - b y + tx - X|^2 + |b x + a y + ty - Y|^2
    double sxx = 0, sa = 0, sb = 0;
    for (int p = 0; p < k; ++p) {
      const double x = s[2 * p] - mx, y = s[2 * p + 1] - my;
      const double X = dst[2 * p] - MX, Y = dst[2 * p + 1] - MY;
      sxx += x * x + y * y;
      sa += x * X + y * Y;
      sb += x * Y - y * X;
    }
    good = good && sxx > 0.0;
    if (good) {
      const double a = sa / sxx, b = sb / sxx;
      m[0] = a; m[1] = -b; m[2] = MX - a * mx + b * my;
      m[3] = b; m[4] = a;  m[5] = MY - b * mx - a * my;
    }
  } else {
    // full affine: two independent 2-unknown LSQ problems on centred coordinates
    double sxx = 0, sxy = 0, syy = 0, sxX = 0, syX = 0, sxY = 0, syY = 0;
    for (int p = 0; p < k; ++p) {
      const double x = s[2 * p] - mx, y = s[2 * p + 1] - my;
      const double X = dst[2 * p] - MX, Y = dst[2 * p + 1] - MY;
      sxx += x * x; sxy += x * y; syy += y * y;
      sxX += x * X; syX += y * X; sxY += x * Y; syY += y * Y;
    }
    const double det = sxx * syy - sxy * sxy;
    good = good && fabs(det) > 1e-12 * (sxx * syy + 1e-300);
    if (good) {
      const double a = (sxX * syy - syX * sxy) / det, b = (syX * sxx - sxX * sxy) / det;
      const double c = (sxY * syy - syY * sxy) / det, d = (syY * sxx - sxY * sxy) / det;
      m[0] = a; m[1] = b; m[2] = MX - a * mx - b * my;
      m[3] = c; m[4] = d; m[5] = MY - c * mx - d * my;
    }
  }
  for (int q = 0; q < 6; ++q) good = good && isfinite(m[q]);
  for (int q = 0; q < 6; ++q) mat[(long)i * 6 + q] = good ? m[q] : 0.0;
  ok[i] = good ? 1 : 0;
  return good;
}

// cv::saturate_cast<int>(double) == cvRound (cvtsd2si: nearest-even, 0x80000000 when out of range)
__device__ __forceinline__ int cv_round(double v) {
  if (!(v >= -2147483648.0 && v < 2147483648.0)) return (int)0x80000000;
  return (int)rint(v);
}
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// cv::borderInterpolate; border codes follow cv2.BORDER_*: 0 constant, 1 replicate, 2 reflect, 3 wrap, 4 reflect_101
__device__ __forceinline__ int border_interp(int p, int len, int border) {
  if ((unsigned)p < (unsigned)len) return p;
  if (border == 1) return p < 0 ? 0 : len - 1;
  if (border == 2 || border == 4) {
    const int delta = border == 4;
    if (len == 1) return 0;
    do {
      if (p < 0) p = -p - 1 + delta;
      else p = len - 1 - (p - len) - delta;
    } while ((unsigned)p >= (unsigned)len);
    return p;
  }
  if (border == 3) {
    if (p < 0) p -= ((p - len + 1) / len) * len;
    if (p >= len) p %= len;
    return p;
  }
  return -1;  // constant
}

// Six consecutive bytes (two RGB pixels) at byte offset `off` of `base`: one aligned 12-byte load + a funnel shift
// instead of six byte loads.  The caller guarantees that (off & ~3) + 12 stays inside the allocation.
__device__ __forceinline__ unsigned long long load6(const uint8_t* __restrict__ base, long off) {
  const long a = off & ~3L;
  const unsigned sh = (unsigned)(off & 3) * 8u;
  const unsigned* q = reinterpret_cast<const unsigned*>(base + a);
  const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
  const unsigned long long lo = (unsigned long long)d0 | ((unsigned long long)d1 << 32);
  return sh ? (lo >> sh) | ((unsigned long long)d2 << (64u - sh)) : lo;
}

template <int PX>
__global__ void __launch_bounds__(256) warp_affine_kernel(
    const uint8_t* __restrict__ images, int n, int h, int w, const int* __restrict__ img_idx,
    const double* __restrict__ mat, const int* __restrict__ ok, const int* __restrict__ paddings,
    int out_h, int out_w, int border, uint8_t* __restrict__ out) {
  const int face = blockIdx.y;
  const int groups_per_row = (out_w + PX - 1) / PX;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = g < groups_per_row * out_h;
  const int y = active ? g / groups_per_row : 0;
  const int x0 = active ? (g - y * groups_per_row) * PX : 0;
  uint8_t* dst = out + (((long)face * out_h + y) * out_w + x0) * 3;
  uint8_t px[PX * 3];
#pragma unroll
  for (int q = 0; q < PX * 3; ++q) px[q] = 0;

  const bool valid = ok == nullptr || ok[face] != 0;       // uniform over the workgroup (one face per blockIdx.y)
  // The inverse map is the same for every lane of the workgroup: one lane inverts the forward transform exactly like
  // cv::warpAffine does (two double divisions), the others read it from LDS.
  __shared__ double sM[6];
  if (valid && threadIdx.x == 0) {
    double M[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) M[q] = mat[(long)face * 6 + q];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
#pragma unroll
    for (int q = 0; q < 6; ++q) sM[q] = M[q];
  }
  __syncthreads();
  if (!active) return;
  if (valid) {
    const int img = img_idx[face];
    int pt = 0, pb = 0, pl = 0, pr = 0;
    if (paddings != nullptr) { pt = paddings[img * 4]; pb = paddings[img * 4 + 1]; pl = paddings[img * 4 + 2]; pr = paddings[img * 4 + 3]; }
    const int sh = h - pt - pb, sw = w - pl - pr;  // un-padded slice (cropper.py:538-539)
    const long sstep = (long)w * 3;
    const long s0off = ((long)img * h + pt) * sstep + (long)pl * 3;
    const uint8_t* S0 = images + s0off;
    const long total = (long)n * h * sstep;          // bytes of the batch: bound of the 12-byte loads
    double M[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) M[q] = sM[q];
    const int X0 = (int)((unsigned)cv_round((M[1] * y + M[2]) * 1024.0) + 16u);
    const int Y0 = (int)((unsigned)cv_round((M[4] * y + M[5]) * 1024.0) + 16u);
    const int width1 = sw - 1 > 0 ? sw - 1 : 0, height1 = sh - 1 > 0 ? sh - 1 : 0;
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int x = x0 + q;
      if (x >= out_w) break;
      const int ad = cv_round(M[0] * x * 1024.0), bd = cv_round(M[3] * x * 1024.0);
      const int X = (int)((unsigned)X0 + (unsigned)ad) >> 5;
      const int Y = (int)((unsigned)Y0 + (unsigned)bd) >> 5;
      const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
      const int fx = X & 31, fy = Y & 31;
      const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32;
      const int w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
      const uint8_t *v0, *v1, *v2, *v3;
      const uint8_t zero[3] = {0, 0, 0};
      if ((unsigned)sx < (unsigned)width1 && (unsigned)sy < (unsigned)height1) {
        const long o0 = s0off + sy * sstep + sx * 3, o1 = o0 + sstep;
        if (((o1 & ~3L) + 12) <= total) {
          // interior: the 2 x 2 neighbourhood is two runs of six bytes — two 12-byte loads instead of twelve byte loads
          const unsigned long long r0 = load6(images, o0), r1 = load6(images, o1);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int a0 = (int)((r0 >> (8 * c)) & 255u), a1 = (int)((r0 >> (8 * c + 24)) & 255u);
            const int a2 = (int)((r1 >> (8 * c)) & 255u), a3 = (int)((r1 >> (8 * c + 24)) & 255u);
            const int acc = a0 * w0 + a1 * w1 + a2 * w2 + a3 * w3;
            const int r = (acc + (1 << 14)) >> 15;
            px[q * 3 + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
          }
          continue;
        }
        v0 = images + o0; v1 = v0 + 3; v2 = v0 + sstep; v3 = v2 + 3;
      } else {
        if (border == 0 && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) continue;  // all-constant: 0
        int sx0, sx1, sy0, sy1;
        if (border == 1) {
          sx0 = sx < 0 ? 0 : (sx < sw ? sx : sw - 1);
          sx1 = sx + 1 < 0 ? 0 : (sx + 1 < sw ? sx + 1 : sw - 1);
          sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
          sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        } else {
          sx0 = border_interp(sx, sw, border); sx1 = border_interp(sx + 1, sw, border);
          sy0 = border_interp(sy, sh, border); sy1 = border_interp(sy + 1, sh, border);
        }
        v0 = (sx0 >= 0 && sy0 >= 0) ? S0 + sy0 * sstep + sx0 * 3 : zero;
        v1 = (sx1 >= 0 && sy0 >= 0) ? S0 + sy0 * sstep + sx1 * 3 : zero;
        v2 = (sx0 >= 0 && sy1 >= 0) ? S0 + sy1 * sstep + sx0 * 3 : zero;
        v3 = (sx1 >= 0 && sy1 >= 0) ? S0 + sy1 * sstep + sx1 * 3 : zero;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int acc = v0[c] * w0 + v1[c] * w1 + v2[c] * w2 + v3[c] * w3;
        const int r = (acc + (1 << 14)) >> 15;
        px[q * 3 + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
      }
    }
  }
  if (PX == 4 && x0 + 4 <= out_w && (out_w & 3) == 0) {
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);  // 12 aligned bytes
    d32[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24);
    d32[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24);
    d32[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((uint32_t)px[11] << 24);
  } else {
    for (int q = 0; q < PX && x0 + q < out_w; ++q) {
      dst[q * 3] = px[q * 3]; dst[q * 3 + 1] = px[q * 3 + 1]; dst[q * 3 + 2] = px[q * 3 + 2];
    }
  }
}

}  // namespace

extern "C" int fcp_estimate_transform_counted(const float* src, const float* dst, int f, int k, int allow_skew,
                                              const int32_t* face_count, double* mat, int32_t* ok,
                                              int64_t* valid_total, fcp_stream_t stream) {
  FCP_REQUIRE(src && dst && mat && ok, "estimate_transform: null pointer");
  FCP_REQUIRE(f > 0 && k >= 2 && k <= 128, "estimate_transform: bad sizes (f=%d, k=%d)", f, k);
  hipLaunchKernelGGL(estimate_transform_kernel, dim3(fcp_cdiv(f, 64)), dim3(64), 0, (hipStream_t)stream, src,
                     dst, f, k, allow_skew, mat, ok, face_count, reinterpret_cast<long long*>(valid_total));
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_estimate_transform(const float* src, const float* dst, int f, int k, int allow_skew,
                                      double* mat, int32_t* ok, fcp_stream_t stream) {
  return fcp_estimate_transform_counted(src, dst, f, k, allow_skew, nullptr, mat, ok, nullptr, stream);
}

extern "C" int fcp_warp_affine_u8(const uint8_t* images, int n, int h, int w, const int32_t* img_idx,
                                  const double* mat, const int32_t* ok, const int32_t* paddings, int f,
                                  int out_h, int out_w, int border, uint8_t* out, fcp_stream_t stream) {
  FCP_REQUIRE(images && img_idx && mat && out, "warp_affine: null pointer");
  FCP_REQUIRE(n > 0 && h > 0 && w > 0 && f > 0 && out_h > 0 && out_w > 0, "warp_affine: bad sizes");
  FCP_REQUIRE(border >= 0 && border <= 4, "warp_affine: unsupported border mode %d", border);
  FCP_REQUIRE(f <= 65535, "warp_affine: at most 65535 faces per call");
  const int groups = ((out_w + 3) / 4) * out_h;
  hipLaunchKernelGGL((warp_affine_kernel<4>), dim3(fcp_cdiv(groups, 256), f), dim3(256), 0,
                     (hipStream_t)stream, images, n, h, w, img_idx, mat, ok, paddings, out_h, out_w, border,
                     out);
  FCP_LAUNCH_OK();
  return 0;
}
