/*
 * A consumer of include/fcp_hip.h written in plain C: no Python, no torch.
 *
 * Reads one case file (written by tests/test_cabi_c_harness.py from seeded inputs and the oracle's expected
 * outputs), uploads it with the HIP runtime, runs fcp_estimate_transform + fcp_warp_affine_u8 on its own HIP
 * stream — the calls a native host would make where the reference runs cv2.estimateAffinePartial2D +
 * cv2.warpAffine (cropper.py:512-547) — and compares: ok flags equal, matrices within 1e-12, crop bytes equal.
 * Also checks the error contract (negative return + fcp_last_error()).  Exit code 0 = all equal.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/cabi/align_crop_check.c \
 *       -Lface-crop-plus_amd/csrc -lfcp_hip -L/opt/rocm/lib -lamdhip64 -lm -o align_crop_check
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "fcp_hip.h"

#define HIP_OK(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return 2;                                                                             \
    }                                                                                       \
  } while (0)

static void* slurp(FILE* fp, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (p == NULL || fread(p, 1, bytes, fp) != bytes) {
    fprintf(stderr, "short read (%zu bytes)\n", bytes);
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s case.bin\n", argv[0]);
    return 2;
  }
  if (fcp_abi_version() != FCP_ABI_VERSION) {
    fprintf(stderr, "library ABI %d, header ABI %d\n", fcp_abi_version(), FCP_ABI_VERSION);
    return 2;
  }
  FILE* fp = fopen(argv[1], "rb");
  if (fp == NULL) {
    perror(argv[1]);
    return 2;
  }
  int32_t hd[10]; /* n h w f k out_h out_w border allow_skew has_pad */
  if (fread(hd, sizeof(int32_t), 10, fp) != 10) return 2;
  const int n = hd[0], h = hd[1], w = hd[2], f = hd[3], k = hd[4], oh = hd[5], ow = hd[6], border = hd[7], skew = hd[8], has_pad = hd[9];
  const size_t img_b = (size_t)n * h * w * 3, idx_b = (size_t)f * 4, lm_b = (size_t)f * k * 2 * 4, tgt_b = (size_t)k * 2 * 4,
               pad_b = (size_t)n * 4 * 4, ok_b = (size_t)f * 4, crop_b = (size_t)f * oh * ow * 3, mat_b = (size_t)f * 6 * 8;
  uint8_t* images = slurp(fp, img_b);
  int32_t* img_idx = slurp(fp, idx_b);
  float* lms = slurp(fp, lm_b);
  float* tgt = slurp(fp, tgt_b);
  int32_t* pads = slurp(fp, pad_b);
  int32_t* ok_ref = slurp(fp, ok_b);
  uint8_t* crop_ref = slurp(fp, crop_b);
  double* mat_ref = slurp(fp, mat_b);
  fclose(fp);

  hipStream_t stream;
  HIP_OK(hipSetDevice(0));
  HIP_OK(hipStreamCreate(&stream));
  void *d_img, *d_idx, *d_lm, *d_tgt, *d_pad, *d_ok, *d_crop, *d_mat;
  HIP_OK(hipMalloc(&d_img, img_b));
  HIP_OK(hipMalloc(&d_idx, idx_b));
  HIP_OK(hipMalloc(&d_lm, lm_b));
  HIP_OK(hipMalloc(&d_tgt, tgt_b));
  HIP_OK(hipMalloc(&d_pad, pad_b));
  HIP_OK(hipMalloc(&d_ok, ok_b));
  HIP_OK(hipMalloc(&d_crop, crop_b));
  HIP_OK(hipMalloc(&d_mat, mat_b));
  HIP_OK(hipMemcpyAsync(d_img, images, img_b, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_idx, img_idx, idx_b, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_lm, lms, lm_b, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_tgt, tgt, tgt_b, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_pad, pads, pad_b, hipMemcpyHostToDevice, stream));

  if (fcp_estimate_transform(d_lm, d_tgt, f, k, skew, d_mat, d_ok, stream) != 0 ||
      fcp_warp_affine_u8(d_img, n, h, w, d_idx, d_mat, d_ok, has_pad ? d_pad : NULL, f, oh, ow, border, d_crop, stream) != 0) {
    fprintf(stderr, "fcp call failed: %s\n", fcp_last_error());
    return 1;
  }
  int32_t* ok = malloc(ok_b);
  uint8_t* crop = malloc(crop_b);
  double* mat = malloc(mat_b);
  HIP_OK(hipMemcpyAsync(ok, d_ok, ok_b, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(crop, d_crop, crop_b, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(mat, d_mat, mat_b, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  int bad = 0;
  size_t face_b = (size_t)oh * ow * 3, live = 0;
  for (int i = 0; i < f; ++i) {
    if (ok[i] != ok_ref[i]) {
      fprintf(stderr, "face %d: ok %d, expected %d\n", i, ok[i], ok_ref[i]);
      ++bad;
      continue;
    }
    if (!ok[i]) continue; /* dropped by the caller (cropper.py:529-531): matrix and crop are unspecified */
    ++live;
    for (int j = 0; j < 6; ++j) {
      const double a = mat[i * 6 + j], b = mat_ref[i * 6 + j];
      if (!(fabs(a - b) <= 1e-12 * fmax(1.0, fabs(b)))) {
        fprintf(stderr, "face %d: mat[%d] = %.17g, expected %.17g\n", i, j, a, b);
        ++bad;
      }
    }
    if (memcmp(crop + i * face_b, crop_ref + i * face_b, face_b) != 0) {
      fprintf(stderr, "face %d: crop bytes differ\n", i);
      ++bad;
    }
  }
  /* error contract: misuse returns a negative code and leaves a message, nothing is launched */
  const int rc = fcp_warp_affine_u8(d_img, n, h, w, d_idx, d_mat, d_ok, NULL, f, oh, ow, 99, d_crop, stream);
  if (rc >= 0 || fcp_last_error() == NULL || fcp_last_error()[0] == '\0') {
    fprintf(stderr, "border=99 was accepted (rc %d)\n", rc);
    ++bad;
  }
  hipFree(d_img); hipFree(d_idx); hipFree(d_lm); hipFree(d_tgt); hipFree(d_pad); hipFree(d_ok); hipFree(d_crop); hipFree(d_mat);
  hipStreamDestroy(stream);
  if (bad) {
    printf("FAILED %d checks\n", bad);
    return 1;
  }
  printf("OK faces=%d live=%zu crop_bytes=%zu\n", f, live, live * face_b);
  return 0;
}
