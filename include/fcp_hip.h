/*
 * fcp_hip.h — C ABI of the MI355X (gfx950) hot path of face-crop-plus.
 *
 * The reference (mantasu/face-crop-plus) is pure Python: it has no FFI layer.
 * Its seam for this path is the object protocol `Cropper` uses on its three
 * models plus `crop_align`.  This header is the native side of
 * that seam: stateless entry points, plain pointers and sizes, every pointer a
 * *device* pointer unless the name ends in `_host`.  All work is enqueued on
 * the caller's HIP stream (`stream` is a hipStream_t passed as void*); nothing
 * synchronises, nothing allocates, nothing keeps global state except the
 * thread-local last-error string.  Every function returns 0 on success and a
 * negative code on misuse (message via fcp_last_error()).
 *
 * Each entry point cites the reference code it replaces (paths relative to
 * src/face_crop_plus/ of the reference tree).
 */
#ifndef FCP_HIP_H
#define FCP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCP_ABI_VERSION 15

typedef void* fcp_stream_t; /* hipStream_t */

int fcp_abi_version(void);
const char* fcp_last_error(void);

/* ------------------------------------------------------------------------
 * Convolution engine (NHWC fp32 tensors, implicit GEMM on the matrix cores).
 * Two arithmetic modes, chosen per filter at pack time (`precision`):
 *   0  exact fp32:  v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain;
 *   1  fp16x3 split: every operand x = hi + lo (two binary16), product expanded as
 *      ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
 *      ~2^-20 relative error per product (fp32-roundoff class) at 5.3x the matrix rate.
 *      Filter rows are pre-scaled by a power of two (`wscale` undoes it exactly).
 * Replaces every nn.Conv2d(+BatchNorm eval)(+ReLU/LeakyReLU)(+residual) the
 * three networks dispatch to ATen: models/_layers.py:64-162 (SSH/FPN/Head),
 * :168-200 (RRDB blocks), :206-368 (BiSeNet blocks), torchvision ResNet-50
 * body (models/retinaface.py:93-99).
 *
 * Tensor layout: activations are NHWC fp32.  `in`/`out`/`res*` may point into
 * a wider buffer: `*_ld` is the channel stride (floats per pixel) of that
 * buffer and the pointer is already offset to the first channel used.  This
 * is how torch.cat (SSH, dense blocks, FFM) is realised with no copy.
 *
 * Filter layout (produced by the host packer, see engine.py pack_conv):
 *   normal mode : [cout_pad][cin/32][kh][kw][32]     cin % 32 == 0 (taps of one 32-channel
 *                                                    slice are consecutive in K: L2 locality)
 *   cin4 mode   : [cout_pad][kh][8][4]               cin <= 4, kw <= 8
 * precision 0 stores fp32; precision 1 stores, for every 32 consecutive K values of a
 * row, 32 binary16 hi parts followed by 32 binary16 lo parts (same 128 bytes).
 * cout_pad = cout rounded up to a multiple of 128 (whatever N tile a launch uses);
 * the padding rows are zero.  BatchNorm (eval) is folded into w and bias.
 *
 * Epilogue (fp32, in this order):
 *   v = acc * wscale[co] + bias[co]        (wscale = 1 for precision 0)
 *   if (res1 && res1_pre)  v += res1[...]
 *   v = v >= 0 ? v : v * act_slope        (act_slope = 1 -> identity, 0 -> ReLU)
 *   v = v * alpha
 *   if (res1 && !res1_pre) v += res1[...]
 *   if (res2)              v = v * alpha2 + res2[...]
 * res1 may have a different spatial size (res1_h,res1_w): it is then read with
 * PyTorch's nearest rule  src = min(floor(dst * (float)src_size/dst_size), src_size-1)
 * (FPN top-down add, _layers.py:137-143).  res2 always has the output's size.
 * ------------------------------------------------------------------------ */
typedef struct fcp_conv_desc {
  const float* in;    /* (n, in_h_phys, in_w_phys, in_ld) */
  const void* w;      /* packed filter */
  const float* bias;  /* [cout] or NULL */
  float* out;         /* (n, out_h, out_w, out_ld) */
  const float* res1;  /* or NULL */
  const float* res2;  /* or NULL */
  const float* wscale; /* [cout] power-of-two filter scales (precision 1) or NULL */
  const float* in2;   /* optional second source of a 1x1 conv (see in2_* below) or NULL */
  int32_t n, in_h, in_w; /* logical input size (after the optional x2 upsample) */
  int32_t cin, in_ld;
  int32_t in_up2;     /* 1: physical input is (in_h/2, in_w/2), read at (h>>1, w>>1)
                         = F.interpolate(scale_factor=2 / exact-2x size, "nearest")
                         (rrdb.py:78-79, _layers.py:338,:343) */
  int32_t cout, kh, kw, stride, pad;
  int32_t out_h, out_w, out_ld;
  int32_t tile_n;     /* N tile: 32, 64 or 128 (128, 192 or 256 with tile_m = 256); filters are padded to 128 rows */
  int32_t cin4;       /* 1: cin4 mode */
  float act_slope, alpha, alpha2;
  int32_t res1_pre, res1_ld, res1_h, res1_w, res2_ld;
  int32_t precision;  /* 0 = fp32 exact, 1 = fp16x3 split (filter packed accordingly) */
  /* Activation tensor formats (precision 1 only): 0 = fp32 NHWC; 1 = "split32": same bytes per
   * element and the same strides, but every group of 32 channels of a pixel is stored as 32 binary16
   * hi parts followed by 32 binary16 lo parts (value = hi + lo, ~22 significant bits).  A K slice of
   * a split32 tensor is byte-for-byte the kernel's LDS operand image, so the consumer conv copies it
   * instead of converting; producers convert once in their epilogue.  Views must start on a
   * 32-channel group and in_ld / out_ld / res*_ld must be multiples of 32. */
  int32_t in_fmt, out_fmt, res1_fmt, res2_fmt;
  int32_t tile_m;     /* 0 / 128: 128-row workgroup tiles; 256: the 256-row, 8-wave kernel (precision 1,
                         split32 input, no cin4 / in_up2; tile_n 128, 192 or 256); 1: the halo-tile kernels
                         for 3x3 / stride 1 / pad 1 convs (8 x 32 pixel patches, precision 1, split32 input,
                         cin >= 64, cout % 8 == 0, wscale / bias 16-byte aligned, |out view| < 4 GiB):
                         tile_n 32 = one pass per 32 filters, cout <= 64; tile_n 64 / 128 = the wide form
                         (column tiles inner, filters through a tap ring): cin % 64 == 0, cout <= 64, or
                         cout <= 128 without residual inputs */
  /* Two-source 1x1 conv (K concatenation): the trailing cin2 of the cin input channels come from
   * in2, a split32 tensor (n, in2_h, in2_w, in2_ld) sampled at (ho*in2_stride, wo*in2_stride); the
   * leading cin - cin2 channels come from `in` as usual.  This is how a ResNet bottleneck's
   * bn3(conv3(o)) + bn_d(downsample(x)) runs as one convolution over [o | x(::s, ::s)] without ever
   * materialising the downsampled identity.  Needs kh = kw = 1, pad 0, precision 1, split32 `in`. */
  int32_t cin2, in2_ld, in2_h, in2_w, in2_stride;
  /* FCP_CONV_FLAT_ADDR (precision 0 only): use 64-bit flat addressing even when every tensor is below
   * 4 GiB.  The flat path is what tensors >= 4 GiB take on their own (RRDB's x4-resolution tail at 1024^2
   * inputs); the flag exists so that it can be exercised — and tested — at small sizes. */
  int32_t flags;
  /* FCP_CONV_BALANCE_TAIL (256-row tiles, tile_m = 256): cut the rows of the last, partial dispatch round into
   * shorter M-tiles that together occupy every CU (instead of 256-row tiles on a fraction of them).  `cu_budget` =
   * CUs the launch may count on (0 = all of the device; a caller that runs two such launches concurrently on two
   * streams passes half).  Neither changes a result. */
  int32_t cu_budget;
  /* Row bands (stride 1): the input view holds band_top real rows above and band_bottom real rows below the rows the
   * output view covers (0 <= each <= pad), instead of the zero padding a conv applies at the edge of its view:
   *     out_h = in_h - band_top - band_bottom + 2 pad - kh + 1,    input row of (output row ho, tap kh_i) = ho - pad + band_top + kh_i.
   * This is how a caller computes rows [a, b) of a larger image's convolution exactly: `in` = rows [a - band_top, b + band_bottom)
   * with band_top = 0 only at the image's top edge (where the zero padding is the right one).  RRDB's dense blocks run
   * band-major this way (rrdb.py: conv1..conv5 of a block on one row band before the next band, the growing concat
   * staying in the memory-side cache; _layers.py:168-200 of the reference).  Same arithmetic per output pixel: bit-identical to
   * the whole-image launch.  Both 0: the ordinary convolution. */
  int32_t band_top, band_bottom;
} fcp_conv_desc;

#define FCP_CONV_FLAT_ADDR 1
#define FCP_CONV_BALANCE_TAIL 2

int fcp_conv2d_nhwc_f32(const fcp_conv_desc* desc, fcp_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused bottleneck chain (precision 1, split32 tensors): conv2 of ResNet block b, conv3 of block b
 * with the identity residual, and conv1 of block b + 1, in ONE launch:
 *     t2  = relu(conv2_3x3(t1) * ws2 + b2)                   c  -> c    (stays in LDS)
 *     out = relu(conv3_1x1(t2) * ws3 + b3 + res)             c  -> 4c   (written once)
 *     t1n = relu(conv1n_1x1(out) * ws1n + b1n)               4c -> cn
 * Replaces torchvision's Bottleneck.forward (the ResNet-50 body retinaface.py:93-99 wraps) from conv2 of one
 * identity block to conv1 of the next: the 4c-channel tensor crosses HBM twice (residual read, result write)
 * instead of four times and the c-channel intermediate never leaves the workgroup.  Results are bit-identical
 * to three fcp_conv2d_nhwc_f32 launches.  Filters are ordinary precision-1 packs (pack_conv): w2
 * [128][9c] (3x3 / stride 1 / pad 1), w3 [4c][c], w1n [cn padded to 128][4c]; BatchNorm folded, so every conv
 * has a bias and a per-filter scale.  All tensors are split32 views (n, h, w, *_ld), 128-byte aligned,
 * *_ld % 32 == 0.  Supported: c = 64 (ResNet-50 layer 1), nout = 256, cn = 64 or 128.
 *
 * Pair forms (w2 == NULL: no conv2, t1 is conv3's input itself, c = 128 or 256 channels):
 *     out = relu(conv3_1x1(t1) * ws3 + b3 [+ res])    c -> nout;     t1n = relu(conv1n_1x1(out) * ws1n + b1n)    nout -> cn
 *   c = 128, nout = 512,  res != NULL, cn = 128   conv3 of a layer-2 identity block + conv1 of the next block
 *   c = 128, nout = 512,  res != NULL, cn = 256   layer 2's last block + layer3.0.conv1
 *   c = 256, nout = 1024, res != NULL, cn = 256   conv3 of a layer-3 identity block + conv1 of the next block
 *   c = 128, nout = 256,  res == NULL, cn = 64    layer1.0: conv3 + downsample as one K-concatenated 1x1 conv over
 *                                                 [conv2 out | pooled stem] (engine.py packs it so), + layer1.1.conv1
 *   c = 384, nout = 512,  res == NULL, cn = 128,  t1b != NULL, cb = 256 (round 5)
 *                                                 layer2.0: conv3 + the 1x1 / 2 downsample as one K-concatenated conv over the TWO
 *                                                 sources [conv2 out (t1: 128 ch) | x(::2, ::2) (t1b: 256 ch at twice the
 *                                                 resolution)], + layer2.1.conv1: the 512-channel block output is written once
 *                                                 and not read back by a separate conv1 launch.  The operand fragments are
 *                                                 loaded straight from the two tensors (no LDS tile).
 *   c = 256, nout = 1024, res != NULL, cn = 0,    w1n == ws1n == b1n == t1n == NULL (round 5): the EXPAND form — conv3 + identity of
 *                                                 a layer-3 block alone, out = relu(conv3(t1) * ws3 + b3 + res), no conv1';
 *                                                 fragments from global memory, two workgroups per CU; bit-identical to
 *                                                 fcp_conv2d_nhwc_f32 with the fused residual epilogue (a tie with it and with the
 *                                                 pair in the A/B: opt-in).
 * ------------------------------------------------------------------------ */
typedef struct fcp_chain_desc {
  const float* t1;    /* conv2's input: c channels */
  const void* w2;  const float* ws2;  const float* b2;
  const void* w3;  const float* ws3;  const float* b3;
  const float* res;   /* identity branch x: 4c channels */
  float* out;         /* 4c channels */
  const void* w1n; const float* ws1n; const float* b1n;
  float* t1n;         /* cn channels */
  const float* t1b;   /* two-source pair form: the trailing cb of conv3's c input channels come from this split32 tensor
                       * (n, t1b_h, t1b_w, t1b_ld), sampled at (y * t1b_stride, x * t1b_stride); t1 holds the leading c - cb.
                       * NULL: one source */
  int32_t n, h, w, c, cn;
  int32_t t1_ld, res_ld, out_ld, t1n_ld;
  int32_t nout;       /* conv3's filters: 4c with conv2; see the pair forms below */
  int32_t tile_m;     /* pixels per workgroup tile: 0 / 128 = 4-wave tiles of 128 consecutive pixels (two workgroups per
                       * CU), 256 = 8-wave tiles where the operand tile fits LDS (the other forms keep 128), 16 = (conv2
                       * forms only) 4-wave tiles that are 8 x 16 pixel patches of one image, conv2's operand staged
                       * once per channel slice as the patch's halo; 32 = the same on 8 waves and 16 x 16 patches (one
                       * workgroup per CU).  Same bits whichever is chosen. */
  int32_t flags;      /* FCP_CHAIN_OUT_EVEN_ONLY (patch form, tile_m = 16): `out` is stored at pixels with even y AND
                       * even x only — for a block whose output is read by nothing but a stride-2 consumer (ResNet-50's
                       * layer1.2: the next block's 1x1 / 2 downsample reads `out`, its conv1 is t1n, computed here): three
                       * quarters of the tensor are never written nor read.  The other pixels of the buffer keep whatever
                       * they held.  t1n is complete either way. */
  int32_t cb, t1b_ld, t1b_h, t1b_w, t1b_stride;   /* geometry of t1b (all 0 without it) */
} fcp_chain_desc;
#define FCP_CHAIN_OUT_EVEN_ONLY 1

int fcp_bottleneck_chain_f16x3(const fcp_chain_desc* desc, fcp_stream_t stream);

/* uint8 NHWC RGB (n,h,w,3) -> fp32 NHWC4 (n,h,w,4): out[c] = (in[c]-sub[c])/div, out[3]=0.
 * Replaces utils.py:222-224 (as_tensor) fused with retinaface.py:450-451 (mean
 * subtraction; the BGR swap is folded into the stem filter's channel order) or
 * with rrdb.py:142 (`.div(255)`).  sub_host: 3 floats on the host. */
int fcp_u8_to_nhwc4_f32(const uint8_t* in, float* out, int64_t npix,
                        const float* sub_host, float div, fcp_stream_t stream);

/* Same, from the reference API's float tensor: fp32 NCHW (n,3,h,w) -> fp32 NHWC4
 * (RetinaFace.predict / RRDBNet.predict take float NCHW, retinaface.py:411, rrdb.py:84). */
int fcp_f32nchw_to_nhwc4_f32(const float* in, float* out, int n, int h, int w,
                             const float* sub_host, float div, fcp_stream_t stream);

/* MaxPool2d(kernel 3, stride 2, pad 1) on NHWC fp32 (torchvision ResNet stem,
 * _layers.py:247).  c % 4 == 0.  The input is dense (c elements per pixel); the output may be a
 * channel slice of a wider buffer (out_ld elements per pixel, out_ld >= c). */
int fcp_maxpool3x3s2_nhwc_f32(const float* in, float* out, int n, int h, int w, int c, int out_ld,
                              int out_h, int out_w, fcp_stream_t stream);
/* Same on split32 tensors (c, out_ld % 32 == 0); the maximum is exact (hi + lo decodes exactly). */
int fcp_maxpool3x3s2_split32(const float* in, float* out, int n, int h, int w, int c, int out_ld,
                             int out_h, int out_w, fcp_stream_t stream);
/* RetinaFace stem in one launch (precision 1): uint8 RGB image -> (x - mean_rgb) -> 7x7 / stride 2 / pad 3
 * conv to 64 channels (BatchNorm folded into wfrag / bias) -> ReLU -> MaxPool2d(3, 2, 1), written as a 64-channel
 * slice (out_ld elements per pixel) of an fp32 (out_fmt 0) or split32 (1) NHWC tensor of size
 * (n, hp, wp) with hs = (h-1)/2+1, hp = (hs-1)/2+1.  Replaces retinaface.py:450-451 + the torchvision ResNet
 * stem (conv1, bn1, relu, maxpool; retinaface.py:93-99).  x - mean is integral, hence exact in binary16: the
 * fp16x3 product needs only a*wh + a*wl.  wfrag: the filter split hi / lo in MFMA fragment order
 * [2 column tiles][11 k-steps][hi, lo][64 lanes][8 binary16], K index = kh*24 + kw*3 + c (RGB), zero-padded
 * (engine.py::pack_stem_fused); wscale: the per-filter power-of-two scale.  mean_rgb: 3 integers in 0..255 (host). */
int fcp_stem7x7s2_relu_pool_u8(const uint8_t* images, int n, int h, int w, const int32_t* mean_rgb,
                               const void* wfrag, const float* bias, const float* wscale, float* out,
                               int out_ld, int out_fmt, fcp_stream_t stream);
/* The same launch continued by conv1 of the first bottleneck (torchvision Bottleneck.forward: relu(bn1(conv1(x))),
 * a 1x1 conv 64 -> 64 on the pooled map; retinaface.py:93-99): t1 = relu(conv1x1(pooled) * ws1 + b1), written as a
 * split32 tensor (n, hp, wp, t1_ld).  The pooled map is still written (the block's downsample branch reads it) but not
 * read back: its hi / lo bytes are conv1's operand while they are still in LDS.  w1: the conv's packed precision-1
 * filter (engine.py::pack_conv: >= 64 rows of [2 channel slices][32 hi | 32 lo binary16]); out_fmt must be 1.  Same K
 * and term order as fcp_conv2d_nhwc_f32 on the stored map: bit-identical to the two launches.  w1 == NULL: the plain stem. */
int fcp_stem7x7s2_relu_pool_conv1_u8(const uint8_t* images, int n, int h, int w, const int32_t* mean_rgb,
                                     const void* wfrag, const float* bias, const float* wscale, float* out,
                                     int out_ld, int out_fmt, const void* w1, const float* ws1, const float* b1,
                                     float* t1, int t1_ld, fcp_stream_t stream);
/* The same stem on an fp32 NHWC4 input (n, h, w, 4; channel 3 ignored) that is already normalised: BiSeNet's ResNet-18 stem
 * (conv1 7x7 / 2 + bn1 + relu + maxpool 3x3 / 2; bise.py:387-393 feeds it, _layers.py:241-247 is the stem).  The input has a lo part,
 * so the patch is staged as hi + lo binary16 planes and a k-step is the full three-term product (al*wh + ah*wl + ah*wh).  wfrag /
 * bias / wscale: engine.py::pack_stem_fused (no channel permutation).  The 256 x 256 x 64 stem map of a 512 x 512 face (537 MB for 32
 * faces) never reaches HBM. */
int fcp_stem7x7s2_relu_pool_f32(const float* x4, int n, int h, int w, const void* wfrag, const float* bias,
                                const float* wscale, float* out, int out_ld, int out_fmt, fcp_stream_t stream);
/* Format converters between fp32 NHWC and split32 (npix pixels of c channels, c % 32 == 0). */
int fcp_f32_to_split32(const float* in, float* out, int64_t npix, int c, fcp_stream_t stream);
int fcp_split32_to_f32(const float* in, float* out, int64_t npix, int c, fcp_stream_t stream);
/* Range guard of the fp16x3 path (no reference counterpart: the reference computes in fp32, _layers.py:16-35 hands it
 * arbitrary checkpoints): *out_max = max(*out_max, max |x|) over a channel-slice view of npix pixels x c channels
 * (pixel pitch ld elements; fmt 0 fp32 / 1 split32; c % 8 == 0), NaN counted as +inf.  *out_max must be initialised
 * (>= 0) by the caller; the update is an atomic max on the device. */
int fcp_absmax_nhwc(const float* x, int64_t npix, int c, int ld, int fmt, float* out_max, fcp_stream_t stream);

/* ------------------------------------------------------------------------
 * RetinaFace post-processing.
 * ------------------------------------------------------------------------ */

/* Fused softmax + analytic PriorBox + decode + strict threshold + ordered
 * compaction.  Replaces retinaface.py:144 (softmax), _layers.py:41-62
 * (PriorBox), retinaface.py:169-178/:204-210/:455-461 (decode + scale) and
 * retinaface.py:264-267 (score > vis mask + gather).
 *
 * head[l]: fused head conv output of pyramid level l (stride 8/16/32), NHWC
 * with 32 channels per pixel: [cls a0(bg,face) a1(bg,face) | box a0(4) a1(4) |
 * landm a0(10) a1(10)]; level sizes are ceil(h/stride) x ceil(w/stride).
 * Outputs (per image, capacity P = total priors, candidates in ascending
 * prior order): cand_score (n,P), cand_box (n,P,4), cand_ldm (n,P,10),
 * cand_prior (n,P) int32, cand_count (n) int32.
 * dense_* are optional (NULL to skip): full (n,P) / (n,P,4) / (n,P,10). */
int fcp_retina_decode(const float* head0, const float* head1, const float* head2,
                      int n, int img_h, int img_w, float vis_threshold,
                      float var0, float var1,
                      float* cand_score, float* cand_box, float* cand_ldm,
                      int32_t* cand_prior, int32_t* cand_count,
                      float* dense_score, float* dense_box, float* dense_ldm,
                      fcp_stream_t stream);

/* Per-image sort (score desc, candidate position asc) + greedy NMS with the
 * reference's +1-pixel IoU and `ovr <= nms_threshold` survival rule
 * (retinaface.py:270-298), followed by take_by_strategy (retinaface.py:363-408).
 * strategy: 0 = all, 1 = best, 2 = largest.
 * workspace: fcp_retina_nms_workspace_bytes(n, cap) bytes (sort keys + sorted
 * boxes), cap = candidate capacity (stride of the cand_* arrays, <= 507904).  Outputs: keep_pos (n,cap) int32 = candidate positions of
 * the kept boxes in keep order, keep_count (n); sel_pos (n,cap) / sel_count (n)
 * = the positions take_by_strategy selects (for "all" identical to keep). */
int64_t fcp_retina_nms_workspace_bytes(int n, int cap);
int fcp_retina_nms_select(const float* cand_score, const float* cand_box,
                          const int32_t* cand_count, int n, int cap,
                          float nms_threshold, int strategy, void* workspace,
                          int32_t* keep_pos, int32_t* keep_count,
                          int32_t* sel_pos, int32_t* sel_count, fcp_stream_t stream);

/* Gather the selected faces into dense, image-major arrays (the return value
 * of RetinaFace.predict, retinaface.py:465-470, minus the D2H copy) and apply
 * the landmark un-padding of cropper.py:822 (paddings (n,4) int32 t,b,l,r or NULL).
 * face_offset: (n+1) int32 exclusive prefix of sel_count (written here;
 * face_offset[n] = number of faces).  out_ldm (max_faces,5,2) f32, out_img
 * (max_faces) int32: rows >= face_offset[n] are zeroed here (no memset needed). */
int fcp_retina_gather_faces(const float* cand_ldm, const int32_t* sel_pos,
                            const int32_t* sel_count, int n, int cap,
                            const int32_t* paddings, int max_faces,
                            int32_t* face_offset, float* out_ldm, int32_t* out_img,
                            fcp_stream_t stream);

/* ------------------------------------------------------------------------
 * Align + crop.
 * ------------------------------------------------------------------------ */

/* Least-squares 2x3 transform from 5 (or k) source points to target points:
 * similarity (4 dof) = cv2.estimateAffinePartial2D, full affine (6 dof) =
 * cv2.estimateAffine2D, both with ransacReprojThreshold=inf (cropper.py:515-527).
 * src (f,k,2) f32, dst (k,2) f32 -> mat (f,6) f64 row-major, ok (f) int32
 * (0 = degenerate / non-finite: the reference drops that face, cropper.py:529-531). */
int fcp_estimate_transform(const float* src, const float* dst, int f, int k,
                           int allow_skew, double* mat, int32_t* ok, fcp_stream_t stream);

/* The same for a fixed-capacity face array whose live length is on the device
 * (the detector's face_offset[n]; RetinaFace.predict's `len(landmarks)`,
 * retinaface.py:465-470): rows >= *face_count (device int32, or NULL = all f)
 * get ok = 0 and a zero matrix, and the number of faces with ok = 1 — the faces
 * crop_align keeps, cropper.py:529-531 — is ADDED to *valid_total (device
 * int64 accumulator, or NULL).  No host read-back, no extra launches. */
int fcp_estimate_transform_counted(const float* src, const float* dst, int f, int k,
                                   int allow_skew, const int32_t* face_count,
                                   double* mat, int32_t* ok, int64_t* valid_total,
                                   fcp_stream_t stream);

/* cv2.warpAffine(image, M, dsize, flags=INTER_LINEAR, borderMode) on the
 * un-padded slice of each face's batch image (cropper.py:533-547): OpenCV's
 * fixed-point algorithm (AB_BITS=10, INTER_BITS=5, 15-bit weights).
 * images (n,h,w,3) u8; img_idx (f) int32; mat (f,6) f64 forward transforms;
 * paddings (n,4) int32 (t,b,l,r) or NULL; border: 0 constant(0), 1 replicate,
 * 2 reflect, 3 wrap, 4 reflect_101 (= cv2.BORDER_*); out (f,out_h,out_w,3) u8. */
int fcp_warp_affine_u8(const uint8_t* images, int n, int h, int w,
                       const int32_t* img_idx, const double* mat, const int32_t* ok,
                       const int32_t* paddings, int f, int out_h, int out_w, int border,
                       uint8_t* out, fcp_stream_t stream);

/* ------------------------------------------------------------------------
 * BiSeNet face parser glue (models/bise.py, _layers.py:206-368).
 * ------------------------------------------------------------------------ */

/* bise.py:387-393: faces (f,h,w,3) uint8 RGB -> `/255`, bilinear to (out_h,out_w)
 * with align_corners=False, `(x-mean)/std` -> fp32 NHWC4 (f,out_h,out_w,4).
 * mean_host/std_host: 3 floats each on the host. */
int fcp_bise_preprocess_u8(const uint8_t* faces, int f, int h, int w, float* out,
                           int out_h, int out_w, const float* mean_host,
                           const float* std_host, fcp_stream_t stream);

/* F.avg_pool2d(x, x.size()[2:]) on an NHWC slice (_layers.py:307,:332,:360):
 * in (n,hw,ld) channels [0,c) -> out (n,c). */
int fcp_avgpool_nhwc_f32(const float* in, int n, int hw, int c, int ld, float* out,
                         fcp_stream_t stream);

/* 1x1 conv on a 1x1 map (+ folded BatchNorm) (+ ReLU / sigmoid): ARM conv_atten,
 * ContextPath conv_avg, FFM conv1/conv2 (_layers.py:308-310,:333,:361-364).
 * out[n][co] = act(scale[co] * dot(w[co,:], in[n,:]) + shift[co]);
 * scale/shift may be NULL; act: 0 none, 1 relu, 2 sigmoid. */
int fcp_fc_f32(const float* in, const float* w, const float* scale, const float* shift,
               int n, int cin, int cout, int act, float* out, fcp_stream_t stream);

/* out = x * scale_nc[n,c] (+ add_nc[n,c]) (+ add_t[n,h,w,c]): torch.mul(feat, atten)
 * followed by `+ avg_up` / `+ feat32_up` / `+ feat` (_layers.py:311,:337,:342,:365-366). */
int fcp_scale_add_nhwc_f32(const float* x, int x_ld, const float* scale_nc,
                           const float* add_nc, const float* add_t, int add_t_ld,
                           int n, int hw, int c, float* out, int out_ld, fcp_stream_t stream);

/* Parse tail (bise.py:212 + :394): logits (f,lh,lw,ld) at 1/8 resolution ->
 * bilinear align_corners=True to (mid_h,mid_w) -> nearest to (out_h,out_w) ->
 * argmax (first maximum) -> labels (f,out_h,out_w) uint8.  counts (f,ncls) int32
 * = per-face class histogram (bise.py:253-254), may be NULL. */
int fcp_parse_tail(const float* logits, int f, int lh, int lw, int ld, int ncls,
                   int mid_h, int mid_w, int out_h, int out_w, uint8_t* labels,
                   int32_t* counts, fcp_stream_t stream);

/* mask = 255 where bit `label` of class_bits is set, else 0 (bise.py:314-319). */
int fcp_label_mask_u8(const uint8_t* labels, int64_t total, uint32_t class_bits,
                      uint8_t* mask, fcp_stream_t stream);

/* ------------------------------------------------------------------------
 * RRDB enhancer tail (rrdb.py:143-144): x4 (4h,4w,ld>=3) fp32 in [0,1] ->
 * bicubic x0.25 (align_corners=False, A=-0.75: taps (-3,19,19,-3)/32) ->
 * clamp(0,1)*255 -> round-half-even -> uint8 RGB (h,w,3).
 * ------------------------------------------------------------------------ */
int fcp_bicubic_down4_u8(const float* x4, int h, int w, int ld, uint8_t* out_rgb,
                         fcp_stream_t stream);

/* ------------------------------------------------------------------------
 * Batch builder (utils.py:273-342, `as_batch`): cv2.resize (INTER_AREA when the
 * image is larger than the batch, else INTER_CUBIC; uint8) of every image of a
 * ragged list + cv2.copyMakeBorder into its slot of one (n,out_h,out_w,3) uint8
 * batch, one launch.  The host computes the geometry of utils.py:316-331 and
 * fills one item per image; the images are packed back to back (RGB, HWC) in
 * one device blob.  items_host is validated here; items_dev is the caller's
 * device copy of the same table.  border = cv2.BORDER_* code as for
 * fcp_warp_affine_u8 (the reference always passes "constant", utils.py:276).
 * ------------------------------------------------------------------------ */
typedef struct fcp_batch_item {
  int64_t src_off;    /* byte offset of the (sh,sw,3) image inside src_blob */
  int32_t sh, sw;     /* source height, width */
  int32_t dh, dw;     /* resized height, width (hh, ww of utils.py:322-331) */
  int32_t top, left;  /* padding in front of the resized image (paddings[0], paddings[2]) */
  int32_t interp;     /* 0 = cv2.INTER_CUBIC, 1 = cv2.INTER_AREA (decimation only) */
  int32_t reserved;
} fcp_batch_item;

int fcp_build_batch_u8(const uint8_t* src_blob, int64_t blob_bytes,
                       const fcp_batch_item* items_host, const fcp_batch_item* items_dev,
                       int n, int out_h, int out_w, int border, uint8_t* out,
                       fcp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FCP_HIP_H */
