// PyTorch-ROCm custom ops over the C ABI of include/fcp_hip.h ("a PyTorch-ROCm extension, TORCH_LIBRARY
// namespace ... stateless ops on the current HIP stream, tensors in/out, no hidden global state").
//
//     torch.ops.fcp.conv2d / bottleneck_chain / retina_decode / nms_select / gather_faces / similarity_from_5pt /
//     warp_affine_u8 / bicubic_down4_round / parse_argmax_hist
//
// Each op validates device / dtype / contiguity with TORCH_CHECK (-> RuntimeError), allocates its outputs with torch's
// caching allocator, borrows its inputs, enqueues the HIP kernels of libfcp_hip.so on at::hip's CURRENT stream and
// returns without synchronising: the ops are stream-ordered and safe to capture in a HIP graph.  Schemas are plain
// (Tensor / int / float / bool), so torch.library can attach fake-tensor implementations for torch.compile.
// The ops are a veneer: all arithmetic lives behind the C ABI, which stays the drop-in boundary.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <tuple>

#include "fcp_hip.h"

namespace {

using at::Tensor;

void* cur_stream() { return (void*)at::hip::getCurrentHIPStream().stream(); }

// Every op makes its first tensor's device current for its duration: the stream it enqueues on and the memory its
// outputs come from are that device's, whatever device the calling thread had selected.
#define FCP_DEVICE_GUARD(t) const c10::OptionalDeviceGuard fcp_guard_(at::device_of(t))

void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", fcp_last_error()); }

const Tensor& dev(const Tensor& t, const char* name, at::ScalarType dt) {
  TORCH_CHECK(t.is_cuda(), name, " must live on the GPU");
  TORCH_CHECK(t.scalar_type() == dt, name, " has dtype ", t.scalar_type(), ", expected ", dt);
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  return t;
}
template <typename T>
const T* optp(const c10::optional<Tensor>& t, const char* name, at::ScalarType dt) {
  if (!t.has_value() || !t->defined()) return nullptr;
  return dev(*t, name, dt).data_ptr<T>();
}

// ---- conv engine.  x / res1 / res2 / out are NHWC buffers (n, h, w, ld); *_c0 selects the first channel of the view.
// `out` (optional) lets the caller write a channel slice of a wider buffer (torch.cat without a copy); otherwise a
// dense (n, oh, ow, cout) tensor is allocated.
Tensor conv2d_impl(const Tensor& x, int64_t x_c0, int64_t cin, const Tensor& w, const c10::optional<Tensor>& bias,
                   const c10::optional<Tensor>& wscale, const c10::optional<Tensor>& res1, int64_t res1_c0,
                   const c10::optional<Tensor>& res2, int64_t res2_c0, const c10::optional<Tensor>& out_, int64_t out_c0,
                   int64_t cout, int64_t kh, int64_t kw, int64_t stride, int64_t pad, double act_slope, double alpha,
                   double alpha2, bool res1_pre, int64_t precision, int64_t in_fmt, int64_t out_fmt, int64_t res1_fmt,
                   int64_t res2_fmt, bool in_up2, bool cin4, int64_t tile_m, int64_t tile_n, const c10::optional<Tensor>& x2,
                   int64_t x2_c0, int64_t cin2, int64_t x2_stride, int64_t flags, int64_t cu_budget, int64_t band_top,
                   int64_t band_bottom) {
  dev(x, "x", at::kFloat);
  FCP_DEVICE_GUARD(x);
  TORCH_CHECK(x.dim() == 4, "x must be NHWC (n, h, w, ld)");
  TORCH_CHECK(w.is_cuda() && w.is_contiguous(), "w must be a contiguous GPU tensor (packed filter)");
  const int64_t n = x.size(0), ph = x.size(1), pw = x.size(2), ld = x.size(3);
  const int64_t ih = in_up2 ? 2 * ph : ph, iw = in_up2 ? 2 * pw : pw;
  const int64_t oh = (ih - band_top - band_bottom + 2 * pad - kh) / stride + 1, ow = (iw + 2 * pad - kw) / stride + 1;
  Tensor out = out_.has_value() && out_->defined() ? *out_ : at::empty({n, oh, ow, cout}, x.options());
  dev(out, "out", at::kFloat);
  TORCH_CHECK(out.dim() == 4 && out.size(0) == n && out.size(1) == oh && out.size(2) == ow && out_c0 + cout <= out.size(3),
              "out has the wrong geometry");
  TORCH_CHECK(x_c0 + (cin4 ? 4 : cin - cin2) <= ld, "input channel view exceeds the buffer");
  fcp_conv_desc d = {};
  d.in = x.data_ptr<float>() + x_c0;
  d.w = w.data_ptr();
  d.bias = optp<float>(bias, "bias", at::kFloat);
  d.wscale = optp<float>(wscale, "wscale", at::kFloat);
  d.out = out.data_ptr<float>() + out_c0;
  d.n = (int)n; d.in_h = (int)ih; d.in_w = (int)iw; d.cin = (int)cin; d.in_ld = (int)ld; d.in_up2 = in_up2;
  d.cout = (int)cout; d.kh = (int)kh; d.kw = (int)kw; d.stride = (int)stride; d.pad = (int)pad;
  d.out_h = (int)oh; d.out_w = (int)ow; d.out_ld = (int)out.size(3);
  d.tile_n = (int)tile_n; d.tile_m = (int)tile_m; d.cin4 = cin4;
  d.act_slope = (float)act_slope; d.alpha = (float)alpha; d.alpha2 = (float)alpha2;
  d.res1_pre = res1_pre; d.precision = (int)precision;
  d.in_fmt = (int)in_fmt; d.out_fmt = (int)out_fmt; d.res1_fmt = (int)res1_fmt; d.res2_fmt = (int)res2_fmt;
  if (res1.has_value() && res1->defined()) {
    dev(*res1, "res1", at::kFloat);
    d.res1 = res1->data_ptr<float>() + res1_c0;
    d.res1_ld = (int)res1->size(3); d.res1_h = (int)res1->size(1); d.res1_w = (int)res1->size(2);
  }
  if (res2.has_value() && res2->defined()) {
    dev(*res2, "res2", at::kFloat);
    d.res2 = res2->data_ptr<float>() + res2_c0;
    d.res2_ld = (int)res2->size(3);
  }
  if (x2.has_value() && x2->defined()) {                 // second source of a 1x1 conv (K concatenation)
    dev(*x2, "x2", at::kFloat);
    d.in2 = x2->data_ptr<float>() + x2_c0;
    d.cin2 = (int)cin2; d.in2_ld = (int)x2->size(3); d.in2_h = (int)x2->size(1); d.in2_w = (int)x2->size(2);
    d.in2_stride = (int)x2_stride;
  }
  d.flags = (int)flags;
  d.cu_budget = (int)cu_budget;
  d.band_top = (int)band_top; d.band_bottom = (int)band_bottom;
  ok(fcp_conv2d_nhwc_f32(&d, cur_stream()), "fcp::conv2d");
  return out;
}

// Two schemas, so that the aliasing is declared truthfully: `conv2d` allocates and returns a fresh dense tensor,
// `conv2d_out` writes channels [out_c0, out_c0 + cout) of the caller's buffer and returns nothing.
Tensor conv2d(const Tensor& x, int64_t x_c0, int64_t cin, const Tensor& w, const c10::optional<Tensor>& bias,
              const c10::optional<Tensor>& wscale, const c10::optional<Tensor>& res1, int64_t res1_c0,
              const c10::optional<Tensor>& res2, int64_t res2_c0, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
              int64_t pad, double act_slope, double alpha, double alpha2, bool res1_pre, int64_t precision, int64_t in_fmt,
              int64_t out_fmt, int64_t res1_fmt, int64_t res2_fmt, bool in_up2, bool cin4, int64_t tile_m, int64_t tile_n,
              const c10::optional<Tensor>& x2, int64_t x2_c0, int64_t cin2, int64_t x2_stride, int64_t flags, int64_t cu_budget,
              int64_t band_top, int64_t band_bottom) {
  return conv2d_impl(x, x_c0, cin, w, bias, wscale, res1, res1_c0, res2, res2_c0, c10::nullopt, 0, cout, kh, kw, stride, pad,
                     act_slope, alpha, alpha2, res1_pre, precision, in_fmt, out_fmt, res1_fmt, res2_fmt, in_up2, cin4, tile_m,
                     tile_n, x2, x2_c0, cin2, x2_stride, flags, cu_budget, band_top, band_bottom);
}

void conv2d_out(const Tensor& x, int64_t x_c0, int64_t cin, const Tensor& w, const c10::optional<Tensor>& bias,
                const c10::optional<Tensor>& wscale, const c10::optional<Tensor>& res1, int64_t res1_c0,
                const c10::optional<Tensor>& res2, int64_t res2_c0, Tensor& out, int64_t out_c0, int64_t cout, int64_t kh,
                int64_t kw, int64_t stride, int64_t pad, double act_slope, double alpha, double alpha2, bool res1_pre,
                int64_t precision, int64_t in_fmt, int64_t out_fmt, int64_t res1_fmt, int64_t res2_fmt, bool in_up2, bool cin4,
                int64_t tile_m, int64_t tile_n, const c10::optional<Tensor>& x2, int64_t x2_c0, int64_t cin2,
                int64_t x2_stride, int64_t flags, int64_t cu_budget, int64_t band_top, int64_t band_bottom) {
  conv2d_impl(x, x_c0, cin, w, bias, wscale, res1, res1_c0, res2, res2_c0, out, out_c0, cout, kh, kw, stride, pad, act_slope,
              alpha, alpha2, res1_pre, precision, in_fmt, out_fmt, res1_fmt, res2_fmt, in_up2, cin4, tile_m, tile_n, x2, x2_c0,
              cin2, x2_stride, flags, cu_budget, band_top, band_bottom);
}

std::tuple<Tensor, Tensor> bottleneck_chain(const Tensor& t1, int64_t t1_c0, const c10::optional<Tensor>& res, int64_t res_c0,
                                            const c10::optional<Tensor>& w2, const c10::optional<Tensor>& ws2,
                                            const c10::optional<Tensor>& b2, const Tensor& w3, const Tensor& ws3,
                                            const Tensor& b3, const c10::optional<Tensor>& w1n,
                                            const c10::optional<Tensor>& ws1n, const c10::optional<Tensor>& b1n, int64_t c, int64_t nout, int64_t cn, int64_t tile_m, int64_t flags,
                                            const c10::optional<Tensor>& t1b, int64_t t1b_c0, int64_t cb, int64_t t1b_stride) {
  dev(t1, "t1", at::kFloat);
  FCP_DEVICE_GUARD(t1);
  const bool two = t1b.has_value() && t1b->defined();              // two-source pair: the trailing cb input channels come from t1b
  TORCH_CHECK(t1.dim() == 4 && t1_c0 + c - (two ? cb : 0) <= t1.size(3), "t1 (n,h,w,>=c) split32 buffer");
  const bool has_res = res.has_value() && res->defined();
  if (has_res) {
    dev(*res, "res", at::kFloat);
    TORCH_CHECK(res->dim() == 4 && res_c0 + nout <= res->size(3), "res (n,h,w,>=nout) split32 buffer");
  }
  Tensor out = at::empty({t1.size(0), t1.size(1), t1.size(2), nout}, t1.options());
  Tensor t1n = at::empty({t1.size(0), t1.size(1), t1.size(2), cn}, t1.options());
  fcp_chain_desc d = {};
  d.t1 = t1.data_ptr<float>() + t1_c0; d.out = out.data_ptr<float>(); d.t1n = t1n.data_ptr<float>();
  if (has_res) { d.res = res->data_ptr<float>() + res_c0; d.res_ld = (int)res->size(3); }
  if (w2.has_value() && w2->defined()) {
    d.w2 = w2->data_ptr(); d.ws2 = optp<float>(ws2, "ws2", at::kFloat); d.b2 = optp<float>(b2, "b2", at::kFloat);
  }
  d.w3 = w3.data_ptr(); d.ws3 = dev(ws3, "ws3", at::kFloat).data_ptr<float>(); d.b3 = dev(b3, "b3", at::kFloat).data_ptr<float>();
  // expand form (cn == 0): conv3 + residual alone, no next conv1 — its filter is absent and t1n comes back with 0 channels
  const bool has_next = w1n.has_value() && w1n->defined();
  TORCH_CHECK(has_next == (cn > 0), "w1n / ws1n / b1n are given exactly when cn > 0 (cn == 0: the expand form)");
  if (has_next) {
    d.w1n = w1n->data_ptr(); d.ws1n = optp<float>(ws1n, "ws1n", at::kFloat); d.b1n = optp<float>(b1n, "b1n", at::kFloat);
    TORCH_CHECK(d.ws1n && d.b1n, "ws1n and b1n accompany w1n");
  } else {
    d.t1n = nullptr;
  }
  d.n = (int)t1.size(0); d.h = (int)t1.size(1); d.w = (int)t1.size(2); d.c = (int)c; d.cn = (int)cn; d.nout = (int)nout;
  d.t1_ld = (int)t1.size(3); d.out_ld = (int)nout; d.t1n_ld = (int)cn; d.tile_m = (int)tile_m; d.flags = (int)flags;
  if (two) {
    dev(*t1b, "t1b", at::kFloat);
    TORCH_CHECK(t1b->dim() == 4 && t1b->size(0) == t1.size(0) && t1b_c0 + cb <= t1b->size(3), "t1b (n,hb,wb,>=cb) split32 buffer");
    d.t1b = t1b->data_ptr<float>() + t1b_c0; d.cb = (int)cb; d.t1b_ld = (int)t1b->size(3);
    d.t1b_h = (int)t1b->size(1); d.t1b_w = (int)t1b->size(2); d.t1b_stride = (int)t1b_stride;
  }
  ok(fcp_bottleneck_chain_f16x3(&d, cur_stream()), "fcp::bottleneck_chain");
  return {out, t1n};
}

// ---- detect_postprocess: three fused head maps -> compacted candidates
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> retina_decode(const Tensor& h0, const Tensor& h1, const Tensor& h2,
                                                                 int64_t img_h, int64_t img_w, double vis, double var0,
                                                                 double var1) {
  dev(h0, "head0", at::kFloat); dev(h1, "head1", at::kFloat); dev(h2, "head2", at::kFloat);
  FCP_DEVICE_GUARD(h0);
  TORCH_CHECK(img_h > 0 && img_w > 0, "image size must be positive");
  TORCH_CHECK(h0.dim() == 4, "head maps are (n, ceil(h/s), ceil(w/s), 32) fp32 NHWC");
  const int64_t n = h0.size(0);
  int64_t P = 0;
  const Tensor* heads[3] = {&h0, &h1, &h2};
  int li = 0;
  for (int64_t s : {8, 16, 32}) {
    const Tensor& h = *heads[li++];
    const int64_t fh = (img_h + s - 1) / s, fw = (img_w + s - 1) / s;
    TORCH_CHECK(h.dim() == 4 && h.size(0) == n && h.size(1) == fh && h.size(2) == fw && h.size(3) == 32,
                "head map of stride ", s, " must be (", n, ", ", fh, ", ", fw, ", 32) for a ", img_h, " x ", img_w, " image, got ",
                h.sizes());
    P += 2 * fh * fw;
  }
  auto f = h0.options();
  auto i = h0.options().dtype(at::kInt);
  Tensor score = at::empty({n, P}, f), box = at::empty({n, P, 4}, f), ldm = at::empty({n, P, 10}, f);
  Tensor prior = at::empty({n, P}, i), count = at::empty({n}, i);
  ok(fcp_retina_decode(h0.data_ptr<float>(), h1.data_ptr<float>(), h2.data_ptr<float>(), (int)n, (int)img_h, (int)img_w,
                       (float)vis, (float)var0, (float)var1, score.data_ptr<float>(), box.data_ptr<float>(),
                       ldm.data_ptr<float>(), prior.data_ptr<int>(), count.data_ptr<int>(), nullptr, nullptr, nullptr,
                       cur_stream()), "fcp::retina_decode");
  return {score, box, ldm, prior, count};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> nms_select(const Tensor& score, const Tensor& box, const Tensor& count,
                                                      double thr, int64_t strategy) {
  dev(score, "cand_score", at::kFloat); dev(box, "cand_box", at::kFloat); dev(count, "cand_count", at::kInt);
  FCP_DEVICE_GUARD(score);
  TORCH_CHECK(score.dim() == 2, "cand_score is (n, cap)");
  const int64_t n = score.size(0), cap = score.size(1);
  TORCH_CHECK(box.dim() == 3 && box.size(0) == n && box.size(1) == cap && box.size(2) == 4, "cand_box must be (", n, ", ", cap,
              ", 4), got ", box.sizes());
  TORCH_CHECK(count.dim() == 1 && count.size(0) == n, "cand_count must be (", n, ",), got ", count.sizes());
  TORCH_CHECK(strategy >= 0 && strategy <= 2, "strategy: 0 all, 1 best, 2 largest");
  auto i = score.options().dtype(at::kInt);
  Tensor ws = at::empty({fcp_retina_nms_workspace_bytes((int)n, (int)cap)}, score.options().dtype(at::kByte));
  Tensor keep_pos = at::empty({n, cap}, i), keep_count = at::empty({n}, i), sel_pos = at::empty({n, cap}, i), sel_count = at::empty({n}, i);
  ok(fcp_retina_nms_select(score.data_ptr<float>(), box.data_ptr<float>(), count.data_ptr<int>(), (int)n, (int)cap, (float)thr,
                           (int)strategy, ws.data_ptr(), keep_pos.data_ptr<int>(), keep_count.data_ptr<int>(),
                           sel_pos.data_ptr<int>(), sel_count.data_ptr<int>(), cur_stream()), "fcp::nms_select");
  return {keep_pos, keep_count, sel_pos, sel_count};
}

std::tuple<Tensor, Tensor, Tensor> gather_faces(const Tensor& ldm, const Tensor& sel_pos, const Tensor& sel_count,
                                                const c10::optional<Tensor>& paddings, int64_t max_faces) {
  dev(ldm, "cand_ldm", at::kFloat); dev(sel_pos, "sel_pos", at::kInt); dev(sel_count, "sel_count", at::kInt);
  FCP_DEVICE_GUARD(ldm);
  TORCH_CHECK(sel_pos.dim() == 2, "sel_pos is (n, cap)");
  const int64_t n = sel_pos.size(0), cap = sel_pos.size(1);
  TORCH_CHECK(ldm.dim() == 3 && ldm.size(0) == n && ldm.size(1) == cap && ldm.size(2) == 10, "cand_ldm must be (", n, ", ", cap,
              ", 10), got ", ldm.sizes());
  TORCH_CHECK(sel_count.dim() == 1 && sel_count.size(0) == n, "sel_count must be (", n, ",), got ", sel_count.sizes());
  TORCH_CHECK(max_faces >= 1, "max_faces must be >= 1");
  if (paddings.has_value() && paddings->defined())
    TORCH_CHECK(paddings->dim() == 2 && paddings->size(0) == n && paddings->size(1) == 4, "paddings must be (", n, ", 4)");
  auto i = ldm.options().dtype(at::kInt);
  Tensor off = at::empty({n + 1}, i), out_ldm = at::empty({max_faces, 5, 2}, ldm.options()), out_img = at::empty({max_faces}, i);
  ok(fcp_retina_gather_faces(ldm.data_ptr<float>(), sel_pos.data_ptr<int>(), sel_count.data_ptr<int>(), (int)n, (int)cap,
                             optp<int>(paddings, "paddings", at::kInt), (int)max_faces, off.data_ptr<int>(),
                             out_ldm.data_ptr<float>(), out_img.data_ptr<int>(), cur_stream()), "fcp::gather_faces");
  return {out_ldm, out_img, off};
}

// face_count: optional device int32 scalar (live rows of a fixed-capacity face array); valid_total: optional device
// int64 scalar the number of ok faces is ADDED to (declared mutable in the schema).
std::tuple<Tensor, Tensor> similarity_from_5pt(const Tensor& src, const Tensor& dst, bool allow_skew,
                                               const c10::optional<Tensor>& face_count,
                                               const c10::optional<Tensor>& valid_total) {
  dev(src, "src", at::kFloat); dev(dst, "dst", at::kFloat);
  FCP_DEVICE_GUARD(src);
  TORCH_CHECK(src.dim() == 3 && src.size(2) == 2 && dst.dim() == 2 && dst.size(0) == src.size(1), "src (f,k,2), dst (k,2)");
  const int64_t f = src.size(0);
  Tensor mat = at::empty({f, 2, 3}, src.options().dtype(at::kDouble)), okf = at::empty({f}, src.options().dtype(at::kInt));
  const int* fc = optp<int>(face_count, "face_count", at::kInt);
  int64_t* vt = const_cast<int64_t*>(optp<int64_t>(valid_total, "valid_total", at::kLong));
  TORCH_CHECK(fc == nullptr || face_count->numel() == 1, "face_count is a scalar");
  TORCH_CHECK(vt == nullptr || valid_total->numel() == 1, "valid_total is a scalar");
  ok(fcp_estimate_transform_counted(src.data_ptr<float>(), dst.data_ptr<float>(), (int)f, (int)src.size(1), allow_skew, fc,
                                    mat.data_ptr<double>(), okf.data_ptr<int>(), vt, cur_stream()),
     "fcp::similarity_from_5pt");
  return {mat, okf};
}

Tensor warp_affine_u8(const Tensor& images, const Tensor& img_idx, const Tensor& mat, const c10::optional<Tensor>& okf,
                      const c10::optional<Tensor>& paddings, int64_t out_w, int64_t out_h, int64_t border) {
  dev(images, "images", at::kByte); dev(img_idx, "img_idx", at::kInt); dev(mat, "mat", at::kDouble);
  FCP_DEVICE_GUARD(images);
  TORCH_CHECK(images.dim() == 4 && images.size(3) == 3, "images (n,h,w,3) uint8");
  const int64_t f = img_idx.size(0);
  TORCH_CHECK(mat.dim() == 3 && mat.size(0) == f && mat.size(1) == 2 && mat.size(2) == 3, "mat must be (", f, ", 2, 3) float64");
  Tensor out = at::empty({f, out_h, out_w, 3}, images.options());
  ok(fcp_warp_affine_u8(images.data_ptr<uint8_t>(), (int)images.size(0), (int)images.size(1), (int)images.size(2),
                        img_idx.data_ptr<int>(), mat.data_ptr<double>(), optp<int>(okf, "ok", at::kInt),
                        optp<int>(paddings, "paddings", at::kInt), (int)f, (int)out_h, (int)out_w, (int)border,
                        out.data_ptr<uint8_t>(), cur_stream()), "fcp::warp_affine_u8");
  return out;
}

Tensor bicubic_down4_round(const Tensor& x4) {
  dev(x4, "x4", at::kFloat);
  FCP_DEVICE_GUARD(x4);
  TORCH_CHECK(x4.dim() == 3 && x4.size(0) % 4 == 0 && x4.size(1) % 4 == 0 && x4.size(2) >= 3, "x4 (4h,4w,ld>=3) fp32");
  const int64_t h = x4.size(0) / 4, w = x4.size(1) / 4;
  Tensor out = at::empty({h, w, 3}, x4.options().dtype(at::kByte));
  ok(fcp_bicubic_down4_u8(x4.data_ptr<float>(), (int)h, (int)w, (int)x4.size(2), out.data_ptr<uint8_t>(), cur_stream()),
     "fcp::bicubic_down4_round");
  return out;
}

std::tuple<Tensor, Tensor> parse_argmax_hist(const Tensor& logits, int64_t ncls, int64_t mid_h, int64_t mid_w, int64_t out_h,
                                             int64_t out_w) {
  dev(logits, "logits", at::kFloat);
  FCP_DEVICE_GUARD(logits);
  TORCH_CHECK(logits.dim() == 4 && logits.size(3) >= ncls, "logits (f,lh,lw,ld>=ncls)");
  const int64_t f = logits.size(0);
  Tensor labels = at::empty({f, out_h, out_w}, logits.options().dtype(at::kByte));
  Tensor counts = at::empty({f, ncls}, logits.options().dtype(at::kInt));
  ok(fcp_parse_tail(logits.data_ptr<float>(), (int)f, (int)logits.size(1), (int)logits.size(2), (int)logits.size(3), (int)ncls,
                    (int)mid_h, (int)mid_w, (int)out_h, (int)out_w, labels.data_ptr<uint8_t>(), counts.data_ptr<int>(), cur_stream()),
     "fcp::parse_argmax_hist");
  return {labels, counts};
}

}  // namespace

TORCH_LIBRARY(fcp, m) {
  m.def("conv2d(Tensor x, int x_c0, int cin, Tensor w, Tensor? bias, Tensor? wscale, Tensor? res1, int res1_c0, Tensor? res2, "
        "int res2_c0, int cout, int kh, int kw, int stride, int pad, float act_slope, float alpha, "
        "float alpha2, bool res1_pre, int precision, int in_fmt, int out_fmt, int res1_fmt, int res2_fmt, bool in_up2, "
        "bool cin4, int tile_m, int tile_n, Tensor? x2, int x2_c0, int cin2, int x2_stride, int flags, int cu_budget=0, int band_top=0, "
        "int band_bottom=0) -> Tensor");
  m.def("conv2d_out(Tensor x, int x_c0, int cin, Tensor w, Tensor? bias, Tensor? wscale, Tensor? res1, int res1_c0, Tensor? res2, "
        "int res2_c0, Tensor(a!) out, int out_c0, int cout, int kh, int kw, int stride, int pad, float act_slope, float alpha, "
        "float alpha2, bool res1_pre, int precision, int in_fmt, int out_fmt, int res1_fmt, int res2_fmt, bool in_up2, "
        "bool cin4, int tile_m, int tile_n, Tensor? x2, int x2_c0, int cin2, int x2_stride, int flags, int cu_budget=0, int band_top=0, "
        "int band_bottom=0) -> ()");
  m.def("bottleneck_chain(Tensor t1, int t1_c0, Tensor? res, int res_c0, Tensor? w2, Tensor? ws2, Tensor? b2, Tensor w3, Tensor ws3, "
        "Tensor b3, Tensor? w1n, Tensor? ws1n, Tensor? b1n, int c, int nout, int cn, int tile_m=0, int flags=0, Tensor? t1b=None, "
        "int t1b_c0=0, int cb=0, int t1b_stride=1) -> (Tensor, Tensor)");
  m.def("retina_decode(Tensor head0, Tensor head1, Tensor head2, int img_h, int img_w, float vis, float var0, float var1) "
        "-> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("nms_select(Tensor cand_score, Tensor cand_box, Tensor cand_count, float nms_threshold, int strategy) "
        "-> (Tensor, Tensor, Tensor, Tensor)");
  m.def("gather_faces(Tensor cand_ldm, Tensor sel_pos, Tensor sel_count, Tensor? paddings, int max_faces) -> (Tensor, Tensor, Tensor)");
  m.def("similarity_from_5pt(Tensor src, Tensor dst, bool allow_skew, Tensor? face_count=None, Tensor(a!)? valid_total=None) "
        "-> (Tensor, Tensor)");
  m.def("warp_affine_u8(Tensor images, Tensor img_idx, Tensor mat, Tensor? ok, Tensor? paddings, int out_w, int out_h, int border) -> Tensor");
  m.def("bicubic_down4_round(Tensor x4) -> Tensor");
  m.def("parse_argmax_hist(Tensor logits, int ncls, int mid_h, int mid_w, int out_h, int out_w) -> (Tensor, Tensor)");
  // ABI the veneer was COMPILED against (struct layouts of include/fcp_hip.h) and the ABI of the libfcp_hip.so it is
  // running on: torch_ops.load() refuses a stale veneer (its shorter structs would be read past their end)
  m.def("abi_version() -> (int, int)", []() -> std::tuple<int64_t, int64_t> {
    return {(int64_t)FCP_ABI_VERSION, (int64_t)fcp_abi_version()};
  });
}

TORCH_LIBRARY_IMPL(fcp, CUDA, m) {
  m.impl("conv2d", &conv2d);
  m.impl("conv2d_out", &conv2d_out);
  m.impl("bottleneck_chain", &bottleneck_chain);
  m.impl("retina_decode", &retina_decode);
  m.impl("nms_select", &nms_select);
  m.impl("gather_faces", &gather_faces);
  m.impl("similarity_from_5pt", &similarity_from_5pt);
  m.impl("warp_affine_u8", &warp_affine_u8);
  m.impl("bicubic_down4_round", &bicubic_down4_round);
  m.impl("parse_argmax_hist", &parse_argmax_hist);
}
