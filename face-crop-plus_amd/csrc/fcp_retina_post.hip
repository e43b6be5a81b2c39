// RetinaFace post-processing on gfx950: fused softmax + analytic priors +
// decode + strict threshold + order-preserving compaction (wave ballot + LDS
// scan), then per-image bitonic sort + tiled greedy NMS + strategy select.
//
// All float arithmetic is written op-by-op in the reference's order and the
// file is built with -ffp-contract=off, so given identical inputs the NMS /
// selection *indices* are bit-exact against the reference (and the oracle).
#include "fcp_common.h"
#include "fcp_hip.h"

namespace {

constexpr int DEC_THREADS = 1024;
constexpr int NMS_THREADS = 1024;
constexpr int SORT_LDS_KEYS = 8192;  // 64 KiB of 8-byte keys
constexpr int NMS_ALIVE_WORDS = SORT_LDS_KEYS - 256;   // alive bitmap of the greedy pass: 64 candidates per word
constexpr int NMS_MAX_CAP = NMS_ALIVE_WORDS * 64;      // 507 904 candidates per image (a 2800^2 frame has 322 k priors)

struct Levels {
  int h[3], w[3], start[4];
};

__device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// decode: one workgroup per image, two passes over its priors (prior p = round * 1024 + thread, so (round, wave, lane)
// order is ascending prior order).  Pass 1 reads only the two class logits of a prior, thresholds the score and leaves
// one ballot count per (round, wave) in LDS; after ONE barrier a wave turns the counts into exclusive offsets; pass 2
// decodes the survivors and writes them at offset + rank-in-wave.  (The first version walked the priors in 17 rounds of
// compute -> ballot -> barrier -> prefix -> write -> barrier -> total -> barrier: 51 barriers per image with every round's
// loads exposed, 45 us for a batch of 64 at 640^2 = 1.5 TB/s.)  The head maps of one image (1-2.75 MB) stay in L2 between
// the passes.
// ---------------------------------------------------------------------------
constexpr int DEC_MAX_ROUNDS = 64;      // rounds per span (one pass bit per round in a 64-bit register)

struct Decoded {
  float score, box[4], ldm[10];
};

__device__ __forceinline__ const float* head_cell(const float* head0, const float* head1, const float* head2, const Levels& lv,
                                                  int img, int p, int& l, int& cell, int& a) {
  l = p >= lv.start[2] ? 2 : (p >= lv.start[1] ? 1 : 0);
  const int q = p - lv.start[l];
  cell = q >> 1; a = q & 1;
  return (l == 0 ? head0 : (l == 1 ? head1 : head2)) + ((long)img * lv.h[l] * lv.w[l] + cell) * 32;
}

// softmax over (bg, face), ATen CPU order: exp(x - max) * (1 / sum)
__device__ __forceinline__ float face_score(const float* hp, int a) {
  const float l0 = hp[2 * a], l1 = hp[2 * a + 1];
  const float mx = fmaxf(l0, l1);
  const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
  const float inv = 1.0f / (e0 + e1);
  return e1 * inv;
}

__device__ __forceinline__ void decode_prior(const float* hp, const Levels& lv, int l, int cell, int a, int img_h, int img_w,
                                             float v0, float v1, Decoded& d) {
  const float fw = (float)img_w, fh = (float)img_h;
  const int wl = lv.w[l];
  const int i = cell / wl, j = cell - i * wl;
  const int step = 8 << l;
  const double ms = (double)((16 << (2 * l)) << a);  // 16,32 | 64,128 | 256,512
  // PriorBox: python-double arithmetic, rounded once to f32 (_layers.py:57-60)
  const float pcx = (float)(((double)j + 0.5) * (double)step / (double)img_w);
  const float pcy = (float)(((double)i + 0.5) * (double)step / (double)img_h);
  const float pw = (float)(ms / (double)img_w);
  const float ph = (float)(ms / (double)img_h);
  d.score = face_score(hp, a);
  const float* bp = hp + 4 + 4 * a;
  const float cx = pcx + (bp[0] * v0) * pw;
  const float cy = pcy + (bp[1] * v0) * ph;
  const float bw = pw * expf(bp[2] * v1);
  const float bh = ph * expf(bp[3] * v1);
  const float x1 = cx - bw / 2.0f, y1 = cy - bh / 2.0f;
  const float x2 = bw + x1, y2 = bh + y1;
  d.box[0] = x1 * fw; d.box[1] = y1 * fh; d.box[2] = x2 * fw; d.box[3] = y2 * fh;
  const float* lp = hp + 12 + 10 * a;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    d.ldm[2 * k] = (pcx + (lp[2 * k] * v0) * pw) * fw;
    d.ldm[2 * k + 1] = (pcy + (lp[2 * k + 1] * v0) * ph) * fh;
  }
}

__global__ void __launch_bounds__(DEC_THREADS) retina_decode_kernel(
    const float* __restrict__ head0, const float* __restrict__ head1, const float* __restrict__ head2,
    Levels lv, int img_h, int img_w, float thr, float v0, float v1, float* __restrict__ cand_score,
    float* __restrict__ cand_box, float* __restrict__ cand_ldm, int* __restrict__ cand_prior,
    int* __restrict__ cand_count, float* __restrict__ dense_score, float* __restrict__ dense_box,
    float* __restrict__ dense_ldm) {
  constexpr int NWAVE = DEC_THREADS / 64;
  __shared__ int cnt[DEC_MAX_ROUNDS * NWAVE];      // ballot count, then exclusive offset, of (round, wave)
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = lv.start[3];
  const int rounds_all = (P + DEC_THREADS - 1) / DEC_THREADS;
  __shared__ int s_total;
  int total = 0;                                     // candidates of the earlier spans (workgroup-uniform)

  // spans of up to 64 rounds (65536 priors): one span for every size up to ~1248^2, more for larger frames
  for (int r0 = 0; r0 < rounds_all; r0 += DEC_MAX_ROUNDS) {
    const int rounds = min(DEC_MAX_ROUNDS, rounds_all - r0);
    // ---- pass 1: scores only
    unsigned long long passbits = 0ull;               // bit r: this thread's prior of round r0 + r passes
    for (int r = 0; r < rounds; ++r) {
      const int p = (r0 + r) * DEC_THREADS + tid;
      bool pass = false;
      if (p < P) {
        int l, cell, a;
        const float* hp = head_cell(head0, head1, head2, lv, img, p, l, cell, a);
        pass = face_score(hp, a) > thr;
      }
      passbits |= pass ? (1ull << r) : 0ull;
      const unsigned long long bal = __ballot(pass);
      if (lane == 0) cnt[r * NWAVE + wave] = __popcll(bal);
    }
    __syncthreads();
    // ---- exclusive prefix over (round, wave): one wave, 64 entries at a time
    if (wave == 0) {
      int carry = total;
      const int n = rounds * NWAVE;
      for (int base = 0; base < n; base += 64) {
        const int idx = base + lane;
        const int v = idx < n ? cnt[idx] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o);
          if (lane >= o) incl += t;
        }
        if (idx < n) cnt[idx] = carry + incl - v;
        carry += __shfl(incl, 63);
      }
      if (lane == 0) s_total = carry;
    }
    __syncthreads();
    total = s_total;
    // ---- pass 2: decode and write the survivors (and everything, when the dense outputs are requested)
    for (int r = 0; r < rounds; ++r) {
      const int p = (r0 + r) * DEC_THREADS + tid;
      const bool pass = (passbits >> r) & 1ull;
      const unsigned long long bal = __ballot(pass);
      if (p >= P || (!pass && dense_score == nullptr)) continue;
      int l, cell, a;
      const float* hp = head_cell(head0, head1, head2, lv, img, p, l, cell, a);
      Decoded dc;
      decode_prior(hp, lv, l, cell, a, img_h, img_w, v0, v1, dc);
      if (dense_score != nullptr) {
        const long d = (long)img * P + p;
        dense_score[d] = dc.score;
#pragma unroll
        for (int k = 0; k < 4; ++k) dense_box[d * 4 + k] = dc.box[k];
#pragma unroll
        for (int k = 0; k < 10; ++k) dense_ldm[d * 10 + k] = dc.ldm[k];
      }
      if (pass) {
        const long d = (long)img * P + cnt[r * NWAVE + wave] + __popcll(bal & ((1ull << lane) - 1ull));
        cand_score[d] = dc.score;
        cand_prior[d] = p;
#pragma unroll
        for (int k = 0; k < 4; ++k) cand_box[d * 4 + k] = dc.box[k];
#pragma unroll
        for (int k = 0; k < 10; ++k) cand_ldm[d * 10 + k] = dc.ldm[k];
      }
    }
    __syncthreads();                                  // cnt and s_total are rewritten by the next span
  }
  if (tid == 0) cand_count[img] = total;
}

// ---------------------------------------------------------------------------
// sort + NMS + strategy: one workgroup per image.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_pairs(unsigned long long* a, int npairs, int j, int k, int tid,
                                              int nthreads, int index_base) {
  for (int idx = tid; idx < npairs; idx += nthreads) {
    const int i = 2 * j * (idx / j) + (idx % j);
    const int l = i + j;
    const bool asc = (((i + index_base) & k) == 0);
    const unsigned long long x = a[i], y = a[l];
    if ((x > y) == asc) { a[i] = y; a[l] = x; }
  }
}

// reference IoU test (retinaface.py:281-292): survivor iff ovr <= thr
__device__ __forceinline__ bool suppressed(float kx1, float ky1, float kx2, float ky2, float karea,
                                           float x1, float y1, float x2, float y2, float area, float thr) {
  const float xx1 = fmaxf(kx1, x1), yy1 = fmaxf(ky1, y1);
  const float xx2 = fminf(kx2, x2), yy2 = fminf(ky2, y2);
  const float w = fmaxf(0.0f, xx2 - xx1 + 1.0f);
  const float h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
  const float a = w * h;
  const float ovr = a / (karea + area - a);
  return !(ovr <= thr);
}

__global__ void __launch_bounds__(NMS_THREADS) retina_nms_kernel(
    const float* __restrict__ cand_score, const float* __restrict__ cand_box,
    const int* __restrict__ cand_count, int cap, int cap_p2, float thr, int strategy,
    unsigned long long* __restrict__ ws_keys, float* __restrict__ ws_box, int* __restrict__ keep_pos,
    int* __restrict__ keep_count, int* __restrict__ sel_pos, int* __restrict__ sel_count) {
  __shared__ unsigned long long lkeys[SORT_LDS_KEYS];  // sort scratch, later alive bitmap + kept boxes
  __shared__ int s_kc;
  __shared__ float s_best_area[NMS_THREADS / 64];
  __shared__ int s_best_rank[NMS_THREADS / 64];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = cand_count[img];
  const float* score = cand_score + (long)img * cap;
  const float4* box = reinterpret_cast<const float4*>(cand_box) + (long)img * cap;
  int* kpos = keep_pos + (long)img * cap;
  int* spos = sel_pos + (long)img * cap;
  if (K <= 0) {
    if (tid == 0) { keep_count[img] = 0; sel_count[img] = 0; }
    return;
  }
  int Kp = 64;
  while (Kp < K) Kp <<= 1;
  unsigned long long* gkeys = ws_keys + (long)img * cap_p2;
  float* sbox = ws_box + (long)img * cap_p2 * 5;  // x1,y1,x2,y2,area in sorted order

  // ---- 1. keys: (~score bits, position) ascending == score desc, position asc
  const bool in_lds = Kp <= SORT_LDS_KEYS;
  unsigned long long* keys = in_lds ? lkeys : gkeys;
  for (int i = tid; i < Kp; i += NMS_THREADS) {
    unsigned long long kv = ~0ull;
    if (i < K) kv = ((unsigned long long)(~__float_as_uint(score[i])) << 32) | (unsigned)i;
    keys[i] = kv;
  }
  __syncthreads();
  // ---- 2. bitonic sort
  if (in_lds) {
    for (int k = 2; k <= Kp; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        bitonic_pairs(keys, Kp >> 1, j, k, tid, NMS_THREADS, 0);
        __syncthreads();
      }
  } else {
    for (int k = 2; k <= Kp; k <<= 1) {
      int j = k >> 1;
      for (; j >= SORT_LDS_KEYS; j >>= 1) {  // strides that span LDS chunks: in global memory
        bitonic_pairs(gkeys, Kp >> 1, j, k, tid, NMS_THREADS, 0);
        __threadfence_block();
        __syncthreads();
      }
      for (int c0 = 0; c0 < Kp; c0 += SORT_LDS_KEYS) {  // remaining strides chunk by chunk in LDS
        for (int i = tid; i < SORT_LDS_KEYS; i += NMS_THREADS) lkeys[i] = gkeys[c0 + i];
        __syncthreads();
        for (int jj = j; jj > 0; jj >>= 1) {
          bitonic_pairs(lkeys, SORT_LDS_KEYS >> 1, jj, k, tid, NMS_THREADS, c0);
          __syncthreads();
        }
        for (int i = tid; i < SORT_LDS_KEYS; i += NMS_THREADS) gkeys[c0 + i] = lkeys[i];
        __syncthreads();
      }
    }
  }
  // ---- 3. sorted boxes + areas to workspace, sorted order to gkeys (lds is reused below)
  for (int i = tid; i < K; i += NMS_THREADS) {
    const unsigned long long kv = keys[i];
    const int pos = (int)(unsigned)(kv & 0xffffffffull);
    const float4 b = box[pos];
    const float area = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
    sbox[i * 5 + 0] = b.x; sbox[i * 5 + 1] = b.y; sbox[i * 5 + 2] = b.z; sbox[i * 5 + 3] = b.w;
    sbox[i * 5 + 4] = area;
    if (in_lds) gkeys[i] = kv;
  }
  __threadfence_block();
  __syncthreads();

  // ---- 4. tiled greedy NMS.  LDS reuse: alive bitmap (Kp/64 words) then kept-box tile.
  unsigned long long* alive = lkeys;                                   // up to NMS_ALIVE_WORDS words
  float* tile = reinterpret_cast<float*>(lkeys + NMS_ALIVE_WORDS);     // 64 * 5 floats
  int* tile_n = reinterpret_cast<int*>(tile + 64 * 5);
  const int nwords = (K + 63) >> 6;
  for (int wd = tid; wd < nwords; wd += NMS_THREADS) {
    const int lo = wd << 6;
    unsigned long long m = 0ull;
    if (lo + 64 <= K) m = ~0ull;
    else if (lo < K) m = (1ull << (K - lo)) - 1ull;
    alive[wd] = m;
  }
  if (tid == 0) s_kc = 0;
  __syncthreads();

  const int ntiles = (K + 63) >> 6;
  for (int t = 0; t < ntiles; ++t) {
    if (wave == 0) {
      const int c = (t << 6) + lane;
      float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f;
      if (c < K) { x1 = sbox[c * 5]; y1 = sbox[c * 5 + 1]; x2 = sbox[c * 5 + 2]; y2 = sbox[c * 5 + 3]; area = sbox[c * 5 + 4]; }
      bool me = (alive[t] >> lane) & 1ull;
      for (int b = 0; b < 64; ++b) {
        const unsigned long long m = __ballot(me);
        if (!((m >> b) & 1ull)) continue;  // wave-uniform
        const float kx1 = __shfl(x1, b), ky1 = __shfl(y1, b), kx2 = __shfl(x2, b), ky2 = __shfl(y2, b);
        const float ka = __shfl(area, b);
        if (lane > b && me && suppressed(kx1, ky1, kx2, ky2, ka, x1, y1, x2, y2, area, thr)) me = false;
      }
      const unsigned long long m = __ballot(me);
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      const int kc = s_kc;
      if (me) {
        kpos[kc + rank] = (int)(unsigned)(gkeys[c] & 0xffffffffull);
        tile[rank * 5 + 0] = x1; tile[rank * 5 + 1] = y1; tile[rank * 5 + 2] = x2; tile[rank * 5 + 3] = y2;
        tile[rank * 5 + 4] = area;
      }
      if (lane == 0) { *tile_n = __popcll(m); s_kc = kc + __popcll(m); }
    }
    __syncthreads();
    const int nk = *tile_n;
    if (nk > 0) {
      for (int wd = t + 1 + wave; wd < nwords && (wd << 6) < K; wd += NMS_THREADS / 64) {
        const unsigned long long m = alive[wd];
        if (m == 0ull) continue;  // wave-uniform
        bool me = (m >> lane) & 1ull;
        if (me) {
          const int c = (wd << 6) + lane;
          const float x1 = sbox[c * 5], y1 = sbox[c * 5 + 1], x2 = sbox[c * 5 + 2], y2 = sbox[c * 5 + 3];
          const float area = sbox[c * 5 + 4];
          for (int q = 0; q < nk; ++q) {
            if (suppressed(tile[q * 5], tile[q * 5 + 1], tile[q * 5 + 2], tile[q * 5 + 3], tile[q * 5 + 4], x1,
                           y1, x2, y2, area, thr)) { me = false; break; }
          }
        }
        const unsigned long long nm = __ballot(me);
        if (lane == 0) alive[wd] = nm;
      }
    }
    __syncthreads();
  }
  const int kc = s_kc;
  if (tid == 0) keep_count[img] = kc;

  // ---- 5. take_by_strategy (retinaface.py:381-400)
  if (strategy == 0) {
    for (int i = tid; i < kc; i += NMS_THREADS) spos[i] = kpos[i];
    if (tid == 0) sel_count[img] = kc;
  } else if (strategy == 1) {
    if (tid == 0) { spos[0] = kpos[0]; sel_count[img] = kc > 0 ? 1 : 0; }
  } else {
    // first maximum of the +1-convention area over the kept boxes (NaN counts as maximal, like torch.argmax)
    float best = -INFINITY; int brank = 0x7fffffff; bool bnan = false;
    __threadfence_block();
    for (int i = tid; i < kc; i += NMS_THREADS) {
      const float4 b = box[kpos[i]];
      const float area = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
      const bool isn = area != area;
      if (bnan) continue;
      if (isn) { bnan = true; best = area; brank = i; }
      else if (brank == 0x7fffffff || area > best) { best = area; brank = i; }
    }
    // wave reduce: order (is_nan desc, area desc, rank asc)
    for (int off = 32; off > 0; off >>= 1) {
      const float oa = __shfl_down(best, off); const int orank = __shfl_down(brank, off);
      const int on = __shfl_down((int)bnan, off);
      bool take;
      if (orank == 0x7fffffff) take = false;
      else if (brank == 0x7fffffff) take = true;
      else if (on != (int)bnan) take = on != 0;
      else if (bnan) take = orank < brank;
      else take = (oa > best) || (oa == best && orank < brank);
      if (take) { best = oa; brank = orank; bnan = on != 0; }
    }
    if (lane == 0) { s_best_area[wave] = best; s_best_rank[wave] = bnan ? -brank - 1 : brank; }
    __syncthreads();
    if (tid == 0) {
      float fb = 0.f; int fr = 0x7fffffff; bool fn = false;
      for (int wv = 0; wv < NMS_THREADS / 64; ++wv) {
        int rk = s_best_rank[wv]; const float ar = s_best_area[wv];
        if (rk == 0x7fffffff) continue;
        const bool isn = rk < 0;
        if (isn) rk = -rk - 1;
        bool take;
        if (fr == 0x7fffffff) take = true;
        else if (isn != fn) take = isn;
        else if (isn) take = rk < fr;
        else take = (ar > fb) || (ar == fb && rk < fr);
        if (take) { fb = ar; fr = rk; fn = isn; }
      }
      if (kc > 0) { spos[0] = kpos[fr]; sel_count[img] = 1; } else sel_count[img] = 0;
    }
  }
}

// One workgroup (a single wave) per image: its face offset is the sum of the selection counts of the images before it
// (n is a batch size: a wave-strided read + a 64-lane reduction), so no block waits for another.  The blocks also clear
// the unused tail [total, max_faces) of the outputs between them — callers need no memset — and block n-1 publishes the
// total in face_offset[n].
__global__ void __launch_bounds__(64) retina_gather_kernel(
    const float* __restrict__ cand_ldm, const int* __restrict__ sel_pos, const int* __restrict__ sel_count,
    int n, int cap, const int* __restrict__ paddings, int max_faces, int* __restrict__ face_offset,
    float* __restrict__ out_ldm, int* __restrict__ out_img) {
  const int img = blockIdx.x, lane = threadIdx.x;
  int before = 0, all = 0;
  for (int i = lane; i < n; i += 64) {
    const int c = sel_count[i];
    all += c;
    before += i < img ? c : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o);
    all += __shfl_xor(all, o);
  }
  const int off = before, cnt = sel_count[img];
  if (lane == 0) {
    face_offset[img] = off;
    if (img == n - 1) face_offset[n] = all;
  }
  const float px = paddings ? (float)paddings[img * 4 + 2] : 0.f;  // left
  const float py = paddings ? (float)paddings[img * 4 + 0] : 0.f;  // top
  for (int e = lane; e < cnt * 10; e += 64) {
    const int k = e / 10, c = e - k * 10;
    const int face = off + k;
    if (face >= max_faces) break;                                   // e is increasing per lane: nothing further fits
    const float v = cand_ldm[((long)img * cap + sel_pos[(long)img * cap + k]) * 10 + c];
    out_ldm[(long)face * 10 + c] = v - ((c & 1) ? py : px);
    if (c == 0) out_img[face] = img;
  }
  for (long face = (long)all + img; face < max_faces; face += n) {   // unused tail: zero, shared out over the blocks
    if (lane < 10) out_ldm[face * 10 + lane] = 0.f;
    if (lane == 10) out_img[face] = 0;
  }
}

inline int pow2_ceil(int v) { int p = 64; while (p < v) p <<= 1; return p; }

}  // namespace

extern "C" int64_t fcp_retina_nms_workspace_bytes(int n, int cap) {
  return (int64_t)n * pow2_ceil(cap) * (8 + 5 * 4);
}

extern "C" int fcp_retina_decode(const float* head0, const float* head1, const float* head2, int n,
                                 int img_h, int img_w, float vis_threshold, float var0, float var1,
                                 float* cand_score, float* cand_box, float* cand_ldm, int32_t* cand_prior,
                                 int32_t* cand_count, float* dense_score, float* dense_box,
                                 float* dense_ldm, fcp_stream_t stream) {
  FCP_REQUIRE(head0 && head1 && head2, "retina_decode: null head pointer");
  FCP_REQUIRE(cand_score && cand_box && cand_ldm && cand_prior && cand_count, "retina_decode: null output");
  FCP_REQUIRE(n > 0 && img_h > 0 && img_w > 0, "retina_decode: bad sizes");
  FCP_REQUIRE((dense_score == nullptr) == (dense_box == nullptr) && (dense_box == nullptr) == (dense_ldm == nullptr),
              "retina_decode: dense outputs must be all set or all NULL");
  Levels lv;
  lv.start[0] = 0;
  for (int l = 0; l < 3; ++l) {
    const int s = 8 << l;
    lv.h[l] = (img_h + s - 1) / s;
    lv.w[l] = (img_w + s - 1) / s;
    lv.start[l + 1] = lv.start[l] + 2 * lv.h[l] * lv.w[l];
  }
  hipLaunchKernelGGL(retina_decode_kernel, dim3(n), dim3(DEC_THREADS), 0, (hipStream_t)stream, head0, head1,
                     head2, lv, img_h, img_w, vis_threshold, var0, var1, cand_score, cand_box, cand_ldm,
                     cand_prior, cand_count, dense_score, dense_box, dense_ldm);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_retina_nms_select(const float* cand_score, const float* cand_box,
                                     const int32_t* cand_count, int n, int cap, float nms_threshold,
                                     int strategy, void* workspace, int32_t* keep_pos, int32_t* keep_count,
                                     int32_t* sel_pos, int32_t* sel_count, fcp_stream_t stream) {
  FCP_REQUIRE(cand_score && cand_box && cand_count && workspace, "retina_nms: null input");
  FCP_REQUIRE(keep_pos && keep_count && sel_pos && sel_count, "retina_nms: null output");
  FCP_REQUIRE(n > 0 && cap > 0, "retina_nms: bad sizes");
  FCP_REQUIRE(cap <= NMS_MAX_CAP, "retina_nms: capacity above %d candidates per image is not supported", NMS_MAX_CAP);
  FCP_REQUIRE(strategy >= 0 && strategy <= 2, "Unsupported startegy: %d", strategy);
  FCP_REQUIRE(((uintptr_t)cand_box & 15) == 0 && ((uintptr_t)workspace & 7) == 0, "retina_nms: misaligned buffers");
  const int cap_p2 = pow2_ceil(cap);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
  float* sbox = reinterpret_cast<float*>(keys + (size_t)n * cap_p2);
  hipLaunchKernelGGL(retina_nms_kernel, dim3(n), dim3(NMS_THREADS), 0, (hipStream_t)stream, cand_score,
                     cand_box, cand_count, cap, cap_p2, nms_threshold, strategy, keys, sbox, keep_pos,
                     keep_count, sel_pos, sel_count);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_retina_gather_faces(const float* cand_ldm, const int32_t* sel_pos,
                                       const int32_t* sel_count, int n, int cap, const int32_t* paddings,
                                       int max_faces, int32_t* face_offset, float* out_ldm,
                                       int32_t* out_img, fcp_stream_t stream) {
  FCP_REQUIRE(cand_ldm && sel_pos && sel_count && face_offset && out_ldm && out_img, "retina_gather: null pointer");
  FCP_REQUIRE(n > 0 && cap > 0 && max_faces > 0, "retina_gather: bad sizes");
  hipLaunchKernelGGL(retina_gather_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, cand_ldm, sel_pos,
                     sel_count, n, cap, paddings, max_faces, face_offset, out_ldm, out_img);
  FCP_LAUNCH_OK();
  return 0;
}
