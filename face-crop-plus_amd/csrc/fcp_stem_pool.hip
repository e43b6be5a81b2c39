// RetinaFace stem in one pass: uint8 image -> (x - mean) -> 7x7 / stride 2 / pad 3 conv (3 -> 64, BatchNorm
// folded) -> ReLU -> 3x3 / stride 2 / pad 1 max-pool, written as a (split32 or fp32) NHWC channel slice.
// Replaces u8_to_nhwc4 + conv + max-pool (reference retinaface.py:450-451 and the torchvision ResNet stem,
// _layers.py / retinaface.py:93-99): the 320x320x64 stem map (1.7 GB for a batch of 64 at 640^2) never
// reaches HBM; only the uint8 image is read and the pooled map written.
//
// Arithmetic.  x - mean is an integer in [-123, 151], exact in binary16, so the activation operand has no
// lo part and the fp16x3 product needs two matrix instructions, a*wh + a*wl (weights pre-scaled per filter by
// a power of two and split hi + lo offline, like every other filter of the fp16x3 path), accumulated in
// fp32 — the same accuracy class as the generic kernels, with a different (kh, kw*3+c) summation order.
//
// Mapping.  A persistent workgroup of 8 waves (two per SIMD) walks patches of 5 x 16 pooled pixels.  A patch
// needs 11 x 33 stem pixels (12 MFMA row tiles of 32) from a 27 x 71 pixel image patch.  The patch is fetched as
// bytes (branch-free, clamped addresses, one patch ahead so the loads fly under the MFMAs; a thread owns 12 bytes of
// ONE patch row, 18 bytes = 6 pixels apart, so its row clamp, channel and mean are fixed and the LDS offsets are
// immediates), converted ONCE to binary16 (x - mean; 0 outside the image = the conv's zero padding) and staged in LDS.
// K = 7 filter rows x 24 (21 = 7 taps x 3 channels, padded): a lane's 8 consecutive K values are 8 consecutive
// binary16 of one staged row, i.e. the MFMA fragment is four aligned dword reads with no arithmetic, requested one
// k-step ahead of the MFMAs that use it.  Wave w owns column tile w & 1 (32 filters; its 22 fragments = 11 k-steps x
// hi / lo stay in registers for the whole kernel) and the row tiles (w >> 1) + 4k.  Raw accumulators go to LDS as fp32
// (-inf for stem pixels outside the stem map: border patches only); the 3 x 3 / 2 max is taken there separably — a
// thread owns two channels of one pooled column: 33 unconditional 8-byte reads, 11 horizontal and 5 vertical three-way
// maxima for 10 outputs, one round on all 512 threads — and scale, bias and ReLU — monotone per channel, so they commute
// with the max — are applied to the 80 pooled pixels only.  Barriers order LDS traffic only (s_waitcnt lgkmcnt(0) +
// s_barrier): neither the output stores nor the prefetch are drained at a barrier.
//
// Where the time goes (tools/stem_probe.sh / stem_ablate.sh, cycles per patch and wave at batch 64, 640^2): MFMAs +
// staging 5700-6400 (4224 is the matrix pipe's share; the 48 staging writes per wave cost ~1900 of it, the fragment
// reads ~1100), pooling + stores 2100-2600, byte fetch 1000-2000 (96 byte-load instructions per patch on the vector
// memory path), commit 1300.  Unaligned dword loads instead of bytes were slower (3 per thread cost more than 12 byte
// loads), recomputing the fetch bookkeeping per patch cost 3x its registers' worth in vector instructions.
#include "fcp_conv_common.h"

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int PH = 5;                          // pooled rows per patch
constexpr int SH = 2 * PH + 1;                 // stem rows per patch: 11
constexpr int IH = 2 * SH + 5;                 // 27 image rows (+1 spare row for the padded K chunk)
constexpr int SPITCH = 68;                     // floats per staged stem pixel (64 + 4: conflict-free pooling)
constexpr int KSTEPS = 11;                     // 22 chunks of 8 K values (7 rows x 3 chunks, +1 zero chunk)

// Patch geometry as a function of the patch width PW (pooled pixels): 16 = the product form, 512 threads, one workgroup per CU;
// 8 = 256 threads and 74 KiB of LDS, TWO workgroups per CU (round 5).  A patch runs as a sequence of phases separated by
// workgroup barriers (commit -> MFMAs -> pooling + stores -> conv1 -> its epilogue), the matrix pipe being busy in one of them
// only (23 % of the cycles in the round-3 probes): the idea was that two independent workgroups per CU put one's MFMA phase beside
// the other's LDS / store phases.  Same work per wave and patch (3 row tiles x 11 k-steps), same K order per stem pixel, same
// bits (tools/stem_hash.py: identical digest) — and 8.8 % SLOWER in an in-call A/B (689.7 vs 634.0 us: +10 % image bytes and +3 %
// stem pixels per pooled pixel for the narrower halo, and no overlap gained: profiles/r05_probes.md section 3).  Kept as a
// build-time geometry (-DFCP_STEM_PW=8) for that record.
template <int PW_>
struct StemGeo {
  static constexpr int PW = PW_;
  static constexpr int SW = 2 * PW + 1;                    // stem columns per patch: 33 | 17
  static constexpr int NSTEM = SH * SW;                    // 363 | 187
  static constexpr int NTILES = (NSTEM + 31) / 32;         // 12 | 6 row tiles of 32 stem pixels
  static constexpr int IWB = (2 * SW + 5) * 3;             // 213 | 117 bytes per image row of the patch
  static constexpr int IPITCH = 6 * SW + 18;               // 216 | 120 binary16 elements per staged image row (a lane's last K chunk ends at 6 SW + 17)
  static constexpr int IN_ELEMS = (IH + 1) * IPITCH;
  static constexpr int NT = 32 * PW;                       // threads: the pooling pass is (32 channel pairs) x (PW pooled columns)
  static constexpr int NGRP = NT / 128;                    // row-tile groups: waves = NGRP x 2 column tiles
  static constexpr int TPR = NT / 28;                      // threads per patch row: 18 | 9 (a whole number of pixels: 6 | 3)
  static constexpr int NLOAD = (IWB + TPR - 1) / TPR;      // 12 | 13 bytes per thread: columns c0 + TPR j of its row
  static constexpr int C1ROWS = (PH * PW + 31) / 32 * 32;  // the patch's pooled pixels padded to MFMA row tiles: 96 | 64
  static constexpr int C1_BYTES = C1ROWS * 2 * 128;        // conv1 operand image: [rows][2 channel slices][128 B]
  static constexpr int WGS = PW <= 8 ? 2 : 1;              // workgroups per CU
  static_assert(TPR % 3 == 0 && IH * TPR <= NT && IPITCH % 2 == 0 && NTILES % NGRP == 0, "stem patch geometry");
};

struct StemParams {
  const uint8_t* img;
  const float* imgf;       // F32IN: fp32 NHWC4 input (n, h, w, 4), already normalised (BiSeNet, bise.py:387-393); channel 3 is ignored
  const uint32_t* wfrag;   // [2 column tiles][11 k-steps][hi, lo][64 lanes][4 dwords]
  const float* bias;
  const float* wscale;
  float* out;
  int n, h, w, hs, ws, hp, wp, out_ld, out_fmt;
  int tiles_y, tiles_x, npatches;
  int mean[3];
  // optional 1x1 conv 64 -> 64 (+ ReLU) on the pooled map, i.e. conv1 of the first bottleneck (HAS_C1 instantiation)
  const char* w1;          // packed fp16x3 filter [>= 64 rows][2 slices][32 hi | 32 lo binary16] (engine.py::pack_conv)
  const float* ws1;
  const float* b1;
  float* t1;               // split32 output (n, hp, wp, t1_ld)
  int t1_ld;
};

constexpr int CPITCH = 68;                     // floats per pixel of conv1's fp32 staging tile (in the stem stage region)
__device__ __forceinline__ int swz1(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain vmcnt, i.e. park every wave
// until the pooled pixels of this patch have reached HBM and the next patch's bytes have arrived.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// F32IN (round 5; BiSeNet's ResNet-18 stem, _layers.py:241-247 of the reference): the input is an fp32 NHWC4 tensor instead of
// bytes minus an integer mean, so the activation has a lo part: the patch is staged as TWO binary16 planes (hi = x rounded toward
// zero, lo = x - hi, the split of every other fp16x3 kernel) and a k-step is three matrix instructions (al*wh + ah*wl + ah*wh)
// instead of two.  Everything else — patch walk, K order (kh, kw * 3 + c), staging, separable max-pool, scale / bias / ReLU on
// the pooled pixels — is the same code.
template <bool HAS_C1, int PW_, bool F32IN = false>
__global__ void __launch_bounds__(StemGeo<PW_>::NT, StemGeo<PW_>::WGS) stem_pool_kernel(const StemParams p) {
  static_assert(!(HAS_C1 && F32IN), "the fp32-input stem has no conv1 tail");
  using G = StemGeo<PW_>;
  constexpr int PW = G::PW, SW = G::SW, NSTEM = G::NSTEM, NTILES = G::NTILES, IWB = G::IWB, IPITCH = G::IPITCH;
  constexpr int IN_ELEMS = G::IN_ELEMS, NT = G::NT, NGRP = G::NGRP, TPR = G::TPR, NLOAD = G::NLOAD, C1ROWS = G::C1ROWS;
  constexpr int KPG = NTILES / NGRP;                                // row tiles per wave: 3
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ __attribute__((aligned(16))) _Float16 inh[IN_ELEMS];   // (x - mean) as binary16: exact integers (F32IN: the hi parts)
  __shared__ __attribute__((aligned(16))) _Float16 inl[F32IN ? IN_ELEMS : 8];   // F32IN: the lo parts
  float* stage = smem;
  char* c1in = reinterpret_cast<char*>(smem + NSTEM * SPITCH);      // HAS_C1: conv1's operand image (split32 rows, swizzled)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  // this wave's column tile (32 filters) and row-tile group; its 22 filter fragments (11 k-steps x hi / lo) stay in
  // registers for the whole kernel
  const int ct = wave & 1, grp = wave >> 1;
  f16x8 wh[KSTEPS], wl[KSTEPS];
#pragma unroll
  for (int q = 0; q < KSTEPS; ++q) {
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(p.wfrag) + ((ct * KSTEPS + q) * 2) * 64 + lane;
    wh[q] = __builtin_bit_cast(f16x8, src[0]);
    wl[q] = __builtin_bit_cast(f16x8, src[64]);
  }

  // pooling pass: thread = (channel pair tid & 31, pooled column tid >> 5): always the same two channels
  const int pc2 = 2 * (tid & 31), pcol = tid >> 5;
  const float ws2a = p.wscale[pc2], ws2b = p.wscale[pc2 + 1], bias2a = p.bias[pc2], bias2b = p.bias[pc2 + 1];
  float ws1a = 0.f, ws1b = 0.f, b1a = 0.f, b1b = 0.f;
  if constexpr (HAS_C1) { ws1a = p.ws1[pc2]; ws1b = p.ws1[pc2 + 1]; b1a = p.b1[pc2]; b1b = p.b1[pc2 + 1]; }
  // HAS_C1: waves 0 .. 2 C1ROWS / 32 - 1 own one 32 x 32 tile of the C1ROWS x 64 conv1 output each (row tile wave >> 1, filters (wave & 1) * 32 ..).
  // Its filter fragments — 2 channel slices x 2 k-halves x hi / lo, 8 KiB for the whole conv, L1-resident — are fetched per
  // patch (the stem's 22 fragments fill the register file).  K order and term order are those of conv_igemm_f16x3_dma
  // (slices ascending; per k-half al*bh, ah*bl, ah*bh): same bits.
  const int c1rt = wave >> 1, c1ct = wave & 1;

  // this thread's share of the image patch: bytes c0 + TPR * j (j = 0 .. NLOAD - 1) of patch row tid / TPR.  TPR is a whole number
  // of pixels, so the channel (and its mean), the row and its clamp are per thread, the pixel column advances by TPR / 3 per
  // byte and the LDS offsets are immediates: four registers of bookkeeping and ~5 vector instructions per byte each for
  // the fetch and the commit (a flat byte index took 48 registers and three times the arithmetic).
  const int frow = tid / TPR, fc0 = tid - frow * TPR;          // rows >= IH (threads 486 ..) fetch nothing
  const int fx0 = fc0 / 3, fch = fc0 - fx0 * 3;
  const int fmean = fch == 0 ? p.mean[0] : (fch == 1 ? p.mean[1] : p.mean[2]);
  _Float16* const fdst = inh + min(frow, IH - 1) * IPITCH + fc0;
  // fetch(): branch-free byte loads from clamped (always valid) addresses, nothing consumed before commit(), so
  // all of them are in flight under the MFMAs of the current patch
  uint8_t pre[F32IN ? 1 : NLOAD];
  float pref[F32IN ? NLOAD : 1];
  unsigned pre_ok = 0u;
  auto fetch = [&](int patch) {
    const int pxi = patch % p.tiles_x;
    const int pyi = (patch / p.tiles_x) % p.tiles_y;
    const int ni = patch / (p.tiles_x * p.tiles_y);
    const int iy0 = 4 * pyi * PH - 5, ix0 = 4 * pxi * PW - 5;
    const int y = iy0 + frow;
    const bool rowok = (unsigned)y < (unsigned)p.h && frow < IH;
    const int yc = min(max(y, 0), p.h - 1);
    const uint8_t* base = p.img + ((long)ni * p.h + yc) * p.w * 3 + fch;
    const float* basef = p.imgf + ((long)ni * p.h + yc) * p.w * 4 + fch;
    pre_ok = 0u;
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      const int x = ix0 + fx0 + (TPR / 3) * j;
      pre_ok |= (rowok && (unsigned)x < (unsigned)p.w) ? (1u << j) : 0u;
      if constexpr (F32IN) pref[j] = basef[(unsigned)(min(max(x, 0), p.w - 1) * 4)];
      else pre[j] = base[(unsigned)(min(max(x, 0), p.w - 1) * 3)];
    }
  };
  _Float16* const fdstl = inl + (F32IN ? min(frow, IH - 1) * IPITCH + fc0 : 0);
  auto commit = [&]() {
    if (frow < IH) {
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) {     // x - mean, or 0 outside the image (the conv's zero padding)
        if (fc0 + TPR * j < IWB) {
          if constexpr (F32IN) {            // hi = x toward zero, lo = x - hi (exact in fp32) toward zero: split8's arithmetic
            const float v = ((pre_ok >> j) & 1u) ? pref[j] : 0.f;
            const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v, 0.f));
            const unsigned lu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fcp_mix_diff<0>(v, hu), 0.f));
            fdst[TPR * j] = __builtin_bit_cast(_Float16, (unsigned short)(hu & 0xFFFFu));
            fdstl[TPR * j] = __builtin_bit_cast(_Float16, (unsigned short)(lu & 0xFFFFu));
          } else {
            fdst[TPR * j] = (_Float16)(float)(((pre_ok >> j) & 1u) ? (int)pre[j] - fmean : 0);
          }
        }
      }
    }
  };

  // zero the spare row / pitch padding once (read by the zero-weight K chunk: must be finite)
  for (int i = tid; i < IN_ELEMS / 2; i += NT) {
    reinterpret_cast<uint32_t*>(inh)[i] = 0u;
    if constexpr (F32IN) reinterpret_cast<uint32_t*>(inl)[i] = 0u;
  }
  __syncthreads();

  // image-patch byte offset of this lane's stem pixel in each of the wave's row tiles (patch independent)
  int abase_t[KPG];
#pragma unroll
  for (int k = 0; k < KPG; ++k) {
    int pix = (grp + NGRP * k) * 32 + (lane & 31);
    pix = pix < NSTEM ? pix : NSTEM - 1;
    const int si = pix / SW, sj = pix - si * SW;
    abase_t[k] = (2 * si) * IPITCH + 6 * sj;
  }

  int patch = blockIdx.x;
  if (patch < p.npatches) fetch(patch);
  for (; patch < p.npatches; patch += gridDim.x) {
    commit();
    lds_barrier();                                      // image patch visible
    const int next = patch + gridDim.x;
    if (next < p.npatches) fetch(next);                 // global loads of the next patch fly under the MFMAs

    const int pxi = patch % p.tiles_x;
    const int pyi = (patch / p.tiles_x) % p.tiles_y;
    const int ni = patch / (p.tiles_x * p.tiles_y);
    const int py0 = pyi * PH, px0 = pxi * PW;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const bool interior = sy0 >= 0 && sy0 + SH <= p.hs && sx0 >= 0 && sx0 + SW <= p.ws;   // workgroup-uniform

#pragma unroll
    for (int k = 0; k < KPG; ++k) {
      const int t = grp + NGRP * k;
      const int abase = abase_t[k];
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      // K chunk of this lane in step q: filter row ch / 3, part ch % 3 with ch = 2 * q + half; 8 consecutive K values = 8
      // consecutive binary16 of one staged image row (4-byte aligned).  The fragment of step q + 1 is requested before the
      // MFMAs of step q.
      auto afrag_of = [&](const _Float16* plane, int q) {
        const int ch = 2 * q + half;
        const int kh = ch / 3, part = ch - kh * 3;
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(plane + abase + kh * IPITCH + 8 * part);
        const u32x4_t raw = {wp[0], wp[1], wp[2], wp[3]};
        return __builtin_bit_cast(f16x8, raw);
      };
      auto afrag = [&](int q) { return afrag_of(inh, q); };
      f16x8 a = afrag(0), al = a;
      if constexpr (F32IN) al = afrag_of(inl, 0);
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) {
        f16x8 an = a, aln = al;
        if (q + 1 < KSTEPS) {
          an = afrag(q + 1);
          if constexpr (F32IN) aln = afrag_of(inl, q + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F32IN) {                                                    // al*wh + ah*wl + ah*wh: the generic kernels' term order
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[q], al, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[q], a, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[q], a, acc, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[q], a, acc, 0, 0, 0);   // filters as the row operand: see below
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[q], a, acc, 0, 0, 0);
        }
        a = an; al = aln;
      }
      // raw accumulators -> stage.  The tile was computed transposed (filters x pixels): a lane holds ONE stem pixel
      // (lane & 31) and filters 8 g + 4 half + 0..3 of its column tile, i.e. four 16-byte staging writes per tile instead
      // of sixteen 4-byte ones, one validity test per lane.  Scale (> 0), bias and ReLU are monotone per channel, so they
      // commute with the max and are applied to the pooled pixels instead of the stem pixels.  Stem pixels outside the
      // stem map are the pool's padding: they are staged as -inf, so the pooling pass reads unconditionally.
      {
        const int row = t * 32 + (lane & 31);
        bool in = true;
        if (!interior) {
          const int si = row / SW, sj = row - si * SW;
          in = (unsigned)(sy0 + si) < (unsigned)p.hs && (unsigned)(sx0 + sj) < (unsigned)p.ws;
        }
        if (row < NSTEM) {
          float* dst = stage + row * SPITCH + ct * 32 + 4 * half;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dst + 8 * g) = in ? f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]}
                                                        : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
    }
    lds_barrier();                                      // stem tile staged; image patch no longer read

    // HAS_C1: conv1's filter fragments (8 x 16 B per lane, L1-resident) are requested here, a pooling pass ahead of use
    f16x8 c1w[8];
    if constexpr (HAS_C1) {
      if (c1rt < C1ROWS / 32) {
        const char* wrow = p.w1 + (size_t)(c1ct * 32 + (lane & 31)) * 256;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            c1w[(sl * 2 + sh) * 2] = *reinterpret_cast<const f16x8*>(wrow + sl * 128 + (2 * sh + half) * 16);
            c1w[(sl * 2 + sh) * 2 + 1] = *reinterpret_cast<const f16x8*>(wrow + sl * 128 + (4 + 2 * sh + half) * 16);
          }
      }
    }
    // max-pool 3x3 / 2 from LDS, separable: a thread owns two channels of one pooled column (5 pooled pixels): the
    // horizontal maxima of the 11 staged rows (33 8-byte reads, unconditional), then the vertical ones — 32 three-input
    // maxima for 10 outputs, one round on all eight waves.
    {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float* sp = stage + (2 * pcol) * SPITCH + pc2;
      f32x2 rm[SH];
#pragma unroll
      for (int si = 0; si < SH; ++si) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(sp + (si * SW) * SPITCH);
        const f32x2 b = *reinterpret_cast<const f32x2*>(sp + (si * SW + 1) * SPITCH);
        const f32x2 c = *reinterpret_cast<const f32x2*>(sp + (si * SW + 2) * SPITCH);
        rm[si][0] = fmaxf(fmaxf(a[0], b[0]), c[0]);
        rm[si][1] = fmaxf(fmaxf(a[1], b[1]), c[1]);
      }
      const int ox = px0 + pcol;
#pragma unroll
      for (int py = 0; py < PH; ++py) {
        const int oy = py0 + py;
        if (oy >= p.hp || ox >= p.wp) continue;
        float v0 = fmaxf(fmaxf(rm[2 * py][0], rm[2 * py + 1][0]), rm[2 * py + 2][0]) * ws2a + bias2a;
        float v1 = fmaxf(fmaxf(rm[2 * py][1], rm[2 * py + 1][1]), rm[2 * py + 2][1]) * ws2b + bias2b;
        v0 = v0 > 0.f ? v0 : 0.f;
        v1 = v1 > 0.f ? v1 : 0.f;
        const long pixel = ((long)ni * p.hp + oy) * p.wp + ox;
        if (p.out_fmt == 1) {
          const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));   // split8's arithmetic
          const unsigned lu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fcp_mix_diff<0>(v0, hu), fcp_mix_diff<1>(v1, hu)));
          char* ob = reinterpret_cast<char*>(p.out) + pixel * p.out_ld * 4 + split_chan_off(pc2);
          *reinterpret_cast<unsigned*>(ob) = hu;
          *reinterpret_cast<unsigned*>(ob + 64) = lu;
          if constexpr (HAS_C1) {                         // the same hi / lo bytes are conv1's operand: row pp, slice pc2 / 32
            const int pp = py * PW + pcol;
            const int q = (pc2 & 31) >> 3, sw = swz1(pp);
            char* trow = c1in + pp * 256 + (pc2 >> 5) * 128 + (pc2 & 7) * 2;
            *reinterpret_cast<unsigned*>(trow + ((q ^ sw) << 4)) = hu;
            *reinterpret_cast<unsigned*>(trow + (((4 + q) ^ sw) << 4)) = lu;
          }
        } else {
          *reinterpret_cast<f32x2*>(p.out + pixel * p.out_ld + pc2) = f32x2{v0, v1};
        }
      }
    }
    if constexpr (HAS_C1) {
      lds_barrier();                                    // conv1 operand complete; the stem stage is no longer read
      // INVARIANT: c1in rows of pooled pixels outside the image (and rows 80..95) are never written and hold stale LDS —
      // possibly NaN / Inf bit patterns.  That is harmless only because the tile is accumulated TRANSPOSED (filters x
      // pixels): a pixel's operand row feeds exactly one accumulator column, and the columns of those rows are never
      // stored.  Any cross-pixel reduction or a change of the tile orientation must zero c1in first.
      if (c1rt < C1ROWS / 32) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int row = c1rt * 32 + (lane & 31), sw = swz1(row);
        const char* arow = c1in + row * 256;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            const f16x8 bh = c1w[(sl * 2 + sh) * 2], bl = c1w[(sl * 2 + sh) * 2 + 1];
            const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + sl * 128 + (((2 * sh + half) ^ sw) << 4));
            const f16x8 al = *reinterpret_cast<const f16x8*>(arow + sl * 128 + (((4 + 2 * sh + half) ^ sw) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc, 0, 0, 0);   // transposed tile (filters x pixels), same
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc, 0, 0, 0);   // products and K order: 16-byte staging writes
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc, 0, 0, 0);
          }
        float* dst = stage + row * CPITCH + c1ct * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(dst + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      }
      lds_barrier();                                    // conv1's fp32 tile staged
      {                                                 // thread = (channel pair, pooled column) like the pooling pass
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const int ox = px0 + pcol;
        f32x2 cv[PH];
#pragma unroll
        for (int py = 0; py < PH; ++py) cv[py] = *reinterpret_cast<const f32x2*>(stage + (py * PW + pcol) * CPITCH + pc2);
#pragma unroll
        for (int py = 0; py < PH; ++py) {
          const int oy = py0 + py;
          if (oy >= p.hp || ox >= p.wp) continue;
          float x0 = cv[py][0] * ws1a + b1a, x1 = cv[py][1] * ws1b + b1b;   // the generic epilogue's expressions (act_slope 0, alpha 1)
          x0 = x0 >= 0.f ? x0 : x0 * 0.f;
          x1 = x1 >= 0.f ? x1 : x1 * 0.f;
          x0 = x0 * 1.f; x1 = x1 * 1.f;
          const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
          const unsigned lu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fcp_mix_diff<0>(x0, hu), fcp_mix_diff<1>(x1, hu)));
          const long pixel = ((long)ni * p.hp + oy) * p.wp + ox;
          char* ob = reinterpret_cast<char*>(p.t1) + pixel * p.t1_ld * 4 + split_chan_off(pc2);
          *reinterpret_cast<unsigned*>(ob) = hu;
          *reinterpret_cast<unsigned*>(ob + 64) = lu;
        }
      }
    }
    lds_barrier();                                      // stage free for the next patch
  }
}

}  // namespace

#ifndef FCP_STEM_PW
#define FCP_STEM_PW 16     // pooled pixels per patch row: 16 = one 8-wave workgroup per CU; 8 = two 4-wave workgroups (round 5: same bits, 8.8 % slower)
#endif

extern "C" int fcp_stem7x7s2_relu_pool_conv1_u8(const uint8_t* images, int n, int h, int w, const int32_t* mean_rgb,
                                                const void* wfrag, const float* bias, const float* wscale, float* out,
                                                int out_ld, int out_fmt, const void* w1, const float* ws1, const float* b1,
                                                float* t1, int t1_ld, fcp_stream_t stream) {
  FCP_REQUIRE(images && wfrag && bias && wscale && out && mean_rgb, "stem: null pointer");
  const bool has_c1 = w1 != nullptr;
  if (has_c1) {
    FCP_REQUIRE(ws1 && b1 && t1, "stem + conv1: conv1 needs its scales, bias and output");
    FCP_REQUIRE(out_fmt == 1, "stem + conv1: the pooled map must be written in split32 (its bytes are conv1's operand)");
    FCP_REQUIRE(t1_ld >= 64 && t1_ld % 32 == 0 && ((uintptr_t)t1 & 127) == 0 && ((uintptr_t)w1 & 15) == 0,
                "stem + conv1: misaligned conv1 filter / output view");
  }
  FCP_REQUIRE(n > 0 && h >= 1 && w >= 1 && (long)h * w * 3 < (1L << 31), "stem: bad image size");
  FCP_REQUIRE((unsigned)out_fmt <= 1u, "stem: out_fmt must be 0 (fp32) or 1 (split32)");
  FCP_REQUIRE(out_ld >= 64 && out_ld % (out_fmt ? 32 : 4) == 0 && ((uintptr_t)out & (out_fmt ? 127 : 15)) == 0,
              "stem: misaligned output view");
  FCP_REQUIRE(((uintptr_t)wfrag & 15) == 0, "stem: filter fragments must be 16-byte aligned");
  for (int c = 0; c < 3; ++c) FCP_REQUIRE(mean_rgb[c] >= 0 && mean_rgb[c] <= 255, "stem: means must be integers in 0..255");
  StemParams p = {};
  p.img = images; p.wfrag = static_cast<const uint32_t*>(wfrag); p.bias = bias; p.wscale = wscale; p.out = out;
  p.n = n; p.h = h; p.w = w;
  p.hs = (h + 6 - 7) / 2 + 1; p.ws = (w + 6 - 7) / 2 + 1;
  p.hp = (p.hs + 2 - 3) / 2 + 1; p.wp = (p.ws + 2 - 3) / 2 + 1;
  p.out_ld = out_ld; p.out_fmt = out_fmt;
  p.tiles_y = fcp_cdiv(p.hp, PH); p.tiles_x = fcp_cdiv(p.wp, StemGeo<FCP_STEM_PW>::PW);
  const long np = (long)n * p.tiles_y * p.tiles_x;
  FCP_REQUIRE(np < (1L << 31) && (long)n * h * w * 3 < (1L << 40), "stem: batch too large");
  p.npatches = (int)np;
  for (int c = 0; c < 3; ++c) p.mean[c] = mean_rgb[c];
  p.w1 = static_cast<const char*>(w1); p.ws1 = ws1; p.b1 = b1; p.t1 = t1; p.t1_ld = t1_ld;
  using G = StemGeo<FCP_STEM_PW>;
  const size_t lds = (size_t)G::NSTEM * SPITCH * 4 + (has_c1 ? G::C1_BYTES : 0);   // stem staging (+ conv1's operand image); the binary16 image patch is a static array
  const int cus = fcp_cu_count() * G::WGS;
  const int grid = (int)(np < cus ? np : cus);
  if (has_c1) {
    FCP_LDS_OPT_IN((&stem_pool_kernel<true, FCP_STEM_PW>), lds);
    hipLaunchKernelGGL((stem_pool_kernel<true, FCP_STEM_PW>), dim3(grid), dim3(G::NT), lds, (hipStream_t)stream, p);
  } else {
    FCP_LDS_OPT_IN((&stem_pool_kernel<false, FCP_STEM_PW>), lds);
    hipLaunchKernelGGL((stem_pool_kernel<false, FCP_STEM_PW>), dim3(grid), dim3(G::NT), lds, (hipStream_t)stream, p);
  }
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_stem7x7s2_relu_pool_f32(const float* x4, int n, int h, int w, const void* wfrag, const float* bias,
                                           const float* wscale, float* out, int out_ld, int out_fmt, fcp_stream_t stream) {
  FCP_REQUIRE(x4 && wfrag && bias && wscale && out, "stem(f32): null pointer");
  FCP_REQUIRE(n > 0 && h >= 1 && w >= 1 && (long)h * w * 4 < (1L << 31), "stem(f32): bad image size");
  FCP_REQUIRE((unsigned)out_fmt <= 1u, "stem(f32): out_fmt must be 0 (fp32) or 1 (split32)");
  FCP_REQUIRE(out_ld >= 64 && out_ld % (out_fmt ? 32 : 4) == 0 && ((uintptr_t)out & (out_fmt ? 127 : 15)) == 0,
              "stem(f32): misaligned output view");
  FCP_REQUIRE(((uintptr_t)wfrag & 15) == 0 && ((uintptr_t)x4 & 15) == 0, "stem(f32): input / filter fragments must be 16-byte aligned");
  using G = StemGeo<FCP_STEM_PW>;
  StemParams p = {};
  p.imgf = x4; p.wfrag = static_cast<const uint32_t*>(wfrag); p.bias = bias; p.wscale = wscale; p.out = out;
  p.n = n; p.h = h; p.w = w;
  p.hs = (h + 6 - 7) / 2 + 1; p.ws = (w + 6 - 7) / 2 + 1;
  p.hp = (p.hs + 2 - 3) / 2 + 1; p.wp = (p.ws + 2 - 3) / 2 + 1;
  p.out_ld = out_ld; p.out_fmt = out_fmt;
  p.tiles_y = fcp_cdiv(p.hp, PH); p.tiles_x = fcp_cdiv(p.wp, G::PW);
  const long np = (long)n * p.tiles_y * p.tiles_x;
  FCP_REQUIRE(np < (1L << 31) && (long)n * h * w * 4 < (1L << 40), "stem(f32): batch too large");
  p.npatches = (int)np;
  const size_t lds = (size_t)G::NSTEM * SPITCH * 4;
  const int cus = fcp_cu_count() * G::WGS;
  const int grid = (int)(np < cus ? np : cus);
  FCP_LDS_OPT_IN((&stem_pool_kernel<false, FCP_STEM_PW, true>), lds);
  hipLaunchKernelGGL((stem_pool_kernel<false, FCP_STEM_PW, true>), dim3(grid), dim3(G::NT), lds, (hipStream_t)stream, p);
  FCP_LAUNCH_OK();
  return 0;
}

extern "C" int fcp_stem7x7s2_relu_pool_u8(const uint8_t* images, int n, int h, int w, const int32_t* mean_rgb,
                                          const void* wfrag, const float* bias, const float* wscale, float* out,
                                          int out_ld, int out_fmt, fcp_stream_t stream) {
  return fcp_stem7x7s2_relu_pool_conv1_u8(images, n, h, w, mean_rgb, wfrag, bias, wscale, out, out_ld, out_fmt, nullptr, nullptr,
                                          nullptr, nullptr, 0, stream);
}
