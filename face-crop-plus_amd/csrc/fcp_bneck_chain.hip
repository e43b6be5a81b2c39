// Fused ResNet bottleneck chain (fp16x3 path, split32 tensors), 64-channel bottlenecks (ResNet-50 layer 1):
//
//     t2  = relu(bn2(conv2_3x3(t1)))                     64 -> 64      (never leaves the workgroup)
//     out = relu(bn3(conv3_1x1(t2)) + x)                 64 -> 256     (written once)
//     t1' = relu(bn1'(conv1'_1x1(out)))                  256 -> CN     (the NEXT block's conv1: CN = 64, or 128 for
//                                                                       layer2.0.conv1)
//
// i.e. torchvision's Bottleneck.forward (retinaface.py:93-99 builds the body from it) from conv2 of block b to
// conv1 of block b+1.  Unfused, those three launches move  t1 + 2 t2 + 2 t2.. = 6.7 GB per block through HBM at
// the bench size (64 x 160 x 160 px: x and out are 1.68 GB each at 4 B per element) and run at the HBM rate;
// fused, a workgroup reads its t1 tile (+ halo, from L2) and its x tile, writes out and t1':  4.2 GB.
//
// One 256-thread workgroup owns 128 consecutive output pixels; two workgroups share a CU (80 KiB of LDS each), so
// one computes while the other waits on memory.
//
//   phase 1   implicit GEMM 128 x 64 x 576 exactly as conv_igemm_f16x3_dma<64, 2> (both operands by LDS-DMA, two
//             stages, taps fastest), epilogue -> T2 in LDS as the split32 operand image [2 slices][128 rows][128 B].
//   chunks    for each group j of 32 output channels of conv3 (8 groups):
//               phase 2  acc2[128 x 32]  = T2 . W3[j]^T                (K = 64; W3 group by LDS-DMA, double-buffered)
//               epilogue out[:, j] = relu(acc2 * ws3 + b3 + x[:, j]) -> HBM, and -> T3 in LDS (operand image)
//               phase 3  acc3[128 x CN] += T3 . W1'[:, j]^T            (K slice j of conv1'; by LDS-DMA)
//   epilogue  t1' = relu(acc3 * ws1 + b1) -> HBM.
//
// Arithmetic is that of the stand-alone kernels, instruction for instruction (al*bh + ah*bl + ah*bh per k-half,
// K slices in ascending order, the same fp32 epilogue expressions, the same hi/lo split of every stored tensor), so
// out and t1' are BIT-IDENTICAL to running the three convolutions separately (tests/test_chain_gpu.py).
//
// The same chunk loop also runs WITHOUT phase 1 ("pair": HAS_C2 = false) on 128-channel inputs, whose operand tile is
// then fetched by LDS-DMA instead of being computed:
//     out = relu(bn3(conv3_1x1(t)) [+ x])   128 -> NOUT;    t1' = relu(bn1'(conv1'_1x1(out)))   NOUT -> CN
//   * NOUT = 512, with residual, CN = 128:  conv3 of a layer-2 identity block + conv1 of the next block: the 512-channel
//     tensor (0.84 GB at the bench size) is not read back by a separate conv1 launch;
//   * NOUT = 256, no residual, CN = 64:  layer1.0's conv3 + downsample (one K-concatenated 1x1 conv over [conv2 out |
//     pooled stem]) + layer1.1's conv1.
// 128 KiB of LDS, one workgroup per CU.
#include "fcp_conv_common.h"

#include <type_traits>

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int FCP_CHAIN_STORE_AUX = 2;   // cache policy of the `out` stores of the chunk loop: nt (0 default / 16 sc1 measured equal)

struct ChainK {
  const float* t1;  unsigned t1_bytes;  int t1_ld;
  const float* w2;  unsigned w2_bytes;  const float* ws2;  const float* b2;
  const float* w3;  unsigned w3_bytes;  const float* ws3;  const float* b3;
  const float* res; int res_ld;
  float* out;       int out_ld;    unsigned out_bytes;
  const float* w1n; unsigned w1n_bytes; const float* ws1n; const float* b1n;
  float* t1n;       int t1n_ld;
  int n, h, w, M;
  int nt_store;
  int out_even;     // patch form: store `out` at even (y, x) only (FCP_CHAIN_OUT_EVEN_ONLY)
  // two-source pair (DIRECT forms): the trailing channels of conv3's input come from t1b, a split32 tensor (n, hb, wb, t1b_ld)
  // sampled at (y * sb, x * sb) — the K concatenation [conv2 out | x(::s, ::s)] of a stride-2 bottleneck's conv3 + downsample
  const float* t1b; unsigned t1b_bytes; int t1b_ld, hb, wb, sb;
};

// Pixels per workgroup tile: BMT = 128 (4 waves, up to two workgroups per CU; the default) or 256 (8 waves, one workgroup per
// CU: every filter byte the tile fetches — W2, the conv3 groups, the conv1' slices: 45 % of a 128-pixel tile's vector-memory
// traffic — serves twice the pixels; an option, not faster: profiles/r03_probes.md).  A wave owns 32 rows either way.
constexpr int C = 64;                 // bottleneck width of the variant with phase 1
constexpr int ROWB = 128;             // bytes per LDS operand row: 32 hi + 32 lo binary16
constexpr int stage_bytes(int bmt) { return (bmt + C) * ROWB; }       // one phase-1 stage (A rows then B rows): 24 | 40 KiB
constexpr int CT_OFF = 0;                       // chunk loop: BMT x 32 fp32 epilogue tile, rewritten in place as T3 (16 | 32 KiB)
constexpr int w1b_off(int bmt) { return bmt * 128; }                  // chunk loop: K slice j of conv1' (CN rows x 128 B)
// conv1' K slices are double-buffered wherever LDS allows (everything but the 128-pixel conv2 form with CN = 128, whose
// chunk buffers live in the 48 KiB of the dead phase-1 stages): the next slice's DMA can then be issued BEFORE the chunk's
// `out` stores, see the chunk loop
constexpr bool w1_double(int bmt, int cn, bool has_c2) { return !has_c2 || w1b_off(bmt) + 2 * cn * 128 + 2 * C * 128 <= 2 * stage_bytes(bmt); }
constexpr int w3b_off(int bmt, int cn, bool has_c2) { return w1b_off(bmt) + (w1_double(bmt, cn, has_c2) ? 2 : 1) * cn * 128; }   // conv3 filter groups, 2 x (CW * 128 B)
// region 0 = phase-1 stages | epilogue tiles | chunk buffers; the operand tile T2 (BMT x CW, 4 B per element) follows it
constexpr int r0_bytes(int bmt, int cw, int cn, bool has_c2) { return has_c2 ? 2 * stage_bytes(bmt) : w3b_off(bmt, cn, has_c2) + 2 * cw * 128; }
// In the pair forms T2 ALIASES the chunk buffers: a wave's T2 fragments are the same for every chunk and live in
// registers, so the tile is only needed until they have been read.  That is what lets the 256-wide pair fit at all
// (144 + 128 KiB otherwise) and brings the 128-wide pairs down to 80 KiB: TWO workgroups per CU, the second one's
// MFMAs under the first one's epilogue (FCP_CHAIN_NOALIAS: the 128-wide pairs as before, 128-144 KiB, one per CU).
constexpr bool alias_t2(int bmt, int cw, int cn, bool has_c2) { return !has_c2; }
// DIRECT (two-source pair forms, round 5): the operand tile never exists in LDS at all — a lane's fragments are 16-byte pieces of
// the split32 pixel rows, i.e. plain buffer loads straight into the registers they live in for the whole chunk loop (a
// 128 x 384-channel tile is 192 KiB: it fits neither LDS nor an alias of the chunk buffers); LDS holds the chunk buffers only.
constexpr int lds_bytes(int bmt, int cw, int cn, bool has_c2, bool direct = false) {   // 80 KiB (two per CU) | 112-160 KiB
  const int r0 = r0_bytes(bmt, cw, cn, has_c2), t2 = bmt * cw * 4;
  if (direct) return r0;
  return !alias_t2(bmt, cw, cn, has_c2) ? r0 + t2 : (r0 > w1b_off(bmt) + t2 ? r0 : w1b_off(bmt) + t2);
}
constexpr int wgs_per_cu(int bmt, int cw, int cn, bool has_c2, bool direct = false) {
  // the two-source DIRECT pair holds 48 fragments + conv1's accumulators: one wave per SIMD; the EXPAND form (cn = 0: conv3 only,
  // no conv1' — no accumulators, no conv1' buffers: 80 KiB, < 256 registers) runs two workgroups per CU
  if (direct) return bmt == 128 && cn == 0 && lds_bytes(bmt, cw, cn, has_c2, true) <= 80 * 1024 ? 2 : 1;
  return bmt == 128 && lds_bytes(bmt, cw, cn, has_c2) <= 80 * 1024 ? 2 : 1;
}

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// PATCH (conv2 forms, 128-pixel tiles): the tile is an 8 x 16 PATCH of one image instead of 128 consecutive pixels, and
// phase 1 stages the patch's (8 + 2) x (16 + 2) halo ONCE per 32-channel slice; the nine taps read it at shifted rows (the
// scheme of conv3x3_halo_f16x3).  The linear form fetches a 128-byte operand row per pixel, tap and slice: 288 of the ~720
// one-KiB vector-memory instructions a tile issues — and the chain kernels are bound by exactly that path
// (profiles/r04_probes.md section 1: same cycles with and without their MFMAs).  The halo form issues 46.  Same K order (channel
// slice outer, taps inner), same terms: bit-identical.
// Direct forms (operand fragments straight from global memory, no LDS tile): the two-source pair (cw2 > 0) and the EXPAND form
// (cn = 0, round 5): conv3 + identity of a block WITHOUT the next block's conv1 — out = relu(conv3(t1) + res) and nothing else.
// Where the pair needs one wave per SIMD (layer 3: 128 accumulator + 128 fragment registers), the expand form needs 128 + 16 and
// 80 KiB, i.e. two workgroups per CU like the forms that sit nearest their floors, and conv1' runs on the 256-row kernel.
constexpr bool direct_form(int cn, int cw2) { return cw2 > 0 || cn == 0; }
template <int CN, int CW, int NOUT, bool HAS_C2, bool HAS_RES, int BMT, bool PATCH = false, int CW2 = 0>
__global__ void __launch_bounds__(2 * BMT, wgs_per_cu(BMT, CW, CN, HAS_C2, direct_form(CN, CW2)) * BMT / 128) bneck_chain_c64(const ChainK p) {
  constexpr bool DIRECT = direct_form(CN, CW2);   // fragments loaded straight from global memory: the two-source pair (CW - CW2 channels from t1, CW2 from t1b)
  static_assert(!DIRECT || (!HAS_C2 && BMT == 128 && CW2 % 32 == 0 && CW2 < CW), "direct forms are pair forms on 128-pixel tiles");
  constexpr bool HAS_P3 = CN > 0;                   // conv1' of the next block (phase 3); false: the expand form
  static_assert(!HAS_C2 || CW == C, "phase 1 is written for 64-channel bottlenecks");
  static_assert(!PATCH || HAS_C2, "the patch form is a conv2 form");
  constexpr int PH = BMT / 16;                      // PATCH: rows of the (PH x 16)-pixel patch: 8 (4 waves) or 16 (8 waves)
  static_assert(BMT == 128 || BMT == 256, "tile height");
  constexpr int NTHR = 2 * BMT;                     // threads: one wave per 32 rows
  constexpr int NW = NTHR / 64;                     // waves
  constexpr int LR = NTHR / 8;                      // rows one DMA pass of the workgroup covers (8 rows per wave instruction)
  constexpr int STAGE = stage_bytes(BMT);
  constexpr int W1B_OFF = w1b_off(BMT);
  constexpr int TN3 = CN / 32;
  constexpr int TN3A = TN3 > 0 ? TN3 : 1;           // array extents (the expand form has no conv1' tiles)
  constexpr int CS = CW / 32;                       // K slices of conv3
  constexpr int NCH = NOUT / 32;                    // groups of 32 conv3 filters
  constexpr int W3CH = CW * 128;                    // bytes of one conv3 filter group in LDS
  constexpr bool ALIAS = alias_t2(BMT, CW, CN, HAS_C2) && !DIRECT;
  constexpr int T2_OFF = ALIAS ? W1B_OFF : r0_bytes(BMT, CW, CN, HAS_C2);
  constexpr int WGS = wgs_per_cu(BMT, CW, CN, HAS_C2, DIRECT);
  constexpr int WPS = WGS * NW / 4;                 // waves per SIMD: 2 = 256 registers per wave
  static_assert(lds_bytes(BMT, CW, CN, HAS_C2, DIRECT) <= 160 * 1024, "LDS budget");
  constexpr bool W1DB = w1_double(BMT, CN, HAS_C2);
  // The next chunk's filter DMAs: one at a time BETWEEN the phase-2 MFMAs, or as a burst at the top of the chunk.  An LDS-DMA
  // instruction holds its wave until the vector-memory path has taken it.  With ONE wave per SIMD (the one-workgroup pair
  // forms) a burst of CS + CN / 32 of them is ~1000-2000 cycles with the matrix pipe idle, and spreading them wins (layer-3
  // pair 451 vs 493 us); with TWO workgroups per CU the other workgroup's wave covers a burst, while spread instructions
  // stretch phase 2 to 5250 cycles for 768 cycles of MFMAs: the burst wins there (layer-2 pair 644 -> 525-548 us).  The
  // conv2 forms (two per CU, few filter pieces) are 1.5 % better spread.  profiles/r03_probes.md.
  constexpr bool SPREAD = W1DB && (HAS_C2 || WPS == 1);
  // Rotated chunk loop (experiment builds, FCP_CHAIN_ROT; the one-wave-per-SIMD pair forms: 512 registers per wave): phase 3
  // of chunk j - 1 is issued BETWEEN the phase-2 MFMAs of chunk j (phase 2 is one dependent chain on a single accumulator,
  // phase 3 is 6 TN3 MFMAs on TN3 independent accumulators); conv1' slice j is then fetched during chunk j and the T3
  // fragments of chunk j are read into registers at the end of chunk j.  Bit-identical (tools/fuzz_chain.py), 10 % fewer
  // cycles per chunk in the probes (8320 -> 7490) and 7 % MORE wall time on the layer-3 pair (440 -> 471 us; the
  // 128 -> 512 -> 256 pair 827-867 -> 820-837): not the default.  profiles/r03_probes.md.
  constexpr bool ROT = false;
  constexpr int PQ = ROT ? TN3 / CS : 0;                // phase-3 MFMAs behind every phase-2 MFMA (6 TN3 against 6 CS)
  static_assert(!ROT || (TN3 % CS == 0 && PQ >= 1), "rotated loop: TN3 must be a multiple of CS");
  constexpr bool W1PRE = !ROT && W1DB && !DIRECT && (HAS_C2 ? CN <= 64 || WPS == 1 : CN <= 128 && WPS == 1);   // conv1' fragments of a chunk requested under phase 2 (registers permitting)
  constexpr int W3B_OFF = w3b_off(BMT, CN, HAS_C2);
  static_assert(!HAS_C2 || W3B_OFF + 2 * W3CH <= 2 * STAGE, "chunk buffers must fit the phase-1 stage region");
  constexpr int NRES = HAS_RES ? 4 : 0;             // residual loads per chunk and thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);

  const int nb = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
  const int tile_m = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int lrow = tid >> 3;                                     // 0..LR-1 (+LR i)
  const int csrc = (tid & 7) ^ swz(lrow);                        // source chunk of LDS position tid & 7
  const int l31 = lane & 31, half = lane >> 5;
  const int rsw = swz(l31);
  int offH[2], offL[2];                                          // fragment chunk offsets inside a 128-byte row, per k-half
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    offH[s] = ((2 * s + half) ^ rsw) << 4;
    offL[s] = ((4 + 2 * s + half) ^ rsw) << 4;
  }
  const int hw = p.h * p.w;
  // tile row -> pixel index in (n, h, w) order, or -1 for a row that has no pixel (past the end / outside the image)
  int pn = 0, py0 = 0, px0 = 0;                                  // PATCH: image and top-left pixel of the 8 x 16 patch
  if constexpr (PATCH) {
    const int txn = (p.w + 15) >> 4, tyn = (p.h + PH - 1) / PH;
    const int r = tile_m / txn;
    px0 = (tile_m - r * txn) * 16;
    pn = r / tyn;
    py0 = (r - pn * tyn) * PH;
  }
  auto pix = [&](int row) -> long {
    if constexpr (PATCH) {
      const int y = py0 + (row >> 4), x = px0 + (row & 15);
      return (y < p.h && x < p.w) ? ((long)pn * p.h + y) * p.w + x : -1L;
    } else {
      const long m = (long)tile_m * BMT + row;
      return m < p.M ? m : -1L;
    }
  };

  // ---- conv2 epilogue (both phase-1 forms): fp32 tile [BMT][64] over the dead phase-1 buffers -> relu(acc * ws2 + b2) -> T2
  auto conv2_epilogue = [&](const f32x16 (&acc1)[2], int wm, int wn) {
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = wm * 64 + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        Cs[row * C + wn * 32 + l31] = acc1[i][rr];
      }
    __syncthreads();
    {
      const int ccol = (tid & 7) * 8;
      float ws8[8], b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ws8[e] = p.ws2[ccol + e];
        b8[e] = p.b2[ccol + e];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = (tid >> 3) + LR * g;
        const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + row * C + ccol);
        const f32x4 b = *reinterpret_cast<const f32x4*>(Cs + row * C + ccol + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = v[e] * ws8[e] + b8[e];
          x = x >= 0.f ? x : x * 0.f;
          v[e] = x * 1.f;
        }
        u32x4_t hi, lo;
        split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
        const int q = (ccol & 31) >> 3, sw = swz(row);
        char* trow = lds + T2_OFF + (ccol >> 5) * (BMT * ROWB) + row * ROWB;
        *reinterpret_cast<u32x4_t*>(trow + ((q ^ sw) << 4)) = hi;
        *reinterpret_cast<u32x4_t*>(trow + (((4 + q) ^ sw) << 4)) = lo;
      }
    }
    __syncthreads();                                             // T2 complete; the fp32 tile is dead
  };

  // =========================================================================================== phase 1: 3x3 conv
  if constexpr (HAS_C2 && PATCH) {
    constexpr int HALO = (PH + 2) * 18;                          // 10 x 18 = 180 | 18 x 18 = 324 halo rows ...
    constexpr int HROWS = (HALO + LR - 1) / LR * LR;             // ... padded to whole DMA passes: 192 | 384
    constexpr int ASL = HROWS * ROWB;                            // one channel slice of the halo patch: 24 KiB
    // filter stages behind the two slices: TPB taps (K slices of 8 KiB) per stage and per workgroup barrier — 2 where the
    // patch form runs two workgroups per CU in 80 KiB, 3 in the 8-wave form
    // (one tap per barrier measured equal in the 4-wave form and 3.5-4 % slower in the 8-wave form: profiles/r04_probes.md 1e)
    constexpr int TPB = BMT == 128 ? 2 : 3, NSTEP = 18 / TPB;
    static_assert(18 % TPB == 0, "taps per barrier must divide the 18 K slices");
    constexpr int BST_OFF = 2 * ASL, BSL = C * ROWB, BSTG = TPB * BSL;
    static_assert(BST_OFF + 2 * BSTG <= T2_OFF + BMT * C * 4, "halo patch + filter stages must fit region 0 + T2");
    f32x16 acc1[2];
    const int wm = wave / 2, wn = wave % 2;                      // 2 x 2 waves of 64 x 32, as in the linear form
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t1), 0, p.t1_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, p.w2_bytes, 0x00020000);
    // ---- the halo patch, both channel slices: 6 passes of 32 rows each (rows >= 180 and pixels outside the image: zero fill)
#pragma unroll
    for (int i = 0; i < HROWS / LR; ++i) {
      const int r = lrow + LR * i;
      const int hy = r / 18, hx = r - hy * 18;
      const int y = py0 - 1 + hy, x = px0 - 1 + hx;
      const bool ok = r < HALO && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
      const unsigned src = ok ? ((unsigned)((pn * p.h + y) * p.w + x) * (unsigned)p.t1_ld + (unsigned)(csrc * 4)) * 4u : 0xFFFFFFFFu;
#pragma unroll
      for (int cs = 0; cs < 2; ++cs)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(lds + cs * ASL + (wave_u * 8 + LR * i) * ROWB), 16,
                                                 (int)(src == 0xFFFFFFFFu ? 0xFFFFFFFFu : src + (unsigned)(cs * 32 * 4)), 0, 0, 0);
    }
    constexpr int B_LD = C / LR;
    unsigned woff[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) woff[i] = (unsigned)(((lrow + LR * i) * (9 * C) + csrc * 4) * 4);
    auto dma_b = [&](int step, int stage) {                      // the TPB K slices of a step
#pragma unroll
      for (int u = 0; u < TPB; ++u)
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds + BST_OFF + stage * BSTG + u * BSL + (wave_u * 8 + LR * i) * ROWB), 16,
                                                   (int)(woff[i] + (unsigned)((step * TPB + u) * BK * 4)), 0, 0, 0);
    };
    dma_b(0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
    // halo row of the lane's pixel in the wave's two 32-row tiles: pixel p = wm * 64 + 32 i + l31 -> (p >> 4, p & 15)
    int hr0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) hr0[i] = (wm * 4 + i * 2 + (l31 >> 4)) * 18 + (l31 & 15);
    const char* Bw = lds + BST_OFF + (wn * 32 + l31) * ROWB;
    static_for<0, NSTEP>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      // the filter slices of step g have landed (with the halo patch, at g = 0) and every wave has read those of step g - 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, TPB>([&](auto uc) {
        constexpr int kt = g * TPB + decltype(uc)::value, cs = kt / 9, tap = kt % 9;   // K order: channel slice outer, taps inner
        const char* Bb = Bw + (g & 1) * BSTG + decltype(uc)::value * BSL;
        f16x8 ah[2][2], al[2][2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int hr = hr0[i] + (tap / 3) * 18 + tap % 3;
          const int sw = swz(hr);
          const char* Ab = lds + cs * ASL + hr * ROWB;
#pragma unroll
          for (int sq = 0; sq < 2; ++sq) {
            ah[sq][i] = *reinterpret_cast<const f16x8*>(Ab + (((2 * sq + half) ^ sw) << 4));
            al[sq][i] = *reinterpret_cast<const f16x8*>(Ab + (((4 + 2 * sq + half) ^ sw) << 4));
          }
        }
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
          bh[sq] = *reinterpret_cast<const f16x8*>(Bb + offH[sq]);
          bl[sq] = *reinterpret_cast<const f16x8*>(Bb + offL[sq]);
        }
        if constexpr (decltype(uc)::value == TPB - 1) {
          // the step's last fragment reads are out: the next step's filters may go into the stage step g - 1 has left.
          // (Issued behind ALL LDS reads of the step: the compiler drains pending LDS-DMA in front of an LDS read.)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (g + 1 < NSTEP) dma_b(g + 1, (g + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int sq = 0; sq < 2; ++sq)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sq][i], bh[sq], acc1[i], 0, 0, 0);
            acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sq][i], bl[sq], acc1[i], 0, 0, 0);
            acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sq][i], bh[sq], acc1[i], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave has consumed the last slice: patch and stages are dead
    __builtin_amdgcn_sched_barrier(0);
    conv2_epilogue(acc1, wm, wn);
  } else if constexpr (HAS_C2) {
    f32x16 acc1[2];
    constexpr int A_LD = BMT / LR, B_LD = C / LR;
    const int wm = wave / 2, wn = wave % 2;                      // (NW / 2) x 2 waves of 64 x 32
    TapPiece tp[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int m = tile_m * BMT + lrow + LR * i;
      unsigned pbase = 0;
      int hi0 = -(1 << 28), wi0 = 0;
      if (m < p.M) {
        const int ni = m / hw;
        const int rem = m - ni * hw;
        const int ho = rem / p.w;
        pbase = (unsigned)(ni * hw);
        hi0 = ho - 1;
        wi0 = rem - ho * p.w - 1;
      }
      tp[i].base = ((pbase + (unsigned)(hi0 * p.w + wi0)) * (unsigned)p.t1_ld + (unsigned)(csrc * 4)) * 4u;
      tp[i].mask = 0u;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const bool ok = (unsigned)(hi0 + q / 3) < (unsigned)p.h && (unsigned)(wi0 + q % 3) < (unsigned)p.w;
        tp[i].mask |= ok ? (1u << q) : 0u;
      }
    }
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t1), 0, p.t1_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, p.w2_bytes, 0x00020000);
    unsigned woff[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) woff[i] = (unsigned)(((lrow + LR * i) * (9 * C) + csrc * 4) * 4);
    unsigned rowoff[A_LD];
    auto set_tap = [&](int tap, int kh_i, int kw_i) {
      const unsigned tapoff = (unsigned)((kh_i * p.w + kw_i) * p.t1_ld) * 4u;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) rowoff[i] = ((tp[i].mask >> tap) & 1u) ? tp[i].base + tapoff : 0xFFFFFFFFu;
    };
    int tap = 0, kh_i = 0, kw_i = 0, c0 = 0;
    auto advance = [&]() {
      ++tap;
      if (++kw_i >= 3) {
        kw_i = 0;
        if (++kh_i >= 3) { kh_i = 0; tap = 0; c0 += BK; }
      }
      set_tap(tap, kh_i, kw_i);
    };
    auto dma_slice = [&](int kt, int stage) {
      char* a = lds + stage * STAGE + wave_u * 8 * ROWB;
      char* b = a + BMT * ROWB;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const unsigned ro = rowoff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + LR * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(c0 * 4)), 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b + LR * i * ROWB), 16,
                                                 (int)(woff[i] + (unsigned)(kt * BK * 4)), 0, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;

    const int aoff = (wm * 64 + l31) * ROWB;
    const int boff = BMT * ROWB + (wn * 32 + l31) * ROWB;
    constexpr int KT = 9 * C / 32;                               // 18 K slices
    // Two stages: the next slice is issued inside the iteration and waited for at its end.  (FCP_CHAIN_C2_STAGES3: three
    // stages, slices fetched TWO ahead — the third stage costs no LDS, it lies in the T2 region, which is only written by the
    // conv2 epilogue — measured equal, 1327-1341 vs 1328-1340 us: with two workgroups per CU the other one covers the DMA
    // round trip already; profiles/r03_probes.md.)
    constexpr int NST = 2;
    static_assert(NST * STAGE <= T2_OFF + BMT * C * 4, "phase-1 stages must fit region 0 + T2");
    set_tap(0, 0, 0);
    dma_slice(0, 0);
    if constexpr (NST == 3) {
      advance();
      dma_slice(1, 1);
    }
    int stage = 0;
    const int kt_end = KT;
    for (int kt = 0; kt < kt_end; ++kt) {
      // slice kt has landed (the one or two younger slices may still fly) and every wave is done with slice kt - 1
      if (NST == 3 && kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* Ab = lds + stage * STAGE + aoff;
      const char* Bb = lds + stage * STAGE + boff;
      f16x8 ah[2][2], al[2][2], bh[2], bl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[s][i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * ROWB + offH[s]);
          al[s][i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * ROWB + offL[s]);
        }
        bh[s] = *reinterpret_cast<const f16x8*>(Bb + offH[s]);
        bl[s] = *reinterpret_cast<const f16x8*>(Bb + offL[s]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NST - 1 < KT) {                                   // into the stage slice kt - 1 has just left
        advance();
        const int nstage = stage == 0 ? NST - 1 : stage - 1;     // (stage + NST - 1) % NST
        dma_slice(kt + NST - 1, nstage);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s], acc1[i], 0, 0, 0);
          acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s], acc1[i], 0, 0, 0);
          acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s], acc1[i], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NST ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave has consumed the last slice: the stages are dead
    __builtin_amdgcn_sched_barrier(0);

    conv2_epilogue(acc1, wm, wn);
  } else if constexpr (!DIRECT) {
    // ---- no conv2: the operand tile is the input itself (CS slices of 128 pixels x 128 B), by LDS-DMA
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t1), 0, p.t1_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < BMT / LR; ++i) {
      const int m = tile_m * BMT + lrow + LR * i;
      const unsigned ro = m < p.M ? ((unsigned)m * (unsigned)p.t1_ld + (unsigned)(csrc * 4)) * 4u : 0xFFFFFFFFu;
#pragma unroll
      for (int sl = 0; sl < CS; ++sl)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(lds + T2_OFF + sl * BMT * ROWB + wave_u * 8 * ROWB + LR * i * ROWB),
                                                 16, (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(sl * 128)), 0, 0, 0);
    }
  }

  // ========================================================================== chunk loop: conv3 (+x, relu) and conv1'
  __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, p.w3_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1n), 0, p.w1n_bytes, 0x00020000);
  // conv3 filter group j: 32 rows x CS K slices = 4 CS pieces of 8 rows.  Wave w moves pieces w PW3 .. w PW3 + PW3 - 1.
  constexpr int PW3 = 4 * CS / NW;                  // per wave: CS (4 waves) | CS / 2 (8 waves)
  constexpr int PW1 = CN / (8 * NW);                // conv1' slice: CN rows, 8 per wave instruction
  static_assert(PW3 >= 1 && (PW1 >= 1 || !HAS_P3), "filter pieces per wave");
  auto dma_w3 = [&](int j, int buf) {
#pragma unroll
    for (int i = 0; i < PW3; ++i) {
      const int g = wave_u * PW3 + i, sl = g >> 2, r = (g & 3) * 8 + (lane >> 3);
      char* dst = lds + W3B_OFF + buf * W3CH + sl * 4096 + (g & 3) * 8 * ROWB;
      const unsigned src = (unsigned)((j * 32 + r) * (CW * 4) + sl * 128 + (((lane & 7) ^ swz(r)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (__attribute__((address_space(3))) void*)dst, 16, (int)src, 0, 0, 0);
    }
  };
  // conv1' K slice j: CN rows.  Wave w moves rows w * CN/NW + 8 i + lane / 8.
  auto dma_w1 = [&](int j, int buf) {
    char* dst = lds + W1B_OFF + buf * (CN * ROWB) + wave_u * (CN / NW) * ROWB;
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int r = wave_u * (CN / NW) + 8 * i + (lane >> 3);
      const unsigned src = (unsigned)(r * (NOUT * 4) + j * 128 + (((lane & 7) ^ swz(r)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (__attribute__((address_space(3))) void*)(dst + 8 * i * ROWB), 16, (int)src, 0, 0, 0);
    }
  };
  // instruction k of {filter group j, conv1' slice j} (PW3 + PW1 per wave), for issue between the phase-2 MFMAs
  constexpr int NDMA = PW3 + PW1;
  auto dma_one = [&](auto kc, int j, int buf) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < PW3) {
      const int g = wave_u * PW3 + k, sl = g >> 2, r = (g & 3) * 8 + (lane >> 3);
      char* dst = lds + W3B_OFF + buf * W3CH + sl * 4096 + (g & 3) * 8 * ROWB;
      const unsigned src = (unsigned)((j * 32 + r) * (CW * 4) + sl * 128 + (((lane & 7) ^ swz(r)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (__attribute__((address_space(3))) void*)dst, 16, (int)src, 0, 0, 0);
    } else {
      constexpr int i = k - PW3;
      const int j1 = ROT ? j - 1 : j, buf1 = ROT ? (j1 & 1) : buf;       // rotated loop: slice j - 1 of the "next" index j
      char* dst = lds + W1B_OFF + buf1 * (CN * ROWB) + wave_u * (CN / NW) * ROWB;
      const int r = wave_u * (CN / NW) + 8 * i + (lane >> 3);
      const unsigned src = (unsigned)(r * (NOUT * 4) + j1 * 128 + (((lane & 7) ^ swz(r)) << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (__attribute__((address_space(3))) void*)(dst + 8 * i * ROWB), 16, (int)src, 0, 0, 0);
    }
  };

  f32x16 acc3[TN3A];
#pragma unroll
  for (int t = 0; t < TN3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc3[t][e] = 0.f;

  const char* a2base = lds + T2_OFF + (wave * 32 + l31) * ROWB;   // + slice * BM * ROWB
  const char* a3base = lds + CT_OFF + (wave * 32 + l31) * ROWB;
  // Epilogue items of this thread: rows erow0, erow0 + 16 OF THE WAVE'S OWN 32 ROWS, channel group eq (8 channels) of the
  // 32-channel chunk.  A wave stages, finishes and re-reads (phase 3) only its own rows of the epilogue tile, so the three
  // steps of a chunk need no workgroup barrier between them and the four waves may drift apart inside a chunk (one wave's
  // MFMAs beside another's epilogue); what the waves share are the filter buffers: ONE barrier per chunk.
  const int eq = lane & 3;
  const int erow0 = wave * 32 + (lane >> 2);
  long rm[2];                                                    // residual pixel of the two items (clamped)
  unsigned so[2];                                                // byte offset of the two items in `out`, 0xFFFFFFFF past the end
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int prow = erow0 + 16 * it;
    const long m = pix(prow);
    rm[it] = m >= 0 ? m : (long)p.M - 1;
    // out_even: a stride-2 consumer reads even (y, x) only — patch origins are even, so the parity is that of the tile row;
    // the dropped stores still issue (exact vmcnt), the hardware discards them: no HBM traffic
    const bool keep = !PATCH || !p.out_even || ((((prow >> 4) | prow) & 1) == 0);
    so[it] = (m >= 0 && keep) ? (unsigned)(m * p.out_ld * 4 + eq * 16) : 0xFFFFFFFFu;
  }
  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

  // ---- vector-memory issue order.  vmcnt retires IN ORDER and an `out` store to HBM takes microseconds to be
  // acknowledged, so no wait may sit behind a store it does not need.  Per chunk j (all counts per thread):
  //     top(j):   wait for D(j) = {conv3 filter group j, conv1' slice j}; barrier; issue D(j+1) into the other buffers
  //     phase 2, epilogue (consumes the residual R(j)); issue c(j+1) x2 (the lane's scale / bias), R(j+1) x NRES
  //     phase 3; issue the stores S(j) x4 through a buffer resource (rows past the end are dropped by the hardware:
  //     the count is exact)
  // D(j+1) is a whole chunk ahead of its use and the ops younger than it at top(j+1) are exactly c(j+1), R(j+1), S(j):
  // vmcnt(6 + NRES) waits for the DMAs without touching the stores, which are first forced two chunks after their
  // issue.  Where LDS has no room for a second conv1' buffer (W1DB false) slice j is issued at top(j) and a second
  // wait + barrier in front of phase 3 drains S(j-1) with it.
  u32x4_t rhi[2], rlo[2];
  auto load_res = [&](int j) {
    if constexpr (!HAS_RES) return;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const char* pb = reinterpret_cast<const char*>(p.res) + rm[it] * p.res_ld * 4 + j * 128 + eq * 16;
      rhi[it] = *reinterpret_cast<const u32x4_t*>(pb);
      rlo[it] = *reinterpret_cast<const u32x4_t*>(pb + 64);
    }
  };
  const int nch = NCH;
  f16x8 pch[2], pcl[2];                                          // rotated loop: T3 fragments of the previous chunk, per k-half
  f16x8 wx[ROT ? TN3 : 1], wy[ROT ? TN3 : 1];                    // rotated loop: conv1' fragments of the previous chunk's slice
  float ws_l = p.ws3[l31], b_l = p.b3[l31];                      // this lane's conv3 channel of chunk 0 (MFMA layout: col = lane & 31)
  constexpr int NCONST = 2;
  f16x8 ah[CS][2], al[CS][2];                                    // phase-2 A fragments: the wave's own 32 rows of T2, [slice][k-half]
  auto read_a2 = [&]() {
#pragma unroll
    for (int sl = 0; sl < CS; ++sl)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        ah[sl][s] = *reinterpret_cast<const f16x8*>(a2base + sl * BMT * ROWB + offH[s]);
        al[sl][s] = *reinterpret_cast<const f16x8*>(a2base + sl * BMT * ROWB + offL[s]);
      }
  };
  if constexpr (DIRECT) {
    // ---- the wave's operand fragments straight from the two source tensors: lane (l31, half) of k-half s of slice sl needs
    //      the 16-byte pieces (2 s + half) [hi] and (4 + 2 s + half) [lo] of its pixel's 128-byte channel-slice record — the
    //      split32 format IS the fragment layout.  Rows past the end read zeros (out-of-range buffer offsets).
    constexpr int CS1 = (CW - CW2) / 32;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t1), 0, p.t1_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t1b), 0, p.t1b_bytes, 0x00020000);
    // chunk 0's filters first: their L2 round trip runs under the fragments' HBM round trip (one wait for both)
    asm volatile("" ::: "memory");
    dma_w3(0, 0);
    if constexpr (W1DB && !ROT) dma_w1(0, 0);
    asm volatile("" ::: "memory");
    const long m = pix(wave * 32 + l31);
    unsigned oa = 0xFFFFFFFFu, ob = 0xFFFFFFFFu;
    if (m >= 0) {
      const int ni = (int)(m / hw);
      const int rem = (int)(m - (long)ni * hw);
      const int y = rem / p.w, x = rem - y * p.w;
      oa = (unsigned)m * (unsigned)p.t1_ld * 4u;
      ob = ((unsigned)(ni * p.hb + y * p.sb) * (unsigned)p.wb + (unsigned)(x * p.sb)) * (unsigned)p.t1b_ld * 4u;
    }
#pragma unroll
    for (int sl = 0; sl < CS; ++sl)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bool first = sl < CS1;                                 // compile-time after unrolling
        const unsigned base = first ? oa : ob;
        const unsigned off = (unsigned)((first ? sl : sl - CS1) * 128);
        const unsigned vh = base == 0xFFFFFFFFu ? base : base + off + (unsigned)((2 * sk + half) << 4);
        const unsigned vl = base == 0xFFFFFFFFu ? base : base + off + (unsigned)((4 + 2 * sk + half) << 4);
        ah[sl][sk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(first ? rs_a : rs_b, (int)vh, 0, 0));
        al[sl][sk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(first ? rs_a : rs_b, (int)vl, 0, 0));
      }
    // all of them landed HERE, outside the chunk loop: a fragment the compiler still believed in flight at the loop header
    // would get a conservative vmcnt wait in front of its first use in EVERY chunk, draining the loop's counted pipeline
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int sl = 0; sl < CS; ++sl)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) asm volatile("" : "+v"(ah[sl][sk]), "+v"(al[sl][sk]));
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (ALIAS) {                                         // T2 shares LDS with the chunk buffers: fragments first, filters after
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_a2();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" ::: "memory");
  if constexpr (!DIRECT) {
    dma_w3(0, 0);
    if constexpr (W1DB && !ROT) dma_w1(0, 0);
  }
  asm volatile("" ::: "memory");
  if (nch > 0) load_res(0);
  __builtin_amdgcn_sched_barrier(0);

  for (int j = 0; j < nch; ++j) {
    const bool more = j + 1 < NCH;
    // ---- top: this chunk's filters have landed (everything younger may fly), every wave is done with chunk j - 1
    // (W1DB false: group j was issued at top(j-1) and already forced by the wait in front of phase 3 of chunk j-1; the
    //  ops younger than that wait — c(j), R(j), S(j-1) — may all stay in flight, which is the same count)
    if (j == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NCONST + NRES) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!SPREAD) {
      if (more) dma_w3(j + 1, (j + 1) & 1);
      if constexpr (W1DB) {
        if (more) dma_w1(j + 1, (j + 1) & 1);
      } else {
        dma_w1(j, 0);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2 operands: T2 (A: the wave's own 32 rows, the same for every chunk — read once, kept in registers) and
    //      filter group j (B), all K slices
    const char* b2base = lds + W3B_OFF + (j & 1) * W3CH + l31 * ROWB;
    if constexpr (!ALIAS && !DIRECT) {
      if (j == 0) read_a2();
    }
    constexpr int BG = (CS <= 4 && (HAS_C2 || WPS == 1)) ? CS : 2;   // slices of filter fragments in flight (all of them up to CS = 4, registers permitting)
    f16x8 bh[2][BG][2], bl[2][BG][2];                            // [buffer][slice in group][k-half]
    auto read_b2 = [&](int buf, int g) {
#pragma unroll
      for (int q = 0; q < BG; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bh[buf][q][s] = *reinterpret_cast<const f16x8*>(b2base + (g * BG + q) * 4096 + offH[s]);
          bl[buf][q][s] = *reinterpret_cast<const f16x8*>(b2base + (g * BG + q) * 4096 + offL[s]);
        }
    };
    read_b2(0, 0);
    // conv1' fragments of this chunk (W1DB: slice j landed with filter group j): requested behind the phase-2 operands,
    // they arrive under the phase-2 MFMAs and phase 3 starts with only its two T3 fragments per k-half to wait for
    f16x8 dh[2][TN3A], dl[2][TN3A];                              // [k-half][column tile]
    if constexpr (W1PRE) {
      const char* w1b = lds + W1B_OFF + (j & 1) * (CN * ROWB);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < TN3; ++t) {
          dh[s][t] = *reinterpret_cast<const f16x8*>(w1b + (t * 32 + l31) * ROWB + offH[s]);
          dl[s][t] = *reinterpret_cast<const f16x8*>(w1b + (t * 32 + l31) * ROWB + offL[s]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
    // rotated loop: phase 3 of chunk j - 1 rides between the phase-2 MFMAs.  Its conv1' fragments live in two register sets
    // X, Y of TN3: k-half 0 needs dh0 (terms 0, 2) and dl0 (term 1), k-half 1 dh1 and dl1; X = dh0 -> dl1, Y = dl0 -> dh1,
    // each reloaded right behind the last MFMA that takes its old contents, TN3 MFMAs before its next use.
    const bool prev = ROT && j > 0;
    const char* w1p = lds + W1B_OFF + ((j + 1) & 1) * (CN * ROWB) + l31 * ROWB;     // conv1' slice j - 1
    auto p3_read = [&](f16x8 (&dst)[ROT ? TN3 : 1], int off) {
#pragma unroll
      for (int t = 0; t < (ROT ? TN3 : 0); ++t) dst[t] = *reinterpret_cast<const f16x8*>(w1p + t * 32 * ROWB + off);
    };
    if constexpr (ROT) {
      if (prev) {
        p3_read(wx, offH[0]);
        p3_read(wy, offL[0]);
      }
    }
    auto p3_step = [&](auto mc) {                                  // MFMA m3 of phase 3 (chunk j - 1), term-major inside a k-half
      constexpr int m3 = decltype(mc)::value;
      constexpr int s3 = m3 / (3 * TN3), term = (m3 % (3 * TN3)) / TN3, t = m3 % TN3;
      if constexpr (m3 == 2 * TN3) p3_read(wy, offH[1]);           // dl0 is dead: Y <- dh1
      if constexpr (m3 == 3 * TN3) p3_read(wx, offL[1]);           // dh0 is dead: X <- dl1
      if constexpr (s3 == 0) {
        if constexpr (term == 0) acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pcl[0], wx[t], acc3[t], 0, 0, 0);
        else if constexpr (term == 1) acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pch[0], wy[t], acc3[t], 0, 0, 0);
        else acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pch[0], wx[t], acc3[t], 0, 0, 0);
      } else {
        if constexpr (term == 0) acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pcl[1], wy[t], acc3[t], 0, 0, 0);
        else if constexpr (term == 1) acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pch[1], wx[t], acc3[t], 0, 0, 0);
        else acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pch[1], wy[t], acc3[t], 0, 0, 0);
      }
    };
    static_for<0, CS / BG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 1 < CS / BG) read_b2((g + 1) & 1, g + 1);  // next group's fragments under this group's MFMAs
      static_for<0, 2 * BG>([&](auto tc) {
        constexpr int q = decltype(tc)::value / 2, s = decltype(tc)::value % 2, sl = g * BG + q, trip = 2 * sl + s;
        static_for<0, 3>([&](auto ec) {
          constexpr int term = decltype(ec)::value;
          if constexpr (term == 0) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sl][s], bh[g & 1][q][s], acc2, 0, 0, 0);
          else if constexpr (term == 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl][s], bl[g & 1][q][s], acc2, 0, 0, 0);
          else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl][s], bh[g & 1][q][s], acc2, 0, 0, 0);
          if constexpr (ROT) {
            __builtin_amdgcn_sched_barrier(0);
            if (prev) static_for<0, PQ>([&](auto pc) { p3_step(std::integral_constant<int, (3 * trip + term) * PQ + decltype(pc)::value>{}); });
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        constexpr int PER = (NDMA + 2 * CS - 1) / (2 * CS);       // DMA instructions per MFMA triple (1, or 2 where CN > 32 CS)
        if constexpr (SPREAD && trip * PER < NDMA) {
          __builtin_amdgcn_sched_barrier(0);
          // rotated loop: the conv1' pieces fetch slice j (consumed by chunk j + 1's phase 3 ride): also in the last chunk
          if (more || (ROT && trip * PER >= PW3)) dma_one(std::integral_constant<int, trip * PER>{}, j + 1, (j + 1) & 1);
          if constexpr (PER == 2 && trip * PER + 1 < NDMA) {
            if (more || (ROT && trip * PER + 1 >= PW3)) dma_one(std::integral_constant<int, trip * PER + 1>{}, j + 1, (j + 1) & 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(PER <= 2, "at most two DMA instructions per MFMA triple");
      });
    });
    // ---- acc2 * ws3 + b3 (per lane: one channel) -> the wave's rows of the fp32 tile.  Channel group q of a row is
    //      stored in the two 16-byte pieces the split32 image of that group will occupy (hi piece q ^ sw, lo piece
    //      (4 + q) ^ sw), so the epilogue rewrites each item in place.
    {
      const int q = l31 >> 3;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = wave * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        const int piece = ((l31 & 4) ? (4 + q) : q) ^ swz(row);
        *reinterpret_cast<float*>(lds + CT_OFF + row * ROWB + (piece << 4) + (l31 & 3) * 4) = acc2[rr] * ws_l + b_l;
      }
    }
    // ---- epilogue of conv3 for this chunk (own rows: LDS accesses of one wave execute in order, no barrier):
    //      out = relu(. + x) -> registers (stored in phase 3) and T3 (in place)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HAS_RES)     // keep the residual's first use HERE: hoisted into phase 2 it would be waited for a phase early
      asm volatile("" : "+v"(rhi[0]), "+v"(rlo[0]), "+v"(rhi[1]), "+v"(rlo[1]));
    u32x4_t ohi[2], olo[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = erow0 + 16 * it;
      const int sw = swz(row);
      char* crow = lds + CT_OFF + row * ROWB;
      const f32x4 a = *reinterpret_cast<const f32x4*>(crow + ((eq ^ sw) << 4));
      const f32x4 b = *reinterpret_cast<const f32x4*>(crow + (((4 + eq) ^ sw) << 4));
      float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      float r[8];
      if constexpr (HAS_RES) join8(rhi[it], rlo[it], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e];
        if constexpr (HAS_RES) x += r[e];
        x = x >= 0.f ? x : x * 0.f;
        v[e] = x * 1.f;
      }
      split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, ohi[it], olo[it]);
      if constexpr (HAS_P3) {                                      // ... and T3, conv1's operand (the expand form has no phase 3)
        *reinterpret_cast<u32x4_t*>(crow + ((eq ^ sw) << 4)) = ohi[it];
        *reinterpret_cast<u32x4_t*>(crow + (((4 + eq) ^ sw) << 4)) = olo[it];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (more) {                                                  // a chunk ahead: the lane's channel constants, the residual
      ws_l = p.ws3[(j + 1) * 32 + l31];
      b_l = p.b3[(j + 1) * 32 + l31];
      load_res(j + 1);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!W1DB) {                                       // single conv1' buffer: slice j was issued at the top
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCONST + NRES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    auto store_out = [&]() {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const unsigned o = so[it] == 0xFFFFFFFFu ? 0xFFFFFFFFu : so[it] + (unsigned)(j * 128);
        __builtin_amdgcn_raw_buffer_store_b128(ohi[it], rs_out, o, 0, FCP_CHAIN_STORE_AUX);     // aux 2: nt (streamed once)
        __builtin_amdgcn_raw_buffer_store_b128(olo[it], rs_out, o == 0xFFFFFFFFu ? o : o + 64u, 0, FCP_CHAIN_STORE_AUX);
      }
    };
    if constexpr (!HAS_P3) {
      store_out();                                                 // expand form: the chunk ends with its stores
    } else if constexpr (ROT) {
      // rotated loop: T3 of this chunk goes to registers now (the tile is overwritten by the next chunk's staging); its
      // phase 3 rides in the next chunk's phase 2.  The chunk's `out` stores go out behind the fragment reads.
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        pch[s] = *reinterpret_cast<const f16x8*>(a3base + offH[s]);
        pcl[s] = *reinterpret_cast<const f16x8*>(a3base + offL[s]);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      store_out();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    } else {
    // ---- phase 3: acc3 += T3 . W1'[:, slice j]^T, one k-half at a time (fragment registers: CN = 128 has none to spare);
    //      this chunk's `out` stores go out behind the first k-half's fragment reads
    const char* w1base = lds + W1B_OFF + (W1DB ? (j & 1) * (CN * ROWB) : 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 ch, cl;
      ch = *reinterpret_cast<const f16x8*>(a3base + offH[s]);
      cl = *reinterpret_cast<const f16x8*>(a3base + offL[s]);
      if constexpr (!W1PRE) {
#pragma unroll
        for (int t = 0; t < TN3; ++t) {
          dh[s][t] = *reinterpret_cast<const f16x8*>(w1base + (t * 32 + l31) * ROWB + offH[s]);
          dl[s][t] = *reinterpret_cast<const f16x8*>(w1base + (t * 32 + l31) * ROWB + offL[s]);
        }
      }
      if (s == 0) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        store_out();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < TN3; ++t) {
        acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, dh[s][t], acc3[t], 0, 0, 0);
        acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, dl[s][t], acc3[t], 0, 0, 0);
        acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, dh[s][t], acc3[t], 0, 0, 0);
      }
    }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (ROT) {
    // phase 3 of the last chunk: its conv1' slice was fetched during the last chunk and has landed (wait + barrier above)
    if (nch > 0) {
      const char* w1l = lds + W1B_OFF + ((NCH - 1) & 1) * (CN * ROWB) + l31 * ROWB;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int t = 0; t < TN3; ++t) {
          wx[t] = *reinterpret_cast<const f16x8*>(w1l + t * 32 * ROWB + offH[s]);
          wy[t] = *reinterpret_cast<const f16x8*>(w1l + t * 32 * ROWB + offL[s]);
        }
#pragma unroll
        for (int t = 0; t < TN3; ++t) {
          acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pcl[s], wx[t], acc3[t], 0, 0, 0);
          acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pch[s], wy[t], acc3[t], 0, 0, 0);
          acc3[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pch[s], wx[t], acc3[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                               // the conv1' buffers are reused by the fp32 tile below
  }

  // ================================================================================ conv1' epilogue -> t1' (HBM)
  float* Cs = smem;                                              // fp32 tile [BMT][64], one 64-column half at a time
#pragma unroll
  for (int hh = 0; hh < CN / 64; ++hh) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = wave * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        Cs[row * 64 + t * 32 + l31] = acc3[CN == 64 ? t : 2 * hh + t][rr];
      }
    __syncthreads();
    const int ccol = (tid & 7) * 8;
    const int co = hh * 64 + ccol;
    float ws8[8], b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ws8[e] = p.ws1n[co + e];
      b8[e] = p.b1n[co + e];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = (tid >> 3) + LR * g;
      const long m = pix(row);
      const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + row * 64 + ccol);
      const f32x4 b = *reinterpret_cast<const f32x4*>(Cs + row * 64 + ccol + 4);
      float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e] * ws8[e] + b8[e];
        x = x >= 0.f ? x : x * 0.f;
        v[e] = x * 1.f;
      }
      if (m < 0) continue;
      u32x4_t hi, lo;
      split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
      char* ob = reinterpret_cast<char*>(p.t1n) + m * p.t1n_ld * 4 + split_chan_off(co);
      if (p.nt_store) {
        __builtin_nontemporal_store(hi, reinterpret_cast<u32x4_t*>(ob));
        __builtin_nontemporal_store(lo, reinterpret_cast<u32x4_t*>(ob + 64));
      } else {
        *reinterpret_cast<u32x4_t*>(ob) = hi;
        *reinterpret_cast<u32x4_t*>(ob + 64) = lo;
      }
    }
    __syncthreads();
  }
}

template <int CN, int CW, int NOUT, bool HAS_C2, bool HAS_RES, int BMT, bool PATCH = false, int CW2 = 0>
int launch(const ChainK& k, hipStream_t s) {
  constexpr int LDS = lds_bytes(BMT, CW, CN, HAS_C2, direct_form(CN, CW2));
  FCP_LDS_OPT_IN((&bneck_chain_c64<CN, CW, NOUT, HAS_C2, HAS_RES, BMT, PATCH, CW2>), LDS);
  const int tiles = PATCH ? k.n * fcp_cdiv(k.h, BMT / 16) * ((k.w + 15) >> 4) : fcp_cdiv(k.M, BMT);
  hipLaunchKernelGGL((bneck_chain_c64<CN, CW, NOUT, HAS_C2, HAS_RES, BMT, PATCH, CW2>), dim3(tiles), dim3(2 * BMT), LDS, s, k);
  FCP_LAUNCH_OK();
  return 0;
}

}  // namespace

extern "C" int fcp_bottleneck_chain_f16x3(const fcp_chain_desc* d, fcp_stream_t stream) {
  FCP_REQUIRE(d != nullptr, "chain: null descriptor");
  const bool has_p3 = d->cn != 0 || d->w1n != nullptr;          // cn = 0 and no conv1' filter: the expand form (conv3 + identity only)
  FCP_REQUIRE(d->t1 && d->w3 && d->ws3 && d->b3 && d->out && (!has_p3 || (d->w1n && d->ws1n && d->b1n && d->t1n)),
              "chain: null pointer (every convolution carries folded-BN bias and filter scales)");
  const bool has_c2 = d->w2 != nullptr;
  FCP_REQUIRE(!has_c2 || (d->ws2 && d->b2), "chain: conv2 needs its scales and bias");
  // supported shapes: (c 64, conv2, nout 256, residual, cn 64 | 128)   (c 128, no conv2, nout 512, residual, cn 128)
  //                   (c 128, no conv2, nout 256, no residual, cn 64)   (c 256, no conv2, nout 1024, residual, cn 256)
  //                   (c 128, no conv2, nout 512, residual, cn 256)
  const int variant = (has_c2 && d->c == 64 && d->nout == 256 && d->res && (d->cn == 64 || d->cn == 128)) ? (d->cn == 64 ? 1 : 2)
                    : (!has_c2 && d->c == 128 && d->nout == 512 && d->res && d->cn == 128) ? 3
                    : (!has_c2 && d->c == 128 && d->nout == 256 && !d->res && d->cn == 64) ? 4
                    : (!has_c2 && d->c == 256 && d->nout == 1024 && d->res && d->cn == 256) ? 5
                    : (!has_c2 && d->c == 128 && d->nout == 512 && d->res && d->cn == 256) ? 6
                    : (!has_c2 && d->t1b && d->c == 384 && d->cb == 256 && d->nout == 512 && !d->res && d->cn == 128) ? 7
                    : (!has_c2 && !d->t1b && d->c == 256 && d->nout == 1024 && d->res && d->cn == 0 && !d->w1n) ? 8 : 0;
  FCP_REQUIRE(variant == 7 || !d->t1b, "chain: a second source (t1b) exists for the two-source pair form only (c 384 = 128 + 256, nout 512, cn 128)");
  FCP_REQUIRE(variant != 0, "chain: unsupported shape (c %d, conv2 %d, nout %d, residual %d, cn %d)", d->c, (int)has_c2, d->nout,
              d->res != nullptr, d->cn);
  FCP_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "chain: bad sizes");
  const long M = (long)d->n * d->h * d->w;
  FCP_REQUIRE(M < (1L << 31), "chain: too many pixels");
  auto aligned = [](const void* p, int ld) { return ((uintptr_t)p & 127) == 0 && ld % 32 == 0; };
  FCP_REQUIRE(aligned(d->t1, d->t1_ld) && (!d->res || aligned(d->res, d->res_ld)) && aligned(d->out, d->out_ld) && (!has_p3 || aligned(d->t1n, d->t1n_ld)),
              "chain: tensors are split32 views: 128-byte aligned, channel stride a multiple of 32");
  FCP_REQUIRE(d->t1_ld >= d->c - (d->t1b ? d->cb : 0) && (!d->res || d->res_ld >= d->nout) && d->out_ld >= d->nout && (!has_p3 || d->t1n_ld >= d->cn), "chain: channel strides too small");
  const unsigned long t1_bytes = (unsigned long)M * d->t1_ld * 4ul;
  FCP_REQUIRE(t1_bytes < 0xFFFFFFF0ul, "chain: t1 must be below 4 GiB");
  ChainK k;
  k.t1 = d->t1; k.t1_bytes = (unsigned)t1_bytes; k.t1_ld = d->t1_ld;
  k.w2 = reinterpret_cast<const float*>(d->w2); k.w2_bytes = 128u * 9 * 64 * 4; k.ws2 = d->ws2; k.b2 = d->b2;
  k.w3 = reinterpret_cast<const float*>(d->w3); k.w3_bytes = (unsigned)(d->nout * d->c * 4); k.ws3 = d->ws3; k.b3 = d->b3;
  k.res = d->res; k.res_ld = d->res_ld; k.out = d->out; k.out_ld = d->out_ld;
  const unsigned long out_bytes = (unsigned long)M * d->out_ld * 4ul;
  FCP_REQUIRE(out_bytes < 0xFFFFFFF0ul, "chain: out must span less than 4 GiB");
  k.out_bytes = (unsigned)out_bytes;
  k.w1n = reinterpret_cast<const float*>(d->w1n); k.w1n_bytes = has_p3 ? (unsigned)(fcp_cdiv(d->cn, 128) * 128 * d->nout * 4) : 0u; k.ws1n = d->ws1n; k.b1n = d->b1n;
  k.t1n = d->t1n; k.t1n_ld = d->t1n_ld;
  k.n = d->n; k.h = d->h; k.w = d->w; k.M = (int)M;
  static const int nt_env = getenv("FCP_NT_STORE") ? atoi(getenv("FCP_NT_STORE")) : 1;
  k.nt_store = nt_env;
  k.t1b = nullptr; k.t1b_bytes = 0; k.t1b_ld = 0; k.hb = 0; k.wb = 0; k.sb = 1;
  if (d->t1b) {
    FCP_REQUIRE(aligned(d->t1b, d->t1b_ld) && d->t1b_ld >= d->cb && d->t1b_stride >= 1 && d->t1b_h > 0 && d->t1b_w > 0 &&
                (long)(d->h - 1) * d->t1b_stride < d->t1b_h && (long)(d->w - 1) * d->t1b_stride < d->t1b_w,
                "chain: t1b must be a split32 view whose (h, w) grid covers the output grid at t1b_stride");
    const unsigned long tb = (unsigned long)d->n * d->t1b_h * d->t1b_w * d->t1b_ld * 4ul;
    FCP_REQUIRE(tb < 0xFFFFFFF0ul, "chain: t1b must be below 4 GiB");
    k.t1b = d->t1b; k.t1b_bytes = (unsigned)tb; k.t1b_ld = d->t1b_ld; k.hb = d->t1b_h; k.wb = d->t1b_w; k.sb = d->t1b_stride;
  }
  k.out_even = (d->flags & FCP_CHAIN_OUT_EVEN_ONLY) ? 1 : 0;
  FCP_REQUIRE(!k.out_even || d->tile_m == 16 || d->tile_m == 32, "chain: FCP_CHAIN_OUT_EVEN_ONLY needs a patch form (tile_m = 16 | 32)");
  hipStream_t s = (hipStream_t)stream;
  // tile height: 128 pixels / 4 waves (two independent workgroups per CU drift against each other: one's MFMAs beside the
  // other's epilogue); d->tile_m = 256 asks for the 8-wave form where the operand tile fits LDS and two waves per SIMD fit
  // the registers (same bits; measured 0-6 % slower: what halving the filter traffic gains, one barrier domain of eight
  // waves loses — profiles/r03_probes.md)
  const bool big = d->tile_m == 256;
  // d->tile_m = 16: the conv2 forms on 8 x 16 patches with a staged halo (PATCH; same bits)
  // d->tile_m = 32: the same with 16 x 16 patches on 8 waves (one workgroup per CU; every filter byte serves 256 pixels)
  const bool patch = d->tile_m == 16, patch2 = d->tile_m == 32;
  FCP_REQUIRE(!(patch || patch2) || has_c2, "chain: tile_m = 16 / 32 (pixel patches) exist for the conv2 forms only");
  switch (variant) {
    case 1: return patch2 ? launch<64, 64, 256, true, true, 256, true>(k, s) : patch ? launch<64, 64, 256, true, true, 128, true>(k, s) : big ? launch<64, 64, 256, true, true, 256>(k, s) : launch<64, 64, 256, true, true, 128>(k, s);
    case 2: return patch2 ? launch<128, 64, 256, true, true, 256, true>(k, s) : patch ? launch<128, 64, 256, true, true, 128, true>(k, s) : big ? launch<128, 64, 256, true, true, 256>(k, s) : launch<128, 64, 256, true, true, 128>(k, s);
    case 3: return big ? launch<128, 128, 512, false, true, 256>(k, s) : launch<128, 128, 512, false, true, 128>(k, s);
    case 5: return launch<256, 256, 1024, false, true, 128>(k, s);
    case 6: return launch<256, 128, 512, false, true, 128>(k, s);     // CN = 256: 128 accumulator registers, one wave per SIMD only
    case 7: return launch<128, 384, 512, false, false, 128, false, 256>(k, s);   // [conv2 out 128 | x(::2, ::2) 256] -> 512 -> 128
    case 8: return launch<0, 256, 1024, false, true, 128>(k, s);                  // expand form: conv3 256 -> 1024 + identity, no conv1'
    default: return big ? launch<64, 128, 256, false, false, 256>(k, s) : launch<64, 128, 256, false, false, 128>(k, s);
  }
}
