// fp16x3 3x3 convolution (stride 1, pad 1) for narrow outputs (cout <= 64; the wide form below: <= 128): halo-tile implicit
// GEMM.  The input may be the nearest x2 of the tensor in memory (in_up2: RRDB's upconv1 / upconv2, round 3).
//
// The generic kernels fetch a 128-byte operand row per output pixel, filter tap and 32-channel slice:
// with N = 32 output channels that is 20 KiB of LDS-DMA per 6 MFMAs of a wave, and the RRDB dense
// blocks (rrdb.py / _layers.py:168-200: 276 of the 351 convs have 32 filters, 69 have 64) run at the
// operand-feed limit.  Here a workgroup owns an 8 x 32 patch of output pixels and stages, per 32-channel
// slice, the (8+2) x (32+2) halo patch ONCE; the nine taps read it at shifted row indices.
//
//  * 8 waves, one workgroup per CU: four COMPUTE waves (wave w: image rows 2w, 2w+1 of the patch = two 32x32 MFMA
//    tiles x 32 filters per pass; cout = 64 runs two passes per slice over the same halo patch) and four LOADER
//    waves that issue every LDS-DMA.  An LDS-DMA instruction occupies its wave until the vector-memory path has
//    taken all 64 lanes' requests (~90 cycles each in a burst of 80 per block: cycle probes put 25 % of the
//    kernel there when the compute waves issued them); a loader wave can sit in that queue while the compute
//    wave of the same SIMD keeps the matrix pipe fed.
//  * LDS: two halo stages [344][128 B] + two filter buffers [9 taps][32][128 B] = 158 KiB.  A "block" =
//    (slice, pass) = 9 taps = 108 MFMAs per wave, ONE barrier per block: the compute waves arrive when they have
//    read the block's last fragments (its buffers are dead), the loader waves when the NEXT block's operands
//    have landed (vmcnt(0)); behind it the loaders refill the dead buffers with the block two ahead.
//  * The nine taps are unrolled: a tap is 12 MFMAs with the 12 ds_read_b128 of the NEXT tap's fragments between
//    them (each read follows the last MFMA that takes the register's old contents), with no scalar bookkeeping or
//    branch in between.  (The first version kept tap / slice / stage as run-time state: rocprof showed 26 % matrix-pipe
//    utilisation, ~1000 idle cycles per tap in compare / select / branch chains between the MFMA groups.)
//    Fragment addresses: for a halo row the four 16-byte chunks a lane needs (hi / lo x k-half) differ only in
//    address bits 5-6, so ONE register per (row offset, kw) pair — 12 per lane — and XOR immediates give all.
//  * Persistent over tiles (grid = min(tiles, CUs)): a workgroup's tiles form one stream of blocks, refilled two
//    blocks ahead ACROSS tile boundaries, so a tile starts with its operands already in LDS and its epilogue (fp32
//    tile in the halo stage that has just died, buffer stores that drop out-of-range pixels) runs while the next
//    tile's second filter block is in flight.
//  * Same arithmetic as the other fp16x3 kernels (al*bh + ah*bl + ah*bh per k-half, channel slices outer,
//    taps inner) => bit-identical results.
//  * Fragment reads are inline-asm ds_read_b128: the compiler's waitcnt pass would otherwise drain the pending
//    LDS-DMA (vmcnt(0)) in front of every LDS read (see fcp_conv_f16x3_big.hip).
#include "fcp_conv_common.h"

#include <type_traits>

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 32;            // output patch
constexpr int HW_ = TW + 2;               // halo patch width (34); height TH + 2 = 10
constexpr int HROWS = 344;                // 340 halo rows padded to a multiple of 8
constexpr int ROWB = 128;
constexpr int A_BYTES = HROWS * ROWB;     // 44032: one halo stage
constexpr int B_BYTES = 9 * 32 * ROWB;    // 36864: one filter buffer (9 taps x 32 filters)
constexpr int B_OFF = 2 * A_BYTES;        // filter buffers follow the two halo stages
constexpr int LDS_TOTAL = 2 * A_BYTES + 2 * B_BYTES;   // 161792
constexpr int A_LD = 11;                  // halo DMA instructions per thread: 2752 16-byte pieces / 256 (last: waves 0..2)

template <int IMM>
__device__ __forceinline__ f16x8 lds_read128i(unsigned addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }
// epilogue LDS accesses, also inline asm: the next tile's first DMA is in flight while the fp32 tile is staged
__device__ __forceinline__ void lds_write32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 lds_read128f(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int TN>   // 32-filter passes per slice: cout <= 32 * TN
__global__ void __launch_bounds__(512, 1) conv3x3_halo_f16x3(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  const int tiles_x = (p.out_w + TW - 1) / TW, tiles_y = (p.out_h + TH - 1) / TH;
  const int ntiles = p.n * tiles_x * tiles_y;
  // persistent workgroups; XCD x (= blockIdx & 7) walks a contiguous range of tiles, so neighbouring patches
  // (which share halo rows) meet in one L2
  const int nb = gridDim.x, bid = blockIdx.x;
  const int per_x = (ntiles + 7) / 8;
  const int xcd = bid & 7, slot = bid >> 3, slots = (nb + 7 - xcd) / 8;   // workgroups of this XCD: slot = 0..slots-1

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wave8 >= 4;                     // waves 4..7 issue the LDS-DMA, waves 0..3 compute
  const int wave_u = wave8 & 3;
  const int tid = threadIdx.x & 255;                  // index within the role
  const int nslices = p.ctiles;
  const int nblocks = nslices * TN;

  // ---- the workgroup's tiles form ONE stream of blocks: a block's filter buffer is refilled two blocks ahead and a
  //      slice's halo stage two slices ahead ACROSS tile boundaries, so a new tile starts with its operands in LDS.
  //      Both roles walk the same stream and meet at the same barriers.
  int it = slot;
  if (it >= per_x || xcd * per_x + it >= ntiles) return;

  if (loader) {
    // =================================================================== loader waves
    const int lrow = tid >> 3;                          // 0..31 (+32 i)
    const int csrc = (tid & 7) ^ swz(lrow);             // rows differ by multiples of 32: one swizzle per thread
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    // filter row n = lrow (+ 32 per pass): the 9 taps of one channel slice are contiguous (9 x 128 B)
    const unsigned wbase = (unsigned)((lrow * p.wrow) * 4 + csrc * 16);
    unsigned abase[A_LD];
    int hyx[A_LD];                                       // halo row lrow + 32 i -> (hy << 8 | hx), or -1 past the patch
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int hr = lrow + 32 * i;
      const int hy = hr / HW_, hx = hr - hy * HW_;
      hyx[i] = hr < (TH + 2) * HW_ ? (hy << 8 | hx) : -1;
    }
    auto tile_setup = [&](int tile) {
      const int tx = tile % tiles_x;
      const int ty = (tile / tiles_x) % tiles_y;
      const int ni = tile / (tiles_x * tiles_y);
      const int y0 = ty * TH, x0 = tx * TW;
      // halo row (hy, hx) -> input pixel (y0 - pad_h + hy, x0 - 1 + hx); pad_h = 1, or 0 when the input view holds a
      // real row above the output view (fcp_conv_desc.band_top)
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int y = y0 - p.pad_h + (hyx[i] >> 8), x = x0 - 1 + (hyx[i] & 255);
        const bool ok = hyx[i] >= 0 && (unsigned)y < (unsigned)p.in_h && (unsigned)x < (unsigned)p.in_w;
        // in_up2: the logical input is the nearest x2 of the physical one (RRDB's upconv1 / upconv2): the halo row of logical
        // pixel (y, x) is physical pixel (y / 2, x / 2); p.ph / p.pw are the physical sizes (= in_h / in_w otherwise)
        const int yp = p.in_up2 ? y >> 1 : y, xp = p.in_up2 ? x >> 1 : x;
        abase[i] = ok ? ((unsigned)((ni * p.ph + yp) * p.pw + xp) * (unsigned)p.in_ld + (unsigned)(csrc * 4)) * 4u : 0xFFFFFFFFu;
      }
    };
    auto dma_halo = [&](int cs, int stage) {             // 11 (waves 0..2) / 10 (wave 3) instructions
      char* a = lds + stage * A_BYTES + wave_u * 8 * ROWB;
#pragma unroll
      for (int i = 0; i < A_LD - 1; ++i) {
        const unsigned ro = abase[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(cs * 128)), 0, 0, 0);
      }
      if (wave_u < 3) {   // rows 320..343
        const unsigned ro = abase[A_LD - 1];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * (A_LD - 1) * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(cs * 128)), 0, 0, 0);
      }
    };
    auto dma_filter = [&](int blk, int buf) {            // block = slice * TN + pass; 9 instructions
      const int cs = blk / TN, pass = blk - cs * TN;
      char* b = lds + B_OFF + buf * B_BYTES + wave_u * 8 * ROWB;
      const unsigned wo = wbase + (unsigned)(pass * 32 * p.wrow * 4 + cs * 9 * 128);
#pragma unroll
      for (int t = 0; t < 9; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b + 32 * t * ROWB), 16,
                                                 (int)(wo + (unsigned)(t * 128)), 0, 0, 0);
    };

    tile_setup(xcd * per_x + it);
    dma_halo(0, 0);
    dma_filter(0, 0);
    dma_halo(1, 1);                                       // nslices >= 2 (launcher)
    dma_filter(1, 1);
    int blk = 0;
    unsigned bbuf = 0, astage = 0;                        // filter buffer of the current block, halo stage of the current slice
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // P: blocks 0 and 1 are in LDS
    for (;;) {
      const bool new_slice = TN == 1 || (blk & 1) != 0;              // the next block starts the next channel slice
      const int cs = blk / TN;
      const bool last_of_tile = blk == nblocks - 1;
      const int nit = it + slots, ntile = xcd * per_x + nit;
      const bool more = nit < per_x && ntile < ntiles;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the NEXT block's operands have landed
      __builtin_amdgcn_s_barrier();                                  // X: ... and this block's buffers are dead
      // refill what just died with the operands of the block two ahead in the stream
      bool halo_delayed = false;
      const int b2 = blk + 2;
      if (b2 < nblocks) dma_filter(b2, (int)bbuf);
      else if (more) dma_filter(b2 - nblocks, (int)bbuf);
      if (new_slice) {
        const int s2 = cs + 2;
        if (s2 < nslices) dma_halo(s2, (int)astage);
        else if (more) {
          if (s2 == nslices) {                                       // next tile's first slice: its addresses replace this tile's
            tile_setup(ntile);
            dma_halo(0, (int)astage);
          } else {
            halo_delayed = true;                                     // next tile's second slice: the freed stage hosts the epilogue first
          }
        }
      }
      const unsigned freed = astage;
      bbuf ^= 1u;
      if (new_slice) astage ^= 1u;
      ++blk;
      if (new_slice && last_of_tile) {
        __builtin_amdgcn_s_barrier();                                  // the compute waves have read the finished tile back from `freed`
        if (!more) return;
        if (halo_delayed) dma_halo(1, (int)freed);
        it = nit;
        blk = 0;
      }
    }
  }

  // ===================================================================== compute waves
  const int xl = lane & 31, half = lane >> 5;
  // ---- fragment addressing (per lane, tile independent)
  // halo rows read by this lane: hr = (2 w + d) * 34 + xl + kw, d = i + kh in 0..3, kw in 0..2.  aaddr[d][kw] is
  // the LDS address (stage 0) of chunk (half ^ swz(hr)); the chunks (2 + half), (4 + half), (6 + half) ^ swz(hr)
  // are that address ^ 0x20 / 0x40 / 0x60 (rows are 128-byte aligned, the chunk index lives in bits 4-6).
  unsigned aaddr[4][3];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int hr = (2 * wave_u + d) * HW_ + xl + kw;
      aaddr[d][kw] = lds0 + (unsigned)(hr * ROWB + ((half ^ swz(hr)) << 4));
    }
  // filter fragments: row xl of tap t at B_OFF + buf * B_BYTES + t * 4096 + xl * 128, chunk c ^ swz(xl)
  const unsigned baddr0 = lds0 + (unsigned)(B_OFF + xl * ROWB + ((half ^ swz(xl)) << 4));

  f16x8 fah[2][2], fal[2][2], fbh[2], fbl[2];   // [tile | k-half][k-half]: ONE set, refilled register by register (see tap_step)

  // fragment `idx` (0..11: k-half 0 {bh, al0, al1, bl, ah0, ah1}, then k-half 1) of tap TAP from halo stage offset
  // `aoff` / filter buffer offset `boff`
  auto read_one = [&](auto tap_c, auto idx_c, unsigned aoff, unsigned boff) {
    constexpr int tap = decltype(tap_c)::value, idx = decltype(idx_c)::value;
    constexpr int kh = tap / 3, kw = tap % 3;
    constexpr int s = idx / 6, r = idx % 6;
    if constexpr (r == 0) fbh[s] = lds_read128i<tap * 4096>((baddr0 + boff) ^ (unsigned)(s ? 0x20 : 0));
    else if constexpr (r == 3) fbl[s] = lds_read128i<tap * 4096>((baddr0 + boff) ^ (unsigned)(s ? 0x60 : 0x40));
    else {
      constexpr int i = (r == 1 || r == 4) ? 0 : 1;
      constexpr bool lo = r < 3;
      const unsigned a = (aaddr[i + kh][kw] + aoff) ^ (unsigned)((lo ? 0x40 : 0) | (s ? 0x20 : 0));
      if constexpr (lo) fal[i][s] = lds_read128i<0>(a);
      else fah[i][s] = lds_read128i<0>(a);
    }
  };

  // output through a buffer resource (out-of-range pixels get offset 0xFFFFFFFF: the hardware drops the store, no
  // divergent branch around it)
  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.in2_bytes, 0x00020000);   // in2_bytes: size of `out` (halo launches)

  int e_ni = 0, e_y0 = 0, e_x0 = 0;                     // coordinates of the tile being computed
  auto tile_coords = [&](int tile) {
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    e_ni = tile / (tiles_x * tiles_y);
    e_y0 = ty * TH; e_x0 = tx * TW;
  };
  tile_coords(xcd * per_x + it);

  f32x16 acc[TN][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][i][e] = 0.f;
  };
  zero_acc();

  __builtin_amdgcn_s_barrier();                         // P: blocks 0 and 1 are in LDS
  __builtin_amdgcn_sched_barrier(0);

  // fragments of (block 0, tap 0)
  static_for<0, 12>([&](auto ic) { read_one(std::integral_constant<int, 0>{}, ic, 0u, 0u); });

  // One tap = 12 MFMAs: six on the k-half-0 fragments, six on the k-half-1 fragments.  There is ONE fragment set: the
  // next tap's read of a fragment follows the LAST MFMA that takes the register's old contents (the matrix pipe has its
  // A / B operands long before the LDS data lands), one or two reads per MFMA, so a wave's LDS requests are spread over
  // the group and land while the other k-half's six MFMAs (~190 cycles) run.  lgkmcnt(6): the older group of six
  // reads is complete.
  auto tap_step = [&](auto tap_c, auto pass_c, unsigned aoff_n, unsigned boff_n) {
    constexpr int tap = decltype(tap_c)::value, pass = decltype(pass_c)::value;
    constexpr int ntap = (tap + 1) % 9;
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      auto rd = [&](auto rc) {
        read_one(std::integral_constant<int, ntap>{}, std::integral_constant<int, 6 * s + decltype(rc)::value>{}, aoff_n, boff_n);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mm = [&](int i, const f16x8& a, const f16x8& b) {
        acc[pass][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[pass][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      mm(0, fal[0][s], fbh[s]); rd(std::integral_constant<int, 1>{});
      mm(1, fal[1][s], fbh[s]); rd(std::integral_constant<int, 2>{});
      mm(0, fah[0][s], fbl[s]);
      mm(1, fah[1][s], fbl[s]); rd(std::integral_constant<int, 3>{});
      mm(0, fah[0][s], fbh[s]); rd(std::integral_constant<int, 4>{});
      mm(1, fah[1][s], fbh[s]); rd(std::integral_constant<int, 5>{}); rd(std::integral_constant<int, 0>{});
    });
  };

  // epilogue of the finished tile through a [256 pixels][32 channels] fp32 tile in halo stage `stage` (just freed)
  auto epilogue = [&](int stage) {
    // A wave stages, reads back and stores only ITS OWN 64 pixels (LDS accesses of one wave execute in order), so the
    // passes need no barrier between the waves; the single barrier at the end tells the loaders that the stage may
    // be refilled.
    const unsigned cs0 = lds0 + (unsigned)(stage * A_BYTES + wave_u * 64 * 128);
    const int eq = lane & 3, er = lane >> 2;                       // 8-channel group, row (+ 16 g) within the wave's 64
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      // this thread's 8 channels of the pass: constants requested first, consumed after the staging
      const int ccol = n * 32 + eq * 8;
      const bool cvalid = ccol < p.cout;
      f32x4 b8[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, w8[2] = {b8[0], b8[0]};
      if (cvalid) {                                                  // cout % 8 == 0 (launcher): whole groups of 8
        if (p.bias != nullptr) { b8[0] = *reinterpret_cast<const f32x4*>(p.bias + ccol); b8[1] = *reinterpret_cast<const f32x4*>(p.bias + ccol + 4); }
        w8[0] = *reinterpret_cast<const f32x4*>(p.wscale + ccol); w8[1] = *reinterpret_cast<const f32x4*>(p.wscale + ccol + 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int row = i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
          lds_write32(cs0 + (unsigned)((row * 32 + xl) * 4), acc[n][i][rr]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        f32x4 va[2], vb[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const unsigned ra = cs0 + (unsigned)(((er + 16 * (2 * gp + g)) * 32 + eq * 8) * 4);
          va[g] = lds_read128f(ra);
          vb[g] = lds_read128f(ra + 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int row = wave_u * 64 + er + 16 * (2 * gp + g);     // pixel of the 8 x 32 patch
          const int y = e_y0 + (row >> 5), x = e_x0 + (row & 31);
          const bool ok = cvalid && y < p.out_h && x < p.out_w;
          const long m = ok ? ((long)e_ni * p.out_h + y) * p.out_w + x : 0;
          float v[8] = {va[g][0], va[g][1], va[g][2], va[g][3], vb[g][0], vb[g][1], vb[g][2], vb[g][3]};
          float r1[8], r2[8];
          if (p.res1 != nullptr) load8(p.res1, m, p.res1_ld, cvalid ? ccol : 0, p.res1_fmt, r1);
          if (p.res2 != nullptr) load8(p.res2, m, p.res2_ld, cvalid ? ccol : 0, p.res2_fmt, r2);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = v[e] * w8[e >> 2][e & 3] + b8[e >> 2][e & 3];
            if (p.res1 != nullptr && p.res1_pre) t += r1[e];
            t = t >= 0.f ? t : t * p.act_slope;
            t = t * p.alpha;
            if (p.res1 != nullptr && !p.res1_pre) t += r1[e];
            if (p.res2 != nullptr) t = t * p.alpha2 + r2[e];
            v[e] = t;
          }
          u32x4_t s0, s1;
          unsigned o0, o1;
          if (p.out_fmt == 1) {
            split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, s0, s1);
            o0 = (unsigned)(m * p.out_ld * 4 + split_chan_off(ccol));
            o1 = o0 + 64u;
          } else {
            s0 = __builtin_bit_cast(u32x4_t, f32x4{v[0], v[1], v[2], v[3]});
            s1 = __builtin_bit_cast(u32x4_t, f32x4{v[4], v[5], v[6], v[7]});
            o0 = (unsigned)((m * p.out_ld + ccol) * 4);
            o1 = o0 + 16u;
          }
          __builtin_amdgcn_raw_buffer_store_b128(s0, rs_out, ok ? o0 : 0xFFFFFFFFu, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(s1, rs_out, ok ? o1 : 0xFFFFFFFFu, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();                                   // every wave has read its pixels back: the stage may be refilled
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- main loop over the stream of blocks (slice, pass): nine unrolled taps each
  int blk = 0;                                          // block index inside the current tile
  unsigned bbuf = 0, astage = 0;                        // filter buffer of the current block, halo stage of the current slice
  auto run_block = [&](auto pass_c) -> bool {           // true: the workgroup is done
    constexpr int pass = decltype(pass_c)::value;
    constexpr bool new_slice = pass == TN - 1;                       // the next block starts the next channel slice
    const unsigned aoff = astage * (unsigned)A_BYTES, boff = bbuf * (unsigned)B_BYTES;
    const bool last_of_tile = blk == nblocks - 1;
    const int nit = it + slots, ntile = xcd * per_x + nit;
    const bool more = nit < per_x && ntile < ntiles;
    static_for<0, 8>([&](auto tc) { tap_step(tc, pass_c, aoff, boff); });
    // tap 8's fragments are (about to be) in registers: the block's buffer (and, after the last pass, the slice's
    // stage) is dead for this wave; the loaders arrive when the next block's operands are in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned aoff_next = new_slice ? (astage ^ 1u) * (unsigned)A_BYTES : aoff;
    const unsigned boff_next = (bbuf ^ 1u) * (unsigned)B_BYTES;
    // (after the workgroup's very last block the "next block" reads fetch stale LDS contents that nothing consumes:
    //  cheaper than a branch around every read of every tap 8; the epilogue's lgkmcnt(0) retires them)
    tap_step(std::integral_constant<int, 8>{}, pass_c, aoff_next, boff_next);
    const unsigned freed = astage;
    bbuf ^= 1u;
    if constexpr (new_slice) astage ^= 1u;
    ++blk;
    if constexpr (new_slice) {
      if (last_of_tile) {
        __builtin_amdgcn_sched_barrier(0);
        epilogue((int)freed);
        if (!more) return true;
        it = nit;
        blk = 0;
        tile_coords(ntile);
        zero_acc();
      }
    }
    return false;
  };
  for (;;) {
    if constexpr (TN == 2) {
      if (run_block(std::integral_constant<int, 0>{})) break;
      if (run_block(std::integral_constant<int, 1>{})) break;
    } else {
      if (run_block(std::integral_constant<int, 0>{})) break;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// "Wide" halo-tile kernel: 3x3 / stride 1 / pad 1 with 33 .. 128 filters (TN = 2 or 4 column tiles of 32), cin % 64 == 0.
//
// The kernel above runs one PASS per 32 filters over a staged halo patch and re-reads the pixel fragments from LDS in every
// pass (one 16-byte read per MFMA).  Here a compute wave keeps its pixel fragments (2 row tiles x hi / lo) for a k-half
// and walks the TN column tiles with them: 4 + 2 TN fragment reads per 6 TN MFMAs (0.5 per MFMA at TN = 4, the 256-row
// kernel's ratio), and the operand DMA per MFMA stays the halo kernel's (one 44 KB patch per 32-channel slice for nine taps).
// The filters of a slice (9 x TN x 4 KB = 147 KB at TN = 4) do not fit beside two halo stages, so they stream through a
// ring of four TAP buffers (TN x 4 KB each; stream tap G lives in slot G % 4).  The stream is: tiles chained, slices
// outer, taps inner.  Tap g's step starts with the ONE barrier of the tap and then reads tap g's own filter fragments
// just in time (two in flight) plus the first fragments of tap g + 1; so behind barrier g the slot of tap g - 1 is dead
// and the loader waves refill it with tap g + 3, which has two tap steps (96 MFMAs per wave at TN = 4) to land: a loader
// arrives at barrier g + 1 when everything it issued BEFORE barrier g has landed (s_waitcnt vmcnt(n) with n = what it issued
// behind barrier g).  A slice's halo stage dies with its tap 8 and takes the slice two ahead.  Eighteen taps = two slices =
// one statically unrolled UNIT (cin % 64 == 0): taps, stages and ring slots are immediates (the four slot bases rotate by
// two per unit).  Same arithmetic and K order as every other fp16x3 kernel (per accumulator and k-half: al*bh, ah*bl,
// ah*bh; slices outer, taps inner) => bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------
// RES: the epilogue supports the two residual inputs (registers: only instantiated for TN = 2).  BT: taps per barrier
// (a block), RS: ring slots.  With the taps of a block read just in time, an interval between two barriers touches the
// slots of BT + 1 taps; the loaders, behind the barrier of block b, put the next BT taps of the stream into the slots of
// block b - 1 and always run RS - BT taps ahead of the block being computed: (RS - BT - 1) / BT whole blocks of slack for a
// DMA to land (2 => a loader waits only for what it issued BEFORE the previous barrier; 1 => for everything).
template <int TN, bool RES, int BT, int RS>
__global__ void __launch_bounds__(512, 1) conv3x3_halo_wide_f16x3(const ConvK p) {
  constexpr int TAPB = TN * 32 * ROWB;                // filter bytes of one tap: TN x 32 rows x 128 B
  constexpr int PRE = RS - BT;                        // taps the loaders run ahead
  constexpr bool SLACK2 = (RS - BT - 1) / BT >= 2;
  static_assert(18 % BT == 0 && 18 % RS == 2 && (RS & (RS - 1)) == 0 && PRE >= BT + 1, "block / ring geometry");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  const int tiles_x = (p.out_w + TW - 1) / TW, tiles_y = (p.out_h + TH - 1) / TH;
  const int ntiles = p.n * tiles_x * tiles_y;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int per_x = (ntiles + 7) / 8;
  const int xcd = bid & 7, slot = bid >> 3, slots = (nb + 7 - xcd) / 8;

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wave8 >= 4;
  const int wave_u = wave8 & 3;
  const int tid = threadIdx.x & 255;
  const int nslices = p.ctiles;                       // even (launcher)
  const int T = 9 * nslices;                          // taps per tile

  int it = slot;
  if (it >= per_x || xcd * per_x + it >= ntiles) return;

  if (loader) {
    // =================================================================== loader waves
    const int lrow = tid >> 3;
    const int csrc = (tid & 7) ^ swz(lrow);
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const unsigned wbase = (unsigned)((lrow * p.wrow) * 4 + csrc * 16);
    unsigned abase[A_LD];
    int hyx[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int hr = lrow + 32 * i;
      const int hy = hr / HW_, hx = hr - hy * HW_;
      hyx[i] = hr < (TH + 2) * HW_ ? (hy << 8 | hx) : -1;
    }
    auto tile_setup = [&](int tile) {
      const int tx = tile % tiles_x;
      const int ty = (tile / tiles_x) % tiles_y;
      const int ni = tile / (tiles_x * tiles_y);
      const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int y = y0 - p.pad_h + (hyx[i] >> 8), x = x0 - 1 + (hyx[i] & 255);
        const bool ok = hyx[i] >= 0 && (unsigned)y < (unsigned)p.in_h && (unsigned)x < (unsigned)p.in_w;
        // in_up2: the logical input is the nearest x2 of the physical one (RRDB's upconv1 / upconv2): the halo row of logical
        // pixel (y, x) is physical pixel (y / 2, x / 2); p.ph / p.pw are the physical sizes (= in_h / in_w otherwise)
        const int yp = p.in_up2 ? y >> 1 : y, xp = p.in_up2 ? x >> 1 : x;
        abase[i] = ok ? ((unsigned)((ni * p.ph + yp) * p.pw + xp) * (unsigned)p.in_ld + (unsigned)(csrc * 4)) * 4u : 0xFFFFFFFFu;
      }
    };
    constexpr int HALO_I = A_LD - 1;                      // halo instructions every loader wave issues (waves 0..2: one more)
    auto dma_halo = [&](int cs, int stage) {
      char* a = lds + stage * A_BYTES + wave_u * 8 * ROWB;
#pragma unroll
      for (int i = 0; i < A_LD - 1; ++i) {
        const unsigned ro = abase[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(cs * 128)), 0, 0, 0);
      }
      if (wave_u < 3) {
        const unsigned ro = abase[A_LD - 1];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 32 * (A_LD - 1) * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(cs * 128)), 0, 0, 0);
      }
    };
    // within-tile tap gt (slice gt / 9, tap gt % 9) into ring slot `rs`: TN instructions (32 filter rows each)
    auto dma_tap = [&](int gt, int rs) {
      const int cs = gt / 9, tap = gt - cs * 9;
      char* b = lds + B_OFF + rs * TAPB + wave_u * 8 * ROWB;
      const unsigned wo = wbase + (unsigned)(cs * 9 * 128 + tap * 128);
#pragma unroll
      for (int k = 0; k < TN; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b + 32 * k * ROWB), 16,
                                                 (int)(wo + (unsigned)(k * 32 * p.wrow * 4)), 0, 0, 0);
    };

    tile_setup(xcd * per_x + it);
    dma_halo(0, 0);
    dma_halo(1, 1);
#pragma unroll
    for (int e = 0; e < PRE; ++e) dma_tap(e, e);            // T >= 18 > PRE
    int g = 0;                                            // within-tile index of the first tap of the block whose barrier comes next
    unsigned G = 0;                                       // the same, counted over the whole stream (ring slot = G % RS)
    int pend = 0;                                         // instructions issued behind the previous barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // P: slices 0, 1 and taps 0 .. PRE - 1 are in LDS
    for (;;) {
      const int nit = it + slots, ntile = xcd * per_x + nit;
      const bool more = nit < per_x && ntile < ntiles;
      if constexpr (SLACK2) {
        // everything issued before the previous barrier has landed (what was issued behind it may still fly)
        if (pend >= BT * TN + HALO_I) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BT * TN + HALO_I) : "memory");
        else if (pend >= BT * TN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BT * TN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                                  // barrier of the block at tap g: the slots of the previous block are dead
      pend = 0;
#pragma unroll
      for (int e = 0; e < BT; ++e) {
        const int ta = g + PRE + e;                                    // filters do not depend on the tile: only its existence matters
        if (ta < T) { dma_tap(ta, (int)((G + (unsigned)(PRE + e)) & (unsigned)(RS - 1))); pend += TN; }
        else if (more) { dma_tap(ta - T, (int)((G + (unsigned)(PRE + e)) & (unsigned)(RS - 1))); pend += TN; }
      }
      const int t8 = (g / 9) * 9 - 1;                                // the last "tap 8 of a slice" before tap g
      if (t8 >= 0 && t8 >= g - BT) {                                 // it was in the previous block: its stage takes the slice two ahead
        const int died = t8 / 9, s2 = died + 2, stage = died & 1;
        if (s2 < nslices) { dma_halo(s2, stage); pend += HALO_I; }
        else if (more) {                                             // s2 == nslices: the next tile's first slice; its addresses replace this tile's
          tile_setup(ntile);
          dma_halo(0, stage);
          pend += HALO_I;
        }
      }
      g += BT; G += (unsigned)BT;
      if (g == T) {
        // the tile's last block is running; its last slice lives in stage 1, which hosts the epilogue before it is refilled
        __builtin_amdgcn_s_barrier();                                  // E: the compute waves have read the finished tile back from stage 1
        if (!more) return;
        dma_halo(1, 1);
        pend += HALO_I;
        it = nit;
        g = 0;
      }
    }
  }

  // ===================================================================== compute waves
  const int xl = lane & 31, half = lane >> 5;
  // Fragment addresses of the halo patch (see the kernel above).  Recomputed at the top of every unit from an opaque copy
  // of the lane index (~100 vector instructions per 864 MFMAs): kept across the epilogue they are what the register
  // allocator spills, and a reload in front of a fragment read stalls the wave's MFMA stream.
  unsigned aaddr[4][3];
  auto make_aaddr = [&]() {
    int xo = xl;
    asm volatile("" : "+v"(xo));
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hr = (2 * wave_u + d) * HW_ + xo + kw;
        aaddr[d][kw] = lds0 + (unsigned)(hr * ROWB + ((half ^ swz(hr)) << 4));
      }
  };
  make_aaddr();
  // filter fragments: row xl (+ 32 j) of ring slot k, chunk c ^ swz(xl); the unit's tap t reads slot t % 4 of bslot[],
  // which rotates by two per unit (18 taps)
  unsigned bslot[RS];
#pragma unroll
  for (int k = 0; k < RS; ++k) bslot[k] = lds0 + (unsigned)(B_OFF + k * TAPB + xl * ROWB + ((half ^ swz(xl)) << 4));

  // pixel fragments of a k-half: [k-half set][row tile] hi and lo; filter fragments: two sets, alternating per column tile
  f16x8 fah[2][2], fal[2][2], fbh[2], fbl[2];

  // tap `tap_c` of the halo stage at immediate offset aoff_c: pixel fragment (row tile i, lo / hi) of k-half s into set s
  auto read_a = [&](auto tap_c, auto s_c, auto i_c, auto lo_c, auto aoff_c) {
    constexpr int tap = decltype(tap_c)::value, sk = decltype(s_c)::value, i = decltype(i_c)::value;
    constexpr bool lo = decltype(lo_c)::value != 0;
    constexpr int kh = tap / 3, kw = tap % 3;
    const unsigned a = aaddr[i + kh][kw] ^ (unsigned)((lo ? 0x40 : 0) | (sk ? 0x20 : 0));
    if constexpr (lo) fal[sk][i] = lds_read128i<decltype(aoff_c)::value>(a);
    else fah[sk][i] = lds_read128i<decltype(aoff_c)::value>(a);
  };
  // filter fragment of column tile j, k-half s, from the ring slot at `base` into set `set`
  auto read_b = [&](unsigned base, auto j_c, auto s_c, auto lo_c, auto set_c) {
    constexpr int j = decltype(j_c)::value, sk = decltype(s_c)::value, set = decltype(set_c)::value;
    constexpr bool lo = decltype(lo_c)::value != 0;
    const unsigned b = base ^ (unsigned)((lo ? 0x40 : 0) | (sk ? 0x20 : 0));
    if constexpr (lo) fbl[set] = lds_read128i<j * 4096>(b);
    else fbh[set] = lds_read128i<j * 4096>(b);
  };

  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.in2_bytes, 0x00020000);   // in2_bytes: size of `out` (halo launches)

  f32x16 acc[TN][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][i][e] = 0.f;
  };
  zero_acc();

  __builtin_amdgcn_s_barrier();                         // P
  __builtin_amdgcn_sched_barrier(0);

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // first fragments of a tile's tap 0 (slice 0 = stage 0): pixel fragments of k-half 0 into set 0, filter fragments of
  // (column tile 0, k-half 0) into set 0
  auto first_frags = [&]() {
    read_a(I0{}, I0{}, I0{}, I1{}, I0{}); read_a(I0{}, I0{}, I1{}, I1{}, I0{});
    read_a(I0{}, I0{}, I0{}, I0{}, I0{}); read_a(I0{}, I0{}, I1{}, I0{}, I0{});
    read_b(bslot[0], I0{}, I0{}, I0{}, I0{}); read_b(bslot[0], I0{}, I0{}, I1{}, I0{});
  };
  first_frags();

  // One tap X = the tap's barrier, then 2 k-halves x TN column tiles x 6 MFMAs.  Group (s, j) uses pixel set s and filter
  // set (s * TN + j) & 1 and, between its MFMAs, requests the NEXT group's filter fragments into the other filter set plus
  // its share of the pixel fragments of the next k-half (k-half 1 of X during s = 0; k-half 0 of the next tap Y during
  // s = 1) into the other pixel set: every register is rewritten a whole group after its last use.
  auto tap_step = [&](auto first_c, auto tap_c, auto aoff_x, unsigned bx, auto ntap_c, auto aoff_n, unsigned bn) {
    if constexpr (decltype(first_c)::value != 0) {       // first tap of a block
      __builtin_amdgcn_s_barrier();                      // the block's operands have landed (loaders); the previous block's slots are dead (this wave)
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      static_for<0, TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int gi = s * TN + j, set = gi & 1, nset = set ^ 1;
        // This group's filter fragments were the first two reads of the previous group; the pixel fragments it requested
        // after them (groups 0 and 1 of a k-half carry two each: all four of the next k-half are under way early) may
        // still be in flight, except at a k-half's first group, which needs its pixel set complete.
        if constexpr (j == 1 || (j == 2 && TN == 4)) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        auto mm = [&](int i, const f16x8& a, const f16x8& b) {
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j][i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        auto rd_b = [&](auto lo_c) {                                // next group's filter fragments
          if constexpr (j + 1 < TN) read_b(bx, std::integral_constant<int, j + 1>{}, sc, lo_c, std::integral_constant<int, nset>{});
          else if constexpr (s == 0) read_b(bx, I0{}, I1{}, lo_c, std::integral_constant<int, nset>{});
          else read_b(bn, I0{}, I0{}, lo_c, std::integral_constant<int, nset>{});
          __builtin_amdgcn_sched_barrier(0);
        };
        auto rd_a = [&](auto idx_c) {                               // next k-half's pixel fragments: lo 0, lo 1, hi 0, hi 1
          constexpr int idx = decltype(idx_c)::value;
          if constexpr (idx < 4) {
            using IC = std::integral_constant<int, idx & 1>;
            using LO = std::integral_constant<int, idx < 2 ? 1 : 0>;
            if constexpr (s == 0) read_a(tap_c, I1{}, IC{}, LO{}, aoff_x);
            else read_a(ntap_c, I0{}, IC{}, LO{}, aoff_n);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        mm(0, fal[s][0], fbh[set]); rd_b(I0{});
        mm(1, fal[s][1], fbh[set]); rd_b(I1{});
        mm(0, fah[s][0], fbl[set]); rd_a(std::integral_constant<int, j < 2 ? 2 * j : 4>{});
        mm(1, fah[s][1], fbl[set]); rd_a(std::integral_constant<int, j < 2 ? 2 * j + 1 : 4>{});
        mm(0, fah[s][0], fbh[set]);
        mm(1, fah[s][1], fbh[set]);
      });
    });
  };

  // epilogue of the finished tile through a [256 pixels][32 channels] fp32 tile in halo stage 1 (just freed), one column
  // tile at a time; a wave stages, reads back and stores only ITS OWN 64 pixels
  auto epilogue = [&]() {
    const int tile = xcd * per_x + it;                               // coordinates of the finished tile (scalar, recomputed: registers)
    const int e_tx = tile % tiles_x, e_ty = (tile / tiles_x) % tiles_y;
    const int e_ni = tile / (tiles_x * tiles_y), e_y0 = e_ty * TH, e_x0 = e_tx * TW;
    const unsigned cs0 = lds0 + (unsigned)(A_BYTES + wave_u * 64 * 128);
    const int eq = lane & 3, er = lane >> 2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the last step's requests for the next tile's fragments
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      asm volatile("" ::: "memory");                                  // one column tile's constants at a time (registers)
      __builtin_amdgcn_sched_barrier(0);
      const int ccol = n * 32 + eq * 8;
      const bool cvalid = ccol < p.cout;
      f32x4 b8[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, w8[2] = {b8[0], b8[0]};
      if (cvalid) {
        if (p.bias != nullptr) { b8[0] = *reinterpret_cast<const f32x4*>(p.bias + ccol); b8[1] = *reinterpret_cast<const f32x4*>(p.bias + ccol + 4); }
        w8[0] = *reinterpret_cast<const f32x4*>(p.wscale + ccol); w8[1] = *reinterpret_cast<const f32x4*>(p.wscale + ccol + 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int row = i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
          lds_write32(cs0 + (unsigned)((row * 32 + xl) * 4), acc[n][i][rr]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gp = 0; gp < (RES ? 2 : 4); ++gp) {                  // RES: two rows per round trip; else one (registers)
        constexpr int GN = RES ? 2 : 1;
        f32x4 va[GN], vb[GN];
#pragma unroll
        for (int g = 0; g < GN; ++g) {
          const unsigned ra = cs0 + (unsigned)(((er + 16 * (GN * gp + g)) * 32 + eq * 8) * 4);
          va[g] = lds_read128f(ra);
          vb[g] = lds_read128f(ra + 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GN; ++g) {
          const int row = wave_u * 64 + er + 16 * (GN * gp + g);
          const int y = e_y0 + (row >> 5), x = e_x0 + (row & 31);
          const bool ok = cvalid && y < p.out_h && x < p.out_w;
          const long m = ok ? ((long)e_ni * p.out_h + y) * p.out_w + x : 0;
          float v[8] = {va[g][0], va[g][1], va[g][2], va[g][3], vb[g][0], vb[g][1], vb[g][2], vb[g][3]};
          if constexpr (RES) {
            float r1[8], r2[8];
            if (p.res1 != nullptr) load8(p.res1, m, p.res1_ld, cvalid ? ccol : 0, p.res1_fmt, r1);
            if (p.res2 != nullptr) load8(p.res2, m, p.res2_ld, cvalid ? ccol : 0, p.res2_fmt, r2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float t = v[e] * w8[e >> 2][e & 3] + b8[e >> 2][e & 3];
              if (p.res1 != nullptr && p.res1_pre) t += r1[e];
              t = t >= 0.f ? t : t * p.act_slope;
              t = t * p.alpha;
              if (p.res1 != nullptr && !p.res1_pre) t += r1[e];
              if (p.res2 != nullptr) t = t * p.alpha2 + r2[e];
              v[e] = t;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {                   // the same expressions without the residual terms
              float t = v[e] * w8[e >> 2][e & 3] + b8[e >> 2][e & 3];
              t = t >= 0.f ? t : t * p.act_slope;
              v[e] = t * p.alpha;
            }
          }
          u32x4_t s0, s1;
          unsigned o0, o1;
          if (p.out_fmt == 1) {
            split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, s0, s1);
            o0 = (unsigned)(m * p.out_ld * 4 + split_chan_off(ccol));
            o1 = o0 + 64u;
          } else {
            s0 = __builtin_bit_cast(u32x4_t, f32x4{v[0], v[1], v[2], v[3]});
            s1 = __builtin_bit_cast(u32x4_t, f32x4{v[4], v[5], v[6], v[7]});
            o0 = (unsigned)((m * p.out_ld + ccol) * 4);
            o1 = o0 + 16u;
          }
          __builtin_amdgcn_raw_buffer_store_b128(s0, rs_out, ok ? o0 : 0xFFFFFFFFu, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(s1, rs_out, ok ? o1 : 0xFFFFFFFFu, 0, 0);
        }
      }
      // the next column tile's staging overwrites what this wave has just read back: LDS accesses of a wave execute in order
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();                                   // E: every wave has read its pixels back: stage 1 may be refilled
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- main loop: units of 18 taps (two slices), every tap / stage / ring slot an immediate
  int unit = 0;                                          // unit index inside the current tile
  const int nunits = nslices / 2;
  for (;;) {
    if (unit != 0 || it != slot) make_aaddr();           // (the first unit's are in registers already)
    static_for<0, 18>([&](auto tc) {
      constexpr int t = decltype(tc)::value, tn = (t + 1) % 18;
      using OX = std::integral_constant<int, (t / 9) * A_BYTES>;
      using ON = std::integral_constant<int, (tn / 9) * A_BYTES>;
      // tap t + 1 of the unit lives in slot (t + 1) % RS; the next unit's tap 0 (t = 17) in slot 18 % RS = 2 of the current rotation
      tap_step(std::integral_constant<int, t % BT == 0 ? 1 : 0>{}, std::integral_constant<int, t % 9>{}, OX{}, bslot[t % RS],
               std::integral_constant<int, tn % 9>{}, ON{}, bslot[(t + 1) % RS]);
    });
    {                                                    // 18 taps = whole ring turns + 2: rotate the slot bases by two
      const unsigned b0 = bslot[0], b1 = bslot[1];
      static_for<0, RS - 2>([&](auto kc) { bslot[decltype(kc)::value] = bslot[decltype(kc)::value + 2]; });
      bslot[RS - 2] = b0; bslot[RS - 1] = b1;
    }
    if (++unit == nunits) {
      const int nit = it + slots, ntile = xcd * per_x + nit;
      const bool more = nit < per_x && ntile < ntiles;
      __builtin_amdgcn_sched_barrier(0);
      epilogue();
      if (!more) break;
      it = nit;
      unit = 0;
      zero_acc();
      // The last tap step requested the next tile's first fragments, but holding them across the epilogue costs 24
      // registers the epilogue needs (the allocator spilled the fragment addresses instead): request them again here.
      first_frags();
    }
  }
}

}  // namespace

namespace fcp_conv {

int launch_f16x3_halo(const ConvK& k, hipStream_t s) {
  const size_t lds = (size_t)LDS_TOTAL;
  const long tiles = (long)k.n * ((k.out_h + TH - 1) / TH) * ((k.out_w + TW - 1) / TW);
  FCP_REQUIRE(tiles < (1L << 31), "conv(halo): too many tiles");
  const int cus = fcp_cu_count();
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  if (k.cout <= 32) {
    FCP_LDS_OPT_IN(&conv3x3_halo_f16x3<1>, lds);
    hipLaunchKernelGGL(conv3x3_halo_f16x3<1>, dim3(grid), dim3(512), lds, s, k);
  } else {
    FCP_LDS_OPT_IN(&conv3x3_halo_f16x3<2>, lds);
    hipLaunchKernelGGL(conv3x3_halo_f16x3<2>, dim3(grid), dim3(512), lds, s, k);
  }
  FCP_LAUNCH_OK();
  return 0;
}

#define FCP_WIDE2_BT 2
constexpr int WIDE2_BT = FCP_WIDE2_BT;   // taps per barrier of the 64-filter form (eight 8 KB slots); tools/wide2_ab.sh: RRDB conv5 610 / 586 / 605 us for 1 / 2 / 3

int launch_f16x3_halo_wide(const ConvK& k, hipStream_t s) {
  const int tn = k.cout <= 64 ? 2 : 4;
  const size_t lds = (size_t)(2 * A_BYTES + (tn == 2 ? 8 : 4) * tn * 32 * ROWB);   // TN = 2: eight 8 KB slots; TN = 4: four 16 KB slots
  const long tiles = (long)k.n * ((k.out_h + TH - 1) / TH) * ((k.out_w + TW - 1) / TW);
  FCP_REQUIRE(tiles < (1L << 31), "conv(halo): too many tiles");
  FCP_REQUIRE(k.ctiles >= 2 && k.ctiles % 2 == 0 && k.cout <= 128, "conv(halo, wide): cin must be a multiple of 64, cout <= 128");
  FCP_REQUIRE(tn == 2 || (k.res1 == nullptr && k.res2 == nullptr), "conv(halo, wide): no residual inputs with more than 64 filters");
  const int cus = fcp_cu_count();
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  if (tn == 2) {
    FCP_LDS_OPT_IN((&conv3x3_halo_wide_f16x3<2, true, WIDE2_BT, 8>), lds);
    hipLaunchKernelGGL((conv3x3_halo_wide_f16x3<2, true, WIDE2_BT, 8>), dim3(grid), dim3(512), lds, s, k);
  } else {
    FCP_LDS_OPT_IN((&conv3x3_halo_wide_f16x3<4, false, 1, 4>), lds);
    hipLaunchKernelGGL((conv3x3_halo_wide_f16x3<4, false, 1, 4>), dim3(grid), dim3(512), lds, s, k);
  }
  FCP_LAUNCH_OK();
  return 0;
}

}  // namespace fcp_conv
