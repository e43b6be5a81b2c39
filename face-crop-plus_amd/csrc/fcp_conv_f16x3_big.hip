// fp16x3 convolution, 256-row tiles, 8 waves, software-pipelined at half-slice granularity.
//
// Same operands and arithmetic as conv_igemm_f16x3_dma (split32 activations and offline-split
// filters moved global -> LDS by `buffer_load ... lds`; al*bh + ah*bl + ah*bh on
// v_mfma_f32_32x32x16_f16, identical accumulation order => identical results), but organised so
// that the matrix pipe never waits for a fragment read or a DMA round trip:
//
//  * tile 256 x BN (BN = 256: 2x4 waves of 128x64; BN = 128 | 192: 4x2 waves of 64x64 | 64x96), one workgroup per
//    CU, two waves per SIMD.  A K slice costs 48 (24) MFMAs per wave = 3072 (1536) matrix cycles per
//    SIMD against one DMA of 64 (48) KiB: half (three quarters of) the operand bytes per FLOP of the
//    128x128 kernel, and a whole iteration for the DMA to land.
//  * two LDS stages, ONE barrier per slice.  Per iteration kt:
//        MFMAs on F0 (slice kt, k-half 0) with the reads F1 <- (slice kt, k-half 1) between them
//        wait lgkmcnt(0), vmcnt(0); s_barrier           -- slice kt+1 visible, slice kt's stage dead
//        MFMAs on F1 with the reads F0 <- (slice kt+1, k-half 0) between them; DMA slice kt+2 -> dead stage,
//        issued BEFORE these MFMAs by waves 4..7 and AFTER them by waves 0..3 (the two waves of a SIMD never sit
//        in the vector-memory queue at the same time)
//    The fragment reads are inline-asm ds_read_b128: the compiler's waitcnt pass would otherwise
//    drain the pending LDS-DMA (vmcnt(0)) in front of every LDS read.
#include "fcp_conv_common.h"

#include <type_traits>

using namespace fcp_conv;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BMB = 256;          // rows per workgroup tile
constexpr int NT = 512;           // threads per workgroup
constexpr int ROWB = 128;         // bytes per LDS row (32 hi + 32 lo binary16)
constexpr int STAGE = 1 << 16;    // stage stride (power of two: the stage toggles by XOR)

template <int IMM>
__device__ __forceinline__ f16x8 lds_read128(unsigned addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int BN>
__global__ void __launch_bounds__(NT, 1) conv_igemm_f16x3_big(const ConvK p) {
  constexpr int WAVES_N = BN == 256 ? 4 : 2;
  constexpr int WAVES_M = 8 / WAVES_N;
  constexpr int WTM = BMB / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_LD = BMB / 64, B_LD = BN / 64;   // DMA instructions per thread and slice

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);

  // Workgroup -> tile.  The grid is consumed in dispatch rounds of `round_size` workgroups (one per CU the launch may
  // count on); inside a round every XCD (workgroup id mod 8) owns a contiguous run of tiles.  Tile order = M-tile major,
  // and the M-tiles are `mfull` tiles of 256 rows followed by tiles of `tail_rows` rows: the last, partial round is
  // made of shorter tiles that together fill all CUs instead of full-height tiles on a fraction of them.  Splitting M
  // never changes a result: every output pixel still sees the same K loop.
  // Persistent: the launch has one workgroup per CU it may count on (`round_size`); workgroup b works through the virtual
  // block ids b, b + round_size, ... of the round-by-round schedule above (same tile -> XCD mapping as one workgroup per
  // tile gave: round_size is a multiple of 8).  Between two tiles the next tile's first two K slices are requested BEFORE
  // the finished tile's epilogue, so a tile no longer starts with an exposed DMA round trip (8-21 k cycles per tile).
  const int nb = p.big_tiles;
  int vb = blockIdx.x;
  int tile_n = 0, m_start = 0, m_end = 0;
  auto decode = [&](int bid) {
    const int rbase = (bid / p.round_size) * p.round_size;
    const int nr = nb - rbase < p.round_size ? nb - rbase : p.round_size;
    const int bi = bid - rbase;
    const int q8 = nr >> 3, r8 = nr & 7, xcd = bi & 7;
    const int logical = rbase + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bi >> 3);
    tile_n = logical % p.grid_n;
    const int tile_m = logical / p.grid_n;
    m_start = tile_m < p.mfull ? tile_m * BMB : p.mfull * BMB + (tile_m - p.mfull) * p.tail_rows;
    const int m_rows = tile_m < p.mfull ? BMB : p.tail_rows;
    m_end = m_start + m_rows < p.M ? m_start + m_rows : p.M;
  };
  decode(vb);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u / WAVES_N, wn = wave_u % WAVES_N;
  const int lrow = tid >> 3;                                     // 0..63 (+64 i)
  const int csrc = (tid & 7) ^ (((lrow >> 1) & 7) ^ ((lrow & 1) << 2));

  const int hw = p.out_h * p.out_w;
  TapPiece tp[A_LD];
  unsigned base2[A_LD];   // second source of a 1x1 conv (channels >= csplit), or unused
  unsigned woff[B_LD];
  auto setup_tile = [&]() {                                      // operand addresses of the tile (tile_n, m_start, m_end)
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = m_start + lrow + 64 * i;
    unsigned pbase = 0;
    int hi0 = -(1 << 28), wi0 = 0;
    base2[i] = 0xFFFFFFFFu;
    if (m < m_end) {
      const int ni = m / hw;
      const int rem = m - ni * hw;
      const int ho = rem / p.out_w;
      const int wo = rem - ho * p.out_w;
      pbase = (unsigned)(ni * p.ph * p.pw);
      hi0 = ho * p.stride - p.pad_h;
      wi0 = wo * p.stride - p.pad;
      if (p.in2 != nullptr)
        base2[i] = (((unsigned)(ni * p.ph2 + ho * p.stride2) * (unsigned)p.pw2 + (unsigned)(wo * p.stride2)) * (unsigned)p.in2_ld +
                    (unsigned)(csrc * 4)) * 4u;
    }
    tp[i] = make_tap_piece<false>(p, pbase, hi0, wi0, (unsigned)(csrc * 4));
  }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) woff[i] = (unsigned)(((tile_n * BN + lrow + 64 * i) * p.wrow + csrc * 4) * 4);
  };
  setup_tile();
  __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2 ? p.in2 : p.in), 0, p.in2_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  unsigned rowoff[A_LD];
  auto set_tap = [&](int tap, int kh_i, int kw_i) {
    const unsigned tapoff = (unsigned)((kh_i * p.pw + kw_i) * p.in_ld) * 4u;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) rowoff[i] = ((tp[i].mask >> tap) & 1u) ? tp[i].base + tapoff : 0xFFFFFFFFu;
  };
  int tap = 0, kh_i = 0, kw_i = 0, c0 = 0;
  auto advance = [&]() {
    ++tap;
    if (++kw_i >= p.kw) {
      kw_i = 0;
      if (++kh_i >= p.kh) { kh_i = 0; tap = 0; c0 += BK; }
    }
    set_tap(tap, kh_i, kw_i);
  };
  // one slice: A_LD + B_LD DMA instructions per wave, each 64 lanes x 16 B = rows 8*wave + 64*i .. +7
  auto dma_slice = [&](int kt, int stage) {
    char* a = lds + stage * STAGE + wave_u * 8 * ROWB;
    char* b = a + BMB * ROWB;
    if (c0 >= p.csplit) {                  // wave-uniform: this slice comes from the second source
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const unsigned ro = base2[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in2, (__attribute__((address_space(3))) void*)(a + 64 * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)((c0 - p.csplit) * 4)), 0, 0, FCP_AUX_A);
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const unsigned ro = rowoff[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void*)(a + 64 * i * ROWB), 16,
                                                 (int)(ro == 0xFFFFFFFFu ? 0xFFFFFFFFu : ro + (unsigned)(c0 * 4)), 0, 0, FCP_AUX_A);
      }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b + 64 * i * ROWB), 16,
                                               (int)(woff[i] + (unsigned)(kt * BK * 4)), 0, 0, FCP_AUX_B);
  };

  f32x16 acc[TM][TN];

  // fragment addresses (stage 0); the stage toggles by XOR with STAGE
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int rsw = (((lane & 31) >> 1) & 7) ^ ((lane & 1) << 2);
  const int half = lane >> 5;
  const unsigned arow = lds0 + (unsigned)((wm * WTM + (lane & 31)) * ROWB);
  const unsigned brow = lds0 + (unsigned)((BMB + wn * WTN + (lane & 31)) * ROWB);
  unsigned aH[2], aL[2], bH[2], bL[2];   // [k-half]
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const unsigned oh = (unsigned)(((2 * s + half) ^ rsw) << 4), ol = (unsigned)(((4 + 2 * s + half) ^ rsw) << 4);
    aH[s] = arow + oh; aL[s] = arow + ol;
    bH[s] = brow + oh; bL[s] = brow + ol;
  }

  // a tile's first two slices in flight (stages 0 and 1 are free: called at kernel start and after a tile's last barrier)
  auto prefetch_tile = [&]() {
    tap = 0; kh_i = 0; kw_i = 0; c0 = 0;
    set_tap(0, 0, 0);
    dma_slice(0, 0);
    if (p.ktiles > 1) {
      advance();
      dma_slice(1, 1);
    }
  };
  prefetch_tile();
  // Rows of a tile the wave owns, in MFMA sub-tiles of 32: TM for a full tile, fewer (down to 0) in a tail tile.  The
  // main loop is instantiated per count (wave-uniform dispatch): a wave with fewer sub-tiles issues fewer MFMAs and
  // fragment reads but the same DMA share and the same barriers, so the two waves of a SIMD (w, w + 4: with the 256-column
  // tile they are the two row halves) split the matrix pipe of that SIMD in proportion to the rows that exist.
  int tm_act = 0;
  auto main_loop = [&](auto tma_c) {
    constexpr int TMA = decltype(tma_c)::value;
    f16x8 fah[2][TMA > 0 ? TMA : 1], fal[2][TMA > 0 ? TMA : 1], fbh[2][TN], fbl[2][TN];   // [fragment set]
  auto read_frags = [&](auto set_c, unsigned stage_xor) {   // set s holds k-half s
    constexpr int set = decltype(set_c)::value;
    static_for<0, TMA>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      fah[set][i] = lds_read128<i * 32 * ROWB>(aH[set] ^ stage_xor);
      fal[set][i] = lds_read128<i * 32 * ROWB>(aL[set] ^ stage_xor);
    });
    static_for<0, TN>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      fbh[set][j] = lds_read128<j * 32 * ROWB>(bH[set] ^ stage_xor);
      fbl[set][j] = lds_read128<j * 32 * ROWB>(bL[set] ^ stage_xor);
    });
  };
  constexpr std::integral_constant<int, 0> SET0{};
  constexpr std::integral_constant<int, 1> SET1{};
  // MFMAs on fragment set CS with the reads of set LS (from the stage at `stage_xor`) between them, one read per
  // TMA * TN * 3 / (2 TMA + 2 TN) MFMAs: the LDS requests of the eight lockstep waves arrive spread over the phase
  // instead of as one burst of 8 x (2 TMA + 2 TN) in front of it.  (Reads past the last slice fetch stale LDS
  // contents nobody consumes: cheaper than a branch.)
  auto mfmas_reads = [&](auto cs_c, auto ls_c, unsigned stage_xor) {
    constexpr int cs = decltype(cs_c)::value, ls = decltype(ls_c)::value;
    constexpr int NM = 3 * TMA * TN, NR = 2 * TMA + 2 * TN;
    static_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int g = m / (TMA * TN), i = (m % (TMA * TN)) / TN, j = m % TN;
      // the filter fragment is the ROW operand: the tile is accumulated transposed (filters x pixels; same products, same
      // K order, same bits), so that a lane ends up with four consecutive filters of ONE pixel per accumulator quad —
      // what the LDS-free epilogue below needs
      if constexpr (g == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[cs][j], fal[cs][i], acc[i][j], 0, 0, 0);
      else if constexpr (g == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbl[cs][j], fah[cs][i], acc[i][j], 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[cs][j], fah[cs][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // read r is issued after MFMA floor((r + 1) * NM / NR) - 1
      static_for<0, NR>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (((r + 1) * NM) / NR - 1 == m
        ) {
          if constexpr (r < 2 * TMA) {
            constexpr int ii = r / 2;
            if constexpr (r % 2 == 0) fah[ls][ii] = lds_read128<ii * 32 * ROWB>(aH[ls] ^ stage_xor);
            else fal[ls][ii] = lds_read128<ii * 32 * ROWB>(aL[ls] ^ stage_xor);
          } else {
            constexpr int jj = (r - 2 * TMA) / 2;
            if constexpr (r % 2 == 0) fbh[ls][jj] = lds_read128<jj * 32 * ROWB>(bH[ls] ^ stage_xor);
            else fbl[ls][jj] = lds_read128<jj * 32 * ROWB>(bL[ls] ^ stage_xor);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
  };

  read_frags(SET0, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  unsigned sx = 0u;                       // XOR of the stage holding slice kt
  for (int kt = 0; kt < p.ktiles; ++kt) {
    // ---- k-half 0 of slice kt on the matrix pipe, k-half 1 on its way to registers
    mfmas_reads(SET0, SET1, sx);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's part of slice kt+1 has landed
    __builtin_amdgcn_s_barrier();                          // slice kt+1 visible; nobody reads slice kt's stage again
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-half 1 of slice kt (k-half 0 of slice kt+1 on its way to registers), and the DMA of slice kt+2 into the
    //      stage that has just died.  An LDS-DMA instruction holds its wave until the vector-memory path has taken the
    //      64 requests: with all eight waves issuing their A_LD + B_LD instructions at once (64 KiB per slice through a
    //      64 B/clk path) cycle probes showed BOTH waves of a SIMD stuck there for 800-1700 cycles per slice with the
    //      matrix pipe idle.  The two waves of a SIMD (w, w+4) therefore take the two jobs in opposite order: one
    //      feeds the matrix pipe while the other sits in the queue.  (Round 3, negative: the same instructions issued
    //      one at a time BETWEEN the MFMAs, spread over the whole window in which the stage is free, are 8-12 % slower —
    //      every one of them stalls its wave's MFMA stream; profiles/r03_probes.md.)
    const bool dma_first = wave_u >= 4;
    if (dma_first && kt + 2 < p.ktiles) {
      advance();
      dma_slice(kt + 2, kt & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas_reads(SET1, SET0, sx ^ (unsigned)STAGE);
    __builtin_amdgcn_sched_barrier(0);
    if (!dma_first && kt + 2 < p.ktiles) {
      advance();
      dma_slice(kt + 2, kt & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    sx ^= (unsigned)STAGE;
  }
  };
  for (;;) {
  // slices 0 and 1 of this tile have landed (and, from the second tile on, the previous tile's stores have left)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  tm_act = (m_end - m_start - wm * WTM + 31) >> 5;
  tm_act = __builtin_amdgcn_readfirstlane(tm_act < 0 ? 0 : (tm_act > TM ? TM : tm_act));
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  static_for<0, TM + 1>([&](auto tc) {
    if (tm_act == decltype(tc)::value) main_loop(tc);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // the finished tile's coordinates; then the next tile's operand addresses and its first two slices, before the epilogue
  const int e_tile_n = tile_n, e_m_start = m_start, e_m_end = m_end, e_tm_act = tm_act;
  const bool more = p.big_persist && vb + p.round_size < nb;
  if (more) {
    vb += p.round_size;
    decode(vb);
    setup_tile();
    prefetch_tile();
  }
  // ---- epilogue straight from the (transposed) accumulators: conv_epilogue_regs (fcp_conv_common.h) — no LDS, no barrier
  conv_epilogue_regs<TM, TN, WTM, WTN>(p, acc, e_m_start, e_m_end, e_tm_act, e_tile_n * BN,
                                       (e_tile_n + 1) * BN < p.cout ? (e_tile_n + 1) * BN : p.cout, wm, wn, lane, hw);
  if (!more) break;
  }   // tiles of this workgroup
}

// M-tile schedule.  Uniform: ceil(M / 256) tiles of 256 rows.  Balanced (descriptor flag FCP_CONV_BALANCE_TAIL): the
// tiles of all FULL dispatch rounds (`cus` workgroups each, one per CU) stay 256 rows high; the rows that are left are cut
// into tiles of R rows (a multiple of 32, <= 256) so that the last round has a tile for (nearly) every CU:
//   e.g. M = 102400, one N-tile, 256 CUs: 400 uniform tiles = 2 rounds (the second on 144 CUs);
//        balanced: 256 tiles of 256 rows + 231 tiles of 160 rows = 1 + ~0.7 rounds.
// A launch with less than one round of uniform tiles is all "tail".  Results do not depend on the schedule.
template <int BN>
int launch(ConvK k, hipStream_t s) {
  const size_t lds = 2 * (size_t)STAGE;   // two stages (the epilogue works from the accumulators: no LDS tile)
  FCP_LDS_OPT_IN((&conv_igemm_f16x3_big<BN>), lds);
  k.grid_n = fcp_cdiv(k.cout, BN);
  int cus = k.cu_budget > 0 ? k.cu_budget : fcp_cu_count();
  cus = cus < 8 ? 8 : (cus & ~7);                                    // rounds are XCD-interleaved: a multiple of 8
  k.round_size = cus;
  const int mt = fcp_cdiv(k.M, BMB);
  k.mfull = mt;
  k.tail_rows = BMB;
  k.grid_m = mt;
  if (k.balance) {
    const long tiles = (long)mt * k.grid_n;
    const long full_rounds = tiles / cus;
    const int mfull = (int)((full_rounds * cus) / k.grid_n);         // whole M-tiles inside the full rounds
    const long rem = (long)k.M - (long)mfull * BMB;
    if (rem > 0) {
      const long slots = cus / k.grid_n > 0 ? cus / k.grid_n : 1;     // M-tiles one round has room for
      long r = ((rem + slots - 1) / slots + 31) / 32 * 32;
      r = r < 32 ? 32 : r;
      if (r < BMB) {
        k.mfull = mfull;
        k.tail_rows = (int)r;
        k.grid_m = mfull + (int)((rem + r - 1) / r);
      }
    }
  }
  k.big_tiles = k.grid_m * k.grid_n;
  // Persistent only when the launch owns the device: with two detector streams (cu_budget = half the CUs each) a launch of
  // cu_budget persistent workgroups cannot spread over the CUs the other stream leaves idle, and that costs more than the
  // hidden tile prologues bring (same box, headline: 3514-3518 faces/s one workgroup per tile, 3256-3269 persistent; single
  // stream 3318-3331 vs 3344-3349).  FCP_BIG_PERSIST=0 / 1 forces it off / on.
  static const int persist_env = getenv("FCP_BIG_PERSIST") ? atoi(getenv("FCP_BIG_PERSIST")) : -1;
  k.big_persist = persist_env < 0 ? (k.cu_budget == 0) : (persist_env != 0);
  hipLaunchKernelGGL((conv_igemm_f16x3_big<BN>), dim3(k.big_persist && k.big_tiles > k.round_size ? k.round_size : k.big_tiles), dim3(NT), lds, s, k);
  FCP_LAUNCH_OK();
  return 0;
}

}  // namespace

namespace fcp_conv {

int launch_f16x3_big(const ConvK& k, int tile_n, hipStream_t s) {
  return tile_n == 256 ? launch<256>(k, s) : tile_n == 192 ? launch<192>(k, s) : launch<128>(k, s);
}

}  // namespace fcp_conv
