// bf16 MFMA GEMM with fused epilogues for the encoder's dense contractions (K5, K6, K9, K11, K12, K13) -- gfx950.
//
//   C[M, N] = sum over K-segments  A_seg[rowmap(m), :] . W_seg[n, :]^T       (both operands K-contiguous)
//
// * operands are bf16 "hi" planes, optionally with "lo" planes (x = hi + lo, |x - hi - lo| <= 2^-17 |x|).
//   TERMS = 1: acc += a_hi b_hi.  TERMS = 3: acc += a_hi b_lo + a_lo b_hi + a_hi b_hi  (split-bf16: the mode
//   that meets the 1e-3 hidden-state parity bound; DESIGN.md "Numerics").  Accumulation is always fp32 MFMA.
// * K-segments let one kernel serve: plain linears (1 segment), linears with a LoRA term
//   ([x | u] . [W | B]^T, 2 segments) and the stride-2 conv stem as an implicit GEMM (3 taps = 3 segments whose
//   source row is 2 s + tap - 1, rows outside the clip reading as zero).
// * two block tiles: 256 x 256 (8 waves as 2 x 4, each 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16; one
//   workgroup per CU, 128 KiB LDS) for the large-N encoder shapes -- at 128 x 128 the L2 -> LDS traffic per MFMA
//   FLOP is twice as high and the kernel sits on the L2 bandwidth instead of the matrix pipe -- and 128 x 128
//   (4 waves as 2 x 2) for N not a multiple of 256.
//   Tiles are staged global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier per
//   K-tile; LDS rows are XOR-swizzled on the SOURCE address (the DMA writes LDS linearly) and on the ds_read_b128,
//   so fragment reads are bank-conflict free.
// * 1-D grid with an XCD-aware remap: consecutive tiles of one A row-panel run on one XCD so the panel is fetched
//   from HBM once and then served by that XCD's L2.
#include <type_traits>
#include "common.h"

namespace {

constexpr int kMaxSeg = 3;

struct GemmArgs {
  int M, N, nseg, tiles_m, tiles_n;
  int m_begin;           // first output row of this launch (a GEMM may be split into a main launch and a finer-tiled tail)
  GemmSeg seg[kMaxSeg];
  GemmOut out;
  const bf16_t* zeros;   // >= 128 zero bytes (padding rows of the conv stem)
};

// block-tile configurations: WM x WN waves, each wave TM x TN tiles of 16 x 16
struct Cfg128 { static constexpr int WM = 2, WN = 2, TM = 4, TN = 4; };
struct Cfg256 { static constexpr int WM = 2, WN = 4, TM = 8, TN = 4; };
struct Cfg128x256 { static constexpr int WM = 2, WN = 4, TM = 4, TN = 4; };   // half-height tiles for the last, partial round

template <int BK> __device__ __forceinline__ int swz(int row);
// BK = 64: 128-byte rows, 8 chunks of 16 B; BK = 32: 64-byte rows, 4 chunks.  See DESIGN.md "LDS images".
template <> __device__ __forceinline__ int swz<64>(int row) { return (row >> 1) & 7; }
template <> __device__ __forceinline__ int swz<32>(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }  // {0,2,3,1}

template <int TERMS, int BK, class CFG>
struct Tile {
  static constexpr int BM = CFG::WM * CFG::TM * 16, BN = CFG::WN * CFG::TN * 16;
  static constexpr int THREADS = CFG::WM * CFG::WN * 64;
  static constexpr int NP = (TERMS == 3) ? 2 : 1;           // planes per operand: hi (+ lo)
  static constexpr int PLANE_A = BM * BK * 2, PLANE_W = BN * BK * 2;   // bytes
  static constexpr int OFF_W = NP * PLANE_A;
  static constexpr int STAGE = NP * (PLANE_A + PLANE_W);
  static constexpr int CPR = BK / 8;                        // 16-byte chunks per row
  static constexpr int ITERS_A = (BM * CPR) / THREADS;      // LDS-DMA instructions per thread per plane
  static constexpr int ITERS_W = (BN * CPR) / THREADS;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Per-thread staging state: the rows a thread copies are the same for every K-tile, so the (row-mapped) source
// pointers are computed once per K-segment and only advanced by BK elements per K-tile.
template <int TERMS, int BK, class CFG>
struct Stager {
  using T = Tile<TERMS, BK, CFG>;
  const bf16_t* a_hi[T::ITERS_A]; const bf16_t* a_lo[T::ITERS_A];
  const bf16_t* w_hi[T::ITERS_W]; const bf16_t* w_lo[T::ITERS_W];
  int si, kk, nk;

  __device__ __forceinline__ void open_segment(const GemmArgs& g, int seg, int m0, int n0, int wave, int lane) {
#ifdef AWT_DIAG_SAME_TILE   // every workgroup streams the same operand tiles: 100 % L2 hits (timing-only)
    m0 = 0; n0 = 0;
#endif
    si = seg; kk = 0;
    const GemmSeg& sg = g.seg[seg];
    nk = sg.K / BK;
#pragma unroll
    for (int it = 0; it < T::ITERS_A; ++it) {
      const int p = it * T::THREADS + wave * 64 + lane;         // linear 16-byte slot in the plane
      const int row = p / T::CPR;
      const int c = (p % T::CPR) ^ swz<BK>(row);                // source chunk that lands in this slot
      int m = m0 + row; m = m < g.M ? m : g.M - 1;              // M tail: clamp (never stored)
      const int grp = m / sg.rows_out, r = m - grp * sg.rows_out;
      const int sr = r * sg.row_mul + sg.row_add;               // conv stem: 2 s + tap - 1
      const bool ok = (sr >= 0) && (sr < sg.rows_in);
      const int64_t aoff = ((int64_t)grp * sg.rows_in + sr) * sg.lda + c * 8;
      a_hi[it] = ok ? sg.a_hi + aoff : nullptr;
      a_lo[it] = (ok && TERMS == 3) ? sg.a_lo + aoff : nullptr;
    }
#pragma unroll
    for (int it = 0; it < T::ITERS_W; ++it) {
      const int p = it * T::THREADS + wave * 64 + lane;
      const int row = p / T::CPR;
      const int c = (p % T::CPR) ^ swz<BK>(row);
      int n = n0 + row; n = n < g.N ? n : g.N - 1;
      const int64_t woff = (int64_t)n * sg.ldw + c * 8;
      w_hi[it] = sg.w_hi + woff;
      w_lo[it] = (TERMS == 3) ? sg.w_lo + woff : nullptr;
    }
  }
  // One LDS-DMA instruction of the current K-tile (piece p of NPIECES).  Issuing the pieces one at a time between
  // MFMA groups matters: a burst of all of them blocks the wave in VMEM issue for about as long as the transfer
  // takes, and since instructions issue in order its MFMAs wait behind them (measured: tile time = DMA time + MFMA
  // time for the burst form).
  static constexpr int NPIECES = T::NP * (T::ITERS_A + T::ITERS_W);
  template <int P>
  __device__ __forceinline__ void stage_piece(const GemmArgs& g, char* stage_base, int wave) {
    static_assert(P >= 0 && P < NPIECES, "piece index");
    constexpr int plane = P % T::NP;                 // 0 = hi, 1 = lo
    constexpr int q = P / T::NP;                     // 0 .. ITERS_A + ITERS_W - 1
    const int koff = kk * BK;
    if constexpr (q < T::ITERS_A) {
      char* dst = stage_base + (q * T::THREADS + wave * 64) * 16 + plane * T::PLANE_A;
      const bf16_t* src = plane == 0 ? a_hi[q] : a_lo[q];
      glds16(src ? (const void*)(src + koff) : (const void*)g.zeros, dst);
    } else {
      constexpr int it = q - T::ITERS_A;
      char* dst = stage_base + T::OFF_W + (it * T::THREADS + wave * 64) * 16 + plane * T::PLANE_W;
      glds16((plane == 0 ? w_hi[it] : w_lo[it]) + koff, dst);
    }
  }
  __device__ __forceinline__ void advance(const GemmArgs& g, int m0, int n0, int wave, int lane) {
    if (++kk == nk && si + 1 < g.nseg) open_segment(g, si + 1, m0, n0, wave, lane);
  }
  // copy the current K-tile into `stage_base`, then step to the next K-tile (opening the next segment if needed)
  __device__ __forceinline__ void stage_and_advance(const GemmArgs& g, char* stage_base, int m0, int n0, int wave, int lane) {
    const int koff = kk * BK;
#pragma unroll
    for (int it = 0; it < T::ITERS_A; ++it) {
      char* dst = stage_base + (it * T::THREADS + wave * 64) * 16;
      glds16(a_hi[it] ? (const void*)(a_hi[it] + koff) : (const void*)g.zeros, dst);
#ifndef AWT_DIAG_SKIP_LO_DMA
      if (TERMS == 3) glds16(a_lo[it] ? (const void*)(a_lo[it] + koff) : (const void*)g.zeros, dst + T::PLANE_A);
#endif
    }
#pragma unroll
    for (int it = 0; it < T::ITERS_W; ++it) {
      char* dst = stage_base + T::OFF_W + (it * T::THREADS + wave * 64) * 16;
      glds16(w_hi[it] + koff, dst);
#ifndef AWT_DIAG_SKIP_LO_DMA
      if (TERMS == 3) glds16(w_lo[it] + koff, dst + T::PLANE_W);
#endif
    }
    if (++kk == nk && si + 1 < g.nseg) open_segment(g, si + 1, m0, n0, wave, lane);
  }
};

// Epilogue on four consecutive columns of one row (the accumulators are transposed through LDS first, so a lane owns
// a 16-byte fp32 / 8-byte bf16 piece of a row and 16 lanes cover 64 contiguous columns).  Side inputs (residual row,
// positional table, saved pre-activation) are loaded by load_side4 one strip AHEAD of their use: in the epilogue every
// wave of the workgroup is past its last MFMA, so a load-then-use per strip would expose its latency eight times.
template <int EPI>
__device__ __forceinline__ float4 load_side4(const GemmOut& o, int m, int n, int M) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m >= M || n >= o.n_valid) return r;
  if (EPI == EPI_F32_RESID) {
    r = *reinterpret_cast<const float4*>(o.resid + (int64_t)m * o.ldo + n);
  } else if (EPI == EPI_F32_GELU_POS) {
    r = *reinterpret_cast<const float4*>(o.pos + (int64_t)(m % o.rows_pos) * o.ldo + n);
  } else if (EPI == EPI_BF16_DGELU) {
    const int64_t off = (int64_t)m * o.ldo + n;
    const uint2 ph = *reinterpret_cast<const uint2*>(o.pre_hi + off);
    uint2 pl = make_uint2(0u, 0u);
    if (o.pre_lo) pl = *reinterpret_cast<const uint2*>(o.pre_lo + off);
    r.x = bf16_to_f32((bf16_t)(ph.x & 0xFFFF)) + bf16_to_f32((bf16_t)(pl.x & 0xFFFF));
    r.y = bf16_to_f32((bf16_t)(ph.x >> 16)) + bf16_to_f32((bf16_t)(pl.x >> 16));
    r.z = bf16_to_f32((bf16_t)(ph.y & 0xFFFF)) + bf16_to_f32((bf16_t)(pl.y & 0xFFFF));
    r.w = bf16_to_f32((bf16_t)(ph.y >> 16)) + bf16_to_f32((bf16_t)(pl.y >> 16));
  }
  return r;
}

template <int EPI>
__device__ __forceinline__ void store_out4(const GemmOut& o, int m, int n, float4 acc, float4 side, int M) {
#ifdef AWT_DIAG_NO_STORE   // timing-only: epilogue arithmetic kept alive, nothing written
  if (acc.x != 123.456f) { asm volatile("" ::"v"(acc.y), "v"(side.x)); return; }
#endif
  if (m >= M || n >= o.n_valid) return;
  float v[4] = {acc.x, acc.y, acc.z, acc.w};
  const float sd[4] = {side.x, side.y, side.z, side.w};
  if (o.bias) {
    const float4 b = *reinterpret_cast<const float4*>(o.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (EPI == EPI_F32 || EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS) {
    float* dst = o.f32 + (int64_t)m * o.ldo + n;
    if (EPI == EPI_F32_RESID) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] += sd[t];
    } else if (EPI == EPI_F32_GELU_POS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = gelu_erf(v[t]) + sd[t];
    }
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    int64_t off;
    if (EPI == EPI_QKV) {
      const int d = o.H * 64;
      const int which = n / d, within = n - which * d;
      const int h = within >> 6, e = within & 63;
      const int b = m / o.S, s = m - b * o.S;
      if (which == 0) { v[0] *= o.scale; v[1] *= o.scale; v[2] *= o.scale; v[3] *= o.scale; }
      off = (int64_t)which * o.plane_stride + (((int64_t)b * o.H + h) * o.S + s) * 64 + e;
    } else {
      off = (int64_t)m * o.ldo + n;
      if (EPI == EPI_BF16_GELU_SAVE) {
        bf16_t ph[4], pl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) split_bf16(v[t], ph[t], pl[t]);
        *reinterpret_cast<uint2*>(o.hi2 + off) = make_uint2(pack2(ph[0], ph[1]), pack2(ph[2], ph[3]));
        if (o.lo2) *reinterpret_cast<uint2*>(o.lo2 + off) = make_uint2(pack2(pl[0], pl[1]), pack2(pl[2], pl[3]));
      }
      if (EPI == EPI_BF16_DGELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float x = sd[t];   // d/dx gelu(x) = Phi(x) + x phi(x)
          v[t] *= 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (EPI == EPI_BF16_GELU || EPI == EPI_BF16_GELU_SAVE) ? gelu_erf(v[t]) : v[t] * o.scale;
      }
    }
    bf16_t hi[4], lo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) split_bf16(v[t], hi[t], lo[t]);
    *reinterpret_cast<uint2*>(o.hi + off) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
    if (o.lo) *reinterpret_cast<uint2*>(o.lo + off) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
  }
}

template <int TERMS, int BK, int EPI, class CFG>
__global__ __launch_bounds__(CFG::WM * CFG::WN * 64, 2) void gemm_kernel(GemmArgs g) {
  using T = Tile<TERMS, BK, CFG>;
  constexpr int TM = CFG::TM, TN = CFG::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // XCD-aware bijective remap: hardware deals block b to XCD b % 8; give each XCD a contiguous run of tiles
  const int nwg = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = g.m_begin + tm * T::BM, n0 = tn * T::BN;

  int ktiles = 0;
  for (int s = 0; s < g.nseg; ++s) ktiles += g.seg[s].K / BK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / CFG::WN, wc = wave - wr * CFG::WN;
  const int frow = lane & 15, fq = lane >> 4;

  Stager<TERMS, BK, CFG> st;
  st.open_segment(g, 0, m0, n0, wave, lane);
  st.stage_and_advance(g, smem, m0, n0, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifndef AWT_GEMM_PIPE
#define AWT_GEMM_PIPE 1
#endif
  if constexpr (TERMS == 3 && AWT_GEMM_PIPE) {
    // ---- software-pipelined K loop (BK = 32: one MFMA k-step per K-tile).
    // A K-tile is TM steps; step i = A row-tile i (hi, lo) x the wave's TN B fragments x 3 products = 12 MFMAs.
    //  * the A fragments of step i + 1 are read from LDS while step i's MFMAs issue (two register sets), so only the
    //    first reads of a tile (B fragments + A_0, right after the barrier) are exposed;
    //  * the LDS-DMA of the NEXT K-tile is issued one piece per step, never as a burst (a burst blocks the wave in VMEM
    //    issue and its MFMAs queue behind it);
    //  * one barrier per K-tile, after the tile's last MFMAs.
    static_assert(BK == 32 && (TM % 2) == 0, "pipelined loop assumes one k-step per tile and an even number of steps");
    using ST = Stager<TERMS, BK, CFG>;
    constexpr int PPS = (ST::NPIECES + TM - 1) / TM;     // DMA pieces per step
    bf16x8 bh[TN], bl[TN], ah[2], al[2];
    const int arow0 = wr * TM * 16 + frow;       // + 16 i
    auto load_b = [&](const char* stage) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wc * TN + j) * 16 + frow;
        const int off = row * 64 + ((fq ^ swz<32>(row)) << 4);
        bh[j] = *reinterpret_cast<const bf16x8*>(stage + T::OFF_W + off);
        bl[j] = *reinterpret_cast<const bf16x8*>(stage + T::OFF_W + T::PLANE_W + off);
      }
    };
    auto load_a = [&](const char* stage, int i, bf16x8& h, bf16x8& l) {
      const int row = arow0 + 16 * i;
      const int off = row * 64 + ((fq ^ swz<32>(row)) << 4);
      h = *reinterpret_cast<const bf16x8*>(stage + off);
      l = *reinterpret_cast<const bf16x8*>(stage + T::PLANE_A + off);
    };
    auto step = [&](auto i_t, const char* cur, char* nxt, bool has_next) {
      constexpr int i = decltype(i_t)::value;
      // a (free) use of this step's A fragments: the compiler's wait for them lands HERE, where they have had a whole
      // step to arrive, instead of behind the next step's reads
      asm volatile("" ::"v"(ah[i & 1]), "v"(al[i & 1]));
#ifndef AWT_DIAG_NO_LDSREAD   // AWT_DIAG_*: timing-only builds of tools/build_variants.sh (wrong results), never shipped
      if constexpr (i + 1 < TM) load_a(cur, i + 1, ah[(i + 1) & 1], al[(i + 1) & 1]);
#endif
#ifdef AWT_DIAG_NO_DMA
      if (false) {
#else
      if (has_next) {
#endif
        if constexpr (i * PPS < ST::NPIECES) st.template stage_piece<(i * PPS < ST::NPIECES ? i * PPS : 0)>(g, nxt, wave);
        if constexpr (PPS > 1 && i * PPS + 1 < ST::NPIECES) st.template stage_piece<(i * PPS + 1 < ST::NPIECES ? i * PPS + 1 : 0)>(g, nxt, wave);
      }
      __builtin_amdgcn_sched_barrier(0);   // reads / DMA of this step are issued before its MFMAs
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < ktiles; ++kt) {
      const char* cur = smem + (kt & 1) * T::STAGE;
      char* nxt = smem + ((kt + 1) & 1) * T::STAGE;
      const bool has_next = kt + 1 < ktiles;
#ifndef AWT_DIAG_NO_LDSREAD
      load_b(cur);
      load_a(cur, 0, ah[0], al[0]);
#else
      if (kt == 0) { load_b(cur); load_a(cur, 0, ah[0], al[0]); ah[1] = ah[0]; al[1] = al[0]; }
#endif
      static_assert(TM == 4 || TM == 8, "steps are unrolled by hand");
      step(std::integral_constant<int, 0>{}, cur, nxt, has_next);
      step(std::integral_constant<int, 1>{}, cur, nxt, has_next);
      step(std::integral_constant<int, 2>{}, cur, nxt, has_next);
      step(std::integral_constant<int, 3>{}, cur, nxt, has_next);
      if constexpr (TM == 8) {
        step(std::integral_constant<int, 4>{}, cur, nxt, has_next);
        step(std::integral_constant<int, 5>{}, cur, nxt, has_next);
        step(std::integral_constant<int, 6>{}, cur, nxt, has_next);
        step(std::integral_constant<int, 7>{}, cur, nxt, has_next);
      }
      if (has_next) st.advance(g, m0, n0, wave, lane);
#ifndef AWT_DIAG_NO_BARRIER
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#endif
    }
  } else {
    // ---- stepped K loop: a K-tile is (BK / 32) x TM steps; step (ks, i) reads A row-tile i and issues TN (x3) MFMAs.
    // The LDS-DMA of the NEXT K-tile is issued one piece per step instead of as a burst at the top of the tile.
    using ST = Stager<TERMS, BK, CFG>;
    constexpr int KS = BK / 32, STEPS = KS * TM;
    constexpr int PPS = (ST::NPIECES + STEPS - 1) / STEPS;     // pieces per step
    for (int kt = 0; kt < ktiles; ++kt) {
      char* cur = smem + (kt & 1) * T::STAGE;
      char* nxt = smem + ((kt + 1) & 1) * T::STAGE;
      const bool has_next = kt + 1 < ktiles;
      const char* a_hi = cur;
      const char* a_lo = cur + T::PLANE_A;
      const char* w_hi = cur + T::OFF_W;
      const char* w_lo = cur + T::OFF_W + T::PLANE_W;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 bh[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = (wc * TN + j) * 16 + frow;
          const int off = row * (BK * 2) + (((ks * 4 + fq) ^ swz<BK>(row)) << 4);
          bh[j] = *reinterpret_cast<const bf16x8*>(w_hi + off);
          if (TERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(w_lo + off);
        }
        auto step = [&](auto i_t) {
          constexpr int i = decltype(i_t)::value;
#ifdef AWT_DIAG_NO_COMPUTE
          return;
#endif
          bf16x8 ah, al;
          const int row = (wr * TM + i) * 16 + frow;
          const int off = row * (BK * 2) + (((ks * 4 + fq) ^ swz<BK>(row)) << 4);
          ah = *reinterpret_cast<const bf16x8*>(a_hi + off);
          if (TERMS == 3) al = *reinterpret_cast<const bf16x8*>(a_lo + off);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#ifndef AWT_DIAG_ONE_MFMA
            if (TERMS == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            }
#else
            asm volatile("" ::"v"(al), "v"(bl[j]));
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
          }
        };
        // steps with their share of the next tile's DMA pieces (compile-time piece indices)
        auto pieces = [&](auto s_t) {
          constexpr int sidx = decltype(s_t)::value;
#ifdef AWT_DIAG_NO_DMA
          if (false) {
#else
          if (has_next) {
#endif
            if constexpr (sidx * PPS + 0 < ST::NPIECES) st.template stage_piece<sidx * PPS + 0>(g, nxt, wave);
            if constexpr (PPS > 1 && sidx * PPS + 1 < ST::NPIECES) st.template stage_piece<(sidx * PPS + 1 < ST::NPIECES ? sidx * PPS + 1 : 0)>(g, nxt, wave);
          }
        };
        auto run = [&](auto ks_t) {
          constexpr int k = decltype(ks_t)::value;
          static_assert(TM == 4 || TM == 8, "steps are unrolled by hand");
          pieces(std::integral_constant<int, k * TM + 0>{}); step(std::integral_constant<int, 0>{});
          pieces(std::integral_constant<int, k * TM + 1>{}); step(std::integral_constant<int, 1>{});
          pieces(std::integral_constant<int, k * TM + 2>{}); step(std::integral_constant<int, 2>{});
          pieces(std::integral_constant<int, k * TM + 3>{}); step(std::integral_constant<int, 3>{});
          if constexpr (TM == 8) {
            pieces(std::integral_constant<int, k * TM + 4>{}); step(std::integral_constant<int, 4>{});
            pieces(std::integral_constant<int, k * TM + 5>{}); step(std::integral_constant<int, 5>{});
            pieces(std::integral_constant<int, k * TM + 6>{}); step(std::integral_constant<int, 6>{});
            pieces(std::integral_constant<int, k * TM + 7>{}); step(std::integral_constant<int, 7>{});
          }
        };
        if (ks == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, (KS > 1 ? 1 : 0)>{});
      }
      if (has_next) st.advance(g, m0, n0, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // epilogue: the C/D layout of the 16x16 MFMA (col = lane & 15, row = (lane >> 4) * 4 + reg) would give 2-4 byte
  // scattered stores; each wave instead transposes one 16-row x 64-column strip at a time through a private LDS
  // patch (the staging buffers are dead: the loop's last barrier has passed) and writes whole 256-byte rows.
  static_assert(TN == 4, "epilogue strips are 64 columns wide");
  constexpr int PITCH = 68;  // floats; 16 x 68 x 4 B = 4352 B per wave
  float* patch = reinterpret_cast<float*>(smem) + wave * (16 * PITCH);
  const int em0 = m0 + wr * TM * 16, en = n0 + wc * 64 + frow * 4;
  float4 side[4], side_next[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) side[it] = load_side4<EPI>(g.out, em0 + fq + 4 * it, en, g.M);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) patch[(fq * 4 + rr) * PITCH + j * 16 + frow] = acc[i][j][rr];
    if (i + 1 < TM) {
#pragma unroll
      for (int it = 0; it < 4; ++it) side_next[it] = load_side4<EPI>(g.out, em0 + (i + 1) * 16 + fq + 4 * it, en, g.M);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = fq + 4 * it;
      const float4 v = *reinterpret_cast<const float4*>(patch + rl * PITCH + frow * 4);
      store_out4<EPI>(g.out, em0 + i * 16 + rl, en, v, side[it], g.M);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) side[it] = side_next[it];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int TERMS, int BK, int EPI, class CFG>
int launch_one(GemmArgs a, hipStream_t s) {
  using T = Tile<TERMS, BK, CFG>;
  static bool attr = false;
  if (!attr) {
    AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<TERMS, BK, EPI, CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T::STAGE));
    attr = true;
  }
  a.tiles_m = (a.M - a.m_begin + T::BM - 1) / T::BM;
  a.tiles_n = (a.N + T::BN - 1) / T::BN;
  hipLaunchKernelGGL((gemm_kernel<TERMS, BK, EPI, CFG>), dim3(a.tiles_m * a.tiles_n), dim3(T::THREADS), 2 * T::STAGE, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int g_force_tile = 0;  // 0 = auto, 128 / 256 = forced (tuning and tests)

// 256 x 256 tiles on 256 CUs: a launch is ceil(tiles / 256) rounds long, and the encoder's shapes leave the last round
// 18 - 58 % full (N = 768: 1125 tiles = 4.39 rounds -> 5).  When it pays, the rows of the partial round are computed
// by a second launch with 128 x 256 tiles, which spreads them over twice as many CUs for half as long.
constexpr int kCUs = 256;
template <int EPI>
int launch_epi(GemmArgs a, int terms, hipStream_t s) {
  const bool big = g_force_tile ? g_force_tile == 256 : (a.N % 256 == 0 && a.M >= 2048);
  a.m_begin = 0;
  if (!big) return terms == 3 ? launch_one<3, 32, EPI, Cfg128>(a, s) : launch_one<1, 64, EPI, Cfg128>(a, s);
  const int tn = a.N / 256, tm = (a.M + 255) / 256, tiles = tm * tn;
  const int full_rounds = tiles / kCUs;
  const int tm_main = full_rounds * kCUs / tn;                 // row panels whose tiles fill whole rounds
  const int rows_left = a.M - tm_main * 256;
  const int tail_tiles = ((rows_left + 127) / 128) * tn;
  // rounds (in units of a 256 x 256 tile's duration): one launch vs main + half-height tail
  const double one = (double)((tiles + kCUs - 1) / kCUs);
  const double split = (double)((tm_main * tn + kCUs - 1) / kCUs) + 0.5 * (double)((tail_tiles + kCUs - 1) / kCUs) + 0.05;
  if (full_rounds >= 1 && rows_left > 0 && split < one) {
    GemmArgs m = a; m.M = tm_main * 256;
    int rc = terms == 3 ? launch_one<3, 32, EPI, Cfg256>(m, s) : launch_one<1, 64, EPI, Cfg256>(m, s);
    if (rc) return rc;
    GemmArgs t = a; t.m_begin = tm_main * 256;
    return terms == 3 ? launch_one<3, 32, EPI, Cfg128x256>(t, s) : launch_one<1, 64, EPI, Cfg128x256>(t, s);
  }
  return terms == 3 ? launch_one<3, 32, EPI, Cfg256>(a, s) : launch_one<1, 64, EPI, Cfg256>(a, s);
}

bf16_t* g_zeros = nullptr;

}  // namespace

void awt_gemm_force_tile(int t) { g_force_tile = t; }

int launch_gemm(awt_ctx* c, int M, int N, const GemmSeg* segs, int nseg, int terms, GemmEpilogue epi, const GemmOut& out,
                hipStream_t s) {
  AWT_REQUIRE(M > 0 && N > 0 && N % 128 == 0, AWT_ERR_INVALID, "gemm: N must be a positive multiple of 128");
  AWT_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, AWT_ERR_INVALID, "gemm: 1..3 K-segments");
  AWT_REQUIRE(terms == 1 || terms == 3, AWT_ERR_INVALID, "gemm: terms must be 1 or 3");
  if (!g_zeros) {
    AWT_HIP_CHECK(hipMalloc((void**)&g_zeros, 256));
    AWT_HIP_CHECK(hipMemset(g_zeros, 0, 256));
  }
  GemmArgs a{};
  a.M = M; a.N = N; a.nseg = nseg; a.out = out; a.zeros = g_zeros;
  double ksum = 0;
  for (int i = 0; i < nseg; ++i) {
    a.seg[i] = segs[i];
    AWT_REQUIRE(segs[i].K > 0 && segs[i].K % 64 == 0, AWT_ERR_INVALID, "gemm: every K-segment must be a positive multiple of 64");
    AWT_REQUIRE(segs[i].a_hi && segs[i].w_hi && (terms == 1 || (segs[i].a_lo && segs[i].w_lo)), AWT_ERR_INVALID, "gemm: null operand plane");
    AWT_REQUIRE(segs[i].lda % 8 == 0 && segs[i].ldw % 8 == 0, AWT_ERR_INVALID, "gemm: leading dimensions must be multiples of 8 elements");
    AWT_REQUIRE(segs[i].rows_out > 0 && segs[i].rows_in > 0, AWT_ERR_INVALID, "gemm: bad row map");
    ksum += segs[i].K;
  }
  ProfScope prof(c, AWT_PROF_GEMM, s, 2.0 * (double)M * (double)out.n_valid * ksum);
  switch (epi) {
    case EPI_F32: return launch_epi<EPI_F32>(a, terms, s);
    case EPI_F32_RESID: return launch_epi<EPI_F32_RESID>(a, terms, s);
    case EPI_BF16: return launch_epi<EPI_BF16>(a, terms, s);
    case EPI_BF16_GELU: return launch_epi<EPI_BF16_GELU>(a, terms, s);
    case EPI_QKV: return launch_epi<EPI_QKV>(a, terms, s);
    case EPI_F32_GELU_POS: return launch_epi<EPI_F32_GELU_POS>(a, terms, s);
    case EPI_BF16_GELU_SAVE: return launch_epi<EPI_BF16_GELU_SAVE>(a, terms, s);
    case EPI_BF16_DGELU: return launch_epi<EPI_BF16_DGELU>(a, terms, s);
  }
  return awt_fail(AWT_ERR_INVALID, "gemm: unknown epilogue");
}
