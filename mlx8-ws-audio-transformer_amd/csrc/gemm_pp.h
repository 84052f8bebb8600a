// Persistent 256 x 256 "ping-pong" GEMM K-loop for gfx950: 8 waves (2 x 4, each 128 x 64 of the tile), BOTH operands through LDS by
// 16-byte LDS-DMA, counted vmcnt, raw s_barrier.  Runs the MLP pair (fc1, fc2) of large inference launches by default; the other single-segment
// GEMMs on request (the K loop saturates the matrix pipe, the time between K loops is each tile's stores leaving the CU).
//
//   C[M, N] = A[M, K] . W[N, K]^T          (K5, K6, K9, K11, K12 of SURVEY.md section 2.1: HF:modeling_whisper.py:284-356, 375-376, 566-567)
//
// Shape of the pipeline (cdna_hip_programming.md section 5, "256^2 8-phase template", rebuilt for this operand format):
// * A K-tile is 128 BYTES of every operand row -- the shipped FMT_F16F8S: alternately the fp16 line ("X") and the e4m3 line ("Y") of 64 consecutive k (below, at
//   the enum); FMT_F16: 64 fp16; FMT_F16F8 (harness only): 32 elements as fp16 | e4m3 | e4m3 = 64 + 32 + 32 bytes -- so an operand half-tile ("region", 128 rows)
//   is 16 KB and a K-tile of both operands 64 KB; two K-tile buffers = 128 KB of LDS.
//   Region image: 128-byte rows, the 16-byte chunk c of row r stored at chunk position c ^ ((r >> 1) & 7): every ds_read_b128 of a 16- or 32-row MFMA
//   fragment (16 distinct rows, one chunk column per 16-lane group) is bank-conflict-free, and one LDS-DMA wave-instruction
//   (1 KB) is 8 rows x 128 bytes = 8 whole cache lines of the source (swizzle on the SOURCE address, linear LDS destination).
// * The wave's 128 x 64 output is four quadrants (64 rows x 32 columns); a PHASE = {fragment reads for one quadrant | one region of
//   LDS-DMA (2 pieces per wave) | counted vmcnt | barrier | the quadrant's MFMAs (256 matrix-pipe cycles) | barrier}.  Waves 4-7 run
//   one barrier interval behind waves 0-3 (they execute one extra s_barrier at tile start, waves 0-3 one at tile end), so on every
//   SIMD one wave is in its MFMA cluster while its partner reads fragments and issues DMA: the matrix pipe never waits for LDS.
// * Regions are the unit of staging.  With the tile rows / columns interleaved so that a region holds exactly what ONE phase reads
//   (A0 / A1 = the first / second 64 rows of both wave rows, B0 / B1 = the first / second 32 columns of all four wave columns):
//       reads    B0(T) in phase 4 of K-tile T-1, A0(T) in phase 1, B1(T) in phase 2, A1(T) in phase 3
//       staging  phase 1: A1(T+1)   phase 2: B0(T+2)   phase 3: A0(T+2)   phase 4: B1(T+2)      (each region two phases after its last read)
//   so every region is requested SIX phases before it is read, one region per phase, and `s_waitcnt vmcnt(10)` (five regions x two
//   pieces still in flight) in every phase is exactly "the region the next phase reads has landed".  Reads happen one phase after the
//   wait that retires them (RAW: wait -> barrier -> read), DMA into a region two phases after its last read (WAR).
// * The K-tile stream is CONTINUOUS across the output tiles of a persistent workgroup: the next tile's first K-tiles are already in
//   flight during the epilogue.  At a tile boundary the two wave groups re-align (so that all eight waves run the epilogue together),
//   the phase-4 staging of the last K-tile is deferred to the next tile's start (it would land in the LDS the epilogue uses), and the
//   epilogue transposes through the last 64 KB of LDS (buffer 1's B1 | A1 regions, both read out by then, + the 32 KB beyond the ring).
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

namespace pp {

constexpr int BM = 256, BN = 256, NT = 512;
constexpr int REGION = 16384, BUF = 65536;
constexpr int OFF_A0 = 0, OFF_B0 = 16384, OFF_B1 = 32768, OFF_A1 = 49152;     // region order inside a K-tile buffer
constexpr int PATCH_BASE = BUF + OFF_B1;                                      // 98304: the epilogue's 8 x 8 KB transposition patches
constexpr int LDS_BYTES = PATCH_BASE + 8 * 8192;                              // 163840 = all of a CU's LDS
// FMT_F16F8S: the f16f8 product on 16 x 16 MFMAs.  The operand rows are SPLIT lines: per 64 consecutive k one 128-byte line of fp16 ("X") followed by one
// 128-byte line of e4m3 ("Y": activations hi8 x 64 | lo8 x 64, weights lo8 x 64 | hi8 x 64), so the K-tile stream alternates X (even K-tiles, always buffer 0:
// two v_mfma_f32_16x16x32_f16 per 16 x 16 tile) and Y (odd K-tiles, always buffer 1: one v_mfma_scale_f32_16x16x128_f8f6f4 per tile, lane block kb = lane >> 4
// multiplying A chunks 2 kb, 2 kb + 1 with the same W chunks -- hi8 x lo8 for kb < 2, lo8 x hi8 above, ONE uniform E8M0 scale per operand because
// 2^-Act 2^(-Wgt - 11) is the scale of both).  Every phase is 256 matrix-pipe cycles in both kinds of K-tile; everything else (regions, staging order, waits)
// is the FMT_F16F8 pipeline unchanged.  Under the package power limit the 16 x 16 shapes hold a higher clock (profiles/r04_mfma_shape_mix.txt).
enum { FMT_F16 = 0, FMT_F16_16 = 1 /* bring-up harness only: the single product on v_mfma_f32_16x16x32_f16 */, FMT_F16F8 = 2, FMT_F16F8S = 3 };
template <int FMT> struct Acc { f32x16 t[4][2]; };                 // 32 x 32 tiles: rows wr * 128 + i * 32, columns wc * 64 + j * 32
template <> struct Acc<FMT_F16_16> { f32x4 t[8][4]; };          // 16 x 16 tiles: rows wr * 128 + i * 16, columns wc * 64 + j * 16
template <> struct Acc<FMT_F16F8S> { f32x4 t[8][4]; };

__host__ __device__ constexpr bool tiles16(int fmt) { return fmt == FMT_F16_16 || fmt == FMT_F16F8S; }
__host__ __device__ constexpr int ktile_elems(int fmt) { return (fmt == FMT_F16F8 || fmt == FMT_F16F8S) ? 32 : 64; }   // K per K-tile (FMT_F16F8S: 64 per X + Y pair)
__host__ __device__ constexpr int elem_bytes(int fmt) { return (fmt == FMT_F16F8 || fmt == FMT_F16F8S) ? 4 : 2; }

// Packed weight image: for column tile bn (256 columns), K-tile kt and half s, one 16 KB region in exactly the LDS image order
// (row rho = (n % 256 / 64) * 32 + n % 32 of half s = n % 64 / 32; chunk c at position c ^ ((rho >> 1) & 7)); FMT_F16F8 chunks:
// 0..3 fp16, 4..5 lo8, 6..7 hi8 (the activation lines carry hi8 before lo8: lane half h of the block-scaled MFMA multiplies
// A chunk 4 + 2 h + q with W chunk 4 + 2 h + q, i.e. hi8 x lo8 for h = 0 and lo8 x hi8 for h = 1).
__host__ __device__ __forceinline__ int64_t w_region_offset(int bn, int kt, int s, int nk) { return (((int64_t)bn * nk + kt) * 2 + s) * REGION; }
__host__ __device__ __forceinline__ int w_row_offset(int n_in_tile, int chunk) {   // byte offset of chunk `chunk` of column n (inside its region)
  const int rho = (n_in_tile >> 6) * 32 + (n_in_tile & 31);
  return rho * 128 + ((chunk ^ ((rho >> 1) & 7)) << 4);
}

struct Args {
  const char* A;          // FMT_F16: fp16 [M][K] (row stride a_row_bytes); FMT_F16F8S: split lines [M][K / 64][X 128 B | Y 128 B]; FMT_F16F8: [M][K / 32][fp16 x 32 | hi8 x 32 | lo8 x 32]
  int64_t a_row_bytes;
  const char* W;          // packed regions (above), N padded to a multiple of 256
  int M, N, K, nk;        // nk = K-tiles per output tile (even)
  int tiles_m, tiles_n, ntiles, gm;
  // start-up stagger: workgroup group g = (blockIdx.x / 8) % 4 starts g * nk * stagger * 128 cycles late (stagger = 4: a quarter of a tile's K loop per
  // group), so that the workgroups of an XCD reach their epilogues -- the output bursts -- at different times instead of all at once; 0 = off
  int stagger;
};

template <int OFF>
__device__ __forceinline__ bf16x8 dsr(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// 16-byte LDS-DMA the compiler does not see (cdna_hip_programming.md 5.7): per-lane 64-bit source / wave-uniform base + per-lane offset
__device__ __forceinline__ void dma_v(const char* src, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void dma_s(const char* sbase, unsigned voff, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst_uniform) : "memory");
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Fragments of one quadrant operand.  FMT_F16F8: A = 64 rows (2 row tiles) x {2 fp16 k-steps, 32 e4m3 bytes as two halves}.
// FMT_F16: A = 2 row tiles x 4 fp16 k-steps.  B = 32 columns of the same.
template <int FMT> struct FragA { bf16x8 v[2][4]; };     // [row tile][FMT_F16F8: h0 h1 f0 f1 | FMT_F16: k-step]
template <int FMT> struct FragB { bf16x8 v[4]; };

template <int FMT> __device__ __forceinline__ void tie(FragA<FMT>& a) {
  asm volatile("" : "+v"(a.v[0][0]), "+v"(a.v[0][1]), "+v"(a.v[0][2]), "+v"(a.v[0][3]), "+v"(a.v[1][0]), "+v"(a.v[1][1]), "+v"(a.v[1][2]), "+v"(a.v[1][3]));
}
template <int FMT> __device__ __forceinline__ void tie(FragB<FMT>& b) { asm volatile("" : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3])); }
__device__ __forceinline__ void lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ i32x8 cat8(bf16x8 lo, bf16x8 hi) {
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  return __builtin_shufflevector(__builtin_bit_cast(i32x4_t, lo), __builtin_bit_cast(i32x4_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// one Y-line product (FMT_F16F8S): a template on the accumulator type so that the 32 x 32 formats' instantiations of the K loop never see the call
template <class ACC>
__device__ __forceinline__ void mfma_y(ACC& c, bf16x8 a0, bf16x8 a1, bf16x8 b0, bf16x8 b1) {
  if constexpr (sizeof(ACC) == 16) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(a0, a1), cat8(b0, b1), c, 0, 0, 0, e8m0(-kF8Act), 0, e8m0(-kF8Wgt - kF8Lo));
}

// The K loop of one persistent workgroup.  EPI is a callable  epi(tm, tn, acc)  invoked by all eight waves at the end of every output
// tile with the wave's accumulators Acc<FMT> (32 x 32 tiles t[i][j]: rows wr * 128 + i * 32, columns wc * 64 + j * 32 of the block tile);
// it may use LDS [PATCH_BASE + wave * 8192, + 8192) and must leave every other LDS byte alone; its vector-memory operations may stay
// in flight (they are older than every DMA the loop waits for afterwards: vmcnt retires in order, so the loop's counted waits then
// wait for them as well -- the price of a store burst is paid at the next tile's first waits, not before its first MFMAs).
// Requirements: nk even and >= 4; A readable for tiles_m * 256 rows (rows >= M feed accumulators that are never stored).
template <int FMT, int DMA_WAVES = 8, class EPI>
__device__ __forceinline__ void kloop(const Args& g, char* smem, EPI&& epi) {
  static_assert(DMA_WAVES == 8 || DMA_WAVES == 2, "staging by all eight waves or by waves 6 and 7");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r32 = lane & 31, half = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int nk = g.nk;
  const int nmine = (g.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (nmine <= 0) return;
  if (g.stagger > 0) {
    const int n = (int)((blockIdx.x >> 3) & 3) * nk * g.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(2);
  }

  // tile order: XCD-contiguous runs, groups of GM row panels (as gemm.hip)
  auto tile_coords = [&](int it, int& tm, int& tn) {
    const int orig = blockIdx.x + it * gridDim.x;
    const int xcd = orig & 7, idx = orig >> 3;
    const int qn = g.ntiles >> 3, rn = g.ntiles & 7;
    const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    const int GM = g.gm;
    const int grp = tile / (GM * g.tiles_n);
    const int gm = min(GM, g.tiles_m - grp * GM);
    const int within = tile - grp * GM * g.tiles_n;
    tn = within / gm; tm = grp * GM + (within - tn * gm);
  };

  // ---- operand streams.  Every region type walks the same sequence of (tile, K-tile) pairs at its own phase; inside a tile a stream
  // advances by one K-tile per staging (A: 128 bytes along the row, W: 2 regions), and all four streams move on to the next tile within
  // the tile's last K-tile pair, at fixed points of the schedule (pair<true> below), to bases computed once per tile.
  // DMA_WAVES = 8: every wave stages two 1 KB pieces of each region.  DMA_WAVES = 2: waves 6 and 7 stage eight pieces each and are the only waves that ever wait
  // on vmcnt in the loop -- the other six waves' epilogue stores then drain behind the next tile's K loop instead of in front of its first DMA waits (vmcnt
  // retires in order, so a wave that stores AND stages has to see its stores acknowledged before the DMA it is waiting for).
  constexpr int PIECES = 16 / DMA_WAVES;                  // 1 KB pieces of a 16 KB region per staging wave
  const bool dma_wave = wave >= 8 - DMA_WAVES;
  const int dw = wave - (8 - DMA_WAVES);                 // index among the staging waves
  const char* ab[2];                                     // wave-uniform: A + (tile row + s * 64) * row bytes + K-tile * 128
  const char* wb[2];                                     // wave-uniform: W region (tile column, K-tile, s)
  // per lane: piece p = dw * PIECES + i covers region rows rho = 8 p + (lane >> 3) = source rows (rho >> 6) * 128 + (rho & 63), chunk (lane & 7) ^ ((rho >> 1) & 7).
  // Consecutive pieces are 8 rows apart (a wave-uniform address step; PIECES <= 8 keeps a wave inside one 64-row half) and flip bit 2 of the swizzle: two offsets.
  unsigned aoff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int rho = (dw * PIECES + e) * 8 + (lane >> 3), c = (lane & 7) ^ ((rho >> 1) & 7);
    aoff[e] = (unsigned)(((rho >> 6) * 128 + (rho & 63) - 8 * e) * g.a_row_bytes + c * 16);     // piece i: + 8 i rows on the scalar base
  }
  const unsigned wvoff = lane * 16;
  auto a_base = [&](int tm, int s) { return g.A + ((int64_t)tm * BM + s * 64) * g.a_row_bytes; };
  auto w_base = [&](int tn, int s) { return g.W + w_region_offset(tn, 0, s, nk); };
  auto stage_a = [&](auto s_t, auto buf_t) {
    constexpr int s = decltype(s_t)::value, buf = decltype(buf_t)::value;
    if (dma_wave) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * BUF + (s ? OFF_A1 : OFF_A0) + dw * PIECES * 1024);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) dma_s(ab[s] + (int64_t)(8 * i) * g.a_row_bytes, aoff[i & 1], dst + i * 1024);
    }
    ab[s] += 128;
  };
  auto stage_b = [&](auto s_t, auto buf_t) {
    constexpr int s = decltype(s_t)::value, buf = decltype(buf_t)::value;
    if (dma_wave) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * BUF + (s ? OFF_B1 : OFF_B0) + dw * PIECES * 1024);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) dma_s(wb[s] + (dw * PIECES + i) * 1024, wvoff, dst + i * 1024);
    }
    wb[s] += 2 * REGION;
  };
  // counted waits of the staging waves: N regions still in flight = N * PIECES pieces (the other waves have nothing of the loop's to wait for)
  auto vmw = [&](auto n_t) { if (dma_wave) vmwait<decltype(n_t)::value * PIECES>(); };
  using R5 = std::integral_constant<int, 5>; using R4 = std::integral_constant<int, 4>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // ---- fragment read addresses of the CURRENT K-tile buffer (flipped once per K-tile; a ds_read's 16-bit immediate holds the region
  // and the row tile): row (wave part + r32) * 128 + ((chunk ^ f) << 4), f = (r32 >> 1) & 7
  const int f = (r32 >> 1) & 7;
  unsigned aaddr[4], baddr[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    // FMT_F16F8: x = 0, 1: fp16 k-step x (chunk 2 x + half); x = 2, 3: e4m3 half q = x - 2 (chunk 4 + 2 half + q).  FMT_F16: k-step x (chunk 2 x + half)
    if constexpr (FMT == FMT_F16_16) {     // 16 x 16 x 32 fragments: row lane & 15 (+ 16 per row tile: immediate), chunk 4 ks + (lane >> 4); x = k-step (x < 2)
      const int r16 = lane & 15, f16 = (r16 >> 1) & 7, chunk = 4 * (x & 1) + (lane >> 4);
      aaddr[x] = lds0 + (wr * 64 + r16) * 128 + ((chunk ^ f16) << 4);
      baddr[x] = lds0 + (wc * 32 + r16) * 128 + ((chunk ^ f16) << 4);
    } else if constexpr (FMT == FMT_F16F8S) {   // x = 0, 1: X line (buffer 0), fp16 k-step x: chunk 4 x + kq;  x = 2, 3: Y line (buffer 1), half q = x - 2 of block kq: chunk 2 kq + q
      const int r16 = lane & 15, f16 = (r16 >> 1) & 7, kq = lane >> 4, chunk = x < 2 ? 4 * x + kq : 2 * kq + (x - 2);
      aaddr[x] = lds0 + (x < 2 ? 0 : BUF) + (wr * 64 + r16) * 128 + ((chunk ^ f16) << 4);
      baddr[x] = lds0 + (x < 2 ? 0 : BUF) + (wc * 32 + r16) * 128 + ((chunk ^ f16) << 4);
    } else {
      const int chunk = (FMT == FMT_F16F8 && x >= 2) ? 4 + 2 * half + (x - 2) : 2 * x + half;
      aaddr[x] = lds0 + (wr * 64 + r32) * 128 + ((chunk ^ f) << 4);
      baddr[x] = lds0 + (wc * 32 + r32) * 128 + ((chunk ^ f) << 4);
    }
  }
  auto flip = [&]() {
    if constexpr (FMT != FMT_F16F8S) {      // FMT_F16F8S: the X addresses always point into buffer 0, the Y addresses into buffer 1
#pragma unroll
      for (int x = 0; x < 4; ++x) { aaddr[x] ^= BUF; baddr[x] ^= BUF; }
    }
  };
  // par_t: parity of the K-tile being read (FMT_F16F8S only: 0 = X line, 1 = Y line)
  auto read_a = [&](auto s_t, FragA<FMT>& a, auto par_t) {
    constexpr int base = decltype(s_t)::value ? OFF_A1 : OFF_A0, par = decltype(par_t)::value;
    static_for<4>([&](auto x_t) {
      constexpr int x = decltype(x_t)::value;
      if constexpr (tiles16(FMT)) {   // a.v[i][x]: row tile 2 i + (x >> 1) of 16 rows, k-step (X) / half (Y) x & 1
        constexpr int ai = (FMT == FMT_F16F8S ? 2 * par : 0) + (x & 1);
        a.v[0][x] = dsr<base + (x >> 1) * 2048>(aaddr[ai]); a.v[1][x] = dsr<base + 4096 + (x >> 1) * 2048>(aaddr[ai]);
      } else { a.v[0][x] = dsr<base>(aaddr[x]); a.v[1][x] = dsr<base + 4096>(aaddr[x]); }
    });
  };
  auto read_b = [&](auto s_t, FragB<FMT>& b, auto par_t) {
    constexpr int base = decltype(s_t)::value ? OFF_B1 : OFF_B0, par = decltype(par_t)::value;
    static_for<4>([&](auto x_t) {
      constexpr int x = decltype(x_t)::value;
      if constexpr (tiles16(FMT)) b.v[x] = dsr<base + (x >> 1) * 2048>(baddr[(FMT == FMT_F16F8S ? 2 * par : 0) + (x & 1)]);   // column tile x >> 1 of 16, k-step / half x & 1
      else b.v[x] = dsr<base>(baddr[x]);
    });
  };

  // e4m3 scales of the concatenated cross-term product: lane half 0 multiplies A hi8 (2^-kF8Act) with W lo8 (2^(-kF8Wgt - 11)),
  // lane half 1 A lo8 (2^(-kF8Act - 11)) with W hi8 (2^-kF8Wgt): one E8M0 byte per lane and operand
  const int sc_a = half ? e8m0(-kF8Act - kF8Lo) : e8m0(-kF8Act);
  const int sc_b = half ? e8m0(-kF8Wgt) : e8m0(-kF8Wgt - kF8Lo);

  Acc<FMT> accs;
  auto& acc = accs.t;
  auto zero_acc = [&]() {
    if constexpr (tiles16(FMT)) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){};
    }
  };
  auto mma = [&](auto sa_t, auto sb_t, const FragA<FMT>& a, const FragB<FMT>& b, auto par_t) {
    constexpr int sa = decltype(sa_t)::value, sb = decltype(sb_t)::value, par = decltype(par_t)::value;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FMT == FMT_F16F8S && par == 1) {     // Y line: both cross terms of 64 k in one block-scaled MFMA per 16 x 16 tile
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          mfma_y(acc[4 * sa + rt][2 * sb + ct], a.v[rt >> 1][(rt & 1) * 2], a.v[rt >> 1][(rt & 1) * 2 + 1], b.v[ct * 2], b.v[ct * 2 + 1]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) asm volatile("" : "+v"(acc[4 * sa + rt][2 * sb]), "+v"(acc[4 * sa + rt][2 * sb + 1]));
    } else if constexpr (tiles16(FMT)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            acc[4 * sa + rt][2 * sb + ct] = mfma16<true>(a.v[rt >> 1][(rt & 1) * 2 + ks], b.v[ct * 2 + ks], acc[4 * sa + rt][2 * sb + ct]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) asm volatile("" : "+v"(acc[4 * sa + rt][2 * sb]), "+v"(acc[4 * sa + rt][2 * sb + 1]));
    } else {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      f32x16 c = acc[2 * sa + rt][sb];
      if constexpr (FMT == FMT_F16F8) {
        c = mfma32<true>(a.v[rt][0], b.v[0], c);
        c = mfma32<true>(a.v[rt][1], b.v[1], c);
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(a.v[rt][2], a.v[rt][3]), cat8(b.v[2], b.v[3]), c, 0, 0, 0, sc_a, 0, sc_b);
      } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) c = mfma32<true>(a.v[rt][x], b.v[x], c);
      }
      acc[2 * sa + rt][sb] = c;
    }
    // the accumulators are opaque here: without it the compiler sinks MFMAs (pure operations) out of their phase, down to the next use of
    // the accumulator a K-tile later, across the barriers, keeping their fragments alive meanwhile (spills)
    asm volatile("" : "+v"(acc[2 * sa][sb]), "+v"(acc[2 * sa + 1][sb]));
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); };

  int tm, tn, tm1, tn1;                                   // this tile, the next tile of this workgroup (the last tile again at the stream's end)
  tile_coords(0, tm, tn);
  tile_coords(nmine > 1 ? 1 : 0, tm1, tn1);
  ab[0] = a_base(tm, 0); ab[1] = a_base(tm, 1); wb[0] = w_base(tn, 0); wb[1] = w_base(tn, 1);

  // ---- prologue: B0, A0, B1, A1 of K-tile 0 and B0, A0 of K-tile 1
  stage_b(I0{}, I0{}); stage_a(I0{}, I0{}); stage_b(I1{}, I0{}); stage_a(I1{}, I0{});
  stage_b(I0{}, I1{}); stage_a(I0{}, I1{});
  vmw(R4{});            // B0(0), A0(0) have landed (this wave's pieces): four of the six regions requested may still be in flight
  bar();

  FragA<FMT> ra; FragB<FMT> rba, rbb;
  // one pair of K-tiles.  LAST: the tile's last pair -- the streams move on to the next tile (B0, A0, B1 before their staging in the even
  // K-tile, A1 before its staging in the odd one), the odd K-tile's phase 4 neither reads ahead nor stages (deferred to the next pre-phase)
  auto pair = [&](auto last_t) {
    constexpr bool LAST = decltype(last_t)::value;
    // ================= even K-tile (buffer 0): B0 is in rba
    read_a(I0{}, ra, I0{}); stage_a(I1{}, I1{}); vmw(R5{}); bar();
    lgkm0(); tie(ra); tie(rba); mma(I0{}, I0{}, ra, rba, I0{}); bar();
    if constexpr (LAST) wb[0] = w_base(tn1, 0);
    read_b(I1{}, rbb, I0{}); stage_b(I0{}, I0{}); vmw(R5{}); bar();
    lgkm0(); tie(rbb); mma(I0{}, I1{}, ra, rbb, I0{}); bar();
    if constexpr (LAST) ab[0] = a_base(tm1, 0);
    read_a(I1{}, ra, I0{}); stage_a(I0{}, I0{}); vmw(R5{}); bar();
    lgkm0(); tie(ra); mma(I1{}, I1{}, ra, rbb, I0{}); bar();
    if constexpr (LAST) wb[1] = w_base(tn1, 1);
    flip();
    read_b(I0{}, rbb, I1{}); stage_b(I1{}, I0{}); vmw(R4{}); bar();          // next K-tile's B0 (buffer 1) into the set M3 just released
    mma(I1{}, I0{}, ra, rba, I0{}); bar();
    // ================= odd K-tile (buffer 1): B0 is in rbb
    if constexpr (LAST) ab[1] = a_base(tm1, 1);
    read_a(I0{}, ra, I1{}); stage_a(I1{}, I0{}); vmw(R5{}); bar();
    lgkm0(); tie(ra); tie(rbb); mma(I0{}, I0{}, ra, rbb, I1{}); bar();
    read_b(I1{}, rba, I1{}); stage_b(I0{}, I1{}); vmw(R5{}); bar();
    lgkm0(); tie(rba); mma(I0{}, I1{}, ra, rba, I1{}); bar();
    read_a(I1{}, ra, I1{}); stage_a(I0{}, I1{}); vmw(R5{}); bar();
    lgkm0(); tie(ra); mma(I1{}, I1{}, ra, rba, I1{}); bar();
    flip();
    if constexpr (!LAST) { read_b(I0{}, rba, I0{}); stage_b(I1{}, I1{}); }
    vmw(R4{}); bar();
    mma(I1{}, I0{}, ra, rbb, I1{}); bar();
  };

  for (int it = 0; it < nmine; ++it) {
    zero_acc();
    if (wr == 1) bar();                                  // waves 4-7 run one barrier interval behind from here on
    // pre-phase: first B0 of the tile, and the region staging that the previous tile's last phase deferred (B1 of K-tile 1)
    read_b(I0{}, rba, I0{});
    stage_b(I1{}, I1{});
    for (int kp = 2; kp < nk; kp += 2) pair(std::false_type{});
    pair(std::true_type{});
    if (wr == 0) bar();                                  // waves 0-3 wait for waves 4-7's last MFMA cluster: all eight run the epilogue together
    epi(tm, tn, accs);
    tm = tm1; tn = tn1;
    tile_coords(it + 2 < nmine ? it + 2 : nmine - 1, tm1, tn1);
    lgkm0();
    bar();                                               // every patch is free again before the next tile stages into buffer 1's B1 / A1
  }
  vmwait<0>();                                           // no LDS-DMA may outlive the workgroup
}

}  // namespace pp
