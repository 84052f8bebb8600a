// bf16 MFMA GEMM with fused epilogues for the encoder's dense contractions (K5, K6, K9, K11, K12, K13, K14) -- gfx950.
//
//   C[M, N] = sum over K-segments  A_seg[rowmap(m), :] . W_seg[n, :]^T
//
// * operands are bf16 "hi" planes, optionally with "lo" planes (x = hi + lo, |x - hi - lo| <= 2^-17 |x|).
//   TERMS = 1: acc += a_hi b_hi.  TERMS = 3: acc += a_hi b_lo + a_lo b_hi + a_hi b_hi  (split-bf16: the mode
//   that meets the 1e-3 hidden-state parity bound; DESIGN.md "Numerics").  Accumulation is always fp32 MFMA.
// * K-segments let one kernel serve: plain linears (1 segment), linears with a LoRA term
//   ([x | u] . [W | B]^T, 2 segments) and the stride-2 conv stem as an implicit GEMM (3 taps = 3 segments whose
//   source row is 2 s + tap - 1, rows outside the clip reading as zero).
// * block tiles: 128 x 256 on 4 waves (1 x 4, each wave 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16), 32 KB of LDS,
//   so TWO independent workgroups share a CU and cover each other's barrier bubbles; 128 x 128 (4 waves, 2 x 2) for N
//   not a multiple of 256 and for small M.
// * the two operands take DIFFERENT routes on purpose.  Only the ACTIVATION tile goes through LDS (16-byte LDS-DMA,
//   double buffered, rows XOR-swizzled on the source address and on the ds_read_b128); the WEIGHT fragments -- static
//   within a step and owned by the library -- are read straight into registers from a fragment-major copy of the
//   weights (w_frag_index in common.h: one coalesced 1 KB block per fragment), one K-tile ahead: no LDS write, LDS read
//   or swizzle for half of the operand bytes.  (The kernel is power-limited, not schedule-limited: DESIGN.md 4.2.)
// * K loop: a K-tile is (BK / 32) x TM steps of TN x TERMS MFMAs; the A fragments of the next step are read from LDS,
//   and one or two of the next K-tile's memory operations (A DMA pieces, W fragment loads) are issued, before each
//   step's MFMAs; one barrier per K-tile.
// * 1-D grid with an XCD-aware remap (each XCD owns a contiguous run of tiles) walked in groups of 4 row panels.
// * epilogue: accumulators are transposed through a per-wave LDS patch so every store is a whole 256 B row
//   segment; side inputs are prefetched one strip ahead.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_pp.h"

#ifndef AWT_GEMM_GM
#define AWT_GEMM_GM 8   // tools/gemm_gm_sweep.py: 6 - 8 row panels per group are ~1 % ahead of 4 and 16 on every encoder shape
#endif

namespace {

#ifndef AWT_GEMM_NT_STORE
#define AWT_GEMM_NT_STORE 1   // f16f8 kernel: operand-plane outputs leave as streaming (non-temporal) stores: whole 256-byte row segments per instruction, read next by
#endif                        // another kernel; -0.3 ms per step.  (NOT for the 4 / 8-byte stores of attention / LayerNorm: those need L2 write combining, +60 % there.)
#ifndef AWT_GEMM_WDEC
#define AWT_GEMM_WDEC 1   // f16f8 kernel: the K-tile barrier waits for the LDS-DMA only; W fragment loads are waited for at their consumers
#endif
int g_gm = AWT_GEMM_GM;   // row panels per tile group (awt_tuning_set "gemm_gm")
constexpr int kMaxSeg = 3;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

struct GemmArgs {
  int M, N, nseg, tiles_m, tiles_n;
  int group_n;           // > 0: walk tiles in groups of group_n column tiles (weight slice L2-resident); 0: groups of GM row panels
  int gm;                // row panels per group (tuning knob "gemm_gm"; default AWT_GEMM_GM)
  GemmSeg seg[kMaxSeg];
  GemmOut out;
  const bf16_t* zeros;   // >= 128 zero bytes (padding rows of the conv stem)
  // batched launch (gemm_kernel<..., BATCH = true>, blockIdx.y = matrix index): element strides of segment 0's activation planes,
  // of its fragment-major weight planes and of the fp32 output / residual between consecutive matrices
  int64_t bs_a, bs_w, bs_o;
};

// block-tile configurations: WM x WN waves, each wave TM x TN tiles of 16 x 16
struct Cfg64 { static constexpr int WM = 2, WN = 2, TM = 2, TN = 4; };         // 64 x 128: small M (a few clips), where the chip is filled by tile count
struct Cfg128 { static constexpr int WM = 2, WN = 2, TM = 4, TN = 4; };
struct CfgW4 { static constexpr int WM = 1, WN = 4, TM = 8, TN = 4; };        // 128 x 256 on 4 waves: two independent workgroups per CU

template <int BK> __device__ __forceinline__ int swz(int row);
// BK = 64: 128-byte rows, 8 chunks of 16 B; BK = 32: 64-byte rows, 4 chunks.  See DESIGN.md "LDS images".
template <> __device__ __forceinline__ int swz<64>(int row) { return (row >> 1) & 7; }
template <> __device__ __forceinline__ int swz<32>(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }  // {0,2,3,1}

// WX (TERMS == 3 only): the weights are exactly representable in the operand type, their lo plane is zero -- it is neither loaded nor multiplied
// (two products per fragment pair: a_hi w + a_lo w).  Used for the split-fp16 mode on checkpoints released in half precision.
template <int TERMS, int BK, class CFG, bool WX = false>
struct Tile {
  static constexpr int BM = CFG::WM * CFG::TM * 16, BN = CFG::WN * CFG::TN * 16;
  static constexpr int THREADS = CFG::WM * CFG::WN * 64;
  static constexpr int NP = (TERMS == 3) ? 2 : 1;           // planes of the A operand: hi (+ lo)
  static constexpr int PLANE_A = BM * BK * 2;               // bytes
  static constexpr int STAGE = NP * PLANE_A;                // only A is staged in LDS
  static constexpr int CPR = BK / 8;                        // 16-byte chunks per row
  static constexpr int ITERS_A = (BM * CPR) / THREADS;      // LDS-DMA instructions per thread per plane
  static constexpr int KS = BK / 32;                        // MFMA k-steps per K-tile
  static constexpr int NA = NP * ITERS_A;                   // A DMA pieces per thread per K-tile
  static constexpr int NPW = WX ? 1 : NP;                   // planes of the W operand actually loaded
  static constexpr int NB = NPW * KS * CFG::TN;             // W fragment loads per lane per K-tile
  static constexpr int LDS_BYTES = (2 * STAGE > CFG::WM * CFG::WN * 16 * 68 * 4) ? 2 * STAGE : CFG::WM * CFG::WN * 16 * 68 * 4;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Per-thread state of the operand streams.  The rows a thread copies are the same for every K-tile, so the (row-mapped)
// A source pointers are computed once per K-segment and advanced by BK elements per K-tile; the W fragment pointer of
// the lane (its n-tile row of fragment blocks) likewise.
template <int TERMS, int BK, class CFG, bool WX = false>
struct Stager {
  using T = Tile<TERMS, BK, CFG, WX>;
  const bf16_t* a_hi[T::ITERS_A]; const bf16_t* a_lo[T::ITERS_A];
  const bf16_t* w_hi; const bf16_t* w_lo;     // lane's 16 bytes of fragment block (n-tile of the wave's first column tile, k-step w_k0)
  int64_t w_tile_stride;                       // elements between consecutive n-tiles = ksteps * 512
  int64_t za = 0, zw = 0;                      // batched launch: this matrix' offset into the activation / weight planes (0 otherwise: folded away)
  int si, kk, nk;

  __device__ __forceinline__ void open_segment(const GemmArgs& g, int seg, int m0, int n0, int wc, int wave, int lane) {
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
      const int64_t aoff = ((int64_t)grp * sg.rows_in + sr) * sg.lda + c * 8 + za;
      a_hi[it] = ok ? sg.a_hi + aoff : nullptr;
      a_lo[it] = (ok && TERMS == 3) ? sg.a_lo + aoff : nullptr;
    }
    w_tile_stride = (int64_t)sg.w_ksteps * 512;
    const int nt0 = (n0 >> 4) + wc * CFG::TN;                   // the wave's first 16-column tile
    const int64_t woff = ((int64_t)nt0 * sg.w_ksteps + sg.w_k0) * 512 + lane * 8 + zw;
    w_hi = sg.w_hi + woff;
    w_lo = (TERMS == 3) ? sg.w_lo + woff : nullptr;
  }
  // memory operation OP of the tile this stager points at: OP < NA is an A LDS-DMA piece, the rest are W fragment loads
  template <int OP>
  __device__ __forceinline__ void issue(const GemmArgs& g, char* stage_base, int wave, bf16x8 (&nbh)[T::KS][CFG::TN], bf16x8 (&nbl)[T::KS][CFG::TN]) {
    static_assert(OP >= 0 && OP < T::NA + T::NB, "operation index");
    if constexpr (OP < T::NA) {
      constexpr int plane = OP % T::NP, it = OP / T::NP;
      char* dst = stage_base + (it * T::THREADS + wave * 64) * 16 + plane * T::PLANE_A;
      const bf16_t* src = plane == 0 ? a_hi[it] : a_lo[it];
      glds16(src ? (const void*)(src + kk * BK) : (const void*)g.zeros, dst);
    } else {
      constexpr int q = OP - T::NA;
      constexpr int plane = q % T::NPW, r = q / T::NPW, j = r % CFG::TN, ks = r / CFG::TN;
      const int64_t off = (int64_t)j * w_tile_stride + (int64_t)(kk * T::KS + ks) * 512;
      if (plane == 0) nbh[ks][j] = *reinterpret_cast<const bf16x8*>(w_hi + off);
      else nbl[ks][j] = *reinterpret_cast<const bf16x8*>(w_lo + off);
    }
  }
  __device__ __forceinline__ void advance(const GemmArgs& g, int m0, int n0, int wc, int wave, int lane) {
    if (++kk == nk && si + 1 < g.nseg) open_segment(g, si + 1, m0, n0, wc, wave, lane);
  }
};

// Epilogue on four consecutive columns of one row (the accumulators are transposed through LDS first, so a lane owns
// a 16-byte fp32 / 8-byte bf16 piece of a row and 16 lanes cover 64 contiguous columns).  Side inputs (residual row,
// positional table, saved pre-activation) are loaded by load_side4 one strip AHEAD of their use: in the epilogue every
// wave of the workgroup is past its last MFMA, so a load-then-use per strip would expose its latency eight times.
template <int EPI, bool FULL = false>
__device__ __forceinline__ float4 load_side4(const GemmOut& o, int m, int n, int M) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!FULL) { if (m >= M || n >= o.n_valid) return r; }
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

// OP = precision of plane outputs (PREC_BF16X3: bf16 hi / lo; PREC_F16X3: fp16 hi / lo; PREC_F16F8: fp16 + two e4m3 planes)
// FULL / b: as store_out8_f8 below (no per-row predicate inside a full tile; the lane's bias values are loaded once per tile)
template <int EPI, int OP, bool FULL>
__device__ __forceinline__ void store_out4(const GemmOut& o, int m, int n, float4 acc, float4 side, float4 b, int M) {
  if constexpr (!FULL) { if (m >= M || n >= o.n_valid) return; }
  float v[4] = {acc.x, acc.y, acc.z, acc.w};
  const float sd[4] = {side.x, side.y, side.z, side.w};
  v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
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
    float inv8 = pow2f(-kF8Act), lim8 = 448.0f * pow2f(-kF8Act);          // PREC_F16F8: exponent of the e4m3 planes by operand role (store_out8_f8)
    if (EPI == EPI_QKV) {
      const int d = o.H * 64;
      const int which = n / d, within = n - which * d;
      inv8 = which == 0 ? pow2f(-kF8Q) : pow2f(-kF8KV); lim8 = which == 0 ? 448.0f * pow2f(-kF8Q) : 448.0f * pow2f(-kF8KV);
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
    if constexpr (OP == PREC_F16F8) {
      bf16_t h[4]; float l[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { h[t] = f32_to_f16(v[t]); l[t] = v[t] - f16_to_f32(h[t]); }
      *reinterpret_cast<uint2*>(o.hi + off) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
      *reinterpret_cast<unsigned*>(o.hi8 + off) = fp8x4_rt(v[0], v[1], v[2], v[3], lim8, inv8);
      *reinterpret_cast<unsigned*>(o.lo8 + off) = fp8x4_rt(l[0], l[1], l[2], l[3], lim8 * pow2f(-kF8Lo), inv8 * pow2f(-kF8Lo));
    } else {
      bf16_t hi[4], lo[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) split16<OP == PREC_F16X3>(v[t], hi[t], lo[t]);
      *reinterpret_cast<uint2*>(o.hi + off) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
      if (o.lo) *reinterpret_cast<uint2*>(o.lo + off) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
    }
  }
}

// Eight consecutive columns of one row in the f16f8 output format: 16-byte fp16 and 8-byte e4m3 stores (the 4-column form
// writes 4-byte pieces of the e4m3 planes: twice the store instructions for the same bytes).  n is a multiple of 8, so the
// eight columns never straddle a head or a q / k / v boundary of EPI_QKV.
// FULL: the tile lies inside [0, M) x [0, n_valid) (decided once per workgroup): no per-row predicate, hence no branch between the stores of a strip --
// with one, hipcc opened every row group with s_waitcnt vmcnt(0), i.e. waited for the previous group's stores to be acknowledged, 16 times per tile.
// b0 / b1: the lane's eight bias values, loaded once per tile (its columns are the same in every strip).
template <int EPI, bool FULL, bool ILV = false>
__device__ __forceinline__ void store_out8_f8(const GemmOut& o, int m, int n, float4 acc0, float4 acc1, float4 side0, float4 side1, float4 b0, float4 b1, int M) {
  // EPI_BF16_GELU_SAVE / EPI_BF16_DGELU: the MLP's forward / backward GEMMs of the training step with backward_terms = 5
  if constexpr (!FULL) { if (m >= M || n >= o.n_valid) return; }
  float v[8] = {acc0.x, acc0.y, acc0.z, acc0.w, acc1.x, acc1.y, acc1.z, acc1.w};
  const float sd[8] = {side0.x, side0.y, side0.z, side0.w, side1.x, side1.y, side1.z, side1.w};
  v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  if (EPI == EPI_F32 || EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS) {
    float* dst = o.f32 + (int64_t)m * o.ldo + n;
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = EPI == EPI_F32_RESID ? v[t] + sd[t] : (EPI == EPI_F32_GELU_POS ? gelu_erf(v[t]) + sd[t] : v[t]);
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    return;
  }
  int64_t off;
  // e4m3 planes of x 2^S: the value is clamped to +-448 2^-S and the conversion applies the scale (common.h fp8x4); S by operand role
  float inv8 = pow2f(-kF8Act), lim8 = 448.0f * pow2f(-kF8Act);
  if (EPI == EPI_QKV) {
    const int d = o.H * 64;
    const int which = n / d, within = n - which * d;
    inv8 = which == 0 ? pow2f(-kF8Q) : pow2f(-kF8KV); lim8 = which == 0 ? 448.0f * pow2f(-kF8Q) : 448.0f * pow2f(-kF8KV);
    const int h = within >> 6, e = within & 63;
    const int b = m / o.S, s = m - b * o.S;
    if (which == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] *= o.scale;
    }
    off = (int64_t)which * o.plane_stride + (((int64_t)b * o.H + h) * o.S + s) * 64 + e;
  } else {
    off = (int64_t)m * o.ldo + n;
    if constexpr (EPI == EPI_BF16_GELU_SAVE) {   // the pre-activation as a bf16 hi / lo pair (what EPI_BF16_DGELU reads back), then the f16f8 planes of gelu(pre)
      bf16_t ph[8], pl[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) split_bf16(v[t], ph[t], pl[t]);
      *reinterpret_cast<uint4*>(o.hi2 + off) = make_uint4(pack2(ph[0], ph[1]), pack2(ph[2], ph[3]), pack2(ph[4], ph[5]), pack2(ph[6], ph[7]));
      if (o.lo2) *reinterpret_cast<uint4*>(o.lo2 + off) = make_uint4(pack2(pl[0], pl[1]), pack2(pl[2], pl[3]), pack2(pl[4], pl[5]), pack2(pl[6], pl[7]));
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = gelu_erf(v[t]);
    } else if constexpr (EPI == EPI_BF16_DGELU) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float x = sd[t];   // d/dx gelu(x) = Phi(x) + x phi(x), x = the saved pre-activation (load_side4)
        v[t] *= 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
        v[t] = __builtin_amdgcn_fmed3f(v[t], -kF16Max, kF16Max);   // a gradient plane: saturate rather than overflow fp16 (common.h f16f8x4 SAT)
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = EPI == EPI_BF16_GELU ? gelu_erf(v[t]) : v[t] * o.scale;
    }
  }
  const float invl8 = inv8 * pow2f(-kF8Lo), liml8 = lim8 * pow2f(-kF8Lo);
  bf16_t h[8]; float l[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { h[t] = f32_to_f16(v[t]); l[t] = v[t] - f16_to_f32(h[t]); }
  if constexpr (ILV) {   // split lines (Act::ilv): the eight columns are an eighth of one 64-element line pair: fp16 bytes 2 e, hi8 128 + e, lo8 192 + e
    char* line = o.ilv + (off >> 6) * 256;
    const int e = (int)(off & 63);
    __builtin_nontemporal_store((i32x4_t){(int)pack2(h[0], h[1]), (int)pack2(h[2], h[3]), (int)pack2(h[4], h[5]), (int)pack2(h[6], h[7])}, reinterpret_cast<i32x4_t*>(line + e * 2));
    __builtin_nontemporal_store((i32x2){(int)fp8x4_rt(v[0], v[1], v[2], v[3], lim8, inv8), (int)fp8x4_rt(v[4], v[5], v[6], v[7], lim8, inv8)}, reinterpret_cast<i32x2*>(line + 128 + e));
    __builtin_nontemporal_store((i32x2){(int)fp8x4_rt(l[0], l[1], l[2], l[3], liml8, invl8), (int)fp8x4_rt(l[4], l[5], l[6], l[7], liml8, invl8)}, reinterpret_cast<i32x2*>(line + 192 + e));
    return;
  }
#if AWT_GEMM_NT_STORE   // the planes are read next by another kernel, not by this one -> streaming stores (profiles/r03_gemm_experiments.txt)
  __builtin_nontemporal_store((i32x4_t){(int)pack2(h[0], h[1]), (int)pack2(h[2], h[3]), (int)pack2(h[4], h[5]), (int)pack2(h[6], h[7])}, reinterpret_cast<i32x4_t*>(o.hi + off));
  if (EPI == EPI_QKV && o.skip_v8 && n >= 2 * o.H * 64) return;
  if (o.hi8) __builtin_nontemporal_store((i32x2){(int)fp8x4_rt(v[0], v[1], v[2], v[3], lim8, inv8), (int)fp8x4_rt(v[4], v[5], v[6], v[7], lim8, inv8)}, reinterpret_cast<i32x2*>(o.hi8 + off));
  __builtin_nontemporal_store((i32x2){(int)fp8x4_rt(l[0], l[1], l[2], l[3], liml8, invl8), (int)fp8x4_rt(l[4], l[5], l[6], l[7], liml8, invl8)}, reinterpret_cast<i32x2*>(o.lo8 + off));
#else
  *reinterpret_cast<uint4*>(o.hi + off) = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
  if (EPI == EPI_QKV && o.skip_v8 && n >= 2 * o.H * 64) return;      // v: fp16 plane only
  if (o.hi8) *reinterpret_cast<uint2*>(o.hi8 + off) = make_uint2(fp8x4_rt(v[0], v[1], v[2], v[3], lim8, inv8), fp8x4_rt(v[4], v[5], v[6], v[7], lim8, inv8));   // null: see store_act4
  *reinterpret_cast<uint2*>(o.lo8 + off) = make_uint2(fp8x4_rt(l[0], l[1], l[2], l[3], liml8, invl8), fp8x4_rt(l[4], l[5], l[6], l[7], liml8, invl8));
#endif
}

template <int TERMS, int BK, int EPI, class CFG, bool F16, bool WX = false, bool BATCH = false>
__global__ __launch_bounds__(CFG::WM * CFG::WN * 64, 2) void gemm_kernel(GemmArgs g) {
  static_assert(!BATCH || EPI == EPI_F32 || EPI == EPI_F32_RESID, "batched launches write fp32 matrices");
  static_assert(!WX || TERMS == 3, "the exact-weight form belongs to the split product");
  using T = Tile<TERMS, BK, CFG, WX>;
  using ST = Stager<TERMS, BK, CFG, WX>;
  constexpr int TM = CFG::TM, TN = CFG::TN, KS = T::KS, STEPS = KS * TM, NOPS = T::NA + T::NB;
  constexpr int OPS_PER_STEP = (NOPS + STEPS - 1) / STEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // XCD-aware bijective remap: hardware deals block b to XCD b % 8; give each XCD a contiguous run of tiles
  const int nwg = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  // Tile order inside an XCD's run (launch_one picks it from the shape):
  //  group_n = 0 (default): groups of GM row panels, row panel fastest -- GM activation panels stay L2-resident and each
  //    weight tile is read by GM workgroups at once.
  //  group_n > 0: groups of group_n column tiles, walked row panel by row panel: the group's weight slice stays in the
  //    XCD's 4 MB L2 for the whole pass over M; the activations are streamed tiles_n / group_n times.
  const int GM = g.gm;
  int tm, tn;
  if (g.group_n > 0) {
    const int grp = tile / (g.group_n * g.tiles_m);
    const int gn = min(g.group_n, g.tiles_n - grp * g.group_n);
    const int within = tile - grp * g.group_n * g.tiles_m;
    tm = within / gn; tn = grp * g.group_n + (within - tm * gn);
  } else {
    const int grp = tile / (GM * g.tiles_n);
    const int gm = min(GM, g.tiles_m - grp * GM);
    const int within = tile - grp * GM * g.tiles_n;
    tn = within / gm; tm = grp * GM + (within - tn * gm);
  }
  const int m0 = tm * T::BM, n0 = tn * T::BN;

  int ktiles = 0;
  for (int s = 0; s < g.nseg; ++s) ktiles += g.seg[s].K / BK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / CFG::WN, wc = wave - wr * CFG::WN;
  const int frow = lane & 15, fq = lane >> 4;

  bf16x8 bh[KS][TN], bl[KS][TN], nbh[KS][TN], nbl[KS][TN], ah[2], al[2];
  ST st;
  if constexpr (BATCH) { st.za = (int64_t)blockIdx.y * g.bs_a; st.zw = (int64_t)blockIdx.y * g.bs_w; }
  st.open_segment(g, 0, m0, n0, wc, wave, lane);
  // prologue: K-tile 0 -- A into LDS stage 0 (burst), W fragments into the "next" registers
  auto issue_all = [&](char* stage) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (st.template issue<I>(g, stage, wave, nbh, nbl), ...); }(std::make_integer_sequence<int, NOPS>{});
  };
  issue_all(smem);
  st.advance(g, m0, n0, wc, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int arow0 = wr * TM * 16 + frow;       // + 16 i
  auto load_a = [&](const char* stage, int ks, int i, bf16x8& h, bf16x8& l) {
    const int row = arow0 + 16 * i;
    const int off = row * (BK * 2) + (((ks * 4 + fq) ^ swz<BK>(row)) << 4);
    h = *reinterpret_cast<const bf16x8*>(stage + off);
    if (TERMS == 3) l = *reinterpret_cast<const bf16x8*>(stage + T::PLANE_A + off);
  };
  // step s = (ks, i): A row-tile i at k-step ks against the wave's TN W fragments
  auto step = [&](auto s_t, const char* cur, char* nxt, bool has_next) {
    constexpr int s = decltype(s_t)::value, ks = s / TM, i = s % TM;
    // a (free) use of this step's A fragments: the compiler's wait for them lands HERE, where they have had a whole
    // step to arrive, instead of behind the next step's reads
    if (TERMS == 3) asm volatile("" ::"v"(ah[s & 1]), "v"(al[s & 1]));
    else asm volatile("" ::"v"(ah[s & 1]));
    if constexpr (s + 1 < STEPS) load_a(cur, (s + 1) / TM, (s + 1) % TM, ah[(s + 1) & 1], al[(s + 1) & 1]);
    if (has_next) {   // this step's share of the next K-tile's memory operations (never a burst: see the header)
      [&]<int... O>(std::integer_sequence<int, O...>) {
        ([&] {
          constexpr int op = s * OPS_PER_STEP + O;
          if constexpr (op < NOPS) st.template issue<op>(g, nxt, wave, nbh, nbl);
        }(), ...);
      }(std::make_integer_sequence<int, OPS_PER_STEP>{});
    }
    __builtin_amdgcn_sched_barrier(0);   // reads / loads of this step are issued before its MFMAs
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (TERMS == 3) {
        if constexpr (!WX) acc[i][j] = mfma16<F16>(ah[s & 1], bl[ks][j], acc[i][j]);
        acc[i][j] = mfma16<F16>(al[s & 1], bh[ks][j], acc[i][j]);
      }
      acc[i][j] = mfma16<F16>(ah[s & 1], bh[ks][j], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int kt = 0; kt < ktiles; ++kt) {
    const char* cur = smem + (kt & 1) * T::STAGE;
    char* nxt = smem + ((kt + 1) & 1) * T::STAGE;
    const bool has_next = kt + 1 < ktiles;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < TN; ++j) { bh[ks][j] = nbh[ks][j]; if (TERMS == 3 && !WX) bl[ks][j] = nbl[ks][j]; }
    load_a(cur, 0, 0, ah[0], al[0]);
    [&]<int... S>(std::integer_sequence<int, S...>) { (step(std::integral_constant<int, S>{}, cur, nxt, has_next), ...); }(std::make_integer_sequence<int, STEPS>{});
    if (has_next) st.advance(g, m0, n0, wc, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: the C/D layout of the 16x16 MFMA (col = lane & 15, row = (lane >> 4) * 4 + reg) would give 2-4 byte
  // scattered stores; each wave instead transposes one 16-row x 64-column strip at a time through a private LDS
  // patch (the staging buffers are dead: the loop's last barrier has passed) and writes whole 256-byte rows.
  static_assert(TN == 4, "epilogue strips are 64 columns wide");
  constexpr int PITCH = 68;  // floats; 16 x 68 x 4 B = 4352 B per wave
  float* patch = reinterpret_cast<float*>(smem) + wave * (16 * PITCH);
  const int em0 = m0 + wr * TM * 16, en = n0 + wc * 64 + frow * 4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.out.bias && en < g.out.n_valid) bias4 = *reinterpret_cast<const float4*>(g.out.bias + en);
  GemmOut out_z;                               // batched launch: the output / residual of this matrix
  if constexpr (BATCH) { out_z = g.out; out_z.f32 += (int64_t)blockIdx.y * g.bs_o; if (out_z.resid) out_z.resid += (int64_t)blockIdx.y * g.bs_o; }
  const GemmOut& gout = BATCH ? out_z : g.out;
  auto strips = [&](auto full_t) {
    constexpr bool FULL = decltype(full_t)::value;
    float4 side[4], side_next[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) side[it] = load_side4<EPI, FULL>(gout, em0 + fq + 4 * it, en, g.M);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) patch[(fq * 4 + rr) * PITCH + j * 16 + frow] = acc[i][j][rr];
      if (i + 1 < TM) {
#pragma unroll
        for (int it = 0; it < 4; ++it) side_next[it] = load_side4<EPI, FULL>(gout, em0 + (i + 1) * 16 + fq + 4 * it, en, g.M);
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rl = fq + 4 * it;
        const float4 v = *reinterpret_cast<const float4*>(patch + rl * PITCH + frow * 4);
        store_out4<EPI, F16 ? PREC_F16X3 : PREC_BF16X3, FULL>(gout, em0 + i * 16 + rl, en, v, side[it], bias4, g.M);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) side[it] = side_next[it];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // wave-uniform: the whole block tile inside the matrix -> no per-row predicates (and no branches) between the stores
  if (m0 + T::BM <= g.M && n0 + T::BN <= g.out.n_valid) strips(std::true_type{}); else strips(std::false_type{});
}


// ================================================================================================ PREC_F16F8
// Same GEMM in the f16f8 operand format (common.h): per 64-deep K-tile and fragment pair
//     a16 w16                 4 k-steps of v_mfma_f32_32x32x16_f16          (32 cycles each)
//     a8 wl8 + al8 w8         2 x v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)   (64 cycles each)
// = 256 matrix-pipe cycles where the split-bf16 kernel spends 384, with the same operand bytes (4 per element: fp16 + two
// e4m3 planes) and ONE barrier per 64 of K instead of two.  Block tiles: 128 x 256 on 4 waves (1 x 4, wave tile 128 x 64 =
// 4 x 2 MFMA tiles of 32 x 32) and 128 x 128 (2 x 2 waves of 64 x 64) for N not a multiple of 256 / small M; 32 KB of LDS per
// stage (A16 16 KB | A8 8 KB | Al8 8 KB), two stages, two workgroups per CU.  Operand routes as in the kernel above: the
// activation planes by 16-byte LDS-DMA into swizzled images, the weight fragments from their fragment-major copies straight
// into registers -- here as a register RING: the registers of k-step ks are reloaded for the next K-tile right after
// their last MFMA, a whole K-tile ahead of their next use, so the weights need 64 VGPRs, not 128.  Every global address is
// a wave-uniform base (SGPR pair) plus a 32-bit per-lane offset.  The W loads are inline asm and every wait is counted by
// hand (vmcnt retires in issue order): hipcc would wait vmcnt(0) at any use of an ordinary load while an LDS-DMA is in
// flight (cdna_hip_programming.md "Projection GEMM at M = 256" item 4b).  The last K-tile prefetches itself again (results
// unused) so that the loop body has no branches.
struct CfgF8W4 { static constexpr int WM = 1, WN = 4, TM = 4, TN = 2; };     // 128 x 256
struct CfgF8Sq { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };     // 128 x 128
// 256 x 256 on 8 waves (one workgroup per CU): two 128 x 256 halves that share nothing in LDS, only the W fragment loads -- the two wave
// rows request the same weight bytes within a few cycles of each other, so the second request is served by the CU's vector L1 and the
// L2 -> CU traffic per output tile drops from 96 KB to 64 KB per 128 x 256 x 64 of work.  Measured (tools/gemm_tile_sweep.py,
// profiles/r02_gemm_tile_sweep.txt): bit-identical results, 2 - 15 % SLOWER than two independent 128 x 256 workgroups per CU (all eight
// waves meet at one barrier per K-tile), so it is not selected automatically; awt_tuning_set("gemm_tile", 512) runs it.
struct CfgF8Big { static constexpr int WM = 2, WN = 4, TM = 4, TN = 2; };

template <int OFF>
__device__ __forceinline__ bf16x8 gload16(unsigned voff, const void* sbase) {
  bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the same, ordering the consumers of the named fragments behind the wait (an MFMA is not a memory operation: "memory" alone does not hold it back)
template <int N> __device__ __forceinline__ void wait_vm(bf16x8& f0, bf16x8& f1) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N) : "memory"); }
// ... the e4m3 W registers of a K-tile: every register the wait retires is tied to it, so no use of one can be scheduled above the wait and no copy of one can
// be taken before it ("memory" + sched_barrier alone left that to the register allocator)
template <int N> __device__ __forceinline__ void wait_vm8(bf16x8& f0, bf16x8& f1, bf16x8& f2, bf16x8& f3, bf16x8& f4, bf16x8& f5, bf16x8& f6, bf16x8& f7) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm4(bf16x8& f0, bf16x8& f1, bf16x8& f2, bf16x8& f3) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "n"(N) : "memory");
}
// LDS fragment reads with hand-counted waits.  With the inline-asm global loads in the loop the compiler's own waitcnt insertion fell
// back to s_waitcnt lgkmcnt(0) in front of every consumer (disassembly), i.e. every prefetched fragment waited for the youngest read
// as well.  lds_read16 issues the read, lgkm_wait<N>(frag) waits until at most N younger LDS operations are outstanding; taking the
// fragment as an in/out operand is what orders its consumers behind the wait.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void lgkm_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int N> __device__ __forceinline__ void lgkm_wait(bf16x8& f0, bf16x8& f1) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N)); }
__device__ __forceinline__ i32x8 cat8(bf16x8 lo, bf16x8 hi) {
  return __builtin_shufflevector(__builtin_bit_cast(i32x4_t, lo), __builtin_bit_cast(i32x4_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// The vector-memory operations a prefetching K-tile issues (for K-tile kt + 1), in program order -- vmcnt retires in this order, so every counted wait of the
// K loop is "the operations issued after the one waited for":
//     position 0 .. NDMA - 1                    the LDS-DMA pieces, DMA_PER_STEP of them ahead of each row tile's MFMAs of k-step 0
//     position NDMA + ks TN + j  (ks < KS)      fp16 W fragment j of k-step ks into the register ring, issued behind k-step ks's last MFMAs
//     position NDMA + KS TN .. N - 1            the e4m3 W planes, issued behind the e4m3 part
// A hand-written count that disagrees with this order waits for the wrong operation (garbage fragments, or -- during bring-up -- a load landing in registers
// the epilogue had reused: a memory fault), so the counts are computed here and nowhere else.
template <int TM, int TN, int NDMA, int NW8, int KS = 4>
struct F8IssueOrder {
  static constexpr int NW16 = KS * TN, N = NDMA + NW16 + NW8;
  static constexpr int DMA_PER_STEP = (NDMA + TM - 1) / TM;
  static constexpr int last_dma = NDMA - 1;
  static constexpr int last_w16(int ks) { return NDMA + ks * TN + TN - 1; }
  static constexpr int last_w8 = N - 1;
  // operations of THIS K-tile already issued when k-step ks's first MFMA group (row tile 0) waits: its own share of the DMA pieces in k-step 0, else every
  // DMA piece plus the ring reloads of the k-steps before it
  static constexpr int issued_at_step(int ks) { return ks == 0 ? DMA_PER_STEP : NDMA + ks * TN; }
  static constexpr int issued_at_f8 = NDMA + NW16;                    // ... when the e4m3 part starts
  static constexpr int younger(int pos_prev_tile, int issued_now) { return (N - 1 - pos_prev_tile) + issued_now; }
  // fp16 W fragments of k-step ks, loaded one K-tile ago: the rest of that K-tile's operations plus what this one has issued (PF) or nothing (last K-tile)
  template <bool PF> static constexpr int w16_wait(int ks) { return younger(last_w16(ks), PF ? issued_at_step(ks) : 0); }
  // e4m3 W planes, the previous K-tile's last operations
  template <bool PF> static constexpr int w8_wait() { return younger(last_w8, PF ? issued_at_f8 : 0); }
  // at the K-tile barrier, after this K-tile's last operation was issued: only its DMA pieces must have landed (AWT_GEMM_WDEC) / also its fp16 fragments
  static constexpr int dma_wait_at_barrier() { return N - 1 - last_dma; }
  static constexpr int w16_wait_at_barrier() { return N - 1 - last_w16(KS - 1); }
  static_assert(w16_wait<true>(0) == (KS - 1) * TN + NW8 + DMA_PER_STEP && w16_wait<true>(KS - 1) == (KS - 1) * TN + NW8 + NDMA && w16_wait<false>(1) == (KS - 2) * TN + NW8, "w16 waits");
  static_assert(w8_wait<true>() == NDMA + NW16 && w8_wait<false>() == 0 && dma_wait_at_barrier() == NW16 + NW8 && w16_wait_at_barrier() == NW8, "w8 / barrier waits");
  static_assert(w16_wait<true>(KS - 1) <= 63, "vmcnt is a 6-bit count");
};

// WX: every weight of the (single) K segment is exactly representable in fp16 (true of checkpoints stored in half precision), so its lo plane is
// zero and the x_hi w_lo cross term vanishes: one e4m3 MFMA per fragment pair (x_lo8 w_hi8) instead of two, and neither the A hi8 image nor the W lo8
// image is moved at all -- 1.5 instead of 2 bf16-MFMA-equivalents and 3 instead of 4 operand bytes per element, with the same result to rounding.
template <int EPI, class CFG, bool MULTI, bool WX = false>
__global__ __launch_bounds__(CFG::WM * CFG::WN * 64, CFG::WM * CFG::WN == 4 ? 2 : 1) void gemm_f8_kernel(GemmArgs g) {
  constexpr int WM = CFG::WM, WN = CFG::WN, TM = CFG::TM, TN = CFG::TN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 64, NT = WM * WN * 64;
  static_assert((BM == 128 && NT == 256) || (BM == 256 && NT == 512), "128 rows per four waves");
  constexpr int PL16 = BM * BK * 2, PL8 = BM * BK, STAGE = PL16 + 2 * PL8;     // 16 + 8 + 8 KB per 128 rows
  constexpr int IT16 = PL16 / 16 / NT, IT8 = PL8 / 16 / NT;                     // LDS-DMA pieces per thread: 4, 2 (+ 2)
  static_assert(!(WX && MULTI), "the exact-weight form is single-segment");
  constexpr int NDMA = IT16 + (WX ? 1 : 2) * IT8;                               // 8 (6)
  constexpr int NW16 = 4 * TN, NW8 = (WX ? 2 : 4) * TN;                                    // W loads per lane per K-tile: fp16 (4 k-steps x TN), e4m3 (2 planes x TN x 2)
  constexpr unsigned kInvalid = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = wave * 64 + lane;

  // tile order: as gemm_kernel (XCD-contiguous runs, groups of GM row panels)
  const int nwg = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qn = nwg >> 3, rn = nwg & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  const int GM = g.gm;
  int tm, tn;
  {
    const int grp = tile / (GM * g.tiles_n);
    const int gm = min(GM, g.tiles_m - grp * GM);
    const int within = tile - grp * GM * g.tiles_n;
    tn = within / gm; tm = grp * GM + (within - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave / WN, wc = wave - wr * WN;
  const int r32 = lane & 31, half = lane >> 5;

  int ktiles = 0;
  for (int s = 0; s < (MULTI ? g.nseg : 1); ++s) ktiles += g.seg[s].K / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){};

  // ---- operand streams: wave-uniform bases (SGPRs, advanced per K-tile) + 32-bit per-lane offsets (fixed per K-segment)
  unsigned a16o[IT16], a8o[IT8];                   // byte offsets of the thread's DMA pieces inside the fp16 / e4m3 planes
  const char *a16b, *a8b, *al8b;                   // plane bases + the current K-tile's column offset
  const char *w16b[TN], *w8b[TN], *wl8b[TN];       // block of n-tile j at the current K-tile
  const unsigned wl16 = lane * 16, wl32 = lane * 32;
  int si = 0, kk = 0, nk = 0;
  const int am0 = m0, an0 = n0;
  auto open_segment = [&](int seg) {
    si = seg; kk = 0;
    const GemmSeg& sg = g.seg[seg];
    nk = sg.K / BK;
    // The per-lane offsets are 32-bit and RELATIVE to the first source row of the tile, which goes into the 64-bit (SGPR) plane bases: a tile
    // spans at most 128 row_mul + 2 source rows, so the offsets stay small however large the activation planes are (a plane of 64+ clips of
    // Whisper-large's MLP hidden exceeds 4 GiB; offsets from the plane base overflowed there).
    const int mb = am0 < g.M ? am0 : g.M - 1;
    const int grp0 = mb / sg.rows_out;
    const int sr0 = (mb - grp0 * sg.rows_out) * sg.row_mul + sg.row_add;
    const int64_t base_row = (int64_t)grp0 * sg.rows_in + (sr0 > 0 ? sr0 : 0);       // wave-uniform; every valid source row of the tile is >= it
#pragma unroll
    for (int it = 0; it < IT16; ++it) {
      const int p = it * NT + tid, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
      int m = am0 + row; m = m < g.M ? m : g.M - 1;
      const int grp = m / sg.rows_out, r = m - grp * sg.rows_out;
      const int sr = r * sg.row_mul + sg.row_add;
      a16o[it] = (sr >= 0 && sr < sg.rows_in) ? (unsigned)((((int64_t)grp * sg.rows_in + sr - base_row) * sg.lda + c * 8) * 2) : kInvalid;
    }
#pragma unroll
    for (int it = 0; it < IT8; ++it) {
      const int p = it * NT + tid, row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
      int m = am0 + row; m = m < g.M ? m : g.M - 1;
      const int grp = m / sg.rows_out, r = m - grp * sg.rows_out;
      const int sr = r * sg.row_mul + sg.row_add;
      a8o[it] = (sr >= 0 && sr < sg.rows_in) ? (unsigned)(((int64_t)grp * sg.rows_in + sr - base_row) * sg.lda + c * 16) : kInvalid;
    }
    a16b = (const char*)sg.a_hi + base_row * sg.lda * 2; a8b = (const char*)sg.a8 + base_row * sg.lda; al8b = (const char*)sg.al8 + base_row * sg.lda;
    const int nt0 = (an0 >> 5) + wc * TN;
    const int64_t w16_ts = (int64_t)sg.w_ksteps * 2 * 1024;          // bytes per 32-row n-tile of the fp16 plane: (K / 16) blocks of 1 KB
    const int64_t w8_ts = (int64_t)(sg.w_ksteps / 2) * 2048;         // bytes per n-tile of an e4m3 plane: (K / 64) blocks of 2 KB
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      w16b[j] = (const char*)sg.w_hi + (nt0 + j) * w16_ts + (int64_t)sg.w_k0 * 2 * 1024;
      w8b[j] = (const char*)sg.w8 + (nt0 + j) * w8_ts + (int64_t)(sg.w_k0 / 2) * 2048;
      wl8b[j] = (const char*)sg.wl8 + (nt0 + j) * w8_ts + (int64_t)(sg.w_k0 / 2) * 2048;
    }
  };
  // point the streams at the next K-tile (the last K-tile of the GEMM points at itself again)
  auto advance = [&]() {
    if (kk + 1 < nk) {
      ++kk;
      a16b += BK * 2; a8b += BK; al8b += BK;
#pragma unroll
      for (int j = 0; j < TN; ++j) { w16b[j] += 4 * 1024; w8b[j] += 2048; wl8b[j] += 2048; }
    } else if (MULTI && si + 1 < g.nseg) {
      open_segment(si + 1);
    }
  };
  // DMA piece OP (0 .. NDMA - 1) of the K-tile the streams point at
  auto dma = [&](auto op_t, char* stage) {
    constexpr int OP = decltype(op_t)::value;
    if constexpr (OP < IT16) {
      const unsigned o = a16o[OP];
      glds16(o != kInvalid ? (const void*)(a16b + o) : (const void*)g.zeros, stage + (OP * NT + wave * 64) * 16);
    } else if constexpr (OP < IT16 + IT8 && !WX) {
      const unsigned o = a8o[OP - IT16];
      glds16(o != kInvalid ? (const void*)(a8b + o) : (const void*)g.zeros, stage + PL16 + ((OP - IT16) * NT + wave * 64) * 16);
    } else {
      constexpr int P8 = OP - IT16 - (WX ? 0 : IT8);
      const unsigned o = a8o[P8];
      glds16(o != kInvalid ? (const void*)(al8b + o) : (const void*)g.zeros, stage + PL16 + PL8 + (P8 * NT + wave * 64) * 16);
    }
  };

  bf16x8 w16[4][TN];              // fp16 B fragments of the K-tile's four k-steps
  bf16x8 w8[TN][2], wl8[TN][2];   // e4m3 operands (32 bytes = two 16-byte halves)
  auto load_w16 = [&](auto ks_t) {
    constexpr int ks = decltype(ks_t)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) w16[ks][j] = gload16<ks * 1024>(wl16, w16b[j]);
  };
  auto load_w8 = [&]() {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      w8[j][0] = gload16<0>(wl32, w8b[j]); w8[j][1] = gload16<16>(wl32, w8b[j]);
      if constexpr (!WX) { wl8[j][0] = gload16<0>(wl32, wl8b[j]); wl8[j][1] = gload16<16>(wl32, wl8b[j]); }
    }
  };

  // ---- prologue: K-tile 0 into stage 0 and into the W registers
  open_segment(0);
  [&]<int... O>(std::integer_sequence<int, O...>) { (dma(std::integral_constant<int, O>{}, smem), ...); }(std::make_integer_sequence<int, NDMA>{});
  [&]<int... S>(std::integer_sequence<int, S...>) { (load_w16(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, 4>{});
  load_w8();
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // A fragment offsets inside a stage: fp16 image row r, chunk (2 ks + half) ^ ((r >> 1) & 7); e4m3 images row r, chunks
  // (2 half + c) ^ ((r >> 2) & 3).  Row tile i adds a multiple of 32 rows, which leaves both swizzle terms unchanged.
  // (kept as absolute LDS byte addresses in the CURRENT stage, flipped at the end of every K-tile; the row tile goes into the ds_read's
  // immediate offset.  Unrolling the K loop by two to make the stage an immediate as well cost 600+ spilled registers.)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(smem);
  unsigned a16a[4], a8a[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) a16a[ks] = lds0 + (wr * TM * 32 + r32) * 128 + (((2 * ks + half) ^ ((r32 >> 1) & 7)) << 4);
#pragma unroll
  for (int c = 0; c < 2; ++c) a8a[c] = lds0 + PL16 + (wr * TM * 32 + r32) * 64 + (((2 * half + c) ^ ((r32 >> 2) & 3)) << 4);

  constexpr int DMA_PER_STEP = (NDMA + TM - 1) / TM;    // every DMA piece is issued during k-step 0, ahead of all W reloads
  using ORD = F8IssueOrder<TM, TN, NDMA, NW8>;          // the issue order of the K-tile's memory operations and every counted wait derived from it
  static_assert(ORD::DMA_PER_STEP == DMA_PER_STEP && ORD::NW16 == NW16, "issue-order table and kernel disagree");
  // One K-tile.  PF (compile time): prefetch K-tile kt + 1 (A by LDS-DMA into the other stage, W into the register ring).  The
  // last K-tile is instantiated without any load: a load still in flight when the loop ends would land in registers the
  // epilogue has already reused (the compiler cannot see that an inline-asm load completes later).
  auto ktile = [&](auto pf_t, int kt) {
    constexpr bool PF = decltype(pf_t)::value;
    constexpr int STG = 0;                                // a16a / a8a already point into the current stage
    char* nxt = smem + ((kt + 1) & 1) * STAGE;
    if constexpr (PF) advance();        // from here on the streams point at K-tile kt + 1 (the W registers still hold K-tile kt)
    // ---- fp16 part: k-step ks, row tile i: one 16-byte A fragment against the wave's TN weight fragments.  A step is only TN = 2
    // MFMAs (64 matrix-pipe cycles) while an LDS read under load takes 100-150 cycles to return, so the fragments are read THREE steps
    // ahead into a four-entry ring (one step ahead, every step stalled on its fragment: the disassembly showed s_waitcnt lgkmcnt(0)
    // in front of every MFMA pair).
    constexpr int AD = MULTI ? 1 : 3;   // the multi-segment form has no registers left for a deeper ring (it spilled)
    bf16x8 af[AD + 1];
    auto read_a16 = [&](auto s_t) {
      constexpr int S2 = decltype(s_t)::value;
      af[S2 % (AD + 1)] = lds_read16<STG * STAGE + (S2 % TM) * 32 * 128>(a16a[S2 / TM]);
    };
    [&]<int... S>(std::integer_sequence<int, S...>) { (read_a16(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, AD>{});
    // e4m3 part operands: X = hi8 image, Y = lo8 image of row tile i (two 16-byte reads each); X(i + 1) is read behind the X MFMAs of
    // row tile i and lands while the Y MFMAs run (128 cycles), and vice versa: no extra registers, no exposed LDS latency after the
    // first row tile, whose reads go out behind the last two fp16 steps (not in the multi-segment form: no registers left, it spilled).
    bf16x8 ax[2], ay[2];
    auto read_x = [&](auto i_t) { constexpr int O = STG * STAGE + decltype(i_t)::value * 32 * 64; ax[0] = lds_read16<O>(a8a[0]); ax[1] = lds_read16<O>(a8a[1]); };
    auto read_y = [&](auto i_t) { constexpr int O = STG * STAGE + decltype(i_t)::value * 32 * 64 + PL8; ay[0] = lds_read16<O>(a8a[0]); ay[1] = lds_read16<O>(a8a[1]); };
    constexpr bool EARLY8 = !MULTI;
    // WX: only the lo8 image is read; row tile I + 1 is read before the MFMAs of row tile I into the other of the two fragment buffers (ax idles)
    auto read_yb = [&](auto i_t) {
      constexpr int I = decltype(i_t)::value, O = STG * STAGE + I * 32 * 64 + PL8;
      if constexpr (I & 1) { ax[0] = lds_read16<O>(a8a[0]); ax[1] = lds_read16<O>(a8a[1]); }
      else { ay[0] = lds_read16<O>(a8a[0]); ay[1] = lds_read16<O>(a8a[1]); }
    };
    [&]<int... S>(std::integer_sequence<int, S...>) {
      ([&] {
        constexpr int ks = S / TM, i = S % TM;
        if constexpr (S + AD < 4 * TM) read_a16(std::integral_constant<int, S + AD>{});
        if constexpr (EARLY8 && !WX && S == 4 * TM - 2) read_x(std::integral_constant<int, 0>{});
        if constexpr (EARLY8 && S == 4 * TM - 1) { if constexpr (WX) read_yb(std::integral_constant<int, 0>{}); else read_y(std::integral_constant<int, 0>{}); }
        if constexpr (PF && ks == 0) {
          [&]<int... O>(std::integer_sequence<int, O...>) {
            ([&] { constexpr int op = i * DMA_PER_STEP + O; if constexpr (op < NDMA) dma(std::integral_constant<int, op>{}, nxt); }(), ...);
          }(std::make_integer_sequence<int, DMA_PER_STEP>{});
        }
        // LDS operations younger than this step's fragment: the (up to) three fragments read ahead, plus the first e4m3 reads
        constexpr int AHEAD = (4 * TM - 1 - S < AD ? 4 * TM - 1 - S : AD) + (EARLY8 && !WX && S >= 4 * TM - 2 ? 2 : 0) + (EARLY8 && S >= 4 * TM - 1 ? 2 : 0);
        lgkm_wait<AHEAD>(af[S % (AD + 1)]);
#if AWT_GEMM_WDEC
        // the fp16 W fragments of k-step ks were loaded a whole K-tile ago (ring); vmcnt retires in issue order, so the wait counts what was
        // issued since: the later k-steps' fragments of this K-tile (3 - ks) TN, its e4m3 planes NW8, and -- when prefetching -- the DMA
        // pieces (all of them from k-step 1 on, this step's share in k-step 0) and the ks TN fragments already reloaded for K-tile kt + 1
        if constexpr (i == 0) wait_vm<ORD::template w16_wait<PF>(ks)>(w16[ks][0], w16[ks][1]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<true>(af[S % (AD + 1)], w16[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PF && i == TM - 1) load_w16(std::integral_constant<int, ks>{});     // ring: this k-step's registers, for K-tile kt + 1
      }(), ...);
    }(std::make_integer_sequence<int, 4 * TM>{});
    // ---- e4m3 part: its W registers were loaded a K-tile ago, behind this K-tile's NDMA + NW16 younger operations
    static_assert(TN == 2, "the tied wait below names the e4m3 W registers of two column tiles");
    if constexpr (WX) wait_vm4<ORD::template w8_wait<PF>()>(w8[0][0], w8[0][1], w8[1][0], w8[1][1]);
    else wait_vm8<ORD::template w8_wait<PF>()>(w8[0][0], w8[0][1], w8[1][0], w8[1][1], wl8[0][0], wl8[0][1], wl8[1][0], wl8[1][1]);
    if constexpr (!EARLY8) { if constexpr (!WX) read_x(std::integral_constant<int, 0>{}); read_y(std::integral_constant<int, 0>{}); }
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        if constexpr (!WX) {
          lgkm_wait<2>(ax[0], ax[1]);                     // younger: Y(I)
          __builtin_amdgcn_sched_barrier(0);
          const i32x8 a8 = cat8(ax[0], ax[1]);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[I][j] = mfma32_f8<e8m0(-kF8Act), e8m0(-kF8Wgt - kF8Lo)>(a8, cat8(wl8[j][0], wl8[j][1]), acc[I][j]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (I + 1 < TM) read_x(std::integral_constant<int, I + 1>{});
        }
        if constexpr (WX) {
          if constexpr (I + 1 < TM) read_yb(std::integral_constant<int, I + 1>{});
          bf16x8 (&yb)[2] = (I & 1) ? ax : ay;
          lgkm_wait<(I + 1 < TM ? 2 : 0)>(yb[0], yb[1]);    // younger: Y(I + 1)
          __builtin_amdgcn_sched_barrier(0);
          const i32x8 al8 = cat8(yb[0], yb[1]);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[I][j] = mfma32_f8<e8m0(-kF8Act - kF8Lo), e8m0(-kF8Wgt)>(al8, cat8(w8[j][0], w8[j][1]), acc[I][j]);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          lgkm_wait<(I + 1 < TM ? 2 : 0)>(ay[0], ay[1]);    // younger: X(I + 1)
          __builtin_amdgcn_sched_barrier(0);
          const i32x8 al8 = cat8(ay[0], ay[1]);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[I][j] = mfma32_f8<e8m0(-kF8Act - kF8Lo), e8m0(-kF8Wgt)>(al8, cat8(w8[j][0], w8[j][1]), acc[I][j]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (I + 1 < TM) read_y(std::integral_constant<int, I + 1>{});
        }
      }(), ...);
    }(std::make_integer_sequence<int, TM>{});
    if constexpr (PF) {
      load_w8();
#if AWT_GEMM_WDEC
      // only the DMA pieces of K-tile kt + 1 must have landed before the barrier; its W fragments (NW16 + NW8 younger loads) stay in flight
      // across it and are waited for where they are consumed, a K-tile after their issue
      wait_vm<ORD::dma_wait_at_barrier()>();
#else
      // the DMA pieces and the fp16 W fragments of K-tile kt + 1 are older than the NW8 e4m3 loads just issued
      wait_vm<ORD::w16_wait_at_barrier()>();
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (PF) {
      const int flip = (kt & 1) ? -STAGE : STAGE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a16a[ks] += flip;
#pragma unroll
      for (int c = 0; c < 2; ++c) a8a[c] += flip;
    }
  };
  for (int kt = 0; kt + 1 < ktiles; ++kt) ktile(std::true_type{}, kt);
  ktile(std::false_type{}, ktiles - 1);

  // ---- epilogue: each wave transposes one 32-row x 64-column strip at a time through a private LDS patch (32 x 32 C/D
  // layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)); a lane then owns EIGHT consecutive columns of a row
  // (8 lanes per row, 8 rows per pass): 32-byte fp32 / 16-byte fp16 / 8-byte e4m3 stores, whole 256-byte row segments per plane.
  static_assert(TN == 2, "epilogue strips are 64 columns wide");
  constexpr int PITCH = 72;   // floats: 288-byte rows keep the two 16-byte reads of a lane 16-byte aligned
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);      // 9216 B per wave
  const int c8 = (lane & 7) * 8, r8 = lane >> 3;
  const int em0 = m0 + wr * TM * 32, en = n0 + wc * 64 + c8;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (g.out.bias && en < g.out.n_valid) { b0 = *reinterpret_cast<const float4*>(g.out.bias + en); b1 = *reinterpret_cast<const float4*>(g.out.bias + en + 4); }
  constexpr bool SIDE = EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS || EPI == EPI_BF16_DGELU;
  auto strips = [&](auto full_t) {
    constexpr bool FULL = decltype(full_t)::value;
    float4 side[4][2], side_next[4][2];
    auto load_sides = [&](float4 (&d)[4][2], int i) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        d[it][0] = load_side4<EPI, FULL>(g.out, em0 + i * 32 + r8 + 8 * it, en, g.M);
        d[it][1] = load_side4<EPI, FULL>(g.out, em0 + i * 32 + r8 + 8 * it, en + 4, g.M);
      }
    };
    if constexpr (SIDE) load_sides(side, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) patch[((rr & 3) + 8 * (rr >> 2) + 4 * half) * PITCH + j * 32 + r32] = acc[i][j][rr];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // the side inputs (residual rows, positional rows) of the NEXT strip are requested before this strip's stores: vmcnt retires in issue
      // order, so a load issued behind the stores would wait for their acknowledgement
      if constexpr (SIDE) { if (i + 1 < TM) load_sides(side_next, i + 1); }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rl = r8 + 8 * it;
        const float4 v0 = *reinterpret_cast<const float4*>(patch + rl * PITCH + c8), v1 = *reinterpret_cast<const float4*>(patch + rl * PITCH + c8 + 4);
        store_out8_f8<EPI, FULL>(g.out, em0 + i * 32 + rl, en, v0, v1, side[it][0], side[it][1], b0, b1, g.M);
      }
      if constexpr (SIDE) {
#pragma unroll
        for (int it = 0; it < 4; ++it) { side[it][0] = side_next[it][0]; side[it][1] = side_next[it][1]; }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // wave-uniform: the whole block tile inside the matrix (every tile of the encoder's shapes; only a ragged last row panel takes the other path)
  if (m0 + BM <= g.M && n0 + BN <= g.out.n_valid) strips(std::true_type{}); else strips(std::false_type{});
}

// ------------------------------------------------------------------------------------------------ the same GEMM on 16 x 16 MFMAs
// Under the package power limit the chip holds a higher clock on the 16 x 16 MFMA shapes than on the 32 x 32 ones at equal matrix-pipe cycles
// (MI355X_MICROARCH.md "DVFS give-back" item 7; measured here on this kernel's own instruction mix with random operands, tools/mfma_power.hip,
// profiles/r04_mfma_shape_mix.txt: 4.32 vs 5.08 us per round = 15 % less time for the identical arithmetic).  gemm_f8s_kernel is gemm_f8_kernel's 128 x 256
// four-wave tile with every product issued as 16 x 16 instructions:
//     a16 w16                  2 k-steps of v_mfma_f32_16x16x32_f16                       (16 cycles each)
//     a8 wl8 + al8 w8          1 x v_mfma_scale_f32_16x16x128_f8f6f4: lane (row r, block kb) holds hi8[32 kb ..] for kb < 2 and lo8[32 (kb - 2) ..] above,
//                              the weight operand lo8 | hi8 in the same block order -- both cross terms of the 64-deep K-tile in one instruction, and since
//                              2^-Act 2^(-Wgt - 11) is the scale of BOTH (hi8 wlo8 and lo8 whi8) ONE E8M0 scale per operand serves all four blocks
// = the same 256 matrix-pipe cycles per 32 x 32 x 64 of work, the same LDS images (A16 | A8 | Al8 by LDS-DMA) and bytes, the same K-loop skeleton (sixteen fp16
// steps of 64 cycles with a four-entry fragment ring, eight e4m3 steps of 128 cycles, W fragments in a register ring reloaded a K-tile ahead, counted waits from
// F8IssueOrder).  What changes is every fragment layout: 16-row A fragments (fp16: row lane & 15, chunk 4 ks + (lane >> 4); e4m3: plane (lane >> 5), chunks
// 2 ((lane >> 4) & 1) + q, images swizzled by (row >> 3) & 1 instead of (row >> 2) & 3), weights from their 16-row fragment-major copies (fp16: w_frag_index;
// e4m3: w8s_index, lo8 and hi8 interleaved per 2 KB block), and a 16 x 16 C layout (col = lane & 15, row = 4 (lane >> 4) + reg) into the epilogue's patch.
template <int N> __device__ __forceinline__ void wait_vm(bf16x8& f0, bf16x8& f1, bf16x8& f2, bf16x8& f3) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "n"(N) : "memory");
}

template <int EPI, bool MULTI>
__global__ __launch_bounds__(256, 2) void gemm_f8s_kernel(GemmArgs g) {
  constexpr int TM = 8, TN = 4, KS = 2;                                  // per wave: 8 x 4 tiles of 16 x 16; two 32-deep fp16 k-steps per K-tile
  constexpr int BM = 128, BN = 256, BK = 64, NT = 256;
  constexpr int PL16 = BM * BK * 2, PL8 = BM * BK, STAGE = PL16 + 2 * PL8;
  constexpr int IT16 = PL16 / 16 / NT, IT8 = PL8 / 16 / NT;             // 4, 2 (+ 2)
  constexpr int NDMA = IT16 + 2 * IT8, NW16 = KS * TN, NW8 = 2 * TN;
  constexpr unsigned kInvalid = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = wave * 64 + lane;

  const int nwg = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qn = nwg >> 3, rn = nwg & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  const int GM = g.gm;
  int tm, tn;
  {
    const int grp = tile / (GM * g.tiles_n);
    const int gm = min(GM, g.tiles_m - grp * GM);
    const int within = tile - grp * GM * g.tiles_n;
    tn = within / gm; tm = grp * GM + (within - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int wc = wave;
  const int r16 = lane & 15, kq = lane >> 4;

  int ktiles = 0;
  for (int s = 0; s < (MULTI ? g.nseg : 1); ++s) ktiles += g.seg[s].K / BK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){};

  unsigned a16o[IT16], a8o[IT8];
  const char *a16b, *a8b, *al8b;
  const char *w16b[TN], *w8b[TN];
  const unsigned wl16 = lane * 16, wl32 = lane * 32;
  int si = 0, kk = 0, nk = 0;
  auto open_segment = [&](int seg) {
    si = seg; kk = 0;
    const GemmSeg& sg = g.seg[seg];
    nk = sg.K / BK;
    const int mb = m0 < g.M ? m0 : g.M - 1;
    const int grp0 = mb / sg.rows_out;
    const int sr0 = (mb - grp0 * sg.rows_out) * sg.row_mul + sg.row_add;
    const int64_t base_row = (int64_t)grp0 * sg.rows_in + (sr0 > 0 ? sr0 : 0);
#pragma unroll
    for (int it = 0; it < IT16; ++it) {
      const int p = it * NT + tid, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
      int m = m0 + row; m = m < g.M ? m : g.M - 1;
      const int grp = m / sg.rows_out, r = m - grp * sg.rows_out;
      const int sr = r * sg.row_mul + sg.row_add;
      a16o[it] = (sr >= 0 && sr < sg.rows_in) ? (unsigned)((((int64_t)grp * sg.rows_in + sr - base_row) * sg.lda + c * 8) * 2) : kInvalid;
    }
#pragma unroll
    for (int it = 0; it < IT8; ++it) {
      const int p = it * NT + tid, row = p >> 2, c = (p & 3) ^ ((row >> 3) & 1);      // the 16-row fragment reads' swizzle (below)
      int m = m0 + row; m = m < g.M ? m : g.M - 1;
      const int grp = m / sg.rows_out, r = m - grp * sg.rows_out;
      const int sr = r * sg.row_mul + sg.row_add;
      a8o[it] = (sr >= 0 && sr < sg.rows_in) ? (unsigned)(((int64_t)grp * sg.rows_in + sr - base_row) * sg.lda + c * 16) : kInvalid;
    }
    a16b = (const char*)sg.a_hi + base_row * sg.lda * 2; a8b = (const char*)sg.a8 + base_row * sg.lda; al8b = (const char*)sg.al8 + base_row * sg.lda;
    const int nt0 = (n0 >> 4) + wc * TN;
    const int64_t w16_ts = (int64_t)sg.w_ksteps * 1024;               // bytes per 16-row n-tile of the fp16 copy: (K / 32) blocks of 1 KB
    const int64_t w8_ts = (int64_t)(sg.w_ksteps / 2) * 2048;          // bytes per n-tile of the e4m3 copy: (K / 64) blocks of 2 KB (lo8 | hi8)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      w16b[j] = (const char*)sg.ws16 + (nt0 + j) * w16_ts + (int64_t)sg.w_k0 * 1024;
      w8b[j] = (const char*)sg.ws8 + (nt0 + j) * w8_ts + (int64_t)(sg.w_k0 / 2) * 2048;
    }
  };
  auto advance = [&]() {
    if (kk + 1 < nk) {
      ++kk;
      a16b += BK * 2; a8b += BK; al8b += BK;
#pragma unroll
      for (int j = 0; j < TN; ++j) { w16b[j] += KS * 1024; w8b[j] += 2048; }
    } else if (MULTI && si + 1 < g.nseg) {
      open_segment(si + 1);
    }
  };
  auto dma = [&](auto op_t, char* stage) {
    constexpr int OP = decltype(op_t)::value;
    if constexpr (OP < IT16) {
      const unsigned o = a16o[OP];
      glds16(o != kInvalid ? (const void*)(a16b + o) : (const void*)g.zeros, stage + (OP * NT + wave * 64) * 16);
    } else if constexpr (OP < IT16 + IT8) {
      const unsigned o = a8o[OP - IT16];
      glds16(o != kInvalid ? (const void*)(a8b + o) : (const void*)g.zeros, stage + PL16 + ((OP - IT16) * NT + wave * 64) * 16);
    } else {
      constexpr int P8 = OP - IT16 - IT8;
      const unsigned o = a8o[P8];
      glds16(o != kInvalid ? (const void*)(al8b + o) : (const void*)g.zeros, stage + PL16 + PL8 + (P8 * NT + wave * 64) * 16);
    }
  };

  bf16x8 w16[KS][TN];             // fp16 B fragments of the K-tile's two k-steps
  bf16x8 w8[TN][2];               // e4m3 operands (32 bytes = two 16-byte halves): lo8 blocks on lanes 0-31, hi8 blocks on lanes 32-63
  auto load_w16 = [&](auto ks_t) {
    constexpr int ks = decltype(ks_t)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) w16[ks][j] = gload16<ks * 1024>(wl16, w16b[j]);
  };
  auto load_w8 = [&]() {
#pragma unroll
    for (int j = 0; j < TN; ++j) { w8[j][0] = gload16<0>(wl32, w8b[j]); w8[j][1] = gload16<16>(wl32, w8b[j]); }
  };

  open_segment(0);
  [&]<int... O>(std::integer_sequence<int, O...>) { (dma(std::integral_constant<int, O>{}, smem), ...); }(std::make_integer_sequence<int, NDMA>{});
  load_w16(std::integral_constant<int, 0>{}); load_w16(std::integral_constant<int, 1>{});
  load_w8();
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // A fragment addresses in the CURRENT stage (flipped per K-tile).  fp16 image: row r16 (+ 16 per row tile: immediate 2048), chunk (4 ks + kq) ^ ((row >> 1) & 7).
  // e4m3 images: plane kq >> 1 (hi8 | lo8), row r16 (+ 16 per row tile: immediate 1024), chunk (2 (kq & 1) + q) ^ ((row >> 3) & 1): per 16-lane group of a
  // ds_read_b128 the rows 0-3 | 12-15 at chunk c and 4-11 at chunk c + 2 (or the mirrored set) then cover all sixteen 16-byte slots of the 256-byte bank row.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(smem);
  unsigned a16a[KS], a8a[2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a16a[ks] = lds0 + r16 * 128 + (((4 * ks + kq) ^ ((r16 >> 1) & 7)) << 4);
#pragma unroll
  for (int q = 0; q < 2; ++q) a8a[q] = lds0 + PL16 + (kq >> 1) * PL8 + r16 * 64 + (((2 * (kq & 1) + q) ^ ((r16 >> 3) & 1)) << 4);

  constexpr int DMA_PER_STEP = (NDMA + TM - 1) / TM;
  using ORD = F8IssueOrder<TM, TN, NDMA, NW8, KS>;
  static_assert(ORD::DMA_PER_STEP == DMA_PER_STEP && ORD::NW16 == NW16, "issue-order table and kernel disagree");
  auto ktile = [&](auto pf_t, int kt) {
    constexpr bool PF = decltype(pf_t)::value;
    char* nxt = smem + ((kt + 1) & 1) * STAGE;
    if constexpr (PF) advance();
    // ---- fp16 part: KS x TM steps of TN MFMAs (64 matrix-pipe cycles), fragments read AD steps ahead into a ring
    constexpr int AD = MULTI ? 1 : 3;
    bf16x8 af[AD + 1];
    auto read_a16 = [&](auto s_t) {
      constexpr int S2 = decltype(s_t)::value;
      af[S2 % (AD + 1)] = lds_read16<(S2 % TM) * 16 * 128>(a16a[S2 / TM]);
    };
    [&]<int... S>(std::integer_sequence<int, S...>) { (read_a16(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, AD>{});
    // e4m3 operand of row tile i (two 16-byte reads): row tile i + 1 is read into the other buffer before the MFMAs of row tile i
    bf16x8 ax[2], ay[2];
    auto read_8 = [&](auto i_t) {
      constexpr int I = decltype(i_t)::value, O = I * 16 * 64;
      if constexpr (I & 1) { ay[0] = lds_read16<O>(a8a[0]); ay[1] = lds_read16<O>(a8a[1]); }
      else { ax[0] = lds_read16<O>(a8a[0]); ax[1] = lds_read16<O>(a8a[1]); }
    };
    constexpr bool EARLY8 = !MULTI;
    [&]<int... S>(std::integer_sequence<int, S...>) {
      ([&] {
        constexpr int ks = S / TM, i = S % TM;
        if constexpr (S + AD < KS * TM) read_a16(std::integral_constant<int, S + AD>{});
        if constexpr (EARLY8 && S == KS * TM - 1) read_8(std::integral_constant<int, 0>{});
        if constexpr (PF && ks == 0) {
          [&]<int... O>(std::integer_sequence<int, O...>) {
            ([&] { constexpr int op = i * DMA_PER_STEP + O; if constexpr (op < NDMA) dma(std::integral_constant<int, op>{}, nxt); }(), ...);
          }(std::make_integer_sequence<int, DMA_PER_STEP>{});
        }
        constexpr int AHEAD = (KS * TM - 1 - S < AD ? KS * TM - 1 - S : AD) + (EARLY8 && S >= KS * TM - 1 ? 2 : 0);
        lgkm_wait<AHEAD>(af[S % (AD + 1)]);
#if AWT_GEMM_WDEC
        if constexpr (i == 0) wait_vm<ORD::template w16_wait<PF>(ks)>(w16[ks][0], w16[ks][1], w16[ks][2], w16[ks][3]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<true>(af[S % (AD + 1)], w16[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PF && i == TM - 1) load_w16(std::integral_constant<int, ks>{});
      }(), ...);
    }(std::make_integer_sequence<int, KS * TM>{});
    // ---- e4m3 part
    wait_vm8<ORD::template w8_wait<PF>()>(w8[0][0], w8[0][1], w8[1][0], w8[1][1], w8[2][0], w8[2][1], w8[3][0], w8[3][1]);
    if constexpr (!EARLY8) read_8(std::integral_constant<int, 0>{});
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        if constexpr (I + 1 < TM) read_8(std::integral_constant<int, I + 1>{});
        bf16x8 (&cur)[2] = (I & 1) ? ay : ax;
        lgkm_wait<(I + 1 < TM ? 2 : 0)>(cur[0], cur[1]);
        __builtin_amdgcn_sched_barrier(0);
        const i32x8 a8 = cat8(cur[0], cur[1]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[I][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, cat8(w8[j][0], w8[j][1]), acc[I][j], 0, 0, 0, e8m0(-kF8Act), 0, e8m0(-kF8Wgt - kF8Lo));
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, TM>{});
    if constexpr (PF) {
      load_w8();
#if AWT_GEMM_WDEC
      wait_vm<ORD::dma_wait_at_barrier()>();
#else
      wait_vm<ORD::w16_wait_at_barrier()>();
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (PF) {
      const int flip = (kt & 1) ? -STAGE : STAGE;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a16a[ks] += flip;
#pragma unroll
      for (int q = 0; q < 2; ++q) a8a[q] += flip;
    }
  };
  for (int kt = 0; kt + 1 < ktiles; ++kt) ktile(std::true_type{}, kt);
  ktile(std::false_type{}, ktiles - 1);

  // ---- epilogue: as gemm_f8_kernel (32-row x 64-column strips through a per-wave LDS patch, eight columns per lane), fed from the 16 x 16 C layout:
  // strip i = row tiles 2 i, 2 i + 1; tile (rt, ct) register reg sits at patch row rt 16 + 4 kq + reg, column ct 16 + r16
  constexpr int PITCH = 72;
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);
  const int c8 = (lane & 7) * 8, r8 = lane >> 3;
  const int em0 = m0, en = n0 + wc * 64 + c8;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (g.out.bias && en < g.out.n_valid) { b0 = *reinterpret_cast<const float4*>(g.out.bias + en); b1 = *reinterpret_cast<const float4*>(g.out.bias + en + 4); }
  constexpr bool SIDE = EPI == EPI_F32_RESID || EPI == EPI_F32_GELU_POS || EPI == EPI_BF16_DGELU;
  auto strips = [&](auto full_t) {
    constexpr bool FULL = decltype(full_t)::value;
    float4 side[4][2], side_next[4][2];
    auto load_sides = [&](float4 (&d)[4][2], int i) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        d[it][0] = load_side4<EPI, FULL>(g.out, em0 + i * 32 + r8 + 8 * it, en, g.M);
        d[it][1] = load_side4<EPI, FULL>(g.out, em0 + i * 32 + r8 + 8 * it, en + 4, g.M);
      }
    };
    if constexpr (SIDE) load_sides(side, 0);
#pragma unroll
    for (int i = 0; i < TM / 2; ++i) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < TN; ++ct)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) patch[(rt * 16 + 4 * kq + reg) * PITCH + ct * 16 + r16] = acc[2 * i + rt][ct][reg];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (SIDE) { if (i + 1 < TM / 2) load_sides(side_next, i + 1); }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rl = r8 + 8 * it;
        const float4 v0 = *reinterpret_cast<const float4*>(patch + rl * PITCH + c8), v1 = *reinterpret_cast<const float4*>(patch + rl * PITCH + c8 + 4);
        store_out8_f8<EPI, FULL>(g.out, em0 + i * 32 + rl, en, v0, v1, side[it][0], side[it][1], b0, b1, g.M);
      }
      if constexpr (SIDE) {
#pragma unroll
        for (int it = 0; it < 4; ++it) { side[it][0] = side_next[it][0]; side[it][1] = side_next[it][1]; }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  if (m0 + BM <= g.M && n0 + BN <= g.out.n_valid) strips(std::true_type{}); else strips(std::false_type{});
}

int g_mfma16 = 1;   // tuning knob "gemm_mfma16": 1 = the 16 x 16 form wherever its weight copies exist (default), 0 = the 32 x 32 kernels only
template <int EPI>
int launch_f8s(GemmArgs a, hipStream_t s) {
  constexpr int lds = 2 * (128 * 64 * 2 + 2 * 128 * 64);
  a.tiles_m = (a.M + 127) / 128;
  a.tiles_n = (a.N + 255) / 256;      // N % 256 != 0: the last column tile is partly empty (its weight rows exist, zero or unread garbage; its stores are masked by n_valid)
  a.group_n = 0; a.gm = g_gm;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f8s_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
  hipLaunchKernelGGL((gemm_f8s_kernel<EPI, false>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

template <int EPI, class CFG>
int launch_f8(GemmArgs a, hipStream_t s) {
  constexpr int BM = CFG::WM * CFG::TM * 32, BN = CFG::WN * CFG::TN * 32;
  constexpr int lds = 2 * (BM * 64 * 2 + 2 * BM * 64);     // two stages of A16 | A8 | Al8 = 64 KB (>= the epilogue patches)
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.group_n = 0; a.gm = g_gm;
  if (a.nseg == 1 && a.seg[0].w_exact16) {
    AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f8_kernel<EPI, CFG, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
    hipLaunchKernelGGL((gemm_f8_kernel<EPI, CFG, false, true>), dim3(a.tiles_m * a.tiles_n), dim3(CFG::WM * CFG::WN * 64), lds, s, a);
  } else if (a.nseg == 1) {
    AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f8_kernel<EPI, CFG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
    hipLaunchKernelGGL((gemm_f8_kernel<EPI, CFG, false>), dim3(a.tiles_m * a.tiles_n), dim3(CFG::WM * CFG::WN * 64), lds, s, a);
  } else {
    AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f8_kernel<EPI, CFG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
    hipLaunchKernelGGL((gemm_f8_kernel<EPI, CFG, true>), dim3(a.tiles_m * a.tiles_n), dim3(CFG::WM * CFG::WN * 64), lds, s, a);
  }
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}



// ================================================================================================ persistent ping-pong kernel (gemm_pp.h)
// 256 x 256 tiles, 8 waves, one workgroup per CU walking its tiles with one continuous K-tile stream; PREC_F16F8, one plain K segment.
// The epilogue is the one above (per-wave LDS transposition of 32 x 64 strips, eight columns per lane, whole row segments per store),
// with the patch in the wave's 8 KB of the last 64 KB of LDS (pitch 64 floats, 16-byte groups XOR-ed by the row's parity against
// bank conflicts on the transposed read) and its stores left in flight while the next tile's K loop starts.
template <int EPI, bool ILV>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(pp::Args g, GemmOut out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, kq = lane >> 4;
  float* patch = reinterpret_cast<float*>(smem + pp::PATCH_BASE + wave * 8192);
  const int c8 = (lane & 7) * 8, r8 = lane >> 3;
  constexpr bool SIDE = EPI == EPI_F32_RESID;
  pp::kloop<pp::FMT_F16F8S>(g, smem, [&](int tm, int tn, pp::Acc<pp::FMT_F16F8S>& accs) {
    auto& acc = accs.t;
    const int m0 = tm * pp::BM, n0 = tn * pp::BN;
    const int em0 = m0 + wr * 128, en = n0 + wc * 64 + c8;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (out.bias && en < out.n_valid) { b0 = *reinterpret_cast<const float4*>(out.bias + en); b1 = *reinterpret_cast<const float4*>(out.bias + en + 4); }
    auto strips = [&](auto full_t) {
      constexpr bool FULL = decltype(full_t)::value;
      float4 side[4][2], side_next[4][2];
      auto load_sides = [&](float4 (&d)[4][2], int i) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          d[it][0] = load_side4<EPI, FULL>(out, em0 + i * 32 + r8 + 8 * it, en, g.M);
          d[it][1] = load_side4<EPI, FULL>(out, em0 + i * 32 + r8 + 8 * it, en + 4, g.M);
        }
      };
      if constexpr (SIDE) load_sides(side, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // strip i = the 16-row tiles 2 i, 2 i + 1 (C layout: column lane & 15, row 4 (lane >> 4) + reg).  Patch position of (row, col): the column rotated by
        // 16 ((row >> 2) & 3) -- the four lane groups of one write then cover all 64 banks -- and XORed with 4 (row & 1) for the 16-byte reads below
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int row = rt * 16 + 4 * kq + reg;
              patch[row * 64 + (((ct * 16 + r16 + 16 * kq) & 63) ^ ((reg & 1) << 2))] = acc[2 * i + rt][ct][reg];
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (SIDE) { if (i + 1 < 4) load_sides(side_next, i + 1); }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rl = r8 + 8 * it, sw = (rl & 1) << 2, rot = 16 * ((rl >> 2) & 3);
          const float4 v0 = *reinterpret_cast<const float4*>(patch + rl * 64 + (((c8 + rot) & 63) ^ sw)), v1 = *reinterpret_cast<const float4*>(patch + rl * 64 + (((c8 + 4 + rot) & 63) ^ sw));
          store_out8_f8<EPI, FULL, ILV>(out, em0 + i * 32 + rl, en, v0, v1, side[it][0], side[it][1], b0, b1, g.M);
        }
        if constexpr (SIDE) {
#pragma unroll
          for (int it = 0; it < 4; ++it) { side[it][0] = side_next[it][0]; side[it][1] = side_next[it][1]; }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    };
    if (m0 + pp::BM <= g.M && n0 + pp::BN <= out.n_valid) strips(std::true_type{}); else strips(std::false_type{});
  });
}

template <int EPI, bool ILV>
int launch_pp(const pp::Args& a, const GemmOut& o, int grid, hipStream_t s) {
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_BYTES)));
  hipLaunchKernelGGL((gemm_pp_kernel<EPI, ILV>), dim3(grid), dim3(pp::NT), pp::LDS_BYTES, s, a, o);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

#ifdef AWT_EXPERIMENTAL_F6   // round-2 experiment: not compiled into the shipped library
// ================================================================================================ PREC_F16F6 (experimental)
// The f16f8 kernel with the two correction planes in FP6 e3m2: per fragment pair the fp16 product (4 x 32 pipe cycles per 64-deep K-tile and
// 32 x 32 tile) plus two block-scaled FP6 products of 32 cycles each = 1.5 bf16-MFMA-equivalents instead of 2, and 3.5 instead of 4 operand
// bytes per element.  128 x 256 on four waves, single K segment, fp32 output (the single-operator path `awt_op_linear`, precision 6); the encoder
// does not use it yet: its producers (LayerNorm, GELU epilogue, attention output) would have to emit 32-consecutive-k groups.
// LDS stage: A16 (16 KB) | A_hi6 (6 KB: 128 rows x 48 B) | A_lo6 (6 KB).  A lane (row r, half h) reads its 24 bytes as three ds_read_b64.
template <int OFF>
__device__ __forceinline__ i32x2 lds_read8a(unsigned addr) {
  i32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ i32x2 gload8(unsigned voff, const void* sbase) {
  i32x2 v;
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF));
  return v;
}
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f6_kernel(GemmArgs g) {
  constexpr int TM = 4, TN = 2, BM = 128, BN = 256, BK = 64, NT = 256;
  constexpr int PL16 = BM * BK * 2, PL6 = BM * 48, STAGE = PL16 + 2 * PL6;     // 16 + 6 + 6 KB
  constexpr int IT16 = PL16 / 16 / NT, IT6 = 2 * PL6 / 16 / NT;                 // 4 + 3 LDS-DMA pieces per thread
  constexpr int NDMA = IT16 + IT6;
  constexpr int NW16 = 4 * TN, NW6 = 2 * TN * 2;                                // W loads per lane per K-tile: fp16, e3m2 (2 planes x TN x (16 B + 8 B))
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = wave * 64 + lane;
  const int nwg = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qn = nwg >> 3, rn = nwg & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  const int GM = g.gm;
  int tm, tn;
  {
    const int grp = tile / (GM * g.tiles_n);
    const int gm = min(GM, g.tiles_m - grp * GM);
    const int within = tile - grp * GM * g.tiles_n;
    tn = within / gm; tm = grp * GM + (within - tn * gm);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int wc = wave;                       // 1 x 4 waves: wave w owns columns 64 w .. 64 w + 63 of the tile
  const int r32 = lane & 31, half = lane >> 5;
  const GemmSeg& sg = g.seg[0];
  const int ktiles = sg.K / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){};

  // ---- A streams: per-thread byte offsets of the DMA pieces (fixed), bases advanced per K-tile
  unsigned a16o[IT16], a6o[IT6];
#pragma unroll
  for (int it = 0; it < IT16; ++it) {
    const int p = it * NT + tid, row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
    int m = m0 + row; m = m < g.M ? m : g.M - 1;
    a16o[it] = (unsigned)(((int64_t)m * sg.lda + c * 8) * 2);
  }
  const int64_t pitch6 = sg.lda / 4 * 3;                          // bytes per row of an e3m2 plane
  const int64_t plane6 = (const char*)sg.al8 - (const char*)sg.a8;   // the lo6 plane follows the hi6 plane (checked on the host: < 4 GB)
#pragma unroll
  for (int it = 0; it < IT6; ++it) {
    const int q = it * NT + tid, pl = q >= 384, within = q - pl * 384, row = within / 3, c = within - row * 3;
    int m = m0 + row; m = m < g.M ? m : g.M - 1;
    a6o[it] = (unsigned)(pl * plane6 + (int64_t)m * pitch6 + c * 16);
  }
  const char* a16b = (const char*)sg.a_hi;
  const char* a6b = (const char*)sg.a8;
  const int nt0 = (n0 >> 5) + wc * TN;
  const int64_t w16_ts = (int64_t)sg.w_ksteps * 2 * 1024;           // bytes per 32-row n-tile of the fp16 plane
  const int64_t w6_ts = (int64_t)(sg.w_ksteps / 2) * 1536;           // bytes per n-tile of an e3m2 plane: (K / 64) blocks of 64 x 24 B
  const char *w16b[TN], *w6b[TN], *wl6b[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    w16b[j] = (const char*)sg.w_hi + (nt0 + j) * w16_ts;
    w6b[j] = (const char*)sg.w8 + (nt0 + j) * w6_ts;
    wl6b[j] = (const char*)sg.wl8 + (nt0 + j) * w6_ts;
  }
  const unsigned wl16 = lane * 16, wl24 = lane * 24;
  int kk = 0;
  auto advance = [&]() {
    if (kk + 1 < ktiles) {
      ++kk;
      a16b += BK * 2; a6b += 48;
#pragma unroll
      for (int j = 0; j < TN; ++j) { w16b[j] += 4 * 1024; w6b[j] += 1536; wl6b[j] += 1536; }
    }
  };
  auto dma = [&](auto op_t, char* stage) {
    constexpr int OP = decltype(op_t)::value;
    if constexpr (OP < IT16) glds16(a16b + a16o[OP], stage + (OP * NT + wave * 64) * 16);
    else glds16(a6b + a6o[OP - IT16], stage + PL16 + ((OP - IT16) * NT + wave * 64) * 16);
  };
  bf16x8 w16[4][TN];
  bf16x8 w6a[TN], wl6a[TN];          // first 16 bytes of the lane's 24
  i32x2 w6c[TN], wl6c[TN];           // last 8
  auto load_w16 = [&](auto ks_t) {
    constexpr int ks = decltype(ks_t)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) w16[ks][j] = gload16<ks * 1024>(wl16, w16b[j]);
  };
  auto load_w6 = [&]() {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      w6a[j] = gload16<0>(wl24, w6b[j]); w6c[j] = gload8<16>(wl24, w6b[j]);
      wl6a[j] = gload16<0>(wl24, wl6b[j]); wl6c[j] = gload8<16>(wl24, wl6b[j]);
    }
  };
  auto cat6 = [](bf16x8 lo, i32x2 hi) -> i32x8 {
    const i32x4_t l = __builtin_bit_cast(i32x4_t, lo);
    return (i32x8){l[0], l[1], l[2], l[3], hi[0], hi[1], 0, 0};
  };

  [&]<int... O>(std::integer_sequence<int, O...>) { (dma(std::integral_constant<int, O>{}, smem), ...); }(std::make_integer_sequence<int, NDMA>{});
  [&]<int... S>(std::integer_sequence<int, S...>) { (load_w16(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, 4>{});
  load_w6();
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(smem);
  unsigned a16a[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) a16a[ks] = lds0 + r32 * 128 + (((2 * ks + half) ^ ((r32 >> 1) & 7)) << 4);
  unsigned a6a = lds0 + PL16 + r32 * 48 + 24 * half;

  constexpr int DMA_PER_STEP = (NDMA + TM - 1) / TM;
  auto ktile = [&](auto pf_t, int kt) {
    constexpr bool PF = decltype(pf_t)::value;
    char* nxt = smem + ((kt + 1) & 1) * STAGE;
    if constexpr (PF) advance();
    constexpr int AD = 3;
    bf16x8 af[AD + 1];
    auto read_a16 = [&](auto s_t) {
      constexpr int S2 = decltype(s_t)::value;
      af[S2 % (AD + 1)] = lds_read16<(S2 % TM) * 32 * 128>(a16a[S2 / TM]);
    };
    [&]<int... S>(std::integer_sequence<int, S...>) { (read_a16(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, AD>{});
    [&]<int... S>(std::integer_sequence<int, S...>) {
      ([&] {
        constexpr int ks = S / TM, i = S % TM;
        if constexpr (S + AD < 4 * TM) read_a16(std::integral_constant<int, S + AD>{});
        if constexpr (PF && ks == 0) {
          [&]<int... O>(std::integer_sequence<int, O...>) {
            ([&] { constexpr int op = i * DMA_PER_STEP + O; if constexpr (op < NDMA) dma(std::integral_constant<int, op>{}, nxt); }(), ...);
          }(std::make_integer_sequence<int, DMA_PER_STEP>{});
        }
        constexpr int AHEAD = (4 * TM - 1 - S < AD ? 4 * TM - 1 - S : AD);
        lgkm_wait<AHEAD>(af[S % (AD + 1)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<true>(af[S % (AD + 1)], w16[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PF && i == TM - 1) load_w16(std::integral_constant<int, ks>{});
      }(), ...);
    }(std::make_integer_sequence<int, 4 * TM>{});
    // ---- e3m2 part: the six 8-byte reads of row tile i + 1 are issued before the four MFMAs of row tile i (two register sets), waits hand-counted
    if constexpr (PF) wait_vm<NDMA + NW16>(); else wait_vm<0>();
    i32x2 xy[2][6];
    auto read_xy = [&](auto i_t, i32x2 (&d)[6]) {
      constexpr int O = decltype(i_t)::value * 32 * 48;
      d[0] = lds_read8a<O>(a6a); d[1] = lds_read8a<O + 8>(a6a); d[2] = lds_read8a<O + 16>(a6a);
      d[3] = lds_read8a<O + PL6>(a6a); d[4] = lds_read8a<O + PL6 + 8>(a6a); d[5] = lds_read8a<O + PL6 + 16>(a6a);
    };
    read_xy(std::integral_constant<int, 0>{}, xy[0]);
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        i32x2 (&cur)[6] = xy[I & 1];
        if constexpr (I + 1 < TM) read_xy(std::integral_constant<int, I + 1>{}, xy[(I + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]) : "n"(I + 1 < TM ? 6 : 0));
        __builtin_amdgcn_sched_barrier(0);
        const i32x8 ax = {cur[0][0], cur[0][1], cur[1][0], cur[1][1], cur[2][0], cur[2][1], 0, 0};
        const i32x8 ay = {cur[3][0], cur[3][1], cur[4][0], cur[4][1], cur[5][0], cur[5][1], 0, 0};
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[I][j] = mfma32_f6<e8m0(-kF6Act), e8m0(-kF6Wgt - kF8Lo)>(ax, cat6(wl6a[j], wl6c[j]), acc[I][j]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[I][j] = mfma32_f6<e8m0(-kF6Act - kF8Lo), e8m0(-kF6Wgt)>(ay, cat6(w6a[j], w6c[j]), acc[I][j]);
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, TM>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PF) {
      load_w6();
      wait_vm<NW6>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (PF) {
      const int flip = (kt & 1) ? -STAGE : STAGE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a16a[ks] += flip;
      a6a += flip;
    }
  };
  for (int kt = 0; kt + 1 < ktiles; ++kt) ktile(std::true_type{}, kt);
  ktile(std::false_type{}, ktiles - 1);

  // ---- epilogue: as gemm_f8_kernel (predicated form)
  constexpr int PITCH = 72;
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);
  const int c8 = (lane & 7) * 8, r8 = lane >> 3;
  const int em0 = m0, en = n0 + wc * 64 + c8;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (g.out.bias && en < g.out.n_valid) { b0 = *reinterpret_cast<const float4*>(g.out.bias + en); b1 = *reinterpret_cast<const float4*>(g.out.bias + en + 4); }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) patch[((rr & 3) + 8 * (rr >> 2) + 4 * half) * PITCH + j * 32 + r32] = acc[i][j][rr];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float4 side[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      side[it][0] = load_side4<EPI>(g.out, em0 + i * 32 + r8 + 8 * it, en, g.M);
      side[it][1] = load_side4<EPI>(g.out, em0 + i * 32 + r8 + 8 * it, en + 4, g.M);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = r8 + 8 * it;
      const float4 v0 = *reinterpret_cast<const float4*>(patch + rl * PITCH + c8), v1 = *reinterpret_cast<const float4*>(patch + rl * PITCH + c8 + 4);
      store_out8_f8<EPI, false>(g.out, em0 + i * 32 + rl, en, v0, v1, side[it][0], side[it][1], b0, b1, g.M);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int EPI>
int launch_f6(GemmArgs a, hipStream_t s) {
  constexpr int lds = 2 * (128 * 64 * 2 + 2 * 128 * 48);     // 56 KB (>= the 36 KB of epilogue patches)
  a.tiles_m = (a.M + 127) / 128;
  a.tiles_n = a.N / 256;
  a.group_n = 0; a.gm = g_gm;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f6_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
  hipLaunchKernelGGL((gemm_f6_kernel<EPI>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

#endif  // AWT_EXPERIMENTAL_F6

int g_force_tile = 0;  // 0 = auto, 64 / 128 / 256 = forced (tuning and tests)
// Tile order (see the kernel).  Measured on the encoder's shapes (tools/gemm_traffic_shapes.sh, profiles/r01_gemm_tile_order.txt):
// column-tile groups of 3 cut the L2 -> fabric reads by 10 - 25 % but run 1.5 % slower end to end than groups of GM = 4 row
// panels, so row-panel groups are the default; AWT_GEMM_GROUP_N=n selects column groups for experiments.
int g_group_n = 0;

// batched launch (launch_gemm_batched): `batch` independent matrices of one shape, blockIdx.y = matrix
template <int TERMS, int BK, int EPI, class CFG>
int launch_batched_one(GemmArgs a, int batch, hipStream_t s) {
  using T = Tile<TERMS, BK, CFG, false>;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<TERMS, BK, EPI, CFG, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)));
  a.tiles_m = (a.M + T::BM - 1) / T::BM;
  a.tiles_n = (a.N + T::BN - 1) / T::BN;
  a.group_n = 0; a.gm = g_gm;
  hipLaunchKernelGGL((gemm_kernel<TERMS, BK, EPI, CFG, false, false, true>), dim3(a.tiles_m * a.tiles_n, batch), dim3(T::THREADS), T::LDS_BYTES, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

template <int TERMS, int BK, int EPI, class CFG, bool F16 = false, bool WX = false>
int launch_one(GemmArgs a, hipStream_t s) {
  using T = Tile<TERMS, BK, CFG, WX>;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<TERMS, BK, EPI, CFG, F16, WX>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)));
  a.tiles_m = (a.M + T::BM - 1) / T::BM;
  a.tiles_n = (a.N + T::BN - 1) / T::BN;
  a.group_n = g_group_n; a.gm = g_gm;
  hipLaunchKernelGGL((gemm_kernel<TERMS, BK, EPI, CFG, F16, WX>), dim3(a.tiles_m * a.tiles_n), dim3(T::THREADS), T::LDS_BYTES, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}


// Tile choice by tile count: a launch should put at least one tile on each of the 512 workgroup slots (256 CUs x 2) --
// with fewer, its duration is one tile's K loop however small M is, so smaller tiles (shorter K-tile steps) win.
//   128 x 256 (4 waves of 128 x 64): the encoder's shapes from ~3 clips up;  128 x 128: N not a multiple of 256 (LoRA
//   projections, N = 384 models) and mid-sized M;  64 x 128: one or two clips (M = 1500 .. 3000) and tiny test shapes.
constexpr int kSlots = 512;
template <int EPI>
int launch_epi(GemmArgs a, int prec, hipStream_t s) {
  if (prec == PREC_F16F6) {
#ifndef AWT_EXPERIMENTAL_F6
    return awt_fail(AWT_ERR_INVALID, "gemm (f16f6): the FP6 cross-term experiment is not part of this build (compile with -DAWT_EXPERIMENTAL_F6)");
#else
    if constexpr (EPI == EPI_F32 || EPI == EPI_F32_RESID) {
      if (a.nseg != 1 || a.N % 256 != 0 || a.seg[0].rows_out != a.M || a.seg[0].rows_in != a.M || a.seg[0].row_mul != 1 || a.seg[0].row_add != 0)
        return awt_fail(AWT_ERR_INVALID, "gemm (f16f6): one plain K segment and N % 256 == 0 only");
      const int64_t gap = (const char*)a.seg[0].al8 - (const char*)a.seg[0].a8;
      if (gap <= 0 || gap >= (1ll << 31)) return awt_fail(AWT_ERR_INVALID, "gemm (f16f6): the lo6 plane must follow the hi6 plane within 2 GB");
      return launch_f6<EPI>(a, s);
    } else return awt_fail(AWT_ERR_INVALID, "gemm (f16f6): fp32 outputs only (experimental single-operator path)");
#endif
  }
  if (prec == PREC_F16F8) {
    {
      // the 16 x 16 MFMA form: one segment that carries its 16-row weight copies and takes no fp16-exact shortcut (multi-segment K loops keep the 32 x 32
      // kernel: their per-segment address state does not fit beside the 16 x 16 form's fragment rings).  It also takes N % 256 != 0 (Whisper-tiny: 384, 1152)
      // when the copies are allocated for whole column tiles and at most a quarter of the last tile's work is padding: measured on M = 48000 (tools/tiny_shapes_probe.py)
      // 136 -> 126 us at N = 1152, 186 -> 170 us at N = 384 / K = 1536 against the 128 x 128 tiles these shapes fell to
      const int n256 = (a.N + 255) / 256 * 256;
      const bool f8s_ok = g_mfma16 && a.nseg == 1 && a.seg[0].ws16 && a.seg[0].ws8 && a.seg[0].a8 && !a.seg[0].w_exact16 &&
                          (a.N % 256 == 0 || (a.seg[0].ws_rows >= n256 && (int64_t)n256 * 3 <= (int64_t)a.N * 4));
      const int64_t t256f = (int64_t)((a.M + 127) / 128) * (f8s_ok ? n256 / 256 : a.N / 256);
      int tile = g_force_tile;
      if (!tile) tile = ((a.N % 256 == 0 || f8s_ok) && t256f >= kSlots) ? 256 : 128;
      if (tile == 512 && a.N % 256 == 0 && a.nseg == 1) return launch_f8<EPI, CfgF8Big>(a, s);
      if (tile == 256 && f8s_ok) return launch_f8s<EPI>(a, s);
      if (tile >= 256 && a.N % 256 == 0) return launch_f8<EPI, CfgF8W4>(a, s);
      return launch_f8<EPI, CfgF8Sq>(a, s);
    }
  }
  const int terms = prec_products(prec);
  if (prec == PREC_F16X3) {      // fp16 hi / lo planes on the same kernels
    const int64_t t256h = (int64_t)((a.M + 127) / 128) * (a.N / 256), t128h = (int64_t)((a.M + 127) / 128) * (a.N / 128);
    int tile = g_force_tile;
    if (!tile) tile = (a.N % 256 == 0 && t256h >= kSlots) ? 256 : (t128h >= kSlots ? 128 : 64);
    if (tile == 256 && a.N % 256 != 0) tile = 128;
    if (a.nseg == 1 && a.seg[0].w_exact16) {     // fp16-exact weights: two products per fragment pair
      if (tile == 256) return launch_one<3, 32, EPI, CfgW4, true, true>(a, s);
      if (tile == 128) return launch_one<3, 32, EPI, Cfg128, true, true>(a, s);
      return launch_one<3, 64, EPI, Cfg64, true, true>(a, s);
    }
    if (tile == 256) return launch_one<3, 32, EPI, CfgW4, true>(a, s);
    if (tile == 128) return launch_one<3, 32, EPI, Cfg128, true>(a, s);
    return launch_one<3, 64, EPI, Cfg64, true>(a, s);
  }
  const int64_t t256 = (int64_t)((a.M + 127) / 128) * (a.N / 256), t128 = (int64_t)((a.M + 127) / 128) * (a.N / 128);
  int tile = g_force_tile;
  if (!tile) tile = (a.N % 256 == 0 && t256 >= kSlots) ? 256 : (t128 >= kSlots ? 128 : 64);
  if (tile == 256 && a.N % 256 != 0) tile = 128;
  if (prec == PREC_F16) {        // one fp16 product (measurement mode): the single-product kernels on fp16 planes
    if (tile == 256) return launch_one<1, 64, EPI, CfgW4, true>(a, s);
    if (tile == 128) return launch_one<1, 64, EPI, Cfg128, true>(a, s);
    return launch_one<1, 64, EPI, Cfg64, true>(a, s);
  }
  if (tile == 256) return terms == 3 ? launch_one<3, 32, EPI, CfgW4>(a, s) : launch_one<1, 64, EPI, CfgW4>(a, s);
  if (tile == 128) return terms == 3 ? launch_one<3, 32, EPI, Cfg128>(a, s) : launch_one<1, 64, EPI, Cfg128>(a, s);
  // few tiles per CU: the K loop is latency-bound (one barrier + one global round trip per K-tile), so the deeper 64-wide
  // K-tile is used for the split product as well (its 64 + 64 weight-fragment registers fit next to 32 accumulators)
  return terms == 3 ? launch_one<3, 64, EPI, Cfg64>(a, s) : launch_one<1, 64, EPI, Cfg64>(a, s);
}


template <int EPI>
int launch_batched_epi(GemmArgs a, int batch, hipStream_t s) {
  // tile by the rows a matrix wastes: 64-row tiles when the last 128-row tile would be at most half full or the launch is small
  const int64_t t256 = (int64_t)((a.M + 127) / 128) * (a.N / 256) * batch, t128 = (int64_t)((a.M + 127) / 128) * (a.N / 128) * batch;
  const int tail = a.M % 128;
  int tile = g_force_tile;
  if (!tile) tile = (tail > 0 && tail <= 64 && a.M < 1024) ? 64 : (a.N % 256 == 0 && t256 >= kSlots) ? 256 : (t128 >= kSlots ? 128 : 64);
  if (tile == 256 && a.N % 256 != 0) tile = 128;
  if (tile == 256) return launch_batched_one<3, 32, EPI, CfgW4>(a, batch, s);
  if (tile == 128) return launch_batched_one<3, 32, EPI, Cfg128>(a, batch, s);
  return launch_batched_one<3, 64, EPI, Cfg64>(a, batch, s);
}

}  // namespace

// C_z [M, N] = A_z [M, K] . W_z [N, K]^T (+ resid_z) for z < batch: split-bf16 operands, fp32 output.  A: row-major hi / lo planes, matrix z at
// element offset z * bs_a; W: fragment-major hi / lo planes of [N, K] (w_frag_index), matrix z at z * bs_w; out / resid: z * bs_o.
int launch_gemm_batched(awt_ctx* c, int batch, int M, int N, int K, const bf16_t* a_hi, const bf16_t* a_lo, int64_t lda, int64_t bs_a, const bf16_t* w_hi,
                        const bf16_t* w_lo, int64_t bs_w, float* out, const float* resid, int64_t ldo, int64_t bs_o, int n_valid, hipStream_t s) {
  AWT_REQUIRE(c && c->zeros && batch > 0 && batch <= 65535 && M > 0 && N > 0 && N % 128 == 0 && K > 0 && K % 64 == 0, AWT_ERR_INVALID,
              "gemm (batched): 1..65535 matrices, N % 128 == 0 and K % 64 == 0 required");
  AWT_REQUIRE(a_hi && a_lo && w_hi && w_lo && out && lda % 8 == 0 && lda >= K && ldo % 4 == 0 && n_valid > 0 && n_valid <= N && n_valid % 4 == 0 && ldo >= n_valid, AWT_ERR_INVALID,
              "gemm (batched): null plane, lda not a multiple of 8, or output columns / pitch not a multiple of 4");
  AWT_REQUIRE(bs_a % 8 == 0 && bs_w % 8 == 0 && bs_o % 4 == 0, AWT_ERR_INVALID, "gemm (batched): matrix strides must keep 16-byte alignment");
  GemmArgs a{};
  a.M = M; a.N = N; a.nseg = 1; a.zeros = (const bf16_t*)c->zeros;
  GemmSeg& sg = a.seg[0];
  sg.a_hi = a_hi; sg.a_lo = a_lo; sg.lda = lda; sg.w_hi = w_hi; sg.w_lo = w_lo; sg.w_ksteps = K / 32; sg.w_k0 = 0; sg.K = K;
  sg.rows_out = sg.rows_in = M; sg.row_mul = 1; sg.row_add = 0;
  a.out.f32 = out; a.out.resid = resid; a.out.ldo = ldo; a.out.n_valid = n_valid;
  a.bs_a = bs_a; a.bs_w = bs_w; a.bs_o = bs_o;
  ProfScope prof(c, AWT_PROF_GEMM, s, 2.0 * (double)batch * (double)M * (double)n_valid * (double)K);
  return resid ? launch_batched_epi<EPI_F32_RESID>(a, batch, s) : launch_batched_epi<EPI_F32>(a, batch, s);
}

// tuning knob "gemm_pp" (include/awt.h): 0 = off, 1 = automatic (default: large inference launches on weights that are not fp16-exact, for the projections in
// awt_api.hip's gemm_pp_mask), 2 = wherever supported.  Measured on the headline step (profiles/r04_gemm_pp16_masks.txt, A/B interleaved in one process): the MLP
// pair on this kernel 46.98 ms per step against 47.50 on the 128 x 256 kernel, QKV + out_proj neutral.
int g_pp_mode = 1;
int g_pp_stagger = 0;   // tuning knob "gemm_pp_stagger": start-up de-phasing of the persistent workgroups (gemm_pp.h Args::stagger), 0 = off
void awt_gemm_set_pp_stagger(int v) { g_pp_stagger = v; }
int awt_gemm_pp_mode() { return g_pp_mode; }
void awt_gemm_set_pp_mode(int v) { g_pp_mode = v; }
// persistent workgroups of a ping-pong launch = CUs of the current device (0 on error)
int gemm_pp_slots() {
  static int n_cu[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (!n_cu[dev & 63]) { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0; n_cu[dev & 63] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }
  return n_cu[dev & 63];
}
bool gemm_pp_supported(int M, int N, int K, int epi) {
  return M > 0 && N > 0 && N % 256 == 0 && K >= 128 && K % 64 == 0 &&
         (epi == EPI_F32 || epi == EPI_F32_RESID || epi == EPI_BF16 || epi == EPI_BF16_GELU || epi == EPI_QKV);
}
int launch_gemm_pp(awt_ctx* c, int M, int N, const GemmSeg& seg, GemmEpilogue epi, const GemmOut& out, hipStream_t s) {
  AWT_REQUIRE(c && gemm_pp_supported(M, N, seg.K, epi), AWT_ERR_INVALID, "gemm (ping-pong): N % 256 == 0, K % 64 == 0, K >= 128 and a supported epilogue required");
  AWT_REQUIRE(seg.a_ilv && seg.w_pp && seg.rows_out == M && seg.rows_in == M && seg.row_mul == 1 && seg.row_add == 0 && seg.w_k0 == 0 && seg.w_ksteps == seg.K / 32 && seg.lda == seg.K,
              AWT_ERR_INVALID, "gemm (ping-pong): one plain, dense K segment over split-line activations and a packed weight image");
  AWT_REQUIRE(!out.ilv || ((epi == EPI_BF16 || epi == EPI_BF16_GELU) && out.ldo == N && out.n_valid == N), AWT_ERR_INVALID, "gemm (ping-pong): a split-line output is a dense [M, N] activation");
  pp::Args a{};
  a.A = seg.a_ilv; a.a_row_bytes = (int64_t)seg.lda * 4; a.W = seg.w_pp;
  a.M = M; a.N = N; a.K = seg.K; a.nk = seg.K / 32;
  a.tiles_m = (M + pp::BM - 1) / pp::BM; a.tiles_n = N / pp::BN; a.ntiles = a.tiles_m * a.tiles_n; a.gm = g_gm; a.stagger = g_pp_stagger;
  const int grid = std::min(a.ntiles, gemm_pp_slots());          // one persistent workgroup per CU (all of its LDS)
  ProfScope prof(c, AWT_PROF_GEMM, s, 2.0 * (double)M * (double)out.n_valid * (double)seg.K);
  switch (epi) {
    case EPI_F32: return launch_pp<EPI_F32, false>(a, out, grid, s);
    case EPI_F32_RESID: return launch_pp<EPI_F32_RESID, false>(a, out, grid, s);
    case EPI_BF16: return out.ilv ? launch_pp<EPI_BF16, true>(a, out, grid, s) : launch_pp<EPI_BF16, false>(a, out, grid, s);
    case EPI_BF16_GELU: return out.ilv ? launch_pp<EPI_BF16_GELU, true>(a, out, grid, s) : launch_pp<EPI_BF16_GELU, false>(a, out, grid, s);
    case EPI_QKV: return launch_pp<EPI_QKV, false>(a, out, grid, s);
    default: break;
  }
  return awt_fail(AWT_ERR_INVALID, "gemm (ping-pong): unsupported epilogue");
}

void awt_gemm_set_mfma16(int v) { g_mfma16 = v; }
void awt_gemm_force_tile(int t) { g_force_tile = t; }
void awt_gemm_set_gm(int v) { g_gm = v > 0 ? v : AWT_GEMM_GM; }

int launch_gemm(awt_ctx* c, int M, int N, const GemmSeg* segs, int nseg, int prec, GemmEpilogue epi, const GemmOut& out,
                hipStream_t s) {
  AWT_REQUIRE(M > 0 && N > 0 && N % 128 == 0, AWT_ERR_INVALID, "gemm: N must be a positive multiple of 128");
  AWT_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, AWT_ERR_INVALID, "gemm: 1..3 K-segments");
  AWT_REQUIRE(prec == PREC_BF16 || prec == PREC_F16 || prec == PREC_BF16X3 || prec == PREC_F16X3 || prec == PREC_F16F8 || prec == PREC_F16F6, AWT_ERR_INVALID, "gemm: unknown operand precision");
  const int terms = prec_products(prec);
  AWT_REQUIRE(c && c->zeros, AWT_ERR_INVALID, "gemm: context without a zero page");
  static const bool env_read = [] { if (const char* e = getenv("AWT_GEMM_GROUP_N")) g_group_n = std::max(0, atoi(e)); return true; }();   // tile-order experiments (tools/)
  (void)env_read;
  GemmArgs a{};
  a.M = M; a.N = N; a.nseg = nseg; a.out = out; a.zeros = (const bf16_t*)c->zeros;
  double ksum = 0;
  for (int i = 0; i < nseg; ++i) {
    a.seg[i] = segs[i];
    AWT_REQUIRE(segs[i].K > 0 && segs[i].K % 64 == 0, AWT_ERR_INVALID, "gemm: every K-segment must be a positive multiple of 64");
    if (prec == PREC_F16F8 || prec == PREC_F16F6) AWT_REQUIRE(segs[i].a_hi && segs[i].w_hi && (segs[i].a8 || (segs[i].w_exact16 && nseg == 1 && prec == PREC_F16F8)) && segs[i].al8 && segs[i].w8 && segs[i].wl8 && segs[i].w_ksteps % 2 == 0 && segs[i].w_k0 % 2 == 0,
                                        AWT_ERR_INVALID, "gemm (f16f8): null operand plane or a K-segment that is not 64-aligned in its weight matrix");
    else AWT_REQUIRE(segs[i].a_hi && segs[i].w_hi && (terms == 1 || (segs[i].a_lo && segs[i].w_lo)), AWT_ERR_INVALID, "gemm: null operand plane");
    AWT_REQUIRE(segs[i].lda % 8 == 0 && segs[i].w_ksteps > 0 && segs[i].w_k0 >= 0 && segs[i].w_k0 + segs[i].K / 32 <= segs[i].w_ksteps, AWT_ERR_INVALID,
                "gemm: lda must be a multiple of 8 and the segment must lie inside its fragment-major weight matrix");
    AWT_REQUIRE(segs[i].rows_out > 0 && segs[i].rows_in > 0, AWT_ERR_INVALID, "gemm: bad row map");
    ksum += segs[i].K;
  }
  ProfScope prof(c, AWT_PROF_GEMM, s, 2.0 * (double)M * (double)out.n_valid * ksum);
  switch (epi) {
    case EPI_F32: return launch_epi<EPI_F32>(a, prec, s);
    case EPI_F32_RESID: return launch_epi<EPI_F32_RESID>(a, prec, s);
    case EPI_BF16: return launch_epi<EPI_BF16>(a, prec, s);
    case EPI_BF16_GELU: return launch_epi<EPI_BF16_GELU>(a, prec, s);
    case EPI_QKV: return launch_epi<EPI_QKV>(a, prec, s);
    case EPI_F32_GELU_POS: return launch_epi<EPI_F32_GELU_POS>(a, prec, s);
    case EPI_BF16_GELU_SAVE: return launch_epi<EPI_BF16_GELU_SAVE>(a, prec, s);
    case EPI_BF16_DGELU: return launch_epi<EPI_BF16_DGELU>(a, prec, s);
  }
  return awt_fail(AWT_ERR_INVALID, "gemm: unknown epilogue");
}
