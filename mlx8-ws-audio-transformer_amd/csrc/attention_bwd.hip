// Flash-style attention backward (part of K14) for head_dim 64, no mask -- gfx950.
//
// Given q (pre-scaled by head_dim^-1/2 and by log2(e), as attention.hip stores it), k, v, the forward's log-sum-exp (log2 domain), dO and delta = rowsum(dO * O):
//   P = exp2(s log2e - lse2),   dV = P^T dO,   dP = dO V^T,   dS = P (dP - delta),   dQ = dS K,   dK = dS^T Q.
// P is recomputed from the saved row statistics instead of storing the S x S scores.
//
// Two launches of one kernel template, neither needs a cross-workgroup sum (deterministic, no atomics):
//   MODE_DQ   lanes = queries (32 per wave), K and V tiles stream through LDS:   dQ^T += K^T dS^T
//   MODE_DKV  lanes = keys    (32 per wave), Q and dO tiles stream through LDS:  dK^T += Q^T dS,  dV^T += dO^T P
// Both follow the forward kernel's shape (attention.hip): the first products are "tile rows x lane-resident
// fragments" on v_mfma_f32_32x32x16_bf16, their 32x32 accumulators (reduction index in registers, lane index on the
// lane) become the B operands of the second products, whose A operands are ds_read_b64_tr_b16 transposes of the same
// LDS tiles (one chunk-XOR-swizzled image serves both the row reads and the transposed reads).
// TERMS = 3: every operand is a hi + lo bf16 pair, three MFMAs per fragment pair (DESIGN.md "Numerics").
// Gradients are written as bf16 hi/lo planes in ROW-MAJOR [B*S, 3*d] (dq | dk | dv, dq already multiplied by the
// head_dim^-1/2 the forward applied to q), which is the K-contiguous A operand of the QKV backward GEMM.
#include <type_traits>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int LW = 32;            // lane-resident rows per wave
constexpr int LB = 128;           // ... per workgroup
constexpr int TB = 64;            // streamed rows per tile
constexpr int PLANE = TB * 64 * 2;

enum { MODE_DQ = 0, MODE_DKV = 1 };

struct BwdArgs {
  const bf16_t *q_hi, *q_lo, *k_hi, *k_lo, *v_hi, *v_lo;   // head-major [B, H, S, 64]
  const bf16_t *do_hi, *do_lo;                              // row-major [B*S, H*64]
  const float* lse2;                                        // [B, H, S]
  float* delta;                                             // [B, H, S]: written by the dq launch (rowsum(dO * O), from o_hi / o_lo), read by the dk / dv launch
  const bf16_t *o_hi, *o_lo;                                // the forward's attention output, row-major like dO
  bf16_t *g_hi, *g_lo;                                      // row-major [B*S, 3*H*64]: dq | dk | dv
  int B, H, S; float qscale;
};

// Chunk swizzle of a [rows][64 x 16-bit] LDS image (128-byte rows, eight 16-byte chunks) that serves BOTH kinds of read: chunk ^ swz(row) with swz = the three
// bits of row >> 1 with bits 0 and 2 exchanged.  ds_read_b128 of a 32x32x16 operand (16 distinct rows per 16-lane group, one chunk column) needs swz to be a
// bijection of (row >> 1) & 7 over those rows -- any bit permutation is; ds_read_b64_tr_b16 (a 32-lane half takes four consecutive rows x four chunks) needs rows
// r and r + 2 (256 bytes apart: the same banks) in different 64-byte halves of their rows, i.e. bit 2 of the swizzle must follow bit 0 of row >> 1.  With the
// plain (row >> 1) & 7 used until round 3 every transposed read was a 2-way bank conflict (SQ_LDS_BANK_CONFLICT = 24 % of the f16f8 kernel's LDS-active cycles).
// Consequences for the offsets below: rows + 8 flip bit 2 of row >> 1 = bit 0 of the swizzle (byte ^ 16), rows + 16 / + 32 change nothing.
__device__ __forceinline__ int swz(int row) { const int x = (row >> 1) & 7; return (x & 2) | ((x & 1) << 2) | (x >> 2); }
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) { glds16_asm(gsrc, lds_wave_base); }   // common.h: invisible to hipcc's vmcnt bookkeeping
__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

struct TileSrc { const bf16_t *hi, *lo; int64_t base; int64_t row_stride; };   // element offsets

// stage two 64-row tiles (T1, T2), hi (+ lo) planes, chunk-swizzled; rows beyond S are clamped (masked later)
template <int TERMS>
__device__ __forceinline__ void stage_tiles(const TileSrc& t1, const TileSrc& t2, int S, int tt, char* stage, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int p = it * kThreads + wave * 64 + lane;
    const int row = p >> 3;
    const int c = (p & 7) ^ swz(row);
    int r = tt * TB + row; r = r < S ? r : S - 1;
    char* dst = stage + (it * kThreads + wave * 64) * 16;
    const int64_t o1 = t1.base + (int64_t)r * t1.row_stride + c * 8;
    const int64_t o2 = t2.base + (int64_t)r * t2.row_stride + c * 8;
    glds16(t1.hi + o1, dst);
    if (TERMS == 3) glds16(t1.lo + o1, dst + PLANE);
    glds16(t2.hi + o2, dst + (TERMS == 3 ? 2 : 1) * PLANE);
    if (TERMS == 3) glds16(t2.lo + o2, dst + 3 * PLANE);
  }
}

// acc (32 x 32, rows = tile rows in registers, cols = lanes) += T[rows] . L^T   with L fragments lane-resident
template <int TERMS>
__device__ __forceinline__ void rows_times_lane(f32x16& acc, const char* t_hi, const char* t_lo, const int (&roff)[4], int sub,
                                                const bf16x8 (&lh)[4], const bf16x8 (&ll)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int off = roff[ks] + sub * 4096;   // the swizzle term repeats every 16 rows: the second sub-tile is a constant away
    const bf16x8 th = *reinterpret_cast<const bf16x8*>(t_hi + off);
    if (TERMS == 3) {
      const bf16x8 tl = *reinterpret_cast<const bf16x8*>(t_lo + off);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, ll[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl, lh[ks], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, lh[ks], acc, 0, 0, 0);
  }
}

// out^T (64 dims x 32 lanes, two 32-dim tiles) += T^T (transposed read of a 32-row sub-tile) . X, where X is a 32 x 32
// accumulator tile whose registers 8 s .. 8 s + 7 form the B fragment of k-step s (cdna_hip_programming.md §3)
template <int TERMS>
__device__ __forceinline__ void trT_times_acc(f32x16 (&out)[2], const char* t_hi, const char* t_lo, int sub, const f32x16& x, int toff, int toffx) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 xh, xl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16_t hi, lo;
      split_bf16(x[8 * s + j], hi, lo);
      xh[j] = (short)hi;
      if (TERMS == 3) xl[j] = (short)lo;
    }
#pragma unroll
    for (int et = 0; et < 2; ++et) {
      // block offsets as in attention.hip: dim-tile et toggles byte bit 6, the +8-row block adds 1024 and flips byte bit 4
      const int cst = sub * 4096 + s * 2048;
      const int off0 = (et == 0 ? toff : toffx) + cst;
      const int off1 = (off0 ^ 16) + 1024;              // rows + 8: swizzle bit 0 flips (see swz)
      const bf16x4 va = tr_read(t_hi + off0), vb = tr_read(t_hi + off1);
      const bf16x8 th = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      if (TERMS == 3) {
        const bf16x4 la = tr_read(t_lo + off0), lb = tr_read(t_lo + off1);
        const bf16x8 tl = {la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
        out[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, xl, out[et], 0, 0, 0);
        out[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl, xh, out[et], 0, 0, 0);
      }
      out[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, xh, out[et], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void load_lane_frags(const bf16_t* hi, const bf16_t* lo, int64_t off, bool want_lo, bf16x8 (&fh)[4], bf16x8 (&fl)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fh[ks] = *reinterpret_cast<const bf16x8*>(hi + off + ks * 16);
    if (want_lo) fl[ks] = *reinterpret_cast<const bf16x8*>(lo + off + ks * 16);
  }
}

__device__ __forceinline__ void store_grad(const BwdArgs& a, const f32x16 (&acc)[2], int64_t row_off, int half, float scale) {
#pragma unroll
  for (int et = 0; et < 2; ++et)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int e = 32 * et + 8 * g4 + 4 * half;
      bf16_t hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf16(acc[et][4 * g4 + j] * scale, hi[j], lo[j]);
      *reinterpret_cast<uint2*>(a.g_hi + row_off + e) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
      if (a.g_lo) *reinterpret_cast<uint2*>(a.g_lo + row_off + e) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
    }
}

template <int TERMS, int MODE, int GT>
__global__ __launch_bounds__(kThreads, 2) void attention_bwd_kernel(BwdArgs a) {
  constexpr int NPL = TERMS == 3 ? 4 : 2;
  constexpr int STAGE = NPL * PLANE + 512;        // + [64] lse2 and [64] delta of the streamed rows (MODE_DKV)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware 1-D grid (see attention.hip): the blocks of one head run on one XCD and share its L2 copy of the tiles
  const int nlb = (a.S + LB - 1) / LB;
  const int nwg = nlb * a.B * a.H;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int bh = logical / nlb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int l0 = (logical - bh * nlb) * LB + wave * LW;
  const int ll_ = lane & 31, half = lane >> 5;
  const int d = a.H * 64;
  const int64_t head_off = (int64_t)bh * a.S * 64;            // head-major planes
  const int64_t rm_off = (int64_t)b * a.S * d + h * 64;        // row-major dO: + row * d

  int lrow = l0 + ll_; lrow = lrow < a.S ? lrow : a.S - 1;
  bf16x8 l1h[4], l1l[4], l2h[4], l2l[4];
  TileSrc t1, t2;
  float lse_l = 0.f, delta_l = 0.f;
  if (MODE == MODE_DQ) {   // lanes: queries.  L1 = q, L2 = dO; tiles: K, V
    load_lane_frags(a.q_hi, a.q_lo, head_off + (int64_t)lrow * 64 + half * 8, TERMS == 3, l1h, l1l);
    load_lane_frags(a.do_hi, a.do_lo, rm_off + (int64_t)lrow * d + half * 8, TERMS == 3, l2h, l2l);
    t1 = TileSrc{a.k_hi, a.k_lo, head_off, 64};
    t2 = TileSrc{a.v_hi, a.v_lo, head_off, 64};
    lse_l = a.lse2[(int64_t)bh * a.S + lrow];
    {   // delta = rowsum(dO * O) of the lane's query row: each lane half holds 32 of its 64 elements (the fragments just loaded); the dk / dv launch reads it back
      bf16x8 oh[4], ol[4];
      load_lane_frags(a.o_hi, a.o_lo, rm_off + (int64_t)lrow * d + half * 8, TERMS == 3, oh, ol);
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dv = bf16_to_f32((bf16_t)l2h[ks][j]) + (TERMS == 3 ? bf16_to_f32((bf16_t)l2l[ks][j]) : 0.f);
          const float ov = bf16_to_f32((bf16_t)oh[ks][j]) + (TERMS == 3 ? bf16_to_f32((bf16_t)ol[ks][j]) : 0.f);
          part += dv * ov;
        }
      delta_l = part + __shfl_xor(part, 32);
      if (half == 0 && l0 + ll_ < a.S) a.delta[(int64_t)bh * a.S + lrow] = delta_l;
    }
  } else {                 // lanes: keys.  L1 = k, L2 = v; tiles: Q, dO
    load_lane_frags(a.k_hi, a.k_lo, head_off + (int64_t)lrow * 64 + half * 8, TERMS == 3, l1h, l1l);
    load_lane_frags(a.v_hi, a.v_lo, head_off + (int64_t)lrow * 64 + half * 8, TERMS == 3, l2h, l2l);
    t1 = TileSrc{a.q_hi, a.q_lo, head_off, 64};
    t2 = TileSrc{a.do_hi, a.do_lo, rm_off, d};
  }

  f32x16 g1[2], g2[2];   // MODE_DQ: g1 = dQ^T.  MODE_DKV: g1 = dK^T, g2 = dV^T
  g1[0] = (f32x16){}; g1[1] = (f32x16){}; g2[0] = (f32x16){}; g2[1] = (f32x16){};

  int roff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) roff[ks] = ll_ * 128 + (((2 * ks + half) ^ swz(ll_)) << 4);
  const int tg = lane >> 4, tli = lane & 15, tqq = tli >> 2, tpp = tli & 3;
  const int trow = 4 * (tg >> 1) + tqq;
  const int toff = trow * 128 + (((2 * (tg & 1) + (tpp >> 1)) ^ swz(trow)) << 4) + 8 * (tpp & 1);
  const int toffx = toff ^ 64;

  const int ntiles = (a.S + TB - 1) / TB;
  auto stage_all = [&](int tt, char* stage) {
    stage_tiles<TERMS>(t1, t2, a.S, tt, stage, wave, lane);
    if (MODE == MODE_DKV && threadIdx.x < 128) {
      int r = tt * TB + (threadIdx.x & 63); r = r < a.S ? r : a.S - 1;
      const float* src = (threadIdx.x < 64 ? a.lse2 : a.delta) + (int64_t)bh * a.S + r;
      reinterpret_cast<float*>(stage + NPL * PLANE)[threadIdx.x] = *src;
    }
  };
  stage_all(0, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto tile = [&](auto tail_t, int tt) {
    constexpr bool TAIL = decltype(tail_t)::value;   // only the last tile can hold streamed rows beyond S
    const char* cur = smem + (tt & 1) * STAGE;
    if (tt + 1 < ntiles) stage_all(tt + 1, smem + ((tt + 1) & 1) * STAGE);
    const char* t1_hi = cur;
    const char* t1_lo = cur + PLANE;
    const char* t2_hi = cur + (TERMS == 3 ? 2 : 1) * PLANE;
    const char* t2_lo = cur + 3 * PLANE;
    const float* stats = reinterpret_cast<const float*>(cur + NPL * PLANE);

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // s (scores) and dp for this 32-row sub-tile: rows = streamed rows (registers), cols = lanes
      f32x16 sacc = (f32x16){}, dpacc = (f32x16){};
      rows_times_lane<TERMS>(sacc, t1_hi, t1_lo, roff, sub, l1h, l1l);
      rows_times_lane<GT>(dpacc, t2_hi, t2_lo, roff, sub, l2h, l2l);
      // MODE_DQ: s^T = K q^T (rows keys), dp^T = V dO^T.   MODE_DKV: s = Q k^T (rows queries), dp = dO v^T.
      f32x16 pacc, dsacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // row inside the 64-row tile (shadows the lane-level trow)
        const float lse = MODE == MODE_DQ ? lse_l : stats[trow];
        const float dl = MODE == MODE_DQ ? delta_l : stats[64 + trow];
        float pv = __builtin_amdgcn_exp2f(sacc[r] - lse);   // q planes carry log2(e): scores are already in log2 units
        if (TAIL && tt * TB + trow >= a.S) pv = 0.f;                        // streamed row beyond S
        pacc[r] = pv;
        dsacc[r] = pv * (dpacc[r] - dl);
      }
      trT_times_acc<GT>(g1, t1_hi, t1_lo, sub, dsacc, toff, toffx);             // dQ^T += K^T dS^T   |  dK^T += Q^T dS
      if (MODE == MODE_DKV) trT_times_acc<GT>(g2, t2_hi, t2_lo, sub, pacc, toff, toffx);   // dV^T += dO^T P
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int tt = 0; tt + 1 < ntiles; ++tt) tile(std::false_type{}, tt);
  if (a.S % TB) tile(std::true_type{}, ntiles - 1);
  else tile(std::false_type{}, ntiles - 1);

  const int row = l0 + ll_;
  if (row < a.S) {
    const int64_t out = ((int64_t)b * a.S + row) * (3 * d) + h * 64;
    if (MODE == MODE_DQ) {
      store_grad(a, g1, out, half, a.qscale);
    } else {
      store_grad(a, g1, out + d, half, 0.6931471805599453f);   // dK = dS^T q and the stored q is q * log2(e)
      store_grad(a, g2, out + 2 * d, half, 1.0f);
    }
  }
}

// GT = products of the GRADIENT contractions (dp, dq, dk, dv): TERMS, or 1 with TERMS = 3 (the opt-in bf16 backward: the
// scores are still recomputed in split-bf16, because p = exp2(s - lse) has to reproduce the forward's probabilities)
template <int TERMS, int MODE, int GT>
int launch_mode(const BwdArgs& a, hipStream_t s) {
  constexpr int lds = 2 * ((TERMS == 3 ? 4 : 2) * PLANE + 512);
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)attention_bwd_kernel<TERMS, MODE, GT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
  dim3 grid(((a.S + LB - 1) / LB) * a.B * a.H);
  hipLaunchKernelGGL((attention_bwd_kernel<TERMS, MODE, GT>), grid, dim3(kThreads), lds, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

}  // namespace

int launch_attention_bwd(awt_ctx* c, const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* k_hi, const bf16_t* k_lo,
                         const bf16_t* v_hi, const bf16_t* v_lo, const bf16_t* o_hi, const bf16_t* o_lo, const bf16_t* do_hi,
                         const bf16_t* do_lo, const float* lse2, float* delta, bf16_t* g_hi, bf16_t* g_lo, int B, int H, int S,
                         float qscale, int terms, int grad_terms, hipStream_t s) {
  AWT_REQUIRE(B > 0 && H > 0 && S > 0, AWT_ERR_INVALID, "attention_bwd: bad shape");
  AWT_REQUIRE(grad_terms == terms || (terms == 3 && grad_terms == 1), AWT_ERR_INVALID, "attention_bwd: grad_terms must equal terms, or be 1 with terms 3");
  AWT_REQUIRE(terms == 1 || terms == 3, AWT_ERR_INVALID, "attention_bwd: terms must be 1 or 3");
  AWT_REQUIRE(q_hi && k_hi && v_hi && o_hi && do_hi && lse2 && delta && g_hi, AWT_ERR_INVALID, "attention_bwd: null argument");
  AWT_REQUIRE(terms == 1 || (q_lo && k_lo && v_lo && o_lo && do_lo && g_lo), AWT_ERR_INVALID, "attention_bwd: lo planes required");
  ProfScope prof(c, AWT_PROF_ATTENTION_BWD, s, 14.0 * (double)B * H * (double)S * S * 64);   // 7 products (s and dp are formed twice)
  // delta = rowsum(dO * O) is formed by the dq launch (its lanes hold the dO rows already) and read back by the dk / dv launch that follows it on the stream
  BwdArgs a{q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, do_hi, do_lo, lse2, delta, o_hi, terms == 3 ? o_lo : nullptr, g_hi, terms == 3 ? g_lo : nullptr, B, H, S, qscale};
  int rc = terms == 1 ? launch_mode<1, MODE_DQ, 1>(a, s) : (grad_terms == 3 ? launch_mode<3, MODE_DQ, 3>(a, s) : launch_mode<3, MODE_DQ, 1>(a, s));
  if (rc) return rc;
  return terms == 1 ? launch_mode<1, MODE_DKV, 1>(a, s) : (grad_terms == 3 ? launch_mode<3, MODE_DKV, 3>(a, s) : launch_mode<3, MODE_DKV, 1>(a, s));
}
