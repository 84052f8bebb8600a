// Flash-style multi-head self-attention forward (K10) for head_dim 64, no mask -- gfx950.
//
//   O[b, s, h, :] = softmax_k( q[b,h,s,:] . k[b,h,k,:] ) v[b,h,k,:]      q is pre-scaled by head_dim^-1/2 in the QKV
//   epilogue, as the reference scales q BEFORE q k^T (HF:modeling_whisper.py:309, 215-238), and additionally by
//   log2(e): the q planes this kernel reads hold q * log2(e), so scores are in log2 units and exp is a bare v_exp_f32.
//
// Structure (per workgroup: 4 waves x 32 query rows = 128 queries of one (batch, head); K/V tiles of 64 keys):
//  * "swapped" first product  S^T = K Q^T  on v_mfma_f32_32x32x16_bf16: the 32x32 result has the QUERY on the lane
//    and 16 keys in registers, so the online-softmax row max / row sum are in-register reductions plus one
//    cross-half exchange, and the exponentiated tile is directly the B operand of the second product
//    O^T += V^T P^T  (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").
//  * K and V tiles are copied global -> LDS by 16-byte LDS-DMA, double buffered, chunk-XOR-swizzled on the source
//    address; K fragments are ds_read_b128 rows, V^T fragments come from the hardware transpose read
//    ds_read_b64_tr_b16 in the permuted key order the P operand has.
//  * TERMS = 3 runs both products in split-bf16 (q, k, v and P as hi + lo planes, three MFMAs per fragment pair).
//  * each wave owns QT = 2 query tiles (64 queries; QT = 1 for small grids): K / V fragments, the LDS tile, its DMA and the barrier are amortised
//    over 96 MFMAs instead of 48;
//  * fp32 running max / sum / output accumulators; S = 1500 is not a multiple of 64: tail keys are masked,
//    tail rows clamped on load and skipped on store.
#include <type_traits>
#include "common.h"

namespace {

constexpr int kThreads = 256;
// QT (template parameter) = 32-query tiles per wave: K / V fragments, LDS tiles and the barrier are amortised over QT x 48
// MFMAs.  QT = 2 for full launches; QT = 1 (twice the workgroups, half the time per K/V tile) when the grid would not
// fill the chip's 512 workgroup slots (one or two clips).
#ifndef AWT_ATTN_KB
#define AWT_ATTN_KB 64
#endif
constexpr int KB = AWT_ATTN_KB;   // keys per tile (32 or 64)
constexpr int NSUB = KB / 32;     // 32-key sub-tiles per tile
constexpr int PLANE = KB * 64 * 2;  // 8 KiB: [64 keys][64 dims] bf16

struct AttnArgs {
  const bf16_t *q_hi, *q_lo, *k_hi, *k_lo, *v_hi, *v_lo;
  bf16_t *o_hi, *o_lo; float* o_f32;
  float* lse;   // optional [B, H, S]: log2-domain log-sum-exp of each score row (saved for the backward pass)
  int B, H, S;
};

// Chunk swizzle of a [rows][64 x 16-bit] LDS image (128-byte rows, eight 16-byte chunks) that serves BOTH kinds of read: chunk ^ swz(row) with swz = the three
// bits of row >> 1 with bits 0 and 2 exchanged.  ds_read_b128 of a 32x32x16 operand (16 distinct rows per 16-lane group, one chunk column) needs swz to be a
// bijection of (row >> 1) & 7 over those rows -- any bit permutation is; ds_read_b64_tr_b16 (a 32-lane half takes four consecutive rows x four chunks) needs rows
// r and r + 2 (256 bytes apart: the same banks) in different 64-byte halves of their rows, i.e. bit 2 of the swizzle must follow bit 0 of row >> 1.  With the
// plain (row >> 1) & 7 used until round 3 every transposed read was a 2-way bank conflict (SQ_LDS_BANK_CONFLICT = 24 % of the f16f8 kernel's LDS-active cycles).
// Consequences for the offsets below: rows + 8 flip bit 2 of row >> 1 = bit 0 of the swizzle (byte ^ 16), rows + 16 / + 32 change nothing.
__device__ __forceinline__ int swz(int row) { const int x = (row >> 1) & 7; return (x & 2) | ((x & 1) << 2) | (x >> 2); }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) { glds16_asm(gsrc, lds_wave_base); }   // common.h: invisible to hipcc's vmcnt bookkeeping

// One K/V tile, global -> LDS.  The source address is a wave-uniform tile base (SGPRs) plus a 32-bit per-lane offset, so the
// eight DMA instructions share two offset VGPRs instead of eight 64-bit address pairs.
template <int TERMS>
__device__ __forceinline__ void stage_kv(const AttnArgs& a, int64_t head_off, int kt, char* stage, int wave, int lane) {
  const int64_t tile_off = head_off + (int64_t)kt * (KB * 64);
  const bf16_t* kh = a.k_hi + tile_off; const bf16_t* kl = a.k_lo + tile_off;
  const bf16_t* vh = a.v_hi + tile_off; const bf16_t* vl = a.v_lo + tile_off;
  const int last = a.S - 1 - kt * KB;          // rows past the sequence end re-read its last key (masked in the tail tile)
#pragma unroll
  for (int it = 0; it < NSUB; ++it) {
    const int p = it * kThreads + wave * 64 + lane;
    const int row = p >> 3;
    const int c = (p & 7) ^ swz(row);
    const unsigned off = (unsigned)(min(row, last) * 64 + c * 8);
    char* dst = stage + (it * kThreads + wave * 64) * 16;
    glds16(kh + off, dst);
    if (TERMS == 3) glds16(kl + off, dst + PLANE);
    glds16(vh + off, dst + (TERMS == 3 ? 2 : 1) * PLANE);
    if (TERMS == 3) glds16(vl + off, dst + 3 * PLANE);
  }
}

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

template <int TERMS, int QT, bool F16>
__global__ __launch_bounds__(kThreads, 2) void attention_kernel(AttnArgs a) {
  constexpr int QW = 32 * QT;       // queries per wave
  constexpr int QB = 4 * QW;        // queries per workgroup
  constexpr int NPL = TERMS == 3 ? 4 : 2;
  constexpr int STAGE = NPL * PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware 1-D grid: the hardware deals block i to XCD i % 8; remap so that each XCD owns a contiguous run of
  // (head, query-block) pairs -- the query blocks of one head then share that XCD's L2 copy of the head's K and V
  // (without this every XCD fetched every head: 5.1 GB of HBM reads per launch against 0.9 GB algorithmic).
  const int nqb = (a.S + QB - 1) / QB;
  const int nwg = nqb * a.B * a.H;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int bh = logical / nqb;                   // b * H + h
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = (logical - bh * nqb) * QB + wave * QW;
  const int64_t head_off = (int64_t)bh * a.S * 64;
  const int ql = lane & 31, half = lane >> 5;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][16 ks + 8 half + j], j = 0..7
  bf16x8 qh[QT][4], qlo[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int q = q0 + 32 * t + ql; q = q < a.S ? q : a.S - 1;
    const int64_t off = head_off + (int64_t)q * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[t][ks] = *reinterpret_cast<const bf16x8*>(a.q_hi + off + ks * 16);
      if (TERMS == 3) qlo[t][ks] = *reinterpret_cast<const bf16x8*>(a.q_lo + off + ks * 16);
    }
  }

  f32x16 oacc[QT][2];
  float m_run[QT], l_run[QT];            // running max (log2 domain) and this half-wave's partial row sum
#pragma unroll
  for (int t = 0; t < QT; ++t) { oacc[t][0] = (f32x16){}; oacc[t][1] = (f32x16){}; m_run[t] = -1.0e30f; l_run[t] = 0.f; }

  // ---- loop-invariant LDS byte offsets (everything else is an immediate): the swizzle term (row >> 1) & 7 does not
  // change when a row moves by 16, so sub-tiles / k-steps / dim-tiles differ by constants or by one XOR bit.
  //   K fragment (row = 32 kt2 + ql, chunk 2 ks + half):  koff[ks] + 4096 kt2
  //   V^T blocks (key0 = 32 kt2 + 16 s + 4 (g >> 1) + qq, chunk 4 et + cc):  dim-tile et toggles chunk bit 2 (byte bit 6),
  //   the +8-key block adds 1024 and flips the swizzle's bit 0:  off0(et) = voff ^ (64 et),  off1(et) = (off0(et) ^ 16) + 1024
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ql * 128 + (((2 * ks + half) ^ swz(ql)) << 4);
  const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
  const int vkey = 4 * (g >> 1) + qq;
  const int voff = vkey * 128 + (((2 * (g & 1) + (pp >> 1)) ^ swz(vkey)) << 4) + 8 * (pp & 1);
  const int voffx = voff ^ 64;

  const int ntiles = (a.S + KB - 1) / KB;
#ifdef AWT_ATTN_STAGGER
  // the two waves of a SIMD (one from each of the CU's two workgroups) alternate between an MFMA phase and a VALU phase of
  // equal length; started together they stay in step and never overlap.  Delay the odd wave slot of the first round.
  {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((int)blockIdx.x < AWT_ATTN_STAGGER_BLOCKS && (hwid & 1)) __builtin_amdgcn_s_sleep(AWT_ATTN_STAGGER);
  }
#endif
  stage_kv<TERMS>(a, head_off, 0, smem, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One K/V tile.  TAIL (compile time) masks keys >= S; it is instantiated only for the last tile, so full tiles carry no
  // compare / select instructions (they were a quarter of the loop's VALU work, and this loop is VALU-bound).
  auto tile = [&](auto tail_t, int kt) {
    constexpr bool TAIL = decltype(tail_t)::value;
    const char* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < ntiles) stage_kv<TERMS>(a, head_off, kt + 1, smem + ((kt + 1) & 1) * STAGE, wave, lane);
    const char* k_hi = cur;
    const char* k_lo = cur + PLANE;
    const char* v_hi = cur + (TERMS == 3 ? 2 : 1) * PLANE;
    const char* v_lo = cur + 3 * PLANE;

    // ---- S^T = K Q^T : two 32-key sub-tiles, rows = keys, cols (lanes) = queries; q carries log2(e), so S is in log2 units
    f32x16 sacc[QT][NSUB];
#pragma unroll
    for (int kt2 = 0; kt2 < NSUB; ++kt2) {
#pragma unroll
      for (int t = 0; t < QT; ++t) sacc[t][kt2] = (f32x16){};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = koff[ks] + kt2 * 4096;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(k_hi + off);
        bf16x8 kl;
        if (TERMS == 3) kl = *reinterpret_cast<const bf16x8*>(k_lo + off);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          if (TERMS == 3) {
            sacc[t][kt2] = mfma32<F16>(kh, qlo[t][ks], sacc[t][kt2]);
            sacc[t][kt2] = mfma32<F16>(kl, qh[t][ks], sacc[t][kt2]);
          }
          sacc[t][kt2] = mfma32<F16>(kh, qh[t][ks], sacc[t][kt2]);
        }
      }
    }

    // ---- online softmax.  C/D layout 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      float tmax = -1.0e30f;
#pragma unroll
      for (int kt2 = 0; kt2 < NSUB; ++kt2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (TAIL) {
            const int key = kt * KB + kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            sacc[t][kt2][r] = key < a.S ? sacc[t][kt2][r] : -1.0e30f;
          }
          tmax = fmaxf(tmax, sacc[t][kt2][r]);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run[t], tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
      m_run[t] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kt2 = 0; kt2 < NSUB; ++kt2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sacc[t][kt2][r] - m_new);
          sacc[t][kt2][r] = pv;
          psum += pv;
        }
      l_run[t] = l_run[t] * alpha + psum;
#pragma unroll
      for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][et][r] *= alpha;
    }

    // ---- O^T += V^T P^T : P registers 8 s .. 8 s + 7 of a sub-tile are the B fragment of k-step s;
    //      element j of lane-half `half` is key 16 s + 8 (j >> 2) + 4 half + (j & 3) of that sub-tile.
    //      V^T fragments come from the transpose read: 16-lane group g covers dims 16 (g & 1) .. + 15 of the dim-tile for
    //      lane-half g >> 1; lane 4 qq + pp of the group supplies row (key) qq, columns 4 pp .. 4 pp + 3.
#pragma unroll
    for (int kt2 = 0; kt2 < NSUB; ++kt2)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ph[QT], pl[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            bf16_t hi, lo;
            split16<F16>(sacc[t][kt2][8 * s2 + j], hi, lo);
            ph[t][j] = (short)hi;
            if (TERMS == 3) pl[t][j] = (short)lo;
          }
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          const int cst = kt2 * 4096 + s2 * 2048;
          const int off0 = (et == 0 ? voff : voffx) + cst;
          const int off1 = (off0 ^ 16) + 1024;          // keys + 8: swizzle bit 0 flips (see swz)
          const bf16x4 va = tr_read(v_hi + off0), vb = tr_read(v_hi + off1);
          const bf16x8 vh = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
          bf16x8 vl;
          if (TERMS == 3) {
            const bf16x4 la = tr_read(v_lo + off0), lb = tr_read(v_lo + off1);
            vl = (bf16x8){la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
          }
#pragma unroll
          for (int t = 0; t < QT; ++t) {
            if (TERMS == 3) {
              oacc[t][et] = mfma32<F16>(vh, pl[t], oacc[t][et]);
              oacc[t][et] = mfma32<F16>(vl, ph[t], oacc[t][et]);
            }
            oacc[t][et] = mfma32<F16>(vh, ph[t], oacc[t][et]);
          }
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int kt = 0; kt + 1 < ntiles; ++kt) tile(std::false_type{}, kt);
  if (a.S % KB) tile(std::true_type{}, ntiles - 1);
  else tile(std::false_type{}, ntiles - 1);

  // ---- normalise and store: lane holds query (lane & 31) of each query tile, dims (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const float l_tot = l_run[t] + __shfl_xor(l_run[t], 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + 32 * t + ql;
    if (a.lse && q < a.S && half == 0) a.lse[(int64_t)bh * a.S + q] = m_run[t] + __builtin_amdgcn_logf(l_tot);
    if (q < a.S) {
      const int64_t row = ((int64_t)b * a.S + q) * (a.H * 64) + h * 64;
#pragma unroll
      for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int e = 32 * et + 8 * g4 + 4 * half;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = oacc[t][et][4 * g4 + j] * inv;
          if (a.o_f32) {
            *reinterpret_cast<float4*>(a.o_f32 + row + e) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            bf16_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split16<F16>(v[j], hi[j], lo[j]);
            *reinterpret_cast<uint2*>(a.o_hi + row + e) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
            if (a.o_lo) *reinterpret_cast<uint2*>(a.o_lo + row + e) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
          }
        }
    }
  }
}

template <int TERMS, int QT, bool F16 = false>
int launch_t(const AttnArgs& a, hipStream_t s) {
  constexpr int lds = 2 * (TERMS == 3 ? 4 : 2) * PLANE;
  constexpr int QB = 4 * 32 * QT;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)attention_kernel<TERMS, QT, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
  dim3 grid(((a.S + QB - 1) / QB) * a.B * a.H);
  hipLaunchKernelGGL((attention_kernel<TERMS, QT, F16>), grid, dim3(kThreads), lds, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

}  // namespace

int launch_attention(awt_ctx* c, const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* k_hi, const bf16_t* k_lo,
                     const bf16_t* v_hi, const bf16_t* v_lo, bf16_t* o_hi, bf16_t* o_lo, float* o_f32, float* lse, int B, int H,
                     int S, int prec, hipStream_t s) {
  AWT_REQUIRE(B > 0 && H > 0 && S > 0, AWT_ERR_INVALID, "attention: bad shape");
  AWT_REQUIRE(prec == PREC_BF16 || prec == PREC_F16 || prec == PREC_BF16X3 || prec == PREC_F16X3, AWT_ERR_INVALID, "attention: precision must be bf16 (1), fp16 (2), bf16x3 (3) or fp16x3 (4) here");
  const int terms = prec_products(prec);
  AWT_REQUIRE(q_hi && k_hi && v_hi && (o_hi || o_f32), AWT_ERR_INVALID, "attention: null plane");
  AWT_REQUIRE(terms == 1 || (q_lo && k_lo && v_lo), AWT_ERR_INVALID, "attention: lo planes required for terms == 3");
  AttnArgs a{q_hi, q_lo, k_hi, k_lo, v_hi, v_lo, o_hi, o_lo, o_f32, lse, B, H, S};
  ProfScope prof(c, AWT_PROF_ATTENTION, s, 4.0 * (double)B * H * (double)S * S * 64);
  // 512 workgroup slots (256 CUs x 2): a launch lasts ceil(workgroups / 512) rounds.  One query tile per wave doubles the
  // workgroups at ~0.55 of the duration each (less K / V amortisation); take it when that is fewer round-equivalents --
  // grids under one round, and grids whose last round would be mostly empty (Whisper-tiny at B = 32: 3 rounds vs 2.75).
  const int64_t wg2 = (int64_t)((S + 255) / 256) * B * H, wg1 = (int64_t)((S + 127) / 128) * B * H;
  const double cost2 = (double)((wg2 + 511) / 512), cost1 = 0.55 * (double)((wg1 + 511) / 512);
  if (prec == PREC_F16X3) return cost1 < cost2 ? launch_t<3, 1, true>(a, s) : launch_t<3, 2, true>(a, s);
  if (prec == PREC_F16) return cost1 < cost2 ? launch_t<1, 1, true>(a, s) : launch_t<1, 2, true>(a, s);
  if (cost1 < cost2) return terms == 3 ? launch_t<3, 1>(a, s) : launch_t<1, 1>(a, s);
  return terms == 3 ? launch_t<3, 2>(a, s) : launch_t<1, 2>(a, s);
}
