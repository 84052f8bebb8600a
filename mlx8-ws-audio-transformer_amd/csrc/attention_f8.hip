// Flash-style self-attention forward (K10) in the f16f8 operand format (common.h PREC_F16F8) -- gfx950.
//
// Same structure as attention.hip (swapped first product S^T = K Q^T with the query on the lane, online softmax in the log2
// domain, O^T += V^T P^T with the exponentiated accumulator tile as B operand, 64-key K/V tiles by 16-byte LDS-DMA, double
// buffered), but every split product  a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi  issues
//     a_hi b_hi               on v_mfma_f32_32x32x16_f16            (fp16 planes, 11 significant bits),
//     a_hi b_lo + a_lo b_hi   on v_mfma_scale_f32_32x32x64_f8f6f4   (e4m3 planes with fixed power-of-two scales),
// i.e. 8 + 4 MFMAs of 32 + 64 cycles per (64-key tile, 32-query tile) and product instead of 24 of 32: 2/3 of the matrix
// pipe time of the split-bf16 kernel, at an error of 2^-16 instead of 2^-17 per product (DESIGN.md "Numerics").
//
// Operand images.  q, k, v arrive head-major [B, H, S, 64] as three planes each: fp16 (128-byte rows), hi8 and lo8 (64-byte
// rows).  The contraction order of the scaled MFMA is free as long as both operands use the same lane -> k assignment
// (tools/mx_probe3.hip), so:
//   S^T cross terms: lane (row l & 31, half l >> 5) holds dims 32 half .. 32 half + 31 of its key (A) / query (B): 32
//     contiguous bytes of a 64-byte row, two ds_read_b128 (K) or two global loads (Q, once per workgroup);
//   O^T cross terms: the B operand is the lane's own 32 probabilities of the 64-key tile, byte 16 kt2 + r = accumulator
//     register r of 32-key sub-tile kt2 = key 32 kt2 + (r & 3) + 8 (r >> 2) + 4 half; the A operand V8^T is gathered in
//     exactly that key order by four ds_read_b64_tr_b8 per 32-dim tile (per 16-lane group a block of 8 rows x 16 bytes, lane
//     2 q + p supplies the address of row q, bytes 8 p .. 8 p + 7, lane i receives column i with row q in byte q).
#include <type_traits>
#include "common.h"

extern int g_attn_shape;

namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int KB = 64;               // keys per tile
constexpr int PL16 = KB * 64 * 2;    // 8 KiB: [64 keys][64 dims] fp16
constexpr int PL8 = KB * 64;         // 4 KiB: [64 keys][64 dims] e4m3
constexpr int STAGE = 2 * PL16 + 4 * PL8;   // K16 | V16 | K8 | Klo8 | V8 | Vlo8 = 32 KiB
constexpr int OFF_K16 = 0, OFF_V16 = PL16, OFF_K8 = 2 * PL16, OFF_KL8 = 2 * PL16 + PL8, OFF_V8 = 2 * PL16 + 2 * PL8, OFF_VL8 = 2 * PL16 + 3 * PL8;

struct Attn8Args {
  const bf16_t *q16, *k16, *v16;                 // fp16 bits
  const uint8_t *q8, *ql8, *k8, *kl8, *v8, *vl8;
  bf16_t* o16; uint8_t *o8, *ol8; float* o_f32;  // output: f16f8 activation planes [B*S, H*64] or fp32
  float* lse;
  int B, H, S;
  char* o_ilv;                                   // ... or the same activation as split lines (common.h Act::ilv), when set
};

__device__ __forceinline__ int swz16(int row) { return (row >> 1) & 7; }        // K fp16 plane: 128-byte rows, 8 chunks, read by ds_read_b128 (32 distinct rows per chunk column)
// V fp16 plane, read by ds_read_b64_tr_b16: a 32-lane half takes FOUR consecutive keys x four 16-byte chunks (8 bytes per lane).  Rows r and r + 2 are 256 bytes apart
// = the same banks, so they must sit in different 64-byte halves of their rows: chunk ^ 4 for the odd row pairs.  (With swz16 on this plane -- the XOR of rows 2, 3
// is 1, which stays inside chunks 0..3 -- every transposed read was a 2-way bank conflict: SQ_LDS_BANK_CONFLICT = 24 % of the LDS-active cycles of the round-3
// kernel, profiles/r03_attention_shapes.txt; 16 of a tile's 32 LDS reads.)
__device__ __forceinline__ int swz_v16(int row) { return ((row >> 1) & 1) << 2; }
__device__ __forceinline__ int swz_k8(int row) { return (row >> 2) & 3; }       // K8 planes: 64-byte rows read by ds_read_b128
__device__ __forceinline__ int swz_v8(int row) { return ((row >> 3) & 1) << 1; }   // V8 planes: rows k and k + 8 land in different 32-byte halves

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) { glds16_asm(gsrc, lds_wave_base); }   // common.h: invisible to hipcc's vmcnt bookkeeping
__device__ __forceinline__ bf16x4 tr_read16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ i32x2 tr_read8(const char* p) {
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)p);
}

// One K/V tile, global -> LDS.  The tile is 8 + 8 groups of 64 sixteen-byte slots in the fp16 planes and 4 groups in each of the
// four e4m3 planes (an LDS-DMA instruction fills one group: wave-uniform base + lane * 16); wave w of NW takes groups w, w + NW, ...
template <int NW>
__device__ __forceinline__ void stage_kv(const Attn8Args& a, int64_t head_off, int kt, char* stage, int wave, int lane) {
  const int64_t tile_off = head_off + (int64_t)kt * (KB * 64);
  const int last = a.S - 1 - kt * KB;          // rows past the sequence end re-read its last key (masked in the tail tile)
  const bf16_t* k16 = a.k16 + tile_off; const bf16_t* v16 = a.v16 + tile_off;
#pragma unroll
  for (int g0 = 0; g0 < 8; g0 += NW) {
    const int grp = g0 + wave;
    if (grp < 8) {
      const int p = grp * 64 + lane;
      const int row = p >> 3;
      const unsigned off = (unsigned)(min(row, last) * 64 + (((p & 7) ^ swz16(row)) << 3));
      const unsigned offv16 = (unsigned)(min(row, last) * 64 + (((p & 7) ^ swz_v16(row)) << 3));
      char* dst = stage + grp * 1024;
      glds16(k16 + off, dst + OFF_K16);
      glds16(v16 + offv16, dst + OFF_V16);
    }
  }
#pragma unroll
  for (int g0 = 0; g0 < 4; g0 += NW) {
    const int grp = g0 + wave;
    if (grp < 4) {
      const int p = grp * 64 + lane;
      const int row = p >> 2, c = p & 3;
      const int r = min(row, last);
      const unsigned offk = (unsigned)(r * 64 + ((c ^ swz_k8(row)) << 4));
      const unsigned offv = (unsigned)(r * 64 + ((c ^ swz_v8(row)) << 4));
      char* dst = stage + grp * 1024;
      glds16(a.k8 + tile_off + offk, dst + OFF_K8);
      glds16(a.kl8 + tile_off + offk, dst + OFF_KL8);
      glds16(a.v8 + tile_off + offv, dst + OFF_V8);
      glds16(a.vl8 + tile_off + offv, dst + OFF_VL8);
    }
  }
}

// probabilities p in [0, 1] -> e4m3 of p 2^E without clamping (p 2^8 <= 256, (p - fp16(p)) 2^19 <= 128): the conversion
// instruction applies the power-of-two scale itself (it divides by its scale operand)
template <int E>
__device__ __forceinline__ int fp8x4_scaled(float a, float b, float c, float d) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  constexpr float inv = pow2f(-E);
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, inv, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, inv, true);
  return __builtin_bit_cast(int, r);
}

template <int QT, int NW>
__global__ __launch_bounds__(64 * NW, (QT == 1 && NW == 6) ? 3 : 2) void attention_f16f8_kernel(Attn8Args a) {
  constexpr int QW = 32 * QT, QB = NW * QW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware 1-D grid (attention.hip): the query blocks of one head share an XCD's L2 copy of the head's K and V
  const int nqb = (a.S + QB - 1) / QB;
  const int nwg = nqb * a.B * a.H;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int bh = logical / nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = (logical - bh * nqb) * QB + wave * QW;
  const int64_t head_off = (int64_t)bh * a.S * 64;
  const int ql = lane & 31, half = lane >> 5;

  // ---- Q fragments: fp16 (lane holds Q[q][16 ks + 8 half + j]) and the two e4m3 operands (dims 32 half .. + 31)
  bf16x8 q16[QT][4];
  i32x8 q8[QT], ql8[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int q = q0 + 32 * t + ql; q = q < a.S ? q : a.S - 1;
    const int64_t off = head_off + (int64_t)q * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q16[t][ks] = *reinterpret_cast<const bf16x8*>(a.q16 + off + half * 8 + ks * 16);
    const uint4 x0 = *reinterpret_cast<const uint4*>(a.q8 + off + half * 32), x1 = *reinterpret_cast<const uint4*>(a.q8 + off + half * 32 + 16);
    const uint4 y0 = *reinterpret_cast<const uint4*>(a.ql8 + off + half * 32), y1 = *reinterpret_cast<const uint4*>(a.ql8 + off + half * 32 + 16);
    q8[t] = (i32x8){(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
    ql8[t] = (i32x8){(int)y0.x, (int)y0.y, (int)y0.z, (int)y0.w, (int)y1.x, (int)y1.y, (int)y1.z, (int)y1.w};
  }

  f32x16 oacc[QT][2];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) { oacc[t][0] = (f32x16){}; oacc[t][1] = (f32x16){}; m_run[t] = -1.0e30f; l_run[t] = 0.f; }

  // ---- loop-invariant LDS byte offsets
  //   K16 fragment (row = 32 kt2 + ql, chunk 2 ks + half): koff[ks] + 4096 kt2          (as attention.hip)
  //   K8 fragment (row = 32 kt2 + ql, chunks 2 half, 2 half + 1): k8off[c] + 2048 kt2    (swz_k8 repeats every 16 rows)
  //   V16^T blocks: voff / voffx as attention.hip
  //   V8^T blocks (read n = 0..3 covers operand bytes 8 n .. 8 n + 7): row = 16 n + 8 (q >> 2) + (q & 3) + 4 half_of_group,
  //     16-byte chunk (g & 1) + 2 et, XOR 2 (q >> 2):  v8off + 1024 n, dim-tile et toggles byte bit 5
  int koff[4], k8off[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ql * 128 + (((2 * ks + half) ^ swz16(ql)) << 4);
#pragma unroll
  for (int c = 0; c < 2; ++c) k8off[c] = ql * 64 + (((2 * half + c) ^ swz_k8(ql)) << 4);
  const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
  const int vkey = 4 * (g >> 1) + qq;
  const int voff = vkey * 128 + (((2 * (g & 1) + (pp >> 1)) ^ swz_v16(vkey)) << 4) + 8 * (pp & 1);
  const int voffx = voff ^ 64;
  const int tq = li >> 1, tp = li & 1;
  const int v8key = 8 * (tq >> 2) + (tq & 3) + 4 * (g >> 1);
  const int v8off = v8key * 64 + ((((g & 1)) ^ swz_v8(v8key)) << 4) + 8 * tp;

  const int ntiles = (a.S + KB - 1) / KB;
  stage_kv<NW>(a, head_off, 0, smem, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto tile = [&](auto tail_t, int kt) {
    constexpr bool TAIL = decltype(tail_t)::value;
    const char* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < ntiles) stage_kv<NW>(a, head_off, kt + 1, smem + ((kt + 1) & 1) * STAGE, wave, lane);

    // ---- S^T = K Q^T
    f32x16 sacc[QT][2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
      for (int t = 0; t < QT; ++t) sacc[t][kt2] = (f32x16){};
      {
        const uint4 x0 = *reinterpret_cast<const uint4*>(cur + OFF_K8 + k8off[0] + kt2 * 2048), x1 = *reinterpret_cast<const uint4*>(cur + OFF_K8 + k8off[1] + kt2 * 2048);
        const uint4 y0 = *reinterpret_cast<const uint4*>(cur + OFF_KL8 + k8off[0] + kt2 * 2048), y1 = *reinterpret_cast<const uint4*>(cur + OFF_KL8 + k8off[1] + kt2 * 2048);
        const i32x8 k8 = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
        const i32x8 kl8 = {(int)y0.x, (int)y0.y, (int)y0.z, (int)y0.w, (int)y1.x, (int)y1.y, (int)y1.z, (int)y1.w};
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          sacc[t][kt2] = mfma32_f8<e8m0(-kF8KV), e8m0(-kF8Q - kF8Lo)>(k8, ql8[t], sacc[t][kt2]);
          sacc[t][kt2] = mfma32_f8<e8m0(-kF8KV - kF8Lo), e8m0(-kF8Q)>(kl8, q8[t], sacc[t][kt2]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(cur + OFF_K16 + koff[ks] + kt2 * 4096);
#pragma unroll
        for (int t = 0; t < QT; ++t) sacc[t][kt2] = mfma32<true>(kh, q16[t][ks], sacc[t][kt2]);
      }
    }

    // ---- online softmax, then the accumulator tile becomes the three P operands in place
    bf16x8 p16[QT][4];        // fp16 B fragments of k-steps (kt2, s2): registers 8 s2 .. 8 s2 + 7 of sub-tile kt2
    i32x8 p8[QT], pl8[QT];    // e4m3 operands: byte 16 kt2 + r
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      float tmax = -1.0e30f;
#pragma unroll
      for (int kt2 = 0; kt2 < 2; ++kt2)
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
      for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float pv[4], lo[4]; bf16_t hb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pv[j] = __builtin_amdgcn_exp2f(sacc[t][kt2][4 * r4 + j] - m_new);
            psum += pv[j];
            hb[j] = f32_to_f16(pv[j]);
            lo[j] = __builtin_fmaf(f16_to_f32(hb[j]), -1.0f, pv[j]);
            p16[t][2 * kt2 + (r4 >> 1)][4 * (r4 & 1) + j] = (short)hb[j];
          }
          p8[t][4 * kt2 + r4] = fp8x4_scaled<kF8P>(pv[0], pv[1], pv[2], pv[3]);
          pl8[t][4 * kt2 + r4] = fp8x4_scaled<kF8P + kF8Lo>(lo[0], lo[1], lo[2], lo[3]);
        }
      l_run[t] = l_run[t] * alpha + psum;
#pragma unroll
      for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][et][r] *= alpha;
    }

    // ---- O^T += V^T P^T : fp16 part (k-steps of 16 keys, V16^T by ds_read_b64_tr_b16) ...
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int et = 0; et < 2; ++et) {
          const int cst = kt2 * 4096 + s2 * 2048;
          const int off0 = (et == 0 ? voff : voffx) + cst;
          const int off1 = (et == 0 ? voff : voffx) + cst + 1024;     // keys + 8: swz_v16 repeats every 4 rows, so the same lane offset (under swz16 it was the other one)
          const bf16x4 va = tr_read16(cur + OFF_V16 + off0), vb = tr_read16(cur + OFF_V16 + off1);
          const bf16x8 vh = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
          for (int t = 0; t < QT; ++t) oacc[t][et] = mfma32<true>(vh, p16[t][2 * kt2 + s2], oacc[t][et]);
        }
    // ---- ... and the two cross terms on the scaled e4m3 MFMA (one instruction covers the tile's 64 keys)
#pragma unroll
    for (int et = 0; et < 2; ++et) {
      i32x8 v8, vl8;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int off = (v8off ^ (et << 5)) + n * 1024;
        const i32x2 x = tr_read8(cur + OFF_V8 + off), y = tr_read8(cur + OFF_VL8 + off);
        v8[2 * n] = x[0]; v8[2 * n + 1] = x[1];
        vl8[2 * n] = y[0]; vl8[2 * n + 1] = y[1];
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        oacc[t][et] = mfma32_f8<e8m0(-kF8KV), e8m0(-kF8P - kF8Lo)>(v8, pl8[t], oacc[t][et]);
        oacc[t][et] = mfma32_f8<e8m0(-kF8KV - kF8Lo), e8m0(-kF8P)>(vl8, p8[t], oacc[t][et]);
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
            uint2 h16; unsigned hi8, lo8;
            f16f8x4<kF8Act>(v, h16, hi8, lo8);
            if (a.o_ilv) { store_ilv4(a.o_ilv, row + e, h16, hi8, lo8); continue; }
            *reinterpret_cast<uint2*>(a.o16 + row + e) = h16;
            if (a.o8) *reinterpret_cast<unsigned*>(a.o8 + row + e) = hi8;
            *reinterpret_cast<unsigned*>(a.ol8 + row + e) = lo8;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ software-pipelined form
// The kernel above runs, per wave and K/V tile, 1024 cycles of MFMA and ~350 VALU instructions (~1400 issue cycles) strictly one
// after the other, and the two waves of a SIMD do not interleave on their own: 3000 cycles per tile where the matrix pipe
// needs 1024.  Here each wave overlaps them itself: iteration k issues the S^T = K Q^T MFMAs of tile k + 1 with the softmax
// VALU work of tile k in the gaps between them (an MFMA occupies the issue port for 8 of its 32 / 64 cycles), then the
// O^T += V^T P^T MFMAs of tile k with the e4m3 packing of P between them.  K is therefore staged two tiles ahead and consumed
// one iteration early (K(k + 2) lands in the buffer K(k) left in iteration k - 1), V one tile ahead; still two 32 KB stages.
// One 32-query tile per wave (the second accumulator tile takes the registers of the second query tile), four waves.
// K (or V) planes of tile kt -> stage: 8 groups of 64 sixteen-byte slots in the fp16 plane, 4 in each e4m3 plane.  Four waves: two
// fp16 groups and one group of each e4m3 plane per wave; eight waves: one fp16 group and one e4m3 group (waves 0-3 hi8, 4-7 lo8).
// RING3 (the single-product P V form): `stage` is the tile's own slot of a three-deep ring -- K slots [K16 | K8 | Klo8] of 16 KB, V slots
// [V16] of 8 KB -- instead of a 32 KB stage shared by K and V.
constexpr int K3_SLOT = PL16 + 2 * PL8, V3_SLOT = PL16, K3_O8 = PL16, K3_OL8 = PL16 + PL8, V3_BASE = 3 * K3_SLOT, RING3_BYTES = 3 * (K3_SLOT + V3_SLOT);
// Per-lane byte offsets of a wave's LDS-DMA pieces inside a [64 keys x 64 dims] tile (rows past `last` re-read row `last`: the tail tile).
// Loop-invariant for every tile but the last one of a sequence whose length is not a multiple of 64.
template <int NW>
struct KvLaneOff {
  unsigned o16[8 / NW];      // K fp16 plane
  unsigned o16v[8 / NW];     // V fp16 plane (its own swizzle: swz_v16)
  unsigned o8k, o8v;         // e4m3 planes: K (ds_read_b128 swizzle) and V (transposing-read swizzle)
};
template <int NW>
__device__ __forceinline__ KvLaneOff<NW> kv_lane_off(int last, int wave, int lane) {
  KvLaneOff<NW> o;
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) {
    const int p = (i * NW + wave) * 64 + lane, row = p >> 3;
    o.o16[i] = (unsigned)(min(row, last) * 64 + (((p & 7) ^ swz16(row)) << 3)) * 2u;
    o.o16v[i] = (unsigned)(min(row, last) * 64 + (((p & 7) ^ swz_v16(row)) << 3)) * 2u;
  }
  const int p = (wave & 3) * 64 + lane, row = p >> 2;
  o.o8k = (unsigned)(min(row, last) * 64 + (((p & 3) ^ swz_k8(row)) << 4));
  o.o8v = (unsigned)(min(row, last) * 64 + (((p & 3) ^ swz_v8(row)) << 4));
  return o;
}
// K (or V) planes of tile kt -> the LDS slot at byte address `stage` (wave-uniform).  Sources are wave-uniform plane bases (SGPRs) + the lane offsets
// above, so a copy costs no VALU instruction.
template <int NW, bool ISK, bool W8 = true, bool RING3 = false>
__device__ __forceinline__ void stage_kv1(const Attn8Args& a, int64_t head_off, int kt, unsigned stage, int wave, const KvLaneOff<NW>& lo) {
  const int64_t tile_off = head_off + (int64_t)kt * (KB * 64);
  const bf16_t* p16 = (ISK ? a.k16 : a.v16) + tile_off;
  const uint8_t* p8 = (ISK ? a.k8 : a.v8) + tile_off;
  const uint8_t* pl8 = (ISK ? a.kl8 : a.vl8) + tile_off;
  constexpr int O16 = RING3 ? 0 : (ISK ? OFF_K16 : OFF_V16), O8 = RING3 ? K3_O8 : (ISK ? OFF_K8 : OFF_V8), OL8 = RING3 ? K3_OL8 : (ISK ? OFF_KL8 : OFF_VL8);
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) glds16_asm_sbase(p16, ISK ? lo.o16[i] : lo.o16v[i], stage + O16 + (i * NW + wave) * 1024);
  if constexpr (!W8) return;          // fp16-only operand (V of the single-product P V): no e4m3 planes to stage
  const int grp = wave & 3;
  const unsigned off = ISK ? lo.o8k : lo.o8v;
  if (NW == 4) {
    glds16_asm_sbase(p8, off, stage + O8 + grp * 1024);
    glds16_asm_sbase(pl8, off, stage + OL8 + grp * 1024);
  } else {
    glds16_asm_sbase(wave < 4 ? p8 : pl8, off, stage + (wave < 4 ? O8 : OL8) + grp * 1024);
  }
}

template <int NW, bool PV8>
__global__ __launch_bounds__(64 * NW, 2) void attention_f16f8_pipe_kernel(Attn8Args a) {
  constexpr int QB = 32 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nqb = (a.S + QB - 1) / QB;
  const int nwg = nqb * a.B * a.H;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int bh = logical / nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = (logical - bh * nqb) * QB + wave * 32;
  const int64_t head_off = (int64_t)bh * a.S * 64;
  const int ql = lane & 31, half = lane >> 5;

  bf16x8 q16[4];
  i32x8 q8, ql8;
  {
    int q = q0 + ql; q = q < a.S ? q : a.S - 1;
    const int64_t off = head_off + (int64_t)q * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q16[ks] = *reinterpret_cast<const bf16x8*>(a.q16 + off + half * 8 + ks * 16);
    const uint4 x0 = *reinterpret_cast<const uint4*>(a.q8 + off + half * 32), x1 = *reinterpret_cast<const uint4*>(a.q8 + off + half * 32 + 16);
    const uint4 y0 = *reinterpret_cast<const uint4*>(a.ql8 + off + half * 32), y1 = *reinterpret_cast<const uint4*>(a.ql8 + off + half * 32 + 16);
    q8 = (i32x8){(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
    ql8 = (i32x8){(int)y0.x, (int)y0.y, (int)y0.z, (int)y0.w, (int)y1.x, (int)y1.y, (int)y1.z, (int)y1.w};
  }
  f32x16 oacc[2] = {(f32x16){}, (f32x16){}};
  float m_run = -1.0e30f, l_run = 0.f;

  int koff[4], k8off[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ql * 128 + (((2 * ks + half) ^ swz16(ql)) << 4);
#pragma unroll
  for (int c = 0; c < 2; ++c) k8off[c] = ql * 64 + (((2 * half + c) ^ swz_k8(ql)) << 4);
  const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
  const int vkey = 4 * (g >> 1) + qq;
  const int voff = vkey * 128 + (((2 * (g & 1) + (pp >> 1)) ^ swz_v16(vkey)) << 4) + 8 * (pp & 1);
  const int voffx = voff ^ 64;
  const int tq = li >> 1, tp = li & 1;
  const int v8key = 8 * (tq >> 2) + (tq & 3) + 4 * (g >> 1);
  const int v8off = v8key * 64 + ((((g & 1)) ^ swz_v8(v8key)) << 4) + 8 * tp;

  const int ntiles = (a.S + KB - 1) / KB;

  // ---- the MFMA slots of one iteration, in issue order.  Every slot reads its LDS operand TWO slots ahead of its MFMA into a
  // three-entry register ring (an LDS read takes 64-128 cycles to return: read just before use, the 28 reads of a tile cost
  // more than its 24 MFMAs), and carries one chunk of VALU work that is issued between the read and the MFMA.
  //   slots  0..11  S(kt + 1)^T = K Q^T: per 32-key sub-tile the two e4m3 cross terms, then the four fp16 k-steps
  //   slots 12..19  O^T += V16^T P16^T  (kt2, s2, et)
  //   slots 20..23  O^T += V8^T Pl8^T + Vl8^T P8^T  (et x 2)
  //                 (PV8 only; without it P V is the single fp16 product: P in [0, 1] rounds to 11 significant bits and the
  //                  normaliser is the sum of the unrounded P -- DESIGN.md "Numerics" for what that costs)
#ifndef AWT_ATTN_DEPTH
#define AWT_ATTN_DEPTH 2
#endif
  constexpr int NSLOT = PV8 ? 24 : 20, DEPTH = AWT_ATTN_DEPTH, RING = DEPTH + 1;   // fragments are read DEPTH slots ahead of their MFMA
  // LDS: PV8 -- two 32 KB stages shared by K and V (K staged two tiles ahead, V one).  !PV8 -- three-deep rings (K three tiles ahead,
  // V two): a tile's LDS-DMA then has two iterations (~4 us) to land instead of one, and the wait at the end of an iteration only covers
  // the loads of the iteration before (counted vmcnt).  With one iteration of slack the timing-only build without the wait ran 18 % faster.
  constexpr bool RING3 = !PV8;
  constexpr int KO8 = RING3 ? K3_O8 : OFF_K8, KOL8 = RING3 ? K3_OL8 : OFF_KL8, VO16 = RING3 ? 0 : OFF_V16;
  // LDS byte addresses (not pointers: the per-lane read addresses below are plain 32-bit integers the compiler cannot re-derive per read)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(smem);
  auto kslot = [&](int t) -> unsigned { return RING3 ? lds0 + (t % 3) * K3_SLOT : lds0 + (t & 1) * STAGE; };
  auto vslot = [&](int t) -> unsigned { return RING3 ? lds0 + V3_BASE + (t % 3) * V3_SLOT : lds0 + (t & 1) * STAGE; };
  // Per-lane read addresses of the K slot / V slot an iteration works on: slot base + the loop-invariant lane offsets, formed ONCE per iteration
  // (eight to ten adds) and pinned in registers; every read then uses one of them plus an immediate offset.  Left to itself the compiler rebuilt
  // slot base + lane offset + constant per read: ~35 VALU instructions per iteration in a kernel that is bound by vector-instruction issue.
  struct RdAddr { unsigned k16[4], k8[2], v0, v1, v8[2]; };
  auto rd_addr = [&](unsigned kb, unsigned vb) {
    RdAddr r;
    constexpr bool PIN = !PV8;      // the every-cross-term form has no registers to spare for pinned addresses (it spilled): the compiler keeps its own way there
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { r.k16[ks] = kb + koff[ks]; if constexpr (PIN) asm volatile("" : "+v"(r.k16[ks])); }
#pragma unroll
    for (int c = 0; c < 2; ++c) { r.k8[c] = kb + k8off[c]; if constexpr (PIN) asm volatile("" : "+v"(r.k8[c])); }
    r.v0 = vb + VO16 + voff; r.v1 = vb + VO16 + voffx;
    if constexpr (PIN) asm volatile("" : "+v"(r.v0), "+v"(r.v1));
    if constexpr (PV8) { r.v8[0] = vb + v8off; r.v8[1] = vb + (v8off ^ 32); }
    else r.v8[0] = r.v8[1] = 0;
    return r;
  };
  auto lds128 = [](unsigned addr) { return *(const __attribute__((address_space(3))) i32x4_t*)(uintptr_t)addr; };
  auto read_slot = [&](auto slot_t, const RdAddr& ra, i32x8& f) {
    constexpr int SLOT = decltype(slot_t)::value;
    if constexpr (SLOT < 12) {
      constexpr int kt2 = SLOT / 6, w = SLOT % 6;
      if constexpr (w < 2) {
        constexpr int pl = (w == 0 ? KO8 : KOL8) + kt2 * 2048;
        const i32x4_t x0 = lds128(ra.k8[0] + pl), x1 = lds128(ra.k8[1] + pl);
        f = (i32x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      } else {
        const i32x4_t x0 = lds128(ra.k16[w - 2] + kt2 * 4096);
        f[0] = x0[0]; f[1] = x0[1]; f[2] = x0[2]; f[3] = x0[3];
      }
    } else if constexpr (SLOT < 20) {
      constexpr int I = SLOT - 12, kt2 = I >> 2, s2 = (I >> 1) & 1, et = I & 1;
      constexpr int cst = kt2 * 4096 + s2 * 2048;
      const unsigned a0 = (et == 0 ? ra.v0 : ra.v1) + cst, a1 = a0 + 1024;     // keys + 8: swz_v16 repeats every 4 rows, so the same lane offset
      const i32x2 va = __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(uintptr_t)a0));
      const i32x2 vb2 = __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(uintptr_t)a1));
      f[0] = va[0]; f[1] = va[1]; f[2] = vb2[0]; f[3] = vb2[1];
    } else {
      constexpr int I = SLOT - 20, et = I >> 1, lo = I & 1;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const i32x2 x = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(uintptr_t)(ra.v8[et] + (lo ? OFF_VL8 : OFF_V8) + n * 1024));
        f[2 * n] = x[0]; f[2 * n + 1] = x[1];
      }
    }
  };
  // LDS-DMA lane offsets: loop-invariant except for the tail tile of a sequence that is not a multiple of 64 keys (rows past its end re-read its last row)
  const KvLaneOff<NW> lo_full = kv_lane_off<NW>(KB - 1, wave, lane);
  auto lane_off = [&](int t) -> KvLaneOff<NW> {
    if ((a.S % KB) != 0 && t == ntiles - 1) return kv_lane_off<NW>(a.S - 1 - t * KB, wave, lane);
    return lo_full;
  };

  // prologue: K(0), V(0) -> stage 0, K(1) -> stage 1; S(0)
  stage_kv1<NW, true, true, RING3>(a, head_off, 0, kslot(0), wave, lane_off(0));
  stage_kv1<NW, false, PV8, RING3>(a, head_off, 0, vslot(0), wave, lane_off(0));
  if (ntiles > 1) stage_kv1<NW, true, true, RING3>(a, head_off, 1, kslot(1), wave, lane_off(1));
  if constexpr (RING3) {
    if (ntiles > 1) stage_kv1<NW, false, PV8, RING3>(a, head_off, 1, vslot(1), wave, lane_off(1));
    if (ntiles > 2) stage_kv1<NW, true, true, RING3>(a, head_off, 2, kslot(2), wave, lane_off(2));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sa[2], sb[2];
  auto qk_mma = [&](auto slot_t, const i32x8& f, f32x16 (&sn)[2]) {
    constexpr int SLOT = decltype(slot_t)::value, kt2 = SLOT / 6, w = SLOT % 6;
    if constexpr (w == 0) sn[kt2] = mfma32_f8<e8m0(-kF8KV), e8m0(-kF8Q - kF8Lo)>(f, ql8, (f32x16){});
    else if constexpr (w == 1) sn[kt2] = mfma32_f8<e8m0(-kF8KV - kF8Lo), e8m0(-kF8Q)>(f, q8, sn[kt2]);
    else {
      const bf16x8 kh = __builtin_bit_cast(bf16x8, (i32x4_t){f[0], f[1], f[2], f[3]});
      sn[kt2] = mfma32<true>(kh, q16[w - 2], sn[kt2]);
    }
  };
  {
    i32x8 ring[RING];
    const RdAddr ra0 = rd_addr(kslot(0), vslot(0));
    [&]<int... P>(std::integer_sequence<int, P...>) { (read_slot(std::integral_constant<int, P>{}, ra0, ring[P]), ...); }(std::make_integer_sequence<int, DEPTH>{});
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        if constexpr (I + DEPTH < 12) read_slot(std::integral_constant<int, I + DEPTH>{}, ra0, ring[(I + DEPTH) % RING]);
        qk_mma(std::integral_constant<int, I>{}, ring[I % RING], sa);
      }(), ...);
    }(std::make_integer_sequence<int, 12>{});
  }
  // iteration 0 stages the next K tile into the slot K(0) was just read from: every wave has to be past its S(0) reads first
  // (the LDS-DMA lands microseconds later, so the race was never observed -- it is closed all the same)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // one iteration: softmax + PV of tile kt (scores in sc), and -- unless LAST -- the scores of tile kt + 1 into sn
  auto iter = [&](auto last_t, auto tail_t, int kt, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
    constexpr bool LAST = decltype(last_t)::value, TAIL = decltype(tail_t)::value;
    constexpr int FIRST = LAST ? 12 : 0;                  // the last tile has no next scores to compute
    const RdAddr ra = rd_addr(kslot(kt + 1), vslot(kt));   // K(kt + 1) for the next scores, V(kt) for this tile's output
    if constexpr (!LAST) {
      if constexpr (RING3) {                              // K(kt + 3) takes K(kt)'s slot, V(kt + 2) takes V(kt - 1)'s
        if (kt + 3 < ntiles) stage_kv1<NW, true, true, true>(a, head_off, kt + 3, kslot(kt), wave, lane_off(kt + 3));
        if (kt + 2 < ntiles) stage_kv1<NW, false, false, true>(a, head_off, kt + 2, vslot(kt + 2), wave, lane_off(kt + 2));
      } else {                                            // K(kt + 2) takes K(kt)'s half of the stage, V(kt + 1) V(kt - 1)'s
        if (kt + 2 < ntiles) stage_kv1<NW, true>(a, head_off, kt + 2, kslot(kt), wave, lane_off(kt + 2));
        stage_kv1<NW, false, PV8>(a, head_off, kt + 1, vslot(kt + 1), wave, lane_off(kt + 1));
      }
    }
    bf16x8 p16[4];
    i32x8 p8, pl8;
    float tmax = -1.0e30f, m_new, alpha, psum = 0.f;
    // VALU chunks: softmax of tile kt (chunks 0..11), e4m3 packing of P (chunks 12..19)
    auto valu_chunk = [&](auto c_t) {
      constexpr int C = decltype(c_t)::value;
      if constexpr (C < 2) {                       // running maximum over sub-tile C
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (TAIL) {
            const int key = kt * KB + C * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            sc[C][r] = key < a.S ? sc[C][r] : -1.0e30f;
          }
          tmax = fmaxf(tmax, sc[C][r]);
        }
        if constexpr (C == 1) {
          tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
          m_new = fmaxf(m_run, tmax);
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
        }
      } else if constexpr (C < 10) {               // exponentials of four accumulator registers, fp16 fragment halves
        constexpr int e = C - 2, kt2 = e >> 2, r4 = e & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pv = __builtin_amdgcn_exp2f(sc[kt2][4 * r4 + j] - m_new);
          sc[kt2][4 * r4 + j] = pv;
          psum += pv;
          p16[2 * kt2 + (r4 >> 1)][4 * (r4 & 1) + j] = (short)f32_to_f16(pv);
        }
      } else if constexpr (C < 12) {               // rescale of one output tile
        constexpr int et = C - 10;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          oacc[et][r] *= alpha;
        }
        if constexpr (et == 1) l_run = l_run * alpha + psum;
      } else if constexpr (C < 20 && PV8) {        // four accumulator registers -> one dword of each e4m3 operand
        constexpr int I = C - 12, c2 = I >> 2, r4 = I & 3;
        float lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = __builtin_fmaf(f16_to_f32((bf16_t)p16[2 * c2 + (r4 >> 1)][4 * (r4 & 1) + j]), -1.0f, sc[c2][4 * r4 + j]);
        p8[4 * c2 + r4] = fp8x4_scaled<kF8P>(sc[c2][4 * r4], sc[c2][4 * r4 + 1], sc[c2][4 * r4 + 2], sc[c2][4 * r4 + 3]);
        pl8[4 * c2 + r4] = fp8x4_scaled<kF8P + kF8Lo>(lo[0], lo[1], lo[2], lo[3]);
      }
    };
    auto mma = [&](auto slot_t, const i32x8& f) {
      constexpr int SLOT = decltype(slot_t)::value;
      if constexpr (SLOT < 12) qk_mma(slot_t, f, sn);
      else if constexpr (SLOT < 20) {
        constexpr int I = SLOT - 12, kt2 = I >> 2, s2 = (I >> 1) & 1, et = I & 1;
        const bf16x8 vh = __builtin_bit_cast(bf16x8, (i32x4_t){f[0], f[1], f[2], f[3]});
        oacc[et] = mfma32<true>(vh, p16[2 * kt2 + s2], oacc[et]);
      } else {
        constexpr int I = SLOT - 20, et = I >> 1, lo = I & 1;
        if constexpr (lo == 0) oacc[et] = mfma32_f8<e8m0(-kF8KV), e8m0(-kF8P - kF8Lo)>(f, pl8, oacc[et]);
        else oacc[et] = mfma32_f8<e8m0(-kF8KV - kF8Lo), e8m0(-kF8P)>(f, p8, oacc[et]);
      }
    };
    i32x8 ring[RING];
    [&]<int... P>(std::integer_sequence<int, P...>) { (read_slot(std::integral_constant<int, FIRST + P>{}, ra, ring[(FIRST + P) % RING]), ...); }(std::make_integer_sequence<int, DEPTH>{});
    if constexpr (LAST) {   // no S^T MFMAs to hide the softmax behind
      [&]<int... C>(std::integer_sequence<int, C...>) { (valu_chunk(std::integral_constant<int, C>{}), ...); }(std::make_integer_sequence<int, 12>{});
    }
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ([&] {
        constexpr int I = FIRST + J;
        if constexpr (I + DEPTH < NSLOT) read_slot(std::integral_constant<int, I + DEPTH>{}, ra, ring[(I + DEPTH) % RING]);
        if constexpr (!LAST || I >= 12) valu_chunk(std::integral_constant<int, I>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, I>{}, ring[I % RING]);
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, NSLOT - FIRST>{});
    if constexpr (RING3 && !LAST) {
      // only the loads of the PREVIOUS iteration have to have landed: this iteration issued NK (fp16 groups + e4m3 groups) K and NV V
      // LDS-DMA instructions per wave (fewer near the end of the key range)
      constexpr int NK = 8 / NW + (NW == 4 ? 2 : 1), NV = 8 / NW;
      if (kt + 3 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NK + NV) : "memory");
      else if (kt + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };
  // tiles in pairs so that the two score buffers swap roles without register copies
  int kt = 0;
  for (; kt + 2 < ntiles; kt += 2) {
    iter(std::false_type{}, std::false_type{}, kt, sa, sb);
    iter(std::false_type{}, std::false_type{}, kt + 1, sb, sa);
  }
  if (kt + 2 == ntiles) {
    iter(std::false_type{}, std::false_type{}, kt, sa, sb);
    if (a.S % KB) iter(std::true_type{}, std::true_type{}, kt + 1, sb, sa);
    else iter(std::true_type{}, std::false_type{}, kt + 1, sb, sa);
  } else {
    if (a.S % KB) iter(std::true_type{}, std::true_type{}, kt, sa, sb);
    else iter(std::true_type{}, std::false_type{}, kt, sa, sb);
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + ql;
  if (a.lse && q < a.S && half == 0) a.lse[(int64_t)bh * a.S + q] = m_run + __builtin_amdgcn_logf(l_tot);
  if (q < a.S) {
    const int64_t row = ((int64_t)b * a.S + q) * (a.H * 64) + h * 64;
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int e = 32 * et + 8 * g4 + 4 * half;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = oacc[et][4 * g4 + j] * inv;
        if (a.o_f32) {
          *reinterpret_cast<float4*>(a.o_f32 + row + e) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 h16; unsigned hi8, lo8;
          f16f8x4<kF8Act>(v, h16, hi8, lo8);
          if (a.o_ilv) { store_ilv4(a.o_ilv, row + e, h16, hi8, lo8); continue; }
          *reinterpret_cast<uint2*>(a.o16 + row + e) = h16;
          if (a.o8) *reinterpret_cast<unsigned*>(a.o8 + row + e) = hi8;
          *reinterpret_cast<unsigned*>(a.ol8 + row + e) = lo8;
        }
      }
  }
}

template <int NW, bool PV8>
int launch_pipe(const Attn8Args& a, hipStream_t s) {
  constexpr int lds = PV8 ? 2 * STAGE : RING3_BYTES;
  constexpr int QB = 32 * NW;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)attention_f16f8_pipe_kernel<NW, PV8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
  dim3 grid(((a.S + QB - 1) / QB) * a.B * a.H);
  hipLaunchKernelGGL((attention_f16f8_pipe_kernel<NW, PV8>), grid, dim3(64 * NW), lds, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

template <int QT, int NW>
int launch_t(const Attn8Args& a, hipStream_t s) {
  constexpr int lds = 2 * STAGE;
  constexpr int QB = NW * 32 * QT;
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)attention_f16f8_kernel<QT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
  dim3 grid(((a.S + QB - 1) / QB) * a.B * a.H);
  hipLaunchKernelGGL((attention_f16f8_kernel<QT, NW>), grid, dim3(64 * NW), lds, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

}  // namespace

int g_attn_shape = 0;
void awt_attn_force_shape(int v) { g_attn_shape = v; }
// must mirror the selection in launch_attention_f16f8: shapes 1 .. 5 and the training (lse) form keep P V's e4m3 cross terms
bool attention_f16f8_reads_v8(bool with_lse) { return with_lse || (g_attn_shape >= 1 && g_attn_shape <= 5); }

int launch_attention_f16f8(awt_ctx* c, const F8Planes& q, const F8Planes& k, const F8Planes& v, const F8Planes& o, float* o_f32,
                           float* lse, int B, int H, int S, hipStream_t s, char* o_ilv) {
  AWT_REQUIRE(B > 0 && H > 0 && S > 0, AWT_ERR_INVALID, "attention: bad shape");
  AWT_REQUIRE(q.p16 && q.hi8 && q.lo8 && k.p16 && k.hi8 && k.lo8 && v.p16 && v.hi8 && v.lo8, AWT_ERR_INVALID, "attention (f16f8): null plane");
  // (v.hi8 / v.lo8 are only dereferenced by the forms attention_f16f8_reads_v8() names; the encoder does not fill them otherwise)
  AWT_REQUIRE(o_f32 || o_ilv || (o.p16 && o.lo8), AWT_ERR_INVALID, "attention (f16f8): null output plane (hi8 may be null: not consumed)");
  AWT_REQUIRE(!o_ilv || (H * 64) % 64 == 0, AWT_ERR_INVALID, "attention (f16f8): split-line output needs whole 64-element line pairs per row");
  Attn8Args a{q.p16, k.p16, v.p16, q.hi8, q.lo8, k.hi8, k.lo8, v.hi8, v.lo8, o.p16, o.hi8, o.lo8, o_f32, lse, B, H, S};
  a.o_ilv = o_f32 ? nullptr : o_ilv;
  ProfScope prof(c, AWT_PROF_ATTENTION, s, 4.0 * (double)B * H * (double)S * S * 64);
  // shapes (awt_tuning_set "attn_shape"): 0 = auto; 1 = 4 waves x 32 queries; 2 = 4 waves x 64 queries; 3 = 6 waves x 32 queries
  // (three waves per SIMD: one wave's softmax VALU work runs beside the others' MFMAs); 4 / 5 = software-pipelined, 4 / 8 waves,
  // every cross term; 6 / 7 = the same with P V as one fp16 product
  const int shape = g_attn_shape;
  if (shape == 1) return launch_t<1, 4>(a, s);
  if (shape == 2) return launch_t<2, 4>(a, s);
  if (shape == 3) return launch_t<1, 6>(a, s);
  if (shape == 4) return launch_pipe<4, true>(a, s);
  if (shape == 5) return launch_pipe<8, true>(a, s);
  if (shape == 6) return launch_pipe<4, false>(a, s);
  if (shape == 7) return launch_pipe<8, false>(a, s);
  // measured at B = 64, H = 12, S = 1500 (tools/attn_bench.py, profiles/r02_attention_shapes.txt): plain 4 x 32 queries 1.28 ms,
  // 4 x 64 queries 1.44 ms (70 spilled registers), 6 x 32 queries 1.68 ms; software-pipelined 4 x 32 queries 1.16 ms, 8 x 32 queries
  // 1.19 ms; pipelined with the single fp16 product for P V (shapes 6 / 7): 0.82 / 0.86 ms.  The kernel is VALU-bound (33 v_exp_f32 +
  // ~260 other VALU instructions against 24 MFMAs per wave and key tile), so dropping the two e4m3 cross terms of P V -- their
  // P_lo = P - fp16(P) arithmetic and e4m3 packing is 40 % of that VALU work -- is what buys the time.  Cost: P and V enter that
  // product with 11 significant bits (error <= 2^-12 |v|_max per output against 2^-15); q k^T keeps its cross terms because an
  // error in a logit is amplified by exp.  Small grids take the four-wave form (twice the workgroups).
  const int64_t wg8 = (int64_t)((S + 255) / 256) * B * H;
  if (lse) return wg8 >= 256 ? launch_pipe<8, true>(a, s) : launch_pipe<4, true>(a, s);   // training keeps every cross term
  return launch_pipe<4, false>(a, s);      // 4 waves beat 8 here (0.82 vs 0.86 ms): two workgroups per CU drift apart, which is what lets VALU and MFMA phases overlap
}
