// Shared device/host helpers for libawt (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/awt.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

#define AWT_WAVE 64

// ---------------------------------------------------------------- operand precisions (awt_encoder_cfg.mfma_terms, `terms` of the operators)
//   PREC_BF16    one bf16 plane, one MFMA per fragment pair (fast, ~4e-3 rel-L2)
//   PREC_BF16X3  bf16 hi + lo planes, three MFMAs: a_hi b_lo + a_lo b_hi + a_hi b_hi  (2^-17 per operand)
//   PREC_F16X3   the same with fp16 planes (11 + 11 significant bits: 2^-23 per operand; operands must stay inside fp16's range)
//   PREC_F16F8   fp16 plane + two e4m3 planes: a_hi b_hi on the fp16 MFMA, the two cross terms a_hi b_lo + a_lo b_hi on the
//                block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the fp16 rate): two MFMA-equivalents per
//                fragment pair instead of three.  hi8 = e4m3(x 2^s), lo8 = e4m3((x - fp16(x)) 2^(s + 11)) with a FIXED
//                power-of-two s per operand role (below), so no reduction pass is needed to write a plane.
//   PREC_F16     one fp16 plane, one MFMA per fragment pair: a measurement mode (bench.py `other_precisions`): 11 significant bits per operand, which misses
//                the 1e-3 max-abs bound by ~2.5x on Whisper-small; runs on the fp16 kernels with the lo plane absent
enum { PREC_BF16 = 1, PREC_F16 = 2, PREC_BF16X3 = 3, PREC_F16X3 = 4, PREC_F16F8 = 5, PREC_F16F6 = 6 };
__host__ __device__ constexpr bool prec_is_f16(int p) { return p == PREC_F16 || p == PREC_F16X3 || p == PREC_F16F8 || p == PREC_F16F6; }
__host__ __device__ constexpr int prec_products(int p) { return (p == PREC_BF16 || p == PREC_F16) ? 1 : 3; }   // split products formed (cross terms may be fp8)
// fixed exponents of the e4m3 planes: |x| 2^s must stay <= 448 (saturates beyond); values below 2^(-6 - s) are subnormal
// (absolute error 2^(-10 - s)), which is far below the cross terms' weight in any dot product they enter
constexpr int kF8Act = -2;     // GEMM A operands (LayerNorm / attention / GELU outputs, im2col rows, LoRA u): |x| <= 1792
constexpr int kF8Wgt = 4;      // GEMM W operands: |w| <= 28
constexpr int kF8Q = 2;        // attention q (already scaled by head_dim^-1/2 log2 e): |q| <= 112
constexpr int kF8KV = 0;       // attention k, v: |x| <= 448
constexpr int kF8P = 8;        // attention probabilities exp2(s - max) in (0, 1]
constexpr int kF8Lo = 11;      // lo planes carry (x - fp16(x)) 2^(s + 11): |x - fp16(x)| <= 2^-11 |x|
__host__ __device__ constexpr int e8m0(int exp2) { return 127 + exp2; }   // E8M0 scale byte of 2^exp2

// ---------------------------------------------------------------- bf16 helpers (device)
__device__ __forceinline__ bf16_t f32_to_bf16(float x) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN; MI355X_MICROARCH correctness table)
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t b) {
  return __builtin_bit_cast(float, ((unsigned)b) << 16);
}
// hi + lo split: x ~= hi + lo with |x - hi - lo| <= 2^-17 |x|
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f32_to_bf16(x);
  lo = f32_to_bf16(x - bf16_to_f32(hi));
}
__device__ __forceinline__ unsigned pack2(bf16_t a, bf16_t b) { return (unsigned)a | ((unsigned)b << 16); }

// ---------------------------------------------------------------- fp16 / fp8 helpers (device)
__device__ __forceinline__ bf16_t f32_to_f16(float x) { _Float16 h = (_Float16)x; return __builtin_bit_cast(bf16_t, h); }   // RNE
__device__ __forceinline__ float f16_to_f32(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
// hi + lo split on a 16-bit carrier: bf16 planes (F16 = false) or fp16 planes (F16 = true)
template <bool F16>
__device__ __forceinline__ void split16(float x, bf16_t& hi, bf16_t& lo) {
  if constexpr (F16) { hi = f32_to_f16(x); lo = f32_to_f16(x - f16_to_f32(hi)); }
  else split_bf16(x, hi, lo);
}
template <bool F16>
__device__ __forceinline__ float join16(bf16_t hi, bf16_t lo) { return F16 ? f16_to_f32(hi) + f16_to_f32(lo) : bf16_to_f32(hi) + bf16_to_f32(lo); }
template <bool F16>
__device__ __forceinline__ float cvt16(bf16_t v) { return F16 ? f16_to_f32(v) : bf16_to_f32(v); }
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// ---- PREC_F16F6 (experimental, single-operator path only): like PREC_F16F8 with the two correction planes in FP6 e3m2 ("bf6", the block-scaled
// MFMA runs it at twice e4m3's rate): 32 consecutive k of a row are 24 bytes, packed by v_cvt_scalef32_2xpk16_bf6_f32 from k 0..15 and k 16..31
// of the group (probed, tools/f6_probe.hip: result = fp6(x / scale), round to nearest, saturates at +-28, element 2 m = a[m], 2 m + 1 = b[m]).
// The order inside a group is immaterial as long as both operands are packed the same way.  Fixed exponents: activations x 2^1, weights x 2^7.
constexpr int kF6Act = 1, kF6Wgt = 7;
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
template <int S>
__device__ __forceinline__ u32x6 bf6x32(const float (&v)[32]) {
  f32x16v a, b;
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = v[i]; b[i] = v[16 + i]; }
  return __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, S >= 0 ? 1.0f / (float)(1ull << (S >= 0 ? S : 0)) : (float)(1ull << (S < 0 ? -S : 0)));
}
template <int SA, int SB>
__device__ __forceinline__ f32x16 mfma32_f6(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 3, 3, 0, SA, 0, SB);
}
// weight images of PREC_F16F6: fp16 plane in w16f8_index order; each e3m2 plane fragment-major, 24 bytes per lane (row n & 31, half (k & 63) >> 5)
__host__ __device__ __forceinline__ int64_t w6_byte_index(int n, int k, int ktiles64) {
  const int nt = n >> 5, r = n & 31, kt = k >> 6, h = (k & 63) >> 5;
  return (((int64_t)nt * ktiles64 + kt) * 64 + (h * 32 + r)) * 24;
}

// block-scaled e4m3 x e4m3 product with one power-of-two scale per operand (E8M0 bytes SA, SB)
template <int SA, int SB>
__device__ __forceinline__ f32x16 mfma32_f8(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, SA, 0, SB);
}
__host__ __device__ constexpr float pow2f(int e) { return e >= 0 ? (float)(1ull << e) : 1.0f / (float)(1ull << -e); }
// four floats -> four e4m3 bytes of x 2^S (round to nearest even).  Neither conversion instruction saturates -- |x 2^S| > 464 comes out as
// NaN from v_cvt_pk_fp8_f32 and from v_cvt_scalef32_pk_fp8_f32 alike (tools/fp8_cvt_probe.hip, profiles/r03_fp8_cvt_probe.txt) -- so the
// value is clamped first, to +-448 2^-S, and the power-of-two scale is applied by the scaled conversion itself (it divides by its scale
// operand): three instructions per pair instead of five, the same bytes for every input (the probe compares the two forms over every
// rounding boundary of 33 binades).
template <int S>
__device__ __forceinline__ unsigned fp8x4(float a, float b, float c, float d) {
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  constexpr float lim = 448.0f * pow2f(-S), inv = pow2f(-S);
  a = __builtin_amdgcn_fmed3f(a, -lim, lim); b = __builtin_amdgcn_fmed3f(b, -lim, lim);
  c = __builtin_amdgcn_fmed3f(c, -lim, lim); d = __builtin_amdgcn_fmed3f(d, -lim, lim);
  s16x2_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, inv, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, inv, true);
  return __builtin_bit_cast(unsigned, r);
}
// the same with the exponent chosen at run time: inv = 2^-S, lim = 448 2^-S (the GEMM epilogue picks S per output column block: q | k, v)
__device__ __forceinline__ unsigned fp8x4_rt(float a, float b, float c, float d, float lim, float inv) {
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  a = __builtin_amdgcn_fmed3f(a, -lim, lim); b = __builtin_amdgcn_fmed3f(b, -lim, lim);
  c = __builtin_amdgcn_fmed3f(c, -lim, lim); d = __builtin_amdgcn_fmed3f(d, -lim, lim);
  s16x2_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, inv, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, inv, true);
  return __builtin_bit_cast(unsigned, r);
}
// the three planes of four consecutive f16f8 elements: fp16 bits (two dwords), hi8 dword, lo8 dword
// SAT: the value is clamped to fp16's finite range first.  The GRADIENT planes of the training step use it: their scale 2^k is chosen from the
// top-level gradient only, and an intermediate gradient more than ~2^10 above it (a LayerNorm backward through an outlier gain) would otherwise
// become inf in the fp16 plane, -inf in the residual and NaN in every adapter gradient downstream; saturated, that element is merely clipped.
constexpr float kF16Max = 65504.0f;
template <int S, bool SAT = false>
__device__ __forceinline__ void f16f8x4(const float (&vin)[4], uint2& h16, unsigned& hi8, unsigned& lo8) {
  bf16_t h[4]; float l[4], v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = SAT ? __builtin_amdgcn_fmed3f(vin[t], -kF16Max, kF16Max) : vin[t];
#pragma unroll
  for (int t = 0; t < 4; ++t) { h[t] = f32_to_f16(v[t]); l[t] = v[t] - f16_to_f32(h[t]); }
  h16 = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
  hi8 = fp8x4<S>(v[0], v[1], v[2], v[3]);
  lo8 = fp8x4<S + kF8Lo>(l[0], l[1], l[2], l[3]);
}

// 16-byte LDS-DMA (global -> LDS without a register stage; the LDS destination is the wave-uniform base + lane * 16) that hipcc does NOT see.
// With the builtin (__builtin_amdgcn_global_load_lds) the compiler knows an LDS write is in flight on the vector-memory counter and, unable to
// prove that a later ordinary LDS read does not alias it, puts `s_waitcnt vmcnt(0)` in front of the first ds_read that follows: in the attention
// kernels that was the first fragment read of the iteration, i.e. every iteration waited for the K / V tiles it had JUST requested for two
// iterations later, and the multi-stage rings bought nothing (disassembly, round 3; cdna_hip_programming.md section 5 "Three .s-level traps" (b)).
// As inline asm the copy is invisible to that bookkeeping; every kernel that uses it waits with its own counted `s_waitcnt vmcnt(N)` in front of
// the barrier that publishes the tile (vmcnt retires in issue order).  M0 (the destination base) is compiler-reserved: saved and restored here.
__device__ __forceinline__ void glds16_asm(const void* gsrc, char* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// The same with the source as a wave-uniform base (SGPR pair) plus a 32-bit per-lane byte offset and the destination as an LDS byte address:
// a loop whose lane offsets are invariant spends no VALU instruction on the copy's addresses (the 64-bit per-lane pointer of the form above
// costs a v_lshl_add_u64 or two per copy).
__device__ __forceinline__ void glds16_asm_sbase(const void* sbase, unsigned lane_byte_off, unsigned lds_wave_addr) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(sbase), "s"(dst) : "memory");
}

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26: five-term polynomial in 1 / (1 + p |x|) times exp(-x^2)) in ~12
// instructions; libm's erff costs ~3x that, and the MLP GEMM epilogue evaluates 295 M of them per layer at B = 64.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
// erf GELU (not the tanh form), as torch.nn.functional.gelu (HF:modeling_whisper.py:618-619, activation_function "gelu")
// One transcendental instead of two (v_rcp_f32 and v_exp_f32 are quarter-rate; the GELU epilogue of fc1 is the largest block of vector work in the encoder step):
// erfc(t) = 2^(-t q(t)) for t = |x| / sqrt 2 <= 4.2 with q a degree-6 polynomial -- a weighted minimax fit of -log2(erfc(t)) / t, |erf error| <= 1.6e-7 when
// evaluated in fp32 (tools/fit_erfc_exp2.py; Abramowitz-Stegun 7.1.26 above: 1.5e-7) -- continued beyond 4.2 (erfc = 2.9e-9 there) along its tangent in the
// exponent (slope 12.5 per unit of t: an upper bound of erfc that falls below 1e-30 by t = 10, so a large negative x gives -0 as it must), and
// gelu(x) = x / 2 + |x| / 2 erf(|x| / sqrt 2) = (h + |h|) - |h| erfc(t),  h = x / 2.
__device__ __forceinline__ float gelu_erf(float x) {
  const float t = fabsf(x) * 0.70710678118654752440f, tc = fminf(t, 4.2f);
  float q = fmaf(-0.00010022104106610641f, tc, 0.0004615735961124301f);
  q = fmaf(q, tc, 0.0023022345267236233f);
  q = fmaf(q, tc, -0.029452508315443993f);
  q = fmaf(q, tc, 0.14896366000175476f);
  q = fmaf(q, tc, 0.9183286428451538f);
  q = fmaf(q, tc, 1.6279137134552002f);
  const float e = __builtin_amdgcn_exp2f(fmaf(tc - t, 12.5f, -tc * q));
  const float h = 0.5f * x;
  return fmaf(-fabsf(h), e, h + fabsf(h));
}

// Fragment-major weight layout.  Every "W-side" GEMM operand is library-owned and static within a step, so it is stored
// the way v_mfma_f32_16x16x32_bf16 consumes it: for each (16-row n-tile, 32-wide k-step) one contiguous 1 KB block holding
// lane l's B fragment W[16 nt + (l & 15)][32 ks + 8 (l >> 4) + 0..7] at bytes 16 l .. 16 l + 15.  A wave then fetches a
// fragment with ONE fully coalesced 16-byte-per-lane global load straight into registers -- the weights never touch LDS
// or the LDS-DMA path, which is the GEMM's bottleneck.
__host__ __device__ __forceinline__ int64_t w_frag_index(int n, int k, int ksteps) {
  const int nt = n >> 4, r = n & 15, ks = k >> 5, kq = (k & 31) >> 3, j = k & 7;
  return (((int64_t)nt * ksteps + ks) * 64 + (kq * 16 + r)) * 8 + j;
}

// f16f8 weights: the fp16 plane in v_mfma_f32_32x32x16_f16 B-fragment order (one 1 KB block per 32-row n-tile and 16-deep
// k-step: lane 32 h + r holds W[32 nt + r][16 ks + 8 h + 0..7]) and the two e4m3 planes in the operand order of
// v_mfma_scale_f32_32x32x64_f8f6f4 (one 2 KB block per n-tile and 64-deep K-tile: lane 32 h + r holds the 32 bytes
// W[32 nt + r][64 kt + 32 h + 0..31]).  Rows are padded to a multiple of 32, K to a multiple of 64.
__host__ __device__ __forceinline__ int64_t w16f8_index(int n, int k, int ksteps16) {
  const int nt = n >> 5, r = n & 31, ks = k >> 4, h = (k & 15) >> 3, j = k & 7;
  return (((int64_t)nt * ksteps16 + ks) * 64 + (h * 32 + r)) * 8 + j;
}
__host__ __device__ __forceinline__ int64_t w8_index(int n, int k, int ktiles64) {
  const int nt = n >> 5, r = n & 31, kt = k >> 6, h = (k & 63) >> 5, b = k & 31;
  return (((int64_t)nt * ktiles64 + kt) * 64 + (h * 32 + r)) * 32 + b;
}

// The 16 x 16 MFMA form of the f16f8 GEMM (gemm_f8s_kernel) reads the fp16 plane in w_frag_index order (16-row n-tiles, fp16 values) and BOTH e4m3 planes from
// one image in the operand order of v_mfma_scale_f32_16x16x128_f8f6f4: per (16-row n-tile, 64-deep K-tile) a 2 KB block, lane 16 kb + r holding the 32 bytes
// lo8[16 nt + r][64 kt + 32 kb ..] for kb < 2 and hi8[16 nt + r][64 kt + 32 (kb - 2) ..] above (the activation operand carries hi8 | lo8 in the same block order)
__host__ __device__ __forceinline__ int64_t w8s_index(int n, int k, int ktiles64, int is_hi8) {
  const int nt = n >> 4, r = n & 15, kt = k >> 6, kb = ((k & 63) >> 5) + (is_hi8 ? 2 : 0), b = k & 31;
  return (((int64_t)nt * ktiles64 + kt) * 64 + (kb * 16 + r)) * 32 + b;
}

// An activation matrix as MFMA operand planes: p16 (bf16 or fp16) + lo16 (split modes) or + hi8 / lo8 (f16f8).
// ilv (PREC_F16F8 only, instead of the three planes): the SPLIT-LINE image the ping-pong GEMM (gemm_pp.h, FMT_F16F8S) stages by whole cache
// lines -- a dense [M][K] matrix, K % 64 == 0, as [M][K / 64] pairs of 128-byte lines: the 64 elements' fp16 (the "X" line), then their hi8 x 64 | lo8 x 64
// (the "Y" line), so that element offset `off` lives in pair off >> 6 (256 bytes) at e = off & 63: fp16 byte 2 e, hi8 byte 128 + e, lo8 byte 192 + e.
struct Act { bf16_t* p16 = nullptr; bf16_t* lo16 = nullptr; uint8_t* hi8 = nullptr; uint8_t* lo8 = nullptr; char* ilv = nullptr; };
__device__ __forceinline__ void store_ilv4(char* ilv, int64_t off, uint2 h16, unsigned hi8, unsigned lo8) {
  char* line = ilv + (off >> 6) * 256;
  const int e = (int)(off & 63);
  *reinterpret_cast<uint2*>(line + e * 2) = h16;
  *reinterpret_cast<unsigned*>(line + 128 + e) = hi8;
  *reinterpret_cast<unsigned*>(line + 192 + e) = lo8;
}
// four consecutive elements at element offset `off` (a multiple of 4) in the planes of precision PREC; S = e4m3 exponent
template <int PREC, int S = kF8Act, bool SAT = false>
__device__ __forceinline__ void store_act4(const Act& o, int64_t off, const float (&v)[4]) {
  if constexpr (PREC == PREC_F16F8) {
    uint2 h16; unsigned hi8, lo8;
    f16f8x4<S, SAT>(v, h16, hi8, lo8);
    if (o.ilv) { store_ilv4(o.ilv, off, h16, hi8, lo8); return; }
    *reinterpret_cast<uint2*>(o.p16 + off) = h16;
    if (o.hi8) *reinterpret_cast<unsigned*>(o.hi8 + off) = hi8;      // null: every consumer runs the fp16-exact-weight GEMM, which never reads this image
    *reinterpret_cast<unsigned*>(o.lo8 + off) = lo8;
  } else {
    bf16_t h[4], l[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) split16<PREC == PREC_F16X3>(v[t], h[t], l[t]);
    *reinterpret_cast<uint2*>(o.p16 + off) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    if (o.lo16) *reinterpret_cast<uint2*>(o.lo16 + off) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
  }
}

// ---------------------------------------------------------------- host-side error plumbing
void awt_set_error(const std::string& msg);
int awt_fail(int code, const std::string& msg);

#define AWT_HIP_CHECK(expr)                                                                      \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return awt_fail(AWT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

#define AWT_REQUIRE(cond, code, msg)         \
  do {                                       \
    if (!(cond)) return awt_fail(code, msg); \
  } while (0)

// ---------------------------------------------------------------- profiling (HIP events on the launch stream)
struct awt_prof_state;
struct awt_ctx {
  int device = 0;
  int prof_on = 0;
  awt_prof_state* prof = nullptr;
  // device-resident constant tables owned by the library, keyed by their parameters
  struct Table;
  Table* tables = nullptr;
  void* zeros = nullptr;   // 256 zero bytes on this device (padding rows of the conv stem's implicit GEMM)
};

// hipFuncSetAttribute is per (function, device): run `set` once per device of this process for the calling site
#define AWT_ONCE_PER_DEVICE(set)                                   \
  do {                                                             \
    static unsigned long long _done = 0;                           \
    int _dev = 0;                                                  \
    AWT_HIP_CHECK(hipGetDevice(&_dev));                            \
    if (!((_done >> (_dev & 63)) & 1ull)) {                        \
      set;                                                         \
      _done |= 1ull << (_dev & 63);                                \
    }                                                              \
  } while (0)

void awt_prof_begin(awt_ctx* c, int klass, hipStream_t s, double flops);
void awt_prof_end(awt_ctx* c, int klass, hipStream_t s);

struct ProfScope {
  awt_ctx* c; int klass; hipStream_t s;
  ProfScope(awt_ctx* c_, int k, hipStream_t s_, double flops = 0.0) : c(c_), klass(k), s(s_) {
    if (c && (c->prof_on >> klass) & 1) awt_prof_begin(c, klass, s, flops);
  }
  ~ProfScope() { if (c && (c->prof_on >> klass) & 1) awt_prof_end(c, klass, s); }
};

// ---------------------------------------------------------------- kernel launchers (defined in the .hip files)
// GEMM:  C[M, N] = sum_seg A_seg[rowmap(m), 0:K_seg] . W_seg[n, 0:K_seg]^T, operands bf16 hi (+ lo when terms == 3)
struct GemmSeg {
  const bf16_t* a_hi; const bf16_t* a_lo; int64_t lda;   // activations, K contiguous
  // weights in FRAGMENT-MAJOR order (w_frag_index below): element (n, k) of a matrix with `w_ksteps` = K_total / 32
  // k-steps; the segment starts at k-step `w_k0` of that matrix
  const bf16_t* w_hi; const bf16_t* w_lo; int w_ksteps, w_k0;
  // PREC_F16F8: a_hi / w_hi are the fp16 planes, a_lo / w_lo are unused, and these are the e4m3 planes (activations row-major
  // with the same lda, weights in w8_index order; w_ksteps / w_k0 still count 32-deep steps of the matrix / of the segment start)
  const uint8_t* a8; const uint8_t* al8; const uint8_t* w8; const uint8_t* wl8;
  int w_exact16;                                         // PREC_F16F8: every weight of the segment is exactly representable in fp16 (its lo8 image is zero)
  // ping-pong kernel (launch_gemm_pp, PREC_F16F8): the activation as split lines (Act::ilv, row pitch lda * 4 bytes) and the weight matrix in
  // the packed region image of gemm_pp.h (launch_pack_weight_pp); both null = the segment is only for the kernels above
  const char* a_ilv; const char* w_pp;
  // 16 x 16 MFMA form (gemm_f8s_kernel, PREC_F16F8): the same weight matrix as 16-row fragment-major copies -- fp16 in w_frag_index order, the two e4m3
  // planes interleaved per (16-row n-tile, 64-deep K-tile) block in w8s_index order (gemm.hip); null = the 32 x 32 kernels
  const bf16_t* ws16; const uint8_t* ws8;
  int ws_rows;        // rows the two copies are allocated for (a multiple of 256 >= N: the 128 x 256 tiles of a matrix with N % 256 != 0 read fragment rows beyond N)
  int K;                                                 // multiple of the kernel's BK
  // source row of output row m:  (m / rows_out) * rows_in + (m % rows_out) * row_mul + row_add ; rows outside
  // [0, rows_in) of their group read as zeros (the conv stem's padding).  Plain GEMM: rows_out = rows_in = M.
  int rows_out, rows_in, row_mul, row_add;
};

enum GemmEpilogue {
  EPI_F32 = 0,        // out_f32[m, n] = acc + bias                        (ldo)
  EPI_F32_RESID = 1,  // out_f32[m, n] = resid[m, n] + acc + bias          (in place allowed)
  EPI_BF16 = 2,       // out hi/lo [m, n] = (acc + bias) * scale
  EPI_BF16_GELU = 3,  // out hi/lo [m, n] = gelu(acc + bias)
  EPI_QKV = 4,        // head-major q|k|v planes, q columns scaled by `scale`
  EPI_F32_GELU_POS = 5,  // out_f32[m, n] = gelu(acc + bias) + pos[m % rows_pos, n]   (conv2 + positional table)
  EPI_BF16_GELU_SAVE = 6,  // as EPI_BF16_GELU, and the pre-activation acc + bias is also written to hi2 / lo2 (training)
  EPI_BF16_DGELU = 7       // out hi/lo [m, n] = acc * gelu'(pre[m, n])  with pre read from pre_hi / pre_lo  (backward)
};

struct GemmOut {
  float* f32; const float* resid; int64_t ldo;
  bf16_t* hi; bf16_t* lo;                 // lo may be null when terms == 1
  uint8_t* hi8; uint8_t* lo8;             // PREC_F16F8: the e4m3 planes of the output (hi = its fp16 plane); same offsets as hi
  char* ilv;                              // PREC_F16F8, EPI_BF16 / EPI_BF16_GELU of launch_gemm_pp: the output as split lines instead (Act::ilv; dense [M][ldo])
  bf16_t* hi2; bf16_t* lo2;               // EPI_BF16_GELU_SAVE: pre-activation planes (same ldo)
  const bf16_t* pre_hi; const bf16_t* pre_lo;   // EPI_BF16_DGELU: saved pre-activation planes (same ldo)
  const float* bias;                      // [N] or null
  const float* pos; int rows_pos;         // EPI_F32_GELU_POS
  float scale;                            // EPI_BF16: all columns; EPI_QKV: q columns
  int n_valid;                            // columns >= n_valid are not stored (N padded up to the tile)
  int S, H;                               // EPI_QKV: m = b * S + s ; plane stride = B*H*S*64
  int64_t plane_stride;
  int skip_v8;                            // EPI_QKV, f16f8: do not write the e4m3 images of v (the single-product P V attention reads only its fp16 plane)
};

// `prec`: PREC_* of the operands (and of plane outputs)
int launch_gemm(awt_ctx* c, int M, int N, const GemmSeg* segs, int nseg, int prec, GemmEpilogue epi, const GemmOut& out,
                hipStream_t s);
// The persistent 256 x 256 ping-pong kernel (gemm_pp.h) for one plain K segment in PREC_F16F8: seg.a_ilv / seg.w_pp, N % 256 == 0, K % 64 == 0, K >= 128,
// the activation buffer readable for ceil(M / 256) * 256 rows.  Epilogues: EPI_F32, EPI_F32_RESID, EPI_BF16, EPI_BF16_GELU, EPI_QKV.
bool gemm_pp_supported(int M, int N, int K, int epi);
int gemm_pp_slots();                                // persistent workgroups of a ping-pong launch = CUs of the current device
int launch_gemm_pp(awt_ctx* c, int M, int N, const GemmSeg& seg, GemmEpilogue epi, const GemmOut& out, hipStream_t s);
size_t gemm_pp_weight_bytes(int N, int K);          // packed image of an [N, K] weight (N padded to 256)
// dst rows row_off .. row_off + N - 1 of a packed [Ntot, K] image = src [N, K] fp32 (f16f8 planes, kF8Wgt exponents); padding rows stay as allocated (zero)
int launch_pack_weight_pp(awt_ctx* c, const float* src, int N, int K, int row_off, char* dst, hipStream_t s);
int awt_gemm_pp_mode();                             // tuning knob "gemm_pp": 0 off (default), 1 automatic, 2 wherever supported
void awt_gemm_set_pp_mode(int v);
void awt_gemm_set_pp_stagger(int v);
void awt_gemm_set_mfma16(int v);                    // tuning knob "gemm_mfma16": 1 (default) = the 16 x 16 MFMA form of the f16f8 GEMM wherever its weight copies exist, 0 = off

// out_f32 (the final layer_norm) or operand planes of precision `prec`
int launch_layernorm(awt_ctx* c, const float* x, const float* gamma, const float* beta, int M, int d, float eps,
                     float* out_f32, const Act& out, int prec, hipStream_t s);
// q, k, v planes: bf16 [B, H, S, 64] (lo may be null for terms == 1); o: bf16 [B*S, H*64] hi/lo
int launch_attention(awt_ctx* c, const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* k_hi, const bf16_t* k_lo,
                     const bf16_t* v_hi, const bf16_t* v_lo, bf16_t* o_hi, bf16_t* o_lo, float* o_f32, float* lse, int B, int H,
                     int S, int terms, hipStream_t s);
int launch_split_f32(awt_ctx* c, const float* x, int64_t n, float scale, bf16_t* hi, bf16_t* lo, hipStream_t s);
// precision-aware form: PREC_BF16 / PREC_BF16X3 / PREC_F16X3 write p16 (+ lo16); PREC_F16F8 writes p16 (fp16), hi8 = e4m3(x 2^f8_exp),
// lo8 = e4m3((x - fp16(x)) 2^(f8_exp + 11))
struct F8Planes { bf16_t* p16; uint8_t* hi8; uint8_t* lo8; };
int launch_split_planes(awt_ctx* c, const float* x, int64_t n, float scale, int prec, int f8_exp, bf16_t* p16, bf16_t* lo16, uint8_t* hi8,
                        uint8_t* lo8, hipStream_t s, char* ilv = nullptr);   // ilv (PREC_F16F8, f8_exp == kF8Act): split lines instead of the planes
int launch_attention_f16f8(awt_ctx* c, const F8Planes& q, const F8Planes& k, const F8Planes& v, const F8Planes& o, float* o_f32,
                           float* lse, int B, int H, int S, hipStream_t s, char* o_ilv = nullptr);   // o_ilv: the output as split lines (Act::ilv) instead of o
// weights: dst(row_off + n, col_off + k) = scale * src[n, c, dt], k = dt * C + c (taps = 1: plain [N, C]); dst is a
// fragment-major matrix with ld / 32 k-steps (ld = its K, a multiple of 32; its row count a multiple of 16)
// prec PREC_F16F8: hi = fp16 plane (w16f8_index order), lo = the hi8 plane, lo8 = the lo8 plane (w8_index order); else hi / lo in
// w_frag_index order (bf16, or fp16 for PREC_F16X3) and lo8 unused
// PREC_F16F6 (single-operator path): x [M, K] -> fp16 plane + two e3m2 planes (row pitch K * 3 / 4 bytes); w [N, K] -> fragment-major images
int launch_split_planes_f6(awt_ctx* c, const float* x, int M, int K, bf16_t* p16, uint8_t* hi6, uint8_t* lo6, hipStream_t s);
int launch_pack_weight_f6(awt_ctx* c, const float* w, int N, int K, bf16_t* w16, uint8_t* hi6, uint8_t* lo6, hipStream_t s);
int launch_pack_weight(awt_ctx* c, const float* src, int N, int C, int taps, int64_t ld, int row_off, int col_off, float scale,
                       bf16_t* hi, bf16_t* lo, uint8_t* lo8, int prec, hipStream_t s, int* inexact = nullptr,    // inexact (PREC_F16F8, device int): set to 1 when a weight is not fp16-exact
                       bf16_t* s16 = nullptr, uint8_t* s8 = nullptr);                                            // PREC_F16F8: also the 16-row copies (w_frag_index / w8s_index) when given
// conv1 im2col: mel f32 [B, C, T] -> A [B*T, K_dst] bf16 hi/lo, k = dt * C + c reads mel[b, c, t + dt - 1]
int launch_im2col_conv1(awt_ctx* c, const float* mel, int B, int C, int T, int K_dst, const Act& out, int prec, hipStream_t s);

int logmel_whisper_impl(awt_ctx* c, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                        int max_valid, int B, int n_frames_out, int n_mels, float* out, void* workspace, size_t ws_bytes, hipStream_t s);
int logmel_generic_impl(awt_ctx* c, const float* pcm, int64_t pcm_stride, int B, int n_samples, int sample_rate, int n_fft,
                        int hop, int n_mels, float f_min, float f_max, float log_eps, float* out, hipStream_t s);
int prepare_waveform_impl(awt_ctx* c, const void* pcm, int pcm_is_i16, int channels, int64_t channel_stride, int64_t sample_stride,
                          int n_in, int sr_in, int sr_out, float* out, int n_out, hipStream_t s);
int64_t resampled_length(int n_in, int sr_in, int sr_out);
int logmel_prepare_impl(awt_ctx* c, int n_fft, int n_mels, double f_min, double f_max, int sample_rate, int slaney);
int resample_prepare_impl(awt_ctx* c, int sr_in, int sr_out);
void awt_free_tables(awt_ctx* c);
void awt_attn_force_shape(int v);   // f16f8 attention workgroup shape: 0 auto, 1 / 2 / 3 (attention_f8.hip)
void awt_gemm_set_gm(int v);       // row panels per tile group of the GEMM tile order (0 = default)
bool attention_f16f8_reads_v8(bool with_lse);   // whether the f16f8 attention kernel that would be selected now stages the e4m3 images of v
void awt_gemm_force_tile(int t);  // 0 auto, 64 / 128 / 256: tuning / tests (awt_tuning_set)

// ---- backward-pass launchers
int launch_attention_bwd(awt_ctx* c, const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* k_hi, const bf16_t* k_lo,
                         const bf16_t* v_hi, const bf16_t* v_lo, const bf16_t* o_hi, const bf16_t* o_lo, const bf16_t* do_hi,
                         const bf16_t* do_lo, const float* lse2, float* delta, bf16_t* g_hi, bf16_t* g_lo, int B, int H, int S,
                         float qscale, int terms, int grad_terms, hipStream_t s);
// dx = dres + LayerNorm_backward(dy; x, gamma)  (fp32), plus bf16 hi/lo planes of dx for the next GEMM; dres may be null
int launch_layernorm_bwd(awt_ctx* c, const float* dy, const float* x, const float* gamma, const float* dres, int M, int d, float eps,
                         float* dx, bf16_t* dx_hi, bf16_t* dx_lo, hipStream_t s, float gscale = 1.0f, const Act* out8 = nullptr);   // gscale multiplies dy; out8: dx as f16f8 planes too
int launch_transpose_f32(awt_ctx* c, const float* src, int N, int C, float* dst, hipStream_t s);   // dst [C, N] = src [N, C]^T
// dst(row_off + c, col_off + n) = scale * src[n, c]   (transposed copy of an [N, C] fp32 matrix into fragment-major planes)
int launch_pack_weight_t(awt_ctx* c, const float* src, int N, int C, int64_t ld, int row_off, int col_off, float scale, bf16_t* hi,
                         bf16_t* lo, hipStream_t s);
// out[j * sj + n * sn] (+)= scale * sum_m X[m, xcol + j] * Y[m, ycol + n]   for j < r, n < ny  (LoRA dA / dB; fp32 result)
int launch_outer_reduce(awt_ctx* c, const bf16_t* x_hi, const bf16_t* x_lo, int64_t ldx, int xcol, int r, const bf16_t* y_hi,
                        const bf16_t* y_lo, int64_t ldy, int ycol, int ny, int M, float scale, float* out, int64_t sj, int64_t sn,
                        float* partial, size_t partial_bytes, int accumulate, hipStream_t s);
size_t outer_reduce_partial_bytes(int M, int r, int ny);
