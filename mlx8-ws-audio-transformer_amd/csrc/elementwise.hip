// HBM-bound row kernels of the encoder: LayerNorm (K8), fp32 -> split-bf16 planes, weight packing, conv1 im2col.
#include "common.h"
#include "gemm_pp.h"

#ifndef AWT_LN_NT_LOAD
#define AWT_LN_NT_LOAD 1   // -0.3 .. -0.4 ms per encoder step, most of it in the GEMM that follows (profiles/r03_gemm_experiments.txt)
#endif

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row, the row lives in registers (d <= 1280): two-pass mean / variance in fp32 exactly like
// torch.nn.functional.layer_norm (eps 1e-5, biased variance; HF:modeling_whisper.py:392,402,642), then affine.
// Output is either fp32 (final layer_norm -> last_hidden_state) or bf16 hi (+ lo) planes for the next GEMM.
constexpr int kLnMaxChunks = 5;  // float4 chunks per lane: d <= 64 * 4 * 5 = 1280 (Whisper large)

template <int PREC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, int d, float eps,
                                                        float* out_f32, Act out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunk = d >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  float4 v[kLnMaxChunks];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
#if AWT_LN_NT_LOAD   // the residual stream is read once here: a streaming load keeps the planes this kernel writes (the next GEMM's operand) cache-resident
    if (c < nchunk) { const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + c)); v[i] = make_float4(t[0], t[1], t[2], t[3]); }
    else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
    v[i] = c < nchunk ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + dd * dd);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)d + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) continue;
    const float4 g = g4[c], bb = b4[c];
    float y[4] = {(v[i].x - mean) * rstd * g.x + bb.x, (v[i].y - mean) * rstd * g.y + bb.y,
                  (v[i].z - mean) * rstd * g.z + bb.z, (v[i].w - mean) * rstd * g.w + bb.w};
    if (out_f32) reinterpret_cast<float4*>(out_f32 + (int64_t)row * d)[c] = make_float4(y[0], y[1], y[2], y[3]);
    else store_act4<PREC>(out, (int64_t)row * d + 4 * c, y);
  }
}

// ------------------------------------------------------------------------------------------------ fp32 -> hi / lo planes
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, int64_t n4, float scale, bf16_t* hi, bf16_t* lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    bf16_t h[4], l[4];
    split_bf16(v.x * scale, h[0], l[0]); split_bf16(v.y * scale, h[1], l[1]); split_bf16(v.z * scale, h[2], l[2]); split_bf16(v.w * scale, h[3], l[3]);
    reinterpret_cast<uint2*>(hi)[i] = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    if (lo) reinterpret_cast<uint2*>(lo)[i] = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
  }
}

template <int PREC>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int64_t n4, float scale, float f8_scale, bf16_t* p16,
                                                           bf16_t* lo16, uint8_t* hi8, uint8_t* lo8, char* ilv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 q = reinterpret_cast<const float4*>(x)[i];
    const float v[4] = {q.x * scale, q.y * scale, q.z * scale, q.w * scale};
    bf16_t h[4], l[4];
    if constexpr (PREC == PREC_F16F8) {
      float lo[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { h[t] = f32_to_f16(v[t]); lo[t] = (v[t] - f16_to_f32(h[t])) * (f8_scale * pow2f(kF8Lo)); }
      const uint2 h16 = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
      const unsigned b_hi = fp8x4<0>(v[0] * f8_scale, v[1] * f8_scale, v[2] * f8_scale, v[3] * f8_scale), b_lo = fp8x4<0>(lo[0], lo[1], lo[2], lo[3]);
      if (ilv) { store_ilv4(ilv, i * 4, h16, b_hi, b_lo); continue; }
      reinterpret_cast<uint2*>(p16)[i] = h16;
      reinterpret_cast<unsigned*>(hi8)[i] = b_hi;
      reinterpret_cast<unsigned*>(lo8)[i] = b_lo;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) split16<PREC == PREC_F16X3>(v[t], h[t], l[t]);
      reinterpret_cast<uint2*>(p16)[i] = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
      if (lo16) reinterpret_cast<uint2*>(lo16)[i] = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight packing
// dst(row_off + n, col_off + k) = src[n, c, dt] * scale with k = dt * C + c  (conv weights [N, C, taps] -> implicit-GEMM
// rows; taps = 1 is a plain [N, C] linear weight), written in the fragment-major layout of w_frag_index().  Only the
// N x (C * taps) region is written: destination buffers are zero-initialised at creation, which provides every padding
// row / column.
template <int PREC>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, int N, int C, int taps, int64_t ld,
                                                          int row_off, int col_off, float scale, bf16_t* hi, bf16_t* lo, uint8_t* lo8,
                                                          int* inexact, bf16_t* s16, uint8_t* s8) {
  const int K = C * taps;
  bool any_inexact = false;       // PREC_F16F8: some weight is not exactly representable in fp16 (its lo8 image is non-zero)
  const int64_t total = (int64_t)N * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / K), k = (int)(i - (int64_t)n * K);
    const int dt = k / C, c = k - dt * C;
    const float v = src[((int64_t)n * C + c) * taps + dt] * scale;
    if constexpr (PREC == PREC_F16F8) {
      const bf16_t h = f32_to_f16(v);
      hi[w16f8_index(row_off + n, col_off + k, (int)(ld >> 4))] = h;
      const int64_t o8 = w8_index(row_off + n, col_off + k, (int)(ld >> 6));
      reinterpret_cast<uint8_t*>(lo)[o8] = (uint8_t)(fp8x4<kF8Wgt>(v, 0.f, 0.f, 0.f) & 0xFF);
      lo8[o8] = (uint8_t)(fp8x4<kF8Wgt + kF8Lo>(v - f16_to_f32(h), 0.f, 0.f, 0.f) & 0xFF);
      if (s16) {   // the 16-row copies of the 16 x 16 MFMA form (gemm_f8s_kernel)
        s16[w_frag_index(row_off + n, col_off + k, (int)(ld >> 5))] = h;
        s8[w8s_index(row_off + n, col_off + k, (int)(ld >> 6), 1)] = reinterpret_cast<uint8_t*>(lo)[o8];
        s8[w8s_index(row_off + n, col_off + k, (int)(ld >> 6), 0)] = lo8[o8];
      }
      any_inexact |= (v != f16_to_f32(h));
    } else {
      bf16_t h, l; split16<PREC == PREC_F16X3>(v, h, l);
      if (PREC == PREC_F16X3) any_inexact |= ((l & 0x7FFF) != 0);        // a non-zero lo element: the weight is not exactly fp16
      const int64_t o = w_frag_index(row_off + n, col_off + k, (int)(ld >> 5));
      hi[o] = h;
      if (lo) lo[o] = l;
    }
  }
  if ((PREC == PREC_F16F8 || PREC == PREC_F16X3) && inexact && __builtin_amdgcn_ballot_w64(any_inexact) != 0 && (threadIdx.x & 63) == 0) atomicOr(inexact, 1);
}

// The packed weight image of the ping-pong GEMM (gemm_pp.h, FMT_F16F8S): per (256-column tile, K-tile, half) one 16 KB region in LDS image order; the K-tiles
// alternate between the X line (fp16 x 64) and the Y line (lo8 x 64 | hi8 x 64) of 64 consecutive k, each row's 16-byte chunks XOR-swizzled by the row.
// One thread per four consecutive k of a row.
__global__ __launch_bounds__(256) void pack_weight_pp_kernel(const float* __restrict__ src, int N, int K, int row_off, char* dst) {
  const int nk = K >> 5;
  const int64_t total = (int64_t)N * (K >> 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / (K >> 2)), k = (int)(i - (int64_t)n * (K >> 2)) * 4;
    const float4 q = *reinterpret_cast<const float4*>(src + (int64_t)n * K + k);
    const float v[4] = {q.x, q.y, q.z, q.w};
    uint2 h16; unsigned hi8, lo8;
    f16f8x4<kF8Wgt>(v, h16, hi8, lo8);
    const int row = row_off + n, bn = row >> 8, nin = row & 255, sgrp = (nin >> 5) & 1, grp = k >> 6, e = k & 63;
    char* rx = dst + pp::w_region_offset(bn, 2 * grp, sgrp, nk);
    char* ry = dst + pp::w_region_offset(bn, 2 * grp + 1, sgrp, nk);
    *reinterpret_cast<uint2*>(rx + pp::w_row_offset(nin, e >> 3) + (e & 7) * 2) = h16;
    *reinterpret_cast<unsigned*>(ry + pp::w_row_offset(nin, e >> 4) + (e & 15)) = lo8;
    *reinterpret_cast<unsigned*>(ry + pp::w_row_offset(nin, 4 + (e >> 4)) + (e & 15)) = hi8;
  }
}

// ------------------------------------------------------------------------------------------------ conv1 im2col
// mel fp32 [B, C, T] -> A [B * T, K_dst] bf16 planes with A[(b, t), dt * C + c] = mel[b, c, t + dt - 1] (zero outside
// the clip: Conv1d padding = 1, HF:modeling_whisper.py:566).  A 64-frame slab of the clip is staged in LDS with
// reads coalesced along t; rows are written as whole 16-byte groups so each row is one contiguous K_dst * 2 B store.
constexpr int kImTile = 64;
template <int PREC>
__global__ __launch_bounds__(256) void im2col_conv1_kernel(const float* __restrict__ mel, int C, int T, int K_dst, Act out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);   // [C][kImTile + 2], pitch kImTile + 3 (odd: conflict-free column reads)
  const int pitch = kImTile + 3;
  const int b = blockIdx.y, t0 = blockIdx.x * kImTile;
  const float* mb = mel + (int64_t)b * C * T;
  for (int i = threadIdx.x; i < C * (kImTile + 2); i += 256) {
    const int c = i / (kImTile + 2), tt = i - c * (kImTile + 2);
    const int t = t0 + tt - 1;
    tile[c * pitch + tt] = (t >= 0 && t < T) ? mb[(int64_t)c * T + t] : 0.f;
  }
  __syncthreads();
  const int groups = K_dst / 8;
  const int cg = C / 8;  // C is a multiple of 8 (80), so a group of 8 never straddles a tap
  for (int i = threadIdx.x; i < kImTile * groups; i += 256) {
    const int tl = i / groups, g = i - tl * groups;
    const int t = t0 + tl;
    if (t >= T) continue;
    const int dt = g / cg, c0 = (g - dt * cg) * 8;
    float v[2][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j >> 2][j & 3] = dt < 3 ? tile[(c0 + j) * pitch + tl + dt] : 0.f;
    const int64_t off = ((int64_t)b * T + t) * K_dst + g * 8;
    store_act4<PREC>(out, off, v[0]);
    store_act4<PREC>(out, off + 4, v[1]);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd.
// Statistics are recomputed from the saved LayerNorm input (fp32), one wave per row, row in registers.
#ifndef AWT_LNB_NT_LOAD
#define AWT_LNB_NT_LOAD 1   // LayerNorm class of the fine-tune step 8.94 -> 8.45 ms (profiles/r03_gemm_experiments.txt)
#endif
// a 16-byte load of data this kernel reads exactly once (streaming when AWT_LNB_NT_LOAD)
__device__ __forceinline__ float4 ldg_once(const float4* p) {
#if AWT_LNB_NT_LOAD
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(t[0], t[1], t[2], t[3]);
#else
  return *p;
#endif
}
// gscale multiplies dy (the power-of-two gradient scale enters at the final LayerNorm); out8: dx also as f16f8 operand planes (the next consumer is
// an f16f8 GEMM: awt_encoder_cfg.backward_terms = 5)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* dres, int M, int d,
                                                            float eps, float* dx, bf16_t* dx_hi, bf16_t* dx_lo, float gscale, Act out8) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunk = d >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  const float4* dyr = reinterpret_cast<const float4*>(dy + (int64_t)row * d);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float4 v[kLnMaxChunks], g[kLnMaxChunks];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < nchunk ? ldg_once(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)d + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nchunk) {
      const float4 gy = ldg_once(dyr + c), gm = g4[c];
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;        // xhat
      g[i] = make_float4(gy.x * gscale * gm.x, gy.y * gscale * gm.y, gy.z * gscale * gm.z, gy.w * gscale * gm.w);
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  const float c1 = s1 / (float)d, c2 = s2 / (float)d;
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) continue;
    float y[4] = {rstd * (g[i].x - c1 - v[i].x * c2), rstd * (g[i].y - c1 - v[i].y * c2),
                  rstd * (g[i].z - c1 - v[i].z * c2), rstd * (g[i].w - c1 - v[i].w * c2)};
    if (dres) {
      const float4 r = ldg_once(reinterpret_cast<const float4*>(dres + (int64_t)row * d) + c);
      y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
    }
    reinterpret_cast<float4*>(dx + (int64_t)row * d)[c] = make_float4(y[0], y[1], y[2], y[3]);
    if (dx_hi) {
      bf16_t hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf16(y[j], hi[j], lo[j]);
      reinterpret_cast<uint2*>(dx_hi + (int64_t)row * d)[c] = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
      if (dx_lo) reinterpret_cast<uint2*>(dx_lo + (int64_t)row * d)[c] = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
    }
    if (out8.p16) store_act4<PREC_F16F8, kF8Act, true>(out8, (int64_t)row * d + 4 * c, y);   // gradient planes saturate instead of overflowing (common.h f16f8x4)
  }
}

// dst [C, N] = src [N, C]^T (fp32; the f16f8 copies of W^T for the backward GEMMs are packed from it)
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int N, int C, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    tile[r][tx] = (n < N && c < C) ? src[(int64_t)n * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, n = n0 + tx;
    if (c < C && n < N) dst[(int64_t)c * N + n] = tile[tx][r];
  }
}

// ------------------------------------------------------------------------------------------------ transposed weight copy
__global__ __launch_bounds__(256) void pack_weight_t_kernel(const float* __restrict__ src, int N, int C, int64_t ld, int row_off,
                                                            int col_off, float scale, bf16_t* hi, bf16_t* lo) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    tile[r][tx] = (n < N && c < C) ? src[(int64_t)n * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, n = n0 + tx;
    if (c < C && n < N) {
      bf16_t h, l; split_bf16(tile[tx][r] * scale, h, l);
      const int64_t o = w_frag_index(row_off + c, col_off + n, (int)(ld >> 5));
      hi[o] = h;
      if (lo) lo[o] = l;
    }
  }
}

// ------------------------------------------------------------------------------------------------ LoRA gradient reductions
// partial[wg][j][n] = sum over the workgroup's 256-row slab of X[m, xcol + j] * Y[m, ycol + n]   (fp32 VALU; the products are
// tiny next to the GEMMs: 2 M r d FLOP per adapter matrix).  X is the skinny operand (r <= 32 columns), staged in LDS;
// each thread owns 4 consecutive Y columns and r x 4 accumulators.  A second kernel sums the slabs in a fixed order,
// so the gradients are bit-reproducible run to run.
constexpr int kOrRows = 256;
constexpr int kOrMaxBlocks = 512;   // workgroups walk the 256-row slabs grid-strided, so the second pass sums <= 512 partials
template <int R>
__global__ __launch_bounds__(256) void outer_reduce_kernel(const bf16_t* x_hi, const bf16_t* x_lo, int64_t ldx, int xcol,
                                                           const bf16_t* y_hi, const bf16_t* y_lo, int64_t ldy, int ycol, int ny, int M,
                                                           float* partial) {
  __shared__ float xs[kOrRows][R];
  const int n = threadIdx.x * 4;
  float acc[R][4];
#pragma unroll
  for (int j = 0; j < R; ++j) { acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f; }
  const int nslab = (M + kOrRows - 1) / kOrRows;
  for (int slab = blockIdx.x; slab < nslab; slab += gridDim.x) {   // fixed slab -> workgroup map: the sum order is reproducible
    const int m0 = slab * kOrRows;
    const int rows = min(kOrRows, M - m0);
    __syncthreads();
    for (int i = threadIdx.x; i < rows * R; i += 256) {
      const int mm = i / R, j = i - mm * R;
      const int64_t o = (int64_t)(m0 + mm) * ldx + xcol + j;
      xs[mm][j] = bf16_to_f32(x_hi[o]) + (x_lo ? bf16_to_f32(x_lo[o]) : 0.f);
    }
    __syncthreads();
    if (n < ny) {
      auto load_y = [&](int mm, float (&y)[4]) {
        const int64_t o = (int64_t)(m0 + mm) * ldy + ycol + n;
        const uint2 yh = *reinterpret_cast<const uint2*>(y_hi + o);
        uint2 yl = make_uint2(0u, 0u);
        if (y_lo) yl = *reinterpret_cast<const uint2*>(y_lo + o);
        y[0] = bf16_to_f32((bf16_t)(yh.x & 0xFFFF)) + bf16_to_f32((bf16_t)(yl.x & 0xFFFF));
        y[1] = bf16_to_f32((bf16_t)(yh.x >> 16)) + bf16_to_f32((bf16_t)(yl.x >> 16));
        y[2] = bf16_to_f32((bf16_t)(yh.y & 0xFFFF)) + bf16_to_f32((bf16_t)(yl.y & 0xFFFF));
        y[3] = bf16_to_f32((bf16_t)(yh.y >> 16)) + bf16_to_f32((bf16_t)(yl.y >> 16));
      };
      auto fma_row = [&](int mm, const float (&y)[4]) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const float xv = xs[mm][j];
          acc[j][0] += xv * y[0]; acc[j][1] += xv * y[1]; acc[j][2] += xv * y[2]; acc[j][3] += xv * y[3];
        }
      };
      int mm = 0;
      for (; mm + 4 <= rows; mm += 4) {       // four rows of Y in flight per thread: the loop is bound by HBM latency, not FMAs
        float y0[4], y1[4], y2[4], y3[4];
        load_y(mm, y0); load_y(mm + 1, y1); load_y(mm + 2, y2); load_y(mm + 3, y3);
        fma_row(mm, y0); fma_row(mm + 1, y1); fma_row(mm + 2, y2); fma_row(mm + 3, y3);   // row order kept: the sum is reproducible
      }
      for (; mm < rows; ++mm) { float y[4]; load_y(mm, y); fma_row(mm, y); }
    }
  }
  if (n < ny) {
    float* p = partial + (int64_t)blockIdx.x * R * ny;
#pragma unroll
    for (int j = 0; j < R; ++j) *reinterpret_cast<float4*>(p + (int64_t)j * ny + n) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
}

// Second pass: out[j, n] = scale * sum over slabs of partial[slab][j][n].  A workgroup takes 32 (j, n) pairs; its 8 groups of 32 threads each
// sum every eighth slab, four loads in flight, and the eight partial sums are combined in a fixed tree (deterministic; a single
// thread walking all ~375 slabs was latency-bound: 89 us per call, 4.3 ms of the fine-tune step).
__global__ __launch_bounds__(256) void outer_reduce_final_kernel(const float* __restrict__ partial, int nslab, int R, int r, int ny, float scale,
                                                                 float* out, int64_t sj, int64_t sn, int accumulate) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + col;
  const bool live = idx < r * ny;
  const int j = live ? idx / ny : 0, n = live ? idx - j * ny : 0;
  const float* p = partial + (int64_t)j * ny + n;
  const int64_t step = (int64_t)R * ny;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = grp;
  for (; s + 24 < nslab; s += 32) {
    a0 += p[(int64_t)s * step]; a1 += p[(int64_t)(s + 8) * step]; a2 += p[(int64_t)(s + 16) * step]; a3 += p[(int64_t)(s + 24) * step];
  }
  for (; s < nslab; s += 8) a0 += p[(int64_t)s * step];
  red[grp][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && live) {
    const float acc = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
    float* o = out + j * sj + n * sn;
    *o = accumulate ? *o + acc * scale : acc * scale;     // accumulate: gradient accumulation over micro-batches
  }
}


#ifdef AWT_EXPERIMENTAL_F6
// ------------------------------------------------------------------------------------------------ PREC_F16F6 operand images (experimental)
// One thread per (row, group of 32 consecutive k): fp16 plane + the two e3m2 planes.  WEIGHT: fragment-major images, exponent kF6Wgt.
template <bool WEIGHT>
__global__ __launch_bounds__(256) void planes_f6_kernel(const float* __restrict__ x, int M, int K, bf16_t* p16, uint8_t* hi6, uint8_t* lo6) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int groups = K >> 5;
  if (idx >= (int64_t)M * groups) return;
  const int m = (int)(idx / groups), g = (int)(idx - (int64_t)m * groups);
  const float* src = x + (int64_t)m * K + 32 * g;
  float v[32], lo[32];
  bf16_t h[32];
#pragma unroll
  for (int i = 0; i < 32; i += 4) { const float4 t = *reinterpret_cast<const float4*>(src + i); v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w; }
#pragma unroll
  for (int i = 0; i < 32; ++i) { h[i] = f32_to_f16(v[i]); lo[i] = v[i] - f16_to_f32(h[i]); }
  constexpr int S = WEIGHT ? kF6Wgt : kF6Act;
  const u32x6 qh = bf6x32<S>(v), ql = bf6x32<S + kF8Lo>(lo);
  uint8_t *dh, *dl;
  if (WEIGHT) {
    const int64_t o = w6_byte_index(m, 32 * g, K >> 6);
    dh = hi6 + o; dl = lo6 + o;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8)
      *reinterpret_cast<uint4*>(p16 + w16f8_index(m, 32 * g + 8 * c8, K >> 4)) =
          make_uint4(pack2(h[8 * c8], h[8 * c8 + 1]), pack2(h[8 * c8 + 2], h[8 * c8 + 3]), pack2(h[8 * c8 + 4], h[8 * c8 + 5]), pack2(h[8 * c8 + 6], h[8 * c8 + 7]));
  } else {
    const int64_t o = (int64_t)m * (K / 4 * 3) + 24 * g;
    dh = hi6 + o; dl = lo6 + o;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8)
      *reinterpret_cast<uint4*>(p16 + (int64_t)m * K + 32 * g + 8 * c8) =
          make_uint4(pack2(h[8 * c8], h[8 * c8 + 1]), pack2(h[8 * c8 + 2], h[8 * c8 + 3]), pack2(h[8 * c8 + 4], h[8 * c8 + 5]), pack2(h[8 * c8 + 6], h[8 * c8 + 7]));
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    *reinterpret_cast<uint2*>(dh + 8 * i) = make_uint2(qh[2 * i], qh[2 * i + 1]);
    *reinterpret_cast<uint2*>(dl + 8 * i) = make_uint2(ql[2 * i], ql[2 * i + 1]);
  }
}
#endif  // AWT_EXPERIMENTAL_F6
}  // namespace

int launch_layernorm(awt_ctx* c, const float* x, const float* gamma, const float* beta, int M, int d, float eps,
                     float* out_f32, const Act& out, int prec, hipStream_t s) {
  AWT_REQUIRE(x && gamma && beta && (out_f32 || out.p16 || out.ilv), AWT_ERR_INVALID, "layernorm: null argument");
  AWT_REQUIRE(out_f32 || prec != PREC_F16F8 || out.lo8 || out.ilv, AWT_ERR_INVALID, "layernorm: f16f8 output needs its lo8 plane (hi8 may be null: not consumed)");
  AWT_REQUIRE(!out.ilv || out_f32 || (prec == PREC_F16F8 && d % 64 == 0), AWT_ERR_INVALID, "layernorm: split lines are an f16f8 format of whole 64-element line pairs");
  AWT_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 64 * 4 * kLnMaxChunks, AWT_ERR_INVALID, "layernorm: d must be a multiple of 4 and <= 1280");
  ProfScope prof(c, AWT_PROF_LAYERNORM, s, 0.0);
  const dim3 grid((M + 3) / 4), block(256);
  if (prec == PREC_F16F8) hipLaunchKernelGGL(layernorm_kernel<PREC_F16F8>, grid, block, 0, s, x, gamma, beta, M, d, eps, out_f32, out);
  else if (prec == PREC_F16X3 || prec == PREC_F16) hipLaunchKernelGGL(layernorm_kernel<PREC_F16X3>, grid, block, 0, s, x, gamma, beta, M, d, eps, out_f32, out);   // PREC_F16: lo16 is null
  else hipLaunchKernelGGL(layernorm_kernel<PREC_BF16X3>, grid, block, 0, s, x, gamma, beta, M, d, eps, out_f32, out);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_split_f32(awt_ctx* c, const float* x, int64_t n, float scale, bf16_t* hi, bf16_t* lo, hipStream_t s) {
  AWT_REQUIRE(x && hi && n > 0 && n % 4 == 0, AWT_ERR_INVALID, "split: n must be a positive multiple of 4");
  ProfScope prof(c, AWT_PROF_OTHER, s, 0.0);
  const int64_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, s, x, n4, scale, hi, lo);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_split_planes(awt_ctx* c, const float* x, int64_t n, float scale, int prec, int f8_exp, bf16_t* p16, bf16_t* lo16, uint8_t* hi8,
                        uint8_t* lo8, hipStream_t s, char* ilv) {
  AWT_REQUIRE(x && (p16 || ilv) && n > 0 && n % 4 == 0, AWT_ERR_INVALID, "split: n must be a positive multiple of 4");
  AWT_REQUIRE(prec == PREC_BF16 || prec == PREC_F16 || prec == PREC_BF16X3 || prec == PREC_F16X3 || prec == PREC_F16F8, AWT_ERR_INVALID, "split: unknown precision");
  AWT_REQUIRE(prec != PREC_F16F8 || (hi8 && lo8) || ilv, AWT_ERR_INVALID, "split: f16f8 needs both e4m3 planes");
  AWT_REQUIRE(!ilv || (prec == PREC_F16F8 && n % 64 == 0), AWT_ERR_INVALID, "split: split lines are an f16f8 format of whole 64-element line pairs");
  AWT_REQUIRE(f8_exp >= -20 && f8_exp <= 20, AWT_ERR_INVALID, "split: bad e4m3 exponent");
  ProfScope prof(c, AWT_PROF_OTHER, s, 0.0);
  const int64_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256); if (grid > 4096) grid = 4096;
  const float f8s = f8_exp >= 0 ? (float)(1u << f8_exp) : 1.0f / (float)(1u << -f8_exp);
  if (prec == PREC_F16F8) hipLaunchKernelGGL(split_planes_kernel<PREC_F16F8>, dim3(grid), dim3(256), 0, s, x, n4, scale, f8s, p16, lo16, hi8, lo8, ilv);
  else if (prec == PREC_F16X3 || prec == PREC_F16) hipLaunchKernelGGL(split_planes_kernel<PREC_F16X3>, dim3(grid), dim3(256), 0, s, x, n4, scale, f8s, p16, prec == PREC_F16 ? nullptr : lo16, hi8, lo8, nullptr);
  else hipLaunchKernelGGL(split_planes_kernel<PREC_BF16X3>, dim3(grid), dim3(256), 0, s, x, n4, scale, f8s, p16, prec == PREC_BF16 ? nullptr : lo16, hi8, lo8, nullptr);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

size_t gemm_pp_weight_bytes(int N, int K) { return (size_t)((N + 255) / 256 * 256) * K * 4; }
int launch_pack_weight_pp(awt_ctx* c, const float* src, int N, int K, int row_off, char* dst, hipStream_t s) {
  AWT_REQUIRE(src && dst && N > 0 && K > 0 && K % 64 == 0 && row_off >= 0, AWT_ERR_INVALID, "pack_weight_pp: K must be a multiple of 64");
  const int64_t total = (int64_t)N * (K / 4);
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_weight_pp_kernel, dim3(grid), dim3(256), 0, s, src, N, K, row_off, dst);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_pack_weight(awt_ctx* c, const float* src, int N, int C, int taps, int64_t ld, int row_off, int col_off, float scale,
                       bf16_t* hi, bf16_t* lo, uint8_t* lo8, int prec, hipStream_t s, int* inexact, bf16_t* s16, uint8_t* s8) {
  AWT_REQUIRE(src && hi && N > 0 && C > 0 && taps > 0 && ld >= col_off + (int64_t)C * taps && ld % 32 == 0, AWT_ERR_INVALID, "pack_weight: bad shape");
  AWT_REQUIRE(prec != PREC_F16F8 || (lo && lo8 && ld % 64 == 0), AWT_ERR_INVALID, "pack_weight: f16f8 needs both e4m3 planes and K a multiple of 64");
  const int64_t total = (int64_t)N * C * taps;
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  if (prec == PREC_F16F8) hipLaunchKernelGGL(pack_weight_kernel<PREC_F16F8>, dim3(grid), dim3(256), 0, s, src, N, C, taps, ld, row_off, col_off, scale, hi, lo, lo8, inexact, s8 ? s16 : nullptr, s8);
  else if (prec == PREC_F16X3 || prec == PREC_F16) hipLaunchKernelGGL(pack_weight_kernel<PREC_F16X3>, dim3(grid), dim3(256), 0, s, src, N, C, taps, ld, row_off, col_off, scale, hi, prec == PREC_F16 ? nullptr : lo, lo8, inexact, nullptr, nullptr);
  else hipLaunchKernelGGL(pack_weight_kernel<PREC_BF16X3>, dim3(grid), dim3(256), 0, s, src, N, C, taps, ld, row_off, col_off, scale, hi, lo, lo8, inexact, nullptr, nullptr);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_im2col_conv1(awt_ctx* c, const float* mel, int B, int C, int T, int K_dst, const Act& out, int prec, hipStream_t s) {
  AWT_REQUIRE(mel && out.p16 && B > 0 && C > 0 && C % 8 == 0 && T > 0 && K_dst % 8 == 0 && K_dst >= 3 * C, AWT_ERR_INVALID, "im2col: bad shape");
  AWT_REQUIRE(prec != PREC_F16F8 || (out.hi8 && out.lo8), AWT_ERR_INVALID, "im2col: f16f8 output needs both e4m3 planes");
  ProfScope prof(c, AWT_PROF_OTHER, s, 0.0);
  const size_t lds = (size_t)C * (kImTile + 3) * sizeof(float);
  const dim3 grid((T + kImTile - 1) / kImTile, B), block(256);
  if (prec == PREC_F16F8) hipLaunchKernelGGL(im2col_conv1_kernel<PREC_F16F8>, grid, block, lds, s, mel, C, T, K_dst, out);
  else if (prec == PREC_F16X3 || prec == PREC_F16) hipLaunchKernelGGL(im2col_conv1_kernel<PREC_F16X3>, grid, block, lds, s, mel, C, T, K_dst, out);
  else hipLaunchKernelGGL(im2col_conv1_kernel<PREC_BF16X3>, grid, block, lds, s, mel, C, T, K_dst, out);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_layernorm_bwd(awt_ctx* c, const float* dy, const float* x, const float* gamma, const float* dres, int M, int d, float eps,
                         float* dx, bf16_t* dx_hi, bf16_t* dx_lo, hipStream_t s, float gscale, const Act* out8) {
  AWT_REQUIRE(dy && x && gamma && dx, AWT_ERR_INVALID, "layernorm_bwd: null argument");
  AWT_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 64 * 4 * kLnMaxChunks, AWT_ERR_INVALID, "layernorm_bwd: d must be a multiple of 4 and <= 1280");
  AWT_REQUIRE(!out8 || (out8->p16 && out8->hi8 && out8->lo8), AWT_ERR_INVALID, "layernorm_bwd: the f16f8 output needs all three planes");
  ProfScope prof(c, AWT_PROF_LAYERNORM, s, 0.0);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, dy, x, gamma, dres, M, d, eps, dx, dx_hi, dx_lo, gscale, out8 ? *out8 : Act{});
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_transpose_f32(awt_ctx* c, const float* src, int N, int C, float* dst, hipStream_t s) {
  AWT_REQUIRE(src && dst && N > 0 && C > 0, AWT_ERR_INVALID, "transpose_f32: bad argument");
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((N + 31) / 32, (C + 31) / 32), dim3(256), 0, s, src, N, C, dst);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int launch_pack_weight_t(awt_ctx* c, const float* src, int N, int C, int64_t ld, int row_off, int col_off, float scale, bf16_t* hi,
                         bf16_t* lo, hipStream_t s) {
  AWT_REQUIRE(src && hi && N > 0 && C > 0 && ld >= col_off + N && ld % 32 == 0, AWT_ERR_INVALID, "pack_weight_t: bad shape");
  hipLaunchKernelGGL(pack_weight_t_kernel, dim3((N + 31) / 32, (C + 31) / 32), dim3(256), 0, s, src, N, C, ld, row_off, col_off, scale, hi, lo);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

size_t outer_reduce_partial_bytes(int M, int r, int ny) {
  const int R = r <= 8 ? 8 : (r <= 16 ? 16 : 32);
  ny = ny < 1024 ? ny : 1024;   // wider blocks are reduced in 1024-column chunks
  const int nslab = (M + kOrRows - 1) / kOrRows;
  return (size_t)(nslab < kOrMaxBlocks ? nslab : kOrMaxBlocks) * R * ny * sizeof(float);
}

static int launch_outer_reduce_1(awt_ctx* c, const bf16_t* x_hi, const bf16_t* x_lo, int64_t ldx, int xcol, int r, const bf16_t* y_hi,
                        const bf16_t* y_lo, int64_t ldy, int ycol, int ny, int M, float scale, float* out, int64_t sj, int64_t sn,
                        float* partial, size_t partial_bytes, int accumulate, hipStream_t s) {
  AWT_REQUIRE(x_hi && y_hi && out && partial && r > 0 && r <= 32 && ny > 0 && ny % 4 == 0 && ny <= 1024, AWT_ERR_INVALID, "outer_reduce: bad shape");
  AWT_REQUIRE(ldy % 4 == 0 && ycol % 4 == 0, AWT_ERR_INVALID, "outer_reduce: Y columns must be 8-byte aligned");
  AWT_REQUIRE(partial_bytes >= outer_reduce_partial_bytes(M, r, ny), AWT_ERR_WORKSPACE, "outer_reduce: partial buffer too small");
  ProfScope prof(c, AWT_PROF_OTHER, s, 0.0);
  const int nslab_all = (M + kOrRows - 1) / kOrRows;
  const int nslab = nslab_all < kOrMaxBlocks ? nslab_all : kOrMaxBlocks;   // = workgroups = partial sums
  const int R = r <= 8 ? 8 : (r <= 16 ? 16 : 32);
  // X columns beyond r inside the R-wide register block read neighbouring (valid, in-row) columns and are discarded below
  AWT_REQUIRE(xcol + R <= ldx, AWT_ERR_INVALID, "outer_reduce: X block exceeds its row");
  if (R == 8) hipLaunchKernelGGL(outer_reduce_kernel<8>, dim3(nslab), dim3(256), 0, s, x_hi, x_lo, ldx, xcol, y_hi, y_lo, ldy, ycol, ny, M, partial);
  else if (R == 16) hipLaunchKernelGGL(outer_reduce_kernel<16>, dim3(nslab), dim3(256), 0, s, x_hi, x_lo, ldx, xcol, y_hi, y_lo, ldy, ycol, ny, M, partial);
  else hipLaunchKernelGGL(outer_reduce_kernel<32>, dim3(nslab), dim3(256), 0, s, x_hi, x_lo, ldx, xcol, y_hi, y_lo, ldy, ycol, ny, M, partial);
  AWT_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(outer_reduce_final_kernel, dim3((r * ny + 31) / 32), dim3(256), 0, s, partial, nslab, R, r, ny, scale, out, sj, sn, accumulate);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

// Y blocks wider than the kernel's 1024 columns (256 threads x 4) are reduced in column chunks (Whisper large: d = 1280)
int launch_outer_reduce(awt_ctx* c, const bf16_t* x_hi, const bf16_t* x_lo, int64_t ldx, int xcol, int r, const bf16_t* y_hi,
                        const bf16_t* y_lo, int64_t ldy, int ycol, int ny, int M, float scale, float* out, int64_t sj, int64_t sn,
                        float* partial, size_t partial_bytes, int accumulate, hipStream_t s) {
  for (int n0 = 0; n0 < ny; n0 += 1024) {
    const int nc = ny - n0 < 1024 ? ny - n0 : 1024;
    int rc = launch_outer_reduce_1(c, x_hi, x_lo, ldx, xcol, r, y_hi, y_lo, ldy, ycol + n0, nc, M, scale, out + (int64_t)n0 * sn, sj, sn,
                                   partial, partial_bytes, accumulate, s);
    if (rc) return rc;
  }
  return AWT_OK;
}

#ifdef AWT_EXPERIMENTAL_F6
int launch_split_planes_f6(awt_ctx* c, const float* x, int M, int K, bf16_t* p16, uint8_t* hi6, uint8_t* lo6, hipStream_t s) {
  AWT_REQUIRE(x && p16 && hi6 && lo6 && M > 0 && K > 0 && K % 64 == 0, AWT_ERR_INVALID, "split_planes_f6: K must be a multiple of 64");
  const int64_t n = (int64_t)M * (K / 32);
  hipLaunchKernelGGL(planes_f6_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, M, K, p16, hi6, lo6);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
int launch_pack_weight_f6(awt_ctx* c, const float* w, int N, int K, bf16_t* w16, uint8_t* hi6, uint8_t* lo6, hipStream_t s) {
  AWT_REQUIRE(w && w16 && hi6 && lo6 && N > 0 && N % 32 == 0 && K > 0 && K % 64 == 0, AWT_ERR_INVALID, "pack_weight_f6: N % 32 == 0 and K % 64 == 0 required");
  const int64_t n = (int64_t)N * (K / 32);
  hipLaunchKernelGGL(planes_f6_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, N, K, w16, hi6, lo6);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
#endif  // AWT_EXPERIMENTAL_F6
