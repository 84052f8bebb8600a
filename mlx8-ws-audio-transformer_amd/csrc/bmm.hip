// Batched matrix products C_z = A_z B_z^T on the MFMA GEMM (gemm.hip, split-bf16 operands) and the row softmax that sits between
// them -- the operators of the decoder's cross-attention in its ABSORBED form (native_decoder.py, scope row f1):
//
//   HF computes K_i = enc W_k,i^T and V_i = enc W_v,i^T over all B x 1500 encoder rows for each of the decoder's layers
//   (HF:modeling_whisper.py:284-356 with `key_value_states`), 2 x 768 x 768 MACs per encoder row and layer, to attend with
//   B x L label rows (L ~ 12, /root/reference/AB/fineTune.py:88-96).  With so few queries the projections can move to the query
//   side:  q_h K_h^T = (q_h W_k,h) enc^T  and  P_h V_h = (P_h enc) W_v,h^T + b_v  (rows of P sum to one), i.e. per clip ONE
//   [H L, d] x [d, S] product, a row softmax, and ONE [H L, S] x [S, d] product against the encoder states themselves.  For
//   H L = 144 that is 5 x fewer MACs than the key / value projections alone, and the [B S, 2 d layers] key / value tensor
//   (7 GB at B = 64 for Whisper-small) and its gradient are never formed.
//
// The per-clip operand (the encoder states, or their transpose) changes every step but is shared by all decoder layers and by
// forward and backward: `awt_bmm_pack` writes it once into the GEMM's fragment-major weight planes, `awt_bmm` then runs
// `batch` GEMMs in one launch (blockIdx.y = matrix).  The same two entry points serve the per-HEAD products with the frozen
// projection blocks (batch = H, operands strided inside the row-major activations).
#include <string>
#include "common.h"

int launch_gemm_batched(awt_ctx* c, int batch, int M, int N, int K, const bf16_t* a_hi, const bf16_t* a_lo, int64_t lda, int64_t bs_a, const bf16_t* w_hi,
                        const bf16_t* w_lo, int64_t bs_w, float* out, const float* resid, int64_t ldo, int64_t bs_o, int n_valid, hipStream_t s);

namespace {
constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }
int pad_to(int x, int m) { return (x + m - 1) / m * m; }

// fp32 sources -> split-bf16 planes of a zero-padded [rows_p, cols_p] matrix per batch entry.  FRAG: fragment-major (w_frag_index: the GEMM's B
// operand); else row-major with pitch cols_p (its A operand).  One thread per 8 consecutive columns = one 16-byte piece of each plane.
//   TRANS = false: element (n, k) = src[z * stride + n * ld + k]                                     (the matrix as it lies in memory)
//   TRANS = true : element (n, k) = (k < k1 ? src : src2)[z * stride + (k < k1 ? k : k - k1) * ld + n]  (given K-MAJOR, optionally as two
//                  matrices stacked along K: P^T | dS^T, or the transposes of per-clip [R, d] blocks) -- consecutive threads take
//                  consecutive n, so the eight reads of a thread are each coalesced across the wave
template <bool FRAG, bool TRANS>
__global__ __launch_bounds__(256) void bmm_planes_kernel(const float* __restrict__ src, const float* __restrict__ src2, int k1, int rows, int cols, int64_t ld,
                                                         int64_t stride, int rows_p, int cols_p, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
  const int groups = cols_p >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows_p * groups) return;
  int n, k0;
  if (TRANS) { k0 = (int)(idx / rows_p) * 8; n = (int)(idx - (int64_t)(k0 >> 3) * rows_p); }
  else { n = (int)(idx / groups); k0 = (int)(idx - (int64_t)n * groups) * 8; }
  float v[8];
  if constexpr (TRANS) {
    const int64_t zo = (int64_t)blockIdx.y * stride + n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      v[j] = (n < rows && k < cols) ? (k < k1 ? src[zo + (int64_t)k * ld] : src2[zo + (int64_t)(k - k1) * ld]) : 0.f;
    }
  } else {
    const float* row = src + (int64_t)blockIdx.y * stride + (int64_t)n * ld;
    if (n < rows && k0 + 8 <= cols && (((uintptr_t)(row + k0)) & 15) == 0) {
      const float4 a = *reinterpret_cast<const float4*>(row + k0), b = *reinterpret_cast<const float4*>(row + k0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (n < rows && k0 + j < cols) ? row[k0 + j] : 0.f;
    }
  }
  bf16_t h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split_bf16(v[j], h[j], l[j]);
  const int64_t base = (int64_t)blockIdx.y * rows_p * cols_p;
  const int64_t o = base + (FRAG ? w_frag_index(n, k0, cols_p >> 5) : (int64_t)n * cols_p + k0);
  *reinterpret_cast<uint4*>(hi + o) = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
  *reinterpret_cast<uint4*>(lo + o) = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
}

template <bool FRAG>
int launch_planes(const float* src, const float* src2, int k1, bool trans, int batch, int rows, int cols, int64_t ld, int64_t stride, int rows_p, int cols_p,
                  bf16_t* hi, bf16_t* lo, hipStream_t s) {
  const int64_t threads = (int64_t)rows_p * (cols_p >> 3);
  const dim3 grid((unsigned)((threads + 255) / 256), batch);
  if (trans) hipLaunchKernelGGL((bmm_planes_kernel<FRAG, true>), grid, dim3(256), 0, s, src, src2 ? src2 : src, k1, rows, cols, ld, stride, rows_p, cols_p, hi, lo);
  else hipLaunchKernelGGL((bmm_planes_kernel<FRAG, false>), grid, dim3(256), 0, s, src, src, cols, rows, cols, ld, stride, rows_p, cols_p, hi, lo);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
// all-threads reduction of a 256-thread workgroup through 4 wave partials (fixed order: deterministic)
template <bool MAX>
__device__ __forceinline__ float block_reduce(float x, float* sh) {
  x = MAX ? wave_max(x) : wave_sum(x);
  __syncthreads();                                   // sh may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  return MAX ? fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// p[r, :] = softmax(scale * s[r, :]) over `cols` columns (pitch ld), in place allowed; one workgroup per row, the row kept in registers
constexpr int kSoftmaxMaxCols = 256 * 16;
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, float* __restrict__ p, int cols, int64_t ld, float scale) {
  __shared__ float sh[4];
  const float* src = s + (int64_t)blockIdx.x * ld;
  float* dst = p + (int64_t)blockIdx.x * ld;
  float v[16];
  float m = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = threadIdx.x + 256 * j;
    v[j] = c < cols ? src[c] * scale : -3.0e38f;
    m = fmaxf(m, v[j]);
  }
  m = block_reduce<true>(m, sh);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int c = threadIdx.x + 256 * j; v[j] = c < cols ? expf(v[j] - m) : 0.f; sum += v[j]; }
  sum = block_reduce<false>(sum, sh);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int c = threadIdx.x + 256 * j; if (c < cols) dst[c] = v[j] * inv; }
}

// ds[r, :] = scale * p[r, :] * (dp[r, :] - sum_c p[r, c] dp[r, c])  (in place on dp allowed)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ ds, int cols,
                                                               int64_t ld, float scale) {
  __shared__ float sh[4];
  const int64_t off = (int64_t)blockIdx.x * ld;
  float pv[16], gv[16];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = threadIdx.x + 256 * j;
    pv[j] = c < cols ? p[off + c] : 0.f;
    gv[j] = c < cols ? dp[off + c] : 0.f;
    dot += pv[j] * gv[j];
  }
  dot = block_reduce<false>(dot, sh);
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int c = threadIdx.x + 256 * j; if (c < cols) ds[off + c] = scale * pv[j] * (gv[j] - dot); }
}
}  // namespace

// ------------------------------------------------------------------------------------------------ C-ABI
extern "C" size_t awt_bmm_packed_bytes(int batch, int N, int K) {
  if (batch <= 0 || N <= 0 || K <= 0) return 0;
  return 2 * align_up((size_t)batch * pad_to(N, 128) * pad_to(K, 64) * 2);
}

namespace {
int bmm_pack_impl(awt_ctx* c, const float* b, const float* b2, int k1, bool trans, int64_t ldb, int64_t stride_b, int batch, int N, int K, void* packed,
                  size_t packed_bytes, void* stream, const char* who) {
  AWT_REQUIRE(c && b && packed && batch > 0 && batch <= 65535 && N > 0 && K > 0 && stride_b >= 0, AWT_ERR_INVALID, std::string(who) + ": bad argument");
  AWT_REQUIRE(trans ? (ldb >= N && k1 > 0 && k1 <= K && (k1 == K || b2)) : ldb >= K, AWT_ERR_INVALID, std::string(who) + ": pitch smaller than a row, or a missing second matrix");
  AWT_REQUIRE(packed_bytes >= awt_bmm_packed_bytes(batch, N, K), AWT_ERR_WORKSPACE, std::string(who) + ": packed buffer too small (awt_bmm_packed_bytes)");
  AWT_REQUIRE(((uintptr_t)packed & 255) == 0, AWT_ERR_INVALID, std::string(who) + ": packed buffer must be 256-byte aligned");
  const int Np = pad_to(N, 128), Kp = pad_to(K, 64);
  bf16_t* hi = (bf16_t*)packed;
  bf16_t* lo = (bf16_t*)((char*)packed + align_up((size_t)batch * Np * Kp * 2));
  ProfScope prof(c, AWT_PROF_OTHER, (hipStream_t)stream, 0.0);
  return launch_planes<true>(b, b2, k1, trans, batch, N, K, ldb, stride_b, Np, Kp, hi, lo, (hipStream_t)stream);
}

int bmm_impl(awt_ctx* c, const float* a, const float* a2, int k1, bool trans, int64_t lda, int64_t stride_a, const void* packed_b, const float* resid, float* out,
             int64_t ldo, int64_t stride_o, int batch, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream, const char* who) {
  AWT_REQUIRE(c && a && packed_b && out && workspace && batch > 0 && batch <= 65535 && M > 0 && N > 0 && K > 0, AWT_ERR_INVALID, std::string(who) + ": bad argument");
  AWT_REQUIRE(trans ? (lda >= M && k1 > 0 && k1 <= K && (k1 == K || a2)) : lda >= K, AWT_ERR_INVALID, std::string(who) + ": pitch smaller than a row, or a missing second matrix");
  AWT_REQUIRE(stride_a >= 0 && N % 4 == 0 && ldo % 4 == 0 && ldo >= N && stride_o % 4 == 0 && ((uintptr_t)out & 15) == 0 && (!resid || ((uintptr_t)resid & 15) == 0),
              AWT_ERR_INVALID, std::string(who) + ": N, the output pitch and the output stride must be multiples of 4 (16-byte rows)");
  AWT_REQUIRE(ws_bytes >= awt_bmm_workspace_bytes(batch, M, K), AWT_ERR_WORKSPACE, std::string(who) + ": workspace too small (awt_bmm_workspace_bytes)");
  AWT_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)packed_b & 255) == 0, AWT_ERR_INVALID, std::string(who) + ": workspace and packed operand must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int Np = pad_to(N, 128), Kp = pad_to(K, 64);
  bf16_t* ah = (bf16_t*)workspace;
  bf16_t* al = (bf16_t*)((char*)workspace + align_up((size_t)batch * M * Kp * 2));
  {
    ProfScope prof(c, AWT_PROF_OTHER, s, 0.0);
    int rc = launch_planes<false>(a, a2, k1, trans, batch, M, K, lda, stride_a, M, Kp, ah, al, s); if (rc) return rc;
  }
  const bf16_t* wh = (const bf16_t*)packed_b;
  const bf16_t* wl = (const bf16_t*)((const char*)packed_b + align_up((size_t)batch * Np * Kp * 2));
  return launch_gemm_batched(c, batch, M, Np, Kp, ah, al, Kp, (int64_t)M * Kp, wh, wl, (int64_t)Np * Kp, out, resid, ldo, stride_o, N, s);
}
}  // namespace

extern "C" int awt_bmm_pack(awt_ctx* c, const float* b, int64_t ldb, int64_t stride_b, int batch, int N, int K, void* packed, size_t packed_bytes, void* stream) {
  return bmm_pack_impl(c, b, nullptr, K, false, ldb, stride_b, batch, N, K, packed, packed_bytes, stream, "bmm_pack");
}
extern "C" int awt_bmm_pack_kmajor(awt_ctx* c, const float* b1, const float* b2, int K1, int64_t ldb, int64_t stride_b, int batch, int N, int K, void* packed,
                                   size_t packed_bytes, void* stream) {
  return bmm_pack_impl(c, b1, b2, K1, true, ldb, stride_b, batch, N, K, packed, packed_bytes, stream, "bmm_pack_kmajor");
}

extern "C" size_t awt_bmm_workspace_bytes(int batch, int M, int K) {
  if (batch <= 0 || M <= 0 || K <= 0) return 0;
  return 2 * align_up((size_t)batch * M * pad_to(K, 64) * 2);
}

extern "C" int awt_bmm(awt_ctx* c, const float* a, int64_t lda, int64_t stride_a, const void* packed_b, const float* resid, float* out, int64_t ldo,
                       int64_t stride_o, int batch, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream) {
  return bmm_impl(c, a, nullptr, K, false, lda, stride_a, packed_b, resid, out, ldo, stride_o, batch, M, N, K, workspace, ws_bytes, stream, "bmm");
}
extern "C" int awt_bmm_kmajor(awt_ctx* c, const float* a1, const float* a2, int K1, int64_t lda, int64_t stride_a, const void* packed_b, const float* resid,
                              float* out, int64_t ldo, int64_t stride_o, int batch, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream) {
  return bmm_impl(c, a1, a2, K1, true, lda, stride_a, packed_b, resid, out, ldo, stride_o, batch, M, N, K, workspace, ws_bytes, stream, "bmm_kmajor");
}

extern "C" int awt_op_softmax_rows(awt_ctx* c, const float* s_in, float* p, int rows, int cols, int64_t ld, float scale, void* stream) {
  AWT_REQUIRE(c && s_in && p && rows > 0 && cols > 0 && cols <= kSoftmaxMaxCols && ld >= cols, AWT_ERR_INVALID, "op_softmax_rows: 1..4096 columns, ld >= cols");
  ProfScope prof(c, AWT_PROF_OTHER, (hipStream_t)stream, 0.0);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s_in, p, cols, ld, scale);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

extern "C" int awt_op_softmax_rows_backward(awt_ctx* c, const float* p, const float* dp, float* ds, int rows, int cols, int64_t ld, float scale, void* stream) {
  AWT_REQUIRE(c && p && dp && ds && rows > 0 && cols > 0 && cols <= kSoftmaxMaxCols && ld >= cols, AWT_ERR_INVALID, "op_softmax_rows_backward: 1..4096 columns, ld >= cols");
  ProfScope prof(c, AWT_PROF_OTHER, (hipStream_t)stream, 0.0);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, p, dp, ds, cols, ld, scale);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
