// Operators of the Whisper DECODER half of the fine-tune step (scope row f1: `WhisperForConditionalGeneration.forward`,
// /root/reference/AB/fineTune.py:131,186-199 -> HF:modeling_whisper.py:416-507,649-797,994-1100), gfx950.
//
// The decoder works on B x L label tokens (L ~ 12 in the reference's data, <= 448), three orders of magnitude fewer rows than the
// encoder's B x 1500 frames, so its shape is different from the encoder's:
//  * linears ([B L, d] x frozen weights, incl. the tied 51 865-row vocabulary projection) are WEIGHT-bound: they run on the
//    encoder's MFMA GEMM (gemm.hip) against weights packed ONCE into fragment-major planes (`awt_weight`), forward and -- with the
//    transposed copy -- backward-to-input; frozen weights get no gradient;
//  * attention (causal self-attention over <= 448 positions, cross-attention of L queries over the 1500 encoder positions) has too
//    few query rows for 32 x 32 MFMA tiles to pay; it is an fp32 kernel with one wave per query row (forward, dq) and one lane
//    per key (dk, dv), reading q / k / v straight out of the row-major outputs of the linears (no head-major copies);
//  * token + position embedding, exact-erf GELU and its derivative, LayerNorm backward, and the cross-entropy over the padded
//    vocabulary (loss, log-sum-exp and d(logits) in one pass over the logits) are row kernels.
// Everything is deterministic (no atomics).
#include <vector>
#include "common.h"

// ------------------------------------------------------------------------------------------------ packed frozen weights
struct awt_weight {
  awt_ctx* ctx = nullptr;
  int N = 0, K = 0, Np = 0, prec = PREC_BF16X3;
  bf16_t *hi = nullptr, *lo = nullptr;        // [Np, K] fragment-major (w_frag_index)
  bf16_t *t_hi = nullptr, *t_lo = nullptr;    // [K, Np] fragment-major: the weight of dx = dy W
  float* bias = nullptr;                      // [Np] (zero-padded) or null
};

namespace {
constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }
int dmalloc0(void** p, size_t bytes) {
  AWT_HIP_CHECK(hipMalloc(p, bytes));
  AWT_HIP_CHECK(hipMemset(*p, 0, bytes));
  return AWT_OK;
}
}  // namespace

extern "C" void awt_weight_destroy(awt_weight* w) {
  if (!w) return;
  for (void* p : {(void*)w->hi, (void*)w->lo, (void*)w->t_hi, (void*)w->t_lo, (void*)w->bias}) if (p) (void)hipFree(p);
  delete w;
}

extern "C" int awt_weight_create(awt_ctx* c, const float* w, const float* bias, int N, int K, int precision, int with_transpose,
                                 void* stream, awt_weight** out) {
  AWT_REQUIRE(c && w && out && N > 0 && K > 0, AWT_ERR_INVALID, "weight_create: bad argument");
  AWT_REQUIRE(K % 64 == 0, AWT_ERR_INVALID, "weight_create: K must be a multiple of 64");
  AWT_REQUIRE(precision == PREC_BF16 || precision == PREC_BF16X3, AWT_ERR_INVALID, "weight_create: precision must be 1 (bf16) or 3 (bf16x3)");
  hipStream_t s = (hipStream_t)stream;
  awt_weight* h = new awt_weight();
  h->ctx = c; h->N = N; h->K = K; h->prec = precision;
  h->Np = (N + 127) / 128 * 128;          // GEMM column tiles are 128 wide; as the K of dx = dy W a multiple of 128 is a multiple of 64
  const size_t plane = (size_t)h->Np * K * 2;
  int rc = dmalloc0((void**)&h->hi, plane);
  if (!rc && precision == PREC_BF16X3) rc = dmalloc0((void**)&h->lo, plane);
  if (!rc) rc = launch_pack_weight(c, w, N, K, 1, K, 0, 0, 1.0f, h->hi, h->lo, nullptr, precision, s);
  if (!rc && with_transpose) {
    rc = dmalloc0((void**)&h->t_hi, plane);
    if (!rc && precision == PREC_BF16X3) rc = dmalloc0((void**)&h->t_lo, plane);
    if (!rc) rc = launch_pack_weight_t(c, w, N, K, h->Np, 0, 0, 1.0f, h->t_hi, h->t_lo, s);
  }
  if (!rc && bias) {
    rc = dmalloc0((void**)&h->bias, (size_t)h->Np * 4);
    if (!rc && hipMemcpyAsync(h->bias, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = awt_fail(AWT_ERR_HIP, "weight_create: bias copy failed");
  }
  if (rc) { awt_weight_destroy(h); return rc; }
  *out = h;
  return AWT_OK;
}
extern "C" int awt_weight_padded_rows(const awt_weight* w) { return w ? w->Np : 0; }

extern "C" size_t awt_linear_workspace_bytes(const awt_weight* w, int M, int backward) {
  if (!w || M <= 0) return 0;
  return 2 * align_up((size_t)M * (backward ? w->Np : w->K) * 2);     // the hi / lo planes of x [M, K] or of dy [M, Np]
}

// y [M, Np] = x [M, K] W^T + bias (+ resid [M, Np]); columns N .. Np - 1 of y are written as 0 + resid
extern "C" int awt_linear_forward(awt_ctx* c, const awt_weight* w, const float* x, const float* resid, float* y, int M, void* workspace,
                                  size_t ws_bytes, void* stream) {
  AWT_REQUIRE(c && w && x && y && workspace && M > 0, AWT_ERR_INVALID, "linear_forward: bad argument");
  AWT_REQUIRE(ws_bytes >= awt_linear_workspace_bytes(w, M, 0), AWT_ERR_WORKSPACE, "linear_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  bf16_t* xh = (bf16_t*)workspace;
  bf16_t* xl = (bf16_t*)((char*)workspace + align_up((size_t)M * w->K * 2));
  int rc = launch_split_planes(c, x, (int64_t)M * w->K, 1.0f, w->prec, 0, xh, xl, nullptr, nullptr, s); if (rc) return rc;
  GemmSeg sg{};
  sg.a_hi = xh; sg.a_lo = w->prec == PREC_BF16X3 ? xl : nullptr; sg.lda = w->K;
  sg.w_hi = w->hi; sg.w_lo = w->lo; sg.w_ksteps = w->K / 32; sg.w_k0 = 0; sg.K = w->K;
  sg.rows_out = M; sg.rows_in = M; sg.row_mul = 1; sg.row_add = 0;
  GemmOut o{}; o.f32 = y; o.resid = resid; o.ldo = w->Np; o.bias = w->bias; o.n_valid = w->Np;
  return launch_gemm(c, M, w->Np, &sg, 1, w->prec, resid ? EPI_F32_RESID : EPI_F32, o, s);
}

// dx [M, K] = dy [M, Np] W   (the columns of dy beyond N must be zero or meet zero weight rows: the transposed copy is zero-padded)
extern "C" int awt_linear_backward_input(awt_ctx* c, const awt_weight* w, const float* dy, float* dx, int M, void* workspace,
                                         size_t ws_bytes, void* stream) {
  AWT_REQUIRE(c && w && dy && dx && workspace && M > 0, AWT_ERR_INVALID, "linear_backward_input: bad argument");
  AWT_REQUIRE(w->t_hi, AWT_ERR_STATE, "linear_backward_input: the weight was created without its transposed copy");
  AWT_REQUIRE(w->K % 128 == 0, AWT_ERR_INVALID, "linear_backward_input: the weight's K (= N of this product) must be a multiple of 128");
  AWT_REQUIRE(ws_bytes >= awt_linear_workspace_bytes(w, M, 1), AWT_ERR_WORKSPACE, "linear_backward_input: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  bf16_t* gh = (bf16_t*)workspace;
  bf16_t* gl = (bf16_t*)((char*)workspace + align_up((size_t)M * w->Np * 2));
  int rc = launch_split_planes(c, dy, (int64_t)M * w->Np, 1.0f, w->prec, 0, gh, gl, nullptr, nullptr, s); if (rc) return rc;
  GemmSeg sg{};
  sg.a_hi = gh; sg.a_lo = w->prec == PREC_BF16X3 ? gl : nullptr; sg.lda = w->Np;
  sg.w_hi = w->t_hi; sg.w_lo = w->t_lo; sg.w_ksteps = w->Np / 32; sg.w_k0 = 0; sg.K = w->Np;
  sg.rows_out = M; sg.rows_in = M; sg.row_mul = 1; sg.row_add = 0;
  GemmOut o{}; o.f32 = dx; o.ldo = w->K; o.n_valid = w->K;
  return launch_gemm(c, M, w->K, &sg, 1, w->prec, EPI_F32, o, s);
}

// ------------------------------------------------------------------------------------------------ row kernels
namespace {

// x[m, :] = tok[ids[m], :] + pos[pos0 + m % L, :]      (HF:modeling_whisper.py:756-770)
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                                    int M, int L, int d, int pos0, int vocab, float* x) {
  const int m = blockIdx.x;
  if (m >= M) return;
  int64_t id = ids[m]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* t = reinterpret_cast<const float4*>(tok + id * d);
  const float4* p = reinterpret_cast<const float4*>(pos + (int64_t)(pos0 + m % L) * d);
  float4* o = reinterpret_cast<float4*>(x + (int64_t)m * d);
  for (int i = threadIdx.x; i < d / 4; i += 256) {
    const float4 a = t[i], b = p[i];
    o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, int64_t n4, float* y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
  }
}
__device__ __forceinline__ float dgelu(float x) {   // d/dx gelu(x) = Phi(x) + x phi(x)
  return 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n4, float* dx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i], g = reinterpret_cast<const float4*>(dy)[i];
    reinterpret_cast<float4*>(dx)[i] = make_float4(g.x * dgelu(v.x), g.y * dgelu(v.y), g.z * dgelu(v.z), g.w * dgelu(v.w));
  }
}

// Cross-entropy over the first `vocab` columns of logits [M, ld] (CrossEntropyLoss(ignore_index = -100), mean over the rows that
// are not ignored; HF:modeling_whisper.py:1079-1086).  One workgroup per row: max, sum of exponentials, then
// row_loss[m] = lse - logit[label] and dlogits[m, :] = (softmax - onehot) * inv_count (zero for ignored rows and padding columns).
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int M, int vocab, int ld,
                                                 const int* __restrict__ count, float* row_loss, float* dlogits) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (int64_t)m * ld;
  float* g = dlogits + (int64_t)m * ld;
  const int64_t lab = labels[m];
  const bool ignored = lab == -100;                                  // the ONLY ignored value (torch.nn.CrossEntropyLoss(ignore_index=-100))
  const bool invalid = !ignored && (lab < 0 || lab >= vocab);        // torch raises for such a target; here the row's loss becomes NaN (never silent)
  float mx = -3.0e38f;
  for (int i = tid; i < vocab; i += 256) mx = fmaxf(mx, z[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = tid; i < vocab; i += 256) sum += __expf(z[i] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float lse = mx + __logf(sum);
  const int n = *count;
  const float scale = (ignored || n <= 0) ? 0.f : 1.0f / (float)n;
  if (tid == 0) row_loss[m] = ignored ? 0.f : (invalid ? __builtin_nanf("") : lse - z[lab]);
  const float inv = 1.0f / sum;
  for (int i = tid; i < ld; i += 256) {
    float v = 0.f;
    if (i < vocab) v = (__expf(z[i] - mx) * inv - (i == lab ? 1.0f : 0.0f)) * scale;
    g[i] = v;
  }
}
__global__ void count_labels_kernel(const int64_t* labels, int M, int vocab, int* count) {
  int n = 0;
  for (int i = threadIdx.x; i < M; i += 64) n += labels[i] != -100 ? 1 : 0;       // out-of-range targets count: their rows poison the loss with NaN (ce_kernel)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if (threadIdx.x == 0) *count = n;
}
// loss = sum(row_loss) / count, summed in a fixed order
__global__ void ce_reduce_kernel(const float* row_loss, int M, const int* count, float* loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < M; i += 64) s += row_loss[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) *loss = *count > 0 ? s / (float)*count : 0.f;
}

// ------------------------------------------------------------------------------------------------ small attention (fp32)
// q: row (b, i) at q + (b Lq + i) ldq + 64 h;  k, v: row (b, j) at k + (b Sk + j) ldk + 64 h;  o like q with ldo.
// score(i, j) = 0.125 q_i . k_j (the reference scales q by head_dim^-1/2 before the product, HF:modeling_whisper.py:309), keys
// j <= i + causal_off only when causal.
struct SmallAttn {
  const float *q, *k, *v; float* o; float* lse;
  const float *dout; float *dq, *dk, *dv; float* delta;
  int B, H, Lq, Sk, ldq, ldk, ldv, ldo, causal, causal_off;
  // attention-probability dropout (torch.nn.MultiheadAttention(dropout=p) in train(): /root/reference/.charles/spectrogram.py:977-985): the softmax
  // output P is multiplied by a keep mask / (1 - p) before P V.  The mask is a pure function of (seed, batch-head, query, key) -- splitmix64 of the flat
  // index, top 24 bits against p -- so the forward and both backward kernels regenerate it instead of storing B H Lq Sk bytes, and a host-side
  // restatement of the same function gives tests the identical mask.  drop_p = 0: no dropout (inv_keep = 1).
  float drop_p, inv_keep; unsigned long long seed;
};
__device__ __forceinline__ float drop_keep(const SmallAttn& a, int bh, int i, int j) {
  if (a.drop_p <= 0.f) return 1.0f;
  unsigned long long z = ((unsigned long long)((long long)bh * a.Lq + i) * (unsigned long long)a.Sk + (unsigned long long)j) + a.seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((float)(unsigned)(z >> 40) * (1.0f / 16777216.0f) >= a.drop_p) ? a.inv_keep : 0.0f;
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

// ---------------------------------------------------------------------------------------------- parameter-gradient reductions
// (trainable consumers of the operators: the UrbanSound8K Transformer classifier, /root/reference/.charles/spectrogram.py:944-1160)
// Column sums over a slab of rows: sums[c] = sum_m a[m, c] (bias gradients) and, for LayerNorm, dgamma[c] = sum_m dy[m, c] xhat[m, c],
// dbeta[c] = sum_m dy[m, c].  One workgroup per slab of kSlabRows rows; wave w takes rows w, w + 4, ...; a lane owns columns
// lane + 64 j.  Partials land in [slab][2][d] and a second kernel adds the slabs in a fixed order (deterministic, no atomics).
constexpr int kSlabRows = 256;
constexpr int kColMaxChunks = 20;          // d <= 1280
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int d, float eps,
                                                            float* __restrict__ partial) {
  __shared__ float red[2][4][64 * 4];      // reused per chunk group of 4
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nch = (d + 63) / 64;
  const int r0 = blockIdx.x * kSlabRows, r1 = min(M, r0 + kSlabRows);
  float dg[kColMaxChunks], db[kColMaxChunks];
#pragma unroll
  for (int j = 0; j < kColMaxChunks; ++j) dg[j] = db[j] = 0.f;
  for (int m = r0 + wave; m < r1; m += 4) {
    const float* xr = x ? x + (int64_t)m * d : nullptr;
    const float* gr = dy + (int64_t)m * d;
    float mean = 0.f, rstd = 1.f;
    float xv[kColMaxChunks];
    if (xr) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < kColMaxChunks; ++j) { const int c = lane + 64 * j; xv[j] = (j < nch && c < d) ? xr[c] : 0.f; s += xv[j]; }
      mean = wave_sum(s) / (float)d;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < kColMaxChunks; ++j) { const int c = lane + 64 * j; const float t = (j < nch && c < d) ? xv[j] - mean : 0.f; sq += t * t; }
      rstd = rsqrtf(wave_sum(sq) / (float)d + eps);
    }
#pragma unroll
    for (int j = 0; j < kColMaxChunks; ++j) {
      const int c = lane + 64 * j;
      if (j < nch && c < d) { const float g = gr[c]; db[j] += g; if (xr) dg[j] += g * (xv[j] - mean) * rstd; }
    }
  }
  // waves -> one partial row per slab
  for (int j0 = 0; j0 < nch; j0 += 4) {
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int j = 0; j < kColMaxChunks; ++j) if (j == j0 + jj) { a = dg[j]; b = db[j]; }
      red[0][wave][jj * 64 + lane] = a; red[1][wave][jj * 64 + lane] = b;
    }
    __syncthreads();
    const int c = j0 * 64 + threadIdx.x;
    if (c < d) {
      const int t = threadIdx.x;
      partial[((int64_t)blockIdx.x * 2 + 0) * d + c] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
      partial[((int64_t)blockIdx.x * 2 + 1) * d + c] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    }
  }
}
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ partial, int nslab, int d, float* out0, float* out1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  float a = 0.f, b = 0.f;
  for (int s = 0; s < nslab; ++s) { a += partial[((int64_t)s * 2 + 0) * d + c]; b += partial[((int64_t)s * 2 + 1) * d + c]; }
  if (out0) out0[c] = a;
  if (out1) out1[c] = b;
}

// Forward.  A workgroup owns QB = 16 query rows of one (batch, head) so that K and V are read once per 16 queries; its four waves
// take the 64-key chunks round robin and keep their own running (max, sum, output) per query, merged at the end.  Per chunk a
// wave (1) has lane j compute the 16 scores of key j against the query tile in LDS (the key row lives in the lane's registers),
// (2) exponentiates with wave-wide max / sum per query and parks the probabilities in LDS, (3) turns its lanes into output
// dimensions and accumulates p V with the chunk's V rows read coalesced and the probabilities broadcast from LDS.
constexpr int QB = 16;

__global__ __launch_bounds__(256) void small_attn_fwd_kernel(SmallAttn a) {
  __shared__ __attribute__((aligned(16))) float qs[QB][64];
  __shared__ __attribute__((aligned(16))) float ps[4][QB][64];
  __shared__ float pm[4][QB], plsum[4][QB];
  __shared__ float po[4][QB][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x, h = bh % a.H, b = bh / a.H;
  const int i0 = blockIdx.y * QB, nq = min(QB, a.Lq - i0);
  const float* kp = a.k + (int64_t)b * a.Sk * a.ldk + 64 * h;
  const float* vp = a.v + (int64_t)b * a.Sk * a.ldv + 64 * h;
  for (int t = threadIdx.x; t < QB * 64; t += 256) {
    const int i = t >> 6, e = t & 63;
    qs[i][e] = i < nq ? a.q[((int64_t)b * a.Lq + i0 + i) * a.ldq + 64 * h + e] * 0.125f : 0.f;
  }
  __syncthreads();
  float m[QB], l[QB], o[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) { m[i] = -3.0e38f; l[i] = 0.f; o[i] = 0.f; }
  const int kmax = a.causal ? min(a.Sk - 1, i0 + nq - 1 + a.causal_off) : a.Sk - 1;      // last key any row of the tile sees
  for (int c = wave; c * 64 <= kmax; c += 4) {
    const int j = c * 64 + lane;
    const float* kr = kp + (int64_t)min(j, a.Sk - 1) * a.ldk;
    float k[64];
#pragma unroll
    for (int e = 0; e < 64; e += 4) { const float4 t = *reinterpret_cast<const float4*>(kr + e); k[e] = t.x; k[e + 1] = t.y; k[e + 2] = t.z; k[e + 3] = t.w; }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 64; e += 4) { const float4 t = *reinterpret_cast<const float4*>(&qs[i][e]); s += t.x * k[e] + t.y * k[e + 1] + t.z * k[e + 2] + t.w * k[e + 3]; }
      const bool ok = j <= kmax && i < nq && (!a.causal || j <= i0 + i + a.causal_off);
      s = ok ? s : -3.0e38f;
      const float m_new = fmaxf(m[i], wave_max(s));
      const float alpha = __expf(m[i] - m_new);
      const float p = ok ? __expf(s - m_new) : 0.f;
      l[i] = l[i] * alpha + wave_sum(p);
      m[i] = m_new;
      o[i] *= alpha;
      ps[wave][i][lane] = p * drop_keep(a, bh, i0 + i, j);          // the row sum l keeps the undropped p: dropout acts on the normalised probabilities
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int j4 = 0; j4 < 64; j4 += 4) {                      // lane = output dimension from here on
      const int jb = c * 64 + j4;
      const float v0 = vp[(int64_t)min(jb, a.Sk - 1) * a.ldv + lane], v1 = vp[(int64_t)min(jb + 1, a.Sk - 1) * a.ldv + lane];
      const float v2 = vp[(int64_t)min(jb + 2, a.Sk - 1) * a.ldv + lane], v3 = vp[(int64_t)min(jb + 3, a.Sk - 1) * a.ldv + lane];
#pragma unroll
      for (int i = 0; i < QB; ++i) {
        const float4 p4 = *reinterpret_cast<const float4*>(&ps[wave][i][j4]);      // zero for masked keys
        o[i] += p4.x * v0 + p4.y * v1 + p4.z * v2 + p4.w * v3;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int i = 0; i < QB; ++i) { po[wave][i][lane] = o[i]; if (lane == 0) { pm[wave][i] = m[i]; plsum[wave][i] = l[i]; } }
  __syncthreads();
  for (int t = threadIdx.x; t < nq * 64; t += 256) {
    const int i = t >> 6, e = t & 63;
    const float M = fmaxf(fmaxf(pm[0][i], pm[1][i]), fmaxf(pm[2][i], pm[3][i]));
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const float f = __expf(pm[w][i] - M); L += plsum[w][i] * f; O += po[w][i][e] * f; }
    a.o[((int64_t)b * a.Lq + i0 + i) * a.ldo + 64 * h + e] = O / L;
    if (a.lse && e == 0) a.lse[(int64_t)bh * a.Lq + i0 + i] = M + __logf(L);
  }
}

// dq (and delta = rowsum(dO * O)): the same tiling; lane j forms ds_ij = p_ij (dO_i . v_j - delta_i) for the 16 rows, then the lanes
// become dimensions and accumulate ds K with the chunk's K rows read coalesced.
__global__ __launch_bounds__(256) void small_attn_dq_kernel(SmallAttn a) {
  __shared__ __attribute__((aligned(16))) float qs[QB][64];
  __shared__ __attribute__((aligned(16))) float gs[QB][64];
  __shared__ __attribute__((aligned(16))) float dss[4][QB][64];
  __shared__ float lses[QB], deltas[QB];
  __shared__ float pq[4][QB][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x, h = bh % a.H, b = bh / a.H;
  const int i0 = blockIdx.y * QB, nq = min(QB, a.Lq - i0);
  const float* kp = a.k + (int64_t)b * a.Sk * a.ldk + 64 * h;
  const float* vp = a.v + (int64_t)b * a.Sk * a.ldv + 64 * h;
  for (int t = threadIdx.x; t < QB * 64; t += 256) {
    const int i = t >> 6, e = t & 63;
    const int64_t ro = ((int64_t)b * a.Lq + i0 + i) * a.ldq + 64 * h + e, oo = ((int64_t)b * a.Lq + i0 + i) * a.ldo + 64 * h + e;
    qs[i][e] = i < nq ? a.q[ro] * 0.125f : 0.f;
    gs[i][e] = i < nq ? a.dout[oo] : 0.f;
  }
  for (int i = wave; i < QB; i += 4) {        // delta_i = dO_i . O_i, one wave per row
    float dlt = 0.f;
    if (i < nq) { const int64_t oo = ((int64_t)b * a.Lq + i0 + i) * a.ldo + 64 * h + lane; dlt = wave_sum(a.dout[oo] * a.o[oo]); }
    if (lane == 0) {
      deltas[i] = dlt;
      lses[i] = i < nq ? a.lse[(int64_t)bh * a.Lq + i0 + i] : 0.f;
      if (i < nq) a.delta[(int64_t)bh * a.Lq + i0 + i] = dlt;
    }
  }
  __syncthreads();
  float dq[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) dq[i] = 0.f;
  const int kmax = a.causal ? min(a.Sk - 1, i0 + nq - 1 + a.causal_off) : a.Sk - 1;
  for (int c = wave; c * 64 <= kmax; c += 4) {
    const int j = c * 64 + lane;
    const float* kr = kp + (int64_t)min(j, a.Sk - 1) * a.ldk;
    const float* vr = vp + (int64_t)min(j, a.Sk - 1) * a.ldv;
    float k[64], v[64];
#pragma unroll
    for (int e = 0; e < 64; e += 4) {
      const float4 t = *reinterpret_cast<const float4*>(kr + e), u = *reinterpret_cast<const float4*>(vr + e);
      k[e] = t.x; k[e + 1] = t.y; k[e + 2] = t.z; k[e + 3] = t.w; v[e] = u.x; v[e + 1] = u.y; v[e + 2] = u.z; v[e + 3] = u.w;
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < 64; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(&qs[i][e]), u = *reinterpret_cast<const float4*>(&gs[i][e]);
        s += t.x * k[e] + t.y * k[e + 1] + t.z * k[e + 2] + t.w * k[e + 3];
        dp += u.x * v[e] + u.y * v[e + 1] + u.z * v[e + 2] + u.w * v[e + 3];
      }
      const bool ok = j <= kmax && i < nq && (!a.causal || j <= i0 + i + a.causal_off);
      dss[wave][i][lane] = ok ? __expf(s - lses[i]) * (dp * drop_keep(a, bh, i0 + i, j) - deltas[i]) : 0.f;   // dP = keep dP'; delta = dO . O = sum_j P_j dP_j either way
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int j4 = 0; j4 < 64; j4 += 4) {
      const int jb = c * 64 + j4;
      const float k0 = kp[(int64_t)min(jb, a.Sk - 1) * a.ldk + lane], k1 = kp[(int64_t)min(jb + 1, a.Sk - 1) * a.ldk + lane];
      const float k2 = kp[(int64_t)min(jb + 2, a.Sk - 1) * a.ldk + lane], k3 = kp[(int64_t)min(jb + 3, a.Sk - 1) * a.ldk + lane];
#pragma unroll
      for (int i = 0; i < QB; ++i) {
        const float4 d4 = *reinterpret_cast<const float4*>(&dss[wave][i][j4]);
        dq[i] += d4.x * k0 + d4.y * k1 + d4.z * k2 + d4.w * k3;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int i = 0; i < QB; ++i) pq[wave][i][lane] = dq[i];
  __syncthreads();
  for (int t = threadIdx.x; t < nq * 64; t += 256) {
    const int i = t >> 6, e = t & 63;
    a.dq[((int64_t)b * a.Lq + i0 + i) * a.ldq + 64 * h + e] = ((pq[0][i][e] + pq[1][i][e]) + (pq[2][i][e] + pq[3][i][e])) * 0.125f;
  }
}

// dk, dv: one lane per key, all query rows of the (batch, head) in a loop; the 64-dimensional accumulators live in registers
__global__ __launch_bounds__(64) void small_attn_dkv_kernel(SmallAttn a) {
  const int lane = threadIdx.x;
  const int chunks = (a.Sk + 63) / 64;
  const int bh = blockIdx.x / chunks, j = (blockIdx.x % chunks) * 64 + lane;
  const int h = bh % a.H, b = bh / a.H;
  const bool live = j < a.Sk;
  const int jj = live ? j : a.Sk - 1;
  // The lane owns key j, but global memory wants a row's 64 dims on neighbouring lanes: tiles go through a padded LDS patch
  // (row pitch 65 words: conflict-free both ways), 16 lanes x float4 per 256-byte row, four rows per instruction.
  __shared__ float tile[64][65];
  const int j0 = (blockIdx.x % chunks) * 64, tr = lane >> 4, tc = (lane & 15) * 4;
  (void)jj;
  float k[64], v[64], dk[64], dv[64];
  auto load_tile = [&](const float* base, int ld, float (&dst)[64]) {
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < 64; r0 += 4) {
      const int r = min(j0 + r0 + tr, a.Sk - 1);
      const float4 t = *reinterpret_cast<const float4*>(base + ((int64_t)b * a.Sk + r) * ld + 64 * h + tc);
      tile[r0 + tr][tc] = t.x; tile[r0 + tr][tc + 1] = t.y; tile[r0 + tr][tc + 2] = t.z; tile[r0 + tr][tc + 3] = t.w;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 64; ++e) dst[e] = tile[lane][e];
  };
  auto store_tile = [&](float* base, int ld, const float (&src)[64]) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 64; ++e) tile[lane][e] = src[e];
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < 64; r0 += 4) {
      const int r = j0 + r0 + tr;
      if (r < a.Sk)
        *reinterpret_cast<float4*>(base + ((int64_t)b * a.Sk + r) * ld + 64 * h + tc) =
            make_float4(tile[r0 + tr][tc], tile[r0 + tr][tc + 1], tile[r0 + tr][tc + 2], tile[r0 + tr][tc + 3]);
    }
  };
  load_tile(a.k, a.ldk, k);
  load_tile(a.v, a.ldv, v);
#pragma unroll
  for (int e = 0; e < 64; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
  const int i0 = a.causal ? max(0, j - a.causal_off) : 0;         // first query row that sees this key
  const int i0w = a.causal ? max(0, (int)(blockIdx.x % chunks) * 64 - a.causal_off) : 0;   // wave-uniform loop start
  for (int i = i0w; i < a.Lq; ++i) {
    const int row = (bh * a.Lq) + i;
    const float* qr = a.q + ((int64_t)b * a.Lq + i) * a.ldq + 64 * h;
    const float* gr = a.dout + ((int64_t)b * a.Lq + i) * a.ldo + 64 * h;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int e = 0; e < 64; e += 4) {      // wave-uniform addresses: scalar loads
      const float4 t = *reinterpret_cast<const float4*>(qr + e), u = *reinterpret_cast<const float4*>(gr + e);
      s += k[e] * t.x + k[e + 1] * t.y + k[e + 2] * t.z + k[e + 3] * t.w;
      dp += v[e] * u.x + v[e + 1] * u.y + v[e + 2] * u.z + v[e + 3] * u.w;
    }
    const bool on = live && i >= i0;
    const float p = on ? __expf(s * 0.125f - a.lse[row]) : 0.f;
    const float keep = drop_keep(a, bh, i, j);
    const float ds = p * (dp * keep - a.delta[row]) * 0.125f;
#pragma unroll
    for (int e = 0; e < 64; e += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qr + e), u = *reinterpret_cast<const float4*>(gr + e);
      dk[e] += ds * t.x; dk[e + 1] += ds * t.y; dk[e + 2] += ds * t.z; dk[e + 3] += ds * t.w;
      dv[e] += p * keep * u.x; dv[e + 1] += p * keep * u.y; dv[e + 2] += p * keep * u.z; dv[e + 3] += p * keep * u.w;
    }
  }
  store_tile(a.dk, a.ldk, dk);
  store_tile(a.dv, a.ldv, dv);
}

int check_small(const SmallAttn& a, const char* who) {
  AWT_REQUIRE(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Sk > 0, AWT_ERR_INVALID, std::string(who) + ": bad shape");
  AWT_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 && a.ldq >= 64 * a.H && a.ldk >= 64 * a.H && a.ldv >= 64 * a.H && a.ldo >= 64 * a.H,
              AWT_ERR_INVALID, std::string(who) + ": row strides must be multiples of 4 and cover H * 64 columns");
  AWT_REQUIRE((int64_t)a.B * a.H * a.Lq < (1ll << 31), AWT_ERR_INVALID, std::string(who) + ": too many query rows");
  AWT_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, AWT_ERR_INVALID, std::string(who) + ": dropout probability must be in [0, 1)");
  return AWT_OK;
}

}  // namespace

extern "C" int awt_op_embed(awt_ctx* c, const int64_t* ids, const float* tok, const float* pos, float* x, int M, int L, int d, int pos0,
                            int vocab, void* stream) {
  AWT_REQUIRE(c && ids && tok && pos && x && M > 0 && L > 0 && d > 0 && d % 4 == 0 && pos0 >= 0 && vocab > 0, AWT_ERR_INVALID, "op_embed: bad argument");
  hipLaunchKernelGGL(embed_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, ids, tok, pos, M, L, d, pos0, vocab, x);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_gelu(awt_ctx* c, const float* x, float* y, int64_t n, void* stream) {
  AWT_REQUIRE(c && x && y && n > 0 && n % 4 == 0, AWT_ERR_INVALID, "op_gelu: n must be a positive multiple of 4");
  int grid = (int)std::min<int64_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n / 4, y);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_gelu_backward(awt_ctx* c, const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  AWT_REQUIRE(c && x && dy && dx && n > 0 && n % 4 == 0, AWT_ERR_INVALID, "op_gelu_backward: n must be a positive multiple of 4");
  int grid = (int)std::min<int64_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dy, n / 4, dx);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_layernorm_backward(awt_ctx* c, const float* dy, const float* x, const float* gamma, const float* dres, float* dx, int M,
                                         int d, float eps, void* stream) {
  return launch_layernorm_bwd(c, dy, x, gamma, dres, M, d, eps, dx, nullptr, nullptr, (hipStream_t)stream);
}
extern "C" size_t awt_op_param_grad_workspace_bytes(int M, int d) { return (size_t)((M + kSlabRows - 1) / kSlabRows) * 2 * (size_t)d * 4; }
extern "C" int awt_op_layernorm_param_grad(awt_ctx* c, const float* dy, const float* x, float* dgamma, float* dbeta, int M, int d, float eps,
                                           void* workspace, size_t ws_bytes, void* stream) {
  AWT_REQUIRE(c && dy && x && dgamma && dbeta && workspace && M > 0 && d > 0 && d <= 64 * kColMaxChunks, AWT_ERR_INVALID,
              "op_layernorm_param_grad: bad argument (d <= 1280)");
  AWT_REQUIRE(ws_bytes >= awt_op_param_grad_workspace_bytes(M, d), AWT_ERR_WORKSPACE, "op_layernorm_param_grad: workspace too small");
  const int nslab = (M + kSlabRows - 1) / kSlabRows;
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, dy, x, M, d, eps, (float*)workspace);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nslab, d, dgamma, dbeta);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_column_sums(awt_ctx* c, const float* a, float* sums, int M, int d, void* workspace, size_t ws_bytes, void* stream) {
  AWT_REQUIRE(c && a && sums && workspace && M > 0 && d > 0 && d <= 64 * kColMaxChunks, AWT_ERR_INVALID, "op_column_sums: bad argument (d <= 1280)");
  AWT_REQUIRE(ws_bytes >= awt_op_param_grad_workspace_bytes(M, d), AWT_ERR_WORKSPACE, "op_column_sums: workspace too small");
  const int nslab = (M + kSlabRows - 1) / kSlabRows;
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, a, (const float*)nullptr, M, d, 0.f, (float*)workspace);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nslab, d, (float*)nullptr, sums);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_cross_entropy(awt_ctx* c, const float* logits, const int64_t* labels, int M, int vocab, int ld, float* loss, float* dlogits,
                                    void* scratch /* >= (M + 1) * 4 bytes */, void* stream) {
  AWT_REQUIRE(c && logits && labels && loss && dlogits && scratch && M > 0 && vocab > 0 && ld >= vocab, AWT_ERR_INVALID, "op_cross_entropy: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int* count = (int*)scratch; float* row_loss = (float*)scratch + 1;
  hipLaunchKernelGGL(count_labels_kernel, dim3(1), dim3(64), 0, s, labels, M, vocab, count);
  hipLaunchKernelGGL(ce_kernel, dim3(M), dim3(256), 0, s, logits, labels, M, vocab, ld, count, row_loss, dlogits);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, s, row_loss, M, count, loss);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

extern "C" int awt_op_attention_small_dropout(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                              float* lse, int B, int H, int Lq, int Sk, int causal, int causal_off, float drop_p, uint64_t seed, void* stream) {
  AWT_REQUIRE(c && q && k && v && o, AWT_ERR_INVALID, "op_attention_small: null argument");
  SmallAttn a{q, k, v, o, lse, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, Lq, Sk, ldq, ldk, ldv, ldo, causal, causal_off, drop_p, drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, seed};
  int rc = check_small(a, "op_attention_small"); if (rc) return rc;
  hipLaunchKernelGGL(small_attn_fwd_kernel, dim3(B * H, (Lq + QB - 1) / QB), dim3(256), 0, (hipStream_t)stream, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_attention_small(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                      float* lse, int B, int H, int Lq, int Sk, int causal, int causal_off, void* stream) {
  return awt_op_attention_small_dropout(c, q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, Lq, Sk, causal, causal_off, 0.0f, 0, stream);
}
extern "C" int awt_op_attention_small_backward_dropout(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                                       const float* dout, int ldo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                                                       int H, int Lq, int Sk, int causal, int causal_off, float drop_p, uint64_t seed, void* stream) {
  AWT_REQUIRE(c && q && k && v && o && dout && lse && delta && dq && dk && dv, AWT_ERR_INVALID, "op_attention_small_backward: null argument");
  SmallAttn a{q, k, v, const_cast<float*>(o), const_cast<float*>(lse), dout, dq, dk, dv, delta, B, H, Lq, Sk, ldq, ldk, ldv, ldo, causal, causal_off, drop_p,
              drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, seed};
  int rc = check_small(a, "op_attention_small_backward"); if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(small_attn_dq_kernel, dim3(B * H, (Lq + QB - 1) / QB), dim3(256), 0, s, a);
  hipLaunchKernelGGL(small_attn_dkv_kernel, dim3(B * H * ((Sk + 63) / 64)), dim3(64), 0, s, a);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
extern "C" int awt_op_attention_small_backward(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                               const float* dout, int ldo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                                               int H, int Lq, int Sk, int causal, int causal_off, void* stream) {
  return awt_op_attention_small_backward_dropout(c, q, ldq, k, ldk, v, ldv, o, dout, ldo, lse, delta, dq, dk, dv, B, H, Lq, Sk, causal, causal_off, 0.0f, 0, stream);
}
