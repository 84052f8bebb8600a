// libawt C-ABI (include/awt.h): handles, weight upload, encoder forward orchestration, profiling hooks.
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
void awt_set_error(const std::string& msg) { g_err = msg; }
int awt_fail(int code, const std::string& msg) { g_err = msg; return code; }
extern "C" const char* awt_last_error(void) { return g_err.c_str(); }
extern "C" const char* awt_version(void) { return "awt 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------ profiling
struct awt_prof_state {
  struct Span { hipEvent_t a, b; };
  std::vector<Span> spans[AWT_PROF_NCLASSES];
  std::vector<Span> pool;
  double flops[AWT_PROF_NCLASSES] = {0, 0, 0, 0, 0};
  Span open[AWT_PROF_NCLASSES];
};

void awt_prof_begin(awt_ctx* c, int klass, hipStream_t s, double flops) {
  awt_prof_state* p = c->prof;
  awt_prof_state::Span sp;
  if (!p->pool.empty()) { sp = p->pool.back(); p->pool.pop_back(); }
  else { (void)hipEventCreate(&sp.a); (void)hipEventCreate(&sp.b); }
  (void)hipEventRecord(sp.a, s);
  p->open[klass] = sp;
  p->flops[klass] += flops;
}
void awt_prof_end(awt_ctx* c, int klass, hipStream_t s) {
  awt_prof_state* p = c->prof;
  (void)hipEventRecord(p->open[klass].b, s);
  p->spans[klass].push_back(p->open[klass]);
}

extern "C" int awt_prof_enable(awt_ctx* c, int on) {
  AWT_REQUIRE(c, AWT_ERR_INVALID, "prof_enable: null ctx");
  if (!c->prof) c->prof = new awt_prof_state();
  c->prof_on = on ? 1 : 0;
  return AWT_OK;
}
extern "C" int awt_prof_collect(awt_ctx* c, int klass, double* total_ms, int64_t* launches, double* flops) {
  AWT_REQUIRE(c && c->prof && klass >= 0 && klass < AWT_PROF_NCLASSES, AWT_ERR_INVALID, "prof_collect: bad argument");
  awt_prof_state* p = c->prof;
  double ms = 0;
  for (auto& sp : p->spans[klass]) {
    AWT_HIP_CHECK(hipEventSynchronize(sp.b));
    float t = 0;
    AWT_HIP_CHECK(hipEventElapsedTime(&t, sp.a, sp.b));
    ms += t;
    p->pool.push_back(sp);
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int64_t)p->spans[klass].size();
  if (flops) *flops = p->flops[klass];
  p->spans[klass].clear();
  p->flops[klass] = 0;
  return AWT_OK;
}

// ------------------------------------------------------------------------------------------------ ctx
extern "C" int awt_ctx_create(int device, awt_ctx** out) {
  AWT_REQUIRE(out, AWT_ERR_INVALID, "ctx_create: null out");
  int n = 0;
  AWT_HIP_CHECK(hipGetDeviceCount(&n));
  AWT_REQUIRE(device >= 0 && device < n, AWT_ERR_INVALID, "ctx_create: no such device");
  AWT_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  AWT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  AWT_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, AWT_ERR_INVALID,
              std::string("libawt is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
  awt_ctx* c = new awt_ctx();
  c->device = device;
  *out = c;
  return AWT_OK;
}
extern "C" void awt_ctx_destroy(awt_ctx* c) {
  if (!c) return;
  awt_free_tables(c);
  if (c->prof) {
    for (int k = 0; k < AWT_PROF_NCLASSES; ++k)
      for (auto& sp : c->prof->spans[k]) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto& sp : c->prof->pool) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    delete c->prof;
  }
  delete c;
}

extern "C" int awt_logmel_whisper(awt_ctx* c, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                                  int max_valid, int B, int n_frames_out, float* out, void* workspace, size_t ws_bytes,
                                  void* stream) {
  return logmel_whisper_impl(c, pcm, pcm_is_i16, pcm_stride, n_valid, max_valid, B, n_frames_out, out, workspace, ws_bytes,
                             (hipStream_t)stream);
}
extern "C" int awt_logmel_generic(awt_ctx* c, const float* pcm, int64_t pcm_stride, int B, int n_samples, int sample_rate,
                                  int n_fft, int hop, int n_mels, float f_min, float f_max, float log_eps, float* out,
                                  void* stream) {
  return logmel_generic_impl(c, pcm, pcm_stride, B, n_samples, sample_rate, n_fft, hop, n_mels, f_min, f_max, log_eps, out,
                             (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ encoder
namespace {

struct Planes {  // a bf16 matrix as hi (+ lo) planes owned by the library
  bf16_t* hi = nullptr; bf16_t* lo = nullptr; int64_t rows = 0, ld = 0;
};
struct Linear {
  Planes w; float* bias = nullptr; int N = 0, K = 0;
};
struct LoraGroup {   // adapters that share one input (q/k/v share LN1's output)
  bool active = false;
  Planes a;          // [128, K_in]: rows slot * r .. slot * r + r - 1 hold adapter `slot`'s A
  Planes b;          // [N_out, kp]: columns slot * r .. hold B (pre-scaling is applied to u, not to B)
  int kp = 0;
};
struct Layer {
  Linear qkv, out, fc1, fc2;
  float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  LoraGroup lq, lo_, l1, l2;
};

constexpr int kConv1K = 256;  // 3 taps x 80 mel bins = 240, zero-padded to a multiple of 64
constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }

}  // namespace

struct awt_encoder {
  awt_ctx* ctx = nullptr;
  awt_encoder_cfg cfg{};
  int planes = 1;          // 1 (bf16) or 2 (hi + lo)
  Linear conv1, conv2;
  float* pos = nullptr;    // [S, d]
  float *lnf_g = nullptr, *lnf_b = nullptr;
  std::vector<Layer> layers;
  std::vector<void*> allocs;
  std::vector<std::string> seen;   // names uploaded so far
  int chunk = 16;
};

namespace {

int dev_alloc(awt_encoder* e, void** p, size_t bytes) {
  AWT_HIP_CHECK(hipMalloc(p, bytes));
  AWT_HIP_CHECK(hipMemset(*p, 0, bytes));
  e->allocs.push_back(*p);
  return AWT_OK;
}
int alloc_planes(awt_encoder* e, Planes* pl, int64_t rows, int64_t ld) {
  pl->rows = rows; pl->ld = ld;
  int rc = dev_alloc(e, (void**)&pl->hi, (size_t)rows * ld * 2); if (rc) return rc;
  if (e->planes == 2) { rc = dev_alloc(e, (void**)&pl->lo, (size_t)rows * ld * 2); if (rc) return rc; }
  return AWT_OK;
}
int alloc_linear(awt_encoder* e, Linear* l, int N, int K) {
  l->N = N; l->K = K;
  int rc = alloc_planes(e, &l->w, N, K); if (rc) return rc;
  return dev_alloc(e, (void**)&l->bias, (size_t)N * 4);
}
int alloc_lora(awt_encoder* e, LoraGroup* g, int slots, int K_in, int N_out) {
  g->active = true;
  g->kp = (int)(((int64_t)slots * e->cfg.lora_rank + 63) / 64 * 64);
  int rc = alloc_planes(e, &g->a, 128, K_in); if (rc) return rc;
  return alloc_planes(e, &g->b, N_out, g->kp);
}

struct Workspace {  // per-chunk buffers carved from the caller's workspace
  float* x; bf16_t *ln[2], *qkv[2], *att[2], *ff[2], *u[2], *a1[2], *h1[2];
  size_t bytes;
};

Workspace carve(const awt_encoder* e, char* base, int Bc) {
  const awt_encoder_cfg& c = e->cfg;
  const size_t P = e->planes, S = c.n_ctx, T = 2 * S, d = c.d_model, f = c.ffn_dim;
  const size_t M = (size_t)Bc * S, Mt = (size_t)Bc * T;
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
  w.x = (float*)take(M * d * 4);
  for (size_t p = 0; p < P; ++p) w.ln[p] = (bf16_t*)take(M * d * 2);
  for (size_t p = 0; p < P; ++p) w.u[p] = (bf16_t*)take(M * 128 * 2);
  // layer-phase buffers; the conv-phase buffers (im2col rows and conv1 output) alias the same region
  const size_t layer_start = off;
  for (size_t p = 0; p < P; ++p) w.qkv[p] = (bf16_t*)take(3 * M * d * 2);
  for (size_t p = 0; p < P; ++p) w.att[p] = (bf16_t*)take(M * d * 2);
  for (size_t p = 0; p < P; ++p) w.ff[p] = (bf16_t*)take(M * f * 2);
  const size_t layer_end = off;
  off = layer_start;
  for (size_t p = 0; p < P; ++p) w.a1[p] = (bf16_t*)take(Mt * kConv1K * 2);
  for (size_t p = 0; p < P; ++p) w.h1[p] = (bf16_t*)take(Mt * d * 2);
  w.bytes = std::max(off, layer_end);
  return w;
}

GemmSeg seg_plain(const bf16_t* a_hi, const bf16_t* a_lo, int64_t lda, const Planes& w, int64_t wcol, int K, int M) {
  GemmSeg s{};
  s.a_hi = a_hi; s.a_lo = a_lo; s.lda = lda;
  s.w_hi = w.hi + wcol; s.w_lo = w.lo ? w.lo + wcol : nullptr; s.ldw = w.ld; s.K = K;
  s.rows_out = M; s.rows_in = M; s.row_mul = 1; s.row_add = 0;
  return s;
}

// y = x W^T (+ LoRA: [x | u] [W | B]^T with u = (alpha / r) x A^T) with the given epilogue
int linear_with_lora(awt_encoder* e, const Workspace& w, bf16_t* const in[2], int64_t ld_in, const Linear& lin, const LoraGroup& lg,
                     int M, GemmEpilogue epi, GemmOut out, hipStream_t s) {
  const int terms = e->cfg.mfma_terms;
  GemmSeg segs[2];
  int nseg = 1;
  segs[0] = seg_plain(in[0], in[1], ld_in, lin.w, 0, lin.K, M);
  if (lg.active) {
    GemmSeg us = seg_plain(in[0], in[1], ld_in, lg.a, 0, lin.K, M);
    GemmOut uo{};
    uo.hi = w.u[0]; uo.lo = e->planes == 2 ? w.u[1] : nullptr; uo.ldo = lg.kp; uo.n_valid = lg.kp;
    uo.scale = e->cfg.lora_alpha / (float)e->cfg.lora_rank;
    int rc = launch_gemm(e->ctx, M, 128, &us, 1, terms, EPI_BF16, uo, s); if (rc) return rc;
    segs[1] = seg_plain(w.u[0], e->planes == 2 ? w.u[1] : nullptr, lg.kp, lg.b, 0, lg.kp, M);
    nseg = 2;
  }
  out.bias = lin.bias;
  out.n_valid = lin.N;
  return launch_gemm(e->ctx, M, lin.N, segs, nseg, terms, epi, out, s);
}

int forward_chunk(awt_encoder* e, const float* mel, int Bc, float* hidden, char* ws_base, hipStream_t s) {
  const awt_encoder_cfg& c = e->cfg;
  const int S = c.n_ctx, T = 2 * S, d = c.d_model, f = c.ffn_dim, H = c.n_heads, terms = c.mfma_terms;
  const int M = Bc * S, Mt = Bc * T;
  const bool two = e->planes == 2;
  Workspace w = carve(e, ws_base, Bc);
  int rc;
  // ---- conv stem (K5-K7)
  rc = launch_im2col_conv1(e->ctx, mel, Bc, c.n_mels, T, kConv1K, w.a1[0], two ? w.a1[1] : nullptr, s); if (rc) return rc;
  {
    GemmSeg sg = seg_plain(w.a1[0], two ? w.a1[1] : nullptr, kConv1K, e->conv1.w, 0, kConv1K, Mt);
    GemmOut o{}; o.hi = w.h1[0]; o.lo = two ? w.h1[1] : nullptr; o.ldo = d; o.bias = e->conv1.bias; o.n_valid = d;
    rc = launch_gemm(e->ctx, Mt, d, &sg, 1, terms, EPI_BF16_GELU, o, s); if (rc) return rc;
  }
  {
    GemmSeg sg[3];
    for (int dt = 0; dt < 3; ++dt) {
      sg[dt] = seg_plain(w.h1[0], two ? w.h1[1] : nullptr, d, e->conv2.w, (int64_t)dt * d, d, M);
      sg[dt].rows_out = S; sg[dt].rows_in = T; sg[dt].row_mul = 2; sg[dt].row_add = dt - 1;
    }
    GemmOut o{}; o.f32 = w.x; o.ldo = d; o.bias = e->conv2.bias; o.n_valid = d; o.pos = e->pos; o.rows_pos = S;
    rc = launch_gemm(e->ctx, M, d, sg, 3, terms, EPI_F32_GELU_POS, o, s); if (rc) return rc;
  }
  // ---- transformer layers (K8-K13)
  bf16_t* lnp[2] = {w.ln[0], two ? w.ln[1] : nullptr};
  bf16_t* attp[2] = {w.att[0], two ? w.att[1] : nullptr};
  bf16_t* ffp[2] = {w.ff[0], two ? w.ff[1] : nullptr};
  const int64_t plane = (int64_t)M * d;
  for (int li = 0; li < c.n_layers; ++li) {
    Layer& L = e->layers[li];
    rc = launch_layernorm(e->ctx, w.x, L.ln1_g, L.ln1_b, M, d, 1e-5f, nullptr, lnp[0], lnp[1], s); if (rc) return rc;
    {
      GemmOut o{}; o.hi = w.qkv[0]; o.lo = two ? w.qkv[1] : nullptr; o.scale = 0.125f; o.S = S; o.H = H; o.plane_stride = plane;
      rc = linear_with_lora(e, w, lnp, d, L.qkv, L.lq, M, EPI_QKV, o, s); if (rc) return rc;
    }
    rc = launch_attention(e->ctx, w.qkv[0], two ? w.qkv[1] : nullptr, w.qkv[0] + plane, two ? w.qkv[1] + plane : nullptr,
                          w.qkv[0] + 2 * plane, two ? w.qkv[1] + 2 * plane : nullptr, attp[0], attp[1], nullptr, Bc, H, S, terms, s);
    if (rc) return rc;
    {
      GemmOut o{}; o.f32 = w.x; o.resid = w.x; o.ldo = d;
      rc = linear_with_lora(e, w, attp, d, L.out, L.lo_, M, EPI_F32_RESID, o, s); if (rc) return rc;
    }
    rc = launch_layernorm(e->ctx, w.x, L.ln2_g, L.ln2_b, M, d, 1e-5f, nullptr, lnp[0], lnp[1], s); if (rc) return rc;
    {
      GemmOut o{}; o.hi = ffp[0]; o.lo = ffp[1]; o.ldo = f;
      rc = linear_with_lora(e, w, lnp, d, L.fc1, L.l1, M, EPI_BF16_GELU, o, s); if (rc) return rc;
    }
    {
      GemmOut o{}; o.f32 = w.x; o.resid = w.x; o.ldo = d;
      rc = linear_with_lora(e, w, ffp, f, L.fc2, L.l2, M, EPI_F32_RESID, o, s); if (rc) return rc;
    }
  }
  return launch_layernorm(e->ctx, w.x, e->lnf_g, e->lnf_b, M, d, 1e-5f, hidden, nullptr, nullptr, s);
}

bool parse_layer(const char* name, int* idx, const char** rest) {
  if (strncmp(name, "layers.", 7) != 0) return false;
  char* end = nullptr;
  long v = strtol(name + 7, &end, 10);
  if (end == name + 7 || *end != '.') return false;
  *idx = (int)v; *rest = end + 1;
  return true;
}

int check_shape(const char* name, const int64_t* shape, int rank, std::initializer_list<int64_t> want) {
  bool ok = rank == (int)want.size();
  int i = 0;
  for (int64_t w : want) { if (ok && shape[i] != w) ok = false; ++i; }
  if (!ok) {
    std::string m = std::string("set_weight: ") + name + " has shape [";
    for (int j = 0; j < rank; ++j) m += (j ? "," : "") + std::to_string(shape[j]);
    m += "], expected [";
    i = 0;
    for (int64_t w : want) m += (i++ ? "," : "") + std::to_string(w);
    return awt_fail(AWT_ERR_INVALID, m + "]");
  }
  return AWT_OK;
}

int copy_f32(float* dst, const float* src, size_t n, hipStream_t s) {
  AWT_HIP_CHECK(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, s));
  return AWT_OK;
}

}  // namespace

extern "C" int awt_encoder_create(awt_ctx* c, const awt_encoder_cfg* cfg, awt_encoder** out) {
  AWT_REQUIRE(c && cfg && out, AWT_ERR_INVALID, "encoder_create: null argument");
  AWT_REQUIRE(cfg->d_model > 0 && cfg->d_model % 128 == 0 && cfg->d_model <= 1024, AWT_ERR_INVALID, "encoder_create: d_model must be a multiple of 128, <= 1024");
  AWT_REQUIRE(cfg->n_heads > 0 && cfg->d_model == cfg->n_heads * 64, AWT_ERR_INVALID, "encoder_create: head_dim (d_model / n_heads) must be 64");
  AWT_REQUIRE(cfg->ffn_dim > 0 && cfg->ffn_dim % 128 == 0, AWT_ERR_INVALID, "encoder_create: ffn_dim must be a multiple of 128");
  AWT_REQUIRE(cfg->n_mels > 0 && cfg->n_mels % 8 == 0 && 3 * cfg->n_mels <= kConv1K, AWT_ERR_INVALID, "encoder_create: n_mels must be a multiple of 8, <= 80");
  AWT_REQUIRE(cfg->n_layers > 0 && cfg->n_ctx > 0, AWT_ERR_INVALID, "encoder_create: n_layers and n_ctx must be positive");
  AWT_REQUIRE(cfg->mfma_terms == 1 || cfg->mfma_terms == 3, AWT_ERR_INVALID, "encoder_create: mfma_terms must be 1 or 3");
  AWT_REQUIRE(cfg->lora_rank >= 0 && cfg->lora_rank <= 32, AWT_ERR_INVALID, "encoder_create: lora_rank must be in 0..32");
  AWT_REQUIRE(cfg->lora_rank == 0 || cfg->lora_targets != 0, AWT_ERR_INVALID, "encoder_create: lora_rank > 0 needs lora_targets");
  awt_encoder* e = new awt_encoder();
  e->ctx = c; e->cfg = *cfg; e->planes = cfg->mfma_terms == 3 ? 2 : 1;
  e->chunk = cfg->chunk_clips > 0 ? cfg->chunk_clips : 16;
  const int d = cfg->d_model, f = cfg->ffn_dim;
  int rc = alloc_linear(e, &e->conv1, d, kConv1K);
  if (!rc) rc = alloc_linear(e, &e->conv2, d, 3 * d);
  if (!rc) rc = dev_alloc(e, (void**)&e->pos, (size_t)cfg->n_ctx * d * 4);
  if (!rc) rc = dev_alloc(e, (void**)&e->lnf_g, (size_t)d * 4);
  if (!rc) rc = dev_alloc(e, (void**)&e->lnf_b, (size_t)d * 4);
  e->layers.resize(cfg->n_layers);
  const bool lora = cfg->lora_rank > 0;
  for (int i = 0; i < cfg->n_layers && !rc; ++i) {
    Layer& L = e->layers[i];
    rc = alloc_linear(e, &L.qkv, 3 * d, d);
    if (!rc) rc = alloc_linear(e, &L.out, d, d);
    if (!rc) rc = alloc_linear(e, &L.fc1, f, d);
    if (!rc) rc = alloc_linear(e, &L.fc2, d, f);
    float** lnp[4] = {&L.ln1_g, &L.ln1_b, &L.ln2_g, &L.ln2_b};
    for (int k = 0; k < 4 && !rc; ++k) rc = dev_alloc(e, (void**)lnp[k], (size_t)d * 4);
    if (lora && !rc && (cfg->lora_targets & (AWT_LORA_Q | AWT_LORA_K | AWT_LORA_V))) rc = alloc_lora(e, &L.lq, 3, d, 3 * d);
    if (lora && !rc && (cfg->lora_targets & AWT_LORA_OUT)) rc = alloc_lora(e, &L.lo_, 1, d, d);
    if (lora && !rc && (cfg->lora_targets & AWT_LORA_FC1)) rc = alloc_lora(e, &L.l1, 1, d, f);
    if (lora && !rc && (cfg->lora_targets & AWT_LORA_FC2)) rc = alloc_lora(e, &L.l2, 1, f, d);
  }
  if (rc) { awt_encoder_destroy(e); return rc; }
  *out = e;
  return AWT_OK;
}

extern "C" void awt_encoder_destroy(awt_encoder* e) {
  if (!e) return;
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

extern "C" int awt_encoder_set_weight(awt_encoder* e, const char* name, const float* data, const int64_t* shape, int rank,
                                      void* stream) {
  AWT_REQUIRE(e && name && data && shape && rank >= 1 && rank <= 3, AWT_ERR_INVALID, "set_weight: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const awt_encoder_cfg& c = e->cfg;
  const int d = c.d_model, f = c.ffn_dim, r = c.lora_rank;
  int rc = AWT_ERR_INVALID;
  auto pack = [&](const Planes& pl, int N, int C, int taps, int row_off, int col_off) {
    return launch_pack_weight(e->ctx, data, N, C, taps, pl.ld, row_off, col_off, 1.0f, pl.hi, pl.lo, s);
  };
  std::string nm(name);
  if (nm == "conv1.weight") { rc = check_shape(name, shape, rank, {d, c.n_mels, 3}); if (!rc) rc = pack(e->conv1.w, d, c.n_mels, 3, 0, 0); }
  else if (nm == "conv1.bias") { rc = check_shape(name, shape, rank, {d}); if (!rc) rc = copy_f32(e->conv1.bias, data, d, s); }
  else if (nm == "conv2.weight") { rc = check_shape(name, shape, rank, {d, d, 3}); if (!rc) rc = pack(e->conv2.w, d, d, 3, 0, 0); }
  else if (nm == "conv2.bias") { rc = check_shape(name, shape, rank, {d}); if (!rc) rc = copy_f32(e->conv2.bias, data, d, s); }
  else if (nm == "embed_positions.weight") { rc = check_shape(name, shape, rank, {c.n_ctx, d}); if (!rc) rc = copy_f32(e->pos, data, (size_t)c.n_ctx * d, s); }
  else if (nm == "layer_norm.weight") { rc = check_shape(name, shape, rank, {d}); if (!rc) rc = copy_f32(e->lnf_g, data, d, s); }
  else if (nm == "layer_norm.bias") { rc = check_shape(name, shape, rank, {d}); if (!rc) rc = copy_f32(e->lnf_b, data, d, s); }
  else {
    int li = -1; const char* rest = nullptr;
    if (!parse_layer(name, &li, &rest) || li < 0 || li >= c.n_layers)
      return awt_fail(AWT_ERR_INVALID, std::string("set_weight: unknown parameter ") + name);
    Layer& L = e->layers[li];
    std::string rs(rest);
    struct Proj { const char* key; Linear* lin; int row_off; int N; int K; LoraGroup* lg; int slot; uint32_t bit; };
    Proj projs[] = {
        {"self_attn.q_proj", &L.qkv, 0, d, d, &L.lq, 0, AWT_LORA_Q},   {"self_attn.k_proj", &L.qkv, d, d, d, &L.lq, 1, AWT_LORA_K},
        {"self_attn.v_proj", &L.qkv, 2 * d, d, d, &L.lq, 2, AWT_LORA_V}, {"self_attn.out_proj", &L.out, 0, d, d, &L.lo_, 0, AWT_LORA_OUT},
        {"fc1", &L.fc1, 0, f, d, &L.l1, 0, AWT_LORA_FC1},              {"fc2", &L.fc2, 0, d, f, &L.l2, 0, AWT_LORA_FC2}};
    bool found = false;
    for (const Proj& p : projs) {
      const std::string key(p.key);
      if (rs == key + ".weight") { found = true; rc = check_shape(name, shape, rank, {p.N, p.K}); if (!rc) rc = pack(p.lin->w, p.N, p.K, 1, p.row_off, 0); }
      else if (rs == key + ".bias") {
        found = true;
        if (key == "self_attn.k_proj") return awt_fail(AWT_ERR_INVALID, "set_weight: k_proj has no bias (HF:modeling_whisper.py:279)");
        rc = check_shape(name, shape, rank, {p.N}); if (!rc) rc = copy_f32(p.lin->bias + p.row_off, data, p.N, s);
      } else if (rs == key + ".lora_A" || rs == key + ".lora_B") {
        found = true;
        if (r == 0 || !(c.lora_targets & p.bit) || !p.lg->active)
          return awt_fail(AWT_ERR_STATE, std::string("set_weight: ") + name + " given but this adapter is not enabled in awt_encoder_cfg");
        if (rs.back() == 'A') { rc = check_shape(name, shape, rank, {r, p.K}); if (!rc) rc = pack(p.lg->a, r, p.K, 1, p.slot * r, 0); }
        else { rc = check_shape(name, shape, rank, {p.N, r}); if (!rc) rc = pack(p.lg->b, p.N, r, 1, p.row_off, p.slot * r); }
      }
      if (found) break;
    }
    if (!found) {
      struct Vec { const char* key; float* dst; };
      Vec vecs[] = {{"self_attn_layer_norm.weight", L.ln1_g}, {"self_attn_layer_norm.bias", L.ln1_b},
                    {"final_layer_norm.weight", L.ln2_g}, {"final_layer_norm.bias", L.ln2_b}};
      for (const Vec& v : vecs)
        if (rs == v.key) { found = true; rc = check_shape(name, shape, rank, {d}); if (!rc) rc = copy_f32(v.dst, data, d, s); break; }
    }
    if (!found) return awt_fail(AWT_ERR_INVALID, std::string("set_weight: unknown parameter ") + name);
  }
  if (rc == AWT_OK && std::find(e->seen.begin(), e->seen.end(), nm) == e->seen.end()) e->seen.push_back(nm);
  return rc;
}

static int require_weights(const awt_encoder* e) {
  // 5 stem/table entries + 2 final LN + 15 per layer (k_proj has no bias) must have been uploaded; adapters are optional
  size_t base = 0;
  for (const std::string& n : e->seen) if (n.find("lora_") == std::string::npos) ++base;
  const size_t want = 7 + 15 * (size_t)e->cfg.n_layers;
  if (base < want)
    return awt_fail(AWT_ERR_STATE, "encoder: " + std::to_string(base) + " of " + std::to_string(want) + " parameters uploaded; call awt_encoder_set_weight for every state-dict key first");
  return AWT_OK;
}

extern "C" size_t awt_encoder_workspace_bytes(const awt_encoder* e, int B) {
  if (!e || B <= 0) return 0;
  return carve(e, nullptr, std::min(B, e->chunk)).bytes;
}

extern "C" int awt_encoder_forward(awt_encoder* e, const float* mel, int B, int n_frames, float* hidden, void* workspace,
                                   size_t ws_bytes, void* stream) {
  AWT_REQUIRE(e && mel && hidden && workspace && B > 0, AWT_ERR_INVALID, "encoder_forward: bad argument");
  const int T = 2 * e->cfg.n_ctx;
  if (n_frames != T)  // HF:modeling_whisper.py:612-616
    return awt_fail(AWT_ERR_VALUE, "Whisper expects the mel input features to be of length " + std::to_string(T) + ", but found " +
                                       std::to_string(n_frames) + ". Make sure to pad the input mel features to " + std::to_string(T) + ".");
  int rc = require_weights(e); if (rc) return rc;
  AWT_REQUIRE(ws_bytes >= awt_encoder_workspace_bytes(e, B), AWT_ERR_WORKSPACE, "encoder_forward: workspace too small");
  AWT_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)mel & 15) == 0 && ((uintptr_t)hidden & 15) == 0, AWT_ERR_INVALID,
              "encoder_forward: workspace must be 256-byte aligned, tensors 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += e->chunk) {
    const int Bc = std::min(e->chunk, B - b0);
    rc = forward_chunk(e, mel + (size_t)b0 * e->cfg.n_mels * T, Bc, hidden + (size_t)b0 * e->cfg.n_ctx * e->cfg.d_model,
                       (char*)workspace, s);
    if (rc) return rc;
  }
  return AWT_OK;
}

extern "C" size_t awt_audio_encode_workspace_bytes(const awt_encoder* e, int B) {
  if (!e || B <= 0) return 0;
  const int Bc = std::min(B, e->chunk);
  return carve(e, nullptr, Bc).bytes + align_up((size_t)Bc * e->cfg.n_mels * 2 * e->cfg.n_ctx * 4) + awt_logmel_workspace_bytes(Bc);
}

extern "C" int awt_audio_encode(awt_encoder* e, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                                int max_valid, int B, float* features_out, float* hidden, void* workspace, size_t ws_bytes,
                                void* stream) {
  AWT_REQUIRE(e && pcm && hidden && workspace && B > 0, AWT_ERR_INVALID, "audio_encode: bad argument");
  AWT_REQUIRE(e->cfg.n_mels == 80, AWT_ERR_INVALID, "audio_encode: the Whisper front-end has 80 mel bins");
  int rc = require_weights(e); if (rc) return rc;
  AWT_REQUIRE(ws_bytes >= awt_audio_encode_workspace_bytes(e, B), AWT_ERR_WORKSPACE, "audio_encode: workspace too small");
  AWT_REQUIRE(((uintptr_t)workspace & 255) == 0, AWT_ERR_INVALID, "audio_encode: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int T = 2 * e->cfg.n_ctx;
  const int chunk = std::min(B, e->chunk);
  char* base = (char*)workspace;
  const size_t enc_bytes = carve(e, nullptr, chunk).bytes;
  float* mel_buf = (float*)(base + enc_bytes);
  void* lm_ws = base + enc_bytes + align_up((size_t)chunk * 80 * T * 4);
  const size_t esz = pcm_is_i16 ? 2 : 4;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int Bc = std::min(chunk, B - b0);
    float* mel = features_out ? features_out + (size_t)b0 * 80 * T : mel_buf;
    rc = logmel_whisper_impl(e->ctx, (const char*)pcm + (size_t)b0 * pcm_stride * esz, pcm_is_i16, pcm_stride,
                             n_valid ? n_valid + b0 : nullptr, max_valid, Bc, T, mel, lm_ws, awt_logmel_workspace_bytes(Bc), s);
    if (rc) return rc;
    rc = forward_chunk(e, mel, Bc, hidden + (size_t)b0 * e->cfg.n_ctx * e->cfg.d_model, base, s);
    if (rc) return rc;
  }
  return AWT_OK;
}

// ------------------------------------------------------------------------------------------------ single operators
extern "C" size_t awt_op_linear_workspace_bytes(int M, int N, int K) {
  return 2 * align_up((size_t)M * K * 2) + 2 * align_up((size_t)N * K * 2);
}
extern "C" int awt_op_linear(awt_ctx* c, const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                             int terms, void* workspace, size_t ws_bytes, void* stream) {
  AWT_REQUIRE(c && x && w && y && workspace, AWT_ERR_INVALID, "op_linear: null argument");
  AWT_REQUIRE(M > 0 && N > 0 && N % 128 == 0 && K > 0 && K % 64 == 0, AWT_ERR_INVALID, "op_linear: N % 128 == 0 and K % 64 == 0 required");
  AWT_REQUIRE(ws_bytes >= awt_op_linear_workspace_bytes(M, N, K), AWT_ERR_WORKSPACE, "op_linear: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  bf16_t* xh = (bf16_t*)base;                 base += align_up((size_t)M * K * 2);
  bf16_t* xl = (bf16_t*)base;                 base += align_up((size_t)M * K * 2);
  bf16_t* wh = (bf16_t*)base;                 base += align_up((size_t)N * K * 2);
  bf16_t* wl = (bf16_t*)base;
  int rc = launch_split_f32(c, x, (int64_t)M * K, xh, xl, s); if (rc) return rc;
  rc = launch_split_f32(c, w, (int64_t)N * K, wh, wl, s); if (rc) return rc;
  GemmSeg sg{};
  sg.a_hi = xh; sg.a_lo = xl; sg.lda = K; sg.w_hi = wh; sg.w_lo = wl; sg.ldw = K; sg.K = K;
  sg.rows_out = M; sg.rows_in = M; sg.row_mul = 1; sg.row_add = 0;
  GemmOut o{}; o.f32 = y; o.ldo = N; o.bias = bias; o.n_valid = N;
  return launch_gemm(c, M, N, &sg, 1, terms, EPI_F32, o, s);
}

extern "C" int awt_op_layernorm(awt_ctx* c, const float* x, const float* gamma, const float* beta, float* y, int M, int d,
                                float eps, void* stream) {
  return launch_layernorm(c, x, gamma, beta, M, d, eps, y, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" size_t awt_op_attention_workspace_bytes(int B, int H, int S) { return 6 * align_up((size_t)B * H * S * 64 * 2); }
extern "C" int awt_op_attention(awt_ctx* c, const float* q, const float* k, const float* v, float* o, int B, int H, int S,
                                int terms, void* workspace, size_t ws_bytes, void* stream) {
  AWT_REQUIRE(c && q && k && v && o && workspace, AWT_ERR_INVALID, "op_attention: null argument");
  AWT_REQUIRE(ws_bytes >= awt_op_attention_workspace_bytes(B, H, S), AWT_ERR_WORKSPACE, "op_attention: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)B * H * S * 64;
  const size_t pb = align_up((size_t)n * 2);
  bf16_t* pl[6];
  for (int i = 0; i < 6; ++i) pl[i] = (bf16_t*)((char*)workspace + i * pb);
  const float* src[3] = {q, k, v};
  for (int i = 0; i < 3; ++i) { int rc = launch_split_f32(c, src[i], n, pl[2 * i], pl[2 * i + 1], s); if (rc) return rc; }
  return launch_attention(c, pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], nullptr, nullptr, o, B, H, S, terms, s);
}
