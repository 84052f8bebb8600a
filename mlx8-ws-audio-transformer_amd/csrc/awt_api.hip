// libawt C-ABI (include/awt.h): handles, weight upload, encoder forward orchestration, profiling hooks.
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "common.h"
#include "comm.h"

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
  double flops[AWT_PROF_NCLASSES] = {};
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

static int g_pp_mask = 12;   // tuning knob "gemm_pp_mask": which of a layer's four projections may take the ping-pong GEMM (default: fc1 + fc2, profiles/r04_gemm_pp16_masks.txt)
extern "C" int awt_tuning_set(const char* key, int value) {
  AWT_REQUIRE(key, AWT_ERR_INVALID, "tuning_set: null key");
  if (!strcmp(key, "gemm_tile")) {
    AWT_REQUIRE(value == 0 || value == 64 || value == 128 || value == 256 || value == 512, AWT_ERR_INVALID, "tuning_set: gemm_tile must be 0 (auto), 64, 128, 256 or 512");
    awt_gemm_force_tile(value);
    return AWT_OK;
  }
  if (!strcmp(key, "gemm_gm")) {
    AWT_REQUIRE(value >= 0 && value <= 64, AWT_ERR_INVALID, "tuning_set: gemm_gm must be 0 (default) .. 64");
    awt_gemm_set_gm(value);
    return AWT_OK;
  }
  if (!strcmp(key, "gemm_pp")) {
    AWT_REQUIRE(value >= 0 && value <= 2, AWT_ERR_INVALID, "tuning_set: gemm_pp must be 0 (off, default), 1 (automatic) or 2 (wherever supported)");
    awt_gemm_set_pp_mode(value);
    return AWT_OK;
  }
  if (!strcmp(key, "gemm_pp_mask")) {
    AWT_REQUIRE(value >= 0 && value <= 15, AWT_ERR_INVALID, "tuning_set: gemm_pp_mask is a bit set 0 .. 15 (1 qkv, 2 out_proj, 4 fc1, 8 fc2; fc2 only together with fc1)");
    g_pp_mask = value;
    return AWT_OK;
  }
  if (!strcmp(key, "gemm_pp_stagger")) {
    AWT_REQUIRE(value >= 0 && value <= 16, AWT_ERR_INVALID, "tuning_set: gemm_pp_stagger must be 0 (off) .. 16");
    awt_gemm_set_pp_stagger(value);
    return AWT_OK;
  }
  if (!strcmp(key, "gemm_mfma16")) {
    AWT_REQUIRE(value == 0 || value == 1, AWT_ERR_INVALID, "tuning_set: gemm_mfma16 must be 1 (default: the f16f8 GEMM's 16 x 16 MFMA form where it applies) or 0 (32 x 32 only)");
    awt_gemm_set_mfma16(value);
    return AWT_OK;
  }
  if (!strcmp(key, "attn_shape")) {
    AWT_REQUIRE(value >= 0 && value <= 9, AWT_ERR_INVALID, "tuning_set: attn_shape must be 0 (auto) or 1 .. 9");
    awt_attn_force_shape(value);
    return AWT_OK;
  }
  AWT_REQUIRE(false, AWT_ERR_INVALID, "tuning_set: unknown key");
}

extern "C" int awt_prof_enable(awt_ctx* c, int on) {
  AWT_REQUIRE(c, AWT_ERR_INVALID, "prof_enable: null ctx");
  if (!c->prof) c->prof = new awt_prof_state();
  c->prof_on = on;   // bit k enables kernel class k (AWT_PROF_*)
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
  if (hipMalloc(&c->zeros, 256) != hipSuccess || hipMemset(c->zeros, 0, 256) != hipSuccess) {
    delete c;
    AWT_REQUIRE(false, AWT_ERR_HIP, "ctx_create: could not allocate the zero page");
  }
  // the two Whisper front-ends' tables (DFT basis of 400, Slaney banks of 80 and 128 bins) are built here, so that no compute call
  // of the hot path allocates or synchronises; other front-end configurations: awt_logmel_prepare / awt_resample_prepare
  int rc = logmel_prepare_impl(c, 400, 80, 0.0, 8000.0, 16000, 1);
  if (!rc) rc = logmel_prepare_impl(c, 400, 128, 0.0, 8000.0, 16000, 1);
  if (rc) { awt_ctx_destroy(c); return rc; }
  *out = c;
  return AWT_OK;
}
extern "C" int awt_logmel_prepare(awt_ctx* c, int n_fft, int n_mels, float f_min, float f_max, int sample_rate, int slaney) {
  return logmel_prepare_impl(c, n_fft, n_mels, (double)f_min, (double)f_max, sample_rate, slaney);
}
extern "C" int awt_resample_prepare(awt_ctx* c, int sr_in, int sr_out) { return resample_prepare_impl(c, sr_in, sr_out); }
extern "C" void awt_ctx_destroy(awt_ctx* c) {
  if (!c) return;
  awt_free_tables(c);
  if (c->zeros) (void)hipFree(c->zeros);
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
  return logmel_whisper_impl(c, pcm, pcm_is_i16, pcm_stride, n_valid, max_valid, B, n_frames_out, 80, out, workspace, ws_bytes,
                             (hipStream_t)stream);
}
extern "C" int awt_logmel_whisper_mels(awt_ctx* c, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                                       int max_valid, int B, int n_frames_out, int n_mels, float* out, void* workspace, size_t ws_bytes,
                                       void* stream) {
  return logmel_whisper_impl(c, pcm, pcm_is_i16, pcm_stride, n_valid, max_valid, B, n_frames_out, n_mels, out, workspace, ws_bytes,
                             (hipStream_t)stream);
}
extern "C" int awt_logmel_generic(awt_ctx* c, const float* pcm, int64_t pcm_stride, int B, int n_samples, int sample_rate,
                                  int n_fft, int hop, int n_mels, float f_min, float f_max, float log_eps, float* out,
                                  void* stream) {
  return logmel_generic_impl(c, pcm, pcm_stride, B, n_samples, sample_rate, n_fft, hop, n_mels, f_min, f_max, log_eps, out,
                             (hipStream_t)stream);
}

extern "C" int64_t awt_resampled_length(int n_in, int sr_in, int sr_out) {
  if (n_in <= 0 || sr_in <= 0 || sr_out <= 0) return 0;
  return resampled_length(n_in, sr_in, sr_out);
}
extern "C" int awt_prepare_waveform(awt_ctx* c, const void* pcm, int pcm_is_i16, int channels, int64_t channel_stride,
                                    int64_t sample_stride, int n_in, int sr_in, int sr_out, float* out, int n_out, void* stream) {
  return prepare_waveform_impl(c, pcm, pcm_is_i16, channels, channel_stride, sample_stride, n_in, sr_in, sr_out, out, n_out,
                               (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ encoder
namespace {

struct Planes {  // a weight matrix as operand planes owned by the library: hi (+ lo); f16f8: hi = fp16, lo = hi8 and x8 = lo8 planes
  bf16_t* hi = nullptr; bf16_t* lo = nullptr; uint8_t* x8 = nullptr; int64_t rows = 0, ld = 0;
  // PREC_F16F8 inference: one device flag per separately uploaded row block (q | k | v): "a weight of this block is not exactly fp16";
  // exact16 = no flag set = the lo8 image is all zero and the GEMM drops that cross term (gemm.hip, WX)
  int* d_inexact = nullptr; bool exact16 = false;
  int64_t s_rows = 0;                             // rows s16 / s8 are allocated for: `rows` rounded up to whole 256-column tiles
  bf16_t* s16 = nullptr; uint8_t* s8 = nullptr;   // f16f8: the 16-row fragment copies the 16 x 16 MFMA form of the GEMM reads (w_frag_index / w8s_index; common.h)
  char* pp = nullptr;   // PREC_F16F8 inference: the same matrix in the packed region image of the ping-pong GEMM (gemm_pp.h), N % 256 == 0 only
};
struct Linear {
  Planes w; float* bias = nullptr; int N = 0, K = 0;
};
struct LoraGroup {   // adapters that share one input (q/k/v share LN1's output)
  bool active = false;
  Planes a;          // [128, K_in]: rows slot * r .. slot * r + r - 1 hold adapter `slot`'s A
  Planes b;          // [N_out, kp]: columns slot * r .. hold B (pre-scaling is applied to u, not to B)
  Planes bT;         // training: [128, N_out] = B^T  (weight of du = dy B)
  Planes aT;         // training: [K_in, kp]  = (alpha / r) A^T  (weight of the adapter term of dx)
  int kp = 0;
};
struct Layer {
  Linear qkv, out, fc1, fc2;
  Planes qkvT, outT, fc1T, fc2T;   // training: transposed copies [K, N] of the frozen weights (dX = dY W)
  Planes fc1T8, fc2T8;             // backward_terms = 5: the MLP's transposed copies in the f16f8 weight format (the MLP's two backward GEMMs run in it)
  Planes fc1_8, fc2_8;             // ... and its forward weights in the same format (the training forward's fc1 / fc2 run in it too)
  float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  LoraGroup lq, lo_, l1, l2;
};

// conv1 as a GEMM: K = 3 taps x n_mels, zero-padded to a multiple of 64 (80 mels: 256, 128 mels: 384)
int conv1_k(int n_mels) { return (3 * n_mels + 63) / 64 * 64; }
constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }

}  // namespace

struct awt_encoder {
  awt_ctx* ctx = nullptr;
  awt_encoder_cfg cfg{};
  int planes = 1;          // 2-byte-per-element planes per matrix: 1 (bf16) or 2 (hi + lo; f16f8: fp16 + the two e4m3 planes)
  int prec = PREC_BF16X3;  // operand precision of the forward pass (cfg.mfma_terms: common.h PREC_*)
  Linear conv1, conv2;
  float* pos = nullptr;    // [S, d]
  float *lnf_g = nullptr, *lnf_b = nullptr;
  std::vector<Layer> layers;
  std::vector<void*> allocs;
  std::vector<std::string> seen;   // names uploaded so far
  int chunk = 16;
  awt_comm* comm = nullptr;        // AWT_BWD_ALLREDUCE: adapter gradients are averaged over this communicator's ranks
  int comm_groups = 2;
  bool mlp_f8 = false;             // cfg.backward_terms == 5: the MLP's backward GEMMs in f16f8 (the others in split-bf16)
  float* wt_tmp = nullptr;         // [ffn_dim * d_model] fp32: W^T of the matrix being uploaded (packed into fc1T8 / fc2T8)
  int grad_scale_log2 = 0;         // awt_encoder_set_grad_scale_log2: the backward pass carries 2^k x the gradient (fp16 planes), adapter gradients leave unscaled
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
  if (e->prec == PREC_F16F8) {
    pl->x8 = (uint8_t*)pl->lo + (size_t)rows * ld;
    pl->s_rows = (rows + 255) / 256 * 256;
    rc = dev_alloc(e, (void**)&pl->s16, (size_t)pl->s_rows * ld * 2); if (rc) return rc;
    rc = dev_alloc(e, (void**)&pl->s8, (size_t)pl->s_rows * ld * 2); if (rc) return rc;
  }
  return AWT_OK;
}
int alloc_planes_f8(awt_encoder* e, Planes* pl, int64_t rows, int64_t ld) {   // fp16 plane + the two e4m3 planes, whatever the forward precision
  pl->rows = rows; pl->ld = ld;
  int rc = dev_alloc(e, (void**)&pl->hi, (size_t)rows * ld * 2); if (rc) return rc;
  rc = dev_alloc(e, (void**)&pl->lo, (size_t)rows * ld * 2); if (rc) return rc;
  pl->x8 = (uint8_t*)pl->lo + (size_t)rows * ld;
  pl->s_rows = (rows + 255) / 256 * 256;
  rc = dev_alloc(e, (void**)&pl->s16, (size_t)pl->s_rows * ld * 2); if (rc) return rc;
  rc = dev_alloc(e, (void**)&pl->s8, (size_t)pl->s_rows * ld * 2); if (rc) return rc;
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
  rc = alloc_planes(e, &g->b, N_out, g->kp); if (rc) return rc;
  if (e->cfg.training) {
    rc = alloc_planes(e, &g->bT, 128, N_out); if (rc) return rc;
    rc = alloc_planes(e, &g->aT, K_in, g->kp); if (rc) return rc;
  }
  return AWT_OK;
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
  for (size_t p = 0; p < P; ++p) w.a1[p] = (bf16_t*)take(Mt * conv1_k(c.n_mels) * 2);
  for (size_t p = 0; p < P; ++p) w.h1[p] = (bf16_t*)take(Mt * d * 2);
  w.bytes = std::max(off, layer_end);
  // the ping-pong GEMM reads whole 256-row panels of its activation: up to 255 rows past the last buffer's end (never stored)
  if (e->prec == PREC_F16F8 && !c.training) w.bytes += align_up((size_t)256 * std::max(f, (size_t)conv1_k(c.n_mels)) * 4);
  return w;
}

// the two 2-byte planes of an activation buffer of `elems` elements as operand planes of precision `prec`
Act make_act(bf16_t* p0, bf16_t* p1, size_t elems, int prec) {
  Act a;
  a.p16 = p0;
  if (prec == PREC_F16F8) { a.hi8 = (uint8_t*)p1; a.lo8 = a.hi8 ? a.hi8 + elems : nullptr; }
  else a.lo16 = p1;
  return a;
}
Act act_offset(const Act& a, int64_t elems) {
  Act r;
  r.p16 = a.p16 + elems;
  r.lo16 = a.lo16 ? a.lo16 + elems : nullptr;
  r.hi8 = a.hi8 ? a.hi8 + elems : nullptr;
  r.lo8 = a.lo8 ? a.lo8 + elems : nullptr;
  return r;
}
void set_out(GemmOut& o, const Act& a) { o.hi = a.p16; o.lo = a.lo16; o.hi8 = a.hi8; o.lo8 = a.lo8; o.ilv = a.ilv; }
// the same buffer pair as one split-line image (Act::ilv): the two 2-byte planes are adjacent, 4 bytes per element in all
Act make_ilv(bf16_t* p0) { Act a; a.ilv = (char*)p0; return a; }

GemmSeg seg_plain(const Act& a, int64_t lda, const Planes& w, int64_t wcol, int K, int M) {
  GemmSeg s{};
  s.a_hi = a.p16; s.a_lo = a.lo16; s.a8 = a.hi8; s.al8 = a.lo8; s.lda = lda;
  s.w_hi = w.hi; s.w_lo = w.lo; s.w8 = (const uint8_t*)w.lo; s.wl8 = w.x8;
  s.w_ksteps = (int)(w.ld / 32); s.w_k0 = (int)(wcol / 32); s.K = K;
  s.rows_out = M; s.rows_in = M; s.row_mul = 1; s.row_add = 0;
  s.w_exact16 = w.exact16 ? 1 : 0;
  s.a_ilv = a.ilv; s.w_pp = w.pp;
  s.ws16 = w.s16; s.ws8 = w.s8; s.ws_rows = (int)w.s_rows;
  return s;
}
GemmSeg seg_plain(const bf16_t* a_hi, const bf16_t* a_lo, int64_t lda, const Planes& w, int64_t wcol, int K, int M) {   // bf16 planes (backward pass)
  Act a; a.p16 = const_cast<bf16_t*>(a_hi); a.lo16 = const_cast<bf16_t*>(a_lo);
  return seg_plain(a, lda, w, wcol, K, M);
}

// y = x W^T (+ LoRA: [x | u] [W | B]^T with u = (alpha / r) x A^T) with the given epilogue
int linear_with_lora(awt_encoder* e, bf16_t* const u[2], bf16_t* const in[2], int64_t ld_in, const Linear& lin, const LoraGroup& lg,
                     int M, GemmEpilogue epi, GemmOut out, hipStream_t s) {
  const int terms = e->prec;
  const Act ain = make_act(in[0], in[1], (size_t)M * ld_in, terms);
  GemmSeg segs[2];
  int nseg = 1;
  segs[0] = seg_plain(ain, ld_in, lin.w, 0, lin.K, M);
  if (lg.active) {
    GemmSeg us = seg_plain(ain, ld_in, lg.a, 0, lin.K, M);
    const Act au = make_act(u[0], u[1], (size_t)M * 128, terms);
    GemmOut uo{};
    set_out(uo, au); uo.ldo = lg.kp; uo.n_valid = lg.kp;
    uo.scale = e->cfg.lora_alpha / (float)e->cfg.lora_rank;
    int rc = launch_gemm(e->ctx, M, 128, &us, 1, terms, EPI_BF16, uo, s); if (rc) return rc;
    segs[1] = seg_plain(au, lg.kp, lg.b, 0, lg.kp, M);
    nseg = 2;
  }
  out.bias = lin.bias;
  out.n_valid = lin.N;
  return launch_gemm(e->ctx, M, lin.N, segs, nseg, terms, epi, out, s);
}

// y = x W^T on the persistent ping-pong kernel: x as split lines (make_ilv), W's packed image; no adapter
int linear_pp(awt_encoder* e, bf16_t* in0, const Linear& lin, int M, GemmEpilogue epi, GemmOut out, hipStream_t s) {
  GemmSeg sg = seg_plain(make_ilv(in0), lin.K, lin.w, 0, lin.K, M);
  out.bias = lin.bias;
  out.n_valid = lin.N;
  return launch_gemm_pp(e->ctx, M, lin.N, sg, epi, out, s);
}
// whether this linear of an inference forward over M rows runs on the ping-pong kernel (tuning knob "gemm_pp": 0 never, 1 automatic, 2 wherever
// it can): automatic = a launch of at least one 256 x 256 tile per CU, and not the fp16-exact weights the one-cross-term GEMM (gemm.hip, WX) is faster on
bool use_pp(const awt_encoder* e, const Linear& lin, const LoraGroup& lg, int M, int epi) {
  const int mode = awt_gemm_pp_mode();
  if (mode == 0 || e->prec != PREC_F16F8 || !lin.w.pp || lg.active || !gemm_pp_supported(M, lin.N, lin.K, epi)) return false;
  if (mode == 2) return true;
  // automatic: weights that are not fp16-exact, at least one tile per CU, and a tile count that fills its rounds -- the workgroups are persistent and every
  // tile takes the same time, so t tiles on c CUs finish after ceil(t / c) tile times: more than 20 % of that idle (e.g. 282 tiles on 256 CUs: two rounds for
  // 1.1 rounds of work; measured at B = 16: GEMM class 9.83 vs 9.27 ms) loses to the 128 x 256 kernel, whose tiles the dispatcher packs as they come
  const int64_t t = (int64_t)((M + 255) / 256) * (lin.N / 256), c = gemm_pp_slots();
  return !lin.w.exact16 && c > 0 && t >= c && ((t + c - 1) / c) * c * 5 <= t * 6;
}

// Per-layer activation buffers.  Inference: every layer reuses one set (residual stream updated in place).
// Training: one set per layer, kept for awt_encoder_backward.
struct LayerBufs {
  float *x_in, *x_mid, *x_out;            // residual stream before the layer, after attention, after the MLP
  bf16_t *ln1[2], *qkv[2], *att[2], *ln2[2], *pre[2], *ff[2], *u[2];
  bf16_t *uo[2], *u1[2], *u2[2];          // u = (alpha / r) x A^T of the out_proj / fc1 / fc2 adapters (inference: alias u; training: kept)
  float* lse;                             // [B, H, S] or null
};

struct TrainWs {   // carved from the caller's `saved` buffer
  std::vector<LayerBufs> layer;
  bf16_t *a1[2], *h1[2], *ff[2];          // conv-phase scratch, MLP hidden (not needed by the q/k/v-adapter backward)
  float* x_final;
  // backward scratch
  float *dx_a, *dx_b, *dln, *delta, *partial;
  bf16_t *dxp[2], *dpre[2], *datt[2], *dqkv[2], *du[2];
  size_t partial_bytes, bytes;
};

TrainWs carve_train(const awt_encoder* e, char* base, int B) {
  const awt_encoder_cfg& c = e->cfg;
  const size_t P = e->planes, S = c.n_ctx, T = 2 * S, d = c.d_model, f = c.ffn_dim, H = c.n_heads;
  const size_t M = (size_t)B * S, Mt = (size_t)B * T;
  TrainWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
  auto planes = [&](bf16_t* (&dst)[2], size_t elems) { dst[0] = dst[1] = nullptr; for (size_t p = 0; p < P; ++p) dst[p] = (bf16_t*)take(elems * 2); };
  w.layer.resize(c.n_layers);
  float* x = (float*)take(M * d * 4);
  for (int li = 0; li < c.n_layers; ++li) {
    LayerBufs& L = w.layer[li];
    L.x_in = x;
    L.x_mid = (float*)take(M * d * 4);
    L.x_out = (float*)take(M * d * 4);
    x = L.x_out;
    planes(L.ln1, M * d); planes(L.qkv, 3 * M * d); planes(L.att, M * d); planes(L.ln2, M * d); planes(L.pre, M * f);
    planes(L.u, M * 128);
    const bool has_o = e->layers[li].lo_.active, has_1 = e->layers[li].l1.active, has_2 = e->layers[li].l2.active;
    if (has_o) planes(L.uo, M * 128); else { L.uo[0] = L.u[0]; L.uo[1] = L.u[1]; }
    if (has_1) planes(L.u1, M * 128); else { L.u1[0] = L.u[0]; L.u1[1] = L.u[1]; }
    if (has_2) { planes(L.u2, M * 128); planes(L.ff, M * f); } else { L.u2[0] = L.u[0]; L.u2[1] = L.u[1]; L.ff[0] = L.ff[1] = nullptr; }
    L.lse = (float*)take((size_t)B * H * S * 4);
  }
  w.x_final = x;
  planes(w.ff, M * f);
  // conv-phase scratch aliases the backward scratch (disjoint in time)
  const size_t mark = off;
  planes(w.a1, Mt * conv1_k(e->cfg.n_mels)); planes(w.h1, Mt * d);
  const size_t conv_end = off;
  off = mark;
  w.dx_a = (float*)take(M * d * 4); w.dx_b = (float*)take(M * d * 4); w.dln = (float*)take(M * d * 4);
  w.delta = (float*)take((size_t)B * H * S * 4);
  planes(w.dxp, M * d); planes(w.dpre, M * f); planes(w.datt, M * d); planes(w.dqkv, 3 * M * d); planes(w.du, M * 128);
  w.partial_bytes = outer_reduce_partial_bytes((int)M, c.lora_rank > 0 ? c.lora_rank : 1, 1024);   // Y blocks are reduced in chunks of <= 1024 columns
  w.partial = (float*)take(w.partial_bytes);
  w.bytes = std::max(off, conv_end);
  for (int li = 0; li < c.n_layers; ++li) for (int p = 0; p < 2; ++p) if (!w.layer[li].ff[p]) w.layer[li].ff[p] = w.ff[p];   // kept per layer only under an fc2 adapter
  return w;
}

// conv stem (K5-K7): mel [Bc, n_mels, T] -> residual stream x [Bc * S, d] fp32
int conv_stem(awt_encoder* e, const float* mel, int Bc, bf16_t* const a1[2], bf16_t* const h1[2], float* x, hipStream_t s) {
  const awt_encoder_cfg& c = e->cfg;
  const int S = c.n_ctx, T = 2 * S, d = c.d_model, terms = e->prec;
  const int M = Bc * S, Mt = Bc * T;
  const int k1 = conv1_k(c.n_mels);
  const Act aa1 = make_act(a1[0], a1[1], (size_t)Mt * k1, terms), ah1 = make_act(h1[0], h1[1], (size_t)Mt * d, terms);
  int rc = launch_im2col_conv1(e->ctx, mel, Bc, c.n_mels, T, k1, aa1, terms, s); if (rc) return rc;
  {
    GemmSeg sg = seg_plain(aa1, k1, e->conv1.w, 0, k1, Mt);
    GemmOut o{}; set_out(o, ah1); o.ldo = d; o.bias = e->conv1.bias; o.n_valid = d;
    rc = launch_gemm(e->ctx, Mt, d, &sg, 1, terms, EPI_BF16_GELU, o, s); if (rc) return rc;
  }
  GemmSeg sg[3];
  for (int dt = 0; dt < 3; ++dt) {
    sg[dt] = seg_plain(ah1, d, e->conv2.w, (int64_t)dt * d, d, M);
    sg[dt].rows_out = S; sg[dt].rows_in = T; sg[dt].row_mul = 2; sg[dt].row_add = dt - 1;
  }
  GemmOut o{}; o.f32 = x; o.ldo = d; o.bias = e->conv2.bias; o.n_valid = d; o.pos = e->pos; o.rows_pos = S;
  return launch_gemm(e->ctx, M, d, sg, 3, terms, EPI_F32_GELU_POS, o, s);
}

// one transformer layer (K8-K13) on the given buffers; `save` also keeps the MLP pre-activation and the softmax statistics
int encoder_layer(awt_encoder* e, Layer& L, const LayerBufs& b, int Bc, bool save, hipStream_t s) {
  const awt_encoder_cfg& c = e->cfg;
  const int S = c.n_ctx, d = c.d_model, f = c.ffn_dim, H = c.n_heads, terms = e->prec;
  const int M = Bc * S;
  const int64_t plane = (int64_t)M * d;
  // which of the four linears run on the persistent ping-pong kernel (inference only): their inputs are then written as split lines by the
  // producer (LayerNorm, the attention epilogue, fc1's GELU epilogue) instead of as three planes -- same bytes, same buffers
  const bool pp_qkv = !save && (g_pp_mask & 1) && use_pp(e, L.qkv, L.lq, M, EPI_QKV), pp_out = !save && (g_pp_mask & 2) && use_pp(e, L.out, L.lo_, M, EPI_F32_RESID);
  const bool pp_fc1 = !save && (g_pp_mask & 4) && use_pp(e, L.fc1, L.l1, M, EPI_BF16_GELU), pp_fc2 = pp_fc1 && (g_pp_mask & 8) && use_pp(e, L.fc2, L.l2, M, EPI_F32_RESID);
  // an activation whose only consumer is a GEMM on fp16-exact weights (gemm.hip, WX) needs no hi8 image: the producers skip that plane
  auto feeds = [&](Act a, const Linear& lin, const LoraGroup& lg) { if (terms == PREC_F16F8 && lin.w.exact16 && !lg.active) a.hi8 = nullptr; return a; };
  const Act aqkv = make_act(b.qkv[0], b.qkv[1], 3 * (size_t)plane, terms);
  const Act aatt = pp_out ? make_ilv(b.att[0]) : feeds(make_act(b.att[0], b.att[1], (size_t)plane, terms), L.out, L.lo_);
  int rc = launch_layernorm(e->ctx, b.x_in, L.ln1_g, L.ln1_b, M, d, 1e-5f, nullptr,
                            pp_qkv ? make_ilv(b.ln1[0]) : feeds(make_act(b.ln1[0], b.ln1[1], (size_t)plane, terms), L.qkv, L.lq), terms, s); if (rc) return rc;
  {
    GemmOut o{}; set_out(o, aqkv); o.scale = 0.125f * 1.4426950408889634f;   // head_dim^-1/2 and log2(e): see attention.hip
    o.S = S; o.H = H; o.plane_stride = plane;
    o.skip_v8 = terms == PREC_F16F8 && !attention_f16f8_reads_v8(save);
    rc = pp_qkv ? linear_pp(e, b.ln1[0], L.qkv, M, EPI_QKV, o, s) : linear_with_lora(e, b.u, b.ln1, d, L.qkv, L.lq, M, EPI_QKV, o, s);
    if (rc) return rc;
  }
  if (terms == PREC_F16F8) {
    const Act ak = act_offset(aqkv, plane), av = act_offset(aqkv, 2 * plane);
    rc = launch_attention_f16f8(e->ctx, F8Planes{aqkv.p16, aqkv.hi8, aqkv.lo8}, F8Planes{ak.p16, ak.hi8, ak.lo8}, F8Planes{av.p16, av.hi8, av.lo8},
                                F8Planes{aatt.p16, aatt.hi8, aatt.lo8}, nullptr, save ? b.lse : nullptr, Bc, H, S, s, aatt.ilv);
  } else {
    rc = launch_attention(e->ctx, b.qkv[0], b.qkv[1], b.qkv[0] + plane, b.qkv[1] ? b.qkv[1] + plane : nullptr, b.qkv[0] + 2 * plane,
                          b.qkv[1] ? b.qkv[1] + 2 * plane : nullptr, b.att[0], b.att[1], nullptr, save ? b.lse : nullptr, Bc, H, S, terms, s);
  }
  if (rc) return rc;
  {
    GemmOut o{}; o.f32 = b.x_mid; o.resid = b.x_in; o.ldo = d;
    rc = pp_out ? linear_pp(e, b.att[0], L.out, M, EPI_F32_RESID, o, s) : linear_with_lora(e, b.uo, b.att, d, L.out, L.lo_, M, EPI_F32_RESID, o, s);
    if (rc) return rc;
  }
  if (save && e->mlp_f8) {   // backward_terms = 5: the MLP of the training step in the f16f8 operand format (no fc1 / fc2 adapters in this mode)
    const Act aln2 = make_act(b.ln2[0], b.ln2[1], (size_t)plane, PREC_F16F8), aff = make_act(b.ff[0], b.ff[1], (size_t)M * f, PREC_F16F8);
    rc = launch_layernorm(e->ctx, b.x_mid, L.ln2_g, L.ln2_b, M, d, 1e-5f, nullptr, aln2, PREC_F16F8, s); if (rc) return rc;
    GemmSeg s1 = seg_plain(aln2, d, L.fc1_8, 0, d, M);
    GemmOut o1{}; set_out(o1, aff); o1.ldo = f; o1.n_valid = f; o1.bias = L.fc1.bias; o1.hi2 = b.pre[0]; o1.lo2 = b.pre[1]; o1.scale = 1.0f;
    rc = launch_gemm(e->ctx, M, f, &s1, 1, PREC_F16F8, EPI_BF16_GELU_SAVE, o1, s); if (rc) return rc;
    GemmSeg s2 = seg_plain(aff, f, L.fc2_8, 0, f, M);
    GemmOut o2{}; o2.f32 = b.x_out; o2.resid = b.x_mid; o2.ldo = d; o2.n_valid = d; o2.bias = L.fc2.bias;
    return launch_gemm(e->ctx, M, d, &s2, 1, PREC_F16F8, EPI_F32_RESID, o2, s);
  }
  rc = launch_layernorm(e->ctx, b.x_mid, L.ln2_g, L.ln2_b, M, d, 1e-5f, nullptr,
                        pp_fc1 ? make_ilv(b.ln2[0]) : feeds(make_act(b.ln2[0], b.ln2[1], (size_t)plane, terms), L.fc1, L.l1), terms, s); if (rc) return rc;
  {
    GemmOut o{}; set_out(o, pp_fc2 ? make_ilv(b.ff[0]) : feeds(make_act(b.ff[0], b.ff[1], (size_t)M * f, terms), L.fc2, L.l2)); o.ldo = f; o.hi2 = b.pre[0]; o.lo2 = b.pre[1];
    rc = pp_fc1 ? linear_pp(e, b.ln2[0], L.fc1, M, EPI_BF16_GELU, o, s)
                : linear_with_lora(e, b.u1, b.ln2, d, L.fc1, L.l1, M, save ? EPI_BF16_GELU_SAVE : EPI_BF16_GELU, o, s);
    if (rc) return rc;
  }
  GemmOut o{}; o.f32 = b.x_out; o.resid = b.x_mid; o.ldo = d;
  return pp_fc2 ? linear_pp(e, b.ff[0], L.fc2, M, EPI_F32_RESID, o, s) : linear_with_lora(e, b.u2, b.ff, f, L.fc2, L.l2, M, EPI_F32_RESID, o, s);
}

int forward_chunk(awt_encoder* e, const float* mel, int Bc, float* hidden, char* ws_base, hipStream_t s) {
  const awt_encoder_cfg& c = e->cfg;
  const int M = Bc * c.n_ctx, d = c.d_model;
  const bool two = e->planes == 2;
  Workspace w = carve(e, ws_base, Bc);
  bf16_t* a1[2] = {w.a1[0], two ? w.a1[1] : nullptr};
  bf16_t* h1[2] = {w.h1[0], two ? w.h1[1] : nullptr};
  int rc = conv_stem(e, mel, Bc, a1, h1, w.x, s); if (rc) return rc;
  LayerBufs b{};
  b.x_in = b.x_mid = b.x_out = w.x;
  for (int p = 0; p < 2; ++p) {
    const bool on = p == 0 || two;
    b.ln1[p] = b.ln2[p] = on ? w.ln[p] : nullptr; b.qkv[p] = on ? w.qkv[p] : nullptr; b.att[p] = on ? w.att[p] : nullptr;
    b.ff[p] = on ? w.ff[p] : nullptr; b.u[p] = b.uo[p] = b.u1[p] = b.u2[p] = on ? w.u[p] : nullptr; b.pre[p] = nullptr;
  }
  for (int li = 0; li < c.n_layers; ++li) { rc = encoder_layer(e, e->layers[li], b, Bc, false, s); if (rc) return rc; }
  return launch_layernorm(e->ctx, w.x, e->lnf_g, e->lnf_b, M, d, 1e-5f, hidden, Act{}, e->prec, s);
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
  AWT_REQUIRE(cfg->d_model > 0 && cfg->d_model % 128 == 0 && cfg->d_model <= 1280, AWT_ERR_INVALID, "encoder_create: d_model must be a multiple of 128, <= 1280");
  AWT_REQUIRE(cfg->n_heads > 0 && cfg->d_model == cfg->n_heads * 64, AWT_ERR_INVALID, "encoder_create: head_dim (d_model / n_heads) must be 64");
  AWT_REQUIRE(cfg->ffn_dim > 0 && cfg->ffn_dim % 128 == 0, AWT_ERR_INVALID, "encoder_create: ffn_dim must be a multiple of 128");
  AWT_REQUIRE(cfg->n_mels > 0 && cfg->n_mels % 8 == 0 && cfg->n_mels <= 128, AWT_ERR_INVALID, "encoder_create: n_mels must be a multiple of 8, <= 128");
  AWT_REQUIRE(cfg->n_layers > 0 && cfg->n_ctx > 0, AWT_ERR_INVALID, "encoder_create: n_layers and n_ctx must be positive");
  AWT_REQUIRE(cfg->mfma_terms == PREC_BF16 || cfg->mfma_terms == PREC_F16 || cfg->mfma_terms == PREC_BF16X3 || cfg->mfma_terms == PREC_F16X3 || cfg->mfma_terms == PREC_F16F8, AWT_ERR_INVALID,
              "encoder_create: mfma_terms must be 1 (bf16), 2 (fp16), 3 (bf16x3), 4 (fp16x3) or 5 (f16f8)");
  AWT_REQUIRE(!cfg->training || cfg->mfma_terms == PREC_BF16 || cfg->mfma_terms == PREC_BF16X3, AWT_ERR_INVALID,
              "encoder_create: training keeps its activations as bf16 planes: mfma_terms must be 1 or 3 (gradients do not fit fp16's range unscaled)");
  AWT_REQUIRE(cfg->backward_terms == 0 || cfg->backward_terms == cfg->mfma_terms || ((cfg->backward_terms == 1 || cfg->backward_terms == PREC_F16F8) && cfg->mfma_terms == 3),
              AWT_ERR_INVALID, "encoder_create: backward_terms must be 0 (= mfma_terms), mfma_terms, 1, or 5 (f16f8 MLP backward) with mfma_terms = 3");
  AWT_REQUIRE(cfg->backward_terms != PREC_F16F8 || (cfg->training && !(cfg->lora_targets & (AWT_LORA_FC1 | AWT_LORA_FC2))), AWT_ERR_INVALID,
              "encoder_create: backward_terms = 5 runs the MLP's backward GEMMs in f16f8: training mode, and no adapters on fc1 / fc2");
  AWT_REQUIRE(cfg->lora_rank >= 0 && cfg->lora_rank <= 32, AWT_ERR_INVALID, "encoder_create: lora_rank must be in 0..32");
  AWT_REQUIRE(cfg->lora_rank == 0 || cfg->lora_targets != 0, AWT_ERR_INVALID, "encoder_create: lora_rank > 0 needs lora_targets");
  AWT_REQUIRE(!cfg->training || cfg->lora_rank > 0, AWT_ERR_INVALID, "encoder_create: training mode needs adapters (lora_rank > 0)");
  awt_encoder* e = new awt_encoder();
  e->ctx = c; e->cfg = *cfg; e->prec = cfg->mfma_terms; e->planes = (cfg->mfma_terms == PREC_BF16 || cfg->mfma_terms == PREC_F16) ? 1 : 2;
  e->chunk = cfg->chunk_clips > 0 ? cfg->chunk_clips : 64;
  e->mlp_f8 = cfg->backward_terms == PREC_F16F8;
  const int d = cfg->d_model, f = cfg->ffn_dim;
  int rc = alloc_linear(e, &e->conv1, d, conv1_k(cfg->n_mels));
  if (!rc && e->mlp_f8) rc = dev_alloc(e, (void**)&e->wt_tmp, (size_t)f * d * 4);
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
    if ((e->prec == PREC_F16F8 || e->prec == PREC_F16X3) && !cfg->training)   // four zero-initialised "not fp16-exact" flags per projection matrix (set_weight)
      for (Linear* lin : {&L.qkv, &L.out, &L.fc1, &L.fc2}) if (!rc) rc = dev_alloc(e, (void**)&lin->w.d_inexact, 4 * sizeof(int));
    if (e->prec == PREC_F16F8 && !cfg->training)     // the packed images of the ping-pong GEMM (a second copy of each weight: 4 bytes per element)
      for (Linear* lin : {&L.qkv, &L.out, &L.fc1, &L.fc2})
        if (!rc && lin->N % 256 == 0 && lin->K % 64 == 0 && lin->K >= 128) rc = dev_alloc(e, (void**)&lin->w.pp, gemm_pp_weight_bytes(lin->N, lin->K));
    if (cfg->training) {
      if (!rc) rc = alloc_planes(e, &L.qkvT, d, 3 * d);
      if (!rc) rc = alloc_planes(e, &L.outT, d, d);
      if (!rc) rc = alloc_planes(e, &L.fc1T, d, f);
      if (!rc) rc = alloc_planes(e, &L.fc2T, f, d);
      if (!rc && e->mlp_f8) rc = alloc_planes_f8(e, &L.fc1T8, d, f);
      if (!rc && e->mlp_f8) rc = alloc_planes_f8(e, &L.fc2T8, f, d);
      if (!rc && e->mlp_f8) rc = alloc_planes_f8(e, &L.fc1_8, f, d);
      if (!rc && e->mlp_f8) rc = alloc_planes_f8(e, &L.fc2_8, d, f);
    }
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
    return launch_pack_weight(e->ctx, data, N, C, taps, pl.ld, row_off, col_off, 1.0f, pl.hi, pl.lo, pl.x8, e->prec, s, nullptr, pl.s16, pl.s8);
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
    struct Proj { const char* key; Linear* lin; int row_off; int N; int K; LoraGroup* lg; int slot; uint32_t bit; Planes* wT; Planes* wT8; Planes* w8; };
    Proj projs[] = {
        {"self_attn.q_proj", &L.qkv, 0, d, d, &L.lq, 0, AWT_LORA_Q, &L.qkvT, nullptr, nullptr},   {"self_attn.k_proj", &L.qkv, d, d, d, &L.lq, 1, AWT_LORA_K, &L.qkvT, nullptr, nullptr},
        {"self_attn.v_proj", &L.qkv, 2 * d, d, d, &L.lq, 2, AWT_LORA_V, &L.qkvT, nullptr, nullptr}, {"self_attn.out_proj", &L.out, 0, d, d, &L.lo_, 0, AWT_LORA_OUT, &L.outT, nullptr, nullptr},
        {"fc1", &L.fc1, 0, f, d, &L.l1, 0, AWT_LORA_FC1, &L.fc1T, &L.fc1T8, &L.fc1_8},            {"fc2", &L.fc2, 0, d, f, &L.l2, 0, AWT_LORA_FC2, &L.fc2T, &L.fc2T8, &L.fc2_8}};
    const float lscale = r > 0 ? c.lora_alpha / (float)r : 0.f;
    bool found = false;
    for (const Proj& p : projs) {
      const std::string key(p.key);
      if (rs == key + ".weight") {
        found = true; rc = check_shape(name, shape, rank, {p.N, p.K});
        if (!rc && (e->prec == PREC_F16F8 || e->prec == PREC_F16X3) && !c.training && p.lin->w.d_inexact) {
          // fp16-exact weights (checkpoints stored in half precision) let the GEMM drop one cross term: find out now, at upload time
          Planes& pl = p.lin->w;
          int* flag = pl.d_inexact + p.row_off / p.N;
          int host[4] = {0, 0, 0, 0};
          if (hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess) rc = awt_fail(AWT_ERR_HIP, "set_weight: flag reset failed");
          if (!rc) rc = launch_pack_weight(e->ctx, data, p.N, p.K, 1, pl.ld, p.row_off, 0, 1.0f, pl.hi, pl.lo, pl.x8, e->prec, s, flag, pl.s16, pl.s8);
          if (!rc && (hipMemcpyAsync(host, pl.d_inexact, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
            rc = awt_fail(AWT_ERR_HIP, "set_weight: flag read-back failed");
          if (!rc) pl.exact16 = !(host[0] | host[1] | host[2] | host[3]);
        } else if (!rc) rc = pack(p.lin->w, p.N, p.K, 1, p.row_off, 0);
        if (!rc && p.lin->w.pp) rc = launch_pack_weight_pp(e->ctx, data, p.N, p.K, p.row_off, p.lin->w.pp, s);
        if (!rc && c.training)   // W^T: [K, N_total], this projection's columns start at row_off
          rc = launch_pack_weight_t(e->ctx, data, p.N, p.K, p.wT->ld, 0, p.row_off, 1.0f, p.wT->hi, p.wT->lo, s);
        if (!rc && e->mlp_f8 && p.wT8) {   // the same W^T [K, N] in the f16f8 weight format: transposed into the scratch matrix, then packed like a forward weight
          rc = launch_transpose_f32(e->ctx, data, p.N, p.K, e->wt_tmp, s);
          if (!rc) rc = launch_pack_weight(e->ctx, e->wt_tmp, p.K, p.N, 1, p.wT8->ld, 0, 0, 1.0f, p.wT8->hi, p.wT8->lo, p.wT8->x8, PREC_F16F8, s, nullptr, p.wT8->s16, p.wT8->s8);
          if (!rc) rc = launch_pack_weight(e->ctx, data, p.N, p.K, 1, p.w8->ld, 0, 0, 1.0f, p.w8->hi, p.w8->lo, p.w8->x8, PREC_F16F8, s, nullptr, p.w8->s16, p.w8->s8);
        }
      }
      else if (rs == key + ".bias") {
        found = true;
        if (key == "self_attn.k_proj") return awt_fail(AWT_ERR_INVALID, "set_weight: k_proj has no bias (HF:modeling_whisper.py:279)");
        rc = check_shape(name, shape, rank, {p.N}); if (!rc) rc = copy_f32(p.lin->bias + p.row_off, data, p.N, s);
      } else if (rs == key + ".lora_A" || rs == key + ".lora_B") {
        found = true;
        if (r == 0 || !(c.lora_targets & p.bit) || !p.lg->active)
          return awt_fail(AWT_ERR_STATE, std::string("set_weight: ") + name + " given but this adapter is not enabled in awt_encoder_cfg");
        if (rs.back() == 'A') {
          rc = check_shape(name, shape, rank, {r, p.K}); if (!rc) rc = pack(p.lg->a, r, p.K, 1, p.slot * r, 0);
          if (!rc && c.training) rc = launch_pack_weight_t(e->ctx, data, r, p.K, p.lg->aT.ld, 0, p.slot * r, lscale, p.lg->aT.hi, p.lg->aT.lo, s);
        } else {
          rc = check_shape(name, shape, rank, {p.N, r}); if (!rc) rc = pack(p.lg->b, p.N, r, 1, p.row_off, p.slot * r);
          if (!rc && c.training) rc = launch_pack_weight_t(e->ctx, data, p.N, r, p.lg->bT.ld, p.slot * r, p.row_off, 1.0f, p.lg->bT.hi, p.lg->bT.lo, s);
        }
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

extern "C" int awt_encoder_exact16_matrices(const awt_encoder* e, int* n_exact, int* n_total) {
  AWT_REQUIRE(e && n_exact && n_total, AWT_ERR_INVALID, "encoder_exact16_matrices: null argument");
  int ex = 0, tot = 0;
  for (const Layer& L : e->layers)
    for (const Linear* lin : {&L.qkv, &L.out, &L.fc1, &L.fc2}) { ++tot; if (lin->w.exact16) ++ex; }
  *n_exact = ex; *n_total = tot;
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
  AWT_REQUIRE(e->cfg.n_mels == 80 || e->cfg.n_mels == 128, AWT_ERR_INVALID, "audio_encode: the Whisper front-end has 80 or 128 (large-v3) mel bins");
  int rc = require_weights(e); if (rc) return rc;
  AWT_REQUIRE(ws_bytes >= awt_audio_encode_workspace_bytes(e, B), AWT_ERR_WORKSPACE, "audio_encode: workspace too small");
  AWT_REQUIRE(((uintptr_t)workspace & 255) == 0, AWT_ERR_INVALID, "audio_encode: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int T = 2 * e->cfg.n_ctx;
  const int chunk = std::min(B, e->chunk);
  char* base = (char*)workspace;
  const size_t enc_bytes = carve(e, nullptr, chunk).bytes;
  float* mel_buf = (float*)(base + enc_bytes);
  void* lm_ws = base + enc_bytes + align_up((size_t)chunk * e->cfg.n_mels * T * 4);
  const size_t esz = pcm_is_i16 ? 2 : 4;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int Bc = std::min(chunk, B - b0);
    float* mel = features_out ? features_out + (size_t)b0 * e->cfg.n_mels * T : mel_buf;
    rc = logmel_whisper_impl(e->ctx, (const char*)pcm + (size_t)b0 * pcm_stride * esz, pcm_is_i16, pcm_stride,
                             n_valid ? n_valid + b0 : nullptr, max_valid, Bc, T, e->cfg.n_mels, mel, lm_ws, awt_logmel_workspace_bytes(Bc), s);
    if (rc) return rc;
    rc = forward_chunk(e, mel, Bc, hidden + (size_t)b0 * e->cfg.n_ctx * e->cfg.d_model, base, s);
    if (rc) return rc;
  }
  return AWT_OK;
}

// ------------------------------------------------------------------------------------------------ single operators
extern "C" size_t awt_op_linear_workspace_bytes(int M, int N, int K) {
  const size_t npad = ((size_t)N + 255) / 256 * 256;      // the 16-row w copies cover whole 256-column tiles
  return 2 * align_up((size_t)M * K * 2) + 2 * align_up((size_t)N * K * 2) + 256 + 2 * align_up(npad * K * 2);     // x planes, w planes, a flag word (fp16-exact weights), the 16-row w copies (f16f8)
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
  if (terms == PREC_F16F6) {   // experimental: fp16 plane + two e3m2 planes (the second 2-byte plane's space holds both, 0.75 B per element each)
#ifndef AWT_EXPERIMENTAL_F6
    return awt_fail(AWT_ERR_INVALID, "op_linear (f16f6): the FP6 cross-term experiment is not part of this build (compile with -DAWT_EXPERIMENTAL_F6)");
#else
    AWT_REQUIRE(N % 256 == 0, AWT_ERR_INVALID, "op_linear (f16f6): N % 256 == 0 required");
    Act ax6; ax6.p16 = xh; ax6.hi8 = (uint8_t*)xl; ax6.lo8 = (uint8_t*)xl + (size_t)M * K / 4 * 3;
    uint8_t* w6 = (uint8_t*)wl; uint8_t* wl6 = w6 + (size_t)N * K / 4 * 3;
    int rc6 = launch_split_planes_f6(c, x, M, K, xh, ax6.hi8, ax6.lo8, s); if (rc6) return rc6;
    rc6 = launch_pack_weight_f6(c, w, N, K, wh, w6, wl6, s); if (rc6) return rc6;
    Planes pw6; pw6.hi = wh; pw6.lo = (bf16_t*)w6; pw6.x8 = wl6; pw6.rows = N; pw6.ld = K;
    GemmSeg sg6 = seg_plain(ax6, K, pw6, 0, K, M);
    GemmOut o6{}; o6.f32 = y; o6.ldo = N; o6.bias = bias; o6.n_valid = N;
    return launch_gemm(c, M, N, &sg6, 1, terms, EPI_F32, o6, s);
#endif
  }
  if (terms == PREC_F16F8 && awt_gemm_pp_mode() == 2 && gemm_pp_supported(M, N, K, EPI_F32)) {
    // the persistent ping-pong kernel (tuning knob "gemm_pp" = 2): x as split lines over the two x planes' space (its 256-row panel reads beyond M
    // stay inside the workspace: the packed weight image of N >= 256 rows follows), the weight in the packed region image over the two w planes' space
    Act ai; ai.ilv = (char*)xh;
    int rcp = launch_split_planes(c, x, (int64_t)M * K, 1.0f, terms, kF8Act, nullptr, nullptr, nullptr, nullptr, s, ai.ilv); if (rcp) return rcp;
    if (hipMemsetAsync(wh, 0, gemm_pp_weight_bytes(N, K), s) != hipSuccess) return awt_fail(AWT_ERR_HIP, "op_linear: weight image reset failed");
    rcp = launch_pack_weight_pp(c, w, N, K, 0, (char*)wh, s); if (rcp) return rcp;
    Planes pwp; pwp.pp = (char*)wh; pwp.rows = N; pwp.ld = K;
    GemmSeg sgp = seg_plain(ai, K, pwp, 0, K, M);
    GemmOut op{}; op.f32 = y; op.ldo = N; op.bias = bias; op.n_valid = N;
    return launch_gemm_pp(c, M, N, sgp, EPI_F32, op, s);
  }
  const Act ax = make_act(xh, xl, (size_t)M * K, terms);
  int rc = launch_split_planes(c, x, (int64_t)M * K, 1.0f, terms, kF8Act, xh, xl, ax.hi8, ax.lo8, s); if (rc) return rc;
  Planes pw; pw.hi = wh; pw.lo = wl; pw.x8 = (uint8_t*)wl + (size_t)N * K; pw.rows = N; pw.ld = K;
  if (terms == PREC_F16F8 || terms == PREC_F16X3) {   // as awt_encoder_set_weight does: fp16-exact weights take the GEMM without the x_hi w_lo product; the flag is the
    int* flag = (int*)((char*)wl + align_up((size_t)N * K * 2));      // word of the workspace's 256-byte tail
    int host = 1;
    if (hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess) return awt_fail(AWT_ERR_HIP, "op_linear: flag reset failed");
    if (terms == PREC_F16F8) {
      pw.s_rows = ((int64_t)N + 255) / 256 * 256;
      pw.s16 = (bf16_t*)((char*)flag + 256); pw.s8 = (uint8_t*)pw.s16 + align_up((size_t)pw.s_rows * K * 2);
      if (N % 256 != 0) {   // the rows beyond N are read by the last column tile (never stored): keep them finite
        if (hipMemsetAsync(pw.s16, 0, 2 * align_up((size_t)pw.s_rows * K * 2), s) != hipSuccess) return awt_fail(AWT_ERR_HIP, "op_linear: weight copy reset failed");
      }
    }
    rc = launch_pack_weight(c, w, N, K, 1, K, 0, 0, 1.0f, wh, wl, pw.x8, terms, s, flag, pw.s16, pw.s8); if (rc) return rc;
    if (hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return awt_fail(AWT_ERR_HIP, "op_linear: flag read-back failed");
    pw.exact16 = host == 0;
  } else {
    rc = launch_pack_weight(c, w, N, K, 1, K, 0, 0, 1.0f, wh, wl, pw.x8, terms, s); if (rc) return rc;     // fragment-major
  }
  GemmSeg sg = seg_plain(ax, K, pw, 0, K, M);
  GemmOut o{}; o.f32 = y; o.ldo = N; o.bias = bias; o.n_valid = N;
  return launch_gemm(c, M, N, &sg, 1, terms, EPI_F32, o, s);
}

extern "C" int awt_op_layernorm(awt_ctx* c, const float* x, const float* gamma, const float* beta, float* y, int M, int d,
                                float eps, void* stream) {
  return launch_layernorm(c, x, gamma, beta, M, d, eps, y, Act{}, PREC_BF16X3, (hipStream_t)stream);
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
  const int f8exp[3] = {kF8Q, kF8KV, kF8KV};
  for (int i = 0; i < 3; ++i) {   // the kernel expects q * log2(e); f16f8: the second 2-byte plane holds the two e4m3 planes
    uint8_t* b8 = (uint8_t*)pl[2 * i + 1];
    int rc = launch_split_planes(c, src[i], n, i == 0 ? 1.4426950408889634f : 1.0f, terms, f8exp[i], pl[2 * i], pl[2 * i + 1], b8, b8 + n, s);
    if (rc) return rc;
  }
  if (terms == PREC_F16F8) {
    F8Planes P[3];
    for (int i = 0; i < 3; ++i) P[i] = F8Planes{pl[2 * i], (uint8_t*)pl[2 * i + 1], (uint8_t*)pl[2 * i + 1] + n};
    return launch_attention_f16f8(c, P[0], P[1], P[2], F8Planes{nullptr, nullptr, nullptr}, o, nullptr, B, H, S, s);
  }
  return launch_attention(c, pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], nullptr, nullptr, o, nullptr, B, H, S, terms, s);
}

// ------------------------------------------------------------------------------------------------ LoRA fine-tune step
extern "C" int awt_encoder_set_grad_scale_log2(awt_encoder* e, int k) {
  AWT_REQUIRE(e && e->cfg.training, AWT_ERR_STATE, "encoder_set_grad_scale_log2: encoder was not created with cfg.training");
  AWT_REQUIRE(k >= -60 && k <= 60, AWT_ERR_INVALID, "encoder_set_grad_scale_log2: k must be in -60 .. 60");
  e->grad_scale_log2 = k;
  return AWT_OK;
}

extern "C" size_t awt_encoder_train_workspace_bytes(const awt_encoder* e, int B) {
  if (!e || B <= 0 || !e->cfg.training) return 0;
  return carve_train(e, nullptr, B).bytes;
}

namespace {
// elements of one layer's adapter gradients: for every enabled target in the order q, k, v, out, fc1, fc2: dA [r, K_in] then dB [N_out, r]
size_t lora_grads_per_layer(const awt_encoder_cfg& c) {
  const size_t r = c.lora_rank, d = c.d_model, f = c.ffn_dim;
  size_t n = 0;
  for (uint32_t bit : {AWT_LORA_Q, AWT_LORA_K, AWT_LORA_V, AWT_LORA_OUT}) if (c.lora_targets & bit) n += 2 * r * d;
  if (c.lora_targets & AWT_LORA_FC1) n += r * d + f * r;
  if (c.lora_targets & AWT_LORA_FC2) n += r * f + d * r;
  return n;
}
}  // namespace

extern "C" size_t awt_encoder_lora_grad_count(const awt_encoder* e) {
  if (!e || e->cfg.lora_rank <= 0) return 0;
  return (size_t)e->cfg.n_layers * lora_grads_per_layer(e->cfg);
}

extern "C" int awt_encoder_forward_train(awt_encoder* e, const float* mel, int B, int n_frames, float* hidden, void* saved,
                                         size_t saved_bytes, void* stream) {
  AWT_REQUIRE(e && mel && hidden && saved && B > 0, AWT_ERR_INVALID, "encoder_forward_train: bad argument");
  AWT_REQUIRE(e->cfg.training, AWT_ERR_STATE, "encoder_forward_train: encoder was not created with cfg.training");
  const int T = 2 * e->cfg.n_ctx;
  if (n_frames != T)
    return awt_fail(AWT_ERR_VALUE, "Whisper expects the mel input features to be of length " + std::to_string(T) + ", but found " +
                                       std::to_string(n_frames) + ". Make sure to pad the input mel features to " + std::to_string(T) + ".");
  int rc = require_weights(e); if (rc) return rc;
  AWT_REQUIRE(saved_bytes >= awt_encoder_train_workspace_bytes(e, B), AWT_ERR_WORKSPACE, "encoder_forward_train: saved buffer too small");
  AWT_REQUIRE(((uintptr_t)saved & 255) == 0, AWT_ERR_INVALID, "encoder_forward_train: saved buffer must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  TrainWs w = carve_train(e, (char*)saved, B);
  rc = conv_stem(e, mel, B, w.a1, w.h1, w.layer[0].x_in, s); if (rc) return rc;
  for (int li = 0; li < e->cfg.n_layers; ++li) { rc = encoder_layer(e, e->layers[li], w.layer[li], B, true, s); if (rc) return rc; }
  return launch_layernorm(e->ctx, w.x_final, e->lnf_g, e->lnf_b, B * e->cfg.n_ctx, e->cfg.d_model, 1e-5f, hidden, Act{}, e->prec, s);
}

extern "C" int awt_encoder_set_comm(awt_encoder* e, awt_comm* m, int groups) {
  AWT_REQUIRE(e, AWT_ERR_INVALID, "encoder_set_comm: null encoder");
  AWT_REQUIRE(groups >= 1, AWT_ERR_INVALID, "encoder_set_comm: groups must be >= 1");
  AWT_REQUIRE(!m || m->ctx == e->ctx, AWT_ERR_INVALID, "encoder_set_comm: communicator and encoder belong to different contexts");
  e->comm = m;
  e->comm_groups = std::min(std::min(groups, 4), e->cfg.n_layers);
  return AWT_OK;
}

extern "C" int awt_encoder_backward(awt_encoder* e, const float* d_hidden, int B, void* saved, size_t saved_bytes, float* lora_grads,
                                    size_t n_grads, void* stream) {
  return awt_encoder_backward_ex(e, d_hidden, B, saved, saved_bytes, lora_grads, n_grads, 0u, stream);
}

extern "C" int awt_encoder_backward_ex(awt_encoder* e, const float* d_hidden, int B, void* saved, size_t saved_bytes, float* lora_grads,
                                       size_t n_grads, uint32_t flags, void* stream) {
  AWT_REQUIRE(e && d_hidden && saved && lora_grads && B > 0, AWT_ERR_INVALID, "encoder_backward: bad argument");
  AWT_REQUIRE(!(flags & ~(uint32_t)(AWT_BWD_ACCUMULATE | AWT_BWD_ALLREDUCE)), AWT_ERR_INVALID, "encoder_backward: unknown flag");
  AWT_REQUIRE(!(flags & AWT_BWD_ALLREDUCE) || e->comm, AWT_ERR_STATE, "encoder_backward: AWT_BWD_ALLREDUCE needs awt_encoder_set_comm first");
  const int accumulate = (flags & AWT_BWD_ACCUMULATE) ? 1 : 0;
  const bool exchange = (flags & AWT_BWD_ALLREDUCE) != 0;
  AWT_REQUIRE(e->cfg.training, AWT_ERR_STATE, "encoder_backward: encoder was not created with cfg.training");
  AWT_REQUIRE(saved_bytes >= awt_encoder_train_workspace_bytes(e, B), AWT_ERR_WORKSPACE, "encoder_backward: saved buffer too small");
  AWT_REQUIRE(n_grads == awt_encoder_lora_grad_count(e), AWT_ERR_INVALID, "encoder_backward: lora_grads has the wrong element count");
  hipStream_t s = (hipStream_t)stream;
  const awt_encoder_cfg& c = e->cfg;
  const int S = c.n_ctx, d = c.d_model, f = c.ffn_dim, H = c.n_heads, terms = c.mfma_terms, r = c.lora_rank;
  // products of the gradient contractions: the forward's (default), or one bf16 product per fragment pair (opt-in fast backward)
  // backward_terms = 5: the MLP's two backward GEMMs run in f16f8 (fp16 + two e4m3 planes, 2 MFMA-equivalents instead of 3), everything else in split-bf16
  const bool mlp8 = e->mlp_f8;
  const int gterms = mlp8 ? PREC_BF16X3 : (c.backward_terms ? c.backward_terms : terms);
  // the pass carries 2^k x the gradient from the final LayerNorm on (fp16 planes: awt_encoder_set_grad_scale_log2); the adapter gradients leave unscaled
  const float gscale = ldexpf(1.0f, e->grad_scale_log2), inv_gscale = ldexpf(1.0f, -e->grad_scale_log2);
  const int M = B * S;
  const int64_t plane = (int64_t)M * d;
  const float lscale = c.lora_alpha / (float)r;
  TrainWs w = carve_train(e, (char*)saved, B);
  const size_t per_layer = lora_grads_per_layer(c);
  int rc;

  // One adapter group (adapters that share an input): du = dy B into w.du, then per adapter dA = lscale du^T x_in and dB = dy^T u at
  // `gp` (advanced).  Forward: u = lscale x_in A^T (saved), y = x_in W^T + b + u B^T.  The group's du stays in w.du for the caller's
  // dX product, which takes it as a second K-segment against (lscale A)^T.
  struct Slot { uint32_t bit; int col; int n_out; };
  auto group_backward = [&](LoraGroup& lg, const bf16_t* dy0, const bf16_t* dy1, int ld_dy, int n_out_total, const Slot* slots, int nslot, bf16_t* const xin[2],
                            int k_in, bf16_t* const u[2], float*& gp) -> int {
    if (!lg.active) return AWT_OK;
    {
      GemmSeg sg = seg_plain(dy0, dy1, ld_dy, lg.bT, 0, n_out_total, M);
      GemmOut o{}; o.hi = w.du[0]; o.lo = w.du[1]; o.ldo = lg.kp; o.n_valid = lg.kp; o.scale = 1.0f;
      int rc2 = launch_gemm(e->ctx, M, 128, &sg, 1, gterms, EPI_BF16, o, s); if (rc2) return rc2;
    }
    for (int i = 0; i < nslot; ++i) {          // adapter i of the group owns rows / columns i r .. i r + r - 1 of A / B, u and du (set_weight)
      if (!(c.lora_targets & slots[i].bit)) continue;
      float* dA = gp;
      float* dB = dA + (size_t)r * k_in;
      int rc2 = launch_outer_reduce(e->ctx, w.du[0], w.du[1], lg.kp, i * r, r, xin[0], xin[1], k_in, 0, k_in, M, lscale * inv_gscale, dA, k_in, 1,
                                    w.partial, w.partial_bytes, accumulate, s);
      if (rc2) return rc2;
      rc2 = launch_outer_reduce(e->ctx, u[0], u[1], lg.kp, i * r, r, dy0, dy1, ld_dy, slots[i].col, slots[i].n_out, M, inv_gscale, dB, 1, r,
                                w.partial, w.partial_bytes, accumulate, s);
      if (rc2) return rc2;
      gp = dB + (size_t)slots[i].n_out * r;
    }
    return AWT_OK;
  };
  // slot order inside a group follows the order the forward packs A rows / B columns: the enabled targets of the group, in order
  const Slot qkv_slots[3] = {{AWT_LORA_Q, 0, d}, {AWT_LORA_K, d, d}, {AWT_LORA_V, 2 * d, d}};
  const Slot out_slot[1] = {{AWT_LORA_OUT, 0, d}}, fc1_slot[1] = {{AWT_LORA_FC1, 0, f}}, fc2_slot[1] = {{AWT_LORA_FC2, 0, d}};

  // final LayerNorm
  float* dx = w.dx_a; float* dx_other = w.dx_b;
  // f16f8 images of d(x_out) (the fc2 backward GEMM's operand) and of d(pre) (the fc1 backward GEMM's) live in the same bytes as their bf16 hi / lo planes
  const Act dxp8 = make_act(w.dxp[0], w.dxp[1], (size_t)M * d, PREC_F16F8), dpre8 = make_act(w.dpre[0], w.dpre[1], (size_t)M * f, PREC_F16F8);
  rc = mlp8 ? launch_layernorm_bwd(e->ctx, d_hidden, w.x_final, e->lnf_g, nullptr, M, d, 1e-5f, dx, nullptr, nullptr, s, gscale, &dxp8)
            : launch_layernorm_bwd(e->ctx, d_hidden, w.x_final, e->lnf_g, nullptr, M, d, 1e-5f, dx, w.dxp[0], w.dxp[1], s, gscale);
  if (rc) return rc;
  for (int li = c.n_layers - 1; li >= 0; --li) {
    Layer& L = e->layers[li];
    const LayerBufs& b = w.layer[li];
    // gradient layout of the layer: q, k, v, out, fc1, fc2 (enabled ones); the groups are visited fc2, fc1, out, qkv
    float* g_qkv = lora_grads + (size_t)li * per_layer;
    size_t n_qkv = 0;
    for (const Slot& sl : qkv_slots) if (c.lora_targets & sl.bit) n_qkv += (size_t)2 * r * d;
    float* g_out = g_qkv + n_qkv;
    float* g_fc1 = g_out + ((c.lora_targets & AWT_LORA_OUT) ? (size_t)2 * r * d : 0);
    float* g_fc2 = g_fc1 + ((c.lora_targets & AWT_LORA_FC1) ? (size_t)r * d + (size_t)f * r : 0);
    // ---- MLP: dpre = ([dx | du2] [W2 | lscale A2]) * gelu'(pre) ; dln = [dpre | du1] [W1 | lscale A1] ; dx_mid = dx + LN2_bwd(dln)
    rc = group_backward(L.l2, w.dxp[0], w.dxp[1], d, d, fc2_slot, 1, b.ff, f, b.u2, g_fc2); if (rc) return rc;
    if (mlp8) {   // no fc1 / fc2 adapters in this mode (encoder_create): one K segment each
      GemmSeg s8 = seg_plain(dxp8, d, L.fc2T8, 0, d, M);
      GemmOut o{}; set_out(o, dpre8); o.ldo = f; o.n_valid = f; o.pre_hi = b.pre[0]; o.pre_lo = b.pre[1]; o.scale = 1.0f;
      rc = launch_gemm(e->ctx, M, f, &s8, 1, PREC_F16F8, EPI_BF16_DGELU, o, s); if (rc) return rc;
      s8 = seg_plain(dpre8, f, L.fc1T8, 0, f, M);
      GemmOut o2{}; o2.f32 = w.dln; o2.ldo = d; o2.n_valid = d;
      rc = launch_gemm(e->ctx, M, d, &s8, 1, PREC_F16F8, EPI_F32, o2, s); if (rc) return rc;
    } else {
    {
      GemmSeg sg[2];
      sg[0] = seg_plain(w.dxp[0], w.dxp[1], d, L.fc2T, 0, d, M);
      if (L.l2.active) sg[1] = seg_plain(w.du[0], w.du[1], L.l2.kp, L.l2.aT, 0, L.l2.kp, M);
      GemmOut o{}; o.hi = w.dpre[0]; o.lo = w.dpre[1]; o.ldo = f; o.n_valid = f; o.pre_hi = b.pre[0]; o.pre_lo = b.pre[1];
      rc = launch_gemm(e->ctx, M, f, sg, L.l2.active ? 2 : 1, gterms, EPI_BF16_DGELU, o, s); if (rc) return rc;
    }
    rc = group_backward(L.l1, w.dpre[0], w.dpre[1], f, f, fc1_slot, 1, b.ln2, d, b.u1, g_fc1); if (rc) return rc;
    {
      GemmSeg sg[2];
      sg[0] = seg_plain(w.dpre[0], w.dpre[1], f, L.fc1T, 0, f, M);
      if (L.l1.active) sg[1] = seg_plain(w.du[0], w.du[1], L.l1.kp, L.l1.aT, 0, L.l1.kp, M);
      GemmOut o{}; o.f32 = w.dln; o.ldo = d; o.n_valid = d;
      rc = launch_gemm(e->ctx, M, d, sg, L.l1.active ? 2 : 1, gterms, EPI_F32, o, s); if (rc) return rc;
    }
    }
    rc = launch_layernorm_bwd(e->ctx, w.dln, b.x_mid, L.ln2_g, dx, M, d, 1e-5f, dx_other, w.dxp[0], w.dxp[1], s); if (rc) return rc;
    std::swap(dx, dx_other);   // dx = d(loss)/d(x_mid)
    // ---- attention: datt = [dx_mid | duo] [Wo | lscale Ao] ; (dq, dk, dv) = attention_bwd
    rc = group_backward(L.lo_, w.dxp[0], w.dxp[1], d, d, out_slot, 1, b.att, d, b.uo, g_out); if (rc) return rc;
    {
      GemmSeg sg[2];
      sg[0] = seg_plain(w.dxp[0], w.dxp[1], d, L.outT, 0, d, M);
      if (L.lo_.active) sg[1] = seg_plain(w.du[0], w.du[1], L.lo_.kp, L.lo_.aT, 0, L.lo_.kp, M);
      GemmOut o{}; o.hi = w.datt[0]; o.lo = w.datt[1]; o.ldo = d; o.n_valid = d; o.scale = 1.0f;
      rc = launch_gemm(e->ctx, M, d, sg, L.lo_.active ? 2 : 1, gterms, EPI_BF16, o, s); if (rc) return rc;
    }
    rc = launch_attention_bwd(e->ctx, b.qkv[0], b.qkv[1], b.qkv[0] + plane, b.qkv[1] ? b.qkv[1] + plane : nullptr, b.qkv[0] + 2 * plane,
                              b.qkv[1] ? b.qkv[1] + 2 * plane : nullptr, b.att[0], b.att[1], w.datt[0], w.datt[1], b.lse, w.delta,
                              w.dqkv[0], w.dqkv[1], B, H, S, 0.125f, terms, gterms, s);
    if (rc) return rc;
    rc = group_backward(L.lq, w.dqkv[0], w.dqkv[1], 3 * d, 3 * d, qkv_slots, 3, b.ln1, d, b.u, g_qkv); if (rc) return rc;
    // ---- gradient exchange: layers [li, group_hi) are final -- average them over the ranks on the side stream while the
    //      lower layers' backward continues on `s`
    if (exchange) {
      const int G = e->comm_groups, Lr = c.n_layers;
      const int grp = (int)((int64_t)li * G / Lr);                       // layer li belongs to group grp (0 = lowest layers)
      const int lo_layer = (int)(((int64_t)grp * Lr + G - 1) / G);       // first layer of that group
      if (li == lo_layer) {
        const int hi_layer = (int)(((int64_t)(grp + 1) * Lr + G - 1) / G);
        rc = comm_reduce_async(e->comm, lora_grads + (size_t)li * per_layer, (size_t)(hi_layer - li) * per_layer, s);
        if (rc) return rc;
      }
    }
    if (li == 0) break;   // nothing below the first adapter needs a gradient
    // ---- dln = [dqkv | du] [Wqkv | lscale A]  ; dx_in = dx_mid + LN1_bwd(dln)
    {
      GemmSeg sg[2];
      sg[0] = seg_plain(w.dqkv[0], w.dqkv[1], 3 * d, L.qkvT, 0, 3 * d, M);
      if (L.lq.active) sg[1] = seg_plain(w.du[0], w.du[1], L.lq.kp, L.lq.aT, 0, L.lq.kp, M);
      GemmOut o{}; o.f32 = w.dln; o.ldo = d; o.n_valid = d;
      rc = launch_gemm(e->ctx, M, d, sg, L.lq.active ? 2 : 1, gterms, EPI_F32, o, s); if (rc) return rc;
    }
    rc = mlp8 ? launch_layernorm_bwd(e->ctx, w.dln, b.x_in, L.ln1_g, dx, M, d, 1e-5f, dx_other, nullptr, nullptr, s, 1.0f, &dxp8)     // feeds the layer below's fc2 backward GEMM
              : launch_layernorm_bwd(e->ctx, w.dln, b.x_in, L.ln1_g, dx, M, d, 1e-5f, dx_other, w.dxp[0], w.dxp[1], s);
    if (rc) return rc;
    std::swap(dx, dx_other);
  }
  if (exchange) { rc = comm_join(e->comm, s); if (rc) return rc; }
  return AWT_OK;
}
