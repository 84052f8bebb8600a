// Log-mel front-end kernels (K1-K4 Whisper, K15 UrbanSound) for gfx950.
//
// What is computed (reference: HF:models/whisper/feature_extraction_whisper.py:105-168, HF:audio_utils.py:809-1017,
// /root/reference/.charles/spectrogram.py:79-87,160-162):
//   reflect-padded framed STFT (periodic Hann folded into the DFT basis) -> |X|^2 -> triangular mel bank ->
//   log10(max(., 1e-10)) [Whisper] or ln(. + eps) [UrbanSound] -> (Whisper) max(., clipmax - 8), (. + 4) / 4.
//
// Design for MI355X:
//  * the DFT is a [frames x n_fft] x [n_fft x 2*bins] contraction on the fp64 matrix pipe
//    (v_mfma_f64_16x16x4_f64): fp64 keeps every bin within ~1e-7 of the reference's float64 NumPy path even for
//    bins 8 decades below the clip maximum, where an fp32 STFT (the reference's own torch path included) is
//    off by up to 4e-5.  At 0.13 GFLOP per 4 s clip this is ~1 % of the encoder's time.
//  * PCM (int16 or fp32) is read once, coalesced, into LDS; frames overlap (hop < n_fft) so the A operand is
//    read straight out of the staged PCM with a one-word-per-hop pad that spreads the 16 frame rows over banks.
//  * the basis lives in L2 (1.3 MB for n_fft 400) in MFMA-B fragment order: one coalesced 512 B load per wave
//    per k-step per part.
//  * only LIVE frames (those that can see a real sample) are computed: for a 4 s clip in the reference's
//    30 s window that is 402 of 3000 frames; the rest is a per-clip constant written by the finalize pass.
//  * per-clip maximum via an order-preserving uint atomicMax; a second, purely HBM-bound pass applies
//    max(., clipmax-8), (.+4)/4 and fills the padding frames.
#include <math.h>
#include <vector>
#include <map>
#include <tuple>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

struct LogmelParams {
  const void* pcm; int pcm_is_i16; int64_t pcm_stride;
  const int32_t* n_valid; int max_valid;
  int L;               // length of the zero-padded signal the reference frames (480000 / 64000 / n_samples)
  int n_fft, hop, n_bins, n_bin_tiles, nbp;   // nbp: row pitch of the power tile in doubles (odd multiple)
  int n_mels, T_out;
  const double* basis;       // [n_bin_tiles][2][n_fft/4][64]
  const int* mel_start; const int* mel_count; const int* mel_off; const double* mel_w;
  int log_mode;              // 0: log10(max(x, 1e-10)) ; 1: ln(x + eps)
  double log_eps;
  float* out; unsigned* clip_max;   // clip_max may be null (generic)
};

__device__ __forceinline__ unsigned float_order_key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_from_key(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

__device__ __forceinline__ int live_frames(int n_valid, int n_fft, int hop, int T_out) {
  // frame t reads padded-signal samples [t*hop - n_fft/2, t*hop + n_fft/2); it sees a real sample iff t*hop - n_fft/2 < n_valid
  if (n_valid <= 0) return 0;
  int n = (n_valid + n_fft / 2 + hop - 1) / hop;
  return n < T_out ? n : T_out;
}

template <int FTILES>
__global__ __launch_bounds__(kThreads) void logmel_stage1_kernel(LogmelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FT = 16 * FTILES;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FT;
  int nv = p.n_valid ? p.n_valid[b] : p.max_valid;
  nv = nv < p.L ? nv : p.L;
  const int n_live = live_frames(nv, p.n_fft, p.hop, p.T_out);
  if (f0 >= n_live) return;

  const int span = (FT - 1) * p.hop + p.n_fft;
  const int span_pad = span + span / p.hop + 1;
  float* pcm_lds = reinterpret_cast<float*>(smem);
  double* pow_lds = reinterpret_cast<double*>(smem + (((size_t)span_pad * 4 + 15) & ~(size_t)15));

  // ---- stage PCM: padded-signal sample j0 + i, reflected at both ends of the L-sample signal, zero beyond n_valid
  const int j0 = f0 * p.hop - p.n_fft / 2;
  const int16_t* pcm16 = reinterpret_cast<const int16_t*>(p.pcm) + (int64_t)b * p.pcm_stride;
  const float* pcm32 = reinterpret_cast<const float*>(p.pcm) + (int64_t)b * p.pcm_stride;
  for (int i = threadIdx.x; i < span; i += kThreads) {
    int j = j0 + i;
    if (j < 0) j = -j;
    if (j >= p.L) j = 2 * (p.L - 1) - j;
    float v = 0.f;
    if (j >= 0 && j < nv) v = p.pcm_is_i16 ? (float)pcm16[j] * (1.0f / 32768.0f) : pcm32[j];
    pcm_lds[i + i / p.hop] = v;
  }
  __syncthreads();

  // ---- DFT on the fp64 matrix pipe: wave w owns bin tiles w, w + 4, ...
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ksteps = p.n_fft / 4;
  const int arow = lane & 15, ak = lane >> 4;
  for (int bt = wave; bt < p.n_bin_tiles; bt += kWaves) {
    f64x4 acc_re[FTILES], acc_im[FTILES];
#pragma unroll
    for (int ft = 0; ft < FTILES; ++ft) { acc_re[ft] = (f64x4){0, 0, 0, 0}; acc_im[ft] = (f64x4){0, 0, 0, 0}; }
    const double* bc = p.basis + ((size_t)(bt * 2 + 0) * ksteps) * 64 + lane;
    const double* bs = p.basis + ((size_t)(bt * 2 + 1) * ksteps) * 64 + lane;
    int nq = 0, next_hop = p.hop;   // nq = (4 ks) / hop, kept incrementally (hop % 4 == 0, so it is wave-uniform)
    for (int ks = 0; ks < ksteps; ++ks) {
      const double vc = bc[(size_t)ks * 64];
      const double vs = bs[(size_t)ks * 64];
      if (4 * ks == next_hop) { ++nq; next_hop += p.hop; }
      const int n = 4 * ks + ak;
#pragma unroll
      for (int ft = 0; ft < FTILES; ++ft) {
        const int fl = 16 * ft + arow;
        const double a = (double)pcm_lds[fl * (p.hop + 1) + n + nq];
        acc_re[ft] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vc, acc_re[ft], 0, 0, 0);
        acc_im[ft] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vs, acc_im[ft], 0, 0, 0);
      }
    }
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    const int bin = bt * 16 + (lane & 15);
#pragma unroll
    for (int ft = 0; ft < FTILES; ++ft) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fl = 16 * ft + (lane >> 4) + 4 * r;
        // the reference stores the spectrum as complex64 before |.|^2 in float64 (HF:audio_utils.py:944,982)
        const double re = (double)(float)acc_re[ft][r];
        const double im = (double)(float)acc_im[ft][r];
        pow_lds[(size_t)fl * p.nbp + bin] = re * re + im * im;
      }
    }
  }
  __syncthreads();

  // ---- mel contraction (sparse triangular filters) + log ; lanes run along frames so stores coalesce along t
  float vmax = -3.0e38f;
  float* outb = p.out + (int64_t)b * p.n_mels * p.T_out;
  for (int idx = threadIdx.x; idx < FT * p.n_mels; idx += kThreads) {
    const int fl = idx % FT, m = idx / FT;
    const int t = f0 + fl;
    const int s0 = p.mel_start[m], cnt = p.mel_count[m];
    const double* w = p.mel_w + p.mel_off[m];
    const double* pw = pow_lds + (size_t)fl * p.nbp + s0;
    double acc = 0.0;
    for (int i = 0; i < cnt; ++i) acc += w[i] * pw[i];
    float v;
    if (p.log_mode == 0) v = (float)log10(acc > 1e-10 ? acc : 1e-10);
    else v = (float)log(acc + p.log_eps);
    if (t < n_live) {
      outb[(int64_t)m * p.T_out + t] = v;
      vmax = fmaxf(vmax, v);
    }
  }
  if (p.clip_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0 && vmax > -3.0e38f) atomicMax(p.clip_max + b, float_order_key(vmax));
  }
}

__global__ void logmel_init_max_kernel(unsigned* clip_max, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) clip_max[i] = float_order_key(-10.0f);  // log10(mel_floor): the value of every all-zero frame
}

// Whisper normalisation + padding fill: out = (max(raw, clipmax - 8) + 4) / 4 in fp32, exactly the reference's op order
// (HF:feature_extraction_whisper.py:129-130,159-165).  Pure streaming: 4 frames (16 B) per lane.
__global__ __launch_bounds__(256) void logmel_finalize_kernel(float* out, const unsigned* clip_max, const int32_t* n_valid,
                                                              int max_valid, int L, int n_fft, int hop, int n_mels, int T_out) {
  const int b = blockIdx.y;
  int nv = n_valid ? n_valid[b] : max_valid;
  nv = nv < L ? nv : L;
  const int n_live = live_frames(nv, n_fft, hop, T_out);
  const float thr = float_from_key(clip_max[b]) - 8.0f;
  const int quads = T_out / 4;
  float4* o4 = reinterpret_cast<float4*>(out + (int64_t)b * n_mels * T_out);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_mels * quads; i += gridDim.x * blockDim.x) {
    const int t = (i % quads) * 4;
    float4 v = make_float4(-10.f, -10.f, -10.f, -10.f);
    if (t + 3 < n_live) v = o4[i];
    else if (t < n_live) {
      const float* src = reinterpret_cast<const float*>(o4 + i);
      v.x = src[0];
      if (t + 1 < n_live) v.y = src[1];
      if (t + 2 < n_live) v.z = src[2];
    }
    v.x = (fmaxf(v.x, thr) + 4.0f) * 0.25f; v.y = (fmaxf(v.y, thr) + 4.0f) * 0.25f;
    v.z = (fmaxf(v.z, thr) + 4.0f) * 0.25f; v.w = (fmaxf(v.w, thr) + 4.0f) * 0.25f;
    o4[i] = v;
  }
}

// ------------------------------------------------------------------------------------------ host-side tables
double hz_to_mel(double f, bool slaney) {
  if (!slaney) return 2595.0 * log10(1.0 + f / 700.0);                  // HF:audio_utils.py:466-467
  if (f >= 1000.0) return 15.0 + log(f / 1000.0) * (27.0 / log(6.4));   // :471-479
  return 3.0 * f / 200.0;
}
double mel_to_hz(double m, bool slaney) {
  if (!slaney) return 700.0 * (pow(10.0, m / 2595.0) - 1.0);            // :502-503
  if (m >= 15.0) return 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0));   // :507-515
  return 200.0 * m / 3.0;
}
std::vector<double> linspace(double a, double b, int n) {
  std::vector<double> v(n);
  const double step = (b - a) / (n - 1);
  for (int i = 0; i < n; ++i) v[i] = a + i * step;
  v[n - 1] = b;
  return v;
}

struct MelTable { int* start; int* count; int* off; double* w; };
struct BasisTable { double* basis; int n_bin_tiles; };
// polyphase resampling filter: taps[phase][K] fp32 (torchaudio's kernel), and per phase the range of non-zero taps
struct ResampleTable { float* taps; int* first; int* count; int orig, out, width, K; };

}  // namespace

struct awt_ctx::Table {
  std::map<int, BasisTable> basis;                                                   // key n_fft
  std::map<std::tuple<int, int, int, int, int, int>, MelTable> mel;                  // n_bins, n_mels, fmin*8, fmax*8, sr, slaney
  std::map<std::pair<int, int>, ResampleTable> resample;                             // (sr_in, sr_out) / gcd
  std::vector<void*> allocs;
};

void awt_free_tables(awt_ctx* c) {
  if (!c->tables) return;
  for (void* p : c->tables->allocs) (void)hipFree(p);
  delete c->tables;
  c->tables = nullptr;
}

namespace {

int get_basis(awt_ctx* c, int n_fft, hipStream_t s, BasisTable* out) {
  if (!c->tables) c->tables = new awt_ctx::Table();
  auto it = c->tables->basis.find(n_fft);
  if (it != c->tables->basis.end()) { *out = it->second; return AWT_OK; }
  const int n_bins = n_fft / 2 + 1;
  const int nbt = (n_bins + 15) / 16;
  const int ksteps = n_fft / 4;
  std::vector<double> h((size_t)nbt * 2 * ksteps * 64);
  // periodic Hann folded in: np.hanning(n_fft + 1)[:-1] = 0.5 - 0.5 cos(2 pi n / n_fft)   (HF:audio_utils.py:778-792)
  for (int bt = 0; bt < nbt; ++bt)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = 4 * ks + (lane >> 4);
        const int bin = 16 * bt + (lane & 15);
        double vc = 0.0, vs = 0.0;
        if (bin < n_bins) {
          const double w = 0.5 - 0.5 * cos(2.0 * M_PI * n / n_fft);
          const long long ph = ((long long)bin * n) % n_fft;  // exact argument reduction
          const double ang = 2.0 * M_PI * (double)ph / n_fft;
          vc = w * cos(ang);
          vs = -w * sin(ang);
        }
        h[((size_t)(bt * 2 + 0) * ksteps + ks) * 64 + lane] = vc;
        h[((size_t)(bt * 2 + 1) * ksteps + ks) * 64 + lane] = vs;
      }
  BasisTable t{};
  t.n_bin_tiles = nbt;
  AWT_HIP_CHECK(hipMalloc((void**)&t.basis, h.size() * sizeof(double)));
  c->tables->allocs.push_back(t.basis);
  AWT_HIP_CHECK(hipMemcpy(t.basis, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  c->tables->basis[n_fft] = t;
  *out = t;
  return AWT_OK;
}

int get_mel(awt_ctx* c, int n_bins, int n_mels, double fmin, double fmax, int sr, bool slaney, MelTable* out) {
  if (!c->tables) c->tables = new awt_ctx::Table();
  auto key = std::make_tuple(n_bins, n_mels, (int)lrint(fmin * 8), (int)lrint(fmax * 8), sr, (int)slaney);
  auto it = c->tables->mel.find(key);
  if (it != c->tables->mel.end()) { *out = it->second; return AWT_OK; }
  // mel_filter_bank (HF:audio_utils.py:696-729): filters triangular in Hz, centres equally spaced in mel
  std::vector<double> mel_f = linspace(hz_to_mel(fmin, slaney), hz_to_mel(fmax, slaney), n_mels + 2);
  std::vector<double> ff(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel_to_hz(mel_f[i], slaney);
  std::vector<double> fft_f = linspace(0.0, (double)(sr / 2), n_bins);
  std::vector<int> start(n_mels), count(n_mels), off(n_mels);
  std::vector<double> w;
  for (int m = 0; m < n_mels; ++m) {
    int first = -1, last = -1;
    std::vector<double> col(n_bins);
    const double enorm = slaney ? 2.0 / (ff[m + 2] - ff[m]) : 1.0;   // Slaney norm goes with the Slaney scale here
    for (int k = 0; k < n_bins; ++k) {
      const double down = -(ff[m] - fft_f[k]) / (ff[m + 1] - ff[m]);
      const double up = (ff[m + 2] - fft_f[k]) / (ff[m + 2] - ff[m + 1]);
      double v = down < up ? down : up;
      v = v > 0.0 ? v : 0.0;
      col[k] = v * enorm;
      if (v > 0.0) { if (first < 0) first = k; last = k; }
    }
    if (first < 0) { first = 0; last = -1; }
    start[m] = first; count[m] = last - first + 1; off[m] = (int)w.size();
    for (int k = first; k <= last; ++k) w.push_back(col[k]);
  }
  if (w.empty()) w.push_back(0.0);
  MelTable t{};
  AWT_HIP_CHECK(hipMalloc((void**)&t.start, n_mels * sizeof(int)));  c->tables->allocs.push_back(t.start);
  AWT_HIP_CHECK(hipMalloc((void**)&t.count, n_mels * sizeof(int)));  c->tables->allocs.push_back(t.count);
  AWT_HIP_CHECK(hipMalloc((void**)&t.off, n_mels * sizeof(int)));    c->tables->allocs.push_back(t.off);
  AWT_HIP_CHECK(hipMalloc((void**)&t.w, w.size() * sizeof(double))); c->tables->allocs.push_back(t.w);
  AWT_HIP_CHECK(hipMemcpy(t.start, start.data(), n_mels * sizeof(int), hipMemcpyHostToDevice));
  AWT_HIP_CHECK(hipMemcpy(t.count, count.data(), n_mels * sizeof(int), hipMemcpyHostToDevice));
  AWT_HIP_CHECK(hipMemcpy(t.off, off.data(), n_mels * sizeof(int), hipMemcpyHostToDevice));
  AWT_HIP_CHECK(hipMemcpy(t.w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
  c->tables->mel[key] = t;
  *out = t;
  return AWT_OK;
}

template <int FTILES>
int launch_stage1(const LogmelParams& p, int B, int max_live, hipStream_t s) {
  constexpr int FT = 16 * FTILES;
  const int span = (FT - 1) * p.hop + p.n_fft;
  const int span_pad = span + span / p.hop + 1;
  const size_t lds = (((size_t)span_pad * 4 + 15) & ~(size_t)15) + (size_t)FT * p.nbp * 8;
  AWT_REQUIRE(lds <= 160 * 1024, AWT_ERR_INVALID, "logmel: frame tile does not fit LDS");
  AWT_ONCE_PER_DEVICE(AWT_HIP_CHECK(hipFuncSetAttribute((const void*)logmel_stage1_kernel<FTILES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
  dim3 grid((max_live + FT - 1) / FT, B);
  if (grid.x == 0) return AWT_OK;
  hipLaunchKernelGGL(logmel_stage1_kernel<FTILES>, grid, dim3(kThreads), lds, s, p);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int host_live_frames(int n_valid, int n_fft, int hop, int T_out) {
  if (n_valid <= 0) return 0;
  int n = (n_valid + n_fft / 2 + hop - 1) / hop;
  return n < T_out ? n : T_out;
}

}  // namespace

size_t awt_logmel_workspace_bytes(int B) { return (((size_t)(B > 0 ? B : 1) * sizeof(unsigned)) + 255) & ~(size_t)255; }

// Builds (once per context) the device tables a front-end configuration needs: the DFT basis of n_fft and the mel filter bank.
// Called from awt_ctx_create for the two Whisper front-ends and from awt_logmel_prepare for any other configuration, so that the
// compute entry points neither allocate nor synchronise (a configuration that was never prepared is still built at first use).
int logmel_prepare_impl(awt_ctx* c, int n_fft, int n_mels, double f_min, double f_max, int sample_rate, int slaney) {
  AWT_REQUIRE(c, AWT_ERR_INVALID, "logmel_prepare: null context");
  AWT_REQUIRE(n_fft == 400 || n_fft == 512 || n_fft == 1024, AWT_ERR_INVALID, "logmel_prepare: n_fft must be 400, 512 or 1024");
  AWT_REQUIRE(n_mels > 0 && n_mels <= 128 && sample_rate > 0, AWT_ERR_INVALID, "logmel_prepare: n_mels must be in 1..128");
  BasisTable bt; MelTable mt;
  int rc = get_basis(c, n_fft, nullptr, &bt); if (rc) return rc;
  return get_mel(c, n_fft / 2 + 1, n_mels, f_min, f_max, sample_rate, slaney != 0, &mt);
}
int resample_prepare_impl(awt_ctx* c, int sr_in, int sr_out);

int logmel_whisper_impl(awt_ctx* c, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                        int max_valid, int B, int n_frames_out, int n_mels, float* out, void* workspace, size_t ws_bytes, hipStream_t s) {
  AWT_REQUIRE(c && pcm && out && workspace, AWT_ERR_INVALID, "logmel_whisper: null argument");
  AWT_REQUIRE(B > 0 && n_frames_out > 0 && n_frames_out % 4 == 0, AWT_ERR_INVALID, "logmel_whisper: B > 0 and n_frames_out % 4 == 0 required");
  AWT_REQUIRE(max_valid >= 0 && (pcm_stride >= max_valid || B == 1), AWT_ERR_INVALID, "logmel_whisper: pcm_stride < max_valid");
  AWT_REQUIRE(ws_bytes >= awt_logmel_workspace_bytes(B), AWT_ERR_WORKSPACE, "logmel_whisper: workspace too small");
  AWT_REQUIRE(((uintptr_t)out & 15) == 0, AWT_ERR_INVALID, "logmel_whisper: out must be 16-byte aligned");
  AWT_REQUIRE(n_mels == 80 || n_mels == 128, AWT_ERR_INVALID, "logmel_whisper: n_mels must be 80 (tiny .. large-v2) or 128 (large-v3)");
  const int n_fft = 400, hop = 160;
  BasisTable bt; MelTable mt;
  int rc = get_basis(c, n_fft, s, &bt); if (rc) return rc;
  rc = get_mel(c, n_fft / 2 + 1, n_mels, 0.0, 8000.0, 16000, true, &mt); if (rc) return rc;
  LogmelParams p{};
  p.pcm = pcm; p.pcm_is_i16 = pcm_is_i16; p.pcm_stride = pcm_stride; p.n_valid = n_valid; p.max_valid = max_valid;
  p.L = n_frames_out * hop; p.n_fft = n_fft; p.hop = hop; p.n_bins = n_fft / 2 + 1; p.n_bin_tiles = bt.n_bin_tiles;
  p.nbp = bt.n_bin_tiles * 16 + 1; p.n_mels = n_mels; p.T_out = n_frames_out; p.basis = bt.basis;
  p.mel_start = mt.start; p.mel_count = mt.count; p.mel_off = mt.off; p.mel_w = mt.w;
  p.log_mode = 0; p.log_eps = 0.0; p.out = out; p.clip_max = reinterpret_cast<unsigned*>(workspace);
  const int mv = max_valid < p.L ? max_valid : p.L;
  const int max_live = host_live_frames(mv, n_fft, hop, n_frames_out);
  ProfScope prof(c, AWT_PROF_LOGMEL, s, 2.0 * 2.0 * n_fft * p.n_bins * (double)max_live * B);
  hipLaunchKernelGGL(logmel_init_max_kernel, dim3((B + 255) / 256), dim3(256), 0, s, p.clip_max, B);
  AWT_HIP_CHECK(hipGetLastError());
  rc = launch_stage1<2>(p, B, max_live, s); if (rc) return rc;
  const int work = n_mels * (n_frames_out / 4);
  int gx = (work + 255) / 256; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(logmel_finalize_kernel, dim3(gx, B), dim3(256), 0, s, out, p.clip_max, n_valid, max_valid, p.L, n_fft, hop,
                     n_mels, n_frames_out);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}

int logmel_generic_impl(awt_ctx* c, const float* pcm, int64_t pcm_stride, int B, int n_samples, int sample_rate, int n_fft,
                        int hop, int n_mels, float f_min, float f_max, float log_eps, float* out, hipStream_t s) {
  AWT_REQUIRE(c && pcm && out, AWT_ERR_INVALID, "logmel_generic: null argument");
  AWT_REQUIRE(B > 0 && n_samples > n_fft / 2, AWT_ERR_INVALID, "logmel_generic: need B > 0 and n_samples > n_fft / 2 (reflect padding)");
  AWT_REQUIRE(n_fft == 400 || n_fft == 512 || n_fft == 1024, AWT_ERR_INVALID, "logmel_generic: n_fft must be 400, 512 or 1024");
  AWT_REQUIRE(hop > 0 && hop % 4 == 0 && hop <= n_fft, AWT_ERR_INVALID, "logmel_generic: hop must be a multiple of 4 and <= n_fft");
  AWT_REQUIRE(n_mels > 0 && n_mels <= 128, AWT_ERR_INVALID, "logmel_generic: n_mels must be in 1..128");
  AWT_REQUIRE(f_min >= 0 && f_max > f_min && f_max <= sample_rate / 2, AWT_ERR_INVALID, "logmel_generic: need 0 <= f_min < f_max <= sr/2");
  BasisTable bt; MelTable mt;
  int rc = get_basis(c, n_fft, s, &bt); if (rc) return rc;
  rc = get_mel(c, n_fft / 2 + 1, n_mels, f_min, f_max, sample_rate, false, &mt); if (rc) return rc;
  LogmelParams p{};
  p.pcm = pcm; p.pcm_is_i16 = 0; p.pcm_stride = pcm_stride; p.n_valid = nullptr; p.max_valid = n_samples;
  p.L = n_samples; p.n_fft = n_fft; p.hop = hop; p.n_bins = n_fft / 2 + 1; p.n_bin_tiles = bt.n_bin_tiles;
  p.nbp = bt.n_bin_tiles * 16 + 1; p.n_mels = n_mels; p.T_out = 1 + n_samples / hop; p.basis = bt.basis;
  p.mel_start = mt.start; p.mel_count = mt.count; p.mel_off = mt.off; p.mel_w = mt.w;
  p.log_mode = 1; p.log_eps = (double)log_eps; p.out = out; p.clip_max = nullptr;
  ProfScope prof(c, AWT_PROF_LOGMEL, s, 2.0 * 2.0 * n_fft * p.n_bins * (double)p.T_out * B);
  // every frame of the (un-padded) generic front-end is live: n_valid == L makes live_frames() == T_out
  return launch_stage1<1>(p, B, p.T_out, s);
}

// ------------------------------------------------------------------------------------------ mono mix + resampling (K16)
// Stands behind /root/reference/.charles/spectrogram.py:146-157: channel mean, torchaudio.transforms.Resample(sr -> 16000)
// with its defaults (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), zero-pad / truncate.  torchaudio evaluates
// the filter as a stride-`orig` conv1d over all K = 2 width + orig taps of each of the `out` phases; only the taps inside
// the Hann window (|t| < 6) are non-zero, so each output sample here sums 13 - 73 taps instead of 15 - 475.
// HBM-bound and tiny (0.13 - 0.77 MB read, 0.26 MB written per 4 s clip): one thread per output sample.
namespace {

template <bool I16>
__global__ __launch_bounds__(256) void resample_mono_kernel(const void* pcm, int channels, int64_t cstride, int64_t sstride,
                                                            int n_in, ResampleTable t, int n_res, float* out, int n_out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  if (n >= n_res) { out[n] = 0.f; return; }
  auto sample = [&](int m) {            // channel mean of input sample m, as torch.mean(waveform, dim=0): sum, then divide
    float sum = 0.f;
    for (int c = 0; c < channels; ++c) {
      const int64_t off = c * cstride + m * sstride;
      sum += I16 ? (float)reinterpret_cast<const int16_t*>(pcm)[off] * (1.0f / 32768.0f) : reinterpret_cast<const float*>(pcm)[off];
    }
    return channels > 1 ? sum / (float)channels : sum;
  };
  if (t.taps == nullptr) { out[n] = sample(n); return; }   // equal rates: the reference skips the resampler
  const int j = n / t.out, ph = n - j * t.out;
  const int base = j * t.orig - t.width;                   // tap k multiplies input sample base + k (zero outside the clip)
  const float* taps = t.taps + (int64_t)ph * t.K;
  const int k0 = t.first[ph], k1 = k0 + t.count[ph];
  float acc = 0.f;
  for (int k = max(k0, -base); k < k1; ++k) {
    const int m = base + k;
    if (m >= n_in) break;
    acc = fmaf(taps[k], sample(m), acc);
  }
  out[n] = acc;
}

int get_resample(awt_ctx* c, int sr_in, int sr_out, ResampleTable* out) {
  if (!c->tables) c->tables = new awt_ctx::Table();
  int a = sr_in, b = sr_out;
  while (b) { const int r = a % b; a = b; b = r; }
  const int orig = sr_in / a, outp = sr_out / a;
  auto it = c->tables->resample.find({orig, outp});
  if (it != c->tables->resample.end()) { *out = it->second; return AWT_OK; }
  // torchaudio.functional._get_sinc_resample_kernel, dtype=None: float64 arithmetic (the phase term i / new is an int64
  // tensor divided by an int, i.e. rounded to float32 first), result rounded to float32
  const double lpw = 6.0, rolloff = 0.99;
  const double basef = (double)(orig < outp ? orig : outp) * rolloff;
  const int width = (int)ceil(lpw * orig / basef);
  const int K = 2 * width + orig;
  std::vector<float> taps((size_t)outp * K);
  std::vector<int> first(outp), count(outp);
  for (int i = 0; i < outp; ++i) {
    const double phase = (double)((float)(-i) / (float)outp);
    int lo = K, hi = -1;
    for (int k = 0; k < K; ++k) {
      double t = (phase + (double)(k - width) / (double)orig) * basef;
      t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
      const double cw = cos(t * M_PI / lpw / 2.0);
      const double window = cw * cw;
      const double tp = t * M_PI;
      const double sinc = tp == 0.0 ? 1.0 : sin(tp) / tp;
      const float v = (float)(sinc * window * (basef / (double)orig));
      taps[(size_t)i * K + k] = v;
      if (v != 0.f) { if (k < lo) lo = k; hi = k; }
    }
    first[i] = hi < 0 ? 0 : lo; count[i] = hi < 0 ? 0 : hi - lo + 1;
  }
  ResampleTable t{}; t.orig = orig; t.out = outp; t.width = width; t.K = K;
  AWT_HIP_CHECK(hipMalloc((void**)&t.taps, taps.size() * sizeof(float))); c->tables->allocs.push_back(t.taps);
  AWT_HIP_CHECK(hipMalloc((void**)&t.first, outp * sizeof(int)));         c->tables->allocs.push_back(t.first);
  AWT_HIP_CHECK(hipMalloc((void**)&t.count, outp * sizeof(int)));         c->tables->allocs.push_back(t.count);
  AWT_HIP_CHECK(hipMemcpy(t.taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice));
  AWT_HIP_CHECK(hipMemcpy(t.first, first.data(), outp * sizeof(int), hipMemcpyHostToDevice));
  AWT_HIP_CHECK(hipMemcpy(t.count, count.data(), outp * sizeof(int), hipMemcpyHostToDevice));
  c->tables->resample[{orig, outp}] = t;
  *out = t;
  return AWT_OK;
}

}  // namespace

int64_t resampled_length(int n_in, int sr_in, int sr_out) {   // ceil(new * n / orig), torchaudio's target_length
  int a = sr_in, b = sr_out;
  while (b) { const int r = a % b; a = b; b = r; }
  const int64_t orig = sr_in / a, outp = sr_out / a;
  return ((int64_t)outp * n_in + orig - 1) / orig;
}

int resample_prepare_impl(awt_ctx* c, int sr_in, int sr_out) {
  AWT_REQUIRE(c, AWT_ERR_INVALID, "resample_prepare: null context");
  AWT_REQUIRE(sr_in >= 1000 && sr_in <= 768000 && sr_out >= 1000 && sr_out <= 768000, AWT_ERR_INVALID, "resample_prepare: sample rates must be in 1 kHz .. 768 kHz");
  if (sr_in == sr_out) return AWT_OK;
  ResampleTable t{};
  return get_resample(c, sr_in, sr_out, &t);
}

int prepare_waveform_impl(awt_ctx* c, const void* pcm, int pcm_is_i16, int channels, int64_t channel_stride, int64_t sample_stride,
                          int n_in, int sr_in, int sr_out, float* out, int n_out, hipStream_t s) {
  AWT_REQUIRE(c && pcm && out, AWT_ERR_INVALID, "prepare_waveform: null argument");
  AWT_REQUIRE(channels >= 1 && channels <= 8 && n_in > 0 && n_out > 0, AWT_ERR_INVALID, "prepare_waveform: need 1..8 channels, n_in > 0, n_out > 0");
  AWT_REQUIRE(sr_in >= 1000 && sr_in <= 768000 && sr_out >= 1000 && sr_out <= 768000, AWT_ERR_INVALID, "prepare_waveform: sample rates must be in 1 kHz .. 768 kHz");
  ResampleTable t{};
  int n_res = n_in;
  if (sr_in != sr_out) {
    int rc = get_resample(c, sr_in, sr_out, &t); if (rc) return rc;
    AWT_REQUIRE((int64_t)t.out * t.K <= (1 << 22), AWT_ERR_INVALID, "prepare_waveform: rate pair needs too large a filter table");
    const int64_t r = resampled_length(n_in, sr_in, sr_out);
    n_res = r > n_out ? n_out : (int)r;
  } else if (n_res > n_out) {
    n_res = n_out;
  }
  ProfScope prof(c, AWT_PROF_LOGMEL, s, 0.0);
  const dim3 grid((n_out + 255) / 256);
  if (pcm_is_i16) hipLaunchKernelGGL(resample_mono_kernel<true>, grid, dim3(256), 0, s, pcm, channels, channel_stride, sample_stride, n_in, t, n_res, out, n_out);
  else hipLaunchKernelGGL(resample_mono_kernel<false>, grid, dim3(256), 0, s, pcm, channels, channel_stride, sample_stride, n_in, t, n_res, out, n_out);
  AWT_HIP_CHECK(hipGetLastError());
  return AWT_OK;
}
