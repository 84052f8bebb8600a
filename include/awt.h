/* awt.h -- C-ABI of libawt.so: the MI355X-native (gfx950) log-mel front-end + Whisper-style audio encoder.
 *
 * The reference (AdamBeedell/MLX8-WS-Audio-Transformer) has no FFI / plugin interface for this path: it is
 * reached through ordinary Python call sites into `transformers`.  Each entry point below
 * names the reference call it stands behind; the Python modules under mlx8-ws-audio-transformer_amd/ wrap them so that those
 * call sites keep their names, argument meaning and error behaviour (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions
 *  - every function returns AWT_OK (0) or a negative awt_status; nothing throws across the ABI;
 *    awt_last_error() returns a thread-local message owned by the library;
 *  - all data pointers are CALLER-OWNED DEVICE pointers (e.g. torch tensors' data_ptr()); the library never
 *    frees caller memory and never allocates inside a compute call (tables and converted weights are
 *    allocated in *_create / *_set_weight / awt_*_prepare; only a front-end configuration that was never prepared is
 *    built at its first use);
 *  - all work is enqueued on the caller's hipStream_t (passed as void*; NULL = default stream);
 *    no hidden synchronisation except where a function says so (awt_prof_collect);
 *  - a handle is not thread-safe; distinct handles are independent; one awt_ctx per (process, device).
 */
#ifndef AWT_H_
#define AWT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum awt_status {
  AWT_OK = 0,
  AWT_ERR_INVALID = -1,      /* bad argument (shape, alignment, NULL, unsupported config)            */
  AWT_ERR_HIP = -2,          /* a HIP runtime call failed; message carries hipGetErrorString          */
  AWT_ERR_WORKSPACE = -3,    /* caller's workspace is smaller than *_workspace_bytes()                */
  AWT_ERR_STATE = -4,        /* weights missing / handle misuse                                       */
  AWT_ERR_VALUE = -5         /* value error the reference raises too (e.g. mel length != 2*S)         */
} awt_status;

typedef struct awt_ctx awt_ctx;
typedef struct awt_encoder awt_encoder;

const char* awt_last_error(void);
const char* awt_version(void);

int awt_ctx_create(int device, awt_ctx** out);
void awt_ctx_destroy(awt_ctx* c);

/* ------------------------------------------------------------------------------------------------------
 * Whisper log-mel (K1-K4).  Stands behind `processor(audio, sampling_rate=16000)` /
 * `WhisperFeatureExtractor.__call__`:
 *   /root/reference/AB/fineTune.py:88, AB/wavToWhisper.py:55, AB/fineTuneMidiTester.py:33,
 *   .charles/music2midi/model.py:100-104 -> HF:models/whisper/feature_extraction_whisper.py:105-168,193-346.
 * pcm        [B] clips, `pcm_stride` elements apart; int16 (decoded as x/32768, the reference loaders'
 *            convention) when pcm_is_i16 != 0, else float32.
 * n_valid    device int32[B] = samples of each clip that are real (rest is zero padding), or NULL = max_valid
 *            for every clip.  Samples at index >= n_valid[b] are never read.
 * max_valid  host-side upper bound of n_valid (sizes the grid); clips longer than 160*n_frames_out samples
 *            are truncated like the reference does (feature_extraction_whisper.py:300-305).
 * n_frames_out  3000 = the reference's behaviour (clip zero-padded to 30 s); 400 = trimmed 4 s mode.
 * out        float32 [B, 80, n_frames_out].
 * workspace  >= awt_logmel_workspace_bytes(B) bytes, 16-byte aligned.
 */
size_t awt_logmel_workspace_bytes(int B);
int awt_logmel_whisper(awt_ctx* c, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                       int max_valid, int B, int n_frames_out, float* out, void* workspace, size_t ws_bytes,
                       void* stream);
/* Same with the number of mel bins explicit: 80 (`WhisperFeatureExtractor()` defaults, tiny .. large-v2) or 128
 * (`feature_size=128`, large-v3); out is [B, n_mels, n_frames_out]. */
int awt_logmel_whisper_mels(awt_ctx* c, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                            int max_valid, int B, int n_frames_out, int n_mels, float* out, void* workspace, size_t ws_bytes,
                            void* stream);

/* UrbanSound log-mel (K15).  Stands behind `torch.log(mel_spectrogram(waveform) + 1e-6)`:
 *   /root/reference/.charles/spectrogram.py:79-87,160-162 (torchaudio MelSpectrogram: periodic Hann(n_fft),
 *   center/reflect, power 2, HTK mel, no norm).
 * pcm float32 [B] clips of n_samples (already mono / padded / trimmed: spectrogram.py:145-157),
 * out float32 [B, n_mels, 1 + n_samples / hop].  Supported: n_fft in {400, 512, 1024}, n_mels <= 128. */
int awt_logmel_generic(awt_ctx* c, const float* pcm, int64_t pcm_stride, int B, int n_samples, int sample_rate,
                       int n_fft, int hop, int n_mels, float f_min, float f_max, float log_eps, float* out,
                       void* stream);

/* Mono mix + sample-rate conversion + pad / trim (K16).  Stands behind /root/reference/.charles/spectrogram.py:146-157:
 *   `torch.mean(waveform, dim=0, keepdim=True)` when there is more than one channel, then
 *   `torchaudio.transforms.Resample(orig_freq=sr, new_freq=SAMPLE_RATE)(waveform)` when sr differs (transform defaults:
 *   sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), then zero-pad / truncate to `n_out` samples.
 * pcm: `channels` channels of `n_in` samples, int16 (scaled by 1/32768 as torchaudio.load does) or float32; element
 *   (c, i) is at pcm[c * channel_stride + i * sample_stride] -- planar [C][n] is (n, 1), interleaved WAV data is (1, C).
 * out: float32 [n_out].  The resampled clip has awt_resampled_length() = ceil(n_in * sr_out / sr_in) samples; output
 *   samples beyond it are zero.  sr_in == sr_out skips the filter, as the reference does. */
/* Device tables (DFT basis + mel bank of a front-end configuration; polyphase taps of a rate pair) are built once per context.
 * awt_ctx_create builds the two Whisper front-ends' tables; call these for any other configuration BEFORE the first compute call that
 * uses it, so that compute calls neither allocate nor synchronise (an unprepared configuration is built at first use, with a
 * hipMalloc and a blocking copy inside that one call).  slaney != 0: Slaney scale + area normalisation (Whisper); 0: HTK, no norm. */
int awt_logmel_prepare(awt_ctx* c, int n_fft, int n_mels, float f_min, float f_max, int sample_rate, int slaney);
int awt_resample_prepare(awt_ctx* c, int sr_in, int sr_out);
int64_t awt_resampled_length(int n_in, int sr_in, int sr_out);
int awt_prepare_waveform(awt_ctx* c, const void* pcm, int pcm_is_i16, int channels, int64_t channel_stride,
                         int64_t sample_stride, int n_in, int sr_in, int sr_out, float* out, int n_out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Whisper-style audio encoder (K5-K13).  Stands behind `encoder(input_features).last_hidden_state`
 * (`WhisperModel.get_encoder()`, .charles/music2midi/model.py:33,109-110; implicitly AB/fineTune.py:199)
 * -> HF:models/whisper/modeling_whisper.py:592-646 (+ :379-413, :284-356, :215-238).
 */
typedef struct awt_encoder_cfg {
  int32_t d_model;          /* 384 tiny, 512 base, 768 small, 1024 medium, 1280 large; multiple of 128, <= 1280 */
  int32_t n_layers;
  int32_t n_heads;          /* head_dim = d_model / n_heads must be 64                                     */
  int32_t ffn_dim;          /* multiple of 128                                                             */
  int32_t n_mels;           /* 80 (tiny .. large-v2) or 128 (large-v3); multiple of 8, <= 128              */
  int32_t n_ctx;            /* S = max_source_positions: 1500 (reference) or 200 (trimmed); mel T = 2*S    */
  int32_t mfma_terms;       /* operand precision of the forward pass:
                               1 bf16: one bf16 MFMA per fragment pair (fast; ~4e-3 rel-L2 vs fp32, misses the 1e-3 bound);
                               2 fp16: one fp16 MFMA per fragment pair (11 significant bits per operand: max-abs ~2.5e-3 on
                                 Whisper-small, misses the bound as well; a measurement mode, inference only);
                               3 bf16x3: split-bf16 hi + lo planes, three MFMAs (2^-17 per operand; the training format);
                               4 fp16x3: split-fp16 planes, three MFMAs (2^-23 per operand; operands within fp16's range);
                               5 f16f8: fp16 plane + two e4m3 planes, the two cross terms on the block-scaled fp8 MFMA: two
                                 MFMA-equivalents per fragment pair (2^-16 per operand); inference only.  Its attention keeps
                                 the cross terms of q k^T and runs P V as one fp16 product           */
  int32_t lora_rank;        /* 0 = no adapters; else 1..64                                                 */
  float lora_alpha;         /* adapter scale = lora_alpha / lora_rank                                      */
  uint32_t lora_targets;    /* bit mask of AWT_LORA_*                                                      */
  int32_t chunk_clips;      /* clips processed per kernel wave (0 = library default)                       */
  int32_t training;         /* != 0: keep transposed copies of the frozen weights so awt_encoder_backward can run
                               (mfma_terms 1 or 3 in this mode)                                             */
  int32_t backward_terms;   /* products of the backward pass' gradient contractions: 0 = mfma_terms (default: gradients
                               to 3e-5 of fp32 autograd), 1 with mfma_terms = 3 = one bf16 product (the usual
                               mixed-precision trade: ~0.5 % gradient error, 1.4x faster step); the attention scores are
                               recomputed in split-bf16 either way.  5 with mfma_terms = 3: the MLP of the training step -- fc1 / fc2
                               forward and their two backward GEMMs (d pre = (dx W2) gelu'(pre), d ln2 = d pre W1) -- runs in the
                               f16f8 operand format (2 instead of 3 MFMA-equivalents, 2^-16 per operand like split-bf16), everything
                               else in split-bf16; fp16 planes want gradients of order one: see awt_encoder_set_grad_scale_log2.
                               No fc1 / fc2 adapters.                                                                       */
} awt_encoder_cfg;

enum { AWT_LORA_Q = 1, AWT_LORA_K = 2, AWT_LORA_V = 4, AWT_LORA_OUT = 8, AWT_LORA_FC1 = 16, AWT_LORA_FC2 = 32 };

int awt_encoder_create(awt_ctx* c, const awt_encoder_cfg* cfg, awt_encoder** out);
void awt_encoder_destroy(awt_encoder* e);

/* Upload one parameter by its HF state-dict key (relative to the encoder module): `conv1.weight`,
 * `layers.{i}.self_attn.{q,k,v,out}_proj.{weight,bias}`, `layers.{i}.{self_attn_layer_norm,final_layer_norm}.*`,
 * `layers.{i}.fc{1,2}.*`, `layer_norm.*`, `embed_positions.weight`, plus build-defined `<module>.lora_A` [r, in] and
 * `<module>.lora_B` [out, r].  `data` is a float32 device pointer of `rank`-d `shape`; the library converts it to
 * its internal operand planes on `stream` and keeps its own copy.  In the f16f8 inference format the upload of a
 * projection matrix (q / k / v / out_proj / fc1 / fc2) also finds out whether every element is exactly representable in
 * fp16 -- true of checkpoints released in half precision -- and, to read that one flag back, synchronises `stream`;
 * such a matrix's GEMM then drops the identically-zero x_hi w_lo cross term. */
int awt_encoder_set_weight(awt_encoder* e, const char* hf_name, const float* data, const int64_t* shape, int rank,
                           void* stream);

/* How many of the encoder's projection matrices (per layer: the fused q|k|v block, out_proj, fc1, fc2) hold only fp16-representable values
 * -- i.e. run the exact-weight GEMM form with one cross term fewer -- and how many such matrices there are.  Measured at upload time
 * (awt_encoder_set_weight); bench.py derives its MFMA products per fragment pair and algorithmic bytes from this, not from a flag.
 * n_exact = 0 in the modes that never look (bf16, bf16x3, training). */
int awt_encoder_exact16_matrices(const awt_encoder* e, int* n_exact, int* n_total);

size_t awt_encoder_workspace_bytes(const awt_encoder* e, int B);

/* input_features float32 [B, n_mels, 2*n_ctx] -> last_hidden_state float32 [B, n_ctx, d_model].
 * `n_frames` must equal 2*n_ctx, otherwise AWT_ERR_VALUE (the reference raises ValueError,
 * modeling_whisper.py:612-616). */
int awt_encoder_forward(awt_encoder* e, const float* input_features, int B, int n_frames, float* last_hidden_state,
                        void* workspace, size_t ws_bytes, void* stream);

/* ---- LoRA fine-tune step (K13/K14; build-defined adapters): the encoder half of
 * `trainer.train()` (/root/reference/AB/fineTune.py:186-199), with the frozen base weights and trainable A/B.
 * awt_encoder_forward_train is awt_encoder_forward for the whole batch at once, keeping every activation the backward
 * pass needs in `saved` (>= awt_encoder_train_workspace_bytes(e, B) bytes, 256-byte aligned, caller-owned, must stay
 * untouched until awt_encoder_backward has been enqueued on the same stream).
 * awt_encoder_backward takes d(loss)/d(last_hidden_state) [B, n_ctx, d_model] and writes the gradients of every
 * adapter into `lora_grads` (float32, awt_encoder_lora_grad_count(e) elements): for each layer in order, for each
 * enabled target in the order q_proj, k_proj, v_proj, out_proj, fc1, fc2: dA [r, in_features] then dB [out_features, r], row-major
 * (in / out = d_model except fc1: out = ffn_dim, fc2: in = ffn_dim).  Gradients of frozen weights and of the input features are
 * not produced (nothing below the first adapter needs them). */
/* The backward pass carries 2^k x the gradient from the final LayerNorm on and divides the adapter gradients by 2^k on the way out (exact: powers of two).
 * With backward_terms = 5 the host picks k so that the largest |d loss / d hidden| becomes ~2^6: fp16 planes hold gradients of order one at full precision,
 * 1e-6 ones only as subnormals.  Default 0; any backward_terms accepts it. */
int awt_encoder_set_grad_scale_log2(awt_encoder* e, int k);
size_t awt_encoder_train_workspace_bytes(const awt_encoder* e, int B);
size_t awt_encoder_lora_grad_count(const awt_encoder* e);
int awt_encoder_forward_train(awt_encoder* e, const float* input_features, int B, int n_frames, float* last_hidden_state,
                              void* saved, size_t saved_bytes, void* stream);
int awt_encoder_backward(awt_encoder* e, const float* d_last_hidden_state, int B, void* saved, size_t saved_bytes,
                         float* lora_grads, size_t n_grads, void* stream);
/* Same with flags.  AWT_BWD_ACCUMULATE: lora_grads += the gradients (gradient accumulation over micro-batches) instead
 * of overwriting.  AWT_BWD_ALLREDUCE: after awt_encoder_set_comm, the gradients (after accumulation) are averaged over
 * the communicator's ranks inside this call, layer group by layer group: the group holding the upper layers is
 * all-reduced on the communicator's side stream while the lower layers' backward still runs on `stream`; `stream` waits
 * for the last group before anything enqueued after this call reads lora_grads (one flat buffer, reduced
 * in place, overlapped with the remaining backward). */
enum { AWT_BWD_ACCUMULATE = 1, AWT_BWD_ALLREDUCE = 2 };
int awt_encoder_backward_ex(awt_encoder* e, const float* d_last_hidden_state, int B, void* saved, size_t saved_bytes,
                            float* lora_grads, size_t n_grads, uint32_t flags, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange over RCCL / xGMI (build-defined: the reference has no distributed code, SURVEY.md
 * §5 last row, §8a a17, §8e).  One process per GPU; rank 0 calls awt_comm_unique_id and hands the 128 bytes to the other
 * ranks out of band (torch.distributed's store in finetune.py); every rank then calls awt_comm_create (collective).
 * librccl.so is loaded on first use (dlopen), so the library itself loads on hosts without RCCL.
 * awt_allreduce_{sum,mean}_f32: in-place ncclAllReduce (ncclSum / ncclAvg) of `n` floats at device pointer `buf`,
 * enqueued on `stream`.  The fine-tune step's exchange is ONE flat buffer (r = 16 on q, v of Whisper-small: 589 824
 * floats = 2.36 MB): latency-bound, so it is never split finer than awt_encoder_set_comm's layer groups. */
typedef struct awt_comm awt_comm;
enum { AWT_COMM_ID_BYTES = 128 };
int awt_comm_unique_id(void* id_out /* host, AWT_COMM_ID_BYTES */);
int awt_comm_create(awt_ctx* c, const void* id /* host, AWT_COMM_ID_BYTES */, int rank, int world, awt_comm** out);
void awt_comm_destroy(awt_comm* m);
int awt_comm_world(const awt_comm* m);
int awt_allreduce_sum_f32(awt_comm* m, float* buf, size_t n, void* stream);
int awt_allreduce_mean_f32(awt_comm* m, float* buf, size_t n, void* stream);
/* Timing of the bucket reductions that awt_encoder_backward_ex issues on the communicator's side stream (AWT_BWD_ALLREDUCE): enable != 0 records a HIP
 * event pair around each one from now on (at most 8 between read-outs); the call returns, and clears, what was recorded since the previous call --
 * milliseconds on the side stream and bytes per bucket, in issue order -- so that a multi-GPU run can show that, and how long, the exchange ran. */
int awt_comm_bucket_stats(awt_comm* m, int enable, double* ms, int64_t* bytes, int max, int* n_out);
/* Attach a communicator to an encoder for AWT_BWD_ALLREDUCE; `groups` >= 1 layer groups (2 = upper / lower half;
 * clamped to the layer count).  m = NULL detaches. */
int awt_encoder_set_comm(awt_encoder* e, awt_comm* m, int groups);

/* PCM -> hidden states in one call (log-mel + encoder, chunked so intermediates stay cache-resident):
 * the batched form of `WhisperAudioEncoder.forward` (.charles/music2midi/model.py:42-123).
 * `input_features_out` may be NULL; when given it receives float32 [B, n_mels, 2*n_ctx]. */
size_t awt_audio_encode_workspace_bytes(const awt_encoder* e, int B);
int awt_audio_encode(awt_encoder* e, const void* pcm, int pcm_is_i16, int64_t pcm_stride, const int32_t* n_valid,
                     int max_valid, int B, float* input_features_out, float* last_hidden_state, void* workspace,
                     size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Single-operator entry points (what the encoder is made of; used by the per-kernel parity tests).
 * Row-major float32 in / out; bf16 splitting happens inside.  `terms` = 1 or 3 as in awt_encoder_cfg. */
int awt_op_linear(awt_ctx* c, const float* x /*[M,K]*/, const float* w /*[N,K]*/, const float* bias /*[N] or NULL*/,
                  float* y /*[M,N]*/, int M, int N, int K, int terms, void* workspace, size_t ws_bytes, void* stream);
size_t awt_op_linear_workspace_bytes(int M, int N, int K);
int awt_op_layernorm(awt_ctx* c, const float* x /*[M,d]*/, const float* gamma, const float* beta, float* y, int M, int d,
                     float eps, void* stream);
/* q (already scaled), k, v: float32 [B, H, S, 64]; o: float32 [B, S, H*64] */
int awt_op_attention(awt_ctx* c, const float* q, const float* k, const float* v, float* o, int B, int H, int S,
                     int terms, void* workspace, size_t ws_bytes, void* stream);
size_t awt_op_attention_workspace_bytes(int B, int H, int S);

/* ------------------------------------------------------------------------------------------------------
 * Decoder-side operators of the fine-tune step (scope row "next" #1): what `WhisperForConditionalGeneration.forward` runs after
 * the encoder -- /root/reference/AB/fineTune.py:131,186-199 -> HF:modeling_whisper.py:416-507 (decoder layer), :649-797
 * (decoder), :994-1100 (shift labels, tied projection, cross-entropy).  B x L label tokens (L ~ 12, <= 448) against frozen
 * weights: the linears are weight-bound GEMMs on the encoder's MFMA kernel over weights packed once (`awt_weight`); attention is
 * an fp32 row kernel (causal self-attention, cross-attention over the 1500 encoder positions) that reads q / k / v in place from
 * the row-major outputs of the linears.  All float32 in / out, row-major, caller-owned device memory, deterministic.
 */
typedef struct awt_weight awt_weight;
/* Packs a frozen nn.Linear weight [N, K] (+ bias [N] or NULL) once: N is zero-padded to a multiple of 128 (awt_weight_padded_rows),
 * K must be a multiple of 64; precision 1 (bf16) or 3 (bf16x3); with_transpose keeps the transposed copy that
 * awt_linear_backward_input needs (then K must be a multiple of 128). */
int awt_weight_create(awt_ctx* c, const float* w, const float* bias, int N, int K, int precision, int with_transpose, void* stream,
                      awt_weight** out);
void awt_weight_destroy(awt_weight* w);
int awt_weight_padded_rows(const awt_weight* w);
size_t awt_linear_workspace_bytes(const awt_weight* w, int M, int backward);   /* backward != 0: for awt_linear_backward_input */
/* y [M, Np] = x [M, K] W^T + bias (+ resid [M, Np], may alias y)      F.linear / the residual adds of HF:modeling_whisper.py:468-500 */
int awt_linear_forward(awt_ctx* c, const awt_weight* w, const float* x, const float* resid, float* y, int M, void* workspace,
                       size_t ws_bytes, void* stream);
/* dx [M, K] = dy [M, Np] W                                             (frozen weight: no weight gradient) */
int awt_linear_backward_input(awt_ctx* c, const awt_weight* w, const float* dy, float* dx, int M, void* workspace, size_t ws_bytes,
                              void* stream);
/* Batched products C_z [M, N] = A_z [M, K] B_z [N, K]^T (+ resid_z, may alias C_z), z < batch, split-bf16 operands on the MFMA GEMM, one launch.
 * The operators of the decoder's cross-attention in its ABSORBED form (HF:modeling_whisper.py:284-356 with `key_value_states`, re-associated:
 * q_h K_h^T = (q_h W_k,h) enc^T and P_h V_h = (P_h enc) W_v,h^T + b_v, so that with L ~ 12 label rows per clip the key / value projections
 * of the B x 1500 encoder rows are never computed; csrc/bmm.hip).  B_z is written once by awt_bmm_pack into the GEMM's weight layout (zero-padded
 * to N % 128 == 0, K % 64 == 0) and reused by every product against it; A_z, resid_z and C_z are row-major fp32 with pitches lda / ldo and strides
 * stride_a / stride_o between matrices (elements; strided views inside larger matrices are fine).  N, ldo, stride_o: multiples of 4. */
size_t awt_bmm_packed_bytes(int batch, int N, int K);
int awt_bmm_pack(awt_ctx* c, const float* b, int64_t ldb, int64_t stride_b, int batch, int N, int K, void* packed, size_t packed_bytes, void* stream);
size_t awt_bmm_workspace_bytes(int batch, int M, int K);
int awt_bmm(awt_ctx* c, const float* a, int64_t lda, int64_t stride_a, const void* packed_b, const float* resid, float* out, int64_t ldo,
            int64_t stride_o, int batch, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream);
/* The same with an operand given K-MAJOR (i.e. transposed in memory), optionally as two matrices stacked along K:
 * X[n, k] = (k < K1 ? x1 : x2)[z * stride + (k < K1 ? k : k - K1) * ld + n]; x2 may be NULL when K1 == K.  The backward's
 * d(enc)_b = [P_b^T | dS_b^T] [d(context)_b ; q~_b] reads all four matrices where they lie. */
int awt_bmm_pack_kmajor(awt_ctx* c, const float* b1, const float* b2, int K1, int64_t ldb, int64_t stride_b, int batch, int N, int K, void* packed,
                        size_t packed_bytes, void* stream);
int awt_bmm_kmajor(awt_ctx* c, const float* a1, const float* a2, int K1, int64_t lda, int64_t stride_a, const void* packed_b, const float* resid, float* out,
                   int64_t ldo, int64_t stride_o, int batch, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream);
/* p[r, :] = softmax(scale * s[r, :]) over cols <= 4096 columns (pitch ld; p may alias s), and its backward
 * ds[r, :] = scale * p[r, :] * (dp[r, :] - sum_c p[r, c] dp[r, c]) (ds may alias dp)                      HF:modeling_whisper.py:226-231 */
int awt_op_softmax_rows(awt_ctx* c, const float* s, float* p, int rows, int cols, int64_t ld, float scale, void* stream);
int awt_op_softmax_rows_backward(awt_ctx* c, const float* p, const float* dp, float* ds, int rows, int cols, int64_t ld, float scale, void* stream);
/* x[m, :] = embed_tokens[ids[m], :] + embed_positions[pos0 + m % L, :]     HF:modeling_whisper.py:756-770 */
int awt_op_embed(awt_ctx* c, const int64_t* ids, const float* tok, const float* pos, float* x, int M, int L, int d, int pos0, int vocab,
                 void* stream);
int awt_op_gelu(awt_ctx* c, const float* x, float* y, int64_t n, void* stream);                          /* exact erf GELU */
int awt_op_gelu_backward(awt_ctx* c, const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* dx = dres + d(LayerNorm(x) * gamma + beta)/dx . dy   (frozen affine: no dgamma / dbeta; dres NULL or the gradient arriving over the
 * residual connection; dx may alias dres) */
int awt_op_layernorm_backward(awt_ctx* c, const float* dy, const float* x, const float* gamma, const float* dres, float* dx, int M, int d,
                              float eps, void* stream);
/* Parameter gradients of trainable consumers of the operators (the UrbanSound8K Transformer classifier trains every weight,
 * /root/reference/.charles/spectrogram.py:1059-1130).  dgamma[c] = sum_m dy[m, c] xhat[m, c], dbeta[c] = sum_m dy[m, c];
 * column_sums: sums[c] = sum_m a[m, c] (bias gradients).  Deterministic two-pass reductions; d <= 1280. */
size_t awt_op_param_grad_workspace_bytes(int M, int d);
int awt_op_layernorm_param_grad(awt_ctx* c, const float* dy, const float* x, float* dgamma, float* dbeta, int M, int d, float eps,
                                void* workspace, size_t ws_bytes, void* stream);
int awt_op_column_sums(awt_ctx* c, const float* a, float* sums, int M, int d, void* workspace, size_t ws_bytes, void* stream);
/* CrossEntropyLoss(ignore_index = -100, mean) over the first `vocab` of `ld` columns: *loss and d(loss)/d(logits) [M, ld]
 * (padding columns zero).  scratch: (M + 1) * 4 bytes.                    HF:modeling_whisper.py:1079-1086
 * Only -100 is ignored.  Any other target outside [0, vocab) -- torch raises for it -- makes *loss NaN (the call itself cannot fail
 * without a synchronisation); the Python wrapper validates the labels on the host side and raises before launching. */
int awt_op_cross_entropy(awt_ctx* c, const float* logits, const int64_t* labels, int M, int vocab, int ld, float* loss, float* dlogits,
                         void* scratch, void* stream);
/* softmax(0.125 q k^T [+ causal mask]) v for head_dim 64 with few query rows.  Row (b, i) of q at q + (b Lq + i) ldq + 64 h; row
 * (b, j) of k / v at k + (b Sk + j) ldk + 64 h; o like q with ldo.  causal != 0: key j is visible to query i iff
 * j <= i + causal_off (causal_off = Sk - Lq for cached decoding).  lse [B, H, Lq] (optional) feeds the backward pass.
 * HF:modeling_whisper.py:215-238, 284-356 (q scaled by head_dim^-1/2; eager attention with the causal mask). */
int awt_op_attention_small(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                           float* lse, int B, int H, int Lq, int Sk, int causal, int causal_off, void* stream);
/* dq (layout of q), dk / dv (layout of k / v: every element of the B x Sk x H x 64 blocks is written); delta [B, H, Lq] scratch */
int awt_op_attention_small_backward(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                    const float* dout, int ldo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                                    int H, int Lq, int Sk, int causal, int causal_off, void* stream);
/* The same pair with attention-probability dropout, what torch.nn.MultiheadAttention(dropout = p) applies in train()
 * (/root/reference/.charles/spectrogram.py:977-985): P <- P * keep / (1 - p) after the softmax, before P V.  keep is a pure function of
 * (seed, batch * H + head, query, key) -- splitmix64 of ((bh * Lq + i) * Sk + j) + seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019, top 24 bits
 * as a uniform in [0, 1), kept when >= p -- regenerated by the forward and both backward kernels from the same (drop_p, seed), never stored;
 * urbansound_classifier.attention_keep_mask restates it on the host for tests.  drop_p = 0 is the plain pair. */
int awt_op_attention_small_dropout(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                   float* lse, int B, int H, int Lq, int Sk, int causal, int causal_off, float drop_p, uint64_t seed, void* stream);
int awt_op_attention_small_backward_dropout(awt_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                            const float* dout, int ldo, const float* lse, float* delta, float* dq, float* dk, float* dv, int B,
                                            int H, int Lq, int Sk, int causal, int causal_off, float drop_p, uint64_t seed, void* stream);

/* Process-wide tuning / test hooks.  key "gemm_tile": 0 = choose the GEMM block tile from the shape (default), 64 / 128 / 256 =
 * force the 64 x 128, 128 x 128 or 128 x 256 tile (256 falls back to 128 when N is not a multiple of 256) so that tests can
 * drive every tiling on small shapes; 512 = the 256 x 256 eight-wave tile of the f16f8 GEMM (same results, slower).
 * "gemm_gm": row panels per tile-order group (0 = default).  "attn_shape": workgroup shape of the f16f8 attention kernel, 0 =
 * automatic; 1..5 keep the e4m3 cross terms of P V (plain 4x32 / 4x64 / 6x32 queries, software-pipelined 4 / 8 waves), 6 / 7 =
 * software-pipelined 4 / 8 waves with P V as one fp16 product (what 0 selects for inference).  Only the last choice changes
 * results (within the tolerances of DESIGN.md section 3); every other value is bit-neutral.
 * "gemm_pp": the persistent 256 x 256 eight-wave "ping-pong" f16f8 GEMM (csrc/gemm_pp.h: both operands by LDS-DMA, split-line
 * activations, 16 x 16 MFMAs, one workgroup per CU walking its tiles): 0 = off, 1 = automatic (default: inference launches of at
 * least one tile per CU whose tile count fills its rounds of persistent workgroups to 5/6 or better, on weights that are not fp16-exact), 2 = wherever it applies (N % 256 == 0, K % 64 == 0, K >= 128, no adapter; also
 * awt_op_linear).  "gemm_pp_mask": which projections of a layer may take it, a bit set (1 qkv, 2 out_proj, 4 fc1, 8 fc2; fc2 only with
 * fc1; default 12 = the MLP pair, where it is measurably ahead).  Same products and the same accumulation order per output as the
 * 128 x 256 kernel's 16 x 16 form: bit-identical results where both apply.
 * "gemm_pp_stagger": start-up de-phasing of the persistent workgroups in sixteenths of a tile's K loop per group of CUs (0 = off, default;
 * measured: no gain, profiles/r04_gemm_pp16_epilogue.txt).
 * "gemm_mfma16": 1 (default) = the 128 x 256 f16f8 GEMM issues its products as 16 x 16 MFMAs (both e4m3 cross terms in one block-scaled
 * instruction), 0 = the 32 x 32 form (results differ by the fp32 summation order only). */
int awt_tuning_set(const char* key, int value);

/* ------------------------------------------------------------------------------------------------------
 * In-library kernel timing with HIP events on the caller's stream (bench.py's `roofline` leg).
 * awt_prof_enable(c, mask) makes every launch of the kernel classes whose bit (1 << AWT_PROF_x) is set in `mask`
 * record an event pair (mask 0 = off);
 * awt_prof_collect synchronises the events and returns accumulated milliseconds and launch count
 * for one class, then resets that class. */
enum { AWT_PROF_LOGMEL = 0, AWT_PROF_GEMM = 1, AWT_PROF_ATTENTION = 2, AWT_PROF_LAYERNORM = 3, AWT_PROF_OTHER = 4,
       AWT_PROF_ATTENTION_BWD = 5,     /* the attention backward launches (fine-tune step), apart from the forward kernel */
       AWT_PROF_NCLASSES = 6 };
int awt_prof_enable(awt_ctx* c, int mask);
int awt_prof_collect(awt_ctx* c, int klass, double* total_ms, int64_t* launches, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* AWT_H_ */
