/*
 * rvc_amd.h -- C ABI of librvc_amd.so: the MI355X (gfx950) kernels of the RVC inference hot path.
 *
 * The reference (codename0og/codename-rvc-fork-3) is 100 % Python: it has no FFI / plugin interface,
 * the boundary it offers is Python method signatures.  Each entry point below replaces
 * the body of one reference call; the reference-side binding a maintainer would add is a ctypes stub,
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer named *_dev is a device (HBM) address owned by the caller (e.g. torch tensor.data_ptr());
 *     *_host pointers are host addresses; nothing here allocates caller-visible memory;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); every launch is
 *     asynchronous on it; 0 / NULL = the default stream;
 *   - every function returns 0 on success, non-zero on error; rvc_last_error() describes the last failure
 *     of the calling thread;
 *   - a handle is bound to the device that was current when it was created; distinct handles are independent.
 *     rvc_decoder_forward only READS the handle (weights, tables) and works in the caller's workspace, so several threads
 *     may run forwards of one handle concurrently, each with its own stream and workspace (VoiceConverter.convert_batch
 *     does); create / set_tensor / finalize / set_tap / destroy must not overlap any other call on the same handle;
 *   - tensors are dense, row-major, fp32 unless said otherwise.
 */
#ifndef RVC_AMD_H
#define RVC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVC_AMD_ABI_VERSION 4   /* 3: round-5 additions (K3f, K12-K14, branch streams); 4: round-6 additions (K3f / K3d with one-term taps, K3f's runtime switch, K3u, K10b); no existing signature changed */

/* ---- library ------------------------------------------------------------------------------------ */

int rvc_abi_version(void);
/* NUL-terminated description of the calling thread's last error ("" if none). */
const char *rvc_last_error(void);

/* ---- K1: feature retrieval --------------------------------------------------------------------- *
 * Replaces `index.search(npy, k=8)` + the weighting/blend of
 * rvc/infer/pipeline.py:497-507 (`Pipeline._retrieve_speaker_embeddings`), where `index` is the faiss
 * index read at pipeline.py:555 and `big_npy = index.reconstruct_n(0, ntotal)` (pipeline.py:556).
 * Search is exact brute-force squared L2 over big_npy (faiss IndexFlat semantics: ascending distance,
 * ties -> lower id).  The returned distances are sum_k (q_k - x_k)^2 evaluated in fp32 in a fixed order (faiss' flat
 * scanner form), identical whichever internal regime finds the candidates:
 *   <= 64 queries            one streaming pass over the fp32 index (HBM-bound)
 *   more, >= 16384 rows      fp16 matrix-core screening pass with a proven error bound -> candidate superset -> exact
 *                            re-scoring (csrc/knn_screen.hip); a query whose candidate list overflows is answered by an
 *                            exact scan inside the same call
 *   otherwise                fp32 matrix-core GEMM with per-lane top-8 lists
 */

/* Per-index derived data, built once when the index is loaded ("aux" blob, opaque, caller-owned device memory):
 * ||x_n||^2, an fp16 copy of the rows, and the index-wide maxima the screening bound needs. */
int rvc_knn_index_aux_bytes(int64_t n_rows, int dim, size_t *bytes);
int rvc_knn_index_build(const float *index_dev, int64_t n_rows, int dim, void *aux_dev, size_t aux_bytes, void *stream);

/* bytes of scratch rvc_knn_search needs for (n_rows, n_queries, dim).  In the fp16-screened regime (> 64 queries, >= 16384 rows)
 * this is ~42 KB per query (8192 candidate ids + up to 2048 sample minima + the fp16 copy of the query): 67 MB at the 1599
 * queries of a 30 s clip.  The scratch is per concurrent search -- a caller with several streams in flight holds one per
 * stream -- and grows linearly with n_queries: split very long query sets rather than sizing for them. */
int rvc_knn_workspace_bytes(int64_t n_rows, int64_t n_queries, int dim, int k, size_t *bytes);

/* out_d2_dev [n_queries,k] squared distances ascending; out_ids_dev [n_queries,k] int64 row ids (-1 where the index has
 * fewer than k rows). k must be 8, dim % 32 == 0.  aux_dev: what rvc_knn_index_build filled for this index. */
int rvc_knn_search(const float *index_dev, const void *aux_dev, int64_t n_rows, int dim,
                   const float *queries_dev, int64_t n_queries, int k,
                   float *out_d2_dev, int64_t *out_ids_dev,
                   void *workspace_dev, size_t workspace_bytes, void *stream);

/* Exact ranking of caller-supplied candidates: for every query the best k of the rows listed in cand_ids_dev
 * [n_queries][cap] (int32 row ids, negative = empty) by sum_k (q_k - x_k)^2, ties -> lower id; -1 / +inf where fewer than k
 * candidates exist.  This is the last step of every rvc_knn_search regime, exported for the inverted-file mode: a faiss
 * `IVF{n},Flat` index searched with nprobe lists (extract_index.py:62-64, pipeline.py:499) is "nearest nprobe centroids
 * (rvc_knn_search over the centroids), then rank the members of those lists".  dim: 256, 512, 768 or 1024. */
int rvc_knn_rank_candidates(const float *index_dev, const void *aux_dev, int64_t n_rows, int dim, const float *queries_dev,
                            int64_t n_queries, const int32_t *cand_ids_dev, int cap, int k, float *out_d2_dev,
                            int64_t *out_ids_dev, void *stream);

/* Test hook, per calling thread (searches issued by other host threads keep choosing by shape): 0 = choose the regime by
 * shape (default), 1 = never screen (fp32 GEMM / streaming only), 2 = screen whenever the shape allows it (>= 4096 rows,
 * dim a multiple of 256).  Results do not depend on it.  Candidate ids handed to rvc_knn_rank_candidates that are negative
 * or >= n_rows are ignored. */
int rvc_knn_set_mode(int mode);

/* pipeline.py:500-506: w = (1/d2)^2, w /= sum(w); out = index_rate * sum_k w_k * index[id_k] + (1-index_rate) * feats.
 * feats_dev/out_dev: [n_queries, dim] (may alias).  Neighbours with id < 0 (index smaller than k) are left out of the
 * sums; d2 == 0 is not guarded, as in the reference (an exact duplicate of a query gives inf/inf there too). */
int rvc_knn_blend(const float *index_dev, int dim, const float *feats_dev, const float *d2_dev,
                  const int64_t *ids_dev, int64_t n_queries, int k, float index_rate,
                  float *out_dev, void *stream);

/* ---- K4: RMVPE log-mel front end --------------------------------------------------------------- *
 * Replaces `MelSpectrogram.forward(audio, center=True)` as configured at
 * rvc/lib/predictors/RMVPE.py:438 (n_fft = win = 1024, hop 160, 128 HTK mel bands 30-8000 Hz,
 * log(clamp(., 1e-5)); forward at RMVPE.py:388-417) and the reflect pad of the frame axis to a multiple
 * of 32 done in `mel2hidden` (RMVPE.py:452-455).
 * audio_dev: [batch, n_samples]; mel_dev: [batch, 128, n_frames_padded] where n_frames = n_samples/160 + 1 and
 * n_frames_padded >= n_frames is the row stride (columns >= n_frames are filled by reflection).
 */
int rvc_logmel_workspace_bytes(int batch, int64_t n_samples, size_t *bytes);
int rvc_logmel_rmvpe(const float *audio_dev, int batch, int64_t n_samples, float *mel_dev,
                     int64_t n_frames_padded, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- K4b: the same transform as a handle, for the training-side features --------------------------------------- *
 * Replaces `spectrogram_torch` / `mel_spectrogram_torch` of rvc/train/mel_processing.py:53-146 (feature side of
 * rvc/train/extract): reflect pad `pad` samples per side, frames of n_fft every hop (center = False
 * in torch.stft terms), periodic hann(win_length) centred in n_fft, |X| = sqrt(re^2 + im^2 + mag_eps),
 * mel = log(max(M |X|, log_floor)) with the caller's mel matrix M [n_mels][n_fft/2+1] (host, fp32; librosa.filters.mel in
 * the reference, mel_processing.py:113-119).  The reference's values: n_fft 2048, hop 480 (sr / 100), win 2048,
 * pad (n_fft - hop) / 2, mag_eps 1e-6, log_floor 1e-5, 128 bands at 48 kHz (rvc/configs/48000.json).
 * audio_dev [batch][n_samples]; mel_dev [batch][n_mels][n_frames] and/or spec_dev [batch][n_fft/2+1][n_frames] (either may
 * be NULL); n_frames from rvc_mel_frames.  The handle is read-only after creation. */
typedef struct rvc_mel rvc_mel;
int rvc_mel_create(int n_fft, int hop, int win_length, int pad, float mag_eps, float log_floor, const float *mel_host, int n_mels,
                   rvc_mel **out);
int rvc_mel_destroy(rvc_mel *h);
int rvc_mel_frames(const rvc_mel *h, int64_t n_samples, int64_t *n_frames);
int rvc_mel_workspace_bytes(const rvc_mel *h, int batch, int64_t n_samples, size_t *bytes);
int rvc_mel_forward(const rvc_mel *h, const float *audio_dev, int batch, int64_t n_samples, float *mel_dev, float *spec_dev,
                    void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- input front end: rational resampling --------------------------------------------------------------------- *
 * Replaces `librosa.resample(audio, orig_sr=sr, target_sr=sample_rate, res_type="soxr_vhq")` of `load_audio` /
 * `load_audio_infer` (rvc/lib/utils.py:21-50, 53-85; soxr is third-party and absent: parity unpinned, the filter is this
 * build's own Kaiser design).  y[j] = sum_m h[m] * xup[j * down + (n_taps - 1) / 2 - m] with xup the input zero-stuffed by
 * `up` (scipy.signal.upfirdn / resample_poly convention); float64 throughout. h_dev: the caller's FIR, gain `up`. */
int rvc_resample_poly_f64(const double *x_dev, int64_t n_in, int up, int down, const double *h_dev, int64_t n_taps,
                          double *y_dev, int64_t n_out, void *stream);

/* ---- K6: zero-phase high-pass ------------------------------------------------------------------ *
 * Replaces `signal.filtfilt(bh, ah, audio)` (rvc/infer/pipeline.py:562; 5th-order Butterworth, pipeline.py:23-28)
 * with SciPy's defaults (padtype "odd", padlen 18, lfilter_zi initial conditions), float64, on the device.
 * coef_host (host memory, 17 doubles): b[6], a[6] (a[0] == 1), zi[5] = scipy.signal.lfilter_zi(b, a).
 * x_dev, y_dev: [n] float64 (must not alias).  Agrees with SciPy to the conditioning of the recurrence (~5e-8,
 * see csrc/filtfilt.hip). */
int rvc_filtfilt_workspace_bytes(int64_t n, size_t *bytes);
int rvc_filtfilt_order5(const double *x_dev, int64_t n, const double *coef_host, double *y_dev,
                        void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- K5: RMVPE BiGRU recurrence -------------------------------------------------------------- *
 * Replaces the sequential part of `self.gru(x)[0]` (rvc/lib/predictors/RMVPE.py:515-536, nn.GRU(384, 256,
 * num_layers=1, batch_first, bidirectional)).  The caller computes the input projections for all steps with one
 * GEMM: gi_dev [batch][n_steps][2][768] = W_ih x_t + b_ih (direction 0 = forward, 1 = reverse; gate order r,z,n).
 * whhT_dev [2][256][768] = W_hh transposed per direction, bhh_dev [2][768]; out_dev [batch][n_steps][512]
 * (forward hidden in [0,256), reverse in [256,512), as torch lays it out).
 * workspace_dev: rvc_bigru_workspace_bytes() bytes of scratch for the multi-workgroup variant (4 workgroups per
 * direction exchange h through it); NULL selects the single-workgroup-per-direction variant.
 * The multi-workgroup variant needs its 8 workgroups running at the same time.  A launch that shares the GPU with other
 * streams may see them start late; every wait is bounded, and a (batch item, direction) whose wait ran out is
 * recomputed by the single-workgroup recurrence inside the same call (device-side decision, no host round trip).
 * rvc_bigru_status reports, after synchronising `stream`, how many (batch item, direction) pairs of the LAST forward on
 * that workspace were recomputed; rvc_bigru_set_spin_limit changes the bound (polls per wait; 0 restores the default
 * 2^22) -- a test hook to force the recompute path. */
int rvc_bigru_workspace_bytes(int batch, size_t *bytes);
int rvc_bigru_status(const void *workspace_dev, int batch, int *n_redone_host, void *stream);
int rvc_bigru_set_spin_limit(unsigned polls);
int rvc_bigru_forward(const float *gi_dev, const float *whhT_dev, const float *bhh_dev, float *out_dev,
                      int batch, int64_t n_steps, int hidden, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- K7: softmax attention (HuBERT encoder layers, TextEncoder relative-position layers) ------- *
 * Replaces (a) the softmax(Q K^T * scale) V of transformers' HubertAttention inside
 * `model(feats)["last_hidden_state"]` (call site rvc/infer/pipeline.py:450, wrapper rvc/lib/utils.py:31-34; third-party
 * arithmetic, no mask, no dropout at inference) and (b) `MultiHeadAttention.attention` of the TextEncoder
 * (rvc/lib/algorithm/attentions.py:101-141 with the window-10 relative-position terms of :115-141, 143-180; unmasked,
 * i.e. all frames valid).
 * qkv_dev [batch][n_frames][3][n_heads][head_dim] is the output of ONE fused q/k/v projection GEMM (the caller's);
 * out_dev [batch][n_frames][n_heads * head_dim] is the layout the output projection consumes.  head_dim: 64 or 96.
 * emb_rel_k_dev / emb_rel_v_dev: [21][head_dim] relative-position embeddings shared by the heads (both or neither):
 *   score[i][j] += scale * q_i . emb_rel_k[j - i + 10],  out_i += sum_r p[i][i + r - 10] emb_rel_v[r],  |j - i| <= 10.
 * fp32 throughout (matrix cores in exact-fp32 mode); exp is evaluated as 2^(x log2 e) on the hardware exp unit.
 * head_dim 64 without relative terms (HuBERT) runs both GEMMs on the bf16 matrix cores instead, every fp32 operand split exactly into
 * three bf16 (six products, fp32 accumulate: fp32-level results, max abs error vs float64 ~4e-6 at 1599 frames); its 8-wave workgroups
 * request the CU's whole LDS like every kernel of this library that issues bf16 matrix instructions.
 * workspace_dev: rvc_attention_workspace_bytes() bytes (partial results when the keys are split; the K / V fragment slab of the bf16 form). */
int rvc_attention_workspace_bytes(int batch, int64_t n_frames, int n_heads, int head_dim, size_t *bytes);
int rvc_attention_qkv_f32(const float *qkv_dev, const float *emb_rel_k_dev, const float *emb_rel_v_dev, float *out_dev,
                          int batch, int64_t n_frames, int n_heads, int head_dim, float scale, void *workspace_dev,
                          size_t workspace_bytes, void *stream);

/* ---- K8: conv epilogue of the RMVPE U-Net ------------------------------------------------------ *
 * out = relu(x + bias[c]) + res, the tail of `ConvBlockRes.forward` (rvc/lib/predictors/RMVPE.py:25-64: conv ->
 * BatchNorm (eval, folded by the caller: bias = folded shift) -> ReLU, and after the second conv `+ shortcut(x)` / `+ x`).
 * x_dev/out_dev/res_dev [batch][channels][inner] (inner = H*W, a multiple of 4); bias_dev [channels] or NULL;
 * res_dev NULL for no residual; relu 0/1; out_dev may alias x_dev. */
int rvc_bias_relu_add_f32(const float *x_dev, const float *bias_dev, const float *res_dev, float *out_dev, int batch,
                          int channels, int64_t inner, int relu, void *stream);

/* out = gelu(GroupNorm(num_groups = channels)(x)) -- the normalisation + activation behind the first feature-extractor conv of
 * `transformers`' HubertModel (pipeline.py:450; HubertGroupNormConvLayer: Conv1d -> GroupNorm(512, 512) -> GELU): per (batch,
 * channel) row mean / variance over `length` (float64 accumulation, biased variance, eps inside the square root), affine
 * gamma / beta [channels] (NULL: 1 / 0), exact erf GELU.  x_dev / out_dev [batch][channels][length]; out_dev may alias x_dev. */
int rvc_rownorm_gelu_f32(const float *x_dev, const float *gamma_dev, const float *beta_dev, float *out_dev, int batch, int channels,
                         int64_t length, float eps, void *stream);

/* WaveNet gate of the flow (rvc/lib/algorithm/commons.py:142-157, modules.py:93-97): acts = tanh(x[:H]) * sigmoid(x[H:]).
 * x_dev [batch][2*hidden][length] (the conditioning is already inside: the producing conv adds it as its bias),
 * out_dev [batch][hidden][length]. */
int rvc_gate_tanh_sigmoid_f32(const float *x_dev, float *out_dev, int batch, int hidden, int64_t length, void *stream);

/* Scheduling hint: how many utterances the caller keeps in flight on separate streams (default 1).  With more than one,
 * kernels stop shrinking their tiles to balance a launch across the CUs on its own -- the other streams fill the idle
 * block slots, and the larger tiles re-read their weights from L2 half as often.  No effect on results.
 * This one is the process-wide default (used by rvc_conv1d_forward and by decoder handles without a hint of their own);
 * rvc_decoder_set_concurrency_hint below sets it per handle, so that two converters in one process do not share it. */
int rvc_set_concurrency_hint(int utterances_in_flight);

/* ---- K2/K3: vocoder ("dec" of Synthesizer) ---------------------------------------------------- *
 * Replaces `self.dec(z * x_mask, nsff0, g=g)` at rvc/lib/algorithm/synthesizers.py:254-258, i.e.
 *   RVC_DEC_NSF    HiFiGANNSFGenerator.forward  rvc/lib/algorithm/generators/hifigan_nsf.py:173-207
 *   RVC_DEC_MRF    HiFiGANMRFGenerator.forward  rvc/lib/algorithm/generators/hifigan_mrf.py:339-366
 *   RVC_DEC_REFINE RefineGANGenerator.forward   rvc/lib/algorithm/generators/refinegan.py:368-405
 * Weights are handed over once, by state-dict name with weight-norm already folded (w = g*v/||v||),
 * fp32, host memory; the library repacks them into its own HBM layouts.
 */
typedef struct rvc_decoder rvc_decoder;

enum { RVC_DEC_NSF = 0, RVC_DEC_MRF = 1, RVC_DEC_REFINE = 2 };

typedef struct rvc_decoder_config {
    int kind;                 /* RVC_DEC_* */
    int sample_rate;          /* 32000 / 40000 / 48000 */
    int in_channels;          /* inter_channels, 192 */
    int upsample_initial_channel; /* 512 */
    int gin_channels;         /* 256 */
    int n_ups;                /* number of upsample stages, <= 8 */
    int upsample_rates[8];
    int upsample_kernel_sizes[8];
    int n_res_kernels;        /* 3 */
    int res_kernel_sizes[4];  /* 3,7,11 */
    int res_dilations[4];     /* 1,3,5 (same for every kernel size) */
    int n_res_dilations;      /* 3 */
    int weight_storage;       /* 0: fp32.  1: the ResBlock / MRF-layer conv weights (98 % of the vocoder's weight bytes) are kept
                                 in HBM as bf16 and widened to fp32 inside the conv kernel (BASELINE cfg 4: "bf16 weights ...
                                 alt ResBlock kernel path"); the arithmetic stays fp32.  NSF / MRF only.
                                 This is a NUMERICS mode (bf16-VALUED taps), not a footprint reduction: the layers that run on
                                 the bf16 matrix cores keep fragment slabs of those taps (three bf16 per transformed tap: 126 B
                                 per (c_out, c_in) pair at 11 taps against 44 B of fp32 taps), so the handle holds MORE weight
                                 bytes than with fp32 storage. */
} rvc_decoder_config;

int rvc_decoder_create(const rvc_decoder_config *cfg, rvc_decoder **out);
/* name: state-dict key without the "dec." prefix, e.g. "ups.0.weight", "resblocks.3.convs1.1.bias". */
int rvc_decoder_set_tensor(rvc_decoder *dec, const char *name, const float *data_host,
                           const int64_t *shape, int ndim);
/* call after the last set_tensor: checks that every tensor arrived, builds derived tables */
int rvc_decoder_finalize(rvc_decoder *dec);
int rvc_decoder_destroy(rvc_decoder *dec);

/* samples produced per input frame (prod(upsample_rates)) */
int rvc_decoder_upp(const rvc_decoder *dec);
int rvc_decoder_workspace_bytes(const rvc_decoder *dec, int batch, int64_t n_frames, size_t *bytes);

/* Explicit noise inputs, in the order and shapes the reference draws them (SURVEY §7 hard part 1):
 *   NSF:    src_randn_dev [batch, T*upp]                         (hifigan.py:223; the rand of :189 is zeroed)
 *   MRF:    src_rand_dev [batch, 9] (hifigan_mrf.py:143), src_randn_dev [batch, T*upp, 9] (hifigan_mrf.py:172)
 *   REFINE: src_rand_dev [batch, 1], src_randn_dev [batch, T*upp, 1], adain_randn_dev = the 24 AdaIN draws
 *           concatenated in call order (refinegan.py:111)
 * Unused pointers may be NULL. */
typedef struct rvc_decoder_noise {
    const float *src_rand_dev;
    const float *src_randn_dev;
    const float *adain_randn_dev;
} rvc_decoder_noise;

/* z_dev [batch, in_channels, T] (already multiplied by x_mask), f0_dev [batch, T], g_dev [batch, gin_channels],
 * out_dev [batch, T*upp] = tanh(conv_post(...)). */
int rvc_decoder_forward(rvc_decoder *dec, const float *z_dev, const float *f0_dev, const float *g_dev,
                        const rvc_decoder_noise *noise, int batch, int64_t n_frames, float *out_dev,
                        void *workspace_dev, size_t workspace_bytes, void *stream);

/* Per-handle scheduling hint (see rvc_set_concurrency_hint); 0 = follow the process-wide default.  May be called while
 * forwards of this handle are running on other threads: it only affects forwards that start afterwards. */
int rvc_decoder_set_concurrency_hint(rvc_decoder *dec, int utterances_in_flight);

/* The ResBlock branches of a stage (`xs += self.resblocks[i * self.num_kernels + j](x)`, hifigan_nsf.py:195-203 /
 * hifigan_mrf.py:353-361) depend on the stage's input only.  With side_streams > 0 (or -1: one per branch after the first) a
 * SHORT stage (batch x channels x samples <= 3 rounds of the 256 CUs' 128 x 256 blocks; stage 0 of a 30 s clip at 48 k is
 * 1.17 rounds, so every launch leaves most of the chip idle through its second round) runs its branches side by side on
 * streams owned by the handle, each branch's last launch adding into the running sum after its predecessor's; the caller's
 * stream waits for the last one, so the call is ordered on `stream` exactly as before and results are bit-identical.
 * Default 0 (every launch on the caller's stream): one forward on an idle device gets 5 % (30 s) to 27 % (3 s) shorter, a
 * pipeline that already keeps two utterances in flight does not, and traced kernel durations then contain the time the branches
 * share the chip (profiles/r05_branch_streams.txt).
 * Changes the workspace size (every stream owns two ping-pong buffers): query rvc_decoder_workspace_bytes afterwards. */
int rvc_decoder_set_branch_parallel(rvc_decoder *dec, int side_streams);

/* Debug hook for the parity tests: after stage `stage` of the next forward calls, copy that stage's output
 * ([batch][C_stage][L_stage], the mean of the three ResBlocks) to tap_dev; stage -1 = har_source [batch][T*upp].
 * tap_dev = NULL switches it off. */
int rvc_decoder_set_tap(rvc_decoder *dec, int stage, float *tap_dev);

/* ---- generic fp32 conv1d (the decoder's work-horse, exported for unit tests) -------------------- *
 * y[b,co,t] = out_scale * ( bias[co] + sum_{ci,k} w[co,ci,k] * act(x[b,ci,t + (k - (K-1)/2)*dil]) + res[b,co,t] + acc[b,co,t] )
 * with act = leaky_relu(slope_in) (slope_in = 1 -> identity), zero padding, K odd.
 * w_packed_dev is the [K][C_in][C_out] repack produced by rvc_conv1d_pack_weight.
 * res_dev / acc_dev / bias_dev may be NULL.  C_in % 8 == 0, C_out % 32 == 0. */
int rvc_conv1d_pack_weight(const float *w_host, int c_out, int c_in, int k, float *w_packed_dev, void *stream);
int rvc_conv1d_forward(const float *x_dev, const float *w_packed_dev, const float *bias_dev,
                       const float *res_dev, const float *acc_dev, float *y_dev,
                       int batch, int c_in, int c_out, int64_t length, int k, int dilation,
                       float slope_in, float out_scale, void *stream);

/* ---- multi-GPU: index replication ---------------------------------------------------------------- *
 * The path shards by utterance, one process per GPU (the reference's own multi-GPU precedent is the file striding of
 * rvc/train/extract/extract.py:141-152, 198); its only exchange is making `big_npy` (pipeline.py:556) resident on
 * every GPU, which replaces the per-call `faiss.read_index` of pipeline.py:555 on each worker.  These entries wrap RCCL
 * (bound at run time, so a single-GPU process never loads it):
 *   rank 0:      rvc_comm_unique_id(id)  ->  the host side hands the 128 bytes to every rank (TCP store, file, ...)
 *   every rank:  rvc_comm_create(id, n_ranks, rank, &comm)      collective, on the rank's own current device
 *                rvc_index_broadcast(comm, buf, bytes, root, stream)   ONE ncclBroadcast of raw bytes, in place
 *                rvc_checksum64(buf, bytes, out2, stream)       replica == root's?  (compare the two words across ranks)
 *                rvc_comm_destroy(comm)
 * A communicator is bound to the device current at creation; calls on one communicator must not be concurrent. */
typedef struct rvc_comm rvc_comm;
#define RVC_COMM_ID_BYTES 128
int rvc_comm_unique_id(unsigned char *id_host /* [RVC_COMM_ID_BYTES] */);
int rvc_comm_create(const unsigned char *id_host, int n_ranks, int rank, rvc_comm **out);
int rvc_comm_destroy(rvc_comm *comm);
/* n_ranks as RCCL reports it (ncclCommCount), this rank, RCCL's version code, and which librccl was bound; comm may be
 * NULL (library facts only) and any output pointer may be NULL. */
int rvc_comm_info(const rvc_comm *comm, int *n_ranks, int *rank, int *rccl_version, char *library_path,
                  size_t library_path_bytes);
int rvc_index_broadcast(rvc_comm *comm, void *buf_dev, size_t bytes, int root, void *stream);
/* out2_dev[0] = sum of the buffer's little-endian 32-bit words, out2_dev[1] = sum of (i + 1) * word_i, both mod 2^64
 * (a trailing 1-3 bytes are zero-extended into a last word).  Computed in HBM: the index never crosses PCIe. */
int rvc_checksum64(const void *buf_dev, size_t bytes, uint64_t *out2_dev, void *stream);

/* ---- K10: conv2d 3x3 (padding 1) / 1x1, stride 1, of RMVPE's U-Net blocks (RMVPE.py:13-287) ---------------------------- *
 * y = act(conv(x) + bias) + res  with act = ReLU when relu != 0 -- ConvBlockRes with its BatchNorms folded into the weights
 * (eval mode): conv -> BN -> ReLU, then the skip path added after the activation.  fp32 on the matrix cores, k-ordered
 * accumulation; deep levels split K over several workgroups and sum the partials in a fixed order (bit-reproducible).
 * x [batch][C_in][H][W], y / res [batch][C_out][H][W]; C_in a multiple of 8, W a power of two in 4..128.
 * w_packed: rvc_conv2d_packed_floats() floats filled by rvc_conv2d_pack_weight from torch's [C_out][C_in][kh][kw].
 * workspace: rvc_conv2d_workspace_bytes() bytes (0 for shapes that do not split), caller-owned, reusable across calls on one
 * stream. */
int rvc_conv2d_packed_floats(int c_out, int c_in, int kh, int kw, size_t *out);
int rvc_conv2d_pack_weight(const float *w_host, int c_out, int c_in, int kh, int kw, float *w_dev, void *stream);
int rvc_conv2d_workspace_bytes(int batch, int c_in, int c_out, int height, int width, int kh, int kw, size_t *out);
int rvc_conv2d_forward(const float *x_dev, const float *w_packed_dev, const float *bias_dev, const float *res_dev,
                       float *y_dev, int batch, int c_in, int c_out, int height, int width, int kh, int kw, int relu,
                       void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- K10b: the same 3x3 conv on the bf16 matrix cores, every fp32 operand split exactly into three bf16 (six products of order
 * <= 2^-16, fp32 accumulate: fp32-level results, not bit-equal to K10's k-ordered fp32 chain).  Replaces the same ConvBlockRes convs
 * (RMVPE.py:13-64) for the shapes rvc_conv2d_bf16x3_supported() returns 1 for: C_in a multiple of 16, C_out <= 64 or a multiple of
 * 128, W a power of two in 4..128 (and <= 256 / 128 / 64 pixels for <= 32 / <= 64 / more output channels); 3x3 only.
 * u: rvc_conv2d_bf16x3_weight_bytes() bytes filled by rvc_conv2d_bf16x3_pack_weight from torch's HOST [C_out][C_in][3][3].
 * workspace: rvc_conv2d_bf16x3_workspace_bytes() bytes (the K-split deep levels; partials summed in a fixed order). */
int rvc_conv2d_bf16x3_supported(int c_in, int c_out, int height, int width);
int rvc_conv2d_bf16x3_weight_bytes(int c_out, int c_in, int kh, int kw, size_t *bytes);
int rvc_conv2d_bf16x3_pack_weight(const float *w_host, int c_out, int c_in, int kh, int kw, void *u_dev, void *stream);
int rvc_conv2d_bf16x3_workspace_bytes(int batch, int c_in, int c_out, int height, int width, size_t *out);
int rvc_conv2d_bf16x3_forward(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev, float *y_dev,
                              int batch, int c_in, int c_out, int height, int width, int relu, void *workspace_dev,
                              size_t workspace_bytes, void *stream);

/* ---- the same conv in its fast form (unit-test entry of what the decoder uses for its ResBlock layers) ---------------- *
 * Winograd / Toom-Cook over groups of taps -- F(4,3) for 3 taps, F(4,4) for 7 and 11: identical mathematics, 1.5 / 3.5 / 5.25
 * multiply-adds per output instead of 3 / 7 / 11; fp32 throughout, one layer agrees with float64 to ~5e-7 relative RMS (direct
 * fp32 form: ~2e-7).
 * K in {3, 7, 11}, dilation 1..5, C_in a multiple of 8, C_out a multiple of 32, leaky slope in [0, 1], C_in * L < 2^29.
 * u_dev: 3 G * C_in * C_out floats laid out [3 G][C_in / 2][C_out][2] (the taps, zero-padded to a multiple of three, input
 * channel pairs interleaved) filled by rvc_conv1d_wino_pack_weight from the [C_out][C_in][K] host weights. */
int rvc_conv1d_wino_pack_weight(const float *w_host, int c_out, int c_in, int k, float *u_dev, void *stream);
int rvc_conv1d_wino_forward(const float *x_dev, const float *u_dev, const float *bias_dev, const float *res_dev,
                            const float *acc_dev, float *y_dev, int batch, int c_in, int c_out, int64_t length, int k,
                            int dilation, float slope_in, float out_scale, void *stream);

/* ---- ... and on the bf16 matrix cores with fp32-exact operands (what the decoder uses for its 7- / 11-tap layers at
 * >= 64 channels and its 3-tap layers at c_out % 128 == 0) ------------------------------------------------------------------------------------------------------- *
 * The same F(4,4) form with every fp32 operand split exactly into three bf16 numbers (8 + 8 + 8 significand bits) and the six
 * products of order <= 2^-16 formed by v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped terms are below 2^-23
 * of a product.  One layer agrees with float64 as closely as the fp32 Winograd form does.
 * K in {7, 11}, dilation 1..5, C_in a multiple of 16, C_out a multiple of 64, leaky slope in [0, 1], C_in * L < 2^29;
 * K = 3 (F(4,3): six points, one tap group) with C_out a multiple of 128.
 * u_dev: rvc_conv1d_winobf_weight_bytes() bytes -- the tap transform, evaluated in float64 on the host, rounded to fp32, split
 * into three bf16 and laid out as matrix-instruction fragments by rvc_conv1d_winobf_pack_weight, in the order the kernel that will
 * read them consumes them: C_out % 128 == 0 runs csrc/winobf2.hip (one transform point per wave, 128-channel x 64-column blocks),
 * other C_out csrc/winobf.hip (64 x 128 blocks) -- a slab is only valid for the (C_out, C_in, K) it was packed for.
 * Both kernels request the CU's whole LDS (like the bf16x3 GEMM below): no other workgroup ever shares their CU. */
int rvc_conv1d_winobf_weight_bytes(int c_out, int c_in, int k, size_t *bytes);
int rvc_conv1d_winobf_pack_weight(const float *w_host, int c_out, int c_in, int k, void *u_dev, void *stream);
int rvc_conv1d_winobf_forward(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev,
                              const float *acc_dev, float *y_dev, int batch, int c_in, int c_out, int64_t length, int k,
                              int dilation, float slope_in, float out_scale, void *stream);

/* ---- K3f: one ResBlock (dilated conv -> conv) pair of the 32- / 64-channel stages in ONE launch ------------------------------- *
 * Replaces, per dilation d of ResBlock.forward (rvc/lib/algorithm/residuals.py:75-86; the MRF layer of
 * rvc/lib/algorithm/generators/hifigan_mrf.py has the same body):
 *     xt = leaky_relu(x, slope); xt = conv1_d(xt); xt = leaky_relu(xt, slope); xt = conv2(xt); x = xt + x
 * plus the generator's running sum over its parallel ResBlocks and the final 1 / n (hifigan_nsf.py:196-205):
 *     y = out_scale * (conv2(leaky(conv1_d(leaky(x)) + b1)) + b2 + x [+ acc])
 * Direct form on the bf16 matrix cores, every fp32 operand (taps, activations, the intermediate) split exactly into three
 * bf16 and the six products of order <= 2^-16 accumulated in fp32 (csrc/resblock_bf.hip); the intermediate stays in LDS.
 * C in {32, 64} with K in {3, 7, 11}, or C = 128 with K in {3, 7}; dilation 1..5 (conv2's is 1), leaky slope in [0, 1],
 * C * L * 4 < 2^31; x and y must not alias.
 * u_dev: rvc_resblock_bf16x3_weight_bytes() bytes filled by rvc_resblock_bf16x3_pack_weight from the two [C][C][K] host weights.
 * Persistent: one 8-wave workgroup per CU that requests the CU's whole LDS (the rule for every bf16 matrix kernel here). */
int rvc_resblock_bf16x3_weight_bytes(int c, int k, size_t *bytes);
int rvc_resblock_bf16x3_pack_weight(const float *w1_host, const float *w2_host, int c, int k, void *u_dev, void *stream);
int rvc_resblock_bf16x3_forward(const float *x_dev, const void *u_dev, const float *b1_dev, const float *b2_dev,
                                const float *acc_dev, float *y_dev, int batch, int c, int64_t length, int k, int dilation,
                                float slope, float out_scale, void *stream);
/* The same pair with bf16-VALUED taps -- the weights of a handle created with weight_storage = 1 (BASELINE cfg 4: "MRF-HiFi-GAN,
 * bf16 weights, alt ResBlock kernel path"; MRFLayer.forward, rvc/lib/algorithm/generators/hifigan_mrf.py:13-83, is the body above).
 * A bf16-valued tap IS the first term of its split (w = w_0 exactly), so w x = w_0 x_0 + w_0 x_1 + w_0 x_2 is the whole product:
 * THREE matrix products per multiply-add instead of six, one 1 KiB tap fragment per (tap, 16 input channels, 32 output channels)
 * instead of three; activations and the intermediate keep their exact three-way split, so the result is the fp32 result of the
 * pair on the rounded taps to the same ~2^-23 as the six-product form.  pack_weight ROUNDS the fp32 taps it is given to bf16
 * (round to nearest even, what weight_storage = 1 and torch's .bfloat16() do).  Shapes and rules as above. */
int rvc_resblock_bf16w_weight_bytes(int c, int k, size_t *bytes);
int rvc_resblock_bf16w_pack_weight(const float *w1_host, const float *w2_host, int c, int k, void *u_dev, void *stream);
int rvc_resblock_bf16w_forward(const float *x_dev, const void *u_dev, const float *b1_dev, const float *b2_dev,
                               const float *acc_dev, float *y_dev, int batch, int c, int64_t length, int k, int dilation,
                               float slope, float out_scale, void *stream);
/* ---- K3d: ONE square conv with bf16-VALUED taps, direct form, for the layers the fused pair cannot hold in LDS ------------------ *
 * Replaces one conv of the same ResBlock / MRFLayer body (rvc/lib/algorithm/residuals.py:75-86, hifigan_mrf.py:13-83) at
 * C = 256 (K = 3, 7, 11) and C = 128 (K = 11; K = 3 / 7 accepted too) in a decoder handle created with weight_storage = 1:
 *     y = out_scale * (conv_d(leaky(x, slope)) + bias [+ res] [+ acc])
 * One-term taps (rounded to bf16 by pack_weight, round to nearest even) against activations split exactly into three bf16: three
 * matrix products per multiply-add, no Winograd transform; persistent 8-wave workgroups walking (64 columns x all channels) tiles
 * (csrc/convbf1.hip).  dilation 1..5, C * L * 4 < 2^31; y must not alias x (res and acc may alias y). */
int rvc_conv1d_bf16w_weight_bytes(int c, int k, size_t *bytes);
int rvc_conv1d_bf16w_pack_weight(const float *w_host, int c, int k, void *u_dev, void *stream);
int rvc_conv1d_bf16w_forward(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev,
                             const float *acc_dev, float *y_dev, int batch, int c, int64_t length, int k, int dilation,
                             float slope_in, float out_scale, void *stream);
/* ---- K3u: the vocoder's upsampling step on the bf16 matrix cores (exact bf16x3 operands) -------------------------------------- *
 * Replaces, per stage i of HiFiGANNSFGenerator.forward / HiFiGANMRFGenerator.forward (rvc/lib/algorithm/generators/hifigan_nsf.py:184-193,
 * hifigan_mrf.py:349-357):   x = leaky_relu(x, slope); x = ups[i](x); x = x + noise_convs[i](har_source)
 *     y[co][t] = bias[co] + ConvTranspose1d(leaky(x))[co][t] + sum_k noise_w[co][k] har[t nc_stride + k - nc_pad]
 * ConvTranspose1d(c_in, c_out, ksize, stride = rate, padding = pad) with ksize <= 2 rate in polyphase form (a GEMM with rows
 * (channel, phase), two taps); the noise conv (one input channel, nc_k taps, stride nc_stride; nc_k = 0: none) enters as
 * (rate - 1) nc_stride + nc_k < 64 extra GEMM rows read straight from har_source [batch][har_len], plus a row of ones whose taps are the
 * bias (so pack_weight takes it: ups bias + noise-conv bias [c_out]); without a noise conv bias_dev is added on the way out.
 * rate in {2, 8, 10, 12}, c_in % 64 == 0; length_out = (length_in - 1) rate - 2 pad + ksize.  Every fp32 operand split exactly into
 * three bf16, six products, fp32 accumulate (csrc/upsbf.hip); persistent whole-CU workgroups. */
int rvc_upsample_bf16x3_weight_bytes(int c_in, int c_out, int rate, int ksize, int nc_k, int nc_stride, size_t *bytes);
int rvc_upsample_bf16x3_pack_weight(const float *up_w_host, const float *noise_w_host, const float *bias_host, int c_in, int c_out,
                                    int rate, int ksize, int nc_k, int nc_stride, void *u_dev, void *stream);
int rvc_upsample_bf16x3_forward(const float *x_dev, const float *har_dev, int64_t har_len, const void *u_dev, const float *bias_dev,
                                float *y_dev, int batch, int c_in, int c_out, int64_t length_in, int rate, int ksize, int pad,
                                int nc_k, int nc_stride, int nc_pad, float slope_in, void *stream);
/* Process-wide runtime switch for K3f inside rvc_decoder_finalize (default 1): decoder handles finalized while it is 0 keep the
 * (conv, conv) pairs of their narrow stages on the unfused kernels (csrc/resblock.hip, winobf.hip, wino.hip) -- the fall-back an
 * operator reaches for without rebuilding the library.  The entry points above are not affected. */
int rvc_resblock_bf16x3_set_enabled(int enabled);

/* ---- K11: fp32 GEMM / strided conv1d on the bf16 matrix cores with fp32-exact operands ------------------------------------ *
 * Replaces the fp32 library GEMMs behind `transformers`' HubertModel at rvc/infer/pipeline.py:450: the attention / FFN
 * projections (nn.Linear: y = act(x W^T + b) + res, x [n_rows][in] row-major) and the stride-2 convolutions of the feature
 * extractor (nn.Conv1d without padding + GELU, x [batch][C_in][L] channel-major).  Every fp32 operand is split exactly into three
 * bf16 numbers and the six products of order <= 2^-16 are accumulated in fp32 (csrc/gemmbf.hip).
 * The kernel launches one 8-wave workgroup per CU and requests the CU's whole LDS: a workgroup issuing bf16 matrix instructions
 * that shares a CU with a workgroup of the fp32 Winograd kernel behind rvc_decoder_forward corrupts THAT kernel's results
 * (measured on MI355X, profiles/r03_mfma_cohabitation.txt), so this one never shares; results agree with float64 as
 * closely as an fp32 GEMM does.  act: 0 none, 1 GELU (erf form, what torch.nn.functional.gelu computes).
 * Weights: rvc_gemm_bf16x3_weight_bytes() bytes filled by rvc_gemm_bf16x3_pack_weight from the [out][in] (linear, conv_taps = 1) or
 * [C_out][C_in][taps] (conv, conv_taps = taps, k_total = taps * C_in) host tensor.  out / C_out a multiple of 128, in / C_in of 16. * rvc_conv1d_bf16x3 also takes c_in = 1 with k <= 16, no padding, batch 1 (HuBERT's first layer: Conv1d(1, 512, 10, stride 5)); the
 * kernel reads nothing beyond x[l_in - 1].  rvc_gemm_bf16x3_weight_bytes / _pack_weight pad such a weight to one k16 step.
 */
int rvc_gemm_bf16x3_weight_bytes(int m, int k_total, size_t *bytes);
int rvc_gemm_bf16x3_pack_weight(const float *w_host, int m, int k_total, int conv_taps, void *a_dev, void *stream);
int rvc_linear_bf16x3(const float *x_dev, const void *a_dev, const float *bias_dev, const float *res_dev, float *y_dev,
                      int64_t n_rows, int in_features, int out_features, int act, void *stream);
int rvc_conv1d_bf16x3(const float *x_dev, const void *a_dev, const float *bias_dev, float *y_dev, int batch, int c_in,
                      int c_out, int64_t l_in, int k, int stride, int padding, int act, void *stream);

/* ---- K12: the GEMMs of HuBERT's transformer layers with BOTH operands pre-split ------------------------------------------------ *
 * Replaces the nn.Linear / LayerNorm / GELU modules of `transformers`' HubertEncoderLayer (q/k/v, attention output, feed-forward)
 * behind rvc/infer/pipeline.py:450.  Same exact bf16x3 arithmetic as K11; the activations arrive as three bf16 planes
 * [split][n_rows_padded][features] (their sum is the fp32 value exactly), so the kernel's main loop is LDS-DMA + matrix
 * instructions only (csrc/linbf.hip).  Rows n_rows .. n_rows_padded - 1 of a plane are never read into a stored result and
 * never written: they need no initialisation.
 * rvc_split_rows_bf16x3: fp32 [n_rows][k] -> planes.
 * rvc_linear_bf16x3_presplit: a_dev = rvc_gemm_bf16x3_pack_weight's slab of the [out][in] weight.  mode 0: y_dev [n_rows][out]
 *   = x W^T + bias; mode 1: ys_dev planes [3][n_rows_padded][out] of gelu(x W^T + bias) (erf form); mode 2: y_dev
 *   [k_parts][n_rows][out] partial sums over k_parts equal slices of in_features (no bias) -- the grid filler for the 768-wide
 *   outputs; mode 3: y_dev [n_rows][out] = gelu(x W^T + bias).  in_features a multiple of 32 (and of 32 k_parts), n_rows_padded a
 *   multiple of 128.  One 8-wave workgroup per CU, whole LDS.
 * rvc_bias_residual_layernorm_bf16x3: y = LayerNorm(sum of the parts + bias + res) * gamma + beta over `features` (256, 768 or 1024),
 *   written as fp32 (y_dev, may be NULL) and as planes (ys_dev, may be NULL): HubertEncoderLayer's `hidden = layer_norm(hidden +
 *   dropout(attn))` / `final_layer_norm(hidden + feed_forward(hidden))` fused with the split-K reduction. */
int rvc_split_rows_bf16x3(const float *x_dev, void *xs_dev, int64_t n_rows, int64_t n_rows_padded, int k, void *stream);
int rvc_linear_bf16x3_presplit(const void *xs_dev, const void *a_dev, const float *bias_dev, float *y_dev, void *ys_dev,
                               int64_t n_rows, int64_t n_rows_padded, int in_features, int out_features, int mode, int k_parts,
                               void *stream);
int rvc_bias_residual_layernorm_bf16x3(const float *parts_dev, int n_parts, const float *bias_dev, const float *res_dev,
                                       const float *gamma_dev, const float *beta_dev, float eps, float *y_dev, void *ys_dev,
                                       int64_t n_rows, int64_t n_rows_padded, int features, void *stream);

/* ---- K12 / K13: HuBERT's feature extractor on time-major frames ---------------------------------------------------------------- *
 * Replaces `transformers`' HubertFeatureEncoder behind rvc/infer/pipeline.py:450 (conv_layers[0] = HubertGroupNormConvLayer:
 * Conv1d(1, 512, 10, stride 5, bias = False) -> GroupNorm(512, 512) -> GELU; conv_layers[1..6] = HubertNoLayerNormConvLayer:
 * Conv1d(512, 512, 3 or 2, stride 2, bias = False) -> GELU) for batch 1.  The activations travel TIME-MAJOR, [frame][channel], as
 * three bf16 planes [split][frames_padded][channels]: the window of a strided conv -- taps x channels values -- is then one
 * contiguous run starting stride x channels after its predecessor's, i.e. the conv IS K12's GEMM with a different row stride
 * and needs no im2col, no transposes and no fp32 copy of the 196 MB first-layer output.
 * rvc_hubert_conv0_frames_bf16x3 (csrc/hubert_front.hip): layer 0 from the 16 kHz samples; the conv is evaluated twice (float64
 *   statistics over the clip, then normalise + GELU + split) instead of being stored.  w_dev [channels][taps] fp32, taps = 10,
 *   channels a multiple of 64; workspace of rvc_hubert_conv0_workspace_bytes(channels); ys_dev planes with
 *   n_frames_padded >= (n_samples - taps) / stride + 1 rows.
 * rvc_conv1d_frames_bf16x3 (csrc/linbf.hip): xs_dev planes [3][n_frames_in_padded][channels]; a_dev = rvc_gemm_bf16x3_pack_weight's
 *   slab of the [out][taps * channels] matrix W[o][k * channels + c] = conv.weight[o][c][k]; output frames (n_frames_in - taps) /
 *   stride + 1; modes 0 / 1 / 3 of rvc_linear_bf16x3_presplit (ys_dev planes [3][n_frames_out_padded][out_channels] or y_dev
 *   [frames_out][out_channels] fp32).  taps * channels a multiple of 32, stride * channels of 8, out_channels of 128,
 *   n_frames_out_padded of 128.  Rows of the last 128-frame tile beyond the input are read (inside the planes' allocation, or
 *   as zeros beyond it) but their results are not stored. */
int rvc_hubert_conv0_workspace_bytes(int channels, size_t *bytes);
int rvc_hubert_conv0_frames_bf16x3(const float *wav_dev, int64_t n_samples, const float *w_dev, int channels, int taps, int stride,
                                   const float *gamma_dev, const float *beta_dev, float eps, void *workspace_dev,
                                   size_t workspace_bytes, void *ys_dev, int64_t n_frames_padded, void *stream);
int rvc_conv1d_frames_bf16x3(const void *xs_dev, int64_t n_frames_in, int64_t n_frames_in_padded, int channels, int taps, int stride,
                             const void *a_dev, const float *bias_dev, float *y_dev, void *ys_dev, int64_t n_frames_out_padded,
                             int out_channels, int mode, void *stream);

/* ---- K14: HuBERT's convolutional position embedding -------------------------------------------------------------------------- *
 * Replaces `transformers`' HubertPositionalConvEmbedding.forward behind rvc/infer/pipeline.py:450 (weight-normed Conv1d(D, D, 128,
 * padding = 64, groups = 16) -> HubertSamePadLayer (drop the last frame) -> GELU) for one clip, time-major:
 *   y_dev[t][o] = gelu(bias[o] + sum_{k, c} W[o][c][k] * x_dev[t + k - padding][group(o) * CG + c]),  t = 0 .. n_frames - 1,
 * x_dev / y_dev [n_frames][d] fp32 (the caller adds the residual and applies encoder.layer_norm, e.g. with
 * rvc_bias_residual_layernorm_bf16x3 and n_parts = 1).  bf16 matrix cores, fp32 operands split exactly into three bf16
 * (csrc/posconv.hip): a block keeps its 128 + taps - 1 frames of one group's channels in LDS for the whole K loop, the weights
 * stream through by LDS-DMA.  d / groups = 48 or 64, taps <= 128.  a_dev: rvc_posconv_bf16x3_weight_bytes() bytes filled by
 * rvc_posconv_bf16x3_pack_weight from the host tensor conv.weight [d][d / groups][taps] (weight norm folded).
 * One 8-wave workgroup per CU, whole LDS. */
int rvc_posconv_bf16x3_weight_bytes(int d, int groups, int taps, size_t *bytes);
int rvc_posconv_bf16x3_pack_weight(const float *w_host, int d, int groups, int taps, void *a_dev, void *stream);
int rvc_posconv_gelu_bf16x3(const float *x_dev, const void *a_dev, const float *bias_dev, float *y_dev, int64_t n_frames, int d,
                            int groups, int taps, int padding, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RVC_AMD_H */
