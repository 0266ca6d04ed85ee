/*
 * gww.h -- C ABI of libgww.so, the MI355X (gfx950) hot path of GW-Whisper.
 *
 * The reference (chayanchatterjee/GW-Whisper) is pure Python; its hot path is
 * the object protocol of three third-party classes.
 * There is therefore no FFI in the reference to mirror symbol-for-symbol; each
 * entry point below names the reference call it replaces (file:line relative to
 * the reference tree, "HF:" = the transformers package the reference imports).
 * The Python shim in gw_whisper_amd/ binds these with ctypes and presents the
 * reference's own call surface (WhisperFeatureExtractor / WhisperEncoder /
 * peft get_peft_model); INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless said otherwise
 *   - every launch is asynchronous on the caller's hipStream_t (passed as void*)
 *   - return value: 0 = ok, negative = error (gww_last_error() has the text)
 *   - the library keeps no global mutable state besides the per-thread error
 *     string; handles are thread-compatible (one handle per thread/stream)
 *   - plain C types only: no torch, no C++ in the signatures
 */
#ifndef GWW_H
#define GWW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GWW_VERSION 107  /* 0.1.7: + gww_dora_merge_batch_f32 (all adapted projections of a step in one launch), gww_conv1_gelu_bf16 (conv1 read from the [B, 80, T] feature layout); 0.1.6: + gww_qadapter_cnn_backward_f32 / _workspace_bytes (the Q-adapter CNN's backward as HIP kernels); 0.1.5: + gww_gemm_bf16_v4_split (explicit column split; no environment switch is read by the library any more); 0.1.4: gww_mlp_fused_bf16 / gww_attn_out_mlp_fused_bf16 with the q / k / v tail return x_next over x (x_out keeps x_new); 0.1.3: + gww_logmel_host_f32 (fork-safe CPU twin of the front end); 0.1.2: + whitening kernels, gww_qadapter_tail_f32, gww_attention_bwd_log2q_bf16, gww_lnqkv_fused_bf16, gww_attn_out_mlp_fused_bf16, gww_mlp_pack_op_bf16; the gww_mlp_pack_bf16 stream carries W1 / 8 and 8 W2 */

#define GWW_OK 0
#define GWW_ERR_ARG (-1)      /* bad argument (shape, null pointer, unsupported size) */
#define GWW_ERR_HIP (-2)      /* a HIP runtime call failed */
#define GWW_ERR_WORKSPACE (-3)/* caller's workspace too small */
#define GWW_ERR_STATE (-4)    /* handle not ready (weights not set) */

/* compute precision of the encoder path */
#define GWW_PREC_BF16 0  /* bf16 MFMA operands, fp32 accumulate / LN / softmax / residual */
#define GWW_PREC_F32 1   /* fp32 MFMA (v_mfma_f32_16x16x4_f32): exact fp32, the parity gate */

int gww_version(void);
const char* gww_last_error(void);

/* --------------------------------------------------------------------------
 * Front end: log-mel features.
 * Replaces WhisperFeatureExtractor.__call__ as used at
 *   Signal_vs_Noise/src/dataset.py:20-21,40 ; Glitch_classification/src/dataset.py:46
 *   (impl HF:models/whisper/feature_extraction_whisper.py:135-168,193-346).
 * wave   [n_seg, wave_stride] fp32, the first n_samples of each row are the
 *        16 kHz samples (zero padded / truncated to 480000 like HF does)
 * out    [n_seg, 80, 3000] fp32  == input_features
 * seg_max[n_seg] fp32 scratch (per-segment max of the raw log10 mel)
 * Exact shortcut: frames that only see zero padding are filled with the one
 * constant HF would produce; only ceil((n_samples+200)/160) frames run a DFT.
 * -------------------------------------------------------------------------- */
typedef struct gww_frontend gww_frontend;
int gww_frontend_create(gww_frontend** out);          /* uploads window / twiddle / filterbank tables */
void gww_frontend_destroy(gww_frontend* fe);
int gww_logmel_f32(gww_frontend* fe, const float* wave, int n_seg, int n_samples,
                   long wave_stride, float* out, float* seg_max, void* stream);
/* The same front end on the HOST: wave and out are HOST pointers, nothing here touches the GPU (no HIP call, no
 * handle, no global mutable state), so it may be called from forked DataLoader worker processes -- where the
 * reference calls the extractor: Signal_vs_Noise/src/dataset.py:12,20-21 under src/train.py:224-225
 * (num_workers 12).  Same arithmetic and the same dead-frame shortcut; double-precision FFT, fp32 result. */
int gww_logmel_host_f32(const float* wave, int n_seg, int n_samples, long wave_stride, float* out);

/* --------------------------------------------------------------------------
 * Encoder.  Replaces WhisperEncoder.forward as built at
 *   Signal_vs_Noise/src/train.py:227-228,240 ; Glitch_classification/src/train.py:159 ;
 *   MLGWSC-1/train.py:660-663 ; MLGWSC-1/inference.py:408-410
 *   (impl HF:models/whisper/modeling_whisper.py:592-646, layers :379-413,
 *    attention :284-356 / :215-238).
 * -------------------------------------------------------------------------- */
typedef struct {
  int d_model;   /* 384 / 512 / 768 ... multiple of 64 */
  int n_layers;
  int n_heads;   /* d_model / 64 (head_dim is 64 for every Whisper size) */
  int ffn;       /* 4 * d_model */
  int n_mels;    /* 80 */
  int t_in;      /* 3000 mel frames  -> t_in/2 tokens */
} gww_enc_cfg;

/* fp32 master weights, HF layout ([out,in] linears, [out,in,3] convs). */
typedef struct {
  const float* conv1_w; const float* conv1_b;   /* [d,80,3] [d] */
  const float* conv2_w; const float* conv2_b;   /* [d,d,3]  [d] */
  const float* pos;                             /* [t_in/2, d] embed_positions.weight */
  const float* ln_w; const float* ln_b;         /* final layer_norm */
} gww_enc_globals;

typedef struct {
  const float* ln1_w; const float* ln1_b;       /* self_attn_layer_norm */
  const float* q_w; const float* q_b;           /* [d,d] [d]  (DoRA-merged when adapted) */
  const float* k_w;                             /* [d,d], no bias (HF:modeling_whisper.py:279) */
  const float* v_w; const float* v_b;
  const float* o_w; const float* o_b;           /* out_proj */
  const float* ln2_w; const float* ln2_b;       /* final_layer_norm */
  const float* fc1_w; const float* fc1_b;       /* [ffn,d] [ffn] */
  const float* fc2_w; const float* fc2_b;       /* [d,ffn] [d] */
} gww_enc_layer;

typedef struct gww_encoder gww_encoder;

int gww_encoder_create(const gww_enc_cfg* cfg, gww_encoder** out);
void gww_encoder_destroy(gww_encoder* enc);
/* (Re)pack the weights into the library-owned kernel layouts (bf16 [N,K] panels,
 * q scale folded in, conv taps flattened).  Call again after an optimizer step /
 * DoRA merge.  Asynchronous on `stream`. */
int gww_encoder_set_weights(gww_encoder* enc, const gww_enc_globals* g,
                            const gww_enc_layer* layers, int n_layers, void* stream);
/* Re-pack only the weight groups that changed since the last full gww_encoder_set_weights (an optimizer step on
 * the DoRA parameters touches the q / k / v / out projections only).  globals_or_null: stem / positions / final
 * LayerNorm, NULL if unchanged.  layer_dirty[i]: bit 0 = q, k, v projections + self_attn_layer_norm, bit 1 =
 * out_proj, bit 2 = fc1 + final_layer_norm, bit 3 = fc2; pointers of clean groups are not read. */
int gww_encoder_update_weights(gww_encoder* enc, const gww_enc_globals* globals_or_null,
                               const gww_enc_layer* layers, int n_layers, const unsigned* layer_dirty,
                               void* stream);
/* bytes of caller-owned scratch gww_encoder_forward needs for `batch` segments */
size_t gww_encoder_workspace_bytes(const gww_encoder* enc, int batch, int precision);
/* mel [batch,80,3000] fp32 (== input_features).  Either output may be NULL:
 *   last_hidden [batch, 1500, d] fp32 (== .last_hidden_state)
 *   last_token  [batch, d]       fp32 (== .last_hidden_state[:, -1, :],
 *                Signal_vs_Noise/src/model.py:25-26) */
int gww_encoder_forward(gww_encoder* enc, const float* mel, int batch, int precision,
                        void* workspace, size_t workspace_bytes,
                        float* last_hidden, float* last_token, void* stream);

/* Dual-stream split (off by default): batches of >= 64 segments are processed as two independent
 * half batches on two library-owned streams forked from / joined to the caller's stream, so the
 * HBM-bound kernels of one half overlap the MFMA-bound kernels of the other on different CUs.
 * Changes gww_encoder_workspace_bytes; results are bit-identical (segments are independent). */
int gww_encoder_set_split(gww_encoder* enc, int on);

/* Optional per-kernel timing of the forward (hipEvents on the caller's stream around every
 * launch; ~30 events per forward).  Classes: gww_encoder_trace_classes() entries named by
 * gww_encoder_trace_class_name(i).  gww_encoder_trace_read sums elapsed ms and launch counts per
 * class since the last read (it blocks until the recorded events completed). */
int gww_encoder_trace_enable(gww_encoder* enc, int on);
int gww_encoder_trace_read(gww_encoder* enc, float* ms, int* counts);
int gww_encoder_trace_classes(void);
const char* gww_encoder_trace_class_name(int i);

/* --------------------------------------------------------------------------
 * DoRA.  Replaces peft's DoRA Linear (peft 0.12.0 tuners/lora/dora.py) created at
 *   Signal_vs_Noise/src/train.py:263-264 ; MLGWSC-1/train.py:695-696.
 *   W' = W0 + s B A ; n = ||W'|| per output row ; W_eff = (m/n)[:,None] W'
 * w0 [d_out,d_in], a [r,d_in], b [d_out,r], m [d_out]  ->  w_eff [d_out,d_in],
 * norm_out [d_out] (may be NULL).  All fp32.
 * -------------------------------------------------------------------------- */
int gww_dora_merge_f32(const float* w0, const float* a, const float* b, const float* m,
                       float scaling, int d_out, int d_in, int r,
                       float* w_eff, float* norm_out, void* stream);

/* --------------------------------------------------------------------------
 * conv1 of the stem straight from the HF feature layout (bf16 path, kernel-level entry; the encoder calls the same
 * kernel).  Replaces  nn.functional.gelu(self.conv1(input_features))  of HF:models/whisper/modeling_whisper.py:619-620
 * (reached from Signal_vs_Noise/src/model.py:25, Glitch_classification/src/model.py, MLGWSC-1/inference.py:353-392).
 *   mel [B, 80, T] fp32, conv1_w [d, 80, 3] fp32, conv1_b [d] fp32; w_scratch_bf16: d * 256 bf16 of scratch (packed
 *   taps); c1_out [B, T + 2, d] bf16 token-major, rows 0 and T + 1 of every segment zero (the padding conv2 reads).
 *   d in {384, 512, 768, 1024}.
 * -------------------------------------------------------------------------- */
int gww_conv1_gelu_bf16(const float* mel, const float* conv1_w, const float* conv1_b, void* w_scratch_bf16,
                        void* c1_out, int B, int T, int d, void* stream);

/* The same merge for every adapted projection of a model in ONE launch: an optimizer step changes all of them at once
 * (12 modules on whisper-tiny, 36 on whisper-small: Signal_vs_Noise/src/train.py:263-264 targets q, k, v of every
 * layer), and a launch per 384 x 384 matrix is launch-bound.  Items are independent; pointers as above. */
typedef struct {
  const float* w0;      /* [d_out, d_in] frozen base weight */
  const float* a;       /* [r, d_in]     lora_A.weight */
  const float* b;       /* [d_out, r]    lora_B.weight */
  const float* m;       /* [d_out]       lora_magnitude_vector (ones for plain LoRA) */
  float* w_eff;         /* [d_out, d_in] out */
  float* norm_out;      /* [d_out] out, may be NULL */
  float scaling;        /* lora_alpha / r */
  int d_out, d_in, r;
} gww_dora_merge_item;
int gww_dora_merge_batch_f32(const gww_dora_merge_item* items, int n, void* stream);

/* --------------------------------------------------------------------------
 * DoRA training step (bf16).  Replaces loss.backward() through the frozen encoder with DoRA
 * adapters (Signal_vs_Noise/src/train.py:163-168 with the model of :263-269): the forward keeps
 * the activations in a caller-owned arena, the backward returns the A / B / magnitude gradients
 * of the listed projections (peft 0.12.0 dora.py semantics: the weight norm is detached) and,
 * optionally, the gradient w.r.t. the conv-stem output.  The MLP head, the loss and the
 * optimizer stay in torch on the GPU.
 * -------------------------------------------------------------------------- */
typedef struct {
  int layer;            /* encoder layer index */
  int proj;             /* 0 q_proj, 1 k_proj, 2 v_proj, 3 out_proj */
  int r;                /* LoRA rank (8) */
  float scaling;        /* lora_alpha / r */
  const float* A;       /* [r, d]   lora_A.weight */
  const float* B;       /* [d, r]   lora_B.weight */
  const float* mag;     /* [d]      lora_magnitude_vector */
  const float* nrm;     /* [d]      ||W0 + s B A|| rows (norm_out of gww_dora_merge_f32) */
  float* dA;            /* gradients, ACCUMULATED into: zero them once per step */
  float* dB;
  float* dm;
} gww_dora_target;

size_t gww_train_saved_bytes(const gww_encoder* enc, int batch);
size_t gww_train_workspace_bytes(const gww_encoder* enc, int batch);
int gww_encoder_train_forward(gww_encoder* enc, const float* mel, int batch, void* workspace,
                              size_t workspace_bytes, void* saved, size_t saved_bytes,
                              float* last_hidden, int pooled, void* stream);
int gww_encoder_train_backward(gww_encoder* enc, int batch, void* workspace, size_t workspace_bytes,
                               const void* saved, size_t saved_bytes, const float* d_last_hidden,
                               const gww_dora_target* targets, int n_targets, float* d_x0, float* d_mel,
                               int pooled, void* stream);
/* pooled != 0: the caller only uses token T-1 of the output, as every classifier of the reference does
 * (Signal_vs_Noise/src/model.py:25-26 `last_hidden_state[:, -1, :]`).  last_hidden and d_last_hidden are then
 * [batch, d]; everything above the last layer's attention (out_proj, LN2, fc1, GELU, fc2, final LN and their
 * backward) runs on those `batch` rows instead of batch*T, and the attention backward skips the query tiles
 * without gradient.  The forward and the backward of one step must use the same `pooled`. */
/* d_x0 (optional): fp32 [batch*T, d] gradient w.r.t. the conv-stem output.  d_mel (optional): fp32
 * [batch, n_mels, t_in] gradient w.r.t. the input features through the conv stem -- the encoder call is
 * differentiable w.r.t. its input, as MLGWSC-1/train.py:494-504 (trainable Q-adapter in front of the frozen
 * encoder) requires. */

/* --------------------------------------------------------------------------
 * Kernel-level entry points (used by the parity tests and by the Python
 * autograd shim; same conventions).
 * -------------------------------------------------------------------------- */
/* y[M,d] (bf16 if out_bf16 else fp32) = LayerNorm(x[M,d] fp32) * w + b, eps 1e-5 */
int gww_layernorm(const float* x, const float* w, const float* b, void* y, int out_bf16,
                  long M, int d, void* stream);
/* C[M,N] = A[M,K] @ W[N,K]^T + bias ; epilogue: 0 none, 1 exact GELU, 2 += resid (fp32 out).
 * bf16 variant: A, W bf16; C bf16 (epilogue 0/1) or fp32 (epilogue 2).
 * f32 variant : everything fp32. */
int gww_gemm_bf16(const void* A, const void* W, const float* bias, const float* resid, void* C,
                  long M, int N, int K, int epilogue, void* stream);
/* The 256 x 256 x 64 kernel of the wide encoders (csrc/gemm_v4.hip; what gww_gemm_bf16 picks for N % 256 == 0,
 * N <= 3072, K % 128 == 0, M % 256 == 0) with the column split of its work items given explicitly: n_split = 0 is the
 * automatic choice, n_split > 0 must divide N / 256.  Results do not depend on n_split bit for bit (tests). */
int gww_gemm_bf16_v4_split(const void* A, const void* W, const float* bias, const float* resid, void* C,
                           long M, int N, int K, int epilogue, int n_split, void* stream);
/* A-stationary bf16 GEMM for K in {256, 384, 512}, N % 128 == 0 (QKV / fc1 / out_proj at
 * whisper-tiny/base): C = epi(f(A) @ W^T + bias), C bf16, epilogue 0 (bias) or 1 (GELU).
 *   ln_u == NULL : A is bf16 [M,K], W the plain bf16 [N,K] panel.
 *   ln_u != NULL : A is the fp32 residual stream x [M,K].  In ONE pass the kernel forms
 *                  x_new = x + delta (delta bf16 [M,K] or NULL), writes it to x_out (fp32 [M,K],
 *                  may be NULL, must not alias A) and applies LayerNorm (eps 1e-5,
 *                  HF:modeling_whisper.py:392,402) algebraically: W must be the gain-folded
 *                  panel and ln_u / ln_cb the vectors produced by gww_ln_fold_weights; `bias`
 *                  is ignored (it is inside ln_cb).
 * Rows of C must be ALLOCATED up to the next multiple of 256: whole 256-row panels are stored
 * unconditionally (rows >= M are scratch). */
int gww_gemm_astat_bf16(const void* A, const void* delta, float* x_out, const float* ln_u,
                        const float* ln_cb, const void* W, const float* bias, void* C, long M, int N,
                        int K, int epilogue, void* stream);
/* Fold a LayerNorm into the Linear that follows it: w fp32 [N,K], ln_w / ln_b [K], bias [N] or NULL
 *   w_folded[n][k] = bf16(scale * ln_w[k] * w[n][k]),   u[n] = sum_k w_folded[n][k],
 *   cb[n] = scale * (bias[n] + sum_k ln_b[k] * w[n][k]). */
int gww_ln_fold_weights(const float* w, const float* ln_w, const float* ln_b, const float* bias, float scale,
                        int N, int K, void* w_folded_bf16, float* u, float* cb, void* stream);
/* Full-N bf16 GEMM for the long-K, N = d contractions (fc2, conv2): one workgroup owns complete
 * 128-row x N output panels, so A is read from HBM once.  N in {384, 512}, K % 32 == 0, C bf16
 * [M,N] with rows allocated up to the next multiple of 128; epilogue 0 (bias) or 1 (GELU). */
int gww_gemm_fulln_bf16(const void* A, const void* W, const float* bias, void* C, long M, int N, int K,
                        int epilogue, void* stream);
/* Fused MLP block at d_model = 384 (HF:modeling_whisper.py:401-407, residual add deferred to the consumer):
 *   x_out = x + delta;   C = bf16( fc2( gelu( fc1( LayerNorm(x_out) ) ) ) + b2 )
 * x fp32 [M,384], delta bf16 [M,384], x_out fp32 [M,384] (must not alias x); ln_u / ln_cb [F] from
 * gww_ln_fold_weights of fc1; Wt = gww_mlp_pack_bf16 stream; C bf16 with rows allocated up to the next multiple of
 * 128.  F % 128 == 0, F <= 1536.  The [M,F] activation never leaves the CU; GELU is x * sigmoid(odd quintic),
 * |err| <= 2.6e-5 against the erf form.
 * With qkv_out != NULL the NEXT layer's self_attn_layer_norm + q / k / v projection (HF:modeling_whisper.py:392,
 * 303-318) is appended: x_next = x + delta + bf16(C) (the residual stream entering the next layer) is then written
 * BACK OVER x (every 128-row panel has finished reading its rows of x by then; x_out keeps x + delta, the block's
 * intermediate stream -- the kernel's second residual seam must not run in place), C is not written, and qkv_out bf16 [M (rows padded to 128), NQ] = LayerNorm(x_next) Wqkv'^T + cb with
 * qkv_u / qkv_cb from gww_ln_fold_weights of that projection (its panel appended to the stream by
 * gww_mlp_pack_bf16).  NQ % 128 == 0, NQ <= 1536. */
int gww_mlp_fused_bf16(float* x, const void* delta, float* x_out, const float* ln_u, const float* ln_cb,
                       const void* Wt, const float* b2, void* C, long M, int d, int F, const float* qkv_u,
                       const float* qkv_cb, void* qkv_out, int NQ, void* stream);
/* Pre-tile the weights of gww_mlp_fused_bf16: w1_folded bf16 [F,384] (gww_ln_fold_weights), w2 bf16 [384,F],
 * optionally the next layer's folded q / k / v panel bf16 [NQ,384] -> out bf16, 2*384*F (+ NQ*384) elements, as the
 * sequence of swizzled 16-KiB LDS images the kernel streams. */
int gww_mlp_pack_bf16(const void* w1_folded, const void* w2, const void* wqkv_folded_or_null, void* out, int d, int F,
                      int NQ, void* stream);
/* The attention output projection fused IN FRONT of gww_mlp_fused_bf16 (HF:modeling_whisper.py:353-356, 396-398): ctx bf16
 * [M,384] is the attention context, x_out = x + bf16(ctx W_o^T + bo) (the value the stand-alone out_proj + deferred
 * residual add produce), then the block as gww_mlp_fused_bf16 describes, incl. the optional q / k / v tail.  Wt =
 * gww_mlp_pack_op_bf16: the 18 tiles of W_o bf16 [384,384] in front of the gww_mlp_pack_bf16 stream. */
int gww_attn_out_mlp_fused_bf16(float* x, const void* ctx, const float* bo, float* x_out, const float* ln_u,
                                const float* ln_cb, const void* Wt, const float* b2, void* C, long M, int d, int F,
                                const float* qkv_u, const float* qkv_cb, void* qkv_out, int NQ, void* stream);
/* ... and, for the LAST layer, with the encoder's final LayerNorm (HF:modeling_whisper.py:642) as the epilogue:
 * y fp32 [M,384] = LayerNorm(x_mid + bf16(mlp(LayerNorm2(x_mid)) + b2); lnf_w, lnf_b), x_mid = x + bf16(ctx W_o^T + bo)
 * (x_mid fp32 [M,384] is written too; it aliases neither x nor y).  Wt = gww_mlp_pack_op_bf16 without a q / k / v panel. */
int gww_attn_out_mlp_final_bf16(const float* x, const void* ctx, const float* bo, float* x_mid, const float* ln_u,
                                const float* ln_cb, const void* Wt, const float* b2, const float* lnf_w, const float* lnf_b,
                                float* y, long M, int d, int F, void* stream);
int gww_mlp_pack_op_bf16(const void* wo, const void* w1_folded, const void* w2, const void* wqkv_folded_or_null, void* out,
                         int d, int F, int NQ, void* stream);
/* self_attn_layer_norm + q / k / v projection of a residual stream with no pending delta (layer 0, fed by the conv stem;
 * HF:modeling_whisper.py:392, 303-318) at d_model = 384, on the panel prologue and the q / k / v tail of
 * gww_mlp_fused_bf16:  qkv_out bf16 [M (rows padded to 128), NQ] = LayerNorm(x) Wqkv'^T + cb.  x fp32 [M,384] is only read;
 * qkv_u / qkv_cb from gww_ln_fold_weights; Wt = gww_mlp_pack_bf16(NULL, NULL, wqkv_folded, ., 384, 0, NQ).
 * NQ % 128 == 0, NQ <= 1536. */
int gww_lnqkv_fused_bf16(const float* x, const float* qkv_u, const float* qkv_cb, const void* Wt, void* qkv_out, long M,
                         int d, int NQ, void* stream);
/* Q-transform front end #2 (ml4gw QScan as used by MLGWSC-1/train.py:117-122,135-154; PARITY UNPINNED: ml4gw is not
 * vendored, pinned or installed -- the kernels follow oracle/qscan.py).  The host builds the static tiling once
 * (gw_whisper_amd/qscan.py): rows = int [n_rows][6] (plane, ntiles, windowsize, first data index, energy offset,
 * window offset) plane by plane in frequency order; order = row indices sorted by ntiles; class_ranges_host = [first,
 * last) positions in `order` for ntiles 128, 256, 512, 1024, 2048; window = the bisquare windows back to back.
 * fseries: fp32 [B, ld] (re, im) forward-normalised one-sided spectrum with the positive frequencies doubled (one
 * gww_gemm_f32 against the real-DFT matrix).  energy: fp32 [B, e_total] median-normalised tile energies; plane_max:
 * uint [n_planes] (float bits of each plane's largest energy over the whole batch; zeroed by the call). */
int gww_qscan_energy_f32(const float* fseries, int ld, int B, const int* rows, const int* order,
                         const int* class_ranges_host, const float* window, float* energy, long e_total,
                         unsigned int* plane_max, int n_planes, void* stream);
/* Select the plane with the largest energy over the batch (on the device) and resample it to out fp32 [B, F, T]
 * with PyTorch's bicubic rules (time per row, then frequency).  plane_rows: device int [n_planes][2] = first row,
 * row count.  chosen (optional): device int receiving the selected plane. */
int gww_qscan_interp_f32(const float* energy, long e_total, const int* rows, const int* plane_rows, int n_planes,
                         const unsigned int* plane_max, int B, int F, int T, float* out, int* chosen, void* stream);
/* Tail of the reference's QTransformAdapter (MLGWSC-1/train.py:146-153, inference.py:345-350) as ONE kernel:
 * AdaptiveAvgPool2d((F, T)) of the adapter CNN's output y fp32 [B, Hin, Win], then scale * y + bias, then
 * * film_gamma[i] + film_beta[i]; the result goes straight into the stacked [B, D, F, T] feature tensor: element
 * (b, f, t) at out[b * out_batch_stride + f * T + t] (the caller offsets `out` to detector i).  scale / bias / gamma_i /
 * beta_i are device scalars (no host sync).  T % 4 == 0, Win <= 4096. */
int gww_qadapter_tail_f32(const float* y, int B, int Hin, int Win, const float* scale, const float* bias,
                          const float* gamma_i, const float* beta_i, float* out, long out_batch_stride, int F, int T,
                          void* stream);
/* The adapter's small CNN (`self.freq_adapter`: Conv2d(1,c1,3,p1) ReLU MaxPool2d(2) Conv2d(c1,c2,3,p1) ReLU MaxPool2d(2)
 * Conv2d(c2,c3,3,p1) ReLU Conv2d(c3,1,1); MLGWSC-1/train.py:117-122 with c = 32 / 64 / 128, MLGWSC-1/inference.py:320-330
 * with c = 16 / 32 / 64) as three HIP launches: a VALU kernel for the one-channel input layer and an implicit-GEMM MFMA
 * kernel (bf16-pair operands, fp32 accumulation: about 1e-5 relative to the fp32 the reference computes in) with ReLU /
 * max-pool resp. ReLU + the 1 x 1 convolution in its epilogue.
 *   gww_qadapter_cnn_pack_f32: the torch parameters (device fp32: w1 [c1,1,3,3] b1 [c1] w2 [c2,c1,3,3] b2 [c2]
 *     w3 [c3,c2,3,3] b3 [c3] w4 [1,c3,1,1] b4 [1]) -> `packed` (gww_qadapter_cnn_packed_bytes bytes; 0 = channel widths
 *     the reference does not have)
 *   gww_qadapter_cnn_forward_f32: qspec fp32 [B, H, W] (the Q-scan map, H % 32 == 0, W % 128 == 0) -> y fp32
 *     [B, H/4, W/4] (the input of gww_qadapter_tail_f32); workspace: gww_qadapter_cnn_workspace_bytes, caller-owned */
size_t gww_qadapter_cnn_packed_bytes(int c1, int c2, int c3);
size_t gww_qadapter_cnn_workspace_bytes(int B, int H, int W, int c1, int c2);
int gww_qadapter_cnn_pack_f32(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                              const float* b3, const float* w4, const float* b4, int c1, int c2, int c3, void* packed,
                              void* stream);
int gww_qadapter_cnn_forward_f32(const float* qspec, int B, int H, int W, const void* packed, int c1, int c2, int c3,
                                 void* workspace, size_t workspace_bytes, float* y, void* stream);
/* The CNN's BACKWARD (the adapter is trained through the frozen encoder, MLGWSC-1/train.py:494-504; replaces torch
 * autograd through nn.Conv2d / nn.MaxPool2d, i.e. MIOpen): dy fp32 [B, H/4, W/4] = the gradient of
 * gww_qadapter_cnn_forward_f32's y -> the gradients of the eight torch parameters in their torch shapes (fp32, overwritten):
 * dw1 [c1,1,3,3] db1 [c1] dw2 [c2,c1,3,3] db2 [c2] dw3 [c3,c2,3,3] db3 [c3] dw4 [1,c3,1,1] db4 [1].  `packed` is the
 * forward's blob, w2 / w3 the raw fp32 parameters (packed here, transposed, for the data gradients).  Nothing was saved by
 * the forward: the activations are recomputed.  The Q-scan input needs no gradient (the reference computes it under
 * no_grad).  W <= 512. */
size_t gww_qadapter_cnn_backward_workspace_bytes(int B, int H, int W, int c1, int c2, int c3);
int gww_qadapter_cnn_backward_f32(const float* qspec, const float* dy, int B, int H, int W, const void* packed,
                                  const float* w2, const float* w3, int c1, int c2, int c3, void* workspace,
                                  size_t workspace_bytes, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                                  float* db3, float* dw4, float* db4, void* stream);
/* Whitening of the search pipeline's strain (MLGWSC-1/inference.py:56-137 -> PyCBC 2.4.0 TimeSeries.psd / welch /
 * inverse_spectrum_truncation; PARITY UNPINNED -- PyCBC is not installed, the kernels follow oracle/whiten.py).
 * gww_welch_power_f32: |rDFT|^2 * scale of the windowed Welch segments from their (re, im) rows (a gww_gemm_f32 against
 * the windowed real-DFT matrix), DC / Nyquist halved.  gww_column_median_f32: numpy.median over the segments per
 * frequency bin (values >= 0).  gww_fir_f32: the inverse-spectrum-truncated whitening filter applied as a FIR in the
 * time domain, out[d][n] = sum_u g[d][u] xp[d][n + u] (xp circularly padded by the caller, taps zero-padded to a
 * multiple of 4). */
int gww_welch_power_f32(const float* spec, long ld, long n_seg, int n_bins, float scale, float* power, void* stream);
int gww_column_median_f32(const float* power, long n_seg, int n_bins, float* median, void* stream);
int gww_fir_f32(const float* xp, long xp_stride, const float* g, int taps4, int D, float* out, long out_stride,
                long n_out, void* stream);
/* Trigger clustering on the device (MLGWSC-1/inference.py:140-166 `get_clusters` behind :484-487): windows whose score
 * exceeds trigger_threshold, in time order, join the running cluster unless they lie more than cluster_threshold seconds
 * behind the previous trigger; per cluster the time and value of its first maximum.  times fp64 [n] (the reference's
 * stamps), scores fp32 [n]; out_times fp64 / out_vals fp32 [max_clusters], *out_count = clusters found (may exceed
 * max_clusters: only the first max_clusters are stored).  One launch per trigger list (segment). */
int gww_cluster_triggers_f64(const double* times, const float* scores, long n, float trigger_threshold,
                             double cluster_threshold, double* out_times, float* out_vals, int* out_count,
                             int max_clusters, void* stream);
int gww_gemm_f32(const float* A, const float* W, const float* bias, const float* resid, float* C,
                 long M, int N, int K, int epilogue, void* stream);
/* softmax(q k^T) v per head, q pre-scaled; qkv [B,T,3*d] (q|k|v), ctx [B,T,d]; head_dim 64 */
int gww_attention_bf16(const void* qkv, void* ctx, int B, int T, int n_heads, void* stream);
int gww_attention_f32(const float* qkv, float* ctx, int B, int T, int n_heads, void* stream);
/* the kernel of the encoder's bf16 paths (k_attention_dma_bf16): q must be in log2 units, i.e. projected with
 * log2(e) / 8 instead of 1 / 8 (gww_encoder_set_weights folds that into every packed bf16 q panel); the running
 * reference enters the score accumulators through one extra MFMA, K / V tiles by LDS-DMA; lse optional, natural log */
int gww_attention_log2q_bf16(const void* qkv, void* ctx, float* lse_or_null, int B, int T, int n_heads, void* stream);
/* attention backward: dqkv [B,T,3d] from qkv, ctx (forward output), dctx and the forward's lse [B,H,T]
 * (gww_attention_lse_bf16 below); d_scratch: B * H * (T + ceil(T / 64)) fp32 words (row dots + live-tile flags:
 * query tiles whose dctx rows are all zero are skipped, which is most of them under last-token pooling) */
int gww_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                           float* d_scratch, void* dqkv, int B, int T, int n_heads, void* stream);
/* the same for a q section in log2 units (projected with log2(e) / 8, what gww_attention_log2q_bf16 takes): the q
 * section of dqkv is the gradient with respect to that stored q */
int gww_attention_bwd_log2q_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                                 float* d_scratch, void* dqkv, int B, int T, int n_heads, void* stream);
/* forward attention that also returns the row log-sum-exp lse [B,H,T] */
int gww_attention_lse_bf16(const void* qkv, void* ctx, float* lse, int B, int T, int n_heads, void* stream);
/* LayerNorm backward: dx (+)= dLN/dx . dy   (dy fp32 or bf16; optional bf16 copy of the result) */
int gww_layernorm_bwd(const float* x, const float* gamma, const void* dy, int dy_is_f32, float* dx,
                      int accumulate, void* dx_bf16, long M, int d, void* stream);
/* bf16 GELU: out = gelu(z) (dgelu == NULL) or out = dgelu * gelu'(z); n % 8 == 0 */
int gww_gelu_bf16(const void* z, const void* dgelu_or_null, void* out, long n, void* stream);
/* DoRA parameter gradients of one [d,d] projection (see train_ops.hip) */
int gww_dora_grads(const void* X, long ldx, const void* dY, const void* Y, long ldy, const float* bias_st,
                   float yscale, float scaling, const float* A, const float* B, const float* mag,
                   const float* nrm, float* dA, float* dB, float* dm, long M, int d, int r, void* stream);
/* The same for np (1..3) projections that read the SAME X -- q, k and v of a layer (peft 0.12.0 dora.py; targets of
 * Signal_vs_Noise/src/train.py:230-237) -- in one pass on the matrix cores (dora_grads.hip); d in {384, 512}, r = 8.
 * Projection p reads dY / Y at column col_off[p] of rows with stride ldy; all arrays have np entries (host memory,
 * device pointers inside).  Gradients are accumulated. */
int gww_dora_grads_multi(const void* X, long ldx, const void* dY, const void* Y, long ldy, int np,
                         const long* col_off, const float* const* bias_st, const float* yscale,
                         const float* scaling, const float* const* A, const float* const* B,
                         const float* const* mag, const float* const* nrm, float* const* dA, float* const* dB,
                         float* const* dm, long M, int d, void* stream);
/* fp32 -> bf16 (round to nearest even), n elements */
int gww_cast_f32_bf16(const float* x, void* y, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GWW_H */
