// Encoder handle: weight packing + the forward launch sequence (host side of the
// C ABI).  Replaces WhisperEncoder.forward (HF:modeling_whisper.py:592-646):
//   conv1+GELU -> conv2(stride 2)+GELU + pos -> L x [LN, QKV, MHSA, out_proj+res,
//   LN, fc1+GELU, fc2+res] -> final LN.
//
// HBM layout of one forward (batch B, T_in = 3000 frames, T = 1500 tokens):
//   melT  [B, T_in+2, 80]   token-major mel, zero rows = Conv1d padding   (+tail pad)
//   c1    [B, T_in+2, d]    conv1 output, token-major, zero rows = padding (+1 row)
//   x     [B*T, d]   fp32   residual stream (never leaves fp32)
//   h     [B*T, d]          LayerNorm output (GEMM A operand)
//   qkv   [B*T, 3d]         q | k | v, q pre-scaled by 1/8
//   ctx   [B*T, d]          attention context
//   f1    [B*T, ffn]        fc1 + GELU
// In GWW_PREC_BF16 the activation tensors other than x are bf16; in GWW_PREC_F32
// everything is fp32.  All of it lives in the caller's workspace; the handle owns
// only the packed weights.
#include "common.h"
#include "epilogue.h"

#include <stdlib.h>
#include <vector>

namespace gww {
#ifdef GWW_LAB
long lab_int(const char* name, long dflt) {   // the laboratory build's only environment reader (common.h)
  const char* e = getenv(name);
  return e ? atol(e) : dflt;
}
#endif

int launch_transpose_bf16(const void* in, void* out, int R, int Cn, hipStream_t s);
int launch_ln_bwd(const float* x, const float* gamma, const void* dy, int dy_f32, float* dx, int accumulate,
                  void* dx_bf16, long M, int d, hipStream_t s);
int launch_gelu_bf16(const void* z, const void* df, void* out, long n, hipStream_t s);
int launch_sub_f32_bf16(const float* a, const float* b, void* out, long n, hipStream_t s);
int launch_stem_dz2(const void* dxb, const void* z2, void* out, int B, int T, int d, hipStream_t s);
int launch_stem_dz1(const void* col, const void* z1, void* out, int B, int T, int Tin, int d, hipStream_t s);
int launch_stem_dmel(const void* col1, float* dmel, int B, int Tin, int C, int Kp, hipStream_t s);
int launch_mlp_pack(const void* w1_folded, const void* w2, const void* wqkv_folded, void* out, int d, int F, int NQ,
                    hipStream_t s, const void* wo = nullptr);
int launch_mlp_fused(const float* x, const void* delta, float* x_out, const float* ln_u, const float* ln_cb,
                     const void* Wt, const float* b2, void* C, long M, int d, int F, hipStream_t s,
                     const float* q_u = nullptr, const float* q_cb = nullptr, void* q_out = nullptr, int NQ = 0,
                     float* x_next_out = nullptr, const float* bo = nullptr, bool keep_x_new = true);
int launch_mlp_fused_final(const float* x, const void* ctx, float* x_mid, const float* ln_u, const float* ln_cb,
                           const void* Wt, const float* b2, const float* bo, const float* lnf_w, const float* lnf_b, float* y,
                           long M, int d, int F, hipStream_t s, bool keep_x_new = true);
int launch_lnqkv_fused(const float* x, const float* q_u, const float* q_cb, const void* Wt, void* q_out, long M, int d,
                       int NQ, hipStream_t s);
int launch_add_delta_f32(const float* x, const void* delta_bf16, float* out, long n, hipStream_t s);
int launch_dora_grads(const void* X, long ldx, const void* dY, const void* Y, long ldy, const float* bias_st,
                      float yscale, float scaling, const float* A, const float* Bm, const float* mag,
                      const float* nrm, float* dA, float* dB, float* dm, long M, int d, int r, hipStream_t s,
                      void* scratch = nullptr, size_t scratch_bytes = 0);
int launch_dora_grads_multi(const void* X, long ldx, const void* dY, const void* Y, long ldy, int np,
                            const long* col_off, const float* const* bias_st, const float* yscale,
                            const float* scaling, const float* const* A, const float* const* Bm,
                            const float* const* mag, const float* const* nrm, float* const* dA, float* const* dB,
                            float* const* dm, long M, int d, hipStream_t s, void* scratch, size_t scratch_bytes);
size_t dora_grads_scratch_bytes(int np, int d);
int launch_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* D,
                              void* dqkv, int B, int T, int H, hipStream_t s, bool q_log2);
int launch_mel_to_tokens(const float* mel, void* out, int out_bf16, int B, int C, int T, hipStream_t s);
}

using namespace gww;

namespace {

struct LayerW {
  // bf16 panels
  unsigned short *wqkv, *wo, *w1, *w2;
  // fp32 panels (parity path)
  float *wqkv32, *wo32, *w132, *w232;
  float *bqkv, *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;
  float* bqkv16;   // q | k | v bias of the bf16 panels: the q part carries log2(e) like the packed bf16 q weights
  // LayerNorm-folded panels for the A-stationary GEMMs (gain folded into W, see gemm_astat.hip)
  unsigned short *wqkv_ln, *w1_ln;
  unsigned short* wmlp;   // fused-MLP weight stream (d = 384): mlp_fused.hip
  unsigned short* wqkv_st; // the folded q / k / v panel alone as a tile stream (layer 0: launch_lnqkv_fused)
  unsigned short* wmlp_op; // the fused-MLP stream with the W_o tiles in front (inference: out_proj fused into the block)
  float *uqkv, *cbqkv, *u1, *cb1;
  // transposed bf16 panels [K][N] for the dX GEMMs of the training backward
  unsigned short *wqkvT, *woT, *w1T, *w2T;
};

constexpr int kConv1Kpad = 256;   // 3 * 80 = 240 padded to a multiple of 64

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace

// kernel classes of one forward, for the optional per-kernel event trace (bench.py roofline)
enum : int { TR_MEL = 0, TR_CONV1, TR_CONV2, TR_QKV, TR_ATTN, TR_OUT, TR_FC1, TR_FC2, TR_LN, TR_MLP, TR_MLPQKV, TR_LNROWS, TR_MLPFIN, TR_COUNT };

struct TraceSpan { int cls; hipEvent_t a, b; };

struct gww_encoder {
  gww_enc_cfg cfg{};
  bool ready = false;
  bool trace = false;
  std::vector<TraceSpan> spans;     // recorded since the last read
  std::vector<hipEvent_t> pool;     // reusable events
  // dual-stream split: two half batches on two library-owned streams, so HBM-bound kernels of one
  // half overlap MFMA-bound kernels of the other on different CUs
  int split = 0;                    // 0: off, 1: on for batch >= 2 * kSplitMin
  hipStream_t s2[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_skew = nullptr, ev_join[2] = {nullptr, nullptr};
  char* blob = nullptr;
  size_t blob_bytes = 0;
  unsigned short *c1w = nullptr, *c2w = nullptr;
  unsigned short *c1wT = nullptr, *c2wT = nullptr;   // [Kpad, d] / [3 d, d]: input-gradient GEMMs of the stem
  float *c1w32 = nullptr, *c2w32 = nullptr;
  float *c1b = nullptr, *c2b = nullptr, *pos = nullptr, *lnw = nullptr, *lnb = nullptr;
  std::vector<LayerW> layers;
};

extern "C" int gww_encoder_create(const gww_enc_cfg* cfg, gww_encoder** out) {
  GWW_REQUIRE(cfg && out, "gww_encoder_create: NULL argument");
  const int d = cfg->d_model, L = cfg->n_layers, H = cfg->n_heads, F = cfg->ffn, C = cfg->n_mels;
  GWW_REQUIRE(d > 0 && d % 128 == 0 && d <= 1280, "gww_encoder_create: d_model=%d must be a multiple of 128 <= 1280", d);
  GWW_REQUIRE(H * 64 == d, "gww_encoder_create: n_heads=%d * 64 != d_model=%d (Whisper head_dim is 64)", H, d);
  GWW_REQUIRE(L > 0 && F > 0 && F % 64 == 0, "gww_encoder_create: bad n_layers=%d / ffn=%d", L, F);
  GWW_REQUIRE(C == 80, "gww_encoder_create: n_mels=%d (only 80 is supported)", C);
  GWW_REQUIRE(cfg->t_in > 0 && cfg->t_in % 2 == 0, "gww_encoder_create: t_in=%d must be even", cfg->t_in);
  const int T = cfg->t_in / 2;

  gww_encoder* e = new gww_encoder();
  e->cfg = *cfg;
  // carve one allocation
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  const size_t o_c1w = take((size_t)d * kConv1Kpad * 2), o_c2w = take((size_t)((d + 255) / 256 * 256) * 3 * d * 2);   // rows d .. : zero padding (k_gemm_bf16_v4 takes N % 256 == 0)
  const size_t o_c1w32 = take((size_t)d * kConv1Kpad * 4), o_c2w32 = take((size_t)d * 3 * d * 4);
  const size_t o_c1wT = take((size_t)d * kConv1Kpad * 2), o_c2wT = take((size_t)d * 3 * d * 2);
  const size_t o_c1b = take(d * 4), o_c2b = take(d * 4), o_pos = take((size_t)T * d * 4);
  const size_t o_lnw = take(d * 4), o_lnb = take(d * 4);
  struct LO { size_t wqkv, wo, w1, w2, wqkv32, wo32, w132, w232, bqkv, bqkv16, bo, b1, b2, ln1w, ln1b, ln2w, ln2b,
                     wqkv_ln, w1_ln, uqkv, cbqkv, u1, cb1, wqkvT, woT, w1T, w2T, wmlp, wqkv_st, wmlp_op; };
  std::vector<LO> lo(L);
  for (int i = 0; i < L; ++i) {
    lo[i].wqkv = take((size_t)3 * d * d * 2);
    lo[i].wo = take((size_t)d * d * 2);
    lo[i].w1 = take((size_t)F * d * 2);
    lo[i].w2 = take((size_t)d * F * 2);
    lo[i].wqkv32 = take((size_t)3 * d * d * 4);
    lo[i].wo32 = take((size_t)d * d * 4);
    lo[i].w132 = take((size_t)F * d * 4);
    lo[i].w232 = take((size_t)d * F * 4);
    lo[i].bqkv = take(3 * d * 4);
    lo[i].bqkv16 = take(3 * d * 4);
    lo[i].bo = take(d * 4);
    lo[i].b1 = take(F * 4);
    lo[i].b2 = take(d * 4);
    lo[i].ln1w = take(d * 4);
    lo[i].ln1b = take(d * 4);
    lo[i].ln2w = take(d * 4);
    lo[i].ln2b = take(d * 4);
    lo[i].wqkv_ln = take((size_t)3 * d * d * 2);
    lo[i].w1_ln = take((size_t)F * d * 2);
    lo[i].wmlp = take(((size_t)2 * F * d + (size_t)3 * d * d) * 2);   // fc1' + fc2 (+ the next layer's q / k / v panel)
    lo[i].wqkv_st = take(i == 0 ? (size_t)3 * d * d * 2 : 16);        // layer 0's own panel as a stream
    lo[i].wmlp_op = take(((size_t)d * d + (size_t)2 * F * d + (size_t)3 * d * d) * 2);   // W_o + the stream above
    lo[i].uqkv = take(3 * d * 4);
    lo[i].cbqkv = take(3 * d * 4);
    lo[i].u1 = take(F * 4);
    lo[i].cb1 = take(F * 4);
    lo[i].wqkvT = take((size_t)3 * d * d * 2);
    lo[i].woT = take((size_t)d * d * 2);
    lo[i].w1T = take((size_t)F * d * 2);
    lo[i].w2T = take((size_t)d * F * 2);
  }
  hipError_t err = hipMalloc(&e->blob, off);
  if (err != hipSuccess) {
    delete e;
    return fail(GWW_ERR_HIP, "hipMalloc(%zu bytes of packed weights) failed: %s", off, hipGetErrorString(err));
  }
  e->blob_bytes = off;
  char* p = e->blob;
  e->c1w = (unsigned short*)(p + o_c1w);
  e->c2w = (unsigned short*)(p + o_c2w);
  e->c1wT = (unsigned short*)(p + o_c1wT);
  e->c2wT = (unsigned short*)(p + o_c2wT);
  e->c1w32 = (float*)(p + o_c1w32);
  e->c2w32 = (float*)(p + o_c2w32);
  e->c1b = (float*)(p + o_c1b);
  e->c2b = (float*)(p + o_c2b);
  e->pos = (float*)(p + o_pos);
  e->lnw = (float*)(p + o_lnw);
  e->lnb = (float*)(p + o_lnb);
  e->layers.resize(L);
  for (int i = 0; i < L; ++i) {
    LayerW& w = e->layers[i];
    w.wqkv = (unsigned short*)(p + lo[i].wqkv);
    w.wo = (unsigned short*)(p + lo[i].wo);
    w.w1 = (unsigned short*)(p + lo[i].w1);
    w.w2 = (unsigned short*)(p + lo[i].w2);
    w.wqkv32 = (float*)(p + lo[i].wqkv32);
    w.wo32 = (float*)(p + lo[i].wo32);
    w.w132 = (float*)(p + lo[i].w132);
    w.w232 = (float*)(p + lo[i].w232);
    w.bqkv = (float*)(p + lo[i].bqkv);
    w.bqkv16 = (float*)(p + lo[i].bqkv16);
    w.bo = (float*)(p + lo[i].bo);
    w.b1 = (float*)(p + lo[i].b1);
    w.b2 = (float*)(p + lo[i].b2);
    w.ln1w = (float*)(p + lo[i].ln1w);
    w.ln1b = (float*)(p + lo[i].ln1b);
    w.ln2w = (float*)(p + lo[i].ln2w);
    w.ln2b = (float*)(p + lo[i].ln2b);
    w.wqkv_ln = (unsigned short*)(p + lo[i].wqkv_ln);
    w.w1_ln = (unsigned short*)(p + lo[i].w1_ln);
    w.wmlp = (unsigned short*)(p + lo[i].wmlp);
    w.wqkv_st = (unsigned short*)(p + lo[i].wqkv_st);
    w.wmlp_op = (unsigned short*)(p + lo[i].wmlp_op);
    w.uqkv = (float*)(p + lo[i].uqkv);
    w.cbqkv = (float*)(p + lo[i].cbqkv);
    w.u1 = (float*)(p + lo[i].u1);
    w.cb1 = (float*)(p + lo[i].cb1);
    w.wqkvT = (unsigned short*)(p + lo[i].wqkvT);
    w.woT = (unsigned short*)(p + lo[i].woT);
    w.w1T = (unsigned short*)(p + lo[i].w1T);
    w.w2T = (unsigned short*)(p + lo[i].w2T);
  }
  *out = e;
  return GWW_OK;
}

extern "C" int gww_encoder_trace_enable(gww_encoder* e, int on) {
  GWW_REQUIRE(e != nullptr, "gww_encoder_trace_enable: NULL handle");
  e->trace = on != 0;
  return GWW_OK;
}

// Sum the elapsed time (ms) and the launch count per kernel class since the last read; blocks until
// the recorded events have completed.  ms / counts: arrays of gww_encoder_trace_classes() entries.
extern "C" int gww_encoder_trace_read(gww_encoder* e, float* ms, int* counts) {
  GWW_REQUIRE(e && ms && counts, "gww_encoder_trace_read: NULL argument");
  for (int i = 0; i < TR_COUNT; ++i) { ms[i] = 0.f; counts[i] = 0; }
  for (TraceSpan& sp : e->spans) {
    GWW_HIP(hipEventSynchronize(sp.b));
    float t = 0.f;
    GWW_HIP(hipEventElapsedTime(&t, sp.a, sp.b));
    ms[sp.cls] += t;
    counts[sp.cls] += 1;
    e->pool.push_back(sp.a);
    e->pool.push_back(sp.b);
  }
  e->spans.clear();
  return GWW_OK;
}

extern "C" int gww_encoder_trace_classes(void) { return TR_COUNT; }
extern "C" const char* gww_encoder_trace_class_name(int i) {
  static const char* names[TR_COUNT] = {"mel_to_tokens", "conv1_gelu", "conv2_gelu_pos", "ln+qkv_proj", "attention",
                                        "out_proj", "ln+fc1_gelu", "fc2", "final_layernorm", "mlp_fused(ln+fc1+gelu+fc2)",
                                        "mlp_fused+next_ln_qkv", "layernorm_rows(B pooled rows)", "mlp_fused+final_layernorm"};
  return (i >= 0 && i < TR_COUNT) ? names[i] : "?";
}

extern "C" void gww_encoder_destroy(gww_encoder* e) {
  if (!e) return;
  for (TraceSpan& sp : e->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (hipEvent_t ev : e->pool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 2; ++i) {
    if (e->s2[i]) (void)hipStreamDestroy(e->s2[i]);
    if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
  }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->blob) (void)hipFree(e->blob);
  delete e;
}

// Weight groups of one layer (bit mask of gww_encoder_update_weights): what has to be re-packed when a
// parameter of the group changed.
//   1 = q / k / v projections + their biases + self_attn_layer_norm   (QKV panels, LN-folded panel, transposes)
//   2 = out_proj          4 = fc1 + final_layer_norm          8 = fc2
static int pack_weights(gww_encoder* e, const gww_enc_globals* g, const gww_enc_layer* layers, int n_layers,
                        const unsigned* dirty, hipStream_t s) {
  const int d = e->cfg.d_model, F = e->cfg.ffn, C = e->cfg.n_mels, T = e->cfg.t_in / 2;
  const float qs = 0.125f;   // head_dim^-0.5 = 64^-0.5, exact power of two (HF:modeling_whisper.py:309)
  // Two dependency phases, one k_prep_batch launch each (per 40 ops): `pb` reads only the caller's fp32 tensors, `pt` (the
  // transposes for the backward's dX GEMMs) reads panels `pb` wrote.  A DoRA step on whisper-tiny used to enqueue ~75 launches
  // of 3 - 6 us here.
  PrepBatch pb(s), pt(s);
  auto pack = [&](const float* w, unsigned short* o16, float* o32, int N, int Cin, int taps, int Kpad,
                  float scale) -> int {
    GWW_TRY(pb.pack(w, o16, 1, N, Cin, taps, Kpad, scale));
    GWW_TRY(pb.pack(w, o32, 0, N, Cin, taps, Kpad, scale));
    return GWW_OK;
  };
  if (g) {
    GWW_REQUIRE(g->conv1_w && g->conv1_b && g->conv2_w && g->conv2_b && g->pos && g->ln_w && g->ln_b,
                "gww_encoder_set_weights: NULL global weight");
    GWW_TRY(pack(g->conv1_w, e->c1w, e->c1w32, d, C, 3, kConv1Kpad, 1.f));
    GWW_TRY(pack(g->conv2_w, e->c2w, e->c2w32, d, d, 3, 3 * d, 1.f));
    if (d % 256 != 0) GWW_HIP(hipMemsetAsync(e->c2w + (size_t)d * 3 * d, 0, (size_t)((d + 255) / 256 * 256 - d) * 3 * d * 2, s));
    GWW_TRY(pt.transpose(e->c1w, e->c1wT, d, kConv1Kpad));
    GWW_TRY(pt.transpose(e->c2w, e->c2wT, d, 3 * d));
    GWW_TRY(pb.copy(g->conv1_b, e->c1b, d, 1.f));
    GWW_TRY(pb.copy(g->conv2_b, e->c2b, d, 1.f));
    GWW_HIP(hipMemcpyAsync(e->pos, g->pos, (size_t)T * d * 4, hipMemcpyDeviceToDevice, s));
    GWW_TRY(pb.copy(g->ln_w, e->lnw, d, 1.f));
    GWW_TRY(pb.copy(g->ln_b, e->lnb, d, 1.f));
  }
  for (int i = 0; i < n_layers; ++i) {
    const unsigned m = dirty ? dirty[i] : 15u;
    if (!m) continue;
    const gww_enc_layer& L = layers[i];
    LayerW& w = e->layers[i];
    const size_t dd = (size_t)d * d;
    if (m & 1u) {
      GWW_REQUIRE(L.ln1_w && L.ln1_b && L.q_w && L.q_b && L.k_w && L.v_w && L.v_b,
                  "gww_encoder_set_weights: NULL attention weight in layer %d", i);
      // q in LOG2 UNITS on every bf16 panel (forward kernels of attention.hip: p = exp2(s) with no multiply per score;
      // attention_bwd.hip and the DoRA-gradient scale of q follow suit): log2(e) / 8 instead of 1 / 8.  The fp32 parity
      // panels keep natural units.
      const float qs16 = attention_log2q_enabled() ? qs * 1.44269504088896340736f : qs;
      GWW_TRY(pb.pack(L.q_w, w.wqkv, 1, d, d, 1, d, qs16));
      GWW_TRY(pb.pack(L.q_w, w.wqkv32, 0, d, d, 1, d, qs));
      GWW_TRY(pack(L.k_w, w.wqkv + dd, w.wqkv32 + dd, d, d, 1, d, 1.f));
      GWW_TRY(pack(L.v_w, w.wqkv + 2 * dd, w.wqkv32 + 2 * dd, d, d, 1, d, 1.f));
      GWW_TRY(pb.copy(L.q_b, w.bqkv, d, qs));
      GWW_TRY(pb.copy(nullptr, w.bqkv + d, d, 0.f));   // k_proj has no bias
      GWW_TRY(pb.copy(L.v_b, w.bqkv + 2 * d, d, 1.f));
      GWW_TRY(pb.copy(L.q_b, w.bqkv16, d, qs16));
      GWW_TRY(pb.copy(nullptr, w.bqkv16 + d, d, 0.f));
      GWW_TRY(pb.copy(L.v_b, w.bqkv16 + 2 * d, d, 1.f));
      GWW_TRY(pb.copy(L.ln1_w, w.ln1w, d, 1.f));
      GWW_TRY(pb.copy(L.ln1_b, w.ln1b, d, 1.f));
      // gain-folded panel + correction vectors for the algebraic LayerNorm of the A-stationary GEMM
      // (the A-stationary inference path feeds k_attention_l2_bf16, which takes q in log2 units: log2(e) rides in
      // the q panel, one rounding of the fp32 product instead of a second one on bf16 q)
      GWW_TRY(pb.ln_fold(L.q_w, L.ln1_w, L.ln1_b, L.q_b, qs16, d, d, w.wqkv_ln, w.uqkv, w.cbqkv));
      GWW_TRY(pb.ln_fold(L.k_w, L.ln1_w, L.ln1_b, nullptr, 1.f, d, d, w.wqkv_ln + dd, w.uqkv + d, w.cbqkv + d));
      GWW_TRY(pb.ln_fold(L.v_w, L.ln1_w, L.ln1_b, L.v_b, 1.f, d, d, w.wqkv_ln + 2 * dd, w.uqkv + 2 * d,
                             w.cbqkv + 2 * d));
      GWW_TRY(pt.transpose(w.wqkv, w.wqkvT, 3 * d, d));   // [N][K] -> [K][N] for the backward dX GEMM
    }
    if (m & 2u) {
      GWW_REQUIRE(L.o_w && L.o_b, "gww_encoder_set_weights: NULL out_proj weight in layer %d", i);
      GWW_TRY(pack(L.o_w, w.wo, w.wo32, d, d, 1, d, 1.f));
      GWW_TRY(pb.copy(L.o_b, w.bo, d, 1.f));
      GWW_TRY(pt.transpose(w.wo, w.woT, d, d));
    }
    if (m & 4u) {
      GWW_REQUIRE(L.ln2_w && L.ln2_b && L.fc1_w && L.fc1_b, "gww_encoder_set_weights: NULL fc1 weight in layer %d", i);
      GWW_TRY(pack(L.fc1_w, w.w1, w.w132, F, d, 1, d, 1.f));
      GWW_TRY(pb.copy(L.fc1_b, w.b1, F, 1.f));
      GWW_TRY(pb.copy(L.ln2_w, w.ln2w, d, 1.f));
      GWW_TRY(pb.copy(L.ln2_b, w.ln2b, d, 1.f));
      GWW_TRY(pb.ln_fold(L.fc1_w, L.ln2_w, L.ln2_b, L.fc1_b, 1.f, F, d, w.w1_ln, w.u1, w.cb1));
      GWW_TRY(pt.transpose(w.w1, w.w1T, F, d));
    }
    if (m & 8u) {
      GWW_REQUIRE(L.fc2_w && L.fc2_b, "gww_encoder_set_weights: NULL fc2 weight in layer %d", i);
      GWW_TRY(pack(L.fc2_w, w.w2, w.w232, d, F, 1, F, 1.f));
      GWW_TRY(pb.copy(L.fc2_b, w.b2, d, 1.f));
      GWW_TRY(pt.transpose(w.w2, w.w2T, d, F));
    }
  }
  GWW_TRY(pb.flush());
  GWW_TRY(pt.flush());
  // the fused-MLP weight stream of layer i: folded fc1 panel, fc2 and the folded q / k / v panel of layer i + 1
  if (d == 384 && F % 128 == 0 && F <= 1536) {
    for (int i = 0; i < n_layers; ++i) {
      const unsigned m = dirty ? dirty[i] : 15u, mn = i + 1 < n_layers ? (dirty ? dirty[i + 1] : 15u) : 0u;
      if (!(m & 14u) && !(mn & 1u)) continue;   // (bit 1: out_proj, in front of the wmlp_op stream)
      LayerW& w = e->layers[i];
      const void* wq_next = i + 1 < n_layers ? e->layers[i + 1].wqkv_ln : nullptr;
      // Only the stream the active path consumes is packed: the inference and the training forward both run the block with
      // out_proj in front (wmlp_op); the stream without it serves the debug paths GWW_GENERIC_PATH bits 4 / 7 alone (a
      // DoRA step used to pay two full fc1 + fc2 + q/k/v stream packs per layer, 2 x 3.2 MB of writes, for one consumer).
      static const bool plain_stream = (lab_int("GWW_GENERIC_PATH", 0) & (16 | 128)) != 0;
      if (plain_stream && ((m & 12u) || (mn & 1u))) GWW_TRY(launch_mlp_pack(w.w1_ln, w.w2, wq_next, w.wmlp, d, F, 3 * d, s));
      GWW_TRY(launch_mlp_pack(w.w1_ln, w.w2, wq_next, w.wmlp_op, d, F, 3 * d, s, w.wo));
    }
    if (!dirty || (dirty[0] & 1u))   // layer 0's folded q / k / v panel alone (no MLP in front of it)
      GWW_TRY(launch_mlp_pack(nullptr, nullptr, e->layers[0].wqkv_ln, e->layers[0].wqkv_st, d, 0, 3 * d, s));
  }
  return GWW_OK;
}

extern "C" int gww_encoder_set_weights(gww_encoder* e, const gww_enc_globals* g, const gww_enc_layer* layers,
                                       int n_layers, void* stream) {
  GWW_REQUIRE(e && g && layers, "gww_encoder_set_weights: NULL argument");
  GWW_REQUIRE(n_layers == e->cfg.n_layers, "gww_encoder_set_weights: got %d layers, handle has %d", n_layers,
              e->cfg.n_layers);
  GWW_TRY(pack_weights(e, g, layers, n_layers, nullptr, (hipStream_t)stream));
  e->ready = true;
  return GWW_OK;
}

// Re-pack only what changed since the last (full) gww_encoder_set_weights: `globals` may be NULL (stem, positions,
// final LayerNorm unchanged); layer_dirty[i] is the group mask above (0 = layer untouched; pointers of clean
// groups are not read).  A DoRA step touches group 1 (and 2) only: 15 small kernels per layer instead of 40.
extern "C" int gww_encoder_update_weights(gww_encoder* e, const gww_enc_globals* globals_or_null,
                                          const gww_enc_layer* layers, int n_layers, const unsigned* layer_dirty,
                                          void* stream) {
  GWW_REQUIRE(e && layers && layer_dirty, "gww_encoder_update_weights: NULL argument");
  GWW_REQUIRE(n_layers == e->cfg.n_layers, "gww_encoder_update_weights: got %d layers, handle has %d", n_layers,
              e->cfg.n_layers);
  if (!e->ready) return fail(GWW_ERR_STATE, "gww_encoder_update_weights: call gww_encoder_set_weights first");
  return pack_weights(e, globals_or_null, layers, n_layers, layer_dirty, (hipStream_t)stream);
}

namespace {
struct WsLayout {
  size_t melT, c1, x, x2, h, d2, qkv, ctx, f1, total;
};
WsLayout ws_layout(const gww_enc_cfg& c, int B, int precision) {
  const size_t es = precision == GWW_PREC_BF16 ? 2 : 4;
  const size_t d = c.d_model, F = c.ffn, Tin = c.t_in, T = c.t_in / 2, C = c.n_mels;
  WsLayout w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  // row-indexed activations are padded so the large-M GEMM can store whole 256-row panels
  // unconditionally (rows past B*T are scratch); +512 covers conv2's remapped garbage rows
  const size_t Mp = ((size_t)B * T + 255) / 256 * 256 + 512;
  w.melT = take(((size_t)B * (Tin + 2) * C + kConv1Kpad) * es);
  w.c1 = take((((size_t)B * (Tin + 2) + 255) / 256 * 256 + 520) * d * es);   // (+ 520: conv2's 256-row panels read 2 * 255 + 3 rows past the last segment)
  w.x = take(Mp * d * 4);
  w.x2 = take(Mp * d * 4);      // ping-pong partner of x for the fused residual-add prologue
  w.h = take(Mp * d * es);      // LayerNorm output, or out_proj delta on the A-stationary path
  w.d2 = take(Mp * d * es);     // fc2 delta on the A-stationary path
  w.qkv = take(Mp * 3 * d * es);
  w.ctx = take(Mp * d * es);
  w.f1 = take(Mp * F * es);
  w.total = off;
  return w;
}
}  // namespace

constexpr int kSplitMin = 32;   // segments per half below which splitting does not pay

static bool use_split(const gww_encoder* e, int batch) { return e->split && batch >= 2 * kSplitMin; }

extern "C" size_t gww_encoder_workspace_bytes(const gww_encoder* e, int batch, int precision) {
  if (!e || batch <= 0) return 0;
  if (use_split(e, batch)) {
    const int b0 = batch / 2;
    return ws_layout(e->cfg, b0, precision).total + ws_layout(e->cfg, batch - b0, precision).total;
  }
  return ws_layout(e->cfg, batch, precision).total;
}

extern "C" int gww_encoder_set_split(gww_encoder* e, int on) {
  GWW_REQUIRE(e != nullptr, "gww_encoder_set_split: NULL handle");
  if (on && !e->s2[0]) {
    for (int i = 0; i < 2; ++i) {
      GWW_HIP(hipStreamCreateWithFlags(&e->s2[i], hipStreamNonBlocking));
      GWW_HIP(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
    }
    GWW_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    GWW_HIP(hipEventCreateWithFlags(&e->ev_skew, hipEventDisableTiming));
  }
  e->split = on ? 1 : 0;
  return GWW_OK;
}

static int forward_impl(gww_encoder* e, const float* mel, int batch, int precision, void* workspace,
                        size_t workspace_bytes, float* last_hidden, float* last_token, void* stream,
                        hipEvent_t skew_event = nullptr) {
  GWW_REQUIRE(e && mel, "gww_encoder_forward: NULL argument");
  if (!e->ready) return fail(GWW_ERR_STATE, "gww_encoder_forward: weights not set");
  GWW_REQUIRE(precision == GWW_PREC_BF16 || precision == GWW_PREC_F32, "gww_encoder_forward: bad precision %d",
              precision);
  GWW_REQUIRE(batch >= 0, "gww_encoder_forward: batch < 0");
  GWW_REQUIRE(last_hidden || last_token, "gww_encoder_forward: no output requested");
  if (batch == 0) return GWW_OK;
  GWW_REQUIRE((((uintptr_t)mel) & 15) == 0 && (((uintptr_t)workspace) & 255) == 0,
              "gww_encoder_forward: mel must be 16-byte and workspace 256-byte aligned");
  const WsLayout w = ws_layout(e->cfg, batch, precision);
  if (!workspace || workspace_bytes < w.total)
    return fail(GWW_ERR_WORKSPACE, "gww_encoder_forward: workspace %zu bytes < required %zu", workspace_bytes,
                w.total);
  hipStream_t s = (hipStream_t)stream;
  const bool bf = precision == GWW_PREC_BF16;
  const size_t es = bf ? 2 : 4;
  const int d = e->cfg.d_model, F = e->cfg.ffn, Tin = e->cfg.t_in, T = Tin / 2, C = e->cfg.n_mels, H = e->cfg.n_heads;
  const int B = batch;
  char* base = (char*)workspace;
  void* melT = base + w.melT;
  void* c1 = base + w.c1;
  float* x = (float*)(base + w.x);
  void* h = base + w.h;
  void* qkv = base + w.qkv;
  void* ctx = base + w.ctx;
  void* f1 = base + w.f1;
  const long M = (long)B * T;

  // optional per-kernel event trace: TR(cls, launch-expression)
  auto tr_begin = [&](int cls) -> int {
    if (!e->trace) return GWW_OK;
    TraceSpan sp{cls, nullptr, nullptr};
    for (hipEvent_t* ev : {&sp.a, &sp.b}) {
      if (!e->pool.empty()) { *ev = e->pool.back(); e->pool.pop_back(); }
      else GWW_HIP(hipEventCreate(ev));
    }
    GWW_HIP(hipEventRecord(sp.a, s));
    e->spans.push_back(sp);
    return GWW_OK;
  };
  auto tr_end = [&]() -> int {
    if (!e->trace) return GWW_OK;
    GWW_HIP(hipEventRecord(e->spans.back().b, s));
    return GWW_OK;
  };
#define TR(cls, expr)        \
  do {                       \
    GWW_TRY(tr_begin(cls));  \
    GWW_TRY(expr);           \
    GWW_TRY(tr_end());       \
  } while (0)

  // generic GEMM dispatch on precision
  auto gemm = [&](const void* A, long lda, const void* W16, const float* W32, const float* bias,
                  const float* resid, const float* pos, void* Cout, long Mr, int N, int K, int epi,
                  int rpb) -> int {
    return bf ? launch_gemm_bf16(A, lda, W16, bias, resid, pos, Cout, Mr, N, K, epi, rpb, s, /*rows_padded_256=*/1)
              : launch_gemm_f32((const float*)A, lda, W32, bias, resid, pos, (float*)Cout, Mr, N, K, epi, rpb, s);
  };

  // ---- stem
  // bf16, the widths of whisper-tiny / -base / -small / -medium: conv1 reads the [B, 80, T] features itself (conv1_mel.hip:
  // no token-major copy of the input, W-stationary, one launch); otherwise the transposition kernel + a GEMM over its rows
  static const int generic_mask = (int)lab_int("GWW_GENERIC_PATH", 0);   // (0 in the product build)
  const bool conv1_direct = bf && conv1_mel_supported(C, d, kConv1Kpad) && !(generic_mask & 2);
  if (!conv1_direct) {
    TR(TR_MEL, launch_mel_to_tokens(mel, melT, bf ? 1 : 0, B, C, Tin, s));
    GWW_HIP(hipMemsetAsync((char*)melT + (size_t)B * (Tin + 2) * C * es, 0, kConv1Kpad * es, s));
    GWW_HIP(hipMemsetAsync(c1, 0, (size_t)d * es, s));   // zero row 0 of batch 0 (token -1)
  }
  // A-stationary kernels (A panel in registers, fused residual-add + LayerNorm prologue) for K = d <= 512
  // GWW_GENERIC_PATH (debug aid): bit 0 = generic layer GEMMs, bit 1 = generic conv1, bit 2 = generic conv2, bit 3 = unfused MLP,
  // bit 4 = stand-alone QKV, bit 5 = no pooled last layer, bit 6 = layer 0's LN1 + QKV by the LN-fused A-stationary GEMM, bit 7 = stand-alone out_proj, bit 8 = stand-alone final LayerNorm, bit 9 = A-stationary layer GEMMs at d = 512
  // d = 512 (whisper-base): since round 3 the LayerNorm kernel + the 256 x 256 GEMM (k_gemm_bf16_v3) beat the LN-fused
  // A-stationary layer GEMMs there (8.78 against 9.33 ms per 64 segments; bit 9 of the mask brings them back)
  const bool astat = bf && (d == 384 || (d == 512 && (generic_mask & 512))) && F % 128 == 0 && !(generic_mask & 1);
  const bool mlp_fused = astat && d == 384 && F <= 1536 && !(generic_mask & 8);   // bit 3 = separate fc1 / fc2 kernels
  const bool fuse_qkv = mlp_fused && !(generic_mask & 16);                          // bit 4 = stand-alone LN1 + QKV kernels
  bool qkv_done = false;
  // only the last token wanted: the last layer runs on B rows above its attention (bit 5 of the mask disables it)
  const bool pooled = astat && !last_hidden && last_token && T >= 3 && !(generic_mask & 32);
  if (conv1_direct)
    TR(TR_CONV1, launch_conv1_mel(mel, e->c1w, e->c1b, c1, B, Tin, d, s));
  else if (bf && d % 128 == 0 && !(generic_mask & 2))
    TR(TR_CONV1, launch_gemm_astat(melT, C, nullptr, nullptr, nullptr, nullptr, e->c1w, e->c1b, c1,
                                   (long)B * (Tin + 2), d, kConv1Kpad, EPI_CONV1, Tin + 2, s));
  else
    TR(TR_CONV1, gemm(melT, C, e->c1w, e->c1w32, e->c1b, nullptr, nullptr, c1, (long)B * (Tin + 2), d, kConv1Kpad,
                      EPI_CONV1, Tin + 2));
  if (bf && d % 128 == 0 && d <= 3072 && !(generic_mask & 4))   // the eight-phase 256 x 256 GEMM over overlapping rows (gemm_v4.hip)
    TR(TR_CONV2, launch_gemm_bf16_v4(c1, 2L * d, e->c2w, e->c2b, nullptr, x, (long)B * (T + 1), (d + 255) / 256 * 256, 3 * d,
                                     EPI_CONV2, s, 0, e->pos, T + 1, d, x + (((size_t)B * T + 255) / 256 * 256) * d));
  else if (bf && (d == 384 || d == 512) && !(generic_mask & 1024))
    TR(TR_CONV2, launch_gemm_fulln(c1, 2L * d, e->c2w, e->c2b, e->pos, x, (long)B * (T + 1), d, 3 * d, EPI_CONV2, T + 1, s));
  else
    TR(TR_CONV2, gemm(c1, 2L * d, e->c2w, e->c2w32, e->c2b, nullptr, e->pos, x, (long)B * (T + 1), d, 3 * d, EPI_CONV2,
                      T + 1));
  const bool q_log2 = attention_log2q_enabled();   // the LN-folded q panel carries log2(e) (pack_weights)
  float* xc = x;                       // current residual stream
  const void* pending = nullptr;       // bf16 delta not yet added to xc (A-stationary path)
  if (astat) {
    // Deferred residual: out_proj / fc2 emit a bf16 delta; the NEXT LayerNorm prologue does
    // x_new = x + delta (written to the ping-pong buffer), LN(x_new) -> GEMM operand.
    float* xn = (float*)(base + w.x2);
    void* d1 = h;
    void* d2 = base + w.d2;
    for (int i = 0; i < e->cfg.n_layers; ++i) {
      const LayerW& L = e->layers[i];
      if (!qkv_done) {
        if (i == 0 && !pending && fuse_qkv && !(generic_mask & 64)) {
          // layer 0 (no delta pending behind the conv stem): the fused kernel's panel prologue + q / k / v tail
          TR(TR_QKV, launch_lnqkv_fused(xc, L.uqkv, L.cbqkv, L.wqkv_st, qkv, M, d, 3 * d, s));
        } else {
          TR(TR_QKV, launch_gemm_astat(xc, d, pending, pending ? xn : nullptr, L.uqkv, L.cbqkv, L.wqkv_ln, nullptr, qkv, M,
                                       3 * d, d, EPI_BIAS, 0, s));
          if (pending) { float* t = xc; xc = xn; xn = t; }
        }
      }
      qkv_done = false;
      if (i == 0 && skew_event) GWW_HIP(hipEventRecord(skew_event, s));   // the other half batch starts here
      if (pooled && i == e->cfg.n_layers - 1) {
        // ---- pooled forward: only token T-1 is wanted (Signal_vs_Noise/src/model.py:25-26) and everything above
        // the last attention is row-wise.  Attention for the one query tile that holds token T-1, then out_proj /
        // LN2 / fc1 / GELU / fc2 / final LayerNorm on the B last-token rows (xc is complete here: the QKV
        // prologue folded the pending delta in).
        TR(TR_ATTN, launch_attention_bf16(qkv, ctx, B, T, H, s, nullptr, /*last_tile_only=*/true, q_log2));
        float* xl = xn;                        // [B, d] x rows (b, T-1)        (the ping-pong buffer is free now)
        float* xm = xn + (size_t)B * d;        // [B, d] x_mid
        float* xf = xn + 2 * (size_t)B * d;    // [B, d] layer output
        GWW_HIP(hipMemcpy2DAsync(xl, (size_t)d * 4, xc + (size_t)(T - 1) * d, (size_t)T * d * 4, (size_t)d * 4, B,
                                 hipMemcpyDeviceToDevice, s));
        TR(TR_OUT, launch_gemm_bf16((const unsigned short*)ctx + (size_t)(T - 1) * d, (long)T * d, L.wo, L.bo, xl, nullptr,
                                    xm, B, d, d, EPI_RESID, 0, s, 0));
        TR(TR_LNROWS, launch_layernorm(xm, L.ln2w, L.ln2b, d1, 1, B, d, s));
        TR(TR_FC1, launch_gemm_bf16(d1, d, L.w1, L.b1, nullptr, nullptr, f1, B, F, d, EPI_GELU, 0, s, 0));
        TR(TR_FC2, launch_gemm_bf16(f1, F, L.w2, L.b2, xm, nullptr, xf, B, d, F, EPI_RESID, 0, s, 0));
        TR(TR_LNROWS, launch_layernorm_rows(xf, d, e->lnw, e->lnb, last_token, B, d, s, nullptr));
        return GWW_OK;
      }
      TR(TR_ATTN, launch_attention_bf16(qkv, ctx, B, T, H, s, nullptr, false, q_log2));
      // out_proj fused in front of the MLP block (ctx is the A operand of a GEMM into the block's idle output
      // accumulators; the bf16 delta never reaches HBM): needs no delta pending on xc, which holds on this path
      const bool op = mlp_fused && fuse_qkv && !pending && !(generic_mask & 128);
      if (!op)
        TR(TR_OUT, launch_gemm_astat(ctx, d, nullptr, nullptr, nullptr, nullptr, L.wo, L.bo, d1, M, d, d, EPI_BIAS, 0, s));
      if (mlp_fused && fuse_qkv && i + 1 < e->cfg.n_layers) {
        // ... and the next layer's LN1 + q / k / v projection appended: xn receives x_next (no delta pending)
        const LayerW& Ln = e->layers[i + 1];
        // (x_next comes back in xc itself: xn only holds x_new, the block's intermediate residual stream)
        TR(TR_MLPQKV, launch_mlp_fused(xc, op ? ctx : d1, xn, L.u1, L.cb1, op ? L.wmlp_op : L.wmlp, L.b2, nullptr, M, d, F, s,
                                       Ln.uqkv, Ln.cbqkv, qkv, 3 * d, nullptr, op ? L.bo : nullptr, /*keep_x_new=*/false));
        pending = nullptr;
        qkv_done = true;
        continue;
      }
      static const bool fuse_final = !(lab_int("GWW_GENERIC_PATH", 0) & 256);   // bit 8: stand-alone final LayerNorm
      if (mlp_fused && op && fuse_final && i == e->cfg.n_layers - 1 && last_hidden) {
        // the LAST block with the encoder's final LayerNorm as its epilogue: last_hidden_state comes straight out of the
        // kernel (no bf16 delta, no second read of the residual stream, no LayerNorm launch); the pooled token is row
        // T - 1 of it
        TR(TR_MLPFIN, launch_mlp_fused_final(xc, ctx, xn, L.u1, L.cb1, L.wmlp_op, L.b2, L.bo, e->lnw, e->lnb, last_hidden, M, d,
                                             F, s, /*keep_x_new=*/false));
        if (last_token)
          GWW_HIP(hipMemcpy2DAsync(last_token, (size_t)d * 4, last_hidden + (size_t)(T - 1) * d, (size_t)T * d * 4, (size_t)d * 4,
                                   B, hipMemcpyDeviceToDevice, s));
        return GWW_OK;
      }
      if (mlp_fused) {
        // LN2 + fc1 + GELU + fc2 in one kernel: the [M, ffn] activation never leaves the CU
        TR(TR_MLP, launch_mlp_fused(xc, op ? ctx : d1, xn, L.u1, L.cb1, op ? L.wmlp_op : L.wmlp, L.b2, d2, M, d, F, s, nullptr,
                                    nullptr, nullptr, 0, nullptr, op ? L.bo : nullptr));
        { float* t = xc; xc = xn; xn = t; }
      } else {
        TR(TR_FC1, launch_gemm_astat(xc, d, d1, xn, L.u1, L.cb1, L.w1_ln, nullptr, f1, M, F, d, EPI_GELU, 0, s));
        { float* t = xc; xc = xn; xn = t; }
        TR(TR_FC2, launch_gemm_fulln(f1, F, L.w2, L.b2, nullptr, d2, M, d, F, EPI_BIAS, 0, s));
      }
      pending = d2;
    }
  } else {
    // only the last token wanted (bf16 generic path, e.g. whisper-small): same pooled last layer as above
    const bool pooled_g = bf && !last_hidden && last_token && T >= 3 && !(generic_mask & 32);
    for (int i = 0; i < e->cfg.n_layers; ++i) {
      const LayerW& L = e->layers[i];
      TR(TR_LN, launch_layernorm(x, L.ln1w, L.ln1b, h, bf ? 1 : 0, M, d, s));
      TR(TR_QKV, gemm(h, d, L.wqkv, L.wqkv32, bf ? L.bqkv16 : L.bqkv, nullptr, nullptr, qkv, M, 3 * d, d, EPI_BIAS, 0));
      if (pooled_g && i == e->cfg.n_layers - 1) {
        TR(TR_ATTN, launch_attention_bf16(qkv, ctx, B, T, H, s, nullptr, /*last_tile_only=*/true, q_log2));
        float* xs2 = (float*)(base + w.x2);    // [B, d] x rows (b, T-1) | x_mid | layer output
        float* xl = xs2;
        float* xm = xs2 + (size_t)B * d;
        float* xf = xs2 + 2 * (size_t)B * d;
        GWW_HIP(hipMemcpy2DAsync(xl, (size_t)d * 4, x + (size_t)(T - 1) * d, (size_t)T * d * 4, (size_t)d * 4, B,
                                 hipMemcpyDeviceToDevice, s));
        TR(TR_OUT, launch_gemm_bf16((const unsigned short*)ctx + (size_t)(T - 1) * d, (long)T * d, L.wo, L.bo, xl, nullptr,
                                    xm, B, d, d, EPI_RESID, 0, s, 0));
        TR(TR_LNROWS, launch_layernorm(xm, L.ln2w, L.ln2b, h, 1, B, d, s));
        TR(TR_FC1, launch_gemm_bf16(h, d, L.w1, L.b1, nullptr, nullptr, f1, B, F, d, EPI_GELU, 0, s, 0));
        TR(TR_FC2, launch_gemm_bf16(f1, F, L.w2, L.b2, xm, nullptr, xf, B, d, F, EPI_RESID, 0, s, 0));
        TR(TR_LNROWS, launch_layernorm_rows(xf, d, e->lnw, e->lnb, last_token, B, d, s, nullptr));
        return GWW_OK;
      }
      if (bf) TR(TR_ATTN, launch_attention_bf16(qkv, ctx, B, T, H, s, nullptr, false, q_log2));
      else TR(TR_ATTN, launch_attention_f32((const float*)qkv, (float*)ctx, B, T, H, s));
      TR(TR_OUT, gemm(ctx, d, L.wo, L.wo32, L.bo, x, nullptr, x, M, d, d, EPI_RESID, 0));
      TR(TR_LN, launch_layernorm(x, L.ln2w, L.ln2b, h, bf ? 1 : 0, M, d, s));
      TR(TR_FC1, gemm(h, d, L.w1, L.w132, L.b1, nullptr, nullptr, f1, M, F, d, EPI_GELU, 0));
      TR(TR_FC2, gemm(f1, F, L.w2, L.w232, L.b2, x, nullptr, x, M, d, F, EPI_RESID, 0));
    }
  }
  // ---- final LayerNorm (HF:modeling_whisper.py:642), with the last pending delta folded in;
  // callers pool token T-1 (Signal_vs_Noise/src/model.py:25-26): that row alone is a fast output
  if (last_hidden) TR(TR_LN, launch_layernorm(xc, e->lnw, e->lnb, last_hidden, 0, M, d, s, pending));
  if (last_token)
    TR(TR_LNROWS, launch_layernorm_rows(xc + (long)(T - 1) * d, (long)T * d, e->lnw, e->lnb, last_token, B, d, s,
                                    pending ? (const char*)pending + (size_t)(T - 1) * d * 2 : nullptr));
#undef TR
  return GWW_OK;
}


extern "C" int gww_encoder_forward(gww_encoder* e, const float* mel, int batch, int precision, void* workspace,
                                   size_t workspace_bytes, float* last_hidden, float* last_token,
                                   void* stream) {
  GWW_REQUIRE(e && mel, "gww_encoder_forward: NULL argument");
  if (batch <= 0 || !use_split(e, batch))
    return forward_impl(e, mel, batch, precision, workspace, workspace_bytes, last_hidden, last_token, stream);
  // two independent half batches on two streams, forked from / joined to the caller's stream
  hipStream_t s = (hipStream_t)stream;
  const int d = e->cfg.d_model, T = e->cfg.t_in / 2;
  const int b[2] = {batch / 2, batch - batch / 2};
  const size_t w0 = ws_layout(e->cfg, b[0], precision).total, w1 = ws_layout(e->cfg, b[1], precision).total;
  if (!workspace || workspace_bytes < w0 + w1)
    return fail(GWW_ERR_WORKSPACE, "gww_encoder_forward: workspace %zu bytes < required %zu", workspace_bytes, w0 + w1);
  GWW_HIP(hipEventRecord(e->ev_fork, s));
  int off = 0;
  static const int skew = (int)lab_int("GWW_SPLIT_SKEW", 0);   // tuning aid (lab build)
  for (int i = 0; i < 2; ++i) {
    GWW_HIP(hipStreamWaitEvent(e->s2[i], e->ev_fork, 0));
    if (i == 1 && skew) GWW_HIP(hipStreamWaitEvent(e->s2[1], e->ev_skew, 0));
    const int rc = forward_impl(e, mel + (size_t)off * e->cfg.n_mels * e->cfg.t_in, b[i], precision,
                                (char*)workspace + (i ? w0 : 0), i ? w1 : w0,
                                last_hidden ? last_hidden + (size_t)off * T * d : nullptr,
                                last_token ? last_token + (size_t)off * d : nullptr, e->s2[i],
                                (i == 0 && skew) ? e->ev_skew : nullptr);
    if (rc != GWW_OK) return rc;
    GWW_HIP(hipEventRecord(e->ev_join[i], e->s2[i]));
    GWW_HIP(hipStreamWaitEvent(s, e->ev_join[i], 0));
    off += b[i];
  }
  return GWW_OK;
}


// =====================================================================================
// DoRA training step (bf16): forward that keeps the activations the backward needs, and the
// backward through the whole layer stack down to the residual stream entering layer 0.
// Plain per-op path (LayerNorm kernel, generic GEMM with residual epilogue): the weights are
// frozen, so there are no weight-gradient GEMMs -- only dX GEMMs against transposed panels,
// the flash-attention backward and the rank-r DoRA parameter gradients.
//
// saved arena (caller-owned), per layer l:
//   x_in[l] f32 [Mp,d] | h1 bf16 [Mp,d] | qkv bf16 [Mp,3d] | lse f32 [B,H,T] | ctx bf16 [Mp,d] |
//   x_mid f32 [Mp,d] | z bf16 [Mp,ffn]                      and x_in[L] = input of the final LayerNorm
namespace {
struct SavedLayout {
  size_t layer_stride, x_in, h1, qkv, lse, ctx, x_mid, z, total;
};
SavedLayout saved_layout(const gww_enc_cfg& c, int B) {
  const size_t d = c.d_model, F = c.ffn, T = c.t_in / 2, H = c.n_heads;
  const size_t Mp = ((size_t)B * T + 255) / 256 * 256 + 512;
  SavedLayout s{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  s.x_in = take(Mp * d * 4);
  s.h1 = take(Mp * d * 2);
  s.qkv = take(Mp * 3 * d * 2);
  s.lse = take((size_t)B * H * T * 4);
  s.ctx = take(Mp * d * 2);
  s.x_mid = take(Mp * d * 4);
  s.z = take(Mp * F * 2);
  s.layer_stride = off;
  s.total = off * c.n_layers + align_up(Mp * d * 4);   // + x_in[L]
  return s;
}
struct TrainWs {
  size_t melT, c1, h2, f1, d2, dx, dxb, dbig, dh, dctx, dqkv, Dv, z1, col1, dgs, dgs_bytes, total;
};
TrainWs train_ws(const gww_enc_cfg& c, int B) {
  const size_t d = c.d_model, F = c.ffn, Tin = c.t_in, T = c.t_in / 2, C = c.n_mels, H = c.n_heads;
  const size_t Mp = ((size_t)B * T + 255) / 256 * 256 + 512;
  TrainWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  w.melT = take(((size_t)B * (Tin + 2) * C + kConv1Kpad) * 2);
  w.c1 = take((((size_t)B * (Tin + 2) + 255) / 256 * 256 + 520) * d * 2);
  w.h2 = take(Mp * d * 2);      // per-op path: LN2 output; fused path: out_proj delta (fwd), recomputed LN1 output (bwd)
  w.f1 = take(Mp * F * 2);      // per-op path: gelu(fc1); fused path: recomputed pre-GELU fc1 output (bwd)
  w.d2 = take(Mp * d * 2);      // fused path: the last layer's fc2 delta
  w.dx = take(Mp * d * 4);
  w.dxb = take(Mp * d * 2);
  w.dbig = take(Mp * F * 2);
  w.dh = take(Mp * d * 2);
  w.dctx = take(Mp * d * 2);
  w.dqkv = take(Mp * 3 * d * 2);
  w.Dv = take((size_t)B * H * (T + (T + 63) / 64) * 4);   // row dots + live-tile flags
  w.z1 = take(((size_t)B * (Tin + 2) + 256) * d * 2);            // stem backward: conv1 pre-activation / its gradient
  w.col1 = take(((size_t)B * (Tin + 2) + 256) * kConv1Kpad * 2); // stem backward: conv1 taps side by side
  w.dgs_bytes = (d == 384 || d == 512) ? dora_grads_scratch_bytes(3, (int)d)        // DoRA-gradient partial sums
                : d == 768 ? dora_grads_scratch_bytes(1, (int)d) : 0;
  w.dgs = take(w.dgs_bytes);
  w.total = off;
  return w;
}
}  // namespace

// d = 384: the training forward runs on the fused inference kernels (GWW_TRAIN_FUSED=0: the per-op forward of round 1)
static bool train_fused(const gww_enc_cfg& c) {
  static const bool off = lab_int("GWW_TRAIN_FUSED", 1) == 0;
  return !off && c.d_model == 384 && c.ffn % 128 == 0 && c.ffn <= 1536;
}

extern "C" size_t gww_train_saved_bytes(const gww_encoder* e, int batch) {
  return (e && batch > 0) ? saved_layout(e->cfg, batch).total : 0;
}
extern "C" size_t gww_train_workspace_bytes(const gww_encoder* e, int batch) {
  return (e && batch > 0) ? train_ws(e->cfg, batch).total : 0;
}

extern "C" int gww_encoder_train_forward(gww_encoder* e, const float* mel, int batch, void* workspace,
                                         size_t workspace_bytes, void* saved, size_t saved_bytes,
                                         float* last_hidden, int pooled, void* stream) {
  GWW_REQUIRE(e && mel && workspace && saved && last_hidden, "gww_encoder_train_forward: NULL argument");
  if (!e->ready) return fail(GWW_ERR_STATE, "gww_encoder_train_forward: weights not set");
  GWW_REQUIRE(batch > 0, "gww_encoder_train_forward: batch must be positive");
  const SavedLayout sl = saved_layout(e->cfg, batch);
  const TrainWs w = train_ws(e->cfg, batch);
  if (workspace_bytes < w.total || saved_bytes < sl.total)
    return fail(GWW_ERR_WORKSPACE, "gww_encoder_train_forward: workspace %zu / saved %zu bytes < required %zu / %zu",
                workspace_bytes, saved_bytes, w.total, sl.total);
  hipStream_t s = (hipStream_t)stream;
  const int d = e->cfg.d_model, F = e->cfg.ffn, Tin = e->cfg.t_in, T = Tin / 2, C = e->cfg.n_mels, H = e->cfg.n_heads;
  const int B = batch, L = e->cfg.n_layers;
  const long M = (long)B * T;
  char* base = (char*)workspace;
  char* sv = (char*)saved;
  void* melT = base + w.melT;
  void* c1 = base + w.c1;
  void* h2 = base + w.h2;
  void* f1 = base + w.f1;
  auto x_in = [&](int l) -> float* { return (float*)(sv + (l < L ? (size_t)l * sl.layer_stride + sl.x_in : (size_t)L * sl.layer_stride)); };
  // ---- stem (same kernels as inference) -> x_in[0]
  GWW_TRY(launch_mel_to_tokens(mel, melT, 1, B, C, Tin, s));
  GWW_HIP(hipMemsetAsync((char*)melT + (size_t)B * (Tin + 2) * C * 2, 0, kConv1Kpad * 2, s));
  GWW_HIP(hipMemsetAsync(c1, 0, (size_t)d * 2, s));
  if (train_fused(e->cfg)) {   // the inference stem kernels (A-stationary conv1, full-N conv2)
    GWW_TRY(launch_gemm_astat(melT, C, nullptr, nullptr, nullptr, nullptr, e->c1w, e->c1b, c1, (long)B * (Tin + 2), d,
                              kConv1Kpad, EPI_CONV1, Tin + 2, s));
    GWW_TRY(launch_gemm_bf16_v4(c1, 2L * d, e->c2w, e->c2b, nullptr, x_in(0), (long)B * (T + 1), (d + 255) / 256 * 256, 3 * d,
                                EPI_CONV2, s, 0, e->pos, T + 1, d, (float*)h2));   // (h2 is idle here: the scratch row of the garbage rows)
  } else {
    GWW_TRY(launch_gemm_bf16(melT, C, e->c1w, e->c1b, nullptr, nullptr, c1, (long)B * (Tin + 2), d, kConv1Kpad, EPI_CONV1,
                             Tin + 2, s, 0));
    GWW_TRY(launch_gemm_bf16(c1, 2L * d, e->c2w, e->c2b, nullptr, e->pos, x_in(0), (long)B * (T + 1), d, 3 * d, EPI_CONV2,
                             T + 1, s, 1));
  }
  const bool fast = (d == 384 || d == 512) && F % 128 == 0;   // A-stationary kernel for the K = d GEMMs without a residual
  if (train_fused(e->cfg)) {
    // ---- fused forward (d = 384): the INFERENCE kernels -- LayerNorm-folded A-stationary q/k/v GEMM for layer 0, flash
    // attention (+ lse), out_proj as a bf16 delta, fused MLP + the next layer's LN1 + q/k/v -- writing what the
    // backward needs straight into the arena: x_in[l], qkv, lse, ctx, x_mid (= x_in + out_proj, the fused kernel's
    // x_new).  LN1 / LN2 outputs and the pre-GELU fc1 output are NOT kept: the backward recomputes them (that is what
    // the reference's gradient_checkpointing_enable() at MLGWSC-1/train.py:662 trades, too).
    void* d1 = h2;
    void* d2 = base + w.d2;
    const bool q_log2 = attention_log2q_enabled();
    for (int l = 0; l < L; ++l) {
      const LayerW& W = e->layers[l];
      char* lb = sv + (size_t)l * sl.layer_stride;
      void* qkv = lb + sl.qkv;
      float* lse = (float*)(lb + sl.lse);
      void* ctx = lb + sl.ctx;
      float* x_mid = (float*)(lb + sl.x_mid);
      void* z = lb + sl.z;
      if (l == 0) {
        // layer 0: LN1 + q / k / v on the fused block's panel prologue + tail (k_mlp_fused<2, false>), as the inference
        // forward does (round 3 ran the LayerNorm kernel + the plain A-stationary GEMM here: 41 + 112 us at 64 segments)
        GWW_TRY(launch_lnqkv_fused(x_in(0), W.uqkv, W.cbqkv, W.wqkv_st, qkv, M, d, 3 * d, s));
      }
      if (pooled && l == L - 1) {
        // only the query tile that holds token T - 1 is needed (forward and backward): the other rows of ctx / lse stay
        // zero so that the backward's row dots see finite values
        GWW_HIP(hipMemsetAsync(ctx, 0, (size_t)M * d * 2, s));
        GWW_HIP(hipMemsetAsync(lse, 0, (size_t)B * H * T * 4, s));
        GWW_TRY(launch_attention_bf16(qkv, ctx, B, T, H, s, lse, /*last_tile_only=*/true, q_log2));
        float* xl = (float*)(base + w.dx);
        GWW_HIP(hipMemcpy2DAsync(xl, (size_t)d * 4, x_in(l) + (size_t)(T - 1) * d, (size_t)T * d * 4, (size_t)d * 4, B,
                                 hipMemcpyDeviceToDevice, s));
        GWW_TRY(launch_gemm_bf16((const unsigned short*)ctx + (size_t)(T - 1) * d, (long)T * d, W.wo, W.bo, xl, nullptr,
                                 x_mid, B, d, d, EPI_RESID, 0, s, 0));
        GWW_TRY(launch_layernorm(x_mid, W.ln2w, W.ln2b, d1, 1, B, d, s));
        GWW_TRY(launch_gemm_bf16(d1, d, W.w1, W.b1, nullptr, nullptr, z, B, F, d, EPI_BIAS, 0, s, 0));
        GWW_TRY(launch_gelu_bf16(z, nullptr, f1, (((long)B * F + 7) / 8) * 8, s));
        GWW_TRY(launch_gemm_bf16(f1, F, W.w2, W.b2, x_mid, nullptr, x_in(L), B, d, F, EPI_RESID, 0, s, 0));
        GWW_TRY(launch_layernorm(x_in(L), e->lnw, e->lnb, last_hidden, 0, B, d, s));
        return GWW_OK;
      }
      GWW_TRY(launch_attention_bf16(qkv, ctx, B, T, H, s, lse, false, q_log2));
      // out_proj fused in front of the block as on the inference path: x_mid = x_in + bf16(ctx W_o^T + bo) comes out of
      // the kernel's seam (the backward needs ctx and x_mid, never the delta)
      static const bool op = !(lab_int("GWW_GENERIC_PATH", 0) & 128);
      if (!op)
        GWW_TRY(launch_gemm_astat(ctx, d, nullptr, nullptr, nullptr, nullptr, W.wo, W.bo, d1, M, d, d, EPI_BIAS, 0, s));
      const void* a2 = op ? (const void*)ctx : (const void*)d1;
      const void* st2 = op ? W.wmlp_op : W.wmlp;
      const float* bo2 = op ? W.bo : nullptr;
      if (l + 1 < L) {
        const LayerW& Wn = e->layers[l + 1];
        void* qkv_n = sv + (size_t)(l + 1) * sl.layer_stride + sl.qkv;
        GWW_TRY(launch_mlp_fused(x_in(l), a2, x_mid, W.u1, W.cb1, st2, W.b2, nullptr, M, d, F, s, Wn.uqkv, Wn.cbqkv, qkv_n,
                                 3 * d, x_in(l + 1), bo2));
      } else {
        GWW_TRY(launch_mlp_fused(x_in(l), a2, x_mid, W.u1, W.cb1, st2, W.b2, d2, M, d, F, s, nullptr, nullptr, nullptr, 0,
                                 nullptr, bo2));
        GWW_TRY(launch_add_delta_f32(x_mid, d2, x_in(L), M * d, s));
      }
    }
    GWW_TRY(launch_layernorm(x_in(L), e->lnw, e->lnb, last_hidden, 0, M, d, s));
    return GWW_OK;
  }
  for (int l = 0; l < L; ++l) {
    const LayerW& W = e->layers[l];
    char* lb = sv + (size_t)l * sl.layer_stride;
    void* h1 = lb + sl.h1;
    void* qkv = lb + sl.qkv;
    float* lse = (float*)(lb + sl.lse);
    void* ctx = lb + sl.ctx;
    float* x_mid = (float*)(lb + sl.x_mid);
    void* z = lb + sl.z;
    GWW_TRY(launch_layernorm(x_in(l), W.ln1w, W.ln1b, h1, 1, M, d, s));
    if (fast) GWW_TRY(launch_gemm_astat(h1, d, nullptr, nullptr, nullptr, nullptr, W.wqkv, W.bqkv16, qkv, M, 3 * d, d, EPI_BIAS, 0, s));
    else GWW_TRY(launch_gemm_bf16(h1, d, W.wqkv, W.bqkv16, nullptr, nullptr, qkv, M, 3 * d, d, EPI_BIAS, 0, s, 1));
    GWW_TRY(launch_attention_bf16(qkv, ctx, B, T, H, s, lse, false, attention_log2q_enabled()));
    if (pooled && l == L - 1) {
      // Only token T-1 of the output is used (Signal_vs_Noise/src/model.py:25-26), and past the last attention
      // every op is row-wise: run out_proj / LN2 / fc1 / GELU / fc2 / final LN on the B last-token rows alone.
      // x_mid, z and x_in[L] of this layer are saved COMPACT ([B, .]) -- the pooled backward expects exactly that.
      float* xl = (float*)(base + w.dx);   // x_in[L-1] rows (b, T-1); the gradient buffers are idle in the forward
      GWW_HIP(hipMemcpy2DAsync(xl, (size_t)d * 4, x_in(l) + (size_t)(T - 1) * d, (size_t)T * d * 4, (size_t)d * 4, B,
                               hipMemcpyDeviceToDevice, s));
      GWW_TRY(launch_gemm_bf16((const unsigned short*)ctx + (size_t)(T - 1) * d, (long)T * d, W.wo, W.bo, xl, nullptr,
                               x_mid, B, d, d, EPI_RESID, 0, s, 0));
      GWW_TRY(launch_layernorm(x_mid, W.ln2w, W.ln2b, h2, 1, B, d, s));
      GWW_TRY(launch_gemm_bf16(h2, d, W.w1, W.b1, nullptr, nullptr, z, B, F, d, EPI_BIAS, 0, s, 0));
      GWW_TRY(launch_gelu_bf16(z, nullptr, f1, (((long)B * F + 7) / 8) * 8, s));
      GWW_TRY(launch_gemm_bf16(f1, F, W.w2, W.b2, x_mid, nullptr, x_in(L), B, d, F, EPI_RESID, 0, s, 0));
      GWW_TRY(launch_layernorm(x_in(L), e->lnw, e->lnb, last_hidden, 0, B, d, s));
      return GWW_OK;
    }
    GWW_TRY(launch_gemm_bf16(ctx, d, W.wo, W.bo, x_in(l), nullptr, x_mid, M, d, d, EPI_RESID, 0, s, 1));
    GWW_TRY(launch_layernorm(x_mid, W.ln2w, W.ln2b, h2, 1, M, d, s));
    if (fast) GWW_TRY(launch_gemm_astat(h2, d, nullptr, nullptr, nullptr, nullptr, W.w1, W.b1, z, M, F, d, EPI_BIAS, 0, s));
    else GWW_TRY(launch_gemm_bf16(h2, d, W.w1, W.b1, nullptr, nullptr, z, M, F, d, EPI_BIAS, 0, s, 1));
    GWW_TRY(launch_gelu_bf16(z, nullptr, f1, ((M * F + 7) / 8) * 8, s));
    GWW_TRY(launch_gemm_bf16(f1, F, W.w2, W.b2, x_mid, nullptr, x_in(l + 1), M, d, F, EPI_RESID, 0, s, 1));
  }
  GWW_TRY(launch_layernorm(x_in(L), e->lnw, e->lnb, last_hidden, 0, M, d, s));
  return GWW_OK;
}

// d_last_hidden: fp32 [B*T, d] gradient of the loss w.r.t. last_hidden_state.
// targets: DoRA-adapted projections whose A / B / magnitude gradients are wanted; the gradient buffers
// are ACCUMULATED into (zero them once per step).  d_x0 (optional, fp32 [B*T, d]): gradient w.r.t. the
// residual stream entering layer 0 (the conv stem output).  d_mel (optional, fp32 [B, n_mels, t_in]): gradient
// w.r.t. the input features, through the conv stem (MLGWSC-1/train.py:494-504 trains its Q-adapter through
// the frozen encoder).
extern "C" int gww_encoder_train_backward(gww_encoder* e, int batch, void* workspace, size_t workspace_bytes,
                                          const void* saved, size_t saved_bytes, const float* d_last_hidden,
                                          const gww_dora_target* targets, int n_targets, float* d_x0,
                                          float* d_mel, int pooled, void* stream) {
  GWW_REQUIRE(e && workspace && saved && d_last_hidden, "gww_encoder_train_backward: NULL argument");
  GWW_REQUIRE(batch > 0 && n_targets >= 0 && (n_targets == 0 || targets), "gww_encoder_train_backward: bad argument");
  const SavedLayout sl = saved_layout(e->cfg, batch);
  const TrainWs w = train_ws(e->cfg, batch);
  if (workspace_bytes < w.total || saved_bytes < sl.total)
    return fail(GWW_ERR_WORKSPACE, "gww_encoder_train_backward: workspace / saved arena too small");
  hipStream_t s = (hipStream_t)stream;
  const int d = e->cfg.d_model, F = e->cfg.ffn, T = e->cfg.t_in / 2, H = e->cfg.n_heads;
  const int B = batch, L = e->cfg.n_layers;
  const long M = (long)B * T;
  char* base = (char*)workspace;
  const char* sv = (const char*)saved;
  float* dx = (float*)(base + w.dx);
  void* dxb = base + w.dxb;
  void* dbig = base + w.dbig;
  void* dh = base + w.dh;
  void* dctx = base + w.dctx;
  void* dqkv = base + w.dqkv;
  float* Dv = (float*)(base + w.Dv);
  auto x_in = [&](int l) -> const float* { return (const float*)(sv + (l < L ? (size_t)l * sl.layer_stride + sl.x_in : (size_t)L * sl.layer_stride)); };
  for (int i = 0; i < n_targets; ++i) {
    const gww_dora_target& t = targets[i];
    GWW_REQUIRE(t.layer >= 0 && t.layer < L && t.proj >= 0 && t.proj <= 3, "gww_encoder_train_backward: bad target %d", i);
    GWW_REQUIRE(t.A && t.B && t.mag && t.nrm && t.dA && t.dB && t.dm, "gww_encoder_train_backward: NULL pointer in target %d", i);
  }
  // dX GEMMs: A-stationary kernel for the K <= 512 contractions, full-N kernel for the long-K, N = d ones
  // (d = 384 / 512); generic tiles otherwise.  All buffers are padded to whole 256-row panels.
  const bool fast = (d == 384 || d == 512) && F % 128 == 0;
  auto gemm_dx = [&](const void* A, long lda, const void* Wt, void* Cout, int N, int K) -> int {
    // the wide product of the MLP backward (d(fc1 output) = d(out) W2: N = ffn, K = d) on the 256 x 256 x 64 kernel: 113 GFLOP
    // in ~130 us against ~200 on the A-stationary kernel, whose 12 n-tile epilogues per panel run with nothing beside them
    if (fast && lda == K && N % 256 == 0 && N >= 1024 && K % 128 == 0) {
      const int rc = launch_gemm_bf16_v4(A, lda, Wt, nullptr, nullptr, Cout, M, N, K, EPI_BIAS, s);
      if (rc != -1) return rc;
    }
    if (fast && lda == K && (K == 384 || K == 512) && N % 128 == 0)
      return launch_gemm_astat(A, lda, nullptr, nullptr, nullptr, nullptr, Wt, nullptr, Cout, M, N, K, EPI_BIAS, 0, s);
    if (fast && N == d && K % 64 == 0 && K > 512)
      return launch_gemm_fulln(A, lda, Wt, nullptr, nullptr, Cout, M, N, K, EPI_BIAS, 0, s);
    return launch_gemm_bf16(A, lda, Wt, nullptr, nullptr, nullptr, Cout, M, N, K, EPI_BIAS, 0, s, 1);
  };
  // the stored q is  q_ysc * (W' x + b): 1 / 8 (head_dim^-0.5), times log2(e) when the bf16 panels carry log2 units
  const float q_ysc = attention_log2q_enabled() ? 0.125f * 1.44269504088896340736f : 0.125f;
  bool multi_ok = (d == 384 || d == 512) && lab_int("GWW_DORA_OLD", 0) == 0;
  for (int i = 0; i < n_targets; ++i) multi_ok = multi_ok && targets[i].r == 8;
  // final LayerNorm backward -> dx (grad w.r.t. x_in[L]); pooled: on the B last-token rows only
  GWW_TRY(launch_ln_bwd(x_in(L), e->lnw, d_last_hidden, 1, dx, 0, dxb, pooled ? B : M, d, s));
  for (int l = L - 1; l >= 0; --l) {
    const LayerW& W = e->layers[l];
    const char* lb = sv + (size_t)l * sl.layer_stride;
    const void* h1 = lb + sl.h1;
    const void* qkv = lb + sl.qkv;
    const float* lse = (const float*)(lb + sl.lse);
    const void* ctx = lb + sl.ctx;
    const float* x_mid = (const float*)(lb + sl.x_mid);
    const void* z = lb + sl.z;
    const bool fused = train_fused(e->cfg);
    if (fused) {
      // the fused forward kept neither LN1(x_in) (the X operand of the q / k / v adapter gradients) nor the pre-GELU
      // fc1 output: LN1 is recomputed here, fc1 inside the GELU-backward GEMM below
      GWW_TRY(launch_layernorm(x_in(l), W.ln1w, W.ln1b, base + w.h2, 1, M, d, s));
      h1 = base + w.h2;
    }
    if (pooled && l == L - 1) {
      // ---- last layer of a pooled step: everything above the attention lives on the B last-token rows
      // (x_mid, z, x_in[L] were saved compact by the pooled forward); the attention backward then sees a dctx
      // that is zero except for row T-1 of every segment and skips the dead query tiles.
      GWW_TRY(launch_gemm_bf16(dxb, d, W.w2T, nullptr, nullptr, nullptr, dbig, B, F, d, EPI_BIAS, 0, s, 0));
      GWW_TRY(launch_gelu_bf16(z, dbig, dbig, (((long)B * F + 7) / 8) * 8, s));
      GWW_TRY(launch_gemm_bf16(dbig, F, W.w1T, nullptr, nullptr, nullptr, dh, B, d, F, EPI_BIAS, 0, s, 0));
      GWW_TRY(launch_ln_bwd(x_mid, W.ln2w, dh, 0, dx, 1, dxb, B, d, s));
      const unsigned short* ctx_last = (const unsigned short*)ctx + (size_t)(T - 1) * d;
      bool have_y = false;
      for (int i = 0; i < n_targets; ++i) {
        const gww_dora_target& t = targets[i];
        if (t.layer != l || t.proj != 3) continue;
        if (!have_y) {   // y = x_mid - x_in on the last-token rows
          float* xl = (float*)dbig;
          GWW_HIP(hipMemcpy2DAsync(xl, (size_t)d * 4, x_in(l) + (size_t)(T - 1) * d, (size_t)T * d * 4, (size_t)d * 4,
                                   B, hipMemcpyDeviceToDevice, s));
          GWW_TRY(launch_sub_f32_bf16(x_mid, xl, dh, (long)B * d, s));
          have_y = true;
        }
        GWW_TRY(launch_dora_grads(ctx_last, (long)T * d, dxb, dh, d, W.bo, 1.0f, t.scaling, t.A, t.B, t.mag, t.nrm,
                                  t.dA, t.dB, t.dm, B, d, t.r, s, base + w.dgs, w.dgs_bytes));
      }
      // d(ctx) rows (b, T-1) -> the dense, otherwise zero dctx
      GWW_TRY(launch_gemm_bf16(dxb, d, W.woT, nullptr, nullptr, nullptr, dh, B, d, d, EPI_BIAS, 0, s, 0));
      GWW_HIP(hipMemsetAsync(dctx, 0, (size_t)M * d * 2, s));
      GWW_HIP(hipMemcpy2DAsync((unsigned short*)dctx + (size_t)(T - 1) * d, (size_t)T * d * 2, dh, (size_t)d * 2,
                               (size_t)d * 2, B, hipMemcpyDeviceToDevice, s));
      // the residual gradient likewise: compact dx -> row T-1 of a zero dense dx
      GWW_HIP(hipMemcpyAsync(dbig, dx, (size_t)B * d * 4, hipMemcpyDeviceToDevice, s));
      GWW_HIP(hipMemsetAsync(dx, 0, (size_t)M * d * 4, s));
      GWW_HIP(hipMemcpy2DAsync(dx + (size_t)(T - 1) * d, (size_t)T * d * 4, dbig, (size_t)d * 4, (size_t)d * 4, B,
                               hipMemcpyDeviceToDevice, s));
    } else {
    // fc2 / GELU / fc1 / LN2   (x_out = x_mid + fc2(gelu(fc1(LN2(x_mid)))))
    GWW_TRY(gemm_dx(dxb, d, W.w2T, dbig, F, d));
    if (fused) {
      // recompute: LN2(x_mid) (LayerNorm kernel, into the idle dctx buffer) -> fc1 as a plain A-stationary GEMM whose
      // epilogue applies gelu'(pre-activation) to the gradient in place: neither the pre-activation nor a separate
      // GELU-backward pass touches HBM
      GWW_TRY(launch_layernorm(x_mid, W.ln2w, W.ln2b, dctx, 1, M, d, s));
      GWW_TRY(launch_gemm_astat(dctx, d, dbig, nullptr, nullptr, nullptr, W.w1, W.b1, dbig, M, F, d, EPI_DGELU, 0, s));
    } else {
      GWW_TRY(launch_gelu_bf16(z, dbig, dbig, ((M * F + 7) / 8) * 8, s));
    }
    GWW_TRY(gemm_dx(dbig, F, W.w1T, dh, d, F));
    GWW_TRY(launch_ln_bwd(x_mid, W.ln2w, dh, 0, dx, 1, dxb, M, d, s));
    // out_proj / attention / QKV / LN1   (x_mid = x_in + out_proj(attn(qkv(LN1(x_in)))))
    {   // out_proj DoRA targets: x = ctx, dy = d(x_mid) (= dxb), y = x_mid - x_in (rebuilt into dh, free here)
      bool have_y = false;
      for (int i = 0; i < n_targets; ++i) {
        const gww_dora_target& t = targets[i];
        if (t.layer != l || t.proj != 3) continue;
        if (!have_y) {
          GWW_TRY(launch_sub_f32_bf16(x_mid, x_in(l), dh, M * d, s));
          have_y = true;
        }
        GWW_TRY(launch_dora_grads(ctx, d, dxb, dh, d, W.bo, 1.0f, t.scaling, t.A, t.B, t.mag, t.nrm, t.dA, t.dB, t.dm,
                                  M, d, t.r, s, base + w.dgs, w.dgs_bytes));
      }
    }
    GWW_TRY(gemm_dx(dxb, d, W.woT, dctx, d, d));
    }
    GWW_TRY(launch_attention_bwd_bf16(qkv, ctx, dctx, lse, Dv, dqkv, B, T, H, s, attention_log2q_enabled()));
    if (multi_ok) {
      // q / k / v adapters of this layer read the same h1: one pass over h1, dqkv and qkv on the matrix cores
      long off[3];
      const float *bias[3], *Aa[3], *Bb[3], *mg[3], *nr[3];
      float ysc[3], scl[3], *dAa[3], *dBb[3], *dmm[3];
      int np = 0;
      for (int i = 0; i < n_targets; ++i) {
        const gww_dora_target& t = targets[i];
        if (t.layer != l || t.proj == 3) continue;
        GWW_REQUIRE(np < 3, "gww_encoder_train_backward: duplicate q/k/v target in layer %d", l);
        off[np] = (long)t.proj * d;
        bias[np] = W.bqkv16 + off[np];
        ysc[np] = t.proj == 0 ? q_ysc : 1.0f;
        scl[np] = t.scaling;
        Aa[np] = t.A; Bb[np] = t.B; mg[np] = t.mag; nr[np] = t.nrm;
        dAa[np] = t.dA; dBb[np] = t.dB; dmm[np] = t.dm;
        ++np;
      }
      if (np > 0)
        GWW_TRY(launch_dora_grads_multi(h1, d, dqkv, qkv, 3L * d, np, off, bias, ysc, scl, Aa, Bb, mg, nr, dAa, dBb, dmm, M,
                                        d, s, base + w.dgs, w.dgs_bytes));
    } else {
      for (int i = 0; i < n_targets; ++i) {
        const gww_dora_target& t = targets[i];
        if (t.layer != l || t.proj == 3) continue;
        const long off = (long)t.proj * d;   // q | k | v section
        GWW_TRY(launch_dora_grads(h1, d, (const unsigned short*)dqkv + off, (const unsigned short*)qkv + off, 3L * d,
                                  W.bqkv16 + off, t.proj == 0 ? q_ysc : 1.0f, t.scaling, t.A, t.B, t.mag, t.nrm, t.dA,
                                  t.dB, t.dm, M, d, t.r, s, base + w.dgs, w.dgs_bytes));
      }
    }
    // below layer 0 the gradient only continues into the conv stem: skip it when nobody asked for d_x0 / d_mel
    if (l == 0 && !d_x0 && !d_mel) break;
    GWW_TRY(gemm_dx(dqkv, 3L * d, W.wqkvT, dh, d, 3 * d));
    GWW_TRY(launch_ln_bwd(x_in(l), W.ln1w, dh, 0, dx, 1, dxb, M, d, s));
  }
  if (d_x0) GWW_HIP(hipMemcpyAsync(d_x0, dx, (size_t)M * d * 4, hipMemcpyDeviceToDevice, s));
  if (d_mel) {
    // ---- conv stem backward: x0 = gelu(conv2(gelu(conv1(mel)))) + pos (melT and c1 of the forward are still
    // in the workspace); the pre-activations are recomputed by the same GEMMs with a plain bias epilogue
    const int Tin = e->cfg.t_in, C = e->cfg.n_mels;
    GWW_REQUIRE(B <= 512, "gww_encoder_train_backward: d_mel supports batch <= 512");
    const void* melT = base + w.melT;
    const void* c1 = base + w.c1;
    void* z1 = base + w.z1;
    void* col1 = base + w.col1;
    const long M2 = (long)B * (T + 1), M1 = (long)B * (Tin + 2);
    GWW_TRY(launch_gemm_bf16(c1, 2L * d, e->c2w, e->c2b, nullptr, nullptr, dh, M2, d, 3 * d, EPI_BIAS, 0, s, 0));       // z2
    GWW_TRY(launch_stem_dz2(dxb, dh, dctx, B, T, d, s));                                                              // dz2
    GWW_TRY(launch_gemm_bf16(dctx, d, e->c2wT, nullptr, nullptr, nullptr, dqkv, M2, 3 * d, d, EPI_BIAS, 0, s, 0));     // col
    GWW_TRY(launch_gemm_bf16(melT, C, e->c1w, e->c1b, nullptr, nullptr, z1, M1, d, kConv1Kpad, EPI_BIAS, 0, s, 0));    // z1
    GWW_TRY(launch_stem_dz1(dqkv, z1, z1, B, T, Tin, d, s));                                                          // dz1
    GWW_TRY(launch_gemm_bf16(z1, d, e->c1wT, nullptr, nullptr, nullptr, col1, M1, kConv1Kpad, d, EPI_BIAS, 0, s, 0));  // col1
    GWW_TRY(launch_stem_dmel(col1, d_mel, B, Tin, C, kConv1Kpad, s));
  }
  return GWW_OK;
}
