// Shared device/host helpers for libgww (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gww.h"
#include <vector>

namespace gww {

// ---- error plumbing -------------------------------------------------------
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);

#define GWW_HIP(expr)                                                                 \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return ::gww::fail(GWW_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                         \
  } while (0)

#define GWW_LAUNCH_CHECK() GWW_HIP(hipGetLastError())

#define GWW_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) return ::gww::fail(GWW_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define GWW_TRY(expr)          \
  do {                         \
    int _rc = (expr);          \
    if (_rc != GWW_OK) return _rc; \
  } while (0)

// ---- vector types ---------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

// fp32 -> bf16 bits, round to nearest even.  A plain cast lowers to
// v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN.
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
// two values -> one dword, ONE v_cvt_pk_bf16_f32 (the scalar casts above cost three instructions per pair)
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU, nn.functional.gelu default
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for the bf16 path: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the
// bf16 rounding of the result); ~12 VALU ops instead of ocml erff's ~40.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  const float erf_abs = fmaf(-p, e, 1.0f);              // erf(|x|/sqrt2)
  const float h = 0.5f * x;
  return fmaf(h, copysignf(erf_abs, x), h);             // 0.5 x (1 + erf(x/sqrt2))
}

// x * sigmoid(p(x)), p an odd quintic fitted to the erf GELU (the fused block's form, mlp_fused.hip): |err| <= 2.6e-5,
// 7 plain VALU operations + v_exp_f32 + v_rcp_f32 -- a third cheaper than the erf polynomial of gelu_fast
__device__ __forceinline__ float gelu_sig4(float x) {
  const float s = fminf(x * x, 64.0f);
  float q = fmaf(s, 0.0010148164f, -0.1067791331f);
  q = fmaf(s, q, -2.3011178f);
  const float e = __builtin_amdgcn_exp2f(x * q);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// gelu'(z) = Phi(z) + z phi(z) for the training backward: the same 7.1.26 polynomial, whose exp(-z^2 / 2) factor is
// sqrt(2 pi) phi(z) -- 16 VALU operations (two transcendental) where ocml erff + expf take about sixty
__device__ __forceinline__ float dgelu_fast(float z) {
  const float az = fabsf(z) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);   // exp(-z^2 / 2)
  const float erf_abs = fmaf(-p, e, 1.0f);
  return fmaf(0.5f, copysignf(erf_abs, z), fmaf(z * 0.3989422804f, e, 0.5f));
}

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Laboratory switches (kernel variants kept for A/B measurements, tuning aids): read from the environment ONLY in the
// -DGWW_LAB build that tools/ use (make LAB=1 -> libgww_lab.so).  In the product library they are compile-time constants:
// no environment variable changes what libgww.so computes or launches (no global mutable state).
#ifdef GWW_LAB
long lab_int(const char* name, long dflt);   // encoder.hip
#else
constexpr long lab_int(const char*, long dflt) { return dflt; }
#endif

// ---- kernels launched from more than one translation unit ------------------
// (definitions in the .hip files; every launcher returns a gww status code)
int launch_layernorm(const float* x, const float* w, const float* b, void* y, int out_bf16,
                     long M, int d, hipStream_t s, const void* delta_bf16 = nullptr);
int launch_layernorm_rows(const float* x, long row_stride, const float* w, const float* b, float* y,
                          long M, int d, hipStream_t s, const void* delta_bf16 = nullptr);
int launch_ln_fold(const float* w, const float* g, const float* bl, const float* bias, float scale, int N, int K,
                   void* wp, float* u, float* cb, hipStream_t s);
int launch_gemm_astat(const void* A, long lda, const void* delta, float* x_out, const float* ln_u,
                      const float* ln_cb, const void* W, const float* bias, void* C, long M, int N, int K,
                      int epi, int rows_per_batch, hipStream_t s, long c_panel_rows = 0);
int launch_cast_f32_bf16(const float* x, void* y, long n, hipStream_t s);
bool conv1_mel_supported(int n_mels, int d, int kpad);
int launch_conv1_mel(const float* mel, const void* W, const float* bias, void* c1, int B, int T, int d, hipStream_t s);
int launch_pack_weight(const float* w, void* out, int out_bf16, int N, int C, int taps, int Kpad,
                       float scale, hipStream_t s);
int launch_scale_copy(const float* in, float* out, int n, float scale, hipStream_t s);

// Batched weight preparation (elementwise.hip::k_prep_batch): the small row-wise kernels of a weight update collected
// into one launch per dependency phase.  Ops of ONE batch must not read each other's outputs; flush() between phases.
enum { PREP_PACK16, PREP_PACK32, PREP_COPY, PREP_LNFOLD, PREP_TRANSPOSE, PREP_DORA };
constexpr int kPrepMaxOps = 40;      // 40 x 80-byte descriptors + the prefix table stay below the 4-KiB kernel-argument limit
constexpr int kPrepMaxRank = 1024;   // LoRA rank the DoRA body stages in LDS
struct PrepOp {
  const void *a, *b, *c, *d;   // inputs   (kind-specific, see PrepBatch's methods)
  void *o0, *o1, *o2;          // outputs
  int kind, blocks, N, K;      // workgroups of the op; rows; columns (Kpad / d_in / Cn)
  int C, taps;                 // pack: input channels, taps; DoRA: C = rank
  float scale;
  int pad_;
};
struct PrepArgs {
  int n;
  int first[kPrepMaxOps + 1];  // first workgroup of op i; first[n] = grid size
  PrepOp op[kPrepMaxOps];
};
int launch_prep_batch(const PrepArgs& P, hipStream_t s);
struct PrepBatch {
  std::vector<PrepArgs> args;  // nothing is launched before flush(): a full table opens the next one
  hipStream_t s;
  explicit PrepBatch(hipStream_t st) : s(st) {}
  int add(const PrepOp& o);
  int flush();                 // one launch per table, in order
  int pack(const float* w, void* out, int out_bf16, int N, int C, int taps, int Kpad, float scale);
  int copy(const float* in, float* out, int n, float scale);
  int ln_fold(const float* w, const float* g, const float* bl, const float* bias, float scale, int N, int K, void* wp,
              float* u, float* cb);
  int transpose(const void* in, void* out, int R, int Cn);
  int dora(const float* w0, const float* a, const float* b, const float* m, float scaling, int d_out, int d_in, int r,
           float* w_eff, float* norm_out);
};
int launch_gemm_bf16(const void* A, long lda, const void* W, const float* bias, const float* resid,
                     const float* pos, void* C, long M, int N, int K, int epi, int rows_per_batch,
                     hipStream_t s, int rows_padded_256 = 0);
int launch_gemm_bf16_v4(const void* A, long lda, const void* W, const float* bias, const float* resid, void* C, long M,
                        int N, int K, int epi, hipStream_t s, int force_split = 0, const float* pos = nullptr,
                        int rows_per_batch = 0, int n_real = 0, float* dump = nullptr);
int launch_gemm_fulln(const void* A, long lda, const void* W, const float* bias, const float* pos, void* C,
                      long M, int N, int K, int epi, int rows_per_batch, hipStream_t s);
int launch_gemm_f32(const float* A, long lda, const float* W, const float* bias, const float* resid,
                    const float* pos, float* C, long M, int N, int K, int epi, int rows_per_batch,
                    hipStream_t s);
int launch_attention_bf16(const void* qkv, void* ctx, int B, int T, int H, hipStream_t s, float* lse = nullptr,
                          bool last_tile_only = false, bool q_log2 = false);
int launch_attention_w64_bf16(const void* qkv, void* ctx, int B, int T, int H, hipStream_t s, float* lse);
bool attention_log2q_enabled();   // the inference path packs q in log2 units (attention.hip)
int launch_attention_f32(const float* qkv, float* ctx, int B, int T, int H, hipStream_t s);
int launch_conv1_bf16(const float* mel, const void* w_packed, const float* bias, void* out,
                      int B, int T, int n_mels, int d, hipStream_t s);
int launch_conv1_f32(const float* mel, const float* w_packed, const float* bias, float* out,
                     int B, int T, int n_mels, int d, hipStream_t s);

// GEMM epilogues
enum : int {
  EPI_BIAS = 0,        // C = acc + bias                    (bf16 / f32 store)
  EPI_GELU = 1,        // C = gelu(acc + bias)
  EPI_RESID = 2,       // C(f32) = resid + acc + bias       (may alias resid)
  EPI_CONV2 = 3,       // conv2: gelu(acc+bias) + pos[t], rows remapped (see gemm_bf16.hip)
};

}  // namespace gww
