// Row-wise / elementwise kernels: LayerNorm, casts, weight packing, DoRA merge.
// All HBM-bound; one wave per row with float2 lanes (d is a multiple of 128 for
// every Whisper size: 384 / 512 / 768 / 1024 / 1280).
#include "common.h"

namespace gww {

thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---------------------------------------------------------------- LayerNorm
// HF:modeling_whisper.py:392,402,642 (nn.LayerNorm, eps 1e-5, affine).
// NV = d / 128 float2 per lane.  Two-pass (mean, then centred variance) in registers.
template <int NV, bool OUT_BF16>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, long row_stride,
                                                   const unsigned short* __restrict__ delta,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ b, void* __restrict__ y,
                                                   long M) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int d = NV * 128;
  const float2* xr = reinterpret_cast<const float2*>(x + row * row_stride);
  float2 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = xr[lane + 64 * j];
    if (delta) {   // deferred residual add: x + bf16 delta of the last GEMM (same row layout)
      const unsigned int dv = reinterpret_cast<const unsigned int*>(delta + row * row_stride)[lane + 64 * j];
      v[j].x += bf2f((unsigned short)(dv & 0xffff));
      v[j].y += bf2f((unsigned short)(dv >> 16));
    }
    s += v[j].x + v[j].y;
  }
  const float mean = wave_sum(s) * (1.0f / d);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j].x -= mean;
    v[j].y -= mean;
    q += v[j].x * v[j].x + v[j].y * v[j].y;
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / d) + 1e-5f);
  const float2* w2 = reinterpret_cast<const float2*>(w);
  const float2* b2 = reinterpret_cast<const float2*>(b);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const float2 ww = w2[lane + 64 * j], bb = b2[lane + 64 * j];
    const float o0 = v[j].x * rstd * ww.x + bb.x;
    const float o1 = v[j].y * rstd * ww.y + bb.y;
    if (OUT_BF16) {
      reinterpret_cast<unsigned int*>(y)[row * (d / 2) + lane + 64 * j] = pack2bf(o0, o1);
    } else {
      reinterpret_cast<float2*>(y)[row * (d / 2) + lane + 64 * j] = make_float2(o0, o1);
    }
  }
}

// The same for d % 256 == 0 (whisper-base / -small: 24 launches per whisper-small forward, 8-9 % of it) with 16 bytes per lane on
// the read side and 8 / 16 on the write side: N4 = d / 256 float4 per lane.
template <int N4, bool OUT_BF16>
__global__ __launch_bounds__(256) void k_layernorm4(const float* __restrict__ x, long row_stride,
                                                    const unsigned short* __restrict__ delta,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    void* __restrict__ y, long M) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int d = N4 * 256;
  const float4* xr = reinterpret_cast<const float4*>(x + row * row_stride);
  float4 v[N4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < N4; ++j) {
    v[j] = xr[lane + 64 * j];
    if (delta) {
      const u32x2 dv = reinterpret_cast<const u32x2*>(delta + row * row_stride)[lane + 64 * j];
      v[j].x += bf2f((unsigned short)(dv[0] & 0xffff)); v[j].y += bf2f((unsigned short)(dv[0] >> 16));
      v[j].z += bf2f((unsigned short)(dv[1] & 0xffff)); v[j].w += bf2f((unsigned short)(dv[1] >> 16));
    }
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = wave_sum(s) * (1.0f / d);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < N4; ++j) {
    v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
    q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / d) + 1e-5f);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
  for (int j = 0; j < N4; ++j) {
    const float4 ww = w4[lane + 64 * j], bb = b4[lane + 64 * j];
    const float o0 = v[j].x * rstd * ww.x + bb.x, o1 = v[j].y * rstd * ww.y + bb.y;
    const float o2 = v[j].z * rstd * ww.z + bb.z, o3 = v[j].w * rstd * ww.w + bb.w;
    if (OUT_BF16) {
      reinterpret_cast<u32x2*>(y)[row * (d / 4) + lane + 64 * j] = u32x2{pack2bf(o0, o1), pack2bf(o2, o3)};
    } else {
      reinterpret_cast<float4*>(y)[row * (d / 4) + lane + 64 * j] = make_float4(o0, o1, o2, o3);
    }
  }
}

template <bool OUT_BF16>
static int ln_dispatch(const float* x, long row_stride, const unsigned short* delta, const float* w,
                       const float* b, void* y, long M, int d, hipStream_t s) {
  GWW_REQUIRE(d % 128 == 0 && d >= 128 && d <= 1280, "layernorm: d=%d must be a multiple of 128 <= 1280", d);
  if (M == 0) return GWW_OK;
  dim3 grid((unsigned)cdiv(M, 4)), block(256);
  // (same arithmetic in the same order per ELEMENT; the row sums group four elements instead of two: last-bit differences of
  //  mean / rstd are possible between the two forms, each is deterministic)
  if (d % 256 == 0 && row_stride % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0 &&
      (!delta || (((uintptr_t)delta) & 7) == 0)) {
    switch (d / 256) {
      case 1: hipLaunchKernelGGL((k_layernorm4<1, OUT_BF16>), grid, block, 0, s, x, row_stride, delta, w, b, y, M); break;
      case 2: hipLaunchKernelGGL((k_layernorm4<2, OUT_BF16>), grid, block, 0, s, x, row_stride, delta, w, b, y, M); break;
      case 3: hipLaunchKernelGGL((k_layernorm4<3, OUT_BF16>), grid, block, 0, s, x, row_stride, delta, w, b, y, M); break;
      case 4: hipLaunchKernelGGL((k_layernorm4<4, OUT_BF16>), grid, block, 0, s, x, row_stride, delta, w, b, y, M); break;
      default: hipLaunchKernelGGL((k_layernorm4<5, OUT_BF16>), grid, block, 0, s, x, row_stride, delta, w, b, y, M); break;
    }
    GWW_LAUNCH_CHECK();
    return GWW_OK;
  }
#define GWW_LN_CASE(NV)                                                                          \
  case NV:                                                                                       \
    hipLaunchKernelGGL((k_layernorm<NV, OUT_BF16>), grid, block, 0, s, x, row_stride, delta, w, b, y, M); \
    break;
  switch (d / 128) {
    GWW_LN_CASE(1) GWW_LN_CASE(2) GWW_LN_CASE(3) GWW_LN_CASE(4) GWW_LN_CASE(5)
    GWW_LN_CASE(6) GWW_LN_CASE(7) GWW_LN_CASE(8) GWW_LN_CASE(9) GWW_LN_CASE(10)
  }
#undef GWW_LN_CASE
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

int launch_layernorm(const float* x, const float* w, const float* b, void* y, int out_bf16, long M, int d,
                     hipStream_t s, const void* delta) {
  const unsigned short* dl = (const unsigned short*)delta;
  return out_bf16 ? ln_dispatch<true>(x, d, dl, w, b, y, M, d, s) : ln_dispatch<false>(x, d, dl, w, b, y, M, d, s);
}

// fp32 LayerNorm of M rows spaced row_stride apart (the last-token fast path:
// only row 1499 of every segment is consumed, Signal_vs_Noise/src/model.py:25-26)
int launch_layernorm_rows(const float* x, long row_stride, const float* w, const float* b, float* y,
                          long M, int d, hipStream_t s, const void* delta) {
  return ln_dispatch<false>(x, row_stride, (const unsigned short*)delta, w, b, y, M, d, s);
}

// ---------------------------------------------------------------- casts / packing
__global__ __launch_bounds__(256) void k_cast_f32_bf16(const float* __restrict__ x,
                                                       unsigned short* __restrict__ y, long n) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long step = (long)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += step) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    u32x2 o = {pack2bf(v.x, v.y), pack2bf(v.z, v.w)};
    *reinterpret_cast<u32x2*>(y + i) = o;
  }
  // tail (n not a multiple of 4): the last partial group
  if (i < n && i + 3 >= n) {
    for (long j = i; j < n; ++j) y[j] = f2bf(x[j]);
  }
}

int launch_cast_f32_bf16(const float* x, void* y, long n, hipStream_t s) {
  if (n == 0) return GWW_OK;
  GWW_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0, "cast: pointers must be 16/8-byte aligned");
  long blocks = cdiv(cdiv(n, 4), 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_cast_f32_bf16, dim3((unsigned)blocks), dim3(256), 0, s, x,
                     reinterpret_cast<unsigned short*>(y), n);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// Pack a [N, C, taps] fp32 weight (taps = 1 for linears, 3 for the convs) into the
// kernel layout [N, Kpad] with k = tap * C + c, zero for k >= taps*C, scaled.
template <typename OutT>
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, OutT* __restrict__ out, int N, int C,
                                                 int taps, int Kpad, float scale, int block, int nblocks) {
  const long total = (long)N * Kpad;
  for (long i = (long)block * 256 + threadIdx.x; i < total; i += (long)nblocks * 256) {
    const int n = (int)(i / Kpad), k = (int)(i - (long)n * Kpad);
    float v = 0.f;
    if (k < taps * C) {
      const int tap = k / C, c = k - tap * C;
      v = w[((long)n * C + c) * taps + tap] * scale;
    }
    if constexpr (sizeof(OutT) == 2) {
      out[i] = f2bf(v);
    } else {
      out[i] = v;
    }
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, OutT* __restrict__ out,
                                                     int N, int C, int taps, int Kpad, float scale) {
  pack_weight_body<OutT>(w, out, N, C, taps, Kpad, scale, (int)blockIdx.x, (int)gridDim.x);
}

int launch_pack_weight(const float* w, void* out, int out_bf16, int N, int C, int taps, int Kpad,
                       float scale, hipStream_t s) {
  long blocks = cdiv((long)N * Kpad, 256);
  if (blocks > 4096) blocks = 4096;
  if (out_bf16)
    hipLaunchKernelGGL(k_pack_weight<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, s, w,
                       reinterpret_cast<unsigned short*>(out), N, C, taps, Kpad, scale);
  else
    hipLaunchKernelGGL(k_pack_weight<float>, dim3((unsigned)blocks), dim3(256), 0, s, w,
                       reinterpret_cast<float*>(out), N, C, taps, Kpad, scale);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// out[i] = in ? in[i] * scale : 0
__global__ __launch_bounds__(256) void k_scale_copy(const float* __restrict__ in, float* __restrict__ out,
                                                    int n, float scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in ? in[i] * scale : 0.f;
}

int launch_scale_copy(const float* in, float* out, int n, float scale, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, in, out, n, scale);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- LayerNorm folding
// For the GEMMs that apply LayerNorm algebraically (gemm_astat.hip): per output row n
//   W'[n][k] = bf16(scale * g[k] * W[n][k]),  u[n] = sum_k W'[n][k],
//   cb[n]    = scale * (bias[n] + sum_k b_ln[k] * W[n][k])
__device__ __forceinline__ void ln_fold_body(const float* __restrict__ w, const float* __restrict__ g,
                                             const float* __restrict__ bl, const float* __restrict__ bias,
                                             float scale, int K, unsigned short* __restrict__ wp,
                                             float* __restrict__ u, float* __restrict__ cb, int n, float* red) {
  float su = 0.f, sc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float wv = w[(long)n * K + k];
    const unsigned short q = f2bf(scale * g[k] * wv);
    wp[(long)n * K + k] = q;
    su += bf2f(q);
    sc = fmaf(bl[k], wv, sc);
  }
  su = wave_sum(su);
  sc = wave_sum(sc);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = su;
    red[4 + (threadIdx.x >> 6)] = sc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u[n] = red[0] + red[1] + red[2] + red[3];
    cb[n] = scale * ((bias ? bias[n] : 0.f) + red[4] + red[5] + red[6] + red[7]);
  }
}

__global__ __launch_bounds__(256) void k_ln_fold(const float* __restrict__ w, const float* __restrict__ g,
                                                 const float* __restrict__ bl, const float* __restrict__ bias,
                                                 float scale, int K, unsigned short* __restrict__ wp,
                                                 float* __restrict__ u, float* __restrict__ cb) {
  __shared__ float red[8];
  ln_fold_body(w, g, bl, bias, scale, K, wp, u, cb, (int)blockIdx.x, red);
}

int launch_ln_fold(const float* w, const float* g, const float* bl, const float* bias, float scale, int N, int K,
                   void* wp, float* u, float* cb, hipStream_t s) {
  hipLaunchKernelGGL(k_ln_fold, dim3((unsigned)N), dim3(256), 0, s, w, g, bl, bias, scale, K,
                     reinterpret_cast<unsigned short*>(wp), u, cb);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- DoRA merge (K11)
// peft 0.12.0 tuners/lora/dora.py: W' = W0 + s B A ; n = ||W'||_2 per output row ;
// W_eff = (m / n)[:, None] * W'.  One workgroup per output row; fp32.
__device__ __forceinline__ void dora_merge_body(const float* __restrict__ w0, const float* __restrict__ a,
                                                const float* __restrict__ b, const float* __restrict__ m,
                                                float scaling, int d_in, int r, float* __restrict__ w_eff,
                                                float* __restrict__ norm_out, int row, float* sm) {
  float* brow = sm;            // [r] scaled B row, then [4] partials
  float* red = sm + r;
  for (int j = threadIdx.x; j < r; j += 256) brow[j] = scaling * b[(long)row * r + j];
  __syncthreads();
  float ss = 0.f;
  for (int c = threadIdx.x; c < d_in; c += 256) {
    float v = w0[(long)row * d_in + c];
    for (int j = 0; j < r; ++j) v = fmaf(brow[j], a[(long)j * d_in + c], v);
    w_eff[(long)row * d_in + c] = v;       // W' for now; rescaled below
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  const float g = m[row] / nrm;
  if (threadIdx.x == 0 && norm_out) norm_out[row] = nrm;
  for (int c = threadIdx.x; c < d_in; c += 256) w_eff[(long)row * d_in + c] *= g;   // same thread wrote it
}

__global__ __launch_bounds__(256) void k_dora_merge(const float* __restrict__ w0, const float* __restrict__ a,
                                                    const float* __restrict__ b, const float* __restrict__ m,
                                                    float scaling, int d_in, int r, float* __restrict__ w_eff,
                                                    float* __restrict__ norm_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  dora_merge_body(w0, a, b, m, scaling, d_in, r, w_eff, norm_out, (int)blockIdx.x, sm);
}

// ---------------------------------------------------------------- batched weight preparation
// A DoRA optimizer step changes 12 (tiny) .. 36 (small) projection weights, and every one of them used to cost a merge, six
// panel packs, eight bias / gain copies, three LayerNorm folds and a transpose: ~110 launches of 3 - 6 us each per step, a
// twentieth of the whisper-tiny step.  The same bodies run here as ONE launch per dependency phase: the host collects
// descriptors (kernel argument, no device table to copy), a workgroup finds its op by a scalar walk over the prefix table.
__global__ __launch_bounds__(256) void k_prep_batch(const PrepArgs P) {
  __shared__ __attribute__((aligned(16))) float sm[kPrepMaxRank + 8];
  __shared__ unsigned short tile[64][66];
  int id = 0;
  while (id + 1 < P.n && (int)blockIdx.x >= P.first[id + 1]) ++id;
  const PrepOp& o = P.op[id];
  const int lb = (int)blockIdx.x - P.first[id], nb = P.first[id + 1] - P.first[id];
  switch (o.kind) {
    case PREP_PACK16:
      pack_weight_body<unsigned short>((const float*)o.a, (unsigned short*)o.o0, o.N, o.C, o.taps, o.K, o.scale, lb, nb);
      break;
    case PREP_PACK32:
      pack_weight_body<float>((const float*)o.a, (float*)o.o0, o.N, o.C, o.taps, o.K, o.scale, lb, nb);
      break;
    case PREP_COPY: {
      const int i = lb * 256 + threadIdx.x;
      if (i < o.N) ((float*)o.o0)[i] = o.a ? ((const float*)o.a)[i] * o.scale : 0.f;
      break;
    }
    case PREP_LNFOLD:
      ln_fold_body((const float*)o.a, (const float*)o.b, (const float*)o.c, (const float*)o.d, o.scale, o.K,
                   (unsigned short*)o.o0, (float*)o.o1, (float*)o.o2, lb, sm);
      break;
    case PREP_TRANSPOSE: {
      const int nbx = (o.K + 63) / 64;
      const int r0 = (lb / nbx) * 64, c0 = (lb % nbx) * 64, R = o.N, Cn = o.K;
      const unsigned short* in = (const unsigned short*)o.a;
      unsigned short* out = (unsigned short*)o.o0;
      for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < R && c0 + c < Cn) ? in[(long)(r0 + r) * Cn + c0 + c] : (unsigned short)0;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < R && c0 + c < Cn) out[(long)(c0 + c) * R + r0 + r] = tile[r][c];
      }
      break;
    }
    case PREP_DORA:
      dora_merge_body((const float*)o.a, (const float*)o.b, (const float*)o.c, (const float*)o.d, o.scale, o.K, o.C,
                      (float*)o.o0, (float*)o.o1, lb, sm);
      break;
    default: break;
  }
}

int launch_prep_batch(const PrepArgs& P, hipStream_t s) {
  if (P.n == 0 || P.first[P.n] == 0) return GWW_OK;
  hipLaunchKernelGGL(k_prep_batch, dim3((unsigned)P.first[P.n]), dim3(256), 0, s, P);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

int PrepBatch::add(const PrepOp& o) {
  if (o.blocks <= 0) return GWW_OK;
  if (args.empty() || args.back().n == kPrepMaxOps) {
    args.emplace_back();
    args.back().n = 0;
    args.back().first[0] = 0;
  }
  PrepArgs& t = args.back();
  t.op[t.n] = o;
  t.first[t.n + 1] = t.first[t.n] + o.blocks;
  ++t.n;
  return GWW_OK;
}

int PrepBatch::flush() {
  for (const PrepArgs& t : args) GWW_TRY(launch_prep_batch(t, s));
  args.clear();
  return GWW_OK;
}

int PrepBatch::pack(const float* w, void* out, int out_bf16, int N, int C, int taps, int Kpad, float scale) {
  PrepOp o{};
  o.kind = out_bf16 ? PREP_PACK16 : PREP_PACK32;
  o.a = w; o.o0 = out; o.N = N; o.C = C; o.taps = taps; o.K = Kpad; o.scale = scale;
  long b = cdiv((long)N * Kpad, 256);
  o.blocks = (int)(b > 1024 ? 1024 : b);
  return add(o);
}

int PrepBatch::copy(const float* in, float* out, int n, float scale) {
  PrepOp o{};
  o.kind = PREP_COPY; o.a = in; o.o0 = out; o.N = n; o.scale = scale; o.blocks = (int)cdiv(n, 256);
  return add(o);
}

int PrepBatch::ln_fold(const float* w, const float* g, const float* bl, const float* bias, float scale, int N, int K,
                       void* wp, float* u, float* cb) {
  PrepOp o{};
  o.kind = PREP_LNFOLD; o.a = w; o.b = g; o.c = bl; o.d = bias; o.scale = scale; o.N = N; o.K = K;
  o.o0 = wp; o.o1 = u; o.o2 = cb; o.blocks = N;
  return add(o);
}

int PrepBatch::transpose(const void* in, void* out, int R, int Cn) {
  PrepOp o{};
  o.kind = PREP_TRANSPOSE; o.a = in; o.o0 = out; o.N = R; o.K = Cn; o.blocks = (int)(cdiv(R, 64) * cdiv(Cn, 64));
  return add(o);
}

int PrepBatch::dora(const float* w0, const float* a, const float* b, const float* m, float scaling, int d_out, int d_in,
                    int r, float* w_eff, float* norm_out) {
  PrepOp o{};
  o.kind = PREP_DORA; o.a = w0; o.b = a; o.c = b; o.d = m; o.scale = scaling; o.N = d_out; o.K = d_in; o.C = r;
  o.o0 = w_eff; o.o1 = norm_out; o.blocks = d_out;
  return add(o);
}

}  // namespace gww

using namespace gww;

extern "C" int gww_dora_merge_f32(const float* w0, const float* a, const float* b, const float* m,
                                  float scaling, int d_out, int d_in, int r, float* w_eff,
                                  float* norm_out, void* stream) {
  GWW_REQUIRE(w0 && a && b && m && w_eff, "gww_dora_merge_f32: NULL argument");
  GWW_REQUIRE(d_out > 0 && d_in > 0 && r > 0 && r <= 1024, "gww_dora_merge_f32: bad shape %dx%d r=%d", d_out, d_in, r);
  hipLaunchKernelGGL(k_dora_merge, dim3((unsigned)d_out), dim3(256), (r + 4) * sizeof(float),
                     (hipStream_t)stream, w0, a, b, m, scaling, d_in, r, w_eff, norm_out);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

extern "C" int gww_dora_merge_batch_f32(const gww_dora_merge_item* items, int n, void* stream) {
  GWW_REQUIRE(items || n == 0, "gww_dora_merge_batch_f32: NULL items");
  GWW_REQUIRE(n >= 0, "gww_dora_merge_batch_f32: n < 0");
  PrepBatch pb((hipStream_t)stream);
  for (int i = 0; i < n; ++i) {
    const gww_dora_merge_item& t = items[i];
    GWW_REQUIRE(t.w0 && t.a && t.b && t.m && t.w_eff, "gww_dora_merge_batch_f32: NULL pointer in item %d", i);
    GWW_REQUIRE(t.d_out > 0 && t.d_in > 0 && t.r > 0 && t.r <= kPrepMaxRank,
                "gww_dora_merge_batch_f32: bad shape %dx%d r=%d in item %d", t.d_out, t.d_in, t.r, i);
    GWW_TRY(pb.dora(t.w0, t.a, t.b, t.m, t.scaling, t.d_out, t.d_in, t.r, t.w_eff, t.norm_out));
  }
  return pb.flush();
}

extern "C" int gww_ln_fold_weights(const float* w, const float* ln_w, const float* ln_b, const float* bias, float scale,
                                   int N, int K, void* w_folded_bf16, float* u, float* cb, void* stream) {
  GWW_REQUIRE(w && ln_w && ln_b && w_folded_bf16 && u && cb, "gww_ln_fold_weights: NULL argument");
  GWW_REQUIRE(N > 0 && K > 0, "gww_ln_fold_weights: bad shape");
  return launch_ln_fold(w, ln_w, ln_b, bias, scale, N, K, w_folded_bf16, u, cb, (hipStream_t)stream);
}

extern "C" int gww_layernorm(const float* x, const float* w, const float* b, void* y, int out_bf16, long M,
                             int d, void* stream) {
  GWW_REQUIRE(x && w && b && y, "gww_layernorm: NULL argument");
  GWW_REQUIRE(M >= 0, "gww_layernorm: M < 0");
  return launch_layernorm(x, w, b, y, out_bf16, M, d, (hipStream_t)stream);
}

extern "C" int gww_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
  GWW_REQUIRE(x && y && n >= 0, "gww_cast_f32_bf16: bad argument");
  return launch_cast_f32_bf16(x, y, n, (hipStream_t)stream);
}

extern "C" int gww_version(void) { return GWW_VERSION; }
extern "C" const char* gww_last_error(void) { return g_err; }
