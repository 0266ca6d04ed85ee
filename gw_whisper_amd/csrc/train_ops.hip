// Row-wise / elementwise kernels of the DoRA training step (backward of LayerNorm and GELU,
// bf16 transposes for the dX GEMMs, and the rank-r DoRA parameter gradients).
//
// Reference semantics: torch autograd through HF:modeling_whisper.py:379-413 with peft 0.12.0
// DoRA Linear (tuners/lora/dora.py: the weight norm is DETACHED), i.e. with g = m / ||W'||:
//     y  = g * (W' x) + b,   W' = W0 + s B A
//     dm = sum_rows dy * (W' x) / ||W'|| = sum_rows dy * (y - b) / m
//     dB = s * (g * dy)^T (x A^T)            [out, r]
//     dA = s * ((g * dy) B)^T x              [r, in]
#include "common.h"

#include <stdlib.h>

namespace gww {

// ---------------------------------------------------------------- LayerNorm backward
// dx (+)= rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  one wave per row, NV float2 per lane.
// DY_F32: dy is fp32 (grad of last_hidden_state) else bf16 (grad from a dX GEMM).
template <int NV, bool DY_F32, bool ACCUM>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                const void* __restrict__ dy, float* dx,
                                                unsigned short* __restrict__ dx_bf16, long M) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int d = NV * 128;
  const float2* xr = reinterpret_cast<const float2*>(x + row * d);
  const float2* g2 = reinterpret_cast<const float2*>(gamma);
  float2 v[NV], gy[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = xr[lane + 64 * j];
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
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float d0, d1;
    if constexpr (DY_F32) {
      const float2 t = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(dy) + row * d)[lane + 64 * j];
      d0 = t.x; d1 = t.y;
    } else {
      const unsigned int t = reinterpret_cast<const unsigned int*>(reinterpret_cast<const unsigned short*>(dy) + row * d)[lane + 64 * j];
      d0 = bf2f((unsigned short)(t & 0xffff)); d1 = bf2f((unsigned short)(t >> 16));
    }
    const float2 gg = g2[lane + 64 * j];
    gy[j].x = d0 * gg.x; gy[j].y = d1 * gg.y;
    v[j].x *= rstd; v[j].y *= rstd;                 // xhat
    a += gy[j].x + gy[j].y;
    b += gy[j].x * v[j].x + gy[j].y * v[j].y;
  }
  a = wave_sum(a) * (1.0f / d);
  b = wave_sum(b) * (1.0f / d);
  float2* dxr = reinterpret_cast<float2*>(dx + row * d);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float o0 = rstd * (gy[j].x - a - v[j].x * b), o1 = rstd * (gy[j].y - a - v[j].y * b);
    if constexpr (ACCUM) {
      const float2 old = dxr[lane + 64 * j];
      o0 += old.x; o1 += old.y;
    }
    dxr[lane + 64 * j] = make_float2(o0, o1);
    if (dx_bf16) reinterpret_cast<unsigned int*>(dx_bf16 + row * d)[lane + 64 * j] = pack2bf(o0, o1);
  }
}

int launch_ln_bwd(const float* x, const float* gamma, const void* dy, int dy_f32, float* dx, int accumulate,
                  void* dx_bf16, long M, int d, hipStream_t s) {
  GWW_REQUIRE(d % 128 == 0 && d >= 128 && d <= 1280, "ln_bwd: d=%d must be a multiple of 128 <= 1280", d);
  if (M == 0) return GWW_OK;
  dim3 grid((unsigned)cdiv(M, 4)), block(256);
  unsigned short* db = (unsigned short*)dx_bf16;
#define GWW_LNB(NV)                                                                                               \
  case NV:                                                                                                        \
    if (dy_f32 && accumulate) hipLaunchKernelGGL((k_ln_bwd<NV, true, true>), grid, block, 0, s, x, gamma, dy, dx, db, M);   \
    else if (dy_f32) hipLaunchKernelGGL((k_ln_bwd<NV, true, false>), grid, block, 0, s, x, gamma, dy, dx, db, M);          \
    else if (accumulate) hipLaunchKernelGGL((k_ln_bwd<NV, false, true>), grid, block, 0, s, x, gamma, dy, dx, db, M);      \
    else hipLaunchKernelGGL((k_ln_bwd<NV, false, false>), grid, block, 0, s, x, gamma, dy, dx, db, M);                     \
    break;
  switch (d / 128) {
    GWW_LNB(1) GWW_LNB(2) GWW_LNB(3) GWW_LNB(4) GWW_LNB(5) GWW_LNB(6) GWW_LNB(7) GWW_LNB(8) GWW_LNB(9) GWW_LNB(10)
  }
#undef GWW_LNB
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- GELU forward / backward (bf16, 8 per lane)
template <bool BWD>
__global__ __launch_bounds__(256) void k_gelu_bf16(const unsigned short* __restrict__ z,
                                                   const unsigned short* __restrict__ df, unsigned short* out,
                                                   long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const u32x4 zv = reinterpret_cast<const u32x4*>(z)[i];
    u32x4 dv = {0u, 0u, 0u, 0u};
    if constexpr (BWD) dv = reinterpret_cast<const u32x4*>(df)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z0 = bf2f((unsigned short)(zv[j] & 0xffff)), z1 = bf2f((unsigned short)(zv[j] >> 16));
      float r0, r1;
      if constexpr (BWD) {
        // gelu'(z) = Phi(z) + z phi(z)
        const float p0 = dgelu_fast(z0);
        const float p1 = dgelu_fast(z1);
        r0 = bf2f((unsigned short)(dv[j] & 0xffff)) * p0;
        r1 = bf2f((unsigned short)(dv[j] >> 16)) * p1;
      } else {
        r0 = gelu_fast(z0);
        r1 = gelu_fast(z1);
      }
      o[j] = pack2bf(r0, r1);
    }
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
}

int launch_gelu_bf16(const void* z, const void* df, void* out, long n, hipStream_t s) {
  GWW_REQUIRE(n % 8 == 0, "gelu_bf16: n must be a multiple of 8");
  if (n == 0) return GWW_OK;
  long blocks = cdiv(n / 8, 256);
  if (blocks > 8192) blocks = 8192;
  if (df)
    hipLaunchKernelGGL(k_gelu_bf16<true>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)z,
                       (const unsigned short*)df, (unsigned short*)out, n / 8);
  else
    hipLaunchKernelGGL(k_gelu_bf16<false>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)z,
                       (const unsigned short*)nullptr, (unsigned short*)out, n / 8);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- bf16 transpose [R][C] -> [C][R]
__global__ __launch_bounds__(256) void k_transpose_bf16(const unsigned short* __restrict__ in,
                                                        unsigned short* __restrict__ out, int R, int Cn) {
  __shared__ unsigned short tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < Cn) ? in[(long)(r0 + r) * Cn + c0 + c] : (unsigned short)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < Cn) out[(long)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

int launch_transpose_bf16(const void* in, void* out, int R, int Cn, hipStream_t s) {
  dim3 grid((unsigned)cdiv(Cn, 64), (unsigned)cdiv(R, 64));
  hipLaunchKernelGGL(k_transpose_bf16, grid, dim3(256), 0, s, (const unsigned short*)in, (unsigned short*)out, R, Cn);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- conv stem backward (gradient w.r.t. the mel input)
// HF:modeling_whisper.py:618-619: x0 = gelu(conv2(gelu(conv1(mel)))) + pos.  Both convolutions run as GEMMs over
// overlapping rows of token-major padded buffers (encoder.hip), so their input gradients are a GEMM against
// the transposed panel ("col" rows = taps side by side) followed by a gather over the taps -- no atomics.
__device__ __forceinline__ float gelu_grad(float z) {   // Phi(z) + z phi(z)
  return dgelu_fast(z);
}

// dz2[b (T+1) + t] = t < T ? bf16(dx0[b T + t]) * gelu'(z2[b (T+1) + t]) : 0        (8 columns per thread)
__global__ __launch_bounds__(256) void k_stem_dz2(const unsigned short* __restrict__ dxb,
                                                  const unsigned short* __restrict__ z2,
                                                  unsigned short* __restrict__ out, int T, int d8, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long row = i / d8;
    const int c8 = (int)(i - row * d8);
    const long b = row / (T + 1);
    const int t = (int)(row - b * (T + 1));
    u32x4 o = {0u, 0u, 0u, 0u};
    if (t < T) {
      const u32x4 g = reinterpret_cast<const u32x4*>(dxb)[(b * T + t) * d8 + c8];
      const u32x4 z = reinterpret_cast<const u32x4*>(z2)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack2bf(bf2f((unsigned short)(g[j] & 0xffff)) * gelu_grad(bf2f((unsigned short)(z[j] & 0xffff))),
                       bf2f((unsigned short)(g[j] >> 16)) * gelu_grad(bf2f((unsigned short)(z[j] >> 16))));
    }
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
}

// conv2 (stride 2, taps k = 0..2 read padded row p = 2 t + k).  col [B (T+1), 3 d]: col[b, t][k d + ci] =
// sum_co dz2[b, t, co] W2[co, ci, k].  Gather into the conv1 output row t1 (padded row p = t1 + 1), times
// gelu'(z1):   p even: col[b, p/2][ci] (p/2 < T) + col[b, p/2 - 1][2 d + ci];   p odd: col[b, (p-1)/2][d + ci].
// z1 / out rows: b (Tin + 2) + t1 (the unshifted conv1 GEMM rows); rows t1 >= Tin are zeroed.  In place (out == z1) ok.
__global__ __launch_bounds__(256) void k_stem_dz1(const unsigned short* __restrict__ col,
                                                  const unsigned short* z1, unsigned short* out, int T, int Tin,
                                                  int d, long n8) {
  const int d8 = d / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long row = i / d8;
    const int c8 = (int)(i - row * d8);
    const long b = row / (Tin + 2);
    const int t1 = (int)(row - b * (Tin + 2));
    u32x4 o = {0u, 0u, 0u, 0u};
    if (t1 < Tin) {
      const int p = t1 + 1;
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = 0.f;
      auto add = [&](long crow, int tap) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(col + (crow * 3 + tap) * d + 8 * c8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[2 * j] += bf2f((unsigned short)(v[j] & 0xffff));
          g[2 * j + 1] += bf2f((unsigned short)(v[j] >> 16));
        }
      };
      if (p & 1) {
        add(b * (T + 1) + (p - 1) / 2, 1);
      } else {
        if (p / 2 < T) add(b * (T + 1) + p / 2, 0);
        add(b * (T + 1) + p / 2 - 1, 2);
      }
      const u32x4 z = reinterpret_cast<const u32x4*>(z1)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack2bf(g[2 * j] * gelu_grad(bf2f((unsigned short)(z[j] & 0xffff))),
                       g[2 * j + 1] * gelu_grad(bf2f((unsigned short)(z[j] >> 16))));
    }
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
}

// conv1 (taps k read padded mel row p = t1 + k = time + 1).  col1 [B (Tin+2), Kp]: col1[b, t1][k C + c] =
// sum_co dz1[b, t1, co] W1[co, c, k].   dmel[b, c, tau] = sum_k col1[b, tau + 1 - k][k C + c], 0 <= tau + 1 - k < Tin.
// One workgroup = 64 time steps of one segment, staged through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void k_stem_dmel(const unsigned short* __restrict__ col1, float* __restrict__ dmel,
                                                   int Tin, int C, int Kp) {
  constexpr int TT = 64;
  extern __shared__ unsigned short sm[];             // [TT + 2][3 C + 2]
  const int ld = 3 * C + 2;
  const int b = blockIdx.y, tau0 = blockIdx.x * TT;
  for (int i = threadIdx.x; i < (TT + 2) * 3 * C; i += 256) {
    const int rl = i / (3 * C), k = i - rl * 3 * C;
    const int t1 = tau0 - 1 + rl;
    sm[rl * ld + k] = (t1 >= 0 && t1 < Tin) ? col1[((long)b * (Tin + 2) + t1) * Kp + k] : (unsigned short)0;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C * TT; o += 256) {
    const int c = o / TT, tl = o - c * TT;
    const int tau = tau0 + tl;
    if (tau >= Tin) continue;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) v += bf2f(sm[(tl + 2 - k) * ld + k * C + c]);   // row tau + 1 - k  ->  local tl + 2 - k
    dmel[((long)b * C + c) * Tin + tau] = v;
  }
}

int launch_stem_dz2(const void* dxb, const void* z2, void* out, int B, int T, int d, hipStream_t s) {
  GWW_REQUIRE(d % 8 == 0, "stem_dz2: d must be a multiple of 8");
  const long n8 = (long)B * (T + 1) * (d / 8);
  long blocks = cdiv(n8, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_stem_dz2, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)dxb,
                     (const unsigned short*)z2, (unsigned short*)out, T, d / 8, n8);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

int launch_stem_dz1(const void* col, const void* z1, void* out, int B, int T, int Tin, int d, hipStream_t s) {
  GWW_REQUIRE(d % 8 == 0 && Tin == 2 * T, "stem_dz1: bad shape");
  const long n8 = (long)B * (Tin + 2) * (d / 8);
  long blocks = cdiv(n8, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_stem_dz1, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)col,
                     (const unsigned short*)z1, (unsigned short*)out, T, Tin, d, n8);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

int launch_stem_dmel(const void* col1, float* dmel, int B, int Tin, int C, int Kp, hipStream_t s) {
  GWW_REQUIRE(3 * C <= Kp, "stem_dmel: col1 row shorter than 3 taps");
  const size_t lds = (size_t)(64 + 2) * (3 * C + 2) * 2;
  hipLaunchKernelGGL(k_stem_dmel, dim3((unsigned)cdiv(Tin, 64), (unsigned)B), dim3(256), lds, s,
                     (const unsigned short*)col1, dmel, Tin, C, Kp);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- bf16(a - b) of two fp32 tensors
// out_proj's output is never stored on its own (it is fused into the residual add): its DoRA magnitude
// gradient needs y = x_mid - x_in, rebuilt here.
__global__ __launch_bounds__(256) void k_sub_f32_bf16(const float* __restrict__ a, const float* __restrict__ b,
                                                      unsigned short* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
    const f32x4 bv = reinterpret_cast<const f32x4*>(b)[i];
    u32x2 o;
    o[0] = pack2bf(av[0] - bv[0], av[1] - bv[1]);
    o[1] = pack2bf(av[2] - bv[2], av[3] - bv[3]);
    reinterpret_cast<u32x2*>(out)[i] = o;
  }
}

// out = x + delta (fp32 residual stream + bf16 deferred delta): x_in[L] of the fused training forward
__global__ __launch_bounds__(256) void k_add_delta_f32(const float* __restrict__ x, const unsigned short* __restrict__ dl,
                                                       float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const u32x2 dv = reinterpret_cast<const u32x2*>(dl)[i];
    v[0] += bf2f((unsigned short)(dv[0] & 0xffff));
    v[1] += bf2f((unsigned short)(dv[0] >> 16));
    v[2] += bf2f((unsigned short)(dv[1] & 0xffff));
    v[3] += bf2f((unsigned short)(dv[1] >> 16));
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

int launch_add_delta_f32(const float* x, const void* delta_bf16, float* out, long n, hipStream_t s) {
  GWW_REQUIRE(n % 4 == 0, "add_delta_f32: n must be a multiple of 4");
  if (n == 0) return GWW_OK;
  long blocks = cdiv(n / 4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_add_delta_f32, dim3((unsigned)blocks), dim3(256), 0, s, x, (const unsigned short*)delta_bf16, out, n / 4);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

int launch_sub_f32_bf16(const float* a, const float* b, void* out, long n, hipStream_t s) {
  GWW_REQUIRE(n % 4 == 0, "sub_f32_bf16: n must be a multiple of 4");
  if (n == 0) return GWW_OK;
  long blocks = cdiv(n / 4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_sub_f32_bf16, dim3((unsigned)blocks), dim3(256), 0, s, a, b, (unsigned short*)out, n / 4);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- DoRA parameter gradients
// One [d, d] target (q / k / v / out projection).  X [M, d] bf16 (row stride ldx), dY / Y [M, d] bf16
// sections with row stride ldy (e.g. inside dqkv / qkv), bias_st [d] in STORED units, yscale = dy_true /
// dy_stored (1/8 for the pre-scaled q section).  Accumulates (atomicAdd) into dA [r, d], dB [d, r], dm [d]:
// the caller zeroes them once per step.  Each workgroup walks row tiles of 32 and keeps its partial sums
// in registers; one pass of atomics per workgroup at the end.
template <int D, int R>
__global__ __launch_bounds__(256) void k_dora_grads(const unsigned short* __restrict__ X, long ldx,
                                                    const unsigned short* __restrict__ dY,
                                                    const unsigned short* __restrict__ Y, long ldy,
                                                    const float* __restrict__ bias_st, float yscale, float scaling,
                                                    const float* __restrict__ A, const float* __restrict__ Bm,
                                                    const float* __restrict__ mag, const float* __restrict__ nrm,
                                                    float* dA, float* dB, float* dm, long M) {
  constexpr int RT = D <= 512 ? 32 : (D <= 1024 ? 16 : 8);   // rows per tile (two fp32 tiles must fit in LDS)
  constexpr int PER = D * R / 256;             // dA / dB outputs per thread
  constexpr int CC = (D + 255) / 256;          // dm columns per thread
  static_assert(D * R % 256 == 0 && D % 8 == 0 && RT * R <= 256, "shape");
  __shared__ float xs[RT][D + 1];
  __shared__ float gs[RT][D + 1];              // g * dy_true
  __shared__ float us[RT][R];                  // x A^T
  __shared__ float ws[RT][R];                  // (g dy) B
  const int tid = threadIdx.x;
  float accA[PER], accB[PER], accM[CC];
#pragma unroll
  for (int i = 0; i < CC; ++i) accM[i] = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) accA[i] = accB[i] = 0.f;

  for (long r0 = (long)blockIdx.x * RT; r0 < M; r0 += (long)gridDim.x * RT) {
    __syncthreads();
    // stage the tile (8 bf16 per load); fold g and yscale into dy; per-column dm partials
    for (int i = tid; i < RT * D / 8; i += 256) {
      const int rr = i / (D / 8), c8 = (i - rr * (D / 8)) * 8;
      const long row = r0 + rr;
      u32x4 xv = {0u, 0u, 0u, 0u}, dv = {0u, 0u, 0u, 0u};
      if (row < M) {
        xv = *reinterpret_cast<const u32x4*>(X + row * ldx + c8);
        dv = *reinterpret_cast<const u32x4*>(dY + row * ldy + c8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xs[rr][c8 + 2 * j] = bf2f((unsigned short)(xv[j] & 0xffff));
        xs[rr][c8 + 2 * j + 1] = bf2f((unsigned short)(xv[j] >> 16));
        const int c = c8 + 2 * j;
        gs[rr][c] = bf2f((unsigned short)(dv[j] & 0xffff)) * yscale * (mag[c] / nrm[c]);
        gs[rr][c + 1] = bf2f((unsigned short)(dv[j] >> 16)) * yscale * (mag[c + 1] / nrm[c + 1]);
      }
    }
    // dm: thread -> columns tid, tid + 256, ...
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      const int c = tid + 256 * cc;
      if (c < D) {
        float sacc = 0.f;
        for (int rr = 0; rr < RT; ++rr) {
          const long row = r0 + rr;
          if (row < M) {
            const float dyv = bf2f(dY[row * ldy + c]);
            const float yv = bf2f(Y[row * ldy + c]);
            sacc = fmaf(dyv, yv - bias_st[c], sacc);
          }
        }
        accM[cc] += sacc;
      }
    }
    __syncthreads();
    // u = x A^T, w = (g dy) B : RT * R outputs each, one (row, j) per thread (RT * R <= 256)
    if (tid < RT * R) {
      const int rr = tid / R, j = tid - rr * R;
      float su = 0.f, sw = 0.f;
      for (int k = 0; k < D; ++k) {
        su = fmaf(xs[rr][k], A[(long)j * D + k], su);
        sw = fmaf(gs[rr][k], Bm[(long)k * R + j], sw);
      }
      us[rr][j] = su;
      ws[rr][j] = sw;
    }
    __syncthreads();
    // dB[n][j] += sum_rows gs[row][n] * us[row][j] ; dA[j][k] += sum_rows ws[row][j] * xs[row][k]
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int o = tid + 256 * i;            // flat index into [D][R] (dB) and [R][D] (dA)
      const int nB = o / R, jB = o - nB * R;
      const int jA = o / D, kA = o - jA * D;
      float sb = 0.f, sa = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < RT; ++rr) {
        sb = fmaf(gs[rr][nB], us[rr][jB], sb);
        sa = fmaf(ws[rr][jA], xs[rr][kA], sa);
      }
      accB[i] += sb;
      accA[i] += sa;
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int o = tid + 256 * i;
    atomicAdd(dB + o, scaling * accB[i]);
    atomicAdd(dA + o, scaling * accA[i]);
  }
#pragma unroll
  for (int cc = 0; cc < CC; ++cc) {
    const int c = tid + 256 * cc;
    if (c < D) atomicAdd(dm + c, accM[cc] / mag[c]);
  }
}

// Register-blocked form for d <= 768 (whisper-tiny / base / small): same contract as k_dora_grads.
//   * A and B^T are staged once per workgroup in LDS (rows padded by 4 floats: conflict-free float4 reads),
//   * u = x A^T and w = (g dy) B: one 384..768-long dot product per thread, float4 LDS reads of both operands,
//   * dB / dA: a thread owns whole columns (c, all 8 ranks): per row one read of gs / xs and two broadcast float4
//     of u / w feed 16 FMAs (the first kernel re-read two LDS words per FMA and re-loaded A / B from L1 per k),
//   * dm rides in the same column pass.
template <int D>
__global__ __launch_bounds__(256) void k_dora_grads_rb(const unsigned short* __restrict__ X, long ldx,
                                                       const unsigned short* __restrict__ dY,
                                                       const unsigned short* __restrict__ Y, long ldy,
                                                       const float* __restrict__ bias_st, float yscale, float scaling,
                                                       const float* __restrict__ A, const float* __restrict__ Bm,
                                                       const float* __restrict__ mag, const float* __restrict__ nrm,
                                                       float* dA, float* dB, float* dm, long M) {
  constexpr int R = 8, RT = D <= 512 ? 16 : 8, LD = D + 4;   // rows per tile: everything must fit in 160 KB of LDS
  constexpr int CC = (D + 255) / 256;
  constexpr int NCH = RT * D / 8 / 256;      // 16-byte chunks of each operand per thread per tile
  static_assert(D % 8 == 0 && D <= 768 && (RT * D / 8) % 256 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) float sm_dg[];
  float* xs = sm_dg;               // [RT][LD]
  float* gs = xs + RT * LD;        // [RT][LD]   g * dy_true
  float* At = gs + RT * LD;        // [R][LD]    A
  float* Bt = At + R * LD;         // [R][LD]    B^T
  float* us = Bt + R * LD;         // [RT][R]
  float* ws = us + RT * R;         // [RT][R]
  float* gl = ws + RT * R;         // [D]        yscale * mag / nrm
  unsigned short* ys = reinterpret_cast<unsigned short*>(gl + D);   // [RT][D] bf16 (the dm term)
  const int tid = threadIdx.x;
  for (int i = tid; i < R * D; i += 256) {
    const int j = i / D, k = i - j * D;
    At[j * LD + k] = A[i];
    Bt[j * LD + k] = Bm[(long)k * R + j];
  }
  for (int i = tid; i < D; i += 256) gl[i] = yscale * (mag[i] / nrm[i]);
  float accA[CC][R], accB[CC][R], accM[CC], bcol[CC];
#pragma unroll
  for (int cc = 0; cc < CC; ++cc) {
    accM[cc] = 0.f;
    const int c = tid + 256 * cc;
    bcol[cc] = c < D ? bias_st[c] : 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) accA[cc][j] = accB[cc][j] = 0.f;
  }
  const int which = tid / (RT * R), prr = (tid % (RT * R)) >> 3, pj = tid & 7;   // phase-3 role: (u | w, row, rank)

  // the next tile's X / dY / Y chunks are in flight while the current tile is being reduced
  u32x4 px[NCH], pd[NCH], py[NCH];
  auto prefetch = [&](long r0) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int i = tid + 256 * q;
      const int rr = i / (D / 8), c8 = (i - rr * (D / 8)) * 8;
      const long row = r0 + rr;
      px[q] = pd[q] = py[q] = u32x4{0u, 0u, 0u, 0u};
      if (row < M) {
        px[q] = *reinterpret_cast<const u32x4*>(X + row * ldx + c8);
        pd[q] = *reinterpret_cast<const u32x4*>(dY + row * ldy + c8);
        py[q] = *reinterpret_cast<const u32x4*>(Y + row * ldy + c8);
      }
    }
  };
  const long step = (long)gridDim.x * RT;
  long r0 = (long)blockIdx.x * RT;
  if (r0 < M) prefetch(r0);
  for (; r0 < M; r0 += step) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int i = tid + 256 * q;
      const int rr = i / (D / 8), c8 = (i - rr * (D / 8)) * 8;
      const u32x4 xv = px[q], dv = pd[q];
      f32x4 x0, x1, g0, g1;
      x0[0] = bf2f((unsigned short)(xv[0] & 0xffff)); x0[1] = bf2f((unsigned short)(xv[0] >> 16));
      x0[2] = bf2f((unsigned short)(xv[1] & 0xffff)); x0[3] = bf2f((unsigned short)(xv[1] >> 16));
      x1[0] = bf2f((unsigned short)(xv[2] & 0xffff)); x1[1] = bf2f((unsigned short)(xv[2] >> 16));
      x1[2] = bf2f((unsigned short)(xv[3] & 0xffff)); x1[3] = bf2f((unsigned short)(xv[3] >> 16));
      g0[0] = bf2f((unsigned short)(dv[0] & 0xffff)); g0[1] = bf2f((unsigned short)(dv[0] >> 16));
      g0[2] = bf2f((unsigned short)(dv[1] & 0xffff)); g0[3] = bf2f((unsigned short)(dv[1] >> 16));
      g1[0] = bf2f((unsigned short)(dv[2] & 0xffff)); g1[1] = bf2f((unsigned short)(dv[2] >> 16));
      g1[2] = bf2f((unsigned short)(dv[3] & 0xffff)); g1[3] = bf2f((unsigned short)(dv[3] >> 16));
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(gl + c8), s1 = *reinterpret_cast<const f32x4*>(gl + c8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g0[e] *= s0[e];
        g1[e] *= s1[e];
      }
      *reinterpret_cast<f32x4*>(xs + rr * LD + c8) = x0;
      *reinterpret_cast<f32x4*>(xs + rr * LD + c8 + 4) = x1;
      *reinterpret_cast<f32x4*>(gs + rr * LD + c8) = g0;
      *reinterpret_cast<f32x4*>(gs + rr * LD + c8 + 4) = g1;
      *reinterpret_cast<u32x4*>(ys + rr * D + c8) = py[q];
    }
    __syncthreads();
    if (r0 + step < M) prefetch(r0 + step);
    if (which < 2) {   // u[rr][j] = x[rr] . A[j]   |   w[rr][j] = gs[rr] . B[:, j]
      const float* a = (which ? gs : xs) + prr * LD;
      const float* b = (which ? Bt : At) + pj * LD;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int k = 0; k < D; k += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + k);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + k);
        acc[0] = fmaf(av[0], bv[0], acc[0]); acc[1] = fmaf(av[1], bv[1], acc[1]);
        acc[2] = fmaf(av[2], bv[2], acc[2]); acc[3] = fmaf(av[3], bv[3], acc[3]);
      }
      (which ? ws : us)[prr * R + pj] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      const int c = tid + 256 * cc;
      if (c < D) {
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) {
          const float g = gs[rr * LD + c], x = xs[rr * LD + c];
          const f32x4 u0 = *reinterpret_cast<const f32x4*>(us + rr * R), u1 = *reinterpret_cast<const f32x4*>(us + rr * R + 4);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + rr * R), w1 = *reinterpret_cast<const f32x4*>(ws + rr * R + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            accB[cc][j] = fmaf(g, u0[j], accB[cc][j]);
            accB[cc][4 + j] = fmaf(g, u1[j], accB[cc][4 + j]);
            accA[cc][j] = fmaf(w0[j], x, accA[cc][j]);
            accA[cc][4 + j] = fmaf(w1[j], x, accA[cc][4 + j]);
          }
          accM[cc] = fmaf(g, bf2f(ys[rr * D + c]) - bcol[cc], accM[cc]);   // g dy_true (y - b); g = 0 on rows past M
        }
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < CC; ++cc) {
    const int c = tid + 256 * cc;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        atomicAdd(dB + (long)c * R + j, scaling * accB[cc][j]);
        atomicAdd(dA + (long)j * D + c, scaling * accA[cc][j]);
      }
      // accM = sum g_col dy_st (y_st - b_st) with g_col = yscale mag / nrm:  dm = sum dy_st (y_st - b_st) / mag
      atomicAdd(dm + c, accM[cc] / (gl[c] * mag[c]));
    }
  }
}

int launch_dora_grads_multi(const void* X, long ldx, const void* dY, const void* Y, long ldy, int np,
                            const long* col_off, const float* const* bias_st, const float* yscale,
                            const float* scaling, const float* const* A, const float* const* Bm,
                            const float* const* mag, const float* const* nrm, float* const* dA, float* const* dB,
                            float* const* dm, long M, int d, hipStream_t s, void* scratch, size_t scratch_bytes);
size_t dora_grads_scratch_bytes(int np, int d);

int launch_dora_grads(const void* X, long ldx, const void* dY, const void* Y, long ldy, const float* bias_st,
                      float yscale, float scaling, const float* A, const float* Bm, const float* mag,
                      const float* nrm, float* dA, float* dB, float* dm, long M, int d, int r, hipStream_t s,
                      void* scratch, size_t scratch_bytes) {
  GWW_REQUIRE(r == 8 && (d == 128 || d == 384 || d == 512 || d == 768 || d == 1024 || d == 1280),
              "dora_grads: only r = 8 and d in {128, 384, 512, 768, 1024, 1280} (got d=%d r=%d)", d, r);
  if (M == 0) return GWW_OK;
  static const int old_kernel = (int)lab_int("GWW_DORA_OLD", 0);   // comparison aid: 1 = VALU
  // kernels only, 2 = register-blocked VALU kernel instead of the MFMA kernel
  if ((d == 384 || d == 512 || d == 768) && !old_kernel) {   // matrix-core kernel (dora_grads.hip)
    const long off = 0;
    return launch_dora_grads_multi(X, ldx, dY, Y, ldy, 1, &off, &bias_st, &yscale, &scaling, &A, &Bm, &mag, &nrm, &dA, &dB,
                                   &dm, M, d, s, scratch, scratch_bytes);
  }
  if (d <= 768 && d != 128 && old_kernel != 1) {
    static const long nb_env = lab_int("GWW_DORA_BLOCKS", 0);   // tuning aid (lab build)
    const int rt = d <= 512 ? 16 : 8;
    long nb = cdiv(M, rt);
    const long nb_max = nb_env > 0 ? nb_env : 256;     // one workgroup per CU: fewer contended atomics at the end
    if (nb > nb_max) nb = nb_max;
    const size_t lds = (size_t)((2 * rt + 2 * 8) * (d + 4) + 2 * rt * 8 + d) * 4 + (size_t)rt * d * 2;
#define GWW_DGR(DD)                                                                                                  \
  do {                                                                                                               \
    GWW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dora_grads_rb<DD>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
    hipLaunchKernelGGL((k_dora_grads_rb<DD>), dim3((unsigned)nb), dim3(256), lds, s, (const unsigned short*)X, ldx,  \
                       (const unsigned short*)dY, (const unsigned short*)Y, ldy, bias_st, yscale, scaling, A, Bm,    \
                       mag, nrm, dA, dB, dm, M);                                                                     \
  } while (0)
    if (d == 384) GWW_DGR(384);
    else if (d == 512) GWW_DGR(512);
    else GWW_DGR(768);
#undef GWW_DGR
    GWW_LAUNCH_CHECK();
    return GWW_OK;
  }
  long blocks = cdiv(M, d <= 512 ? 32 : (d <= 1024 ? 16 : 8));
  if (blocks > 1024) blocks = 1024;
#define GWW_DG(DD)                                                                                                \
  hipLaunchKernelGGL((k_dora_grads<DD, 8>), dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)X, ldx, \
                     (const unsigned short*)dY, (const unsigned short*)Y, ldy, bias_st, yscale, scaling, A, Bm, mag, \
                     nrm, dA, dB, dm, M)
  if (d == 128) GWW_DG(128);
  else if (d == 384) GWW_DG(384);
  else if (d == 512) GWW_DG(512);
  else if (d == 768) GWW_DG(768);
  else if (d == 1024) GWW_DG(1024);
  else GWW_DG(1280);
#undef GWW_DG
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_layernorm_bwd(const float* x, const float* gamma, const void* dy, int dy_is_f32, float* dx,
                                 int accumulate, void* dx_bf16, long M, int d, void* stream) {
  GWW_REQUIRE(x && gamma && dy && dx, "gww_layernorm_bwd: NULL argument");
  return launch_ln_bwd(x, gamma, dy, dy_is_f32, dx, accumulate, dx_bf16, M, d, (hipStream_t)stream);
}

extern "C" int gww_gelu_bf16(const void* z, const void* dgelu_or_null, void* out, long n, void* stream) {
  GWW_REQUIRE(z && out, "gww_gelu_bf16: NULL argument");
  return launch_gelu_bf16(z, dgelu_or_null, out, n, (hipStream_t)stream);
}

extern "C" int gww_dora_grads(const void* X, long ldx, const void* dY, const void* Y, long ldy, const float* bias_st,
                              float yscale, float scaling, const float* A, const float* B, const float* mag,
                              const float* nrm, float* dA, float* dB, float* dm, long M, int d, int r,
                              void* stream) {
  GWW_REQUIRE(X && dY && Y && bias_st && A && B && mag && nrm && dA && dB && dm, "gww_dora_grads: NULL argument");
  return launch_dora_grads(X, ldx, dY, Y, ldy, bias_st, yscale, scaling, A, B, mag, nrm, dA, dB, dm, M, d, r,
                           (hipStream_t)stream, nullptr, 0);
}
