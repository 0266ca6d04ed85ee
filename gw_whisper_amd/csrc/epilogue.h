// Shared GEMM epilogue: one lane owns 4 consecutive output columns n..n+3 of row m
// (the "swapped operand" MFMA layout used by gemm_bf16.hip and gemm_f32.hip).
#pragma once
#include "common.h"

namespace gww {

enum : int { EPI_CONV1 = 4 };   // conv1: gelu(acc+bias) -> padded token-major buffer, row m + 1
enum : int { EPI_DGELU = 5 };   // training backward (gemm_astat.hip): C = mul * gelu'(acc + bias), mul bf16 [M, N] (may alias C)

template <int EPI, bool BF16OUT>
__device__ __forceinline__ void epilogue_store4(f32x4 acc, long m, int n, long M, int N,
                                                const float* __restrict__ bias,
                                                const float* resid,
                                                const float* __restrict__ pos, void* C,
                                                int rows_per_batch, int valid_rows) {
  if (m >= M || n >= N) return;
  auto gelu_erf = [](float v) { return BF16OUT ? ::gww::gelu_fast(v) : ::gww::gelu_erf(v); };
  float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  float v0 = acc[0] + bv.x, v1 = acc[1] + bv.y, v2 = acc[2] + bv.z, v3 = acc[3] + bv.w;
  if constexpr (EPI == EPI_BIAS || EPI == EPI_GELU) {
    if constexpr (EPI == EPI_GELU) {
      v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
    }
    if constexpr (BF16OUT) {
      u32x2 o = {pack2bf(v0, v1), pack2bf(v2, v3)};
      *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(C) + m * N + n) = o;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + m * N + n) = make_float4(v0, v1, v2, v3);
    }
  } else if constexpr (EPI == EPI_RESID) {
    const float4 r = *reinterpret_cast<const float4*>(resid + m * N + n);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + m * N + n) =
        make_float4(r.x + v0, r.y + v1, r.z + v2, r.w + v3);
  } else if constexpr (EPI == EPI_CONV2) {
    // rows m = b * rows_per_batch + t; t == valid_rows is the per-batch garbage row
    const long b = m / rows_per_batch;
    const int t = (int)(m - b * rows_per_batch);
    if (t >= valid_rows) return;
    const float4 p = *reinterpret_cast<const float4*>(pos + (long)t * N + n);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + (b * valid_rows + t) * N + n) =
        make_float4(gelu_erf(v0) + p.x, gelu_erf(v1) + p.y, gelu_erf(v2) + p.z, gelu_erf(v3) + p.w);
  } else if constexpr (EPI == EPI_CONV1) {
    // rows m = b * rows_per_batch + t (rows_per_batch = T + 2); output row m + 1 of the
    // padded token-major buffer; t >= valid_rows lands on a zero-pad row
    const long b = m / rows_per_batch;
    const int t = (int)(m - b * rows_per_batch);
    const bool live = t < valid_rows;
    if constexpr (BF16OUT) {
      u32x2 o = {live ? pack2bf(gelu_erf(v0), gelu_erf(v1)) : 0u, live ? pack2bf(gelu_erf(v2), gelu_erf(v3)) : 0u};
      *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(C) + (m + 1) * N + n) = o;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + (m + 1) * N + n) =
          live ? make_float4(gelu_erf(v0), gelu_erf(v1), gelu_erf(v2), gelu_erf(v3)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}


}  // namespace gww
