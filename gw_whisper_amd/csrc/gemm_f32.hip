// fp32 MFMA GEMM (v_mfma_f32_16x16x4_f32): the exact-fp32 parity path of the encoder
// (GWW_PREC_F32).  Same contract, operand layout and epilogues as gemm_bf16.hip; the
// MFMA is bit-for-bit a k-ordered fmaf chain, so results match an fp32 CPU reference
// to accumulation-order noise.  64 x 64 x 32 tiles, 256 threads = 2 x 2 waves of
// 32 x 32; single LDS buffer (this path is the checker's twin, not the fast path).
#include "common.h"
#include "epilogue.h"

namespace gww {

constexpr int FM = 64, FN = 64, FK = 32, FLD = FK + 1;

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32(const float* __restrict__ A, long lda,
                                                  const float* __restrict__ W, const float* __restrict__ bias,
                                                  const float* resid, const float* __restrict__ pos, float* C,
                                                  long M, int N, int K, int rows_per_batch, int valid_rows,
                                                  int tiles_n) {
  __shared__ float As[FM][FLD];
  __shared__ float Ws[FN][FLD];
  const long tm = blockIdx.x / tiles_n;
  const int tn = (int)(blockIdx.x - tm * tiles_n);
  const long m0 = tm * FM;
  const int n0 = tn * FN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < K; k0 += FK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 4;
      long ar = m0 + row;
      if (ar >= M) ar = M - 1;
      int wr = n0 + row;
      if (wr >= N) wr = N - 1;
      const float4 av = *reinterpret_cast<const float4*>(A + ar * lda + k0 + kc);
      const float4 wv = *reinterpret_cast<const float4*>(W + (long)wr * K + k0 + kc);
      As[row][kc] = av.x; As[row][kc + 1] = av.y; As[row][kc + 2] = av.z; As[row][kc + 3] = av.w;
      Ws[row][kc] = wv.x; Ws[row][kc + 1] = wv.y; Ws[row][kc + 2] = wv.z; Ws[row][kc + 3] = wv.w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FK; kk += 4) {
      float af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = As[wm * 32 + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = Ws[wn * 32 + j * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long m = m0 + wm * 32 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
      epilogue_store4<EPI, false>(acc[i][j], m, n, M, N, bias, resid, pos, C, rows_per_batch, valid_rows);
    }
  }
}

int launch_gemm_f32(const float* A, long lda, const float* W, const float* bias, const float* resid,
                    const float* pos, float* C, long M, int N, int K, int epi, int rows_per_batch,
                    hipStream_t s) {
  GWW_REQUIRE(A && W && C, "gemm_f32: NULL operand");
  GWW_REQUIRE(K % FK == 0 && K > 0, "gemm_f32: K=%d must be a positive multiple of %d", K, FK);
  GWW_REQUIRE(N % 4 == 0 && N > 0, "gemm_f32: N=%d must be a positive multiple of 4", N);
  GWW_REQUIRE(lda % 4 == 0, "gemm_f32: lda=%ld must be a multiple of 4", lda);
  GWW_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 15) == 0,
              "gemm_f32: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  const int tiles_n = (int)cdiv(N, FN);
  const long n_tiles = cdiv(M, FM) * tiles_n;
  GWW_REQUIRE(n_tiles < 2147483647L, "gemm_f32: grid too large");
  int valid_rows = 0;
  if (epi == EPI_CONV2) {
    GWW_REQUIRE(pos && rows_per_batch > 1, "gemm_f32: conv2 epilogue needs pos and rows_per_batch");
    valid_rows = rows_per_batch - 1;
  } else if (epi == EPI_CONV1) {
    GWW_REQUIRE(rows_per_batch > 2, "gemm_f32: conv1 epilogue needs rows_per_batch");
    valid_rows = rows_per_batch - 2;
  } else if (epi == EPI_RESID) {
    GWW_REQUIRE(resid != nullptr, "gemm_f32: residual epilogue needs resid");
  }
  dim3 grid((unsigned)n_tiles), block(256);
#define GWW_GEMM_CASE(E)                                                                                  \
  case E:                                                                                                 \
    hipLaunchKernelGGL((k_gemm_f32<E>), grid, block, 0, s, A, lda, W, bias, resid, pos, C, M, N, K,       \
                       rows_per_batch, valid_rows, tiles_n);                                              \
    break;
  switch (epi) {
    GWW_GEMM_CASE(EPI_BIAS) GWW_GEMM_CASE(EPI_GELU) GWW_GEMM_CASE(EPI_RESID)
    GWW_GEMM_CASE(EPI_CONV2) GWW_GEMM_CASE(EPI_CONV1)
    default:
      return fail(GWW_ERR_ARG, "gemm_f32: unknown epilogue %d", epi);
  }
#undef GWW_GEMM_CASE
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_gemm_f32(const float* A, const float* W, const float* bias, const float* resid, float* C,
                            long M, int N, int K, int epilogue, void* stream) {
  GWW_REQUIRE(epilogue >= 0 && epilogue <= 2, "gww_gemm_f32: epilogue must be 0, 1 or 2");
  return launch_gemm_f32(A, K, W, bias, resid, nullptr, C, M, N, K, epilogue, 0, (hipStream_t)stream);
}
