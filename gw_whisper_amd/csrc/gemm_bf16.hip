// bf16 MFMA GEMM  C[M,N] = A[M,K] @ W[N,K]^T  with fused epilogues (K3/K4/K6/K8/K9
// of SURVEY.md section 2.1): every dense contraction of the Whisper encoder.
//
//   * A is [M, K] bf16 with an arbitrary row stride `lda` (elements).  The two
//     Conv1d(k=3) layers are run as GEMMs over OVERLAPPING rows of a token-major,
//     zero-padded activation: conv1 lda = 80, K = 240 (padded to 256 with zero
//     weights); conv2 (stride 2) lda = 2 d, K = 3 d.  No im2col buffer exists.
//   * W is the packed [N, K] bf16 panel (k contiguous) -- the natural MFMA operand.
//   * 128 x 128 x 64 tiles, 256 threads = 2 x 2 waves of 64 x 64, 16x16x32 MFMA
//     (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
//   * operands are computed "swapped" (D = W_tile . A_tile^T) so each lane ends
//     up with 4 CONSECUTIVE output columns of one row: 8-byte bf16 / 16-byte fp32
//     vector stores and vector bias / residual loads in the epilogue.
//   * LDS tiles are [row][64] bf16 (128-B rows) with the 16-B chunk index XORed by
//     (row >> 1) & 7: ds_read_b128 fragment reads are bank-conflict free.
//   * register-staged double buffering: global loads of tile k+1 are issued before
//     the MFMAs of tile k and written to the other LDS buffer after them; one
//     barrier per k-tile.
//   * blockIdx is remapped so each XCD (blockIdx % 8 share an L2) owns a contiguous
//     range of tiles with the N tiles of one row panel adjacent: the A panel is
//     fetched from HBM once and re-read from that XCD's L2.
#include "common.h"
#include "epilogue.h"

namespace gww {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ int swz_off(int row, int chunk) {   // byte offset inside a [128][64] bf16 tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16(const unsigned short* __restrict__ A, long lda,
                                                      const unsigned short* __restrict__ W,
                                                      const float* __restrict__ bias,
                                                      const float* resid,
                                                      const float* __restrict__ pos, void* C,
                                                      long M, int N, int K, int rows_per_batch,
                                                      int valid_rows, int tiles_n, long n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BM * BK * 2];   // 64 KB
  constexpr int TILE_BYTES = BM * BK * 2;
  auto As = [&](int buf) -> unsigned char* { return lds + buf * TILE_BYTES; };
  auto Bs = [&](int buf) -> unsigned char* { return lds + (2 + buf) * TILE_BYTES; };

  // XCD-aware bijective remap: blocks with equal (blockIdx % 8) share an L2
  long bid = blockIdx.x;
  {
    const long q = n_tiles / 8, r = n_tiles % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const long tm = bid / tiles_n;
  const int tn = (int)(bid - tm * tiles_n);
  const long m0 = tm * BM;
  const int n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // global -> register staging: 4 chunks of A and 4 of W per thread per k-tile
  const unsigned short* a_src[4];
  const unsigned short* w_src[4];
  int st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, kc = c & 7;
    long ar = m0 + row;
    if (ar >= M) ar = M - 1;
    int wr = n0 + row;
    if (wr >= N) wr = N - 1;
    a_src[i] = A + ar * lda + kc * 8;
    w_src[i] = W + (long)wr * K + kc * 8;
    st_off[i] = swz_off(row, kc);
  }
  u32x4 ra[4], rw[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(a_src[i] + k0);
      rw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + k0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(As(buf) + st_off[i]) = ra[i];
      *reinterpret_cast<u32x4*>(Bs(buf) + st_off[i]) = rw[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(As(buf) + swz_off(r, ks * 4 + fk));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + frow;
        wf[j] = *reinterpret_cast<const bf16x8*>(Bs(buf) + swz_off(r, ks * 4 + fk));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // D[n][m] layout: lane -> m = lane & 15, n = (lane >> 4) * 4 + reg
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      epilogue_store4<EPI, true>(acc[i][j], m, n, M, N, bias, resid, pos, C, rows_per_batch, valid_rows);
    }
  }
}

int launch_gemm_bf16(const void* A, long lda, const void* W, const float* bias, const float* resid,
                     const float* pos, void* C, long M, int N, int K, int epi, int rows_per_batch,
                     hipStream_t s) {
  GWW_REQUIRE(A && W && C, "gemm_bf16: NULL operand");
  GWW_REQUIRE(K % BK == 0 && K > 0, "gemm_bf16: K=%d must be a positive multiple of %d", K, BK);
  GWW_REQUIRE(N % 4 == 0 && N > 0, "gemm_bf16: N=%d must be a positive multiple of 4", N);
  GWW_REQUIRE(lda % 8 == 0, "gemm_bf16: lda=%ld must be a multiple of 8 (16-byte rows)", lda);
  GWW_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 15) == 0,
              "gemm_bf16: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  const int tiles_n = (int)cdiv(N, BN);
  const long n_tiles = cdiv(M, BM) * tiles_n;
  GWW_REQUIRE(n_tiles < 2147483647L, "gemm_bf16: grid too large");
  int valid_rows = 0;
  if (epi == EPI_CONV2) {
    GWW_REQUIRE(pos && rows_per_batch > 1, "gemm_bf16: conv2 epilogue needs pos and rows_per_batch");
    valid_rows = rows_per_batch - 1;
  } else if (epi == EPI_CONV1) {
    GWW_REQUIRE(rows_per_batch > 2, "gemm_bf16: conv1 epilogue needs rows_per_batch");
    valid_rows = rows_per_batch - 2;
  } else if (epi == EPI_RESID) {
    GWW_REQUIRE(resid != nullptr, "gemm_bf16: residual epilogue needs resid");
  }
  dim3 grid((unsigned)n_tiles), block(256);
#define GWW_GEMM_CASE(E)                                                                              \
  case E:                                                                                             \
    hipLaunchKernelGGL((k_gemm_bf16<E>), grid, block, 0, s, (const unsigned short*)A, lda,            \
                       (const unsigned short*)W, bias, resid, pos, C, M, N, K, rows_per_batch,        \
                       valid_rows, tiles_n, n_tiles);                                                 \
    break;
  switch (epi) {
    GWW_GEMM_CASE(EPI_BIAS) GWW_GEMM_CASE(EPI_GELU) GWW_GEMM_CASE(EPI_RESID)
    GWW_GEMM_CASE(EPI_CONV2) GWW_GEMM_CASE(EPI_CONV1)
    default:
      return fail(GWW_ERR_ARG, "gemm_bf16: unknown epilogue %d", epi);
  }
#undef GWW_GEMM_CASE
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------- mel transpose
// [B, C=80, T=3000] fp32 (HF input_features layout) -> token-major [B, T + 2, C]
// with zero rows 0 and T+1 (the Conv1d padding), bf16 or fp32.
template <typename OutT>
__global__ __launch_bounds__(256) void k_mel_to_tokens(const float* __restrict__ mel, OutT* __restrict__ out,
                                                       int C, int T) {
  __shared__ float tile[64][81];
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const float* src = mel + (long)b * C * T;
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, tt = i & 63;
    tile[tt][c] = (t0 + tt < T) ? src[(long)c * T + t0 + tt] : 0.f;
  }
  __syncthreads();
  OutT* dst = out + ((long)b * (T + 2) + 1 + t0) * C;
  for (int i = threadIdx.x; i < 64 * C; i += 256) {
    const int tt = i / C, c = i - tt * C;
    if (t0 + tt < T) {
      if constexpr (sizeof(OutT) == 2) dst[(long)tt * C + c] = f2bf(tile[tt][c]);
      else dst[(long)tt * C + c] = tile[tt][c];
    }
  }
  if (blockIdx.x == 0) {
    OutT* z0 = out + (long)b * (T + 2) * C;
    OutT* z1 = out + ((long)b * (T + 2) + T + 1) * C;
    for (int i = threadIdx.x; i < C; i += 256) {
      z0[i] = OutT(0);
      z1[i] = OutT(0);
    }
  }
}

int launch_mel_to_tokens(const float* mel, void* out, int out_bf16, int B, int C, int T, hipStream_t s) {
  GWW_REQUIRE(C <= 80, "mel_to_tokens: n_mels=%d > 80 unsupported", C);
  if (B == 0) return GWW_OK;
  dim3 grid((unsigned)cdiv(T, 64), (unsigned)B), block(256);
  if (out_bf16)
    hipLaunchKernelGGL(k_mel_to_tokens<unsigned short>, grid, block, 0, s, mel, (unsigned short*)out, C, T);
  else
    hipLaunchKernelGGL(k_mel_to_tokens<float>, grid, block, 0, s, mel, (float*)out, C, T);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_gemm_bf16(const void* A, const void* W, const float* bias, const float* resid, void* C,
                             long M, int N, int K, int epilogue, void* stream) {
  GWW_REQUIRE(epilogue >= 0 && epilogue <= 2, "gww_gemm_bf16: epilogue must be 0, 1 or 2");
  return launch_gemm_bf16(A, K, W, bias, resid, nullptr, C, M, N, K, epilogue, 0, (hipStream_t)stream);
}
