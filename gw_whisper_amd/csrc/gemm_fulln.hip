// "Full-N" bf16 MFMA GEMM for the long-K, narrow-N contractions of the encoder
// (fc2: K = 4d, N = d;  conv2: K = 3d, N = d):     C[M,N] = epi(A[M,K] @ W[N,K]^T + b)
//
// Why a third GEMM: with N = d one workgroup can own COMPLETE output rows, so the big
// operand A (fc1's activations, 1.2 GB at B = 256) is read from HBM exactly once.  The
// 256 x 128 tile kernel re-read each A panel N/128 = 3 times and, with a reuse distance
// of many MB per XCD, those re-reads came from HBM: 3.5 GB of traffic for a 1.2 GB
// operand (profiles/r01_*).
//
//   * workgroup tile 128 (M) x N (384 or 512) x 64 (K), 512 threads = 2 x 4 waves of
//     64 x N/4, v_mfma_f32_16x16x32_bf16, swapped operands (row m on the lane, 4
//     consecutive n in registers).
//   * A and W tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, whole 128-byte
//     lines per row) into a double buffer, counted vmcnt + one raw s_barrier per k-tile
//     (a 32-deep-K / 4-stage build is kept behind GWW_FN_BK=32: its 64-byte row pieces
//     double the request count and measured slower).  The 16-B chunk index is XORed with
//     a row key on the SOURCE address and again on the ds_read_b128 side: conflict-free.
//   * epilogue: the finished 128 x N tile is staged in the (now idle) ring memory and
//     written with whole-row, 1 KiB-per-instruction stores; conv2 adds the position
//     table and drops the per-batch garbage row there (fp32 output, two column halves).
#include "common.h"
#include "epilogue.h"

namespace gww {

#ifndef GWW_FN_BK
#define GWW_FN_BK 64
#endif
constexpr int FN_BM = 128, FN_BK = GWW_FN_BK, FN_NST = (FN_BK == 32 ? 4 : 2), FN_D = FN_NST - 1;
constexpr int FN_ROWB = FN_BK * 2;            // bytes per LDS row (64 or 128)
constexpr int FN_PROWS = 1024 / FN_ROWB;      // rows per 1-KiB LDS-DMA piece (16 or 8)
constexpr int FN_CPR = FN_ROWB / 16;          // 16-byte chunks per row (4 or 8)

__device__ __forceinline__ int fn_key(int row) { return FN_BK == 32 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7); }
__device__ __forceinline__ int fn_swz(int row, int chunk) { return row * FN_ROWB + ((chunk ^ fn_key(row)) << 4); }

template <int N>
__device__ __forceinline__ void fn_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NW = n-tiles (16 wide) per wave; N = 64 * NW (384 -> 6, 512 -> 8)
template <int EPI, int NW>
__global__ __launch_bounds__(512, 1) void k_gemm_fulln(const unsigned short* __restrict__ A, long lda,
                                                       const unsigned short* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ pos, void* __restrict__ C,
                                                       long M, int K, int rows_per_batch, int valid_rows) {
  constexpr int N = 64 * NW;
  constexpr int A_BYTES = FN_BM * FN_ROWB;
  constexpr int W_BYTES = N * FN_ROWB;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int A_PIECES = A_BYTES / 1024 / 8;        // LDS-DMA pieces per wave per k-tile
  constexpr int W_PIECES = W_BYTES / 1024 / 8;
  constexpr int GLDS = A_PIECES + W_PIECES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[FN_NST * STAGE];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const long m0 = (long)blockIdx.x * FN_BM;
  const int nk = K / FN_BK;

  // LDS-DMA sources: one piece = FN_PROWS rows x FN_ROWB bytes; lane -> row lane / FN_CPR, chunk position lane % FN_CPR
  const unsigned short* a_src[A_PIECES];
  const unsigned short* w_src[W_PIECES];
#pragma unroll
  for (int j = 0; j < A_PIECES; ++j) {
    const int row = FN_PROWS * (A_PIECES * wave + j) + lane / FN_CPR;
    const int chunk = (lane % FN_CPR) ^ fn_key(row);
    long ar = m0 + row;
    if (ar >= M) ar = M - 1;
    a_src[j] = A + ar * lda + chunk * 8;
  }
#pragma unroll
  for (int j = 0; j < W_PIECES; ++j) {
    const int wrow = FN_PROWS * (W_PIECES * wave + j) + lane / FN_CPR;
    const int wchunk = (lane % FN_CPR) ^ fn_key(wrow);
    w_src[j] = W + (long)wrow * K + wchunk * 8;
  }
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % FN_NST) * STAGE;
    const int k0 = kt * FN_BK;
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j)
      __builtin_amdgcn_global_load_lds((g_ptr)(a_src[j] + k0), (lds_ptr)(st + (A_PIECES * wave + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < W_PIECES; ++j)
      __builtin_amdgcn_global_load_lds((g_ptr)(w_src[j] + k0), (lds_ptr)(st + A_BYTES + (W_PIECES * wave + j) * 1024),
                                       16, 0, 0);
  };

  f32x4 acc[4][NW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < FN_D; ++p)
    if (p < nk) issue(p);
  const int frow = lane & 15, fk = lane >> 4;
  auto read_frags = [&](const unsigned char* As, int ks, bf16x8 (&af)[4], bf16x8 (&wf)[NW]) {
    const unsigned char* Ws = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(As + fn_swz(wm * 64 + i * 16 + frow, ks * 4 + fk));
#pragma unroll
    for (int j = 0; j < NW; ++j)
      wf[j] = *reinterpret_cast<const bf16x8*>(Ws + fn_swz(wn * 16 * NW + j * 16 + frow, ks * 4 + fk));
  };
  auto mfma_step = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[NW]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  };
  if constexpr (FN_BK == 64) {
    // Software pipeline over 32-deep k-steps with two fragment register sets: the ds_reads of step
    // s+1 are in flight while the MFMAs of step s issue, also across the k-tile barrier.
    bf16x8 afA[4], wfA[NW], afB[4], wfB[NW];
    fn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (1 < nk) issue(1);
    read_frags(lds, 0, afA, wfA);
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* cur = lds + (kt & 1) * STAGE;
      read_frags(cur, 1, afB, wfB);
      mfma_step(afA, wfA);
      if (kt + 1 < nk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this buffer is refilled after the barrier
        fn_wait_vmcnt<0>();                                   // tile kt+1 (the only group in flight) landed
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        read_frags(lds + ((kt + 1) & 1) * STAGE, 0, afA, wfA);
      }
      mfma_step(afB, wfB);
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + FN_D <= nk) fn_wait_vmcnt<GLDS * (FN_D - 1)>();
      else fn_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + FN_D < nk) issue(kt + FN_D);
      bf16x8 af[4], wf[NW];
      read_frags(lds + (kt % FN_NST) * STAGE, 0, af, wf);
      mfma_step(af, wf);
    }
  }

  // ---- epilogue through the idle ring memory
  __builtin_amdgcn_s_barrier();   // every wave is done reading the last k-tile
  if constexpr (EPI == EPI_BIAS || EPI == EPI_GELU) {
    constexpr int STRIDE = (N + 8) * 2;                 // bytes per staged bf16 row
    static_assert(FN_BM * STRIDE <= FN_NST * STAGE, "staging does not fit the ring");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ml = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int n = wn * 16 * NW + j * 16 + (lane >> 4) * 4;
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
        if constexpr (EPI == EPI_GELU) {
          v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3);
        }
        u32x2 o = {pack2bf(v0, v1), pack2bf(v2, v3)};
        *reinterpret_cast<u32x2*>(lds + ml * STRIDE + n * 2) = o;
      }
    }
    __syncthreads();
    constexpr int CPR = N / 8;                          // 16-byte chunks per row
    unsigned short* Cb = reinterpret_cast<unsigned short*>(C);
    for (int idx = tid; idx < FN_BM * CPR; idx += 512) {
      const int row = idx / CPR, ch = idx - row * CPR;
      const u32x4 u = *reinterpret_cast<const u32x4*>(lds + row * STRIDE + ch * 16);
      *reinterpret_cast<u32x4*>(Cb + (m0 + row) * N + ch * 8) = u;   // rows past M: caller-padded scratch
    }
  } else {
    // conv2: gelu(acc + bias) + pos[t]  ->  fp32 [b * valid + t][N]; rows m = b * rows_per_batch + t
    constexpr int HALF = N / 2;
    constexpr int STRIDE = (HALF + 4) * 4;              // bytes per staged fp32 half row
    static_assert(FN_BM * STRIDE <= FN_NST * STAGE, "staging does not fit the ring");
    float* Cf = reinterpret_cast<float*>(C);
    // output row and position row of each of the 128 tile rows, computed ONCE (one 64-bit division per row instead of
    // one per 16-byte chunk in the copy loops below): dst_row = b * valid + t, or -1 for the per-batch garbage row
    int* row_dst = reinterpret_cast<int*>(lds + FN_BM * STRIDE);          // [128] after the staged half tile
    int* row_t = row_dst + FN_BM;
    static_assert(FN_BM * STRIDE + 2 * FN_BM * 4 <= FN_NST * STAGE, "row table does not fit the ring");
    if (tid < FN_BM) {
      const long m = m0 + tid;
      const long b = m / rows_per_batch;
      const int t = (int)(m - b * rows_per_batch);
      row_t[tid] = t;
      row_dst[tid] = (m < M && t < valid_rows) ? (int)(b * valid_rows + t) : -1;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if ((wn >> 1) == h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ml = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const int n = wn * 16 * NW + j * 16 + (lane >> 4) * 4;
            const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 o = make_float4(gelu_fast(acc[i][j][0] + bv.x), gelu_fast(acc[i][j][1] + bv.y),
                                         gelu_fast(acc[i][j][2] + bv.z), gelu_fast(acc[i][j][3] + bv.w));
            *reinterpret_cast<float4*>(lds + ml * STRIDE + (n - h * HALF) * 4) = o;
          }
        }
      }
      __syncthreads();
      constexpr int CPR = HALF / 4;                     // float4 chunks per half row
      for (int idx = tid; idx < FN_BM * CPR; idx += 512) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const int dst = row_dst[row];
        if (dst >= 0) {
          float4 v = *reinterpret_cast<const float4*>(lds + row * STRIDE + ch * 16);
          const float4 p = *reinterpret_cast<const float4*>(pos + (long)row_t[row] * N + h * HALF + ch * 4);
          v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
          *reinterpret_cast<float4*>(Cf + (long)dst * N + h * HALF + ch * 4) = v;
        }
      }
      __syncthreads();
    }
  }
}

// C rows must be allocated up to the next multiple of 128 for the bias / GELU epilogues.
int launch_gemm_fulln(const void* A, long lda, const void* W, const float* bias, const float* pos, void* C,
                      long M, int N, int K, int epi, int rows_per_batch, hipStream_t s) {
  GWW_REQUIRE(A && W && C, "gemm_fulln: NULL operand");
  GWW_REQUIRE(N == 384 || N == 512, "gemm_fulln: N=%d unsupported (384 or 512)", N);
  GWW_REQUIRE(K % FN_BK == 0 && K >= FN_BK, "gemm_fulln: K=%d must be a multiple of %d", K, FN_BK);
  GWW_REQUIRE(lda % 8 == 0, "gemm_fulln: lda must be a multiple of 8");
  GWW_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 15) == 0,
              "gemm_fulln: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  int valid_rows = 0;
  if (epi == EPI_CONV2) {
    GWW_REQUIRE(pos && rows_per_batch > 1, "gemm_fulln: conv2 epilogue needs pos and rows_per_batch");
    valid_rows = rows_per_batch - 1;
  }
  dim3 grid((unsigned)cdiv(M, FN_BM)), block(512);
#define GWW_FN_LAUNCH(E, NW)                                                                              \
  hipLaunchKernelGGL((k_gemm_fulln<E, NW>), grid, block, 0, s, (const unsigned short*)A, lda,             \
                     (const unsigned short*)W, bias, pos, C, M, K, rows_per_batch, valid_rows)
#define GWW_FN_N(E)                     \
  do {                                  \
    if (N == 384) GWW_FN_LAUNCH(E, 6);  \
    else GWW_FN_LAUNCH(E, 8);           \
  } while (0)
  if (epi == EPI_BIAS) GWW_FN_N(EPI_BIAS);
  else if (epi == EPI_GELU) GWW_FN_N(EPI_GELU);
  else if (epi == EPI_CONV2) GWW_FN_N(EPI_CONV2);
  else return fail(GWW_ERR_ARG, "gemm_fulln: unsupported epilogue %d", epi);
#undef GWW_FN_N
#undef GWW_FN_LAUNCH
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_gemm_fulln_bf16(const void* A, const void* W, const float* bias, void* C, long M, int N, int K,
                                   int epilogue, void* stream) {
  GWW_REQUIRE(epilogue == 0 || epilogue == 1, "gww_gemm_fulln_bf16: epilogue must be 0 (bias) or 1 (GELU)");
  return launch_gemm_fulln(A, K, W, bias, nullptr, C, M, N, K, epilogue, 0, (hipStream_t)stream);
}
