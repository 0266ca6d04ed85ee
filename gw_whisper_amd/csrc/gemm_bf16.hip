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

#include <stdlib.h>

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

// =====================================================================================
// v2: persistent row-panel kernel for large M.
//   * 256 x 128 x 64 tiles, 512 threads = 4 x 2 waves of 64 x 64 (2 waves per SIMD).
//   * one workgroup owns a 256-row panel and walks a contiguous range of N tiles; the
//     (n, k) iteration space is flattened so the global->LDS pipeline never drains
//     inside the panel and the epilogue stores of tile n overlap the loads of tile n+1.
//   * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 16 B per lane) into a
//     3-stage ring, two k-tiles in flight; the ring is retired with COUNTED
//     s_waitcnt vmcnt(N) + a raw s_barrier (one per k-tile), never vmcnt(0) in the loop.
//     The LDS image is lane-linear, so the XOR swizzle is applied to the per-lane
//     SOURCE chunk and undone on the ds_read_b128 side (same involution).
//   * bias is staged in LDS once, so the in-loop epilogue issues no VGPR-destination
//     loads (those would force vmcnt(0) and drain the ring); stores are unconditional
//     (the caller pads every activation buffer to a multiple of 256 rows), which keeps
//     the per-wave VMEM count exact for the counted waits.
constexpr int BM2 = 256, BN2 = 128, BK2 = 64, NSTAGE = 3;
constexpr int A2_BYTES = BM2 * BK2 * 2, W2_BYTES = BN2 * BK2 * 2, STAGE2_BYTES = A2_BYTES + W2_BYTES;
constexpr int GLDS_PER_TILE = 6;     // per thread: 4 (A) + 2 (W)
constexpr int STORES_PER_TILE = 16;  // per thread: 4 x 4 accumulator tiles, one vector store each

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void k_gemm_bf16_v2(const unsigned short* __restrict__ A, long lda,
                                                         const unsigned short* __restrict__ W,
                                                         const float* __restrict__ bias, const float* resid,
                                                         const float* __restrict__ pos, void* C, long M, int N,
                                                         int K, int rows_per_batch, int valid_rows, int tiles_n,
                                                         int n_split) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE2_BYTES + 1536 * 4];
  float* lds_bias = reinterpret_cast<float*>(lds + NSTAGE * STAGE2_BYTES);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;

  const int panel = blockIdx.x / n_split, split = blockIdx.x - panel * n_split;
  const int nt0 = (int)((long)split * tiles_n / n_split), nt1 = (int)((long)(split + 1) * tiles_n / n_split);
  const long m0 = (long)panel * BM2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = K / BK2;
  const int total = (nt1 - nt0) * nk;

  // bias for this block's column range -> LDS (read back in the epilogue)
  for (int i = tid; i < (nt1 - nt0) * BN2; i += 512) lds_bias[i] = bias ? bias[nt0 * BN2 + i] : 0.f;

  // per-lane global sources of the LDS-DMA pieces (1 KiB = 8 rows x 128 B per wave-instruction)
  const unsigned short* a_src[4];
  long w_off[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * (4 * wave + j) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    long ar = m0 + row;
    if (ar >= M) ar = M - 1;           // rows past M only feed rows past M
    a_src[j] = A + ar * lda + chunk * 8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * (2 * wave + j) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    w_off[j] = (long)row * K + chunk * 8;
  }
  auto issue = [&](int it) {
    const int stage = it % NSTAGE;
    const int nn = nt0 + it / nk, k0 = (it % nk) * BK2;
    unsigned char* sa = lds + stage * STAGE2_BYTES + (4 * wave) * 1024;
    unsigned char* sw = lds + stage * STAGE2_BYTES + A2_BYTES + (2 * wave) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((g_ptr)(a_src[j] + k0), (lds_ptr)(sa + j * 1024), 16, 0, 0);
    const unsigned short* wb = W + (long)nn * BN2 * K + k0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((g_ptr)(wb + w_off[j]), (lds_ptr)(sw + j * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) issue(0);
  if (total > 1) issue(1);
  const int frow = lane & 15, fk = lane >> 4;
  bool stores_pending = false;   // epilogue stores were issued after the youngest LDS-DMA group
  for (int it = 0; it < total; ++it) {
    // retire this wave's pieces of tile `it` (everything older than the youngest in-flight group)
    if (it + 1 < total) {
      if (stores_pending) wait_vmcnt<GLDS_PER_TILE + STORES_PER_TILE>();
      else wait_vmcnt<GLDS_PER_TILE>();
    } else {
      wait_vmcnt<0>();
    }
    stores_pending = false;
    __builtin_amdgcn_s_barrier();
    if (it + 2 < total) issue(it + 2);
    const unsigned char* As = lds + (it % NSTAGE) * STAGE2_BYTES;
    const unsigned char* Ws = As + A2_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(As + swz_off(wm * 64 + i * 16 + frow, ks * 4 + fk));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const bf16x8*>(Ws + swz_off(wn * 64 + j * 16 + frow, ks * 4 + fk));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if ((it + 1) % nk == 0) {
      const int nn = nt0 + it / nk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = wn * 64 + j * 16 + (lane >> 4) * 4;
          const float4 bv = *reinterpret_cast<const float4*>(lds_bias + (nn - nt0) * BN2 + nl);
          f32x4 v = acc[i][j];
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          epilogue_store4<EPI, true>(v, m, nn * BN2 + nl, 0x7fffffffffffffffL, N, nullptr, resid, pos, C,
                                     rows_per_batch, valid_rows);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      stores_pending = true;
    }
  }
}

// v3: 256 x 256 tiles for the wide panels of whisper-base / -small (N % 256 == 0).  v2's 256 x 128 tile needs 48 KB of
// operands per 4.2 MFLOP -- 47 B per clock and CU at the MFMA rate; the square tile needs 64 KB per 8.4 MFLOP.
// 512 threads = 2 x 4 waves of 128 x 64 (8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16).  What the ablation builds
// (GWW_G3_ABL, tools/gemm_exp.py) showed on the first form of this kernel (two 64-KB stages, BK = 64: 640-700 TFLOP/s
// where MFMAs + barriers alone run 1 500-1 870): the eight LDS-DMA pieces per wave and k-tile cost ~60 issue cycles each
// and had to go out in one burst behind the tile's second barrier (-25 %), the 32 eight-byte stores per lane of the
// epilogue another 25-30 %.  Hence:
//   * BK = 32, FOUR 32-KB stages, three k-tiles in flight: the four DMA pieces of tile it + 3 are issued one per eight
//     MFMAs inside tile it (their stage was read in tile it - 1: ONE barrier per k-tile), counted vmcnt;
//   * LDS rows are 64 B: the 16-byte chunk p of row R sits at p ^ 2 ((R >> 3) & 1) -- conflict-free for the lane
//     groups of ds_read_b128 (rows R, R + 4, R + 8, R + 12 of a 16-row fragment share a bank row);
//   * bf16 epilogue: v_permlane16_swap pairs neighbouring 16-column tiles so that every lane stores 16 bytes (64
//     contiguous bytes per row and instruction) -- 16 stores per lane instead of 32 of eight bytes.
// Work item = (column split, row panel), split-major, contiguous per XCD (blockIdx % 8): the blocks of an XCD walk the
// panels of one column split together, whose W slice (<= 1.5 MB) stays in that L2.
#ifndef GWW_G3_ABL
#define GWW_G3_ABL 0   // diagnostic builds only (wrong results): 1 = no epilogue stores, 2 = no LDS-DMA / ring waits, 4 = no fragment reads
#endif
constexpr int BM3 = 256, BN3 = 256, BK3 = 32, NST3 = 4;
constexpr int A3_BYTES = BM3 * BK3 * 2, W3_BYTES = BN3 * BK3 * 2, STAGE3_BYTES = A3_BYTES + W3_BYTES;   // 16 + 16 KB
constexpr int GLDS3 = 4;       // per thread and k-tile: 2 (A) + 2 (W)

__device__ __forceinline__ int swz3(int row, int chunk) {   // byte offset inside a [256][32] bf16 tile (64-byte rows)
  return row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4);
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void k_gemm_bf16_v3(const unsigned short* __restrict__ A, long lda,
                                                         const unsigned short* __restrict__ W,
                                                         const float* __restrict__ bias, const float* resid,
                                                         void* C, long M, int N, int K, int tiles_n, int n_split) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST3 * STAGE3_BYTES + 1536 * 4];
  float* lds_bias = reinterpret_cast<float*>(lds + NST3 * STAGE3_BYTES);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;
  constexpr bool BF16OUT = EPI != EPI_RESID;
  unsigned short* __restrict__ const Cb = reinterpret_cast<unsigned short*>(C);   // (bf16 output never aliases an operand)
  constexpr int STORES3 = BF16OUT ? 16 : 32;   // per thread and output tile

  const int panels = (int)(gridDim.x / n_split);
  int item;
  {
    const int nb = (int)gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    item = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int split = item / panels, panel = item - split * panels;
  const int nt0 = (int)((long)split * tiles_n / n_split), nt1 = (int)((long)(split + 1) * tiles_n / n_split);
  const long m0 = (long)panel * BM3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nk = K / BK3;
  const int total = (nt1 - nt0) * nk;

  for (int i = tid; i < (nt1 - nt0) * BN3; i += 512) lds_bias[i] = bias ? bias[nt0 * BN3 + i] : 0.f;

  // LDS-DMA pieces: 1 KiB = 16 rows x 64 B; lane l lands at (row l >> 2, position l & 3), which holds chunk
  // (l & 3) ^ 2 ((row >> 3) & 1) of the row
  const unsigned short* a_src[2];
  long w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * (2 * wave + j) + (lane >> 2);
    const int chunk = (lane & 3) ^ (((row >> 3) & 1) << 1);
    long ar = m0 + row;
    if (ar >= M) ar = M - 1;           // rows past M only feed rows past M
    a_src[j] = A + ar * lda + chunk * 8;
    w_off[j] = (long)row * K + chunk * 8;
  }
  auto issue_piece = [&](int it, int j) {   // piece j (0, 1: A; 2, 3: W) of k-tile it
    if (GWW_G3_ABL & 2) return;
    const int stage = it & (NST3 - 1);
    const int nn = nt0 + it / nk, k0 = (it % nk) * BK3;
    unsigned char* sa = lds + stage * STAGE3_BYTES + (2 * wave) * 1024;
    if (j < 2) {
      __builtin_amdgcn_global_load_lds((g_ptr)(a_src[j] + k0), (lds_ptr)(sa + j * 1024), 16, 0, 0);
    } else {
      const unsigned short* wb = W + (long)nn * BN3 * K + k0;
      __builtin_amdgcn_global_load_lds((g_ptr)(wb + w_off[j - 2]), (lds_ptr)(sa + A3_BYTES + (j - 2) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < NST3 - 1; ++p)
    if (p < total)
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_piece(p, j);
  const int frow = lane & 15, fk = lane >> 4;
  // ---- the two waves of a SIMD alternate: waves 4 .. 7 (group 1; wave w + 4 shares wave w's SIMD) run ONE PHASE behind
  // waves 0 .. 3 (one extra barrier at the start, one for group 0 at the end), and a k-tile is two phases -- LOAD (the
  // tile's 12 fragment reads, the four LDS-DMA pieces of tile it + 3, nothing for the matrix pipe) and MFMA (32 MFMAs,
  // registers only).  Between two barriers one wave of every SIMD computes while its partner loads; in the first form of
  // this kernel both did the same thing at the same time and every component was paid in full (profiles/r03_gemm_ablation.md).
  //   * a tile's pieces are waited for by their issuers before the barrier that opens its FIRST reader's (group 0's) LOAD
  //     phase: group 0 waits at the end of MFMA(it - 1), group 1 at the end of LOAD(it - 1);
  //   * fragment reads are drained (lgkmcnt(0)) before the phase ends: the partner group's next LOAD phase requests tile
  //     it + 3 into the stage tile it - 1 was read from.
  const bool g1 = __builtin_amdgcn_readfirstlane(wave) >= 4;
  int stores_age = 0;   // > 0: an epilogue's stores still sit in the queue in front of pieces this wave may wait for
  auto wait_next = [&](int it) {   // this wave's pieces of tile it + 1; younger: tiles it + 2, it + 3 (+ an epilogue's stores)
    if (GWW_G3_ABL & 2) return;
    if (it + 1 < total) {
      if (it + 3 < total) {
        if (stores_age > 0 && !(GWW_G3_ABL & 1)) wait_vmcnt<2 * GLDS3 + STORES3>();
        else wait_vmcnt<2 * GLDS3>();
      } else {
        wait_vmcnt<0>();
      }
    }
    if (stores_age > 0) --stores_age;
  };
  if (!(GWW_G3_ABL & 2)) {
    if (total > 2) wait_vmcnt<2 * GLDS3>();
    else wait_vmcnt<0>();
  }
  if (g1) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < total; ++it) {
    __builtin_amdgcn_s_barrier();                       // ---- LOAD(it)
    const unsigned char* As = lds + (it & (NST3 - 1)) * STAGE3_BYTES;
    const unsigned char* Ws = As + A3_BYTES;
    bf16x8 af[8], wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wf[j] = *reinterpret_cast<const bf16x8*>(Ws + ((GWW_G3_ABL & 4) ? 0 : swz3(wn * 64 + j * 16 + frow, fk)));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(As + ((GWW_G3_ABL & 4) ? 0 : swz3(wm * 128 + i * 16 + frow, fk)));
    if (it + NST3 - 1 < total)
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_piece(it + NST3 - 1, j);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (g1) wait_next(it);
    __builtin_amdgcn_s_barrier();                       // ---- MFMA(it)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if ((it + 1) % nk == 0) {
      const int nn = nt0 + it / nk;
      // the bias of this lane's four column groups: asm reads -- hipcc cannot tell these LDS reads from the LDS-DMA
      // destinations in the same array and would drain the whole ring (vmcnt(0)) in front of them
      f32x4 bvj[4];
      {
        const unsigned ba = (unsigned)(unsigned long long)(lds_ptr)(lds_bias + (nn - nt0) * BN3 + wn * 64 + (lane >> 4) * 4);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                     "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bvj[0]), "=&v"(bvj[1]), "=&v"(bvj[2]), "=&v"(bvj[3]) : "v"(ba));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long m = m0 + wm * 128 + i * 16 + (lane & 15);
        if constexpr (BF16OUT) {
#pragma unroll
          for (int jp = 0; jp < 4; jp += 2) {
            unsigned pk[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const f32x4 bv = bvj[jp + t];
              float v0 = acc[i][jp + t][0] + bv[0], v1 = acc[i][jp + t][1] + bv[1], v2 = acc[i][jp + t][2] + bv[2],
                    v3 = acc[i][jp + t][3] + bv[3];
              if constexpr (EPI == EPI_GELU) { v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3); }
              pk[t][0] = pack2bf(v0, v1);
              pk[t][1] = pack2bf(v2, v3);
              acc[i][jp + t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // lanes of 16-lane row g hold columns 4 g .. 4 g + 3 of each tile; after the swaps: g = 0 / 2 hold eight
            // columns of tile jp (their own four + the next row's), g = 1 / 3 eight columns of tile jp + 1
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            const int g = lane >> 4;
            const int col = nn * BN3 + wn * 64 + (jp + (g & 1)) * 16 + 8 * (g >> 1);
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            if (GWW_G3_ABL & 1) asm volatile("" :: "v"(o));
            else *reinterpret_cast<u32x4*>(Cb + m * N + col) = o;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + (lane >> 4) * 4;
            f32x4 v = acc[i][j] + bvj[j];
            if (GWW_G3_ABL & 1) asm volatile("" :: "v"(v));
            else epilogue_store4<EPI, true>(v, m, nn * BN3 + nl, 0x7fffffffffffffffL, N, nullptr, resid, nullptr, C, 0, 0);
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      // the stores sit behind the pieces of tile it + 3: group 0 still has the waits of tiles it + 1 .. it + 3 ahead that may
      // leave them in flight, group 1 (whose wait for tile it + 1 is already behind it) two
      stores_age = g1 ? 2 : 3;
    }
    if (!g1) wait_next(it);
  }
  if (!g1) __builtin_amdgcn_s_barrier();
}

static int pick_n_split(long panels, int tiles_n) {
  // enough workgroups to fill 256 CUs a few times over, but keep the n-loop long
  int s = 1;
  while (panels * s < 1024 && s < tiles_n && tiles_n % (s * 2) == 0) s *= 2;
  if (panels * s < 512 && tiles_n % 3 == 0 && s * 3 <= tiles_n) s *= 3;
  return s;
}

int launch_gemm_bf16(const void* A, long lda, const void* W, const float* bias, const float* resid,
                     const float* pos, void* C, long M, int N, int K, int epi, int rows_per_batch,
                     hipStream_t s, int rows_padded_256) {
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
  static const bool use_v4 = lab_int("GWW_GEMM_V4", 1) != 0;
  if (use_v4 && rows_padded_256) {   // 256 x 256 x 64 eight-phase form (gemm_v4.hip); -1: not its shape
    const int rc = launch_gemm_bf16_v4(A, lda, W, bias, resid, C, M, N, K, epi, s);
    if (rc != -1) return rc;
  }
  static const bool use_v3 = lab_int("GWW_GEMM_V3", 1) != 0;
  if (use_v3 && rows_padded_256 && N % BN3 == 0 && N <= 12288 && M >= 4096 &&
      (epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID)) {
    // wide panels (whisper-base / -small): 256 x 256 tiles.  A block's bias slice lives in 1536 floats of LDS: at most
    // six n-tiles per block; enough blocks for a few rounds over the 256 CUs
    const long panels = cdiv(M, BM3);
    const int tn3 = N / BN3;
    // column splits: a split's W slice should sit in one XCD's 4-MB L2 beside the streaming A panels (<= 1.5 MB), and its
    // bias slice in 1536 floats of LDS (<= 6 n-tiles)
    long fit = (3L << 19) / ((long)BN3 * K * 2);
    if (fit < 1) fit = 1;
    if (fit > 6) fit = 6;
    int n_split = tn3;
    for (int s2 = 1; s2 <= tn3; ++s2)
      if (tn3 % s2 == 0 && tn3 / s2 <= fit && (panels * s2 >= 768 || s2 == tn3)) { n_split = s2; break; }
    dim3 grid3((unsigned)(panels * n_split)), block3(512);
#define GWW_GEMM3_CASE(E)                                                                             \
  case E:                                                                                             \
    hipLaunchKernelGGL((k_gemm_bf16_v3<E>), grid3, block3, 0, s, (const unsigned short*)A, lda,       \
                       (const unsigned short*)W, bias, resid, C, M, N, K, tn3, n_split);              \
    break;
    switch (epi) {
      GWW_GEMM3_CASE(EPI_BIAS) GWW_GEMM3_CASE(EPI_GELU) GWW_GEMM3_CASE(EPI_RESID)
      default:
        return fail(GWW_ERR_ARG, "gemm_bf16: unknown epilogue %d", epi);
    }
#undef GWW_GEMM3_CASE
    GWW_LAUNCH_CHECK();
    return GWW_OK;
  }
  if (rows_padded_256 && N % BN2 == 0 && N <= 6144 && M >= 4096 && epi != EPI_CONV1) {
    // large-M path; the caller has padded A / C / resid to a multiple of 256 rows.
    // (EPI_CONV1 writes row m + 1 and keeps the bounds-checked kernel.)
    const long panels = cdiv(M, BM2);
    const int tn2 = N / BN2;
    int n_split = pick_n_split(panels, tn2);
    while (cdiv(tn2, n_split) * BN2 > 1536) ++n_split;   // a block's bias slice lives in 1536 floats of LDS
    dim3 grid2((unsigned)(panels * n_split)), block2(512);
#define GWW_GEMM2_CASE(E)                                                                             \
  case E:                                                                                             \
    hipLaunchKernelGGL((k_gemm_bf16_v2<E>), grid2, block2, 0, s, (const unsigned short*)A, lda,       \
                       (const unsigned short*)W, bias, resid, pos, C, M, N, K, rows_per_batch,        \
                       valid_rows, tn2, n_split);                                                     \
    break;
    switch (epi) {
      GWW_GEMM2_CASE(EPI_BIAS) GWW_GEMM2_CASE(EPI_GELU) GWW_GEMM2_CASE(EPI_RESID) GWW_GEMM2_CASE(EPI_CONV2)
      default:
        return fail(GWW_ERR_ARG, "gemm_bf16: unknown epilogue %d", epi);
    }
#undef GWW_GEMM2_CASE
    GWW_LAUNCH_CHECK();
    return GWW_OK;
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
  __shared__ __attribute__((aligned(16))) float tile[64][84];   // 84: rows stay 16-byte aligned, 4-way-free column walks
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const float* src = mel + (long)b * C * T;
  // 16 bytes per lane on both sides when the shapes allow it (T % 4 == 0, C % 8 == 0: the Whisper front end's 3000 x 80):
  // a lane reads four consecutive time samples of one mel bin and writes eight consecutive bins of one token
  const bool vec = (T & 3) == 0 && (C & 7) == 0 && t0 + 64 <= T;
  if (vec) {
    for (int i = threadIdx.x; i < C * 16; i += 256) {
      const int c = i >> 4, q = i & 15;
      const float4 v = *reinterpret_cast<const float4*>(src + (long)c * T + t0 + 4 * q);
      tile[4 * q][c] = v.x; tile[4 * q + 1][c] = v.y; tile[4 * q + 2][c] = v.z; tile[4 * q + 3][c] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < C * 64; i += 256) {
      const int c = i >> 6, tt = i & 63;
      tile[tt][c] = (t0 + tt < T) ? src[(long)c * T + t0 + tt] : 0.f;
    }
  }
  __syncthreads();
  OutT* dst = out + ((long)b * (T + 2) + 1 + t0) * C;
  if (vec && sizeof(OutT) == 2) {
    const int per_row = C >> 3;
    for (int i = threadIdx.x; i < 64 * per_row; i += 256) {
      const int tt = i / per_row, c = 8 * (i - tt * per_row);
      const float4 lo = *reinterpret_cast<const float4*>(&tile[tt][c]), hi = *reinterpret_cast<const float4*>(&tile[tt][c + 4]);
      u32x4 o = {pack2bf(lo.x, lo.y), pack2bf(lo.z, lo.w), pack2bf(hi.x, hi.y), pack2bf(hi.z, hi.w)};
      *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(dst) + (long)tt * C + c) = o;
    }
  } else {
    for (int i = threadIdx.x; i < 64 * C; i += 256) {
      const int tt = i / C, c = i - tt * C;
      if (t0 + tt < T) {
        if constexpr (sizeof(OutT) == 2) dst[(long)tt * C + c] = f2bf(tile[tt][c]);
        else dst[(long)tt * C + c] = tile[tt][c];
      }
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
  return launch_gemm_bf16(A, K, W, bias, resid, nullptr, C, M, N, K, epilogue, 0, (hipStream_t)stream,
                          M % 256 == 0);
}
