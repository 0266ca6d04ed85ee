// k_gemm_bf16_v4: 256 x 256 x 64 bf16 GEMM tile for the wide panels of whisper-base / -small, C = epi(A W^T + bias).
// (HF: nn.Linear of WhisperEncoderLayer, modeling_whisper.py:379-413 -- q/k/v, out_proj, fc1, fc2 at d = 512 / 768.)
//
// What v3 (gemm_bf16.hip) taught: its k-tile is 32 deep, so an LDS-DMA piece (1 KiB per wave-instruction) is 16 rows of
// 64 bytes -- sixteen half lines for the address path -- and every k-tile pays a barrier pair per 32 MFMAs; with
// everything else removed its MFMA + barrier skeleton ran at 1 450-1 600 TFLOP/s, with the loads 940, shipped 720.
// This kernel is the eight-phase choreography of cdna_hip_programming.md section 5 laid out for a persistent n-loop:
//
//   * 8 waves = 4 (M) x 2 (N).  The 256 x 256 tile is four 128 x 128 QUADRANTS (A half x W half); in every phase all
//     waves work on the same quadrant, each on its 32 x 64 piece: 2 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16
//     x 2 k-steps = 16 MFMAs.  A k-tile (64 deep) is four phases, quadrant order (A0,W0) (A1,W0) (A1,W1) (A0,W1):
//     the fragment reads are 12 / 4 / 8 / 0 ds_read_b128 per wave (A0 stays in registers for phase 3, W0 for phase 1,
//     W1 for phase 3), so every half-tile (128 rows x 64 k = 16 KB) is read in exactly ONE phase: A0, W0 in phase 0, A1 in
//     phase 1, W1 in phase 2 -- which frees it for the LDS-DMA of the k-tile after next two phases later.
//   * two 64-KB buffers (k-tile parity), one half-tile (2 LDS-DMA pieces per thread, 8 rows x 128 B = eight FULL lines
//     each) requested per phase: phase 0: A1(t+1), 1: W1(t+1), 2: A0(t+2), 3: W0(t+2) -- five to six phases (1.3-1.5
//     k-tiles) ahead of their first read, four half-tiles in flight: s_waitcnt vmcnt(8) in phases 3, 0, 1 (nothing is
//     needed for phase 3), never 0 inside the loop.  An epilogue's stores sit in the same in-order queue: the three
//     waits behind an epilogue name them as younger operations.
//   * the two waves of a SIMD (wave w and w + 4) run ONE BARRIER apart: a phase is [fragment reads + DMA requests |
//     s_barrier | 16 MFMAs | s_barrier], so between two barriers one wave of every SIMD feeds the matrix pipe while its
//     partner loads.  A wait that retires a half-tile sits in front of the barrier that ends the phase BEFORE the
//     half-tile's first read (by the leading group), and the trailing group's wait is one barrier in front of that read.
//   * 128-byte LDS rows, chunk c of row R at c ^ ((R >> 1) & 7): the four 16-lane groups of a ds_read_b128 each cover a
//     whole 256-byte bank row (two rows x eight chunks) -- conflict-free; the swizzle is applied to the per-lane SOURCE
//     address of the LDS-DMA (the LDS image of a piece is lane-linear).
//   * D = W-fragment x A-fragment: a lane holds four consecutive output columns of one row; v_permlane16_swap pairs
//     two 16-column tiles so that every lane stores 16 bytes (64 contiguous bytes per row and instruction).
//
// Work item = (row panel, column split), panel-major and XCD-contiguous; at most one block per CU, whose k-tile stream
// runs through the epilogues of its tiles AND through its items.  Requires N % 256 == 0, N <= 3072 (the bias vector lives
// in LDS), K % 128 == 0, the row count padded to 256 by the caller (rows_padded_256).

#include "common.h"
#include "epilogue.h"

#include <stdlib.h>
#include <type_traits>

namespace gww {

#ifndef GWW_G4_DEEP
#define GWW_G4_DEEP 0   // measured: no gain (the ring wait is not what bounds the loop), off
#endif
#ifndef GWW_G4_ABL
#define GWW_G4_ABL 0   // diagnostic builds only (wrong results): 1 = no epilogue stores, 2 = no LDS-DMA / ring waits, 4 = no fragment reads
#endif

#ifdef GWW_G4_STAMP
// diagnostic build only: s_memtime ticks per phase and section, summed over all waves -- [group][phase][section]
// (section 0: fragment reads + DMA requests + ring wait + barrier + read wait, 1: the 16 MFMAs, 2: the closing barrier)
__device__ unsigned long long g_stamp_v4[32];
#define G4S_DECL unsigned long long _s4[12] = {0}; unsigned long long _t4 = __builtin_amdgcn_s_memtime();
#define G4S(P, C) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); _s4[(P) * 3 + (C)] += _n - _t4; _t4 = _n; }
#define G4S_FLUSH if (lane == 0) { for (int _q = 0; _q < 12; ++_q) atomicAdd(&g_stamp_v4[(g1 ? 12 : 0) + _q], _s4[_q]); atomicAdd(&g_stamp_v4[24 + (g1 ? 1 : 0)], 1ull); }
#else
#define G4S_DECL
#define G4S(P, C)
#define G4S_FLUSH
#endif

namespace {

constexpr int HT4 = 128 * 64 * 2;   // half-tile: 128 rows x 64 k, bf16
constexpr int BUF4 = 4 * HT4;       // A0 A1 W0 W1
constexpr int OFF_A4 = 0, OFF_W4 = 2 * HT4;
constexpr int BIAS4 = 3072;         // the whole bias vector lives in LDS: N <= 3072 (whisper-small's fc1)

template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I>
using ic = std::integral_constant<int, I>;

}  // namespace

template <int EPI>
__global__ __launch_bounds__(512, 1) void k_gemm_bf16_v4(const unsigned short* __restrict__ A, long lda,
                                                         const unsigned short* __restrict__ W,
                                                         const float* __restrict__ bias, const float* resid,
                                                         void* C, long M, int N, int K, int n_split, int tpi, int n_items,
                                                         const float* __restrict__ pos, int rows_per_batch, int n_real,
                                                         float* dump) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF4 + BIAS4 * 4];
  float* lds_bias = reinterpret_cast<float*>(lds + 2 * BUF4);   // the whole bias vector (N <= BIAS4)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;
  // EPI_CONV2 (the stride-2 convolution of the stem as a GEMM over overlapping rows, HF:modeling_whisper.py:619-624):
  // fp32 out = gelu(acc + bias) + pos[t], row m = b * rows_per_batch + t -> row b * (rows_per_batch - 1) + t, the
  // per-segment garbage row t = rows_per_batch - 1 and the rows past M go to the scratch row `dump` (every store is
  // issued: the ring waits count them); the weight panel is padded to N % 256 == 0 with zero rows, n_real = the true
  // width -- a W half that is all padding gets no MFMAs and no stores
  constexpr bool CONV2 = EPI == EPI_CONV2;
  constexpr bool BF16OUT = EPI != EPI_RESID && !CONV2;
  constexpr bool PRELOAD = EPI == EPI_RESID;
  unsigned short* __restrict__ const Cb = reinterpret_cast<unsigned short*>(C);
  float* const Cf = reinterpret_cast<float*>(C);   // (EPI_RESID: may alias resid)
  constexpr int SQ = BF16OUT ? 4 : 8;              // stores per thread and output QUADRANT

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const bool g1 = wave >= 4;
  const int nk = K >> 6;

  // ---- the work items of this block.  Item = (row panel, column split), panel-major (the splits of a panel run side by
  // side and share its A panel through the L2); the items are cut into eight contiguous chunks, one per XCD (blockIdx % 8),
  // and the blocks of an XCD walk their chunk with the stride of their number: one launch of at most one block per CU, the
  // k-tile stream of a block runs THROUGH its items (the ring requests the next item's first k-tiles under the last
  // MFMAs of this one: no pipeline fill and drain per item -- whisper-base's items are one or two tiles of eight k-tiles)
  int first_item, item_stride, n_my;
  {
    const int G = (int)gridDim.x, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int per = n_items >> 3, rem = n_items & 7;
    const int c0 = xcd * per + (xcd < rem ? xcd : rem), clen = per + (xcd < rem ? 1 : 0);
    item_stride = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    first_item = c0 + bi;
    n_my = (clen - bi + item_stride - 1) / item_stride;
  }
  if (n_my <= 0) return;   // (block-uniform, in front of every barrier: a CU count that is no multiple of 8 leaves such blocks)
  const int n_tiles = n_my * tpi;
  const int total = n_tiles * nk;
  // tile i of this block -> first row and column tile
  auto desc = [&](int i, long& m0, int& ncol) {
    const int j = i / tpi, nn_i = i - j * tpi;
    const int item = first_item + j * item_stride;
    const int panel = item / n_split, split = item - panel * n_split;
    m0 = (long)panel * 256;
    ncol = split * tpi + nn_i;
  };

  for (int i = tid; i < N; i += 512) lds_bias[i] = (bias && i < n_real) ? bias[i] : 0.f;

  // ---- LDS-DMA: half-tile h of an operand = 16 pieces of 8 rows; wave w requests pieces 2 w, 2 w + 1 (rows 16 w ..).
  // Lane l lands at row l >> 3, position l & 7 of its piece, which holds chunk (l & 7) ^ ((row >> 1) & 7) of that row.
  const unsigned short* a_src[2];
  const unsigned short* w_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * wave + 8 * j + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_src[j] = A + (long)row * lda + chunk * 8;
    w_src[j] = W + (long)row * K + chunk * 8;
  }
  const long a_half = 128 * lda, w_half = 128L * K;
  // half-tile `which` (0: A0, 1: A1, 2: W0, 3: W1) into buffer BUF; o = element offset of its k-tile in A resp. W
  auto stage = [&](auto buf_c, auto which_c, long o) {
    constexpr int BUF = decltype(buf_c)::value, which = decltype(which_c)::value;
    if (GWW_G4_ABL & 2) return;
    unsigned char* dst = lds + BUF * BUF4 + which * HT4 + wave * 2048;
    if constexpr (which < 2) {
      o += which * a_half;
      __builtin_amdgcn_global_load_lds((g_ptr)(a_src[0] + o), (lds_ptr)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((g_ptr)(a_src[1] + o), (lds_ptr)(dst + 1024), 16, 0, 0);
    } else {
      o += (which - 2) * w_half;
      __builtin_amdgcn_global_load_lds((g_ptr)(w_src[0] + o), (lds_ptr)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((g_ptr)(w_src[1] + o), (lds_ptr)(dst + 1024), 16, 0, 0);
    }
  };

  // ---- fragment read addresses: lane (r = l & 15, q = l >> 4) reads row r, chunk 4 ks + q of a 16-row fragment
  const int fr = lane & 15, fq = lane >> 4;
  const int lo0 = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);   // k-step 0; k-step 1 is lo0 ^ 64
  const unsigned char* a_rd[2][2];   // [buffer][k-step]
  const unsigned char* w_rd[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int lo = (GWW_G4_ABL & 4) ? 0 : (ks ? (lo0 ^ 64) : lo0);
      a_rd[b][ks] = lds + b * BUF4 + OFF_A4 + wr * 4096 + lo;
      w_rd[b][ks] = lds + b * BUF4 + OFF_W4 + wc * 8192 + lo;
    }

  f32x4 acc[2][2][2][4];   // [A half][W half][m-tile][n-tile]
  // PRE (bf16 outputs): A0 of the NEXT k-tile is read in phase 3 of this one (fragment reads per phase 8 / 4 / 8 / 4 instead
  // of 12 / 4 / 8 / 0); the fp32-residual form has no 16 registers left for the second A0 set
  constexpr bool PRE = BF16OUT;   // (conv2: its epilogue holds the position rows and the GELU results side by side -- no room for the second A0 set)
  // DEEP (with PRE): every half-tile is requested as early as its LDS slot allows -- two phases behind the slot's last read:
  // phase 0: W1(t+1), 1: A0(t+2), 2: W0(t+2), 3: A1(t+2) -- SIX phases (1.5 k-tiles) ahead of the first read, five half-tiles
  // in flight behind every wait (vmcnt(10)); the A panel comes from HBM, and the ring wait was where its latency showed
  constexpr bool DEEP = PRE && GWW_G4_DEEP;
  constexpr int RING = DEEP ? 10 : 8;   // LDS-DMA operations a wait leaves in flight: five resp. four half-tiles
  bf16x8 af0[PRE ? 2 : 1][2][2], af1[2][2], wf[4][2];

  // ---- epilogue of one quadrant, in the LOAD section of the phase behind its last MFMAs (the partner wave of the SIMD is
  // in its MFMA section): bias (+ GELU), pack, store; EPI_RESID: the fp32 residual tile was loaded INTO the accumulators
  // before the n-tile's first MFMA, so this is add-bias + store, and the next n-tile's residual is requested into the
  // registers just freed (asm loads: counted by hand with the ring, retired by the waits in front of their first MFMA)
  auto preload_resid = [&](auto ah_c, auto wh_c, long m0, int nn) {
    constexpr int ah = decltype(ah_c)::value, wh = decltype(wh_c)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = m0 + ah * 128 + wr * 32 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* p = resid + m * N + (nn * 256 + wh * 128 + wc * 64 + j * 16 + (lane >> 4) * 4);
        f32x4 r;   // (a plain rename: the ISA must show no copy between this load and the MFMA that reads it -- build() checks)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
        acc[ah][wh][i][j] = r;
      }
    }
  };
  // (m0, nn): the finished tile; (pre, m0n, nnn): fp32 residual -- the tile whose residual goes into the freed registers
  auto epi_quadrant = [&](auto ah_c, auto wh_c, long m0, int nn, int pre, long m0n, int nnn) {
    constexpr int ah = decltype(ah_c)::value, wh = decltype(wh_c)::value;
    f32x4 bvj[4];
    if constexpr (!CONV2) {   // (conv2 reads its bias tile by tile: the 16 registers hold the position rows there)
      const unsigned ba = (unsigned)(unsigned long long)(lds_ptr)(lds_bias + nn * 256 + wh * 128 + wc * 64 + (lane >> 4) * 4);
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                   "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(bvj[0]), "=&v"(bvj[1]), "=&v"(bvj[2]), "=&v"(bvj[3]) : "v"(ba));
    }
    const int cbase = nn * 256 + wh * 128 + wc * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = m0 + ah * 128 + wr * 32 + i * 16 + (lane & 15);
      if constexpr (BF16OUT) {
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          unsigned pk[2][2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 bv = bvj[jp + u];
            const f32x4 a = acc[ah][wh][i][jp + u];
            float v0 = a[0] + bv[0], v1 = a[1] + bv[1], v2 = a[2] + bv[2], v3 = a[3] + bv[3];
            if constexpr (EPI == EPI_GELU) { v0 = gelu_sig4(v0); v1 = gelu_sig4(v1); v2 = gelu_sig4(v2); v3 = gelu_sig4(v3); }
            pk[u][0] = pack2bf(v0, v1);
            pk[u][1] = pack2bf(v2, v3);
            acc[ah][wh][i][jp + u] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          // lanes of 16-lane row g hold columns 4 g .. 4 g + 3 of each tile; after the swaps g = 0 / 2 hold eight columns
          // of tile jp (their own four + the next row's), g = 1 / 3 eight columns of tile jp + 1
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          const int g = lane >> 4;
          const int col = cbase + (jp + (g & 1)) * 16 + 8 * (g >> 1);
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          if (GWW_G4_ABL & 1) asm volatile("" ::"v"(o));
          else if (GWW_G4_ABL & 64)   // diagnostic (wrong layout): the same stores as 8 rows x 128 B = eight FULL lines per instruction
            *reinterpret_cast<u32x4*>(Cb + (m - (lane & 15) + (jp >> 1) * 8 + (lane >> 3)) * N + cbase + (lane & 7) * 8) = o;
          else *reinterpret_cast<u32x4*>(Cb + m * N + col) = o;
        }
      } else if constexpr (CONV2) {
        // (a quadrant of padding columns -- cbase >= n_real -- still issues its stores, into the scratch row: the ring waits
        // count SQ stores per quadrant; its columns are folded back into the row so that pos / C are never read or written
        // past n_real)
        const bool padq = cbase >= n_real;
        const int cfold = padq ? cbase - (n_real - 64) : 0;
        const unsigned mu = (unsigned)m, rpb = (unsigned)rows_per_batch;   // (M < 2^31 is checked at launch)
        const unsigned b = mu / rpb;
        const int t = (int)(mu - b * rpb);
        const bool live = m < M && t < rows_per_batch - 1 && !padq;
        float* const orow = live ? Cf + ((long)b * (rows_per_batch - 1) + t) * n_real : dump;
        const float* prow = pos + (long)(live ? t : 0) * n_real;
        const int col0 = cbase - cfold + (lane >> 4) * 4;
        // Round 4: ONE exposed round trip per quadrant instead of four.  The position rows of this 16-row group (4 vectors) are
        // requested first; the GELU of all four column tiles runs IN PLACE in the accumulator registers meanwhile (36 VALU
        // operations per lane and tile: ~0.6 us, the latency of an L2 hit); then add + store.  (Two tiles at a time with the
        // wait right behind 0.2 us of arithmetic -- round 3's form, 8 registers of position values instead of 16 -- left every
        // pair waiting ~0.8 us with the partner wave's MFMA section long over.)
        f32x4 pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = *reinterpret_cast<const f32x4*>(prow + col0 + j * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 bj = *reinterpret_cast<const f32x4*>(lds_bias + nn * 256 + wh * 128 + wc * 64 + (lane >> 4) * 4 + j * 16);
          const f32x4 a = acc[ah][wh][i][j] + bj;
          acc[ah][wh][i][j] = f32x4{gelu_sig4(a[0]), gelu_sig4(a[1]), gelu_sig4(a[2]), gelu_sig4(a[3])};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(orow + col0 + j * 16) = acc[ah][wh][i][j] + pv[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[ah][wh][i][j] + bvj[j];
          if (GWW_G4_ABL & 1) asm volatile("" ::"v"(v));
          else *reinterpret_cast<f32x4*>(Cf + m * N + (cbase + j * 16 + (lane >> 4) * 4)) = v;
        }
      }
    }
    if constexpr (!BF16OUT) {
      if (PRELOAD && pre && !(GWW_G4_ABL & 1)) {
        preload_resid(ah_c, wh_c, m0n, nnn);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ah][wh][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // ---- the tiles around the stream position: cur = the tile of k-tile t, prev / next its neighbours in this block's order
  long m0c, m0p = 0, m0n = 0;
  int ncolc, ncolp = 0, ncoln = 0, tile_i = 0, has_next = n_tiles > 1;
  desc(0, m0c, ncolc);
  if (has_next) desc(1, m0n, ncoln);

  if constexpr (BF16OUT) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else if (!PRELOAD || (GWW_G4_ABL & 1)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    preload_resid(ic<0>{}, ic<0>{}, m0c, ncolc); preload_resid(ic<1>{}, ic<0>{}, m0c, ncolc);
    preload_resid(ic<1>{}, ic<1>{}, m0c, ncolc); preload_resid(ic<0>{}, ic<1>{}, m0c, ncolc);
  }

  // ---- prologue: what phases (-1, 2) .. would have requested: k-tile 0 whole, A0 and W0 of k-tile 1 (nk >= 2)
  {
    const long a0 = m0c * lda, w0 = (long)ncolc * 256 * K;
    stage(ic<0>{}, ic<0>{}, a0); stage(ic<0>{}, ic<2>{}, w0); stage(ic<0>{}, ic<1>{}, a0); stage(ic<0>{}, ic<3>{}, w0);
    stage(ic<1>{}, ic<0>{}, a0 + 64); stage(ic<1>{}, ic<2>{}, w0 + 64);
    if constexpr (DEEP) stage(ic<1>{}, ic<1>{}, a0 + 64);
  }
  if (!(GWW_G4_ABL & 2)) vm_wait<RING>();   // A0, W0 of k-tile 0 (and the residual preloads in front of the pieces)
  else vm_wait<0>();
  __builtin_amdgcn_s_barrier();
  // A0 of k-tile 0 for the leading phase (in the loop it is read one phase early, in phase 3 of the k-tile before)
  if constexpr (PRE) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) af0[0][i][ks] = *reinterpret_cast<const bf16x8*>(a_rd[0][ks] + i * 2048);
  }
  if (g1) __builtin_amdgcn_s_barrier();

  // ---- stream cursors (scalar, advanced inside the last MFMA section of a k-tile, where the wave's issue slots are idle):
  // k-tile t is k-tile kt of tile `cur`; (ao1, wo1) / (ao2, wo2) = element offsets of k-tiles t + 1 / t + 2 in A resp. W,
  // CLAMPED to the last k-tile of the block's stream -- past the end the ring keeps requesting that tile into slots nobody reads
  // again, so every wait of the loop sees the same queue (no tail cases; one vmcnt(0) before the kernel ends)
  int kt = 0;
  const long w_tile = 256L * K;
  long ao1 = m0c * lda + 64, wo1 = ncolc * w_tile + 64;   // nk >= 2
  int k2 = 2, tile2 = 0;
  long m02 = m0c;
  int ncol2 = ncolc;
  if (total <= 2) k2 = 1;
  else if (k2 == nk) { k2 = 0; tile2 = 1; m02 = m0n; ncol2 = ncoln; }
  long ao2 = m02 * lda + ((long)k2 << 6), wo2 = ncol2 * w_tile + ((long)k2 << 6);
  G4S_DECL

  // one wait of the ring: the youngest RING LDS-DMA operations (+ EXTRA stores of epilogue quadrants, when `epi`) stay in flight
  auto ring_wait = [&](int epi, auto extra_c) {
    constexpr int EXTRA = decltype(extra_c)::value;
    if (GWW_G4_ABL & 2) return;
    if (EXTRA > 0 && epi && !(GWW_G4_ABL & 1)) vm_wait<(RING + EXTRA > 63 ? 63 : RING + EXTRA)>();
    else vm_wait<RING>();
  };

  auto ktile = [&](int t, auto buf_c) {
    constexpr int B = decltype(buf_c)::value;
    const int last = (B == 1) && kt == nk - 1;        // this k-tile completes tile cur (nk is even: only in buffer 1)
    const int first = (B == 0) && kt == 0 && t > 0;   // the k-tile before completed tile prev
    const int second = (B == 1) && kt == 1 && t > 1;  // ... the k-tile before that one did

    auto mid = [&](auto p_c) {   // between the two sections of a phase
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      G4S(decltype(p_c)::value, 0)
      __builtin_amdgcn_s_setprio(1);
    };
    auto mfma16 = [&](auto ah_c, auto wh_c, auto&& tail) {
      constexpr int AH = decltype(ah_c)::value, WH = decltype(wh_c)::value;
      constexpr int PH = AH == 0 ? (WH == 0 ? 0 : 3) : (WH == 0 ? 1 : 2);
      // conv2 at d = 384: the second column tile's W1 half is zero padding -- its two phases keep their barriers and
      // requests (the stream stays uniform) but issue no MFMAs
      if (!(CONV2 && WH == 1 && ncolc * 256 + 128 >= n_real)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[AH][WH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], AH ? af1[i][ks] : af0[PRE ? B : 0][i][ks],
                                                                          acc[AH][WH][i][j], 0, 0, 0);
      }
      tail();
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      G4S(PH, 1)
      __builtin_amdgcn_s_barrier();
      G4S(PH, 2)
    };
    auto none = []() {};

    // ---- phase 0: quadrant (A0, W0).  W0 fragments (PRE: A0's were read in the phase before)
    if constexpr (B == 0) { if (first) epi_quadrant(ic<0>{}, ic<1>{}, m0p, ncolp, 1, m0c, ncolc); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j][ks] = *reinterpret_cast<const bf16x8*>(w_rd[B][ks] + j * 2048);
    if constexpr (!PRE) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) af0[0][i][ks] = *reinterpret_cast<const bf16x8*>(a_rd[B][ks] + i * 2048);
    }
    if constexpr (DEEP) {   // requests W1(t+1); retires A1(t)
      stage(ic<1 - B>{}, ic<3>{}, wo1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 0) ring_wait(first, ic<4 * SQ>{});
      else ring_wait(second, ic<SQ>{});
    } else {                // requests A1(t+1); retires A1(t)
      stage(ic<1 - B>{}, ic<1>{}, ao1);
      __builtin_amdgcn_sched_barrier(0);
      ring_wait(first, ic<(B == 0 ? 4 * SQ : 0)>{});
    }
    mid(ic<0>{});
    mfma16(ic<0>{}, ic<0>{}, none);

    // ---- phase 1: quadrant (A1, W0).  A1 fragments
    if constexpr (B == 1) { if (last) epi_quadrant(ic<0>{}, ic<0>{}, m0c, ncolc, has_next, m0n, ncoln); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) af1[i][ks] = *reinterpret_cast<const bf16x8*>(a_rd[B][ks] + HT4 + i * 2048);
    if constexpr (DEEP) {   // requests A0(t+2); retires W1(t)
      stage(ic<B>{}, ic<0>{}, ao2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 1) ring_wait(last, ic<SQ>{});
      else ring_wait(first, ic<4 * SQ>{});
    } else {                // requests W1(t+1); retires W1(t)
      stage(ic<1 - B>{}, ic<3>{}, wo1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 1) ring_wait(last, ic<SQ>{});
      else ring_wait(first, ic<3 * SQ>{});
    }
    mid(ic<1>{});
    mfma16(ic<1>{}, ic<0>{}, none);

    // ---- phase 2: quadrant (A1, W1).  W1 fragments
    if constexpr (B == 1) { if (last) epi_quadrant(ic<1>{}, ic<0>{}, m0c, ncolc, has_next, m0n, ncoln); }
    // (conv2: the fragments of an all-padding W1 half are read all the same -- skipping them under a branch made hipcc
    // spill 24 registers into the hand-counted vmcnt queue: 0.56 -> 0.89 ms)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j][ks] = *reinterpret_cast<const bf16x8*>(w_rd[B][ks] + HT4 + j * 2048);
    if constexpr (DEEP) {   // requests W0(t+2); retires A0(t+1) (read in phase 3)
      stage(ic<B>{}, ic<2>{}, wo2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 1) ring_wait(last, ic<2 * SQ>{});
      else ring_wait(first, ic<3 * SQ>{});
    } else {                // requests A0(t+2); retires A0(t+1) (PRE: read in phase 3; fp32 residual: the preloads of quadrant 2)
      stage(ic<B>{}, ic<0>{}, ao2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 1) ring_wait(last, ic<2 * SQ>{});
      else ring_wait(first, ic<2 * SQ>{});
    }
    mid(ic<2>{});
    mfma16(ic<1>{}, ic<1>{}, none);

    // ---- phase 3: quadrant (A0, W1).  PRE: A0 fragments of k-tile t + 1
    if constexpr (B == 1) { if (last) epi_quadrant(ic<1>{}, ic<1>{}, m0c, ncolc, has_next, m0n, ncoln); }
    if constexpr (PRE) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) af0[1 - B][i][ks] = *reinterpret_cast<const bf16x8*>(a_rd[1 - B][ks] + i * 2048);
    }
    if constexpr (DEEP) {   // requests A1(t+2); retires W0(t+1)
      stage(ic<B>{}, ic<1>{}, ao2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 1) ring_wait(last, ic<3 * SQ>{});
      else ring_wait(first, ic<2 * SQ>{});
    } else {                // requests W0(t+2); retires W0(t+1) (and, fp32 residual, the preloads of quadrant 3: no allowance)
      stage(ic<B>{}, ic<2>{}, wo2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (B == 1) ring_wait(last, ic<3 * SQ>{});
      else ring_wait(first, ic<(PRELOAD ? 0 : SQ)>{});
    }
    mid(ic<3>{});
    mfma16(ic<0>{}, ic<1>{}, [&]() {
      // cursors of the next k-tile (scalar work under the MFMAs)
      ++kt;
      if (kt == nk) {   // the next k-tile opens the next tile
        kt = 0;
        m0p = m0c; ncolp = ncolc; m0c = m0n; ncolc = ncoln;
        ++tile_i;
        has_next = tile_i + 1 < n_tiles;
        if (has_next) desc(tile_i + 1, m0n, ncoln);
      }
      ao1 = ao2; wo1 = wo2;
      if (t + 3 < total) {
        ++k2;
        if (k2 == nk) { k2 = 0; ++tile2; desc(tile2, m02, ncol2); }
        ao2 = m02 * lda + ((long)k2 << 6);
        wo2 = ncol2 * w_tile + ((long)k2 << 6);
      }
    });
  };

  for (int t = 0; t < total; t += 2) {
    ktile(t, ic<0>{});
    ktile(t + 1, ic<1>{});
  }
  vm_wait<0>();   // the clamped requests behind the last k-tile: no LDS-DMA may be in flight when the workgroup ends
  G4S_FLUSH
  epi_quadrant(ic<0>{}, ic<1>{}, m0p, ncolp, 0, 0L, 0);   // (the last k-tile's cursor step made the last tile `prev`)
  if (!g1) __builtin_amdgcn_s_barrier();
}

// returns GWW_OK after a launch, -1 when the shape is not this kernel's (the caller falls back to v3 / v2)
// force_split > 0 (the kernel-level entry point gww_gemm_bf16_v4_split): that column split (a divisor of N / 256)
// instead of the automatic choice -- an output element's accumulation order does not depend on it, which is what the
// split-invariance test checks bit for bit
int launch_gemm_bf16_v4(const void* A, long lda, const void* W, const float* bias, const float* resid, void* C, long M,
                        int N, int K, int epi, hipStream_t s, int force_split, const float* pos, int rows_per_batch,
                        int n_real, float* dump) {
  // (no lower bound on M: a segment's rows must not depend on how many segments share the launch -- the batch-independence
  // property the tests check bit for bit -- so the kernel choice may depend on N and K only)
  if (N % 256 != 0 || K % 128 != 0 || N > BIAS4 || M < 1) return -1;
  if (!(epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_CONV2)) return -1;
  if (epi == EPI_CONV2) {
    GWW_REQUIRE(M < 2147483647L, "gemm_bf16_v4: conv2 row count too large");
    GWW_REQUIRE(pos && rows_per_batch > 1 && n_real > 0 && n_real <= N && N - n_real < 256 && n_real % 128 == 0 && dump,
                "gemm_bf16_v4: conv2 epilogue needs pos, rows_per_batch, n_real (N - 255 .. N, a multiple of 128) and a scratch row");
  } else {
    n_real = N;
  }
  const long panels = cdiv(M, 256);
  const int tn = N / 256;
  // column splits: a split's W slice should sit in one XCD's 4-MB L2 beside the streaming A panels (<= 1.5 MB).  Among the
  // splits that allow it the one with the best load balance wins: the items of an XCD (an eighth of them) are dealt to
  // its CUs round-robin and all take the same time, so the launch lasts ceil(items per XCD / CUs per XCD) item times --
  // whisper-small's q/k/v at B = 64: 376 panels x 3 splits = 141 items per XCD on 32 CUs = 5 rounds for 4.4 (88 %),
  // x 9 splits = 423 = 14 rounds for 13.2 (94 %)
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, cus = 0;
    GWW_HIP(hipGetDevice(&dev));
    GWW_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));   // (no stream operation: capture-safe)
    n_cu = cus > 0 ? cus : 256;
  }
  long fit = (3L << 19) / (256L * K * 2);
  if (fit < 1) fit = 1;
  if (fit > 6) fit = 6;
  int n_split = tn;
  {
    const long cu_x = n_cu >= 8 ? n_cu / 8 : 1;
    double best = -1.0;
    for (int s2 = 1; s2 <= tn; ++s2) {
      if (tn % s2 != 0 || (tn / s2 > fit && s2 != tn)) continue;
      const long per_x = cdiv(panels * s2, 8);
      const double eff = (double)per_x / (double)(cdiv(per_x, cu_x) * cu_x);
      if (eff > best + 0.01) { best = eff; n_split = s2; }   // ties: the fewer, longer items
    }
  }
  if (force_split > 0) {
    GWW_REQUIRE(force_split <= tn && tn % force_split == 0, "gemm_bf16_v4: column split %d does not divide %d column tiles",
                force_split, tn);
    n_split = force_split;
  }
  const long n_items = panels * n_split;
  GWW_REQUIRE(n_items < 2147483647L, "gemm_bf16: grid too large");
  // one block per CU at most (140 KB of LDS each); a block's k-tile stream runs through its items
  dim3 grid((unsigned)(n_items < n_cu ? n_items : n_cu)), block(512);
  const int tpi = tn / n_split;
#define GWW_GEMM4_CASE(E)                                                                             \
  case E:                                                                                             \
    hipLaunchKernelGGL((k_gemm_bf16_v4<E>), grid, block, 0, s, (const unsigned short*)A, lda,         \
                       (const unsigned short*)W, bias, resid, C, M, N, K, n_split, tpi, (int)n_items, pos,     \
                       rows_per_batch, n_real, dump);                                                     \
    break;
  switch (epi) {
    GWW_GEMM4_CASE(EPI_BIAS) GWW_GEMM4_CASE(EPI_GELU) GWW_GEMM4_CASE(EPI_RESID) GWW_GEMM4_CASE(EPI_CONV2)
    default:
      return -1;
  }
#undef GWW_GEMM4_CASE
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

extern "C" int gww_gemm_bf16_v4_split(const void* A, const void* W, const float* bias, const float* resid, void* C, long M,
                                      int N, int K, int epilogue, int n_split, void* stream) {
  GWW_REQUIRE(epilogue >= 0 && epilogue <= 2, "gww_gemm_bf16_v4_split: epilogue must be 0, 1 or 2");
  GWW_REQUIRE(M % 256 == 0, "gww_gemm_bf16_v4_split: M must be a multiple of 256 (got %ld)", M);
  GWW_REQUIRE(epilogue != 2 || resid != nullptr, "gww_gemm_bf16_v4_split: residual epilogue needs resid");
  const int rc = gww::launch_gemm_bf16_v4(A, K, W, bias, resid, C, M, N, K, epilogue, (hipStream_t)stream, n_split, nullptr, 0, 0, nullptr);
  if (rc == -1) return gww::fail(GWW_ERR_ARG, "gww_gemm_bf16_v4_split: not a shape of the 256 x 256 x 64 kernel (N=%d K=%d)", N, K);
  return rc;
}

#ifdef GWW_G4_STAMP
extern "C" int gww_debug_stamps_v4(unsigned long long* out32, int reset) {
  GWW_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(gww::g_stamp_v4), sizeof(unsigned long long) * 32));
  if (reset) {
    unsigned long long z[32] = {0};
    GWW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gww::g_stamp_v4), z, sizeof(z)));
  }
  return GWW_OK;
}
#endif
