// "A-stationary" bf16 MFMA GEMM for the K <= 512 contractions of the encoder
// (QKV, fc1, out_proj at whisper-tiny/base; conv1):   C[M,N] = epi(f(A)[M,K] @ W[N,K]^T + b)
//
// Why a second GEMM: at d = 384 these GEMMs have 77..307 FLOP per HBM byte, i.e. they
// sit at or below the gfx950 ridge (~400 FLOP/B).  A tile kernel that streams A tiles
// per N tile re-reads the A panel N/128 times; with 32 workgroups per XCD the reuse
// distance (9 MB) exceeds the 4 MB L2, so those re-reads go to Infinity Cache / HBM
// and set the speed.  Here the A panel never leaves the CU:
//
//   * one workgroup = 256 rows = 8 waves x 32 rows.  Each wave keeps ITS 32 rows of A
//     for the WHOLE K in registers as MFMA operand fragments (K/16 x 4 VGPRs = 96 at
//     K = 384), read from HBM once, in whole 128-byte lines (8 lanes per row), and
//     transposed into fragment order through a small wave-private LDS slice.
//   * fused prologue (AMODE_LN): the deferred residual add x_new = x + delta (delta = bf16
//     output of the previous out_proj / fc2 GEMM) and LayerNorm are folded into the GEMM in
//     ONE pass over x.  LayerNorm is applied algebraically: the operand is the raw shifted
//     row a = bf16(x_new - c_m) (c_m = mean of the row's first 32 values, so |a| ~ sigma), the
//     gain is folded into the packed weight W' = bf16(g * W) and
//         out[m][n] = rstd_m * (acc[m][n] - mean'_m * u[n]) + cb[n],
//         u[n] = sum_k W'[n][k],   cb[n] = bias[n] + sum_k b_ln[k] W[n][k]
//     with the exact fp32 row statistics mean'_m, rstd_m of (x_new - c_m) gathered in the same
//     pass (gww_ln_fold_weights builds W', u, cb).  The LayerNorm kernel, its bf16 round
//     trip, the second read of x and the residual read-modify-write of the GEMM epilogues
//     all disappear.
//   * only W moves through shared LDS: [128 n][64 k] tiles, global_load_lds (16 B/lane)
//     into a 7-deep ring (a whole n-tile ahead, so epilogue stores never block the ring), counted vmcnt + one raw s_barrier per k-tile; W stays L2
//     resident (<= 1.2 MB) and costs 16 KB per 1024 MFMA cycles per CU.
//   * v_mfma_f32_32x32x16_bf16 with swapped operands (D = W_tile . A_frag^T): the row m
//     sits on the lane.  The epilogue transposes each 32 x 64 output block through the
//     wave-private LDS slice so every global store instruction writes whole 128-byte
//     lines (8 line requests per instruction instead of 32: the scattered 16-byte
//     pieces of the accumulator layout were request-rate bound, profiles/r01_pmc_*).
//   * the (n-tile, k-tile) space is flattened: the ring never drains inside a block and
//     the stores of n-tile t overlap the loads / MFMAs of t+1.
#include "common.h"
#include "epilogue.h"

#include <stdlib.h>

namespace gww {

constexpr int AS_WAVES = 8;                      // waves per workgroup (32 rows each); 1 workgroup per CU
constexpr int AS_THREADS = AS_WAVES * 64;
constexpr int AS_BM = AS_WAVES * 32, AS_BN = 128, AS_BK = 64;
constexpr int AS_NST = 7;                       // W ring depth: a whole n-tile (6 k-tiles at K=384) in flight
constexpr int AS_D = AS_NST - 1;                // k-tiles in flight
constexpr int AS_W_BYTES = AS_BN * AS_BK * 2;   // 16 KB
constexpr int AS_GLDS = 16 / AS_WAVES;          // LDS-DMA pieces per thread per k-tile
constexpr int AS_STORES = 8;                    // 16-byte global stores per thread per n-tile
constexpr int AS_SLICE_STRIDE = 144;            // bytes per row of the wave-private [32][64] bf16 slice (+16 pad)
constexpr int AS_SLICE_BYTES = 32 * AS_SLICE_STRIDE;

__device__ __forceinline__ int as_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N>
__device__ __forceinline__ void as_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

enum : int { AMODE_BF16 = 0, AMODE_LN = 1 };

// ---- optional in-kernel phase stamps (diagnostic build only: -DGWW_STAMP); totals per phase in cycles
#ifdef GWW_STAMP
__device__ unsigned long long g_stamp[8];
#define STAMP_DECL unsigned long long _t0 = __builtin_amdgcn_s_memtime(); unsigned long long _acc[6] = {0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                   \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long _t1 = __builtin_amdgcn_s_memtime();   \
    _acc[i] += _t1 - _t0;                                          \
    _t0 = _t1;                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#define STAMP_FLUSH                                                                  \
  if (lane == 0) {                                                                   \
    for (int _q = 0; _q < 6; ++_q) atomicAdd(&g_stamp[_q], _acc[_q]);                \
    atomicAdd(&g_stamp[7], 1ull);                                                    \
  }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// KT = K / 64 (k-tiles); A fragments: af[4 S + j] holds k = 64 S + 32 hh + 8 j .. +7 of row m.
template <int EPI, int AMODE, int KT, bool HAS_DELTA>
__global__ __launch_bounds__(AS_THREADS, 1) void k_gemm_astat(const void* Aany, long lda,
                                                       const unsigned short* delta,
                                                       float* x_out,
                                                       const float* __restrict__ ln_u,
                                                       const float* __restrict__ ln_cb,
                                                       const unsigned short* __restrict__ W,
                                                       const float* __restrict__ bias, unsigned short* __restrict__ C,
                                                       long M, int N, int tiles_n, int n_split,
                                                       int rows_per_batch, int valid_rows, long c_panel_rows) {
  constexpr int K = KT * 64;
  constexpr int OFF_BIAS = AS_NST * AS_W_BYTES;
  constexpr int OFF_U = OFF_BIAS + 1536 * 4;      // LN mode: column sums u[n] of the gain-folded weight
  constexpr int OFF_SLICE = OFF_U + (AMODE == AMODE_LN ? 1536 * 4 : 0);
  __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_SLICE + AS_WAVES * AS_SLICE_BYTES];
  float* lds_bias = reinterpret_cast<float*>(lds + OFF_BIAS);
  float* lds_u = reinterpret_cast<float*>(lds + OFF_U);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;

  const int panel = blockIdx.x / n_split, split = blockIdx.x - panel * n_split;
  const int nt0 = (int)((long)split * tiles_n / n_split), nt1 = (int)((long)(split + 1) * tiles_n / n_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const long m_base = (long)panel * AS_BM + wave * 32;   // first row of this wave
  const int total = (nt1 - nt0) * KT;
  unsigned char* slice = lds + OFF_SLICE + wave * AS_SLICE_BYTES;   // wave-private transpose buffer
  // coalesced row/chunk roles: one instruction covers 8 rows x 128 B
  const int crow = lane >> 3, cchunk = lane & 7;

  // ---- W ring: per-lane source offsets of the two 1-KiB pieces this wave stages per k-tile
  long w_off[AS_GLDS];
#pragma unroll
  for (int j = 0; j < AS_GLDS; ++j) {
    const int row = 8 * (AS_GLDS * wave + j) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    w_off[j] = (long)row * K + chunk * 8;
  }
  auto issue = [&](int it) {
    const int stage = it % AS_NST;
    const int nn = nt0 + it / KT, k0 = (it % KT) * AS_BK;
    unsigned char* sw = lds + stage * AS_W_BYTES + (AS_GLDS * wave) * 1024;
    const unsigned short* wb = W + (long)nn * AS_BN * K + k0;
#pragma unroll
    for (int j = 0; j < AS_GLDS; ++j)
      __builtin_amdgcn_global_load_lds((g_ptr)(wb + w_off[j]), (lds_ptr)(sw + j * 1024), 16, 0, 0);
  };

  if constexpr (AMODE == AMODE_LN) {
    for (int i = tid; i < (nt1 - nt0) * AS_BN; i += AS_THREADS) {
      lds_bias[i] = ln_cb[nt0 * AS_BN + i];
      lds_u[i] = ln_u[nt0 * AS_BN + i];
    }
  } else {
    for (int i = tid; i < (nt1 - nt0) * AS_BN; i += AS_THREADS) lds_bias[i] = bias ? bias[nt0 * AS_BN + i] : 0.f;
  }
  STAMP_DECL
  // ---- A fragments (whole K) for this wave's 32 rows
  bf16x8 af[KT * 4];
  float row_rstd = 1.f, row_mean = 0.f;   // AMODE_LN: statistics of this lane's row m (accumulator layout)
  long grow[4];   // clamped global rows this lane touches in the coalesced passes
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long g = m_base + 8 * i + crow;
    grow[i] = g < M ? g : M - 1;
  }
  if constexpr (AMODE == AMODE_BF16) {
    const unsigned short* A = reinterpret_cast<const unsigned short*>(Aany);
#pragma unroll
    for (int S = 0; S < KT; ++S) {
      u32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(A + grow[i] * lda + 64 * S + 8 * cchunk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<u32x4*>(slice + (8 * i + crow) * AS_SLICE_STRIDE + cchunk * 16) = v[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 u = *reinterpret_cast<const u32x4*>(slice + r * AS_SLICE_STRIDE + (4 * hh + j) * 16);
        asm volatile("" : "+v"(u)::"memory");   // materialise before the slice is overwritten
        af[4 * S + j] = __builtin_bit_cast(bf16x8, u);
      }
    }
  } else {
    // x_new = x (+ delta) -> written back; operand a = bf16(x_new - c); exact fp32 statistics of
    // (x_new - c) for the algebraic LayerNorm (HF:modeling_whisper.py:392,402; eps 1e-5).
    // Lane (crow, cchunk) owns float4 #cchunk of a 32-float half slice of rows 8 i + crow.
    const float* X = reinterpret_cast<const float*>(Aany);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, cshift[4];
    // lda == K here (checked by the launcher): one base pointer per row group, every other offset is an immediate
    // (48 / 64 separate 64-bit addresses were what spilled in this prologue)
    const float* xrow[4];
    const unsigned short* drow[4];
    float* orow_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xrow[i] = X + grow[i] * K + 4 * cchunk;
      drow[i] = delta + grow[i] * K + 4 * cchunk;
      orow_[i] = x_out + grow[i] * K + 4 * cchunk;
    }
#pragma unroll
    for (int S = 0; S < KT; ++S) {
      // one 32-float half slice of a k-tile per group: 4 (+4) loads issued back to back (branch-free: HAS_DELTA is a
      // template flag), then consumed.  (Whole k-tiles -- 8 + 8 loads, 48 registers in flight beside the growing A
      // panel -- pushed the LayerNorm variants past 256 registers: 42 - 153 spilled VGPRs, some reloaded in the main
      // loop; with two waves per SIMD the extra round trips hide behind the other wave.)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        __builtin_amdgcn_sched_barrier(0);
        float4 xv[4];
        u32x2 dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xv[i] = *reinterpret_cast<const float4*>(xrow[i] + 64 * S + 32 * h2);
          if constexpr (HAS_DELTA) dv[i] = *reinterpret_cast<const u32x2*>(drow[i] + 64 * S + 32 * h2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 v = xv[i];
          if constexpr (HAS_DELTA) {
            v.x += bf2f((unsigned short)(dv[i][0] & 0xffff));
            v.y += bf2f((unsigned short)(dv[i][0] >> 16));
            v.z += bf2f((unsigned short)(dv[i][1] & 0xffff));
            v.w += bf2f((unsigned short)(dv[i][1] >> 16));
            // x_new written back (rows past M are clamped duplicates of row M-1: same value, benign)
            *reinterpret_cast<float4*>(orow_[i] + 64 * S + 32 * h2) = v;
          }
          if (S == 0 && h2 == 0) {   // shift = mean of the row's first 32 values (any shift is exact algebra)
            float t = (v.x + v.y) + (v.z + v.w);
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            cshift[i] = t * (1.0f / 32.0f);
          }
          v.x -= cshift[i]; v.y -= cshift[i]; v.z -= cshift[i]; v.w -= cshift[i];
          s1[i] += (v.x + v.y) + (v.z + v.w);
          s2[i] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          u32x2 o = {pack2bf(v.x, v.y), pack2bf(v.z, v.w)};
          *reinterpret_cast<u32x2*>(slice + (8 * i + crow) * AS_SLICE_STRIDE + h2 * 64 + cchunk * 8) = o;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 u = *reinterpret_cast<const u32x4*>(slice + r * AS_SLICE_STRIDE + (4 * hh + j) * 16);
        asm volatile("" : "+v"(u)::"memory");
        af[4 * S + j] = __builtin_bit_cast(bf16x8, u);
      }
    }
    STAMP(0);
    // row statistics -> the lanes that own the row in the accumulator layout (via the slice memory)
    float* stat = reinterpret_cast<float*>(slice);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = s1[i], b = s2[i];
      a += __shfl_xor(a, 1, 64); b += __shfl_xor(b, 1, 64);
      a += __shfl_xor(a, 2, 64); b += __shfl_xor(b, 2, 64);
      a += __shfl_xor(a, 4, 64); b += __shfl_xor(b, 4, 64);
      const float mean = a * (1.0f / K);
      const float var = fmaxf(b * (1.0f / K) - mean * mean, 0.f);
      if (cchunk == 0) {
        stat[8 * i + crow] = rsqrtf(var + 1e-5f);
        stat[32 + 8 * i + crow] = mean;
      }
    }
    row_rstd = stat[r];
    row_mean = stat[32 + r];
    asm volatile("" : "+v"(row_rstd), "+v"(row_mean)::"memory");
  }
  // every ordinary load / store above is retired before the LDS-DMA ring starts counting
  as_wait_vmcnt<0>();
  STAMP(1);
#pragma unroll
  for (int p = 0; p < AS_D; ++p)
    if (p < total) issue(p);

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  const int ntiles = nt1 - nt0;
  for (int nti = 0; nti < ntiles; ++nti) {
#pragma unroll
    for (int S = 0; S < KT; ++S) {
      const int it = nti * KT + S;
      // retire this wave's pieces of tile `it`; tiles it+1 .. it+D-1 may stay in flight.  The
      // epilogue stores of the previous n-tile sit between them and the newest group at S == 0.
      if (it + AS_D <= total) {
        // D >= KT: every k-tile of the NEXT n-tile is already in flight when an epilogue issues its
        // stores, so those stores may stay outstanding for a whole n-tile (any smaller count is
        // merely conservative: vmcnt retires in order)
        // (k-tiles S >= D of this n-tile were issued AFTER those stores: no allowance for them -- K = 512)
        if (nti > 0 && S < AS_D) as_wait_vmcnt<AS_GLDS * (AS_D - 1) + AS_STORES>();
        else as_wait_vmcnt<AS_GLDS * (AS_D - 1)>();
      } else {
        as_wait_vmcnt<0>();   // tail: fewer groups in flight than the constant assumes
      }
      __builtin_amdgcn_s_barrier();
      STAMP(2);
      if (it + AS_D < total) issue(it + AS_D);
      const unsigned char* Ws = lds + (it % AS_NST) * AS_W_BYTES;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(Ws + as_swz(32 * t + r, 4 * hh + j));
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], af[4 * S + j], acc[t], 0, 0, 0);
      }
      STAMP(3);
    }
    // ---- epilogue: bias (+GELU) -> bf16 -> wave-private LDS transpose -> whole-line stores
    const int nn = nt0 + nti;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // EPI_DGELU: the incoming-gradient lines of this half are requested HERE, in front of the half's gelu' arithmetic (~1 300
      // VALU instructions), and awaited behind it.  As plain loads next to their use (round 3) hipcc's vmcnt(0) in front of
      // the multiply drained the six weight tiles in flight twice per n-tile with nothing to do meanwhile -- as long as the
      // n-tile's MFMAs.  (asm: hipcc sinks a plain load back to its use.)
      constexpr bool G_EARLY = EPI == EPI_DGELU && KT <= 6;   // (K = 512: the 16 extra live registers spill)
      u32x4 gq[4];
      if constexpr (G_EARLY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long orow = m_base + 8 * i + crow;
          const long mrow = orow < M ? orow : M - 1;
          const unsigned short* gp = delta + mrow * N + nn * AS_BN + 64 * half + 8 * cchunk;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gq[i]) : "v"(gp) : "memory");
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * half + tt;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if ((c & 1) == 0) __builtin_amdgcn_sched_barrier(0);   // keep the bias / u LDS reads from piling up in VGPRs
          const int nl = 32 * t + 8 * c + 4 * hh;
          const float4 bv = *reinterpret_cast<const float4*>(lds_bias + nti * AS_BN + nl);
          float v0 = acc[t][4 * c], v1 = acc[t][4 * c + 1], v2 = acc[t][4 * c + 2], v3 = acc[t][4 * c + 3];
          if constexpr (AMODE == AMODE_LN) {   // LayerNorm applied algebraically (see header)
            const float4 uv = *reinterpret_cast<const float4*>(lds_u + nti * AS_BN + nl);
            v0 = row_rstd * (v0 - row_mean * uv.x); v1 = row_rstd * (v1 - row_mean * uv.y);
            v2 = row_rstd * (v2 - row_mean * uv.z); v3 = row_rstd * (v3 - row_mean * uv.w);
          }
          v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
          if constexpr (EPI == EPI_GELU || EPI == EPI_CONV1) {
            v0 = gelu_fast(v0); v1 = gelu_fast(v1); v2 = gelu_fast(v2); v3 = gelu_fast(v3);
          }
          if constexpr (EPI == EPI_DGELU) {   // gelu'(z) = Phi(z) + z phi(z)  (train_ops.hip::k_gelu_bf16<true>)
            // dgelu_fast (common.h): 16 VALU operations where ocml erff + expf took about sixty -- this epilogue's VALU work
            // was five times the n-tile's MFMA time
            auto dg = [](float z) { return dgelu_fast(z); };
            v0 = dg(v0); v1 = dg(v1); v2 = dg(v2); v3 = dg(v3);
          }
          u32x2 o = {pack2bf(v0, v1), pack2bf(v2, v3)};
          *reinterpret_cast<u32x2*>(slice + r * AS_SLICE_STRIDE + (32 * tt + 8 * c + 4 * hh) * 2) = o;
          acc[t][4 * c] = 0.f; acc[t][4 * c + 1] = 0.f; acc[t][4 * c + 2] = 0.f; acc[t][4 * c + 3] = 0.f;
        }
      }
      if constexpr (G_EARLY)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(gq[0]), "+v"(gq[1]), "+v"(gq[2]), "+v"(gq[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 u = *reinterpret_cast<const u32x4*>(slice + (8 * i + crow) * AS_SLICE_STRIDE + cchunk * 16);
        long orow = m_base + 8 * i + crow;
        if constexpr (EPI == EPI_CONV1) {
          // rows m = b * rows_per_batch + t -> padded token-major row m + 1; t >= valid is a zero-pad row
          const int t = (int)(orow % rows_per_batch);
          if (t >= valid_rows) u = u32x4{0u, 0u, 0u, 0u};
          orow += 1;
        }
        // row-major [M][N], or panel-major [N/64][c_panel_rows][64]: there every store instruction
        // writes 1 KiB of contiguous memory and the consumer reads whole [rows][64] tiles
        unsigned short* dst = c_panel_rows ? C + ((long)(2 * nn + half) * c_panel_rows + orow) * 64 + 8 * cchunk
                                           : C + orow * N + nn * AS_BN + 64 * half + 8 * cchunk;
        if constexpr (EPI == EPI_DGELU) {   // times the incoming gradient (whole-line loads of the same [M, N] layout)
          u32x4 g;
          if constexpr (G_EARLY) {
            g = gq[i];
          } else {
            const long mrow = orow < M ? orow : M - 1;
            g = *reinterpret_cast<const u32x4*>(delta + mrow * N + nn * AS_BN + 64 * half + 8 * cchunk);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            u[q] = pack2bf(bf2f((unsigned short)(u[q] & 0xffff)) * bf2f((unsigned short)(g[q] & 0xffff)),
                           bf2f((unsigned short)(u[q] >> 16)) * bf2f((unsigned short)(g[q] >> 16)));
        }
        *reinterpret_cast<u32x4*>(dst) = u;
      }
    }
    STAMP(4);
  }
  STAMP_FLUSH
}

static int as_pick_split(long panels, int tiles_n) {
  int s = 1;
  while (panels * s < 768 && s * 2 <= tiles_n && tiles_n % (s * 2) == 0) s *= 2;
  if (panels * s < 512 && tiles_n % 3 == 0 && s * 3 <= tiles_n) s *= 3;
  return s;
}

// bf16 A [M, lda]                                     (ln_u == nullptr), or
// fp32 residual stream x [M, lda] (+ bf16 delta [M, lda], x_out written back) with fused LayerNorm:
// then W must be the gain-folded panel and ln_u / ln_cb the vectors built by gww_ln_fold_weights.
// C is bf16 [>= roundup(M,256) (+1 for conv1), N]: whole 256-row panels are stored unconditionally.
int launch_gemm_astat(const void* A, long lda, const void* delta, float* x_out, const float* ln_u,
                      const float* ln_cb, const void* W, const float* bias, void* C, long M, int N, int K,
                      int epi, int rows_per_batch, hipStream_t s, long c_panel_rows) {
  GWW_REQUIRE(A && W && C, "gemm_astat: NULL operand");
  GWW_REQUIRE(K == 256 || K == 384 || K == 512, "gemm_astat: K=%d unsupported", K);
  GWW_REQUIRE(N % AS_BN == 0 && N > 0, "gemm_astat: N=%d must be a multiple of 128", N);
  GWW_REQUIRE(lda % 8 == 0, "gemm_astat: lda must be a multiple of 8");
  GWW_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 15) == 0,
              "gemm_astat: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  const long panels = cdiv(M, AS_BM);
  const int tiles_n = N / AS_BN;
  // with a delta every workgroup of a row panel rewrites the same x_new values: keep the split minimal
  int n_split = (delta && epi != EPI_DGELU) ? 1 : as_pick_split(panels, tiles_n);
  while (cdiv(tiles_n, n_split) > 12) n_split *= 2;   // lds_bias / lds_u hold 1536 columns
  int valid_rows = 0;
  if (epi == EPI_CONV1) {
    GWW_REQUIRE(rows_per_batch > 2, "gemm_astat: conv1 epilogue needs rows_per_batch");
    valid_rows = rows_per_batch - 2;
  } else {
    rows_per_batch = 1;
  }
  const bool ln = ln_u != nullptr;
  GWW_REQUIRE(!ln || ln_cb, "gemm_astat: ln_u and ln_cb go together");
  GWW_REQUIRE(ln || epi == EPI_DGELU || (!delta && !x_out), "gemm_astat: delta / x_out need the LayerNorm prologue");
  GWW_REQUIRE(epi != EPI_DGELU || (delta && !x_out && !ln && !c_panel_rows), "gemm_astat: the gelu-backward epilogue takes the "
              "incoming gradient through `delta` (plain bf16 A operand, row-major C)");
  GWW_REQUIRE(!ln || lda == K, "gemm_astat: fused LayerNorm needs lda == K");
  GWW_REQUIRE(epi == EPI_DGELU || (delta == nullptr) == (x_out == nullptr), "gemm_astat: delta and x_out go together");
  GWW_REQUIRE(!x_out || (const void*)x_out != A, "gemm_astat: x_out must not alias the input stream");
  dim3 grid((unsigned)(panels * n_split)), block(AS_THREADS);
#define GWW_AS_LAUNCH(E, AM, KT, HD)                                                                        \
  hipLaunchKernelGGL((k_gemm_astat<E, AM, KT, HD>), grid, block, 0, s, A, lda, (const unsigned short*)delta,  \
                     x_out, ln_u, ln_cb, (const unsigned short*)W, bias, (unsigned short*)C, M, N, tiles_n, \
                     n_split, rows_per_batch, valid_rows, c_panel_rows)
#define GWW_AS_K2(E, AM, HD)                        \
  do {                                              \
    if (K == 384) GWW_AS_LAUNCH(E, AM, 6, HD);      \
    else if (K == 512) GWW_AS_LAUNCH(E, AM, 8, HD); \
    else GWW_AS_LAUNCH(E, AM, 4, HD);               \
  } while (0)
#define GWW_AS_K(E, AM)                              \
  do {                                               \
    if (AM == AMODE_LN && delta) GWW_AS_K2(E, AM, true); \
    else GWW_AS_K2(E, AM, false);                    \
  } while (0)
  if (ln) {
    if (epi == EPI_BIAS) GWW_AS_K(EPI_BIAS, AMODE_LN);
    else if (epi == EPI_GELU) GWW_AS_K(EPI_GELU, AMODE_LN);
    else return fail(GWW_ERR_ARG, "gemm_astat: LN-fused variant supports bias / GELU epilogues only");
  } else {
    if (epi == EPI_BIAS) GWW_AS_K(EPI_BIAS, AMODE_BF16);
    else if (epi == EPI_GELU) GWW_AS_K(EPI_GELU, AMODE_BF16);
    else if (epi == EPI_CONV1) GWW_AS_K(EPI_CONV1, AMODE_BF16);
    else if (epi == EPI_DGELU) GWW_AS_K2(EPI_DGELU, AMODE_BF16, false);
    else return fail(GWW_ERR_ARG, "gemm_astat: unsupported epilogue %d", epi);
  }
#undef GWW_AS_K
#undef GWW_AS_K2
#undef GWW_AS_LAUNCH
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_gemm_astat_bf16(const void* A, const void* delta, float* x_out, const float* ln_u,
                                   const float* ln_cb, const void* W, const float* bias, void* C, long M, int N,
                                   int K, int epilogue, void* stream) {
  GWW_REQUIRE(epilogue == 0 || epilogue == 1, "gww_gemm_astat_bf16: epilogue must be 0 (bias) or 1 (GELU)");
  GWW_REQUIRE((ln_u == nullptr) == (ln_cb == nullptr), "gww_gemm_astat_bf16: ln_u and ln_cb go together");
  static const long dbg_panel = lab_int("GWW_ASTAT_PANEL", 0);   // tuning aid (lab build)
  return launch_gemm_astat(A, K, delta, x_out, ln_u, ln_cb, W, bias, C, M, N, K, epilogue, 0, (hipStream_t)stream,
                           dbg_panel ? (M + 255) / 256 * 256 : 0);
}

#ifdef GWW_STAMP
extern "C" int gww_debug_stamps(unsigned long long* out8, int reset) {
  GWW_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gww::g_stamp), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0};
    GWW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gww::g_stamp), z, sizeof(z)));
  }
  return GWW_OK;
}
#endif
