// gw_whisper_amd -- conv1 of the Whisper stem read straight from the HF feature layout (bf16 path).
//
//   c1[b, 1 + t, :] = gelu(bias + sum_{tap, c} W[:, c, tap] * mel[b, c, t + tap - 1])        (Conv1d k = 3, padding 1:
//   HF:models/whisper/modeling_whisper.py:619-620, called from Signal_vs_Noise/src/model.py:25 through WhisperEncoder)
//
// Until round 4 this was two launches: k_mel_to_tokens ([B, 80, 3000] fp32 -> token-major bf16 [B, 3002, 80]) and an
// A-stationary GEMM over overlapping rows of that copy (gemm_astat.hip, EPI_CONV1), 0.088 + 0.347 ms per 256 segments for
// an op that moves 0.84 GB (0.13 ms at the HBM rate) and needs 0.06 ms of the matrix pipe: one 256-row workgroup per CU
// whose A prologue, twelve k-tiles and three GELU epilogues run one after the other.  Here:
//
//   * W-STATIONARY: K = 240 is tiny, so each of the 8 waves keeps its 48 (d = 384, 768) or 64 (d = 512, 1024) output
//     channels' whole weight slice in registers as the A operand of v_mfma_f32_16x16x32_bf16 (96 / 128 VGPRs, loaded once
//     per workgroup) and the workgroup is persistent over (segment, 128-token chunk) items: no weight ring, no k-tile
//     barriers, one s_barrier per chunk.
//   * the transposition rides on the LDS write: the chunk's 80 x 130 mel samples are read along time (coalesced), packed
//     to bf16 pairs and written token-major (176-byte rows: the 16 token lanes of a ds_read_b128 group fall on distinct
//     16-byte slots), double-buffered -- the loads of chunk i + 1 are in flight while chunk i is computed.  The B operand
//     of k-step s is then ONE 16-byte read at (token + tap(s, lane)) * 176 + 2 c(s, lane): 80 % 8 == 0, so an 8-group
//     of k never straddles a tap.  k >= 240 (zero weights) reads tap 0's finite data.
//   * D[channel][token]: a lane holds four consecutive channels of one token -> bias + GELU + one 8-byte store per
//     16 x 16 tile, no transpose; the 8 waves of a workgroup fill the 768-byte (d = 384) rows between them.
//   * the zero rows 0 and T + 1 of every segment (Conv1d padding of conv2) are written by the first / last chunk.
#include "common.h"

#ifndef GWW_C1M_TABLE
#define GWW_C1M_TABLE 0   // 1 (round 4 experiment): GELU through a 1 024-interval Phi table in LDS -- more exact (7.4e-6) and 5 % SLOWER
                          // (0.276 against 0.262 ms per 256 segments): 96 random 8-byte LDS reads per lane and chunk cost more than the
                          // 192 quarter-rate transcendentals they replace
#endif

namespace gww {

namespace {
constexpr int C1M_TOK = 128;          // tokens per chunk
constexpr int C1M_ROWS = C1M_TOK + 2; // + the two halo samples
constexpr int C1M_RS = 176;           // bytes per tile row: 80 bf16 + pad, 11 x 16
constexpr int C1M_BUF = 132 * C1M_RS;
constexpr int C1M_C = 80, C1M_KPAD = 256;
constexpr int C1M_PAIRS = (C1M_C / 2) * C1M_ROWS;             // (channel pair, time) items per chunk
constexpr int C1M_NPF = (C1M_PAIRS + 511) / 512;
}  // namespace

template <int NCB>   // 16-channel blocks per wave
__global__ __launch_bounds__(512, 1) void k_conv1_mel(const float* __restrict__ mel, const unsigned short* __restrict__ W,
                                                      const float* __restrict__ bias, unsigned short* __restrict__ c1,
                                                      int T, int d, int chunks_per_seg, int n_items) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[2 * C1M_BUF];
#if GWW_C1M_TABLE
  // GELU by table: Phi on [-8, 8) in 1 024 intervals with its forward difference beside it -- gelu(x) = x (Phi_i + f dPhi_i),
  // linear interpolation error <= h^2 / 8 max |Phi''| = 7.4e-6 (the sigmoid-quintic form: 2.6e-5).  Seven plain VALU operations
  // and one 8-byte LDS read per value instead of seven + v_exp_f32 + v_rcp_f32 (quarter rate): the epilogue is what bounds
  // this kernel.  8 KB, filled once per (persistent) workgroup.
  __shared__ __attribute__((aligned(16))) float2 phi_tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 512) {
    const float x0 = -8.0f + i * (1.0f / 64.0f), x1 = x0 + (1.0f / 64.0f);
    const float p0 = 0.5f * (1.0f + erff(x0 * 0.70710678118654752440f)), p1 = 0.5f * (1.0f + erff(x1 * 0.70710678118654752440f));
    phi_tab[i] = make_float2(p0, p1 - p0);
  }
  auto gelu_tab = [&](float x) -> float {
    float t = fmaf(x, 64.0f, 512.0f);
    t = __builtin_amdgcn_fmed3f(t, 0.0f, 1023.9999f);
    const unsigned i = (unsigned)t;                 // (truncation: t >= 0)
    const float2 pd = phi_tab[i];
    return x * fmaf(__builtin_amdgcn_fractf(t), pd.y, pd.x);
  };
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int chb = (int)blockIdx.y * (128 * NCB) + wave * 16 * NCB;   // first channel of this wave

  // ---- stationary operands
  bf16x8 wf[NCB][8];
  f32x4 bv[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
      wf[cb][s] = *reinterpret_cast<const bf16x8*>(W + (long)(chb + 16 * cb + col) * C1M_KPAD + 32 * s + 8 * g);
    bv[cb] = *reinterpret_cast<const f32x4*>(bias + chb + 16 * cb + 4 * g);
  }
  int boff[8];   // tile offset of this lane's 8 k values of k-step s, relative to its token's row
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k0 = 32 * s + 8 * g;
    const int tap = k0 / C1M_C, c = k0 - tap * C1M_C;
    boff[s] = col * C1M_RS + (k0 < 3 * C1M_C ? tap * C1M_RS + 2 * c : 0);
  }

  // The chunk's samples: requested from CLAMPED addresses (no branch per load, nothing waits here), the out-of-range ones
  // (time -1 / T of the segment's edges, the tail of the item list) zeroed when they are packed into the tile.
  float pf[C1M_NPF][2];
  auto request = [&](int item) {
    const int b = item / chunks_per_seg, t0 = (item - b * chunks_per_seg) * C1M_TOK;
    const float* src = mel + (long)b * C1M_C * T;
#pragma unroll
    for (int i = 0; i < C1M_NPF; ++i) {
      const int e = i * 512 + tid;
      int cp = e / C1M_ROWS;
      const int ti = e - cp * C1M_ROWS;
      cp = cp < C1M_C / 2 ? cp : C1M_C / 2 - 1;
      int t = t0 - 1 + ti;
      t = t < 0 ? 0 : (t < T ? t : T - 1);
      const float* p = src + (long)(2 * cp) * T + t;
      pf[i][0] = p[0];
      pf[i][1] = p[T];
    }
  };
  auto deposit = [&](int buf, int item) {
    const int t0 = (item % chunks_per_seg) * C1M_TOK;
#pragma unroll
    for (int i = 0; i < C1M_NPF; ++i) {
      const int e = i * 512 + tid;
      const int cp = e / C1M_ROWS, ti = e - cp * C1M_ROWS;
      const int t = t0 - 1 + ti;
      const bool ok = t >= 0 && t < T;
      if (e < C1M_PAIRS)
        *reinterpret_cast<unsigned int*>(tile + buf * C1M_BUF + ti * C1M_RS + 4 * cp) = ok ? pack2bf(pf[i][0], pf[i][1]) : 0u;
    }
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  request(item);
  deposit(0, item);
  // Touch the stationary operands here: hipcc then waits for their loads in front of the loop.  Left pending into the loop
  // header, its counter model re-waits for them in every iteration with counts that also drain the chunk-(i + 1) requests.
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(wf[cb][s]));
    asm volatile("" : "+v"(bv[cb]));
  }
  __syncthreads();
  int cur = 0;
  for (; item < n_items; item += gridDim.x) {
    const int nxt_item = item + gridDim.x;
    const bool more = nxt_item < n_items;
    if (more) request(nxt_item);
    const int b = item / chunks_per_seg, tc = item - b * chunks_per_seg, t0 = tc * C1M_TOK;
    const unsigned char* tb_base = tile + cur * C1M_BUF;
    unsigned short* out = c1 + ((long)b * (T + 2) + 1 + t0) * d + chb + 4 * g;
    const int n_tok = T - t0 < C1M_TOK ? T - t0 : C1M_TOK;
    // (fully unrolled, no trip count: with a loop around the stores hipcc drains the chunk-(i + 1) loads in front of it)
#pragma unroll
    for (int tb = 0; tb < C1M_TOK / 16; ++tb) {
      bf16x8 bfr[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(tb_base + tb * 16 * C1M_RS + boff[s]);
      f32x4 acc[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][s], bfr[s], acc[cb], 0, 0, 0);
      const int tok = tb * 16 + col;
      if (tok < n_tok) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const f32x4 v = acc[cb] + bv[cb];
          // (x sigma(quintic): the epilogue is VALU-bound -- 96 values per lane per chunk against 3 072 cycles of MFMAs)
#if GWW_C1M_TABLE
          const u32x2 o = {pack2bf(gelu_tab(v[0]), gelu_tab(v[1])), pack2bf(gelu_tab(v[2]), gelu_tab(v[3]))};
#else
          const u32x2 o = {pack2bf(gelu_sig4(v[0]), gelu_sig4(v[1])), pack2bf(gelu_sig4(v[2]), gelu_sig4(v[3]))};
#endif
          *reinterpret_cast<u32x2*>(out + (long)tok * d + 16 * cb) = o;
        }
      }
    }
    // Conv1d padding rows of conv2: row 0 and row T + 1 of the segment
    if ((tc == 0 || tc == chunks_per_seg - 1) && lane < 4 * NCB) {
      unsigned short* z = c1 + ((long)b * (T + 2) + (tc == 0 ? 0 : T + 1)) * d + chb + 4 * lane;
      *reinterpret_cast<u32x2*>(z) = u32x2{0u, 0u};
      if (chunks_per_seg == 1) *reinterpret_cast<u32x2*>(c1 + ((long)b * (T + 2) + T + 1) * d + chb + 4 * lane) = u32x2{0u, 0u};
    }
    if (more) deposit(cur ^ 1, nxt_item);
    __syncthreads();
    cur ^= 1;
  }
}

bool conv1_mel_supported(int n_mels, int d, int kpad) {
  return n_mels == C1M_C && kpad == C1M_KPAD && (d == 384 || d == 512 || d == 768 || d == 1024);
}

// mel [B, 80, T] fp32 (HF input_features), W [d, 256] bf16 with k = tap * 80 + c (zero for k >= 240), bias [d] fp32
// -> c1 [B, T + 2, d] bf16 incl. its zero rows.
int launch_conv1_mel(const float* mel, const void* W, const float* bias, void* c1, int B, int T, int d, hipStream_t s) {
  GWW_REQUIRE(mel && W && bias && c1, "conv1_mel: NULL operand");
  GWW_REQUIRE(conv1_mel_supported(C1M_C, d, C1M_KPAD), "conv1_mel: d=%d unsupported", d);
  GWW_REQUIRE(T > 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)c1) & 7) == 0 && (((uintptr_t)bias) & 15) == 0,
              "conv1_mel: bad shape or alignment");
  if (B == 0) return GWW_OK;
  const int cps = (int)cdiv(T, C1M_TOK);
  const long n_items = (long)B * cps;
  GWW_REQUIRE(n_items < (1l << 30), "conv1_mel: too many chunks");
  const int ncb = (d % 192 == 0) ? 3 : 4, nsplit = d / (128 * ncb);
  const int cus = 256 / nsplit;
  dim3 grid((unsigned)(n_items < cus ? n_items : cus), (unsigned)nsplit);
  if (ncb == 3)
    hipLaunchKernelGGL(k_conv1_mel<3>, grid, dim3(512), 0, s, mel, (const unsigned short*)W, bias, (unsigned short*)c1, T, d,
                       cps, (int)n_items);
  else
    hipLaunchKernelGGL(k_conv1_mel<4>, grid, dim3(512), 0, s, mel, (const unsigned short*)W, bias, (unsigned short*)c1, T, d,
                       cps, (int)n_items);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_conv1_gelu_bf16(const float* mel, const float* conv1_w, const float* conv1_b, void* w_scratch_bf16,
                                   void* c1_out, int B, int T, int d, void* stream) {
  GWW_REQUIRE(mel && conv1_w && conv1_b && w_scratch_bf16 && c1_out, "gww_conv1_gelu_bf16: NULL argument");
  GWW_REQUIRE(B >= 0 && T > 0, "gww_conv1_gelu_bf16: bad shape B=%d T=%d", B, T);
  GWW_REQUIRE(conv1_mel_supported(80, d, 256), "gww_conv1_gelu_bf16: d=%d unsupported (384, 512, 768, 1024)", d);
  GWW_TRY(launch_pack_weight(conv1_w, w_scratch_bf16, 1, d, 80, 3, 256, 1.f, (hipStream_t)stream));
  return launch_conv1_mel(mel, w_scratch_bf16, conv1_b, c1_out, B, T, d, (hipStream_t)stream);
}
