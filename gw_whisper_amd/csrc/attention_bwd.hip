// Flash-style attention BACKWARD (for the DoRA training step): given dO, recompute
// P = exp(S - LSE) from Q, K and the forward's log-sum-exp instead of storing the
// 1500 x 1500 scores, and form
//     D_i = sum_d dO_id O_id,  dP = dO V^T,  dS = P * (dP - D),
//     dV = P^T dO,  dK = dS^T Q,  dQ = dS K          (q is pre-scaled, so no extra factor).
// Layouts as the forward: qkv / dqkv [B, T, 3 d] (q | k | v), ctx / dctx [B, T, d], lse / D [B, H, T].
//
// Two kernels, no atomics (results are bitwise reproducible):
//   k_attn_bwd_dq   one workgroup = 128 queries of one (b, h), loops over 64-key tiles
//   k_attn_bwd_dkv  one workgroup = 128 keys of one (b, h), loops over 64-query tiles
// Both reuse the forward's dataflow: the "swapped" 32x32x16 products keep the stationary index
// (query resp. key) on the lane, the fp32 accumulator of the score-like products becomes, after a
// pairwise bf16 conversion, the B operand of the following product (no LDS round trip), and the
// transposed A operands come from row-major LDS tiles through ds_read_b64_tr_b16.  Tiles that are
// read both by rows and transposed are kept as two LDS images (chunk-XOR resp. bit-6-XOR swizzle).
#include "common.h"

#include <type_traits>

namespace gww {

namespace {
constexpr int DH = 64, TB = 128, KB = 64;
#ifndef GWW_ATTBWD_DMA
#define GWW_ATTBWD_DMA 1   // round 4: k_attn_bwd_dq stages K / V by LDS-DMA (no staging registers, no ds_write_b128) and keeps ONE
                           // image of K that serves the row reads of S = K Q^T and the transposed reads of dQ = dS K (dual_off
                           // below): 32 KB of LDS; k_attn_bwd_dkv likewise for Q and dO (237 -> 171 registers, 65 -> 33 KB).
                           // Measured (tools/attbwd_exp.py, rowdot + dq + dkv at B = 64, bit-identical results): 1.051 -> 1.028-1.034 ms;
                           // dq at three workgroups per CU (168 registers: the accumulator start values have to go and 15 registers
                           // still spill) was slower -- not kept
#endif
#ifndef GWW_ATTBWD_DQ_WAVES
#define GWW_ATTBWD_DQ_WAVES 2   // workgroups per CU of k_attn_bwd_dq
#endif
constexpr float kLog2e = 1.44269504088896340736f;
constexpr int TILE_BYTES = KB * DH * 2;   // 8 KB

__device__ __forceinline__ int row_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int tr_off(int row, int colbyte) { return row * 128 + (colbyte ^ (((row >> 1) & 1) << 6)); }
// One image of a row-major [64][64] bf16 tile for BOTH access patterns: 16-byte chunk c of row R sits at c ^ dual_x(R),
// dual_x = (k & 1) << 2 | k >> 1 with k = (R >> 1) & 7.
//  * ds_read_b128 row reads (16 lanes = rows R .. R + 15 of one chunk): the 8 values of k are a permutation of 0 .. 7, the
//    two rows of a k differ in bit 7 of the address -> 16 distinct 16-byte slots of the 256-byte bank sweep;
//  * ds_read_b64_tr_b16 (32 lanes = 4 rows x 64 bytes): rows R, R + 1 differ in bit 7, the row pairs (R, R + 1) / (R + 2, R + 3)
//    in bit 6 (k & 1 goes to chunk bit 2) and the lanes of a row fill 64 contiguous bytes -> 32 distinct 8-byte slots.
__device__ __forceinline__ int dual_x(int row) { const int k = (row >> 1) & 7; return ((k & 1) << 2) | (k >> 1); }
__device__ __forceinline__ int dual_off(int row, int chunk) { return row * 128 + ((chunk ^ dual_x(row)) << 4); }
__device__ __forceinline__ int dual_tr_off(int row, int colbyte) { return row * 128 + (colbyte ^ (dual_x(row) << 4)); }

__device__ __forceinline__ bf16x8 cvt8(const f32x16& a, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[base + j];
  return r;
}

// transposed A operand (32x32x16): rows r0 + {4 hh + (j&3) + 8 (j>>2)} of a row-major [64][64] bf16 tile,
// column 32 n + (lane & 31)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int row0, int n, int lane) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
  const int hh = lane >> 5;
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int cb = 64 * n + (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;
  const int r = row0 + 4 * hh + tr_q;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + tr_off(r, cb)));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + tr_off(r + 8, cb)));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
__device__ __forceinline__ bf16x8 tr_frag_dual(const unsigned char* tile, int row0, int n, int lane) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
  const int hh = lane >> 5;
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int cb = 64 * n + (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;
  const int r = row0 + 4 * hh + tr_q;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + dual_tr_off(r, cb)));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + dual_tr_off(r + 8, cb)));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
}  // namespace

// D[b, h, t] = sum_d dO[b, t, h*64 + d] * O[b, t, h*64 + d]
// nz[(b H + h) * n64 + t / 64] |= 1 when row t of dO (head h) is not all zero: the dQ / dK,dV kernels skip the
// query tiles whose dO is zero -- the classifier heads pool ONE token (Signal_vs_Noise/src/model.py:25-26), so
// in the last layer 23 of the 24 query tiles carry no gradient at all.
__global__ __launch_bounds__(256) void k_attn_rowdot(const unsigned short* __restrict__ o,
                                                     const unsigned short* __restrict__ d_o,
                                                     float* __restrict__ D, unsigned int* __restrict__ nz, int T,
                                                     int H, long rows) {
  // one 8-lane group per (row, head): 8 lanes x 8 bf16 = 64
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  if (g >= rows * H) return;
  const long row = g / H;
  const int h = (int)(g - row * H);
  const long off = row * (long)H * DH + h * DH + sub * 8;
  const u32x4 a = *reinterpret_cast<const u32x4*>(o + off);
  const u32x4 b = *reinterpret_cast<const u32x4*>(d_o + off);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s += bf2f((unsigned short)(a[j] & 0xffff)) * bf2f((unsigned short)(b[j] & 0xffff));
    s += bf2f((unsigned short)(a[j] >> 16)) * bf2f((unsigned short)(b[j] >> 16));
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  unsigned int any = ((b[0] | b[1] | b[2] | b[3]) & 0x7fff7fffu) != 0u;   // +-0 count as zero
  any |= __shfl_xor(any, 1, 64);
  any |= __shfl_xor(any, 2, 64);
  any |= __shfl_xor(any, 4, 64);
  if (sub == 0) {
    const long bidx = row / T;
    const int t = (int)(row - bidx * T);
    D[(bidx * H + h) * T + t] = s;
    // (a plain store: the flags are zeroed before the launch and every writer writes the same 1 -- 64 rows share a flag, and
    //  600 k device-scope atomics on 9 k addresses were half of this kernel's time)
    if (any) nz[(bidx * H + h) * ((T + KB - 1) / KB) + t / KB] = 1u;
  }
}

// ------------------------------------------------------------------------------------ dQ
// L2Q (the training step: q in log2 units, ssc = 1): the row constants enter as the INITIAL ACCUMULATORS of the two
// score-like products -- S' = q k - lse and dP' = dO v - D leave their MFMA chains ready, p = v_exp_f32(S') and
// dS = p dP' are two vector instructions per element (round 3: fma, exp, subtract, two multiplies and a key-mask
// select); the 1 / log2(e) of dS is applied once to the finished dQ; only the ragged last key tile masks its keys.
template <bool L2Q>
__global__ __launch_bounds__(256, GWW_ATTBWD_DQ_WAVES) void k_attn_bwd_dq(const unsigned short* __restrict__ qkv,
                                                        const unsigned short* __restrict__ dctx,
                                                        const float* __restrict__ lse,
                                                        const float* __restrict__ Dv,
                                                        unsigned short* __restrict__ dqkv, int T, int H,
                                                        int q_tiles, const unsigned int* __restrict__ nz,
                                                        float ssc, float gsc) {
  constexpr bool DMA = GWW_ATTBWD_DMA != 0;
  // per stage: K row image, K transposed-read image, V row image (DMA: one dual-use K image + the V row image)
  constexpr int IMGS = DMA ? 2 : 3;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * IMGS * TILE_BYTES];   // 48 KB (DMA: 32 KB)
  auto Kr = [&](int buf) -> unsigned char* { return lds + (buf * IMGS + 0) * TILE_BYTES; };
  auto Kt = [&](int buf) -> unsigned char* { return lds + (buf * IMGS + (DMA ? 0 : 1)) * TILE_BYTES; };
  auto Vr = [&](int buf) -> unsigned char* { return lds + (buf * IMGS + IMGS - 1) * TILE_BYTES; };
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware work order (attention.hip): the query tiles of one (b, h) stream the same K / V -- keep them on one XCD
  const unsigned nblk = gridDim.x, per = nblk >> 3;
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
  const int qt = wid % q_tiles, bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H, d = H * DH;
  const long rs = 3L * d;
  const unsigned short* base = qkv + (long)b * T * rs;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;
  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * TB + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;
  {   // dO of these 128 queries all zero (wave-uniform: the flags are per block): dQ = 0, nothing to stream
    const int n64 = (T + KB - 1) / KB;
    const unsigned int* f = nz + (long)bh * n64;
    const unsigned int live = f[2 * qt] | (2 * qt + 1 < n64 ? f[2 * qt + 1] : 0u);
    if (!live) {
      if (q_row < T) {
        unsigned short* orow = dqkv + ((long)b * T + q_row) * rs + h * DH;
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x2*>(orow + 8 * c + 4 * hh) = u32x2{0u, 0u};
      }
      return;
    }
  }

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * rs + 16 * s + 8 * hh);
    dof[s] = *reinterpret_cast<const bf16x8*>(dctx + ((long)b * T + q_ld) * d + h * DH + 16 * s + 8 * hh);
  }
  const float lse_q = lse[((long)b * H + h) * T + q_ld] * kLog2e;
  const float D_q = Dv[((long)b * H + h) * T + q_ld];

  int st_row[2], st_chunk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    st_row[i] = c >> 3;
    st_chunk[i] = c & 7;
  }
  u32x4 rk[2], rv[2];
  auto gload = [&](int kt) {
    if constexpr (DMA) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kt * KB + st_row[i];
      if (key >= T) key = T - 1;
      rk[i] = *reinterpret_cast<const u32x4*>(kp + (long)key * rs + st_chunk[i] * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(vp + (long)key * rs + st_chunk[i] * 8);
    }
  };
  auto lstore = [&](int buf) {
    if constexpr (DMA) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(Kr(buf) + row_off(st_row[i], st_chunk[i])) = rk[i];
      *reinterpret_cast<u32x4*>(Kt(buf) + tr_off(st_row[i], st_chunk[i] * 16)) = rk[i];
      *reinterpret_cast<u32x4*>(Vr(buf) + row_off(st_row[i], st_chunk[i])) = rv[i];
    }
  };
  // DMA: piece j of wave w = rows 8 (2 w + j) .. + 7 of the tile, 16 bytes per lane (position lane & 7 of row lane >> 3); the
  // images stay lane-linear in LDS and the swizzles are applied to the per-lane SOURCE chunk (an XOR is its own inverse).
  // Rows past T - 1 read row T - 1 (their scores are masked / their dS rows multiply nothing that is kept).
  // (32-bit per-lane byte offsets + a wave-uniform tile base: 64-bit pointers per piece spilled at three workgroups per CU)
  unsigned koff[2], voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
    koff[j] = (unsigned)(row * (int)rs * 2 + ((pos ^ dual_x(row)) << 4));
    voff[j] = (unsigned)(row * (int)rs * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
  }
  auto dma = [&](int kt, int buf) {
    if constexpr (!DMA) return;
    const unsigned char* kb = reinterpret_cast<const unsigned char*>(kp + (long)kt * KB * rs);
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(vp + (long)kt * KB * rs);
    unsigned char* dk = Kr(buf) + 2 * wave * 1024;
    unsigned char* dv = Vr(buf) + 2 * wave * 1024;
    if ((kt + 1) * KB <= T) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_global_load_lds((g_ptr)(kb + koff[j]), (lds_ptr)(dk + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(vb + voff[j]), (lds_ptr)(dv + j * 1024), 16, 0, 0);
      }
    } else {   // ragged last tile: rows past T - 1 read row T - 1
      const int last_row = T - 1 - kt * KB;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
        const int rc = row < last_row ? row : last_row;
        const unsigned ko = (unsigned)(rc * (int)rs * 2 + ((pos ^ dual_x(row)) << 4));
        const unsigned vo = (unsigned)(rc * (int)rs * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
        __builtin_amdgcn_global_load_lds((g_ptr)(kb + ko), (lds_ptr)(dk + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(vb + vo), (lds_ptr)(dv + j * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 dqt[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) dqt[n][j] = 0.f;

  const int n_kt = (T + KB - 1) / KB;
  // the row constants as accumulator start values (the query sits on the lane: all 16 registers hold the same value)
  f32x16 c_lse, c_D;
  // (at three workgroups per CU = 168 registers they are subtracted per element instead: 32 registers for 64 v_sub per tile)
  constexpr bool CD = L2Q && GWW_ATTBWD_DQ_WAVES < 3;
#pragma unroll
  for (int j = 0; j < 16; ++j) { c_lse[j] = CD ? -lse_q : 0.f; c_D[j] = CD ? -D_q : 0.f; }
  gload(0);
  lstore(0);
  dma(0, 0);
  __syncthreads();   // (DMA: hipcc drains the LDS-DMA in front of the barrier)
  // (the stage is a compile-time constant of the tile body -- the loop runs over tile PAIRS -- so that every LDS address is a
  //  per-lane base + immediate)
  auto tile = [&](int kt, auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
    if (kt + 1 < n_kt) { gload(kt + 1); dma(kt + 1, buf ^ 1); }
    const bool mask_tile = kt == n_kt - 1 && (T % KB) != 0;   // wave-uniform
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x16 st, dpt;
      if constexpr (CD) {
        st = c_lse;
        dpt = c_D;
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { st[j] = 0.f; dpt[j] = 0.f; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kr(buf) + (DMA ? dual_off(32 * g + r, 2 * s + hh) : row_off(32 * g + r, 2 * s + hh)));
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vr(buf) + row_off(32 * g + r, 2 * s + hh));
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dpt, 0, 0, 0);
      }
      // dS^T = P^T * (dP^T - D), P^T = exp(S^T - LSE); keys >= T contribute nothing
      if constexpr (L2Q) {
        if (mask_tile) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
            st[j] = (key < T) ? __builtin_amdgcn_exp2f(CD ? st[j] : st[j] - lse_q) * (CD ? dpt[j] : dpt[j] - D_q) : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) st[j] = __builtin_amdgcn_exp2f(CD ? st[j] : st[j] - lse_q) * (CD ? dpt[j] : dpt[j] - D_q);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
          const float p = (key < T) ? __builtin_amdgcn_exp2f(fmaf(st[j], ssc, -lse_q)) : 0.f;
          st[j] = p * (dpt[j] - D_q);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsf = cvt8(st, 8 * s);
#pragma unroll
        for (int n = 0; n < 2; ++n)
          dqt[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(DMA ? tr_frag_dual(Kt(buf), 32 * g + 16 * s, n, lane) : tr_frag(Kt(buf), 32 * g + 16 * s, n, lane), dsf, dqt[n], 0, 0, 0);
      }
    }
    if (kt + 1 < n_kt) lstore(buf ^ 1);
    __syncthreads();
  };
  {
    int kt = 0;
    for (; kt + 1 < n_kt; kt += 2) {
      tile(kt, std::integral_constant<int, 0>{});
      tile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < n_kt) tile(kt, std::integral_constant<int, 0>{});
  }
  if (q_row < T) {
    unsigned short* orow = dqkv + ((long)b * T + q_row) * rs + h * DH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x2 o = {pack2bf(dqt[n][4 * c] * gsc, dqt[n][4 * c + 1] * gsc), pack2bf(dqt[n][4 * c + 2] * gsc, dqt[n][4 * c + 3] * gsc)};
        *reinterpret_cast<u32x2*>(orow + 32 * n + 8 * c + 4 * hh) = o;
      }
  }
}

// ------------------------------------------------------------------------------------ dK, dV
// L2Q: -lse log2(e) and -D of the tile's 64 queries are staged NEGATED and read from LDS straight into the accumulators of
// S and dP (the query rows sit in the registers here); dS = p dP' without its 1 / log2(e), which the finished dK gets once.
#ifndef GWW_ATTBWD_DKV_WAVES
#define GWW_ATTBWD_DKV_WAVES 2   // workgroups per CU of k_attn_bwd_dkv (DMA form: 171 registers, 33 KB of LDS; 3 per CU = 168 registers, 6 spilled:
                                 // 1.021 against 1.028 - 1.034 ms for rowdot + dq + dkv, tools/attbwd_exp.py -- not kept)
#endif
template <bool L2Q>
__global__ __launch_bounds__(256, GWW_ATTBWD_DKV_WAVES) void k_attn_bwd_dkv(const unsigned short* __restrict__ qkv,
                                                         const unsigned short* __restrict__ dctx,
                                                         const float* __restrict__ lse,
                                                         const float* __restrict__ Dv,
                                                         unsigned short* __restrict__ dqkv, int T, int H,
                                                         int k_tiles, const unsigned int* __restrict__ nz,
                                                         float ssc, float gsc) {
  // per stage: Q row image, Q transposed-read image, dO row image, dO transposed-read image, lse[64], D[64]
  // (DMA, round 4: ONE dual-use image each for Q and dO -- dual_off above -- staged by LDS-DMA: 33 instead of 65 KB, no staging
  //  registers, no ds_write_b128)
  constexpr bool DMA = GWW_ATTBWD_DMA != 0;
  constexpr int IMG = DMA ? 2 : 4;
  constexpr int STAGE = IMG * TILE_BYTES + 512;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  auto Qr = [&](int buf) -> unsigned char* { return lds + buf * STAGE; };
  auto Qt = [&](int buf) -> unsigned char* { return lds + buf * STAGE + (DMA ? 0 : TILE_BYTES); };
  auto Or = [&](int buf) -> unsigned char* { return lds + buf * STAGE + (DMA ? 1 : 2) * TILE_BYTES; };
  auto Ot = [&](int buf) -> unsigned char* { return lds + buf * STAGE + (DMA ? 1 : 3) * TILE_BYTES; };
  auto Ls = [&](int buf) -> float* { return reinterpret_cast<float*>(lds + buf * STAGE + IMG * TILE_BYTES); };
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned nblk = gridDim.x, per = nblk >> 3;   // XCD-aware order: the key tiles of one (b, h) stream the same Q / dO
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
  const int ktile = wid % k_tiles, bh = wid / k_tiles;
  const int b = bh / H, h = bh - b * H, d = H * DH;
  const long rs = 3L * d;
  const unsigned short* base = qkv + (long)b * T * rs;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;
  const unsigned short* dop = dctx + (long)b * T * d + h * DH;
  const float* lsep = lse + ((long)b * H + h) * T;
  const float* Dp = Dv + ((long)b * H + h) * T;
  const int r = lane & 31, hh = lane >> 5;
  const int key_row = ktile * TB + wave * 32 + r;
  const int key_ld = key_row < T ? key_row : T - 1;

  bf16x8 kf[4], vf[4];   // B operands: K[key = r][dh = 16 s + 8 hh + j], V likewise
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = *reinterpret_cast<const bf16x8*>(kp + (long)key_ld * rs + 16 * s + 8 * hh);
    vf[s] = *reinterpret_cast<const bf16x8*>(vp + (long)key_ld * rs + 16 * s + 8 * hh);
  }

  int st_row[2], st_chunk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    st_row[i] = c >> 3;
    st_chunk[i] = c & 7;
  }
  u32x4 rq[2], ro[2];
  float rl = 0.f;
  auto gload = [&](int qt) {
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int q = qt * KB + st_row[i];
        if (q >= T) q = T - 1;
        rq[i] = *reinterpret_cast<const u32x4*>(qp + (long)q * rs + st_chunk[i] * 8);
        ro[i] = *reinterpret_cast<const u32x4*>(dop + (long)q * d + st_chunk[i] * 8);
      }
    }
    if (tid < 128) {
      const int q = qt * KB + (tid & 63);
      // queries past T get LSE = +inf (P = 0) and D = 0; L2Q: both negated (accumulator start values)
      rl = (tid < 64) ? (q < T ? lsep[q] * kLog2e : INFINITY) : (q < T ? Dp[q] : 0.f);
      if constexpr (L2Q) rl = -rl;
    }
  };
  // DMA: piece j of wave w = tile rows 8 (2 w + j) .. + 7, 16 bytes per lane; the dual-use swizzle on the per-lane SOURCE chunk;
  // queries past T - 1 read row T - 1 (their P is 0: lse = +inf)
  auto dma = [&](int qt, int buf) {
    if constexpr (DMA) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
        int q = qt * KB + row;
        if (q >= T) q = T - 1;
        const int ch = (pos ^ dual_x(row)) << 3;
        __builtin_amdgcn_global_load_lds((g_ptr)(qp + (long)q * rs + ch), (lds_ptr)(Qr(buf) + (2 * wave + j) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(dop + (long)q * d + ch), (lds_ptr)(Or(buf) + (2 * wave + j) * 1024), 16, 0, 0);
      }
    }
  };
  auto lstore = [&](int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<u32x4*>(Qr(buf) + row_off(st_row[i], st_chunk[i])) = rq[i];
        *reinterpret_cast<u32x4*>(Qt(buf) + tr_off(st_row[i], st_chunk[i] * 16)) = rq[i];
        *reinterpret_cast<u32x4*>(Or(buf) + row_off(st_row[i], st_chunk[i])) = ro[i];
        *reinterpret_cast<u32x4*>(Ot(buf) + tr_off(st_row[i], st_chunk[i] * 16)) = ro[i];
      }
    }
    if (tid < 128) Ls(buf)[tid] = rl;   // [0,64): lse * log2e, [64,128): D
  };

  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) { dkt[n][j] = 0.f; dvt[n][j] = 0.f; }

  // only the query tiles whose dO is not all zero contribute (dV += P^T dO, dS = P (dO V^T - D) with D = 0 there)
  const int n_qt = (T + KB - 1) / KB;
  const unsigned int* live = nz + (long)bh * n_qt;
  auto next_live = [&](int q) {
    while (q < n_qt && !live[q]) ++q;
    return q;
  };
  int qt = next_live(0), buf = 0;
  if (qt < n_qt) {
    gload(qt);
    dma(qt, 0);
    lstore(0);
  }
  __syncthreads();   // (DMA: hipcc drains the LDS-DMA in front of the barrier)
  while (qt < n_qt) {
    const int qn = next_live(qt + 1);
    if (qn < n_qt) { gload(qn); dma(qn, buf ^ 1); }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // S[q][key], dP[q][key]: rows q = 32 g + (reg&3) + 8 (reg>>2) + 4 hh in registers, key on the lane
      f32x16 sm, dpm, ds;
      if constexpr (L2Q) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // -lse, -D of the rows in registers 4 c .. 4 c + 3: the chains' start values
          const float4 l4 = *reinterpret_cast<const float4*>(Ls(buf) + 32 * g + 8 * c + 4 * hh);
          const float4 d4 = *reinterpret_cast<const float4*>(Ls(buf) + 64 + 32 * g + 8 * c + 4 * hh);
          sm[4 * c] = l4.x; sm[4 * c + 1] = l4.y; sm[4 * c + 2] = l4.z; sm[4 * c + 3] = l4.w;
          dpm[4 * c] = d4.x; dpm[4 * c + 1] = d4.y; dpm[4 * c + 2] = d4.z; dpm[4 * c + 3] = d4.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sm[j] = 0.f; dpm[j] = 0.f; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ro_ = DMA ? dual_off(32 * g + r, 2 * s + hh) : row_off(32 * g + r, 2 * s + hh);
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qr(buf) + ro_);
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(Or(buf) + ro_);
        sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], sm, 0, 0, 0);
        dpm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[s], dpm, 0, 0, 0);
      }
      if constexpr (L2Q) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          sm[j] = __builtin_amdgcn_exp2f(sm[j]);
          ds[j] = sm[j] * dpm[j];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 l4 = *reinterpret_cast<const float4*>(Ls(buf) + 32 * g + 8 * c + 4 * hh);
          const float4 d4 = *reinterpret_cast<const float4*>(Ls(buf) + 64 + 32 * g + 8 * c + 4 * hh);
          const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sm[4 * c + e], ssc, -ll[e]));
            sm[4 * c + e] = p;
            ds[4 * c + e] = p * (dpm[4 * c + e] - dd[e]);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = cvt8(sm, 8 * s), dsf = cvt8(ds, 8 * s);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          dvt[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(DMA ? tr_frag_dual(Ot(buf), 32 * g + 16 * s, n, lane) : tr_frag(Ot(buf), 32 * g + 16 * s, n, lane), pf, dvt[n], 0, 0, 0);
          dkt[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(DMA ? tr_frag_dual(Qt(buf), 32 * g + 16 * s, n, lane) : tr_frag(Qt(buf), 32 * g + 16 * s, n, lane), dsf, dkt[n], 0, 0, 0);
        }
      }
    }
    if (qn < n_qt) lstore(buf ^ 1);
    __syncthreads();
    qt = qn;
    buf ^= 1;
  }
  if (key_row < T) {
    unsigned short* krow = dqkv + ((long)b * T + key_row) * rs + d + h * DH;
    unsigned short* vrow = krow + d;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x2 ok = {pack2bf(dkt[n][4 * c] * gsc, dkt[n][4 * c + 1] * gsc), pack2bf(dkt[n][4 * c + 2] * gsc, dkt[n][4 * c + 3] * gsc)};
        u32x2 ov = {pack2bf(dvt[n][4 * c], dvt[n][4 * c + 1]), pack2bf(dvt[n][4 * c + 2], dvt[n][4 * c + 3])};
        *reinterpret_cast<u32x2*>(krow + 32 * n + 8 * c + 4 * hh) = ok;
        *reinterpret_cast<u32x2*>(vrow + 32 * n + 8 * c + 4 * hh) = ov;
      }
  }
}

// qkv, ctx, dctx bf16; lse [B,H,T] from the forward; D scratch of B H (T + ceil(T / 64)) floats (row dots, then
// the live-tile flags); dqkv [B,T,3d] out
// q_log2: the q section of qkv is in log2 units (projected with log2(e) / 8, what the forward kernels of attention.hip
// take): S = q_l2 k / log2(e), so P = exp2(q_l2 k - lse log2(e)) needs no multiply and dS is scaled by 1 / log2(e) once --
// dq then is the gradient with respect to the STORED q (dS K / log2(e)) and dk = dS^T q_l2 / log2(e).
int launch_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* D,
                              void* dqkv, int B, int T, int H, hipStream_t s, bool q_log2) {
  const float ssc = q_log2 ? 1.0f : kLog2e, gsc = q_log2 ? 1.0f / kLog2e : 1.0f;
  GWW_REQUIRE(qkv && ctx && dctx && lse && D && dqkv, "attention_bwd: NULL operand");
  GWW_REQUIRE(B >= 0 && T > 0 && H > 0, "attention_bwd: bad shape");
  if (B == 0) return GWW_OK;
  const long rows = (long)B * T;
  unsigned int* nz = reinterpret_cast<unsigned int*>(D + rows * H);   // [B, H, ceil(T / 64)] live-tile flags
  GWW_HIP(hipMemsetAsync(nz, 0, sizeof(unsigned int) * B * H * cdiv(T, KB), s));
  hipLaunchKernelGGL(k_attn_rowdot, dim3((unsigned)cdiv(rows * H * 8, 256)), dim3(256), 0, s,
                     (const unsigned short*)ctx, (const unsigned short*)dctx, D, nz, T, H, rows);
  GWW_LAUNCH_CHECK();
  const int tiles = (T + TB - 1) / TB;
  const long blocks = (long)tiles * B * H;
  GWW_REQUIRE(blocks < 2147483647L, "attention_bwd: grid too large");
#define GWW_ATTBWD_LAUNCH(L)                                                                                       \
  hipLaunchKernelGGL(k_attn_bwd_dq<L>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)qkv,          \
                     (const unsigned short*)dctx, lse, D, (unsigned short*)dqkv, T, H, tiles, nz, ssc, gsc);         \
  GWW_LAUNCH_CHECK();                                                                                               \
  hipLaunchKernelGGL(k_attn_bwd_dkv<L>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)qkv,         \
                     (const unsigned short*)dctx, lse, D, (unsigned short*)dqkv, T, H, tiles, nz, ssc, gsc);         \
  GWW_LAUNCH_CHECK();
  if (q_log2) { GWW_ATTBWD_LAUNCH(true) } else { GWW_ATTBWD_LAUNCH(false) }
#undef GWW_ATTBWD_LAUNCH
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_attention_bwd_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                                      float* d_scratch, void* dqkv, int B, int T, int n_heads, void* stream) {
  return launch_attention_bwd_bf16(qkv, ctx, dctx, lse, d_scratch, dqkv, B, T, n_heads, (hipStream_t)stream, false);
}
extern "C" int gww_attention_bwd_log2q_bf16(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                                            float* d_scratch, void* dqkv, int B, int T, int n_heads, void* stream) {
  return launch_attention_bwd_bf16(qkv, ctx, dctx, lse, d_scratch, dqkv, B, T, n_heads, (hipStream_t)stream, true);
}
