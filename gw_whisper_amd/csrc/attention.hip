// Multi-head self-attention core (K7 of SURVEY.md section 2.1), flash style:
//   ctx[b, t, h*64 : h*64+64] = softmax(q k^T) v      non-causal, no mask, head_dim 64
// HF:modeling_whisper.py:215-238 with scaling = 1.0 (:348) -- q arrives pre-scaled
// (head_dim^-0.5 is folded into the packed q_proj weights, an exact power of two).
//
// Layout: qkv [B, T, 3 d] (q | k | v), ctx [B, T, d].
//
// bf16 kernel, per workgroup 128 query rows of one (b, h): 4 waves x 32 rows.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): the query row
//     sits on the LANE, the 32 keys of a tile in the lane's accumulator registers, so
//     the online-softmax max / sum are in-lane plus ONE exchange with lane ^ 32.
//   * O^T = V^T P^T: the fp32 S^T accumulator, converted pairwise to bf16, IS the B
//     operand of the PV product (no LDS round trip for P); the V^T A-operand comes
//     from the row-major V tile in LDS through ds_read_b64_tr_b16.  O^T keeps the
//     query on the lane too, so the rescale factor and the final 1/l are per lane.
//   * K tile [64 keys][64] bf16 in LDS with the 16-B chunk XOR-swizzle (conflict-free
//     ds_read_b128), V tile with bit-6 XOR (conflict-free transposed reads).
//   * K/V tiles are register-staged and double-buffered: the global loads of tile
//     j+1 fly during the MFMAs of tile j; one barrier per tile.
// fp32 kernel: same dataflow on v_mfma_f32_32x32x2_f32 (exact fp32), 32-key tiles.
#include "common.h"

#include <type_traits>

namespace gww {

constexpr int DH = 64;
constexpr int QB = 128;   // query rows per workgroup
constexpr int KB = 64;    // keys per tile (bf16 kernel)
constexpr float kLog2e = 1.44269504088896340736f;

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int colbyte) { return row * 128 + (colbyte ^ (((row >> 1) & 1) << 6)); }

__device__ __forceinline__ bf16x8 cvt8(const f32x16& a, int base) {
  const u32x4 r = {pack2bf(a[base], a[base + 1]), pack2bf(a[base + 2], a[base + 3]),
                   pack2bf(a[base + 4], a[base + 5]), pack2bf(a[base + 6], a[base + 7])};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(256, 2) void k_attention_bf16(const unsigned short* __restrict__ qkv,
                                                           unsigned short* __restrict__ ctx,
                                                           float* __restrict__ lse, int T, int H,
                                                           int q_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * KB * DH * 2];   // 32 KB
  constexpr int TILE_BYTES = KB * DH * 2;
  auto Ks = [&](int buf) -> unsigned char* { return lds + buf * TILE_BYTES; };
  auto Vs = [&](int buf) -> unsigned char* { return lds + (2 + buf) * TILE_BYTES; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware work order: workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Give every
  // XCD a contiguous range of (b, h, q-tile) so that the q-tiles of one head -- which all stream the same
  // K / V (384 KB) -- run on ONE XCD back to back and K / V come from HBM once instead of once per XCD.
  const unsigned nblk = gridDim.x, per = nblk >> 3;
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
  const int qt = wid % q_tiles;
  const int bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * DH;
  const long row_stride = 3L * d;
  const unsigned short* base = qkv + (long)b * T * row_stride;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;

  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * QB + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;

  // Q fragments: B operand of K Q^T -> Q[q = r][dh = 16 s + 8 hh + j]
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * row_stride + 16 * s + 8 * hh);

  // staging: 2 chunks (16 B) of K and 2 of V per thread per tile
  int st_row[2], st_chunk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    st_row[i] = c >> 3;
    st_chunk[i] = c & 7;
  }
  u32x4 rk[2], rv[2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kt * KB + st_row[i];
      if (key >= T) key = T - 1;
      rk[i] = *reinterpret_cast<const u32x4*>(kp + (long)key * row_stride + st_chunk[i] * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(vp + (long)key * row_stride + st_chunk[i] * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(Ks(buf) + k_off(st_row[i], st_chunk[i])) = rk[i];
      *reinterpret_cast<u32x4*>(Vs(buf) + v_off(st_row[i], st_chunk[i] * 16)) = rv[i];
    }
  };

  f32x16 ot[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) ot[n][j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int n_kt = (T + KB - 1) / KB;
  gload(0);
  lstore(0);
  __syncthreads();

  // per-lane constants of the transposed V read: lane i of a 16-lane group supplies
  // row (i >> 2), columns 4 (i & 3) .. +3 of a 4 x 16 block
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_colbyte = (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;

  // one key tile; MASKED (compile time) only for the ragged last tile, so the full tiles carry no per-score
  // select.  The running maximum is refreshed only when some row's tile maximum exceeds it by more than
  // kDefer (any reference value is exact algebra; p <= e^kDefer keeps bf16 / fp32 far from overflow): the
  // 16 rescale multiplies of O and the extra exp leave most tiles.
  constexpr float kDefer = 8.0f;
  auto tile = [&](int kt, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const int buf = kt & 1;
    if (kt + 1 < n_kt) gload(kt + 1);

    // ---- S^T = K Q^T : st[g][reg] = score(key = 32 g + (reg&3) + 8 (reg>>2) + 4 hh, q = r)
    f32x16 st[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int j = 0; j < 16; ++j) st[g][j] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks(buf) + k_off(32 * g + r, 2 * s + hh));
        st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[g], 0, 0, 0);
      }
    }
    if constexpr (MASKED) {   // keys >= T
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (key >= T) st[g][j] = -INFINITY;
        }
    }
    // ---- online softmax, query row on the lane
    float tmax = st[0][0];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, st[g][j]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__builtin_amdgcn_ballot_w64(tmax > m_run + kDefer) != 0) {   // wave-uniform
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) ot[n][j] *= alpha;
    }
    const float ms = m_run * kLog2e;
    float psum = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[g][j], kLog2e, -ms));
        st[g][j] = p;
        psum += p;
      }
    l_run += psum;

    // ---- O^T += V^T P^T : B operand = bf16(st) registers 8 s .. 8 s + 7 (k-step s);
    //      A operand element j <-> key 32 g + 16 s + 8 (j>>2) + 4 hh + (j&3), dh = 32 n + r
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = cvt8(st[g], 8 * s);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int key0 = 32 * g + 16 * s + 4 * hh + tr_q;
          const int cb = 64 * n + tr_colbyte;
          typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (lds_bf16x4_ptr)(Vs(buf) + v_off(key0, cb)));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (lds_bf16x4_ptr)(Vs(buf) + v_off(key0 + 8, cb)));
          bf16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[n], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < n_kt) lstore(buf ^ 1);
    __syncthreads();
  };
  for (int kt = 0; kt + 1 < n_kt; ++kt) tile(kt, std::false_type{});
  if ((T % KB) != 0) tile(n_kt - 1, std::true_type{});
  else tile(n_kt - 1, std::false_type{});

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  // log-sum-exp of the row (training: the backward recomputes P = exp(S - LSE))
  if (lse && q_row < T && hh == 0) lse[((long)b * H + h) * T + q_row] = m_run + __logf(l_tot);
  if (q_row < T) {
    unsigned short* orow = ctx + ((long)b * T + q_row) * d + h * DH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dh = 32 * n + 8 * c + 4 * hh;
        u32x2 o = {pack2bf(ot[n][4 * c] * inv, ot[n][4 * c + 1] * inv),
                   pack2bf(ot[n][4 * c + 2] * inv, ot[n][4 * c + 3] * inv)};
        *reinterpret_cast<u32x2*>(orow + dh) = o;
      }
  }
}

int launch_attention_bf16(const void* qkv, void* ctx, int B, int T, int H, hipStream_t s, float* lse) {
  GWW_REQUIRE(qkv && ctx, "attention_bf16: NULL operand");
  GWW_REQUIRE(B >= 0 && T > 0 && H > 0, "attention_bf16: bad shape B=%d T=%d H=%d", B, T, H);
  GWW_REQUIRE((((uintptr_t)qkv) & 15) == 0 && (((uintptr_t)ctx) & 15) == 0, "attention_bf16: 16-byte alignment");
  if (B == 0) return GWW_OK;
  const int q_tiles = (T + QB - 1) / QB;
  const long blocks = (long)q_tiles * B * H;
  GWW_REQUIRE(blocks < 2147483647L, "attention_bf16: grid too large");
  hipLaunchKernelGGL(k_attention_bf16, dim3((unsigned)blocks), dim3(256), 0, s,
                     (const unsigned short*)qkv, (unsigned short*)ctx, lse, T, H, q_tiles);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ------------------------------------------------------------------ fp32 twin
// v_mfma_f32_32x32x2_f32: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31].
constexpr int FKB = 32;          // keys per tile
constexpr int FLDS = DH + 1;     // padded fp32 row

__global__ __launch_bounds__(256) void k_attention_f32(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                       int T, int H, int q_tiles) {
  __shared__ float Ks[FKB][FLDS];
  __shared__ float Vs[FKB][FLDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x % q_tiles;
  const int bh = blockIdx.x / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * DH;
  const long row_stride = 3L * d;
  const float* base = qkv + (long)b * T * row_stride;
  const float* qp = base + h * DH;
  const float* kp = base + d + h * DH;
  const float* vp = base + 2 * d + h * DH;
  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * QB + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;

  float qf[32];   // Q[q = r][dh = 2 s + hh]
#pragma unroll
  for (int s = 0; s < 32; ++s) qf[s] = qp[(long)q_ld * row_stride + 2 * s + hh];

  f32x16 ot[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) ot[n][j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int n_kt = (T + FKB - 1) / FKB;
  for (int kt = 0; kt < n_kt; ++kt) {
    // stage 32 keys x 64 dh of K and V (fp32): 2048 floats each, 8 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, row = c >> 4, col = (c & 15) * 4;
      int key = kt * FKB + row;
      if (key >= T) key = T - 1;
      const float4 kv = *reinterpret_cast<const float4*>(kp + (long)key * row_stride + col);
      const float4 vv = *reinterpret_cast<const float4*>(vp + (long)key * row_stride + col);
      Ks[row][col] = kv.x; Ks[row][col + 1] = kv.y; Ks[row][col + 2] = kv.z; Ks[row][col + 3] = kv.w;
      Vs[row][col] = vv.x; Vs[row][col + 1] = vv.y; Vs[row][col + 2] = vv.z; Vs[row][col + 3] = vv.w;
    }
    __syncthreads();
    f32x16 st;
#pragma unroll
    for (int j = 0; j < 16; ++j) st[j] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r][2 * s + hh], qf[s], st, 0, 0, 0);
    if (kt == n_kt - 1 && (T % FKB) != 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int key = kt * FKB + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (key >= T) st[j] = -INFINITY;
      }
    }
    float tmax = st[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, st[j]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      st[j] = expf(st[j] - m_new);
      psum += st[j];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 16; ++j) ot[n][j] *= alpha;
    // O^T[dh][q] += V^T[dh][key] P^T[key][q]; MFMA step rho pairs k=0 <-> key_a(rho), k=1 <-> key_a(rho)+4
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) {
      const int key = (rho & 3) + 8 * (rho >> 2) + 4 * hh;
#pragma unroll
      for (int n = 0; n < 2; ++n)
        ot[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key][32 * n + r], st[rho], ot[n], 0, 0, 0);
    }
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_row < T) {
    float* orow = ctx + ((long)b * T + q_row) * d + h * DH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dh = 32 * n + 8 * c + 4 * hh;
        *reinterpret_cast<float4*>(orow + dh) = make_float4(ot[n][4 * c] * inv, ot[n][4 * c + 1] * inv,
                                                            ot[n][4 * c + 2] * inv, ot[n][4 * c + 3] * inv);
      }
  }
}

int launch_attention_f32(const float* qkv, float* ctx, int B, int T, int H, hipStream_t s) {
  GWW_REQUIRE(qkv && ctx, "attention_f32: NULL operand");
  GWW_REQUIRE(B >= 0 && T > 0 && H > 0, "attention_f32: bad shape B=%d T=%d H=%d", B, T, H);
  GWW_REQUIRE((((uintptr_t)qkv) & 15) == 0 && (((uintptr_t)ctx) & 15) == 0, "attention_f32: 16-byte alignment");
  if (B == 0) return GWW_OK;
  const int q_tiles = (T + QB - 1) / QB;
  const long blocks = (long)q_tiles * B * H;
  GWW_REQUIRE(blocks < 2147483647L, "attention_f32: grid too large");
  hipLaunchKernelGGL(k_attention_f32, dim3((unsigned)blocks), dim3(256), 0, s, qkv, ctx, T, H, q_tiles);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_attention_bf16(const void* qkv, void* ctx, int B, int T, int n_heads, void* stream) {
  return launch_attention_bf16(qkv, ctx, B, T, n_heads, (hipStream_t)stream);
}
extern "C" int gww_attention_lse_bf16(const void* qkv, void* ctx, float* lse, int B, int T, int n_heads, void* stream) {
  GWW_REQUIRE(lse != nullptr, "gww_attention_lse_bf16: lse is NULL");
  return launch_attention_bf16(qkv, ctx, B, T, n_heads, (hipStream_t)stream, lse);
}
extern "C" int gww_attention_f32(const float* qkv, float* ctx, int B, int T, int n_heads, void* stream) {
  return launch_attention_f32(qkv, ctx, B, T, n_heads, (hipStream_t)stream);
}
