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

#include <stdlib.h>
#include <type_traits>

#ifndef GWW_ATT_EPI16
#define GWW_ATT_EPI16 1   // context rows stored in 16-byte pieces (lane pairs exchange halves) instead of 8-byte ones
#endif
namespace gww {

#ifdef GWW_STAMP
__device__ unsigned long long g_stamp_att[8];
#define ASTAMP_DECL unsigned long long _t0 = __builtin_amdgcn_s_memtime(); unsigned long long _acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define ASTAMP(i)                                                  \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long _t1 = __builtin_amdgcn_s_memtime();   \
    _acc[i] += _t1 - _t0;                                          \
    _t0 = _t1;                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#define ASTAMP_FLUSH                                                                 \
  if (lane == 0) {                                                                   \
    for (int _q = 0; _q < 7; ++_q) atomicAdd(&g_stamp_att[_q], _acc[_q]);            \
    atomicAdd(&g_stamp_att[7], 1ull);                                                \
  }
#else
#define ASTAMP_DECL
#define ASTAMP(i)
#define ASTAMP_FLUSH
#endif

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

// NW waves x 32 query rows per workgroup.  NW = 8 (256 rows, one workgroup per CU) streams every K / V tile
// once per 256 queries: at 128 rows per workgroup the kernel needed 32 B/clk/CU of K / V from L2 -- the whole L2
// bandwidth of an XCD -- and spent 60 % of its cycles waiting for the next tile (tools/stamp_att.py).
#ifndef GWW_ATT_MINBLK
#define GWW_ATT_MINBLK 2   // tuning aid: 2 lets the allocator use 168 VGPRs (3 waves / SIMD); 4 forces 128 (spills)
#endif
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? GWW_ATT_MINBLK : 1) void k_attention_bf16(const unsigned short* __restrict__ qkv,
                                                           unsigned short* __restrict__ ctx,
                                                           float* __restrict__ lse, int T, int H,
                                                           int q_tiles, int qt0) {
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
  const int qt = qt0 + wid % q_tiles;   // q_tiles = query tiles in THIS launch, starting at tile qt0
  const int bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * DH;
  const long row_stride = 3L * d;
  const unsigned short* base = qkv + (long)b * T * row_stride;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;

  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * (NW * 32) + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;

  // Q fragments: B operand of K Q^T -> Q[q = r][dh = 16 s + 8 hh + j]
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * row_stride + 16 * s + 8 * hh);

  // staging: 512 chunks (16 B) of K and of V per tile, NCH per thread
  constexpr int NCH = 512 / (NW * 64);
  int st_row[NCH], st_chunk[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + NW * 64 * i;
    st_row[i] = c >> 3;
    st_chunk[i] = c & 7;
  }
  u32x4 rk[NCH], rv[NCH];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int key = kt * KB + st_row[i];
      if (key >= T) key = T - 1;
      rk[i] = *reinterpret_cast<const u32x4*>(kp + (long)key * row_stride + st_chunk[i] * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(vp + (long)key * row_stride + st_chunk[i] * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      *reinterpret_cast<u32x4*>(Ks(buf) + k_off(st_row[i], st_chunk[i])) = rk[i];
      *reinterpret_cast<u32x4*>(Vs(buf) + v_off(st_row[i], st_chunk[i] * 16)) = rv[i];
    }
  };

  ASTAMP_DECL
  f32x16 ot[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 16; ++j) ot[n][j] = 0.f;
  float m_run = -INFINITY;
  // The softmax denominator rides on the matrix pipe: one extra MFMA per k-step multiplies P^T by an all-ones A
  // operand, so every register of `lt` holds sum_k bf16(p[k]) of the lane's query (both key halves included) --
  // 32 v_add_f32 per tile leave the VALU, which is the busier pipe here, for 4 MFMAs on the idler one; the
  // normalisation then uses exactly the rounded probabilities the numerator uses.
  f32x16 lt;
#pragma unroll
  for (int j = 0; j < 16; ++j) lt[j] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

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
    ASTAMP(1);

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
    ASTAMP(2);
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
      m_run = m_new;
#pragma unroll
      for (int j = 0; j < 16; ++j) lt[j] *= alpha;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) ot[n][j] *= alpha;
    }
    const float ms = m_run * kLog2e;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) st[g][j] = __builtin_amdgcn_exp2f(fmaf(st[g][j], kLog2e, -ms));

    ASTAMP(3);
    // ---- O^T += V^T P^T : B operand = bf16(st) registers 8 s .. 8 s + 7 (k-step s);
    //      A operand element j <-> key 32 g + 16 s + 8 (j>>2) + 4 hh + (j&3), dh = 32 n + r
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = cvt8(st[g], 8 * s);
        lt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lt, 0, 0, 0);
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
    ASTAMP(4);
    if (kt + 1 < n_kt) lstore(buf ^ 1);
    ASTAMP(5);
    __syncthreads();
    ASTAMP(6);
  };
  for (int kt = 0; kt + 1 < n_kt; ++kt) tile(kt, std::false_type{});
  if ((T % KB) != 0) tile(n_kt - 1, std::true_type{});
  else tile(n_kt - 1, std::false_type{});

  const float l_tot = lt[0];
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
  ASTAMP(0);
  ASTAMP_FLUSH
}

#ifdef GWW_LAB   // laboratory variant: compiled into libgww_lab.so only (make LAB=1)
// ---------------------------------------------------------------------------------------------------------
// The kernel the inference path launches (q in log2 units).  Same dataflow and occupancy as k_attention_bf16 above
// (one tile = S, softmax, P V back to back; three waves per SIMD supply the overlap), with the instruction stream
// of a 64-key tile cut from ~225 to ~170 -- the kernel is bound by instruction issue, not by memory or LDS
// (DESIGN.md section 4):
//   * scores arrive in log2 units (log2(e) / 8 is folded into the packed q panel: one rounding of the fp32
//     product instead of a multiply per score) and the running reference -m enters THROUGH THE MATRIX PIPE: the
//     accumulator chain of each 32-key half starts with one extra MFMA  ones[key][k] x mref[k][q], where mref
//     holds -m of the lane's query as a bf16 hi + lo pair in k = 0, 1 (16 significant bits; whatever the pair
//     sums to IS the reference -- any reference is exact algebra as long as numerator and denominator share it).
//     p = v_exp_f32(s) then needs no subtract / multiply-add per score (31 VALU instructions per tile) and no
//     persistent 16-register C operand (what cost the pipelined kernel below its third wave);
//   * the reference moves only when some row's tile maximum exceeds it by more than 2^kDeferL2 (deferred rescale,
//     rare path: O, l and the tile's scores are re-based there); the FIRST tile is scored against 0 and then always
//     re-based to its own row maximum, so there is no -inf reference and no underflow for uniformly negative rows;
//   * K / V tile addresses are a scalar tile base + a per-lane 32-bit offset that never changes (a second offset
//     set clamps the rows of the ragged last tile), and every LDS address is a per-lane base + an immediate: the
//     tile loop is unrolled by two so the buffer parity is a compile-time constant.
// VAR (tuning aid, GWW_ATT_VAR): bit 0 = K / V global loads issued after the exponentials instead of at the top of the
// tile; bit 1 = softmax denominator by 32 v_add_f32 per tile instead of the ones-MFMA (frees its 16 accumulators)
template <int NW, int VAR>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 1) void k_attention_l2_bf16(const unsigned short* __restrict__ qkv,
                                                                              unsigned short* __restrict__ ctx,
                                                                              float* __restrict__ lse, int T, int H,
                                                                              int q_tiles, int qt0) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * KB * DH * 2];   // K0 | K1 | V0 | V1, 8 KB each
  constexpr int TILE_BYTES = KB * DH * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned nblk = gridDim.x, per = nblk >> 3;
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;   // XCD-aware order
  const int qt = qt0 + wid % q_tiles;
  const int bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * DH;
  const long row_stride = 3L * d;
  const unsigned short* base = qkv + (long)b * T * row_stride;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;
  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * (NW * 32) + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;
  bf16x8 qf[4];   // B operand of K Q^T: Q[q = r][dh = 16 s + 8 hh + j], log2 units
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * row_stride + 16 * s + 8 * hh);

  // ---- staging: 512 16-byte chunks of K and of V per tile, NCH per thread
  constexpr int NCH = 512 / (NW * 64);
  const int n_kt = (T + KB - 1) / KB;
  unsigned off_full[NCH];   // BYTE offset of the thread's chunk from the tile base: base (scalar) + zext(offset) addressing
  int lk_off[NCH], lv_off[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + NW * 64 * i, row = c >> 3, chunk = c & 7;
    off_full[i] = (unsigned)(row * (int)row_stride * 2 + chunk * 16);
    lk_off[i] = k_off(row, chunk);
    lv_off[i] = 2 * TILE_BYTES + v_off(row, chunk * 16);
  }
  u32x4 rk[NCH], rv[NCH];
  auto gload = [&](int kt) {
    const char* kb = reinterpret_cast<const char*>(kp + (long)kt * KB * row_stride);   // wave-uniform: scalar registers
    const char* vb = reinterpret_cast<const char*>(vp + (long)kt * KB * row_stride);
    if (kt != n_kt - 1 || (T % KB) == 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        rk[i] = *reinterpret_cast<const u32x4*>(kb + off_full[i]);
        rv[i] = *reinterpret_cast<const u32x4*>(vb + off_full[i]);
      }
    } else {   // ragged last tile (once per workgroup): rows past T - 1 are clamped to it (never read past the tensor)
      const int last_row = T - 1 - kt * KB;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = tid + NW * 64 * i;
        int row = c >> 3;
        row = row < last_row ? row : last_row;
        const unsigned o = (unsigned)(row * (int)row_stride * 2 + (c & 7) * 16);
        rk[i] = *reinterpret_cast<const u32x4*>(kb + o);
        rv[i] = *reinterpret_cast<const u32x4*>(vb + o);
      }
    }
  };
  auto lstore = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      *reinterpret_cast<u32x4*>(lds + BUF * TILE_BYTES + lk_off[i]) = rk[i];
      *reinterpret_cast<u32x4*>(lds + BUF * TILE_BYTES + lv_off[i]) = rv[i];
    }
  };

  // ---- per-lane LDS bases of the fragment reads (everything else is an immediate)
  // K: row 32 g + r, chunk (2 s + hh) ^ ((r >> 1) & 7)        V^T: row 32 g + 16 s + 4 hh + tr_q (+ 8), column bytes
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_colbyte = (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;
  const unsigned char* kbase[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) kbase[s] = lds + k_off(r, 2 * s + hh);
  const unsigned char* vbase[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) vbase[n] = lds + 2 * TILE_BYTES + v_off(4 * hh + tr_q, 64 * n + tr_colbyte);

  f32x16 ot[2], lt;
#pragma unroll
  for (int j = 0; j < 16; ++j) { ot[0][j] = 0.f; ot[1][j] = 0.f; lt[j] = 0.f; }
  bf16x8 ones, mref;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ones[j] = (__bf16)1.0f; mref[j] = (__bf16)0.0f; }
  float l_run = 0.f;   // VAR bit 1: this lane's half of the denominator
  float m_run = 0.f;   // the reference the scores are taken against, log2 units (== -(hi + lo) of mref)
  constexpr float kDeferL2 = 8.0f * kLog2e;

  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
  auto tile = [&](int kt, auto buf_c, auto first_c, auto masked_c) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool FIRST = decltype(first_c)::value, MASKED = decltype(masked_c)::value;
    constexpr bool LATE = (VAR & 1) != 0, VSUM = (VAR & 2) != 0;
    if constexpr (!LATE) { if (kt + 1 < n_kt) gload(kt + 1); }
    // ---- S^T - m = [ones | K] [mref | Q]^T : st[g][reg] <-> key 32 g + (reg & 3) + 8 (reg >> 2) + 4 hh, q = r
    f32x16 st[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x16 z;
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = 0.f;
      if constexpr (FIRST) st[g] = z;
      else st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, mref, z, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase[s] + BUF * TILE_BYTES + g * (32 * 128));
        st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[g], 0, 0, 0);
      }
    }
    if constexpr (MASKED) {   // keys >= T of the ragged last tile
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (key >= T) st[g][j] = -INFINITY;
        }
    }
    float tmax = st[0][0];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, st[g][j]);
    {   // the other 32 keys of the row sit in lane ^ 32: one v_permlane32_swap instead of an LDS round trip (ds_bpermute)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
      tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (FIRST || __builtin_amdgcn_ballot_w64(tmax > kDeferL2) != 0) {   // wave-uniform, rare after the first tile
      // move the reference of every row that needs it: the new one is what the bf16 pair can represent
      const float want = m_run + (FIRST ? tmax : fmaxf(tmax, 0.f));
      const __bf16 hi = (__bf16)(-want);
      const __bf16 lo = (__bf16)(-want - (float)hi);
      const float m_new = -((float)hi + (float)lo);
      const float dm = m_new - m_run;
      const float alpha = __builtin_amdgcn_exp2f(-dm);
      m_run = m_new;
      if constexpr (VSUM && !FIRST) l_run *= alpha;
      mref[0] = hh == 0 ? hi : (__bf16)0.0f;
      mref[1] = hh == 0 ? lo : (__bf16)0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if constexpr (!FIRST) {   // first tile: O = l = 0, and 2^-dm may overflow for uniformly negative rows
          if constexpr (!VSUM) lt[j] *= alpha;
          ot[0][j] *= alpha;
          ot[1][j] *= alpha;
        }
        st[0][j] -= dm;
        st[1][j] -= dm;
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) st[g][j] = __builtin_amdgcn_exp2f(st[g][j]);
    // the next tile's K / V leave for the registers here (issue late, write after the P V product: the staging
    // registers are not live across the score phase, whose 8 K fragments need the room)
    if constexpr (LATE) { if (kt + 1 < n_kt) gload(kt + 1); }
    if constexpr (VSUM) {
      float ps = 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) ps += st[g][j];
      l_run += ps;
    }
    // ---- O^T += V^T P^T, l += 1^T P^T : B operand = bf16(st) registers 8 s .. 8 s + 7 of k-step s
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = cvt8(st[g], 8 * s);
        if constexpr (!VSUM) lt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lt, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const unsigned char* vb = vbase[n] + BUF * TILE_BYTES + (32 * g + 16 * s) * 128;
          const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)vb);
          const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb + 8 * 128));
          bf16x8 vf;
          vf[0] = lo4[0]; vf[1] = lo4[1]; vf[2] = lo4[2]; vf[3] = lo4[3];
          vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
          ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[n], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < n_kt) lstore(std::integral_constant<int, BUF ^ 1>{});
    __syncthreads();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;

  gload(0);
  lstore(P0{});
  __syncthreads();
  const bool ragged = (T % KB) != 0;
  if (n_kt == 1) {
    if (ragged) tile(0, P0{}, Yes{}, Yes{});
    else tile(0, P0{}, Yes{}, No{});
  } else {
    tile(0, P0{}, Yes{}, No{});
    int kt = 1;
    for (; kt + 2 <= n_kt - 1; kt += 2) {
      tile(kt, P1{}, No{}, No{});
      tile(kt + 1, P0{}, No{}, No{});
    }
    if (kt == n_kt - 2) {   // two tiles left (kt odd)
      tile(kt, P1{}, No{}, No{});
      if (ragged) tile(kt + 1, P0{}, No{}, Yes{});
      else tile(kt + 1, P0{}, No{}, No{});
    } else {                // one tile left
      if (ragged) tile(kt, P1{}, No{}, Yes{});
      else tile(kt, P1{}, No{}, No{});
    }
  }

  const float l_tot = (VAR & 2) ? l_run + __shfl_xor(l_run, 32, 64) : lt[0];
  const float inv = 1.0f / l_tot;
  if (lse && q_row < T && hh == 0)
    lse[((long)b * H + h) * T + q_row] = (m_run + __log2f(l_tot)) * 0.69314718055994530942f;   // natural log
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

#endif  // GWW_LAB
// ---------------------------------------------------------------------------------------------------------
// LDS-DMA form of k_attention_l2_bf16 (GWW_ATT_VAR = 4 / 5): K / V tiles go global -> LDS directly
// (global_load_lds_dwordx4, the XOR swizzles applied on the per-lane SOURCE address, the LDS image stays lane-linear),
// so there are no staging registers (16 VGPRs) and no ds_write_b128 (4 per wave and tile, 13 issue cycles each); the
// softmax denominator is summed on the VALU (no ones-MFMA accumulator: another 16 VGPRs).  The kernel then fits 128
// registers, i.e. FOUR waves per SIMD (MINW = 4) instead of three: the loop is bound by issue stalls that more
// resident waves can fill (PMC: something issues only 70 % of the cycles at three waves).
// Two LDS buffers: the DMA of tile kt + 1 is issued at the top of tile kt into the buffer tile kt - 1 used (every wave
// left it before the barrier that ended tile kt - 1); __syncthreads() at the end of the tile is vmcnt(0) + s_barrier
// (hipcc drains LDS-DMA in front of it), after which the tile is visible to every wave.
// MSUM: the denominator by the ones-MFMA (16 accumulators, 4 MFMAs per tile) instead of 32 v_add_f32 per tile
// NOMAX: no row maximum in the steady state.  The tile's probabilities are exponentiated against the current reference
// straight away and their row sum (needed anyway) is the overflow detector: any p above 2^kDefer -- or an inf -- makes
// the sum exceed the trigger; only then (rare, wave-uniform) the scores are recomputed from the K tile still in LDS,
// the exact row maximum is taken and O, l and the reference are re-based.  23 v_max per tile leave the hot path.
#ifndef GWW_ATT_NBUF
#define GWW_ATT_NBUF 2   // K / V buffers of k_attention_dma_bf16.  3 (round 4 experiment, tools/attfwd_exp.py NBUF=2 NBUF=3): the
                         // LDS-DMA of tile kt + 2 issued at the top of tile kt behind counted waits -- correct, 4 % SLOWER (1.073 /
                         // 1.084 against 1.031 / 1.042 ms on one box): the K / V pieces are L2 hits that land within a tile time
#endif
template <int MINW, bool MSUM, bool NOMAX>
__global__ __launch_bounds__(256, MINW) void k_attention_dma_bf16(const unsigned short* __restrict__ qkv,
                                                                 unsigned short* __restrict__ ctx,
                                                                 float* __restrict__ lse, int T, int H, int q_tiles,
                                                                 int qt0) {
  // NB = 3 (round 4 experiment, off): THREE K / V buffers, the LDS-DMA of tile kt + 2 issued at the top of tile kt and only tile
  // kt + 1's pieces awaited at its end (asm DMA + counted vmcnt: hipcc's __syncthreads() drains every LDS-DMA in flight) -- a
  // tile's pieces get two tile times to arrive instead of one.  48 KB per workgroup: three workgroups still fit a CU.
  constexpr int NB = GWW_ATT_NBUF;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NB * 2 * KB * DH * 2];   // K0 .. | V0 .., 8 KB each
  constexpr int TILE_BYTES = KB * DH * 2, NW = 4;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned nblk = gridDim.x, per = nblk >> 3;
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;   // XCD-aware order
  const int qt = qt0 + wid % q_tiles;
  const int bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * DH;
  const long row_stride = 3L * d;
  const unsigned short* base = qkv + (long)b * T * row_stride;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;
  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * (NW * 32) + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;
  const bool wave_live = qt * (NW * 32) + wave_u * 32 < T;   // wave-uniform
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * row_stride + 16 * s + 8 * hh);

  // ---- DMA roles: piece j (0, 1) of this wave = tile rows 8 (2 wave + j) .. + 7; lane l lands at LDS (row l >> 3,
  // 16-byte position l & 7), which must hold the chunk the swizzled reads expect there
  const int n_kt = (T + KB - 1) / KB;
  const bool ragged = (T % KB) != 0;
  unsigned koff[2], voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
    koff[j] = (unsigned)(row * (int)row_stride * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
    voff[j] = (unsigned)(row * (int)row_stride * 2 + ((pos ^ (((row >> 1) & 1) << 2)) << 4));
  }
  auto dma = [&](int kt, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const char* kb = reinterpret_cast<const char*>(kp + (long)kt * KB * row_stride);   // wave-uniform
    const char* vb = reinterpret_cast<const char*>(vp + (long)kt * KB * row_stride);
    unsigned char* dk = lds + BUF * TILE_BYTES + (2 * wave_u) * 1024;
    unsigned char* dv = dk + NB * TILE_BYTES;
    if constexpr (NB == 3) {
      // one M0 (wave-uniform LDS destination) and one SGPR base per image; the piece enters as the instruction's immediate,
      // which the hardware adds to the LDS address -- its global side is folded into the per-lane offset
      const unsigned mk = (unsigned)(unsigned long long)(lds_ptr)dk, mv = (unsigned)(unsigned long long)(lds_ptr)dv;
      unsigned ko[2], vo[2];
      if (kt != n_kt - 1 || !ragged) {
        ko[0] = koff[0]; ko[1] = koff[1]; vo[0] = voff[0]; vo[1] = voff[1];
      } else {   // ragged last tile: rows past T - 1 read row T - 1
        const int last_row = T - 1 - kt * KB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
          const int rc = row < last_row ? row : last_row;
          ko[j] = (unsigned)(rc * (int)row_stride * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
          vo[j] = (unsigned)(rc * (int)row_stride * 2 + ((pos ^ (((row >> 1) & 1) << 2)) << 4));
        }
      }
      // (one M0 per piece: an instruction immediate would be added to the GLOBAL address too, and a clamped row of the ragged
      //  tile has no 1 KiB to give back -- the first version of this path wrapped its offset below zero and faulted on T = 1)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(ko[j]), "s"(kb), "s"(mk + 1024u * j) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo[j]), "s"(vb), "s"(mv + 1024u * j) : "memory");
      }
    } else if (kt != n_kt - 1 || !ragged) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        __builtin_amdgcn_global_load_lds((g_ptr)(kb + koff[j]), (lds_ptr)(dk + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(vb + voff[j]), (lds_ptr)(dv + j * 1024), 16, 0, 0);
      }
    } else {   // ragged last tile: rows past T - 1 read row T - 1 (masked afterwards; never past the tensor)
      const int last_row = T - 1 - kt * KB;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
        const int rc = row < last_row ? row : last_row;
        const unsigned ko = (unsigned)(rc * (int)row_stride * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
        const unsigned vo = (unsigned)(rc * (int)row_stride * 2 + ((pos ^ (((row >> 1) & 1) << 2)) << 4));
        __builtin_amdgcn_global_load_lds((g_ptr)(kb + ko), (lds_ptr)(dk + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(vb + vo), (lds_ptr)(dv + j * 1024), 16, 0, 0);
      }
    }
  };

  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_colbyte = (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;
  const unsigned char* kbase[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) kbase[s] = lds + k_off(r, 2 * s + hh);
  const unsigned char* vbase[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) vbase[n] = lds + NB * TILE_BYTES + v_off(4 * hh + tr_q, 64 * n + tr_colbyte);

  f32x16 ot[2], lt;
#pragma unroll
  for (int j = 0; j < 16; ++j) { ot[0][j] = 0.f; ot[1][j] = 0.f; lt[j] = 0.f; }
  bf16x8 ones, mref;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ones[j] = (__bf16)1.0f; mref[j] = (__bf16)0.0f; }
  float l_run = 0.f, m_run = 0.f;
  constexpr float kDeferL2 = 8.0f * kLog2e;
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

  auto tile = [&](int kt, auto buf_c, auto first_c, auto masked_c, bool masked_rt = false) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
    // (NB = 3: the ragged-tile flag is a wave-uniform RUNTIME value -- the steady loop then holds three tile bodies and nothing
    //  else; as a template flag the remainder paths added six more copies and 162 spilled registers around them)
    const bool MASKED = NB == 3 ? masked_rt : decltype(masked_c)::value;
    if (kt + NB - 1 < n_kt) dma(kt + NB - 1, std::integral_constant<int, (BUF + NB - 1) % NB>{});
    // Work that cannot contribute is skipped (wave-uniform tests; T = 1500: 2 % + 2 % of the launch): a wave whose 32 query
    // rows all lie past T - 1 (the last query tile's tail) only requests its pieces and joins the barrier; the ragged last
    // key tile computes its second key half only if a valid key lies in it (1500 = 23 x 64 + 28: it does not)
    const bool g1_live = !MASKED || (T - kt * KB) > 32;
    if (wave_live) {
    f32x16 st[2];
    auto scores = [&]() {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x16 z;
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j] = 0.f;
        if (MASKED && g == 1 && !g1_live) {
#pragma unroll
          for (int j = 0; j < 16; ++j) st[1][j] = -INFINITY;
          continue;
        }
        if constexpr (FIRST) st[g] = z;
        else st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, mref, z, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase[s] + BUF * TILE_BYTES + g * (32 * 128));
          st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[g], 0, 0, 0);
        }
      }
      if (MASKED) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
            if (key >= T) st[g][j] = -INFINITY;
          }
      }
    };
    // exact row maximum of the tile (both key halves), then move the reference of the rows that need it
    auto rebase = [&]() {
      float tmax = st[0][0];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, st[g][j]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      if (FIRST || NOMAX || __builtin_amdgcn_ballot_w64(tmax > kDeferL2) != 0) {
        const float want = m_run + (FIRST ? tmax : fmaxf(tmax, 0.f));
        const __bf16 hi = (__bf16)(-want);
        const __bf16 lo = (__bf16)(-want - (float)hi);
        const float m_new = -((float)hi + (float)lo);
        const float dm = m_new - m_run;
        const float alpha = __builtin_amdgcn_exp2f(-dm);
        m_run = m_new;
        if constexpr (!FIRST && !MSUM) l_run *= alpha;
        mref[0] = hh == 0 ? hi : (__bf16)0.0f;
        mref[1] = hh == 0 ? lo : (__bf16)0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if constexpr (!FIRST) {
            ot[0][j] *= alpha;
            ot[1][j] *= alpha;
            if constexpr (MSUM) lt[j] *= alpha;
          }
          st[0][j] -= dm;
          st[1][j] -= dm;
        }
      }
    };
    scores();
    if constexpr (FIRST || !NOMAX) rebase();
    float ps = 0.f;
    auto exps = [&]() {
      ps = 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          st[g][j] = __builtin_amdgcn_exp2f(st[g][j]);
          if constexpr (!MSUM || NOMAX) ps += st[g][j];
        }
    };
    exps();
    if constexpr (NOMAX && !FIRST) {
      // 2^kDeferL2 = e^8 = 2981: one probability above it, or an inf, lifts the half-row sum over the trigger
      if (__builtin_amdgcn_ballot_w64(!(ps <= 2981.0f)) != 0) {   // wave-uniform, rare
        scores();
        rebase();
        exps();
      }
    }
    if constexpr (!MSUM) l_run += ps;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (MASKED && g == 1 && !g1_live) continue;   // (its probabilities are exp2(-inf) = 0)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = cvt8(st[g], 8 * s);
        if constexpr (MSUM) lt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lt, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const unsigned char* vb = vbase[n] + BUF * TILE_BYTES + (32 * g + 16 * s) * 128;
          const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)vb);
          const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb + 8 * 128));
          bf16x8 vf;
          vf[0] = lo4[0]; vf[1] = lo4[1]; vf[2] = lo4[2]; vf[3] = lo4[3];
          vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
          ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[n], 0, 0, 0);
        }
      }
    }
    }   // wave_live
    if constexpr (NB == 3) {   // tile kt + 1's four pieces have landed; tile kt + 2's (issued at this tile's top) may stay in flight
      if (kt + 2 < n_kt) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      __syncthreads();   // vmcnt(0): this wave's pieces of tile kt + 1 have landed; barrier: everybody's have
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;

  if constexpr (NB == 3) {
    using P2 = std::integral_constant<int, 2>;
    dma(0, P0{});
    if (n_kt > 1) {
      dma(1, P1{});
      asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // (touch the Q fragments here: hipcc then waits for their loads in front of the loop; left pending into the loop header,
    //  its counter model re-waits for them in every tile with counts that also drain the K / V requests in flight)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(qf[s4]));
    tile(0, P0{}, Yes{}, No{}, n_kt == 1 && ragged);
    // steady state: buffers 1, 2, 0, 1, ... (compile-time in every tile body; the guards are scalar branches)
    for (int kt = 1; kt < n_kt; kt += 3) {
      tile(kt, P1{}, No{}, No{}, ragged && kt == n_kt - 1);
      if (kt + 1 < n_kt) tile(kt + 1, P2{}, No{}, No{}, ragged && kt + 1 == n_kt - 1);
      if (kt + 2 < n_kt) tile(kt + 2, P0{}, No{}, No{}, ragged && kt + 2 == n_kt - 1);
    }
  } else {
    dma(0, P0{});
    __syncthreads();
    if (n_kt == 1) {
      if (ragged) tile(0, P0{}, Yes{}, Yes{});
      else tile(0, P0{}, Yes{}, No{});
    } else {
      tile(0, P0{}, Yes{}, No{});
      int kt = 1;
      for (; kt + 2 <= n_kt - 1; kt += 2) {
        tile(kt, P1{}, No{}, No{});
        tile(kt + 1, P0{}, No{}, No{});
      }
      if (kt == n_kt - 2) {
        tile(kt, P1{}, No{}, No{});
        if (ragged) tile(kt + 1, P0{}, No{}, Yes{});
        else tile(kt + 1, P0{}, No{}, No{});
      } else {
        if (ragged) tile(kt, P1{}, No{}, Yes{});
        else tile(kt, P1{}, No{}, No{});
      }
    }
  }

  float l_tot;
  if constexpr (MSUM) {
    l_tot = lt[0];
  } else {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  if (lse && q_row < T && hh == 0)
    lse[((long)b * H + h) * T + q_row] = (m_run + __log2f(l_tot)) * 0.69314718055994530942f;
#if GWW_ATT_EPI16
  {
    // whole 16-byte row pieces: lanes l and l + 32 hold neighbouring 4-column groups of the same row, so one
    // v_permlane32_swap per packed word hands the low lane columns 8 c .. 8 c + 7 of an even c and the high lane those of
    // the odd c + 1 -- 4 stores of 16 bytes per lane instead of 8 of 8 bytes (the epilogue tail is store-issue bound)
    unsigned short* orow = ctx + ((long)b * T + q_row) * d + h * DH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const int c0 = 2 * cp, c1 = 2 * cp + 1;
        const unsigned a0 = pack2bf(ot[n][4 * c0] * inv, ot[n][4 * c0 + 1] * inv), a1 = pack2bf(ot[n][4 * c0 + 2] * inv, ot[n][4 * c0 + 3] * inv);
        const unsigned b0 = pack2bf(ot[n][4 * c1] * inv, ot[n][4 * c1 + 1] * inv), b1 = pack2bf(ot[n][4 * c1 + 2] * inv, ot[n][4 * c1 + 3] * inv);
        // swap(x, y): x's upper 32 lanes <-> y's lower 32 lanes
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        // low lanes: s?[0] = own group of c0, s?[1] = the partner's (hh = 1) group of c0; high lanes: s?[0] = the partner's
        // (hh = 0) group of c1, s?[1] = own group of c1
        u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if (q_row < T) *reinterpret_cast<u32x4*>(orow + 32 * n + 8 * (hh ? c1 : c0)) = o;
      }
  }
#else
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
#endif
}

#ifdef GWW_LAB   // laboratory variant: compiled into libgww_lab.so only (make LAB=1)
// ---------------------------------------------------------------------------------------------------------
// Two-waves-per-SIMD "ping-pong" form (GWW_ATT_VAR = 8; correct -- every test_attention_log2q case runs it too -- but
// MEASURED SLOWER than the three-waves-per-SIMD default: 1.10 - 1.21 ms against 0.98 - 1.03 per whisper-tiny layer at
// B = 256, with any of the priority settings below; kept as the starting point of a hand-placed version): a workgroup = 8 waves = 256 query rows of one
// (b, h); waves w and w + 4 share a SIMD and run ONE BARRIER apart.  A 64-key tile is two sections per wave:
//   M section: S(j) = K(j) Q^T (9 - 10 MFMAs, the running reference enters as the first of them) and, software-pipelined,
//              O += V(j-1)^T P(j-1)^T (8 MFMAs) -- fragment reads, MFMAs, the LDS-DMA request of tile j + 2, nothing else;
//   V section: the softmax of tile j on the vector ALU -- 32 v_exp_f32, the row sum, 16 v_cvt_pk -- and nothing else.
// Between two barriers one wave of every SIMD is in its M section and its partner in its V section: the matrix pipe and
// the vector ALU of a SIMD work at the same time BY CONSTRUCTION.  The three-waves-per-SIMD kernel above relies on waves of
// independent workgroups happening to be out of phase, with one vmcnt(0) + barrier per tile, and measured 42 % matrix-pipe
// busy with the vector ALU 53 % busy (profiles/r02_pmc_attention.md) -- each unit idle most of the time the other works.
// K / V ring of four tiles each (64 KB of LDS, one workgroup per CU): the tile two ahead is requested in the M section of
// tile j into the slot tile j - 2 left (its V was last read one barrier earlier), counted vmcnt(2), never 0 in the loop.
// Arithmetic identical to k_attention_dma_bf16<3, false, true>: q in log2 units, reference through the matrix pipe, no
// row maximum in the steady state (the row sum is the overflow detector), denominators on the VALU.
constexpr int PP_SLOTS = 4;
#ifndef GWW_PP_PRIO
#define GWW_PP_PRIO 0   // 0: no priority changes; 1: the M section at priority 1; 2: waves 4 .. 7 at priority 1 throughout; 3: the V section at priority 1
#endif
template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 256 : 512, 1) void k_attention_pp_bf16(const unsigned short* __restrict__ qkv,
                                                              unsigned short* __restrict__ ctx,
                                                              float* __restrict__ lse, int T, int H, int q_tiles,
                                                              int qt0) {
  constexpr int TILE_BYTES = KB * DH * 2;   // 8 KB
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PP_SLOTS * TILE_BYTES];   // K ring | V ring
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool g1 = wave >= 4;
  // MODE 1 (interleaved): FOUR waves, one per SIMD, each with the whole 512-register file (two score sets + both fragment
  // sets + O do not fit 256 registers: 49 spilled, and a spill is an uncounted entry of the in-order vmcnt queue)
  constexpr int NWV = MODE == 1 ? 4 : 8, PPW = 8 / NWV;   // waves per workgroup, LDS-DMA pieces per wave, operand and tile
  const unsigned nblk = gridDim.x, per = nblk >> 3;
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;   // XCD-aware order
  const int qt = qt0 + wid % q_tiles;
  const int bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * DH;
  const long row_stride = 3L * d;
  const unsigned short* base = qkv + (long)b * T * row_stride;
  const unsigned short* qp = base + h * DH;
  const unsigned short* kp = base + d + h * DH;
  const unsigned short* vp = base + 2 * d + h * DH;
  const int r = lane & 31, hh = lane >> 5;
  const int q_row = qt * (NWV * 32) + wave * 32 + r;
  const int q_ld = q_row < T ? q_row : T - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * row_stride + 16 * s + 8 * hh);

  // ---- LDS-DMA: a tile is 8 pieces of 8 rows; wave w requests piece w of K and of V.  Lane l lands at (row l >> 3,
  // 16-byte position l & 7), which must hold the chunk the swizzled reads expect there
  const int n_kt = (T + KB - 1) / KB;
  const bool ragged = (T % KB) != 0;
  const int dpos = lane & 7;
  int drow[PPW];
  unsigned koff[PPW], voff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    drow[j] = 8 * (wave + NWV * j) + (lane >> 3);
    koff[j] = (unsigned)(drow[j] * (int)row_stride * 2 + ((dpos ^ ((drow[j] >> 1) & 7)) << 4));
    voff[j] = (unsigned)(drow[j] * (int)row_stride * 2 + ((dpos ^ (((drow[j] >> 1) & 1) << 2)) << 4));
  }
  auto dma = [&](int kt) {
    const char* kb = reinterpret_cast<const char*>(kp + (long)kt * KB * row_stride);   // wave-uniform
    const char* vb = reinterpret_cast<const char*>(vp + (long)kt * KB * row_stride);
    if (kt != n_kt - 1 || !ragged) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        unsigned char* dk = lds + (kt & (PP_SLOTS - 1)) * TILE_BYTES + (wave + NWV * j) * 1024;
        __builtin_amdgcn_global_load_lds((g_ptr)(kb + koff[j]), (lds_ptr)dk, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(vb + voff[j]), (lds_ptr)(dk + PP_SLOTS * TILE_BYTES), 16, 0, 0);
      }
    } else {   // ragged last tile: rows past T - 1 read row T - 1 (masked afterwards; never past the tensor)
      const int last_row = T - 1 - kt * KB;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        unsigned char* dk = lds + (kt & (PP_SLOTS - 1)) * TILE_BYTES + (wave + NWV * j) * 1024;
        const int rc = drow[j] < last_row ? drow[j] : last_row;
        const unsigned ko = (unsigned)(rc * (int)row_stride * 2 + ((dpos ^ ((drow[j] >> 1) & 7)) << 4));
        const unsigned vo = (unsigned)(rc * (int)row_stride * 2 + ((dpos ^ (((drow[j] >> 1) & 1) << 2)) << 4));
        __builtin_amdgcn_global_load_lds((g_ptr)(kb + ko), (lds_ptr)dk, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((g_ptr)(vb + vo), (lds_ptr)(dk + PP_SLOTS * TILE_BYTES), 16, 0, 0);
      }
    }
  };

  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_colbyte = (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;
  const unsigned char* kbase[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) kbase[s] = lds + k_off(r, 2 * s + hh);
  const unsigned char* vbase[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) vbase[n] = lds + PP_SLOTS * TILE_BYTES + v_off(4 * hh + tr_q, 64 * n + tr_colbyte);

  f32x16 ot[2], st[2];
#pragma unroll
  for (int j = 0; j < 16; ++j) { ot[0][j] = 0.f; ot[1][j] = 0.f; }
  bf16x8 ones, mref, pf[2][2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ones[j] = (__bf16)1.0f; mref[j] = (__bf16)0.0f; }
  float l_run = 0.f, m_run = 0.f;
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

  ASTAMP_DECL
  // S(kt) from the K tile in slot SLOT
  auto scores = [&](auto slot_c, auto first_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x16 z;
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = 0.f;
      if constexpr (FIRST) st[g] = z;
      else st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, mref, z, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase[s] + SLOT * TILE_BYTES + g * (32 * 128));
        st[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[g], 0, 0, 0);
      }
    }
  };
  // O += V(slot)^T P^T with the probabilities of the tile before (pf).  The transposed fragment reads are inline asm: through
  // the intrinsic hipcc cannot tell them from the LDS-DMA destinations and puts s_waitcnt vmcnt(0) in front of the first
  // one -- a drain of the ring in every M section.  All sixteen go out first (ahead of the score MFMAs), one lgkmcnt(0) that
  // names the fragments as its operands stands between them and the MFMAs that read them.
  u32x2 vfr[2][2][2][2];   // [g][s][n][lo / hi half]
  auto pv_reads = [&](int slot) {   // (not a generic lambda: hipcc rejects asm operands that name captured arrays in those)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const unsigned va = (unsigned)(unsigned long long)(lds_ptr)const_cast<unsigned char*>(vbase[n]) + slot * TILE_BYTES;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vfr[g][s][n][0]) : "v"(va), "n"((32 * g + 16 * s) * 128));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vfr[g][s][n][1]) : "v"(va), "n"((32 * g + 16 * s + 8) * 128));
        }
    }
  };
  auto pv_mfma = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vfr[0][0][0][0]), "+v"(vfr[0][0][0][1]), "+v"(vfr[0][0][1][0]), "+v"(vfr[0][0][1][1]),
                   "+v"(vfr[0][1][0][0]), "+v"(vfr[0][1][0][1]), "+v"(vfr[0][1][1][0]), "+v"(vfr[0][1][1][1]),
                   "+v"(vfr[1][0][0][0]), "+v"(vfr[1][0][0][1]), "+v"(vfr[1][0][1][0]), "+v"(vfr[1][0][1][1]),
                   "+v"(vfr[1][1][0][0]), "+v"(vfr[1][1][0][1]), "+v"(vfr[1][1][1][0]), "+v"(vfr[1][1][1][1]));
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const u32x4 v4 = {vfr[g][s][n][0][0], vfr[g][s][n][0][1], vfr[g][s][n][1][0], vfr[g][s][n][1][1]};
          ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v4), pf[g][s], ot[n], 0, 0, 0);
        }
  };
  // ---- the M section of a steady-state tile with every instruction PLACED (inline asm, program order = issue order): left to
  // hipcc the section read two K fragments, waited lgkmcnt(0), issued one MFMA, read the next two ... -- the LDS latency in
  // front of every MFMA: 1 090 ticks per tile where the 18 MFMAs need 650 (tools/stamp_att.py).  Here: the eight K reads and
  // the first eight V reads go out back to back, the score MFMAs start as soon as the K fragments are in (the second half of
  // the V reads rides behind the first score MFMAs), the P V MFMAs follow without a gap.
  auto m_section = [&](int kslot, int vslot) {   // (not a generic lambda: asm operands name captured arrays)
    u32x4 kfr[2][4];
    unsigned ka[4], va[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ka[i] = (unsigned)(unsigned long long)(lds_ptr)const_cast<unsigned char*>(kbase[i]) + kslot * TILE_BYTES;
#pragma unroll
    for (int n = 0; n < 2; ++n) va[n] = (unsigned)(unsigned long long)(lds_ptr)const_cast<unsigned char*>(vbase[n]) + vslot * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(kfr[0][i]) : "v"(ka[i]));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kfr[1][i]) : "v"(ka[i]));
    }
#define GWW_PP_VREAD(G, S, N)                                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vfr[G][S][N][0]) : "v"(va[N]), "n"((32 * G + 16 * S) * 128));         \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vfr[G][S][N][1]) : "v"(va[N]), "n"((32 * G + 16 * S + 8) * 128));
    GWW_PP_VREAD(0, 0, 0) GWW_PP_VREAD(0, 0, 1) GWW_PP_VREAD(0, 1, 0) GWW_PP_VREAD(0, 1, 1)
    // the K fragments (the eight oldest reads) are in; the eight V reads behind them may still fly
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(kfr[0][0]), "+v"(kfr[0][1]), "+v"(kfr[0][2]), "+v"(kfr[0][3]), "+v"(kfr[1][0]), "+v"(kfr[1][1]),
                   "+v"(kfr[1][2]), "+v"(kfr[1][3]));
#define GWW_PP_MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
    // (hipcc may materialise `ones` / `mref` right in front: a VALU write of an MFMA source needs two wait states)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(st[0]) : "v"(ones), "v"(mref));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(st[1]) : "v"(ones), "v"(mref));
    GWW_PP_MFMA(st[0], kfr[0][0], qf[0]);
    GWW_PP_MFMA(st[1], kfr[1][0], qf[0]);
    GWW_PP_VREAD(1, 0, 0) GWW_PP_VREAD(1, 0, 1) GWW_PP_VREAD(1, 1, 0) GWW_PP_VREAD(1, 1, 1)
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      GWW_PP_MFMA(st[0], kfr[0][i], qf[i]);
      GWW_PP_MFMA(st[1], kfr[1][i], qf[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vfr[0][0][0][0]), "+v"(vfr[0][0][0][1]), "+v"(vfr[0][0][1][0]), "+v"(vfr[0][0][1][1]),
                   "+v"(vfr[0][1][0][0]), "+v"(vfr[0][1][0][1]), "+v"(vfr[0][1][1][0]), "+v"(vfr[0][1][1][1]),
                   "+v"(vfr[1][0][0][0]), "+v"(vfr[1][0][0][1]), "+v"(vfr[1][0][1][0]), "+v"(vfr[1][0][1][1]),
                   "+v"(vfr[1][1][0][0]), "+v"(vfr[1][1][0][1]), "+v"(vfr[1][1][1][0]), "+v"(vfr[1][1][1][1]));
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const u32x4 v4 = {vfr[g][sk][n][0][0], vfr[g][sk][n][0][1], vfr[g][sk][n][1][0], vfr[g][sk][n][1][1]};
          GWW_PP_MFMA(ot[n], v4, pf[g][sk]);
        }
#undef GWW_PP_MFMA
#undef GWW_PP_VREAD
  };
  // exact row maximum of the tile (both key halves), then move the reference
  auto rebase = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    float tmax = st[0][0];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, st[g][j]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
      tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float want = m_run + (FIRST ? tmax : fmaxf(tmax, 0.f));
    const __bf16 hi = (__bf16)(-want);
    const __bf16 lo = (__bf16)(-want - (float)hi);
    const float m_new = -((float)hi + (float)lo);
    const float dm = m_new - m_run;
    const float alpha = __builtin_amdgcn_exp2f(-dm);
    m_run = m_new;
    if constexpr (!FIRST) l_run *= alpha;
    mref[0] = hh == 0 ? hi : (__bf16)0.0f;
    mref[1] = hh == 0 ? lo : (__bf16)0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if constexpr (!FIRST) {
        ot[0][j] *= alpha;
        ot[1][j] *= alpha;
      }
      st[0][j] -= dm;
      st[1][j] -= dm;
    }
  };
  float ps = 0.f;
  auto exps = [&]() {
    ps = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        st[g][j] = __builtin_amdgcn_exp2f(st[g][j]);
        ps += st[g][j];
      }
  };

  if constexpr (MODE == 1) {
    // ================= interleaved form (GWW_ATT_VAR = 9): no wave-against-wave alternation.  What the ping-pong form showed
    // is that the MFMAs of one wave and the VALU stream of its partner do NOT share a SIMD's vector-issue port well (the
    // partner's section is simply appended: profiles/r03_attention_pp.md), while ONE wave's own VALU instructions do run in
    // the shadow of its own MFMAs (the fused MLP block's GELU: 34 cycles per MFMA with eight VALU operations behind each).
    // So every wave software-pipelines its own tiles: iteration kt issues S(kt) = K(kt) Q^T -- ten asm MFMAs in program
    // order -- with the softmax of tile kt - 1 (exp, row sum, bf16 convert) cut into ten chunks placed between them, then
    // O += V(kt-1)^T P(kt-1)^T.  Two score sets (sA, sB) alternate as "being produced" / "being exponentiated".
    f32x16 sA[2], sB[2];
    auto scores_into = [&](f32x16 (&sc)[2], int kslot, bool first) {   // compiler-scheduled: prologue, epilogue, rare path
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x16 z;
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j] = 0.f;
        if (first) sc[g] = z;
        else sc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, mref, z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase[i] + kslot * TILE_BYTES + g * (32 * 128));
          sc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[i], sc[g], 0, 0, 0);
        }
      }
    };
    auto mask_tail = [&](f32x16 (&sc)[2], int kt) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (key >= T) sc[g][j] = -INFINITY;
        }
    };
    // exact row maximum of the tile, new reference; returns the shift of the reference (scores computed against the old one
    // must be lowered by it)
    auto rebase_on = [&](f32x16 (&sc)[2], bool first) -> float {
      float tmax = sc[0][0];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, sc[g][j]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float want = m_run + (first ? tmax : fmaxf(tmax, 0.f));
      const __bf16 hi = (__bf16)(-want);
      const __bf16 lo = (__bf16)(-want - (float)hi);
      const float m_new = -((float)hi + (float)lo);
      const float dm = m_new - m_run;
      const float alpha = __builtin_amdgcn_exp2f(-dm);
      m_run = m_new;
      if (!first) l_run *= alpha;
      mref[0] = hh == 0 ? hi : (__bf16)0.0f;
      mref[1] = hh == 0 ? lo : (__bf16)0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (!first) {
          ot[0][j] *= alpha;
          ot[1][j] *= alpha;
        }
        sc[0][j] -= dm;
        sc[1][j] -= dm;
      }
      return dm;
    };
    auto exps_on = [&](f32x16 (&sc)[2]) {
      ps = 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          sc[g][j] = __builtin_amdgcn_exp2f(sc[g][j]);
          ps += sc[g][j];
        }
    };
    auto cvt_all = [&](f32x16 (&sc)[2]) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) pf[g][sk] = cvt8(sc[g], 8 * sk);
    };
    // one steady-state iteration: cur <- S(kt) (asm MFMAs), prv = S(kt - 1) -> probabilities (chunks between the MFMAs),
    // then O += V(kt - 1)^T P(kt - 1)^T
    auto il_iter = [&](f32x16 (&prv)[2], f32x16 (&cur)[2], int kt) {
      const int kslot = kt & (PP_SLOTS - 1), vslot = (kt - 1) & (PP_SLOTS - 1);
      u32x4 kfr[2][4];
      unsigned ka[4], va[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) ka[i] = (unsigned)(unsigned long long)(lds_ptr)const_cast<unsigned char*>(kbase[i]) + kslot * TILE_BYTES;
#pragma unroll
      for (int n = 0; n < 2; ++n) va[n] = (unsigned)(unsigned long long)(lds_ptr)const_cast<unsigned char*>(vbase[n]) + vslot * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(kfr[0][i]) : "v"(ka[i]));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kfr[1][i]) : "v"(ka[i]));
      }
#define GWW_IL_VREAD(G, S, N)                                                                                                       \
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vfr[G][S][N][0]) : "v"(va[N]), "n"((32 * G + 16 * S) * 128));       \
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vfr[G][S][N][1]) : "v"(va[N]), "n"((32 * G + 16 * S + 8) * 128));
      // (the V fragments are requested later, into the registers the K fragments leave: sixteen more live registers spill)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(kfr[0][0]), "+v"(kfr[0][1]), "+v"(kfr[0][2]), "+v"(kfr[0][3]), "+v"(kfr[1][0]), "+v"(kfr[1][1]),
                     "+v"(kfr[1][2]), "+v"(kfr[1][3]));
      ASTAMP(0);
      ps = 0.f;
      // chunk C of the softmax of tile kt - 1: exponentials and row sum of scores [32 C / 10, 32 (C + 1) / 10), a bf16 convert
      // where eight neighbours are complete; the empty asm pins it between the MFMA before and the MFMA behind it (it names
      // what the chunk produced AND what the next chunk will read, so hipcc can move neither across it)
#define GWW_IL_CHUNK(C)                                                                                       \
      {                                                                                                       \
        _Pragma("unroll") for (int idx = (C) * 32 / 10; idx < ((C) + 1) * 32 / 10; ++idx) {                   \
          float tv = prv[idx >> 4][idx & 15];                                                                 \
          tv = __builtin_amdgcn_exp2f(tv);                                                                    \
          ps += tv;                                                                                           \
          prv[idx >> 4][idx & 15] = tv;                                                                       \
        }                                                                                                     \
        if ((C) == 3) pf[0][0] = cvt8(prv[0], 0);                                                             \
        if ((C) == 5) pf[0][1] = cvt8(prv[0], 8);                                                             \
        if ((C) == 8) pf[1][0] = cvt8(prv[1], 0);                                                             \
        if ((C) == 9) pf[1][1] = cvt8(prv[1], 8);                                                             \
        asm volatile("" : "+v"(ps), "+v"(prv[0]), "+v"(prv[1]), "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1])); \
      }
// (s_nop 1 in front of every asm MFMA: above 256 registers hipcc parks values in AGPRs and fetches them with v_accvgpr_read
// right in front of the statement -- a VALU write of an MFMA source needs two wait states; tools/audit_asm_mfma.py)
#define GWW_IL_MFMA(ACC, A, B) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define GWW_IL_MFMA_O(ACC, A, B) GWW_IL_MFMA(ACC, A, B)
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(cur[0]) : "v"(ones), "v"(mref));
      GWW_IL_CHUNK(0)
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(cur[1]) : "v"(ones), "v"(mref));
      GWW_IL_CHUNK(1)
      GWW_IL_MFMA(cur[0], kfr[0][0], qf[0]);
      GWW_IL_CHUNK(2)
      GWW_IL_MFMA(cur[1], kfr[1][0], qf[0]);
      GWW_IL_CHUNK(3)
      GWW_IL_MFMA(cur[0], kfr[0][1], qf[1]);
      GWW_IL_CHUNK(4)
      GWW_IL_MFMA(cur[1], kfr[1][1], qf[1]);
      GWW_IL_VREAD(0, 0, 0) GWW_IL_VREAD(0, 0, 1) GWW_IL_VREAD(0, 1, 0) GWW_IL_VREAD(0, 1, 1)
      GWW_IL_CHUNK(5)
      GWW_IL_MFMA(cur[0], kfr[0][2], qf[2]);
      GWW_IL_CHUNK(6)
      GWW_IL_MFMA(cur[1], kfr[1][2], qf[2]);
      GWW_IL_CHUNK(7)
      GWW_IL_MFMA(cur[0], kfr[0][3], qf[3]);
      GWW_IL_CHUNK(8)
      GWW_IL_MFMA(cur[1], kfr[1][3], qf[3]);
      GWW_IL_VREAD(1, 0, 0) GWW_IL_VREAD(1, 0, 1) GWW_IL_VREAD(1, 1, 0) GWW_IL_VREAD(1, 1, 1)
      GWW_IL_CHUNK(9)
      ASTAMP(1);
      // 2^(8 log2 e) = e^8 = 2981: one probability above it, or an inf, lifts the half-row sum over the trigger
      if (__builtin_amdgcn_ballot_w64(!(ps <= 2981.0f)) != 0) {   // wave-uniform, rare: exact maximum, re-based O, l, reference
        scores_into(prv, vslot, false);
        const float dm = rebase_on(prv, false);
#pragma unroll
        for (int j = 0; j < 16; ++j) { cur[0][j] -= dm; cur[1][j] -= dm; }   // S(kt) was formed against the old reference
        exps_on(prv);
        cvt_all(prv);
      }
      l_run += ps;
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vfr[0][0][0][0]), "+v"(vfr[0][0][0][1]), "+v"(vfr[0][0][1][0]), "+v"(vfr[0][0][1][1]),
                     "+v"(vfr[0][1][0][0]), "+v"(vfr[0][1][0][1]), "+v"(vfr[0][1][1][0]), "+v"(vfr[0][1][1][1]),
                     "+v"(vfr[1][0][0][0]), "+v"(vfr[1][0][0][1]), "+v"(vfr[1][0][1][0]), "+v"(vfr[1][0][1][1]),
                     "+v"(vfr[1][1][0][0]), "+v"(vfr[1][1][0][1]), "+v"(vfr[1][1][1][0]), "+v"(vfr[1][1][1][1]));
      asm volatile("s_nop 1");   // the converts wrote MFMA sources
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int sk = 0; sk < 2; ++sk)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const u32x4 v4 = {vfr[g][sk][n][0][0], vfr[g][sk][n][0][1], vfr[g][sk][n][1][0], vfr[g][sk][n][1][1]};
            GWW_IL_MFMA_O(ot[n], v4, pf[g][sk]);
          }
      ASTAMP(2);
#undef GWW_IL_MFMA
#undef GWW_IL_MFMA_O
#undef GWW_IL_CHUNK
#undef GWW_IL_VREAD
      if (kt + 2 < n_kt) {
        dma(kt + 2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt + 1 landed (this wave's pieces); kt + 2 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      ASTAMP(3);
      __builtin_amdgcn_s_barrier();
      ASTAMP(4);
    };

    dma(0);
    if (n_kt > 1) dma(1);
    if (n_kt > 2) {
      dma(2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    scores_into(sA, 0, true);
    if (n_kt == 1 && ragged) mask_tail(sA, 0);
    rebase_on(sA, true);
    int kt = 1;
    for (; kt + 1 < n_kt; kt += 2) {
      il_iter(sA, sB, kt);
      il_iter(sB, sA, kt + 1);
    }
    if (kt < n_kt) {
      il_iter(sA, sB, kt);
      ++kt;
    }
    // the last tile: its softmax and P V (scheduled by hipcc)
    {
      const bool last_in_b = ((n_kt - 1) & 1) != 0;   // tile t sits in sA for even t
      auto finish = [&](f32x16 (&sc)[2]) {
        if (ragged && n_kt > 1) mask_tail(sc, n_kt - 1);
        exps_on(sc);
        if (n_kt > 1 && __builtin_amdgcn_ballot_w64(!(ps <= 2981.0f)) != 0) {
          scores_into(sc, (n_kt - 1) & (PP_SLOTS - 1), false);
          if (ragged) mask_tail(sc, n_kt - 1);
          rebase_on(sc, false);
          exps_on(sc);
        }
        l_run += ps;
        cvt_all(sc);
      };
      if (last_in_b) finish(sB); else finish(sA);
      pv_reads((n_kt - 1) & (PP_SLOTS - 1));
      pv_mfma();
      ASTAMP(6);
      ASTAMP_FLUSH
    }
  } else {
  // ---- one tile = M section | barrier | V section | barrier
  auto tile = [&](int kt, auto slot_c, auto first_c, auto masked_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr bool FIRST = decltype(first_c)::value, MASKED = decltype(masked_c)::value;
    // M section: matrix pipe only (+ the request of the tile two ahead)
    ASTAMP(5);
    if (GWW_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (FIRST) scores(slot_c, first_c);
    else m_section(SLOT, (SLOT + PP_SLOTS - 1) & (PP_SLOTS - 1));
    if (kt + 2 < n_kt) {
      dma(kt + 2);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // tile kt + 1 landed (this wave's pieces); kt + 2 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (GWW_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (g1) { ASTAMP(2); } else { ASTAMP(0); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ASTAMP(4);
    // V section: vector ALU only
    if (GWW_PP_PRIO == 3) __builtin_amdgcn_s_setprio(1);
    if constexpr (MASKED) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (key >= T) st[g][j] = -INFINITY;
        }
    }
    if constexpr (FIRST) rebase(first_c);
    exps();
    if constexpr (!FIRST) {
      // 2^(8 log2 e) = e^8 = 2981: one probability above it, or an inf, lifts the half-row sum over the trigger
      if (__builtin_amdgcn_ballot_w64(!(ps <= 2981.0f)) != 0) {   // wave-uniform, rare: exact maximum, re-based O, l, reference
        scores(slot_c, first_c);
        if constexpr (MASKED) {
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int key = kt * KB + 32 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
              if (key >= T) st[g][j] = -INFINITY;
            }
        }
        rebase(first_c);
        exps();
      }
    }
    l_run += ps;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int s = 0; s < 2; ++s) pf[g][s] = cvt8(st[g], 8 * s);
    if (GWW_PP_PRIO == 3) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (g1) { ASTAMP(3); } else { ASTAMP(1); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using Yes = std::true_type;
  using No = std::false_type;
  auto tile_rt = [&](int kt, bool masked) {   // slot by the tile number (the remainder of the unrolled loop)
    switch (kt & 3) {
      case 0: if (masked) tile(kt, std::integral_constant<int, 0>{}, No{}, Yes{}); else tile(kt, std::integral_constant<int, 0>{}, No{}, No{}); break;
      case 1: if (masked) tile(kt, std::integral_constant<int, 1>{}, No{}, Yes{}); else tile(kt, std::integral_constant<int, 1>{}, No{}, No{}); break;
      case 2: if (masked) tile(kt, std::integral_constant<int, 2>{}, No{}, Yes{}); else tile(kt, std::integral_constant<int, 2>{}, No{}, No{}); break;
      default: if (masked) tile(kt, std::integral_constant<int, 3>{}, No{}, Yes{}); else tile(kt, std::integral_constant<int, 3>{}, No{}, No{}); break;
    }
  };

  dma(0);
  if (n_kt > 1) {
    dma(1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (g1) __builtin_amdgcn_s_barrier();   // the trailing group: one barrier behind
  if (GWW_PP_PRIO == 2 && g1) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (n_kt == 1 && ragged) tile(0, std::integral_constant<int, 0>{}, Yes{}, Yes{});
  else tile(0, std::integral_constant<int, 0>{}, Yes{}, No{});
  {
    int kt = 1;
    for (; kt + 4 <= n_kt - 1; kt += 4) {   // whole groups of four full tiles, slots 1 2 3 0 (kt = 1 mod 4)
      tile(kt, std::integral_constant<int, 1>{}, No{}, No{});
      tile(kt + 1, std::integral_constant<int, 2>{}, No{}, No{});
      tile(kt + 2, std::integral_constant<int, 3>{}, No{}, No{});
      tile(kt + 3, std::integral_constant<int, 0>{}, No{}, No{});
    }
    for (; kt < n_kt; ++kt) tile_rt(kt, ragged && kt == n_kt - 1);
  }
  // the last tile's P V (the M section of a tile that does not exist)
  pv_reads((n_kt - 1) & 3);
  pv_mfma();
  if (!g1) __builtin_amdgcn_s_barrier();   // the leading group: the barrier its partner is one behind with
  ASTAMP(6);
  ASTAMP_FLUSH
  }   // MODE

  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  if (lse && q_row < T && hh == 0)
    lse[((long)b * H + h) * T + q_row] = (m_run + __log2f(l_tot)) * 0.69314718055994530942f;
  {
    unsigned short* orow = ctx + ((long)b * T + q_row) * d + h * DH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const int c0 = 2 * cp, c1 = 2 * cp + 1;
        const unsigned a0 = pack2bf(ot[n][4 * c0] * inv, ot[n][4 * c0 + 1] * inv), a1 = pack2bf(ot[n][4 * c0 + 2] * inv, ot[n][4 * c0 + 3] * inv);
        const unsigned b0 = pack2bf(ot[n][4 * c1] * inv, ot[n][4 * c1 + 1] * inv), b1 = pack2bf(ot[n][4 * c1 + 2] * inv, ot[n][4 * c1 + 3] * inv);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if (q_row < T) *reinterpret_cast<u32x4*>(orow + 32 * n + 8 * (hh ? c1 : c0)) = o;
      }
  }
}

#endif  // GWW_LAB
// last_tile_only: compute only the query tile that holds token T - 1 (the other rows of ctx are left untouched) --
// the pooled forward needs nothing else of the last layer's attention.
// q_log2: q was projected with log2(e) / 8 instead of 1 / 8 (every bf16 q panel of the encoder is packed that way unless
// GWW_ATT_LOG2Q=0) -> k_attention_dma_bf16 (default) or, GWW_ATT_VAR = 0 .. 3, the register-staged k_attention_l2_bf16.
// Measured per whisper-tiny layer at B = 256 (tools/run/att_ab.py, interleaved in one process, three boxes):
//   k_attention_bf16 (natural q, round 1)          1.16 - 1.25 ms
//   k_attention_l2_bf16 VAR 0 / 2                  1.18 / 1.06 - 1.13   (VAR 1: 1.41 - 1.50, VAR 3: 1.19 - 1.21)
//   k_attention_dma_bf16 VAR 4 / 5 / 6 / 7         1.00 - 1.06 / 1.08 / 1.07 - 1.12 / 1.02 - 1.04
// (an earlier software-pipelined two-waves-per-SIMD kernel measured 1.20 ms and was removed.)
bool attention_log2q_enabled() {
  static const bool off = lab_int("GWW_ATT_LOG2Q", 1) == 0;
  return !off;
}

int launch_attention_bf16(const void* qkv, void* ctx, int B, int T, int H, hipStream_t s, float* lse,
                          bool last_tile_only, bool q_log2) {
  GWW_REQUIRE(qkv && ctx, "attention_bf16: NULL operand");
  GWW_REQUIRE(B >= 0 && T > 0 && H > 0, "attention_bf16: bad shape B=%d T=%d H=%d", B, T, H);
  GWW_REQUIRE((((uintptr_t)qkv) & 15) == 0 && (((uintptr_t)ctx) & 15) == 0, "attention_bf16: 16-byte alignment");
  if (B == 0) return GWW_OK;
  // k_attention_w64_bf16 (attention_w64.hip: 64 query rows per wave, one wave per SIMD, the softmax hand-placed into the
  // MFMA gaps) is correct -- every test_attention_log2q case passes on it -- and MEASURED SLOWER than the kernel below:
  // 1.15 - 1.16 ms against 0.98 - 0.99 per whisper-tiny layer at B = 256 (profiles/r04_attention_w64.md: 56 % matrix-pipe
  // busy inside a workgroup's life, but a fifth of that life is the prologue / epilogue no second workgroup covers at one
  // wave per SIMD).  It is compiled into the laboratory build only (make LAB=1, GWW_ATT_W64=1).
#ifdef GWW_LAB
  if (q_log2 && !last_tile_only && lab_int("GWW_ATT_W64", 0) != 0) return launch_attention_w64_bf16(qkv, ctx, B, T, H, s, lse);
#endif
  const unsigned short* in = (const unsigned short*)qkv;
  unsigned short* out = (unsigned short*)ctx;
#ifdef GWW_LAB
  const int nw_env = (int)lab_int("GWW_ATT_WAVES", 0);   // tuning aid (natural-q kernel): 4 or 8
  const int var_env = (int)lab_int("GWW_ATT_VAR", 7);    // read per call: in-process A/B
  const bool pp = q_log2 && var_env >= 8;   // the two-waves-per-SIMD ping-pong kernel: 256 query rows per workgroup
  const int nw = ((nw_env == 8 && !q_log2) || (pp && var_env == 8)) ? 8 : 4;
#else
  const int nw = 4;
#endif
  const int all_tiles = (T + nw * 32 - 1) / (nw * 32);
  const int q_tiles = last_tile_only ? 1 : all_tiles, qt0 = last_tile_only ? all_tiles - 1 : 0;
  const long blocks = (long)q_tiles * B * H;
  GWW_REQUIRE(blocks < 2147483647L, "attention_bf16: grid too large");
#define GWW_DMA(MW, MS, NM) hipLaunchKernelGGL((k_attention_dma_bf16<MW, MS, NM>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, lse, T, H, q_tiles, qt0)
#ifdef GWW_LAB
  if (pp && var_env >= 9) {
    hipLaunchKernelGGL((k_attention_pp_bf16<1>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, lse, T, H, q_tiles, qt0);
  } else if (pp) {
    hipLaunchKernelGGL((k_attention_pp_bf16<0>), dim3((unsigned)blocks), dim3(512), 0, s, in, out, lse, T, H, q_tiles, qt0);
  } else if (q_log2) {
    const int var = var_env & 7;
#define GWW_L2(VV) hipLaunchKernelGGL((k_attention_l2_bf16<4, VV>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, lse, T, H, q_tiles, qt0)
    switch (var) {
      case 0: GWW_L2(0); break;
      case 1: GWW_L2(1); break;
      case 2: GWW_L2(2); break;
      case 3: GWW_L2(3); break;
      case 4: GWW_DMA(3, false, false); break;
      case 5: GWW_DMA(4, false, false); break;
      case 6: GWW_DMA(3, true, false); break;
      default: GWW_DMA(3, false, true); break;
    }
#undef GWW_L2
  } else if (nw == 8) {
    hipLaunchKernelGGL(k_attention_bf16<8>, dim3((unsigned)blocks), dim3(512), 0, s, in, out, lse, T, H, q_tiles, qt0);
  } else
#else
  if (q_log2) {   // log2-unit q, 128 query rows per workgroup: the pooled last layer's query tile
    GWW_DMA(3, false, true);
  } else
#endif
  {               // natural-unit q: the kernel-level entry point gww_attention_bf16
    hipLaunchKernelGGL(k_attention_bf16<4>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, lse, T, H, q_tiles, qt0);
  }
#undef GWW_DMA
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

#ifdef GWW_STAMP
extern "C" int gww_debug_stamps_att(unsigned long long* out8, int reset) {
  GWW_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gww::g_stamp_att), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0};
    GWW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gww::g_stamp_att), z, sizeof(z)));
  }
  return GWW_OK;
}
#endif

extern "C" int gww_attention_bf16(const void* qkv, void* ctx, int B, int T, int n_heads, void* stream) {
  return launch_attention_bf16(qkv, ctx, B, T, n_heads, (hipStream_t)stream);
}
extern "C" int gww_attention_lse_bf16(const void* qkv, void* ctx, float* lse, int B, int T, int n_heads, void* stream) {
  GWW_REQUIRE(lse != nullptr, "gww_attention_lse_bf16: lse is NULL");
  return launch_attention_bf16(qkv, ctx, B, T, n_heads, (hipStream_t)stream, lse);
}

extern "C" int gww_attention_log2q_bf16(const void* qkv, void* ctx, float* lse_or_null, int B, int T, int n_heads,
                                        void* stream) {
  return launch_attention_bf16(qkv, ctx, B, T, n_heads, (hipStream_t)stream, lse_or_null, false, true);
}
extern "C" int gww_attention_f32(const float* qkv, float* ctx, int B, int T, int n_heads, void* stream) {
  return launch_attention_f32(qkv, ctx, B, T, n_heads, (hipStream_t)stream);
}
