// k_attention_w64_bf16: the MHSA core (HF:modeling_whisper.py:215-238, 284-356; softmax(q k^T) v, head_dim 64, no mask)
// with 64 query rows per wave and ONE wave per SIMD -- the structure the round-1 .. round-3 notes end on.
//
// Why: k_attention_dma_bf16 (attention.hip; 32 query rows per wave, three waves per SIMD) is bound by instruction issue:
// per 64-key tile a wave issues 18 MFMAs (576 matrix-pipe cycles) next to ~150 other instructions, every K / V fragment
// read, address and wait feeds ONE 32-row query block.  Here a wave owns TWO query blocks and the whole 512-register
// file; every K / V fragment read feeds two MFMAs, and the wave's single instruction stream is a hand-placed software
// pipeline over UNITS of 32 keys (half a tile): step u issues
//     S(u+1) = K(u+1) Q^T    10 MFMAs (the first of each chain is the reference / mask product, below)
//     O     += V(u-1)^T P(u-1)^T     8 MFMAs
// and, in the gaps behind those 18 MFMAs, the softmax of unit u on the vector ALU: 32 v_exp_f32, 32 v_add_f32 (row
// sums), 16 v_cvt_pk_bf16_f32 per lane -- two exponentials, two adds and one convert per gap, none of them reading a
// result of the same or the previous gap -- plus the unit's 12 fragment reads and LDS-DMA requests.  Every instruction
// of the loop body is a volatile asm statement or pinned by a scheduling barrier, so program order IS issue order.
//
// Arithmetic: q in log2 units (p = v_exp_f32(s)), the reference -m enters through the matrix pipe (A = "ones" rows,
// B = [-m_hi, -m_lo] of the query row), denominators by v_add_f32 -- as k_attention_dma_bf16<3, false, true>.  The
// reference is the exact row maximum of the FIRST 32 keys and is never moved inside the loop: there is no row maximum,
// no trigger test and NO cold block in the steady state (k_attention_dma_bf16 re-bases when a tile's row sum exceeds
// e^8; here a join behind such a block made hipcc copy ~60 loop-carried registers per step on the hot side).  p may
// exceed 1 -- fp32 / bf16 keep their relative precision -- and only a row whose denominator leaves the fp32 range is
// wrong; it is detected ONCE, behind the loop (l huge / inf / NaN), and then the wave redoes its 64 rows with the
// textbook online softmax (slow, plain code; tests force it: spike, staircase).  The key mask of the ragged last tile rides in the reference product --
// element 2 of a key's "ones" row is 1 for keys >= T and the reference column carries -1e30 there -- so there is no
// masked variant of the body, and a unit past the end of the sequence (the pipeline computes S one unit ahead) is
// simply fully masked.
//
// K / V tiles ([64 keys][64] bf16, the swizzled images of attention.hip) arrive by LDS-DMA into two rings of four
// slots; K three tiles ahead, V two, one s_waitcnt vmcnt(4) + s_barrier per tile, never vmcnt(0) in the loop.
#include "common.h"

#include <type_traits>

namespace gww {

#ifndef GWW_W64_ABL
#define GWW_W64_ABL 0   // diagnostic builds only (tools/att_w64_exp.py; wrong results by design, only the time matters):
                        // 1 = no v_exp, 2 = no row-sum adds, 4 = no converts, 8 = no fragment reads in the steps, 16 = no
                        // LDS-DMA requests, 32 = no ring wait / barrier per tile, 64 = no S MFMAs, 128 = no O MFMAs
#endif

#ifdef GWW_W64_STAMP
// diagnostic build only (tools/att_w64_exp.py STAMP=1): s_memtime ticks per phase, summed over the waves.  s_memtime is a
// fixed-rate counter on this system: the SHARES are what counts.
__device__ unsigned long long g_stamp_w64[8];
#define W_STAMP_DECL unsigned long long _ts = __builtin_amdgcn_s_memtime(); unsigned long long _ta[7] = {0, 0, 0, 0, 0, 0, 0};
#define W_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long _tn = __builtin_amdgcn_s_memtime(); _ta[i] += _tn - _ts; _ts = _tn; __builtin_amdgcn_sched_barrier(0); } while (0)
#define W_STAMP_FLUSH if (lane == 0) { for (int _q = 0; _q < 7; ++_q) atomicAdd(&g_stamp_w64[_q], _ta[_q]); atomicAdd(&g_stamp_w64[7], 1ull); }
#else
#define W_STAMP_DECL
#define W_STAMP(i)
#define W_STAMP_FLUSH
#endif

namespace {
constexpr int W_DH = 64;
constexpr int W_TILE = 64 * W_DH * 2;         // 8 KB: one K or V tile
constexpr int W_SLOTS = 4;
constexpr int W_VOFF = W_SLOTS * W_TILE;      // V ring behind the K ring
constexpr int W_LDS = 2 * W_SLOTS * W_TILE;   // 64 KB
constexpr float W_LMAX = 1.0e30f;             // a row denominator above it (2^100; or inf / NaN) sends the wave to the exact path

template <int I>
using wic = std::integral_constant<int, I>;

__device__ __forceinline__ int w_koff(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int w_voff(int row, int colbyte) { return row * 128 + (colbyte ^ (((row >> 1) & 1) << 6)); }

// ---- the loop body's instructions (volatile: their order is the program's)
#define W_EXP(D, S) do { if (!(GWW_W64_ABL & 1)) asm volatile("v_exp_f32 %0, %1" : "=v"(D) : "v"(S)); else asm volatile("" : "=v"(D) : "v"(S)); } while (0)
#define W_ADD(ACC, X) do { if (!(GWW_W64_ABL & 2)) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ACC) : "v"(X)); else asm volatile("" : "+v"(ACC) : "v"(X)); } while (0)
#define W_CVT(D, LO, HI) do { if (!(GWW_W64_ABL & 4)) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(D) : "v"(LO), "v"(HI)); else asm volatile("" : "=v"(D) : "v"(LO), "v"(HI)); } while (0)
// S chains: D (architectural registers) = A B (+ D); the first MFMA of a chain takes the literal 0 as C
#define W_MFMA_Z(D, A, B) do { if (!(GWW_W64_ABL & 64)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(D) : "v"(A), "v"(B)); else asm volatile("" : "=&v"(D) : "v"(A), "v"(B)); } while (0)
#define W_MFMA_SQ(D, A, B) do { if (!(GWW_W64_ABL & 64)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(A), "a"(B)); else asm volatile("" : "+v"(D) : "v"(A), "a"(B)); } while (0)
// O and l chains: accumulators in the accumulator file.  Register classes are chosen so that hipcc never copies in front
// of an asm MFMA: what plain code touches INSIDE the loop (S, P, the K / V fragments hipcc loads) lives in architectural
// registers; O, l and the Q fragments, which only asm MFMAs name between the prologue and the epilogue, in the accumulator
// file (while a cold re-base block inside the loop still scaled O in plain code, hipcc kept O in architectural registers
// and copied all 16 into the accumulator file in front of every MFMA)
#define W_MFMA_O(D, A, B) do { if (!(GWW_W64_ABL & 128)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "v"(A), "v"(B)); else asm volatile("" : "+a"(D) : "v"(A), "v"(B)); } while (0)
// outside the loop hipcc may put a register copy directly in front of an asm MFMA: two wait states inside the statement
#define W_MFMA_O_PAD(D, A, B) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "v"(A), "v"(B))
#define W_PIN() __builtin_amdgcn_sched_barrier(0)
}  // namespace

__global__ __launch_bounds__(256, 1) void k_attention_w64_bf16(const unsigned short* __restrict__ qkv,
                                                             unsigned short* __restrict__ ctx,
                                                             float* __restrict__ lse, int T, int H, int q_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[W_LDS];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware work order (attention.hip): the query tiles of one (b, h) -- which stream the same K / V -- sit on one XCD
  const unsigned nblk = gridDim.x, per = nblk >> 3;
  const unsigned wid = blockIdx.x < 8 * per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
  const int qt = wid % q_tiles;
  const int bh = wid / q_tiles;
  const int b = bh / H, h = bh - b * H;
  const int d = H * W_DH;
  const long row_stride = 3L * d;
  const unsigned short* base = qkv + (long)b * T * row_stride;
  const unsigned short* qp = base + h * W_DH;
  const unsigned short* kp = base + d + h * W_DH;
  const unsigned short* vp = base + 2 * d + h * W_DH;
  const int r = lane & 31, hh = lane >> 5;
  const int n_kt = (T + 63) >> 6;
  const bool ragged = (T & 63) != 0;

  // ---- LDS-DMA: piece j (0, 1) of this wave = tile rows 8 (2 wave + j) .. + 7; lane l lands at (row l >> 3, 16-byte
  // position l & 7) of its piece and must fetch the chunk the swizzled reads expect there (source-side swizzle)
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr)lds;
  unsigned koff[2], voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
    koff[j] = (unsigned)(row * (int)row_stride * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
    voff[j] = (unsigned)(row * (int)row_stride * 2 + ((pos ^ (((row >> 1) & 1) << 2)) << 4));
  }
  auto dma_piece = [&](const unsigned short* src, unsigned off, unsigned dst) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(src), "s"(dst) : "memory");
  };
  // one tile of K (which = 0) or V (1) into the ring: src = first row of the tile, dst = LDS byte address of this wave's
  // first piece; rag: the ragged last tile, whose rows past T - 1 read row T - 1 (never past the tensor)
  auto dma_pair = [&](const unsigned short* src, unsigned dst, bool rag, int last_row, int which) __attribute__((always_inline)) {
    if (__builtin_expect(!rag, 1)) {
      dma_piece(src, which ? voff[0] : koff[0], dst);
      dma_piece(src, which ? voff[1] : koff[1], dst + 1024);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * wave + j) + (lane >> 3), pos = lane & 7;
        const int rc = row < last_row ? row : last_row;
        const unsigned sw = which ? (unsigned)((pos ^ (((row >> 1) & 1) << 2)) << 4) : (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
        dma_piece(src, (unsigned)(rc * (int)row_stride * 2) + sw, dst + j * 1024);
      }
    }
  };
  // tiles past the end re-request the last one (never read unmasked)
  const int last_row_rag = T - 1 - (n_kt - 1) * 64;
  auto dma_tile = [&](int kt, int slot, int which) __attribute__((always_inline)) {
    if (kt > n_kt - 1) kt = n_kt - 1;
    const unsigned short* src = (which ? vp : kp) + (long)kt * 64 * row_stride;
    const unsigned dst = lds_base + (unsigned)((which ? W_VOFF : 0) + slot * W_TILE + 2 * wave * 1024);
    dma_pair(src, dst, kt == n_kt - 1 && ragged, last_row_rag, which);
  };

  W_STAMP_DECL
  // prologue requests: K(0), K(1), K(2), V(0), V(1)
  dma_tile(0, 0, 0); dma_tile(0, 0, 1); dma_tile(1, 1, 0); dma_tile(1, 1, 1); dma_tile(2, 2, 0);

  // ---- Q fragments (B operand of K Q^T): Q[q = 32 qb + r][dh = 16 s + 8 hh + j]; rows past T - 1 duplicate row T - 1
  const int q_row0 = qt * 256 + wave * 64 + r;
  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q_row = q_row0 + 32 * qb;
    const int q_ld = q_row < T ? q_row : T - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + (long)q_ld * row_stride + 16 * s + 8 * hh);
  }

  // ---- fragment addresses (attention.hip's images): K rows by ds_read_b128, V^T by ds_read_b64_tr_b16
  const unsigned char* kaddr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) kaddr[s] = lds + w_koff(r, 2 * s + hh);
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_colbyte = (((lane & 31) >> 4) * 16 + 4 * tr_p) * 2;
  const unsigned char* vaddr[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) vaddr[n] = lds + W_VOFF + w_voff(4 * hh + tr_q, 64 * n + tr_colbyte);

  f32x16 st[2][2];       // [unit parity][query block]: S^T (key in the registers, query on the lane)
  bf16x8 kf[4];          // [k-step]: K fragments of the unit being scored; each is re-read (next unit but one... the unit
                         // after it) right behind its second MFMA
  bf16x8 vf[2][2];       // [dh block][k-step of the unit]: V^T fragments, re-read behind their second MFMA likewise
  u32x4 pf[2][2][2];     // [unit parity][query block][k-step of the unit]: bf16 P^T, B operand of V^T P^T
  f32x16 oacc[2][2];     // [dh block][query block]: O^T
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int j = 0; j < 16; ++j) oacc[n][qb][j] = 0.f;
  float m_run[2] = {0.f, 0.f};
  // row sums of the (bf16-rounded) probabilities THROUGH THE MATRIX PIPE: l^T += ones P^T, every register of lsum[qb] ends
  // up holding the denominator of the lane's query row.  The loop is bound by vector-instruction issue (592 cycles per
  // step with v_add_f32 row sums against 576 cycles of MFMAs, tools/att_w64_exp.py); four more MFMAs per step take 32
  // v_add_f32 off that side: 496 issue cycles against 704 matrix cycles.
  f32x16 lsum[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int j = 0; j < 16; ++j) lsum[qb][j] = 0.f;
  u32x4 all_ones;   // (defined by asm: a constant hipcc knows it re-materialises by v_mov directly in front of the asm MFMAs)
  {
    unsigned w_;
    asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(w_));
    all_ones = u32x4{w_, w_, w_, w_};
    asm volatile("" : "+v"(all_ones));
  }
  // reference column of a query block (B operand, k = 0: -m_hi, 1: -m_lo, 2: the mask value; lanes of half 1 hold k >= 8)
  const unsigned mask_word = hh == 0 ? 0x0000F14Au : 0u;   // bf16(-1e30) at k = 2
  u32x4 mref[2] = {u32x4{0u, mask_word, 0u, 0u}, u32x4{0u, mask_word, 0u, 0u}};
  const unsigned ones_w0 = hh == 0 ? 0x3F803F80u : 0u;     // k = 0, 1: 1.0
  // "ones" rows of unit (tile kt, key half kb): element 2 = 1.0 for keys past T - 1
  auto ones_of = [&](int kt, int kb) __attribute__((always_inline)) -> u32x4 {
    const int key = kt * 64 + 32 * kb + r;
    return u32x4{ones_w0, (hh == 0 && key >= T) ? 0x00003F80u : 0u, 0u, 0u};
  };

  auto load_k = [&](int slot, int kb) __attribute__((always_inline)) {   // K fragments of a unit -> kf
#pragma unroll
    for (int s = 0; s < 4; ++s)
      kf[s] = *reinterpret_cast<const bf16x8*>(kaddr[s] + slot * W_TILE + kb * 4096);
  };
  auto load_v1 = [&](const unsigned char* const (&vcur_)[2], int kb, int n, int kl) __attribute__((always_inline)) {   // one V^T fragment of a unit -> vf[n][kl]
    const unsigned char* vb = vcur_[n] + (32 * kb + 16 * kl) * 128;
    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)vb);
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb + 8 * 128));
    bf16x8 v;
    v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
    v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
    vf[n][kl] = v;
  };

  // ---- the first unit of the workgroup: scored against reference 0, its exact row maximum becomes the reference of the
  // whole pass (plain compiler-scheduled code, once per workgroup)
  auto first_reference = [&]() __attribute__((always_inline)) {
    const u32x4 ones = ones_of(0, 0);
    bf16x8 kt_f[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kt_f[s] = *reinterpret_cast<const bf16x8*>(kaddr[s]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 z;
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = 0.f;
      f32x16 s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, mref[qb]), z, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_f[s], qf[qb][s], s_, 0, 0, 0);
      float tmax = s_[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, s_[j]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const __bf16 hi = (__bf16)(-tmax);
      const __bf16 lo = (__bf16)(-tmax - (float)hi);
      const float m_new = -((float)hi + (float)lo);
      m_run[qb] = m_new;
      const unsigned w0 = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
      mref[qb][0] = hh == 0 ? w0 : 0u;
#pragma unroll
      for (int j = 0; j < 16; ++j) s_[j] -= m_new;
      st[0][qb] = s_;   // the pipeline's first step exponentiates it
    }
    asm volatile("s_nop 7" ::: "memory");   // compiler VALU writes (mref, st) -> the asm MFMAs that read them
  };

  // ---- one step of the pipeline, unit u = (tile kt, key half KB), parity PAR = KB:
  //   MFMAs : S(u+1) -> st[1-PAR] with kf (unit u+1 = (kt + KB, 1 - KB)), then O += V(u-1)^T P(u-1)^T
  //   VALU  : softmax of st[PAR] -> pf[PAR], row sums
  //   LDS   : kf <- unit u+2 = (kt + 1, KB);  vf <- unit u
  //   DMA   : KB = 0: K(kt + 3);  KB = 1: V(kt + 2)
  // The ring slot of a tile is kt & 3: kcur / vcur = the fragment addresses inside the slots of K(kt + 1) / V(kt), set
  // once per tile (six VALU additions), so one tile body serves the whole loop.
  const unsigned char* kcur[4];
  const unsigned char* vcur[2];
  const unsigned short* req_src[2];   // [0]: K(kt + 3), [1]: V(kt + 2) -- this tile's two requests
  unsigned req_dst[2];
  bool req_rag[2];
  auto set_slots = [&](int kt) __attribute__((always_inline)) {
    const int ko = ((kt + 1) & 3) * W_TILE, vo = (kt & 3) * W_TILE;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      int t = kt + 3 - w;
      const int slot = t & 3;
      if (t > n_kt - 1) t = n_kt - 1;
      req_src[w] = (w ? vp : kp) + (long)t * 64 * row_stride;
      req_dst[w] = lds_base + (unsigned)((w ? W_VOFF : 0) + slot * W_TILE + 2 * wave * 1024);
      req_rag[w] = ragged && t == n_kt - 1;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) kcur[s] = kaddr[s] + ko;
#pragma unroll
    for (int n = 0; n < 2; ++n) vcur[n] = vaddr[n] + vo;
  };
  auto step = [&](auto kb_c, int kt) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_c)::value, PAR = KB, NXT = 1 - KB;
    u32x4 ones_n = ones_of(kt + KB, NXT);              // "ones" rows of unit u+1
    float E[2][4];
    asm volatile("s_nop 1" : "+v"(ones_n));            // (a VALU result as an MFMA operand: hipcc pads nothing in front of asm)
    W_PIN();
#define W_GAP(G)                                                                                          \
  {                                                                                                       \
    if constexpr ((G) < 16) { W_EXP(E[0][(G) & 3], st[PAR][0][(G) & 15]); }                                 \
    if constexpr ((G) < 16) { W_EXP(E[1][(G) & 3], st[PAR][1][(G) & 15]); }                                 \
    if constexpr ((G) >= 2 && (G) <= 16 && (G) % 2 == 0) {                                                             \
      constexpr int k_ = ((G) - 2) / 2;                                                                   \
      W_CVT(pf[PAR][0][k_ >> 2][k_ & 3], E[0][(2 * k_) & 3], E[0][(2 * k_ + 1) & 3]);                       \
    }                                                                                                     \
    if constexpr ((G) >= 3 && (G) <= 17 && (G) % 2 == 1) {                                                             \
      constexpr int k_ = ((G) - 3) / 2;                                                                   \
      W_CVT(pf[PAR][1][k_ >> 2][k_ & 3], E[1][(2 * k_) & 3], E[1][(2 * k_ + 1) & 3]);                       \
    }                                                                                                     \
    /* fragment reads, each right behind the second (last) MFMA that takes the register's old content: K of unit u+2   \
       behind MFMAs 3 / 5 / 7 / 9, V^T of unit u (two transposed reads each) behind MFMAs 12 / 13 / 16 / 17 */        \
    if constexpr (GWW_W64_ABL & 8) {                                                                      \
    } else if constexpr ((G) == 3 || (G) == 5 || (G) == 7 || (G) == 9) {                                  \
      kf[((G) - 3) / 2] = *reinterpret_cast<const bf16x8*>(kcur[((G) - 3) / 2] + KB * 4096);                \
    } else if constexpr ((G) == 12 || (G) == 13) {                                                        \
      load_v1(vcur, KB, (G) - 12, 0);                                                                     \
    } else if constexpr ((G) == 16 || (G) == 17) {                                                        \
      load_v1(vcur, KB, (G) - 16, 1);                                                                     \
    }                                                                                                     \
    if constexpr ((G) == 19 && !(GWW_W64_ABL & 16)) dma_pair(req_src[KB], req_dst[KB], req_rag[KB], last_row_rag, KB); \
    W_PIN();                                                                                              \
  }
    // S(u+1): reference / mask product, then the four k-steps, the two query blocks interleaved
    W_MFMA_Z(st[NXT][0], ones_n, mref[0]);        W_GAP(0)
    W_MFMA_Z(st[NXT][1], ones_n, mref[1]);        W_GAP(1)
    W_MFMA_SQ(st[NXT][0], kf[0], qf[0][0]);  W_GAP(2)
    W_MFMA_SQ(st[NXT][1], kf[0], qf[1][0]);  W_GAP(3)
    W_MFMA_SQ(st[NXT][0], kf[1], qf[0][1]);  W_GAP(4)
    W_MFMA_SQ(st[NXT][1], kf[1], qf[1][1]);  W_GAP(5)
    W_MFMA_SQ(st[NXT][0], kf[2], qf[0][2]);  W_GAP(6)
    W_MFMA_SQ(st[NXT][1], kf[2], qf[1][2]);  W_GAP(7)
    W_MFMA_SQ(st[NXT][0], kf[3], qf[0][3]);  W_GAP(8)
    W_MFMA_SQ(st[NXT][1], kf[3], qf[1][3]);  W_GAP(9)
    // O += V(u-1)^T P(u-1)^T: the unit's two k-steps x two dh blocks x two query blocks
    { W_MFMA_O(oacc[0][0], vf[0][0], pf[NXT][0][0]); }  W_GAP(10)
    { W_MFMA_O(oacc[1][0], vf[1][0], pf[NXT][0][0]); }  W_GAP(11)
    { W_MFMA_O(oacc[0][1], vf[0][0], pf[NXT][1][0]); }  W_GAP(12)
    { W_MFMA_O(oacc[1][1], vf[1][0], pf[NXT][1][0]); }  W_GAP(13)
    { W_MFMA_O(oacc[0][0], vf[0][1], pf[NXT][0][1]); }  W_GAP(14)
    { W_MFMA_O(oacc[1][0], vf[1][1], pf[NXT][0][1]); }  W_GAP(15)
    { W_MFMA_O(oacc[0][1], vf[0][1], pf[NXT][1][1]); }  W_GAP(16)
    { W_MFMA_O(oacc[1][1], vf[1][1], pf[NXT][1][1]); }  W_GAP(17)
    // l^T += ones P(u-1)^T
    W_MFMA_O(lsum[0], all_ones, pf[NXT][0][0]);  W_GAP(18)
    W_MFMA_O(lsum[1], all_ones, pf[NXT][1][0]);  W_GAP(19)
    W_MFMA_O(lsum[0], all_ones, pf[NXT][0][1]);  W_GAP(20)
    W_MFMA_O(lsum[1], all_ones, pf[NXT][1][1]);  W_GAP(21)
#undef W_GAP
  };

  // ---- prologue: every request landed, unit 0 scored against reference 0 and re-based exactly
  W_STAMP(0);   // requests, Q loads, address set-up
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  W_STAMP(1);   // first tiles' wait + barrier
  first_reference();
  load_k(0, 1);   // unit 1
  asm volatile("s_nop 7" ::: "memory");

  // (the first step has no unit u-1: its O product runs on zero operands, so the loop has ONE body and every register
  // keeps one home -- a peeled first tile made hipcc re-home accumulators / Q fragments with copies directly in front of
  // asm MFMAs, which it cannot pad)
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int kl = 0; kl < 2; ++kl) {
      pf[1][n][kl] = u32x4{0u, 0u, 0u, 0u};
      vf[n][kl] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
    }
  W_STAMP(2);   // first unit's reference
#pragma clang loop unroll(disable)   // (no peeled first iteration: a second copy of the body re-homes registers with copies in front of asm MFMAs)
  for (int kt = 0; kt < n_kt; ++kt) {
    if (kt > 0 && !(GWW_W64_ABL & 32)) {   // K(kt + 1), V(kt) landed (the four pieces requested during tile kt - 1 may still fly)
      if (GWW_W64_ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    W_STAMP(3);   // ring wait + barrier per tile
    set_slots(kt);
    step(wic<0>{}, kt);
    step(wic<1>{}, kt);
    W_STAMP(4);   // the tile's two steps
  }
  // the last unit's O product (its P in pf[1], its V^T fragments in vf[1])
  W_MFMA_O_PAD(oacc[0][0], vf[0][0], pf[1][0][0]);
  W_MFMA_O_PAD(oacc[1][0], vf[1][0], pf[1][0][0]);
  W_MFMA_O_PAD(oacc[0][1], vf[0][0], pf[1][1][0]);
  W_MFMA_O_PAD(oacc[1][1], vf[1][0], pf[1][1][0]);
  W_MFMA_O_PAD(oacc[0][0], vf[0][1], pf[1][0][1]);
  W_MFMA_O_PAD(oacc[1][0], vf[1][1], pf[1][0][1]);
  W_MFMA_O_PAD(oacc[0][1], vf[0][1], pf[1][1][1]);
  W_MFMA_O_PAD(oacc[1][1], vf[1][1], pf[1][1][1]);
  W_MFMA_O_PAD(lsum[0], all_ones, pf[1][0][0]);
  W_MFMA_O_PAD(lsum[1], all_ones, pf[1][1][0]);
  W_MFMA_O_PAD(lsum[0], all_ones, pf[1][0][1]);
  W_MFMA_O_PAD(lsum[1], all_ones, pf[1][1][1]);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // no LDS-DMA in flight at the end; MFMA -> readers

  // ---- the deferred overflow check.  The pass above never moves its reference (the exact row maximum of the FIRST 32
  // keys): a later score may exceed it, p = 2^(s - m) > 1 is as accurate in fp32 / bf16 as p <= 1, and only a row whose
  // sum leaves the fp32 range (a score more than ~100 log2 units above every one of the first 32 keys) is wrong.  Such a
  // row shows in its denominator: l is then huge, inf or NaN.  If any row of the wave is affected, the wave redoes its
  // 64 rows with the textbook online softmax (exact running maximum per 64-key tile): plain compiler-scheduled code on
  // a wave-private LDS region, K / V by ordinary loads -- slow, and essentially never taken on real activations.
  float l_tot[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    l_tot[qb] = lsum[qb][0];   // (the product summed over both lane halves' keys)
  }
  W_STAMP(5);   // last unit's products, drain
  __builtin_amdgcn_s_barrier();   // every wave has left the rings: the slow path below owns 16 KB of them per wave
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(l_tot[0] <= W_LMAX) || !(l_tot[1] <= W_LMAX)) != 0, 0)) {
    unsigned char* kw = lds + wave * (2 * W_TILE);   // wave-private K tile | V tile
    unsigned char* vw = kw + W_TILE;
    float mx[2] = {-3.0e38f, -3.0e38f};
    l_tot[0] = 0.f; l_tot[1] = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[n][qb][j] = 0.f;
    for (int kt = 0; kt < n_kt; ++kt) {
      // stage the tile: 512 16-byte chunks of K and of V, 8 of each per lane (rows past T - 1 duplicate row T - 1)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
        int key = kt * 64 + row;
        if (key >= T) key = T - 1;
        const u32x4 kv = *reinterpret_cast<const u32x4*>(kp + (long)key * row_stride + ch * 8);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(vp + (long)key * row_stride + ch * 8);
        *reinterpret_cast<u32x4*>(kw + w_koff(row, ch)) = kv;
        *reinterpret_cast<u32x4*>(vw + w_voff(row, ch * 16)) = vv;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8 kt_f[4], vt_f[2][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) kt_f[s] = *reinterpret_cast<const bf16x8*>(kw + w_koff(32 * kb + r, 2 * s + hh));
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int kl = 0; kl < 2; ++kl) {
            const unsigned char* vb = vw + w_voff(4 * hh + tr_q, 64 * n + tr_colbyte) + (32 * kb + 16 * kl) * 128;
            const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)vb);
            const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb + 8 * 128));
            bf16x8 v;
            v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
            v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
            vt_f[n][kl] = v;
          }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          f32x16 s_;
#pragma unroll
          for (int j = 0; j < 16; ++j) s_[j] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_f[s], qf[qb][s], s_, 0, 0, 0);
          float tmax = -3.0e38f;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int key = kt * 64 + 32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh;
            if (key >= T) s_[j] = -3.0e38f;
            tmax = fmaxf(tmax, s_[j]);
          }
          {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          }
          const float m_new = fmaxf(mx[qb], tmax);
          const float alpha = __builtin_amdgcn_exp2f(mx[qb] - m_new);   // (first tile: 2^-huge = 0 on O = l = 0)
          mx[qb] = m_new;
          float ps = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            s_[j] = __builtin_amdgcn_exp2f(s_[j] - m_new);
            ps += s_[j];
          }
          {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps), __float_as_uint(ps), false, false);
            ps = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
          }
          l_tot[qb] = l_tot[qb] * alpha + ps;
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) oacc[n][qb][j] *= alpha;
#pragma unroll
          for (int kl = 0; kl < 2; ++kl) {
            const u32x4 pw = {pack2bf(s_[8 * kl], s_[8 * kl + 1]), pack2bf(s_[8 * kl + 2], s_[8 * kl + 3]),
                              pack2bf(s_[8 * kl + 4], s_[8 * kl + 5]), pack2bf(s_[8 * kl + 6], s_[8 * kl + 7])};
#pragma unroll
            for (int n = 0; n < 2; ++n)
              oacc[n][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt_f[n][kl], __builtin_bit_cast(bf16x8, pw), oacc[n][qb], 0, 0, 0);
          }
        }
      }
    }
    m_run[0] = mx[0];
    m_run[1] = mx[1];
  }

  // ---- epilogue: 1 / l, context rows in 16-byte pieces (attention.hip's exchange of neighbouring column groups)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q_row = q_row0 + 32 * qb;
    const float inv = 1.0f / l_tot[qb];
    if (lse && q_row < T && hh == 0)
      lse[((long)b * H + h) * T + q_row] = (m_run[qb] + __log2f(l_tot[qb])) * 0.69314718055994530942f;
    unsigned short* orow = ctx + ((long)b * T + q_row) * d + h * W_DH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const int c0 = 2 * cp, c1 = 2 * cp + 1;
        const f32x16& o = oacc[n][qb];
        const unsigned a0 = pack2bf(o[4 * c0] * inv, o[4 * c0 + 1] * inv), a1 = pack2bf(o[4 * c0 + 2] * inv, o[4 * c0 + 3] * inv);
        const unsigned b0 = pack2bf(o[4 * c1] * inv, o[4 * c1 + 1] * inv), b1 = pack2bf(o[4 * c1 + 2] * inv, o[4 * c1 + 3] * inv);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 ov = {s0[0], s1[0], s0[1], s1[1]};
        if (q_row < T) *reinterpret_cast<u32x4*>(orow + 32 * n + 8 * (hh ? c1 : c0)) = ov;
      }
  }
  W_STAMP(6);   // barrier, overflow check, epilogue stores (issue)
  W_STAMP_FLUSH
}

// q must be in log2 units (every bf16 q panel of the encoder is packed that way).  All query tiles of the launch.
int launch_attention_w64_bf16(const void* qkv, void* ctx, int B, int T, int H, hipStream_t s, float* lse) {
  const int q_tiles = (T + 255) / 256;
  const long blocks = (long)q_tiles * B * H;
  GWW_REQUIRE(blocks < 2147483647L, "attention_w64: grid too large");
  GWW_REQUIRE(64L * 3 * H * W_DH * 2 < (1L << 31), "attention_w64: row stride too large");
  hipLaunchKernelGGL(k_attention_w64_bf16, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)qkv,
                     (unsigned short*)ctx, lse, T, H, q_tiles);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

#ifdef GWW_W64_STAMP
extern "C" int gww_debug_stamps_w64(unsigned long long* out8, int reset) {
  GWW_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gww::g_stamp_w64), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0};
    GWW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gww::g_stamp_w64), z, sizeof(z)));
  }
  return GWW_OK;
}
#endif
