// Fused MLP block of one encoder layer at d_model = 384 (whisper-tiny), bf16 MFMA:
//
//     x_new = x + delta                      (deferred residual of out_proj, written back, fp32)
//     out   = fc2( gelu( fc1( LayerNorm(x_new) ) ) ) + b2          (bf16 delta for the next layer)
//
// HF:modeling_whisper.py:401-407 (final_layer_norm -> fc1 -> activation_fn -> fc2; the residual add of
// :407 is deferred to the consumer exactly like the A-stationary GEMMs do, gemm_astat.hip).
//
// Why: as two kernels (LN+fc1+GELU, then fc2) the [M, ffn] activation makes a 2.4 GB HBM round trip per
// layer at B = 256 and the GELU epilogue (VALU) cannot overlap any MFMA.  Here the ffn activation never
// leaves the CU and most of the GELU work runs beside the fc2 MFMAs:
//
//   * one workgroup = 128 rows = 4 waves x 32 rows, ONE wave per SIMD with the whole 512-register file:
//     A operand (LayerNorm'd x, whole K = 384) 96 VGPRs, fc1 chunk accumulator [32 x 128] 64, output
//     accumulator [32 x 384] 192.
//   * the ffn dimension is walked in chunks of 128:  S = a W1'[chunk]^T (96 MFMAs), LayerNorm algebra +
//     bias + GELU on S in registers, and the bf16 result IS the B operand of the second product
//     O += P W2[:, chunk]^T (96 MFMAs) -- the accumulator-as-operand trick of attention.hip; W2 is packed
//     with bits 2 and 3 of k swapped inside every 16-group so that its fragments are plain 16-byte reads.
//   * W1' / W2 stream through ONE ring of [128 rows][64 k] bf16 tiles (16 KB, XOR-swizzled, LDS-DMA with
//     source-side swizzle, counted vmcnt, one raw s_barrier per tile): 12 tiles per chunk, 16 MFMAs per
//     wave per tile.  Waits run one tile ahead so the first fragments of the next tile are read before
//     its barrier.
//   * prologue / epilogue as in gemm_astat.hip: whole-line loads of x / delta, wave-private LDS transposes,
//     single-pass algebraic LayerNorm, whole-line stores of the bf16 output.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace gww {

namespace {
constexpr int MF_D = 384;                  // d_model this kernel is built for
constexpr int MF_KT = MF_D / 64;           // 6 k-tiles of fc1
constexpr int MF_OT = MF_D / 32;           // 12 output sub-tiles of fc2
constexpr int MF_WAVES = 4, MF_THREADS = 256, MF_BM = 128;
#ifndef GWW_MF_EXP
#define GWW_MF_EXP 0   // diagnostic builds only: 1 = no DMA in the loop, 2 = no GELU, 4 = no fragment reads in the loop, 8 = stream folded onto its first 8 tiles, 16 = GELU bias not read from LDS, 32 = GELU transcendentals replaced by multiplies, 64 = no ring wait / barrier per tile
#endif
#ifndef GWW_MF_ABL
#define GWW_MF_ABL 0   // diagnostic builds only (wrong results by design, only the time matters): 1 = q/k/v tail without its global
                       // stores, 2 = without its n-tile epilogues, 4 = without DMA issue / ring waits, 8 = without barriers;
                       // 16 = seams without their stores, 32 = without their loads; 64 = OP prologue without the ctx loads;
                       // 128 = out_proj GEMM without DMA issue / ring waits
#endif
#ifndef GWW_MF_OLDDMA
#define GWW_MF_OLDDMA 0   // diagnostic: round 1's eight-instruction DMA issue
#endif
#ifndef GWW_MF_AHEAD
#define GWW_MF_AHEAD 3    // tiles in flight; 3 makes the ring four stages deep and every stage a compile-time constant
#endif
#ifndef GWW_MF_SCHED
#define GWW_MF_SCHED 1  // 1 (shipped): the GELU is placed by hand into the MFMA gaps, four values per gap (gelu_slice), every
                        // main-loop MFMA is a volatile asm statement, the fc1 accumulators live in architectural registers
                        // and start from the folded bias, W1' / W2 are packed x 1/8 / x 8.  0: round 1's form (builtin MFMAs,
                        // scheduler hints that hipcc turns into one ~30-instruction lump per four MFMAs): 64.6 against 42.6
                        // cycles per MFMA in the main loop (profiles/r02_mlp_phase_stamps.md).
#endif
#ifndef GWW_MF_VACC
#define GWW_MF_VACC 1   // (with SCHED) fc1 accumulators in architectural registers, MFMAs as asm
#endif
#ifndef GWW_MF_NTSTORE
#define GWW_MF_NTSTORE 1   // q / k / v tail: non-temporal output stores (they retire sooner: the ring wait behind them is shorter, -2 % on the launch)
#endif
#ifndef GWW_MF_NTLOAD
#define GWW_MF_NTLOAD 1   // streaming panel loads (x, delta / ctx, x_new) with the non-temporal hint
#endif
#if GWW_MF_NTLOAD
#define MF_NT " nt"
#else
#define MF_NT ""
#endif
#ifndef GWW_MF_PADNOP
#define GWW_MF_PADNOP 0   // 1: two wait states in front of the asm MFMAs of the out_proj GEMM and the q / k / v tail (hipcc may place
                          // a register copy directly in front of an MFMA it cannot see inside an asm statement) -- rounds 2 / 3.
                          // 0 (round 4): without them; build() runs tools/audit_asm_mfma.py over the ISA of every instantiation
                          // and fails on the first MFMA with an operand written too close in front of it.  1 152 s_nop per wave:
                          // 1.430 / 1.428 -> 1.407 / 1.414 ms for <1, true> (tools/mlp_exp2.py PADNOP=1 PADNOP=0, twice each)
#endif
#if GWW_MF_PADNOP
#define MF_PADNOP "s_nop 1\n\t"
#else
#define MF_PADNOP ""
#endif
#ifndef GWW_MF_XLDS
#define GWW_MF_XLDS 0   // 1: XACC's x_next / x_new / y leave through the wave-private LDS slices in ROW order (whole 128-byte lines, 8
                        // rows per store instruction) instead of from the accumulator layout (32-byte row pieces per lane pair).
                        // Measured 1.449 against 1.438 ms: the 0.08 ms the 48 stores of a panel cost (ABL=272) is not their
                        // request count -- it is the in-order vmcnt queue again (the third tail tile's ring wait needs them retired)
#endif
#ifndef GWW_MF_XSTNT
#define GWW_MF_XSTNT 0   // XACC: x_next / y stores with the non-temporal hint
#endif
#ifndef GWW_MF_XNT
#define GWW_MF_XNT 0   // 1: the XACC x request with the non-temporal hint too.  Measured 1.490 against 1.414 ms: a lane pair reads
                       // 32 bytes of a row per instruction and the four instructions of a 32-column block share each row's
                       // 128-byte line -- without the hint the line stays in the vector L1 for the other three
#endif
#if GWW_MF_XNT
#define MF_XNT MF_NT
#else
#define MF_XNT ""
#endif
#ifndef GWW_MF_DMAGAP
#define GWW_MF_DMAGAP 1   // (with SCHED) DMA pieces of a riding tile in the gaps the GELU schedule leaves empty (0: one per step)
#endif
#ifndef GWW_MF_TAIL2
#define GWW_MF_TAIL2 0   // 1 (round 4 experiment, tools/mlp_exp2.py TAIL2=1): the q / k / v tail as 64-column CHUNKS of three
                         // fc1-format tiles, software-pipelined like the main loop -- the pack / LDS transpose / stores of chunk
                         // c - 1 and the bias preload of chunk c + 1 ride in the MFMA gaps of chunk c.  Correct (the fp64 tests
                         // pass on it) and NOT faster: 1.565 against 1.556 ms (profiles/r04_mlp_tail2.md) -- the epilogues it
                         // hides were already overlapping the tail's real bound, the in-order vmcnt queue behind its output
                         // stores -- and six more spilled registers.  0 (shipped): round 3's tail, n-tiles of 128 columns.
#endif
#ifndef GWW_MF_NORM
#define GWW_MF_NORM 1   // 1: the A operand is normalised once per panel, a^ = bf16((a - mean') rstd), so the fc1 / q,k,v outputs need
                        // only + cb (0: round 1's per-value LayerNorm algebra rstd (acc - mean' u) + cb: two more VALU
                        // instructions and an LDS read of u per activation value -- the GELU path is what bounds the main loop)
#endif
#ifndef GWW_MF_XACC
#define GWW_MF_XACC 1   // (OP, MODE 1 / 3) round 4: the residual stream lives in the OUTPUT ACCUMULATORS for the whole block.  x is
                        // requested straight into the 192 accumulator registers at kernel start (accumulator layout: a lane owns
                        // half of ITS row), the out_proj GEMM accumulates on top (x_new = x + ctx W_o^T in fp32: the delta is no
                        // longer rounded to bf16), fc2 accumulates on top of that (x_next), and both seams become register work:
                        // row statistics in-lane + one cross-half add, the bf16 A operand by v_permlane32_swap (lane r and lane
                        // r + 32 exchange their halves of a 32-column block: no LDS transpose), x_next / y stored from the
                        // accumulator layout.  Gone: the x load of seam 1 in row order, the x_new store, its re-read in seam 2
                        // (1.2 GB of the launch's 4.0 GB at B = 256 and two exposed HBM round trips per panel), 96 LDS
                        // transposes per seam.  x_new is still written when the caller keeps it (training: x_mid).
#endif
constexpr int MF_AHEAD = GWW_MF_AHEAD;   // tiles in flight ahead of the one being computed
// Ring: AHEAD + 1 slots of one 16-KiB tile, one s_barrier per tile.  (One barrier per two tiles with one more slot was
// built and measured in round 2: 1.154 vs 1.159 ms plain, 1.653 vs 1.653 ms with q/k/v -- no gain, removed.)
constexpr int MF_NST_MAX = MF_AHEAD + 1;
constexpr int MF_TILE = 128 * 64 * 2;      // 16 KB
constexpr int MF_GL = 16 / MF_WAVES;       // LDS-DMA pieces per thread per tile
constexpr int MF_FMAX = 1536;              // largest ffn (cb / u staged in LDS)
constexpr int MF_SLICE_STRIDE = 144, MF_SLICE_BYTES = 32 * MF_SLICE_STRIDE;
constexpr int MF_OFF_CB = MF_NST_MAX * MF_TILE;
constexpr int MF_OFF_U = MF_OFF_CB + MF_FMAX * 4;
constexpr int MF_OFF_B2 = MF_OFF_U + MF_FMAX * 4;
constexpr int MF_OFF_BO = MF_OFF_B2 + MF_D * 4;     // out_proj bias (OP mode)
constexpr int MF_OFF_QCB = MF_OFF_BO + MF_D * 4;    // folded bias of the appended q / k / v panel (staged at kernel start)
constexpr int MF_OFF_SLICE = MF_OFF_QCB + MF_FMAX * 4;
constexpr int MF_LDS = MF_OFF_SLICE + 2 * MF_WAVES * MF_SLICE_BYTES;   // two slices per wave: the seams alternate (below)
static_assert(MF_LDS <= 160 * 1024, "LDS budget");

template <int N>
using MF_FL = std::integral_constant<int, N>;   // flat index of a tile in the unrolled stream (its ring stage, mod 4)

// sum over the 8 lanes of a row group (lane & 7): three DPP adds -- quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
// (lane i <-> 7 - i of its 8: the other quad).  __shfl_xor lowers to ds_bpermute_b32 here: an LDS round trip per step
// in a dependent chain, and a lane-index register per step (three of the kernel's spills).
__device__ __forceinline__ float mf_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <int N>
__device__ __forceinline__ void mf_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// x * sigmoid(p(x)),  p an odd quintic fitted to the erf GELU: |err| <= 2.6e-5 on the whole real line
// (x^2 clamped at 64 keeps p monotone); 7 plain VALU ops + v_exp_f32 + v_rcp_f32.
__device__ __forceinline__ float gelu_sig(float x) {
  const float s = fminf(x * x, 64.0f);
  float q = fmaf(s, 0.0010148164f, -0.1067791331f);     // -log2(e) * (a5 s + a3)
  q = fmaf(s, q, -2.3011178f);                          // -log2(e) * a1
#if GWW_MF_EXP & 32   // diagnostic: the two transcendentals replaced by plain multiplies (wrong values, same count)
  const float e = (x * q) * 0.37f;
  return x * ((1.0f + e) * 0.91f);
#else
  const float e = __builtin_amdgcn_exp2f(x * q);        // exp(-p(x))
  return x * __builtin_amdgcn_rcpf(1.0f + e);
#endif
}

// ---- optional in-kernel phase stamps (diagnostic build only: -DGWW_STAMP), cycles per phase
#ifdef GWW_STAMP
__device__ unsigned long long g_stamp_mlp[24];
// GWW_STAMP=2: only the prologue / main loop / epilogue boundaries (stamps 0, 4, 3) -- no sched_barrier inside the main
// loop, the kernel runs as the production build does.  Entries 21 / 22: the wave's life in s_memrealtime (100 MHz) and
// s_memtime (shader clock) ticks: their ratio is the clock the chip actually held under this kernel.
#define MSTAMP_DECL unsigned long long _t0 = __builtin_amdgcn_s_memtime(); unsigned long long _acc[20] = {0}; \
  const unsigned long long _tb = _t0, _rb = __builtin_amdgcn_s_memrealtime();
#define MSTAMP(i)                                                  \
  do {                                                             \
    if (GWW_STAMP >= 2 && (i) != 0 && (i) != 3 && (i) != 4 && (i) != 5 && !((i) >= 100)) break; \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long _t1 = __builtin_amdgcn_s_memtime();   \
    _acc[(i) >= 100 ? (i) - 100 : (i)] += _t1 - _t0;               \
    _t0 = _t1;                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#define TSTAMP(i) do { if (GWW_STAMP == 3) MSTAMP(100 + (i)); } while (0)   /* q/k/v tail detail (mode 3) */
#define MSTAMP_FLUSH                                                                 \
  if (lane == 0) {                                                                   \
    for (int _q = 0; _q < 20; ++_q) atomicAdd(&g_stamp_mlp[_q], _acc[_q]);           \
    atomicAdd(&g_stamp_mlp[21], __builtin_amdgcn_s_memrealtime() - _rb);             \
    atomicAdd(&g_stamp_mlp[22], __builtin_amdgcn_s_memtime() - _tb);                 \
    atomicAdd(&g_stamp_mlp[23], 1ull);                                                \
  }
#else
#define MSTAMP_DECL
#define MSTAMP(i)
#define TSTAMP(i)
#define MSTAMP_FLUSH
#endif
}  // namespace

// QKV = true appends the NEXT layer's  LayerNorm1 + q / k / v projection  to the block: the epilogue turns into the
// prologue of a second A-stationary GEMM (x_next = x_new + out is formed, written back over x_new and normalised
// while the output is still in registers; the 192 output-accumulator registers are free by then), whose weight
// tiles simply continue the same stream.  The standalone LN+QKV kernel's 10 B/element HBM round trip of the
// residual stream disappears.
// MODE 0: the MLP block.  MODE 1: + the next layer's LN1 + q / k / v (QKV).  MODE 2 (LNQ): ONLY LayerNorm + q / k / v of a
// residual stream that has no pending delta (layer 0, fed by the conv stem): the prologue without delta / store, then the
// q / k / v tail on a stream that holds just those 54 tiles -- the panel prologue and the tail are the same code the
// block kernel runs, instead of the LN-fused A-stationary GEMM (which hipcc spills).
// OP: the attention output projection is fused IN FRONT of the block: `delta` is then the attention context ctx (bf16
// [M, 384]) and the stream starts with the 18 tiles of W_o; ctx is the A operand of a GEMM into the (still idle) output
// accumulators, whose result + bo meets the residual stream in the seam code (x_new = x + bf16(ctx W_o^T + bo)).  The
// stand-alone out_proj kernel and the bf16 delta round trip through HBM disappear.
template <int MODE, bool OP>
__global__ __launch_bounds__(MF_THREADS, 1) void k_mlp_fused(const float* X, const unsigned short* delta, float* x_out,
                                                            const float* __restrict__ ln_u,
                                                            const float* __restrict__ ln_cb,
                                                            const unsigned short* __restrict__ Wt,
                                                            const float* __restrict__ b2,
                                                            unsigned short* __restrict__ C, long M, int F,
                                                            int stagger_ticks, const float* __restrict__ q_u,
                                                            const float* __restrict__ q_cb,
                                                            unsigned short* __restrict__ q_out, int NQ,
                                                            float* x_next, const float* __restrict__ bo, int keep_x_new) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[MF_LDS];
  constexpr bool QKV = MODE == 1 || MODE == 2, LNQ = MODE == 2, FIN = MODE == 3;
  constexpr bool XACC = GWW_MF_XACC && GWW_MF_NORM && GWW_MF_SCHED && OP && (MODE == 1 || MODE == 3);
  constexpr int MF_NST = MF_AHEAD + 1;
  float* lds_cb = reinterpret_cast<float*>(lds + MF_OFF_CB);
  float* lds_u = reinterpret_cast<float*>(lds + MF_OFF_U);
  float* lds_b2 = reinterpret_cast<float*>(lds + MF_OFF_B2);
  float* lds_bo = reinterpret_cast<float*>(lds + MF_OFF_BO);
  float* lds_qcb = reinterpret_cast<float*>(lds + MF_OFF_QCB);
  constexpr int OP_TILES = OP ? 3 * MF_KT : 0;   // W_o tiles in front of the stream
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const long m_base = (long)blockIdx.x * MF_BM + wave * 32;
  unsigned char* slice = lds + MF_OFF_SLICE + wave * MF_SLICE_BYTES;
  unsigned char* const slice0 = slice;
  const int crow = lane >> 3, cchunk = lane & 7;

  // ---- ring: the weights arrive pre-tiled (gww_mlp_pack_bf16): tile (c, idx) is 16 contiguous KiB that
  // already hold the swizzled LDS image, in the order the loop consumes them, so the whole weight stream
  // is one linear walk; piece j of this wave = 1 KiB = one wave-instruction
  // SGPR base + per-lane 32-bit byte offset (no VALU address arithmetic per piece: a VALU write into a
  // register an in-flight MFMA still reads stalls the wave); M0 = wave-uniform LDS destination.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane_off = (unsigned)lane * 16u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr)lds;
  // Piece j of a tile: the wave's four pieces are 4 contiguous KiB in the stream AND in the ring, so one SGPR address
  // pair and one M0 value serve all four and j only enters as the instruction's immediate offset (which the hardware
  // adds to both the global and the LDS address): three instructions per piece (round 1 recomputed source, destination
  // and saved / restored M0 per piece: eight).  Nothing else in this kernel reads M0.
  auto issue_piece_c = [&](int tile, int stage, auto j_c) {
    constexpr int J = decltype(j_c)::value;
    const unsigned dst = lds_base + (unsigned)(stage * MF_TILE + MF_GL * wave_u * 1024);
    if (GWW_MF_EXP & 8) tile &= 7;   // diagnostic: the whole stream collapses onto 128 KB (always L2-hot)
    const unsigned short* src = Wt + ((long)tile * (MF_TILE / 2) + MF_GL * wave_u * 512);
    const unsigned lo = lane_off;   // (a generic lambda's asm operands cannot name captures directly)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :
                 : "v"(lo), "s"(src), "s"(dst), "n"(J * 1024)
                 : "memory");
  };
  auto issue_piece = [&](int tile, int stage, int j) {   // j is a constant after unrolling; the asm operand must be one before
#if GWW_MF_OLDDMA
    const unsigned dst = lds_base + (unsigned)(stage * MF_TILE + (MF_GL * wave_u + j) * 1024);
    const unsigned short* src = Wt + ((long)tile * (MF_TILE / 2) + (MF_GL * wave_u + j) * 512);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(dst) : "memory");
#else
    if (j == 0) issue_piece_c(tile, stage, std::integral_constant<int, 0>{});
    else if (j == 1) issue_piece_c(tile, stage, std::integral_constant<int, 1>{});
    else if (j == 2) issue_piece_c(tile, stage, std::integral_constant<int, 2>{});
    else issue_piece_c(tile, stage, std::integral_constant<int, 3>{});
#endif
  };
  auto issue = [&](int tile, int stage) {
#pragma unroll
    for (int j = 0; j < MF_GL; ++j) issue_piece(tile, stage, j);
  };

  MSTAMP_DECL
  // the bias tables of the block -> LDS.  XACC: called BEHIND the x / ctx requests (below), so that their round trip -- a
  // microsecond of L2 latency, plain loads hipcc waits for one by one -- runs under the panel's HBM round trip instead of in
  // front of it; the first reader (the seam behind the out_proj GEMM) sits behind eighteen tile barriers
  auto stage_tables = [&]() {
    for (int i = tid; i < F; i += MF_THREADS) {
      lds_cb[i] = GWW_MF_SCHED ? ln_cb[i] * 0.125f : ln_cb[i];   // SCHED: the fc1 accumulators hold S / 8 (gelu_slice)
      // (u is only read by round 1's per-value LayerNorm algebra; with NORM the table stays unwritten here, so the final-LN
      // staging below -- MODE 3 keeps its gain / bias in the same place -- has no second writer in another wave)
      if (!GWW_MF_NORM) lds_u[i] = ln_u[i];
    }
    if (!GWW_MF_NORM && FIN) __syncthreads();
    if (!LNQ)
      for (int i = tid; i < MF_D; i += MF_THREADS) lds_b2[i] = XACC ? b2[i] + bo[i] : b2[i];   // XACC: O holds x + ctx W_o^T + fc2, both biases are added when it is read
    if (OP)
      for (int i = tid; i < MF_D; i += MF_THREADS) lds_bo[i] = bo[i];
    if (FIN)   // gain / bias of the encoder's final LayerNorm (q_u / q_cb carry them in this mode), in the unused u table
      for (int i = tid; i < MF_D; i += MF_THREADS) { lds_u[i] = q_u[i]; lds_u[MF_D + i] = q_cb[i]; }
    if (QKV && GWW_MF_NORM)   // (a load issued behind the second seam's stores would wait for them: staged here, 6 KB of LDS)
      for (int i = tid; i < NQ; i += MF_THREADS) lds_qcb[i] = q_cb[i];
  };
  if constexpr (!XACC) stage_tables();
  // De-phase the first round of workgroups (later ones inherit the offsets as CUs free up): panels take
  // the same time everywhere, so without this every CU is in its HBM phase (prologue / epilogue) at the
  // same moment and idles HBM during the MFMA phase.
  if (stagger_ticks > 0 && blockIdx.x < 256) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 15) * stagger_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }

  // The ring is filled FIRST: its AHEAD tiles land while the prologue runs (they used to be requested only after the
  // prologue's last store had retired, and their L2 latency was exposed in front of the first tile).  Loads, stores
  // and LDS-DMA retire in issue order, so the prologue's counted waits are unaffected by these OLDER operations, and
  // the first ring wait (everything but the youngest AHEAD - 1 tiles... of a queue whose youngest entries are then the
  // prologue's stores) at worst waits for more than it needs.
#pragma unroll
  for (int p = 0; p < MF_AHEAD; ++p) issue(p, p);
  // accumulator layout of O: lane (r, hh) owns row m_base + r, columns 32 t + 8 cc + 4 hh .. + 3 in O[t][4 cc ..]
  const long xrow_l = m_base + r < M ? m_base + r : M - 1;
  const unsigned xoff = (unsigned)xrow_l * (unsigned)(MF_D * 4) + 16u * (unsigned)hh;
  f32x4 xp[XACC ? MF_OT : 1][4];
  if constexpr (XACC) {
    // The residual stream goes straight INTO the accumulator file (48 loads of 16 bytes per lane, 32-byte row pieces
    // per lane pair), in front of the ctx loads: loads retire in issue order, so the first counted ctx wait below also
    // proves every x piece has landed -- long before the out_proj GEMM's first MFMA reads them as its C operand.  The four
    // pieces of an accumulator tile are joined into O[t] BEHIND that wait (a register-sequence rename when hipcc's coalescer
    // cooperates, sixteen v_accvgpr_mov otherwise -- either way on landed data; tools/audit_asm_preload.py, build()).
#pragma unroll
    for (int t = 0; t < MF_OT; ++t)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if (GWW_MF_ABL & 32) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(xp[t][cc][0]));   // (diagnostic: no x loads)
        else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" MF_XNT : "=a"(xp[t][cc]) : "v"(xoff), "s"(X), "n"((32 * t + 8 * cc) * 4) : "memory");
  }
  // ---- prologue: x_new = x + delta (written back), a = bf16(x_new - c), exact fp32 row statistics
  bf16x8 af[MF_KT * 4];
  float row_rstd, row_mean;
  long grow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long g = m_base + 8 * i + crow;
    grow[i] = g < M ? g : M - 1;
  }
  if constexpr (XACC) {
    // ---- XACC prologue: the A fragments of the out_proj GEMM come STRAIGHT from global memory -- af[4 S + j] of lane (r, hh)
    // is ctx[row r][64 S + 32 hh + 8 j .. + 7], 16 contiguous bytes; the four requests j = 0 .. 3 of a k-tile share each row's
    // 128-byte line through the vector L1 (no nt hint), as the x requests above do.  No staging registers, no LDS transpose
    // (24 ds_write_b128 + 24 ds_read_b128 per lane and their hand-counted waits in the row-order form below).
    const unsigned coff = (unsigned)xrow_l * (unsigned)(MF_D * 2) + 64u * (unsigned)hh;
#pragma unroll
    for (int S = 0; S < MF_KT; ++S)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(af[4 * S + j]) : "v"(coff), "s"(delta), "n"((64 * S + 8 * j) * 2) : "memory");
    stage_tables();
    // one wait for the panel: ring tiles, x, ctx and the tables (loads retire in issue order; hipcc's own waits for the table
    // values already imply it -- this one is for the reader).  The fences keep the fragments' first use behind it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MF_KT * 4; ++i) asm volatile("" : "+v"(af[i]));
    row_rstd = 1.f; row_mean = 0.f;   // set by the seam behind the out_proj GEMM
  } else if constexpr (OP) {
    // ---- OP prologue: the attention context panel (bf16, read once in whole 128-byte lines: 16 bytes per lane, 8 lanes
    // per row of a 64-column k-tile) goes through the wave-private slice into the A fragments of the out_proj GEMM.  All
    // 24 loads of a lane in flight at once (asm + counted waits, as below): load k is complete once 23 - k younger ones
    // are outstanding.
    const unsigned short* crow_p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) crow_p[i] = delta + grow[i] * MF_D + 8 * cchunk;
    u32x4 cv[MF_KT][4];
#pragma unroll
    for (int S = 0; S < MF_KT; ++S)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (GWW_MF_ABL & 64) { cv[S][i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; continue; }
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" MF_NT : "=v"(cv[S][i]) : "v"(crow_p[i]), "n"(64 * S * 2) : "memory");
      }
#pragma unroll
    for (int S = 0; S < MF_KT; ++S) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!(GWW_MF_ABL & 64)) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cv[S][i]) : "n"(23 - (4 * S + i)));
        *reinterpret_cast<u32x4*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + cchunk * 16) = cv[S][i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 u = *reinterpret_cast<const u32x4*>(slice + r * MF_SLICE_STRIDE + (4 * hh + j) * 16);
        asm volatile("" : "+v"(u)::"memory");
        af[4 * S + j] = __builtin_bit_cast(bf16x8, u);
      }
    }
    row_rstd = 1.f; row_mean = 0.f;   // set by the seam behind the out_proj GEMM
  } else
  {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, cshift[4];
    // All loads of three k-tiles (48 per lane, 144 registers) are issued before the first use -- two exposed
    // round trips per panel instead of six; the accumulators are not live yet, the register file is free.
    // hipcc sinks plain loads back next to their uses, so the loads are asm and counted by hand: loads and
    // stores retire in order, pair k (x, delta) is complete once at most 46 - 2 k younger loads + the k
    // x_new stores issued since are outstanding.
    const float* xrow[4];
    const unsigned short* drow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xrow[i] = X + grow[i] * MF_D + 4 * cchunk;
      drow[i] = LNQ ? nullptr : delta + grow[i] * MF_D + 4 * cchunk;
    }
#pragma unroll
    for (int S3 = 0; S3 < MF_KT; S3 += 3) {
      f32x4 xv[3][2][4];
      u32x2 dv[3][2][4];
#pragma unroll
      for (int S = 0; S < 3; ++S)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" MF_NT
                         : "=v"(xv[S][h2][i]) : "v"(xrow[i]), "n"((64 * (S3 + S) + 32 * h2) * 4) : "memory");
            if (!LNQ)
              asm volatile("global_load_dwordx2 %0, %1, off offset:%2" MF_NT
                           : "=v"(dv[S][h2][i]) : "v"(drow[i]), "n"((64 * (S3 + S) + 32 * h2) * 2) : "memory");
          }
#pragma unroll
      for (int Sl = 0; Sl < 3; ++Sl) {
        const int S = S3 + Sl;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 v;
            if (LNQ) {   // no delta, no write-back: load k of the batch of 24 is complete once 23 - k younger loads are outstanding
              asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xv[Sl][h2][i]) : "n"(23 - (8 * Sl + 4 * h2 + i)));
              v = xv[Sl][h2][i];
            } else {
              asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xv[Sl][h2][i]), "+v"(dv[Sl][h2][i]) : "n"(46 - (8 * Sl + 4 * h2 + i)));
              v = xv[Sl][h2][i];
              v[0] += bf2f((unsigned short)(dv[Sl][h2][i][0] & 0xffff));
              v[1] += bf2f((unsigned short)(dv[Sl][h2][i][0] >> 16));
              v[2] += bf2f((unsigned short)(dv[Sl][h2][i][1] & 0xffff));
              v[3] += bf2f((unsigned short)(dv[Sl][h2][i][1] >> 16));
              *reinterpret_cast<f32x4*>(x_out + grow[i] * MF_D + 64 * S + 32 * h2 + 4 * cchunk) = v;
              asm volatile("" ::: "memory");   // keep the store count of the next wait exact
            }
            if (S == 0 && h2 == 0) {
              float t = (v[0] + v[1]) + (v[2] + v[3]);
              t = mf_sum8(t);
              cshift[i] = t * (1.0f / 32.0f);
            }
            v[0] -= cshift[i]; v[1] -= cshift[i]; v[2] -= cshift[i]; v[3] -= cshift[i];
            s1[i] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + h2 * 64 + cchunk * 8) = o;
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x4 u = *reinterpret_cast<const u32x4*>(slice + r * MF_SLICE_STRIDE + (4 * hh + j) * 16);
          asm volatile("" : "+v"(u)::"memory");
          af[4 * S + j] = __builtin_bit_cast(bf16x8, u);
        }
      }
    }
    TSTAMP(8);
    float* stat = reinterpret_cast<float*>(slice);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = s1[i], b = s2[i];
      a = mf_sum8(a); b = mf_sum8(b);
      const float mean = a * (1.0f / MF_D);
      const float var = fmaxf(b * (1.0f / MF_D) - mean * mean, 0.f);
      if (cchunk == 0) {
        stat[8 * i + crow] = rsqrtf(var + 1e-5f);
        stat[32 + 8 * i + crow] = mean;
      }
    }
    row_rstd = stat[r];
    row_mean = stat[32 + r];
    asm volatile("" : "+v"(row_rstd), "+v"(row_mean)::"memory");
    TSTAMP(9);
  }
#if GWW_MF_NORM
  // normalise the panel in place (every lane owns the fragments of ITS row r): 96 registers x 5 instructions, once per panel
  auto normalise_af = [&]() {
    const float nm = -row_mean * row_rstd;
#pragma unroll
    for (int i = 0; i < MF_KT * 4; ++i) {
      u32x4 w = __builtin_bit_cast(u32x4, af[i]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w[q] = pack2bf(fmaf(bf2f((unsigned short)(w[q] & 0xffff)), row_rstd, nm), fmaf(bf2f((unsigned short)(w[q] >> 16)), row_rstd, nm));
      af[i] = __builtin_bit_cast(bf16x8, w);
    }
  };
  if (!OP) normalise_af();   // (OP: after the seam that forms x_new)
#endif
  MSTAMP(0);

  // per-lane LDS offsets of the W fragments inside a tile
  //   fc1 tile [64 n][128 k], 256-byte rows, 16-byte chunk c stored at c ^ (row & 15):
  //     sub-tile tl (rows 32 tl + r), k-step (Sl, j) of the tile: chunk 8 Sl + 4 hh + j
  //   fc2 tile [128 n2][64 k], 128-byte rows, chunk c stored at c ^ ((row >> 1) & 7): k-step s: chunk 2 s + hh
  int off1[8], off2[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) off1[j] = r * 256 + ((((j >> 2) * 8 + 4 * hh + (j & 3)) ^ (r & 15)) << 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) off2[j] = r * 128 + (((2 * j + hh) ^ ((r >> 1) & 7)) << 4);
  //   QKV tile [128 n][64 k], 128-byte rows, same swizzle as fc2; k-step j of the tile: chunk 4 hh + j
  int offq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) offq[j] = r * 128 + (((4 * hh + j) ^ ((r >> 1) & 7)) << 4);

  f32x16 sacc[4], oacc[MF_OT];
  if constexpr (XACC) {   // the x pieces (landed: behind the prologue's ctx waits) become the accumulator tiles
#pragma unroll
    for (int t = 0; t < MF_OT; ++t) {
      asm volatile("" : "+a"(xp[t][0]), "+a"(xp[t][1]), "+a"(xp[t][2]), "+a"(xp[t][3]));
      const auto lo = __builtin_shufflevector(xp[t][0], xp[t][1], 0, 1, 2, 3, 4, 5, 6, 7);
      const auto hi = __builtin_shufflevector(xp[t][2], xp[t][3], 0, 1, 2, 3, 4, 5, 6, 7);
      oacc[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    }
  } else {
#pragma unroll
    for (int t = 0; t < MF_OT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) oacc[t][j] = 0.f;
  }
  u32x4 pf[4][2];   // bf16 operand fragments of fc2: pf[t][s] = gelu(S[t]) registers 8 s .. 8 s + 7

  // Activation piece p = 4 t + cc of chunk c: LayerNorm algebra + bias + GELU on registers 4 cc .. 4 cc + 3 of
  // S[t], one value per call so that the VALU work can be spread over the gaps between MFMAs.
  float a_u[2][4], a_b[2][4], a_v[2][4];
  auto act_begin = [&](int slot, int c, int p) {
    if (GWW_MF_EXP & 2) return;
    if (GWW_MF_SCHED) {   // the folded bias is the accumulators' initial value (preload_bias): nothing to add
      a_b[slot][0] = a_b[slot][1] = a_b[slot][2] = a_b[slot][3] = 0.f;
      return;
    }
    const int nl = 128 * c + 32 * (p >> 2) + 8 * (p & 3) + 4 * hh;
#if !GWW_MF_NORM
    const float4 uv = *reinterpret_cast<const float4*>(lds_u + nl);
    a_u[slot][0] = uv.x; a_u[slot][1] = uv.y; a_u[slot][2] = uv.z; a_u[slot][3] = uv.w;
#endif
    if (GWW_MF_EXP & 16) {   // diagnostic: no LDS read of the bias in front of the GELU chain
      a_b[slot][0] = a_b[slot][1] = a_b[slot][2] = a_b[slot][3] = row_rstd;
      return;
    }
    const float4 bv = *reinterpret_cast<const float4*>(lds_cb + nl);
    a_b[slot][0] = bv.x; a_b[slot][1] = bv.y; a_b[slot][2] = bv.z; a_b[slot][3] = bv.w;
  };
  auto act_val = [&](int slot, int p, int e) {
    if (GWW_MF_EXP & 2) {   // diagnostic: no GELU arithmetic, the fc1 result is only rounded and packed (the fc1 MFMAs stay live)
      a_v[slot][e] = sacc[p >> 2][4 * (p & 3) + e];
      return;
    }
#if GWW_MF_SCHED
    a_v[slot][e] = 0.125f * gelu_sig(8.0f * sacc[p >> 2][4 * (p & 3) + e]);   // S / 8 in, gelu / 8 out (both exact)
#elif GWW_MF_NORM
    a_v[slot][e] = gelu_sig(sacc[p >> 2][4 * (p & 3) + e] + a_b[slot][e]);
#else
    a_v[slot][e] = gelu_sig(fmaf(row_rstd, fmaf(-row_mean, a_u[slot][e], sacc[p >> 2][4 * (p & 3) + e]), a_b[slot][e]));
#endif
  };
  auto act_end = [&](int slot, int p) {
    const int t = p >> 2, cc = p & 3;
    pf[t][cc >> 1][2 * (cc & 1)] = pack2bf(a_v[slot][0], a_v[slot][1]);
    pf[t][cc >> 1][2 * (cc & 1) + 1] = pack2bf(a_v[slot][2], a_v[slot][3]);
    asm volatile("" : "+v"(pf[t][cc >> 1][2 * (cc & 1)]), "+v"(pf[t][cc >> 1][2 * (cc & 1) + 1]));
  };
  // the four values of a piece are independent chains (ILP 4 hides the v_exp / v_rcp latency); act_end pins
  // the result here (hipcc would sink the whole computation to its first use)
  auto act_piece = [&](int slot, int c, int p) {
    act_begin(slot, c, p);
    act_val(slot, p, 0); act_val(slot, p, 1); act_val(slot, p, 2); act_val(slot, p, 3);
    act_end(slot, p);
  };

  // ---- GELU as a stream of single instructions placed BY HAND into the MFMA gaps (GWW_MF_SCHED=1, default).
  // hipcc ignores the sched_group_barrier hints for this dependent chain: it emitted the ~30 instructions of a step's
  // values (+ the DMA issue) as ONE lump behind the first MFMA of the step and left 1-2 instructions in the other three
  // gaps -- the matrix pipe idled ~90 of every 216 cycles (ISA of round 1; a build without the GELU ran 37 % faster).
  // One wave per SIMD hides about five single-issue instructions per v_mfma_f32_32x32x16 gap, so the 16 values x 12
  // operations of an S index are cut into 48 slices of 4 -- a phase has 3 tiles x 16 MFMAs = 48 gaps: gap g carries
  // operations 2 j, 2 j + 1 (j = g % 6) of values 2 k and 2 k + 1 (k = g / 6), interleaved a b a b (one independent
  // instruction between dependent ones), and a sched_barrier(0) after every gap pins the placement.
  // What one wave per SIMD hides behind a v_mfma_f32_32x32x16_bf16 was measured in isolation (tools/ubench/mfma_gap.hip,
  // profiles/r02_mfma_gap.txt): five INDEPENDENT 4-cycle VALU instructions are free (32.8 cycles per MFMA with 4, 33.3
  // with 5, 35 with 6), but an instruction that depends on the one issued just before it costs 8 cycles, not 4 (a chain
  // a b a b over two values: 34.5 cycles with 4 per gap, 42 with 5, 47.5 with 6), v_exp / v_rcp cost 8 and so does every
  // v_accvgpr_read.  Round 2's first hand placement interleaved TWO values (a b a b): every instruction paid the 8-cycle
  // dependent price and the loop ran no faster than the compiler's lumps.  This one works on FOUR values per gap -- a gap
  // holds the same operation on values 4 k, 4 k + 1 (pair A, operation j) and the previous operation on 4 k + 2, 4 k + 3
  // (pair B, operation j - 1): nothing in a gap depends on anything in the same or the previous gap's tail, and a gap
  // never carries more than two transcendentals.  A phase has 3 tiles x 16 MFMAs = 48 gaps = 4 groups of 12; the 9
  // operations of a value (clamp(S^2), two fma, S q, exp2, + 1, rcp, S s, pack -- no bias add, no v_min: see
  // preload_bias and the S / 8 note in gelu_slice) fill gaps 0 .. 8 (A) and 1 .. 9 (B); gaps 9 .. 11 of every group
  // are where the riding tile's DMA pieces go.
  float g_t[4] = {0.f, 0.f, 0.f, 0.f}, g_w[4] = {0.f, 0.f, 0.f, 0.f}, g_q[4] = {0.f, 0.f, 0.f, 0.f};
#define MF_FENCE()                                                                                                       \
  asm volatile("" : "+v"(g_t[0]), "+v"(g_t[1]), "+v"(g_t[2]), "+v"(g_t[3]), "+v"(g_w[0]), "+v"(g_w[1]), "+v"(g_w[2]),   \
               "+v"(g_w[3]), "+v"(g_q[0]), "+v"(g_q[1]), "+v"(g_q[2]), "+v"(g_q[3]))
  // The folded fc1 bias cb is the INITIAL VALUE of the fc1 accumulators: the 16 values of S index t of 64-column chunk c
  // (register e holds column 64 c + 32 (t & 1) + 8 (e >> 2) + 4 hh + (e & 3)) are read from LDS straight into the
  // accumulator registers, one ds_read_b128 per gap, in the phase in which that accumulator pair is idle (its GELU has
  // been consumed, the chunk's first MFMA is three tiles away).  The per-value bias add (one VALU instruction per
  // activation value) and its LDS read in front of the GELU chain are gone.
  auto preload_bias = [&](int t, int chunk, int q) {
    const float4 bv = *reinterpret_cast<const float4*>(lds_cb + 64 * chunk + 32 * (t & 1) + 8 * q + 4 * hh);
    sacc[t][4 * q] = bv.x; sacc[t][4 * q + 1] = bv.y; sacc[t][4 * q + 2] = bv.z; sacc[t][4 * q + 3] = bv.w;
  };
  auto gelu_slice = [&](int t, int g) {   // t, g are compile-time constants after unrolling
    if (GWW_MF_EXP & 2) return;
    const int k = g / 12, j = g % 12;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int op = j - pr;
      if (op < 0 || op > 8) continue;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int i = 2 * pr + sl, v = 4 * k + i;
        const float S_ = sacc[t][v];
        float& T_ = g_t[i];
        float& W_ = g_w[i];
        float& Q_ = g_q[i];
        // x = 8 S_:  s' = min(x^2, 64) / 64 = clamp(S_^2) is ONE instruction (the VOP3 clamp modifier),
        // -log2(e) p(x) x = S_ (C1 + s' (C3 + s' C5)) with the powers of 8 folded into the constants
        if (op == 0) asm("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(W_) : "v"(S_));
        else if (op == 1) Q_ = fmaf(W_, 33.2535038f, -54.67091615f);
        else if (op == 2) Q_ = fmaf(W_, Q_, -18.4089424f);
        else if (op == 3) W_ = S_ * Q_;
        else if (op == 4) W_ = __builtin_amdgcn_exp2f(W_);
        else if (op == 5) W_ = 1.0f + W_;
        else if (op == 6) W_ = __builtin_amdgcn_rcpf(W_);
        else if (op == 7) T_ = S_ * W_;
        else if (op == 8 && sl == 1) {
          pf[t][k >> 1][2 * (k & 1) + pr] = pack2bf(g_t[2 * pr], g_t[2 * pr + 1]);
          asm volatile("" : "+v"(pf[t][k >> 1][2 * (k & 1) + pr]));   // the pack stays in this gap
        }
      }
    }
  };

  // tile 0 landed (younger tiles may still be in flight)
  TSTAMP(10);
  mf_wait_vmcnt<0>();   // tiles 0 .. AHEAD - 1 (requested at the kernel's start) and the prologue's stores
  __builtin_amdgcn_s_barrier();
  TSTAMP(11);

  // Software-pipelined schedule over 64-column ffn chunks c' = 0 .. n-1 (n = ffn / 64, even):
  //   G1(c') : three fc1 tiles [64 n][128 k] -> S[c' & 1]  (two 32-column accumulators)
  //   G2(c') : three fc2 tiles [128 n2][64 k], O += gelu(S(c')) W2^T, P = pf[c' & 1]
  //   stream : G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(n-1) G2(n-2) | G2(n-1)
  // The GELU of chunk c' (16 values per lane = 4 pieces) is split: its first half rides under G2(c' - 1), its
  // second half under G1(c' + 1) -- every tile carries 1.33 values per step (about 5 VALU instructions per MFMA
  // gap, which is what one wave per SIMD can hide) instead of half the tiles carrying twice that.  S and P are
  // double buffered by the chunk parity: no more registers than the 128-column schedule needed.
  // W fragments are read one step ahead of their use, across tile boundaries too (the ring waits run one tile
  // ahead, so tile it + 1 is complete and visible while tile it is being computed).
  bf16x8 wf[2][4];
  if (!LNQ && !OP) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      wf[0][t] = *reinterpret_cast<const bf16x8*>(lds + (t & 1) * 8192 + off1[t >> 1]);
  }
  if (OP) {   // first W_o tile ([128 n][64 k] image, the q / k / v tile format)
#pragma unroll
    for (int u = 0; u < 4; ++u) wf[0][u] = *reinterpret_cast<const bf16x8*>(lds + u * 4096 + offq[0]);
  }

  const int nck = F / 64;                                             // 64-column chunks
  const int total = OP_TILES + 6 * nck + (QKV ? (NQ / 128) * MF_KT : 0);   // tiles in the weight stream
  int stage = 0;   // ring stage of the tile being computed
  int it = 0;      // flat index of the tile being computed

  // fragment u of the FIRST step of a tile of the given kind (0 fc1, 1 fc2, 2 q/k/v): byte offset in the tile
  auto first_off = [&](int kind, int u) -> int {
    return kind == 0 ? (u & 1) * 8192 + off1[u >> 1] : (kind == 1 ? u * 4096 + off2[0] : u * 4096 + offq[0]);
  };

  // One tile = 4 steps of 4 MFMAs.  KIND 0: fc1 tile kt3 of the chunk whose accumulators are S[2 PAR + tl];
  // KIND 1: fc2 tile ng with P = pf[2 PAR + ..].  ride_t >= 0: S index whose GELU values ride in this tile
  // (values v0 .. of its four pieces, 16 per S index, spread 1-1-2 over the 12 steps of a phase);
  // next_kind: kind of the tile that follows (its first fragments are prefetched in the last step).
  auto run_tile = [&](auto flat_c, auto kind_c, auto par_c, auto idx3_c, auto ride_t_c, int ride_cpair, int next_kind,
                      auto pre_c, int pre_chunk, auto pad_c) {
    constexpr int KIND = decltype(kind_c)::value, PAR = decltype(par_c)::value, IDX3 = decltype(idx3_c)::value;
    constexpr int RIDE_T = decltype(ride_t_c)::value;
    // PAD: the asm MFMAs of the tiles outside the main loop carry two wait states in front: there hipcc may place register
    // copies (v_accvgpr_mov / _write into an accumulator tile it re-homes between the loop and the epilogue) directly in
    // front of an MFMA it cannot recognise inside an asm statement; the build's ISA audit checks that none is left.
    constexpr bool PAD = decltype(pad_c)::value != 0;
    // pad_c == 2: no ring wait in front of this tile.  Behind a seam the wave has consumed loads that were issued AFTER the
    // DMA pieces of the next two tiles: operations retire in issue order, so those pieces have landed -- while a counted
    // wait here would also wait for the seam's last stores, which are younger than the pieces and have no business
    // holding up a tile that does not need them (they get two tile times until the first tile whose pieces are younger).
    constexpr bool NOWAIT = decltype(pad_c)::value == 2;
    constexpr int PRE = decltype(pre_c)::value;   // >= 0: parity of the accumulator pair whose bias is preloaded in this tile
    // A four-stage ring makes the stage of every tile of the unrolled 12-tile body a compile-time constant (flat tile
    // index mod 4): every fragment address is then base register + immediate (no v_or_b32 per read: 8 cycles each
    // beside an MFMA, tools/ubench/mfma_gap.hip) and the DMA destination needs no wrap-around arithmetic.
    constexpr int ST = (12 % MF_NST == 0) ? ((decltype(flat_c)::value + OP_TILES) % MF_NST) : -1;   // (the unrolled body is 12 tiles: static for 4- and 6-stage rings)
    if (ST >= 0) stage = ST;
    if constexpr (!(GWW_MF_EXP & 64)) {   // (64: diagnostic, no ring wait / barrier -- only meaningful together with 1)
      if (!NOWAIT) mf_wait_vmcnt<MF_GL * (MF_AHEAD - 2)>();
      __builtin_amdgcn_s_barrier();
    }
    MSTAMP(1);
    const int dma_tile = it + MF_AHEAD < total ? it + MF_AHEAD : total - 1;
    const int dma_stage = stage + MF_AHEAD >= MF_NST ? stage + MF_AHEAD - MF_NST : stage + MF_AHEAD;
    const int stage_next = stage + 1 == MF_NST ? 0 : stage + 1;
    MSTAMP(2);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      bf16x8(&cur)[4] = wf[sub & 1];          // every tile has 4 steps: the parity is static
      bf16x8(&nxt)[4] = wf[(sub + 1) & 1];
      const unsigned char* Wn = lds + (sub == 3 ? stage_next : stage) * MF_TILE;
      // GELU values of this step: step j = 4 IDX3 + sub of the phase takes values [16 j / 12, 16 (j + 1) / 12)
      const int j12 = 4 * IDX3 + sub;
      const int v0 = (16 * j12) / 12, v1 = (16 * (j12 + 1)) / 12;
      __builtin_amdgcn_sched_barrier(0);
      // DMA pieces: in a riding tile they go into the gaps the GELU schedule leaves (nearly) empty -- gaps 9, 10, 11 of
      // every 12-gap group carry at most the pack, gap 0 only pair A's first operation: phase gaps 9 .. 12, 21 .. 24,
      // 33 .. 36, i.e. tile-local gaps 9 .. 12 / 5 .. 8 / 1 .. 4 of tiles 0 / 1 / 2 of the phase (an LDS-DMA piece costs
      // ~27 cycles of issue when it shares a gap with a full slice, tools/ubench/mfma_gap.hip); elsewhere one per step
      constexpr bool DMA_IN_GAPS = GWW_MF_SCHED && GWW_MF_DMAGAP && RIDE_T >= 0;
      if (!(GWW_MF_EXP & 1) && !DMA_IN_GAPS) issue_piece(dma_tile, dma_stage, sub);
      if (!GWW_MF_SCHED && RIDE_T >= 0) {
#pragma unroll
        for (int v = v0; v < v1; ++v) {
          const int p = 4 * RIDE_T + (v >> 2), e = v & 3;
          if (e == 0) act_begin(0, ride_cpair, p);
          act_val(0, p, e);
          if (e == 3) act_end(0, p);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (KIND == 0) {
          // ---- fc1: S[2 PAR + tl] (+)= W1' tile rows 32 tl .. . a[k-step]
          const int tl = u & 1;
          const int ks = 2 * sub + (u >> 1);                       // k-step of the tile, 0 .. 7
          const int afi = 4 * (2 * IDX3 + (ks >> 2)) + (ks & 3);   // af[4 S + j]
          if (GWW_MF_SCHED && GWW_MF_VACC) {
            // fc1 accumulates in ARCHITECTURAL registers (asm: hipcc gives a builtin MFMA's result to the accumulator file
            // and then copies all 16 values out with v_accvgpr_read_b32 for the GELU -- 8 cycles each beside an MFMA,
            // tools/ubench/mfma_gap.hip).  The GELU reads S two or more MFMAs after the last one that wrote it (the
            // slices are pinned into their gaps), which covers the MFMA-write -> VALU-read interval hipcc cannot see here.
            if (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[2 * PAR + tl]) : "v"(cur[u]), "v"(af[afi]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[2 * PAR + tl]) : "v"(cur[u]), "v"(af[afi]));
          } else if (IDX3 == 0 && ks == 0) {
            f32x16 z;
#pragma unroll
            for (int j = 0; j < 16; ++j) z[j] = 0.f;
            sacc[2 * PAR + tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[u], af[afi], z, 0, 0, 0);
          } else {
            sacc[2 * PAR + tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[u], af[afi], sacc[2 * PAR + tl], 0, 0, 0);
          }
        } else {
          // ---- fc2: O[4 ng + u] += W2 tile (rows 32 u ..) . P[k-step];  ng = IDX3
          if (GWW_MF_SCHED && PAD)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                         : "+a"(oacc[4 * IDX3 + u]) : "v"(cur[u]), "v"(pf[2 * PAR + (sub >> 1)][sub & 1]));
          else if (GWW_MF_SCHED)   // asm: volatile statements keep their order, so no fence has to name an MFMA operand (below)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                         : "+a"(oacc[4 * IDX3 + u]) : "v"(cur[u]), "v"(pf[2 * PAR + (sub >> 1)][sub & 1]));
          else
            oacc[4 * IDX3 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                cur[u], __builtin_bit_cast(bf16x8, pf[2 * PAR + (sub >> 1)][sub & 1]), oacc[4 * IDX3 + u], 0, 0, 0);
        }
        // Every MFMA of the riding tiles is a volatile asm statement and so is the fence behind each slice: their order is
        // the program's.  The fence names ONLY the GELU state: a fence that (re)defined the next MFMA's W fragment made
        // hipcc pad every MFMA with an s_nop (it must assume a VALU write two wait states in front of an MFMA read; 3.5
        // cycles per gap, tools/ubench/mfma_gap.hip).  The fragment's own "definition" below sits in front of the slice:
        // the counted LDS wait for MFMA u + 1 lands there and the slice fills the wait states.
        if (GWW_MF_SCHED && RIDE_T >= 0) {
          if (u < 3) asm volatile("" : "+v"(cur[u + 1]));
          else if (sub < 3) asm volatile("" : "+v"(nxt[0]));
        }
        if (!(GWW_MF_EXP & 4)) {
          int off;
          if (sub == 3) off = first_off(next_kind, u);
          else off = KIND == 0 ? (u & 1) * 8192 + off1[2 * (sub + 1) + (u >> 1)] : u * 4096 + off2[sub + 1];
          nxt[u] = *reinterpret_cast<const bf16x8*>(Wn + off);
        }
        if (GWW_MF_SCHED && RIDE_T >= 0) {
          if (PRE >= 0 && 4 * sub + u < 8) preload_bias(2 * PRE + ((4 * sub + u) >> 2), pre_chunk, (4 * sub + u) & 3);
          gelu_slice(RIDE_T, 16 * IDX3 + 4 * sub + u);
          if (DMA_IN_GAPS && !(GWW_MF_EXP & 1)) {
            const int piece = 4 * sub + u - (9 - 4 * IDX3);
            if (piece >= 0 && piece < 4) issue_piece(dma_tile, dma_stage, piece);
          }
          MF_FENCE();
        }
      }
      if (GWW_MF_SCHED) continue;
      // pipeline of the step: MFMA, fragment read, a slice of the VALU work -- four times
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage_next;
    ++it;
    MSTAMP(8 + (KIND == 0 ? IDX3 : 3 + IDX3) + 6 * PAR);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using IM = std::integral_constant<int, -1>;

  // ---- the seam: residual + bf16(accumulated output + bias) -> new residual stream, LayerNorm statistics, A fragments.
  // Per 64-column chunk np the output tile (rounded to bf16 exactly like the stand-alone kernels' delta) goes through the
  // wave-private slice into row order, new = src + it is formed from whole-line reads of src, written to dst and shifted /
  // measured / packed into the A fragments of k-tile np.  Used behind fc2 (src = x_new, dst = x_next, bias = b2: the
  // operand of the next layer's q / k / v) and -- OP mode -- behind the fused out_proj (src = x, dst = x_new, bias = bo:
  // the operand of fc1).  The ring tiles in flight are older than these loads: hipcc's own vmcnt waits retire them first.
  // Stores go out chunk by chunk (8 per 64-column chunk) and the second batch of loads is requested BEFORE the stores of
  // the chunk just finished: loads, stores and LDS-DMA retire in issue order, so a wait for a load that sits behind 24
  // fresh stores also waits for those stores to reach the L2 (round 2 order: 24 stores, then 24 loads, then the waits --
  // store drain + load latency in series, twice per seam; GWW_MF_ABL = 16 / 32 / 48 builds).  Now the waits name the
  // stores issued since the request as younger operations and leave them in flight.  That needs EXACT queue counts: loads
  // and stores are asm, every lane of every store is live.  Rows past M are clamped duplicates of row M - 1 on both
  // sides: every duplicate computes and stores the same value, which is harmless as long as the seam does not run in
  // place (seam_dst != seam_src: the launcher guarantees it, x_next goes to the buffer x came from).  Addresses are one
  // 32-bit row offset per lane and row + a scalar base (M * 1536 bytes < 4 GiB is checked at launch).
  auto seam = [&](const float* seam_src, float* seam_dst, const float* seam_bias, auto&& after_request) {
      unsigned roff[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) roff[i] = (unsigned)grow[i] * (unsigned)(MF_D * 4) + 16u * (unsigned)cchunk;

      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, cshift[4];
      // the x_new lines of three 64-column chunks are requested together (24 loads per lane; with the 192 output
      // accumulators still live there is room for no more) -- two exposed round trips instead of six; asm +
      // hand-counted vmcnt as in the prologue
      // (second batch parked in accumulator registers: by then the operand fragments fill the arch VGPRs and
      //  hipcc would otherwise copy the just-requested registers away BEFORE the data has landed; half of the output
      //  accumulators are free at that point)
      f32x4 xn4[3][2][4], xa4[3][2][4];
      auto request = [&](int np0) {   // the 24 loads of chunks np0 .. np0 + 2
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (GWW_MF_ABL & 32) { xn4[q][h2][i] = f32x4{0.5f, -0.25f, 1.f, 0.f}; xa4[q][h2][i] = f32x4{0.5f, -0.25f, 1.f, 0.f}; continue; }
              if (np0 == 0)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" MF_NT
                             : "=v"(xn4[q][h2][i]) : "v"(roff[i]), "s"(seam_src), "n"((64 * q + 32 * h2) * 4) : "memory");
              else
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" MF_NT
                             : "=a"(xa4[q][h2][i]) : "v"(roff[i]), "s"(seam_src), "n"((64 * (3 + q) + 32 * h2) * 4) : "memory");
            }
      };
      auto store_chunk = [&](int np) {   // x_next of chunk np: 8 stores, every lane live (the s_nop covers the interval in
                                         // which the store still reads its data registers)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (GWW_MF_ABL & 16) asm volatile("" :: "v"(xn4[np % 3][h2][i]));
            else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1"
                              : : "v"(roff[i]), "v"(xn4[np % 3][h2][i]), "s"(seam_dst), "n"((64 * np + 32 * h2) * 4) : "memory");
          }
      };
#pragma unroll
      for (int np = 0; np < MF_KT; ++np) {
        // the chunks alternate between two wave-private slices: chunk np + 1's transposed output tile can be written while
        // chunk np's operand fragments are still being read back (one slice serialised the six chunks on their LDS round trips)
        unsigned char* const slice = slice0 + (np & 1) * (MF_WAVES * MF_SLICE_BYTES);
        if (np == 0) {
          request(0);
          after_request();   // work whose own memory latency runs beside the first batch's
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * np + tt;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int nl = 32 * t + 8 * cc + 4 * hh;
            const float4 bv = *reinterpret_cast<const float4*>(seam_bias + nl);
            u32x2 o = {pack2bf(oacc[t][4 * cc] + bv.x, oacc[t][4 * cc + 1] + bv.y),
                       pack2bf(oacc[t][4 * cc + 2] + bv.z, oacc[t][4 * cc + 3] + bv.w)};
            *reinterpret_cast<u32x2*>(slice + r * MF_SLICE_STRIDE + (32 * tt + 8 * cc + 4 * hh) * 2) = o;
          }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // load k = 8 (np % 3) + 4 h2 + i of its batch: the 23 - k younger loads of the batch and the 8 stores of every
            // chunk finished since the request (the second batch is requested in front of chunk 2's stores) may be outstanding
            const int ST_SINCE = 8 * (np % 3) + (np >= 3 ? 8 : 0);
            f32x4 v;
            if (np < 3) {
              if (!(GWW_MF_ABL & 32)) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xn4[np % 3][h2][i]) : "n"(ST_SINCE + 23 - (8 * (np % 3) + 4 * h2 + i)));
              v = xn4[np % 3][h2][i];
            } else {
              if (!(GWW_MF_ABL & 32)) asm volatile("s_waitcnt vmcnt(%1)" : "+a"(xa4[np % 3][h2][i]) : "n"(ST_SINCE + 23 - (8 * (np % 3) + 4 * h2 + i)));
              v = xa4[np % 3][h2][i];
            }
            const u32x2 dv = *reinterpret_cast<const u32x2*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + h2 * 64 + cchunk * 8);
            v[0] += bf2f((unsigned short)(dv[0] & 0xffff));
            v[1] += bf2f((unsigned short)(dv[0] >> 16));
            v[2] += bf2f((unsigned short)(dv[1] & 0xffff));
            v[3] += bf2f((unsigned short)(dv[1] >> 16));
            xn4[np % 3][h2][i] = v;      // x_next, stored once the batch is consumed
            if (np == 0 && h2 == 0) {
              float t = (v[0] + v[1]) + (v[2] + v[3]);
              t = mf_sum8(t);
              cshift[i] = t * (1.0f / 32.0f);
            }
            v[0] -= cshift[i]; v[1] -= cshift[i]; v[2] -= cshift[i]; v[3] -= cshift[i];
            s1[i] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + h2 * 64 + cchunk * 8) = o;
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x4 u = *reinterpret_cast<const u32x4*>(slice + r * MF_SLICE_STRIDE + (4 * hh + j) * 16);
          asm volatile("" : "+v"(u)::"memory");
          af[4 * np + j] = __builtin_bit_cast(bf16x8, u);
        }
        if (np == 2) request(3);   // batch 0 consumed: batch 1 is requested first, chunk 2's stores go behind it
        store_chunk(np);
      }
      float* stat = reinterpret_cast<float*>(slice0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = s1[i], b = s2[i];
        a = mf_sum8(a); b = mf_sum8(b);
        const float mean = a * (1.0f / MF_D);
        const float var = fmaxf(b * (1.0f / MF_D) - mean * mean, 0.f);
        if (cchunk == 0) {
          stat[8 * i + crow] = rsqrtf(var + 1e-5f);
          stat[32 + 8 * i + crow] = mean;
        }
      }
      row_rstd = stat[r];
      row_mean = stat[32 + r];
      asm volatile("" : "+v"(row_rstd), "+v"(row_mean)::"memory");
      };
  // ---- XACC seam: O (+ bias) IS the new residual stream, in accumulator layout.  Pass A: optional store (x_new for the
  // training path, x_next behind fc2), shifted row sums in-lane, one cross-half add.  Pass B: (v - mean) rstd -> bf16 pairs;
  // lane r (hh = 0) owns columns 8 cc .. + 3 of every 32-block, lane r + 32 columns 8 cc + 4 .. + 7, and the A fragment
  // af[4 S + cc] of lane (r, hh) is k = 64 S + 32 hh + 8 cc .. + 7: the low lane needs the partner's half of the EVEN block
  // 2 S, the high lane the partner's half of the ODD block 2 S + 1 -- exactly one v_permlane32_swap per packed word pair
  // (attention.hip's GWW_ATT_EPI16 epilogue uses the same exchange).  One rounding, no LDS, no second normalise pass.
  auto seam_x = [&](const float* sb, float* dst) {
    if constexpr (XACC) {
    float s1 = 0.f, s2 = 0.f, csh = 0.f;
#pragma unroll
    for (int t = 0; t < MF_OT; ++t)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float4 bv = *reinterpret_cast<const float4*>(sb + 32 * t + 8 * cc + 4 * hh);
        f32x4 v = {oacc[t][4 * cc] + bv.x, oacc[t][4 * cc + 1] + bv.y, oacc[t][4 * cc + 2] + bv.z, oacc[t][4 * cc + 3] + bv.w};
        if (dst && !(GWW_MF_ABL & 16)) {
          if (GWW_MF_XLDS) {   // into the slice in accumulator layout (rows 144 bytes apart: conflict-free for the 16-lane groups)
            *reinterpret_cast<f32x4*>(slice0 + (t & 1) * (MF_WAVES * MF_SLICE_BYTES) + r * MF_SLICE_STRIDE + (8 * cc + 4 * hh) * 4) = v;
          } else {
            f32x4* const sp = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst) + xoff + (32 * t + 8 * cc) * 4);
            if (GWW_MF_XSTNT) __builtin_nontemporal_store(v, sp);
            else *sp = v;
          }
        }
        if (GWW_MF_XLDS && dst && !(GWW_MF_ABL & 16) && cc == 3) {   // the 32-column block back in row order: 8 rows x 128 bytes per store
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(slice0 + (t & 1) * (MF_WAVES * MF_SLICE_BYTES) + (8 * i + crow) * MF_SLICE_STRIDE + cchunk * 16);
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst) + (unsigned)grow[i] * (unsigned)(MF_D * 4) + 16u * (unsigned)cchunk + t * 128) = u;
          }
        }
        if (t == 0 && cc == 0) {   // one shift per ROW: the low lane's first value, for both halves
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[0]), false, false);
          csh = hh ? __uint_as_float(sw[0]) : v[0];   // (high lanes: [0] = the low lanes' operand)
        }
        v[0] -= csh; v[1] -= csh; v[2] -= csh; v[3] -= csh;
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    {
      const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
      s1 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
      s2 = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    const float mean_s = s1 * (1.0f / MF_D);
    row_rstd = rsqrtf(fmaxf(s2 * (1.0f / MF_D) - mean_s * mean_s, 0.f) + 1e-5f);
    row_mean = csh + mean_s;
    const float nm = -row_mean * row_rstd;
    // (a fence between the passes: without it hipcc keeps all 192 sums O + bias of pass A alive for pass B -- fifty spills)
    asm volatile("" : "+a"(oacc[0]), "+a"(oacc[1]), "+a"(oacc[2]), "+a"(oacc[3]), "+a"(oacc[4]), "+a"(oacc[5]), "+a"(oacc[6]),
                      "+a"(oacc[7]), "+a"(oacc[8]), "+a"(oacc[9]), "+a"(oacc[10]), "+a"(oacc[11]));
#pragma unroll
    for (int S = 0; S < MF_KT; ++S)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float4 be = *reinterpret_cast<const float4*>(sb + 64 * S + 8 * cc + 4 * hh);
        const float4 bo_ = *reinterpret_cast<const float4*>(sb + 64 * S + 32 + 8 * cc + 4 * hh);
        const f32x16& E = oacc[2 * S];
        const f32x16& O = oacc[2 * S + 1];
        const unsigned e0 = pack2bf(fmaf(E[4 * cc] + be.x, row_rstd, nm), fmaf(E[4 * cc + 1] + be.y, row_rstd, nm));
        const unsigned e1 = pack2bf(fmaf(E[4 * cc + 2] + be.z, row_rstd, nm), fmaf(E[4 * cc + 3] + be.w, row_rstd, nm));
        const unsigned o0 = pack2bf(fmaf(O[4 * cc] + bo_.x, row_rstd, nm), fmaf(O[4 * cc + 1] + bo_.y, row_rstd, nm));
        const unsigned o1 = pack2bf(fmaf(O[4 * cc + 2] + bo_.z, row_rstd, nm), fmaf(O[4 * cc + 3] + bo_.w, row_rstd, nm));
        // swap(x, y): x's upper 32 lanes <-> y's lower 32 lanes.  Low lanes: [0] own half of the even block, [1] the
        // partner's; high lanes: [0] the partner's half of the odd block, [1] own
        const auto w0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
        const auto w1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
        const u32x4 f = {w0[0], w1[0], w0[1], w1[1]};
        af[4 * S + cc] = __builtin_bit_cast(bf16x8, f);
      }
      }
  };
  if constexpr (OP) {
    // ======== out_proj in front of the block: O[n][m] = sum_k W_o[n][k] ctx[m][k], 3 n-tiles x 6 k-tiles of the q / k / v
    // tile format, accumulated into the (zeroed, still idle) output accumulators O[4 nt + u]
    static_assert(12 % MF_NST == 0, "OP mode relies on a ring whose depth divides the 12-tile body (4 or 6 stages)");
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
      for (int kt = 0; kt < MF_KT; ++kt) {
        const int itile = nt * MF_KT + kt;               // compile-time: the stage is static
        if (!(GWW_MF_ABL & 128)) mf_wait_vmcnt<MF_GL * (MF_AHEAD - 2)>();
        __builtin_amdgcn_s_barrier();
        const int st = itile % MF_NST, st_next = (itile + 1) % MF_NST, dma_st = (itile + MF_AHEAD) % MF_NST;
        const int dma_tile = itile + MF_AHEAD < total ? itile + MF_AHEAD : total - 1;
        const bool last = itile == OP_TILES - 1;          // its last step prefetches the first fc1 tile's fragments
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int q = 4 * kt + sub;
          bf16x8(&cur)[4] = wf[q & 1];
          bf16x8(&nxt)[4] = wf[(q + 1) & 1];
          const unsigned char* Wn = lds + (sub == 3 ? st_next : st) * MF_TILE;
          if (!(GWW_MF_ABL & 128)) issue_piece(dma_tile, dma_st, sub);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            asm volatile(MF_PADNOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(oacc[4 * nt + u]) : "v"(cur[u]), "v"(af[4 * kt + sub]));
            const int off = (last && sub == 3) ? (u & 1) * 8192 + off1[u >> 1] : u * 4096 + offq[(sub + 1) & 3];
            nxt[u] = *reinterpret_cast<const bf16x8*>(Wn + off);
          }
        }
      }
    }
    it = OP_TILES;
    stage = OP_TILES % MF_NST;
    TSTAMP(14);
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(oacc[0]), "+a"(oacc[1]), "+a"(oacc[2]), "+a"(oacc[3]), "+a"(oacc[4]), "+a"(oacc[5]), "+a"(oacc[6]),
                   "+a"(oacc[7]), "+a"(oacc[8]), "+a"(oacc[9]), "+a"(oacc[10]), "+a"(oacc[11]));
    if constexpr (XACC) {
      // x_new = x + ctx W_o^T + bo sits in O (bo is added on the fly): LN2 statistics and the fc1 operand straight from the
      // registers; fc2 keeps accumulating on top of it
      seam_x(lds_bo, (keep_x_new && !(GWW_MF_ABL & 256)) ? x_out : nullptr);   // (256: diagnostic, the ABI entry without x_new)
    } else {
    // x_new = x + bf16(ctx W_o^T + bo): written to x_out, LN2 statistics, A fragments
    seam(X, x_out, lds_bo, [] {});
#if GWW_MF_NORM
    normalise_af();
#endif
#pragma unroll
    for (int t = 0; t < MF_OT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) oacc[t][j] = 0.f;   // fc2 accumulates from zero
    }
    TSTAMP(15);
  }
  if constexpr (!LNQ) {   // ======== the MLP stream (MODE 2 has none: straight to the q / k / v tail)
  if (GWW_MF_SCHED) {   // bias of chunks 0 and 1 into S[0..1] and S[2..3] (later chunks: in the tiles marked below)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) preload_bias(t, t >> 1, q);
  }
  // G1(0): no GELU to carry yet; its first half is then computed in the open (once per 128 rows)
  using IW = std::conditional_t<OP && !XACC && !(GWW_MF_ABL & 32), I2, I1>;   // behind the out_proj seam: tiles 0 / 1 have landed (XACC: its seam loads nothing, the ring waits are the regular ones)
  run_tile(MF_FL<0>{}, I0{}, I0{}, I0{}, IM{}, 0, 0, IM{}, 0, IW{});
  run_tile(MF_FL<1>{}, I0{}, I0{}, I1{}, IM{}, 0, 0, IM{}, 0, IW{});
  run_tile(MF_FL<2>{}, I0{}, I0{}, I2{}, IM{}, 0, 0, IM{}, 0, I1{});
  // (hipcc cannot see that the asm MFMAs write S: the MFMA-write -> VALU-read interval is padded by hand)
  TSTAMP(12);
  if (GWW_MF_SCHED) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]));
  act_piece(0, 0, 0); act_piece(0, 0, 1); act_piece(0, 0, 2); act_piece(0, 0, 3);
  TSTAMP(13);
  int b = 0;
  for (; b + 2 < nck; b += 2) {
    // ---- block b (parity 0): G1(b + 1) -> S[2..3] carrying the second half of GELU(b) (S[1]);
    //      G2(b) with pf[0..1] carrying the first half of GELU(b + 1) (S[2])
    run_tile(MF_FL<3>{}, I0{}, I1{}, I0{}, I1{}, b >> 1, 0, IM{}, 0, I0{});
    run_tile(MF_FL<4>{}, I0{}, I1{}, I1{}, I1{}, b >> 1, 0, IM{}, 0, I0{});
    run_tile(MF_FL<5>{}, I0{}, I1{}, I2{}, I1{}, b >> 1, 1, IM{}, 0, I0{});
    run_tile(MF_FL<6>{}, I1{}, I0{}, I0{}, I2{}, b >> 1, 1, I0{}, b + 2, I0{});
    run_tile(MF_FL<7>{}, I1{}, I0{}, I1{}, I2{}, b >> 1, 1, IM{}, 0, I0{});
    run_tile(MF_FL<8>{}, I1{}, I0{}, I2{}, I2{}, b >> 1, 0, IM{}, 0, I0{});
    // ---- block b + 1 (parity 1): G1(b + 2) -> S[0..1] carrying the second half of GELU(b + 1) (S[3]);
    //      G2(b + 1) with pf[2..3] carrying the first half of GELU(b + 2) (S[0])
    run_tile(MF_FL<9>{}, I0{}, I0{}, I0{}, I3{}, b >> 1, 0, IM{}, 0, I0{});
    run_tile(MF_FL<10>{}, I0{}, I0{}, I1{}, I3{}, b >> 1, 0, IM{}, 0, I0{});
    run_tile(MF_FL<11>{}, I0{}, I0{}, I2{}, I3{}, b >> 1, 1, IM{}, 0, I0{});
    run_tile(MF_FL<12>{}, I1{}, I1{}, I0{}, I0{}, (b >> 1) + 1, 1, I1{}, b + 3, I0{});
    run_tile(MF_FL<13>{}, I1{}, I1{}, I1{}, I0{}, (b >> 1) + 1, 1, IM{}, 0, I0{});
    run_tile(MF_FL<14>{}, I1{}, I1{}, I2{}, I0{}, (b >> 1) + 1, 0, IM{}, 0, I0{});
  }
  // ---- last pair (b = n - 2): G1(n - 1) + second half of GELU(n - 2); G2(n - 2) + first half of GELU(n - 1);
  //      then the second half of GELU(n - 1) in the open (nothing left to hide it under) and G2(n - 1)
  run_tile(MF_FL<3>{}, I0{}, I1{}, I0{}, I1{}, b >> 1, 0, IM{}, 0, I1{});
  run_tile(MF_FL<4>{}, I0{}, I1{}, I1{}, I1{}, b >> 1, 0, IM{}, 0, I1{});
  run_tile(MF_FL<5>{}, I0{}, I1{}, I2{}, I1{}, b >> 1, 1, IM{}, 0, I1{});
  run_tile(MF_FL<6>{}, I1{}, I0{}, I0{}, I2{}, b >> 1, 1, IM{}, 0, I1{});
  run_tile(MF_FL<7>{}, I1{}, I0{}, I1{}, I2{}, b >> 1, 1, IM{}, 0, I1{});
  run_tile(MF_FL<8>{}, I1{}, I0{}, I2{}, I2{}, b >> 1, 1, IM{}, 0, I1{});
  if (GWW_MF_SCHED) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]));
  act_piece(0, b >> 1, 12); act_piece(0, b >> 1, 13); act_piece(0, b >> 1, 14); act_piece(0, b >> 1, 15);
  run_tile(MF_FL<9>{}, I1{}, I1{}, I0{}, IM{}, 0, 1, IM{}, 0, I1{});
  run_tile(MF_FL<10>{}, I1{}, I1{}, I1{}, IM{}, 0, 1, IM{}, 0, I1{});
  run_tile(MF_FL<11>{}, I1{}, I1{}, I2{}, IM{}, 0, QKV ? (GWW_MF_TAIL2 ? 0 : 2) : 1, IM{}, 0, I1{});

  MSTAMP(4);
  if (GWW_MF_SCHED)   // asm MFMAs -> accumulator reads of the epilogue (as above); the operands keep the reads behind the pad
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(oacc[0]), "+a"(oacc[1]), "+a"(oacc[2]), "+a"(oacc[3]), "+a"(oacc[4]), "+a"(oacc[5]), "+a"(oacc[6]),
                   "+a"(oacc[7]), "+a"(oacc[8]), "+a"(oacc[9]), "+a"(oacc[10]), "+a"(oacc[11]));
  }   // ======== !LNQ
  if constexpr (FIN && XACC) {
    // ---- MODE 3 with the residual stream in O: x_fin = O + (b2 + bo); row statistics as in seam_x, then
    // y = (x_fin - mean) rstd g + b goes to last_hidden_state from the accumulator layout (HF:modeling_whisper.py:642)
    mf_wait_vmcnt<0>();   // the ring's re-reads issued past the end
    float* y_out = reinterpret_cast<float*>(C);
    float s1 = 0.f, s2 = 0.f, csh = 0.f;
#pragma unroll
    for (int t = 0; t < MF_OT; ++t)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float4 bv = *reinterpret_cast<const float4*>(lds_b2 + 32 * t + 8 * cc + 4 * hh);
        f32x4 v = {oacc[t][4 * cc] + bv.x, oacc[t][4 * cc + 1] + bv.y, oacc[t][4 * cc + 2] + bv.z, oacc[t][4 * cc + 3] + bv.w};
        if (t == 0 && cc == 0) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[0]), false, false);
          csh = hh ? __uint_as_float(sw[0]) : v[0];
        }
        v[0] -= csh; v[1] -= csh; v[2] -= csh; v[3] -= csh;
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    {
      const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
      s1 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
      s2 = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    const float mean_s = s1 * (1.0f / MF_D);
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / MF_D) - mean_s * mean_s, 0.f) + 1e-5f);
    const float mean = csh + mean_s;
    // (a fence between the passes: without it hipcc keeps all 192 sums O + bias of pass A alive for pass B -- fifty spills)
    asm volatile("" : "+a"(oacc[0]), "+a"(oacc[1]), "+a"(oacc[2]), "+a"(oacc[3]), "+a"(oacc[4]), "+a"(oacc[5]), "+a"(oacc[6]),
                      "+a"(oacc[7]), "+a"(oacc[8]), "+a"(oacc[9]), "+a"(oacc[10]), "+a"(oacc[11]));
#pragma unroll
    for (int t = 0; t < MF_OT; ++t)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int col = 32 * t + 8 * cc + 4 * hh;
        const float4 bv = *reinterpret_cast<const float4*>(lds_b2 + col);
        const float4 g4 = *reinterpret_cast<const float4*>(lds_u + col);
        const float4 b4 = *reinterpret_cast<const float4*>(lds_u + MF_D + col);
        const f32x4 y = {fmaf((oacc[t][4 * cc] + bv.x - mean) * rstd, g4.x, b4.x), fmaf((oacc[t][4 * cc + 1] + bv.y - mean) * rstd, g4.y, b4.y),
                         fmaf((oacc[t][4 * cc + 2] + bv.z - mean) * rstd, g4.z, b4.z), fmaf((oacc[t][4 * cc + 3] + bv.w - mean) * rstd, g4.w, b4.w)};
        // rows past M are clamped duplicates of row M - 1: every duplicate stores the same value
        if (GWW_MF_XLDS) {
          *reinterpret_cast<f32x4*>(slice0 + (t & 1) * (MF_WAVES * MF_SLICE_BYTES) + r * MF_SLICE_STRIDE + (8 * cc + 4 * hh) * 4) = y;
          if (cc == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 u = *reinterpret_cast<const f32x4*>(slice0 + (t & 1) * (MF_WAVES * MF_SLICE_BYTES) + (8 * i + crow) * MF_SLICE_STRIDE + cchunk * 16);
              *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(y_out) + (unsigned)grow[i] * (unsigned)(MF_D * 4) + 16u * (unsigned)cchunk + t * 128) = u;
            }
          }
        } else if (GWW_MF_XSTNT) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(y_out) + xoff + (32 * t + 8 * cc) * 4));
        else *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(y_out) + xoff + (32 * t + 8 * cc) * 4) = y;
      }
  } else if constexpr (FIN) {
    // ---- MODE 3, the LAST block of the encoder: the epilogue is the final LayerNorm (HF:modeling_whisper.py:642).
    // x_fin = x_new + bf16(fc2 output + b2) is formed exactly as the second seam forms x_next (same roundings as the
    // stand-alone delta + LayerNorm kernels), but all 48 row pieces of the panel stay in registers -- the accumulator
    // tiles free 32 registers per chunk, a chunk's pieces take 32 -- until the row statistics are complete; then
    // y = (x_fin - mean) rstd g + b goes straight to last_hidden_state.  The bf16 delta (0.3 GB per launch at B = 256),
    // its re-read, the second read of the residual stream and the LayerNorm launch disappear.
    mf_wait_vmcnt<0>();   // the ring's re-reads issued past the end
    float* y_out = reinterpret_cast<float*>(C);
    unsigned roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) roff[i] = (unsigned)grow[i] * (unsigned)(MF_D * 4) + 16u * (unsigned)cchunk;
    f32x4 xh[MF_KT][2][4];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, cshift[4];
    auto request = [&](int np0) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (np0 == 0)
              asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" MF_NT
                           : "=v"(xh[q][h2][i]) : "v"(roff[i]), "s"(x_out), "n"((64 * q + 32 * h2) * 4) : "memory");
            else
              asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" MF_NT
                           : "=a"(xh[3 + q][h2][i]) : "v"(roff[i]), "s"(x_out), "n"((64 * (3 + q) + 32 * h2) * 4) : "memory");
          }
    };
    request(0);
#pragma unroll
    for (int np = 0; np < MF_KT; ++np) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * np + tt;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int nl = 32 * t + 8 * cc + 4 * hh;
          const float4 bv = *reinterpret_cast<const float4*>(lds_b2 + nl);
          u32x2 o = {pack2bf(oacc[t][4 * cc] + bv.x, oacc[t][4 * cc + 1] + bv.y),
                     pack2bf(oacc[t][4 * cc + 2] + bv.z, oacc[t][4 * cc + 3] + bv.w)};
          *reinterpret_cast<u32x2*>(slice + r * MF_SLICE_STRIDE + (32 * tt + 8 * cc + 4 * hh) * 2) = o;
        }
      }
      if (np == 2) request(3);   // into the accumulator registers chunks 0 .. 2 have just freed
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // load k = 8 (np % 3) + 4 h2 + i of its batch; batch 1 is requested in front of chunk 2's waits: its 24 loads
          // are younger than every load of batch 0 still outstanding then
          if (np < 2) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xh[np][h2][i]) : "n"(23 - (8 * np + 4 * h2 + i)));
          else if (np == 2) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xh[np][h2][i]) : "n"(24 + 23 - (16 + 4 * h2 + i)));
          else asm volatile("s_waitcnt vmcnt(%1)" : "+a"(xh[np][h2][i]) : "n"(23 - (8 * (np - 3) + 4 * h2 + i)));
          f32x4 v = xh[np][h2][i];
          const u32x2 dv = *reinterpret_cast<const u32x2*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + h2 * 64 + cchunk * 8);
          v[0] += bf2f((unsigned short)(dv[0] & 0xffff));
          v[1] += bf2f((unsigned short)(dv[0] >> 16));
          v[2] += bf2f((unsigned short)(dv[1] & 0xffff));
          v[3] += bf2f((unsigned short)(dv[1] >> 16));
          if (np == 0 && h2 == 0) {
            float t = (v[0] + v[1]) + (v[2] + v[3]);
            t = mf_sum8(t);
            cshift[i] = t * (1.0f / 32.0f);
          }
          v[0] -= cshift[i]; v[1] -= cshift[i]; v[2] -= cshift[i]; v[3] -= cshift[i];
          s1[i] += (v[0] + v[1]) + (v[2] + v[3]);
          s2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          xh[np][h2][i] = v;      // x_fin - shift, kept until the statistics are complete
        }
    }
    float rstd[4], mean_s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = s1[i], b = s2[i];
      a = mf_sum8(a); b = mf_sum8(b);
      mean_s[i] = a * (1.0f / MF_D);
      rstd[i] = rsqrtf(fmaxf(b * (1.0f / MF_D) - mean_s[i] * mean_s[i], 0.f) + 1e-5f);
    }
#pragma unroll
    for (int np = 0; np < MF_KT; ++np)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int col = 64 * np + 32 * h2 + 4 * cchunk;
        const float4 g4 = *reinterpret_cast<const float4*>(lds_u + col);
        const float4 b4 = *reinterpret_cast<const float4*>(lds_u + MF_D + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = xh[np][h2][i];
          const f32x4 y = {fmaf((v[0] - mean_s[i]) * rstd[i], g4.x, b4.x), fmaf((v[1] - mean_s[i]) * rstd[i], g4.y, b4.y),
                           fmaf((v[2] - mean_s[i]) * rstd[i], g4.z, b4.z), fmaf((v[3] - mean_s[i]) * rstd[i], g4.w, b4.w)};
          // rows past M are clamped duplicates of row M - 1: every duplicate stores the same value (y_out != x_out)
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(y_out) + roff[i] + (64 * np + 32 * h2) * 4) = y;
        }
      }
  } else if constexpr (!QKV) {
    mf_wait_vmcnt<0>();   // the re-reads issued past the end
    // ---- epilogue: + b2 -> bf16 -> wave-private LDS transpose -> whole-line stores
#pragma unroll
    for (int np = 0; np < MF_OT / 2; ++np) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * np + tt;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int nl = 32 * t + 8 * cc + 4 * hh;
          const float4 bv = *reinterpret_cast<const float4*>(lds_b2 + nl);
          u32x2 o = {pack2bf(oacc[t][4 * cc] + bv.x, oacc[t][4 * cc + 1] + bv.y),
                     pack2bf(oacc[t][4 * cc + 2] + bv.z, oacc[t][4 * cc + 3] + bv.w)};
          *reinterpret_cast<u32x2*>(slice + r * MF_SLICE_STRIDE + (32 * tt + 8 * cc + 4 * hh) * 2) = o;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 u = *reinterpret_cast<const u32x4*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + cchunk * 16);
        const long orow = m_base + 8 * i + crow;
        *reinterpret_cast<u32x4*>(C + orow * MF_D + np * 64 + 8 * cchunk) = u;
      }
    }
  } else {
    // ---- epilogue == prologue of the next layer's LN1 + QKV GEMM.  Per 64-column chunk np: the fc2 output (+ b2,
    // rounded to bf16 exactly like the stand-alone kernel's delta) goes through the wave-private slice into row
    // order, x_next = x_new + out is formed from whole-line re-reads of x_new (written by this wave in the
    // prologue), written to x_next (== x_out, i.e. IN PLACE, on the inference path; a buffer of its own on the training
    // path, which keeps x_new as x_mid), and shifted / measured / packed into the A fragments of k-tile np.
    // The tiles already in flight are older than these loads: hipcc's own vmcnt waits retire them first.
    if constexpr (XACC) {
      // x_next = O + (b2 + bo), stored from the accumulator layout (48 stores per lane, all lanes live: the first two tail
      // tiles count them), LN1 statistics and the q / k / v operand from the registers
      seam_x(lds_b2, x_next);
    } else if constexpr (!LNQ) {   // the seam (MODE 2: the prologue already produced the normalised operand of LN1)
      seam(x_out, x_next, lds_b2, [] {});
#if GWW_MF_NORM
    normalise_af();
#else
      // (round 1's per-value LayerNorm algebra reads u and cb of the panel from the fc1 tables' place)
      for (int i = tid; i < NQ; i += MF_THREADS) { lds_qcb[i] = q_cb[i]; lds_u[i] = q_u[i]; }
      mf_wait_vmcnt<0>();
#endif
    } else {
      mf_wait_vmcnt<0>();             // MODE 2: every tile issued so far has landed (this wave's pieces)
    }
    // Behind the seam no counted wait is needed: its loads were issued after the DMA pieces of the first two tail tiles and
    // have been consumed, so those pieces have landed (issue-order retirement), and its last x_next stores stay in flight.
    __builtin_amdgcn_s_barrier();     // everybody's pieces visible

    MSTAMP(5);
#if GWW_MF_TAIL2
    // ---- second GEMM: qkv[32 rows, NQ] = LN1(x_next) Wqkv'^T as NQ / 64 column CHUNKS of three fc1-format tiles
    // ([64 n][128 k], gww_mlp_pack_bf16).  Chunk c accumulates into the accumulator pair sacc[2 (c & 1)], [.. + 1] (preloaded with
    // its folded bias); in the MFMA gaps of its 48 MFMAs ride, for the OTHER pair: the epilogue of chunk c - 1 (gaps 4 .. 11:
    // pack + ds_write_b64 of the eight 4-column pieces; 22 .. 25: the four transposed row reads; 37 .. 40: the four 16-byte
    // stores, behind the tile's LDS-DMA pieces) and the bias preload of chunk c + 1 (gaps 12 .. 19).  Nothing of a chunk's
    // epilogue is in the open any more except the last one's.
    // vmcnt: a chunk's four stores are issued in its third tile BEHIND that tile's four DMA pieces, so for the ring waits
    // at the top of the next chunk's first and second tile they are younger than the awaited pieces and may stay in
    // flight (+ 4); at the third tile they are older and long retired.
    {
    const int nchunks = NQ / 64;
    u32x4 tq[4];                      // the four transposed row pieces of the chunk being stored
#pragma unroll
    for (int i = 0; i < 4; ++i) tq[i] = u32x4{0u, 0u, 0u, 0u};
    auto q_preload = [&](int t, int chunk, int q) {   // folded bias of chunk `chunk` -> accumulator sacc[t], registers 4 q ..
      const float4 bv = *reinterpret_cast<const float4*>(lds_qcb + 64 * chunk + 32 * (t & 1) + 8 * q + 4 * hh);
      sacc[t][4 * q] = bv.x; sacc[t][4 * q + 1] = bv.y; sacc[t][4 * q + 2] = bv.z; sacc[t][4 * q + 3] = bv.w;
    };
    // one gap of the ride: G = 0 .. 47 within the chunk; Q = the pair that rides (the other one), chunk = the chunk being computed
    auto q_ride = [&](auto q_c, auto g_c, int chunk, auto mode_c) {
      constexpr int Q = decltype(q_c)::value, G = decltype(g_c)::value, RM = decltype(mode_c)::value;   // 1: epilogue + preload, 2: preload only
      if constexpr (RM == 1 && G >= 4 && G < 12) {
        constexpr int tt = (G - 4) >> 2, cc = (G - 4) & 3;
        const f32x16& a = sacc[2 * Q + tt];
        u32x2 o = {pack2bf(a[4 * cc], a[4 * cc + 1]), pack2bf(a[4 * cc + 2], a[4 * cc + 3])};
        *reinterpret_cast<u32x2*>(slice + r * MF_SLICE_STRIDE + (32 * tt + 8 * cc + 4 * hh) * 2) = o;
      }
      if constexpr (RM >= 1 && G >= 12 && G < 20) {
        if (chunk + 1 < nchunks) q_preload(2 * Q + ((G - 12) >> 2), chunk + 1, (G - 12) & 3);
      }
      if constexpr (RM == 1 && G >= 22 && G < 26) {
        tq[G - 22] = *reinterpret_cast<const u32x4*>(slice + (8 * (G - 22) + crow) * MF_SLICE_STRIDE + cchunk * 16);
        asm volatile("" : "+v"(tq[G - 22]));
      }
      if constexpr (RM == 1 && G >= 37 && G < 41) {
        const long orow = m_base + 8 * (G - 37) + crow;
        if (!(GWW_MF_ABL & 1))
          __builtin_nontemporal_store(tq[G - 37], reinterpret_cast<u32x4*>(q_out + orow * NQ + (chunk - 1) * 64 + 8 * cchunk));
      }
    };
    // one tile of the tail: FLAT = tile index within the tail (its ring stage follows), PAR = chunk parity, IDX3 = tile of the
    // chunk, RM = what rides (0 nothing), EXTRA = stores the ring wait may leave in flight, NOWAIT = no ring wait (behind the seam)
    auto run_qtile = [&](auto flat_c, auto par_c, auto idx3_c, auto mode_c, int chunk, auto extra_c, auto nowait_c) {
      constexpr int FLAT = decltype(flat_c)::value, PAR = decltype(par_c)::value, IDX3 = decltype(idx3_c)::value;
      constexpr int RM = decltype(mode_c)::value, EXTRA = decltype(extra_c)::value;
      constexpr bool NOWAIT = decltype(nowait_c)::value != 0;
      constexpr int ST = (12 % MF_NST == 0) ? ((FLAT + OP_TILES) % MF_NST) : -1;   // (6 F / 64 MLP tiles in front: a multiple of 4)
      if (ST >= 0) stage = ST;
      if (!(GWW_MF_ABL & 4)) {
        if (!NOWAIT) mf_wait_vmcnt<MF_GL*(MF_AHEAD - 2) + ((GWW_MF_ABL & 3) ? 0 : EXTRA)>();
      }
      if (!(GWW_MF_ABL & 8)) __builtin_amdgcn_s_barrier();
      TSTAMP(6);
      const int dma_tile = it + MF_AHEAD < total ? it + MF_AHEAD : total - 1;
      const int dma_stage = stage + MF_AHEAD >= MF_NST ? stage + MF_AHEAD - MF_NST : stage + MF_AHEAD;
      const int stage_next = stage + 1 == MF_NST ? 0 : stage + 1;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        bf16x8(&cur)[4] = wf[sub & 1];
        bf16x8(&nxt)[4] = wf[(sub + 1) & 1];
        const unsigned char* Wn = lds + (sub == 3 ? stage_next : stage) * MF_TILE;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tl = u & 1, ks = 2 * sub + (u >> 1);
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[2 * PAR + tl]) : "v"(cur[u]), "v"(af[8 * IDX3 + ks]));
          if (u < 3) asm volatile("" : "+v"(cur[u + 1]));
          else if (sub < 3) asm volatile("" : "+v"(nxt[0]));
          {
            const int off = sub == 3 ? first_off(0, u) : (u & 1) * 8192 + off1[2 * (sub + 1) + (u >> 1)];
            nxt[u] = *reinterpret_cast<const bf16x8*>(Wn + off);
          }
          if constexpr (RM != 0) {
            if (sub == 0 && u == 0) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 0>{}, chunk, mode_c);
            else if (sub == 0 && u == 1) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 1>{}, chunk, mode_c);
            else if (sub == 0 && u == 2) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 2>{}, chunk, mode_c);
            else if (sub == 0 && u == 3) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 3>{}, chunk, mode_c);
            else if (sub == 1 && u == 0) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 4>{}, chunk, mode_c);
            else if (sub == 1 && u == 1) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 5>{}, chunk, mode_c);
            else if (sub == 1 && u == 2) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 6>{}, chunk, mode_c);
            else if (sub == 1 && u == 3) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 7>{}, chunk, mode_c);
            else if (sub == 2 && u == 0) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 8>{}, chunk, mode_c);
            else if (sub == 2 && u == 1) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 9>{}, chunk, mode_c);
            else if (sub == 2 && u == 2) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 10>{}, chunk, mode_c);
            else if (sub == 2 && u == 3) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 11>{}, chunk, mode_c);
            else if (sub == 3 && u == 0) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 12>{}, chunk, mode_c);
            else if (sub == 3 && u == 1) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 13>{}, chunk, mode_c);
            else if (sub == 3 && u == 2) q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 14>{}, chunk, mode_c);
            else q_ride(MF_FL<1 - PAR>{}, MF_FL<16 * IDX3 + 15>{}, chunk, mode_c);
          }
          // the tile's four LDS-DMA pieces: in the gaps the main loop uses (IDX3 0: 9 .. 12, 1: 5 .. 8, 2: 1 .. 4)
          if (!(GWW_MF_ABL & 4)) {
            const int piece = 4 * sub + u - (9 - 4 * IDX3);
            if (piece >= 0 && piece < 4) issue_piece(dma_tile, dma_stage, piece);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stage = stage_next;
      ++it;
      TSTAMP(7);
    };
    using NW = std::conditional_t<LNQ || (GWW_MF_ABL & 32), I0, I1>;   // behind the seam the first two tiles have landed (above)
    if (LNQ) {   // first-step fragments of the first tile (the MLP stream's last tile prefetches them otherwise)
#pragma unroll
      for (int u = 0; u < 4; ++u) wf[0][u] = *reinterpret_cast<const bf16x8*>(lds + stage * MF_TILE + first_off(0, u));
    }
    // bias of chunk 0 (in the open, once per panel)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) q_preload(t, 0, q);
    // chunk 0: nothing to store yet; the bias of chunk 1 rides
    run_qtile(MF_FL<0>{}, I0{}, I0{}, I2{}, 0, I0{}, NW{});
    run_qtile(MF_FL<1>{}, I0{}, I1{}, I2{}, 0, I0{}, NW{});
    run_qtile(MF_FL<2>{}, I0{}, I2{}, I2{}, 0, I0{}, I0{});
    // chunk 1: epilogue of chunk 0 rides (no stores of an earlier chunk in the queue yet)
    run_qtile(MF_FL<3>{}, I1{}, I0{}, I1{}, 1, I0{}, I0{});
    run_qtile(MF_FL<4>{}, I1{}, I1{}, I1{}, 1, I0{}, I0{});
    run_qtile(MF_FL<5>{}, I1{}, I2{}, I1{}, 1, I0{}, I0{});
    using I4 = std::integral_constant<int, 4>;
    for (int c = 2; c < nchunks; c += 4) {
      run_qtile(MF_FL<6>{}, I0{}, I0{}, I1{}, c, I4{}, I0{});
      run_qtile(MF_FL<7>{}, I0{}, I1{}, I1{}, c, I4{}, I0{});
      run_qtile(MF_FL<8>{}, I0{}, I2{}, I1{}, c, I0{}, I0{});
      run_qtile(MF_FL<9>{}, I1{}, I0{}, I1{}, c + 1, I4{}, I0{});
      run_qtile(MF_FL<10>{}, I1{}, I1{}, I1{}, c + 1, I4{}, I0{});
      run_qtile(MF_FL<11>{}, I1{}, I2{}, I1{}, c + 1, I0{}, I0{});
      run_qtile(MF_FL<12>{}, I0{}, I0{}, I1{}, c + 2, I4{}, I0{});
      run_qtile(MF_FL<13>{}, I0{}, I1{}, I1{}, c + 2, I4{}, I0{});
      run_qtile(MF_FL<14>{}, I0{}, I2{}, I1{}, c + 2, I0{}, I0{});
      run_qtile(MF_FL<15>{}, I1{}, I0{}, I1{}, c + 3, I4{}, I0{});
      run_qtile(MF_FL<16>{}, I1{}, I1{}, I1{}, c + 3, I4{}, I0{});
      run_qtile(MF_FL<17>{}, I1{}, I2{}, I1{}, c + 3, I0{}, I0{});
    }
    // the last chunk's epilogue, in the open (its pair: parity of nchunks - 1 = 1)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sacc[2]), "+v"(sacc[3]));   // asm MFMA -> VALU read
    if (!(GWW_MF_ABL & 2)) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const f32x16& a = sacc[2 + tt];
          u32x2 o = {pack2bf(a[4 * cc], a[4 * cc + 1]), pack2bf(a[4 * cc + 2], a[4 * cc + 3])};
          *reinterpret_cast<u32x2*>(slice + r * MF_SLICE_STRIDE + (32 * tt + 8 * cc + 4 * hh) * 2) = o;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 u = *reinterpret_cast<const u32x4*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + cchunk * 16);
        const long orow = m_base + 8 * i + crow;
        if (GWW_MF_ABL & 1) { asm volatile("" :: "v"(u)); continue; }
        __builtin_nontemporal_store(u, reinterpret_cast<u32x4*>(q_out + orow * NQ + (nchunks - 1) * 64 + 8 * cchunk));
      }
    }
    // the ring's re-reads issued past the end must have landed before this workgroup's LDS is handed on; they are older
    // than the last two chunks' 8 output stores, which need not be waited for (vmcnt counts in issue order)
    mf_wait_vmcnt<(GWW_MF_ABL & 3) ? 0 : 8>();
    }
#else
    // ---- second GEMM: qkv[32 rows, NQ] = LN1(x_next) Wqkv'^T, n-tiles of 128 columns, 6 k-tiles each
    const int T0 = OP_TILES + 6 * nck;
    if (LNQ) {   // first-step fragments of the first tile (the MLP stream's last tile prefetches them otherwise)
#pragma unroll
      for (int u = 0; u < 4; ++u) wf[0][u] = *reinterpret_cast<const bf16x8*>(lds + stage * MF_TILE + u * 4096 + offq[0]);
    }
    f32x16 acc[4];
    for (int nt = 0; nt < NQ / 128; ++nt) {
#if GWW_MF_SCHED
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float4 bv = *reinterpret_cast<const float4*>(lds_qcb + 128 * nt + 32 * t + 8 * cc + 4 * hh);
          acc[t][4 * cc] = bv.x; acc[t][4 * cc + 1] = bv.y; acc[t][4 * cc + 2] = bv.z; acc[t][4 * cc + 3] = bv.w;
        }
#endif
#pragma unroll
      for (int kt = 0; kt < MF_KT; ++kt) {
        // the DMA group being waited for (tile it + 1) was issued during tile it + 1 - AHEAD; the 8 output stores of the
        // previous n-tile were issued in front of (nt, 0): they are YOUNGER than that group while kt <= AHEAD - 2 and may
        // then stay in flight
        constexpr int TAIL_ST = (GWW_MF_ABL & 3) ? 0 : 8;   // output stores per n-tile in the vmcnt queue
        if (GWW_MF_ABL & 4) {}
        else if (XACC && nt == 0 && kt <= MF_AHEAD - 2) mf_wait_vmcnt<MF_GL*(MF_AHEAD - 2) + ((GWW_MF_ABL & 16) ? 0 : 4 * MF_OT)>();   // the seam's 48 x_next stores are younger than the awaited pieces
        else if (!LNQ && !(GWW_MF_ABL & 32) && nt == 0 && kt <= MF_AHEAD - 2) {}   // landed before the seam's loads (above)
        else if (nt > 0 && kt <= MF_AHEAD - 2) mf_wait_vmcnt<MF_GL*(MF_AHEAD - 2) + TAIL_ST>();
        else mf_wait_vmcnt<MF_GL*(MF_AHEAD - 2)>();
        if (!(GWW_MF_ABL & 8)) __builtin_amdgcn_s_barrier();
        TSTAMP(6);
        const int itq = T0 + nt * MF_KT + kt;
        const int dma_tile = itq + MF_AHEAD < total ? itq + MF_AHEAD : total - 1;
        const int dma_stage = stage + MF_AHEAD >= MF_NST ? stage + MF_AHEAD - MF_NST : stage + MF_AHEAD;
        const int stage_next = stage + 1 == MF_NST ? 0 : stage + 1;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int q = 4 * kt + sub;
          bf16x8(&cur)[4] = wf[q & 1];
          bf16x8(&nxt)[4] = wf[(q + 1) & 1];
          const unsigned char* Wn = lds + (sub == 3 ? stage_next : stage) * MF_TILE + offq[(sub + 1) & 3];
          __builtin_amdgcn_sched_barrier(0);
          if (!(GWW_MF_ABL & 4)) issue_piece(dma_tile, dma_stage, sub);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
#if GWW_MF_SCHED
            // accumulators in architectural registers, preloaded with the folded bias (below): the n-tile epilogue is a
            // pack + store, no v_accvgpr_read_b32 (8 cycles each in the open), no bias read / add per value.  asm: see
            // the main loop; two wait states in front cover register copies hipcc may place there (ISA audit in build())
            asm volatile(MF_PADNOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(cur[u]), "v"(af[4 * kt + sub]));
#else
            if (kt == 0 && sub == 0) {
              f32x16 z;
#pragma unroll
              for (int j = 0; j < 16; ++j) z[j] = 0.f;
              acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[u], af[4 * kt + sub], z, 0, 0, 0);
            } else {
              acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[u], af[4 * kt + sub], acc[u], 0, 0, 0);
            }
#endif
            nxt[u] = *reinterpret_cast<const bf16x8*>(Wn + u * 4096);
          }
#if !GWW_MF_SCHED   // (asm MFMAs and the fragment reads keep their program order without hints)
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage_next;
        TSTAMP(7);
      }
      // n-tile epilogue: LayerNorm algebra + bias -> bf16 -> slice transpose -> whole-line stores (8 per wave)
#if GWW_MF_SCHED
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));   // asm MFMA -> VALU read
#endif
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (GWW_MF_ABL & 2) break;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * half + tt;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int nl = 128 * nt + 32 * t + 8 * cc + 4 * hh;
#if GWW_MF_SCHED
            const float v0 = acc[t][4 * cc], v1 = acc[t][4 * cc + 1], v2 = acc[t][4 * cc + 2], v3 = acc[t][4 * cc + 3];
            (void)nl;
#elif GWW_MF_NORM
            const float4 bv = *reinterpret_cast<const float4*>(lds_qcb + nl);
            const float v0 = acc[t][4 * cc] + bv.x, v1 = acc[t][4 * cc + 1] + bv.y;
            const float v2 = acc[t][4 * cc + 2] + bv.z, v3 = acc[t][4 * cc + 3] + bv.w;
#else
            const float4 bv = *reinterpret_cast<const float4*>(lds_qcb + nl);
            const float4 uv = *reinterpret_cast<const float4*>(lds_u + nl);
            const float v0 = fmaf(row_rstd, fmaf(-row_mean, uv.x, acc[t][4 * cc]), bv.x);
            const float v1 = fmaf(row_rstd, fmaf(-row_mean, uv.y, acc[t][4 * cc + 1]), bv.y);
            const float v2 = fmaf(row_rstd, fmaf(-row_mean, uv.z, acc[t][4 * cc + 2]), bv.z);
            const float v3 = fmaf(row_rstd, fmaf(-row_mean, uv.w, acc[t][4 * cc + 3]), bv.w);
#endif
            u32x2 o = {pack2bf(v0, v1), pack2bf(v2, v3)};
            *reinterpret_cast<u32x2*>(slice + r * MF_SLICE_STRIDE + (32 * tt + 8 * cc + 4 * hh) * 2) = o;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 u = *reinterpret_cast<const u32x4*>(slice + (8 * i + crow) * MF_SLICE_STRIDE + cchunk * 16);
          const long orow = m_base + 8 * i + crow;
          if (GWW_MF_ABL & 1) { asm volatile("" :: "v"(u)); continue; }
#if GWW_MF_NTSTORE
          __builtin_nontemporal_store(u, reinterpret_cast<u32x4*>(q_out + orow * NQ + nt * 128 + 64 * half + 8 * cchunk));
#else
          *reinterpret_cast<u32x4*>(q_out + orow * NQ + nt * 128 + 64 * half + 8 * cchunk) = u;
#endif
        }
      }
    }
    // the ring's re-reads issued past the end must have landed before this workgroup's LDS is handed on; they are older
    // than the last n-tile's 8 output stores, which need not be waited for (vmcnt counts in issue order)
    mf_wait_vmcnt<(GWW_MF_ABL & 3) ? 0 : 8>();
#endif
  }
  MSTAMP(3);
  MSTAMP_FLUSH
}

// bf16 x 2^k is exact (no mantissa change) short of the exponent range's ends: the rescaled panels carry the same
// information and every product of the rescaled GEMMs equals the unscaled one times an exact power of two.
__device__ __forceinline__ u32x4 mf_scale_bf16x8(u32x4 v, float f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = __uint_as_float(v[j] << 16) * f, hi = __uint_as_float(v[j] & 0xffff0000u) * f;
    v[j] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
  }
  return v;
}

// Pre-tile the two weight panels of the block into the stream k_mlp_fused consumes:
//   out[c][idx] = one 16-KiB LDS image
//   idx < 6 : nh = idx / 3, kt3 = idx % 3: [64 rows][128 k] = W1'[128 c + 64 nh + row][128 kt3 + k] (gain-folded
//             fc1, [F, 384]); 16-byte chunks XOR-swizzled by row & 15
//   idx >= 6: i2 = idx - 6, kh = i2 / 3, ng = i2 % 3: [128 rows][64 k] = W2[128 ng + row][128 c + 64 kh + swap23(k)];
//             chunks XOR-swizzled by (row >> 1) & 7
// swap23 exchanges bits 2 and 3 of k: the operand order of the accumulator-as-operand product (header).
// Optionally followed by the NEXT layer's LayerNorm-folded q / k / v panel (k_mlp_fused<true>): n-tile major,
// six [128 n][64 k] images per n-tile, chunks XOR-swizzled by (row >> 1) & 7.
__global__ __launch_bounds__(256) void k_mlp_pack(const unsigned short* __restrict__ w1,
                                                  const unsigned short* __restrict__ w2,
                                                  const unsigned short* __restrict__ wq,
                                                  unsigned short* __restrict__ out, int F, long n_chunks16,
                                                  const unsigned short* __restrict__ wo) {
  const int nck = F / 64, mlp_tiles = 6 * nck, op_tiles = wo ? 3 * MF_KT : 0;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n_chunks16; g += (long)gridDim.x * 256) {
    const int tile_all = (int)(g >> 10), within = (int)(g & 1023);
    const int row = within >> 3, chunk = (within & 7) ^ ((row >> 1) & 7);
    const int tile = tile_all - op_tiles;
    u32x4 v;
    if (tile < 0) {
      // W_o [384, 384] in front of the stream (OP mode): n-tile major, 6 k-tiles each, [128 n][64 k] images
      const int nt = tile_all / MF_KT, kt = tile_all - nt * MF_KT;
      v = *reinterpret_cast<const u32x4*>(wo + (long)(128 * nt + row) * MF_D + 64 * kt + 8 * chunk);
    } else if (tile >= mlp_tiles) {
#if GWW_MF_TAIL2
      // appended LN1-folded q / k / v panel [NQ, 384] in the fc1 tile format: 64-column chunk cq, three [64 n][128 k] tiles,
      // 16 chunks per row, chunk ch stored at ch ^ (row & 15)
      const int it2 = tile - mlp_tiles, cq = it2 / 3, k3 = it2 - 3 * cq;
      const int row2 = within >> 4, ch = (within & 15) ^ (row2 & 15);
      v = *reinterpret_cast<const u32x4*>(wq + (long)(64 * cq + row2) * MF_D + 128 * k3 + 8 * ch);
#else
      // appended LN1-folded q / k / v panel [NQ, 384]: n-tile major, 6 k-tiles each, [128 n][64 k] images
      const int it2 = tile - mlp_tiles, nt = it2 / MF_KT, kt = it2 - nt * MF_KT;
      v = *reinterpret_cast<const u32x4*>(wq + (long)(128 * nt + row) * MF_D + 64 * kt + 8 * chunk);
#endif
    } else {
      // stream order of k_mlp_fused:  G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(n-1) G2(n-2) | G2(n-1)
      int kind, cp, idx3;
      if (tile < 3) { kind = 0; cp = 0; idx3 = tile; }
      else {
        const int k = tile - 3, blk = k / 6, w = k - 6 * blk;
        if (blk == nck - 1) { kind = 1; cp = blk; idx3 = w; }
        else if (w < 3) { kind = 0; cp = blk + 1; idx3 = w; }
        else { kind = 1; cp = blk; idx3 = w - 3; }
      }
      if (kind == 0) {
        // fc1 tile [64 n][128 k]: 16 chunks per row, chunk ch stored at ch ^ (row & 15); k-third idx3
        const int row2 = within >> 4, ch = (within & 15) ^ (row2 & 15);
        v = *reinterpret_cast<const u32x4*>(w1 + (long)(64 * cp + row2) * MF_D + 128 * idx3 + 8 * ch);
        if (GWW_MF_SCHED) v = mf_scale_bf16x8(v, 0.125f);   // S / 8 in the accumulators (header of gelu_slice)
      } else {
        // fc2 tile [128 n2][64 k]: n-group idx3, k = the 64 ffn columns of chunk cp, bits 2 / 3 of k swapped
        const unsigned short* src = w2 + (long)(128 * idx3 + row) * F + 64 * cp;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int p = 8 * chunk + j;
          e[j] = src[(p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1)];
        }
        v = u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                  (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
        if (GWW_MF_SCHED) v = mf_scale_bf16x8(v, 8.0f);     // ... and gelu / 8 in the fc2 operand: fc2 x 8 restores it
      }
    }
    reinterpret_cast<u32x4*>(out)[g] = v;
  }
}

// w1_folded bf16 [F, 384], w2 bf16 [384, F] (+ wqkv_folded bf16 [NQ, 384] or NULL) -> out bf16,
// 2 * 384 * F (+ NQ * 384) elements
int launch_mlp_pack(const void* w1_folded, const void* w2, const void* wqkv_folded, void* out, int d, int F, int NQ,
                    hipStream_t s, const void* wo) {
  GWW_REQUIRE(d == MF_D && F % 128 == 0 && F >= 0 && (F > 0 || wqkv_folded),
              "mlp_pack: d must be 384 and ffn a multiple of 128 (0: only the q / k / v panel, for launch_lnqkv_fused)");
  GWW_REQUIRE(!wqkv_folded || (NQ > 0 && NQ % 128 == 0), "mlp_pack: the q / k / v panel needs NQ %% 128 == 0");
  GWW_REQUIRE(!wo || F > 0, "mlp_pack: an out_proj panel goes in front of an MLP stream");
  const long n16 = ((wo ? (long)MF_D * MF_D : 0) + 2L * MF_D * F + (wqkv_folded ? (long)NQ * MF_D : 0)) / 8;
  hipLaunchKernelGGL(k_mlp_pack, dim3((unsigned)cdiv(n16, 256)), dim3(256), 0, s, (const unsigned short*)w1_folded,
                     (const unsigned short*)w2, (const unsigned short*)wqkv_folded, (unsigned short*)out, F, n16,
                     (const unsigned short*)wo);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// x fp32 [M, 384], delta bf16 [M, 384], x_out fp32 [M, 384] (!= x); Wt = launch_mlp_pack of the gain-folded fc1
// panel and the fc2 panel, ln_u / ln_cb from gww_ln_fold_weights; C bf16 [>= roundup(M, 128), 384] (whole
// 128-row panels are stored).
// bo != NULL (OP): `delta` is the attention context ctx (bf16 [M, 384]) and Wt starts with the W_o tiles
// (launch_mlp_pack(..., wo)): x_out = x + bf16(ctx W_o^T + bo), then the block as above.
int launch_mlp_fused(const float* x, const void* delta, float* x_out, const float* ln_u, const float* ln_cb,
                     const void* Wt, const float* b2, void* C, long M, int d, int F, hipStream_t s,
                     const float* q_u, const float* q_cb, void* q_out, int NQ, float* x_next_out, const float* bo,
                     bool keep_x_new = true) {
  GWW_REQUIRE(x && delta && x_out && ln_u && ln_cb && Wt && b2, "mlp_fused: NULL operand");
  GWW_REQUIRE(d == MF_D, "mlp_fused: built for d_model = 384 (got %d)", d);
  GWW_REQUIRE(F % 128 == 0 && F > 0 && F <= MF_FMAX, "mlp_fused: ffn = %d must be a multiple of 128, <= 1536", F);
  GWW_REQUIRE((const void*)x_out != (const void*)x, "mlp_fused: x_out must not alias x");
  GWW_REQUIRE(M * (long)(MF_D * 4) < (1L << 32), "mlp_fused: M = %ld rows exceed the 32-bit row offsets of the seams", M);
  const bool qkv = q_out != nullptr;
  GWW_REQUIRE(qkv || C, "mlp_fused: no output");
  GWW_REQUIRE(!qkv || (q_u && q_cb && NQ > 0 && NQ % 128 == 0 && NQ <= MF_FMAX),
              "mlp_fused: the fused q / k / v projection needs u, cb and NQ %% 128 == 0, NQ <= 1536");
  GWW_REQUIRE(((((uintptr_t)x) | ((uintptr_t)delta) | ((uintptr_t)x_out) | ((uintptr_t)Wt) | ((uintptr_t)C) |
                ((uintptr_t)q_out)) & 15) == 0, "mlp_fused: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  const long panels = cdiv(M, MF_BM);
  // stagger only when there is more than one round of workgroups to keep de-phased; 100 MHz ticks
  static const int stagger_env = (int)lab_int("GWW_MLP_STAGGER", 0);   // (lab build; measured: no effect)
  const int stagger = panels >= 512 ? stagger_env : 0;
#define GWW_MF_LAUNCH(QQ, OO, ...)                                                                                        \
  hipLaunchKernelGGL((k_mlp_fused<QQ, OO>), dim3((unsigned)panels), dim3(MF_THREADS), 0, s, x, (const unsigned short*)delta, \
                     x_out, ln_u, ln_cb, (const unsigned short*)Wt, b2, (unsigned short*)C, M, F, stagger, __VA_ARGS__, bo, keep_x_new ? 1 : 0)
  if (qkv) {
    // x_next goes to its own buffer (training: the saved activations) or back over x, whose rows each workgroup has
    // finished reading long before it writes them; never over x_out: the seam's unmasked stores of the clamped rows past M
    // are only harmless while it does not run in place (k_mlp_fused, seam)
    float* xnx = x_next_out ? x_next_out : const_cast<float*>(x);
    GWW_REQUIRE((((uintptr_t)xnx) & 15) == 0 && (const void*)xnx != (const void*)x_out, "mlp_fused: x_next must not alias x_out");
    if (bo) GWW_MF_LAUNCH(1, true, q_u, q_cb, (unsigned short*)q_out, NQ, xnx);
    else GWW_MF_LAUNCH(1, false, q_u, q_cb, (unsigned short*)q_out, NQ, xnx);
  } else {
    if (bo) GWW_MF_LAUNCH(0, true, nullptr, nullptr, nullptr, 0, nullptr);
    else GWW_MF_LAUNCH(0, false, nullptr, nullptr, nullptr, 0, nullptr);
  }
#undef GWW_MF_LAUNCH
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// The LAST block of the encoder with the final LayerNorm as its epilogue (k_mlp_fused<3, true>): ctx bf16 [M, 384] and x fp32
// [M, 384] in, y fp32 [M, 384] = LayerNorm_final(x + bf16(ctx W_o^T + bo) + bf16(mlp(...) + b2)) out; x_mid (fp32 [M, 384],
// != x, != y) receives the block's intermediate residual stream.  Wt = launch_mlp_pack(w1_folded, w2, NULL, ., wo).
int launch_mlp_fused_final(const float* x, const void* ctx, float* x_mid, const float* ln_u, const float* ln_cb,
                           const void* Wt, const float* b2, const float* bo, const float* lnf_w, const float* lnf_b, float* y,
                           long M, int d, int F, hipStream_t s, bool keep_x_new = true) {
  GWW_REQUIRE(x && ctx && x_mid && ln_u && ln_cb && Wt && b2 && bo && lnf_w && lnf_b && y, "mlp_fused_final: NULL operand");
  GWW_REQUIRE(d == MF_D, "mlp_fused_final: built for d_model = 384 (got %d)", d);
  GWW_REQUIRE(F % 128 == 0 && F > 0 && F <= MF_FMAX, "mlp_fused_final: ffn = %d must be a multiple of 128, <= 1536", F);
  GWW_REQUIRE((const void*)x_mid != (const void*)x && (const void*)y != (const void*)x_mid, "mlp_fused_final: x_mid must alias neither x nor y");
  GWW_REQUIRE(M * (long)(MF_D * 4) < (1L << 32), "mlp_fused_final: M = %ld rows exceed the 32-bit row offsets of the seams", M);
  GWW_REQUIRE(((((uintptr_t)x) | ((uintptr_t)ctx) | ((uintptr_t)x_mid) | ((uintptr_t)Wt) | ((uintptr_t)y)) & 15) == 0,
              "mlp_fused_final: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  hipLaunchKernelGGL((k_mlp_fused<3, true>), dim3((unsigned)cdiv(M, MF_BM)), dim3(MF_THREADS), 0, s, x, (const unsigned short*)ctx,
                     x_mid, ln_u, ln_cb, (const unsigned short*)Wt, b2, reinterpret_cast<unsigned short*>(y), M, F, 0, lnf_w,
                     lnf_b, (unsigned short*)nullptr, 0, (float*)nullptr, bo, keep_x_new ? 1 : 0);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// LayerNorm + q / k / v projection of a residual stream WITHOUT a pending delta (layer 0): q_out bf16 [>= roundup(M, 128),
// NQ] = LN(x) Wqkv'^T + cb with the LayerNorm folded into Wt = launch_mlp_pack(NULL, NULL, wqkv_folded, ., 384, 0, NQ)
// and q_u / q_cb (gww_ln_fold_weights).  x is only read.
int launch_lnqkv_fused(const float* x, const float* q_u, const float* q_cb, const void* Wt, void* q_out, long M, int d,
                       int NQ, hipStream_t s) {
  GWW_REQUIRE(x && q_u && q_cb && Wt && q_out, "lnqkv_fused: NULL operand");
  GWW_REQUIRE(d == MF_D, "lnqkv_fused: built for d_model = 384 (got %d)", d);
  GWW_REQUIRE(NQ > 0 && NQ % 128 == 0 && NQ <= MF_FMAX, "lnqkv_fused: NQ = %d must be a multiple of 128, <= 1536", NQ);
  GWW_REQUIRE(((((uintptr_t)x) | ((uintptr_t)Wt) | ((uintptr_t)q_out)) & 15) == 0, "lnqkv_fused: operands must be 16-byte aligned");
  if (M == 0) return GWW_OK;
  const long panels = cdiv(M, MF_BM);
  hipLaunchKernelGGL((k_mlp_fused<2, false>), dim3((unsigned)panels), dim3(MF_THREADS), 0, s, x, (const unsigned short*)nullptr,
                     (float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const unsigned short*)Wt,
                     (const float*)nullptr, (unsigned short*)nullptr, M, 0, 0, q_u, q_cb, (unsigned short*)q_out, NQ,
                     (float*)nullptr, (const float*)nullptr, 0);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

#ifdef GWW_STAMP
extern "C" int gww_debug_stamps_mlp(unsigned long long* out8, int reset) {
  GWW_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gww::g_stamp_mlp), sizeof(unsigned long long) * 24));
  if (reset) {
    unsigned long long z[24] = {0};
    GWW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gww::g_stamp_mlp), z, sizeof(z)));
  }
  return GWW_OK;
}
#endif

extern "C" int gww_mlp_pack_bf16(const void* w1_folded, const void* w2, const void* wqkv_folded_or_null, void* out, int d,
                                 int F, int NQ, void* stream) {
  GWW_REQUIRE(out && ((w1_folded && w2) || (F == 0 && wqkv_folded_or_null)), "gww_mlp_pack_bf16: NULL argument");
  return launch_mlp_pack(w1_folded, w2, wqkv_folded_or_null, out, d, F, NQ, (hipStream_t)stream, nullptr);
}

extern "C" int gww_mlp_pack_op_bf16(const void* wo, const void* w1_folded, const void* w2, const void* wqkv_folded_or_null,
                                    void* out, int d, int F, int NQ, void* stream) {
  GWW_REQUIRE(wo && w1_folded && w2 && out, "gww_mlp_pack_op_bf16: NULL argument");
  return launch_mlp_pack(w1_folded, w2, wqkv_folded_or_null, out, d, F, NQ, (hipStream_t)stream, wo);
}

extern "C" int gww_attn_out_mlp_fused_bf16(float* x, const void* ctx, const float* bo, float* x_out, const float* ln_u,
                                           const float* ln_cb, const void* Wt, const float* b2, void* C, long M, int d,
                                           int F, const float* qkv_u, const float* qkv_cb, void* qkv_out, int NQ,
                                           void* stream) {
  GWW_REQUIRE(bo, "gww_attn_out_mlp_fused_bf16: NULL out_proj bias");
  return launch_mlp_fused(x, ctx, x_out, ln_u, ln_cb, Wt, b2, C, M, d, F, (hipStream_t)stream, qkv_u, qkv_cb, qkv_out, NQ,
                          nullptr, bo);
}

extern "C" int gww_attn_out_mlp_final_bf16(const float* x, const void* ctx, const float* bo, float* x_mid, const float* ln_u,
                                          const float* ln_cb, const void* Wt, const float* b2, const float* lnf_w,
                                          const float* lnf_b, float* y, long M, int d, int F, void* stream) {
  return launch_mlp_fused_final(x, ctx, x_mid, ln_u, ln_cb, Wt, b2, bo, lnf_w, lnf_b, y, M, d, F, (hipStream_t)stream);
}

extern "C" int gww_lnqkv_fused_bf16(const float* x, const float* qkv_u, const float* qkv_cb, const void* Wt, void* qkv_out,
                                    long M, int d, int NQ, void* stream) {
  return launch_lnqkv_fused(x, qkv_u, qkv_cb, Wt, qkv_out, M, d, NQ, (hipStream_t)stream);
}

extern "C" int gww_mlp_fused_bf16(float* x, const void* delta, float* x_out, const float* ln_u,
                                  const float* ln_cb, const void* Wt, const float* b2, void* C, long M, int d,
                                  int F, const float* qkv_u, const float* qkv_cb, void* qkv_out, int NQ,
                                  void* stream) {
  return launch_mlp_fused(x, delta, x_out, ln_u, ln_cb, Wt, b2, C, M, d, F, (hipStream_t)stream, qkv_u, qkv_cb, qkv_out,
                          NQ, nullptr, nullptr);
}
