// DoRA parameter gradients on the matrix cores (d = 384 / 512, r = 8), up to three projections that share
// their input in ONE pass -- q, k and v of a layer all read h1 = LN1(x) (peft 0.12.0 tuners/lora/dora.py; the
// reference adapts q/k/v [+ out_proj]: Signal_vs_Noise/src/train.py:230-237, MLGWSC-1/train.py:695).
//
// Per projection p (x [M, d], dy_p, y_p [M, d] bf16; g_p[c] = yscale_p mag_p[c] / nrm_p[c]):
//   u      = x A_p^T                       [M, 8]
//   v_p    = dy_p (g_p . B_p)              [M, 8]
//   dB_p  += scaling g_p[c] (dy_p^T u)     [d, 8]
//   dA_p  += scaling (v_p^T x)             [8, d]
//   dm_p  += (sum_rows dy_p y_p - b_p sum_rows dy_p) / mag_p
// The rank-8 factors make this 2.4 GFLOP per projection against 220 MB of operands: the kernel is HBM-bound
// (x + 2 NP operand matrices, read once) as long as the contractions stay off the VALU, which is where the
// register-blocked kernel of train_ops.hip spends its time.
//
// One workgroup (4 waves) streams 32-row tiles:
//   staging   x, dy_p -> LDS as ONE dual-use image each (row reads AND ds_read_b64_tr_b16 transposed reads,
//             256-byte rows with the chunk XOR of cdna_hip_programming.md T10 (b)); sum dy_p y_p on the fly in the
//             staging threads (a thread keeps one 8-column chunk for the whole kernel)
//   phase A   wave 0: u_all[32 rows][n = 8 p + r] = x A_all^T ; wave 1 + p: v_p -> columns 8 p .. 8 p + 7 of v_all.
//             The weights (bf16) live in registers as B operands for the whole kernel.  u_all / v_all go to LDS
//             as bf16 [n][row].
//   phase B   each wave owns D / 128 column blocks of 32:  dA_all^T[j][n] += x^T[j][rows] v_all[rows][n]   (1 MFMA
//             chain for all projections) and dB_all[c][n] += sum_p dy_p^T[c][rows] u_p*[rows][n], where u_p* is
//             u_all with the lanes outside 8 p .. 8 p + 7 zeroed and lane 24 + p set to 1 -- so column 24 + p of
//             the same accumulator collects sum_rows dy_p (the bias term of dm).
// Accumulators stay in registers over all tiles of the workgroup; each workgroup then stores its partial sums to a
// slab of scratch memory and k_dora_reduce adds the (at most 256) slabs into the gradients.
#include "common.h"

namespace gww {

namespace {
struct DoraProj {
  const float* A;      // [8, d]
  const float* Bm;     // [d, 8]
  const float* mag;    // [d]
  const float* nrm;    // [d]
  const float* bias;   // [d] (of the stored y)
  float* dA;
  float* dB;
  float* dm;
  float yscale, scaling;
  long col_off;        // column offset of this projection inside dY / Y rows
};
struct DoraProjs {
  DoraProj p[3];
};

// byte offset of 16-byte chunk ch (0..15) of row `row` inside a [32][128] bf16 sub-image (256-byte rows)
__device__ __forceinline__ int dual_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_p;

// A operand (32x32x16) = transposed block: M = column 32 jb + (lane & 31), K = rows row0 + {4 hh + (j & 3) + 8 (j >> 2)}
__device__ __forceinline__ bf16x8 tr_frag_dual(const unsigned char* img, int row0, int jb, int lane) {
  const int hh = lane >> 5, g16 = (lane & 31) >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* sub = img + (jb >> 2) * (32 * 256);
  const int ch = 4 * (jb & 3) + 2 * g16 + (p >> 1);
  const int r = row0 + 4 * hh + q;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_p)(sub + dual_off(r, ch) + 8 * (p & 1)));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_p)(sub + dual_off(r + 8, ch) + 8 * (p & 1)));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
}  // namespace

// WLDS: the phase-A weights live in LDS instead of registers (d = 768: 192 registers of weights beside 192 of
// accumulators would spill)
template <int D, int NP, bool WLDS = false>
__global__ __launch_bounds__(256) void k_dora_grads_mfma(const unsigned short* __restrict__ X, long ldx,
                                                         const unsigned short* __restrict__ dY,
                                                         const unsigned short* __restrict__ Y, long ldy,
                                                         const DoraProjs pr, long M, float* __restrict__ scratch) {
  constexpr int R = 32, CH = D / 8, IMG = R * D * 2, KS = D / 16, JB = D / 128;
  constexpr int RP = D <= 512 ? 4 : 2;   // staging row phases: RP * CH threads load, each its chunk of R / RP rows
  static_assert(D % 128 == 0 && RP * CH <= 256 && NP >= 1 && NP <= 3, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Us = lds + (1 + NP) * IMG;   // [32 n][32 rows] bf16
  unsigned char* Vs = Us + 2048;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hh = lane >> 5;

  // ---- phase-A weights of this wave as B operands: lane (n, hh), k-step s holds W[k = 16 s + 8 hh + j][n]
  constexpr int WROW = D * 2 + 16;              // LDS weight row (bytes), padded against bank conflicts
  unsigned char* Wl = Vs + 2048;                // WLDS: [8 NP rows of A_all | NP x 8 rows of (g . B)^T], bf16 [row][k]
  bf16x8 wf[WLDS ? 1 : KS];
  const bool w_lane = wave == 0 ? n < 8 * NP : (n >> 3) == wave - 1;          // lanes with a non-zero weight column
  const int w_row = wave == 0 ? (n < 8 * NP ? n : 0) : 8 * NP + 8 * (wave - 1) + (n & 7);
  if constexpr (WLDS) {
    for (int i = tid; i < 8 * NP * D; i += 256) {
      const int row = i / D, k = i - row * D;
      const DoraProj& P = pr.p[row >> 3];
      const int r = row & 7;
      *reinterpret_cast<__bf16*>(Wl + row * WROW + 2 * k) = (__bf16)P.A[(long)r * D + k];
      *reinterpret_cast<__bf16*>(Wl + (8 * NP + row) * WROW + 2 * k) =
          (__bf16)(P.yscale * (P.mag[k] / P.nrm[k]) * P.Bm[(long)k * 8 + r]);
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[s][j] = (__bf16)0.f;
    if (wave == 0) {
      if (n < 8 * NP) {
        const float* a = pr.p[n >> 3].A + (long)(n & 7) * D + 8 * hh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(a + 16 * s), hi = *reinterpret_cast<const f32x4*>(a + 16 * s + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { wf[s][j] = (__bf16)lo[j]; wf[s][4 + j] = (__bf16)hi[j]; }
        }
      }
    } else if (wave <= NP) {
      const DoraProj& P = pr.p[wave - 1];
      if ((n >> 3) == wave - 1) {
        const int r = n & 7;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * hh + j;
            wf[s][j] = (__bf16)(P.yscale * (P.mag[k] / P.nrm[k]) * P.Bm[(long)k * 8 + r]);
          }
      }
    }
  }

  // ---- staging role: 16-byte chunk c8 of rows ph, ph + RP, ... (threads past RP CH idle: D = 384, 768 -> wave 3)
  const bool stager = tid < RP * CH;
  const int c8 = tid % CH, ph = tid / CH;
  const int st_sub = (c8 >> 4) * (32 * 256), st_ch = c8 & 15;
  float dyy[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) dyy[p][j] = 0.f;

  // ---- phase-B role: column blocks jb = wave * JB + i; per-lane selectors of the masked u operand
  f32x16 accA[JB], accB[JB];
#pragma unroll
  for (int i = 0; i < JB; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) { accA[i][j] = 0.f; accB[i][j] = 0.f; }
  unsigned int fill[NP];
  bool own[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    own[p] = (n >> 3) == p;
    fill[p] = n == 24 + p ? 0x3f803f80u : 0u;   // bf16 1.0 pairs
  }
  // lanes of v_all nobody writes only feed ignored output columns, but keep them finite (U and V: 4 KB)
  for (int i = tid; i < 1024; i += 256) reinterpret_cast<unsigned int*>(Us)[i] = 0u;

  const long n_tiles = (M + R - 1) / R;
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const long r0 = t * R;
    // ---------------- staging
    if (stager) {
      // 2 rows per trip and no unrolling: the compiler would otherwise hoist all 8 rows' loads (7 x 32 registers)
#pragma unroll 1
      for (int qq = 0; qq < R / RP / 2; ++qq) {
        u32x4 vx[2], vd[NP][2], vy[NP][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const long row = r0 + RP * (2 * qq + i) + ph;
          const bool ok = row < M;
          vx[i] = ok ? *reinterpret_cast<const u32x4*>(X + row * ldx + 8 * c8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const long o = row * ldy + pr.p[p].col_off + 8 * c8;
            vd[p][i] = ok ? *reinterpret_cast<const u32x4*>(dY + o) : u32x4{0u, 0u, 0u, 0u};
            vy[p][i] = ok ? *reinterpret_cast<const u32x4*>(Y + o) : u32x4{0u, 0u, 0u, 0u};
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = RP * (2 * qq + i) + ph;
          const int off = st_sub + dual_off(row, st_ch);
          *reinterpret_cast<u32x4*>(lds + off) = vx[i];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<u32x4*>(lds + (1 + p) * IMG + off) = vd[p][i];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              dyy[p][2 * w] = fmaf(bf_lo(vd[p][i][w]), bf_lo(vy[p][i][w]), dyy[p][2 * w]);
              dyy[p][2 * w + 1] = fmaf(bf_hi(vd[p][i][w]), bf_hi(vy[p][i][w]), dyy[p][2 * w + 1]);
            }
          }
        }
      }
    }
    __syncthreads();
    // ---------------- phase A: u_all (wave 0) / v_p (wave 1 + p), rows on M, n on the lane
    if (wave <= NP) {
      const unsigned char* img = lds + wave * IMG;
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(img + (s >> 3) * (32 * 256) + dual_off(n, 2 * (s & 7) + hh));
        if constexpr (WLDS) {
          u32x4 wv = *reinterpret_cast<const u32x4*>(Wl + w_row * WROW + (16 * s + 8 * hh) * 2);
          if (!w_lane) wv = u32x4{0u, 0u, 0u, 0u};
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, wv), acc, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[s], acc, 0, 0, 0);
        }
      }
      // acc[4 c + e] = out[row 8 c + 4 hh + e][n]  ->  [n][row] bf16
      if (wave == 0 || (n >> 3) == wave - 1) {
        unsigned char* dst = (wave == 0 ? Us : Vs) + n * 64 + 8 * hh;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<u32x2*>(dst + 16 * c) = u32x2{pack2bf(acc[4 * c], acc[4 * c + 1]), pack2bf(acc[4 * c + 2], acc[4 * c + 3])};
      }
    }
    __syncthreads();
    // ---------------- phase B
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // B operands: rows 16 ks + {4 hh + (j & 3) + 8 (j >> 2)} of column n
      u32x4 uf, vf;
      {
        const u32x2 u0 = *reinterpret_cast<const u32x2*>(Us + n * 64 + 32 * ks + 8 * hh);
        const u32x2 u1 = *reinterpret_cast<const u32x2*>(Us + n * 64 + 32 * ks + 16 + 8 * hh);
        const u32x2 v0 = *reinterpret_cast<const u32x2*>(Vs + n * 64 + 32 * ks + 8 * hh);
        const u32x2 v1 = *reinterpret_cast<const u32x2*>(Vs + n * 64 + 32 * ks + 16 + 8 * hh);
        uf = u32x4{u0[0], u0[1], u1[0], u1[1]};
        vf = u32x4{v0[0], v0[1], v1[0], v1[1]};
      }
      bf16x8 ub[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        u32x4 m;
#pragma unroll
        for (int w = 0; w < 4; ++w) m[w] = own[p] ? uf[w] : fill[p];
        ub[p] = __builtin_bit_cast(bf16x8, m);
      }
      const bf16x8 vb = __builtin_bit_cast(bf16x8, vf);
#pragma unroll
      for (int i = 0; i < JB; ++i) {
        const int jb = wave * JB + i;
        accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_dual(lds, 16 * ks, jb, lane), vb, accA[i], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_dual(lds + (1 + p) * IMG, 16 * ks, jb, lane), ub[p],
                                                            accB[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---------------- epilogue: acc[4 c + e] = out[column 32 jb + 8 c + 4 hh + e][n].  Every workgroup leaves its
  // partial sums in its own slab of `scratch` (plain 16-byte stores); k_dora_reduce adds the slabs up.  (fp32
  // atomics straight into dA / dB / dm cost ~300 us per call: 6 M device-scope atomics on 20 k addresses.)
  //   slab: PA [NP][8][D] | PB [NP][8][D] | PS [NP][D] | PY [NP][RP][D]
  float* slab = scratch + (long)blockIdx.x * (NP * D * (17 + RP));
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (own[p] || n == 24 + p) {
      const int r = n & 7;
#pragma unroll
      for (int i = 0; i < JB; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int col = 32 * (wave * JB + i) + 8 * c + 4 * hh;
          const f32x4 va = {accA[i][4 * c], accA[i][4 * c + 1], accA[i][4 * c + 2], accA[i][4 * c + 3]};
          const f32x4 vb = {accB[i][4 * c], accB[i][4 * c + 1], accB[i][4 * c + 2], accB[i][4 * c + 3]};
          if (own[p]) {
            *reinterpret_cast<f32x4*>(slab + (p * 8 + r) * D + col) = va;
            *reinterpret_cast<f32x4*>(slab + NP * 8 * D + (p * 8 + r) * D + col) = vb;
          } else {
            *reinterpret_cast<f32x4*>(slab + 2 * NP * 8 * D + p * D + col) = vb;   // sum_rows dy_p
          }
        }
    }
    if (stager) {
      float* py = slab + 2 * NP * 8 * D + NP * D + (p * RP + ph) * D + 8 * c8;
      *reinterpret_cast<f32x4*>(py) = f32x4{dyy[p][0], dyy[p][1], dyy[p][2], dyy[p][3]};
      *reinterpret_cast<f32x4*>(py + 4) = f32x4{dyy[p][4], dyy[p][5], dyy[p][6], dyy[p][7]};
    }
  }
}

// element e of the slab layout, summed over the nb workgroup slabs, scaled and added to its gradient
template <int D, int NP>
__global__ __launch_bounds__(256) void k_dora_reduce(const float* __restrict__ scratch, int nb, const DoraProjs pr) {
  constexpr int RP = D <= 512 ? 4 : 2;
  constexpr int E = NP * D * (17 + RP);
  __shared__ float part[4][64];
  const int tid = threadIdx.x, el = tid & 63, sl = tid >> 6;
  const int e = blockIdx.x * 64 + el;
  float sum = 0.f;
  if (e < E)
    for (int w = sl; w < nb; w += 4) sum += scratch[(long)w * E + e];
  part[sl][el] = sum;
  __syncthreads();
  if (sl != 0 || e >= E) return;
  sum = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
  if (e < 2 * NP * 8 * D) {
    const bool isB = e >= NP * 8 * D;
    const int q = isB ? e - NP * 8 * D : e;
    const int p = q / (8 * D), r = (q - p * 8 * D) / D, col = q % D;
    const DoraProj& P = pr.p[p];
    if (isB) atomicAdd(P.dB + (long)col * 8 + r, P.scaling * P.yscale * (P.mag[col] / P.nrm[col]) * sum);
    else atomicAdd(P.dA + (long)r * D + col, P.scaling * sum);
  } else {
    const int q = e - 2 * NP * 8 * D;
    if (q < NP * D) {
      const int p = q / D, col = q % D;
      atomicAdd(pr.p[p].dm + col, -pr.p[p].bias[col] * sum / pr.p[p].mag[col]);   // - b sum_rows dy
    } else {
      const int q2 = q - NP * D;
      const int p = q2 / (RP * D), col = q2 % D;
      atomicAdd(pr.p[p].dm + col, sum / pr.p[p].mag[col]);                        // sum_rows dy y
    }
  }
}

// scratch the multi-projection kernel wants: one slab of partial sums per workgroup (at most 256 workgroups)
size_t dora_grads_scratch_bytes(int np, int d) { return (size_t)256 * np * d * 21 * sizeof(float); }   // >= 17 + RP

// np projections (1..3) that read the same X; d in {384, 512}.  Gradients are ACCUMULATED.  scratch (optional):
// dora_grads_scratch_bytes(np, d) bytes of device memory; without it the slabs come from hipMallocAsync.
int launch_dora_grads_multi(const void* X, long ldx, const void* dY, const void* Y, long ldy, int np,
                            const long* col_off, const float* const* bias_st, const float* yscale,
                            const float* scaling, const float* const* A, const float* const* Bm,
                            const float* const* mag, const float* const* nrm, float* const* dA, float* const* dB,
                            float* const* dm, long M, int d, hipStream_t s, void* scratch, size_t scratch_bytes) {
  GWW_REQUIRE(np >= 1 && np <= 3 && (d == 384 || d == 512 || (d == 768 && np == 1)),
              "dora_grads_multi: np=%d d=%d unsupported (d 384 / 512 with up to 3 projections, d 768 with one)", np, d);
  GWW_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "dora_grads_multi: row strides must be multiples of 8 elements");
  if (M == 0) return GWW_OK;
  DoraProjs pr{};
  for (int p = 0; p < np; ++p) {
    GWW_REQUIRE(A[p] && Bm[p] && mag[p] && nrm[p] && bias_st[p] && dA[p] && dB[p] && dm[p] && col_off[p] % 8 == 0,
                "dora_grads_multi: bad projection %d", p);
    pr.p[p] = DoraProj{A[p], Bm[p], mag[p], nrm[p], bias_st[p], dA[p], dB[p], dm[p], yscale[p], scaling[p], col_off[p]};
  }
  static const long nb_env = lab_int("GWW_DORA_BLOCKS", 0);   // tuning aid (lab build)
  long nb = cdiv(M, 32);
  const long nb_max = nb_env > 0 && nb_env < 256 ? nb_env : 256;   // one workgroup per CU (LDS); one slab each
  if (nb > nb_max) nb = nb_max;
  const size_t lds = (size_t)(1 + np) * 32 * d * 2 + 4096 + (d > 512 ? (size_t)16 * np * (d * 2 + 16) : 0);
  // per-workgroup partial sums: the caller's scratch (encoder workspace) or a stream-ordered allocation
  const size_t need = dora_grads_scratch_bytes(np, d);
  float* slabs = (float*)scratch;
  bool own_alloc = false;
  if (!slabs || scratch_bytes < need) {
    GWW_HIP(hipMallocAsync((void**)&slabs, need, s));
    own_alloc = true;
  }
  const int n_el = np * d * (17 + (d <= 512 ? 4 : 2));
#define GWW_DGM(DD, NPP)                                                                                            \
  do {                                                                                                              \
    GWW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dora_grads_mfma<DD, NPP, (DD > 512)>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
    hipLaunchKernelGGL((k_dora_grads_mfma<DD, NPP, (DD > 512)>), dim3((unsigned)nb), dim3(256), lds, s,             \
                       (const unsigned short*)X, ldx, (const unsigned short*)dY, (const unsigned short*)Y, ldy, pr, \
                       M, slabs);                                                                                   \
    hipLaunchKernelGGL((k_dora_reduce<DD, NPP>), dim3((unsigned)cdiv(n_el, 64)), dim3(256), 0, s, slabs, (int)nb,   \
                       pr);                                                                                         \
  } while (0)
  if (d == 768) {
    GWW_DGM(768, 1);
  } else if (d == 384) {
    if (np == 1) GWW_DGM(384, 1); else if (np == 2) GWW_DGM(384, 2); else GWW_DGM(384, 3);
  } else {
    if (np == 1) GWW_DGM(512, 1); else if (np == 2) GWW_DGM(512, 2); else GWW_DGM(512, 3);
  }
#undef GWW_DGM
  GWW_LAUNCH_CHECK();
  if (own_alloc) GWW_HIP(hipFreeAsync(slabs, s));
  return GWW_OK;
}

}  // namespace gww

using namespace gww;

extern "C" int gww_dora_grads_multi(const void* X, long ldx, const void* dY, const void* Y, long ldy, int np,
                                    const long* col_off, const float* const* bias_st, const float* yscale,
                                    const float* scaling, const float* const* A, const float* const* B,
                                    const float* const* mag, const float* const* nrm, float* const* dA,
                                    float* const* dB, float* const* dm, long M, int d, void* stream) {
  GWW_REQUIRE(X && dY && Y && col_off && bias_st && yscale && scaling && A && B && mag && nrm && dA && dB && dm,
              "gww_dora_grads_multi: NULL argument");
  return launch_dora_grads_multi(X, ldx, dY, Y, ldy, np, col_off, bias_st, yscale, scaling, A, B, mag, nrm, dA, dB, dm,
                                 M, d, (hipStream_t)stream, nullptr, 0);
}
