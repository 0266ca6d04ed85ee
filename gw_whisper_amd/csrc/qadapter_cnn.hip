// The small CNN of the Q-transform adapter as hand-written gfx950 kernels:
//
//   Conv2d(1, c1, 3, pad 1) ReLU MaxPool2d(2)  Conv2d(c1, c2, 3, pad 1) ReLU MaxPool2d(2)  Conv2d(c2, c3, 3, pad 1) ReLU
//   Conv2d(c3, 1, 1)
//
// reference: MLGWSC-1/train.py:117-122 (c = 32 / 64 / 128 on a 128 x 128 Q-scan) and MLGWSC-1/inference.py:320-330
// (c = 16 / 32 / 64 on 512 x 512), both `self.freq_adapter`.  Three launches, nothing of the 3 x 3 stacks is a library
// call (round 2 ran torch.nn -> MIOpen here: 35 ms per 256 two-detector windows for the inference variant):
//
//   k_qcnn_conv1   1 -> c1: nine taps per output, pure VALU (K = 9 is no matrix shape), one thread per POOLED pixel (a 4 x 4
//                  input patch in registers, the weights as scalar operands), ReLU + 2 x 2 max fused, channels-last output
//   k_qcnn_conv3x3 c_in -> c_out as an implicit GEMM on the matrix cores, D[c_out][pixel] = sum_k W[c_out][k] X[k][pixel],
//                  k = (tap, c_in): a 32-pixel x 8-row tile per workgroup, its (8 + 2) x 34 input patch and the weight
//                  fragments in LDS, each wave two image rows (= one pooled row) x all output channels;  POOL: ReLU + 2 x 2
//                  max in the epilogue (rows in-lane, columns by one DPP max);  FUSE: ReLU + the trailing 1 x 1 convolution
//                  (a dot product over the accumulator registers + one cross-half add), fp32 map out.
//
// BACKWARD (round 4; the adapter trains through the frozen encoder, MLGWSC-1/train.py:494-504): gww_qadapter_cnn_backward_f32.
// Nothing is saved by the forward; the backward recomputes the two pooled activations and then
//   k_qcnn_conv3x3<MODE 3>  conv3 again with dz3 = [z3 > 0] w4 dy and relu(z3) dy as its epilogue (-> dw4 by a channel sum)
//   k_qcnn_wgrad            dw[co][ci][tap] = sum_pixels dz[p][co] a[p + tap][ci], db = sum dz: fp32 VALU, exact, one tap per
//                           workgroup, register accumulators over a strip of image rows, fp32 atomics at the end
//   k_qcnn_conv3x3<MODE 2>  the data gradient da = conv^T(dz) = the SAME implicit GEMM on the transposed, tap-flipped
//                           weights (k_qcnn_pack with transpose = 1), fp32 channels-last out
//   k_qcnn_conv3x3<MODE 4>  conv2 again; its epilogue routes da2 through ReLU + 2 x 2 max (first maximum in scan order, as
//                           torch) into dz2
//   k_qcnn_conv1_bwd        conv1 again per pooled pixel, the same routing, dw1 / db1 by wave sums + atomics
// so that no library (MIOpen / aten) convolution runs in a training step either.
//
// Precision: the reference runs this CNN in fp32.  Activations and weights are carried as bf16 PAIRS (hi = bf16(v),
// lo = bf16(v - hi): 16 significant bits, the same bytes per element as fp32) and every product is three bf16 MFMAs,
// hi.hi + hi.lo + lo.hi with fp32 accumulation -- 3/16 of the cost of the fp32 MFMA at a relative error of about 1e-5
// (tests/test_gpu_qscan.py holds the stack to 1e-4 of the fp64 torch CNN).
#include "common.h"

namespace gww {
namespace {

constexpr int QC_TW = 32;   // tile width in pixels (the MFMA's 32 columns)
constexpr int QC_TR = 8;    // tile rows: 4 waves x 2 rows
constexpr int QC_PW = QC_TW + 2;

__device__ __forceinline__ unsigned short bf_hi(float v) { return f2bf(v); }
__device__ __forceinline__ unsigned short bf_lo(float v, unsigned short hi) { return f2bf(v - bf2f(hi)); }

// ---- conv1 (1 -> C1) + ReLU + 2 x 2 max pool.  in fp32 [B, H, W]; out channels-last bf16 pairs: hi plane
// [B, H/2, W/2, C1] followed by the lo plane of the same shape.  One thread per pooled pixel.
template <int C1>
__global__ __launch_bounds__(256) void k_qcnn_conv1(const float* __restrict__ in, const float* __restrict__ w,
                                                    const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                    int H, int W, long plane_elems) {
  const int Hp = H / 2, Wp = W / 2;
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Hp * Wp) return;
  const int py = idx / Wp, px = idx - py * Wp;
  const float* img = in + (long)b * H * W;
  float p[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gy = 2 * py - 1 + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gx = 2 * px - 1 + c;
      p[r][c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(long)gy * W + gx] : 0.f;
    }
  }
  unsigned short* oh = out + (((long)b * Hp + py) * Wp + px) * C1;
  unsigned short* ol = oh + plane_elems;
#pragma unroll
  for (int c8 = 0; c8 < C1; c8 += 8) {
    unsigned hi[4], lo[4];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int c = c8 + cc;
      float m = -3.4e38f;
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
          float a = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a = fmaf(w[c * 9 + ky * 3 + kx], p[oy + ky][ox + kx], a);
          m = fmaxf(m, a);
        }
      m = fmaxf(m + bias[c], 0.f);     // relu(max(conv) + b) == max(relu(conv + b))
      const unsigned short h = bf_hi(m), l = bf_lo(m, h);
      if (cc & 1) { hi[cc >> 1] |= (unsigned)h << 16; lo[cc >> 1] |= (unsigned)l << 16; }
      else { hi[cc >> 1] = h; lo[cc >> 1] = l; }
    }
    *reinterpret_cast<u32x4*>(oh + c8) = u32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4*>(ol + c8) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  }
}

// ---- 3 x 3 convolution CIN -> COUT as an implicit GEMM (header).
// act   channels-last bf16 pairs [2 planes][B, H, W, CIN]
// wfrag the weights as MFMA A-operand fragments, fragment f = ((ct * 9 + tap) * KS + s) * 2 + plane, 1 KiB each: lane
//       (r, h) holds W[32 ct + r][tap][16 s + 8 h .. + 8]  (gww_qadapter_cnn_pack_f32)
// POOL: out = channels-last bf16 pairs [2][B, H/2, W/2, COUT] of max-pooled ReLU;  FUSE: out = fp32 [B, H, W] of
// w4 . relu(conv) + b4.   grid (W / 32, H / 8, B), 256 threads.
// MODE 0 (FUSE): out = fp32 [B, H, W] of w4 . relu(conv) + b4;  1 (POOL): out = channels-last bf16 pairs of max-pooled ReLU;
// 2 (RAW): out = fp32 channels-last [B, H, W, cout_real] of conv (+ bias if given; += the old content if `accumulate`):
//   the data gradient;  3 (BWD_FUSE): aux = dy fp32 [B, H, W]; out = pairs [B, H, W, COUT] of dz = [z > 0] w4 dy, out2 =
//   pairs of relu(z) dy;  4 (BWD_POOL): aux = fp32 channels-last [B, H/2, W/2, COUT], the gradient of the POOLED output;
//   out = pairs [B, H, W, COUT] of it routed to the first maximum of each 2 x 2 window where that maximum is positive.
// act_cs: elements per pixel of the INPUT tensor (>= CIN: a channel slice of a wider tensor is read through the pointer).
enum : int { QM_FUSE = 0, QM_POOL = 1, QM_RAW = 2, QM_BWD_FUSE = 3, QM_BWD_POOL = 4 };
template <int CIN, int COUT, int MODE, bool WLDS>
__global__ __launch_bounds__(256) void k_qcnn_conv3x3(const unsigned short* __restrict__ act, long act_plane, int act_cs,
                                                      const unsigned short* __restrict__ wfrag,
                                                      const float* __restrict__ bias, const float* __restrict__ w4,
                                                      void* __restrict__ outp, long out_plane, int H, int W,
                                                      const float* __restrict__ aux, void* __restrict__ out2p,
                                                      int cout_real, int accumulate) {
  constexpr bool POOL = MODE == QM_POOL;
  constexpr int P = CIN / 8;            // 16-byte chunks per pixel and plane
  constexpr int PXROW = 16 / P;         // pixels per 256-byte bank row
  constexpr int KS = CIN / 16;          // k-steps per tap
  constexpr int CT = COUT / 32;         // output-channel tiles
  constexpr int PATCH = (QC_TR + 2) * QC_PW * CIN * 2;   // bytes per plane
  constexpr int NFRAG = CT * 9 * KS * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
  unsigned char* patch = qlds;                       // [2 planes][(TR + 2)][34][CIN] bf16, chunks swizzled
  unsigned char* wl = qlds + 2 * PATCH;              // [NFRAG][1 KiB]  (WLDS)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int x0 = blockIdx.x * QC_TW, y0 = blockIdx.y * QC_TR, b = blockIdx.z;

  // ---- stage the input patch (zero outside the image: the convolution's padding) and the weight fragments
  constexpr int NCH = (QC_TR + 2) * QC_PW * P;       // chunks per plane
  for (int i = tid; i < 2 * NCH; i += 256) {
    const int pl = i >= NCH, j = pl ? i - NCH : i;
    const int c = j % P, q = j / P, px = q % QC_PW, pr = q / QC_PW;
    const int gy = y0 - 1 + pr, gx = x0 - 1 + px;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (gy >= 0 && gy < H && gx >= 0 && gx < W)
      v = *reinterpret_cast<const u32x4*>(act + pl * act_plane + (((long)b * H + gy) * W + gx) * act_cs + 8 * c);
    *reinterpret_cast<u32x4*>(patch + pl * PATCH + (pr * QC_PW + px) * CIN * 2 + 16 * (c ^ ((px / PXROW) % P))) = v;
  }
  if (WLDS)
    for (int i = tid; i < NFRAG * 64; i += 256)
      *reinterpret_cast<u32x4*>(wl + (long)i * 16) = *reinterpret_cast<const u32x4*>(wfrag + (long)i * 8);
  __syncthreads();

  // ---- accumulators start from the bias: lane = pixel, register e = channel (e & 3) + 8 (e >> 2) + 4 h of the tile
  f32x16 acc[2][CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float bv = bias ? bias[32 * ct + (e & 3) + 8 * (e >> 2) + 4 * h] : 0.f;
      acc[0][ct][e] = bv;
      acc[1][ct][e] = bv;
    }
  const int ry = 2 * wave;              // this wave's two rows of the tile
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 xb[2][2];                  // [row][plane]: channels 16 s + 8 h .. of pixel r + dx, row ry + row + dy
#pragma unroll
      for (int row = 0; row < 2; ++row) {
        const int px = r + dx, pr = ry + row + dy;
        const int off = (pr * QC_PW + px) * CIN * 2 + 16 * ((2 * s + h) ^ ((px / PXROW) % P));
        xb[row][0] = *reinterpret_cast<const bf16x8*>(patch + off);
        xb[row][1] = *reinterpret_cast<const bf16x8*>(patch + PATCH + off);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int f = ((ct * 9 + tap) * KS + s) * 2;
        bf16x8 wh, wlo;
        if (WLDS) {
          wh = *reinterpret_cast<const bf16x8*>(wl + f * 1024 + lane * 16);
          wlo = *reinterpret_cast<const bf16x8*>(wl + (f + 1) * 1024 + lane * 16);
        } else {
          wh = *reinterpret_cast<const bf16x8*>(wfrag + (long)f * 512 + lane * 8);
          wlo = *reinterpret_cast<const bf16x8*>(wfrag + (long)(f + 1) * 512 + lane * 8);
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          acc[row][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xb[row][0], acc[row][ct], 0, 0, 0);
          acc[row][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xb[row][1], acc[row][ct], 0, 0, 0);
          acc[row][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xb[row][0], acc[row][ct], 0, 0, 0);
        }
      }
    }
  }

  if constexpr (POOL) {
    // ReLU + 2 x 2 max: the two rows in-lane, the two columns by one cross-lane max (lane ^ 1); even lanes store
    const int Hp = H / 2, Wp = W / 2;
    unsigned short* oh = reinterpret_cast<unsigned short*>(outp) + (((long)b * Hp + (y0 + ry) / 2) * Wp + (x0 + r) / 2) * COUT;
    unsigned short* ol = oh + out_plane;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float m[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = fmaxf(fmaxf(acc[0][ct][e], acc[1][ct][e]), 0.f);
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        m[e] = fmaxf(v, o);
      }
      if (!(lane & 1)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned short hh[4], ll[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { hh[q] = bf_hi(m[4 * g + q]); ll[q] = bf_lo(m[4 * g + q], hh[q]); }
          const int c = 32 * ct + 8 * g + 4 * h;
          *reinterpret_cast<u32x2*>(oh + c) = u32x2{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16)};
          *reinterpret_cast<u32x2*>(ol + c) = u32x2{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16)};
        }
      }
    }
  } else if constexpr (MODE == QM_RAW) {
    float* out = reinterpret_cast<float*>(outp);
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      float* op = out + (((long)b * H + y0 + ry + row) * W + x0 + r) * cout_real;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 32 * ct + 8 * g + 4 * h;
          if (c < cout_real) {
            f32x4 v = {acc[row][ct][4 * g], acc[row][ct][4 * g + 1], acc[row][ct][4 * g + 2], acc[row][ct][4 * g + 3]};
            if (accumulate) v += *reinterpret_cast<const f32x4*>(op + c);
            *reinterpret_cast<f32x4*>(op + c) = v;
          }
        }
    }
  } else if constexpr (MODE == QM_BWD_FUSE) {
    unsigned short* o1 = reinterpret_cast<unsigned short*>(outp);
    unsigned short* o2 = reinterpret_cast<unsigned short*>(out2p);
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      const long pix = ((long)b * H + y0 + ry + row) * W + x0 + r;
      const float dyv = aux[pix];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 32 * ct + 8 * g + 4 * h;
          unsigned short zh[4], zl[4], rh[4], rl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float z = acc[row][ct][4 * g + q];
            const float dz = z > 0.f ? w4[c + q] * dyv : 0.f, rz = fmaxf(z, 0.f) * dyv;
            zh[q] = bf_hi(dz); zl[q] = bf_lo(dz, zh[q]);
            rh[q] = bf_hi(rz); rl[q] = bf_lo(rz, rh[q]);
          }
          *reinterpret_cast<u32x2*>(o1 + pix * COUT + c) = u32x2{(unsigned)zh[0] | ((unsigned)zh[1] << 16), (unsigned)zh[2] | ((unsigned)zh[3] << 16)};
          *reinterpret_cast<u32x2*>(o1 + out_plane + pix * COUT + c) = u32x2{(unsigned)zl[0] | ((unsigned)zl[1] << 16), (unsigned)zl[2] | ((unsigned)zl[3] << 16)};
          *reinterpret_cast<u32x2*>(o2 + pix * COUT + c) = u32x2{(unsigned)rh[0] | ((unsigned)rh[1] << 16), (unsigned)rh[2] | ((unsigned)rh[3] << 16)};
          *reinterpret_cast<u32x2*>(o2 + out_plane + pix * COUT + c) = u32x2{(unsigned)rl[0] | ((unsigned)rl[1] << 16), (unsigned)rl[2] | ((unsigned)rl[3] << 16)};
        }
    }
  } else if constexpr (MODE == QM_BWD_POOL) {
    // the wave's two rows are one pooled row; lanes 2 p / 2 p + 1 are the window's two columns.  Scan order of a window:
    // (row 0, col 0), (row 0, col 1), (row 1, col 0), (row 1, col 1) -- torch's max_pool2d backward takes the FIRST maximum
    unsigned short* o1 = reinterpret_cast<unsigned short*>(outp);
    const int Hp = H / 2, Wp = W / 2;
    const float* gp = aux + (((long)b * Hp + (y0 + ry) / 2) * Wp + (x0 + r) / 2) * COUT;
    const bool odd = lane & 1;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * ct + 8 * g + 4 * h;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + c);
        unsigned short dh[2][4], dl[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v0 = acc[0][ct][4 * g + q], v1 = acc[1][ct][4 * g + q];
          const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v0), 0xB1, 0xF, 0xF, true));
          const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v1), 0xB1, 0xF, 0xF, true));
          const float m = fmaxf(fmaxf(v0, v1), fmaxf(p0, p1));
          // positions in scan order: even lane owns #0 (v0) and #2 (v1), its partner #1 (p0) and #3 (p1); odd lane owns #1, #3
          const float s0 = odd ? p0 : v0, s1 = odd ? v0 : p0, s2 = odd ? p1 : v1;
          const bool first0 = s0 == m, first1 = !first0 && s1 == m, first2 = !first0 && !first1 && s2 == m;
          const bool first3 = !first0 && !first1 && !first2;
          const bool win_row0 = odd ? first1 : first0, win_row1 = odd ? first3 : first2;
          const float d0 = (m > 0.f && win_row0) ? gv[q] : 0.f, d1 = (m > 0.f && win_row1) ? gv[q] : 0.f;
          dh[0][q] = bf_hi(d0); dl[0][q] = bf_lo(d0, dh[0][q]);
          dh[1][q] = bf_hi(d1); dl[1][q] = bf_lo(d1, dh[1][q]);
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const long pix = ((long)b * H + y0 + ry + row) * W + x0 + r;
          *reinterpret_cast<u32x2*>(o1 + pix * COUT + c) = u32x2{(unsigned)dh[row][0] | ((unsigned)dh[row][1] << 16), (unsigned)dh[row][2] | ((unsigned)dh[row][3] << 16)};
          *reinterpret_cast<u32x2*>(o1 + out_plane + pix * COUT + c) = u32x2{(unsigned)dl[row][0] | ((unsigned)dl[row][1] << 16), (unsigned)dl[row][2] | ((unsigned)dl[row][3] << 16)};
        }
      }
  } else {
    // ReLU + the 1 x 1 convolution to one channel: a dot product over this lane's 16 x CT channels, + the other half's
    float* out = reinterpret_cast<float*>(outp);
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      float y = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          y = fmaf(w4[32 * ct + (e & 3) + 8 * (e >> 2) + 4 * h], fmaxf(acc[row][ct][e], 0.f), y);
      y += __shfl_xor(y, 32, 64);
      if (h == 0) out[((long)b * H + y0 + ry + row) * W + x0 + r] = y + w4[COUT];   // b4 rides behind the COUT weights
    }
  }
}

// fp32 [COUT, CIN, 3, 3] -> MFMA A-operand fragments as bf16 pairs (layout: k_qcnn_conv3x3)
// transpose: the operand of the data gradient -- CIN / COUT are then the convolution^T's (CIN = the forward's output
// channels): W'[cout][cin][tap] = w[cin][cout][8 - tap] of the forward weight w [CIN, cout_real, 3, 3]; rows cout >= cout_real
// (COUT padded to 32) are zero
__global__ __launch_bounds__(256) void k_qcnn_pack(const float* __restrict__ w, unsigned short* __restrict__ out, int CIN,
                                                   int COUT, int transpose, int cout_real) {
  const int KS = CIN / 16, nfrag = (COUT / 32) * 9 * KS * 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)nfrag * 512; i += (long)gridDim.x * 256) {
    const int f = (int)(i >> 9), within = (int)(i & 511), lane = within >> 3, j = within & 7;
    const int plane = f & 1, s = (f >> 1) % KS, tap = ((f >> 1) / KS) % 9, ct = (f >> 1) / KS / 9;
    const int cout = 32 * ct + (lane & 31), cin = 16 * s + 8 * (lane >> 5) + j;
    const float v = !transpose ? w[((long)cout * CIN + cin) * 9 + tap]
                               : (cout < cout_real ? w[((long)cin * cout_real + cout) * 9 + (8 - tap)] : 0.f);
    const unsigned short hi = f2bf(v);
    out[i] = plane ? f2bf(v - bf2f(hi)) : hi;
  }
}

template <int CIN, int COUT, int MODE>
int launch_conv3x3(const unsigned short* act, long act_plane, const unsigned short* wfrag, const float* bias,
                   const float* w4, void* out, long out_plane, int B, int H, int W, hipStream_t s, int act_cs = CIN,
                   const float* aux = nullptr, void* out2 = nullptr, int cout_real = COUT, int accumulate = 0) {
  constexpr int patch = 2 * (QC_TR + 2) * QC_PW * CIN * 2;
  constexpr int wbytes = (COUT / 32) * 9 * (CIN / 16) * 2 * 1024;
  constexpr bool WLDS = patch + wbytes <= 128 * 1024;
  constexpr int lds = patch + (WLDS ? wbytes : 0);
  static_assert(lds <= 160 * 1024, "conv3x3: the input patch does not fit the LDS");
  auto kern = k_qcnn_conv3x3<CIN, COUT, MODE, WLDS>;
  if (lds > 64 * 1024) GWW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(W / QC_TW, H / QC_TR, B), dim3(256), lds, s, act, act_plane, act_cs, wfrag, bias, w4, out,
                     out_plane, H, W, aux, out2, cout_real, accumulate);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---- weight gradient of a 3 x 3 convolution: dw[co][ci][tap] = sum_{b,y,x} dz[b,y,x,co] a[b, y + dy - 1, x + dx - 1, ci]
// (torch layout [COUT, CIN, 3, 3]), db[co] = sum dz.  a, dz: channels-last bf16 pairs (hi + lo = the 16-bit values the
// forward computed with).  fp32 VALU, exact products: grid (row strips, 9 taps); a workgroup stages one image row of dz and
// the matching (shifted) row of a in LDS as fp32, thread t owns output channel co = t % COUT and the ACC = COUT CIN / 256
// input channels of group t / COUT, accumulates over its strip of rows in registers and adds them to dw with fp32 atomics
// (9 x strips x COUT x CIN adds in all: a few MB).  W <= 256.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_qcnn_wgrad(const unsigned short* __restrict__ a, long a_plane,
                                                    const unsigned short* __restrict__ dz, long dz_plane,
                                                    float* __restrict__ dw, float* __restrict__ db, int H, int W,
                                                    long n_rows, int rows_per_wg) {
  constexpr int ACC = COUT * CIN / 256, GROUPS = 256 / COUT;
  static_assert(ACC >= 1 && ACC * GROUPS == CIN, "wgrad: COUT CIN must be a multiple of 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
  float* dz_s = reinterpret_cast<float*>(wl);               // [W][COUT]
  float* a_s = dz_s + (long)W * COUT;                       // [W + 2][CIN], column x + 1 <-> pixel x (zero borders)
  const int tid = threadIdx.x, tap = blockIdx.y, dy = tap / 3, dx = tap % 3;
  const int co = tid % COUT, cig = tid / COUT;
  float acc[ACC], bsum = 0.f;
#pragma unroll
  for (int k = 0; k < ACC; ++k) acc[k] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < n_rows ? r0 + rows_per_wg : n_rows;
  for (long row = r0; row < r1; ++row) {
    const long b = row / H;
    const int y = (int)(row - b * H), ya = y + dy - 1;
    __syncthreads();
    for (int i = tid; i < W * COUT / 8; i += 256) {
      const long off = row * W * COUT + 8L * i;
      const u32x4 hi = *reinterpret_cast<const u32x4*>(dz + off), lo = *reinterpret_cast<const u32x4*>(dz + dz_plane + off);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dz_s[8 * i + 2 * q] = bf2f((unsigned short)(hi[q] & 0xffff)) + bf2f((unsigned short)(lo[q] & 0xffff));
        dz_s[8 * i + 2 * q + 1] = bf2f((unsigned short)(hi[q] >> 16)) + bf2f((unsigned short)(lo[q] >> 16));
      }
    }
    const bool in_img = ya >= 0 && ya < H;
    for (int i = tid; i < (W + 2) * CIN / 8; i += 256) {
      const int px = (8 * i) / CIN - 1, c = (8 * i) % CIN;     // pixel of this chunk (-1 and W: the zero borders)
      u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
      if (in_img && px >= 0 && px < W) {
        const long off = ((b * H + ya) * W + px) * CIN + c;
        hi = *reinterpret_cast<const u32x4*>(a + off);
        lo = *reinterpret_cast<const u32x4*>(a + a_plane + off);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a_s[8 * i + 2 * q] = bf2f((unsigned short)(hi[q] & 0xffff)) + bf2f((unsigned short)(lo[q] & 0xffff));
        a_s[8 * i + 2 * q + 1] = bf2f((unsigned short)(hi[q] >> 16)) + bf2f((unsigned short)(lo[q] >> 16));
      }
    }
    __syncthreads();
    for (int x = 0; x < W; ++x) {
      const float d = dz_s[x * COUT + co];
      const float* ap = a_s + (x + dx) * CIN + cig * ACC;
      if (tap == 4 && cig == 0) bsum += d;
#pragma unroll
      for (int k = 0; k < ACC; ++k) acc[k] = fmaf(d, ap[k], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < ACC; ++k) atomicAdd(dw + ((long)co * CIN + cig * ACC + k) * 9 + tap, acc[k]);
  if (tap == 4 && cig == 0) atomicAdd(db + co, bsum);
}

// out[c] += sum over n pixels of a channels-last bf16-pair tensor [n][C] (C <= 256, 256 % C == 0): dw4 from relu(z3) dy
__global__ __launch_bounds__(256) void k_qcnn_chan_sum(const unsigned short* __restrict__ v, long plane, long n, int C,
                                                       float* __restrict__ out) {
  const int c = threadIdx.x % C, sub = threadIdx.x / C, nsub = 256 / C;
  float s = 0.f;
  for (long p = (long)blockIdx.x * nsub + sub; p < n; p += (long)gridDim.x * nsub)
    s += bf2f(v[p * C + c]) + bf2f(v[plane + p * C + c]);
  atomicAdd(out + c, s);
}
// out[0] += sum of n floats (db4 = sum dy)
__global__ __launch_bounds__(256) void k_qcnn_sum_f32(const float* __restrict__ v, long n, float* __restrict__ out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// ---- conv1 backward: one thread per POOLED pixel, as the forward: the four convolution outputs of the window are
// recomputed per channel, da1 (fp32 channels-last [B, H/2, W/2, C1], the gradient of the pooled output) goes to the first
// maximum if it is positive; dw1[c][ky][kx] += g in[...], db1[c] += g: wave sums, then one atomic per wave and value.
template <int C1>
__global__ __launch_bounds__(256) void k_qcnn_conv1_bwd(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ da,
                                                        float* __restrict__ dw, float* __restrict__ db, int H, int W) {
  const int Hp = H / 2, Wp = W / 2;
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < Hp * Wp;
  const int py = live ? idx / Wp : 0, px = live ? idx - py * Wp : 0;
  const float* img = in + (long)b * H * W;
  float p[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gy = 2 * py - 1 + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gx = 2 * px - 1 + c;
      p[r][c] = (live && gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(long)gy * W + gx] : 0.f;
    }
  }
  const float* gp = da + (((long)b * Hp + py) * Wp + px) * C1;
  // per-workgroup sums in LDS first: every workgroup adds into the SAME 10 C1 global words, and atomics on one address
  // serialise at the memory side (one add per wave and value -- 5 M adds on 320 addresses -- took 1.5 ms)
  __shared__ float red[C1 * 10];
  for (int i = threadIdx.x; i < C1 * 10; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int c = 0; c < C1; ++c) {
    float v[4];
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
      for (int ox = 0; ox < 2; ++ox) {
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc = fmaf(w[c * 9 + ky * 3 + kx], p[oy + ky][ox + kx], acc);
        v[2 * oy + ox] = acc;
      }
    const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    const int pos = v[0] == m ? 0 : (v[1] == m ? 1 : (v[2] == m ? 2 : 3));   // first maximum in scan order
    const float g = (live && m + bias[c] > 0.f) ? gp[c] : 0.f;               // relu(max + b) as the forward
    const int oy = pos >> 1, ox = pos & 1;
    float contrib[10];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        // p[oy + ky][ox + kx] with a runtime (oy, ox): select among the four candidates (no dynamic register indexing)
        const float a00 = p[ky][kx], a01 = p[ky][kx + 1], a10 = p[ky + 1][kx], a11 = p[ky + 1][kx + 1];
        const float sel = oy ? (ox ? a11 : a10) : (ox ? a01 : a00);
        contrib[ky * 3 + kx] = g * sel;
      }
    contrib[9] = g;
    // sums over the 16 lanes of a DPP row (four butterfly steps on the vector ALU; __shfl_xor is an LDS round trip per step
    // and made this kernel the most expensive of the whole training step), then one atomic per row and value
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float v2 = contrib[k];
      v2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
      v2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
      v2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), 0x141, 0xF, 0xF, true));   // row_half_mirror
      v2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), 0x140, 0xF, 0xF, true));   // row_mirror
      if ((threadIdx.x & 15) == 0 && v2 != 0.f) atomicAdd(&red[c * 10 + k], v2);   // LDS
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C1 * 10; i += 256) {
    const int c = i / 10, k = i - 10 * c;
    if (red[i] != 0.f) atomicAdd(k < 9 ? dw + c * 9 + k : db + c, red[i]);
  }
}

template <int CIN, int COUT>
int launch_wgrad(const unsigned short* a, long a_plane, const unsigned short* dz, long dz_plane, float* dw, float* db, int B,
                 int H, int W, hipStream_t s) {
  GWW_REQUIRE(W <= 256, "qadapter_cnn backward: maps wider than 256 pixels at the convolution's resolution are not supported");
  const long n_rows = (long)B * H;
  const int rows_per_wg = 16;
  const int lds = (W * COUT + (W + 2) * CIN) * 4;
  auto kern = k_qcnn_wgrad<CIN, COUT>;
  if (lds > 64 * 1024) GWW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(n_rows, rows_per_wg), 9), dim3(256), lds, s, a, a_plane, dz, dz_plane, dw, db,
                     H, W, n_rows, rows_per_wg);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

struct QcnnPacked {   // offsets (bytes) into the packed blob
  size_t w1, b1, w2, b2, w3, b3, w4, total;
};
QcnnPacked qcnn_layout(int c1, int c2, int c3) {
  QcnnPacked p{};
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 255) / 256 * 256; return at; };
  p.w1 = take((size_t)c1 * 9 * 4);
  p.b1 = take((size_t)c1 * 4);
  p.w2 = take((size_t)(c2 / 32) * 9 * (c1 / 16) * 2 * 1024);
  p.b2 = take((size_t)c2 * 4);
  p.w3 = take((size_t)(c3 / 32) * 9 * (c2 / 16) * 2 * 1024);
  p.b3 = take((size_t)c3 * 4);
  p.w4 = take((size_t)(c3 + 1) * 4);   // w4 [c3] followed by b4
  p.total = o;
  return p;
}
bool qcnn_supported(int c1, int c2, int c3) {
  return (c1 == 16 && c2 == 32 && c3 == 64) || (c1 == 32 && c2 == 64 && c3 == 128);
}

}  // namespace
}  // namespace gww

using namespace gww;

extern "C" size_t gww_qadapter_cnn_packed_bytes(int c1, int c2, int c3) {
  return qcnn_supported(c1, c2, c3) ? qcnn_layout(c1, c2, c3).total : 0;
}

extern "C" size_t gww_qadapter_cnn_workspace_bytes(int B, int H, int W, int c1, int c2) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const size_t a1 = (size_t)B * (H / 2) * (W / 2) * c1 * 4, a2 = (size_t)B * (H / 4) * (W / 4) * c2 * 4;   // bf16 pairs
  return (a1 + 255) / 256 * 256 + (a2 + 255) / 256 * 256;
}

extern "C" int gww_qadapter_cnn_pack_f32(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                         const float* b3, const float* w4, const float* b4, int c1, int c2, int c3,
                                         void* packed, void* stream) {
  GWW_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && w4 && b4 && packed, "gww_qadapter_cnn_pack_f32: NULL argument");
  GWW_REQUIRE(qcnn_supported(c1, c2, c3), "gww_qadapter_cnn_pack_f32: channels %d / %d / %d (the reference has 16 / 32 / 64 and 32 / 64 / 128)", c1, c2, c3);
  hipStream_t s = (hipStream_t)stream;
  const QcnnPacked p = qcnn_layout(c1, c2, c3);
  char* base = (char*)packed;
  GWW_HIP(hipMemcpyAsync(base + p.w1, w1, (size_t)c1 * 9 * 4, hipMemcpyDeviceToDevice, s));
  GWW_HIP(hipMemcpyAsync(base + p.b1, b1, (size_t)c1 * 4, hipMemcpyDeviceToDevice, s));
  GWW_HIP(hipMemcpyAsync(base + p.b2, b2, (size_t)c2 * 4, hipMemcpyDeviceToDevice, s));
  GWW_HIP(hipMemcpyAsync(base + p.b3, b3, (size_t)c3 * 4, hipMemcpyDeviceToDevice, s));
  GWW_HIP(hipMemcpyAsync(base + p.w4, w4, (size_t)c3 * 4, hipMemcpyDeviceToDevice, s));
  GWW_HIP(hipMemcpyAsync(base + p.w4 + (size_t)c3 * 4, b4, 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w2, (unsigned short*)(base + p.w2), c1, c2, 0, c2);
  GWW_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w3, (unsigned short*)(base + p.w3), c2, c3, 0, c3);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

extern "C" int gww_qadapter_cnn_forward_f32(const float* qspec, int B, int H, int W, const void* packed, int c1, int c2,
                                            int c3, void* workspace, size_t workspace_bytes, float* y,
                                            void* stream) {
  GWW_REQUIRE(qspec && packed && workspace && y, "gww_qadapter_cnn_forward_f32: NULL argument");
  GWW_REQUIRE(qcnn_supported(c1, c2, c3), "gww_qadapter_cnn_forward_f32: channels %d / %d / %d", c1, c2, c3);
  GWW_REQUIRE(B >= 0 && H > 0 && W > 0 && H % 32 == 0 && W % 128 == 0,
              "gww_qadapter_cnn_forward_f32: H %% 32 == 0 and W %% 128 == 0 required (got %d x %d)", H, W);
  if (workspace_bytes < gww_qadapter_cnn_workspace_bytes(B, H, W, c1, c2))
    return fail(GWW_ERR_WORKSPACE, "gww_qadapter_cnn_forward_f32: workspace %zu < %zu bytes", workspace_bytes,
                gww_qadapter_cnn_workspace_bytes(B, H, W, c1, c2));
  if (B == 0) return GWW_OK;
  GWW_REQUIRE(B <= 65535, "gww_qadapter_cnn_forward_f32: at most 65535 maps per call");
  hipStream_t s = (hipStream_t)stream;
  const QcnnPacked p = qcnn_layout(c1, c2, c3);
  const char* base = (const char*)packed;
  const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4;
  const long pl1 = (long)B * H1 * W1 * c1, pl2 = (long)B * H2 * W2 * c2;
  unsigned short* a1 = (unsigned short*)workspace;
  unsigned short* a2 = (unsigned short*)((char*)workspace + ((size_t)pl1 * 4 + 255) / 256 * 256);
  const float* w1 = (const float*)(base + p.w1);
  const float* b1 = (const float*)(base + p.b1);
  const dim3 g1((unsigned)cdiv((long)H1 * W1, 256), (unsigned)B);
  if (c1 == 16) hipLaunchKernelGGL(k_qcnn_conv1<16>, g1, dim3(256), 0, s, qspec, w1, b1, a1, H, W, pl1);
  else hipLaunchKernelGGL(k_qcnn_conv1<32>, g1, dim3(256), 0, s, qspec, w1, b1, a1, H, W, pl1);
  GWW_LAUNCH_CHECK();
  const unsigned short* w2 = (const unsigned short*)(base + p.w2);
  const unsigned short* w3 = (const unsigned short*)(base + p.w3);
  const float* b2 = (const float*)(base + p.b2);
  const float* b3 = (const float*)(base + p.b3);
  const float* w4 = (const float*)(base + p.w4);
  if (c1 == 16) {
    GWW_TRY((launch_conv3x3<16, 32, QM_POOL>(a1, pl1, w2, b2, nullptr, a2, pl2, B, H1, W1, s)));
    GWW_TRY((launch_conv3x3<32, 64, QM_FUSE>(a2, pl2, w3, b3, w4, y, 0, B, H2, W2, s)));
  } else {
    GWW_TRY((launch_conv3x3<32, 64, QM_POOL>(a1, pl1, w2, b2, nullptr, a2, pl2, B, H1, W1, s)));
    GWW_TRY((launch_conv3x3<64, 128, QM_FUSE>(a2, pl2, w3, b3, w4, y, 0, B, H2, W2, s)));
  }
  return GWW_OK;
}

namespace {
struct QcnnBwdWs {
  size_t a1, a2, dz3, rz3, da2, dz2, da1, w3t, w2t, total;
};
QcnnBwdWs qcnn_bwd_layout(int B, int H, int W, int c1, int c2, int c3) {
  QcnnBwdWs w{};
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 255) / 256 * 256; return at; };
  const size_t p1 = (size_t)B * (H / 2) * (W / 2), p2 = (size_t)B * (H / 4) * (W / 4);
  const int c1p = c1 < 32 ? 32 : c1, c2p = c2 < 32 ? 32 : c2;
  w.a1 = take(p1 * c1 * 4);
  w.a2 = take(p2 * c2 * 4);
  w.dz3 = take(p2 * c3 * 4);
  w.rz3 = take(p2 * c3 * 4);
  w.da2 = take(p2 * c2 * 4);
  w.dz2 = take(p1 * c2 * 4);
  w.da1 = take(p1 * c1 * 4);
  w.w3t = take((size_t)(c2p / 32) * 9 * (c3 / 16) * 2 * 1024);
  w.w2t = take((size_t)(c1p / 32) * 9 * (c2 / 16) * 2 * 1024);
  w.total = o;
  return w;
}
}  // namespace

extern "C" size_t gww_qadapter_cnn_backward_workspace_bytes(int B, int H, int W, int c1, int c2, int c3) {
  if (B <= 0 || H <= 0 || W <= 0 || !qcnn_supported(c1, c2, c3)) return 0;
  return qcnn_bwd_layout(B, H, W, c1, c2, c3).total;
}

extern "C" int gww_qadapter_cnn_backward_f32(const float* qspec, const float* dy, int B, int H, int W, const void* packed,
                                             const float* w2, const float* w3, int c1, int c2, int c3, void* workspace,
                                             size_t workspace_bytes, float* dw1, float* db1, float* dw2, float* db2,
                                             float* dw3, float* db3, float* dw4, float* db4, void* stream) {
  GWW_REQUIRE(qspec && dy && packed && w2 && w3 && workspace && dw1 && db1 && dw2 && db2 && dw3 && db3 && dw4 && db4,
              "gww_qadapter_cnn_backward_f32: NULL argument");
  GWW_REQUIRE(qcnn_supported(c1, c2, c3), "gww_qadapter_cnn_backward_f32: channels %d / %d / %d", c1, c2, c3);
  GWW_REQUIRE(B >= 0 && H > 0 && W > 0 && H % 32 == 0 && W % 128 == 0 && W <= 512,
              "gww_qadapter_cnn_backward_f32: H %% 32 == 0, W %% 128 == 0, W <= 512 required (got %d x %d)", H, W);
  const QcnnBwdWs L = qcnn_bwd_layout(B > 0 ? B : 1, H, W, c1, c2, c3);
  if (workspace_bytes < L.total)
    return fail(GWW_ERR_WORKSPACE, "gww_qadapter_cnn_backward_f32: workspace %zu < %zu bytes", workspace_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  GWW_HIP(hipMemsetAsync(dw1, 0, (size_t)c1 * 9 * 4, s));
  GWW_HIP(hipMemsetAsync(db1, 0, (size_t)c1 * 4, s));
  GWW_HIP(hipMemsetAsync(dw2, 0, (size_t)c2 * c1 * 9 * 4, s));
  GWW_HIP(hipMemsetAsync(db2, 0, (size_t)c2 * 4, s));
  GWW_HIP(hipMemsetAsync(dw3, 0, (size_t)c3 * c2 * 9 * 4, s));
  GWW_HIP(hipMemsetAsync(db3, 0, (size_t)c3 * 4, s));
  GWW_HIP(hipMemsetAsync(dw4, 0, (size_t)c3 * 4, s));
  GWW_HIP(hipMemsetAsync(db4, 0, 4, s));
  if (B == 0) return GWW_OK;
  GWW_REQUIRE(B <= 65535, "gww_qadapter_cnn_backward_f32: at most 65535 maps per call");
  const QcnnPacked p = qcnn_layout(c1, c2, c3);
  const char* base = (const char*)packed;
  char* ws = (char*)workspace;
  const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4;
  const long n1 = (long)B * H1 * W1, n2 = (long)B * H2 * W2;
  const long pl_a1 = n1 * c1, pl_a2 = n2 * c2, pl_z3 = n2 * c3, pl_z2 = n1 * c2;
  unsigned short* a1 = (unsigned short*)(ws + L.a1);
  unsigned short* a2 = (unsigned short*)(ws + L.a2);
  unsigned short* dz3 = (unsigned short*)(ws + L.dz3);
  unsigned short* rz3 = (unsigned short*)(ws + L.rz3);
  float* da2 = (float*)(ws + L.da2);
  unsigned short* dz2 = (unsigned short*)(ws + L.dz2);
  float* da1 = (float*)(ws + L.da1);
  unsigned short* w3t = (unsigned short*)(ws + L.w3t);
  unsigned short* w2t = (unsigned short*)(ws + L.w2t);
  const float* w1f = (const float*)(base + p.w1);
  const float* b1f = (const float*)(base + p.b1);
  const unsigned short* w2f = (const unsigned short*)(base + p.w2);
  const unsigned short* w3f = (const unsigned short*)(base + p.w3);
  const float* b2f = (const float*)(base + p.b2);
  const float* b3f = (const float*)(base + p.b3);
  const float* w4f = (const float*)(base + p.w4);
  // ---- the two pooled activations again (the forward saves nothing)
  const dim3 g1((unsigned)cdiv((long)H1 * W1, 256), (unsigned)B);
  if (c1 == 16) hipLaunchKernelGGL(k_qcnn_conv1<16>, g1, dim3(256), 0, s, qspec, w1f, b1f, a1, H, W, pl_a1);
  else hipLaunchKernelGGL(k_qcnn_conv1<32>, g1, dim3(256), 0, s, qspec, w1f, b1f, a1, H, W, pl_a1);
  GWW_LAUNCH_CHECK();
  // db4 = sum dy
  hipLaunchKernelGGL(k_qcnn_sum_f32, dim3(256), dim3(256), 0, s, dy, n2, db4);
  GWW_LAUNCH_CHECK();
  const int c1p = c1 < 32 ? 32 : c1;
  if (c1 == 16) {
    GWW_TRY((launch_conv3x3<16, 32, QM_POOL>(a1, pl_a1, w2f, b2f, nullptr, a2, pl_a2, B, H1, W1, s)));
    // conv3 again -> dz3, relu(z3) dy
    GWW_TRY((launch_conv3x3<32, 64, QM_BWD_FUSE>(a2, pl_a2, w3f, b3f, w4f, dz3, pl_z3, B, H2, W2, s, 32, dy, rz3)));
    GWW_TRY((launch_wgrad<32, 64>(a2, pl_a2, dz3, pl_z3, dw3, db3, B, H2, W2, s)));
    // da2 = conv^T(dz3)
    hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w3, w3t, 64, 32, 1, 32);
    GWW_LAUNCH_CHECK();
    GWW_TRY((launch_conv3x3<64, 32, QM_RAW>(dz3, pl_z3, w3t, nullptr, nullptr, da2, 0, B, H2, W2, s, 64, nullptr, nullptr, 32, 0)));
    // conv2 again -> dz2
    GWW_TRY((launch_conv3x3<16, 32, QM_BWD_POOL>(a1, pl_a1, w2f, b2f, nullptr, dz2, pl_z2, B, H1, W1, s, 16, da2)));
    GWW_TRY((launch_wgrad<16, 32>(a1, pl_a1, dz2, pl_z2, dw2, db2, B, H1, W1, s)));
    // da1 = conv^T(dz2): 16 real output channels, padded to the MFMA's 32 rows
    hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w2, w2t, 32, c1p, 1, 16);
    GWW_LAUNCH_CHECK();
    GWW_TRY((launch_conv3x3<32, 32, QM_RAW>(dz2, pl_z2, w2t, nullptr, nullptr, da1, 0, B, H1, W1, s, 32, nullptr, nullptr, 16, 0)));
    hipLaunchKernelGGL(k_qcnn_conv1_bwd<16>, g1, dim3(256), 0, s, qspec, w1f, b1f, da1, dw1, db1, H, W);
    GWW_LAUNCH_CHECK();
  } else {
    GWW_TRY((launch_conv3x3<32, 64, QM_POOL>(a1, pl_a1, w2f, b2f, nullptr, a2, pl_a2, B, H1, W1, s)));
    GWW_TRY((launch_conv3x3<64, 128, QM_BWD_FUSE>(a2, pl_a2, w3f, b3f, w4f, dz3, pl_z3, B, H2, W2, s, 64, dy, rz3)));
    GWW_TRY((launch_wgrad<64, 128>(a2, pl_a2, dz3, pl_z3, dw3, db3, B, H2, W2, s)));
    // da2 = conv^T(dz3): the 128 input channels of the transposed convolution in two slices of 64 (the LDS patch of a
    // 128-channel input does not fit), the second accumulating
    const size_t half = (size_t)(64 / 32) * 9 * (64 / 16) * 2 * 1024 / 2;   // elements of one slice's fragments
    for (int sl = 0; sl < 2; ++sl) {
      hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w3 + (size_t)sl * 64 * c2 * 9, w3t + sl * half, 64, 64, 1, 64);
      GWW_LAUNCH_CHECK();
      GWW_TRY((launch_conv3x3<64, 64, QM_RAW>(dz3 + 64 * sl, pl_z3, w3t + sl * half, nullptr, nullptr, da2, 0, B, H2, W2, s, 128,
                                              nullptr, nullptr, 64, sl)));
    }
    GWW_TRY((launch_conv3x3<32, 64, QM_BWD_POOL>(a1, pl_a1, w2f, b2f, nullptr, dz2, pl_z2, B, H1, W1, s, 32, da2)));
    GWW_TRY((launch_wgrad<32, 64>(a1, pl_a1, dz2, pl_z2, dw2, db2, B, H1, W1, s)));
    hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w2, w2t, 64, 32, 1, 32);
    GWW_LAUNCH_CHECK();
    GWW_TRY((launch_conv3x3<64, 32, QM_RAW>(dz2, pl_z2, w2t, nullptr, nullptr, da1, 0, B, H1, W1, s, 64, nullptr, nullptr, 32, 0)));
    hipLaunchKernelGGL(k_qcnn_conv1_bwd<32>, g1, dim3(256), 0, s, qspec, w1f, b1f, da1, dw1, db1, H, W);
    GWW_LAUNCH_CHECK();
  }
  // dw4[c] = sum relu(z3[c]) dy
  hipLaunchKernelGGL(k_qcnn_chan_sum, dim3(512), dim3(256), 0, s, rz3, pl_z3, n2, c3, dw4);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}
