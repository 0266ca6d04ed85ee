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
template <int CIN, int COUT, bool POOL, bool WLDS>
__global__ __launch_bounds__(256) void k_qcnn_conv3x3(const unsigned short* __restrict__ act, long act_plane,
                                                      const unsigned short* __restrict__ wfrag,
                                                      const float* __restrict__ bias, const float* __restrict__ w4,
                                                      void* __restrict__ outp, long out_plane, int H, int W) {
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
      v = *reinterpret_cast<const u32x4*>(act + pl * act_plane + (((long)b * H + gy) * W + gx) * CIN + 8 * c);
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
      const float bv = bias[32 * ct + (e & 3) + 8 * (e >> 2) + 4 * h];
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
__global__ __launch_bounds__(256) void k_qcnn_pack(const float* __restrict__ w, unsigned short* __restrict__ out, int CIN,
                                                   int COUT) {
  const int KS = CIN / 16, nfrag = (COUT / 32) * 9 * KS * 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)nfrag * 512; i += (long)gridDim.x * 256) {
    const int f = (int)(i >> 9), within = (int)(i & 511), lane = within >> 3, j = within & 7;
    const int plane = f & 1, s = (f >> 1) % KS, tap = ((f >> 1) / KS) % 9, ct = (f >> 1) / KS / 9;
    const int cout = 32 * ct + (lane & 31), cin = 16 * s + 8 * (lane >> 5) + j;
    const float v = w[((long)cout * CIN + cin) * 9 + tap];
    const unsigned short hi = f2bf(v);
    out[i] = plane ? f2bf(v - bf2f(hi)) : hi;
  }
}

template <int CIN, int COUT, bool POOL>
int launch_conv3x3(const unsigned short* act, long act_plane, const unsigned short* wfrag, const float* bias,
                   const float* w4, void* out, long out_plane, int B, int H, int W, hipStream_t s) {
  constexpr int patch = 2 * (QC_TR + 2) * QC_PW * CIN * 2;
  constexpr int wbytes = (COUT / 32) * 9 * (CIN / 16) * 2 * 1024;
  constexpr bool WLDS = patch + wbytes <= 128 * 1024;
  constexpr int lds = patch + (WLDS ? wbytes : 0);
  auto kern = k_qcnn_conv3x3<CIN, COUT, POOL, WLDS>;
  if (lds > 64 * 1024) GWW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(W / QC_TW, H / QC_TR, B), dim3(256), lds, s, act, act_plane, wfrag, bias, w4, out,
                     out_plane, H, W);
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
  hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w2, (unsigned short*)(base + p.w2), c1, c2);
  GWW_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_qcnn_pack, dim3(64), dim3(256), 0, s, w3, (unsigned short*)(base + p.w3), c2, c3);
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
    GWW_TRY((launch_conv3x3<16, 32, true>(a1, pl1, w2, b2, nullptr, a2, pl2, B, H1, W1, s)));
    GWW_TRY((launch_conv3x3<32, 64, false>(a2, pl2, w3, b3, w4, y, 0, B, H2, W2, s)));
  } else {
    GWW_TRY((launch_conv3x3<32, 64, true>(a1, pl1, w2, b2, nullptr, a2, pl2, B, H1, W1, s)));
    GWW_TRY((launch_conv3x3<64, 128, false>(a2, pl2, w3, b3, w4, y, 0, B, H2, W2, s)));
  }
  return GWW_OK;
}
