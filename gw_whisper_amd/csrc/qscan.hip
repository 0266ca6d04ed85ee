// Q-transform front end #2: the constant-Q tile energies and the spectrogram that
// ml4gw's QScan feeds to the reference's QTransformAdapter (MLGWSC-1/train.py:117-122,135-154).
//
// PARITY UNPINNED: ml4gw is neither vendored nor pinned by the reference and is not installed anywhere this build
// runs; these kernels follow the restatement in oracle/qscan.py (Chatterji 2004 tiling as in GWpy / ml4gw).
//
// Dataflow (fp32):
//   strain [B, N]  --gww_gemm_f32 against the forward-normalised real-DFT matrix-->  fseries [B, 2 (N/2+1)] (re, im)
//   k_qscan_tiles : one workgroup = one (Q plane, frequency row) x 4 batch items.  The bisquare-windowed slice of
//                   the spectrum (<= 695 bins) is staged in LDS; every thread owns ntiles / 128 time samples and
//                   evaluates  z[t] = sum_k c_k e^{2 pi i k t / ntiles}  with twiddles from an LDS table (the
//                   zero padding + ifftshift of the reference only add a unit-modulus phase); energies go to LDS,
//                   a bitonic sort gives the median (mean of the two middle values: ntiles is a power of two),
//                   the normalised energies are written once and the plane's maximum is tracked with atomicMax.
//   k_qscan_interp: picks the plane with the largest normalised energy over the WHOLE batch (QScan's rule) on the
//                   device and resamples its [rows, ntiles(row)] energies to [F, T] with PyTorch's bicubic rules
//                   (align_corners = false, A = -0.75): time first, per row, then frequency.
#include "common.h"

namespace gww {

namespace {
constexpr int QS_THREADS = 128;
constexpr int QS_NB = 4;                 // batch items per workgroup
constexpr int QS_MAXW = 704;             // largest window (bins), rounded up

// row table entry (ints): plane, ntiles, windowsize, first data index, energy offset, window offset
constexpr int QR_PLANE = 0, QR_N = 1, QR_WS = 2, QR_IDX0 = 3, QR_EOFF = 4, QR_WOFF = 5, QR_STRIDE = 6;

template <int TPT>   // time samples per thread = ntiles / 128
__global__ __launch_bounds__(QS_THREADS) void k_qscan_tiles(const float* __restrict__ fseries, int ld, int B,
                                                          const int* __restrict__ rows,
                                                          const int* __restrict__ order, int row0,
                                                          const float* __restrict__ window, float* __restrict__ energy,
                                                          long e_total, unsigned int* __restrict__ plane_max) {
  constexpr int N = TPT * QS_THREADS;
  __shared__ float2 tw[N];                       // e^{2 pi i m / N}
  __shared__ float2 cs[QS_MAXW][QS_NB];          // windowed spectrum slice, 4 batch items side by side
  __shared__ float sortbuf[QS_NB][N];
  const int tid = threadIdx.x;
  const int* R = rows + (long)order[row0 + blockIdx.x] * QR_STRIDE;
  const int plane = R[QR_PLANE], ws = R[QR_WS], idx0 = R[QR_IDX0], eoff = R[QR_EOFF], woff = R[QR_WOFF];
  const int b0 = blockIdx.y * QS_NB;
  for (int m = tid; m < N; m += QS_THREADS) {
    float s, c;
    __sincosf(6.28318530717958647692f * (float)m / (float)N, &s, &c);
    tw[m] = make_float2(c, s);
  }
  for (int i = tid; i < ws * QS_NB; i += QS_THREADS) {
    const int k = i / QS_NB, bb = i - k * QS_NB;
    float2 v = make_float2(0.f, 0.f);
    if (b0 + bb < B) {
      const float w = window[woff + k];
      const float2 x = *reinterpret_cast<const float2*>(fseries + (long)(b0 + bb) * ld + 2 * (idx0 + k));
      v = make_float2(w * x.x, w * x.y);
    }
    cs[k][bb] = v;
  }
  __syncthreads();
  float ar[TPT][QS_NB], ai[TPT][QS_NB];
#pragma unroll
  for (int j = 0; j < TPT; ++j)
#pragma unroll
    for (int bb = 0; bb < QS_NB; ++bb) ar[j][bb] = ai[j][bb] = 0.f;
  for (int k = 0; k < ws; ++k) {
    float2 c[QS_NB];
#pragma unroll
    for (int bb = 0; bb < QS_NB; ++bb) c[bb] = cs[k][bb];
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
      const int t = tid + QS_THREADS * j;
      const float2 w = tw[(k * t) & (N - 1)];
#pragma unroll
      for (int bb = 0; bb < QS_NB; ++bb) {
        ar[j][bb] = fmaf(c[bb].x, w.x, fmaf(-c[bb].y, w.y, ar[j][bb]));
        ai[j][bb] = fmaf(c[bb].x, w.y, fmaf(c[bb].y, w.x, ai[j][bb]));
      }
    }
  }
  const float inv_n2 = 1.0f / ((float)N * (float)N);
  float e[TPT][QS_NB];
#pragma unroll
  for (int j = 0; j < TPT; ++j)
#pragma unroll
    for (int bb = 0; bb < QS_NB; ++bb) {
      e[j][bb] = (ar[j][bb] * ar[j][bb] + ai[j][bb] * ai[j][bb]) * inv_n2;
      sortbuf[bb][tid + QS_THREADS * j] = e[j][bb];
    }
  __syncthreads();
  // bitonic sort of the four arrays (N a power of two >= 128)
  for (int k2 = 2; k2 <= N; k2 <<= 1) {
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int i = tid; i < N; i += QS_THREADS) {
        const int ixj = i ^ j2;
        if (ixj > i) {
          const bool up = (i & k2) == 0;
#pragma unroll
          for (int bb = 0; bb < QS_NB; ++bb) {
            const float a = sortbuf[bb][i], b = sortbuf[bb][ixj];
            if ((a > b) == up) {
              sortbuf[bb][i] = b;
              sortbuf[bb][ixj] = a;
            }
          }
        }
      }
      __syncthreads();
    }
  }
  float vmax = 0.f;
#pragma unroll
  for (int bb = 0; bb < QS_NB; ++bb) {
    if (b0 + bb >= B) continue;
    const float med = 0.5f * (sortbuf[bb][N / 2 - 1] + sortbuf[bb][N / 2]);
    const float inv = 1.0f / med;
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
      const float v = e[j][bb] * inv;
      energy[(long)(b0 + bb) * e_total + eoff + tid + QS_THREADS * j] = v;
      vmax = fmaxf(vmax, v);
    }
  }
  vmax = wave_max(vmax);
  if ((tid & 63) == 0) atomicMax(plane_max + plane, __float_as_uint(vmax));   // energies are >= 0: uint order == float order
}

__device__ __forceinline__ void cubic4(float t, float w[4]) {   // PyTorch upsample_bicubic2d, A = -0.75
  const float A = -0.75f;
  auto c1 = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
  auto c2 = [&](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
  w[0] = c2(t + 1.f); w[1] = c1(t); w[2] = c1(1.f - t); w[3] = c2(2.f - t);
}

__global__ __launch_bounds__(256) void k_qscan_interp(const float* __restrict__ energy, long e_total,
                                                      const int* __restrict__ rows, const int* __restrict__ plane_rows,
                                                      int n_planes, const unsigned int* __restrict__ plane_max,
                                                      int F, int T, float* __restrict__ out, int* __restrict__ chosen) {
  // plane with the largest energy over the whole batch (first one on ties, as torch.argmax)
  int best = 0;
  unsigned int bm = plane_max[0];
  for (int p = 1; p < n_planes; ++p)
    if (plane_max[p] > bm) { bm = plane_max[p]; best = p; }
  if (chosen && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *chosen = best;
  const int r0 = plane_rows[2 * best], nr = plane_rows[2 * best + 1];
  const int b = blockIdx.y;
  const float* E = energy + (long)b * e_total;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < F * T; o += gridDim.x * 256) {
    const int fo = o / T, to = o - fo * T;
    // frequency axis: nr rows -> F
    float wf[4];
    int fi;
    if (nr == F) { fi = fo; wf[0] = 0.f; wf[1] = 1.f; wf[2] = 0.f; wf[3] = 0.f; }
    else {
      const float src = ((float)fo + 0.5f) * ((float)nr / (float)F) - 0.5f;
      const float fl = floorf(src);
      fi = (int)fl;
      cubic4(src - fl, wf);
    }
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int rr = fi - 1 + a;
      rr = rr < 0 ? 0 : (rr > nr - 1 ? nr - 1 : rr);
      const int* R = rows + (long)(r0 + rr) * QR_STRIDE;
      const int n = R[QR_N];
      const float* er = E + R[QR_EOFF];
      float v;
      if (n == T) v = er[to];
      else {
        const float src = ((float)to + 0.5f) * ((float)n / (float)T) - 0.5f;
        const float fl = floorf(src);
        const int ti = (int)fl;
        float wt[4];
        cubic4(src - fl, wt);
        v = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int tt = ti - 1 + c;
          tt = tt < 0 ? 0 : (tt > n - 1 ? n - 1 : tt);
          v = fmaf(wt[c], er[tt], v);
        }
      }
      acc = fmaf(wf[a], v, acc);
    }
    out[((long)b * F + fo) * T + to] = acc;
  }
}
}  // namespace

}  // namespace gww

using namespace gww;

// fseries: fp32 [B, ld] = (re, im) pairs of the forward-normalised one-sided spectrum (positive frequencies doubled).
// rows: device int table [n_rows][6] = plane, ntiles, windowsize, first data index, energy offset, window offset,
// plane by plane in frequency order; order: device int [n_rows], the row indices sorted by ntiles; class_ranges
// (HOST): for the five ntiles classes 128, 256, 512, 1024, 2048 the [first, last) positions in `order`.  energy: fp32 [B, e_total] (normalised tile energies); plane_max: uint [n_planes], zeroed by the call.
extern "C" int gww_qscan_energy_f32(const float* fseries, int ld, int B, const int* rows, const int* order,
                                    const int* class_ranges_host, const float* window, float* energy, long e_total, unsigned int* plane_max,
                                    int n_planes, void* stream) {
  GWW_REQUIRE(fseries && rows && order && class_ranges_host && window && energy && plane_max,
              "gww_qscan_energy_f32: NULL argument");
  GWW_REQUIRE(B >= 0 && n_planes > 0 && ld % 2 == 0, "gww_qscan_energy_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  GWW_HIP(hipMemsetAsync(plane_max, 0, sizeof(unsigned int) * n_planes, s));
  if (B == 0) return GWW_OK;
  const unsigned by = (unsigned)cdiv(B, QS_NB);
#define GWW_QS(TPT, cls)                                                                                        \
  do {                                                                                                          \
    const int r0 = class_ranges_host[2 * cls], r1 = class_ranges_host[2 * cls + 1];                             \
    if (r1 > r0)                                                                                                \
      hipLaunchKernelGGL((k_qscan_tiles<TPT>), dim3((unsigned)(r1 - r0), by), dim3(QS_THREADS), 0, s, fseries, ld, B, \
                         rows, order, r0, window, energy, e_total, plane_max);                                         \
  } while (0)
  GWW_QS(1, 0); GWW_QS(2, 1); GWW_QS(4, 2); GWW_QS(8, 3); GWW_QS(16, 4);
#undef GWW_QS
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// plane_rows: device int [n_planes][2] = first row, row count of each Q plane (rows of a plane contiguous, ordered by
// frequency).  out: fp32 [B, F, T].  chosen (optional, device int): index of the selected plane.
extern "C" int gww_qscan_interp_f32(const float* energy, long e_total, const int* rows, const int* plane_rows,
                                    int n_planes, const unsigned int* plane_max, int B, int F, int T, float* out,
                                    int* chosen, void* stream) {
  GWW_REQUIRE(energy && rows && plane_rows && plane_max && out, "gww_qscan_interp_f32: NULL argument");
  GWW_REQUIRE(B >= 0 && F > 0 && T > 0 && n_planes > 0, "gww_qscan_interp_f32: bad argument");
  if (B == 0) return GWW_OK;
  hipLaunchKernelGGL(k_qscan_interp, dim3((unsigned)cdiv((long)F * T, 256 * 4), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, energy, e_total, rows, plane_rows, n_planes, plane_max, F, T, out, chosen);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Tail of the reference's QTransformAdapter (MLGWSC-1/train.py:146-153, inference.py:345-350), one kernel:
//     y = AdaptiveAvgPool2d((F, T))(cnn_out);  y = scale * y + bias;  y = y * film_gamma[i] + film_beta[i]
// cnn_out fp32 [B, Hin, Win] (one channel: the 1x1 convolution's output) -> out fp32 [.., F, T] written straight into
// the stacked [B, D, F, T] feature tensor (batch stride given; detector i selects the base pointer on the host): the
// three elementwise passes and torch.stack of the reference shape (4 x 960 KB read + written per window and detector)
// become one 960 KB write.  HBM-write-bound: 4 F T bytes out per (window, detector), Hin Win 4 bytes in.
// PyTorch's adaptive pooling regions: rows [floor(f Hin / F), ceil((f + 1) Hin / F)), columns likewise.
namespace gww {
namespace {
constexpr int QT_MAXW = 4096;
__global__ __launch_bounds__(256) void k_qadapter_tail(const float* __restrict__ y, int Hin, int Win,
                                                       const float* __restrict__ scale, const float* __restrict__ bias,
                                                       const float* __restrict__ gamma_i, const float* __restrict__ beta_i,
                                                       float* __restrict__ out, long out_bstride, int F, int Tn) {
  __shared__ float rp[QT_MAXW];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int r0 = (f * Hin) / F, r1 = ((f + 1) * Hin + F - 1) / F;
  const float* yb = y + (long)b * Hin * Win;
  for (int c = tid; c < Win; c += 256) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += yb[(long)r * Win + c];
    rp[c] = s;
  }
  __syncthreads();
  const float g = gamma_i[0];
  const float a = scale[0] * g, c0 = bias[0] * g + beta_i[0];
  const float inv_r = 1.0f / (float)(r1 - r0);
  float* orow = out + (long)b * out_bstride + (long)f * Tn;
  for (int t4 = tid; t4 < Tn / 4; t4 += 256) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = 4 * t4 + e;
      const int k0 = (int)(((long)t * Win) / Tn), k1 = (int)((((long)t + 1) * Win + Tn - 1) / Tn);
      float s = 0.f;
      for (int k = k0; k < k1; ++k) s += rp[k];
      v[e] = fmaf(s * (inv_r / (float)(k1 - k0)), a, c0);
    }
    *reinterpret_cast<float4*>(orow + 4 * t4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
}  // namespace
}  // namespace gww

// y: fp32 [B, Hin, Win]; scale, bias, gamma_i, beta_i: device scalars (gamma_i / beta_i already point at detector i);
// out: fp32, element (b, f, t) at out[b * out_batch_stride + f * T + t].  T % 4 == 0, Win <= 4096, out 16-byte aligned.
extern "C" int gww_qadapter_tail_f32(const float* y, int B, int Hin, int Win, const float* scale, const float* bias,
                                     const float* gamma_i, const float* beta_i, float* out, long out_batch_stride,
                                     int F, int T, void* stream) {
  GWW_REQUIRE(y && scale && bias && gamma_i && beta_i && out, "gww_qadapter_tail_f32: NULL argument");
  GWW_REQUIRE(B >= 0 && Hin > 0 && Win > 0 && Win <= gww::QT_MAXW && F > 0 && F <= 65535 && T > 0 && T % 4 == 0,
              "gww_qadapter_tail_f32: bad shape B=%d Hin=%d Win=%d F=%d T=%d", B, Hin, Win, F, T);
  GWW_REQUIRE((((uintptr_t)out) & 15) == 0 && out_batch_stride % 4 == 0, "gww_qadapter_tail_f32: out must be 16-byte aligned");
  if (B == 0) return GWW_OK;
  for (int b0 = 0; b0 < B; b0 += 65535) {   // gridDim.y limit
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    hipLaunchKernelGGL(gww::k_qadapter_tail, dim3((unsigned)F, (unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       y + (long)b0 * Hin * Win, Hin, Win, scale, bias, gamma_i, beta_i, out + (long)b0 * out_batch_stride,
                       out_batch_stride, F, T);
  }
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}
