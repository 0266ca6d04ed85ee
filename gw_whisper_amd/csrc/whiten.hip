// Whitening of the search pipeline's strain (reference MLGWSC-1/inference.py:56-137 -> PyCBC
// 2.4.0 welch / inverse_spectrum_truncation; PARITY UNPINNED: PyCBC is not installed anywhere this build runs, the
// kernels follow oracle/whiten.py).  Device side of gw_whisper_amd/whiten.py:
//
//   k_welch_power    |rDFT(hann . segment)|^2 with DC / Nyquist halved, from the (re, im) rows a fp32-MFMA GEMM against
//                    the windowed real-DFT matrix produced                                  (HBM-bound, elementwise)
//   k_column_median  per-frequency MEDIAN over the Welch segments (numpy.median: mean of the two middle values for an
//                    even count) by a 4-pass, 8-bit radix select on the float bit patterns (power >= 0: uint order ==
//                    float order); one workgroup per frequency bin, no sort, no scratch        (L2-resident re-reads)
//   k_fir_f32        the whitening filter applied in the TIME domain: inverse-spectrum truncation makes it a short
//                    FIR (the reference's max_filter_duration = 0.25 s = 512 taps at 2048 Hz, plus the tail the
//                    |.| of its spectrum adds), so  white[n] = sum_u g[u] x[n + u]  streams the strain once through
//                    LDS with O(1) extra memory -- 2 N (2 K + 1) flops on the fp32 VALU (15 GFLOP per detector-hour:
//                    an N-point FFT pair is not needed and nothing of size N is held besides the strain itself)
#include "common.h"

namespace gww {
namespace {

__global__ __launch_bounds__(256) void k_welch_power(const float* __restrict__ spec, long ld, float* __restrict__ pw,
                                                     long n_seg, int n_bins, float scale) {
  const long total = n_seg * n_bins;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long s = i / n_bins;
    const int k = (int)(i - s * n_bins);
    const float re = spec[s * ld + 2 * k], im = spec[s * ld + 2 * k + 1];
    float p = (re * re + im * im) * scale;
    if (k == 0 || k == n_bins - 1) p *= 0.5f;   // "halve the DC and Nyquist components" (pycbc.psd.welch)
    pw[i] = p;
  }
}

// value of rank `rank` (0-based, ascending) among col[0], col[stride], ..., n values, all >= 0
__device__ float radix_select(const float* __restrict__ col, long stride, long n, long rank, unsigned* hist) {
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned v = __float_as_uint(col[i * stride]);
      if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    // every thread walks the 256 buckets (uniform result, no broadcast needed)
    long acc = 0;
    unsigned b = 0;
    for (; b < 256; ++b) {
      const long c = hist[b];
      if (acc + c > rank) break;
      acc += c;
    }
    rank -= acc;
    prefix |= b << shift;
    mask |= 255u << shift;
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

__global__ __launch_bounds__(256) void k_column_median(const float* __restrict__ pw, long n_seg, int n_bins,
                                                       float* __restrict__ med) {
  __shared__ unsigned hist[256];
  const int k = blockIdx.x;
  const float hi = radix_select(pw + k, n_bins, n_seg, n_seg / 2, hist);
  float m = hi;
  if ((n_seg & 1) == 0) m = 0.5f * (radix_select(pw + k, n_bins, n_seg, n_seg / 2 - 1, hist) + hi);
  if (threadIdx.x == 0) med[k] = m;
}

// out[d][n] = sum_{u = 0}^{taps - 1} g[d][u] xp[d][n + u],  n in [0, n_out); taps % 4 == 1 is NOT required: the host
// pads g with zeros to a multiple of 4 (taps4) and xp accordingly.  One workgroup = FIR_BLK outputs, 4 per thread.
constexpr int FIR_BLK = 1024;
__global__ __launch_bounds__(256) void k_fir_f32(const float* __restrict__ xp, long xp_stride, const float* __restrict__ g,
                                                 int taps4, float* __restrict__ out, long out_stride, long n_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                          // FIR_BLK + taps4 samples
  float* gs = sm + FIR_BLK + taps4;        // taps4 coefficients
  const int d = blockIdx.y, tid = threadIdx.x;
  const long n0 = (long)blockIdx.x * FIR_BLK;
  const float* xd = xp + (long)d * xp_stride + n0;
  const long avail = n_out - n0 < FIR_BLK ? n_out - n0 : FIR_BLK;       // outputs of this block
  for (int i = tid; i < FIR_BLK + taps4; i += 256) xs[i] = i < avail + taps4 ? xd[i] : 0.f;   // xp holds n_out + taps4 samples
  for (int i = tid; i < taps4; i += 256) gs[i] = g[(long)d * taps4 + i];
  __syncthreads();
  const int base = 4 * tid;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  float4 w = *reinterpret_cast<const float4*>(xs + base);   // window xs[base + u .. base + u + 3]
  for (int u = 0; u < taps4; u += 4) {
    const float4 c = *reinterpret_cast<const float4*>(gs + u);
    const float4 nx = *reinterpret_cast<const float4*>(xs + base + u + 4);
    a0 = fmaf(c.x, w.x, a0); a1 = fmaf(c.x, w.y, a1); a2 = fmaf(c.x, w.z, a2); a3 = fmaf(c.x, w.w, a3);
    a0 = fmaf(c.y, w.y, a0); a1 = fmaf(c.y, w.z, a1); a2 = fmaf(c.y, w.w, a2); a3 = fmaf(c.y, nx.x, a3);
    a0 = fmaf(c.z, w.z, a0); a1 = fmaf(c.z, w.w, a1); a2 = fmaf(c.z, nx.x, a2); a3 = fmaf(c.z, nx.y, a3);
    a0 = fmaf(c.w, w.w, a0); a1 = fmaf(c.w, nx.x, a1); a2 = fmaf(c.w, nx.y, a2); a3 = fmaf(c.w, nx.z, a3);
    w = nx;
  }
  float* od = out + (long)d * out_stride + n0;
  if (base + 3 < avail) *reinterpret_cast<float4*>(od + base) = make_float4(a0, a1, a2, a3);
  else {
    if (base < avail) od[base] = a0;
    if (base + 1 < avail) od[base + 1] = a1;
    if (base + 2 < avail) od[base + 2] = a2;
  }
}
}  // namespace
}  // namespace gww

using namespace gww;

// spec: fp32 [n_seg, ld] rows of (re, im) pairs, bins 0 .. n_bins - 1; power: fp32 [n_seg, n_bins] = (re^2 + im^2) * scale,
// DC and Nyquist halved.
extern "C" int gww_welch_power_f32(const float* spec, long ld, long n_seg, int n_bins, float scale, float* power, void* stream) {
  GWW_REQUIRE(spec && power && n_seg >= 0 && n_bins > 1 && ld >= 2L * n_bins, "gww_welch_power_f32: bad argument");
  if (n_seg == 0) return GWW_OK;
  long blocks = cdiv(n_seg * n_bins, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_welch_power, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, spec, ld, power, n_seg, n_bins, scale);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// power: fp32 [n_seg, n_bins], every value >= 0 (and not NaN); median: fp32 [n_bins] = numpy.median(power, axis=0).
extern "C" int gww_column_median_f32(const float* power, long n_seg, int n_bins, float* median, void* stream) {
  GWW_REQUIRE(power && median && n_seg > 0 && n_bins > 0, "gww_column_median_f32: bad argument");
  hipLaunchKernelGGL(k_column_median, dim3((unsigned)n_bins), dim3(256), 0, (hipStream_t)stream, power, n_seg, n_bins, median);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}

// xp: fp32 [D, xp_stride] with at least n_out + taps4 valid samples per row; g: fp32 [D, taps4] (taps4 % 4 == 0, zero
// padded); out: fp32 [D, out_stride], out[d][n] = sum_u g[d][u] xp[d][n + u] for n < n_out.  16-byte aligned rows.
extern "C" int gww_fir_f32(const float* xp, long xp_stride, const float* g, int taps4, int D, float* out, long out_stride,
                           long n_out, void* stream) {
  GWW_REQUIRE(xp && g && out && D > 0 && D <= 65535 && taps4 > 0 && taps4 % 4 == 0 && taps4 <= 16384 && n_out >= 0,
              "gww_fir_f32: bad argument (taps4=%d D=%d)", taps4, D);
  GWW_REQUIRE(xp_stride % 4 == 0 && out_stride % 4 == 0 && ((((uintptr_t)xp) | ((uintptr_t)out) | ((uintptr_t)g)) & 15) == 0,
              "gww_fir_f32: rows must be 16-byte aligned");
  if (n_out == 0) return GWW_OK;
  const size_t lds = (size_t)(FIR_BLK + 2 * taps4) * sizeof(float);
  GWW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fir_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_fir_f32, dim3((unsigned)cdiv(n_out, (long)FIR_BLK), (unsigned)D), dim3(256), lds, (hipStream_t)stream, xp,
                     xp_stride, g, taps4, out, out_stride, n_out);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}
