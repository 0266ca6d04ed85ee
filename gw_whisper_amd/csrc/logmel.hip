// Log-mel front end (K1 + K2 of SURVEY.md section 2.1) for gfx950.
//
// Replaces WhisperFeatureExtractor.__call__ (reference call sites
// Signal_vs_Noise/src/dataset.py:20-21; HF:feature_extraction_whisper.py:135-168):
//   zero-pad to 480000 -> reflect-pad 200 -> 400-pt periodic-Hann DFT, hop 160 ->
//   |X|^2 (frames 0..2999) -> mel[80,201] @ P -> log10(max(.,1e-10)) ->
//   per-segment max -> max(x, max-8) -> (x+4)/4.
//
// HBM-bound: 64 KB in + 960 KB out per segment.  Two kernels:
//   k_logmel_frames    DFT + mel + log10 for the LIVE frames only (frames that can
//                      see a non-zero sample: ceil((n+200)/160), 102 for 1 s), raw
//                      log-mel written in place, per-segment max via atomicMax
//   k_logmel_finalize  streams the whole [80,3000] row: clamp to max-8, affine,
//                      constant fill of the dead frames (float4 stores)
// The DFT uses the real-input symmetry x[n] +- x[400-n] (half the MACs) with the
// 400-entry twiddle table and the windowed frames in LDS; fp32 throughout.
#include "common.h"

#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace gww {

constexpr int kNfft = 400;
constexpr int kHop = 160;
constexpr int kNfreq = 201;
constexpr int kNmel = 80;
constexpr int kFrames = 3000;
constexpr int kChunk = 480000;
constexpr int kFT = 8;                      // frames per workgroup
constexpr int kSpan = kHop * (kFT - 1) + kNfft;  // 1520 samples per tile

struct Tables {
  float* win;      // [400]
  float* cost;     // [400] cos(2 pi k / 400)
  float* sint;     // [400]
  float* fbT;      // [201][80] filterbank, transposed so consecutive mels are contiguous
};

__device__ __forceinline__ unsigned int fkey(float f) {
  unsigned int b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned int k) {
  unsigned int b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

// grid (tiles, n_seg), 256 threads
__global__ __launch_bounds__(256) void k_logmel_frames(const float* __restrict__ wave, long stride,
                                                       int n_eff, int live, Tables tb,
                                                       float* __restrict__ out,
                                                       unsigned int* __restrict__ seg_max) {
  __shared__ __attribute__((aligned(16))) float xs[kSpan];
  __shared__ __attribute__((aligned(16))) float ev[kFT][204];   // ev[f][0..200]
  __shared__ __attribute__((aligned(16))) float od[kFT][204];   // od[f][0..199], od[.][0]=0
  __shared__ float ct[kNfft];
  __shared__ float st[kNfft];
  __shared__ float pw[kFT][kNfreq + 3];
  __shared__ float red[4];

  const int tid = threadIdx.x;
  const int seg = blockIdx.y;
  const int t0 = blockIdx.x * kFT;
  const float* x = wave + (long)seg * stride;

  // stage the samples of this tile (reflect at both ends of the 480000 buffer, zeros past n_eff)
  for (int i = tid; i < kSpan; i += 256) {
    int p = t0 * kHop - kNfft / 2 + i;       // index into the unpadded 480000 buffer
    if (p < 0) p = -p;
    if (p >= kChunk) p = 2 * (kChunk - 1) - p;
    xs[i] = (p < n_eff) ? x[p] : 0.0f;
  }
  for (int i = tid; i < kNfft; i += 256) {
    ct[i] = tb.cost[i];
    st[i] = tb.sint[i];
  }
  __syncthreads();
  // windowed even / odd parts
  for (int i = tid; i < kFT * 204; i += 256) {
    int f = i / 204, n = i - f * 204;
    float e = 0.f, o = 0.f;
    if (n <= 200) {
      float a = xs[f * kHop + n] * tb.win[n];
      if (n == 0 || n == 200) {
        e = a;
      } else {
        float b = xs[f * kHop + kNfft - n] * tb.win[kNfft - n];
        e = a + b;
        o = a - b;
      }
    }
    ev[f][n] = e;
    od[f][n] = o;
  }
  __syncthreads();

  if (tid < kNfreq) {
    const int k = tid;
    float re[kFT], im[kFT];
#pragma unroll
    for (int f = 0; f < kFT; ++f) re[f] = im[f] = 0.f;
    int idx = 0;  // (k * n) mod 400
    for (int n = 0; n < 204; n += 4) {
      float c[4], s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[j] = ct[idx];
        s[j] = st[idx];
        idx += k;
        if (idx >= kNfft) idx -= kNfft;
      }
#pragma unroll
      for (int f = 0; f < kFT; ++f) {
        const float4 e4 = *reinterpret_cast<const float4*>(&ev[f][n]);
        const float4 o4 = *reinterpret_cast<const float4*>(&od[f][n]);
        re[f] = fmaf(e4.x, c[0], re[f]);
        re[f] = fmaf(e4.y, c[1], re[f]);
        re[f] = fmaf(e4.z, c[2], re[f]);
        re[f] = fmaf(e4.w, c[3], re[f]);
        im[f] = fmaf(o4.x, s[0], im[f]);
        im[f] = fmaf(o4.y, s[1], im[f]);
        im[f] = fmaf(o4.z, s[2], im[f]);
        im[f] = fmaf(o4.w, s[3], im[f]);
      }
    }
#pragma unroll
    for (int f = 0; f < kFT; ++f) pw[f][k] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();

  // mel projection + log10; thread -> (frame, mel) pairs, mel fastest for coalesced fbT reads
  float lmax = -INFINITY;
  for (int i = tid; i < kFT * kNmel; i += 256) {
    const int f = i / kNmel, m = i - f * kNmel;
    const int t = t0 + f;
    float acc = 0.f;
    for (int k = 0; k < kNfreq; ++k) acc = fmaf(tb.fbT[k * kNmel + m], pw[f][k], acc);
    // log10 in fp64, rounded once: ocml's log10f is 1 ulp off at 1e-10 (-10.000001) where
    // torch (and the oracle) give the correctly rounded -10.0 that the dead frames use
    const float lg = (float)log10((double)fmaxf(acc, 1e-10f));
    if (t < live) {
      out[((long)seg * kNmel + m) * kFrames + t] = lg;
      lmax = fmaxf(lmax, lg);
    }
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > -INFINITY) atomicMax(&seg_max[seg], fkey(m));
  }
}

// ---- the same stage on the fp32 matrix cores (default; GWW_LOGMEL_VALU=1 selects the kernel above) ----
// The VALU kernel issues ~8.5 k instructions per wave for 8 frames (one thread per frequency bin, 3264 FMAs each):
// instruction issue, not HBM, bounds it (273 us per 256 segments against ~60 us of traffic).  Here a workgroup
// owns 32 frames and both contractions are v_mfma_f32_32x32x2_f32 chains (exact fp32 products, fp32 accumulate):
//   re[f][k] = sum_n ev[f][n] cos(2 pi k n / 400),  im[f][k] = sum_n od[f][n] sin(...)      n = 0..200
//       A operand = ev / od rows from LDS (frame on M), B operand = ONE twiddle-table read per lane per MFMA
//       (index k n mod 400 advanced by 2 k per step) -- no [201 x 201] table exists anywhere;
//   mel[f][m] = sum_k P[f][k] fb[k][m]: A = the power tile in LDS, B = the filterbank read coalesced from L2.
// log10 (fp64, as above), the per-segment maximum, and a transpose through LDS so that every store instruction
// writes 32 consecutive frames (128 B) of a mel row instead of 4 bytes each into 64 different rows.
constexpr int kMT = 32;                           // frames per workgroup
constexpr int kMThreads = 512;                    // 8 waves: 7 DFT bin blocks + latency hiding for the row-wise phases
constexpr int kMSpan = kHop * (kMT - 1) + kNfft;  // 5360 samples per tile
constexpr int kEvS = 205;                         // row stride of ev / od / pw: odd -> conflict-free column reads

__global__ __launch_bounds__(kMThreads) void k_logmel_frames_mfma(const float* __restrict__ wav, long stride, int n_eff,
                                                            int live, Tables tb, float* __restrict__ out,
                                                            unsigned int* __restrict__ seg_max) {
  __shared__ __attribute__((aligned(16))) float xs[kMSpan];       // 21.4 KB; reused as the [80][33] output tile
  __shared__ float ev[kMT * kEvS];                               // 26.2 KB; reused as the power tile
  __shared__ float od[kMT * kEvS];
  __shared__ float ct[kNfft];
  __shared__ float st[kNfft];
  __shared__ float red[kMThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int seg = blockIdx.y;
  const int t0 = blockIdx.x * kMT;
  const float* x = wav + (long)seg * stride;

  for (int i = tid; i < kMSpan; i += kMThreads) {
    int p = t0 * kHop - kNfft / 2 + i;
    if (p < 0) p = -p;
    if (p >= kChunk) p = 2 * (kChunk - 1) - p;
    xs[i] = (p >= 0 && p < n_eff) ? x[p] : 0.0f;
  }
  for (int i = tid; i < kNfft; i += kMThreads) {
    ct[i] = tb.cost[i];
    st[i] = tb.sint[i];
  }
  __syncthreads();
  // windowed even / odd parts; columns 201..204 are zero (the k-loop runs to 202)
  for (int i = tid; i < kMT * kEvS; i += kMThreads) {
    const int f = i / kEvS, n = i - f * kEvS;
    float e = 0.f, o = 0.f;
    if (n <= 200) {
      const float a = xs[f * kHop + n] * tb.win[n];
      if (n == 0 || n == 200) {
        e = a;
      } else {
        const float b = xs[f * kHop + kNfft - n] * tb.win[kNfft - n];
        e = a + b;
        o = a - b;
      }
    }
    ev[i] = e;
    od[i] = o;
  }
  __syncthreads();

  // ---- DFT: 7 blocks of 32 bins, one per wave (wave 7 idles here); the re and im chains of a block share the twiddle
  // index.  acc[4 c + e] <-> frame 8 c + 4 hh + e, bin on the lane
  constexpr int NBLK = 7, NSTEP = 101;   // 2 n per step: n = 0 .. 201
  f32x16 re, im;
#pragma unroll
  for (int j = 0; j < 16; ++j) { re[j] = 0.f; im[j] = 0.f; }
  if (wave < NBLK) {
    int bin = 32 * wave + r;
    if (bin > 200) bin = 200;                  // padding lanes recompute bin 200; never stored
    int idx = (bin * hh) % kNfft;              // n = hh at step 0
    const int inc = (2 * bin) % kNfft;
    const float* evr = ev + r * kEvS + hh;
    const float* odr = od + r * kEvS + hh;
    auto step = [&](int s) {
      re = __builtin_amdgcn_mfma_f32_32x32x2f32(evr[2 * s], ct[idx], re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_32x32x2f32(odr[2 * s], st[idx], im, 0, 0, 0);
      idx += inc;
      idx -= idx >= kNfft ? kNfft : 0;
    };
    // a real loop (4 steps per trip): fully unrolled this is tens of KB of straight-line code
#pragma unroll 1
    for (int s = 0; s + 3 < NSTEP; s += 4) {
      step(s);
      step(s + 1);
      step(s + 2);
      step(s + 3);
    }
    step(NSTEP - 1);   // NSTEP = 101 = 4 * 25 + 1
  }
  __syncthreads();   // everyone is done reading ev / od
  // power tile pw[f][k] over ev; columns 201..204 zero
  float* pw = ev;
  {
    const int k = 32 * wave + r;
    if (wave < NBLK && k < kEvS) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 8 * c + 4 * hh + e;
          pw[f * kEvS + k] = k <= 200 ? re[4 * c + e] * re[4 * c + e] + im[4 * c + e] * im[4 * c + e] : 0.f;
        }
    }
  }
  __syncthreads();

  // ---- mel projection: 3 blocks of 32 mels on waves 0..2; acc[4 c + e] <-> frame 8 c + 4 hh + e, mel on the lane
  float* tile = xs;                            // [80][33] raw log-mel of this tile
  float lmax = -INFINITY;
  if (wave < 3) {
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    int m = 32 * wave + r;
    const bool live_m = m < kNmel;
    if (!live_m) m = kNmel - 1;
    const float* fb = tb.fbT + m;
    // k = 2 s + hh; k = 201 (last step, upper half-wave) meets a zero power column.  Four filterbank values are
    // requested before the four MFMAs that use them.
    const float* pwr = pw + r * kEvS + hh;
    const float* fbk = fb + (long)hh * kNmel;
#pragma unroll 1
    for (int s = 0; s + 3 < NSTEP; s += 4) {
      float b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = fbk[(long)(2 * (s + u)) * kNmel];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pwr[2 * (s + u)], b[u], acc, 0, 0, 0);
    }
    {
      const int k = 2 * (NSTEP - 1) + hh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pw[r * kEvS + k], fb[(k <= 200 ? k : 200) * kNmel], acc, 0, 0, 0);
    }
    if (live_m) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[m * 33 + 8 * c + 4 * hh + e] = acc[4 * c + e];
    }
  }
  __syncthreads();
  // log10 in fp64, rounded once (see the VALU kernel), spread over all 256 threads
  for (int i = tid; i < kNmel * kMT; i += kMThreads) {
    const int m = i >> 5, f = i & 31;
    const float lg = (float)log10((double)fmaxf(tile[m * 33 + f], 1e-10f));
    tile[m * 33 + f] = lg;
    if (t0 + f < live) lmax = fmaxf(lmax, lg);
  }
  lmax = wave_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int w = 1; w < kMThreads / 64; ++w) m = fmaxf(m, red[w]);
    if (m > -INFINITY) atomicMax(&seg_max[seg], fkey(m));
  }
  // whole 128-byte runs of a mel row per store instruction
  for (int i = tid; i < kNmel * kMT; i += kMThreads) {
    const int m = i >> 5, f = i & 31;
    if (t0 + f < live) out[((long)seg * kNmel + m) * kFrames + t0 + f] = tile[m * 33 + f];
  }
}

// grid (80, n_seg), 256 threads: one [3000] row per workgroup
__global__ __launch_bounds__(256) void k_logmel_finalize(float* __restrict__ out, int live,
                                                         const unsigned int* __restrict__ seg_max) {
  const int seg = blockIdx.y, m = blockIdx.x;
  float mx = fkey_inv(seg_max[seg]);
  mx = fmaxf(mx, live < kFrames ? -10.0f : mx);     // dead frames contribute log10(1e-10) = -10
  const float floor_v = mx - 8.0f;
  const float padv = (fmaxf(-10.0f, floor_v) + 4.0f) * 0.25f;
  float4* row = reinterpret_cast<float4*>(out + ((long)seg * kNmel + m) * kFrames);
  for (int i = threadIdx.x; i < kFrames / 4; i += 256) {
    const int t = i * 4;
    float4 v;
    if (t + 3 < live) {
      v = row[i];
      v.x = (fmaxf(v.x, floor_v) + 4.0f) * 0.25f;
      v.y = (fmaxf(v.y, floor_v) + 4.0f) * 0.25f;
      v.z = (fmaxf(v.z, floor_v) + 4.0f) * 0.25f;
      v.w = (fmaxf(v.w, floor_v) + 4.0f) * 0.25f;
    } else if (t >= live) {
      v = make_float4(padv, padv, padv, padv);
    } else {
      v = row[i];
      v.x = (t + 0 < live) ? (fmaxf(v.x, floor_v) + 4.0f) * 0.25f : padv;
      v.y = (t + 1 < live) ? (fmaxf(v.y, floor_v) + 4.0f) * 0.25f : padv;
      v.z = (t + 2 < live) ? (fmaxf(v.z, floor_v) + 4.0f) * 0.25f : padv;
      v.w = (t + 3 < live) ? (fmaxf(v.w, floor_v) + 4.0f) * 0.25f : padv;
    }
    row[i] = v;
  }
}

// ---- host side: tables (HF:audio_utils.py:448-520,638-731 slaney scale + norm) ----
static double hz2mel(double f) {
  return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0;
}
static double mel2hz(double m) {
  return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0;
}

}  // namespace gww

using namespace gww;

struct gww_frontend {
  Tables tb{};
  float* blob = nullptr;
};

extern "C" int gww_frontend_create(gww_frontend** out) {
  GWW_REQUIRE(out != nullptr, "gww_frontend_create: out is NULL");
  std::vector<float> h(3 * kNfft + kNfreq * kNmel);
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < kNfft; ++k) {
    h[k] = (float)(0.5 - 0.5 * cos(2.0 * pi * k / kNfft));  // torch.hann_window(400) periodic
    h[kNfft + k] = (float)cos(2.0 * pi * k / kNfft);
    h[2 * kNfft + k] = (float)sin(2.0 * pi * k / kNfft);
  }
  // filterbank [201][80] (already "transposed" relative to mel.T @ P)
  const double mel_min = hz2mel(0.0), mel_max = hz2mel(8000.0);
  std::vector<double> ff(kNmel + 2);
  for (int i = 0; i < kNmel + 2; ++i) ff[i] = mel2hz(mel_min + (mel_max - mel_min) * i / (kNmel + 1));
  for (int k = 0; k < kNfreq; ++k) {
    const double fk = 8000.0 * k / (kNfreq - 1);
    for (int m = 0; m < kNmel; ++m) {
      const double down = (fk - ff[m]) / (ff[m + 1] - ff[m]);
      const double up = (ff[m + 2] - fk) / (ff[m + 2] - ff[m + 1]);
      double v = down < up ? down : up;
      if (v < 0) v = 0;
      v *= 2.0 / (ff[m + 2] - ff[m]);
      h[3 * kNfft + k * kNmel + m] = (float)v;
    }
  }
  gww_frontend* fe = new gww_frontend();
  hipError_t e = hipMalloc(&fe->blob, h.size() * sizeof(float));
  if (e != hipSuccess) {
    delete fe;
    return fail(GWW_ERR_HIP, "hipMalloc(frontend tables) failed: %s", hipGetErrorString(e));
  }
  e = hipMemcpy(fe->blob, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(fe->blob);
    delete fe;
    return fail(GWW_ERR_HIP, "hipMemcpy(frontend tables) failed: %s", hipGetErrorString(e));
  }
  fe->tb.win = fe->blob;
  fe->tb.cost = fe->blob + kNfft;
  fe->tb.sint = fe->blob + 2 * kNfft;
  fe->tb.fbT = fe->blob + 3 * kNfft;
  *out = fe;
  return GWW_OK;
}

extern "C" void gww_frontend_destroy(gww_frontend* fe) {
  if (!fe) return;
  if (fe->blob) (void)hipFree(fe->blob);
  delete fe;
}

extern "C" int gww_logmel_f32(gww_frontend* fe, const float* wave, int n_seg, int n_samples,
                              long wave_stride, float* out, float* seg_max, void* stream) {
  GWW_REQUIRE(fe && wave && out && seg_max, "gww_logmel_f32: NULL argument");
  GWW_REQUIRE(n_seg >= 0 && n_samples >= 0, "gww_logmel_f32: negative size");
  GWW_REQUIRE(wave_stride >= (n_samples < kChunk ? n_samples : kChunk),
              "gww_logmel_f32: wave_stride %ld < n_samples %d", wave_stride, n_samples);
  if (n_seg == 0) return GWW_OK;
  hipStream_t s = (hipStream_t)stream;
  const int n_eff = n_samples < kChunk ? n_samples : kChunk;   // HF truncates at 30 s
  int live = kFrames;
  if (n_eff < kChunk - kNfft) {
    live = (n_eff + kNfft / 2 + kHop - 1) / kHop;
    if (live > kFrames) live = kFrames;
    if (live < 1) live = 1;
  }
  GWW_HIP(hipMemsetAsync(seg_max, 0, sizeof(float) * (size_t)n_seg, s));
  static const bool valu_kernel = lab_int("GWW_LOGMEL_VALU", 0) != 0;   // comparison aid (lab build)
  if (valu_kernel) {
    dim3 g1((unsigned)cdiv(live, kFT), (unsigned)n_seg);
    hipLaunchKernelGGL(k_logmel_frames, g1, dim3(256), 0, s, wave, wave_stride, n_eff, live, fe->tb, out,
                       reinterpret_cast<unsigned int*>(seg_max));
  } else {
    dim3 g1((unsigned)cdiv(live, kMT), (unsigned)n_seg);
    hipLaunchKernelGGL(k_logmel_frames_mfma, g1, dim3(kMThreads), 0, s, wave, wave_stride, n_eff, live, fe->tb, out,
                       reinterpret_cast<unsigned int*>(seg_max));
  }
  GWW_LAUNCH_CHECK();
  dim3 g2(kNmel, (unsigned)n_seg);
  hipLaunchKernelGGL(k_logmel_finalize, g2, dim3(256), 0, s, out, live,
                     reinterpret_cast<const unsigned int*>(seg_max));
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}
