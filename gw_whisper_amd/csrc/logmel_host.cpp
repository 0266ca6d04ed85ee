// Host (CPU) log-mel front end: the fork-safe twin of gww_logmel_f32.
//
// Replaces WhisperFeatureExtractor.__call__ where the reference calls it from forked DataLoader workers
// (Signal_vs_Noise/src/dataset.py:12,20-21 under src/train.py:224-225, num_workers 12): a worker may not touch
// the GPU, so this entry point is plain C++ -- no HIP call, no device memory, no global mutable state (the tables
// are function-local statics, built once, thread safe since C++11).  Same arithmetic as
// HF:models/whisper/feature_extraction_whisper.py:135-168: zero-pad to 480000 -> reflect-pad 200 -> 400-point
// periodic-Hann DFT, hop 160 -> |X|^2 (frames 0..2999) -> mel[80,201] @ P -> log10(max(., 1e-10)) -> per-segment
// max -> max(x, max - 8) -> (x + 4) / 4, with the same exact shortcut as the device kernel: frames that only see
// zero padding are the one constant HF produces for them.
//
// The DFT is a 200-point complex mixed-radix FFT (5 x 5 x 4 x 2, decimation in time) of the even / odd samples of
// the windowed frame followed by the real-input untangling step; double precision, rounded to fp32 once per mel.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/gww.h"

namespace gww {
int fail(int code, const char* fmt, ...);   // elementwise.hip (per-thread error text)
}

namespace {

constexpr int kNfft = 400, kHalf = 200, kHop = 160, kNfreq = 201, kNmel = 80, kFrames = 3000, kChunk = 480000;
constexpr double kPi = 3.14159265358979323846;

struct Cpx {
  double re, im;
};

struct HostTables {
  double win[kNfft];
  Cpx w200[kHalf];        // exp(-2 pi i k / 200)
  Cpx w400[kHalf + 1];    // exp(-2 pi i k / 400), k = 0..200
  // sparse filterbank: HF's dense [201, 80] slaney triangles touch at most a few mels per bin
  int fb_first[kNfreq], fb_count[kNfreq];
  std::vector<float> fb_val;   // the fp32-cast weights HF multiplies with
  std::vector<int> fb_ofs;

  static double hz2mel(double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; }
  static double mel2hz(double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; }

  HostTables() {
    for (int k = 0; k < kNfft; ++k) win[k] = (double)(float)(0.5 - 0.5 * cos(2.0 * kPi * k / kNfft));
    for (int k = 0; k < kHalf; ++k) w200[k] = {cos(2.0 * kPi * k / kHalf), -sin(2.0 * kPi * k / kHalf)};
    for (int k = 0; k <= kHalf; ++k) w400[k] = {cos(2.0 * kPi * k / kNfft), -sin(2.0 * kPi * k / kNfft)};
    // HF:audio_utils.py:638-731 (slaney scale + slaney norm, 0..8000 Hz)
    const double mel_min = hz2mel(0.0), mel_max = hz2mel(8000.0);
    double ff[kNmel + 2];
    for (int i = 0; i < kNmel + 2; ++i) ff[i] = mel2hz(mel_min + (mel_max - mel_min) * i / (kNmel + 1));
    fb_ofs.resize(kNfreq);
    for (int k = 0; k < kNfreq; ++k) {
      const double fk = 8000.0 * k / (kNfreq - 1);
      fb_first[k] = 0;
      fb_count[k] = 0;
      fb_ofs[k] = (int)fb_val.size();
      int last = -1;
      float row[kNmel];
      for (int m = 0; m < kNmel; ++m) {
        const double down = (fk - ff[m]) / (ff[m + 1] - ff[m]);
        const double up = (ff[m + 2] - fk) / (ff[m + 2] - ff[m + 1]);
        double v = down < up ? down : up;
        if (v < 0) v = 0;
        v *= 2.0 / (ff[m + 2] - ff[m]);
        row[m] = (float)v;
        if (row[m] != 0.0f) {
          if (last < 0) fb_first[k] = m;
          last = m;
        }
      }
      if (last >= 0) {
        fb_count[k] = last - fb_first[k] + 1;
        for (int m = fb_first[k]; m <= last; ++m) fb_val.push_back(row[m]);
      }
    }
  }
};

const HostTables& tables() {
  static const HostTables t;
  return t;
}

// out[0..n) = DFT_n(in[0], in[stride], ...); n divides 200, twiddles from the 200-entry table.
void fft_rec(int n, const Cpx* in, int stride, Cpx* out, const HostTables& tb) {
  if (n == 1) {
    out[0] = in[0];
    return;
  }
  const int p = (n % 5 == 0) ? 5 : (n % 4 == 0) ? 4 : 2;
  const int m = n / p;
  for (int q = 0; q < p; ++q) fft_rec(m, in + (long)q * stride, stride * p, out + q * m, tb);
  const int tw = kHalf / n;   // w_n^j = w200[j * tw]
  for (int k = 0; k < m; ++k) {
    Cpx t[5];
    for (int q = 0; q < p; ++q) {
      const Cpx a = out[q * m + k];
      const Cpx w = tb.w200[(q * k * tw) % kHalf];
      t[q] = {a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re};
    }
    for (int r = 0; r < p; ++r) {
      double sr = t[0].re, si = t[0].im;
      for (int q = 1; q < p; ++q) {
        const Cpx w = tb.w200[((q * r) % p) * (kHalf / p)];
        sr += t[q].re * w.re - t[q].im * w.im;
        si += t[q].re * w.im + t[q].im * w.re;
      }
      out[k + r * m] = {sr, si};
    }
  }
}

// power[k] = |rfft(frame)[k]|^2, k = 0..200, of one windowed 400-sample frame
void frame_power(const double* frame, double* power, const HostTables& tb) {
  Cpx z[kHalf], Z[kHalf];
  for (int n = 0; n < kHalf; ++n) z[n] = {frame[2 * n], frame[2 * n + 1]};
  fft_rec(kHalf, z, 1, Z, tb);
  for (int k = 0; k <= kHalf; ++k) {
    const Cpx a = Z[k % kHalf];
    const Cpx b = {Z[(kHalf - k) % kHalf].re, -Z[(kHalf - k) % kHalf].im};
    const Cpx e = {0.5 * (a.re + b.re), 0.5 * (a.im + b.im)};       // FFT of the even samples
    const Cpx o = {0.5 * (a.im - b.im), -0.5 * (a.re - b.re)};      // FFT of the odd samples: (a - b) / (2i)
    const Cpx w = tb.w400[k];
    const double re = e.re + o.re * w.re - o.im * w.im;
    const double im = e.im + o.re * w.im + o.im * w.re;
    power[k] = re * re + im * im;
  }
}

}  // namespace

extern "C" int gww_logmel_host_f32(const float* wave, int n_seg, int n_samples, long wave_stride, float* out) {
  if (!wave || !out) return gww::fail(GWW_ERR_ARG, "gww_logmel_host_f32: NULL argument");
  if (n_seg < 0 || n_samples < 0) return gww::fail(GWW_ERR_ARG, "gww_logmel_host_f32: negative size");
  const int n_eff = n_samples < kChunk ? n_samples : kChunk;   // HF truncates at 30 s
  if (wave_stride < n_eff)
    return gww::fail(GWW_ERR_ARG, "gww_logmel_host_f32: wave_stride %ld < n_samples %d", wave_stride, n_samples);
  const HostTables& tb = tables();
  int live = kFrames;
  if (n_eff < kChunk - kNfft) {
    live = (n_eff + kNfft / 2 + kHop - 1) / kHop;
    if (live > kFrames) live = kFrames;
    if (live < 1) live = 1;
  }
  std::vector<float> raw((size_t)kNmel * live);
  for (int s = 0; s < n_seg; ++s) {
    const float* x = wave + (long)s * wave_stride;
    float* o = out + (size_t)s * kNmel * kFrames;
    // sample of the zero-padded, reflect-padded buffer at padded index j (buffer index j - 200)
    auto sample = [&](long j) -> double {
      long i = j - kNfft / 2;
      if (i < 0) i = -i;                                  // reflect (no edge repeat)
      if (i >= kChunk) i = 2L * (kChunk - 1) - i;
      return i < n_eff ? (double)x[i] : 0.0;
    };
    float seg_max = -INFINITY;
    for (int t = 0; t < live; ++t) {
      double frame[kNfft], power[kNfreq], mel[kNmel];
      for (int n = 0; n < kNfft; ++n) frame[n] = sample((long)t * kHop + n) * tb.win[n];
      frame_power(frame, power, tb);
      for (int m = 0; m < kNmel; ++m) mel[m] = 0.0;
      for (int k = 0; k < kNfreq; ++k) {
        const float pk = (float)power[k];                 // HF's magnitudes are fp32
        const float* v = tb.fb_val.data() + tb.fb_ofs[k];
        for (int c = 0; c < tb.fb_count[k]; ++c) mel[tb.fb_first[k] + c] += (double)v[c] * (double)pk;
      }
      for (int m = 0; m < kNmel; ++m) {
        float v = (float)mel[m];
        if (v < 1e-10f) v = 1e-10f;
        const float lg = (float)log10((double)v);
        raw[(size_t)m * live + t] = lg;
        if (lg > seg_max) seg_max = lg;
      }
    }
    if (live < kFrames && -10.0f > seg_max) seg_max = -10.0f;   // the dead frames' log10(1e-10) takes part in the max
    const float floor_v = seg_max - 8.0f;
    const float dead = ((-10.0f > floor_v ? -10.0f : floor_v) + 4.0f) / 4.0f;
    for (int m = 0; m < kNmel; ++m) {
      float* row = o + (size_t)m * kFrames;
      const float* r = raw.data() + (size_t)m * live;
      for (int t = 0; t < live; ++t) row[t] = ((r[t] > floor_v ? r[t] : floor_v) + 4.0f) / 4.0f;
      for (int t = live; t < kFrames; ++t) row[t] = dead;
    }
  }
  return GWW_OK;
}
