// Trigger clustering of the search pipeline on the device.
//
// Reference: MLGWSC-1/inference.py:140-166 (`get_clusters`) fed by :484-487 (`evaluate_slices` keeps the windows whose
// score exceeds `trigger_threshold`, one `.item()` per window): time-ordered triggers closer than `cluster_threshold`
// seconds to their predecessor join its cluster; a cluster is reported as the time and value of its FIRST maximum.
// Here the per-window scores never leave the GPU before they are clustered: one wave walks the score array 64 windows at a
// time, the trigger mask of a step is a ballot, and the (rare) set bits are folded into a wave-uniform running cluster --
// a sequential algorithm whose state is four scalars, so one wave is the natural shape; n / 64 steps of a few
// instructions (120 000 windows: about 2 000 steps).  Times are the reference's float64 stamps, compared in float64.
#include "common.h"

namespace gww {
namespace {

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(64) void k_cluster_triggers(const double* __restrict__ times, const float* __restrict__ scores,
                                                         long n, float trigger_threshold, double cluster_threshold,
                                                         double* __restrict__ out_t, float* __restrict__ out_v,
                                                         int* __restrict__ out_count, int max_clusters) {
  const int lane = threadIdx.x;
  bool have = false;
  double last_t = 0.0, best_t = 0.0;
  float best_v = 0.f;
  int count = 0;
  auto emit = [&]() {
    if (lane == 0 && count < max_clusters) { out_t[count] = best_t; out_v[count] = best_v; }
    ++count;
  };
  for (long base = 0; base < n; base += 64) {
    const long i = base + lane;
    const float s = i < n ? scores[i] : -INFINITY;
    const double t = i < n ? times[i] : 0.0;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(s > trigger_threshold);
    while (mask) {
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      const double tl = readlane_f64(t, l);
      const float vl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), l));
      if (have && (tl - last_t) > cluster_threshold) {
        emit();
        have = false;
      }
      if (!have) { best_t = tl; best_v = vl; have = true; }
      else if (vl > best_v) { best_t = tl; best_v = vl; }     // strictly greater: np.argmax keeps the first maximum
      last_t = tl;
    }
  }
  if (have) emit();
  if (lane == 0) *out_count = count;
}

}  // namespace
}  // namespace gww

using namespace gww;

extern "C" int gww_cluster_triggers_f64(const double* times, const float* scores, long n, float trigger_threshold,
                                        double cluster_threshold, double* out_times, float* out_vals, int* out_count,
                                        int max_clusters, void* stream) {
  GWW_REQUIRE(out_count && (n == 0 || (times && scores)) && (max_clusters == 0 || (out_times && out_vals)),
              "gww_cluster_triggers_f64: NULL argument");
  GWW_REQUIRE(n >= 0 && max_clusters >= 0, "gww_cluster_triggers_f64: negative size");
  hipLaunchKernelGGL(k_cluster_triggers, dim3(1), dim3(64), 0, (hipStream_t)stream, times, scores, n, trigger_threshold,
                     cluster_threshold, out_times, out_vals, out_count, max_clusters);
  GWW_LAUNCH_CHECK();
  return GWW_OK;
}
