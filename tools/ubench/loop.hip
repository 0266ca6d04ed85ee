// Micro-benchmark of the A-stationary main loop: W ring (LDS-DMA) + barrier + ds_read_b128 + MFMA.
// Variants switch off one ingredient at a time to see what bounds the loop.
//   hipcc --offload-arch=gfx950 -O3 -o loop loop.hip && ./loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// FLAGS: 1 = issue glds, 2 = barrier+wait, 4 = ds_read W frags (else constants), 8 = MFMA, 16 = setprio
template <int WAVES, int NST, int FLAGS>
__global__ __launch_bounds__(WAVES * 64, 1) void k_loop(const unsigned short* __restrict__ W, float* out, int iters) {
  constexpr int D = NST - 1, KT = 6, GL = 16 / WAVES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * 16384];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  long w_off[GL];
  for (int j = 0; j < GL; ++j) {
    const int row = 8 * (GL * wave + j) + (lane >> 3);
    w_off[j] = (long)row * 384 + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
  }
  auto issue = [&](int it) {
    if (!(FLAGS & 1)) return;
    const int nn = (it / KT) % 9, k0 = (it % KT) * 64;
    unsigned char* sw = lds + (it % NST) * 16384 + (GL * wave) * 1024;
    const unsigned short* wb = W + (long)nn * 128 * 384 + k0;
#pragma unroll
    for (int j = 0; j < GL; ++j) __builtin_amdgcn_global_load_lds((g_ptr)(wb + w_off[j]), (lds_ptr)(sw + j * 1024), 16, 0, 0);
  };
  bf16x8 af[24];
#pragma unroll
  for (int q = 0; q < 24; ++q) af[q] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + q, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  for (int p = 0; p < D; ++p) issue(p);
  for (int it0 = 0; it0 < iters; it0 += KT) {
#pragma unroll
    for (int S = 0; S < KT; ++S) {
      const int it = it0 + S;
      if (FLAGS & 2) {
        if (FLAGS & 1) wait_vm<GL * (D - 1)>();
        __builtin_amdgcn_s_barrier();
      }
      issue(it + D);
      const unsigned char* Ws = lds + (it % NST) * 16384;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (FLAGS & 4) wf[t] = *reinterpret_cast<const bf16x8*>(Ws + swz(32 * t + r, 4 * hh + j));
          else wf[t] = af[(4 * S + j + t) % 24];
        }
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(1);
        if (FLAGS & 8) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], af[4 * S + j], acc[t], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(wf[t]));
        }
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  wait_vm<0>();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
  out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int WAVES, int NST, int FLAGS>
void run(const char* name, const unsigned short* W, float* out, int blocks_per_cu) {
  const int iters = 54 * 6, blocks = 256 * blocks_per_cu;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_loop<WAVES, NST, FLAGS>), dim3(blocks), dim3(WAVES * 64), 0, 0, W, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  }
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double cyc = ms * 1e-3 * 2.4e9 / iters;    // nominal cycles per k-tile iteration per CU round
  const double flop = 2.0 * blocks * WAVES * 32.0 * 128 * 64 * iters;
  printf("%-44s waves %d nst %d  %.3f ms  %7.0f cyc/iter  %6.0f TF\n", name, WAVES, NST, ms, cyc / blocks_per_cu, flop / ms / 1e9);
}

int main() {
  unsigned short* W; float* out;
  CK(hipMalloc(&W, 1152 * 384 * 2)); CK(hipMemset(W, 0x3c, 1152 * 384 * 2));
  CK(hipMalloc(&out, 256 * 8 * 64 * 4 * 4));
  run<8, 7, 1 | 2 | 4 | 8>("full (glds+barrier+ds_read+mfma)", W, out, 1);
  run<8, 7, 1 | 2 | 4 | 8 | 16>("full + setprio", W, out, 1);
  run<8, 7, 2 | 4 | 8>("no glds", W, out, 1);
  run<8, 7, 4 | 8>("no glds, no barrier", W, out, 1);
  run<8, 7, 8>("mfma only (register operands)", W, out, 1);
  run<8, 7, 1 | 2 | 8>("glds+barrier+mfma, no ds_read", W, out, 1);
  run<8, 7, 1 | 2 | 4>("glds+barrier+ds_read, no mfma", W, out, 1);
  run<8, 7, 1 | 2>("glds+barrier only", W, out, 1);
  run<4, 7, 1 | 2 | 4 | 8>("full, 4 waves (1/SIMD), 1 block/CU", W, out, 1);
  run<4, 3, 1 | 2 | 4 | 8>("full, 4 waves, nst 3, 2 blocks/CU", W, out, 2);
  run<4, 7, 8>("mfma only, 4 waves", W, out, 1);
  return 0;
}
