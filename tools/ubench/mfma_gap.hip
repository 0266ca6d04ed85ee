// Microbenchmark: what does one wave per SIMD hide in the gap behind a v_mfma_f32_32x32x16_bf16?
// One workgroup of 4 waves (one per SIMD), a loop of 16 MFMAs per iteration, K filler instructions of a given kind after
// every MFMA, cycles per MFMA from s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_gap mfma_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { F_FMA = 0, F_FMA_DEP = 1, F_EXP = 2, F_ACCREAD = 3, F_DSREAD = 4, F_CVT = 5, F_SALU = 6, F_MIX = 7, F_FMA_VACC = 8, F_DSREAD_USE = 9, F_GELU = 10, F_NOP = 11, F_WAIT = 12, F_GAP = 13, F_GAP_DMA = 14, F_VOR = 15 };

template <int KIND, int K, bool ACC_V>
__global__ __launch_bounds__(256, 1) void k_gap(unsigned long long* out, int iters, float seed, const float* gbuf) {
  __shared__ __attribute__((aligned(16))) float lds[8192 + 4096];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * i;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed - j); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = seed * (threadIdx.x + j);
  float c = seed * 0.5f;
  f32x4 frag[4];
  for (int j = 0; j < 4; ++j) frag[j] = f32x4{seed, seed, seed, seed};
  const unsigned laddr = (threadIdx.x & 63) * 16;
  int sacc = 0;
  const unsigned lane_off = (threadIdx.x & 63) * 16u;
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned dma_dst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + 32768u + wave_u * 4096u;
  const float* dma_src = gbuf + wave_u * 1024;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (ACC_V) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k & 7]) : "v"(c));
        if (KIND == F_FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
        if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k & 7]));
        if (KIND == F_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(v[k & 7]));
        if (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[k & 3]) : "v"(laddr), "n"(1024 * (k & 7)));
        if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(c));
        if (KIND == F_SALU) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
        if (KIND == F_MIX) {   // the fused MLP's mix per gap: 1 read, the rest plain VALU with one transcendental
          if (k == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[u & 3]) : "v"(laddr), "n"(1024 * (u & 7)));
          else if (k == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
          else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(u + k) & 7]) : "v"(c));
        }
        if (KIND == F_DSREAD_USE) {   // a read and, one gap later, a VALU use of it behind a counted wait
          if (k == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[u & 3]) : "v"(laddr), "n"(1024 * (u & 7)));
          else if (k == 1) asm volatile("s_waitcnt lgkmcnt(1)\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(v[u & 7]) : "v"(frag[(u + 3) & 3][0]), "v"(c));
          else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(u + k) & 7]) : "v"(c));
        }
        if (KIND == F_NOP) asm volatile("s_nop 0");
        if (KIND == F_WAIT) asm volatile("s_waitcnt lgkmcnt(7)");
        if (KIND == F_VOR) asm volatile("v_or_b32 %0, %1, %2" : "=v"(v[k & 7]) : "s"(sacc), "v"(laddr));
        if (KIND == F_GAP || KIND == F_GAP_DMA) {   // the planned gap: counted wait, one fragment read, K - 2 independent VALU (an exp in 1 gap of 3)
          if (k == 0) asm volatile("s_waitcnt lgkmcnt(3)");
          else if (k == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[u & 3]) : "v"(laddr), "n"(1024 * (u & 7)));
          else if (k == 2 && (u % 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
          else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(u + k) & 7]) : "v"(c));
          if (KIND == F_GAP_DMA && k == K - 1 && (u & 3) == 3)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(lane_off), "s"(dma_src), "s"(dma_dst), "n"(1024 * ((u >> 2) & 3)) : "memory");
        }
        if (KIND == F_GELU) {   // K steps of a dependent GELU-like chain on TWO values a b a b (the kernel's shape)
          const int st = (2 * u + k / 2) % 10, w = k & 1;
          if (st == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v[w]));
          else if (st == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[w]));
          else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[w]) : "v"(c));
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int j = 0; j < 16; ++j) s += acc[t][j];
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int j = 0; j < 4; ++j) s += frag[j][0] + frag[j][3];
  if (s == 12345.678f) out[2] = sacc;   // keep everything live
  if (threadIdx.x == 0) out[0] = t1 - t0;
}


// The fused MLP's riding phase, gap by gap (48 gaps = 4 groups of 12; pair A of a group runs operation j in gap j, pair B
// operation j - 1): MFMA | counted wait | fragment read | the slice | (every 4th gap) one LDS-DMA piece.
// VARIANT bits: 1 no DMA, 2 transcendentals replaced by v_mul, 4 no s_nop in front of the MFMA, 8 no fragment read / wait,
// 16 literal-constant fma replaced by register fma, 32 accumulators alternate between two registers only (fc1 shape)
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k_gelu(unsigned long long* out, int iters, float seed, const float* gbuf) {
  __shared__ __attribute__((aligned(16))) float lds[8192 + 4096];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * i;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed - j); }
  f32x4 frag8[8];
  for (int j = 0; j < 8; ++j) frag8[j] = f32x4{seed, seed, seed, seed};
  float S[4], W[4], Q[4], T[4];
  for (int j = 0; j < 4; ++j) { S[j] = seed * (threadIdx.x + j); W[j] = Q[j] = T[j] = 0.f; }
  float c = seed * 0.5f;
  unsigned pk = 0;
  f32x4 frag[4];
  for (int j = 0; j < 4; ++j) frag[j] = f32x4{seed, seed, seed, seed};
  const unsigned laddr = (threadIdx.x & 63) * 16;
  const unsigned lane_off = (threadIdx.x & 63) * 16u;
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned dma_dst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + 32768u + wave_u * 4096u;
  const float* dma_src = gbuf + wave_u * 1024;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 48; ++g) {
      const int u = g & 3;
      if (!(VARIANT & 4)) asm volatile("s_nop 0");
      if (VARIANT & 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
      if (!(VARIANT & 8)) {
        if (VARIANT & 64) asm volatile("s_waitcnt lgkmcnt(5)");      // fragment reads six gaps ahead of their use (8 buffers)
        else if (VARIANT & 128) asm volatile("s_waitcnt lgkmcnt(3)"); // four gaps ahead
        else asm volatile("s_waitcnt lgkmcnt(2)");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag8[g & 7]) : "v"(laddr), "n"(2048));
      }
      const int j = g % 12;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int op = j - pr;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int i = 2 * pr + sl;
          if (op == 0) asm volatile("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(W[i]) : "v"(S[i]));
          if (op == 1) {
            if (VARIANT & 16) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(Q[i]) : "v"(W[i]), "v"(c));
            else asm volatile("v_fmamk_f32 %0, %1, 0x42050396, %2" : "=v"(Q[i]) : "v"(W[i]), "v"(c));
          }
          if (op == 2) {
            if (VARIANT & 16) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(Q[i]) : "v"(W[i]), "v"(c));
            else asm volatile("v_fmaak_f32 %0, %1, %0, 0xc1934584" : "+v"(Q[i]) : "v"(W[i]));
          }
          if (op == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(W[i]) : "v"(S[i]), "v"(Q[i]));
          if (op == 4) {
            if (VARIANT & 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(W[i]) : "v"(c));
            else asm volatile("v_exp_f32 %0, %0" : "+v"(W[i]));
          }
          if (op == 5) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(W[i]));
          if (op == 6) {
            if (VARIANT & 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(W[i]) : "v"(c));
            else asm volatile("v_rcp_f32 %0, %0" : "+v"(W[i]));
          }
          if (op == 7) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(T[i]) : "v"(S[i]), "v"(W[i]));
          if (op == 8 && sl == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(T[2 * pr]), "v"(T[2 * pr + 1]));
        }
      }
      if (!(VARIANT & 1) && u == 0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(lane_off), "s"(dma_src), "s"(dma_dst), "n"(1024) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int j = 0; j < 16; ++j) s += acc[t][j];
  for (int j = 0; j < 4; ++j) s += T[j] + W[j] + Q[j] + frag[j][0] + frag[j][3];
  for (int j = 0; j < 8; ++j) s += frag8[j][0] + frag8[j][3];
  if (s == 12345.678f) out[2] = pk;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int VARIANT>
static void run_gelu(const char* name, unsigned long long* d) {
  const int iters = 700;
  hipLaunchKernelGGL((k_gelu<VARIANT>), dim3(1), dim3(256), 0, 0, d, iters, 1e-3f, (const float*)(d + 8));
  hipLaunchKernelGGL((k_gelu<VARIANT>), dim3(1), dim3(256), 0, 0, d, iters, 1e-3f, (const float*)(d + 8));
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("riding phase, %-52s: %6.1f cycles per MFMA\n", name, (double)h / (48.0 * iters));
}

template <int KIND, int K, bool ACC_V>
static void run(const char* name, unsigned long long* d) {
  const int iters = 2000;
  hipLaunchKernelGGL((k_gap<KIND, K, ACC_V>), dim3(1), dim3(256), 0, 0, d, iters, 1e-3f, (const float*)(d + 8));
  hipLaunchKernelGGL((k_gap<KIND, K, ACC_V>), dim3(1), dim3(256), 0, 0, d, iters, 1e-3f, (const float*)(d + 8));
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-28s K=%d acc in %s: %6.1f cycles per MFMA\n", name, K, ACC_V ? "VGPR" : "AGPR", (double)h / (16.0 * iters));
}

#define SWEEP(KIND, NAME, V)                                                                     \
  run<KIND, 0, V>(NAME, d); run<KIND, 2, V>(NAME, d); run<KIND, 4, V>(NAME, d); run<KIND, 5, V>(NAME, d); \
  run<KIND, 6, V>(NAME, d); run<KIND, 8, V>(NAME, d);

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 + 65536);
  hipMemset(d, 0, 64 + 65536);
  SWEEP(F_FMA, "independent v_fma", false)
  SWEEP(F_FMA, "independent v_fma", true)
  SWEEP(F_FMA_DEP, "dependent v_fma chain", false)
  SWEEP(F_EXP, "v_exp_f32", false)
  SWEEP(F_ACCREAD, "v_accvgpr_read", false)
  SWEEP(F_DSREAD, "ds_read_b128", false)
  SWEEP(F_CVT, "v_cvt_pk_bf16_f32", false)
  SWEEP(F_SALU, "s_add_i32", false)
  SWEEP(F_MIX, "1 ds_read + 1 exp + fma", false)
  SWEEP(F_DSREAD_USE, "ds_read, used a gap later", false)
  SWEEP(F_GELU, "GELU-like chain a b a b", false)
  SWEEP(F_NOP, "s_nop 0", false)
  SWEEP(F_WAIT, "s_waitcnt (nothing pending)", false)
  SWEEP(F_VOR, "v_or_b32 v, s, v", false)
  SWEEP(F_GAP, "wait + read + (K-2) VALU", false)
  SWEEP(F_GAP_DMA, "same + DMA piece per 4 gaps", false)
  run_gelu<0>("as in the kernel", d);
  run_gelu<1>("no DMA", d);
  run_gelu<2>("transcendentals -> v_mul", d);
  run_gelu<3>("no DMA, transcendentals -> v_mul", d);
  run_gelu<4>("no s_nop", d);
  run_gelu<16>("literal fma -> register fma", d);
  run_gelu<32>("two alternating VGPR accumulators (fc1 shape)", d);
  run_gelu<128>("fragment wait lgkmcnt(3)", d);
  run_gelu<64>("fragment wait lgkmcnt(5)", d);
  run_gelu<64 | 4>("fragment wait lgkmcnt(5), no s_nop", d);
  run_gelu<8>("no fragment read / wait", d);
  run_gelu<9>("no fragment read / wait, no DMA", d);
  run_gelu<1 | 2 | 4 | 16>("no DMA, no trans, no nop, no literals", d);
  hipFree(d);
  return 0;
}
