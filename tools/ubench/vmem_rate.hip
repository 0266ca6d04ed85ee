// Microbenchmark: what one CU (8 waves, 1 workgroup per CU) sustains in 1-KiB VMEM wave-instructions -- LDS-DMA loads
// (global_load_lds_dwordx4) and 16-byte-per-lane stores -- for the access shapes of the GEMM kernels.
//   mode 0: LDS-DMA, 8 rows x 128 B per piece, source re-read from a 64-KB window per block (L2-resident)
//   mode 1: LDS-DMA, 8 rows x 128 B, streaming a large buffer (HBM)
//   mode 2: LDS-DMA, 16 rows x 64 B per piece (row stride 1536 B), L2-resident window
//   mode 3: stores, 16 rows x 64 B per instruction (row stride 4608 B), streaming
//   mode 4: stores, 8 rows x 128 B per instruction, streaming
//   mode 5: stores, 16 rows x 64 B, into a 64-KB window per block (L2-resident)
//   mode 6: stores, 4 rows x 256 B per instruction (fp32 rows), streaming
// Build: hipcc --offload-arch=gfx950 -O3 -o vmem_rate vmem_rate.hip ; run: ./vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* g_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(512, 1) void k_rate(const unsigned char* src, unsigned char* dst, int iters, long bytes_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned char* sb = src + (long)blockIdx.x * bytes_per_block;
  unsigned char* db = dst + (long)blockIdx.x * bytes_per_block;
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long piece = ((long)it * 8 + u) * 8 + wave;   // 1 KiB pieces of this block, 8 waves side by side
      if constexpr (MODE == 0 || MODE == 1) {
        const long off = MODE == 0 ? (piece & 63) * 1024 : piece * 1024;
        __builtin_amdgcn_global_load_lds((g_ptr)(sb + off + lane * 16), (lds_ptr)(lds + ((piece & 127) * 1024)), 16, 0, 0);
      } else if constexpr (MODE == 2) {
        const long off = ((piece & 31) * 16 + (lane >> 2)) * 1536 + (lane & 3) * 16;
        __builtin_amdgcn_global_load_lds((g_ptr)(sb + off), (lds_ptr)(lds + ((piece & 127) * 1024)), 16, 0, 0);
      } else if constexpr (MODE == 3) {
        const long off = (piece * 16 + (lane >> 2)) * 4608 % bytes_per_block + (lane & 3) * 16;
        *reinterpret_cast<u32x4*>(db + off) = v;
      } else if constexpr (MODE == 4) {
        *reinterpret_cast<u32x4*>(db + piece * 1024 + lane * 16) = v;
      } else if constexpr (MODE == 5) {
        const long off = ((piece & 3) * 16 + (lane >> 2)) * 1024 + (lane & 3) * 16;
        *reinterpret_cast<u32x4*>(db + off) = v;
      } else {
        const long off = (piece * 4 + (lane >> 4)) * 3072 % bytes_per_block + (lane & 15) * 16;
        *reinterpret_cast<u32x4*>(db + off) = v;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 123 && iters < 0) dst[0] = 1;
}

template <int MODE, int INFLIGHT>
void run(const char* name, unsigned char* src, unsigned char* dst, long bpb) {
  const int iters = (int)(bpb / 65536);   // 64 KiB per block and iteration
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_rate<MODE, INFLIGHT><<<256, 512>>>(src, dst, iters, bpb);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k_rate<MODE, INFLIGHT><<<256, 512>>>(src, dst, iters, bpb);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = 256.0 * iters * 65536;
  printf("%-58s inflight %2d: %7.3f ms  %6.2f TB/s chip  %5.1f ns per 1-KiB instruction and CU\n", name, INFLIGHT, ms,
         bytes / ms / 1e9, ms * 1e6 / (iters * 64.0));
}

int main() {
  const long bpb = 16L << 20;   // 16 MiB per block, 4 GiB total
  unsigned char *src, *dst;
  hipMalloc(&src, 256 * bpb); hipMalloc(&dst, 256 * bpb);
  hipMemset(src, 1, 256 * bpb); hipMemset(dst, 0, 256 * bpb);
  run<0, 8>("LDS-DMA 8 x 128 B, L2 window", src, dst, bpb);
  run<0, 16>("LDS-DMA 8 x 128 B, L2 window", src, dst, bpb);
  run<1, 8>("LDS-DMA 8 x 128 B, streaming", src, dst, bpb);
  run<1, 16>("LDS-DMA 8 x 128 B, streaming", src, dst, bpb);
  run<1, 32>("LDS-DMA 8 x 128 B, streaming", src, dst, bpb);
  run<2, 8>("LDS-DMA 16 x 64 B, L2 window", src, dst, bpb);
  run<3, 16>("stores 16 x 64 B (stride 4608), streaming", src, dst, bpb);
  run<4, 16>("stores 8 x 128 B, streaming", src, dst, bpb);
  run<6, 16>("stores 4 x 256 B (stride 3072), streaming", src, dst, bpb);
  run<5, 16>("stores 16 x 64 B, L2 window", src, dst, bpb);
  run<4, 4>("stores 8 x 128 B, streaming", src, dst, bpb);
  return 0;
}
