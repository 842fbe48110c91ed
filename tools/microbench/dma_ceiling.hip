// dma_ceiling.hip — per-CU ceiling of the global->LDS DMA path (global_load_lds_dwordx4) and of plain
// global_load_dwordx4, from an L2-resident working set, as a function of waves per CU and pieces in flight.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_ceiling tools/microbench/dma_ceiling.hip && /tmp/dma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t __attribute__((address_space(3))) * lds_u32p;
typedef const uint32_t __attribute__((address_space(1))) * glb_u32p;

// MODE 0: LDS-DMA, rows of 128 B gathered like a GEMM operand tile (8 rows x 128 B per piece, row stride `ld` bytes)
// MODE 1: LDS-DMA, fully linear 1 KiB pieces
// MODE 2: plain global_load_dwordx4 into VGPRs (linear)
template <int MODE, int PIECES>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t region_bytes, int ld, int iters,
                                                  float* sink, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const char* base = src + (size_t)(blockIdx.x / share) * region_bytes;
  float accum = 0.f;
  for (int it = 0; it < iters; ++it) {
    const size_t koff = (size_t)(it & 7) * 128;  // walk 8 K-tiles of 128 B then wrap (stays in L2)
#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) {
      const int piece = wave + nw * pc;
      if (MODE == 0) {
        const char* g = base + (size_t)(piece * 8 + (lane >> 3)) * ld + koff + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds((glb_u32p)g, (lds_u32p)(smem + piece * 1024), 16, 0, 0);
      } else if (MODE == 3) {
        const char* g = base + (size_t)(piece * 16 + (lane >> 2)) * ld + (size_t)(it & 15) * 64 + (lane & 3) * 16;
        __builtin_amdgcn_global_load_lds((glb_u32p)g, (lds_u32p)(smem + piece * 1024), 16, 0, 0);
      } else if (MODE == 1) {
        const char* g = base + (size_t)piece * 1024 + (size_t)(it & 7) * (size_t)(nw * PIECES) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((glb_u32p)g, (lds_u32p)(smem + piece * 1024), 16, 0, 0);
      } else {
        const char* g = base + (size_t)piece * 1024 + (size_t)(it & 7) * (size_t)(nw * PIECES) * 1024 + lane * 16;
        float4 v = *reinterpret_cast<const float4*>(g);
        accum += v.x + v.y + v.z + v.w;
      }
    }
    if (MODE != 2) __builtin_amdgcn_s_waitcnt(0x0070 | (PIECES & 15));  // keep the newest PIECES pieces in flight
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  if (accum == 1.2345e33f) sink[0] = accum;
}

template <int MODE, int PIECES>
static void run(const char* name, int waves, const char* d_src, float* d_sink, int ld, int share = 1) {
  const int blocks = 256;
  const int iters = 4096;
  const size_t region = 1 << 20;  // 1 MiB per block: 32 blocks per XCD -> but only ~64 KiB x 8 K-tiles touched per block
  const size_t smem = (size_t)waves * PIECES * 1024;
  auto k = dma_kernel<MODE, PIECES>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), smem, 0, d_src, region, ld, 64, d_sink, share);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), smem, 0, d_src, region, ld, iters, d_sink, share);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * waves * PIECES * 1024.0 * iters;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-28s share=%d waves/CU=%d pieces=%d : %8.1f GB/s total  %6.1f GB/s/CU  %5.1f B/clk/CU @2.4GHz\n", name, share, waves, PIECES, gbs,
         gbs / 256, gbs / 256 / 2.4);
}

int main() {
  char* d_src;
  float* d_sink;
  const size_t total = (size_t)256 << 20;
  hipMalloc(&d_src, total);
  hipMemset(d_src, 1, total);
  hipMalloc(&d_sink, 16);
  // 8 waves x 4 pieces = 32 KiB per iteration per block (one BK=32 stage of the 256x256 GEMM tile)
  for (int share : {1, 4, 8, 32, 256}) {
    run<0, 4>("gather 8x128B ld=1536", 8, d_src, d_sink, 1536, share);
    run<3, 4>("gather 16x64B ld=1536", 8, d_src, d_sink, 1536, share);
    run<3, 4>("gather 16x64B ld=1600", 8, d_src, d_sink, 1600, share);
  }
  run<0, 2>("gather 8x128B ld=1536", 8, d_src, d_sink, 1536, 1);
  run<3, 2>("gather 16x64B ld=1536", 8, d_src, d_sink, 1536, 1);
  run<1, 4>("linear", 8, d_src, d_sink, 0, 1);
  run<1, 4>("linear", 8, d_src, d_sink, 0, 8);
  return 0;
}
