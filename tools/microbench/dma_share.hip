// dma_share.hip — does LOCK-STEP sharing of operand panels between the blocks of one XCD throttle the LDS-DMA stream?
// 256 blocks x 8 waves.  Blocks on one XCD (b % 8) are split into groups of `share`; a group streams the SAME
// 393 KiB panel (256 rows x 768 bf16, 12 K-slices of 32 KiB), either all at the same K-slice (rot=0, what a
// row-major GEMM grid does) or each member starting at a different slice (rot=1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t __attribute__((address_space(3))) * lds_u32p;
typedef const uint32_t __attribute__((address_space(1))) * glb_u32p;

__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int share, int rot, int iters, int fresh) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;       // j = 0..31 within the XCD
  const int grp = j / share, mem = j % share;
  const size_t panel = 256 * 1536;
  const int KT = 12;
  const int r0 = rot ? (mem * KT) / share : 0;
  for (int it = 0; it < iters; ++it) {
    // `fresh`: move to a new panel every KT iterations (first-touch misses, like a GEMM walking down M)
    const size_t pidx = (size_t)(xcd * 32 + grp) + (fresh ? (size_t)(it / KT) * 256 : 0);
    const char* base = src + (pidx % 640) * panel;
    const int kt = (it + r0) % KT;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int piece = wave + 8 * pc;  // 32 pieces = 256 rows x 128 B
      const char* g = base + (size_t)(piece * 8 + (lane >> 3)) * 1536 + kt * 128 + (lane & 7) * 16;
      __builtin_amdgcn_global_load_lds((glb_u32p)g, (lds_u32p)(smem + ((it & 1) * 32 + piece) * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0074);  // previous slice landed, newest 4 pieces in flight
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
}

int main() {
  char* d;
  const size_t total = (size_t)640 * 256 * 1536;
  hipMalloc(&d, total);
  hipMemset(d, 1, total);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int fresh : {0, 1})
    for (int share : {1, 2, 4, 8})
      for (int rot : {0, 1}) {
        if (share == 1 && rot) continue;
        const int iters = 12 * 200;
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, d, share, rot, 24, fresh);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, d, share, rot, iters, fresh);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double gbs = 256.0 * 32768.0 * iters / (ms * 1e-3) / 1e9;
        printf("fresh=%d share=%d rot=%d : %8.1f GB/s  %5.1f B/clk/CU\n", fresh, share, rot, gbs, gbs / 256 / 2.4);
      }
  return 0;
}
