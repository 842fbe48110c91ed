// mfma_dma_mix.hip — how much do an MFMA wave and an LDS-DMA wave on the SAME SIMD slow each other down, and does the
// addressing form of the DMA matter?   8 waves per block (1 block per CU): waves 0-3 = MFMA loop, waves 4-7 = DMA loop
// (wave w and w+4 share a SIMD).  MODE: 0 = global_load_lds 64-bit vaddr, 1 = global_load_lds saddr + 32-bit voffset,
// 2 = raw.buffer.load.lds (SRD + 32-bit voffset).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t __attribute__((address_space(3))) * lds_u32p;
typedef const uint32_t __attribute__((address_space(1))) * glb_u32p;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int int32x4;

__device__ void llvm_amdgcn_raw_buffer_load_lds(int32x4 rsrc, lds_u32p lds_ptr, int size, int voffset, int soffset, int offset,
                                                int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

template <int MODE>
__global__ __launch_bounds__(512) void mix(const char* __restrict__ src, int iters, int do_mfma, int do_dma, float* sink,
                                           unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (!do_mfma) return;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * lane); b[j] = (__bf16)(0.002f * j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 1.2345e33f) sink[0] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t0;
  } else {
    if (!do_dma) return;
    const int w = wave - 4;
    // per block a 64 KiB region (L2-resident across the chip: 16 MiB total), 8 rows x 128 B per piece
    const char* base = src + (size_t)blockIdx.x * 65536;
    const uint32_t voff = (uint32_t)((lane >> 3) * 128 + (lane & 7) * 16);
    int32x4 rsrc;
    {
      const uint64_t p = (uint64_t)base;
      rsrc[0] = (int)(p & 0xffffffffu); rsrc[1] = (int)((p >> 32) & 0xffffu); rsrc[2] = 65536; rsrc[3] = 0x00020000;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {  // 8 pieces per iteration per wave = 32 KiB per iteration per block
        const uint32_t off = (uint32_t)(((it & 1) * 32 + pc * 4 + w) * 1024);
        char* dst = smem + (pc * 4 + w) * 1024;
        if (MODE == 0) {
          __builtin_amdgcn_global_load_lds((glb_u32p)(base + off + voff), (lds_u32p)dst, 16, 0, 0);
        } else if (MODE == 1) {
          const char* sb = base + off;  // wave-uniform
          uint32_t keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(sb), "s"((uint32_t)(uintptr_t)(lds_u32p)dst) : "memory");
        } else {
          llvm_amdgcn_raw_buffer_load_lds(rsrc, (lds_u32p)dst, 16, (int)(off + voff), 0, 0, 0);
        }
      }
      if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else __builtin_amdgcn_s_waitcnt(0x0078);
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else __builtin_amdgcn_s_waitcnt(0x0070);
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memtime() - t0;
  }
}

template <int MODE>
void run(const char* name, const char* d_src, float* d_sink, unsigned long long* d_cyc, int do_mfma, int do_dma) {
  const int iters = 2000;
  auto k = mix<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipMemset(d_cyc, 0, 256 * 8 * 8);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, d_src, 50, do_mfma, do_dma, d_sink, d_cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, d_src, iters, do_mfma, do_dma, d_sink, d_cyc);
  hipDeviceSynchronize();
  unsigned long long h[256 * 8];
  hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double cm = 0, cd = 0; int nm = 0, nd = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) { if (h[b*8+w] == 0) continue; if (w < 4) { cm += h[b*8+w]; ++nm; } else { cd += h[b*8+w]; ++nd; } }
  printf("%-34s mfma=%d dma=%d :", name, do_mfma, do_dma);
  if (nm) printf("  MFMA %.1f cyc/instr", cm / nm / (iters * 32.0));
  if (nd) printf("  DMA %.1f cyc/piece/wave -> %.1f B/clk/CU", cd / nd / (iters * 8.0), 4.0 * 1024.0 / (cd / nd / (iters * 8.0)));
  printf("\n");
}

int main() {
  char* d_src; float* d_sink; unsigned long long* d_cyc;
  hipMalloc(&d_src, (size_t)256 * 65536 * 2); hipMemset(d_src, 1, (size_t)256 * 65536 * 2);
  hipMalloc(&d_sink, 16); hipMalloc(&d_cyc, 256 * 8 * 8);
  run<0>("global_load_lds vaddr64", d_src, d_sink, d_cyc, 1, 0);
  run<0>("global_load_lds vaddr64", d_src, d_sink, d_cyc, 0, 1);
  run<0>("global_load_lds vaddr64", d_src, d_sink, d_cyc, 1, 1);
  run<1>("global_load_lds saddr+voff", d_src, d_sink, d_cyc, 0, 1);
  run<1>("global_load_lds saddr+voff", d_src, d_sink, d_cyc, 1, 1);
  run<2>("buffer_load lds (SRD)", d_src, d_sink, d_cyc, 0, 1);
  run<2>("buffer_load lds (SRD)", d_src, d_sink, d_cyc, 1, 1);
  return 0;
}
