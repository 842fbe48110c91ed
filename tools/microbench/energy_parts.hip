// energy_parts.hip — one data-movement / arithmetic resource of the GEMM step at a time, looped for a few seconds, so that
// tools/power_clocks.py --cmd can put a socket power and a clock next to its RATE: marginal energy per unit = (P - P_idle) / rate.
//     hipcc -O3 --offload-arch=gfx950 -o /tmp/energy_parts tools/microbench/energy_parts.hip && /tmp/energy_parts <mode> <seconds>
// modes (256 CUs x 8 waves each unless noted):
//   lds_read   every wave streams ds_read_b128 (conflict-free, 1 KiB per instruction) out of a 64 KiB LDS region            -> LDS GB/s
//   l2_dma     global_load_lds_dwordx4 (the GEMM's LDS-DMA) of a 1 MiB region shared by the CUs of an XCD (L2 hits)      -> L2->LDS GB/s
//   hbm_read   every wave streams 16-byte loads through its own slice of an 8 GiB buffer (>> the 256 MiB MALL)           -> HBM read GB/s
//   hbm_copy   the same with a 16-byte store of every value to a second buffer                                           -> HBM read + write GB/s
//   valu       v_pk_fma_f32 stream, 8 independent chains per lane                                                        -> GFLOP/s (fp32)
//   mfma       v_mfma_f32_32x32x16_bf16 on N(0,1) bf16 operands held in registers, a GEMM tile's operand rotation        -> TFLOP/s
//   mfma_lds   the same MFMA stream with its operands RE-READ from LDS every k-step at the persistent GEMM's ratio
//              (6 ds_read_b128 per 8 MFMAs, N(0,1) data in LDS)                                                          -> TFLOP/s + LDS GB/s
//   mfma_lds_dma   mfma_lds + one 64 KiB K-tile of LDS-DMA (L2 hits) per 4 k-steps: the GEMM's K loop without an epilogue -> TFLOP/s + GB/s
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline float gauss(uint32_t& s) {
  float v = 0.f;
  for (int i = 0; i < 4; ++i) { s = mix32(s + 0x9e3779b9u); v += (float)(s >> 8) * (1.0f / 16777216.0f); }
  return (v - 2.0f) * 1.7320508f;
}

__device__ inline void fill_lds_gauss(char* smem, int bytes, uint32_t seed) {
  __bf16* p = reinterpret_cast<__bf16*>(smem);
  for (int i = threadIdx.x; i < bytes / 2; i += blockDim.x) { uint32_t s = seed + i * 2654435761u; p[i] = (__bf16)gauss(s); }
  __syncthreads();
}

__global__ __launch_bounds__(512) void k_lds_read(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fill_lds_gauss(smem, 65536, blockIdx.x * 977u);
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
  u32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  for (int it = 0; it < iters; ++it) {
    u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                 "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(base) : "memory");
    a0 ^= v0 ^ v4; a1 ^= v1 ^ v5; a2 ^= v2 ^ v6; a3 ^= v3 ^ v7;
  }
  const u32x4 r = a0 ^ a1 ^ a2 ^ a3;
  if ((r[0] ^ r[1] ^ r[2] ^ r[3]) == 0x12345677u) sink[0] = 1.f;
}

__device__ __forceinline__ void dma_piece(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(512) void k_l2_dma(int iters, const char* src, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* base = src + (size_t)(blockIdx.x & 7) * (1u << 20);  // one 1 MiB region per XCD: every CU of the XCD reads the same bytes
  for (int it = 0; it < iters; ++it) {
    const uint32_t off = ((uint32_t)(it & 15) * 65536u + wave * 8192u + lane * 16u) & ((1u << 20) - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(base, off + j * 1024u, lds0 + wave * 8192u + j * 1024u);  // 64 KiB per workgroup and trip = one K-tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (smem[threadIdx.x] == 77 && iters < 0) sink[0] = 1.f;
}

__global__ __launch_bounds__(512) void k_hbm(int iters, const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, float* sink, int copy) {
  const size_t per_wg = n16 / gridDim.x;
  const u32x4* s = src + (size_t)blockIdx.x * per_wg;
  u32x4* d = dst + (size_t)blockIdx.x * per_wg;
  u32x4 acc = {0, 0, 0, 0};
  size_t i = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const size_t k = (i + (size_t)j * 512) & (per_wg - 1); v[j] = __builtin_nontemporal_load(s + k); }  // (per_wg is a power of two)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (copy) { const size_t k = (i + (size_t)j * 512) & (per_wg - 1); __builtin_nontemporal_store(v[j], d + k); }
      else acc ^= v[j];
    }
    i += 2048;
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345677u) sink[0] = 1.f;
}

__global__ __launch_bounds__(512) void k_valu(int iters, float* sink) {
  f32x2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f32x2{0.001f * (threadIdx.x + i), 0.002f * i};
  const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
  if (s == 1.2345e33f) sink[0] = s;
}

// MODE 0: registers only; 1: operands re-read from LDS per k-step (6 reads per 8 MFMAs); 2: + one K-tile of LDS-DMA per 4 k-steps
template <int MODE>
__global__ __launch_bounds__(512) void k_mfma(int iters, const char* src, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  if (MODE >= 1) fill_lds_gauss(smem, 131072, blockIdx.x * 977u);
  bf16x8 a[2][4], b[2][2];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) a[s][i][j] = (__bf16)gauss(seed);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 8; ++j) b[s][i][j] = (__bf16)(0.05f * gauss(seed));
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  typedef __attribute__((address_space(3))) const bf16x8* lds_frag_p;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t rd = lds0 + lane * 16 + (wave & 1) * 16384;  // the 8 waves share fragments the way 2 x 4 wave tiles do (conflict-free linear reads)
  const char* base = src + (size_t)(blockIdx.x & 7) * (1u << 20);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (MODE >= 1) {
        const uint32_t o = rd + ((it * 2 + s) & 7) * 6144u;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[s][i] = *reinterpret_cast<lds_frag_p>((uintptr_t)(o + i * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i) b[s][i] = *reinterpret_cast<lds_frag_p>((uintptr_t)(o + 4096 + i * 1024));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][mi], b[s][ni], acc[mi * 2 + ni], 0, 0, 0);
    }
    if (MODE == 2 && (it & 1) == 1) {  // every 4 k-steps: one 64 KiB K-tile by LDS-DMA (8 pieces per wave) into the other half of the LDS
      const uint32_t off = ((uint32_t)((it >> 1) & 15) * 65536u + wave * 8192u + lane * 16u) & ((1u << 20) - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) dma_piece(base, off + j * 1024u, lds0 + 65536u + wave * 8192u + j * 1024u);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
  if (s == 1.2345e33f) sink[0] = s;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "mfma";
  const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* sink;
  CHECK(hipMalloc(&sink, 4));
  char* l2buf = nullptr;
  CHECK(hipMalloc(&l2buf, 8u << 20));
  CHECK(hipMemset(l2buf, 1, 8u << 20));
  u32x4 *hsrc = nullptr, *hdst = nullptr;
  const size_t hbm_bytes = (size_t)8 << 30, n16 = hbm_bytes / 16;
  const bool hbm = !strncmp(mode, "hbm", 3);
  if (hbm) {
    CHECK(hipMalloc(&hsrc, hbm_bytes));
    CHECK(hipMemset(hsrc, 3, hbm_bytes));
    if (!strcmp(mode, "hbm_copy")) CHECK(hipMalloc(&hdst, hbm_bytes));
  }
  double unit_per_launch = 0, unit2_per_launch = 0;
  const char *unit = "", *unit2 = nullptr;
  int iters = 20000;
  auto launch = [&]() {
    if (!strcmp(mode, "lds_read")) {
      CHECK(hipFuncSetAttribute((const void*)k_lds_read, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      hipLaunchKernelGGL(k_lds_read, dim3(cus), dim3(512), 65536, 0, iters, sink);
      unit_per_launch = (double)cus * 8 * iters * 8 * 1024; unit = "LDS read GB/s";
    } else if (!strcmp(mode, "l2_dma")) {
      CHECK(hipFuncSetAttribute((const void*)k_l2_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      hipLaunchKernelGGL(k_l2_dma, dim3(cus), dim3(512), 65536, 0, iters, l2buf, sink);
      unit_per_launch = (double)cus * iters * 65536.0; unit = "L2->LDS DMA GB/s";
    } else if (hbm) {
      iters = 4000;
      const int copy = hdst != nullptr;
      hipLaunchKernelGGL(k_hbm, dim3(cus * 4), dim3(512), 0, 0, iters, hsrc, hdst, n16, sink, copy);
      unit_per_launch = (double)cus * 4 * 512 * iters * 4 * 16 * (copy ? 2 : 1); unit = copy ? "HBM read+write GB/s" : "HBM read GB/s";
    } else if (!strcmp(mode, "valu")) {
      hipLaunchKernelGGL(k_valu, dim3(cus), dim3(512), 0, 0, iters, sink);
      unit_per_launch = (double)cus * 512 * iters * 64.0 * 4; unit = "fp32 GFLOP/s (pk_fma)";
    } else {
      const int m = !strcmp(mode, "mfma") ? 0 : !strcmp(mode, "mfma_lds") ? 1 : 2;
      const double fl = (double)cus * 8 * iters * 16.0 * (2.0 * 32 * 32 * 16);
      unit_per_launch = fl; unit = "MFMA GFLOP/s";
      if (m == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(cus), dim3(512), 0, 0, iters, l2buf, sink);
      else if (m == 1) {
        CHECK(hipFuncSetAttribute((const void*)k_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        hipLaunchKernelGGL(k_mfma<1>, dim3(cus), dim3(512), 131072, 0, iters, l2buf, sink);
        unit2_per_launch = (double)cus * 8 * iters * 12.0 * 1024; unit2 = "LDS read GB/s";
      } else {
        CHECK(hipFuncSetAttribute((const void*)k_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        hipLaunchKernelGGL(k_mfma<2>, dim3(cus), dim3(512), 131072, 0, iters, l2buf, sink);
        unit2_per_launch = (double)cus * 8 * iters * 12.0 * 1024 + (double)cus * (iters / 2) * 65536.0; unit2 = "LDS read + DMA write GB/s";
      }
    }
  };
  launch();
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  float ms_total = 0.f, ms_last = 0.f;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CHECK(hipEventRecord(e0, 0));
    for (int k = 0; k < 4; ++k) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_last, e0, e1));
    ms_total += ms_last;
    n += 4;
  }
  printf("ENERGY_PART {\"mode\": \"%s\", \"launches\": %d, \"rate\": %.2f, \"unit\": \"%s\", \"rate_last4\": %.2f", mode, n,
         unit_per_launch * n / (ms_total * 1e-3) / 1e9, unit, unit_per_launch * 4 / (ms_last * 1e-3) / 1e9);
  if (unit2) printf(", \"rate2\": %.2f, \"unit2\": \"%s\"", unit2_per_launch * n / (ms_total * 1e-3) / 1e9, unit2);
  printf("}\n");
  return 0;
}
