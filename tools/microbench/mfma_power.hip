// mfma_power.hip — what the chip sustains on v_mfma_f32_32x32x16_bf16 from REGISTERS ONLY (no LDS, no global traffic in the loop) as a
// function of the operand data: the matrix pipe's rate under the 1400 W socket power cap.  Run under tools/power_clocks.py --cmd so that
// socket power / gfxclk / PPT residency are sampled while it loops.
//     hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_power tools/microbench/mfma_power.hip && /tmp/mfma_power <pattern> <seconds> [waves_per_simd]
// pattern 0: all-zero operands; 1: one small constant per lane; 2: N(0,1) bf16 operands, the SAME a / b registers for every MFMA;
//         3: N(0,1) bf16 operands, a 256 x 256 tile's register traffic — 4 A x 2 B fragments per k-step (8 MFMAs), two such sets alternating
//            (what a GEMM main loop presents to the pipe: every MFMA sees other operands than the one before it).
// Each wave keeps 8 independent 32 x 32 accumulators (no dependent-issue stalls); 1 or 2 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ inline float gauss(uint32_t& s) {  // sum of 4 uniforms, variance-normalised: close enough to N(0,1) for a toggle-rate experiment
  float v = 0.f;
  for (int i = 0; i < 4; ++i) { s = mix32(s + 0x9e3779b9u); v += (float)(s >> 8) * (1.0f / 16777216.0f); }
  return (v - 2.0f) * 1.7320508f;
}

template <int PATTERN>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  uint32_t seed = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a[2][4], b[2][2];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) a[s][i][j] = PATTERN == 0 ? (__bf16)0.f : PATTERN == 1 ? (__bf16)(0.001f * lane) : (__bf16)gauss(seed);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 8; ++j) b[s][i][j] = PATTERN == 0 ? (__bf16)0.f : PATTERN == 1 ? (__bf16)(0.002f * j) : (__bf16)(0.05f * gauss(seed));
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (PATTERN == 3) acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][mi], b[s][ni], acc[mi * 2 + ni], 0, 0, 0);
          else acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[0][0], acc[mi * 2 + ni], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
  if (s == 1.2345e33f) sink[0] = s;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int pattern = argc > 1 ? atoi(argv[1]) : 3;
  const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
  const int wps = argc > 3 ? atoi(argv[3]) : 2;
  float* sink;
  CHECK(hipMalloc(&sink, 4));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, threads = 256 * wps, iters = 20000;
  auto launch = [&]() {
    switch (pattern) {
      case 0: hipLaunchKernelGGL(mfma_loop<0>, dim3(cus), dim3(threads), 0, 0, iters, sink); break;
      case 1: hipLaunchKernelGGL(mfma_loop<1>, dim3(cus), dim3(threads), 0, 0, iters, sink); break;
      case 2: hipLaunchKernelGGL(mfma_loop<2>, dim3(cus), dim3(threads), 0, 0, iters, sink); break;
      default: hipLaunchKernelGGL(mfma_loop<3>, dim3(cus), dim3(threads), 0, 0, iters, sink); break;
    }
  };
  launch();
  CHECK(hipDeviceSynchronize());
  const double flop_per_launch = (double)cus * (threads / 64) * iters * 16.0 * (2.0 * 32 * 32 * 16);
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
  printf("pattern %d  waves/SIMD %d  CUs %d: %d launches, mean %.1f TFLOP/s over the run, last 4 launches %.1f TFLOP/s (peak at 2.4 GHz: %.0f)\n", pattern, wps, cus, n,
         flop_per_launch * n / (ms_total * 1e-3) / 1e12, flop_per_launch * 4 / (ms_last * 1e-3) / 1e12, cus * 4 * 1024 * 2.4e9 / 1e12);
  return 0;
}
