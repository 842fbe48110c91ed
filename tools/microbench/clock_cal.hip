// clock_cal.hip — what does s_memtime tick at, and what shader clock does a heavy MFMA kernel really get?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
  out[0] = __builtin_amdgcn_s_memtime() - t0;
}
// every wave: N dependent-free MFMAs; reports memtime ticks and clock64 per wave
__global__ __launch_bounds__(512) void mfma_burn(int iters, unsigned long long* out, float* sink) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.37f * (threadIdx.x % 7)); b[j] = (__bf16)(0.11f * j - 0.3f); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long c1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][3];
  if (s == 1.234e33f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; }
}
int main() {
  unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long h[2]; float ms;
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 1000ull, d); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 100000000ull, d); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("idle spin: %llu memtime ticks in %.3f ms -> %.1f MHz\n", h[0], ms, h[0] / (ms * 1e3));
  for (int rep = 0; rep < 3; ++rep) {
    const int iters = 200000;  // per wave: 1.6M MFMAs; 2 waves/SIMD -> 3.2M x 32 cycles per SIMD
    hipEventRecord(e0); hipLaunchKernelGGL(mfma_burn, dim3(256), dim3(512), 0, 0, iters, d, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mfma_cycles_per_simd = 2.0 * iters * 8 * 32;
    printf("mfma burn: %.3f ms; memtime ticks %llu (%.1f MHz), clock64 %llu; MFMA-bound shader clock >= %.0f MHz; %.0f TFLOP/s\n", ms, h[0],
           h[0] / (ms * 1e3), h[1], mfma_cycles_per_simd / (ms * 1e3), 256.0 * 8 * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
