// atomic_rmw.hip — can the fp32 residual read-modify-write of the out-projection / MLP-down epilogues be handed to the L2's atomic units?
// x[i] += v[i] over a 155 MB fp32 array (the ViT-B/16 residual stream at B = 256), three ways: (a) load + add + store, 16 B per lane;
// (b) global_atomic_add_f32 without return, one dword per lane, a wave-instruction = 256 contiguous bytes; (c) the same with the 4 floats of
// a lane's float4 as 4 atomics (what an epilogue that keeps its 16-byte register layout would issue).  Times and checks the results.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench/atomic_rmw.hip -o /tmp/atomic_rmw && /tmp/atomic_rmw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void rmw_plain(float4* x, const float4* v, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 a = x[i], b = v[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    x[i] = a;
  }
}
__global__ void rmw_atomic_dword(float* x, const float* v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) unsafeAtomicAdd(x + i, v[i]);
}
__global__ void rmw_atomic_4(float* x, const float4* v, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const float4 b = v[i];
    unsafeAtomicAdd(x + 4 * i + 0, b.x);
    unsafeAtomicAdd(x + 4 * i + 1, b.y);
    unsafeAtomicAdd(x + 4 * i + 2, b.z);
    unsafeAtomicAdd(x + 4 * i + 3, b.w);
  }
}

int main() {
  const size_t n = (size_t)50432 * 768;
  float *x, *v;
  hipMalloc(&x, n * 4);
  hipMalloc(&v, n * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000) * 0.001f;
  hipMemcpy(v, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[3] = {"load + add + store (float4)", "global_atomic_add_f32, dword per lane", "4 dword atomics per lane (float4 layout)"};
  for (int grid : {2048, 8192}) {
    for (int k = 0; k < 3; ++k) {
      hipMemset(x, 0, n * 4);
      auto run = [&]() {
        if (k == 0) hipLaunchKernelGGL(rmw_plain, dim3(grid), dim3(256), 0, 0, (float4*)x, (const float4*)v, n / 4);
        else if (k == 1) hipLaunchKernelGGL(rmw_atomic_dword, dim3(grid), dim3(256), 0, 0, x, v, n);
        else hipLaunchKernelGGL(rmw_atomic_4, dim3(grid), dim3(256), 0, 0, x, (const float4*)v, n / 4);
      };
      for (int i = 0; i < 3; ++i) run();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) run();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), x, 4 * 1000, hipMemcpyDeviceToHost);
      printf("grid %5d  %-44s %7.1f us per pass  (%.2f TB/s of 3 x %.0f MB)  x[7] = %.4f (expect %.4f)\n", grid, names[k], ms * 100, 3.0 * n * 4 / (ms * 100) / 1e6,
             n * 4 / 1e6, h[7], 13 * 0.007f);
      for (size_t i = 0; i < 1000; ++i) h[i] = (float)(i % 1000) * 0.001f;
    }
  }
  return 0;
}
