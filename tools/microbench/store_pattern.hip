// store_pattern.hip — how fast MI355X takes 477 MB of whole-cache-line fp32 stores (global_store_dwordx4, 1 KB per wave-instruction) as a function of WHERE
// the chip's workgroups write at the same time.  Question behind it (r05, csrc/attention_probs_lse.hip): the probabilities kernel with everything but its
// stores switched off takes 112 us for the bytes a fill_ writes in 70 — is that the store instructions or the address pattern (3072 workgroups, each walking
// its own 155 KB block)?
//     hipcc -O3 --offload-arch=gfx950 -o /tmp/store_pattern tools/microbench/store_pattern.hip && /tmp/store_pattern
// Patterns (all write the same 3072 x 155236-byte blocks = [256, 12, 197, 197] fp32, every float exactly once, 1 KB-aligned chunks, ragged ends float by float
// omitted: chunks are clipped to the tensor, which is allocated with slack):
//   0  fill: grid-stride, wave-instruction i of the whole grid writes chunk i (the compact moving window a fill kernel produces)
//   1  one workgroup (8 waves) per BLOCK, walking its block front to back, 8 KB per round (wave w: chunk 8 r + w)      [the kernel's pattern]
//   2  one workgroup per 25 KB BAND (7 per block, the last one short), bands in memory order: workgroup i writes band i
//   3  as 2, but the 7 bands of a block go to ONE XCD: workgroup i -> xcd = i & 7, j = i >> 3, block = 8 (j / 7) + xcd, band = j % 7
//   4  as 1 with 4 waves per workgroup (twice the workgroups per CU)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kBlocks = 3072, kS = 197, kBlockFloats = kS * kS, kBandFloats = 32 * kS;

__device__ inline void store_chunks(float* base, long long first_float, long long n_floats, int wave, int nwaves, int lane, float val) {
  // write floats [first_float, first_float + n_floats) of `base` in 1 KB-aligned chunks (clipped at both ends to 16-byte units: the slack absorbs it)
  const long long c0 = first_float >> 8, c1 = (first_float + n_floats + 255) >> 8;
  for (long long c = c0 + wave; c < c1; c += nwaves) {
    const long long f = c * 256 + lane * 4;
    if (f + 4 > first_float && f < first_float + n_floats) *reinterpret_cast<f32x4*>(base + f) = f32x4{val, val, val, val};
  }
}

template <int PATTERN, int NW>
__global__ __launch_bounds__(NW * 64) void store_kernel(float* base, float val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (PATTERN == 0) {
    const long long total = (long long)kBlocks * kBlockFloats / 256;
    for (long long c = (long long)blockIdx.x * NW + wave; c < total; c += (long long)gridDim.x * NW)
      *reinterpret_cast<f32x4*>(base + c * 256 + lane * 4) = f32x4{val, val, val, val};
  } else if constexpr (PATTERN == 1 || PATTERN == 4) {
    const long long first = (long long)blockIdx.x * kBlockFloats;
    for (int band = 0; band < 7; ++band) {
      const int nf = (band < 6 ? 32 : kS - 192) * kS;
      store_chunks(base, first + (long long)band * kBandFloats, nf, wave, NW, lane, val);
    }
  } else {
    int block, band;
    if constexpr (PATTERN == 2) { block = blockIdx.x / 7; band = blockIdx.x % 7; }
    else { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; block = 8 * (j / 7) + x; band = j % 7; }
    const int nf = (band < 6 ? 32 : kS - 192) * kS;
    store_chunks(base, (long long)block * kBlockFloats + (long long)band * kBandFloats, nf, wave, NW, lane, val);
  }
}

template <typename F>
static float timed(F launch, int n = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / n;
}

int main() {
  const size_t floats = (size_t)kBlocks * kBlockFloats + 4096;
  float* buf = nullptr;
  if (hipMalloc(&buf, floats * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  float* base = buf + 1024;  // slack in front for clipped chunks
  const double mb = (double)kBlocks * kBlockFloats * 4 / 1e6;
  for (int rnd = 0; rnd < 2; ++rnd) {
    float t;
    t = timed([&] { hipLaunchKernelGGL((store_kernel<0, 4>), dim3(256 * 8), dim3(256), 0, 0, base, 1.f); });
    printf("pattern 0  fill, grid-stride chunks                          %7.1f us  %5.2f TB/s\n", t, mb / t);
    t = timed([&] { hipLaunchKernelGGL((store_kernel<1, 8>), dim3(kBlocks), dim3(512), 0, 0, base, 2.f); });
    printf("pattern 1  workgroup (8 waves) per 155 KB block              %7.1f us  %5.2f TB/s\n", t, mb / t);
    t = timed([&] { hipLaunchKernelGGL((store_kernel<4, 4>), dim3(kBlocks), dim3(256), 0, 0, base, 2.f); });
    printf("pattern 4  workgroup (4 waves) per 155 KB block              %7.1f us  %5.2f TB/s\n", t, mb / t);
    t = timed([&] { hipLaunchKernelGGL((store_kernel<2, 8>), dim3(kBlocks * 7), dim3(512), 0, 0, base, 3.f); });
    printf("pattern 2  workgroup per 25 KB band, memory order            %7.1f us  %5.2f TB/s\n", t, mb / t);
    t = timed([&] { hipLaunchKernelGGL((store_kernel<3, 8>), dim3(kBlocks * 7), dim3(512), 0, 0, base, 4.f); });
    printf("pattern 3  workgroup per band, a block's bands on one XCD    %7.1f us  %5.2f TB/s\n", t, mb / t);
    t = timed([&] { hipLaunchKernelGGL((store_kernel<2, 4>), dim3(kBlocks * 7), dim3(256), 0, 0, base, 3.f); });
    printf("pattern 2  ... with 4 waves per workgroup                    %7.1f us  %5.2f TB/s\n", t, mb / t);
  }
  hipFree(buf);
  return 0;
}
