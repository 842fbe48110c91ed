// evalops.hip — zero-shot classification / retrieval read-outs built on the towers' embeddings:
// the torch expressions of examples/flava/native/utils.py:100-160 (`_zero_shot_classifier`, `_accuracy`, `run_imagenet_zero_shot`)
// and examples/flava/coco_zero_shot.py:24-31,78-90 (`compute_recall`, normalised similarity) as three row kernels.
// A top-k hit test needs no sort: "target is among the k largest" == "fewer than k entries beat the target's score".
#include "common.h"

namespace mmamd {
namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {  // red: 4 floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// out[g, :] = normalize(mean_t normalize(x[g*T + t, :]))   (utils.py:108-111; plain x / |x|, no epsilon, like the reference)
constexpr int kMaxPerLane = 16;  // d <= 64 * 16 * ... per wave stride: d <= 4096 with 256 threads
__global__ void __launch_bounds__(256) group_mean_normalize_kernel(const float* __restrict__ x, int T, int d, float* __restrict__ out) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  float acc[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) acc[i] = 0.f;
  for (int t = 0; t < T; ++t) {
    const float* row = x + ((size_t)g * T + t) * d;
    float v[kMaxPerLane];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const int k = threadIdx.x + i * 256;
      v[i] = k < d ? row[k] : 0.f;
      ss += v[i] * v[i];
    }
    const float n = sqrtf(block_sum_256(ss, red));
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) acc[i] += v[i] / n;
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    acc[i] = acc[i] / (float)T;
    ss += acc[i] * acc[i];
  }
  const float n = sqrtf(block_sum_256(ss, red));
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int k = threadIdx.x + i * 256;
    if (k < d) out[(size_t)g * d + k] = acc[i] / n;
  }
}

// y = scale * x / |x|   (utils.py:141-142: features /= norm; 100.0 * features)
__global__ void __launch_bounds__(256) scale_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int d,
                                                              float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * d;
  float ss = 0.f;
  for (int k = lane; k < d; k += 64) ss += xr[k] * xr[k];
  const float n = sqrtf(wave_sum(ss));
  float* yr = y + (size_t)row * d;
  for (int k = lane; k < d; k += 64) yr[k] = scale * (xr[k] / n);
}

// rank[r] = how many entries of row r beat the target's score (ties: the lower index wins, as a stable descending sort would)
__global__ void __launch_bounds__(256) target_rank_kernel(const float* __restrict__ s, int64_t ld, const int64_t* __restrict__ target,
                                                          int C, int32_t* __restrict__ rank) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const int64_t t = target ? target[r] : (int64_t)r;
  if (t < 0 || t >= C) {  // uniform per block
    if (threadIdx.x == 0) rank[r] = C;
    return;
  }
  const float* row = s + (size_t)r * ld;
  const float st = row[t];
  int cnt = 0;
  for (int j = threadIdx.x; j < C; j += 256) {
    const float v = row[j];
    cnt += (v > st) || (v == st && j < t);
  }
  const float total = block_sum_256((float)cnt, red);  // C < 2^24: exact in fp32
  if (threadIdx.x == 0) rank[r] = (int32_t)total;
}

}  // namespace
}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_group_mean_normalize(const float* x, int G, int T, int d, float* out, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && out && G >= 0 && T > 0 && d > 0, MMAMD_E_BADARG, "group_mean_normalize: bad argument");
  MMAMD_CHECK_ARG(d <= 256 * kMaxPerLane, MMAMD_E_UNSUPPORTED, "group_mean_normalize: d=%d above %d", d, 256 * kMaxPerLane);
  if (G == 0) return 0;
  hipLaunchKernelGGL(group_mean_normalize_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, x, T, d, out);
  return launch_status("group_mean_normalize");
}

extern "C" int mmamd_scale_normalize(const float* x, float* y, int rows, int d, float scale, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && y && rows >= 0 && d > 0, MMAMD_E_BADARG, "scale_normalize: bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(scale_normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, d, scale);
  return launch_status("scale_normalize");
}

extern "C" int mmamd_target_rank(const float* scores, int64_t ld, const int64_t* target, int R, int C, int32_t* rank,
                                 mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(scores && rank && R >= 0 && C > 0 && ld >= C, MMAMD_E_BADARG, "target_rank: bad argument");
  MMAMD_CHECK_ARG(C < (1 << 24), MMAMD_E_UNSUPPORTED, "target_rank: C=%d above 2^24", C);
  if (R == 0) return 0;
  hipLaunchKernelGGL(target_rank_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, scores, ld, target, C, rank);
  return launch_status("target_rank");
}
