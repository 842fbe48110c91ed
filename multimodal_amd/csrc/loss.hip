// loss.hip — contrastive logits + cross entropy in fp32 (gfx950).
//   logits kernel : one wave per 32x32 logit tile on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32, a
//                   k-ordered fmaf chain, 1/16 of the bf16 MFMA rate) — the loss is 0.001 % of the
//                   step's FLOPs, so it is kept in fp32 to leave the argmax/ordering of the logits
//                   as close to the reference's fp32 matmul as the features allow.
//   ce_rows kernel: one wave per logit row: max / log-sum-exp / label pick / smoothing term.
//   ce_reduce     : one block: masked mean (or sum) over rows, both directions, final average.
// Reference: modules/losses/contrastive_loss_with_temperature.py:81,90-107.
#include "common.h"

namespace mmamd {

// dir 0: logits_a = a . b_all^T ; dir 1: logits_b = b . a_all^T
__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ a_all, const float* __restrict__ b_all,
                                                     int ld_all, const float* __restrict__ logit_scale, int B, int WB,
                                                     int E, float* __restrict__ logits_a, float* __restrict__ logits_b) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int dir = blockIdx.z;
  const float* L = dir == 0 ? a : b;          // [B,E], row stride E
  const float* R = dir == 0 ? b_all : a_all;  // [WB,E], row stride ld_all
  float* out = dir == 0 ? logits_a : logits_b;
  const int i0 = blockIdx.y * 32;
  const int j0 = (blockIdx.x * 4 + wv) * 32;
  if (j0 >= WB) return;  // wave-uniform
  const int half = lane >> 5;
  int ri = i0 + (lane & 31); ri = ri < B ? ri : B - 1;
  int rj = j0 + (lane & 31); rj = rj < WB ? rj : WB - 1;
  const float* lp = L + (size_t)ri * E;
  const float* rp = R + (size_t)rj * ld_all;
  const bool vec = ((E & 3) == 0) && ((ld_all & 3) == 0) && ((reinterpret_cast<uintptr_t>(L) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(R) & 15) == 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int e0 = 0; e0 < E; e0 += 8) {
    const int e = e0 + 4 * half;
    f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
    if (vec && e + 3 < E) {
      x = load4(lp + e);
      y = load4(rp + e);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e + u < E) { x[u] = lp[e + u]; y[u] = rp[e + u]; }
    }
    // k-slot (half) of MFMA #u holds feature index e0 + 4*half + u for BOTH operands
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u], acc, 0, 0, 0);
  }
  const float T = expf(*logit_scale);
  const int j = j0 + (lane & 31);
  if (j < WB) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < B) out[(size_t)i * WB + j] = acc[r] * T;
    }
  }
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits_a,
                                                      const float* __restrict__ logits_b, int B, int WB,
                                                      int label_offset, const uint8_t* __restrict__ row_mask,
                                                      float smoothing, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= 2 * B) return;
  const int dir = gw / B, row = gw - dir * B;
  if (row_mask != nullptr && row_mask[row] == 0) {
    if (lane == 0) ws[gw] = 0.f;
    return;
  }
  const float* lr = (dir == 0 ? logits_a : logits_b) + (size_t)row * WB;
  float m = -INFINITY;
  for (int j = lane; j < WB; j += 64) m = fmaxf(m, lr[j]);
  m = wave_max(m);
  float se = 0.f, sl = 0.f;
  for (int j = lane; j < WB; j += 64) {
    const float v = lr[j];
    se += expf(v - m);
    sl += v;
  }
  se = wave_sum(se);
  sl = wave_sum(sl);
  if (lane == 0) {
    const float lse = m + logf(se);
    const float nll = lse - lr[label_offset + row];
    const float smooth = lse - sl / (float)WB;
    ws[gw] = (1.f - smoothing) * nll + smoothing * smooth;
  }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ ws, int B,
                                                        const uint8_t* __restrict__ row_mask, int reduction,
                                                        float* __restrict__ out3) {
  __shared__ float red[3][4];
  float sa = 0.f, sb = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const bool valid = row_mask == nullptr || row_mask[i] != 0;
    if (valid) { sa += ws[i]; sb += ws[B + i]; cnt += 1.f; }
  }
  sa = wave_sum(sa); sb = wave_sum(sb); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; red[2][threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float la = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    float lb = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float n = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    if (reduction == MMAMD_REDUCE_MEAN) { la /= n; lb /= n; }  // n == 0 -> nan, as F.cross_entropy
    out3[0] = (la + lb) * 0.5f;
    out3[1] = la;
    out3[2] = lb;
  }
}

}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_contrastive_fwd(const float* a, const float* b, const float* a_all, const float* b_all,
                                     int ld_all, const float* logit_scale, int B, int WB, int E, int label_offset,
                                     const uint8_t* row_mask, float label_smoothing, int reduction, float* logits_a,
                                     float* logits_b, float* out3, float* ws, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(a && b && a_all && b_all && logit_scale && logits_a && logits_b && out3 && ws, MMAMD_E_BADARG,
                  "contrastive_fwd: null pointer");
  MMAMD_CHECK_ARG(B > 0 && WB >= B && E > 0 && ld_all >= E, MMAMD_E_BADARG,
                  "contrastive_fwd: bad sizes B=%d WB=%d E=%d ld_all=%d", B, WB, E, ld_all);
  MMAMD_CHECK_ARG(label_offset >= 0 && label_offset + B <= WB, MMAMD_E_BADARG,
                  "contrastive_fwd: labels %d..%d outside [0,%d)", label_offset, label_offset + B, WB);
  MMAMD_CHECK_ARG(reduction == MMAMD_REDUCE_MEAN || reduction == MMAMD_REDUCE_SUM, MMAMD_E_UNSUPPORTED,
                  "contrastive_fwd: reduction must be mean or sum");
  hipStream_t st = (hipStream_t)stream;
  const dim3 g1((WB + 127) / 128, (B + 31) / 32, 2);
  hipLaunchKernelGGL(logits_kernel, g1, dim3(256), 0, st, a, b, a_all, b_all, ld_all, logit_scale, B, WB, E, logits_a, logits_b);
  hipLaunchKernelGGL(ce_rows_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, logits_a, logits_b, B, WB, label_offset,
                     row_mask, label_smoothing, ws);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, ws, B, row_mask, reduction, out3);
  return launch_status("contrastive_fwd");
}
