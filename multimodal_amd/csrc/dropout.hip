// dropout.hip — training-time dropout and stochastic depth (reference: nn.Dropout in modules/layers/mlp.py:40,59-60,
// modules/layers/transformer.py:64-70,86-93, torchvision.ops.StochasticDepth(mode="row") at transformer.py:64-67).
//
//   out[i] = (res ? res[i] : 0) + x[i] * keep(i) / (1 - p)
//
// keep() comes from a counter-based generator — Philox4x32-10 keyed by the 64-bit `seed`, counter = (element group, site) — so a mask is
// a pure function of (seed, site, index): the backward regenerates it instead of storing it (dx = dy * keep / (1 - p): the same kernel
// with res = NULL), and oracle/philox.py restates the generator in numpy to pin every mask bit.
//   mode 0 (dropout):          one decision per ELEMENT; the 4 outputs of counter (i / 4, site) serve elements 4 (i / 4) .. + 3
//   mode 1 (stochastic depth): one decision per SAMPLE, group = elements per sample; sample s uses output (s & 3) of counter (s / 4, site)
// keep <=> r >= floor(p * 2^32) on the 32-bit output r (integer compare: no float rounding in the decision).
// HBM-bound elementwise work: 16 bytes per lane, one Philox block per 4 elements (~60 integer ops: hidden under the memory time).
#include "common.h"

namespace mmamd {

template <typename TX, typename TO, int MODE>
__global__ __launch_bounds__(256) void dropout_kernel(const TX* x, const float* __restrict__ res, TO* out,  // x and out may alias (in-place form)
                                                      uint8_t* __restrict__ mask_out, long long n4, long long group, uint32_t thresh,
                                                      float scale, uint32_t k0, uint32_t k1, uint32_t site) {
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n4; g += (long long)gridDim.x * 256) {
    bool keep[4];
    if constexpr (MODE == 0) {
      const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), site, 0u, k0, k1);
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = r.v[j] >= thresh;
    } else {
      const long long s = (4 * g) / group;  // group % 4 == 0 (checked by the launcher): the 4 elements share their sample
      const Philox4 r = philox4x32_10((uint32_t)(s >> 2), (uint32_t)(s >> 34), site, 0u, k0, k1);
      const bool k = r.v[s & 3] >= thresh;
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = k;
    }
    const f32x4 xv = load4(x + 4 * g);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (res != nullptr) o = load4(res + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += keep[j] ? xv[j] * scale : 0.f;
    store4(out + 4 * g, o);
    if (mask_out != nullptr) {
      const uint32_t m = (uint32_t)keep[0] | ((uint32_t)keep[1] << 8) | ((uint32_t)keep[2] << 16) | ((uint32_t)keep[3] << 24);
      *reinterpret_cast<uint32_t*>(mask_out + 4 * g) = m;
    }
  }
}

}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_dropout(const void* x, int x_dtype, const float* residual, void* out, int out_dtype, uint8_t* mask_out, int64_t n,
                             int64_t group, float p, uint64_t seed, uint32_t site, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && out && n >= 0, MMAMD_E_BADARG, "dropout: bad argument");
  MMAMD_CHECK_ARG(p >= 0.f && p < 1.f, MMAMD_E_BADARG, "dropout: p = %g must be in [0, 1)", (double)p);
  MMAMD_CHECK_ARG(n % 4 == 0 && (group == 0 || (group > 0 && group % 4 == 0 && n % group == 0)), MMAMD_E_UNSUPPORTED,
                  "dropout: n = %lld (and the per-sample group) must be multiples of 4", (long long)n);
  MMAMD_CHECK_ARG((x_dtype == MMAMD_F32 || x_dtype == MMAMD_BF16) && (out_dtype == MMAMD_F32 || out_dtype == MMAMD_BF16), MMAMD_E_BADARG, "dropout: bad dtype code");
  MMAMD_CHECK_ARG(aligned16(residual) && ((uintptr_t)x & (x_dtype == MMAMD_F32 ? 15 : 7)) == 0 && ((uintptr_t)out & (out_dtype == MMAMD_F32 ? 15 : 7)) == 0 &&
                      ((uintptr_t)mask_out & 3) == 0, MMAMD_E_ALIGN, "dropout: pointers must be aligned to 4 elements");
  if (n == 0) return 0;
  const long long n4 = n / 4;
  const uint32_t thresh = dropout_threshold(p);  // keep <=> r >= thresh: P(keep) = 1 - thresh / 2^32
  const float scale = 1.0f / (1.0f - p);
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
  hipStream_t st = (hipStream_t)stream;
#define DROP_LAUNCH(TX, TO)                                                                                                                  \
  do {                                                                                                                                       \
    if (group == 0) hipLaunchKernelGGL((dropout_kernel<TX, TO, 0>), dim3(grid), dim3(256), 0, st, (const TX*)x, residual, (TO*)out, mask_out, \
                                       n4, (long long)0, thresh, scale, k0, k1, site);                                                       \
    else hipLaunchKernelGGL((dropout_kernel<TX, TO, 1>), dim3(grid), dim3(256), 0, st, (const TX*)x, residual, (TO*)out, mask_out, n4,      \
                            (long long)group, thresh, scale, k0, k1, site);                                                                  \
  } while (0)
  if (x_dtype == MMAMD_F32 && out_dtype == MMAMD_F32) DROP_LAUNCH(float, float);
  else if (x_dtype == MMAMD_F32) DROP_LAUNCH(float, bf16);
  else if (out_dtype == MMAMD_F32) DROP_LAUNCH(bf16, float);
  else DROP_LAUNCH(bf16, bf16);
#undef DROP_LAUNCH
  return launch_status("dropout");
}
