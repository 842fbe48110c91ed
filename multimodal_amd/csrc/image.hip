// image.hip — the input side of the image tower: decoded uint8 RGB pixels of a ragged batch ->
// Resize(bicubic) + crop + ToTensor + Normalize (+ im2col for the patch-embedding GEMM), replacing the per-image PIL / torch
// host loop of torchmultimodal/transforms/clip_transform.py:326-352.
//
// Byte / integer work, HBM-bound and small: no MFMA, no LDS.  The resampling is Pillow's (Resample.c) two 8-bit passes with its
// fixed-point coefficients (22 fractional bits, built by the host in double precision exactly as precompute_coeffs does):
//   pass H  source view rows [row0, row0+nrows) x the crop_w output columns the crop keeps  -> uint8 tmp [nrows][crop_w][3]
//   pass V  the crop_h output rows from tmp -> uint8 -> x/255 -> (x - mean)/std in fp32 (IEEE divide, as torch's CPU kernels)
//           -> any of: fp32 [B,3,crop_h,crop_w] (what the reference returns), bf16 patch rows [B*G2, kpad] (column
//           (c*P+py)*P+px: the GEMM operand), uint8 [B,crop_h,crop_w,3] (the resized crop itself).
// Each thread owns one output byte (H) / one output pixel (V); neighbouring threads read neighbouring, overlapping source windows,
// so the loads coalesce; the uint8 intermediate between the passes is part of the algorithm (Pillow rounds there).
#include "common.h"

namespace mmamd {
namespace {

constexpr int kPrec = 32 - 8 - 2;
constexpr int kDesc = 16;  // int64 words per image, see mmamd.h

__device__ __forceinline__ uint8_t clip8(int acc) {
  const int v = acc >> kPrec;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ void __launch_bounds__(256) resample_h_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ tables,
                                                         uint8_t* __restrict__ tmp, int crop_w) {
  const int64_t* d = desc + (size_t)blockIdx.z * kDesc;
  const int r = blockIdx.y;
  if (r >= (int)d[5]) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= crop_w * 3) return;
  const int x = t / 3, c = t - x * 3;
  const int ks = (int)d[8];
  const int32_t* kk = tables + d[6] + (size_t)x * ks;
  const int32_t* bd = tables + d[7] + 2 * x;
  const int x0 = bd[0], n = bd[1];
  const int px = (int)d[13];
  const uint8_t* src = reinterpret_cast<const uint8_t*>(d[0]) + (size_t)(d[4] + r) * d[1] + (size_t)x0 * px + c;
  int acc = 1 << (kPrec - 1);
  for (int k = 0; k < n; ++k) acc += (int)src[(size_t)k * px] * kk[k];
  tmp[d[12] + ((size_t)r * crop_w + x) * 3 + c] = clip8(acc);
}

__global__ void __launch_bounds__(256) resample_v_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ tables,
                                                         const uint8_t* __restrict__ tmp, int crop_h, int crop_w, float m0, float m1,
                                                         float m2, float s0, float s1, float s2, float* __restrict__ out_f32,
                                                         bf16* __restrict__ patches, int P, int kpad, uint8_t* __restrict__ out_u8) {
  const int64_t* d = desc + (size_t)blockIdx.z * kDesc;
  const int y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= crop_w) return;
  const int b = blockIdx.z;
  const int ks = (int)d[11];
  const int32_t* kk = tables + d[9] + (size_t)y * ks;
  const int32_t* bd = tables + d[10] + 2 * y;
  const int y0 = bd[0], n = bd[1];
  const uint8_t* src = tmp + d[12] + ((size_t)y0 * crop_w + x) * 3;
  const size_t rs = (size_t)crop_w * 3;
  int a0 = 1 << (kPrec - 1), a1 = a0, a2 = a0;
  for (int k = 0; k < n; ++k) {
    const int w = kk[k];
    const uint8_t* p = src + (size_t)k * rs;
    a0 += (int)p[0] * w;
    a1 += (int)p[1] * w;
    a2 += (int)p[2] * w;
  }
  const uint8_t u[3] = {clip8(a0), clip8(a1), clip8(a2)};
  if (out_u8) {
    uint8_t* o = out_u8 + (((size_t)b * crop_h + y) * crop_w + x) * 3;
    o[0] = u[0]; o[1] = u[1]; o[2] = u[2];
  }
  if (!out_f32 && !patches) return;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  const int gw = patches ? crop_w / P : 0, gh = patches ? crop_h / P : 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u[c], 255.0f), mean[c]), sd[c]);
    if (out_f32) out_f32[(((size_t)b * 3 + c) * crop_h + y) * crop_w + x] = v;
    if (patches) {
      const size_t row = ((size_t)b * gh + y / P) * gw + x / P;
      patches[row * kpad + (c * P + y % P) * P + x % P] = (bf16)v;
    }
  }
}

}  // namespace
}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_image_resample(const int64_t* desc, const int32_t* tables, uint8_t* tmp, int B, int crop_h, int crop_w,
                                    int max_rows, const float* mean, const float* std, float* out_f32, void* patches, int P,
                                    int kpad, uint8_t* out_u8, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(desc && tables && tmp && mean && std && B >= 0 && crop_h > 0 && crop_w > 0 && max_rows > 0, MMAMD_E_BADARG,
                  "image_resample: bad argument");
  MMAMD_CHECK_ARG(out_f32 || patches || out_u8, MMAMD_E_BADARG, "image_resample: no output requested");
  MMAMD_CHECK_ARG(B <= 65535 && max_rows <= 65535 && crop_h <= 65535, MMAMD_E_UNSUPPORTED,
                  "image_resample: B=%d / rows=%d / crop_h=%d above the 65535 grid limit", B, max_rows, crop_h);
  if (patches)
    MMAMD_CHECK_ARG(P > 0 && crop_h % P == 0 && crop_w % P == 0 && kpad >= 3 * P * P, MMAMD_E_BADARG,
                    "image_resample: crop %dx%d is not a grid of %d-pixel patches with kpad=%d >= 3*P*P", crop_h, crop_w, P, kpad);
  for (int c = 0; c < 3; ++c) MMAMD_CHECK_ARG(std[c] != 0.f, MMAMD_E_BADARG, "image_resample: std[%d] is zero", c);
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(resample_h_kernel, dim3((crop_w * 3 + 255) / 256, max_rows, B), dim3(256), 0, st, desc, tables, tmp, crop_w);
  hipLaunchKernelGGL(resample_v_kernel, dim3((crop_w + 255) / 256, crop_h, B), dim3(256), 0, st, desc, tables, tmp, crop_h, crop_w,
                     mean[0], mean[1], mean[2], std[0], std[1], std[2], out_f32, (bf16*)patches, P, kpad, out_u8);
  return launch_status("image_resample");
}
