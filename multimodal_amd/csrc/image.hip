// image.hip — the input side of the image tower: decoded uint8 RGB pixels of a ragged batch ->
// Resize(bicubic) + crop + ToTensor + Normalize (+ im2col for the patch-embedding GEMM), replacing the per-image PIL / torch
// host loop of torchmultimodal/transforms/clip_transform.py:326-352.
//
// Byte / integer work, HBM-bound by nature and small: no MFMA.  The resampling is Pillow's (Resample.c) two 8-bit passes with its
// fixed-point coefficients (22 fractional bits, built by the host in double precision exactly as precompute_coeffs does):
//   pass H  source view rows [row0, row0+nrows) x the crop_w output columns the crop keeps  -> uint8 tmp [nrows][crop_w][3]
//   pass V  the crop_h output rows from tmp -> uint8 -> value table [3][256] (the host tabulates ToTensor + Normalize, or FLAVA's
//           map_pixels, in IEEE fp32: 256 possible results per channel) -> any of: fp32 [B,3,crop_h,crop_w] (what the reference
//           returns), bf16 patch rows [B*G2, kpad] (column (c*P+py)*P+px: the GEMM operand), uint8 [B,crop_h,crop_w,3] (the resized
//           crop itself).
// The uint8 intermediate between the passes is part of the algorithm (Pillow rounds there).  Two implementations of each pass:
//   tiled   (the one that runs for ordinary images) H: a block stages the source segment of R rows (and the image's coefficient
//           table) in LDS with aligned 16-byte loads, one thread per output pixel walks its taps out of LDS for 4 rows at a time
//           (RGB: four taps = 12 bytes per LDS read), the R x crop_w x 3 result tile goes back through LDS as dword stores.
//           V: one thread per 4 output pixels = 12 consecutive bytes = one 12-byte load per tap row, 12 accumulators, value-table
//           lookups out of LDS, vector stores (f32x4 / bf16x4).
//   direct  one thread per output byte (H) / pixel (V) straight from global memory: any crop width, any row length.
// Algorithmic bytes per image: source segment rows + 2 x tmp + outputs.
#include "common.h"

namespace mmamd {
namespace {

constexpr int kPrec = 32 - 8 - 2;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
constexpr int kDesc = 16;  // int64 words per image, see mmamd.h
constexpr int kTapBlock = 8, kTapBlockV = 5;  // taps whose coefficients / source words are fetched together

// Products are pixel (8 bits) x coefficient (|k| < 2^23: the weights of one output sum to 2^22 and the bicubic lobes add < 30 %),
// so the full-rate 24-bit multiply-add is exact.
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
  for (int k = 0; k < n; ++k) acc += __mul24((int)src[(size_t)k * px], kk[k]);
  tmp[d[12] + ((size_t)r * crop_w + x) * 3 + c] = clip8(acc);
}

__global__ void __launch_bounds__(256) resample_v_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ tables,
                                                         const uint8_t* __restrict__ tmp, int crop_h, int crop_w,
                                                         const float* __restrict__ lut, float* __restrict__ out_f32,
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
    a0 += __mul24((int)p[0], w);
    a1 += __mul24((int)p[1], w);
    a2 += __mul24((int)p[2], w);
  }
  const uint8_t u[3] = {clip8(a0), clip8(a1), clip8(a2)};
  if (out_u8) {
    uint8_t* o = out_u8 + (((size_t)b * crop_h + y) * crop_w + x) * 3;
    o[0] = u[0]; o[1] = u[1]; o[2] = u[2];
  }
  if (!out_f32 && !patches) return;
  const int gw = patches ? crop_w / P : 0, gh = patches ? crop_h / P : 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = lut[c * 256 + u[c]];
    if (out_f32) out_f32[(((size_t)b * 3 + c) * crop_h + y) * crop_w + x] = v;
    if (patches) {
      const size_t row = ((size_t)b * gh + y / P) * gw + x / P;
      patches[row * kpad + (c * P + y % P) * P + x % P] = (bf16)v;
    }
  }
}


// The tap walk of the tiled H pass for one block: thread = output pixel, 4 rows x 3 channels of accumulators per trip.
// PX = bytes per source pixel when known at compile time (3: the tap offsets fold into the ds_read immediates), 0 = use px.
// COEF_LDS: the image's coefficient table was staged at LDS byte offset coef_lds (else it is read from global memory).
template <int PX, bool COEF_LDS>
__device__ __forceinline__ void h_tile_taps(int out_tile, int coef_lds, const int64_t* d, const int32_t* tables, const int32_t* bd,
                                            const uint8_t* row_first, int crop_w, int rows, int seg_pitch, int out_pitch, int seg0,
                                            int px_dyn) {
  extern __shared__ __align__(16) uint8_t lds[];  // all LDS addressing below is by integer offset: keeps the ds_ instructions
  const int px = PX ? PX : px_dyn;
  const int ks = (int)d[8];
  const int32_t* lds32 = reinterpret_cast<const int32_t*>(lds);
  for (int x = threadIdx.x; x < crop_w; x += 256) {
    const int32_t* kk = tables + d[6] + (size_t)x * ks;  // used when the coefficient table did not fit the LDS budget
    const int kl = (coef_lds >> 2) + x * ks;
    const int x0 = bd[2 * x], n = bd[2 * x + 1];
    const int off = x0 * px - seg0;
    for (int rb = 0; rb < rows; rb += 4) {
      int acc[4][3];
      int base[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j][0] = acc[j][1] = acc[j][2] = 1 << (kPrec - 1);
        const int r = min(rb + j, rows - 1);
        base[j] = r * seg_pitch + (int)(reinterpret_cast<uintptr_t>(row_first + (size_t)r * d[1]) & 15) + off;
      }
      for (int k0 = 0; k0 < n; k0 += kTapBlock) {  // the coefficients of a tap block are fetched together, then used from registers
        int w[kTapBlock];
#pragma unroll
        for (int k = 0; k < kTapBlock; ++k) {  // unconditional loads (index clamped into the row), zero weight past the taps
          const int kc = min(k0 + k, ks - 1);
          const int c = COEF_LDS ? lds32[kl + kc] : kk[kc];
          w[k] = (k0 + k < n) ? c : 0;
        }
        if (PX == 3) {
          // RGB: four neighbouring taps are 12 consecutive bytes -> one (unaligned) 16-byte LDS read per tap quad and row; a quad
          // that straddles the last tap reads bytes past the window (inside the LDS allocation) under zero weights
#pragma unroll
          for (int k = 0; k < kTapBlock; k += 4) {
            if (k0 + k < n) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                uint4 v;
                __builtin_memcpy(&v, lds + base[j] + k * 3, 16);
                acc[j][0] += __mul24((int)(v.x & 0xffu), w[k]);
                acc[j][1] += __mul24((int)((v.x >> 8) & 0xffu), w[k]);
                acc[j][2] += __mul24((int)((v.x >> 16) & 0xffu), w[k]);
                acc[j][0] += __mul24((int)(v.x >> 24), w[k + 1]);
                acc[j][1] += __mul24((int)(v.y & 0xffu), w[k + 1]);
                acc[j][2] += __mul24((int)((v.y >> 8) & 0xffu), w[k + 1]);
                acc[j][0] += __mul24((int)((v.y >> 16) & 0xffu), w[k + 2]);
                acc[j][1] += __mul24((int)(v.y >> 24), w[k + 2]);
                acc[j][2] += __mul24((int)(v.z & 0xffu), w[k + 2]);
                acc[j][0] += __mul24((int)((v.z >> 8) & 0xffu), w[k + 3]);
                acc[j][1] += __mul24((int)((v.z >> 16) & 0xffu), w[k + 3]);
                acc[j][2] += __mul24((int)(v.z >> 24), w[k + 3]);
              }
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < kTapBlock; ++k) {
            if (k0 + k < n) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int p = base[j] + k * px;
                acc[j][0] += __mul24((int)lds[p], w[k]);
                acc[j][1] += __mul24((int)lds[p + 1], w[k]);
                acc[j][2] += __mul24((int)lds[p + 2], w[k]);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) base[j] += kTapBlock * px;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (rb + j < rows) {
          const int o = out_tile + (rb + j) * out_pitch + x * 3;
          lds[o] = clip8(acc[j][0]); lds[o + 1] = clip8(acc[j][1]); lds[o + 2] = clip8(acc[j][2]);
        }
    }
  }
}

// ---- tiled passes ---------------------------------------------------------------------------------------------------------
// H: block = R consecutive needed rows of one image.  LDS: R x seg_pitch source bytes | R x out_pitch result bytes.
__global__ void __launch_bounds__(256) resample_h_tiled_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ tables,
                                                               uint8_t* __restrict__ tmp, int crop_w, int R, int seg_pitch,
                                                               int coef_cap) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int64_t* d = desc + (size_t)blockIdx.y * kDesc;
  const int nrows = (int)d[5];
  const int r0 = blockIdx.x * R;
  if (r0 >= nrows) return;
  const int rows = min(R, nrows - r0);
  const int px = (int)d[13];
  const int32_t* bd = tables + d[7];
  const int seg0 = bd[0] * px;                                                   // first source byte of the row the crop reads
  const int seg1 = (bd[2 * (crop_w - 1)] + bd[2 * (crop_w - 1) + 1]) * px;        // one past the last
  const int out_pitch = (crop_w * 3 + 15) & ~15;
  uint8_t* out_tile = lds + (size_t)R * seg_pitch;
  // stage: per row, the 16-byte chunks covering [seg0, seg1) -- aligned down, so the first chunk may start before the segment
  // (never before the 16-byte line that holds its first byte)
  const uint8_t* row_first = reinterpret_cast<const uint8_t*>(d[0]) + (size_t)(d[4] + r0) * d[1] + seg0;
  const int cpr = seg_pitch >> 4;  // 16-byte chunks per staged row
  for (int i = threadIdx.x; i < rows * cpr; i += 256) {
    const int r = i / cpr, c = i - r * cpr;
    const uint8_t* row = row_first + (size_t)r * d[1];
    const int mis = (int)(reinterpret_cast<uintptr_t>(row) & 15);
    if (c < ((mis + (seg1 - seg0) + 15) >> 4))
      reinterpret_cast<uint4*>(lds + (size_t)r * seg_pitch)[c] = reinterpret_cast<const uint4*>(row - mis)[c];
  }
  // the image's horizontal coefficient table rides along when it fits its LDS budget (coef_cap ints): no global load in the tap walk
  const int ncoef = crop_w * (int)d[8];
  const int coef_lds = ncoef <= coef_cap ? R * (seg_pitch + out_pitch) : -1;
  if (coef_lds >= 0) {
    const int32_t* g = tables + d[6];
    int32_t* l = reinterpret_cast<int32_t*>(lds + coef_lds);
    for (int i = threadIdx.x; i < ncoef; i += 256) l[i] = g[i];
  }
  __syncthreads();
  if (px == 3 && coef_lds >= 0)
    h_tile_taps<3, true>(R * seg_pitch, coef_lds, d, tables, bd, row_first, crop_w, rows, seg_pitch, out_pitch, seg0, 3);
  else if (px == 3)
    h_tile_taps<3, false>(R * seg_pitch, coef_lds, d, tables, bd, row_first, crop_w, rows, seg_pitch, out_pitch, seg0, 3);
  else if (coef_lds >= 0)
    h_tile_taps<0, true>(R * seg_pitch, coef_lds, d, tables, bd, row_first, crop_w, rows, seg_pitch, out_pitch, seg0, px);
  else
    h_tile_taps<0, false>(R * seg_pitch, coef_lds, d, tables, bd, row_first, crop_w, rows, seg_pitch, out_pitch, seg0, px);
  __syncthreads();
  // tmp rows are crop_w*3 bytes (a multiple of 4 on this path) at a 16-byte aligned image offset: dword stores
  const int row_dw = crop_w * 3 / 4;
  uint32_t* t32 = reinterpret_cast<uint32_t*>(tmp + d[12] + (size_t)r0 * crop_w * 3);
  for (int i = threadIdx.x; i < rows * row_dw; i += 256) {
    const int r = i / row_dw, c = i - r * row_dw;
    t32[i] = reinterpret_cast<const uint32_t*>(out_tile + (size_t)r * out_pitch)[c];
  }
}

// V: thread = 4 pixels of one output row (crop_w % 4 == 0); threads enumerate the crop row-major, or patch-major when the bf16
// patch rows are written as vectors (VEC_PATCH: P % 4 == 0).
template <bool VEC_PATCH>
__global__ void __launch_bounds__(256) resample_v_tiled_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ tables,
                                                               const uint8_t* __restrict__ tmp, int crop_h, int crop_w,
                                                               const float* __restrict__ lut_g, float* __restrict__ out_f32,
                                                               bf16* __restrict__ patches, int P, int kpad, uint8_t* __restrict__ out_u8) {
  // the value table (3 channels x 256 byte values) rides in LDS: every output element is one lookup
  __shared__ float lut[3][256];
  if (out_f32 || patches) {
    lut[0][threadIdx.x] = lut_g[threadIdx.x];
    lut[1][threadIdx.x] = lut_g[256 + threadIdx.x];
    lut[2][threadIdx.x] = lut_g[512 + threadIdx.x];
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int64_t* d = desc + (size_t)b * kDesc;
  const int q = crop_w >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= crop_h * q) return;
  int y, x4;
  if (VEC_PATCH) {  // patch-major: consecutive lanes walk (patch, py, px/4), so a wave's bf16x4 stores of one channel are contiguous
    const int pq = P >> 2, per_patch = P * pq, gwp = crop_w / P;
    const int patch = idx / per_patch, rem = idx - patch * per_patch;
    const int py = rem / pq, gy = patch / gwp;
    y = gy * P + py;
    x4 = (patch - gy * gwp) * pq + (rem - py * pq);
  } else {
    y = idx / q;
    x4 = idx - y * q;
  }
  const int ks = (int)d[11];
  const int32_t* kk = tables + d[9] + (size_t)y * ks;
  const int y0 = tables[d[10] + 2 * y], n = tables[d[10] + 2 * y + 1];
  const int row_dw = crop_w * 3 / 4;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(tmp + d[12]) + (size_t)y0 * row_dw + x4 * 3;
  int acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 1 << (kPrec - 1);
  for (int k0 = 0; k0 < n; k0 += kTapBlockV) {
    int w[kTapBlockV];
    uint32_t v[kTapBlockV][3];
#pragma unroll
    for (int k = 0; k < kTapBlockV; ++k) {  // all loads of the tap block first (a short row range is clamped; its weight is 0)
      const int kc = min(k0 + k, n - 1);
      w[k] = (k0 + k < n) ? kk[kc] : 0;
      const u32x3_a4 t = *reinterpret_cast<const u32x3_a4*>(src + (size_t)kc * row_dw);  // one 12-byte load, dense across the lanes
      v[k][0] = t[0]; v[k][1] = t[1]; v[k][2] = t[2];
    }
#pragma unroll
    for (int k = 0; k < kTapBlockV; ++k)
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] += __mul24((int)((v[k][i >> 2] >> (8 * (i & 3))) & 0xffu), w[k]);
  }
  uint8_t u[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) u[i] = clip8(acc[i]);
  if (out_u8) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out_u8 + ((size_t)b * crop_h + y) * crop_w * 3) + x4 * 3;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o[j] = (uint32_t)u[4 * j] | ((uint32_t)u[4 * j + 1] << 8) | ((uint32_t)u[4 * j + 2] << 16) | ((uint32_t)u[4 * j + 3] << 24);
  }
  if (!out_f32 && !patches) return;
  const int x = x4 * 4;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lut[c][u[3 * j + c]];
    if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + (((size_t)b * 3 + c) * crop_h + y) * crop_w + x) = v;
    if (patches) {
      const int gw = crop_w / P, gh = crop_h / P;
      if (VEC_PATCH) {  // P % 4 == 0: the 4 pixels sit in one patch row
        const size_t row = ((size_t)b * gh + y / P) * gw + x / P;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x4*>(patches + row * kpad + (c * P + y % P) * P + x % P) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const size_t row = ((size_t)b * gh + y / P) * gw + (x + j) / P;
          patches[row * kpad + (c * P + y % P) * P + (x + j) % P] = (bf16)v[j];
        }
      }
    }
  }
}

}  // namespace
}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_image_resample(const int64_t* desc, const int32_t* tables, uint8_t* tmp, int B, int crop_h, int crop_w,
                                    int max_rows, int max_seg_bytes, int max_coef_ints, const float* lut, float* out_f32,
                                    void* patches, int P, int kpad, uint8_t* out_u8, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(desc && tables && tmp && B >= 0 && crop_h > 0 && crop_w > 0 && max_rows > 0, MMAMD_E_BADARG,
                  "image_resample: bad argument");
  MMAMD_CHECK_ARG(out_f32 || patches || out_u8, MMAMD_E_BADARG, "image_resample: no output requested");
  MMAMD_CHECK_ARG(lut || !(out_f32 || patches), MMAMD_E_BADARG, "image_resample: float outputs need the value table");
  MMAMD_CHECK_ARG(B <= 65535 && max_rows <= 65535 && crop_h <= 65535, MMAMD_E_UNSUPPORTED,
                  "image_resample: B=%d / rows=%d / crop_h=%d above the 65535 grid limit", B, max_rows, crop_h);
  if (patches)
    MMAMD_CHECK_ARG(P > 0 && crop_h % P == 0 && crop_w % P == 0 && kpad >= 3 * P * P, MMAMD_E_BADARG,
                    "image_resample: crop %dx%d is not a grid of %d-pixel patches with kpad=%d >= 3*P*P", crop_h, crop_w, P, kpad);
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool quad = crop_w % 4 == 0;  // tmp rows are whole dwords
  // H pass: tiled when a row segment (+ up to 15 bytes of alignment slack, rounded to 16) and the result tile fit the 64 KiB LDS
  const int seg_pitch = (max_seg_bytes + 15 + 15) & ~15;
  const int out_pitch = (crop_w * 3 + 15) & ~15;
  // ints of LDS for the horizontal coefficient table: what the batch needs, up to 16 KiB (224 columns x 18 taps, i.e. down-scales
  // to 4.25x); larger tables are read from global memory by the blocks that have them
  const int coef_cap = max_coef_ints <= 0 ? 0 : (max_coef_ints < 4096 ? (max_coef_ints + 3) & ~3 : 4096);
  int R = 0;
  if (quad && max_seg_bytes > 0)
    for (int r = 8; r >= 1; r >>= 1)
      if ((size_t)r * (seg_pitch + out_pitch) + coef_cap * 4 <= 60 * 1024) { R = r; break; }
  if (R) {
    hipLaunchKernelGGL(resample_h_tiled_kernel, dim3((max_rows + R - 1) / R, B), dim3(256),
                       (size_t)R * (seg_pitch + out_pitch) + coef_cap * 4, st, desc, tables, tmp, crop_w, R, seg_pitch, coef_cap);
  } else {
    hipLaunchKernelGGL(resample_h_kernel, dim3((crop_w * 3 + 255) / 256, max_rows, B), dim3(256), 0, st, desc, tables, tmp, crop_w);
  }
  if (quad) {
    const dim3 grid((crop_h * (crop_w / 4) + 255) / 256, B);
    if (patches && P % 4 == 0 && kpad % 4 == 0)
      hipLaunchKernelGGL((resample_v_tiled_kernel<true>), grid, dim3(256), 0, st, desc, tables, tmp, crop_h, crop_w, lut, out_f32,
                         (bf16*)patches, P, kpad, out_u8);
    else
      hipLaunchKernelGGL((resample_v_tiled_kernel<false>), grid, dim3(256), 0, st, desc, tables, tmp, crop_h, crop_w, lut, out_f32,
                         (bf16*)patches, P, kpad, out_u8);
  } else {
    hipLaunchKernelGGL(resample_v_kernel, dim3((crop_w + 255) / 256, crop_h, B), dim3(256), 0, st, desc, tables, tmp, crop_h, crop_w,
                       lut, out_f32, (bf16*)patches, P, kpad, out_u8);
  }
  return launch_status("image_resample");
}
