// common.h — shared device/host helpers for libmmamd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mmamd.h"
#include "../../include/mmamd_debug.h"

namespace mmamd {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWave = 64;
constexpr int kChipCUs = 256;  // MI355X: 8 XCDs x 32 CUs

// CUs a launch on `st` may occupy: 256, or the popcount of the mask of a stream made by mmamd_stream_create_cu_mask
int stream_cus(hipStream_t st);

// thread-local last-error string, written by the host-side launchers
void set_error(const char* fmt, ...);

#define MMAMD_CHECK_ARG(cond, code, ...)  \
  do {                                    \
    if (!(cond)) {                        \
      ::mmamd::set_error(__VA_ARGS__);    \
      return (code);                      \
    }                                     \
  } while (0)

// per-launcher launch counters (mmamd_debug_launch_count): how an A/B tool checks that its knob took effect without a profiler -- VERDICT r04 hygiene
void count_launch(const char* what);

inline int launch_status(const char* what) {
  count_launch(what);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// One-time opt-in to > 64 KiB of dynamic LDS (160 KiB per CU on gfx950), PER DEVICE: the attribute belongs to the function on the
// current device, so a second GPU in the same process needs its own call (`mask`: one bit per device ordinal).
inline int opt_in_lds(const void* kern, int smem, unsigned long long& mask) {
  if (smem <= 64 * 1024) return 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%d B of dynamic LDS): %s", smem, hipGetErrorString(e)); return (int)e; }
  mask |= bit;
  return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

// load 4 consecutive elements as fp32
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// Philox4x32-10 (Salmon et al., SC'11; Random123): the counter-based generator behind the training-time dropout masks (csrc/dropout.hip,
// the attention-probability dropout of the general attention kernels).  oracle/philox.py restates it and is pinned to the Random123 KATs.
struct Philox4 {
  uint32_t v[4];
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
// keep <=> r >= floor(p * 2^32): P(keep) = 1 - thresh / 2^32 (integer compare, no float rounding in the decision)
inline uint32_t dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}
// Dropout on attention probabilities: element (b, h, q, key) of a [B, H, Sq, Sk] map uses output word key & 3 of the Philox block with
// counter ((b H + h) Sq + q) * ceil(Sk / 4) + key / 4 (64-bit) and (site, 0).  drop_thresh = 0: no dropout.
struct AttnDrop {
  uint32_t thresh, k0, k1, site;
  float scale;
};
__device__ __forceinline__ Philox4 attn_drop_block(const AttnDrop& d, long long row, int sk4, int key) {
  const unsigned long long idx = (unsigned long long)row * (unsigned long long)sk4 + (unsigned long long)(key >> 2);
  return philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), d.site, 0u, d.k0, d.k1);
}

__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
  bf16x4 r;
  r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
  *reinterpret_cast<bf16x4*>(p) = r;
}

}  // namespace mmamd
