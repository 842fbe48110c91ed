// rowops.hip — HBM-bound row kernels of the dual-encoder path (gfx950).
//   layernorm, ViT assemble(+CLS,+pos)+ln_pre, token embedding, patch extraction,
//   pooled-row LN + projection (+L2 normalize), L2 normalize, scalar clamp, dtype convert.
// All are one-wave-per-row (64 lanes, 16-byte vector accesses) unless noted; none reshapes work
// into a GEMM — the roofline that bounds them is HBM bandwidth.
#include <type_traits>

#include "common.h"

namespace mmamd {
extern int g_ln_rev;        // attention.hip (mmamd_debug_set_attn_variant(3110 + r))
extern int g_ln_nt_policy;  // attention.hip (mmamd_debug_set_attn_variant(3100 + p)): 0 = by size, 1 = never, 2 = always
// LayerNorm reads its fp32 input NON-TEMPORALLY when the tensor is larger than what the 256 MiB MALL keeps next to the bf16 output: the input rows then
// stream through without evicting the output rows the next GEMM is about to read.  Measured per step, same-box alternating A/B
// (profiles/r03_cache_policy_ab.txt): ViT-B/16 B = 256 (148 + 38 MiB) 13.66 -> 13.43 ms, ViT-L/14 (257 MiB) -1.0 %, FLAVA (148 MiB) -0.8 %;
// ViT-B/32 (37 MiB: the input is still cached from the GEMM that wrote it) +0.7 %, CoCa B = 128 (128 MiB) +0.3 % with one row per wave and -0.4 % with
// the two-rows-per-wave kernel these inputs now take (layernorm_grouped_rows_kernel) — hence the 128 MiB threshold.
// A stream with a CU budget (mmamd_stream_set_cus) shares the chip -- and the MALL -- with the other half's stream: its tensors count for the
// whole chip's worth (a half-batch of 74 MiB beside another half-batch of 74 MiB streams the same 148 MiB past the MALL).
static bool ln_nontemporal(size_t x_bytes, hipStream_t st) {
  if (g_ln_nt_policy != 0) return g_ln_nt_policy == 2;
  return x_bytes * (size_t)kChipCUs / (size_t)stream_cus(st) >= ((size_t)128 << 20);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row cached in registers (MAXV float4 per lane), two-pass statistics
// ---------------------------------------------------------------------------------------------
template <typename TIN, typename TOUT, int MAXV, int NT = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        TOUT* __restrict__ y, int rows, int d,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d4 = d >> 2;
  const TIN* xr = x + (size_t)row * d;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      if constexpr (NT != 0 && std::is_same<TIN, float>::value) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4 * c));
      else v[i] = load4(xr + 4 * c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = v[i][j] - mean;
        q += t * t;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  TOUT* yr = y + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      const f32x4 g = load4(gamma + 4 * c), b = load4(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      store4(yr + 4 * c, o);
    }
  }
}

template <typename TIN, typename TOUT>
static int launch_layernorm(const void* x, const float* g, const float* b, void* y, int rows, int d,
                            float eps, hipStream_t st) {
  const int d4 = d / 4;
  const dim3 grid((rows + 3) / 4), block(256);
  if (std::is_same<TIN, float>::value && ln_nontemporal((size_t)rows * d * 4, st)) {
    if (d4 <= 128)
      hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, 2, 1>), grid, block, 0, st, (const TIN*)x, g, b, (TOUT*)y, rows, d, eps);
    else if (d4 <= 256)
      hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, 4, 1>), grid, block, 0, st, (const TIN*)x, g, b, (TOUT*)y, rows, d, eps);
    else
      hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, 8, 1>), grid, block, 0, st, (const TIN*)x, g, b, (TOUT*)y, rows, d, eps);
    return launch_status("layernorm");
  }
  if (d4 <= 128)
    hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, 2>), grid, block, 0, st, (const TIN*)x, g, b, (TOUT*)y, rows, d, eps);
  else if (d4 <= 256)
    hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, 4>), grid, block, 0, st, (const TIN*)x, g, b, (TOUT*)y, rows, d, eps);
  else
    hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, 8>), grid, block, 0, st, (const TIN*)x, g, b, (TOUT*)y, rows, d, eps);
  return launch_status("layernorm");
}

// ---------------------------------------------------------------------------------------------
// Grouped LayerNorm with an optional residual add in front: per problem  x += delta (bf16, when given; x fp32 in place), y = LN(x) (bf16).
// One wave per row, both problems (the two towers of a dual encoder: different row counts and widths) in ONE launch — the text tower's
// 14 us small-grid launches of r02 ride along with the ViT rows, and with `delta` the fp32 read-modify-write of the residual stream moves
// out of the out-projection / MLP-down GEMM epilogues (which then store a bf16 tile) into this streaming kernel (r02 VERDICT item 2).
struct LnProb {
  float* x;
  const bf16* delta;
  const float* gamma;
  const float* beta;
  bf16* y;
  int rows, d;
  float eps;
  int pad_;
};
struct LnGroupArgs {
  LnProb p[2];
  int nprob;
  int blocks0;  // blocks (4 rows each) of problem 0
  int rev;      // the blocks walk the rows from the last to the first
};

template <int MAXV, bool HAS_DELTA, int NT = 0>
__global__ __launch_bounds__(256) void add_layernorm_grouped_kernel(const LnGroupArgs a) {
  const int lane = threadIdx.x & 63;
  int blk = blockIdx.x, pi = 0;
  if (blk >= a.blocks0) { blk -= a.blocks0; pi = 1; }
  float* __restrict__ x = a.p[pi].x;
  const bf16* __restrict__ delta = a.p[pi].delta;
  const float* __restrict__ gamma = a.p[pi].gamma;
  const float* __restrict__ beta = a.p[pi].beta;
  bf16* __restrict__ y = a.p[pi].y;
  const int rows = a.p[pi].rows, d = a.p[pi].d;
  const float eps = a.p[pi].eps;
  const int row = blk * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d4 = d >> 2;
  float* xr = x + (size_t)row * d;
  f32x4 v[MAXV];
  float s = 0.f;
  if (HAS_DELTA && delta != nullptr) {  // wave-uniform
    const bf16* dr = delta + (size_t)row * d;
    f32x4 dv[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {  // all loads of the row in flight before the first use
      const int c = lane + 64 * i;
      if (c < d4) {
        v[i] = load4(xr + 4 * c);
        dv[i] = load4(dr + 4 * c);
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] += dv[i][j];
      if (c < d4) store4(xr + 4 * c, v[i]);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        if constexpr (NT != 0) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4 * c));
        else v[i] = load4(xr + 4 * c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  if (y == nullptr) return;  // add only (the last layer's MLP-down has no LayerNorm behind it)
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = v[i][j] - mean;
        q += t * t;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  bf16* yr = y + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      const f32x4 g = load4(gamma + 4 * c), b = load4(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      store4(yr + 4 * c, o);
    }
  }
}

// The grouped LayerNorm for LARGE inputs (non-temporal loads, see ln_nontemporal): RPW rows per wave, the loads of all of them in flight before
// the first reduction (twice the bytes in flight per wave of the one-row kernel).  Same arithmetic per row: bit-identical results.  Same-box
// alternating A/B of the headline step (profiles/r03_cache_policy_ab.txt): 2 rows per wave -0.05 ... -0.12 ms, 3 and 4 rows no better than 1.
template <int MAXV, int RPW>
__global__ __launch_bounds__(256) void layernorm_grouped_rows_kernel(const LnGroupArgs a) {
  const int lane = threadIdx.x & 63;
  int blk = a.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x, pi = 0;
  const int b0 = (a.p[0].rows + 4 * RPW - 1) / (4 * RPW);
  if (blk >= b0) { blk -= b0; pi = 1; }
  const float* __restrict__ x = a.p[pi].x;
  const float* __restrict__ gamma = a.p[pi].gamma;
  const float* __restrict__ beta = a.p[pi].beta;
  bf16* __restrict__ y = a.p[pi].y;
  const int rows = a.p[pi].rows, d = a.p[pi].d;
  const float eps = a.p[pi].eps;
  const int row0 = blk * (4 * RPW) + (threadIdx.x >> 6) * RPW;
  if (row0 >= rows) return;
  const int d4 = d >> 2;
  f32x4 v[RPW][MAXV];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      v[r][i] = (row0 + r < rows && c < d4) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (size_t)(row0 + r) * d + 4 * c)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    if (row0 + r >= rows) break;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + 64 * i < d4) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + 64 * i < d4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = v[r][i][j] - mean;
          q += t * t;
        }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    bf16* yr = y + (size_t)(row0 + r) * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        const f32x4 g = load4(gamma + 4 * c), b = load4(beta + 4 * c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[r][i][j] - mean) * rstd * g[j] + b[j];
        store4(yr + 4 * c, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ViT assemble: x[b,0]=cls+pos[0]; x[b,1+i]=patch_emb[b,i]+pos[1+i]; then ln_pre  -> fp32
// ---------------------------------------------------------------------------------------------
template <typename TPE, int MAXV>
__global__ __launch_bounds__(256) void vit_assemble_ln_kernel(const TPE* __restrict__ pe,
                                                              const float* __restrict__ cls,
                                                              const float* __restrict__ pos,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float eps, float* __restrict__ x,
                                                              int B, int G2, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int S = G2 + 1;
  if (row >= B * S) return;
  const int b = row / S, s = row - b * S;
  const int d4 = d >> 2;
  f32x4 v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      f32x4 t = (s == 0) ? load4(cls + 4 * c) : load4(pe + ((size_t)b * G2 + (s - 1)) * d + 4 * c);
      const f32x4 p = load4(pos + (size_t)s * d + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] += p[j];
      v[i] = t;
      sum += (t[0] + t[1]) + (t[2] + t[3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum(sum) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = v[i][j] - mean;
        q += t * t;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  float* xr = x + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      const f32x4 g = load4(gamma + 4 * c), bb = load4(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
      store4(xr + 4 * c, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ViT stem behind the fused patch-embedding GEMM (mmamd_patch_embed_gemm wrote x[b,1+i] = conv(patch i) + pos[1+i] in place):
// x[b,0] = cls + pos[0]; x = ln_pre(x) (fp32, in place) and, when asked, hn = LayerNorm(x; gamma1, beta1) (bf16) = norm1 of the first
// encoder layer in the same pass — the row never leaves the registers between the two normalisations.  Same arithmetic, same
// summation order as vit_assemble_ln_kernel / layernorm_kernel.
// ---------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void vit_cls_lnpre_ln_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos0,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               const float* __restrict__ gamma1, const float* __restrict__ beta1, float eps1,
                                                               bf16* __restrict__ hn, int rows, int S, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int s = row % S;
  const int d4 = d >> 2;
  float* xr = x + (size_t)row * d;
  f32x4 v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      f32x4 t;
      if (s == 0) {
        t = load4(cls + 4 * c);
        const f32x4 p = load4(pos0 + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] += p[j];
      } else {
        t = load4(xr + 4 * c);
      }
      v[i] = t;
      sum += (t[0] + t[1]) + (t[2] + t[3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  auto normalise = [&](const float* g_, const float* b_, float e_) {  // v = LN(v) (rows beyond d4 stay zero)
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(sm) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = v[i][j] - mean;
          q += t * t;
        }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + e_);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        const f32x4 g = load4(g_ + 4 * c), bb = load4(b_ + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
      }
    }
  };
  (void)sum;
  normalise(gamma, beta, eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) store4(xr + 4 * c, v[i]);
  }
  if (hn == nullptr) return;
  normalise(gamma1, beta1, eps1);
  bf16* hr = hn + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) store4(hr + 4 * c, v[i]);
  }
}

// ---------------------------------------------------------------------------------------------
// token embedding gather + positional embedding  -> fp32
// ---------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ ids,
                                                           const TT* __restrict__ table,
                                                           const float* __restrict__ pos,
                                                           float* __restrict__ x, int rows, int S,
                                                           int d, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int s = row % S;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const TT* tr = table + (size_t)id * d;
  const float* pr = pos + (size_t)s * d;
  float* xr = x + (size_t)row * d;
  for (int c = lane; c < (d >> 2); c += 64) {
    f32x4 t = load4(tr + 4 * c);
    const f32x4 p = load4(pr + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] += p[j];
    store4(xr + 4 * c, t);
  }
}

// ---------------------------------------------------------------------------------------------
// patch extraction: out[(b*G+gy)*G+gx, k=(c*P+py)*P+px] = img[b,c,gy*P+py,gx*P+px]; zero pad k>=C*P*P
// one thread per 4 consecutive k
// ---------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void patchify_kernel(const TI* __restrict__ img,
                                                       bf16* __restrict__ out, int B, int C, int HW,
                                                       int P, int Kpad) {
  const int G = HW / P;
  const int K = C * P * P;
  const int kq = Kpad >> 2;  // groups of 4 per row
  const size_t total = (size_t)B * G * G * kq;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (size_t)gridDim.x * blockDim.x) {
    const int k0 = (int)(t % kq) * 4;
    const size_t prow = t / kq;
    const int gx = (int)(prow % G);
    const int gy = (int)((prow / G) % G);
    const int b = (int)(prow / ((size_t)G * G));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((P & 3) == 0 && k0 + 3 < K) {
      const int c = k0 / (P * P), r = k0 - c * P * P, py = r / P, px = r - py * P;
      v = load4(img + (((size_t)b * C + c) * HW + (gy * P + py)) * HW + gx * P + px);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (k < K) {
          const int c = k / (P * P), r = k - c * P * P, py = r / P, px = r - py * P;
          v[j] = to_f32(img[(((size_t)b * C + c) * HW + (gy * P + py)) * HW + gx * P + px]);
        }
      }
    }
    store4(out + prow * Kpad + k0, v);
  }
}

// ---------------------------------------------------------------------------------------------
// pooled row -> LN -> projection (-> L2 normalize)
//   pool_ln_kernel : one wave per sample: EOT index = first argmax of the ids (or 0), LayerNorm of that row -> h[B,d] fp32
//   proj_f32_kernel: out[B,E] = h[B,d] . P on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), one wave per 32x32 tile;
//                    P element (k,e) at proj[k*sk + e*se]: [d,E] parameter (sk=E,se=1) or Linear weight [E,d] (sk=1,se=d)
//   (the first version did the projection as a block-per-sample mat-vec that re-read the whole matrix per sample:
//    276 us per call in the r01 rocprof trace)
// ---------------------------------------------------------------------------------------------
// (r02: the first pool_ln_kernel walked the row three times with dependent scalar loads, 82 us for 256 rows; the first proj_f32_kernel
//  gave one wave a whole 32 x 32 x d tile with a load -> 4 MFMA -> load chain, 106-152 us for 0.2 GFLOP.  Now the row is read once
//  into registers, and the contraction is split over the 4 waves of a block with 32 k's of loads in flight per wave.)
template <int MAXV>
__global__ __launch_bounds__(256) void pool_ln_kernel(const float* __restrict__ x, int S, int d,
                                                      const int64_t* __restrict__ ids, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float* __restrict__ h, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int best_i = 0;
  if (ids != nullptr) {
    long long best_v = INT64_MIN;
    best_i = 0x7fffffff;
    for (int s = lane; s < S; s += 64) {
      const long long v = ids[(size_t)b * S + s];
      if (v > best_v) { best_v = v; best_i = s; }  // ascending s: keeps the first maximum
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const long long ov = __shfl_xor(best_v, off);
      const int oi = __shfl_xor(best_i, off);
      if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
    }
  }
  const float* xr = x + ((size_t)b * S + best_i) * d;
  float* hr = h + (size_t)b * d;
  if constexpr (MAXV > 0) {  // d % 4 == 0, d <= 256 * MAXV: the row lives in registers (one batch of 16-byte loads)
    const int d4 = d >> 2;
    f32x4 v[MAXV];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < d4 ? load4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s1) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < d4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float t = v[i][j] - mean; s2 += t * t; }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        const f32x4 g = load4(gamma + 4 * c), bb = load4(beta + 4 * c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
        store4(hr + 4 * c, o);
      }
    }
  } else {
    float s1 = 0.f;
    for (int k = lane; k < d; k += 64) s1 += xr[k];
    const float mean = wave_sum(s1) / (float)d;
    float s2 = 0.f;
    for (int k = lane; k < d; k += 64) { const float t = xr[k] - mean; s2 += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)d + eps);
    for (int k = lane; k < d; k += 64) hr[k] = (xr[k] - mean) * rstd * gamma[k] + beta[k];
  }
}

// out[B,E] = h[B,d] . P in exact fp32 (v_mfma_f32_32x32x2_f32).  One block per 32 x 32 output tile; its 4 waves split the contraction
// (wave w: k in [w kq, (w+1) kq), kq a multiple of 32) and sum their partial tiles through LDS in wave order (deterministic).
__global__ __launch_bounds__(256) void proj_f32_kernel(const float* __restrict__ h, const float* __restrict__ proj, int sk,
                                                       int se, float* __restrict__ out, int B, int d, int E) {
  __shared__ float red[4][32][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int half = lane >> 5;
  int ri = i0 + (lane & 31); ri = ri < B ? ri : B - 1;
  int rj = j0 + (lane & 31); rj = rj < E ? rj : E - 1;
  const float* lp = h + (size_t)ri * d;
  const float* rp = proj + (size_t)rj * se;
  const bool vec = (sk == 1) && ((d & 3) == 0) && ((se & 3) == 0) && ((reinterpret_cast<uintptr_t>(proj) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(h) & 15) == 0);
  const bool xvec = ((d & 3) == 0) && ((reinterpret_cast<uintptr_t>(h) & 15) == 0);
  const int kq = ((d + 127) / 128) * 32;  // per-wave share of the contraction, multiple of 32
  const int kbeg = wv * kq, kend = (kbeg + kq) < d ? (kbeg + kq) : d;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    f32x4 xv[4], yv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // all loads of 32 k's first: one memory round trip per 16 MFMAs instead of one per 4
      const int k = k0 + 8 * q + 4 * half;
      xv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      yv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (xvec && k + 3 < kend) {
        xv[q] = load4(lp + k);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k + u < kend) xv[q][u] = lp[k + u];
      }
      if (vec && k + 3 < kend) {
        yv[q] = load4(rp + k);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k + u < kend) yv[q][u] = rp[(size_t)(k + u) * sk];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[q][u], yv[q][u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wv][(r & 3) + 8 * (r >> 2) + 4 * half][lane & 31] = acc[r];
  __syncthreads();
  for (int t = threadIdx.x; t < 32 * 32; t += 256) {
    const int i = t >> 5, j = t & 31;
    if (i0 + i < B && j0 + j < E) out[(size_t)(i0 + i) * E + j0 + j] = ((red[0][i][j] + red[1][i][j]) + red[2][i][j]) + red[3][i][j];
  }
}

__global__ __launch_bounds__(256) void l2_normalize_inplace_f32_kernel(float* __restrict__ x, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = x + (size_t)row * d;
  float ss = 0.f;
  for (int k = lane; k < d; k += 64) { const float v = xr[k]; ss += v * v; }
  const float sc = 1.0f / fmaxf(sqrtf(wave_sum(ss)), eps);
  for (int k = lane; k < d; k += 64) xr[k] *= sc;
}


// ---------------------------------------------------------------------------------------------
// BERT embeddings: word[id] + position[pos] + token_type[type], then LayerNorm  -> fp32   (wave per token)
// (modules/layers/text_embedding.py:74-104; position ids default to 0..S-1, token types to 0)
// ---------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids,
                                                            const int64_t* __restrict__ pos_ids, const float* __restrict__ word,
                                                            const float* __restrict__ pos, const float* __restrict__ type,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ x, int rows, int S, int d, int vocab,
                                                            int max_pos, int n_types) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  long long pi = pos_ids ? pos_ids[row] : (row % S);
  pi = pi < 0 ? 0 : (pi >= max_pos ? max_pos - 1 : pi);
  long long ti = type_ids ? type_ids[row] : 0;
  ti = ti < 0 ? 0 : (ti >= n_types ? n_types - 1 : ti);
  const int d4 = d >> 2;
  f32x4 v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < d4) {
      const f32x4 a = load4(word + (size_t)id * d + 4 * c), b = load4(pos + (size_t)pi * d + 4 * c),
                  t = load4(type + (size_t)ti * d + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = (a[j] + b[j]) + t[j];
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  if (gamma == nullptr) {  // training forward keeps the pre-LayerNorm sum (its backward needs it): no normalisation here
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) store4(x + (size_t)row * d + 4 * c, v[i]);
    }
    return;
  }
  const float mean = wave_sum(sum) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float u = v[i][j] - mean; q += u * u; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      const f32x4 g = load4(gamma + 4 * c), bb = load4(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
      store4(x + (size_t)row * d + 4 * c, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// FLAVA image embeddings: patch embeddings (optionally blended with the mask token), CLS, + position  -> fp32, no LN
// (models/flava/image_encoder.py:139-177)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flava_image_embed_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                                                const float* __restrict__ pos, const int64_t* __restrict__ pmask,
                                                                const float* __restrict__ mask_token, float* __restrict__ x,
                                                                int B, int G2, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int hc = cls != nullptr ? 1 : 0;  // no CLS row: CoCa's ViT (layers/patch_embedding.py, include_cls_embed=False)
  const int S = G2 + hc;
  if (row >= B * S) return;
  const int b = row / S, s = row - b * S;
  const int pi = s - hc;
  float w = 0.f;
  if (pi >= 0 && pmask != nullptr && mask_token != nullptr) w = (float)pmask[(size_t)b * G2 + pi];
  for (int c = lane; c < (d >> 2); c += 64) {
    f32x4 t;
    if (pi < 0) {
      t = load4(cls + 4 * c);
    } else {
      t = load4(pe + ((size_t)b * G2 + pi) * d + 4 * c);
      if (w != 0.f) {
        const f32x4 mt = load4(mask_token + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = t[j] * (1.f - w) + mt[j] * w;
      }
    }
    const f32x4 pp = load4(pos + (size_t)s * d + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] += pp[j];
    store4(x + (size_t)row * d + 4 * c, t);
  }
}

// out[B,E] = act(rows . W^T + bias): rows i at h + i*ldh (e.g. the CLS row of every sample), W [E,d] fp32 (Linear weight),
// fp32 MFMA; act 0 = none, 1 = tanh  (Pooler: modules/losses/flava.py:84-97; image/text projections: models/flava/model.py:243-264)
__global__ __launch_bounds__(256) void rows_linear_f32_kernel(const float* __restrict__ h, size_t ldh, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int act, float* __restrict__ out,
                                                              int B, int d, int E) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32;
  const int j0 = (blockIdx.x * 4 + wv) * 32;
  if (j0 >= E) return;
  const int half = lane >> 5;
  int ri = i0 + (lane & 31); ri = ri < B ? ri : B - 1;
  int rj = j0 + (lane & 31); rj = rj < E ? rj : E - 1;
  const float* lp = h + (size_t)ri * ldh;
  const float* rp = W + (size_t)rj * d;
  const bool vec = ((d & 3) == 0) && ((ldh & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(h) & 15) == 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // four k-steps (32 columns) per trip, all eight loads issued before the first MFMA: the loop was one dependent load -> MFMA chain per 8 columns
  // (41 us for a 256 x 768 x 768 pooler: latency, not arithmetic; r05).  Same products in the same order: bit-identical results.
  for (int k0 = 0; k0 < d; k0 += 32) {
    f32x4 xv[4], yv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = k0 + 8 * s + 4 * half;
      xv[s] = f32x4{0.f, 0.f, 0.f, 0.f}; yv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (vec && k + 3 < d) {
        xv[s] = load4(lp + k);
        yv[s] = load4(rp + k);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k + u < d) { xv[s][u] = lp[k + u]; yv[s][u] = rp[k + u]; }
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s][u], yv[s][u], acc, 0, 0, 0);
  }
  const int j = j0 + (lane & 31);
  if (j < E) {
    const float bj = bias ? bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < B) {
        float v = acc[r] + bj;
        if (act == 1) v = tanhf(v);
        else if (act == 2) v = fmaxf(v, 0.f);
        out[(size_t)i * E + j] = v;
      }
    }
  }
}


// key-padding mask as the attention kernel wants it (uint8, 0 = masked key) from ids (!= pad) or from a 0/1 mask of any of
// the dtypes callers hold (modules/encoders/bert_text_encoder.py:86-91; utils/attention.py:13-52 keeps "0 = ignore")
__global__ __launch_bounds__(256) void key_mask_kernel(const void* __restrict__ src, int kind, long long pad, uint8_t* __restrict__ out,
                                                       long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool keep;
  if (kind == 0) keep = reinterpret_cast<const int64_t*>(src)[i] != pad;
  else if (kind == 1) keep = reinterpret_cast<const float*>(src)[i] != 0.f;
  else if (kind == 2) keep = reinterpret_cast<const int64_t*>(src)[i] != 0;
  else keep = reinterpret_cast<const uint8_t*>(src)[i] != 0;
  out[i] = keep ? 1 : 0;
}


// CoCa text embeddings: x[b, s] = token[ids[b, s]] + pos[s] for s < S_ids, x[b, S_ids] = cls + pos[S_ids] (cls optional)
// (models/coca/text_decoder.py:67-88)
__global__ __launch_bounds__(256) void coca_text_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                              const float* __restrict__ pos, const float* __restrict__ cls,
                                                              float* __restrict__ x, int B, int S_ids, int d, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int S = S_ids + (cls != nullptr ? 1 : 0);
  if (row >= B * S) return;
  const int b = row / S, s = row - b * S;
  const float* src;
  if (s < S_ids) {
    long long id = ids[(size_t)b * S_ids + s];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    src = table + (size_t)id * d;
  } else {
    src = cls;
  }
  for (int c = lane; c < (d >> 2); c += 64) {
    const f32x4 a = load4(src + 4 * c), pp = load4(pos + (size_t)s * d + 4 * c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a[j] + pp[j];
    store4(x + (size_t)row * d + 4 * c, o);
  }
}

// CoCaTextDecoder.build_mask (models/coca/text_decoder.py:178-194) as a uint8 [B, S+1, S+1] attend-mask: causal everywhere;
// the CLS query (last row) additionally sees key 0 always and key j >= 1 only if token j-1 is not padding (the reference's
// F.pad(..., (1, 0, S, 0)) shifts the padding mask by one column).  kind as in key_mask (0: ids != pad, 1/2/3: mask != 0).
__global__ __launch_bounds__(256) void coca_text_mask_kernel(const void* __restrict__ src, int kind, long long pad, uint8_t* __restrict__ out,
                                                             int B, int S) {
  const int T = S + 1;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * T * T) return;
  const int b = (int)(i / (T * T));
  const int rem = (int)(i - (long long)b * T * T);
  const int qi = rem / T, kj = rem - qi * T;
  bool keep = kj <= qi;
  if (keep && qi == S && kj >= 1) {
    const size_t o = (size_t)b * S + (kj - 1);
    if (kind == 0) keep = reinterpret_cast<const int64_t*>(src)[o] != pad;
    else if (kind == 1) keep = reinterpret_cast<const float*>(src)[o] != 0.f;
    else if (kind == 2) keep = reinterpret_cast<const int64_t*>(src)[o] != 0;
    else keep = reinterpret_cast<const uint8_t*>(src)[o] != 0;
  }
  out[i] = keep ? 1 : 0;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void l2_normalize_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                           int rows, int d, float eps, int ldy) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + (size_t)row * d;
  float ss = 0.f;
  for (int k = lane; k < d; k += 64) { const float v = to_f32(xr[k]); ss += v * v; }
  const float sc = 1.0f / fmaxf(sqrtf(wave_sum(ss)), eps);
  TO* yr = y + (size_t)row * ldy;  // ldy > d: the rows land in one half of the packed [B, 2E] gather block
  for (int k = lane; k < d; k += 64) yr[k] = (TO)(to_f32(xr[k]) * sc);
}

__global__ void clamp_scalar_kernel(float* p, int has_min, float lo, int has_max, float hi) {
  float v = *p;
  if (has_min) v = fmaxf(v, lo);
  if (has_max) v = fminf(v, hi);
  *p = v;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_kernel(const TI* __restrict__ s, TO* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = (TO)to_f32(s[i]);
}

// ---------------------------------------------------------------------------------------------
// FLAVA position-embedding interpolation (models/flava/image_encoder.py:102-137): the [n_side x n_side] grid of patch position
// embeddings resampled to [h0 x w0] with F.interpolate(mode="bicubic", align_corners=False, scale_factor=(sh, sw)); row 0 (CLS) is
// copied.  The resampling itself is PyTorch's upsample_bicubic2d (aten/src/ATen/native/UpSampleBicubic2d.cpp, torch 2.10 — not part
// of the reference tree): source coordinate (o + 0.5) / scale - 0.5, cubic-convolution weights with A = -0.75, taps clamped to the
// grid, x pass then y pass in fp32.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
__global__ __launch_bounds__(256) void bicubic_pos_embed_kernel(const float* __restrict__ pos, int n_side, int d, float* __restrict__ out,
                                                                int h0, int w0, float inv_sh, float inv_sw) {
  const int row = blockIdx.x;  // output row: 0 = CLS, 1 + oy*w0 + ox
  if (row == 0) {
    for (int c = threadIdx.x; c < d; c += 256) out[c] = pos[c];
    return;
  }
  const int oy = (row - 1) / w0, ox = (row - 1) - oy * w0;
  const float ry = (oy + 0.5f) * inv_sh - 0.5f, rx = (ox + 0.5f) * inv_sw - 0.5f;
  const float fy = floorf(ry), fx = floorf(rx);
  const int iy = (int)fy, ix = (int)fx;
  float wy[4], wx[4];
  cubic_weights(ry - fy, wy);
  cubic_weights(rx - fx, wx);
  for (int c = threadIdx.x; c < d; c += 256) {
    float col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int y = iy - 1 + i;
      y = y < 0 ? 0 : (y > n_side - 1 ? n_side - 1 : y);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int x = ix - 1 + j;
        x = x < 0 ? 0 : (x > n_side - 1 ? n_side - 1 : x);
        acc += pos[(size_t)(1 + y * n_side + x) * d + c] * wx[j];
      }
      col[i] = acc;
    }
    out[(size_t)row * d + c] = col[0] * wy[0] + col[1] * wy[1] + col[2] * wy[2] + col[3] * wy[3];
  }
}

// RoBERTa-style position ids (BERTTextEmbeddings.create_position_ids_from_input_ids, modules/layers/text_embedding.py:55-68): non-padding
// tokens are numbered 1, 2, ... from the left, + pad_id; padding tokens get pad_id.  One thread per row (S <= 512: a serial scan).
__global__ __launch_bounds__(64) void offset_position_ids_kernel(const long long* __restrict__ ids, long long pad_id, long long* __restrict__ out, int B, int S) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  long long run = 0;
  for (int s_ = 0; s_ < S; ++s_) {
    const bool tok = ids[(size_t)b * S + s_] != pad_id;
    run += tok ? 1 : 0;
    out[(size_t)b * S + s_] = (tok ? run : 0) + pad_id;
  }
}

// labels[i] = keep[i] ? labels[i] : fill   (FLAVAForPreTraining: image_labels[~image_patches_mask] = -1, models/flava/model.py:340-343)
__global__ __launch_bounds__(256) void mask_labels_kernel(long long* __restrict__ labels, const uint8_t* __restrict__ keep, long long fill, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n && keep[i] == 0) labels[i] = fill;
}

}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_add_layernorm_grouped(const mmamd_ln_problem* probs, int nprob, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(probs != nullptr && nprob >= 1 && nprob <= 2, MMAMD_E_BADARG, "add_layernorm_grouped: 1 or 2 problems");
  LnGroupArgs a;
  a.nprob = 0;
  a.blocks0 = 0;
  a.rev = g_ln_rev;
  int total = 0, dmax = 0;
  for (int i = 0; i < nprob; ++i) {
    const mmamd_ln_problem& q = probs[i];
    MMAMD_CHECK_ARG(q.x && q.rows >= 0 && q.d > 0 && (q.y == nullptr || (q.gamma && q.beta)), MMAMD_E_BADARG, "add_layernorm_grouped: bad argument (problem %d)", i);
    MMAMD_CHECK_ARG(q.y != nullptr || q.delta != nullptr, MMAMD_E_BADARG, "add_layernorm_grouped: problem %d has neither an output nor a delta", i);
    MMAMD_CHECK_ARG(q.d % 4 == 0 && q.d <= 2048, MMAMD_E_UNSUPPORTED, "add_layernorm_grouped: d=%d must be a multiple of 4 and <= 2048", q.d);
    MMAMD_CHECK_ARG(aligned16(q.x) && aligned16(q.delta) && aligned16(q.gamma) && aligned16(q.beta) && aligned16(q.y) && (q.d * 2) % 16 == 0,
                    MMAMD_E_ALIGN, "add_layernorm_grouped: rows must be 16-byte aligned");
    if (q.rows == 0) continue;
    LnProb& p = a.p[a.nprob];
    p.x = q.x; p.delta = (const bf16*)q.delta; p.gamma = q.gamma; p.beta = q.beta; p.y = (bf16*)q.y; p.rows = q.rows; p.d = q.d; p.eps = q.eps; p.pad_ = 0;
    const int blocks = (q.rows + 3) / 4;
    if (a.nprob == 0) a.blocks0 = blocks;
    total += blocks;
    if (q.d > dmax) dmax = q.d;
    ++a.nprob;
  }
  if (a.nprob == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  bool has_delta = false;
  for (int i = 0; i < a.nprob; ++i) has_delta = has_delta || a.p[i].delta != nullptr;
  size_t xbytes = 0;
  for (int i = 0; i < a.nprob; ++i) xbytes += (size_t)a.p[i].rows * a.p[i].d * 4;
  const bool ntp = ln_nontemporal(xbytes, st);
  bool all_y = true;
  for (int i = 0; i < a.nprob; ++i) all_y = all_y && a.p[i].y != nullptr;
  if (ntp && !has_delta && all_y) {  // large inputs: two rows per wave
    int tot2 = 0;
    for (int i = 0; i < a.nprob; ++i) tot2 += (a.p[i].rows + 7) / 8;
    if (dmax <= 512) hipLaunchKernelGGL((layernorm_grouped_rows_kernel<2, 2>), dim3(tot2), dim3(256), 0, st, a);
    else if (dmax <= 768) hipLaunchKernelGGL((layernorm_grouped_rows_kernel<3, 2>), dim3(tot2), dim3(256), 0, st, a);
    else if (dmax <= 1024) hipLaunchKernelGGL((layernorm_grouped_rows_kernel<4, 2>), dim3(tot2), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((layernorm_grouped_rows_kernel<8, 2>), dim3(tot2), dim3(256), 0, st, a);
    return launch_status("add_layernorm_grouped");
  }
#define LN_LAUNCH(MV)                                                                                                  \
  do {                                                                                                                 \
    if (has_delta) hipLaunchKernelGGL((add_layernorm_grouped_kernel<MV, true>), dim3(total), dim3(256), 0, st, a);     \
    else if (ntp) hipLaunchKernelGGL((add_layernorm_grouped_kernel<MV, false, 1>), dim3(total), dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((add_layernorm_grouped_kernel<MV, false>), dim3(total), dim3(256), 0, st, a);              \
  } while (0)
  if (dmax <= 512) LN_LAUNCH(2);
  else if (dmax <= 768) LN_LAUNCH(3);
  else if (dmax <= 1024) LN_LAUNCH(4);
  else LN_LAUNCH(8);
#undef LN_LAUNCH
  return launch_status("add_layernorm_grouped");
}

extern "C" int mmamd_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta,
                               void* y, int y_dtype, int rows, int d, float eps, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && gamma && beta && y && rows >= 0 && d > 0, MMAMD_E_BADARG, "layernorm: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0 && d <= 8192, MMAMD_E_UNSUPPORTED, "layernorm: d=%d must be a multiple of 4 and <= 8192", d);
  MMAMD_CHECK_ARG(aligned16(x) && aligned16(gamma) && aligned16(beta) && (y_dtype == MMAMD_F32 ? aligned16(y) : ((uintptr_t)y & 7) == 0),
                  MMAMD_E_ALIGN, "layernorm: pointers must be 16-byte aligned");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == MMAMD_F32 && y_dtype == MMAMD_BF16) {
    if (d <= 2048 && (d * 2) % 16 == 0 && ln_nontemporal((size_t)rows * d * 4, st)) {  // large input: the two-rows-per-wave kernel (same arithmetic)
      mmamd_ln_problem q;
      q.x = const_cast<float*>(reinterpret_cast<const float*>(x)); q.delta = nullptr; q.gamma = gamma; q.beta = beta; q.y = y; q.rows = rows; q.d = d; q.eps = eps;
      return mmamd_add_layernorm_grouped(&q, 1, stream);
    }
    return launch_layernorm<float, bf16>(x, gamma, beta, y, rows, d, eps, st);
  }
  if (x_dtype == MMAMD_F32 && y_dtype == MMAMD_F32) return launch_layernorm<float, float>(x, gamma, beta, y, rows, d, eps, st);
  if (x_dtype == MMAMD_BF16 && y_dtype == MMAMD_BF16) return launch_layernorm<bf16, bf16>(x, gamma, beta, y, rows, d, eps, st);
  if (x_dtype == MMAMD_BF16 && y_dtype == MMAMD_F32) return launch_layernorm<bf16, float>(x, gamma, beta, y, rows, d, eps, st);
  MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "layernorm: bad dtype code");
}

extern "C" int mmamd_vit_assemble_ln(const void* pe, int pe_dtype, const float* cls, const float* pos,
                                     const float* gamma, const float* beta, float eps, float* x, int B,
                                     int G2, int d, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(pe && cls && pos && gamma && beta && x && B >= 0 && G2 > 0 && d > 0, MMAMD_E_BADARG, "vit_assemble_ln: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0 && d <= 2048, MMAMD_E_UNSUPPORTED, "vit_assemble_ln: d=%d must be a multiple of 4 and <= 2048", d);
  if (B == 0) return 0;
  const int rows = B * (G2 + 1);
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int d4 = d / 4;
#define LAUNCH_ASM(T, MV) hipLaunchKernelGGL((vit_assemble_ln_kernel<T, MV>), grid, block, 0, st, (const T*)pe, cls, pos, gamma, beta, eps, x, B, G2, d)
  if (pe_dtype == MMAMD_BF16) {
    if (d4 <= 128) LAUNCH_ASM(bf16, 2); else if (d4 <= 256) LAUNCH_ASM(bf16, 4); else LAUNCH_ASM(bf16, 8);
  } else if (pe_dtype == MMAMD_F32) {
    if (d4 <= 128) LAUNCH_ASM(float, 2); else if (d4 <= 256) LAUNCH_ASM(float, 4); else LAUNCH_ASM(float, 8);
  } else {
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "vit_assemble_ln: bad dtype code");
  }
#undef LAUNCH_ASM
  return launch_status("vit_assemble_ln");
}

extern "C" int mmamd_vit_cls_lnpre_ln(float* x, const float* cls, const float* pos0, const float* gamma, const float* beta, float eps,
                                      const float* gamma1, const float* beta1, float eps1, void* hn, int B, int S, int d, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && cls && pos0 && gamma && beta && B >= 0 && S > 0 && d > 0 && (hn == nullptr || (gamma1 && beta1)), MMAMD_E_BADARG,
                  "vit_cls_lnpre_ln: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0 && d <= 2048, MMAMD_E_UNSUPPORTED, "vit_cls_lnpre_ln: d=%d must be a multiple of 4 and <= 2048", d);
  MMAMD_CHECK_ARG(aligned16(x) && aligned16(cls) && aligned16(pos0) && aligned16(gamma) && aligned16(beta) && aligned16(gamma1) && aligned16(beta1) &&
                      aligned16(hn) && (d * 2) % 16 == 0,
                  MMAMD_E_ALIGN, "vit_cls_lnpre_ln: rows must be 16-byte aligned");
  if (B == 0) return 0;
  const int rows = B * S;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int d4 = d / 4;
#define LAUNCH_VCL(MV) hipLaunchKernelGGL((vit_cls_lnpre_ln_kernel<MV>), grid, block, 0, st, x, cls, pos0, gamma, beta, eps, gamma1, beta1, eps1, (bf16*)hn, rows, S, d)
  if (d4 <= 128) LAUNCH_VCL(2); else if (d4 <= 192) LAUNCH_VCL(3); else if (d4 <= 256) LAUNCH_VCL(4); else LAUNCH_VCL(8);
#undef LAUNCH_VCL
  return launch_status("vit_cls_lnpre_ln");
}

extern "C" int mmamd_embed_tokens(const int64_t* ids, const void* table, int table_dtype, const float* pos,
                                  float* x, int B, int S, int d, int vocab, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(ids && table && pos && x && B >= 0 && S > 0 && d > 0 && vocab > 0, MMAMD_E_BADARG, "embed_tokens: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0, MMAMD_E_UNSUPPORTED, "embed_tokens: d=%d must be a multiple of 4", d);
  if (B == 0) return 0;
  const int rows = B * S;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (table_dtype == MMAMD_F32)
    hipLaunchKernelGGL((embed_tokens_kernel<float>), grid, block, 0, st, ids, (const float*)table, pos, x, rows, S, d, vocab);
  else if (table_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((embed_tokens_kernel<bf16>), grid, block, 0, st, ids, (const bf16*)table, pos, x, rows, S, d, vocab);
  else
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "embed_tokens: bad dtype code");
  return launch_status("embed_tokens");
}

extern "C" int mmamd_patchify(const void* images, int img_dtype, void* patches, int B, int C, int HW, int P,
                              int Kpad, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(images && patches && B >= 0 && C > 0 && HW > 0 && P > 0 && HW % P == 0, MMAMD_E_BADARG, "patchify: bad argument");
  MMAMD_CHECK_ARG(Kpad % 4 == 0 && Kpad >= C * P * P, MMAMD_E_BADARG, "patchify: Kpad=%d must be a multiple of 4 and >= C*P*P", Kpad);
  if (B == 0) return 0;
  const int G = HW / P;
  const size_t total = (size_t)B * G * G * (Kpad / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (img_dtype == MMAMD_F32)
    hipLaunchKernelGGL((patchify_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)images, (bf16*)patches, B, C, HW, P, Kpad);
  else if (img_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((patchify_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)images, (bf16*)patches, B, C, HW, P, Kpad);
  else
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "patchify: bad dtype code");
  return launch_status("patchify");
}

extern "C" int mmamd_pool_ln_proj(const float* x, int S, int d, const int64_t* ids, const float* gamma,
                                  const float* beta, float eps, const float* proj, int proj_sk, int proj_se,
                                  float* out, int B, int E, int normalize, float* ws, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && gamma && beta && proj && out && ws && S > 0 && d > 0 && B >= 0 && E > 0, MMAMD_E_BADARG, "pool_ln_proj: bad argument");
  MMAMD_CHECK_ARG(proj_se == 1 || proj_sk == 1, MMAMD_E_UNSUPPORTED, "pool_ln_proj: projection must be contiguous along k or e");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 pgrid((B + 3) / 4), pblock(256);
  const bool rowvec = d % 4 == 0 && aligned16(x) && aligned16(gamma) && aligned16(beta) && aligned16(ws);
  if (rowvec && d <= 512) hipLaunchKernelGGL((pool_ln_kernel<2>), pgrid, pblock, 0, st, x, S, d, ids, gamma, beta, eps, ws, B);
  else if (rowvec && d <= 1024) hipLaunchKernelGGL((pool_ln_kernel<4>), pgrid, pblock, 0, st, x, S, d, ids, gamma, beta, eps, ws, B);
  else if (rowvec && d <= 2048) hipLaunchKernelGGL((pool_ln_kernel<8>), pgrid, pblock, 0, st, x, S, d, ids, gamma, beta, eps, ws, B);
  else hipLaunchKernelGGL((pool_ln_kernel<0>), pgrid, pblock, 0, st, x, S, d, ids, gamma, beta, eps, ws, B);
  hipLaunchKernelGGL(proj_f32_kernel, dim3((E + 31) / 32, (B + 31) / 32), dim3(256), 0, st, ws, proj, proj_sk, proj_se, out, B, d, E);
  if (normalize)
    hipLaunchKernelGGL(l2_normalize_inplace_f32_kernel, dim3((B + 3) / 4), dim3(256), 0, st, out, B, E, 1e-12f);
  return launch_status("pool_ln_proj");
}

extern "C" int mmamd_l2_normalize_ld(const void* x, int x_dtype, void* y, int y_dtype, int ldy, int rows, int d, float eps,
                                     mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && y && rows >= 0 && d > 0 && ldy >= d, MMAMD_E_BADARG, "l2_normalize: bad argument");
  if (rows == 0) return 0;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == MMAMD_F32 && y_dtype == MMAMD_F32)
    hipLaunchKernelGGL((l2_normalize_kernel<float, float>), grid, block, 0, st, (const float*)x, (float*)y, rows, d, eps, ldy);
  else if (x_dtype == MMAMD_BF16 && y_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((l2_normalize_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)x, (bf16*)y, rows, d, eps, ldy);
  else if (x_dtype == MMAMD_BF16 && y_dtype == MMAMD_F32)
    hipLaunchKernelGGL((l2_normalize_kernel<bf16, float>), grid, block, 0, st, (const bf16*)x, (float*)y, rows, d, eps, ldy);
  else if (x_dtype == MMAMD_F32 && y_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((l2_normalize_kernel<float, bf16>), grid, block, 0, st, (const float*)x, (bf16*)y, rows, d, eps, ldy);
  else
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "l2_normalize: bad dtype code");
  return launch_status("l2_normalize");
}

extern "C" int mmamd_l2_normalize(const void* x, int x_dtype, void* y, int y_dtype, int rows, int d, float eps,
                                  mmamd_stream_t stream) {
  return mmamd_l2_normalize_ld(x, x_dtype, y, y_dtype, d, rows, d, eps, stream);
}

extern "C" int mmamd_clamp_scalar(float* p, int has_min, float lo, int has_max, float hi, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(p != nullptr, MMAMD_E_BADARG, "clamp_scalar: null pointer");
  hipLaunchKernelGGL(clamp_scalar_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, has_min, lo, has_max, hi);
  return launch_status("clamp_scalar");
}

extern "C" int mmamd_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(src && dst && n >= 0, MMAMD_E_BADARG, "convert: bad argument");
  if (n == 0) return 0;
  const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == MMAMD_F32 && dst_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((convert_kernel<float, bf16>), dim3(blocks), dim3(256), 0, st, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == MMAMD_BF16 && dst_dtype == MMAMD_F32)
    hipLaunchKernelGGL((convert_kernel<bf16, float>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == MMAMD_F32 && dst_dtype == MMAMD_F32)  // same-type copy: packing several parameters into one buffer
    hipLaunchKernelGGL((convert_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (src_dtype == MMAMD_BF16 && dst_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((convert_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, n);
  else
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "convert: unsupported dtype pair");
  return launch_status("convert");
}

// ---------------------------------------------------------------------------------------------
// token mean: out[b][c] = mean over tokens first .. S-1 of x[b][s][c] (GlobalAveragePooler, modules/encoders/vision_transformer.py:117-127:
// the CLS row is skipped).  One thread per 4 columns of a sample, tokens summed in order (deterministic), coalesced across the row.
// ---------------------------------------------------------------------------------------------
namespace mmamd {
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int S, int d, int first) {
  const int b = blockIdx.y, c4 = blockIdx.x * 256 + threadIdx.x;
  if (4 * c4 >= d) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* base = x + (size_t)b * S * d + 4 * c4;
  for (int s = first; s < S; ++s) {
    const f32x4 v = load4(base + (size_t)s * d);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
  const float inv = 1.0f / (float)(S - first);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] *= inv;
  store4(out + (size_t)b * d + 4 * c4, acc);
}
}  // namespace mmamd

extern "C" int mmamd_token_mean(const float* x, float* out, int B, int S, int d, int first, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && out && B >= 0 && S > 0 && d > 0 && first >= 0 && first < S, MMAMD_E_BADARG, "token_mean: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0 && aligned16(x) && aligned16(out), MMAMD_E_ALIGN, "token_mean: d %% 4 == 0 and 16-byte aligned pointers");
  if (B == 0) return 0;
  hipLaunchKernelGGL(mmamd::token_mean_kernel, dim3((d / 4 + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, out, S, d, first);
  return mmamd::launch_status("token_mean");
}

extern "C" int mmamd_bert_embed_ln(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                                   const float* pos, const float* type, const float* gamma, const float* beta, float eps, float* x,
                                   int B, int S, int d, int vocab, int max_pos, int n_types, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(ids && word && pos && type && x && (gamma == nullptr) == (beta == nullptr) && B >= 0 && S > 0 && d > 0, MMAMD_E_BADARG, "bert_embed_ln: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0 && d <= 2048, MMAMD_E_UNSUPPORTED, "bert_embed_ln: d=%d must be a multiple of 4 and <= 2048", d);
  MMAMD_CHECK_ARG(pos_ids != nullptr || S <= max_pos, MMAMD_E_BADARG, "bert_embed_ln: sequence longer than the position table");
  if (B == 0) return 0;
  const int rows = B * S;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int d4 = d / 4;
#define LAUNCH_BE(MV) hipLaunchKernelGGL((bert_embed_ln_kernel<MV>), grid, block, 0, st, ids, type_ids, pos_ids, word, pos, type, gamma, beta, eps, x, rows, S, d, vocab, max_pos, n_types)
  if (d4 <= 128) LAUNCH_BE(2); else if (d4 <= 256) LAUNCH_BE(4); else LAUNCH_BE(8);
#undef LAUNCH_BE
  return launch_status("bert_embed_ln");
}

extern "C" int mmamd_flava_image_embed(const float* patch_emb, const float* cls, const float* pos, const int64_t* patches_mask,
                                       const float* mask_token, float* x, int B, int G2, int d, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(patch_emb && pos && x && B >= 0 && G2 > 0 && d > 0 && d % 4 == 0, MMAMD_E_BADARG, "flava_image_embed: bad argument");
  if (B == 0) return 0;
  const int rows = B * (G2 + (cls ? 1 : 0));
  hipLaunchKernelGGL(flava_image_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, patch_emb, cls, pos,
                     patches_mask, mask_token, x, B, G2, d);
  return launch_status("flava_image_embed");
}

extern "C" int mmamd_rows_linear_f32(const float* h, int64_t ldh, const float* W, const float* bias, int act, float* out, int B,
                                     int d, int E, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(h && W && out && B >= 0 && d > 0 && E > 0 && ldh >= d && act >= 0 && act <= 2, MMAMD_E_BADARG, "rows_linear_f32: bad argument");
  if (B == 0) return 0;
  hipLaunchKernelGGL(rows_linear_f32_kernel, dim3((E + 127) / 128, (B + 31) / 32), dim3(256), 0, (hipStream_t)stream, h, (size_t)ldh,
                     W, bias, act, out, B, d, E);
  return launch_status("rows_linear_f32");
}

extern "C" int mmamd_key_mask(const void* src, int kind, int64_t pad_id, uint8_t* out, int64_t n, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(src && out && n >= 0 && kind >= 0 && kind <= 3, MMAMD_E_BADARG, "key_mask: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(key_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, kind, (long long)pad_id,
                     out, (long long)n);
  return launch_status("key_mask");
}

extern "C" int mmamd_coca_text_embed(const int64_t* ids, const float* table, const float* pos, const float* cls, float* x, int B,
                                     int S_ids, int d, int vocab, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(ids && table && pos && x && B >= 0 && S_ids > 0 && d > 0 && d % 4 == 0 && vocab > 0, MMAMD_E_BADARG, "coca_text_embed: bad argument");
  if (B == 0) return 0;
  const int rows = B * (S_ids + (cls ? 1 : 0));
  hipLaunchKernelGGL(coca_text_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, table, pos, cls, x, B, S_ids, d, vocab);
  return launch_status("coca_text_embed");
}

extern "C" int mmamd_coca_text_mask(const void* src, int kind, int64_t pad_id, uint8_t* out, int B, int S, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(src && out && B >= 0 && S > 0 && kind >= 0 && kind <= 3, MMAMD_E_BADARG, "coca_text_mask: bad argument");
  if (B == 0) return 0;
  const long long n = (long long)B * (S + 1) * (S + 1);
  hipLaunchKernelGGL(coca_text_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, kind, (long long)pad_id, out, B, S);
  return launch_status("coca_text_mask");
}

extern "C" int mmamd_bicubic_pos_embed(const float* pos, int n_side, int d, float* out, int h0, int w0, float scale_h, float scale_w,
                                       mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(pos && out && n_side > 0 && d > 0 && h0 > 0 && w0 > 0 && scale_h > 0.f && scale_w > 0.f, MMAMD_E_BADARG, "bicubic_pos_embed: bad argument");
  hipLaunchKernelGGL(bicubic_pos_embed_kernel, dim3(1 + h0 * w0), dim3(256), 0, (hipStream_t)stream, pos, n_side, d, out, h0, w0, 1.0f / scale_h,
                     1.0f / scale_w);
  return launch_status("bicubic_pos_embed");
}

extern "C" int mmamd_mask_labels(int64_t* labels, const uint8_t* keep, int64_t fill, int64_t n, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(labels && keep && n >= 0, MMAMD_E_BADARG, "mask_labels: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(mask_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long*)labels, keep,
                     (long long)fill, (long long)n);
  return launch_status("mask_labels");
}

extern "C" int mmamd_offset_position_ids(const int64_t* ids, int64_t pad_id, int64_t* out, int B, int S, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(ids && out && B >= 0 && S > 0, MMAMD_E_BADARG, "offset_position_ids: bad argument");
  if (B == 0) return 0;
  hipLaunchKernelGGL(offset_position_ids_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const long long*)ids, (long long)pad_id,
                     (long long*)out, B, S);
  return launch_status("offset_position_ids");
}
