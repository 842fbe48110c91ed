// backward.hip — row / elementwise kernels of the training step: what torch autograd runs for
// LayerNorm, QuickGELU / GELU, bias adds, F.normalize and the embedding lookups of the CLIP towers, plus the bf16 transpose that
// lets the NT MFMA GEMM compute weight gradients (dW = dY^T X: both operands are needed with the token index contiguous).
// All HBM-bound: wave per row, float4 / bf16x4 accesses, fp32 accumulation; column reductions are two-stage (no atomics),
// the embedding-table gradient uses fp32 atomics (rows collide by construction).
#include <type_traits>

#include "common.h"

namespace mmamd {

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  y = (x - mu) rstd * gamma + beta.  dx = rstd (g - mean(g) - xh mean(g xh)), g = dy * gamma, xh = (x - mu) rstd
// dx (+= add) in fp32;  per-block partial column sums of dy*xh (dgamma) and dy (dbeta) -> part[block][2][d]
// ---------------------------------------------------------------------------------------------
template <typename TD, int MAXV>
__global__ __launch_bounds__(256, MAXV <= 3 ? 4 : MAXV <= 4 ? 3 : 1) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const TD* __restrict__ dy, const float* __restrict__ add,
                                                            float* __restrict__ dx, bf16* __restrict__ dx_bf16, float* __restrict__ part,
                                                            int rows, int d, float eps, int with_cs) {
  // with_cs: also the column sums of the OUTPUT dx (third partial slot) — dx is the dY of the Linear that produced this LayerNorm's
  // input stream (out-projection / MLP-down), so its bias gradient needs no pass of its own
  __shared__ float red[4][3][MAXV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d4 = d >> 2;
  f32x4 gacc[MAXV], bacc[MAXV], cacc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) { gacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * d;
    const TD* dr = dy + (size_t)row * d;
    // every load of the row is issued before the first reduction: one memory round trip per row instead of three dependent ones
    // (x -> statistics -> dy -> means -> add): the first form ran at 2.7 TB/s, latency-bound with 2 waves per SIMD
    // (dy stays in its storage type until it is used: a bf16 dy then holds 2 registers per chunk instead of 4 while the row's loads are in flight --
    //  the bf16 instantiation at d = 768 needed 183 registers, one step over the 168 that allow three waves per SIMD: 179 vs 128 us, r05)
    typedef typename std::conditional<std::is_same<TD, float>::value, f32x4, bf16x4>::type raw4;
    f32x4 xv[MAXV], gv[MAXV], av[MAXV];
    raw4 graw[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      xv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) graw[i][j] = (TD)0.f;
      if (c < d4) {
        xv[i] = load4(xr + 4 * c);
        graw[i] = *reinterpret_cast<const raw4*>(dr + 4 * c);
        if (add != nullptr) av[i] = load4(add + (size_t)row * d + 4 * c);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);  // lanes past d hold zeros
    const float mean = wave_sum(sum) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float u = xv[i][j] - mean; q += u * u; }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        const f32x4 gm = load4(gamma + 4 * c);  // L1-resident: kept out of the registers the row loads need
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[i][j] - mean) * rstd;
          const float dyj = (float)graw[i][j];
          xv[i][j] = xh;
          gv[i][j] = dyj * gm[j];
          sg += gv[i][j];
          sgx += gv[i][j] * xh;
          gacc[i][j] += dyj * xh;
          bacc[i][j] += dyj;
        }
      }
    }
    const float mg = wave_sum(sg) / (float)d, mgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (gv[i][j] - mg - xv[i][j] * mgx) + av[i][j];
        if (with_cs) {
#pragma unroll
          for (int j = 0; j < 4; ++j) cacc[i][j] += o[j];
        }
        store4(dx + (size_t)row * d + 4 * c, o);
        if (dx_bf16 != nullptr) store4(dx_bf16 + (size_t)row * d + 4 * c, o);  // MFMA operand of the next dgrad / wgrad
      }
    }
  }
  // block partials: 4 waves -> LDS -> wave 0 sums and writes part[blockIdx][{gamma, beta}][d]
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[wave][0][4 * c + j] = gacc[i][j]; red[wave][1][4 * c + j] = bacc[i][j]; red[wave][2][4 * c + j] = cacc[i][j]; }
  }
  __syncthreads();
  const int nslot = with_cs ? 3 : 2;  // part layout [G][nslot][d]
  for (int c = threadIdx.x; c < d; c += 256) {
    part[((size_t)blockIdx.x * nslot + 0) * d + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
    part[((size_t)blockIdx.x * nslot + 1) * d + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
    if (with_cs) part[((size_t)blockIdx.x * 3 + 2) * d + c] = (red[0][2][c] + red[1][2][c]) + (red[2][2][c] + red[3][2][c]);
  }
}

// out[c] = sum_g part[g][c]  (second stage of every column reduction).  Block = 32 columns (8 threads x 4 columns, 16-byte loads) x 32
// partial-row groups, the loop unrolled by four so each thread keeps 4 independent loads in flight: the first form (thread per
// column, 8 groups) was latency-bound — 22 us average over the 596 calls of a CLIP training step, 4 % of the step.
// (out1 / out2 / seg: the result is split into segments of `seg` columns that go to up to three separate arrays — the LayerNorm backward's
// dgamma | dbeta | column sums of dx; out1 == NULL: one array)
__device__ __forceinline__ void colsum_stage2_body(const float* __restrict__ part, int G, int n, float* __restrict__ out, float* __restrict__ out1,
                                                   float* __restrict__ out2, int seg, int block) {
  __shared__ f32x4 red[32][8];
  const int cq = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int c = block * 32 + cq * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if ((n & 3) == 0) {
    if (c < n) {
      int g = grp;
      for (; g + 96 < G; g += 128) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(part + (size_t)g * n + c);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(part + (size_t)(g + 32) * n + c);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(part + (size_t)(g + 64) * n + c);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(part + (size_t)(g + 96) * n + c);
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
      }
      for (; g < G; g += 32) s0 += *reinterpret_cast<const f32x4*>(part + (size_t)g * n + c);
    }
  } else {
    for (int g = grp; g < G; g += 32)
      for (int j = 0; j < 4; ++j)
        if (c + j < n) s0[j] += part[(size_t)g * n + c + j];
  }
  red[grp][cq] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int st = 16; st >= 1; st >>= 1) {
    if (grp < st) red[grp][cq] += red[grp + st][cq];
    __syncthreads();
  }
  if (grp == 0)
    for (int j = 0; j < 4; ++j)
      if (c + j < n) {
        if (out1 == nullptr) {
          out[c + j] = red[0][cq][j];
        } else {
          const int which = (c + j) / seg, cc = (c + j) - which * seg;
          (which == 0 ? out : which == 1 ? out1 : out2)[cc] = red[0][cq][j];
        }
      }
}

__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ part, int G, int n, float* __restrict__ out,
                                                            float* __restrict__ out1 = nullptr, float* __restrict__ out2 = nullptr, int seg = 0) {
  colsum_stage2_body(part, G, n, out, out1, out2, seg, blockIdx.x);
}

// The same reduction for up to 64 independent (partials -> up to three result arrays) jobs in ONE launch (blockIdx.y = job): the LayerNorm backward calls of a
// layer stack park their partials and the stack's backward reduces all of them at its end (r05: 53 launches of ~25 us on the training step's critical path,
// each with 72 workgroups on 256 CUs, become two).  Same arithmetic per job as colsum_stage2_kernel -> bit-identical results.
struct ColsumJobs {
  mmamd_colsum_job j[64];
};
__global__ __launch_bounds__(256) void colsum_stage2_batched_kernel(const ColsumJobs jobs) {
  const mmamd_colsum_job& jb = jobs.j[blockIdx.y];
  if ((int)blockIdx.x * 32 >= jb.n) return;  // (whole workgroup: no barrier is skipped by part of it)
  colsum_stage2_body(jb.part, jb.G, jb.n, jb.out0, jb.out1, jb.out2, jb.seg, blockIdx.x);
}

// column sums of x [rows, n] (bias gradients): stage 1.  Workgroup g owns rows [g*rpb, (g+1)*rpb); a thread owns one 16-byte column
// chunk (8 bf16 / 4 fp32) and, when the row is narrower than 256 chunks, one of 256/nch row lanes, so a wave-instruction reads
// whole contiguous row segments; partial sums of the row lanes meet in LDS.  part[g][n].  (The first version gave each thread 4
// columns and a grid-strided row walk: 66-106 us for the 77-310 MB operands of a ViT-B/16 layer, 1-3 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const T* __restrict__ x, int rows, int n, float* __restrict__ part, int rpb) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[256 * VEC];
  const int nch = n / VEC;  // launcher guarantees n % VEC == 0
  const int r0 = blockIdx.x * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
  const int rpar = nch >= 256 ? 1 : 256 / nch;
  const int ch0 = nch >= 256 ? (int)threadIdx.x : (int)threadIdx.x % nch, rl = nch >= 256 ? 0 : (int)threadIdx.x / nch;
  for (int cb = 0; cb < nch; cb += 256) {  // one trip unless the row has more than 256 chunks
    const int ch = cb + ch0;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (ch < nch && rl < rpar) {
      int r = r0 + rl;
      if constexpr (sizeof(T) == 2) {  // four rows in flight per thread (the dependent-looking accumulate loop kept one)
        for (; r + 3 * rpar < r1; r += 4 * rpar) {
          const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(x + (size_t)r * n + ch * VEC);
          const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + rpar) * n + ch * VEC);
          const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 2 * rpar) * n + ch * VEC);
          const bf16x8 v3 = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + 3 * rpar) * n + ch * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) { acc[j] += (float)v0[j]; acc[j] += (float)v1[j]; acc[j] += (float)v2[j]; acc[j] += (float)v3[j]; }
        }
      }
      for (; r < r1; r += rpar) {
        if constexpr (sizeof(T) == 2) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (size_t)r * n + ch * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += (float)v[j];
        } else {
          const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * n + ch * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += v[j];
        }
      }
    }
    if (rpar > 1) {
      __syncthreads();
      if (ch < nch && rl < rpar)
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[(rl * nch + ch) * VEC + j] = acc[j];
      __syncthreads();
      if (rl == 0 && ch < nch) {
        for (int q = 1; q < rpar; ++q)
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += red[(q * nch + ch) * VEC + j];
      }
    }
    if (rl == 0 && ch < nch)
#pragma unroll
      for (int j = 0; j < VEC; ++j) part[(size_t)blockIdx.x * n + ch * VEC + j] = acc[j];
  }
}

// generic fallback (n not a multiple of the 16-byte chunk): thread per column, grid-strided rows
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1_slow_kernel(const T* __restrict__ x, int rows, int n, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) s += to_f32(x[(size_t)r * n + c]);
  part[(size_t)blockIdx.y * n + c] = s;
}

// ---------------------------------------------------------------------------------------------
// activations: g = act(u) (training forward keeps u) and du = dg * act'(u)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_value(float u, int act) {
  if (act == MMAMD_ACT_QUICKGELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
  return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
}
__device__ __forceinline__ float act_grad(float u, int act) {
  if (act == MMAMD_ACT_QUICKGELU) {  // d/du [u s(1.702 u)] = s + 1.702 u s (1 - s)
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
    return sg * (1.0f + 1.702f * u * (1.0f - sg));
  }
  const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752f));  // d/du [u Phi(u)] = Phi + u phi
  return cdf + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}
// dz = dy where the ReLU output y is positive (classifier MLP of FLAVAForClassification, fp32)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16* __restrict__ u, bf16* __restrict__ g, int64_t n4, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = load4(u + 4 * i);
    store4(g + 4 * i, f32x4{act_value(v[0], act), act_value(v[1], act), act_value(v[2], act), act_value(v[3], act)});
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ u, const bf16* __restrict__ dg, bf16* __restrict__ du,
                                                      int64_t n4, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = load4(u + 4 * i), d = load4(dg + 4 * i);
    store4(du + 4 * i, f32x4{d[0] * act_grad(v[0], act), d[1] * act_grad(v[1], act), d[2] * act_grad(v[2], act), d[3] * act_grad(v[3], act)});
  }
}

// stand-alone activation module (modules/layers/activation.py:24-25 called on its own): any element count, fp32 or bf16
template <typename T>
__global__ __launch_bounds__(256) void act_elem_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = to_f32<T>(x[i]);
    out[i] = (T)(dy == nullptr ? act_value(v, act) : to_f32<T>(dy[i]) * act_grad(v, act));
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 transpose with zero padding: dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < ld_dst   (64 x 64 tiles through LDS)
// ---------------------------------------------------------------------------------------------
// 64 x 64 tiles through LDS, 8-byte global accesses on both sides (bf16 source) — and, optionally, the column sums of the
// source on the way (colpart[blockIdx.y][c] = sum of this tile's 64 rows of column c): every dY that is transposed for a
// weight-gradient GEMM is also the operand of a bias gradient, so the separate column-sum pass over it disappears.
template <typename TS>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const TS* __restrict__ src, int64_t ld_src, bf16* __restrict__ dst,
                                                                int rows, int cols, int ld_dst, float* __restrict__ colpart) {
  __shared__ bf16 tile[64][68];  // row pitch 136 B: 8-byte aligned rows
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  const int lr = t >> 4, lc = (t & 15) * 4;  // load: 16 threads x 4 columns per row, 16 rows per pass
  const bool vec = ((ld_src & 3) == 0) && (c0 + 64 <= cols) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int r = r0 + ps * 16 + lr;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      if (vec) v = load4(src + (size_t)r * ld_src + c0 + lc);
      else
        for (int j = 0; j < 4; ++j)
          if (c0 + lc + j < cols) v[j] = to_f32(src[(size_t)r * ld_src + c0 + lc + j]);
    }
    bf16x4 b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j] = (bf16)v[j]; csum[j] += (float)b[j]; }
    *reinterpret_cast<bf16x4*>(&tile[ps * 16 + lr][lc]) = b;
  }
  __syncthreads();
  // store: thread = (dst row = source column sc, 4 consecutive dst columns = source rows sr..sr+3)
  const int sc = t >> 4, sr = (t & 15) * 4;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int c = c0 + ps * 16 + sc;
    const int r = r0 + sr;
    if (c < cols && r < ld_dst) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = tile[sr + j][ps * 16 + sc];
      if (r + 3 < ld_dst) *reinterpret_cast<bf16x4*>(dst + (size_t)c * ld_dst + r) = o;
      else
        for (int j = 0; j < 4 && r + j < ld_dst; ++j) dst[(size_t)c * ld_dst + r + j] = o[j];
    }
  }
  if (colpart != nullptr) {
    // the 16 threads with the same (t & 15) hold partial sums of the same 4 columns over different rows: reduce through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(&tile[0][0]);  // 16 x 64 floats = 4 KiB <= 8.7 KiB
#pragma unroll
    for (int j = 0; j < 4; ++j) red[lr * 64 + lc + j] = csum[j];
    __syncthreads();
    if (t < 64) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) sum += red[i * 64 + t];
      if (c0 + t < cols) colpart[(size_t)blockIdx.y * cols + c0 + t] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight pack for a training step: up to 64 fp32 [rows, cols] matrices (the Linear weights of a layer stack) -> their bf16 copies [rows, cols]
// (the forward GEMMs' W operand) AND their bf16 transposes [cols, ld_t] (the dgrad GEMMs' operand: dX = dY W = gemm(dY, W^T)), ONE launch instead
// of a convert and a transpose launch per weight per step (the 99 + 96 five-microsecond launches of a CLIP ViT-B/16 training step).
// A workgroup owns one 64 x 64 tile; tensor / tile from the prefix table in the kernel arguments.  Same rounding as mmamd_convert /
// mmamd_transpose_to_bf16 (one fp32 -> bf16 conversion per element): bit-identical copies.
// ---------------------------------------------------------------------------------------------
struct PackDesc {
  const float* src;
  bf16* nt;   // [rows, cols] or NULL
  bf16* tr;   // [cols, ld_t] (columns >= rows zero-filled up to ld_t) or NULL
  int rows, cols, ld_t;
  int tile0;  // first tile id of this tensor
};
struct PackArgs {
  PackDesc d[64];
  int n, total;
};
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
  __shared__ bf16 tile[64][68];
  int ti = 0;
  {
    int lo = 0, hi = a.n - 1;  // last tensor whose tile0 <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.d[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    ti = lo;
  }
  const PackDesc& p = a.d[ti];
  const int tiles_c = (p.cols + 63) >> 6;
  const int tid_ = (int)blockIdx.x - p.tile0;
  const int r0 = (tid_ / tiles_c) * 64, c0 = (tid_ - (tid_ / tiles_c) * tiles_c) * 64;
  const int t = threadIdx.x;
  const int lr = t >> 4, lc = (t & 15) * 4;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int r = r0 + ps * 16 + lr;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < p.rows) {
      if (c0 + lc + 3 < p.cols && (p.cols & 3) == 0) v = *reinterpret_cast<const f32x4*>(p.src + (size_t)r * p.cols + c0 + lc);
      else
        for (int j = 0; j < 4; ++j)
          if (c0 + lc + j < p.cols) v[j] = p.src[(size_t)r * p.cols + c0 + lc + j];
    }
    bf16x4 b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = (bf16)v[j];
    *reinterpret_cast<bf16x4*>(&tile[ps * 16 + lr][lc]) = b;
    if (p.nt != nullptr && r < p.rows) {
      if (c0 + lc + 3 < p.cols && (p.cols & 3) == 0) *reinterpret_cast<bf16x4*>(p.nt + (size_t)r * p.cols + c0 + lc) = b;
      else
        for (int j = 0; j < 4; ++j)
          if (c0 + lc + j < p.cols) p.nt[(size_t)r * p.cols + c0 + lc + j] = b[j];
    }
  }
  if (p.tr == nullptr) return;
  __syncthreads();
  const int sc = t >> 4, sr = (t & 15) * 4;  // dst row = source column sc (+16 ps), 4 consecutive dst columns = source rows sr..sr+3
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int c = c0 + ps * 16 + sc, r = r0 + sr;
    if (c < p.cols && r < p.ld_t) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = tile[sr + j][ps * 16 + sc];  // rows >= p.rows were staged as zeros
      if (r + 3 < p.ld_t) *reinterpret_cast<bf16x4*>(p.tr + (size_t)c * p.ld_t + r) = o;
      else
        for (int j = 0; j < 4 && r + j < p.ld_t; ++j) p.tr[(size_t)c * p.ld_t + r + j] = o[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// F.normalize backward: y = x / max(|x|, eps);  dx = (dy - y (y . dy)) / max(|x|, eps)      (wave per row, fp32)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * d;
  const float* gr = dy + (size_t)row * d;
  float ss = 0.f, sd = 0.f;
  for (int c = lane; c < d; c += 64) { ss += xr[c] * xr[c]; sd += xr[c] * gr[c]; }
  ss = wave_sum(ss); sd = wave_sum(sd);
  const float nrm = sqrtf(ss);
  const float den = fmaxf(nrm, eps);
  // below eps the forward divides by the constant eps: plain scaling
  const float k = nrm > eps ? sd / (den * den * den) : 0.f;
  for (int c = lane; c < d; c += 64) dx[(size_t)row * d + c] = gr[c] / den - xr[c] * k;
}

// dst[idx[i], :] += src[i, :]  (fp32 atomics: token-embedding gradient; pooled-row gradient scattered into the sequence)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int n, int d,
                                                               float* __restrict__ dst, int64_t dst_rows) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t r = idx[i];
  if (r < 0 || r >= dst_rows) return;
  for (int c = lane; c < d; c += 64) atomicAdd(dst + (size_t)r * d + c, src[(size_t)i * d + c]);
}

}  // namespace mmamd

using namespace mmamd;

// workgroups of the LayerNorm backward = what is resident at once (the kernel walks rows with a grid stride, so a second round would be a tail):
// 4 per CU where the instantiation fits four waves per SIMD (d <= 768, but more than 512: 120 registers, 36 KB LDS), 3 per CU otherwise
extern "C" int mmamd_layernorm_bwd_groups(int rows, int d) {
  const int cap = (d > 512 && d <= 768) ? 1024 : 768;
  return rows < 4 * cap ? (rows + 3) / 4 : cap;
}

extern "C" int mmamd_layernorm_bwd(const float* x, const float* gamma, const void* dy, int dy_dtype, const float* add, float* dx,
                                   void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, float* ws, int rows, int d, float eps,
                                   mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && gamma && dy && dx && ws && rows > 0 && d > 0 && (dgamma != nullptr) == (dbeta != nullptr), MMAMD_E_BADARG, "layernorm_bwd: bad argument");
  MMAMD_CHECK_ARG(d % 4 == 0 && d <= 2048, MMAMD_E_UNSUPPORTED, "layernorm_bwd: d=%d must be a multiple of 4 and <= 2048", d);
  hipStream_t st = (hipStream_t)stream;
  const int G = mmamd_layernorm_bwd_groups(rows, d);  // ws: (G + 1) * 3 * d floats
  const int d4 = d / 4, cs = dx_colsum != nullptr, ns = cs ? 3 : 2;
#define LNB(T, MV) hipLaunchKernelGGL((layernorm_bwd_kernel<T, MV>), dim3(G), dim3(256), 0, st, x, gamma, (const T*)dy, add, dx, (bf16*)dx_bf16, ws, rows, d, eps, cs)
  if (dy_dtype == MMAMD_F32) { if (d4 <= 128) LNB(float, 2); else if (d4 <= 192) LNB(float, 3); else if (d4 <= 256) LNB(float, 4); else LNB(float, 8); }
  else if (dy_dtype == MMAMD_BF16) { if (d4 <= 128) LNB(bf16, 2); else if (d4 <= 192) LNB(bf16, 3); else if (d4 <= 256) LNB(bf16, 4); else LNB(bf16, 8); }
  else MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "layernorm_bwd: bad dy dtype");
#undef LNB
  // part layout [G][ns][d] = G rows of width ns*d: dgamma | dbeta | (column sums of dx)
  // (the three results go straight to their arrays: the three 3 KB device-to-device copies this used to end with were 150 of the 182
  //  copies of a CLIP training step, 0.7 ms of serialised 5 us operations)
  // dgamma == dbeta == NULL: the caller reduces the partials later (mmamd_colsum_stage2_batched: job {ws, G, ns * d, dgamma, dbeta, dx_colsum, seg = d})
  if (dgamma != nullptr) hipLaunchKernelGGL(colsum_stage2_kernel, dim3((ns * d + 31) / 32), dim3(256), 0, st, ws, G, ns * d, dgamma, dbeta, dx_colsum, d);
  return launch_status("layernorm_bwd");
}

extern "C" int mmamd_colsum_stage2_batched(const mmamd_colsum_job* jobs, int njobs, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(njobs >= 0 && (jobs != nullptr || njobs == 0), MMAMD_E_BADARG, "colsum_stage2_batched: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += 64) {
    ColsumJobs a;
    const int nj = njobs - j0 < 64 ? njobs - j0 : 64;
    int nmax = 0;
    for (int i = 0; i < nj; ++i) {
      const mmamd_colsum_job& jb = jobs[j0 + i];
      MMAMD_CHECK_ARG(jb.part && jb.out0 && jb.G > 0 && jb.n > 0 && (jb.out1 == nullptr || jb.seg > 0), MMAMD_E_BADARG, "colsum_stage2_batched: bad job %d", j0 + i);
      a.j[i] = jb;
      if (jb.n > nmax) nmax = jb.n;
    }
    hipLaunchKernelGGL(colsum_stage2_batched_kernel, dim3((nmax + 31) / 32, nj), dim3(256), 0, st, a);
  }
  return launch_status("colsum_stage2_batched");
}

namespace mmamd {
// FEW rows of a VERY wide matrix (the positional-embedding gradient: column sums of d_asm viewed as [B, S * w] = [256, 151 296]): one row per workgroup
// (colsum_stage1_kernel) writes as many partial bytes as it reads -- 155 MB in, 155 MB of partials out, 155 MB back in: 100 + 23 us.  Here a thread owns
// one 16-byte chunk of columns and walks a row group with four loads in flight; the partials are [row groups <= 8][n].  r05.
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* __restrict__ x, int rows, int n, float* __restrict__ part, int rpg) {
  constexpr int VEC = 16 / sizeof(T);
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type vec_t;
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch * VEC >= n) return;
  const int r0 = blockIdx.y * rpg, r1 = r0 + rpg < rows ? r0 + rpg : rows;
  const T* p = x + (size_t)ch * VEC;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    const vec_t v0 = *reinterpret_cast<const vec_t*>(p + (size_t)r * n), v1 = *reinterpret_cast<const vec_t*>(p + (size_t)(r + 1) * n);
    const vec_t v2 = *reinterpret_cast<const vec_t*>(p + (size_t)(r + 2) * n), v3 = *reinterpret_cast<const vec_t*>(p + (size_t)(r + 3) * n);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc[j] += (float)v0[j]; acc[j] += (float)v1[j]; acc[j] += (float)v2[j]; acc[j] += (float)v3[j]; }
  }
  for (; r < r1; ++r) {
    const vec_t v = *reinterpret_cast<const vec_t*>(p + (size_t)r * n);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] += (float)v[j];
  }
  float* o = part + (size_t)blockIdx.y * n + (size_t)ch * VEC;
#pragma unroll
  for (int j = 0; j < VEC; ++j) o[j] = acc[j];
}
}  // namespace mmamd

static int g_colsum_wide = 1;
extern "C" int mmamd_debug_set_colsum_wide(int on) { g_colsum_wide = on != 0; return 0; }

extern "C" int mmamd_colsum(const void* x, int dtype, int rows, int n, float* out, float* ws, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && out && ws && rows > 0 && n > 0, MMAMD_E_BADARG, "colsum: bad argument");
  MMAMD_CHECK_ARG(dtype == MMAMD_F32 || dtype == MMAMD_BF16, MMAMD_E_BADARG, "colsum: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == MMAMD_F32 ? 4 : 8;
  int G;  // ws: min(1024, rows) * n floats
  if (g_colsum_wide && n % vec == 0 && aligned16(x) && rows <= 2048 && (long long)n >= 64LL * rows && n >= 16384) {
    const int rg = rows >= 64 ? 8 : 1;  // row groups: 8 x (n / vec / 256) workgroups keep every CU loading
    const int rpg = (rows + rg - 1) / rg;
    G = (rows + rpg - 1) / rpg;
    const dim3 grid((n / vec + 255) / 256, G);
    if (dtype == MMAMD_F32) hipLaunchKernelGGL((colsum_wide_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows, n, ws, rpg);
    else hipLaunchKernelGGL((colsum_wide_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, rows, n, ws, rpg);
  } else if (n % vec == 0 && aligned16(x)) {
    const int rpb = (rows + 1023) / 1024;
    G = (rows + rpb - 1) / rpb;
    if (dtype == MMAMD_F32) hipLaunchKernelGGL((colsum_stage1_kernel<float>), dim3(G), dim3(256), 0, st, (const float*)x, rows, n, ws, rpb);
    else hipLaunchKernelGGL((colsum_stage1_kernel<bf16>), dim3(G), dim3(256), 0, st, (const bf16*)x, rows, n, ws, rpb);
  } else {
    G = rows < 256 ? rows : 256;
    const dim3 grid((n + 255) / 256, G);
    if (dtype == MMAMD_F32) hipLaunchKernelGGL((colsum_stage1_slow_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows, n, ws);
    else hipLaunchKernelGGL((colsum_stage1_slow_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, rows, n, ws);
  }
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((n + 31) / 32), dim3(256), 0, st, ws, G, n, out);
  return launch_status("colsum");
}

extern "C" int mmamd_act_fwd(const void* u, void* g, int64_t n, int act, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(u && g && n >= 0 && n % 4 == 0 && (act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF), MMAMD_E_BADARG, "act_fwd: bad argument");
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (bf16*)g, n4, act);
  return launch_status("act_fwd");
}

extern "C" int mmamd_act_bwd(const void* u, const void* dg, void* du, int64_t n, int act, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(u && dg && du && n >= 0 && n % 4 == 0 && (act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF), MMAMD_E_BADARG, "act_bwd: bad argument");
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)u, (const bf16*)dg, (bf16*)du, n4, act);
  return launch_status("act_bwd");
}

extern "C" int mmamd_activation(const void* x, const void* dy, void* out, int dtype, int64_t n, int act, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && out && n >= 0 && (act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF), MMAMD_E_BADARG, "activation: bad argument");
  MMAMD_CHECK_ARG(dtype == MMAMD_F32 || dtype == MMAMD_BF16, MMAMD_E_BADARG, "activation: bad dtype %d", dtype);
  if (n == 0) return 0;
  const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  if (dtype == MMAMD_F32) hipLaunchKernelGGL((act_elem_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)out, n, act);
  else hipLaunchKernelGGL((act_elem_kernel<bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, (bf16*)out, n, act);
  return launch_status("activation");
}

extern "C" int mmamd_transpose_to_bf16(const void* src, int src_dtype, int64_t ld_src, void* dst, int rows, int cols, int ld_dst,
                                       float* colsum, float* ws, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, MMAMD_E_BADARG, "transpose: bad argument");
  MMAMD_CHECK_ARG(ld_dst % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0, MMAMD_E_ALIGN, "transpose: dst rows must be 8-byte aligned");
  MMAMD_CHECK_ARG(colsum == nullptr || ws != nullptr, MMAMD_E_BADARG, "transpose: colsum needs a workspace");
  const dim3 grid((cols + 63) / 64, (ld_dst + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  float* part = colsum ? ws : nullptr;  // [grid.y][cols]
  if (src_dtype == MMAMD_BF16) hipLaunchKernelGGL((transpose_to_bf16_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)src, ld_src, (bf16*)dst, rows, cols, ld_dst, part);
  else if (src_dtype == MMAMD_F32) hipLaunchKernelGGL((transpose_to_bf16_kernel<float>), grid, dim3(256), 0, st, (const float*)src, ld_src, (bf16*)dst, rows, cols, ld_dst, part);
  else MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "transpose: bad dtype");
  if (colsum != nullptr) hipLaunchKernelGGL(colsum_stage2_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, ws, (int)grid.y, cols, colsum);
  return launch_status("transpose_to_bf16");
}

extern "C" int mmamd_pack_weights(const mmamd_pack_desc* descs, int n, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(descs != nullptr && n >= 1 && n <= 64, MMAMD_E_BADARG, "pack_weights: 1 .. 64 tensors per call, got %d", n);
  PackArgs a;
  a.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const mmamd_pack_desc& q = descs[i];
    MMAMD_CHECK_ARG(q.src && q.rows > 0 && q.cols > 0 && (q.nt || q.tr), MMAMD_E_BADARG, "pack_weights: tensor %d: bad argument", i);
    MMAMD_CHECK_ARG(q.tr == nullptr || (q.ld_t >= q.rows && q.ld_t % 4 == 0), MMAMD_E_BADARG, "pack_weights: tensor %d: ld_t must be >= rows and a multiple of 4", i);
    MMAMD_CHECK_ARG(aligned16(q.src) && ((uintptr_t)q.nt & 7) == 0 && ((uintptr_t)q.tr & 7) == 0, MMAMD_E_ALIGN, "pack_weights: tensor %d: alignment", i);
    PackDesc& d = a.d[i];
    d.src = q.src; d.nt = (bf16*)q.nt; d.tr = (bf16*)q.tr; d.rows = q.rows; d.cols = q.cols; d.ld_t = q.tr ? q.ld_t : 0; d.tile0 = tiles;
    const int rows_t = q.tr ? (q.ld_t > q.rows ? q.ld_t : q.rows) : q.rows;  // the zero tail of the transposes is written too
    tiles += ((rows_t + 63) / 64) * ((q.cols + 63) / 64);
  }
  a.total = tiles;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status("pack_weights");
}

extern "C" int mmamd_l2_normalize_bwd(const float* x, const float* dy, float* dx, int rows, int d, float eps, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && dy && dx && rows >= 0 && d > 0, MMAMD_E_BADARG, "l2_normalize_bwd: bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(l2_normalize_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, d, eps);
  return launch_status("l2_normalize_bwd");
}

extern "C" int mmamd_scatter_add_rows(const float* src, const int64_t* idx, int n, int d, float* dst, int64_t dst_rows, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(src && idx && dst && n >= 0 && d > 0 && dst_rows > 0, MMAMD_E_BADARG, "scatter_add_rows: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, idx, n, d, dst, dst_rows);
  return launch_status("scatter_add_rows");
}

extern "C" int mmamd_relu_bwd(const float* y, const float* dy, float* dz, int64_t n, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(y && dy && dz && n >= 0, MMAMD_E_BADARG, "relu_bwd: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dz, (long long)n);
  return launch_status("relu_bwd");
}
