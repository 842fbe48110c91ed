// gemm.hip — bf16 "NT" GEMM with fused epilogues on the gfx950 matrix cores.
//
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N])        A, W bf16; accumulate fp32
//
// This one kernel family is ~96 % of the hot path's FLOPs: packed QKV in-projection, attention
// out-projection (+residual), MLP up (+QuickGELU) and down (+residual), and the patch-embedding
// conv expressed as a GEMM.  Both operands are K-contiguous (torch Linear weight layout), so both
// MFMA operand fragments are 16-byte contiguous LDS reads.
//
// Structure (CDNA4-first, not a CUDA tiling):
//   * v_mfma_f32_32x32x16_bf16, operands SWAPPED: the MFMA "A" operand is the W tile (rows = n) and the
//     "B" operand is the activation tile (rows = m), so D[n][m]: each lane owns ONE output row m and
//     4 consecutive columns n per accumulator group -> bias is a float4, bf16 results pack to 8 bytes
//     and, after one v_permlane32_swap per dword, to one 16-byte store per lane (guide T21).
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round
//     trip).  The DMA destination is lane-linear, so the bank swizzle is applied to the per-lane
//     SOURCE address and undone on the ds_read_b128 side (guide rule 21).  Swizzle: a tile row is
//     128 B (BK = 64 bf16); two rows share a 256-byte bank row of 16 slots; slot' = slot ^ (bankrow & 15)
//     -> every 16-lane ds_read_b128 group hits 16 distinct slots (conflict-free).
//   * 2-stage LDS ring, ONE barrier per K-tile: the DMA for tile k+1 is issued right after the barrier
//     that retires tile k-1's reads, and is only waited for (vmcnt(0)) at the next barrier, a full
//     compute phase later.
//   * XCD-aware bijective block remap: consecutive tile ids land on ONE XCD so the blocks sharing an
//     activation row-panel / the weight matrix hit the same 4 MiB L2.
#include <type_traits>

#include "common.h"

namespace mmamd {

typedef uint32_t __attribute__((address_space(3))) * lds_u32p;
typedef const uint32_t __attribute__((address_space(1))) * glb_u32p;
typedef __attribute__((ext_vector_type(4))) int int32x4;

// LDS-DMA through a buffer descriptor: per-lane 32-bit byte offset (constant over the K loop) + a SCALAR K offset.
// The 64-bit-vaddr form (global_load_lds v[a:a+1], off) needs one v_lshl_add_u64 per piece; measured
// (tools/microbench/mfma_dma_mix.hip) that VALU traffic beside a busy matrix pipe cuts the DMA stream of a CU from
// 57 to 21 B/clk and was the reason every schedule of this kernel stalled at ~40 % MFMA utilisation.  The SRD /
// saddr forms need no VALU at all and run at the full 57 B/clk next to full-rate MFMAs.
__device__ void llvm_amdgcn_raw_buffer_load_lds(int32x4 rsrc, lds_u32p lds_ptr, int size, int voffset, int soffset,
                                                int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ int32x4 make_srd(const void* base, uint32_t bytes) {
  const uint64_t p = reinterpret_cast<uint64_t>(base);
  int32x4 r;
  r[0] = (int)(uint32_t)(p & 0xffffffffu);
  r[1] = (int)(uint32_t)((p >> 32) & 0xffffu);  // stride 0 (raw buffer)
  r[2] = (int)bytes;                              // num_records in bytes: reads past the end return 0
  r[3] = 0x00020000;
  return r;
}

struct GemmArgs {
  const bf16* A;
  const bf16* W;
  const float* bias;
  const void* R;
  void* C;
  int M, N, K;
  int lda, ldw, ldr, ldc;
  int act;
  int tiles_n;
  int res_mode;             // 0: C += R;  1 / 2: C *= QuickGELU'(R) / GELU'(R) (backward of the MLP: R = saved pre-activation, bf16)
  int kt_chunk;             // split-K (weight gradients): K-tiles (of 64) per split, blockIdx.y = split; 0 = no split
  long long c_split_stride; // elements between the partial outputs of consecutive splits
  int split_flat;           // split-K with the split index folded into blockIdx.x (1-D grid of tiles * splits, split-major): 0 = blockIdx.y
  void* C2;                 // training forward of the MLP: second bf16 output act2(bf16(C)) next to the pre-activation C (NULL = none)
  int ldc2, act2;
  // LayerNorm folded into the GEMMs either side of it (file header of the "LN fold" section below).
  // producer (fp32 residual GEMM): bf16 copy of the output rows + per-row partial (sum, sum of squares) of every 64-column block
  bf16* Xh = nullptr;
  int ldxh = 0;
  float* st_out = nullptr;  // [M][nslot_out][2]
  int nslot_out = 0;
  // consumer (A = that bf16 copy, W = gamma-scaled weight): C = rstd_m (acc - mu_m c1[n]) + bias[n], then the activation
  const float* st_in = nullptr;  // [M][nslot_in][2]
  int nslot_in = 0;
  const float* c1 = nullptr;     // [N] row sums of the gamma-scaled bf16 weight
  float inv_d = 0.f, ln_eps = 0.f;
  // persistent kernel: workgroups that walk one tile fewer than the others (the last round of tiles is partial) start up to `stagger`
  // clock ticks late, spread evenly, so that the CUs stop draining their C tiles in lock-step (0 = off)
  int stagger = 0;
  // patch-embedding mode of the persistent kernel (A_MODE = 1): A is a bf16 IMAGE [B,3,hw,hw]; the row m = b*g2 + gy*g + gx of the im2col
  // matrix and its K index (c*p + py)*p + px are resolved by the LDS-DMA source addresses (16-byte pieces = 8 pixels of one image row),
  // the output row is b*(g2+1) + 1 + (m - b*g2) (row 0 of every image is the CLS token, written elsewhere) and R = positional embedding
  // rows 1..g2 (row index (m mod g2) + 1), models/clip/image_encoder.py:91-106
  int i2c_g2 = 0, i2c_g = 0, i2c_p = 0, i2c_hw = 0;
  int i2c_lcr = 0;   // log2(16-byte chunks per patch row) = log2(p / 8)
  int i2c_ltpc = 0;  // log2(K-tiles per channel) = log2(p*p / 64)
  int i2c_rpk = 0;   // image rows per K-tile = 64 / p
};

// x * sigmoid(1.702 x) with the hardware exp2 / rcp (1 ulp each; the result is rounded to bf16 anyway).  A plain
// `/` compiles to the ~10-instruction IEEE division sequence: measured at 29 % of the MLP-up GEMM's time.
__device__ __forceinline__ float quick_gelu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
}

// nn.GELU() = 0.5 v (1 + erf(v / sqrt 2)).  libm's erff is ~60 instructions with branches (it doubled the epilogue's register
// use); this is the Abramowitz-Stegun 7.1.26 rational form on the hardware rcp / exp2: |erf error| <= 1.5e-7, i.e. the result
// differs from the exact GELU by < 1e-7 |v| -- far below the bf16 rounding of the value that is stored.  Branch-free.
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float h = 0.5f * v * (poly * t) * __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);  // 0.5 v (1 - erf|x|)
  return v >= 0.f ? v - h : h;
}

// d/du [u sigmoid(1.702 u)] and d/du [u Phi(u)]: the factor the MLP's backward multiplies the incoming gradient with
__device__ __forceinline__ float act_grad(float u, int mode) {
  if (mode == 1) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
    return sg * (1.0f + 1.702f * u * (1.0f - sg));
  }
  const float x = fabsf(u) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);  // exp(-u^2/2)
  const float tail = 0.5f * (poly * t) * ex;                               // 0.5 (1 - erf|x|)
  const float cdf = u >= 0.f ? 1.0f - tail : tail;
  return cdf + u * 0.3989422804014327f * ex;
}
__device__ __forceinline__ float combine_res(float v, float r, int mode) { return mode == 0 ? v + r : v * act_grad(r, mode); }

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == MMAMD_ACT_QUICKGELU) return quick_gelu(v);
  if (act == MMAMD_ACT_GELU_ERF) return gelu_erf(v);
  return v;
}

// second output of the dual-store GEMM: the activation of the 8 bf16 values just stored to C (row m, columns n..n+7)
__device__ __forceinline__ void store_act_copy(const GemmArgs& p, uint4 v, int m, int n) {
  if (p.C2 == nullptr) return;  // wave-uniform
  bf16x8 a8 = __builtin_bit_cast(bf16x8, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) a8[j] = (bf16)apply_act((float)a8[j], p.act2);
  *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C2) + (size_t)m * p.ldc2 + n) = __builtin_bit_cast(uint4, a8);
}

// ---------------------------------------------------------------------------------------------------------
// LN fold.  A pre-norm block computes  y = LN(x) W^T + b  with  LN(x) = (x - mu) rstd gamma + beta  per token row.  Algebra:
//     y[m][n] = rstd_m ( sum_k x[m][k] (gamma_k W[n][k])  -  mu_m sum_k gamma_k W[n][k] )  +  ( sum_k beta_k W[n][k] + b[n] )
//             = rstd_m ( acc[m][n] - mu_m c1[n] ) + c2[n]          acc = xh . W'^T,  xh = bf16(x),  W' = bf16(gamma (.) W)
// so the LayerNorm pass (a 155 MB fp32 read + 77 MB bf16 write per call at ViT-B/16, B = 256) disappears: the GEMM that PRODUCES x
// (attention out-projection / MLP down-projection with the fp32 residual) also writes xh and, per row and 64-column block, the
// partial sums (sum x, sum x^2) of the fp32 values; the GEMM that CONSUMES LN(x) reads xh as its A operand and finishes the
// statistics (mu, rstd from the N/64 partials of its rows, fixed summation order -> bit-reproducible) in its epilogue.
// Numerics: x instead of LN(x) is rounded to bf16, which scales the operand rounding error of a token by sqrt(1 + mu^2/sigma^2)
// (tools/ln_fold_numerics.py); var = E[x^2] - mu^2 in fp32.  Parity at the headline size: tests/test_gpu_headline_parity.py.
// ---------------------------------------------------------------------------------------------------------
// st_lds (FOLD == 1, persistent kernel): the statistics of the tile's rows, [BM][nslot][2] floats, DMA'd into LDS during the K loop
// bias_lds (persistent kernel, FOLD == 0): the tile's 256 bias values, DMA'd into LDS during the K loop (the eight 16-byte global loads
// per lane at the head of the epilogue were an exposed L2 round trip per tile)
template <int MI, int NI, int TM, int TN, int FOLD, int ACT, bool STLDS = false>
__device__ __forceinline__ void bias_or_lnfold(f32x16 (&acc)[NI][MI], const GemmArgs& p, int m0, int n0, int wm, int wn, int lane,
                                               const char* st_lds = nullptr, const char* bias_lds = nullptr) {
  const int l31 = lane & 31, half = lane >> 5;
  if constexpr (FOLD == 1) {
    // the 128 accumulators leave ~100 VGPRs to this code: statistics first (2 * MI live values), then one column group at a time with
    // a scheduling fence between groups (hipcc otherwise hoists the c1 / c2 loads of all 8 groups to the top: 64 more live registers)
    float rs[MI], nmr[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float s1 = 0.f, s2 = 0.f;
      if constexpr (STLDS) {
        const float* s = reinterpret_cast<const float*>(st_lds) + (size_t)(wm * TM + mi * 32 + l31) * (size_t)(2 * p.nslot_in);
        for (int k = 0; k < p.nslot_in; k += 2) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(s + 2 * k);
          s1 += v[0]; s2 += v[1];
          s1 += v[2]; s2 += v[3];
        }
      } else {
        int m = m0 + wm * TM + mi * 32 + l31;
        m = m < p.M ? m : p.M - 1;
        const float* s = p.st_in + (size_t)m * (size_t)(2 * p.nslot_in);
#pragma unroll 4
        for (int k = 0; k < p.nslot_in; k += 2) {  // nslot is even (host check); fixed order: block 0, 1, 2, ...
          const f32x4 v = load4(s + 2 * k);
          s1 += v[0]; s2 += v[1];
          s1 += v[2]; s2 += v[3];
        }
      }
      const float mean = s1 * p.inv_d;
      rs[mi] = __builtin_amdgcn_rsqf(fmaxf(fmaf(s2, p.inv_d, -mean * mean), 0.f) + p.ln_eps);
      nmr[mi] = -mean * rs[mi];  // acc' = rs acc + (-mu rs) c1 + c2
      __builtin_amdgcn_sched_barrier(0);
    }
    // one 32-column half (4 column groups) at a time: its 8 c1 / c2 loads are issued together (ONE L2 round trip per half; a fence
    // per group made it one per group: +10 % on the ViT MLP-up GEMM, 2 x on the text tower's), 32 live registers
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      f32x4 cv[4], bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * TN + ni * 32 + 4 * half + 8 * g;
        cv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n + 3 < p.N) { cv[g] = load4(p.c1 + n); bv[g] = load4(p.bias + n); }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = fmaf(rs[mi], acc[ni][mi][4 * g + j], fmaf(nmr[mi], cv[g][j], bv[g][j]));
            // QuickGELU right here (the callers skip their own pass when FOLD == 1)
            acc[ni][mi][4 * g + j] = ACT == MMAMD_ACT_QUICKGELU ? quick_gelu(v) : v;
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (p.bias != nullptr) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * TN + ni * 32 + 4 * half + 8 * g;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias_lds != nullptr) bv = *reinterpret_cast<const f32x4*>(bias_lds + (wn * TN + ni * 32 + 4 * half + 8 * g) * 4);  // block-uniform
        else if (n + 3 < p.N) bv = load4(p.bias + n);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ni][mi][4 * g + j] += bv[j];
      }
  }
}

// producer side.  CANONICAL summation order of a row's 64-column block (every epilogue reproduces it, so a row's statistics are
// bit-identical whichever kernel the dispatcher picks for a batch size): column quads q_c = (v0 + v1) + (v2 + v3), c = 0..15;
// u_c = q_c + q_(c+8), c = 0..7 (the two 32-column halves); then a butterfly over c with partners c^1, c^2, c^4.  Sum of squares alike
// with quads fma(v0, v0, v1 v1) + fma(v2, v2, v3 v3).
__device__ __forceinline__ void lnfold_store_acc(const GemmArgs& p, f32x4 v, int m, int n, bool ok, float& s1, float& s2) {
  if (ok) store4(p.Xh + (size_t)m * p.ldxh + n, v);  // bf16 copy of the 4 values
  s1 += (v[0] + v[1]) + (v[2] + v[3]);
  s2 += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
}
__device__ __forceinline__ void lnfold_butterfly8(float& a, float& b) {  // partners c^1, c^2, c^4 = lanes ^1, ^2, ^4
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
}

static int g_gemm_variant = 0;
// per cent of a tile time; measured (tools/gemm_variant_bench.py --staggers 0,30,60,90,120, profiles/r02_gemm_stagger.txt): 60 is the best or
// within 1 % of it on every shape whose last round of tiles is partial (ViT MLP-up -3.6 %, out-proj -8 %, text MLP-up -10.6 %, patch -7.6 %)
static int g_gemm_stagger = 60;
static unsigned long long* g_gemm_trace = nullptr;

// Epilogue shared by the tiled kernels.  Lane owns row m = .. + (lane&31); accumulator regs 4g..4g+3 are columns
// n = .. + 8g + 4*(lane>>5) + {0..3}.
template <int MI, int NI, int TM, int TN, bool OUT_F32, int ACT, int FOLD = 0>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[NI][MI], const GemmArgs& p, int m0, int n0, int wm, int wn,
                                              int lane) {
  const int l31 = lane & 31, half = lane >> 5;
  // pass 1: bias (depends on n only) or the folded LayerNorm.  pass 2: activation behind ONE uniform branch.  pass 3: residual + store.
  bias_or_lnfold<MI, NI, TM, TN, FOLD, ACT>(acc, p, m0, n0, wm, wn, lane);
  if constexpr (ACT == MMAMD_ACT_QUICKGELU && FOLD != 1) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[ni][mi][r];
          acc[ni][mi][r] = quick_gelu(v);
        }
  } else if constexpr (ACT == MMAMD_ACT_GELU_ERF) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[ni][mi][r];
          acc[ni][mi][r] = gelu_erf(v);
        }
  }
  const bool has_res = p.R != nullptr;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * TM + mi * 32 + l31;
    const bool mok = m < p.M;
    // LN fold (producer): the canonical order (lnfold_store_acc) in this layout: quad c = 2 g + half of 32-column half ni, so
    // u_c = ua[g] (sum over ni, in-lane), partner c^1 = the other lane half, partners c^2 / c^4 = accumulator groups g^1 / g^2
    float ua[4] = {0.f, 0.f, 0.f, 0.f}, ub[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nb = n0 + wn * TN + ni * 32 + 4 * half;
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        const bool ok = mok && (n + 3 < p.N);
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = acc[ni][mi][4 * g + j];
        if (has_res && ok) {
          f32x4 rv;
          if constexpr (OUT_F32) rv = load4(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + n);
          else rv = load4(reinterpret_cast<const bf16*>(p.R) + (size_t)m * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = OUT_F32 ? t[j] + rv[j] : combine_res(t[j], rv[j], p.res_mode);
        }
        if constexpr (OUT_F32) {
          if (ok) store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, t);
          if constexpr (FOLD == 2) {
            if (ok) store4(p.Xh + (size_t)m * p.ldxh + n, t);
            ua[g] += (t[0] + t[1]) + (t[2] + t[3]);
            ub[g] += fmaf(t[0], t[0], t[1] * t[1]) + fmaf(t[2], t[2], t[3] * t[3]);
          }
        }
        v[g] = t;
      }
      if constexpr (!OUT_F32) {
        // pack to bf16 and widen the stores: groups (g, g+1) -> one 16-byte store per lane (T21)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          bf16x4 pa, pb;
#pragma unroll
          for (int j = 0; j < 4; ++j) { pa[j] = (bf16)v[g][j]; pb[j] = (bf16)v[g + 1][j]; }
          uint2 ua = __builtin_bit_cast(uint2, pa), ub = __builtin_bit_cast(uint2, pb);
          // lanes 32-63 of `ua` <-> lanes 0-31 of `ub`
          auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
          const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          // lower half now holds columns 8g..8g+7 of its row, upper half columns 8(g+1)..8(g+1)+7
          const int n = n0 + wn * TN + ni * 32 + 8 * (g + half);
          if (mok && n + 7 < p.N) {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = o;
            store_act_copy(p, o, m, n);
          }
        }
      }
    }
    if constexpr (OUT_F32 && FOLD == 2) {
      {  // TN == 64: the wave's columns are ONE 64-column block
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ua[g] += __shfl_xor(ua[g], 32);
          ub[g] += __shfl_xor(ub[g], 32);
        }
        const float ls1 = (ua[0] + ua[1]) + (ua[2] + ua[3]), ls2 = (ub[0] + ub[1]) + (ub[2] + ub[3]);
        if (mok && half == 0) {
          const int slot = (n0 + wn * TN) >> 6;
          *reinterpret_cast<f32x2*>(p.st_out + ((size_t)m * p.nslot_out + slot) * 2) = f32x2{ls1, ls2};
        }
      }
    }
  }
}

// LDS-staged epilogue (used by the pipelined kernels).  The MFMA layout gives every lane 4 consecutive columns of
// ONE row, so a direct store instruction touches 32 different rows with 32 bytes each — measured (ablation: no epilogue)
// at 30-50 % of the kernel time.  Here every wave transposes its sub-tile through a private LDS strip, 32 rows at a
// time, and then stores/loads FULL rows: one wave-instruction covers 8 rows x 128 B (bf16) or 4 rows x 256 B (fp32),
// i.e. whole cache lines; the fp32 residual is read with the same row-contiguous pattern.
template <int MI, int NI, int TM, int TN, bool OUT_F32, int ACT, int ABL = 0, int FOLD = 0>
__device__ __forceinline__ void gemm_epilogue_lds(f32x16 (&acc)[NI][MI], const GemmArgs& p, int m0, int n0, int wm,
                                                  int wn, int lane, int wave, char* smem) {
  static_assert(TN == 64, "row strip below is laid out for 64-column wave tiles");
  const int l31 = lane & 31, half = lane >> 5;
  bias_or_lnfold<MI, NI, TM, TN, FOLD, ACT>(acc, p, m0, n0, wm, wn, lane);
  if constexpr (ACT == MMAMD_ACT_QUICKGELU && FOLD != 1) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[ni][mi][r];
          acc[ni][mi][r] = quick_gelu(v);
        }
  }
  // erf-GELU is applied where the values are packed for the strip (32x32 at a time): as one pass over all 256
  // accumulators its temporaries spilled (282 VGPRs)
  constexpr int ROWB = OUT_F32 ? (TN * 4 + 16) : (TN * 2 + 16);  // padded strip row: 272 B / 144 B (conflict-free b128)
  const int nw0 = n0 + wn * TN;
  const bool has_res = OUT_F32 && p.R != nullptr;
  f32x4 rr[8], rn[8];
  auto res_load = [&](int mi, f32x4 (&dst)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = m0 + wm * TM + mi * 32 + it * 4 + (lane >> 4), n = nw0 + (lane & 15) * 4;
      dst[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < p.M && n + 3 < p.N) dst[it] = load4(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + n);
    }
  };
  if constexpr (OUT_F32) {
    if (has_res) res_load(0, rr);  // in flight across the barrier and the first transpose
  }
  __syncthreads();  // every wave is done reading the operand stages: LDS can be reused
  char* strip = smem + wave * (32 * ROWB);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int mrow0 = m0 + wm * TM + mi * 32;
    if constexpr (OUT_F32) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 t;
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = ACT == MMAMD_ACT_GELU_ERF ? gelu_erf(acc[ni][mi][4 * g + j]) : acc[ni][mi][4 * g + j];
          *reinterpret_cast<f32x4*>(strip + l31 * ROWB + (ni * 32 + 8 * g + 4 * half) * 4) = t;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 vv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it)  // 4 rows x 256 B per wave-instruction
        vv[it] = *reinterpret_cast<const f32x4*>(strip + (it * 4 + (lane >> 4)) * ROWB + (lane & 15) * 16);
      // residual of the NEXT 32-row slab is requested before this slab is consumed: the (otherwise fully exposed)
      // HBM/L2 round trip of every slab overlaps the previous slab's adds and stores (r01 trace: 39k of 72k ticks)
      if (has_res && mi + 1 < MI) res_load(mi + 1, rn);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = mrow0 + it * 4 + (lane >> 4), n = nw0 + (lane & 15) * 4;
        const bool ok = m < p.M && n + 3 < p.N && ((ABL & 16) == 0 || vv[it][0] == 1.2345678e33f);
        f32x4 v = vv[it];
        if (ok) {
          if (has_res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rr[it][j];
          }
          store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, v);
        }
        if constexpr (FOLD == 2) {  // LN fold (producer): 16 lanes hold the 64 columns of row m
          float s1 = 0.f, s2 = 0.f;
          lnfold_store_acc(p, v, m, n, ok, s1, s2);  // lane c = lane & 15 holds quad c
          s1 += __shfl_xor(s1, 8);                   // u_c = q_c + q_(c+8)
          s2 += __shfl_xor(s2, 8);
          lnfold_butterfly8(s1, s2);
          if (m < p.M && (lane & 15) == 0)
            *reinterpret_cast<f32x2*>(p.st_out + ((size_t)m * p.nslot_out + (nw0 >> 6)) * 2) = f32x2{s1, s2};
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) rr[it] = rn[it];
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          bf16x4 pa, pb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float va = acc[ni][mi][4 * g + j], vb = acc[ni][mi][4 * (g + 1) + j];
            if constexpr (ACT == MMAMD_ACT_GELU_ERF) { va = gelu_erf(va); vb = gelu_erf(vb); }
            pa[j] = (bf16)va; pb[j] = (bf16)vb;
          }
          uint2 ua = __builtin_bit_cast(uint2, pa), ub = __builtin_bit_cast(uint2, pb);
          auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
          // lower half: columns 8g..8g+7 of its row; upper half: columns 8(g+1)..8(g+1)+7
          *reinterpret_cast<uint4*>(strip + l31 * ROWB + (ni * 32 + 8 * (g + half)) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {  // 8 rows x 128 B per wave-instruction
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        uint4 v = *reinterpret_cast<const uint4*>(strip + row * ROWB + c * 16);
        const int m = mrow0 + row, n = nw0 + c * 8;
        if (m < p.M && n + 7 < p.N && ((ABL & 16) == 0 || v.x == 0x12345678u)) {
          if (p.R != nullptr) {
            const uint4 rr = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.R) + (size_t)m * p.ldr + n);
            bf16x8 a8 = __builtin_bit_cast(bf16x8, v), r8 = __builtin_bit_cast(bf16x8, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = (bf16)combine_res((float)a8[j], (float)r8[j], p.res_mode);
            v = __builtin_bit_cast(uint4, a8);
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = v;
          store_act_copy(p, v, m, n);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip reads done before the next 32 rows overwrite it
  }
}

// LDS-DMA piece through inline asm: 1 KiB (64 lanes x 16 B) from per-lane global addresses to the wave-uniform LDS
// byte address `lds_dst`.  hipcc does not model it (no LDS-alias drain of lgkmcnt before it, no vmcnt bookkeeping):
// completion is counted by hand with s_waitcnt vmcnt(N).  M0 is saved/restored inside the statement (guide 5.7).
__device__ __forceinline__ void dma_piece(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Same piece with the address split as SCALAR base (SGPR pair) + per-lane 32-bit byte offset: no VALU per piece.
__device__ __forceinline__ void dma_piece_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  // M0 is written and consumed inside the statement and NOT restored: nothing else in these kernels uses M0 (no LDS-DMA
  // builtin, no s_movrel, no GWS), so the guide's save/restore pair (2 of 5 scalar instructions per piece) is dropped.
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// 16-byte global store with an explicit cache policy.  POLICY 0: plain (line stays in the XCD's L2); 1: sc1 (written
// through and dropped from L2 - the C tile is never re-read by this kernel, so it should not evict operand panels);
// 2: nt.  The trailing s_nop covers the data-register hazard of an asm store (guide 5.7).
template <int POLICY>
__device__ __forceinline__ void store16(void* ptr, uint4 v) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  const u32x4_t w = __builtin_bit_cast(u32x4_t, v);
  if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(w) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(ptr), "v"(w) : "memory");
  else *reinterpret_cast<uint4*>(ptr) = v;
}

// BM x BN block tile, WM x WN waves, BK = 64
template <int BM, int BN, int WM, int WN, bool OUT_F32, int ACT, bool SGB, int FOLD = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // 1-KiB DMA pieces per wave per stage
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  static_assert(NW % 4 == 0, "swizzle phase below assumes the wave count is a multiple of 4");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- block -> tile, XCD-aware (block b runs on XCD b % 8: give each XCD a contiguous id range)
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  // ---- DMA source offsets (bytes from A / W) for this lane; piece i = wave + NW*j covers tile rows 8i..8i+7
  // LDS position of lane: bank row Rr = 4i + (lane>>4), slot' = lane&15; it must hold slot = slot' ^ (Rr&15)
  const int sw = (4 * (wave & 3) + (lane >> 4)) & 15;
  const int slot = (lane & 15) ^ sw;
  const int row8 = 2 * (lane >> 4) + (slot >> 3);  // row inside the 8-row piece
  const int chunk = slot & 7;                      // 16-byte chunk inside the 128-byte row
  uint32_t a_off[A_INSTR], b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    int r = m0 + 8 * (wave + NW * j) + row8;
    r = r < p.M ? r : p.M - 1;
    a_off[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    int r = n0 + 8 * (wave + NW * j) + row8;
    r = r < p.N ? r : p.N - 1;
    b_off[j] = ((uint32_t)r * (uint32_t)p.ldw + chunk * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  auto issue_stage = [&](int buf, int kt) __attribute__((always_inline)) {  // scalar base + K offset, per-lane 32-bit offset: no VALU per piece
    const uint32_t dst = lds0 + buf * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) dma_piece_s(Ab + (size_t)kt * 128, a_off[j], dst + NW * j * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) dma_piece_s(Wb + (size_t)kt * 128, b_off[j], dst + A_BYTES + NW * j * 1024);
  };

  // ---- fragment read offsets: lane reads row (lane&31) of a 32-row block, 16-byte chunk 2t + (lane>>5)
  const int l31 = lane & 31, half = lane >> 5;
  const int hsw = l31 >> 1;
  int roff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) roff[t] = hsw * 256 + (((((l31 & 1) << 3) | (2 * t + half)) ^ hsw) << 4);

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int KT = p.K >> 6;
  issue_stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of tile kt have landed
    __syncthreads();                                   // ... everyone's; and tile kt-1's reads are done
    if (kt + 1 < KT) issue_stage((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE + (wm * TM) * 128;
    const char* sb = smem + (kt & 1) * STAGE + A_BYTES + (wn * TN) * 128;
    // register double buffer: fragments of k-step t+1 are in flight while the MFMAs of k-step t issue
    bf16x8 xa[2][MI], wb[2][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wb[0][ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 32 * 128 + roff[0]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xa[0][mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 32 * 128 + roff[0]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      if (t < 3) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          wb[nxt][ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 32 * 128 + roff[t + 1]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          xa[nxt][mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 32 * 128 + roff[t + 1]);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[cur][ni], xa[cur][mi], acc[ni][mi], 0, 0, 0);
    }
    if constexpr (SGB) {
      // pin the software pipeline hipcc otherwise collapses (it re-uses the fragment registers and issues
      // every ds_read AFTER the MFMAs of the step): fragments(t=0); then per k-step one ds_read of step t+1
      // behind each of the first NI+MI MFMAs of step t.   masks: MFMA = 0x008, DS read = 0x100
      constexpr int NF = NI + MI, NM = NI * MI;
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    }
  }

  gemm_epilogue<MI, NI, TM, TN, OUT_F32, ACT, FOLD>(acc, p, m0, n0, wm, wn, lane);
}


// ---------------------------------------------------------------------------------------------------------
// Pipelined kernel ("P"): same tile geometry / LDS image / epilogue as above, different schedule.
//   * the K loop is ROTATED across the barrier: the 4th k-step's MFMAs of tile k are issued AFTER the barrier
//     that publishes tile k+1, so they cover the barrier release, the LDS-DMA issue for tile k+2 and the
//     latency of the first fragment reads of tile k+1 (in the plain loop the matrix pipe idles through all
//     three on every K-tile: SQ_WAIT_ANY was 35 % of wave cycles there);
//   * the 8 DMA pieces of the next tile are interleaved one-per-MFMA instead of issued as a burst;
//   * block -> tile order is grouped (GM row-panels x all column tiles per group) inside each XCD's contiguous
//     id range, so the ~32 blocks an XCD runs concurrently share GM activation panels and ~32/GM weight tiles
//     in its 4 MiB L2 (the row-major order re-fetched the whole weight matrix every 32 blocks: FETCH_SIZE was
//     3-6x the algorithmic bytes).
// ABL (ablation bit mask, perf experiments only — results are WRONG for ABL != 0): 1 = no DMA in the loop,
// 2 = no MFMA, 4 = no epilogue, 8 = no fragment reads, 16 = no global accesses in the epilogue
// TNM ("TN" operands, weight gradients): C[M,N] = sum_t A[t][m] W[t][n] with A = [K, lda] and W = [K, ldw] ROW-major over the
// contraction index t (dW = dY^T X straight from the row-major dY and X: no transposed bf16 copies in HBM).  The LDS image of a
// K-tile is then [64 t][256 cols], stored as 256-byte units of [4 t][32 cols] (two [4][16] blocks) in [t/4][cols/32] order — the
// DMA lays it out through its per-lane source addresses — and every MFMA operand is two ds_read_b64_tr_b16 (4 + 4 contraction
// indices of one column per lane; the two 16-lane groups of a half-wave read one contiguous 256-byte unit: conflict-free).
template <int BM, int BN, int WM, int WN, bool OUT_F32, int ACT, int GM, int ABL = 0, bool LDSEPI = true, bool TNM = false, int SCH = 0, int FOLD = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel_p(const GemmArgs p, const int tiles_m,
                                                                        unsigned long long* trace = nullptr) {
  static_assert(!TNM || (BM == 256 && BN == 256), "TN image below is laid out for 256-column operand tiles");
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  constexpr int NF = NI + MI, NM = NI * MI, NDMA = A_INSTR + B_INSTR;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 4 == 0, "tile/wave geometry");
  static_assert(NM >= NF && NM >= NDMA, "interleave below needs one MFMA per fragment read / DMA piece");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // split-K, flat form: the XCD-contiguous id enumerates (split, tile) split-major, so the ~32 workgroups an XCD runs are ONE split's
  // neighbouring tiles: they stream the same contraction rows at the same time and share the operand panels in that XCD's L2
  int split = (int)blockIdx.y;
  if (p.split_flat) {
    const int ntile = tiles_m * p.tiles_n;
    split = bid / ntile;
    bid -= split * ntile;
  }
  int tm, tn;
  {
    const int per_group = GM * p.tiles_n;
    const int grp = bid / per_group, within = bid - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  const int sw = (4 * (wave & 3) + (lane >> 4)) & 15;
  const int slot = (lane & 15) ^ sw;
  const int row8 = 2 * (lane >> 4) + (slot >> 3);
  const int chunk = slot & 7;
  uint32_t a_off[A_INSTR], b_off[B_INSTR];
  if constexpr (TNM) {
    // piece pi (1 KiB = 4 units) holds contraction rows 4*(pi>>1)..+3, columns 128*(pi&1)..+127: lane -> unit lane>>4, block
    // (lane>>3)&1, row (lane>>1)&3, 16-byte half of the block row lane&1.  Columns past the matrix are clamped (they only feed
    // rows / columns of C that are never stored).
    const int pu = lane >> 4, pcb = (lane >> 3) & 1, prow = (lane >> 1) & 3, phr = lane & 1;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int pi = wave + NW * j;
      int c = m0 + (4 * (pi & 1) + pu) * 32 + pcb * 16 + phr * 8;
      c = c + 8 <= p.M ? c : p.M - 8;
      a_off[j] = ((uint32_t)(4 * (pi >> 1) + prow) * (uint32_t)p.lda + c) * 2u;
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const int pi = wave + NW * j;
      int c = n0 + (4 * (pi & 1) + pu) * 32 + pcb * 16 + phr * 8;
      c = c + 8 <= p.N ? c : p.N - 8;
      b_off[j] = ((uint32_t)(4 * (pi >> 1) + prow) * (uint32_t)p.ldw + c) * 2u;
    }
  } else {
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      int r = m0 + 8 * (wave + NW * j) + row8;
      r = r < p.M ? r : p.M - 1;
      a_off[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      int r = n0 + 8 * (wave + NW * j) + row8;
      r = r < p.N ? r : p.N - 1;
      b_off[j] = ((uint32_t)r * (uint32_t)p.ldw + chunk * 8) * 2u;
    }
  }
  // split-K: this block owns K-tiles [kt0, kt0 + KT) and writes its own partial output (no bias / residual / activation)
  const int kt0 = p.kt_chunk > 0 ? split * p.kt_chunk : 0;
  const size_t a_step = TNM ? (size_t)64 * p.lda * 2 : 128, w_step = TNM ? (size_t)64 * p.ldw * 2 : 128;  // bytes per K-tile
  const char* Ab = reinterpret_cast<const char*>(p.A) + (size_t)kt0 * a_step;
  const char* Wb = reinterpret_cast<const char*>(p.W) + (size_t)kt0 * w_step;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  // piece i of a stage (i < A_INSTR: activation rows, else weight rows): scalar base + K offset, per-lane 32-bit offset
  auto issue_piece = [&](int buf, int kt, int i) __attribute__((always_inline)) {
    const uint32_t dst = lds0 + buf * STAGE + (i < A_INSTR ? (wave + NW * i) * 1024 : A_BYTES + (wave + NW * (i - A_INSTR)) * 1024);
    if (i < A_INSTR) dma_piece_s(Ab + (size_t)kt * a_step, a_off[i], dst);
    else dma_piece_s(Wb + (size_t)kt * w_step, b_off[i - A_INSTR], dst);
  };
  auto issue_stage = [&](int buf, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) issue_piece(buf, kt, i);
  };

  const int l31 = lane & 31, half = lane >> 5;
  const int hsw = l31 >> 1;
  // per-lane LDS byte addresses of the fragment reads, fully precomputed per (buffer, k-step): the K loop is unrolled by
  // two so the buffer is a compile-time constant and NO address VALU is left inside the loop (every VALU instruction
  // beside the MFMA stream waits for an issue gap of the matrix pipe; the 8 v_add per K-tile cost ~25 % of the loop)
  uint32_t ra[2][4], rb[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (TNM) {
        // k-step t = contraction rows 16t..16t+15: this lane's slots 0-3 are rows 16t + 4*half + {0..3} (unit row 4t + half),
        // slots 4-7 the same + 8 (unit row + 2, i.e. + 4096 bytes); 32-column block b of the operand tile = unit column b
        const uint32_t ro = (uint32_t)((4 * t + half) * 8) * 256 + ((lane >> 4) & 1) * 128 + ((lane & 15) >> 2) * 32 + (lane & 3) * 8;
        ra[bf][t] = (uint32_t)(uintptr_t)(lds_u32p)smem + bf * STAGE + (wm * TM / 32) * 256 + ro;
        rb[bf][t] = (uint32_t)(uintptr_t)(lds_u32p)smem + bf * STAGE + A_BYTES + (wn * TN / 32) * 256 + ro;
      } else {
        const uint32_t ro = hsw * 256 + (((((l31 & 1) << 3) | (2 * t + half)) ^ hsw) << 4);
        ra[bf][t] = (uint32_t)(uintptr_t)(lds_u32p)smem + bf * STAGE + (wm * TM) * 128 + ro;
        rb[bf][t] = (uint32_t)(uintptr_t)(lds_u32p)smem + bf * STAGE + A_BYTES + (wn * TN) * 128 + ro;
      }
    }
  typedef __attribute__((address_space(3))) const bf16x8* lds_frag_p;
  typedef __attribute__((ext_vector_type(4))) short s16x4_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_p;
  auto tr_frag = [&](uint32_t addr) -> bf16x8 {  // two transpose reads: contraction rows r..r+3 and r+8..r+11 of this lane's column
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_p>((uintptr_t)addr));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_p>((uintptr_t)(addr + 4096)));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  bf16x8 xa0[MI], wb0[NI], xa1[MI], wb1[NI];  // two fragment sets, statically named (no runtime indexing)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 8; ++j) xa1[mi][j] = (bf16)0.f;  // first rotated MFMA group multiplies zeros
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 8; ++j) wb1[ni][j] = (bf16)0.f;

  auto load_frags = [&](auto bufc, int t, bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
    if constexpr ((ABL & 8) != 0) return;
    constexpr int BF = decltype(bufc)::value;
    if constexpr (TNM) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wb[ni] = tr_frag(rb[BF][t] + ni * 256);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xa[mi] = tr_frag(ra[BF][t] + mi * 256);
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wb[ni] = *reinterpret_cast<lds_frag_p>((uintptr_t)(rb[BF][t] + ni * 32 * 128));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<lds_frag_p>((uintptr_t)(ra[BF][t] + mi * 32 * 128));
    }
  };
  auto mma = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
    if constexpr ((ABL & 2) != 0) {  // keep the fragment reads alive without the matrix work
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wb[ni]));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(xa[mi]));
      return;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
  };
  // one K-tile between two barriers.  On entry: tile kt is visible in LDS, (xa1, wb1) hold the LAST k-step of
  // tile kt-1 (or zeros).  On exit: (xa1, wb1) hold the last k-step of tile kt, everything else is consumed.
  auto mma_one = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI], int i) {
    acc[i / MI][i % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i / MI], xa[i % MI], acc[i / MI][i % MI], 0, 0, 0);
  };
  auto tile_body = [&](int kt, auto bufc, const bool has_next) __attribute__((always_inline)) {
    constexpr int BF = decltype(bufc)::value;
    const bool NEXT = has_next && (ABL & 1) == 0;
    // segment 1 (hand-ordered; the asm DMA is invisible to sched_group_barrier): this tile's first fragments, then the
    // previous tile's last k-step with one DMA piece of tile kt+1 behind every MFMA
    load_frags(bufc, 0, xa0, wb0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      mma_one(xa1, wb1, i);
      if (NEXT && i < NDMA) issue_piece(BF ^ 1, kt + 1, i);  // wave-uniform scalar branch
      __builtin_amdgcn_sched_barrier(0);
    }
    // segment 2 (compiler-scheduled under the pattern below): k-steps 0..2 with the next step's reads interleaved
    load_frags(bufc, 1, xa1, wb1);
    mma(xa0, wb0);
    load_frags(bufc, 2, xa0, wb0);
    mma(xa1, wb1);
    load_frags(bufc, 3, xa1, wb1);
    mma(xa0, wb0);
    if constexpr (ABL != 0 || SCH == 2) return;  // ablations / experiment: leave the order to the compiler
    if constexpr (SCH == 1) {  // experiment: all fragment reads of the next k-step in one burst, then the MFMAs
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x100, (TNM ? 2 : 1) * NF, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      }
      return;
    }
    if constexpr (SCH == 3) {  // experiment: reads first, in pairs, each pair followed by one MFMA
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x100, TNM ? 2 : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if constexpr (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TNM ? 2 : 1, 0);
      }
      if constexpr (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
    }
  };
  int tix = 0;
  unsigned long long* tr = nullptr;
  if constexpr ((ABL & 64) != 0) {
    if (trace != nullptr && blockIdx.x < 64 && (wave & 3) == 0) tr = trace + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 256;
  }
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr ((ABL & 64) != 0) {
      if (tr != nullptr && tix < 255) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) tr[1 + tix] = t;
        ++tix;
      }
    }
  };
  stamp();
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  auto sync_tile = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the tile have landed
    __syncthreads();                                   // ... everyone's; and the previous tile's reads are done
  };

  if constexpr ((ABL & 32) != 0) {
    // experiment: break the lock-step of the first wave of blocks (all CUs otherwise hit their store bursts together)
    if (blockIdx.x < 256) {
      const int phase = (blockIdx.x >> 3) & 3;  // 4 phases inside every XCD
      for (int i = 0; i < phase * 2; ++i) __builtin_amdgcn_s_sleep(127);  // 127*64 cycles ~ 3.4 us each -> ~T/4 per phase
    }
  }
  // even (checked by the launcher): two tiles per trip, the buffer index is a compile-time constant
  const int KT = p.kt_chunk > 0 ? ((p.K >> 6) - kt0 < p.kt_chunk ? (p.K >> 6) - kt0 : p.kt_chunk) : (p.K >> 6);
  issue_stage(0, 0);
#pragma unroll 1
  for (int kt = 0; kt < KT; kt += 2) {
    sync_tile();
    stamp();
    tile_body(kt, B0{}, true);
    sync_tile();
    stamp();
    tile_body(kt + 1, B1{}, kt + 2 < KT);
  }
  stamp();
  mma(xa1, wb1);

  if constexpr ((ABL & 4) != 0) {
    float ssum = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) ssum += acc[ni][mi][r];
    if (ssum == 1.2345678e33f) reinterpret_cast<float*>(p.C)[0] = ssum;
    return;
  }
  if constexpr (FOLD != 0) {  // never split-K: the kernel argument is used as is (a local copy of the enlarged struct is not promoted to SGPRs)
    if constexpr (LDSEPI) gemm_epilogue_lds<MI, NI, TM, TN, OUT_F32, ACT, ABL, FOLD>(acc, p, m0, n0, wm, wn, lane, wave, smem);
    else gemm_epilogue<MI, NI, TM, TN, OUT_F32, ACT, FOLD>(acc, p, m0, n0, wm, wn, lane);
  } else {
    GemmArgs pe = p;
    if (p.kt_chunk > 0) pe.C = reinterpret_cast<float*>(p.C) + (size_t)split * (size_t)p.c_split_stride;
    if constexpr (LDSEPI) gemm_epilogue_lds<MI, NI, TM, TN, OUT_F32, ACT, ABL, 0>(acc, pe, m0, n0, wm, wn, lane, wave, smem);
    else gemm_epilogue<MI, NI, TM, TN, OUT_F32, ACT, 0>(acc, pe, m0, n0, wm, wn, lane);
  }
  if constexpr ((ABL & 64) != 0) {
    stamp();                                           // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores retired
    stamp();
    if (tr != nullptr && lane == 0) tr[0] = (unsigned long long)tix;
  }
}



#ifdef MMAMD_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------
// Ping-pong kernel ("G"): 256 x 256 tile, BK = 64, 2-stage LDS ring, 8 waves (2 x 4), 128 x 64 per wave — same tile, LDS image,
// DMA pieces and epilogue as "P", different time structure.
//
// Why: in "P" every wave interleaves its fragment reads and DMA issue 1:1 with its own MFMAs, and the two waves of a SIMD
// do the same thing at the same time: the ablations put MFMA + DMA at 1613 TF/s-equivalent and MFMA + DMA + ds_read at 1190 —
// the reads steal issue slots from the matrix pipe.  Here the two waves of a SIMD (wave w and w + 4, i.e. wm = 0 / wm = 1) run
// ONE BARRIER APART: while one group issues an uninterrupted burst of 8 MFMAs at raised priority, the other one issues its
// ds_reads and DMA pieces, then they swap.  Each K-tile is four phases (one 64 x 32 quadrant of the wave tile x all of K = 64):
//
//     phase : LOAD  { fragment reads of this phase; 2 DMA pieces }  s_barrier  MFMA { 8 x mfma_32x32x16 }  s_barrier
//     quadrants (m-half, n-block): (0,0) (0,1) (1,1) (1,0)  ->  A fragments are read in phases 0 and 2, W fragments in 0 and 1
//                                                               (both W sets stay in registers; phase 3 reads nothing)
//
// Because the W units of a stage are dead after phase 1 and the A units after phase 2, the DMA stream runs a whole K-tile
// ahead inside a 2-stage ring.  Units (2 pieces per wave each): W-lo/W-hi = weight rows 0-127 / 128-255 of the tile,
// A-lo/A-hi = activation rows 0-127 (read only by group 0) / 128-255 (only by group 1).  Issue schedule, tile t in stage t&1:
//     phase 0(t): W-hi(t+1)   phase 1(t): A-lo(t+1)   phase 2(t): A-hi(t+1)   phase 3(t): W-lo(t+2)
// Slots (= barrier intervals): group 0 reads tile t in slots 8t, 8t+2, 8t+4 and group 1 in 8t+1, 8t+3, 8t+5; a read issued in
// slot L is retired by its wave's lgkmcnt(0) at the start of slot L+1, i.e. before the barrier that ends L+1: a unit may be
// restaged from slot L+2 on.  W (last read 8t+3) -> restaged in slots 8t+6.. (phase 3) ok; A-lo (8t+4) -> 8t+10..; A-hi (8t+5)
// -> 8t+12.. ok.
// MEASURED (MI355X, qkv shape, TF/s-equivalent): full 1007 (P 1043, PP 1088); without epilogue 1197 (P 1320); MFMA + barriers 1758;
// MFMA + reads 1345; MFMA + DMA 1625; loads only 1932.  The fragment reads cost the same ~25 % as in P even though another wave
// issues them: not an issue-slot effect.  Kept as an experiment (variants 30/31/34-38).
// RAW: every wave waits `vmcnt(2)` (its two youngest pieces = W-lo(t+2) may stay in flight) at the end of
// slot 8t+7 — the end of the MFMA section of phase 3 for group 0, of the LOAD section of phase 3 for group 1 — and tile t+1 is
// first read in slot 8t+8, behind the barrier.  Never vmcnt(0) in the loop except at the tail where phase 3 issues nothing.
template <bool OUT_F32, int ACT, int GM, bool TRACE = false, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_nt_kernel_g(const GemmArgs p, const int tiles_m, unsigned long long* trace) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;  // 128 x 64 per wave: MI 4, NI 2
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;             // 32 KiB + 32 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tm, tn;
  {
    const int per_group = GM * p.tiles_n;
    const int grp = bid / per_group, within = bid - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm is also the ping-pong group: waves w and w+4 share a SIMD

  // DMA source offsets (same image as "P": bank swizzle applied on the source side)
  const int sw = (4 * (wave & 3) + (lane >> 4)) & 15;
  const int slot = (lane & 15) ^ sw;
  const int row8 = 2 * (lane >> 4) + (slot >> 3);
  const int chunk = slot & 7;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int r = m0 + 8 * (wave + NW * j) + row8;
    r = r < p.M ? r : p.M - 1;
    a_off[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
    int c = n0 + 8 * (wave + NW * j) + row8;
    c = c < p.N ? c : p.N - 1;
    b_off[j] = ((uint32_t)c * (uint32_t)p.ldw + chunk * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  // unit u of K-tile kt into stage buf: 0 = A-lo, 1 = A-hi, 2 = W-lo, 3 = W-hi (pieces j = 2(u&1), 2(u&1)+1 of the operand)
  auto issue_unit = [&](int buf, int kt, int u) __attribute__((always_inline)) {
    if constexpr ((ABL & 1) != 0) { if (kt > 0) return; }  // ablation: no DMA after the prologue
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = 2 * (u & 1) + e;
      if (u < 2) dma_piece_s(Ab + (size_t)kt * 128, a_off[j], lds0 + buf * STAGE + (wave + NW * j) * 1024);
      else dma_piece_s(Wb + (size_t)kt * 128, b_off[j], lds0 + buf * STAGE + A_BYTES + (wave + NW * j) * 1024);
    }
  };

  const int l31 = lane & 31, half = lane >> 5;
  const int hsw = l31 >> 1;
  uint32_t ra[2][4], rb[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t ro = hsw * 256 + (((((l31 & 1) << 3) | (2 * t + half)) ^ hsw) << 4);
      ra[bf][t] = lds0 + bf * STAGE + (wm * TM) * 128 + ro;
      rb[bf][t] = lds0 + bf * STAGE + A_BYTES + (wn * TN) * 128 + ro;
    }
  typedef __attribute__((address_space(3))) const bf16x8* lds_frag_p;

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  bf16x8 xa[2][4], w0[4], w1[4];  // A fragments of the current m-half (2 row blocks x 4 k-steps); both W column blocks

  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  const int KT = p.K >> 6;  // even (checked by the launcher)
  int tix = 0;
  unsigned long long* tr = nullptr;
  if constexpr (TRACE) {
    if (trace != nullptr && blockIdx.x < 64 && (wave & 3) == 0) tr = trace + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 256;
  }
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (TRACE) {
      if (tr != nullptr && tix < 255) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) tr[1 + tix] = t;
        ++tix;
      }
    }
  };

  // one phase.  BF: stage of the current K-tile (compile-time), PH: phase 0..3, kt: current K-tile
  auto phase = [&](auto bufc, auto phc, int kt) __attribute__((always_inline)) {
    constexpr int BF = decltype(bufc)::value, PH = decltype(phc)::value;
    constexpr int MH = (PH >= 2) ? 1 : 0;             // m-half of the wave tile
    constexpr int NB = (PH == 1 || PH == 2) ? 1 : 0;  // n-block
    // ---- LOAD section
    if constexpr (PH == 0 && (ABL & 8) == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) w0[t] = *reinterpret_cast<lds_frag_p>((uintptr_t)(rb[BF][t]));
    }
    if constexpr (PH == 1 && (ABL & 8) == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) w1[t] = *reinterpret_cast<lds_frag_p>((uintptr_t)(rb[BF][t] + 32 * 128));
    }
    if constexpr ((PH == 0 || PH == 2) && (ABL & 8) == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) xa[mi][t] = *reinterpret_cast<lds_frag_p>((uintptr_t)(ra[BF][t] + (2 * MH + mi) * 32 * 128));
    }
    __builtin_amdgcn_sched_barrier(0);
    bool issued_ahead = true;
    if constexpr (PH == 0) { if (kt + 1 < KT) issue_unit(BF ^ 1, kt + 1, 3); }
    if constexpr (PH == 1) { if (kt + 1 < KT) issue_unit(BF ^ 1, kt + 1, 0); }
    if constexpr (PH == 2) { if (kt + 1 < KT) issue_unit(BF ^ 1, kt + 1, 1); }
    if constexpr (PH == 3) {
      issued_ahead = kt + 2 < KT;
      if (issued_ahead) issue_unit(BF, kt + 2, 2);
      if (wm == 1) {  // group 1: this LOAD section is slot 8t+7
        if (issued_ahead) __builtin_amdgcn_s_waitcnt(0x0F72 | 0x0000);  // vmcnt(2)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
      }
    }
    stamp();
    bar();
    stamp();
    // ---- MFMA section
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this phase's (and older) fragment reads
    stamp();
    __builtin_amdgcn_s_setprio(1);
    if constexpr ((ABL & 2) == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[NB][2 * MH + mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(NB ? w1[t] : w0[t], xa[mi][t], acc[NB][2 * MH + mi], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) { asm volatile("" ::"v"(NB ? w1[t] : w0[t])); asm volatile("" ::"v"(xa[0][t])); asm volatile("" ::"v"(xa[1][t])); }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (PH == 3) {
      if (wm == 0) {  // group 0: this MFMA section is slot 8t+7
        if (issued_ahead) __builtin_amdgcn_s_waitcnt(0x0F72);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
      }
    }
    stamp();
    bar();
    stamp();
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;

  stamp();
  // prologue: all of tile 0, W-lo of tile 1
  issue_unit(0, 0, 2);
  issue_unit(0, 0, 3);
  issue_unit(0, 0, 0);
  issue_unit(0, 0, 1);
  if (KT > 1) {
    issue_unit(1, 1, 2);
    __builtin_amdgcn_s_waitcnt(0x0F72);
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  bar();
  if (wm == 1) bar();  // group 1 runs one slot behind
#pragma unroll 1
  for (int kt = 0; kt < KT; kt += 2) {
    phase(B0{}, P0{}, kt);
    phase(B0{}, P1{}, kt);
    phase(B0{}, P2{}, kt);
    phase(B0{}, P3{}, kt);
    phase(B1{}, P0{}, kt + 1);
    phase(B1{}, P1{}, kt + 1);
    phase(B1{}, P2{}, kt + 1);
    phase(B1{}, P3{}, kt + 1);
  }
  if (wm == 0) bar();  // barrier counts of the two groups match again
  stamp();
  if constexpr ((ABL & 4) != 0) {
    float ssum = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) ssum += acc[ni][mi][r];
    if (ssum == 1.2345678e33f) reinterpret_cast<float*>(p.C)[0] = ssum;
    return;
  }
  gemm_epilogue_lds<MI, NI, TM, TN, OUT_F32, ACT>(acc, p, m0, n0, wm, wn, lane, wave, smem);
  if constexpr (TRACE) {
    stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    if (tr != nullptr && lane == 0) tr[0] = (unsigned long long)tix;
  }
}

#endif  // MMAMD_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------
// Deep-ring kernel ("Q"): 256 x 256 tile, BK = 32, FOUR-stage LDS ring (4 x 32 KiB), 8 waves.
// Why: with a 2-stage ring the DMA for K-tile k+1 is issued at the barrier of tile k and drained (vmcnt(0)) at the
// barrier of tile k+1, so at most one tile is in flight and the L2->LDS stream stops between tiles; measured, the
// load skeleton alone (no MFMA) ran at ~24 B/clk/CU and barely overlapped the matrix work.  Here three 32-wide
// stages (96 KiB per CU) are ALWAYS in flight: stage s+3 is issued right after the barrier that retires stage s-1,
// and the wait before the next barrier is a COUNTED vmcnt (8 = the pieces of the two younger stages), never 0.
// One raw s_barrier per stage; fragment reads are software-pipelined one k16-step deep ACROSS the barrier.
// Stage image: tile row = 64 B (4 chunks of 16 B), 4 rows per 256-B bank row, slot' = slot ^ (bankrow & 3)
// (conflict-free for ds_read_b128 lane groups; applied on the DMA source address, undone on the read).
template <bool OUT_F32, int ACT, int GM, int SCHED = 0>
__global__ __launch_bounds__(512) void gemm_bf16_nt_kernel_q(const GemmArgs p, const int tiles_m) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;  // 128 x 64 per wave: MI 4, NI 2
  constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;               // 16 KiB + 16 KiB
  constexpr int NF = NI + MI, NM = NI * MI;                              // 6 fragment reads, 8 MFMAs per k16-step
  extern __shared__ __attribute__((aligned(16))) char smem[];            // 4 * STAGE = 128 KiB

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tm, tn;
  {
    const int per_group = GM * p.tiles_n;
    const int grp = bid / per_group, within = bid - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  // DMA: a 1-KiB piece = 16 tile rows; piece i of the A (or W) half covers rows 16i..16i+15; wave w moves pieces
  // w and w+8 of each half.  LDS position of lane: bank row 4i + (lane>>4), slot' = lane&15.
  const int slot = (lane & 15) ^ (lane >> 4);          // (bank row & 3) == lane>>4
  const int row16 = 4 * (lane >> 4) + (slot >> 2);     // row inside the 16-row piece
  const int chunk = slot & 3;                          // 16-byte chunk inside the 64-byte row
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int r = m0 + 16 * (wave + NW * j) + row16;
    r = r < p.M ? r : p.M - 1;
    a_off[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
    int rn = n0 + 16 * (wave + NW * j) + row16;
    rn = rn < p.N ? rn : p.N - 1;
    b_off[j] = ((uint32_t)rn * (uint32_t)p.ldw + chunk * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  auto issue_stage = [&](int st) __attribute__((always_inline)) {
    char* sbase = smem + (st & 3) * STAGE;
    const uint32_t kbytes = (uint32_t)st * 64u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((glb_u32p)(Ab + a_off[j] + kbytes), (lds_u32p)(sbase + (wave + NW * j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((glb_u32p)(Wb + b_off[j] + kbytes),
                                       (lds_u32p)(sbase + A_BYTES + (wave + NW * j) * 1024), 16, 0, 0);
  };

  // fragment read: row l31 of a 32-row block (8 bank rows), chunk 2t + half; bank row & 3 == (l31 >> 2) & 3
  const int l31 = lane & 31, half = lane >> 5;
  int roff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    roff[t] = (l31 >> 2) * 256 + (((((l31 & 3) << 2) | (2 * t + half)) ^ ((l31 >> 2) & 3)) << 4);

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  bf16x8 xa0[MI], wb0[NI], xa1[MI], wb1[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 8; ++j) xa1[mi][j] = (bf16)0.f;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < 8; ++j) wb1[ni][j] = (bf16)0.f;

  auto load_frags = [&](const char* sa, const char* sb, int t, bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wb[ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 32 * 64 + roff[t]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 32 * 64 + roff[t]);
  };
  auto mma = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
  };
  // one 32-wide stage between two barriers.  On entry (xa1, wb1) = 2nd k16-step of the previous stage (or zeros).
  auto mma_range = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI], int i0, int i1) {
#pragma unroll
    for (int i = 0; i < NM; ++i)
      if (i >= i0 && i < i1)
        acc[i / MI][i % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i / MI], xa[i % MI], acc[i / MI][i % MI], 0, 0, 0);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  auto stage_body = [&](int st, auto issue_next) __attribute__((always_inline)) {
    const char* sa = smem + (st & 3) * STAGE + (wm * TM) * 64;
    const char* sb = smem + (st & 3) * STAGE + A_BYTES + (wn * TN) * 64;
    if constexpr (SCHED == 1) {
      // k16-step = [1 MFMA | all 6 fragment reads of the NEXT step (front-loaded: 7 MFMAs of cover) | MFMA | DMA piece |
      // 3 MFMA | DMA piece | 3 MFMA]; the 4 DMA pieces of stage st+3 are spread over the whole stage instead of bursting
      constexpr bool ISS = decltype(issue_next)::value;
      const uint32_t dst = lds0 + ((st + 3) & 3) * STAGE + wave * 1024;
      const uint32_t kb = (uint32_t)(st + 3) * 64u;
      mma_range(xa1, wb1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);  // the wait for THIS step's fragments stays in front of the new reads
      load_frags(sa, sb, 0, xa0, wb0);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa1, wb1, 1, 2);
      if constexpr (ISS) dma_piece(Ab + a_off[0] + kb, dst);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa1, wb1, 2, 5);
      if constexpr (ISS) dma_piece(Ab + a_off[1] + kb, dst + NW * 1024);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa1, wb1, 5, 8);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa0, wb0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(sa, sb, 1, xa1, wb1);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa0, wb0, 1, 2);
      if constexpr (ISS) dma_piece(Wb + b_off[0] + kb, dst + A_BYTES);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa0, wb0, 2, 5);
      if constexpr (ISS) dma_piece(Wb + b_off[1] + kb, dst + A_BYTES + NW * 1024);
      __builtin_amdgcn_sched_barrier(0);
      mma_range(xa0, wb0, 5, 8);
      return;
    }
    // program order: DMA pieces BEFORE the fragment reads (an LDS-DMA issued behind pending ds_reads makes hipcc
    // drain lgkmcnt first), then everything is re-interleaved behind the previous stage's last MFMA group
    if constexpr (decltype(issue_next)::value) issue_stage(st + 3);
    load_frags(sa, sb, 0, xa0, wb0);
    mma(xa1, wb1);
    load_frags(sa, sb, 1, xa1, wb1);
    mma(xa0, wb0);
    if constexpr (decltype(issue_next)::value) {
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM - 3, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
  };

  const int NS = p.K >> 5;  // stages (>= 2 since K % 64 == 0)
#pragma unroll 1
  for (int st = 0; st < 3 && st < NS; ++st) issue_stage(st);
  // steady state: two younger stages (8 pieces of this wave) stay in flight across the barrier
  int st = 0;
#pragma unroll 1
  for (; st + 3 < NS; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0078);  // vmcnt(8) lgkmcnt(0) (gfx9 encoding: vm[3:0] | exp<<4 | lgkm<<8 | vm[5:4]<<14)
    __builtin_amdgcn_s_barrier();
    stage_body(st, std::true_type{});
  }
  // drain: the last three stages, nothing left to issue
#pragma unroll 1
  for (; st < NS; ++st) {
    const int younger = NS - 1 - st;
    if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0078);       // vmcnt(8) lgkmcnt(0)
    else if (younger == 1) __builtin_amdgcn_s_waitcnt(0x0074);  // vmcnt(4) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    stage_body(st, std::false_type{});
  }
  mma(xa1, wb1);
  gemm_epilogue_lds<MI, NI, TM, TN, OUT_F32, ACT>(acc, p, m0, n0, wm, wn, lane, wave, smem);
}


// ---------------------------------------------------------------------------------------------------------
// Staggered kernel ("S"): the Q kernel's geometry (256 x 256 tile, BK = 32, 4-stage 128 KiB ring, 8 waves = 2 per
// SIMD) with the two wave halves running HALF A PERIOD APART.  Ablations of the lock-step kernels showed the MFMA
// loop alone sustains 77 % of peak but drops to 40 % as soon as fragment reads + DMA are added: both waves of a SIMD
// hit their LDS waits at the same time (matrix pipe idle) and then fight for the pipe at the same time.  Here every
// stage is split into a LOAD section (4 DMA pieces of stage s+3, the 12 fragment reads of stage s, counted vmcnt +
// lgkmcnt drain) and a MATRIX section (16 MFMAs under s_setprio 1), separated by raw s_barriers; waves 4-7 execute
// one extra barrier up front, so on each SIMD one wave is always in its matrix section while its partner loads:
//        interval:   0      1      2      3      4
//        waves 0-3:  L(0)   M(0)   L(1)   M(1)   L(2) ...
//        waves 4-7:  -      L(0)   M(0)   L(1)   M(1) ...
// Hazards (s = stage, buffer = s & 3, all barriers are whole-workgroup):
//   RAW  stage x is first read in L(x); its pieces were issued in L(x-3) and are waited for (vmcnt(8) = the two
//        younger stages may stay in flight) in L(x-1), which ends with a barrier BEFORE any L(x) starts.
//   WAR  the DMA for stage x (issued in L(x-3)) overwrites stage x-4, whose last reads (other half's L(x-4)) were
//        drained (lgkmcnt(0)) before the barrier that ends that interval, i.e. before L(x-3) of either half starts.
// TRACE: waves 0 and 4 of the first 64 blocks stamp s_memtime at every section boundary into `trace`
// ([block][2 waves][256] u64) — diagnostic variant 14, see tools/gemm_trace.py
template <bool OUT_F32, int ACT, int GM, bool TRACE = false, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_nt_kernel_s(const GemmArgs p, const int tiles_m, unsigned long long* trace = nullptr) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tm, tn;
  {
    const int per_group = GM * p.tiles_n;
    const int grp = bid / per_group, within = bid - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  const int slot = (lane & 15) ^ (lane >> 4);
  const int row16 = 4 * (lane >> 4) + (slot >> 2);
  const int chunk = slot & 3;
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int r = m0 + 16 * (wave + NW * j) + row16;
    r = r < p.M ? r : p.M - 1;
    a_off[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
    int rn = n0 + 16 * (wave + NW * j) + row16;
    rn = rn < p.N ? rn : p.N - 1;
    b_off[j] = ((uint32_t)rn * (uint32_t)p.ldw + chunk * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  auto issue_stage = [&](int st) __attribute__((always_inline)) {
    char* sbase = smem + (st & 3) * STAGE;
    const uint32_t kbytes = (uint32_t)st * 64u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((glb_u32p)(Ab + a_off[j] + kbytes), (lds_u32p)(sbase + (wave + NW * j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((glb_u32p)(Wb + b_off[j] + kbytes),
                                       (lds_u32p)(sbase + A_BYTES + (wave + NW * j) * 1024), 16, 0, 0);
  };

  const int l31 = lane & 31, half = lane >> 5;
  int roff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    roff[t] = (l31 >> 2) * 256 + (((((l31 & 3) << 2) | (2 * t + half)) ^ ((l31 >> 2) & 3)) << 4);

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  bf16x8 xa[2][MI], wb[2][NI];  // both k16-steps of one stage
  int tix = 0;
  unsigned long long* tr = nullptr;
  if constexpr (TRACE) {
    if (trace != nullptr && blockIdx.x < 64 && (wave & 3) == 0) tr = trace + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 256;
  }
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (TRACE) {
      if (tr != nullptr && tix < 255) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) tr[1 + tix] = t;
        ++tix;
      }
    }
  };

  // LOAD section of stage st: DMA of stage st+3 first (an LDS-DMA behind pending ds_reads would make hipcc drain them)
  auto load_section = [&](int st, auto issue_next, auto waitcode) __attribute__((always_inline)) {
    if constexpr (decltype(issue_next)::value && (ABL & 1) == 0) issue_stage(st + 3);
    if constexpr (TRACE) { __builtin_amdgcn_sched_barrier(0); stamp(); }
    const char* sa = smem + (st & 3) * STAGE + (wm * TM) * 64;
    const char* sb = smem + (st & 3) * STAGE + A_BYTES + (wn * TN) * 64;
    if constexpr ((ABL & 8) == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wb[t][ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 32 * 64 + roff[t]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xa[t][mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 32 * 64 + roff[t]);
      }
    }
    if constexpr (TRACE) {
      __builtin_amdgcn_sched_barrier(0); stamp();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
      __builtin_amdgcn_sched_barrier(0); stamp();
    }
    __builtin_amdgcn_s_waitcnt(decltype(waitcode)::value);  // fragments in registers; next stage's own pieces landed
    __builtin_amdgcn_sched_barrier(0);
  };
  auto matrix_section = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[t][ni], xa[t][mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using W8 = std::integral_constant<int, 0x0078>;  // vmcnt(8) lgkmcnt(0)
  using W4 = std::integral_constant<int, 0x0074>;  // vmcnt(4) lgkmcnt(0)
  using W0 = std::integral_constant<int, 0x0070>;  // vmcnt(0) lgkmcnt(0)

  stamp();
  const int NS = p.K >> 5;
#pragma unroll 1
  for (int st = 0; st < 3 && st < NS; ++st) issue_stage(st);
  // stage 0 visible to everyone before the first load section
  if (NS >= 3) __builtin_amdgcn_s_waitcnt(0x0078); else if (NS == 2) __builtin_amdgcn_s_waitcnt(0x0074); else __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the stagger: waves 4-7 run one interval behind
  stamp();

  int st = 0;
#pragma unroll 1
  for (; st + 3 < NS; ++st) {
    load_section(st, std::true_type{}, W8{});
    stamp();
    __builtin_amdgcn_s_barrier();
    stamp();
    matrix_section();
    stamp();
    __builtin_amdgcn_s_barrier();
    stamp();
  }
#pragma unroll 1
  for (; st < NS; ++st) {
    const int younger = NS - 2 - st;  // issued stages younger than st+1
    if (younger >= 2) load_section(st, std::false_type{}, W8{});
    else if (younger == 1) load_section(st, std::false_type{}, W4{});
    else load_section(st, std::false_type{}, W0{});
    __builtin_amdgcn_s_barrier();
    matrix_section();
    __builtin_amdgcn_s_barrier();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the stagger barrier
  stamp();
  gemm_epilogue_lds<MI, NI, TM, TN, OUT_F32, ACT>(acc, p, m0, n0, wm, wn, lane, wave, smem);
  stamp();
  if constexpr (TRACE) {
    if (tr != nullptr && lane == 0) tr[0] = (unsigned long long)tix;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Persistent pipelined kernel ("PP", the production kernel): the P schedule, but ONE workgroup per CU walks the tiles
// vb = blockIdx.x, + gridDim.x, ...  The r01 section trace of P showed 6 % of every tile in the exposed first-stage
// load and 22-55 % in the epilogue; here the first K-tile of the NEXT output tile is DMA'd (into ring buffer 0, free
// during the last K-tile) behind the last K-tile's MFMAs, so it lands while the epilogue runs out of ring buffer 1.
//   * epilogue strips live in ring buffer 1 only (32 rows x 128 B per wave and pass, fp32 tiles in two column halves)
//   * the fp32 residual of pass i+1 is requested before pass i is consumed
//   * tile order: same XCD-contiguous, GM-grouped order as P, applied to the virtual block id (gridDim.x % 8 == 0)
// WM x WN waves: 2 x 4 = eight 128x64 wave tiles (two waves per SIMD), or 2 x 2 = four 128x128 wave tiles (ONE wave per
// SIMD, 256 accumulator registers in the unified VGPR/AGPR file): 8 instead of 12 fragment reads per 16 MFMAs.
template <bool OUT_F32, int ACT, int GM, int WM = 2, int WN = 4, int STP = 0, int RDP = 0, int FOLD = 0, int RES_DEPTH = 1, bool BLDS = false, int A_MODE = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel_pp(const GemmArgs p, const int tiles_m, const int ntiles) {
  constexpr int BM = 256, BN = 256, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW, NDMA = A_INSTR + B_INSTR, NF = NI + MI, NM = NI * MI;
  static_assert(NM >= NDMA && NM >= NF && NW % 4 == 0, "interleave needs one MFMA per DMA piece / fragment read");
  constexpr int CH = TN / 64;  // 128-byte column chunks of a wave tile row in bf16
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, half = lane >> 5;

  auto tile_of = [&](int vb, int& tm, int& tn) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7, loc = vb >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per_group = GM * p.tiles_n;
    const int grp = id / per_group, within = id - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  };

  // DMA source offsets of this lane (see kernel above for the swizzle): depend on the tile, not on K
  const int sw = (4 * (wave & 3) + (lane >> 4)) & 15;
  const int slot = (lane & 15) ^ sw;
  const int row8 = 2 * (lane >> 4) + (slot >> 3);
  const int chunk = slot & 7;
  auto tile_offsets = [&](int tm, int tn, uint32_t (&ao)[A_INSTR], uint32_t (&bo)[B_INSTR]) {
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      int r = tm * BM + 8 * (wave + NW * j) + row8;
      r = r < p.M ? r : p.M - 1;
      if constexpr (A_MODE == 1) {  // patch row r = (b, gy, gx); this lane's 16-byte chunk = 8 pixels of patch row py_l of the K-tile
        const int b = r / p.i2c_g2, t = r - b * p.i2c_g2;
        const int gy = t / p.i2c_g, gx = t - gy * p.i2c_g;
        const int py_l = chunk >> p.i2c_lcr, px0 = (chunk & ((1 << p.i2c_lcr) - 1)) * 8;
        ao[j] = ((uint32_t)((b * 3 * p.i2c_hw + gy * p.i2c_p + py_l) * p.i2c_hw) + (uint32_t)(gx * p.i2c_p + px0)) * 2u;
      } else {
        ao[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
      }
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      int r = tn * BN + 8 * (wave + NW * j) + row8;
      r = r < p.N ? r : p.N - 1;
      bo[j] = ((uint32_t)r * (uint32_t)p.ldw + chunk * 8) * 2u;
    }
  };
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  uint32_t a_off[A_INSTR], b_off[B_INSTR], a_nxt[A_INSTR], b_nxt[B_INSTR];
  auto issue_piece = [&](int buf, int kt, int i) __attribute__((always_inline)) {
    const uint32_t dst = lds0 + buf * STAGE + (i < A_INSTR ? (wave + NW * i) * 1024 : A_BYTES + (wave + NW * (i - A_INSTR)) * 1024);
    if (i < A_INSTR) {
      if constexpr (A_MODE == 1) {  // K-tile kt = channel kt >> ltpc, image rows (kt & mask) * rpk .. + rpk - 1 of every patch
        const size_t koff = ((size_t)(kt >> p.i2c_ltpc) * p.i2c_hw * p.i2c_hw + (size_t)(kt & ((1 << p.i2c_ltpc) - 1)) * p.i2c_rpk * p.i2c_hw) * 2u;
        dma_piece_s(Ab + koff, a_off[i], dst);
      } else {
        dma_piece_s(Ab + (size_t)kt * 128, a_off[i], dst);
      }
    } else {
      dma_piece_s(Wb + (size_t)kt * 128, b_off[i - A_INSTR], dst);
    }
  };
  // output / residual row of GEMM row m (A_MODE = 1: the token row behind its image's CLS row; positional-embedding row of the patch)
  auto c_row = [&](int m) -> size_t { if constexpr (A_MODE == 1) return (size_t)m + (size_t)(m / p.i2c_g2) + 1; else return (size_t)m; };
  auto r_row = [&](int m) -> size_t { if constexpr (A_MODE == 1) return (size_t)(m % p.i2c_g2) + 1; else return (size_t)m; };

  const int hsw = l31 >> 1;
  uint32_t ra[2][4], rb[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t ro = hsw * 256 + (((((l31 & 1) << 3) | (2 * t + half)) ^ hsw) << 4);
      ra[bf][t] = lds0 + bf * STAGE + (wm * TM) * 128 + ro;
      rb[bf][t] = lds0 + bf * STAGE + A_BYTES + (wn * TN) * 128 + ro;
    }
  typedef __attribute__((address_space(3))) const bf16x8* lds_frag_p;

  f32x16 acc[NI][MI];
  bf16x8 xa0[MI], wb0[NI], xa1[MI], wb1[NI];
  auto load_frags = [&](auto bufc, int t, bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
    constexpr int BF = decltype(bufc)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wb[ni] = *reinterpret_cast<lds_frag_p>((uintptr_t)(rb[BF][t] + ni * 32 * 128));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<lds_frag_p>((uintptr_t)(ra[BF][t] + mi * 32 * 128));
  };
  auto mma = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
  };
  auto mma_one = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI], int i) {
    acc[i / MI][i % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i / MI], xa[i % MI], acc[i / MI][i % MI], 0, 0, 0);
  };
  // one K-tile out of ring buffer BF; behind the first MFMA group one DMA piece each of (ktsrc -> buffer BF^1)
  auto tile_body = [&](auto bufc, int ktsrc, const bool issue) __attribute__((always_inline)) {
    constexpr int BF = decltype(bufc)::value;
    load_frags(bufc, 0, xa0, wb0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      mma_one(xa1, wb1, i);
      if (issue && i < NDMA) issue_piece(BF ^ 1, ktsrc, i);  // wave-uniform scalar branch
      __builtin_amdgcn_sched_barrier(0);
    }
    load_frags(bufc, 1, xa1, wb1);
    mma(xa0, wb0);
    load_frags(bufc, 2, xa0, wb0);
    mma(xa1, wb1);
    load_frags(bufc, 3, xa1, wb1);
    mma(xa0, wb0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if constexpr (RDP == 0) {         // one read of the next k-step behind each of the first NF MFMAs
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
      } else if constexpr (RDP == 1) {  // all NF reads in one burst behind the first MFMA (maximum cover)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 0);
      } else {                          // two reads behind each of the first NF/2 MFMAs
#pragma unroll
        for (int i = 0; i < NF / 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NF / 2, 0);
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  auto sync_tile = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int KT = p.K >> 6;  // even (launcher)
  int vb = blockIdx.x;
  int tm, tn;
  tile_of(vb, tm, tn);
  tile_offsets(tm, tn, a_off, b_off);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) issue_piece(0, 0, i);
  if (p.stagger > 0) {
    const int heavy = ntiles % (int)gridDim.x;  // workgroups 0 .. heavy-1 walk one tile more: they start at once
    if (heavy > 0 && (int)blockIdx.x >= heavy) {
      const long long delay = (long long)p.stagger * ((int)blockIdx.x - heavy + 1) / ((int)gridDim.x - heavy);
      const long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(32);
    }
  } else if (p.stagger < 0) {
    // experiment (mmamd_debug_set_gemm_stagger(1000 + percent)): EVERY workgroup is delayed, the 32 of an XCD spread evenly over
    // [0, |stagger|) -- de-synchronises the chip-wide store bursts at the price of up to |stagger| of makespan
    const long long delay = (long long)(-p.stagger) * ((int)blockIdx.x >> 3) / (((int)gridDim.x + 7) >> 3);
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(32);
  }

  while (true) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int nvb = vb + gridDim.x;
    const bool more = nvb < ntiles;
    int ntm = 0, ntn = 0;
    if (more) {
      tile_of(nvb, ntm, ntn);
      tile_offsets(ntm, ntn, a_nxt, b_nxt);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 8; ++j) xa1[mi][j] = (bf16)0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 8; ++j) wb1[ni][j] = (bf16)0.f;

#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
      sync_tile();
      if constexpr (FOLD == 1) {
        // LN fold (consumer): the block statistics of this tile's 256 rows ([256][nslot][2] floats, one contiguous 256 * nslot * 8 B
        // block of p.st_in) go to LDS behind the ring by LDS-DMA, issued after the first barrier of the tile (every wave has left the
        // previous tile's epilogue, which read the previous statistics) and waited for by the next sync_tile: the epilogue finds
        // them in LDS instead of paying nslot/2 dependent L2 round trips per row
        if (kt == 0) {
          const uint32_t total = (uint32_t)p.nslot_in * (BM * 8u), limit = (uint32_t)(p.M - m0) * (uint32_t)p.nslot_in * 8u - 16u;
          const char* sbase = reinterpret_cast<const char*>(p.st_in) + (size_t)m0 * (size_t)p.nslot_in * 8u;
          for (uint32_t pc = wave; pc * 1024u < total; pc += NW) {
            uint32_t off = pc * 1024u + lane * 16u;
            off = off < limit ? off : limit;  // rows past M (last tile) read the last valid 16 bytes: never stored anyway
            dma_piece_s(sbase, off, lds0 + 2 * STAGE + pc * 1024u);
          }
        }
      }
      if constexpr (BLDS && FOLD != 1) {
        // the tile's 256 bias values (1 KiB = one DMA piece, issued by wave 0) land in LDS behind the ring while the K loop runs;
        // same hazards as the statistics above: issued after the tile's first barrier, waited for by the next sync_tile
        if (kt == 0 && wave == 0 && p.bias != nullptr) {
          const uint32_t limit = (uint32_t)(p.N - n0) * 4u - 16u;
          uint32_t off = lane * 16u;
          off = off < limit ? off : limit;  // columns past N (never stored) re-read the last valid 16 bytes
          dma_piece_s(reinterpret_cast<const char*>(p.bias) + (size_t)n0 * 4u, off, lds0 + 2 * STAGE);
        }
      }
      tile_body(B0{}, kt + 1, true);
      const bool last = kt + 2 >= KT;
      if (last && more) {  // this tile's loads are all issued: switch the DMA source to the next tile's first K-tile
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) a_off[j] = a_nxt[j];
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) b_off[j] = b_nxt[j];
      }
      sync_tile();
      tile_body(B1{}, last ? 0 : kt + 2, !last || more);
    }
    mma(xa1, wb1);  // flush the rotated last k-step

    // ---------------- epilogue (LDS strips in ring buffer 1; buffer 0 is receiving the next tile) ----------------
    // fp32 residual: a ring of RD passes of loads in flight, the first RD issued here (before the bias / barrier / first transpose).
    // Measured (tools/gemm_variant_bench.py --variants 60,61,62,63, r02): depth 1 = 2 (out-proj 100.7 / 100.5 us), depth 3 and 4
    // LOSE (115 / 133 us: 21 spilled registers and more loads queued per CU) -- the residual's latency is not what the fp32
    // epilogue waits for; the default stays 1.
    constexpr int RD = OUT_F32 ? RES_DEPTH : 1;
    const int nw0 = n0 + wn * TN;
    const int rrow = lane >> 3, rc = (lane & 7) * 4;  // fp32 read-back: 8 rows x 128 B per wave-instruction
    f32x4 rq[RD][4];
    auto res_load = [&](int pass, f32x4 (&dst)[4]) __attribute__((always_inline)) {  // pass = mi * NI + ni
      const int mi = pass / NI, ni = pass - mi * NI;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = m0 + wm * TM + mi * 32 + it * 8 + rrow, n = nw0 + ni * 32 + rc;
        dst[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M && n + 3 < p.N) dst[it] = load4(reinterpret_cast<const float*>(p.R) + r_row(m) * p.ldr + n);
      }
    };
    if constexpr (OUT_F32) {
      if (p.R != nullptr) {
#pragma unroll
        for (int d = 0; d < RD; ++d) res_load(d, rq[d]);
      }
    }
    bias_or_lnfold<MI, NI, TM, TN, FOLD, ACT, FOLD == 1>(acc, p, m0, n0, wm, wn, lane, smem + 2 * STAGE,
                                                         (BLDS && FOLD != 1) ? smem + 2 * STAGE : nullptr);
    if constexpr (ACT == MMAMD_ACT_QUICKGELU && FOLD != 1) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][mi][r] = quick_gelu(acc[ni][mi][r]);
    }
    constexpr int ROWB = 144;  // 128-byte strip rows + 16 B pad (conflict-free b128 both ways)
    char* strip = smem + STAGE + wave * (32 * ROWB);
    __syncthreads();  // every wave has finished reading the last K-tile out of ring buffer 1
    if constexpr (OUT_F32) {
      const bool has_res = p.R != nullptr;
      float fs1[4], fs2[4];  // LN fold (producer): row partials of the 4 rows this lane touches per pass, over the wave's TN columns
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int pass = mi * NI + ni;
          if constexpr (FOLD == 2) {
            if (ni == 0) {
#pragma unroll
              for (int it = 0; it < 4; ++it) fs1[it] = fs2[it] = 0.f;
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = ACT == MMAMD_ACT_GELU_ERF ? gelu_erf(acc[ni][mi][4 * g + j]) : acc[ni][mi][4 * g + j];
            *reinterpret_cast<f32x4*>(strip + l31 * ROWB + (8 * g + 4 * half) * 4) = t;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          f32x4 vv[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) vv[it] = *reinterpret_cast<const f32x4*>(strip + (it * 8 + rrow) * ROWB + rc * 4);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int m = m0 + wm * TM + mi * 32 + it * 8 + rrow, n = nw0 + ni * 32 + rc;
            const bool ok = m < p.M && n + 3 < p.N;
            f32x4 v = vv[it];
            if (ok) {
              if (has_res) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rq[pass % RD][it][j];
              }
              store16<STP>(reinterpret_cast<float*>(p.C) + c_row(m) * p.ldc + n, __builtin_bit_cast(uint4, v));
            }
            if constexpr (FOLD == 2) {  // 8 lanes hold 32 columns of row m in this pass
              lnfold_store_acc(p, v, m, n, ok, fs1[it], fs2[it]);  // lane c = lane & 7: quad c of pass 0, quad c + 8 of pass 1
              if (ni == NI - 1) {
                lnfold_butterfly8(fs1[it], fs2[it]);
                if (m < p.M && (lane & 7) == 0)
                  *reinterpret_cast<f32x2*>(p.st_out + ((size_t)m * p.nslot_out + (nw0 >> 6)) * 2) = f32x2{fs1[it], fs2[it]};
              }
            }
          }
          if (has_res && pass + RD < MI * NI) res_load(pass + RD, rq[pass % RD]);  // refill the slot this pass just consumed
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {  // 64 columns (128 B of bf16) per pass
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
              const int ni = 2 * ch + nn;
              bf16x4 pa, pb;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float va = acc[ni][mi][4 * g + j], vb = acc[ni][mi][4 * (g + 1) + j];
                if constexpr (ACT == MMAMD_ACT_GELU_ERF) { va = gelu_erf(va); vb = gelu_erf(vb); }
                pa[j] = (bf16)va; pb[j] = (bf16)vb;
              }
              uint2 ua = __builtin_bit_cast(uint2, pa), ub = __builtin_bit_cast(uint2, pb);
              auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
              auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
              *reinterpret_cast<uint4*>(strip + l31 * ROWB + (nn * 32 + 8 * (g + half)) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int it = 0; it < 4; ++it) {  // 8 rows x 128 B per wave-instruction
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            uint4 v = *reinterpret_cast<const uint4*>(strip + row * ROWB + c * 16);
            const int m = m0 + wm * TM + mi * 32 + row, n = nw0 + ch * 64 + c * 8;
            if (m < p.M && n + 7 < p.N) {
              if (p.R != nullptr) {
                const uint4 rr = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.R) + (size_t)m * p.ldr + n);
                bf16x8 a8 = __builtin_bit_cast(bf16x8, v), r8 = __builtin_bit_cast(bf16x8, rr);
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = (bf16)combine_res((float)a8[j], (float)r8[j], p.res_mode);
                v = __builtin_bit_cast(uint4, a8);
              }
              store16<STP>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n, v);
              store_act_copy(p, v, m, n);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (!more) break;
    vb = nvb;
    tm = ntm;
    tn = ntn;
  }
}

// ---------------------------------------------------------------------------------------------------------
// GROUPED persistent kernel ("PPG"): the persistent kernel above walking the CONCATENATED tile lists of up to two GEMM problems that
// share the epilogue kind (activation, output type) but not the shapes -- the same projection of the two towers of the dual encoder
// (ViT [50432 x N x 768] and text [19712 x N' x 512] at cfg 2).  One launch instead of two on two streams: the second problem's tiles
// fill the partial last round of the first (a persistent workgroup owns its CU for the whole launch, so a second stream's kernels could
// only run in whatever the first one's tail left over -- in-bench every ViT kernel ran 17-60 % slower than alone and the text tower's
// 24-50 us GEMMs took 150 us on average).  Per tile the problem's fields are (re)loaded from the kernel argument segment (scalar loads);
// K, the leading dimensions, the operand bases and the epilogue pointers all change at a problem boundary, and the first K-tile of the NEXT
// tile -- possibly of the other problem -- is DMA'd behind the last K-tile of the current one exactly as in the single-problem kernel.
struct GemmProblem {
  const bf16* A;
  const bf16* W;
  const float* bias;
  const void* R;
  void* C;
  int M, N, K;
  int lda, ldw, ldr, ldc;
  int tiles_m, tiles_n;
};
struct GemmGroupArgs {
  GemmProblem prob[2];
  int nprob;
  int tile_start[3];  // tile_start[i] = first tile id of problem i; tile_start[nprob] = total
  int stagger;
};

template <bool OUT_F32, int ACT, int GM>
__global__ __launch_bounds__(512) void gemm_bf16_nt_kernel_ppg(const GemmGroupArgs g) {
  constexpr int WM = 2, WN = 4, STP = OUT_F32 ? 0 : 2, RDP = 0;
  constexpr int BM = 256, BN = 256, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW, NDMA = A_INSTR + B_INSTR, NF = NI + MI, NM = NI * MI;
  static_assert(NM >= NDMA && NM >= NF && NW % 4 == 0, "interleave needs one MFMA per DMA piece / fragment read");
  constexpr int CH = TN / 64;  // 128-byte column chunks of a wave tile row in bf16
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, half = lane >> 5;

  // Tile walk.  Workgroup b runs on XCD b & 7 and is the (b >> 3)-th of that XCD's `nwx` workgroups.  Every XCD owns a contiguous slice of
  // EACH problem's tile list (an eighth of it: operand panels stay in that XCD's L2) and walks its slice of problem 0 first, then its
  // slice of problem 1 -- so the eight XCDs carry equal WORK, not just equal tile counts (a split of the concatenated list gave the
  // last XCDs only the second problem's shorter tiles: K = 512 against 768, they idled a quarter of the launch).
  const int xcd = (int)blockIdx.x & 7, wl = (int)blockIdx.x >> 3;
  const int nwx = ((int)gridDim.x - xcd + 7) >> 3;
  int xb0, xc0, xb1 = 0, xc1 = 0;  // this XCD's slice [base, base + count) of problem 0 / 1
  {
    const int n0 = g.tile_start[1] - g.tile_start[0], q0 = n0 >> 3, r0 = n0 & 7;
    xc0 = q0 + (xcd < r0 ? 1 : 0);
    xb0 = xcd < r0 ? xcd * (q0 + 1) : r0 * (q0 + 1) + (xcd - r0) * q0;
    if (g.nprob > 1) {
      const int n1 = g.tile_start[2] - g.tile_start[1], q1 = n1 >> 3, r1 = n1 & 7;
      xc1 = q1 + (xcd < r1 ? 1 : 0);
      xb1 = xcd < r1 ? xcd * (q1 + 1) : r1 * (q1 + 1) + (xcd - r1) * q1;
    }
  }
  const int nx = xc0 + xc1;  // tiles of this XCD
  if (wl >= nx) return;      // (whole workgroup: before any barrier)
  // XCD-local tile index -> (problem, tile): the GM-grouped order inside each problem's own tile grid
  auto tile_of = [&](int l, int& sel, int& tm, int& tn) __attribute__((always_inline)) {
    sel = l >= xc0 ? 1 : 0;
    const int id = sel ? xb1 + (l - xc0) : xb0 + l;
    const int tiles_m = g.prob[sel].tiles_m;
    const int per_group = GM * g.prob[sel].tiles_n;
    const int grp = id / per_group, within = id - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  };

  // DMA source offsets of this lane (see kernel above for the swizzle): depend on the tile, not on K
  const int sw = (4 * (wave & 3) + (lane >> 4)) & 15;
  const int slot = (lane & 15) ^ sw;
  const int row8 = 2 * (lane >> 4) + (slot >> 3);
  const int chunk = slot & 7;
  auto tile_offsets = [&](const GemmProblem& p, int tm, int tn, uint32_t (&ao)[A_INSTR], uint32_t (&bo)[B_INSTR]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      int r = tm * BM + 8 * (wave + NW * j) + row8;
      r = r < p.M ? r : p.M - 1;
      ao[j] = ((uint32_t)r * (uint32_t)p.lda + chunk * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      int r = tn * BN + 8 * (wave + NW * j) + row8;
      r = r < p.N ? r : p.N - 1;
      bo[j] = ((uint32_t)r * (uint32_t)p.ldw + chunk * 8) * 2u;
    }
  };
  int vb = wl, sel = 0;  // XCD-local tile index of the current tile
  int tm, tn;
  tile_of(vb, sel, tm, tn);
  GemmProblem p = g.prob[sel];  // the CURRENT tile's problem (scalar loads from the kernel argument segment, refreshed per tile)
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  uint32_t a_off[A_INSTR], b_off[B_INSTR], a_nxt[A_INSTR], b_nxt[B_INSTR];
  auto issue_piece = [&](int buf, int kt, int i) __attribute__((always_inline)) {
    const uint32_t dst = lds0 + buf * STAGE + (i < A_INSTR ? (wave + NW * i) * 1024 : A_BYTES + (wave + NW * (i - A_INSTR)) * 1024);
    if (i < A_INSTR) dma_piece_s(Ab + (size_t)kt * 128, a_off[i], dst);
    else dma_piece_s(Wb + (size_t)kt * 128, b_off[i - A_INSTR], dst);
  };

  const int hsw = l31 >> 1;
  uint32_t ra[2][4], rb[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t ro = hsw * 256 + (((((l31 & 1) << 3) | (2 * t + half)) ^ hsw) << 4);
      ra[bf][t] = lds0 + bf * STAGE + (wm * TM) * 128 + ro;
      rb[bf][t] = lds0 + bf * STAGE + A_BYTES + (wn * TN) * 128 + ro;
    }
  typedef __attribute__((address_space(3))) const bf16x8* lds_frag_p;

  f32x16 acc[NI][MI];
  bf16x8 xa0[MI], wb0[NI], xa1[MI], wb1[NI];
  auto load_frags = [&](auto bufc, int t, bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
    constexpr int BF = decltype(bufc)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wb[ni] = *reinterpret_cast<lds_frag_p>((uintptr_t)(rb[BF][t] + ni * 32 * 128));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<lds_frag_p>((uintptr_t)(ra[BF][t] + mi * 32 * 128));
  };
  auto mma = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
  };
  auto mma_one = [&](bf16x8 (&xa)[MI], bf16x8 (&wb)[NI], int i) {
    acc[i / MI][i % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i / MI], xa[i % MI], acc[i / MI][i % MI], 0, 0, 0);
  };
  // one K-tile out of ring buffer BF; behind the first MFMA group one DMA piece each of (ktsrc -> buffer BF^1)
  auto tile_body = [&](auto bufc, int ktsrc, const bool issue) __attribute__((always_inline)) {
    constexpr int BF = decltype(bufc)::value;
    load_frags(bufc, 0, xa0, wb0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      mma_one(xa1, wb1, i);
      if (issue && i < NDMA) issue_piece(BF ^ 1, ktsrc, i);  // wave-uniform scalar branch
      __builtin_amdgcn_sched_barrier(0);
    }
    load_frags(bufc, 1, xa1, wb1);
    mma(xa0, wb0);
    load_frags(bufc, 2, xa0, wb0);
    mma(xa1, wb1);
    load_frags(bufc, 3, xa1, wb1);
    mma(xa0, wb0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if constexpr (RDP == 0) {         // one read of the next k-step behind each of the first NF MFMAs
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
      } else if constexpr (RDP == 1) {  // all NF reads in one burst behind the first MFMA (maximum cover)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 0);
      } else {                          // two reads behind each of the first NF/2 MFMAs
#pragma unroll
        for (int i = 0; i < NF / 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NF / 2, 0);
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  auto sync_tile = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  int KT = p.K >> 6;  // even (launcher), per problem
  tile_offsets(p, tm, tn, a_off, b_off);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) issue_piece(0, 0, i);
  if (g.stagger > 0) {
    const int heavy = nx % nwx;  // this XCD's workgroups 0 .. heavy-1 walk one tile more: they start at once
    if (heavy > 0 && wl >= heavy) {
      const long long delay = (long long)g.stagger * (wl - heavy + 1) / (nwx - heavy);
      const long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(32);
    }
  }

  while (true) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int nvb = vb + nwx;
    const bool more = nvb < nx;
    int ntm = 0, ntn = 0, nsel = sel;
    const char *Abn = Ab, *Wbn = Wb;
    if (more) {
      tile_of(nvb, nsel, ntm, ntn);
      tile_offsets(g.prob[nsel], ntm, ntn, a_nxt, b_nxt);
      Abn = reinterpret_cast<const char*>(g.prob[nsel].A);
      Wbn = reinterpret_cast<const char*>(g.prob[nsel].W);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 8; ++j) xa1[mi][j] = (bf16)0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 8; ++j) wb1[ni][j] = (bf16)0.f;

#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
      sync_tile();
      tile_body(B0{}, kt + 1, true);
      const bool last = kt + 2 >= KT;
      if (last && more) {  // this tile's loads are all issued: switch the DMA source to the next tile's first K-tile
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) a_off[j] = a_nxt[j];
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) b_off[j] = b_nxt[j];
        Ab = Abn;  // ... which may belong to the other problem
        Wb = Wbn;
      }
      sync_tile();
      tile_body(B1{}, last ? 0 : kt + 2, !last || more);
    }
    mma(xa1, wb1);  // flush the rotated last k-step

    // ---------------- epilogue (LDS strips in ring buffer 1; buffer 0 is receiving the next tile) ----------------
    // fp32 residual: a ring of RD passes of loads in flight, the first RD issued here (before the bias / barrier / first transpose).
    // Measured (tools/gemm_variant_bench.py --variants 60,61,62,63, r02): depth 1 = 2 (out-proj 100.7 / 100.5 us), depth 3 and 4
    // LOSE (115 / 133 us: 21 spilled registers and more loads queued per CU) -- the residual's latency is not what the fp32
    // epilogue waits for; the default stays 1.
    constexpr int RD = 1;
    const int nw0 = n0 + wn * TN;
    const int rrow = lane >> 3, rc = (lane & 7) * 4;  // fp32 read-back: 8 rows x 128 B per wave-instruction
    f32x4 rq[RD][4];
    auto res_load = [&](int pass, f32x4 (&dst)[4]) __attribute__((always_inline)) {  // pass = mi * NI + ni
      const int mi = pass / NI, ni = pass - mi * NI;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = m0 + wm * TM + mi * 32 + it * 8 + rrow, n = nw0 + ni * 32 + rc;
        dst[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M && n + 3 < p.N) dst[it] = load4(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + n);
      }
    };
    if constexpr (OUT_F32) {
      if (p.R != nullptr) {
#pragma unroll
        for (int d = 0; d < RD; ++d) res_load(d, rq[d]);
      }
    }
    if (p.bias != nullptr) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n = n0 + wn * TN + ni * 32 + 4 * half + 8 * g4;
          f32x4 bv = {0.f, 0.f, 0.f, 0.f};
          if (n + 3 < p.N) bv = load4(p.bias + n);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ni][mi][4 * g4 + j] += bv[j];
        }
    }
    if constexpr (ACT == MMAMD_ACT_QUICKGELU) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][mi][r] = quick_gelu(acc[ni][mi][r]);
    }
    constexpr int ROWB = 144;  // 128-byte strip rows + 16 B pad (conflict-free b128 both ways)
    char* strip = smem + STAGE + wave * (32 * ROWB);
    __syncthreads();  // every wave has finished reading the last K-tile out of ring buffer 1
    if constexpr (OUT_F32) {
      const bool has_res = p.R != nullptr;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int pass = mi * NI + ni;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = ACT == MMAMD_ACT_GELU_ERF ? gelu_erf(acc[ni][mi][4 * g4 + j]) : acc[ni][mi][4 * g4 + j];
            *reinterpret_cast<f32x4*>(strip + l31 * ROWB + (8 * g4 + 4 * half) * 4) = t;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          f32x4 vv[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) vv[it] = *reinterpret_cast<const f32x4*>(strip + (it * 8 + rrow) * ROWB + rc * 4);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int m = m0 + wm * TM + mi * 32 + it * 8 + rrow, n = nw0 + ni * 32 + rc;
            const bool ok = m < p.M && n + 3 < p.N;
            f32x4 v = vv[it];
            if (ok) {
              if (has_res) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rq[pass % RD][it][j];
              }
              store16<STP>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, __builtin_bit_cast(uint4, v));
            }
          }
          if (has_res && pass + RD < MI * NI) res_load(pass + RD, rq[pass % RD]);  // refill the slot this pass just consumed
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {  // 64 columns (128 B of bf16) per pass
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4 += 2) {
              const int ni = 2 * ch + nn;
              bf16x4 pa, pb;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float va = acc[ni][mi][4 * g4 + j], vb2 = acc[ni][mi][4 * (g4 + 1) + j];
                if constexpr (ACT == MMAMD_ACT_GELU_ERF) { va = gelu_erf(va); vb2 = gelu_erf(vb2); }
                pa[j] = (bf16)va; pb[j] = (bf16)vb2;
              }
              uint2 ua = __builtin_bit_cast(uint2, pa), ub = __builtin_bit_cast(uint2, pb);
              auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
              auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
              *reinterpret_cast<uint4*>(strip + l31 * ROWB + (nn * 32 + 8 * (g4 + half)) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int it = 0; it < 4; ++it) {  // 8 rows x 128 B per wave-instruction
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            uint4 v = *reinterpret_cast<const uint4*>(strip + row * ROWB + c * 16);
            const int m = m0 + wm * TM + mi * 32 + row, n = nw0 + ch * 64 + c * 8;
            if (m < p.M && n + 7 < p.N) {
              if (p.R != nullptr) {
                const uint4 rr = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.R) + (size_t)m * p.ldr + n);
                bf16x8 a8 = __builtin_bit_cast(bf16x8, v), r8 = __builtin_bit_cast(bf16x8, rr);
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = (bf16)((float)a8[j] + (float)r8[j]);
                v = __builtin_bit_cast(uint4, a8);
              }
              store16<STP>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n, v);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (!more) break;
    vb = nvb;
    tm = ntm;
    tn = ntn;
    if (nsel != sel) {  // wave-uniform: the walk crossed into the next problem
      sel = nsel;
      p = g.prob[sel];
    }
    KT = p.K >> 6;
  }
}


#ifdef MMAMD_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------
// "W" kernel (experiment, r02: correct, bit-equal to the production kernels, and SLOWER -- qkv 200 vs 160 us, MLP-up 277 vs 252,
// MLP-down 316 vs 270; PMC profiles/r02_pmc_gemm_w.txt: MFMA pipe 41 % busy vs 53 %, 57 % of wave cycles in issue stalls;
// six DMA pieces per 16 MFMAs per wave against eight per 32 in the 8-wave kernels): TWO workgroups per CU.  The lock-step kernels above stop the matrix pipe for the whole per-tile epilogue (22-55 % of a
// tile's time at K = 768: bias / activation VALU, the LDS transpose, the stores) because all eight waves of the CU's one workgroup are
// in it together.  Here a workgroup is 4 waves (one per SIMD) on a 256 x 128 tile (2 x 2 waves of 128 x 64, the same wave tile and
// fragment traffic as above) with a 3-stage BK = 32 LDS-DMA ring (72 KiB), so two workgroups are resident per CU with independent
// barriers: they drift out of phase, and while one is in its prologue / epilogue the other one's K loop owns the matrix pipe.
//   LDS image of a K-tile: 64-byte rows (32 bf16), 4 rows per 256-byte bank row = 16 slots of 16 B; logical slot s = 4 (r & 3) + c
//   (c = 16-byte chunk) is stored at s ^ (R & 15), R = r >> 2 -- the permutation is applied to the DMA SOURCE address (the LDS
//   destination of a piece is lane-linear) and undone by the ds_read_b128 address: the 16 lanes of every read group hit 16 slots.
//   RAW: tile kt's pieces (issued two iterations earlier) are waited with vmcnt(6) (tile kt+1's six may stay in flight) + barrier.
//   WAR: tile kt+2 goes into the stage read during iteration kt-1; every wave has passed iteration kt's barrier by then.
// SCH 0: DMA burst behind the barrier, then reads, then MFMAs.  SCH 1: one DMA piece behind each of the first six MFMAs (the burst is
// ~6 x 60-100 cycles in front of a 512-cycle MFMA block).  SCH 2: SCH 1 + raised priority over the MFMA block.
template <bool OUT_F32, int ACT, int GM, int SCH = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel_w(const GemmArgs p, const int tiles_m) {
  constexpr int BM = 256, BN = 128, WN = 2, NW = 4, TM = 128, TN = 64, MI = 4, NI = 2;
  constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64, A_INSTR = 4, B_INSTR = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tm, tn;
  {
    const int per_group = GM * p.tiles_n;
    const int grp = bid / per_group, within = bid - grp * per_group;
    const int gm0 = grp * GM;
    const int rows = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tn = within / rows;
    tm = gm0 + (within - tn * rows);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, half = lane >> 5;

  // DMA source offsets: piece q = wave + 4 j covers image rows 16 q .. 16 q + 15; lane -> bank row R = 4 q + (lane >> 4), stored
  // slot lane & 15 holds logical slot s = (lane & 15) ^ (R & 15); (R & 15) = (4 wave + (lane >> 4)) & 15 for every j
  const int sxr = (4 * wave + (lane >> 4)) & 15;
  const int ls = (lane & 15) ^ sxr;
  const int prow = 4 * (lane >> 4) + (ls >> 2), pchunk = ls & 3;
  uint32_t a_off[A_INSTR], b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    int r = m0 + 16 * (wave + NW * j) + prow;
    r = r < p.M ? r : p.M - 1;
    a_off[j] = ((uint32_t)r * (uint32_t)p.lda + pchunk * 8) * 2u;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    int r = n0 + 16 * (wave + NW * j) + prow;
    r = r < p.N ? r : p.N - 1;
    b_off[j] = ((uint32_t)r * (uint32_t)p.ldw + pchunk * 8) * 2u;
  }
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)smem;
  auto issue_tile = [&](int stage, int kt) __attribute__((always_inline)) {
    const uint32_t dst = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) dma_piece_s(Ab + (size_t)kt * 64, a_off[j], dst + NW * j * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) dma_piece_s(Wb + (size_t)kt * 64, b_off[j], dst + A_BYTES + NW * j * 1024);
  };

  // fragment read offsets inside a stage: operand row r, k-step t -> chunk c = 2 t + half
  uint32_t ra[MI][2], rb[NI][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int r = wm * TM + mi * 32 + l31, R = r >> 2;
      ra[mi][t] = R * 256 + (((((r & 3) << 2) | (2 * t + half)) ^ (R & 15)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int r = wn * TN + ni * 32 + l31, R = r >> 2;
      rb[ni][t] = A_BYTES + R * 256 + (((((r & 3) << 2) | (2 * t + half)) ^ (R & 15)) << 4);
    }
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int KT = p.K >> 5;  // >= 2 (K % 64 == 0)
  issue_tile(0, 0);
  issue_tile(1, 1);
  auto body = [&](auto stagec, int kt) __attribute__((always_inline)) {
    constexpr int ST = decltype(stagec)::value;
    if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile kt landed; tile kt+1's six pieces may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool issue = kt + 2 < KT;
    if constexpr (SCH == 0) {
      if (issue) issue_tile((ST + 2) % 3, kt + 2);
    }
    const char* sb = smem + ST * STAGE;
    bf16x8 xa[2][MI], wb[2][NI];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wb[t][ni] = *reinterpret_cast<const bf16x8*>(sb + rb[ni][t]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xa[t][mi] = *reinterpret_cast<const bf16x8*>(sb + ra[mi][t]);
    }
    if constexpr (SCH == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[t][ni], xa[t][mi], acc[ni][mi], 0, 0, 0);
    } else {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SCH == 2) __builtin_amdgcn_s_setprio(1);
      const uint32_t dst = lds0 + ((ST + 2) % 3) * STAGE + wave * 1024;
#pragma unroll
      for (int i = 0; i < NI * MI; ++i) {
        acc[i / MI][i % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[0][i / MI], xa[0][i % MI], acc[i / MI][i % MI], 0, 0, 0);
        if (issue && i < A_INSTR + B_INSTR) {  // wave-uniform scalar branch
          if (i < A_INSTR) dma_piece_s(Ab + (size_t)(kt + 2) * 64, a_off[i], dst + NW * i * 1024);
          else dma_piece_s(Wb + (size_t)(kt + 2) * 64, b_off[i - A_INSTR], dst + A_BYTES + NW * (i - A_INSTR) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[1][ni], xa[1][mi], acc[ni][mi], 0, 0, 0);
      if constexpr (SCH == 2) __builtin_amdgcn_s_setprio(0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int kt = 0; kt < KT; kt += 3) {
    body(S0{}, kt);
    if (kt + 1 < KT) body(S1{}, kt + 1);
    if (kt + 2 < KT) body(S2{}, kt + 2);
  }
  gemm_epilogue_lds<MI, NI, TM, TN, OUT_F32, ACT>(acc, p, m0, n0, wm, wn, lane, wave, smem);
}

#endif  // MMAMD_EXPERIMENTS

// plain one-thread-per-output kernel: on-device cross-check for the MFMA kernels (tests / debugging)
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_naive_kernel(const GemmArgs p) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= p.M || n >= p.N) return;
  float acc = 0.f;
  for (int k = 0; k < p.K; ++k) acc = fmaf((float)p.A[(size_t)m * p.lda + k], (float)p.W[(size_t)n * p.ldw + k], acc);
  if (p.bias) acc += p.bias[n];
  acc = apply_act(acc, p.act);
  if constexpr (OUT_F32) {
    if (p.R) acc += reinterpret_cast<const float*>(p.R)[(size_t)m * p.ldr + n];
    reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = acc;
  } else {
    if (p.R) acc += (float)reinterpret_cast<const bf16*>(p.R)[(size_t)m * p.ldr + n];
    reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + n] = (bf16)acc;
    if (p.C2) reinterpret_cast<bf16*>(p.C2)[(size_t)m * p.ldc2 + n] = (bf16)apply_act((float)(bf16)acc, p.act2);
  }
}

template <int BM, int BN, int WM, int WN, bool OUT_F32, int ACT, bool SGB, int FOLD = 0>
static int launch_tiled(GemmArgs& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, OUT_F32, ACT, SGB, FOLD>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, st, p);
  return launch_status("gemm_bf16");
}

template <int BM, int BN, int WM, int WN, bool OUT_F32, int ACT, int GM, int ABL = 0, bool LDSEPI = true, int FOLD = 0>
static int launch_tiled_p(GemmArgs& p, hipStream_t st) {
  // the pipelined kernel walks the K-tiles in pairs (compile-time buffer index): odd tile counts take the plain kernel
  if (((p.K >> 6) & 1) != 0) return launch_tiled<BM, BN, WM, WN, OUT_F32, ACT, true, FOLD>(p, st);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = gemm_bf16_nt_kernel_p<BM, BN, WM, WN, OUT_F32, ACT, GM, ABL, LDSEPI, false, 0, FOLD>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, st, p, tiles_m,
                     (ABL & 64) != 0 ? g_gemm_trace : nullptr);
  return launch_status("gemm_bf16_p");
}

template <bool OUT_F32, int ACT, int GM, int SCHED = 0>
static int launch_tiled_q(GemmArgs& p, hipStream_t st) {
  constexpr int smem = 4 * 512 * 64;
  auto kern = gemm_bf16_nt_kernel_q<OUT_F32, ACT, GM, SCHED>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(512), smem, st, p, tiles_m);
  return launch_status("gemm_bf16_q");
}

template <bool OUT_F32, int ACT, int GM, bool TRACE = false, int ABL = 0>
static int launch_tiled_s(GemmArgs& p, hipStream_t st) {
  constexpr int smem = 4 * 512 * 64;
  auto kern = gemm_bf16_nt_kernel_s<OUT_F32, ACT, GM, TRACE, ABL>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(512), smem, st, p, tiles_m, TRACE ? g_gemm_trace : nullptr);
  return launch_status("gemm_bf16_s");
}

#ifdef MMAMD_EXPERIMENTS
template <bool OUT_F32, int ACT, int GM, bool TRACE = false, int ABL = 0>
static int launch_tiled_g(GemmArgs& p, hipStream_t st) {
  if (((p.K >> 6) & 1) != 0) return launch_tiled<256, 256, 2, 4, OUT_F32, ACT, true>(p, st);
  constexpr int smem = 2 * 512 * 128;
  auto kern = gemm_bf16_nt_kernel_g<OUT_F32, ACT, GM, TRACE, ABL>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(512), smem, st, p, tiles_m, TRACE ? g_gemm_trace : nullptr);
  return launch_status("gemm_bf16_g");
}

#endif  // MMAMD_EXPERIMENTS

template <bool OUT_F32, int ACT, int GM, int WM = 2, int WN = 4, int STP = 0, int RDP = 0, int FOLD = 0, int RES_DEPTH = 1, bool BLDS = false, int A_MODE = 0>
static int launch_tiled_pp(GemmArgs& p, hipStream_t st) {
  if constexpr (A_MODE == 0) {
    if (((p.K >> 6) & 1) != 0) return launch_tiled<256, 256, 2, 4, OUT_F32, ACT, true, FOLD>(p, st);
  }
  constexpr int smem = 2 * 512 * 128 + (FOLD == 1 ? 256 * 16 * 8 : (BLDS ? 1024 : 0));  // + the tile's row statistics (nslot <= 16: K <= 1024) / bias
  if (FOLD == 1 && p.nslot_in > 16) return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, GM, 0, true, FOLD>(p, st);
  auto kern = gemm_bf16_nt_kernel_pp<OUT_F32, ACT, GM, WM, WN, STP, RDP, FOLD, RES_DEPTH, BLDS, A_MODE>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int ntiles = tiles_m * p.tiles_n;
  const int cus = stream_cus(st);                   // 256, or the CU partition of a masked stream (multiple of 8: whole XCD slices)
  const int grid = ntiles < cus ? ntiles : cus;  // one persistent workgroup per CU
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, st, p, tiles_m, ntiles);
  return launch_status("gemm_bf16_pp");
}

template <bool OUT_F32, int ACT>
static int launch_grouped(GemmGroupArgs& g, hipStream_t st) {
  constexpr int smem = 2 * 512 * 128;
  auto kern = gemm_bf16_nt_kernel_ppg<OUT_F32, ACT, 8>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int ntiles = g.tile_start[g.nprob];
  const int cus = stream_cus(st);
  const int grid = ntiles < cus ? ntiles : cus;  // one persistent workgroup per CU
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g);
  return launch_status("gemm_bf16_grouped");
}

#ifdef MMAMD_EXPERIMENTS
template <bool OUT_F32, int ACT, int GM, int SCH = 0>
static int launch_tiled_w(GemmArgs& p, hipStream_t st) {
  constexpr int smem = 3 * (256 + 128) * 64;
  auto kern = gemm_bf16_nt_kernel_w<OUT_F32, ACT, GM, SCH>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 127) / 128;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(256), smem, st, p, tiles_m);
  return launch_status("gemm_bf16_w");
}

#endif  // MMAMD_EXPERIMENTS

template <bool OUT_F32, int ACT, int FOLD = 0>
static int dispatch_variant(GemmArgs& p, hipStream_t st) {
  int v = FOLD != 0 ? 0 : g_gemm_variant;  // the LN-fold epilogues exist in the default-policy kernels only
  if (v == 0) {
    // Default policy (measured per shape with tools/kernel_bench.py):
    //  * the pipelined 256x256 kernel whenever its grid reaches a good fraction of the 256 CUs, else 128x128 tiles;
    //  * wave quantisation: one workgroup per CU, so a grid of r = tiles/256 rounds with a small fractional part pays a
    //    whole extra round (N = 768 GEMMs at B = 256: 591 tiles = 2.31 rounds -> 3).  Then the row range is split: the
    //    first floor(r) full rounds run on 256x256 tiles, the remaining rows on 128x128 tiles (4x the workgroups, ~1/4
    //    the time each), as a second launch on the same stream.
    const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + 255) / 256;
    const long t256 = (long)tiles_m * tiles_n;
    const int cus = stream_cus(st);              // CU partition of a masked stream, else 256
    const long t_eq = t256 * kChipCUs / cus;     // tile count scaled to a whole chip: the thresholds below were measured on 256 CUs
    // row-range split for the wave-quantisation tail: returns true when it launched (rc holds the status)
    int rc = 0;
    auto try_split = [&](bool big_pp) -> bool {
      const long full = t256 / cus, rem = t256 - full * cus;
      if (!(full >= 1 && full <= 4 && rem > 0 && rem <= cus / 2)) return false;
      const int m_tiles_big = (int)((full * cus) / tiles_n);  // whole row-panels that fit in the full rounds
      if (!(m_tiles_big >= 1 && m_tiles_big < tiles_m)) return false;
      GemmArgs a = p, b = p;
      const size_t rows = (size_t)m_tiles_big * 256;
      a.M = (int)rows;
      b.M = p.M - (int)rows;
      b.A = p.A + rows * p.lda;
      const size_t esz = OUT_F32 ? 4 : 2;
      b.C = reinterpret_cast<char*>(p.C) + rows * p.ldc * esz;
      if (p.R != nullptr) b.R = reinterpret_cast<const char*>(p.R) + rows * p.ldr * esz;
      if (p.C2 != nullptr) b.C2 = reinterpret_cast<char*>(p.C2) + rows * p.ldc2 * 2;
      if (p.Xh != nullptr) b.Xh = p.Xh + rows * p.ldxh;
      if (p.st_out != nullptr) b.st_out = p.st_out + rows * (size_t)(2 * p.nslot_out);
      if (p.st_in != nullptr) b.st_in = p.st_in + rows * (size_t)(2 * p.nslot_in);
      rc = big_pp ? launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, FOLD>(a, st) : launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 0, true, FOLD>(a, st);
      if (rc == 0) rc = launch_tiled<128, 128, 2, 2, OUT_F32, ACT, true, FOLD>(b, st);
      return true;
    };
    // (re-measured with warm clocks, tools/kernel_bench.py: the persistent kernel wins from ~400 tiles up at every K — out-proj 625 vs 616 vs
    //  602 TF/s for PP / P / P + split, text MLP-up 691 / 663 / 620, patch embedding 863 / 851 / 838 — and the row-range split only pays
    //  with long K: MLP-down 891 with it, 836-838 without)
    // (half a round of 256 x 256 tiles with a long K: four times as many 128 x 128 tiles balance better -- [9856 x 768 x 3072], the text MLP-down
    //  of FLAVA / CoCa at B = 128, 117 tiles: 55 us vs 63; profiles/r02_gemm_flava_coca_shapes.txt)
    if (t_eq < 96 || (t_eq < 128 && p.K >= 2048)) {
      v = 6;
    } else if ((p.K & 127) == 0 && t_eq >= 400) {
      v = 18;  // many tiles per CU: the persistent kernel hides each tile's first-stage load behind the previous epilogue
      if (p.K >= 2048 && try_split(true)) return rc;
      // the large-M and the small-M launches get different instantiations (tile-order group 8 / 4: equal speed), so that a kernel name in a
      // rocprof summary is ONE shape class — the bench's dominant kernel, gemm_bf16_nt_kernel_pp<false, QuickGELU, 8, ...>, is the ViT MLP-up only
      if (p.M < 32768) return launch_tiled_pp<OUT_F32, ACT, 4, 2, 4, OUT_F32 ? 0 : 2, 0, FOLD>(p, st);
    } else {
      v = 7;
      if (p.K >= 2048 && try_split(false)) return rc;
    }
  }
  if constexpr (FOLD != 0) {
    switch (v) {
      case 6: return launch_tiled<128, 128, 2, 2, OUT_F32, ACT, true, FOLD>(p, st);
      case 7: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 0, true, FOLD>(p, st);
      default: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, FOLD>(p, st);
    }
  } else {
    switch (v) {
      case 1: return launch_tiled<256, 256, 2, 4, OUT_F32, ACT, false>(p, st);
      case 2: return launch_tiled<128, 128, 2, 2, OUT_F32, ACT, false>(p, st);
      case 5: return launch_tiled<256, 256, 2, 4, OUT_F32, ACT, true>(p, st);
      case 6: return launch_tiled<128, 128, 2, 2, OUT_F32, ACT, true>(p, st);
      case 7: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8>(p, st);
      // bf16 C tiles are stored non-temporal (measured +6-7 % on the qkv / MLP-up GEMMs: the 128 KiB a block writes per
      // tile no longer competes with the operand panels for the XCD's L2); the in-place fp32 residual update stays plain
      case 18: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2>(p, st);
      case 70: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, 0, 1, true>(p, st);  // bias through LDS (DMA'd during the K loop)
#ifdef MMAMD_EXPERIMENTS
      // one wave per SIMD (4 waves, ONE workgroup per CU by LDS: 96 KiB ring), the 8-wave kernels' 128 x 64 wave tile: what a kernel with a
      // second accumulator set (512 registers per wave) would have as its main loop
      case 81: return launch_tiled<256, 128, 2, 2, OUT_F32, ACT, true>(p, st);
      case 82: return launch_tiled<128, 256, 1, 4, OUT_F32, ACT, true>(p, st);
      case 60: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, 0, 1>(p, st);  // fp32 residual prefetch ring depth 1 .. 4
      case 61: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, 0, 2>(p, st);
      case 62: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, 0, 3>(p, st);
      case 63: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 0, 0, 4>(p, st);
      case 50: return launch_tiled_w<OUT_F32, ACT, 8>(p, st);   // two workgroups per CU, 256 x 128 tiles, BK = 32 ring of 3
      case 51: return launch_tiled_w<OUT_F32, ACT, 8, 1>(p, st);
      case 52: return launch_tiled_w<OUT_F32, ACT, 8, 2>(p, st);
#endif
#ifdef MMAMD_EXPERIMENTS
      case 20: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, 1>(p, st);  // C stores sc1 (write-through, not kept in L2)
      case 21: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, 2>(p, st);  // C stores nt
      case 22: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, 0>(p, st);  // C stores plain
      case 23: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 1>(p, st);  // fragment reads in one burst
      case 24: return launch_tiled_pp<OUT_F32, ACT, 8, 2, 4, OUT_F32 ? 0 : 2, 2>(p, st);  // fragment reads 2 per MFMA
      case 25: return launch_tiled_pp<OUT_F32, ACT, 4, 2, 4, OUT_F32 ? 0 : 2>(p, st);   // tile-order group of 4 row panels
      case 26: return launch_tiled_pp<OUT_F32, ACT, 16, 2, 4, OUT_F32 ? 0 : 2>(p, st);  // ... 16
      case 27: return launch_tiled_pp<OUT_F32, ACT, 2, 2, 4, OUT_F32 ? 0 : 2>(p, st);   // ... 2
#endif
#ifdef MMAMD_EXPERIMENTS  // schedule experiments, ablations (WRONG results for 1xx except 132/164) and traces: see DESIGN.md 4.1
      case 30: return launch_tiled_g<OUT_F32, ACT, 8>(p, st);        // ping-pong kernel "G" (measured: not faster than P/PP)
      case 31: return launch_tiled_g<OUT_F32, ACT, 8, true>(p, st);  // ping-pong kernel with s_memtime stamps
      case 34: return launch_tiled_g<OUT_F32, ACT, 8, false, 4>(p, st);   // G ablations: no epilogue
      case 35: return launch_tiled_g<OUT_F32, ACT, 8, false, 5>(p, st);   //   no epilogue, no DMA
      case 36: return launch_tiled_g<OUT_F32, ACT, 8, false, 12>(p, st);  //   no epilogue, no fragment reads
      case 37: return launch_tiled_g<OUT_F32, ACT, 8, false, 6>(p, st);   //   no epilogue, no MFMA
      case 38: return launch_tiled_g<OUT_F32, ACT, 8, false, 13>(p, st);  //   MFMA + barriers only
      // (tried and removed: the same kernel as 4 waves x (128 x 128) — one wave per SIMD, 256 AGPR accumulators, a third less LDS read
      //  traffic per MFMA: its main loop alone ran at 1187 TF/s-equivalent against 1300 for the 8-wave form on the qkv shape)
      case 3: return launch_tiled<256, 128, 4, 2, OUT_F32, ACT, false>(p, st);
      case 4: return launch_tiled<128, 256, 2, 4, OUT_F32, ACT, false>(p, st);
      case 9: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 0, false>(p, st);  // direct-store epilogue
      case 10: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 1>(p, st);
      case 11: return launch_tiled_q<OUT_F32, ACT, 8>(p, st);
      case 12: return launch_tiled_q<OUT_F32, ACT, 8, 1>(p, st);
      case 13: return launch_tiled_s<OUT_F32, ACT, 8>(p, st);
      case 14: return launch_tiled_s<OUT_F32, ACT, 8, true>(p, st);
      case 15: return launch_tiled_s<OUT_F32, ACT, 8, true, 1>(p, st);
      case 16: return launch_tiled_s<OUT_F32, ACT, 8, true, 8>(p, st);
      case 101: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 1>(p, st);
      case 102: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 2>(p, st);
      case 104: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 4>(p, st);
      case 105: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 5>(p, st);
      case 106: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 6>(p, st);
      case 108: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 8>(p, st);
      case 112: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 12>(p, st);
      case 113: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 13>(p, st);
      case 114: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 14>(p, st);
      case 116: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 16>(p, st);
      case 132: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 32>(p, st);
      case 164: return launch_tiled_p<256, 256, 2, 4, OUT_F32, ACT, 8, 64>(p, st);
#endif
      default: set_error("gemm: unknown variant %d (experimental variants need -DMMAMD_EXPERIMENTS)", v); return MMAMD_E_BADARG;
    }
  }
}

template <bool OUT_F32>
static int dispatch(GemmArgs& p, hipStream_t st) {
  if (g_gemm_variant == 99) {
    hipLaunchKernelGGL((gemm_naive_kernel<OUT_F32>), dim3((p.N + 63) / 64, (p.M + 3) / 4), dim3(256), 0, st, p);
    return launch_status("gemm_naive");
  }
  if constexpr (OUT_F32) {
    if (p.Xh != nullptr) return dispatch_variant<true, MMAMD_ACT_NONE, 2>(p, st);  // LN fold, producer (host: act == NONE)
  } else {
    if (p.st_in != nullptr) {  // LN fold, consumer
      switch (p.act) {
        case MMAMD_ACT_NONE: return dispatch_variant<false, MMAMD_ACT_NONE, 1>(p, st);
        case MMAMD_ACT_QUICKGELU: return dispatch_variant<false, MMAMD_ACT_QUICKGELU, 1>(p, st);
        default: return dispatch_variant<false, MMAMD_ACT_GELU_ERF, 1>(p, st);
      }
    }
  }
  switch (p.act) {
    case MMAMD_ACT_NONE: return dispatch_variant<OUT_F32, MMAMD_ACT_NONE>(p, st);
    case MMAMD_ACT_QUICKGELU: return dispatch_variant<OUT_F32, MMAMD_ACT_QUICKGELU>(p, st);
    default: return dispatch_variant<OUT_F32, MMAMD_ACT_GELU_ERF>(p, st);
  }
}

}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_set_gemm_variant(int variant) {
  g_gemm_variant = variant;
  return 0;
}
extern "C" int mmamd_get_gemm_variant(void) { return g_gemm_variant; }
extern "C" int mmamd_debug_set_gemm_stagger(int percent) {
  g_gemm_stagger = percent < 0 ? 0 : percent;
  return 0;
}

extern "C" int mmamd_debug_set_gemm_trace(void* buf) {
  g_gemm_trace = reinterpret_cast<unsigned long long*>(buf);
  return 0;
}

struct LnFoldArgs {  // see "LN fold" at the top of this file
  void* xh = nullptr; int ldxh = 0; float* st_out = nullptr;                         // producer
  const float* st_in = nullptr; int nslot_in = 0; const float* c1 = nullptr; float eps = 0.f;  // consumer
};
static int gemm_bf16_impl(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual, int ldr, void* C,
                          int ldc, int out_dtype, int M, int N, int K, int act, void* C2, int ldc2, int act2, mmamd_stream_t stream,
                          const LnFoldArgs* lf = nullptr);

extern "C" int mmamd_gemm_bf16_res_stats(const void* A, int lda, const void* W, int ldw, const float* bias, const float* residual, int ldr,
                                         float* C, int ldc, void* Xh, int ldxh, float* stats, int M, int N, int K, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(Xh && stats, MMAMD_E_BADARG, "gemm_res_stats: null output");
  MMAMD_CHECK_ARG(N % 128 == 0, MMAMD_E_UNSUPPORTED, "gemm_res_stats: N=%d must be a multiple of 128 (whole 64-column statistic blocks, even count)", N);
  MMAMD_CHECK_ARG(ldxh >= N && ldxh % 4 == 0 && (reinterpret_cast<uintptr_t>(Xh) & 7) == 0 && (reinterpret_cast<uintptr_t>(stats) & 15) == 0,
                  MMAMD_E_ALIGN, "gemm_res_stats: bf16 copy needs 8-byte aligned rows, the statistics 16-byte alignment");
  LnFoldArgs lf;
  lf.xh = Xh; lf.ldxh = ldxh; lf.st_out = stats;
  return gemm_bf16_impl(A, lda, W, ldw, bias, residual, ldr, C, ldc, MMAMD_F32, M, N, K, MMAMD_ACT_NONE, nullptr, 0, 0, stream, &lf);
}

extern "C" int mmamd_gemm_bf16_lnfold(const void* Xh, int lda, const void* Wg, int ldw, const float* c1, const float* c2, const float* stats,
                                      int nslot, float eps, void* C, int ldc, int M, int N, int K, int act, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(c1 && c2 && stats, MMAMD_E_BADARG, "gemm_lnfold: null argument");
  MMAMD_CHECK_ARG(nslot > 0 && nslot % 2 == 0 && nslot * 64 == K, MMAMD_E_BADARG,
                  "gemm_lnfold: %d statistic blocks of 64 columns do not cover the normalised width K=%d", nslot, K);
  MMAMD_CHECK_ARG(aligned16(c1) && aligned16(stats), MMAMD_E_ALIGN, "gemm_lnfold: c1 / stats must be 16-byte aligned");
  MMAMD_CHECK_ARG(act == MMAMD_ACT_NONE || act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF, MMAMD_E_BADARG, "gemm_lnfold: bad activation code %d", act);
  LnFoldArgs lf;
  lf.st_in = stats; lf.nslot_in = nslot; lf.c1 = c1; lf.eps = eps;
  return gemm_bf16_impl(Xh, lda, Wg, ldw, c2, nullptr, 0, C, ldc, MMAMD_BF16, M, N, K, act, nullptr, 0, 0, stream, &lf);
}

extern "C" int mmamd_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual,
                               int ldr, void* C, int ldc, int out_dtype, int M, int N, int K, int act,
                               mmamd_stream_t stream) {
  return gemm_bf16_impl(A, lda, W, ldw, bias, residual, ldr, C, ldc, out_dtype, M, N, K, act, nullptr, 0, 0, stream);
}

extern "C" int mmamd_patch_embed_gemm(const void* image, const void* W, int ldw, const float* pos, float* x, int B, int patch, int image_size,
                                      int width, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(image && W && pos && x && B >= 0 && width > 0, MMAMD_E_BADARG, "patch_embed_gemm: bad argument");
  MMAMD_CHECK_ARG((patch == 16 || patch == 32) && image_size > 0 && image_size % patch == 0 && image_size % 8 == 0, MMAMD_E_UNSUPPORTED,
                  "patch_embed_gemm: patch %d / image %d (16- and 32-pixel patches of an image side that is a multiple of 8)", patch, image_size);
  MMAMD_CHECK_ARG(width % 8 == 0 && ldw >= 3 * patch * patch && ldw % 8 == 0, MMAMD_E_UNSUPPORTED, "patch_embed_gemm: width %d / ldw %d", width, ldw);
  MMAMD_CHECK_ARG(aligned16(image) && aligned16(W) && aligned16(pos) && aligned16(x), MMAMD_E_ALIGN, "patch_embed_gemm: pointers must be 16-byte aligned");
  const int g = image_size / patch;
  MMAMD_CHECK_ARG((uint64_t)B * 3u * image_size * image_size * 2u < (1ull << 32) && (uint64_t)width * ldw * 2u < (1ull << 32), MMAMD_E_UNSUPPORTED,
                  "patch_embed_gemm: operand exceeds the 4 GiB 32-bit DMA offset range");
  if (B == 0) return 0;
  GemmArgs p;
  p.A = (const bf16*)image; p.W = (const bf16*)W; p.bias = nullptr; p.R = pos; p.C = x;
  p.M = B * g * g; p.N = width; p.K = 3 * patch * patch; p.lda = p.K; p.ldw = ldw; p.ldr = width; p.ldc = width; p.act = MMAMD_ACT_NONE; p.tiles_n = 0;
  p.kt_chunk = 0; p.c_split_stride = 0; p.res_mode = 0; p.C2 = nullptr; p.ldc2 = 0; p.act2 = 0; p.split_flat = 0; p.stagger = 0;
  p.i2c_g2 = g * g; p.i2c_g = g; p.i2c_p = patch; p.i2c_hw = image_size;
  p.i2c_lcr = patch == 16 ? 1 : 2; p.i2c_ltpc = patch == 16 ? 2 : 4; p.i2c_rpk = 64 / patch;
  return launch_tiled_pp<true, MMAMD_ACT_NONE, 8, 2, 4, 0, 0, 0, 1, false, 1>(p, (hipStream_t)stream);
}

extern "C" int mmamd_gemm_bf16_grouped(const mmamd_gemm_problem* probs, int nprob, int out_dtype, int act, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(probs != nullptr && nprob >= 1 && nprob <= 2, MMAMD_E_BADARG, "gemm_grouped: 1 or 2 problems, got %d", nprob);
  MMAMD_CHECK_ARG(out_dtype == MMAMD_F32 || out_dtype == MMAMD_BF16, MMAMD_E_BADARG, "gemm_grouped: bad out_dtype %d", out_dtype);
  MMAMD_CHECK_ARG(act == MMAMD_ACT_NONE || act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF, MMAMD_E_BADARG, "gemm_grouped: bad activation code %d", act);
  // one persistent launch needs every problem on the persistent kernel's K granularity (two 64-deep K-tiles per loop trip) and enough tiles
  // in total to give each CU more than one; anything else runs as the separate launches the grouped call stands for (same results: a tile's
  // arithmetic does not depend on which launch computes it -- tests/test_gpu_grouped_gemm.py)
  bool group = nprob == 2 && g_gemm_variant == 0;
  long tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const mmamd_gemm_problem& q = probs[i];
    MMAMD_CHECK_ARG(q.M >= 0 && q.N > 0 && q.K > 0 && q.W && (q.M == 0 || (q.A && q.C)), MMAMD_E_BADARG, "gemm_grouped: problem %d: bad argument", i);
    if ((q.K & 127) != 0 || q.M == 0) group = false;
    tiles += (long)((q.M + 255) / 256) * ((q.N + 255) / 256);
  }
  hipStream_t st = (hipStream_t)stream;
  if (!group || tiles < 2L * stream_cus(st)) {
    for (int i = 0; i < nprob; ++i) {
      const mmamd_gemm_problem& q = probs[i];
      if (q.M == 0) continue;  // an empty problem (its pointers may be NULL)
      if (int rc = gemm_bf16_impl(q.A, q.lda, q.W, q.ldw, q.bias, q.R, q.ldr, q.C, q.ldc, out_dtype, q.M, q.N, q.K, act, nullptr, 0, 0, stream)) return rc;
    }
    return 0;
  }
  GemmGroupArgs g;
  g.nprob = nprob;
  g.tile_start[0] = 0;
  for (int i = 0; i < nprob; ++i) {
    const mmamd_gemm_problem& q = probs[i];
    MMAMD_CHECK_ARG(q.N % 8 == 0, MMAMD_E_UNSUPPORTED, "gemm_grouped: N=%d must be a multiple of 8", q.N);
    MMAMD_CHECK_ARG(q.lda >= q.K && q.ldw >= q.K && q.ldc >= q.N && (!q.R || q.ldr >= q.N), MMAMD_E_BADARG, "gemm_grouped: leading dimension too small");
    MMAMD_CHECK_ARG(q.lda % 8 == 0 && q.ldw % 8 == 0 && q.ldc % 8 == 0 && (!q.R || q.ldr % 8 == 0), MMAMD_E_ALIGN,
                    "gemm_grouped: leading dimensions must be multiples of 8 elements");
    MMAMD_CHECK_ARG(aligned16(q.A) && aligned16(q.W) && aligned16(q.C) && aligned16(q.R) && aligned16(q.bias), MMAMD_E_ALIGN,
                    "gemm_grouped: base pointers must be 16-byte aligned");
    MMAMD_CHECK_ARG((uint64_t)q.M * (uint64_t)q.lda * 2u < (1ull << 32) && (uint64_t)q.N * (uint64_t)q.ldw * 2u < (1ull << 32),
                    MMAMD_E_UNSUPPORTED, "gemm_grouped: operand exceeds the 4 GiB 32-bit DMA offset range");
    GemmProblem& d = g.prob[i];
    d.A = (const bf16*)q.A; d.W = (const bf16*)q.W; d.bias = q.bias; d.R = q.R; d.C = q.C;
    d.M = q.M; d.N = q.N; d.K = q.K; d.lda = q.lda; d.ldw = q.ldw; d.ldr = q.ldr; d.ldc = q.ldc;
    d.tiles_m = (q.M + 255) / 256; d.tiles_n = (q.N + 255) / 256;
    g.tile_start[i + 1] = g.tile_start[i] + d.tiles_m * d.tiles_n;
  }
  for (int i = nprob; i < 2; ++i) { g.prob[i] = g.prob[0]; g.tile_start[i + 1] = g.tile_start[nprob]; }
  {
    const long long t_tile = (long long)(probs[0].K / 64) * 3500 + (out_dtype == MMAMD_F32 ? 27000 : 8000) + (act != MMAMD_ACT_NONE ? 8000 : 0);
    g.stagger = (int)(t_tile * (g_gemm_stagger % 1000) / 100);
  }
  if (out_dtype == MMAMD_F32) {
    switch (act) {
      case MMAMD_ACT_NONE: return launch_grouped<true, MMAMD_ACT_NONE>(g, st);
      case MMAMD_ACT_QUICKGELU: return launch_grouped<true, MMAMD_ACT_QUICKGELU>(g, st);
      default: return launch_grouped<true, MMAMD_ACT_GELU_ERF>(g, st);
    }
  }
  switch (act) {
    case MMAMD_ACT_NONE: return launch_grouped<false, MMAMD_ACT_NONE>(g, st);
    case MMAMD_ACT_QUICKGELU: return launch_grouped<false, MMAMD_ACT_QUICKGELU>(g, st);
    default: return launch_grouped<false, MMAMD_ACT_GELU_ERF>(g, st);
  }
}

extern "C" int mmamd_gemm_bf16_dual(const void* A, int lda, const void* W, int ldw, const float* bias, void* U, int ldu, void* G,
                                    int ldg, int M, int N, int K, int act, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(U && G, MMAMD_E_BADARG, "gemm_dual: null output");
  MMAMD_CHECK_ARG(act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF, MMAMD_E_BADARG, "gemm_dual: bad activation code %d", act);
  MMAMD_CHECK_ARG(ldg >= N && ldg % 8 == 0 && aligned16(G), MMAMD_E_ALIGN, "gemm_dual: second output must be 16-byte aligned with ldg %% 8 == 0");
  return gemm_bf16_impl(A, lda, W, ldw, bias, nullptr, 0, U, ldu, MMAMD_BF16, M, N, K, MMAMD_ACT_NONE, G, ldg, act, stream);
}

static int gemm_bf16_impl(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual, int ldr, void* C,
                          int ldc, int out_dtype, int M, int N, int K, int act, void* C2, int ldc2, int act2, mmamd_stream_t stream,
                          const LnFoldArgs* lf) {
  MMAMD_CHECK_ARG(A && W && C, MMAMD_E_BADARG, "gemm: null pointer");
  MMAMD_CHECK_ARG(M >= 0 && N > 0 && K > 0, MMAMD_E_BADARG, "gemm: bad sizes M=%d N=%d K=%d", M, N, K);
  MMAMD_CHECK_ARG(K % 64 == 0, MMAMD_E_UNSUPPORTED, "gemm: K=%d must be a multiple of 64 (pad the operands)", K);
  MMAMD_CHECK_ARG(N % 8 == 0, MMAMD_E_UNSUPPORTED, "gemm: N=%d must be a multiple of 8", N);
  MMAMD_CHECK_ARG(lda >= K && ldw >= K && ldc >= N && (!residual || ldr >= N), MMAMD_E_BADARG, "gemm: leading dimension too small");
  MMAMD_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), MMAMD_E_ALIGN,
                  "gemm: leading dimensions must be multiples of 8 elements");
  MMAMD_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(residual) && aligned16(bias), MMAMD_E_ALIGN,
                  "gemm: base pointers must be 16-byte aligned");
  MMAMD_CHECK_ARG((uint64_t)M * (uint64_t)lda * 2u < (1ull << 32) && (uint64_t)N * (uint64_t)ldw * 2u < (1ull << 32),
                  MMAMD_E_UNSUPPORTED, "gemm: operand exceeds the 4 GiB 32-bit DMA offset range");
  MMAMD_CHECK_ARG(act >= MMAMD_ACT_NONE && act <= MMAMD_ACT_MUL_GELU_GRAD, MMAMD_E_BADARG, "gemm: bad activation code %d", act);
  if (M == 0) return 0;
  GemmArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.R = residual; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc; p.act = act; p.tiles_n = 0;
  p.kt_chunk = 0; p.c_split_stride = 0; p.res_mode = 0;
  p.C2 = C2; p.ldc2 = ldc2; p.act2 = act2; p.split_flat = 0;
  {
    // start-up stagger of the persistent kernel as a fraction (g_gemm_stagger, per cent) of the estimated tile time in shader ticks:
    // ~3500 ticks per 64-deep K-tile + the epilogue (bf16 tile ~8k, + QuickGELU / erf-GELU ~8k, fp32 + residual ~27k)
    const long long t_tile = (long long)(K / 64) * 3500 + (out_dtype == MMAMD_F32 ? 27000 : 8000) +
                             ((act == MMAMD_ACT_QUICKGELU || act == MMAMD_ACT_GELU_ERF) ? 8000 : 0);
    p.stagger = g_gemm_stagger >= 1000 ? -(int)(t_tile * (g_gemm_stagger - 1000) / 100) : (int)(t_tile * g_gemm_stagger / 100);
  }
  if (lf != nullptr) {
    p.Xh = (bf16*)lf->xh; p.ldxh = lf->ldxh; p.st_out = lf->st_out; p.nslot_out = N / 64;
    p.st_in = lf->st_in; p.nslot_in = lf->nslot_in; p.c1 = lf->c1; p.inv_d = 1.0f / (float)K; p.ln_eps = lf->eps;
  }
  if (act == MMAMD_ACT_MUL_QUICKGELU_GRAD || act == MMAMD_ACT_MUL_GELU_GRAD) {
    MMAMD_CHECK_ARG(out_dtype == MMAMD_BF16 && residual != nullptr, MMAMD_E_BADARG,
                    "gemm: the activation-gradient epilogue needs bf16 output and the saved pre-activation as `residual`");
    p.res_mode = act == MMAMD_ACT_MUL_QUICKGELU_GRAD ? 1 : 2;
    p.act = MMAMD_ACT_NONE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MMAMD_F32) return dispatch<true>(p, st);
  if (out_dtype == MMAMD_BF16) return dispatch<false>(p, st);
  MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "gemm: bad out_dtype %d", out_dtype);
}

// column-wise sum of `splits` partial outputs (second stage of the split-K weight-gradient GEMM)
namespace mmamd {
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, long long n, float* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 acc = load4(part + i);
  for (int s = 1; s < splits; ++s) {
    const f32x4 v = load4(part + (size_t)s * n + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
  store4(out + i, acc);
}
}  // namespace mmamd

// FLAT (default): 1-D grid of tiles * splits, split-major after the XCD-contiguous remap — an XCD then runs neighbouring tiles of ONE split,
// which stream the same contraction rows at the same time and share operand panels in its L2 (the 2-D grid scattered a split's tiles over all
// XCDs: PMC FETCH_SIZE 3x the operand bytes on the MLP-up gradient).  Measured (tools/wgrad_bench.py --sched): -3...-10 % on every shape.
template <bool TNM, int SCH = 0, int GMV = 8, bool FLAT = true>
static int gemm_splitk_impl(const void* A, int lda, const void* W, int ldw, float* C, float* ws, int M, int N, int K, int splits,
                            mmamd_stream_t stream) {
  const int KT = K / 64;
  int chunk = (KT + splits - 1) / splits;
  chunk += chunk & 1;  // even number of K-tiles per split (the K loop is unrolled by two); KT is even, so is the remainder
  const int nsplit = (KT + chunk - 1) / chunk;
  GemmArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = nullptr; p.R = nullptr; p.C = nsplit == 1 ? C : ws;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldr = 0; p.ldc = N; p.act = MMAMD_ACT_NONE;
  p.kt_chunk = chunk; p.c_split_stride = (long long)M * N; p.res_mode = 0;
  p.C2 = nullptr; p.ldc2 = 0; p.act2 = 0; p.split_flat = FLAT ? 1 : 0;
  constexpr int smem = 2 * 512 * 128;
  auto kern = gemm_bf16_nt_kernel_p<256, 256, 2, 4, true, MMAMD_ACT_NONE, GMV, 0, true, TNM, SCH>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (M + 255) / 256;
  p.tiles_n = (N + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  if (FLAT) hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n * nsplit), dim3(512), smem, st, p, tiles_m, nullptr);
  else hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n, nsplit), dim3(512), smem, st, p, tiles_m, nullptr);
  if (nsplit > 1) {
    const long long n = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, ws, nsplit, n, C);
  }
  return launch_status("gemm_bf16_splitk");
}

extern "C" int mmamd_gemm_bf16_splitk(const void* A, int lda, const void* W, int ldw, float* C, float* ws, int M, int N, int K,
                                      int splits, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(A && W && C && ws && M > 0 && N > 0 && K > 0 && splits >= 1, MMAMD_E_BADARG, "gemm_splitk: bad argument");
  MMAMD_CHECK_ARG(K % 128 == 0, MMAMD_E_UNSUPPORTED, "gemm_splitk: K=%d must be a multiple of 128 (pad the operands)", K);
  MMAMD_CHECK_ARG(N % 8 == 0 && (M * (long long)N) % 4 == 0, MMAMD_E_UNSUPPORTED, "gemm_splitk: N=%d must be a multiple of 8", N);
  MMAMD_CHECK_ARG(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, MMAMD_E_BADARG, "gemm_splitk: bad leading dimension");
  MMAMD_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(ws), MMAMD_E_ALIGN, "gemm_splitk: base pointers must be 16-byte aligned");
  MMAMD_CHECK_ARG((uint64_t)M * (uint64_t)lda * 2u < (1ull << 32) && (uint64_t)N * (uint64_t)ldw * 2u < (1ull << 32),
                  MMAMD_E_UNSUPPORTED, "gemm_splitk: operand exceeds the 4 GiB 32-bit DMA offset range");
  return gemm_splitk_impl<false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);
}

extern "C" int mmamd_gemm_bf16_tn_splitk(const void* A, int lda, const void* W, int ldw, float* C, float* ws, int M, int N, int K,
                                         int splits, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(A && W && C && ws && M > 0 && N > 0 && K > 0 && splits >= 1, MMAMD_E_BADARG, "gemm_tn_splitk: bad argument");
  MMAMD_CHECK_ARG(K % 128 == 0, MMAMD_E_UNSUPPORTED, "gemm_tn_splitk: contraction length K=%d must be a multiple of 128", K);
  MMAMD_CHECK_ARG(M % 8 == 0 && N % 8 == 0, MMAMD_E_UNSUPPORTED, "gemm_tn_splitk: M=%d and N=%d must be multiples of 8", M, N);
  MMAMD_CHECK_ARG(lda >= M && ldw >= N && lda % 8 == 0 && ldw % 8 == 0, MMAMD_E_BADARG, "gemm_tn_splitk: bad leading dimension");
  MMAMD_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(ws), MMAMD_E_ALIGN, "gemm_tn_splitk: base pointers must be 16-byte aligned");
  MMAMD_CHECK_ARG((uint64_t)64 * (uint64_t)lda * 2u < (1ull << 31) && (uint64_t)64 * (uint64_t)ldw * 2u < (1ull << 31), MMAMD_E_UNSUPPORTED,
                  "gemm_tn_splitk: leading dimension too large for the 32-bit DMA offsets");
#ifdef MMAMD_EXPERIMENTS  // fragment-read placement experiments of the TN main loop (mmamd_set_gemm_variant(40 .. 43))
  if (g_gemm_variant == 40) return gemm_splitk_impl<true, 0, 8, false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);  // 2-D grid (split = blockIdx.y)
  if (g_gemm_variant == 41) return gemm_splitk_impl<true, 1, 8, false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);
  if (g_gemm_variant == 42) return gemm_splitk_impl<true, 2, 8, false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);
  if (g_gemm_variant == 43) return gemm_splitk_impl<true, 3, 8, false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);
  if (g_gemm_variant == 44) return gemm_splitk_impl<true, 0, 1, false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);  // tile order: column tile innermost
  if (g_gemm_variant == 45) return gemm_splitk_impl<true, 0, 2, false>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);
  if (g_gemm_variant == 46) return gemm_splitk_impl<true, 0, 1, true>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);  // flat grid, split-major, column tile innermost
  if (g_gemm_variant == 47) return gemm_splitk_impl<true, 0, 8, true>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);  // flat grid, row tile innermost
#endif
  // (fragment-read placement variants 40-43 differ by less than the run-to-run spread of a 20-launch loop — the same kernel measured 295 and
  //  252 us depending on its position in the loop — and the training step time is unchanged by them: the MFMA-first order stays)
  return gemm_splitk_impl<true>(A, lda, W, ldw, C, ws, M, N, K, splits, stream);
}
