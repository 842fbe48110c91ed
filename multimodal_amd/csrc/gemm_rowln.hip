// gemm_rowln.hip — out-projection + residual + the NEXT LayerNorm in one launch (r06).
//
//     X[M, N] (fp32, in place) = A[M, K] (bf16) . W[N, K]^T (bf16) + bias[N] + X          -- the attention block's residual update
//     Y[M, N] (bf16)           = LayerNorm(X row; gamma, beta, eps)                        -- norm2, the MLP's input
//
// Why: in the headline step the out-projection is HBM-bound, not MFMA-bound (69.8 GFLOP in ~112 us for both towers: 490 MB at the ~5 TB/s a
// read+write stream reaches), and the LayerNorm behind it (49 us) spends its time re-reading the 195 MB of fp32 X the GEMM has just written.  A
// workgroup that owns WHOLE ROWS (a 64 x N tile, N = 768 / 512) has the row statistics in its registers when the tile is done, writes Y next to X
// and the re-read disappears: 584 MB instead of 490 + 292.  The price is operand reuse -- W (N x K bf16, 1.2 MB) is streamed from the L2 once per
// 64-row tile instead of once per 256-row tile (~1.1 GB of L2 -> LDS DMA per launch, at 35 TB/s peak, 10 pJ/B) -- which only a GEMM that is
// nowhere near the matrix pipe's limit can afford: out-projections (K = N), not the MLP's (K = 4 N).
//
// Structure (one workgroup = 8 waves per CU, persistent over its round-robin share of the tiles of up to two problems):
//   * K loop in steps of 32: a stage = 64 x 32 of A (4 KiB) + N x 32 of W (48 KiB at N = 768), `global_load_lds_dwordx4` straight into a
//     three-slot LDS ring (156 KiB), 64-byte rows with the 16-byte chunk position XORed by (row >> 2) & 3 on the SOURCE side (conflict-free
//     ds_read_b128 fragments).  The ring runs ACROSS tiles: the first two stages of the next tile are in flight during a tile's epilogue.
//     One barrier per step; counted `s_waitcnt vmcnt` (in-order queue).
//   * waves 2 (32-row halves) x 4 (N / 4 column slabs); v_mfma_f32_32x32x16_bf16 with the operands swapped (lane = row of A, registers = columns),
//     the same k order as the persistent GEMM kernels of gemm.hip: X is BIT-IDENTICAL to mmamd_gemm_bf16's.
//   * epilogue: accumulators -> wave-private LDS strip (inside the ring slot the tile's last stage has just left) -> whole 128-byte row segments;
//     + bias + residual (buffer descriptors, a two-pass load window), X stored, the lane keeps its 4 rows x (N / 32) x 4 values; two-pass row
//     statistics (8-lane butterflies, then the four column slabs through LDS); Y stored as bf16.
// LayerNorm arithmetic as rowops.hip: mean = sum / N, rstd = 1 / sqrtf(sum((x - mean)^2) / N + eps), y = (x - mean) * rstd * gamma + beta; only the
// summation ORDER of the two statistics differs from the wave-per-row kernel (fp32 rounding, ~1e-7 relative).
// Replaces out_proj + residual + norm2 of nn.TransformerEncoderLayer(norm_first=True) (reference models/clip/image_encoder.py:108,
// models/clip/text_encoder.py:121).
#include <type_traits>

#include "common.h"

namespace mmamd {

typedef uint32_t __attribute__((address_space(3))) * rl_lds_u32p;
typedef __attribute__((ext_vector_type(4))) uint32_t rl_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t rl_u32x2;

struct RowLnProb {
  const bf16* A;
  const bf16* W;  // K-step-major: [K / 32][N][32] (mmamd_pack_w_ksteps)
  const float* bias;
  float* X;
  const float* gamma;
  const float* beta;
  bf16* Y;
  int M, N, K;
  float eps;
  int tiles;  // ceil(M / 64)
  int pad_;
};
struct RowLnArgs {
  RowLnProb p[2];
  int nprob;
  int tiles_total;
};

constexpr int kRlBM = 64, kRlBK = 32, kRlMaxN = 768;
constexpr int kRlSlot = kRlBM * 64 + kRlMaxN * 64;  // 53248 B: A rows then W rows, 64 B (32 bf16) each
constexpr int kRlSlots = 3;
constexpr int kRlStripRow = 144;                      // 32 fp32 + 16 B pad: conflict-free b128 writes by row
constexpr int kRlStrip = 32 * kRlStripRow;            // per wave
constexpr int kRlRed = 8 * kRlStrip;                  // row-statistics exchange behind the eight strips: 2 x [64 rows][4 slabs] floats
constexpr int kRlPar = kRlRed + 2 * 64 * 4 * 4;     // gamma [N], then beta at + 4 kRlMaxN, DMA'd per tile (an L2 hit): the Y loop reads them without a vector-memory wait
static_assert(kRlPar + 2 * kRlMaxN * 4 <= kRlSlot, "strips + statistics + LayerNorm parameters fit the free ring slot");

__device__ __forceinline__ void rl_dma(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// wait until at most n (0..7, wave-uniform) of this wave's vector-memory operations are outstanding
__device__ __forceinline__ void rl_wait_le(int n) {
  switch (n) {
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__device__ __forceinline__ void rl_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// sum over the 8 lanes that share a row of the row-major epilogue layout (lane & 7 = 16-byte column group)
__device__ __forceinline__ float rl_sum8(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

// ABL (timing experiments, results WRONG; mmamd_debug_set_gemm_knob(5, code)): 1 = no X / Y stores, 2 = no residual loads, 4 = no operand DMA, 8 = no fragment reads / MFMA, 16 = K walk rotated per workgroup
template <int ABL>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(const RowLnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(rl_lds_u32p)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int rg = wave & 1, cg = wave >> 1;
  const int nmine = (int)blockIdx.x < g.tiles_total ? (g.tiles_total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (nmine == 0) return;
  const int tiles0 = g.p[0].tiles;

  // ---- issue cursor: the next stage to DMA (tile index ij of this workgroup, K step ik) ----------------------------------------------------
  int ij = 0, ik = 0, islot = 0;
  int i_pi = 0, i_row0 = 0, i_ks = 0, i_pw = 0, i_K2 = 0;
  const char* i_a = nullptr;   // A + row0 * K * 2 (tile base)
  const char* i_w = nullptr;   // W + (this wave's first W row) * K * 2
  uint32_t i_voff_a = 0;
  int i_wstep = 0;
  const int dr = lane >> 2;                                           // row of the lane inside a 16-row piece
  const uint32_t dchunk = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);  // source chunk of the lane's 16 bytes (swizzle undone on the read side)
  const uint32_t voff_w = (uint32_t)(dr * 64) + dchunk;  // W is packed by K step: a piece is 1 KiB of consecutive bytes (whole cache lines)
  auto cursor_tile = [&]() {  // derive the issue cursor's per-tile values from ij
    const int t = (int)blockIdx.x + ij * (int)gridDim.x;
    i_pi = (g.nprob > 1 && t >= tiles0) ? 1 : 0;
    const RowLnProb& P = g.p[i_pi];
    i_row0 = (t - (i_pi ? tiles0 : 0)) * kRlBM;
    i_ks = P.K / kRlBK;
    i_pw = P.N >> 7;  // W pieces (16 rows each) per wave per stage: N / 16 / 8
    i_K2 = P.K * 2;
    i_a = reinterpret_cast<const char*>(P.A) + (size_t)i_row0 * i_K2;
    i_w = reinterpret_cast<const char*>(P.W) + (size_t)(wave * i_pw * 16) * 64;
    i_wstep = P.N * 64;  // bytes of one K step's W slice
    int ar = wave * 16 + dr;  // rows past M: clamped to the last row (never stored)
    if (i_row0 + ar >= P.M) ar = P.M - 1 - i_row0;
    i_voff_a = (uint32_t)(ar * i_K2) + dchunk;
  };
  auto issue = [&]() -> int {  // DMA this wave's pieces of the cursor's stage; returns their number (0: nothing left)
    if (ij >= nmine) return 0;
    const uint32_t dst = lds0 + (uint32_t)(islot * kRlSlot);
    // ABL & 16 (experiment): every workgroup starts its K walk at a different step (the 32 CUs of an XCD otherwise sweep the SAME W slices through the
    // same L2 channels at the same time); the accumulation order rotates with it, so X is no longer bit-identical to mmamd_gemm_bf16's
    int ike = ik;
    if constexpr ((ABL & 16) != 0) {
      ike = ik + (int)(((blockIdx.x >> 3) * (unsigned)i_ks) >> 5);
      ike = ike >= i_ks ? ike - i_ks : ike;
    }
    const char* ws = i_w + (size_t)ike * i_wstep;
    int n = i_pw;
#pragma unroll
    for (int i = 0; i < kRlMaxN / 128; ++i)
      if (i < i_pw && (ABL & 4) == 0) rl_dma(ws + i * 1024, voff_w, dst + 4096u + (uint32_t)((wave * i_pw + i) * 1024));
    if (wave < 4) {
      if constexpr ((ABL & 4) == 0) rl_dma(i_a + ike * 64, i_voff_a, dst + (uint32_t)(wave * 1024));
      ++n;
    }
    islot = islot + 1 == kRlSlots ? 0 : islot + 1;
    if (++ik == i_ks) {
      ik = 0;
      ++ij;
      if (ij < nmine) cursor_tile();
    }
    return n;
  };

  // lane constants of the fragment reads: 16-byte chunk (2 s + half) of row r sits at chunk position (2 s + half) ^ ((r >> 2) & 3)
  const int fsw = (l31 >> 2) & 3;
  const int a_off0 = (rg * 32 + l31) * 64 + (((0 + half) ^ fsw) << 4), a_off1 = (rg * 32 + l31) * 64 + (((2 + half) ^ fsw) << 4);
  // row-major epilogue layout: the lane holds rows it * 8 + rrow (it = 0..3) of the wave's 32, columns 32 ct + rc .. rc + 3 of its slab
  const int rrow = lane >> 3, rc = (lane & 7) * 4;

  cursor_tile();
  int n_next;
  (void)issue();
  n_next = issue();
  int cslot = 0, skip = 0;

  auto run_tile = [&](auto nctc, int pi, int row0) __attribute__((always_inline)) {
    constexpr int NCT = decltype(nctc)::value;  // 32-column MFMA tiles per wave: N / 128
    constexpr int CW = NCT * 32;
    const RowLnProb& P = g.p[pi];
    const int N = NCT * 128;
    const int KS = P.K / kRlBK;
    const int w_base = (cg * CW + l31) * 64 + 4096;
    const int w_off0 = w_base + (((0 + half) ^ fsw) << 4), w_off1 = w_base + (((2 + half) ^ fsw) << 4);
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll 1
    for (int k = 0; k < KS; ++k) {
      if (skip > 0) --skip;  // the epilogue before this step waited for loads YOUNGER than this stage's DMA: it has landed (in-order queue)
      else rl_wait_le(n_next);
      rl_barrier();
      n_next = issue();  // into the slot every wave has finished reading (the stage of step k - 1)
      const char* sb = smem + cslot * kRlSlot;
      if constexpr ((ABL & 8) == 0) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sb + a_off0), a1 = *reinterpret_cast<const bf16x8*>(sb + a_off1);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(sb + w_off0 + ct * 2048), w1 = *reinterpret_cast<const bf16x8*>(sb + w_off1 + ct * 2048);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a1, acc[ct], 0, 0, 0);
      }
      }
      cslot = cslot + 1 == kRlSlots ? 0 : cslot + 1;
    }
    // ------------------------------------------------------------- epilogue -------------------------------------------------------------
    // the slot of the last stage is free once every wave has left the K loop: strips + statistics live there (the other two slots hold the next
    // tile's first stages, in flight)
    const int fslot = cslot == 0 ? kRlSlots - 1 : cslot - 1;
    rl_barrier();
    char* strip = smem + fslot * kRlSlot + wave * kRlStrip;
    float* red = reinterpret_cast<float*>(smem + fslot * kRlSlot + kRlRed);
    const int wrow0 = row0 + rg * 32;
    const long long rows_left = (long long)P.M - wrow0;
    const int xspan = rows_left > 0 ? (int)(rows_left * N * 4) : 0, yspan = xspan >> 1;
    __amdgpu_buffer_rsrc_t x_srd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(P.X) + (size_t)wrow0 * N * 4, 0, xspan, 0x00020000);
    __amdgpu_buffer_rsrc_t y_srd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(P.Y) + (size_t)wrow0 * N * 2, 0, yspan, 0x00020000);
    const uint32_t xo = (uint32_t)((rrow * N + cg * CW + rc) * 4);
    // gamma / beta -> LDS (waves 0 .. 2 N / 256 - 1, one 1 KiB piece each): older than every load the wave waits for below, visible to all after the
    // statistics barriers.  bias: all of the lane's vectors now (a load issued inside the pass loop would be the YOUNGEST operation when it is needed:
    // vmcnt(0), which also drains the X stores of the pass before -- read off the first build's ISA)
    {
      const int npar = N >> 8;
      if (wave < 2 * npar) {
        const bool isb = wave >= npar;
        const int pc = isb ? wave - npar : wave;
        rl_dma(reinterpret_cast<const char*>(isb ? P.beta : P.gamma) + pc * 1024, (uint32_t)(lane * 16),
               lds0 + (uint32_t)(fslot * kRlSlot + kRlPar + (isb ? kRlMaxN * 4 : 0) + pc * 1024));
      }
    }
    f32x4 bvec[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bvec[ct] = load4(P.bias + cg * CW + rc + ct * 32);
    rl_u32x4 rw[2][4];
    auto win_load = [&](int ct, int w) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if constexpr ((ABL & 2) == 0) rw[w][it] = __builtin_amdgcn_raw_buffer_load_b128(x_srd, xo + (uint32_t)((it * 8 * N + ct * 32) * 4), 0, 0);
        else rw[w][it] = rl_u32x4{0u, 0u, 0u, 0u};
      }
    };
    win_load(0, 0);
    if constexpr (NCT > 1) win_load(1, 1);
    f32x4 xv[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = acc[ct][4 * g4 + j];
        *reinterpret_cast<f32x4*>(strip + l31 * kRlStripRow + (8 * g4 + 4 * half) * 4) = t;
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 vv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) vv[it] = *reinterpret_cast<const f32x4*>(strip + (it * 8 + rrow) * kRlStripRow + rc * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the strip is rewritten by the next pass)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < 4; ++it) xv[ct][it] = (vv[it] + bvec[ct]) + __builtin_bit_cast(f32x4, rw[ct & 1][it]);
      __builtin_amdgcn_sched_barrier(0);
      if (ct + 2 < NCT) win_load(ct + 2, ct & 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((ABL & 1) == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rl_u32x4, xv[ct][it]), x_srd, xo + (uint32_t)((it * 8 * N + ct * 32) * 4), 0, 16);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // row statistics, two passes (rows it * 8 + rrow of the wave's 32; the row's other three column slabs are in the waves cg' != cg of this rg)
    float s[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float a = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) a += (xv[ct][it][0] + xv[ct][it][1]) + (xv[ct][it][2] + xv[ct][it][3]);
      s[it] = rl_sum8(a);
    }
    if ((lane & 7) == 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) red[(rg * 32 + it * 8 + rrow) * 4 + cg] = s[it];
    }
    rl_barrier();
    float mean[4], rstd[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(red + (rg * 32 + it * 8 + rrow) * 4);
      mean[it] = ((r[0] + r[1]) + (r[2] + r[3])) / (float)N;
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = xv[ct][it][j] - mean[it];
          q += u * u;
        }
      s[it] = rl_sum8(q);
    }
    if ((lane & 7) == 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) red[256 + (rg * 32 + it * 8 + rrow) * 4 + cg] = s[it];
    }
    rl_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(red + 256 + (rg * 32 + it * 8 + rrow) * 4);
      rstd[it] = 1.0f / sqrtf(((r[0] + r[1]) + (r[2] + r[3])) / (float)N + P.eps);
    }
    const float* gam = reinterpret_cast<const float*>(smem + fslot * kRlSlot + kRlPar) + cg * CW + rc;
    const float* bet = gam + kRlMaxN;
    const uint32_t yo = (uint32_t)((rrow * N + cg * CW + rc) * 2);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const f32x4 gv = load4(gam + ct * 32), bv = load4(bet + ct * 32);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)((xv[ct][it][j] - mean[it]) * rstd[it] * gv[j] + bv[j]);
        if constexpr ((ABL & 1) == 0)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rl_u32x2, o), y_srd, yo + (uint32_t)((it * 8 * N + ct * 32) * 2), 0, 0);
      }
    }
    skip = 2;
  };

#pragma unroll 1
  for (int j = 0; j < nmine; ++j) {
    const int t = (int)blockIdx.x + j * (int)gridDim.x;
    const int pi = (g.nprob > 1 && t >= tiles0) ? 1 : 0;
    const int row0 = (t - (pi ? tiles0 : 0)) * kRlBM;
    if (g.p[pi].N == 768) run_tile(std::integral_constant<int, 6>{}, pi, row0);
    else run_tile(std::integral_constant<int, 4>{}, pi, row0);
  }
}

// W [N, K] row-major -> [K / 32][N][32]: the W slice of one K step of gemm_rowln_kernel is one contiguous block (its DMA pieces read whole 128-byte
// lines; from the row-major layout every piece took 64 bytes out of each of 16 lines and the L2 -> L1 traffic doubled: 137 vs ... us of DMA per launch)
__global__ __launch_bounds__(256) void pack_w_ksteps_kernel(const bf16* __restrict__ w, bf16* __restrict__ out, int N, int K) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte chunk (8 bf16) per thread
  const int kc8 = K >> 3;
  if (idx >= (long long)N * kc8) return;
  const int n = (int)(idx / kc8), kc = (int)(idx - (long long)n * kc8);
  const int step = kc >> 2, c = kc & 3;
  *reinterpret_cast<uint4*>(out + ((size_t)step * N + n) * 32 + c * 8) = *reinterpret_cast<const uint4*>(w + (size_t)n * K + kc * 8);
}

static int g_rowln_abl = 0;  // mmamd_debug_set_gemm_knob(5, code): timing ablations of the kernel (results wrong unless 0)
void set_rowln_ablation(int code) { g_rowln_abl = code; }

}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_gemm_bf16_residual_ln_supported(int M, int N, int K) {
  return (N == 512 || N == 768) && K % kRlBK == 0 && K >= 3 * kRlBK && K <= 4096 && M > 0 && (long long)M * N * 4 <= 0x7fffffffLL &&
         (long long)M * K * 2 <= 0xffffffffLL;
}

extern "C" int mmamd_pack_w_ksteps(const void* W, int N, int K, void* out, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(W && out && N > 0 && K > 0 && K % kRlBK == 0, MMAMD_E_BADARG, "pack_w_ksteps: bad argument (K must be a multiple of 32)");
  MMAMD_CHECK_ARG(aligned16(W) && aligned16(out), MMAMD_E_ALIGN, "pack_w_ksteps: bases must be 16-byte aligned");
  const long long chunks = (long long)N * (K / 8);
  hipLaunchKernelGGL(pack_w_ksteps_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)W, (bf16*)out, N, K);
  return launch_status("pack_w_ksteps");
}

extern "C" int mmamd_gemm_bf16_residual_ln_grouped(const mmamd_gemm_ln_problem* probs, int nprob, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(probs != nullptr && nprob >= 1 && nprob <= 2, MMAMD_E_BADARG, "gemm_bf16_residual_ln_grouped: 1 or 2 problems");
  RowLnArgs a;
  a.nprob = 0;
  a.tiles_total = 0;
  for (int i = 0; i < nprob; ++i) {
    const mmamd_gemm_ln_problem& q = probs[i];
    MMAMD_CHECK_ARG(q.A && q.W && q.bias && q.X && q.gamma && q.beta && q.Y && q.M >= 0 && q.N > 0 && q.K > 0, MMAMD_E_BADARG,
                    "gemm_bf16_residual_ln_grouped: bad argument (problem %d)", i);
    if (q.M == 0) continue;
    MMAMD_CHECK_ARG(mmamd_gemm_bf16_residual_ln_supported(q.M, q.N, q.K), MMAMD_E_UNSUPPORTED,
                    "gemm_bf16_residual_ln_grouped: problem %d (M=%d N=%d K=%d): N must be 512 or 768, K a multiple of 32 in [96, 4096], M N < 2^29", i, q.M,
                    q.N, q.K);
    MMAMD_CHECK_ARG(aligned16(q.A) && aligned16(q.W) && aligned16(q.bias) && aligned16(q.X) && aligned16(q.gamma) && aligned16(q.beta) && aligned16(q.Y),
                    MMAMD_E_ALIGN, "gemm_bf16_residual_ln_grouped: bases must be 16-byte aligned");
    RowLnProb& p = a.p[a.nprob];
    p.A = (const bf16*)q.A; p.W = (const bf16*)q.W; p.bias = q.bias; p.X = q.X; p.gamma = q.gamma; p.beta = q.beta; p.Y = (bf16*)q.Y;
    p.M = q.M; p.N = q.N; p.K = q.K; p.eps = q.eps; p.tiles = (q.M + kRlBM - 1) / kRlBM; p.pad_ = 0;
    a.tiles_total += p.tiles;
    ++a.nprob;
  }
  if (a.nprob == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int cus = stream_cus(st);
  const int grid = a.tiles_total < cus ? a.tiles_total : cus;
  constexpr int smem = kRlSlots * kRlSlot;
  static unsigned long long mask[12] = {0};
#define RL_LAUNCH(ABL, SLOT)                                                                           \
  do {                                                                                                  \
    if (int e = opt_in_lds((const void*)gemm_rowln_kernel<ABL>, smem, mask[SLOT])) return e;            \
    hipLaunchKernelGGL((gemm_rowln_kernel<ABL>), dim3(grid), dim3(512), smem, st, a);                   \
  } while (0)
  switch (g_rowln_abl) {
    case 0: RL_LAUNCH(0, 0); break;
    case 1: RL_LAUNCH(1, 1); break;
    case 2: RL_LAUNCH(2, 2); break;
    case 3: RL_LAUNCH(3, 3); break;
    case 4: RL_LAUNCH(4, 4); break;
    case 8: RL_LAUNCH(8, 5); break;
    case 11: RL_LAUNCH(11, 6); break;
    case 7: RL_LAUNCH(7, 7); break;
    case 16: RL_LAUNCH(16, 8); break;
    case 27: RL_LAUNCH(27, 9); break;
    case 19: RL_LAUNCH(19, 10); break;
    default: MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "gemm_bf16_residual_ln_grouped: unknown ablation %d", g_rowln_abl);
  }
#undef RL_LAUNCH
  return launch_status("gemm_bf16_residual_ln_grouped");
}
