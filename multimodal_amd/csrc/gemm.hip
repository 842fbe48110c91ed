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
#include "common.h"

namespace mmamd {

typedef uint32_t __attribute__((address_space(3))) * lds_u32p;
typedef const uint32_t __attribute__((address_space(1))) * glb_u32p;

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
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == MMAMD_ACT_QUICKGELU) return v / (1.0f + __expf(-1.702f * v));
  if (act == MMAMD_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  return v;
}

static int g_gemm_variant = 0;

// BM x BN block tile, WM x WN waves, BK = 64
template <int BM, int BN, int WM, int WN, bool OUT_F32, int ACT, bool SGB>
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

  auto issue_stage = [&](int buf, int kt) {
    char* sbase = smem + buf * STAGE;
    const uint32_t kbytes = (uint32_t)kt * 128u;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((glb_u32p)(Ab + a_off[j] + kbytes),
                                       (lds_u32p)(sbase + (wave + NW * j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((glb_u32p)(Wb + b_off[j] + kbytes),
                                       (lds_u32p)(sbase + A_BYTES + (wave + NW * j) * 1024), 16, 0, 0);
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

  // ---- epilogue: lane owns row m = .. + l31; acc regs 4g..4g+3 are columns n = .. + 8g + 4*half + {0..3}
  // pass 1: bias (depends on n only).  pass 2: activation behind ONE uniform branch.  pass 3: residual + store.
  if (p.bias != nullptr) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * TN + ni * 32 + 4 * half + 8 * g;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (n + 3 < p.N) bv = load4(p.bias + n);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ni][mi][4 * g + j] += bv[j];
      }
  }
  if constexpr (ACT == MMAMD_ACT_QUICKGELU) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[ni][mi][r];
          acc[ni][mi][r] = v / (1.0f + __expf(-1.702f * v));
        }
  } else if constexpr (ACT == MMAMD_ACT_GELU_ERF) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[ni][mi][r];
          acc[ni][mi][r] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        }
  }
  const bool has_res = p.R != nullptr;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * TM + mi * 32 + l31;
    const bool mok = m < p.M;
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
          for (int j = 0; j < 4; ++j) t[j] += rv[j];
        }
        if constexpr (OUT_F32) {
          if (ok) store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, t);
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
          if (mok && n + 7 < p.N)
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = o;
        }
      }
    }
  }
}

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
  }
}

template <int BM, int BN, int WM, int WN, bool OUT_F32, int ACT, bool SGB>
static int launch_tiled(GemmArgs& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, OUT_F32, ACT, SGB>;
  static bool attr_done = false;  // one-time opt-in to >64 KiB dynamic LDS (160 KiB per CU on gfx950)
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { set_error("gemm: hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, st, p);
  return launch_status("gemm_bf16");
}

template <bool OUT_F32, int ACT>
static int dispatch_variant(GemmArgs& p, hipStream_t st) {
  int v = g_gemm_variant;
  if (v == 0) {
    // default policy: big tile when the grid still fills the chip, else the 128x128 tile
    const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    v = (t256 >= 256) ? 1 : 2;
  }
  switch (v) {
    case 1: return launch_tiled<256, 256, 2, 4, OUT_F32, ACT, false>(p, st);
    case 2: return launch_tiled<128, 128, 2, 2, OUT_F32, ACT, false>(p, st);
    case 3: return launch_tiled<256, 128, 4, 2, OUT_F32, ACT, false>(p, st);
    case 4: return launch_tiled<128, 256, 2, 4, OUT_F32, ACT, false>(p, st);
    case 5: return launch_tiled<256, 256, 2, 4, OUT_F32, ACT, true>(p, st);
    case 6: return launch_tiled<128, 128, 2, 2, OUT_F32, ACT, true>(p, st);
    default: set_error("gemm: unknown variant %d", v); return MMAMD_E_BADARG;
  }
}

template <bool OUT_F32>
static int dispatch(GemmArgs& p, hipStream_t st) {
  if (g_gemm_variant == 99) {
    hipLaunchKernelGGL((gemm_naive_kernel<OUT_F32>), dim3((p.N + 63) / 64, (p.M + 3) / 4), dim3(256), 0, st, p);
    return launch_status("gemm_naive");
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

extern "C" int mmamd_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual,
                               int ldr, void* C, int ldc, int out_dtype, int M, int N, int K, int act,
                               mmamd_stream_t stream) {
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
  MMAMD_CHECK_ARG(act >= MMAMD_ACT_NONE && act <= MMAMD_ACT_GELU_ERF, MMAMD_E_BADARG, "gemm: bad activation code %d", act);
  if (M == 0) return 0;
  GemmArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.R = residual; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc; p.act = act; p.tiles_n = 0;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == MMAMD_F32) return dispatch<true>(p, st);
  if (out_dtype == MMAMD_BF16) return dispatch<false>(p, st);
  MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "gemm: bad out_dtype %d", out_dtype);
}
