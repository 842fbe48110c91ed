// conv.hip — the DALL-E dVAE encoder of FLAVA's image codebook (reference torchmultimodal/models/flava/model.py:583-744) on
// the gfx950 matrix cores: 3x3 / 1x1 convolutions as an implicit GEMM over a zero-bordered NHWC grid, plus the small kernels
// around it (7x7 stem im2col, 2x2 max pool, argmax).
//
// Layout.  An activation tensor is bf16 rows [B * GH * GW, C] with GH = H + 2, GW = W + 2: every image carries a one-pixel ZERO
// border (= the convolution's padding), and the buffer has GW + 1 readable guard rows in front and behind (their contents never
// reach a stored value: only border positions, which are stored as zeros, read them).  Output pixel m of a
// 3x3 convolution is then  sum over the 9 taps (dy, dx)  A[m + dy*GW + dx, :] . W_tap^T : nine row-shifted GEMMs that share
// one accumulator, i.e. ONE GEMM whose K loop walks (tap, channel chunk) — no im2col copy.  Outputs are computed for border
// positions too and stored as zeros (the next layer's padding).  A 1x1 convolution is the same kernel with one tap.
// The kernel is the plain double-buffered LDS-DMA GEMM of gemm.hip (256-row tiles, BN = 256 / 128 / 64 for the narrow hidden
// widths of the encoder) with the tap walk in the scalar base address of the activation stream.
// Epilogue: + bias, + bf16 residual (the block's identity path; post_gain is folded into the last conv's weights by the host),
// border rows -> 0, then any of: bf16 C, bf16 relu(C) (second output: the next conv's input), fp32 C (the final logits).
#include "common.h"

namespace mmamd {

typedef uint32_t __attribute__((address_space(3))) * lds_u32p_c;

struct ConvArgs {
  const bf16* A;
  const bf16* W;
  const float* bias;
  const bf16* R;
  void* C;
  bf16* C2;
  int M, N, Cin, ntaps;
  int lda, ldw, ldr, ldc, ldc2;
  int relu_c, out_f32;
  int gh, gw;
  int tiles_n;
  long long tap_off[9];
};

__device__ __forceinline__ void conv_dma_piece(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(const ConvArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 4 == 0, "tile / wave geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int bid = blockIdx.x;
  {  // XCD-aware: block b runs on XCD b % 8; give each XCD a contiguous id range (neighbouring row panels share the weights in L2)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  // DMA source offsets (same swizzled LDS image as gemm.hip: piece i = wave + NW*j covers tile rows 8i..8i+7)
  const int sw = (4 * (wave & 3) + (lane >> 4)) & 15;
  const int slot = (lane & 15) ^ sw;
  const int row8 = 2 * (lane >> 4) + (slot >> 3);
  const int chunk = slot & 7;
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
  const int kpt = p.Cin >> 6;  // K-tiles per tap
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p_c)smem;
  // K walk: channel chunk OUTER, tap INNER.  The nine taps of one 64-channel chunk read the same ~(256 + 2 GW) activation rows shifted by
  // at most GW + 1, back to back, so all but the first find them in the XCD's L2; with the taps outermost every tap re-streamed the whole
  // [256 rows x Cin] panel after megabytes of other traffic (PMC on the first group-1 conv: FETCH_SIZE 3.6 GB for an 852 MB activation
  // tensor, TCC hit rate 27 %).  The weight's K order stays (tap, channel): only the tile walk changes.
  auto issue_stage = [&](int buf, int kt) {
    const int kin = kt / p.ntaps, tap = kt - kin * p.ntaps;  // scalar: once per K-tile
    const char* abase = Ab + p.tap_off[tap] + (size_t)kin * 128;
    const char* wbase = Wb + ((size_t)tap * kpt + kin) * 128;
    const uint32_t dst = lds0 + buf * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) conv_dma_piece(abase, a_off[j], dst + NW * j * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) conv_dma_piece(wbase, b_off[j], dst + A_BYTES + NW * j * 1024);
  };

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

  const int KT = kpt * p.ntaps;
  issue_stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) issue_stage((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE + (wm * TM) * 128;
    const char* sb = smem + (kt & 1) * STAGE + A_BYTES + (wn * TN) * 128;
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
        for (int ni = 0; ni < NI; ++ni) wb[nxt][ni] = *reinterpret_cast<const bf16x8*>(sb + ni * 32 * 128 + roff[t + 1]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xa[nxt][mi] = *reinterpret_cast<const bf16x8*>(sa + mi * 32 * 128 + roff[t + 1]);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[cur][ni], xa[cur][mi], acc[ni][mi], 0, 0, 0);
    }
    {  // keep the fragment reads of k-step t+1 interleaved with the MFMAs of k-step t (hipcc otherwise sinks them)
      constexpr int NF = NI + MI, NM = NI * MI;
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        constexpr int PAIRS = NM < NF ? NM : NF;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NF > NM) __builtin_amdgcn_sched_group_barrier(0x100, NF - NM, 0);
        if constexpr (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    }
  }

  // ---- epilogue: lane owns row m (one grid position); accumulator regs 4g..4g+3 = columns n + 8g + 4*half + {0..3}
  const int img = p.gh * p.gw;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * TM + mi * 32 + l31;
    const bool mok = m < p.M;
    bool border = false;
    if (img > 0) {
      const int rr = m % img;
      const int y = rr / p.gw, x = rr - y * p.gw;
      border = y == 0 || y == p.gh - 1 || x == 0 || x == p.gw - 1;
    }
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
        if (p.bias != nullptr && n + 3 < p.N) {
          const f32x4 bv = load4(p.bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] += bv[j];
        }
        if (p.R != nullptr && ok) {
          const f32x4 rv = load4(p.R + (size_t)m * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] += rv[j];
        }
        if (border) t = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.out_f32) {
          if (ok) store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, t);
        }
        v[g] = t;
      }
      if (!p.out_f32) {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          bf16x4 pa, pb, ra, rb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = v[g][j], b = v[g + 1][j];
            pa[j] = (bf16)(p.relu_c ? fmaxf(a, 0.f) : a);
            pb[j] = (bf16)(p.relu_c ? fmaxf(b, 0.f) : b);
            ra[j] = (bf16)fmaxf((float)(bf16)a, 0.f);  // relu of the value as the first output stores it
            rb[j] = (bf16)fmaxf((float)(bf16)b, 0.f);
          }
          const int n = n0 + wn * TN + ni * 32 + 8 * (g + half);
          {
            const uint2 ua = __builtin_bit_cast(uint2, pa), ub = __builtin_bit_cast(uint2, pb);
            auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);  // lanes 32-63 of a <-> lanes 0-31 of b
            auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
            if (mok && n + 7 < p.N)
              *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
          if (p.C2 != nullptr) {
            const uint2 ua = __builtin_bit_cast(uint2, ra), ub = __builtin_bit_cast(uint2, rb);
            auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
            if (mok && n + 7 < p.N) *reinterpret_cast<uint4*>(p.C2 + (size_t)m * p.ldc2 + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// 7x7 stem (3 -> n_hid channels, padding 3): im2col of the fp32 NCHW image into bf16 rows of the padded grid.  Row m = grid
// position (b, y, x) incl. the border; columns (c, ky, kx) in the weight's own [n_in][kw][kw] order, zero-padded to kpad.  Border
// rows are written too (their GEMM output is discarded by the border mask).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dalle_stem_im2col_kernel(const float* __restrict__ img, bf16* __restrict__ cols, int B, int C, int H, int W,
                                                                int kw, int kpad, long long rows) {
  const int gh = H + 2, gw = W + 2;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // wave per grid position
  if (row >= rows) return;
  const int b = (int)(row / (gh * gw));
  const int rr = (int)(row - (long long)b * gh * gw);
  const int y = rr / gw - 1, x = rr % gw - 1;  // image coordinates of this grid position
  const int pad = (kw - 1) / 2, kk = kw * kw;
  for (int k = lane; k < kpad; k += 64) {
    float v = 0.f;
    if (k < C * kk) {
      const int c = k / kk, r2 = k - c * kk;
      const int ky = r2 / kw, kx = r2 - ky * kw;
      const int iy = y + ky - pad, ix = x + kx - pad;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(((size_t)b * C + c) * H + iy) * W + ix];
    }
    cols[row * kpad + k] = (bf16)v;
  }
}

// 2x2 max pool of a padded-grid tensor [B, H+2, W+2, C] -> [B, H/2+2, W/2+2, C] (zero border), plus the ReLU copy of the result
// (relu(maxpool(x)) = the next block's first conv input; either output may be NULL)
__global__ __launch_bounds__(256) void dalle_maxpool_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, bf16* __restrict__ yr, int B, int H,
                                                            int W, int C) {
  const int gh = H + 2, gw = W + 2, oh = H / 2 + 2, ow = W / 2 + 2;
  const long long row = blockIdx.x;  // output grid position
  const int b = (int)(row / (oh * ow));
  const int rr = (int)(row - (long long)b * oh * ow);
  const int oy = rr / ow, ox = rr % ow;
  const bool border = oy == 0 || oy == oh - 1 || ox == 0 || ox == ow - 1;
  const int iy = 2 * (oy - 1) + 1, ix = 2 * (ox - 1) + 1;  // top-left input grid position of the window
  for (int c = threadIdx.x * 8; c < C; c += 256 * 8) {
    bf16x8 o, orl;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = (bf16)0.f; orl[j] = (bf16)0.f; }
    if (!border) {
      const bf16* p00 = x + (((size_t)b * gh + iy) * gw + ix) * C + c;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(p00), bq = *reinterpret_cast<const bf16x8*>(p00 + C);
      const bf16x8 cq = *reinterpret_cast<const bf16x8*>(p00 + (size_t)gw * C), d = *reinterpret_cast<const bf16x8*>(p00 + (size_t)gw * C + C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float m = fmaxf(fmaxf((float)a[j], (float)bq[j]), fmaxf((float)cq[j], (float)d[j]));
        o[j] = (bf16)m;
        orl[j] = (bf16)fmaxf(m, 0.f);
      }
    }
    if (y != nullptr) *reinterpret_cast<bf16x8*>(y + row * C + c) = o;
    if (yr != nullptr) *reinterpret_cast<bf16x8*>(yr + row * C + c) = orl;
  }
}

// ids[b, y, x] = argmax over the V channels of the fp32 logits at interior grid position (b, y+1, x+1)  (first maximum wins, like
// torch.argmax on the reference's [B, V, H, W] logits: models/flava/model.py:733-735)
__global__ __launch_bounds__(256) void dalle_argmax_kernel(const float* __restrict__ logits, long long* __restrict__ ids, int B, int H, int W, int V) {
  const int gw = W + 2, gh = H + 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pos = (long long)blockIdx.x * 4 + wave;
  if (pos >= (long long)B * H * W) return;
  const int b = (int)(pos / (H * W));
  const int r = (int)(pos - (long long)b * H * W);
  const int y = r / W, x = r - y * W;
  const float* row = logits + (((size_t)b * gh + y + 1) * gw + x + 1) * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = row[c];
    if (v > best) { best = v; bi = c; }  // per lane the indices ascend: the first maximum is kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) ids[pos] = bi;
}

// kernel-ready copy of a DalleConv2d parameter: dst[o*ld + idx] = gain * src[o][c][t], idx = t*n_in + c (tap-major, the implicit
// GEMM's K order) or c*taps + t (the parameter's own order: the 7x7 stem, matching dalle_stem_im2col); idx >= n_in*taps -> 0
template <typename TD>
__global__ __launch_bounds__(256) void dalle_pack_kernel(const float* __restrict__ src, TD* __restrict__ dst, int n_out, int n_in, int taps, int ld,
                                                         float gain, int tap_major) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_out * ld) return;
  const int o = (int)(i / ld), idx = (int)(i - (long long)o * ld);
  float v = 0.f;
  if (idx < n_in * taps) {
    const int c = tap_major ? idx % n_in : idx / taps, t = tap_major ? idx / n_in : idx % taps;
    v = gain * src[((size_t)o * n_in + c) * taps + t];
  }
  dst[i] = (TD)v;
}

// in-place softmax over the V channels of every grid row (DalleVAEEncoder.get_codebook_probs, models/flava/model.py:737-739:
// nn.Softmax(dim=1) on the NCHW logits = a row softmax on the NHWC rows).  Wave per row, three passes over L2-resident data.
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ x, long long rows, int V) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* r = x + row * V;
  float m = -INFINITY;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, r[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += __expf(r[c] - m);
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int c = lane; c < V; c += 64) r[c] = __expf(r[c] - m) * inv;
}

template <int BM, int BN, int WM, int WN>
static int launch_conv(ConvArgs& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_kernel<BM, BN, WM, WN>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, st, p);
  return launch_status("conv_gemm");
}

}  // namespace mmamd

using namespace mmamd;

extern "C" int mmamd_conv_gemm_bf16(const void* A, int lda, const int64_t* tap_row_offsets, int ntaps, const void* W, int ldw,
                                    const float* bias, const void* residual, int ldr, void* C, int ldc, int out_dtype, void* C_relu,
                                    int ldc_relu, int relu_c, int M, int N, int Cin, int grid_h, int grid_w, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(A && W && C && tap_row_offsets, MMAMD_E_BADARG, "conv_gemm: null pointer");
  MMAMD_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && ntaps >= 1 && ntaps <= 9, MMAMD_E_BADARG, "conv_gemm: bad sizes M=%d N=%d Cin=%d taps=%d", M, N, Cin, ntaps);
  MMAMD_CHECK_ARG(Cin % 64 == 0 && N % 8 == 0, MMAMD_E_UNSUPPORTED, "conv_gemm: Cin=%d must be a multiple of 64 and N=%d of 8", Cin, N);
  MMAMD_CHECK_ARG(lda >= Cin && lda % 8 == 0 && ldw >= ntaps * Cin && ldw % 8 == 0 && ldc >= N && ldc % 8 == 0 && (!residual || (ldr >= N && ldr % 8 == 0)) &&
                      (!C_relu || (ldc_relu >= N && ldc_relu % 8 == 0)),
                  MMAMD_E_BADARG, "conv_gemm: bad leading dimension");
  MMAMD_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(residual) && aligned16(C_relu) && aligned16(bias), MMAMD_E_ALIGN,
                  "conv_gemm: base pointers must be 16-byte aligned");
  MMAMD_CHECK_ARG(out_dtype == MMAMD_BF16 || (out_dtype == MMAMD_F32 && !C_relu && !relu_c), MMAMD_E_BADARG, "conv_gemm: fp32 output has no ReLU forms");
  MMAMD_CHECK_ARG((uint64_t)M * (uint64_t)lda * 2u < (1ull << 32) && (uint64_t)N * (uint64_t)ldw * 2u < (1ull << 32), MMAMD_E_UNSUPPORTED,
                  "conv_gemm: operand exceeds the 4 GiB 32-bit DMA offset range (split the batch)");
  MMAMD_CHECK_ARG((grid_h == 0 && grid_w == 0) || (grid_h >= 3 && grid_w >= 3 && M % (grid_h * grid_w) == 0), MMAMD_E_BADARG, "conv_gemm: bad grid %dx%d", grid_h, grid_w);
  ConvArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.R = (const bf16*)residual; p.C = C; p.C2 = (bf16*)C_relu;
  p.M = M; p.N = N; p.Cin = Cin; p.ntaps = ntaps; p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc; p.ldc2 = ldc_relu;
  p.relu_c = relu_c; p.out_f32 = out_dtype == MMAMD_F32; p.gh = grid_h; p.gw = grid_w; p.tiles_n = 0;
  for (int t = 0; t < 9; ++t) p.tap_off[t] = t < ntaps ? (long long)tap_row_offsets[t] * lda * 2 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 64) return launch_conv<256, 64, 4, 1>(p, st);
  if (N <= 128) return launch_conv<256, 128, 4, 2>(p, st);
  return launch_conv<256, 256, 2, 4>(p, st);
}

extern "C" int mmamd_dalle_stem_im2col(const float* images, void* cols, int B, int C, int H, int W, int kw, int kpad, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(images && cols && B > 0 && C > 0 && H > 0 && W > 0 && kw >= 1 && (kw & 1) && kpad >= C * kw * kw && kpad % 64 == 0, MMAMD_E_BADARG,
                  "dalle_stem_im2col: bad argument");
  const long long rows = (long long)B * (H + 2) * (W + 2);
  MMAMD_CHECK_ARG(rows < (1ll << 31), MMAMD_E_UNSUPPORTED, "dalle_stem_im2col: too many rows");
  hipLaunchKernelGGL(dalle_stem_im2col_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, images, (bf16*)cols, B, C, H, W, kw, kpad, rows);
  return launch_status("dalle_stem_im2col");
}

extern "C" int mmamd_dalle_maxpool2(const void* x, void* y, void* y_relu, int B, int H, int W, int C, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && (y || y_relu) && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, MMAMD_E_BADARG, "dalle_maxpool2: bad argument");
  const long long rows = (long long)B * (H / 2 + 2) * (W / 2 + 2);
  hipLaunchKernelGGL(dalle_maxpool_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, (bf16*)y_relu, B, H, W, C);
  return launch_status("dalle_maxpool2");
}

extern "C" int mmamd_dalle_argmax(const float* logits, int64_t* ids, int B, int H, int W, int V, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(logits && ids && B > 0 && H > 0 && W > 0 && V > 0, MMAMD_E_BADARG, "dalle_argmax: bad argument");
  const long long n = (long long)B * H * W;
  hipLaunchKernelGGL(dalle_argmax_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, (long long*)ids, B, H, W, V);
  return launch_status("dalle_argmax");
}

extern "C" int mmamd_dalle_pack(const float* src, void* dst, int dst_dtype, int n_out, int n_in, int taps, int ld_dst, float gain, int tap_major,
                                mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(src && dst && n_out > 0 && n_in > 0 && taps > 0 && ld_dst >= n_in * taps, MMAMD_E_BADARG, "dalle_pack: bad argument");
  const long long n = (long long)n_out * ld_dst;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dst_dtype == MMAMD_BF16) hipLaunchKernelGGL((dalle_pack_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n_out, n_in, taps, ld_dst, gain, tap_major);
  else if (dst_dtype == MMAMD_F32) hipLaunchKernelGGL((dalle_pack_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, n_out, n_in, taps, ld_dst, gain, tap_major);
  else MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "dalle_pack: bad dtype");
  return launch_status("dalle_pack");
}

extern "C" int mmamd_row_softmax_(float* x, int64_t rows, int V, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(x && rows >= 0 && V > 0, MMAMD_E_BADARG, "row_softmax: bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, V);
  return launch_status("row_softmax");
}
