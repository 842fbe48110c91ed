// loss.hip — contrastive logits + cross entropy in fp32 (gfx950).
//   logits kernel : one wave per 32x32 logit tile on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32, a
//                   k-ordered fmaf chain, 1/16 of the bf16 MFMA rate) — the loss is 0.001 % of the
//                   step's FLOPs, so it is kept in fp32 to leave the argmax/ordering of the logits
//                   as close to the reference's fp32 matmul as the features allow.
//   ce_rows kernel: one wave per logit row: max / log-sum-exp / label pick / smoothing term.
//   ce_reduce     : one block: masked mean (or sum) over rows, both directions, final average.
// Reference: modules/losses/contrastive_loss_with_temperature.py:81,90-107.
#include "common.h"

namespace mmamd {

// dir 0: logits_a = a . b_all^T ; dir 1: logits_b = b . a_all^T
__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ a_all, const float* __restrict__ b_all,
                                                     int ld_all, const float* __restrict__ logit_scale, int B, int WB,
                                                     int E, float* __restrict__ logits_a, float* __restrict__ logits_b, int ld_loc) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int dir = blockIdx.z;
  const float* L = dir == 0 ? a : b;          // [B,E], row stride ld_loc (E, or 2E when a / b are halves of the packed block)
  const float* R = dir == 0 ? b_all : a_all;  // [WB,E], row stride ld_all
  float* out = dir == 0 ? logits_a : logits_b;
  const int i0 = blockIdx.y * 32;
  const int j0 = (blockIdx.x * 4 + wv) * 32;
  if (j0 >= WB) return;  // wave-uniform
  const int half = lane >> 5;
  int ri = i0 + (lane & 31); ri = ri < B ? ri : B - 1;
  int rj = j0 + (lane & 31); rj = rj < WB ? rj : WB - 1;
  const float* lp = L + (size_t)ri * ld_loc;
  const float* rp = R + (size_t)rj * ld_all;
  const bool vec = ((E & 3) == 0) && ((ld_all & 3) == 0) && ((ld_loc & 3) == 0) && ((reinterpret_cast<uintptr_t>(L) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(R) & 15) == 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int e0 = 0; e0 < E; e0 += 8) {
    const int e = e0 + 4 * half;
    f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
    if (vec && e + 3 < E) {
      x = load4(lp + e);
      y = load4(rp + e);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e + u < E) { x[u] = lp[e + u]; y[u] = rp[e + u]; }
    }
    // k-slot (half) of MFMA #u holds feature index e0 + 4*half + u for BOTH operands
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u], acc, 0, 0, 0);
  }
  const float T = expf(*logit_scale);
  const int j = j0 + (lane & 31);
  if (j < WB) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < B) out[(size_t)i * WB + j] = acc[r] * T;
    }
  }
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits_a,
                                                      const float* __restrict__ logits_b, int B, int WB,
                                                      int label_offset, const uint8_t* __restrict__ row_mask,
                                                      float smoothing, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= 2 * B) return;
  const int dir = gw / B, row = gw - dir * B;
  if (row_mask != nullptr && row_mask[row] == 0) {
    if (lane == 0) ws[gw] = 0.f;
    return;
  }
  const float* lr = (dir == 0 ? logits_a : logits_b) + (size_t)row * WB;
  float m = -INFINITY;
  for (int j = lane; j < WB; j += 64) m = fmaxf(m, lr[j]);
  m = wave_max(m);
  float se = 0.f, sl = 0.f;
  for (int j = lane; j < WB; j += 64) {
    const float v = lr[j];
    se += expf(v - m);
    sl += v;
  }
  se = wave_sum(se);
  sl = wave_sum(sl);
  if (lane == 0) {
    const float lse = m + logf(se);
    const float nll = lse - lr[label_offset + row];
    const float smooth = lse - sl / (float)WB;
    ws[gw] = (1.f - smoothing) * nll + smoothing * smooth;
  }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ ws, int B,
                                                        const uint8_t* __restrict__ row_mask, int reduction,
                                                        float* __restrict__ out3) {
  __shared__ float red[3][4];
  float sa = 0.f, sb = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const bool valid = row_mask == nullptr || row_mask[i] != 0;
    if (valid) { sa += ws[i]; sb += ws[B + i]; cnt += 1.f; }
  }
  sa = wave_sum(sa); sb = wave_sum(sb); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; red[2][threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float la = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    float lb = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float n = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    if (reduction == MMAMD_REDUCE_MEAN) { la /= n; lb /= n; }  // n == 0 -> nan, as F.cross_entropy
    out3[0] = (la + lb) * 0.5f;
    out3[1] = la;
    out3[2] = lb;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Backward of the contrastive loss (first slice): gradient of
//   loss = 0.5 (CE(logits_a, y) + CE(logits_b, y)),  logits_a = T a b_all^T,  logits_b = T b a_all^T,  T = exp(s)
// (modules/losses/contrastive_loss_with_temperature.py:81-107) with respect to a, b, the gathered a_all / b_all and s.
// ---------------------------------------------------------------------------------------------------------
// G[dir][row, j] = w_dir * (softmax(logits)[j] - (1 - eps) [j == label] - eps / WB) for kept rows, 0 for masked rows;
// w_a = (0.5 g[0] + g[1]) / n, w_b = (0.5 g[0] + g[2]) / n (mean; n = kept rows) where g = d(out3) from upstream.
// ws[gw] = sum_j G * logits (this row's share of d logit_scale).  One wave per (dir, row).
__global__ __launch_bounds__(256) void ce_grad_rows_kernel(const float* __restrict__ logits_a, const float* __restrict__ logits_b,
                                                           int B, int WB, int label_offset, const uint8_t* __restrict__ row_mask,
                                                           float smoothing, int reduction, const float* __restrict__ gout3,
                                                           float* __restrict__ G_a, float* __restrict__ G_b,
                                                           float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= 2 * B) return;
  const int dir = gw / B, row = gw - dir * B;
  float* g = (dir == 0 ? G_a : G_b) + (size_t)row * WB;
  if (row_mask != nullptr && row_mask[row] == 0) {
    for (int j = lane; j < WB; j += 64) g[j] = 0.f;
    if (lane == 0) ws[gw] = 0.f;
    return;
  }
  float n = (float)B;
  if (row_mask != nullptr) {
    float c = 0.f;
    for (int i = lane; i < B; i += 64) c += row_mask[i] != 0 ? 1.f : 0.f;
    n = wave_sum(c);
  }
  const float up = 0.5f * gout3[0] + gout3[1 + dir];
  const float w = reduction == MMAMD_REDUCE_MEAN ? up / n : up;
  const float* lr = (dir == 0 ? logits_a : logits_b) + (size_t)row * WB;
  float m = -INFINITY;
  for (int j = lane; j < WB; j += 64) m = fmaxf(m, lr[j]);
  m = wave_max(m);
  float se = 0.f;
  for (int j = lane; j < WB; j += 64) se += expf(lr[j] - m);
  se = wave_sum(se);
  const float inv = 1.0f / se;
  const int label = label_offset + row;
  const float unif = smoothing / (float)WB;
  float ds = 0.f;
  for (int j = lane; j < WB; j += 64) {
    const float v = lr[j];
    const float gj = w * (expf(v - m) * inv - (j == label ? 1.f - smoothing : 0.f) - unif);
    g[j] = gj;
    ds += gj * v;
  }
  ds = wave_sum(ds);
  if (lane == 0) ws[gw] = ds;
}

__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ ws, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += ws[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// C[m, n] = exp(*log_alpha) * sum_k X(m, k) Y(n, k) (+ R[m, n]) on the exact-f32 MFMA, operands addressed by element strides:
// X(m, k) = X[m*sxm + k*sxk], Y(n, k) = Y[n*syn + k*syk] — so that G.b_all, G^T.a etc. need no transposed copies.
__global__ __launch_bounds__(256) void f32_gemm_strided_kernel(const float* __restrict__ X, long long sxm, long long sxk,
                                                               const float* __restrict__ Y, long long syn, long long syk,
                                                               const float* __restrict__ log_alpha, const float* __restrict__ R,
                                                               int ldr, float* __restrict__ C, int ldc, int M, int N, int K) {
  // One 32 x 32 tile of C per workgroup, its contraction split over the four waves (r05): these are the 0.07-0.2 GFLOP gradients of the projections and
  // of the loss, ten of them at the head of a CLIP training step's backward with nothing to overlap -- a wave per tile walking all of K was a chain of
  // K / 32 dependent memory round trips (58 us for K = 256 .. 768).  Wave w takes k in [w Kq, (w + 1) Kq); the four partial tiles meet in LDS and are added
  // in wave order (deterministic).
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m0 = blockIdx.y * 32;
  const int n0 = blockIdx.x * 32;
  const int half = lane >> 5;
  int rm = m0 + (lane & 31); rm = rm < M ? rm : M - 1;
  int rn = n0 + (lane & 31); rn = rn < N ? rn : N - 1;
  const float* xp = X + (size_t)rm * sxm;
  const float* yp = Y + (size_t)rn * syn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int Kq = (((K + 3) / 4) + 7) & ~7;  // per-wave share, a multiple of the 8 contraction steps one load group covers
  const int kb = wv * Kq, ke = kb + Kq < K ? kb + Kq : K;
  // 32 contraction steps per trip, all 32 loads of the trip issued before the first MFMA
  for (int k0 = kb; k0 < ke; k0 += 32) {
    float xs[16], ys[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 8 * i + 4 * half + u;
        xs[4 * i + u] = k < ke ? xp[(size_t)k * sxk] : 0.f;
        ys[4 * i + u] = k < ke ? yp[(size_t)k * syk] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[i], ys[i], acc, 0, 0, 0);
  }
  if (wv > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wv != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
  const float alpha = log_alpha ? expf(*log_alpha) : 1.f;
  const int n = n0 + (lane & 31);
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < M) {
        float v = acc[r] * alpha;
        if (R != nullptr) v += R[(size_t)m * ldr + n];
        C[(size_t)m * ldc + n] = v;
      }
    }
  }
}


// backward of mmamd_cross_entropy: dlogits[row, j] = g / n_kept * (softmax(logits[row])[j] - [j == label]) for kept rows, 0 for
// ignored rows and for the padding columns [V, ldd).  g = *gout (upstream gradient of the mean loss), n_kept = *cnt.
template <typename TD>
__global__ __launch_bounds__(256) void ce_generic_bwd_kernel(const float* __restrict__ logits, size_t ld, const int64_t* __restrict__ labels,
                                                             int N, int V, long long ignore, const float* __restrict__ gout,
                                                             const float* __restrict__ cnt, TD* __restrict__ dlogits, size_t ldd) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TD* out = dlogits + (size_t)row * ldd;
  const long long lab = labels[row];
  if (lab == ignore || lab < 0 || lab >= V) {
    for (int j = tid; j < (int)ldd; j += 256) out[j] = (TD)0.f;
    return;
  }
  const float* x = logits + (size_t)row * ld;
  // pass 1 (r05: ONE read for max and sum -- running (max, sum relative to it) per thread over 16-byte chunks, merged at the end; it was two reads),
  // pass 2: the gradient, 4 columns per thread (8- / 16-byte stores; it was one 2- / 4-byte store per thread)
  const bool vec = (V & 3) == 0 && (ld & 3) == 0 && (ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0;
  float m = -INFINITY, se = 0.f;
  if (vec) {
    for (int j = tid * 4; j < V; j += 1024) {
      const f32x4 v = load4(x + j);
      const float mn = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
      if (mn != -INFINITY) {  // (a thread whose elements so far are all -inf keeps (-inf, 0): m - mn would be NaN)
        se = se * __expf(m - mn) + ((__expf(v[0] - mn) + __expf(v[1] - mn)) + (__expf(v[2] - mn) + __expf(v[3] - mn)));
        m = mn;
      }
    }
  } else {
    for (int j = tid; j < V; j += 256) {
      const float v = x[j], mn = fmaxf(m, v);
      if (mn != -INFINITY) {  // masked (-inf) logits before the thread's first finite one: keep (-inf, 0), do not form -inf - -inf
        se = se * __expf(m - mn) + __expf(v - mn);
        m = mn;
      }
    }
  }
  const float wm = wave_max(m);
  se = wave_sum(m == -INFINITY ? 0.f : se * __expf(m - wm));
  __shared__ float redm[4];
  if (lane == 0) { red[wave] = se; redm[wave] = wm; }
  __syncthreads();
  const float M = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  float tot = 0.f;
  for (int w4 = 0; w4 < 4; ++w4) tot += redm[w4] == -INFINITY ? 0.f : red[w4] * __expf(redm[w4] - M);
  const float inv = 1.0f / tot;
  const float w = gout[0] / cnt[0];
  if (vec) {
    for (int j = tid * 4; j < (int)ldd; j += 1024) {
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (j < V) {
        const f32x4 v = load4(x + j);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = w * (__expf(v[i] - M) * inv - (j + i == lab ? 1.f : 0.f));
      }
      store4(out + j, g);
    }
  } else {
    for (int j = tid; j < (int)ldd; j += 256) {
      float g = 0.f;
      if (j < V) g = w * (__expf(x[j] - M) * inv - (j == lab ? 1.f : 0.f));
      out[j] = (TD)g;
    }
  }
}

}  // namespace mmamd

using namespace mmamd;


// ---------------------------------------------------------------------------------------------------------
// FLAVA masked-prediction / ITM heads (modules/losses/flava.py:110-238)
// ---------------------------------------------------------------------------------------------------------
// Order-preserving compaction of the labelled positions: for labels [B, L] (and an optional per-sample keep flag) emit, for
// every kept (b, l) in row-major order, the source row b*seq_S + tok_offset + l and its label.  This is what the reference
// does with boolean indexing (`hidden_states[masked_tokens, :]`, `labels[masked_tokens]`, `sequence[pos_mask]`): one block,
// ballot + popcount prefix per 1024-element chunk (B*L is at most a few 10^4).
__global__ __launch_bounds__(1024) void select_tokens_kernel(const int64_t* __restrict__ labels, const uint8_t* __restrict__ row_keep,
                                                             long long ignore, int B, int L, int seq_S, int tok_offset,
                                                             int* __restrict__ idx_out, int64_t* __restrict__ label_out,
                                                             int* __restrict__ count_out) {
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = B * L;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int start = 0; start < total; start += 1024) {
    const int i = start + tid;
    bool flag = false;
    int b = 0, l = 0;
    long long lab = 0;
    if (i < total) {
      b = i / L; l = i - b * L;
      lab = labels[i];
      flag = (row_keep == nullptr || row_keep[b] != 0) && lab != ignore;
    }
    const unsigned long long ballot = __ballot(flag);
    const int lanepos = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(ballot);
    __syncthreads();
    int prefix = 0, chunk = 0;
    for (int w = 0; w < 16; ++w) { const int c = wave_cnt[w]; if (w < wave) prefix += c; chunk += c; }
    if (flag) {
      const int pos = base + prefix + lanepos;
      idx_out[pos] = b * seq_S + tok_offset + l;
      if (label_out) label_out[pos] = lab;
    }
    __syncthreads();
    if (tid == 0) base += chunk;
    __syncthreads();
  }
  if (tid == 0) count_out[0] = base;
}

// dst[i, :] = src[idx[i], :] (fp32 rows `row_stride` floats apart) as fp32 or bf16 — wave per row
template <typename TO>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, size_t row_stride, const int* __restrict__ idx,
                                                          int n, int d, TO* __restrict__ dst, const int64_t* __restrict__ zero_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* s = src + (size_t)idx[row] * row_stride;
  const bool zero = zero_rows != nullptr && zero_rows[row] != 0;  // e.g. masked patches: their embedding gets no gradient
  for (int c = lane; c < (d >> 2); c += 64) store4(dst + (size_t)row * d + 4 * c, zero ? f32x4{0.f, 0.f, 0.f, 0.f} : load4(s + 4 * c));
}

// nn.CrossEntropyLoss(ignore_index) rows: ws[row] = lse - logit[label] (0 for ignored rows), ws[N + row] = 1 / 0 kept flag
__global__ __launch_bounds__(256) void ce_generic_rows_kernel(const float* __restrict__ logits, size_t ld, const int64_t* __restrict__ labels,
                                                              int N, int V, long long ignore, float* __restrict__ ws) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long lab = labels[row];
  if (lab == ignore || lab < 0 || lab >= V) {  // out-of-range labels are an error in torch; they are dropped here
    if (tid == 0) { ws[row] = 0.f; ws[N + row] = 0.f; }
    return;
  }
  const float* x = logits + (size_t)row * ld;
  // ONE pass over the row (r05; it was two: 1.9 GB of CoCa's [9728, 49408] caption logits read twice, 768 us): every thread keeps a running
  // (max, sum of exp relative to it) over its 16-byte chunks, the 256 pairs are merged at the end.  The result differs from the two-pass form
  // by the rounding of the rescales (~1e-7 relative).
  float m = -INFINITY, se = 0.f;
  if ((V & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
    for (int j = tid * 4; j < V; j += 1024) {
      const f32x4 v = load4(x + j);
      const float cm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
      const float mn = fmaxf(m, cm);
      if (mn != -INFINITY) {  // (first finite chunk: 0 * exp(-inf) = 0; all -inf so far: keep (-inf, 0) -- m - mn would be NaN)
        se = se * __expf(m - mn) + ((__expf(v[0] - mn) + __expf(v[1] - mn)) + (__expf(v[2] - mn) + __expf(v[3] - mn)));
        m = mn;
      }
    }
  } else {
    for (int j = tid; j < V; j += 256) {
      const float v = x[j], mn = fmaxf(m, v);
      if (mn != -INFINITY) {  // masked (-inf) logits before the thread's first finite one: keep (-inf, 0), do not form -inf - -inf
        se = se * __expf(m - mn) + __expf(v - mn);
        m = mn;
      }
    }
  }
  // merge: threads that saw no element hold (-inf, 0) and drop out (an all -inf row gives NaN like torch)
  const float wm = wave_max(m);
  se = wave_sum(m == -INFINITY ? 0.f : se * __expf(m - wm));
  __shared__ float redm[4];
  if (lane == 0) { red[wave] = se; redm[wave] = wm; }
  __syncthreads();
  if (tid == 0) {
    const float M = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float tot = 0.f;
    for (int w = 0; w < 4; ++w) tot += redm[w] == -INFINITY ? 0.f : red[w] * __expf(redm[w] - M);
    ws[row] = (M + logf(tot)) - x[lab];
    ws[N + row] = 1.f;
  }
}

__global__ __launch_bounds__(256) void ce_generic_reduce_kernel(const float* __restrict__ ws, int N, float* __restrict__ out) {
  __shared__ float rs[4], rc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f, c = 0.f;
  for (int i = tid; i < N; i += 256) { s += ws[i]; c += ws[N + i]; }
  s = wave_sum(s); c = wave_sum(c);
  if (lane == 0) { rs[wave] = s; rc[wave] = c; }
  __syncthreads();
  if (tid == 0) out[0] = ((rs[0] + rs[1]) + (rs[2] + rs[3])) / ((rc[0] + rc[1]) + (rc[2] + rc[3]));  // 0/0 = NaN, like torch
}

extern "C" int mmamd_contrastive_fwd(const float* a, const float* b, const float* a_all, const float* b_all,
                                     int ld_all, const float* logit_scale, int B, int WB, int E, int label_offset,
                                     const uint8_t* row_mask, float label_smoothing, int reduction, float* logits_a,
                                     float* logits_b, float* out3, float* ws, mmamd_stream_t stream) {
  return mmamd_contrastive_fwd_ld(a, b, E, a_all, b_all, ld_all, logit_scale, B, WB, E, label_offset, row_mask, label_smoothing, reduction,
                                  logits_a, logits_b, out3, ws, stream);
}

extern "C" int mmamd_contrastive_fwd_ld(const float* a, const float* b, int ld_local, const float* a_all, const float* b_all,
                                        int ld_all, const float* logit_scale, int B, int WB, int E, int label_offset,
                                        const uint8_t* row_mask, float label_smoothing, int reduction, float* logits_a,
                                        float* logits_b, float* out3, float* ws, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(a && b && a_all && b_all && logit_scale && logits_a && logits_b && out3 && ws, MMAMD_E_BADARG,
                  "contrastive_fwd: null pointer");
  MMAMD_CHECK_ARG(B > 0 && WB >= B && E > 0 && ld_all >= E && ld_local >= E, MMAMD_E_BADARG,
                  "contrastive_fwd: bad sizes B=%d WB=%d E=%d ld_all=%d ld_local=%d", B, WB, E, ld_all, ld_local);
  MMAMD_CHECK_ARG(label_offset >= 0 && label_offset + B <= WB, MMAMD_E_BADARG,
                  "contrastive_fwd: labels %d..%d outside [0,%d)", label_offset, label_offset + B, WB);
  MMAMD_CHECK_ARG(reduction == MMAMD_REDUCE_MEAN || reduction == MMAMD_REDUCE_SUM, MMAMD_E_UNSUPPORTED,
                  "contrastive_fwd: reduction must be mean or sum");
  hipStream_t st = (hipStream_t)stream;
  const dim3 g1((WB + 127) / 128, (B + 31) / 32, 2);
  hipLaunchKernelGGL(logits_kernel, g1, dim3(256), 0, st, a, b, a_all, b_all, ld_all, logit_scale, B, WB, E, logits_a, logits_b, ld_local);
  hipLaunchKernelGGL(ce_rows_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, logits_a, logits_b, B, WB, label_offset,
                     row_mask, label_smoothing, ws);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, ws, B, row_mask, reduction, out3);
  return launch_status("contrastive_fwd");
}

extern "C" int mmamd_select_tokens(const int64_t* labels, const uint8_t* row_keep, int64_t ignore_index, int B, int L, int seq_S,
                                   int tok_offset, int32_t* idx_out, int64_t* label_out, int32_t* count_out, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(labels && idx_out && count_out && B >= 0 && L > 0 && seq_S > 0 && tok_offset >= 0 && tok_offset + L <= seq_S,
                  MMAMD_E_BADARG, "select_tokens: bad argument");
  MMAMD_CHECK_ARG((int64_t)B * seq_S < (1ll << 31), MMAMD_E_UNSUPPORTED, "select_tokens: more than 2^31 source rows");
  hipLaunchKernelGGL(select_tokens_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, row_keep, (long long)ignore_index, B, L,
                     seq_S, tok_offset, idx_out, label_out, count_out);
  return launch_status("select_tokens");
}

extern "C" int mmamd_gather_rows(const float* src, int64_t row_stride, const int32_t* idx, int n, int d, void* dst, int dst_dtype,
                                 const int64_t* zero_rows, mmamd_stream_t stream) {
  if (n == 0) return 0;
  MMAMD_CHECK_ARG(src && idx && dst && n > 0 && d > 0 && d % 4 == 0 && row_stride >= d && row_stride % 4 == 0, MMAMD_E_BADARG,
                  "gather_rows: bad argument");
  MMAMD_CHECK_ARG(aligned16(src) && aligned16(dst), MMAMD_E_ALIGN, "gather_rows: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  const dim3 grid((n + 3) / 4), block(256);
  if (dst_dtype == MMAMD_F32)
    hipLaunchKernelGGL((gather_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, src, (size_t)row_stride, idx, n, d, (float*)dst, zero_rows);
  else if (dst_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((gather_rows_kernel<bf16>), grid, block, 0, (hipStream_t)stream, src, (size_t)row_stride, idx, n, d, (bf16*)dst, zero_rows);
  else
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "gather_rows: bad dst_dtype %d", dst_dtype);
  return launch_status("gather_rows");
}

extern "C" int mmamd_cross_entropy(const float* logits, int64_t ld, const int64_t* labels, int N, int V, int64_t ignore_index,
                                   float* out_loss, float* ws, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(out_loss && N >= 0 && V > 0 && ld >= V && (N == 0 || (logits && labels && ws)), MMAMD_E_BADARG,
                  "cross_entropy: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0)
    hipLaunchKernelGGL(ce_generic_rows_kernel, dim3(N), dim3(256), 0, st, logits, (size_t)ld, labels, N, V, (long long)ignore_index, ws);
  hipLaunchKernelGGL(ce_generic_reduce_kernel, dim3(1), dim3(256), 0, st, ws, N, out_loss);
  return launch_status("cross_entropy");
}

static int launch_f32_gemm(const float* X, long long sxm, long long sxk, const float* Y, long long syn, long long syk,
                           const float* log_alpha, const float* R, int ldr, float* C, int ldc, int M, int N, int K, hipStream_t st) {
  hipLaunchKernelGGL(f32_gemm_strided_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, st, X, sxm, sxk, Y, syn, syk, log_alpha,
                     R, ldr, C, ldc, M, N, K);
  return launch_status("contrastive_bwd gemm");
}

extern "C" int mmamd_contrastive_bwd(const float* a, const float* b, const float* a_all, const float* b_all, int ld_all,
                                     const float* logit_scale, const float* logits_a, const float* logits_b, int B, int WB, int E,
                                     int label_offset, const uint8_t* row_mask, float label_smoothing, int reduction,
                                     const float* grad_out3, float* G_a, float* G_b, float* grad_a, float* grad_b,
                                     const float* add_a, const float* add_b, int ld_add, float* grad_a_all, float* grad_b_all,
                                     int ld_grad_all, int all_row0, int all_rows, float* grad_logit_scale, float* ws,
                                     mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(a && b && a_all && b_all && logit_scale && logits_a && logits_b && grad_out3 && G_a && G_b && grad_a && grad_b &&
                      grad_logit_scale && ws && B > 0 && WB >= B && E > 0 && ld_all >= E,
                  MMAMD_E_BADARG, "contrastive_bwd: bad argument");
  MMAMD_CHECK_ARG(label_offset >= 0 && label_offset + B <= WB, MMAMD_E_BADARG, "contrastive_bwd: labels out of range");
  MMAMD_CHECK_ARG((grad_a_all == nullptr) == (grad_b_all == nullptr) && all_row0 >= 0 && all_rows >= 0 && all_row0 + all_rows <= WB,
                  MMAMD_E_BADARG, "contrastive_bwd: bad gathered-gradient range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_grad_rows_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, logits_a, logits_b, B, WB, label_offset, row_mask,
                     label_smoothing, reduction, grad_out3, G_a, G_b, ws);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, st, ws, 2 * B, grad_logit_scale);
  int rc;
  if (grad_a_all != nullptr && all_rows > 0) {
    // d b_all[j] = T sum_i G_a[i, j] a[i],  d a_all[j] = T sum_i G_b[i, j] b[i]  for gathered rows j in [all_row0, all_row0 + all_rows).
    // These run FIRST so that add_a / add_b may point into grad_a_all / grad_b_all (own block joining the direct terms).
    rc = launch_f32_gemm(G_a + all_row0, 1, WB, a, 1, E, logit_scale, nullptr, 0, grad_b_all, ld_grad_all, all_rows, E, B, st);
    if (rc) return rc;
    rc = launch_f32_gemm(G_b + all_row0, 1, WB, b, 1, E, logit_scale, nullptr, 0, grad_a_all, ld_grad_all, all_rows, E, B, st);
    if (rc) return rc;
  }
  // d a = T G_a b_all (+ add_a),  d b = T G_b a_all (+ add_b):  X(m,k) = G[m*WB + k], Y(n,k) = all[k*ld_all + n]
  rc = launch_f32_gemm(G_a, WB, 1, b_all, 1, ld_all, logit_scale, add_a, ld_add, grad_a, E, B, E, WB, st);
  if (rc) return rc;
  rc = launch_f32_gemm(G_b, WB, 1, a_all, 1, ld_all, logit_scale, add_b, ld_add, grad_b, E, B, E, WB, st);
  if (rc) return rc;
  return 0;
}

extern "C" int mmamd_f32_gemm_strided(const float* X, int64_t sxm, int64_t sxk, const float* Y, int64_t syn, int64_t syk, const float* R,
                                      int ldr, float* C, int ldc, int M, int N, int K, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(X && Y && C && M > 0 && N > 0 && K > 0 && ldc >= N && (!R || ldr >= N), MMAMD_E_BADARG, "f32_gemm_strided: bad argument");
  return launch_f32_gemm(X, sxm, sxk, Y, syn, syk, nullptr, R, ldr, C, ldc, M, N, K, (hipStream_t)stream);
}

extern "C" int mmamd_cross_entropy_bwd(const float* logits, int64_t ld, const int64_t* labels, int N, int V, int64_t ignore_index,
                                       const float* grad_out, void* dlogits, int dlogits_dtype, int64_t ldd, float* ws,
                                       mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(logits && labels && grad_out && dlogits && ws && N > 0 && V > 0 && ld >= V && ldd >= V, MMAMD_E_BADARG,
                  "cross_entropy_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // kept-row count: the forward's row pass again (cheap next to the softmax below) then a one-block sum of the flags
  hipLaunchKernelGGL(ce_generic_rows_kernel, dim3(N), dim3(256), 0, st, logits, (size_t)ld, labels, N, V, (long long)ignore_index, ws);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, st, ws + N, N, ws + 2 * (size_t)N);
  if (dlogits_dtype == MMAMD_BF16)
    hipLaunchKernelGGL((ce_generic_bwd_kernel<bf16>), dim3(N), dim3(256), 0, st, logits, (size_t)ld, labels, N, V, (long long)ignore_index,
                       grad_out, ws + 2 * (size_t)N, (bf16*)dlogits, (size_t)ldd);
  else if (dlogits_dtype == MMAMD_F32)
    hipLaunchKernelGGL((ce_generic_bwd_kernel<float>), dim3(N), dim3(256), 0, st, logits, (size_t)ld, labels, N, V, (long long)ignore_index,
                       grad_out, ws + 2 * (size_t)N, (float*)dlogits, (size_t)ldd);
  else
    MMAMD_CHECK_ARG(false, MMAMD_E_BADARG, "cross_entropy_bwd: bad dlogits dtype");
  return launch_status("cross_entropy_bwd");
}
