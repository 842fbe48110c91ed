// attention.hip — multi-head self-attention forward, head dim 64, S <= 288 (gfx950).
//
// Every sequence on the path is short (77 text, 50/197/257 ViT, 275 FLAVA fusion), so the whole K and V
// of one (batch, head) fit in LDS (<= 2 x 41 KiB): they are staged ONCE per workgroup and every query tile
// of that head re-reads them from LDS only.
//
// One workgroup (4 waves) per (batch, head).  K is staged row-major (padded rows, conflict-free
// ds_read_b128), V is staged TRANSPOSED (V^T[d][key], row stride = 2 (mod 4) dwords -> conflict-free
// ds_read_b64).  Each wave owns 32 query rows at a time and walks the keys in tiles of 32 with a running
// max / running sum (flash-style), which keeps the live state at ~110 VGPRs (4 waves per SIMD) instead of
// the ~450 a materialised [32 x S] score strip costs.  BOTH products are computed swapped:
//     S^T = K . Q^T   (MFMA A = K rows,   B = Q rows)  -> lane owns ONE query, 16 keys of the tile:
//                      row max / row sum are in-lane reductions + one cross-half shuffle
//     O^T = V^T . P^T (MFMA A = V^T rows, B = P rows)  -> the P^T fragment is exactly the packed bf16 score
//                      registers the lane already holds (no LDS round trip, no permute), and the lane ends
//                      up with 4 consecutive output channels of its query row -> 8-byte stores.
// The MFMA sums over its 16 k-slots; as long as slot j of lane-half h carries the SAME key for both
// operands the order of keys inside a slot group is irrelevant, which is what makes the register reuse
// above legal.  exp2 with log2(e)/sqrt(dh) folded into the score scale.
// Replaces F.scaled_dot_product_attention under nn.MultiheadAttention (reference call sites:
// models/clip/image_encoder.py:108, models/clip/text_encoder.py:121 with is_causal=True).
#include <type_traits>

#include "common.h"

namespace mmamd {

constexpr int kDh = 64;
constexpr int kKStride = 72;  // bf16 elements per K row in LDS (144 B: 16-B aligned, 9 slots -> conflict-free)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ds_read_b64_tr_b16 (gfx950 LDS transpose read): within each 16-lane group, lane i points at 4 contiguous bf16 — row i>>2, columns
// 4(i&3)..+3 of a [4][16] block — and receives column i of that block (its 4 rows).  With a row-major [token][channel] image and
// tr_off() below, a lane gets the 4 consecutive TOKENS of ONE channel an MFMA operand with the token index as k needs: no transposed
// copy of the tile in LDS (and none of the 2-byte scattered stores that build one).  Addresses must be 8-byte aligned.
__device__ __forceinline__ uint2 lds_tr_b64(const bf16* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}
// element offset of this lane's 4-element piece inside a [rows][stride] image, relative to (row0, col0) of a [4 + 4*half][32] block:
// lane -> row ((lane&15)>>2) + 4*(lane>>5), column 16*((lane>>4)&1) + 4*(lane&3); the lane then owns channel col0 + (lane&31)
__device__ __forceinline__ int tr_off(int lane, int stride) {
  return (((lane & 15) >> 2) + 4 * (lane >> 5)) * stride + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
}

// Persistent kernel: gridDim.x workgroups (2 per CU) walk the (batch, head) items.  While item i is being computed out
// of LDS, the K/V rows of item i + grid are already in flight into registers (issue-early / write-late staging: the
// r01 ablation showed ~50 of 150 us were the exposed K/V load latency at the head of every workgroup), and each wave's
// next Q tile is prefetched the same way.
// Softmax VALU diet (the first version spent ~270 instructions per 32-key tile for 8 MFMAs): the score scale is folded
// into the exp2 argument (one FMA), masking code exists only in the peeled tail / diagonal tile, and the O / l rescale
// runs only when some row's maximum grows by more than 2^8 (deferred max: P stays <= 256, exact in the normalisation).
// ABL (timing experiments only, results WRONG): 1 = V staged row-major, 2 = no exp, 4 = no K/V global loads, 128 = serial key loop;
// 8 = phase timers (results right; `lse` receives 8 floats of s_memtime cycles per wave instead of the log-sum-exp)
template <int NKT, bool CAUSAL, int ABL = 0, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void attention_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                            int S, int H, int BH, float scale_log2e, float* __restrict__ lse,
                                                            int lse_stride, int lse_tile) {
  constexpr int SP = NKT * 32;   // padded key count
  constexpr int VS = SP + 4;     // V^T row stride (elements): (SP+4)/2 dwords = 2 (mod 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);                       // [SP][kKStride]
  // V: row-major [SP][kKStride] read through ds_read_b64_tr_b16 (no transposed copy, staged with the same 16-byte stores as K);
  // S > 256 keeps the transposed [64][VS] image (two workgroups per CU only fit with its smaller footprint)
  constexpr bool VTR = NKT <= 8 && (ABL & 1) == 0;  // (ABL & 256 keeps it)
  bf16* Vt = reinterpret_cast<bf16*>(smem + SP * kKStride * 2);   // VTR: [SP][kKStride], else [64][VS]

  const int D = H * kDh;
  // ABL & 256 (layout experiment): qkv given HEAD-major, [3H][B*S][64] — every (batch, head) slice of q, k, v is one contiguous block
  constexpr bool HM = (ABL & 256) != 0;
  const size_t Mtot = (size_t)(BH / H) * S;
  const size_t row_stride = HM ? (size_t)kDh : (size_t)3 * D;
  const size_t k_off = HM ? (size_t)H * Mtot * kDh : (size_t)D, v_off = 2 * k_off;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nqt = (S + 31) >> 5;
  const int srow = tid >> 3, schunk = tid & 7;

  auto item_base = [&](int item) {
    if constexpr (HM) return qkv + ((size_t)(item % H) * Mtot + (size_t)(item / H) * S) * kDh;
    else return qkv + (size_t)(item / H) * S * row_stride + (item % H) * kDh;
  };

  // NW waves stage NW*8 rows per pass; with NW = 8 (16 waves per CU at 2 resident workgroups) the waits of one wave (43 % of
  // wave-cycles in the 4-wave form: PMC SQ_WAIT_ANY) are covered by three others on the same SIMD instead of one
  constexpr int RPP = NW * 8, NPASS = (SP + RPP - 1) / RPP;
  bf16x8 kreg[NPASS], vreg[NPASS];
  auto load_item = [&](const bf16* base) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = srow + RPP * i;
#pragma unroll
      for (int j = 0; j < 8; ++j) { kreg[i][j] = (bf16)0.f; vreg[i][j] = (bf16)0.f; }
      if (r < S && (ABL & 4) == 0) {
        kreg[i] = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + k_off + schunk * 8);
        vreg[i] = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + v_off + schunk * 8);
      }
    }
  };
  auto store_item = [&]() {  // rows >= S are zero so padded keys contribute exact 0
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = srow + RPP * i;
      if (r >= SP) continue;
      *reinterpret_cast<bf16x8*>(Ks + r * kKStride + schunk * 8) = kreg[i];
      if constexpr (VTR) {
        *reinterpret_cast<bf16x8*>(Vt + r * kKStride + schunk * 8) = vreg[i];
      } else if constexpr ((ABL & 1) != 0) {
        *reinterpret_cast<bf16x8*>(Vt + r * 64 + schunk * 8) = vreg[i];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(schunk * 8 + j) * VS + r] = vreg[i][j];
      }
    }
  };
  auto load_q = [&](const bf16* base, int qt, bf16x8 (&qf)[4]) {
    const int q = qt * 32 + l31;
    const int qc = q < S ? q : S - 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * row_stride + 16 * t + 8 * half);
  };

  // V operand of the PV MFMA: channel nt*32 + l31, keys key0..key0+3 and key0+8..key0+11 with key0 = kt*32 + 16*jj + 4*half
  const bf16* vtr = Vt + tr_off(lane, kKStride);
  auto read_v = [&](int kt, int jj, int nt) -> bf16x8 {
    uint2 v0, v1;
    if constexpr (VTR) {
      const bf16* vp = vtr + (kt * 32 + 16 * jj) * kKStride + nt * 32;
      v0 = lds_tr_b64(vp);
      v1 = lds_tr_b64(vp + 8 * kKStride);
    } else {
      const bf16* vrow = Vt + (nt * 32 + l31) * VS + kt * 32 + 16 * jj + 4 * half;
      v0 = *reinterpret_cast<const uint2*>(vrow);
      v1 = *reinterpret_cast<const uint2*>(vrow + 8);
    }
    u32x4 vw;
    vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
    return __builtin_bit_cast(bf16x8, vw);
  };

  int item = blockIdx.x;
  if (item >= BH) return;
  load_item(item_base(item));
  bf16x8 qcur[4], qnext[4];
  load_q(item_base(item), wave, qcur);

  constexpr bool TIMED = (ABL & 8) != 0;
  uint64_t tacc[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> uint64_t { if constexpr (TIMED) return __builtin_amdgcn_s_memtime(); else return 0; };
  const uint64_t t_begin = now();
  for (; item < BH; item += gridDim.x) {
    const bf16* base = item_base(item);
    const int b = item / H, h = item - b * H;
    const uint64_t t0 = now();
    store_item();
    const uint64_t t1 = now();
    __syncthreads();
    const uint64_t t2 = now();
    tacc[0] += t1 - t0; tacc[1] += t2 - t1;
    const int nitem = item + gridDim.x;
    if (nitem < BH) load_item(item_base(nitem));  // in flight during the whole compute phase below

    for (int qt = wave; qt < nqt; qt += NW) {
      // prefetch the Q fragments this wave needs next: its next tile of this item, else its first tile of the next item
      if (qt + NW < nqt) load_q(base, qt + NW, qnext);
      else if (nitem < BH && wave < nqt) load_q(item_base(nitem), wave, qnext);
      const int q = qt * 32 + l31;

      float m = -INFINITY, lsum = 0.f;  // running reference (scaled log2 domain) and running sum of this lane's row
      f32x16 ot[2];                     // ot[nt][r] = O[q][channel nt*32 + (r&3) + 8*(r>>2) + 4*half]
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[nt][r] = 0.f;
      const uint64_t ta = now();

      auto tile = [&](int kt, auto masked) {
        // ---- S^T tile: st[r] = score(query q, key kt*32 + (r&3) + 8*(r>>2) + 4*half)
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + 16 * t);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qcur[t], st, 0, 0, 0);
        }
        if constexpr (decltype(masked)::value) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= S || (CAUSAL && key > q)) st[r] = -INFINITY;
          }
        }
        float tmax = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float ts = tmax * scale_log2e;
        // deferred max: move the reference only when a row grew by more than 2^8 (first tile: m = -inf -> always)
        if (__any(ts > m + 8.0f)) {
          const float m_new = fmaxf(m, ts);
          const float alpha = __builtin_amdgcn_exp2f(m - m_new);
          m = m_new;
          lsum *= alpha;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[nt][r] *= alpha;
        }
        // ---- P = exp2(s*c - m), packed to bf16 pairs: pk[2g], pk[2g+1] = the 4 keys of accumulator group g
        uint32_t pk[8];
        float psum = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr ((ABL & 2) != 0) e[j] = st[4 * g + j];
            else e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[4 * g + j], scale_log2e, -m));
            psum += e[j];
          }
          bf16x2 p0, p1;
          p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
          pk[2 * g] = __builtin_bit_cast(uint32_t, p0);
          pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
        }
        lsum += psum;
        // ---- O^T += V^T . P^T over this tile's 32 keys (two MFMA k-groups of 16)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          // slots 0-3 = keys key0..key0+3, slots 4-7 = keys key0+8..key0+11 (per lane half), both operands
          const int key0 = kt * 32 + 16 * jj + 4 * half;
          u32x4 pw;
          pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
          const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_v(kt, jj, nt), pf, ot[nt], 0, 0, 0);
        }
      };

      if constexpr (!CAUSAL && (ABL & ~(8 | 256)) == 0) {
        // Software-pipelined key loop (fully unrolled).  The serial form below spends 43 % of its wave-cycles in s_waitcnt (r01 PMC,
        // profiles/r01_pmc_attention_fwd_before_pipelining.txt): ds_read -> 4 chained MFMAs -> ~100 softmax VALU -> ds_read -> 4 MFMAs, nothing
        // overlapping inside a wave.  Here QK^T of tile kt+1 is issued BEFORE the softmax of tile kt (its K fragments were read one
        // tile earlier) and the V fragments of tile kt are read before its softmax, so the matrix pipe and the LDS work under the VALU.
        auto read_k = [&](int kt, bf16x8 (&kf)[4]) {
          const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
          for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(krow + 16 * t);
        };
        auto qk = [&](const bf16x8 (&kf)[4]) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t], qcur[t], acc, 0, 0, 0);
          return acc;
        };
        bf16x8 kf[4];
        f32x16 st_next;
        auto body = [&](int kt, auto last) {
          constexpr bool kLast = decltype(last)::value;
          f32x16 st = st_next;
          bf16x8 vf[2][2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) vf[jj][nt] = read_v(kt, jj, nt);
          if constexpr (!kLast) {
            st_next = qk(kf);
            read_k(kt + 2 < NKT ? kt + 2 : NKT - 1, kf);  // (the clamped re-read of the last tile is never used)
          } else {  // only the last tile can hold padded keys
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (key >= S) st[r] = -INFINITY;
            }
          }
          float tmax = st[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
          tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
          const float ts = tmax * scale_log2e;
          if (__any(ts > m + 8.0f)) {
            const float m_new = fmaxf(m, ts);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            lsum *= alpha;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int r = 0; r < 16; ++r) ot[nt][r] *= alpha;
          }
          uint32_t pk[8];
          f32x2 ps2 = {0.f, 0.f};
          const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m, -m};
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            f32x2 a = {st[2 * g], st[2 * g + 1]};
            a = __builtin_elementwise_fma(a, sc2, nm2);
            f32x2 e;
            e[0] = __builtin_amdgcn_exp2f(a[0]);
            e[1] = __builtin_amdgcn_exp2f(a[1]);
            ps2 += e;
            bf16x2 p;
            p[0] = (bf16)e[0]; p[1] = (bf16)e[1];
            pk[g] = __builtin_bit_cast(uint32_t, p);
          }
          lsum += ps2[0] + ps2[1];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            u32x4 pw;
            pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jj][nt], pf, ot[nt], 0, 0, 0);
          }
        };
        read_k(0, kf);
        st_next = qk(kf);
        if constexpr (NKT > 1) read_k(1, kf);
#pragma unroll 1
        for (int kt = 0; kt < NKT - 1; ++kt) body(kt, std::false_type{});
        body(NKT - 1, std::true_type{});
      } else {
      // tiles that cannot contain a dead key run the mask-free body; the (at most one) partial / diagonal tile is peeled
      const int kt_end = CAUSAL ? (qt + 1 < NKT ? qt + 1 : NKT) : NKT;
      const int kt_last = kt_end - 1;
      const bool last_masked = ((kt_last + 1) * 32 > S) || CAUSAL;
      const int kt_plain = last_masked ? kt_last : kt_end;
#pragma unroll 1
      for (int kt = 0; kt < kt_plain; ++kt) tile(kt, std::false_type{});
      if (last_masked) tile(kt_last, std::true_type{});

      }
      // ---- normalise and store: lane owns row q, channels nt*32 + 8g + 4*half + {0..3}
      lsum += __shfl_xor(lsum, 32);
      const float inv = 1.0f / lsum;
      const uint64_t tb = now();
      tacc[2] += tb - ta;
      // training: log2-domain log-sum-exp for the backward kernels (m is the running REFERENCE, not necessarily the max: m + log2(sum) is exact either way)
      if (!TIMED && lse != nullptr && half == 0 && q < S) lse[((size_t)b * H + h) * lse_stride + (size_t)(q >> 5) * lse_tile + (q & 31)] = m + __builtin_amdgcn_logf(lsum);
      if (q < S) {
        bf16* orow = out + ((size_t)b * S + q) * D + h * kDh;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ot[nt][4 * g + j] * inv;
            store4(orow + nt * 32 + 8 * g + 4 * half, o);
          }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) qcur[t] = qnext[t];
      // compiler barrier: without it the scheduler stretches live ranges across the query-tile seam and the pipelined key loop
      // spills 56 VGPRs at NKT = 7
      asm volatile("" ::: "memory");
      if constexpr (TIMED) tacc[3] += now() - tb;
    }
    const uint64_t t3 = now();
    __syncthreads();  // every wave is done with this item's K/V before the next item overwrites LDS
    tacc[4] += now() - t3;
  }
  if constexpr (TIMED) {
    tacc[5] = now() - t_begin;
    if (lane == 0 && lse != nullptr)
      for (int i = 0; i < 6; ++i) lse[((size_t)blockIdx.x * NW + wave) * 8 + i] = (float)tacc[i];
  }
}


// ---------------------------------------------------------------------------------------------------------
// Attention variant that ALSO returns the probabilities and honours a key-padding mask — what FLAVA's encoders need
// (modules/layers/attention.py:185-241 returns `attn`, models/flava/image_encoder.py:217-222 always asks for it;
// BERTTextEncoder masks padded keys, modules/encoders/bert_text_encoder.py:86-91).  Two passes over the key tiles per
// 32-query tile: pass 1 = row max / row sum (QK^T only), pass 2 = recompute QK^T, emit NORMALISED probabilities
// (fp32 or bf16, [B,H,S,S]) and accumulate P.V.  Same swapped-operand register layout as the kernel above.
// key_mask: uint8 [B,S], 0 = masked key (NULL = no mask).  HBM-bound by the S^2 probability write.

// 4 consecutive probabilities to a row that is only element-aligned (S = 197: 788-byte rows).  gfx950 global stores take any
// dword-aligned address for dwordx4 (unaligned access mode), so the fp32 case is one 16-byte store.
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void store_probs4(float* dst, f32x4 v) { *reinterpret_cast<f32x4_a4*>(dst) = v; }
__device__ __forceinline__ void store_probs4(bf16* dst, f32x4 v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) dst[j] = (bf16)v[j];
}

template <int NKT, typename TP, bool PIPE = true>
__global__ __launch_bounds__(256) void attention_probs_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ key_mask,
                                                              bf16* __restrict__ out, TP* __restrict__ probs, int S, int H,
                                                              float scale_log2e) {
  constexpr int SP = NKT * 32;
  constexpr int VS = SP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vt = reinterpret_cast<bf16*>(smem + SP * kKStride * 2);
  float* Mk = reinterpret_cast<float*>(smem + SP * kKStride * 2 + 64 * VS * 2);  // additive key mask: 0 or -inf, [SP]
  // per-wave 32x32 fp32 strip (row pitch 36 floats: conflict-free b128 both ways): the probability tile is transposed through
  // it so that one wave-instruction stores 8 rows x 128 contiguous bytes instead of 32 rows x 2 x 4 bytes
  constexpr int PSTR = 36;
  float* Ps = reinterpret_cast<float*>(smem + SP * kKStride * 2 + 64 * VS * 2 + SP * 4) + (threadIdx.x >> 6) * (32 * PSTR);

  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * kDh;
  const size_t row_stride = (size_t)3 * D;
  const bf16* base = qkv + (size_t)b * S * row_stride + h * kDh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  for (int r = tid >> 3; r < SP; r += 32) {
    const int c = tid & 7;
    bf16x8 kv, vv;
#pragma unroll
    for (int j = 0; j < 8; ++j) { kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; }
    if (r < S) {
      kv = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + D + c * 8);
      vv = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + 2 * D + c * 8);
    }
    *reinterpret_cast<bf16x8*>(Ks + r * kKStride + c * 8) = kv;
#pragma unroll
    for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * VS + r] = vv[j];
  }
  for (int k = tid; k < SP; k += 256)
    Mk[k] = (k < S && (key_mask == nullptr || key_mask[(size_t)b * S + k] != 0)) ? 0.f : -INFINITY;
  __syncthreads();

  const int l31 = lane & 31, half = lane >> 5;
  const int nqt = (S + 31) >> 5;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q = qt * 32 + l31;
    const int qc = q < S ? q : S - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * row_stride + 16 * t + 8 * half);

    auto scores = [&](int kt, f32x16& st) {  // masked, scaled scores (log2 domain) of this lane's 16 keys of tile kt
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + 16 * t);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], st, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = st[r] * scale_log2e + Mk[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
    };

    float m = -INFINITY, lsum = 0.f, inv = 0.f;
    f32x16 ot[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[nt][r] = 0.f;
    TP* pbase = probs + ((size_t)b * H + h) * S * S;
    // One key tile of pass 1 (row max / sum) and of pass 2 (normalised probabilities out, P.V) on finished scores `st`; the arithmetic
    // per row is the same sequence of operations in the serial and the pipelined loops (bit-identical results).
    auto stats_tile = [&](const f32x16& st) {
      float tmax = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m, tmax);
      // a fully masked prefix keeps m = -inf: exp2(-inf - (-inf)) would be NaN, so guard the rescale factor
      const float alpha = (m_new == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ps += (m_new == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st[r] - m_new);
      lsum = lsum * alpha + ps;
      m = m_new;
    };
    auto read_vf = [&](int kt, bf16x8 (&vf)[2][2]) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16* vrow = Vt + (nt * 32 + l31) * VS + kt * 32 + 16 * jj + 4 * half;
          const uint2 v0 = *reinterpret_cast<const uint2*>(vrow);
          const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + 8);
          u32x4 vw;
          vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
          vf[jj][nt] = __builtin_bit_cast(bf16x8, vw);
        }
    };
    auto emit_tile = [&](int kt, const f32x16& st, const bf16x8 (&vf)[2][2]) {
      uint32_t pk[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(st[4 * g + j] - m) * inv;
        if (probs != nullptr) *reinterpret_cast<f32x4*>(Ps + l31 * PSTR + 8 * g + 4 * half) = f32x4{e[0], e[1], e[2], e[3]};
        bf16x2 p0, p1;
        p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
        pk[2 * g] = __builtin_bit_cast(uint32_t, p0);
        pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
      }
      // P.V first: its four MFMAs cover the strip's LDS write -> read round trip (LDS operations of one wave execute in order, so the
      // reads below see the strip without a wait, and the next tile's strip writes come after them)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        u32x4 pw;
        pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jj][nt], pf, ot[nt], 0, 0, 0);
      }
      if (probs != nullptr) {  // strip -> global: lane = (row lane>>3 of 8, 4 keys at (lane&7)*4); rows are only 4-byte aligned
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
          const f32x4 pv = *reinterpret_cast<const f32x4*>(Ps + row * PSTR + c4);
          const int qq = qt * 32 + row, key = kt * 32 + c4;
          if (qq < S) {
            TP* dst = pbase + (size_t)qq * S + key;
            if (key + 3 < S) {
              store_probs4(dst, pv);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (key + j < S) dst[j] = (TP)pv[j];
            }
          }
        }
      }
    };
    if constexpr (PIPE) {
      // Software-pipelined key loops (as in attention_fwd_kernel): the QK^T MFMAs of tile kt+1 are issued before the softmax VALU of
      // tile kt and the K fragments are read a tile ahead, so the matrix pipe and the LDS work under the VALU instead of in front of it.
      // Measured (tools/probs_bench.py, B = 128, bit-identical outputs): S = 197 117 us either way (the fp32 probability write, 238 MB, is
      // what the kernel waits for: 65 us without it), S = 275 267 vs 274 us, 174 vs 181 us without probabilities.
      auto read_k = [&](int kt, bf16x8 (&kf)[4]) {
        const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
        for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(krow + 16 * t);
      };
      auto qk = [&](const bf16x8 (&kf)[4]) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t], qf[t], acc, 0, 0, 0);
        return acc;
      };
      auto mask_scale = [&](int kt, f32x16& st) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(Mk + kt * 32 + 8 * g + 4 * half);
#pragma unroll
          for (int j = 0; j < 4; ++j) st[4 * g + j] = st[4 * g + j] * scale_log2e + mk[j];
        }
      };
      bf16x8 kf[4];
      f32x16 st_next;
      read_k(0, kf);
      st_next = qk(kf);
      if constexpr (NKT > 1) read_k(1, kf);
#pragma unroll 1
      for (int kt = 0; kt < NKT; ++kt) {
        f32x16 st = st_next;
        if (kt + 1 < NKT) {
          st_next = qk(kf);
          read_k(kt + 2 < NKT ? kt + 2 : 0, kf);  // (past the end: tile 0 again, for pass 2)
        }
        mask_scale(kt, st);
        stats_tile(st);
      }
      lsum += __shfl_xor(lsum, 32);
      inv = 1.0f / lsum;  // all keys masked -> 0 * inf = NaN, as the reference's softmax of an all -inf row
      if constexpr (NKT == 1) read_k(0, kf);
      st_next = qk(kf);  // kf holds tile 0 again
      if constexpr (NKT > 1) read_k(1, kf);
#pragma unroll 1
      for (int kt = 0; kt < NKT; ++kt) {
        f32x16 st = st_next;
        bf16x8 vf[2][2];
        read_vf(kt, vf);
        if (kt + 1 < NKT) {
          st_next = qk(kf);
          read_k(kt + 2 < NKT ? kt + 2 : NKT - 1, kf);
        }
        mask_scale(kt, st);
        emit_tile(kt, st, vf);
      }
    } else {
#pragma unroll 1
      for (int kt = 0; kt < NKT; ++kt) {
        f32x16 st;
        scores(kt, st);
        stats_tile(st);
      }
      lsum += __shfl_xor(lsum, 32);
      inv = 1.0f / lsum;
#pragma unroll 1
      for (int kt = 0; kt < NKT; ++kt) {
        f32x16 st;
        scores(kt, st);
        bf16x8 vf[2][2];
        read_vf(kt, vf);
        emit_tile(kt, st, vf);
      }
    }
    if (q < S) {
      bf16* orow = out + ((size_t)b * S + q) * D + h * kDh;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ot[nt][4 * g + j];
          store4(orow + nt * 32 + 8 * g + 4 * half, o);
        }
    }
  }
}

static int g_attn_probs_serial = 0;  // mmamd_debug_set_attn_variant(512): the serial key loops (A/B of the pipelined form; same results)
static int g_attn_probs_twopass = 0;  // mmamd_debug_set_attn_variant(514): unmasked fp32 probabilities from the two-pass kernel too (515: back to flash + one pass)

template <int NKT, typename TP, bool PIPE = true>
static int launch_attn_probs(const void* qkv, const uint8_t* key_mask, void* out, void* probs, int B, int S, int H, float scale,
                             hipStream_t st) {
  if constexpr (PIPE) {
    if (g_attn_probs_serial) return launch_attn_probs<NKT, TP, false>(qkv, key_mask, out, probs, B, S, H, scale, st);
  }
  constexpr int SP = NKT * 32;
  constexpr int smem = SP * kKStride * 2 + 64 * (SP + 4) * 2 + SP * 4 + 4 * 32 * 36 * 4;
  auto kern = attention_probs_kernel<NKT, TP, PIPE>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), smem, st, (const bf16*)qkv, key_mask, (bf16*)out, (TP*)probs, S, H,
                     scale * 1.4426950408889634f);
  return launch_status("attention_probs_fwd");
}


// ---------------------------------------------------------------------------------------------------------
// General attention (CoCa: cross-attention with Sq != Sk, the attention pooler's 96-wide heads and batch-shared learned
// queries, causal decoders with a padding-aware mask).  Same two-pass structure and register layout as the kernel above,
// templated on the head dimension; q / k / v are separate strided bf16 matrices:
//   q row (b, i) at q + b*q_bs + i*ldq (+ h*DH)     q_bs = 0: the same queries for every sample (AttentionPooler)
//   k/v row (b, j) at k/v + b*kv_bs + j*ldk/ldv (+ h*DH)
// Masks (all optional, combined): causal (key j <= query i), key_mask uint8 [B,Sk], full_mask uint8 [B or 1, Sq, Sk]
// (0 = masked; fm_bs = 0 broadcasts one [Sq,Sk] mask over the batch).
struct AttnX {
  const bf16 *q, *k, *v;
  bf16* out;
  void* probs;
  const uint8_t *key_mask, *full_mask;
  float* lse;  // optional [B,H,Sq]: log2-domain log-sum-exp of the scaled scores (saved for the backward kernels)
  long long q_bs, kv_bs, fm_bs;
  int ldq, ldk, ldv, ldo, Sq, Sk, H, causal;
  int km_last = 0;  // 1: key_mask applies to the last query row only (argument `causal` bit 1)
  float scale_log2e;
  AttnDrop drop;  // training-time dropout on the normalised probabilities (thresh = 0: none); common.h
  // head_mask of the reference's scaled_dot_product_attention (modules/layers/attention.py:236-237): fp32, multiplied into the probabilities AFTER
  // softmax and dropout (what is returned and what multiplies V); element strides of its [b, h, q, k] broadcast (0 = broadcast dimension)
  const float* hmask = nullptr;
  long long hm_sb = 0, hm_sh = 0, hm_sq = 0, hm_sk = 0;
};

template <int NKT, int DH, typename TP>
__global__ __launch_bounds__(256) void attention_x_kernel(const AttnX p) {
  constexpr int SP = NKT * 32;
  constexpr int KS = DH + 8;   // bf16 per K row in LDS: (DH+8)*2 B = an odd number of 16-B slots -> conflict-free b128 reads
  constexpr int VS = SP + 4;
  constexpr int CPR = DH / 8;  // 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vt = reinterpret_cast<bf16*>(smem + SP * KS * 2);
  float* Mk = reinterpret_cast<float*>(smem + SP * KS * 2 + DH * VS * 2);

  const int bh = blockIdx.x;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Sq = p.Sq, Sk = p.Sk;
  const bf16* kb = p.k + (size_t)b * p.kv_bs + h * DH;
  const bf16* vb = p.v + (size_t)b * p.kv_bs + h * DH;
  const bf16* qb = p.q + (size_t)b * p.q_bs + h * DH;

  // K / V staging: every 16-byte load of the item is issued before the first is stored to LDS, unconditionally on a clamped row (r06: the loop
  // loaded inside `if (r < Sk)` and stored at once -- one exposed global round trip per trip, 8 to 14 of them in a row before any compute)
  {
    constexpr int NIT = (SP * CPR + 255) / 256;
    bf16x8 kr[NIT], vr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * 256;
      i = i < SP * CPR ? i : SP * CPR - 1;
      const int r = i / CPR, c = i - r * CPR;
      const int rc = r < Sk ? r : Sk - 1;
      kr[it] = *reinterpret_cast<const bf16x8*>(kb + (size_t)rc * p.ldk + c * 8);
      vr[it] = *reinterpret_cast<const bf16x8*>(vb + (size_t)rc * p.ldv + c * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      if (i < SP * CPR) {
        const int r = i / CPR, c = i - r * CPR;
        bf16x8 kv = kr[it], vv = vr[it];
        if (r >= Sk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; }
        }
        *reinterpret_cast<bf16x8*>(Ks + r * KS + c * 8) = kv;
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * VS + r] = vv[j];
      }
    }
  }
  for (int k = tid; k < SP; k += 256)
    Mk[k] = (k < Sk && (p.key_mask == nullptr || p.key_mask[(size_t)b * Sk + k] != 0)) ? 0.f : -INFINITY;
  __syncthreads();

  const int l31 = lane & 31, half = lane >> 5;
  const int nqt = (Sq + 31) >> 5;
  const uint8_t* fm = p.full_mask ? p.full_mask + (size_t)b * p.fm_bs : nullptr;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q = qt * 32 + l31;
    const int qc = q < Sq ? q : Sq - 1;
    bf16x8 qf[DH / 16];
#pragma unroll
    for (int t = 0; t < DH / 16; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qc * p.ldq + 16 * t + 8 * half);
    const int kt_hi = p.causal ? (qt + 1 < NKT ? qt + 1 : NKT) : NKT;  // tiles above the diagonal are fully masked

    auto scores = [&](int kt, f32x16& st) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const bf16* krow = Ks + (kt * 32 + l31) * KS + 8 * half;
#pragma unroll
      for (int t = 0; t < DH / 16; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + 16 * t);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], st, 0, 0, 0);
      }
      // full [Sq, Sk] mask: the 4 consecutive keys of a register group as ONE (unaligned) 32-bit load -- the byte-per-score form was 16 dependent global
      // loads per lane and tile in both passes (CoCa's padding-aware causal mask: 122 us average for the S = 77 decoder self-attention, r04)
      uint32_t fmw[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
      if (fm != nullptr && Sk >= 4) {
        // branch-free (r06): the word is loaded from a clamped key index and shifted, so that the four loads of a tile are straight-line code, issued
        // together and waited for once -- inside the per-lane `key0 + 3 < Sk` branch each of them was waited for where it was issued
        const uint8_t* row = fm + (size_t)qc * Sk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key0 = kt * 32 + 8 * g + 4 * half;
          const int kc = key0 + 3 < Sk ? key0 : Sk - 4;
          uint32_t w;
          __builtin_memcpy(&w, row + kc, 4);
          const int sh = key0 - kc;                      // 0 inside the row; 1..3 at its ragged end; >= 4 past it
          fmw[g] = sh < 4 ? w >> (8 * sh) : 0u;          // keys >= Sk read as masked (Mk is -inf there anyway)
        }
      } else if (fm != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key0 = kt * 32 + 8 * g + 4 * half;
          const uint8_t* src = fm + (size_t)qc * Sk + key0;
          if (key0 + 3 < Sk) {
            __builtin_memcpy(&fmw[g], src, 4);
          } else {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (key0 + j < Sk) w |= (uint32_t)src[j] << (8 * j);
            fmw[g] = w;  // keys >= Sk read as masked (Mk is -inf there anyway)
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float add = Mk[key];
        if (p.km_last && qc != Sq - 1) add = key < Sk ? 0.f : -INFINITY;  // the key-padding mask binds the LAST query row only (CoCa's CLS row)
        if (p.causal && key > qc) add = -INFINITY;
        if (((fmw[r >> 2] >> (8 * (r & 3))) & 0xffu) == 0) add = -INFINITY;
        st[r] = st[r] * p.scale_log2e + add;
      }
    };

    float m = -INFINITY, lsum = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < kt_hi; ++kt) {
      f32x16 st;
      scores(kt, st);
      float tmax = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m, tmax);
      const float alpha = (m_new == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ps += (m_new == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st[r] - m_new);
      lsum = lsum * alpha + ps;
      m = m_new;
    }
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / lsum;
    if (p.lse != nullptr && half == 0 && q < Sq) p.lse[((size_t)b * p.H + h) * Sq + q] = m + __builtin_amdgcn_logf(lsum);

    f32x16 ot[DH / 32];
#pragma unroll
    for (int nt = 0; nt < DH / 32; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[nt][r] = 0.f;
    TP* prow = p.probs ? reinterpret_cast<TP*>(p.probs) + (((size_t)b * p.H + h) * Sq + (size_t)qc) * Sk : nullptr;
#pragma unroll 1
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt >= kt_hi) {  // causal tail: probabilities are exactly 0, nothing to accumulate
        if (prow != nullptr && q < Sq) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key < Sk) prow[key] = (TP)0.f;
          }
        }
        continue;
      }
      f32x16 st;
      scores(kt, st);
      uint32_t pk[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(st[4 * g + j] - m) * inv;
        if (p.drop.thresh != 0) {  // (uniform) P' = P * keep / (1 - p): what the reference returns and multiplies with V (attention.py:234-239)
          const Philox4 rr = attn_drop_block(p.drop, ((long long)b * p.H + h) * Sq + qc, (Sk + 3) >> 2, kt * 32 + 8 * g + 4 * half);
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = rr.v[j] >= p.drop.thresh ? e[j] * p.drop.scale : 0.f;
        }
        if (p.hmask != nullptr) {  // (uniform)
          const float* hm = p.hmask + (long long)b * p.hm_sb + (long long)h * p.hm_sh + (long long)qc * p.hm_sq;
          const int key = kt * 32 + 8 * g + 4 * half;
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] *= key + j < Sk ? hm[(long long)(key + j) * p.hm_sk] : 0.f;
        }
        if (prow != nullptr && q < Sq) {
          const int key = kt * 32 + 8 * g + 4 * half;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (key + j < Sk) prow[key + j] = (TP)e[j];
        }
        bf16x2 p0, p1;
        p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
        pk[2 * g] = __builtin_bit_cast(uint32_t, p0);
        pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int key0 = kt * 32 + 16 * jj + 4 * half;
        u32x4 pw;
        pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int nt = 0; nt < DH / 32; ++nt) {
          const bf16* vrow = Vt + (nt * 32 + l31) * VS + key0;
          const uint2 v0 = *reinterpret_cast<const uint2*>(vrow);
          const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + 8);
          u32x4 vw;
          vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
          ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, ot[nt], 0, 0, 0);
        }
      }
    }
    if (q < Sq) {
      bf16* orow = p.out + ((size_t)b * Sq + q) * p.ldo + h * DH;
#pragma unroll
      for (int nt = 0; nt < DH / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ot[nt][4 * g + j];
          store4(orow + nt * 32 + 8 * g + 4 * half, o);
        }
    }
  }
}

template <int NKT, int DH, typename TP>
static int launch_attn_x(const AttnX& p, int B, hipStream_t st) {
  constexpr int SP = NKT * 32;
  constexpr int smem = SP * (DH + 8) * 2 + DH * (SP + 4) * 2 + SP * 4;
  auto kern = attention_x_kernel<NKT, DH, TP>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  hipLaunchKernelGGL(kern, dim3(B * p.H), dim3(256), smem, st, p);
  return launch_status("attention_x_fwd");
}

template <int DH, typename TP>
static int dispatch_attn_x(const AttnX& p, int B, hipStream_t st) {
  switch ((p.Sk + 31) / 32) {
    case 1: return launch_attn_x<1, DH, TP>(p, B, st);
    case 2: return launch_attn_x<2, DH, TP>(p, B, st);
    case 3: return launch_attn_x<3, DH, TP>(p, B, st);
    case 4: return launch_attn_x<4, DH, TP>(p, B, st);
    case 5: return launch_attn_x<5, DH, TP>(p, B, st);
    case 6: return launch_attn_x<6, DH, TP>(p, B, st);
    case 7: return launch_attn_x<7, DH, TP>(p, B, st);
    case 8: return launch_attn_x<8, DH, TP>(p, B, st);
    case 9: return launch_attn_x<9, DH, TP>(p, B, st);
  }
  set_error("attention_x: Sk=%d > 288 not supported", p.Sk);
  return MMAMD_E_UNSUPPORTED;
}


// ---------------------------------------------------------------------------------------------------------
// Attention backward: head dim 64, packed qkv like the forward, no atomics — two kernels, each
// with the forward's structure (one workgroup per (batch, head), whole head in LDS, 32-row tiles per wave, every product on
// v_mfma_f32_32x32x16_bf16 with the "lane owns one row" layouts):
//   P = exp2(s2 - L2[q])            s2 = scale*log2e * q.k (+ mask), L2 = the forward's saved log2-sum-exp
//   Dq[q] = sum_d dO[q,d] O[q,d];   dP = dO V^T;   dS = P (dP - Dq)
//   dQ = scale dS K  (kernel 1: lane = query, loop over key tiles; K rows and V rows in LDS, K^T operands by transpose reads)
//   dV = P^T dO,  dK = scale dS^T Q  (kernel 2: lane = key, loop over query tiles; Q rows and dO rows in LDS, Q^T / dO^T operands by
//   transpose reads).  The first version kept explicit transposed copies (94 / 124 KiB of LDS: one workgroup = one wave per SIMD per
//   CU, built with 2-byte scattered stores) and ran the vision shape in 317 + 415 us against the forward's 108.
// ---------------------------------------------------------------------------------------------------------
template <int NKT, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attention_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                               const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                               bf16* __restrict__ dqkv, int S, int H, float scale,
                                                               const uint8_t* __restrict__ key_mask) {
  constexpr int SP = NKT * 32;
  constexpr int VS = SP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);                                  // [SP][72]
  bf16* Vs = reinterpret_cast<bf16*>(smem + SP * kKStride * 2);              // [SP][72]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * kDh;
  const size_t row_stride = (size_t)3 * D;
  const bf16* base = qkv + (size_t)b * S * row_stride + h * kDh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int r = tid >> 3; r < SP; r += (int)(blockDim.x >> 3)) {
    const int c = tid & 7;
    bf16x8 kv, vv;
#pragma unroll
    for (int j = 0; j < 8; ++j) { kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; }
    if (r < S) {
      kv = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + D + c * 8);
      vv = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + 2 * D + c * 8);
    }
    *reinterpret_cast<bf16x8*>(Ks + r * kKStride + c * 8) = kv;
    *reinterpret_cast<bf16x8*>(Vs + r * kKStride + c * 8) = vv;
  }
  __syncthreads();
  const int l31 = lane & 31, half = lane >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const int nqt = (S + 31) >> 5;
  const bf16* ktr = Ks + tr_off(lane, kKStride);  // K^T fragments by transpose reads of the row-major K image
  for (int qt = wave; qt < nqt; qt += (int)(blockDim.x >> 6)) {
    const int q = qt * 32 + l31;
    const int qc = q < S ? q : S - 1;
    bf16x8 qf[4], dof[4];
    float dq_part = 0.f;
    const bf16* orow = O + ((size_t)b * S + qc) * D + h * kDh + 8 * half;
    const bf16* dorow = dO + ((size_t)b * S + qc) * D + h * kDh + 8 * half;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * row_stride + 16 * t + 8 * half);
      dof[t] = *reinterpret_cast<const bf16x8*>(dorow + 16 * t);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + 16 * t);
#pragma unroll
      for (int j = 0; j < 8; ++j) dq_part += (float)dof[t][j] * (float)of[j];
    }
    const float Dq = dq_part + __shfl_xor(dq_part, 32);
    const float L2 = lse[((size_t)b * H + h) * S + qc];
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int kt_hi = CAUSAL ? (qt + 1 < NKT ? qt + 1 : NKT) : NKT;
#pragma unroll 1
    for (int kt = 0; kt < kt_hi; ++kt) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
      const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
      const bf16* vrow = Vs + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + 16 * t), qf[t], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(vrow + 16 * t), dof[t], dp, 0, 0, 0);
      }
      uint32_t pk[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const int key = kt * 32 + 8 * g + 4 * half + j;
          const bool ok = key < S && (!CAUSAL || key <= qc) && (key_mask == nullptr || key_mask[(size_t)b * S + key] != 0);
          const float pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2) : 0.f;
          e[j] = pr * (dp[r] - Dq);
        }
        bf16x2 p0, p1;
        p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
        pk[2 * g] = __builtin_bit_cast(uint32_t, p0);
        pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int key0 = kt * 32 + 16 * jj + 4 * half;
        u32x4 pw;
        pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          // slots 0-3 = channel nt*32+l31 of keys key0..key0+3, slots 4-7 = keys key0+8..key0+11 (key0 includes 4*half, as tr_off does)
          const bf16* kp = ktr + (kt * 32 + 16 * jj) * kKStride + nt * 32;
          const uint2 v0 = lds_tr_b64(kp);
          const uint2 v1 = lds_tr_b64(kp + 8 * kKStride);
          u32x4 vw;
          vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), dsf, acc[nt], 0, 0, 0);
        }
      }
    }
    if (q < S) {
      bf16* dst = dqkv + ((size_t)b * S + q) * row_stride + h * kDh;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = acc[nt][4 * g + j] * scale;
          store4(dst + nt * 32 + 8 * g + 4 * half, o);
        }
    }
  }
}

template <int NKT, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attention_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                                const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                                bf16* __restrict__ dqkv, int S, int H, float scale,
                                                                const uint8_t* __restrict__ key_mask) {
  constexpr int SP = NKT * 32;
  constexpr int VS = SP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Qs = reinterpret_cast<bf16*>(smem);                                           // [SP][72]
  bf16* dOs = reinterpret_cast<bf16*>(smem + SP * kKStride * 2);                      // [SP][72]
  float* L2s = reinterpret_cast<float*>(smem + 2 * SP * kKStride * 2);                // [SP]
  float* Dqs = L2s + SP;                                                                   // [SP]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * kDh;
  const size_t row_stride = (size_t)3 * D;
  const bf16* base = qkv + (size_t)b * S * row_stride + h * kDh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int k = tid; k < SP; k += (int)blockDim.x) { L2s[k] = INFINITY; Dqs[k] = 0.f; }
  __syncthreads();
  for (int r = tid >> 3; r < SP; r += (int)(blockDim.x >> 3)) {
    const int c = tid & 7;
    bf16x8 qv, dv, ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) { qv[j] = (bf16)0.f; dv[j] = (bf16)0.f; ov[j] = (bf16)0.f; }
    if (r < S) {
      qv = *reinterpret_cast<const bf16x8*>(base + (size_t)r * row_stride + c * 8);
      dv = *reinterpret_cast<const bf16x8*>(dO + ((size_t)b * S + r) * D + h * kDh + c * 8);
      ov = *reinterpret_cast<const bf16x8*>(O + ((size_t)b * S + r) * D + h * kDh + c * 8);
    }
    *reinterpret_cast<bf16x8*>(Qs + r * kKStride + c * 8) = qv;
    *reinterpret_cast<bf16x8*>(dOs + r * kKStride + c * 8) = dv;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) part += (float)dv[j] * (float)ov[j];
    // the 8 threads of a row are 8 consecutive lanes: sum their partial dot products
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    if (c == 0 && r < S) { Dqs[r] = part; L2s[r] = lse[((size_t)b * H + h) * S + r]; }
  }
  __syncthreads();
  const int l31 = lane & 31, half = lane >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const int nkt = (S + 31) >> 5;
  const bf16* qtr = Qs + tr_off(lane, kKStride);   // Q^T / dO^T fragments by transpose reads of the row-major images
  const bf16* dotr = dOs + tr_off(lane, kKStride);
  for (int kt = wave; kt < nkt; kt += (int)(blockDim.x >> 6)) {
    const int key = kt * 32 + l31;
    const int kc = key < S ? key : S - 1;
    const bool key_live = key < S && (key_mask == nullptr || key_mask[(size_t)b * S + kc] != 0);  // a masked key has P = 0: dK = dV = 0
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)kc * row_stride + D + 16 * t + 8 * half);
      vf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)kc * row_stride + 2 * D + 16 * t + 8 * half);
    }
    f32x16 dv_acc[2], dk_acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv_acc[nt][r] = 0.f; dk_acc[nt][r] = 0.f; }
    const int qt_lo = CAUSAL ? kt : 0;  // queries before the key tile never see it
#pragma unroll 1
    for (int qt = qt_lo; qt < NKT; ++qt) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
      const bf16* qrow = Qs + (qt * 32 + l31) * kKStride + 8 * half;
      const bf16* drow = dOs + (qt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qrow + 16 * t), kf[t], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(drow + 16 * t), vf[t], dp, 0, 0, 0);
      }
      uint32_t pk[8], dk[8];
      // (r06) the queries' log-sum-exp / D values as 16-byte LDS reads one group ahead, exp2 unconditional, the key test a select afterwards: the
      // per-element `ok ? exp2(.. - L2s[q]) : 0` compiled to an exec-masked branch with a 4-byte LDS read and lgkmcnt(0) inside, per element
      f32x4 Lq = *reinterpret_cast<const f32x4*>(L2s + qt * 32 + 4 * half), Dq = *reinterpret_cast<const f32x4*>(Dqs + qt * 32 + 4 * half);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 Lc = Lq, Dc = Dq;
        if (g < 3) {
          Lq = *reinterpret_cast<const f32x4*>(L2s + qt * 32 + 8 * (g + 1) + 4 * half);
          Dq = *reinterpret_cast<const f32x4*>(Dqs + qt * 32 + 8 * (g + 1) + 4 * half);
        }
        float e[4], f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const int q = qt * 32 + 8 * g + 4 * half + j;
          const bool ok = key_live && (!CAUSAL || key <= q);
          float pr, dd;  // (L2 = +inf for padded queries -> 0)
          if constexpr (CAUSAL) {  // (per-element branch form: faster where half a tile is masked, see the single-pass kernel)
            pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2s[q]) : 0.f;
            dd = Dqs[q];
          } else {
            pr = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -Lc[j])) : 0.f;
            dd = Dc[j];
          }
          e[j] = pr;
          f[j] = pr * (dp[r] - dd);
        }
        bf16x2 p0, p1, d0, d1;
        p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
        d0[0] = (bf16)f[0]; d0[1] = (bf16)f[1]; d1[0] = (bf16)f[2]; d1[1] = (bf16)f[3];
        pk[2 * g] = __builtin_bit_cast(uint32_t, p0); pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
        dk[2 * g] = __builtin_bit_cast(uint32_t, d0); dk[2 * g + 1] = __builtin_bit_cast(uint32_t, d1);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int q0 = qt * 32 + 16 * jj + 4 * half;
        u32x4 pw, dw;
        pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
        dw[0] = dk[4 * jj + 0]; dw[1] = dk[4 * jj + 1]; dw[2] = dk[4 * jj + 2]; dw[3] = dk[4 * jj + 3];
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), dsf = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int toff = (qt * 32 + 16 * jj) * kKStride + nt * 32;  // queries q0..q0+3 / q0+8..q0+11 of channel nt*32+l31
          const uint2 a0 = lds_tr_b64(dotr + toff), a1 = lds_tr_b64(dotr + toff + 8 * kKStride);
          const uint2 b0 = lds_tr_b64(qtr + toff), b1 = lds_tr_b64(qtr + toff + 8 * kKStride);
          u32x4 aw, bw;
          aw[0] = a0.x; aw[1] = a0.y; aw[2] = a1.x; aw[3] = a1.y;
          bw[0] = b0.x; bw[1] = b0.y; bw[2] = b1.x; bw[3] = b1.y;
          dv_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), pf, dv_acc[nt], 0, 0, 0);
          dk_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw), dsf, dk_acc[nt], 0, 0, 0);
        }
      }
    }
    if (key < S) {
      bf16* dst = dqkv + ((size_t)b * S + key) * row_stride + h * kDh;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 ok_, ov_;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ok_[j] = dk_acc[nt][4 * g + j] * scale; ov_[j] = dv_acc[nt][4 * g + j]; }
          store4(dst + D + nt * 32 + 8 * g + 4 * half, ok_);
          store4(dst + 2 * D + nt * 32 + 8 * g + 4 * half, ov_);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// FUSED attention backward (r04; S <= 256).  The two kernels above each stage half of a head's operands, fetch the other half as
// fragment-shaped global loads (32 rows x 32 B per wave-instruction) and store their results 8 bytes per lane: ~1 GB of HBM traffic per
// ViT-B/16 launch pair for 618 MB of operands, at poor coalescing.  Here ONE persistent workgroup per CU walks the (batch, head) items:
//   * the four row images Q, K, V, dO of the head (+ log-sum-exp, D = sum dO.O, key mask) are staged ONCE, as whole 128-byte row segments;
//   * wave w runs the dK / dV role for key tile w (lane = key: S^T, dP^T, dV += P^T dO, dK += dS^T Q) and then the dQ role for query tile w
//     (lane = query: S, dP, dS, dQ += dS K), both out of the same images -- every MFMA operand is an LDS read (row fragments by ds_read_b128,
//     transposed ones by ds_read_b64_tr_b16), nothing fragment-shaped touches HBM.  (Both roles at once on 16 waves would cap a wave at 128
//     registers: the dK / dV role needs 162 and spilled 100 of them.)
//   * the results go back through the dead images (dQ over Q, dK over K, dV over V) and leave as whole rows [dq | dk | dv].
// Same arithmetic per element as the two-kernel form (the products, their operand roundings and the summation order inside a tile pair are
// unchanged), so the gradients are bit-identical to it (tests/test_gpu_backward_kernels.py).
// ---------------------------------------------------------------------------------------------------------
template <int NKT, bool CAUSAL, int NWH>
__global__ __launch_bounds__(NWH * 64) void attention_bwd_fused_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                                          const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                                          bf16* __restrict__ dqkv, int S, int H, int BH, float scale,
                                                                          const uint8_t* __restrict__ key_mask) {
  static_assert(NKT <= NWH, "one tile per wave and role");
  constexpr int SP = NKT * 32;
  constexpr int NT = NWH * 64;  // threads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Qs = reinterpret_cast<bf16*>(smem);
  bf16* Ks = Qs + SP * kKStride;
  bf16* Vs = Ks + SP * kKStride;
  bf16* dOs = Vs + SP * kKStride;
  float* L2s = reinterpret_cast<float*>(dOs + SP * kKStride);
  float* Dqs = L2s + SP;
  uint8_t* Mk = reinterpret_cast<uint8_t*>(Dqs + SP);  // 1 = key may be attended
  const int D = H * kDh;
  const size_t row_stride = (size_t)3 * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const int nt_s = (S + 31) >> 5;  // live tiles (<= NKT)

  for (int item = blockIdx.x; item < BH; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const bf16* base = qkv + (size_t)b * S * row_stride + h * kDh;
    // ---- stage: 8 threads per row, 16 bytes each of q, k, v, dO, O
    for (int r = tid >> 3; r < SP; r += NT >> 3) {
      const int c = tid & 7;
      bf16x8 qv, kv, vv, dv, ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) { qv[j] = (bf16)0.f; kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; dv[j] = (bf16)0.f; ov[j] = (bf16)0.f; }
      if (r < S) {
        const bf16* row = base + (size_t)r * row_stride + c * 8;
        qv = *reinterpret_cast<const bf16x8*>(row);
        kv = *reinterpret_cast<const bf16x8*>(row + D);
        vv = *reinterpret_cast<const bf16x8*>(row + 2 * D);
        dv = *reinterpret_cast<const bf16x8*>(dO + ((size_t)b * S + r) * D + h * kDh + c * 8);
        ov = *reinterpret_cast<const bf16x8*>(O + ((size_t)b * S + r) * D + h * kDh + c * 8);
      }
      *reinterpret_cast<bf16x8*>(Qs + r * kKStride + c * 8) = qv;
      *reinterpret_cast<bf16x8*>(Ks + r * kKStride + c * 8) = kv;
      *reinterpret_cast<bf16x8*>(Vs + r * kKStride + c * 8) = vv;
      *reinterpret_cast<bf16x8*>(dOs + r * kKStride + c * 8) = dv;
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) part += (float)dv[j] * (float)ov[j];
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      if (c == 0) {
        Dqs[r] = r < S ? part : 0.f;
        L2s[r] = r < S ? lse[((size_t)b * H + h) * S + r] : INFINITY;  // +inf: a padded query has P = exp2(-inf) = 0
        Mk[r] = (r < S && (key_mask == nullptr || key_mask[(size_t)b * S + r] != 0)) ? 1 : 0;
      }
    }
    __syncthreads();

    f32x16 dq_acc[2], dv_acc[2], dk_acc[2];
    const int tile = wave;
    if (tile < nt_s) {
      {
        // ---------------- dQ role: lane = query ----------------
        const int qt = tile;
        const int q = qt * 32 + l31;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          qf[t] = *reinterpret_cast<const bf16x8*>(Qs + q * kKStride + 16 * t + 8 * half);
          dof[t] = *reinterpret_cast<const bf16x8*>(dOs + q * kKStride + 16 * t + 8 * half);
        }
        const float Dq = Dqs[q], L2 = L2s[q];
        const bf16* ktr = Ks + tr_off(lane, kKStride);
        const int kt_hi = CAUSAL ? (qt + 1 < nt_s ? qt + 1 : nt_s) : nt_s;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dq_acc[nt][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < kt_hi; ++kt) {
          f32x16 st, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
          const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
          const bf16* vrow = Vs + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + 16 * t), qf[t], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(vrow + 16 * t), dof[t], dp, 0, 0, 0);
          }
          uint32_t pk[8];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t mk4 = *reinterpret_cast<const uint32_t*>(Mk + kt * 32 + 8 * g + 4 * half);  // the 4 keys of this group
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * g + j;
              const int key = kt * 32 + 8 * g + 4 * half + j;
              const bool ok = ((mk4 >> (8 * j)) & 1u) != 0 && (!CAUSAL || key <= q);
              const float pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2) : 0.f;
              e[j] = pr * (dp[r] - Dq);
            }
            bf16x2 p0, p1;
            p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
            pk[2 * g] = __builtin_bit_cast(uint32_t, p0);
            pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            u32x4 pw;
            pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const bf16* kp = ktr + (kt * 32 + 16 * jj) * kKStride + nt * 32;
              const uint2 v0 = lds_tr_b64(kp);
              const uint2 v1 = lds_tr_b64(kp + 8 * kKStride);
              u32x4 vw;
              vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
              dq_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), dsf, dq_acc[nt], 0, 0, 0);
            }
          }
        }
      }
      {
        // ---------------- dK / dV role: lane = key ----------------
        const int kt = tile;
        const int key = kt * 32 + l31;
        const bool key_live = Mk[key] != 0;  // a masked / padded key has P = 0: dK = dV = 0
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          kf[t] = *reinterpret_cast<const bf16x8*>(Ks + key * kKStride + 16 * t + 8 * half);
          vf[t] = *reinterpret_cast<const bf16x8*>(Vs + key * kKStride + 16 * t + 8 * half);
        }
        const bf16* qtr = Qs + tr_off(lane, kKStride);
        const bf16* dotr = dOs + tr_off(lane, kKStride);
        const int qt_lo = CAUSAL ? kt : 0;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { dv_acc[nt][r] = 0.f; dk_acc[nt][r] = 0.f; }
#pragma unroll 1
        for (int qt = qt_lo; qt < nt_s; ++qt) {
          f32x16 st, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
          const bf16* qrow = Qs + (qt * 32 + l31) * kKStride + 8 * half;
          const bf16* drow = dOs + (qt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qrow + 16 * t), kf[t], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(drow + 16 * t), vf[t], dp, 0, 0, 0);
          }
          uint32_t pk[8], dk[8];
          f32x4 Lq = *reinterpret_cast<const f32x4*>(L2s + qt * 32 + 4 * half), Dq = *reinterpret_cast<const f32x4*>(Dqs + qt * 32 + 4 * half);  // (r06: see the dK / dV kernel above)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 Lc = Lq, Dc = Dq;
            if (g < 3) {
              Lq = *reinterpret_cast<const f32x4*>(L2s + qt * 32 + 8 * (g + 1) + 4 * half);
              Dq = *reinterpret_cast<const f32x4*>(Dqs + qt * 32 + 8 * (g + 1) + 4 * half);
            }
            float e[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * g + j;
              const int q = qt * 32 + 8 * g + 4 * half + j;
              const bool ok = key_live && (!CAUSAL || key <= q);
              float pr, dd;
              if constexpr (CAUSAL) {  // (the diagonal tiles mask half their elements: the per-element branch form measured faster -- S = 77: 55.4 vs 63 us)
                pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2s[q]) : 0.f;
                dd = Dqs[q];
              } else {
                pr = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -Lc[j])) : 0.f;
                dd = Dc[j];
              }
              e[j] = pr;
              f[j] = pr * (dp[r] - dd);
            }
            bf16x2 p0, p1, d0, d1;
            p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
            d0[0] = (bf16)f[0]; d0[1] = (bf16)f[1]; d1[0] = (bf16)f[2]; d1[1] = (bf16)f[3];
            pk[2 * g] = __builtin_bit_cast(uint32_t, p0); pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
            dk[2 * g] = __builtin_bit_cast(uint32_t, d0); dk[2 * g + 1] = __builtin_bit_cast(uint32_t, d1);
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            u32x4 pw, dw;
            pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
            dw[0] = dk[4 * jj + 0]; dw[1] = dk[4 * jj + 1]; dw[2] = dk[4 * jj + 2]; dw[3] = dk[4 * jj + 3];
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), dsf = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int toff = (qt * 32 + 16 * jj) * kKStride + nt * 32;
              const uint2 a0 = lds_tr_b64(dotr + toff), a1 = lds_tr_b64(dotr + toff + 8 * kKStride);
              const uint2 b0 = lds_tr_b64(qtr + toff), b1 = lds_tr_b64(qtr + toff + 8 * kKStride);
              u32x4 aw, bw;
              aw[0] = a0.x; aw[1] = a0.y; aw[2] = a1.x; aw[3] = a1.y;
              bw[0] = b0.x; bw[1] = b0.y; bw[2] = b1.x; bw[3] = b1.y;
              dv_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), pf, dv_acc[nt], 0, 0, 0);
              dk_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw), dsf, dk_acc[nt], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();  // every role is done reading the images: they become the output staging area
    if (tile < nt_s) {
      const int row = tile * 32 + l31;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = nt * 32 + 8 * g + 4 * half;
          f32x4 a, c, e;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = dq_acc[nt][4 * g + j] * scale; c[j] = dk_acc[nt][4 * g + j] * scale; e[j] = dv_acc[nt][4 * g + j]; }
          store4(Qs + row * kKStride + col, a);   // dQ over Q
          store4(Ks + row * kKStride + col, c);   // dK over K
          store4(Vs + row * kKStride + col, e);   // dV over V
        }
    }
    __syncthreads();
    // ---- whole rows out: [dq | dk | dv], 8 threads x 16 bytes per 128-byte segment
    bf16* obase = dqkv + (size_t)b * S * row_stride + h * kDh;
    for (int r = tid >> 3; r < S; r += NT >> 3) {
      const int c = tid & 7;
      bf16* orow = obase + (size_t)r * row_stride + c * 8;
      *reinterpret_cast<bf16x8*>(orow) = *reinterpret_cast<const bf16x8*>(Qs + r * kKStride + c * 8);
      *reinterpret_cast<bf16x8*>(orow + D) = *reinterpret_cast<const bf16x8*>(Ks + r * kKStride + c * 8);
      *reinterpret_cast<bf16x8*>(orow + 2 * D) = *reinterpret_cast<const bf16x8*>(Vs + r * kKStride + c * 8);
    }
    __syncthreads();  // the next item's staging overwrites the images
  }
}

// ---------------------------------------------------------------------------------------------------------
// SINGLE-PASS attention backward (r04, S <= 256): every (query tile, key tile) pair is visited ONCE.  The fused kernel above still computes
// S, P, dP and dS of a pair twice (once per role: 28 MFMAs and 32 exponentials per pair); here the wave that owns key tile kt computes them once --
// S^T and dP^T (lane = key), dV_kt += P^T dO, dK_kt += dS^T Q -- and HANDS dS OVER, as a 32 x 32 bf16 tile in an LDS mailbox, to the wave that owns
// query tile qt, which adds dQ_qt += dS K_kt to its own registers (20 MFMAs and 16 exponentials per pair).  Skewed schedule: in step s wave w produces
// pair (qt = (w + s) mod T, kt = w) and, behind ONE workgroup barrier, consumes pair (qt = w, kt = (w - s) mod T) -- T = live tiles; every wave
// produces and consumes exactly one pair per step, no two waves touch the same accumulator, the summation order of every dQ / dK / dV element is
// fixed (deterministic, no atomics).  Mailboxes are double-buffered over the step parity, so one barrier per step separates writer and reader.
//   LDS: Q, K, dO row images [SP][72] (V is only ever needed by its key tile's owner: fragments straight from global), log-sum-exp / D / key-mask rows,
//   2 x T mailboxes of [32 keys][40] bf16 (the consumer reads dS^T fragments by ds_read_b64_tr_b16, exactly like the K^T fragments).  139 KiB at S = 197:
//   one 8-wave workgroup per CU.  Results leave through the dead images as whole rows, like the fused kernel.
// Same products, operand roundings and per-pair arithmetic as the other two forms; dQ sums its key tiles in another order (fp32 rounding only).
// ---------------------------------------------------------------------------------------------------------
template <int NKT, bool CAUSAL>
__global__ __launch_bounds__(512) void attention_bwd_sp_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O, const bf16* __restrict__ dO,
                                                               const float* __restrict__ lse, bf16* __restrict__ dqkv, int S, int H, int BH,
                                                               float scale, const uint8_t* __restrict__ key_mask) {
  static_assert(NKT <= 8, "one tile per wave");
  constexpr int SP = NKT * 32, NT = 512, TS = 40;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Qs = reinterpret_cast<bf16*>(smem);
  bf16* Ks = Qs + SP * kKStride;
  bf16* dOs = Ks + SP * kKStride;
  float* L2s = reinterpret_cast<float*>(dOs + SP * kKStride);
  float* Dqs = L2s + SP;
  uint8_t* Mk = reinterpret_cast<uint8_t*>(Dqs + SP);
  bf16* mail = reinterpret_cast<bf16*>(smem + 3 * SP * kKStride * 2 + 2 * SP * 4 + ((SP + 15) & ~15));  // [2][NKT][32][TS]
  const int D = H * kDh;
  const size_t row_stride = (size_t)3 * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const float c2 = scale * 1.4426950408889634f;
  const int T = (S + 31) >> 5;  // live tiles (<= NKT)

  for (int item = blockIdx.x; item < BH; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const bf16* base = qkv + (size_t)b * S * row_stride + h * kDh;
    // Staging with EVERY global load of the item in flight before the first one is consumed (r05): the row loop used to be a rolled loop of
    // ceil(SP / 64) trips, each a round trip for its four rows' chunks and then a second, dependent one for the log-sum-exp (and a third for the key
    // mask) -- seven or more serial round trips to memory per (batch, head) item, ~10 of its 25 us at S = 197; the K / V fragments of this wave's key
    // tile came after that.  Now: this wave's K / V fragments, then all row chunks, log-sum-exp values and mask bytes, then the LDS image.  The loads
    // are UNCONDITIONAL on clamped addresses (rows / keys past S read row S - 1 and are zeroed afterwards): a load inside a divergent branch makes the
    // compiler drain the whole queue (s_waitcnt vmcnt(0)) where the branch joins.
    const int tile = wave;
    const bool live = tile < T;
    // this wave's key tile: K and V fragments (lane = key), V straight from global
    bf16x8 kf[4], vf[4];
    {
      const int key = tile * 32 + l31;
      const bool in = live && key < S;
      const bf16* krow = base + (size_t)(key < S ? key : S - 1) * row_stride + 8 * half;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        kf[t] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * t);
        vf[t] = *reinterpret_cast<const bf16x8*>(krow + 2 * D + 16 * t);
      }
      constexpr int NI = (SP + (NT >> 3) - 1) / (NT >> 3);
      const int c = tid & 7;
      bf16x8 qv[NI], kv[NI], dv[NI], ov[NI];
      float lv[NI];
      uint8_t mv[NI];
      const uint8_t* mbase = key_mask != nullptr ? key_mask + (size_t)b * S : reinterpret_cast<const uint8_t*>(lse);  // (no mask: any readable byte)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = (tid >> 3) + i * (NT >> 3);
        const int rr = r < S ? r : S - 1;
        const bf16* row = base + (size_t)rr * row_stride + c * 8;
        qv[i] = *reinterpret_cast<const bf16x8*>(row);
        kv[i] = *reinterpret_cast<const bf16x8*>(row + D);
        dv[i] = *reinterpret_cast<const bf16x8*>(dO + ((size_t)b * S + rr) * D + h * kDh + c * 8);
        ov[i] = *reinterpret_cast<const bf16x8*>(O + ((size_t)b * S + rr) * D + h * kDh + c * 8);
        lv[i] = lse[((size_t)b * H + h) * S + rr];
        mv[i] = mbase[rr];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          kf[t][j] = in ? kf[t][j] : (bf16)0.f;
          vf[t][j] = in ? vf[t][j] : (bf16)0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = (tid >> 3) + i * (NT >> 3);
        const bool rin = r < S;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          qv[i][j] = rin ? qv[i][j] : (bf16)0.f;
          kv[i][j] = rin ? kv[i][j] : (bf16)0.f;
          dv[i][j] = rin ? dv[i][j] : (bf16)0.f;
        }
        if (r < SP) {
          *reinterpret_cast<bf16x8*>(Qs + r * kKStride + c * 8) = qv[i];
          *reinterpret_cast<bf16x8*>(Ks + r * kKStride + c * 8) = kv[i];
          *reinterpret_cast<bf16x8*>(dOs + r * kKStride + c * 8) = dv[i];
        }
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)dv[i][j] * (float)ov[i][j];
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        if (c == 0 && r < SP) {
          Dqs[r] = rin ? part : 0.f;
          L2s[r] = rin ? lv[i] : INFINITY;
          Mk[r] = (rin && (key_mask == nullptr || mv[i] != 0)) ? 1 : 0;
        }
      }
    }
    __syncthreads();
    const int key = tile * 32 + l31;
    const bool key_live = live && Mk[key] != 0;
    f32x16 dq_acc[2], dv_acc[2], dk_acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dq_acc[nt][r] = 0.f; dv_acc[nt][r] = 0.f; dk_acc[nt][r] = 0.f; }
    const bf16* qtr = Qs + tr_off(lane, kKStride);
    const bf16* dotr = dOs + tr_off(lane, kKStride);
    const bf16* ktr = Ks + tr_off(lane, kKStride);
    const int trm = tr_off(lane, TS);

#pragma unroll 1
    for (int s_ = 0; s_ < T; ++s_) {
      bf16* mbuf = mail + (size_t)(s_ & 1) * NKT * 32 * TS;
      // ---------------- produce: pair (qt, kt = tile), lane = key ----------------
      if (live) {
        int qt = tile + s_;
        qt = qt >= T ? qt - T : qt;
        if (!CAUSAL || tile <= qt) {
          f32x16 st, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
          const bf16* qrow = Qs + (qt * 32 + l31) * kKStride + 8 * half;
          const bf16* drow = dOs + (qt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qrow + 16 * t), kf[t], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(drow + 16 * t), vf[t], dp, 0, 0, 0);
          }
          uint32_t pk[8], dk[8];
          bf16* mrow = mbuf + (size_t)qt * 32 * TS + l31 * TS + 4 * half;
          // the 16 queries' log-sum-exp and D values: eight 16-byte LDS reads up front (r06).  The per-element form `ok ? exp2(st c2 - L2s[q]) : 0` compiled
          // to an exec-masked branch per element with a 4-byte LDS read and `s_waitcnt lgkmcnt(0)` INSIDE it -- 32 exposed LDS round trips per pair and wave
          // (read off the ISA).  exp2 runs unconditionally (rows past S hold +inf -> 0; masked keys are selected away afterwards): straight-line code.
          // (two groups' worth in flight: all eight at once needs 32 registers the kernel does not have -- 32 B of scratch when tried)
          f32x4 Lq = *reinterpret_cast<const f32x4*>(L2s + qt * 32 + 4 * half), Dq = *reinterpret_cast<const f32x4*>(Dqs + qt * 32 + 4 * half);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 Lc = Lq, Dc = Dq;
            if (g < 3) {
              Lq = *reinterpret_cast<const f32x4*>(L2s + qt * 32 + 8 * (g + 1) + 4 * half);
              Dq = *reinterpret_cast<const f32x4*>(Dqs + qt * 32 + 8 * (g + 1) + 4 * half);
            }
            float e[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * g + j;
              const int q = qt * 32 + 8 * g + 4 * half + j;
              const bool ok = key_live && (!CAUSAL || key <= q);
              float pr, dd;
              if constexpr (CAUSAL) {  // (the diagonal tiles mask half their elements: the per-element branch form measured faster -- S = 77: 55.4 vs 63 us)
                pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2s[q]) : 0.f;
                dd = Dqs[q];
              } else {
                pr = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -Lc[j])) : 0.f;
                dd = Dc[j];
              }
              e[j] = pr;
              f[j] = pr * (dp[r] - dd);
            }
            bf16x2 p0, p1, d0, d1;
            p0[0] = (bf16)e[0]; p0[1] = (bf16)e[1]; p1[0] = (bf16)e[2]; p1[1] = (bf16)e[3];
            d0[0] = (bf16)f[0]; d0[1] = (bf16)f[1]; d1[0] = (bf16)f[2]; d1[1] = (bf16)f[3];
            pk[2 * g] = __builtin_bit_cast(uint32_t, p0); pk[2 * g + 1] = __builtin_bit_cast(uint32_t, p1);
            dk[2 * g] = __builtin_bit_cast(uint32_t, d0); dk[2 * g + 1] = __builtin_bit_cast(uint32_t, d1);
            // dS[q = 8g + 4half + 0..3][key] -> mailbox row `key`, 4 consecutive query columns (8 bytes)
            *reinterpret_cast<uint2*>(mrow + 8 * g) = make_uint2(dk[2 * g], dk[2 * g + 1]);
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            u32x4 pw, dw;
            pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
            dw[0] = dk[4 * jj + 0]; dw[1] = dk[4 * jj + 1]; dw[2] = dk[4 * jj + 2]; dw[3] = dk[4 * jj + 3];
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), dsf = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int toff = (qt * 32 + 16 * jj) * kKStride + nt * 32;
              const uint2 a0 = lds_tr_b64(dotr + toff), a1 = lds_tr_b64(dotr + toff + 8 * kKStride);
              const uint2 b0 = lds_tr_b64(qtr + toff), b1 = lds_tr_b64(qtr + toff + 8 * kKStride);
              u32x4 aw, bw;
              aw[0] = a0.x; aw[1] = a0.y; aw[2] = a1.x; aw[3] = a1.y;
              bw[0] = b0.x; bw[1] = b0.y; bw[2] = b1.x; bw[3] = b1.y;
              dv_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), pf, dv_acc[nt], 0, 0, 0);
              dk_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw), dsf, dk_acc[nt], 0, 0, 0);
            }
          }
        }
      }
      __syncthreads();  // the step's mailboxes are written (and, double-buffered, nobody still reads the ones of two steps ago)
      // ---------------- consume: pair (qt = tile, kt), lane = query ----------------
      if (live) {
        int kt = tile - s_;
        kt = kt < 0 ? kt + T : kt;
        if (!CAUSAL || kt <= tile) {
          const bf16* mt = mbuf + (size_t)tile * 32 * TS + trm;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const uint2 d0 = lds_tr_b64(mt + (16 * jj) * TS), d1 = lds_tr_b64(mt + (16 * jj + 8) * TS);
            u32x4 dw;
            dw[0] = d0.x; dw[1] = d0.y; dw[2] = d1.x; dw[3] = d1.y;
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, dw);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const bf16* kp = ktr + (kt * 32 + 16 * jj) * kKStride + nt * 32;
              const uint2 v0 = lds_tr_b64(kp), v1 = lds_tr_b64(kp + 8 * kKStride);
              u32x4 vw;
              vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
              dq_acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), dsf, dq_acc[nt], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();  // every wave is done reading the images: they become the output staging area
    if (live) {
      const int row = tile * 32 + l31;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = nt * 32 + 8 * g + 4 * half;
          f32x4 a, c, e;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = dq_acc[nt][4 * g + j] * scale; c[j] = dk_acc[nt][4 * g + j] * scale; e[j] = dv_acc[nt][4 * g + j]; }
          store4(Qs + row * kKStride + col, a);    // dQ over Q
          store4(Ks + row * kKStride + col, c);    // dK over K
          store4(dOs + row * kKStride + col, e);   // dV over dO
        }
    }
    __syncthreads();
    bf16* obase = dqkv + (size_t)b * S * row_stride + h * kDh;
    for (int r = tid >> 3; r < S; r += NT >> 3) {
      const int c = tid & 7;
      bf16* orow = obase + (size_t)r * row_stride + c * 8;
      *reinterpret_cast<bf16x8*>(orow) = *reinterpret_cast<const bf16x8*>(Qs + r * kKStride + c * 8);
      *reinterpret_cast<bf16x8*>(orow + D) = *reinterpret_cast<const bf16x8*>(Ks + r * kKStride + c * 8);
      *reinterpret_cast<bf16x8*>(orow + 2 * D) = *reinterpret_cast<const bf16x8*>(dOs + r * kKStride + c * 8);
    }
    __syncthreads();  // the next item's staging overwrites the images
  }
}

template <int NKT>
static int launch_attn_bwd_sp(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H, int causal,
                              float scale, hipStream_t st, const uint8_t* key_mask) {
  constexpr int SP = NKT * 32;
  constexpr int smem = 3 * SP * kKStride * 2 + 2 * SP * 4 + ((SP + 15) & ~15) + 2 * NKT * 32 * 40 * 2;
  auto kc = attention_bwd_sp_kernel<NKT, true>;
  auto kn = attention_bwd_sp_kernel<NKT, false>;
  static unsigned long long mc = 0, mn = 0;
  if (int rc_attr = opt_in_lds((const void*)kc, smem, mc)) return rc_attr;
  if (int rc_attr = opt_in_lds((const void*)kn, smem, mn)) return rc_attr;
  const int per_cu = (160 * 1024) / smem >= 2 ? 2 : 1;
  const int slots = per_cu * stream_cus(st);
  const int BH = B * H;
  const int grid = BH < slots ? BH : slots;
  if (causal) hipLaunchKernelGGL(kc, dim3(grid), dim3(512), smem, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, BH, scale, key_mask);
  else hipLaunchKernelGGL(kn, dim3(grid), dim3(512), smem, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, BH, scale, key_mask);
  return launch_status("attention_bwd_sp");
}

template <int NKT, int NWH>
static int launch_attn_bwd_fused(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H, int causal,
                                 float scale, hipStream_t st, const uint8_t* key_mask) {
  constexpr int SP = NKT * 32;
  constexpr int smem = 4 * SP * kKStride * 2 + 2 * SP * 4 + SP;
  auto kc = attention_bwd_fused_kernel<NKT, true, NWH>;
  auto kn = attention_bwd_fused_kernel<NKT, false, NWH>;
  static unsigned long long mc = 0, mn = 0;
  if (int rc_attr = opt_in_lds((const void*)kc, smem, mc)) return rc_attr;
  if (int rc_attr = opt_in_lds((const void*)kn, smem, mn)) return rc_attr;
  const int per_cu = (160 * 1024) / smem >= 2 ? 2 : 1;
  const int slots = per_cu * stream_cus(st);
  const int BH = B * H;
  const int grid = BH < slots ? BH : slots;
  if (causal) hipLaunchKernelGGL(kc, dim3(grid), dim3(NWH * 64), smem, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, BH, scale, key_mask);
  else hipLaunchKernelGGL(kn, dim3(grid), dim3(NWH * 64), smem, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, BH, scale, key_mask);
  return launch_status("attention_bwd_fused");
}

template <int NKT>
static int launch_attn_bwd(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H, int causal,
                           float scale, hipStream_t st, const uint8_t* key_mask) {
  constexpr int SP = NKT * 32;
  // the kernels take any multiple of 64 threads (tiles are dealt round-robin to the waves): 8 waves per workgroup from 5 tiles up (ViT-B/16: 7 tiles in
  // one round, 16 waves per CU instead of 8: 365-391 us vs 398-405 us for both kernels at B = 256), 4 below (text S = 77: 67 vs 71 us); same results
  constexpr int kThreads = NKT >= 5 ? 512 : 256;
  constexpr int smem1 = 2 * SP * kKStride * 2;               // K, V rows (two workgroups per CU up to S = 256)
  constexpr int smem2 = 2 * SP * kKStride * 2 + 2 * SP * 4;  // Q, dO rows + lse / Dq
  auto k1c = attention_bwd_dq_kernel<NKT, true>;
  auto k1n = attention_bwd_dq_kernel<NKT, false>;
  auto k2c = attention_bwd_dkv_kernel<NKT, true>;
  auto k2n = attention_bwd_dkv_kernel<NKT, false>;
  static unsigned long long m1c = 0, m1n = 0, m2c = 0, m2n = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds((const void*)k1c, smem1, m1c)) return rc_attr;
  if (int rc_attr = opt_in_lds((const void*)k1n, smem1, m1n)) return rc_attr;
  if (int rc_attr = opt_in_lds((const void*)k2c, smem2, m2c)) return rc_attr;
  if (int rc_attr = opt_in_lds((const void*)k2n, smem2, m2n)) return rc_attr;
  if (causal) {
    hipLaunchKernelGGL(k1c, dim3(B * H), dim3(kThreads), smem1, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, scale, key_mask);
    hipLaunchKernelGGL(k2c, dim3(B * H), dim3(kThreads), smem2, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, scale, key_mask);
  } else {
    hipLaunchKernelGGL(k1n, dim3(B * H), dim3(kThreads), smem1, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, scale, key_mask);
    hipLaunchKernelGGL(k2n, dim3(B * H), dim3(kThreads), smem2, st, (const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, (bf16*)dqkv, S, H, scale, key_mask);
  }
  return launch_status("attention_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// Long sequences (S > 288: 384-pixel ViTs after interpolate_pos_encoding, 512-token BERT inputs): the keys no longer fit LDS at once,
// so K / V stream through it in chunks of 128 keys (row-major images, V operands by transpose reads).  A workgroup owns one (batch,
// head) and 128 queries (one 32-query tile per wave) and makes TWO passes over the chunks: pass 1 = row max and row sum (QK^T only),
// pass 2 = recompute the scores, optionally emit the NORMALISED probabilities ([B,H,S,S], what FLAVA's encoders return) and
// accumulate P.V with the final normalisation already applied — no rescaling of O, and the same arithmetic whether or not the
// probabilities are wanted.  Self-attention on the packed qkv layout of the kernels above; causal and key-padding masks; a fully
// masked row is NaN like the reference's softmax.  Forward only (no lse): training on S > 288 raises.
template <typename TP, bool CAUSAL>
__global__ __launch_bounds__(256) void attention_long_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ key_mask,
                                                             bf16* __restrict__ out, TP* __restrict__ probs, int S, int H, float c2) {
  constexpr int KC = 128;
  __shared__ __attribute__((aligned(16))) bf16 Ks[KC * kKStride];
  __shared__ __attribute__((aligned(16))) bf16 Vs[KC * kKStride];
  __shared__ float Mk[KC];  // 0 for a key that may be attended, -inf otherwise (past the end or masked)
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * kDh;
  const size_t row_stride = (size_t)3 * D;
  const bf16* base = qkv + (size_t)b * S * row_stride + h * kDh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nqt = (S + 31) >> 5;
  const int qt = blockIdx.y * 4 + wave;
  const bool active = qt < nqt;  // wave-uniform
  const int q = qt * 32 + l31;
  const int qc = q < S ? q : S - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * row_stride + 16 * t + 8 * half);
  const int nchunks = (S + KC - 1) / KC;
  // causal: chunks entirely above this workgroup's last query are never needed
  const int q_hi = (blockIdx.y * 4 + 4) * 32 - 1;
  const int nch = CAUSAL ? ((q_hi < S ? q_hi : S - 1) / KC + 1) : nchunks;

  auto stage = [&](int c, bool with_v) {
    for (int r = tid >> 3; r < KC; r += 32) {
      const int ch = tid & 7, key = c * KC + r;
      bf16x8 kv, vv;
#pragma unroll
      for (int j = 0; j < 8; ++j) { kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; }
      if (key < S) {
        kv = *reinterpret_cast<const bf16x8*>(base + (size_t)key * row_stride + D + ch * 8);
        if (with_v) vv = *reinterpret_cast<const bf16x8*>(base + (size_t)key * row_stride + 2 * D + ch * 8);
      }
      *reinterpret_cast<bf16x8*>(Ks + r * kKStride + ch * 8) = kv;
      if (with_v) *reinterpret_cast<bf16x8*>(Vs + r * kKStride + ch * 8) = vv;
      if (ch == 0) Mk[r] = (key < S && (key_mask == nullptr || key_mask[(size_t)b * S + key] != 0)) ? 0.f : -INFINITY;
    }
  };
  // scores of query q against the 32 keys of tile kt of chunk c, scaled into the log2 domain, masked keys = -inf
  auto scores = [&](int c, int kt) -> f32x16 {
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const bf16* krow = Ks + (kt * 32 + l31) * kKStride + 8 * half;
#pragma unroll
    for (int t = 0; t < 4; ++t) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + 16 * t), qf[t], st, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = st[r] * c2 + Mk[kl];
      if (CAUSAL && c * KC + kl > q) v = -INFINITY;
      st[r] = v;
    }
    return st;
  };

  // ---- pass 1: running maximum and sum of exp2 per query row
  float m = -INFINITY, l = 0.f;
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
    stage(c, false);
    __syncthreads();
    if (!active) continue;
#pragma unroll 1
    for (int kt = 0; kt < KC / 32; ++kt) {
      if (c * KC + kt * 32 >= S) break;
      const f32x16 st = scores(c, kt);
      float tmax = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m, tmax);
      if (m_new == -INFINITY) continue;  // every key so far is masked: nothing to add
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(st[r] - m_new);
      l = l * __builtin_amdgcn_exp2f(m - m_new) + ps;
      m = m_new;
    }
  }
  l += __shfl_xor(l, 32);
  // a fully masked row: the reference's softmax of all -inf is NaN; 0 * inf reproduces it in every probability and in O
  const float inv = 1.0f / l;

  // ---- pass 2: probabilities (optional) and O = P V
  f32x16 ot[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[nt][r] = 0.f;
  const bf16* vtr = Vs + tr_off(lane, kKStride);
  TP* prow = probs != nullptr ? probs + (((size_t)b * H + h) * S + qc) * S : nullptr;
  for (int c = 0; c < nchunks; ++c) {  // (causal: chunks past nch only get their zero probabilities written)
    const bool need = c < nch;
    __syncthreads();
    if (need) stage(c, true);
    __syncthreads();
    if (!active) continue;
#pragma unroll 1
    for (int kt = 0; kt < KC / 32; ++kt) {
      if (c * KC + kt * 32 >= S) break;
      f32x16 pr;
      if (need) {
        const f32x16 st = scores(c, kt);
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = (l > 0.f ? __builtin_amdgcn_exp2f(st[r] - m) : 0.f) * inv;  // l == 0: 0 * inf = NaN
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = 0.f;
      }
      if (prow != nullptr && q < S) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key = c * KC + kt * 32 + 8 * g + 4 * half;
          if (key + 3 < S) {
            store_probs4(prow + key, f32x4{pr[4 * g], pr[4 * g + 1], pr[4 * g + 2], pr[4 * g + 3]});
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (key + j < S) prow[key + j] = (TP)pr[4 * g + j];
          }
        }
      }
      if (!need) continue;
      uint32_t pk[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        bf16x2 p2;
        p2[0] = (bf16)pr[2 * g]; p2[1] = (bf16)pr[2 * g + 1];
        pk[g] = __builtin_bit_cast(uint32_t, p2);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        u32x4 pw;
        pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16* vp = vtr + (kt * 32 + 16 * jj) * kKStride + nt * 32;
          const uint2 v0 = lds_tr_b64(vp), v1 = lds_tr_b64(vp + 8 * kKStride);
          u32x4 vw;
          vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
          ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, ot[nt], 0, 0, 0);
        }
      }
    }
  }
  if (active && q < S) {
    bf16* orow = out + ((size_t)b * S + q) * D + h * kDh;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ot[nt][4 * g + j];
        store4(orow + nt * 32 + 8 * g + 4 * half, o);
      }
  }
}

static int launch_attn_long(const void* qkv, const uint8_t* key_mask, void* out, void* probs, int probs_dtype, int B, int S, int H, int causal,
                            float scale, hipStream_t st) {
  const dim3 grid(B * H, ((S + 31) / 32 + 3) / 4), block(256);
  const float c2 = scale * 1.4426950408889634f;
#define ATTN_LONG(TP, C) hipLaunchKernelGGL((attention_long_kernel<TP, C>), grid, block, 0, st, (const bf16*)qkv, key_mask, (bf16*)out, (TP*)probs, S, H, c2)
  if (probs != nullptr && probs_dtype == MMAMD_BF16) { if (causal) ATTN_LONG(bf16, true); else ATTN_LONG(bf16, false); }
  else { if (causal) ATTN_LONG(float, true); else ATTN_LONG(float, false); }
#undef ATTN_LONG
  return launch_status("attention_long");
}

static int g_attn_variant = 0;
static int g_attn_bwd_variant = 0;  // mmamd_debug_set_attn_variant(4000 two kernels | 4001 single pass | 4002 fused two-role | 4003 default): backward form only

template <int NKT, bool CAUSAL, int ABL = 0>
static int launch_attn(const void* qkv, void* out, int B, int S, int H, float scale, hipStream_t st, float* lse = nullptr, int lse_stride = 0) {
  constexpr int SP = NKT * 32;
  constexpr int smem = SP * kKStride * 2 + ((NKT <= 8 && (ABL & 1) == 0) ? SP * kKStride * 2 : 64 * (SP + 4) * 2);
  constexpr int NW = 4;  // 8 waves per workgroup needs <= 128 VGPRs to be resident twice per CU: the kernel uses ~170
  auto kern = attention_fwd_kernel<NKT, CAUSAL, ABL, NW>;
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to > 64 KiB dynamic LDS
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc_attr;
  const int BH = B * H;
  const int slots = 2 * stream_cus(st);
  const int grid = BH < slots ? BH : slots;  // 2 resident workgroups per CU (LDS-limited), persistent over the items
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, (const bf16*)qkv, (bf16*)out, S, H, BH,
                     scale * 1.4426950408889634f, lse, lse_stride > 0 ? lse_stride : S, lse_stride > 0 ? 32 * S : 32);
  return launch_status("attention_fwd");
}

}  // namespace mmamd

using namespace mmamd;

namespace mmamd {
// attention_ring.hip: LDS-DMA ring kernel (S <= 224), up to two problems per launch
bool attn_ring_supports(int S);
extern int g_attn_ring_abl;
extern int g_attn_ring_depth_cap;
int g_ln_rev = 0;        // mmamd_debug_set_attn_variant(3110 + r): grouped LayerNorm walks its rows 1 = last to first (A/B)
int g_ln_nt_policy = 0;  // mmamd_debug_set_attn_variant(3100 + p): LayerNorm x loads 0 = by size (default), 1 = never non-temporal, 2 = always (A/B)
int launch_attn_ring(const void* const* qkv, void* const* out, float* const* lse, const int* B, const int* S, const int* H, const int* causal,
                     const float* scale, int nprob, hipStream_t st, const int* lse_stride = nullptr);
// attention_probs_lse.hip: normalised probabilities from q, k and the saved log-sum-exp, stored as whole 256-byte segments
bool attn_probs_lse_supports(int S);
extern int g_probs_lse_abl, g_probs_lse_pad, g_probs_lse_waves;
int launch_attn_probs_lse(const void* qkv, float* probs, int B, int S, int H, float scale, hipStream_t st, const float* lse = nullptr);
}  // namespace mmamd

extern "C" int mmamd_debug_set_attn_variant(int v) {
  if (v == 512 || v == 513) {  // attention_probs_fwd: 512 = serial key loops, 513 = back to the pipelined default
    g_attn_probs_serial = v == 512;
    return 0;
  }
  if (v >= 5000 && v < 5100) { g_probs_lse_abl = v - 5000; return 0; }   // probabilities-from-lse kernel: ablation bits (MMAMD_EXPERIMENTS builds)
  if (v >= 5100 && v < 5300) { g_probs_lse_pad = v - 5100; return 0; }
  if (v == 5300 || v == 5308 || v == 5316) { g_probs_lse_waves = v - 5300; return 0; }  // ... waves per workgroup (A/B)   // ... extra LDS per workgroup in KiB (occupancy A/B)
  if (v == 514 || v == 515) {  // attention_probs_fwd without a key mask: 514 = the two-pass kernel, 515 = back to flash forward + probabilities pass
    g_attn_probs_twopass = v == 514;
    return 0;
  }
  if (v >= 2000 && v < 3000) {  // ring kernel ablations (timing only): 2000 + {1: no DMA, 2: no key loops, 4: no Q loads, 8: no O stores}
    g_attn_ring_abl = v - 2000;
    return 0;
  }
  if (v >= 3100 && v < 3103) {
    g_ln_nt_policy = v - 3100;
    return 0;
  }
  if (v >= 3110 && v < 3112) {
    g_ln_rev = v - 3110;
    return 0;
  }
  if (v >= 4000 && v < 4004) {  // attention BACKWARD form (the forward keeps its default)
    g_attn_bwd_variant = v == 4003 ? 0 : v;
    return 0;
  }
  if (v >= 3000 && v < 3008) {  // ring kernel: cap of the ring depth (0 = none), timing A/B only
    g_attn_ring_depth_cap = v - 3000;
    return 0;
  }
  g_attn_variant = v;
  return 0;
}

static int attention_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, int causal, float scale, mmamd_stream_t stream,
                              int lse_stride = 0);


extern "C" int mmamd_attention_fwd(const void* qkv, void* out, int B, int S, int H, int causal, float scale,
                                   mmamd_stream_t stream) {
  return attention_fwd_impl(qkv, out, nullptr, B, S, H, causal, scale, stream);
}

extern "C" int mmamd_attention_fwd_lse(const void* qkv, void* out, float* lse, int B, int S, int H, int causal, float scale,
                                       mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(lse != nullptr, MMAMD_E_BADARG, "attention_fwd_lse: lse is NULL");
  return attention_fwd_impl(qkv, out, lse, B, S, H, causal, scale, stream);
}

// lse_stride: 0 = the dense [B, H, S] log-sum-exp layout; S * S = PARKED in a [B, H, S, S] probability tensor: query q of a head at float
// (q / 32) * 32 S + q % 32 of the head's block (the first 32 floats of the 32-row band its probabilities will fill; attention_probs_lse.hip)
static int attention_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, int causal, float scale, mmamd_stream_t stream,
                              int lse_stride) {
  MMAMD_CHECK_ARG(qkv && out && B >= 0 && S > 0 && H > 0, MMAMD_E_BADARG, "attention: bad argument");
  MMAMD_CHECK_ARG(S <= 288 || lse == nullptr, MMAMD_E_UNSUPPORTED, "attention: S=%d > 288 has no training forward (the streaming kernel does not save the log-sum-exp)", S);
  MMAMD_CHECK_ARG(aligned16(qkv) && aligned16(out), MMAMD_E_ALIGN, "attention: pointers must be 16-byte aligned");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (S > 288) return launch_attn_long(qkv, nullptr, out, nullptr, MMAMD_F32, B, S, H, causal, scale, st);
  const int nkt = (S + 31) / 32;
  // default for S <= 224: the LDS-DMA ring kernel (attention_ring.hip); mmamd_debug_set_attn_variant(1000) keeps the r02 register-staged kernel (A/B)
  if (g_attn_variant == 0 && attn_ring_supports(S))
    return launch_attn_ring(&qkv, &out, &lse, &B, &S, &H, &causal, &scale, 1, st, lse_stride > 0 ? &lse_stride : nullptr);
  if (g_attn_variant != 0 && g_attn_variant != 1000 && nkt == 7 && !causal) {  // ablations, vision shape only
    switch (g_attn_variant) {
      case 1: return launch_attn<7, false, 1>(qkv, out, B, S, H, scale, st);
      case 2: return launch_attn<7, false, 2>(qkv, out, B, S, H, scale, st);
      case 3: return launch_attn<7, false, 3>(qkv, out, B, S, H, scale, st);
      case 4: return launch_attn<7, false, 4>(qkv, out, B, S, H, scale, st);
      case 7: return launch_attn<7, false, 7>(qkv, out, B, S, H, scale, st);
      case 8: return launch_attn<7, false, 8>(qkv, out, B, S, H, scale, st, lse);
      case 128: return launch_attn<7, false, 128>(qkv, out, B, S, H, scale, st, lse);
      case 136: return launch_attn<7, false, 136>(qkv, out, B, S, H, scale, st, lse);
      case 256: return launch_attn<7, false, 256>(qkv, out, B, S, H, scale, st, lse);
    }
  }
#define ATTN_CASE(N)                                                              \
  case N:                                                                         \
    return causal ? launch_attn<N, true>(qkv, out, B, S, H, scale, st, lse, lse_stride)       \
                  : launch_attn<N, false>(qkv, out, B, S, H, scale, st, lse, lse_stride);
  switch (nkt) {
    ATTN_CASE(1) ATTN_CASE(2) ATTN_CASE(3) ATTN_CASE(4) ATTN_CASE(5) ATTN_CASE(6) ATTN_CASE(7) ATTN_CASE(8) ATTN_CASE(9)
  }
#undef ATTN_CASE
  MMAMD_CHECK_ARG(false, MMAMD_E_UNSUPPORTED, "attention: unsupported S=%d", S);
}

extern "C" int mmamd_attention_fwd_grouped(const mmamd_attn_problem* probs, int nprob, float scale, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(probs != nullptr && nprob >= 1 && nprob <= 2, MMAMD_E_BADARG, "attention_fwd_grouped: 1 or 2 problems");
  const void* qkv[2]; void* out[2]; float* lse[2]; int B[2], S[2], H[2], causal[2]; float sc[2];
  bool ring = g_attn_variant == 0;
  for (int i = 0; i < nprob; ++i) {
    const mmamd_attn_problem& p = probs[i];
    MMAMD_CHECK_ARG(p.qkv && p.out && p.B >= 0 && p.S > 0 && p.H > 0, MMAMD_E_BADARG, "attention_fwd_grouped: bad argument (problem %d)", i);
    MMAMD_CHECK_ARG(aligned16(p.qkv) && aligned16(p.out), MMAMD_E_ALIGN, "attention_fwd_grouped: pointers must be 16-byte aligned");
    qkv[i] = p.qkv; out[i] = p.out; lse[i] = p.lse; B[i] = p.B; S[i] = p.S; H[i] = p.H; causal[i] = p.causal; sc[i] = scale;
    ring = ring && attn_ring_supports(p.S);
  }
  if (ring) return launch_attn_ring(qkv, out, lse, B, S, H, causal, sc, nprob, (hipStream_t)stream);
  for (int i = 0; i < nprob; ++i)  // shapes the ring kernel does not take: the launches this call stands for
    if (int rc = attention_fwd_impl(qkv[i], out[i], lse[i], B[i], S[i], H[i], causal[i], scale, stream)) return rc;
  return 0;
}

extern "C" int mmamd_attention_probs_fwd(const void* qkv, const uint8_t* key_mask, void* out, void* probs, int probs_dtype,
                                         int B, int S, int H, float scale, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(qkv && out && B >= 0 && S > 0 && H > 0, MMAMD_E_BADARG, "attention_probs: bad argument");
  MMAMD_CHECK_ARG(aligned16(qkv) && aligned16(out), MMAMD_E_ALIGN, "attention_probs: pointers must be 16-byte aligned");
  MMAMD_CHECK_ARG(probs == nullptr || probs_dtype == MMAMD_F32 || probs_dtype == MMAMD_BF16, MMAMD_E_BADARG, "attention_probs: bad probs dtype");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (S > 288) return launch_attn_long(qkv, key_mask, out, probs, probs_dtype, B, S, H, 0, scale, st);
  // No key-padding mask, fp32 (or no) probabilities of a sequence long enough for it to pay (no probabilities: 73 vs 150 us at S = 197, B = 256;
  // with them: (S = 129: 113 vs 130 us, S = 197: 218 vs 255 us, S = 275 at B = 128: 215 vs 270 us;
  // S = 77: 57 vs 46 us -- profiles/r05_probs_lse_bench_v3.txt) -> the flash forward, which parks each query's log-sum-exp in the first 32 floats of
  // the band of the probability tensor its row belongs to, then ONE pass that turns q.k into normalised probabilities and streams them out as whole
  // cache lines (attention_probs_lse.hip).  mmamd_debug_set_attn_variant(514) keeps the two-pass kernel below for every case (A/B).
  // (one predicate for both cases: the attention output must not depend on whether the probabilities were asked for)
  if (key_mask == nullptr && g_attn_probs_twopass == 0 && g_attn_variant == 0 && S >= 112 && attn_probs_lse_supports(S) &&
      (probs == nullptr || probs_dtype == MMAMD_F32)) {
    if (int rc = attention_fwd_impl(qkv, out, (float*)probs, B, S, H, 0, scale, stream, S * S)) return rc;
    return probs != nullptr ? launch_attn_probs_lse(qkv, (float*)probs, B, S, H, scale, st) : 0;
  }
  const int nkt = (S + 31) / 32;
#define ATTNP_CASE(N)                                                                                              \
  case N:                                                                                                          \
    return probs_dtype == MMAMD_F32 ? launch_attn_probs<N, float>(qkv, key_mask, out, probs, B, S, H, scale, st)   \
                                    : launch_attn_probs<N, bf16>(qkv, key_mask, out, probs, B, S, H, scale, st);
  switch (nkt) {
    ATTNP_CASE(1) ATTNP_CASE(2) ATTNP_CASE(3) ATTNP_CASE(4) ATTNP_CASE(5) ATTNP_CASE(6) ATTNP_CASE(7) ATTNP_CASE(8) ATTNP_CASE(9)
  }
#undef ATTNP_CASE
  MMAMD_CHECK_ARG(false, MMAMD_E_UNSUPPORTED, "attention_probs: unsupported S=%d", S);
}

extern "C" int mmamd_attention_probs_from_lse(const void* qkv, const float* lse, void* probs, int B, int S, int H, float scale,
                                              mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(qkv && lse && probs && B >= 0 && S > 0 && H > 0, MMAMD_E_BADARG, "attention_probs_from_lse: bad argument");
  MMAMD_CHECK_ARG(aligned16(qkv), MMAMD_E_ALIGN, "attention_probs_from_lse: qkv must be 16-byte aligned");
  MMAMD_CHECK_ARG(attn_probs_lse_supports(S), MMAMD_E_UNSUPPORTED, "attention_probs_from_lse: S=%d is not served (64 .. 288, not a multiple of 8)", S);
  if (B == 0) return 0;
  return launch_attn_probs_lse(qkv, (float*)probs, B, S, H, scale, (hipStream_t)stream, lse);
}

static AttnDrop make_attn_drop(float p, uint64_t seed, uint32_t site) {
  AttnDrop d;
  d.thresh = p > 0.f ? dropout_threshold(p) : 0u;
  if (p > 0.f && d.thresh == 0u) d.thresh = 1u;  // (p below 2^-32 still means "dropout on")
  d.k0 = (uint32_t)seed; d.k1 = (uint32_t)(seed >> 32); d.site = site;
  d.scale = 1.0f / (1.0f - p);
  return d;
}

static int attention_x_fwd_impl(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask, int64_t full_mask_batch_stride,
                                int causal, void* out, int ldo, void* probs, int probs_dtype, float* lse, int B, int Sq, int Sk, int H,
                                int head_dim, float scale, float drop_p, uint64_t seed, uint32_t site, mmamd_stream_t stream,
                                const float* head_mask = nullptr, const int64_t* hm_strides = nullptr);

extern "C" int mmamd_attention_x_fwd(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                     int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                     int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                                     float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, mmamd_stream_t stream) {
  return attention_x_fwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              ldo, probs, probs_dtype, lse, B, Sq, Sk, H, head_dim, scale, 0.f, 0, 0, stream);
}

extern "C" int mmamd_attention_x_fwd_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                               int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                               int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                                               float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, const float* head_mask,
                                               int64_t hm_stride_b, int64_t hm_stride_h, int64_t hm_stride_q, int64_t hm_stride_k,
                                               mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(head_mask != nullptr && hm_stride_b >= 0 && hm_stride_h >= 0 && hm_stride_q >= 0 && hm_stride_k >= 0, MMAMD_E_BADARG,
                  "attention_x: head_mask must be given with non-negative element strides");
  const int64_t hms[4] = {hm_stride_b, hm_stride_h, hm_stride_q, hm_stride_k};
  return attention_x_fwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              ldo, probs, probs_dtype, lse, B, Sq, Sk, H, head_dim, scale, 0.f, 0, 0, stream, head_mask, hms);
}

extern "C" int mmamd_attention_x_fwd_dropout(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                             int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                             int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                                             float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed,
                                             uint32_t site, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, MMAMD_E_BADARG, "attention_x: dropout p = %g must be in [0, 1)", (double)drop_p);
  return attention_x_fwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              ldo, probs, probs_dtype, lse, B, Sq, Sk, H, head_dim, scale, drop_p, seed, site, stream);
}

extern "C" int mmamd_attention_x_fwd_dropout_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                                       int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                                       int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                                                       float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, float drop_p,
                                                       uint64_t seed, uint32_t site, const float* head_mask, int64_t hm_stride_b,
                                                       int64_t hm_stride_h, int64_t hm_stride_q, int64_t hm_stride_k, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, MMAMD_E_BADARG, "attention_x: dropout p = %g must be in [0, 1)", (double)drop_p);
  MMAMD_CHECK_ARG(head_mask != nullptr && hm_stride_b >= 0 && hm_stride_h >= 0 && hm_stride_q >= 0 && hm_stride_k >= 0, MMAMD_E_BADARG,
                  "attention_x: head_mask must be given with non-negative element strides");
  const int64_t hms[4] = {hm_stride_b, hm_stride_h, hm_stride_q, hm_stride_k};
  return attention_x_fwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              ldo, probs, probs_dtype, lse, B, Sq, Sk, H, head_dim, scale, drop_p, seed, site, stream, head_mask, hms);
}

static int attention_x_fwd_impl(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask, int64_t full_mask_batch_stride,
                                int causal, void* out, int ldo, void* probs, int probs_dtype, float* lse, int B, int Sq, int Sk, int H,
                                int head_dim, float scale, float drop_p, uint64_t seed, uint32_t site, mmamd_stream_t stream,
                                const float* head_mask, const int64_t* hm_strides) {
  MMAMD_CHECK_ARG(q && k && v && out && B >= 0 && Sq > 0 && Sk > 0 && H > 0, MMAMD_E_BADARG, "attention_x: bad argument");
  MMAMD_CHECK_ARG(head_dim == 64 || head_dim == 96, MMAMD_E_UNSUPPORTED, "attention_x: head_dim=%d (64 and 96 are built)", head_dim);
  MMAMD_CHECK_ARG(Sk <= 288, MMAMD_E_UNSUPPORTED, "attention_x: Sk=%d > 288 not supported", Sk);
  MMAMD_CHECK_ARG(causal >= 0 && causal <= 3, MMAMD_E_BADARG, "attention_x: causal is a 2-bit flag (1 = causal, 2 = key mask on the last query row only)");
  MMAMD_CHECK_ARG(!(causal & 1) || Sq == Sk, MMAMD_E_BADARG, "attention_x: causal needs Sq == Sk");
  MMAMD_CHECK_ARG(ldq >= H * head_dim && ldk >= H * head_dim && ldv >= H * head_dim && ldo >= H * head_dim, MMAMD_E_BADARG,
                  "attention_x: leading dimension smaller than H*head_dim");
  MMAMD_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && q_batch_stride % 8 == 0 && kv_batch_stride % 8 == 0 &&
                      aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out),
                  MMAMD_E_ALIGN, "attention_x: rows must be 16-byte aligned");
  if (B == 0) return 0;
  AttnX p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.out = (bf16*)out; p.probs = probs;
  p.key_mask = key_mask; p.full_mask = full_mask; p.lse = lse;
  p.q_bs = q_batch_stride; p.kv_bs = kv_batch_stride; p.fm_bs = full_mask_batch_stride;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.Sq = Sq; p.Sk = Sk; p.H = H; p.causal = causal & 1; p.km_last = (causal >> 1) & 1;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.drop = make_attn_drop(drop_p, seed, site);
  p.hmask = head_mask;
  if (head_mask != nullptr) { p.hm_sb = hm_strides[0]; p.hm_sh = hm_strides[1]; p.hm_sq = hm_strides[2]; p.hm_sk = hm_strides[3]; }
  hipStream_t st = (hipStream_t)stream;
  const bool pf32 = probs == nullptr || probs_dtype == MMAMD_F32;
  if (head_dim == 64) return pf32 ? dispatch_attn_x<64, float>(p, B, st) : dispatch_attn_x<64, bf16>(p, B, st);
  return pf32 ? dispatch_attn_x<96, float>(p, B, st) : dispatch_attn_x<96, bf16>(p, B, st);
}

extern "C" int mmamd_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const uint8_t* key_mask, void* dqkv,
                                   int B, int S, int H, int causal, float scale, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(qkv && out && dout && lse && dqkv && B >= 0 && S > 0 && H > 0, MMAMD_E_BADARG, "attention_bwd: bad argument");
  MMAMD_CHECK_ARG(S <= 288, MMAMD_E_UNSUPPORTED, "attention_bwd: S=%d > 288 not supported", S);
  MMAMD_CHECK_ARG(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), MMAMD_E_ALIGN, "attention_bwd: pointers must be 16-byte aligned");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // non-causal heads of five and more tiles (ViT-B/16: 7): the single-pass kernel -- 292-311 us against 308-322 us for the two-role fused kernel and 364-398 us
  // for the two-kernel form at B = 256 (tools/attn_bwd_bench.py); causal heads keep the fused kernel (the skewed schedule idles on the pairs above the diagonal:
  // text tower 76 vs 55 us).  mmamd_debug_set_attn_variant(4001) forces it wherever it is built, 4002 the fused kernel, 4000 the two kernels.
  if (g_attn_bwd_variant == 4001 || (g_attn_bwd_variant == 0 && !causal)) {
    switch ((S + 31) / 32) {
      case 3: if (g_attn_bwd_variant == 4001) return launch_attn_bwd_sp<3>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask); break;
      case 5: return launch_attn_bwd_sp<5>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 6: return launch_attn_bwd_sp<6>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 7: return launch_attn_bwd_sp<7>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 8: return launch_attn_bwd_sp<8>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    }
  }
  if (g_attn_bwd_variant != 4000) {  // the fused kernel (S <= 256); mmamd_debug_set_attn_variant(4000) keeps the two-kernel form for the A/B (codes 2000-3999 belong to the ring kernel's ablations)
    switch ((S + 31) / 32) {
      case 1: return launch_attn_bwd_fused<1, 4>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 2: return launch_attn_bwd_fused<2, 4>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 3: return launch_attn_bwd_fused<3, 4>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 4: return launch_attn_bwd_fused<4, 4>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 5: return launch_attn_bwd_fused<5, 8>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 6: return launch_attn_bwd_fused<6, 8>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 7: return launch_attn_bwd_fused<7, 8>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      case 8: return launch_attn_bwd_fused<8, 8>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
      default: break;
    }
  }
  switch ((S + 31) / 32) {
    case 1: return launch_attn_bwd<1>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 2: return launch_attn_bwd<2>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 3: return launch_attn_bwd<3>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 4: return launch_attn_bwd<4>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 5: return launch_attn_bwd<5>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 6: return launch_attn_bwd<6>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 7: return launch_attn_bwd<7>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 8: return launch_attn_bwd<8>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
    case 9: return launch_attn_bwd<9>(qkv, out, dout, lse, dqkv, B, S, H, causal, scale, st, key_mask);
  }
  MMAMD_CHECK_ARG(false, MMAMD_E_UNSUPPORTED, "attention_bwd: unsupported S=%d", S);
}
