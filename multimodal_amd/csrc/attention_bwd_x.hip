// attention_bwd_x.hip — backward of the GENERAL attention (csrc/attention.hip: attention_x_kernel): separate strided q / k / v,
// Sq != Sk, head dim 64 or 96, causal / key-padding / full masks, batch-shared queries (their gradient comes back per sample and
// is summed by the caller).  Same math and layouts as attention_bwd_{dq,dkv}_kernel; two kernels, no atomics:
//   dQ  : one workgroup per (batch, head); K rows, K^T and V rows of the whole head in LDS; a wave owns 32 queries per step
//   dK,dV: one workgroup per (batch, head); a wave owns 32 keys per round; Q, Q^T, dO, dO^T are staged in chunks of 128 queries
//          (the pooler's 257 queries x 96-wide heads do not fit LDS at once), accumulators persist across the chunks of a round
// Reference: torch autograd of F.scaled_dot_product_attention in MultiHeadAttentionWithCache / MultiHeadSelfAttention
// (modules/layers/multi_head_attention.py:69-71,165-167).
#include "common.h"

namespace mmamd {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct AttnXB {
  const bf16 *q, *k, *v, *O, *dO;
  const float* lse;
  bf16 *dq, *dk, *dv;
  const uint8_t *key_mask, *full_mask;
  long long q_bs, kv_bs, fm_bs, dq_bs, dkv_bs;
  int ldq, ldk, ldv, ldo, lddq, lddk, lddv, Sq, Sk, H, causal;
  int km_last = 0;  // 1: key_mask applies to the last query row only (argument `causal` bit 1), as in the forward
  float scale;
  AttnDrop drop;  // the forward's dropout on the probabilities (thresh = 0: none): O = P' V with P' = P keep / (1 - p), so
                  // dV = P'^T dO, dP = (dO V^T) keep / (1 - p), dS = P (dP - D) with D = sum dO O (unchanged form)
  // the forward's head_mask (reference modules/layers/attention.py:236-237: attn = attn * head_mask AFTER softmax and dropout; a constant): it multiplies
  // exactly where the dropout factor does -- P' = P m, dV = P'^T dO, dP = (dO V^T) m, D unchanged.  fp32, element strides of its [b, h, q, k] broadcast
  const float* hmask = nullptr;
  long long hm_sb = 0, hm_sh = 0, hm_sq = 0, hm_sk = 0;
};

template <int DH, bool DROP, bool HM>
__global__ __launch_bounds__(256) void attention_x_bwd_dq_kernel(const AttnXB p) {
  constexpr int KS = DH + 8;
  constexpr int CPR = DH / 8;
  const int Sq = p.Sq, Sk = p.Sk;
  const int SP = ((Sk + 31) >> 5) << 5, VS = SP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = Ks + (size_t)SP * KS;
  bf16* Kt = Vs + (size_t)SP * KS;
  uint8_t* Mk = reinterpret_cast<uint8_t*>(Kt + (size_t)DH * VS);  // 1 = key may be attended
  const int bh = blockIdx.x;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16* kb = p.k + (size_t)b * p.kv_bs + h * DH;
  const bf16* vb = p.v + (size_t)b * p.kv_bs + h * DH;
  const bf16* qb = p.q + (size_t)b * p.q_bs + h * DH;
  // K / V staging, four trips at a time: the eight 16-byte loads of a group are issued together, unconditionally on clamped rows, then stored (r06: a
  // load inside `if (r < Sk)` followed by its stores was one exposed global round trip per trip -- eight in a row at Sk = 256 before any compute)
  for (int i0 = tid; i0 < SP * CPR; i0 += 4 * 256) {
    bf16x8 kr[4], vr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + u * 256;
      i = i < SP * CPR ? i : SP * CPR - 1;
      const int r = i / CPR, c = i - r * CPR;
      const int rc = r < Sk ? r : Sk - 1;
      kr[u] = *reinterpret_cast<const bf16x8*>(kb + (size_t)rc * p.ldk + c * 8);
      vr[u] = *reinterpret_cast<const bf16x8*>(vb + (size_t)rc * p.ldv + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      if (i < SP * CPR) {
        const int r = i / CPR, c = i - r * CPR;
        bf16x8 kv = kr[u], vv = vr[u];
        if (r >= Sk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; }
        }
        *reinterpret_cast<bf16x8*>(Ks + r * KS + c * 8) = kv;
        *reinterpret_cast<bf16x8*>(Vs + r * KS + c * 8) = vv;
#pragma unroll
        for (int j = 0; j < 8; ++j) Kt[(c * 8 + j) * VS + r] = kv[j];
      }
    }
  }
  for (int k = tid; k < SP; k += 256) Mk[k] = (k < Sk && (p.key_mask == nullptr || p.key_mask[(size_t)b * Sk + k] != 0)) ? 1 : 0;
  __syncthreads();
  const int l31 = lane & 31, half = lane >> 5;
  const float c2 = p.scale * 1.4426950408889634f;
  const int nqt = (Sq + 31) >> 5, nkt = SP >> 5;
  const uint8_t* fm = p.full_mask ? p.full_mask + (size_t)b * p.fm_bs : nullptr;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q = qt * 32 + l31;
    const int qc = q < Sq ? q : Sq - 1;
    bf16x8 qf[DH / 16], dof[DH / 16];
    float dpart = 0.f;
    const bf16* orow = p.O + ((size_t)b * Sq + qc) * p.ldo + h * DH + 8 * half;
    const bf16* dorow = p.dO + ((size_t)b * Sq + qc) * p.ldo + h * DH + 8 * half;
#pragma unroll
    for (int t = 0; t < DH / 16; ++t) {
      qf[t] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qc * p.ldq + 16 * t + 8 * half);
      dof[t] = *reinterpret_cast<const bf16x8*>(dorow + 16 * t);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + 16 * t);
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += (float)dof[t][j] * (float)of[j];
    }
    const float Dq = dpart + __shfl_xor(dpart, 32);
    const float L2 = p.lse[((size_t)b * p.H + h) * Sq + qc];
    f32x16 acc[DH / 32];
#pragma unroll
    for (int nt = 0; nt < DH / 32; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int kt_hi = p.causal ? (qt + 1 < nkt ? qt + 1 : nkt) : nkt;
#pragma unroll 1
    for (int kt = 0; kt < kt_hi; ++kt) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
      const bf16* krow = Ks + (kt * 32 + l31) * KS + 8 * half;
      const bf16* vrow = Vs + (kt * 32 + l31) * KS + 8 * half;
#pragma unroll
      for (int t = 0; t < DH / 16; ++t) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + 16 * t), qf[t], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(vrow + 16 * t), dof[t], dp, 0, 0, 0);
      }
      uint32_t pk[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float e[4];
        Philox4 rr = {{0u, 0u, 0u, 0u}};
        if constexpr (DROP) rr = attn_drop_block(p.drop, ((long long)b * p.H + h) * Sq + qc, (Sk + 3) >> 2, kt * 32 + 8 * g + 4 * half);
        uint32_t fmw = 0x01010101u;  // the group's 4 consecutive full-mask bytes as one (unaligned) 32-bit load, like the forward kernel
        if (fm != nullptr) {
          const int key0 = kt * 32 + 8 * g + 4 * half;
          const uint8_t* src = fm + (size_t)qc * Sk + key0;
          if (key0 + 3 < Sk) {
            __builtin_memcpy(&fmw, src, 4);
          } else {
            fmw = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (key0 + j < Sk) fmw |= (uint32_t)src[j] << (8 * j);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const int key = kt * 32 + 8 * g + 4 * half + j;
          bool ok = ((p.km_last && qc != Sq - 1) ? key < Sk : Mk[key] != 0) && (!p.causal || key <= qc);
          ok = ok && ((fmw >> (8 * j)) & 0xffu) != 0;
          const float pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2) : 0.f;
          float dpr = dp[r];
          if constexpr (DROP) dpr = rr.v[j] >= p.drop.thresh ? dpr * p.drop.scale : 0.f;
          if constexpr (HM) dpr *= ok ? p.hmask[(long long)b * p.hm_sb + (long long)h * p.hm_sh + (long long)qc * p.hm_sq + (long long)key * p.hm_sk] : 0.f;
          e[j] = pr * (dpr - Dq);
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
        for (int nt = 0; nt < DH / 32; ++nt) {
          const bf16* ktrow = Kt + (nt * 32 + l31) * VS + key0;
          const uint2 v0 = *reinterpret_cast<const uint2*>(ktrow);
          const uint2 v1 = *reinterpret_cast<const uint2*>(ktrow + 8);
          u32x4 vw;
          vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), dsf, acc[nt], 0, 0, 0);
        }
      }
    }
    if (q < Sq) {
      bf16* dst = p.dq + (size_t)b * p.dq_bs + (size_t)q * p.lddq + h * DH;
#pragma unroll
      for (int nt = 0; nt < DH / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = acc[nt][4 * g + j] * p.scale;
          store4(dst + nt * 32 + 8 * g + 4 * half, o);
        }
    }
  }
}

template <int DH, bool DROP, bool HM>
__global__ __launch_bounds__(256) void attention_x_bwd_dkv_kernel(const AttnXB p) {
  constexpr int KS = DH + 8;
  constexpr int CPR = DH / 8;
  constexpr int QC = 128, QVS = QC + 4;  // queries staged per chunk; row pitch of the transposed images
  const int Sq = p.Sq, Sk = p.Sk;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* Qs = reinterpret_cast<bf16*>(smem);          // [QC][KS]
  bf16* dOs = Qs + QC * KS;                           // [QC][KS]
  bf16* Qt = dOs + QC * KS;                           // [DH][QVS]
  bf16* dOt = Qt + DH * QVS;                          // [DH][QVS]
  float* L2s = reinterpret_cast<float*>(dOt + DH * QVS);  // [QC]
  float* Dqs = L2s + QC;                                   // [QC]
  const int bh = blockIdx.x;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16* kb = p.k + (size_t)b * p.kv_bs + h * DH;
  const bf16* vb = p.v + (size_t)b * p.kv_bs + h * DH;
  const bf16* qb = p.q + (size_t)b * p.q_bs + h * DH;
  const int l31 = lane & 31, half = lane >> 5;
  const float c2 = p.scale * 1.4426950408889634f;
  const int nkt = (Sk + 31) >> 5, nchunk = (Sq + QC - 1) / QC;
  const uint8_t* fm = p.full_mask ? p.full_mask + (size_t)b * p.fm_bs : nullptr;
  // Q / dO (row-major and transposed images), D = rowsum(dO * O) and the log-sum-exp of one chunk of <= 128 queries -> LDS.  Two trips at a time: the six
  // 16-byte loads of a pair are issued together, unconditionally on clamped rows, before the first store (r06: the predicated form paid one exposed global
  // round trip per trip, and the cross-attention's second round of key tiles staged the same chunk again)
  auto stage_chunk = [&](int qbase) __attribute__((always_inline)) {
    __syncthreads();  // the previous chunk (or round) is consumed
    for (int i = tid; i < QC; i += 256) { L2s[i] = INFINITY; Dqs[i] = 0.f; }
    __syncthreads();
    constexpr int NTRIP = QC * CPR / 256;
    static_assert(QC * CPR % 256 == 0 && NTRIP % 2 == 0, "chunk staging: whole pairs of trips");
#pragma unroll
    for (int t0 = 0; t0 < NTRIP; t0 += 2) {
      bf16x8 qr[2], dr[2], orr[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + (t0 + u) * 256;
        const int r = i / CPR, c = i - r * CPR;
        const int q = qbase + r < Sq ? qbase + r : Sq - 1;
        qr[u] = *reinterpret_cast<const bf16x8*>(qb + (size_t)q * p.ldq + c * 8);
        dr[u] = *reinterpret_cast<const bf16x8*>(p.dO + ((size_t)b * Sq + q) * p.ldo + h * DH + c * 8);
        orr[u] = *reinterpret_cast<const bf16x8*>(p.O + ((size_t)b * Sq + q) * p.ldo + h * DH + c * 8);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + (t0 + u) * 256;
        const int r = i / CPR, c = i - r * CPR;
        const int q = qbase + r;
        bf16x8 qv = qr[u], dv = dr[u], ov = orr[u];
        if (q >= Sq) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { qv[j] = (bf16)0.f; dv[j] = (bf16)0.f; ov[j] = (bf16)0.f; }
        }
        *reinterpret_cast<bf16x8*>(Qs + r * KS + c * 8) = qv;
        *reinterpret_cast<bf16x8*>(dOs + r * KS + c * 8) = dv;
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          Qt[(c * 8 + j) * QVS + r] = qv[j];
          dOt[(c * 8 + j) * QVS + r] = dv[j];
          part += (float)dv[j] * (float)ov[j];
        }
        if (q < Sq) atomicAdd(&Dqs[r], part);  // the CPR chunks of a row are not one aligned lane group when CPR = 12: LDS atomics
        if (c == 0 && q < Sq) L2s[r] = p.lse[((size_t)b * p.H + h) * Sq + q];
      }
    }
    __syncthreads();
  };
  for (int r0 = 0; r0 < nkt; r0 += 4) {
    const int kt = r0 + wave;
    const bool active = kt < nkt;
    const int key = kt * 32 + l31;
    const int kc = key < Sk ? key : Sk - 1;
    const bool key_live = active && key < Sk && (p.key_mask == nullptr || p.key_mask[(size_t)b * Sk + kc] != 0);
    bf16x8 kf[DH / 16], vf[DH / 16];
#pragma unroll
    for (int t = 0; t < DH / 16; ++t) {
      kf[t] = *reinterpret_cast<const bf16x8*>(kb + (size_t)kc * p.ldk + 16 * t + 8 * half);
      vf[t] = *reinterpret_cast<const bf16x8*>(vb + (size_t)kc * p.ldv + 16 * t + 8 * half);
    }
    f32x16 dv_acc[DH / 32], dk_acc[DH / 32];
#pragma unroll
    for (int nt = 0; nt < DH / 32; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv_acc[nt][r] = 0.f; dk_acc[nt][r] = 0.f; }
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      const int qbase = ch * QC;
      if (nchunk > 1 || r0 == 0) stage_chunk(qbase);  // one chunk (Sq <= 128): staged once, every round of key tiles reads the same images
      if (active) {
        const int nq_here = (Sq - qbase) < QC ? (Sq - qbase) : QC;
        const int nqt = (nq_here + 31) >> 5;
#pragma unroll 1
        for (int qt = 0; qt < nqt; ++qt) {
          if (p.causal && qbase + qt * 32 + 31 < kt * 32) continue;  // every query of the tile precedes every key of the tile
          f32x16 st, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
          const bf16* qrow = Qs + (qt * 32 + l31) * KS + 8 * half;
          const bf16* drow = dOs + (qt * 32 + l31) * KS + 8 * half;
#pragma unroll
          for (int t = 0; t < DH / 16; ++t) {
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qrow + 16 * t), kf[t], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(drow + 16 * t), vf[t], dp, 0, 0, 0);
          }
          uint32_t pk[8], dk[8];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float e[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * g + j;
              const int ql = qt * 32 + 8 * g + 4 * half + j;  // query index inside the chunk
              const int q = qbase + ql;
              bool ok = ((p.km_last && q != Sq - 1) ? (active && key < Sk) : key_live) && q < Sq && (!p.causal || key <= q);
              if (ok && fm != nullptr) ok = fm[(size_t)q * Sk + key] != 0;
              const float pr = ok ? __builtin_amdgcn_exp2f(st[r] * c2 - L2s[ql]) : 0.f;
              float pd = pr, dpr = dp[r];
              if constexpr (DROP) {  // lane = key: every query of the tile needs its own Philox block (word key & 3)
                const int qq = q < Sq ? q : Sq - 1;
                const Philox4 rr = attn_drop_block(p.drop, ((long long)b * p.H + h) * Sq + qq, (Sk + 3) >> 2, kc);
                const bool keep = rr.v[kc & 3] >= p.drop.thresh;
                pd = keep ? pr * p.drop.scale : 0.f;
                dpr = keep ? dpr * p.drop.scale : 0.f;
              }
              if constexpr (HM) {
                const float m = ok ? p.hmask[(long long)b * p.hm_sb + (long long)h * p.hm_sh + (long long)q * p.hm_sq + (long long)key * p.hm_sk] : 0.f;
                pd *= m;
                dpr *= m;
              }
              e[j] = pd;
              f[j] = pr * (dpr - Dqs[ql]);
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
            for (int nt = 0; nt < DH / 32; ++nt) {
              const bf16* dorow = dOt + (nt * 32 + l31) * QVS + q0;
              const bf16* qtrow = Qt + (nt * 32 + l31) * QVS + q0;
              const uint2 a0 = *reinterpret_cast<const uint2*>(dorow), a1 = *reinterpret_cast<const uint2*>(dorow + 8);
              const uint2 b0 = *reinterpret_cast<const uint2*>(qtrow), b1 = *reinterpret_cast<const uint2*>(qtrow + 8);
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
    if (active && key < Sk) {
      bf16* dkd = p.dk + (size_t)b * p.dkv_bs + (size_t)key * p.lddk + h * DH;
      bf16* dvd = p.dv + (size_t)b * p.dkv_bs + (size_t)key * p.lddv + h * DH;
#pragma unroll
      for (int nt = 0; nt < DH / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 ok_, ov_;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ok_[j] = dk_acc[nt][4 * g + j] * p.scale; ov_[j] = dv_acc[nt][4 * g + j]; }
          store4(dkd + nt * 32 + 8 * g + 4 * half, ok_);
          store4(dvd + nt * 32 + 8 * g + 4 * half, ov_);
        }
    }
  }
}

template <int DH, bool DROP, bool HM = false>
static int launch_x_bwd(const AttnXB& p, int B, hipStream_t st) {
  const int SP = ((p.Sk + 31) / 32) * 32;
  const int smem1 = 2 * SP * (DH + 8) * 2 + DH * (SP + 4) * 2 + SP;
  constexpr int smem2 = 2 * 128 * (DH + 8) * 2 + 2 * DH * 132 * 2 + 2 * 128 * 4;
  if (smem1 > 160 * 1024) { set_error("attention_x_bwd: Sk=%d with head_dim=%d needs %d B of LDS (> 160 KiB)", p.Sk, DH, smem1); return MMAMD_E_UNSUPPORTED; }
  auto k1 = attention_x_bwd_dq_kernel<DH, DROP, HM>;
  auto k2 = attention_x_bwd_dkv_kernel<DH, DROP, HM>;
  // per-device opt-in to > 64 KiB dynamic LDS; the dQ kernel's size varies with Sk, so it opts in to the 160 KiB maximum once
  static unsigned long long m1 = 0, m2 = 0;
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(k1), smem1 > 64 * 1024 ? 160 * 1024 : 0, m1)) return rc_attr;
  if (int rc_attr = opt_in_lds(reinterpret_cast<const void*>(k2), smem2, m2)) return rc_attr;
  hipLaunchKernelGGL(k1, dim3(B * p.H), dim3(256), smem1, st, p);
  hipLaunchKernelGGL(k2, dim3(B * p.H), dim3(256), smem2, st, p);
  return launch_status("attention_x_bwd");
}

}  // namespace mmamd

using namespace mmamd;

static int attention_x_bwd_impl(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask, int64_t full_mask_batch_stride,
                                int causal, const void* out, const void* dout, int ldo, const float* lse, void* dq, int lddq, void* dk, void* dv,
                                int lddk, int lddv, int B, int Sq, int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed,
                                uint32_t site, mmamd_stream_t stream, const float* head_mask = nullptr, const int64_t* hm_strides = nullptr);

extern "C" int mmamd_attention_x_bwd(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                     int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                     int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo,
                                     const float* lse, void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq, int Sk,
                                     int H, int head_dim, float scale, mmamd_stream_t stream) {
  return attention_x_bwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              dout, ldo, lse, dq, lddq, dk, dv, lddk, lddv, B, Sq, Sk, H, head_dim, scale, 0.f, 0, 0, stream);
}

extern "C" int mmamd_attention_x_bwd_dropout(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                             int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                             int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo,
                                             const float* lse, void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq,
                                             int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed, uint32_t site,
                                             mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, MMAMD_E_BADARG, "attention_x_bwd: dropout p = %g must be in [0, 1)", (double)drop_p);
  return attention_x_bwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              dout, ldo, lse, dq, lddq, dk, dv, lddk, lddv, B, Sq, Sk, H, head_dim, scale, drop_p, seed, site, stream);
}

extern "C" int mmamd_attention_x_bwd_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                               int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                               int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo,
                                               const float* lse, void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq,
                                               int Sk, int H, int head_dim, float scale, const float* head_mask, int64_t hm_stride_b,
                                               int64_t hm_stride_h, int64_t hm_stride_q, int64_t hm_stride_k, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(head_mask != nullptr && hm_stride_b >= 0 && hm_stride_h >= 0 && hm_stride_q >= 0 && hm_stride_k >= 0, MMAMD_E_BADARG,
                  "attention_x_bwd: head_mask must be given with non-negative element strides");
  const int64_t hms[4] = {hm_stride_b, hm_stride_h, hm_stride_q, hm_stride_k};
  return attention_x_bwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              dout, ldo, lse, dq, lddq, dk, dv, lddk, lddv, B, Sq, Sk, H, head_dim, scale, 0.f, 0, 0, stream, head_mask, hms);
}

extern "C" int mmamd_attention_x_bwd_dropout_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                                       int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                                       int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo,
                                                       const float* lse, void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq,
                                                       int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed, uint32_t site,
                                                       const float* head_mask, int64_t hm_stride_b, int64_t hm_stride_h, int64_t hm_stride_q,
                                                       int64_t hm_stride_k, mmamd_stream_t stream) {
  MMAMD_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, MMAMD_E_BADARG, "attention_x_bwd: dropout p = %g must be in [0, 1)", (double)drop_p);
  MMAMD_CHECK_ARG(head_mask != nullptr && hm_stride_b >= 0 && hm_stride_h >= 0 && hm_stride_q >= 0 && hm_stride_k >= 0, MMAMD_E_BADARG,
                  "attention_x_bwd: head_mask must be given with non-negative element strides");
  const int64_t hms[4] = {hm_stride_b, hm_stride_h, hm_stride_q, hm_stride_k};
  return attention_x_bwd_impl(q, ldq, q_batch_stride, k, v, ldk, ldv, kv_batch_stride, key_mask, full_mask, full_mask_batch_stride, causal, out,
                              dout, ldo, lse, dq, lddq, dk, dv, lddk, lddv, B, Sq, Sk, H, head_dim, scale, drop_p, seed, site, stream, head_mask, hms);
}

static int attention_x_bwd_impl(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask, int64_t full_mask_batch_stride,
                                int causal, const void* out, const void* dout, int ldo, const float* lse, void* dq, int lddq, void* dk, void* dv,
                                int lddk, int lddv, int B, int Sq, int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed,
                                uint32_t site, mmamd_stream_t stream, const float* head_mask, const int64_t* hm_strides) {
  MMAMD_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && B >= 0 && Sq > 0 && Sk > 0 && H > 0, MMAMD_E_BADARG,
                  "attention_x_bwd: bad argument");
  MMAMD_CHECK_ARG(head_dim == 64 || head_dim == 96, MMAMD_E_UNSUPPORTED, "attention_x_bwd: head_dim=%d (64 and 96 are built)", head_dim);
  MMAMD_CHECK_ARG(Sk <= 288, MMAMD_E_UNSUPPORTED, "attention_x_bwd: Sk=%d > 288 not supported", Sk);
  MMAMD_CHECK_ARG(causal >= 0 && causal <= 3, MMAMD_E_BADARG, "attention_x_bwd: causal is a 2-bit flag (1 = causal, 2 = key mask on the last query row only)");
  MMAMD_CHECK_ARG(!(causal & 1) || Sq == Sk, MMAMD_E_BADARG, "attention_x_bwd: causal needs Sq == Sk");
  const int D = H * head_dim;
  MMAMD_CHECK_ARG(ldq >= D && ldk >= D && ldv >= D && ldo >= D && lddq >= D && lddk >= D && lddv >= D, MMAMD_E_BADARG,
                  "attention_x_bwd: leading dimension smaller than H*head_dim");
  MMAMD_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 &&
                      q_batch_stride % 8 == 0 && kv_batch_stride % 8 == 0 && aligned16(q) && aligned16(k) && aligned16(v) &&
                      aligned16(out) && aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv),
                  MMAMD_E_ALIGN, "attention_x_bwd: rows must be 16-byte aligned");
  if (B == 0) return 0;
  AttnXB p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.O = (const bf16*)out; p.dO = (const bf16*)dout; p.lse = lse;
  p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv; p.key_mask = key_mask; p.full_mask = full_mask;
  p.q_bs = q_batch_stride; p.kv_bs = kv_batch_stride; p.fm_bs = full_mask_batch_stride;
  p.dq_bs = (long long)Sq * lddq; p.dkv_bs = (long long)Sk * lddk;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.Sq = Sq; p.Sk = Sk; p.H = H; p.causal = causal & 1; p.km_last = (causal >> 1) & 1; p.scale = scale;
  p.drop.thresh = drop_p > 0.f ? (dropout_threshold(drop_p) ? dropout_threshold(drop_p) : 1u) : 0u;
  p.drop.k0 = (uint32_t)seed; p.drop.k1 = (uint32_t)(seed >> 32); p.drop.site = site; p.drop.scale = 1.0f / (1.0f - drop_p);
  MMAMD_CHECK_ARG(lddk == lddv, MMAMD_E_BADARG, "attention_x_bwd: dk and dv must share their row pitch");
  hipStream_t st = (hipStream_t)stream;
  if (head_mask != nullptr) {
    p.hmask = head_mask; p.hm_sb = hm_strides[0]; p.hm_sh = hm_strides[1]; p.hm_sq = hm_strides[2]; p.hm_sk = hm_strides[3];
    if (p.drop.thresh != 0) return head_dim == 64 ? launch_x_bwd<64, true, true>(p, B, st) : launch_x_bwd<96, true, true>(p, B, st);  // r06: both
    return head_dim == 64 ? launch_x_bwd<64, false, true>(p, B, st) : launch_x_bwd<96, false, true>(p, B, st);
  }
  if (p.drop.thresh != 0) return head_dim == 64 ? launch_x_bwd<64, true>(p, B, st) : launch_x_bwd<96, true>(p, B, st);
  return head_dim == 64 ? launch_x_bwd<64, false>(p, B, st) : launch_x_bwd<96, false>(p, B, st);
}
