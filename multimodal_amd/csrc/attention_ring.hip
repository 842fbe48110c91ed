// attention_ring.hip — multi-head self-attention forward for short sequences (S <= 208, head dim 64) built around
// LDS-DMA staging: a dedicated LOADER wave streams K / V of the coming (batch, head) items into a ring of LDS slots
// with `global_load_lds_dwordx4` (no VGPR round trip, no ds_write, no prefetch registers) while the other seven waves
// of the workgroup compute out of slots that have landed.  One workgroup (8 waves) per CU, persistent.
//
// Why (r02 VERDICT item 1): attention.hip's attention_fwd_kernel sat at 0.40 of the HBM roofline (96.8 us in the
// ViT-B/16 B = 256 step for 310 MB).  Its K/V went global -> VGPR -> ds_write; the 56 prefetch registers cost the
// third wave per SIMD, 21 of 87 us were exposed K/V load latency and 7 query tiles over 4 waves idled 12.5 %.
//
// Structure
//   * ROUND = G whole items (G = 7 / nqt, nqt = ceil(S / 32) query tiles per item): ViT S = 197 -> one item per round,
//     query tile w on compute wave w (7 tiles on 7 waves: no 7-over-4 imbalance); text S = 77 -> two items per round
//     on 6 waves.  A round's K / V / Q is one ring ENTRY; the ring holds as many entries as fit 160 KiB (2 at S = 197: 2 x 78 KiB).
//   * The loader issues entry r + nring - 1 right after the barrier that retires entry r - 1 and arrives at the
//     barrier that ends round r only when entry r + 1 has landed (counted s_waitcnt vmcnt: its queue holds nothing
//     but its own DMA pieces, in order).  The next entry (78 KB at S = 197) is in flight per CU while one is being computed.
//   * Q and O go through LDS in FULL 128-byte rows as well (r03 ablation of the first version of this kernel, which loaded the Q
//     fragments and stored O in MFMA-fragment shape — 32 rows x 16..32 B per wave-instruction: 111.7 us as built, 82.8 without the Q
//     loads, 69.6 without the O stores, 58.0 without both; profiles/r03_attn_ring_ablation.txt): the loader DMAs the item's Q rows
//     behind its K / V rows, a compute wave reads its 32 x 64 Q tile with four ds_read_b128, and when the tile is done it writes the
//     normalised bf16 O tile over its own (now dead, wave-private) Q rows, reads it back as whole rows and stores 8 rows x 128 B per
//     wave-instruction.
//   * LDS image per item: K rows, V rows, then Q rows, 128 B per row, UNPADDED (the DMA destination is lane-linear), bank
//     swizzle on the per-lane SOURCE address and undone on the read side (guide rule 21):
//         chunk position = chunk ^ f(row),  f(row) = (((row >> 1) & 1) << 2) | ((row >> 2) & 3)
//     -> the 16 lanes of a ds_read_b128 group (K fragments: rows {0-3, 12-15, 20-27} x one chunk) and the 32 lanes of a
//     ds_read_b64_tr_b16 group (V fragments: 4 rows x 4 chunks) each cover all 16 slots of the 256-byte bank row.
//   * K / V rows >= S are not staged (the images hold roundup8(S) rows); the last key tile reads whatever follows.  Scores of
//     keys >= S are replaced by -inf before they are used (P = 0 exactly) and the V elements of such keys are ANDed to
//     zero in the peeled last tile, so nothing read beyond the item reaches the output.
//   * Compute waves: the software-pipelined key loop of attention_fwd_kernel (QK^T of tile k+1 issued before the
//     softmax of tile k, swapped products, deferred rescale, packed bf16 P as the MFMA operand) — the same arithmetic in
//     the same order, so non-causal results are BIT-IDENTICAL to attention_fwd_kernel's (tests/test_gpu_attention_ring.py).
//     Causal items skip tiles above the diagonal.
//   * Up to two problems per launch (ViT + text tower of one layer): the workgroups walk problem 0's items, then
//     problem 1's, in the same persistent launch (mmamd_attention_fwd_grouped).
// Replaces F.scaled_dot_product_attention under nn.MultiheadAttention (reference call sites:
// models/clip/image_encoder.py:108, models/clip/text_encoder.py:121 with is_causal=True).
#include <type_traits>

#include "common.h"

namespace mmamd {

typedef uint32_t __attribute__((address_space(3))) * lds_u32p_r;
typedef __attribute__((ext_vector_type(4))) uint32_t ru32x4;
typedef __attribute__((ext_vector_type(4))) short rs16x4;

struct AttnRingProb {
  const bf16* qkv;
  bf16* out;
  float* lse;
  int S, H, BH, causal;
  int nqt;          // query tiles (32 rows) per item, <= 7
  int rows8;        // staged rows per K / V image: S rounded up to 8
  int G;            // items per round
  int nring;        // ring entries
  int slot_bytes;   // one item: (2 * rows8 + 32 * nqt) * 128
  int entry_bytes;  // G * slot_bytes
  float scale_log2e;
  int lse_stride;   // floats between the log-sum-exp rows of consecutive (batch, head) items: S, or S * S when the rows are parked in the
                    // item's own block of a [B, H, S, S] probability tensor (mmamd_attention_probs_fwd)
  int lse_tile;     // floats between the 32-query groups of a row: 32 (dense), or 32 * S when parked (query q at (q / 32) * 32 S + q % 32:
                    // the first 32 floats of the band its probabilities will fill)
  int pad_;
};
struct AttnRingArgs {
  AttnRingProb p[2];
  int nprob;
};

// 1 KiB LDS-DMA piece: lane l writes 16 B at lds_dst + 16 l from sbase + voff (scalar base + 32-bit per-lane offset: no VALU).
// M0 is written and consumed inside the statement; nothing else in this kernel uses M0.
template <int NT = 0>
__device__ __forceinline__ void ring_dma_piece(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  if constexpr (NT != 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; rounded DOWN to a multiple of 8,
// which only waits for up to 7 more pieces than necessary)
__device__ __forceinline__ void ring_wait_vm_le(int n) {
  if (n >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
  else if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else if (n >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void ring_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint2 ring_tr_b64(const char* p) {
  const rs16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

// bank-swizzle term of a row of a 128-byte-row image: the 16-byte chunk c of row r sits at chunk position c ^ ring_f(r)
__device__ __forceinline__ int ring_f(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

constexpr int kRingWaves = 8;  // 7 compute + 1 loader

// ABL, results right: 4096 / 8192 = O rows stored non-temporal / sc1, 16384 = K / V / Q loaded non-temporal (profiles/r03_cache_policy_ab.txt: non-temporal O
// stores take the isolated kernel from 77 to 67-70 us and cost the out-projection that reads O next just as much: 13.69 vs 13.66 ms per step).
// 2048 (results wrong) = no Q rows in the ring entry, which makes room for a third entry: 58.4 vs 57.7 us without O stores, 66.4 vs 65.6 with - a deeper
// ring does not pay (tools/attn_ring_depth.py).
// ABL (timing experiments, results WRONG): 1 = no K/V/Q DMA, 2 = no key loops, 8 = no O stores, 32 = no exp2, 64 = K / V fragments read once per
// unit instead of per tile, 128 = no MFMA (scores / outputs come from moves), 256 = no cross-half max exchange.  ABL & 4 (results right): phase timers —
// `lse` of problem 0 receives 8 floats of s_memtime ticks per wave: compute waves {key loops, O transpose + store, barrier wait, total},
// loader {DMA issue, landing wait, barrier wait, total}.
// UNR: the key loop of non-causal items with 7 (ViT-B/16, S = 197) or 2 (ViT-B/32, S = 50) key tiles is fully unrolled: tile offsets become
// ds_read immediates and the score registers of consecutive tiles need no copies (~20 of ~100 VALU instructions per tile)
// UNR = 2: the unrolled loop in TWO PHASES per key tile — QK^T of tile k+1 beside the exponentials of tile k, then P.V of tile k beside the row sums
// of tile k and the row maximum / rescale decision of tile k+1 — so that both MFMA groups have independent VALU work to issue between them
// (r03 ablation: matrix 14.6 us, exp2 9.6 us and the other VALU 28 us of the 52 us compute time were purely additive: the compiler had
// scheduled each tile's 8 MFMAs in one cluster and ~65 VALU instructions behind it).  Same arithmetic, same order per row.
template <int ABL, int UNR>
__global__ __launch_bounds__(kRingWaves * 64, 2) void attention_ring_kernel(const AttnRingArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p_r)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

#pragma unroll 1
  for (int pi = 0; pi < args.nprob; ++pi) {
    const bf16* __restrict__ qkv = args.p[pi].qkv;
    bf16* __restrict__ out = args.p[pi].out;
    float* __restrict__ lse = args.p[pi].lse;
    const int S = args.p[pi].S, H = args.p[pi].H, BH = args.p[pi].BH;
    const bool causal = args.p[pi].causal != 0;
    const int nqt = args.p[pi].nqt, rows8 = args.p[pi].rows8, G = args.p[pi].G, nring = args.p[pi].nring;
    const int slot_bytes = args.p[pi].slot_bytes, entry_bytes = args.p[pi].entry_bytes;
    const float scale_log2e = args.p[pi].scale_log2e;
    const int lse_stride = args.p[pi].lse_stride, lse_tile = args.p[pi].lse_tile;
    const int D = H * 64;
    const int rsb = 3 * D * 2;  // bytes between consecutive tokens of qkv
    const int Nloc = (int)blockIdx.x < BH ? (BH - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;  // items of this workgroup
    const int R = (Nloc + G - 1) / G;                                                           // rounds
    if (R == 0) continue;
    constexpr bool TIMED = (ABL & 4) != 0;
    auto now = [&]() -> uint64_t { if constexpr (TIMED) return __builtin_amdgcn_s_memtime(); else return 0; };
    uint64_t tacc[4] = {0, 0, 0, 0};
    const uint64_t t_begin = now();

    if (wave == kRingWaves - 1) {
      // ================================================= loader =================================================
      const int np = rows8 >> 3;            // 1 KiB pieces per K (and per V) image
      const int npq = nqt * 4;              // pieces of the Q image (32 * nqt rows)
      const int nfull = S >> 3;             // pieces whose 8 rows all exist
      constexpr int DNT = (ABL & 16384) ? 1 : 0;  // non-temporal K / V / Q loads
      constexpr bool NOQ = (ABL & 2048) != 0;  // timing experiment: no Q rows in the ring (the compute waves read K rows as their Q)
      const int E1 = 2 * np + (NOQ ? 0 : npq);  // pieces per item
      const int E = G * E1;                 // pieces of a full entry
      const int lr8 = lane >> 3, c8 = lane & 7;
      // full pieces: lane offset relative to the piece's first row; the swizzle term of row 8p + lr8 is ring_f(lr8) ^ (2 (p & 1))
      const uint32_t vo_even = (uint32_t)(lr8 * rsb + ((c8 ^ ring_f(lr8)) << 4));
      const uint32_t vo_odd = vo_even ^ 32u;
      const int step = 8 * rsb;             // bytes between the first rows of consecutive pieces
      auto cum = [&](int e) {  // pieces issued once entries 0..e are out
        return e >= R - 1 ? (R - 1) * E + (Nloc - G * (R - 1)) * E1 : (e + 1) * E;
      };
      int issued = 0, next_e = 0, eslot = 0;
      auto issue_image = [&](const char* sb, uint32_t d0, int npieces) {  // npieces >= nfull pieces of one image (K, V or Q) of one item
        const char* sp = sb;
        uint32_t d = d0;
        int p = 0;
#pragma unroll 1
        for (; p + 1 < nfull; p += 2) {
          ring_dma_piece<DNT>(sp, vo_even, d);
          ring_dma_piece<DNT>(sp + step, vo_odd, d + 1024);
          sp += 2 * step;
          d += 2048;
        }
        if (p < nfull) {
          ring_dma_piece<DNT>(sp, vo_even, d);
          ++p;
        }
#pragma unroll 1
        for (; p < npieces; ++p) {  // rows >= S: the source row is clamped to S - 1 (K / V: never used; Q: as the register-staged kernel did)
          const int row = 8 * p + lr8;
          const int srow = row < S ? row : S - 1;
          ring_dma_piece<DNT>(sb, (uint32_t)(srow * rsb + ((c8 ^ ring_f(row)) << 4)), d0 + (uint32_t)(p * 1024));
        }
      };
      auto issue_entry = [&]() {
        const uint32_t ebase = lds0 + (uint32_t)(eslot * entry_bytes);
        for (int gi = 0; gi < G; ++gi) {
          const int n = G * next_e + gi;
          if (n >= Nloc) break;
          const int item = (int)blockIdx.x + n * (int)gridDim.x;
          const int b = item / H, h = item - b * H;
          const char* src = reinterpret_cast<const char*>(qkv + ((size_t)b * S * 3 * D + (size_t)h * 64));
          const uint32_t dst = ebase + (uint32_t)(gi * slot_bytes);
          if constexpr ((ABL & 1) == 0) {
            if constexpr (!NOQ) issue_image(src, dst + (uint32_t)(2 * rows8 * 128), npq);  // Q (read first by the compute waves)
            issue_image(src + (size_t)D * 2, dst, np);                         // K
            issue_image(src + (size_t)D * 4, dst + (uint32_t)(rows8 * 128), np);  // V
          }
        }
        issued = cum(next_e);
        ++next_e;
        eslot = eslot + 1 == nring ? 0 : eslot + 1;
      };
      while (next_e < R && next_e < nring) issue_entry();
      if constexpr ((ABL & 1) == 0) ring_wait_vm_le(issued - cum(0));  // entry 0 landed
      ring_barrier();
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        const uint64_t t0 = now();
        if (r > 0 && next_e < R && next_e < r + nring) issue_entry();  // the slot of entry r-1 is free since the last barrier
        const uint64_t t1 = now();
        if constexpr ((ABL & 1) == 0)
          if (r + 1 < R) ring_wait_vm_le(issued - cum(r + 1));        // entry r+1 landed before anyone starts round r+1
        const uint64_t t2 = now();
        ring_barrier();
        if constexpr (TIMED) { tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += now() - t2; }
      }
    } else {
      // ================================================= compute =================================================
      // lane constants of the LDS reads (byte offsets inside an image).  K / Q fragment t of a 32-row tile: row l31, chunk 2t + half
      const int fk = ring_f(l31);
      int ko[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) ko[t] = l31 * 128 + (((2 * t + half) ^ fk) << 4);
      // V fragment (jj, nt), halves v0 / v1 (keys +0 / +8): the lane points at row lr (+8), columns nt*32 + 16*((lane>>4)&1) + 4*(lane&3) .. +3
      const int lr = ((lane & 15) >> 2) + 4 * (lane >> 5);
      const int fv = (((lane >> 3) & 1) << 2) | (lane >> 5);
      const int cv = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
      int vo[2][2];  // [nt][v1]
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int v1 = 0; v1 < 2; ++v1) vo[nt][v1] = (lr + 8 * v1) * 128 + (((cv ^ fv) ^ (4 * nt) ^ (2 * v1)) << 4) + 8 * (lane & 1);
      // O tile through the wave's own Q rows: the lane writes row l31, chunks 4nt + g (8 bytes at + 8 half); reads back / stores row
      // (lane >> 3) + 8 j, chunk lane & 7
      const int ow = l31 * 128 + 8 * half;
      const int orr = (lane >> 3) * 128, oc = lane & 7;

      // Static priority for the younger wave of each SIMD (waves 4-6 share SIMDs with waves 0-2): phase timers showed one wave of every pair
      // finishing its 7 key tiles in ~70 k ticks and the other in ~97 k whatever the arbitration; with the raise the pair's total drops 3.7 %
      // (132.9 k vs 138.0 k ticks per launch; alternating the priority per key tile measures the same, 132.6 k).  ABL & 512 switches it off.
      if constexpr ((ABL & 512) == 0) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
      }
      const int di = wave / nqt, qt = wave - di * nqt;  // this wave's item of the round and its query tile (fixed per problem)
      bool wactive = wave < G * nqt;
      // timing experiments (results WRONG): 32768 = the seventh query tile of a 7-tile item (5 valid rows at S = 197) is not computed; 65536 = the second
      // tiles of SIMD 0 / 1 (waves 4, 5) are not computed -- which SIMD sets the round time (r06, profiles/r06_attn_tail_ablation.txt)
      if constexpr ((ABL & 32768) != 0) wactive = wactive && !(nqt == 7 && qt == 6);
      if constexpr ((ABL & 65536) != 0) wactive = wactive && !(nqt == 7 && (qt == 4 || qt == 5));
      const int q = qt * 32 + l31;
      const int kt_end = causal ? qt + 1 : nqt;
      ring_barrier();
      int eslot = 0;
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        const int n = G * r + di;
        const uint64_t tr0 = now();
        uint64_t tr1 = tr0, tr2 = tr0;
        if (wactive && n < Nloc) {
          const char* Kb = smem + eslot * entry_bytes + di * slot_bytes;
          const char* Vb = Kb + rows8 * 128;
          char* Qb = (ABL & 2048) ? const_cast<char*>(Kb) + (qt < 6 ? qt : 5) * 4096
                                  : const_cast<char*>(Vb) + rows8 * 128 + qt * 4096;  // this wave's 32 Q rows; later its O tile
          bf16x8 qcur[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) qcur[t] = *reinterpret_cast<const bf16x8*>(Qb + ko[t]);

          float m = -INFINITY, lsum = 0.f;
          f32x16 ot[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) ot[nt][i] = 0.f;

          auto read_k = [&](int kt, bf16x8 (&kf)[4]) {
            if constexpr ((ABL & 64) != 0) { if (kt > 1) return; }
            const char* kp = Kb + kt * 4096;
#pragma unroll
            for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(kp + ko[t]);
          };
          auto qk = [&](const bf16x8 (&kf)[4]) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if constexpr ((ABL & 128) != 0) { acc[t] += (float)kf[t][0] * (float)qcur[t][0]; }
              else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t], qcur[t], acc, 0, 0, 0);
            }
            return acc;
          };
          bf16x8 kf[4];
          f32x16 st_next;
          auto body = [&](int kt, auto last, auto nc) {  // nc: compile-time tile count of a fully unrolled non-causal item, 0 = run-time loop
            constexpr bool kLast = decltype(last)::value;
            constexpr int NC = decltype(nc)::value;
            const int kend = NC > 0 ? NC : kt_end;
            const bool csl = NC > 0 ? false : causal;
            f32x16 st = st_next;
            bf16x8 vf[2][2];
            const char* vp = Vb + (((ABL & 64) != 0) ? 0 : kt * 4096);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                uint2 v0 = ring_tr_b64(vp + jj * 2048 + vo[nt][0]);
                uint2 v1 = ring_tr_b64(vp + jj * 2048 + vo[nt][1]);
                if constexpr (kLast) {  // keys >= S: their V rows were never staged -> exact zeros
                  const int key0 = kt * 32 + 16 * jj + 4 * half;
                  const uint32_t ma = (key0 + 0 < S ? 0x0000ffffu : 0u) | (key0 + 1 < S ? 0xffff0000u : 0u);
                  const uint32_t mb = (key0 + 2 < S ? 0x0000ffffu : 0u) | (key0 + 3 < S ? 0xffff0000u : 0u);
                  const uint32_t mc = (key0 + 8 < S ? 0x0000ffffu : 0u) | (key0 + 9 < S ? 0xffff0000u : 0u);
                  const uint32_t md = (key0 + 10 < S ? 0x0000ffffu : 0u) | (key0 + 11 < S ? 0xffff0000u : 0u);
                  v0.x &= ma; v0.y &= mb; v1.x &= mc; v1.y &= md;
                }
                ru32x4 vw;
                vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
                vf[jj][nt] = __builtin_bit_cast(bf16x8, vw);
              }
            if constexpr (!kLast) {
              st_next = qk(kf);
              read_k(kt + 2 < kend ? kt + 2 : kend - 1, kf);  // (the clamped re-read of the last tile is never used)
            } else {  // only the last tile of a unit holds padded keys (non-causal) or keys above the diagonal (causal)
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (key >= S || (csl && key > q)) st[i] = -INFINITY;
              }
            }
            float tmax = st[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, st[i]);
            // (v_permlane32_swap instead of this ds_bpermute measured the same, 73.4 vs 74.0 us; hipcc folds a swap of a register with itself away)
            if constexpr ((ABL & 256) == 0) tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float ts = tmax * scale_log2e;
            if (__any(ts > m + 8.0f)) {  // deferred max: the reference moves only when some row grew by more than 2^8
              const float m_new = fmaxf(m, ts);
              const float alpha = __builtin_amdgcn_exp2f(m - m_new);
              m = m_new;
              lsum *= alpha;
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[nt][i] *= alpha;
            }
            uint32_t pk[8];
            f32x2 ps2 = {0.f, 0.f};
            const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m, -m};
#pragma unroll
            for (int g = 0; g < 8; ++g) {
              f32x2 a = {st[2 * g], st[2 * g + 1]};
              a = __builtin_elementwise_fma(a, sc2, nm2);
              f32x2 e;
              if constexpr ((ABL & 32) != 0) { e = a; }
              else {
              e[0] = __builtin_amdgcn_exp2f(a[0]);
              e[1] = __builtin_amdgcn_exp2f(a[1]);
              }
              ps2 += e;
              bf16x2 p;
              p[0] = (bf16)e[0]; p[1] = (bf16)e[1];
              pk[g] = __builtin_bit_cast(uint32_t, p);
            }
            lsum += ps2[0] + ps2[1];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              ru32x4 pw;
              pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
              const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                if constexpr ((ABL & 128) != 0) { ot[nt][jj] += (float)vf[jj][nt][0] * (float)pf[0] + (float)vf[jj][nt][7] * (float)pf[7]; }
                else ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jj][nt], pf, ot[nt], 0, 0, 0);
              }
            }
          };
          auto tiles_phased = [&](auto nc) {
            constexpr int N = decltype(nc)::value;
            auto mask_last = [&](f32x16& sc) {  // the last tile holds the padded keys
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int key = (N - 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (key >= S) sc[i] = -INFINITY;
              }
            };
            auto row_max_scaled = [&](const f32x16& sc) {
              float tmax = sc[0];
#pragma unroll
              for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, sc[i]);
              tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
              return tmax * scale_log2e;
            };
            read_k(0, kf);
            f32x16 st = qk(kf);
            if constexpr (N > 1) read_k(1, kf);
            if constexpr (N == 1) mask_last(st);
            {  // tile 0: the reference starts at -inf, so the first tile always sets it (O and the row sum are still zero)
              const float ts = row_max_scaled(st);
              m = fmaxf(m, ts);
            }
#pragma unroll
            for (int kt = 0; kt < N; ++kt) {
              if constexpr ((ABL & 1024) != 0) {  // the two waves of a SIMD take turns at priority 1, one key tile each
                if ((wave >= 4) == ((kt & 1) != 0)) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
              }
              bf16x8 vf[2][2];
              const char* vp = Vb + kt * 4096;
#pragma unroll
              for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                  uint2 v0 = ring_tr_b64(vp + jj * 2048 + vo[nt][0]);
                  uint2 v1 = ring_tr_b64(vp + jj * 2048 + vo[nt][1]);
                  if (kt == N - 1) {  // keys >= S: their V rows were never staged -> exact zeros
                    const int key0 = kt * 32 + 16 * jj + 4 * half;
                    const uint32_t ma = (key0 + 0 < S ? 0x0000ffffu : 0u) | (key0 + 1 < S ? 0xffff0000u : 0u);
                    const uint32_t mb = (key0 + 2 < S ? 0x0000ffffu : 0u) | (key0 + 3 < S ? 0xffff0000u : 0u);
                    const uint32_t mc = (key0 + 8 < S ? 0x0000ffffu : 0u) | (key0 + 9 < S ? 0xffff0000u : 0u);
                    const uint32_t md = (key0 + 10 < S ? 0x0000ffffu : 0u) | (key0 + 11 < S ? 0xffff0000u : 0u);
                    v0.x &= ma; v0.y &= mb; v1.x &= mc; v1.y &= md;
                  }
                  ru32x4 vw;
                  vw[0] = v0.x; vw[1] = v0.y; vw[2] = v1.x; vw[3] = v1.y;
                  vf[jj][nt] = __builtin_bit_cast(bf16x8, vw);
                }
              // ---- phase 1: QK^T of tile kt+1 (matrix pipe) beside the exponentials of tile kt (VALU)
              f32x16 stn;
              if (kt + 1 < N) {
                stn = qk(kf);
                if (kt + 2 < N) read_k(kt + 2, kf);
              }
              uint32_t pk[8];
              f32x2 ev[8];
              const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m, -m};
#pragma unroll
              for (int g = 0; g < 8; ++g) {
                f32x2 a = {st[2 * g], st[2 * g + 1]};
                a = __builtin_elementwise_fma(a, sc2, nm2);
                ev[g][0] = __builtin_amdgcn_exp2f(a[0]);
                ev[g][1] = __builtin_amdgcn_exp2f(a[1]);
                bf16x2 p;
                p[0] = (bf16)ev[g][0]; p[1] = (bf16)ev[g][1];
                pk[g] = __builtin_bit_cast(uint32_t, p);
              }
              // ---- phase 2: P.V of tile kt (matrix pipe) beside the row sums of tile kt and the row maximum of tile kt+1 (VALU)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                ru32x4 pw;
                pw[0] = pk[4 * jj + 0]; pw[1] = pk[4 * jj + 1]; pw[2] = pk[4 * jj + 2]; pw[3] = pk[4 * jj + 3];
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) ot[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jj][nt], pf, ot[nt], 0, 0, 0);
              }
              f32x2 ps2 = {0.f, 0.f};
#pragma unroll
              for (int g = 0; g < 8; ++g) ps2 += ev[g];
              lsum += ps2[0] + ps2[1];
              if (kt + 1 < N) {
                if (kt + 1 == N - 1) mask_last(stn);
                const float ts = row_max_scaled(stn);
                if constexpr (UNR == 3) {  // interleave hints: one MFMA, then its share of the VALU work, per phase
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                  }
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                  }
                }
                if (__any(ts > m + 8.0f)) {  // deferred max (after P.V of tile kt: O is rescaled)
                  const float m_new = fmaxf(m, ts);
                  const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                  m = m_new;
                  lsum *= alpha;
#pragma unroll
                  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[nt][i] *= alpha;
                }
                st = stn;
              }
            }
          };
          auto tiles_unrolled = [&](auto nc) {
            constexpr int N = decltype(nc)::value;
            read_k(0, kf);
            st_next = qk(kf);
            if constexpr (N > 1) read_k(1, kf);
#pragma unroll
            for (int kt = 0; kt < N - 1; ++kt) body(kt, std::false_type{}, nc);
            body(N - 1, std::true_type{}, nc);
          };
          if constexpr ((ABL & 2) == 0) {
            if (UNR >= 2 && !causal && nqt == 7) {
              tiles_phased(std::integral_constant<int, 7>{});
            } else if (UNR == 1 && !causal && nqt == 7) {
              tiles_unrolled(std::integral_constant<int, 7>{});
            } else if (UNR && !causal && nqt == 2) {
              tiles_unrolled(std::integral_constant<int, 2>{});
            } else {
              read_k(0, kf);
              st_next = qk(kf);
              if (kt_end > 1) read_k(1, kf);
#pragma unroll 1
              for (int kt = 0; kt < kt_end - 1; ++kt) body(kt, std::false_type{}, std::integral_constant<int, 0>{});
              body(kt_end - 1, std::true_type{}, std::integral_constant<int, 0>{});
            }
          } else {
            lsum = 1.f + (float)qcur[0][0];
          }

          // ---- normalise, transpose the O tile through the wave's Q rows, store whole rows
          tr1 = now();
          lsum += __shfl_xor(lsum, 32);
          const float inv = 1.0f / lsum;
          const int item = (int)blockIdx.x + n * (int)gridDim.x;
          const int b = item / H, h = item - b * H;
          if (!TIMED && lse != nullptr && half == 0 && q < S) lse[(size_t)item * lse_stride + (size_t)qt * lse_tile + l31] = m + __builtin_amdgcn_logf(lsum);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              bf16x4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = (bf16)(ot[nt][4 * g + j] * inv);
              *reinterpret_cast<bf16x4*>(Qb + ow + (((4 * nt + g) ^ fk) << 4)) = o;
            }
          if constexpr ((ABL & 8) == 0) {
            bf16* obase = out + ((size_t)b * S + qt * 32) * D + h * 64 + oc * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int row = (lane >> 3) + 8 * j;
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(Qb + orr + j * 1024 + ((oc ^ ring_f(row)) << 4));
              if (qt * 32 + row < S) {
                if constexpr ((ABL & 4096) != 0) {
                  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
                  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(obase + (size_t)row * D), "v"(__builtin_bit_cast(u32x4_t, v)) : "memory");
                } else if constexpr ((ABL & 8192) != 0) {
                  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
                  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(obase + (size_t)row * D), "v"(__builtin_bit_cast(u32x4_t, v)) : "memory");
                } else {
                  *reinterpret_cast<bf16x8*>(obase + (size_t)row * D) = v;
                }
              }
            }
          }
          tr2 = now();
        }
        ring_barrier();
        if constexpr (TIMED) { tacc[0] += tr1 - tr0; tacc[1] += tr2 - tr1; tacc[2] += now() - tr2; }
        eslot = eslot + 1 == nring ? 0 : eslot + 1;
      }
    }
    if constexpr (TIMED) {
      tacc[3] = now() - t_begin;
      if (pi == 0 && lane == 0 && lse != nullptr)
        for (int i = 0; i < 4; ++i) lse[((size_t)blockIdx.x * kRingWaves + wave) * 8 + i] = (float)tacc[i];
    }
  }
}

// host side ----------------------------------------------------------------------------------------------------------------
int g_attn_ring_abl = 0;
int g_attn_ring_depth_cap = 0;  // mmamd_debug_set_attn_variant(3000 + n): at most n ring entries (0 = as many as fit)
static bool ring_prob_setup(AttnRingProb& p, const void* qkv, void* out, float* lse, int B, int S, int H, int causal, float scale, int& smem,
                            int lse_stride) {
  const int nqt = (S + 31) / 32;
  if (nqt > 7 || S < 1) return false;
  const int rows8 = (S + 7) & ~7;
  p.qkv = (const bf16*)qkv; p.out = (bf16*)out; p.lse = lse;
  p.S = S; p.H = H; p.BH = B * H; p.causal = causal ? 1 : 0;
  p.nqt = nqt; p.rows8 = rows8;
  p.G = 7 / nqt;
  p.slot_bytes = (2 * rows8 + 32 * nqt) * 128;  // K rows, V rows, Q rows (the V over-read of the last key tile ends inside the Q rows)
#ifdef MMAMD_EXPERIMENTS
  if (g_attn_ring_abl == 400 || g_attn_ring_abl == 401) p.slot_bytes = 2 * rows8 * 128;  // "no Q rows" timing experiment
#endif
  p.entry_bytes = p.G * p.slot_bytes;
  int nring = (160 * 1024) / p.entry_bytes;
  if (nring > 4) nring = 4;
  if (g_attn_ring_depth_cap >= 2 && nring > g_attn_ring_depth_cap) nring = g_attn_ring_depth_cap;  // A/B of the ring depth (tools/attn_ring_depth.py)
  if (nring < 2) return false;
  p.nring = nring;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.lse_stride = lse_stride > 0 ? lse_stride : S;
  p.lse_tile = lse_stride > 0 ? 32 * S : 32;  // (a stride is only ever given for the parked layout)
  p.pad_ = 0;
  int need = nring * p.entry_bytes;
#ifdef MMAMD_EXPERIMENTS
  if (g_attn_ring_abl == 400 || g_attn_ring_abl == 401) need += 4096;  // the V over-read of the last entry
#endif
  if (need > smem) smem = need;
  return true;
}

// true if mmamd_attention_fwd's shape is served by the ring kernel: at most 7 query tiles and two ring entries in 160 KiB (S <= 208)
bool attn_ring_supports(int S) {
  if (S < 1 || (S + 31) / 32 > 7) return false;
  const int nqt = (S + 31) / 32, rows8 = (S + 7) & ~7;
  return 2 * (7 / nqt) * (2 * rows8 + 32 * nqt) * 128 <= 160 * 1024;
}
// (defined above ring_prob_setup) timing experiments: ABL bits of the kernel (1, 2, 8 and their sums), + 16 = runtime key loop for every shape

template <int ABL, int UNR>
static int launch_ring_t(const AttnRingArgs& a, int grid, int smem, hipStream_t st) {
  static unsigned long long attr_mask = 0;  // per-device one-time opt-in to 160 KiB of dynamic LDS (the attribute is a maximum)
  auto kern = attention_ring_kernel<ABL, UNR>;
  if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_mask)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kRingWaves * 64), smem, st, a);
  return launch_status("attention_ring");
}

int launch_attn_ring(const void* const* qkv, void* const* out, float* const* lse, const int* B, const int* S, const int* H, const int* causal,
                     const float* scale, int nprob, hipStream_t st, const int* lse_stride) {
  AttnRingArgs a;
  a.nprob = 0;
  int smem = 0, maxbh = 0;
  for (int i = 0; i < nprob; ++i) {
    if (B[i] == 0) continue;
    if (!ring_prob_setup(a.p[a.nprob], qkv[i], out[i], lse ? lse[i] : nullptr, B[i], S[i], H[i], causal[i], scale[i], smem,
                         lse_stride ? lse_stride[i] : 0)) {
      set_error("attention_ring: S=%d is not served by the ring kernel", S[i]);
      return MMAMD_E_UNSUPPORTED;
    }
    if (a.p[a.nprob].BH > maxbh) maxbh = a.p[a.nprob].BH;
    ++a.nprob;
  }
  if (a.nprob == 0) return 0;
  const int cus = stream_cus(st);
  const int grid = maxbh < cus ? maxbh : cus;
#ifdef MMAMD_EXPERIMENTS  // ablations for tools/attn_ring_ablate.py (python -m multimodal_amd.build with MMAMD_EXPERIMENTS=1)
  switch (g_attn_ring_abl) {
    case 1: return launch_ring_t<1, 1>(a, grid, smem, st);
    case 2: return launch_ring_t<2, 1>(a, grid, smem, st);
    case 3: return launch_ring_t<3, 1>(a, grid, smem, st);
    case 8: return launch_ring_t<8, 1>(a, grid, smem, st);
    case 9: return launch_ring_t<9, 1>(a, grid, smem, st);
    case 10: return launch_ring_t<10, 1>(a, grid, smem, st);
    case 11: return launch_ring_t<11, 1>(a, grid, smem, st);
    case 4: return launch_ring_t<4, 1>(a, grid, smem, st);
    case 300: return launch_ring_t<0, 1>(a, grid, smem, st);
    case 301: return launch_ring_t<0, 3>(a, grid, smem, st);
    case 309: return launch_ring_t<9, 2>(a, grid, smem, st);
    case 310: return launch_ring_t<9, 3>(a, grid, smem, st);
    case 304: return launch_ring_t<4, 2>(a, grid, smem, st);
    case 302: return launch_ring_t<512, 2>(a, grid, smem, st);
    case 303: return launch_ring_t<1024, 2>(a, grid, smem, st);
    case 305: return launch_ring_t<512 + 4, 2>(a, grid, smem, st);
    case 306: return launch_ring_t<1024 + 4, 2>(a, grid, smem, st);
    case 41: return launch_ring_t<9 + 32, 1>(a, grid, smem, st);
    case 73: return launch_ring_t<9 + 64, 1>(a, grid, smem, st);
    case 137: return launch_ring_t<9 + 128, 1>(a, grid, smem, st);
    case 265: return launch_ring_t<9 + 256, 1>(a, grid, smem, st);
    case 105: return launch_ring_t<9 + 32 + 64, 1>(a, grid, smem, st);
    case 233: return launch_ring_t<9 + 32 + 64 + 128, 1>(a, grid, smem, st);
    case 201: return launch_ring_t<9 + 64 + 128, 1>(a, grid, smem, st);
    case 320: return launch_ring_t<32768, 2>(a, grid, smem, st);
    case 321: return launch_ring_t<65536, 2>(a, grid, smem, st);
    case 322: return launch_ring_t<32768 + 65536, 2>(a, grid, smem, st);
    case 410: return launch_ring_t<4096, 2>(a, grid, smem, st);
    case 412: return launch_ring_t<16384, 2>(a, grid, smem, st);
    case 413: return launch_ring_t<4096 + 16384, 2>(a, grid, smem, st);
    case 411: return launch_ring_t<8192, 2>(a, grid, smem, st);
    case 400: return launch_ring_t<2048 + 8, 2>(a, grid, smem, st);
    case 401: return launch_ring_t<2048, 2>(a, grid, smem, st);
    case 402: return launch_ring_t<8, 2>(a, grid, smem, st);
    case 16: return launch_ring_t<0, 0>(a, grid, smem, st);
    case 17: return launch_ring_t<1, 0>(a, grid, smem, st);
    case 25: return launch_ring_t<9, 0>(a, grid, smem, st);
  }
#endif
  return launch_ring_t<0, 2>(a, grid, smem, st);
}

}  // namespace mmamd
