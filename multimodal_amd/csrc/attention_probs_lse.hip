// attention_probs_lse.hip — the attention PROBABILITIES FLAVA's encoders hand out (reference modules/layers/attention.py:220-239 returns `attn`,
// models/flava/image_encoder.py:217-222 always asks for it) as a pass of their own behind the flash forward:
//     P[b, h, q, k] = exp2(scale * log2(e) * q.k - lse[b, h, q])          (lse: the log2-domain log-sum-exp the flash kernel saved)
// One pass (QK^T once, no row statistics), fp32, no key-padding mask.  The job is the STORE: [B, H, S, S] fp32 is 477 MB per launch at B = 256, S = 197, and a
// row of 197 floats (788 B) is only 4-byte aligned, so the 128-byte row segments of a 32 x 32 MFMA tile never cover a cache line — the two-pass kernel
// (attention.hip, attention_probs_kernel) spends ~150 us on segment stores that a plain fill of the same bytes does in 70 (profiles/r04_attn_probs_ablation.txt).
// Here the 32 rows x S keys of a query tile — CONTIGUOUS in memory, 32 * S floats — are assembled in LDS and streamed out as the linear memory image:
//   * one workgroup (8 waves) per (batch, head); the head's Q rows staged once in LDS (128-byte rows, the ring kernel's bank swizzle), wave w's K
//     fragments of key tile w in registers for the whole head;
//   * per query tile, wave w computes key tile w (4 MFMAs, 16 exp2 per lane) and writes its 32 x 32 block into the BAND image with four ds_write_b128:
//     band rows are padded to `pitch` floats, pitch a multiple of 4 with pitch / 4 odd, so the eight rows of a write group tile all 32 banks;
//   * after one barrier all waves stream the band: a wave-instruction stores 64 CONSECUTIVE floats of the linear image, 256-byte aligned in memory
//     (global_store_dword: two whole cache lines; only the first and last segment of a band are partial), read back from LDS with ds_read_b32
//     (consecutive lanes -> consecutive banks; the row pad shifts the address by pitch - S at a row seam);
//   * two band buffers: the next tile's blocks are written while slower waves still stream the previous band (one barrier per tile).
// The log-sum-exp arrives PARKED in the head's own block of the probability tensor (its first S floats; the flash kernels take a row stride): no
// workspace in the C-ABI, and only this workgroup ever touches the block — it reads the S values into LDS before its first store.
#include "common.h"

namespace mmamd {

struct ProbsLseArgs {
  const bf16* qkv;
  float* probs;
  int S, H;
  float scale_log2e;
  int pitch;       // floats per band row in LDS
  int rows_q;      // staged Q rows: S rounded up to 8
  uint32_t magic;  // ceil(2^32 / S): row of a linear band index g is umulhi(g, magic) (exact for g < 32 S)
};

__device__ __forceinline__ int pl_f(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }  // attention_ring.hip's ring_f

// NO global load may be pending inside the query-tile loop: vmcnt counts loads and stores together and they do not retire in order with each other,
// so waiting for a load means waiting for every store before it -- the first form of this kernel prefetched the next tile's Q fragments from memory
// and drained the previous band's 25 KB of stores at every tile (4.7 us per band, 198 us per launch at B = 256, S = 197;
// profiles/r05_probs_lse_bench.txt).  Everything the loop reads is on chip before it starts: wave w keeps the K fragments of ITS key tile(s) in
// registers for the whole head (loaded once, fragment-shaped, straight from memory), the head's Q rows sit in LDS (128-byte rows, the ring kernel's
// bank swizzle), the log-sum-exp too; the loop is LDS reads, MFMA, exp2, LDS writes, one barrier and stores that nobody waits for.
template <int NKT>
__global__ __launch_bounds__(512, 2) void attention_probs_lse_kernel(const ProbsLseArgs a) {
  constexpr int KPW = (NKT + 7) / 8;  // key tiles per wave (wave w: tiles w, w + 8)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = a.S, H = a.H, pitch = a.pitch, rows_q = a.rows_q;
  char* Qimg = smem;                                                        // [rows_q][128 B], chunk c of row r at position c ^ pl_f(r)
  float* Ls = reinterpret_cast<float*>(smem + rows_q * 128);                // [32 NKT] log-sum-exp per query (0 past S)
  float* band0 = reinterpret_cast<float*>(smem + rows_q * 128 + NKT * 128);  // 2 x [32][pitch]
  const int band_floats = 32 * pitch;

  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const size_t row_stride = (size_t)3 * D;
  const bf16* base = a.qkv + (size_t)b * S * row_stride + h * 64;
  float* phead = a.probs + (size_t)bh * S * S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // ---- everything the head needs from memory, once: Q rows -> LDS, this wave's K fragments -> registers, the parked log-sum-exp -> LDS
  for (int r = tid >> 3; r < rows_q; r += 64) {
    const int c = tid & 7;
    const int rs = r < S ? r : S - 1;  // (rows past S repeat row S - 1: finite, their band rows are never streamed)
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(base + (size_t)rs * row_stride + c * 8);
    *reinterpret_cast<bf16x8*>(Qimg + r * 128 + ((c ^ pl_f(r)) << 4)) = qv;
  }
  bf16x8 kf[KPW][4];  // K fragment t of key tile kt: row kt*32 + l31, elements 16 t + 8 half .. + 7 (keys past S repeat key S - 1: their columns are never stored)
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int key = (wave + 8 * i) * 32 + l31;
    const bf16* kp = base + (size_t)(key < S ? key : S - 1) * row_stride + D + 8 * half;
#pragma unroll
    for (int t = 0; t < 4; ++t) kf[i][t] = *reinterpret_cast<const bf16x8*>(kp + 16 * t);
  }
  for (int q = tid; q < NKT * 32; q += 512) Ls[q] = q < S ? phead[q] : 0.f;
  // lane constants: Q fragment t of a 32-row tile = row l31, chunk 2t + half (swizzled)
  const int fq = pl_f(l31);
  int qo[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qo[t] = l31 * 128 + (((2 * t + half) ^ fq) << 4);
  const int nqt = (S + 31) >> 5;
  const uint32_t pad = (uint32_t)(pitch - S);
  const uint64_t head_abs = (uint64_t)(reinterpret_cast<uintptr_t>(phead) >> 2);  // absolute float index of the head's block
  __syncthreads();

#pragma unroll 1
  for (int qt = 0; qt < nqt; ++qt) {
    float* band = band0 + (qt & 1) * band_floats;
    const float L = Ls[qt * 32 + l31];
    // the last query tile reads Q rows up to 32 nqt - 1 >= rows_q: they land in Ls / the band buffers (junk, possibly changing under the read) --
    // only band rows >= S depend on them, and those are never streamed
    const char* qp = Qimg + qt * 4096;
    bf16x8 qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(qp + qo[t]);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int kt = wave + 8 * i;
      if (kt < NKT) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][t], qf[t], acc, 0, 0, 0);
        // acc[4g + j]: key kt*32 + 8g + 4 half + j of query qt*32 + l31
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 e;
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[4 * g + j], a.scale_log2e, -L));
          const int col = kt * 32 + 8 * g + 4 * half;
          if (col < pitch) *reinterpret_cast<f32x4*>(band + l31 * pitch + col) = e;  // (columns S .. pitch - 1 hold junk that is never streamed)
        }
      }
    }
    __syncthreads();  // the band is complete; everyone has finished streaming the band before the last one (the buffer written next)
    // ---- stream: linear image of min(32, S - 32 qt) rows x S floats, in 64-float segments aligned to 256 B in memory
    const int n = (S - qt * 32 < 32 ? S - qt * 32 : 32) * S;
    const int ph = (int)((head_abs + (uint64_t)qt * 32u * (uint64_t)S) & 63u);  // floats between the segment grid and the band's first element
    float* sbase = phead + (size_t)qt * 32 * S - ph;                            // segment 0 starts here (its first `ph` lanes belong to the band before)
    const int nseg = (ph + n + 63) >> 6;
    // Segment s (wave-uniform) starts at linear index g0 = 64 s - ph; with S >= 64 it crosses at most one row seam, so the row of its first element,
    // the seam lane and the store address are SCALAR work and a lane pays three VALU operations for its LDS address (the first form computed a
    // per-lane umulhi / mul / 64-bit address per element: ~250 VALU per wave and band, which alone was ~90 us of SIMD time per launch)
    const int lane_pad = lane;
#pragma unroll 1
    for (int s0 = wave; s0 < nseg; s0 += 32) {
      float v[4];
      bool ok[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sg = s0 + 8 * i;
        const int g0 = sg * 64 - ph;                                  // < 0 only for segment 0
        const uint32_t row0 = __umulhi((uint32_t)(g0 > 0 ? g0 : 0), a.magic);
        const int seam = (int)(row0 + 1u) * S - g0;                   // first lane that belongs to the next row (>= 64: none)
        const int g = g0 + lane_pad;
        ok[i] = sg < nseg && (uint32_t)g < (uint32_t)n;
        const uint32_t idx = (uint32_t)(g + (int)(row0 * pad)) + (lane_pad >= seam ? pad : 0u);
        v[i] = band[ok[i] ? idx : 0u];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* seg = sbase + (size_t)(s0 + 8 * i) * 64;               // wave-uniform pointer: scalar base + the lane's 4-byte offset
        if (ok[i]) seg[lane] = v[i];
      }
    }
  }
}

// pitch: S rounded up to 4 floats, + 4 when pitch / 4 is even (ds_write_b128 is serviced in groups of 8 consecutive lanes = 8 band rows at one column:
// with pitch = 4 (mod 8) their 16-byte slots tile the 32 banks)
static int probs_lse_pitch(int S) {
  int p = (S + 3) & ~3;
  if (((p >> 2) & 1) == 0) p += 4;
  return p;
}
static int probs_lse_smem(int S) {
  const int nkt = (S + 31) / 32, rows_q = (S + 7) & ~7;
  return rows_q * 128 + nkt * 128 + 2 * 32 * probs_lse_pitch(S) * 4;
}

// S >= 64: a 64-float store segment then crosses at most one row seam (the streaming loop's scalar row arithmetic); shorter sequences keep the two-pass kernel
bool attn_probs_lse_supports(int S) { return S >= 64 && S <= 288 && probs_lse_smem(S) <= 160 * 1024; }

template <int NKT>
static int launch_probs_lse_t(const ProbsLseArgs& a, int BH, int smem, hipStream_t st) {
  static unsigned long long attr_mask = 0;
  auto kern = attention_probs_lse_kernel<NKT>;
  if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc;
  hipLaunchKernelGGL(kern, dim3(BH), dim3(512), smem, st, a);
  return launch_status("attention_probs_lse");
}

// probs [B, H, S, S] fp32; on entry the first S floats of every (b, h) block hold that head's log2-domain log-sum-exp
int launch_attn_probs_lse(const void* qkv, float* probs, int B, int S, int H, float scale, hipStream_t st) {
  if (!attn_probs_lse_supports(S)) {
    set_error("attention_probs_lse: S=%d is not served", S);
    return MMAMD_E_UNSUPPORTED;
  }
  ProbsLseArgs a;
  a.qkv = (const bf16*)qkv; a.probs = probs; a.S = S; a.H = H;
  a.scale_log2e = scale * 1.4426950408889634f;
  a.pitch = probs_lse_pitch(S);
  a.rows_q = (S + 7) & ~7;
  a.magic = (uint32_t)(((1ull << 32) + (uint64_t)S - 1) / (uint64_t)S);
  const int smem = probs_lse_smem(S);
  switch ((S + 31) / 32) {
    case 1: return launch_probs_lse_t<1>(a, B * H, smem, st);
    case 2: return launch_probs_lse_t<2>(a, B * H, smem, st);
    case 3: return launch_probs_lse_t<3>(a, B * H, smem, st);
    case 4: return launch_probs_lse_t<4>(a, B * H, smem, st);
    case 5: return launch_probs_lse_t<5>(a, B * H, smem, st);
    case 6: return launch_probs_lse_t<6>(a, B * H, smem, st);
    case 7: return launch_probs_lse_t<7>(a, B * H, smem, st);
    case 8: return launch_probs_lse_t<8>(a, B * H, smem, st);
    case 9: return launch_probs_lse_t<9>(a, B * H, smem, st);
  }
  set_error("attention_probs_lse: unsupported S=%d", S);
  return MMAMD_E_UNSUPPORTED;
}

}  // namespace mmamd
