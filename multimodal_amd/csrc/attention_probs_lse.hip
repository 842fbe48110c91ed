// attention_probs_lse.hip — the attention PROBABILITIES FLAVA's encoders hand out (reference modules/layers/attention.py:220-239 returns `attn`,
// models/flava/image_encoder.py:217-222 always asks for it) as a pass of their own behind the flash forward:
//     P[b, h, q, k] = exp2(scale * log2(e) * q.k - lse[b, h, q])          (lse: the log2-domain log-sum-exp the flash kernel saved)
// One pass (QK^T once, no row statistics), fp32, no key-padding mask.  The job is the STORE: [B, H, S, S] fp32 is 477 MB per launch at B = 256, S = 197, and a
// row of 197 floats (788 B) is only 4-byte aligned, so the 128-byte row segments of a 32 x 32 MFMA tile never cover a cache line — the two-pass kernel
// (attention.hip, attention_probs_kernel) spends ~150 us on segment stores that a plain fill of the same bytes does in 70 (profiles/r04_attn_probs_ablation.txt).
// Here the 32 rows x S keys of a query tile — a BAND, contiguous in memory, 32 * S floats — are assembled in LDS AS THE LINEAR MEMORY IMAGE and streamed out:
//   * one workgroup (8 waves) per (batch, head); the head's Q rows staged once in LDS (128-byte rows, the ring kernel's bank swizzle), wave w's K
//     fragments of key tile w in registers for the whole head, the head's log-sum-exp in LDS: NO global load is pending inside the query-tile loop
//     (vmcnt counts loads and stores together and they do not retire in order with each other, so waiting for a load means waiting for every store
//     issued before it);
//   * per query tile, wave w computes key tile w (4 MFMAs, 16 exp2 per lane) and writes its 32 x 32 block into the band image: float (row r, key k) at
//     band[shift + r S + k] — no row padding, the image IS what memory will hold — with ds_write_b32 (the 32 lanes of a write group are 32 rows of
//     one column: banks (r S + c) mod 32, conflict-free for odd S, gcd(S, 32)-way otherwise; S % 8 == 0 keeps the two-pass kernel);
//   * `shift` = (address of the band's first float / 4) mod 4: a 16-byte aligned chunk of MEMORY is then a 16-byte aligned chunk of LDS, and the
//     streaming phase is ds_read_b128 -> global_store_dwordx4, lane l of segment s moving chunk 64 s + l: 1 KB = eight whole cache lines per
//     wave-instruction, no per-element address arithmetic.  Only the two chunks that straddle the band's ends are stored float by float;
//   * two band buffers: the next tile's blocks are written while slower waves still stream the previous band (one barrier per tile).
// How it got here (profiles/r05_probs_lse_*.txt, B = 256, S = 197, the kernel alone; a fill_ of the bytes: 70 us):
//   v1  198 us  the next tile's Q fragments prefetched from memory inside the loop: every tile drained the previous band's stores (the vmcnt rule above);
//   v2  160 us  everything on chip before the loop, but 256-byte segments of dword stores out of a row-PADDED image: ~22 instructions per segment and
//               wave, 83 us of instruction issue on their own;
//   v3  135 us  this form.  Ablations: stores alone 112 us, compute + set-up alone 51 us, LDS reads of the streaming phase ~0;
//   v4  170 us  a workgroup per BAND (29 KB of LDS, four workgroups = 32 waves per CU, a head's bands on one XCD): the stores alone 105 us, but every band
//               now waits for its own K / Q loads behind the chip's store traffic (compute + set-up 70 us) and the two do not overlap.  Dropped.
//   tools/microbench/store_pattern.hip: bare dwordx4 stores in this kernel's address pattern take 87 us from 32 waves per CU (a grid-stride fill: 97 us);
//   the address pattern is not what separates the kernel from the fill.
// The log-sum-exp arrives PARKED in the probability tensor itself — query q of a head in float (q / 32) * 32 S + q % 32 of the head's block, the first 32
// floats of the band its probabilities will fill (the flash kernels take the strides): no workspace in the C-ABI, and only this workgroup ever touches the
// block — it reads the values into LDS before its first store.  Or as a dense [B, H, S] array a training forward saved (mmamd_attention_probs_from_lse).
#include "common.h"

namespace mmamd {

struct ProbsLseArgs {
  const bf16* qkv;
  float* probs;
  const float* lse;  // log2-domain log-sum-exp: query q of head bh at lse[bh * lse_stride + (q / 32) * lse_tile + q % 32] -- the probability tensor itself
  int lse_stride;    // (stride S * S, tile 32 S: parked by the flash forward in the first 32 floats of each band) or a dense [B, H, S] array saved by
  int lse_tile;      // a training forward (stride S, tile 32)
  int S, H;
  float scale_log2e;
  int rows_q;        // staged Q rows: S rounded up to 8
  int band_bytes;    // one band buffer: 32 S floats + the shift + one dump slot, rounded up to 16 B
};

__device__ __forceinline__ int pl_f(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }  // attention_ring.hip's ring_f

// ABL (timing experiments, results WRONG; built only with MMAMD_EXPERIMENTS=1, tools/probs_lse_ablate.py): 1 = no global stores, 2 = no compute phase (no Q reads,
// MFMA, exp2, band writes), 4 = no LDS reads in the streaming phase (a constant is stored), 8 = no barrier
// NW: waves per workgroup.  The LDS image allows two workgroups per CU either way; 16 waves per workgroup = 32 waves per CU in the streaming phase
// (stores in flight are what the store rate follows: tools/microbench/store_pattern.hip, profiles/r05_probs_lse_ablation_v3.txt) while waves >= NKT
// sit out the compute phase.
template <int NKT, int ABL = 0, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 16 ? 8 : 2) void attention_probs_lse_kernel(const ProbsLseArgs a) {
  constexpr int KPW = (NKT + NW - 1) / NW;  // key tiles per wave (wave w: tiles w, w + NW)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = a.S, H = a.H, rows_q = a.rows_q;
  char* Qimg = smem;                                            // [rows_q][128 B], chunk c of row r at position c ^ pl_f(r)
  float* Ls = reinterpret_cast<float*>(smem + rows_q * 128);    // [32 NKT] log-sum-exp per query (0 past S)
  char* band0 = smem + rows_q * 128 + NKT * 128;                // 2 band buffers of band_bytes

  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * 64;
  const size_t row_stride = (size_t)3 * D;
  const bf16* base = a.qkv + (size_t)b * S * row_stride + h * 64;
  float* phead = a.probs + (size_t)bh * S * S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // ---- everything the head needs from memory, once: Q rows -> LDS, this wave's K fragments -> registers, the log-sum-exp -> LDS
  for (int r = tid >> 3; r < rows_q; r += NW * 8) {
    const int c = tid & 7;
    const int rs = r < S ? r : S - 1;  // (rows past S repeat row S - 1: finite, their band rows are never streamed)
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(base + (size_t)rs * row_stride + c * 8);
    *reinterpret_cast<bf16x8*>(Qimg + r * 128 + ((c ^ pl_f(r)) << 4)) = qv;
  }
  bf16x8 kf[KPW][4];  // K fragment t of key tile kt: row kt*32 + l31, elements 16 t + 8 half .. + 7 (keys past S repeat key S - 1: never stored)
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int key = (wave + NW * i) * 32 + l31;
    const bf16* kp = base + (size_t)(key < S ? key : S - 1) * row_stride + D + 8 * half;
#pragma unroll
    for (int t = 0; t < 4; ++t) kf[i][t] = *reinterpret_cast<const bf16x8*>(kp + 16 * t);
  }
  const float* lrow = a.lse + (size_t)bh * a.lse_stride;
  for (int q = tid; q < NKT * 32; q += NW * 64) Ls[q] = q < S ? lrow[(size_t)(q >> 5) * a.lse_tile + (q & 31)] : 0.f;
  // lane constants: Q fragment t of a 32-row tile = row l31, chunk 2t + half (swizzled)
  const int fq = pl_f(l31);
  int qo[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qo[t] = l31 * 128 + (((2 * t + half) ^ fq) << 4);
  const int nqt = (S + 31) >> 5;
  const uint64_t head_abs = (uint64_t)(reinterpret_cast<uintptr_t>(phead) >> 2);  // absolute float index of the head's block
  const int wrow = (l31 * S + 4 * half) * 4;   // byte offset of this lane's first float of a key tile inside the band image (before shift / tile)
  const int dump = 32 * S * 4 + 16;            // byte offset of the dump slot behind the image (+ the largest shift): junk of keys >= S goes there
  __syncthreads();

#pragma unroll 1
  for (int qt = 0; qt < nqt; ++qt) {
    char* band = band0 + (qt & 1) * a.band_bytes;
    const uint64_t a0 = head_abs + (uint64_t)qt * 32u * (uint64_t)S;  // absolute float index of the band's first element
    const int ph = (int)(a0 & 255u);                                  // floats between the 1 KB grid of memory and the band's first element
    const int sh4 = (ph & 3) * 4;                                     // the image starts `shift` floats into the buffer
    const float L = Ls[qt * 32 + l31];
    // the last query tile reads Q rows up to 32 nqt - 1 >= rows_q: they land in Ls / the band buffers (junk, possibly changing under the read) --
    // only band rows >= S depend on them, and those are never streamed
    const char* qp = Qimg + qt * 4096;
    bf16x8 qf[4];
    if constexpr ((ABL & 2) == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(qp + qo[t]);
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int kt = wave + NW * i;
      if ((ABL & 2) == 0 && kt < NKT) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][t], qf[t], acc, 0, 0, 0);
        // acc[4g + j]: key kt*32 + 8g + 4 half + j of query qt*32 + l31  ->  band float l31 S + key
        char* wp = band + sh4 + wrow + kt * 128;
        if (kt * 32 + 32 <= S) {  // wave-uniform: every key of the tile exists
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              *reinterpret_cast<float*>(wp + (8 * g + j) * 4) = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[4 * g + j], a.scale_log2e, -L));
        } else {  // the tile that holds keys >= S: their values go to the dump slot (in the unpadded image they would land in the next row)
          const int key0 = kt * 32 + 4 * half;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[4 * g + j], a.scale_log2e, -L));
              char* dst = key0 + 8 * g + j < S ? wp + (8 * g + j) * 4 : band + dump;
              *reinterpret_cast<float*>(dst) = e;
            }
        }
      }
    }
    if constexpr ((ABL & 8) == 0) __syncthreads();  // the band is complete; everyone has finished streaming the band before the last one (the buffer written next)
    // ---- stream: the linear image of min(32, S - 32 qt) rows x S floats in 16-byte chunks on memory's 16-byte grid; chunk c = 64 s + lane of
    //      segment s covers band floats 4 c - ph .. + 3 and sits at LDS byte band + sh4 + 4 (4 c - ph) = band + 16 c - 4 (ph - shift): 16-byte aligned
    const int n = (S - qt * 32 < 32 ? S - qt * 32 : 32) * S;
    float* sbase = phead + (size_t)qt * 32 * S - ph;  // the 1 KB-aligned address segment 0 starts at (its first ph floats belong to what lies before the band)
    const char* lbase = band + sh4 - 4 * ph;          // LDS address of band float -ph, i.e. of chunk 0
    const int nchunk = (ph + n + 3) >> 2;
    const int nseg = (nchunk + 63) >> 6;
#pragma unroll 1
    for (int s0 = wave; s0 < nseg; s0 += 4 * NW) {
      f32x4 v[4];
      int g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = (s0 + NW * i) * 64 + lane;
        g[i] = 4 * c - ph;  // first band float of the chunk
        const bool inside = g[i] + 3 >= 0 && g[i] < n;  // (chunks of segments >= nseg start past n)
        if constexpr ((ABL & 4) != 0) v[i] = f32x4{(float)c, 0.f, 0.f, 0.f};
        else v[i] = *reinterpret_cast<const f32x4*>(inside ? lbase + 16 * c : band);  // (outside: any valid aligned address)
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = (s0 + NW * i) * 64 + lane;
        float* dst = sbase + 4 * (size_t)(uint32_t)c;
        if constexpr ((ABL & 1) != 0) {
          if (v[i][0] == 1.2345e33f) *reinterpret_cast<f32x4*>(dst) = v[i];
        } else if (g[i] >= 0 && g[i] + 4 <= n) {
          *reinterpret_cast<f32x4*>(dst) = v[i];
        } else if (g[i] + 3 >= 0 && g[i] < n) {  // a chunk across the band's first or last float (at most two per band): float by float
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g[i] + j >= 0 && g[i] + j < n) dst[j] = v[i][j];
        }
      }
    }
  }
}

// LDS per workgroup: Q image, log-sum-exp, two band buffers (32 S floats + up to 3 floats of shift + the dump slot, rounded up to 16 B)
static int probs_lse_band_bytes(int S) { return (32 * S * 4 + 16 + 16 + 15) & ~15; }
static int probs_lse_smem(int S) {
  const int nkt = (S + 31) / 32, rows_q = (S + 7) & ~7;
  return rows_q * 128 + nkt * 128 + 2 * probs_lse_band_bytes(S);
}

// S % 8 != 0: the band writes (32 rows of one column per ds_write_b32 group) are at most 4-way bank-conflicted; S >= 64 keeps the head worth a workgroup.
// Other lengths keep the two-pass kernel.
bool attn_probs_lse_supports(int S) { return S >= 64 && S <= 288 && (S & 7) != 0 && probs_lse_smem(S) <= 160 * 1024; }

int g_probs_lse_abl = 0;   // mmamd_debug_set_attn_variant(5000 + bits): ablations of the S = 193 .. 224 instantiation (MMAMD_EXPERIMENTS builds)
int g_probs_lse_pad = 0;   // mmamd_debug_set_attn_variant(5100 + KiB): extra dynamic LDS per workgroup (occupancy A/B: 10 -> one workgroup per CU)

// waves per workgroup: 0 = by length (16 from 7 key tiles up: S = 197 127-142 vs 135-137 us, S = 275 120-136 vs 134-141 us; S = 129 75-81 vs 66-69 us
// the other way: profiles/r05_probs_lse_waves_ab.txt); mmamd_debug_set_attn_variant(5308 / 5316) forces 8 / 16 (A/B)
int g_probs_lse_waves = 0;

template <int NKT, int ABL = 0, int NW = 8>
static int launch_probs_lse_t(const ProbsLseArgs& a, int grid, int smem, hipStream_t st) {
  if constexpr (NW == 8 && ABL == 0) {
    if (g_probs_lse_waves == 16 || (g_probs_lse_waves == 0 && NKT >= 7)) return launch_probs_lse_t<NKT, 0, 16>(a, grid, smem, st);
  }
  static unsigned long long attr_mask = 0;
  auto kern = attention_probs_lse_kernel<NKT, ABL, NW>;
  if (g_probs_lse_pad > 0 && smem + g_probs_lse_pad * 1024 <= 160 * 1024) { smem += g_probs_lse_pad * 1024; attr_mask = 0; }
  if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), smem, attr_mask)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, a);
  return launch_status("attention_probs_lse");
}

// probs [B, H, S, S] fp32.  lse == NULL: on entry the first 32 floats of every band of `probs` hold the log2-domain log-sum-exp of the band's queries
// (mmamd_attention_probs_fwd); else lse is a dense [B, H, S] array (mmamd_attention_probs_from_lse)
int launch_attn_probs_lse(const void* qkv, float* probs, int B, int S, int H, float scale, hipStream_t st, const float* lse) {
  if (!attn_probs_lse_supports(S)) {
    set_error("attention_probs_lse: S=%d is not served", S);
    return MMAMD_E_UNSUPPORTED;
  }
  ProbsLseArgs a;
  a.qkv = (const bf16*)qkv; a.probs = probs; a.S = S; a.H = H;
  a.lse = lse ? lse : probs; a.lse_stride = lse ? S : S * S; a.lse_tile = lse ? 32 : 32 * S;
  a.scale_log2e = scale * 1.4426950408889634f;
  a.rows_q = (S + 7) & ~7;
  a.band_bytes = probs_lse_band_bytes(S);
  const int smem = probs_lse_smem(S);
  const int grid = B * H;
  switch ((S + 31) / 32) {
    case 2: return launch_probs_lse_t<2>(a, grid, smem, st);
    case 3: return launch_probs_lse_t<3>(a, grid, smem, st);
    case 4: return launch_probs_lse_t<4>(a, grid, smem, st);
    case 5: return launch_probs_lse_t<5>(a, grid, smem, st);
    case 6: return launch_probs_lse_t<6>(a, grid, smem, st);
    case 7:
#ifdef MMAMD_EXPERIMENTS
      switch (g_probs_lse_abl) {
        case 1: return launch_probs_lse_t<7, 1>(a, grid, smem, st);
        case 2: return launch_probs_lse_t<7, 2>(a, grid, smem, st);
        case 3: return launch_probs_lse_t<7, 3>(a, grid, smem, st);
        case 4: return launch_probs_lse_t<7, 4>(a, grid, smem, st);
        case 5: return launch_probs_lse_t<7, 5>(a, grid, smem, st);
        case 6: return launch_probs_lse_t<7, 6>(a, grid, smem, st);
        case 8: return launch_probs_lse_t<7, 8>(a, grid, smem, st);
        case 14: return launch_probs_lse_t<7, 14>(a, grid, smem, st);
      }
#endif
      return launch_probs_lse_t<7>(a, grid, smem, st);
    case 8: return launch_probs_lse_t<8>(a, grid, smem, st);
    case 9: return launch_probs_lse_t<9>(a, grid, smem, st);
  }
  set_error("attention_probs_lse: unsupported S=%d", S);
  return MMAMD_E_UNSUPPORTED;
}

}  // namespace mmamd
