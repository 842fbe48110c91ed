/* mmamd_debug.h -- bench, diagnostic and experiment hooks of libmmamd.so.  NOT part of the drop-in surface (include/mmamd.h):
 * nothing in multimodal_amd/models or multimodal_amd/modules calls these; tools/, tests/ and bench.py's probes do.  A reference
 * maintainer integrating the path does not need this header.  Results never change through any of these switches unless the
 * comment of the switch says so (ablation variants). */
#ifndef MMAMD_DEBUG_H
#define MMAMD_DEBUG_H

#include "mmamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* --- GEMM / attention kernel-variant selectors and experiment knobs ------------------------------------------------------------------- */
/* Select a GEMM kernel variant at run time (0 = default).  Test/bench hook; variants are bit-compatible
 * in what they compute, they differ in tiling/pipelining only. */
int mmamd_set_gemm_variant(int variant);
int mmamd_get_gemm_variant(void);
/* Start-up stagger of the persistent GEMM, in per cent of the estimated time of one tile (default 60; 0 = off): the workgroups that walk one tile
 * fewer than the others (the last round of tiles is partial) start up to that much later, spread evenly — it de-synchronises the
 * C-tile store bursts of the 256 CUs and is free as long as it stays below one tile time.  Results do not change.
 * 1000 + percent (experiment, single-problem persistent kernel only): EVERY workgroup is delayed, the 32 of an XCD spread over 0 .. percent of a tile
 * time — measured to cost as much makespan as the spread-out bursts save. */
int mmamd_debug_set_gemm_stagger(int percent);
/* Experiment knobs of the GEMM launchers (results never change; knob 0: tile-order group of the grouped persistent kernel -- 0 = by the stream's
 * CU budget, 4, 8;  knob 1: start-up stagger policy of the grouped kernel -- 0 = light workgroups only, 1 = every workgroup by its slack;  knob 4 (r06): column tiles
 * per chunk of the persistent kernels' tile order -- 0 = by the size of W, -1 = no chunking, k = k tiles). */
int mmamd_debug_set_gemm_knob(int knob, int value);
/* Host-side enumeration of the persistent GEMM kernels' tile order for a tiles_m x tiles_n grid of 256 x 256 tiles (gm / cn <= 0: the launchers' own choice for
 * contraction length K): out[2 id] = row tile, out[2 id + 1] = column tile for id = 0 .. tiles_m * tiles_n - 1; returns the column-chunk width used (or a
 * negative error).  No device work: the CPU test enumerates it and asserts every tile is visited exactly once. */
int mmamd_debug_tile_order(int tiles_m, int tiles_n, int K, int gm, int cn, int* out);
/* W [N, K] bf16 row-major (leading dimension ldw) -> MFMA-fragment order for the direct-W GEMM kernels: ceil(N / 32) x (K / 16) blocks of 1 KiB,
 * block (nb, ks) = 64 lanes x 16 B, lane (l = lane & 31, h = lane >> 5) holds W[32 nb + l][16 ks + 8 h .. + 7] (rows >= N: zeros).  Wp: ceil(N / 32) * 32 * K
 * bf16.  A layout of the static operand of torch's nn.Linear inside TransformerEncoderLayer (models/clip/image_encoder.py:65-77). */
int mmamd_pack_w_frag(const void* W, int ldw, int N, int K, void* Wp, mmamd_stream_t stream);
/* Experiment: fragment-order copy of W used by the direct-W GEMM variants (84, 85) of the following mmamd_gemm_bf16 calls (NULL = none). */
int mmamd_debug_set_gemm_wp(const void* Wp);
/* Diagnostic: device buffer of 64*2*256 uint64 that GEMM variant 14 fills with s_memtime stamps (NULL = off). */
int mmamd_debug_set_gemm_trace(void* buf);
/* Diagnostic: attention ablation variant (timing experiments; non-zero values compute WRONG results). */
int mmamd_debug_set_attn_variant(int v);
/* A/B: 0 = mmamd_colsum never takes its few-rows / very-wide form (every shape goes through the row-per-workgroup form, as before r05); 1 = default. */
int mmamd_debug_set_colsum_wide(int on);

/* --- CU-mask streams, per-stream CU budget (r02 / r04 experiments: profiles/r02_cu_partition_sweep.txt, r04_cu_budget_sweep.txt,
 * r04_phased_schedule_ab.txt), placement probe --------------------------------------------------------------------------------------- */
/* Streams confined to a subset of the CUs (hipExtStreamCreateWithCUMask): the two towers of the dual encoder are independent until
 * the loss (reference models/clip/model.py:70-71 runs them one after the other); here each gets its own CU partition so that neither
 * tower's persistent kernels queue behind the other's.  mask: `words` 32-bit words, bit i = CU i in the runtime's CU numbering.
 * mmamd_stream_cus(stream): CUs a launch on the stream may occupy (256 for any stream not created here) -- what the persistent
 * GEMM / attention kernels size their grids with.  mmamd_debug_cu_census: blocks x {XCC_ID, HW_ID} of a spinning grid (placement probe). */
int mmamd_stream_create_cu_mask(const uint32_t* mask, int words, mmamd_stream_t* out);
int mmamd_stream_destroy(mmamd_stream_t stream);
int mmamd_stream_cus(mmamd_stream_t stream);
/* CU BUDGET of an ordinary stream (no mask): persistent kernels launched on it use `cus` workgroups (a multiple of 8; 0 or >= 256 clears it)
 * instead of one per CU of the chip.  Two streams with a budget of 128 each run their persistent GEMM / attention kernels side by side on
 * disjoint CUs, and a phase shift between the two lets one stream's HBM-bound kernels (LayerNorm, attention, residual epilogues) run while the
 * other's matrix-bound main loops leave the memory system idle: the phased half-batch schedule of the dual encoder (each half-batch of
 * reference models/clip/model.py:65-74 is independent of the other until the loss).  Host-side state, read when a launch is enqueued. */
int mmamd_stream_set_cus(mmamd_stream_t stream, int cus);
int mmamd_debug_cu_census(int* out, int blocks, long long spin_ticks, mmamd_stream_t stream);

/* --- launch counters: how many times the launcher named `what` (the string its errors carry: "attention_probs_lse", "attention_ring",
 * "gemm_bf16_splitk", "layernorm_bwd", "colsum_stage2_batched", ...) has enqueued since the library was loaded.  The A/B tools assert on the difference
 * around each arm that the knob they flipped took effect (no profiler needed). */
unsigned long long mmamd_debug_launch_count(const char* what);

/* --- timing helper for bench.py: HIP events on the SAME stream the kernels run on ------------
 * mmamd_timer_create returns an opaque handle (two hipEvents); start/stop record on `stream`;
 * elapsed_ms synchronises on the stop event (host-side call, not capturable). */
void* mmamd_timer_create(void);
void mmamd_timer_destroy(void* t);
int mmamd_timer_start(void* t, mmamd_stream_t stream);
int mmamd_timer_stop(void* t, mmamd_stream_t stream);
int mmamd_timer_elapsed_ms(void* t, float* ms_host);

#ifdef __cplusplus
}
#endif
#endif /* MMAMD_DEBUG_H */
