// valu_rates.hip — issue cost of the VALU instructions the attention softmax is made of, per SIMD, with 1 and 2 waves per SIMD,
// alone and mixed (does the transcendental pipe overlap plain VALU / MFMA?).  Prints shader cycles per wave-instruction per SIMD
// assuming the clock measured by a v_fma_f32 loop at 2 waves per SIMD = 2 cycles per instruction (guide: SIMD-32).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o tools/microbench/valu_rates && tools/microbench/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
enum { FMA, EXP, MAX3, CVT, PKFMA, PKADD, MOV, MIX_EXP_FMA3, MIX_EXP_FMA1, MFMA_ONLY, MFMA_EXP2, MFMA_FMA4, ROLE_EXP_FMA, ADD, MUL, ROLE_MFMA_FMA, ROLE_MFMA_EXP, MFMA_DEP_FMA8, NKIND };
static const char* kNames[] = {"v_fma_f32", "v_exp_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_mov_b32",
                               "1 exp + 3 fma (per 4 instr)", "1 exp + 1 fma (per 2 instr)", "mfma 32x32x16 only (per mfma)",
                               "mfma + 2 exp (per group)", "mfma + 4 fma (per group)", "roles: even waves exp, odd waves fma", "v_add_f32", "v_mul_f32", "roles: waves 0-3 mfma, waves 4-7 fma", "roles: waves 0-3 mfma, waves 4-7 exp", "mfma + 8 fma (per group, dependent mfma chain)"};

template <int KIND>
__global__ void burn(int iters, float* sink, float seed) {
  float r[8], p[8];
  for (int i = 0; i < 8; ++i) { r[i] = seed * (i + 1) + threadIdx.x * 1e-3f; p[i] = r[i] * 0.5f; }
  f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.37f * (threadIdx.x % 7)); b[j] = (__bf16)(0.11f * j - 0.3f); }
  const float c = 0.999f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (KIND == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(p[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(p[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
        REP8(X)
#undef X
      } else if constexpr (KIND == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(p[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == CVT) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(p[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&r[i]) : "v"(*(double*)&p[0]), "v"(*(double*)&p[2]));
        X(0) X(2) X(4) X(6) X(0) X(2) X(4) X(6)
#undef X
      } else if constexpr (KIND == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&r[i]) : "v"(*(double*)&p[0]));
        X(0) X(2) X(4) X(6) X(0) X(2) X(4) X(6)
#undef X
      } else if constexpr (KIND == MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(p[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == MIX_EXP_FMA3) {
        asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %4, %5\n v_fma_f32 %8, %8, %4, %5\n v_fma_f32 %9, %9, %4, %5"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(c), "v"(p[0]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
      } else if constexpr (KIND == MIX_EXP_FMA1) {
        asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %4, %5\n v_exp_f32 %8, %8\n v_fma_f32 %9, %9, %4, %5"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(c), "v"(p[0]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
      } else if constexpr (KIND == MFMA_ONLY) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      } else if constexpr (KIND == MFMA_EXP2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(r[0]), "+v"(r[1]));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(r[2]), "+v"(r[3]));
      } else if constexpr (KIND == MFMA_FMA4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(c), "v"(p[0]));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(c), "v"(p[0]));
      } else if constexpr (KIND == ROLE_MFMA_FMA || KIND == ROLE_MFMA_EXP) {
        if (wave < 4) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if constexpr (KIND == ROLE_MFMA_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(p[i]));
          REP8(X) REP8(X)
#undef X
        } else {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
          REP8(X)
#undef X
        }
      } else if constexpr (KIND == MFMA_DEP_FMA8) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(p[i]));
        REP8(X)
#undef X
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(p[i]));
        REP8(X)
#undef X
      } else if constexpr (KIND == ROLE_EXP_FMA) {
        // waves 0-3 (first on each SIMD) run exp, waves 4-7 run fma: does a SIMD overlap one wave's transcendentals with the other's VALU?
        if (wave < 4) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
          REP8(X)
#undef X
        } else {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(p[i]));
          REP8(X)
#undef X
        }
      }
    }
  }
  float s = acc[0];
  for (int i = 0; i < 8; ++i) s += r[i];
  if (s == 1.234e33f) sink[0] = s;
}

template <int KIND>
static double run(int threads, int iters, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(burn<KIND>, dim3(256), dim3(threads), 0, 0, iters / 10, sink, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(burn<KIND>, dim3(256), dim3(threads), 0, 0, iters, sink, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best * 1e6;  // ns
}

int main() {
  float* sink; hipMalloc(&sink, 16);
  const int iters = 20000;
  // per-kind: instructions (or groups) per loop iteration per wave
  const double per_iter[NKIND] = {32, 32, 32, 32, 32, 32, 32, 8 /*groups of 4 -> 8 instr: count instr*/ * 4, 8 * 4, 8, 8, 8, 32, 32, 32, 8, 8, 8};
  double ns_fma2 = run<FMA>(512, iters, sink);
  const double clk = (2.0 * iters * 32 * 2.0) / ns_fma2;  // GHz if v_fma_f32 = 2 cycles per instruction per SIMD, 2 waves per SIMD
  printf("calibration: v_fma_f32, 2 waves/SIMD: %.1f us -> %.2f GHz if 2 cycles per wave-instruction\n", ns_fma2 / 1e3, clk);
#define RUN(K)                                                                                                              \
  for (int w = 1; w <= 2; ++w) {                                                                                            \
    const double ns = run<K>(256 * w, iters, sink);                                                                         \
    printf("%-40s %d wave(s)/SIMD: %8.1f us  -> %6.2f cycles per instr(group) per SIMD\n", kNames[K], w, ns / 1e3,          \
           ns * clk / (w * iters * per_iter[K]));                                                                           \
  }
  RUN(FMA) RUN(ADD) RUN(MUL) RUN(EXP) RUN(MAX3) RUN(CVT) RUN(PKFMA) RUN(PKADD) RUN(MOV) RUN(MIX_EXP_FMA3) RUN(MIX_EXP_FMA1) RUN(MFMA_ONLY) RUN(MFMA_EXP2) RUN(MFMA_FMA4)
  RUN(MFMA_DEP_FMA8)
  {
    const double ns = run<ROLE_MFMA_FMA>(512, iters, sink);
    printf("%-40s 2 wave(s)/SIMD: %8.1f us  (per iteration: 8 mfma in waves 0-3, 64 fma in waves 4-7; mfma alone / fma alone at 1 wave per SIMD above)\n", kNames[ROLE_MFMA_FMA], ns / 1e3);
    const double ns2 = run<ROLE_MFMA_EXP>(512, iters, sink);
    printf("%-40s 2 wave(s)/SIMD: %8.1f us  (per iteration: 8 mfma in waves 0-3, 32 exp in waves 4-7)\n", kNames[ROLE_MFMA_EXP], ns2 / 1e3);
  }
  {
    const double ns = run<ROLE_EXP_FMA>(512, iters, sink);
    printf("%-40s 2 wave(s)/SIMD: %8.1f us  (exp alone at 1 wave/SIMD and fma alone at 1 wave/SIMD above)\n", kNames[ROLE_EXP_FMA], ns / 1e3);
  }
  return 0;
}
