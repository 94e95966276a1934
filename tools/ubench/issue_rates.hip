// Micro-benchmarks of per-SIMD issue rates on gfx950 (cycles per wave64 instruction), used to
// budget the attention inner loop.  hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x16 acc0 = {0}, acc1 = {0};
  bf16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1}, fb = fa;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // 64 v_exp_f32, 8 independent chains
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 1) {   // 64 v_add_f32
      REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 2) {   // 64 v_cvt_pk_bf16_f32
      REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %1, %2, %3\n v_cvt_pk_bf16_f32 %2, %3, %4\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %4, %5, %6\n v_cvt_pk_bf16_f32 %5, %6, %7\n v_cvt_pk_bf16_f32 %6, %7, %0\n v_cvt_pk_bf16_f32 %7, %0, %1"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 3) {   // 64 v_max3_f32
      REP8(asm volatile("v_max3_f32 %0, %1, %2, %3\n v_max3_f32 %1, %2, %3, %4\n v_max3_f32 %2, %3, %4, %5\n v_max3_f32 %3, %4, %5, %6\n v_max3_f32 %4, %5, %6, %7\n v_max3_f32 %5, %6, %7, %0\n v_max3_f32 %6, %7, %0, %1\n v_max3_f32 %7, %0, %1, %2"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 4) {   // 16 MFMA 32x32x16, two independent accumulators
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);)
    } else if (MODE == 5) {   // 16 x (MFMA + 6 v_add_f32)
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
           asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
           asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 6) {   // 16 x (MFMA + 2 v_exp + 4 v_add)
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
           asm volatile("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_exp_f32 %3, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
           asm volatile("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_exp_f32 %3, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 7) {   // 64 v_sub+v_exp pairs? : 32 v_exp + 32 v_add interleaved
      REP8(asm volatile("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %2\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %4\n v_exp_f32 %4, %4\n v_add_f32 %5, %5, %6\n v_exp_f32 %6, %6\n v_add_f32 %7, %7, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int blocks_per_cu) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 2048 * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 2000;
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // one block = 4 waves = 1 wave per SIMD; waves per SIMD = blocks_per_cu
  const double inst_per_simd = (double)iters * per_iter * blocks_per_cu;
  printf("%-34s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD; s_memtime ticks/instr (1 wave) %.2f\n", name,
         blocks_per_cu, ms, ms * 1e6 / inst_per_simd, (double)c / ((double)iters * per_iter));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_exp_f32", 64, w);
    run<1>("v_add_f32", 64, w);
    run<2>("v_cvt_pk_bf16_f32", 64, w);
    run<3>("v_max3_f32", 64, w);
    run<4>("mfma_32x32x16_bf16", 16, w);
    run<5>("mfma + 6 v_add (per group of 7)", 16, w);
    run<6>("mfma + 2 v_exp + 4 v_add (per group)", 16, w);
    run<7>("v_exp/v_add alternating", 64, w);
  }
  return 0;
}
