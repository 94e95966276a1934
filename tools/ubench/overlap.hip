// How well do MFMA and VALU work overlap on one SIMD of gfx950?  (2 or more waves per SIMD)
//  mode A: every wave runs groups of (1 MFMA + k VALU)
//  mode B: waves alternate roles by block parity: even blocks MFMA-only, odd blocks VALU-only
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ADD6 "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n"
#define EXP2ADD4 "v_exp_f32 %0, %0\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_exp_f32 %3, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

// MODE: 0 = mfma only, 1 = 6 add only, 2 = mfma+6add same wave, 3 = role split (even block mfma x2, odd block 12 add),
//       4 = mfma + (2exp+4add) same wave, 5 = role split with exp mix, 6 = (2exp+4add) only
//       7 = 2 mfma then 12 add same wave, 8 = mfma+6add with setprio(1) around mfma
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x16 acc0 = {0}, acc1 = {0};
  bf16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1}, fb = fa;
  const bool odd = blockIdx.x & 1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
      } else if (MODE == 1) {
        asm volatile(ADD6 OPS); asm volatile(ADD6 OPS);
      } else if (MODE == 6) {
        asm volatile(EXP2ADD4 OPS); asm volatile(EXP2ADD4 OPS);
      } else if (MODE == 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        asm volatile(ADD6 OPS);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        asm volatile(ADD6 OPS);
      } else if (MODE == 4) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        asm volatile(EXP2ADD4 OPS);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        asm volatile(EXP2ADD4 OPS);
      } else if (MODE == 3 || MODE == 5) {
        if (!odd) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        } else if (MODE == 3) {
          asm volatile(ADD6 OPS); asm volatile(ADD6 OPS); asm volatile(ADD6 OPS); asm volatile(ADD6 OPS);
        } else {
          asm volatile(EXP2ADD4 OPS); asm volatile(EXP2ADD4 OPS); asm volatile(EXP2ADD4 OPS); asm volatile(EXP2ADD4 OPS);
        }
      } else if (MODE == 7) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        asm volatile(ADD6 OPS); asm volatile(ADD6 OPS);
      } else if (MODE == 8) {
        __builtin_amdgcn_s_setprio(1);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile(ADD6 OPS);
        __builtin_amdgcn_s_setprio(1);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile(ADD6 OPS);
      }
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wps) {
  float* out;
  (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 1000, blocks = 256 * wps;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // "pair" = 2 MFMA and/or 12 VALU of work; per SIMD there are wps waves each doing iters*8 pairs
  // (role split: half the waves do 4 MFMA, half 24 VALU per step -> same total work per 2 waves as modes 2/4)
  printf("%-44s waves/SIMD=%d %.3f ms -> %.1f cycles(2.1GHz) per [2 MFMA + 12 VALU] per SIMD\n", name, wps, ms,
         ms * 1e-3 * 2.1e9 / ((double)iters * 8 * wps));
  (void)hipFree(out);
}

int main() {
  for (int w = 2; w <= 4; w += 2) {
    run<0>("mfma only (2 per step)", w);
    run<1>("12 v_add only", w);
    run<6>("4 v_exp + 8 v_add only", w);
    run<2>("same wave: (mfma, 6 add) x2", w);
    run<7>("same wave: 2 mfma, 12 add", w);
    run<8>("same wave: (mfma@prio1, 6 add) x2", w);
    run<3>("role split: even 4 mfma | odd 24 add", w);
    run<4>("same wave: (mfma, 2 exp + 4 add) x2", w);
    run<5>("role split: even 4 mfma | odd 8exp+16add", w);
  }
  return 0;
}
