// Issue-only model of one self-attention "unit" (d = 40): 7 MFMAs (4 P.V + 3 QK^T) with the softmax VALU
// pieces between them, no memory, no branches.  Gives the floor the real kernel can approach.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define PIECE(i) asm volatile("v_exp_f32 %1, %3\n\tv_exp_f32 %2, %4\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" \
                              : "=&v"(p[i]), "=&v"(e0), "=&v"(e1) : "v"(s[2 * (i)]), "v"(s[2 * (i) + 1]));
#define MAX4(m, c) asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "+v"(m) : "v"(s[4 * (c)]), "v"(s[4 * (c) + 1]), "v"(s[4 * (c) + 2]), "v"(s[4 * (c) + 3]));
#define MF(acc) if (PRIO) __builtin_amdgcn_s_setprio(1); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0); if (PRIO) __builtin_amdgcn_s_setprio(0);

template <int MODE, int PRIO>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  f32x16 o0 = {0}, o1 = {0}, sq = {0}, s;
  for (int r = 0; r < 16; ++r) s[r] = threadIdx.x * 1e-3f + r;
  bf16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1}, fb = fa;
  uint32_t p[8];
  float e0, e1, mxa = 0, mxb = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {          // the kernel's schedule: P P Q P Q P Q, pieces spread
      MF(o0) PIECE(0)
      MF(o1) PIECE(1) MAX4(mxa, 0)
      MF(sq) PIECE(2)
      MF(o0) PIECE(3) MAX4(mxa, 1)
      MF(sq) PIECE(4)
      MF(o1) PIECE(5) MAX4(mxb, 2)
      MF(sq) PIECE(6) PIECE(7) MAX4(mxb, 3)
    } else if (MODE == 1) {   // MFMAs only
      MF(o0) MF(o1) MF(sq) MF(o0) MF(sq) MF(o1) MF(sq)
    } else if (MODE == 2) {   // VALU only
      PIECE(0) PIECE(1) MAX4(mxa, 0) PIECE(2) PIECE(3) MAX4(mxa, 1) PIECE(4) PIECE(5) MAX4(mxb, 2) PIECE(6) PIECE(7) MAX4(mxb, 3)
    } else if (MODE == 3) {   // all MFMAs first, then all VALU
      MF(o0) MF(o1) MF(sq) MF(o0) MF(sq) MF(o1) MF(sq)
      PIECE(0) PIECE(1) MAX4(mxa, 0) PIECE(2) PIECE(3) MAX4(mxa, 1) PIECE(4) PIECE(5) MAX4(mxb, 2) PIECE(6) PIECE(7) MAX4(mxb, 3)
    }
    // feed something back so nothing is hoisted
    asm volatile("" : "+v"(s[0]), "+v"(mxa), "+v"(mxb));
  }
  float acc = mxa + mxb;
  for (int r = 0; r < 16; ++r) acc += o0[r] + o1[r] + sq[r];
  for (int r = 0; r < 8; ++r) acc += (float)p[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int PRIO>
void run(const char* name, int wps) {
  float* out;
  (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 4000, blocks = 256 * wps;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, PRIO><<<blocks, 256>>>(out, 10);
  (void)hipDeviceSynchronize();
  // clock reference: time the MFMA-only loop right before
  (void)hipEventRecord(e0);
  k<MODE, PRIO><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s prio=%d waves/SIMD=%d %.3f ms -> %.2f ns per unit per SIMD\n", name, PRIO, wps, ms, ms * 1e6 / ((double)iters * wps));
  (void)hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<1, 0>("7 MFMA only", w);
    run<2, 0>("VALU only (16 exp, 8 cvt, 8 max3)", w);
    run<0, 0>("interleaved", w);
    run<0, 1>("interleaved", w);
    run<3, 0>("MFMA block then VALU block", w);
    run<3, 1>("MFMA block then VALU block", w);
  }
  return 0;
}
