// tools/ubench/mfma_power.hip -- the matrix pipe's sustained rate on REGISTER operands (no memory, no LDS) as a function of the operand
// DATA: the chip is power-limited under dense MFMA issue and clocks with the bits it multiplies (cdna_hip_programming.md rule 25).
// This is the ceiling any bf16 GEMM-class kernel of this build can approach on real activations, whatever its operand path.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power && tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ inline unsigned lcg(unsigned& h) { h = h * 1664525u + 1013904223u; return h; }
// fill: 0 zeros, 1 uniform [-1, 1) (full-range mantissa and sign), 2 N(0,1)-like (sum of 4 uniforms), 3 sign-constant |.| of 1
__device__ inline unsigned pair(unsigned& h, int fill) {
  if (fill == 0) return 0u;
  float v[2];
  for (int i = 0; i < 2; ++i) {
    float u = (float)(int)(lcg(h) >> 8) * (1.0f / 8388608.0f) - 1.0f;
    if (fill == 2) { for (int k = 0; k < 3; ++k) u += (float)(int)(lcg(h) >> 8) * (1.0f / 8388608.0f) - 1.0f; u *= 0.866f; }
    if (fill == 3) u = fabsf(u);
    v[i] = u;
  }
  union { __bf16 b[2]; unsigned u; } x;
  x.b[0] = (__bf16)v[0]; x.b[1] = (__bf16)v[1];
  return x.u;
}

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int fill) {
  unsigned h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    u32x4 ua, ub;
    for (int j = 0; j < 4; ++j) { ua[j] = pair(h, fill); ub[j] = pair(h, fill); }
    a[i] = __builtin_bit_cast(bf16x8, ua); b[i] = __builtin_bit_cast(bf16x8, ub);
  }
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
      // keep the accumulators bounded without touching the issue pattern much: every 64 iterations rescale
      if ((it & 63) == 63) for (int i = 0; i < 16; ++i) acc[i] *= 1e-3f;
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + q) & 3], b[i], acc[i], 0, 0, 0);
      if ((it & 63) == 63) for (int i = 0; i < 4; ++i) acc[i] *= 1e-3f;
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  }
  if (s == 123.456f) out[0] = s;
}

int main() {
  float* d; hipMalloc(&d, 4);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount, iters = 1600000;   // ~0.4 s per run: long enough for the power management to settle
  const char* names[4] = {"zeros", "uniform [-1,1)", "normal-like", "uniform [0,1) (sign constant)"};
  printf("# %d CUs, 8 waves per CU (2 per SIMD), %d iterations x 16 MFMA-equivalents of 16 K flop per wave\n", blocks, iters);
  for (int shape = 0; shape < 2; ++shape)
    for (int fill = 0; fill < 4; ++fill)
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, 200, fill); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, 200, fill);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, iters, fill); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, iters, fill);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 8 * iters * (shape == 0 ? 16 * 16384.0 : 8 * 32768.0);
        if (rep == 1) printf("%-10s %-32s %8.2f ms  %7.0f TFLOP/s\n", shape == 0 ? "16x16x32" : "32x32x16", names[fill], ms, fl / ms / 1e9);
      }
  return 0;
}
