// What does the inner loop of a "tile in LDS, weight slices straight from L2 into registers" chain reach on gfx950?
// (the mapping proposed for the token-local chains.)
//
// One persistent block of 4 waves per CU (one wave per SIMD).  The activation tile X[BM][C] (bf16) sits in LDS, XOR
// swizzled per 128-byte line like igemm's operand tiles.  Wave w owns the output columns [w C/4, (w+1) C/4): per 32-deep
// k-step it loads its NI = C/64 weight fragments (1 KB each, fragment-major stream: contiguous per wave) from global
// memory / L2 into VGPRs -- no LDS round trip, nobody else needs them --, reads the MI = BM/16 activation fragments from
// LDS, and issues NI x MI v_mfma_f32_16x16x32_bf16.  Next k-step's fragments are requested before this one's MFMAs.
//   LDS read traffic: MI KB per NI x MI MFMAs (C = 320, BM = 128: 0.2 KB / MFMA; the round-3 chains: 1 KB / MFMA)
//   L2 traffic:       NI KB per NI x MI MFMAs per wave, i.e. C*C*2 bytes per BM rows and layer
// MODE bit 0: weight loads on, bit 1: LDS fragment reads on (off: the fragments of k-step 0 are reused).
// Prints TFLOP/s over all CUs for (C, BM) = (320, 128) and (640, 64).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/chain_core.hip -o tools/ubench/chain_core && tools/ubench/chain_core
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int C, int BM, int MODE>
__global__ __launch_bounds__(256, 1) void core(const uint16_t* __restrict__ wstream, float* __restrict__ out, int layers, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NI = C / 64, MI = BM / 16, KS = C / 32;
  constexpr int ROW = C * 2;                       // bytes per activation row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // fill the tile (any finite bf16 pattern)
  for (int i = tid; i < BM * C / 2; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + ((i * 2654435761u) & 0x007f007fu);
  __syncthreads();
  // X fragment of rows 16 i + fr, k-step ks: chunk (4 ks + fq) of the row, swizzled inside its 128-byte line
  int x_rd[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) x_rd[i] = (i * 16 + fr) * ROW + ((fq ^ ((i * 16 + fr) & 7)) << 4);
  f32x4 acc[NI][MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // weight stream: [layer][ks][wave][j][lane] 16 bytes
  const u32x4* wp = reinterpret_cast<const u32x4*>(wstream) + (size_t)wave * NI * 64 + lane;
  constexpr size_t KSTEP = (size_t)4 * NI * 64;    // u32x4 per k-step
  bf16x8 wa[NI], wb[NI], xa[MI], xb[MI];
  auto load_w = [&](bf16x8 (&w)[NI], size_t step) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (MODE & 1) w[j] = __builtin_bit_cast(bf16x8, wp[step * KSTEP + j * 64]);
    }
  };
  auto load_x = [&](bf16x8 (&x)[MI], int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // chunk index 4 ks + fq: line = ks >> 1, chunk-in-line = 4 (ks & 1) + fq (the XOR only touches the low 3 bits)
      const int a = x_rd[i] ^ ((ks & 1) << 6);
      if (MODE & 2) x[i] = *reinterpret_cast<const bf16x8*>(smem + a + (ks >> 1) * 128);
    }
  };
  auto mfmas = [&](const bf16x8 (&w)[NI], const bf16x8 (&x)[MI]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], x[i], acc[j][i], 0, 0, 0);
  };
  {
    const u32x4 z = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
    for (int j = 0; j < NI; ++j) wa[j] = wb[j] = __builtin_bit_cast(bf16x8, z);
#pragma unroll
    for (int i = 0; i < MI; ++i) xa[i] = xb[i] = __builtin_bit_cast(bf16x8, z);
  }
  const size_t steps_per_layer = KS;
  size_t step = 0;
  const size_t total_steps = (size_t)layers * steps_per_layer;      // the stream is cyclic over the layers
  load_w(wa, 0);
  load_x(xa, 0);
  for (int t = 0; t < tiles * layers; ++t) {
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      size_t s1 = step + 1; if (s1 >= total_steps) s1 -= total_steps;
      load_w(wb, s1);
      load_x(xb, ks + 1);
      mfmas(wa, xa);
      __builtin_amdgcn_sched_barrier(0);           // (keeps the scheduler from hoisting later k-steps' reads: register pressure)
      size_t s2 = s1 + 1; if (s2 >= total_steps) s2 -= total_steps;
      load_w(wa, s2);
      load_x(xa, (ks + 2) % KS);
      mfmas(wb, xb);
      __builtin_amdgcn_sched_barrier(0);
      step = s2;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int C, int BM, int MODE>
void run(const uint16_t* w, float* out, int cus, int layers, int tiles) {
  const int lds = BM * C * 2;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&core<C, BM, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((core<C, BM, MODE>), dim3(cus), dim3(256), lds, 0, w, out, layers, 2);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((core<C, BM, MODE>), dim3(cus), dim3(256), lds, 0, w, out, layers, tiles);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double flops = 2.0 * cus * (double)tiles * layers * BM * C * C;
  printf("C=%4d BM=%3d layers=%d  weights %s  LDS reads %s : %8.3f ms  %7.1f TFLOP/s  (%5.1f GB/s of weights per CU)\n", C, BM, layers,
         (MODE & 1) ? "on " : "off", (MODE & 2) ? "on " : "off", best, flops / best / 1e9,
         (MODE & 1) ? (double)tiles * layers * C * C * 2.0 / best / 1e6 : 0.0);
}

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t wbytes = (size_t)4 * 640 * 640 * 2;     // up to four layers of 640 x 640
  uint16_t* w; float* out;
  CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&out, (size_t)cus * 256 * 4));
  std::vector<uint16_t> hw(wbytes / 2);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 25));
  CK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice));
  printf("%d CUs\n", cus);
  for (int layers : {2, 4}) {
    run<320, 128, 0>(w, out, cus, layers, 60);
    run<320, 128, 2>(w, out, cus, layers, 60);
    run<320, 128, 1>(w, out, cus, layers, 60);
    run<320, 128, 3>(w, out, cus, layers, 60);
  }
  for (int layers : {2, 4}) {
    run<640, 64, 0>(w, out, cus, layers, 30);
    run<640, 64, 2>(w, out, cus, layers, 30);
    run<640, 64, 1>(w, out, cus, layers, 30);
    run<640, 64, 3>(w, out, cus, layers, 30);
  }
  run<640, 128, 3>(w, out, cus, 2, 30);      // (160 KB tile: all of the LDS; registers: 320 accumulators)
  return 0;
}
