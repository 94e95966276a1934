// Does a counted `s_waitcnt vmcnt(N)` retire LDS-DMA loads (buffer_load ... lds) IN ISSUE ORDER on gfx950 when their
// latencies differ (an HBM miss followed by L2 hits), and with stores in the queue?  Every counted-vmcnt ring in csrc/
// assumes it.  (the rare corrupt rows of lin_chain_kernel.)
//
// Each wave, per iteration:  sentinel -> LDS slot 0;  DMA "cold" (1 KB from a pseudo-random place of a 2 GiB buffer: an
// HBM miss, slower still while other streams keep HBM busy) -> slot 0;  then K "hot" operations that complete quickly
// (MODE 0: LDS-DMAs of one L2-resident KB into slot 1.., MODE 1: global stores of 16 B per lane to a small buffer,
// MODE 2: alternating);  s_waitcnt vmcnt(K)  -- in-order retirement means the cold DMA has landed --;  read slot 0 and
// compare with what the cold source holds (the buffer is filled with f(address)).  A mismatch = the wait was satisfied
// while the OLDEST operation was still in flight.  MODE 3: the same with vmcnt(0) (must never fail).
// Run alone and with a bandwidth hog on a second stream.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/vmcnt_order.hip -o tools/ubench/vmcnt_order && tools/ubench/vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void fill(uint32_t* p, size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 32);
}
__global__ void hog(const u32x4* a, u32x4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <int MODE, int K>
__global__ __launch_bounds__(256, 1) void probe(const uint32_t* __restrict__ cold, size_t cold_kb, const uint32_t* __restrict__ hot,
                                                uint32_t* __restrict__ sink, unsigned long long* __restrict__ bad, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* my = smem + wave * (K + 2) * 1024;
  const __amdgpu_buffer_rsrc_t rs_cold = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(cold), (short)0, (int)0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hot = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(hot), (short)0, 64 << 10, 0x00020000);
  uint64_t rng = 0x9E3779B97F4A7C15ull * (blockIdx.x * 4 + wave + 1);
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    const size_t kb = (size_t)((rng >> 20) % cold_kb);                  // which KB of the cold buffer
    // sentinel, visible to the wave itself before the DMA is issued
    reinterpret_cast<u32x4*>(my)[lane] = (u32x4){0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (the buffer form, as in csrc/: descriptor + 32-bit lane offset + scalar offset)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cold, (__attribute__((address_space(3))) void*)my, 16, (unsigned)(lane * 16), (int)(kb * 1024), 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool as_store = MODE == 1 || (MODE == 2 && (k & 1));
      if (as_store) {
        u32x4 v = {(unsigned)it, (unsigned)k, (unsigned)lane, 0u};
        // plain 16-byte store to a small (L2-resident) buffer
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(sink + ((size_t)(blockIdx.x * 4 + wave) * K + k) * 256 + lane * 4), "v"(v) : "memory");
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_hot, (__attribute__((address_space(3))) void*)(my + (k + 1) * 1024), 16, (unsigned)(lane * 16), k * 1024, 0, 0);
      }
    }
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
    const u32x4 got = reinterpret_cast<const u32x4*>(my)[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const size_t w0 = kb * 256 + lane * 4;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) ok &= got[j] == ((uint32_t)((w0 + j) * 2654435761u) ^ (uint32_t)((w0 + j) >> 32));
    nbad += __popcll(__ballot(!ok)) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // nothing of this iteration survives into the next
  }
  if (lane == 0 && nbad) atomicAdd(bad, nbad);
}

template <int MODE, int K>
void run(const char* what, const uint32_t* cold, size_t cold_kb, const uint32_t* hot, uint32_t* sink, unsigned long long* bad, int cus, int iters,
         hipStream_t side, const u32x4* ha, u32x4* hb, size_t hn) {
  for (int with_hog = 0; with_hog < 2; ++with_hog) {
    CK(hipMemset(bad, 0, 8));
    if (with_hog)
      for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(hog, dim3(2048), dim3(256), 0, side, ha, hb, hn);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(cus), dim3(256), 4 * (K + 2) * 1024, 0, cold, cold_kb, hot, sink, bad, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("%-34s K=%2d  %s : %llu of %llu waits let the oldest DMA pass unfinished\n", what, K, with_hog ? "HBM busy" : "idle    ", h,
           (unsigned long long)cus * 4 * iters);
  }
}

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t cold_bytes = ((size_t)2 << 30) - (1 << 20), cold_kb = cold_bytes >> 10;     // (below the 2 GiB of a 32-bit buffer offset)
  uint32_t *cold, *hot, *sink; unsigned long long* bad; u32x4 *ha, *hb;
  const size_t hog_bytes = (size_t)1 << 30;
  CK(hipMalloc(&cold, cold_bytes)); CK(hipMalloc(&hot, 64 << 10)); CK(hipMalloc(&sink, (size_t)cus * 4 * 16 * 1024)); CK(hipMalloc(&bad, 8));
  CK(hipMalloc(&ha, hog_bytes)); CK(hipMalloc(&hb, hog_bytes));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, cold_bytes / 4);
  CK(hipMemset(hot, 1, 64 << 10)); CK(hipMemset(ha, 2, hog_bytes));
  CK(hipDeviceSynchronize());
  hipStream_t side; CK(hipStreamCreate(&side));
  const int iters = 20000;
  run<3, 4>("drained (vmcnt 0), control", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<0, 1>("cold DMA + hot DMAs", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<0, 4>("cold DMA + hot DMAs", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<0, 12>("cold DMA + hot DMAs", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<1, 1>("cold DMA + stores", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<1, 4>("cold DMA + stores", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<1, 12>("cold DMA + stores", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  run<2, 12>("cold DMA + DMAs and stores mixed", cold, cold_kb, hot, sink, bad, cus, iters, side, ha, hb, hog_bytes / 16);
  return 0;
}
