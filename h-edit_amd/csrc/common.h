// Shared device/host helpers for the h-Edit HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

// ---- the 16-bit storage format of activations and weights in HBM = the MFMA operand type.
// Default: bfloat16, what BASELINE configs[1] names and what every measured figure of this repository is quoted on.
// -DHEDIT_STORE_F16 compiles the SAME kernels on IEEE half (libhedit_hip_f16.so, build.py --f16): same MFMA rate, three more
// mantissa bits -- the eps error of the SD UNet against the fp32 reference falls from 1.2e-2 to 1.7e-3 (DESIGN.md 5.0 item 3,
// storage emulation), i.e. the "fp16 tolerance" of the reference's own half-precision mode (text-guided/main_p2p.py:106 loads fp32;
// diffusers' torch_dtype=float16 is the mode users run).  The type names keep their bf16 spelling: `bf16_t` = 16 raw storage
// bits, `bf16x8` = one MFMA operand register pair, pack_bf16x2 / bf16_to_f32 = the storage conversions of the build.
// The split-bf16 "precise" GEMMs of the reward networks (pnet.hip) stay bf16 triples in either build (half's lo part would go
// subnormal below 0.1): they use the *_always helpers and igemm_kernel's OPB instantiations.
typedef uint16_t bf16_t;  // raw 16-bit storage element in memory
#if defined(HEDIT_STORE_F16)
#define HEDIT_F16 1
typedef _Float16 st_elem_t;
#define MFMA_16x16x32_ST __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MFMA_32x32x16_ST __builtin_amdgcn_mfma_f32_32x32x16_f16
#define MFMA_ST_SFX "f16"
#else
#define HEDIT_F16 0
typedef __bf16 st_elem_t;
#define MFMA_16x16x32_ST __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MFMA_32x32x16_ST __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define MFMA_ST_SFX "bf16"
#endif
typedef __attribute__((ext_vector_type(8))) st_elem_t bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 true_bf16x8;      // operands of the split-bf16 GEMMs, either build
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

// ---- error plumbing: thread-local message, negative return codes, never throws across the ABI
void hedit_set_error(const std::string& msg);
// Called from the catch (...) of every extern "C" entry point (they are function-try-blocks): turns whatever the C++
// runtime threw inside the library (std::bad_alloc from a handle's containers, ...) into an error code + message.
int hedit_abi_catch() noexcept;
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are PER DEVICE: a process that drives several GPUs (or
// switches devices) must not inherit the first device's answer.  Both are remembered per (kernel, device) / per device
// behind a mutex; a launch pays one hipGetDevice and a small lookup.
int hedit_dyn_lds(const void* kernel, int bytes);      // HEDIT_OK, or HEDIT_ERR_HIP with the message set
int hedit_cu_count(int* cus);                          // CUs of the CURRENT device
// Test switch (hedit_test_set_flags in include/hedit.h; tests/test_gpu_ring_hazard.py): bit 0 = the kernels with counted
// vmcnt rings (igemm, ffn_chain, self_attn) run their DRAINED twin -- every ring wait vmcnt(0) --, bit 1 = self-attention
// takes the exact online-softmax pass only (no pinned-shift pass), bit 2 = the pixel UNet's other GroupNorm-statistics path, bit 3 = the
// one-shot igemm_kernel instead of the persistent kernels (pgemm.hip / pconv.hip).  0 in every product path; never read on a device.
int hedit_test_flags();
inline bool hedit_test_drained() { return (hedit_test_flags() & 1) != 0; }
#define HEDIT_OK 0
#define HEDIT_ERR_ARG (-1)
#define HEDIT_ERR_HIP (-2)
#define HEDIT_ERR_STATE (-3)

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      hedit_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
      return HEDIT_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

#define ARG_CHECK(cond, msg)                                                                  \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      hedit_set_error(std::string("bad argument: ") + (msg) + " [" #cond "]");                \
      return HEDIT_ERR_ARG;                                                                   \
    }                                                                                         \
  } while (0)

#define LAUNCH_CHECK()                                                                        \
  do {                                                                                        \
    hipError_t _e = hipGetLastError();                                                        \
    if (_e != hipSuccess) {                                                                   \
      hedit_set_error(std::string("kernel launch failed: ") + hipGetErrorString(_e) + " at " + \
                      __FILE__ + ":" + std::to_string(__LINE__));                             \
      return HEDIT_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

// ---- bfloat16 <-> f32 proper (round to nearest even), usable on host and device: the split-bf16 path and the C entries that
// are bf16 by contract in either build
__host__ __device__ inline float bf16_to_f32_always(bf16_t h) {
  union { uint32_t u; float f; } v;
  v.u = ((uint32_t)h) << 16;
  return v.f;
}
__host__ __device__ inline bf16_t f32_to_bf16_always(float f) {
  union { uint32_t u; float f; } v;
  v.f = f;
  uint32_t u = v.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// ---- storage element <-> f32 (round to nearest even), host and device
#if HEDIT_F16
__host__ __device__ inline float bf16_to_f32(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__host__ __device__ inline bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
#else
__host__ __device__ inline float bf16_to_f32(bf16_t h) { return bf16_to_f32_always(h); }
__host__ __device__ inline bf16_t f32_to_bf16(float f) { return f32_to_bf16_always(f); }
#endif
// native conversion: lets the compiler emit v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (round-to-nearest-even)
typedef __attribute__((ext_vector_type(2))) st_elem_t bf16x2_t;
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  union { bf16x2_t v; uint32_t u; } x;
  x.v = (bf16x2_t){(st_elem_t)lo, (st_elem_t)hi};
  return x.u;
}
// 1.0 in the storage format, and the storage bits of a value that is exactly representable in it (attn.hip's folded shift)
#if HEDIT_F16
constexpr bf16_t ST_ONE_BITS = 0x3C00;
__device__ __forceinline__ uint32_t st_exact_bits(float v) { return (uint32_t)__builtin_bit_cast(bf16_t, (_Float16)v); }
#else
constexpr bf16_t ST_ONE_BITS = 0x3F80;
__device__ __forceinline__ uint32_t st_exact_bits(float v) { return __builtin_bit_cast(uint32_t, v) >> 16; }
#endif
// the two storage elements of a dword as fp32 (bfloat16: one shift / one mask; half: v_cvt_f32_f16)
__device__ __forceinline__ float st_lo(uint32_t u) {
#if HEDIT_F16
  return (float)__builtin_bit_cast(bf16x2_t, u)[0];
#else
  return __builtin_bit_cast(float, u << 16);
#endif
}
__device__ __forceinline__ float st_hi(uint32_t u) {
#if HEDIT_F16
  return (float)__builtin_bit_cast(bf16x2_t, u)[1];
#else
  return __builtin_bit_cast(float, u & 0xffff0000u);
#endif
}
__device__ inline float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ inline float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// exact-GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output
// resolution of 2^-9): one v_rcp + one v_exp + a 5-term Horner chain instead of libm's erff
__device__ inline float gelu_erf_fast(float x) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float erf_abs = 1.0f - poly * e;
  const float erfv = copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erfv);
}

// v * gelu(x) for two values at once on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32).  GEGLU epilogues are
// VALU-bound (a 256 x 256 FF1 tile spends as long in this function as in its K = 320 MFMA loop), and transcendental
// instructions issue at quarter rate, so erf is taken from Abramowitz-Stegun 7.1.28,
//   erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16,  |error| <= 3e-7,
// one v_rcp per element and no v_exp (7.1.26, used by gelu_erf_fast above, needs both): |gelu error| < 1e-6 absolute,
// far below the bf16 output resolution.  The 1/sqrt(2) of z = |x| / sqrt(2) is folded into the coefficients; a
// denominator that overflows to +inf gives erf = 1, as it should.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ inline f32x2 mul_gelu2(f32x2 v, f32x2 x) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  f32x2 q = {5.3829750e-06f, 5.3829750e-06f};                                        // a6 * 2^-3
  q = __builtin_elementwise_fma(q, ax, (f32x2){4.8890634e-05f, 4.8890634e-05f});     // a5 * 2^-2.5
  q = __builtin_elementwise_fma(q, ax, (f32x2){3.8003575e-05f, 3.8003575e-05f});     // a4 * 2^-2
  q = __builtin_elementwise_fma(q, ax, (f32x2){3.2776263e-03f, 3.2776263e-03f});     // a3 * 2^-1.5
  q = __builtin_elementwise_fma(q, ax, (f32x2){2.1141006e-02f, 2.1141006e-02f});     // a2 * 2^-1
  q = __builtin_elementwise_fma(q, ax, (f32x2){4.9867347e-02f, 4.9867347e-02f});     // a1 * 2^-0.5
  q = __builtin_elementwise_fma(q, ax, (f32x2){1.f, 1.f});
  q *= q;
  q *= q;
  q *= q;
  q *= q;
  const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  // Phi(x) = 0.5 + sign(x) * 0.5 * erf|.| ,  0.5 * erf|.| = 0.5 - 0.5 * r
  const f32x2 h = __builtin_elementwise_fma(r, (f32x2){-0.5f, -0.5f}, (f32x2){0.5f, 0.5f});
  const f32x2 hs = {copysignf(h[0], x[0]), copysignf(h[1], x[1])};
  return v * x * (hs + 0.5f);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf16_to_f32((bf16_t)(v.x & 0xffff)); f[1] = bf16_to_f32((bf16_t)(v.x >> 16));
  f[2] = bf16_to_f32((bf16_t)(v.y & 0xffff)); f[3] = bf16_to_f32((bf16_t)(v.y >> 16));
  f[4] = bf16_to_f32((bf16_t)(v.z & 0xffff)); f[5] = bf16_to_f32((bf16_t)(v.z >> 16));
  f[6] = bf16_to_f32((bf16_t)(v.w & 0xffff)); f[7] = bf16_to_f32((bf16_t)(v.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

static inline int ew_grid(long work_items) {
  long g = (work_items + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
