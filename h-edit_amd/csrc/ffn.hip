// The GEGLU feed-forward of a BasicTransformerBlock as ONE kernel per row tile:
//     out = x + FF2( GEGLU( FF1( LayerNorm(x) ) ) )            (diffusers FeedForward with GEGLU; oracle/sd_unet.py)
// for the C = 320 level of the SD UNet.  Unfused this is three launches (LayerNorm, FF1+GEGLU, FF2+residual) that move
// the [M][1280] hidden activation and the normalised rows through HBM (10 passes over an [M][320] tensor where 2
// are needed) and run at HBM speed; here nothing but x and out touches HBM.
//
// gfx950 mapping -- "rows stay in registers, weights stream":
//   * a block = 4 waves (one per SIMD, up to 512 registers each), a wave owns 32 rows for the whole kernel and works
//     in v_mfma_f32_32x32x16_bf16 (weights = the 32-row operand, the wave's 32 rows = the 32-column operand):
//       Xn  LayerNorm(x) as activation fragments               20 k-steps x 4     80 AGPR
//       O   the FF2 accumulators (320 x 32 fp32)               10 blocks x 16   160 AGPR
//       S   FF1 accumulator of one (value16 | gate16) pair, double-buffered       32 VGPR
//       H   the GEGLU output of the pair = one 16-deep FF2 activation fragment     4 VGPR
//     The FF1 rows are interleaved (value16 | gate16) at load time, so value and gate of a hidden unit sit in the same
//     lane, 8 registers apart; the 8 GEGLU results of a lane ARE its activation fragment of the FF2 k-step once the
//     16 hidden units of the pair are renumbered -- a permutation folded into the FF2 weight packing.  The hidden
//     activation never leaves the register file.
//     (A one-wave-per-SIMD kernel has only its own MFMAs to hide its other instructions behind; a 32-cycle 32x32x16
//     MFMA covers about five of them where a 16-cycle 16x16x32 covers one or two -- measured)
//   * the weights (2.4 MB bf16) are packed once, at load time, into a STREAM of 10 KB units in consumption order --
//     per pair t: two units of FF1 fragments (k-steps 0-9, 10-19) and the unit of FF2 fragments of pair t-2 -- every
//     fragment stored as the lane-linear 1 KB image its ds_read_b128 wants (conflict-free).  The kernel copies units
//     global -> LDS by lane-linear DMA (buffer_load ... lds) into a ring of 15 (150 KB), four pairs ahead, with a
//     counted vmcnt and ONE barrier per pair (30 MFMAs); every wave reads every unit.  All tiles read the same stream,
//     which fits the 4 MB L2 of an XCD.
//   * per pair and wave: 20 FF1 + 10 FF2 MFMAs, 30 fragment reads, 8 DMA pieces and the 124 VALU operations of the
//     previous pair's GEGLU, dealt out 4 per MFMA.
// Every output row depends on its own input row only and the summation order is fixed, so results do not depend on
// the batch.
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int FC = 320;                 // channels
constexpr int FKS = FC / 16;            // 16-deep k-steps of FF1
constexpr int FNB = FC / 32;            // 32-wide output blocks of FF2
constexpr int FH = 4 * FC;              // hidden units
constexpr int FPAIRS = FH / 16;         // (value16 | gate16) row pairs of FF1 = 16-deep k-steps of FF2
constexpr int UNIT = 10240;             // bytes: ten 1 KB fragments
constexpr int ITER_BYTES = 3 * UNIT;    // one pair iteration: FF1 k-steps 0-9 | 10-19 | FF2 of pair t-2
constexpr int BANKS = 5, AHEAD = 4;     // ring of five iterations, DMA four ahead
constexpr int NITER = FPAIRS + 2;       // + two draining iterations (FF2 of the last two pairs)
constexpr int STREAM_ITERS = NITER + AHEAD;   // the DMA runs this far past the end (zero units)
constexpr int BIAS_OFF = BANKS * ITER_BYTES;
constexpr int BIAS_BYTES = FPAIRS * 128;       // per pair and lane half: 8 value + 8 gate biases in register order
constexpr int LDS_TOTAL = BIAS_OFF + BIAS_BYTES;
constexpr int ROWS_PER_WAVE = 32, BLOCK_ROWS = 128;
constexpr int PPW = 8;                  // DMA pieces per wave and iteration (30 pieces; the two surplus slots re-load the last)
static_assert(LDS_TOTAL <= 160 * 1024, "ring + bias table must fit the LDS");
static_assert(BIAS_BYTES % 1024 == 0, "bias table in whole DMA pieces");

// hidden unit (inside its pair) of accumulator register r (0..7) in lane half hi: the row of a 32x32 MFMA result
__host__ __device__ inline int ffn_unit_of_reg(int r, int hi) { return (r >> 2) * 8 + hi * 4 + (r & 3); }

__device__ __forceinline__ void unpack8v(const u32x4& v, float* f) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f[2 * q] = __builtin_bit_cast(float, v[q] << 16);
    f[2 * q + 1] = __builtin_bit_cast(float, v[q] & 0xffff0000u);
  }
}

// one thread per 16-byte piece of the stream: iteration it, unit u (0, 1: FF1 of pair it; 2: FF2 of pair it-2),
// fragment f (0..9), lane l -> 8 consecutive k of one weight row
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, bf16_t* __restrict__ stream) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)STREAM_ITERS * (ITER_BYTES / 16);
  if (idx >= total) return;
  const int it = (int)(idx / (ITER_BYTES / 16));
  const int off = (int)(idx - (long)it * (ITER_BYTES / 16)) * 16;
  const int u = off / UNIT, f = (off % UNIT) / 1024, l = (off % 1024) / 16;
  const int row = l & 31, hi = l >> 5;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  bool mine = true;          // does this call own the piece?  (w1 call: FF1 units and every zero unit; w2 call: FF2 units)
  if (u < 2) {
    const int t = it;
    if (t < FPAIRS) {
      if (!w1) return;
      const int ks = u * 10 + f;
      const int src = row < 16 ? t * 16 + row : FH + t * 16 + (row - 16);        // packed row: value16 | gate16
      const float* s = w1 + (long)src * FC + ks * 16 + hi * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s[e];
    } else {
      mine = w1 != nullptr;
    }
  } else {
    const int t = it - 2;
    if (t >= 0 && t < FPAIRS) {
      if (!w2) return;
      const int n = f * 32 + row;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = w2[(long)n * FH + t * 16 + ffn_unit_of_reg(e, hi)];
    } else {
      mine = w1 != nullptr;
    }
  }
  if (!mine) return;
  uint4 o = pack8(v);
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(stream) + idx * 16) = o;
}

// FF1 bias in accumulator-register order: [pair][lane half][8 value | 8 gate]
__global__ __launch_bounds__(256) void ffn_pack_bias_kernel(const float* __restrict__ b1, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= BIAS_BYTES / 4) return;
  const int t = idx >> 5, hi = (idx >> 4) & 1, r = idx & 15;
  out[idx] = r < 8 ? b1[t * 16 + ffn_unit_of_reg(r, hi)] : b1[FH + t * 16 + ffn_unit_of_reg(r - 8, hi)];
}

struct FfnKernelParams {
  const bf16_t* x; long ldx;
  const float* gamma; const float* beta; float eps;
  const bf16_t* stream;
  const float* bias1p;
  const float* bias2;
  bf16_t* out; long ldo;
  int M;
};

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// MFMAs as inline asm so that the register file of the operands is OURS to choose: the FF1 accumulators live in
// VGPRs (the GEGLU arithmetic reads them; the compiler's MFMA form would park them in AGPRs behind a v_accvgpr_read
// per value), the FF2 accumulators and the LayerNorm fragments in AGPRs (240 registers nothing but MFMAs touch until
// the epilogue).  What the compiler does not do for an asm MFMA is hazard padding, so every hazard is excluded by
// construction: an FF1 accumulator is first read by the VALU a whole iteration after its last write, an FF2
// accumulator only in the epilogue (behind s_nops and a re-definition, see there), consecutive MFMAs on one
// accumulator are the hardware's back-to-back accumulate (same destination and C), and no VALU result feeds an MFMA
// closer than several bundles (the H fragments are pinned an iteration early).
__device__ __forceinline__ void mfma_s(f32x16& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(a));
}
// first MFMA of an FF1 chain: the accumulator start (bias) is a separate, read-only operand -- a v_mov into the
// accumulator right in front of an asm MFMA would be a VALU-write -> MFMA-read hazard nobody pads.  The other way
// round (the MFMA still reading C while something overwrites it) is excluded by keeping C's registers live for
// several more MFMAs (the empty asm statement at bundle 9)
__device__ __forceinline__ void mfma_s0(f32x16& acc, const bf16x8& w, const bf16x8& a, const f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(w), "a"(a), "v"(c));
}
// first MFMA of the second FF1 chain: C = 0 (inline constant)
__device__ __forceinline__ void mfma_z(f32x16& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "a"(a));
}
__device__ __forceinline__ void mfma_o(f32x16& acc, const bf16x8& w, const bf16x8& h) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(h));
}

// GEGLU of one FF1 pair = 8 elements per lane (registers e and e + 8 of the pair's accumulator are value and gate of
// hidden unit ffn_unit_of_reg(e, lane half)), as a list of 124 single VALU operations in stage-major order
// (operation k: stage k / 8 of element k % 8, then four bf16 packs), so that the kernel can hand them out a few per
// MFMA.  With v = value, x = gate:
//   v gelu(x),   gelu(x) = x Phi(x) = max(x, 0) - |x| r(|x|),   r(z) = erfc(z / sqrt 2) / 2 = q(z)^-16,
// q a degree-5 polynomial (the Abramowitz-Stegun 7.1.28 form, refitted with the 1/2 and the 1/sqrt 2 folded in:
// |gelu error| < 5e-6 absolute, three decimal orders below the bf16 resolution of the result; tools/gelu_fit.py).
// Plain fp32 operations: packed fp32 VALU is slow beside MFMAs.  No cancellation anywhere: for large |x| the
// second term vanishes (q^16 overflows to +inf, 1/inf = 0).
struct Gelu8 {
  float x[8], q[8], g[8];
  uint32_t h[4];       // the FF2 activation fragment: h[j] = bf16 pair of elements 2j, 2j + 1
};
constexpr int GELU_OPS = 15 * 8 + 4;
// (the empty volatile asm pins each result where it is written: instruction selection otherwise sinks an operation
//  down to its consumer, out of the bundle it was meant to fill; the operation itself stays compiler-visible, so its
//  hazards and waits are the compiler's business)
#define GELU_PIN(v) asm volatile("" : "+v"(v))
// The FF1 result of a pair arrives as TWO partial accumulators (even / odd k-steps, see the kernel): value and gate
// are their sums, one more add each.
template <int K>
__device__ __forceinline__ void gelu_op(Gelu8& g, const f32x16& Se, const f32x16& So) {
  if constexpr (K < 120) {
    constexpr int st = K / 8, e = K % 8;
    if constexpr (st == 0) { g.x[e] = Se[e + 8] + So[e + 8]; GELU_PIN(g.x[e]); }
    else if constexpr (st <= 10) {
      const float x = g.x[e];
      if constexpr (st == 1) g.q[e] = __builtin_fmaf(9.73478169e-05f, __builtin_fabsf(x), -1.02176718e-04f);
      else if constexpr (st == 2) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 3.62392071e-03f);
      else if constexpr (st == 3) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 2.19443815e-02f);
      else if constexpr (st == 4) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 5.21099924e-02f);
      else if constexpr (st == 5) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 1.04427174e+00f);
      else if constexpr (st <= 9) g.q[e] = g.q[e] * g.q[e];
      else g.q[e] = __builtin_amdgcn_rcpf(g.q[e]);
      GELU_PIN(g.q[e]);
    } else {
      if constexpr (st == 11) g.g[e] = __builtin_fmaxf(g.x[e], 0.f);
      else if constexpr (st == 12) g.g[e] = __builtin_fmaf(-__builtin_fabsf(g.x[e]), g.q[e], g.g[e]);
      else if constexpr (st == 13) { g.q[e] = Se[e] + So[e]; GELU_PIN(g.q[e]); }
      else g.g[e] = g.g[e] * g.q[e];
      GELU_PIN(g.g[e]);
    }
  } else {
    constexpr int p = K - 120;
    g.h[p] = pack_bf16x2(g.g[2 * p], g.g[2 * p + 1]);
    GELU_PIN(g.h[p]);
  }
}
// operations [lo, hi) of the list
template <int LO, int HI>
__device__ __forceinline__ void gelu_ops(Gelu8& g, const f32x16& Se, const f32x16& So) {
  static_for<HI - LO>([&](auto k) { gelu_op<LO + decltype(k)::value>(g, Se, So); });
}

__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnKernelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 31, hi = lane >> 5;
  const int m_wave = blockIdx.x * BLOCK_ROWS + wave * ROWS_PER_WAVE;

#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.stream), (short)0, (int)(STREAM_ITERS * ITER_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias1p), (short)0, (int)BIAS_BYTES, 0x00020000);
#endif
  // DMA piece k (0..7) of this wave for stream iteration `it` -> ring bank `bank`: 1 KB piece q = wave + 4 k of the
  // iteration's 30 (k = 7 of the waves 2, 3 would be pieces 30, 31: they re-load piece 29 -- same bytes, same place)
  unsigned dma_voff[2];
  dma_voff[0] = (unsigned)(wave * 1024 + lane * 16);
  dma_voff[1] = (unsigned)((wave + 28 > 29 ? 29 : wave + 28) * 1024 + lane * 16);
  const int dma_last = (wave + 28 > 29 ? 29 : wave + 28) * 1024;
  auto dma_piece = [&](int it, int bank, int k) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (k < 7)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + bank * ITER_BYTES + wave * 1024 + k * 4096), 16,
                                               dma_voff[0], it * ITER_BYTES + k * 4096, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + bank * ITER_BYTES + dma_last), 16, dma_voff[1],
                                               it * ITER_BYTES, 0, 0);
#else
    (void)it; (void)bank; (void)k;
#endif
  };

  // ---- prologue: start the weight stream, then LayerNorm this wave's rows into fragment registers
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = wave; i < BIAS_BYTES / 1024; i += 4)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(smem + BIAS_OFF + i * 1024), 16,
                                             (unsigned)(i * 1024 + lane * 16), 0, 0, 0);
#endif
#pragma unroll
  for (int it = 0; it < AHEAD; ++it)
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_piece(it, it, k);

  bf16x8 Xn[FKS];        // lane (row lm, half hi): LayerNorm(x)[row][16 ks + 8 hi .. + 7]
  {
    u32x4 raw[FKS];
    int row = m_wave + lm;
    if (row > p.M - 1) row = p.M - 1;
    const bf16_t* src = p.x + (long)row * p.ldx + hi * 8;
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) raw[ks] = *reinterpret_cast<const u32x4*>(src + ks * 16);
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      float f[8];
      unpack8v(raw[ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[e];
    }
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)FC;
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      float f[8];
      unpack8v(raw[ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
    }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q / (float)FC + p.eps);
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      const float4* g4 = reinterpret_cast<const float4*>(p.gamma + ks * 16 + hi * 8);
      const float4* b4 = reinterpret_cast<const float4*>(p.beta + ks * 16 + hi * 8);
      const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float f[8], o[8];
      unpack8v(raw[ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f[e] - mean) * rstd * gg[e] + bb[e];
      const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      Xn[ks] = __builtin_bit_cast(bf16x8, pk);
      asm volatile("" : "+a"(Xn[ks]));       // home in the AGPR file from here on (every use is an MFMA operand)
    }
  }

  f32x16 O[FNB];
#pragma unroll
  for (int nb = 0; nb < FNB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");           // iterations 0 .. AHEAD-1 and the biases are in LDS

  // One pair iteration = 30 "bundles" of [1 MFMA | 3-4 GEGLU operations | a fragment read LEAD MFMAs ahead (the last
  // LEAD reads fetch the first fragments of the next iteration) | eight of them a DMA piece], pinned by sched_barrier so
  // that the VALU and LDS work sits in the MFMAs' shadow instead of in a block of its own.  MFMA order: two FF1
  // k-steps, one FF2 output block, ten times.  At the iteration boundary: [my DMA pieces of the iteration after next
  // have landed: vmcnt(2 iterations in flight)] [my reads of the iteration just finished have returned: lgkmcnt(the
  // LEAD newest = next iteration's)] barrier; the finished iteration's bank is then refilled during the next one.
  constexpr int LEAD = 8;
  constexpr int NB = 30;
  int it = 0, bank = 0;                      // stream iteration being consumed, its ring bank
  int frag_rd = lane * 16;                   // LDS address of this lane's 16 bytes in fragment 0 of the current bank
  int bias_rd = BIAS_OFF + hi * 64;          // accumulator start of the NEXT FF1 pair to be fetched
  auto boundary = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"((AHEAD - 2) * PPW), "n"(LEAD) : "memory");
  };
  // fragment j (0..29) of an iteration in MFMA order: j % 3 < 2 -> FF1 k-step 2 (j / 3) + j % 3; else FF2 block j / 3
  auto frag_off = [](int j) __attribute__((always_inline)) {
    return (j % 3 < 2) ? (2 * (j / 3) + j % 3) * 1024 : 2 * UNIT + (j / 3) * 1024;
  };
  auto rd = [&](int base, int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(smem + (base + frag_off(j)));
  };
  auto rdb = [&]() __attribute__((always_inline)) {
    f32x16 b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (bias_rd + q * 16));
      b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3];
    }
    return b;
  };

  // FF1 accumulators of even / odd pairs, each as TWO chains (even / odd k-steps): a dependent MFMA can issue only when
  // its predecessor has left the pipeline (about two issue slots of a 32x32x16), so one chain of 20 back-to-back
  // accumulations would run at 2/3 of the MFMA rate; with two chains and the FF2 MFMA in between, every chain is
  // revisited each third MFMA.  The GEGLU adds the two halves.
  f32x16 Sa, Sb, Sa2, Sb2;
  u32x4 Ha, Hb;                              // FF2 activation fragments (GEGLU of even / odd pairs)
  bf16x8 pre[LEAD];                          // first fragments of the iteration about to start
  f32x16 binit;                              // accumulator start of the FF1 pair about to start
#pragma unroll
  for (int r = 0; r < 16; ++r) { Sa[r] = 0.f; Sb[r] = 0.f; Sa2[r] = 0.f; Sb2[r] = 0.f; }
  Ha = (u32x4){0u, 0u, 0u, 0u};
  Hb = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < LEAD; ++j) pre[j] = rd(frag_rd, j);
  binit = rdb();
  bias_rd += 128;

  // iteration t: FF1 of pair t -> Scur; GEGLU of pair t-1 (Sprev) -> Hnew; FF2 of pair t-2 with Hold.  F1 / GELU false
  // in the two draining iterations.  Hnew and Hold are the same variable: the fragment of pair t-2 has been consumed by
  // the time the GEGLU of pair t (same parity) writes it, an iteration later -- Hnew is assembled at the very end.
  auto iteration = [&](auto f1_c, auto gelu_c, f32x16& Scur, f32x16& Scur2, const f32x16& Sprev, const f32x16& Sprev2, u32x4& Hgelu,
                       const u32x4& Hff2) __attribute__((always_inline)) {
    constexpr bool F1 = decltype(f1_c)::value, GELU = decltype(gelu_c)::value;
    const int base = frag_rd;
    const int nbank = bank == BANKS - 1 ? 0 : bank + 1;
    const int pbank = bank == 0 ? BANKS - 1 : bank - 1;
    const int nbase = lane * 16 + nbank * ITER_BYTES;
    Gelu8 G;
    bf16x8 fr[NB + LEAD];
#pragma unroll
    for (int j = 0; j < LEAD; ++j) fr[j] = pre[j];
    f32x16 bnext;
    static_for<NB>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      // one counted wait per three MFMAs instead of the compiler's one per MFMA: the fragments of bundles b .. b+2 have
      // arrived when at most the LEAD - 3 youngest LDS reads are outstanding (s_waitcnt lgkmcnt only: vmcnt / expcnt at max)
      if constexpr (b % 3 == 0) __builtin_amdgcn_s_waitcnt(0xC07F | ((LEAD - 3) << 8));
      if constexpr (b % 3 < 2) {
        constexpr int ks = 2 * (b / 3) + b % 3;
        if constexpr (F1) {
          if constexpr (ks == 0) mfma_s0(Scur, fr[b], Xn[ks], binit);
          else if constexpr (ks == 1) mfma_z(Scur2, fr[b], Xn[ks]);
          else if constexpr (ks % 2 == 0) mfma_s(Scur, fr[b], Xn[ks]);
          else mfma_s(Scur2, fr[b], Xn[ks]);
        }
      } else {
        mfma_o(O[b / 3], fr[b], __builtin_bit_cast(bf16x8, Hff2));
      }
      if constexpr (F1 && b == 9) asm volatile("" ::"v"(binit));        // (see mfma_s0: keeps the C operand's registers intact)
      if constexpr (b + LEAD < NB) fr[b + LEAD] = rd(base, b + LEAD);
      else pre[b + LEAD - NB] = rd(nbase, b + LEAD - NB);
      if constexpr (b >= 10 && b < 14) {                                  // accumulator start of the next pair, 4 x 16 bytes
        constexpr int q = b - 10;
        const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (bias_rd + q * 16));
        bnext[4 * q] = v[0]; bnext[4 * q + 1] = v[1]; bnext[4 * q + 2] = v[2]; bnext[4 * q + 3] = v[3];
      }
      if constexpr (b % 3 == 1 && b / 3 < PPW) dma_piece(it + AHEAD, pbank, b / 3);
      if constexpr (GELU) gelu_ops<(b * GELU_OPS) / NB, ((b + 1) * GELU_OPS) / NB>(G, Sprev, Sprev2);
      __builtin_amdgcn_sched_barrier(0);
    });
    boundary();
    if constexpr (GELU) {
      Hgelu = (u32x4){G.h[0], G.h[1], G.h[2], G.h[3]};
      asm volatile("" : "+v"(Hgelu));
    }
    binit = bnext;
    if (it + 2 < FPAIRS) bias_rd += 128;       // (stay inside the table in the last iterations; their fetch is not used)
    ++it;
    bank = nbank;
    frag_rd = nbase;
  };
  using T = std::true_type;
  using F = std::false_type;
  for (int tt = 0; tt < FPAIRS / 2; ++tt) {
    iteration(T{}, T{}, Sa, Sa2, Sb, Sb2, Hb, Ha);     // even pair t: FF1 -> Sa; GEGLU of pair t-1 (Sb) -> Hb; FF2 of pair t-2 (Ha)
    iteration(T{}, T{}, Sb, Sb2, Sa, Sa2, Ha, Hb);     // odd pair
  }
  iteration(F{}, T{}, Sa, Sa2, Sb, Sb2, Hb, Ha);       // GEGLU of the last pair, FF2 of the one before
  iteration(F{}, F{}, Sb, Sb2, Sa, Sa2, Ha, Hb);       // FF2 of the last pair

  // ---- epilogue: bf16(O + bias2), staged per wave in LDS (the ring is free once every wave is here), then whole rows:
  // + x (residual) in fp32, rounded again, 16-byte stores.  (The s_nops: the last asm MFMAs must have retired before
  // the first v_accvgpr_read -- an MFMA-write -> VALU-read hazard the compiler cannot see; the empty asm statements
  // re-define every accumulator behind them, so no read can be scheduled up into the MFMA stream.)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int nb = 0; nb < FNB; ++nb) asm volatile("" : "+a"(O[nb]));
  constexpr int PITCH = FC * 2 + 16;                     // bytes per staged row
  constexpr int STAGE = 24576;
  static_assert(ROWS_PER_WAVE * PITCH <= STAGE && 4 * STAGE <= BIAS_OFF, "epilogue staging must fit the ring");
  char* stage = smem + wave * STAGE;
#pragma unroll
  for (int nb = 0; nb < FNB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nb * 32 + q * 8 + hi * 4;            // registers 4q .. 4q+3 = output features n .. n+3 of row lm
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.bias2 + n);
      u32x2 o;
      o[0] = pack_bf16x2(O[nb][4 * q] + b2[0], O[nb][4 * q + 1] + b2[1]);
      o[1] = pack_bf16x2(O[nb][4 * q + 2] + b2[2], O[nb][4 * q + 3] + b2[3]);
      *reinterpret_cast<u32x2*>(stage + lm * PITCH + n * 2) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (a wave reads back only what it wrote itself)
  constexpr int CPR = FC / 8;                            // 16-byte pieces per row
  constexpr int ITER = ROWS_PER_WAVE * CPR / 64;
  static_assert(ROWS_PER_WAVE * CPR % 64 == 0, "rows must divide evenly over the lanes");
  auto add2 = [](uint32_t a, uint32_t b) __attribute__((always_inline)) {
    return pack_bf16x2(bf16_to_f32((bf16_t)(a & 0xffff)) + bf16_to_f32((bf16_t)(b & 0xffff)),
                       bf16_to_f32((bf16_t)(a >> 16)) + bf16_to_f32((bf16_t)(b >> 16)));
  };
  u32x4 res[ITER], ov[ITER];
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = lane + i * 64;
    const int rl = idx / CPR, c = idx - rl * CPR;
    int row = m_wave + rl;
    if (row > p.M - 1) row = p.M - 1;
    res[i] = *reinterpret_cast<const u32x4*>(p.x + (long)row * p.ldx + c * 8);
    ov[i] = *reinterpret_cast<const u32x4*>(stage + rl * PITCH + c * 16);
  }
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = lane + i * 64;
    const int rl = idx / CPR, c = idx - rl * CPR;
    const int row = m_wave + rl;
    u32x4 o;
    o[0] = add2(ov[i][0], res[i][0]);
    o[1] = add2(ov[i][1], res[i][1]);
    o[2] = add2(ov[i][2], res[i][2]);
    o[3] = add2(ov[i][3], res[i][3]);
    if (row < p.M) *reinterpret_cast<u32x4*>(p.out + (long)row * p.ldo + c * 8) = o;
  }
}

}  // namespace

int ffn_fused_channels() { return FC; }
size_t ffn_stream_bytes() { return (size_t)STREAM_ITERS * ITER_BYTES; }
size_t ffn_bias_bytes() { return BIAS_BYTES; }

int ffn_pack_launch(const float* w1, const float* w2, bf16_t* stream, hipStream_t st) {
  const long total = (long)STREAM_ITERS * (ITER_BYTES / 16);
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w1, w2, stream);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int ffn_pack_bias_launch(const float* b1, float* out, hipStream_t st) {
  hipLaunchKernelGGL(ffn_pack_bias_kernel, dim3(cdiv(BIAS_BYTES / 4, 256)), dim3(256), 0, st, b1, out);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int ffn_fused_launch(const FfnParams& f, hipStream_t st) {
  ARG_CHECK(f.C == FC, "ffn: the fused feed-forward exists for C = 320");
  ARG_CHECK(f.M > 0 && f.ldx % 8 == 0 && f.ldo % 8 == 0, "ffn: rows must be 16-byte aligned");
  ARG_CHECK(f.x && f.out && f.stream && f.bias1p && f.bias2 && f.gamma && f.beta, "ffn: null");
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    attr_set = true;
  }
  FfnKernelParams k;
  k.x = f.x; k.ldx = f.ldx; k.gamma = f.gamma; k.beta = f.beta; k.eps = f.eps;
  k.stream = f.stream; k.bias1p = f.bias1p; k.bias2 = f.bias2; k.out = f.out; k.ldo = f.ldo; k.M = f.M;
  hipLaunchKernelGGL(ffn_fused_kernel, dim3(cdiv(f.M, BLOCK_ROWS)), dim3(256), LDS_TOTAL, st, k);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
