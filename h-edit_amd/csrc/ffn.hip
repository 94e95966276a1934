// The token-local tail of a BasicTransformerBlock at the C = 320 level of the SD UNet as ONE kernel per row tile:
//
//     t2  = attn2.to_out(a) + t1                       [optional leading linear layer + residual]
//     t3  = t2 + FF2( GEGLU( FF1( LayerNorm(t2) ) ) )  (diffusers FeedForward with GEGLU; oracle/sd_unet.py)
//     out = proj_out(t3) + x                           [optional trailing linear layer + residual]
//
// Unfused this is five launches (to_out GEMM, LayerNorm, FF1+GEGLU, FF2+residual, proj_out GEMM) that move the
// [M][1280] hidden activation and four [M][320] intermediates through HBM and run at HBM speed; here only a, t1, x
// and out touch HBM.  hedit_k_ffn_fused is the middle line alone.
//
// gfx950 mapping -- "rows stay in registers, weights stream":
//   * a block = 4 waves (one per SIMD, up to 512 registers each), a wave owns 32 rows for the whole kernel and works
//     in v_mfma_f32_32x32x16_bf16 (weights = the 32-row operand, the wave's 32 rows = the 32-column operand):
//       Xn  the activation fragments of the running layer      20 k-steps x 4     80 AGPR
//       O   the 320 x 32 fp32 accumulator of the row tile      10 blocks x 16   160 AGPR
//       S   FF1 accumulators of one (value16 | gate16) pair, two chains, double-buffered  64 VGPR
//       H   the GEGLU output of the pair = one 16-deep FF2 activation fragment     4 VGPR
//     O carries the residual stream: it starts as t1 (+ bias), the leading linear accumulates onto it, LayerNorm reads
//     it, FF2 accumulates onto it, and the trailing linear starts a new one from x.  An accumulator in MFMA result
//     layout IS the activation-fragment layout of the next layer once that layer's input features are renumbered
//     inside their groups of 16 -- a permutation folded into the weight packing (ffn_unit_of_reg) -- so no
//     activation ever leaves the register file between the layers.  The FF1 rows are interleaved (value16 | gate16)
//     at load time, so value and gate of a hidden unit sit in the same lane, 8 registers apart.
//     (A one-wave-per-SIMD kernel has only its own MFMAs to hide its other instructions behind: DESIGN.md section 5.)
//   * the weights (2.4 - 2.9 MB bf16) are packed once, at load time, into a STREAM of 30 KB iterations in consumption
//     order -- linear layers: 30 fragments (k-step major, output block minor); feed-forward pair t: 20 FF1 fragments
//     and the 10 FF2 fragments of pair t-2 -- every fragment stored as the lane-linear 1 KB image its ds_read_b128
//     wants (conflict-free).  The kernel copies iterations global -> LDS by lane-linear DMA (buffer_load ... lds) into
//     a ring of five (150 KB), four ahead, with a counted vmcnt and ONE barrier per iteration (30 MFMAs); every wave
//     reads every fragment.  All tiles read the same stream, which fits the 4 MB L2 of an XCD.
//   * per feed-forward pair and wave: 20 FF1 + 10 FF2 MFMAs, 30 fragment reads, 8 DMA pieces and the 124 VALU
//     operations of the previous pair's GEGLU, dealt out 4 per MFMA.
// Every output row depends on its own input rows only and the summation order is fixed, so results do not depend on
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
constexpr int NITER = FPAIRS + 2;       // feed-forward: + two draining iterations (FF2 of the last two pairs)
constexpr int LIN_ITERS = 7;            // a 320 -> 320 linear layer: 200 fragments = 6 x 30 + 20
constexpr int LIN_FRAGS = FKS * FNB;
// stream sections: [leading linear][feed-forward][trailing linear][AHEAD iterations of zeros the DMA runs into]
__host__ __device__ constexpr int stream_iters(bool pre, bool post) { return (pre ? LIN_ITERS : 0) + NITER + (post ? LIN_ITERS : 0) + AHEAD; }
constexpr int BIAS_OFF = BANKS * ITER_BYTES;
constexpr int BIAS_BYTES = FPAIRS * 128;       // per pair and lane half: 8 value + 8 gate biases in register order
constexpr int LDS_TOTAL = BIAS_OFF + BIAS_BYTES;
constexpr int ROWS_PER_WAVE = 32, BLOCK_ROWS = 128;
constexpr int PPW = 8;                  // DMA pieces per wave and iteration (30 pieces; the two surplus slots re-load the last)
static_assert(LDS_TOTAL <= 160 * 1024, "ring + bias table must fit the LDS");
static_assert(LIN_FRAGS == 6 * 30 + 20, "linear layer = six full iterations + 20 fragments");
static_assert(BIAS_BYTES % 1024 == 0, "bias table in whole DMA pieces");

// hidden unit (inside its pair) of accumulator register r (0..7) in lane half hi: the row of a 32x32 MFMA result
__host__ __device__ inline int ffn_unit_of_reg(int r, int hi) { return (r >> 2) * 8 + hi * 4 + (r & 3); }

__device__ __forceinline__ void unpack8v(const u32x4& v, float* f) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f[2 * q] = st_lo(v[q]);
    f[2 * q + 1] = st_hi(v[q]);
  }
}

// one thread per 16-byte piece of the stream.  which: 0 = leading linear (w [C][C], natural k order: its input comes
// from HBM), 1 = FF1 (w [8C][C]), 2 = FF2 (w [C][4C]), 3 = trailing linear (w [C][C]); every layer whose input is an
// accumulator of this kernel (FF1, FF2, trailing linear) takes its k in register order (ffn_unit_of_reg).  The call
// for `which` = 1 also zeroes everything that belongs to no layer (pad fragments, draining iterations, DMA overrun).
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w, int which, int pre, int post, bf16_t* __restrict__ stream) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)stream_iters(pre, post) * (ITER_BYTES / 16);
  if (idx >= total) return;
  const int it = (int)(idx / (ITER_BYTES / 16));
  const int off = (int)(idx - (long)it * (ITER_BYTES / 16)) * 16;
  const int j = off / 1024, l = (off % 1024) / 16;            // fragment of the iteration (stored order), lane
  const int row = l & 31, hi = l >> 5;
  const int ff0 = pre ? LIN_ITERS : 0, post0 = ff0 + NITER;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  int owner = 1;             // which call writes this piece
  if (it < ff0 || (it >= post0 && it < post0 + (post ? LIN_ITERS : 0))) {
    const bool lead = it < ff0;
    const int jg = (lead ? it : it - post0) * 30 + j;         // fragment of the layer: k-step jg / 10, output block jg % 10
    if (jg < LIN_FRAGS) {
      owner = lead ? 0 : 3;
      if (which == owner) {
        const int ks = jg / FNB, nb = jg % FNB;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[(long)(nb * 32 + row) * FC + ks * 16 + (lead ? hi * 8 + e : ffn_unit_of_reg(e, hi))];
      }
    }
  } else if (it < post0) {
    const int t = it - ff0;
    if (j < 20) {                                             // FF1 fragment: k-step j of pair t
      if (t < FPAIRS) {
        owner = 1;
        if (which == 1) {
          const int src = row < 16 ? t * 16 + row : FH + t * 16 + (row - 16);        // packed row: value16 | gate16
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = w[(long)src * FC + j * 16 + ffn_unit_of_reg(e, hi)];
        }
      }
    } else if (t - 2 >= 0 && t - 2 < FPAIRS) {                // FF2 fragment: output block j - 20 of pair t - 2
      owner = 2;
      if (which == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[(long)((j - 20) * 32 + row) * FH + (t - 2) * 16 + ffn_unit_of_reg(e, hi)];
      }
    }
  }
  if (which != owner) return;
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

struct ChainKernelParams {
  const bf16_t* a; long lda;        // PRE: input rows of the leading linear layer
  const bf16_t* r1; long ldr1;      // PRE: residual added to it; else: the feed-forward's input rows (LayerNorm input and residual)
  const bf16_t* r2; long ldr2;      // POST: residual added to the trailing linear layer
  const float* bias_pre;
  const float* gamma; const float* beta; float eps;
  const bf16_t* stream;
  const float* bias1p;
  const float* bias2;
  const float* bias_post;
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
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(a));
}
// first MFMA of an FF1 chain: the accumulator start (bias) is a separate, read-only operand -- a v_mov into the
// accumulator right in front of an asm MFMA would be a VALU-write -> MFMA-read hazard nobody pads.  The other way
// round (the MFMA still reading C while something overwrites it) is excluded by keeping C's registers live for
// several more MFMAs (the empty asm statement at bundle 9)
__device__ __forceinline__ void mfma_s0(f32x16& acc, const bf16x8& w, const bf16x8& a, const f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, %3" : "=&v"(acc) : "v"(w), "a"(a), "v"(c));
}
// first MFMA of the second FF1 chain: C = 0 (inline constant)
__device__ __forceinline__ void mfma_z(f32x16& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "a"(a));
}
__device__ __forceinline__ void mfma_o(f32x16& acc, const bf16x8& w, const bf16x8& h) {
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(h));
}
// linear layers: accumulator AND activation fragment in AGPRs
__device__ __forceinline__ void mfma_l(f32x16& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, %0" : "+a"(acc) : "v"(w), "a"(a));
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

// DRAIN: the test twin of the wait schedule (tests/test_gpu_ring_hazard.py) -- every ring hand-over waits vmcnt(0) lgkmcnt(0)
// and every in-iteration fragment wait lgkmcnt(0); same arithmetic, same order.
template <bool PRE, bool POST, bool DRAIN = false>
__global__ __launch_bounds__(256, 1) void ffn_chain_kernel(ChainKernelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 31, hi = lane >> 5;
  const int m_wave = blockIdx.x * BLOCK_ROWS + wave * ROWS_PER_WAVE;
  int row_c = m_wave + lm;                                   // this lane's row, clamped for the loads
  if (row_c > p.M - 1) row_c = p.M - 1;

#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.stream), (short)0, (int)(stream_iters(PRE, POST) * ITER_BYTES), 0x00020000);
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

  // ---- prologue: start the weight stream; the accumulator starts as the residual stream, the leading layer's input
  // rows go into fragment registers
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = wave; i < BIAS_BYTES / 1024; i += 4)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(smem + BIAS_OFF + i * 1024), 16,
                                             (unsigned)(i * 1024 + lane * 16), 0, 0, 0);
#endif
#pragma unroll
  for (int it = 0; it < AHEAD; ++it)
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_piece(it, it, k);

  // O[nb][4 q + j] of lane (lm, hi) = feature 32 nb + 8 q + 4 hi + j of row lm  (32x32 MFMA result layout)
  f32x16 O[FNB];
  // O = rows of `src` (bf16) + bias.  A lane owns one row, so every global access is its own request: 16-byte loads
  // (chunk 2k + hi of the row, 20 per lane) instead of the 40 8-byte pieces the accumulator layout asks for, and one
  // exchange between the lane halves puts the pieces where they belong (the lane of half `hi` needs half `hi` of EVERY
  // chunk: it keeps that half of its own chunks and swaps the other for the partner lane's).
  auto load_rows = [&](const bf16_t* src, long ld, const float* bias) __attribute__((always_inline)) {
    const bf16_t* rp = src + (long)row_c * ld + hi * 8;
    u32x4 raw[FKS];
#pragma unroll
    for (int k = 0; k < FKS; ++k) raw[k] = *reinterpret_cast<const u32x4*>(rp + k * 16);
#pragma unroll
    for (int k = 0; k < FKS; ++k) {
      // raw[k] = chunk 2k + hi = [low half: dwords 0, 1 | high half: dwords 2, 3]; after the swaps: x = this lane's half of
      // chunk 2k, y = its half of chunk 2k + 1
      const auto s0 = __builtin_amdgcn_permlane32_swap(raw[k][0], raw[k][2], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(raw[k][1], raw[k][3], false, false);
      const unsigned half[2][2] = {{s0[0], s1[0]}, {s0[1], s1[1]}};
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int c = 2 * k + c2, nb = c / 4, q = c % 4;
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (bias) bb = *reinterpret_cast<const f32x4*>(bias + nb * 32 + q * 8 + hi * 4);
        O[nb][4 * q] = st_lo(half[c2][0]) + bb[0];
        O[nb][4 * q + 1] = st_hi(half[c2][0]) + bb[1];
        O[nb][4 * q + 2] = st_lo(half[c2][1]) + bb[2];
        O[nb][4 * q + 3] = st_hi(half[c2][1]) + bb[3];
      }
    }
  };
  // make the accumulators readable by the VALU behind asm MFMAs: the last MFMAs must have retired before the first
  // v_accvgpr_read (an MFMA-write -> VALU-read hazard the compiler cannot see), and the empty asm statements re-define
  // every accumulator behind the s_nops so that no read can be scheduled up into the MFMA stream
  auto settle = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int nb = 0; nb < FNB; ++nb) asm volatile("" : "+a"(O[nb]));
  };
  // and the other way round: VALU-written accumulators / fragments in front of asm MFMAs
  auto publish = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < FNB; ++nb) asm volatile("" : "+a"(O[nb]));
    asm volatile("s_nop 3" ::: "memory");
  };

  bf16x8 Xn[FKS];        // lane (row lm, half hi): activation fragment of k-step ks (8 consecutive k, or 8 registers of O)
  load_rows(p.r1, p.ldr1, PRE ? p.bias_pre : nullptr);
  if constexpr (PRE) {
    const bf16_t* src = p.a + (long)row_c * p.lda + hi * 8;
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(src + ks * 16);
      Xn[ks] = __builtin_bit_cast(bf16x8, raw);
      asm volatile("" : "+a"(Xn[ks]));       // home in the AGPR file from here on (every use is an MFMA operand)
    }
  }
  // Xn = bf16 of the accumulator (the next layer's k order is the register order)
  auto frags_from_acc = [&](auto ln_c, float mean, float rstd) __attribute__((always_inline)) {
    constexpr bool LN = decltype(ln_c)::value;
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      float o[8];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int q = 2 * (ks % 2) + h2;
        f32x4 gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LN) {
          gg = *reinterpret_cast<const f32x4*>(p.gamma + (ks / 2) * 32 + q * 8 + hi * 4);
          bb = *reinterpret_cast<const f32x4*>(p.beta + (ks / 2) * 32 + q * 8 + hi * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = O[ks / 2][4 * q + j];
          o[4 * h2 + j] = LN ? (v - mean) * rstd * gg[j] + bb[j] : v;
        }
      }
      const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      Xn[ks] = __builtin_bit_cast(bf16x8, pk);
      asm volatile("" : "+a"(Xn[ks]));
    }
  };

  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");           // iterations 0 .. AHEAD-1 and the biases are in LDS

  // One iteration = 30 "bundles" of [1 MFMA | (feed-forward) 4 GEGLU operations | a fragment read LEAD MFMAs ahead (the
  // last LEAD reads fetch the first fragments of the next iteration) | eight of them a DMA piece], pinned by
  // sched_barrier so that the VALU and LDS work sits in the MFMAs' shadow instead of in a block of its own.  At the
  // iteration boundary: [my DMA pieces of the iteration after next have landed: vmcnt(2 iterations in flight)] [my
  // reads of the iteration just finished have returned: lgkmcnt(the LEAD newest = next iteration's)] barrier; the
  // finished iteration's bank is then refilled during the next one.
  constexpr int LEAD = 8;
  constexpr int NB = 30;
  int it = 0, bank = 0;                      // stream iteration being consumed, its ring bank
  int frag_rd = lane * 16;                   // LDS address of this lane's 16 bytes in fragment 0 of the current bank
  int bias_rd = BIAS_OFF + hi * 64;          // accumulator start of the NEXT FF1 pair to be fetched
  auto boundary = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"(DRAIN ? 0 : (AHEAD - 2) * PPW), "n"(DRAIN ? 0 : LEAD) : "memory");
  };
  // LDS offset of the fragment MFMA j of an iteration uses.  Linear layer: stored in MFMA order.  Feed-forward (MFMA
  // order: two FF1 k-steps, one FF2 output block, ten times): FF1 k-steps 0..19, then the FF2 blocks.
  auto frag_off = [](bool ff, int j) __attribute__((always_inline)) {
    return !ff ? j * 1024 : ((j % 3 < 2) ? (2 * (j / 3) + j % 3) * 1024 : 2 * UNIT + (j / 3) * 1024);
  };
  auto rd = [&](int base, bool ff, int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(smem + (base + frag_off(ff, j)));
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
  bf16x8 pre[LEAD];                          // first fragments of the iteration about to start
#pragma unroll
  for (int j = 0; j < LEAD; ++j) pre[j] = rd(frag_rd, !PRE, j);
  auto advance = [&](int nbank, int nbase) __attribute__((always_inline)) {
    ++it;
    bank = nbank;
    frag_rd = nbase;
  };

  // ---- a linear layer: iteration IDX (0..6) of O[nb] += W[32 nb + ., k-step] Xn[k-step], fragment jg = 30 IDX + b =
  // 10 ks + nb; the ten accumulators are visited round-robin (no dependent MFMAs back to back)
  auto lin_iteration = [&](auto idx_c, auto next_ff_c) __attribute__((always_inline)) {
    constexpr int IDX = decltype(idx_c)::value;
    constexpr bool NEXT_FF = decltype(next_ff_c)::value;
    constexpr int CNT = IDX < LIN_ITERS - 1 ? NB : LIN_FRAGS - (LIN_ITERS - 1) * NB;
    const int base = frag_rd;
    const int nbank = bank == BANKS - 1 ? 0 : bank + 1;
    const int pbank = bank == 0 ? BANKS - 1 : bank - 1;
    const int nbase = lane * 16 + nbank * ITER_BYTES;
    bf16x8 fr[NB + LEAD];
#pragma unroll
    for (int j = 0; j < LEAD; ++j) fr[j] = pre[j];
    static_for<CNT>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      constexpr int jg = IDX * NB + b;
      if constexpr (b % 3 == 0) __builtin_amdgcn_s_waitcnt(0xC07F | ((DRAIN ? 0 : LEAD - 3) << 8));
      mfma_l(O[jg % FNB], fr[b], Xn[jg / FNB]);
      if constexpr (b + LEAD < CNT) fr[b + LEAD] = rd(base, false, b + LEAD);
      else pre[b + LEAD - CNT] = rd(nbase, NEXT_FF, b + LEAD - CNT);
      if constexpr (b % 2 == 1 && b / 2 < PPW) dma_piece(it + AHEAD, pbank, b / 2);
      __builtin_amdgcn_sched_barrier(0);
    });
    boundary();
    advance(nbank, nbase);
  };
  auto linear_layer = [&](auto next_ff_c) __attribute__((always_inline)) {
    using I = std::false_type;
    lin_iteration(std::integral_constant<int, 0>{}, I{});
    lin_iteration(std::integral_constant<int, 1>{}, I{});
    lin_iteration(std::integral_constant<int, 2>{}, I{});
    lin_iteration(std::integral_constant<int, 3>{}, I{});
    lin_iteration(std::integral_constant<int, 4>{}, I{});
    lin_iteration(std::integral_constant<int, 5>{}, I{});
    lin_iteration(std::integral_constant<int, 6>{}, next_ff_c);
  };

  // ================================================================ leading linear layer: O (= t1 + bias) += W_pre a
  if constexpr (PRE) {
    publish();
    linear_layer(std::true_type{});
  }

  // ================================================================ LayerNorm of the accumulator -> fragments; O += FF2 bias
  {
    if constexpr (PRE) settle();
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < FNB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += O[nb][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)FC;
    float q = 0.f;
#pragma unroll
    for (int nb = 0; nb < FNB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = O[nb][r] - mean; q += d * d; }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q / (float)FC + p.eps);
    frags_from_acc(std::true_type{}, mean, rstd);
#pragma unroll
    for (int nb = 0; nb < FNB; ++nb)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.bias2 + nb * 32 + qq * 8 + hi * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) O[nb][4 * qq + j] += b2[j];
      }
    publish();
  }

  // ================================================================ feed-forward
  // FF1 accumulators of even / odd pairs, each as TWO chains (even / odd k-steps) so that no MFMA follows its own
  // predecessor directly; the GEGLU adds the two halves.
  f32x16 Sa, Sb, Sa2, Sb2;
  u32x4 Ha, Hb;                              // FF2 activation fragments (GEGLU of even / odd pairs)
  f32x16 binit;                              // accumulator start of the FF1 pair about to start
#pragma unroll
  for (int r = 0; r < 16; ++r) { Sa[r] = 0.f; Sb[r] = 0.f; Sa2[r] = 0.f; Sb2[r] = 0.f; }
  Ha = (u32x4){0u, 0u, 0u, 0u};
  Hb = (u32x4){0u, 0u, 0u, 0u};
  binit = rdb();
  bias_rd += 128;
  int pair = 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // iteration t: FF1 of pair t -> Scur; GEGLU of pair t-1 (Sprev) -> Hgelu; FF2 of pair t-2 with Hff2.  F1 / GELU false
  // in the two draining iterations.
  auto ff_iteration = [&](auto f1_c, auto gelu_c, auto next_ff_c, f32x16& Scur, f32x16& Scur2, const f32x16& Sprev, const f32x16& Sprev2,
                          u32x4& Hgelu, const u32x4& Hff2) __attribute__((always_inline)) {
    constexpr bool F1 = decltype(f1_c)::value, GELU = decltype(gelu_c)::value, NEXT_FF = decltype(next_ff_c)::value;
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
      if constexpr (b % 3 == 0) __builtin_amdgcn_s_waitcnt(0xC07F | ((DRAIN ? 0 : LEAD - 3) << 8));
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
      if constexpr (b + LEAD < NB) fr[b + LEAD] = rd(base, true, b + LEAD);
      else pre[b + LEAD - NB] = rd(nbase, NEXT_FF, b + LEAD - NB);
      if constexpr (F1 && b >= 10 && b < 14) {                            // accumulator start of the next pair, 4 x 16 bytes
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
    if constexpr (F1) {
      binit = bnext;
      ++pair;
      if (pair + 1 < FPAIRS) bias_rd += 128;       // (stay inside the table: the last pair's fetch is not used)
    }
    advance(nbank, nbase);
  };
  using T = std::true_type;
  using F = std::false_type;
  for (int tt = 0; tt < FPAIRS / 2; ++tt) {
    ff_iteration(T{}, T{}, T{}, Sa, Sa2, Sb, Sb2, Hb, Ha);     // even pair t: FF1 -> Sa; GEGLU of pair t-1 (Sb) -> Hb; FF2 of pair t-2 (Ha)
    ff_iteration(T{}, T{}, T{}, Sb, Sb2, Sa, Sa2, Ha, Hb);     // odd pair
  }
  ff_iteration(F{}, T{}, T{}, Sa, Sa2, Sb, Sb2, Hb, Ha);       // GEGLU of the last pair, FF2 of the one before
  ff_iteration(F{}, F{}, F{}, Sb, Sb2, Sa, Sa2, Ha, Hb);       // FF2 of the last pair (the next iteration, if any, is a linear one)

  // ================================================================ trailing linear layer: O = x + bias + W_post bf16(O)
  if constexpr (POST) {
    settle();
    frags_from_acc(std::false_type{}, 0.f, 1.f);
    load_rows(p.r2, p.ldr2, p.bias_post);
    publish();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the residual loads are waited for by their uses; the DMA count restarts clean)
    // (the read-ahead is taken afresh: carried through the transition above it would sit in scratch)
#pragma unroll
    for (int j = 0; j < LEAD; ++j) pre[j] = rd(frag_rd, false, j);
    linear_layer(std::false_type{});
  }

  // ---- epilogue: bf16(O), staged per wave in LDS (the ring is free once every wave is here), then whole rows as
  // 16-byte stores
  settle();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  constexpr int PITCH = FC * 2 + 16;                     // bytes per staged row
  constexpr int STAGE = 24576;
  static_assert(ROWS_PER_WAVE * PITCH <= STAGE && 4 * STAGE <= BIAS_OFF, "epilogue staging must fit the ring");
  char* stage = smem + wave * STAGE;
#pragma unroll
  for (int nb = 0; nb < FNB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x2 o;
      o[0] = pack_bf16x2(O[nb][4 * q], O[nb][4 * q + 1]);
      o[1] = pack_bf16x2(O[nb][4 * q + 2], O[nb][4 * q + 3]);
      *reinterpret_cast<u32x2*>(stage + lm * PITCH + (nb * 32 + q * 8 + hi * 4) * 2) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (a wave reads back only what it wrote itself)
  constexpr int CPR = FC / 8;                            // 16-byte pieces per row
  constexpr int ITER = ROWS_PER_WAVE * CPR / 64;
  static_assert(ROWS_PER_WAVE * CPR % 64 == 0, "rows must divide evenly over the lanes");
  u32x4 ov[ITER];
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = lane + i * 64;
    const int rl = idx / CPR, c = idx - rl * CPR;
    ov[i] = *reinterpret_cast<const u32x4*>(stage + rl * PITCH + c * 16);
  }
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = lane + i * 64;
    const int rl = idx / CPR, c = idx - rl * CPR;
    const int row = m_wave + rl;
    if (row < p.M) *reinterpret_cast<u32x4*>(p.out + (long)row * p.ldo + c * 8) = ov[i];
  }
}

}  // namespace

int ffn_fused_channels() { return FC; }
size_t ffn_stream_bytes(int pre, int post) { return (size_t)stream_iters(pre != 0, post != 0) * ITER_BYTES; }
size_t ffn_bias_bytes() { return BIAS_BYTES; }

int ffn_pack_launch(const float* w, int which, int pre, int post, bf16_t* stream, hipStream_t st) {
  ARG_CHECK(w && stream && which >= 0 && which <= 3, "ffn_pack: args");
  ARG_CHECK((which != 0 || pre) && (which != 3 || post), "ffn_pack: this stream has no such layer");
  const long total = (long)stream_iters(pre != 0, post != 0) * (ITER_BYTES / 16);
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, which, pre, post, stream);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int ffn_pack_bias_launch(const float* b1, float* out, hipStream_t st) {
  hipLaunchKernelGGL(ffn_pack_bias_kernel, dim3(cdiv(BIAS_BYTES / 4, 256)), dim3(256), 0, st, b1, out);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

template <bool PRE, bool POST>
static int launch_chain(const ChainKernelParams& k, hipStream_t st) {
  if (hedit_test_drained()) {
    if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&ffn_chain_kernel<PRE, POST, true>), LDS_TOTAL)) return rc;
    hipLaunchKernelGGL((ffn_chain_kernel<PRE, POST, true>), dim3(cdiv(k.M, BLOCK_ROWS)), dim3(256), LDS_TOTAL, st, k);
    LAUNCH_CHECK();
    return HEDIT_OK;
  }
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&ffn_chain_kernel<PRE, POST>), LDS_TOTAL)) return rc;
  hipLaunchKernelGGL((ffn_chain_kernel<PRE, POST>), dim3(cdiv(k.M, BLOCK_ROWS)), dim3(256), LDS_TOTAL, st, k);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int ffn_fused_launch(const FfnParams& f, hipStream_t st) {
  ARG_CHECK(f.C == FC, "ffn: the fused feed-forward exists for C = 320");
  ARG_CHECK(f.M > 0 && f.ldx % 8 == 0 && f.ldo % 8 == 0, "ffn: rows must be 16-byte aligned");
  ARG_CHECK(f.x && f.out && f.stream && f.bias1p && f.bias2 && f.gamma && f.beta, "ffn: null");
  const bool pre = f.a != nullptr, post = f.r2 != nullptr;
  ARG_CHECK(pre == post, "ffn: the chain exists with both outer linear layers or with neither");
  ChainKernelParams k{};
  k.r1 = f.x; k.ldr1 = f.ldx; k.gamma = f.gamma; k.beta = f.beta; k.eps = f.eps;
  k.stream = f.stream; k.bias1p = f.bias1p; k.bias2 = f.bias2; k.out = f.out; k.ldo = f.ldo; k.M = f.M;
  if (pre) {
    ARG_CHECK(f.lda % 8 == 0 && f.ldr2 % 4 == 0 && f.ldx % 4 == 0 && f.bias_pre && f.bias_post, "ffn chain: alignment / biases");
    k.a = f.a; k.lda = f.lda; k.r2 = f.r2; k.ldr2 = f.ldr2; k.bias_pre = f.bias_pre; k.bias_post = f.bias_post;
    return launch_chain<true, true>(k, st);
  }
  return launch_chain<false, false>(k, st);
}
