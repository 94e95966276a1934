// The GEGLU feed-forward of a BasicTransformerBlock as ONE kernel per row tile:
//     out = x + FF2( GEGLU( FF1( LayerNorm(x) ) ) )            (diffusers FeedForward with GEGLU; oracle/sd_unet.py)
// for the C = 320 level of the SD UNet.  Unfused this is four launches (LayerNorm, FF1+GEGLU, FF2+residual) that move
// the [M][1280] hidden activation and the normalised rows through HBM (10 passes over an [M][320] tensor where 2
// are needed) and run at HBM speed; here nothing but x and out touches HBM.
//
// gfx950 mapping -- "rows stay in registers, weights stream":
//   * a block = 4 waves (one per SIMD, up to 512 registers each), a wave owns 32 rows for the whole kernel:
//       A   LayerNorm(x) as the MFMA activation operand      2 x 10 fragments   80 VGPR
//       O   the FF2 accumulators (32 x 320 fp32)              20 x 2 tiles     160 VGPR
//       S   FF1 accumulators of one (value, gate) column pair, double-buffered  32 VGPR
//       H   the GEGLU output of 32 hidden units as the FF2 activation operand    8 VGPR
//     The FF1 rows are interleaved (value16 | gate16) at load time, so value and gate of a hidden unit sit in the same
//     lane and register; the MFMA result layout of two such pairs IS the activation-fragment layout of a 32-deep
//     FF2 k-step once the hidden units are renumbered inside their group of 32 -- a permutation folded into the FF2
//     weight packing.  The hidden activation never leaves the register file.
//   * the weights (2.4 MB bf16) are packed once, at load time, into a STREAM of 120 slots of 20 KB in consumption order
//     (two FF1 pairs, then the FF2 group they complete, one group behind), each slot stored as the LDS image the
//     fragment reads want (XOR-swizzled 16-byte pieces).  The kernel copies slots global -> LDS by lane-linear DMA
//     (buffer_load ... lds) into a ring of seven, six slots (120 KB) ahead, with a counted vmcnt and one barrier per
//     slot; every wave reads every slot (ds_read_b128, each weight fragment feeds two MFMAs).  All tiles read the same
//     stream, which fits the 4 MB L2 of an XCD.
//   * per slot and wave: 40 MFMA 16x16x32 + 20 fragment reads + 5 DMA issues, and the GEGLU arithmetic of the
//     previous pair (packed-fp32 erf, common.h) in the MFMA shadow.
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
constexpr int FKS = FC / 32;            // 32-deep k-steps of FF1
constexpr int FNT = FC / 16;            // 16-wide output tiles of FF2
constexpr int FH = 4 * FC;              // hidden units
constexpr int FPAIRS = FH / 16;         // (value16 | gate16) row pairs of FF1
constexpr int FGROUPS = FH / 32;        // 32-deep k-steps of FF2
constexpr int SLOT = 64 * FC;           // bytes: 32 rows x C (FF1 pair) = C rows x 32 (FF2 group)
constexpr int PIECES = SLOT / 1024;     // DMA instructions per slot
constexpr int RING = 6, AHEAD = 5;        // ring positions are compile-time constants: two group iterations = one lap
constexpr int NSLOTS = FPAIRS + FGROUPS + 1;      // + the all-zero "group -1" slot of the first iteration
constexpr int STREAM_SLOTS = NSLOTS + AHEAD + 1;  // the DMA runs this far past the end (zero slots)
constexpr int BIAS_OFF = RING * SLOT;
constexpr int BIAS_BYTES = 12288;       // 2560 packed FF1 biases, zero-padded to twelve 1 KB DMA pieces
constexpr int LDS_TOTAL = BIAS_OFF + BIAS_BYTES;
constexpr int ROWS_PER_WAVE = 32, BLOCK_ROWS = 128;
static_assert(PIECES % 4 == 0, "a slot must split evenly over the four waves");
constexpr int PPW = PIECES / 4;         // DMA pieces per wave and slot

// hidden unit held by k-slot kk (0..31) of group g:  lane group fq = kk / 8 holds elements e = kk % 8;
// e < 4 come from the group's first FF1 pair, e >= 4 from its second
__host__ __device__ inline int ffn_hidden_of_slot(int g, int kk) {
  const int c = kk >> 3, e = kk & 7;
  return 32 * g + (e >= 4 ? 16 : 0) + c * 4 + (e & 3);
}

__device__ __forceinline__ void unpack8v(const u32x4& v, float* f) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f[2 * q] = __builtin_bit_cast(float, v[q] << 16);
    f[2 * q + 1] = __builtin_bit_cast(float, v[q] & 0xffff0000u);
  }
}

// one thread per 16-byte piece of the stream
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, bf16_t* __restrict__ stream) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)STREAM_SLOTS * (SLOT / 16);
  if (idx >= total) return;
  const int slot = (int)(idx / (SLOT / 16));
  const int off = (int)(idx - (long)slot * (SLOT / 16)) * 16;      // byte offset inside the slot
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (slot < NSLOTS) {
    // which pair / group is it?
    int t = -1, g = -1;
    if (slot == NSLOTS - 1) g = FGROUPS - 1;
    else {
      const int q = slot / 3, r = slot % 3;      // iteration q: pairs 2q, 2q+1, group q-1 (q = 0: stays zero)
      if (r == 2) g = q - 1; else t = 2 * q + r;
    }
    const bool zero_slot = (t < 0 && g < 0);
    if (w1 && t >= 0) {
      // FF1 pair image: seg (64 k) x [32 rows][128 B], 16-byte piece p of row r holds k-chunk p ^ (r & 7)
      const int seg = off / 4096, row = (off % 4096) / 128, p = (off % 128) / 16;
      const int c = p ^ (row & 7);
      const int u = row;                                       // packed row 32 t + u: u < 16 value, else gate
      const int src = u < 16 ? t * 16 + u : FH + t * 16 + (u - 16);
      const float* s = w1 + (long)src * FC + seg * 64 + c * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s[e];
    } else if (w2 && g >= 0) {
      // FF2 group image: [C rows][64 B], piece p of row n holds k-chunk p ^ ((-(n >> 2)) & 3)
      const int n = off / 64, p = (off % 64) / 16;
      const int c = p ^ ((-(n >> 2)) & 3);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = w2[(long)n * FH + ffn_hidden_of_slot(g, c * 8 + e)];
    } else if (!zero_slot) {
      return;      // the other weight's call fills this slot
    }
  }
  uint4 o = pack8(v);
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(stream) + idx * 16) = o;
}

// FF1 bias in packed row order (value16 | gate16), zero-padded
__global__ __launch_bounds__(256) void ffn_pack_bias_kernel(const float* __restrict__ b1, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= BIAS_BYTES / 4) return;
  float v = 0.f;
  if (idx < 2 * FH) {
    const int t = idx >> 5, u = idx & 31;
    v = b1[u < 16 ? t * 16 + u : FH + t * 16 + (u - 16)];
  }
  out[idx] = v;
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
// VGPRs (the GEGLU arithmetic reads them; the compiler's MFMA form would park them in AGPRs behind 32 v_accvgpr_read
// per pair), the FF2 accumulators and the LayerNorm fragments in AGPRs (240 registers nothing but MFMAs touch until
// the epilogue).  What the compiler does not do for an asm MFMA is hazard padding, so every hazard is excluded by
// construction: an FF1 accumulator is first read 20+ MFMAs after its last write, an FF2 accumulator only in the
// epilogue (behind s_nops and a re-definition, see there), each accumulator chain is revisited every 4th (FF1) /
// 120th (FF2) MFMA, and no VALU result feeds an MFMA closer than a few bundles.
__device__ __forceinline__ void mfma_v(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(a));
}
// first MFMA of an FF1 chain: the accumulator start (bias) is a separate, read-only operand -- a v_mov into the
// accumulator right in front of an asm MFMA would be a VALU-write -> MFMA-read hazard nobody pads.  The other way
// round (the MFMA still reading C while something overwrites it) is excluded by keeping C's registers live for
// another eight MFMAs (the empty asm statements at bundle 12)
__device__ __forceinline__ void mfma_v0(f32x4& acc, const bf16x8& w, const bf16x8& a, const f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(w), "a"(a), "v"(c));
}
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

// GEGLU of one FF1 pair = 8 elements per lane (2 row tiles x 4 hidden units), as a list of 108 single VALU
// operations in stage-major order (operation k: stage k / 8 of element k % 8, then four bf16 packs), so that the
// kernel can hand them out two per MFMA.  With v = value, x = gate:
//   v gelu(x),   gelu(x) = x Phi(x) = max(x, 0) - |x| r(|x|),   r(z) = erfc(z / sqrt 2) / 2 = q(z)^-16,
// q a degree-5 polynomial (the Abramowitz-Stegun 7.1.28 form, refitted with the 1/2 and the 1/sqrt 2 folded in:
// |gelu error| < 5e-6 absolute, three decimal orders below the bf16 resolution of the result; tools/gelu_fit.py).
// Plain fp32 operations: packed fp32 VALU is slow beside MFMAs.  No cancellation anywhere: for large |x| the
// second term vanishes (q^16 overflows to +inf, 1/inf = 0).
struct Gelu8 {
  float q[8], g[8];
  uint32_t h[4];       // bf16 pairs: h[2 i + half] = elements (i, 2 half), (i, 2 half + 1)
};
constexpr int GELU_OPS = 13 * 8 + 4;
// (the empty volatile asm pins each result where it is written: instruction selection otherwise sinks an operation
//  down to its consumer, out of the bundle it was meant to fill; the operation itself stays compiler-visible, so its
//  hazards and waits are the compiler's business)
#define GELU_PIN(v) asm volatile("" : "+v"(v))
template <int K>
__device__ __forceinline__ void gelu_op(Gelu8& g, const f32x4 (&S)[2][2]) {
  if constexpr (K < 104) {
    constexpr int st = K / 8, e = K % 8, i = e / 4, r = e % 4;
    const float x = S[1][i][r];
    if constexpr (st == 0) g.q[e] = __builtin_fmaf(9.73478169e-05f, __builtin_fabsf(x), -1.02176718e-04f);
    else if constexpr (st == 1) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 3.62392071e-03f);
    else if constexpr (st == 2) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 2.19443815e-02f);
    else if constexpr (st == 3) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 5.21099924e-02f);
    else if constexpr (st == 4) g.q[e] = __builtin_fmaf(g.q[e], __builtin_fabsf(x), 1.04427174e+00f);
    else if constexpr (st <= 8) g.q[e] = g.q[e] * g.q[e];
    else if constexpr (st == 9) g.q[e] = __builtin_amdgcn_rcpf(g.q[e]);
    if constexpr (st <= 9) GELU_PIN(g.q[e]);
    else {
      if constexpr (st == 10) g.g[e] = __builtin_fmaxf(x, 0.f);
      else if constexpr (st == 11) g.g[e] = __builtin_fmaf(-__builtin_fabsf(x), g.q[e], g.g[e]);
      else g.g[e] = g.g[e] * S[0][i][r];
      GELU_PIN(g.g[e]);
    }
  } else {
    constexpr int p = K - 104;
    g.h[p] = pack_bf16x2(g.g[2 * p], g.g[2 * p + 1]);
    GELU_PIN(g.h[p]);
  }
}
// operations [lo, hi) of the list
template <int LO, int HI>
__device__ __forceinline__ void gelu_ops(Gelu8& g, const f32x4 (&S)[2][2]) {
  static_for<HI - LO>([&](auto k) { gelu_op<LO + decltype(k)::value>(g, S); });
}

__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnKernelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m_wave = blockIdx.x * BLOCK_ROWS + wave * ROWS_PER_WAVE;

#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.stream), (short)0, (int)(STREAM_SLOTS * SLOT), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias1p), (short)0, (int)BIAS_BYTES, 0x00020000);
#endif
  const unsigned dma_voff = (unsigned)(wave * PPW * 1024 + lane * 16);
  // piece i of this wave's share of stream slot s -> ring position pos
  auto dma_piece = [&](int s, int pos, int i) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + pos * SLOT + (wave * PPW + i) * 1024), 16, dma_voff,
                                             s * SLOT + i * 1024, 0, 0);
#else
    (void)s; (void)pos; (void)i;
#endif
  };

  // ---- prologue: start the weight stream, then LayerNorm this wave's rows into fragment registers
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 3; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(smem + BIAS_OFF + (wave * 3 + i) * 1024), 16,
                                             (unsigned)((wave * 3 + i) * 1024 + lane * 16), 0, 0, 0);
#endif
#pragma unroll
  for (int s = 0; s < AHEAD; ++s)
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_piece(s, s, i);

  bf16x8 A[2][FKS];
  {
    u32x4 raw[2][FKS];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = m_wave + 16 * i + fr;
      if (row > p.M - 1) row = p.M - 1;
      const bf16_t* src = p.x + (long)row * p.ldx + fq * 8;
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks) raw[i][ks] = *reinterpret_cast<const u32x4*>(src + ks * 32);
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks) {
        float f[8];
        unpack8v(raw[i][ks], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      mean[i] = s / (float)FC;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks) {
        float f[8];
        unpack8v(raw[i][ks], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean[i]; q += d * d; }
      }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      rstd[i] = rsqrtf(q / (float)FC + p.eps);
    }
#pragma unroll
    for (int ks = 0; ks < FKS; ++ks) {
      const float4* g4 = reinterpret_cast<const float4*>(p.gamma + ks * 32 + fq * 8);
      const float4* b4 = reinterpret_cast<const float4*>(p.beta + ks * 32 + fq * 8);
      const float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float f[8], o[8];
        unpack8v(raw[i][ks], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[e] - mean[i]) * rstd[i] * gg[e] + bb[e];
        const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        A[i][ks] = __builtin_bit_cast(bf16x8, pk);
        asm volatile("" : "+a"(A[i][ks]));       // home in the AGPR file from here on (every use is an MFMA operand)
      }
    }
  }

  f32x4 O[FNT][2];
#pragma unroll
  for (int nt = 0; nt < FNT; ++nt)
#pragma unroll
    for (int i = 0; i < 2; ++i) O[nt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a slot
  const int w1_lo = fr * 128 + ((fq ^ (fr & 7)) << 4);                    // FF1 pair image, even k-step (odd: ^ 64)
  const int w2_lo = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);           // FF2 group image
  int bias_rd = BIAS_OFF + fq * 16;                                       // packed bias of the next FF1 pair to start
  // fragment n (0 .. 19) of an FF1 pair slot: k-step n / 2, (value | gate) tile n % 2; of an FF2 group slot: output tile n.
  // ds_read offsets are 16-bit immediates, so each lane offset exists twice: for ring positions 0-2 and 3-5 (the
  // empty asm keeps the compiler from re-deriving them with a v_add per read)
  constexpr int HALF = (RING / 2) * SLOT;
  static_assert(HALF < 65536, "half a ring must be addressable by a ds_read offset");
  int w1e[2], w1o[2], w2b[2];       // (LDS byte offsets)
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    w1e[hf] = hf * HALF + w1_lo;
    w1o[hf] = hf * HALF + (w1_lo ^ 64);
    w2b[hf] = hf * HALF + w2_lo;
    asm volatile("" : "+v"(w1e[hf]), "+v"(w1o[hf]), "+v"(w2b[hf]));
  }
  auto rd1 = [&](int pos, int n) __attribute__((always_inline)) {
    const int ks = n >> 1, j = n & 1, hf = pos / (RING / 2);
    return *reinterpret_cast<const bf16x8*>(smem + (((ks & 1) ? w1o[hf] : w1e[hf]) + (pos - hf * (RING / 2)) * SLOT + (ks >> 1) * 4096 + j * 2048));
  };
  auto rd2 = [&](int pos, int n) __attribute__((always_inline)) {
    const int hf = pos / (RING / 2);
    return *reinterpret_cast<const bf16x8*>(smem + (w2b[hf] + (pos - hf * (RING / 2)) * SLOT + n * 1024));
  };
  // accumulator start of an FF1 pair: the packed bias
  auto rdb = [&](int j) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(smem + bias_rd + j * 64); };

  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");           // slots 0 .. AHEAD-1 and the biases are in LDS

  // One slot = 40 "bundles" of [1 MFMA | 1-2 GEGLU operations | every other one a fragment read 16 bundles ahead (the
  // last eight reads of a slot fetch the first fragments of the next one) | five of them a DMA piece], pinned by
  // sched_barrier so that the VALU and LDS work sits in the MFMAs' shadow instead of in a block of its own.  At a
  // slot boundary: [my DMA pieces of the slot after next have landed: vmcnt(3 slots in flight)] [my reads of the
  // slot just finished have returned: lgkmcnt(the 8 newest = next slot's)] barrier; the finished slot's ring
  // position is then refilled during the next slot.
  constexpr int LEAD = 8;                    // fragment reads in flight ahead of their MFMAs
  constexpr int NB = 40;                     // bundles (MFMAs) per slot
  int s = 0;                                 // stream slot being consumed
  auto boundary = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"((AHEAD - 2) * PPW), "n"(LEAD) : "memory");
  };

  f32x4 Se0[2][2], Se1[2][2], So0[2][2], So1[2][2];      // FF1 accumulators of the even / odd group iteration
  u32x2 hlo[2];
  bf16x8 pre[LEAD];                          // first fragments of the slot about to start
  f32x4 binit[2];                            // accumulator start of the FF1 pair about to start
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) So1[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) hlo[i] = (u32x2){0u, 0u};
#pragma unroll
  for (int n = 0; n < LEAD; ++n) pre[n] = rd1(0, n);
  binit[0] = rdb(0);
  binit[1] = rdb(1);
  bias_rd += 128;

  // group iteration g on ring positions PA, PA+1, PA+2: slot A = FF1 pair 2g -> S0, slot B = FF1 pair 2g+1 -> S1,
  // slot C = FF2 group g-1 (group -1: zero weights); GEGLU of pair 2g-1 (Sp = S1 of the previous iteration, completes
  // group g-1) during A and the first half of B, GEGLU of pair 2g (S0 -> next hlo) during the second half of B and C.
  // LAST: the slot after C is the final FF2 slot (group 39) instead of another slot A.
  auto iteration = [&](auto pa_c, auto last_c, const f32x4 (&Sp)[2][2], f32x4 (&S0)[2][2], f32x4 (&S1)[2][2]) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int PA = decltype(pa_c)::value, PB = PA + 1, PC = PA + 2, PN = (PA + 3) % RING, PP = (PA + RING - 1) % RING;
    Gelu8 G1, G0;
    bf16x8 fa[20 + LEAD], fb[20 + LEAD], fc[20 + LEAD];
#pragma unroll
    for (int n = 0; n < LEAD; ++n) fa[n] = pre[n];
    f32x4 bB[2], bA[2];
    u32x4 H[2];      // activation fragments of FF2 group g-1, assembled (and pinned) long before slot C reads them
    // ---------------- slot A
    static_for<NB>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      constexpr int n = b / 2, ks = n / 2, j = n % 2, i = b % 2;
      if constexpr (ks == 0) mfma_v0(S0[j][i], fa[n], A[i][ks], binit[j]);
      else mfma_v(S0[j][i], fa[n], A[i][ks]);
      if constexpr (b == 12) asm volatile("" ::"v"(binit[0]), "v"(binit[1]));     // (see mfma_v0: keeps the C operand's registers intact)
      if constexpr (b % 2 == 0) {
        if constexpr (n + LEAD < 20) fa[n + LEAD] = rd1(PA, n + LEAD);
        else fb[n + LEAD - 20] = rd1(PB, n + LEAD - 20);
      }
      if constexpr (b == 19) bB[0] = rdb(0);
      if constexpr (b == 21) { bB[1] = rdb(1); bias_rd += 128; }
      if constexpr (b % 8 == 4) dma_piece(s + AHEAD, PP, b / 8);
      gelu_ops<(b * GELU_OPS) / 60, ((b + 1) * GELU_OPS) / 60>(G1, Sp);
      __builtin_amdgcn_sched_barrier(0);
    });
    boundary();
    // ---------------- slot B
    static_for<NB>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      constexpr int n = b / 2, ks = n / 2, j = n % 2, i = b % 2;
      if constexpr (ks == 0) mfma_v0(S1[j][i], fb[n], A[i][ks], bB[j]);
      else mfma_v(S1[j][i], fb[n], A[i][ks]);
      if constexpr (b == 12) asm volatile("" ::"v"(bB[0]), "v"(bB[1]));
      if constexpr (b % 2 == 0) {
        if constexpr (n + LEAD < 20) fb[n + LEAD] = rd1(PB, n + LEAD);
        else fc[n + LEAD - 20] = rd2(PC, n + LEAD - 20);
      }
      if constexpr (b % 8 == 4) dma_piece(s + 1 + AHEAD, PA, b / 8);
      if constexpr (b < 20) gelu_ops<((b + 40) * GELU_OPS) / 60, ((b + 41) * GELU_OPS) / 60>(G1, Sp);
      else gelu_ops<((b - 20) * GELU_OPS) / 60, ((b - 19) * GELU_OPS) / 60>(G0, S0);
      if constexpr (b == 19) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          H[i2] = (u32x4){hlo[i2][0], hlo[i2][1], G1.h[2 * i2], G1.h[2 * i2 + 1]};
          asm volatile("" : "+v"(H[i2]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    boundary();
    // ---------------- slot C
    static_for<NB>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      constexpr int n = b / 2, i = b % 2;
      mfma_a(O[n][i], fc[n], __builtin_bit_cast(bf16x8, H[i]));
      if constexpr (b % 2 == 0) {
        if constexpr (n + LEAD < 20) fc[n + LEAD] = rd2(PC, n + LEAD);
        else if constexpr (LAST) pre[n + LEAD - 20] = rd2(PN, n + LEAD - 20);
        else pre[n + LEAD - 20] = rd1(PN, n + LEAD - 20);
      }
      if constexpr (!LAST) {
        if constexpr (b == 19) bA[0] = rdb(0);
        if constexpr (b == 21) { bA[1] = rdb(1); bias_rd += 128; }
      }
      if constexpr (b % 8 == 4) dma_piece(s + 2 + AHEAD, PB, b / 8);
      gelu_ops<((b + 20) * GELU_OPS) / 60, ((b + 21) * GELU_OPS) / 60>(G0, S0);
      __builtin_amdgcn_sched_barrier(0);
    });
    boundary();
#pragma unroll
    for (int i = 0; i < 2; ++i) hlo[i] = (u32x2){G0.h[2 * i], G0.h[2 * i + 1]};
    if constexpr (!LAST) { binit[0] = bA[0]; binit[1] = bA[1]; }
    s += 3;
  };
  using P0 = std::integral_constant<int, 0>;
  using P3 = std::integral_constant<int, 3>;
  for (int gg = 0; gg < FGROUPS / 2 - 1; ++gg) {
    iteration(P0{}, std::false_type{}, So1, Se0, Se1);
    iteration(P3{}, std::false_type{}, Se1, So0, So1);
  }
  iteration(P0{}, std::false_type{}, So1, Se0, Se1);
  iteration(P3{}, std::true_type{}, Se1, So0, So1);
  // ---- tail: GEGLU of the last pair, then the last FF2 slot (ring position 0 again)
  {
    Gelu8 G1;
    gelu_ops<0, GELU_OPS>(G1, So1);
    u32x4 H[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      H[i] = (u32x4){hlo[i][0], hlo[i][1], G1.h[2 * i], G1.h[2 * i + 1]};
      asm volatile("" : "+v"(H[i]));
    }
    asm volatile("s_nop 1");       // VALU write -> (asm) MFMA read
    bf16x8 fc[20];
#pragma unroll
    for (int n = 0; n < LEAD; ++n) fc[n] = pre[n];
    static_for<NB>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      constexpr int n = b / 2, i = b % 2;
      mfma_a(O[n][i], fc[n], __builtin_bit_cast(bf16x8, H[i]));
      if constexpr (b % 2 == 0 && n + LEAD < 20) fc[n + LEAD] = rd2(0, n + LEAD);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- epilogue: bf16(O + bias2), staged per wave in LDS (the ring is free once every wave is here), then whole rows:
  // + x (residual) in fp32, rounded again, 16-byte stores.  (The s_nops: the last asm MFMAs must have retired before
  // the first v_accvgpr_read -- an MFMA-write -> VALU-read hazard the compiler cannot see; the empty asm statements
  // re-define every accumulator behind them, so no read can be scheduled up into the MFMA stream.)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int nt = 0; nt < FNT; ++nt)
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+a"(O[nt][i]));
  constexpr int PITCH = FC * 2 + 16;                     // bytes per staged row
  constexpr int STAGE = 24576;
  static_assert(ROWS_PER_WAVE * PITCH <= STAGE && 4 * STAGE <= BIAS_OFF, "epilogue staging must fit the ring");
  char* stage = smem + wave * STAGE;
#pragma unroll
  for (int nt = 0; nt < FNT; ++nt) {
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.bias2 + nt * 16 + fq * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 v = O[nt][i] + b2;
      u32x2 o;
      o[0] = pack_bf16x2(v[0], v[1]);
      o[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(stage + (16 * i + fr) * PITCH + (nt * 16 + fq * 4) * 2) = o;
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
  for (int it = 0; it < ITER; ++it) {
    const int idx = lane + it * 64;
    const int rl = idx / CPR, c = idx - rl * CPR;
    int row = m_wave + rl;
    if (row > p.M - 1) row = p.M - 1;
    res[it] = *reinterpret_cast<const u32x4*>(p.x + (long)row * p.ldx + c * 8);
    ov[it] = *reinterpret_cast<const u32x4*>(stage + rl * PITCH + c * 16);
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = lane + it * 64;
    const int rl = idx / CPR, c = idx - rl * CPR;
    const int row = m_wave + rl;
    u32x4 o;
    o[0] = add2(ov[it][0], res[it][0]);
    o[1] = add2(ov[it][1], res[it][1]);
    o[2] = add2(ov[it][2], res[it][2]);
    o[3] = add2(ov[it][3], res[it][3]);
    if (row < p.M) *reinterpret_cast<u32x4*>(p.out + (long)row * p.ldo + c * 8) = o;
  }
}

}  // namespace

int ffn_fused_channels() { return FC; }
size_t ffn_stream_bytes() { return (size_t)STREAM_SLOTS * SLOT; }
size_t ffn_bias_bytes() { return BIAS_BYTES; }

int ffn_pack_launch(const float* w1, const float* w2, bf16_t* stream, hipStream_t st) {
  const long total = (long)STREAM_SLOTS * (SLOT / 16);
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
