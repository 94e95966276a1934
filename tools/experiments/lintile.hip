// EXPERIMENT (round 4), not part of the product library: built into the side library of tools/experiments/build_xffn.sh and
// timed against csrc/linchain.hip by tools/experiments/lin_bench.py.  Outcome: correct (same bits as
// linchain.hip for the residual stream, 5e-5 apart for q / k / v^T; rows independent bit for bit) but no faster: 408 us against
// 410 for the two-layer chain and 899 against 611 for the four-layer one at M = 491 520 -- with 80 KB of LDS per block there is
// no room to stage the next tile, two blocks per CU do not hide five dependent memory round trips per tile, and smaller tiles
// pay for it in weight traffic (every block streams every weight).
// The same at the 640-channel level (the default build, LINTILE_C = 640; -DLINTILE_C=320 for the above): 64-row tile = all 160 KB
// of LDS with its staging buffer, ONE block per CU, rolled k loop.  Correct on the first run (1.7e-3 / 2.3e-3 against fp32), and
// again no win: 346 us against the 384 us of the three launches the two-layer chain replaces, 789 us against ~715 for the
// four-layer one (tools/experiments/lin_bench640.py, gpurun_out/r04/lin640.txt).  The k loops are ~36 us of a 105 us tile; the rest
// is tile I/O, LayerNorm and four staged outputs with nothing to overlap them -- what linchain.hip's hand-pipelined schedule
// buys at C = 320 and what a C = 640 version would have to rebuild inside an LDS that the tile alone fills.
//
// The token-local projections AROUND the two attentions of a BasicTransformerBlock at the C = 320 level of the SD UNet (what
// linchain.hip computes; oracle/sd_unet.py: Transformer2DModel.norm / proj_in, BasicTransformerBlock norm1 / attn1.to_q|k|v,
// attn1.to_out, norm2, attn2.to_q), in round 4's mapping -- "the row tile in LDS, the weights straight from L2 into registers":
//
//   two layers:   t1 = attn1.to_out(a) + b + t0     (written out: the residual stream);   q2 = attn2.to_q( LayerNorm(t1) )
//   four layers:  t0 = proj_in( GroupNorm(x) ) + b  (GroupNorm applied while the tile is staged, from per-(image, channel)
//                 scale / shift);   q | k = attn1.to_q | to_k ( LayerNorm(t0) );   v^T = attn1.to_v( LayerNorm(t0) )^T
//
// gfx950 mapping
//   * a block = 4 waves owns a tile of 64 rows; its activation tile X[64][320] (bf16, 40 KB, 16-byte chunks XOR-swizzled
//     inside 128-byte groups) lives in LDS and is rewritten once per layer; wave w owns the output COLUMNS [80 w, 80 w + 80)
//     of every layer.  Per 32-deep k-step a wave reads the tile's 4 activation fragments from LDS (4 KB for 20 MFMAs) and
//     takes its 5 weight fragments (1 KB each, packed fragment-major at load time: one coalesced 16-byte load per lane) from
//     global memory / L2 into VGPRs -- no LDS round trip, no LDS-DMA, no hand-counted wait anywhere in this file: every load
//     is one the compiler counts.  v_mfma_f32_16x16x32_bf16 with the weight fragment as the A operand (gemm.hip's choice:
//     a lane ends up with 4 consecutive columns of one row), asm so that the accumulators and activation fragments sit in
//     AGPRs.  tools/ubench/chain_core.hip: this inner loop alone reaches 1.5 PFLOP/s on the chip (linchain.hip's, every
//     weight fragment through LDS to every wave: bound by the LDS port at half of that).
//   * TWO blocks per CU (80 KB of LDS and 256 registers each): these chains are HBM-bound (8 / 10 bytes per row and channel
//     against 2 / 4 layers), so what matters is that the tile I/O of one block runs under the arithmetic of the other --
//     the hardware interleaves the two blocks, nothing is hand-pipelined.  (The weights are read twice per CU: 400 KB per
//     128 rows and layer pair, 100 GB/s per CU at the HBM-bound rate; chain_core measures that much arriving.)
//   * tile I/O: coalesced 16-byte global accesses by all 256 threads through registers into / out of the swizzled LDS image;
//     results go through the second buffer B (residual rows in, result rows out, v^T staged transposed: 128 contiguous bytes
//     per feature).
// Every output row depends on its own input row only and every summation order is fixed.
#include <type_traits>
#include <utility>

#include "../../h-edit_amd/csrc/common.h"
#include "../../h-edit_amd/csrc/kernels.h"
#include "lintile_decl.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#ifndef LINTILE_C
#define LINTILE_C 640
#endif
constexpr int TC = LINTILE_C;            // channels: 320 (the experiment of DESIGN.md 5.0 item 2) or 640
constexpr int TBPC = TC <= 320 ? 2 : 1;  // blocks per CU (LDS: two tiles per block)
constexpr int TBM = 64;                  // rows per tile
constexpr int TNW = 4;                   // waves = column slices
constexpr int TWN = TC / TNW;            // 80 output columns per wave
constexpr int TNI = TWN / 16;            // 5 weight fragments per wave and k-step
constexpr int TMI = TBM / 16;            // 4 activation fragments per k-step
constexpr int TKS = TC / 32;             // 10 k-steps per layer
constexpr int TSET = TNW * TNI * 1024;   // stream bytes per k-step (20 KB)
constexpr int TROW = TC * 2;             // bytes per activation row
constexpr int TTILE = TBM * TROW;        // 40960
constexpr int TX_OFF = 0, TB_OFF = TTILE;
constexpr int TLDS = 2 * TTILE;          // 80 KB at C = 320 (two blocks per CU), 160 KB at C = 640
constexpr int TCPR = TC / 8;             // 16-byte chunks per row
constexpr int TIO = TBM * TCPR / 256;    // chunks per thread and tile
constexpr int TPART = TC <= 320 ? 1 : 2, TIOP = TIO / TPART;
static_assert(TBPC * TLDS <= 160 * 1024, "LDS per CU");
constexpr int TDIVM = (1 << 18) / TCPR + 1;      // q / TCPR == (q * TDIVM) >> 18 for q < TBM * TCPR
static_assert(TBM * TCPR % 256 == 0 && TKS % 2 == 0, "tile chunks divide over the block; double-buffered weight sets");

// w [C][C] fp32 -> layer `layer` of the stream: [layer][k-step][wave][fragment j][lane] 16 bytes = W[80 wave + 16 j + (lane & 15)]
// [32 ks + 8 (lane >> 4) .. + 8], times `scale`
__global__ __launch_bounds__(256) void lin_tile_pack_kernel(const float* __restrict__ w, int layer, float scale, bf16_t* __restrict__ stream) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= TKS * (TSET / 16)) return;
  const int ks = idx / (TSET / 16), r = idx - ks * (TSET / 16);
  const int wave = r / (TNI * 64), j = (r / 64) % TNI, lane = r & 63;
  const int n = wave * TWN + j * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = scale * w[(long)n * TC + k0 + e];
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(stream) + ((size_t)layer * TKS * TSET + (size_t)idx * 16)) = pack8(v);
}

struct TileParams {
  const bf16_t* a; long lda;
  const bf16_t* r1; long ldr1;
  const float* bias_pre; const float* gamma; const float* beta; float eps;
  const bf16_t* stream;
  bf16_t* out_mid; long ldmid;
  bf16_t* out_p[2]; long ldp[2];
  bf16_t* out; long ldo;
  const float* gn_ss; int rows_per_image;
  int M;
};

template <class F, int... I>
__device__ __forceinline__ void tstatic_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void tstatic_for(F&& f) {
  tstatic_for_impl(f, std::make_integer_sequence<int, N>{});
}

// byte offset of 16-byte chunk c of row m in the swizzled row-major tile
__device__ __forceinline__ int tswz(int m, int c) { return m * TROW + (((c & ~7) | ((c ^ m) & 7)) << 4); }

// NPOST: layers behind the LayerNorm (1: attn2.to_q; 3: q, k, v^T).  GNIN: GroupNorm'd input, no residual, v^T transposed.
template <int NPOST, bool GNIN>
__global__ __launch_bounds__(256, TBPC) void lin_tile_kernel(TileParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int ntiles = (p.M + TBM - 1) / TBM;
  constexpr int NL = 1 + NPOST, NPOS = NL * TKS;

  int xrd[TMI];                                   // activation fragment i: rows 16 i + fr, k chunk 4 ks + fq
#pragma unroll
  for (int i = 0; i < TMI; ++i) {
    const int m = i * 16 + fr;
    xrd[i] = m * TROW + ((fq ^ (m & 7)) << 4);
  }
  const int ncol0 = wave * TWN + fq * 4;          // the lane's 4 consecutive columns of block j: ncol0 + 16 j, rows 16 i + fr
  auto pc_off = [&](int i, int j) __attribute__((always_inline)) {
    const int m = i * 16 + fr, n = ncol0 + j * 16;
    return tswz(m, n >> 3) + ((n & 4) << 1);
  };
  auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); return v; };

  // ---- weights: two sets of the wave's 5 fragments, the next k-step requested while this one's MFMAs run; the stream is
  // read cyclically over the tiles of this block
  // (buffer loads: descriptor + one lane offset + a SCALAR stream offset per load -- with flat addresses the compiler
  //  materialises the 64-bit address of every fragment of a tile's 20 / 40 k-steps as loop invariants and spills them)
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.stream), (short)0, NPOS * TSET, 0x00020000);
#endif
  const unsigned w_voff = (unsigned)(wave * TNI * 1024 + lane * 16);
  bf16x8 wr[2][TNI];
  int ppos = 0;                                   // byte offset of the stream k-step the next load fetches
  auto load_set = [&](auto s_) __attribute__((always_inline)) {
    constexpr int s = decltype(s_)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    const int so = opaque(ppos);
#pragma unroll
    for (int j = 0; j < TNI; ++j) wr[s][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff, so + j * 1024, 0));
#endif
    ppos = ppos + TSET == NPOS * TSET ? 0 : ppos + TSET;
  };
  load_set(std::integral_constant<int, 0>{});

  f32x4 acc[TNI][TMI];
  auto k_loop = [&]() __attribute__((always_inline)) {
    bf16x8 xa[TMI], xb[TMI];
    auto read_x = [&](bf16x8 (&xf)[TMI], int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TMI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(smem + TX_OFF + ((xrd[i] ^ ((ks & 1) << 6)) + (ks >> 1) * 128));
    };
    read_x(xa, 0);
    // one k-step: request the next weight set, read the next activation fragments, TNI x TMI MFMAs.  FIRST: the accumulators
    // start from the inline constant 0.  Every asm MFMA opens with s_nop 1: the compiler parks weight sets in AGPRs across
    // the epilogues and copies them back (v_accvgpr_read) right in front of their first use, and it pads no hazard for an
    // instruction inside asm -- without the wait states the first MFMA behind such a copy reads the PREVIOUS fragment.
    auto k_step = [&](auto par_, auto first_, int ks) __attribute__((always_inline)) {
      constexpr int par = decltype(par_)::value;
      constexpr bool first = decltype(first_)::value;
      load_set(std::integral_constant<int, par ^ 1>{});            // (behind the last k-step of a layer: the next layer's / tile's first)
      bf16x8 (&xc)[TMI] = par ? xb : xa;
      bf16x8 (&xn)[TMI] = par ? xa : xb;
      if (ks + 1 < TKS) read_x(xn, ks + 1);
#pragma unroll
      for (int j = 0; j < TNI; ++j)
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
          if constexpr (first) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc[j][i]) : "v"(wr[par][j]), "a"(xc[i]));
          else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wr[par][j]), "a"(xc[i]));
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    // (the first pair peeled, the rest a rolled loop of pairs: fully unrolled, the four layers of the C = 640 form are
    //  3 200 MFMAs of straight-line code, more than the instruction cache holds)
    k_step(std::integral_constant<int, 0>{}, std::true_type{}, 0);
    k_step(std::integral_constant<int, 1>{}, std::false_type{}, 1);
#pragma unroll 1
    for (int ks = 2; ks < TKS; ks += 2) {
      k_step(std::integral_constant<int, 0>{}, std::false_type{}, ks);
      k_step(std::integral_constant<int, 1>{}, std::false_type{}, ks + 1);
    }
    // the accumulators of the last asm MFMAs become readable by the VALU
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < TNI; ++j)
#pragma unroll
      for (int i = 0; i < TMI; ++i) asm volatile("" : "+a"(acc[j][i]));
  };

  // ---- coalesced tile I/O: chunk q = t * 256 + tid -> (row q / 40, chunk q % 40); bounds-checked buffer accesses
#if defined(__HIP_DEVICE_COMPILE__)
  auto rsrc = [&](const bf16_t* ptr, long ld, int cols) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ptr), (short)0, (int)((((long)p.M - 1) * ld + cols) * 2), 0x00020000);
  };
#endif
  auto io_rc = [&](int t, int& r, int& c) __attribute__((always_inline)) {
    const int q = opaque(t * 256) + tid;
    r = (q * TDIVM) >> 18;                        // q / TCPR
    c = q - r * TCPR;
  };
  // (the tile moves in TPART pieces of TIOP chunks per thread: a whole 64 x 640 tile is 80 registers per thread and spilled)
  auto load_tile = [&](const bf16_t* src, long ld, int row0, u32x4 (&buf)[TIOP], int t0) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, ld, TC);
    const int ld2 = opaque((int)ld * 2);
#pragma unroll
    for (int t = 0; t < TIOP; ++t) {
      int r, c;
      io_rc(t0 + t, r, c);
      buf[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((row0 + r) * ld2 + c * 16), 0, 0));
    }
#else
    (void)src; (void)ld; (void)row0; (void)buf; (void)t0;
#endif
  };
  auto put_tile = [&](int tile_off, const u32x4 (&buf)[TIOP], int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < TIOP; ++t) {
      int r, c;
      io_rc(t0 + t, r, c);
      *reinterpret_cast<u32x4*>(smem + tile_off + tswz(r, c)) = buf[t];
    }
  };
  auto store_tile = [&](int tile_off, bf16_t* dst, long ld, int row0) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = rsrc(dst, ld, TC);
    const int ld2 = opaque((int)ld * 2);
    constexpr int GRP = 5;
#pragma unroll
    for (int t0 = 0; t0 < TIO; t0 += GRP) {
      u32x4 buf[GRP];
      unsigned off[GRP];
#pragma unroll
      for (int t = 0; t < GRP; ++t) {
        int r, c;
        io_rc(t0 + t, r, c);
        buf[t] = *reinterpret_cast<const u32x4*>(smem + tile_off + tswz(r, c));
        off[t] = (unsigned)((row0 + r) * ld2 + c * 16);
      }
#pragma unroll
      for (int t = 0; t < GRP; ++t) __builtin_amdgcn_raw_buffer_store_b128(buf[t], rs, off[t], 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    (void)tile_off; (void)dst; (void)ld; (void)row0;
#endif
  };
  // the accumulator as bf16 into a tile (the lane's 8-byte pieces)
  auto acc_to_tile = [&](int tile_off) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TNI; ++j)
#pragma unroll
      for (int i = 0; i < TMI; ++i) {
        u32x2 o;
        o[0] = pack_bf16x2(acc[j][i][0], acc[j][i][1]);
        o[1] = pack_bf16x2(acc[j][i][2], acc[j][i][3]);
        *reinterpret_cast<u32x2*>(smem + tile_off + pc_off(i, j)) = o;
      }
  };
  float2* const scratch = reinterpret_cast<float2*>(smem + TX_OFF);        // [wave][row] statistics, in X while X is dead

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TBM;
    // ================================================================ the tile's input rows -> X (GroupNorm on the way), residual rows -> B
#pragma unroll
    for (int part = 0; part < TPART; ++part) {
      const int t0 = part * TIOP;
      u32x4 ba[TIOP];
      load_tile(p.a, p.lda, row0, ba, t0);
      if constexpr (GNIN) {
        // x * scale + shift with the (scale, shift) pairs of the row's image, [image][C][2] fp32 (L2-resident)
#pragma unroll
        for (int t = 0; t < TIOP; ++t) {
          int r, c;
          io_rc(t0 + t, r, c);
          int row = row0 + r;
          row = row < p.M ? row : p.M - 1;
          const f32x4* ss = reinterpret_cast<const f32x4*>(p.gn_ss + ((long)(row / p.rows_per_image) * TC + c * 8) * 2);
          float x[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[2 * e] = __builtin_bit_cast(float, ba[t][e] << 16);
            x[2 * e + 1] = __builtin_bit_cast(float, ba[t][e] & 0xffff0000u);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 s4 = ss[e];                                     // channels 2 e, 2 e + 1
            x[2 * e] = __builtin_fmaf(x[2 * e], s4[0], s4[1]);
            x[2 * e + 1] = __builtin_fmaf(x[2 * e + 1], s4[2], s4[3]);
          }
          ba[t] = (u32x4){pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])};
        }
        put_tile(TX_OFF, ba, t0);
      } else {
        put_tile(TX_OFF, ba, t0);
        load_tile(p.r1, p.ldr1, row0, ba, t0);
        put_tile(TB_OFF, ba, t0);
      }
    }
    __syncthreads();

    // ================================================================ first layer: + bias (+ residual), LayerNorm -> X, result -> B -> HBM
    k_loop();
    {
      float rs[TMI], rq[TMI];
#pragma unroll
      for (int i = 0; i < TMI; ++i) rs[i] = rq[i] = 0.f;
#pragma unroll
      for (int j = 0; j < TNI; ++j) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias_pre + ncol0 + j * 16);
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
          float r4[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (!GNIN) {
            const u32x2 u = *reinterpret_cast<const u32x2*>(smem + TB_OFF + pc_off(i, j));
            r4[0] = __builtin_bit_cast(float, u[0] << 16); r4[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
            r4[2] = __builtin_bit_cast(float, u[1] << 16); r4[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = (acc[j][i][e] + bb[e]) + r4[e];
            acc[j][i][e] = x;                       // (the accumulator registers carry the fp32 row values on)
            rs[i] += x;
            rq[i] = __builtin_fmaf(x, x, rq[i]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TMI; ++i) {
        rs[i] += __shfl_xor(rs[i], 16, 64); rq[i] += __shfl_xor(rq[i], 16, 64);
        rs[i] += __shfl_xor(rs[i], 32, 64); rq[i] += __shfl_xor(rq[i], 32, 64);
      }
      __syncthreads();        // every wave is through its k loop: X is dead and carries the statistics from here
      if (fq == 0) {
#pragma unroll
        for (int i = 0; i < TMI; ++i) scratch[wave * TBM + i * 16 + fr] = make_float2(rs[i], rq[i]);
      }
      __syncthreads();
      float mean[TMI], rstd[TMI];
#pragma unroll
      for (int i = 0; i < TMI; ++i) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < TNW; ++w) {
          const float2 t = scratch[w * TBM + i * 16 + fr];
          s += t.x; q += t.y;
        }
        mean[i] = s / (float)TC;
        float var = q / (float)TC - mean[i] * mean[i];
        var = var < 0.f ? 0.f : var;
        rstd[i] = rsqrtf(var + p.eps);
      }
      __syncthreads();        // the statistics have been read: X takes the normalised rows
#pragma unroll
      for (int j = 0; j < TNI; ++j) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(p.gamma + ncol0 + j * 16);
        const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + ncol0 + j * 16);
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
          const int off = pc_off(i, j);
          u32x2 o, n2;
          o[0] = pack_bf16x2(acc[j][i][0], acc[j][i][1]);
          o[1] = pack_bf16x2(acc[j][i][2], acc[j][i][3]);
          *reinterpret_cast<u32x2*>(smem + TB_OFF + off) = o;
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[j][i][e] - mean[i]) * rstd[i], gg[e], be[e]);
          n2[0] = pack_bf16x2(y[0], y[1]);
          n2[1] = pack_bf16x2(y[2], y[3]);
          *reinterpret_cast<u32x2*>(smem + TX_OFF + off) = n2;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();          // LayerNorm rows in X, result rows in B
    store_tile(TB_OFF, p.out_mid, p.ldmid, row0);

    // ================================================================ the layers behind the LayerNorm: result -> B -> HBM
#pragma unroll
    for (int l = 0; l < NPOST; ++l) {
      k_loop();
      __syncthreads();        // B has been read out by everybody (each wave's stores precede its k loop)
      if (GNIN && l == NPOST - 1) {
        // v^T: staged transposed, [feature][64 rows] (128 bytes per feature, 16-byte chunks XORed with the feature's low bits)
#pragma unroll
        for (int j = 0; j < TNI; ++j)
#pragma unroll
          for (int i = 0; i < TMI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int n = ncol0 + j * 16 + e, m = i * 16 + fr;
              *reinterpret_cast<bf16_t*>(smem + TB_OFF + n * 128 + ((((m >> 3) ^ n) & 7) << 4) + (m & 7) * 2) =
                  (bf16_t)(pack_bf16x2(acc[j][i][e], 0.f) & 0xffffu);
            }
        __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.out, (short)0, (int)(((long)(TC - 1) * p.ldo + p.M) * 2), 0x00020000);
        const int ld2 = opaque((int)p.ldo * 2);
        constexpr int GRP = 5;
#pragma unroll
        for (int t0 = 0; t0 < TIO; t0 += GRP) {
          u32x4 buf[GRP];
          unsigned off[GRP];
#pragma unroll
          for (int t = 0; t < GRP; ++t) {
            const int q = opaque((t0 + t) * 256) + tid, n = q >> 3, c = q & 7;       // feature, 8-row chunk
            buf[t] = *reinterpret_cast<const u32x4*>(smem + TB_OFF + n * 128 + (((c ^ n) & 7) << 4));
            off[t] = row0 + c * 8 < p.M ? (unsigned)(n * ld2 + (row0 + c * 8) * 2) : 0x7ffffff0u;
          }
#pragma unroll
          for (int t = 0; t < GRP; ++t) __builtin_amdgcn_raw_buffer_store_b128(buf[t], rs, off[t], 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      } else {
        acc_to_tile(TB_OFF);
        __syncthreads();
        bf16_t* dst = (GNIN || NPOST > 1) && l < NPOST - 1 ? p.out_p[l] : p.out;
        const long ld = (GNIN || NPOST > 1) && l < NPOST - 1 ? p.ldp[l] : p.ldo;
        store_tile(TB_OFF, dst, ld, row0);
      }
    }
    __syncthreads();          // B and X are free for the next tile
  }
}

template <int NPOST, bool GNIN>
int launch_tile(const TileParams& k, hipStream_t st) {
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&lin_tile_kernel<NPOST, GNIN>), TLDS)) return rc;
  int cus = 0;
  if (int rc = hedit_cu_count(&cus)) return rc;
  const int ntiles = cdiv(k.M, TBM);
  hipLaunchKernelGGL((lin_tile_kernel<NPOST, GNIN>), dim3(ntiles < TBPC * cus ? ntiles : TBPC * cus), dim3(256), TLDS, st, k);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

}  // namespace

size_t lin_tile_stream_bytes(int layers) { return (size_t)layers * TKS * TSET; }

int lin_tile_pack_launch(const float* w, int layer, float scale, int layers, bf16_t* stream, hipStream_t st) {
  ARG_CHECK(w && stream && (layers == 2 || layers == 4) && layer >= 0 && layer < layers, "lin_tile_pack: args");
  hipLaunchKernelGGL(lin_tile_pack_kernel, dim3(cdiv(TKS * (TSET / 16), 256)), dim3(256), 0, st, w, layer, scale, stream);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int lin_tile_launch(const LinChainParams& c, hipStream_t st) {
  ARG_CHECK(c.C == TC, "lin_tile: built for another channel count");
  ARG_CHECK(c.M > 0 && c.a && c.stream && c.gamma && c.beta && c.bias_pre && c.out_mid && c.out, "lin_tile: null");
  ARG_CHECK(c.lda % 8 == 0 && c.ldmid % 8 == 0 && c.ldo % 8 == 0, "lin_tile: rows must be 16-byte aligned");
  TileParams k{};
  k.a = c.a; k.lda = c.lda; k.bias_pre = c.bias_pre; k.gamma = c.gamma; k.beta = c.beta; k.eps = c.eps;
  k.stream = c.stream; k.M = c.M; k.out_mid = c.out_mid; k.ldmid = c.ldmid; k.out = c.out; k.ldo = c.ldo;
  long ldmax = c.lda > c.ldmid ? c.lda : c.ldmid;
  if (!c.gn_ss) {
    ARG_CHECK(c.r1 && c.ldr1 % 8 == 0, "lin_tile: residual rows");
    if (c.ldo > ldmax) ldmax = c.ldo;
    if (c.ldr1 > ldmax) ldmax = c.ldr1;
    ARG_CHECK((long)c.M * ldmax * 2 < (1L << 31), "lin_tile: tensor beyond the 2 GB buffer window");
    k.r1 = c.r1; k.ldr1 = c.ldr1;
    return launch_tile<1, false>(k, st);
  }
  ARG_CHECK(c.rows_per_image > 0 && c.M % c.rows_per_image == 0 && c.M % 8 == 0 && c.out_q && c.out_k && c.ldq % 8 == 0 && c.ldk % 8 == 0,
            "lin_tile: GroupNorm'd input form (whole images, M % 8 == 0; q, k, v^T outputs)");
  if (c.ldq > ldmax) ldmax = c.ldq;
  if (c.ldk > ldmax) ldmax = c.ldk;
  ARG_CHECK((long)c.M * ldmax * 2 < (1L << 31) && (long)TC * c.ldo * 2 < (1L << 31), "lin_tile: tensor beyond the 2 GB buffer window");
  k.gn_ss = c.gn_ss; k.rows_per_image = c.rows_per_image;
  k.out_p[0] = c.out_q; k.ldp[0] = c.ldq; k.out_p[1] = c.out_k; k.ldp[1] = c.ldk;
  return launch_tile<3, true>(k, st);
}
