// EXPERIMENT (round 4), not part of the product library: built into a side library by tools/experiments/build_xffn.sh and
// timed against csrc/ffn.hip by tools/experiments/xffn_bench.py.  Outcome: correct (1.7e-3 vs the
// fp32 reference, rows independent bit for bit) and its inner loop is the fast one (tools/ubench/chain_core.hip: 1.5 PFLOP/s),
// but as a whole 821 TFLOP/s against 939 for ffn.hip: with one wave per SIMD the ~900 VALU instructions of a chunk's
// bias + GELU + bf16 hand-over cost as long as its 250 MFMAs, and hiding them needs a second FF1 accumulator set the
// 256 AGPRs do not have next to Y.
//
// The token-local tail of a BasicTransformerBlock at the C = 320 level of the SD UNet as ONE kernel per row tile -- round 4's
// mapping, "the tile in LDS, the weights straight from L2 into registers" (what ffn.hip computes, oracle/sd_unet.py
// BasicTransformerBlock / Transformer2DModel.proj_out):
//
//     t2  = attn2.to_out(a) + b + t1
//     t3  = t2 + FF2( GEGLU( FF1( LayerNorm(t2) ) ) ) + b2          (diffusers FeedForward with GEGLU)
//     out = proj_out(t3) + b + x
//
// Why a second mapping.  ffn.hip keeps 32 ROWS per wave in registers and streams every weight fragment through LDS to
// every wave: each MFMA consumes its own 1 KB fragment read, four waves = 128 B/clk = the LDS port, and the kernel stops at
// 0.34-0.41 of the MFMA rate.  Here the roles are swapped:
//   * a block = 4 waves (one per SIMD) owns a tile of BM = 80 rows.  The activation tile X[80][320] (bf16, 50 KB) lives in
//     LDS, written once per layer; wave w owns the output COLUMNS [80 w, 80 w + 80) of every layer.
//   * per 32-deep k-step a wave reads the tile's 5 activation fragments from LDS (5 KB for 25 MFMAs: 0.2 KB per MFMA
//     instead of 1 KB) and takes its 5 weight fragments (16 x 32, 1 KB each) straight from global memory / L2 into
//     VGPRs -- nobody else needs them, so they never touch LDS.  The weights are packed at load time into a stream in
//     consumption order whose k-step holds, per wave, 5 contiguous lane-linear KB (one coalesced 16-byte load per lane
//     and fragment); the kernel keeps a ring of five k-steps in registers, four ahead (100 VGPRs), loads the compiler
//     counts itself (plain loads, no LDS-DMA anywhere in this kernel, so no wait is ever hand-counted).
//     tools/ubench/chain_core.hip: this inner loop alone runs at 1.5 PFLOP/s over the chip (96-100 GB/s of weights per
//     CU arrive this way; the LDS-DMA path saturates at 55).
//   * v_mfma_f32_16x16x32_bf16 with the weight fragment as the A operand, like gemm.hip: a lane ends up with 4
//     consecutive output columns of one row, the layout of the bias / residual / LayerNorm passes and of the 8-byte LDS
//     writes that hand a layer's result to the next one as activation tile.
//   * the feed-forward runs in 8 chunks of 160 hidden units: FF1 chunk (320 value | gate rows interleaved so that a
//     lane holds (v, g) pairs) -> bias -> v * gelu(g) in registers -> bf16 G[80][160] in LDS -> FF2 chunk accumulates
//     onto Y, which starts as t2 + b2 (the residual stream stays in fp32 registers for the whole block).
//   * tile I/O: the three input tiles (a, t1, x) come in with coalesced 16-byte loads through registers into the
//     swizzled LDS image, the result leaves the same way; t1 and x land behind the arithmetic that precedes their use.
// LDS: X 50 KB | B 50 KB (t1 -> x -> result) | G 30 KB (384-byte rows: whole 128-byte swizzle groups) | 2.5 KB LayerNorm scratch.
// Registers (one wave per SIMD, 512): accumulators Y, F 2 x 100 AGPR, activation fragments 2 x 20 AGPR, weight ring 100 VGPR --
// 96-row tiles (2 x 120 + 48 + 100) left the allocator ~30 registers short and it spilled loop-invariant addresses, whose
// reloads inside the chunk loop drained the weight ring; 80 rows divide the bench's M = 120 / 96 / 48 x 4096 evenly.
// Every output row depends on its own input rows only and every summation order is fixed (k ascending in the MFMA chains,
// LayerNorm statistics: a lane's 20 values in order, the four lanes of a row by xor-shuffle, the four waves in order),
// so results do not depend on the batch.
#include <type_traits>
#include <utility>

#include "../../h-edit_amd/csrc/common.h"
#include "../../h-edit_amd/csrc/kernels.h"
#include "xffn_decl.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int XC = 320;                  // channels
constexpr int XBM = 80;                  // rows per tile
constexpr int XNW = 4;                   // waves = column slices
constexpr int XWN = XC / XNW;            // 80 output columns per wave
constexpr int XNI = XWN / 16;            // 5 weight fragments per wave and k-step
constexpr int XMI = XBM / 16;            // 6 activation fragments per k-step
constexpr int XKS = XC / 32;             // 10 k-steps of a 320-deep layer
constexpr int XHC = 160;                 // hidden units per feed-forward chunk
constexpr int XNCH = 4 * XC / XHC;       // 8 chunks
constexpr int XKS2 = XHC / 32;           // 5 k-steps of an FF2 chunk
constexpr int XSET = XNW * XNI * 1024;   // stream bytes per k-step (20 KB)
constexpr int XNPOS = XKS + XNCH * (XKS + XKS2) + XKS;       // 140 k-steps per tile
constexpr int XRING = 5;                 // k-steps of weights in registers (every phase is a multiple: static ring index)
constexpr int XROW = XC * 2;             // bytes per activation row
constexpr int XTILE = XBM * XROW;        // 61440
constexpr int XGROW = 384;               // bytes per G row (160 bf16 = 320 B, padded to whole 128-byte groups)
constexpr int X_OFF = 0, B_OFF = XTILE, G_OFF = 2 * XTILE, S_OFF = G_OFF + XBM * XGROW;
constexpr int XLDS = S_OFF + XNW * XBM * 8;
static_assert(XLDS <= 160 * 1024, "tiles + scratch must fit the LDS");
static_assert(XKS % XRING == 0 && XKS2 % XRING == 0, "every phase a multiple of the register ring");
constexpr int XIOR = XBM % 6 == 0 ? 6 : 5;   // rows per I/O group (XIOR x 40 chunks <= 256 threads)
static_assert(XBM % XIOR == 0 && XIOR * (XC / 8) <= 256, "tile I/O: groups of XIOR rows");

__host__ __device__ constexpr size_t xffn_stream_bytes_c() { return (size_t)XNPOS * XSET; }

// k-step `pos` of the stream (0 .. XNPOS-1): which layer, which k-step inside it
//   [0, 10)                      leading linear (attn2.to_out)
//   [10 + 15 h, 10 + 15 h + 10)  FF1 of hidden chunk h;   the next 5: FF2 of chunk h
//   [130, 140)                   trailing linear (proj_out)
// one thread per 16-byte piece.  which: 0 = leading linear (w [C][C]), 1 = FF1 (w [8C][C]: value rows, then gate rows),
// 2 = FF2 (w [C][4C]), 3 = trailing linear (w [C][C]); every piece belongs to exactly one of the four calls.
__global__ __launch_bounds__(256) void xffn_pack_kernel(const float* __restrict__ w, int which, bf16_t* __restrict__ stream) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)XNPOS * (XSET / 16);
  if (idx >= total) return;
  const int pos = (int)(idx / (XSET / 16));
  const int r = (int)(idx - (long)pos * (XSET / 16));
  const int wave = r / (XNI * 64), j = (r / 64) % XNI, lane = r & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int n = wave * XWN + j * 16 + fr;                 // row of the layer's (possibly interleaved) weight matrix
  int owner, ks;
  long src_row, src_ld, k0;
  if (pos < XKS) {
    owner = 0; ks = pos; src_row = n; src_ld = XC; k0 = ks * 32;
  } else if (pos >= XNPOS - XKS) {
    owner = 3; ks = pos - (XNPOS - XKS); src_row = n; src_ld = XC; k0 = ks * 32;
  } else {
    const int q = pos - XKS, h = q / (XKS + XKS2), s = q % (XKS + XKS2);
    if (s < XKS) {                                         // FF1 chunk h: packed row n = 2 u + gate of hidden unit h*160 + u
      owner = 1; ks = s;
      const int u = n >> 1, gate = n & 1;
      src_row = (gate ? 4 * XC : 0) + h * XHC + u; src_ld = XC; k0 = ks * 32;
    } else {                                               // FF2 chunk h: output column n, hidden units h*160 + ks*32 ...
      owner = 2; ks = s - XKS;
      src_row = n; src_ld = 4 * XC; k0 = (long)h * XHC + ks * 32;
    }
  }
  if (owner != which) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w[src_row * src_ld + k0 + fq * 8 + e];
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(stream) + idx * 16) = pack8(v);
}

// ff.net.0.proj.bias [8C] -> the order of the packed FF1 rows: out[h][n] = n even ? b[h*160 + n/2] : b[4C + h*160 + n/2]
__global__ __launch_bounds__(256) void xffn_pack_bias_kernel(const float* __restrict__ b1, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 8 * XC) return;
  const int h = i / XC, n = i % XC;
  out[i] = b1[((n & 1) ? 4 * XC : 0) + h * XHC + (n >> 1)];
}

struct XParams {
  const bf16_t* a; long lda;             // leading layer input
  const bf16_t* t1; long ldt1;           // its residual
  const bf16_t* x; long ldx;             // residual of the trailing layer
  const float* bias_pre; const float* gamma; const float* beta; float eps;
  const bf16_t* stream; const float* bias1p; const float* bias2; const float* bias_post;
  bf16_t* out; long ldo;
  int M;
};

template <class F, int... I>
__device__ __forceinline__ void xstatic_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void xstatic_for(F&& f) {
  xstatic_for_impl(f, std::make_integer_sequence<int, N>{});
}

// byte offset of 16-byte chunk c of row m in a swizzled row-major tile (chunk index XORed with the row inside its 128-byte group)
__device__ __forceinline__ int xswz(int m, int c, int row_bytes) { return m * row_bytes + (((c & ~7) | ((c ^ m) & 7)) << 4); }

__global__ __launch_bounds__(256, 1) void xffn_kernel(XParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int ntiles = (p.M + XBM - 1) / XBM;

  // ---- fragment read offsets (activation fragment i: rows 16 i + fr, k chunk 4 ks + fq) for the two row strides
  int xrd[XMI], grd[XMI];
#pragma unroll
  for (int i = 0; i < XMI; ++i) {
    const int m = i * 16 + fr;
    xrd[i] = m * XROW + ((fq ^ (m & 7)) << 4);
    grd[i] = m * XGROW + ((fq ^ (m & 7)) << 4);
  }
  // ---- this lane's 4 consecutive columns of block j: n = 80 wave + 16 j + 4 fq; 8-byte pieces of rows 16 i + fr
  const int ncol0 = wave * XWN + fq * 4;
  auto pc_off = [&](int i, int j, int row_bytes) __attribute__((always_inline)) {      // byte offset of the lane's piece (i, j) in a tile
    const int m = i * 16 + fr, n = ncol0 + j * 16;
    return xswz(m, n >> 3, row_bytes) + ((n & 4) << 1);
  };

  // ---- weight ring: set s of the ring = the wave's 5 fragments of one k-step
  const u32x4* const wbase = reinterpret_cast<const u32x4*>(p.stream) + (size_t)wave * XNI * 64 + lane;
  bf16x8 wr[XRING][XNI];
  int ppos = 0;                                        // stream k-step the NEXT ring load fetches (cyclic over the tiles)
  auto load_set = [&](auto s_) __attribute__((always_inline)) {
    constexpr int s = decltype(s_)::value;
    const u32x4* src = wbase + (size_t)ppos * (XSET / 16);
#pragma unroll
    for (int j = 0; j < XNI; ++j) wr[s][j] = __builtin_bit_cast(bf16x8, src[j * 64]);
    ppos = ppos + 1 == XNPOS ? 0 : ppos + 1;
  };
  xstatic_for<XRING - 1>([&](auto s_) { load_set(s_); });

  // one phase = NK k-steps over the activation tile at `tile_off` (fragment offsets rd), accumulating onto acc (FRESH: the
  // first k-step starts from the inline constant 0 -- nobody writes 120 zeros).  The MFMAs are inline asm so that the
  // register files are ours to choose: accumulators and activation fragments in AGPRs (200 + 40 of 256), weight ring in VGPRs -- left to itself the allocator
  // mixes the files, runs out of one of them and spills into the k loops, whose reloads drain the weight ring
  // (vmcnt(0) between MFMAs).  What the compiler does not do for asm MFMAs is hazard padding: settle() below.
  auto k_loop = [&](auto nk_, auto fresh_, f32x4 (&acc)[XNI][XMI], int tile_off, const int (&rd)[XMI]) __attribute__((always_inline)) {
    constexpr int NK = decltype(nk_)::value;
    constexpr bool FRESH = decltype(fresh_)::value;
    bf16x8 xa[XMI], xb[XMI];
    auto read_x = [&](bf16x8 (&xf)[XMI], int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < XMI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(smem + tile_off + ((rd[i] ^ ((ks & 1) << 6)) + (ks >> 1) * 128));
    };
    read_x(xa, 0);
    xstatic_for<NK>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      load_set(std::integral_constant<int, (ks + XRING - 1) % XRING>{});       // four k-steps ahead, into the set consumed last
      bf16x8 (&xc)[XMI] = (ks & 1) ? xb : xa;
      bf16x8 (&xn)[XMI] = (ks & 1) ? xa : xb;
      if constexpr (ks + 1 < NK) read_x(xn, ks + 1);
#pragma unroll
      for (int j = 0; j < XNI; ++j)
#pragma unroll
        for (int i = 0; i < XMI; ++i) {
          if constexpr (FRESH && ks == 0) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc[j][i]) : "v"(wr[ks % XRING][j]), "a"(xc[i]));
          else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wr[ks % XRING][j]), "a"(xc[i]));
        }
      __builtin_amdgcn_sched_barrier(0);             // (keeps later k-steps' reads where they are: register pressure)
    });
  };
  // the accumulators of the last asm MFMAs become readable by the VALU (8-pass MFMA: the result lands 16+ cycles after issue)
  auto settle = [&](f32x4 (&acc)[XNI][XMI]) __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < XNI; ++j)
#pragma unroll
      for (int i = 0; i < XMI; ++i) asm volatile("" : "+a"(acc[j][i]));
  };
  // ---- coalesced tile I/O by the first XIOR x 40 threads: thread -> (row tid / 40 of a group of XIOR rows, chunk tid % 40), XBM / XIOR
  // groups per tile; bounds-checked buffer accesses (rows beyond M read as zeros / are not written: no branches).  The row
  // stride is made opaque per use: hoisted in front of the tile loop the 16 offsets of every site would live in scratch.
  constexpr int XIOG = XBM / XIOR;
  const int io_r = tid / (XC / 8), io_c = tid - io_r * (XC / 8);
  const bool io_on = tid < XIOR * (XC / 8);
  auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); return v; };
#if defined(__HIP_DEVICE_COMPILE__)
  auto rsrc = [&](const bf16_t* ptr, long ld) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ptr), (short)0, (int)(((long)p.M - 1) * ld * 2 + XROW), 0x00020000);
  };
#endif
  auto load_tile = [&](const bf16_t* src, long ld, int row0, u32x4 (&buf)[XIOG]) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, ld);
    const int ld2 = opaque((int)ld * 2);
    const unsigned voff = io_on ? (unsigned)((row0 + io_r) * ld2 + io_c * 16) : 0x7ffffff0u;
#pragma unroll
    for (int t = 0; t < XIOG; ++t) buf[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, t * XIOR * ld2, 0));
#else
    (void)src; (void)ld; (void)row0; (void)buf;
#endif
  };
  auto put_tile = [&](int tile_off, const u32x4 (&buf)[XIOG]) __attribute__((always_inline)) {
    if (io_on) {
#pragma unroll
      for (int t = 0; t < XIOG; ++t) *reinterpret_cast<u32x4*>(smem + tile_off + xswz(opaque(t * XIOR) + io_r, io_c, XROW)) = buf[t];
    }
  };
  auto store_tile = [&](int tile_off, bf16_t* dst, long ld, int row0) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = rsrc(dst, ld);
    const int ld2 = opaque((int)ld * 2);
    const unsigned voff = io_on ? (unsigned)((row0 + io_r) * ld2 + io_c * 16) : 0x7ffffff0u;
    constexpr int GRP = 4;
#pragma unroll
    for (int t0 = 0; t0 < XIOG; t0 += GRP) {
      u32x4 buf[GRP];
#pragma unroll
      for (int t = 0; t < GRP; ++t) buf[t] = *reinterpret_cast<const u32x4*>(smem + tile_off + xswz(opaque((t0 + t) * XIOR) + (io_on ? io_r : 0), io_c % (XC / 8), XROW));
#pragma unroll
      for (int t = 0; t < GRP; ++t) __builtin_amdgcn_raw_buffer_store_b128(buf[t], rs, voff, (t0 + t) * XIOR * ld2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    (void)tile_off; (void)dst; (void)ld; (void)row0;
#endif
  };
  auto unpack4 = [](const u32x2& u, float (&f)[4]) __attribute__((always_inline)) {
    f[0] = __builtin_bit_cast(float, u[0] << 16); f[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, u[1] << 16); f[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
  };
  auto pack4 = [](const f32x4& v) __attribute__((always_inline)) {
    u32x2 o;
    o[0] = pack_bf16x2(v[0], v[1]);
    o[1] = pack_bf16x2(v[2], v[3]);
    return o;
  };
  float2* const scratch = reinterpret_cast<float2*>(smem + S_OFF);       // [wave][row]: (sum, sum of squares) of the wave's 80 columns

  f32x4 Y[XNI][XMI], F[XNI][XMI];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * XBM;
    {   // the tile's rows of a -> X and of t1 -> B
      u32x4 ba[XIOG], bt[XIOG];
      load_tile(p.a, p.lda, row0, ba);
      load_tile(p.t1, p.ldt1, row0, bt);
      // the per-column parameters of the first epilogue (b_pre | gamma | beta | b2) go through the G region, which is idle
      // until the first feed-forward chunk: read from global memory where they are used, each load would be YOUNGER than
      // the weight sets in flight, and waiting for it would drain the ring (one in-order queue)
      f32x4 pv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int idx = t * 256 + tid;                        // float4 index over the four vectors of 80 float4 each
        const int which = idx / (XC / 4), c4 = idx - which * (XC / 4);
        const float* src = which == 0 ? p.bias_pre : (which == 1 ? p.gamma : (which == 2 ? p.beta : p.bias2));
        pv[t] = idx < XC ? *reinterpret_cast<const f32x4*>(src + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      put_tile(X_OFF, ba);
      put_tile(B_OFF, bt);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t * 256 + tid < XC) *reinterpret_cast<f32x4*>(smem + G_OFF + (t * 256 + tid) * 16) = pv[t];
    }
    __syncthreads();

    // ================================================================ t2 = to_out(a) + b + t1;  LayerNorm -> X;  Y = t2 + b2
    k_loop(std::integral_constant<int, XKS>{}, std::true_type{}, Y, X_OFF, xrd);
    settle(Y);
    float rs[XMI], rq[XMI];
#pragma unroll
    for (int i = 0; i < XMI; ++i) rs[i] = rq[i] = 0.f;
#pragma unroll
    for (int j = 0; j < XNI; ++j) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(smem + G_OFF + (ncol0 + j * 16) * 4);
#pragma unroll
      for (int i = 0; i < XMI; ++i) {
        float r4[4];
        unpack4(*reinterpret_cast<const u32x2*>(smem + B_OFF + pc_off(i, j, XROW)), r4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (Y[j][i][e] + bb[e]) + r4[e];
          Y[j][i][e] = v;
          rs[i] += v;
          rq[i] = __builtin_fmaf(v, v, rq[i]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);       // (one column block at a time: keeps the live set of these passes small)
    }
#pragma unroll
    for (int i = 0; i < XMI; ++i) {
      rs[i] += __shfl_xor(rs[i], 16, 64); rq[i] += __shfl_xor(rq[i], 16, 64);
      rs[i] += __shfl_xor(rs[i], 32, 64); rq[i] += __shfl_xor(rq[i], 32, 64);
      if (fq == 0) scratch[wave * XBM + i * 16 + fr] = make_float2(rs[i], rq[i]);
    }
    __syncthreads();        // statistics of all four column slices are in LDS; everybody is done reading X and B
    {   // t1 has been consumed: the rows of x take its place in B (needed by the trailing layer's epilogue)
      u32x4 bx[XIOG];
      load_tile(p.x, p.ldx, row0, bx);
      put_tile(B_OFF, bx);
    }
    {
      float mean[XMI], rstd[XMI];
#pragma unroll
      for (int i = 0; i < XMI; ++i) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < XNW; ++w) {
          const float2 v = scratch[w * XBM + i * 16 + fr];
          s += v.x; q += v.y;
        }
        mean[i] = s / (float)XC;
        float var = q / (float)XC - mean[i] * mean[i];
        var = var < 0.f ? 0.f : var;
        rstd[i] = rsqrtf(var + p.eps);
      }
#pragma unroll
      for (int j = 0; j < XNI; ++j) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(smem + G_OFF + (XC + ncol0 + j * 16) * 4);
        const f32x4 be = *reinterpret_cast<const f32x4*>(smem + G_OFF + (2 * XC + ncol0 + j * 16) * 4);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(smem + G_OFF + (3 * XC + ncol0 + j * 16) * 4);
#pragma unroll
        for (int i = 0; i < XMI; ++i) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf((Y[j][i][e] - mean[i]) * rstd[i], gg[e], be[e]);
          *reinterpret_cast<u32x2*>(smem + X_OFF + pc_off(i, j, XROW)) = pack4(o);
          Y[j][i] = Y[j][i] + b2;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();        // LayerNorm(t2) is in X

    // ================================================================ feed-forward, 8 chunks of 160 hidden units
#pragma unroll 1
    for (int h = 0; h < XNCH; ++h) {
      f32x4 b1v[XNI];                              // requested in front of the k loop: older than the weight sets in flight at its use
#pragma unroll
      for (int j = 0; j < XNI; ++j) b1v[j] = *reinterpret_cast<const f32x4*>(p.bias1p + h * XC + ncol0 + j * 16);
      k_loop(std::integral_constant<int, XKS>{}, std::true_type{}, F, X_OFF, xrd);
      settle(F);
      // bias, v * gelu(g): the lane's 4 columns of block j are (v, g, v, g) of hidden units 40 wave + 8 j + 2 fq + {0, 1}
#pragma unroll
      for (int j = 0; j < XNI; ++j) {
        const f32x4 bb = b1v[j];
#pragma unroll
        for (int i = 0; i < XMI; ++i) {
          const f32x4 v = F[j][i] + bb;
          const f32x2 g2 = mul_gelu2((f32x2){v[0], v[2]}, (f32x2){v[1], v[3]});
          const int m = i * 16 + fr;
          *reinterpret_cast<uint32_t*>(smem + G_OFF + xswz(m, wave * XNI + j, XGROW) + fq * 4) = pack_bf16x2(g2[0], g2[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();      // G is complete
      k_loop(std::integral_constant<int, XKS2>{}, std::false_type{}, Y, G_OFF, grd);
      __syncthreads();      // G may be overwritten
    }

    // ================================================================ out = proj_out(t3) + b + x
    settle(Y);
#pragma unroll
    for (int j = 0; j < XNI; ++j)
#pragma unroll
      for (int i = 0; i < XMI; ++i) *reinterpret_cast<u32x2*>(smem + X_OFF + pc_off(i, j, XROW)) = pack4(Y[j][i]);
    __syncthreads();        // t3 is in X
    f32x4 bpv[XNI];
#pragma unroll
    for (int j = 0; j < XNI; ++j) bpv[j] = *reinterpret_cast<const f32x4*>(p.bias_post + ncol0 + j * 16);
    k_loop(std::integral_constant<int, XKS>{}, std::true_type{}, F, X_OFF, xrd);
    settle(F);
#pragma unroll
    for (int j = 0; j < XNI; ++j) {
      const f32x4 bb = bpv[j];
#pragma unroll
      for (int i = 0; i < XMI; ++i) {
        const int off = B_OFF + pc_off(i, j, XROW);
        float r4[4];
        unpack4(*reinterpret_cast<const u32x2*>(smem + off), r4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (F[j][i][e] + bb[e]) + r4[e];
        *reinterpret_cast<u32x2*>(smem + off) = pack4(o);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();        // the result tile is in B; X is free for the next tile's rows
    store_tile(B_OFF, p.out, p.ldo, row0);
    __syncthreads();        // (B is read out before the next tile's t1 rows are written into it)
  }
}

}  // namespace

size_t xffn_stream_bytes() { return xffn_stream_bytes_c(); }
size_t xffn_bias_bytes() { return (size_t)8 * XC * sizeof(float); }

int xffn_pack_launch(const float* w, int which, bf16_t* stream, hipStream_t st) {
  ARG_CHECK(w && stream && which >= 0 && which <= 3, "xffn_pack: args");
  const long total = (long)XNPOS * (XSET / 16);
  hipLaunchKernelGGL(xffn_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, which, stream);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int xffn_pack_bias_launch(const float* b1, float* out, hipStream_t st) {
  ARG_CHECK(b1 && out, "xffn_pack_bias: args");
  hipLaunchKernelGGL(xffn_pack_bias_kernel, dim3(cdiv(8 * XC, 256)), dim3(256), 0, st, b1, out);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int xffn_launch(const FfnParams& f, hipStream_t st) {
  ARG_CHECK(f.C == XC, "xffn: exists for C = 320");
  ARG_CHECK(f.M > 0 && f.a && f.x && f.r2 && f.out && f.stream && f.bias1p && f.bias2 && f.gamma && f.beta && f.bias_pre && f.bias_post, "xffn: null");
  ARG_CHECK(f.lda % 8 == 0 && f.ldx % 8 == 0 && f.ldr2 % 8 == 0 && f.ldo % 8 == 0, "xffn: rows must be 16-byte aligned");
  XParams k{};
  k.a = f.a; k.lda = f.lda; k.t1 = f.x; k.ldt1 = f.ldx; k.x = f.r2; k.ldx = f.ldr2;
  k.bias_pre = f.bias_pre; k.gamma = f.gamma; k.beta = f.beta; k.eps = f.eps;
  k.stream = f.stream; k.bias1p = f.bias1p; k.bias2 = f.bias2; k.bias_post = f.bias_post;
  k.out = f.out; k.ldo = f.ldo; k.M = f.M;
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&xffn_kernel), XLDS)) return rc;
  int cus = 0;
  if (int rc = hedit_cu_count(&cus)) return rc;
  const int ntiles = cdiv(f.M, XBM);
  hipLaunchKernelGGL(xffn_kernel, dim3(ntiles < cus ? ntiles : cus), dim3(256), XLDS, st, k);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
