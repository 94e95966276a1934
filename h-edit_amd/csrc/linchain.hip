// The token-local projections AROUND the two attentions of a BasicTransformerBlock at the C = 320 level of the SD UNet,
// each group as ONE kernel per 128-row tile (oracle/sd_unet.py: Transformer2DModel.norm / proj_in, BasicTransformerBlock
// norm1 / attn1.to_q|k|v, attn1.to_out, norm2, attn2.to_q):
//
//   two layers:   t1 = attn1.to_out(a) + t0         (written out: the residual stream)
//                 q2 = attn2.to_q( LayerNorm(t1) )
//   four layers:  t0 = proj_in( GroupNorm(x) )      (GroupNorm applied on the fly from per-(image, channel) scale / shift)
//                 q | k = attn1.to_q | to_k ( LayerNorm(t0) ),   v^T = attn1.to_v( LayerNorm(t0) )^T
//
// Unfused these are 3 / 6 launches that move five / nine [M][320] activations through HBM at HBM speed (K = 320 GEMMs
// are memory-bound); here the inputs are read once and only the results the attention kernels need are written.
//
// gfx950 mapping -- the scheme of ffn.hip ("rows stay in registers, weights stream") with the tile I/O done the way the
// memory system wants it:
//   * a block = 4 waves (one per SIMD), a wave owns 32 rows; v_mfma_f32_32x32x16_bf16 with the weights as the 32-row
//     operand: Xn = the layer's input fragments (80 AGPR), O = the 320 x 32 fp32 accumulator (160 AGPR).  An accumulator
//     in MFMA result layout is the next layer's fragment layout once that layer's k is renumbered inside its groups of
//     16 (lin_unit_of_reg, folded into the weight packing).
//   * weights: a stream of 10 KB iterations (one 16-deep k-step = ten 1 KB lane-linear fragment images, 20 iterations
//     per layer) copied global -> LDS by DMA into a ring of seven, six ahead (60 KB in flight = what an LDS-DMA's
//     ~0.85 us latency needs at the consumption rate of 10 KB per 320 cycles), counted vmcnt, one barrier per iteration.
//   * tile I/O through a wave-private 20 KB staging area in LDS: a lane of the MFMA layout owns ONE ROW, so direct
//     global accesses are 64 separate requests per instruction (measured: 60 such loads + 40 such stores cost more than
//     the arithmetic of the whole chain).  Inputs: DMA of whole rows (lanes fetch consecutive 16-byte chunks) into the
//     staging area with the chunk index XOR-swizzled by the row, then ds_read of the fragment / accumulator layout.
//     Outputs: bf16(O) written to the staging area in the same swizzle, read back as row pieces, 16-byte coalesced
//     stores.  v^T leaves through the same area staged transposed per wave ([320 features][32 rows]: 64 contiguous
//     bytes of the [C][M] result per feature).
// Every output row depends on its own input row only and the summation order is fixed.
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int LC = 320;                  // channels
constexpr int LKS = LC / 16;             // k-steps = iterations per layer
constexpr int LNB = LC / 32;             // 32-wide output blocks = MFMAs (and fragments) per iteration
constexpr int IT_BYTES = LNB * 1024;
constexpr int RING = 7, AHEAD = 6;
constexpr int PPW = 3;                   // DMA pieces per wave and iteration (10 pieces; the surplus slots re-load the last)
constexpr int LEAD = 8;                  // fragment reads run this many MFMAs ahead
constexpr int ROW_BYTES = LC * 2;
constexpr int STAGE_OFF = RING * IT_BYTES;
constexpr int STAGE_BYTES = 32 * ROW_BYTES;            // per wave
constexpr int PAR_OFF = STAGE_OFF + 4 * STAGE_BYTES;   // gamma | beta | bias of the first layer (fp32)
constexpr int SS_OFF = PAR_OFF + 3 * LC * 4;           // GroupNorm (scale, shift) pairs of the tile's image, 3 x 1 KB DMA pieces
constexpr int SS_DMAS = 3;
constexpr int LDS_TOTAL = SS_OFF + SS_DMAS * 1024;
constexpr int BLOCK_ROWS = 128;
constexpr int CPR = LC / 8;              // 16-byte chunks per row
static_assert(LDS_TOTAL <= 160 * 1024, "ring + staging must fit the LDS");
static_assert(LNB + 2 >= LEAD && LEAD <= LNB, "the read-ahead reaches into the next iteration only");

__host__ __device__ constexpr int stream_iters(int layers) { return layers * LKS; }
// input feature (inside its group of 16) held by accumulator register r (0..7) of lane half hi
__host__ __device__ inline int lin_unit_of_reg(int r, int hi) { return (r >> 2) * 8 + hi * 4 + (r & 3); }
__host__ __device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(256) void lin_pack_kernel(const float* __restrict__ w, int layer, float scale, int layers, bf16_t* __restrict__ stream) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)stream_iters(layers) * (IT_BYTES / 16);
  if (idx >= total) return;
  const int it = (int)(idx / (IT_BYTES / 16));
  const int r = (int)(idx - (long)it * (IT_BYTES / 16));
  const int nb = r >> 6, l = r & 63, row = l & 31, hi = l >> 5;
  const int L = it / LKS, ks = it % LKS;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (L != layer) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = scale * w[(long)(nb * 32 + row) * LC + ks * 16 + (L == 0 ? hi * 8 + e : lin_unit_of_reg(e, hi))];
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(stream) + idx * 16) = pack8(v);
}

struct LinKernelParams {
  const bf16_t* a; long lda;
  const bf16_t* r1; long ldr1;
  const float* bias_pre;
  const float* gamma; const float* beta; float eps;
  const bf16_t* stream;
  bf16_t* out_mid; long ldmid;
  bf16_t* out_p[2]; long ldp[2];
  bf16_t* out; long ldo;
  const float* gn_ss; int rows_per_image;
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

// accumulator and activation fragment in AGPRs, the weight fragment in VGPRs (asm: the register files are ours to choose;
// what the compiler does not do for an asm MFMA is hazard padding -- see settle / publish below and ffn.hip)
__device__ __forceinline__ void mfma_l(f32x16& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, %0" : "+a"(acc) : "v"(w), "a"(a));
}
// first k-step of a layer: C = 0 (inline constant)
__device__ __forceinline__ void mfma_l0(f32x16& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_32x32x16_" MFMA_ST_SFX " %0, %1, %2, 0" : "=&a"(acc) : "v"(w), "a"(a));
}

// NPOST: layers behind the LayerNorm (1 or 3).  GNIN: the first layer's input is GroupNorm'd on the fly and there is no
// residual.  VT: the last result is written transposed.
//
// PERSISTENT: one block per CU walks over its tiles, so that
//   * the weight stream never stops -- it is read cyclically, the DMA of the next tile's first iterations is simply the
//     continuation of the ring (no per-tile prefill, no drain);
//   * the tile I/O overlaps the arithmetic: the next tile's input rows are DMA'd into the staging area while the last
//     layer runs, the residual rows while the first layer runs (the residual is added BEHIND the first layer), and the
//     stores of a result are in flight while the next layer runs.  The staging area is used strictly one thing after
//     the other: [input rows -> fragments] [final result -> stores] [residual rows -> accumulator] [first result ->
//     stores] ([q, k -> stores]) [next input rows ...].
// The bursts between two layers (row DMA, stores) sit in the same vector-memory queue as the weight DMA, so the counted
// waits of the ring are widened while a burst is younger than the iteration waited for: the first four iterations of a
// layer that follows E such operations allow E more outstanding ones (template parameter of the layer).  What this rests
// on: loads retire in issue order (what every counted vmcnt ring rests on); the weights those four waits are about were
// issued before the burst, a whole transition earlier; and a staged row is first touched a full layer (>= 7 us) after
// its DMA was issued and behind waits that no longer tolerate it.  A window may be WIDER than the burst in front of it
// (it then waits for less, which is only safe when everything it protects has landed anyway) but never narrower, so the
// bursts are sized per position (E_FIRST / E_MID / E_LAST) and issued for every tile alike -- bounds-checked buffer stores
// go out whether or not the row exists.  The FIRST tile of a block has no burst in front of its first layer; it needs
// none: the prologue drains the queue completely (vmcnt(0)) before the first barrier, so iterations 0 .. AHEAD-1 of the
// stream have landed whatever the first windows tolerate.
// (Round 3 put 20 identical place-holder DMAs in front of the first tile instead, to make its burst as long as every
//  other tile's, and waited vmcnt(burst).  LLVM's dead-store elimination collapses identical LDS-DMAs to the same LDS
//  address into ONE, the wait then tolerated 19 operations that did not exist, i.e. the 18 pieces of the stream's head
//  were never waited for: a first-tile race, invisible while the weights hit L2 and the prologue's own work gave them
//  time, visible as stale 32-column weight blocks under memory load -- what tests/test_gpu_chain_hazard.py caught, and what
//  made the in-layer variant of round 3, whose prologue is shorter, fail in every full sampling run.  DESIGN.md 5.0.)
// Carrying the row DMA inside the layers instead was tried and is NOT what this file does: DESIGN.md 5.0.
// SSG: the (scale, shift) pairs are read per row from global memory (images that are not whole 128-row tiles).
// SCHED: 0 = the product schedule described above.  The other two exist for tests/test_gpu_chain_hazard.py and are never
// launched by the executor:
//   1 = DRAINED: every counted wait of the kernel becomes vmcnt(0) lgkmcnt(0) in front of its barrier, so nothing in
//       LDS is ever read while ANY vector-memory operation of the wave is outstanding -- the schedule whose correctness
//       needs no argument about queue order or timing.  Same arithmetic, same order: its results are the reference
//       bits the product schedule is compared with under memory load.
//   2 = IN-LAYER (only with -DHEDIT_LINCHAIN_INLAYER, tools/chain_hazard.sh): round 3's dropped variant, the row DMA
//       carried inside the layers two pieces per iteration with per-iteration windows.
template <int NPOST, bool GNIN, bool VT, bool SSG, int SCHED>
__global__ __launch_bounds__(256, 1) void lin_chain_kernel(LinKernelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 31, hi = lane >> 5;
  constexpr int NL = 1 + NPOST, NIT = NL * LKS;
  const int ntiles = (p.M + BLOCK_ROWS - 1) / BLOCK_ROWS;
  float* const par = reinterpret_cast<float*>(smem + PAR_OFF);      // gamma | beta | bias_pre

#if defined(__HIP_DEVICE_COMPILE__)
  // inputs through bounds-checked descriptors: rows beyond M read as zeros
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.stream), (short)0, (int)(NIT * IT_BYTES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.a), (short)0, (int)((long)p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(GNIN ? p.a : p.r1), (short)0,
                                                                        (int)((long)p.M * (GNIN ? p.lda : p.ldr1) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GNIN ? p.gn_ss : p.gamma), (short)0,
                                                                         GNIN ? (int)((p.M / (GNIN ? p.rows_per_image : 1)) * LC * 8) : 16, 0x00020000);
#endif
  // weight DMA: piece k (0..2) of this wave for stream iteration `sit` -> ring bank `bank`: 1 KB piece q = wave + 4 k of the
  // iteration's ten (q = 10, 11 re-load piece 9: same bytes, same place)
  const int q2 = wave + 8 > 9 ? 9 : wave + 8;
  const unsigned dma_voff = (unsigned)(lane * 16);
  auto dma_piece = [&](int sit, int bank, int k) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int q = k < 2 ? wave + 4 * k : q2;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + bank * IT_BYTES + q * 1024), 16, dma_voff,
                                             sit * IT_BYTES + q * 1024, 0, 0);
#else
    (void)sit; (void)bank; (void)k;
#endif
  };
  // ---- the tile I/O goes through the wave's staging area as 20 pieces of 64 consecutive 16-byte slots; slot s of row r
  // holds chunk s ^ swz(r) (the XOR stays inside a group of eight chunks = one 128-byte line).  Per piece i this lane's
  // slot is (row pr[i], byte pc[i] inside the row): the global offset of a piece is row * ld + pc, for loads and stores.
  char* const stage = smem + STAGE_OFF + wave * STAGE_BYTES;
  constexpr int ROW_DMAS = 32 * CPR / 64;
  static_assert(32 * CPR % 64 == 0, "rows must divide evenly over the lanes");
  int prc[ROW_DMAS];                 // row << 16 | byte inside the row
#pragma unroll
  for (int i = 0; i < ROW_DMAS; ++i) {
    const int s = i * 64 + lane;
    const int r = s / CPR;
    prc[i] = (r << 16) | (((s - r * CPR) ^ swz(r)) * 16);
  }
  auto piece_off = [&](int i, int ld2) __attribute__((always_inline)) { return (unsigned)((prc[i] >> 16) * ld2 + (prc[i] & 0xffff)); };
  // (the row stride as an opaque value per use: otherwise the compiler computes the 20 offsets of every load / store site
  //  once, in front of the tile loop, and keeps ~100 registers' worth of them in scratch)
  auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); return v; };
  // this lane's own row in the staging area: the eight 16-byte slots of a 128-byte line in swizzled order (+ the lane
  // half's 8 bytes: accumulator layout), and the four slot pairs the fragment reads cycle through
  int slot8[8], slotx[4];
#pragma unroll
  for (int k = 0; k < 8; ++k) slot8[k] = STAGE_OFF + wave * STAGE_BYTES + lm * ROW_BYTES + ((k ^ swz(lm)) * 16) + hi * 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) slotx[k] = STAGE_OFF + wave * STAGE_BYTES + lm * ROW_BYTES + (((2 * k + hi) ^ swz(lm)) * 16);

  // piece i of the wave's 32 rows of a tile (res: the residual tensor instead of the input tensor)
  auto row_piece = [&](auto res_c, int i, int ld2, int soff) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(decltype(res_c)::value ? rs_r : rs_a, (__attribute__((address_space(3))) void*)(stage + i * 1024), 16,
                                             piece_off(i, ld2), soff, 0, 0);
#else
    (void)i; (void)ld2; (void)soff;
#endif
  };
  auto stage_rows = [&](auto res_c, long ld, int tile) __attribute__((always_inline)) {
    const int soff = (int)((long)(tile * BLOCK_ROWS + wave * 32) * ld * 2);
    const int ld2 = opaque((int)ld * 2);
#pragma unroll
    for (int i = 0; i < ROW_DMAS; ++i) row_piece(res_c, i, ld2, soff);
  };
  // the (scale, shift) pairs of the tile's image -> SS_OFF (every wave fetches the same 2.5 KB: identical bytes)
  auto ss_piece = [&](int k, int tile) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int img = (tile * BLOCK_ROWS) / p.rows_per_image;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_ss, (__attribute__((address_space(3))) void*)(smem + SS_OFF + k * 1024), 16, dma_voff,
                                             img * (LC * 8) + k * 1024, 0, 0);
#else
    (void)k; (void)tile;
#endif
  };
  auto stage_ss = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < SS_DMAS; ++k) ss_piece(k, tile);
  };
  // O[nb][4 q + j] of lane (lm, hi) = feature 32 nb + 8 q + 4 hi + j of row lm  (32x32 MFMA result layout)
  f32x16 O[LNB];
  bf16x8 Xn[LKS];        // lane (row lm, half hi): activation fragment of k-step ks (8 consecutive k, or 8 registers of O)
  // (see ffn.hip) the accumulators become readable by the VALU behind asm MFMAs
  auto settle = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) asm volatile("" : "+a"(O[nb]));
  };
  // the first layer's fragments from the staged input rows (GNIN: x * scale + shift on the way)
  // (scale, shift): from the staged pairs of the tile's image (SSG false), or -- images that are not whole tiles, e.g.
  // 24 x 24 tokens -- per row from global memory (`rtile`: the tile the staged rows belong to)
  auto read_xn = [&](int rtile) __attribute__((always_inline)) {
    const float* ssg = nullptr;
    if constexpr (GNIN && SSG) {
      int row = rtile * BLOCK_ROWS + wave * 32 + lm;
      if (row > p.M - 1) row = p.M - 1;
      ssg = p.gn_ss + ((long)(row / p.rows_per_image) * LC + hi * 8) * 2;
    }
#pragma unroll
    for (int ks = 0; ks < LKS; ++ks) {
      u32x4 raw = *reinterpret_cast<const u32x4*>(smem + slotx[ks & 3] + (ks >> 2) * 128);
      if constexpr (GNIN) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[2 * e] = st_lo(raw[e]);
          x[2 * e + 1] = st_hi(raw[e]);
        }
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          f32x4 c;                                                                                             // channels 2 e2, 2 e2 + 1
          if constexpr (!SSG) c = *reinterpret_cast<const f32x4*>(smem + SS_OFF + (ks * 16 + hi * 8 + 2 * e2) * 8);
          else c = *reinterpret_cast<const f32x4*>(ssg + ks * 32 + e2 * 4);
          x[2 * e2] = __builtin_fmaf(x[2 * e2], c[0], c[1]);
          x[2 * e2 + 1] = __builtin_fmaf(x[2 * e2 + 1], c[2], c[3]);
        }
        raw = (u32x4){pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])};
      }
      Xn[ks] = __builtin_bit_cast(bf16x8, raw);
      asm volatile("" : "+a"(Xn[ks]));       // home in the AGPR file from here on (every use is an MFMA operand)
      if constexpr (SSG) __builtin_amdgcn_sched_barrier(0);      // (the global loads of one k-step at a time: register pressure)
    }
    asm volatile("s_nop 3" ::: "memory");    // (VALU-written fragments in front of asm MFMAs)
  };

  // ---- one iteration = ten bundles of [MFMA | a fragment read LEAD MFMAs ahead (the last LEAD fetch the first fragments
  // of the next iteration; not across layers) | three of them a DMA piece], pinned by sched_barrier.  The first iteration
  // of a layer starts the accumulators from the inline constant 0 (nobody writes 160 zeros).  At the iteration
  // boundary: [my pieces of iteration it + 2 have landed: vmcnt(the newest AHEAD - 2 iterations, + the burst in front of
  // the layer while it is younger than that)] [my reads of the iteration just finished have returned: lgkmcnt(the LEAD
  // newest = next iteration's)] barrier; the finished bank is refilled next.
  int sit = 0, bank = 0;                     // stream iteration being consumed (mod NIT), its ring bank
  int frag_rd = lane * 16;
  bf16x8 pre[LEAD];
  // ROWS (SCHED 2 only): row DMA carried by this layer, two pieces per iteration from its first iteration on (0: none;
  // 1: the residual rows of this tile; 2: the input rows (and scale / shift pairs) of tile `rtile`)
  auto linear_layer = [&](auto extra_c, auto rows_c, int rtile) __attribute__((always_inline)) {
    constexpr int EXTRA = decltype(extra_c)::value;
    constexpr int ROWS = SCHED == 2 ? decltype(rows_c)::value : 0;
    constexpr int NP = ROWS == 0 ? 0 : ROW_DMAS + (ROWS == 2 && GNIN ? SS_DMAS : 0);      // pieces carried
    constexpr int PITER = (NP + 1) / 2;                                                    // iterations that carry pieces
    static_assert(PITER + 4 <= LKS, "the carried pieces must be out four iterations before the layer ends");
    static_assert((AHEAD - 2) * PPW + EXTRA + (ROWS ? 8 : 0) <= 63, "vmcnt is a 6-bit counter");
    const int r_ld2 = opaque((int)(ROWS == 1 ? p.ldr1 : p.lda) * 2);
    const int r_soff = (int)((long)(rtile * BLOCK_ROWS + wave * 32) * (ROWS == 1 ? p.ldr1 : p.lda) * 2);
    (void)r_ld2; (void)r_soff;
#pragma unroll
    for (int j = 0; j < LEAD; ++j) pre[j] = *reinterpret_cast<const bf16x8*>(smem + frag_rd + j * 1024);
    static_for<LKS>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      const int base = frag_rd;
      const int nbank = bank == RING - 1 ? 0 : bank + 1;
      const int pbank = bank == 0 ? RING - 1 : bank - 1;
      const int nbase = lane * 16 + nbank * IT_BYTES;
      int dit = sit + AHEAD;
      if (dit >= NIT) dit -= NIT;
      bf16x8 fr[LNB + LEAD];
#pragma unroll
      for (int j = 0; j < LEAD; ++j) fr[j] = pre[j];
      static_for<LNB>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        // lgkmcnt only: fragments b .. b+2 are here when at most the reads of b+3 .. b+LEAD-1 are outstanding (the last
        // iteration of a layer issues none beyond its own fragments)
        if constexpr (b % 3 == 0) {
          constexpr int newer = (ks + 1 < LKS ? b + LEAD - 1 : (b + LEAD - 1 < LNB - 1 ? b + LEAD - 1 : LNB - 1)) - (b + 2);
          __builtin_amdgcn_s_waitcnt(0xC07F | ((newer > 0 ? newer : 0) << 8));
        }
        if constexpr (ks == 0) mfma_l0(O[b], fr[b], Xn[ks]);
        else mfma_l(O[b], fr[b], Xn[ks]);
        if constexpr (b + LEAD < LNB) fr[b + LEAD] = *reinterpret_cast<const bf16x8*>(smem + base + (b + LEAD) * 1024);
        else if constexpr (ks + 1 < LKS) pre[b + LEAD - LNB] = *reinterpret_cast<const bf16x8*>(smem + nbase + (b + LEAD - LNB) * 1024);
        if constexpr (b % 3 == 1) dma_piece(dit, pbank, b / 3);
        if constexpr (ROWS != 0 && b % 3 == 2 && b / 3 < 2) {
          constexpr int pi = 2 * ks + b / 3;                 // carried piece of this bundle
          if constexpr (pi < NP) {
            if constexpr (pi < ROW_DMAS) {
              if constexpr (ROWS == 1) row_piece(std::true_type{}, pi, r_ld2, r_soff);
              else row_piece(std::false_type{}, pi, r_ld2, r_soff);
            } else {
              ss_piece(pi - ROW_DMAS, rtile);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (SCHED == 1) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        // operations newer than the last piece of iteration ks + 2 (issued during iteration ks - 4): the weight pieces (and
        // carried pieces) of iterations ks-3 .. ks, and the burst in front of the layer while ks - 3 <= 0
        constexpr int carried = [] {
          int n = 0;
          for (int j = ks - 3; j <= ks; ++j)
            if (j >= 0) n += (2 * j + 2 <= NP) ? 2 : (2 * j + 1 <= NP ? 1 : 0);
          return n;
        }();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"((AHEAD - 2) * PPW + carried + (ks < 4 ? EXTRA : 0)), "n"(LEAD) : "memory");
      }
      sit = sit + 1 == NIT ? 0 : sit + 1;
      bank = nbank;
      frag_rd = nbase;
    });
  };
  // the staged result -> whole-row pieces -> 16-byte coalesced stores: ROW_DMAS bounds-checked buffer stores, issued for
  // every tile alike (rows beyond M fall outside the buffer)
  auto flush_rows = [&](bf16_t* dst, long ld, int tile) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(dst, (short)0, (int)((((long)p.M - 1) * ld + LC) * 2), 0x00020000);
    const int soff = (int)((long)(tile * BLOCK_ROWS + wave * 32) * ld * 2);
    const int ld2 = opaque((int)ld * 2);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (a wave reads back only what it wrote itself)
    constexpr int GRP = 5;
#pragma unroll
    for (int i0 = 0; i0 < ROW_DMAS; i0 += GRP) {
      u32x4 ov[GRP];
#pragma unroll
      for (int i = 0; i < GRP; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i0 + i) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < GRP; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_buffer_store_b128(ov[i], rs_o, piece_off(i0 + i, ld2), soff, 0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staging area may be overwritten from here on
  };
  // bf16(O) -> the wave's staging area (swizzled like the inputs) -> stores
  auto store_rows = [&](bf16_t* dst, long ld, int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x2 o;
        o[0] = pack_bf16x2(O[nb][4 * q], O[nb][4 * q + 1]);
        o[1] = pack_bf16x2(O[nb][4 * q + 2], O[nb][4 * q + 3]);
        *reinterpret_cast<u32x2*>(smem + slot8[(nb * 4 + q) & 7] + ((nb * 4 + q) >> 3) * 128) = o;
      }
    flush_rows(dst, ld, tile);
  };
  // the same for the transposed result: [320 features][32 rows] per wave (64 bytes per feature; features permuted inside
  // their groups of eight so that the two lane halves, four features apart, write different bank halves), then per
  // feature 64 contiguous bytes of the [C][M] result as four 16-byte stores
  auto store_rows_t = [&](bf16_t* dst, long ld, int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = nb * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
        const int x = f & 7;
        const int slot = (f & ~7) + (x < 4 ? x : 4 + ((x + 1) & 3));
        *reinterpret_cast<bf16_t*>(stage + slot * 64 + lm * 2) = (bf16_t)(pack_bf16x2(O[nb][r], 0.f) & 0xffffu);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(dst, (short)0, (int)((long)LC * ld * 2), 0x00020000);
    const int soff = (tile * BLOCK_ROWS + wave * 32) * 2;
    const int ld2 = opaque((int)ld * 2);
#endif
    constexpr int GRP = 5;
#pragma unroll
    for (int i0 = 0; i0 < ROW_DMAS; i0 += GRP) {
      u32x4 ov[GRP];
#pragma unroll
      for (int i = 0; i < GRP; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i0 + i) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < GRP; ++i) {
        const int s = (i0 + i) * 64 + lane;
        const int slot = s >> 2, part = s & 3;
        const int y = slot & 7;
        const int f = (slot & ~7) + (y < 4 ? y : 4 + ((y + 3) & 3));
#if defined(__HIP_DEVICE_COMPILE__)
        // (rows beyond M -- last tile of an M that is not a multiple of 128 -- get an offset outside the buffer: the store is
        //  issued like every other one and dropped by the bounds check)
        const bool live = tile * BLOCK_ROWS + wave * 32 + part * 8 < p.M;
        __builtin_amdgcn_raw_buffer_store_b128(ov[i], rs_o, live ? (unsigned)(f * ld2 + part * 16) : 0x7ffffff0u, soff, 0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- once: the LayerNorm parameters and the first bias into LDS; the first tile's input rows; the stream's head
  for (int i = tid; i < 3 * LC; i += 256) par[i] = i < LC ? p.gamma[i] : (i < 2 * LC ? p.beta[i - LC] : p.bias_pre[i - 2 * LC]);
  int tile = blockIdx.x;
  stage_rows(std::false_type{}, p.lda, tile);
  if constexpr (GNIN) stage_ss(tile);
#pragma unroll
  for (int it = 0; it < AHEAD; ++it)
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_piece(it, it, k);
  if constexpr (SCHED == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * PPW) : "memory");   // my input rows (and the scale / shift pairs) are in LDS
  read_xn(tile);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr bool INL = SCHED == 2;
  constexpr int E_FIRST = INL ? ROW_DMAS : ROW_DMAS + (GNIN ? 0 : ROW_DMAS);                // in front of a first layer: the final stores (+ the residual DMA)
  constexpr int E_MID = ROW_DMAS;                                                            // in front of a middle layer: the stores of the previous result
  constexpr int E_LAST = INL ? ROW_DMAS : ROW_DMAS + ROW_DMAS + (GNIN ? SS_DMAS : 0);       // in front of the last layer: stores + the next tile's input
  if constexpr (!GNIN && !INL) stage_rows(std::true_type{}, p.ldr1, tile);
  // everything issued so far has landed before anybody passes the first barrier: par[], the input rows, and iterations
  // 0 .. AHEAD-1 of the stream in full (vmcnt(0), NOT a counted wait: see the kernel comment)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  while (true) {
    // ================================================================ first layer
    linear_layer(std::integral_constant<int, E_FIRST>{}, std::integral_constant<int, GNIN ? 0 : 1>{}, tile);
    settle();
    // ONE pass over the accumulator: + bias (+ the residual rows from the staging area), the result goes to the staging
    // area (same slot the residual came from: each slot belongs to one lane) and on to HBM, its LayerNorm becomes the
    // next layers' fragments.  Statistics in one sweep (sum and sum of squares, four partial sums each: fp32 over 320
    // values of order one).
    {
      float X[LNB][16];
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < LNB; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int adr = slot8[(nb * 4 + q) & 7] + ((nb * 4 + q) >> 3) * 128;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(par + 2 * LC + nb * 32 + q * 8 + hi * 4);
          float r4[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (!GNIN) {
            const u32x2 u = *reinterpret_cast<const u32x2*>(smem + adr);
            r4[0] = st_lo(u[0]);
            r4[1] = st_hi(u[0]);
            r4[2] = st_lo(u[1]);
            r4[3] = st_hi(u[1]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = (O[nb][4 * q + j] + bb[j]) + r4[j];
            X[nb][4 * q + j] = x;
            s1[j] += x;
            s2[j] = __builtin_fmaf(x, x, s2[j]);
          }
          u32x2 o;
          o[0] = pack_bf16x2(X[nb][4 * q], X[nb][4 * q + 1]);
          o[1] = pack_bf16x2(X[nb][4 * q + 2], X[nb][4 * q + 3]);
          *reinterpret_cast<u32x2*>(smem + adr) = o;
          if (q == 3) __builtin_amdgcn_sched_barrier(0);       // (keeps the live set of this pass to X and one block's operands)
        }
      float s = (s1[0] + s1[1]) + (s1[2] + s1[3]), q = (s2[0] + s2[1]) + (s2[2] + s2[3]);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      const float mean = s / (float)LC;
      float var = q / (float)LC - mean * mean;
      var = var < 0.f ? 0.f : var;
      const float rstd = rsqrtf(var + p.eps);
      const float shift = -mean * rstd;
#pragma unroll
      for (int ks = 0; ks < LKS; ++ks) {
        float o[8];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int qq = 2 * (ks % 2) + h2;
          const f32x4 gg = *reinterpret_cast<const f32x4*>(par + (ks / 2) * 32 + qq * 8 + hi * 4);
          const f32x4 bb = *reinterpret_cast<const f32x4*>(par + LC + (ks / 2) * 32 + qq * 8 + hi * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[4 * h2 + j] = __builtin_fmaf(__builtin_fmaf(X[ks / 2][4 * qq + j], rstd, shift), gg[j], bb[j]);
        }
        const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        Xn[ks] = __builtin_bit_cast(bf16x8, pk);
        asm volatile("" : "+a"(Xn[ks]));
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_nop 3" ::: "memory");
    }
    flush_rows(p.out_mid, p.ldmid, tile);

    // ================================================================ the layers behind the LayerNorm
#pragma unroll 1
    for (int j = 0; j < NPOST - 1; ++j) {
      linear_layer(std::integral_constant<int, E_MID>{}, std::integral_constant<int, 0>{}, tile);
      settle();
      store_rows(p.out_p[j], p.ldp[j], tile);
    }
    // the next tile's input rows travel while the last layer runs (behind the last tile: the same rows once more, so that
    // every tile puts the same number of operations into the queue)
    const int ntile = tile + (int)gridDim.x;
    const int ltile = ntile < ntiles ? ntile : tile;
    if constexpr (!INL) {
      stage_rows(std::false_type{}, p.lda, ltile);
      if constexpr (GNIN) stage_ss(ltile);
    }
    linear_layer(std::integral_constant<int, E_LAST>{}, std::integral_constant<int, 2>{}, ltile);
    settle();
    read_xn(ltile);                                        // the fragments of the NEXT tile's first layer (Xn is free now)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (VT) store_rows_t(p.out, p.ldo, tile);
    else store_rows(p.out, p.ldo, tile);
    if constexpr (!GNIN && !INL) stage_rows(std::true_type{}, p.ldr1, ltile);
    if (ntile >= ntiles) break;
    tile = ntile;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (DMA still in flight lands in LDS that must still be this block's)
}

template <int NPOST, bool GNIN, bool VT, bool SSG, int SCHED>
int launch_impl(const LinKernelParams& k, hipStream_t st) {
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&lin_chain_kernel<NPOST, GNIN, VT, SSG, SCHED>), LDS_TOTAL)) return rc;
  int cus = 0;
  if (int rc = hedit_cu_count(&cus)) return rc;
  const int ntiles = cdiv(k.M, BLOCK_ROWS);
  hipLaunchKernelGGL((lin_chain_kernel<NPOST, GNIN, VT, SSG, SCHED>), dim3(ntiles < cus ? ntiles : cus), dim3(256), LDS_TOTAL, st, k);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

template <int NPOST, bool GNIN, bool VT, bool SSG>
int launch(const LinKernelParams& k, int sched, hipStream_t st) {
  if (sched == 0) return launch_impl<NPOST, GNIN, VT, SSG, 0>(k, st);
  if (sched == 1) return launch_impl<NPOST, GNIN, VT, SSG, 1>(k, st);
#if defined(HEDIT_LINCHAIN_INLAYER)
  if constexpr (!SSG)
    if (sched == 2) return launch_impl<NPOST, GNIN, VT, SSG, 2>(k, st);
#endif
  hedit_set_error("lin_chain: unknown schedule (2 = in-layer row DMA exists only in the -DHEDIT_LINCHAIN_INLAYER test build)");
  return HEDIT_ERR_ARG;
}

}  // namespace

size_t lin_chain_stream_bytes(int layers) { return (size_t)stream_iters(layers) * IT_BYTES; }

int lin_chain_pack_launch(const float* w, int layer, float scale, int layers, bf16_t* stream, hipStream_t st) {
  ARG_CHECK(w && stream && (layers == 2 || layers == 4) && layer >= 0 && layer < layers, "lin_chain_pack: args");
  const long total = (long)stream_iters(layers) * (IT_BYTES / 16);
  hipLaunchKernelGGL(lin_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, layer, scale, layers, stream);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// one launch: every row tensor inside the 2 GiB window of a buffer descriptor's 32-bit offsets
static int lin_chain_launch_one(const LinChainParams& c, int sched, hipStream_t st) {
  ARG_CHECK((long)c.M * c.lda * 2 < (1L << 31) && (long)c.M * c.ldmid * 2 < (1L << 31) && (c.gn_ss || (long)c.M * c.ldo * 2 < (1L << 31)),
            "lin_chain: tensor beyond the 2 GB buffer window");
  LinKernelParams k{};
  k.a = c.a; k.lda = c.lda; k.bias_pre = c.bias_pre; k.gamma = c.gamma; k.beta = c.beta; k.eps = c.eps;
  k.stream = c.stream; k.M = c.M; k.out_mid = c.out_mid; k.ldmid = c.ldmid; k.out = c.out; k.ldo = c.ldo;
  if (!c.gn_ss) {
    ARG_CHECK((long)c.M * c.ldr1 * 2 < (1L << 31), "lin_chain: residual rows beyond the 2 GB buffer window");
    k.r1 = c.r1; k.ldr1 = c.ldr1;
    return launch<1, false, false, false>(k, sched, st);
  }
  ARG_CHECK((long)c.M * c.ldq * 2 < (1L << 31) && (long)c.M * c.ldk * 2 < (1L << 31) && (long)LC * c.ldo * 2 < (1L << 31),
            "lin_chain: output beyond the 2 GB buffer window");
  k.gn_ss = c.gn_ss; k.rows_per_image = c.rows_per_image;
  k.out_p[0] = c.out_q; k.ldp[0] = c.ldq; k.out_p[1] = c.out_k; k.ldp[1] = c.ldk;
  // images of whole 128-row tiles: the (scale, shift) pairs of a tile are staged once; else they are read per row
  return c.rows_per_image % BLOCK_ROWS == 0 ? launch<3, true, true, false>(k, sched, st) : launch<3, true, true, true>(k, sched, st);
}

int lin_chain_launch(const LinChainParams& c, hipStream_t st) { return lin_chain_launch_sched(c, 0, st); }

int lin_chain_launch_sched(const LinChainParams& c, int sched, hipStream_t st) {
  ARG_CHECK(c.C == LC, "lin_chain: exists for C = 320");
  ARG_CHECK(c.M > 0 && c.a && c.stream && c.gamma && c.beta && c.bias_pre && c.out_mid && c.out, "lin_chain: null");
  ARG_CHECK(c.lda % 8 == 0 && c.ldmid % 8 == 0 && c.ldo % 8 == 0, "lin_chain: rows must be 16-byte aligned");
  if (!c.gn_ss) ARG_CHECK(c.r1 && c.ldr1 % 8 == 0, "lin_chain: residual rows");
  else
    ARG_CHECK(c.rows_per_image > 0 && c.M % c.rows_per_image == 0 && c.M % 8 == 0 && c.out_q && c.out_k && c.ldq % 8 == 0 && c.ldk % 8 == 0,
              "lin_chain: GroupNorm'd input form (whole images, M % 8 == 0; q, k, v^T outputs)");
  // Rows are independent and a launch addresses its tensors through 32-bit buffer offsets, so a batch whose widest row
  // tensor passes 2 GiB (M >= 1.67 M rows of 320 dense channels: > 400 rows of 64 x 64 tokens) runs as several launches
  // over row ranges -- whole tiles, and whole images where the GroupNorm pairs are per image.  Same bits as one launch.
  long ldmax = c.lda > c.ldmid ? c.lda : c.ldmid;
  if (!c.gn_ss) { if (c.ldo > ldmax) ldmax = c.ldo; if (c.ldr1 > ldmax) ldmax = c.ldr1; }
  else { if (c.ldq > ldmax) ldmax = c.ldq; if (c.ldk > ldmax) ldmax = c.ldk; }
  long cap = ((1L << 31) - 1) / (ldmax * 2);
  if (c.M <= cap) return lin_chain_launch_one(c, sched, st);
  long unit = BLOCK_ROWS;
  if (c.gn_ss) {      // lcm(128, rows_per_image): ranges start on an image boundary AND a tile boundary
    long a = unit, b = c.rows_per_image;
    while (b) { const long t = a % b; a = b; b = t; }
    unit = unit / a * c.rows_per_image;
  }
  cap = cap / unit * unit;
  ARG_CHECK(cap > 0, "lin_chain: one image's rows do not fit the 2 GB buffer window");
  for (long r0 = 0; r0 < c.M; r0 += cap) {
    LinChainParams p = c;
    p.M = (int)(c.M - r0 < cap ? c.M - r0 : cap);
    p.a = c.a + r0 * c.lda;
    p.out_mid = c.out_mid + r0 * c.ldmid;
    if (!c.gn_ss) {
      p.r1 = c.r1 + r0 * c.ldr1;
      p.out = c.out + r0 * c.ldo;
    } else {
      p.gn_ss = c.gn_ss + (r0 / c.rows_per_image) * LC * 2;
      p.out_q = c.out_q + r0 * c.ldq;
      p.out_k = c.out_k + r0 * c.ldk;
      p.out = c.out + r0;                       // v^T [C][ldo]: a column range
    }
    if (int rc = lin_chain_launch_one(p, sched, st)) return rc;
  }
  return HEDIT_OK;
}
