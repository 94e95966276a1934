// Implicit-GEMM on MFMA for every dense contraction of the SD UNet: linear / 1x1-conv layers and
// 3x3 convolutions (stride 1, stride 2, and 3x3 on a 2x nearest-upsampled input) over NHWC bf16
// activations.   C[M][N] = A[M][K] * W[N][K]^T (+bias[n]) (+residual[m][n]),  fp32 accumulate.
//
// gfx950 mapping
//   * v_mfma_f32_16x16x32_bf16; the WEIGHT fragment is fed as the MFMA "A" operand and the
//     activation fragment as "B", so a lane ends up holding 4 consecutive output channels n of
//     one row m -> 8-byte bf16x4 / 16-byte f32x4 epilogue stores along the contiguous NHWC axis.
//   * two block shapes: 256 threads = 4 waves (2 x 2), tile 128 x BN x 64, two LDS stages, two
//     blocks per CU; and 512 threads = 8 waves (4 x 2), tile 256 x BN x 64, three LDS stages, one
//     block per CU (long K loops and the FF1+GEGLU launches, see launch_igemm).  BN = 128 or 160
//     (every SD-1.x channel count is a multiple of 160, so no n-tile is wasted).
//   * operands go global -> LDS by DMA (global_load_lds, 16 B per lane, no VGPR round trip and no
//     ds_write).  A DMA instruction fills 8 rows x 128 B lane-linearly, so the bank swizzle is
//     applied on the SOURCE side: slot p of row r receives chunk p ^ (r & 7), the same involution
//     the ds_read_b128 fragment reads use (conflict-free).  Per-lane source pointers carry the conv
//     gather: centre pixel + wave-uniform tap delta, a 9-bit tap mask and a zero page for the
//     border / ragged rows -- no exec-mask branches.
//   * the K loop is pipelined per 32-deep k-step: fragment reads of the next step, the pointer
//     arithmetic of a later tile's DMA and its issue are all woven between the MFMAs of the
//     current step (sched_group_barrier), one barrier per K-tile in mid-tile.  4-wave kernel: DMA
//     one tile ahead, drained at the barrier.  8-wave kernel: ring of three stages, DMA two tiles
//     ahead, counted vmcnt -- the queue never drains.
//   * split-K (grid.y) for the low-resolution, weight-heavy layers: fp32 partial slabs + a
//     reduce/epilogue kernel.
//   * optional epilogue extras, each a template parameter so that the plain instantiations stay
//     byte-identical: GNS = GroupNorm pair statistics of the stored tile (gnstat.h; 128-column tile),
//     OPB = bfloat16 operands in the half-storage build (the split-bf16 GEMMs of pnet.hip).
#include <type_traits>

#include "common.h"
#include "gnstat.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // (not HIP's uint4 struct: aggregate
                                                               //  copies become memcpy and defeat SROA)
constexpr int BM0 = 128;   // row tile of the 4-wave kernel (also used by the host heuristics)
constexpr int BK = 64;

// BM = 128: 256 threads (2x2 waves), 2 LDS stages, 2 blocks per CU.
// BM = 256: 512 threads (4x2 waves), 3 LDS stages (LDS-DMA only), 1 block per CU: 28 % less
//           operand traffic per MFMA and two K-tiles in flight.
// DEEP (BM = 128 only): the three-stage ring of the 256-row tile for the 128-row tile, ONE block per CU -- for launches with at most
//           one block per CU, where the two-stage loop has nothing to overlap its DMA round trip with (a K-tile is 0.17 us of MFMA
//           work behind a ~1 us request): two K-tiles in flight instead of one.
template <int BM, int BN, bool DEEP = false>
struct Smem {
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int W_BYTES = BN * BK * 2;
  static constexpr int STAGE = A_BYTES + W_BYTES;
  static constexpr int STAGES = (DEEP || (BM == 256 && BN <= 160)) ? 3 : 2;
  static constexpr int EPI = BM * (BN + 8) * 2;          // the epilogue stages the output tile in the same LDS
  static constexpr int TOTAL = STAGES * STAGE > EPI ? STAGES * STAGE : EPI;
  // row-sharing 3x3 conv (kernel mode 4): two activation tiles + a ring of three weight tiles + one zero row per
  // activation tile, placed at the same distance behind each of them
  static constexpr int RS_W0 = 2 * A_BYTES;
  static constexpr int RS_ZERO = RS_W0 + 3 * W_BYTES;               // zero row of activation tile 0 (tile 1: + A_BYTES)
  static constexpr int RS_TOTAL = RS_ZERO + A_BYTES + 128;
};

// CHUNK: the in-block form of the canonical K-chunking (see gemm_canonical_chunk): the fp32 sum is formed
// chunk by chunk -- every p.chunk_kt K-tiles the running accumulators are added to a second register set
// and restart from zero -- which is bit for bit what the split-K path computes (one chunk per slab, slabs
// added in order by splitk_reduce_kernel).  The result of a layer therefore does not depend on whether a
// launch was large enough to skip the split, i.e. not on the batch.
// DRAIN (tests/test_gpu_ring_hazard.py only): every counted wait of the operand rings becomes vmcnt(0) -- same arithmetic in
// the same order, but nothing in LDS is read while any DMA of the wave is in flight.  Its output is the reference the
// product schedule (DRAIN = false) must reproduce bit for bit under memory load (the protocol that found round 3's race).
// GNS (128-column tile): the epilogue also writes the GroupNorm pair statistics of the stored tile (gnstat.h) to p.gn_part.  A
// template parameter, not a run-time branch: the instantiations without it are the kernels of the SD loop, untouched.
// OPB (half-storage build only, HEDIT_STORE_F16): the operands are bfloat16 whatever the storage format -- the [hi | hi | lo]
// triples of the split-bf16 GEMMs (pnet.hip), which only ever take the fp32 slab / raw-product exit of the epilogue.
template <bool OPB>
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
#if HEDIT_F16
  if constexpr (OPB)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(true_bf16x8, a), __builtin_bit_cast(true_bf16x8, b), c, 0, 0, 0);
#endif
  return MFMA_16x16x32_ST(a, b, c, 0, 0, 0);
}

template <int BM, int BN, int MODE, bool CHUNK, bool DRAIN = false, bool DEEP = false, bool GNS = false, bool OPB = false>
__global__ __launch_bounds__(BM * 2, 2) void igemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = BM * 2;       // threads: 4 or 8 waves, each owning a 64 x BN/2 sub-tile
  constexpr int NW = NT / 64;
  constexpr int NI = BN / 32;      // 16-wide n sub-tiles per wave
  constexpr int MI = 4;            // 16-wide m sub-tiles per wave
  constexpr int A_CH = BM * 8 / NT;                    // 16-byte chunks (or 8-row DMA groups) per thread / wave
  constexpr int W_GROUPS = BN / 8;
  constexpr int W_CH = (W_GROUPS + NW - 1) / NW;       // 8 waves x 3 > 20 groups: the surplus re-loads the last group
  using S = Smem<BM, BN, DEEP>;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): LDS-DMA targets, tile roles
  const int wm = wave >> 1, wn = wave & 1;      // (BM/64) x 2 waves

  // XCD-aware tile order.  Workgroup b is observed to run on XCD b % 8 (speed only, never
  // correctness): give every XCD one contiguous chunk of the tile sequence so that tiles which
  // share an operand panel hit the same 4 MiB L2.  Within the sequence the n-tiles of one m-tile
  // are adjacent when the activation operand is the big one (they re-use its rows), and the
  // m-tiles of one n-tile are adjacent when the weight panel is the big one.
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  int v;
  {
    const int nb = tiles_m * tiles_n, bid = blockIdx.x;
    const int xcd = bid & 7, seq = bid >> 3, q = nb >> 3, r = nb & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
  }
  const int tile_m = p.n_fastest ? v / tiles_n : v % tiles_m;
  const int tile_n = p.n_fastest ? v % tiles_n : v / tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int split = blockIdx.y;
  const int kt_begin = split * p.kt_per_split;
  int kt_end = kt_begin + p.kt_per_split;
  const int kt_total = p.K / BK;
  if (kt_end > kt_total) kt_end = kt_total;

  // ---- per-thread staging state, fixed for the whole K loop.  Everything that can be decided
  // once is decided here so that the K loop issues (almost) nothing but loads, LDS traffic and
  // MFMAs.  Operands are addressed through buffer descriptors (base in SGPRs + a 32-bit byte offset
  // per lane + a wave-uniform scalar offset): the conv gather is "centre pixel offset + uniform
  // tap delta" gated by a 9-bit tap mask, and an out-of-range row / padded tap simply gets an offset
  // beyond the descriptor's size, which the hardware answers with zeros (no exec-mask branches, no
  // 64-bit pointer arithmetic, one address dword per lane).
  constexpr unsigned OOB = 0x80000000u;    // > any operand size accepted by gemm_launch
  const bf16_t* const zero_page = p.zeros; // (the epilogue's residual prefetch still selects a pointer)
  unsigned a_off[A_CH];
  unsigned a_mask[A_CH];     // conv: bit t = tap t is inside the image; linear: ~0 / 0
  int a_oy[A_CH], a_ox[A_CH];
  int a_lds[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    // register staging: thread -> (row, chunk) = (id>>3, id&7), written to the swizzled slot.
    // LDS-DMA staging: one wave instruction fills 8 rows x 128 B linearly (lane L -> row L>>3,
    // slot L&7), so the swizzle is applied on the SOURCE side: slot p of row r must receive
    // chunk p ^ (r & 7)  (same involution the fragment reads use).
    const int row = (wave * A_CH + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (lane >> 3);
    const int m = m0 + row;
    const bool ok = m < p.M;
    a_lds[i] = (wave * A_CH + i) * 1024;
    a_oy[i] = a_ox[i] = 0;
    if (MODE == 0) {
      a_off[i] = (unsigned)(((long)m * p.lda + c * 8) * 2);
      a_mask[i] = ok ? ~0u : 0u;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = m / hw;
      const int r = m - b * hw;
      const int oy = r / p.Wout;
      const int ox = r - oy * p.Wout;
      if (MODE == 3 || MODE == 5) {
        // 3x3 on the 2x nearest-upsampled image: tap (dy,dx) of output pixel (oy,ox) reads input
        // pixel ((oy+dy)>>1, (ox+dx)>>1) = (oy>>1, ox>>1) + (fy, fx) with fy = -1/0 for even oy
        // (dy = -1 / else) and 0/+1 for odd oy (else / dy = +1): the gather is again "centre
        // pointer + small offset", only the offset depends on the pixel's parity bits.
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int uy = oy + t / 3 - 1, ux = ox + t % 3 - 1;
          const unsigned in = (unsigned)ok & (unsigned)((unsigned)uy < (unsigned)(2 * p.Hin)) & (unsigned)((unsigned)ux < (unsigned)(2 * p.Win));
          mk |= in << t;
        }
        a_mask[i] = mk;
        a_off[i] = (unsigned)(((((long)b * p.Hin + (oy >> 1)) * p.Win + (ox >> 1)) * p.Cin + c * 8) * 2);
        a_oy[i] = oy & 1;
        a_ox[i] = ox & 1;
      } else {
        const int cy = MODE == 2 ? oy * 2 + p.asym : oy, cx = MODE == 2 ? ox * 2 + p.asym : ox;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
          const unsigned in = (unsigned)ok & (unsigned)((unsigned)iy < (unsigned)p.Hin) & (unsigned)((unsigned)ix < (unsigned)p.Win);
          mk |= in << t;
        }
        a_mask[i] = mk;
        a_off[i] = (unsigned)(((((long)b * p.Hin + cy) * p.Win + cx) * p.Cin + c * 8) * 2);
      }
    }
  }
  unsigned w_off[W_CH];      // out-of-range weight rows: OOB once and for all
  int w_lds[W_CH];
#pragma unroll
  for (int i = 0; i < W_CH; ++i) {
    int wg = wave * W_CH + i;
    if (wg > W_GROUPS - 1) wg = W_GROUPS - 1;
    const int row = wg * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (lane >> 3);
    const int n = n0 + row;
    w_lds[i] = wg * 1024;
    w_off[i] = n < p.N ? (unsigned)(((long)n * p.K + c * 8) * 2) : OOB;
  }
#if defined(__HIP_DEVICE_COMPILE__)     // (the resource type only exists in the device pass)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), (short)0, (int)p.w_bytes, 0x00020000);
#endif

  // The DMA of one K-tile in two halves, so each can be woven into a different MFMA group of the K
  // loop: prep_dma computes the lane offsets of the A_CH activation chunks (tap mask -> offset or OOB;
  // the weight offsets never change) and the two scalar offsets, fire_dma issues the A_CH + W_CH
  // LDS-DMA instructions (M0 + buffer_load ... lds each) into stage `buf`.
  // K-tile order.  linear: k0 = kt*64.  conv: the 9 taps of one 64-channel slab are visited back
  // to back (tap = kt % 9, slab = kt / 9) so the shifted re-reads of the same input pixels are
  // nine consecutive K-tiles apart at most -> they stay in L2.
  struct DmaArgs {
    unsigned a_voff[A_CH];
    int a_soff, w_soff;       // wave-uniform, non-negative byte offsets
  };
  auto prep_dma = [&](int kt, DmaArgs& d) __attribute__((always_inline)) {
    int k0 = kt * BK;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) d.a_voff[i] = a_mask[i] ? a_off[i] : OOB;
      d.a_soff = k0 * 2;
    } else {
      const int tap = kt % 9;
      const int ci0 = (kt / 9) * BK;
      k0 = tap * p.Cin + ci0;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      if (MODE == 3 || MODE == 5) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          const int fy = (a_oy[i] + dy) >> 1, fx = (a_ox[i] + dx) >> 1;      // in {-1, 0, +1}
          const bool ok = (a_mask[i] >> tap) & 1u;
          d.a_voff[i] = ok ? a_off[i] + (unsigned)((fy * p.Win + fx) * p.Cin * 2) : OOB;
        }
        d.a_soff = ci0 * 2;
      } else {
        // the tap delta can be negative and the scalar offset of a buffer access is an unsigned
        // addend outside the range check, so the delta goes into the lane offset (wrapping add)
        const unsigned tapd = (unsigned)((dy * p.Win + dx) * p.Cin * 2);
#pragma unroll
        for (int i = 0; i < A_CH; ++i) d.a_voff[i] = ((a_mask[i] >> tap) & 1u) ? a_off[i] + tapd : OOB;
        d.a_soff = ci0 * 2;
      }
    }
    d.w_soff = k0 * 2;
  };
  auto fire_dma = [&](int buf, const DmaArgs& d) __attribute__((always_inline)) {
    char* sa = smem + buf * S::STAGE;
    char* sw = sa + S::A_BYTES;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(sa + a_lds[i]), 16, d.a_voff[i], d.a_soff, 0, 0);
#pragma unroll
    for (int i = 0; i < W_CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(sw + w_lds[i]), 16, w_off[i], d.w_soff, 0, 0);
#else
    (void)sa; (void)sw; (void)d;
#endif
  };
  auto issue_glds = [&](int kt, int buf) __attribute__((always_inline)) {
    DmaArgs d;
    prep_dma(kt, d);
    fire_dma(buf, d);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 tot[CHUNK ? MI : 1][CHUNK ? NI : 1];
  int next_flush = 0x7fffffff;      // tile index (relative to kt_begin) after which the chunk sum is folded
  if constexpr (CHUNK) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) tot[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    next_flush = p.chunk_kt - 1;
  }
  auto flush = [&]() __attribute__((always_inline)) {
    if constexpr (CHUNK) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          tot[i][j] += acc[i][j];
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  };

  const int fr = lane & 15;   // fragment row within a 16-row sub-tile
  const int fq = lane >> 4;   // k-group (8 bf16 each) within a 32-deep MFMA step

  // fragment read offsets: (row & 7) == (fr & 7) for every sub-tile of this lane, and the second
  // 32-deep k-step only flips chunk bit 2 (byte 64), so two base offsets + compile-time
  // immediates (sub-tile i -> +i*2048 bytes) address all 18 ds_read_b128 of a K-tile
  const int rd_x = ((fq ^ (fr & 7)) << 4);
  const int a_rd = (wm * 64 + fr) * 128 + rd_x;
  const int w_rd = (wn * (BN / 2) + fr) * 128 + rd_x;

  if constexpr (MODE == 4 || MODE == 5) {
    // ---- 3x3 stride-1 conv, the three taps of one kernel row share ONE staged activation tile.
    // Output pixel m, tap (dy, dx) reads input pixel m + dy*W + dx: for a tile of 256 consecutive pixels the
    // taps (dy,-1), (dy,0), (dy,+1) read the SAME 256 input rows shifted by one LDS row, so the tile is staged
    // once per dy (the centre-tap gather, rows outside the image zero-filled by the buffer range check) and the
    // fragment reads of the side taps address row +-1.  With W | 256 the tile starts at x = 0: the rows a shift
    // would take from outside the tile (row -1, row 256) belong to pixels whose side tap is outside the image
    // anyway, and those lanes read a zero row instead.  A K-tile then moves 1/3 activation tile + one weight tile
    // from L2 (31 KB instead of 52 KB at BN = 160) with 13 instead of 21 LDS-DMA instructions per wave and three
    // K-tiles; the MFMA chain of every output element is unchanged (same bits as modes 1 / the split-K path).
    // Mode 5 = the same on the 2x nearest-upsampled image (mode 3's gather): the staged tile is a row of the
    // UPSAMPLED image (input pixel ((oy+dy)>>1, ox>>1) for output-resolution pixel (oy, ox)), so the side taps are
    // again one LDS row away and W below is the output width.
    //   LDS: act tile g&1 (32 KB each) | weight ring, stage = tap column (3 x BN x 128 B) | zero rows
    //   DMA: weights three K-tiles ahead (as the ring loop below); the activation tile of tap row g+1 in two halves
    //        behind the weight DMAs of K-tiles (g-1,2) and (g,0), i.e. >= 2 K-tiles before its first use
    constexpr int AP = S::A_BYTES;
    static_assert(BM == 256 && A_CH == 4, "row-sharing conv: 256-row tile");
    constexpr bool UP = MODE == 5;
    const int Wimg = UP ? p.Wout : p.Win;
    const int G = kt_total / 3;                 // tap rows: 3 per 64-channel slab
    if (tid < 64) reinterpret_cast<uint32_t*>(smem + S::RS_ZERO + (tid >> 5) * AP)[tid & 31] = 0u;
    // side-tap read offsets (relative to the activation tile): row -1 / +1, or the zero row at the image edge
    int a_rdl[MI], a_rdr[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * 64 + i * 16 + fr;
      const int x = (m0 + row) % Wimg;
      const int rl = row - 1, rr = row + 1;
      a_rdl[i] = x == 0 ? S::RS_ZERO + (fq << 4) : rl * 128 + ((fq ^ (rl & 7)) << 4);
      a_rdr[i] = x == Wimg - 1 ? S::RS_ZERO + (fq << 4) : rr * 128 + ((fq ^ (rr & 7)) << 4);
    }
    bf16x8 xa[MI], wa[NI], xb[MI], wb[NI];
    auto read_a = [&](int ab, int col, int ks, bf16x8 (&xf)[MI]) __attribute__((always_inline)) {
      if (col == 1) {
        const char* pa = smem + ab + (a_rd ^ (ks << 6));
#pragma unroll
        for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(pa + i * 2048);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          xf[i] = *reinterpret_cast<const bf16x8*>(smem + ab + ((col == 0 ? a_rdl[i] : a_rdr[i]) ^ (ks << 6)));
      }
    };
    auto read_w = [&](int stage, int ks, bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
      const char* pw = smem + S::RS_W0 + stage * S::W_BYTES + (w_rd ^ (ks << 6));
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
    };
    auto mfmas = [&](const bf16x8 (&xf)[MI], const bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<OPB>(wf[j], xf[i], acc[i][j]);
    };
    // half `part` (DMA groups 2*part, 2*part+1 of this wave) of the activation tile of tap row g -> tile g&1
    struct ASrc { unsigned v[2]; int s; };
    // (register diet, mode 4: the pixel offset is linear in the pixel index, so DMA group i of this wave sits 8 pixels
    //  = i * 8 * Cin * 2 bytes behind group 0; the centre-column validity of the three tap rows is 3 bits per group,
    //  the row parity of the upsampling gather one more)
    const unsigned a_off0 = a_off[0];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      a_ok |= (((a_mask[i] >> 1) & 1u) | (((a_mask[i] >> 4) & 1u) << 1) | (((a_mask[i] >> 7) & 1u) << 2)) << (3 * i);
      if (UP) a_ok |= (unsigned)(a_oy[i] & 1) << (12 + i);
    }
    const unsigned grp_step = (unsigned)(8 * p.Cin * 2);
    const int row_bytes = p.Win * p.Cin * 2;
    auto a_prep = [&](int g, int part, ASrc& d) __attribute__((always_inline)) {
      const int slab = g / 3, dyi = g - slab * 3;
      const bool live = g < G;                  // (one dummy half is issued past the end: keeps the vmcnt pattern)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = part * 2 + q;
        const bool ok = live && ((a_ok >> (3 * i + dyi)) & 1u);
        if (UP) {
          const int fy = ((int)((a_ok >> (12 + i)) & 1u) + dyi - 1) >> 1;       // input row delta in {-1, 0, +1}
          d.v[q] = ok ? a_off[i] + (unsigned)(fy * row_bytes) : OOB;
        } else {
          d.v[q] = ok ? a_off0 + (unsigned)i * grp_step + (unsigned)((dyi - 1) * row_bytes) : OOB;
        }
      }
      d.s = slab * BK * 2;
    };
    auto a_fire = [&](int g, int part, const ASrc& d) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
      char* sa = smem + (g & 1) * AP;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(sa + a_lds[part * 2 + q]), 16, d.v[q], d.s, 0, 0);
#else
      (void)g; (void)part; (void)d;
#endif
    };
    auto w_fire = [&](int kt, int stage) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int slab = kt / 9, tap = kt - slab * 9;
      const int soff = (tap * p.Cin + slab * BK) * 2;
      char* sw = smem + S::RS_W0 + stage * S::W_BYTES;
#pragma unroll
      for (int i = 0; i < W_CH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(sw + w_lds[i]), 16, w_off[i], soff, 0, 0);
#else
      (void)kt; (void)stage;
#endif
    };
    // prologue, in the issue order of the steady state: act(0) | w(0) | w(1) | [act(1) first half, w(2)]
    ASrc asrc;
    a_prep(0, 0, asrc); a_fire(0, 0, asrc);
    a_prep(0, 1, asrc); a_fire(0, 1, asrc);
    w_fire(0, 0);
    w_fire(1, 1);
    a_prep(1, 0, asrc); a_fire(1, 0, asrc);
    w_fire(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRAIN ? 0 : (2 * W_CH + 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    read_a(0, 0, 0, xa);
    read_w(0, 0, wa);
    // one K-tile = tap column `col` of tap row g.  fire: the DMA slot behind the mid-tile barrier.
    auto tile = [&](auto col_c, int g, bool steady) __attribute__((always_inline)) {
      constexpr int col = decltype(col_c)::value;
      const int j = g * 3 + col;
      const int ab = (g & 1) * AP;
      read_a(ab, col, 1, xb);
      read_w(col, 1, wb);
      __builtin_amdgcn_s_setprio(1);
      if (steady && col != 1) a_prep(col == 2 ? g + 2 : g + 1, col == 2 ? 0 : 1, asrc);
      mfmas(xa, wa);
#pragma unroll
      for (int r = 0; r < MI * NI / 2; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);      // offsets of this K-tile's DMA slot
      }
      __builtin_amdgcn_s_setprio(0);
      if (steady) {
        // my DMAs of K-tile j+1 have landed: only the slot issued during K-tile j-1 may still be in flight
        if (col == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : (W_CH)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : (W_CH + 2)) : "memory");
      } else if (col == 0) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : (W_CH + 2)) : "memory");
      } else if (col == 1) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      __builtin_amdgcn_s_setprio(1);
      if (col < 2) {
        read_a(ab, col + 1, 0, xa);
        read_w(col + 1, 0, wa);
      } else if (steady) {
        read_a(ab ^ AP, 0, 0, xa);
        read_w(0, 0, wa);
      }
      if (steady) {
        if (col != 1) a_fire(col == 2 ? g + 2 : g + 1, col == 2 ? 0 : 1, asrc);
        w_fire(j + 3, col);
      }
      mfmas(xb, wb);
      if (steady) {
        constexpr int GR = (MI + NI + 1) / 2;                   // MFMA pairs that carry 2 fragment reads each
#pragma unroll
        for (int r = 0; r < MI * NI / 2; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // 2 MFMA
          if (r < GR) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 ds_read
          } else {
            __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);  // M0 + scalar offset
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 LDS-DMA
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (CHUNK) {
        if (j == next_flush && j + 1 < kt_total) { flush(); next_flush += p.chunk_kt; }
      }
    };
    int g = 0;
    for (; g + 1 < G; ++g) {
      tile(std::integral_constant<int, 0>{}, g, true);
      tile(std::integral_constant<int, 1>{}, g, true);
      tile(std::integral_constant<int, 2>{}, g, true);
    }
    tile(std::integral_constant<int, 0>{}, g, false);
    tile(std::integral_constant<int, 1>{}, g, false);
    tile(std::integral_constant<int, 2>{}, g, false);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS is reused by the epilogue
  } else if constexpr (S::STAGES == 3) {
    // Three LDS stages as a ring with the DMA TWO tiles ahead and a counted vmcnt: the queue never
    // drains (one to two tiles = 52-104 KB in flight), which is what the measured ~0.85 us
    // issue-to-landed time of an LDS-DMA under load needs -- the 2-stage loop below can only give a
    // DMA one tile of MFMAs to land and is bound by exactly that latency (operand traffic without
    // MFMAs runs at 1.6-1.9 PF/s-equivalent, MFMAs without traffic at 1.3-1.6, the two together
    // at 0.9-1.15: they do not overlap).  Same k-step-level weave as the 2-stage loop:
    //   group 1 of tile j = MFMAs of k-step 0 | fragment reads of k-step 1 | pointers of tile j+3
    //   mid-tile          = wait [my DMA of tile j+1 landed: vmcnt(NDMA) lets tile j+2 stay in
    //                       flight] + lgkmcnt(0); barrier -> tile j+1 visible everywhere, stage
    //                       j%3 in registers everywhere
    //   group 2 of tile j = MFMAs of k-step 1 | DMA of tile j+3 into stage j%3 | reads of tile
    //                       j+1's k-step 0
    constexpr int NDMA = A_CH + W_CH;
    const int nk = kt_end - kt_begin;
    bf16x8 xa[MI], wa[NI], xb[MI], wb[NI];
    DmaArgs nsrc;
    auto read_frags = [&](int buf, int ks, bf16x8 (&xf)[MI], bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
      const char* pa = smem + buf * S::STAGE + (a_rd ^ (ks << 6));
      const char* pw = smem + buf * S::STAGE + S::A_BYTES + (w_rd ^ (ks << 6));
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(pa + i * 2048);
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
    };
    auto mfmas = [&](const bf16x8 (&xf)[MI], const bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<OPB>(wf[j], xf[i], acc[i][j]);
    };
    for (int t = 0; t < 3 && t < nk; ++t) issue_glds(kt_begin + t, t);
    if (nk > 0) {
      if (nk >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRAIN ? 0 : (2 * NDMA)) : "memory");
      else if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRAIN ? 0 : (NDMA)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      read_frags(0, 0, xa, wa);
    }
    int j = 0, cur = 0;
    for (; j + 3 < nk; ++j) {          // steady state: tiles j+1 .. j+3 exist
      const int nx = cur == 2 ? 0 : cur + 1;
      read_frags(cur, 1, xb, wb);
      __builtin_amdgcn_s_setprio(1);
      prep_dma(kt_begin + j + 3, nsrc);
      mfmas(xa, wa);
#pragma unroll
      for (int r = 0; r < MI * NI / 2; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);      // pointer arithmetic
      }
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : (NDMA)) : "memory");
      __builtin_amdgcn_s_setprio(1);
      read_frags(nx, 0, xa, wa);
      fire_dma(cur, nsrc);
      mfmas(xb, wb);
#pragma unroll
      for (int r = 0; r < MI * NI / 2; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
        if (r < (MI + NI + 1) / 2) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // 2 ds_read
        } else {
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);    // M0
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // 2 LDS-DMA
        }
      }
      __builtin_amdgcn_s_setprio(0);
      cur = nx;
      if constexpr (CHUNK) {
        if (j == next_flush) { flush(); next_flush += p.chunk_kt; }
      }
    }
    for (; j < nk; ++j) {              // last three tiles: nothing left to fetch
      const int nx = cur == 2 ? 0 : cur + 1;
      read_frags(cur, 1, xb, wb);
      __builtin_amdgcn_s_setprio(1);
      mfmas(xa, wa);
      __builtin_amdgcn_s_setprio(0);
      if (j + 1 < nk) {
        if (j + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : (NDMA)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(nx, 0, xa, wa);
      }
      __builtin_amdgcn_s_setprio(1);
      mfmas(xb, wb);
      __builtin_amdgcn_s_setprio(0);
      cur = nx;
      if constexpr (CHUNK) {
        if (j == next_flush && j + 1 < nk) { flush(); next_flush += p.chunk_kt; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS is reused by the epilogue
  } else {
    // two LDS stages, pipelined at the granularity of a 32-deep k-step: the fragments of the NEXT
    // k-step are read from LDS while the 20 MFMAs of the current one run, across the K-tile
    // boundary too -- a wave never sits in a read-only phase.  Tile k+1 must therefore be visible
    // in the middle of tile k: the barrier sits between the two MFMA groups, after it stage `cur`
    // has been read into registers completely (lgkmcnt(0) first) and takes the DMA of tile k+2.
    const int nk = kt_end - kt_begin;
    bf16x8 xa[MI], wa[NI], xb[MI], wb[NI];
    auto read_frags = [&](int buf, int ks, bf16x8 (&xf)[MI], bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
      const char* pa = smem + buf * S::STAGE + (a_rd ^ (ks << 6));
      const char* pw = smem + buf * S::STAGE + S::A_BYTES + (w_rd ^ (ks << 6));
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(pa + i * 2048);
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
    };
    auto mfma_group = [&](const bf16x8 (&xf)[MI], const bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<OPB>(wf[j], xf[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    };
    // Steady state of one K-tile i (two MFMA groups of MI*NI, one barrier between them):
    //   group 1 = MFMAs of k-step 0, with the fragment reads of k-step 1 in front and the source
    //             pointers of tile i+2 (prep_dma: one select per activation chunk) woven between the MFMAs;
    //   barrier  = tile i+1 has landed everywhere, stage `cur` is in registers everywhere;
    //   group 2 = MFMAs of k-step 1, with the 9 LDS-DMA issues of tile i+2 (into stage cur) and the
    //             fragment reads of tile i+1's k-step 0 woven in.
    // The ~130 non-MFMA instructions a tile's DMA needs thereby sit in issue slots the 16-cycle
    // MFMAs leave free instead of forming a serial phase (they were ~45 % of a wave's K-tile time).
    // sched_group_barrier spells the interleave out for the scheduler.
    DmaArgs nsrc;
    auto weave_prep = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < MI * NI / 2; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);      // 8 VALU | SALU (pointer arithmetic)
      }
    };
    // (the fragment reads come first: the compiler cannot prove that the LDS-DMA writes into stage
    // `cur` do not alias the reads of stage `cur ^ 1`, so reads and DMA issues cannot alternate)
    auto weave_fire = [&]() __attribute__((always_inline)) {
      constexpr int G = MI * NI / 2, GR = (MI + NI + 1) / 2;     // MFMA pairs; pairs that carry 2 reads each
#pragma unroll
      for (int r = 0; r < G; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
        if (r < GR) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // 2 ds_read
        } else {
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);    // M0
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // 2 LDS-DMA
        }
      }
    };
    auto mfmas = [&](const bf16x8 (&xf)[MI], const bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<OPB>(wf[j], xf[i], acc[i][j]);
    };
    if (nk > 0) {
      issue_glds(kt_begin, 0);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (nk > 1) issue_glds(kt_begin + 1, 1);
      read_frags(0, 0, xa, wa);
    }
    int i = 0;
    for (; i + 2 < nk; ++i) {          // steady state: tile i+2 exists, no conditionals inside
      const int cur = i & 1;
      read_frags(cur, 1, xb, wb);
      __builtin_amdgcn_s_setprio(1);
      prep_dma(kt_begin + i + 2, nsrc);
      mfmas(xa, wa);
      weave_prep();
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
      read_frags(cur ^ 1, 0, xa, wa);
      fire_dma(cur, nsrc);
      mfmas(xb, wb);
      weave_fire();
      __builtin_amdgcn_s_setprio(0);
      if constexpr (CHUNK) {
        if (i == next_flush) { flush(); next_flush += p.chunk_kt; }
      }
    }
    for (; i < nk; ++i) {              // last two tiles: nothing left to fetch
      const int cur = i & 1;
      read_frags(cur, 1, xb, wb);
      mfma_group(xa, wa);
      if (i + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(cur ^ 1, 0, xa, wa);
      }
      mfma_group(xb, wb);
      if constexpr (CHUNK) {
        if (i == next_flush && i + 1 < nk) { flush(); next_flush += p.chunk_kt; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS is reused by the epilogue
  }
  if constexpr (CHUNK) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = tot[i][j] + acc[i][j];
  }

  // ---- epilogue.  A lane holds 4 consecutive n of column m = mb + fr.
  if (p.partial) {
    // split-K: fp32 slab, 16-byte stores
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + fq * 4;
        if (n >= p.N) continue;
        float* dst = p.partial + ((long)split * p.M + m) * p.N + n;
        *reinterpret_cast<f32x4*>(dst) = acc[i][j];
      }
    }
    return;
  }
  if (p.geglu) {
    // FF1 with the GEGLU fused: weight rows were interleaved at load time so that sub-tiles
    // (j, j+1) of a wave are (value, gate) of the same 16 output columns:
    // out[m][c] = (v + bv) * gelu(g + bg).  The block writes a 128 x BN/2 tile.
    if constexpr ((NI & 1) == 0) {
      constexpr int ON = BN / 2, CSG = ON + 8;
      bf16_t* sg = reinterpret_cast<bf16_t*>(smem);
      // the bias of every column pair is requested up front, all at once (the fragment registers are free now): inside
      // the loop each pair paid its own dependent global-load round trip, 16 of them in a row on the 256-column tile
      f32x4 bv[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int nb = n0 + wn * (BN / 2) + j * 16 + fq * 4;
        bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias && nb < p.N) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + nb);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ml = wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          const f32x4 v = acc[i][j] + bv[j], g = acc[i][j + 1] + bv[j + 1];      // (no bias: + 0)
          const int ol = wn * (ON / 2) + (j / 2) * 16 + fq * 4;      // output column inside the tile
          uint2 o;
          const f32x2 r0 = mul_gelu2((f32x2){v[0], v[1]}, (f32x2){g[0], g[1]});
          const f32x2 r1 = mul_gelu2((f32x2){v[2], v[3]}, (f32x2){g[2], g[3]});
          o.x = pack_bf16x2(r0[0], r0[1]);
          o.y = pack_bf16x2(r1[0], r1[1]);
          *reinterpret_cast<uint2*>(sg + ml * CSG + ol) = o;
        }
      }
      __syncthreads();
      constexpr int GCH = ON / 8;
      const int on0 = n0 / 2;
      // all LDS reads of a thread first, then its stores: as a rolled loop every 16-byte piece waited for its own LDS
      // round trip before the store could issue
      constexpr int GIT = BM * GCH / NT;
      static_assert(BM * GCH % NT == 0, "geglu tile rows must divide evenly over the block");
      u32x4 og[GIT];
#pragma unroll
      for (int it = 0; it < GIT; ++it) {
        const int idx = tid + it * NT;
        const int ml = idx / GCH, c = idx - ml * GCH;
        og[it] = *reinterpret_cast<const u32x4*>(sg + ml * CSG + c * 8);
      }
#pragma unroll
      for (int it = 0; it < GIT; ++it) {
        const int idx = tid + it * NT;
        const int ml = idx / GCH, c = idx - ml * GCH;
        const int m = m0 + ml, n = on0 + c * 8;
        if (m < p.M && n < p.N / 2) *reinterpret_cast<u32x4*>(p.C + (long)m * p.ldc + n) = og[it];
      }
    }
    return;
  }
  // bf16 output: stage the BM x BN tile in LDS (the K loop is over, its buffers are free) and
  // write it out as whole rows, 16 B per lane, so a wave store instruction covers >= 1 KiB of
  // contiguous NHWC memory instead of sixteen 32-byte fragments.  The residual rows and the bias
  // are requested FIRST, all at once, so their latency hides under the staging pass instead of
  // costing one dependent round trip per 16-byte piece.
  constexpr int CS = BN + 8;                  // padded row stride (elements): 16-B aligned rows
  constexpr int CHUNKS = BN / 8;              // 16-byte chunks per tile row
  constexpr int ITER = BM * CHUNKS / NT;
  static_assert(BM * CHUNKS % NT == 0, "tile rows must divide evenly over the block");
  bf16_t* sc = reinterpret_cast<bf16_t*>(smem);
  static_assert(BM * CS * 2 <= S::TOTAL, "epilogue tile must fit the staging LDS");
  const bool aligned8 = (p.N % 8 == 0) && (p.ldc % 8 == 0) && (p.residual == nullptr || p.ldr % 8 == 0);

  u32x4 resid[ITER];
  if (aligned8 && p.residual) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * NT;
      const int ml = idx / CHUNKS, c = idx - ml * CHUNKS;
      const int m = m0 + ml, n = n0 + c * 8;
      const bf16_t* src = (m < p.M && n < p.N) ? p.residual + (long)m * p.ldr + n : zero_page;
      resid[it] = *reinterpret_cast<const u32x4*>(src);
    }
  }
  f32x4 biasv[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 16 + fq * 4;
    biasv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) biasv[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ml = wm * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nl = wn * (BN / 2) + j * 16 + fq * 4;
      const f32x4 v = acc[i][j] + biasv[j];
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(sc + ml * CS + nl) = o;
    }
  }
  __syncthreads();
  auto add2 = [](uint32_t a, uint32_t b) __attribute__((always_inline)) {
    return pack_bf16x2(bf16_to_f32((bf16_t)(a & 0xffff)) + bf16_to_f32((bf16_t)(b & 0xffff)),
                       bf16_to_f32((bf16_t)(a >> 16)) + bf16_to_f32((bf16_t)(b >> 16)));
  };
  if (aligned8) {
    // all LDS reads of a thread first (every piece is inside the staged tile), then the residual adds, then the
    // stores under their range checks: with the check in front, each piece waited for its own LDS round trip
    u32x4 ov[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * NT;
      const int ml = idx / CHUNKS, c = idx - ml * CHUNKS;
      ov[it] = *reinterpret_cast<const u32x4*>(sc + ml * CS + c * 8);
    }
    if (p.residual) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        ov[it][0] = add2(ov[it][0], resid[it][0]);
        ov[it][1] = add2(ov[it][1], resid[it][1]);
        ov[it][2] = add2(ov[it][2], resid[it][2]);
        ov[it][3] = add2(ov[it][3], resid[it][3]);
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * NT;
      const int ml = idx / CHUNKS, c = idx - ml * CHUNKS;
      const int m = m0 + ml, n = n0 + c * 8;
      if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.C + (long)m * p.ldc + n) = ov[it];
    }
    if constexpr (GNS) {
      // GroupNorm pair statistics of the tile as stored (ov: after the residual add), in the canonical tree of gnstat.h:
      // thread (r, c) holds piece c of rows r + RSTEP * it -- exactly the store mapping above for CHUNKS = 16
      static_assert(BN == 128 && CHUNKS == 16 && ITER == 8, "statistics epilogue: 128-column tile");
      constexpr int RSTEP = NT / CHUNKS;                    // 16 (128-row tile) or 32 (256-row tile)
      constexpr int UNITS = BM / GNS_UNIT;
      constexpr int RED_OFF = BM * CS * 2;                  // behind the staged tile (other threads may still be reading it)
      static_assert(RED_OFF % 16 == 0 && RED_OFF + UNITS * GNS_RED_BYTES <= S::TOTAL, "statistics scratch must fit the LDS");
      float2* red = reinterpret_cast<float2*>(smem + RED_OFF);
      const int r = tid / CHUNKS, c = tid - r * CHUNKS;
      GnPiece t[2];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const GnPiece g = gns_piece(ov[it][0], ov[it][1], ov[it][2], ov[it][3]);
        if (gns_first<RSTEP>(it)) t[gns_slot<RSTEP>(it)] = g; else gns_add(t[gns_slot<RSTEP>(it)], g);
      }
      gns_store_t(red, gns_red_row<RSTEP>(r, 0), c, t[0]);
      gns_store_t(red, gns_red_row<RSTEP>(r, 1), c, t[1]);
      __syncthreads();
      if (tid < UNITS * 64) {
        const int u = tid >> 6, pk = tid & 63;
        const float2 v = gns_fold_unit(red, u, pk);
        if (m0 + u * GNS_UNIT < p.M)         // (M % 128 == 0: a unit is whole or absent)
          *reinterpret_cast<float2*>(p.gn_part + (((long)(m0 / GNS_UNIT + u) * (p.N / 2)) + n0 / 2 + pk) * 2) = v;
      }
    }
  } else {
    // ragged right edge (N % 8 == 4) or rows that are only 8-byte aligned: 8-byte pieces
    for (int idx = tid; idx < BM * CHUNKS * 2; idx += NT) {
      const int ml = idx / (CHUNKS * 2), h = idx - ml * (CHUNKS * 2);
      const int m = m0 + ml, n = n0 + h * 4;
      if (m >= p.M || n >= p.N) continue;
      uint2 o = *reinterpret_cast<const uint2*>(sc + ml * CS + h * 4);
      if (p.residual) {
        const uint2 r = *reinterpret_cast<const uint2*>(p.residual + (long)m * p.ldr + n);
        o.x = add2(o.x, r.x);
        o.y = add2(o.y, r.y);
      }
      *reinterpret_cast<uint2*>(p.C + (long)m * p.ldc + n) = o;
    }
  }
}

// sum the split-K slabs IN ORDER (0 + p0 + p1 + ...: the chain the CHUNK kernel forms in registers) and apply the
// same epilogue arithmetic as igemm_kernel: bf16(sum + bias), then + residual in fp32 and rounded again.
// One thread per 4 consecutive n.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  const long total = (long)p.M * n4;
  if (idx >= total) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (long)m * n4) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.splits; ++s)
    v += *reinterpret_cast<const f32x4*>(p.partial + ((long)s * p.M + m) * p.N + n);
  if (p.raw_f32) {      // plain fp32 products (precise-GEMM path of pnet.hip): no bias / residual / rounding
    *reinterpret_cast<f32x4*>(p.raw_f32 + (long)m * p.N + n) = v;
    return;
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + n);
  v = v + b;
  uint2 o;
  o.x = pack_bf16x2(v[0], v[1]);
  o.y = pack_bf16x2(v[2], v[3]);
  if (p.residual) {
    const uint2 r = *reinterpret_cast<const uint2*>(p.residual + (long)m * p.ldr + n);
    o.x = pack_bf16x2(bf16_to_f32((bf16_t)(o.x & 0xffff)) + bf16_to_f32((bf16_t)(r.x & 0xffff)),
                      bf16_to_f32((bf16_t)(o.x >> 16)) + bf16_to_f32((bf16_t)(r.x >> 16)));
    o.y = pack_bf16x2(bf16_to_f32((bf16_t)(o.y & 0xffff)) + bf16_to_f32((bf16_t)(r.y & 0xffff)),
                      bf16_to_f32((bf16_t)(o.y >> 16)) + bf16_to_f32((bf16_t)(r.y >> 16)));
  }
  *reinterpret_cast<uint2*>(p.C + (long)m * p.ldc + n) = o;
}

// The split-K reduce with the GroupNorm pair statistics (gnstat.h): a block owns one 128-row unit x 128 columns in the piece
// layout of the 128-row igemm tile (thread (r, c): piece c of rows r + 16 it), every element the arithmetic of
// splitk_reduce_kernel -- same output bits, same statistics bits as a launch that folds its chunks in registers.
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(GemmParams p) {
  __shared__ float2 red[32 * 64];
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const int m0 = blockIdx.x * GNS_UNIT, n = blockIdx.y * 128 + c * 8;
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (p.bias) {
    b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
    b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
  }
  auto add2 = [](uint32_t a, uint32_t b) __attribute__((always_inline)) {
    return pack_bf16x2(bf16_to_f32((bf16_t)(a & 0xffff)) + bf16_to_f32((bf16_t)(b & 0xffff)),
                       bf16_to_f32((bf16_t)(a >> 16)) + bf16_to_f32((bf16_t)(b >> 16)));
  };
  GnPiece t[2];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int m = m0 + r + 16 * it;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    for (int s = 0; s < p.splits; ++s) {
      const float* src = p.partial + ((long)s * p.M + m) * p.N + n;
      v0 += *reinterpret_cast<const f32x4*>(src);
      v1 += *reinterpret_cast<const f32x4*>(src + 4);
    }
    v0 = v0 + b0;
    v1 = v1 + b1;
    u32x4 o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
    if (p.residual) {
      const u32x4 rr = *reinterpret_cast<const u32x4*>(p.residual + (long)m * p.ldr + n);
      o[0] = add2(o[0], rr[0]); o[1] = add2(o[1], rr[1]); o[2] = add2(o[2], rr[2]); o[3] = add2(o[3], rr[3]);
    }
    *reinterpret_cast<u32x4*>(p.C + (long)m * p.ldc + n) = o;
    const GnPiece g = gns_piece(o[0], o[1], o[2], o[3]);
    if (gns_first<16>(it)) t[gns_slot<16>(it)] = g; else gns_add(t[gns_slot<16>(it)], g);
  }
  gns_store_t(red, gns_red_row<16>(r, 0), c, t[0]);
  gns_store_t(red, gns_red_row<16>(r, 1), c, t[1]);
  __syncthreads();
  if (tid < 64) {
    const float2 v = gns_fold_unit(red, 0, tid);
    *reinterpret_cast<float2*>(p.gn_part + ((long)blockIdx.x * (p.N / 2) + blockIdx.y * 64 + tid) * 2) = v;
  }
}

template <int BM, int BN, int MODE, bool CHUNK, bool DEEP = false>
int launch_igemm_impl(const GemmParams& p, int splits, hipStream_t st) {
  using S = Smem<BM, BN, DEEP>;
  constexpr int LDS = (MODE == 4 || MODE == 5) ? S::RS_TOTAL : S::TOTAL;
  dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), splits);
  // the statistics epilogue: conv modes on the 128-column tile, launches that write the bf16 tile themselves (a split-K launch
  // leaves them to splitk_reduce_gn_kernel).  No drained twin: the K loop is the one of the GNS = false instantiation.
#if HEDIT_F16
  if (p.op_bf16) {      // (gemm_launch has checked that the launch leaves through the fp32 exit)
    if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, CHUNK, false, DEEP, false, true>), LDS)) return rc;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE, CHUNK, false, DEEP, false, true>), grid, dim3(BM * 2), LDS, st, p);
    LAUNCH_CHECK();
    return HEDIT_OK;
  }
#endif
  if constexpr (BN == 128 && MODE != 0) {
    if (p.gn_part && !p.partial) {
      if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, CHUNK, false, DEEP, true>), LDS)) return rc;
      hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE, CHUNK, false, DEEP, true>), grid, dim3(BM * 2), LDS, st, p);
      LAUNCH_CHECK();
      return HEDIT_OK;
    }
  }
  // (only the loops with counted waits have a drained twin: the two-stage loop waits vmcnt(0) as it is)
  constexpr bool COUNTED = MODE == 4 || MODE == 5 || S::STAGES == 3;
  if constexpr (COUNTED) {
    if (hedit_test_drained()) {
      if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, CHUNK, true, DEEP>), LDS)) return rc;
      hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE, CHUNK, true, DEEP>), grid, dim3(BM * 2), LDS, st, p);
      LAUNCH_CHECK();
      return HEDIT_OK;
    }
  }
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, CHUNK, false, DEEP>), LDS)) return rc;
  hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE, CHUNK, false, DEEP>), grid, dim3(BM * 2), LDS, st, p);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// Tile choice.  The 8-wave 256-row kernel (three-stage ring, DMA two tiles ahead) is 5-8 % faster
// than the 4-wave 128-row kernel when the K loop is long and there is at least one tile per CU
// (3x3 convs, FF2: 1.17-1.26 vs 1.10-1.16 PF/s) and for the fused FF1+GEGLU at every K (+2 / +9 /
// +14 % at K = 320 / 640 / 1280: half as many tiles pay the GELU epilogue's LDS pass); plain
// launches with few K-tiles lose to its longer per-tile prologue / epilogue (K = 320: -3...-9 %).
// (The tile shape never changes a result: every output element is the same k-ordered MFMA chain.)

template <int BN, int MODE>
int launch_igemm(const GemmParams& p, int splits, hipStream_t st) {
  const long tiles256 = (long)cdiv(p.M, 256) * cdiv(p.N, BN);
  const bool big = splits == 1 && (p.geglu || p.K / BK >= 16) && tiles256 >= 200;
  const bool chunk = splits == 1 && p.chunk_kt > 0 && p.chunk_kt < p.K / BK;
  if constexpr (MODE == 1) {
    // stride-1 3x3 on the 256-row tile with an image width that divides it: the row-sharing loop (kernel mode 4)
    if (big && splits == 1 && p.Hout == p.Hin && p.Wout == p.Win && p.Win > 0 && 256 % p.Win == 0) {
      if (!chunk) return launch_igemm_impl<256, BN, 4, false>(p, splits, st);
      // (with the second accumulator set of the chunk fold the 160-column tile spills in this loop, also with a single
      //  set of weight fragments refilled column by column: 128 columns only)
      if constexpr (BN == 128) return launch_igemm_impl<256, BN, 4, true>(p, splits, st);
    }
  }
  if constexpr (MODE == 3) {
    if (big && splits == 1 && p.Wout > 0 && 256 % p.Wout == 0) {
      if (!chunk) return launch_igemm_impl<256, BN, 5, false>(p, splits, st);
      if constexpr (BN == 128) return launch_igemm_impl<256, BN, 5, true>(p, splits, st);
    }
  }
  if (chunk)
    return big ? launch_igemm_impl<256, BN, MODE, true>(p, splits, st) : launch_igemm_impl<128, BN, MODE, true>(p, splits, st);
  if (big) return launch_igemm_impl<256, BN, MODE, false>(p, splits, st);
  // a launch with at most one 128-row block per CU: the three-stage ring (Smem DEEP).  A choice by the launch's size, like the tile
  // shape -- every output element is the same k-ordered MFMA chain either way
  int cus = 256;
  if (int rc = hedit_cu_count(&cus)) return rc;
  const long blocks = (long)cdiv(p.M, 128) * cdiv(p.N, BN) * splits;
  if (blocks <= cus && p.K / BK >= 3) return launch_igemm_impl<128, BN, MODE, false, true>(p, splits, st);
  return launch_igemm_impl<128, BN, MODE, false>(p, splits, st);
}

}  // namespace

int gemm_pick_bn(int N);
static int pick_bn(const GemmParams& p) {
  // FF1 + GEGLU: 256 x 256 tile (8 waves, 64 x 128 each, two LDS stages), +13...18 % on the FF1 shapes over 256 x 128
  // (23 % less operand traffic per MFMA) when it still fills the chip
  if (p.geglu) return (p.N % 256 == 0 && (long)cdiv(p.M, 256) * (p.N / 256) >= 200) ? 256 : 128;
  return gemm_pick_bn(p.N);
}

int gemm_pick_bn(int N) {
  // smallest padded width wins; ties go to the wider tile
  long w160 = (long)cdiv(N, 160) * 160, w128 = (long)cdiv(N, 128) * 128;
  return w160 <= w128 ? 160 : 128;
}

// Canonical K-chunking of a contraction, a function of its NOMINAL shape only (the caller passes the per-image extent
// times GEMM_NOMINAL_BATCH on whichever of M / N carries the batch -- never the actual batch): the fp32 result
// is DEFINED as  ((0 + p_0) + p_1) + ...  with p_c the MFMA chain over the K-tiles [c * chunk, (c+1) * chunk).
// Small launches run one chunk per split-K slab (parallelism for a handful of images), large ones fold the
// chunks in registers (CHUNK kernel, no slab traffic) -- same bits either way, so a row's result does not
// depend on how many other rows share the launch.  Returns the chunk length in K-tiles, 0 = one plain chain.
int gemm_canonical_chunk(int M_nom, int N_nom, int K) {
  const int bn = gemm_pick_bn(N_nom);
  const long tiles = (long)cdiv(M_nom, BM0) * cdiv(N_nom, bn);
  const int kt = K / BK;
  // (contractions shorter than 24 K-tiles stay one chain: splitting K <= 1472 buys a handful of rows less than the reduce launch
  //  and the fold's second accumulator set cost -- one image 9.46 -> 9.23 ms per UNet call, 120 rows -0.3 %, 20 rows -0.5 % against
  //  a threshold of 8; 48 gives the one-image gain back: gpurun_out/r05/kt12.txt)
  if (tiles >= 256 || kt < 24) return 0;
  int s = (int)((256 + tiles - 1) / tiles);     // one block per CU at the nominal batch
  int max_s = kt / 4;                            // keep >= 4 K-tiles per chunk
  if (s > max_s) s = max_s;
  if (s > 16) s = 16;
  if (s <= 1) return 0;
  const int chunk = cdiv(kt, s);
  return chunk >= kt ? 0 : chunk;
}

// How a launch of the ACTUAL shape executes that chunking: the number of split-K slabs (1 = in one launch, chunks
// folded in registers).  Speed only -- both forms produce the same bits.
int gemm_plan_splits(int M, int N, int K, int chunk_kt) {
  const int kt = K / BK;
  if (chunk_kt <= 0 || chunk_kt >= kt) return 1;
  const int s = cdiv(kt, chunk_kt);
  const long tiles = (long)cdiv(M, BM0) * cdiv(N, gemm_pick_bn(N));
  return tiles * s <= 1024 ? s : 1;
}

// (single-kernel C entry points: an explicit split count, no canonical structure)
int gemm_pick_splits(int M, int N, int K, int force) {
  (void)M; (void)N;
  const int kt = K / BK;
  if (force <= 1) return 1;
  return force > kt ? kt : force;
}

size_t gemm_partial_bytes(int M, int N, int splits) {
  return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

static const bf16_t* zero_page_device() {
  static bf16_t* z = nullptr;
  if (!z) {
    void* ptr = nullptr;
    if (hipMalloc(&ptr, 256) != hipSuccess) return nullptr;
    if (hipMemset(ptr, 0, 256) != hipSuccess) return nullptr;
    z = reinterpret_cast<bf16_t*>(ptr);
  }
  return z;
}

int gemm_prepare() { return zero_page_device() ? HEDIT_OK : HEDIT_ERR_HIP; }

int gemm_launch(GemmParams p, int splits, float* partial_ws, hipStream_t st) {
  p.zeros = zero_page_device();
  if (!p.zeros) {
    hedit_set_error("gemm: could not allocate the zero page");
    return HEDIT_ERR_HIP;
  }
  ARG_CHECK(p.K % BK == 0, "gemm: K must be a multiple of 64");
  ARG_CHECK(p.N % 4 == 0, "gemm: N must be a multiple of 4");
  ARG_CHECK(p.mode >= 0 && p.mode <= 3, "gemm: mode");
  if (p.mode != 0) ARG_CHECK(p.Cin % BK == 0 && p.K == 9 * p.Cin, "gemm: conv needs Cin % 64 == 0 and K = 9 Cin");
  ARG_CHECK(p.ldc % 4 == 0 && (p.residual == nullptr || p.ldr % 4 == 0), "gemm: ldc/ldr alignment");
  {
    // operand extents for the buffer descriptors; lane offsets are 32-bit and offsets >= 2^31 mean
    // "zero fill", so an operand must stay below 2 GiB (the largest here: 80 rows x 4096 x 1280 x 2 B
    // = 0.84 GB).  Split a launch over M if that is ever exceeded.
    const double a_b = p.mode == 0 ? (double)p.M * p.lda * 2.0
                                   : (double)(p.M / ((long)p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin * 2.0;
    const double w_b = (double)p.N * p.K * 2.0;
    ARG_CHECK(a_b < 2040.0 * 1048576.0 && w_b < 2040.0 * 1048576.0, "gemm: operand larger than 2 GiB");
    p.a_bytes = (unsigned)a_b;
    p.w_bytes = (unsigned)w_b;
  }
  const int kt = p.K / BK;
  if (splits < 1) splits = 1;
  if (splits > kt) splits = kt;
  if (p.chunk_kt < 0 || p.chunk_kt >= kt) p.chunk_kt = 0;
  if (splits > 1) {
    // split-K executes the canonical chunking when there is one (one chunk per slab), else an even cut
    p.kt_per_split = p.chunk_kt > 0 ? p.chunk_kt : cdiv(kt, splits);
    p.splits = cdiv(kt, p.kt_per_split);
    ARG_CHECK(p.chunk_kt == 0 || p.splits == splits, "gemm: split count does not match the canonical chunking");
  } else {
    p.splits = 1;
    p.kt_per_split = kt;
  }
  splits = p.splits;
  if (p.raw_f32 && splits == 1) {
    ARG_CHECK(!p.geglu, "gemm: raw fp32 output excludes the GEGLU epilogue");
    p.partial = p.raw_f32;      // the split-K slab path with one split IS the fp32 product
  } else if (splits > 1) {
    ARG_CHECK(!p.geglu, "gemm: split-K excludes the GEGLU epilogue");
    ARG_CHECK(partial_ws != nullptr, "gemm: split-K needs a partial workspace");
    p.partial = partial_ws;
  } else {
    p.partial = nullptr;
  }
  ARG_CHECK(!p.op_bf16 || (p.raw_f32 && !p.geglu && !p.gn_part), "gemm: bfloat16-by-contract operands come with the raw fp32 output");
  if (p.gn_part) {
    ARG_CHECK(p.mode != 0 && !p.geglu && !p.raw_f32 && !p.ln_out, "gemm: pair statistics come with the bf16 tile of a convolution");
    ARG_CHECK(p.M % GNS_UNIT == 0 && p.N % 128 == 0 && gemm_pick_bn(p.N) == 128, "gemm: pair statistics need M % 128 == 0 and the 128-column tile");
    ARG_CHECK(p.ldc % 8 == 0 && (p.residual == nullptr || p.ldr % 8 == 0), "gemm: pair statistics need 16-byte rows");
  }
  int bn = pick_bn(p);
  // the in-register chunk fold doubles the accumulators: with the upsampling gather's extra lane state the
  // 160-column tile would spill inside the K loop, so that one combination takes the 128-column tile
  // (the tile shape never changes a result)
  if (splits == 1 && p.chunk_kt > 0 && p.mode == 3) bn = 128;
  // stride-1 3x3 with the chunk fold: the row-sharing loop exists for the 128-column tile only (+6 % over the
  // 160-column tile of the plain loop when the narrower tiles still fill the chip)
  if (splits == 1 && p.chunk_kt > 0 && p.mode == 1 && p.Hout == p.Hin && p.Wout == p.Win && p.Win > 0 && 256 % p.Win == 0 &&
      p.N % 128 == 0 && (long)cdiv(p.M, 256) * (p.N / 128) >= 512)
    bn = 128;
  if (p.geglu) {
    ARG_CHECK(p.N % 32 == 0 && p.mode == 0 && p.residual == nullptr && p.ldc % 8 == 0, "gemm: geglu epilogue needs N % 32 == 0, linear mode, no residual");
    ARG_CHECK(splits == 1 && p.chunk_kt == 0, "gemm: the geglu epilogue takes no K-chunking");
  }
  {
    // unique operand bytes: activations M x (K or Cin), weights N x K
    const double a_bytes = (double)p.M * (p.mode == 0 ? p.K : p.Cin);
    const double w_bytes = (double)p.N * p.K;
    p.n_fastest = a_bytes >= w_bytes ? 1 : 0;
  }
  if (pgemm_supported(p, splits, bn)) return pgemm_launch(p, bn, st);      // persistent ring across tiles (pgemm.hip): same bits
  if (pconv_supported(p, splits, bn)) return pconv_launch(p, bn, st);      // the same for the row-sharing 3x3 loop (pconv.hip)
  int rc;
#define DISPATCH(BNV)                                                     \
  switch (p.mode) {                                                       \
    case 0: rc = launch_igemm<BNV, 0>(p, splits, st); break;              \
    case 1: rc = launch_igemm<BNV, 1>(p, splits, st); break;              \
    case 2: rc = launch_igemm<BNV, 2>(p, splits, st); break;              \
    default: rc = launch_igemm<BNV, 3>(p, splits, st); break;             \
  }
  if (bn == 256) {
    rc = launch_igemm_impl<256, 256, 0, false>(p, 1, st);     // FF1 + GEGLU only (no K-chunking there: 242 VGPRs, the fold would not fit)
  } else if (bn == 160) { DISPATCH(160) } else { DISPATCH(128) }
#undef DISPATCH
  if (rc != HEDIT_OK) return rc;
  if (splits > 1) {
    if (p.ln_out && p.ln_done && !p.raw_f32 && p.ldc == p.N && splitk_reduce_ln_supported(p.N)) {
      *p.ln_done = 1;
      return splitk_reduce_ln_launch(p.partial, p.splits, p.bias, p.residual, p.ldr, p.C, p.ln_out, p.ln_gamma, p.ln_beta, p.M, p.N,
                                     p.ln_eps, st);
    }
    if (p.gn_part) {
      hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3(p.M / GNS_UNIT, p.N / 128), dim3(256), 0, st, p);
      LAUNCH_CHECK();
      return HEDIT_OK;
    }
    long total = (long)p.M * (p.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p);
    LAUNCH_CHECK();
  }
  return HEDIT_OK;
}
