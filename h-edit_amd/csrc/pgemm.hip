// Persistent form of igemm_kernel for the linear / 1x1 layers (mode 0): the SAME 256 x BN x 64 tile, wave layout, MFMA chain and
// epilogue arithmetic (so the same bits), but a block walks over its tiles and the operand ring never drains between them.
//
// Why (profiles/r05_igemm_ablation.txt D, profiles/HISTORY.md): the K = 640 / 1280 projections ran at 536-760 TFLOP/s because every
// tile pays (a) a prologue that waits one HBM round trip with nothing in flight behind it, (b) a K loop whose look-ahead starts from
// empty, and (c) an epilogue (block barrier, LDS staging of the whole tile, stores) during which the CU requests nothing.  Here
//   * one block per CU (8 waves, 4 x 2), each walking the tiles  X0 + c, X0 + c + nbx, ...  of its XCD's chunk of the XCD-aware tile
//     order (the blocks of one XCD work on adjacent tiles at any time, as the one-shot launch does);
//   * the three-stage LDS-DMA ring runs over the FLAT sequence of K-tiles of all of the block's tiles: in K-tile j of a tile the DMA of
//     K-tile j + 3 is issued -- which, for the last three, belongs to the NEXT tile (its row / column offsets are computed a tile
//     ahead).  Two K-tiles (64 KB of activations) are in flight at every moment, also across the epilogue;
//   * the epilogue needs no block barrier and no LDS of its own: after the mid-tile barrier of a tile's last K-tile that K-tile's stage
//     is dead everywhere, and each wave stages its 64 x BN/2 sub-tile, 16 rows at a time, through the 4 KB of that stage that its OWN
//     next DMA will overwrite (a wave-private scratch: LDS instructions of one wave execute in order, so write -> read -> write needs
//     no wait), reads it back as 16-byte row pieces, adds the residual and stores.  Only then does the wave issue the deferred DMA of
//     that stage (K-tile 2 of the next tile) and walk into the next tile's K loop -- the only synchronisation between waves remains the
//     one barrier per K-tile;
//   * bias, residual and output go through buffer descriptors: a masked lane has an out-of-range offset instead of a cleared exec
//     bit, so every wave issues EXACTLY the same number of vector-memory instructions per tile -- which is what lets the ring keep
//     counted vmcnt waits with the epilogue's loads and stores in the queue (gfx950 retires vmcnt in issue order, loads and stores
//     alike: profiles/r04_vmcnt_order.txt).  Issue order per wave and tile, and the waits it implies:
//         K-tile j (0 <= j < nk-1), second half : DMA(f+3)                              [NDMA instructions]
//         K-tile nk-1, after its mid barrier    : bias loads [NI], residual loads [RL]
//         epilogue                              : stores [ST], the NEXT tile's bias row [1 DMA, see below], then the deferred DMA(f+3)
//       mid-tile wait of K-tile j >= 1 : the DMA of K-tile j+1 must have landed, younger than it is only the DMA of K-tile j+2
//                                        -> vmcnt(NDMA)
//       mid-tile wait of K-tile 0      : the DMA of K-tile 1 was issued in the previous tile's K-tile nk-2; younger are that tile's
//                                        bias / residual loads (consumed, hence retired), its ST stores and the deferred DMA
//                                        -> vmcnt(ST + 1 + NDMA)    (first tile: the prologue has already waited for K-tile 1)
//   * the bias row of a tile does not come from global memory when the epilogue needs it (a ~1 us round trip with the matrix pipe idle:
//     all eight waves reach the epilogue together): each wave copies its half row (BN/2 floats) by ONE LDS-DMA into a slot of its own
//     behind the ring -- for the first tile in the prologue, for every later tile right behind the previous epilogue's stores -- and the
//     epilogue reads it with ds_read_b128.  (The residual rows are requested under the last MFMA group and consumed after the first
//     pass's staging.)
//     DRAIN twin (tests/test_gpu_ring_hazard.py): every one of these is vmcnt(0).
// Not here (they keep igemm_kernel): the in-register chunk fold (two accumulator sets leave no room for the residual rows), split-K,
// GEGLU, conv modes, rows that are not 16-byte aligned.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int BK = 64;
constexpr int BM = 256;
constexpr int NT = 512;
constexpr unsigned OOB = 0x80000000u;
// measurement builds only (tools/build_variant.sh ... pgemm "-DPG_ABL=n"; results are WRONG by construction, only the time is used):
// bit 0 stores dropped (out-of-range offsets), bit 1 activation offsets wrapped into 256 KB (every A request an L2 hit),
// bit 2 no operand traffic (every DMA out of range = zero fill), bit 3 no epilogue
#ifndef PG_ABL
#define PG_ABL 0
#endif
constexpr int ABL = PG_ABL;

template <int BN>
struct PS {
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int W_BYTES = BN * BK * 2;
  static constexpr int STAGE = A_BYTES + W_BYTES;
  static constexpr int BIAS0 = 3 * STAGE;                   // 8 wave-private slots of BN/2 floats: the tile's bias row halves
  static constexpr int BIAS_SLOT = BN * 2;
  static constexpr int TOTAL = BIAS0 + 8 * BIAS_SLOT;
  static constexpr int PCH = BN / 16;                       // 16-byte pieces per row of a wave's sub-tile
  static constexpr int NQ = (16 * PCH + 63) / 64;           // piece instructions per 16-row pass
  static constexpr int SCR_STRIDE = BN + 16;                // bytes: BN/2 elements + 16 (bank spread, see the header)
  static_assert(16 * SCR_STRIDE <= 4096 && (NQ * 64 / PCH + 1) * SCR_STRIDE <= 4096, "scratch stays inside the wave's own DMA target");
  static_assert(TOTAL <= 160 * 1024, "ring must fit the LDS");
};

template <int BN, bool RES, bool DRAIN>
__global__ __launch_bounds__(NT, 2) void pgemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // for the epilogue's inline-asm LDS accesses
  using S = PS<BN>;
  constexpr int NI = BN / 32, MI = 4;
  constexpr int A_CH = 4;
  constexpr int W_GROUPS = BN / 8;
  constexpr int W_CH = (W_GROUPS + 7) / 8;
  constexpr int NDMA = A_CH + W_CH;
  constexpr int NQ = S::NQ, PCH = S::PCH;
  constexpr int ST = 4 * NQ;                                // stores per wave and tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- the block's tiles
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nb = tiles_m * tiles_n;
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, cidx = bid >> 3;
  const int q8 = nb >> 3, r8 = nb & 7;
  const int X0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int XN = q8 + (xcd < r8 ? 1 : 0);
  const int nbx = (G - xcd + 7) >> 3;                       // blocks of this launch on my XCD
  if (cidx >= XN) return;
  const int T = (XN - cidx + nbx - 1) / nbx;
  const int nk = p.K / BK;                                  // >= 3 (pgemm_supported)

  const unsigned lda2 = (unsigned)p.lda * 2u, ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
  const unsigned K2 = (unsigned)p.K * 2u;
  // DMA source offsets of one tile: activations rows (wave*4+i)*8 + lane/8, weights rows wg*8 + lane/8, 16-byte chunk (lane&7)^(lane>>3)
  // (the bank swizzle is applied on the source side, as in igemm_kernel)
  const unsigned dchunk = (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
  auto tile_mn = [&](int k, int& m0, int& n0) __attribute__((always_inline)) {
    const int v = X0 + cidx + k * nbx;
    const int tm = p.n_fastest ? v / tiles_n : v % tiles_m;
    const int tn = p.n_fastest ? v % tiles_n : v / tiles_m;
    m0 = tm * BM;
    n0 = tn * BN;
  };
  auto tile_offsets = [&](int k, unsigned (&ao)[A_CH], unsigned (&wo)[W_CH]) __attribute__((always_inline)) {
    int m0, n0;
    tile_mn(k, m0, n0);
    const bool live = k < T;                                // past the end: zero-fill requests keep the vmcnt pattern
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int m = m0 + (wave * A_CH + i) * 8 + (lane >> 3);
      ao[i] = (live && m < p.M) ? (unsigned)m * lda2 + dchunk : OOB;
      if (ABL & 2) ao[i] = live ? (ao[i] & 0x3ffffu) : OOB;
      if (ABL & 4) ao[i] = OOB;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      int wg = wave * W_CH + i;
      if (wg > W_GROUPS - 1) wg = W_GROUPS - 1;
      const int n = n0 + wg * 8 + (lane >> 3);
      wo[i] = (live && n < p.N) ? (unsigned)n * K2 + dchunk : OOB;
      if (ABL & 4) wo[i] = OOB;
    }
  };
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), (short)0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)((unsigned)p.M * ldc2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), (short)0, p.bias ? p.N * 4 : 0, 0x00020000);   // no bias: every read is out of range = 0
  const __amdgpu_buffer_rsrc_t rs_r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.residual), (short)0, RES ? (int)((unsigned)p.M * ldr2) : 0, 0x00020000);
#endif
  auto fire = [&](int buf, const unsigned (&ao)[A_CH], const unsigned (&wo)[W_CH], int soff) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    char* sa = smem + buf * S::STAGE;
    char* sw = sa + S::A_BYTES;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(sa + (wave * A_CH + i) * 1024), 16, ao[i], soff, 0, 0);
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      int wg = wave * W_CH + i;
      if (wg > W_GROUPS - 1) wg = W_GROUPS - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(sw + wg * 1024), 16, wo[i], soff, 0, 0);
    }
#else
    (void)buf; (void)ao; (void)wo; (void)soff;
#endif
  };

  // the wave's half of the bias row of the tile at column n0 -> its slot (BN/8 lanes x 16 B; a lane past N requests out of range = 0)
  auto bias_fire = [&](int n0, bool live) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (lane < BN / 8) {
      const int n = n0 + wn * (BN / 2) + lane * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(smem + S::BIAS0 + wave * S::BIAS_SLOT), 16,
                                               (live && n < p.N) ? (unsigned)n * 4u : OOB, 0, 0, 0);
    }
#else
    (void)n0; (void)live;
#endif
  };

  // fragment reads (igemm_kernel's addressing)
  const int rd_x = ((fq ^ (fr & 7)) << 4);
  const int a_rd = (wm * 64 + fr) * 128 + rd_x;
  const int w_rd = (wn * (BN / 2) + fr) * 128 + rd_x;
  bf16x8 xa[MI], wa[NI], xb[MI], wb[NI];
  auto read_frags = [&](int buf, int ks, bf16x8 (&xf)[MI], bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
    const char* pa = smem + buf * S::STAGE + (a_rd ^ (ks << 6));
    const char* pw = smem + buf * S::STAGE + S::A_BYTES + (w_rd ^ (ks << 6));
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(pa + i * 2048);
#pragma unroll
    for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
  };
  f32x4 acc[MI][NI];
  auto mfmas = [&](const bf16x8 (&xf)[MI], const bf16x8 (&wf)[NI]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = MFMA_16x16x32_ST(wf[j], xf[i], acc[i][j], 0, 0, 0);
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // row pieces of a 16-row pass: piece idx = lane + 64 q -> (row r, 16-byte chunk cc) of the wave's [16][BN/2] slab
  int pr[NQ], pc[NQ];
  bool pv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int idx = lane + 64 * q;
    pr[q] = idx / PCH;
    pc[q] = idx - pr[q] * PCH;
    pv[q] = idx < 16 * PCH;
  }

  unsigned ao_c[A_CH], wo_c[W_CH], ao_n[A_CH], wo_n[W_CH];
  tile_offsets(0, ao_c, wo_c);
  tile_offsets(1, ao_n, wo_n);
  // prologue: K-tiles 0, 1, 2 of the first tile; K-tiles 0 and 1 are waited for (see the header: K-tile 0's mid wait then holds trivially)
  {
    int m0f, n0f;
    tile_mn(0, m0f, n0f);
    bias_fire(n0f, true);
  }
  fire(0, ao_c, wo_c, 0);
  fire(1, ao_c, wo_c, 128);
  fire(2, ao_c, wo_c, 256);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DRAIN ? 0 : NDMA) : "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, xa, wa);
  int cur = 0;

  // one K-tile.  FIRST: K-tile 0 of a tile (its wait tolerates the previous tile's stores); LAST: K-tile nk-1 (no DMA: the stage becomes
  // the epilogue's scratch; no fragment reads of the next K-tile: their registers hold the residual rows instead)
  unsigned ao_f[A_CH], wo_f[W_CH];
  int soff_f = 0;
  f32x4 biasv[NI];
  u32x4 resid[RES ? ST : 1];
  auto ktile = [&](auto first_c, auto last_c, int j, int m0, int n0) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    const int nx = cur == 2 ? 0 : cur + 1;
    read_frags(cur, 1, xb, wb);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (!LAST) {
      // source of the DMA this K-tile issues: K-tile j+3 of this tile, or K-tile j+3-nk of the next one
      const int fk = j + 3;
      const bool nxt = fk >= nk;
      soff_f = (nxt ? fk - nk : fk) * 128;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) ao_f[i] = nxt ? ao_n[i] : ao_c[i];
#pragma unroll
      for (int i = 0; i < W_CH; ++i) wo_f[i] = nxt ? wo_n[i] : wo_c[i];
    }
    mfmas(xa, wa);
#pragma unroll
    for (int r = 0; r < MI * NI / 2; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
      __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);      // selects
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DRAIN ? 0 : (FIRST ? ST + 1 + NDMA : NDMA)) : "memory");
    __builtin_amdgcn_s_setprio(1);
    if constexpr (!LAST) {
      read_frags(nx, 0, xa, wa);
      fire(cur, ao_f, wo_f, soff_f);
      mfmas(xb, wb);
#pragma unroll
      for (int r = 0; r < MI * NI / 2; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // 2 MFMA
        if (r < (MI + NI + 1) / 2) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 ds_read
        } else {
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);  // M0
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // 2 LDS-DMA
        }
      }
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
      // the residual rows of the four passes: in flight under the last MFMA group
      if constexpr (RES) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int m = m0 + wm * 64 + ps * 16 + pr[q], n = n0 + wn * (BN / 2) + pc[q] * 8;
            const bool ok = pv[q] && m < p.M && n < p.N;
            resid[ps * NQ + q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ok ? (unsigned)m * ldr2 + (unsigned)n * 2u : OOB, 0, 0);
          }
      }
#endif
      mfmas(xb, wb);
    }
    __builtin_amdgcn_s_setprio(0);
    cur = nx;
  };
  auto add2 = [](uint32_t a, uint32_t b) __attribute__((always_inline)) {
    return pack_bf16x2(bf16_to_f32((bf16_t)(a & 0xffff)) + bf16_to_f32((bf16_t)(b & 0xffff)),
                       bf16_to_f32((bf16_t)(a >> 16)) + bf16_to_f32((bf16_t)(b >> 16)));
  };
  using TC = std::true_type;
  using FC = std::false_type;

  for (int t = 0; t < T; ++t) {
    int m0, n0;
    tile_mn(t, m0, n0);
    zero_acc();
    ktile(TC{}, FC{}, 0, m0, n0);
    for (int j = 1; j < nk - 1; ++j) ktile(FC{}, FC{}, j, m0, n0);
    ktile(FC{}, TC{}, nk - 1, m0, n0);
    // ---- epilogue.  `cur` is the next tile's K-tile 0 (landed and visible: the last mid barrier), cur+1 its K-tile 1 (in flight or
    // landed), cur+2 = the stage the last K-tile used: dead everywhere, the wave's scratch = its own 4 KB of that stage's A area.
    const int sfree = cur == 0 ? 2 : cur - 1;
    const int scr_off = (int)lds_base + sfree * S::STAGE + wave * 4096;      // LDS byte address of the wave's scratch
    // the bias of the lane's column quads, from the wave's slot (its DMA is older than every ring wait of this tile)
#pragma unroll
    for (int jn = 0; jn < NI; ++jn)
      biasv[jn] = *reinterpret_cast<const f32x4*>(smem + S::BIAS0 + wave * S::BIAS_SLOT + (jn * 16 + fq * 4) * 4);
#pragma unroll
    for (int ps = 0; ps < ((ABL & 8) ? 0 : 4); ++ps) {
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) {
        const f32x4 v = acc[ps][jn] + biasv[jn];
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        // (inline asm: a compiler-visible LDS write that may alias a pending LDS-DMA makes the compiler drain vmcnt(0) in front of it --
        //  the scratch IS a DMA target; the wave's own program order is what makes it safe, see the header)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(scr_off + fr * S::SCR_STRIDE + (jn * 16 + fq * 4) * 2)),
                     "v"(__builtin_bit_cast(unsigned long long, o))
                     : "memory");
      }
      u32x4 ov[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        asm volatile("ds_read_b128 %0, %1" : "=v"(ov[q]) : "v"((unsigned)(scr_off + pr[q] * S::SCR_STRIDE + pc[q] * 16)) : "memory");
      if constexpr (NQ == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[NQ - 1])::"memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov[0]), "+v"(ov[NQ - 1])::"memory");
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if constexpr (RES) {
          const u32x4 rr = resid[ps * NQ + q];
          ov[q][0] = add2(ov[q][0], rr[0]); ov[q][1] = add2(ov[q][1], rr[1]);
          ov[q][2] = add2(ov[q][2], rr[2]); ov[q][3] = add2(ov[q][3], rr[3]);
        }
        const int m = m0 + wm * 64 + ps * 16 + pr[q], n = n0 + wn * (BN / 2) + pc[q] * 8;
        const bool ok = pv[q] && m < p.M && n < p.N && !(ABL & 1);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_buffer_store_b128(ov[q], rs_c, ok ? (unsigned)m * ldc2 + (unsigned)n * 2u : OOB, 0, 0);
#else
        (void)ok;
#endif
      }
    }
    // the deferred DMA: K-tile 2 of the next tile into the scratch's stage, once my scratch reads have completed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      int m0n, n0n;
      tile_mn(t + 1, m0n, n0n);
      bias_fire(n0n, t + 1 < T);
    }
    fire(sfree, ao_n, wo_n, 256);
    read_frags(cur, 0, xa, wa);
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ao_c[i] = ao_n[i];
#pragma unroll
    for (int i = 0; i < W_CH; ++i) wo_c[i] = wo_n[i];
    tile_offsets(t + 2, ao_n, wo_n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill requests past the end still target this block's LDS
}

template <int BN, bool RES>
int pgemm_launch_impl(const GemmParams& p, int grid, hipStream_t st) {
  constexpr int LDS = PS<BN>::TOTAL;
  if (hedit_test_drained()) {
    if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&pgemm_kernel<BN, RES, true>), LDS)) return rc;
    hipLaunchKernelGGL((pgemm_kernel<BN, RES, true>), dim3(grid), dim3(NT), LDS, st, p);
  } else {
    if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&pgemm_kernel<BN, RES, false>), LDS)) return rc;
    hipLaunchKernelGGL((pgemm_kernel<BN, RES, false>), dim3(grid), dim3(NT), LDS, st, p);
  }
  LAUNCH_CHECK();
  return HEDIT_OK;
}

}  // namespace

// Does the persistent kernel take this launch?  (A choice by shape only -- it never changes a result: same MFMA chain, same epilogue
// arithmetic as igemm_kernel.)  p as gemm_launch has prepared it (splits resolved, a_bytes / w_bytes set).
bool pgemm_supported(const GemmParams& p, int splits, int bn) {
  if (hedit_test_flags() & 8) return false;                 // tests: the one-shot kernel for the A/B comparison
  if (p.mode != 0 || splits != 1 || p.geglu || p.partial || p.raw_f32 || p.gn_part || p.op_bf16) return false;
  if (p.chunk_kt > 0 && p.chunk_kt < p.K / BK) return false;
  if (bn != 128 && bn != 160) return false;
  if (p.K / BK < 3) return false;
  if (p.N % 8 != 0 || p.ldc % 8 != 0 || (p.residual && p.ldr % 8 != 0) || p.lda % 8 != 0) return false;
  if ((double)p.M * p.ldc * 2.0 >= 2040.0 * 1048576.0 || (p.residual && (double)p.M * p.ldr * 2.0 >= 2040.0 * 1048576.0)) return false;
  int cus = 256;
  if (hedit_cu_count(&cus)) return false;
  const long tiles = (long)cdiv(p.M, BM) * cdiv(p.N, bn);
  return tiles >= 2L * cus;                                 // at least two tiles per block: something to carry the ring across
}

int pgemm_launch(const GemmParams& p, int bn, hipStream_t st) {
  int cus = 256;
  if (int rc = hedit_cu_count(&cus)) return rc;
  const long tiles = (long)cdiv(p.M, BM) * cdiv(p.N, bn);
  const int grid = (int)(tiles < cus ? tiles : cus);
  if (bn == 160) return p.residual ? pgemm_launch_impl<160, true>(p, grid, st) : pgemm_launch_impl<160, false>(p, grid, st);
  return p.residual ? pgemm_launch_impl<128, true>(p, grid, st) : pgemm_launch_impl<128, false>(p, grid, st);
}
