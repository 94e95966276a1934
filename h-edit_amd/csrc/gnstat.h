// GroupNorm statistics taken where a tensor is PRODUCED (the bf16 output tile of igemm_kernel / the split-K reduce)
// instead of by a pass that reads the tensor again (gn_partial_kernel: 6 % of the face loop's busy time at 8 faces,
// profiles/r05_face_kernel_stats.txt).  Replaces the statistics half of torch's GroupNorm in the ResNet blocks of
// face-swapping/diffusion/diffusion.py:27-33,115-134 (Normalize -> swish -> conv).
//
// What is stored: for every UNIT of 128 consecutive rows (pixels; a unit never straddles two images: HW % 128 == 0) and
// every PAIR of adjacent output channels, (sum, sum of squares) of the bf16 values as stored, fp32:
//     part[(unit * (N / 2) + pair) * 2 + {0, 1}]
// Pairs, not groups: a consumer may normalise the tensor alone (cpg = C / 32 channels per group) or as one half of a skip
// concatenation (cpg = (C + C') / 32), and every cpg of these networks is even -- gn_fold_kernel (norm.hip) assembles
// whatever groups the consumer has from the pairs of one or two producers.
//
// The summation tree of a (unit, pair) is FIXED, a function of the row index inside the unit only, so that every
// producer form -- 128-row tile, 256-row tile, three-stage ring, row-sharing loop, chunk fold, split-K slabs + reduce --
// writes the same bits and an image's statistics do not depend on what else shares the launch:
//     piece(row)  = (a + b, fma(b, b, a * a))                      the pair's two bf16 values of one row
//     T[r]        = ((piece(r) + piece(r + 32)) + piece(r + 64)) + piece(r + 96)          r = 0 .. 31
//     unit        = (..((T[0] + T[16]) + (T[1] + T[17])) + ..) + (T[15] + T[31])
// The thread layout all producers share for it: thread (r = tid / 16, c = tid % 16) owns the 16-byte piece c (8 channels
// = 4 pairs) of rows r, r + RSTEP, r + 2 RSTEP, ... of a 128-column tile (RSTEP = threads / 16: 16 or 32).
// __fadd_rn / __fmul_rn / __fmaf_rn: no contraction the compiler could apply differently in two kernels.  The helpers are
// __host__ __device__ so that tests/test_host_gn_stats.py can run the tree of every producer form on the CPU (host pass: plain
// fp32 operations, the build has -ffp-contract=off).
#pragma once
#include "common.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define GNS_ADD(a, b) __fadd_rn((a), (b))
#define GNS_MUL(a, b) __fmul_rn((a), (b))
#define GNS_FMA(a, b, c) __fmaf_rn((a), (b), (c))
#else
#include <math.h>
#define GNS_ADD(a, b) ((a) + (b))
#define GNS_MUL(a, b) ((a) * (b))
#define GNS_FMA(a, b, c) fmaf((a), (b), (c))
#endif

constexpr int GNS_UNIT = 128;                       // rows per statistics unit
constexpr int GNS_RED_BYTES = 32 * 64 * 8;          // LDS per unit: T[32 rows][64 pairs] of float2

struct GnPiece {
  float s[4], q[4];
};

__host__ __device__ __forceinline__ GnPiece gns_piece(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  GnPiece g;
  const uint32_t w[4] = {w0, w1, w2, w3};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = bf16_to_f32((bf16_t)(w[k] & 0xffff)), b = bf16_to_f32((bf16_t)(w[k] >> 16));
    g.s[k] = GNS_ADD(a, b);
    g.q[k] = GNS_FMA(b, b, GNS_MUL(a, a));
  }
  return g;
}
__host__ __device__ __forceinline__ void gns_add(GnPiece& t, const GnPiece& g) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t.s[k] = GNS_ADD(t.s[k], g.s[k]);
    t.q[k] = GNS_ADD(t.q[k], g.q[k]);
  }
}

// Rows of a tile of `rows` rows (128 or 256) held by thread (r, c) as pieces it = 0 .. rows / RSTEP - 1 (row = r + RSTEP * it):
// which T slot a piece belongs to and whether it opens it.  RSTEP = 16: slot = it & 1 (T[r], T[r + 16]), four pieces each;
// RSTEP = 32: slot = it >> 2 (unit 0 / unit 1), T[r] of that unit.
template <int RSTEP>
__host__ __device__ __forceinline__ constexpr int gns_slot(int it) { return RSTEP == 16 ? (it & 1) : (it >> 2); }
template <int RSTEP>
__host__ __device__ __forceinline__ constexpr bool gns_first(int it) { return RSTEP == 16 ? it < 2 : (it & 3) == 0; }
// LDS row (unit * 32 + r') of slot `slot` for the thread with row index r
template <int RSTEP>
__host__ __device__ __forceinline__ int gns_red_row(int r, int slot) { return RSTEP == 16 ? r + 16 * slot : slot * 32 + r; }

__host__ __device__ __forceinline__ void gns_store_t(float2* red, int red_row, int c, const GnPiece& t) {
#pragma unroll
  for (int k = 0; k < 4; ++k) red[red_row * 64 + c * 4 + k] = make_float2(t.s[k], t.q[k]);
}

// after a barrier: thread `pk` (0 .. 63) of unit `u` folds the 32 T rows of its pair
__host__ __device__ __forceinline__ float2 gns_fold_unit(const float2* red, int u, int pk) {
  const float2* t = red + (long)u * 32 * 64 + pk;
  float2 a = t[0], b = t[16 * 64];
  float s = GNS_ADD(a.x, b.x), q = GNS_ADD(a.y, b.y);
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    a = t[r * 64];
    b = t[(r + 16) * 64];
    s = GNS_ADD(s, GNS_ADD(a.x, b.x));
    q = GNS_ADD(q, GNS_ADD(a.y, b.y));
  }
  return make_float2(s, q);
}
