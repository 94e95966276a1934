// HBM-bound kernels around the GEMMs: GroupNorm(+SiLU), LayerNorm, GEGLU, concat, casts, the
// time-embedding GEMV chain, conv_in / conv_out and weight packing.  NHWC bf16 activations, all
// global accesses 16 B per lane along the contiguous channel axis, fp32 statistics.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {


// ------------------------------------------------------------------ GroupNorm
// stage 1: per (batch, pixel-slab) partial sums per group.  Threads are laid out as
// (row r, channel-vector cv); a thread keeps the 8 channels of its cv in registers across the
// rows of the slab, then the block folds rows and channels into the 32 group sums through LDS.
constexpr int GN_MAXV = 2;   // channel vectors per thread when C/8 > 256 (C up to 4096)

__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                         int HW, int C, int G, int nslab) {
  extern __shared__ float lds[];   // [R][C][2]
  const int CV = C / 8;
  const int slab = blockIdx.x, b = blockIdx.y;
  const int pix_per = (HW + nslab - 1) / nslab;
  const int p0 = slab * pix_per;
  int p1 = p0 + pix_per;
  if (p1 > HW) p1 = HW;
  const int R = CV <= 256 ? 256 / CV : 1;
  const int r = CV <= 256 ? threadIdx.x / CV : 0;
  const int cv0 = CV <= 256 ? threadIdx.x % CV : threadIdx.x;
  const bool active = r < R;
  float s[GN_MAXV][8], q[GN_MAXV][8];
#pragma unroll
  for (int v = 0; v < GN_MAXV; ++v)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[v][j] = q[v][j] = 0.f;
  if (active) {
    const bf16_t* xb = x + (long)b * HW * C;
    int p = p0 + r;
    if (CV <= 256) {
      // (the common shape: one 16-byte chunk per thread and pixel.  Four pixels requested before the first is summed -- the
      //  sums still take the pixels in order, so the bits are those of the plain loop below)
      auto acc1 = [&](const uint4& u) __attribute__((always_inline)) {
        float f[8];
        unpack8(u, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[0][j] += f[j]; q[0][j] += f[j] * f[j]; }
      };
      const bf16_t* xc = xb + cv0 * 8;
      for (; p + 3 * R < p1; p += 4 * R) {
        const uint4 u0 = *reinterpret_cast<const uint4*>(xc + (long)p * C);
        const uint4 u1 = *reinterpret_cast<const uint4*>(xc + (long)(p + R) * C);
        const uint4 u2 = *reinterpret_cast<const uint4*>(xc + (long)(p + 2 * R) * C);
        const uint4 u3 = *reinterpret_cast<const uint4*>(xc + (long)(p + 3 * R) * C);
        acc1(u0); acc1(u1); acc1(u2); acc1(u3);
      }
    }
    for (; p < p1; p += R) {
#pragma unroll
      for (int v = 0; v < GN_MAXV; ++v) {
        int cv = cv0 + v * 256;
        if (cv < CV && (v == 0 || CV > 256)) {
          uint4 u = *reinterpret_cast<const uint4*>(xb + (long)p * C + cv * 8);
          float f[8];
          unpack8(u, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s[v][j] += f[j]; q[v][j] += f[j] * f[j]; }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < GN_MAXV; ++v) {
      int cv = cv0 + v * 256;
      if (cv < CV && (v == 0 || CV > 256)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          lds[((long)r * C + cv * 8 + j) * 2 + 0] = s[v][j];
          lds[((long)r * C + cv * 8 + j) * 2 + 1] = q[v][j];
        }
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x, cpg = C / G;
    float ss = 0.f, qq = 0.f;
    for (int rr = 0; rr < R; ++rr)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        ss += lds[((long)rr * C + c) * 2 + 0];
        qq += lds[((long)rr * C + c) * 2 + 1];
      }
    float* dst = part + (((long)b * nslab + slab) * G + g) * 2;
    dst[0] = ss;
    dst[1] = qq;
  }
}

// stage 2: fold slabs -> mean / rstd per group (one wave per group, lanes stride over slabs, fixed
// summation order), then per-channel scale & shift for this batch item
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ ss,
                                                          int HW, int C, int G, int nslab, float eps,
                                                          float* __restrict__ stats) {
  __shared__ float mean[64], rstd[64];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int g = wave; g < G; g += 4) {
    float s = 0.f, q = 0.f;
    for (int sl = lane; sl < nslab; sl += 64) {
      const float2 v = *reinterpret_cast<const float2*>(part + (((long)b * nslab + sl) * G + g) * 2);
      s += v.x;
      q += v.y;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) {
      const float n = (float)HW * (float)(C / G);
      const float m = s / n;
      float var = q / n - m * m;
      var = var < 0.f ? 0.f : var;
      mean[g] = m;
      rstd[g] = rsqrtf(var + eps);
      if (stats) *reinterpret_cast<float2*>(stats + ((long)b * G + g) * 2) = make_float2(m, rstd[g]);
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = rstd[g] * gamma[c];
    *reinterpret_cast<float2*>(ss + ((long)b * C + c) * 2) = make_float2(sc, beta[c] - mean[g] * sc);
  }
}

// stage 3: y = x*scale + shift (+SiLU)
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       const float* __restrict__ ss, long total_v, int HW, int C, int silu) {
  const int CV = C / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_v; i += (long)gridDim.x * 256) {
    const long pix = i / CV;
    const int cv = (int)(i - pix * CV);
    const int b = (int)(pix / HW);
    uint4 u = *reinterpret_cast<const uint4*>(x + i * 8);
    float f[8];
    unpack8(u, f);
    const float4* t = reinterpret_cast<const float4*>(ss + ((long)b * C + cv * 8) * 2);
    float4 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    f[0] = f[0] * t0.x + t0.y; f[1] = f[1] * t0.z + t0.w;
    f[2] = f[2] * t1.x + t1.y; f[3] = f[3] * t1.z + t1.w;
    f[4] = f[4] * t2.x + t2.y; f[5] = f[5] * t2.z + t2.w;
    f[6] = f[6] * t3.x + t3.y; f[7] = f[7] * t3.z + t3.w;
    if (silu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8(f);
  }
}

// stage 3, row form: a thread keeps ONE group of 8 channels (its 8 scale / shift pairs in registers) and walks the pixels
// of its slab, so an iteration is one 16-byte load and one 16-byte store; the form above re-reads 64 bytes of the
// coefficient table per 16 bytes of data.  Same arithmetic per element.
// The statistics' second stage is folded into this kernel's prologue (no gn_finalize launch on this path): every block
// folds the slab partials of its image into mean / rstd per group -- the same lanes, the same order and the same formulas
// as gn_finalize_kernel, hence the same bits -- and each thread derives the scale / shift pairs of its 8 channels.
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                            const float* __restrict__ part, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int HW, int C, int G, int nstat, float eps,
                                                            int silu, int nslab) {
  __shared__ float mean[64], rstd[64];
  const int CV = C / 8;
  const int slab = blockIdx.x, b = blockIdx.y;
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = wave; g < G; g += 4) {
      float s = 0.f, q = 0.f;
      for (int sl = lane; sl < nstat; sl += 64) {
        const float2 v = *reinterpret_cast<const float2*>(part + (((long)b * nstat + sl) * G + g) * 2);
        s += v.x;
        q += v.y;
      }
      s = wave_sum(s);
      q = wave_sum(q);
      if (lane == 0) {
        const float n = (float)HW * (float)(C / G);
        const float m = s / n;
        float var = q / n - m * m;
        var = var < 0.f ? 0.f : var;
        mean[g] = m;
        rstd[g] = rsqrtf(var + eps);
      }
    }
    __syncthreads();
  }
  const int pix_per = (HW + nslab - 1) / nslab;
  const int p0 = slab * pix_per;
  int p1 = p0 + pix_per;
  if (p1 > HW) p1 = HW;
  const int R = 256 / CV;                     // (launched only for CV <= 256)
  const int r = threadIdx.x / CV;
  const int cv = threadIdx.x - r * CV;
  if (r >= R) return;
  float sc[8], sh[8];
  {
    const int cpg = C / G;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv * 8 + j, g = c / cpg;
      sc[j] = rstd[g] * gamma[c];
      sh[j] = beta[c] - mean[g] * sc[j];
    }
  }
  const float4 t0 = {sc[0], sh[0], sc[1], sh[1]}, t1 = {sc[2], sh[2], sc[3], sh[3]};
  const float4 t2 = {sc[4], sh[4], sc[5], sh[5]}, t3 = {sc[6], sh[6], sc[7], sh[7]};
  const bf16_t* xb = x + (long)b * HW * C + cv * 8;
  bf16_t* yb = y + (long)b * HW * C + cv * 8;
  auto one = [&](const uint4& u, int p) __attribute__((always_inline)) {
    float f[8];
    unpack8(u, f);
    f[0] = f[0] * t0.x + t0.y; f[1] = f[1] * t0.z + t0.w;
    f[2] = f[2] * t1.x + t1.y; f[3] = f[3] * t1.z + t1.w;
    f[4] = f[4] * t2.x + t2.y; f[5] = f[5] * t2.z + t2.w;
    f[6] = f[6] * t3.x + t3.y; f[7] = f[7] * t3.z + t3.w;
    if (silu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
    }
    *reinterpret_cast<uint4*>(yb + (long)p * C) = pack8(f);
  };
  // four pixels requested before the first is touched: with one 16-byte load in flight per thread the pass left the HBM
  // queue half empty (element-wise arithmetic unchanged)
  int p = p0 + r;
  for (; p + 3 * R < p1; p += 4 * R) {
    const uint4 u0 = *reinterpret_cast<const uint4*>(xb + (long)p * C);
    const uint4 u1 = *reinterpret_cast<const uint4*>(xb + (long)(p + R) * C);
    const uint4 u2 = *reinterpret_cast<const uint4*>(xb + (long)(p + 2 * R) * C);
    const uint4 u3 = *reinterpret_cast<const uint4*>(xb + (long)(p + 3 * R) * C);
    one(u0, p); one(u1, p + R); one(u2, p + 2 * R); one(u3, p + 3 * R);
  }
  for (; p < p1; p += R) one(*reinterpret_cast<const uint4*>(xb + (long)p * C), p);
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
constexpr int LN_MAXV = 5;   // C up to 2560
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int CV = C / 8;
  float f[LN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < LN_MAXV; ++v) {
    int cv = lane + v * 64;
    if (cv < CV) {
      uint4 u = *reinterpret_cast<const uint4*>(x + row * C + cv * 8);
      unpack8(u, f[v]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[v][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < LN_MAXV; ++v) {
    int cv = lane + v * 64;
    if (cv < CV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = f[v][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int v = 0; v < LN_MAXV; ++v) {
    int cv = lane + v * 64;
    if (cv < CV) {
      const float4* g4 = reinterpret_cast<const float4*>(gamma + cv * 8);
      const float4* b4 = reinterpret_cast<const float4*>(beta + cv * 8);
      float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
      float o[8];
      o[0] = (f[v][0] - mean) * rstd * g0.x + b0.x; o[1] = (f[v][1] - mean) * rstd * g0.y + b0.y;
      o[2] = (f[v][2] - mean) * rstd * g0.z + b0.z; o[3] = (f[v][3] - mean) * rstd * g0.w + b0.w;
      o[4] = (f[v][4] - mean) * rstd * g1.x + b1.x; o[5] = (f[v][5] - mean) * rstd * g1.y + b1.y;
      o[6] = (f[v][6] - mean) * rstd * g1.z + b1.z; o[7] = (f[v][7] - mean) * rstd * g1.w + b1.w;
      *reinterpret_cast<uint4*>(y + row * C + cv * 8) = pack8(o);
    }
  }
}

// Narrow rows (C <= 320): a whole wave per row would leave 24+ of its 64 lanes idle.  Here 8 lanes
// share a row (5 x 16-byte chunks each at C = 320, consecutive lanes on consecutive chunks = one
// 128-byte line per load), a wave normalises 8 rows, and the two reductions stay inside the 8-lane
// groups.  Same arithmetic order per row as layernorm_kernel up to the reduction tree.
constexpr int LNN_MAXV = 5;
// RED: the rows do not exist yet -- they are the sum of split-K slabs (+ bias, rounded, + residual, rounded: the arithmetic
// of gemm.hip's splitk_reduce_kernel, element for element).  The kernel then forms each 16-byte piece itself, writes it
// to `x` (the GEMM's output, which the residual stream keeps) and normalises it: one launch instead of reduce +
// LayerNorm behind a small split-K GEMM, same bits.
struct LnReduce {
  const float* partial;      // [splits][rows][C]
  int splits;
  const float* bias;         // [C] or null
  const bf16_t* residual;    // [rows][ldr] or null
  int ldr;
};
template <int LANES, bool RED = false>     // 8 / 16 / 32 lanes per row, up to LNN_MAXV 16-byte chunks per lane (C = 320 / 640 / 1280: five each)
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               long rows, int C, float eps, LnReduce rd) {
  const int l8 = threadIdx.x & (LANES - 1);
  const long row = (long)blockIdx.x * (256 / LANES) + threadIdx.x / LANES;
  const bool live = row < rows;
  const int CV = C / 8, NV = CV / LANES;       // chunks per row, chunks per lane
  float f[LNN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < LNN_MAXV; ++v) {
    if (v < NV && live) {
      uint4 u;
      if constexpr (RED) {
        const int n = (l8 + v * LANES) * 8;
        uint32_t w[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          for (int sp = 0; sp < rd.splits; ++sp)
            a += *reinterpret_cast<const f32x4*>(rd.partial + ((long)sp * rows + row) * C + n + half * 4);
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (rd.bias) b = *reinterpret_cast<const f32x4*>(rd.bias + n + half * 4);
          a = a + b;
          uint32_t o0 = pack_bf16x2(a[0], a[1]), o1 = pack_bf16x2(a[2], a[3]);
          if (rd.residual) {
            const uint2 r = *reinterpret_cast<const uint2*>(rd.residual + row * rd.ldr + n + half * 4);
            o0 = pack_bf16x2(bf16_to_f32((bf16_t)(o0 & 0xffff)) + bf16_to_f32((bf16_t)(r.x & 0xffff)),
                             bf16_to_f32((bf16_t)(o0 >> 16)) + bf16_to_f32((bf16_t)(r.x >> 16)));
            o1 = pack_bf16x2(bf16_to_f32((bf16_t)(o1 & 0xffff)) + bf16_to_f32((bf16_t)(r.y & 0xffff)),
                             bf16_to_f32((bf16_t)(o1 >> 16)) + bf16_to_f32((bf16_t)(r.y >> 16)));
          }
          w[half * 2] = o0;
          w[half * 2 + 1] = o1;
        }
        u = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(x + row * C + n) = u;
      } else {
        u = *reinterpret_cast<const uint4*>(x + row * C + (l8 + v * LANES) * 8);
      }
      unpack8(u, f[v]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[v][j];
    }
  }
#pragma unroll
  for (int o = 1; o < LANES; o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < LNN_MAXV; ++v) {
    if (v < NV && live) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = f[v][j] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = 1; o < LANES; o <<= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int v = 0; v < LNN_MAXV; ++v) {
    if (v < NV && live) {
      const int cv = l8 + v * LANES;
      const float4* g4 = reinterpret_cast<const float4*>(gamma + cv * 8);
      const float4* b4 = reinterpret_cast<const float4*>(beta + cv * 8);
      float4 g0 = g4[0], g1 = g4[1], b0 = b4[0], b1 = b4[1];
      float o[8];
      o[0] = (f[v][0] - mean) * rstd * g0.x + b0.x; o[1] = (f[v][1] - mean) * rstd * g0.y + b0.y;
      o[2] = (f[v][2] - mean) * rstd * g0.z + b0.z; o[3] = (f[v][3] - mean) * rstd * g0.w + b0.w;
      o[4] = (f[v][4] - mean) * rstd * g1.x + b1.x; o[5] = (f[v][5] - mean) * rstd * g1.y + b1.y;
      o[6] = (f[v][6] - mean) * rstd * g1.z + b1.z; o[7] = (f[v][7] - mean) * rstd * g1.w + b1.w;
      *reinterpret_cast<uint4*>(y + row * C + cv * 8) = pack8(o);
    }
  }
}

// ------------------------------------------------------------------ GEGLU: y = h * gelu(g), x = [h | g]
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long rows, int inner) {
  const int IV = inner / 8;
  const long total = rows * IV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / IV;
    const int v = (int)(i - row * IV);
    const bf16_t* src = x + row * (2L * inner) + v * 8;
    uint4 uh = *reinterpret_cast<const uint4*>(src);
    uint4 ug = *reinterpret_cast<const uint4*>(src + inner);
    float h[8], g[8];
    unpack8(uh, h);
    unpack8(ug, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] *= gelu_erf_f(g[j]);
    *reinterpret_cast<uint4*>(y + row * inner + v * 8) = pack8(h);
  }
}

// a == nullptr: the left ca columns of y are already in place (their producer wrote them with ldc = ca + cb);
// only b is copied into columns [ca, ca + cb)
__global__ __launch_bounds__(256) void concat_right_kernel(const bf16_t* __restrict__ b, int ca, int cb, bf16_t* __restrict__ y, long rows) {
  const int BV = cb / 8;
  const long total = rows * BV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / BV;
    const int v = (int)(i - row * BV);
    *reinterpret_cast<uint4*>(y + row * (ca + cb) + ca + v * 8) = *reinterpret_cast<const uint4*>(b + row * cb + v * 8);
  }
}

__global__ __launch_bounds__(256) void concat_kernel(const bf16_t* __restrict__ a, int ca, const bf16_t* __restrict__ b, int cb,
                                                     bf16_t* __restrict__ y, long rows) {
  const int CV = (ca + cb) / 8, AV = ca / 8;
  const long total = rows * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / CV;
    const int v = (int)(i - row * CV);
    uint4 u = v < AV ? *reinterpret_cast<const uint4*>(a + row * ca + v * 8)
                     : *reinterpret_cast<const uint4*>(b + row * cb + (v - AV) * 8);
    *reinterpret_cast<uint4*>(y + i * 8) = u;
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = f32_to_bf16(x[i] * scale);
}

// text context fp32 [B][77][dim] -> bf16 [B][80][dim], rows 77..79 zero (16-byte aligned V^T rows)
__global__ __launch_bounds__(256) void ctx_pad_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int dim) {
  const long total = (long)B * HEDIT_CTXP * dim;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % dim);
    const long r = i / dim;
    const int tok = (int)(r % HEDIT_CTXP);
    const long b = r / HEDIT_CTXP;
    y[i] = tok < HEDIT_MAXW ? f32_to_bf16(x[(b * HEDIT_MAXW + tok) * dim + c]) : (bf16_t)0;
  }
}

__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I) {
  const long total = (long)O * 9 * I;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % I);
    const long t = idx / I;
    const int tap = (int)(t % 9);
    const int o = (int)(t / 9);
    out[idx] = f32_to_bf16(w[((long)o * I + i) * 9 + tap]);
  }
}

// rows of the FF1 projection interleaved so that (value, gate) of 16 output columns are adjacent
__global__ __launch_bounds__(256) void pack_geglu_rows_kernel(const float* __restrict__ w, bf16_t* __restrict__ ob, float* __restrict__ of,
                                                              int rows, int K) {
  const long total = (long)rows * K;
  const int half = rows / 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int k = (int)(idx % K);
    const int r = (int)(idx / K);
    const int t = r >> 5, u = r & 31;
    const int src = u < 16 ? t * 16 + u : half + t * 16 + (u - 16);
    const float v = w[(long)src * K + k];
    if (ob) ob[idx] = f32_to_bf16(v);
    if (of) of[idx] = v;
  }
}

// ------------------------------------------------------------------ GEMV (time-embedding chain)
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ W, const float* __restrict__ x, const float* __restrict__ b0,
                                                   const float* __restrict__ b1, float* __restrict__ out, int N, int K, int silu) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    uint4 u = *reinterpret_cast<const uint4*>(W + (long)n * K + k);
    float w[8];
    unpack8(u, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xv = x[k + j];
      if (silu) xv = silu_f(xv);
      acc += w[j] * xv;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) out[n] = acc + (b0 ? b0[n] : 0.f) + (b1 ? b1[n] : 0.f);
}

// sinusoidal timestep embedding, flip_sin_to_cos: [cos | sin](t * 10000^(-i/half))
__global__ void timestep_embed_kernel(float t, float* out, int dim) {
  const int half = dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float fr = expf(-9.210340371976184f * (float)i / (float)half);
    const float a = t * fr;
    out[i] = cosf(a);
    out[half + i] = sinf(a);
  }
}

// the DDPM code base's variant (face model, face-swapping/diffusion/diffusion.py:6-24):
// [sin | cos](t * 10000^(-i/(half-1)))
__global__ void timestep_embed_ddpm_kernel(float t, float* out, int dim) {
  const int half = dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float fr = expf(-9.210340371976184f / (float)(half - 1) * (float)i);
    const float a = t * fr;
    out[i] = sinf(a);
    out[half + i] = cosf(a);
  }
  if ((dim & 1) && threadIdx.x == 0) out[dim - 1] = 0.f;
}

// ------------------------------------------------------------------ conv_in (Cin <= 8, K = 9 Cin tiny -> VALU)
// Block = CI_PIX consecutive pixels x all output channels.  The 9*Cin input taps of the block's pixels
// and the whole (transposed) weight matrix live in LDS; a thread produces 8 consecutive output
// channels of FOUR pixels (the weight vector of a tap is read once for the four: the loop is LDS-read-bound) -> 16-byte
// NHWC stores.  Every output element is the same k-ordered fp32 chain whatever the blocking.
constexpr int CI_PIX = 128;
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      bf16_t* __restrict__ y, int B, int Cin, int H, int Wd, int Cout) {
  extern __shared__ float lds[];
  const int K = Cin * 9;
  float* sw = lds;                    // [K][Cout]   (k = ci*9 + tap)
  float* sx = lds + (long)K * Cout;   // [K][CI_PIX]  (tap-major: the four pixels of a thread are one 16-byte read)
  for (int i = threadIdx.x; i < K * Cout; i += 256) {
    const int co = i / K, k = i - co * K;      // global layout [Cout][Cin][3][3] = [Cout][K]
    sw[k * Cout + co] = w[i];
  }
  const long pix0 = (long)blockIdx.x * CI_PIX;
  const long npix = (long)B * H * Wd;
  for (int i = threadIdx.x; i < CI_PIX * K; i += 256) {
    const int k = i / CI_PIX, pl = i - k * CI_PIX;
    const long pix = pix0 + pl;
    float v = 0.f;
    if (pix < npix) {
      const int ox = (int)(pix % Wd), oy = (int)((pix / Wd) % H), b = (int)(pix / ((long)Wd * H));
      const int ci = k / 9, tap = k - ci * 9;
      const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd) v = x[(((long)b * Cin + ci) * H + iy) * Wd + ix];
    }
    sx[k * CI_PIX + pl] = v;
  }
  __syncthreads();
  const int CV = Cout / 8;
  for (int i = threadIdx.x; i < (CI_PIX / 4) * CV; i += 256) {
    const int pg = i / CV, cv = i - pg * CV;           // pixel group (4 pixels), channel group (8 channels)
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[q][j] = bias[cv * 8 + j];
    for (int k = 0; k < K; ++k) {
      const float4 xv = *reinterpret_cast<const float4*>(sx + k * CI_PIX + pg * 4);
      const float4 w0 = *reinterpret_cast<const float4*>(sw + k * Cout + cv * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(sw + k * Cout + cv * 8 + 4);
      const float xq[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q][0] += xq[q] * w0.x; acc[q][1] += xq[q] * w0.y; acc[q][2] += xq[q] * w0.z; acc[q][3] += xq[q] * w0.w;
        acc[q][4] += xq[q] * w1.x; acc[q][5] += xq[q] * w1.y; acc[q][6] += xq[q] * w1.z; acc[q][7] += xq[q] * w1.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long pix = pix0 + pg * 4 + q;
      if (pix < npix) *reinterpret_cast<uint4*>(y + pix * Cout + cv * 8) = pack8(acc[q]);
    }
  }
}

// ------------------------------------------------------------------ conv_out (Cout <= 4): one wave per pixel
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int B, int H, int Wd, int C, int Cout) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long)B * H * Wd) return;
  const int ox = (int)(pix % Wd);
  const int oy = (int)((pix / Wd) % H);
  const int b = (int)(pix / ((long)Wd * H));
  const int CV = C / 8;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int idx = lane; idx < 9 * CV; idx += 64) {
    const int tap = idx / CV, cv = idx - tap * CV;
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)Wd) continue;
    uint4 u = *reinterpret_cast<const uint4*>(x + (((long)b * H + iy) * Wd + ix) * C + cv * 8);
    float f[8];
    unpack8(u, f);
    for (int co = 0; co < Cout; ++co) {
      uint4 uw = *reinterpret_cast<const uint4*>(w + ((long)co * 9 + tap) * C + cv * 8);
      float ww[8];
      unpack8(uw, ww);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[co] += f[j] * ww[j];
    }
  }
  for (int co = 0; co < Cout; ++co) {
    float v = wave_sum(acc[co]);
    if (lane == 0) y[(((long)b * Cout + co) * H + oy) * Wd + ox] = v + bias[co];
  }
}

// Small images (HW < 1024: the 16 x 16 and 8 x 8 levels), ONE launch: a block owns one (image, group) -- at most 256
// pixels x 80 channels = 40 KB of bf16 pairs, parked in LDS --, sums it in a fixed order (thread-sequential over its
// strided pairs, butterfly over the wave, waves 0..3 in order: a function of the image alone, like every norm here) and
// normalises from LDS.  Three launches (partials, finalize, apply) became one; at one image per call that is 60 launches
// of a UNet pass.
__global__ __launch_bounds__(256) void gn_small_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int HW, int C, int G, float eps, int silu, float* __restrict__ stats) {
  extern __shared__ uint32_t gs_pairs[];
  __shared__ float red[8];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G, ppp = cpg / 2;                 // bf16 pairs per pixel of this group
  const int npair = HW * ppp;
  const uint32_t* xin = reinterpret_cast<const uint32_t*>(x + (long)b * HW * C + g * cpg);
  float s = 0.f, q = 0.f;
  for (int i = tid; i < npair; i += 256) {
    const int pix = i / ppp, j = i - pix * ppp;
    const uint32_t u = xin[(long)pix * (C / 2) + j];
    gs_pairs[i] = u;
    const float a = st_lo(u), c = st_hi(u);
    s += a; s += c;
    q += a * a; q += c * c;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s; red[(tid >> 6) * 2 + 1] = q; }
  __syncthreads();
  s = ((red[0] + red[2]) + red[4]) + red[6];
  q = ((red[1] + red[3]) + red[5]) + red[7];
  const float n = (float)HW * (float)cpg;
  const float m = s / n;
  float var = q / n - m * m;
  var = var < 0.f ? 0.f : var;
  const float rstd = rsqrtf(var + eps);
  if (stats && tid == 0) *reinterpret_cast<float2*>(stats + ((long)b * G + g) * 2) = make_float2(m, rstd);
  uint32_t* yout = reinterpret_cast<uint32_t*>(y + (long)b * HW * C + g * cpg);
  for (int i = tid; i < npair; i += 256) {
    const int pix = i / ppp, j = i - pix * ppp;
    const uint32_t u = gs_pairs[i];
    const int c0 = g * cpg + 2 * j;
    const float sc0 = rstd * gamma[c0], sc1 = rstd * gamma[c0 + 1];
    float a = st_lo(u) * sc0 + (beta[c0] - m * sc0);
    float c = st_hi(u) * sc1 + (beta[c0 + 1] - m * sc1);
    if (silu) { a = silu_f(a); c = silu_f(c); }
    yout[(long)pix * (C / 2) + j] = pack_bf16x2(a, c);
  }
}

// (group, image) -> (sum, sum of squares) from producer-side pair statistics, see groupnorm_from_parts_launch.
// pa [B * units][npa][2], pb [B * units][npb][2] (or null): pair p of the concatenation is pair p of a, or pair p - npa of b.
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ pa, int npa, const float* __restrict__ pb, int npb,
                                                      float* __restrict__ out, int units, int ppg) {
  __shared__ float red[8];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int total = units * ppg;
  float s = 0.f, q = 0.f;
  for (int e = tid; e < total; e += 256) {
    const int u = e / ppg, j = e - u * ppg;
    const int pr = g * ppg + j;
    const long row = (long)b * units + u;
    const float2 v = pr < npa ? *reinterpret_cast<const float2*>(pa + (row * npa + pr) * 2)
                              : *reinterpret_cast<const float2*>(pb + (row * npb + (pr - npa)) * 2);
    s += v.x;
    q += v.y;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s; red[(tid >> 6) * 2 + 1] = q; }
  __syncthreads();
  if (tid == 0) {
    s = ((red[0] + red[2]) + red[4]) + red[6];
    q = ((red[1] + red[3]) + red[5]) + red[7];
    *reinterpret_cast<float2*>(out + ((long)b * gridDim.x + g) * 2) = make_float2(s, q);
  }
}

// Pixel slabs per image for the statistics pass.  A function of (HW, C) only -- never of the batch -- so that the
// fp32 summation order of an image's statistics, and with it every bit of the normalised output, is the same
// whether the image is evaluated alone or in a batch of 120 (the sampler's reconstruction invariant rests on it).
int gn_nslab(int B, int HW, int C) {
  (void)B;
  const int CV = C / 8;
  const int R = CV <= 256 ? 256 / CV : 1;
  int n = HW / (R * 8);         // >= 8 rows per thread
  // 32 slabs up to 64 x 64 pixels (the SD UNet's shapes), 64 from 128 x 128, 128 for the VAE's 512 x 512.  More is not
  // better: every block of the apply pass folds all slabs of its image, and at 256 slabs for 256 x 256 that fold cost more
  // than the statistics pass gained (8 faces: 101 -> 148 us; tools/gn_bench.py, gpurun_out/r04/gn_bench.txt)
  int cap = HW / 256;
  cap = cap < 32 ? 32 : (cap > 64 ? 64 : cap);
  if (HW >= 512 * 512) cap = 128;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return n;
}

}  // namespace

size_t groupnorm_ws_bytes(int B, int HW, int C) {
  const int nslab = gn_nslab(B, HW, C);
  return ((size_t)B * nslab * 64 * 2 + (size_t)B * C * 2) * sizeof(float);
}

int groupnorm_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, int B, int HW,
                     int C, int G, float eps, int silu, float* ws, hipStream_t st, float* stats) {
  ARG_CHECK(C % 8 == 0 && C % G == 0 && G <= 64, "groupnorm: C % 8, C % G, G <= 64");
  ARG_CHECK(C / 8 <= 256 * GN_MAXV, "groupnorm: C too large");
  // small images: one launch (a choice by (HW, C, G) only, never by the batch)
  if (HW < 1024 && (C / G) % 2 == 0 && (size_t)HW * (C / G) * 2 <= 60 * 1024) {
    hipLaunchKernelGGL(gn_small_kernel, dim3(G, B), dim3(256), (size_t)HW * (C / G) * 2, st, x, y, gamma, beta, HW, C, G, eps, silu, stats);
    LAUNCH_CHECK();
    return HEDIT_OK;
  }
  const int nslab = gn_nslab(B, HW, C);
  float* part = ws;
  float* ss = ws + (size_t)B * nslab * 64 * 2;
  const int CV = C / 8;
  const int R = CV <= 256 ? 256 / CV : 1;
  const size_t lds = (size_t)R * C * 2 * sizeof(float);
  ARG_CHECK(lds <= 64 * 1024, "groupnorm: LDS");
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nslab, B), dim3(256), lds, st, x, part, HW, C, G, nslab);
  LAUNCH_CHECK();
  // partial buffer is indexed with stride G (<=64 reserved)
  const long total_v = (long)B * HW * CV;
  const bool rows_form = CV <= 256 && HW >= 1024;      // (smaller images: too few pixels per thread row to pay for the set-up)
  if (!rows_form || stats) {       // the row-form apply folds the slabs itself; the backward pass wants (mean, rstd) written out
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, part, gamma, beta, ss, HW, C, G, nslab, eps, stats);
    LAUNCH_CHECK();
  }
  if (rows_form) {
    // enough slabs for >= ~4096 workgroups over the batch, at least 8 pixels per thread row
    int ns = (int)(4096 / (B > 0 ? B : 1)) + 1;
    const int max_ns = HW / (R * 8) > 0 ? HW / (R * 8) : 1;
    if (ns > max_ns) ns = max_ns;
    {
      // pixels per slab a multiple of the 4 R the unrolled loop takes per trip (speed only: the slabs of this pass only
      // partition the work, the statistics' slabs are gn_nslab's)
      int pix = HW / ns / (4 * R) * (4 * R);
      if (pix < 4 * R) pix = 4 * R;
      ns = (HW + pix - 1) / pix;
    }
    hipLaunchKernelGGL(gn_apply_rows_kernel, dim3(ns, B), dim3(256), 0, st, x, y, part, gamma, beta, HW, C, G, nslab, eps, silu, ns);
    LAUNCH_CHECK();
    return HEDIT_OK;
  }
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ew_grid(total_v)), dim3(256), 0, st, x, y, ss, total_v, HW, C, silu);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// GroupNorm whose statistics were taken by the PRODUCER(s) of x (gnstat.h: pair sums per 128-row unit, written by the igemm
// epilogue / the split-K reduce): gn_fold_kernel assembles (sum, sum of squares) per (image, group) from the pairs of one
// tensor or of the two halves of a skip concatenation [a | b] -- thread-sequential over (unit, pair) in a fixed order,
// butterfly over the wave, waves 0..3 in order: a function of the image alone -- in the layout of one statistics slab, and the
// row-form apply pass finishes as always (mean / rstd formulas and element-wise arithmetic unchanged).  The pass that read
// the tensor for its statistics (gn_partial_kernel) is gone; what is read instead is 1 / 32 of the tensor's bytes.
bool groupnorm_from_parts_supported(int HW, int C, int G) {
  return HW >= 1024 && HW % 128 == 0 && C % 8 == 0 && C / 8 <= 256 && C % G == 0 && (C / G) % 2 == 0 && G <= 64;
}
size_t groupnorm_from_parts_ws_bytes(int B) { return (size_t)B * 64 * 2 * sizeof(float); }

int groupnorm_from_parts_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                float eps, int silu, const float* part_a, int ca, const float* part_b, int cb, float* ws,
                                hipStream_t st) {
  ARG_CHECK(groupnorm_from_parts_supported(HW, C, G), "groupnorm_from_parts: shape");
  ARG_CHECK(part_a && ca > 0 && ca % 2 == 0 && cb >= 0 && cb % 2 == 0 && ca + cb == C && (cb == 0) == (part_b == nullptr),
            "groupnorm_from_parts: the pair statistics must cover the C channels");
  hipLaunchKernelGGL(gn_fold_kernel, dim3(G, B), dim3(256), 0, st, part_a, ca / 2, part_b, cb / 2, ws, HW / 128, C / G / 2);
  LAUNCH_CHECK();
  const int CV = C / 8, R = 256 / CV;
  int ns = (int)(4096 / (B > 0 ? B : 1)) + 1;
  const int max_ns = HW / (R * 8) > 0 ? HW / (R * 8) : 1;
  if (ns > max_ns) ns = max_ns;
  int pix = HW / ns / (4 * R) * (4 * R);
  if (pix < 4 * R) pix = 4 * R;
  ns = (HW + pix - 1) / pix;
  hipLaunchKernelGGL(gn_apply_rows_kernel, dim3(ns, B), dim3(256), 0, st, x, y, ws, gamma, beta, HW, C, G, 1, eps, silu, ns);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// statistics only: the per (image, channel) scale and shift (float2, [B][C]) a consumer applies itself (ffn.hip's
// GroupNorm'd-input chain); same partial sums and summation order as groupnorm_launch
int groupnorm_affine_launch(const bf16_t* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps,
                            float* ws, hipStream_t st, const float** ss_out) {
  ARG_CHECK(C % 8 == 0 && C % G == 0 && G <= 64, "groupnorm: C % 8, C % G, G <= 64");
  ARG_CHECK(C / 8 <= 256 * GN_MAXV, "groupnorm: C too large");
  const int nslab = gn_nslab(B, HW, C);
  float* part = ws;
  float* ss = ws + (size_t)B * nslab * 64 * 2;
  const int CV = C / 8;
  const int R = CV <= 256 ? 256 / CV : 1;
  const size_t lds = (size_t)R * C * 2 * sizeof(float);
  ARG_CHECK(lds <= 64 * 1024, "groupnorm: LDS");
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nslab, B), dim3(256), lds, st, x, part, HW, C, G, nslab);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, part, gamma, beta, ss, HW, C, G, nslab, eps, (float*)nullptr);
  LAUNCH_CHECK();
  *ss_out = ss;
  return HEDIT_OK;
}

// (one selection for both forms: the fused form must normalise with the very kernel the separate launch would use)
static int ln_narrow_lanes(int C) {
  if (C % 64 == 0 && C / 64 <= LNN_MAXV) return 8;
  if (C % 128 == 0 && C / 128 <= LNN_MAXV) return 16;     // C = 640
  if (C % 256 == 0 && C / 256 <= LNN_MAXV) return 32;     // C = 1280
  return 0;
}

bool splitk_reduce_ln_supported(int C) { return C % 8 == 0 && ln_narrow_lanes(C) != 0; }

// out = bf16(sum of slabs + bias) (+ residual), y = LayerNorm(out): see LnReduce
int splitk_reduce_ln_launch(const float* partial, int splits, const float* bias, const bf16_t* residual, int ldr, bf16_t* out,
                            bf16_t* y, const float* gamma, const float* beta, long rows, int C, float eps, hipStream_t st) {
  const int lanes = ln_narrow_lanes(C);
  ARG_CHECK(lanes != 0 && splits >= 1, "splitk_reduce_ln: unsupported row width");
  const LnReduce rd{partial, splits, bias, residual, ldr};
  if (lanes == 8) hipLaunchKernelGGL((layernorm_narrow_kernel<8, true>), dim3(cdiv(rows, 32)), dim3(256), 0, st, out, y, gamma, beta, rows, C, eps, rd);
  else if (lanes == 16) hipLaunchKernelGGL((layernorm_narrow_kernel<16, true>), dim3(cdiv(rows, 16)), dim3(256), 0, st, out, y, gamma, beta, rows, C, eps, rd);
  else hipLaunchKernelGGL((layernorm_narrow_kernel<32, true>), dim3(cdiv(rows, 8)), dim3(256), 0, st, out, y, gamma, beta, rows, C, eps, rd);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int layernorm_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, long rows, int C,
                     float eps, hipStream_t st) {
  ARG_CHECK(C % 8 == 0 && C / 8 <= 64 * LN_MAXV, "layernorm: C");
  const LnReduce none{};
  bf16_t* xx = const_cast<bf16_t*>(x);      // (only the reducing form writes it)
  switch (ln_narrow_lanes(C)) {
    case 8:      // 8 lanes per row
      hipLaunchKernelGGL((layernorm_narrow_kernel<8, false>), dim3(cdiv(rows, 32)), dim3(256), 0, st, xx, y, gamma, beta, rows, C, eps, none);
      LAUNCH_CHECK();
      return HEDIT_OK;
    case 16:     // 16 lanes per row (C = 640)
      hipLaunchKernelGGL((layernorm_narrow_kernel<16, false>), dim3(cdiv(rows, 16)), dim3(256), 0, st, xx, y, gamma, beta, rows, C, eps, none);
      LAUNCH_CHECK();
      return HEDIT_OK;
    case 32:     // 32 lanes per row (C = 1280)
      hipLaunchKernelGGL((layernorm_narrow_kernel<32, false>), dim3(cdiv(rows, 8)), dim3(256), 0, st, xx, y, gamma, beta, rows, C, eps, none);
      LAUNCH_CHECK();
      return HEDIT_OK;
    default: break;
  }
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, y, gamma, beta, rows, C, eps);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int geglu_launch(const bf16_t* x, bf16_t* y, long rows, int inner, hipStream_t st) {
  ARG_CHECK(inner % 8 == 0, "geglu: inner % 8");
  hipLaunchKernelGGL(geglu_kernel, dim3(ew_grid(rows * (inner / 8))), dim3(256), 0, st, x, y, rows, inner);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int concat_launch(const bf16_t* a, int ca, const bf16_t* b, int cb, bf16_t* y, long rows, hipStream_t st) {
  ARG_CHECK(ca % 8 == 0 && cb % 8 == 0, "concat: channels % 8");
  if (a == nullptr) {
    hipLaunchKernelGGL(concat_right_kernel, dim3(ew_grid(rows * (cb / 8))), dim3(256), 0, st, b, ca, cb, y, rows);
    LAUNCH_CHECK();
    return HEDIT_OK;
  }
  hipLaunchKernelGGL(concat_kernel, dim3(ew_grid(rows * ((ca + cb) / 8))), dim3(256), 0, st, a, ca, b, cb, y, rows);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int f32_to_bf16_launch(const float* x, bf16_t* y, long n, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, y, n, 1.0f);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int ctx_pad_launch(const float* x, bf16_t* y, int B, int dim, hipStream_t st) {
  hipLaunchKernelGGL(ctx_pad_kernel, dim3(ew_grid((long)B * HEDIT_CTXP * dim)), dim3(256), 0, st, x, y, B, dim);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int pack_linear_launch(const float* w, bf16_t* out, long n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, st, w, out, n, scale);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int pack_conv3x3_launch(const float* w, bf16_t* out, int O, int I, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(ew_grid((long)O * 9 * I)), dim3(256), 0, st, w, out, O, I);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int pack_geglu_rows_launch(const float* w, bf16_t* out_bf16, float* out_f32, int rows, int K, hipStream_t st) {
  ARG_CHECK(rows % 32 == 0, "geglu pack: rows % 32");
  hipLaunchKernelGGL(pack_geglu_rows_kernel, dim3(ew_grid((long)rows * K)), dim3(256), 0, st, w, out_bf16, out_f32, rows, K);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int gemv_launch(const bf16_t* W, const float* x, const float* b0, const float* b1, float* out, int N, int K,
                int silu, hipStream_t st) {
  ARG_CHECK(K % 8 == 0, "gemv: K % 8");
  hipLaunchKernelGGL(gemv_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, W, x, b0, b1, out, N, K, silu);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int timestep_embed_launch(float t, float* out, int dim, hipStream_t st) {
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(1), dim3(256), 0, st, t, out, dim);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// x[b] <- x[src[b]] for the rows with src[b] != b (Plug-and-Play feature injection: target rows take the source
// row's activations); rows that are read are never written (src[src[b]] == src[b])
__global__ __launch_bounds__(256) void copy_rows_kernel(bf16_t* __restrict__ x, const int* __restrict__ src, long row_v) {
  const int b = blockIdx.y;
  const int s = src[b];
  if (s == b) return;
  const uint4* from = reinterpret_cast<const uint4*>(x) + (long)s * row_v;
  uint4* to = reinterpret_cast<uint4*>(x) + (long)b * row_v;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_v; i += (long)gridDim.x * 256) to[i] = from[i];
}

int copy_rows_launch(bf16_t* x, const int* src, int B, long row_elems, hipStream_t st) {
  ARG_CHECK(row_elems % 8 == 0, "copy_rows: row size must be a multiple of 8 elements");
  const long row_v = row_elems / 8;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid(row_v) > 256 ? 256 : ew_grid(row_v), B), dim3(256), 0, st, x, src, row_v);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int timestep_embed_ddpm_launch(float t, float* out, int dim, hipStream_t st) {
  ARG_CHECK(dim >= 4, "timestep_embed_ddpm: dim >= 4");
  hipLaunchKernelGGL(timestep_embed_ddpm_kernel, dim3(1), dim3(256), 0, st, t, out, dim);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int conv_in_launch(const float* x, const float* w, const float* bias, bf16_t* y, int B, int Cin, int H, int W,
                   int Cout, hipStream_t st) {
  ARG_CHECK(Cout % 8 == 0, "conv_in: Cout % 8");
  const size_t lds = ((size_t)Cin * 9 * Cout + CI_PIX * (size_t)Cin * 9) * sizeof(float);
  ARG_CHECK(lds <= 160 * 1024, "conv_in: Cin*9*(Cout+128) floats must fit 160 KiB of LDS");
  if (lds > 64 * 1024)
    if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&conv_in_kernel), 160 * 1024)) return rc;      // (the ceiling once, per device)
  hipLaunchKernelGGL(conv_in_kernel, dim3(cdiv((long)B * H * W, CI_PIX)), dim3(256), lds, st, x, w, bias, y, B, Cin, H, W, Cout);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// fp32 products [B*HW][ld] of a small-N conv GEMM (+ bias) -> fp32 NCHW [B][Cout][HW]
__global__ __launch_bounds__(256) void rows_to_nchw_kernel(const float* __restrict__ src, const float* __restrict__ bias,
                                                           float* __restrict__ dst, long total, long HW, int ld, int Cout) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, px = i - b * HW;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * ld);
    for (int c = 0; c < Cout; ++c) dst[(b * Cout + c) * HW + px] = v[c] + (bias ? bias[c] : 0.f);
  }
}

int rows_to_nchw_launch(const float* src, const float* bias, float* dst, int B, long HW, int ld, int Cout, hipStream_t st) {
  ARG_CHECK(Cout >= 1 && Cout <= 4 && ld >= 4 && ld % 4 == 0, "rows_to_nchw: Cout <= 4, ld % 4");
  hipLaunchKernelGGL(rows_to_nchw_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, st, src, bias, dst, (long)B * HW, HW, ld, Cout);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int conv_out_launch(const bf16_t* x, const bf16_t* w, const float* bias, float* y, int B, int H, int W, int C,
                    int Cout, hipStream_t st) {
  ARG_CHECK(C % 8 == 0 && Cout <= 4, "conv_out: C % 8, Cout <= 4");
  hipLaunchKernelGGL(conv_out_kernel, dim3(cdiv((long)B * H * W, 4)), dim3(256), 0, st, x, w, bias, y, B, H, W, C, Cout);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// ------------------------------------------------------------------ VAE helpers
// softmax over the rows of an fp32 score matrix, bf16 probabilities out: one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, long rows, int N, float scale_log2e) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * N;
  float mx = -3.0e38f;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sr + i);
    mx = fmaxf(fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], v[2])), v[3]);
  }
  mx = wave_max(mx) * scale_log2e;
  float sum = 0.f;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sr + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += __builtin_amdgcn_exp2f(v[j] * scale_log2e - mx);
  }
  const float inv = 1.0f / wave_sum(sum);
  bf16_t* pr = p + row * N;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sr + i);
    uint2 o;
    o.x = pack_bf16x2(__builtin_amdgcn_exp2f(v[0] * scale_log2e - mx) * inv, __builtin_amdgcn_exp2f(v[1] * scale_log2e - mx) * inv);
    o.y = pack_bf16x2(__builtin_amdgcn_exp2f(v[2] * scale_log2e - mx) * inv, __builtin_amdgcn_exp2f(v[3] * scale_log2e - mx) * inv);
    *reinterpret_cast<uint2*>(pr + i) = o;
  }
}

// block-diagonal variant for several images stacked along both axes of one score matrix [rows][ld], rows = B*T:
// row r attends only to columns [b*T, (b+1)*T) of its own image b = r / T; every other entry of P is written 0,
// so that one P.V GEMM over K = B*T serves all images.
__global__ __launch_bounds__(256) void softmax_blockdiag_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, long rows, int T,
                                                                int ld, float scale_log2e) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c0 = (int)(row / T) * T;
  const float* sr = s + row * ld + c0;
  float mx = -3.0e38f;
  for (int i = lane * 4; i < T; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sr + i);
    mx = fmaxf(fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], v[2])), v[3]);
  }
  mx = wave_max(mx) * scale_log2e;
  float sum = 0.f;
  for (int i = lane * 4; i < T; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sr + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += __builtin_amdgcn_exp2f(v[j] * scale_log2e - mx);
  }
  const float inv = 1.0f / wave_sum(sum);
  bf16_t* pr = p + row * ld;
  for (int i = lane * 4; i < ld; i += 256) {
    uint2 o = make_uint2(0u, 0u);
    if (i >= c0 && i < c0 + T) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(s + row * ld + i);
      o.x = pack_bf16x2(__builtin_amdgcn_exp2f(v[0] * scale_log2e - mx) * inv, __builtin_amdgcn_exp2f(v[1] * scale_log2e - mx) * inv);
      o.y = pack_bf16x2(__builtin_amdgcn_exp2f(v[2] * scale_log2e - mx) * inv, __builtin_amdgcn_exp2f(v[3] * scale_log2e - mx) * inv);
    }
    *reinterpret_cast<uint2*>(pr + i) = o;
  }
}

int softmax_blockdiag_launch(const float* s, bf16_t* p, long rows, int T, int ld, float scale, hipStream_t st) {
  ARG_CHECK(T % 4 == 0 && ld % 4 == 0 && rows % T == 0, "softmax_blockdiag: T % 4, ld % 4, rows % T");
  hipLaunchKernelGGL(softmax_blockdiag_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, s, p, rows, T, ld, scale * 1.4426950408889634f);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int softmax_rows_launch(const float* s, bf16_t* p, long rows, int N, float scale, hipStream_t st) {
  ARG_CHECK(N % 4 == 0, "softmax_rows: N % 4");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, s, p, rows, N, scale * 1.4426950408889634f);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// per-pixel channel mixing of a small fp32 NCHW tensor: y[b][co] = sum_ci w[co][ci] x[b][ci] + bias[co]  (1x1 conv, C <= 8)
__global__ __launch_bounds__(256) void mix1x1_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int B, int Cin, int Cout, long HW, float pre_scale) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, px = i - b * HW;
    float xv[8];
    for (int ci = 0; ci < Cin; ++ci) xv[ci] = x[(b * Cin + ci) * HW + px] * pre_scale;
    for (int co = 0; co < Cout; ++co) {
      float a = bias ? bias[co] : 0.f;
      for (int ci = 0; ci < Cin; ++ci) a += w[co * Cin + ci] * xv[ci];
      y[(b * Cout + co) * HW + px] = a;
    }
  }
}

int mix1x1_nchw_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, long HW,
                       float pre_scale, hipStream_t st) {
  ARG_CHECK(Cin <= 8 && Cout <= 8, "mix1x1: at most 8 channels");
  hipLaunchKernelGGL(mix1x1_nchw_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, st, x, w, bias, y, B, Cin, Cout, HW, pre_scale);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// encoder tail: moments h [M][Cm] (bf16 NHWC) -> first Cout channels of quant_conv(h), fp32 NCHW
__global__ __launch_bounds__(256) void quant_mean_kernel(const bf16_t* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, int B, long HW, int Cm, int Cout) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, px = i - b * HW;
    float hv[16];
    for (int c = 0; c < Cm; ++c) hv[c] = bf16_to_f32(h[i * Cm + c]);
    for (int co = 0; co < Cout; ++co) {
      float a = bias[co];
      for (int c = 0; c < Cm; ++c) a += w[co * Cm + c] * hv[c];
      y[(b * Cout + co) * HW + px] = a;
    }
  }
}

int quant_mean_launch(const bf16_t* h, const float* w, const float* bias, float* y, int B, long HW, int Cm, int Cout, hipStream_t st) {
  ARG_CHECK(Cm <= 16 && Cout <= Cm, "quant_mean: channels");
  hipLaunchKernelGGL(quant_mean_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, st, h, w, bias, y, B, HW, Cm, Cout);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
