// Input-gradient (vector-Jacobian) kernels of the image decoder: what the style-guidance closure
// of the reference differentiates through (`torch.autograd.grad(loss, xt_prev_opt_style)` with
// x0 -> vae.decode -> CLIP features, text-guided-n-style/inversion/h_edit.py:146-185; SURVEY.md
// section 8 rows a17 / a20).  Only d(image)/d(latent) is needed -- never a weight gradient -- so the
// convolutions' backward is again a forward conv (taps flipped, channels swapped: the same MFMA
// implicit GEMM with a re-packed weight) and the rest is HBM-bound elementwise / reduction work:
// GroupNorm(+SiLU) backward, softmax backward, 2x2 block sums, tile transposes.  NHWC bf16
// gradients, fp32 statistics, deterministic two-stage reductions as in norm.hip.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float silu_grad(float z) {
  const float s = 1.0f / (1.0f + __expf(-z));
  return s * (1.0f + z * (1.0f - s));
}

// ------------------------------------------------------------------ GroupNorm(+SiLU) backward
// forward: xh = (x - mu_g) r_g ; z = gamma xh + beta ; y = silu(z) | z
// backward: t = dz gamma ; dx = r_g (t - mean_g(t) - xh mean_g(t xh))
// stage 1: per (batch, pixel slab) partial sums of t and t*xh per group
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ stats, float* __restrict__ part,
                                                             int HW, int C, int G, int nslab, int silu) {
  extern __shared__ float lds[];   // [R][C][2]
  const int CV = C / 8;
  const int slab = blockIdx.x, b = blockIdx.y;
  const int pix_per = (HW + nslab - 1) / nslab;
  const int p0 = slab * pix_per;
  int p1 = p0 + pix_per;
  if (p1 > HW) p1 = HW;
  const int R = 256 / CV;
  const int r = threadIdx.x / CV, cv = threadIdx.x % CV;
  const int cpg = C / G;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (r < R) {
    float ga[8], be[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv * 8 + j;
      ga[j] = gamma[c];
      be[j] = beta[c];
      const float2 st = *reinterpret_cast<const float2*>(stats + ((long)b * G + c / cpg) * 2);
      mu[j] = st.x;
      rs[j] = st.y;
    }
    const long base = (long)b * HW * C;
    for (int p = p0 + r; p < p1; p += R) {
      float xf[8], df[8];
      unpack8(*reinterpret_cast<const uint4*>(x + base + (long)p * C + cv * 8), xf);
      unpack8(*reinterpret_cast<const uint4*>(dy + base + (long)p * C + cv * 8), df);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xf[j] - mu[j]) * rs[j];
        float dz = df[j];
        if (silu) dz *= silu_grad(ga[j] * xh + be[j]);
        const float t = dz * ga[j];
        s[j] += t;
        q[j] += t * xh;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lds[((long)r * C + cv * 8 + j) * 2 + 0] = s[j];
      lds[((long)r * C + cv * 8 + j) * 2 + 1] = q[j];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x;
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

// stage 2: fold the slabs in a fixed order -> m1 = mean_g(t), m2 = mean_g(t xh) per (batch, group), then
// the per-(batch, channel) coefficients of stage 3.  With A = r gamma, D = beta - mu A (so z = A x + D):
//   dx = A dz - B x + E,   B = r^2 m2,   E = mu r^2 m2 - r m1
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ stats,
                                                              float* __restrict__ coef, int HW, int C, int G, int nslab) {
  __shared__ float m1s[64], m2s[64];
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
      m1s[g] = s / n;
      m2s[g] = q / n;
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float2 st = *reinterpret_cast<const float2*>(stats + ((long)b * G + g) * 2);
    const float mu = st.x, r = st.y;
    const float A = r * gamma[c];
    const float Bc = r * r * m2s[g];
    *reinterpret_cast<float4*>(coef + ((long)b * C + c) * 4) = make_float4(A, Bc, mu * Bc - r * m1s[g], beta[c] - mu * A);
  }
}

// stage 3: dx = A dz - B x + E (+ add), dz = dy * silu'(A x + D).  Same thread layout as stage 1: a thread
// keeps the coefficients of its 8 channels in registers over the pixels of its slab.
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           const bf16_t* __restrict__ add, bf16_t* __restrict__ dx,
                                                           const float* __restrict__ coef, int HW, int C, int nslab, int silu) {
  const int CV = C / 8;
  const int slab = blockIdx.x, b = blockIdx.y;
  const int pix_per = (HW + nslab - 1) / nslab;
  const int p0 = slab * pix_per;
  int p1 = p0 + pix_per;
  if (p1 > HW) p1 = HW;
  const int R = 256 / CV;
  const int r = threadIdx.x / CV, cv = threadIdx.x % CV;
  if (r >= R) return;
  float4 k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = *reinterpret_cast<const float4*>(coef + ((long)b * C + cv * 8 + j) * 4);
  const long base = (long)b * HW * C + cv * 8;
  for (int p = p0 + r; p < p1; p += R) {
    const long o = base + (long)p * C;
    float xf[8], df[8], af[8], out[8];
    unpack8(*reinterpret_cast<const uint4*>(x + o), xf);
    unpack8(*reinterpret_cast<const uint4*>(dy + o), df);
    if (add) unpack8(*reinterpret_cast<const uint4*>(add + o), af);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dz = df[j];
      if (silu) dz *= silu_grad(k[j].x * xf[j] + k[j].w);
      float v = k[j].x * dz - k[j].y * xf[j] + k[j].z;
      if (add) v += af[j];
      out[j] = v;
    }
    *reinterpret_cast<uint4*>(dx + o) = pack8(out);
  }
}

// slabs per image: a function of (HW, C) only, like gn_nslab -- the gradient of an image does not depend on the batch
int gb_nslab(int B, int HW, int C) {
  (void)B;
  const int R = 256 / (C / 8);
  int n = HW / (R * 8);
  int cap = HW / 1024;
  cap = cap < 32 ? 32 : (cap > 128 ? 128 : cap);
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return n;
}

// ------------------------------------------------------------------ softmax backward, one wave per row
// ds = scale * p * (dp - sum_j dp_j p_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ p, const float* __restrict__ dp,
                                                          bf16_t* __restrict__ ds, long rows, int N, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* pr = p + row * N;
  const float* dr = dp + row * N;
  float dot = 0.f;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(dr + i);
    const uint2 u = *reinterpret_cast<const uint2*>(pr + i);
    dot += d[0] * bf16_to_f32((bf16_t)(u.x & 0xffff)) + d[1] * bf16_to_f32((bf16_t)(u.x >> 16)) +
           d[2] * bf16_to_f32((bf16_t)(u.y & 0xffff)) + d[3] * bf16_to_f32((bf16_t)(u.y >> 16));
  }
  dot = wave_sum(dot);
  bf16_t* out = ds + row * N;
  for (int i = lane * 4; i < N; i += 256) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(dr + i);
    const uint2 u = *reinterpret_cast<const uint2*>(pr + i);
    uint2 o;
    o.x = pack_bf16x2(scale * bf16_to_f32((bf16_t)(u.x & 0xffff)) * (d[0] - dot), scale * bf16_to_f32((bf16_t)(u.x >> 16)) * (d[1] - dot));
    o.y = pack_bf16x2(scale * bf16_to_f32((bf16_t)(u.y & 0xffff)) * (d[2] - dot), scale * bf16_to_f32((bf16_t)(u.y >> 16)) * (d[3] - dot));
    *reinterpret_cast<uint2*>(out + i) = o;
  }
}

// ------------------------------------------------------------------ bf16 matrix transpose, 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int C) {
  __shared__ bf16_t tile[64][72];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int r = i >> 3, v = i & 7;
    *reinterpret_cast<uint4*>(&tile[r][v * 8]) = *reinterpret_cast<const uint4*>(src + (long)(r0 + r) * C + c0 + v * 8);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int c = i >> 3, v = i & 7;      // output row c0 + c, columns r0 + v*8 ..
    bf16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = tile[v * 8 + j][c];
    uint4 o;
    o.x = (unsigned)e[0] | ((unsigned)e[1] << 16);
    o.y = (unsigned)e[2] | ((unsigned)e[3] << 16);
    o.z = (unsigned)e[4] | ((unsigned)e[5] << 16);
    o.w = (unsigned)e[6] | ((unsigned)e[7] << 16);
    *reinterpret_cast<uint4*>(dst + (long)(c0 + c) * R + r0 + v * 8) = o;
  }
}

// ------------------------------------------------------------------ nearest-2x upsample backward: 2x2 block sums
__global__ __launch_bounds__(256) void sum2x2_kernel(const bf16_t* __restrict__ du, bf16_t* __restrict__ dx, long total_v,
                                                     int H, int W, int C) {
  const int CV = C / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_v; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const long pix = i / CV;
    const int xx = (int)(pix % W);
    const int yy = (int)((pix / W) % H);
    const long b = pix / ((long)W * H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long src = ((b * 2 * H + 2 * yy + (t >> 1)) * 2 * W + 2 * xx + (t & 1)) * C + cv * 8;
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(du + src), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8(acc);
  }
}

// ------------------------------------------------------------------ weight re-packs for the input-gradient GEMMs
// conv3x3 OIHW fp32 -> bf16 [I][9][O] with the taps flipped: the dgrad conv's weight
__global__ __launch_bounds__(256) void pack_conv3x3_dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I) {
  const long total = (long)I * 9 * O;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int o = (int)(idx % O);
    const long t = idx / O;
    const int tap = (int)(t % 9);
    const int i = (int)(t / 9);
    out[idx] = f32_to_bf16(w[((long)o * I + i) * 9 + (8 - tap)]);
  }
}
// [O][I] fp32 -> bf16 [I][O]
__global__ __launch_bounds__(256) void pack_linear_t_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int O, int I) {
  const long total = (long)O * I;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int o = (int)(idx % O);
    const int i = (int)(idx / O);
    out[idx] = f32_to_bf16(w[(long)o * I + i]);
  }
}
// OIHW fp32 (k x k) -> IOHW fp32 with the taps flipped (k = 1: a plain transpose)
__global__ __launch_bounds__(256) void flip_oihw_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int kk) {
  const long total = (long)O * I * kk;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int tap = (int)(idx % kk);
    const long t = idx / kk;
    const int o = (int)(t % O);
    const int i = (int)(t / O);
    out[idx] = w[((long)o * I + i) * kk + (kk - 1 - tap)];
  }
}

}  // namespace

size_t groupnorm_bwd_ws_bytes(int B, int HW, int C) {
  return ((size_t)B * gb_nslab(B, HW, C) * 64 * 2 + (size_t)B * C * 4) * sizeof(float);
}

int groupnorm_bwd_launch(const bf16_t* x, const bf16_t* dy, const bf16_t* add, bf16_t* dx, const float* gamma,
                         const float* beta, const float* stats, int B, int HW, int C, int G, int silu, float* ws,
                         hipStream_t st) {
  ARG_CHECK(C % 8 == 0 && C % G == 0 && G <= 64 && C / 8 <= 256 && 256 % (C / 8) == 0,
            "groupnorm_bwd: C % 8, C % G, G <= 64, C/8 a divisor of 256");
  const int nslab = gb_nslab(B, HW, C);
  float* part = ws;
  float* coef = ws + (size_t)B * nslab * 64 * 2;
  const int R = 256 / (C / 8);
  const size_t lds = (size_t)R * C * 2 * sizeof(float);
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nslab, B), dim3(256), lds, st, x, dy, gamma, beta, stats, part, HW, C, G,
                     nslab, silu);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, st, part, gamma, beta, stats, coef, HW, C, G, nslab);
  LAUNCH_CHECK();
  // stage 3 wants more, smaller slabs than the reduction: ~16 pixel rows per thread
  int na = HW / (R * 16);
  if (na > 2048) na = 2048;
  if (na < 1) na = 1;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(na, B), dim3(256), 0, st, x, dy, add, dx, coef, HW, C, na, silu);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int softmax_bwd_launch(const bf16_t* p, const float* dp, bf16_t* ds, long rows, int N, float scale, hipStream_t st) {
  ARG_CHECK(N % 4 == 0, "softmax_bwd: N % 4");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, p, dp, ds, rows, N, scale);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int transpose_bf16_launch(const bf16_t* src, bf16_t* dst, int R, int C, hipStream_t st) {
  ARG_CHECK(R % 64 == 0 && C % 64 == 0, "transpose: rows and columns must be multiples of 64");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(C / 64, R / 64), dim3(256), 0, st, src, dst, R, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int sum2x2_launch(const bf16_t* du, bf16_t* dx, int B, int H, int W, int C, hipStream_t st) {
  ARG_CHECK(C % 8 == 0, "sum2x2: C % 8");
  const long total_v = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(sum2x2_kernel, dim3(ew_grid(total_v)), dim3(256), 0, st, du, dx, total_v, H, W, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int pack_conv3x3_dgrad_launch(const float* w_oihw, bf16_t* out, int O, int I, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv3x3_dgrad_kernel, dim3(ew_grid((long)O * I * 9)), dim3(256), 0, st, w_oihw, out, O, I);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int pack_linear_t_launch(const float* w, bf16_t* out, int O, int I, hipStream_t st) {
  hipLaunchKernelGGL(pack_linear_t_kernel, dim3(ew_grid((long)O * I)), dim3(256), 0, st, w, out, O, I);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int flip_oihw_launch(const float* w, float* out, int O, int I, int k, hipStream_t st) {
  hipLaunchKernelGGL(flip_oihw_kernel, dim3(ew_grid((long)O * I * k * k)), dim3(256), 0, st, w, out, O, I, k * k);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
