// Loop-level kernels of the h-Edit sampler: everything the reference does between two UNet calls
// with ~15 tiny torch ops and two .item() host syncs (text-guided/inversion/p2p_h_edit.py:616-699,
// inversion_utils.py:84-119, p2p/ptp_classes.py:44-72) as three launch-bound kernels with no host
// round trip.  fp32 NCHW latents, one workgroup per image (the per-image RMS norms of the
// reconstruction pull are block reductions).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// mu(x, e) = sqrt(ab_prev) * (x - sqrt(1-ab_t) e) / sqrt(ab_t) + dir * e   with e = e_u + w_src (e_c - e_u).
// ONE definition for the sampler's base step and for the inversion that produced its noise maps: the
// reconstruction branch of the loop then retraces the inverted trajectory bit for bit (same operations in the
// same order on the same eps; this unit is built with -ffp-contract=off, so nothing is re-associated).
__device__ __forceinline__ float ddpm_mu(float x, float eu, float ec, const StepCoef& c) {
  const float en = eu + c.w_src * (ec - eu);
  const float x0 = (x - c.sqrt_1m_ab_t * en) / c.sqrt_ab_t;
  return c.sqrt_ab_prev * x0 + c.dir_coef * en;
}

// x_prev[j] = mu(x[j], e) + noise * z.  Batched tensors are [row][image][elems] ("row-major over
// images", identical to the reference layout for one image):
//   rows == 4: P2P layout [x_o|0, x_e|0, x_o|src, x_e|src];
//   rows == 2: no-P2P layout [x_e|0, x_e|src], the same eps drives both branches;
//   xt / x_prev: [2][n_img][elems] = (x^orig, x^edit).
__global__ __launch_bounds__(256) void step_base_kernel(const float* __restrict__ eps, const float* __restrict__ xt,
                                                        const float* __restrict__ z, float* __restrict__ x_prev,
                                                        int n_img, int elems, int rows, StepCoef c) {
  const int img = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < elems; i += gridDim.x * 256) {
    const float zz = z ? z[(long)img * elems + i] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ru = rows == 4 ? j : 0, rc = rows == 4 ? 2 + j : 1;
      const float eu = eps[((long)ru * n_img + img) * elems + i];
      const float ec = eps[((long)rc * n_img + img) * elems + i];
      const long xi = ((long)j * n_img + img) * elems + i;
      float pv = ddpm_mu(xt[xi], eu, ec, c);
      pv = pv + c.noise_coef * zz;
      x_prev[xi] = pv;
    }
  }
}

// One step of the edit-friendly DDPM inversion (text-guided/inversion/ddpm_inversion.py:146-162):
//   z = (x_prev - mu(x_t, e)) / sigma ;  x_prev <- mu + sigma z   (the reference's in-place rewrite of xts[idx]).
// e_u / e_c: [n_img][elems] each (e_c == e_u for an unconditional inversion); sigma = c.noise_coef.
__global__ __launch_bounds__(256) void step_invert_kernel(const float* __restrict__ e_u, const float* __restrict__ e_c,
                                                          const float* __restrict__ xt, float* __restrict__ x_prev,
                                                          float* __restrict__ z_out, int elems, StepCoef c) {
  const int img = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < elems; i += gridDim.x * 256) {
    const long xi = (long)img * elems + i;
    const float mu = ddpm_mu(xt[xi], e_u[xi], e_c[xi], c);
    const float zz = (x_prev[xi] - mu) / c.noise_coef;
    z_out[xi] = zz;
    float pv = mu;
    pv = pv + c.noise_coef * zz;
    x_prev[xi] = pv;
  }
}

// x_out = rec + coeff * corr,  corr = e_tar - e_hat;  k > 0: rec = x_k - rho * sign(x_k - x_base)/N
__global__ __launch_bounds__(1024) void step_update_kernel(const float* __restrict__ e_u_src, const float* __restrict__ e_c_src,
                                                           const float* __restrict__ e_u_tar, const float* __restrict__ e_c_tar,
                                                           long stride_img, const float* __restrict__ x_k,
                                                           const float* __restrict__ x_base, float* __restrict__ x_out,
                                                           int elems, int k_gt0, StepCoef c) {
  __shared__ float red[16];
  const int img = blockIdx.x;
  const long eo = (long)img * stride_img;
  const float* xk = x_k + (long)img * elems;
  const float* xb = x_base + (long)img * elems;
  float* xo = x_out + (long)img * elems;
  float rho = 0.f;
  const float invn = 1.0f / (float)elems;
  if (k_gt0) {
    float sc = 0.f, sg = 0.f;
    for (int i = threadIdx.x; i < elems; i += blockDim.x) {
      const float ehat = e_u_src[eo + i] + c.w_hat * (e_c_src[eo + i] - e_u_src[eo + i]);
      const float etar = e_u_tar[eo + i] + c.w_tar * (e_c_tar[eo + i] - e_u_tar[eo + i]);
      const float corr = etar - ehat;
      sc += corr * corr;
      const float d = xk[i] - xb[i];
      const float g = (d > 0.f ? invn : (d < 0.f ? -invn : 0.f));
      sg += g * g;
    }
    sc = block_sum(sc, red);
    sg = block_sum(sg, red);
    const float nc = sqrtf(sc * invn), ng = sqrtf(sg * invn);
    rho = nc / (ng + 1e-8f) * c.w_rec;
  }
  for (int i = threadIdx.x; i < elems; i += blockDim.x) {
    const float ehat = e_u_src[eo + i] + c.w_hat * (e_c_src[eo + i] - e_u_src[eo + i]);
    const float etar = e_u_tar[eo + i] + c.w_tar * (e_c_tar[eo + i] - e_u_tar[eo + i]);
    const float corr = etar - ehat;
    float rec = xk[i];
    if (k_gt0) {
      const float d = xk[i] - xb[i];
      const float g = (d > 0.f ? invn : (d < 0.f ? -invn : 0.f));
      rec = rec - rho * g;
    }
    xo[i] = rec + c.coeff * corr;
  }
}

struct BlendMaps { const float* p[8]; };

// LocalBlend: word-selected mean of the 16x16 cross maps -> 3x3 max-pool -> nearest upsample ->
// /max -> threshold -> OR(src, tar) -> x_e = x_o + mask (x_e - x_o).  One workgroup per image.
// alpha_sub (optional): the substruct_words layers of LocalBlend (ptp_classes.py:28-38,64-68) -- a second word mask, not
// pooled, thresholded with th_sub, whose complement is multiplied into the blend mask.
__global__ __launch_bounds__(256) void local_blend_kernel(BlendMaps maps, int n_maps, int heads, const float* __restrict__ alpha,
                                                          const float* __restrict__ alpha_sub, const int* __restrict__ enabled,
                                                          float* __restrict__ xt, int n_img, int C, int H, int W, float th,
                                                          float th_sub) {
  __shared__ float m[2][256];
  __shared__ float pooled[2][256];
  __shared__ float sub[2][256];
  __shared__ float red[4];
  __shared__ float mxs[2], mxsub[2];
  const int img = blockIdx.x, p = threadIdx.x;
  if (enabled && !enabled[img]) return;     // this image has no LocalBlend
  auto word_mean = [&](const float* al, int s) {
    float acc = 0.f;
    for (int l = 0; l < n_maps; ++l) {
      const float* base = maps.p[l] + (((long)img * 2 + s) * heads) * 256 * HEDIT_MAXW;
      for (int n = 0; n < HEDIT_MAXW; ++n) {
        const float a = al[s * HEDIT_MAXW + n];
        if (a == 0.f) continue;
        for (int h = 0; h < heads; ++h) acc += base[((long)h * 256 + p) * HEDIT_MAXW + n] * a;
      }
    }
    return acc / (float)(n_maps * heads);
  };
  for (int s = 0; s < 2; ++s) {
    m[s][p] = word_mean(alpha + (long)img * 2 * HEDIT_MAXW, s);
    sub[s][p] = alpha_sub ? word_mean(alpha_sub + (long)img * 2 * HEDIT_MAXW, s) : 0.f;
  }
  __syncthreads();
  const int py = p >> 4, px = p & 15;
  for (int s = 0; s < 2; ++s) {
    float v = -3.4e38f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = py + dy, xx = px + dx;
        if ((unsigned)yy < 16u && (unsigned)xx < 16u) v = fmaxf(v, m[s][yy * 16 + xx]);
      }
    pooled[s][p] = v;
  }
  __syncthreads();
  for (int s = 0; s < 2; ++s) {
    float v = wave_max(pooled[s][p]);
    if ((p & 63) == 0) red[p >> 6] = v;
    __syncthreads();
    if (p == 0) mxs[s] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    if (alpha_sub) {
      v = wave_max(sub[s][p]);
      if ((p & 63) == 0) red[p >> 6] = v;
      __syncthreads();
      if (p == 0) mxsub[s] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      __syncthreads();
    }
  }
  float* xo = xt + (long)img * C * H * W;            // xt is [2][n_img][C][H][W]
  float* xe = xt + ((long)n_img + img) * C * H * W;
  for (int i = p; i < H * W; i += 256) {
    const int y = i / W, x = i - y * W;
    const int sp = ((y * 16) / H) * 16 + (x * 16) / W;
    bool on = (pooled[0][sp] / mxs[0] > th) || (pooled[1][sp] / mxs[1] > th);
    // (an all-zero layer -- no substruct words for this prompt -- is 0 / 0 in the reference: NaN > th is false)
    if (alpha_sub)
      on = on && !((mxsub[0] > 0.f && sub[0][sp] / mxsub[0] > th_sub) || (mxsub[1] > 0.f && sub[1][sp] / mxsub[1] > th_sub));
    if (!on) {
      for (int ch = 0; ch < C; ++ch) xe[(long)ch * H * W + i] = xo[(long)ch * H * W + i];
    } else {
      // x_o + 1.0 * (x_e - x_o): keep the reference's rounding
      for (int ch = 0; ch < C; ++ch) {
        const float a = xo[(long)ch * H * W + i];
        xe[(long)ch * H * W + i] = a + (xe[(long)ch * H * W + i] - a);
      }
    }
  }
}

// ---- style guidance (text-guided-n-style/inversion/h_edit.py:160-185) around the decoder / image-encoder pass
// z0 = (x - sqrt(1-ab) e_tar) / sqrt(ab) * inv_scale : Tweedie x0 at t-1, pre-divided by the VAE scaling factor
__global__ __launch_bounds__(256) void step_tweedie_kernel(const float* __restrict__ e_u_tar, const float* __restrict__ e_c_tar,
                                                           long stride_img, const float* __restrict__ x, float* __restrict__ z0,
                                                           int elems, float w_tar, float sqrt_ab, float sqrt_1m_ab, float inv_scale) {
  const int img = blockIdx.y;
  const long eo = (long)img * stride_img;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < elems; i += gridDim.x * 256) {
    const float etar = e_u_tar[eo + i] + w_tar * (e_c_tar[eo + i] - e_u_tar[eo + i]);
    z0[(long)img * elems + i] = (x[(long)img * elems + i] - sqrt_1m_ab * etar) / sqrt_ab * inv_scale;
  }
}

// x_out = x - rho g,  g = chain * g_z (chain rule through the Tweedie / scaling step),
// rho = rms(correction) / rms(g) * weight -- per image, no epsilon, as the reference writes it
__global__ __launch_bounds__(1024) void step_style_kernel(const float* __restrict__ e_u_src, const float* __restrict__ e_c_src,
                                                          const float* __restrict__ e_u_tar, const float* __restrict__ e_c_tar,
                                                          long stride_img, const float* __restrict__ x, const float* __restrict__ g_z,
                                                          float* __restrict__ x_out, int elems, float w_hat, float w_tar,
                                                          float chain, float weight) {
  __shared__ float red[16];
  const int img = blockIdx.x;
  const long eo = (long)img * stride_img;
  const float* xi = x + (long)img * elems;
  const float* gi = g_z + (long)img * elems;
  float sc = 0.f, sg = 0.f;
  for (int i = threadIdx.x; i < elems; i += blockDim.x) {
    const float ehat = e_u_src[eo + i] + w_hat * (e_c_src[eo + i] - e_u_src[eo + i]);
    const float etar = e_u_tar[eo + i] + w_tar * (e_c_tar[eo + i] - e_u_tar[eo + i]);
    const float corr = etar - ehat;
    sc += corr * corr;
    const float g = gi[i] * chain;
    sg += g * g;
  }
  sc = block_sum(sc, red);
  sg = block_sum(sg, red);
  const float invn = 1.0f / (float)elems;
  const float rho = sqrtf(sc * invn) / sqrtf(sg * invn) * weight;
  for (int i = threadIdx.x; i < elems; i += blockDim.x) x_out[(long)img * elems + i] = xi[i] - rho * (gi[i] * chain);
}

}  // namespace

int step_tweedie_launch(const float* e_u_tar, const float* e_c_tar, long stride_img, const float* x, float* z0,
                        int n_img, int elems, float w_tar, float sqrt_ab, float sqrt_1m_ab, float inv_scale, hipStream_t st) {
  dim3 grid(cdiv(elems, 256) > 64 ? 64 : cdiv(elems, 256), n_img);
  hipLaunchKernelGGL(step_tweedie_kernel, grid, dim3(256), 0, st, e_u_tar, e_c_tar, stride_img, x, z0, elems, w_tar, sqrt_ab,
                     sqrt_1m_ab, inv_scale);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int step_style_launch(const float* e_u_src, const float* e_c_src, const float* e_u_tar, const float* e_c_tar,
                      long stride_img, const float* x, const float* g_z, float* x_out, int n_img, int elems, float w_hat,
                      float w_tar, float chain, float weight, hipStream_t st) {
  hipLaunchKernelGGL(step_style_kernel, dim3(n_img), dim3(1024), 0, st, e_u_src, e_c_src, e_u_tar, e_c_tar, stride_img, x,
                     g_z, x_out, elems, w_hat, w_tar, chain, weight);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int step_base_launch(const float* eps, const float* xt, const float* z, float* x_prev, int n_img, int elems,
                     int rows, StepCoef c, hipStream_t st) {
  ARG_CHECK(rows == 4 || rows == 2, "step_base: eps rows per image must be 4 (P2P) or 2");
  dim3 grid(cdiv(elems, 256) > 64 ? 64 : cdiv(elems, 256), n_img);
  hipLaunchKernelGGL(step_base_kernel, grid, dim3(256), 0, st, eps, xt, z, x_prev, n_img, elems, rows, c);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int step_invert_launch(const float* e_u, const float* e_c, const float* xt, float* x_prev, float* z_out, int n_img,
                       int elems, StepCoef c, hipStream_t st) {
  ARG_CHECK(c.noise_coef > 0.f, "step_invert: sigma must be positive (eta > 0)");
  dim3 grid(cdiv(elems, 256) > 64 ? 64 : cdiv(elems, 256), n_img);
  hipLaunchKernelGGL(step_invert_kernel, grid, dim3(256), 0, st, e_u, e_c, xt, x_prev, z_out, elems, c);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int step_update_launch(const float* e_u_src, const float* e_c_src, const float* e_u_tar, const float* e_c_tar,
                       long stride_img, const float* x_k, const float* x_base, float* x_out, int n_img, int elems,
                       int k_gt0, StepCoef c, hipStream_t st) {
  hipLaunchKernelGGL(step_update_kernel, dim3(n_img), dim3(1024), 0, st, e_u_src, e_c_src, e_u_tar, e_c_tar,
                     stride_img, x_k, x_base, x_out, elems, k_gt0, c);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int local_blend_launch(const float* const* maps, int n_maps, int heads, const float* alpha_layers, const float* alpha_sub,
                       const int* enabled, float* xt, int n_img, int C, int H, int W, float th, float th_sub, hipStream_t st) {
  ARG_CHECK(n_maps >= 1 && n_maps <= 8, "local_blend: 1..8 maps");
  BlendMaps bm;
  for (int i = 0; i < 8; ++i) bm.p[i] = i < n_maps ? maps[i] : nullptr;
  hipLaunchKernelGGL(local_blend_kernel, dim3(n_img), dim3(256), 0, st, bm, n_maps, heads, alpha_layers, alpha_sub, enabled, xt, n_img, C, H,
                     W, th, th_sub);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

// ------------------------------------------------------------------ axis mix (bicubic resize of the style closure)
// out[o][i][x] = sum_j val[i][j] * in[o][idx[i][j]][x],  j < nnz in table order: a sparse linear map along one axis of
// a [outer][n][inner] fp32 tensor.  With the bicubic weights of F.interpolate as the table it is the resize in front
// of the style encoder (clip_guidance/base_clip.py:57-58), with the transposed table its backward -- one thread per
// output element and a fixed summation order, so forward and backward are repeatable and batch-invariant bit for bit
// (torch's bicubic backward accumulates with atomics).
__global__ __launch_bounds__(256) void axis_mix_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ idx,
                                                       const float* __restrict__ val, int nnz, long outer, int n_in, int n_out, int inner) {
  const long total = outer * n_out * inner;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int x = (int)(e % inner);
    const long r = e / inner;
    const int i = (int)(r % n_out);
    const long o = r / n_out;
    const float* src = in + o * n_in * inner + x;
    float acc = 0.f;
    for (int j = 0; j < nnz; ++j) acc += val[i * nnz + j] * src[(long)idx[i * nnz + j] * inner];
    out[e] = acc;
  }
}

int axis_mix_launch(const float* in, float* out, const int* idx, const float* val, int nnz, long outer, int n_in, int n_out, int inner,
                    hipStream_t st) {
  ARG_CHECK(nnz >= 1 && outer >= 1 && n_in >= 1 && n_out >= 1 && inner >= 1, "axis_mix: shape");
  hipLaunchKernelGGL(axis_mix_kernel, dim3(ew_grid(outer * n_out * inner)), dim3(256), 0, st, in, out, idx, val, nnz, outer, n_in, n_out, inner);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
