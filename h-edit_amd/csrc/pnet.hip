// "Precise" building blocks for the reward networks of the guidance closures (ArcFace IR-SE50 identity
// reward, LPIPS-VGG16): networks whose image gradient has to stay at fp32 quality through 50 layers --
// storing activations or weights in bf16 moves that gradient by 7 %, far outside what the
// sampler tolerates.  gfx950 has no TF32-class MFMA, and the exact f32 MFMA runs at 1/16 of the bf16
// rate, so the contraction is done as THREE bf16 MFMA products with fp32 accumulation:
//
//     a = a_hi + a_lo,  w = w_hi + w_lo   (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits kept)
//     a.w ~= a_hi.w_hi + a_hi.w_lo + a_lo.w_hi          (the dropped a_lo.w_lo is 2^-16 relative)
//
// laid out along K so that ONE launch of the tuned implicit-GEMM kernel (gemm.hip) computes it:
// activations are written as channel triples [hi | hi | lo], weights as [hi | lo | hi], the fp32
// products come back through the kernel's raw-fp32 output.  Activations live in HBM as fp32 NHWC; the
// `split3` pass that produces the GEMM operand is also where every per-element operation of the
// networks is fused (BatchNorm affine, PReLU / ReLU and their gradients, squeeze-excitation scaling,
// strided sub-sampling, zero-stuffing for the gradient of a stride-2 convolution).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ bf16_t bf16_rn(float f) { return f32_to_bf16_always(f); }   // bfloat16 in either build (common.h)

// one thread per output element (row, j) of the bf16 operand [rows_out][Kp]; j -> (part, channel)
__global__ __launch_bounds__(256) void split3_kernel(Split3Params s, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / s.Kp;
    const int j = (int)(idx - row * s.Kp);
    const int part = j / s.Cs;
    const int c = j - part * s.Cs;
    bf16_t o = 0;
    if (part < 3 && c < s.C) {
      // output pixel -> input pixel
      long in_row = row;
      bool live = true;
      int b = 0;
      if (s.geo != 0 || s.pq_img) {
        const int Ho = s.geo == 1 ? s.H / 2 : (s.geo == 2 ? s.H * 2 : s.H);
        const int Wo = s.geo == 1 ? s.W / 2 : (s.geo == 2 ? s.W * 2 : s.W);
        b = (int)(row / ((long)Ho * Wo));
        const int r = (int)(row - (long)b * Ho * Wo);
        const int oy = r / Wo, ox = r - oy * Wo;
        if (s.geo == 1) {
          in_row = ((long)b * s.H + oy * 2) * s.W + ox * 2;
        } else if (s.geo == 2) {
          live = ((oy | ox) & 1) == 0;
          in_row = ((long)b * s.H + (oy >> 1)) * s.W + (ox >> 1);
        }
      }
      float v = 0.f;
      if (live) {
        const float x = s.x[in_row * s.ldx + c];
        const int pi = s.pq_img ? b * s.C + c : c;
        switch (s.op) {
          case P_COPY: v = x; break;
          case P_AFFINE: v = s.p[pi] * x + s.q[pi]; break;
          case P_PRELU: { const float t = x + (s.q ? s.q[c] : 0.f); v = t > 0.f ? t : s.p[c] * t; break; }
          case P_PRELU_GRAD: { const float t = s.z[in_row * s.ldx + c] + (s.q ? s.q[c] : 0.f); v = x * (t > 0.f ? 1.f : s.p[c]); break; }
          case P_RELU: { const float t = x + (s.q ? s.q[c] : 0.f); v = t > 0.f ? t : 0.f; break; }
          case P_RELU_GRAD: { const float t = s.z[in_row * s.ldx + c] + (s.q ? s.q[c] : 0.f); v = t > 0.f ? x : 0.f; break; }
          case P_QGELU: { const float t = x + (s.q ? s.q[c] : 0.f); v = t / (1.0f + __expf(-1.702f * t)); break; }
          case P_QGELU_GRAD: {
            const float t = s.z[in_row * s.ldx + c] + (s.q ? s.q[c] : 0.f);
            const float sg = 1.0f / (1.0f + __expf(-1.702f * t));
            v = x * (sg * (1.0f + 1.702f * t * (1.0f - sg)));
            break;
          }
          default: v = x;
        }
      }
      const bf16_t hi = bf16_rn(v);
      o = part == (s.worder ? 1 : 2) ? bf16_rn(v - bf16_to_f32_always(hi)) : hi;
    }
    s.out[idx] = o;
  }
}

// the same with 8 consecutive columns per thread (16-byte stores, two 16-byte loads): geometry with C % 8 == 0, Cs % 8 == 0
__global__ __launch_bounds__(256) void split3_v8_kernel(Split3Params s, long total8) {
  const int K8 = s.Kp / 8, Cs8 = s.Cs / 8;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (long)gridDim.x * 256) {
    const long row = idx / K8;
    const int j8 = (int)(idx - row * K8);
    const int part = j8 / Cs8;
    const int c = (j8 - part * Cs8) * 8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (part < 3 && c < s.C) {
      long in_row = row;
      bool live = true;
      int b = 0;
      if (s.geo != 0 || s.pq_img) {
        const int Ho = s.geo == 1 ? s.H / 2 : (s.geo == 2 ? s.H * 2 : s.H);
        const int Wo = s.geo == 1 ? s.W / 2 : (s.geo == 2 ? s.W * 2 : s.W);
        b = (int)(row / ((long)Ho * Wo));
        const int r = (int)(row - (long)b * Ho * Wo);
        const int oy = r / Wo, ox = r - oy * Wo;
        if (s.geo == 1) {
          in_row = ((long)b * s.H + oy * 2) * s.W + ox * 2;
        } else if (s.geo == 2) {
          live = ((oy | ox) & 1) == 0;
          in_row = ((long)b * s.H + (oy >> 1)) * s.W + (ox >> 1);
        }
      }
      if (live) {
        float v[8], zz[8], pp[8], qq[8];
        const f32x4* xp = reinterpret_cast<const f32x4*>(s.x + in_row * s.ldx + c);
        const f32x4 x0 = xp[0], x1 = xp[1];
        const int pi = s.pq_img ? b * s.C + c : c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pp[e] = s.p ? s.p[(s.op == P_AFFINE ? pi : c) + e] : 0.f;
          qq[e] = s.q ? s.q[(s.op == P_AFFINE ? pi : c) + e] : 0.f;
          zz[e] = 0.f;
        }
        if (s.op == P_PRELU_GRAD || s.op == P_RELU_GRAD || s.op == P_QGELU_GRAD) {
          const f32x4* zp = reinterpret_cast<const f32x4*>(s.z + in_row * s.ldx + c);
          const f32x4 z0 = zp[0], z1 = zp[1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { zz[e] = z0[e]; zz[4 + e] = z1[e]; }
        }
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y;
          switch (s.op) {
            case P_AFFINE: y = pp[e] * v[e] + qq[e]; break;
            case P_PRELU: { const float t = v[e] + qq[e]; y = t > 0.f ? t : pp[e] * t; break; }
            case P_PRELU_GRAD: { const float t = zz[e] + qq[e]; y = v[e] * (t > 0.f ? 1.f : pp[e]); break; }
            case P_RELU: { const float t = v[e] + qq[e]; y = t > 0.f ? t : 0.f; break; }
            case P_RELU_GRAD: { const float t = zz[e] + qq[e]; y = t > 0.f ? v[e] : 0.f; break; }
            case P_QGELU: { const float t = v[e] + qq[e]; y = t / (1.0f + __expf(-1.702f * t)); break; }
            case P_QGELU_GRAD: {
              const float t = zz[e] + qq[e];
              const float sg = 1.0f / (1.0f + __expf(-1.702f * t));
              y = v[e] * (sg * (1.0f + 1.702f * t * (1.0f - sg)));
              break;
            }
            default: y = v[e];
          }
          const bf16_t hi = bf16_rn(y);
          const bf16_t ov = part == (s.worder ? 1 : 2) ? bf16_rn(y - bf16_to_f32_always(hi)) : hi;
          if (e & 1) w[e >> 1] |= (uint32_t)ov << 16; else w[e >> 1] = ov;
        }
        o = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    *reinterpret_cast<uint4*>(s.out + idx * 8) = o;
  }
}

// weights: fp32 torch layout [O][I][k][k] (k = 1 or 3; linear = k 1) -> bf16 GEMM operand with parts (hi, lo, hi)
//   forward:  out[o][tap][part*Cs + i]  = split(w[o][i][tap] * scale[o])
//   dgrad:    out[i][tap][part*Cs + o]  = split(w[o][i][kk-1-tap] * scale[o])      (taps flipped, roles of I / O swapped)
// perm_hw > 0 (linear after a flatten of an NCHW tensor): input index i = ch * perm_hw + hw is re-addressed as
// hw * perm_c + ch, the order of the NHWC activations.  Rows beyond the real row count (N padded to 4) are zero.
__global__ __launch_bounds__(256) void pack_split3_w_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                            bf16_t* __restrict__ out, int O, int I, int kk, int dgrad, int Cs,
                                                            int Kp, int rows_out, int perm_hw, int perm_c, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int j = (int)(idx % Kp);
    const long rt = idx / Kp;
    const int tap = (int)(rt % kk);
    const int n = (int)(rt / kk);
    const int part = j / Cs, c = j - part * Cs;
    const int NC = dgrad ? O : I;       // channels along K
    const int NR = dgrad ? I : O;       // output rows
    bf16_t ov = 0;
    if (part < 3 && c < NC && n < NR) {
      const int o = dgrad ? c : n;
      int i = dgrad ? n : c;
      const int t = dgrad ? kk - 1 - tap : tap;
      if (perm_hw > 0) {                // i is in NHWC flatten order -> torch's NCHW flatten order
        const int hw = i / perm_c, ch = i - hw * perm_c;
        i = ch * perm_hw + hw;
      }
      float v = w[((long)o * I + i) * kk + t];
      if (scale) v *= scale[o];
      const bf16_t hi = bf16_rn(v);
      ov = part == 1 ? bf16_rn(v - bf16_to_f32_always(hi)) : hi;
    }
    (void)rows_out;
    out[idx] = ov;
  }
}

// BatchNorm (eval): p = gamma / sqrt(var + eps), q = beta - mean * p;  rep > 1 tiles the result rep times
// (the affine in front of a flatten: index hw * C + c)
__global__ void bn_affine_kernel(const float* g, const float* b, const float* m, const float* v, float eps, float* p, float* q,
                                 int C, int rep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * rep) return;
  const int c = i % C;
  const float s = g[c] / sqrtf(v[c] + eps);
  p[i] = s;
  q[i] = b[c] - m[c] * s;
}
// folded bias of Linear + BatchNorm1d: q = (bias - mean) * p + beta
__global__ void bn_fold_bias_kernel(const float* bias, const float* g, const float* b, const float* m, const float* v, float eps,
                                    float* p, float* q, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float s = g[c] / sqrtf(v[c] + eps);
  p[c] = s;
  q[c] = ((bias ? bias[c] : 0.f) - m[c]) * s + b[c];
}

// y = act(x + q): materialised activation (network stem), op = P_PRELU / P_RELU
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ q,
                                                  float* __restrict__ y, long total, int C, int op) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float t = x[i] + (q ? q[c] : 0.f);
    y[i] = t > 0.f ? t : (op == P_PRELU ? p[c] * t : 0.f);
  }
}

// ---- squeeze-excitation.  Pixel slabs: partial[b][slab][c] = sum over the slab's pixels of u (se_nslab(HW) slabs, a
// function of HW only); the fc kernels fold the slabs in order.  One block per (64-channel slice, image, slab).
__global__ __launch_bounds__(256) void se_pool_kernel(const float* __restrict__ u, float* __restrict__ part, int HW, int C, int nslab) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, sl = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const int per = (HW + nslab - 1) / nslab, p0 = sl * per, p1 = p0 + per < HW ? p0 + per : HW;
  float s = 0.f;
  if (c < C)
    for (int p = p0 + r; p < p1; p += 4) s += u[((long)b * HW + p) * C + c];
  red[r][threadIdx.x & 63] = s;
  __syncthreads();
  if (r == 0 && c < C) part[((long)b * nslab + sl) * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// h = relu(W1 pool), s = sigmoid(W2 h): one block per image (C <= 512, C/16 <= 32)
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ part, const float* __restrict__ bias, const float* __restrict__ w1,
                                                    const float* __restrict__ w2, float* __restrict__ hbuf, float* __restrict__ sbuf, int C,
                                                    int R, int nslab, float inv_hw) {
  __shared__ float pl[512], hh[32];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int sl = 0; sl < nslab; ++sl) a += part[((long)b * nslab + sl) * C + c];
    pl[c] = a * inv_hw + bias[c];          // mean_hw(u + bias)
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = wv; r < R; r += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w1[(long)r * C + c] * pl[c];
    a = wave_sum(a);
    if (lane == 0) { hh[r] = a > 0.f ? a : 0.f; hbuf[(long)b * R + r] = hh[r]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += w2[(long)c * R + r] * hh[r];
    sbuf[(long)b * C + c] = 1.0f / (1.0f + __expf(-a));
  }
}
// Y = (u + bias) * s[b][c] + shortcut;  shortcut = sc + sc_bias (conv) or X at (stride y, stride x) (identity)
__global__ __launch_bounds__(256) void se_combine_kernel(const float* __restrict__ u, const float* __restrict__ bias, const float* __restrict__ s,
                                                         const float* __restrict__ sc, const float* __restrict__ sc_bias,
                                                         const float* __restrict__ X, int stride, float* __restrict__ Y, int B, int Ho,
                                                         int Wo, int C) {
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int b = (int)(pix / ((long)Ho * Wo));
    float sh;
    if (sc) {
      sh = sc[i] + sc_bias[c];
    } else {
      const int r = (int)(pix - (long)b * Ho * Wo);
      const int oy = r / Wo, ox = r - oy * Wo;
      sh = X[(((long)b * Ho * stride + oy * stride) * (Wo * stride) + ox * stride) * C + c];
    }
    Y[i] = (u[i] + bias[c]) * s[(long)b * C + c] + sh;
  }
}
// backward: partial[b][slab][c] = sum over the slab of dY * (u + bias)
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const float* __restrict__ dY, const float* __restrict__ u, const float* __restrict__ bias,
                                                            float* __restrict__ part, int HW, int C, int nslab) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, sl = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const int per = (HW + nslab - 1) / nslab, p0 = sl * per, p1 = p0 + per < HW ? p0 + per : HW;
  float s = 0.f;
  if (c < C) {
    const float bc = bias[c];
    for (int p = p0 + r; p < p1; p += 4) {
      const long i = ((long)b * HW + p) * C + c;
      s += dY[i] * (u[i] + bc);
    }
  }
  red[r][threadIdx.x & 63] = s;
  __syncthreads();
  if (r == 0 && c < C) part[((long)b * nslab + sl) * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// r[b][c] = (W1^T ((W2^T (gs * s (1-s))) * (h > 0)))[c] / HW : the pooled path of the SE gradient, spread back over the pixels
__global__ __launch_bounds__(256) void se_fc_bwd_kernel(const float* __restrict__ part, const float* __restrict__ sbuf, const float* __restrict__ hbuf,
                                                        const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ rbuf,
                                                        int C, int R, int nslab, float inv_hw) {
  __shared__ float dt[512], dh[32];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float s = sbuf[(long)b * C + c];
    float gs = 0.f;
    for (int sl = 0; sl < nslab; ++sl) gs += part[((long)b * nslab + sl) * C + c];
    dt[c] = gs * s * (1.0f - s);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = wv; r < R; r += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w2[(long)c * R + r] * dt[c];
    a = wave_sum(a);
    if (lane == 0) dh[r] = hbuf[(long)b * R + r] > 0.f ? a : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += w1[(long)r * C + c] * dh[r];
    rbuf[(long)b * C + c] = a * inv_hw;
  }
}
// dX = p[c] * dxa + shortcut gradient;  identity shortcut: dY scattered to the (stride y, stride x) grid; conv shortcut: dsc likewise
__global__ __launch_bounds__(256) void unit_bwd_combine_kernel(const float* __restrict__ dxa, const float* __restrict__ p,
                                                               const float* __restrict__ dsc, int stride, float* __restrict__ dX, int B, int H,
                                                               int W, int C) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int b = (int)(pix / ((long)H * W));
    const int r = (int)(pix - (long)b * H * W);
    const int y = r / W, x = r - y * W;
    float v = p[c] * dxa[i];
    if (y % stride == 0 && x % stride == 0)
      v += dsc[(((long)b * (H / stride) + y / stride) * (W / stride) + x / stride) * C + c];
    dX[i] = v;
  }
}
// y[i] = x[i] * p[i % n]   (gradient of the affine in front of the flatten)
__global__ __launch_bounds__(256) void scale_cols_kernel(const float* __restrict__ x, const float* __restrict__ p, float* __restrict__ y,
                                                         long total, int n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) y[i] = x[i] * p[i % n];
}

// ---- IDLoss.extract_feats front end (arcface_model.py:40-46): crop [35:223, 32:220] of the 256 x 256 image, adaptive
// average pool to 112 x 112 (window of output o: [floor(o*188/112), ceil((o+1)*188/112)) ); NCHW fp32 -> NHWC fp32
constexpr int FP_Y0 = 35, FP_X0 = 32, FP_IN = 188, FP_OUT = 112, FP_S = 256;
__device__ __forceinline__ int fp_lo(int o) { return (o * FP_IN) / FP_OUT; }
__device__ __forceinline__ int fp_hi(int o) { return ((o + 1) * FP_IN + FP_OUT - 1) / FP_OUT; }
__global__ __launch_bounds__(256) void face_pool_kernel(const float* __restrict__ img, float* __restrict__ out, int B) {
  const long total = (long)B * FP_OUT * FP_OUT * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const long pix = i / 3;
    const int b = (int)(pix / (FP_OUT * FP_OUT));
    const int r = (int)(pix - (long)b * FP_OUT * FP_OUT);
    const int oy = r / FP_OUT, ox = r - oy * FP_OUT;
    const int y0 = fp_lo(oy), y1 = fp_hi(oy), x0 = fp_lo(ox), x1 = fp_hi(ox);
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) s += img[(((long)b * 3 + c) * FP_S + FP_Y0 + y) * FP_S + FP_X0 + x];
    out[i] = s / (float)((y1 - y0) * (x1 - x0));
  }
}
// d image[b][c][Y][X] = sum over the output windows covering the pixel of d out / window size; 0 outside the crop.  g: [B*112*112][ldg]
__global__ __launch_bounds__(256) void face_pool_bwd_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dimg, int B) {
  const long total = (long)B * 3 * FP_S * FP_S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int X = (int)(i % FP_S), Y = (int)((i / FP_S) % FP_S);
    const int c = (int)((i / ((long)FP_S * FP_S)) % 3), b = (int)(i / ((long)3 * FP_S * FP_S));
    const int y = Y - FP_Y0, x = X - FP_X0;
    float s = 0.f;
    if ((unsigned)y < (unsigned)FP_IN && (unsigned)x < (unsigned)FP_IN) {
      const int oyc = (y * FP_OUT) / FP_IN, oxc = (x * FP_OUT) / FP_IN;
      for (int oy = oyc - 1; oy <= oyc + 1; ++oy) {
        if ((unsigned)oy >= (unsigned)FP_OUT || y < fp_lo(oy) || y >= fp_hi(oy)) continue;
        const int hy = fp_hi(oy) - fp_lo(oy);
        for (int ox = oxc - 1; ox <= oxc + 1; ++ox) {
          if ((unsigned)ox >= (unsigned)FP_OUT || x < fp_lo(ox) || x >= fp_hi(ox)) continue;
          const int hx = fp_hi(ox) - fp_lo(ox);
          s += g[(((long)b * FP_OUT + oy) * FP_OUT + ox) * ldg + c] / (float)(hy * hx);
        }
      }
    }
    dimg[i] = s;
  }
}

// ---- identity loss head (arcface_model.py:48-67, model_irse.py:44-48): f = raw + bias; e = f / |f| (Backbone), e2 = e / |e|
// (F.normalize), cos = <e2, ref>, loss_b = 1 - cos.  d(scale * loss_b) / d f via the two normalisations.  One block per image.
__global__ __launch_bounds__(256) void cos_head_kernel(const float* __restrict__ raw, const float* __restrict__ bias, const float* __restrict__ ref,
                                                       int ref_stride, float* __restrict__ feat, float* __restrict__ loss,
                                                       float* __restrict__ df, int D, float scale) {
  __shared__ float red[4];
  __shared__ float f[512];
  const int b = blockIdx.x;
  auto bsum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float ss = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    f[i] = raw[(long)b * D + i] + bias[i];
    ss += f[i] * f[i];
  }
  const float n1 = sqrtf(bsum(ss));
  float s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) { f[i] = f[i] / n1; s2 += f[i] * f[i]; }       // e
  const float n2 = fmaxf(sqrtf(bsum(s2)), 1e-12f);
  float dot = 0.f;
  const float* r = ref ? ref + (long)b * ref_stride : nullptr;
  for (int i = threadIdx.x; i < D; i += 256) {
    f[i] = f[i] / n2;                                                                        // e2
    if (feat) feat[(long)b * D + i] = f[i];
    if (r) dot += f[i] * r[i];
  }
  if (!r) return;
  const float cs = bsum(dot);
  if (threadIdx.x == 0 && loss) loss[b] = 1.0f - cs;
  if (df) {
    // d(-cos)/d e2 = -ref;  through e2 = e / n2: (g - e2 <g, e2>) / n2;  through e = f / n1 likewise (e2 is the direction of both)
    const float k = -scale / (n1 * n2);
    for (int i = threadIdx.x; i < D; i += 256) df[(long)b * D + i] = k * (r[i] - cs * f[i]);
  }
}

// ---- LPIPS-VGG pieces (lpips 0.1.4: ScalingLayer, vgg16 slices, normalize_tensor, lin layers, spatial_average)
// (x - shift_c) / scale_c, NCHW fp32 -> NHWC fp32 [B*H*W][3]
__global__ __launch_bounds__(256) void lpips_prep_kernel(const float* __restrict__ img, float* __restrict__ out, float3 shift, float3 scale,
                                                         int B, int H, int W) {
  const long total = (long)B * H * W * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const long pix = i / 3;
    const int b = (int)(pix / ((long)H * W));
    const long r = pix - (long)b * H * W;
    const float sh = c == 0 ? shift.x : (c == 1 ? shift.y : shift.z), sc = c == 0 ? scale.x : (c == 1 ? scale.y : scale.z);
    out[i] = (img[((long)b * 3 + c) * H * W + r] - sh) / sc;
  }
}
// d img[b][c][pix] = g[pix][c] / scale_c ; g: [B*H*W][ldg]
__global__ __launch_bounds__(256) void lpips_prep_bwd_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dimg, float3 scale, int B,
                                                             int H, int W) {
  const long total = (long)B * 3 * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i % ((long)H * W);
    const int c = (int)((i / ((long)H * W)) % 3), b = (int)(i / ((long)3 * H * W));
    const float sc = c == 0 ? scale.x : (c == 1 ? scale.y : scale.z);
    dimg[i] = g[((long)b * H * W + r) * ldg + c] / sc;
  }
}
// 2 x 2 / stride 2 max pooling of a pre-activation tensor (ReLU and the per-channel bias commute with it); idx = winner 0..3
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ z, float* __restrict__ zp, uint8_t* __restrict__ idx, int B,
                                                       int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int b = (int)(pix / ((long)Ho * Wo));
    const int r = (int)(pix - (long)b * Ho * Wo);
    const int oy = r / Wo, ox = r - oy * Wo;
    const float* src = z + (((long)b * H + oy * 2) * W + ox * 2) * C + c;
    float best = src[0];
    int bi = 0;
    const float v1 = src[C], v2 = src[(long)W * C], v3 = src[(long)W * C + C];
    if (v1 > best) { best = v1; bi = 1; }
    if (v2 > best) { best = v2; bi = 2; }
    if (v3 > best) { best = v3; bi = 3; }
    zp[i] = best;
    idx[i] = (uint8_t)bi;
  }
}
// G[full] = add[full] (or 0) + (the pixel won its window ? g_pooled : 0)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ gp, const uint8_t* __restrict__ idx,
                                                           const float* __restrict__ add, float* __restrict__ G, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int b = (int)(pix / ((long)H * W));
    const int r = (int)(pix - (long)b * H * W);
    const int y = r / W, x = r - y * W;
    const long pi = (((long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c;
    float v = add ? add[i] : 0.f;
    if (idx[pi] == (uint8_t)((y & 1) * 2 + (x & 1))) v += gp[pi];
    G[i] = v;
  }
}
// One wave per pixel.  F = relu(z + bias); n = |F| + 1e-10; Fn = F / n.
//   src == nullptr: write Fn to fn_out (the source image's normalised features)
//   else: d = sum_c w_c (Fn - src)^2 -> dpix[pixel]; dF (gradient of  gscale * mean_hw d  w.r.t. F) -> dF
__global__ __launch_bounds__(256) void lpips_head_kernel(const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ src,
                                                         long src_img_stride, const float* __restrict__ w, float* __restrict__ fn_out,
                                                         float* __restrict__ dpix, float* __restrict__ dF, long npix, int HW, int C,
                                                         float gscale) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const float* zr = z + pix * C;
  float f[8];                       // C <= 512
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + k * 64;
    f[k] = 0.f;
    if (c < C) {
      const float t = zr[c] + bias[c];
      f[k] = t > 0.f ? t : 0.f;
      ss += f[k] * f[k];
    }
  }
  const float nrm = sqrtf(wave_sum(ss));
  const float n = nrm + 1e-10f;
  if (!src) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lane + k * 64;
      if (c < C) fn_out[pix * C + c] = f[k] / n;
    }
    return;
  }
  const long b = pix / HW;
  const float* sr = src + b * src_img_stride + (pix - b * HW) * C;
  float g[8], d = 0.f, gf = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + k * 64;
    g[k] = 0.f;
    if (c < C) {
      const float df = f[k] / n - sr[c];
      d += w[c] * df * df;
      g[k] = 2.0f * w[c] * df * gscale;      // d / d Fn
      gf += g[k] * f[k];
    }
  }
  d = wave_sum(d);
  gf = wave_sum(gf);
  if (lane == 0) dpix[pix] = d;
  if (dF) {
    // Fn = F / (|F| + eps):  dF = g / n - F (g . F) / (|F| n^2)
    const float k2 = nrm > 0.f ? gf / (nrm * n * n) : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lane + k * 64;
      if (c < C) dF[pix * C + c] = g[k] / n - f[k] * k2;
    }
  }
}
// acc[b] (+)= mean over the image's pixels of dpix, summed in a fixed order: one block per image
__global__ __launch_bounds__(256) void lpips_reduce_kernel(const float* __restrict__ dpix, float* __restrict__ acc, int HW, int first) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) s += dpix[(long)b * HW + p];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) / (float)HW;
    acc[b] = first ? v : acc[b] + v;
  }
}

}  // namespace

static inline dim3 pgrid(long total) { return dim3(ew_grid(total)); }

int lpips_prep_launch(const float* img, float* out, const float* shift3, const float* scale3, int B, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(lpips_prep_kernel, pgrid((long)B * H * W * 3), dim3(256), 0, st, img, out, make_float3(shift3[0], shift3[1], shift3[2]),
                     make_float3(scale3[0], scale3[1], scale3[2]), B, H, W);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int lpips_prep_bwd_launch(const float* g, int ldg, float* dimg, const float* scale3, int B, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(lpips_prep_bwd_kernel, pgrid((long)B * 3 * H * W), dim3(256), 0, st, g, ldg, dimg,
                     make_float3(scale3[0], scale3[1], scale3[2]), B, H, W);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int maxpool2_launch(const float* z, float* zp, uint8_t* idx, int B, int H, int W, int C, hipStream_t st) {
  ARG_CHECK(H % 2 == 0 && W % 2 == 0, "maxpool2: even size");
  hipLaunchKernelGGL(maxpool2_kernel, pgrid((long)B * (H / 2) * (W / 2) * C), dim3(256), 0, st, z, zp, idx, B, H, W, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int maxpool2_bwd_launch(const float* gp, const uint8_t* idx, const float* add, float* G, int B, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(maxpool2_bwd_kernel, pgrid((long)B * H * W * C), dim3(256), 0, st, gp, idx, add, G, B, H, W, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int lpips_head_launch(const float* z, const float* bias, const float* src, long src_img_stride, const float* w, float* fn_out, float* dpix,
                      float* dF, int B, int HW, int C, float gscale, hipStream_t st) {
  ARG_CHECK(C <= 512, "lpips_head: C <= 512");
  const long npix = (long)B * HW;
  hipLaunchKernelGGL(lpips_head_kernel, dim3(cdiv(npix, 4)), dim3(256), 0, st, z, bias, src, src_img_stride, w, fn_out, dpix, dF, npix, HW, C,
                     gscale);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int lpips_reduce_launch(const float* dpix, float* acc, int B, int HW, int first, hipStream_t st) {
  hipLaunchKernelGGL(lpips_reduce_kernel, dim3(B), dim3(256), 0, st, dpix, acc, HW, first);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int split3_launch(const Split3Params& s, long rows_out, hipStream_t st) {
  ARG_CHECK(s.Kp % 64 == 0 && 3 * s.Cs <= s.Kp + 0 && s.C <= s.Cs, "split3: operand geometry");
  const long total = rows_out * s.Kp;
  if (s.C % 8 == 0 && s.Cs % 8 == 0 && s.ldx % 4 == 0) {
    hipLaunchKernelGGL(split3_v8_kernel, pgrid(total / 8), dim3(256), 0, st, s, total / 8);
    LAUNCH_CHECK();
    return HEDIT_OK;
  }
  hipLaunchKernelGGL(split3_kernel, pgrid(total), dim3(256), 0, st, s, total);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int pack_split3_w_launch(const float* w, const float* scale, bf16_t* out, int O, int I, int k, int dgrad, int Cs, int Kp,
                         int rows_out, int perm_hw, int perm_c, hipStream_t st) {
  const long total = (long)rows_out * k * k * Kp;
  hipLaunchKernelGGL(pack_split3_w_kernel, pgrid(total), dim3(256), 0, st, w, scale, out, O, I, k * k, dgrad, Cs, Kp, rows_out, perm_hw,
                     perm_c, total);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int bn_affine_launch(const float* g, const float* b, const float* m, const float* v, float eps, float* p, float* q, int C, int rep,
                     hipStream_t st) {
  hipLaunchKernelGGL(bn_affine_kernel, dim3(cdiv((long)C * rep, 256)), dim3(256), 0, st, g, b, m, v, eps, p, q, C, rep);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int bn_fold_bias_launch(const float* bias, const float* g, const float* b, const float* m, const float* v, float eps, float* p, float* q,
                        int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_bias_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, bias, g, b, m, v, eps, p, q, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int act_launch(const float* x, const float* p, const float* q, float* y, long total, int C, int op, hipStream_t st) {
  hipLaunchKernelGGL(act_kernel, pgrid(total), dim3(256), 0, st, x, p, q, y, total, C, op);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int se_nslab(int HW) {
  int n = HW / 64;
  return n < 1 ? 1 : (n > 32 ? 32 : n);
}
int se_pool_launch(const float* u, float* part, int B, int HW, int C, hipStream_t st) {
  const int ns = se_nslab(HW);
  hipLaunchKernelGGL(se_pool_kernel, dim3(cdiv(C, 64), B, ns), dim3(256), 0, st, u, part, HW, C, ns);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int se_fc_launch(const float* part, const float* bias, const float* w1, const float* w2, float* hbuf, float* sbuf, int B, int HW, int C,
                 int R, hipStream_t st) {
  ARG_CHECK(C <= 512 && R <= 32, "se_fc: C <= 512, C / 16 <= 32");
  hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(256), 0, st, part, bias, w1, w2, hbuf, sbuf, C, R, se_nslab(HW), 1.0f / (float)HW);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int se_combine_launch(const float* u, const float* bias, const float* s, const float* sc, const float* sc_bias, const float* X, int stride,
                      float* Y, int B, int Ho, int Wo, int C, hipStream_t st) {
  hipLaunchKernelGGL(se_combine_kernel, pgrid((long)B * Ho * Wo * C), dim3(256), 0, st, u, bias, s, sc, sc_bias, X, stride, Y, B, Ho, Wo, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int se_bwd_reduce_launch(const float* dY, const float* u, const float* bias, float* part, int B, int HW, int C, hipStream_t st) {
  const int ns = se_nslab(HW);
  hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3(cdiv(C, 64), B, ns), dim3(256), 0, st, dY, u, bias, part, HW, C, ns);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int se_fc_bwd_launch(const float* part, const float* sbuf, const float* hbuf, const float* w1, const float* w2, float* rbuf, int B, int C,
                     int R, int HW, hipStream_t st) {
  ARG_CHECK(C <= 512 && R <= 32, "se_fc_bwd: C <= 512, C / 16 <= 32");
  hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(B), dim3(256), 0, st, part, sbuf, hbuf, w1, w2, rbuf, C, R, se_nslab(HW), 1.0f / (float)HW);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int unit_bwd_combine_launch(const float* dxa, const float* p, const float* dsc, int stride, float* dX, int B, int H, int W, int C,
                            hipStream_t st) {
  hipLaunchKernelGGL(unit_bwd_combine_kernel, pgrid((long)B * H * W * C), dim3(256), 0, st, dxa, p, dsc, stride, dX, B, H, W, C);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int scale_cols_launch(const float* x, const float* p, float* y, long total, int n, hipStream_t st) {
  hipLaunchKernelGGL(scale_cols_kernel, pgrid(total), dim3(256), 0, st, x, p, y, total, n);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int face_pool_launch(const float* img, float* out, int B, hipStream_t st) {
  hipLaunchKernelGGL(face_pool_kernel, pgrid((long)B * 112 * 112 * 3), dim3(256), 0, st, img, out, B);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int face_pool_bwd_launch(const float* g, int ldg, float* dimg, int B, hipStream_t st) {
  hipLaunchKernelGGL(face_pool_bwd_kernel, pgrid((long)B * 3 * 256 * 256), dim3(256), 0, st, g, ldg, dimg, B);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
int cos_head_launch(const float* raw, const float* bias, const float* ref, int ref_stride, float* feat, float* loss, float* df, int B, int D,
                    float scale, hipStream_t st) {
  ARG_CHECK(D <= 512, "cos_head: feature dimension <= 512");
  hipLaunchKernelGGL(cos_head_kernel, dim3(B), dim3(256), 0, st, raw, bias, ref, ref_stride, feat, loss, df, D, scale);
  LAUNCH_CHECK();
  return HEDIT_OK;
}
