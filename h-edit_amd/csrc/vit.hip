// The style encoder of the combined text + style task as a native executor: `CLIPEncoder.get_gram_matrix_residual`
// of the reference's text-guided-n-style/clip_guidance/base_clip.py:55-66 on the slice of clip_guidance/clip/model.py it
// exercises (VisionTransformer :195-237 through `encode_image_with_features` :339-365): patch embedding, class token +
// positional embedding, ln_pre and the first `layers` ResidualAttentionBlocks (LayerNorm, multi-head attention,
// QuickGELU MLP, :153-190); the Gram matrix of the last block's patch tokens, its residual against the style
// reference's Gram matrix, the Frobenius norm -- TOGETHER WITH the gradient w.r.t. the (CLIP-normalised, resized)
// image, which is what the style closure of inversion/h_edit.py:162-182 pulls back into the VAE decoder.
// SURVEY.md section 8 row a19 / boundary entry `hedit_vit_gram_fwd_bwd`.
//
// Arithmetic: fp32 token stream, LayerNorm / softmax / QuickGELU in fp32, every contraction a three-term split-bf16
// product with fp32 accumulation (pnet.hip) -- finer than the reference's fp16 CLIP (model.py:414-435).  Attention
// over the 197 tokens of an image runs in LDS-resident fp32 kernels (one workgroup per image and head).
// Parameters by the OpenAI CLIP state_dict names (`visual.conv1.weight`, `visual.transformer.resblocks.0.attn.in_proj_weight` ...).
#include "pnet.h"

namespace {

constexpr float LN_EPS = 1e-5f;

// ---- patches.  img [B][3][R][R] -> X [B*P*P][3*p*p], column = (c, ky, kx): conv1 with kernel = stride = p is a linear map
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ X, int B, int R, int p, int bwd) {
  const int P = R / p, K = 3 * p * p;
  const long total = (long)B * P * P * K;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % K);
    const long row = i / K;
    const int b = (int)(row / (P * P)), pr = (int)(row % (P * P));
    const int c = k / (p * p), ky = (k / p) % p, kx = k % p;
    const long src = (((long)b * 3 + c) * R + (pr / P) * p + ky) * R + (pr % P) * p + kx;
    if (bwd) const_cast<float*>(img)[src] = X[i];      // the inverse permutation (every pixel belongs to exactly one patch)
    else X[i] = img[src];
  }
}
// T0[b][0] = cls + pos[0]; T0[b][1+l] = E[b][l] + pos[1+l]
__global__ __launch_bounds__(256) void tokens_kernel(const float* __restrict__ E, const float* __restrict__ cls, const float* __restrict__ pos,
                                                     float* __restrict__ T, int B, int L, int W) {
  const long total = (long)B * L * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int w = (int)(i % W);
    const long row = i / W;
    const int b = (int)(row / L), l = (int)(row % L);
    T[i] = (l == 0 ? cls[w] : E[((long)b * (L - 1) + l - 1) * W + w]) + pos[(long)l * W + w];
  }
}
// dE[b][l] = dT[b][1+l]
__global__ __launch_bounds__(256) void drop_cls_kernel(const float* __restrict__ dT, float* __restrict__ dE, int B, int L, int W) {
  const long total = (long)B * (L - 1) * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int w = (int)(i % W);
    const long row = i / W;
    const int b = (int)(row / (L - 1)), l = (int)(row % (L - 1));
    dE[i] = dT[((long)b * L + l + 1) * W + w];
  }
}

// ---- LayerNorm over W (one wave per row), statistics kept for the backward pass
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                     float* __restrict__ y, float* __restrict__ stats, long rows, int W) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * W;
  float s = 0.f;
  for (int i = lane; i < W; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)W;
  float q = 0.f;
  for (int i = lane; i < W; i += 64) { const float d = xr[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)W + LN_EPS);
  for (int i = lane; i < W; i += 64) y[row * W + i] = (xr[i] - mean) * rstd * g[i] + b[i];
  if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}
// dx = add + rstd * (dy g - mean(dy g) - xhat mean(dy g xhat))
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ g,
                                                     const float* __restrict__ stats, const float* __restrict__ add, float* __restrict__ dx,
                                                     long rows, int W) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  const float* xr = x + row * W;
  const float* dr = dy + row * W;
  float a = 0.f, c = 0.f;
  for (int i = lane; i < W; i += 64) {
    const float t = dr[i] * g[i];
    a += t;
    c += t * (xr[i] - mean) * rstd;
  }
  a = wave_sum(a) / (float)W;
  c = wave_sum(c) / (float)W;
  for (int i = lane; i < W; i += 64) {
    const float xh = (xr[i] - mean) * rstd;
    const float v = rstd * (dr[i] * g[i] - a - xh * c);
    dx[row * W + i] = (add ? add[row * W + i] : 0.f) + v;
  }
}
// out = res + raw + bias
__global__ __launch_bounds__(256) void add_bias_res_kernel(const float* __restrict__ raw, const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ out, long total, int W) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[i] = res[i] + raw[i] + bias[i % W];
}

// ---- attention over the L tokens of one image, head dimension 64, fp32, K / V (or Q / dA) of one (image, head) resident in
// LDS.  qkv: [B*L][3W] raw (bias added on load), head h uses columns [h*64, h*64+64) of the q / k / v thirds.  Rows are
// padded to 68 floats: 16-byte aligned, and the four 16-lane groups of a ds_read_b128 with lane = row hit all 64 banks.
// Workgroups: (head, image, slice) -- the query rows (forward, dq) or the keys (dk / dv) are cut into ASPLIT slices so
// that 12 heads x 8 images fill the chip; every output element is still produced by exactly one wave in a fixed order.
constexpr int HD = 64, HP = 68, LMAX = 200;
typedef __attribute__((ext_vector_type(4))) float fl4;
__device__ __forceinline__ void load_head(const float* __restrict__ src, int ld, const float* __restrict__ bias, int col0, int L, float* dst,
                                          float mul) {
  for (int i = threadIdx.x; i < L * (HD / 4); i += 256) {
    const int r = i >> 4, c = (i & 15) * 4;
    fl4 v = *reinterpret_cast<const fl4*>(src + (long)r * ld + col0 + c);
    if (bias) v += *reinterpret_cast<const fl4*>(bias + col0 + c);
    *reinterpret_cast<fl4*>(dst + r * HP + c) = v * mul;
  }
}
// Two register-resident 64-float rows against one LDS row: the LDS row is read once for both (two independent FMA chains)
__device__ __forceinline__ void dot64x2(const fl4 (&a0)[16], const fl4 (&a1)[16], const float* __restrict__ row, float& s0, float& s1) {
  s0 = 0.f;
  s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const fl4 b = *reinterpret_cast<const fl4*>(row + 4 * c);
    s0 += a0[c][0] * b[0] + a0[c][1] * b[1] + a0[c][2] * b[2] + a0[c][3] * b[3];
    s1 += a1[c][0] * b[0] + a1[c][1] * b[1] + a1[c][2] * b[2] + a1[c][3] * b[3];
  }
}
// o0[lane] = sum_j w0[j] M[j][lane], o1 likewise (w in LDS, four at a time as a broadcast; M read once for both)
__device__ __forceinline__ void wsum2(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ M, int L, int lane,
                                      float& o0, float& o1) {
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;      // two partial chains per output: even / odd quads, folded at the end in a fixed order
  int j = 0;
  for (; j + 8 <= L; j += 8) {
    const fl4 p0 = *reinterpret_cast<const fl4*>(w0 + j), p1 = *reinterpret_cast<const fl4*>(w1 + j);
    const fl4 r0 = *reinterpret_cast<const fl4*>(w0 + j + 4), r1 = *reinterpret_cast<const fl4*>(w1 + j + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = M[(j + e) * HP + lane], m2 = M[(j + 4 + e) * HP + lane];
      a0 += p0[e] * m; a1 += p1[e] * m;
      b0 += r0[e] * m2; b1 += r1[e] * m2;
    }
  }
  for (; j < L; ++j) { const float m = M[j * HP + lane]; a0 += w0[j] * m; a1 += w1[j] * m; }
  o0 = a0 + b0;
  o1 = a1 + b1;
}
__device__ __forceinline__ void slice_of(int L, int z, int nz, int& lo, int& hi) {
  const int per = (L + nz - 1) / nz;
  lo = z * per;
  hi = lo + per < L ? lo + per : L;
}
// a head's 64-float row of an [rows][ld] tensor (+bias) times mul, every lane gets the whole row (broadcast loads)
__device__ __forceinline__ void load_row(const float* __restrict__ src, const float* __restrict__ bias, float mul, fl4 (&r)[16]) {
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    fl4 v = *reinterpret_cast<const fl4*>(src + 4 * c);
    if (bias) v += *reinterpret_cast<const fl4*>(bias + 4 * c);
    r[c] = v * mul;
  }
}

// Every wave works on TWO rows at a time (i0 and i0 + 4; the second is a clamped duplicate when the slice runs out, its
// results are then not stored): the LDS rows are read once for both and the FMA chains are independent.
// A[b][i][h*64 + d] = sum_j softmax_j(scale q_i . k_j) v_j[d];  lse[b][h][i] = log-sum-exp of the scaled scores
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ bias, float* __restrict__ A,
                                                       float* __restrict__ lse, int L, int W, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + LMAX * HP;
  float* ps = Vs + LMAX * HP;          // [8][LMAX]
  const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
  const float* base = qkv + (long)b * L * 3 * W;
  load_head(base, 3 * W, bias, W + h * HD, L, Ks, 1.f);
  load_head(base, 3 * W, bias, 2 * W + h * HD, L, Vs, 1.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* pa = ps + (2 * wv) * LMAX;
  float* pb = pa + LMAX;
  int lo, hi;
  slice_of(L, blockIdx.z, gridDim.z, lo, hi);
  for (int i0 = lo + wv; i0 < hi; i0 += 8) {
    const bool two = i0 + 4 < hi;
    const int i1 = two ? i0 + 4 : i0;
    fl4 qa[16], qb[16];
    load_row(base + (long)i0 * 3 * W + h * HD, bias + h * HD, scale, qa);
    load_row(base + (long)i1 * 3 * W + h * HD, bias + h * HD, scale, qb);
    float ma = -3.0e38f, mb = -3.0e38f;
    for (int j = lane; j < L; j += 64) {
      float sa, sb;
      dot64x2(qa, qb, Ks + j * HP, sa, sb);
      pa[j] = sa; pb[j] = sb;
      ma = fmaxf(ma, sa); mb = fmaxf(mb, sb);
    }
    ma = wave_max(ma); mb = wave_max(mb);
    float suma = 0.f, sumb = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float ea = __expf(pa[j] - ma), eb = __expf(pb[j] - mb);
      pa[j] = ea; pb[j] = eb;
      suma += ea; sumb += eb;
    }
    suma = wave_sum(suma); sumb = wave_sum(sumb);
    float oa, ob;
    wsum2(pa, pb, Vs, L, lane, oa, ob);
    A[((long)b * L + i0) * W + h * HD + lane] = oa / suma;
    if (lane == 0) lse[((long)b * heads + h) * L + i0] = ma + __logf(suma);
    if (two) {
      A[((long)b * L + i1) * W + h * HD + lane] = ob / sumb;
      if (lane == 0) lse[((long)b * heads + h) * L + i1] = mb + __logf(sumb);
    }
  }
}
// dq_i = scale * sum_j ds_ij k_j,  ds_ij = p_ij (dA_i . v_j - D_i),  D_i = dA_i . A_i
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ A,
                                                          const float* __restrict__ dA, const float* __restrict__ lse, float* __restrict__ dqkv,
                                                          int L, int W, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + LMAX * HP;
  float* ps = Vs + LMAX * HP;
  const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
  const float* base = qkv + (long)b * L * 3 * W;
  load_head(base, 3 * W, bias, W + h * HD, L, Ks, 1.f);
  load_head(base, 3 * W, bias, 2 * W + h * HD, L, Vs, 1.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* pa = ps + (2 * wv) * LMAX;
  float* pb = pa + LMAX;
  int lo, hi;
  slice_of(L, blockIdx.z, gridDim.z, lo, hi);
  for (int i0 = lo + wv; i0 < hi; i0 += 8) {
    const bool two = i0 + 4 < hi;
    const int i1 = two ? i0 + 4 : i0;
    const long ra = ((long)b * L + i0) * W + h * HD, rb = ((long)b * L + i1) * W + h * HD;
    fl4 qa[16], qb[16], ga[16], gb[16];
    load_row(base + (long)i0 * 3 * W + h * HD, bias + h * HD, scale, qa);
    load_row(base + (long)i1 * 3 * W + h * HD, bias + h * HD, scale, qb);
    load_row(dA + ra, nullptr, 1.f, ga);
    load_row(dA + rb, nullptr, 1.f, gb);
    const float Da = wave_sum(dA[ra + lane] * A[ra + lane]), Db = wave_sum(dA[rb + lane] * A[rb + lane]);
    const float la = lse[((long)b * heads + h) * L + i0], lb = lse[((long)b * heads + h) * L + i1];
    for (int j = lane; j < L; j += 64) {
      float sa, sb, da, db;
      dot64x2(qa, qb, Ks + j * HP, sa, sb);
      dot64x2(ga, gb, Vs + j * HP, da, db);
      pa[j] = __expf(sa - la) * (da - Da);
      pb[j] = __expf(sb - lb) * (db - Db);
    }
    float oa, ob;
    wsum2(pa, pb, Ks, L, lane, oa, ob);
    dqkv[((long)b * L + i0) * 3 * W + h * HD + lane] = oa * scale;
    if (two) dqkv[((long)b * L + i1) * 3 * W + h * HD + lane] = ob * scale;
  }
}
// dv_j = sum_i p_ij dA_i ;  dk_j = sum_i ds_ij (scale q_i)
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ A,
                                                           const float* __restrict__ dA, const float* __restrict__ lse, float* __restrict__ dqkv,
                                                           int L, int W, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;                      // [L][68] q * scale
  float* Gs = sm + LMAX * HP;          // [L][68] dA
  float* ps = Gs + LMAX * HP;          // [8][LMAX] p_ij   (two keys per wave)
  float* ds = ps + 8 * LMAX;           // [8][LMAX] ds_ij
  float* Dl = ds + 8 * LMAX;           // [LMAX][2]: D_i, lse_i
  const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
  const float* base = qkv + (long)b * L * 3 * W;
  load_head(base, 3 * W, bias, h * HD, L, Qs, scale);
  load_head(dA + (long)b * L * W, W, nullptr, h * HD, L, Gs, 1.f);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = wv; i < L; i += 4) {
    const long ro = ((long)b * L + i) * W + h * HD + lane;
    const float Di = wave_sum(dA[ro] * A[ro]);
    if (lane == 0) { Dl[i * 2] = Di; Dl[i * 2 + 1] = lse[((long)b * heads + h) * L + i]; }
  }
  __syncthreads();
  float* pa = ps + (2 * wv) * LMAX;
  float* pb = pa + LMAX;
  float* da = ds + (2 * wv) * LMAX;
  float* db = da + LMAX;
  int lo, hi;
  slice_of(L, blockIdx.z, gridDim.z, lo, hi);
  for (int j0 = lo + wv; j0 < hi; j0 += 8) {
    const bool two = j0 + 4 < hi;
    const int j1 = two ? j0 + 4 : j0;
    fl4 ka[16], kb[16], va[16], vb[16];
    load_row(base + (long)j0 * 3 * W + W + h * HD, bias + W + h * HD, 1.f, ka);
    load_row(base + (long)j1 * 3 * W + W + h * HD, bias + W + h * HD, 1.f, kb);
    load_row(base + (long)j0 * 3 * W + 2 * W + h * HD, bias + 2 * W + h * HD, 1.f, va);
    load_row(base + (long)j1 * 3 * W + 2 * W + h * HD, bias + 2 * W + h * HD, 1.f, vb);
    for (int i = lane; i < L; i += 64) {
      float sa, sb, ga, gb;
      dot64x2(ka, kb, Qs + i * HP, sa, sb);
      dot64x2(va, vb, Gs + i * HP, ga, gb);
      const float ea = __expf(sa - Dl[i * 2 + 1]), eb = __expf(sb - Dl[i * 2 + 1]);
      pa[i] = ea; pb[i] = eb;
      da[i] = ea * (ga - Dl[i * 2]); db[i] = eb * (gb - Dl[i * 2]);
    }
    float ka_o, kb_o, va_o, vb_o;
    wsum2(da, db, Qs, L, lane, ka_o, kb_o);          // Qs already holds q * scale
    wsum2(pa, pb, Gs, L, lane, va_o, vb_o);
    dqkv[((long)b * L + j0) * 3 * W + W + h * HD + lane] = ka_o;
    dqkv[((long)b * L + j0) * 3 * W + 2 * W + h * HD + lane] = va_o;
    if (two) {
      dqkv[((long)b * L + j1) * 3 * W + W + h * HD + lane] = kb_o;
      dqkv[((long)b * L + j1) * 3 * W + 2 * W + h * HD + lane] = vb_o;
    }
  }
}

// ---- Gram matrix pieces.  Ft[w][l] = T[b][1+l][w] for l < L-1, zero up to Lp (one image)
__global__ __launch_bounds__(256) void tokens_t_kernel(const float* __restrict__ T, float* __restrict__ Ft, int L, int W, int Lp) {
  const long total = (long)W * Lp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int l = (int)(i % Lp), w = (int)(i / Lp);
    Ft[i] = l < L - 1 ? T[(long)(l + 1) * W + w] : 0.f;
  }
}
// R = G - Gref (in place over G) and |R|_F in two deterministic stages: GR_BLOCKS blocks sum fixed contiguous chunks, one wave
// folds their partial sums in order
constexpr int GR_BLOCKS = 64;
__global__ __launch_bounds__(256) void gram_residual_kernel(float* __restrict__ G, const float* __restrict__ Gref, float* __restrict__ part, long n) {
  __shared__ float red[4];
  const long per = (n + GR_BLOCKS - 1) / GR_BLOCKS, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float r = G[i] - Gref[i];
    G[i] = r;
    s += r * r;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void gram_norm_kernel(const float* __restrict__ part, float* __restrict__ nrm, float* __restrict__ loss) {
  float t = 0.f;
  for (int i = 0; i < GR_BLOCKS; ++i) t += part[i];
  *nrm = sqrtf(t);
  *loss = *nrm;
}
// dT[0][:] = 0 ; dT[1+l][:] = (2 * scale / |R|) * FR[l][:]      (d |F^T F - Gref|_F / dF = 2 F R / |R| for symmetric R)
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ FR, const float* __restrict__ nrm, float* __restrict__ dT, int L,
                                                       int W, float scale) {
  const long total = (long)L * W;
  const float k = *nrm > 0.f ? 2.0f * scale / *nrm : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int l = (int)(i / W);
    dT[i] = l == 0 ? 0.f : k * FR[i - W];
  }
}

struct VBlock {
  float *ln1g, *ln1b, *ln2g, *ln2b, *win, *bin, *wo, *bo, *wfc, *bfc, *wp, *bp;
  PConv in, out, fc, proj;
};

inline dim3 egrid(long total) { return dim3(ew_grid(total)); }
constexpr size_t ATTN_LDS_FWD = (size_t)(2 * LMAX * HP + 8 * LMAX) * 4;
constexpr size_t ATTN_LDS_DQ = ATTN_LDS_FWD;
constexpr size_t ATTN_LDS_DKV = (size_t)(2 * LMAX * HP + 16 * LMAX + 2 * LMAX) * 4;
// slices of the query rows / keys per (image, head): fill the 256 CUs in ONE round (a function of the launch only: every
// output element is produced by the same per-row arithmetic whichever slice it falls in)
inline int attn_slices(int heads, int B) {
  int z = 256 / (heads * B);
  return z < 1 ? 1 : (z > 4 ? 4 : z);
}

}  // namespace

struct hedit_vit : ParamStore {
  hedit_vit_cfg cfg;
  int L = 0, P = 0;
  float *conv_w = nullptr, *cls = nullptr, *pos = nullptr, *lnpg = nullptr, *lnpb = nullptr;
  PConv conv;
  std::vector<VBlock> blocks;
  bool finalized = false;
};

namespace {

struct BTape { float *Tin, *st1, *qkv, *A, *lse, *Tmid, *st2, *H; };

int ln_fwd(PF& f, const float* x, const float* g, const float* b, long rows, int W, float** y, float** stats) {
  TRY(palloc(f, y, (size_t)rows * W));
  TRY(palloc(f, stats, (size_t)rows * 2));
  if (!f.dry()) {
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, f.st, x, g, b, *y, *stats, rows, W);
    LAUNCH_CHECK();
  }
  return HEDIT_OK;
}

// the M rows of a [M][C] fp32 tensor as a split operand and through one linear layer: out raw [M][N]
int lin(PF& f, const float* x, int C, int op, const float* q, const float* z, const PConv& c, bool dgrad, long M, float** out) {
  bf16_t* A;
  TRY(op_split(f, x, C, op, nullptr, q, 0, z, 0, 1, 1, M, &A));
  TRY(pgemm(f, A, c, dgrad, 0, 1, 1, M, out));
  f.ar.free(A);
  return HEDIT_OK;
}

// img [B][3][R][R] (CLIP-normalised).  gram_out: write the Gram matrices [B][W][W]; else gref -> loss [B], d_img
int run(hedit_vit* h, const float* img, const float* gref, long gref_stride, int B, float scale, float* gram_out, float* loss, float* d_img,
        void* ws, size_t ws_bytes, hipStream_t st, bool dry, bool want_grad, size_t* peak) {
  PF f{B, st, Arena{}};
  f.ar.dry = dry;
  f.ar.base = reinterpret_cast<char*>(ws);
  f.ar.cap = ws_bytes;
  const int W = h->cfg.width, L = h->L, R = h->cfg.input_resolution, p = h->cfg.patch_size, heads = h->cfg.heads;
  const int K0 = 3 * p * p;
  const long Mp = (long)B * (L - 1), M = (long)B * L;
  const float ascale = 1.0f / sqrtf((float)(W / heads));
  const bool grad = want_grad;
  float *X0, *E, *T0, *T, *stp;
  TRY(palloc(f, &X0, (size_t)Mp * K0));
  if (!dry) { hipLaunchKernelGGL(patchify_kernel, egrid(Mp * K0), dim3(256), 0, st, img, X0, B, R, p, 0); LAUNCH_CHECK(); }
  TRY(lin(f, X0, K0, P_COPY, nullptr, nullptr, h->conv, false, Mp, &E));
  f.ar.free(X0);
  TRY(palloc(f, &T0, (size_t)M * W));
  if (!dry) { hipLaunchKernelGGL(tokens_kernel, egrid(M * W), dim3(256), 0, st, E, h->cls, h->pos, T0, B, L, W); LAUNCH_CHECK(); }
  f.ar.free(E);
  TRY(ln_fwd(f, T0, h->lnpg, h->lnpb, M, W, &T, &stp));
  if (!grad) { f.ar.free(T0); f.ar.free(stp); }
  std::vector<BTape> tape(h->blocks.size());
  for (size_t i = 0; i < h->blocks.size(); ++i) {
    const VBlock& k = h->blocks[i];
    BTape& t = tape[i];
    float *a, *raw, *m;
    t.Tin = T;
    TRY(ln_fwd(f, T, k.ln1g, k.ln1b, M, W, &a, &t.st1));
    TRY(lin(f, a, W, P_COPY, nullptr, nullptr, k.in, false, M, &t.qkv));
    f.ar.free(a);
    TRY(palloc(f, &t.A, (size_t)M * W));
    TRY(palloc(f, &t.lse, (size_t)B * heads * L));
    if (!dry) {
      hipLaunchKernelGGL(attn_fwd_kernel, dim3(heads, B, attn_slices(heads, B)), dim3(256), ATTN_LDS_FWD, st, t.qkv, k.bin, t.A, t.lse, L, W, ascale);
      LAUNCH_CHECK();
    }
    TRY(lin(f, t.A, W, P_COPY, nullptr, nullptr, k.out, false, M, &raw));
    TRY(palloc(f, &t.Tmid, (size_t)M * W));
    if (!dry) { hipLaunchKernelGGL(add_bias_res_kernel, egrid(M * W), dim3(256), 0, st, raw, k.bo, t.Tin, t.Tmid, M * W, W); LAUNCH_CHECK(); }
    f.ar.free(raw);
    TRY(ln_fwd(f, t.Tmid, k.ln2g, k.ln2b, M, W, &m, &t.st2));
    TRY(lin(f, m, W, P_COPY, nullptr, nullptr, k.fc, false, M, &t.H));
    f.ar.free(m);
    TRY(lin(f, t.H, 4 * W, P_QGELU, k.bfc, nullptr, k.proj, false, M, &raw));
    float* Tn;
    TRY(palloc(f, &Tn, (size_t)M * W));
    if (!dry) { hipLaunchKernelGGL(add_bias_res_kernel, egrid(M * W), dim3(256), 0, st, raw, k.bp, t.Tmid, Tn, M * W, W); LAUNCH_CHECK(); }
    f.ar.free(raw);
    if (!grad) {
      f.ar.free(t.Tin); f.ar.free(t.st1); f.ar.free(t.qkv); f.ar.free(t.A); f.ar.free(t.lse); f.ar.free(t.Tmid); f.ar.free(t.st2); f.ar.free(t.H);
    }
    T = Tn;
  }
  // ---- Gram matrix of the patch tokens, per image
  const int Lp = (L - 1 + 63) / 64 * 64;
  float *Ft, *G, *nrm = nullptr, *dT = nullptr, *gpart = nullptr;
  TRY(palloc(f, &Ft, (size_t)W * Lp));
  if (grad) { TRY(palloc(f, &nrm, (size_t)B)); TRY(palloc(f, &gpart, (size_t)GR_BLOCKS)); TRY(palloc(f, &dT, (size_t)M * W)); }
  PConv gc;                       // "weights" = the second activation operand of F^T F / F R
  for (int b = 0; b < B; ++b) {
    const float* Tb = T + (size_t)b * L * W;
    if (!dry) { hipLaunchKernelGGL(tokens_t_kernel, egrid((long)W * Lp), dim3(256), 0, st, Tb, Ft, L, W, Lp); LAUNCH_CHECK(); }
    bf16_t *A1, *A2;
    TRY(op_split(f, Ft, Lp, P_COPY, nullptr, nullptr, 0, nullptr, 0, 1, 1, W, &A1));
    {
      Split3Params s{};
      s.x = Ft; s.ldx = Lp; s.op = P_COPY; s.Kp = split_kp(Lp); s.Cs = split_cs(Lp); s.C = Lp; s.B = 1; s.H = 1; s.W = 1; s.worder = 1;
      TRY(palloc(f, &A2, (size_t)W * s.Kp));
      s.out = A2;
      if (!dry) TRY(split3_launch(s, W, st));
    }
    gc.O = W; gc.I = Lp; gc.k = 1; gc.wf = A2; gc.rows_f = W;
    PF f1 = f;                    // one image: the GEMM's nominal batch logic sees a single "image" of W rows
    f1.B = 1;
    TRY(pgemm(f1, A1, gc, false, 0, 1, 1, W, &G));
    f.ar = f1.ar;
    f.ar.free(A1);
    f.ar.free(A2);
    if (!grad) {
      if (!dry) HIP_TRY(hipMemcpyAsync(gram_out + (size_t)b * W * W, G, (size_t)W * W * sizeof(float), hipMemcpyDeviceToDevice, st));
      f.ar.free(G);
      continue;
    }
    if (!dry) {
      hipLaunchKernelGGL(gram_residual_kernel, dim3(GR_BLOCKS), dim3(256), 0, st, G, gref + (size_t)b * gref_stride, gpart, (long)W * W);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(gram_norm_kernel, dim3(1), dim3(1), 0, st, gpart, nrm + b, loss + b);
      LAUNCH_CHECK();
    }
    // d loss / d F = 2 F R / |R| : [L-1][W] = F [L-1][W] . R [W][W]   (R symmetric: its rows serve as the "weights")
    bf16_t *AF, *AR;
    float* FR;
    TRY(op_split(f, Tb + W, W, P_COPY, nullptr, nullptr, 0, nullptr, 0, 1, 1, L - 1, &AF));
    {
      Split3Params s{};
      s.x = G; s.ldx = W; s.op = P_COPY; s.Kp = split_kp(W); s.Cs = split_cs(W); s.C = W; s.B = 1; s.H = 1; s.W = 1; s.worder = 1;
      TRY(palloc(f, &AR, (size_t)W * s.Kp));
      s.out = AR;
      if (!dry) TRY(split3_launch(s, W, st));
    }
    gc.O = W; gc.I = W; gc.wf = AR; gc.rows_f = W;
    f1 = f;
    f1.B = 1;
    TRY(pgemm(f1, AF, gc, false, 0, 1, 1, L - 1, &FR));
    f.ar = f1.ar;
    f.ar.free(AF);
    f.ar.free(AR);
    f.ar.free(G);
    if (!dry) {
      hipLaunchKernelGGL(gram_bwd_kernel, egrid((long)L * W), dim3(256), 0, st, FR, nrm + b, dT + (size_t)b * L * W, L, W, scale);
      LAUNCH_CHECK();
    }
    f.ar.free(FR);
  }
  f.ar.free(Ft);
  f.ar.free(T);
  if (!grad) {
    if (peak) *peak = f.ar.peak;
    return HEDIT_OK;
  }
  f.ar.free(nrm);
  f.ar.free(gpart);
  // ---- backward through the blocks
  for (int i = (int)h->blocks.size() - 1; i >= 0; --i) {
    const VBlock& k = h->blocks[i];
    BTape& t = tape[i];
    float *dG, *dm, *dTm, *dAo, *dqkv, *da, *dTi;
    TRY(lin(f, dT, W, P_COPY, nullptr, nullptr, k.proj, true, M, &dG));                          // d QuickGELU output
    TRY(lin(f, dG, 4 * W, P_QGELU_GRAD, k.bfc, t.H, k.fc, true, M, &dm));                        // d LN2 output
    f.ar.free(dG);
    f.ar.free(t.H);
    TRY(palloc(f, &dTm, (size_t)M * W));
    if (!dry) { hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, t.Tmid, dm, k.ln2g, t.st2, dT, dTm, M, W); LAUNCH_CHECK(); }
    f.ar.free(dm); f.ar.free(dT); f.ar.free(t.Tmid); f.ar.free(t.st2);
    TRY(lin(f, dTm, W, P_COPY, nullptr, nullptr, k.out, true, M, &dAo));                         // d attention output
    TRY(palloc(f, &dqkv, (size_t)M * 3 * W));
    if (!dry) {
      hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(heads, B, attn_slices(heads, B)), dim3(256), ATTN_LDS_DQ, st, t.qkv, k.bin, t.A, dAo, t.lse, dqkv, L, W, ascale);
      LAUNCH_CHECK();
      hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(heads, B, attn_slices(heads, B)), dim3(256), ATTN_LDS_DKV, st, t.qkv, k.bin, t.A, dAo, t.lse, dqkv, L, W, ascale);
      LAUNCH_CHECK();
    }
    f.ar.free(dAo); f.ar.free(t.qkv); f.ar.free(t.A); f.ar.free(t.lse);
    TRY(lin(f, dqkv, 3 * W, P_COPY, nullptr, nullptr, k.in, true, M, &da));                      // d LN1 output
    f.ar.free(dqkv);
    TRY(palloc(f, &dTi, (size_t)M * W));
    if (!dry) { hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, t.Tin, da, k.ln1g, t.st1, dTm, dTi, M, W); LAUNCH_CHECK(); }
    f.ar.free(da); f.ar.free(dTm); f.ar.free(t.Tin); f.ar.free(t.st1);
    dT = dTi;
  }
  // ln_pre, class token, patch embedding
  float *dT0, *dE, *dX0;
  TRY(palloc(f, &dT0, (size_t)M * W));
  if (!dry) { hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, T0, dT, h->lnpg, stp, (const float*)nullptr, dT0, M, W); LAUNCH_CHECK(); }
  f.ar.free(dT); f.ar.free(T0); f.ar.free(stp);
  TRY(palloc(f, &dE, (size_t)Mp * W));
  if (!dry) { hipLaunchKernelGGL(drop_cls_kernel, egrid(Mp * W), dim3(256), 0, st, dT0, dE, B, L, W); LAUNCH_CHECK(); }
  f.ar.free(dT0);
  TRY(lin(f, dE, W, P_COPY, nullptr, nullptr, h->conv, true, Mp, &dX0));
  f.ar.free(dE);
  if (!dry) { hipLaunchKernelGGL(patchify_kernel, egrid(Mp * K0), dim3(256), 0, st, d_img, dX0, B, R, p, 1); LAUNCH_CHECK(); }
  f.ar.free(dX0);
  if (peak) *peak = f.ar.peak;
  return HEDIT_OK;
}

}  // namespace

extern "C" {

int hedit_vit_create(const hedit_vit_cfg* cfg, hedit_vit** out) try {
  ARG_CHECK(cfg && out, "null");
  ARG_CHECK(cfg->width % 64 == 0 && cfg->heads > 0 && cfg->width / cfg->heads == 64, "vit: head dimension must be 64");
  ARG_CHECK(cfg->layers >= 1 && cfg->input_resolution % cfg->patch_size == 0 && (3 * cfg->patch_size * cfg->patch_size) % 64 == 0,
            "vit: layers / patch geometry");
  const int P = cfg->input_resolution / cfg->patch_size, L = P * P + 1;
  ARG_CHECK(L <= LMAX, "vit: at most 200 tokens");
  TRY(gemm_prepare());
  hedit_vit* h = new hedit_vit();
  h->cfg = *cfg;
  h->P = P; h->L = L;
  const int W = cfg->width, p = cfg->patch_size;
  h->conv_w = f32conv(h, "visual.conv1.weight", W, 3, p);
  h->cls = vec(h, "visual.class_embedding", W);
  h->pos = dalloc<float>(h, (size_t)L * W);
  add_slot(h, "visual.positional_embedding", 0, h->pos, (size_t)L * W, L, W, 2, L, W, 1, 1);
  h->lnpg = vec(h, "visual.ln_pre.weight", W);
  h->lnpb = vec(h, "visual.ln_pre.bias", W);
  auto mat = [&](const std::string& name, int O, int I) {
    float* d = dalloc<float>(h, (size_t)O * I);
    add_slot(h, name, 0, d, (size_t)O * I, O, I, 2, O, I, 1, 1);
    return d;
  };
  for (int i = 0; i < cfg->layers; ++i) {
    const std::string pre = "visual.transformer.resblocks." + std::to_string(i);
    VBlock k{};
    k.ln1g = vec(h, pre + ".ln_1.weight", W); k.ln1b = vec(h, pre + ".ln_1.bias", W);
    k.win = mat(pre + ".attn.in_proj_weight", 3 * W, W); k.bin = vec(h, pre + ".attn.in_proj_bias", 3 * W);
    k.wo = mat(pre + ".attn.out_proj.weight", W, W); k.bo = vec(h, pre + ".attn.out_proj.bias", W);
    k.ln2g = vec(h, pre + ".ln_2.weight", W); k.ln2b = vec(h, pre + ".ln_2.bias", W);
    k.wfc = mat(pre + ".mlp.c_fc.weight", 4 * W, W); k.bfc = vec(h, pre + ".mlp.c_fc.bias", 4 * W);
    k.wp = mat(pre + ".mlp.c_proj.weight", W, 4 * W); k.bp = vec(h, pre + ".mlp.c_proj.bias", W);
    h->blocks.push_back(k);
  }
  if (h->alloc_failed) {
    hedit_set_error("hipMalloc failed while creating the ViT prefix");
    store_free(h);
    delete h;
    return HEDIT_ERR_HIP;
  }
  *out = h;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

void hedit_vit_destroy(hedit_vit* h) try {
  if (!h) return;
  store_free(h);
  delete h;
} catch (...) { (void)hedit_abi_catch(); }

int hedit_vit_num_params(const hedit_vit* h) { return h ? (int)h->slots.size() : 0; }
const char* hedit_vit_param_name(const hedit_vit* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
} catch (...) { (void)hedit_abi_catch(); return nullptr; }
int hedit_vit_param_shape(const hedit_vit* h, int i, int* ndim, int* dims4) try {
  ARG_CHECK(h && ndim && dims4 && i >= 0 && i < (int)h->slots.size(), "param index");
  *ndim = h->slots[i].ndim;
  for (int k = 0; k < 4; ++k) dims4[k] = h->slots[i].dims[k];
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }
int hedit_vit_load(hedit_vit* h, const char* name, const float* w, size_t numel, void* stream) try {
  ARG_CHECK(h && name && w, "null");
  h->finalized = false;
  return store_load(h, "ViT", name, w, numel, reinterpret_cast<hipStream_t>(stream));
} catch (...) { return hedit_abi_catch(); }
int hedit_vit_missing(const hedit_vit* h) { return h ? store_missing(h) : -1; }

int hedit_vit_finalize(hedit_vit* h, void* stream) try {
  ARG_CHECK(h, "null");
  if (store_missing(h) != 0) {
    hedit_set_error("ViT has " + std::to_string(store_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int W = h->cfg.width, p = h->cfg.patch_size;
  // conv1 weight [W][3][p][p] flattened = a linear map over (c, ky, kx): k = 1 with I = 3 p^2
  TRY(make_pconv(h, h->conv, h->conv_w, nullptr, W, 3 * p * p, 1, 0, 0, st));
  for (VBlock& k : h->blocks) {
    TRY(make_pconv(h, k.in, k.win, nullptr, 3 * W, W, 1, 0, 0, st));
    TRY(make_pconv(h, k.out, k.wo, nullptr, W, W, 1, 0, 0, st));
    TRY(make_pconv(h, k.fc, k.wfc, nullptr, 4 * W, W, 1, 0, 0, st));
    TRY(make_pconv(h, k.proj, k.wp, nullptr, W, 4 * W, 1, 0, 0, st));
  }
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&attn_fwd_kernel), (int)ATTN_LDS_FWD)) return rc;
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&attn_bwd_dq_kernel), (int)ATTN_LDS_DQ)) return rc;
  if (int rc = hedit_dyn_lds(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel), (int)ATTN_LDS_DKV)) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  if (h->alloc_failed) { hedit_set_error("hipMalloc failed while packing the ViT weights"); return HEDIT_ERR_HIP; }
  h->finalized = true;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

size_t hedit_vit_workspace_bytes(hedit_vit* h, int B) try {
  if (!h || B < 1) return 0;
  size_t peak = 0;
  if (run(h, nullptr, nullptr, 0, B, 1.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, true, &peak) != HEDIT_OK) return 0;
  return peak + 4096;
} catch (...) { (void)hedit_abi_catch(); return 0; }

/* image fp32 [B][3][R][R], CLIP-normalised and resized -> gram fp32 [B][W][W]: F^T F of the patch tokens after the last
 * kept block (the style reference's side of get_gram_matrix_residual, base_clip.py:60-65) */
int hedit_vit_gram(hedit_vit* h, const float* image, int B, float* gram, void* workspace, size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && image && gram && workspace && B >= 1, "vit_gram args");
  if (!h->finalized) { hedit_set_error("call hedit_vit_finalize after loading the parameters"); return HEDIT_ERR_STATE; }
  return run(h, image, nullptr, 0, B, 1.f, gram, nullptr, nullptr, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), false,
             false, nullptr);
} catch (...) { return hedit_abi_catch(); }

/* loss[b] = | Gram(image_b) - gram_ref |_F and d_image = d(scale * sum_b loss[b]) / d image in ONE call
 * (torch.linalg.norm(get_gram_matrix_residual(x)) + torch.autograd.grad of inversion/h_edit.py:170-179).
 * gram_ref [W][W] shared by the batch (ref_per_image = 0) or [B][W][W]. */
int hedit_vit_gram_fwd_bwd(hedit_vit* h, const float* image, const float* gram_ref, int ref_per_image, int B, float scale, float* loss,
                           float* d_image, void* workspace, size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && image && gram_ref && loss && d_image && workspace && B >= 1, "vit_gram_fwd_bwd args");
  if (!h->finalized) { hedit_set_error("call hedit_vit_finalize after loading the parameters"); return HEDIT_ERR_STATE; }
  const long stride = ref_per_image ? (long)h->cfg.width * h->cfg.width : 0;
  return run(h, image, gram_ref, stride, B, scale, nullptr, loss, d_image, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream),
             false, true, nullptr);
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
