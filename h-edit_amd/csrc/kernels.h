// Internal launcher interface between the kernel files and the UNet executor / C-ABI.
#pragma once
#include "common.h"

// ---------------------------------------------------------------- gemm.hip
struct GemmParams {
  const bf16_t* A;   // mode 0: [M][lda]; conv modes: NHWC [B][Hin][Win][Cin]
  const bf16_t* W;   // [N][K], K-contiguous (conv: k = (ky*3+kx)*Cin + ci)
  int M, N, K;
  int lda;
  int mode;          // 0 linear/1x1, 1 conv3x3 s1 p1, 2 conv3x3 s2 p1, 3 conv3x3 p1 on 2x nearest upsample
  int Hin, Win, Cin, Hout, Wout;
  const float* bias;        // [N] or null
  const bf16_t* residual;   // [M][ldr] or null
  int ldr;
  bf16_t* C;
  int ldc;
  float* partial;    // set by gemm_launch
  int splits, kt_per_split;
  int n_fastest;     // tile order, set by gemm_launch
  const bf16_t* zeros;   // 16-byte-aligned zero page (>= 16 B), set by gemm_launch (residual prefetch of ragged tiles)
  unsigned a_bytes, w_bytes;   // sizes of the A and W operands for the buffer descriptors, set by gemm_launch
  int geglu;             // FF1: rows interleaved (value16|gate16), output [M][N/2] = v * gelu(g)
  int asym;              // mode 2 only: 1 = pad (0,1,0,1) instead of 1 all round (the VAE encoder's
                         // Downsample2D(padding=0): taps at rows 2oy .. 2oy+2)
  float* raw_f32;        // if set: write the plain fp32 products [M][N] here (no bias / residual / bf16 C)
  int chunk_kt;          // canonical K-chunking (gemm_canonical_chunk), in K-tiles of 64; 0 = one plain chain
  // optional: the consumer is LayerNorm(C) -> ln_out.  gemm_launch honours it ONLY when the launch is split-K (the reduce
  // pass then normalises too, one launch less) and reports it through *ln_done; otherwise the caller normalises itself
  const float *ln_gamma, *ln_beta;
  bf16_t* ln_out;
  float ln_eps;
  int* ln_done;
  // optional (conv modes, N % 128 == 0, M % 128 == 0): GroupNorm pair statistics of the stored output, [M / 128][N / 2][2] fp32
  // (gnstat.h): written by whichever kernel writes the bf16 tile, same bits for every execution form
  float* gn_part;
  // the operands are bfloat16 whatever the storage format of the build (the split-bf16 triples of pnet.hip; raw_f32 output only).
  // No effect in the bfloat16 build.
  int op_bf16;
};
#ifndef GEMM_NOMINAL_BATCH
#define GEMM_NOMINAL_BATCH 4   // the canonical chunking is sized for this many rows of the batch dimension
#endif
int gemm_prepare();   // allocates the zero page (call once outside any timed / captured region)
int gemm_pick_bn(int N);
int gemm_pick_splits(int M, int N, int K, int force);      // explicit split count clamped to the K-tiles (C entry points)
int gemm_canonical_chunk(int M_nom, int N_nom, int K);     // batch-independent definition of the fp32 summation order
int gemm_plan_splits(int M, int N, int K, int chunk_kt);   // slabs this launch uses to execute it (1 = in registers)
size_t gemm_partial_bytes(int M, int N, int splits);
int gemm_launch(GemmParams p, int splits, float* partial_ws, hipStream_t st);
// pgemm.hip: the persistent form of the linear (mode 0) launches -- same bits as igemm_kernel, chosen by shape inside gemm_launch
bool pgemm_supported(const GemmParams& p, int splits, int bn);
int pgemm_launch(const GemmParams& p, int bn, hipStream_t st);
// pconv.hip: the persistent form of the row-sharing 3x3 launches (kernel modes 4 / 5) -- same bits, chosen by shape inside gemm_launch
bool pconv_supported(const GemmParams& p, int splits, int bn);
int pconv_launch(const GemmParams& p, int bn, hipStream_t st);

// ---------------------------------------------------------------- ffn.hip
// The token-local tail of a transformer block in one kernel (C = ffn_fused_channels() only):
//     t2 = a W_pre^T + bias_pre + x          (only if a != nullptr: attn2.to_out + residual)
//     t3 = t2 + FF2(GEGLU(FF1(LayerNorm(t2))))          (t2 = x without the leading layer)
//     out = t3 W_post^T + bias_post + r2     (only with the leading layer: proj_out + residual)
// The weights come as a pre-packed stream (ffn_pack_launch, one call per layer: which = 0 leading linear [C][C],
// 1 ff.net.0.proj.weight [8C][C] -- call it first or alone, it also zeroes the padding --, 2 ff.net.2.weight [C][4C],
// 3 trailing linear [C][C]) and the FF1 bias in packed order (ffn_pack_bias_launch).
struct FfnParams {
  const bf16_t* x; long ldx;     // [M][ldx]: residual of the leading layer, or LayerNorm input AND residual without it
  const float* gamma; const float* beta; float eps;
  const bf16_t* stream;          // ffn_stream_bytes(pre, post)
  const float* bias1p;           // ffn_bias_bytes()
  const float* bias2;            // [C]
  bf16_t* out; long ldo;
  int M, C;
  const bf16_t* a; long lda;     // leading layer input [M][lda] or nullptr
  const bf16_t* r2; long ldr2;   // residual of the trailing layer [M][ldr2] (with the leading layer only)
  const float* bias_pre; const float* bias_post;
};
// ---------------------------------------------------------------- linchain.hip
// The projections around an attention of the C = 320 level as one kernel:
//   mid = a W_pre^T + bias_pre + r1 (written to out_mid);  out = LN(mid) W_0^T           [gn_ss == nullptr]
//   mid = (a * scale + shift) W_pre^T + bias_pre;  q, k = LN(mid) W_0^T, LN(mid) W_1^T;  out = (LN(mid) W_2^T)^T   [gn_ss set]
struct LinChainParams {
  const bf16_t* a; long lda;
  const bf16_t* r1; long ldr1;
  const float* bias_pre;
  const float* gamma; const float* beta; float eps;
  const bf16_t* stream;          // lin_chain_stream_bytes(2 / 4 layers)
  bf16_t* out_mid; long ldmid;
  bf16_t* out; long ldo;         // [M][ldo], or with gn_ss: [C][ldo] (transposed)
  int M, C;
  const float* gn_ss; int rows_per_image;      // [images][C][2] (scale, shift) of the GroupNorm in front
  bf16_t* out_q; long ldq; bf16_t* out_k; long ldk;
};
int lin_chain_launch(const LinChainParams& c, hipStream_t st);
// test entry (tests/test_gpu_chain_hazard.py): sched 0 = the product schedule, 1 = every wait drained to vmcnt(0) (the reference bits)
int lin_chain_launch_sched(const LinChainParams& c, int sched, hipStream_t st);
size_t lin_chain_stream_bytes(int layers);        // layers = 2 (to_out, to_q) or 4 (proj_in, to_q, to_k, to_v)
// w [C][C] fp32 -> layer `layer` of the stream (layer 0 takes its input from HBM: natural k order; the others read an
// accumulator: register order).  The stream is CYCLIC (layers * 20 iterations of 10 KB, no padding, no overrun: the
// kernel wraps around to iteration 0 for the next tile), and every byte of it is written by exactly one layer's pack
// call -- a stream is complete once all `layers` calls have run; nothing is zeroed by anybody.
// Largest launch: M * ld * 2 bytes of every row tensor below 2 GiB (32-bit buffer offsets), i.e. M < 1 677 721 rows of
// 320 channels with dense rows = 409 rows of 64 x 64 tokens (the bench runs 120); lin_chain_launch refuses more.
int lin_chain_pack_launch(const float* w, int layer, float scale, int layers, bf16_t* stream, hipStream_t st);
int ffn_fused_channels();
size_t ffn_stream_bytes(int pre, int post);
size_t ffn_bias_bytes();
int ffn_pack_launch(const float* w, int which, int pre, int post, bf16_t* stream, hipStream_t st);
int ffn_pack_bias_launch(const float* b1, float* out, hipStream_t st);
int ffn_fused_launch(const FfnParams& f, hipStream_t st);

#define HEDIT_MAXW 77
#define HEDIT_WPAD 96
#define HEDIT_CTXP 80           // context rows per batch item in K / V^T buffers (77 + zero pad)

// ---------------------------------------------------------------- norm.hip
// GroupNorm over NHWC bf16 [B][HW][C] with G groups; deterministic two-stage statistics.
size_t groupnorm_ws_bytes(int B, int HW, int C);
int groupnorm_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, int B,
                     int HW, int C, int G, float eps, int silu, float* ws, hipStream_t st,
                     float* stats = nullptr);   // stats: optional [B][G][2] (mean, rstd) for the backward pass
// GroupNorm on producer-side pair statistics (gnstat.h; GemmParams::gn_part of the launch(es) that wrote x = [a | b])
bool groupnorm_from_parts_supported(int HW, int C, int G);
size_t groupnorm_from_parts_ws_bytes(int B);
int groupnorm_from_parts_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                float eps, int silu, const float* part_a, int ca, const float* part_b, int cb, float* ws,
                                hipStream_t st);
int groupnorm_affine_launch(const bf16_t* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps,
                            float* ws, hipStream_t st, const float** ss_out);   // *ss_out: [B][C] float2 (scale, shift), inside ws
int layernorm_launch(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, long rows,
                     int C, float eps, hipStream_t st);
// reduce + LayerNorm in one launch behind a split-K GEMM whose output feeds a LayerNorm (small batches): out = the GEMM's
// bf16 result (sum of the fp32 slabs in order + bias, rounded, + residual, rounded -- splitk_reduce_kernel's arithmetic), y = LayerNorm(out)
// by the kernel layernorm_launch would pick.  Same bits as the two launches.
bool splitk_reduce_ln_supported(int C);
int splitk_reduce_ln_launch(const float* partial, int splits, const float* bias, const bf16_t* residual, int ldr, bf16_t* out,
                            bf16_t* y, const float* gamma, const float* beta, long rows, int C, float eps, hipStream_t st);
int geglu_launch(const bf16_t* x, bf16_t* y, long rows, int inner, hipStream_t st);
// y [rows][ca+cb] = [a | b]; a == nullptr: the left part is already in place, only b is copied
int concat_launch(const bf16_t* a, int ca, const bf16_t* b, int cb, bf16_t* y, long rows, hipStream_t st);
int f32_to_bf16_launch(const float* x, bf16_t* y, long n, hipStream_t st);
int ctx_pad_launch(const float* x, bf16_t* y, int B, int dim, hipStream_t st);
// out[n] = sum_k W[n][k] * act(x[k]) + b0[n] + b1[n]   (act = SiLU if silu), x/out fp32, W bf16
int gemv_launch(const bf16_t* W, const float* x, const float* b0, const float* b1, float* out,
                int N, int K, int silu, hipStream_t st);
int copy_rows_launch(bf16_t* x, const int* src, int B, long row_elems, hipStream_t st);   // x[b] <- x[src[b]]
int timestep_embed_launch(float t, float* out, int dim, hipStream_t st);
int timestep_embed_ddpm_launch(float t, float* out, int dim, hipStream_t st);   // [sin | cos], exponent / (half-1)
// conv_in: x fp32 NCHW [B][Cin<=8][H][W], w fp32 [Cout][Cin][3][3] -> y NHWC bf16 [B][H][W][Cout]
int conv_in_launch(const float* x, const float* w, const float* bias, bf16_t* y, int B, int Cin,
                   int H, int W, int Cout, hipStream_t st);
// conv_out: x NHWC bf16 [B][H][W][C], w bf16 [Cout<=4][9][C] -> y fp32 NCHW [B][Cout][H][W]
int conv_out_launch(const bf16_t* x, const bf16_t* w, const float* bias, float* y, int B, int H,
                    int W, int C, int Cout, hipStream_t st);
// fp32 [B*HW][ld] (ld >= 4, the first Cout <= 4 columns) + bias -> fp32 NCHW [B][Cout][HW]: tail of the MFMA route of conv_out
int rows_to_nchw_launch(const float* src, const float* bias, float* dst, int B, long HW, int ld, int Cout, hipStream_t st);
// VAE helpers: row softmax of fp32 scores (p = softmax(scale * s), bf16), 1x1 channel mixing of a
// small fp32 NCHW tensor (x scaled by pre_scale first), encoder tail (first Cout channels of quant_conv)
int softmax_rows_launch(const float* s, bf16_t* p, long rows, int N, float scale, hipStream_t st);
// several images stacked in one score matrix [rows = B*T][ld = B*T]: softmax inside each image's diagonal block, 0 elsewhere
int softmax_blockdiag_launch(const float* s, bf16_t* p, long rows, int T, int ld, float scale, hipStream_t st);
int mix1x1_nchw_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, long HW,
                       float pre_scale, hipStream_t st);
int quant_mean_launch(const bf16_t* h, const float* w, const float* bias, float* y, int B, long HW, int Cm, int Cout, hipStream_t st);
// weight packing (fp32 source tensors in torch layouts -> bf16 GEMM layouts), optional scale
int pack_linear_launch(const float* w, bf16_t* out, long n, float scale, hipStream_t st);
int pack_conv3x3_launch(const float* w_oihw, bf16_t* out, int O, int I, hipStream_t st);
// GEGLU row interleave: out row t*32+u <- in row (u<16 ? t*16+u : half + t*16+u-16); K = 1 for the bias
int pack_geglu_rows_launch(const float* w, bf16_t* out_bf16, float* out_f32, int rows, int K, hipStream_t st);

// ---------------------------------------------------------------- grad.hip (decoder input-gradient pieces)
// GroupNorm(+SiLU) backward: x = the forward input, stats = (mean, rstd) [B][G][2] saved by
// groupnorm_launch, dy = gradient w.r.t. the output; dx = input gradient (+ add, if given)
size_t groupnorm_bwd_ws_bytes(int B, int HW, int C);
int groupnorm_bwd_launch(const bf16_t* x, const bf16_t* dy, const bf16_t* add, bf16_t* dx, const float* gamma,
                         const float* beta, const float* stats, int B, int HW, int C, int G, int silu, float* ws,
                         hipStream_t st);
// ds = scale * p * (dp - rowsum(dp * p))
int softmax_bwd_launch(const bf16_t* p, const float* dp, bf16_t* ds, long rows, int N, float scale, hipStream_t st);
int transpose_bf16_launch(const bf16_t* src, bf16_t* dst, int R, int C, hipStream_t st);   // [R][C] -> [C][R]
// du [B][2H][2W][C] -> dx [B][H][W][C]: backward of the 2x nearest upsample
int sum2x2_launch(const bf16_t* du, bf16_t* dx, int B, int H, int W, int C, hipStream_t st);
// weights of the input-gradient GEMMs: conv3x3 OIHW -> bf16 [I][9][O] taps flipped; linear [O][I] -> bf16 [I][O];
// k x k OIHW fp32 -> IOHW fp32 taps flipped
int pack_conv3x3_dgrad_launch(const float* w_oihw, bf16_t* out, int O, int I, hipStream_t st);
int pack_linear_t_launch(const float* w, bf16_t* out, int O, int I, hipStream_t st);
int flip_oihw_launch(const float* w, float* out, int O, int I, int k, hipStream_t st);

// ---------------------------------------------------------------- attn.hip
struct SelfAttnParams {
  const bf16_t* q; int ldq;     // [B*N][ldq], head h at column h*d ; pre-scaled by scale*log2(e)
  const bf16_t* k; int ldk;     // [B*N][ldk]
  const bf16_t* vt; long ldvt;  // V^T: [heads*d][B*N]  (row = h*d + dd, col = b*N + token)
  bf16_t* out; int ldo;         // [B*N][ldo]
  int B, N, heads, d;
  const int* qk_src;            // [B] batch index whose q,k are used (P2P self-replace) or null
  const int* kv_src;            // [B] batch index whose k,v are used (MasaCtrl mutual self-attention) or null
  int test_flags;               // hedit_test_flags() at launch (0 in the product): bit 0 drained ring waits, bit 1 exact pass only
};
int self_attn_launch(const SelfAttnParams& p, hipStream_t st);

struct CrossAttnParams {
  const bf16_t* q; int ldq;     // [B*N][ldq] pre-scaled
  const bf16_t* k; int ldk;     // [B*77][ldk]
  const bf16_t* vt; long ldvt;  // [heads*d][B*77]
  bf16_t* out; int ldo;
  int B, N, heads, d;
  // P2P: n_pairs (src,tar) pairs; items not in any pair are plain.  Tables per pair.
  int n_pairs;
  const int* pair_src;          // [n_pairs] batch index of the source-conditional row
  const int* pair_tar;          // [n_pairs]
  const bf16_t* mixT;           // [n_pairs][96][96]: mixT[n][w] = A[w][n]  (P_new = P_src A + bvec*P_tar)
  const float* bvec;            // [n_pairs][96]
  float* store;                 // [n_pairs][2][heads][N][77] fp32 accumulators or null
  const int* singles;           // [n_single] batch rows that are not part of any pair
  int n_single;
};
int cross_attn_launch(const CrossAttnParams& p, hipStream_t st);

// materialised probabilities (the host-language controller hook, hedit_unet_set_attn_hook): probs = softmax(Q K^T) as
// fp32 [B*heads][N][M] (reference layout: ptp_utils.py:98), then out = probs . V.  kstride = rows per batch item of K / V^T
// (N for self-attention, HEDIT_CTXP for the context), M = the keys that count (N or 77).
struct AttnProbsParams {
  const bf16_t* q; int ldq;     // [B*N][ldq] pre-scaled by scale*log2(e)
  const bf16_t* k; int ldk;     // [B*kstride][ldk]
  const bf16_t* vt; long ldvt;  // [heads*d][B*kstride]
  float* probs;                 // [B*heads][N][M]
  bf16_t* out; int ldo;         // [B*N][ldo]
  int B, N, M, kstride, heads, d;
  // attn_self_store_launch only: the rows that are materialised (pair p: pair_src[p], pair_tar[p]) and whose q / k they use
  const int *pair_src, *pair_tar, *qk_src;
};
int attn_probs_launch(const AttnProbsParams& p, hipStream_t st);
int attn_self_store_launch(const AttnProbsParams& p, hipStream_t st);
int attn_apply_launch(const AttnProbsParams& p, hipStream_t st);

// ---------------------------------------------------------------- step.hip
struct StepCoef {
  float sqrt_ab_t, sqrt_1m_ab_t;     // of the current timestep t
  float sqrt_ab_prev;                // sqrt(abar_prev)
  float dir_coef;                    // sqrt(1-abar_prev - eta^2 var)  or sqrt(1-abar_prev) (ddim-inv)
  float noise_coef;                  // eta*sqrt(var) or eta (ddim-inv); 0 when eta == 0
  float w_src, w_hat, w_tar;         // CFG weights
  float coeff;                       // edit coefficient of the correction term
  float w_rec;                       // reconstruction weight (k > 0)
};
// base pass: eps [rows][n_img][elems], rows = [x_o|null, x_e|null, x_o|src, x_e|src] -> x_prev [2][n_img][elems]
int step_base_launch(const float* eps, const float* xt, const float* z, float* x_prev, int n_img,
                     int elems, int eps_rows_per_img, StepCoef c, hipStream_t st);
// inversion step: z = (x_prev - mu) / sigma, x_prev <- mu + sigma z, with step_base's mu (sigma = c.noise_coef)
int step_invert_launch(const float* e_u, const float* e_c, const float* xt, float* x_prev, float* z_out, int n_img,
                       int elems, StepCoef c, hipStream_t st);
int step_update_launch(const float* e_u_src, const float* e_c_src, const float* e_u_tar,
                       const float* e_c_tar, long stride_img, const float* x_k, const float* x_base,
                       float* x_out, int n_img, int elems, int k_gt0, StepCoef c, hipStream_t st);
// style guidance (n-style h_edit.py:160-185): Tweedie x0 / scaling_factor for the decoder, and the update
// x - rho g with g = chain * g_z, rho = rms(correction) / rms(g) * weight (per image)
int step_tweedie_launch(const float* e_u_tar, const float* e_c_tar, long stride_img, const float* x, float* z0,
                        int n_img, int elems, float w_tar, float sqrt_ab, float sqrt_1m_ab, float inv_scale, hipStream_t st);
int step_style_launch(const float* e_u_src, const float* e_c_src, const float* e_u_tar, const float* e_c_tar,
                      long stride_img, const float* x, const float* g_z, float* x_out, int n_img, int elems, float w_hat,
                      float w_tar, float chain, float weight, hipStream_t st);
// out[o][i][x] = sum_j val[i][j] in[o][idx[i][j]][x] over a [outer][n_in][inner] fp32 tensor (tables [n_out][nnz])
int axis_mix_launch(const float* in, float* out, const int* idx, const float* val, int nnz, long outer, int n_in, int n_out, int inner,
                    hipStream_t st);
int local_blend_launch(const float* const* maps, int n_maps, int heads, const float* alpha_layers, const float* alpha_sub,
                       const int* enabled, float* xt, int n_img, int C, int H, int W, float th, float th_sub, hipStream_t st);

// ---------------------------------------------------------------- pnet.hip ("precise" fp32-quality building blocks)
// ops of the fused element-wise stage that produces a split-bf16 GEMM operand (and of act_launch)
enum { P_COPY = 0, P_AFFINE = 1, P_PRELU = 2, P_PRELU_GRAD = 3, P_RELU = 4, P_RELU_GRAD = 5, P_QGELU = 6, P_QGELU_GRAD = 7 };
struct Split3Params {
  const float* x; int ldx;     // fp32 input rows [rows_in][ldx] (NHWC pixels x channels)
  const float* z;              // P_*_GRAD: the pre-activation the mask is taken from (same geometry as x)
  const float* p; const float* q;   // per-channel parameters (q may be null for the activations), or per (image, channel)
  int pq_img;                  // 1: p / q are [B][C]
  int op;
  bf16_t* out; int Kp, Cs, C;  // bf16 [rows_out][Kp]: columns [0,C) hi, [Cs,Cs+C) hi, [2Cs,2Cs+C) lo, rest 0
  int geo;                     // 0: rows_out = rows_in; 1: every second pixel of every second row; 2: zero-stuffed to 2H x 2W
  int B, H, W;                 // INPUT geometry (needed for geo != 0 or pq_img)
  int worder;                  // 1: parts (hi, lo, hi) -- the "weight" side of a product of two activation tensors
};
int split3_launch(const Split3Params& s, long rows_out, hipStream_t st);
int pack_split3_w_launch(const float* w, const float* scale, bf16_t* out, int O, int I, int k, int dgrad, int Cs, int Kp,
                         int rows_out, int perm_hw, int perm_c, hipStream_t st);
int bn_affine_launch(const float* g, const float* b, const float* m, const float* v, float eps, float* p, float* q, int C, int rep,
                     hipStream_t st);
int bn_fold_bias_launch(const float* bias, const float* g, const float* b, const float* m, const float* v, float eps, float* p, float* q,
                        int C, hipStream_t st);
int act_launch(const float* x, const float* p, const float* q, float* y, long total, int C, int op, hipStream_t st);
int se_nslab(int HW);          // pixel slabs of the SE reductions: partial buffers are [B][se_nslab(HW)][C]
int se_pool_launch(const float* u, float* part, int B, int HW, int C, hipStream_t st);
int se_fc_launch(const float* part, const float* bias, const float* w1, const float* w2, float* hbuf, float* sbuf, int B, int HW, int C,
                 int R, hipStream_t st);
int se_combine_launch(const float* u, const float* bias, const float* s, const float* sc, const float* sc_bias, const float* X, int stride,
                      float* Y, int B, int Ho, int Wo, int C, hipStream_t st);
int se_bwd_reduce_launch(const float* dY, const float* u, const float* bias, float* part, int B, int HW, int C, hipStream_t st);
int se_fc_bwd_launch(const float* part, const float* sbuf, const float* hbuf, const float* w1, const float* w2, float* rbuf, int B, int C,
                     int R, int HW, hipStream_t st);
int unit_bwd_combine_launch(const float* dxa, const float* p, const float* dsc, int stride, float* dX, int B, int H, int W, int C,
                            hipStream_t st);
int scale_cols_launch(const float* x, const float* p, float* y, long total, int n, hipStream_t st);
int face_pool_launch(const float* img, float* out, int B, hipStream_t st);
int face_pool_bwd_launch(const float* g, int ldg, float* dimg, int B, hipStream_t st);
int cos_head_launch(const float* raw, const float* bias, const float* ref, int ref_stride, float* feat, float* loss, float* df, int B, int D,
                    float scale, hipStream_t st);
// LPIPS-VGG pieces
int lpips_prep_launch(const float* img, float* out, const float* shift3, const float* scale3, int B, int H, int W, hipStream_t st);
int lpips_prep_bwd_launch(const float* g, int ldg, float* dimg, const float* scale3, int B, int H, int W, hipStream_t st);
int maxpool2_launch(const float* z, float* zp, uint8_t* idx, int B, int H, int W, int C, hipStream_t st);
int maxpool2_bwd_launch(const float* gp, const uint8_t* idx, const float* add, float* G, int B, int H, int W, int C, hipStream_t st);
int lpips_head_launch(const float* z, const float* bias, const float* src, long src_img_stride, const float* w, float* fn_out, float* dpix,
                      float* dF, int B, int HW, int C, float gscale, hipStream_t st);
int lpips_reduce_launch(const float* dpix, float* acc, int B, int HW, int first, hipStream_t st);
