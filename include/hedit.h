/* libhedit_hip.so -- C ABI of the MI355X-native h-Edit hot path.
 *
 * The reference (nktoan/h-edit) is pure Python on third-party CUDA libraries and has NO FFI of
 * its own; the boundary below is therefore build-defined and sits exactly
 * where the reference's Python hands work to "the device":
 *
 *   hedit_unet_forward      replaces  model.unet(x, t, encoder_hidden_states=..,
 *                                     cross_attention_kwargs=..).sample
 *                           call sites text-guided/inversion/p2p_h_edit.py:98,123,245,281,315,458,
 *                                     484,492,613,644,652 and ddpm_inversion.py:130,132
 *                           including the P2P processor/controller body that the reference runs
 *                           inside every attention layer (text-guided/p2p/ptp_utils.py:65-122,
 *                           text-guided/p2p/ptp_classes.py:91-108,135-150,202-227)
 *   hedit_step_base         replaces  CFG mix + reverse_step  (p2p_h_edit.py:614-622 /
 *                                     inversion/inversion_utils.py:84-119)
 *   hedit_step_invert       replaces  one step of the edit-friendly DDPM inversion: z_t = (x_{t-1} - mu_t) / sigma_t and
 *                                     the in-place rewrite x_{t-1} <- mu_t + sigma_t z_t
 *                                     (text-guided/inversion/ddpm_inversion.py:146-162), with hedit_step_base's
 *                                     mu so that the sampler's reconstruction branch retraces it bit for bit
 *   hedit_step_update       replaces  the three CFG mixes, correction, L1 reconstruction pull with
 *                                     its two .item() syncs, and the x_{t-1} update
 *                                     (p2p_h_edit.py:654-692; twins :317-353, :494-514, :125-147)
 *   hedit_unet_set_attn_hook  opens   the reference's hook point for a controller written in the host language: the call
 *                                     `self.controller(attention_probs, is_cross, self.place_in_unet, save_attn)` of
 *                                     text-guided/p2p/ptp_utils.py:98-106 on materialised probabilities (slow path)
 *   hedit_local_blend       replaces  LocalBlend.__call__/get_mask (p2p/ptp_classes.py:44-72)
 *   hedit_local_blend_sub   the same  with substruct_words (p2p/ptp_classes.py:28-38,64-68)
 *   hedit_p2p_plan          carries   the per-call edit rule of the registered controller / editor: Prompt-to-Prompt
 *                                     (ptp_classes.py:194-227), MasaCtrl (masactrl/masactrl.py:53-69: kv_src) and
 *                                     Plug-and-Play (plug_n_play/pnp_utils.py:29-154: qk_first_block, feat_src)
 *   hedit_vae_encode/decode replaces  model.vae.encode(x).latent_dist.mode() / model.vae.decode(z).sample
 *                                     (main_p2p.py:159,263)
 *   hedit_vae_decode_vjp,   replace   the decoder part of torch.autograd.grad(loss, latents) in the style closure
 *   _keep + _backward                 (text-guided-n-style/inversion/h_edit.py:146-185)
 *   hedit_step_tweedie,     replace   reverse_step_pred_x0 and the rho-normalised style update
 *   hedit_step_style                  (inversion_utils.py:128-140, h_edit.py:167-183)
 *   hedit_ddpm_forward      replaces  Model.forward(x, t) of face-swapping/diffusion/diffusion.py:294-341
 *                                     (call sites face-swapping/inversion/h_edit_R.py:71,96,118, sde_inversion.py:121)
 *   hedit_k_*               single-kernel entry points used by the parity tests
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every pointer is a DEVICE pointer unless the
 *     parameter name starts with h_ .  The caller (PyTorch-ROCm host code) owns every buffer
 *     including the workspace; the library owns only the opaque handle and its packed weights.
 *   - stream-ordered: nothing here synchronises the device; `stream` is a hipStream_t.
 *   - returns 0 on success, a negative code otherwise; hedit_last_error() gives the message of
 *     the last failure on the calling thread.  Never throws, never exits.
 *   - activations inside the UNet are NHWC bf16; latents/eps at the boundary are fp32 NCHW,
 *     exactly the tensors the reference loop holds.
 */
#ifndef HEDIT_H
#define HEDIT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEDIT_MAX_LEVELS 8

typedef struct hedit_unet hedit_unet;

/* Architecture of a UNet2DConditionModel (diffusers naming). */
typedef struct {
  int in_channels, out_channels, sample_size;
  int n_levels;
  int block_out_channels[HEDIT_MAX_LEVELS];
  int down_has_attn[HEDIT_MAX_LEVELS]; /* CrossAttnDownBlock2D (1) / DownBlock2D (0)            */
  int up_has_attn[HEDIT_MAX_LEVELS];   /* CrossAttnUpBlock2D (1) / UpBlock2D (0), up-block order */
  int layers_per_block;
  int cross_attention_dim;
  int heads;                           /* "attention_head_dim" of SD-1.x = number of heads      */
  int norm_num_groups;
} hedit_unet_cfg;

/* Per-call Prompt-to-Prompt plan = the state the reference's controller holds, compiled by the
 * host for ONE UNet call (hedit/p2p/plan.py).  All arrays are device memory. */
typedef struct {
  int mode;                 /* 0: controller off; 1: apply edits; 2: apply edits + accumulate store */
  int n_pairs;              /* (source-conditional row, target-conditional row) pairs in the batch  */
  const int32_t* pair_src;  /* [n_pairs] */
  const int32_t* pair_tar;  /* [n_pairs] */
  const int32_t* singles;   /* [n_single] batch rows not in any pair                                */
  int n_single;
  const int32_t* qk_src;    /* [B] self-attention replacement map for layers with <= 32*32 tokens
                               (row b uses q,k of row qk_src[b]); NULL outside the self window      */
  const void* mixT;         /* bf16 [n_pairs][96][96], mixT[n][w] = A[w][n]                         */
  const float* bvec;        /* [n_pairs][96]:  P_new = P_src . A + bvec * P_tar                     */
  float* const* h_store;    /* HOST array of n_store device pointers, one per cross-attention layer
                               with <= 32*32 tokens in call order; each [n_pairs][2][heads][N][77]  */
  int n_store;
  /* MasaCtrl mutual self-attention (text-guided/masactrl/masactrl.py:53-69): in the transformer blocks with
   * index >= kv_first_block (call order, 16 in SD-1.x) row b of every self-attention uses the keys and values
   * of row kv_src[b] (its own queries).  NULL = off. */
  const int32_t* kv_src;    /* [B] */
  int kv_first_block;
  /* Plug-and-Play injection (text-guided/plug_n_play/pnp_utils.py:29-154): the qk_src map above, restricted to the
   * transformer blocks with index >= qk_first_block and to layers with <= qk_max_tokens tokens (0 = the P2P
   * default, 32*32); and feat_src: before conv2 of ResNet block number feat_resblock (call order) row b takes the
   * activations of row feat_src[b], i.e. its conv2 output becomes the source row's (pnp_utils.py:131-140).  NULL = off. */
  int qk_first_block, qk_max_tokens;
  const int32_t* feat_src;  /* [B] */
  int feat_resblock;
  /* The reference's AttentionStore also keeps the SELF maps of the layers with <= 32*32 tokens (ptp_classes.py:135-150).
   * Nothing on the h-Edit path reads them, so the fused path stores them only on request: HOST array of n_store_self
   * device pointers, one per such self-attention layer in call order, each [n_pairs][2][heads][N][N] fp32, accumulated
   * on mode-2 passes (post-edit: inside the self window the target row's map is the source's).  NULL = off. */
  float* const* h_store_self;
  int n_store_self;
} hedit_p2p_plan;

typedef struct {
  float sqrt_ab_t, sqrt_1m_ab_t, sqrt_ab_prev, dir_coef, noise_coef;
  float w_src, w_hat, w_tar, coeff, w_rec;
} hedit_step_coef;

const char* hedit_last_error(void);
int hedit_version(void);

/* ---- UNet ------------------------------------------------------------------------------- */
int hedit_unet_create(const hedit_unet_cfg* cfg, hedit_unet** out);
void hedit_unet_destroy(hedit_unet* h);
/* number of parameters the handle expects; name of the i-th one (diffusers state_dict key) */
int hedit_unet_num_params(const hedit_unet* h);
const char* hedit_unet_param_name(const hedit_unet* h, int i);
/* torch shape of the i-th parameter: *ndim in 1..4, dims4[0..ndim) */
int hedit_unet_param_shape(const hedit_unet* h, int i, int* ndim, int* dims4);
/* hand one fp32 parameter tensor (torch layout, device memory) to the library, which packs it
 * into its own bf16/fp32 GEMM layout on `stream` */
int hedit_unet_load(hedit_unet* h, const char* name, const float* w, size_t numel, void* stream);
/* 1 if parameter i is kept as an unscaled bf16 copy: a bf16-rounded source gives the same packed bits (the weight
 * broadcast of the multi-GPU start-up sends those tensors as bf16, the rest as fp32) */
int hedit_unet_param_bf16_exact(const hedit_unet* h, int i);
/* 0 when every parameter has been loaded, else the count still missing */
int hedit_unet_missing(const hedit_unet* h);
size_t hedit_unet_workspace_bytes(hedit_unet* h, int B, int height, int width);
/* x: fp32 [B][Cin][H][W]; t: scalar timestep shared by the batch; ctx: fp32 [B][77][ctx_dim];
 * eps_out: fp32 [B][Cout][H][W] */
int hedit_unet_forward(hedit_unet* h, const float* x, float t, const float* ctx, int B, int height,
                       int width, const hedit_p2p_plan* plan, float* eps_out, void* workspace,
                       size_t workspace_bytes, void* stream);
/* Host-language attention controller -- the reference's hook point, text-guided/p2p/ptp_utils.py:98-106
 * (`self.controller(attention_probs, is_cross, self.place_in_unet, save_attn)` between the softmax and the product with V).
 * With a hook set, every attention layer of hedit_unet_forward writes its probabilities to HBM as fp32
 * [batch_heads = B * heads][n_query][n_key] (batch-major like `head_to_batch_dim`; n_key = 77 for the context), calls
 * fn -- which may rewrite them in place with work queued on `stream` -- and multiplies the result with V.  place: 0 down,
 * 1 mid, 2 up; layer counts the calls of one forward.  fn returns 0, anything else aborts the forward.  The in-kernel
 * edits of the plan are NOT applied on this path (pass plan = NULL or a mode-0 plan); fn = NULL restores the fused kernels.
 * hedit_unet_workspace_bytes accounts for the probabilities while a hook is set. */
typedef int (*hedit_attn_hook_fn)(void* user, float* probs, int batch_heads, int n_query, int n_key, int is_cross, int place,
                                  int layer, void* stream);
int hedit_unet_set_attn_hook(hedit_unet* h, hedit_attn_hook_fn fn, void* user);
/* number of cross-attention layers with <= 32*32 tokens (= plan.n_store) and their geometry */
int hedit_unet_num_store_layers(const hedit_unet* h, int height, int width);
int hedit_unet_store_layer_info(const hedit_unet* h, int height, int width, int i, int* tokens,
                                int* place /*0 down,1 mid,2 up*/);

/* Sampled launch timing for bench.py's roofline: while enabled, every kernel launch of
 * hedit_unet_forward is bracketed by a HIP event pair on the launch stream (up to max_records
 * launches).  kind: 0 conv3x3 GEMM, 1 linear/1x1 GEMM, 2 self-attention, 3 cross-attention,
 * 4 Group/LayerNorm, 5 other.  total_flops is the ALGORITHMIC count (2MNK; 4 B N Nk C for
 * attention).  Call hedit_prof_collect only after synchronising the stream. */
int hedit_prof_enable(hedit_unet* h, int on, int max_records);
int hedit_prof_reset(hedit_unet* h);
int hedit_prof_collect(hedit_unet* h, int kind, double* total_ms, double* total_flops, int64_t* count);
/* algorithmic HBM bytes of the sampled launches of a class: every operand read once, every result written once
 * (what roofline.traffic, a PMC measurement, is compared with) */
int hedit_prof_collect_bytes(hedit_unet* h, int kind, double* total_bytes);
/* one row of 8 doubles per sampled launch, in launch order: kind, ms, flops, bytes, M, N, K, tag (GEMM launches only:
 * mode | 8 GEGLU | 16 residual | 32 split-K | 64 chunked K); after synchronising the stream */
int hedit_prof_records(hedit_unet* h, double* rows, int max_rows, int* n_rows);

/* ---- sampler steps ------------------------------------------------------------------------
 * Batched tensors are [row][image][elems]: for one image this is exactly the reference layout.
 *   eps of the base pass: rows = 4 -> [x_o|null, x_e|null, x_o|src, x_e|src] (P2P loops),
 *                         rows = 2 -> [x_e|null, x_e|src] (no-P2P loops)
 *   xt / x_prev: [2][n_img][elems] = (x^orig, x^edit);   z: [n_img][elems] (may be NULL)
 *   step_update: the four eps operands are pointers to image 0, image i sits stride_img floats
 *   further; x_k / x_base / x_out: [n_img][elems].
 *   local_blend: h_maps = HOST array of n_maps device pointers to 16x16 cross maps, each
 *   [n_img][2][heads][256][77]; alpha_layers [n_img][2][77]; enabled [n_img] or NULL. */
int hedit_step_base(const float* eps, const float* xt, const float* z, float* x_prev, int n_img,
                    int elems, int eps_rows_per_img, const hedit_step_coef* c, void* stream);
/* inversion step: e_u, e_c, xt, x_prev (in/out), z_out: [n_img][elems]; sigma = c->noise_coef > 0; the CFG mix
 * uses c->w_src (pass e_c = e_u for an unconditional inversion). */
int hedit_step_invert(const float* e_u, const float* e_c, const float* xt, float* x_prev, float* z_out,
                      int n_img, int elems, const hedit_step_coef* c, void* stream);
int hedit_step_update(const float* e_u_src, const float* e_c_src, const float* e_u_tar,
                      const float* e_c_tar, int64_t stride_img, const float* x_k,
                      const float* x_base, float* x_out, int n_img, int elems, int k_gt0,
                      const hedit_step_coef* c, void* stream);
/* Style guidance of the text + style loop (text-guided-n-style/inversion/h_edit.py:160-185), the latent-side
 * arithmetic either side of the decoder / image-encoder pass:
 *   step_tweedie: z0 = (x - sqrt(1-ab) e_tar) / sqrt(ab) * inv_scale, e_tar = e_u_tar + w_tar (e_c_tar - e_u_tar)
 *                 (reverse_step_pred_x0, inversion_utils.py:128-140, and the 1/0.18215 of h_edit.py:173)
 *   step_style:   x_out = x - rho g, g = chain * g_z, rho = rms(correction) / rms(g) * weight per image
 *                 (h_edit.py:179-183; g_z = d loss / d z0 from hedit_vae_decode_vjp behind the image encoder,
 *                 chain = inv_scale / sqrt(ab))
 * eps operands as in step_update: pointer to image 0, image i at + i * stride_img; x / z0 / g_z / x_out [n_img][elems] */
int hedit_step_tweedie(const float* e_u_tar, const float* e_c_tar, int64_t stride_img, const float* x, float* z0,
                       int n_img, int elems, float w_tar, float sqrt_ab, float sqrt_1m_ab, float inv_scale, void* stream);
int hedit_step_style(const float* e_u_src, const float* e_c_src, const float* e_u_tar, const float* e_c_tar,
                     int64_t stride_img, const float* x, const float* g_z, float* x_out, int n_img, int elems,
                     float w_hat, float w_tar, float chain, float weight, void* stream);
/* A sparse linear map along one axis of a [outer][n_in][inner] fp32 tensor: out[o][i][x] = sum_j val[i][j] in[o][idx[i][j]][x]
 * (tables [n_out][nnz], fixed summation order).  With torch's bicubic weights it replaces F.interpolate(im, (224, 224),
 * mode="bicubic") in front of the style encoder (text-guided-n-style/clip_guidance/base_clip.py:57-58) and, with the
 * transposed table, its backward -- repeatable and batch-invariant, where torch's bicubic backward uses atomics. */
int hedit_axis_mix(const float* in, float* out, const int32_t* idx, const float* val, int nnz, int64_t outer, int n_in, int n_out,
                   int inner, void* stream);
int hedit_local_blend(float* const* h_maps, int n_maps, int heads, const float* alpha_layers,
                      const int32_t* enabled, float* xt, int n_img, int C, int H, int W, float th,
                      void* stream);
/* the same with LocalBlend's substruct_words (ptp_classes.py:28-38,64-68): substruct_layers [n_img][2][77] marks the words
 * whose (un-pooled) map, thresholded with th_sub = th[1], is cut out of the blend mask */
int hedit_local_blend_sub(float* const* h_maps, int n_maps, int heads, const float* alpha_layers,
                          const float* substruct_layers, const int32_t* enabled, float* xt, int n_img, int C, int H,
                          int W, float th, float th_sub, void* stream);

/* ---- image autoencoder (SD-1.x AutoencoderKL): the steps either side of the editing loop ---------
 * replaces `model.vae.encode(image).latent_dist.mode()` (text-guided/main_p2p.py:159; also
 * p2p/ptp_classes.py:351-373) and `model.vae.decode(1 / 0.18215 * latents).sample`
 * (text-guided/main_p2p.py:263).  The 0.18215 scaling stays with the caller, as in the reference.
 * Parameters are addressed by their diffusers state_dict names (hedit_vae_param_name enumerates
 * them); fp32 device tensors in, packed to bf16 GEMM layouts on load.  Activations bf16 NHWC. */
typedef struct hedit_vae hedit_vae;
typedef struct hedit_vae_cfg {
  int in_channels;              /* 3 */
  int latent_channels;          /* 4 */
  int n_levels;                 /* number of entries used in block_out_channels (<= 4) */
  int block_out_channels[4];    /* 128,256,512,512; multiples of 64 */
  int layers_per_block;         /* 2 */
  int norm_num_groups;          /* 32 */
} hedit_vae_cfg;
int hedit_vae_create(const hedit_vae_cfg* cfg, hedit_vae** out);
void hedit_vae_destroy(hedit_vae* h);
int hedit_vae_num_params(const hedit_vae* h);
const char* hedit_vae_param_name(const hedit_vae* h, int i);
int hedit_vae_param_shape(const hedit_vae* h, int i, int* ndim, int* dims4);
int hedit_vae_load(hedit_vae* h, const char* name, const float* dev_w, size_t numel, void* stream);
int hedit_vae_missing(const hedit_vae* h);
/* workspace for a decode (encode = 0), an encode (encode = 1) or a decode_vjp (encode = 2) of B images
 * whose LATENT is h x w */
size_t hedit_vae_workspace_bytes(hedit_vae* h, int B, int latent_h, int latent_w, int encode);
/* z: fp32 [B][latent_channels][h][w] (already divided by the scaling factor)
 * -> image fp32 [B][in_channels][h*f][w*f], f = 2^(n_levels-1) */
int hedit_vae_decode(hedit_vae* h, const float* z, int B, int latent_h, int latent_w, float* image,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Vector-Jacobian product of the decoder: d_z = (d image / d z)^T d_image, what
 * `torch.autograd.grad(loss, latents)` pulls through `vae.decode` inside the reference's style-guidance
 * closure (text-guided-n-style/inversion/h_edit.py:146-185: x0 -> model.vae.decode -> image encoder ->
 * loss -> autograd.grad).  One call runs the forward (keeping block inputs and GroupNorm statistics
 * in the workspace) and the backward; no state survives the call.
 * z [B][latent_channels][h][w], d_image [B][in_channels][h*f][w*f] -> d_z like z; image (optional, may be
 * NULL) receives the decoded image as hedit_vae_decode would. */
int hedit_vae_decode_vjp(hedit_vae* h, const float* z, const float* d_image, int B, int latent_h, int latent_w,
                         float* d_z, float* image, void* workspace, size_t workspace_bytes, void* stream);
/* The same product in two calls, so that the caller's image encoder can run between them without a second
 * decoder forward: decode_keep = hedit_vae_decode that leaves its tape in the workspace (sized with
 * hedit_vae_workspace_bytes(.., 2)); decode_backward consumes it.  One outstanding forward per handle: a
 * further decode_keep drops the previous tape; the workspace must not be written in between. */
int hedit_vae_decode_keep(hedit_vae* h, const float* z, int B, int latent_h, int latent_w, float* image,
                          void* workspace, size_t workspace_bytes, void* stream);
int hedit_vae_decode_backward(hedit_vae* h, const float* d_image, float* d_z, void* workspace, void* stream);
/* image fp32 [B][in_channels][H][W] -> mean of the latent distribution, fp32 [B][latent_channels][H/f][W/f] */
int hedit_vae_encode(hedit_vae* h, const float* image, int B, int height, int width, float* mean,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- single kernels (parity tests) ------------------------------------------------------- */
/* C[M][N] = A . W^T (+bias)(+residual); mode 0 linear, 1 conv3x3 s1, 2 conv3x3 s2, 3 conv3x3 on
 * 2x nearest-upsampled input.  bf16 in/out.  splits: 0 = auto.  partial_ws may be NULL if the
 * call resolves to one split (query with hedit_k_gemm_ws_bytes). */
size_t hedit_k_gemm_ws_bytes(int M, int N, int K, int splits);
/* Batch-independent summation order.  The fp32 result of a contraction is DEFINED by a K-chunking that depends
 * on the layer's nominal shape only (per-image extent x 4 on the batch dimension): chunk sums are MFMA chains,
 * added in order.  hedit_k_gemm_canonical_chunk returns the chunk length in 64-deep K-tiles (0 = one chain);
 * hedit_k_gemm_plan_splits how many split-K slabs a launch of the ACTUAL shape uses to execute it (1 = chunks
 * folded in registers).  Both forms give the same bits, which hedit_k_gemm exposes for the tests:
 * splits > 0: that many slabs + reduce; splits < 0: the same chunking folded in one launch; 0: one plain chain. */
int hedit_k_gemm_canonical_chunk(int M_nominal, int N_nominal, int K);
int hedit_k_gemm_plan_splits(int M, int N, int K, int chunk_kt);
int hedit_k_gemm(const void* A, const void* W, const float* bias, const void* residual, void* C,
                 int M, int N, int K, int lda, int ldc, int ldr, int mode, int Hin, int Win,
                 int Cin, int Hout, int Wout, int splits, void* partial_ws, void* stream);
/* GroupNorm statistics taken by the producer (csrc/gnstat.h) -- what replaces the statistics pass of torch's GroupNorm in
 * the ResNet blocks of the pixel UNet (reference face-swapping/diffusion/diffusion.py:27-33 Normalize, :115-134 ResnetBlock).
 * hedit_k_conv_gn = hedit_k_gemm for a 3x3 convolution (mode 1..3; M % 128 == 0, N % 128 == 0 with the 128-column tile) that
 *   also writes gn_part[M / 128][N / 2][2] (fp32): (sum, sum of squares) of the stored bf16 output per unit of 128 rows and
 *   pair of adjacent channels, in a fixed summation tree -- the same bits whether the launch runs as one chain, folds its
 *   canonical chunks in registers (splits < 0) or goes through split-K slabs (splits > 0), and whatever shares the batch.
 * hedit_k_groupnorm_from_parts: y = GroupNorm(x) (+SiLU) for x [B][HW][C] = the channel concatenation of tensors with pair
 *   statistics part_a (ca channels) and part_b (cb channels, or null / 0); HW >= 1024, HW % 128 == 0, (C / G) even.
 *   ws: B * 512 bytes. */
int hedit_k_conv_gn(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N, int K, int ldc,
                    int ldr, int mode, int Hin, int Win, int Cin, int Hout, int Wout, int splits, void* partial_ws, float* gn_part,
                    void* stream);
int hedit_k_groupnorm_from_parts(const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                 float eps, int silu, const float* part_a, int ca, const float* part_b, int cb, void* ws,
                                 void* stream);
/* FF1 of the transformer feed-forward with the GEGLU fused into the GEMM epilogue (diffusers
 * GEGLU.forward: hidden, gate = proj(x).chunk(2, -1); out = hidden * gelu(gate)).
 * hedit_k_pack_geglu interleaves the fp32 [2*inner][K] projection weight (and [2*inner] bias) of the
 * checkpoint layout into the row order the kernel wants; hedit_k_gemm_geglu then computes
 * C[M][inner] (bf16) from A[M][K] (bf16).  inner % 16 == 0, K % 8 == 0. */
int hedit_k_pack_geglu(const float* w, const float* bias, void* w_packed_bf16, float* bias_packed,
                       int inner, int K, void* stream);
int hedit_k_gemm_geglu(const void* A, const void* w_packed, const float* bias_packed, void* C, int M,
                       int inner, int K, int lda, int ldc, void* stream);
/* The token-local tail of a BasicTransformerBlock in one launch (csrc/ffn.hip), for C == hedit_k_ffn_channels() (320).
 * hedit_k_ffn_fused: out[M][C] = x + ff.net.2( GEGLU( ff.net.0.proj( LayerNorm(x) ) ) )  -- replaces diffusers'
 *   `hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states` (oracle/sd_unet.py transformer block).
 * hedit_k_ffn_chain: the same with the linear layers either side of it,
 *   t2 = attn2.to_out.0(a) + t1;  t3 = t2 + ff(norm3(t2));  out = proj_out(t3) + x
 *   (oracle/sd_unet.py: the cross-attention output projection + residual in front, Transformer2DModel.proj_out +
 *   residual behind), x / t1 / a / out bf16 with row strides that are multiples of 8.
 * hedit_k_ffn_pack turns the checkpoint tensors w1 = ff.net.0.proj.weight [8C][C], b1 = ff.net.0.proj.bias [8C],
 * w2 = ff.net.2.weight [C][4C] (and w_pre = attn2.to_out.0.weight [C][C], w_post = proj_out.weight [C][C], both or
 * neither; fp32, device) into the weight stream (hedit_k_ffn_stream_bytes(with_outer_layers)) and the packed FF1 bias
 * (hedit_k_ffn_bias_bytes). */
int hedit_k_ffn_channels(void);
size_t hedit_k_ffn_stream_bytes(int with_outer_layers);
size_t hedit_k_ffn_bias_bytes(void);
int hedit_k_ffn_pack(const float* w1, const float* b1, const float* w2, const float* w_pre, const float* w_post, void* stream_out,
                     float* bias1_out, void* stream);
int hedit_k_ffn_fused(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* w_stream,
                      const float* bias1_packed, const float* bias2, void* out, int64_t ldo, int M, int C, void* stream);
int hedit_k_ffn_chain(const void* a, int64_t lda, const void* t1, int64_t ldt1, const void* x, int64_t ldx, const float* bias_pre,
                      const float* gamma, const float* beta, float eps, const void* w_stream, const float* bias1_packed,
                      const float* bias2, const float* bias_post, void* out, int64_t ldo, int M, int C, void* stream);
/* The projections around an attention of the same level in one launch (csrc/linchain.hip), two forms:
 *   n_out = 1 (gn_ss == NULL):  mid = attn1.to_out.0(a) + r1 (written to out_mid);  out[M][C] = attn2.to_q( norm2(mid) )
 *   n_out = 3 (gn_ss set):      mid = proj_in( a * scale + shift ) (GroupNorm applied on the fly; written to out_mid);
 *                               q, k = attn1.to_q / to_k ( norm1(mid) ) -> out_q, out_k;  out[C][ldo] = attn1.to_v(norm1(mid))^T
 *   (oracle/sd_unet.py: Transformer2DModel.norm / proj_in and BasicTransformerBlock attn1 / attn2 projections).
 * hedit_k_lin_chain_pack: checkpoint tensors [C][C] (fp32, device) -> weight stream; scale0 multiplies w0 (the softmax
 * scale folded into the query projection); w1, w2: both (n_out = 3) or neither.
 * hedit_k_groupnorm_affine: ss_out [B][C][2] = (scale, shift) per image and channel of GroupNorm(x). */
size_t hedit_k_lin_chain_stream_bytes(int n_out);
int hedit_k_lin_chain_pack(const float* w_pre, const float* w0, const float* w1, const float* w2, float scale0, void* stream_out,
                           void* stream);
int hedit_k_lin_chain(const void* a, int64_t lda, const void* r1, int64_t ldr1, const float* gn_ss, int rows_per_image,
                      const float* bias_pre, const float* gamma, const float* beta, float eps, const void* w_stream, void* out_mid,
                      int64_t ldmid, void* out_q, int64_t ldq, void* out_k, int64_t ldk, void* out, int64_t ldo, int M, int C,
                      void* stream);
/* Test entry (tests/test_gpu_chain_hazard.py): hedit_k_lin_chain with the kernel's wait schedule chosen -- sched 0 = the
 * product's (counted vmcnt windows around the bursts between the layers), 1 = every wait drained to vmcnt(0) lgkmcnt(0)
 * in front of its barrier: same arithmetic, no reliance on queue order or timing, i.e. the bits the product schedule must
 * reproduce under any memory load.  Never called by the executor. */
int hedit_k_lin_chain_sched(const void* a, int64_t lda, const void* r1, int64_t ldr1, const float* gn_ss, int rows_per_image,
                            const float* bias_pre, const float* gamma, const float* beta, float eps, const void* w_stream, void* out_mid,
                            int64_t ldmid, void* out_q, int64_t ldq, void* out_k, int64_t ldk, void* out, int64_t ldo, int M, int C,
                            int sched, void* stream);
/* The 16-bit storage format of this build: 0 = bfloat16 (libhedit_hip.so: the default, BASELINE configs[1], every benchmark
 * figure), 1 = IEEE half (libhedit_hip_f16.so, `build.py --f16`: the same kernels with -DHEDIT_STORE_F16).  Wherever this header
 * says "bf16" for a buffer of a kernel-level entry or for the P2P mix tables, read "the storage format of the build"; the
 * executors' own boundaries (images, latents, eps, checkpoints) are fp32 in either build. */
int hedit_storage_is_f16(void);
/* Test switch (tests/test_gpu_ring_hazard.py), process-wide, 0 by default and in every product path.
 * bit 0: the kernels whose operand rings are retired by COUNTED s_waitcnt vmcnt(N) -- igemm_kernel's three-stage ring and
 *        row-sharing 3x3 loop, ffn_chain_kernel, self_attn_kernel -- launch their drained twin: every ring wait is vmcnt(0)
 *        (and lgkmcnt(0)) in front of its barrier, same arithmetic in the same order.  The product schedule must reproduce
 *        those bits under any memory load.  (lin_chain_kernel has its own entry above.)
 * bit 1: self-attention takes the exact online-softmax pass only (no pinned-shift pass) -- the reference bits of the
 *        denominator-check / redo logic.
 * bit 2: the pixel UNet (hedit_ddpm_forward) takes the OTHER of its two GroupNorm statistics paths (statistics pass over the
 *        tensor / pair statistics from the producing convolution's epilogue, csrc/gnstat.h): tests/test_gpu_gn_stats.py
 *        compares the two on one library.
 * bit 3: gemm launches that would run a persistent kernel (csrc/pgemm.hip, csrc/pconv.hip) run the one-shot igemm_kernel of the
 *        same tile instead -- the same bits by construction; tools/pgemm_ab.py / tools/pconv_ab.py compare them (torch.equal)
 *        and time both on one box. */
int hedit_test_set_flags(int flags);
size_t hedit_k_groupnorm_ws_bytes(int B, int HW, int C);
int hedit_k_groupnorm_affine(const void* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, void* ws,
                             float* ss_out, void* stream);
int hedit_k_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int B, int HW,
                      int C, int G, float eps, int silu, void* ws, void* stream);
int hedit_k_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows,
                      int C, float eps, void* stream);
int hedit_k_geglu(const void* x, void* y, int64_t rows, int inner, void* stream);
int hedit_k_self_attn(const void* q, int ldq, const void* k, int ldk, const void* vt, int64_t ldvt,
                      void* out, int ldo, int B, int N, int heads, int d, const int32_t* qk_src, const int32_t* kv_src,
                      void* stream);
int hedit_k_cross_attn(const void* q, int ldq, const void* k, int ldk, const void* vt, int64_t ldvt,
                       void* out, int ldo, int B, int N, int heads, int d,
                       const hedit_p2p_plan* plan, float* store, void* stream);
/* the two kernels of the hook path: probs = softmax2(q k^T) as fp32 [B*heads][N][M] (q pre-scaled by scale * log2 e;
 * k: [B*kstride][ldk], the first M rows of every batch item count), then out = probs . v (vt: [heads*d][B*kstride]) */
int hedit_k_attn_probs(const void* q, int ldq, const void* k, int ldk, float* probs, int B, int N, int M, int kstride,
                       int heads, int d, void* stream);
int hedit_k_attn_apply(const float* probs, const void* vt, int64_t ldvt, void* out, int ldo, int B, int N, int M,
                       int kstride, int heads, int d, void* stream);
int hedit_k_pack_conv3x3(const float* w_oihw, void* out, int O, int I, void* stream);
int hedit_k_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pixel-space DDPM UNet of the face-swapping task: `Model.forward(x, t)` of
 * face-swapping/diffusion/diffusion.py:192-341, evaluated by face-swapping/inversion/h_edit_R.py:71,96,118
 * and inversion/sde_inversion.py:121.  Parameters by the state_dict names of that class
 * (temb.dense.0.weight, down.{i}.block.{j}.conv1.weight, down.{i}.attn.{j}.q.weight, mid.attn_1.proj_out.bias,
 * up.{i}.upsample.conv.weight, norm_out.weight, ...), fp32 device tensors in torch layouts.
 * attn_level[i] != 0 <=> the level's resolution is in the config's attn_resolutions. */
typedef struct hedit_ddpm hedit_ddpm;
typedef struct {
  int in_channels, out_ch, ch, n_levels;
  int ch_mult[8];
  int attn_level[8];
  int num_res_blocks, image_size;
} hedit_ddpm_cfg;
int hedit_ddpm_create(const hedit_ddpm_cfg* cfg, hedit_ddpm** out);
void hedit_ddpm_destroy(hedit_ddpm* h);
int hedit_ddpm_num_params(const hedit_ddpm* h);
const char* hedit_ddpm_param_name(const hedit_ddpm* h, int i);
int hedit_ddpm_param_shape(const hedit_ddpm* h, int i, int* ndim, int* dims4);
int hedit_ddpm_load(hedit_ddpm* h, const char* name, const float* dev_w, size_t numel, void* stream);
int hedit_ddpm_missing(const hedit_ddpm* h);
size_t hedit_ddpm_workspace_bytes(hedit_ddpm* h, int B);
/* x fp32 [B][in_channels][S][S], one timestep t for the batch (the reference passes ones(n) * t)
 * -> eps fp32 [B][out_ch][S][S] */
int hedit_ddpm_forward(hedit_ddpm* h, const float* x, float t, int B, float* eps, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Identity reward of the face-swapping task: `IDLoss.get_cosine_loss(image)` of face-swapping/arcface/
 * arcface_model.py:40-67 (crop [35:223, 32:220] -> adaptive average pool 112 -> IR-SE50, model_irse.py:9-48 /
 * helpers.py:28-119 -> l2-normalised feature -> 1 - cos against the reference face's feature) fused with the gradient
 * w.r.t. the image that inversion/h_edit_R.py:103-106 takes from it with torch.autograd.grad.  fp32-quality
 * arithmetic (three-term split-bf16 products with fp32 accumulation, fp32 activations).  Parameters by the
 * reference's state_dict names (`input_layer.0.weight`, `body.3.res_layer.4.running_var`, `output_layer.3.weight`,
 * ...; the integer `num_batches_tracked` buffers are not parameters), fp32 device tensors in torch layouts;
 * hedit_irse50_finalize folds the BatchNorms and packs the GEMM operands once after loading. */
typedef struct hedit_irse hedit_irse;
int hedit_irse50_create(hedit_irse** out);
void hedit_irse50_destroy(hedit_irse* h);
int hedit_irse50_num_params(const hedit_irse* h);
const char* hedit_irse50_param_name(const hedit_irse* h, int i);
int hedit_irse50_param_shape(const hedit_irse* h, int i, int* ndim, int* dims4);
int hedit_irse50_load(hedit_irse* h, const char* name, const float* dev_w, size_t numel, void* stream);
int hedit_irse50_missing(const hedit_irse* h);
int hedit_irse50_finalize(hedit_irse* h, void* stream);
size_t hedit_irse50_workspace_bytes(hedit_irse* h, int B);
/* image fp32 [B][3][256][256] in [-1, 1] -> feat fp32 [B][512] = F.normalize(IDLoss.extract_feats(image)) */
int hedit_irse50_features(hedit_irse* h, const float* image, int B, float* feat, void* workspace,
                          size_t workspace_bytes, void* stream);
/* loss[b] = 1 - cos(feature(image_b), ref_feat); d_image [B][3][256][256] = d(scale * sum_b loss[b]) / d image
 * (scale = 1 / B: the reference's batch mean).  ref_feat: l2-normalised [512] (ref_per_image = 0) or [B][512]. */
int hedit_irse50_cos_fwd_bwd(hedit_irse* h, const float* image, const float* ref_feat, int ref_per_image, int B,
                             float scale, float* loss, float* d_image, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * Perceptual reward of the face-swapping task: `LPIPS_Loss.get_lpips_loss(x)` of face-swapping/arcface/
 * arcface_model.py:69-94 = lpips.LPIPS(net='vgg')(x, src).mean() (lpips==0.1.4: ScalingLayer, VGG16 taps relu1_2 ...
 * relu5_3, channel-unit normalisation, squared difference, 1x1 lin weights, spatial mean, sum over taps) fused with
 * the gradient w.r.t. x that inversion/h_edit_R.py:124-132 takes with torch.autograd.grad.  Parameters by the lpips
 * state_dict names (`net.slice1.0.weight` ... `net.slice5.28.bias`, `lin0.model.1.weight` ... `lin4.model.1.weight`),
 * fp32 device tensors in torch layouts.  Image height / width: multiples of 16. */
typedef struct hedit_lpips hedit_lpips;
int hedit_lpips_create(hedit_lpips** out);
void hedit_lpips_destroy(hedit_lpips* h);
int hedit_lpips_num_params(const hedit_lpips* h);
const char* hedit_lpips_param_name(const hedit_lpips* h, int i);
int hedit_lpips_param_shape(const hedit_lpips* h, int i, int* ndim, int* dims4);
int hedit_lpips_load(hedit_lpips* h, const char* name, const float* dev_w, size_t numel, void* stream);
int hedit_lpips_missing(const hedit_lpips* h);
int hedit_lpips_finalize(hedit_lpips* h, void* stream);
size_t hedit_lpips_feature_floats(int height, int width);
size_t hedit_lpips_workspace_bytes(hedit_lpips* h, int B, int height, int width);
/* src fp32 [B][3][H][W] in [-1, 1] -> feats fp32 [B][hedit_lpips_feature_floats(H, W)] (normalised tap features) */
int hedit_lpips_source(hedit_lpips* h, const float* src, int B, int height, int width, float* feats,
                       void* workspace, size_t workspace_bytes, void* stream);
/* loss[b] = LPIPS(x_b, source); d_x = d(scale * sum_b loss[b]) / d x (scale = 1 / B: the reference's batch mean).
 * src_feats: one source shared by the batch (src_per_image = 0) or one per image. */
int hedit_lpips_fwd_bwd(hedit_lpips* h, const float* x, const float* src_feats, int src_per_image, int B, int height,
                        int width, float scale, float* loss, float* d_x, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Style encoder of the text + style task: `CLIPEncoder.get_gram_matrix_residual` of text-guided-n-style/clip_guidance/
 * base_clip.py:55-66 on the prefix of the CLIP ViT it exercises (clip/model.py:195-237,339-365: patch embedding, class +
 * positional embedding, ln_pre, the first `layers` ResidualAttentionBlocks): Gram matrix F^T F of the last block's patch
 * tokens, its residual against the style reference's Gram matrix and the Frobenius norm, fused with the gradient w.r.t.
 * the CLIP-normalised, resized image (what inversion/h_edit.py:170-179 pulls back into the decoder).  Parameters by the
 * OpenAI CLIP state_dict names (`visual.conv1.weight`, `visual.transformer.resblocks.{i}.attn.in_proj_weight`, ...).
 * width / heads = 64; at most 200 tokens. */
typedef struct hedit_vit hedit_vit;
typedef struct { int width, layers, heads, patch_size, input_resolution; } hedit_vit_cfg;
int hedit_vit_create(const hedit_vit_cfg* cfg, hedit_vit** out);
void hedit_vit_destroy(hedit_vit* h);
int hedit_vit_num_params(const hedit_vit* h);
const char* hedit_vit_param_name(const hedit_vit* h, int i);
int hedit_vit_param_shape(const hedit_vit* h, int i, int* ndim, int* dims4);
int hedit_vit_load(hedit_vit* h, const char* name, const float* dev_w, size_t numel, void* stream);
int hedit_vit_missing(const hedit_vit* h);
int hedit_vit_finalize(hedit_vit* h, void* stream);
size_t hedit_vit_workspace_bytes(hedit_vit* h, int B);
/* image fp32 [B][3][R][R] (CLIP-normalised, R = input_resolution) -> gram fp32 [B][width][width] */
int hedit_vit_gram(hedit_vit* h, const float* image, int B, float* gram, void* workspace, size_t workspace_bytes,
                   void* stream);
/* loss[b] = |Gram(image_b) - gram_ref|_F; d_image = d(scale * sum_b loss[b]) / d image.  gram_ref [width][width]
 * shared by the batch (ref_per_image = 0) or [B][width][width]. */
int hedit_vit_gram_fwd_bwd(hedit_vit* h, const float* image, const float* gram_ref, int ref_per_image, int B, float scale,
                           float* loss, float* d_image, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
