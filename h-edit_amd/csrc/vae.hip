// The SD-1.x image autoencoder (diffusers AutoencoderKL) as a native executor on the kernels of
// gemm.hip / norm.hip: the steps either side of the editing loop -- `vae.encode(img).latent_dist
// .mode()` before the inversion (reference text-guided/main_p2p.py:159) and `vae.decode(latents)`
// after it (main_p2p.py:263).  Same conventions as unet.hip:
// parameters are loaded by their diffusers state_dict names, activations are NHWC bf16 in a
// caller-provided workspace, one C call per pass.
//
// Architecture (diffusers 0.18 `AutoencoderKL`, `Encoder` / `Decoder`, `UNetMidBlock2D` with one
// single-head `Attention`, `DownEncoderBlock2D` / `UpDecoderBlock2D`, resnet eps 1e-6, no time
// embedding).  The encoder's Downsample2D(padding=0) pads (0,1,0,1) -> GemmParams.asym.
//
// Mid-block attention (one head of C = 512 channels over h*w tokens) does not fit the flash
// kernel's head sizes; it runs once per image per pass, so it is done with the GEMM kernel:
// fp32 scores S = Q K^T, a row-softmax pass, O = P V with V^T from a swapped-operand GEMM.  Two
// exact simplifications: the key bias adds a per-row constant to S (softmax-invariant) and is
// dropped; the value bias passes through P (rows sum to 1) and is folded into the output bias.
#include "blocks.h"

namespace {

struct VMid {
  VRes r0, r1;
  VAttn at;
};

struct VStage {
  std::vector<VRes> res;
  bf16_t* samp_w = nullptr;   // down: stride-2 conv; up: conv after the 2x nearest upsample
  bf16_t* samp_t = nullptr;   // up: its input-gradient weight
  float* samp_b = nullptr;
  int ch = 0;
};

}  // namespace

struct hedit_vae : ParamStore {
  hedit_vae_cfg cfg;
  // encoder
  float *e_in_w = nullptr, *e_in_b = nullptr, *e_gn_g = nullptr, *e_gn_b = nullptr, *e_out_b = nullptr;
  bf16_t* e_out_w = nullptr;
  std::vector<VStage> down;
  VMid e_mid;
  float *quant_w = nullptr, *quant_b = nullptr;
  // decoder
  float *pq_w = nullptr, *pq_b = nullptr, *d_in_w = nullptr, *d_in_b = nullptr, *d_gn_g = nullptr, *d_gn_b = nullptr,
        *d_out_b = nullptr;
  bf16_t* d_out_w = nullptr;
  std::vector<VStage> up;
  VMid d_mid;
  // input-gradient twins of the decoder's head / tail weights, and a zero bias vector
  float *pq_t = nullptr, *d_out_t = nullptr, *zero_bias = nullptr;
  bf16_t* d_in_t = nullptr;
  // the outstanding tape of hedit_vae_decode_keep (a DecodeTape), consumed by hedit_vae_decode_backward
  void* tape = nullptr;
  void (*tape_free)(void*) = nullptr;
};

namespace {

VMid make_mid(hedit_vae* h, const std::string& pre, int C, bool grad = false) {
  VMid m;
  m.r0 = make_res(h, pre + ".resnets.0", C, C, grad);
  m.at = make_attn(h, pre + ".attentions.0", C, grad);
  m.r1 = make_res(h, pre + ".resnets.1", C, C, grad);
  return m;
}


struct MidRec {
  ResRec r0, r1;
  AttnRec at;
};

int mid(VF& f, const VMid& m, bf16_t** x, int H, int W, MidRec* rec = nullptr) {
  bf16_t *a, *b, *c;
  TRY(resblock(f, m.r0, *x, H, W, &a, rec ? &rec->r0 : nullptr));
  if (!rec) f.ar.free(*x);
  TRY(attention(f, m.at, a, H, W, &b, rec ? &rec->at : nullptr));
  if (!rec) f.ar.free(a);
  TRY(resblock(f, m.r1, b, H, W, &c, rec ? &rec->r1 : nullptr));
  if (!rec) f.ar.free(b);
  *x = c;
  return HEDIT_OK;
}

int mid_bwd(VF& f, const MidRec& rec, bf16_t** d) {
  bf16_t *a, *b, *c;
  TRY(resblock_bwd(f, rec.r1, *d, &a));
  f.ar.free(*d);
  TRY(attention_bwd(f, rec.at, a, &b));
  f.ar.free(a);
  TRY(resblock_bwd(f, rec.r0, b, &c));
  f.ar.free(b);
  *d = c;
  return HEDIT_OK;
}

// Everything a decoder backward pass needs from its forward: the arena (with the kept buffers still
// allocated in it) and the per-block records.
struct DecodeTape {
  VF f;
  MidRec mrec{};
  std::vector<ResRec> rrec;
  bf16_t* x = nullptr;       // input of conv_norm_out
  float* st_out = nullptr;   // its GroupNorm statistics
  int B = 0, lh = 0, lw = 0, H = 0, W = 0;
  void* ws = nullptr;
};

// forward; grad = keep block inputs, conv1 outputs and GroupNorm statistics for decode_backward
int decode_forward(hedit_vae* h, DecodeTape& T, const float* z, float* image, bool grad) {
  VF& f = T.f;
  const int B = f.B;
  hipStream_t st = f.st;
  const hedit_vae_cfg& c = h->cfg;
  const int L = c.n_levels, LC = c.latent_channels;
  int H = T.lh, W = T.lw;
  int ch = c.block_out_channels[L - 1];
  float* z2;
  TRY(aalloc(f, &z2, (size_t)B * LC * H * W));
  RUN(f, mix1x1_nchw_launch(z, h->pq_w, h->pq_b, z2, B, LC, LC, (long)H * W, 1.0f, st));
  bf16_t* x;
  TRY(aalloc(f, &x, (size_t)B * H * W * ch));
  RUN(f, conv_in_launch(z2, h->d_in_w, h->d_in_b, x, B, LC, H, W, ch, st));
  f.ar.free(z2);
  TRY(mid(f, h->d_mid, &x, H, W, grad ? &T.mrec : nullptr));
  for (int i = 0; i < L; ++i) {
    const VStage& s = h->up[i];
    for (const VRes& r : s.res) {
      bf16_t* y;
      if (grad) {
        T.rrec.emplace_back();
        TRY(resblock(f, r, x, H, W, &y, &T.rrec.back()));
      } else {
        TRY(resblock(f, r, x, H, W, &y));
        f.ar.free(x);
      }
      x = y;
    }
    if (s.samp_w) {
      bf16_t* y;
      TRY(aalloc(f, &y, (size_t)B * H * W * 4 * s.ch));
      TRY(conv3x3(f, x, H, W, s.ch, s.samp_w, s.ch, s.samp_b, nullptr, y, 3));
      f.ar.free(x);   // the upsampling conv is linear in x: nothing to keep
      x = y;
      H *= 2; W *= 2;
    }
    ch = s.ch;
  }
  bf16_t* xn;
  TRY(aalloc(f, &xn, (size_t)B * H * W * ch));
  TRY(groupnorm(f, x, xn, h->d_gn_g, h->d_gn_b, H * W, ch, 1, grad ? &T.st_out : nullptr));
  if (!grad) f.ar.free(x);
  if (image) TRY(conv_out(f, xn, H, W, ch, h->d_out_w, h->d_out_b, c.in_channels, image));
  f.ar.free(xn);
  T.x = x;
  T.H = H; T.W = W;
  return HEDIT_OK;
}

// d_z = (d image / d z)^T d_image from a tape made by decode_forward(grad = true)
int decode_backward(hedit_vae* h, DecodeTape& T, const float* d_image, float* d_z) {
  VF& f = T.f;
  const int B = f.B;
  hipStream_t st = f.st;
  const hedit_vae_cfg& c = h->cfg;
  const int L = c.n_levels, LC = c.latent_channels;
  int H = T.H, W = T.W;
  int ch = c.block_out_channels[0];
  bf16_t *d, *t;
  TRY(aalloc(f, &t, (size_t)B * H * W * ch));
  RUN(f, conv_in_launch(d_image, h->d_out_t, h->zero_bias, t, B, c.in_channels, H, W, ch, st));
  TRY(aalloc(f, &d, (size_t)B * H * W * ch));
  TRY(groupnorm_bwd(f, T.x, t, nullptr, d, h->d_gn_g, h->d_gn_b, T.st_out, H * W, ch, 1));
  f.ar.free(t);
  f.ar.free(T.x);
  size_t ri = T.rrec.size();
  for (int i = L - 1; i >= 0; --i) {
    const VStage& s = h->up[i];
    if (s.samp_w) {
      // d(2x upsample) = dgrad conv at the high resolution, then 2x2 block sums
      bf16_t *du, *dx;
      TRY(aalloc(f, &du, (size_t)B * H * W * s.ch));
      TRY(conv3x3(f, d, H, W, s.ch, s.samp_t, s.ch, nullptr, nullptr, du, 1));
      f.ar.free(d);
      H /= 2; W /= 2;
      TRY(aalloc(f, &dx, (size_t)B * H * W * s.ch));
      RUN(f, sum2x2_launch(du, dx, B, H, W, s.ch, st));
      f.ar.free(du);
      d = dx;
    }
    for (size_t j = 0; j < s.res.size(); ++j) {
      const ResRec& rec = T.rrec[--ri];
      bf16_t* dx;
      TRY(resblock_bwd(f, rec, d, &dx));
      f.ar.free(d);
      f.ar.free(rec.h1); f.ar.free(rec.st1); f.ar.free(rec.st2);
      f.ar.free(const_cast<bf16_t*>(rec.x));   // this block's input: last use
      d = dx;
    }
  }
  TRY(mid_bwd(f, T.mrec, &d));
  ch = c.block_out_channels[L - 1];
  float* dz2;
  TRY(aalloc(f, &dz2, (size_t)B * LC * H * W));
  if (LC == 4) {
    TRY(conv_out(f, d, H, W, ch, h->d_in_t, nullptr, LC, dz2));
  } else {
    RUN(f, conv_out_launch(d, h->d_in_t, h->zero_bias, dz2, B, H, W, ch, LC, st));
  }
  f.ar.free(d);
  RUN(f, mix1x1_nchw_launch(dz2, h->pq_t, nullptr, d_z, B, LC, LC, (long)H * W, 1.0f, st));
  f.ar.free(dz2);
  return HEDIT_OK;
}

void tape_init(DecodeTape& T, hedit_vae* h, int B, int lh, int lw, void* ws, size_t ws_bytes, hipStream_t st, bool dry) {
  T.f.groups = h->cfg.norm_num_groups; T.f.B = B; T.f.st = st;
  T.f.ar.dry = dry;
  T.f.ar.base = reinterpret_cast<char*>(ws);
  T.f.ar.cap = ws_bytes;
  T.B = B; T.lh = lh; T.lw = lw; T.ws = ws;
}

// one call: forward, and with d_z != null also the backward
int decode_impl(hedit_vae* h, const float* z, int B, int lh, int lw, float* image, void* ws, size_t ws_bytes,
                hipStream_t st, bool dry, size_t* peak, const float* d_image = nullptr, float* d_z = nullptr) {
  DecodeTape T;
  tape_init(T, h, B, lh, lw, ws, ws_bytes, st, dry);
  TRY(decode_forward(h, T, z, image, d_z != nullptr));
  if (d_z) TRY(decode_backward(h, T, d_image, d_z));
  if (peak) *peak = T.f.ar.peak;
  return HEDIT_OK;
}

int encode_impl(hedit_vae* h, const float* image, int B, int IH, int IW, float* mean, void* ws, size_t ws_bytes,
                hipStream_t st, bool dry, size_t* peak) {
  VF f{h->cfg.norm_num_groups, B, st, Arena{}};
  f.ar.dry = dry;
  f.ar.base = reinterpret_cast<char*>(ws);
  f.ar.cap = ws_bytes;
  const hedit_vae_cfg& c = h->cfg;
  const int L = c.n_levels, LC = c.latent_channels;
  int H = IH, W = IW;
  int ch = c.block_out_channels[0];
  bf16_t* x;
  TRY(aalloc(f, &x, (size_t)B * H * W * ch));
  RUN(f, conv_in_launch(image, h->e_in_w, h->e_in_b, x, B, c.in_channels, H, W, ch, st));
  for (int i = 0; i < L; ++i) {
    const VStage& s = h->down[i];
    for (const VRes& r : s.res) {
      bf16_t* y;
      TRY(resblock(f, r, x, H, W, &y));
      f.ar.free(x);
      x = y;
    }
    ch = s.ch;
    if (s.samp_w) {
      bf16_t* y;
      TRY(aalloc(f, &y, (size_t)B * (H / 2) * (W / 2) * ch));
      TRY(conv3x3(f, x, H, W, ch, s.samp_w, ch, s.samp_b, nullptr, y, 2));
      f.ar.free(x);
      x = y;
      H /= 2; W /= 2;
    }
  }
  TRY(mid(f, h->e_mid, &x, H, W));
  bf16_t *xn, *mom;
  TRY(aalloc(f, &xn, (size_t)B * H * W * ch));
  TRY(groupnorm(f, x, xn, h->e_gn_g, h->e_gn_b, H * W, ch, 1));
  f.ar.free(x);
  TRY(aalloc(f, &mom, (size_t)B * H * W * 2 * LC));
  TRY(conv3x3(f, xn, H, W, ch, h->e_out_w, 2 * LC, h->e_out_b, nullptr, mom, 1));
  f.ar.free(xn);
  // latent_dist.mode() == mean == first LC channels of quant_conv(moments)
  RUN(f, quant_mean_launch(mom, h->quant_w, h->quant_b, mean, B, (long)H * W, 2 * LC, LC, st));
  f.ar.free(mom);
  if (peak) *peak = f.ar.peak;
  return HEDIT_OK;
}

}  // namespace

// ==================================================================================== C ABI
extern "C" {

int hedit_vae_create(const hedit_vae_cfg* cfg, hedit_vae** out) try {
  ARG_CHECK(cfg && out, "null");
  ARG_CHECK(cfg->n_levels >= 1 && cfg->n_levels <= 4, "n_levels in 1..4");
  ARG_CHECK(cfg->in_channels >= 1 && cfg->in_channels <= 4, "in_channels in 1..4");
  ARG_CHECK(cfg->latent_channels >= 1 && cfg->latent_channels <= 8, "latent_channels in 1..8");
  ARG_CHECK(cfg->layers_per_block >= 1 && cfg->layers_per_block <= 4, "layers_per_block in 1..4");
  for (int i = 0; i < cfg->n_levels; ++i) {
    const int c = cfg->block_out_channels[i];
    ARG_CHECK(c % 64 == 0 && c % cfg->norm_num_groups == 0, "block_out_channels: multiples of 64 and of norm_num_groups");
  }
  TRY(gemm_prepare());
  hedit_vae* h = new hedit_vae();
  h->cfg = *cfg;
  const int L = cfg->n_levels, LC = cfg->latent_channels;
  const int* bc = cfg->block_out_channels;
  // ---- encoder
  h->e_in_w = f32conv(h, "encoder.conv_in.weight", bc[0], cfg->in_channels, 3);
  h->e_in_b = vec(h, "encoder.conv_in.bias", bc[0]);
  int ch = bc[0];
  for (int i = 0; i < L; ++i) {
    VStage s;
    const std::string pre = "encoder.down_blocks." + std::to_string(i);
    for (int j = 0; j < cfg->layers_per_block; ++j) {
      s.res.push_back(make_res(h, pre + ".resnets." + std::to_string(j), j == 0 ? ch : bc[i], bc[i]));
    }
    ch = bc[i];
    s.ch = ch;
    if (i < L - 1) {
      s.samp_w = conv3(h, pre + ".downsamplers.0.conv.weight", ch, ch);
      s.samp_b = vec(h, pre + ".downsamplers.0.conv.bias", ch);
    }
    h->down.push_back(s);
  }
  h->e_mid = make_mid(h, "encoder.mid_block", ch);
  h->e_gn_g = vec(h, "encoder.conv_norm_out.weight", ch);
  h->e_gn_b = vec(h, "encoder.conv_norm_out.bias", ch);
  h->e_out_w = conv3(h, "encoder.conv_out.weight", 2 * LC, ch);
  h->e_out_b = vec(h, "encoder.conv_out.bias", 2 * LC);
  h->quant_w = f32conv(h, "quant_conv.weight", 2 * LC, 2 * LC, 1);
  h->quant_b = vec(h, "quant_conv.bias", 2 * LC);
  // ---- decoder
  h->pq_w = f32conv(h, "post_quant_conv.weight", LC, LC, 1);
  h->pq_t = twin<float>(h, 3, (size_t)LC * LC);
  h->pq_b = vec(h, "post_quant_conv.bias", LC);
  ch = bc[L - 1];
  h->d_in_w = f32conv(h, "decoder.conv_in.weight", ch, LC, 3);
  h->d_in_t = twin<bf16_t>(h, 2, (size_t)ch * LC * 9);
  h->d_in_b = vec(h, "decoder.conv_in.bias", ch);
  h->d_mid = make_mid(h, "decoder.mid_block", ch, true);
  for (int i = 0; i < L; ++i) {
    VStage s;
    const int co = bc[L - 1 - i];
    const std::string pre = "decoder.up_blocks." + std::to_string(i);
    for (int j = 0; j < cfg->layers_per_block + 1; ++j) {
      s.res.push_back(make_res(h, pre + ".resnets." + std::to_string(j), j == 0 ? ch : co, co, true));
    }
    ch = co;
    s.ch = ch;
    if (i < L - 1) {
      s.samp_w = conv3(h, pre + ".upsamplers.0.conv.weight", ch, ch);
      s.samp_t = twin<bf16_t>(h, 2, (size_t)ch * ch * 9);
      s.samp_b = vec(h, pre + ".upsamplers.0.conv.bias", ch);
    }
    h->up.push_back(s);
  }
  h->d_gn_g = vec(h, "decoder.conv_norm_out.weight", ch);
  h->d_gn_b = vec(h, "decoder.conv_norm_out.bias", ch);
  h->d_out_w = conv3(h, "decoder.conv_out.weight", cfg->in_channels, ch);
  h->d_out_t = twin<float>(h, 3, (size_t)cfg->in_channels * ch * 9);
  h->d_out_b = vec(h, "decoder.conv_out.bias", cfg->in_channels);
  {
    int zc = 8;
    for (int i = 0; i < L; ++i) zc = bc[i] > zc ? bc[i] : zc;
    h->zero_bias = dalloc<float>(h, zc);
    if (h->zero_bias && hipMemset(h->zero_bias, 0, zc * sizeof(float)) != hipSuccess) h->alloc_failed = true;
  }
  if (h->alloc_failed) {
    hedit_set_error("hipMalloc failed while creating the VAE");
    hedit_vae_destroy(h);
    return HEDIT_ERR_HIP;
  }
  *out = h;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

static void drop_tape(hedit_vae* h) {
  if (h->tape && h->tape_free) h->tape_free(h->tape);
  h->tape = nullptr;
}

void hedit_vae_destroy(hedit_vae* h) try {
  if (!h) return;
  drop_tape(h);
  store_free(h);
  delete h;
} catch (...) { (void)hedit_abi_catch(); }

int hedit_vae_num_params(const hedit_vae* h) { return h ? (int)h->slots.size() : 0; }
const char* hedit_vae_param_name(const hedit_vae* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
} catch (...) { (void)hedit_abi_catch(); return nullptr; }
int hedit_vae_param_shape(const hedit_vae* h, int i, int* ndim, int* dims4) try {
  ARG_CHECK(h && ndim && dims4 && i >= 0 && i < (int)h->slots.size(), "param index");
  *ndim = h->slots[i].ndim;
  for (int k = 0; k < 4; ++k) dims4[k] = h->slots[i].dims[k];
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_vae_load(hedit_vae* h, const char* name, const float* w, size_t numel, void* stream) try {
  ARG_CHECK(h && name && w, "null");
  return store_load(h, "VAE", name, w, numel, reinterpret_cast<hipStream_t>(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_vae_missing(const hedit_vae* h) { return h ? store_missing(h) : -1; }

static int check_latent(const hedit_vae* h, int lh, int lw) {
  ARG_CHECK(lh >= 1 && lw >= 1 && (lh * lw) % 64 == 0, "latent h*w must be a multiple of 64 (attention tokens)");
  (void)h;
  return HEDIT_OK;
}

size_t hedit_vae_workspace_bytes(hedit_vae* h, int B, int latent_h, int latent_w, int encode) try {
  if (!h || B < 1 || check_latent(h, latent_h, latent_w) != HEDIT_OK) return 0;
  size_t peak = 0;
  const int f = 1 << (h->cfg.n_levels - 1);
  float dummy = 0.f;   // only its address matters in the dry run (selects the gradient pass)
  int rc = encode == 1 ? encode_impl(h, nullptr, B, latent_h * f, latent_w * f, nullptr, nullptr, 0, nullptr, true, &peak)
         : encode == 2 ? decode_impl(h, nullptr, B, latent_h, latent_w, nullptr, nullptr, 0, nullptr, true, &peak, &dummy, &dummy)
                       : decode_impl(h, nullptr, B, latent_h, latent_w, nullptr, nullptr, 0, nullptr, true, &peak);
  return rc == HEDIT_OK ? peak + 4096 : 0;
} catch (...) { (void)hedit_abi_catch(); return 0; }

int hedit_vae_decode(hedit_vae* h, const float* z, int B, int latent_h, int latent_w, float* image, void* workspace,
                     size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && z && image && workspace, "null");
  ARG_CHECK(B >= 1, "B");
  TRY(check_latent(h, latent_h, latent_w));
  if (hedit_vae_missing(h) != 0) {
    hedit_set_error("VAE has " + std::to_string(hedit_vae_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  return decode_impl(h, z, B, latent_h, latent_w, image, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream),
                     false, nullptr);
} catch (...) { return hedit_abi_catch(); }

int hedit_vae_decode_vjp(hedit_vae* h, const float* z, const float* d_image, int B, int latent_h, int latent_w, float* d_z,
                         float* image, void* workspace, size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && z && d_image && d_z && workspace, "null");
  ARG_CHECK(B >= 1, "B");
  TRY(check_latent(h, latent_h, latent_w));
  if (hedit_vae_missing(h) != 0) {
    hedit_set_error("VAE has " + std::to_string(hedit_vae_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  return decode_impl(h, z, B, latent_h, latent_w, image, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream),
                     false, nullptr, d_image, d_z);
} catch (...) { return hedit_abi_catch(); }

int hedit_vae_decode_keep(hedit_vae* h, const float* z, int B, int latent_h, int latent_w, float* image, void* workspace,
                          size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && z && image && workspace, "null");
  ARG_CHECK(B >= 1, "B");
  TRY(check_latent(h, latent_h, latent_w));
  if (hedit_vae_missing(h) != 0) {
    hedit_set_error("VAE has " + std::to_string(hedit_vae_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  drop_tape(h);
  DecodeTape* T = new DecodeTape();
  tape_init(*T, h, B, latent_h, latent_w, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), false);
  const int rc = decode_forward(h, *T, z, image, true);
  if (rc != HEDIT_OK) {
    delete T;
    return rc;
  }
  h->tape = T;
  h->tape_free = [](void* p) { delete reinterpret_cast<DecodeTape*>(p); };
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_vae_decode_backward(hedit_vae* h, const float* d_image, float* d_z, void* workspace, void* stream) try {
  ARG_CHECK(h && d_image && d_z && workspace, "null");
  if (!h->tape) {
    hedit_set_error("hedit_vae_decode_backward: no forward is being kept (call hedit_vae_decode_keep first; one backward per forward)");
    return HEDIT_ERR_STATE;
  }
  DecodeTape* T = reinterpret_cast<DecodeTape*>(h->tape);
  if (T->ws != workspace) {
    hedit_set_error("hedit_vae_decode_backward: not the workspace the kept forward ran in");
    return HEDIT_ERR_ARG;
  }
  T->f.st = reinterpret_cast<hipStream_t>(stream);
  const int rc = decode_backward(h, *T, d_image, d_z);
  drop_tape(h);
  return rc;
} catch (...) { return hedit_abi_catch(); }

int hedit_vae_encode(hedit_vae* h, const float* image, int B, int height, int width, float* mean, void* workspace,
                     size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && image && mean && workspace, "null");
  ARG_CHECK(B >= 1, "B");
  const int f = 1 << (h->cfg.n_levels - 1);
  ARG_CHECK(height % f == 0 && width % f == 0, "image size must be divisible by 2^(levels-1)");
  TRY(check_latent(h, height / f, width / f));
  if (hedit_vae_missing(h) != 0) {
    hedit_set_error("VAE has " + std::to_string(hedit_vae_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  return encode_impl(h, image, B, height, width, mean, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream),
                     false, nullptr);
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
