// The pixel-space DDPM UNet of the face-swapping task as a native executor: the eps-network that
// `h_Edit_R` (reference face-swapping/inversion/h_edit_R.py:71,96,118) and the SDE inversion
// (inversion/sde_inversion.py:121) evaluate, i.e. `Model.forward(x, t)` of
// face-swapping/diffusion/diffusion.py:192-341 (blocks :27-189) -- SURVEY.md section 8 rows a21 / a22.
// Built from the blocks of blocks.h on the kernels of gemm.hip / norm.hip, like vae.hip: parameters
// by their state_dict names (`down.{i}.block.{j}.conv1.weight`, `mid.attn_1.q.weight`, ...), NHWC
// bf16 activations in a caller-provided workspace, one C call per evaluation.
//
// Architecture: sinusoidal timestep embedding ([sin|cos], diffusion.py:6-24) -> two dense layers;
// conv_in; per level `num_res_blocks` ResNet blocks (GroupNorm(32, eps 1e-6) + swish, the timestep
// projection added after conv1) each followed by a single-head spatial attention at the listed
// resolutions, stride-2 conv with (0,1,0,1) padding between levels; mid block; the mirrored up path
// with skip concatenation and 2x nearest upsample + conv; GroupNorm + swish + conv_out.
// One timestep per call (the reference always passes `ones(n) * t`).
#include "blocks.h"

namespace {

struct DLevel {
  std::vector<VRes> block;
  std::vector<VAttn> attn;      // empty, or one per block
  bf16_t* samp_w = nullptr;     // down: stride-2 conv, up: conv after the 2x upsample
  float* samp_b = nullptr;
  int ch = 0;
};

}  // namespace

struct hedit_ddpm : ParamStore {
  hedit_ddpm_cfg cfg;
  int temb_ch = 0;
  bf16_t *t0_w = nullptr, *t1_w = nullptr;
  float *t0_b = nullptr, *t1_b = nullptr;
  float *in_w = nullptr, *in_b = nullptr;
  std::vector<DLevel> down, up;     // up[i] = the reference's up[i] (level i; executed from the last to level 0)
  VRes mid1, mid2;
  VAttn mida;
  float *no_g = nullptr, *no_b = nullptr, *out_b = nullptr;
  bf16_t* out_w = nullptr;
};

namespace {

// GroupNorm statistics from the producing convolutions (gnstat.h) -- default of this build, and hedit_test_set_flags bit 2
// flips it so that tests/test_gpu_gn_stats.py can compare both paths on one library
constexpr bool DDPM_GN_FUSE_DEFAULT = true;
bool ddpm_gn_fuse() { return DDPM_GN_FUSE_DEFAULT != ((hedit_test_flags() & 4) != 0); }

int forward_impl(hedit_ddpm* h, const float* x, float t, int B, float* out, void* ws, size_t ws_bytes, hipStream_t st,
                 bool dry, size_t* peak) {
  VF f{32, B, st, Arena{}};
  f.gn_fuse = ddpm_gn_fuse();
  f.ar.dry = dry;
  f.ar.base = reinterpret_cast<char*>(ws);
  f.ar.cap = ws_bytes;
  const hedit_ddpm_cfg& c = h->cfg;
  const int L = c.n_levels, nrb = c.num_res_blocks, ch = c.ch, tc = h->temb_ch;
  // timestep embedding -> dense0 -> swish -> dense1 (diffusion.py:296-300)
  float *emb, *t0, *temb;
  TRY(aalloc(f, &emb, (size_t)ch));
  TRY(aalloc(f, &t0, (size_t)tc));
  TRY(aalloc(f, &temb, (size_t)tc));
  RUN(f, timestep_embed_ddpm_launch(t, emb, ch, st));
  RUN(f, gemv_launch(h->t0_w, emb, h->t0_b, nullptr, t0, tc, ch, 0, st));
  RUN(f, gemv_launch(h->t1_w, t0, h->t1_b, nullptr, temb, tc, tc, 1, st));

  int H = c.image_size, W = c.image_size;
  // st: the tensor's GroupNorm pair statistics, taken by the convolution that produced it (gnstat.h), or nullptr: every
  // GroupNorm whose input carries them skips its own statistics pass over the tensor
  struct Skip { bf16_t* p; int ch; float* st; };
  std::vector<Skip> hs;
  auto stat_of = [](const Skip& s) { return GnStat{s.st, s.ch, nullptr, 0}; };
  bf16_t* x0;
  TRY(aalloc(f, &x0, (size_t)B * H * W * ch));
  RUN(f, conv_in_launch(x, h->in_w, h->in_b, x0, B, c.in_channels, H, W, ch, st));
  hs.push_back({x0, ch, nullptr});
  for (int i = 0; i < L; ++i) {
    const DLevel& lv = h->down[i];
    for (int j = 0; j < nrb; ++j) {
      bf16_t* y;
      float* yst = nullptr;
      const GnStat xs = stat_of(hs.back());
      TRY(resblock(f, lv.block[j], hs.back().p, H, W, &y, nullptr, temb, nullptr, 0, &xs, &yst));
      if (!lv.attn.empty()) {
        bf16_t* a;
        TRY(attention(f, lv.attn[j], y, H, W, &a));
        f.ar.free(y);
        if (yst) f.ar.free(yst);
        yst = nullptr;
        y = a;
      }
      hs.push_back({y, lv.ch, yst});
    }
    if (lv.samp_w) {
      bf16_t* y;
      float* yst = nullptr;
      TRY(aalloc(f, &y, (size_t)B * (H / 2) * (W / 2) * lv.ch));
      if (gn_stats_shape(f, (H / 2) * (W / 2), lv.ch)) TRY(aalloc(f, &yst, (size_t)B * (H / 2) * (W / 2) / 128 * lv.ch));
      TRY(conv3x3(f, hs.back().p, H, W, lv.ch, lv.samp_w, lv.ch, lv.samp_b, nullptr, y, 2, 0, yst));
      H /= 2; W /= 2;
      hs.push_back({y, lv.ch, yst});
    }
  }
  // middle (the last skip stays on the stack: it is popped by the first up block)
  bf16_t *m1, *m2, *cur;
  float* cur_st = nullptr;
  int cur_ch = hs.back().ch;
  {
    const GnStat xs = stat_of(hs.back());
    TRY(resblock(f, h->mid1, hs.back().p, H, W, &m1, nullptr, temb, nullptr, 0, &xs));
  }
  TRY(attention(f, h->mida, m1, H, W, &m2));
  f.ar.free(m1);
  TRY(resblock(f, h->mid2, m2, H, W, &cur, nullptr, temb, nullptr, 0, nullptr, &cur_st));
  f.ar.free(m2);
  // up path.  As in unet.hip, the block that produces `cur` writes it straight into the left columns of the next
  // concatenation buffer; only the skip half is copied.
  bool in_cat = false;
  for (int i = L - 1; i >= 0; --i) {
    const DLevel& lv = h->up[i];
    for (int j = 0; j < nrb + 1; ++j) {
      const Skip s = hs.back();
      hs.pop_back();
      bf16_t *cat, *y;
      const size_t M = (size_t)B * H * W;
      if (in_cat) {
        cat = cur;
        RUN(f, concat_launch(nullptr, cur_ch, s.p, s.ch, cat, (long)M, st));
      } else {
        TRY(aalloc(f, &cat, M * (cur_ch + s.ch)));
        RUN(f, concat_launch(cur, cur_ch, s.p, s.ch, cat, (long)M, st));
        f.ar.free(cur);
      }
      f.ar.free(s.p);
      const bool to_cat = j < nrb;          // the next consumer is the next block of this level
      bf16_t* dst = nullptr;
      int ldd = 0;
      if (to_cat) {
        ldd = lv.ch + hs.back().ch;
        TRY(aalloc(f, &dst, M * ldd));
      }
      // statistics of the concatenation = the pairs of its two producers (both or nothing)
      const GnStat xs = (cur_st && s.st) ? GnStat{cur_st, cur_ch, s.st, s.ch} : GnStat{};
      // the output's statistics, unless its consumer is the upsampling convolution
      const bool want_st = !(j == nrb && lv.samp_w);
      float* yst = nullptr;
      if (!lv.attn.empty()) {
        TRY(resblock(f, lv.block[j], cat, H, W, &y, nullptr, temb, nullptr, 0, &xs));
        f.ar.free(cat);
        bf16_t* a;
        TRY(attention(f, lv.attn[j], y, H, W, &a, nullptr, dst, ldd));
        f.ar.free(y);
        y = a;
      } else {
        TRY(resblock(f, lv.block[j], cat, H, W, &y, nullptr, temb, dst, ldd, &xs, want_st ? &yst : nullptr));
        f.ar.free(cat);
      }
      if (cur_st) f.ar.free(cur_st);
      if (s.st) f.ar.free(s.st);
      cur = y;
      cur_st = yst;
      cur_ch = lv.ch;
      in_cat = to_cat;
    }
    if (lv.samp_w) {
      const int ldd = cur_ch + hs.back().ch;     // left half of the next level's first concatenation
      bf16_t* y;
      float* yst = nullptr;
      TRY(aalloc(f, &y, (size_t)B * H * W * 4 * ldd));
      if (gn_stats_shape(f, 4 * H * W, cur_ch)) TRY(aalloc(f, &yst, (size_t)B * H * W * 4 / 128 * cur_ch));
      TRY(conv3x3(f, cur, H, W, cur_ch, lv.samp_w, cur_ch, lv.samp_b, nullptr, y, 3, ldd, yst));
      f.ar.free(cur);
      if (cur_st) f.ar.free(cur_st);
      cur = y;
      cur_st = yst;
      H *= 2; W *= 2;
      in_cat = true;
    }
  }
  bf16_t* xn;
  TRY(aalloc(f, &xn, (size_t)B * H * W * cur_ch));
  {
    const GnStat xs{cur_st, cur_ch, nullptr, 0};
    TRY(groupnorm(f, cur, xn, h->no_g, h->no_b, H * W, cur_ch, 1, nullptr, cur_st ? &xs : nullptr));
  }
  if (cur_st) f.ar.free(cur_st);
  f.ar.free(cur);
  TRY(conv_out(f, xn, H, W, cur_ch, h->out_w, h->out_b, c.out_ch, out));
  f.ar.free(xn);
  f.ar.free(temb); f.ar.free(t0); f.ar.free(emb);
  if (peak) *peak = f.ar.peak;
  return HEDIT_OK;
}

}  // namespace

extern "C" {

int hedit_ddpm_create(const hedit_ddpm_cfg* cfg, hedit_ddpm** out) try {
  ARG_CHECK(cfg && out, "null");
  ARG_CHECK(cfg->n_levels >= 1 && cfg->n_levels <= 8, "n_levels in 1..8");
  ARG_CHECK(cfg->in_channels >= 1 && cfg->in_channels <= 8 && cfg->out_ch >= 1 && cfg->out_ch <= 4, "in_channels <= 8, out_ch <= 4");
  ARG_CHECK(cfg->ch % 64 == 0, "ch must be a multiple of 64");
  ARG_CHECK(cfg->num_res_blocks >= 1 && cfg->num_res_blocks <= 4, "num_res_blocks in 1..4");
  const int L = cfg->n_levels;
  ARG_CHECK(cfg->image_size % (1 << (L - 1)) == 0, "image_size must be divisible by 2^(n_levels-1)");
  int res = cfg->image_size;
  for (int i = 0; i < L; ++i) {
    ARG_CHECK(cfg->ch_mult[i] >= 1, "ch_mult >= 1");
    if (cfg->attn_level[i]) ARG_CHECK((res * res) % 64 == 0, "attention levels need h*w % 64 == 0");
    if (i < L - 1) res /= 2;
  }
  ARG_CHECK((res * res) % 64 == 0, "the mid block's h*w must be a multiple of 64");
  TRY(gemm_prepare());
  hedit_ddpm* h = new hedit_ddpm();
  h->cfg = *cfg;
  const int ch = cfg->ch, tc = 4 * ch, nrb = cfg->num_res_blocks;
  h->temb_ch = tc;
  const BlockNames& nm = NAMES_DDPM;
  h->t0_w = lin(h, "temb.dense.0.weight", tc, ch);
  h->t0_b = vec(h, "temb.dense.0.bias", tc);
  h->t1_w = lin(h, "temb.dense.1.weight", tc, tc);
  h->t1_b = vec(h, "temb.dense.1.bias", tc);
  h->in_w = f32conv(h, "conv_in.weight", ch, cfg->in_channels, 3);
  h->in_b = vec(h, "conv_in.bias", ch);
  auto mult_in = [&](int i) { return i == 0 ? 1 : cfg->ch_mult[i - 1]; };   // in_ch_mult = (1,) + ch_mult
  int block_in = ch;
  for (int i = 0; i < L; ++i) {
    DLevel lv;
    const std::string pre = "down." + std::to_string(i);
    block_in = ch * mult_in(i);
    const int block_out = ch * cfg->ch_mult[i];
    for (int j = 0; j < nrb; ++j) {
      lv.block.push_back(make_res(h, pre + ".block." + std::to_string(j), block_in, block_out, false, nm, tc));
      block_in = block_out;
      if (cfg->attn_level[i]) lv.attn.push_back(make_attn(h, pre + ".attn." + std::to_string(j), block_in, false, nm));
    }
    lv.ch = block_in;
    if (i != L - 1) {
      lv.samp_w = conv3(h, pre + ".downsample.conv.weight", block_in, block_in);
      lv.samp_b = vec(h, pre + ".downsample.conv.bias", block_in);
    }
    h->down.push_back(lv);
  }
  h->mid1 = make_res(h, "mid.block_1", block_in, block_in, false, nm, tc);
  h->mida = make_attn(h, "mid.attn_1", block_in, false, nm);
  h->mid2 = make_res(h, "mid.block_2", block_in, block_in, false, nm, tc);
  h->up.resize(L);
  for (int i = L - 1; i >= 0; --i) {
    DLevel lv;
    const std::string pre = "up." + std::to_string(i);
    const int block_out = ch * cfg->ch_mult[i];
    int skip_in = ch * cfg->ch_mult[i];
    for (int j = 0; j < nrb + 1; ++j) {
      if (j == nrb) skip_in = ch * mult_in(i);
      lv.block.push_back(make_res(h, pre + ".block." + std::to_string(j), block_in + skip_in, block_out, false, nm, tc));
      block_in = block_out;
      if (cfg->attn_level[i]) lv.attn.push_back(make_attn(h, pre + ".attn." + std::to_string(j), block_in, false, nm));
    }
    lv.ch = block_in;
    if (i != 0) {
      lv.samp_w = conv3(h, pre + ".upsample.conv.weight", block_in, block_in);
      lv.samp_b = vec(h, pre + ".upsample.conv.bias", block_in);
    }
    h->up[i] = lv;
  }
  h->no_g = vec(h, "norm_out.weight", block_in);
  h->no_b = vec(h, "norm_out.bias", block_in);
  h->out_w = conv3(h, "conv_out.weight", cfg->out_ch, block_in);
  h->out_b = vec(h, "conv_out.bias", cfg->out_ch);
  if (h->alloc_failed) {
    hedit_set_error("hipMalloc failed while creating the DDPM UNet");
    hedit_ddpm_destroy(h);
    return HEDIT_ERR_HIP;
  }
  *out = h;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

void hedit_ddpm_destroy(hedit_ddpm* h) try {
  if (!h) return;
  store_free(h);
  delete h;
} catch (...) { (void)hedit_abi_catch(); }

int hedit_ddpm_num_params(const hedit_ddpm* h) { return h ? (int)h->slots.size() : 0; }
const char* hedit_ddpm_param_name(const hedit_ddpm* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
} catch (...) { (void)hedit_abi_catch(); return nullptr; }
int hedit_ddpm_param_shape(const hedit_ddpm* h, int i, int* ndim, int* dims4) try {
  ARG_CHECK(h && ndim && dims4 && i >= 0 && i < (int)h->slots.size(), "param index");
  *ndim = h->slots[i].ndim;
  for (int k = 0; k < 4; ++k) dims4[k] = h->slots[i].dims[k];
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }
int hedit_ddpm_load(hedit_ddpm* h, const char* name, const float* w, size_t numel, void* stream) try {
  ARG_CHECK(h && name && w, "null");
  return store_load(h, "DDPM UNet", name, w, numel, reinterpret_cast<hipStream_t>(stream));
} catch (...) { return hedit_abi_catch(); }
int hedit_ddpm_missing(const hedit_ddpm* h) { return h ? store_missing(h) : -1; }

size_t hedit_ddpm_workspace_bytes(hedit_ddpm* h, int B) try {
  if (!h || B < 1) return 0;
  size_t peak = 0;
  const int rc = forward_impl(h, nullptr, 0.f, B, nullptr, nullptr, 0, nullptr, true, &peak);
  return rc == HEDIT_OK ? peak + 4096 : 0;
} catch (...) { (void)hedit_abi_catch(); return 0; }

int hedit_ddpm_forward(hedit_ddpm* h, const float* x, float t, int B, float* eps, void* workspace, size_t workspace_bytes,
                       void* stream) try {
  ARG_CHECK(h && x && eps && workspace, "null");
  ARG_CHECK(B >= 1, "B");
  if (store_missing(h) != 0) {
    hedit_set_error("DDPM UNet has " + std::to_string(store_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  return forward_impl(h, x, t, B, eps, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), false, nullptr);
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
