// The identity reward of the face-swapping task as a native executor: `IDLoss.get_cosine_loss` of the reference's
// face-swapping/arcface/arcface_model.py:40-67 -- crop + adaptive pool -> IR-SE50 backbone (facial_recognition/
// model_irse.py:9-48, helpers.py:28-119: BatchNorm in eval mode, PReLU, squeeze-excitation, strided shortcuts)
// -> l2-normalised 512-d feature -> 1 - cos(feature, reference feature) -- TOGETHER WITH its gradient w.r.t. the image,
// which is all the h-Edit loop takes from it (inversion/h_edit_R.py:103-111).  SURVEY.md section 8 row a23 /
// boundary entry `hedit_irse50_cos_fwd_bwd`.
//
// fp32-quality arithmetic on the bf16 matrix cores: every contraction is the three-term split product of pnet.hip
// on the implicit-GEMM kernel of gemm.hip, activations stay fp32 in HBM; BatchNorms are folded (into the following
// split pass when they precede a zero-padded convolution, into the weights when they follow one).  Parameters are
// addressed by the reference's state_dict names (`input_layer.0.weight`, `body.3.res_layer.5.fc1.weight`, ...).
#include "pnet.h"

namespace {

constexpr float BN_EPS = 1e-5f;
constexpr int IR_STAGES[4][3] = {{64, 64, 3}, {64, 128, 4}, {128, 256, 14}, {256, 512, 3}};   // helpers.py:37-44 (IR-50)

struct BN { float *g, *b, *m, *v; };

struct IUnit {
  int cin, depth, stride;
  bool conv_sc;
  BN bn1, bn2, bnsc;
  float *w1, *prelu, *w2, *fc1, *fc2, *wsc;
  float *p1, *q1, *p2, *q2, *psc, *qsc;      // folded BatchNorms: y = p x + q
  PConv c1, c2, sc;
};

}  // namespace

struct hedit_irse : ParamStore {
  float *in_w = nullptr, *in_prelu = nullptr, *in_p = nullptr, *in_q = nullptr;
  BN in_bn{}, out_bn2{}, out_bn1{};
  PConv in_c, lin;
  std::vector<IUnit> units;
  float *lin_w = nullptr, *lin_b = nullptr;
  float *out_p = nullptr, *out_q = nullptr;          // BatchNorm2d in front of the flatten, tiled over the 49 pixels
  float *lin_p = nullptr, *lin_q = nullptr;          // Linear bias + BatchNorm1d folded
  bool finalized = false;
};

namespace {

BN make_bn(ParamStore* h, const std::string& pre, int C) {
  BN b;
  b.g = vec(h, pre + ".weight", C);
  b.b = vec(h, pre + ".bias", C);
  b.m = vec(h, pre + ".running_mean", C);
  b.v = vec(h, pre + ".running_var", C);
  return b;
}

struct UnitTape { float *X, *z1, *u, *s, *hb; int H, W; };

// X [B][H][W][cin] -> Y [B][H/s][W/s][depth] (allocated; X is NOT freed: it is the tape's).  helpers.py:97-119
int unit_fwd(PF& f, const IUnit& u, const float* X, int H, int W, float** Yout, UnitTape* tape) {
  const int B = f.B, Ho = H / u.stride, Wo = W / u.stride;
  const long M = (long)B * H * W, M2 = (long)B * Ho * Wo;
  bf16_t *A1, *A2, *A3 = nullptr;
  float *z1, *uu, *pool, *hb, *sb, *sc = nullptr, *Y;
  TRY(op_split(f, X, u.cin, P_AFFINE, u.p1, u.q1, 0, nullptr, 0, H, W, M, &A1));               // BatchNorm -> conv1 operand
  TRY(pgemm(f, A1, u.c1, false, 1, H, W, M, &z1));
  f.ar.free(A1);
  TRY(op_split(f, z1, u.depth, P_PRELU, u.prelu, nullptr, 0, nullptr, 0, H, W, M, &A2));      // PReLU -> conv2 operand
  TRY(pgemm(f, A2, u.c2, false, u.stride == 2 ? 2 : 1, H, W, M2, &uu));                       // BatchNorm folded: u + q2
  f.ar.free(A2);
  TRY(palloc(f, &pool, (size_t)B * se_nslab(Ho * Wo) * u.depth));
  TRY(palloc(f, &hb, (size_t)B * (u.depth / 16)));
  TRY(palloc(f, &sb, (size_t)B * u.depth));
  if (!f.dry()) {
    TRY(se_pool_launch(uu, pool, B, Ho * Wo, u.depth, f.st));
    TRY(se_fc_launch(pool, u.q2, u.fc1, u.fc2, hb, sb, B, Ho * Wo, u.depth, u.depth / 16, f.st));
  }
  f.ar.free(pool);
  if (u.conv_sc) {
    TRY(op_split(f, X, u.cin, P_COPY, nullptr, nullptr, 0, nullptr, u.stride == 2 ? 1 : 0, H, W, M, &A3));
    TRY(pgemm(f, A3, u.sc, false, 0, Ho, Wo, M2, &sc));
    f.ar.free(A3);
  }
  TRY(palloc(f, &Y, (size_t)M2 * u.depth));
  if (!f.dry()) TRY(se_combine_launch(uu, u.q2, sb, sc, u.qsc, X, u.stride, Y, B, Ho, Wo, u.depth, f.st));
  if (sc) f.ar.free(sc);
  *tape = UnitTape{const_cast<float*>(X), z1, uu, sb, hb, H, W};
  *Yout = Y;
  return HEDIT_OK;
}

// dY [B][H/s][W/s][depth] -> dX [B][H][W][cin] (allocated; dY is NOT freed)
int unit_bwd(PF& f, const IUnit& u, const UnitTape& t, const float* dY, float** dXout) {
  const int B = f.B, H = t.H, W = t.W, Ho = H / u.stride, Wo = W / u.stride;
  const long M = (long)B * H * W, M2 = (long)B * Ho * Wo;
  float *gs, *rb, *dz, *dxa, *dsc = nullptr, *dX;
  bf16_t *A, *A2, *A3;
  TRY(palloc(f, &gs, (size_t)B * se_nslab(Ho * Wo) * u.depth));
  TRY(palloc(f, &rb, (size_t)B * u.depth));
  if (!f.dry()) {
    TRY(se_bwd_reduce_launch(dY, t.u, u.q2, gs, B, Ho * Wo, u.depth, f.st));
    TRY(se_fc_bwd_launch(gs, t.s, t.hb, u.fc1, u.fc2, rb, B, u.depth, u.depth / 16, Ho * Wo, f.st));
  }
  // d(u + q2) = dY * s + r, zero-stuffed to the input resolution when conv2 was strided, then conv2's input gradient
  TRY(op_split(f, dY, u.depth, P_AFFINE, t.s, rb, 1, nullptr, u.stride == 2 ? 2 : 0, Ho, Wo, M2, &A));
  TRY(pgemm(f, A, u.c2, true, 1, H, W, M, &dz));
  f.ar.free(A);
  f.ar.free(gs);
  f.ar.free(rb);
  TRY(op_split(f, dz, u.depth, P_PRELU_GRAD, u.prelu, nullptr, 0, t.z1, 0, H, W, M, &A2));
  f.ar.free(dz);
  TRY(pgemm(f, A2, u.c1, true, 1, H, W, M, &dxa));
  f.ar.free(A2);
  const float* dshort = dY;
  if (u.conv_sc) {
    TRY(op_split(f, dY, u.depth, P_COPY, nullptr, nullptr, 0, nullptr, 0, Ho, Wo, M2, &A3));
    TRY(pgemm(f, A3, u.sc, true, 0, Ho, Wo, M2, &dsc));
    f.ar.free(A3);
    dshort = dsc;
  }
  TRY(palloc(f, &dX, (size_t)M * u.cin));
  if (!f.dry()) TRY(unit_bwd_combine_launch(dxa, u.p1, dshort, u.stride, dX, B, H, W, u.cin, f.st));
  f.ar.free(dxa);
  if (dsc) f.ar.free(dsc);
  *dXout = dX;
  return HEDIT_OK;
}

// image [B][3][256][256] -> feat (optional, l2-normalised [B][512]); with ref: loss [B] = 1 - cos, d_image = d(scale * sum_b loss_b)/d image
int run(hedit_irse* h, const float* image, const float* ref, int ref_stride, int B, float* feat, float* loss, float* d_image, float scale,
        void* ws, size_t ws_bytes, hipStream_t st, bool dry, size_t* peak) {
  PF f{B, st, Arena{}};
  f.ar.dry = dry;
  f.ar.base = reinterpret_cast<char*>(ws);
  f.ar.cap = ws_bytes;
  const bool grad = d_image != nullptr || dry;
  const long M0 = (long)B * 112 * 112;
  float *a0, *z0, *x0;
  bf16_t* A;
  TRY(palloc(f, &a0, (size_t)M0 * 3));
  if (!dry) TRY(face_pool_launch(image, a0, B, st));
  TRY(op_split(f, a0, 3, P_COPY, nullptr, nullptr, 0, nullptr, 0, 112, 112, M0, &A));
  f.ar.free(a0);
  TRY(pgemm(f, A, h->in_c, false, 1, 112, 112, M0, &z0));          // BatchNorm folded into the weights: z0 + in_q
  f.ar.free(A);
  TRY(palloc(f, &x0, (size_t)M0 * 64));
  if (!dry) TRY(act_launch(z0, h->in_prelu, h->in_q, x0, M0 * 64, 64, P_PRELU, st));
  std::vector<UnitTape> tapes(h->units.size());
  float* X = x0;
  int H = 112, W = 112;
  for (size_t i = 0; i < h->units.size(); ++i) {
    float* Y;
    TRY(unit_fwd(f, h->units[i], X, H, W, &Y, &tapes[i]));
    H /= h->units[i].stride; W /= h->units[i].stride;
    if (!grad) {      // features only: nothing is kept
      f.ar.free(tapes[i].z1); f.ar.free(tapes[i].u); f.ar.free(tapes[i].s); f.ar.free(tapes[i].hb);
      f.ar.free(X);
    }
    X = Y;
  }
  if (!grad) f.ar.free(z0);
  // output_layer: BatchNorm2d -> (Dropout: eval) -> Flatten -> Linear -> BatchNorm1d ; then the two normalisations + cosine
  const int K = 7 * 7 * 512;
  float *fr, *df = nullptr;
  TRY(op_split(f, X, K, P_AFFINE, h->out_p, h->out_q, 0, nullptr, 0, 1, 1, B, &A));
  TRY(pgemm(f, A, h->lin, false, 0, 1, 1, B, &fr));
  f.ar.free(A);
  if (grad) TRY(palloc(f, &df, (size_t)B * 512));
  if (!dry) TRY(cos_head_launch(fr, h->lin_q, ref, ref_stride, feat, loss, grad ? df : nullptr, B, 512, scale, st));
  f.ar.free(fr);
  if (grad) {
    float *dflat, *dX;
    TRY(op_split(f, df, 512, P_COPY, nullptr, nullptr, 0, nullptr, 0, 1, 1, B, &A));
    TRY(pgemm(f, A, h->lin, true, 0, 1, 1, B, &dflat));
    f.ar.free(A);
    f.ar.free(df);
    TRY(palloc(f, &dX, (size_t)B * K));
    if (!dry) TRY(scale_cols_launch(dflat, h->out_p, dX, (long)B * K, K, st));
    f.ar.free(dflat);
    f.ar.free(X);
    for (int i = (int)h->units.size() - 1; i >= 0; --i) {
      float* dXn;
      TRY(unit_bwd(f, h->units[i], tapes[i], dX, &dXn));
      f.ar.free(dX);
      f.ar.free(tapes[i].z1); f.ar.free(tapes[i].u); f.ar.free(tapes[i].s); f.ar.free(tapes[i].hb);
      if (i > 0) f.ar.free(tapes[i].X);
      dX = dXn;
    }
    // stem: x0 = prelu(z0 + in_q); conv (3 -> 64) input gradient has 3 (padded to 4) columns
    float* da0;
    TRY(op_split(f, dX, 64, P_PRELU_GRAD, h->in_prelu, h->in_q, 0, z0, 0, 112, 112, M0, &A));
    f.ar.free(dX);
    f.ar.free(x0);
    f.ar.free(z0);
    TRY(pgemm(f, A, h->in_c, true, 1, 112, 112, M0, &da0));
    f.ar.free(A);
    if (!dry) TRY(face_pool_bwd_launch(da0, h->in_c.rows_b, d_image, B, st));
    f.ar.free(da0);
  } else {
    f.ar.free(X);
  }
  if (peak) *peak = f.ar.peak;
  return HEDIT_OK;
}

}  // namespace

extern "C" {

int hedit_irse50_create(hedit_irse** out) try {
  ARG_CHECK(out, "null");
  TRY(gemm_prepare());
  hedit_irse* h = new hedit_irse();
  h->in_w = f32conv(h, "input_layer.0.weight", 64, 3, 3);
  h->in_bn = make_bn(h, "input_layer.1", 64);
  h->in_prelu = vec(h, "input_layer.2.weight", 64);
  int idx = 0;
  for (int s = 0; s < 4; ++s) {
    const int cin0 = IR_STAGES[s][0], depth = IR_STAGES[s][1], n = IR_STAGES[s][2];
    for (int j = 0; j < n; ++j, ++idx) {
      IUnit u{};
      u.cin = j == 0 ? cin0 : depth;
      u.depth = depth;
      u.stride = j == 0 ? 2 : 1;
      u.conv_sc = u.cin != depth;
      const std::string pre = "body." + std::to_string(idx);
      if (u.conv_sc) {
        u.wsc = f32conv(h, pre + ".shortcut_layer.0.weight", depth, u.cin, 1);
        u.bnsc = make_bn(h, pre + ".shortcut_layer.1", depth);
      }
      u.bn1 = make_bn(h, pre + ".res_layer.0", u.cin);
      u.w1 = f32conv(h, pre + ".res_layer.1.weight", depth, u.cin, 3);
      u.prelu = vec(h, pre + ".res_layer.2.weight", depth);
      u.w2 = f32conv(h, pre + ".res_layer.3.weight", depth, depth, 3);
      u.bn2 = make_bn(h, pre + ".res_layer.4", depth);
      u.fc1 = f32conv(h, pre + ".res_layer.5.fc1.weight", depth / 16, depth, 1);
      u.fc2 = f32conv(h, pre + ".res_layer.5.fc2.weight", depth, depth / 16, 1);
      u.p1 = dalloc<float>(h, u.cin); u.q1 = dalloc<float>(h, u.cin);
      u.p2 = dalloc<float>(h, depth); u.q2 = dalloc<float>(h, depth);
      u.psc = dalloc<float>(h, depth); u.qsc = dalloc<float>(h, depth);
      h->units.push_back(u);
    }
  }
  h->out_bn2 = make_bn(h, "output_layer.0", 512);
  h->lin_w = dalloc<float>(h, (size_t)512 * 25088);
  add_slot(h, "output_layer.3.weight", 0, h->lin_w, (size_t)512 * 25088, 512, 25088, 2, 512, 25088, 1, 1);
  h->lin_b = vec(h, "output_layer.3.bias", 512);
  h->out_bn1 = make_bn(h, "output_layer.4", 512);
  h->in_p = dalloc<float>(h, 64); h->in_q = dalloc<float>(h, 64);
  h->out_p = dalloc<float>(h, 25088); h->out_q = dalloc<float>(h, 25088);
  h->lin_p = dalloc<float>(h, 512); h->lin_q = dalloc<float>(h, 512);
  if (h->alloc_failed) {
    hedit_set_error("hipMalloc failed while creating the IR-SE50 backbone");
    store_free(h);
    delete h;
    return HEDIT_ERR_HIP;
  }
  *out = h;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

void hedit_irse50_destroy(hedit_irse* h) try {
  if (!h) return;
  store_free(h);
  delete h;
} catch (...) { (void)hedit_abi_catch(); }

int hedit_irse50_num_params(const hedit_irse* h) { return h ? (int)h->slots.size() : 0; }
const char* hedit_irse50_param_name(const hedit_irse* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
} catch (...) { (void)hedit_abi_catch(); return nullptr; }
int hedit_irse50_param_shape(const hedit_irse* h, int i, int* ndim, int* dims4) try {
  ARG_CHECK(h && ndim && dims4 && i >= 0 && i < (int)h->slots.size(), "param index");
  *ndim = h->slots[i].ndim;
  for (int k = 0; k < 4; ++k) dims4[k] = h->slots[i].dims[k];
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }
int hedit_irse50_load(hedit_irse* h, const char* name, const float* w, size_t numel, void* stream) try {
  ARG_CHECK(h && name && w, "null");
  h->finalized = false;
  return store_load(h, "IR-SE50", name, w, numel, reinterpret_cast<hipStream_t>(stream));
} catch (...) { return hedit_abi_catch(); }
int hedit_irse50_missing(const hedit_irse* h) { return h ? store_missing(h) : -1; }

/* fold the BatchNorms and build the split-bf16 GEMM operands (forward and input-gradient); call once after loading */
int hedit_irse50_finalize(hedit_irse* h, void* stream) try {
  ARG_CHECK(h, "null");
  if (store_missing(h) != 0) {
    hedit_set_error("IR-SE50 has " + std::to_string(store_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TRY(bn_affine_launch(h->in_bn.g, h->in_bn.b, h->in_bn.m, h->in_bn.v, BN_EPS, h->in_p, h->in_q, 64, 1, st));
  TRY(make_pconv(h, h->in_c, h->in_w, h->in_p, 64, 3, 3, 0, 0, st));
  for (IUnit& u : h->units) {
    TRY(bn_affine_launch(u.bn1.g, u.bn1.b, u.bn1.m, u.bn1.v, BN_EPS, u.p1, u.q1, u.cin, 1, st));
    TRY(bn_affine_launch(u.bn2.g, u.bn2.b, u.bn2.m, u.bn2.v, BN_EPS, u.p2, u.q2, u.depth, 1, st));
    TRY(make_pconv(h, u.c1, u.w1, nullptr, u.depth, u.cin, 3, 0, 0, st));
    TRY(make_pconv(h, u.c2, u.w2, u.p2, u.depth, u.depth, 3, 0, 0, st));
    if (u.conv_sc) {
      TRY(bn_affine_launch(u.bnsc.g, u.bnsc.b, u.bnsc.m, u.bnsc.v, BN_EPS, u.psc, u.qsc, u.depth, 1, st));
      TRY(make_pconv(h, u.sc, u.wsc, u.psc, u.depth, u.cin, 1, 0, 0, st));
    }
  }
  TRY(bn_affine_launch(h->out_bn2.g, h->out_bn2.b, h->out_bn2.m, h->out_bn2.v, BN_EPS, h->out_p, h->out_q, 512, 49, st));
  TRY(bn_fold_bias_launch(h->lin_b, h->out_bn1.g, h->out_bn1.b, h->out_bn1.m, h->out_bn1.v, BN_EPS, h->lin_p, h->lin_q, 512, st));
  // Linear over the NCHW flatten (index c * 49 + hw) of our NHWC activations (index hw * 512 + c)
  TRY(make_pconv(h, h->lin, h->lin_w, h->lin_p, 512, 25088, 1, 49, 512, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h->alloc_failed) { hedit_set_error("hipMalloc failed while packing the IR-SE50 weights"); return HEDIT_ERR_HIP; }
  h->finalized = true;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

size_t hedit_irse50_workspace_bytes(hedit_irse* h, int B) try {
  if (!h || B < 1) return 0;
  size_t peak = 0;
  if (run(h, nullptr, nullptr, 0, B, nullptr, nullptr, nullptr, 1.f, nullptr, 0, nullptr, true, &peak) != HEDIT_OK) return 0;
  return peak + 4096;
} catch (...) { (void)hedit_abi_catch(); return 0; }

/* image fp32 [B][3][256][256] in [-1, 1] -> feat fp32 [B][512], the l2-normalised identity feature
 * (IDLoss.extract_feats followed by F.normalize, arcface_model.py:40-52) */
int hedit_irse50_features(hedit_irse* h, const float* image, int B, float* feat, void* workspace, size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && image && feat && workspace && B >= 1, "irse50_features args");
  if (!h->finalized) { hedit_set_error("call hedit_irse50_finalize after loading the parameters"); return HEDIT_ERR_STATE; }
  return run(h, image, nullptr, 0, B, feat, nullptr, nullptr, 1.f, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), false, nullptr);
} catch (...) { return hedit_abi_catch(); }

/* loss[b] = 1 - cos(feature(image_b), ref_feat) and d_image = d(scale * sum_b loss[b]) / d image in ONE call
 * (IDLoss.get_cosine_loss + torch.autograd.grad of h_edit_R.py:103-106: scale = 1 / B reproduces the batch mean).
 * ref_feat: l2-normalised fp32 [512] shared by the batch (ref_per_image = 0) or [B][512]. */
int hedit_irse50_cos_fwd_bwd(hedit_irse* h, const float* image, const float* ref_feat, int ref_per_image, int B, float scale,
                             float* loss, float* d_image, void* workspace, size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && image && ref_feat && loss && d_image && workspace && B >= 1, "irse50_cos_fwd_bwd args");
  if (!h->finalized) { hedit_set_error("call hedit_irse50_finalize after loading the parameters"); return HEDIT_ERR_STATE; }
  return run(h, image, ref_feat, ref_per_image ? 512 : 0, B, nullptr, loss, d_image, scale, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), false, nullptr);
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
