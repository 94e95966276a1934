// The perceptual reward of the face-swapping task as a native executor: `LPIPS_Loss.get_lpips_loss` of the reference's
// face-swapping/arcface/arcface_model.py:69-94 = lpips.LPIPS(net='vgg')(x, src).mean() (third-party package lpips==0.1.4,
// absent offline; its published algorithm is restated here: ScalingLayer -> torchvision VGG16 features, taps relu1_2,
// relu2_2, relu3_3, relu4_3, relu5_3 -> channel-unit-normalised features -> squared difference -> non-negative 1x1
// "lin" weights -> spatial mean -> sum over the five taps) TOGETHER WITH its gradient w.r.t. x, which is all
// inversion/h_edit_R.py:124-132 takes from it.  SURVEY.md section 8 row a24.  The source image's normalised features
// are computed once (hedit_lpips_source) and reused by every call.
//
// Same arithmetic family as irse.hip: fp32 activations, three-term split-bf16 contractions on the implicit-GEMM kernel
// (pnet.hip / gemm.hip); ReLU, bias and the 2 x 2 max pooling are fused into the passes that build the GEMM operands.
// Parameters by the lpips state_dict names (`net.slice1.0.weight`, ..., `net.slice5.28.bias`, `lin0.model.1.weight`, ...).
#include "pnet.h"

namespace {

constexpr int VGG_N = 13;
constexpr int VGG_CH[VGG_N][2] = {{3, 64}, {64, 64}, {64, 128}, {128, 128}, {128, 256}, {256, 256}, {256, 256},
                                  {256, 512}, {512, 512}, {512, 512}, {512, 512}, {512, 512}, {512, 512}};
constexpr int VGG_SLICE[VGG_N] = {1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5};
constexpr int VGG_IDX[VGG_N] = {0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28};       // torchvision vgg16.features indices
// tap after conv l (0-based) -> lin index, or -1; a 2 x 2 max pool follows taps 0..3
inline int tap_of(int l) { return l == 1 ? 0 : (l == 3 ? 1 : (l == 6 ? 2 : (l == 9 ? 3 : (l == 12 ? 4 : -1)))); }

}  // namespace

struct hedit_lpips : ParamStore {
  float* w[VGG_N] = {};
  float* b[VGG_N] = {};
  PConv conv[VGG_N];
  float* lin[5] = {};
  float shift[3] = {-0.030f, -0.088f, -0.188f}, scale[3] = {0.458f, 0.448f, 0.450f};   // lpips ScalingLayer buffers
  bool finalized = false;
};

namespace {

size_t feature_floats(int H, int W) {
  size_t n = 0;
  int h = H, w = W;
  for (int l = 0; l < VGG_N; ++l) {
    if (tap_of(l) >= 0) {
      n += (size_t)h * w * VGG_CH[l][1];
      h /= 2; w /= 2;
    }
  }
  return n;
}

struct LTape { float* z; float* zp; uint8_t* idx; int H, W; };     // pre-activation of a conv; pooled copy + winners after taps

// x [B][3][H][W].  src == nullptr: write the normalised tap features to feat_out ([B][feature_floats]); else loss[b] and
// d_x = d(scale * sum_b loss[b]) / d x.
int run(hedit_lpips* h, const float* x, const float* src, long src_stride, int B, int H0, int W0, float scale, float* feat_out,
        float* loss, float* d_x, void* ws, size_t ws_bytes, hipStream_t st, bool dry, bool want_grad, size_t* peak) {
  PF f{B, st, Arena{}};
  f.ar.dry = dry;
  f.ar.base = reinterpret_cast<char*>(ws);
  f.ar.cap = ws_bytes;
  const bool grad = want_grad;
  const size_t per_img = feature_floats(H0, W0);
  float* a0;
  TRY(palloc(f, &a0, (size_t)B * H0 * W0 * 3));
  if (!dry) TRY(lpips_prep_launch(x, a0, h->shift, h->scale, B, H0, W0, st));
  LTape tp[VGG_N] = {};
  float* headG[5] = {};            // gradient of the objective w.r.t. the tap's feature map F (fp32), kept until the backward reaches it
  const float* in = a0;            // input of the next conv: a0, or a pre-activation whose bias + ReLU are applied in the split pass
  int in_layer = -1;               // the layer whose bias goes with `in`
  int H = H0, W = W0;
  size_t foff = 0;
  for (int l = 0; l < VGG_N; ++l) {
    const int cin = VGG_CH[l][0], cout = VGG_CH[l][1];
    const long M = (long)B * H * W;
    bf16_t* A;
    TRY(op_split(f, in, cin, in_layer < 0 ? P_COPY : P_RELU, nullptr, in_layer < 0 ? nullptr : h->b[in_layer], 0, nullptr, 0, H, W, M, &A));
    if (l == 0) f.ar.free(a0);
    float* z;
    TRY(pgemm(f, A, h->conv[l], false, 1, H, W, M, &z));
    f.ar.free(A);
    tp[l] = LTape{z, nullptr, nullptr, H, W};
    if (!grad && l > 0) f.ar.free(const_cast<float*>(in));      // features only: the consumed input is not kept
    in = z;
    in_layer = l;
    const int t = tap_of(l);
    if (t >= 0) {
      const int HW = H * W;
      float *dpix = nullptr, *dF = nullptr;
      if (src) {
        TRY(palloc(f, &dpix, (size_t)M));
        if (grad) TRY(palloc(f, &dF, (size_t)M * cout));
        if (!dry) {
          TRY(lpips_head_launch(z, h->b[l], src + foff, src_stride, h->lin[t], nullptr, dpix, dF, B, HW, cout, scale / (float)HW, st));
          TRY(lpips_reduce_launch(dpix, loss, B, HW, t == 0 ? 1 : 0, st));
        }
        f.ar.free(dpix);
        headG[t] = dF;
      } else if (!dry) {
        // feat_out is [B][per_img]: the tap's block of image b starts at b * per_img + foff
        for (int b = 0; b < B; ++b)
          TRY(lpips_head_launch(z + (size_t)b * HW * cout, h->b[l], nullptr, 0, h->lin[t], feat_out + (size_t)b * per_img + foff, nullptr,
                                nullptr, 1, HW, cout, 0.f, st));
      }
      foff += (size_t)HW * cout;
      if (t < 4) {
        TRY(palloc(f, &tp[l].zp, (size_t)M / 4 * cout));
        TRY(palloc(f, &tp[l].idx, (size_t)M / 4 * cout));
        if (!dry) TRY(maxpool2_launch(z, tp[l].zp, tp[l].idx, B, H, W, cout, st));
        if (!grad) { f.ar.free(z); f.ar.free(tp[l].idx); }
        in = tp[l].zp;
        H /= 2; W /= 2;
      }
    }
  }
  if (!grad) {
    f.ar.free(tp[VGG_N - 1].z);
    if (peak) *peak = f.ar.peak;
    return HEDIT_OK;
  }
  // ---- backward.  G = gradient w.r.t. the activation relu(z_l + b_l) of layer l
  float* G = headG[4];
  for (int l = VGG_N - 1; l >= 0; --l) {
    const int cin = VGG_CH[l][0], cout = VGG_CH[l][1];
    const LTape& t = tp[l];
    const long M = (long)B * t.H * t.W;
    bf16_t* A;
    TRY(op_split(f, G, cout, P_RELU_GRAD, nullptr, h->b[l], 0, t.z, 0, t.H, t.W, M, &A));      // d z_l = G * [z_l + b_l > 0]
    f.ar.free(G);
    f.ar.free(t.z);
    float* dIn;
    TRY(pgemm(f, A, h->conv[l], true, 1, t.H, t.W, M, &dIn));       // gradient w.r.t. this conv's input activation
    f.ar.free(A);
    if (l == 0) {
      if (!dry) TRY(lpips_prep_bwd_launch(dIn, h->conv[0].rows_b, d_x, h->scale, B, t.H, t.W, st));
      f.ar.free(dIn);
      break;
    }
    const int pt = tap_of(l - 1);
    if (pt >= 0) {
      // the input was pool(relu(z_{l-1} + b)): scatter to the winners and add the tap's own gradient
      const LTape& pl = tp[l - 1];
      float* Gn;
      TRY(palloc(f, &Gn, (size_t)B * pl.H * pl.W * cin));
      if (!dry) TRY(maxpool2_bwd_launch(dIn, pl.idx, headG[pt], Gn, B, pl.H, pl.W, cin, st));
      f.ar.free(dIn);
      f.ar.free(headG[pt]);
      f.ar.free(pl.zp);
      f.ar.free(pl.idx);
      G = Gn;
    } else {
      G = dIn;
    }
  }
  if (peak) *peak = f.ar.peak;
  return HEDIT_OK;
}

}  // namespace

extern "C" {

int hedit_lpips_create(hedit_lpips** out) try {
  ARG_CHECK(out, "null");
  TRY(gemm_prepare());
  hedit_lpips* h = new hedit_lpips();
  for (int l = 0; l < VGG_N; ++l) {
    const std::string pre = "net.slice" + std::to_string(VGG_SLICE[l]) + "." + std::to_string(VGG_IDX[l]);
    h->w[l] = f32conv(h, pre + ".weight", VGG_CH[l][1], VGG_CH[l][0], 3);
    h->b[l] = vec(h, pre + ".bias", VGG_CH[l][1]);
  }
  const int tapc[5] = {64, 128, 256, 512, 512};
  for (int t = 0; t < 5; ++t) h->lin[t] = f32conv(h, "lin" + std::to_string(t) + ".model.1.weight", 1, tapc[t], 1);
  if (h->alloc_failed) {
    hedit_set_error("hipMalloc failed while creating the LPIPS network");
    store_free(h);
    delete h;
    return HEDIT_ERR_HIP;
  }
  *out = h;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

void hedit_lpips_destroy(hedit_lpips* h) try {
  if (!h) return;
  store_free(h);
  delete h;
} catch (...) { (void)hedit_abi_catch(); }

int hedit_lpips_num_params(const hedit_lpips* h) { return h ? (int)h->slots.size() : 0; }
const char* hedit_lpips_param_name(const hedit_lpips* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
} catch (...) { (void)hedit_abi_catch(); return nullptr; }
int hedit_lpips_param_shape(const hedit_lpips* h, int i, int* ndim, int* dims4) try {
  ARG_CHECK(h && ndim && dims4 && i >= 0 && i < (int)h->slots.size(), "param index");
  *ndim = h->slots[i].ndim;
  for (int k = 0; k < 4; ++k) dims4[k] = h->slots[i].dims[k];
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }
int hedit_lpips_load(hedit_lpips* h, const char* name, const float* w, size_t numel, void* stream) try {
  ARG_CHECK(h && name && w, "null");
  h->finalized = false;
  return store_load(h, "LPIPS", name, w, numel, reinterpret_cast<hipStream_t>(stream));
} catch (...) { return hedit_abi_catch(); }
int hedit_lpips_missing(const hedit_lpips* h) { return h ? store_missing(h) : -1; }

int hedit_lpips_finalize(hedit_lpips* h, void* stream) try {
  ARG_CHECK(h, "null");
  if (store_missing(h) != 0) {
    hedit_set_error("LPIPS has " + std::to_string(store_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int l = 0; l < VGG_N; ++l) TRY(make_pconv(h, h->conv[l], h->w[l], nullptr, VGG_CH[l][1], VGG_CH[l][0], 3, 0, 0, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h->alloc_failed) { hedit_set_error("hipMalloc failed while packing the LPIPS weights"); return HEDIT_ERR_HIP; }
  h->finalized = true;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

/* floats per image of the normalised tap features hedit_lpips_source writes (H, W multiples of 16) */
size_t hedit_lpips_feature_floats(int height, int width) { return feature_floats(height, width); }

size_t hedit_lpips_workspace_bytes(hedit_lpips* h, int B, int height, int width) try {
  if (!h || B < 1 || height % 16 || width % 16) return 0;
  size_t peak = 0;
  float dummy = 0.f;
  if (run(h, nullptr, &dummy, 0, B, height, width, 1.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, true, &peak) != HEDIT_OK) return 0;
  return peak + 4096;
} catch (...) { (void)hedit_abi_catch(); return 0; }

/* src fp32 [B][3][H][W] in [-1, 1] -> feats fp32 [B][hedit_lpips_feature_floats(H, W)]: the channel-unit-normalised
 * VGG features of the five taps, what every later hedit_lpips_fwd_bwd call compares against */
int hedit_lpips_source(hedit_lpips* h, const float* src, int B, int height, int width, float* feats, void* workspace,
                       size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && src && feats && workspace && B >= 1 && height % 16 == 0 && width % 16 == 0, "lpips_source args");
  if (!h->finalized) { hedit_set_error("call hedit_lpips_finalize after loading the parameters"); return HEDIT_ERR_STATE; }
  return run(h, src, nullptr, 0, B, height, width, 1.f, feats, nullptr, nullptr, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), false, false, nullptr);
} catch (...) { return hedit_abi_catch(); }

/* loss[b] = LPIPS(x_b, source) and d_x = d(scale * sum_b loss[b]) / d x in ONE call (get_lpips_loss + autograd.grad of
 * h_edit_R.py:124-132; scale = 1 / B reproduces the batch mean).  src_feats from hedit_lpips_source: one image shared by
 * the batch (src_per_image = 0) or one per image. */
int hedit_lpips_fwd_bwd(hedit_lpips* h, const float* x, const float* src_feats, int src_per_image, int B, int height, int width,
                        float scale, float* loss, float* d_x, void* workspace, size_t workspace_bytes, void* stream) try {
  ARG_CHECK(h && x && src_feats && loss && d_x && workspace && B >= 1 && height % 16 == 0 && width % 16 == 0, "lpips_fwd_bwd args");
  if (!h->finalized) { hedit_set_error("call hedit_lpips_finalize after loading the parameters"); return HEDIT_ERR_STATE; }
  return run(h, x, src_feats, src_per_image ? (long)feature_floats(height, width) : 0, B, height, width, scale, nullptr, loss, d_x,
             workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), false, true, nullptr);
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
