// Blocks shared by the convolutional executors (vae.hip: SD image autoencoder; ddpm.hip: pixel-space
// DDPM UNet): parameter store addressed by state_dict names, ResNet block (optionally with a timestep-
// embedding bias), single-head spatial attention done with GEMMs, and their input-gradient passes.
// Header-only, every includer gets its own copy in an anonymous namespace.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/hedit.h"
#include "common.h"
#include "exec.h"
#include "kernels.h"

#define TRY(expr)                        \
  do {                                   \
    int _rc = (expr);                    \
    if (_rc != HEDIT_OK) return _rc;     \
  } while (0)


namespace {

struct VSlot {
  std::string name;
  int kind;        // 0 fp32 copy, 1 linear / 1x1 -> bf16, 2 conv3x3 OIHW -> bf16 [O][9][I]
  void* dst;
  size_t numel;
  int O, I;
  bool loaded;
  int ndim;
  int dims[4];
  // input-gradient twin (decoder only), filled by the same hedit_vae_load call:
  // 0 none, 1 bf16 [I][O], 2 bf16 [I][9][O] taps flipped, 3 fp32 IOHW taps flipped
  int tkind = 0;
  void* tdst = nullptr;
};

struct VRes {
  int cin, cout;
  float *n1g, *n1b, *n2g, *n2b, *c1b, *c2b, *sc_b;
  bf16_t *conv1, *conv2, *sc_w;
  bf16_t *conv1_t = nullptr, *conv2_t = nullptr, *sc_t = nullptr;   // input-gradient weights (decoder)
  int temb_ch = 0;                                                   // > 0: timestep-embedding projection
  bf16_t* temb_w = nullptr;
  float* temb_b = nullptr;
};

struct VAttn {
  int C;
  float *gn_g, *gn_b, *q_b, *k_b, *v_b, *o_b;
  bf16_t *w_q, *w_k, *w_v, *w_o;
  bf16_t *w_q_t = nullptr, *w_k_t = nullptr, *w_v_t = nullptr, *w_o_t = nullptr;
};

// parameters of one network, owned device copies; the public handles derive from this
struct ParamStore {
  bool alloc_failed = false;
  std::vector<void*> owned;
  std::vector<VSlot> slots;
  std::map<std::string, int> index;
};

template <class T>
T* dalloc(ParamStore* h, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16) != hipSuccess) {
    h->alloc_failed = true;
    return nullptr;
  }
  h->owned.push_back(p);
  return reinterpret_cast<T*>(p);
}

void add_slot(ParamStore* h, const std::string& name, int kind, void* dst, size_t numel, int O, int I, int ndim,
              int d0, int d1, int d2, int d3) {
  VSlot s{name, kind, dst, numel, O, I, false, ndim, {d0, d1, d2, d3}};
  h->index[name] = (int)h->slots.size();
  h->slots.push_back(s);
}
float* vec(ParamStore* h, const std::string& name, int n) {
  float* d = dalloc<float>(h, n);
  add_slot(h, name, 0, d, n, 0, 0, 1, n, 1, 1, 1);
  return d;
}
float* f32conv(ParamStore* h, const std::string& name, int O, int I, int k) {   // kept fp32, torch layout
  float* d = dalloc<float>(h, (size_t)O * I * k * k);
  add_slot(h, name, 0, d, (size_t)O * I * k * k, O, I, 4, O, I, k, k);
  return d;
}
bf16_t* lin(ParamStore* h, const std::string& name, int O, int I, bool conv1x1 = false) {
  bf16_t* d = dalloc<bf16_t>(h, (size_t)O * I);
  add_slot(h, name, 1, d, (size_t)O * I, O, I, conv1x1 ? 4 : 2, O, I, 1, 1);
  return d;
}
bf16_t* conv3(ParamStore* h, const std::string& name, int O, int I) {
  // at least 4 rows (zero beyond O): a conv with <= 4 output channels can then run as an N = 4 GEMM
  const int rows = O < 4 ? 4 : O;
  bf16_t* d = dalloc<bf16_t>(h, (size_t)rows * I * 9);
  if (d && rows != O && hipMemset(d, 0, (size_t)rows * I * 9 * sizeof(bf16_t)) != hipSuccess) h->alloc_failed = true;
  add_slot(h, name, 2, d, (size_t)O * I * 9, O, I, 4, O, I, 3, 3);
  return d;
}

// attach an input-gradient twin to the slot just added
template <class T>
T* twin(ParamStore* h, int tkind, size_t n) {
  T* d = dalloc<T>(h, n);
  h->slots.back().tkind = tkind;
  h->slots.back().tdst = d;
  return d;
}

// Parameter naming of the two families: diffusers (AutoencoderKL) and the DDPM code base of the face model
struct BlockNames {
  const char* shortcut;                 // 1x1 shortcut conv of a ResNet block
  const char *a_norm, *a_q, *a_k, *a_v, *a_o;
  bool attn_conv1x1;                    // attention projections stored as 1x1 convs (4-d weights)
};
constexpr BlockNames NAMES_DIFFUSERS{".conv_shortcut", ".group_norm", ".to_q", ".to_k", ".to_v", ".to_out.0", false};
constexpr BlockNames NAMES_DDPM{".nin_shortcut", ".norm", ".q", ".k", ".v", ".proj_out", true};

// temb_ch > 0: the block has `temb_proj` (Linear temb_ch -> cout) whose output is added after conv1
VRes make_res(ParamStore* h, const std::string& pre, int cin, int cout, bool grad = false,
              const BlockNames& nm = NAMES_DIFFUSERS, int temb_ch = 0) {
  VRes r{};
  r.cin = cin; r.cout = cout;
  r.n1g = vec(h, pre + ".norm1.weight", cin);
  r.n1b = vec(h, pre + ".norm1.bias", cin);
  r.conv1 = conv3(h, pre + ".conv1.weight", cout, cin);
  if (grad) r.conv1_t = twin<bf16_t>(h, 2, (size_t)cout * cin * 9);
  r.c1b = vec(h, pre + ".conv1.bias", cout);
  if (temb_ch > 0) {
    r.temb_ch = temb_ch;
    r.temb_w = lin(h, pre + ".temb_proj.weight", cout, temb_ch);
    r.temb_b = vec(h, pre + ".temb_proj.bias", cout);
  }
  r.n2g = vec(h, pre + ".norm2.weight", cout);
  r.n2b = vec(h, pre + ".norm2.bias", cout);
  r.conv2 = conv3(h, pre + ".conv2.weight", cout, cout);
  if (grad) r.conv2_t = twin<bf16_t>(h, 2, (size_t)cout * cout * 9);
  r.c2b = vec(h, pre + ".conv2.bias", cout);
  if (cin != cout) {
    r.sc_w = lin(h, pre + nm.shortcut + ".weight", cout, cin, true);
    if (grad) r.sc_t = twin<bf16_t>(h, 1, (size_t)cout * cin);
    r.sc_b = vec(h, pre + nm.shortcut + ".bias", cout);
  }
  return r;
}

VAttn make_attn(ParamStore* h, const std::string& pre, int C, bool grad = false, const BlockNames& nm = NAMES_DIFFUSERS) {
  VAttn a{};
  a.C = C;
  const bool c4 = nm.attn_conv1x1;
  a.gn_g = vec(h, pre + nm.a_norm + ".weight", C);
  a.gn_b = vec(h, pre + nm.a_norm + ".bias", C);
  a.w_q = lin(h, pre + nm.a_q + ".weight", C, C, c4);
  if (grad) a.w_q_t = twin<bf16_t>(h, 1, (size_t)C * C);
  a.q_b = vec(h, pre + nm.a_q + ".bias", C);
  a.w_k = lin(h, pre + nm.a_k + ".weight", C, C, c4);
  if (grad) a.w_k_t = twin<bf16_t>(h, 1, (size_t)C * C);
  a.k_b = vec(h, pre + nm.a_k + ".bias", C);      // loaded for completeness; softmax-invariant (see vae.hip header)
  a.w_v = lin(h, pre + nm.a_v + ".weight", C, C, c4);
  if (grad) a.w_v_t = twin<bf16_t>(h, 1, (size_t)C * C);
  a.v_b = vec(h, pre + nm.a_v + ".bias", C);
  a.w_o = lin(h, pre + nm.a_o + ".weight", C, C, c4);
  if (grad) a.w_o_t = twin<bf16_t>(h, 1, (size_t)C * C);
  a.o_b = vec(h, pre + nm.a_o + ".bias", C);
  return a;
}

struct VF {
  int groups;      // GroupNorm groups of the network
  int B;
  hipStream_t st;
  Arena ar;
  bool gn_fuse = false;   // GroupNorm statistics from the producing convolution's epilogue where the shape allows (gnstat.h)
  bool dry() const { return ar.dry; }
};

// Producer-side pair statistics (gnstat.h) of a tensor, or of the two halves [a | b] of a skip concatenation; a == nullptr: none
struct GnStat {
  const float* a = nullptr;
  int ca = 0;
  const float* b = nullptr;
  int cb = 0;
};
// does a convolution output [B * HW][C] carry statistics?  A function of the layer's shape only (never of the batch): the
// statistics' summation order, like every other one here, must not depend on what shares the launch
inline bool gn_stats_shape(const VF& f, int HW, int C) {
  return f.gn_fuse && HW >= 1024 && HW % 128 == 0 && C % 128 == 0 && gemm_pick_bn(C) == 128;
}

#define RUN(f, expr)            \
  do {                          \
    if (!(f).dry()) TRY(expr);  \
  } while (0)

template <class T>
int aalloc(VF& f, T** out, size_t n) {
  *out = reinterpret_cast<T*>(f.ar.alloc(n * sizeof(T)));
  if (!*out) {
    hedit_set_error("workspace too small (need more than " + std::to_string(f.ar.cap) + " bytes)");
    return HEDIT_ERR_ARG;
  }
  return HEDIT_OK;
}

// batch_in: which extent carries the batch (1 = M, 2 = N, 0 = neither); the K-chunking comes from the per-image
// extent times a fixed nominal batch (gemm_canonical_chunk), so results do not depend on the batch size
int run_gemm(VF& f, GemmParams p, int batch_in = 1, int batch = 0) {
  if (batch <= 0) batch = f.B;
  if (!p.raw_f32 && batch_in >= 0) {
    const int mn = batch_in == 1 ? p.M / batch * GEMM_NOMINAL_BATCH : p.M;
    const int nn = batch_in == 2 ? p.N / batch * GEMM_NOMINAL_BATCH : p.N;
    p.chunk_kt = gemm_canonical_chunk(mn, nn, p.K);
  }
  const int splits = gemm_plan_splits(p.M, p.N, p.K, p.chunk_kt);
  float* part = nullptr;
  if (splits > 1) TRY(aalloc(f, &part, (size_t)splits * p.M * p.N));
  RUN(f, gemm_launch(p, splits, part, f.st));
  if (part) f.ar.free(part);
  return HEDIT_OK;
}

int linear(VF& f, const bf16_t* A, int M, int K, const bf16_t* W, int N, const float* bias, const bf16_t* residual,
           bf16_t* C, int ldc) {
  GemmParams p{};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = K; p.mode = 0;
  p.bias = bias; p.residual = residual; p.ldr = N; p.C = C; p.ldc = ldc;
  return run_gemm(f, p);
}

// mode 1: stride 1; 2: stride 2 with pad (0,1,0,1); 3: on the 2x nearest-upsampled input
// gn_part: [M / 128][Cout / 2][2] pair statistics of Y (only for shapes gn_stats_shape accepts)
int conv3x3(VF& f, const bf16_t* X, int Hin, int Win, int Cin, const bf16_t* W, int Cout, const float* bias,
            const bf16_t* residual, bf16_t* Y, int mode, int ldy = 0, float* gn_part = nullptr) {
  GemmParams p{};
  p.gn_part = gn_part;
  p.mode = mode;
  p.asym = mode == 2 ? 1 : 0;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = mode == 2 ? Hin / 2 : (mode == 3 ? Hin * 2 : Hin);
  p.Wout = mode == 2 ? Win / 2 : (mode == 3 ? Win * 2 : Win);
  p.A = X; p.W = W; p.M = f.B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin; p.lda = Cin;
  p.bias = bias; p.residual = residual; p.ldr = Cout; p.C = Y; p.ldc = ldy ? ldy : Cout;
  return run_gemm(f, p);
}

// conv3x3 (stride 1, pad 1) with Cout <= 4 output channels, fp32 NCHW result: the [*][4] fp32 products of an N = 4 MFMA
// GEMM (weights [4][9 C], rows beyond Cout zero) + a bias / layout pass; the one-wave-per-pixel kernel remains for
// channel counts that are not a multiple of 64.
int conv_out(VF& f, const bf16_t* x, int H, int W, int C, const bf16_t* w, const float* bias, int Cout, float* y) {
  if (C % 64 != 0) {
    RUN(f, conv_out_launch(x, w, bias, y, f.B, H, W, C, Cout, f.st));
    return HEDIT_OK;
  }
  const size_t M = (size_t)f.B * H * W;
  float* prod;
  TRY(aalloc(f, &prod, M * 4));
  GemmParams p{};
  p.mode = 1; p.Hin = H; p.Win = W; p.Cin = C; p.Hout = H; p.Wout = W;
  p.A = x; p.W = w; p.M = (int)M; p.N = 4; p.K = 9 * C; p.lda = C; p.raw_f32 = prod; p.ldc = 4;
  TRY(run_gemm(f, p));
  RUN(f, rows_to_nchw_launch(prod, bias, y, f.B, (long)H * W, 4, Cout, f.st));
  f.ar.free(prod);
  return HEDIT_OK;
}

// stats: if non-null, *stats receives a kept [B][G][2] (mean, rstd) buffer for the backward pass
// xs: producer-side pair statistics of x, if its producer(s) took them
int groupnorm(VF& f, const bf16_t* x, bf16_t* y, const float* g, const float* b, int HW, int C, int silu,
              float** stats = nullptr, const GnStat* xs = nullptr) {
  float *ws, *sb = nullptr;
  if (xs && xs->a && !stats && xs->ca + xs->cb == C && (xs->cb == 0 || xs->b) && groupnorm_from_parts_supported(HW, C, f.groups)) {
    TRY(aalloc(f, &ws, groupnorm_from_parts_ws_bytes(f.B) / sizeof(float)));
    RUN(f, groupnorm_from_parts_launch(x, y, g, b, f.B, HW, C, f.groups, 1e-6f, silu, xs->a, xs->ca, xs->b, xs->cb, ws, f.st));
    f.ar.free(ws);
    return HEDIT_OK;
  }
  if (stats) {
    TRY(aalloc(f, &sb, (size_t)f.B * 64 * 2));
    *stats = sb;
  }
  TRY(aalloc(f, &ws, groupnorm_ws_bytes(f.B, HW, C) / sizeof(float)));
  RUN(f, groupnorm_launch(x, y, g, b, f.B, HW, C, f.groups, 1e-6f, silu, ws, f.st, sb));
  f.ar.free(ws);
  return HEDIT_OK;
}

// what the backward pass needs from a forward block (all buffers stay allocated in the arena)
struct ResRec {
  const VRes* r;
  const bf16_t* x;
  bf16_t* h1;
  float *st1, *st2;
  int H, W;
};
struct AttnRec {
  const VAttn* a;
  const bf16_t* x;
  bf16_t *xn, *q, *k;
  float* st;
  int H, W;
};

// x [M][cin] -> *out [M][cout] (allocated here; x is NOT freed)
// temb: the network's timestep embedding (fp32 [temb_ch], shared by the batch) for blocks with temb_proj:
// h = conv1(.) + conv1.bias + temb_proj(silu(temb)), folded into the conv's epilogue bias
// dst / ldd: write the result into an existing buffer with row stride ldd (the left columns of the next skip
// concatenation) instead of allocating a contiguous [M][cout] one
// xs: pair statistics of x (GnStat), if any; ys: if non-null, *ys receives the pair statistics of the output (allocated here,
// freed by the caller) or nullptr when the shape carries none
int resblock(VF& f, const VRes& r, const bf16_t* x, int H, int W, bf16_t** out, ResRec* rec = nullptr,
             const float* temb = nullptr, bf16_t* dst = nullptr, int ldd = 0, const GnStat* xs = nullptr, float** ys = nullptr) {
  const size_t M = (size_t)f.B * H * W;
  if (ys) *ys = nullptr;
  const bool st_out = !rec && gn_stats_shape(f, H * W, r.cout);
  float* st1 = nullptr;
  bf16_t *a1, *h1, *a2, *sc = nullptr, *y;
  float* bias1 = r.c1b;
  float* tb = nullptr;
  if (r.temb_ch > 0) {
    if (!temb && !f.dry()) {
      hedit_set_error("resblock: this block needs the timestep embedding");
      return HEDIT_ERR_ARG;
    }
    TRY(aalloc(f, &tb, (size_t)r.cout));
    RUN(f, gemv_launch(r.temb_w, temb, r.temb_b, r.c1b, tb, r.cout, r.temb_ch, 1, f.st));
    bias1 = tb;
  }
  if (rec) {
    // kept buffers first, so the temporaries freed below do not fragment around them
    TRY(aalloc(f, &h1, M * r.cout));
    *rec = ResRec{&r, x, h1, nullptr, nullptr, H, W};
  }
  TRY(aalloc(f, &a1, M * r.cin));
  TRY(groupnorm(f, x, a1, r.n1g, r.n1b, H * W, r.cin, 1, rec ? &rec->st1 : nullptr, rec ? nullptr : xs));
  if (!rec) TRY(aalloc(f, &h1, M * r.cout));
  if (st_out) TRY(aalloc(f, &st1, M / 128 * r.cout));
  TRY(conv3x3(f, a1, H, W, r.cin, r.conv1, r.cout, bias1, nullptr, h1, 1, 0, st1));
  f.ar.free(a1);
  if (tb) f.ar.free(tb);
  TRY(aalloc(f, &a2, M * r.cout));
  {
    const GnStat hs{st1, r.cout, nullptr, 0};
    TRY(groupnorm(f, h1, a2, r.n2g, r.n2b, H * W, r.cout, 1, rec ? &rec->st2 : nullptr, st1 ? &hs : nullptr));
  }
  if (st1) f.ar.free(st1);
  if (!rec) f.ar.free(h1);
  const bf16_t* res = x;
  if (r.sc_w) {
    TRY(aalloc(f, &sc, M * r.cout));
    TRY(linear(f, x, (int)M, r.cin, r.sc_w, r.cout, r.sc_b, nullptr, sc, r.cout));
    res = sc;
  }
  if (dst) {
    y = dst;
  } else {
    TRY(aalloc(f, &y, M * r.cout));
  }
  float* sty = nullptr;
  if (ys && st_out) {
    TRY(aalloc(f, &sty, M / 128 * r.cout));
    *ys = sty;
  }
  TRY(conv3x3(f, a2, H, W, r.cout, r.conv2, r.cout, r.c2b, res, y, 1, dst ? ldd : 0, sty));
  f.ar.free(a2);
  if (sc) f.ar.free(sc);
  *out = y;
  return HEDIT_OK;
}

// single-head attention over the T = H*W tokens of every image; x [B*T][C] -> *out (x is NOT freed)
int attention(VF& f, const VAttn& a, const bf16_t* x, int H, int W, bf16_t** out, AttnRec* rec = nullptr,
              bf16_t* dst = nullptr, int ldd = 0) {
  const int C = a.C, T = H * W, B = f.B;
  const size_t M = (size_t)B * T;
  bf16_t *xn, *q, *k, *vt, *pb, *o, *y;
  float *s, *ob, *st = nullptr;
  TRY(aalloc(f, &xn, M * C));
  TRY(groupnorm(f, x, xn, a.gn_g, a.gn_b, T, C, 0, rec ? &st : nullptr));
  TRY(aalloc(f, &q, M * C));
  TRY(linear(f, xn, (int)M, C, a.w_q, C, a.q_b, nullptr, q, C));
  TRY(aalloc(f, &k, M * C));
  TRY(linear(f, xn, (int)M, C, a.w_k, C, nullptr, nullptr, k, C));
  TRY(aalloc(f, &o, M * C));
  // Images per pass: when the stacked score matrix is small (B*T <= 4096) all images go through ONE
  // Q.K^T GEMM, a block-diagonal softmax (zeros off the diagonal blocks) and ONE P.V GEMM over K = G*T --
  // the off-diagonal products are wasted MFMA work (a few GFLOP) bought back many times over in launches
  // (the pixel UNet has six such layers at 16x16 / 8x8 tokens).  Large T (the VAE's 64x64) goes image by image.
  int G = 4096 / T;
  G = G < 1 ? 1 : (G > B ? B : G);
  TRY(aalloc(f, &vt, (size_t)C * T * G));
  TRY(aalloc(f, &s, (size_t)T * G * T * G));
  TRY(aalloc(f, &pb, (size_t)T * G * T * G));
  for (int b = 0; b < B; b += G) {
    const int g = B - b < G ? B - b : G;
    const int TG = T * g;
    const bf16_t* xb = xn + (size_t)b * T * C;
    {   // V^T [C][TG] = W_v . xn_b^T
      GemmParams p{};
      p.A = a.w_v; p.W = xb; p.M = C; p.N = TG; p.K = C; p.lda = C; p.C = vt; p.ldc = TG;
      TRY(run_gemm(f, p, 2, g));
    }
    {   // S [TG][TG] = q_b . k_b^T in fp32
      GemmParams p{};
      p.A = q + (size_t)b * T * C; p.W = k + (size_t)b * T * C; p.M = TG; p.N = TG; p.K = C; p.lda = C;
      p.raw_f32 = s; p.ldc = TG;
      TRY(run_gemm(f, p));
    }
    if (g == 1) {
      RUN(f, softmax_rows_launch(s, pb, T, T, 1.0f / sqrtf((float)C), f.st));
    } else {
      RUN(f, softmax_blockdiag_launch(s, pb, TG, T, TG, 1.0f / sqrtf((float)C), f.st));
    }
    {   // O_b [TG][C] = P . V
      GemmParams p{};
      p.A = pb; p.W = vt; p.M = TG; p.N = C; p.K = TG; p.lda = TG; p.C = o + (size_t)b * T * C; p.ldc = C;
      // one plain chain over the stacked keys: the other images' probabilities are exact zeros, so an image's
      // result is the chain over its own keys whatever else is stacked beside it
      TRY(run_gemm(f, p, -1));
    }
  }
  f.ar.free(pb); f.ar.free(s); f.ar.free(vt);
  if (rec) {
    *rec = AttnRec{&a, x, xn, q, k, st, H, W};
  } else {
    f.ar.free(k); f.ar.free(q); f.ar.free(xn);
  }
  // output bias with the value bias folded in: o_b' = o_b + W_o . v_b
  TRY(aalloc(f, &ob, (size_t)C));
  RUN(f, gemv_launch(a.w_o, a.v_b, a.o_b, nullptr, ob, C, C, 0, f.st));
  if (dst) {
    y = dst;
  } else {
    TRY(aalloc(f, &y, M * C));
  }
  TRY(linear(f, o, (int)M, C, a.w_o, C, ob, x, y, dst ? ldd : C));
  f.ar.free(ob); f.ar.free(o);
  *out = y;
  return HEDIT_OK;
}

int groupnorm_bwd(VF& f, const bf16_t* x, const bf16_t* dy, const bf16_t* add, bf16_t* dx, const float* g, const float* b,
                  const float* stats, int HW, int C, int silu) {
  float* ws;
  TRY(aalloc(f, &ws, groupnorm_bwd_ws_bytes(f.B, HW, C) / sizeof(float)));
  RUN(f, groupnorm_bwd_launch(x, dy, add, dx, g, b, stats, f.B, HW, C, f.groups, silu, ws, f.st));
  f.ar.free(ws);
  return HEDIT_OK;
}

// dy [M][cout] -> *dx [M][cin] (allocated here; dy is NOT freed)
int resblock_bwd(VF& f, const ResRec& rec, const bf16_t* dy, bf16_t** dx_out) {
  const VRes& r = *rec.r;
  const int H = rec.H, W = rec.W;
  const size_t M = (size_t)f.B * H * W;
  bf16_t *da2, *dh1, *da1, *dsc = nullptr, *dx;
  TRY(aalloc(f, &da2, M * r.cout));
  TRY(conv3x3(f, dy, H, W, r.cout, r.conv2_t, r.cout, nullptr, nullptr, da2, 1));
  TRY(aalloc(f, &dh1, M * r.cout));
  TRY(groupnorm_bwd(f, rec.h1, da2, nullptr, dh1, r.n2g, r.n2b, rec.st2, H * W, r.cout, 1));
  f.ar.free(da2);
  TRY(aalloc(f, &da1, M * r.cin));
  TRY(conv3x3(f, dh1, H, W, r.cout, r.conv1_t, r.cin, nullptr, nullptr, da1, 1));
  f.ar.free(dh1);
  const bf16_t* add = dy;
  if (r.sc_w) {
    TRY(aalloc(f, &dsc, M * r.cin));
    TRY(linear(f, dy, (int)M, r.cout, r.sc_t, r.cin, nullptr, nullptr, dsc, r.cin));
    add = dsc;
  }
  TRY(aalloc(f, &dx, M * r.cin));
  TRY(groupnorm_bwd(f, rec.x, da1, add, dx, r.n1g, r.n1b, rec.st1, H * W, r.cin, 1));
  f.ar.free(da1);
  if (dsc) f.ar.free(dsc);
  *dx_out = dx;
  return HEDIT_OK;
}

// Backward of the single-head attention.  With P = softmax(scale Q K^T), O = P V, Y = O Wo^T + X:
//   dO = dY Wo ; dV = P^T dO ; dP = dO V^T ; dS = scale P (dP - rowsum(dP P)) ; dQ = dS K ; dK = dS^T Q
// P is recomputed per image from the kept q, k.  The dropped key / value biases stay dropped: a
// row-constant in dP cancels inside dS, and rows of dS sum to zero so K's bias cannot reach dQ.
int attention_bwd(VF& f, const AttnRec& rec, const bf16_t* dy, bf16_t** dx_out) {
  const VAttn& a = *rec.a;
  const int C = a.C, T = rec.H * rec.W, B = f.B;
  const size_t M = (size_t)B * T;
  const float scale = 1.0f / sqrtf((float)C);
  bf16_t *dO, *v, *dq, *dk, *dv, *pb, *ds, *tt, *ct, *dxn, *dx;
  float *s, *dp;
  TRY(aalloc(f, &dO, M * C));
  TRY(linear(f, dy, (int)M, C, a.w_o_t, C, nullptr, nullptr, dO, C));
  TRY(aalloc(f, &v, M * C));
  TRY(linear(f, rec.xn, (int)M, C, a.w_v, C, nullptr, nullptr, v, C));
  TRY(aalloc(f, &dq, M * C));
  TRY(aalloc(f, &dk, M * C));
  TRY(aalloc(f, &dv, M * C));
  TRY(aalloc(f, &s, (size_t)T * T));
  TRY(aalloc(f, &dp, (size_t)T * T));
  TRY(aalloc(f, &pb, (size_t)T * T));
  TRY(aalloc(f, &ds, (size_t)T * T));
  TRY(aalloc(f, &tt, (size_t)T * T));   // P^T, then dS^T
  TRY(aalloc(f, &ct, (size_t)C * T));   // dO^T, K^T, Q^T in turn
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * T * C;
    GemmParams p{};
    p.A = rec.q + o; p.W = rec.k + o; p.M = T; p.N = T; p.K = C; p.lda = C; p.raw_f32 = s; p.ldc = T;
    TRY(run_gemm(f, p, 0));
    RUN(f, softmax_rows_launch(s, pb, T, T, scale, f.st));
    p = GemmParams{};   // dP = dO V^T
    p.A = dO + o; p.W = v + o; p.M = T; p.N = T; p.K = C; p.lda = C; p.raw_f32 = dp; p.ldc = T;
    TRY(run_gemm(f, p, 0));
    RUN(f, softmax_bwd_launch(pb, dp, ds, T, T, scale, f.st));
    // dV = P^T dO
    RUN(f, transpose_bf16_launch(pb, tt, T, T, f.st));
    RUN(f, transpose_bf16_launch(dO + o, ct, T, C, f.st));
    p = GemmParams{};
    p.A = tt; p.W = ct; p.M = T; p.N = C; p.K = T; p.lda = T; p.C = dv + o; p.ldc = C;
    TRY(run_gemm(f, p, 0));
    // dQ = dS K
    RUN(f, transpose_bf16_launch(rec.k + o, ct, T, C, f.st));
    p = GemmParams{};
    p.A = ds; p.W = ct; p.M = T; p.N = C; p.K = T; p.lda = T; p.C = dq + o; p.ldc = C;
    TRY(run_gemm(f, p, 0));
    // dK = dS^T Q
    RUN(f, transpose_bf16_launch(ds, tt, T, T, f.st));
    RUN(f, transpose_bf16_launch(rec.q + o, ct, T, C, f.st));
    p = GemmParams{};
    p.A = tt; p.W = ct; p.M = T; p.N = C; p.K = T; p.lda = T; p.C = dk + o; p.ldc = C;
    TRY(run_gemm(f, p, 0));
  }
  f.ar.free(ct); f.ar.free(tt); f.ar.free(ds); f.ar.free(pb); f.ar.free(dp); f.ar.free(s);
  f.ar.free(v); f.ar.free(dO);
  // d(xn) = dQ Wq + dK Wk + dV Wv, accumulated through the GEMM's residual input
  TRY(aalloc(f, &dxn, M * C));
  TRY(linear(f, dq, (int)M, C, a.w_q_t, C, nullptr, nullptr, dxn, C));
  TRY(linear(f, dk, (int)M, C, a.w_k_t, C, nullptr, dxn, dq, C));    // dq's buffer is free again
  TRY(linear(f, dv, (int)M, C, a.w_v_t, C, nullptr, dq, dxn, C));
  f.ar.free(dv); f.ar.free(dk); f.ar.free(dq);
  TRY(aalloc(f, &dx, M * C));
  TRY(groupnorm_bwd(f, rec.x, dxn, dy, dx, a.gn_g, a.gn_b, rec.st, T, C, 0));
  f.ar.free(dxn);
  *dx_out = dx;
  return HEDIT_OK;
}

// ---- generic parameter access behind the C ABI of every executor
inline int store_load(ParamStore* h, const char* what, const char* name, const float* w, size_t numel, hipStream_t st) {
  auto it = h->index.find(name);
  if (it == h->index.end()) {
    hedit_set_error(std::string("unknown ") + what + " parameter: " + name);
    return HEDIT_ERR_ARG;
  }
  VSlot& s = h->slots[it->second];
  if (s.numel != numel) {
    hedit_set_error(std::string("size mismatch for ") + name + ": expected " + std::to_string(s.numel) + ", got " + std::to_string(numel));
    return HEDIT_ERR_ARG;
  }
  if (s.kind == 0) {
    HIP_TRY(hipMemcpyAsync(s.dst, w, numel * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else if (s.kind == 1) {
    TRY(pack_linear_launch(w, reinterpret_cast<bf16_t*>(s.dst), (long)numel, 1.0f, st));
  } else {
    TRY(pack_conv3x3_launch(w, reinterpret_cast<bf16_t*>(s.dst), s.O, s.I, st));
  }
  if (s.tkind == 1) {
    TRY(pack_linear_t_launch(w, reinterpret_cast<bf16_t*>(s.tdst), s.O, s.I, st));
  } else if (s.tkind == 2) {
    TRY(pack_conv3x3_dgrad_launch(w, reinterpret_cast<bf16_t*>(s.tdst), s.O, s.I, st));
  } else if (s.tkind == 3) {
    TRY(flip_oihw_launch(w, reinterpret_cast<float*>(s.tdst), s.O, s.I, s.dims[2], st));
  }
  s.loaded = true;
  return HEDIT_OK;
}
inline int store_missing(const ParamStore* h) {
  int m = 0;
  for (auto& s : h->slots) m += s.loaded ? 0 : 1;
  return m;
}
inline void store_free(ParamStore* h) {
  for (void* p : h->owned) if (p) (void)hipFree(p);
  h->owned.clear();
}

}  // namespace
