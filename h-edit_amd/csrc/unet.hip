// The SD-1.x eps-network as a native executor: owns the packed bf16 weights, plans activation
// memory inside a caller-provided workspace (first-fit arena, stream-ordered reuse) and walks the
// network issuing the HIP kernels of gemm.hip / norm.hip / attn.hip on one stream.  One C call
// per UNet evaluation replaces the ~700 eager launches of the reference stack.
//
// Architecture = diffusers' UNet2DConditionModel as used by SD-1.4/1.5;
// parameter names are the diffusers state_dict keys so real checkpoints load unchanged.
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

struct Slot {
  std::string name;
  int kind;        // 0 fp32 copy, 1 linear -> bf16 (scaled), 2 conv3x3 OIHW -> bf16 [O][9][I],
                   // 3 / 4: FF1 weight / bias with GEGLU (value16|gate16) row interleave,
                   // 5: layer `which` of the block-tail weight stream (ffn.hip), 6: the FF1 bias in that kernel's packed
                   // order, 7: layer `which` of a projection-chain weight stream of `lay` layers (linchain.hip)
  void* dst;
  size_t numel;
  int O, I;
  float scale;
  bool loaded;
  int ndim;
  int dims[4];
  int which = 0, lay = 0;
};

struct Res {
  int cin, cout, temb_off;
  float *n1g, *n1b, *n2g, *n2b, *conv2_b, *sc_b;
  bf16_t *conv1, *conv2, *sc_w;
};

struct Attn {
  int C;
  float *gn_g, *gn_b, *pin_b, *ln1g, *ln1b, *ln2g, *ln2b, *ln3g, *ln3b, *o1_b, *o2_b, *ff1_b, *ff2_b, *pout_b;
  bf16_t *pin, *w_qk, *w_v1, *w_o1, *w_q2, *w_k2, *w_v2, *w_o2, *ff1, *ff2, *pout;
  // C == ffn_fused_channels(): the token-local layers run as three chain kernels (linchain.hip x 2, ffn.hip) on these weight streams
  // (+ the packed FF1 bias); pin / w_qk / w_v1 / w_o1 / w_q2 / w_o2 / ff1 / ff2 / ff1_b / pout are then not materialised
  bf16_t *frs = nullptr, *k1s = nullptr, *ffs = nullptr;
  float* ff1_bp = nullptr;
  int ctx_off = 0;      // first row of this block's attn2.to_k / to_v in the UNet-wide matrices (hedit_unet::wk2_all / wv2_all)
};

struct Block {
  std::vector<Res> res;
  std::vector<Attn> attn;
  bool has_attn = false, has_sampler = false;
  bf16_t* samp_w = nullptr;
  float* samp_b = nullptr;
  int ch = 0;
};

}  // namespace

struct hedit_unet {
  hedit_unet_cfg cfg;
  std::vector<void*> owned;
  std::vector<Slot> slots;
  std::map<std::string, int> index;
  int temb_dim = 0, temb_total = 0;
  // stem / head
  float *conv_in_w = nullptr, *conv_in_b = nullptr, *te1_b = nullptr, *te2_b = nullptr;
  bf16_t *te1_w = nullptr, *te2_w = nullptr, *temb_w_all = nullptr, *conv_out_w = nullptr;
  float *temb_b_all = nullptr, *conv1_b_all = nullptr, *gn_out_g = nullptr, *gn_out_b = nullptr, *conv_out_b = nullptr;
  std::vector<Block> down, up;
  Res mid_res[2];
  Attn mid_attn;
  int32_t* iota = nullptr;
  int iota_cap = 0;
  // sampled launch timing (bench.py roofline): HIP event pairs on the launch stream
  bool prof_on = false;
  std::vector<hipEvent_t> prof_pool;
  struct ProfRec { int kind; double flops, bytes; int e0, e1; int m = 0, n = 0, k = 0, tag = 0; };
  std::vector<ProfRec> prof_recs;
  size_t prof_next = 0;
  // host-language attention controller (hedit_unet_set_attn_hook): when set, every attention layer materialises its
  // probabilities, hands them to the hook and multiplies whatever comes back with V (attn.hip, slow path)
  hedit_attn_hook_fn hook = nullptr;
  void* hook_user = nullptr;
  // attn2.to_k / attn2.to_v of EVERY transformer block, stacked along the output channels: the context projections depend
  // on the prompt only, so one forward computes them for all blocks in two launches ([B*80][ctx_n] and its transpose form)
  // instead of two small launches per block (32 per call on SD-1.x)
  bf16_t *wk2_all = nullptr, *wv2_all = nullptr;
  int ctx_n = 0, ctx_next = 0;
};

namespace {

template <class T>
T* dalloc(hedit_unet* h, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16) != hipSuccess) return nullptr;
  h->owned.push_back(p);
  return reinterpret_cast<T*>(p);
}

void add_slot(hedit_unet* h, const std::string& name, int kind, void* dst, size_t numel, int O = 0, int I = 0, float scale = 1.f) {
  Slot s{name, kind, dst, numel, O, I, scale, false, 1, {(int)numel, 1, 1, 1}};
  if (kind == 1) { s.ndim = 2; s.dims[0] = O; s.dims[1] = I; }
  if (kind == 2) { s.ndim = 4; s.dims[0] = O; s.dims[1] = I; s.dims[2] = 3; s.dims[3] = 3; }
  // 1x1 convolutions keep their 4-D torch shape
  if (kind == 1 && (name.find("proj_in.weight") != std::string::npos || name.find("proj_out.weight") != std::string::npos ||
                    name.find("conv_shortcut.weight") != std::string::npos)) {
    s.ndim = 4; s.dims[2] = 1; s.dims[3] = 1;
  }
  h->index[name] = (int)h->slots.size();
  h->slots.push_back(s);
}
float* f32p(hedit_unet* h, const std::string& name, size_t n, float* dst = nullptr) {
  if (!dst) dst = dalloc<float>(h, n);
  add_slot(h, name, 0, dst, n);
  return dst;
}
bf16_t* linp(hedit_unet* h, const std::string& name, int O, int I, bf16_t* dst = nullptr, float scale = 1.f) {
  if (!dst) dst = dalloc<bf16_t>(h, (size_t)O * I);
  add_slot(h, name, 1, dst, (size_t)O * I, O, I, scale);
  return dst;
}
bf16_t* conv3p(hedit_unet* h, const std::string& name, int O, int I) {
  bf16_t* dst = dalloc<bf16_t>(h, (size_t)O * I * 9);
  add_slot(h, name, 2, dst, (size_t)O * I * 9, O, I);
  return dst;
}

Res make_res(hedit_unet* h, const std::string& pre, int cin, int cout, int& temb_off) {
  Res r{};
  r.cin = cin; r.cout = cout; r.temb_off = temb_off;
  r.n1g = f32p(h, pre + ".norm1.weight", cin);
  r.n1b = f32p(h, pre + ".norm1.bias", cin);
  r.conv1 = conv3p(h, pre + ".conv1.weight", cout, cin);
  f32p(h, pre + ".conv1.bias", cout, h->conv1_b_all + temb_off);
  linp(h, pre + ".time_emb_proj.weight", cout, h->temb_dim, h->temb_w_all + (size_t)temb_off * h->temb_dim);
  f32p(h, pre + ".time_emb_proj.bias", cout, h->temb_b_all + temb_off);
  r.n2g = f32p(h, pre + ".norm2.weight", cout);
  r.n2b = f32p(h, pre + ".norm2.bias", cout);
  r.conv2 = conv3p(h, pre + ".conv2.weight", cout, cout);
  r.conv2_b = f32p(h, pre + ".conv2.bias", cout);
  if (cin != cout) {
    r.sc_w = linp(h, pre + ".conv_shortcut.weight", cout, cin);
    r.sc_b = f32p(h, pre + ".conv_shortcut.bias", cout);
  }
  temb_off += cout;
  return r;
}

Attn make_attn(hedit_unet* h, const std::string& pre, int C) {
  Attn a{};
  a.C = C;
  const int ctx = h->cfg.cross_attention_dim;
  const int d = C / h->cfg.heads;
  const float qscale = 1.0f / sqrtf((float)d) * 1.4426950408889634f;   // softmax scale * log2(e)
  a.gn_g = f32p(h, pre + ".norm.weight", C);
  a.gn_b = f32p(h, pre + ".norm.bias", C);
  // At the level ffn.hip / linchain.hip exist for, the token-local layers of the block run as three kernels -- GroupNorm apply +
  // proj_in + norm1 + q | k | v^T;  attn1.to_out + residual + norm2 + attn2.to_q;  attn2.to_out + residual + norm3 +
  // feed-forward + proj_out + residual -- and their weights go into those kernels' streams (slot kind 5), the FF1 bias
  // into its packed form (kind 6); none of them is kept as a GEMM operand.
  const bool chain = C == ffn_fused_channels();
  auto stream_slot = [&](const std::string& name, bf16_t* stream, int lay, int which, int O, int I, float scale, int ndim) {
    add_slot(h, name, lay ? 7 : 5, stream, (size_t)O * I, O, I, scale);
    Slot& sl = h->slots.back();
    sl.which = which; sl.lay = lay; sl.ndim = ndim;
    sl.dims[0] = O; sl.dims[1] = I; sl.dims[2] = 1; sl.dims[3] = 1;
  };
  const std::string tb = pre + ".transformer_blocks.0";
  if (chain) {
    a.frs = dalloc<bf16_t>(h, lin_chain_stream_bytes(4) / sizeof(bf16_t));
    a.k1s = dalloc<bf16_t>(h, lin_chain_stream_bytes(2) / sizeof(bf16_t));
    a.ffs = dalloc<bf16_t>(h, ffn_stream_bytes(1, 1) / sizeof(bf16_t));
    a.ff1_bp = dalloc<float>(h, ffn_bias_bytes() / sizeof(float));
    stream_slot(pre + ".proj_in.weight", a.frs, 4, 0, C, C, 1.f, 4);
  } else {
    a.pin = linp(h, pre + ".proj_in.weight", C, C);
  }
  a.pin_b = f32p(h, pre + ".proj_in.bias", C);
  a.ln1g = f32p(h, tb + ".norm1.weight", C);
  a.ln1b = f32p(h, tb + ".norm1.bias", C);
  if (chain) {
    stream_slot(tb + ".attn1.to_q.weight", a.frs, 4, 1, C, C, qscale, 2);
    stream_slot(tb + ".attn1.to_k.weight", a.frs, 4, 2, C, C, 1.f, 2);
    stream_slot(tb + ".attn1.to_v.weight", a.frs, 4, 3, C, C, 1.f, 2);
    stream_slot(tb + ".attn1.to_out.0.weight", a.k1s, 2, 0, C, C, 1.f, 2);
  } else {
    a.w_qk = dalloc<bf16_t>(h, (size_t)2 * C * C);
    linp(h, tb + ".attn1.to_q.weight", C, C, a.w_qk, qscale);
    linp(h, tb + ".attn1.to_k.weight", C, C, a.w_qk + (size_t)C * C);
    a.w_v1 = linp(h, tb + ".attn1.to_v.weight", C, C);
    a.w_o1 = linp(h, tb + ".attn1.to_out.0.weight", C, C);
  }
  a.o1_b = f32p(h, tb + ".attn1.to_out.0.bias", C);
  a.ln2g = f32p(h, tb + ".norm2.weight", C);
  a.ln2b = f32p(h, tb + ".norm2.bias", C);
  if (chain) stream_slot(tb + ".attn2.to_q.weight", a.k1s, 2, 1, C, C, qscale, 2);
  else a.w_q2 = linp(h, tb + ".attn2.to_q.weight", C, C, nullptr, qscale);
  a.ctx_off = h->ctx_next;
  h->ctx_next += C;
  a.w_k2 = linp(h, tb + ".attn2.to_k.weight", C, ctx, h->wk2_all + (size_t)a.ctx_off * ctx);
  a.w_v2 = linp(h, tb + ".attn2.to_v.weight", C, ctx, h->wv2_all + (size_t)a.ctx_off * ctx);
  if (chain) {
    stream_slot(tb + ".attn2.to_out.0.weight", a.ffs, 0, 0, C, C, 1.f, 2);
  } else {
    a.w_o2 = linp(h, tb + ".attn2.to_out.0.weight", C, C);
  }
  a.o2_b = f32p(h, tb + ".attn2.to_out.0.bias", C);
  a.ln3g = f32p(h, tb + ".norm3.weight", C);
  a.ln3b = f32p(h, tb + ".norm3.bias", C);
  if (chain) {
    stream_slot(tb + ".ff.net.0.proj.weight", a.ffs, 0, 1, 8 * C, C, 1.f, 2);
    add_slot(h, tb + ".ff.net.0.proj.bias", 6, a.ff1_bp, (size_t)8 * C);
    stream_slot(tb + ".ff.net.2.weight", a.ffs, 0, 2, C, 4 * C, 1.f, 2);
  } else {
    a.ff1 = linp(h, tb + ".ff.net.0.proj.weight", 8 * C, C);
    h->slots.back().kind = 3;
    a.ff1_b = f32p(h, tb + ".ff.net.0.proj.bias", 8 * C);
    h->slots.back().kind = 4;
    a.ff2 = linp(h, tb + ".ff.net.2.weight", C, 4 * C);
  }
  a.ff2_b = f32p(h, tb + ".ff.net.2.bias", C);
  if (chain) {
    stream_slot(pre + ".proj_out.weight", a.ffs, 0, 3, C, C, 1.f, 4);
  } else {
    a.pout = linp(h, pre + ".proj_out.weight", C, C);
  }
  a.pout_b = f32p(h, pre + ".proj_out.bias", C);
  return a;
}

// ------------------------------------------------------------------------------ forward
struct Fwd {
  hedit_unet* h;
  int B;
  hipStream_t st;
  Arena ar;
  const hedit_p2p_plan* plan;
  const float* temb_all;     // fused (time_emb_proj(silu(temb)) + conv1.bias) of every ResBlock
  const bf16_t* ctxb;        // [B*80][ctx_dim]
  const bf16_t* k2_all = nullptr;    // [B*80][ctx_n]: attn2 keys of every block (columns ctx_off .. ctx_off + C)
  const bf16_t* vt2_all = nullptr;   // [ctx_n][B*80]: attn2 values, transposed
  int store_idx = 0;
  int store_self_idx = 0;
  int tblock = 0;            // transformer blocks visited so far in this call (MasaCtrl / PnP layer gates)
  int rblock = 0;            // ResNet blocks visited so far (PnP feature injection)
  int place = 0;             // 0 down, 1 mid, 2 up: where the transformer block being executed sits (the hook's argument)
  int hook_layer = 0;        // attention layers handed to the hook so far in this call
  bool dry() const { return ar.dry; }
};

#define RUN(f, expr)            \
  do {                          \
    if (!(f).dry()) TRY(expr);  \
  } while (0)

// kernel classes of the sampled profiler
enum { PK_CONV = 0, PK_LINEAR = 1, PK_SELF_ATTN = 2, PK_CROSS_ATTN = 3, PK_NORM = 4, PK_OTHER = 5, PK_COUNT = 6 };

struct ProfScope {
  hedit_unet* h;
  hipStream_t st;
  int rec = -1;
  ProfScope(Fwd& f, int kind, double flops, double bytes = 0.0);
  ~ProfScope() {
    if (rec >= 0) (void)hipEventRecord(h->prof_pool[h->prof_recs[rec].e1], st);
  }
};

// bytes = ALGORITHMIC HBM bytes of the launch(es) in the scope: every operand read once, every result written once
ProfScope::ProfScope(Fwd& f, int kind, double flops, double bytes) : h(f.h), st(f.st) {
  if (f.dry() || !h->prof_on || h->prof_next + 2 > h->prof_pool.size()) return;
  hedit_unet::ProfRec r;
  r.kind = kind; r.flops = flops; r.bytes = bytes; r.e0 = (int)h->prof_next; r.e1 = (int)h->prof_next + 1;
  h->prof_next += 2;
  h->prof_recs.push_back(r);
  rec = (int)h->prof_recs.size() - 1;
  (void)hipEventRecord(h->prof_pool[r.e0], st);
}

template <class T>
int aalloc(Fwd& f, T** out, size_t n) {
  *out = reinterpret_cast<T*>(f.ar.alloc(n * sizeof(T)));
  if (!*out) {
    hedit_set_error("workspace too small (need more than " + std::to_string(f.ar.cap) + " bytes)");
    return HEDIT_ERR_ARG;
  }
  return HEDIT_OK;
}

// batch_in: which extent of the GEMM carries the batch (1 = M, 2 = N, 0 = neither).  The K-chunking is taken from
// the per-image extent times a fixed nominal batch, so the summation order of a layer is a property of the
// layer, not of the launch (see gemm_canonical_chunk).
// unique operand bytes + result bytes of one GEMM launch (bf16 operands; conv: the input image once, not nine times)
double gemm_alg_bytes(const Fwd& f, const GemmParams& p) {
  const double a = p.mode == 0 ? (double)p.M * p.K : (double)f.B * p.Hin * p.Win * p.Cin;
  const double c = p.raw_f32 ? 4.0 * p.M * p.N : 2.0 * p.M * (p.geglu ? p.N / 2 : p.N);
  return 2.0 * a + 2.0 * (double)p.N * p.K + c + (p.residual ? 2.0 * p.M * p.N : 0.0);
}

int run_gemm(Fwd& f, GemmParams p, int batch_in = 1) {
  if (!p.raw_f32 && !p.geglu) {
    const int mn = batch_in == 1 ? p.M / f.B * GEMM_NOMINAL_BATCH : p.M;
    const int nn = batch_in == 2 ? p.N / f.B * GEMM_NOMINAL_BATCH : p.N;
    p.chunk_kt = gemm_canonical_chunk(mn, nn, p.K);
  }
  const int splits = gemm_plan_splits(p.M, p.N, p.K, p.chunk_kt);
  float* part = nullptr;
  if (splits > 1) TRY(aalloc(f, &part, (size_t)splits * p.M * p.N));
  {
    ProfScope ps(f, p.mode == 0 ? PK_LINEAR : PK_CONV, 2.0 * p.M * p.N * p.K, gemm_alg_bytes(f, p));
    if (ps.rec >= 0) {
      auto& r = f.h->prof_recs[ps.rec];
      r.m = p.M; r.n = p.N; r.k = p.K;
      r.tag = p.mode | (p.geglu ? 8 : 0) | (p.residual ? 16 : 0) | (splits > 1 ? 32 : 0) | (p.chunk_kt ? 64 : 0);
    }
    RUN(f, gemm_launch(p, splits, part, f.st));
  }
  if (part) f.ar.free(part);
  return HEDIT_OK;
}

int linear(Fwd& f, const bf16_t* A, int M, int K, const bf16_t* W, int N, const float* bias,
           const bf16_t* residual, bf16_t* C, int ldc, int batch_in = 1) {
  GemmParams p{};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = K; p.mode = 0;
  p.bias = bias; p.residual = residual; p.ldr = N; p.C = C; p.ldc = ldc;
  return run_gemm(f, p, batch_in);
}

// C = A W^T + bias (+ residual), then y = LayerNorm(C).  Behind a split-K launch (small batches) the reduce pass
// normalises too; otherwise the LayerNorm is its own launch, as before -- the same bits either way (kernels.h).
int linear_ln(Fwd& f, const bf16_t* A, int M, int K, const bf16_t* W, int N, const float* bias, const bf16_t* residual, bf16_t* C,
              const float* g, const float* b, float eps, bf16_t* y) {
  GemmParams p{};
  p.A = A; p.W = W; p.M = M; p.N = N; p.K = K; p.lda = K; p.mode = 0;
  p.bias = bias; p.residual = residual; p.ldr = N; p.C = C; p.ldc = N;
  int done = 0;
  p.ln_gamma = g; p.ln_beta = b; p.ln_out = y; p.ln_eps = eps; p.ln_done = &done;
  TRY(run_gemm(f, p));
  if (!done) {
    ProfScope ps(f, PK_NORM, 0.0, 4.0 * M * N);
    RUN(f, layernorm_launch(C, y, g, b, (long)M, N, eps, f.st));
  }
  return HEDIT_OK;
}

int conv3x3(Fwd& f, const bf16_t* X, int Hin, int Win, int Cin, const bf16_t* W, int Cout, const float* bias,
            const bf16_t* residual, bf16_t* Y, int mode, int ldy = 0) {
  GemmParams p{};
  p.mode = mode;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin;
  p.Hout = mode == 2 ? Hin / 2 : (mode == 3 ? Hin * 2 : Hin);
  p.Wout = mode == 2 ? Win / 2 : (mode == 3 ? Win * 2 : Win);
  p.A = X; p.W = W; p.M = f.B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin; p.lda = Cin;
  p.bias = bias; p.residual = residual; p.ldr = Cout; p.C = Y; p.ldc = ldy ? ldy : Cout;
  return run_gemm(f, p);
}

int groupnorm(Fwd& f, const bf16_t* x, bf16_t* y, const float* g, const float* b, int HW, int C, float eps, int silu) {
  float* ws;
  TRY(aalloc(f, &ws, groupnorm_ws_bytes(f.B, HW, C) / sizeof(float)));
  {
    ProfScope ps(f, PK_NORM, 0.0, 6.0 * f.B * (double)HW * C);
    RUN(f, groupnorm_launch(x, y, g, b, f.B, HW, C, f.h->cfg.norm_num_groups, eps, silu, ws, f.st));
  }
  f.ar.free(ws);
  return HEDIT_OK;
}

// x [M][cin] -> *out [M][cout] (allocated here; x is NOT freed)
// dst / ldd: write the result into an existing buffer with row stride ldd (the left columns of the next
// skip concatenation) instead of allocating a contiguous [M][cout] one
int resblock(Fwd& f, const Res& r, const bf16_t* x, int H, int W, bf16_t** out, bf16_t* dst = nullptr, int ldd = 0) {
  const size_t M = (size_t)f.B * H * W;
  bf16_t *a1, *h1, *a2, *sc = nullptr, *y;
  TRY(aalloc(f, &a1, M * r.cin));
  TRY(groupnorm(f, x, a1, r.n1g, r.n1b, H * W, r.cin, 1e-5f, 1));
  TRY(aalloc(f, &h1, M * r.cout));
  TRY(conv3x3(f, a1, H, W, r.cin, r.conv1, r.cout, f.temb_all + r.temb_off, nullptr, h1, 1));
  f.ar.free(a1);
  TRY(aalloc(f, &a2, M * r.cout));
  TRY(groupnorm(f, h1, a2, r.n2g, r.n2b, H * W, r.cout, 1e-5f, 1));
  f.ar.free(h1);
  const bf16_t* res = x;
  if (r.sc_w) {
    TRY(aalloc(f, &sc, M * r.cout));
    TRY(linear(f, x, (int)M, r.cin, r.sc_w, r.cout, r.sc_b, nullptr, sc, r.cout));
    res = sc;
  }
  {   // Plug-and-Play feature injection: the target rows' conv2 input becomes the source rows'
    const hedit_p2p_plan* pl = (f.plan && f.plan->mode > 0) ? f.plan : nullptr;
    if (pl && pl->feat_src && f.rblock == pl->feat_resblock) RUN(f, copy_rows_launch(a2, pl->feat_src, f.B, (long)H * W * r.cout, f.st));
    ++f.rblock;
  }
  if (dst) {
    y = dst;
  } else {
    TRY(aalloc(f, &y, M * r.cout));
  }
  TRY(conv3x3(f, a2, H, W, r.cout, r.conv2, r.cout, r.conv2_b, res, y, 1, dst ? ldd : 0));
  f.ar.free(a2);
  if (sc) f.ar.free(sc);
  *out = y;
  return HEDIT_OK;
}

// One attention layer through the host-language controller: probabilities to HBM, the hook (which may rewrite them in
// place, on the launch stream), then probabilities . V.  Reference: ptp_utils.py:98-106.
int hooked_attention(Fwd& f, const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* vt, long ldvt, bf16_t* out, int ldo,
                     int N, int Mk, int kstride, int heads, int d, int is_cross) {
  AttnProbsParams ap{};
  ap.q = q; ap.ldq = ldq; ap.k = k; ap.ldk = ldk; ap.vt = vt; ap.ldvt = ldvt; ap.out = out; ap.ldo = ldo;
  ap.B = f.B; ap.N = N; ap.M = Mk; ap.kstride = kstride; ap.heads = heads; ap.d = d;
  TRY(aalloc(f, &ap.probs, (size_t)f.B * heads * N * Mk));
  const int layer = f.hook_layer++;
  if (!f.dry()) {
    TRY(attn_probs_launch(ap, f.st));
    const int rc = f.h->hook(f.h->hook_user, ap.probs, f.B * heads, N, Mk, is_cross, f.place, layer, f.st);
    if (rc != 0) {
      hedit_set_error("unet: the attention hook failed at layer " + std::to_string(layer) + " (code " + std::to_string(rc) + ")");
      return HEDIT_ERR_ARG;
    }
    TRY(attn_apply_launch(ap, f.st));
  }
  f.ar.free(ap.probs);
  return HEDIT_OK;
}

// x [M][C] -> *out [M][C] (allocated here; x is NOT freed)
int transformer(Fwd& f, const Attn& a, const bf16_t* x, int H, int W, bf16_t** out, bf16_t* dst = nullptr, int ldd = 0) {
  const int C = a.C, N = H * W, B = f.B, heads = f.h->cfg.heads, d = C / heads;
  const int ctx_dim = f.h->cfg.cross_attention_dim;
  const size_t M = (size_t)B * N;
  const hedit_p2p_plan* pl = (f.plan && f.plan->mode > 0) ? f.plan : nullptr;
  bf16_t *xn, *t0, *tn, *qk, *vt, *ao, *t1, *q2, *k2, *vt2, *t2, *gf, *t3, *y;

  const bool chain = a.ffs != nullptr;
  auto chain_prof = [&](ProfScope& ps, int layers, int tag) {
    if (ps.rec >= 0) {
      auto& r = f.h->prof_recs[ps.rec];
      r.m = (int)M; r.n = C; r.k = layers * C; r.tag = tag;
    }
  };
  TRY(aalloc(f, &t0, M * C));
  TRY(aalloc(f, &qk, M * 2 * C));
  TRY(aalloc(f, &vt, M * C));
  if (chain) {
    // GroupNorm statistics, then normalisation + proj_in -> norm1 -> q | k | v^T in ONE kernel (linchain.hip): x is read once,
    // t0 (the residual stream), q, k and v^T are written, nothing else touches HBM
    float* ws;
    const float* ss = nullptr;
    TRY(aalloc(f, &ws, groupnorm_ws_bytes(B, N, C) / sizeof(float)));
    { ProfScope ps(f, PK_NORM, 0.0, 2.0 * M * C); RUN(f, groupnorm_affine_launch(x, a.gn_g, a.gn_b, B, N, C, f.h->cfg.norm_num_groups, 1e-6f, ws, f.st, &ss)); }
    LinChainParams lc{};
    lc.a = x; lc.lda = C; lc.gn_ss = ss; lc.rows_per_image = N; lc.bias_pre = a.pin_b;
    lc.gamma = a.ln1g; lc.beta = a.ln1b; lc.eps = 1e-5f; lc.stream = a.frs;
    lc.out_mid = t0; lc.ldmid = C; lc.out_q = qk; lc.ldq = 2 * C; lc.out_k = qk + C; lc.ldk = 2 * C; lc.out = vt; lc.ldo = (long)M;
    lc.M = (int)M; lc.C = C;
    {
      ProfScope ps(f, PK_LINEAR, 2.0 * M * 4.0 * C * C, 10.0 * M * C + 8.0 * C * C);      // x read; t0, q, k, v^T written; weights once
      chain_prof(ps, 4, 129);                                                               // tag 129: GroupNorm .. q | k | v^T
      RUN(f, lin_chain_launch(lc, f.st));
    }
    f.ar.free(ws);
  } else {
    TRY(aalloc(f, &xn, M * C));
    TRY(groupnorm(f, x, xn, a.gn_g, a.gn_b, N, C, 1e-6f, 0));
    // ---- proj_in, norm1 and the self-attention projections
    TRY(aalloc(f, &tn, M * C));
    TRY(linear_ln(f, xn, (int)M, C, a.pin, C, a.pin_b, nullptr, t0, a.ln1g, a.ln1b, 1e-5f, tn));
    f.ar.free(xn);
    TRY(linear(f, tn, (int)M, C, a.w_qk, 2 * C, nullptr, nullptr, qk, 2 * C));
    TRY(linear(f, a.w_v1, C, C, tn, (int)M, nullptr, nullptr, vt, (int)M, 2));   // V^T = W_v . X^T
    f.ar.free(tn);
  }
  TRY(aalloc(f, &ao, M * C));
  {
    SelfAttnParams sp{};
    sp.q = qk; sp.ldq = 2 * C; sp.k = qk + C; sp.ldk = 2 * C; sp.vt = vt; sp.ldvt = (long)M;
    sp.out = ao; sp.ldo = C; sp.B = B; sp.N = N; sp.heads = heads; sp.d = d;
    sp.qk_src = (pl && pl->qk_src && N <= (pl->qk_max_tokens > 0 ? pl->qk_max_tokens : 1024) && f.tblock >= pl->qk_first_block)
                    ? pl->qk_src : nullptr;
    sp.kv_src = (pl && pl->kv_src && f.tblock >= pl->kv_first_block) ? pl->kv_src : nullptr;
    ++f.tblock;
    if (f.h->hook) {
      TRY(hooked_attention(f, sp.q, sp.ldq, sp.k, sp.ldk, sp.vt, sp.ldvt, ao, C, N, N, N, heads, d, 0));
    } else {
      ProfScope ps(f, PK_SELF_ATTN, 4.0 * B * (double)N * N * C, 8.0 * M * C);      // q, k, v^T read, out written
      RUN(f, self_attn_launch(sp, f.st));
    }
    // the reference's store also keeps the <= 32 x 32 self maps; on request (plan.h_store_self) they are materialised for the
    // conditional rows beside the fused kernel, which never forms them
    if (N <= 1024) {
      if (pl && pl->mode == 2 && pl->h_store_self && f.store_self_idx < pl->n_store_self && !f.h->hook) {
        AttnProbsParams ap{};
        ap.q = sp.q; ap.ldq = sp.ldq; ap.k = sp.k; ap.ldk = sp.ldk; ap.probs = pl->h_store_self[f.store_self_idx];
        ap.B = pl->n_pairs; ap.N = N; ap.M = N; ap.kstride = N; ap.heads = heads; ap.d = d;
        ap.pair_src = pl->pair_src; ap.pair_tar = pl->pair_tar; ap.qk_src = sp.qk_src;
        RUN(f, attn_self_store_launch(ap, f.st));
      }
      f.store_self_idx++;
    }
  }
  f.ar.free(qk);
  f.ar.free(vt);
  TRY(aalloc(f, &t1, M * C));
  TRY(aalloc(f, &q2, M * C));
  if (chain) {
    // attn1.to_out + residual -> norm2 -> attn2.to_q in ONE kernel: t1 (the residual stream) and the query are written
    LinChainParams lc{};
    lc.a = ao; lc.lda = C; lc.r1 = t0; lc.ldr1 = C; lc.bias_pre = a.o1_b;
    lc.gamma = a.ln2g; lc.beta = a.ln2b; lc.eps = 1e-5f; lc.stream = a.k1s;
    lc.out_mid = t1; lc.ldmid = C; lc.out = q2; lc.ldo = C; lc.M = (int)M; lc.C = C;
    ProfScope ps(f, PK_LINEAR, 2.0 * M * 2.0 * C * C, 8.0 * M * C + 4.0 * C * C);          // ao, t0 read; t1, q written
    chain_prof(ps, 2, 130);                                                                 // tag 130: attn1.to_out .. attn2.to_q
    RUN(f, lin_chain_launch(lc, f.st));
  } else {
    // ---- attn1.to_out + residual, norm2; cross-attention (P2P edits + store happen inside the kernel)
    TRY(aalloc(f, &tn, M * C));
    TRY(linear_ln(f, ao, (int)M, C, a.w_o1, C, a.o1_b, t0, t1, a.ln2g, a.ln2b, 1e-5f, tn));
    TRY(linear(f, tn, (int)M, C, a.w_q2, C, nullptr, nullptr, q2, C));
    f.ar.free(tn);
  }
  f.ar.free(ao);
  f.ar.free(t0);
  const int MC = B * HEDIT_CTXP;
  k2 = const_cast<bf16_t*>(f.k2_all) + a.ctx_off;                      // row stride ctx_n
  vt2 = const_cast<bf16_t*>(f.vt2_all) + (size_t)a.ctx_off * MC;
  const int ldk2 = f.h->ctx_n;
  TRY(aalloc(f, &ao, M * C));
  {
    CrossAttnParams cp{};
    cp.q = q2; cp.ldq = C; cp.k = k2; cp.ldk = ldk2; cp.vt = vt2; cp.ldvt = MC; cp.out = ao; cp.ldo = C;
    cp.B = B; cp.N = N; cp.heads = heads; cp.d = d;
    const bool stored_layer = N <= 1024;
    if (pl) {
      cp.n_pairs = pl->n_pairs; cp.pair_src = pl->pair_src; cp.pair_tar = pl->pair_tar;
      cp.mixT = reinterpret_cast<const bf16_t*>(pl->mixT); cp.bvec = pl->bvec;
      cp.singles = pl->singles; cp.n_single = pl->n_single;
      cp.store = (pl->mode == 2 && stored_layer && pl->h_store && f.store_idx < pl->n_store) ? pl->h_store[f.store_idx] : nullptr;
    } else {
      cp.n_pairs = 0; cp.singles = f.h->iota; cp.n_single = B;
    }
    if (stored_layer) f.store_idx++;
    if (f.h->hook) {
      TRY(hooked_attention(f, cp.q, cp.ldq, cp.k, cp.ldk, cp.vt, cp.ldvt, ao, C, N, HEDIT_MAXW, HEDIT_CTXP, heads, d, 1));
    } else {
      ProfScope ps(f, PK_CROSS_ATTN, 4.0 * B * (double)N * HEDIT_MAXW * C, 4.0 * M * C + 4.0 * MC * C);
      RUN(f, cross_attn_launch(cp, f.st));
    }
  }
  f.ar.free(q2);
  if (chain) {
    // attn2.to_out + residual -> LayerNorm -> FF1 -> GEGLU -> FF2 + residual -> proj_out + residual in ONE kernel: rows in
    // registers, weights streamed (ffn.hip).  The result goes where proj_out's would (dst / ldd: the next concatenation).
    if (dst) {
      y = dst;
    } else {
      TRY(aalloc(f, &y, M * C));
    }
    FfnParams fp{};
    fp.a = ao; fp.lda = C; fp.x = t1; fp.ldx = C; fp.r2 = x; fp.ldr2 = C;
    fp.bias_pre = a.o2_b; fp.bias_post = a.pout_b;
    fp.gamma = a.ln3g; fp.beta = a.ln3b; fp.eps = 1e-5f;
    fp.stream = a.ffs; fp.bias1p = a.ff1_bp; fp.bias2 = a.ff2_b; fp.out = y; fp.ldo = dst ? ldd : C; fp.M = (int)M; fp.C = C;
    {
      ProfScope ps(f, PK_LINEAR, 2.0 * M * 14.0 * C * C, 8.0 * M * C + 28.0 * C * C);      // a, t1, x read, out written; weights once
      chain_prof(ps, 14, 128);                                                              // tag 128: the fused block tail
      RUN(f, ffn_fused_launch(fp, f.st));
    }
    f.ar.free(ao);
    f.ar.free(t1);
    *out = y;
    return HEDIT_OK;
  }
  TRY(aalloc(f, &t2, M * C));
  // ---- attn2.to_out + residual, norm3, GEGLU feed-forward
  TRY(aalloc(f, &tn, M * C));
  TRY(linear_ln(f, ao, (int)M, C, a.w_o2, C, a.o2_b, t1, t2, a.ln3g, a.ln3b, 1e-5f, tn));
  f.ar.free(ao);
  f.ar.free(t1);
  TRY(aalloc(f, &gf, M * 4 * C));
  {
    GemmParams gp{};
    gp.A = tn; gp.W = a.ff1; gp.M = (int)M; gp.N = 8 * C; gp.K = C; gp.lda = C; gp.mode = 0;
    gp.bias = a.ff1_b; gp.residual = nullptr; gp.ldr = 0; gp.C = gf; gp.ldc = 4 * C; gp.geglu = 1;
    ProfScope ps(f, PK_LINEAR, 2.0 * gp.M * gp.N * gp.K, gemm_alg_bytes(f, gp));
    RUN(f, gemm_launch(gp, 1, nullptr, f.st));
  }
  f.ar.free(tn);
  TRY(aalloc(f, &t3, M * C));
  TRY(linear(f, gf, (int)M, 4 * C, a.ff2, C, a.ff2_b, t2, t3, C));
  f.ar.free(gf);
  f.ar.free(t2);

  if (dst) {
    y = dst;
  } else {
    TRY(aalloc(f, &y, M * C));
  }
  TRY(linear(f, t3, (int)M, C, a.pout, C, a.pout_b, x, y, dst ? ldd : C));
  f.ar.free(t3);
  *out = y;
  return HEDIT_OK;
}

struct Skip { bf16_t* p; int C; };

int forward_impl(hedit_unet* h, const float* x, float t, const float* ctx, int B, int H0, int W0,
                 const hedit_p2p_plan* plan, float* eps_out, void* ws, size_t ws_bytes, hipStream_t st,
                 bool dry, size_t* peak) {
  const hedit_unet_cfg& c = h->cfg;
  Fwd f;
  f.h = h; f.B = B; f.st = st; f.plan = plan;
  f.ar.base = reinterpret_cast<char*>(ws); f.ar.cap = ws_bytes; f.ar.dry = dry;
  const int ch0 = c.block_out_channels[0];

  if (!dry && h->iota_cap < B) {
    hedit_set_error("batch larger than " + std::to_string(h->iota_cap) + " rows");
    return HEDIT_ERR_ARG;
  }

  // ---- text context -> bf16, padded to 80 rows per item
  bf16_t* ctxb;
  TRY(aalloc(f, &ctxb, (size_t)B * HEDIT_CTXP * c.cross_attention_dim));
  RUN(f, ctx_pad_launch(ctx, ctxb, B, c.cross_attention_dim, st));
  f.ctxb = ctxb;
  {
    // attn2 keys / values of ALL transformer blocks, two launches; they stay live for the whole pass (the blocks read
    // their column / row slice): 2 * B * 80 * ctx_n * 2 bytes of workspace = 4 MB per batch row for SD-1.5 (ctx_n = 12480),
    // 0.48 GB at the bench's 120 rows, accounted for by hedit_unet_workspace_bytes like every other arena allocation
    const int MC = B * HEDIT_CTXP;
    bf16_t *k2a, *vt2a;
    TRY(aalloc(f, &k2a, (size_t)MC * h->ctx_n));
    TRY(aalloc(f, &vt2a, (size_t)MC * h->ctx_n));
    TRY(linear(f, ctxb, MC, c.cross_attention_dim, h->wk2_all, h->ctx_n, nullptr, nullptr, k2a, h->ctx_n));
    TRY(linear(f, h->wv2_all, h->ctx_n, c.cross_attention_dim, ctxb, MC, nullptr, nullptr, vt2a, MC, 2));
    f.k2_all = k2a;
    f.vt2_all = vt2a;
  }

  // ---- time embedding chain (the timestep is shared by the batch, so this is one GEMV chain)
  float *te0, *te1, *te2, *temb_all;
  TRY(aalloc(f, &te0, (size_t)ch0));
  TRY(aalloc(f, &te1, (size_t)h->temb_dim));
  TRY(aalloc(f, &te2, (size_t)h->temb_dim));
  TRY(aalloc(f, &temb_all, (size_t)h->temb_total));
  RUN(f, timestep_embed_launch(t, te0, ch0, st));
  RUN(f, gemv_launch(h->te1_w, te0, h->te1_b, nullptr, te1, h->temb_dim, ch0, 0, st));
  RUN(f, gemv_launch(h->te2_w, te1, h->te2_b, nullptr, te2, h->temb_dim, h->temb_dim, 1, st));
  RUN(f, gemv_launch(h->temb_w_all, te2, h->temb_b_all, h->conv1_b_all, temb_all, h->temb_total, h->temb_dim, 1, st));
  f.temb_all = temb_all;

  // ---- stem
  int H = H0, W = W0;
  bf16_t* cur;
  TRY(aalloc(f, &cur, (size_t)B * H * W * ch0));
  RUN(f, conv_in_launch(x, h->conv_in_w, h->conv_in_b, cur, B, c.in_channels, H, W, ch0, st));
  std::vector<Skip> skips;
  skips.push_back({cur, ch0});

  // ---- down path
  for (int i = 0; i < c.n_levels; ++i) {
    Block& blk = h->down[i];
    for (size_t j = 0; j < blk.res.size(); ++j) {
      bf16_t* y;
      TRY(resblock(f, blk.res[j], cur, H, W, &y));
      // cur stays alive as a skip
      cur = y;
      if (blk.has_attn) {
        bf16_t* z;
        f.place = 0;
        TRY(transformer(f, blk.attn[j], cur, H, W, &z));
        f.ar.free(cur);
        cur = z;
      }
      skips.push_back({cur, blk.ch});
    }
    if (blk.has_sampler) {
      bf16_t* y;
      TRY(aalloc(f, &y, (size_t)B * (H / 2) * (W / 2) * blk.ch));
      TRY(conv3x3(f, cur, H, W, blk.ch, blk.samp_w, blk.ch, blk.samp_b, nullptr, y, 2));
      H /= 2; W /= 2;
      cur = y;
      skips.push_back({cur, blk.ch});
    }
  }

  // ---- mid (cur is the last skip: do not free it here)
  {
    bf16_t *y, *z, *w;
    TRY(resblock(f, h->mid_res[0], cur, H, W, &y));
    f.place = 1;
    TRY(transformer(f, h->mid_attn, y, H, W, &z));
    f.ar.free(y);
    TRY(resblock(f, h->mid_res[1], z, H, W, &w));
    f.ar.free(z);
    cur = w;
  }
  int cur_c = c.block_out_channels[c.n_levels - 1];

  // ---- up path.  Every concatenation [cur | skip] is laid out before its left half is produced: the block that
  // computes `cur` (ResNet conv2, transformer proj_out or the upsampling conv) writes it straight into the left
  // columns of the next concatenation buffer (row stride = its full width), so only the skip half is copied.
  bool in_cat = false;         // cur already sits in `cat_next`
  bf16_t* cat_next = nullptr;
  for (int i = 0; i < c.n_levels; ++i) {
    Block& blk = h->up[i];
    for (size_t j = 0; j < blk.res.size(); ++j) {
      Skip s = skips.back();
      skips.pop_back();
      bf16_t* cat;
      const size_t M = (size_t)B * H * W;
      if (in_cat) {
        cat = cat_next;
        { ProfScope ps(f, PK_OTHER, 0.0); RUN(f, concat_launch(nullptr, cur_c, s.p, s.C, cat, (long)M, st)); }
      } else {
        TRY(aalloc(f, &cat, M * (cur_c + s.C)));
        { ProfScope ps(f, PK_OTHER, 0.0); RUN(f, concat_launch(cur, cur_c, s.p, s.C, cat, (long)M, st)); }
        f.ar.free(cur);
      }
      f.ar.free(s.p);
      // where does this block's output go next?
      const bool last_res = j + 1 == blk.res.size();
      const bool to_cat = !last_res;                      // next consumer is the next ResNet block of this level
      bf16_t* dst = nullptr;
      int ldd = 0;
      if (to_cat) {
        ldd = blk.ch + skips.back().C;
        TRY(aalloc(f, &dst, M * ldd));
      }
      bf16_t* y;
      if (blk.has_attn) {
        TRY(resblock(f, blk.res[j], cat, H, W, &y));
        f.ar.free(cat);
        bf16_t* z;
        f.place = 2;
        TRY(transformer(f, blk.attn[j], y, H, W, &z, dst, ldd));
        f.ar.free(y);
        cur = z;
      } else {
        TRY(resblock(f, blk.res[j], cat, H, W, &y, dst, ldd));
        f.ar.free(cat);
        cur = y;
      }
      cur_c = blk.ch;
      in_cat = to_cat;
      cat_next = dst;
    }
    if (blk.has_sampler) {
      // the upsampled tensor is the left half of the next level's first concatenation
      const size_t M4 = (size_t)B * H * W * 4;
      const int ldd = blk.ch + skips.back().C;
      bf16_t* dst;
      TRY(aalloc(f, &dst, M4 * ldd));
      TRY(conv3x3(f, cur, H, W, blk.ch, blk.samp_w, blk.ch, blk.samp_b, nullptr, dst, 3, ldd));
      f.ar.free(cur);
      H *= 2; W *= 2;
      cur = dst;
      in_cat = true;
      cat_next = dst;
    }
  }

  // ---- head
  {
    bf16_t* a;
    TRY(aalloc(f, &a, (size_t)B * H * W * ch0));
    TRY(groupnorm(f, cur, a, h->gn_out_g, h->gn_out_b, H * W, ch0, 1e-5f, 1));
    if (c.out_channels != 4 || ch0 % 64 != 0) {
      RUN(f, conv_out_launch(a, h->conv_out_w, h->conv_out_b, eps_out, B, H, W, ch0, c.out_channels, st));
    } else {
      // conv_out as an N = 4 MFMA GEMM (fp32 products) + bias / NCHW pass: the one-wave-per-pixel kernel re-reads the
      // activation nine times (0.99 ms at 120 rows)
      float* prod;
      const size_t M = (size_t)B * H * W;
      TRY(aalloc(f, &prod, M * 4));
      GemmParams p{};
      p.mode = 1; p.Hin = H; p.Win = W; p.Cin = ch0; p.Hout = H; p.Wout = W;
      p.A = a; p.W = h->conv_out_w; p.M = (int)M; p.N = 4; p.K = 9 * ch0; p.lda = ch0; p.raw_f32 = prod; p.ldc = 4;
      TRY(run_gemm(f, p));
      RUN(f, rows_to_nchw_launch(prod, h->conv_out_b, eps_out, B, (long)H * W, 4, 4, st));
      f.ar.free(prod);
    }
    f.ar.free(a);
    f.ar.free(cur);
  }
  if (peak) *peak = f.ar.peak;
  return HEDIT_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int hedit_unet_create(const hedit_unet_cfg* cfg, hedit_unet** out) try {
  ARG_CHECK(cfg && out, "null");
  ARG_CHECK(cfg->n_levels >= 2 && cfg->n_levels <= HEDIT_MAX_LEVELS, "n_levels");
  ARG_CHECK(cfg->in_channels <= 8 && cfg->out_channels <= 4, "in/out channels");
  ARG_CHECK(cfg->cross_attention_dim % 64 == 0, "cross_attention_dim % 64");
  for (int i = 0; i < cfg->n_levels; ++i) {
    ARG_CHECK(cfg->block_out_channels[i] % 64 == 0, "block_out_channels % 64");
    ARG_CHECK(cfg->block_out_channels[i] % cfg->norm_num_groups == 0, "channels % groups");
    const int d = cfg->block_out_channels[i] / cfg->heads;
    ARG_CHECK(d == 32 || d == 40 || d == 64 || d == 80 || d == 160, "head dim must be one of 32,40,64,80,160");
  }
  TRY(gemm_prepare());
  hedit_unet* h = new hedit_unet();
  h->cfg = *cfg;
  const int* ch = cfg->block_out_channels;
  const int L = cfg->layers_per_block, n = cfg->n_levels;
  h->temb_dim = ch[0] * 4;

  // channel plan of every ResBlock (needed up-front to size the fused time-embedding projection)
  int total = 0;
  {
    int outc = ch[0];
    for (int i = 0; i < n; ++i) { outc = ch[i]; total += L * outc; }
    total += 2 * ch[n - 1];
    for (int i = 0; i < n; ++i) total += (L + 1) * ch[n - 1 - i];
  }
  h->temb_total = total;
  h->temb_w_all = dalloc<bf16_t>(h, (size_t)total * h->temb_dim);
  h->temb_b_all = dalloc<float>(h, total);
  h->conv1_b_all = dalloc<float>(h, total);

  h->conv_in_w = f32p(h, "conv_in.weight", (size_t)ch[0] * cfg->in_channels * 9);
  {
    Slot& cs = h->slots.back();
    cs.ndim = 4; cs.dims[0] = ch[0]; cs.dims[1] = cfg->in_channels; cs.dims[2] = 3; cs.dims[3] = 3;
  }
  h->conv_in_b = f32p(h, "conv_in.bias", ch[0]);
  h->te1_w = linp(h, "time_embedding.linear_1.weight", h->temb_dim, ch[0]);
  h->te1_b = f32p(h, "time_embedding.linear_1.bias", h->temb_dim);
  h->te2_w = linp(h, "time_embedding.linear_2.weight", h->temb_dim, h->temb_dim);
  h->te2_b = f32p(h, "time_embedding.linear_2.bias", h->temb_dim);

  for (int i = 0; i < n; ++i) {
    if (cfg->down_has_attn[i]) h->ctx_n += L * ch[i];
    if (cfg->up_has_attn[i]) h->ctx_n += (L + 1) * ch[n - 1 - i];
  }
  h->ctx_n += ch[n - 1];
  h->wk2_all = dalloc<bf16_t>(h, (size_t)h->ctx_n * cfg->cross_attention_dim);
  h->wv2_all = dalloc<bf16_t>(h, (size_t)h->ctx_n * cfg->cross_attention_dim);

  int toff = 0;
  int outc = ch[0];
  for (int i = 0; i < n; ++i) {
    const int inc = outc;
    outc = ch[i];
    Block b;
    b.ch = outc;
    b.has_attn = cfg->down_has_attn[i] != 0;
    const std::string pre = "down_blocks." + std::to_string(i);
    for (int j = 0; j < L; ++j) {
      b.res.push_back(make_res(h, pre + ".resnets." + std::to_string(j), j == 0 ? inc : outc, outc, toff));
      if (b.has_attn) b.attn.push_back(make_attn(h, pre + ".attentions." + std::to_string(j), outc));
    }
    if (i != n - 1) {
      b.has_sampler = true;
      b.samp_w = conv3p(h, pre + ".downsamplers.0.conv.weight", outc, outc);
      b.samp_b = f32p(h, pre + ".downsamplers.0.conv.bias", outc);
    }
    h->down.push_back(b);
  }
  h->mid_attn = make_attn(h, "mid_block.attentions.0", ch[n - 1]);
  h->mid_res[0] = make_res(h, "mid_block.resnets.0", ch[n - 1], ch[n - 1], toff);
  h->mid_res[1] = make_res(h, "mid_block.resnets.1", ch[n - 1], ch[n - 1], toff);
  outc = ch[n - 1];
  for (int i = 0; i < n; ++i) {
    const int prev = outc;
    outc = ch[n - 1 - i];
    const int inp = ch[n - 1 - (i + 1 < n ? i + 1 : n - 1)];
    Block b;
    b.ch = outc;
    b.has_attn = cfg->up_has_attn[i] != 0;
    const std::string pre = "up_blocks." + std::to_string(i);
    for (int j = 0; j < L + 1; ++j) {
      const int skip = j == L ? inp : outc;
      const int rin = j == 0 ? prev : outc;
      b.res.push_back(make_res(h, pre + ".resnets." + std::to_string(j), rin + skip, outc, toff));
      if (b.has_attn) b.attn.push_back(make_attn(h, pre + ".attentions." + std::to_string(j), outc));
    }
    if (i != n - 1) {
      b.has_sampler = true;
      b.samp_w = conv3p(h, pre + ".upsamplers.0.conv.weight", outc, outc);
      b.samp_b = f32p(h, pre + ".upsamplers.0.conv.bias", outc);
    }
    h->up.push_back(b);
  }
  h->gn_out_g = f32p(h, "conv_norm_out.weight", ch[0]);
  h->gn_out_b = f32p(h, "conv_norm_out.bias", ch[0]);
  h->conv_out_w = conv3p(h, "conv_out.weight", cfg->out_channels, ch[0]);
  h->conv_out_b = f32p(h, "conv_out.bias", cfg->out_channels);

  {
    // identity row list for "all rows are singles" (controller off), sized once: nothing is allocated or copied
    // synchronously inside hedit_unet_forward
    constexpr int IOTA = 16384;
    std::vector<int32_t> v(IOTA);
    for (int i = 0; i < IOTA; ++i) v[i] = i;
    h->iota = dalloc<int32_t>(h, IOTA);
    if (h->iota && hipMemcpy(h->iota, v.data(), IOTA * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) h->iota = nullptr;
    h->iota_cap = h->iota ? IOTA : 0;
  }
  if (h->ctx_next != h->ctx_n) {      // every block must have taken exactly its slice of the stacked attn2.to_k / to_v matrices
    hedit_set_error("internal: stacked context-projection plan mismatch");
    delete h;
    return HEDIT_ERR_STATE;
  }
  if (toff != total) {
    hedit_set_error("internal: time-embedding plan mismatch");
    delete h;
    return HEDIT_ERR_STATE;
  }
  for (void* p : h->owned)
    if (!p) { hedit_set_error("hipMalloc failed while creating the UNet"); return HEDIT_ERR_HIP; }
  for (auto& s : h->slots)
    if (!s.dst) { hedit_set_error("hipMalloc failed for " + s.name); return HEDIT_ERR_HIP; }
  *out = h;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

void hedit_unet_destroy(hedit_unet* h) try {
  if (!h) return;
  for (void* p : h->owned) (void)hipFree(p);
  for (hipEvent_t e : h->prof_pool) (void)hipEventDestroy(e);
  delete h;
} catch (...) { (void)hedit_abi_catch(); }

int hedit_unet_num_params(const hedit_unet* h) { return h ? (int)h->slots.size() : 0; }
const char* hedit_unet_param_name(const hedit_unet* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i].name.c_str();
} catch (...) { (void)hedit_abi_catch(); return nullptr; }

int hedit_unet_param_shape(const hedit_unet* h, int i, int* ndim, int* dims4) try {
  ARG_CHECK(h && ndim && dims4 && i >= 0 && i < (int)h->slots.size(), "param index");
  *ndim = h->slots[i].ndim;
  for (int k = 0; k < 4; ++k) dims4[k] = h->slots[i].dims[k];
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

/* 1: the executor keeps this parameter as an unscaled bf16 copy (matrices, convolutions), so handing it over rounded to
   bf16 loses nothing -- what the start-up weight broadcast uses to halve its blob; 0: kept in fp32 or scaled while packed */
int hedit_unet_param_bf16_exact(const hedit_unet* h, int i) try {
  if (!h || i < 0 || i >= (int)h->slots.size()) return 0;
  const Slot& s = h->slots[i];
  return (((s.kind == 1 || s.kind == 7) && s.scale == 1.f) || s.kind == 2 || s.kind == 3 || s.kind == 5) ? 1 : 0;
} catch (...) { (void)hedit_abi_catch(); return 0; }

int hedit_unet_load(hedit_unet* h, const char* name, const float* w, size_t numel, void* stream) try {
  ARG_CHECK(h && name && w, "null");
  auto it = h->index.find(name);
  if (it == h->index.end()) {
    hedit_set_error(std::string("unknown parameter: ") + name);
    return HEDIT_ERR_ARG;
  }
  Slot& s = h->slots[it->second];
  if (s.numel != numel) {
    hedit_set_error(std::string("size mismatch for ") + name + ": expected " + std::to_string(s.numel) + ", got " + std::to_string(numel));
    return HEDIT_ERR_ARG;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (s.kind == 0) {
    HIP_TRY(hipMemcpyAsync(s.dst, w, numel * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else if (s.kind == 1) {
    TRY(pack_linear_launch(w, reinterpret_cast<bf16_t*>(s.dst), (long)numel, s.scale, st));
  } else if (s.kind == 3) {
    TRY(pack_geglu_rows_launch(w, reinterpret_cast<bf16_t*>(s.dst), nullptr, s.O, s.I, st));
  } else if (s.kind == 4) {
    TRY(pack_geglu_rows_launch(w, nullptr, reinterpret_cast<float*>(s.dst), (int)numel, 1, st));
  } else if (s.kind == 5) {
    TRY(ffn_pack_launch(w, s.which, 1, 1, reinterpret_cast<bf16_t*>(s.dst), st));
  } else if (s.kind == 6) {
    TRY(ffn_pack_bias_launch(w, reinterpret_cast<float*>(s.dst), st));
  } else if (s.kind == 7) {
    TRY(lin_chain_pack_launch(w, s.which, s.scale, s.lay, reinterpret_cast<bf16_t*>(s.dst), st));
  } else {
    TRY(pack_conv3x3_launch(w, reinterpret_cast<bf16_t*>(s.dst), s.O, s.I, st));
  }
  s.loaded = true;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_unet_missing(const hedit_unet* h) try {
  if (!h) return -1;
  int m = 0;
  for (auto& s : h->slots) m += s.loaded ? 0 : 1;
  return m;
} catch (...) { return hedit_abi_catch(); }

size_t hedit_unet_workspace_bytes(hedit_unet* h, int B, int height, int width) try {
  if (!h) return 0;
  size_t peak = 0;
  if (forward_impl(h, nullptr, 0.f, nullptr, B, height, width, nullptr, nullptr, nullptr, 0, nullptr, true, &peak) != HEDIT_OK) return 0;
  return peak + 4096;
} catch (...) { (void)hedit_abi_catch(); return 0; }

int hedit_unet_forward(hedit_unet* h, const float* x, float t, const float* ctx, int B, int height, int width,
                       const hedit_p2p_plan* plan, float* eps_out, void* workspace, size_t workspace_bytes,
                       void* stream) try {
  ARG_CHECK(h && x && ctx && eps_out && workspace, "null");
  ARG_CHECK(B >= 1, "B");
  const int div = 1 << (h->cfg.n_levels - 1);
  ARG_CHECK(height % div == 0 && width % div == 0, "latent size must be divisible by 2^(levels-1)");
  const int lowest = (height / div) * (width / div);
  ARG_CHECK(lowest % 64 == 0, "lowest-resolution level must have a multiple of 64 tokens");
  if (hedit_unet_missing(h) != 0) {
    hedit_set_error("UNet has " + std::to_string(hedit_unet_missing(h)) + " unloaded parameters");
    return HEDIT_ERR_STATE;
  }
  ARG_CHECK(!(h->hook && plan && plan->mode > 0), "an attention hook excludes the in-kernel edits of a plan (pass plan = NULL)");
  if (plan && plan->mode > 0) {
    ARG_CHECK(plan->n_pairs >= 0 && plan->n_pairs + plan->n_single > 0, "plan rows");
    ARG_CHECK(plan->n_pairs == 0 || (plan->pair_src && plan->pair_tar && plan->mixT && plan->bvec), "plan tables");
  }
  return forward_impl(h, x, t, ctx, B, height, width, plan, eps_out, workspace, workspace_bytes,
                      reinterpret_cast<hipStream_t>(stream), false, nullptr);
} catch (...) { return hedit_abi_catch(); }

int hedit_prof_enable(hedit_unet* h, int on, int max_records) try {
  ARG_CHECK(h, "null");
  if (on && (int)h->prof_pool.size() < 2 * max_records) {
    const size_t want = (size_t)2 * max_records;
    while (h->prof_pool.size() < want) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      h->prof_pool.push_back(e);
    }
  }
  h->prof_on = on != 0;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_prof_reset(hedit_unet* h) try {
  ARG_CHECK(h, "null");
  h->prof_recs.clear();
  h->prof_next = 0;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

/* call after the stream has been synchronised */
int hedit_prof_collect(hedit_unet* h, int kind, double* total_ms, double* total_flops, int64_t* count) try {
  ARG_CHECK(h && total_ms && total_flops && count && kind >= 0 && kind < PK_COUNT, "prof args");
  double ms = 0, fl = 0;
  int64_t n = 0;
  for (auto& r : h->prof_recs) {
    if (r.kind != kind) continue;
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->prof_pool[r.e0], h->prof_pool[r.e1]));
    ms += t;
    fl += r.flops;
    ++n;
  }
  *total_ms = ms; *total_flops = fl; *count = n;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

/* one row per sampled launch, in launch order: kind, ms, flops, bytes, M, N, K, tag (GEMM launches: mode | 8 geglu |
   16 residual | 32 split-K | 64 chunked); call after the stream has been synchronised */
int hedit_prof_records(hedit_unet* h, double* rows, int max_rows, int* n_rows) try {
  ARG_CHECK(h && rows && n_rows && max_rows >= 0, "prof args");
  int n = 0;
  for (auto& r : h->prof_recs) {
    if (n >= max_rows) break;
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->prof_pool[r.e0], h->prof_pool[r.e1]));
    double* o = rows + (size_t)n * 8;
    o[0] = r.kind; o[1] = t; o[2] = r.flops; o[3] = r.bytes; o[4] = r.m; o[5] = r.n; o[6] = r.k; o[7] = r.tag;
    ++n;
  }
  *n_rows = n;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

/* algorithmic HBM bytes (operands read once, results written once) of the sampled launches of a class */
int hedit_prof_collect_bytes(hedit_unet* h, int kind, double* total_bytes) try {
  ARG_CHECK(h && total_bytes && kind >= 0 && kind < PK_COUNT, "prof args");
  double b = 0;
  for (auto& r : h->prof_recs)
    if (r.kind == kind) b += r.bytes;
  *total_bytes = b;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

static void store_layers(const hedit_unet* h, int height, int width, std::vector<std::pair<int, int>>& v) {
  const hedit_unet_cfg& c = h->cfg;
  int H = height, W = width;
  for (int i = 0; i < c.n_levels; ++i) {
    if (c.down_has_attn[i])
      for (int j = 0; j < c.layers_per_block; ++j)
        if (H * W <= 1024) v.push_back({H * W, 0});
    if (i != c.n_levels - 1) { H /= 2; W /= 2; }
  }
  if (H * W <= 1024) v.push_back({H * W, 1});
  for (int i = 0; i < c.n_levels; ++i) {
    if (c.up_has_attn[i])
      for (int j = 0; j < c.layers_per_block + 1; ++j)
        if (H * W <= 1024) v.push_back({H * W, 2});
    if (i != c.n_levels - 1) { H *= 2; W *= 2; }
  }
}

int hedit_unet_set_attn_hook(hedit_unet* h, hedit_attn_hook_fn fn, void* user) try {
  ARG_CHECK(h, "unet handle");
  h->hook = fn;
  h->hook_user = fn ? user : nullptr;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_unet_num_store_layers(const hedit_unet* h, int height, int width) try {
  if (!h) return 0;
  std::vector<std::pair<int, int>> v;
  store_layers(h, height, width, v);
  return (int)v.size();
} catch (...) { return hedit_abi_catch(); }

int hedit_unet_store_layer_info(const hedit_unet* h, int height, int width, int i, int* tokens, int* place) try {
  ARG_CHECK(h && tokens && place, "null");
  std::vector<std::pair<int, int>> v;
  store_layers(h, height, width, v);
  ARG_CHECK(i >= 0 && i < (int)v.size(), "layer index");
  *tokens = v[i].first;
  *place = v[i].second;
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
