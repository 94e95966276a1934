// C-ABI glue: error reporting, sampler-step entry points and the single-kernel entry points the
// parity tests call (see include/hedit.h).
#include "../../include/hedit.h"
#include <exception>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <utility>

#include "common.h"
#include "kernels.h"

static thread_local std::string g_last_error;
void hedit_set_error(const std::string& msg) { g_last_error = msg; }

int hedit_abi_catch() noexcept {
  try {
    try {
      throw;
    } catch (const std::bad_alloc&) {
      g_last_error = "out of host memory inside libhedit_hip";
    } catch (const std::exception& e) {
      g_last_error = std::string("C++ exception inside libhedit_hip: ") + e.what();
    } catch (...) {
      g_last_error = "unknown C++ exception inside libhedit_hip";
    }
  } catch (...) {
    // even the message could not be stored; the code alone reports the failure
  }
  return HEDIT_ERR_STATE;
}

#include <atomic>
static std::atomic<int> g_test_flags{0};
int hedit_test_flags() { return g_test_flags.load(std::memory_order_relaxed); }

namespace {
std::mutex g_dev_mu;
std::set<std::pair<const void*, int>> g_lds_set;      // (kernel, device) pairs whose dynamic-LDS limit was raised
std::map<int, int> g_cus;                              // device -> CU count
}  // namespace

int hedit_dyn_lds(const void* kernel, int bytes) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (g_lds_set.count({kernel, dev})) return HEDIT_OK;
  HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  g_lds_set.insert({kernel, dev});
  return HEDIT_OK;
}

int hedit_cu_count(int* cus) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto it = g_cus.find(dev);
  if (it == g_cus.end()) {
    int n = 0;
    HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    it = g_cus.emplace(dev, n).first;
  }
  *cus = it->second;
  return HEDIT_OK;
}

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline StepCoef to_coef(const hedit_step_coef* c) {
  StepCoef k;
  k.sqrt_ab_t = c->sqrt_ab_t; k.sqrt_1m_ab_t = c->sqrt_1m_ab_t; k.sqrt_ab_prev = c->sqrt_ab_prev;
  k.dir_coef = c->dir_coef; k.noise_coef = c->noise_coef;
  k.w_src = c->w_src; k.w_hat = c->w_hat; k.w_tar = c->w_tar; k.coeff = c->coeff; k.w_rec = c->w_rec;
  return k;
}

extern "C" {

const char* hedit_last_error(void) { return g_last_error.c_str(); }
int hedit_version(void) { return 1; }

int hedit_step_base(const float* eps, const float* xt, const float* z, float* x_prev, int n_img, int elems,
                    int eps_rows_per_img, const hedit_step_coef* c, void* stream) try {
  ARG_CHECK(eps && xt && x_prev && c && n_img > 0 && elems > 0, "step_base args");
  return step_base_launch(eps, xt, z, x_prev, n_img, elems, eps_rows_per_img, to_coef(c), S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_step_invert(const float* e_u, const float* e_c, const float* xt, float* x_prev, float* z_out, int n_img,
                      int elems, const hedit_step_coef* c, void* stream) try {
  ARG_CHECK(e_u && e_c && xt && x_prev && z_out && c && n_img > 0 && elems > 0, "step_invert args");
  return step_invert_launch(e_u, e_c, xt, x_prev, z_out, n_img, elems, to_coef(c), S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_step_update(const float* e_u_src, const float* e_c_src, const float* e_u_tar, const float* e_c_tar,
                      int64_t stride_img, const float* x_k, const float* x_base, float* x_out, int n_img,
                      int elems, int k_gt0, const hedit_step_coef* c, void* stream) try {
  ARG_CHECK(e_u_src && e_c_src && e_u_tar && e_c_tar && x_k && x_base && x_out && c, "step_update args");
  return step_update_launch(e_u_src, e_c_src, e_u_tar, e_c_tar, (long)stride_img, x_k, x_base, x_out, n_img,
                            elems, k_gt0, to_coef(c), S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_step_tweedie(const float* e_u_tar, const float* e_c_tar, int64_t stride_img, const float* x, float* z0,
                       int n_img, int elems, float w_tar, float sqrt_ab, float sqrt_1m_ab, float inv_scale, void* stream) try {
  ARG_CHECK(e_u_tar && e_c_tar && x && z0 && sqrt_ab > 0.f, "step_tweedie args");
  return step_tweedie_launch(e_u_tar, e_c_tar, (long)stride_img, x, z0, n_img, elems, w_tar, sqrt_ab, sqrt_1m_ab, inv_scale,
                             S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_step_style(const float* e_u_src, const float* e_c_src, const float* e_u_tar, const float* e_c_tar,
                     int64_t stride_img, const float* x, const float* g_z, float* x_out, int n_img, int elems, float w_hat,
                     float w_tar, float chain, float weight, void* stream) try {
  ARG_CHECK(e_u_src && e_c_src && e_u_tar && e_c_tar && x && g_z && x_out, "step_style args");
  return step_style_launch(e_u_src, e_c_src, e_u_tar, e_c_tar, (long)stride_img, x, g_z, x_out, n_img, elems, w_hat, w_tar,
                           chain, weight, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_local_blend(float* const* h_maps, int n_maps, int heads, const float* alpha_layers,
                      const int32_t* enabled, float* xt, int n_img, int C, int H, int W, float th, void* stream) try {
  ARG_CHECK(h_maps && alpha_layers && xt, "local_blend args");
  return local_blend_launch(const_cast<const float* const*>(h_maps), n_maps, heads, alpha_layers, nullptr, enabled, xt, n_img, C, H, W, th,
                            0.f, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_local_blend_sub(float* const* h_maps, int n_maps, int heads, const float* alpha_layers, const float* substruct_layers,
                          const int32_t* enabled, float* xt, int n_img, int C, int H, int W, float th, float th_sub, void* stream) try {
  ARG_CHECK(h_maps && alpha_layers && substruct_layers && xt, "local_blend_sub args");
  return local_blend_launch(const_cast<const float* const*>(h_maps), n_maps, heads, alpha_layers, substruct_layers, enabled, xt, n_img, C,
                            H, W, th, th_sub, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_axis_mix(const float* in, float* out, const int32_t* idx, const float* val, int nnz, int64_t outer, int n_in, int n_out, int inner,
                   void* stream) try {
  ARG_CHECK(in && out && idx && val, "axis_mix args");
  return axis_mix_launch(in, out, idx, val, nnz, (long)outer, n_in, n_out, inner, S(stream));
} catch (...) { return hedit_abi_catch(); }

size_t hedit_k_gemm_ws_bytes(int M, int N, int K, int splits) try {
  const int s = gemm_pick_splits(M, N, K, splits);
  return gemm_partial_bytes(M, N, s);
} catch (...) { (void)hedit_abi_catch(); return 0; }

/* the canonical (batch-independent) chunking of a layer and the slab count a launch of the actual shape uses */
int hedit_k_gemm_canonical_chunk(int M_nominal, int N_nominal, int K) { return gemm_canonical_chunk(M_nominal, N_nominal, K); }
int hedit_k_gemm_plan_splits(int M, int N, int K, int chunk_kt) { return gemm_plan_splits(M, N, K, chunk_kt); }

int hedit_k_gemm(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N,
                 int K, int lda, int ldc, int ldr, int mode, int Hin, int Win, int Cin, int Hout, int Wout,
                 int splits, void* partial_ws, void* stream) try {
  ARG_CHECK(A && W && C, "gemm args");
  GemmParams p{};
  p.A = reinterpret_cast<const bf16_t*>(A);
  p.W = reinterpret_cast<const bf16_t*>(W);
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.mode = mode;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout;
  p.bias = bias;
  p.residual = reinterpret_cast<const bf16_t*>(residual);
  p.ldr = ldr;
  p.C = reinterpret_cast<bf16_t*>(C);
  p.ldc = ldc;
  if (splits < 0) {
    // the same chunking as `-splits` slabs, folded in registers by one launch (bit-identical by construction)
    const int kt = K / 64;
    const int s = gemm_pick_splits(M, N, K, -splits);
    p.chunk_kt = (kt + s - 1) / s;
    return gemm_launch(p, 1, nullptr, S(stream));
  }
  const int s = gemm_pick_splits(M, N, K, splits);
  return gemm_launch(p, s, reinterpret_cast<float*>(partial_ws), S(stream));
} catch (...) { return hedit_abi_catch(); }

/* hedit_k_gemm for a convolution (mode 1..3) that also writes the GroupNorm pair statistics of its output (csrc/gnstat.h) */
int hedit_k_conv_gn(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N, int K, int ldc,
                    int ldr, int mode, int Hin, int Win, int Cin, int Hout, int Wout, int splits, void* partial_ws, float* gn_part,
                    void* stream) try {
  ARG_CHECK(A && W && C && gn_part, "conv_gn args");
  GemmParams p{};
  p.A = reinterpret_cast<const bf16_t*>(A);
  p.W = reinterpret_cast<const bf16_t*>(W);
  p.M = M; p.N = N; p.K = K; p.lda = Cin; p.mode = mode;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout;
  p.bias = bias;
  p.residual = reinterpret_cast<const bf16_t*>(residual);
  p.ldr = ldr;
  p.C = reinterpret_cast<bf16_t*>(C);
  p.ldc = ldc;
  p.gn_part = gn_part;
  if (splits < 0) {
    const int kt = K / 64;
    const int s = gemm_pick_splits(M, N, K, -splits);
    p.chunk_kt = (kt + s - 1) / s;
    return gemm_launch(p, 1, nullptr, S(stream));
  }
  const int s = gemm_pick_splits(M, N, K, splits);
  return gemm_launch(p, s, reinterpret_cast<float*>(partial_ws), S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_groupnorm_from_parts(const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                 float eps, int silu, const float* part_a, int ca, const float* part_b, int cb, void* ws,
                                 void* stream) try {
  ARG_CHECK(x && y && gamma && beta && part_a && ws, "groupnorm_from_parts args");
  return groupnorm_from_parts_launch(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), gamma, beta, B, HW, C, G,
                                     eps, silu, part_a, ca, part_b, cb, reinterpret_cast<float*>(ws), S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_pack_geglu(const float* w, const float* bias, void* w_packed_bf16, float* bias_packed, int inner, int K,
                       void* stream) try {
  ARG_CHECK(w && w_packed_bf16, "pack_geglu args");
  int rc = pack_geglu_rows_launch(w, reinterpret_cast<bf16_t*>(w_packed_bf16), nullptr, 2 * inner, K, S(stream));
  if (rc != HEDIT_OK || !bias) return rc;
  ARG_CHECK(bias_packed, "pack_geglu: bias_packed");
  return pack_geglu_rows_launch(bias, nullptr, bias_packed, 2 * inner, 1, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_gemm_geglu(const void* A, const void* w_packed, const float* bias_packed, void* C, int M, int inner, int K,
                       int lda, int ldc, void* stream) try {
  ARG_CHECK(A && w_packed && C, "gemm_geglu args");
  GemmParams p{};
  p.A = reinterpret_cast<const bf16_t*>(A);
  p.W = reinterpret_cast<const bf16_t*>(w_packed);
  p.M = M; p.N = 2 * inner; p.K = K; p.lda = lda; p.mode = 0;
  p.bias = bias_packed;
  p.C = reinterpret_cast<bf16_t*>(C);
  p.ldc = ldc;
  p.geglu = 1;
  return gemm_launch(p, 1, nullptr, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_ffn_channels(void) { return ffn_fused_channels(); }
size_t hedit_k_ffn_stream_bytes(int with_outer_layers) { return ffn_stream_bytes(with_outer_layers, with_outer_layers); }
size_t hedit_k_ffn_bias_bytes(void) { return ffn_bias_bytes(); }

int hedit_k_ffn_pack(const float* w1, const float* b1, const float* w2, const float* w_pre, const float* w_post, void* stream_out,
                     float* bias1_out, void* stream) try {
  ARG_CHECK(w1 && b1 && w2 && stream_out && bias1_out && ((w_pre == nullptr) == (w_post == nullptr)), "ffn_pack args");
  const int outer = w_pre != nullptr;
  bf16_t* so = reinterpret_cast<bf16_t*>(stream_out);
  int rc = ffn_pack_launch(w1, 1, outer, outer, so, S(stream));
  if (rc == HEDIT_OK) rc = ffn_pack_launch(w2, 2, outer, outer, so, S(stream));
  if (rc == HEDIT_OK && outer) rc = ffn_pack_launch(w_pre, 0, 1, 1, so, S(stream));
  if (rc == HEDIT_OK && outer) rc = ffn_pack_launch(w_post, 3, 1, 1, so, S(stream));
  if (rc != HEDIT_OK) return rc;
  return ffn_pack_bias_launch(b1, bias1_out, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_ffn_fused(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* w_stream,
                      const float* bias1_packed, const float* bias2, void* out, int64_t ldo, int M, int C, void* stream) try {
  FfnParams f{};
  f.x = reinterpret_cast<const bf16_t*>(x); f.ldx = (long)ldx; f.gamma = gamma; f.beta = beta; f.eps = eps;
  f.stream = reinterpret_cast<const bf16_t*>(w_stream); f.bias1p = bias1_packed; f.bias2 = bias2;
  f.out = reinterpret_cast<bf16_t*>(out); f.ldo = (long)ldo; f.M = M; f.C = C;
  return ffn_fused_launch(f, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_ffn_chain(const void* a, int64_t lda, const void* t1, int64_t ldt1, const void* x, int64_t ldx, const float* bias_pre,
                      const float* gamma, const float* beta, float eps, const void* w_stream, const float* bias1_packed, const float* bias2,
                      const float* bias_post, void* out, int64_t ldo, int M, int C, void* stream) try {
  ARG_CHECK(a && t1 && x, "ffn_chain args");
  FfnParams f{};
  f.a = reinterpret_cast<const bf16_t*>(a); f.lda = (long)lda;
  f.x = reinterpret_cast<const bf16_t*>(t1); f.ldx = (long)ldt1;
  f.r2 = reinterpret_cast<const bf16_t*>(x); f.ldr2 = (long)ldx;
  f.bias_pre = bias_pre; f.bias_post = bias_post;
  f.gamma = gamma; f.beta = beta; f.eps = eps;
  f.stream = reinterpret_cast<const bf16_t*>(w_stream); f.bias1p = bias1_packed; f.bias2 = bias2;
  f.out = reinterpret_cast<bf16_t*>(out); f.ldo = (long)ldo; f.M = M; f.C = C;
  return ffn_fused_launch(f, S(stream));
} catch (...) { return hedit_abi_catch(); }

size_t hedit_k_lin_chain_stream_bytes(int n_out) { return lin_chain_stream_bytes(n_out == 3 ? 4 : 2); }

int hedit_k_lin_chain_pack(const float* w_pre, const float* w0, const float* w1, const float* w2, float scale0, void* stream_out,
                           void* stream) try {
  ARG_CHECK(w_pre && w0 && stream_out && ((w1 == nullptr) == (w2 == nullptr)), "lin_chain_pack args");
  const int layers = w1 ? 4 : 2;
  bf16_t* so = reinterpret_cast<bf16_t*>(stream_out);
  int rc = lin_chain_pack_launch(w_pre, 0, 1.f, layers, so, S(stream));
  if (rc == HEDIT_OK) rc = lin_chain_pack_launch(w0, 1, scale0, layers, so, S(stream));
  if (rc == HEDIT_OK && w1) rc = lin_chain_pack_launch(w1, 2, 1.f, layers, so, S(stream));
  if (rc == HEDIT_OK && w1) rc = lin_chain_pack_launch(w2, 3, 1.f, layers, so, S(stream));
  return rc;
} catch (...) { return hedit_abi_catch(); }

int hedit_k_lin_chain(const void* a, int64_t lda, const void* r1, int64_t ldr1, const float* gn_ss, int rows_per_image,
                      const float* bias_pre, const float* gamma, const float* beta, float eps, const void* w_stream, void* out_mid,
                      int64_t ldmid, void* out_q, int64_t ldq, void* out_k, int64_t ldk, void* out, int64_t ldo, int M, int C,
                      void* stream) try {
  LinChainParams c{};
  c.a = reinterpret_cast<const bf16_t*>(a); c.lda = (long)lda; c.r1 = reinterpret_cast<const bf16_t*>(r1); c.ldr1 = (long)ldr1;
  c.bias_pre = bias_pre; c.gamma = gamma; c.beta = beta; c.eps = eps; c.stream = reinterpret_cast<const bf16_t*>(w_stream);
  c.out_mid = reinterpret_cast<bf16_t*>(out_mid); c.ldmid = (long)ldmid; c.out = reinterpret_cast<bf16_t*>(out); c.ldo = (long)ldo;
  c.M = M; c.C = C; c.gn_ss = gn_ss; c.rows_per_image = rows_per_image;
  c.out_q = reinterpret_cast<bf16_t*>(out_q); c.ldq = (long)ldq; c.out_k = reinterpret_cast<bf16_t*>(out_k); c.ldk = (long)ldk;
  return lin_chain_launch(c, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_storage_is_f16(void) { return HEDIT_F16; }

int hedit_test_set_flags(int flags) try {
  ARG_CHECK(flags >= 0 && flags <= 15, "hedit_test_set_flags: bit 0 drained ring waits, bit 1 exact self-attention pass, bit 2 pixel-UNet GroupNorm statistics path flipped, bit 3 no persistent linear kernel");
  g_test_flags.store(flags, std::memory_order_relaxed);
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_k_lin_chain_sched(const void* a, int64_t lda, const void* r1, int64_t ldr1, const float* gn_ss, int rows_per_image,
                            const float* bias_pre, const float* gamma, const float* beta, float eps, const void* w_stream, void* out_mid,
                            int64_t ldmid, void* out_q, int64_t ldq, void* out_k, int64_t ldk, void* out, int64_t ldo, int M, int C,
                            int sched, void* stream) try {
  LinChainParams c{};
  c.a = reinterpret_cast<const bf16_t*>(a); c.lda = (long)lda; c.r1 = reinterpret_cast<const bf16_t*>(r1); c.ldr1 = (long)ldr1;
  c.bias_pre = bias_pre; c.gamma = gamma; c.beta = beta; c.eps = eps; c.stream = reinterpret_cast<const bf16_t*>(w_stream);
  c.out_mid = reinterpret_cast<bf16_t*>(out_mid); c.ldmid = (long)ldmid; c.out = reinterpret_cast<bf16_t*>(out); c.ldo = (long)ldo;
  c.M = M; c.C = C; c.gn_ss = gn_ss; c.rows_per_image = rows_per_image;
  c.out_q = reinterpret_cast<bf16_t*>(out_q); c.ldq = (long)ldq; c.out_k = reinterpret_cast<bf16_t*>(out_k); c.ldk = (long)ldk;
  return lin_chain_launch_sched(c, sched, S(stream));
} catch (...) { return hedit_abi_catch(); }

size_t hedit_k_groupnorm_ws_bytes(int B, int HW, int C) { return groupnorm_ws_bytes(B, HW, C); }

int hedit_k_groupnorm_affine(const void* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, void* ws,
                             float* ss_out, void* stream) try {
  ARG_CHECK(x && gamma && beta && ws && ss_out, "groupnorm_affine args");
  const float* ss = nullptr;
  int rc = groupnorm_affine_launch(reinterpret_cast<const bf16_t*>(x), gamma, beta, B, HW, C, G, eps, reinterpret_cast<float*>(ws), S(stream), &ss);
  if (rc != HEDIT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(ss_out, ss, (size_t)B * C * 2 * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
  return HEDIT_OK;
} catch (...) { return hedit_abi_catch(); }

int hedit_k_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int G,
                      float eps, int silu, void* ws, void* stream) try {
  ARG_CHECK(x && y && gamma && beta && ws, "groupnorm args");
  return groupnorm_launch(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), gamma, beta, B, HW, C, G,
                          eps, silu, reinterpret_cast<float*>(ws), S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                      void* stream) try {
  ARG_CHECK(x && y && gamma && beta, "layernorm args");
  return layernorm_launch(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), gamma, beta, (long)rows, C, eps, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_geglu(const void* x, void* y, int64_t rows, int inner, void* stream) try {
  ARG_CHECK(x && y, "geglu args");
  return geglu_launch(reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), (long)rows, inner, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_self_attn(const void* q, int ldq, const void* k, int ldk, const void* vt, int64_t ldvt, void* out, int ldo,
                      int B, int N, int heads, int d, const int32_t* qk_src, const int32_t* kv_src, void* stream) try {
  ARG_CHECK(q && k && vt && out, "self_attn args");
  SelfAttnParams p{};
  p.q = reinterpret_cast<const bf16_t*>(q); p.ldq = ldq;
  p.k = reinterpret_cast<const bf16_t*>(k); p.ldk = ldk;
  p.vt = reinterpret_cast<const bf16_t*>(vt); p.ldvt = (long)ldvt;
  p.out = reinterpret_cast<bf16_t*>(out); p.ldo = ldo;
  p.B = B; p.N = N; p.heads = heads; p.d = d; p.qk_src = qk_src; p.kv_src = kv_src;
  return self_attn_launch(p, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_cross_attn(const void* q, int ldq, const void* k, int ldk, const void* vt, int64_t ldvt, void* out, int ldo,
                       int B, int N, int heads, int d, const hedit_p2p_plan* plan, float* store, void* stream) try {
  ARG_CHECK(q && k && vt && out && plan, "cross_attn args");
  CrossAttnParams p{};
  p.q = reinterpret_cast<const bf16_t*>(q); p.ldq = ldq;
  p.k = reinterpret_cast<const bf16_t*>(k); p.ldk = ldk;
  p.vt = reinterpret_cast<const bf16_t*>(vt); p.ldvt = (long)ldvt;
  p.out = reinterpret_cast<bf16_t*>(out); p.ldo = ldo;
  p.B = B; p.N = N; p.heads = heads; p.d = d;
  p.n_pairs = plan->n_pairs; p.pair_src = plan->pair_src; p.pair_tar = plan->pair_tar;
  p.mixT = reinterpret_cast<const bf16_t*>(plan->mixT); p.bvec = plan->bvec;
  p.singles = plan->singles; p.n_single = plan->n_single;
  p.store = store;
  return cross_attn_launch(p, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_attn_probs(const void* q, int ldq, const void* k, int ldk, float* probs, int B, int N, int M, int kstride,
                       int heads, int d, void* stream) try {
  ARG_CHECK(q && k && probs, "attn_probs args");
  AttnProbsParams p{};
  p.q = reinterpret_cast<const bf16_t*>(q); p.ldq = ldq;
  p.k = reinterpret_cast<const bf16_t*>(k); p.ldk = ldk;
  p.probs = probs; p.B = B; p.N = N; p.M = M; p.kstride = kstride; p.heads = heads; p.d = d;
  return attn_probs_launch(p, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_attn_apply(const float* probs, const void* vt, int64_t ldvt, void* out, int ldo, int B, int N, int M,
                       int kstride, int heads, int d, void* stream) try {
  ARG_CHECK(probs && vt && out, "attn_apply args");
  AttnProbsParams p{};
  p.probs = const_cast<float*>(probs);
  p.vt = reinterpret_cast<const bf16_t*>(vt); p.ldvt = (long)ldvt;
  p.out = reinterpret_cast<bf16_t*>(out); p.ldo = ldo;
  p.B = B; p.N = N; p.M = M; p.kstride = kstride; p.heads = heads; p.d = d;
  return attn_apply_launch(p, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_pack_conv3x3(const float* w, void* out, int O, int I, void* stream) try {
  ARG_CHECK(w && out, "pack args");
  return pack_conv3x3_launch(w, reinterpret_cast<bf16_t*>(out), O, I, S(stream));
} catch (...) { return hedit_abi_catch(); }

int hedit_k_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) try {
  ARG_CHECK(x && y, "cast args");
  return f32_to_bf16_launch(x, reinterpret_cast<bf16_t*>(y), (long)n, S(stream));
} catch (...) { return hedit_abi_catch(); }

}  // extern "C"
