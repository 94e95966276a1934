// C entry points of the xffn experiment (linked only into the side library tools/experiments/build_xffn.sh builds).
#include "../../include/hedit.h"
#include "../../h-edit_amd/csrc/common.h"
#include "../../h-edit_amd/csrc/kernels.h"
#include "xffn_decl.h"
#include "lintile_decl.h"
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
extern "C" {
size_t hedit_k_xffn_stream_bytes(void) { return xffn_stream_bytes(); }
size_t hedit_k_xffn_bias_bytes(void) { return xffn_bias_bytes(); }

int hedit_k_xffn_pack(const float* w1, const float* b1, const float* w2, const float* w_pre, const float* w_post, void* stream_out,
                      float* bias1_out, void* stream) try {
  ARG_CHECK(w1 && b1 && w2 && w_pre && w_post && stream_out && bias1_out, "xffn_pack args");
  bf16_t* so = reinterpret_cast<bf16_t*>(stream_out);
  int rc = xffn_pack_launch(w_pre, 0, so, S(stream));
  if (rc == HEDIT_OK) rc = xffn_pack_launch(w1, 1, so, S(stream));
  if (rc == HEDIT_OK) rc = xffn_pack_launch(w2, 2, so, S(stream));
  if (rc == HEDIT_OK) rc = xffn_pack_launch(w_post, 3, so, S(stream));
  if (rc == HEDIT_OK) rc = xffn_pack_bias_launch(b1, bias1_out, S(stream));
  return rc;
} catch (...) { return hedit_abi_catch(); }

int hedit_k_xffn_chain(const void* a, int64_t lda, const void* t1, int64_t ldt1, const void* x, int64_t ldx, const float* bias_pre,
                       const float* gamma, const float* beta, float eps, const void* w_stream, const float* bias1_packed,
                       const float* bias2, const float* bias_post, void* out, int64_t ldo, int M, int C, void* stream) try {
  FfnParams f{};
  f.a = reinterpret_cast<const bf16_t*>(a); f.lda = (long)lda; f.x = reinterpret_cast<const bf16_t*>(t1); f.ldx = (long)ldt1;
  f.r2 = reinterpret_cast<const bf16_t*>(x); f.ldr2 = (long)ldx; f.bias_pre = bias_pre; f.bias_post = bias_post;
  f.gamma = gamma; f.beta = beta; f.eps = eps; f.stream = reinterpret_cast<const bf16_t*>(w_stream); f.bias1p = bias1_packed; f.bias2 = bias2;
  f.out = reinterpret_cast<bf16_t*>(out); f.ldo = (long)ldo; f.M = M; f.C = C;
  return xffn_launch(f, S(stream));
} catch (...) { return hedit_abi_catch(); }

size_t hedit_k_lin_tile_stream_bytes(int n_out) { return lin_tile_stream_bytes(n_out == 3 ? 4 : 2); }

int hedit_k_lin_tile_pack(const float* w_pre, const float* w0, const float* w1, const float* w2, float scale0, void* stream_out,
                          void* stream) try {
  ARG_CHECK(w_pre && w0 && stream_out && ((w1 == nullptr) == (w2 == nullptr)), "lin_tile_pack args");
  const int layers = w1 ? 4 : 2;
  bf16_t* so = reinterpret_cast<bf16_t*>(stream_out);
  int rc = lin_tile_pack_launch(w_pre, 0, 1.f, layers, so, S(stream));
  if (rc == HEDIT_OK) rc = lin_tile_pack_launch(w0, 1, scale0, layers, so, S(stream));
  if (rc == HEDIT_OK && w1) rc = lin_tile_pack_launch(w1, 2, 1.f, layers, so, S(stream));
  if (rc == HEDIT_OK && w1) rc = lin_tile_pack_launch(w2, 3, 1.f, layers, so, S(stream));
  return rc;
} catch (...) { return hedit_abi_catch(); }

int hedit_k_lin_tile(const void* a, int64_t lda, const void* r1, int64_t ldr1, const float* gn_ss, int rows_per_image,
                     const float* bias_pre, const float* gamma, const float* beta, float eps, const void* w_stream, void* out_mid,
                     int64_t ldmid, void* out_q, int64_t ldq, void* out_k, int64_t ldk, void* out, int64_t ldo, int M, int C,
                     void* stream) try {
  LinChainParams c{};
  c.a = reinterpret_cast<const bf16_t*>(a); c.lda = (long)lda; c.r1 = reinterpret_cast<const bf16_t*>(r1); c.ldr1 = (long)ldr1;
  c.bias_pre = bias_pre; c.gamma = gamma; c.beta = beta; c.eps = eps; c.stream = reinterpret_cast<const bf16_t*>(w_stream);
  c.out_mid = reinterpret_cast<bf16_t*>(out_mid); c.ldmid = (long)ldmid; c.out = reinterpret_cast<bf16_t*>(out); c.ldo = (long)ldo;
  c.M = M; c.C = C; c.gn_ss = gn_ss; c.rows_per_image = rows_per_image;
  c.out_q = reinterpret_cast<bf16_t*>(out_q); c.ldq = (long)ldq; c.out_k = reinterpret_cast<bf16_t*>(out_k); c.ldk = (long)ldk;
  return lin_tile_launch(c, S(stream));
} catch (...) { return hedit_abi_catch(); }

}
