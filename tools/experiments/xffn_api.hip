// C entry points of the xffn experiment (linked only into the side library tools/experiments/build_xffn.sh builds).
#include "../../include/hedit.h"
#include "../../h-edit_amd/csrc/common.h"
#include "../../h-edit_amd/csrc/kernels.h"
#include "xffn_decl.h"
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

}
