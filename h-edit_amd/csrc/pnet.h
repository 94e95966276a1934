// Host-side helpers of the "precise" executors (irse.hip, lpips.hip) on the kernels of pnet.hip: operand geometry of the
// three-term split, packed weights for the forward and the input-gradient GEMM, the fused element-wise -> operand pass and
// the GEMM call (batch-independent summation order, like every GEMM of the library).  Header-only, like blocks.h.
#pragma once
#include "blocks.h"

namespace {

inline int split_cs(int C) { return (3 * C) % 64 == 0 ? C : (3 * C <= 64 ? C : (C + 63) / 64 * 64); }
inline int split_kp(int C) { return (3 * split_cs(C) + 63) / 64 * 64; }

// one convolution / linear layer as a precise GEMM: operands for the forward and for the input gradient
struct PConv {
  int O = 0, I = 0, k = 1;
  bf16_t *wf = nullptr, *wb = nullptr;
  int rows_f = 0, rows_b = 0;     // GEMM N (padded to a multiple of 4)
};

struct PF {
  int B;
  hipStream_t st;
  Arena ar;
  bool dry() const { return ar.dry; }
};

template <class T>
int palloc(PF& f, T** out, size_t n) {
  *out = reinterpret_cast<T*>(f.ar.alloc(n * sizeof(T)));
  if (!*out) {
    hedit_set_error("workspace too small (need more than " + std::to_string(f.ar.cap) + " bytes)");
    return HEDIT_ERR_ARG;
  }
  return HEDIT_OK;
}

int make_pconv(ParamStore* h, PConv& c, const float* w, const float* scale, int O, int I, int k, int perm_hw, int perm_c, hipStream_t st) {
  c.O = O; c.I = I; c.k = k;
  c.rows_f = (O + 3) / 4 * 4;
  c.rows_b = (I + 3) / 4 * 4;
  const size_t nf = (size_t)c.rows_f * k * k * split_kp(I), nb = (size_t)c.rows_b * k * k * split_kp(O);
  if (!c.wf) c.wf = dalloc<bf16_t>(h, nf);
  if (!c.wb) c.wb = dalloc<bf16_t>(h, nb);
  if (!c.wf || !c.wb) { hedit_set_error("hipMalloc failed for a packed weight"); return HEDIT_ERR_HIP; }
  TRY(pack_split3_w_launch(w, scale, c.wf, O, I, k, 0, split_cs(I), split_kp(I), c.rows_f, perm_hw, perm_c, st));
  TRY(pack_split3_w_launch(w, scale, c.wb, O, I, k, 1, split_cs(O), split_kp(O), c.rows_b, perm_hw, perm_c, st));
  return HEDIT_OK;
}

// fp32 [rows_in][C] -> split bf16 operand (allocated here)
int op_split(PF& f, const float* x, int C, int op, const float* p, const float* q, int pq_img, const float* z, int geo, int H, int W,
             long rows_in, bf16_t** out) {
  Split3Params s{};
  s.x = x; s.ldx = C; s.z = z; s.p = p; s.q = q; s.pq_img = pq_img; s.op = op;
  s.Kp = split_kp(C); s.Cs = split_cs(C); s.C = C; s.geo = geo; s.B = f.B; s.H = H; s.W = W;
  const long rows_out = geo == 1 ? rows_in / 4 : (geo == 2 ? rows_in * 4 : rows_in);
  TRY(palloc(f, out, (size_t)rows_out * s.Kp));
  s.out = *out;
  if (!f.dry()) TRY(split3_launch(s, rows_out, f.st));
  return HEDIT_OK;
}

// out fp32 [M][rows] = A . W^T ; mode 0: linear / 1x1 over M rows; 1: 3x3 s1 p1; 2: 3x3 s2 p1 (Hin x Win input)
int pgemm(PF& f, const bf16_t* A, const PConv& c, bool dgrad, int mode, int Hin, int Win, long M, float** out) {
  const int Cin = dgrad ? c.O : c.I;
  const int N = dgrad ? c.rows_b : c.rows_f;
  const int Kp = split_kp(Cin);
  GemmParams p{};
  p.A = A; p.W = dgrad ? c.wb : c.wf; p.M = (int)M; p.N = N; p.lda = Kp; p.mode = mode;
  p.K = (mode == 0 ? 1 : 9) * Kp;
  p.Hin = Hin; p.Win = Win; p.Cin = Kp;
  p.Hout = mode == 2 ? Hin / 2 : Hin; p.Wout = mode == 2 ? Win / 2 : Win;
  p.ldc = N;
  TRY(palloc(f, out, (size_t)M * N));
  p.raw_f32 = *out;
  p.op_bf16 = 1;
  // batch-independent summation order, as everywhere (gemm_canonical_chunk): the batch is in M
  p.chunk_kt = gemm_canonical_chunk((int)(M / f.B) * GEMM_NOMINAL_BATCH, N, p.K);
  const int splits = gemm_plan_splits(p.M, p.N, p.K, p.chunk_kt);
  float* part = nullptr;
  if (splits > 1) TRY(palloc(f, &part, (size_t)splits * M * N));
  if (!f.dry()) TRY(gemm_launch(p, splits, part, f.st));
  if (part) f.ar.free(part);
  return HEDIT_OK;
}


}  // namespace
