// Attention for the SD UNet on MFMA (v_mfma_f32_32x32x16_bf16), with the Prompt-to-Prompt edits
// applied inside the kernels so probabilities are never materialised in HBM.
//
// Common structure (per wave: 32 query rows, workgroup = 4 waves = 128 query rows):
//   * "swapped" QK^T: S^T[kv][q] = mfma(A = K tile, B = Q^T) -> a lane owns ONE query column
//     (q = lane & 31) and 16 of the 32 kv rows of the tile; the other 16 live in lane ^ 32.  Row
//     max / row sum are in-register reductions + one cross-half shuffle.
//   * P feeds the PV MFMA straight from registers: O^T[d][q] = mfma(A = V^T tile, B = P^T).  The
//     B-operand wants 8 consecutive k per lane, a lane holds kv = (r&3) + 8(r>>2) + 4*half; since
//     the contraction order over kv is free, the V^T fragment is simply read with the SAME
//     permutation (two 8-byte LDS reads), so no cross-lane traffic is needed.
//   * V arrives transposed (V^T[h*d + dd][b*N + token]) from a GEMM with swapped operands.
//   * Q is pre-scaled by softmax_scale * log2(e) (folded into W_q), so exp2 is used directly.
//
// self_attn: flash / online softmax over 64-row KV tiles staged in LDS (registers prefetch the
//   next tile).  P2P self-attention replacement (ptp_classes.py:194-200: P_tar <- P_src) needs no
//   probabilities at all: the target row just uses the SOURCE row's Q and K (qk_src[b]).
// cross_attn: 77 (padded 96) keys, whole K/V^T in LDS, exact softmax.  For a (src,tar) pair the
//   wave computes P_src, then P_new = P_src . A + bvec * P_tar with the per-step 96x96 mixing
//   matrix on MFMA (Replace / Refine / Reweight and the cross_replace_alpha blend all fold into
//   (A, bvec), see hedit/p2p/plan.py), accumulates the post-edit maps into the fp32 store when
//   asked, and finishes with P.V for both rows.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int D>
struct HeadCfg {
  static constexpr int DP = (D + 15) / 16 * 16;   // K-dim of QK^T, padded
  static constexpr int DK = DP / 16;
  static constexpr int DT = (D + 31) / 32;        // 32-row tiles of O^T
  static constexpr int KS = DP + 8;               // LDS row stride of K (elements)
  static constexpr int DCH = D / 8;               // 16-byte chunks per K row
};

__device__ __forceinline__ bf16x8 pack_p8(const float* p) {
  union { uint32_t u[4]; bf16x8 v; } x;
  x.u[0] = pack_bf16x2(p[0], p[1]);
  x.u[1] = pack_bf16x2(p[2], p[3]);
  x.u[2] = pack_bf16x2(p[4], p[5]);
  x.u[3] = pack_bf16x2(p[6], p[7]);
  return x.v;
}

// read the V^T-style A-fragment: row `row`, kv columns {c0..c0+3} and {c0+8..c0+11}
__device__ __forceinline__ bf16x8 read_perm_frag(const bf16_t* base, int row, int stride, int c0) {
  union { uint2 h[2]; bf16x8 v; } x;
  const bf16_t* p = base + row * stride + c0;
  x.h[0] = *reinterpret_cast<const uint2*>(p);
  x.h[1] = *reinterpret_cast<const uint2*>(p + 8);
  return x.v;
}

// ============================================================================ self-attention
template <int D>
__global__ __launch_bounds__(256, (D <= 80 ? 2 : 1)) void self_attn_kernel(SelfAttnParams p) {
  using H = HeadCfg<D>;
  constexpr int VS = 68;
  __shared__ __attribute__((aligned(16))) bf16_t sK[64 * H::KS];
  __shared__ __attribute__((aligned(16))) bf16_t sV[H::DT * 32 * VS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bqk = p.qk_src ? p.qk_src[b] : b;
  const int q_row = blockIdx.x * 128 + wave * 32 + ql;
  const bool q_ok = q_row < p.N;

  // zero LDS once: pad columns of K (>= D) and pad rows of V^T (>= D) must be finite zeros
  for (int i = tid; i < 64 * H::KS / 8; i += 256) reinterpret_cast<uint4*>(sK)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < H::DT * 32 * VS / 4; i += 256) reinterpret_cast<uint2*>(sV)[i] = make_uint2(0, 0);

  // Q fragments (B operand): lane holds Q[q][ks*16 + hi*8 .. +8]
  bf16x8 qf[H::DK];
  {
    const bf16_t* qp = p.q + ((long)bqk * p.N + (q_ok ? q_row : 0)) * p.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < H::DK; ++ks) {
      const int c = ks * 16 + hi * 8;
      union { uint4 u; bf16x8 v; } x;
      x.u = make_uint4(0, 0, 0, 0);
      if (q_ok && c < D) x.u = *reinterpret_cast<const uint4*>(qp + c);
      qf[ks] = x.v;
    }
  }

  constexpr int K_IT = (64 * H::DCH + 255) / 256;
  constexpr int V_IT = (D * 8 + 255) / 256;
  uint4 kreg[K_IT], vreg[V_IT];
  // per-thread staging coordinates, fixed for the whole KV loop
  const bf16_t* kptr[K_IT];
  const bf16_t* vptr[V_IT];
  int klds[K_IT], vlds[V_IT];
  bool kok[K_IT], vok[V_IT];
  {
    const bf16_t* kbase = p.k + (long)bqk * p.N * p.ldk + h * D;
    const bf16_t* vbase = p.vt + (long)h * D * p.ldvt + (long)b * p.N;
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / H::DCH, c = idx - row * H::DCH;
      kok[i] = idx < 64 * H::DCH;
      kptr[i] = kbase + (long)row * p.ldk + c * 8;
      klds[i] = row * H::KS + c * 8;
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, c = idx & 7;
      vok[i] = idx < D * 8;
      vptr[i] = vbase + (long)row * p.ldvt + c * 8;
      vlds[i] = row * VS + c * 8;
    }
  }
  const long kstep = 64L * p.ldk;

  auto load_regs = [&](int t) {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kok[i]) v = *reinterpret_cast<const uint4*>(kptr[i] + t * kstep);
      kreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (vok[i]) v = *reinterpret_cast<const uint4*>(vptr[i] + t * 64);
      vreg[i] = v;
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < K_IT; ++i)
      if (kok[i]) *reinterpret_cast<uint4*>(sK + klds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < V_IT; ++i)
      if (vok[i]) {
        uint2* dst = reinterpret_cast<uint2*>(sV + vlds[i]);
        dst[0] = make_uint2(vreg[i].x, vreg[i].y);
        dst[1] = make_uint2(vreg[i].z, vreg[i].w);
      }
  };

  f32x16 o[H::DT];
#pragma unroll
  for (int t = 0; t < H::DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = p.N / 64;
  __syncthreads();          // zero-fill done
  load_regs(0);
  store_lds();
  __syncthreads();

  // Lazy rescale: the running maximum (log2 domain) is only raised -- and O, l rescaled -- when
  // some row's tile maximum exceeds it by more than RESCALE_THR; until then probabilities are
  // formed against the stale maximum and are bounded by 2^RESCALE_THR (bf16 has fp32's exponent
  // range, fp32 accumulators: no precision is lost).  Saves an O-wide VALU pass on most tiles.
  constexpr float RESCALE_THR = 10.0f;

  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) load_regs(t + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < H::DK; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (sub * 32 + ql) * H::KS + ks * 16 + hi * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (!__all(mx - m_run <= RESCALE_THR)) {
        // every P.V accumulated so far is complete at this point, so O and l are the only state
        // still expressed against the old maximum
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < H::DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      float pr[16];
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pr[r] = fast_exp2(s[r] - m_run); ls += pr[r]; }
      l_run += ls;
      const bf16x8 pf0 = pack_p8(pr), pf1 = pack_p8(pr + 8);
#pragma unroll
      for (int dt = 0; dt < H::DT; ++dt) {
        const bf16x8 v0 = read_perm_frag(sV, dt * 32 + ql, VS, sub * 32 + 4 * hi);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf0, o[dt], 0, 0, 0);
        const bf16x8 v1 = read_perm_frag(sV, dt * 32 + ql, VS, sub * 32 + 16 + 4 * hi);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf1, o[dt], 0, 0, 0);
      }
    }
    __syncthreads();
    if (t + 1 < ntiles) store_lds();
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    bf16_t* op = p.out + ((long)b * p.N + q_row) * p.ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < H::DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * hi;
        if (d0 < D) {
          uint2 w;
          w.x = pack_bf16x2(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
          w.y = pack_bf16x2(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
          *reinterpret_cast<uint2*>(op + d0) = w;
        }
      }
  }
}

// ============================================================================ cross-attention
constexpr int CTXP = HEDIT_CTXP;
constexpr int WS = 100;    // LDS row stride (elements) of the 96-wide V^T and mix tiles

template <int D>
struct CrossSmem {
  using H = HeadCfg<D>;
  static constexpr int K_BYTES = 96 * H::KS * 2;
  static constexpr int V_BYTES = H::DT * 32 * WS * 2;
  static constexpr int MIX_BYTES = 96 * WS * 2;
  static constexpr int BV_BYTES = 96 * 4;
  static constexpr int TOTAL = K_BYTES + V_BYTES + MIX_BYTES + BV_BYTES;
};

template <int D>
__global__ __launch_bounds__(256, (D <= 80 ? 2 : 1)) void cross_attn_kernel(CrossAttnParams p) {
  using H = HeadCfg<D>;
  using S = CrossSmem<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem);
  bf16_t* sV = reinterpret_cast<bf16_t*>(smem + S::K_BYTES);
  bf16_t* sM = reinterpret_cast<bf16_t*>(smem + S::K_BYTES + S::V_BYTES);
  float* sB = reinterpret_cast<float*>(smem + S::K_BYTES + S::V_BYTES + S::MIX_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y;
  const int item = blockIdx.z;
  const bool is_pair = item < p.n_pairs;
  const int b0 = is_pair ? p.pair_src[item] : p.singles[item - p.n_pairs];
  const int b1 = is_pair ? p.pair_tar[item] : -1;
  const int q_row = blockIdx.x * 128 + wave * 32 + ql;
  const bool q_ok = q_row < p.N;

  for (int i = tid; i < (S::K_BYTES + S::V_BYTES) / 8; i += 256) reinterpret_cast<uint2*>(smem)[i] = make_uint2(0, 0);
  if (is_pair) {
    // mixing matrix (already transposed: mixT[n][w]) and bvec of this pair
    const bf16_t* src = p.mixT + (long)item * 96 * 96;
    for (int i = tid; i < 96 * 12; i += 256) {
      const int row = i / 12, c = i - row * 12;
      const uint4 u = *reinterpret_cast<const uint4*>(src + row * 96 + c * 8);
      uint2* dst = reinterpret_cast<uint2*>(sM + row * WS + c * 8);
      dst[0] = make_uint2(u.x, u.y);
      dst[1] = make_uint2(u.z, u.w);
    }
    if (tid < 96) sB[tid] = p.bvec[(long)item * 96 + tid];
  }
  __syncthreads();

  auto stage = [&](int b) {
    const bf16_t* kbase = p.k + (long)b * CTXP * p.ldk + h * D;
    for (int idx = tid; idx < CTXP * H::DCH; idx += 256) {
      const int row = idx / H::DCH, c = idx - row * H::DCH;
      *reinterpret_cast<uint4*>(sK + row * H::KS + c * 8) =
          *reinterpret_cast<const uint4*>(kbase + (long)row * p.ldk + c * 8);
    }
    const bf16_t* vbase = p.vt + (long)h * D * p.ldvt + (long)b * CTXP;
    for (int idx = tid; idx < D * (CTXP / 8); idx += 256) {
      const int row = idx / (CTXP / 8), c = idx - row * (CTXP / 8);
      const uint4 u = *reinterpret_cast<const uint4*>(vbase + (long)row * p.ldvt + c * 8);
      uint2* dst = reinterpret_cast<uint2*>(sV + row * WS + c * 8);
      dst[0] = make_uint2(u.x, u.y);
      dst[1] = make_uint2(u.z, u.w);
    }
  };

  auto load_q = [&](int b, bf16x8* qf) {
    const bf16_t* qp = p.q + ((long)b * p.N + (q_ok ? q_row : 0)) * p.ldq + h * D;
#pragma unroll
    for (int ks = 0; ks < H::DK; ++ks) {
      const int c = ks * 16 + hi * 8;
      union { uint4 u; bf16x8 v; } x;
      x.u = make_uint4(0, 0, 0, 0);
      if (q_ok && c < D) x.u = *reinterpret_cast<const uint4*>(qp + c);
      qf[ks] = x.v;
    }
  };

  // exact softmax over the 77 valid keys; returns normalised probabilities in pr[3][16]
  auto probs = [&](const bf16x8* qf, float (*pr)[16]) {
    float mx = -1e30f;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < H::DK; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (nt * 32 + ql) * H::KS + ks * 16 + hi * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float v = n < HEDIT_MAXW ? s[r] : -1e30f;
        pr[nt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ls = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { pr[nt][r] = fast_exp2(pr[nt][r] - mx); ls += pr[nt][r]; }
    ls += __shfl_xor(ls, 32, 64);
    const float inv = 1.0f / ls;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[nt][r] *= inv;
  };

  auto accumulate_store = [&](int which, float (*pr)[16]) {
    if (p.store == nullptr || !q_ok) return;
    float* dst = p.store + ((((long)item * 2 + which) * p.heads + h) * p.N + q_row) * HEDIT_MAXW;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (n < HEDIT_MAXW) dst[n] += pr[nt][r];
      }
  };

  auto pv_and_write = [&](int b, const bf16x8 (*pf)[2]) {
    f32x16 o[H::DT];
#pragma unroll
    for (int dt = 0; dt < H::DT; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 vf = read_perm_frag(sV, dt * 32 + ql, WS, nt * 32 + 16 * s2 + 4 * hi);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[nt][s2], o[dt], 0, 0, 0);
        }
    }
    if (q_ok) {
      bf16_t* op = p.out + ((long)b * p.N + q_row) * p.ldo + h * D;
#pragma unroll
      for (int dt = 0; dt < H::DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = dt * 32 + 8 * g + 4 * hi;
          if (d0 < D) {
            uint2 w;
            w.x = pack_bf16x2(o[dt][g * 4 + 0], o[dt][g * 4 + 1]);
            w.y = pack_bf16x2(o[dt][g * 4 + 2], o[dt][g * 4 + 3]);
            *reinterpret_cast<uint2*>(op + d0) = w;
          }
        }
    }
  };

  bf16x8 qf[H::DK];
  float pr[3][16];
  bf16x8 pf[3][2];

  // ---- first (or only) row
  stage(b0);
  __syncthreads();
  load_q(b0, qf);
  probs(qf, pr);
  if (is_pair) accumulate_store(0, pr);
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) { pf[nt][0] = pack_p8(pr[nt]); pf[nt][1] = pack_p8(pr[nt] + 8); }
  pv_and_write(b0, pf);
  if (!is_pair) return;

  // ---- mixed source part of the target probabilities: mix[n][q] = sum_w mixT[n][w] P_src[w][q]
  f32x16 mix[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mix[nt][r] = 0.f;
#pragma unroll
    for (int wt = 0; wt < 3; ++wt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 mf = read_perm_frag(sM, nt * 32 + ql, WS, wt * 32 + 16 * s2 + 4 * hi);
        mix[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf, pf[wt][s2], mix[nt], 0, 0, 0);
      }
  }
  __syncthreads();          // everyone is done with the source K / V^T
  stage(b1);
  __syncthreads();
  load_q(b1, qf);
  probs(qf, pr);
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sB + nt * 32 + 8 * g + 4 * hi);
#pragma unroll
      for (int j = 0; j < 4; ++j) pr[nt][g * 4 + j] = mix[nt][g * 4 + j] + bv[j] * pr[nt][g * 4 + j];
    }
  accumulate_store(1, pr);
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) { pf[nt][0] = pack_p8(pr[nt]); pf[nt][1] = pack_p8(pr[nt] + 8); }
  pv_and_write(b1, pf);
}

template <int D>
int launch_cross(const CrossAttnParams& p, hipStream_t st) {
  using S = CrossSmem<D>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_kernel<D>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S::TOTAL));
    attr_set = true;
  }
  dim3 grid(cdiv(p.N, 128), p.heads, p.n_pairs + p.n_single);
  hipLaunchKernelGGL((cross_attn_kernel<D>), grid, dim3(256), S::TOTAL, st, p);
  LAUNCH_CHECK();
  return HEDIT_OK;
}

}  // namespace

int self_attn_launch(const SelfAttnParams& p, hipStream_t st) {
  ARG_CHECK(p.N % 64 == 0, "self_attn: N must be a multiple of 64");
  ARG_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldvt % 8 == 0 && p.ldo % 4 == 0, "self_attn: strides");
  dim3 grid(cdiv(p.N, 128), p.heads, p.B);
#define SA(DV) case DV: hipLaunchKernelGGL((self_attn_kernel<DV>), grid, dim3(256), 0, st, p); break;
  switch (p.d) {
    SA(32) SA(40) SA(64) SA(80) SA(160)
    default:
      hedit_set_error("self_attn: unsupported head dim " + std::to_string(p.d) + " (32,40,64,80,160)");
      return HEDIT_ERR_ARG;
  }
#undef SA
  LAUNCH_CHECK();
  return HEDIT_OK;
}

int cross_attn_launch(const CrossAttnParams& p, hipStream_t st) {
  ARG_CHECK(p.n_pairs + p.n_single > 0, "cross_attn: nothing to do");
  ARG_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldvt % 8 == 0 && p.ldo % 4 == 0, "cross_attn: strides");
  switch (p.d) {
    case 32: return launch_cross<32>(p, st);
    case 40: return launch_cross<40>(p, st);
    case 64: return launch_cross<64>(p, st);
    case 80: return launch_cross<80>(p, st);
    case 160: return launch_cross<160>(p, st);
    default:
      hedit_set_error("cross_attn: unsupported head dim " + std::to_string(p.d) + " (32,40,64,80,160)");
      return HEDIT_ERR_ARG;
  }
}
