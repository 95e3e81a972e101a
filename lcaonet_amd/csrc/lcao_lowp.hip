// lcao_lowp.hip — BASELINE config 3: the fused edge kernels on the 16-bit matrix path (v_mfma_f32_16x16x32_{bf16,f16},
// 16x the fp32 matrix rate per instruction), fp32 accumulation.  Two precisions, one template (PREC):
//   PREC_BF16X3  every fp32 operand split into two bf16 terms, three MFMAs per product: ~1e-5 from fp32 (the fp32 bar)
//   PREC_F16     operands rounded once to fp16 (11 significant bits), ONE MFMA per product: ~3e-4, inside config 3's
//                stated tolerance of 1e-3, which a single bf16 MFMA (8 bits: 3.6e-3 on the MLP alone) does not meet
#include "lcao_mfma.h"

enum { PREC_BF16X3 = 0, PREC_F16 = 1 };

// ===================================================================================================
// lcao_coeff_mlp_contract_bf16x3 / _f16 — BASELINE config 3 ("bf16 MFMA on the per-edge coeffs x hidden GEMM"):
// the same fused kernel as lcao_coeff_mlp_contract with both Dense layers on v_mfma_f32_16x16x32_bf16.
// Plain bf16 operands cost 3.6e-3 relative error on this MLP (worse than config 3's own 1e-3 bound), so every fp32
// operand is split into two bf16 terms, x = xh + xl (16 mantissa bits), and a product runs as three MFMAs with fp32
// accumulation:  wh.xl + wl.xh + wh.xh  (the dropped wl.xl is 2^-16 of the product): 6-9e-6 relative error on the
// two-layer MLP at 3/16 of the fp32 matrix time ("bf16x3").  Weights are split once per workgroup while staging
// (hi/lo planes, 139 KB of LDS at C = V = 128); activations are split in registers.
//   layer 1, k-step s (32 inputs):  lane (n, g) feeds X[n][32s + 8g + j], j < 8      (natural order, 32 B per lane)
//   layer 2, k-step s (32 hidden):  lane (n, g) feeds h tile 2s regs 0..3 (features 32s + 4g + j) and h tile 2s+1
//                                   regs 0..3 (features 32s + 16 + 4g + j): the W2 planes are stored column-permuted
//                                   to match, so layers still chain register to register.
// ===================================================================================================
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
#define MFMA16BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

template <int PREC> struct LowP;
template <> struct LowP<PREC_BF16X3> { typedef __bf16 T; static constexpr int PLANES = 2; };
template <> struct LowP<PREC_F16> { typedef _Float16 T; static constexpr int PLANES = 1; };

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8v& hi, bf16x8v& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    hi[j] = h;
    lo[j] = (__bf16)(x[j] - (float)h);
  }
}

// Weight planes in LDS: [plane][row][K + 8] 16-bit elements (272-B rows at K = 128: the 16 rows of a fragment hit 16
// distinct 16-B slots).  PERMUTE stores the columns in the order a chained layer delivers its inputs (see header).
template <int PREC, bool PERMUTE>
__device__ __forceinline__ void stage_weights_lowp(typename LowP<PREC>::T* __restrict__ planes, const float* __restrict__ W,
                                                   int rows, int K, float scale = 1.0f) {
  typedef typename LowP<PREC>::T T;
  const int ld = K + 8;
  T* hi = planes;
  T* lo = planes + rows * ld;
  for (int i = threadIdx.x; i < rows * K; i += blockDim.x) {
    const int rr = i / K, pos = i - rr * K;
    int col = pos;
    if (PERMUTE) {  // pos = 32 s + 8 g + j  ->  column 32 s + (j < 4 ? 4 g + j : 16 + 4 g + (j - 4))
      const int s = pos >> 5, g = (pos >> 3) & 3, j = pos & 7;
      col = 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
    }
    const float w = W[(int64_t)rr * K + col] * scale;
    const T h = (T)w;
    hi[rr * ld + pos] = h;
    if (PREC == PREC_BF16X3) lo[rr * ld + pos] = (T)(w - (float)h);
  }
}

// one Dense layer on a 16-row tile: out[m] += W[16m.., :] . in, input given as 8 floats per 32-wide k-step in B-operand order
template <int PREC, int KS, int MT>
__device__ __forceinline__ void lowp_layer(const typename LowP<PREC>::T* __restrict__ planes, int rows, int ld, int r, int q,
                                           const float (&in)[KS][8], f32x4v (&out)[MT]) {
  if constexpr (PREC == PREC_BF16X3) {
    const __bf16* wh = planes;
    const __bf16* wl = planes + rows * ld;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8v xh, xl;
      split8(in[s], xh, xl);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8v a_h = *(const bf16x8v*)(wh + (16 * m + r) * ld + 32 * s + 8 * q);
        const bf16x8v a_l = *(const bf16x8v*)(wl + (16 * m + r) * ld + 32 * s + 8 * q);
        out[m] = MFMA16BF(a_h, xl, out[m]);
        out[m] = MFMA16BF(a_l, xh, out[m]);
        out[m] = MFMA16BF(a_h, xh, out[m]);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f16x8v xh;
#pragma unroll
      for (int j = 0; j < 8; ++j) xh[j] = (_Float16)in[s][j];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f16x8v a_h = *(const f16x8v*)(planes + (16 * m + r) * ld + 32 * s + 8 * q);
        out[m] = MFMA16H(a_h, xh, out[m]);
      }
    }
  }
}

template <int PREC, int C, int V, int NL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_coeff_mlp_lowp(
    const float* __restrict__ cst, const float* __restrict__ rb, const int32_t* __restrict__ order,
    const int32_t* __restrict__ cst_row, const int32_t* __restrict__ l_of_d, const float* __restrict__ W1,
    const float* __restrict__ W2, int64_t E, int D, float* __restrict__ A, float* __restrict__ Bout,
    const float* __restrict__ gate, const int32_t* __restrict__ src_p, float* __restrict__ Ag, float* __restrict__ gram,
    unsigned* __restrict__ tile_counter) {
  static_assert(C % 32 == 0 && V % 32 == 0, "k-steps are 32 wide");
  typedef typename LowP<PREC>::T T;
  constexpr int KS1 = C / 32, KS2 = V / 32, MT = V / 16, LD1 = C + 8, LD2 = V + 8, WAVES = THREADS / 64;
  constexpr int NP = LowP<PREC>::PLANES;
  extern __shared__ unsigned char lds_raw[];
  T* w1p = (T*)lds_raw;
  T* w2p = w1p + NP * V * LD1;
  // log2(e) folded into the first layer (lcao_mfma.h: g(x') = SiLU(x) log2 e for x' = x log2 e, so the second layer's
  // input and output carry the factor and it leaves through the radial multiplier): one VALU multiply less per activation
  stage_weights_lowp<PREC, false>(w1p, W1, V, C, LCAO_LOG2E);
  stage_weights_lowp<PREC, true>(w2p, W2, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  const int64_t first_dynamic = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  for (; tile < n_tiles;) {
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const int64_t e = order ? order[pc] : pc;
    const int64_t ec = cst_row ? (int64_t)cst_row[pc] : e;   // gathered coefficient rows (see k_coeff_mlp)
    const float* xrow = cst + ec * (int64_t)D * C + 8 * q;
    const float* rbrow = rb + e * D;
    f32x4v acc[NL][MT];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[l][m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v x[KS1][2];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      x[s][0] = *(const f32x4v*)(xrow + 32 * s);
      x[s][1] = *(const f32x4v*)(xrow + 32 * s + 4);
    }
    for (int d = 0; d < D; ++d) {
      asm volatile("" ::: "memory");  // the LDS fragments are loop-invariant: keep them from being hoisted (spills)
      f32x4v h[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) h[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
      float xin[KS1][8];
#pragma unroll
      for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xin[s][j] = x[s][0][j];
          xin[s][4 + j] = x[s][1][j];
        }
      lowp_layer<PREC, KS1, MT>(w1p, V, LD1, r, q, xin, h);
      if (d + 1 < D) {
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
          x[s][0] = *(const f32x4v*)(xrow + (d + 1) * C + 32 * s);
          x[s][1] = *(const f32x4v*)(xrow + (d + 1) * C + 32 * s + 4);
        }
      }
      float hin[KS2][8];
#pragma unroll
      for (int s = 0; s < KS2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hin[s][j] = lcao_silu_log2e(h[2 * s][j]);
          hin[s][4 + j] = lcao_silu_log2e(h[2 * s + 1][j]);
        }
      f32x4v o[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) o[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
      lowp_layer<PREC, KS2, MT>(w2p, V, LD2, r, q, hin, o);
      const float w = rbrow[d] * LCAO_INV_LOG2E;
      const int l = l_of_d[d];
#pragma unroll
      for (int ll = 0; ll < NL; ++ll) {
        if (l == ll) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[ll][m] += w * lcao_silu_log2e4(o[m]);
        }
      }
    }
    // epilogue: identical to k_coeff_mlp (B, Gram of the ungated A, gated copy or raw A)
    f32x4v bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      bsum[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int l = 0; l < NL; ++l) bsum[m] += acc[l][m];
    }
    if (gram) {
      float g[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) g[i] = 0.f;
      int gi = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int l2 = l; l2 < 4; ++l2) {
          if (l < NL && l2 < NL) {
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int j = 0; j < 4; ++j) t += acc[l][m][j] * acc[l2][m][j];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            g[gi] = t;
          }
          ++gi;
        }
      if (valid && q == 0) {
        float* gp = gram + p * 10;
#pragma unroll
        for (int i = 0; i < 10; ++i) gp[i] = g[i];
      }
    }
    if (valid) {
      float* Bp = Bout + p * (int64_t)V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        *(f32x4v*)(Bp + 16 * m) = bsum[m];
        if (Ag) {
          const f32x4v gt = *(const f32x4v*)(gate + (int64_t)src_p[p] * V + 16 * m + 4 * q);
          float* Agp = Ag + p * (int64_t)NL * V + 4 * q;
#pragma unroll
          for (int l = 0; l < NL; ++l) *(f32x4v*)(Agp + l * V + 16 * m) = acc[l][m] * gt;
        } else {
          float* Ap = A + p * (int64_t)NL * V + 4 * q;
#pragma unroll
          for (int l = 0; l < NL; ++l) *(f32x4v*)(Ap + l * V + 16 * m) = acc[l][m];
        }
      }
    }
    unsigned nxt = 0;
    if (lane == 0) nxt = atomicAdd(tile_counter, 1u);
    tile = first_dynamic + (int64_t)__builtin_amdgcn_readfirstlane(nxt);
  }
  if (lane == 0) lcao_tile_queue_leave(tile_counter, gridDim.x * (THREADS / 64));   // the last wave re-arms the queue
}

template <int PREC, int C, int V, int NL>
static int launch_coeff_mlp_lowp(const float* cst, const float* rb, const int32_t* order, const int32_t* cst_row,
                                 const int32_t* l_of_d, const float* W1, const float* W2, int64_t E, int D, float* A,
                                 float* Bout, const float* gate, const int32_t* src_p, float* Ag, float* gram,
                                 unsigned* counter, hipStream_t st) {
  constexpr int THREADS = 512;
  constexpr int NP = LowP<PREC>::PLANES;
  size_t lds = (size_t(NP) * V * (C + 8) + size_t(NP) * V * (V + 8)) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_coeff_mlp_lowp<PREC, C, V, NL, THREADS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return int(e);
    attr_set = true;
  }
  constexpr int WAVES = THREADS / 64;
  int64_t tiles = (E + 15) / 16;
  int64_t groups = (tiles + WAVES - 1) / WAVES;
  int64_t cap = (int64_t)lcao_num_cu() * ((lds > 80 * 1024) ? 1 : 2);
  if (groups > cap) groups = cap;
  hipLaunchKernelGGL((k_coeff_mlp_lowp<PREC, C, V, NL, THREADS>), dim3((unsigned)groups), dim3(THREADS), lds, st, cst, rb,
                     order, cst_row, l_of_d, W1, W2, E, D, A, Bout, gate, src_p, Ag, gram, counter);
  return lcao_launch_status();
}

template <int PREC>
static int coeff_mlp_contract_lowp(const float* cst, const float* rb, const int32_t* order, const int32_t* cst_row,
                                   const int32_t* l_of_d, const float* W1, const float* W2, const float* gate,
                                   const int32_t* src_p, int64_t n_edges, int n_orb, int C, int V, int n_l, float* A, float* B,
                                   float* Ag, float* gram, void* workspace, size_t ws_bytes, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || C <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(cst); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(W1); LCAO_CHECK_PTR(W2);
  LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(workspace);
  if (Ag) { LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(gram); } else { LCAO_CHECK_PTR(A); }
  if (ws_bytes < 2 * sizeof(unsigned)) return LCAO_ERR_INVALID;   // {drawn, left} of the tile queue
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctr = (unsigned*)workspace;
#define LP_CASE(C_, V_, NL_)                                                                                       \
  if (C == C_ && V == V_ && n_l == NL_)                                                                            \
    return launch_coeff_mlp_lowp<PREC, C_, V_, NL_>(cst, rb, order, cst_row, l_of_d, W1, W2, n_edges, n_orb, A, B, gate, \
                                                    src_p, Ag, Ag ? gram : nullptr, ctr, st);
#define LP_CV(C_, V_) LP_CASE(C_, V_, 1) LP_CASE(C_, V_, 2) LP_CASE(C_, V_, 3) LP_CASE(C_, V_, 4)
  LP_CV(128, 128) LP_CV(64, 64) LP_CV(32, 32)
#undef LP_CV
#undef LP_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_coeff_mlp_contract_bf16x3(const float* cst, const float* rb, const int32_t* order,
                                              const int32_t* cst_row, const int32_t* l_of_d, const float* W1, const float* W2,
                                              const float* gate, const int32_t* src_p, int64_t n_edges, int n_orb, int C,
                                              int V, int n_l, float* A, float* B, float* Ag, float* gram,
                                              void* workspace, size_t ws_bytes, void* stream) {
  return coeff_mlp_contract_lowp<PREC_BF16X3>(cst, rb, order, cst_row, l_of_d, W1, W2, gate, src_p, n_edges, n_orb, C, V, n_l,
                                              A, B, Ag, gram, workspace, ws_bytes, stream);
}

extern "C" int lcao_coeff_mlp_contract_f16(const float* cst, const float* rb, const int32_t* order, const int32_t* cst_row,
                                           const int32_t* l_of_d, const float* W1, const float* W2, const float* gate,
                                           const int32_t* src_p, int64_t n_edges, int n_orb, int C, int V, int n_l, float* A,
                                           float* B, float* Ag, float* gram, void* workspace, size_t ws_bytes, void* stream) {
  return coeff_mlp_contract_lowp<PREC_F16>(cst, rb, order, cst_row, l_of_d, W1, W2, gate, src_p, n_edges, n_orb, C, V, n_l, A,
                                           B, Ag, gram, workspace, ws_bytes, stream);
}

// ===================================================================================================
// low-precision forms of the two-body kernels (same operand treatment; `precision="bf16x3"` / `"f16"`)
// ===================================================================================================
template <int PREC, int V>
__global__ __launch_bounds__(256) void k_edge_lcao_weight_lowp(const float* __restrict__ tbw, const float* __restrict__ B,
                                                               const float* __restrict__ W3, const float* __restrict__ Wb,
                                                               int64_t E, float* __restrict__ bw) {
  typedef typename LowP<PREC>::T T;
  constexpr int MT = V / 16, KS = V / 32, LD = V + 8, NP = LowP<PREC>::PLANES;
  extern __shared__ unsigned char lds_raw[];
  T* w3p = (T*)lds_raw;
  T* wbp = w3p + NP * V * LD;
  stage_weights_lowp<PREC, false>(w3p, W3, V, V);
  stage_weights_lowp<PREC, true>(wbp, Wb, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    float xin[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f32x4v a = *(const f32x4v*)(tbw + pc * V + 32 * s + 8 * q);
      const f32x4v b = *(const f32x4v*)(tbw + pc * V + 32 * s + 8 * q + 4);
      xin[s][0] = a[0]; xin[s][1] = a[1]; xin[s][2] = a[2]; xin[s][3] = a[3];
      xin[s][4] = b[0]; xin[s][5] = b[1]; xin[s][6] = b[2]; xin[s][7] = b[3];
    }
    f32x4v bb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) bb[m] = *(const f32x4v*)(B + pc * V + 16 * m + 4 * q);
    f32x4v ft[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ft[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    lowp_layer<PREC, KS, MT>(w3p, V, LD, r, q, xin, ft);
    float n2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = bb[m][j] + bb[m][j] * ft[m][j];
        ft[m][j] = u;
        n2 += u * u;
      }
    n2 += __shfl_xor(n2, 16);
    n2 += __shfl_xor(n2, 32);
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    float lin[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lin[s][j] = ft[2 * s][j] * inv;
        lin[s][4 + j] = ft[2 * s + 1][j] * inv;
      }
    f32x4v o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    lowp_layer<PREC, KS, MT>(wbp, V, LD, r, q, lin, o);
    if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(bw + p * V + 16 * m + 4 * q) = o[m];
    }
  }
}

template <int PREC, int V>
__global__ __launch_bounds__(256) void k_edge_node_message_lowp(
    const float* __restrict__ bw, const float* __restrict__ Ps, const float* __restrict__ Pt,
    const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p, const float* __restrict__ Wn2,
    const float* __restrict__ bn2, int64_t E, const int32_t* __restrict__ skip_if /* nullable: return when it reads 1 */,
    float* __restrict__ msg) {
  typedef typename LowP<PREC>::T T;
  constexpr int MT = V / 16, KS = V / 32, LD = V + 8;
  extern __shared__ unsigned char lds_raw[];
  if (skip_if && *skip_if != 0) return;
  T* wp = (T*)lds_raw;
  stage_weights_lowp<PREC, false>(wp, Wn2, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const float* ps = Ps + (int64_t)src_p[pc] * V + 8 * q;
    const float* pt = Pt + (int64_t)dst_p[pc] * V + 8 * q;
    float hin[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f32x4v a = *(const f32x4v*)(ps + 32 * s) + *(const f32x4v*)(pt + 32 * s);
      const f32x4v b = *(const f32x4v*)(ps + 32 * s + 4) + *(const f32x4v*)(pt + 32 * s + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { hin[s][j] = lcao_silu(a[j]); hin[s][4 + j] = lcao_silu(b[j]); }
    }
    f32x4v o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);
    lowp_layer<PREC, KS, MT>(wp, V, LD, r, q, hin, o);
    if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4v w = *(const f32x4v*)(bw + p * V + 16 * m + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] *= lcao_silu(o[m][j]);
        *(f32x4v*)(msg + p * V + 16 * m + 4 * q) = w;
      }
    }
  }
}

// "one centre node per MFMA column" form of the message kernel with the E -> N sum fused in (see lcao_mfma.hip,
// k_edge_node_message_cols): step t multiplies the t-th out-edge of 16 consecutive centre nodes, the node sum is a per-lane
// accumulation.  Chosen on the device by lcao_node_column_hint; the edge-tiled kernel above + k_rows_segment_sum serve
// graphs with uneven degrees.
template <int PREC, int V>
__global__ __launch_bounds__(256) void k_edge_node_message_cols_lowp(
    const float* __restrict__ bw, const float* __restrict__ Ps, const float* __restrict__ Pt,
    const int32_t* __restrict__ dst_p, const int32_t* __restrict__ out_ptr, const float* __restrict__ Wn2,
    const float* __restrict__ bn2, int n_nodes, int n_edges, const int32_t* __restrict__ hint, float* __restrict__ agg) {
  typedef typename LowP<PREC>::T T;
  constexpr int MT = V / 16, KS = V / 32, LD = V + 8;
  extern __shared__ unsigned char lds_raw[];
  if (hint && *hint == 0) return;
  const bool by_degree = hint && *hint == 2;
  T* wp = (T*)lds_raw;
  stage_weights_lowp<PREC, false>(wp, Wn2, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n_groups = (n_nodes + 15) >> 4;
  for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
    const bool node_ok = 16 * g + r < n_nodes;
    const int slot = node_ok ? 16 * g + r : n_nodes - 1;
    const int s = by_degree ? hint[1 + slot] : slot;          // nodes grouped by degree (lcao_node_column_hint, mode 2)
    const int b = out_ptr[s];
    const int deg = node_ok ? out_ptr[s + 1] - b : 0;
    int Tn = deg;
    Tn = max(Tn, __shfl_xor(Tn, 1)); Tn = max(Tn, __shfl_xor(Tn, 2)); Tn = max(Tn, __shfl_xor(Tn, 4)); Tn = max(Tn, __shfl_xor(Tn, 8));
    Tn = __builtin_amdgcn_readfirstlane(Tn);
    f32x4v acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    if (Tn > 0) {
      auto edge_at = [&](int t) { return deg > 0 ? b + min(t, deg - 1) : min(b, n_edges - 1); };
      // PF: the target rows of step t + 1 are requested before step t's GEMM (fp16: registers allow it); bf16x3 (three MFMAs
      // and two operand splits per product) has no room for that at two waves per SIMD and reads them at the top of the step
      constexpr bool PF = PREC != PREC_BF16X3;
      float hin[KS][8];
      const float* pr = Ps + (int64_t)s * V + 8 * q;   // the centre's own row: re-read per step (an L1 hit; 32 VGPRs saved)
      auto first_layer = [&](int d) {
        const float* pt = Pt + (int64_t)d * V + 8 * q;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const f32x4v a0 = *(const f32x4v*)(pt + 32 * k) + *(const f32x4v*)(pr + 32 * k);
          const f32x4v a1 = *(const f32x4v*)(pt + 32 * k + 4) + *(const f32x4v*)(pr + 32 * k + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { hin[k][j] = lcao_silu(a0[j]); hin[k][4 + j] = lcao_silu(a1[j]); }
        }
      };
      int d_cur = dst_p[edge_at(0)], d_nxt = dst_p[edge_at(1)];
      if (PF) first_layer(d_cur);
      for (int t = 0; t < Tn; ++t) {
        asm volatile("" ::: "memory");
        const int64_t pw = edge_at(t);
        const float live = t < deg ? 1.0f : 0.0f;
        f32x4v wv[MT], o[MT];
        if (!PF) first_layer(d_cur);
        f32x4v ntv[PF ? KS : 1][2];
        if (PF) {
          const float* pt = Pt + (int64_t)d_nxt * V + 8 * q;
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            ntv[PF ? k : 0][0] = *(const f32x4v*)(pt + 32 * k) + *(const f32x4v*)(pr + 32 * k);
            ntv[PF ? k : 0][1] = *(const f32x4v*)(pt + 32 * k + 4) + *(const f32x4v*)(pr + 32 * k + 4);
          }
        }
        d_cur = d_nxt;
        d_nxt = dst_p[edge_at(t + 2)];
#pragma unroll
        for (int m = 0; m < MT; ++m) wv[m] = *(const f32x4v*)(bw + pw * V + 16 * m + 4 * q);
#pragma unroll
        for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);
        lowp_layer<PREC, KS, MT>(wp, V, LD, r, q, hin, o);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[m][j] += (wv[m][j] * live) * lcao_silu(o[m][j]);
        if (PF) {
#pragma unroll
          for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              hin[k][j] = lcao_silu(ntv[PF ? k : 0][0][j]);
              hin[k][4 + j] = lcao_silu(ntv[PF ? k : 0][1][j]);
            }
        }
      }
    }
    if (node_ok) {
      float* out = agg + (int64_t)s * V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(out + 16 * m) = acc[m];
    }
  }
}

// out[s, :] = sum of rows[out_ptr[s] : out_ptr[s+1], :]  (one wave per centre node, lanes over channels, fixed order);
// returns at once when *skip_if reads 1
__global__ void k_rows_segment_sum(const float* __restrict__ rows, const int32_t* __restrict__ out_ptr, int n_nodes, int V,
                                   const int32_t* __restrict__ skip_if, float* __restrict__ out) {
  if (skip_if && *skip_if != 0) return;
  const int lane = threadIdx.x & 63;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], p1 = out_ptr[s + 1];
    for (int ch = 4 * lane; ch < V; ch += 256) {
      f32x4v a = {0.f, 0.f, 0.f, 0.f};
      for (int p = p0; p < p1; ++p) a += *(const f32x4v*)(rows + (int64_t)p * V + ch);
      *(f32x4v*)(out + (int64_t)s * V + ch) = a;
    }
  }
}

template <int PREC>
static int edge_lcao_weight_lowp(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges, int V,
                                 float* bw, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(tbw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(W3); LCAO_CHECK_PTR(Wb); LCAO_CHECK_PTR(bw);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(2) * LowP<PREC>::PLANES * V * (V + 8) * 2;
#define LWL_CASE(V_)                                                                                                \
  if (V == V_) {                                                                                                    \
    static bool done = false;                                                                                       \
    int rc = set_lds_attr(k_edge_lcao_weight_lowp<PREC, V_>, lds, done);                                            \
    if (rc != LCAO_OK) return rc;                                                                                   \
    hipLaunchKernelGGL((k_edge_lcao_weight_lowp<PREC, V_>), dim3(persistent_groups(n_edges, lds)), dim3(256), lds, st, tbw, \
                       B, W3, Wb, n_edges, bw);                                                                     \
    return lcao_launch_status();                                                                                    \
  }
  LWL_CASE(128) LWL_CASE(64) LWL_CASE(32)
#undef LWL_CASE
  return LCAO_ERR_UNSUPPORTED;
}

template <int PREC>
static int edge_node_message_lowp(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                  const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V,
                                  float* msg, void* stream, const int32_t* column_hint = nullptr) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(bw); LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p);
  LCAO_CHECK_PTR(Wn2); LCAO_CHECK_PTR(bn2); LCAO_CHECK_PTR(msg);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(LowP<PREC>::PLANES) * V * (V + 8) * 2;
#define NML_CASE(V_)                                                                                                 \
  if (V == V_) {                                                                                                     \
    static bool done = false;                                                                                        \
    int rc = set_lds_attr(k_edge_node_message_lowp<PREC, V_>, lds, done);                                            \
    if (rc != LCAO_OK) return rc;                                                                                    \
    hipLaunchKernelGGL((k_edge_node_message_lowp<PREC, V_>), dim3(persistent_groups(n_edges, lds)), dim3(256), lds, st, bw, \
                       Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, column_hint, msg);                                   \
    return lcao_launch_status();                                                                                     \
  }
  NML_CASE(128) NML_CASE(64) NML_CASE(32)
#undef NML_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_edge_lcao_weight_bf16x3(const float* tbw, const float* B, const float* W3, const float* Wb,
                                            int64_t n_edges, int V, float* bw, void* stream) {
  return edge_lcao_weight_lowp<PREC_BF16X3>(tbw, B, W3, Wb, n_edges, V, bw, stream);
}
extern "C" int lcao_edge_lcao_weight_f16(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges,
                                         int V, float* bw, void* stream) {
  return edge_lcao_weight_lowp<PREC_F16>(tbw, B, W3, Wb, n_edges, V, bw, stream);
}
extern "C" int lcao_edge_node_message_bf16x3(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                             const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges,
                                             int V, float* msg, void* stream) {
  return edge_node_message_lowp<PREC_BF16X3>(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg, stream);
}
extern "C" int lcao_edge_node_message_f16(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                          const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V,
                                          float* msg, void* stream) {
  return edge_node_message_lowp<PREC_F16>(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg, stream);
}

template <int PREC>
static int edge_node_message_agg_lowp(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                      const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                                      int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* msg,
                                      float* agg, void* stream) {
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0 || V % 4) return LCAO_ERR_INVALID;
  if (n_nodes == 0) return LCAO_OK;
  LCAO_CHECK_PTR(agg); LCAO_CHECK_PTR(out_ptr);
  hipStream_t st = (hipStream_t)stream;
  if (n_edges == 0) {
    hipError_t me = hipMemsetAsync(agg, 0, sizeof(float) * (size_t)n_nodes * V, st);
    return me == hipSuccess ? LCAO_OK : int(me);
  }
  LCAO_CHECK_PTR(msg);
  // uneven degrees: message rows, then the per-node sum (both return at once when the hint reads 1)
  int rc = edge_node_message_lowp<PREC>(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg, stream, column_hint);
  if (rc != LCAO_OK) return rc;
  hipLaunchKernelGGL(k_rows_segment_sum, dim3(lcao_blocks(n_nodes, 4, 65536)), dim3(256), 0, st, msg, out_ptr, (int)n_nodes, V,
                     column_hint, agg);
  if (!column_hint) return lcao_launch_status();
  size_t lds = size_t(LowP<PREC>::PLANES) * V * (V + 8) * 2;
  int64_t per_cu = (lds > 80 * 1024) ? 1 : ((lds > 53 * 1024) ? 2 : 3);
  int64_t cgroups = (((n_nodes + 15) >> 4) + 3) / 4, cap = (int64_t)lcao_num_cu() * per_cu;
  if (cgroups > cap) cgroups = cap;
#define NMC_CASE(V_)                                                                                                  \
  if (V == V_) {                                                                                                      \
    static bool done = false;                                                                                         \
    rc = set_lds_attr(k_edge_node_message_cols_lowp<PREC, V_>, lds, done);                                            \
    if (rc != LCAO_OK) return rc;                                                                                     \
    hipLaunchKernelGGL((k_edge_node_message_cols_lowp<PREC, V_>), dim3((unsigned)cgroups), dim3(256), lds, st, bw, Ps, Pt, \
                       dst_p, out_ptr, Wn2, bn2, (int)n_nodes, (int)n_edges, column_hint, agg);                       \
    return lcao_launch_status();                                                                                      \
  }
  NMC_CASE(128) NMC_CASE(64) NMC_CASE(32)
#undef NMC_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_edge_node_message_agg_bf16x3(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                                 const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2,
                                                 const float* bn2, int64_t n_nodes, int64_t n_edges, int V,
                                                 const int32_t* column_hint, float* msg, float* agg, void* stream) {
  return edge_node_message_agg_lowp<PREC_BF16X3>(bw, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, n_nodes, n_edges, V,
                                                 column_hint, msg, agg, stream);
}

extern "C" int lcao_edge_node_message_agg_f16(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                              const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2,
                                              const float* bn2, int64_t n_nodes, int64_t n_edges, int V,
                                              const int32_t* column_hint, float* msg, float* agg, void* stream) {
  return edge_node_message_agg_lowp<PREC_F16>(bw, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, n_nodes, n_edges, V, column_hint,
                                              msg, agg, stream);
}
