// lcao_mfma.hip — the dense contractions of the LCAOConv hot path on the gfx950 matrix cores.
//
// All kernels use ONE formulation (exact fp32: v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain):
//
//      Y^T (features x rows) = W (features x K)  .  X^T (K x rows)
//
//   * the weight tile is the MFMA "A" operand, read from LDS where the whole matrix lives for the
//     lifetime of a persistent workgroup (row stride K+4 floats, so the 16 rows of a fragment land on
//     different 16-B bank slots);
//   * 16 data rows (edges / nodes) are the "B" operand: lane (r = lane&15, q = lane>>4) owns data row r and
//     feeds k = 16a + 4q + b at MFMA step (a, b), i.e. it loads the 16-byte chunk X[r][16a+4q .. +3] straight
//     from global memory (every byte of a row is consumed by the wave that fetched it);
//   * the accumulator of feature tile m then holds, in lane (r, q) register j, feature 16m + 4q + j of data
//     row r — which is exactly the B-operand layout of the NEXT layer (k = 16t + 4q + j at step (t, j)), so a
//     chain of layers runs register to register: no LDS round trip, no barrier, no transposition.
//   Waves never talk to each other after the weights are staged: one __syncthreads per kernel.
#include "lcao_common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static int g_num_cu = 0;
static inline int lcao_num_cu() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cu = prop.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

// stage a (rows x K) row-major weight block into LDS with row stride K+4 (all threads of the workgroup)
__device__ __forceinline__ void stage_weights(float* __restrict__ dst, const float* __restrict__ W, int rows, int K,
                                              int ldw) {
  const int chunks = K >> 2;
  for (int i = threadIdx.x; i < rows * chunks; i += blockDim.x) {
    int rr = i / chunks, cc = i - rr * chunks;
    *(f32x4v*)(dst + rr * (K + 4) + 4 * cc) = *(const f32x4v*)(W + (int64_t)rr * ldw + 4 * cc);
  }
}

// one dense layer on a 16-row tile held in B-operand layout:  out[m] (+)= W[16m.., :] . in
//   KT = K/16 input chunks (f32x4 each), MT = N/16 output tiles.  Two output tiles are interleaved so that
//   consecutive MFMAs never wait on their own accumulator (40-cycle dependent latency vs 32-cycle issue).
template <int KT, int MT>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ wl, int ldw /*K+4*/, int r, int q,
                                           const f32x4v (&in)[KT], f32x4v (&out)[MT]) {
  static_assert(MT % 2 == 0, "feature tiles are processed in pairs");
#pragma unroll
  for (int m = 0; m < MT; m += 2) {
    const float* w0 = wl + (16 * m + r) * ldw + 4 * q;
    const float* w1 = w0 + 16 * ldw;
#pragma unroll
    for (int a = 0; a < KT; ++a) {
      f32x4v f0 = *(const f32x4v*)(w0 + 16 * a);
      f32x4v f1 = *(const f32x4v*)(w1 + 16 * a);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        out[m] = MFMA16(f0[b], in[a][b], out[m]);
        out[m + 1] = MFMA16(f1[b], in[a][b], out[m + 1]);
      }
    }
  }
}

// ===================================================================================================
// lcao_dense: Y = act(X W^T + bias) + residual.   MFMA path for K % 16 == 0 && N % 32 == 0, VALU otherwise.
// grid.x = N/64 column slices (the slice's weights live in LDS), grid.y = persistent row groups.
// ===================================================================================================
template <int KT>
__global__ __launch_bounds__(256) void k_dense_mfma(const float* __restrict__ X, int64_t ldx,
                                                    const float* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ R, int64_t ldr, float* __restrict__ Y,
                                                    int64_t ldy, int64_t M, int N, int act) {
  constexpr int K = KT * 16;
  extern __shared__ float lds[];
  const int n0 = blockIdx.x * 64;
  const int ncols = min(64, N - n0);  // multiple of 32
  stage_weights(lds, W + (int64_t)n0 * K, ncols, K, K);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (M + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.y * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.y * 4) {
    int64_t row = tile * 16 + r;
    bool valid = row < M;
    const float* xr = X + (valid ? row : M - 1) * ldx + 4 * q;
    f32x4v x[KT];
#pragma unroll
    for (int a = 0; a < KT; ++a) x[a] = *(const f32x4v*)(xr + 16 * a);
    f32x4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    if (ncols == 64) {
      mfma_layer<KT, 4>(lds, K + 4, r, q, x, acc);
    } else {
      f32x4v acc2[2] = {acc[0], acc[1]};
      mfma_layer<KT, 2>(lds, K + 4, r, q, x, acc2);
      acc[0] = acc2[0];
      acc[1] = acc2[1];
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (16 * t < ncols) {
          int n = n0 + 16 * t + 4 * q;
          f32x4v v = acc[t];
          if (bias) v += *(const f32x4v*)(bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = lcao_act(v[j], act);
          if (R) v += *(const f32x4v*)(R + row * ldr + n);
          *(f32x4v*)(Y + row * ldy + n) = v;
        }
      }
    }
  }
}

// VALU fallback for arbitrary K, N (tiny heads like out_size = 1, odd widths in the reference's tests):
// one wave per output row, lanes over output columns, X row broadcast from L1.
__global__ void k_dense_valu(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                             const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr,
                             float* __restrict__ Y, int64_t ldy, int64_t M, int K, int N, int act) {
  int lane = threadIdx.x & 63;
  int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; row < M; row += nwave) {
    const float* xr = X + row * ldx;
    for (int n = lane; n < N; n += 64) {
      const float* wr = W + (int64_t)n * K;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
      if (bias) acc += bias[n];
      acc = lcao_act(acc, act);
      if (R) acc += R[row * ldr + n];
      Y[row * ldy + n] = acc;
    }
  }
}

template <int KT>
static int launch_dense_mfma(const float* X, int64_t ldx, const float* W, const float* bias, const float* R,
                             int64_t ldr, float* Y, int64_t ldy, int64_t M, int N, int act, hipStream_t st) {
  constexpr int K = KT * 16;
  size_t lds = size_t(64) * (K + 4) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_dense_mfma<KT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return int(e);
    attr_set = true;
  }
  int nslice = (N + 63) / 64;
  int64_t tiles = (M + 15) / 16;
  int64_t groups = (tiles + 3) / 4;
  int64_t cap = (int64_t)lcao_num_cu() * 4 / nslice;
  if (cap < 1) cap = 1;
  if (groups > cap) groups = cap;
  hipLaunchKernelGGL(k_dense_mfma<KT>, dim3(nslice, (unsigned)groups), dim3(256), lds, st, X, ldx, W, bias, R, ldr,
                     Y, ldy, M, N, act);
  return lcao_launch_status();
}

extern "C" int lcao_dense(const float* X, int64_t ldx, const float* W, const float* bias, const float* residual,
                          int64_t ldr, float* Y, int64_t ldy, int64_t M, int K, int N, int act, void* stream) {
  LCAO_CHECK_SIZE(M);
  if (K <= 0 || N <= 0 || act < 0 || act > 2) return LCAO_ERR_INVALID;
  if (M == 0) return LCAO_OK;
  LCAO_CHECK_PTR(X); LCAO_CHECK_PTR(W); LCAO_CHECK_PTR(Y);
  hipStream_t st = (hipStream_t)stream;
  bool vec_ok = (ldx % 4 == 0) && (ldy % 4 == 0) && (!residual || ldr % 4 == 0) &&
                ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0) && ((uintptr_t)W % 16 == 0) &&
                (!bias || (uintptr_t)bias % 16 == 0) && (!residual || (uintptr_t)residual % 16 == 0);
  if (vec_ok && K % 16 == 0 && N % 32 == 0 && K <= 512) {
#define DENSE_CASE(KT_)                                                                                  \
  case KT_:                                                                                              \
    return launch_dense_mfma<KT_>(X, ldx, W, bias, residual, ldr, Y, ldy, M, N, act, st)
    switch (K / 16) {
      DENSE_CASE(1); DENSE_CASE(2); DENSE_CASE(3); DENSE_CASE(4); DENSE_CASE(5); DENSE_CASE(6); DENSE_CASE(7);
      DENSE_CASE(8); DENSE_CASE(12); DENSE_CASE(16); DENSE_CASE(24); DENSE_CASE(32);
      default: break;
    }
#undef DENSE_CASE
  }
  hipLaunchKernelGGL(k_dense_valu, dim3(lcao_blocks(M, 4, 65536)), dim3(256), 0, st, X, ldx, W, bias, residual, ldr,
                     Y, ldy, M, K, N, act);
  return lcao_launch_status();
}

// ===================================================================================================
// lcao_coeff_mlp_contract — lcaonet.py:170 (f_coeffs: two bias-free Dense + SiLU on every (edge, orbital)
// row) fused with the radial contraction of lcaonet.py:173-183,200-203.  The (E, D, V) tensor
// f_coeffs(cst) — 4.7 GB per layer at the 1 M-edge config — never exists: a wave keeps 16 edges on its
// lanes, walks the D orbitals of those edges, runs both layers register to register on the matrix cores
// and folds rb[e,d] * SiLU(.) into per-l accumulators A[e, l, :] that are written once.
//   HBM per edge: read 4*D*C (cst) + 4*D (rb); write 4*(n_l+1)*V.   FLOPs per edge: 2*D*(C*V + V*V).
// LDS: W1 (V x (C+4)) and W2 (V x (V+4)) fp32, staged once per persistent workgroup.
// ===================================================================================================
template <int C, int V, int NL>
__global__ __launch_bounds__(256) void k_coeff_mlp(const float* __restrict__ cst, const float* __restrict__ rb,
                                                   const int32_t* __restrict__ order,
                                                   const int32_t* __restrict__ l_of_d,
                                                   const float* __restrict__ wmul /* (E,D) or null */,
                                                   const float* __restrict__ W1, const float* __restrict__ W2,
                                                   int64_t E, int D, int accumulate, float* __restrict__ A,
                                                   float* __restrict__ Bout) {
  constexpr int KT1 = C / 16, MT = V / 16;
  extern __shared__ float lds[];
  float* w1s = lds;
  float* w2s = lds + V * (C + 4);
  stage_weights(w1s, W1, V, C, C);
  stage_weights(w2s, W2, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const int64_t e = order ? order[pc] : pc;
    const float* xrow = cst + e * (int64_t)D * C + 4 * q;
    const float* rbrow = rb + e * D;
    f32x4v acc[NL][MT];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[l][m] = f32x4v{0.f, 0.f, 0.f, 0.f};

    f32x4v x[KT1];
#pragma unroll
    for (int a = 0; a < KT1; ++a) x[a] = *(const f32x4v*)(xrow + 16 * a);

    for (int d = 0; d < D; ++d) {
      // the weight fragments are loop-invariant LDS loads: without this fence LICM hoists all of them
      // (2 x 64 ds_read_b128 = 512 VGPRs) out of the orbital loop and the kernel spills.
      asm volatile("" ::: "memory");
      f32x4v h[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) h[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
      mfma_layer<KT1, MT>(w1s, C + 4, r, q, x, h);
      // prefetch the next orbital's coefficient row while layer 2 runs
      if (d + 1 < D) {
#pragma unroll
        for (int a = 0; a < KT1; ++a) x[a] = *(const f32x4v*)(xrow + (d + 1) * C + 16 * a);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) h[m][j] = lcao_silu(h[m][j]);
      f32x4v o[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) o[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
      mfma_layer<MT, MT>(w2s, V + 4, r, q, h, o);
      float w = rbrow[d];
      if (wmul) w *= wmul[e * D + d];
      const int l = l_of_d[d];  // wave-uniform
#pragma unroll
      for (int ll = 0; ll < NL; ++ll) {
        if (l == ll) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ll][m][j] += w * lcao_silu(o[m][j]);
        }
      }
    }
    if (valid) {
      float* Ap = A + p * (int64_t)NL * V + 4 * q;
      float* Bp = Bout + p * (int64_t)V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4v bsum = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          bsum += acc[l][m];
          f32x4v v = acc[l][m];
          if (accumulate) v += *(const f32x4v*)(Ap + l * V + 16 * m);
          *(f32x4v*)(Ap + l * V + 16 * m) = v;
        }
        *(f32x4v*)(Bp + 16 * m) = bsum;
      }
    }
  }
}

template <int C, int V, int NL>
static int launch_coeff_mlp(const float* cst, const float* rb, const int32_t* order, const int32_t* l_of_d,
                            const float* wmul, const float* W1, const float* W2, int64_t E, int D, int accumulate,
                            float* A, float* Bout, hipStream_t st) {
  size_t lds = (size_t(V) * (C + 4) + size_t(V) * (V + 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_coeff_mlp<C, V, NL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return int(e);
    attr_set = true;
  }
  int64_t tiles = (E + 15) / 16;
  int64_t groups = (tiles + 3) / 4;
  int64_t per_cu = (lds > 80 * 1024) ? 1 : 2;
  int64_t cap = (int64_t)lcao_num_cu() * per_cu;
  if (groups > cap) groups = cap;
  hipLaunchKernelGGL((k_coeff_mlp<C, V, NL>), dim3((unsigned)groups), dim3(256), lds, st, cst, rb, order, l_of_d,
                     wmul, W1, W2, E, D, accumulate, A, Bout);
  return lcao_launch_status();
}

template <int C, int V>
static int dispatch_coeff_mlp_nl(int n_l, const float* cst, const float* rb, const int32_t* order,
                                 const int32_t* l_of_d, const float* wmul, const float* W1, const float* W2,
                                 int64_t E, int D, int accumulate, float* A, float* Bout, hipStream_t st) {
  switch (n_l) {
    case 1: return launch_coeff_mlp<C, V, 1>(cst, rb, order, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, st);
    case 2: return launch_coeff_mlp<C, V, 2>(cst, rb, order, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, st);
    case 3: return launch_coeff_mlp<C, V, 3>(cst, rb, order, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, st);
    case 4: return launch_coeff_mlp<C, V, 4>(cst, rb, order, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, st);
    default: return LCAO_ERR_INVALID;
  }
}

extern "C" size_t lcao_lds_bytes_coeff_mlp(int C, int V, int valence) {
  (void)valence;
  return (size_t(V) * (C + 4) + size_t(V) * (V + 4)) * sizeof(float);
}

extern "C" int lcao_coeff_mlp_contract(const float* cst, const float* rb, const int32_t* order,
                                       const int32_t* l_of_d, const float* valence_ed, const float* W1,
                                       const float* W2, int64_t n_edges, int n_orb, int C, int V, int n_l,
                                       int valence, float* A, float* B, float* Bv, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || C <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(cst); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(W1); LCAO_CHECK_PTR(W2);
  LCAO_CHECK_PTR(A); LCAO_CHECK_PTR(B);
  if (valence) { LCAO_CHECK_PTR(valence_ed); LCAO_CHECK_PTR(Bv); }
  hipStream_t st = (hipStream_t)stream;
  int rc = LCAO_ERR_UNSUPPORTED;
#define CM_CASE(C_, V_)                                                                                          \
  if (C == C_ && V == V_) {                                                                                      \
    rc = dispatch_coeff_mlp_nl<C_, V_>(n_l, cst, rb, order, l_of_d, nullptr, W1, W2, n_edges, n_orb, 0, A, B, st); \
    if (rc == LCAO_OK && valence)                                                                                \
      rc = dispatch_coeff_mlp_nl<C_, V_>(n_l, cst, rb, order, l_of_d, valence_ed, W1, W2 + (size_t)V_ * V_,       \
                                         n_edges, n_orb, 1, A, Bv, st);                                          \
    return rc;                                                                                                   \
  }
  CM_CASE(128, 128)
  CM_CASE(64, 64)
  CM_CASE(32, 32)
#undef CM_CASE
  return rc;
}

// ===================================================================================================
// lcao_edge_lcao_weight — lcaonet.py:192 (f_three), :192-204 (inject + contract + F.normalize) and the
// basis_weight product of :209, per 16-edge tile, register to register:
//   ft = W3 . tbw^T ;  u = B * (1 + ft) ;  lw = u / max(|u|, 1e-12) ;  bw = Wb . lw^T
// The row norm is an in-lane sum over the lane's 4*MT features plus two cross-quarter shuffles.
// ===================================================================================================
template <int V>
__global__ __launch_bounds__(256) void k_edge_lcao_weight(const float* __restrict__ tbw, const float* __restrict__ B,
                                                          const float* __restrict__ W3, const float* __restrict__ Wb,
                                                          int64_t E, float* __restrict__ bw) {
  constexpr int MT = V / 16;
  extern __shared__ float lds[];
  float* w3s = lds;
  float* wbs = lds + V * (V + 4);
  stage_weights(w3s, W3, V, V, V);
  stage_weights(wbs, Wb, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");  // keep the LDS weight fragments from being hoisted out of the tile loop
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    f32x4v x[MT], b[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) x[a] = *(const f32x4v*)(tbw + pc * V + 16 * a + 4 * q);
#pragma unroll
    for (int a = 0; a < MT; ++a) b[a] = *(const f32x4v*)(B + pc * V + 16 * a + 4 * q);
    f32x4v ft[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ft[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<MT, MT>(w3s, V + 4, r, q, x, ft);
    float n2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = b[m][j] + b[m][j] * ft[m][j];
        ft[m][j] = u;
        n2 += u * u;
      }
    n2 += __shfl_xor(n2, 16);
    n2 += __shfl_xor(n2, 32);
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
    for (int m = 0; m < MT; ++m) ft[m] *= inv;
    f32x4v o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<MT, MT>(wbs, V + 4, r, q, ft, o);
    if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(bw + p * V + 16 * m + 4 * q) = o[m];
    }
  }
}

// ===================================================================================================
// lcao_edge_node_message — second f_node layer (lcaonet.py:125-126 via :209) on SiLU(Ps[s]+Pt[t]) and the
// product with the basis-weighted LCAO weight: msg = bw * SiLU(Wn2 . h1^T + bn2).
// Ps / Pt are (N, V) node tables (32 MB at the 1 M-edge config: L2 / Infinity-Cache resident row gathers).
// ===================================================================================================
template <int V>
__global__ __launch_bounds__(256) void k_edge_node_message(const float* __restrict__ bw, const float* __restrict__ Ps,
                                                           const float* __restrict__ Pt,
                                                           const int32_t* __restrict__ src_p,
                                                           const int32_t* __restrict__ dst_p,
                                                           const float* __restrict__ Wn2,
                                                           const float* __restrict__ bn2, int64_t E,
                                                           float* __restrict__ msg) {
  constexpr int MT = V / 16;
  extern __shared__ float lds[];
  stage_weights(lds, Wn2, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const float* ps = Ps + (int64_t)src_p[pc] * V + 4 * q;
    const float* pt = Pt + (int64_t)dst_p[pc] * V + 4 * q;
    f32x4v h[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      f32x4v s = *(const f32x4v*)(ps + 16 * a) + *(const f32x4v*)(pt + 16 * a);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[a][j] = lcao_silu(s[j]);
    }
    f32x4v o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);  // bias as the initial accumulator
    mfma_layer<MT, MT>(lds, V + 4, r, q, h, o);
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

template <typename KernelT>
static int set_lds_attr(KernelT kern, size_t lds, bool& done) {
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return int(e);
    done = true;
  }
  return LCAO_OK;
}

static unsigned persistent_groups(int64_t n_rows, size_t lds) {
  int64_t tiles = (n_rows + 15) / 16;
  int64_t groups = (tiles + 3) / 4;
  int64_t per_cu = (lds > 80 * 1024) ? 1 : ((lds > 53 * 1024) ? 2 : 3);
  int64_t cap = (int64_t)lcao_num_cu() * per_cu;
  return unsigned(groups > cap ? cap : groups);
}

extern "C" int lcao_edge_lcao_weight(const float* tbw, const float* B, const float* W3, const float* Wb,
                                     int64_t n_edges, int V, float* bw, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(tbw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(W3); LCAO_CHECK_PTR(Wb); LCAO_CHECK_PTR(bw);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(2) * V * (V + 4) * sizeof(float);
#define LW_CASE(V_)                                                                                          \
  if (V == V_) {                                                                                             \
    static bool done = false;                                                                                \
    int rc = set_lds_attr(k_edge_lcao_weight<V_>, lds, done);                                                \
    if (rc != LCAO_OK) return rc;                                                                            \
    hipLaunchKernelGGL(k_edge_lcao_weight<V_>, dim3(persistent_groups(n_edges, lds)), dim3(256), lds, st, tbw, B, \
                       W3, Wb, n_edges, bw);                                                                 \
    return lcao_launch_status();                                                                             \
  }
  LW_CASE(128) LW_CASE(64) LW_CASE(32)
#undef LW_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_edge_node_message(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                      const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges,
                                      int V, float* msg, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(bw); LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p);
  LCAO_CHECK_PTR(Wn2); LCAO_CHECK_PTR(bn2); LCAO_CHECK_PTR(msg);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(V) * (V + 4) * sizeof(float);
#define NM_CASE(V_)                                                                                            \
  if (V == V_) {                                                                                               \
    static bool done = false;                                                                                  \
    int rc = set_lds_attr(k_edge_node_message<V_>, lds, done);                                                 \
    if (rc != LCAO_OK) return rc;                                                                              \
    hipLaunchKernelGGL(k_edge_node_message<V_>, dim3(persistent_groups(n_edges, lds)), dim3(256), lds, st, bw, Ps, \
                       Pt, src_p, dst_p, Wn2, bn2, n_edges, msg);                                              \
    return lcao_launch_status();                                                                               \
  }
  NM_CASE(128) NM_CASE(64) NM_CASE(32)
#undef NM_CASE
  return LCAO_ERR_UNSUPPORTED;
}
