// lcao_node.hip — node-level (N rows, not E) work of one LCAOConv layer in ONE launch.
//
// lcaonet.py:167 (node_weight), :187 (sigmoid(xk)) and the first f_node layer of :209 split by the identity
//   Dense(cat[x_s, x_t]) = x_s Wa^T + x_t Wb^T   produce three (N, V) node tables per layer:
//     gate = sigmoid(x Wk^T + bk)
//     Ps   = (x Ws^T + bs) Wa^T + b1 = x (Wa Ws)^T + (Wa bs + b1)
//     Pt   = (x Ws^T + bs) Wb^T      = x (Wb Ws)^T +  Wb bs
// The host folds the two linear maps (weights only, cached per layer), so all three tables are one GEMM of x against
// the stacked (3V x H) matrix [Wk; Wa Ws; Wb Ws]: one read of x per 64-column slice (x stays in L2 across the six
// slices), one launch instead of four.  Same tile formulation as lcao_dense (lcao_mfma.h).
#include "lcao_mfma.h"

template <int KT>
__global__ __launch_bounds__(256) void k_node_tables(const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ bias, int64_t M, int V,
                                                     float* __restrict__ gate, float* __restrict__ Ps,
                                                     float* __restrict__ Pt) {
  constexpr int K = KT * 16;
  extern __shared__ float lds[];
  const int N = 3 * V;
  const int n0 = blockIdx.x * 64;
  const int ncols = min(64, N - n0);  // multiple of 32 (V % 32 == 0)
  stage_weights(lds, W + (int64_t)n0 * K, ncols, K, K);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (M + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.y * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.y * 4) {
    asm volatile("" ::: "memory");
    const int64_t row = tile * 16 + r;
    const bool valid = row < M;
    const float* xr = X + (valid ? row : M - 1) * (int64_t)K + 4 * q;
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
          const int n = n0 + 16 * t + 4 * q;       // a 16-column tile never straddles two tables (V % 16 == 0)
          const int which = n / V, col = n - which * V;
          f32x4v v = acc[t] + *(const f32x4v*)(bias + n);
          if (which == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = lcao_sigmoid(v[j]);
          }
          float* out = which == 0 ? gate : (which == 1 ? Ps : Pt);
          if (out) *(f32x4v*)(out + row * V + col) = v;
        }
      }
    }
  }
}

template <int KT>
static int launch_node_tables(const float* X, const float* W, const float* bias, int64_t M, int V, float* gate, float* Ps,
                              float* Pt, hipStream_t st) {
  constexpr int K = KT * 16;
  const size_t lds = size_t(64) * (K + 4) * sizeof(float);
  static bool done = false;
  if (lds > 48 * 1024) {
    int rc = set_lds_attr(k_node_tables<KT>, lds, done);
    if (rc != LCAO_OK) return rc;
  }
  const int nslice = (3 * V + 63) / 64;
  int64_t groups = ((M + 15) / 16 + 3) / 4;
  int64_t cap = (int64_t)lcao_num_cu() * 4 / nslice;
  if (cap < 1) cap = 1;
  if (groups > cap) groups = cap;
  hipLaunchKernelGGL(k_node_tables<KT>, dim3(nslice, (unsigned)groups), dim3(256), lds, st, X, W, bias, M, V, gate, Ps, Pt);
  return lcao_launch_status();
}

extern "C" int lcao_node_tables(const float* x, const float* w_cat, const float* b_cat, int64_t n_rows, int H, int V,
                                float* gate, float* Ps, float* Pt, void* stream) {
  LCAO_CHECK_SIZE(n_rows);
  if (H <= 0 || V <= 0) return LCAO_ERR_INVALID;
  if (H % 16 || V % 32 || H > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_rows == 0) return LCAO_OK;
  LCAO_CHECK_PTR(x); LCAO_CHECK_PTR(w_cat); LCAO_CHECK_PTR(b_cat);
  if (((uintptr_t)x | (uintptr_t)w_cat | (uintptr_t)b_cat | (uintptr_t)gate | (uintptr_t)Ps | (uintptr_t)Pt) % 16)
    return LCAO_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
#define NT_CASE(KT_) \
  case KT_:          \
    return launch_node_tables<KT_>(x, w_cat, b_cat, n_rows, V, gate, Ps, Pt, st)
  switch (H / 16) {
    NT_CASE(1); NT_CASE(2); NT_CASE(3); NT_CASE(4); NT_CASE(6); NT_CASE(8); NT_CASE(12); NT_CASE(16); NT_CASE(24);
    NT_CASE(32);
    default: break;
  }
#undef NT_CASE
  return LCAO_ERR_UNSUPPORTED;
}

// row of the (zdim^2, D, C) species-pair table read by edge p: z[src_p[p]] * zdim + z[dst_p[p]]  (cst_row of
// lcao_coeff_mlp_contract; embed.py:246-249: cst[e] is a function of the two atomic numbers alone)
__global__ void k_pair_rows(const int64_t* __restrict__ z, const int32_t* __restrict__ src_p,
                            const int32_t* __restrict__ dst_p, int64_t E, int zdim, int32_t* __restrict__ row) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p < E) row[p] = (int32_t)z[src_p[p]] * zdim + (int32_t)z[dst_p[p]];
}

extern "C" int lcao_pair_rows(const int64_t* z, const int32_t* src_p, const int32_t* dst_p, int64_t n_edges, int zdim,
                              int32_t* row, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (zdim <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(z); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(row);
  hipLaunchKernelGGL(k_pair_rows, dim3(lcao_blocks(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, z, src_p, dst_p,
                     n_edges, zdim, row);
  return lcao_launch_status();
}
