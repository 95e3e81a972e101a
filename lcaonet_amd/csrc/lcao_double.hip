// lcao_double.hip — the three-body term as a sparse bilinear family that is closed under differentiation, for the
// EXACT double backward of autograd-force training (lcaonet/model/lcaonet.py:310-317, create_graph=True).
//
// lcaonet.py:173-189 per centre node s:   tbw[p, :] = sum_{q in In(s)} sum_l  w[t(p,q), l] * Ag[q, l, :]
// with one scalar weight vector w[t, :] (n_l floats) per triplet t = (out-edge p, in-edge q).  In the inference kernels
// w = Y_l(u_p . u_q) / |.| is recomputed in registers; here it is an INPUT (a T-sized tensor of scalars built by
// ordinary differentiable tensor ops, model/double.py), which makes the op bilinear in (w, Ag) and gives the family
//     spmm  (w, A) -> out[p,:]   = sum_{q,l} w[t,l] A[q,l,:]
//     spmm_t(w, G) -> out[q,l,:] = sum_p     w[t,l] G[p,:]
//     sddmm (G, A) -> out[t,l]   = G[p,:] . A[q,l,:]
// whose derivatives are each other:  d spmm/dw = sddmm(g, A), d spmm/dA = spmm_t(w, g);  d spmm_t/dw = sddmm(G, g),
// d spmm_t/dG = spmm(w, g);  d sddmm/dG = spmm(g, A), d sddmm/dA = spmm_t(g, G).  So autograd can differentiate the
// backward pass again without any further kernel.
//
// Triplet layout ("dense per-node blocks"): node s owns the block tbase[s] + i * m_in(s) + j, i = rank of the out-edge
// p = out_ptr[s] + i, j = rank of the in-edge q = in_pos[in_ptr[s] + j].  The pair q == p (dropped by lcaonet.py:470) has
// a slot too; the caller gives it w = 0.  Any V <= 512; no atomics.
//
// Per centre node the three ops are small dense products (m_out x m_in*n_l x V): for V % 4 == 0 one wave takes a node and
// runs them on fp32 MFMA 16x16x4 tiles with both operands read straight from global memory as float4 (k_trip_*_node; the
// rows of a node are L2-resident after their first touch).  The lane-per-channel kernels below them serve any other V.
#include "lcao_mfma.h"

template <int R>
__global__ void k_trip_spmm(const float* __restrict__ w, const float* __restrict__ A, const int32_t* __restrict__ src_p,
                            const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_pos,
                            const int32_t* __restrict__ out_ptr, const int64_t* __restrict__ tbase, int64_t n_edges,
                            int V, int NL, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; p < n_edges; p += nwave) {
    const int s = src_p[p];
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    const int64_t t0 = tbase[s] + (int64_t)(p - out_ptr[s]) * m_in;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int j = 0; j < m_in; ++j) {
      const int64_t q = in_pos[j0 + j];
      for (int l = 0; l < NL; ++l) {
        const float wv = w[(t0 + j) * NL + l];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          if (ch < V) acc[r] += wv * A[(q * NL + l) * V + ch];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      if (ch < V) out[p * V + ch] = acc[r];
    }
  }
}

// one wave per in-edge q (of node s = dst_p[q], rank j = in_rank[q]); output row (q, l, :) for every l
template <int R, int NLMAX>
__global__ void k_trip_spmm_t(const float* __restrict__ w, const float* __restrict__ G, const int32_t* __restrict__ dst_p,
                              const int32_t* __restrict__ in_rank, const int32_t* __restrict__ in_ptr,
                              const int32_t* __restrict__ out_ptr, const int64_t* __restrict__ tbase, int64_t n_edges,
                              int n_centres, int V, int NL, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t q = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; q < n_edges; q += nwave) {
    const int s = dst_p[q];
    float acc[NLMAX][R];
#pragma unroll
    for (int l = 0; l < NLMAX; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[l][r] = 0.f;
    if (s < n_centres) {   // in-edges of nodes that are not centres here (ghosts of a sharded graph) get zero rows
      const int m_in = in_ptr[s + 1] - in_ptr[s];
      const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
      const int64_t tq = tbase[s] + in_rank[q];
      for (int i = 0; i < m_out; ++i) {
        const int64_t t = tq + (int64_t)i * m_in;
        float g[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          g[r] = ch < V ? G[(int64_t)(p0 + i) * V + ch] : 0.f;
        }
#pragma unroll
        for (int l = 0; l < NLMAX; ++l) {
          if (l < NL) {
            const float wv = w[t * NL + l];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[l][r] += wv * g[r];
          }
        }
      }
    }
#pragma unroll
    for (int l = 0; l < NLMAX; ++l)
      if (l < NL) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          if (ch < V) out[(q * NL + l) * V + ch] = acc[l][r];
        }
      }
  }
}

template <int R>
__global__ void k_trip_sddmm(const float* __restrict__ G, const float* __restrict__ A, const int32_t* __restrict__ src_p,
                             const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_pos,
                             const int32_t* __restrict__ out_ptr, const int64_t* __restrict__ tbase, int64_t n_edges,
                             int V, int NL, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; p < n_edges; p += nwave) {
    const int s = src_p[p];
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    const int64_t t0 = tbase[s] + (int64_t)(p - out_ptr[s]) * m_in;
    float g[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      g[r] = ch < V ? G[p * V + ch] : 0.f;
    }
    for (int j = 0; j < m_in; ++j) {
      const int64_t q = in_pos[j0 + j];
      for (int l = 0; l < NL; ++l) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          if (ch < V) d += g[r] * A[(q * NL + l) * V + ch];
        }
        d = lcao_wave_sum(d);
        if (lane == 0) out[(t0 + j) * NL + l] = d;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Node-blocked MFMA versions.  v_mfma_f32_16x16x4_f32: A operand lane (r, q) = A[r][q], B operand lane (c, q) = B[q][c],
// accumulator lane (c, q)[j] = D[4q + j][c]  (r = c = lane & 15, q = lane >> 4).  The memory-contiguous output dimension
// is always the MFMA row dimension, so a lane's 4 accumulator rows (or the same row of 4 tiles) are one float4 store.
// Rows / columns past the node's extents are clamped duplicates (computed, never stored); only the K dimension is
// zero-padded.
// ---------------------------------------------------------------------------------------------------------------------
// sddmm: D[m = (j,l)][i] = sum_v A[q_j, l, v] G[p_i, v];  k = 16 kk + 4 q + c for MFMA step c of chunk kk (both operands)
template <int NL>
__global__ __launch_bounds__(256) void k_trip_sddmm_node(const float* __restrict__ G, const float* __restrict__ A,
                                                         const int32_t* __restrict__ in_ptr,
                                                         const int32_t* __restrict__ in_pos,
                                                         const int32_t* __restrict__ out_ptr,
                                                         const int64_t* __restrict__ tbase, int n_nodes, int V,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  const int KC = (V + 15) >> 4;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    if (m_out <= 0 || m_in <= 0) continue;
    const int Mtot = m_in * NL;
    const int64_t t0 = tbase[s];
    for (int i0 = 0; i0 < m_out; i0 += 16) {
      const float* gb = G + (int64_t)(p0 + min(i0 + r, m_out - 1)) * V + 4 * q;
      for (int m0 = 0; m0 < Mtot; m0 += 64) {
        const int ntile = min(4, (Mtot - m0 + 15) >> 4);
        const float* ab[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = min(m0 + 16 * t + r, Mtot - 1);
          ab[t] = A + ((int64_t)in_pos[j0 + m / NL] * NL + m % NL) * V + 4 * q;
        }
        f32x4v acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < KC; ++kk) {
          const bool ok = 16 * kk + 4 * q < V;
          const f32x4v z = {0.f, 0.f, 0.f, 0.f};
          const f32x4v b = ok ? *(const f32x4v*)(gb + 16 * kk) : z;
          f32x4v a[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) a[t] = (ok && t < ntile) ? *(const f32x4v*)(ab[t] + 16 * kk) : z;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (t < ntile) acc[t] = MFMA16(a[t][c], b[c], acc[t]);
        }
        if (i0 + r < m_out) {
          float* o = out + (t0 + (int64_t)(i0 + r) * m_in) * NL;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int m = m0 + 16 * t + 4 * q;
            if (t < ntile) {
              if (NL == 4) {
                if (m < Mtot) *(f32x4v*)(o + m) = acc[t];
              } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                  if (m + jj < Mtot) o[m + jj] = acc[t][jj];
              }
            }
          }
        }
      }
    }
  }
}

// spmm: D_cc[4r' .. ][i] with row r of tile cc <-> channel v0 + 4 r + cc;  k = (j,l) = 16 kk + 4 q + c
template <int NL>
__global__ __launch_bounds__(256) void k_trip_spmm_node(const float* __restrict__ w, const float* __restrict__ A,
                                                        const int32_t* __restrict__ in_ptr,
                                                        const int32_t* __restrict__ in_pos,
                                                        const int32_t* __restrict__ out_ptr,
                                                        const int64_t* __restrict__ tbase, int n_nodes, int V,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    if (m_out <= 0) continue;
    const int Ktot = m_in * NL, KC = (Ktot + 15) >> 4;
    const int64_t t0 = tbase[s];
    for (int i0 = 0; i0 < m_out; i0 += 16) {
      const float* wb = w + (t0 + (int64_t)min(i0 + r, m_out - 1) * m_in) * NL;
      for (int v0 = 0; v0 < V; v0 += 64) {
        const int vld = (v0 + 4 * r < V) ? v0 + 4 * r : 0;
        f32x4v acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < KC; ++kk) {
          const int k = 16 * kk + 4 * q;
          f32x4v b, a[4];
          if (NL == 4) {
            b = k < Ktot ? *(const f32x4v*)(wb + k) : f32x4v{0.f, 0.f, 0.f, 0.f};
            const float* ar = A + (int64_t)in_pos[j0 + min(k >> 2, m_in - 1)] * 4 * V + vld;
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = *(const f32x4v*)(ar + c * V);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              b[c] = k + c < Ktot ? wb[k + c] : 0.f;
              const int kc = min(k + c, Ktot - 1);
              a[c] = *(const f32x4v*)(A + ((int64_t)in_pos[j0 + kc / NL] * NL + kc % NL) * V + vld);
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MFMA16(a[c][t], b[c], acc[t]);
        }
        if (i0 + r < m_out) {
          float* o = out + (int64_t)(p0 + i0 + r) * V;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int v = v0 + 16 * q + 4 * jj;
            if (v < V) *(f32x4v*)(o + v) = f32x4v{acc[0][jj], acc[1][jj], acc[2][jj], acc[3][jj]};
          }
        }
      }
    }
  }
}

// spmm_t: D_cc[v][n = (j,l)] = sum_i G[p_i, v] w[t(i,j), l];  k = i = 4 ks + q (no permutation: K is the short out-list)
template <int NL>
__global__ __launch_bounds__(256) void k_trip_spmm_t_node(const float* __restrict__ w, const float* __restrict__ G,
                                                          const int32_t* __restrict__ in_ptr,
                                                          const int32_t* __restrict__ in_pos,
                                                          const int32_t* __restrict__ out_ptr,
                                                          const int64_t* __restrict__ tbase, int n_nodes, int V,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    if (m_in <= 0) continue;
    const int Ntot = m_in * NL, KS = (m_out + 3) >> 2;
    const int64_t t0 = tbase[s];
    for (int n0 = 0; n0 < Ntot; n0 += 16) {
      const int n = min(n0 + r, Ntot - 1);
      const float* wb = w + t0 * NL + n;
      const int64_t orow = (int64_t)in_pos[j0 + n / NL] * NL + n % NL;
      for (int v0 = 0; v0 < V; v0 += 64) {
        const int vld = (v0 + 4 * r < V) ? v0 + 4 * r : 0;
        f32x4v acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
          const int i = 4 * ks + q;
          const float b = i < m_out ? wb[(int64_t)i * Ntot] : 0.f;
          const f32x4v a = *(const f32x4v*)(G + (int64_t)(p0 + min(i, max(m_out - 1, 0))) * V + vld);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = MFMA16(a[t], b, acc[t]);
        }
        if (n0 + r < Ntot) {
          float* o = out + orow * V;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int v = v0 + 16 * q + 4 * jj;
            if (v < V) *(f32x4v*)(o + v) = f32x4v{acc[0][jj], acc[1][jj], acc[2][jj], acc[3][jj]};
          }
        }
      }
    }
  }
}

// rows of in-edges whose target is not a centre here (ghosts of a sharded graph) are zero
__global__ void k_zero_noncentre_rows(const int32_t* __restrict__ dst_p, int64_t n_edges, int n_centres, int row_len,
                                      float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; e < n_edges; e += nwave)
    if (dst_p[e] >= n_centres)
      for (int i = lane; i < row_len; i += 64) out[e * row_len + i] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// The per-edge Gram matrix G[e] = A[e] A[e]^T (n_l x n_l, lcaonet.py:179-184 in Gram form) as a second closed family:
//     rowgram(X, Y) -> out[e,l,m] = sum_v X[e,l,v] Y[e,m,v]
//     rowmix (S, A) -> out[e,l,v] = sum_m S[e,l,m] A[e,m,v]          (or S[e,m,l]: `trans`)
// d rowgram/dX = rowmix(g, Y), d rowgram/dY = rowmix(g^T, X);  d rowmix/dS = rowgram(g, A) (or rowgram(A, g)),
// d rowmix/dA = rowmix(S^T, g).  One wave per edge, lanes over channels; both stream their operands once (HBM-bound).
// ---------------------------------------------------------------------------------------------------------------------
template <int R, int NLMAX>
__global__ void k_rowgram(const float* __restrict__ X, const float* __restrict__ Y, int64_t n_edges, int V, int NL,
                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; e < n_edges; e += nwave) {
    float x[NLMAX][R], y[NLMAX][R];
#pragma unroll
    for (int l = 0; l < NLMAX; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + 64 * r;
        const bool ok = l < NL && ch < V;
        x[l][r] = ok ? X[(e * NL + l) * V + ch] : 0.f;
        y[l][r] = ok ? Y[(e * NL + l) * V + ch] : 0.f;
      }
    // 16 partial dot products per lane -> row sums by DPP -> lane (row k, r) keeps product r -> two cross-row steps
    float mine = 0.f;
#pragma unroll
    for (int l = 0; l < NLMAX; ++l)
#pragma unroll
      for (int m = 0; m < NLMAX; ++m) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) d += x[l][r] * y[m][r];
        d = lcao_row16_sum(d);
        if ((lane & 15) == l * NLMAX + m) mine = d;
      }
    mine += __shfl_xor(mine, 16);
    mine += __shfl_xor(mine, 32);
    const int l = (lane & 15) / NLMAX, m = (lane & 15) % NLMAX;
    if (lane < 16 && l < NL && m < NL) out[(e * NL + l) * NL + m] = mine;
  }
}

template <int R, int NLMAX>
__global__ void k_rowmix(const float* __restrict__ S, const float* __restrict__ A, int64_t n_edges, int V, int NL,
                         int trans, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; e < n_edges; e += nwave) {
    float a[NLMAX][R];
#pragma unroll
    for (int m = 0; m < NLMAX; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + 64 * r;
        a[m][r] = (m < NL && ch < V) ? A[(e * NL + m) * V + ch] : 0.f;
      }
    const float* s = S + e * NL * NL;
#pragma unroll
    for (int l = 0; l < NLMAX; ++l)
      if (l < NL) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
        for (int m = 0; m < NLMAX; ++m)
          if (m < NL) {
            const float sv = trans ? s[m * NL + l] : s[l * NL + m];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] += sv * a[m][r];
          }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          if (ch < V) out[(e * NL + l) * V + ch] = acc[r];
        }
      }
  }
}

static int check_trip(int64_t n_edges, int V, int n_l) {
  if (n_edges < 0 || n_edges >= (int64_t(1) << 31)) return LCAO_ERR_INVALID;
  if (V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  return LCAO_OK;
}

#define TRIP_DISPATCH(KERNEL, ...)                                                        \
  do {                                                                                    \
    const int R_ = (V + 63) / 64;                                                         \
    dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);                                \
    if (R_ == 1) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, st, __VA_ARGS__);        \
    else if (R_ == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, st, __VA_ARGS__);   \
    else if (R_ <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, st, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<8>), grid, block, 0, st, __VA_ARGS__);                \
  } while (0)

#define NODE_DISPATCH(KERNEL, ...)                                                                     \
  do {                                                                                                 \
    dim3 grid(lcao_blocks(n_centres, 4, 65536)), block(256);                                           \
    if (n_l == 1) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, st, __VA_ARGS__);                    \
    else if (n_l == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, st, __VA_ARGS__);               \
    else if (n_l == 3) hipLaunchKernelGGL((KERNEL<3>), grid, block, 0, st, __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, st, __VA_ARGS__);                             \
  } while (0)

static int check_centres(int64_t n_centres) {
  return (n_centres < 0 || n_centres >= (int64_t(1) << 31)) ? LCAO_ERR_INVALID : LCAO_OK;
}

extern "C" int lcao_trip_spmm(const float* w, const float* A, const int32_t* src_p, const int32_t* in_ptr,
                              const int32_t* in_pos, const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges,
                              int64_t n_centres, int V, int n_l, float* out, void* stream) {
  int rc = check_trip(n_edges, V, n_l);
  if (rc != LCAO_OK) return rc;
  if ((rc = check_centres(n_centres)) != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(w); LCAO_CHECK_PTR(A); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(in_ptr); LCAO_CHECK_PTR(in_pos);
  LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  if (V % 4 == 0 && n_centres > 0)
    NODE_DISPATCH(k_trip_spmm_node, w, A, in_ptr, in_pos, out_ptr, tbase, (int)n_centres, V, out);
  else
    TRIP_DISPATCH(k_trip_spmm, w, A, src_p, in_ptr, in_pos, out_ptr, tbase, n_edges, V, n_l, out);
  return lcao_launch_status();
}

extern "C" int lcao_trip_sddmm(const float* G, const float* A, const int32_t* src_p, const int32_t* in_ptr,
                               const int32_t* in_pos, const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges,
                               int64_t n_centres, int V, int n_l, float* out, void* stream) {
  int rc = check_trip(n_edges, V, n_l);
  if (rc != LCAO_OK) return rc;
  if ((rc = check_centres(n_centres)) != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(A); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(in_ptr); LCAO_CHECK_PTR(in_pos);
  LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  if (V % 4 == 0 && n_centres > 0)
    NODE_DISPATCH(k_trip_sddmm_node, G, A, in_ptr, in_pos, out_ptr, tbase, (int)n_centres, V, out);
  else
    TRIP_DISPATCH(k_trip_sddmm, G, A, src_p, in_ptr, in_pos, out_ptr, tbase, n_edges, V, n_l, out);
  return lcao_launch_status();
}

extern "C" int lcao_trip_spmm_t(const float* w, const float* G, const int32_t* dst_p, const int32_t* in_rank,
                                const int32_t* in_ptr, const int32_t* in_pos, const int32_t* out_ptr, const int64_t* tbase,
                                int64_t n_edges, int64_t n_centres, int V, int n_l, float* out, void* stream) {
  int rc = check_trip(n_edges, V, n_l);
  if (rc != LCAO_OK) return rc;
  if ((rc = check_centres(n_centres)) != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(w); LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(in_rank); LCAO_CHECK_PTR(in_ptr);
  LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  if (V % 4 == 0 && n_centres > 0) {
    hipLaunchKernelGGL(k_zero_noncentre_rows, dim3(lcao_blocks(n_edges, 4, 65536)), dim3(256), 0, st, dst_p, n_edges,
                       (int)n_centres, n_l * V, out);
    NODE_DISPATCH(k_trip_spmm_t_node, w, G, in_ptr, in_pos, out_ptr, tbase, (int)n_centres, V, out);
    return lcao_launch_status();
  }
  const int R_ = (V + 63) / 64;
  dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);
#define TT(RR) \
  hipLaunchKernelGGL((k_trip_spmm_t<RR, 4>), grid, block, 0, st, w, G, dst_p, in_rank, in_ptr, out_ptr, tbase, n_edges, \
                     (int)n_centres, V, n_l, out)
  if (R_ == 1) TT(1);
  else if (R_ == 2) TT(2);
  else if (R_ <= 4) TT(4);
  else TT(8);
#undef TT
  return lcao_launch_status();
}

extern "C" int lcao_rowgram(const float* X, const float* Y, int64_t n_edges, int V, int n_l, float* out, void* stream) {
  int rc = check_trip(n_edges, V, n_l);
  if (rc != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(X); LCAO_CHECK_PTR(Y); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  const int R_ = (V + 63) / 64;
  dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);
#define GG(RR) hipLaunchKernelGGL((k_rowgram<RR, 4>), grid, block, 0, st, X, Y, n_edges, V, n_l, out)
  if (R_ == 1) GG(1);
  else if (R_ == 2) GG(2);
  else if (R_ <= 4) GG(4);
  else GG(8);
#undef GG
  return lcao_launch_status();
}

extern "C" int lcao_rowmix(const float* S, const float* A, int64_t n_edges, int V, int n_l, int trans, float* out,
                           void* stream) {
  int rc = check_trip(n_edges, V, n_l);
  if (rc != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(S); LCAO_CHECK_PTR(A); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  const int R_ = (V + 63) / 64;
  dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);
#define MM(RR) hipLaunchKernelGGL((k_rowmix<RR, 4>), grid, block, 0, st, S, A, n_edges, V, n_l, trans, out)
  if (R_ == 1) MM(1);
  else if (R_ == 2) MM(2);
  else if (R_ <= 4) MM(4);
  else MM(8);
#undef MM
  return lcao_launch_status();
}
