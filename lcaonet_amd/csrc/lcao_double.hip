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
// Common shape of the three: the out-list of a node (m_out rows, 18 in a crystal) is covered by PAIRS of 16-row tiles
// that share every load of the long operand (the in-edge rows A, m_in n_l of them), and the loads of step k + 1 are issued
// before the MFMAs of step k — one wave per node has nothing else to hide a gather behind.
//
// four consecutive floats at a 4-byte aligned address (one global_load/store_dwordx4: gfx950 runs with unaligned access on)
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };

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
  const f32x4v z = {0.f, 0.f, 0.f, 0.f};
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    if (m_out <= 0 || m_in <= 0) continue;
    const int Mtot = m_in * NL;
    const int64_t t0 = tbase[s];
    for (int i0 = 0; i0 < m_out; i0 += 32) {
      const bool two = m_out - i0 > 16;
      const float* gb0 = G + (int64_t)(p0 + min(i0 + r, m_out - 1)) * V + 4 * q;
      const float* gb1 = G + (int64_t)(p0 + min(i0 + 16 + r, m_out - 1)) * V + 4 * q;
      for (int m0 = 0; m0 < Mtot; m0 += 64) {
        const int ntile = min(4, (Mtot - m0 + 15) >> 4);
        const float* ab[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = min(m0 + 16 * t + r, Mtot - 1);
          ab[t] = A + ((int64_t)in_pos[j0 + m / NL] * NL + m % NL) * V + 4 * q;
        }
        f32x4v acc[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[0][t] = z; acc[1][t] = z; }
        f32x4v a[4], b0, b1;
        {
          const bool ok = 4 * q < V;
          b0 = ok ? *(const f32x4v*)gb0 : z;
          b1 = (ok && two) ? *(const f32x4v*)gb1 : z;
#pragma unroll
          for (int t = 0; t < 4; ++t) a[t] = (ok && t < ntile) ? *(const f32x4v*)ab[t] : z;
        }
#pragma unroll 1
        for (int kk = 0; kk < KC; ++kk) {
          f32x4v na[4], nb0 = z, nb1 = z;
          const bool ok = kk + 1 < KC && 16 * (kk + 1) + 4 * q < V;
          nb0 = ok ? *(const f32x4v*)(gb0 + 16 * (kk + 1)) : z;
          nb1 = (ok && two) ? *(const f32x4v*)(gb1 + 16 * (kk + 1)) : z;
#pragma unroll
          for (int t = 0; t < 4; ++t) na[t] = (ok && t < ntile) ? *(const f32x4v*)(ab[t] + 16 * (kk + 1)) : z;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (t < ntile) {
                acc[0][t] = MFMA16(a[t][c], b0[c], acc[0][t]);
                if (two) acc[1][t] = MFMA16(a[t][c], b1[c], acc[1][t]);
              }
#pragma unroll
          for (int t = 0; t < 4; ++t) a[t] = na[t];
          b0 = nb0;
          b1 = nb1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = i0 + 16 * h + r;
          if (i < m_out) {
            float* o = out + (t0 + (int64_t)i * m_in) * NL;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int m = m0 + 16 * t + 4 * q;
              if (t < ntile) {
                if (NL == 4) {
                  if (m < Mtot) *(f32x4v*)(o + m) = acc[h][t];
                } else if (m + 4 <= Mtot) {
                  F4U y;
#pragma unroll
                  for (int jj = 0; jj < 4; ++jj) y.v[jj] = acc[h][t][jj];
                  *(F4U*)(o + m) = y;
                } else {
#pragma unroll
                  for (int jj = 0; jj < 4; ++jj)
                    if (m + jj < Mtot) o[m + jj] = acc[h][t][jj];
                }
              }
            }
          }
        }
      }
    }
  }
}

// spmm: D_cc[4r' .. ][i] with row r of tile cc <-> channel v0 + 4 r + cc;  k = (j,l) = 16 kk + 4 q + c.  The row offsets of
// the k index (in_pos lookups) sit in a per-wave LDS table, so that a step's gathers do not wait for an index load first.
#define TRIP_ROWTAB 256
template <int NL>
__global__ __launch_bounds__(256) void k_trip_spmm_node(const float* __restrict__ w, const float* __restrict__ A,
                                                        const int32_t* __restrict__ in_ptr,
                                                        const int32_t* __restrict__ in_pos,
                                                        const int32_t* __restrict__ out_ptr,
                                                        const int64_t* __restrict__ tbase, int n_nodes, int V,
                                                        float* __restrict__ out) {
  __shared__ int rowtab_all[4][TRIP_ROWTAB];
  int* rowtab = rowtab_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  const f32x4v z = {0.f, 0.f, 0.f, 0.f};
  const int NV = (V + 63) >> 6;                            // one wave per (node, 64-channel chunk)
  for (int ws = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; ws < n_nodes * NV; ws += nwave) {
    const int s = ws / NV, v0 = 64 * (ws - s * NV);
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    if (m_out <= 0) continue;
    const int Ktot = m_in * NL, KC = (Ktot + 15) >> 4;
    const int64_t t0 = tbase[s];
    __builtin_amdgcn_wave_barrier();                       // the previous node's reads of the tables are done (in-order LDS)
    for (int k = lane; k < min(Ktot, TRIP_ROWTAB); k += 64) rowtab[k] = in_pos[j0 + k / NL] * NL + k % NL;
    __builtin_amdgcn_wave_barrier();
    auto row_of = [&](int k) -> int64_t {                  // k < Ktot
      return k < TRIP_ROWTAB ? rowtab[k] : in_pos[j0 + k / NL] * NL + k % NL;
    };
    for (int i0 = 0; i0 < m_out; i0 += 32) {
      const bool two = m_out - i0 > 16;
      const float* wb0 = w + (t0 + (int64_t)min(i0 + r, m_out - 1) * m_in) * NL;
      const float* wb1 = w + (t0 + (int64_t)min(i0 + 16 + r, m_out - 1) * m_in) * NL;
      {
        const int vld = (v0 + 4 * r < V) ? v0 + 4 * r : 0;
        f32x4v acc[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[0][t] = z; acc[1][t] = z; }
        f32x4v a[4], b0, b1;
        auto load = [&](int kk, f32x4v (&aa)[4], f32x4v& bb0, f32x4v& bb1) {
          const int k = 16 * kk + 4 * q;
          if (NL == 4 || k + 4 <= Ktot) {                  // rows of w start at any 4-byte boundary when n_l != 4
            const F4U x0 = *(const F4U*)(wb0 + k), x1 = *(const F4U*)(wb1 + k);
#pragma unroll
            for (int c = 0; c < 4; ++c) { bb0[c] = x0.v[c]; bb1[c] = x1.v[c]; }
            if (NL == 4 && k >= Ktot) { bb0 = z; bb1 = z; }
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const bool in = k + c < Ktot;
              bb0[c] = in ? wb0[k + c] : 0.f;
              bb1[c] = in ? wb1[k + c] : 0.f;
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) aa[c] = *(const f32x4v*)(A + row_of(min(k + c, Ktot - 1)) * V + vld);
        };
        for (int kk = 0; kk < KC; ++kk) {
          load(kk, a, b0, b1);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              acc[0][t] = MFMA16(a[c][t], b0[c], acc[0][t]);
              if (two) acc[1][t] = MFMA16(a[c][t], b1[c], acc[1][t]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (i0 + 16 * h + r < m_out) {
            float* o = out + (int64_t)(p0 + i0 + 16 * h + r) * V;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int v = v0 + 16 * q + 4 * jj;
              if (v < V) *(f32x4v*)(o + v) = f32x4v{acc[h][0][jj], acc[h][1][jj], acc[h][2][jj], acc[h][3][jj]};
            }
          }
      }
    }
  }
}

// spmm_t: D_t[v][n = (j,l)] = sum_i G[p_i, v] w[t(i,j), l];  k = i = 4 ks + q (no permutation: K is the short out-list).
// Row r of tile t <-> channel v0 + 16 t + r (scalar loads of the few, cache-resident G rows), so that a lane's four
// accumulator rows are 16 consecutive bytes and the four lanes of a column write one whole 64-byte line per store.
// Two column tiles (32 of the m_in n_l columns) share the loads of the G rows.
template <int NL>
__global__ __launch_bounds__(256) void k_trip_spmm_t_node(const float* __restrict__ w, const float* __restrict__ G,
                                                          const int32_t* __restrict__ in_ptr,
                                                          const int32_t* __restrict__ in_pos,
                                                          const int32_t* __restrict__ out_ptr,
                                                          const int64_t* __restrict__ tbase, int n_nodes, int V,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  const f32x4v z = {0.f, 0.f, 0.f, 0.f};
  for (int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    if (m_in <= 0) continue;
    const int Ntot = m_in * NL, KS = (m_out + 3) >> 2;
    const int64_t t0 = tbase[s];
    for (int n0 = 0; n0 < Ntot; n0 += 32) {
      const bool two = Ntot - n0 > 16;
      const int na_ = min(n0 + r, Ntot - 1), nb_ = min(n0 + 16 + r, Ntot - 1);
      const float* wb0 = w + t0 * NL + na_;
      const float* wb1 = w + t0 * NL + nb_;
      const int64_t orow0 = (int64_t)in_pos[j0 + na_ / NL] * NL + na_ % NL;
      const int64_t orow1 = (int64_t)in_pos[j0 + nb_ / NL] * NL + nb_ % NL;
      for (int v0 = 0; v0 < V; v0 += 64) {
        f32x4v acc[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[0][t] = z; acc[1][t] = z; }
        auto load = [&](int ks, f32x4v& aa, float& bb0, float& bb1) {
          const int i = 4 * ks + q;
          const bool in = i < m_out;
          bb0 = in ? wb0[(int64_t)i * Ntot] : 0.f;
          bb1 = (in && two) ? wb1[(int64_t)i * Ntot] : 0.f;
          const float* gr = G + (int64_t)(p0 + min(i, max(m_out - 1, 0))) * V + v0 + r;
#pragma unroll
          for (int t = 0; t < 4; ++t) aa[t] = (v0 + 16 * t + r < V) ? gr[16 * t] : 0.f;
        };
        f32x4v a = z;
        float b0 = 0.f, b1 = 0.f;
        if (KS > 0) load(0, a, b0, b1);
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
          f32x4v na = z;
          float nb0 = 0.f, nb1 = 0.f;
          if (ks + 1 < KS) load(ks + 1, na, nb0, nb1);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[0][t] = MFMA16(a[t], b0, acc[0][t]);
            if (two) acc[1][t] = MFMA16(a[t], b1, acc[1][t]);
          }
          a = na;
          b0 = nb0;
          b1 = nb1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (n0 + 16 * h + r < Ntot) {
            float* o = out + (h ? orow1 : orow0) * V;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int v = v0 + 16 * t + 4 * q;
              if (v < V) *(f32x4v*)(o + v) = acc[h][t];
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

// ---------------------------------------------------------------------------------------------------------------------
// The per-triplet weights of lcaonet.py:179-184 in Gram form, as a function of c[t] = u_p . u_q (T) and G (E, n_l, n_l):
//     w[t, l] = [p != q] Y_l(c_t) / sqrt(max(Y(c_t)^T G_q Y(c_t), 1e-24))
// with its first derivative (gw -> gc, gG) and the derivative of that map (the second-order terms of force training).
// One thread per in-list position (node s, in-edge q = in_pos[in_ptr[s] + j]) walks the node's out-edges i: slots
// tbase[s] + i m_in + j are consecutive across the threads of a node (coalesced), G_q stays in registers and the sum over
// i that gG needs is a register accumulation — no atomics, fixed order.  The second-order kernel runs the hand-written
// reverse sweep on dual numbers (value + directional derivative): its tangent is the Hessian-vector product.
// ---------------------------------------------------------------------------------------------------------------------
struct TwDual {
  float v, d;
  __device__ TwDual() {}
  __device__ TwDual(float a) : v(a), d(0.f) {}
  __device__ TwDual(float a, float b) : v(a), d(b) {}
};
__device__ __forceinline__ TwDual operator+(TwDual a, TwDual b) { return TwDual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ TwDual operator-(TwDual a, TwDual b) { return TwDual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ TwDual operator*(TwDual a, TwDual b) { return TwDual(a.v * b.v, a.v * b.d + a.d * b.v); }
__device__ __forceinline__ TwDual operator*(float a, TwDual b) { return TwDual(a * b.v, a * b.d); }
__device__ __forceinline__ TwDual operator-(TwDual a, float b) { return TwDual(a.v - b, a.d); }
__device__ __forceinline__ float tw_value(float a) { return a; }
__device__ __forceinline__ float tw_value(TwDual a) { return a.v; }
__device__ __forceinline__ float tw_rsqrt(float a) { return 1.0f / sqrtf(a); }
__device__ __forceinline__ TwDual tw_rsqrt(TwDual a) {
  const float r = 1.0f / sqrtf(a.v);
  return TwDual(r, -0.5f * r * r * r * a.d);
}

#define TW_Y0 0.28209479177387814f
#define TW_Y1 0.48860251190292f
#define TW_Y2A 0.94617469575756f
#define TW_Y2B 0.31539156525252f
#define TW_Y3 0.373176332590115f

// y_l(c), 1/|.| and whether the clamp is inactive
template <class S, int NL>
__device__ __forceinline__ void tw_eval(S c, const S* G, S* y, S& inv, bool& live) {
  y[0] = S(TW_Y0);
  if (NL > 1) y[1] = TW_Y1 * c;
  if (NL > 2) y[2] = TW_Y2A * (c * c) - TW_Y2B;
  if (NL > 3) y[3] = TW_Y3 * (c * (5.0f * (c * c) - 3.0f));
  S n2 = S(0.f);
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int m = 0; m < NL; ++m) n2 = n2 + (y[l] * y[m]) * G[l * NL + m];
  live = tw_value(n2) >= 1e-24f;
  inv = live ? tw_rsqrt(n2) : S(1e12f);
}

// reverse sweep: gw (already masked) -> gc, gG
template <class S, int NL>
__device__ __forceinline__ void tw_grad(S c, const S* G, const float* gw, S& gc, S* gG) {
  S y[NL], inv;
  bool live;
  tw_eval<S, NL>(c, G, y, inv, live);
  S gy[NL], g_inv = S(0.f);
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    gy[l] = gw[l] * inv;
    g_inv = g_inv + gw[l] * y[l];
  }
  const S g_n2 = live ? (-0.5f) * (g_inv * (inv * (inv * inv))) : S(0.f);
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    S acc = S(0.f);
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      gG[l * NL + m] = g_n2 * (y[l] * y[m]);
      acc = acc + (G[l * NL + m] + G[m * NL + l]) * y[m];
    }
    gy[l] = gy[l] + g_n2 * acc;
  }
  gc = S(0.f);
  if (NL > 1) gc = gc + TW_Y1 * gy[1];
  if (NL > 2) gc = gc + (2.0f * TW_Y2A) * (c * gy[2]);
  if (NL > 3) gc = gc + TW_Y3 * ((15.0f * (c * c) - 3.0f) * gy[3]);
}

template <int NL>
__device__ __forceinline__ void tw_store(float* __restrict__ p, const float* v) {
  if (NL == 4) *(f32x4v*)p = f32x4v{v[0], v[1], v[2], v[3]};    // (T, 4) rows are 16-byte aligned
  else
#pragma unroll
    for (int l = 0; l < NL; ++l) p[l] = v[l];
}
template <int NL>
__device__ __forceinline__ void tw_load(const float* __restrict__ p, float* v) {
  if (NL == 4) {
    const f32x4v x = *(const f32x4v*)p;
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3 % NL] = x[3];
  } else
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = p[l];
}

// MODE 0: w;  1: gc, gG from gw;  2: ggw, gc2, gG2 from (hc, hG, gw)
template <int NL, int MODE>
__global__ void k_trip_weight(const float* __restrict__ c, const float* __restrict__ G, const float* __restrict__ gw,
                              const float* __restrict__ hc, const float* __restrict__ hG,
                              const int32_t* __restrict__ dst_p, const int32_t* __restrict__ in_ptr,
                              const int32_t* __restrict__ in_pos, const int32_t* __restrict__ out_ptr,
                              const int64_t* __restrict__ tbase, int64_t n_edges, int n_centres,
                              float* __restrict__ o_w, float* __restrict__ o_c, float* __restrict__ o_G) {
  constexpr int NG = NL * NL;
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const int q = in_pos[k];
  const int s = dst_p[q];
  if (s >= n_centres) {   // in-edge of a node that is no centre here: no triplets; its G row gets a zero gradient
    if (MODE != 0)
#pragma unroll
      for (int a = 0; a < NG; ++a) o_G[(int64_t)q * NG + a] = 0.f;
    return;
  }
  const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
  const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
  float g[NG], dg[NG], accG[NG];
#pragma unroll
  for (int a = 0; a < NG; ++a) {
    g[a] = G[(int64_t)q * NG + a];
    dg[a] = MODE == 2 ? hG[(int64_t)q * NG + a] : 0.f;
    accG[a] = 0.f;
  }
  int64_t t = tbase[s] + (k - j0);
  for (int i = 0; i < m_out; ++i, t += m_in) {
    const float cc = c[t];
    const bool keep = (p0 + i) != q;
    if (MODE == 0) {
      float y[NL], inv;
      bool live;
      tw_eval<float, NL>(cc, g, y, inv, live);
      float wv[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) wv[l] = keep ? y[l] * inv : 0.f;
      tw_store<NL>(o_w + t * NL, wv);
    } else {
      float gwt[NL];
      tw_load<NL>(gw + t * NL, gwt);
#pragma unroll
      for (int l = 0; l < NL; ++l) gwt[l] = keep ? gwt[l] : 0.f;
      if (MODE == 1) {
        float gc, gG[NG];
        tw_grad<float, NL>(cc, g, gwt, gc, gG);
        o_c[t] = gc;
#pragma unroll
        for (int a = 0; a < NG; ++a) accG[a] += gG[a];
      } else {
        TwDual Gd[NG], gc, gG[NG], y[NL], inv;
#pragma unroll
        for (int a = 0; a < NG; ++a) Gd[a] = TwDual(g[a], dg[a]);
        const TwDual cd(cc, hc[t]);
        tw_grad<TwDual, NL>(cd, Gd, gwt, gc, gG);
        o_c[t] = gc.d;
#pragma unroll
        for (int a = 0; a < NG; ++a) accG[a] += gG[a].d;
        bool live;
        tw_eval<TwDual, NL>(cd, Gd, y, inv, live);
        float wv[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) wv[l] = keep ? (y[l] * inv).d : 0.f;             // J h: the gradient wrt gw
        tw_store<NL>(o_w + t * NL, wv);
      }
    }
  }
  if (MODE != 0)
#pragma unroll
    for (int a = 0; a < NG; ++a) o_G[(int64_t)q * NG + a] = accG[a];
}

// ---------------------------------------------------------------------------------------------------------------------
// Pointwise / row-wise pieces of the edge chain with their first and second derivatives in closed form, so that the
// (E, V) tensors of the double backward pass through one kernel per level instead of autograd's decomposition:
//   SiLU (lcaonet.py:142-150 f_node activations):  s1 = sig (1 + x (1 - sig)),  s2 = sig (1 - sig) (2 + x (1 - 2 sig))
//     level 0: y = x sig        level 1: gx = g s1        level 2 (cotangent h of gx): gg = h s1,  gx2 = h g s2
//   row normalisation y = x / max(|x|, eps) (lcaonet.py:196, F.normalize), n = |x|, a = y.k, b = y.g, d = k.g:
//     level 1: gx = (g - y b) / n        level 2 (cotangent k of gx): gg = (k - y a) / n,
//                                        gx2 = ((3 a b - d) y - b k - a g) / n^2;   rows with n <= eps: x / eps, g / eps, 0
// ---------------------------------------------------------------------------------------------------------------------
template <int LEVEL>
__global__ void k_silu_levels(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ h,
                              int64_t n, float* __restrict__ o0, float* __restrict__ o1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    float xv[4], gv[4], hv[4], a0[4], a1[4];
    const bool full = i + 4 <= n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = full || i + j < n;
      xv[j] = ok ? x[i + j] : 0.f;
      gv[j] = (LEVEL >= 1 && ok) ? g[i + j] : 0.f;
      hv[j] = (LEVEL >= 2 && ok) ? h[i + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-xv[j]));
      const float s1 = sg * (1.0f + xv[j] * (1.0f - sg));
      if (LEVEL == 0) a0[j] = xv[j] * sg;
      if (LEVEL == 1) a0[j] = gv[j] * s1;
      if (LEVEL == 2) {
        a0[j] = hv[j] * s1;
        a1[j] = hv[j] * gv[j] * (sg * (1.0f - sg) * (2.0f + xv[j] * (1.0f - 2.0f * sg)));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (full || i + j < n) {
        o0[i + j] = a0[j];
        if (LEVEL == 2) o1[i + j] = a1[j];
      }
  }
}

// f(a, b) = a (c0 + b): the message product bw * h2 (c0 = 0, lcaonet.py:209-211) and the two-body injection B (1 + ft)
// (c0 = 1, lcaonet.py:192-196) with both derivative levels:
//   level 0: out0 = a (c0 + b)     level 1 (cotangent g): out0 = g (c0 + b) [d/da], out1 = g a [d/db]
//   level 2 (cotangents ha, hb of level 1's outputs): out0 = ha (c0 + b) + hb a [d/dg], out1 = hb g [d/da], out2 = ha g [d/db]
template <int LEVEL>
__global__ void k_mulp_levels(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                              const float* __restrict__ ha, const float* __restrict__ hb, float c0, int64_t n,
                              float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const f32x4v av = *(const f32x4v*)(a + i), bv = *(const f32x4v*)(b + i) + c0;
      if (LEVEL == 0) *(f32x4v*)(o0 + i) = av * bv;
      if (LEVEL == 1) {
        const f32x4v gv = *(const f32x4v*)(g + i);
        *(f32x4v*)(o0 + i) = gv * bv;
        *(f32x4v*)(o1 + i) = gv * av;
      }
      if (LEVEL == 2) {
        const f32x4v gv = *(const f32x4v*)(g + i), hav = *(const f32x4v*)(ha + i), hbv = *(const f32x4v*)(hb + i);
        *(f32x4v*)(o0 + i) = hav * bv + hbv * av;
        *(f32x4v*)(o1 + i) = hbv * gv;
        *(f32x4v*)(o2 + i) = hav * gv;
      }
    } else {
      for (int64_t k = i; k < n; ++k) {
        const float av = a[k], bv = b[k] + c0;
        if (LEVEL == 0) o0[k] = av * bv;
        if (LEVEL == 1) { o0[k] = g[k] * bv; o1[k] = g[k] * av; }
        if (LEVEL == 2) { o0[k] = ha[k] * bv + hb[k] * av; o1[k] = hb[k] * g[k]; o2[k] = ha[k] * g[k]; }
      }
    }
  }
}

// The gated copy of lcaonet.py:186 (Ag = A * gate[src], broadcast over l) and the sum over l (B = sum_l A, :200-203) with
// everything their two derivative levels need, as two primitives over (E, L, V) / (E, V) operands:
//   bcast_fma:  out[e,l,:] = X1[e,l,:] y1[e,:]  (+ X2[e,l,:] y2[e,:])  (+ z[e,:])
//   lsum_mul :  out[e,:]   = sum_l X[e,l,:] (Y[e,l,:] or 1)
__global__ void k_bcast_fma(const float* __restrict__ X1, const float* __restrict__ y1, const float* __restrict__ X2,
                            const float* __restrict__ y2, const float* __restrict__ z, int64_t n_edges, int L, int V4,
                            float* __restrict__ out) {
  const int64_t total = n_edges * V4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i / V4;
    const int c = (int)(i - e * V4);
    const int64_t yo = (e * V4 + c) * 4;
    const f32x4v a1 = *(const f32x4v*)(y1 + yo);
    const f32x4v a2 = y2 ? *(const f32x4v*)(y2 + yo) : f32x4v{0.f, 0.f, 0.f, 0.f};
    const f32x4v zz = z ? *(const f32x4v*)(z + yo) : f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      const int64_t xo = ((e * L + l) * V4 + c) * 4;
      f32x4v v = *(const f32x4v*)(X1 + xo) * a1 + zz;
      if (X2) v += *(const f32x4v*)(X2 + xo) * a2;
      *(f32x4v*)(out + xo) = v;
    }
  }
}

__global__ void k_lsum_mul(const float* __restrict__ X, const float* __restrict__ Y, int64_t n_edges, int L, int V4,
                           float* __restrict__ out) {
  const int64_t total = n_edges * V4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i / V4;
    const int c = (int)(i - e * V4);
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
      const int64_t xo = ((e * L + l) * V4 + c) * 4;
      const f32x4v x = *(const f32x4v*)(X + xo);
      acc += Y ? x * *(const f32x4v*)(Y + xo) : x;
    }
    *(f32x4v*)(out + (e * V4 + c) * 4) = acc;
  }
}

template <int R, int LEVEL>
__global__ void k_rownorm_levels(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ k,
                                 int64_t n_rows, int V, float eps, float* __restrict__ o0, float* __restrict__ o1) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; e < n_rows; e += nwave) {
    float xv[R], gv[R], kv[R];
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      const bool ok = ch < V;
      xv[r] = ok ? x[e * V + ch] : 0.f;
      gv[r] = (LEVEL >= 1 && ok) ? g[e * V + ch] : 0.f;
      kv[r] = (LEVEL >= 2 && ok) ? k[e * V + ch] : 0.f;
      n2 += xv[r] * xv[r];
    }
    const float n = sqrtf(lcao_wave_sum(n2));
    const bool live = n > eps;
    const float inv = 1.0f / fmaxf(n, eps);
    float a = 0.f, b = 0.f, d = 0.f;
    if (LEVEL >= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        b += xv[r] * gv[r];
        if (LEVEL >= 2) {
          a += xv[r] * kv[r];
          d += kv[r] * gv[r];
        }
      }
      b = lcao_wave_sum(b) * inv;
      if (LEVEL >= 2) {
        a = lcao_wave_sum(a) * inv;
        d = lcao_wave_sum(d);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      if (ch >= V) continue;
      const float y = xv[r] * inv;
      if (LEVEL == 0) o0[e * V + ch] = y;
      if (LEVEL == 1) o0[e * V + ch] = live ? (gv[r] - y * b) * inv : gv[r] * inv;
      if (LEVEL == 2) {
        o0[e * V + ch] = live ? (kv[r] - y * a) * inv : kv[r] * inv;
        o1[e * V + ch] = live ? ((3.0f * a * b - d) * y - b * kv[r] - a * gv[r]) * (inv * inv) : 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Radial basis R_d(r) (lcaonet/nn/rbf.py:17-182, cutoff.py:10-67) with R', R'' from one evaluation on second-order jets
// (value, d/dr, d^2/dr^2):   level 0: rb[e,d] = R_d(r_e)      level 1: gr[e] = sum_d g[e,d] R_d'(r_e)
//                            level 2 (cotangent h of gr): gg[e,d] = h[e] R_d'(r_e),  gr2[e] = h[e] sum_d g[e,d] R_d''(r_e)
// Same closed forms as lcao_radial_eval / lcao_cutoff_eval (lcao_common.h): factored envelopes, Horner Laguerre series.
// ---------------------------------------------------------------------------------------------------------------------
struct Jet2 {
  float v, a, b;   // f, f', f''
  __device__ Jet2() {}
  __device__ Jet2(float x) : v(x), a(0.f), b(0.f) {}
  __device__ Jet2(float x, float y, float z) : v(x), a(y), b(z) {}
};
__device__ __forceinline__ Jet2 operator+(Jet2 x, Jet2 y) { return Jet2(x.v + y.v, x.a + y.a, x.b + y.b); }
__device__ __forceinline__ Jet2 operator+(Jet2 x, float y) { return Jet2(x.v + y, x.a, x.b); }
__device__ __forceinline__ Jet2 operator*(Jet2 x, Jet2 y) {
  return Jet2(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + 2.0f * x.a * y.a + x.v * y.b);
}
__device__ __forceinline__ Jet2 operator*(float x, Jet2 y) { return Jet2(x * y.v, x * y.a, x * y.b); }
__device__ __forceinline__ Jet2 jet_exp(Jet2 x) {
  const float e = expf(x.v);
  return Jet2(e, e * x.a, e * (x.b + x.a * x.a));
}
__device__ __forceinline__ Jet2 jet_sin(Jet2 x) {
  float sn, cs;
  sincosf(x.v, &sn, &cs);
  return Jet2(sn, cs * x.a, cs * x.b - sn * x.a * x.a);
}
__device__ __forceinline__ Jet2 jet_cos(Jet2 x) {
  float sn, cs;
  sincosf(x.v, &sn, &cs);
  return Jet2(cs, -sn * x.a, -sn * x.b - cs * x.a * x.a);
}
__device__ __forceinline__ Jet2 jet_recip(Jet2 x) {
  const float u = 1.0f / x.v;
  return Jet2(u, -u * u * x.a, 2.0f * u * u * u * x.a * x.a - u * u * x.b);
}

__device__ __forceinline__ Jet2 jet_cutoff(Jet2 r, float rc, int kind) {
  if (kind == LCAO_CUTOFF_NONE) return Jet2(1.0f);
  if (!(r.v <= rc)) return Jet2(0.0f);
  const Jet2 x = (1.0f / rc) * r;
  const Jet2 om = (-1.0f) * x + 1.0f;
  if (kind == LCAO_CUTOFF_ENVELOPE) {
    Jet2 q = 15.0f * x + 10.0f;
    q = q * x + 6.0f;
    q = q * x + 3.0f;
    q = q * x + 1.0f;
    return om * om * om * q;
  } else if (kind == LCAO_CUTOFF_POLYNOMIAL) {
    Jet2 q = 6.0f * x + 3.0f;
    q = q * x + 1.0f;
    return om * om * om * q;
  }
  const Jet2 c = jet_cos(1.57079632679489661923f * x);
  return c * c;
}

__device__ __forceinline__ Jet2 jet_radial(Jet2 r, const float* __restrict__ row, int rbf_kind, float rc) {
  if (rbf_kind == LCAO_RBF_HYDROGEN) {
    const Jet2 zeta = row[0] * r;
    const int l = (int)row[2], k = (int)row[3];
    Jet2 poly(row[4 + k]);
    for (int i = k - 1; i >= 0; --i) poly = poly * zeta + row[4 + i];
    Jet2 f = row[1] * poly;
    for (int i = 0; i < l; ++i) f = f * zeta;
    return f * jet_exp(-0.5f * zeta);
  }
  return jet_sin((row[11] / rc) * r) * jet_recip(r);
}

template <int LEVEL>
__global__ void k_radial_levels(const float* __restrict__ r, const float* __restrict__ g, const float* __restrict__ h,
                                int64_t n_edges, int D, const float* __restrict__ orb_table, int rbf_kind, int cutoff_kind,
                                float rc, float* __restrict__ o_ed, float* __restrict__ o_e) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const Jet2 rj(r[e], 1.0f, 0.0f);
  const Jet2 cw = jet_cutoff(rj, rc, cutoff_kind);
  const float hv = LEVEL == 2 ? h[e] : 0.f;
  float acc = 0.f;
  for (int d = 0; d < D; ++d) {
    const Jet2 f = cw * jet_radial(rj, orb_table + d * LCAO_ORB_STRIDE, rbf_kind, rc);
    if (LEVEL == 0) o_ed[e * D + d] = f.v;
    if (LEVEL == 1) acc += g[e * D + d] * f.a;
    if (LEVEL == 2) {
      o_ed[e * D + d] = hv * f.a;
      acc += g[e * D + d] * f.b;
    }
  }
  if (LEVEL == 1) o_e[e] = acc;
  if (LEVEL == 2) o_e[e] = hv * acc;
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

#define NODE_DISPATCH(KERNEL, ...) NODE_DISPATCH_N(KERNEL, n_centres, __VA_ARGS__)
#define NODE_DISPATCH_N(KERNEL, N_WAVES, ...)                                                          \
  do {                                                                                                 \
    dim3 grid(lcao_blocks((N_WAVES), 4, 65536)), block(256);                                           \
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
    NODE_DISPATCH_N(k_trip_spmm_node, n_centres * ((V + 63) / 64), w, A, in_ptr, in_pos, out_ptr, tbase, (int)n_centres, V, out);
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
                                int64_t n_edges, int64_t n_centres, int V, int n_l, int all_targets_are_centres,
                                float* out, void* stream) {
  int rc = check_trip(n_edges, V, n_l);
  if (rc != LCAO_OK) return rc;
  if ((rc = check_centres(n_centres)) != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(w); LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(in_rank); LCAO_CHECK_PTR(in_ptr);
  LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  if (V % 4 == 0 && n_centres > 0) {
    if (!all_targets_are_centres)
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

#define TW_DISPATCH(MODE, ...)                                                                                  \
  do {                                                                                                          \
    dim3 grid(lcao_blocks(n_edges, 256)), block(256);                                                           \
    if (n_l == 1) hipLaunchKernelGGL((k_trip_weight<1, MODE>), grid, block, 0, st, __VA_ARGS__);                \
    else if (n_l == 2) hipLaunchKernelGGL((k_trip_weight<2, MODE>), grid, block, 0, st, __VA_ARGS__);           \
    else if (n_l == 3) hipLaunchKernelGGL((k_trip_weight<3, MODE>), grid, block, 0, st, __VA_ARGS__);           \
    else hipLaunchKernelGGL((k_trip_weight<4, MODE>), grid, block, 0, st, __VA_ARGS__);                         \
  } while (0)

static int check_tw(int64_t n_edges, int64_t n_centres, int n_l) {
  if (n_edges < 0 || n_edges >= (int64_t(1) << 31)) return LCAO_ERR_INVALID;
  if (n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  return check_centres(n_centres);
}

extern "C" int lcao_trip_weight(const float* c, const float* G, const int32_t* dst_p, const int32_t* in_ptr,
                                const int32_t* in_pos, const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges,
                                int64_t n_centres, int n_l, float* w, void* stream) {
  int rc = check_tw(n_edges, n_centres, n_l);
  if (rc != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(c); LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(in_ptr); LCAO_CHECK_PTR(in_pos);
  LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(w);
  hipStream_t st = (hipStream_t)stream;
  TW_DISPATCH(0, c, G, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dst_p, in_ptr, in_pos, out_ptr,
              tbase, n_edges, (int)n_centres, w, (float*)nullptr, (float*)nullptr);
  return lcao_launch_status();
}

extern "C" int lcao_trip_weight_bwd(const float* gw, const float* c, const float* G, const int32_t* dst_p,
                                    const int32_t* in_ptr, const int32_t* in_pos, const int32_t* out_ptr,
                                    const int64_t* tbase, int64_t n_edges, int64_t n_centres, int n_l, float* gc,
                                    float* gG, void* stream) {
  int rc = check_tw(n_edges, n_centres, n_l);
  if (rc != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(gw); LCAO_CHECK_PTR(c); LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(in_ptr);
  LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(gc); LCAO_CHECK_PTR(gG);
  hipStream_t st = (hipStream_t)stream;
  TW_DISPATCH(1, c, G, gw, (const float*)nullptr, (const float*)nullptr, dst_p, in_ptr, in_pos, out_ptr, tbase, n_edges,
              (int)n_centres, (float*)nullptr, gc, gG);
  return lcao_launch_status();
}

extern "C" int lcao_trip_weight_bwd2(const float* hc, const float* hG, const float* gw, const float* c, const float* G,
                                     const int32_t* dst_p, const int32_t* in_ptr, const int32_t* in_pos,
                                     const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges, int64_t n_centres,
                                     int n_l, float* ggw, float* gc2, float* gG2, void* stream) {
  int rc = check_tw(n_edges, n_centres, n_l);
  if (rc != LCAO_OK) return rc;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(hc); LCAO_CHECK_PTR(hG); LCAO_CHECK_PTR(gw); LCAO_CHECK_PTR(c); LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(dst_p);
  LCAO_CHECK_PTR(in_ptr); LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase); LCAO_CHECK_PTR(ggw);
  LCAO_CHECK_PTR(gc2); LCAO_CHECK_PTR(gG2);
  hipStream_t st = (hipStream_t)stream;
  TW_DISPATCH(2, c, G, gw, hc, hG, dst_p, in_ptr, in_pos, out_ptr, tbase, n_edges, (int)n_centres, ggw, gc2, gG2);
  return lcao_launch_status();
}

extern "C" int lcao_silu_levels(int level, const float* x, const float* g, const float* h, int64_t n, float* out0,
                                float* out1, void* stream) {
  if (n < 0 || level < 0 || level > 2) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(x); LCAO_CHECK_PTR(out0);
  if (level >= 1) LCAO_CHECK_PTR(g);
  if (level >= 2) { LCAO_CHECK_PTR(h); LCAO_CHECK_PTR(out1); }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks((n + 3) / 4, 256, 65536)), block(256);
  if (level == 0) hipLaunchKernelGGL(k_silu_levels<0>, grid, block, 0, st, x, g, h, n, out0, out1);
  else if (level == 1) hipLaunchKernelGGL(k_silu_levels<1>, grid, block, 0, st, x, g, h, n, out0, out1);
  else hipLaunchKernelGGL(k_silu_levels<2>, grid, block, 0, st, x, g, h, n, out0, out1);
  return lcao_launch_status();
}

extern "C" int lcao_rownorm_levels(int level, const float* x, const float* g, const float* k, int64_t n_rows, int V,
                                   float eps, float* out0, float* out1, void* stream) {
  if (n_rows < 0 || V <= 0 || level < 0 || level > 2) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_rows == 0) return LCAO_OK;
  LCAO_CHECK_PTR(x); LCAO_CHECK_PTR(out0);
  if (level >= 1) LCAO_CHECK_PTR(g);
  if (level >= 2) { LCAO_CHECK_PTR(k); LCAO_CHECK_PTR(out1); }
  hipStream_t st = (hipStream_t)stream;
  const int R_ = (V + 63) / 64;
  dim3 grid(lcao_blocks(n_rows, 4, 65536)), block(256);
#define RN(RR, LL) hipLaunchKernelGGL((k_rownorm_levels<RR, LL>), grid, block, 0, st, x, g, k, n_rows, V, eps, out0, out1)
#define RNL(RR)            \
  do {                     \
    if (level == 0) RN(RR, 0); \
    else if (level == 1) RN(RR, 1); \
    else RN(RR, 2);        \
  } while (0)
  if (R_ == 1) RNL(1);
  else if (R_ == 2) RNL(2);
  else if (R_ <= 4) RNL(4);
  else RNL(8);
#undef RNL
#undef RN
  return lcao_launch_status();
}

extern "C" int lcao_radial_levels(int level, const float* r, const float* g, const float* h, int64_t n_edges, int n_orb,
                                  const float* orb_table, int rbf_kind, int cutoff_kind, float rc, float* out_ed,
                                  float* out_e, void* stream) {
  if (n_edges < 0 || n_orb <= 0 || level < 0 || level > 2) return LCAO_ERR_INVALID;
  if (rbf_kind != LCAO_RBF_HYDROGEN && rbf_kind != LCAO_RBF_BESSEL) return LCAO_ERR_INVALID;
  if (cutoff_kind < LCAO_CUTOFF_ENVELOPE || cutoff_kind > LCAO_CUTOFF_NONE) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(r); LCAO_CHECK_PTR(orb_table);
  if (level != 1) LCAO_CHECK_PTR(out_ed);
  if (level >= 1) { LCAO_CHECK_PTR(g); LCAO_CHECK_PTR(out_e); }
  if (level == 2) LCAO_CHECK_PTR(h);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_edges, 256)), block(256);
#define RL(LL) hipLaunchKernelGGL(k_radial_levels<LL>, grid, block, 0, st, r, g, h, n_edges, n_orb, orb_table, rbf_kind, \
                                  cutoff_kind, rc, out_ed, out_e)
  if (level == 0) RL(0);
  else if (level == 1) RL(1);
  else RL(2);
#undef RL
  return lcao_launch_status();
}

extern "C" int lcao_mulp_levels(int level, const float* a, const float* b, const float* g, const float* ha, const float* hb,
                                float c0, int64_t n, float* out0, float* out1, float* out2, void* stream) {
  if (n < 0 || level < 0 || level > 2) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(a); LCAO_CHECK_PTR(b); LCAO_CHECK_PTR(out0);
  if (level >= 1) { LCAO_CHECK_PTR(g); LCAO_CHECK_PTR(out1); }
  if (level == 2) { LCAO_CHECK_PTR(ha); LCAO_CHECK_PTR(hb); LCAO_CHECK_PTR(out2); }
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out0)) & 15) return LCAO_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks((n + 3) / 4, 256, 65536)), block(256);
  if (level == 0) hipLaunchKernelGGL(k_mulp_levels<0>, grid, block, 0, st, a, b, g, ha, hb, c0, n, out0, out1, out2);
  else if (level == 1) hipLaunchKernelGGL(k_mulp_levels<1>, grid, block, 0, st, a, b, g, ha, hb, c0, n, out0, out1, out2);
  else hipLaunchKernelGGL(k_mulp_levels<2>, grid, block, 0, st, a, b, g, ha, hb, c0, n, out0, out1, out2);
  return lcao_launch_status();
}

extern "C" int lcao_bcast_fma(const float* X1, const float* y1, const float* X2, const float* y2, const float* z,
                              int64_t n_edges, int L, int V, float* out, void* stream) {
  if (n_edges < 0 || L < 1 || V <= 0 || V % 4) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(X1); LCAO_CHECK_PTR(y1); LCAO_CHECK_PTR(out);
  if ((X2 == nullptr) != (y2 == nullptr)) return LCAO_ERR_INVALID;
  hipLaunchKernelGGL(k_bcast_fma, dim3(lcao_blocks(n_edges * (V / 4), 256, 65536)), dim3(256), 0, (hipStream_t)stream, X1, y1,
                     X2, y2, z, n_edges, L, V / 4, out);
  return lcao_launch_status();
}

extern "C" int lcao_lsum_mul(const float* X, const float* Y, int64_t n_edges, int L, int V, float* out, void* stream) {
  if (n_edges < 0 || L < 1 || V <= 0 || V % 4) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(X); LCAO_CHECK_PTR(out);
  hipLaunchKernelGGL(k_lsum_mul, dim3(lcao_blocks(n_edges * (V / 4), 256, 65536)), dim3(256), 0, (hipStream_t)stream, X, Y,
                     n_edges, L, V / 4, out);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// The orbital contraction with everything that consumes it, as ONE map with closed-form derivative levels
// (lcaonet.py:170-186, 200-203 in species-pair form; model/double.py::_ASideFn):
//     A[p,l,:] = sum_{d in l} rb[p,d] T[pair(p),d,:]      Ag = A gate[src(p)]      B = sum_l A      G = A A^T  (n_l x n_l)
//   level 0: (rb, T, gate)                          -> Ag, B, G
//   level 1: + cotangents (gAg, gB, gG)             -> g_rb[p,d] = gA[l(d)] . T[d],  g_gate_part[p,:] = sum_l gAg[l] A[l],
//                                                      with gA[l] = gAg[l] gate + gB + sum_m (gG[l,m] + gG[m,l]) A[m]
//   level 2: + cotangents (h_rb, h_gate) of level 1 -> the derivative of  L2 = sum_d h_rb[d] g_rb[d] + h_gate[src] . g_gate_part
//            with respect to gAg, gB, gG, rb, gate[src] — and gA, dA = dL2/dA for the two table gradients
//            dL2/dT = table_grad(gA, h_rb) + table_grad(dA, rb).  With Ah[l] = sum_{d in l} h_rb[d] T[d]:
//                d gAg[l] = gate Ah[l] + h_gate A[l]        d gB = sum_l Ah[l]        d gG[l,m] = Ah[l].A[m] + Ah[m].A[l]
//                d gate_part = sum_l gAg[l] Ah[l]           dA[m] = sum_l S[l,m] Ah[l] + h_gate gAg[m]     d rb[d] = dA[l(d)] . T[d]
// A (E, n_l, V) is never stored: each level rebuilds it from rb and the table rows (L2-resident: zdim^2 x D x V floats).
// The level-1 map does not return dE/dT (the force pass wants positions only), so level 2 has no h_T.
// 16 lanes per edge (4 edges per wave), float4 channel chunks, DPP row reductions; V in {32, 64, 128}.
// ---------------------------------------------------------------------------------------------------------------------
template <int V, int NL, int LEVEL>
__global__ __launch_bounds__(256) void k_aside(
    const float* __restrict__ Tf, const float* __restrict__ rb, const int32_t* __restrict__ l_of_d,
    const int64_t* __restrict__ z, const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p,
    const float* __restrict__ gate, const float* __restrict__ gAg, const float* __restrict__ gB,
    const float* __restrict__ gG, const float* __restrict__ h_rb, const float* __restrict__ h_gate, int64_t E, int D,
    int zdim, float* __restrict__ o_big0, float* __restrict__ o_row0, float* __restrict__ o_gram,
    float* __restrict__ o_rb, float* __restrict__ o_row1, float* __restrict__ o_gA, float* __restrict__ o_dA) {
  constexpr int NC = (V + 63) / 64;
  const int sub = threadIdx.x & 15, rowi = (threadIdx.x & 63) >> 4;
  const bool chv0 = 4 * sub < V;                                   // V = 32: the upper half of the 16 lanes idles
  int64_t p0 = ((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6) * 4;
  const int64_t stride = (((int64_t)gridDim.x * blockDim.x) >> 6) * 4;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (; p0 < E; p0 += stride) {
    const int64_t p = p0 + rowi;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const int s = src_p[pc];
    const int64_t pair = z[s] * zdim + z[dst_p[pc]];
    const float* t = Tf + pair * (int64_t)D * V + 4 * sub;
    f32x4 a[NL][NC], ah[NL][NC];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int c = 0; c < NC; ++c) { a[l][c] = zero4; ah[l][c] = zero4; }
    for (int d = 0; d < D; ++d) {
      const float w = rb[pc * D + d];
      const float wh = LEVEL == 2 ? h_rb[pc * D + d] : 0.f;
      const int l = l_of_d[d];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 tv = chv0 ? *(const f32x4*)(t + d * V + 64 * c) : zero4;
#pragma unroll
        for (int ll = 0; ll < NL; ++ll) {
          a[ll][c] += (l == ll ? w : 0.f) * tv;
          if (LEVEL == 2) ah[ll][c] += (l == ll ? wh : 0.f) * tv;
        }
      }
    }
    float S[NL][NL];
    if (LEVEL >= 1) {
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int m = 0; m < NL; ++m) S[l][m] = gG[(pc * NL + l) * NL + m] + gG[(pc * NL + m) * NL + l];
    }
    f32x4 cot[NL][NC];                      // level 1: gA, level 2: dA — contracted with the table rows afterwards
    float dots[NL][NL];                     // level 0: A.A, level 2: Ah.A
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int m = 0; m < NL; ++m) dots[l][m] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t off = 64 * c + 4 * sub;
      const bool st = valid && chv0;
      const f32x4 ge = chv0 ? *(const f32x4*)(gate + (int64_t)s * V + off) : zero4;
      if (LEVEL == 0) {
        f32x4 b = zero4;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          b += a[l][c];
          if (st) *(f32x4*)(o_big0 + (p * NL + l) * V + off) = a[l][c] * ge;
#pragma unroll
          for (int m = l; m < NL; ++m) {
            const f32x4 q = a[l][c] * a[m][c];
            dots[l][m] += q[0] + q[1] + q[2] + q[3];
          }
        }
        if (st) *(f32x4*)(o_row0 + p * V + off) = b;
      } else {
        const f32x4 gb = chv0 ? *(const f32x4*)(gB + pc * V + off) : zero4;
        f32x4 gag[NL], gA[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) gag[l] = chv0 ? *(const f32x4*)(gAg + (pc * NL + l) * V + off) : zero4;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          f32x4 v = gag[l] * ge + gb;
#pragma unroll
          for (int m = 0; m < NL; ++m) v += S[l][m] * a[m][c];
          gA[l] = v;
        }
        if (LEVEL == 1) {
          f32x4 gpart = zero4;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            gpart += gag[l] * a[l][c];
            cot[l][c] = gA[l];
            if (o_gA && st) *(f32x4*)(o_gA + (p * NL + l) * V + off) = gA[l];
          }
          if (st) *(f32x4*)(o_row1 + p * V + off) = gpart;
        } else {
          const f32x4 hge = (h_gate && chv0) ? *(const f32x4*)(h_gate + (int64_t)s * V + off) : zero4;
          f32x4 dgb = zero4, dgp = zero4;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            dgb += ah[l][c];
            dgp += gag[l] * ah[l][c];
            if (st) *(f32x4*)(o_big0 + (p * NL + l) * V + off) = ge * ah[l][c] + hge * a[l][c];
            f32x4 da = hge * gag[l];
#pragma unroll
            for (int m = 0; m < NL; ++m) {
              da += S[m][l] * ah[m][c];
              const f32x4 q = ah[l][c] * a[m][c];
              dots[l][m] += q[0] + q[1] + q[2] + q[3];
            }
            cot[l][c] = da;
            if (o_gA && st) {
              *(f32x4*)(o_gA + (p * NL + l) * V + off) = gA[l];
              *(f32x4*)(o_dA + (p * NL + l) * V + off) = da;
            }
          }
          if (st) {
            *(f32x4*)(o_row0 + p * V + off) = dgb;
            *(f32x4*)(o_row1 + p * V + off) = dgp;
          }
        }
      }
    }
    if (LEVEL != 1) {
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int m = (LEVEL == 0 ? l : 0); m < NL; ++m) dots[l][m] = lcao_row16_sum(dots[l][m]);
      if (valid && sub == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int m = 0; m < NL; ++m)
            o_gram[(p * NL + l) * NL + m] = LEVEL == 0 ? (l <= m ? dots[l][m] : dots[m][l]) : dots[l][m] + dots[m][l];
      }
    }
    if (LEVEL >= 1) {
      for (int d = 0; d < D; ++d) {
        const int l = l_of_d[d];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const f32x4 tv = chv0 ? *(const f32x4*)(t + d * V + 64 * c) : zero4;
          f32x4 gsel = zero4;
#pragma unroll
          for (int ll = 0; ll < NL; ++ll) gsel = (l == ll) ? cot[ll][c] : gsel;
          acc += gsel[0] * tv[0] + gsel[1] * tv[1] + gsel[2] * tv[2] + gsel[3] * tv[3];
        }
        acc = lcao_row16_sum(acc);
        if (valid && sub == 0) o_rb[p * D + d] = acc;
      }
    }
  }
}

extern "C" int lcao_aside_levels(int level, const float* Tf, const float* rb, const int32_t* l_of_d, const int64_t* z,
                                 const int32_t* src_p, const int32_t* dst_p, const float* gate, const float* gAg,
                                 const float* gB, const float* gG, const float* h_rb, const float* h_gate, int64_t n_edges,
                                 int n_orb, int V, int n_l, int zdim, float* o_big0, float* o_row0, float* o_gram,
                                 float* o_rb, float* o_row1, float* o_gA, float* o_dA, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (level < 0 || level > 2 || n_orb <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1 || zdim <= 0) return LCAO_ERR_INVALID;
  if (V != 32 && V != 64 && V != 128) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Tf); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(z); LCAO_CHECK_PTR(src_p);
  LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(gate);
  if (level == 0) { LCAO_CHECK_PTR(o_big0); LCAO_CHECK_PTR(o_row0); LCAO_CHECK_PTR(o_gram); }
  if (level >= 1) { LCAO_CHECK_PTR(gAg); LCAO_CHECK_PTR(gB); LCAO_CHECK_PTR(gG); LCAO_CHECK_PTR(o_rb); LCAO_CHECK_PTR(o_row1); }
  if (level == 2) {
    LCAO_CHECK_PTR(h_rb); LCAO_CHECK_PTR(o_big0); LCAO_CHECK_PTR(o_row0); LCAO_CHECK_PTR(o_gram);
    if ((o_gA == nullptr) != (o_dA == nullptr)) return LCAO_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_edges, 16, 65536)), block(256);
#define AS_L(V_, NL_, LV_)                                                                                             \
  hipLaunchKernelGGL((k_aside<V_, NL_, LV_>), grid, block, 0, st, Tf, rb, l_of_d, z, src_p, dst_p, gate, gAg, gB, gG, h_rb, \
                     h_gate, n_edges, n_orb, zdim, o_big0, o_row0, o_gram, o_rb, o_row1, o_gA, o_dA)
#define AS_NL(V_, NL_)                                                                                                 \
  if (V == V_ && n_l == NL_) {                                                                                         \
    if (level == 0) AS_L(V_, NL_, 0);                                                                                  \
    else if (level == 1) AS_L(V_, NL_, 1);                                                                             \
    else AS_L(V_, NL_, 2);                                                                                             \
    return lcao_launch_status();                                                                                       \
  }
#define AS_V(V_) AS_NL(V_, 1) AS_NL(V_, 2) AS_NL(V_, 3) AS_NL(V_, 4)
  AS_V(128) AS_V(64) AS_V(32)
#undef AS_V
#undef AS_NL
#undef AS_L
  return LCAO_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
// Two more maps of the edge chain with their derivative levels, each the fusion of two of the above so that the tensor
// between them is never stored:
//   injnorm (lcaonet.py:192-196): y = x / max(|x|, eps) with x = a (1 + b)   (a = B, b = ft; mulp then rownorm)
//     level 1 (cotangent g of y): gx = (g - y (y.g)) / n;  out0 = gx (1 + b) [d/da], out1 = gx a [d/db]
//     level 2 (cotangents ha, hb of level 1's outputs): k = ha (1 + b) + hb a;  out0 = (k - y (y.k)) / n [d/dg];
//             dx = ((3 (y.k)(y.g) - k.g) y - (y.g) k - (y.k) g) / n^2;  out1 = dx (1 + b) + hb gx [d/da],  out2 = dx a + ha gx [d/db]
//   silumul (lcaonet.py:209-211): m = a silu(x)   (a = bw, x = the pre-activation of f_node's last layer; silu then mulp)
//     level 1 (cotangent g): out0 = g s [d/da], out1 = g a s1 [d/dx]
//     level 2 (cotangents ha, hx): out0 = ha s + hx a s1 [d/dg], out1 = hx g s1 [d/da], out2 = ha g s1 + hx g a s2 [d/dx]
// ---------------------------------------------------------------------------------------------------------------------
template <int R, int LEVEL>
__global__ void k_injnorm_levels(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                 const float* __restrict__ ha, const float* __restrict__ hb, int64_t n_rows, int V,
                                 float eps, float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; e < n_rows; e += nwave) {
    float av[R], bv[R], xv[R], gv[R], kv[R], hav[R], hbv[R];
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      const bool ok = ch < V;
      av[r] = ok ? a[e * V + ch] : 0.f;
      bv[r] = ok ? 1.0f + b[e * V + ch] : 0.f;
      gv[r] = (LEVEL >= 1 && ok) ? g[e * V + ch] : 0.f;
      hav[r] = (LEVEL >= 2 && ok) ? ha[e * V + ch] : 0.f;
      hbv[r] = (LEVEL >= 2 && ok) ? hb[e * V + ch] : 0.f;
      xv[r] = av[r] * bv[r];
      kv[r] = hav[r] * bv[r] + hbv[r] * av[r];
      n2 += xv[r] * xv[r];
    }
    const float n = sqrtf(lcao_wave_sum(n2));
    const bool live = n > eps;
    const float inv = 1.0f / fmaxf(n, eps);
    float ad = 0.f, bd = 0.f, d = 0.f;
    if (LEVEL >= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        bd += xv[r] * gv[r];
        if (LEVEL >= 2) {
          ad += xv[r] * kv[r];
          d += kv[r] * gv[r];
        }
      }
      bd = lcao_wave_sum(bd) * inv;
      if (LEVEL >= 2) {
        ad = lcao_wave_sum(ad) * inv;
        d = lcao_wave_sum(d);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      if (ch >= V) continue;
      const float y = xv[r] * inv;
      if (LEVEL == 0) o0[e * V + ch] = y;
      if (LEVEL >= 1) {
        const float gx = live ? (gv[r] - y * bd) * inv : gv[r] * inv;
        if (LEVEL == 1) {
          o0[e * V + ch] = gx * bv[r];
          o1[e * V + ch] = gx * av[r];
        } else {
          const float dx = live ? ((3.0f * ad * bd - d) * y - bd * kv[r] - ad * gv[r]) * (inv * inv) : 0.f;
          o0[e * V + ch] = live ? (kv[r] - y * ad) * inv : kv[r] * inv;
          o1[e * V + ch] = dx * bv[r] + hbv[r] * gx;
          o2[e * V + ch] = dx * av[r] + hav[r] * gx;
        }
      }
    }
  }
}

template <int LEVEL>
__global__ void k_silumul_levels(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ g,
                                 const float* __restrict__ ha, const float* __restrict__ hx, int64_t n,
                                 float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4v av = *(const f32x4v*)(a + i), xv = *(const f32x4v*)(x + i);
    const f32x4v gv = LEVEL >= 1 ? *(const f32x4v*)(g + i) : z4;
    const f32x4v hav = LEVEL == 2 ? *(const f32x4v*)(ha + i) : z4, hxv = LEVEL == 2 ? *(const f32x4v*)(hx + i) : z4;
    f32x4v r0, r1, r2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-xv[j]));
      const float s = xv[j] * sg;
      const float s1 = sg * (1.0f + xv[j] * (1.0f - sg));
      if (LEVEL == 0) r0[j] = av[j] * s;
      if (LEVEL == 1) { r0[j] = gv[j] * s; r1[j] = gv[j] * av[j] * s1; }
      if (LEVEL == 2) {
        const float s2 = sg * (1.0f - sg) * (2.0f + xv[j] * (1.0f - 2.0f * sg));
        r0[j] = hav[j] * s + hxv[j] * av[j] * s1;
        r1[j] = hxv[j] * gv[j] * s1;
        r2[j] = gv[j] * (hav[j] * s1 + hxv[j] * av[j] * s2);
      }
    }
    *(f32x4v*)(o0 + i) = r0;
    if (LEVEL >= 1) *(f32x4v*)(o1 + i) = r1;
    if (LEVEL == 2) *(f32x4v*)(o2 + i) = r2;
  }
}

extern "C" int lcao_injnorm_levels(int level, const float* a, const float* b, const float* g, const float* ha,
                                   const float* hb, int64_t n_rows, int V, float eps, float* out0, float* out1, float* out2,
                                   void* stream) {
  if (n_rows < 0 || V <= 0 || level < 0 || level > 2) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_rows == 0) return LCAO_OK;
  LCAO_CHECK_PTR(a); LCAO_CHECK_PTR(b); LCAO_CHECK_PTR(out0);
  if (level >= 1) { LCAO_CHECK_PTR(g); LCAO_CHECK_PTR(out1); }
  if (level == 2) { LCAO_CHECK_PTR(ha); LCAO_CHECK_PTR(hb); LCAO_CHECK_PTR(out2); }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_rows, 4, 65536)), block(256);
  const int R = (V + 63) / 64;
#define IN_L(RR, LV) hipLaunchKernelGGL((k_injnorm_levels<RR, LV>), grid, block, 0, st, a, b, g, ha, hb, n_rows, V, eps, out0, out1, out2)
#define IN_R(RR) { if (level == 0) IN_L(RR, 0); else if (level == 1) IN_L(RR, 1); else IN_L(RR, 2); }
  if (R == 1) IN_R(1) else if (R == 2) IN_R(2) else if (R <= 4) IN_R(4) else IN_R(8)
#undef IN_R
#undef IN_L
  return lcao_launch_status();
}

extern "C" int lcao_silumul_levels(int level, const float* a, const float* x, const float* g, const float* ha,
                                   const float* hx, int64_t n, float* out0, float* out1, float* out2, void* stream) {
  if (n < 0 || n % 4 || level < 0 || level > 2) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(a); LCAO_CHECK_PTR(x); LCAO_CHECK_PTR(out0);
  if (level >= 1) { LCAO_CHECK_PTR(g); LCAO_CHECK_PTR(out1); }
  if (level == 2) { LCAO_CHECK_PTR(ha); LCAO_CHECK_PTR(hx); LCAO_CHECK_PTR(out2); }
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out0)) & 15) return LCAO_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n / 4, 256, 65536)), block(256);
  if (level == 0) hipLaunchKernelGGL(k_silumul_levels<0>, grid, block, 0, st, a, x, g, ha, hx, n, out0, out1, out2);
  else if (level == 1) hipLaunchKernelGGL(k_silumul_levels<1>, grid, block, 0, st, a, x, g, ha, hx, n, out0, out1, out2);
  else hipLaunchKernelGGL(k_silumul_levels<2>, grid, block, 0, st, a, x, g, ha, hx, n, out0, out1, out2);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// pairsilu (lcaonet.py:142-150, first layer of f_node on the pair): h1[p,:] = silu(Ps[src(p)] + Pt[dst(p)]) — the
// pre-activation (E, V) is rebuilt from the two node tables at every level instead of being stored:
//   level 0: out0 = silu(pre)
//   level 1 (cotangent g of h1): out0 = g s1(pre)           — its segment sums over src / dst are dE/dPs, dE/dPt
//   level 2 (cotangents hPs, hPt (N, V) of those sums): k = hPs[src] + hPt[dst];  out0 = k s1 [d/dg],  out1 = k g s2
//            — the segment sums of out1 are d/dPs, d/dPt
// One float4 of channels per thread; V % 4 == 0.
// ---------------------------------------------------------------------------------------------------------------------
template <int LEVEL>
__global__ void k_pairsilu_levels(const float* __restrict__ Ps, const float* __restrict__ Pt,
                                  const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p,
                                  const float* __restrict__ g, const float* __restrict__ hPs, const float* __restrict__ hPt,
                                  int64_t n_edges, int V4, float* __restrict__ o0, float* __restrict__ o1) {
  const int64_t total = n_edges * V4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i / V4;
    const int c = (int)(i - e * V4);
    const int64_t so = ((int64_t)src_p[e] * V4 + c) * 4, to = ((int64_t)dst_p[e] * V4 + c) * 4;
    const f32x4v x = *(const f32x4v*)(Ps + so) + *(const f32x4v*)(Pt + to);
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4v gv = LEVEL >= 1 ? *(const f32x4v*)(g + i * 4) : z4;
    const f32x4v kv = LEVEL == 2 ? *(const f32x4v*)(hPs + so) + *(const f32x4v*)(hPt + to) : z4;
    f32x4v r0, r1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-x[j]));
      const float s1 = sg * (1.0f + x[j] * (1.0f - sg));
      if (LEVEL == 0) r0[j] = x[j] * sg;
      if (LEVEL == 1) r0[j] = gv[j] * s1;
      if (LEVEL == 2) {
        r0[j] = kv[j] * s1;
        r1[j] = kv[j] * gv[j] * (sg * (1.0f - sg) * (2.0f + x[j] * (1.0f - 2.0f * sg)));
      }
    }
    *(f32x4v*)(o0 + i * 4) = r0;
    if (LEVEL == 2) *(f32x4v*)(o1 + i * 4) = r1;
  }
}

extern "C" int lcao_pairsilu_levels(int level, const float* Ps, const float* Pt, const int32_t* src_p, const int32_t* dst_p,
                                    const float* g, const float* hPs, const float* hPt, int64_t n_edges, int V,
                                    float* out0, float* out1, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (level < 0 || level > 2 || V <= 0 || V % 4) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(out0);
  if (level >= 1) LCAO_CHECK_PTR(g);
  if (level == 2) { LCAO_CHECK_PTR(hPs); LCAO_CHECK_PTR(hPt); LCAO_CHECK_PTR(out1); }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_edges * (V / 4), 256, 65536)), block(256);
  if (level == 0) hipLaunchKernelGGL(k_pairsilu_levels<0>, grid, block, 0, st, Ps, Pt, src_p, dst_p, g, hPs, hPt, n_edges, V / 4, out0, out1);
  else if (level == 1) hipLaunchKernelGGL(k_pairsilu_levels<1>, grid, block, 0, st, Ps, Pt, src_p, dst_p, g, hPs, hPt, n_edges, V / 4, out0, out1);
  else hipLaunchKernelGGL(k_pairsilu_levels<2>, grid, block, 0, st, Ps, Pt, src_p, dst_p, g, hPs, hPt, n_edges, V / 4, out0, out1);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same derivative levels as k_aside with the gradient of the species-pair table in the same pass: the edges are
// visited in the pair-sorted chunk lists of lcao_pair_table_grad (perm, chunk_ptr: <= 64 edges of ONE species pair per
// chunk), one wave per chunk with its lanes over the channels, so that the pair's table rows stay in registers, every
// per-edge quantity is lane-local and the chunk's partial table  sum_p (rb[p,d] dX[p,l(d),:] (+ h_rb[p,d] gA[p,l(d),:]))
// accumulates in registers in list order (dX = gA at level 1, dA at level 2) — gA / dA are never written.  The caller
// sums each pair's partials in order (lcao_segment_sum): no atomics.  D <= DMAX.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lcao_wave_sum_dpp(float v) {
  v = lcao_row16_sum(v);
  const int iv = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), b = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), d = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return (a + b) + (c + d);
}

template <int R, int NL, int DMAX, int LEVEL, bool SWEEP1>   // SWEEP1: the first-order training sweep's operands (order, gram_tri)
__global__ __launch_bounds__(256) void k_aside_pairs(
    const float* __restrict__ Tf, const float* __restrict__ rb, const int32_t* __restrict__ l_of_d,
    const int64_t* __restrict__ z, const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p,
    const float* __restrict__ gate, const float* __restrict__ gAg, const float* __restrict__ gB,
    const float* __restrict__ gG, const float* __restrict__ h_rb, const float* __restrict__ h_gate,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ order,
    int gram_tri, int64_t n_chunks, int D, int V, int zdim, float* __restrict__ o_big0, float* __restrict__ o_row0,
    float* __restrict__ o_gram, float* __restrict__ o_rb, float* __restrict__ o_row1, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; c < n_chunks; c += nwave) {
    const int i0 = chunk_ptr[c], i1 = chunk_ptr[c + 1];
    float acc[DMAX][R], t[DMAX][R];
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
#pragma unroll
      for (int r = 0; r < R; ++r) { acc[d][r] = 0.f; t[d][r] = 0.f; }
    if (i0 < i1) {
      const int pf = perm[i0];
      const int64_t pair = z[src_p[pf]] * zdim + z[dst_p[pf]];
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < D)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int ch = lane + 64 * r;
            t[d][r] = ch < V ? Tf[(pair * D + d) * V + ch] : 0.f;
          }
    }
    for (int i = i0; i < i1; ++i) {
      const int64_t p = __builtin_amdgcn_readfirstlane(perm[i]);
      const int64_t s = __builtin_amdgcn_readfirstlane(src_p[p]);
      const int64_t e = (SWEEP1 && order) ? (int64_t)__builtin_amdgcn_readfirstlane(order[p]) : p;   // row of rb / d/drb
      float ge[R], hge[R], gb[R], gag[NL][R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + 64 * r;
        const bool ok = ch < V;
        ge[r] = ok ? gate[s * V + ch] : 0.f;
        hge[r] = (LEVEL == 2 && h_gate && ok) ? h_gate[s * V + ch] : 0.f;
        gb[r] = ok ? gB[p * V + ch] : 0.f;
#pragma unroll
        for (int l = 0; l < NL; ++l) gag[l][r] = ok ? gAg[(p * NL + l) * V + ch] : 0.f;
      }
      float S[NL][NL];
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int m = 0; m < NL; ++m) {
          if (SWEEP1 && gram_tri) {                        // (E, 10) upper triangle of a 4 x 4 block, Gram coordinates
            const int lo = l < m ? l : m, hi = l < m ? m : l;
            S[l][m] = gG[p * 10 + lcao_tri_idx(lo, hi)] * (l == m ? 2.0f : 1.0f);
          } else {
            S[l][m] = gG[(p * NL + l) * NL + m] + gG[(p * NL + m) * NL + l];
          }
        }
      float a[NL][R], ah[NL][R];
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int r = 0; r < R; ++r) { a[l][r] = 0.f; ah[l][r] = 0.f; }
      float wv[DMAX], whv[DMAX];
      int lv[DMAX];
#pragma unroll
      for (int d = 0; d < DMAX; ++d) {
        wv[d] = d < D ? rb[e * D + d] : 0.f;
        whv[d] = (LEVEL == 2 && d < D) ? h_rb[e * D + d] : 0.f;
        lv[d] = d < D ? l_of_d[d] : -1;
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            a[l][r] += (lv[d] == l ? wv[d] : 0.f) * t[d][r];
            if (LEVEL == 2) ah[l][r] += (lv[d] == l ? whv[d] : 0.f) * t[d][r];
          }
      }
      float gA[NL][R], dX[NL][R];                        // dX: what the table rows are contracted with for d/drb
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float v = gag[l][r] * ge[r] + gb[r];
#pragma unroll
          for (int m = 0; m < NL; ++m) v += S[l][m] * a[m][r];
          gA[l][r] = v;
          if (LEVEL == 1) dX[l][r] = v;
        }
      if (LEVEL == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          float gp = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) gp += gag[l][r] * a[l][r];
          if (ch < V) o_row1[p * V + ch] = gp;
        }
      } else {
        float dots[NL][NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
#pragma unroll
          for (int m = 0; m < NL; ++m) {
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) q += ah[l][r] * a[m][r];
            dots[l][m] = lcao_wave_sum_dpp(q);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          float dgb = 0.f, dgp = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            dgb += ah[l][r];
            dgp += gag[l][r] * ah[l][r];
            float da = hge[r] * gag[l][r];
#pragma unroll
            for (int m = 0; m < NL; ++m) da += S[m][l] * ah[m][r];
            dX[l][r] = da;
            if (ch < V) o_big0[(p * NL + l) * V + ch] = ge[r] * ah[l][r] + hge[r] * a[l][r];
          }
          if (ch < V) {
            o_row0[p * V + ch] = dgb;
            o_row1[p * V + ch] = dgp;
          }
        }
        if (lane < NL * NL) {
          const int l = lane / NL, m = lane % NL;
          float val = 0.f;
#pragma unroll
          for (int ll = 0; ll < NL; ++ll)
#pragma unroll
            for (int mm = 0; mm < NL; ++mm)
              if (ll == l && mm == m) val = dots[ll][mm] + dots[mm][ll];
          o_gram[(p * NL + l) * NL + m] = val;
        }
      }
      float mine = 0.f;                                   // lane d keeps d/drb[d]
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < D) {
          float q = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float sel = 0.f, selg = 0.f;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
              sel = lv[d] == l ? dX[l][r] : sel;
              selg = lv[d] == l ? gA[l][r] : selg;
            }
            q += sel * t[d][r];
            acc[d][r] += wv[d] * sel + (LEVEL == 2 ? whv[d] * selg : 0.f);
          }
          q = lcao_wave_sum_dpp(q);
          mine = lane == d ? q : mine;
        }
      if (lane < D) o_rb[e * D + lane] = mine;
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + 64 * r;
          if (ch < V) partial[(c * D + d) * V + ch] = acc[d][r];
        }
  }
}

extern "C" int lcao_aside_pairs(int level, const float* Tf, const float* rb, const int32_t* l_of_d, const int64_t* z,
                                const int32_t* src_p, const int32_t* dst_p, const float* gate, const float* gAg,
                                const float* gB, const float* gG, const float* h_rb, const float* h_gate,
                                const int32_t* perm, const int32_t* chunk_ptr, const int32_t* order, int gram_tri,
                                int64_t n_chunks, int64_t n_edges, int n_orb, int V, int n_l, int zdim, float* o_big0,
                                float* o_row0, float* o_gram, float* o_rb, float* o_row1, float* partial, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  LCAO_CHECK_SIZE(n_chunks);
  if (level < 1 || level > 2 || n_orb <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1 || zdim <= 0 || V <= 0) return LCAO_ERR_INVALID;
  if ((gram_tri || order) && level != 1) return LCAO_ERR_INVALID;
  if (V > 128 || n_orb > 24) return LCAO_ERR_UNSUPPORTED;
  if (n_chunks == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Tf); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(z); LCAO_CHECK_PTR(src_p);
  LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(gAg); LCAO_CHECK_PTR(gB); LCAO_CHECK_PTR(gG);
  LCAO_CHECK_PTR(perm); LCAO_CHECK_PTR(chunk_ptr); LCAO_CHECK_PTR(o_rb); LCAO_CHECK_PTR(o_row1); LCAO_CHECK_PTR(partial);
  if (level == 2) { LCAO_CHECK_PTR(h_rb); LCAO_CHECK_PTR(o_big0); LCAO_CHECK_PTR(o_row0); LCAO_CHECK_PTR(o_gram); }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_chunks, 4, 65536)), block(256);
#define AP_L(R_, NL_, DM_, LV_) { if (order || gram_tri) AP_S(R_, NL_, DM_, LV_, true); else AP_S(R_, NL_, DM_, LV_, false); }
#define AP_S(R_, NL_, DM_, LV_, S1_)                                                                                    \
  hipLaunchKernelGGL((k_aside_pairs<R_, NL_, DM_, LV_, S1_>), grid, block, 0, st, Tf, rb, l_of_d, z, src_p, dst_p, gate, gAg, gB, \
                     gG, h_rb, h_gate, perm, chunk_ptr, order, gram_tri, n_chunks, n_orb, V, zdim, o_big0, o_row0, o_gram, o_rb, \
                     o_row1, partial)
#define AP_D(R_, NL_, DM_) { if (level == 1) AP_L(R_, NL_, DM_, 1) else AP_S(R_, NL_, DM_, 2, false); return lcao_launch_status(); }
#define AP_NL(R_, NL_)                                                                                                  \
  if (n_l == NL_) {                                                                                                    \
    if (n_orb <= 8) AP_D(R_, NL_, 8) else if (n_orb <= 16) AP_D(R_, NL_, 16) else AP_D(R_, NL_, 24)                      \
  }
#define AP_R(R_) { AP_NL(R_, 1) AP_NL(R_, 2) AP_NL(R_, 3) AP_NL(R_, 4) }
  if (V <= 64) AP_R(1) else AP_R(2)
#undef AP_R
#undef AP_NL
#undef AP_D
#undef AP_L
#undef AP_S
  return LCAO_ERR_INVALID;
}
