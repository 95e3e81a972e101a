// lcao_backward.hip — activation-gradient (vector-Jacobian) kernels of the LCAOConv path: what
// `-torch.autograd.grad(E, pos)` (lcaonet/model/lcaonet.py:310-317, regress_forces with direct_forces=False) needs at
// inference, and the same sweep in training (model/train.py; weight gradients: lcao_wgrad in lcao_mfma.hip,
// pair-table gradient: lcao_pair_table_grad below).
//
// Notation as in lcaonet_amd/model/lcaonet.py (p-space edges, species-pair table Tf, Gram form of the three-body
// normalisation).  g_X denotes dE/dX.  Dense layers are differentiated with lcao_dense on transposed weights; the
// kernels here are the element-wise / per-edge / per-triplet pieces.  No float atomics: every accumulation target is
// owned by exactly one wave, or goes through a dense per-node scratch matrix that a second kernel sums in fixed order.
#include "lcao_common.h"

// out = g * silu'(pre)                      (act == SILU)
// out = g * s (1 - s), s = sigmoid(pre)     (act == SIGMOID)
__global__ void k_act_bwd(const float* __restrict__ g, const float* __restrict__ pre, int64_t n, int act,
                          float* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pre[i];
  float d = 1.0f;
  if (act == LCAO_ACT_SILU) d = lcao_silu_grad(x);
  else if (act == LCAO_ACT_SIGMOID) { const float s = lcao_sigmoid(x); d = s * (1.0f - s); }
  out[i] = g[i] * d;
}

extern "C" int lcao_act_bwd(const float* g, const float* pre, int64_t n, int act, float* out, void* stream) {
  if (n < 0 || act < 0 || act > 2) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g); LCAO_CHECK_PTR(pre); LCAO_CHECK_PTR(out);
  hipLaunchKernelGGL(k_act_bwd, dim3(lcao_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, g, pre, n, act, out);
  return lcao_launch_status();
}

// e4/e5 backward (lcaonet.py:207-213): msg = bw * silu(pre2), agg[s] = sum msg.
//   g_msg[p] = g_agg[src_p[p]];  g_bw = g_msg * silu(pre2);  g_pre2 = g_msg * bw * silu'(pre2)
__global__ void k_message_bwd(const float* __restrict__ g_agg, const int32_t* __restrict__ src_p,
                              const float* __restrict__ bw, const float* __restrict__ pre2, int64_t E, int V,
                              float* __restrict__ g_bw, float* __restrict__ g_pre2) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    const float* ga = g_agg + (int64_t)src_p[p] * V;
    for (int c = lane; c < V; c += 64) {
      const float gm = ga[c], x = pre2[p * V + c];
      g_bw[p * V + c] = gm * lcao_silu(x);
      g_pre2[p * V + c] = gm * bw[p * V + c] * lcao_silu_grad(x);
    }
  }
}

extern "C" int lcao_message_bwd(const float* g_agg, const int32_t* src_p, const float* bw, const float* pre2,
                                int64_t n_edges, int V, float* g_bw, float* g_pre2, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_agg); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(bw); LCAO_CHECK_PTR(pre2); LCAO_CHECK_PTR(g_bw);
  LCAO_CHECK_PTR(g_pre2);
  hipLaunchKernelGGL(k_message_bwd, dim3(lcao_blocks(n_edges, 4, 65536)), dim3(256), 0, (hipStream_t)stream, g_agg,
                     src_p, bw, pre2, n_edges, V, g_bw, g_pre2);
  return lcao_launch_status();
}

// pre1[p] = Ps[src] + Pt[dst]   (the pre-activation of the first f_node layer, needed for silu')
__global__ void k_pair_pre(const float* __restrict__ Ps, const float* __restrict__ Pt,
                           const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p, int64_t E, int V,
                           float* __restrict__ pre1) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    const float* a = Ps + (int64_t)src_p[p] * V;
    const float* b = Pt + (int64_t)dst_p[p] * V;
    for (int c = lane; c < V; c += 64) pre1[p * V + c] = a[c] + b[c];
  }
}

extern "C" int lcao_pair_pre(const float* Ps, const float* Pt, const int32_t* src_p, const int32_t* dst_p,
                             int64_t n_edges, int V, float* pre1, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(pre1);
  hipLaunchKernelGGL(k_pair_pre, dim3(lcao_blocks(n_edges, 4, 65536)), dim3(256), 0, (hipStream_t)stream, Ps, Pt, src_p,
                     dst_p, n_edges, V, pre1);
  return lcao_launch_status();
}

// e3 backward (lcaonet.py:192-204): u = B (1 + ft), lw = u / max(|u|, eps).
//   g_u = (g_lw - lw (lw . g_lw)) / |u|   (0 where |u| <= eps: the clamped branch is constant in direction 0/eps)
//   g_B = g_u (1 + ft);  g_ft = g_u B
template <int R>
__global__ void k_combine_bwd(const float* __restrict__ g_lw, const float* __restrict__ B,
                              const float* __restrict__ ft, int64_t E, int V, float* __restrict__ g_B,
                              float* __restrict__ g_ft) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    float u[R], gl[R], b[R], f[R];
    float n2 = 0.f, dot = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      b[r] = f[r] = u[r] = gl[r] = 0.f;
      if (ch < V) {
        b[r] = B[p * V + ch];
        f[r] = ft[p * V + ch];
        u[r] = b[r] + b[r] * f[r];
        gl[r] = g_lw[p * V + ch];
      }
      n2 += u[r] * u[r];
      dot += u[r] * gl[r];
    }
    n2 = lcao_wave_sum(n2);
    dot = lcao_wave_sum(dot);
    const float nrm = sqrtf(n2);
    const bool live = nrm > 1e-12f;
    const float inv = live ? 1.0f / nrm : 1e12f;
    // lw = u * inv;  g_u = inv * (g_lw - lw (lw . g_lw)) when live, inv * g_lw in the clamped (constant-scale) branch
    const float proj = live ? dot * inv * inv : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      if (ch < V) {
        const float gu = inv * (gl[r] - u[r] * proj);
        g_B[p * V + ch] = gu * (1.0f + f[r]);
        g_ft[p * V + ch] = gu * b[r];
      }
    }
  }
}

extern "C" int lcao_combine_bwd(const float* g_lw, const float* B, const float* ft, int64_t n_edges, int V, float* g_B,
                                float* g_ft, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_lw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(ft); LCAO_CHECK_PTR(g_B); LCAO_CHECK_PTR(g_ft);
  dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int R = (V + 63) / 64;
  if (R == 1) hipLaunchKernelGGL(k_combine_bwd<1>, grid, block, 0, st, g_lw, B, ft, n_edges, V, g_B, g_ft);
  else if (R == 2) hipLaunchKernelGGL(k_combine_bwd<2>, grid, block, 0, st, g_lw, B, ft, n_edges, V, g_B, g_ft);
  else if (R <= 4) hipLaunchKernelGGL(k_combine_bwd<4>, grid, block, 0, st, g_lw, B, ft, n_edges, V, g_B, g_ft);
  else hipLaunchKernelGGL(k_combine_bwd<8>, grid, block, 0, st, g_lw, B, ft, n_edges, V, g_B, g_ft);
  return lcao_launch_status();
}

// add_valence form (lcaonet.py:195-203): u = B (1 + ft_a) + Bv (1 + ft_b), ft = [ft_a | ft_b] (E, 2V):
//   g_B = g_u (1 + ft_a);  g_Bv = g_u (1 + ft_b);  g_ft = [g_u B | g_u Bv]
template <int R>
__global__ void k_combine_bwd_valence(const float* __restrict__ g_lw, const float* __restrict__ B,
                                      const float* __restrict__ Bv, const float* __restrict__ ft, int64_t E, int V,
                                      float* __restrict__ g_B, float* __restrict__ g_Bv, float* __restrict__ g_ft) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    float u[R], gl[R], b[R], bv[R], fa[R], fb[R];
    float n2 = 0.f, dot = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      b[r] = bv[r] = fa[r] = fb[r] = u[r] = gl[r] = 0.f;
      if (ch < V) {
        b[r] = B[p * V + ch];
        bv[r] = Bv[p * V + ch];
        fa[r] = ft[p * 2 * V + ch];
        fb[r] = ft[p * 2 * V + V + ch];
        u[r] = (b[r] + b[r] * fa[r]) + (bv[r] + bv[r] * fb[r]);
        gl[r] = g_lw[p * V + ch];
      }
      n2 += u[r] * u[r];
      dot += u[r] * gl[r];
    }
    n2 = lcao_wave_sum(n2);
    dot = lcao_wave_sum(dot);
    const float nrm = sqrtf(n2);
    const bool live = nrm > 1e-12f;
    const float inv = live ? 1.0f / nrm : 1e12f;
    const float proj = live ? dot * inv * inv : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      if (ch < V) {
        const float gu = inv * (gl[r] - u[r] * proj);
        g_B[p * V + ch] = gu * (1.0f + fa[r]);
        g_Bv[p * V + ch] = gu * (1.0f + fb[r]);
        g_ft[p * 2 * V + ch] = gu * b[r];
        g_ft[p * 2 * V + V + ch] = gu * bv[r];
      }
    }
  }
}

extern "C" int lcao_combine_bwd_valence(const float* g_lw, const float* B, const float* Bv, const float* ft,
                                        int64_t n_edges, int V, float* g_B, float* g_Bv, float* g_ft, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_lw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(Bv); LCAO_CHECK_PTR(ft); LCAO_CHECK_PTR(g_B);
  LCAO_CHECK_PTR(g_Bv); LCAO_CHECK_PTR(g_ft);
  dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int R = (V + 63) / 64;
#define CBV(RR) hipLaunchKernelGGL(k_combine_bwd_valence<RR>, grid, block, 0, st, g_lw, B, Bv, ft, n_edges, V, g_B, g_Bv, g_ft)
  if (R == 1) CBV(1);
  else if (R == 2) CBV(2);
  else if (R <= 4) CBV(4);
  else CBV(8);
#undef CBV
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// e2 backward — three-body term  tbw[p] = sum_{q in In(s), q != p} sum_l w_pql Ag[q,l,:],
//   w_pql = Y_l(c) * inv,  c = u_p . u_q,  inv = 1 / max(sqrt(m), eps),  m = Y^T G_q Y   (G upper-triangular, 10 floats)
// One wave per in-edge q (it owns g_Ag[q], g_gram[q], and its part of g_u[q]); the per-triplet scalar
// g_c = dE/dc goes to the dense per-node scratch gc[tbase[s] + j * m_out + i] (j = index of q in In(s), i = index of p
// in Out(s)) so that a second kernel can form g_u[p] = sum_q g_c u_q without atomics.
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ void k_triplet_bwd_q(const float* __restrict__ g_tbw, const float* __restrict__ Ag,
                                const float* __restrict__ gram, const float* __restrict__ uvec_p,
                                const int32_t* __restrict__ dst_p, const int32_t* __restrict__ in_ptr,
                                const int32_t* __restrict__ in_pos, const int32_t* __restrict__ out_ptr,
                                const int64_t* __restrict__ tbase, int64_t E, int n_own, int V, int NL,
                                float* __restrict__ g_Ag, float* __restrict__ g_gram, float* __restrict__ g_uq,
                                float* __restrict__ gc) {
  int lane = threadIdx.x & 63;
  int64_t jj = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;   // position in the in-edge lists
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n_in_total = in_ptr[n_own];
  for (; jj < n_in_total; jj += nwave) {
    const int q = in_pos[jj];
    const int s = dst_p[q];                    // centre node whose in-list holds q
    const int j = (int)(jj - in_ptr[s]);
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    float a[4][R], ga[4][R];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int ch = lane + 64 * r;
        a[l][r] = (l < NL && ch < V) ? Ag[((int64_t)q * NL + l) * V + ch] : 0.f;
        ga[l][r] = 0.f;
      }
    float G[10], gG[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { G[i] = gram[(int64_t)q * 10 + i]; gG[i] = 0.f; }
    const float qx = uvec_p[3 * (int64_t)q], qy = uvec_p[3 * (int64_t)q + 1], qz = uvec_p[3 * (int64_t)q + 2];
    float gux = 0.f, guy = 0.f, guz = 0.f;
    for (int i = 0; i < m_out; ++i) {
      const int p = p0 + i;
      float gcv = 0.f;
      if (p != q) {
        const float px = uvec_p[3 * (int64_t)p], py = uvec_p[3 * (int64_t)p + 1], pz = uvec_p[3 * (int64_t)p + 2];
        const float c = px * qx + py * qy + pz * qz;
        float y[4], dy[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) { y[l] = (l < NL) ? lcao_y_l0(c, l) : 0.f; dy[l] = (l < NL) ? lcao_y_l0_grad(c, l) : 0.f; }
        // m = |R Y|^2 = Y^T A A^T Y and GY = R^T R Y from the edge's triangular factor (lcao_common.h)
        float GY[4];
        const float m = lcao_factor_quadratic<4>(G, y, GY);
        const float nrm = sqrtf(fmaxf(m, 0.f));
        const bool live = nrm > 1e-12f;
        const float inv = live ? 1.0f / nrm : 1e12f;
        // g_w_l = g_tbw[p] . Ag[q,l]
        float gt[R];
        float gw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          int ch = lane + 64 * r;
          gt[r] = ch < V ? g_tbw[(int64_t)p * V + ch] : 0.f;
#pragma unroll
          for (int l = 0; l < 4; ++l) gw[l] += gt[r] * a[l][r];
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) gw[l] = (l < NL) ? lcao_wave_sum(gw[l]) : 0.f;
        float g_inv = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) g_inv += gw[l] * y[l];
        const float g_m = live ? -0.5f * g_inv * inv * inv * inv : 0.f;
        float gY[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) gY[l] = gw[l] * inv + 2.0f * g_m * GY[l];
#pragma unroll
        for (int l = 0; l < 4; ++l) gcv += gY[l] * dy[l];
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
          for (int l2 = l; l2 < 4; ++l2)
            gG[l * 4 - (l * (l - 1)) / 2 + (l2 - l)] += g_m * (l == l2 ? y[l] * y[l] : 2.0f * y[l] * y[l2]);
        // g_Ag[q,l] += w_l g_tbw[p]
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const float w = y[l] * inv;
#pragma unroll
          for (int r = 0; r < R; ++r) ga[l][r] += w * gt[r];
        }
        gux += gcv * px; guy += gcv * py; guz += gcv * pz;
      }
      if (lane == 0) gc[tbase[s] + (int64_t)j * m_out + i] = gcv;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < NL) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          int ch = lane + 64 * r;
          if (ch < V) g_Ag[((int64_t)q * NL + l) * V + ch] = ga[l][r];
        }
      }
    if (lane < 10) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 10; ++i) v = (lane == i) ? gG[i] : v;
      g_gram[(int64_t)q * 10 + lane] = v;
    }
    if (lane == 0) { g_uq[3 * (int64_t)q] = gux; g_uq[3 * (int64_t)q + 1] = guy; g_uq[3 * (int64_t)q + 2] = guz; }
  }
}

// g_u[p] (as the OUT-edge of its centre) = sum_{q in In(s)} g_c[p,q] u_q ;  one thread per out-edge
__global__ void k_triplet_bwd_p(const float* __restrict__ gc, const float* __restrict__ uvec_p,
                                const int32_t* __restrict__ src_p, const int32_t* __restrict__ in_ptr,
                                const int32_t* __restrict__ in_pos, const int32_t* __restrict__ out_ptr,
                                const int64_t* __restrict__ tbase, int64_t e_own, float* __restrict__ g_up) {
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= e_own) return;
  const int s = src_p[p];
  const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0, i = (int)(p - p0);
  const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int j = 0; j < m_in; ++j) {
    const int q = in_pos[j0 + j];
    const float g = gc[tbase[s] + (int64_t)j * m_out + i];
    gx += g * uvec_p[3 * (int64_t)q]; gy += g * uvec_p[3 * (int64_t)q + 1]; gz += g * uvec_p[3 * (int64_t)q + 2];
  }
  g_up[3 * p] = gx; g_up[3 * p + 1] = gy; g_up[3 * p + 2] = gz;
}

extern "C" int lcao_triplet_bwd(const float* g_tbw, const float* Ag, const float* gram, const float* uvec_p,
                                const int32_t* src_p, const int32_t* dst_p, const int32_t* in_ptr, const int32_t* in_pos,
                                const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges, int64_t e_own, int n_own,
                                int64_t n_in_total, int V, int n_l, float* g_Ag, float* g_gram, float* g_uq, float* g_up,
                                float* gc_scratch, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1 || n_own < 0) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0 || n_own == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_tbw); LCAO_CHECK_PTR(Ag); LCAO_CHECK_PTR(gram); LCAO_CHECK_PTR(uvec_p); LCAO_CHECK_PTR(src_p);
  LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(in_ptr); LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbase);
  LCAO_CHECK_PTR(g_Ag); LCAO_CHECK_PTR(g_gram); LCAO_CHECK_PTR(g_uq); LCAO_CHECK_PTR(g_up); LCAO_CHECK_PTR(gc_scratch);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_in_total > 0 ? n_in_total : 1, 4, 65536)), block(256);
  int R = (V + 63) / 64;
#define TB_LAUNCH(RR)                                                                                               \
  hipLaunchKernelGGL(k_triplet_bwd_q<RR>, grid, block, 0, st, g_tbw, Ag, gram, uvec_p, dst_p, in_ptr, in_pos, out_ptr, \
                     tbase, n_edges, n_own, V, n_l, g_Ag, g_gram, g_uq, gc_scratch)
  if (R == 1) TB_LAUNCH(1);
  else if (R == 2) TB_LAUNCH(2);
  else if (R <= 4) TB_LAUNCH(4);
  else TB_LAUNCH(8);
#undef TB_LAUNCH
  hipLaunchKernelGGL(k_triplet_bwd_p, dim3(lcao_blocks(e_own > 0 ? e_own : 1, 256)), dim3(256), 0, st, gc_scratch, uvec_p,
                     src_p, in_ptr, in_pos, out_ptr, tbase, e_own, g_up);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// e1 backward, species-pair form:  A[l] = sum_{d in l} rb[e,d] Tf[pair,d,:],  Ag = A gate[src],  B = sum_l A,
// gram = A A^T (upper triangle).   Given g_Ag, g_B, g_gram:
//   g_A[l]  = g_Ag[l] gate + g_B + sum_l2 gGsym[l,l2] A[l2]      (gGsym[l,l] = 2 g_gram[l,l], off-diagonal = g_gram)
//   g_rb[e,d] = g_A[l(d)] . Tf[pair,d,:]
//   g_gate_part[p,:] = sum_l g_Ag[l] A[l]      (segment-summed over out-edges of src by the caller)
// One wave per edge.
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ void k_pair_contract_bwd(const float* __restrict__ Tf, const float* __restrict__ rb,
                                    const int32_t* __restrict__ order, const int32_t* __restrict__ l_of_d,
                                    const int64_t* __restrict__ z, const int32_t* __restrict__ src_p,
                                    const int32_t* __restrict__ dst_p, const float* __restrict__ gate,
                                    const float* __restrict__ g_Ag, const float* __restrict__ g_B,
                                    const float* __restrict__ g_gram, int64_t E, int64_t e_own, int D, int V, int NL,
                                    int zdim, float* __restrict__ g_rb, float* __restrict__ g_gate_part,
                                    float* __restrict__ g_A_out) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    const int64_t e = order ? order[p] : p;
    const int s = src_p[p];
    const int64_t pair = z[s] * zdim + z[dst_p[p]];
    const float* t = Tf + pair * (int64_t)D * V;
    float a[4][R], gA[4][R];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) a[l][r] = 0.f;
    for (int d = 0; d < D; ++d) {
      const float w = rb[e * D + d];
      const int l = l_of_d[d];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int ch = lane + 64 * r;
        const float v = ch < V ? w * t[d * V + ch] : 0.f;
        a[0][r] += (l == 0) ? v : 0.f; a[1][r] += (l == 1) ? v : 0.f;
        a[2][r] += (l == 2) ? v : 0.f; a[3][r] += (l == 3) ? v : 0.f;
      }
    }
    float gG[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) gG[i] = g_gram[p * 10 + i];
    const bool own = p < e_own;   // only owned out-edges carry a two-body term (g_B); ghost out-edges feed the 3-body term only
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      const float gt = ch < V ? gate[(int64_t)s * V + ch] : 0.f;
      const float gb = (own && ch < V) ? g_B[p * V + ch] : 0.f;
      float gpart = 0.f;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        float gag = (l < NL && ch < V) ? g_Ag[(p * NL + l) * V + ch] : 0.f;
        gpart += gag * a[l][r];
        float v = gag * gt + ((l < NL) ? gb : 0.f);
#pragma unroll
        for (int l2 = 0; l2 < 4; ++l2) {
          const int lo = l < l2 ? l : l2, hi = l < l2 ? l2 : l;
          const float gs = gG[lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo)] * (l == l2 ? 2.0f : 1.0f);
          v += gs * a[l2][r];
        }
        gA[l][r] = v;
        if (g_A_out && l < NL && ch < V) g_A_out[(p * NL + l) * V + ch] = v;
      }
      if (g_gate_part && ch < V) g_gate_part[p * V + ch] = gpart;
    }
    for (int d = 0; d < D; ++d) {
      const int l = l_of_d[d];
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int ch = lane + 64 * r;
        const float tv = ch < V ? t[d * V + ch] : 0.f;
        const float g = l == 0 ? gA[0][r] : (l == 1 ? gA[1][r] : (l == 2 ? gA[2][r] : gA[3][r]));
        acc += g * tv;
      }
      acc = lcao_wave_sum(acc);
      if (lane == 0) g_rb[e * D + d] += acc;   // accumulated over layers (the caller zeroes it once per forward)
    }
  }
}

// fast form for V in {32, 64, 128}: 16 lanes per edge (4 edges per wave), float4 channel chunks, DPP row reductions
template <int V, int NL, int DC>   // DC = 8: n_orb <= 8, the table block of the edge is kept in registers between the two passes
__global__ __launch_bounds__(256) void k_pair_contract_bwd16(
    const float* __restrict__ Tf, const float* __restrict__ rb, const int32_t* __restrict__ order,
    const int32_t* __restrict__ l_of_d, const int64_t* __restrict__ z, const int32_t* __restrict__ src_p,
    const int32_t* __restrict__ dst_p, const float* __restrict__ gate, const float* __restrict__ g_Ag,
    const float* __restrict__ g_B, const float* __restrict__ g_gram, int64_t E, int64_t e_own, int D, int zdim,
    float* __restrict__ g_rb, float* __restrict__ g_gate_part, float* __restrict__ g_A_out) {
  constexpr int NC = (V + 63) / 64;
  const int sub = threadIdx.x & 15, rowi = (threadIdx.x & 63) >> 4;
  int64_t p0 = ((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6) * 4;
  const int64_t stride = (((int64_t)gridDim.x * blockDim.x) >> 6) * 4;
  for (; p0 < E; p0 += stride) {
    const int64_t p = p0 + rowi;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const int64_t e = order ? order[pc] : pc;
    const int s = src_p[pc];
    const int64_t pair = z[s] * zdim + z[dst_p[pc]];
    const float* t = Tf + pair * (int64_t)D * V + 4 * sub;
    f32x4 a[NL][NC], gA[NL][NC];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int c = 0; c < NC; ++c) a[l][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 tc[DC > 0 ? DC : 1][NC];
    if constexpr (DC > 0) {
#pragma unroll
      for (int d = 0; d < DC; ++d) {
        if (d < D) {
          const float w = rb[e * D + d];
          const int l = l_of_d[d];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            tc[d][c] = (64 * c + 4 * sub) < V ? *(const f32x4*)(t + d * V + 64 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ll = 0; ll < NL; ++ll) a[ll][c] += (l == ll ? w : 0.f) * tc[d][c];
          }
        }
      }
    } else {
      for (int d = 0; d < D; ++d) {
        const float w = rb[e * D + d];
        const int l = l_of_d[d];
#pragma unroll
        for (int ll = 0; ll < NL; ++ll)
          if (l == ll) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
              if ((64 * c + 4 * sub) < V) a[ll][c] += w * *(const f32x4*)(t + d * V + 64 * c);
          }
      }
    }
    float gG[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) gG[i] = g_gram[pc * 10 + i];
    const bool own = pc < e_own;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool chv = (64 * c + 4 * sub) < V;
      const int64_t off = 64 * c + 4 * sub;
      const f32x4 gt = chv ? *(const f32x4*)(gate + (int64_t)s * V + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 gb = (own && chv) ? *(const f32x4*)(g_B + pc * V + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 gpart = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 gag[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        gag[l] = chv ? *(const f32x4*)(g_Ag + (pc * NL + l) * V + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        gpart += gag[l] * a[l][c];
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        f32x4 v = gag[l] * gt + gb;
#pragma unroll
        for (int l2 = 0; l2 < NL; ++l2) {
          const int lo = l < l2 ? l : l2, hi = l < l2 ? l2 : l;
          v += (gG[lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo)] * (l == l2 ? 2.0f : 1.0f)) * a[l2][c];
        }
        gA[l][c] = v;
        if (g_A_out && valid && chv) *(f32x4*)(g_A_out + (p * NL + l) * V + off) = v;
      }
      if (g_gate_part && valid && chv) *(f32x4*)(g_gate_part + p * V + off) = gpart;
    }
    if constexpr (DC > 0) {
#pragma unroll
      for (int d = 0; d < DC; ++d) {
        if (d < D) {
          const int l = l_of_d[d];
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            f32x4 gsel = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ll = 0; ll < NL; ++ll) gsel = (l == ll) ? gA[ll][c] : gsel;
            acc += gsel[0] * tc[d][c][0] + gsel[1] * tc[d][c][1] + gsel[2] * tc[d][c][2] + gsel[3] * tc[d][c][3];
          }
          acc = lcao_row16_sum(acc);
          if (valid && sub == 0) g_rb[e * D + d] += acc;
        }
      }
    } else {
      for (int d = 0; d < D; ++d) {
        const int l = l_of_d[d];
        float acc = 0.f;
#pragma unroll
        for (int ll = 0; ll < NL; ++ll)
          if (l == ll) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
              if ((64 * c + 4 * sub) < V) {
                const f32x4 tv = *(const f32x4*)(t + d * V + 64 * c);
                acc += gA[ll][c][0] * tv[0] + gA[ll][c][1] * tv[1] + gA[ll][c][2] * tv[2] + gA[ll][c][3] * tv[3];
              }
          }
        acc = lcao_row16_sum(acc);
        if (valid && sub == 0) g_rb[e * D + d] += acc;
      }
    }
  }
}

extern "C" int lcao_pair_contract_bwd(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                      const int64_t* z, const int32_t* src_p, const int32_t* dst_p, const float* gate,
                                      const float* g_Ag, const float* g_B, const float* g_gram, int64_t n_edges,
                                      int64_t e_own, int n_orb, int V, int n_l, int zdim, float* g_rb,
                                      float* g_gate_part, float* g_A_out, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1 || zdim <= 0) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Tf); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(z); LCAO_CHECK_PTR(src_p);
  LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(g_Ag); LCAO_CHECK_PTR(g_B); LCAO_CHECK_PTR(g_gram);
  LCAO_CHECK_PTR(g_rb);
  dim3 grid(lcao_blocks(n_edges, 4, 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define PB16(V_, NL_)                                                                                                  \
  if (V == V_ && n_l == NL_) {                                                                                         \
    if (n_orb <= 8)                                                                                                    \
      hipLaunchKernelGGL((k_pair_contract_bwd16<V_, NL_, 8>), dim3(lcao_blocks(n_edges, 16, 65536)), block, 0, st, Tf, rb,  \
                         order, l_of_d, z, src_p, dst_p, gate, g_Ag, g_B, g_gram, n_edges, e_own, n_orb, zdim, g_rb,    \
                         g_gate_part, g_A_out);                                                                        \
    else                                                                                                               \
      hipLaunchKernelGGL((k_pair_contract_bwd16<V_, NL_, 0>), dim3(lcao_blocks(n_edges, 16, 65536)), block, 0, st, Tf, rb,  \
                         order, l_of_d, z, src_p, dst_p, gate, g_Ag, g_B, g_gram, n_edges, e_own, n_orb, zdim, g_rb,    \
                         g_gate_part, g_A_out);                                                                        \
    return lcao_launch_status();                                                                                       \
  }
#define PB16V(V_) PB16(V_, 1) PB16(V_, 2) PB16(V_, 3) PB16(V_, 4)
  PB16V(128) PB16V(64) PB16V(32)
#undef PB16V
#undef PB16
  int R = (V + 63) / 64;
#define PB_LAUNCH(RR)                                                                                                 \
  hipLaunchKernelGGL(k_pair_contract_bwd<RR>, grid, block, 0, st, Tf, rb, order, l_of_d, z, src_p, dst_p, gate, g_Ag, g_B, \
                     g_gram, n_edges, e_own, n_orb, V, n_l, zdim, g_rb, g_gate_part, g_A_out)
  if (R == 1) PB_LAUNCH(1);
  else if (R == 2) PB_LAUNCH(2);
  else if (R <= 4) PB_LAUNCH(4);
  else PB_LAUNCH(8);
#undef PB_LAUNCH
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// basis + geometry backward: g_rb (E,D), g_u (E,3) -> g_vec (E,3) = dE/d(vec_e), vec = pos[t] - pos[s] + shift @ lattice.
//   rb[e,d] = R_d(r):  g_r = sum_d g_rb[e,d] R_d'(r);   u = vec / r:  g_vec = (g_u - u (u . g_u)) / r + g_r u
// R_nl' for the hydrogen basis: R = cw(r) f(zeta), f = C P(zeta) zeta^l exp(-zeta/2), zeta = s r
//   R' = cw'(r) f + cw s C [P' zeta^l + l P zeta^(l-1) - P zeta^l / 2] exp(-zeta/2)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lcao_cutoff_grad(float r, float rc, int kind) {
  if (kind == LCAO_CUTOFF_NONE || !(r <= rc)) return 0.0f;
  const float x = r / rc;
  const float om = 1.0f - x;   // factored like lcao_cutoff_eval: no cancellation towards rc
  if (kind == LCAO_CUTOFF_ENVELOPE) {
    const float x2 = x * x;
    return -105.0f * x2 * x2 * om * om / rc;   // d/dx (1 - 21x^5 + 35x^6 - 15x^7) = -105 x^4 (1 - x)^2
  } else if (kind == LCAO_CUTOFF_POLYNOMIAL) {
    return -30.0f * x * x * om * om / rc;      // d/dx (1 - 6x^5 + 15x^4 - 10x^3) = -30 x^2 (1 - x)^2
  }
  return -0.5f * sinf(r * 3.14159265358979323846f / rc) * 3.14159265358979323846f / rc;
}

__global__ void k_geometry_bwd(const float* __restrict__ dist, const float* __restrict__ uvec,
                               const float* __restrict__ g_rb, const float* __restrict__ g_u, int64_t E, int D,
                               const float* __restrict__ tab, int rbf_kind, int cutoff_kind, float rc,
                               float* __restrict__ g_vec) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float r = dist[e];
  const float cw = lcao_cutoff_eval(r, rc, cutoff_kind), dcw = lcao_cutoff_grad(r, rc, cutoff_kind);
  float g_r = 0.f;
  for (int d = 0; d < D; ++d) {
    const float* row = tab + d * LCAO_ORB_STRIDE;
    float dR;
    if (rbf_kind == LCAO_RBF_HYDROGEN) {
      const float s = row[0], C = row[1];
      const int l = (int)row[2], k = (int)row[3];
      const float zeta = s * r;
      float P = 0.f, dP = 0.f;   // Horner for P and P'
      for (int i = k; i >= 0; --i) { dP = dP * zeta + P; P = P * zeta + row[4 + i]; }
      const float zl = (l == 0) ? 1.0f : (l == 1 ? zeta : (l == 2 ? zeta * zeta : zeta * zeta * zeta));
      const float zlm1 = (l <= 1) ? 1.0f : (l == 2 ? zeta : zeta * zeta);   // zeta^(l-1) (unused when l == 0)
      const float ex = expf(-0.5f * zeta);
      const float f = C * P * zl * ex;
      const float df = C * s * ex * (dP * zl + (l > 0 ? (float)l * P * zlm1 : 0.f) - 0.5f * P * zl);
      dR = dcw * f + cw * df;
    } else {
      const float kk = row[11] / rc;
      const float sn = sinf(kk * r), cs = cosf(kk * r);
      dR = dcw * sn / r + cw * (kk * cs / r - sn / (r * r));
    }
    g_r += g_rb[e * D + d] * dR;
  }
  const float ux = uvec[3 * e], uy = uvec[3 * e + 1], uz = uvec[3 * e + 2];
  const float gx = g_u[3 * e], gy = g_u[3 * e + 1], gz = g_u[3 * e + 2];
  const float dot = ux * gx + uy * gy + uz * gz;
  g_vec[3 * e] = (gx - ux * dot) / r + g_r * ux;
  g_vec[3 * e + 1] = (gy - uy * dot) / r + g_r * uy;
  g_vec[3 * e + 2] = (gz - uz * dot) / r + g_r * uz;
}

extern "C" int lcao_geometry_bwd(const float* dist, const float* uvec, const float* g_rb, const float* g_u,
                                 int64_t n_edges, int n_orb, const float* orb_table, int rbf_kind, int cutoff_kind,
                                 float rc, float* g_vec, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || rbf_kind < 0 || rbf_kind > 1 || cutoff_kind < 0 || cutoff_kind > 3) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(dist); LCAO_CHECK_PTR(uvec); LCAO_CHECK_PTR(g_rb); LCAO_CHECK_PTR(g_u); LCAO_CHECK_PTR(orb_table);
  LCAO_CHECK_PTR(g_vec);
  hipLaunchKernelGGL(k_geometry_bwd, dim3(lcao_blocks(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, dist, uvec, g_rb,
                     g_u, n_edges, n_orb, orb_table, rbf_kind, cutoff_kind, rc, g_vec);
  return lcao_launch_status();
}


// ---------------------------------------------------------------------------------------------------
// gradient of the species-pair table (training):  A[p,l,:] = sum_{d in l} rb[e,d] Tf[pair(p),d,:]  =>
//   g_Tf[pair,d,:] = sum_{p: pair(p) = pair} rb[e,d] g_A[p,l(d),:]
// Edges are visited sorted by pair id (perm); every pair's list is cut into chunks of <= 64 edges (chunk_ptr over the
// sorted list, chunks never span pairs); one wave per chunk writes a partial (D, V) table; the caller sums each pair's
// partials in order (lcao_segment_sum) — deterministic, no atomics.
// ---------------------------------------------------------------------------------------------------
// D accumulators per lane (DMAX orbitals at a time) so that every g_A row of the chunk is read once per pass
template <int R, int DMAX>
__global__ void k_pair_table_grad(const float* __restrict__ g_A, const float* __restrict__ rb,
                                  const int32_t* __restrict__ order, const int32_t* __restrict__ l_of_d,
                                  const int32_t* __restrict__ perm, const int32_t* __restrict__ chunk_ptr,
                                  int64_t n_chunks, int D, int V, int NL, float* __restrict__ partial) {
  int lane = threadIdx.x & 63;
  int64_t c = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; c < n_chunks; c += nwave) {
    const int i0 = chunk_ptr[c], i1 = chunk_ptr[c + 1];
    for (int d0 = 0; d0 < D; d0 += DMAX) {
      float acc[DMAX][R];
#pragma unroll
      for (int dd = 0; dd < DMAX; ++dd)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[dd][r] = 0.f;
      // U edges per iteration: their index loads and their g_A rows are independent and stay in flight together
      constexpr int U = 4;
      for (int i = i0; i < i1; i += U) {
        int pu[U];
        int64_t eu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          pu[u] = perm[min(i + u, i1 - 1)];
          eu[u] = order ? order[pu[u]] : pu[u];
        }
        float ga[U][4][R];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int ch = lane + 64 * r;
              ga[u][l][r] = (l < NL && ch < V && (i + u) < i1) ? g_A[((int64_t)pu[u] * NL + l) * V + ch] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {   // rows are added in list order: the sum is the same as one edge at a time
#pragma unroll
          for (int dd = 0; dd < DMAX; ++dd) {
            const int d = d0 + dd;
            if (d < D) {
              const float w = (i + u) < i1 ? rb[eu[u] * D + d] : 0.f;
              const int l = l_of_d[d];
#pragma unroll
              for (int r = 0; r < R; ++r)
                acc[dd][r] += w * (l == 0 ? ga[u][0][r] : (l == 1 ? ga[u][1][r] : (l == 2 ? ga[u][2][r] : ga[u][3][r])));
            }
          }
        }
      }
#pragma unroll
      for (int dd = 0; dd < DMAX; ++dd) {
        const int d = d0 + dd;
        if (d < D) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int ch = lane + 64 * r;
            if (ch < V) partial[(c * D + d) * V + ch] = acc[dd][r];
          }
        }
      }
    }
  }
}

extern "C" int lcao_pair_table_grad(const float* g_A, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                    const int32_t* perm, const int32_t* chunk_ptr, int64_t n_chunks, int n_orb, int V,
                                    int n_l, float* partial, void* stream) {
  LCAO_CHECK_SIZE(n_chunks);
  if (n_orb <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_chunks == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_A); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(perm); LCAO_CHECK_PTR(chunk_ptr);
  LCAO_CHECK_PTR(partial);
  dim3 grid(lcao_blocks(n_chunks, 4, 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int R = (V + 63) / 64;
#define PTG(RR)                                                                                                        \
  hipLaunchKernelGGL((k_pair_table_grad<RR, 8>), grid, block, 0, st, g_A, rb, order, l_of_d, perm, chunk_ptr, n_chunks,   \
                     n_orb, V, n_l, partial)
  if (R == 1) PTG(1);
  else if (R == 2) PTG(2);
  else if (R <= 4) PTG(4);
  else PTG(8);
#undef PTG
  return lcao_launch_status();
}
