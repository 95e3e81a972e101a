// lcao_generic.hip — shape-generic kernels of the LCAONet hot path (any D, C, V, H).
// HBM-bound elementwise / gather / segmented-reduction work: coalesced row accesses, one wave per row
// or per segment, no float atomics (results are bitwise reproducible run to run).
#include "lcao_common.h"

// ---------------------------------------------------------------------------------------------------
// a1  edge geometry  (lcaonet/model/base.py:33-42)
// ---------------------------------------------------------------------------------------------------
__global__ void k_edge_geometry(const float* __restrict__ pos, const int64_t* __restrict__ idx_s,
                                const int64_t* __restrict__ idx_t, const float* __restrict__ shift,
                                const float* __restrict__ lattice, const int64_t* __restrict__ batch, int64_t E,
                                float* __restrict__ dist, float* __restrict__ uvec) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = idx_s[e], t = idx_t[e];
  const float* L = lattice + 9 * (batch ? batch[s] : 0);
  float sx = shift[3 * e], sy = shift[3 * e + 1], sz = shift[3 * e + 2];
  float v[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float off = sx * L[j] + sy * L[3 + j] + sz * L[6 + j];  // einsum("ni,nij->nj")
    v[j] = (pos[3 * t + j] - pos[3 * s + j]) + off;
  }
  float d = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  dist[e] = d;
  uvec[3 * e] = v[0] / d;
  uvec[3 * e + 1] = v[1] / d;
  uvec[3 * e + 2] = v[2] / d;
}

extern "C" int lcao_edge_geometry(const float* pos, const int64_t* idx_s, const int64_t* idx_t, const float* shift,
                                  const float* lattice, const int64_t* batch, int64_t n_edges, float* dist,
                                  float* uvec, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(pos); LCAO_CHECK_PTR(idx_s); LCAO_CHECK_PTR(idx_t); LCAO_CHECK_PTR(shift);
  LCAO_CHECK_PTR(lattice); LCAO_CHECK_PTR(dist); LCAO_CHECK_PTR(uvec);
  hipLaunchKernelGGL(k_edge_geometry, dim3(lcao_blocks(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                     idx_s, idx_t, shift, lattice, batch, n_edges, dist, uvec);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// a4  cutoff,  a5/a5' radial bases,  a6 angular basis,  a3 angles
// ---------------------------------------------------------------------------------------------------
__global__ void k_cutoff(const float* __restrict__ r, int64_t n, int kind, float rc, float* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = lcao_cutoff_eval(r[i], rc, kind);
}

extern "C" int lcao_cutoff(const float* r, int64_t n, int kind, float rc, float* out, void* stream) {
  LCAO_CHECK_SIZE(n);
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(r); LCAO_CHECK_PTR(out);
  if (kind < 0 || kind > 2) return LCAO_ERR_INVALID;
  hipLaunchKernelGGL(k_cutoff, dim3(lcao_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, r, n, kind, rc, out);
  return lcao_launch_status();
}

// R_nl(r) = cutoff(r) * C_nl * P_nl(zeta) * zeta^l * exp(-zeta/2), zeta = 2 r / (n r0)   (rbf.py:96-103).
// P_nl is summed in expanded monomial form, highest power first, as the reference's lambdified sympy code does.
__device__ __forceinline__ float lcao_hydrogen_rnl(float r, float cw, const float* __restrict__ row) {
  float zeta = row[0] * r;
  int l = (int)row[2], k = (int)row[3];
  float zp[7];
  zp[0] = 1.0f;
#pragma unroll
  for (int i = 1; i < 7; ++i) zp[i] = zp[i - 1] * zeta;
  float poly = 0.0f;
#pragma unroll
  for (int i = 6; i >= 0; --i) {  // static indices only (runtime-indexed arrays would go to scratch)
    if (i <= k) {
      float term = row[4 + i] * zp[i];
      poly = (i == k) ? term : poly + term;
    }
  }
  float zl = (l == 0) ? 1.0f : (l == 1 ? zeta : (l == 2 ? zeta * zeta : zeta * zeta * zeta));
  return cw * row[1] * poly * zl * expf(-zeta / 2.0f);
}

__global__ void k_radial_basis(const float* __restrict__ r, int64_t E, int D, const float* __restrict__ tab,
                               int rbf_kind, int cutoff_kind, float rc, float* __restrict__ rb) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  float x = r[e];
  float cw = lcao_cutoff_eval(x, rc, cutoff_kind);
  for (int d = 0; d < D; ++d) {
    const float* row = tab + d * LCAO_ORB_STRIDE;
    float v;
    if (rbf_kind == LCAO_RBF_HYDROGEN) v = lcao_hydrogen_rnl(x, cw, row);
    else v = cw * sinf(row[11] * x / rc) / x;  // rbf.py:168: cutoff * sin(n pi r / rc) / r
    rb[e * D + d] = v;
  }
}

extern "C" int lcao_radial_basis(const float* r, int64_t n_edges, int n_orb, const float* orb_table, int rbf_kind,
                                 int cutoff_kind, float rc, float* rb, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || rbf_kind < 0 || rbf_kind > 1 || cutoff_kind < 0 || cutoff_kind > 3) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(r); LCAO_CHECK_PTR(orb_table); LCAO_CHECK_PTR(rb);
  hipLaunchKernelGGL(k_radial_basis, dim3(lcao_blocks(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, r,
                     n_edges, n_orb, orb_table, rbf_kind, cutoff_kind, rc, rb);
  return lcao_launch_status();
}

// a1 + a4 + a5 in ONE pass over the edges in p-space (sorted by centre): pos -> dist, unit vectors and the radial basis
// with no `dist` round trip through HBM, and the unit vectors written a second time directly in p-space (what
// lcao_triplet_gemm reads), which removes the separate row permutation.  Arithmetic identical to lcao_edge_geometry +
// lcao_radial_basis.  Algorithmic bytes per edge: 12 (3 int32) + 12 (shift) + 24 (pos, cache-resident) read; 4 + 12 + 12 +
// 4 D written.
__global__ void k_edge_geometry_basis(const float* __restrict__ pos, const int32_t* __restrict__ order,
                                      const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p,
                                      const float* __restrict__ shift, const float* __restrict__ lattice,
                                      const int64_t* __restrict__ batch, int64_t E, int D,
                                      const float* __restrict__ tab, int rbf_kind, int cutoff_kind, float rc,
                                      float* __restrict__ dist, float* __restrict__ uvec, float* __restrict__ uvec_p,
                                      float* __restrict__ rb) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= E) return;
  const int64_t e = order[p];
  const int64_t s = src_p[p], t = dst_p[p];
  const float* L = lattice + 9 * (batch ? batch[s] : 0);
  const float sx = shift[3 * e], sy = shift[3 * e + 1], sz = shift[3 * e + 2];
  float v[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float off = sx * L[j] + sy * L[3 + j] + sz * L[6 + j];
    v[j] = (pos[3 * t + j] - pos[3 * s + j]) + off;
  }
  const float d = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float ux = v[0] / d, uy = v[1] / d, uz = v[2] / d;
  dist[e] = d;
  uvec[3 * e] = ux; uvec[3 * e + 1] = uy; uvec[3 * e + 2] = uz;
  uvec_p[3 * p] = ux; uvec_p[3 * p + 1] = uy; uvec_p[3 * p + 2] = uz;
  const float cw = lcao_cutoff_eval(d, rc, cutoff_kind);
  for (int k = 0; k < D; ++k) {
    const float* row = tab + k * LCAO_ORB_STRIDE;
    float val;
    if (rbf_kind == LCAO_RBF_HYDROGEN) val = lcao_hydrogen_rnl(d, cw, row);
    else val = cw * sinf(row[11] * d / rc) / d;
    rb[e * D + k] = val;
  }
}

extern "C" int lcao_edge_geometry_basis(const float* pos, const int32_t* order, const int32_t* src_p, const int32_t* dst_p,
                                        const float* shift, const float* lattice, const int64_t* batch, int64_t n_edges,
                                        int n_orb, const float* orb_table, int rbf_kind, int cutoff_kind, float rc,
                                        float* dist, float* uvec, float* uvec_p, float* rb, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || rbf_kind < 0 || rbf_kind > 1 || cutoff_kind < 0 || cutoff_kind > 3) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(pos); LCAO_CHECK_PTR(order); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(shift);
  LCAO_CHECK_PTR(lattice); LCAO_CHECK_PTR(orb_table); LCAO_CHECK_PTR(dist); LCAO_CHECK_PTR(uvec); LCAO_CHECK_PTR(uvec_p);
  LCAO_CHECK_PTR(rb);
  hipLaunchKernelGGL(k_edge_geometry_basis, dim3(lcao_blocks(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, pos, order,
                     src_p, dst_p, shift, lattice, batch, n_edges, n_orb, orb_table, rbf_kind, cutoff_kind, rc, dist, uvec,
                     uvec_p, rb);
  return lcao_launch_status();
}

__global__ void k_sph_basis(const float* __restrict__ c, int64_t n, int D, const int32_t* __restrict__ l_of_d,
                            float* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  int64_t t = i / D;
  int d = int(i - t * D);
  out[i] = lcao_y_l0(c[t], l_of_d[d]);
}

extern "C" int lcao_sph_basis(const float* costheta, int64_t n, int n_orb, const int32_t* l_of_d, float* shb,
                              void* stream) {
  LCAO_CHECK_SIZE(n);
  if (n_orb <= 0) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(costheta); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(shb);
  if (n * n_orb >= (int64_t(1) << 40)) return LCAO_ERR_INVALID;
  hipLaunchKernelGGL(k_sph_basis, dim3(lcao_blocks(n * n_orb, 256)), dim3(256), 0, (hipStream_t)stream, costheta,
                     n, n_orb, l_of_d, shb);
  return lcao_launch_status();
}

__global__ void k_triplet_angles(const float* __restrict__ u, const int64_t* __restrict__ e_st,
                                 const int64_t* __restrict__ e_ks, int64_t n, float* __restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = u + 3 * e_st[i];
  const float* b = u + 3 * e_ks[i];
  out[i] = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

extern "C" int lcao_triplet_angles(const float* uvec, const int64_t* e_st, const int64_t* e_ks, int64_t n_trip,
                                   float* cosv, void* stream) {
  if (n_trip < 0) return LCAO_ERR_INVALID;
  if (n_trip == 0) return LCAO_OK;
  LCAO_CHECK_PTR(uvec); LCAO_CHECK_PTR(e_st); LCAO_CHECK_PTR(e_ks); LCAO_CHECK_PTR(cosv);
  hipLaunchKernelGGL(k_triplet_angles, dim3(lcao_blocks(n_trip, 256)), dim3(256), 0, (hipStream_t)stream, uvec,
                     e_st, e_ks, n_trip, cosv);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// row gathers / permutes (16-B vectorised when width % 4 == 0)
// ---------------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ void k_gather_rows(const float* __restrict__ table, const IdxT* __restrict__ idx, int64_t n, int width,
                              float* __restrict__ out) {
  int lane = threadIdx.x & 63;
  int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; row < n; row += nwave) {
    const float* src = table + (int64_t)idx[row] * width;
    float* dst = out + row * width;
    if ((width & 3) == 0) {
      for (int c = lane * 4; c < width; c += 256) *(float4*)(dst + c) = *(const float4*)(src + c);
    } else {
      for (int c = lane; c < width; c += 64) dst[c] = src[c];
    }
  }
}

extern "C" int lcao_gather_rows(const float* table, const int64_t* idx, int64_t n, int width, float* out,
                                void* stream) {
  LCAO_CHECK_SIZE(n);
  if (width <= 0) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(table); LCAO_CHECK_PTR(idx); LCAO_CHECK_PTR(out);
  hipLaunchKernelGGL(k_gather_rows<int64_t>, dim3(lcao_blocks(n, 4, 8192)), dim3(256), 0, (hipStream_t)stream,
                     table, idx, n, width, out);
  return lcao_launch_status();
}

extern "C" int lcao_permute_rows(const float* src, const int32_t* perm, int64_t n, int width, float* out,
                                 void* stream) {
  LCAO_CHECK_SIZE(n);
  if (width <= 0) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(src); LCAO_CHECK_PTR(perm); LCAO_CHECK_PTR(out);
  hipLaunchKernelGGL(k_gather_rows<int32_t>, dim3(lcao_blocks(n, 4, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     perm, n, width, out);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// a8  segment sum over contiguous segments: one wave per segment, lanes over the feature width.
// Deterministic: rows are added in index order.
// ---------------------------------------------------------------------------------------------------
__global__ void k_segment_sum(const float* __restrict__ a, const float* __restrict__ b,
                              const int32_t* __restrict__ ptr, int64_t n_seg, int width, float* __restrict__ out) {
  int lane = threadIdx.x & 63;
  int64_t seg = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; seg < n_seg; seg += nwave) {
    int r0 = ptr[seg], r1 = ptr[seg + 1];
    for (int c = lane; c < width; c += 64) {
      float acc = 0.0f;
      if (b) {
        for (int r = r0; r < r1; ++r) acc += a[(int64_t)r * width + c] * b[(int64_t)r * width + c];
      } else {
        for (int r = r0; r < r1; ++r) acc += a[(int64_t)r * width + c];
      }
      out[seg * width + c] = acc;
    }
  }
}

// 16-byte form for widths 4*WL, WL in {8, 16, 32, 64} lanes per row: 64/WL rows per load instruction, two running sums
// per lane, then a fixed shuffle tree over the row groups (deterministic; the association differs from the row-by-row
// kernel above, not from run to run).  idx (optional): row r of the segment list is a[idx[r]] — a gather fused into the sum.
template <int WL>
__global__ void k_segment_sum_v4(const float* __restrict__ a, const float* __restrict__ b,
                                 const int32_t* __restrict__ idx, const int32_t* __restrict__ ptr, int64_t n_seg,
                                 float* __restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int RP = 64 / WL, width = 4 * WL;
  const int lane = threadIdx.x & 63, g = lane / WL, cl = lane % WL;
  int64_t seg = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; seg < n_seg; seg += nwave) {
    const int r0 = ptr[seg], r1 = ptr[seg + 1];
    f4 s0 = f4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int r = r0 + g;
    for (; r + RP < r1; r += 2 * RP) {
      const int64_t ra = idx ? idx[r] : r, rb_ = idx ? idx[r + RP] : r + RP;
      f4 va = *(const f4*)(a + ra * width + 4 * cl), vb = *(const f4*)(a + rb_ * width + 4 * cl);
      if (b) { va *= *(const f4*)(b + ra * width + 4 * cl); vb *= *(const f4*)(b + rb_ * width + 4 * cl); }
      s0 += va;
      s1 += vb;
    }
    if (r < r1) {
      const int64_t ra = idx ? idx[r] : r;
      f4 va = *(const f4*)(a + ra * width + 4 * cl);
      if (b) va *= *(const f4*)(b + ra * width + 4 * cl);
      s0 += va;
    }
    s0 += s1;
#pragma unroll
    for (int off = 32; off >= WL; off >>= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s0[k] += __shfl_xor(s0[k], off);
    }
    if (g == 0) *(f4*)(out + seg * width + 4 * cl) = s0;
  }
}

__global__ void k_segment_sum_idx(const float* __restrict__ a, const int32_t* __restrict__ idx,
                                  const int32_t* __restrict__ ptr, int64_t n_seg, int width, float* __restrict__ out) {
  int lane = threadIdx.x & 63;
  int64_t seg = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; seg < n_seg; seg += nwave) {
    int r0 = ptr[seg], r1 = ptr[seg + 1];
    for (int c = lane; c < width; c += 64) {
      float acc = 0.0f;
      for (int r = r0; r < r1; ++r) acc += a[(int64_t)idx[r] * width + c];
      out[seg * width + c] = acc;
    }
  }
}

static int launch_segment_sum(const float* a, const float* b, const int32_t* idx, const int32_t* ptr, int64_t n_seg,
                              int width, float* out, void* stream) {
  LCAO_CHECK_SIZE(n_seg);
  if (width <= 0) return LCAO_ERR_INVALID;
  if (n_seg == 0) return LCAO_OK;
  LCAO_CHECK_PTR(a); LCAO_CHECK_PTR(ptr); LCAO_CHECK_PTR(out);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(lcao_blocks(n_seg, 4, 16384)), block(256);
  const bool al = (((uintptr_t)a | (uintptr_t)out | (uintptr_t)(b ? b : a)) % 16) == 0;
#define SS_CASE(WL_)                                                                                        \
  if (al && width == 4 * WL_) {                                                                             \
    hipLaunchKernelGGL(k_segment_sum_v4<WL_>, grid, block, 0, st, a, b, idx, ptr, n_seg, out);              \
    return lcao_launch_status();                                                                            \
  }
  SS_CASE(8) SS_CASE(16) SS_CASE(32) SS_CASE(64)
#undef SS_CASE
  if (idx) {
    if (b) return LCAO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_segment_sum_idx, grid, block, 0, st, a, idx, ptr, n_seg, width, out);
  } else {
    hipLaunchKernelGGL(k_segment_sum, grid, block, 0, st, a, b, ptr, n_seg, width, out);
  }
  return lcao_launch_status();
}

extern "C" int lcao_segment_sum(const float* a, const float* b, const int32_t* ptr, int64_t n_seg, int width,
                                float* out, void* stream) {
  return launch_segment_sum(a, b, nullptr, ptr, n_seg, width, out, stream);
}

extern "C" int lcao_segment_sum_gather(const float* a, const int32_t* idx, const int32_t* ptr, int64_t n_seg, int width,
                                       float* out, void* stream) {
  if (n_seg > 0) LCAO_CHECK_PTR(idx);
  return launch_segment_sum(a, nullptr, idx, ptr, n_seg, width, out, stream);
}

// ---------------------------------------------------------------------------------------------------
// contraction of f_coeffs(cst) with the radial basis, grouped by l  (exact regrouping of lcaonet.py:173-183,200-203)
// one wave per edge; lanes over V.
// ---------------------------------------------------------------------------------------------------
__global__ void k_coeff_contract(const float* __restrict__ cstf, const float* __restrict__ rb,
                                 const int32_t* __restrict__ order, const int32_t* __restrict__ l_of_d,
                                 const float* __restrict__ mask, int64_t mse, int64_t msd, int64_t msv, int64_t E,
                                 int D, int V, int NL, int valence, float* __restrict__ A, float* __restrict__ B,
                                 float* __restrict__ Bv) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int Vt = valence ? 2 * V : V;
  for (; p < E; p += nwave) {
    int64_t e = order ? order[p] : p;
    const float* crow = cstf + e * (int64_t)D * Vt;
    for (int c = lane; c < V; c += 64) {
      float al[LCAO_MAX_L + 1] = {0.f, 0.f, 0.f, 0.f};
      float bsum = 0.f, bvsum = 0.f;
      for (int d = 0; d < D; ++d) {
        float w = rb[e * D + d];
        float cc = crow[d * Vt + c];
        float t = w * cc;
        bsum += t;
        if (valence) {
          float m = mask[e * mse + d * msd + c * msv];
          float tv = w * (m * crow[d * Vt + V + c]);
          bvsum += tv;
          t += tv;
        }
        int l = l_of_d[d];
        al[0] += (l == 0) ? t : 0.f;
        al[1] += (l == 1) ? t : 0.f;
        al[2] += (l == 2) ? t : 0.f;
        al[3] += (l == 3) ? t : 0.f;
      }
      for (int l = 0; l < NL; ++l) A[(p * NL + l) * V + c] = al[l];
      B[p * V + c] = bsum;
      if (valence) Bv[p * V + c] = bvsum;
    }
  }
}

extern "C" int lcao_coeff_contract(const float* cstf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                   const float* mask, int64_t mask_se, int64_t mask_sd, int64_t mask_sv,
                                   int64_t n_edges, int n_orb, int V, int n_l, int valence, float* A, float* B,
                                   float* Bv, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(cstf); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(A); LCAO_CHECK_PTR(B);
  if (valence) { LCAO_CHECK_PTR(mask); LCAO_CHECK_PTR(Bv); }
  hipLaunchKernelGGL(k_coeff_contract, dim3(lcao_blocks(n_edges, 4, 16384)), dim3(256), 0, (hipStream_t)stream,
                     cstf, rb, order, l_of_d, mask, mask_se, mask_sd, mask_sv, n_edges, n_orb, V, n_l, valence, A, B,
                     Bv);
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// three-body term, generic form: one wave per out-edge p, lanes over V (R = ceil(V/64) values per lane),
// loop over the in-edges of the centre node; exact per-triplet L2 norm by wave reduction.
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ void k_triplet_sum(const float* __restrict__ A, const float* __restrict__ gate,
                              const float* __restrict__ uvec, const int32_t* __restrict__ order,
                              const int32_t* __restrict__ src_p, const int32_t* __restrict__ in_ptr,
                              const int32_t* __restrict__ in_pos, int64_t E, int V, int NL,
                              float* __restrict__ tbw) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    int s = src_p[p];
    int64_t ep = order ? order[p] : p;
    float ux = uvec[3 * ep], uy = uvec[3 * ep + 1], uz = uvec[3 * ep + 2];
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    int j0 = in_ptr[s], j1 = in_ptr[s + 1];
    for (int j = j0; j < j1; ++j) {
      int q = in_pos[j];
      if (q == p) continue;  // lcaonet.py:470: only the edge itself is dropped (the back-triplet k == t stays)
      int64_t eq = order ? order[q] : q;
      float c = ux * uvec[3 * eq] + uy * uvec[3 * eq + 1] + uz * uvec[3 * eq + 2];
      float y[LCAO_MAX_L + 1];
#pragma unroll
      for (int l = 0; l <= LCAO_MAX_L; ++l) y[l] = lcao_y_l0(c, l);
      const float* Aq = A + (int64_t)q * NL * V;
      float v[R];
      float n2 = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int ch = lane + 64 * r;
        float t = 0.f;
        if (ch < V) {
          for (int l = 0; l < NL; ++l) t += y[l] * Aq[l * V + ch];
        }
        v[r] = t;
        n2 += t * t;
      }
      n2 = lcao_wave_sum(n2);
      float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);  // F.normalize: v / max(||v||, eps)
      const float* g = gate + (int64_t)src_p[q] * V;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int ch = lane + 64 * r;
        if (ch < V) acc[r] += (v[r] * inv) * g[ch];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      if (ch < V) tbw[p * V + ch] = acc[r];
    }
  }
}

extern "C" int lcao_triplet_sum(const float* A, const float* gate, const float* uvec, const int32_t* order,
                                const int32_t* src_p, const int32_t* in_ptr, const int32_t* in_pos, int64_t n_edges,
                                int V, int n_l, float* tbw, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(A); LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(uvec); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(in_ptr);
  LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(tbw);
  dim3 grid(lcao_blocks(n_edges, 4, 32768)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int R = (V + 63) / 64;
#define LAUNCH_TS(RR)                                                                                            \
  hipLaunchKernelGGL(k_triplet_sum<RR>, grid, block, 0, st, A, gate, uvec, order, src_p, in_ptr, in_pos, n_edges, \
                     V, n_l, tbw)
  if (R == 1) LAUNCH_TS(1);
  else if (R == 2) LAUNCH_TS(2);
  else if (R <= 4) LAUNCH_TS(4);
  else LAUNCH_TS(8);
#undef LAUNCH_TS
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// lw = normalize((1+ft) * B [+ (1+ft_v) * Bv])   (lcaonet.py:192-204); one wave per edge
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ void k_edge_combine(const float* __restrict__ ft, const float* __restrict__ B,
                               const float* __restrict__ Bv, int64_t E, int V, float* __restrict__ lw) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int Vt = Bv ? 2 * V : V;
  for (; p < E; p += nwave) {
    float v[R];
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      float t = 0.f;
      if (ch < V) {
        float b = B[p * V + ch];
        t = b + b * ft[p * Vt + ch];  // cst + cst * f_three(.)  summed over d
        if (Bv) {
          float bv = Bv[p * V + ch];
          t += bv + bv * ft[p * Vt + V + ch];
        }
      }
      v[r] = t;
      n2 += t * t;
    }
    n2 = lcao_wave_sum(n2);
    float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      if (ch < V) lw[p * V + ch] = v[r] * inv;
    }
  }
}

extern "C" int lcao_edge_combine(const float* ft, const float* B, const float* Bv, int64_t n_edges, int V, float* lw,
                                 void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(ft); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(lw);
  dim3 grid(lcao_blocks(n_edges, 4, 32768)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int R = (V + 63) / 64;
  if (R == 1) hipLaunchKernelGGL(k_edge_combine<1>, grid, block, 0, st, ft, B, Bv, n_edges, V, lw);
  else if (R == 2) hipLaunchKernelGGL(k_edge_combine<2>, grid, block, 0, st, ft, B, Bv, n_edges, V, lw);
  else if (R <= 4) hipLaunchKernelGGL(k_edge_combine<4>, grid, block, 0, st, ft, B, Bv, n_edges, V, lw);
  else hipLaunchKernelGGL(k_edge_combine<8>, grid, block, 0, st, ft, B, Bv, n_edges, V, lw);
  return lcao_launch_status();
}

// h1 = SiLU(Ps[s] + Pt[t])
__global__ void k_pair_silu(const float* __restrict__ Ps, const float* __restrict__ Pt,
                            const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p, int64_t E, int V,
                            float* __restrict__ h1) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    const float* a = Ps + (int64_t)src_p[p] * V;
    const float* b = Pt + (int64_t)dst_p[p] * V;
    for (int c = lane; c < V; c += 64) h1[p * V + c] = lcao_silu(a[c] + b[c]);
  }
}

extern "C" int lcao_pair_silu(const float* Ps, const float* Pt, const int32_t* src_p, const int32_t* dst_p,
                              int64_t n_edges, int V, float* h1, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(h1);
  hipLaunchKernelGGL(k_pair_silu, dim3(lcao_blocks(n_edges, 4, 32768)), dim3(256), 0, (hipStream_t)stream, Ps, Pt,
                     src_p, dst_p, n_edges, V, h1);
  return lcao_launch_status();
}

// cst[p,d,c] = scale[d,c] * FE[z_t,d,c] * (1 + U[z_s,c] + Vt[z_t,c]) + shift[d,c]
// (EmbedCoeffs, eval-mode BatchNorm folded into scale/shift; embed.py:246-249).  One wave per edge row-block.
__global__ void k_embed_coeffs(const float* __restrict__ FE, const float* __restrict__ U,
                               const float* __restrict__ Vt, const float* __restrict__ scale,
                               const float* __restrict__ shift, const int64_t* __restrict__ z,
                               const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p, int64_t E, int D,
                               int C, float* __restrict__ cst) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    int64_t zs = z[src_p[p]], zt = z[dst_p[p]];
    const float* fe = FE + zt * D * C;
    float* out = cst + p * (int64_t)D * C;
    for (int c = lane; c < C; c += 64) {
      float zz = U[zs * C + c] + Vt[zt * C + c];
      for (int d = 0; d < D; ++d) {
        float e = fe[d * C + c];
        out[d * C + c] = (e + e * zz) * scale[d * C + c] + shift[d * C + c];
      }
    }
  }
}

// 16-byte form (C % 4 == 0): the D*C/4 float4 slots of an edge are spread over the wave, streaming stores
__global__ void k_embed_coeffs_v4(const float* __restrict__ FE, const float* __restrict__ U,
                                  const float* __restrict__ Vt, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const int64_t* __restrict__ z,
                                  const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p, int64_t E,
                                  int D, int C, float* __restrict__ cst) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int c4n = C >> 2, slots = D * c4n;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    const int64_t zs = z[src_p[p]], zt = z[dst_p[p]];
    const f4* fe = (const f4*)(FE + zt * D * C);
    const f4* u = (const f4*)(U + zs * C);
    const f4* v = (const f4*)(Vt + zt * C);
    f4* out = (f4*)(cst + p * (int64_t)D * C);
    for (int i = lane; i < slots; i += 64) {
      const int c4 = i % c4n;
      const f4 zz = u[c4] + v[c4];
      const f4 e = fe[i];
      const f4 o = (e + e * zz) * ((const f4*)scale)[i] + ((const f4*)shift)[i];
      __builtin_nontemporal_store(o, out + i);
    }
  }
}

extern "C" int lcao_embed_coeffs(const float* FE, const float* U, const float* Vt, const float* scale,
                                 const float* shift, const int64_t* z, const int32_t* src_p, const int32_t* dst_p,
                                 int64_t n_edges, int n_orb, int C, float* cst, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || C <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(FE); LCAO_CHECK_PTR(U); LCAO_CHECK_PTR(Vt); LCAO_CHECK_PTR(scale); LCAO_CHECK_PTR(shift);
  LCAO_CHECK_PTR(z); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(cst);
  const bool v4 = (C % 4 == 0) && (((uintptr_t)FE | (uintptr_t)U | (uintptr_t)Vt | (uintptr_t)scale |
                                     (uintptr_t)shift | (uintptr_t)cst) % 16 == 0);
  if (v4)
    hipLaunchKernelGGL(k_embed_coeffs_v4, dim3(lcao_blocks(n_edges, 4, 32768)), dim3(256), 0, (hipStream_t)stream, FE,
                       U, Vt, scale, shift, z, src_p, dst_p, n_edges, n_orb, C, cst);
  else
    hipLaunchKernelGGL(k_embed_coeffs, dim3(lcao_blocks(n_edges, 4, 32768)), dim3(256), 0, (hipStream_t)stream, FE, U,
                       Vt, scale, shift, z, src_p, dst_p, n_edges, n_orb, C, cst);
  return lcao_launch_status();
}

// literal three-body operator with explicit triplet lists (lcaonet.py:173-188): one wave per triplet
template <int R>
__global__ void k_triplet_literal(const float* __restrict__ cstf, const float* __restrict__ rb,
                                  const float* __restrict__ shb, const float* __restrict__ xk, int64_t ldxk,
                                  const int64_t* __restrict__ tri_k, const int64_t* __restrict__ e_ks,
                                  const float* __restrict__ mask, int64_t mse, int64_t msd, int64_t msv, int64_t T,
                                  int D, int V, int valence, float* __restrict__ w) {
  int lane = threadIdx.x & 63;
  int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int Vt = valence ? 2 * V : V;
  for (; t < T; t += nwave) {
    int64_t e = e_ks[t];
    const float* crow = cstf + e * (int64_t)D * Vt;
    float v[R];
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      float acc = 0.f, accv = 0.f;
      if (ch < V) {
        for (int d = 0; d < D; ++d) {
          float o = rb[e * D + d] * shb[t * D + d];
          acc += o * crow[d * Vt + ch];
          if (valence) accv += o * (crow[d * Vt + V + ch] * mask[e * mse + d * msd + ch * msv]);
        }
      }
      v[r] = acc + accv;
      n2 += v[r] * v[r];
    }
    n2 = lcao_wave_sum(n2);
    float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    const float* g = xk + tri_k[t] * ldxk;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ch = lane + 64 * r;
      if (ch < V) w[t * V + ch] = (v[r] * inv) * lcao_sigmoid(g[ch]);
    }
  }
}

extern "C" int lcao_triplet_literal(const float* cstf, const float* rb, const float* shb, const float* xk,
                                    int64_t ldxk, const int64_t* tri_k, const int64_t* e_ks, const float* mask,
                                    int64_t mask_se, int64_t mask_sd, int64_t mask_sv, int64_t n_trip, int n_orb,
                                    int V, int valence, float* w, void* stream) {
  if (n_trip < 0 || n_orb <= 0 || V <= 0) return LCAO_ERR_INVALID;
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  if (n_trip == 0) return LCAO_OK;
  LCAO_CHECK_PTR(cstf); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(shb); LCAO_CHECK_PTR(xk); LCAO_CHECK_PTR(tri_k);
  LCAO_CHECK_PTR(e_ks); LCAO_CHECK_PTR(w);
  if (valence) LCAO_CHECK_PTR(mask);
  dim3 grid(lcao_blocks(n_trip, 4, 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int R = (V + 63) / 64;
#define LAUNCH_TL(RR)                                                                                              \
  hipLaunchKernelGGL(k_triplet_literal<RR>, grid, block, 0, st, cstf, rb, shb, xk, ldxk, tri_k, e_ks, mask, mask_se, \
                     mask_sd, mask_sv, n_trip, n_orb, V, valence, w)
  if (R == 1) LAUNCH_TL(1);
  else if (R == 2) LAUNCH_TL(2);
  else if (R <= 4) LAUNCH_TL(4);
  else LAUNCH_TL(8);
#undef LAUNCH_TL
  return lcao_launch_status();
}

// ---------------------------------------------------------------------------------------------------
extern "C" int lcao_abi_version(void) { return LCAO_ABI_VERSION; }

extern "C" const char* lcao_error_string(int code) {
  if (code == LCAO_OK) return "ok";
  if (code == LCAO_ERR_INVALID) return "lcaonet_hip: invalid argument";
  if (code == LCAO_ERR_UNSUPPORTED) return "lcaonet_hip: shape not supported by this kernel";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "lcaonet_hip: unknown error";
}

// ---------------------------------------------------------------------------------------------------
// lcao_pair_contract — species-pair form of f_coeffs + radial contraction.
// cst[e] is a function of (z_s, z_t) only (embed.py:246-249) and every layer applies a row-wise MLP to the same
// cst, so f_coeffs(cst)[e] = Tf[z_s, z_t] with Tf a ((Z+1)^2, D, V) table (5.6 MB at Z=36: L2 / Infinity-Cache
// resident).  Per edge p:  A[l,:] = sum_{d in l} rb[e,d] * Tf[pair,d,:],  B = sum_l A,  gram = triangular factor of A A^T,
// Ag[l,:] = A[l,:] * gate[src,:].   One wave per edge, lanes over V (R values per lane); 2.1 KB written per edge.
// ---------------------------------------------------------------------------------------------------
// 16 lanes per edge (4 edges per wave); a lane owns CH = V/16 channels as float4 chunks: chunk c covers channels
// 64*c + 4*sub .. +3, so a 16-lane row reads / writes 256 contiguous bytes per chunk.
template <int V, int NL>
__global__ __launch_bounds__(256) void k_pair_contract(const float* __restrict__ Tf, const float* __restrict__ rb,
                                                       const int32_t* __restrict__ order,
                                                       const int32_t* __restrict__ l_of_d,
                                                       const int64_t* __restrict__ z,
                                                       const int32_t* __restrict__ src_p,
                                                       const int32_t* __restrict__ dst_p,
                                                       const float* __restrict__ gate, int64_t E, int D, int zdim,
                                                       float* __restrict__ Ag, float* __restrict__ B,
                                                       float* __restrict__ gram,
                                                       const int32_t* __restrict__ visit /* or null */) {
  constexpr int NC = (V + 63) / 64;  // float4 chunks per lane
  const int sub = threadIdx.x & 15, rowi = (threadIdx.x & 63) >> 4;
  int64_t p0 = ((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6) * 4;   // wave-uniform: 4 edges per wave
  const int64_t stride = (((int64_t)gridDim.x * blockDim.x) >> 6) * 4;
  for (; p0 < E; p0 += stride) {   // whole waves stay converged for the DPP reductions; rows past E are masked
    const int64_t pi = p0 + rowi;
    const bool valid = pi < E;
    // visit: the order in which edges are WALKED (e.g. sorted by species pair, so that the four edges of a wave and the
    // waves of a workgroup read the same 4 KB table block from L1 instead of L2); every read and write stays indexed by p
    const int64_t p = visit ? (int64_t)visit[valid ? pi : E - 1] : pi;
    const int64_t pc = valid ? p : (visit ? p : E - 1);
    const int64_t e = order ? order[pc] : pc;
    const int s = src_p[pc];
    const int64_t pair = z[s] * zdim + z[dst_p[pc]];
    const float* t = Tf + pair * (int64_t)D * V + 4 * sub;
    f32x4 a[NL][NC];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int c = 0; c < NC; ++c) a[l][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d) {
      const float w = rb[e * D + d];
      const int l = l_of_d[d];  // uniform
      f32x4 v[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c)
        v[c] = (64 * c + 4 * sub) < V ? *(const f32x4*)(t + d * V + 64 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ll = 0; ll < NL; ++ll)
        if (l == ll) {
#pragma unroll
          for (int c = 0; c < NC; ++c) a[ll][c] += w * v[c];
        }
    }
    // triangular factor of the per-l vectors (lcao_common.h: lcao_edge_factor): in-lane partials + 16-lane row reductions
    float g[10];
    {
      float wv[NL][4 * NC];
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k) wv[l][4 * c + k] = a[l][c][k];
      lcao_edge_factor<NL, 4 * NC>(wv, g, [](float t) { return lcao_row16_sum(t); });
    }
    if (valid) {
      if (sub < 10) {
        float gv = 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i) gv = (sub == i) ? g[i] : gv;
        gram[p * 10 + sub] = gv;
      }
      const float* gt = gate + (int64_t)s * V + 4 * sub;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if ((64 * c + 4 * sub) < V) {
          const f32x4 gg = *(const f32x4*)(gt + 64 * c);
          f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            // streaming stores: 2.4 GB per launch that nothing re-reads before it has left every cache
            __builtin_nontemporal_store(a[l][c] * gg, (f32x4*)(Ag + (p * NL + l) * V + 64 * c + 4 * sub));
            bsum += a[l][c];
          }
          __builtin_nontemporal_store(bsum, (f32x4*)(B + p * V + 64 * c + 4 * sub));
        }
      }
    }
  }
}

// any-V fallback of k_pair_contract: one wave per edge, lanes over V (R values per lane)
template <int R>
__global__ void k_pair_contract_any(const float* __restrict__ Tf, const float* __restrict__ rb,
                                    const int32_t* __restrict__ order, const int32_t* __restrict__ l_of_d,
                                    const int64_t* __restrict__ z, const int32_t* __restrict__ src_p,
                                    const int32_t* __restrict__ dst_p, const float* __restrict__ gate, int64_t E, int D,
                                    int V, int NL, int zdim, float* __restrict__ Ag, float* __restrict__ B,
                                    float* __restrict__ gram) {
  int lane = threadIdx.x & 63;
  int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (; p < E; p += nwave) {
    const int64_t e = order ? order[p] : p;
    const int s = src_p[p];
    const int64_t pair = z[s] * zdim + z[dst_p[p]];
    const float* t = Tf + pair * (int64_t)D * V;
    float a[4][R];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) a[l][r] = 0.f;
    for (int d = 0; d < D; ++d) {
      const float w = rb[e * D + d];
      const int l = l_of_d[d];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + 64 * r;
        const float v = ch < V ? w * t[d * V + ch] : 0.f;
        a[0][r] += (l == 0) ? v : 0.f; a[1][r] += (l == 1) ? v : 0.f;
        a[2][r] += (l == 2) ? v : 0.f; a[3][r] += (l == 3) ? v : 0.f;
      }
    }
    {
      float wv[4][R], g[10];   // triangular factor (rows >= n_l are zero vectors and give zero rows)
#pragma unroll
      for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int r = 0; r < R; ++r) wv[l][r] = a[l][r];
      lcao_edge_factor<4, R>(wv, g, [](float t) { return lcao_wave_sum(t); });
      if (lane < 10) {
        float gv = 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i) gv = (lane == i) ? g[i] : gv;
        gram[p * 10 + lane] = gv;
      }
    }
    const float* gt = gate + (int64_t)s * V;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + 64 * r;
      if (ch < V) {
        const float gg = gt[ch];
        float bsum = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l)
          if (l < NL) {
            Ag[(p * NL + l) * V + ch] = a[l][r] * gg;
            bsum += a[l][r];
          }
        B[p * V + ch] = bsum;
      }
    }
  }
}

static int launch_pair_contract(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                const int64_t* z, const int32_t* src_p, const int32_t* dst_p, const float* gate,
                                int64_t n_edges, int n_orb, int V, int n_l, int zdim, float* Ag, float* B, float* gram,
                                const int32_t* visit, void* stream);

extern "C" int lcao_pair_contract(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                  const int64_t* z, const int32_t* src_p, const int32_t* dst_p, const float* gate,
                                  int64_t n_edges, int n_orb, int V, int n_l, int zdim, float* Ag, float* B, float* gram,
                                  void* stream) {
  return launch_pair_contract(Tf, rb, order, l_of_d, z, src_p, dst_p, gate, n_edges, n_orb, V, n_l, zdim, Ag, B, gram,
                              nullptr, stream);
}

extern "C" int lcao_pair_contract_visit(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                        const int64_t* z, const int32_t* src_p, const int32_t* dst_p, const float* gate,
                                        const int32_t* visit, int64_t n_edges, int n_orb, int V, int n_l, int zdim,
                                        float* Ag, float* B, float* gram, void* stream) {
  if (V != 128 && V != 64 && V != 32) return LCAO_ERR_UNSUPPORTED;   // the any-V kernel walks edges in p order
  return launch_pair_contract(Tf, rb, order, l_of_d, z, src_p, dst_p, gate, n_edges, n_orb, V, n_l, zdim, Ag, B, gram,
                              visit, stream);
}

static int launch_pair_contract(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                                const int64_t* z, const int32_t* src_p, const int32_t* dst_p, const float* gate,
                                int64_t n_edges, int n_orb, int V, int n_l, int zdim, float* Ag, float* B, float* gram,
                                const int32_t* visit, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1 || zdim <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Tf); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(z); LCAO_CHECK_PTR(src_p);
  LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(Ag); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(gram);
  dim3 grid(lcao_blocks(n_edges, 16, 65536)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define PC_CASE(V_, NL_)                                                                                          \
  if (V == V_ && n_l == NL_) {                                                                                    \
    hipLaunchKernelGGL((k_pair_contract<V_, NL_>), grid, block, 0, st, Tf, rb, order, l_of_d, z, src_p, dst_p, gate, \
                       n_edges, n_orb, zdim, Ag, B, gram, visit);                                                 \
    return lcao_launch_status();                                                                                  \
  }
#define PC_V(V_) PC_CASE(V_, 1) PC_CASE(V_, 2) PC_CASE(V_, 3) PC_CASE(V_, 4)
  PC_V(128) PC_V(64) PC_V(32)
#undef PC_V
#undef PC_CASE
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  {
    dim3 g2(lcao_blocks(n_edges, 4, 65536));
    int R = (V + 63) / 64;
#define PCA(RR)                                                                                                  \
  hipLaunchKernelGGL(k_pair_contract_any<RR>, g2, block, 0, st, Tf, rb, order, l_of_d, z, src_p, dst_p, gate, n_edges, \
                     n_orb, V, n_l, zdim, Ag, B, gram)
    if (R == 1) PCA(1);
    else if (R == 2) PCA(2);
    else if (R <= 4) PCA(4);
    else PCA(8);
#undef PCA
  }
  return lcao_launch_status();
}
