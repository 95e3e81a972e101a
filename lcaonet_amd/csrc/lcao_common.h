// lcao_common.h — shared device helpers for the LCAONet hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcaonet_hip.h"

#define LCAO_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LCAO_CHECK_PTR(p) \
  do {                    \
    if ((p) == nullptr) return LCAO_ERR_INVALID; \
  } while (0)
#define LCAO_CHECK_SIZE(n) \
  do {                     \
    if ((n) < 0 || (n) >= (int64_t(1) << 31)) return LCAO_ERR_INVALID; \
  } while (0)

static inline int lcao_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LCAO_OK : int(e);
}

static inline unsigned lcao_blocks(int64_t n, int per_block, int64_t cap = (int64_t(1) << 30)) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return unsigned(b);
}

// Zero-fill as a KERNEL launch.  The library issues no hipMemsetAsync: inside a captured HIP graph a memset NODE faulted on
// replay ("Memory access fault by GPU") as soon as the process had mapped a new allocator segment after the capture
// (profiles/r03/tile_queue_suspects.txt, ROCm 7.2); kernel nodes do not.
static __global__ void k_lcao_zero_words(uint32_t* __restrict__ p, int64_t n_words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline int lcao_zero_async(void* p, size_t bytes, hipStream_t st) {   // bytes % 4 == 0
  if (bytes == 0) return LCAO_OK;
  const int64_t n_words = (int64_t)(bytes / 4);
  hipLaunchKernelGGL(k_lcao_zero_words, dim3(lcao_blocks(n_words, 256, 4096)), dim3(256), 0, st, (uint32_t*)p, n_words);
  return lcao_launch_status();
}

__device__ __forceinline__ float lcao_silu(float x) {
  // x * sigmoid(x) in 5 VALU ops: v_mul, v_exp_f32, v_add, v_rcp_f32, v_mul (each transcendental <= 1 ulp).
  // A plain `x / (1 + expf(-x))` compiles to the 10+-instruction IEEE division sequence (v_div_scale/fmas/fixup),
  // which at 64 activations per lane per orbital step left the matrix pipe idle a third of the time.
  // Tails: x -> -inf gives exp = inf, rcp = 0, result -0 (never NaN for finite x).
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

__device__ __forceinline__ float lcao_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float lcao_silu_grad(float x) {
  // d/dx [x sigmoid(x)] = s (1 + x (1 - s))
  const float s = lcao_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float lcao_act(float x, int act) {
  if (act == LCAO_ACT_SILU) return lcao_silu(x);
  if (act == LCAO_ACT_SIGMOID) return lcao_sigmoid(x);
  return x;
}

// Y_l^0(cos theta), l <= 3: the closed forms the reference's sympy pipeline emits (lcaonet/nn/shbf.py:28-42,54-63)
__device__ __forceinline__ float lcao_y_l0(float c, int l) {
  switch (l) {
    case 0: return 0.28209479177387814f;
    case 1: return 0.48860251190292f * c;
    case 2: return 0.94617469575756f * c * c - 0.31539156525252f;
    default: return 0.373176332590115f * c * (5.0f * c * c - 3.0f);
  }
}

// d/dc of Y_l^0(c)
__device__ __forceinline__ float lcao_y_l0_grad(float c, int l) {
  switch (l) {
    case 0: return 0.0f;
    case 1: return 0.48860251190292f;
    case 2: return 2.0f * 0.94617469575756f * c;
    default: return 0.373176332590115f * (15.0f * c * c - 3.0f);
  }
}

// radial cutoff envelopes (lcaonet/nn/cutoff.py:32-34, 41-43, 62-67).  The reference's expanded forms cancel
// catastrophically towards rc (1 - 21 x^5 + 35 x^6 - 15 x^7 has terms of 35 and a value of 3.5e-5 at
// x = 0.99), so its fp32 result carries up to 7e-6 (of an orbital's maximum) of rounding noise against its own fp64 run.
// Both polynomials have a triple root at x = 1; they are evaluated here in the factored, cancellation-free form
//     1 - 21 x^5 + 35 x^6 - 15 x^7 = (1 - x)^3 (1 + 3x + 6x^2 + 10x^3 + 15x^4)          (EnvelopeCutoff, p = 5)
//     1 -  6 x^5 + 15 x^4 - 10 x^3 = (1 - x)^3 (1 + 3x + 6x^2)                          (PolynomialCutoff)
//     0.5 (cos(pi x) + 1)          = cos^2(pi x / 2)                                    (CosineCutoff)
// which is within 3e-7 of the fp64 value everywhere, i.e. the distance to the reference's fp32 tables is the
// reference's own noise (tests/test_gpu_parity.py::test_basis_kernels_vs_reference_tables, flat 1e-5 on the whole grid).
__device__ __forceinline__ float lcao_cutoff_eval(float r, float rc, int kind) {
  if (kind == LCAO_CUTOFF_NONE) return 1.0f;  // user-defined envelope applied by the caller
  if (!(r <= rc)) return 0.0f;
  const float x = r / rc;
  const float om = 1.0f - x;
  if (kind == LCAO_CUTOFF_ENVELOPE) {
    const float q = (((15.0f * x + 10.0f) * x + 6.0f) * x + 3.0f) * x + 1.0f;
    return om * om * om * q;
  } else if (kind == LCAO_CUTOFF_POLYNOMIAL) {
    const float q = (6.0f * x + 3.0f) * x + 1.0f;
    return om * om * om * q;
  } else {
    const float c = cosf(1.57079632679489661923f * x);
    return c * c;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-edge triangular factor (the 10-float record the three-body kernels read).  lcaonet.py:179-184 normalises every
// per-triplet vector v = sum_l y_l A_l directly; the fused kernels need |v| from per-EDGE data.  Round 1-2 kept the Gram
// matrix G = A A^T and took |v|^2 = y^T G y: a difference of large terms when the A_l are correlated and v nearly cancels
// (error eps S^2 / |v|^2, S = sum |y_l| |A_l|: 6x the literal form's on a model with rank-1 f_coeffs.2,
// tests/test_gpu_conditioning.py).  Now the record is R (upper triangular, R^T R = A A^T) from a modified Gram-Schmidt
// sweep over the vectors themselves, and |v|^2 = |R y|^2 = sum_i (sum_{j>=i} R[i][j] y_j)^2 — a sum of squares whose
// inner sums cancel exactly as v does (error eps S / |v|, the literal form's).  Entry (i, j >= i) sits at lcao_tri_idx(i, j).
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int lcao_tri_idx(int l, int l2) {  // l <= l2, row-major upper triangle of a 4x4
  return l * 4 - (l * (l - 1)) / 2 + (l2 - l);
}

// w[l][0..NVAL): this lane's values of the n_l vectors of its edge (DESTROYED); reduce(x) sums x over the lanes that share
// the edge.  R[lcao_tri_idx(i, j)] for i <= j < NL, zero elsewhere.  A zero (or numerically dependent) vector gives a
// zero row: its later coefficients are 0, never NaN.
template <int NL, int NVAL, typename Reduce>
__device__ __forceinline__ void lcao_edge_factor(float (&w)[NL][NVAL], float (&R)[10], Reduce reduce) {
#pragma unroll
  for (int i = 0; i < 10; ++i) R[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    float d[NL];
#pragma unroll
    for (int j = i; j < NL; ++j) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NVAL; ++k) t += w[i][k] * w[j][k];
      d[j] = reduce(t);
    }
    const float rii = sqrtf(d[i]);
    const float inv = rii > 0.f ? 1.0f / rii : 0.f;
    R[lcao_tri_idx(i, i)] = rii;
#pragma unroll
    for (int j = i + 1; j < NL; ++j) {
      const float rij = d[j] * inv;
      R[lcao_tri_idx(i, j)] = rij;
      const float f = rij * inv;
#pragma unroll
      for (int k = 0; k < NVAL; ++k) w[j][k] -= f * w[i][k];
    }
  }
}

// m = |R y|^2 (= y^T A A^T y) and Gy = R^T (R y) (= A A^T y, half the gradient of m with respect to y)
template <int NL>
__device__ __forceinline__ float lcao_factor_quadratic(const float (&R)[10], const float (&y)[4], float (&Gy)[4]) {
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int j = i; j < NL; ++j) t[i] += R[lcao_tri_idx(i, j)] * y[j];
    m += t[i] * t[i];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    Gy[j] = 0.f;
#pragma unroll
    for (int i = 0; i <= j; ++i)
      if (i < NL && j < NL) Gy[j] += R[lcao_tri_idx(i, j)] * t[i];
  }
  return m;
}

// sum within each row of 16 lanes (DPP: two quad permutes, row_half_mirror, row_mirror), result in every lane of the row
__device__ __forceinline__ float lcao_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// sum across the 64 lanes of a wave, result in every lane (DPP within rows of 16, then readlane across rows)
__device__ __forceinline__ float lcao_wave_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
