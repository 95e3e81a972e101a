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
#include <cstdlib>

#include "lcao_mfma.h"

#ifndef TBWD_LINE_STORES
#define TBWD_LINE_STORES 1
#endif

// Diagnostic hook (tools/clock_probe.py): when set, lane 0 of every wave of k_coeff_mlp stores
// {delta s_memtime (shader cycles), delta s_memrealtime (100 MHz ticks)} of its whole tile loop into this buffer
// (4 x uint64 per wave).  NULL in production: the guarded stamps then never execute.
static unsigned long long* g_clock_probe = nullptr;
extern "C" void lcao_debug_set_clock_probe(void* buf) { g_clock_probe = (unsigned long long*)buf; }

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

// N % 128 == 0 and many rows: one workgroup of 8 waves owns 128 output columns (weights = 128 x (K + 4) floats of LDS, two
// workgroups per CU) and every X row is read once; the next tile's rows are requested into the operand registers as soon as
// the GEMM has consumed them, so they are in flight under the epilogue and the stores (as in k_edge_lcao_weight).
template <int KT>
__global__ __launch_bounds__(512) void k_dense_mfma_wide(const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ R, int64_t ldr, float* __restrict__ Y,
                                                         int64_t ldy, int64_t M, int N, int act) {
  constexpr int K = KT * 16;
  extern __shared__ float lds[];
  const int n0 = blockIdx.x * 128;
  stage_weights(lds, W + (int64_t)n0 * K, 128, K, K);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (M + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.y * 8;
  int64_t tile = (int64_t)blockIdx.y * 8 + wave;
  f32x4v x[KT];
  if (tile < n_tiles) {
    const float* xr = X + min(tile * 16 + r, M - 1) * ldx + 4 * q;
#pragma unroll
    for (int a = 0; a < KT; ++a) x[a] = *(const f32x4v*)(xr + 16 * a);
  }
  for (; tile < n_tiles; tile += stride) {
    asm volatile("" ::: "memory");
    const int64_t row = tile * 16 + r;
    const bool valid = row < M;
    f32x4v acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<KT, 8>(lds, K + 4, r, q, x, acc);
    if (tile + stride < n_tiles) {
      const float* xr = X + min((tile + stride) * 16 + r, M - 1) * ldx + 4 * q;
#pragma unroll
      for (int a = 0; a < KT; ++a) x[a] = *(const f32x4v*)(xr + 16 * a);
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int n = n0 + 16 * t + 4 * q;
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

// Few output columns (the last layer of the output head: out_size = 1 or 2): 16 lanes per row, each lane a float4 stride of
// the row, one DPP row reduction per column — the lanes-over-columns fallback above leaves 63 of 64 lanes idle there.
__global__ __launch_bounds__(256) void k_dense_narrow(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ R,
                                                      int64_t ldr, float* __restrict__ Y, int64_t ldy, int64_t M, int K,
                                                      int N, int act) {
  const int sub = threadIdx.x & 15;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t rc = row < M ? row : M - 1;
  const float* xr = X + rc * ldx;
  for (int n = 0; n < N; ++n) {
    const float* wr = W + (int64_t)n * K;
    float acc = 0.f;
    for (int k = 4 * sub; k < K; k += 64) {
      const f32x4v x = *(const f32x4v*)(xr + k), w = *(const f32x4v*)(wr + k);
      acc += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
    }
    acc = lcao_row16_sum(acc);
    if (sub == 0 && row < M) {
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
  if constexpr (KT <= 8) if (N % 128 == 0 && M >= 16384) {   // wide form: whole 128-column slices, X rows read once
    const size_t lds_w = size_t(128) * (K + 4) * sizeof(float);
    static bool attr_w = false;
    if (!attr_w && lds_w > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)k_dense_mfma_wide<KT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_w);
      if (e != hipSuccess) return int(e);
      attr_w = true;
    }
    const int nsl = N / 128;
    int64_t groups = ((M + 15) / 16 + 7) / 8;
    int64_t cap = (int64_t)lcao_num_cu() * 2 / nsl;
    if (cap < 1) cap = 1;
    if (groups > cap) groups = cap;
    hipLaunchKernelGGL(k_dense_mfma_wide<KT>, dim3(nsl, (unsigned)groups), dim3(512), lds_w, st, X, ldx, W, bias, R, ldr,
                       Y, ldy, M, N, act);
    return lcao_launch_status();
  }
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
  if (N <= 4 && K % 4 == 0 && ldx % 4 == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)W % 16 == 0) {
    hipLaunchKernelGGL(k_dense_narrow, dim3(lcao_blocks(M, 16)), dim3(256), 0, st, X, ldx, W, bias, residual, ldr, Y, ldy,
                       M, K, N, act);
    return lcao_launch_status();
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
template <int C, int V, int NL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_coeff_mlp(const float* __restrict__ cst, const float* __restrict__ rb,
                                                   const int32_t* __restrict__ order,
                                                   const int32_t* __restrict__ cst_row /* p -> row of cst, or null */,
                                                   const int32_t* __restrict__ l_of_d,
                                                   const float* __restrict__ wmul /* (E,D) or null */,
                                                   const float* __restrict__ W1, const float* __restrict__ W2,
                                                   int64_t E, int D, int accumulate, float* __restrict__ A,
                                                   float* __restrict__ Bout, const float* __restrict__ gate,
                                                   const int32_t* __restrict__ src_p, float* __restrict__ Ag,
                                                   float* __restrict__ gram, unsigned* __restrict__ tile_counter,
                                                   unsigned long long* __restrict__ clk) {
  constexpr int KT1 = C / 16, MT = V / 16;
  extern __shared__ float lds[];
  float* w1s = lds;
  float* w2s = lds + V * (C + 4);
  // SiLU(h) = h / (1 + 2^(-h log2 e)): W1 is staged pre-multiplied by log2 e, so layer 1 yields h' = h log2 e and the
  // activation is g(h') = h' rcp(1 + exp2(-h')) = SiLU(h) log2 e — four instructions (v_exp_f32 takes the negation as
  // a source modifier) instead of five.  The factor cancels through the chain: layer 2 sees W2 g(h') = (W2 SiLU(h))
  // log2 e = o', the second activation gives SiLU(o) log2 e, and 1 / log2 e is folded into the orbital weight rb[e,d].
  stage_weights_scaled(w1s, W1, V, C, C, LCAO_LOG2E);
  stage_weights(w2s, W2, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  constexpr int WAVES = THREADS / 64;
  unsigned long long t0 = 0, r0 = 0;
  if (clk) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  // Dynamic tile queue: the first tile of a wave is static, every further one is drawn from a device-scope
  // counter.  The queue re-arms ITSELF (lcao_tile_queue_leave: the last wave to leave zeroes it), so a launch needs no
  // memset before it: a hipMemsetAsync NODE in a captured HIP graph faults on replay once a new allocator segment has been
  // mapped (profiles/r03/tile_queue_suspects.txt; the library issues none, lcao_common.h: lcao_zero_async).  With a static split the two waves that share a SIMD finished 1.5 ms apart
  // (the older wave wins MFMA arbitration, completes its share early and leaves the other to run alone);
  // drawing tiles keeps both resident until the queue is empty.  Every wave exits once it draws tile >= n_tiles.
#if LCAO_ABL & 16
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef LCAO_STAGGER
  // one-time stagger of the second wave of every SIMD (waves 4-7) by LCAO_STAGGER x ~8k cycles
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {
#pragma unroll
    for (int i = 0; i < LCAO_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const int64_t first_dynamic = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  for (; tile < n_tiles;) {
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const int64_t e = order ? order[pc] : pc;
    // cst_row: the coefficient rows are GATHERED here (e.g. from the (z_s, z_t) table all edges of a species pair share,
    // LCAOEmbedding.forward_pair_table) instead of being materialised per edge; every (edge, orbital) row still runs
    // through both layers below
    const int64_t ec = cst_row ? (int64_t)cst_row[pc] : e;
    const float* xrow = cst + ec * (int64_t)D * C + 4 * q;
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
      // the wave in an MFMA stream outranks its SIMD partner's VALU phase for the shared vector issue port
      // (+2 % measured, tools/k2_ablate.py); LCAO_ABL & 8 removes it for A/B runs
#if !(LCAO_ABL & 8)
      __builtin_amdgcn_s_setprio(1);
#endif
      mfma_layer<KT1, MT>(w1s, C + 4, r, q, x, h);
#if !(LCAO_ABL & 8)
      __builtin_amdgcn_s_setprio(0);
#endif
      // prefetch the next orbital's coefficient row while layer 2 runs
      if (d + 1 < D && !(LCAO_ABL & 2)) {
#pragma unroll
        for (int a = 0; a < KT1; ++a) x[a] = *(const f32x4v*)(xrow + (d + 1) * C + 16 * a);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) h[m] = K2_ACT4(h[m]);
      f32x4v o[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) o[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
#if !(LCAO_ABL & 8)
      __builtin_amdgcn_s_setprio(1);
#endif
      mfma_layer<MT, MT>(w2s, V + 4, r, q, h, o);
#if !(LCAO_ABL & 8)
      __builtin_amdgcn_s_setprio(0);
#endif
      float w = rbrow[d] * LCAO_INV_LOG2E;
      if (wmul) w *= wmul[e * D + d];
      const int l = l_of_d[d];  // wave-uniform
#pragma unroll
      for (int ll = 0; ll < NL; ++ll) {
        if (l == ll) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[ll][m] += w * K2_ACT4(o[m]);
        }
      }
    }
    // Bout of a pass is the sum over l of THIS pass' accumulators (B in pass 1, Bv in the valence pass)
    f32x4v bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      bsum[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int l = 0; l < NL; ++l) bsum[m] += acc[l][m];
    }
    if (accumulate) {  // valence pass: add the first pass' A (written ungated) before finalising
      const float* Ain = A + pc * (int64_t)NL * V + 4 * q;
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[l][m] += *(const f32x4v*)(Ain + l * V + 16 * m);
    }
    if (valid) {
      float* Ap = A + p * (int64_t)NL * V + 4 * q;
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
#pragma unroll
          for (int l = 0; l < NL; ++l) *(f32x4v*)(Ap + l * V + 16 * m) = acc[l][m];
        }
      }
    }
    if (gram) {
      // triangular factor of the (ungated) per-l vectors (lcao_common.h: lcao_edge_factor), computed IN PLACE on the
      // accumulators after their last use above: in-lane partial sums over this lane's 4*MT features, the four quarters of
      // the edge summed with two cross-lane steps.  Feeds the per-triplet norm.
      float(&wv)[NL][4 * MT] = reinterpret_cast<float(&)[NL][4 * MT]>(acc);
      float g[10];
      lcao_edge_factor<NL, 4 * MT>(wv, g, [](float t) {
        t += __shfl_xor(t, 16);
        return t + __shfl_xor(t, 32);
      });
      if (valid && q == 0) {
        float* gp = gram + p * 10;
#pragma unroll
        for (int i = 0; i < 10; ++i) gp[i] = g[i];
      }
    }
    if (!tile_counter) {   // static striding (LCAO_K2_STATIC=1: diagnostic)
      tile += first_dynamic;
      continue;
    }
    unsigned nxt = 0;
    if (lane == 0) nxt = atomicAdd(tile_counter, 1u);
    tile = first_dynamic + (int64_t)__builtin_amdgcn_readfirstlane(nxt);
  }
  // every wave passes here exactly once, after its last draw: the last one re-arms the queue for the next launch
  if (tile_counter && lane == 0) lcao_tile_queue_leave(tile_counter, gridDim.x * WAVES);
  if (clk && lane == 0) {
    unsigned long long* o = clk + 4 * ((size_t)blockIdx.x * WAVES + wave);
    o[0] = __builtin_amdgcn_s_memtime() - t0;
    o[1] = __builtin_amdgcn_s_memrealtime() - r0;
    o[2] = r0;  // absolute 100 MHz stamps: loop start / end (kernel span = max end - min start)
    o[3] = r0 + o[1];
  }
}

// block size of k_coeff_mlp: 512 threads = 2 waves per SIMD sharing the LDS weights (one wave's SiLU / accumulate
// VALU work runs under the other's MFMAs) when the register budget allows (n_l <= 3: <= 256 VGPRs), else 256.
// LCAO_K2_THREADS=256|512 overrides (benchmarking).
static int coeff_mlp_threads(int n_l) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("LCAO_K2_THREADS");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 256 || forced == 512) return forced;
  return n_l <= 3 ? 512 : 256;
}

template <int C, int V, int NL, int THREADS>
static int launch_coeff_mlp_t(const float* cst, const float* rb, const int32_t* order, const int32_t* cst_row,
                              const int32_t* l_of_d, const float* wmul, const float* W1, const float* W2, int64_t E, int D, int accumulate,
                              float* A, float* Bout, const float* gate, const int32_t* src_p, float* Ag,
                              float* gram, unsigned* counter, hipStream_t st) {
  size_t lds = (size_t(V) * (C + 4) + size_t(V) * (V + 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_coeff_mlp<C, V, NL, THREADS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return int(e);
    attr_set = true;
  }
  constexpr int WAVES = THREADS / 64;
  int64_t tiles = (E + 15) / 16;
  int64_t groups = (tiles + WAVES - 1) / WAVES;
  int64_t per_cu = (lds > 80 * 1024) ? 1 : 2;
  int64_t cap = (int64_t)lcao_num_cu() * per_cu;
  if (groups > cap) groups = cap;
  static int k2_static = -1;
  if (k2_static < 0) { const char* e_ = getenv("LCAO_K2_STATIC"); k2_static = (e_ && atoi(e_)) ? 1 : 0; }
  if (k2_static) counter = nullptr;   // diagnostic: static striding, no queue
#ifdef LCAO_DIAG_MEMSET_REARM   // diagnostic build only: the queue re-armed by a memset in front of the launch (a memset NODE in a capture)
  if (counter && hipMemsetAsync(counter, 0, 2 * sizeof(unsigned), st) != hipSuccess) return LCAO_ERR_INVALID;
#endif
  hipLaunchKernelGGL((k_coeff_mlp<C, V, NL, THREADS>), dim3((unsigned)groups), dim3(THREADS), lds, st, cst, rb, order,
                     cst_row, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, gate, src_p, Ag, gram, counter, g_clock_probe);
  return lcao_launch_status();
}

template <int C, int V, int NL>
static int launch_coeff_mlp(const float* cst, const float* rb, const int32_t* order, const int32_t* cst_row,
                            const int32_t* l_of_d, const float* wmul, const float* W1, const float* W2, int64_t E, int D, int accumulate,
                            float* A, float* Bout, const float* gate, const int32_t* src_p, float* Ag, float* gram,
                            unsigned* counter, hipStream_t st) {
  if (coeff_mlp_threads(NL) == 512)
    return launch_coeff_mlp_t<C, V, NL, 512>(cst, rb, order, cst_row, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, gate,
                                             src_p, Ag, gram, counter, st);
  return launch_coeff_mlp_t<C, V, NL, 256>(cst, rb, order, cst_row, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, gate,
                                           src_p, Ag, gram, counter, st);
}

template <int C, int V>
static int dispatch_coeff_mlp_nl(int n_l, const float* cst, const float* rb, const int32_t* order,
                                 const int32_t* cst_row, const int32_t* l_of_d, const float* wmul, const float* W1, const float* W2,
                                 int64_t E, int D, int accumulate, float* A, float* Bout, const float* gate,
                                 const int32_t* src_p, float* Ag, float* gram, unsigned* counter, hipStream_t st) {
#define CM_NL(NL_) \
  case NL_:        \
    return launch_coeff_mlp<C, V, NL_>(cst, rb, order, cst_row, l_of_d, wmul, W1, W2, E, D, accumulate, A, Bout, gate, src_p, Ag, gram, counter, st)
  switch (n_l) {
    CM_NL(1); CM_NL(2); CM_NL(3); CM_NL(4);
#undef CM_NL
    default: return LCAO_ERR_INVALID;
  }
}

extern "C" size_t lcao_lds_bytes_coeff_mlp(int C, int V, int valence) {
  (void)valence;
  return (size_t(V) * (C + 4) + size_t(V) * (V + 4)) * sizeof(float);
}

extern "C" int lcao_coeff_mlp_contract(const float* cst, const float* rb, const int32_t* order,
                                       const int32_t* cst_row, const int32_t* l_of_d, const float* valence_ed, const float* W1,
                                       const float* W2, const float* gate, const int32_t* src_p, int64_t n_edges,
                                       int n_orb, int C, int V, int n_l, int valence, float* A, float* B, float* Bv,
                                       float* Ag, float* gram, void* workspace, size_t ws_bytes, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || C <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(cst); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(W1); LCAO_CHECK_PTR(W2);
  LCAO_CHECK_PTR(B);
  if (valence) { LCAO_CHECK_PTR(valence_ed); LCAO_CHECK_PTR(Bv); LCAO_CHECK_PTR(A); }
  if (Ag) { LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(gram); } else { LCAO_CHECK_PTR(A); }
  LCAO_CHECK_PTR(workspace);
  if (ws_bytes < 4 * sizeof(unsigned)) return LCAO_ERR_INVALID;  // {drawn, left} per pass (lcao_tile_queue_leave)
  unsigned* ctr = (unsigned*)workspace;
  hipStream_t st = (hipStream_t)stream;
  int rc = LCAO_ERR_UNSUPPORTED;
  // no valence: one pass, finalised (gated + Gram) directly.  valence: pass 1 writes the raw A of the plain half
  // [c], pass 2 runs the [cv] half of W2 with rb * valence as the orbital weight, adds pass 1's A and finalises.
#define CM_CASE(C_, V_)                                                                                            \
  if (C == C_ && V == V_) {                                                                                        \
    if (!valence)                                                                                                  \
      return dispatch_coeff_mlp_nl<C_, V_>(n_l, cst, rb, order, cst_row, l_of_d, nullptr, W1, W2, n_edges, n_orb, 0, \
                                           A, B, gate, src_p, Ag, Ag ? gram : nullptr, ctr, st);                   \
    rc = dispatch_coeff_mlp_nl<C_, V_>(n_l, cst, rb, order, cst_row, l_of_d, nullptr, W1, W2, n_edges, n_orb, 0, A, \
                                       B, nullptr, nullptr, nullptr, nullptr, ctr, st);                            \
    if (rc != LCAO_OK) return rc;                                                                                  \
    return dispatch_coeff_mlp_nl<C_, V_>(n_l, cst, rb, order, cst_row, l_of_d, valence_ed, W1, W2 + (size_t)V_ * V_, \
                                         n_edges, n_orb, 1, A, Bv, gate, src_p, Ag, Ag ? gram : nullptr, ctr + 2,  \
                                         st);                                                                      \
  }
  CM_CASE(128, 128)
  CM_CASE(64, 64)
  CM_CASE(32, 32)
  CM_CASE(64, 128)   // emb_size_coeff != emb_size_conv: every pair of {32, 64, 128}
  CM_CASE(128, 64)
  CM_CASE(32, 64)
  CM_CASE(64, 32)
  CM_CASE(32, 128)
  CM_CASE(128, 32)
#undef CM_CASE
  return rc;
}

// ===================================================================================================
// lcao_edge_lcao_weight — lcaonet.py:192 (f_three), :192-204 (inject + contract + F.normalize) and the
// basis_weight product of :209, per 16-edge tile, register to register:
//   ft = W3 . tbw^T ;  u = B * (1 + ft) ;  lw = u / max(|u|, 1e-12) ;  bw = Wb . lw^T
// The row norm is an in-lane sum over the lane's 4*MT features plus two cross-quarter shuffles.
// ===================================================================================================
template <int V, int THREADS>
__global__ __launch_bounds__(THREADS) void k_edge_lcao_weight(const float* __restrict__ tbw,
                                                              const float* __restrict__ B,
                                                              const float* __restrict__ W3,
                                                              const float* __restrict__ Wb, int64_t E,
                                                              float* __restrict__ bw,
                                                              float* __restrict__ ft_out /* or null */) {
  constexpr int MT = V / 16, WAVES = THREADS / 64;
  extern __shared__ float lds[];
  float* w3s = lds;
  float* wbs = lds + V * (V + 4);
  stage_weights(w3s, W3, V, V, V);
  stage_weights(wbs, Wb, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  // the inputs of the next tile are fetched into the registers of the operand that has just been consumed
  // (x after the first GEMM, b after the combine), so loads are in flight under the second GEMM and the store
  f32x4v x[MT], b[MT];
  if (tile < n_tiles) {
    const int64_t pc = min(tile * 16 + r, E - 1);
#pragma unroll
    for (int a = 0; a < MT; ++a) x[a] = *(const f32x4v*)(tbw + pc * V + 16 * a + 4 * q);
#pragma unroll
    for (int a = 0; a < MT; ++a) b[a] = *(const f32x4v*)(B + pc * V + 16 * a + 4 * q);
  }
  for (; tile < n_tiles; tile += stride) {
    asm volatile("" ::: "memory");  // keep the LDS weight fragments from being hoisted out of the tile loop
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t nxt = tile + stride;
    const int64_t pn = min(nxt * 16 + r, E - 1);
    f32x4v ft[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ft[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<MT, MT>(w3s, V + 4, r, q, x, ft);
    if (ft_out && valid) {  // kept for the reverse sweep (forces / training)
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(ft_out + p * V + 16 * m + 4 * q) = ft[m];
    }
    if (nxt < n_tiles) {
#pragma unroll
      for (int a = 0; a < MT; ++a) x[a] = *(const f32x4v*)(tbw + pn * V + 16 * a + 4 * q);
    }
    float n2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = b[m][j] + b[m][j] * ft[m][j];
        ft[m][j] = u;
        n2 += u * u;
      }
    if (nxt < n_tiles) {
#pragma unroll
      for (int a = 0; a < MT; ++a) b[a] = *(const f32x4v*)(B + pn * V + 16 * a + 4 * q);
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
                                                           float* __restrict__ msg,
                                                           float* __restrict__ pre2_out /* or null */) {
  constexpr int MT = V / 16;
  extern __shared__ float lds[];
  stage_weights(lds, Wn2, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  // software pipeline: the node rows of tile t+1 are gathered while tile t is on the matrix cores; their indices
  // were read one iteration earlier
  f32x4v h[MT];
  int32_t sn = 0, dn = 0;
  if (tile < n_tiles) {
    const int64_t pc = min(tile * 16 + r, E - 1);
    const float* ps = Ps + (int64_t)src_p[pc] * V + 4 * q;
    const float* pt = Pt + (int64_t)dst_p[pc] * V + 4 * q;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      f32x4v s = *(const f32x4v*)(ps + 16 * a) + *(const f32x4v*)(pt + 16 * a);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[a][j] = lcao_silu(s[j]);
    }
    const int64_t pn = min((tile + stride) * 16 + r, E - 1);
    sn = src_p[pn];
    dn = dst_p[pn];
  }
  for (; tile < n_tiles; tile += stride) {
    asm volatile("" ::: "memory");
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const bool more = tile + stride < n_tiles;
    f32x4v ns[MT], nt[MT];
    if (more) {
      const float* ps = Ps + (int64_t)sn * V + 4 * q;
      const float* pt = Pt + (int64_t)dn * V + 4 * q;
#pragma unroll
      for (int a = 0; a < MT; ++a) ns[a] = *(const f32x4v*)(ps + 16 * a);
#pragma unroll
      for (int a = 0; a < MT; ++a) nt[a] = *(const f32x4v*)(pt + 16 * a);
      const int64_t pn2 = min((tile + 2 * stride) * 16 + r, E - 1);
      sn = src_p[pn2];
      dn = dst_p[pn2];
    }
    f32x4v o[MT], wv[MT];
    const int64_t pw = valid ? p : E - 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) wv[m] = *(const f32x4v*)(bw + pw * V + 16 * m + 4 * q);  // in flight under the GEMM
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);  // bias as the initial accumulator
    __builtin_amdgcn_s_setprio(1);  // the wave on the matrix cores outranks its SIMD partner's SiLU phase
    mfma_layer<MT, MT>(lds, V + 4, r, q, h, o);
    __builtin_amdgcn_s_setprio(0);
    if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4v w = wv[m];
        if (pre2_out) *(f32x4v*)(pre2_out + p * V + 16 * m + 4 * q) = o[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] *= lcao_silu(o[m][j]);
        *(f32x4v*)(msg + p * V + 16 * m + 4 * q) = w;
      }
    }
    if (more) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        f32x4v s = ns[a] + nt[a];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[a][j] = lcao_silu(s[j]);
      }
    }
  }
}

// ===================================================================================================
// lcao_edge_node_message_agg — lcao_edge_node_message with the node scatter of lcaonet.py:207-213 fused in:
//   agg[s, :] = sum_{p in Out(s)} bw[p, :] * SiLU(Wn2 . SiLU(Ps[s] + Pt[dst(p)]) + bn2)
// Edges are sorted by centre, so a wave takes CHUNKS OF WHOLE CENTRE NODES (`chunk_nodes` consecutive nodes, ~10 tiles of
// 16 edges; only a chunk's last tile is partly empty) and walks their edges tile by tile.  After the GEMM a lane (r, q)
// holds 32 message features of edge r; the rows of one node are summed across r with a masked 16-lane DPP reduction
// (a tile spans 1-3 nodes), a node that continues into the next tile is carried in registers, and a finished node row is
// stored once.  msg (E, V) — 1 KB per edge and layer — and the lcao_segment_sum launch disappear.  Fixed order (tile by
// tile, fixed DPP tree): bitwise reproducible; no atomics.  `agg` must be zero-filled by the launcher (edgeless nodes).
// ===================================================================================================
template <int V>
__global__ __launch_bounds__(256) void k_edge_node_message_agg(
    const float* __restrict__ bw, const float* __restrict__ Ps, const float* __restrict__ Pt,
    const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p, const int32_t* __restrict__ out_ptr,
    const float* __restrict__ Wn2, const float* __restrict__ bn2, int n_nodes, int chunk_nodes,
    const int32_t* __restrict__ hint, float* __restrict__ pre2_out /* (E, V) or null */, float* __restrict__ agg) {
  constexpr int MT = V / 16;
  extern __shared__ float lds[];
  if (hint && *hint != 0) return;   // even degrees: k_edge_node_message_cols (launched next) writes every row of agg
  stage_weights(lds, Wn2, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n_chunks = (n_nodes + chunk_nodes - 1) / chunk_nodes;
  const int cstride = gridDim.x * 4;
  // first non-empty chunk at or after c (wave-uniform): edges [b, e)
  auto seek = [&](int& c, int& b, int& e) -> bool {
    while (c < n_chunks) {
      b = out_ptr[c * chunk_nodes];
      e = out_ptr[min((c + 1) * chunk_nodes, n_nodes)];
      if (e > b) return true;
      c += cstride;
    }
    return false;
  };
  int chunk = blockIdx.x * 4 + wave, pb = 0, pend = 0;
  bool have = seek(chunk, pb, pend);
  f32x4v h[MT], carry[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) carry[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
  int s_cur = 0;
  int nchunk = chunk, npb = 0, npend = 0;
  bool nhave = false;
  int s_nxt = 0, d_nxt = 0;
  if (have) {
    const int pc = min(pb + r, pend - 1);
    s_cur = src_p[pc];
    const float* ps = Ps + (int64_t)s_cur * V + 4 * q;
    const float* pt = Pt + (int64_t)dst_p[pc] * V + 4 * q;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      f32x4v sv = *(const f32x4v*)(ps + 16 * a) + *(const f32x4v*)(pt + 16 * a);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[a][j] = (LCAO_ABL & 2048) ? sv[j] : lcao_silu(sv[j]);
    }
    npb = pb + 16; npend = pend; nhave = true;
    if (npb >= pend) { nchunk = chunk + cstride; nhave = seek(nchunk, npb, npend); }
    if (nhave) {
      const int pn = min(npb + r, npend - 1);
      s_nxt = src_p[pn];
      d_nxt = (LCAO_ABL & 4096) ? s_nxt : dst_p[pn];
    }
  }
  while (have) {
    asm volatile("" ::: "memory");
    const bool valid = pb + r < pend;
    const int64_t pw = min(pb + r, pend - 1);
    // node rows of the next tile: requested now, consumed after this tile's GEMM
    f32x4v ns[MT], nt[MT];
    int n2chunk = nchunk, n2pb = 0, n2pend = 0;
    bool n2have = false;
    int s_n2 = 0, d_n2 = 0;
    if (nhave) {
      const float* ps = Ps + (int64_t)s_nxt * V + 4 * q;
      const float* pt = Pt + (int64_t)d_nxt * V + 4 * q;
#pragma unroll
      for (int a = 0; a < MT; ++a) ns[a] = *(const f32x4v*)(ps + 16 * a);
#pragma unroll
      for (int a = 0; a < MT; ++a) nt[a] = *(const f32x4v*)(pt + 16 * a);
      n2pb = npb + 16; n2pend = npend; n2have = true;
      if (n2pb >= npend) { n2chunk = nchunk + cstride; n2have = seek(n2chunk, n2pb, n2pend); }
      if (n2have) {
        const int pn2 = min(n2pb + r, n2pend - 1);
        s_n2 = src_p[pn2];
        d_n2 = (LCAO_ABL & 4096) ? s_n2 : dst_p[pn2];
      }
    }
    f32x4v o[MT], wv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wv[m] = *(const f32x4v*)(bw + pw * V + 16 * m + 4 * q);  // in flight under the GEMM
    const int eb = out_ptr[s_cur], ee = out_ptr[s_cur + 1];   // first / one-past-last edge of this lane's centre node
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);
    __builtin_amdgcn_s_setprio(1);
    mfma_layer<MT, MT>(lds, V + 4, r, q, h, o);
    __builtin_amdgcn_s_setprio(0);
    if (pre2_out && valid) {   // the pre-activation of f_node.2, kept for the reverse sweep
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(pre2_out + pw * V + 16 * m + 4 * q) = o[m];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[m][j] = valid ? wv[m][j] * ((LCAO_ABL & 2048) ? o[m][j] : lcao_silu(o[m][j])) : 0.f;   // message rows; 0 past the chunk
    // segmented sum over the 16 edges of the tile, segment by segment (wave-uniform loop over the segment heads found by
    // a ballot; 1-3 per tile).  Only the first segment can continue a node from the previous tile and only the last can
    // continue into the next: their node bounds come from two vector loads issued before the GEMM (eb / ee), so the
    // loop itself touches no memory.
    const int s_prev = __shfl_up(s_cur, 1);                                        // (all lanes active: outside the ?:)
    const unsigned long long heads =
        __ballot(valid & ((r == 0) | (s_cur != s_prev))) & 0xFFFFull;              // bit r: edge r opens a segment (row 0)
    const int n_valid = min(16, pend - pb);
    const int first_beg = __builtin_amdgcn_readfirstlane(eb);
    const int last_end = __builtin_amdgcn_readlane(ee, n_valid - 1);
    unsigned rem = (unsigned)heads;
    if (LCAO_ABL & 1024) {   // ablation: no segmented reduction (one guarded store keeps the GEMM alive)
      rem = 0;
      float chk = (float)(first_beg + last_end);
#pragma unroll
      for (int m = 0; m < MT; ++m) chk += o[m][0] + o[m][1] + o[m][2] + o[m][3];
      if (chk == 12345.678f) agg[lane] = chk;
    }
    while (rem) {
      const int a = __builtin_ctz(rem);
      rem &= rem - 1;
      const int b = rem ? __builtin_ctz(rem) : n_valid;           // segment = lanes [a, b)
      const bool mine = r >= a && r < b;                          // (lanes past n_valid hold zeros already)
      const bool fresh = a > 0 || first_beg >= pb;                // the node's first edge lies in this tile: no carry
      const bool done = rem != 0 || last_end <= pb + 16;          // ... and its last one too: the row is complete
      const int node = __builtin_amdgcn_readlane(s_cur, a);
      float* out = agg + (int64_t)node * V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4v t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = lcao_row16_sum(mine ? o[m][j] : 0.f);
        if (!fresh) t += carry[m];
        if (done) {
          if (r == 0) *(f32x4v*)(out + 16 * m) = t;
        } else {
          carry[m] = t;
        }
      }
    }
    // rotate the pipeline
    if (nhave) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        f32x4v sv = ns[a] + nt[a];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[a][j] = (LCAO_ABL & 2048) ? sv[j] : lcao_silu(sv[j]);
      }
    }
    have = nhave; chunk = nchunk; pb = npb; pend = npend; s_cur = s_nxt;
    nhave = n2have; nchunk = n2chunk; npb = n2pb; npend = n2pend; s_nxt = s_n2; d_nxt = d_n2;
  }
}

// ===================================================================================================
// lcao_edge_node_message_agg, "one centre node per MFMA column" form.  A wave takes 16 consecutive centre nodes; in
// step t its 16 data rows are the t-th out-edge of each of them.  The E -> N sum of lcaonet.py:207-213 is then a
// per-lane accumulation over t — no cross-lane reduction, no carry, no segment bookkeeping, Ps[s] is loaded once per
// group, and nodes without edges simply store their zero accumulator (no memset).  The price is padding: a group runs
// max-degree steps, so the form is used only when the graph's degrees are even enough (lcao_node_column_hint: crystals,
// fixed-k neighbour lists; molecules of mixed size keep the edge-tiled kernel above).  The fp32 MFMA stream and VALU
// work do not overlap on this SIMD (profiles/r02/mfma_valu_overlap.txt): this form issues ~330 VALU instructions per
// 256 MFMAs where the edge-tiled one issues ~970.
// ===================================================================================================
__device__ __forceinline__ f32x4v lcao_silu4(f32x4v x) {
  f32x4v e;
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = __expf(-x[j]);
  e = e + 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
  return x * e;
}

template <int V>
__global__ __launch_bounds__(256) void k_edge_node_message_cols(
    const float* __restrict__ bw, const float* __restrict__ Ps, const float* __restrict__ Pt,
    const int32_t* __restrict__ dst_p, const int32_t* __restrict__ out_ptr, const float* __restrict__ Wn2,
    const float* __restrict__ bn2, int n_nodes, int n_edges, const int32_t* __restrict__ hint,
    float* __restrict__ pre2_out /* (E, V) or null */, float* __restrict__ agg) {
  constexpr int MT = V / 16;
  extern __shared__ float lds[];
  if (hint && *hint == 0) return;   // uneven degrees: the edge-tiled kernel has produced agg
  const bool by_degree = hint && *hint == 2;
  stage_weights(lds, Wn2, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n_groups = (n_nodes + 15) >> 4;
  for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
    const bool node_ok = 16 * g + r < n_nodes;
    const int slot = node_ok ? 16 * g + r : n_nodes - 1;
    const int s = by_degree ? hint[1 + slot] : slot;          // hint[0] == 2: nodes grouped by degree (lcao_node_column_hint)
    const int b = out_ptr[s];
    const int deg = node_ok ? out_ptr[s + 1] - b : 0;
    int T = deg;   // max over the 16 columns (the four quarters hold the same 16 values)
    T = max(T, __shfl_xor(T, 1)); T = max(T, __shfl_xor(T, 2)); T = max(T, __shfl_xor(T, 4)); T = max(T, __shfl_xor(T, 8));
    T = __builtin_amdgcn_readfirstlane(T);
    f32x4v acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    if (T > 0) {
      // edge of step t for this column, clamped into the node's own list (or to any valid edge for an edgeless column)
      auto edge_at = [&](int t) { return deg > 0 ? b + min(t, deg - 1) : min(b, n_edges - 1); };
      f32x4v ps[MT], h[MT];
      {
        const float* pr = Ps + (int64_t)s * V + 4 * q;
        const float* pt = Pt + (int64_t)dst_p[edge_at(0)] * V + 4 * q;
#pragma unroll
        for (int a = 0; a < MT; ++a) ps[a] = *(const f32x4v*)(pr + 16 * a);
#pragma unroll
        for (int a = 0; a < MT; ++a) h[a] = lcao_silu4(ps[a] + *(const f32x4v*)(pt + 16 * a));
      }
      int d_nxt = dst_p[edge_at(1)];
      for (int t = 0; t < T; ++t) {
        asm volatile("" ::: "memory");
        const int64_t pw = edge_at(t);
        const float live = t < deg ? 1.0f : 0.0f;
        // the target rows of step t + 1: requested now, consumed after this step's GEMM (re-read of the last row at the end)
        f32x4v nt[MT], wv[MT], o[MT];
        {
          const float* pt = Pt + (int64_t)d_nxt * V + 4 * q;
#pragma unroll
          for (int a = 0; a < MT; ++a) nt[a] = *(const f32x4v*)(pt + 16 * a);
        }
        d_nxt = dst_p[edge_at(t + 2)];
#pragma unroll
        for (int m = 0; m < MT; ++m) wv[m] = *(const f32x4v*)(bw + pw * V + 16 * m + 4 * q);   // in flight under the GEMM
#pragma unroll
        for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);   // bias as the initial accumulator
        __builtin_amdgcn_s_setprio(1);
        mfma_layer<MT, MT>(lds, V + 4, r, q, h, o);
        __builtin_amdgcn_s_setprio(0);
        if (pre2_out && t < deg) {
#pragma unroll
          for (int m = 0; m < MT; ++m) *(f32x4v*)(pre2_out + pw * V + 16 * m + 4 * q) = o[m];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] += (wv[m] * live) * lcao_silu4(o[m]);
#pragma unroll
        for (int a = 0; a < MT; ++a) h[a] = lcao_silu4(ps[a] + nt[a]);
      }
    }
    if (node_ok) {
      float* out = agg + (int64_t)s * V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(out + 16 * m) = acc[m];
    }
  }
}

// hint[0] = 1 when running every group of 16 consecutive centre nodes for its maximum degree costs at most 10 % more
// 16-edge steps than the edge list itself has.  Otherwise the nodes are grouped by degree instead: hint[1 .. n_nodes] gets
// a node order sorted by degree (counting sort; the order inside a degree class is arbitrary, which cannot change any result:
// a node's sum depends on its own column only) and hint[0] = 2 when THAT grouping stays within the 10 % — molecules of
// mixed size, surfaces.  0: neither (a few hubs among many low-degree nodes): the edge-tiled kernel runs.
// One workgroup; a few microseconds when the natural order passes, three passes over the nodes otherwise; built once per
// edge list.
#define LCAO_HINT_BINS 1024
__global__ __launch_bounds__(1024) void k_node_column_hint(const int32_t* __restrict__ out_ptr, int n_nodes, int n_edges,
                                                          int32_t* __restrict__ hint) {
  __shared__ long long part[1024];
  __shared__ int bins[LCAO_HINT_BINS];
  __shared__ int verdict;
  const int n_groups = (n_nodes + 15) >> 4;
  auto block_sum = [&](long long v) {
    part[threadIdx.x] = v;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    const long long total = part[0];
    __syncthreads();
    return total;
  };
  auto fits = [&](long long steps) { return steps * 160 <= (long long)n_edges * 11 + 160; };
  long long steps = 0;
  for (int g = threadIdx.x; g < n_groups; g += 1024) {
    int mx = 0;
    for (int i = 16 * g; i < min(16 * g + 16, n_nodes); ++i) mx = max(mx, out_ptr[i + 1] - out_ptr[i]);
    steps += mx;
  }
  steps = block_sum(steps);
  if (fits(steps)) {
    if (threadIdx.x == 0) hint[0] = 1;
    return;
  }
  // counting sort by degree, highest first
  for (int i = threadIdx.x; i < LCAO_HINT_BINS; i += 1024) bins[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_nodes; i += 1024) atomicAdd(&bins[min(out_ptr[i + 1] - out_ptr[i], LCAO_HINT_BINS - 1)], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int d = LCAO_HINT_BINS - 1; d >= 0; --d) {
      const int c = bins[d];
      bins[d] = run;
      run += c;
    }
  }
  __syncthreads();
  int32_t* order = hint + 1;
  for (int i = threadIdx.x; i < n_nodes; i += 1024)
    order[atomicAdd(&bins[min(out_ptr[i + 1] - out_ptr[i], LCAO_HINT_BINS - 1)], 1)] = i;
  __threadfence_block();
  __syncthreads();
  steps = 0;
  for (int g = threadIdx.x; g < n_groups; g += 1024) {
    int mx = 0;
    for (int i = 16 * g; i < min(16 * g + 16, n_nodes); ++i) {
      const int s = order[i];
      mx = max(mx, out_ptr[s + 1] - out_ptr[s]);
    }
    steps += mx;
  }
  steps = block_sum(steps);
  if (threadIdx.x == 0) hint[0] = fits(steps) ? 2 : 0;
  (void)verdict;
}

extern "C" int lcao_node_column_hint(const int32_t* out_ptr, int64_t n_nodes, int64_t n_edges, int32_t* hint, void* stream) {
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_SIZE(n_edges);
  LCAO_CHECK_PTR(hint);
  if (n_nodes > 0) LCAO_CHECK_PTR(out_ptr);
  hipLaunchKernelGGL(k_node_column_hint, dim3(1), dim3(1024), 0, (hipStream_t)stream, out_ptr, (int)n_nodes, (int)n_edges, hint);
  return lcao_launch_status();
}

static int launch_edge_node_message_agg(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                        const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                                        int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* pre2,
                                        float* agg, void* stream) {
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_nodes == 0) return LCAO_OK;
  LCAO_CHECK_PTR(agg);
  hipStream_t st = (hipStream_t)stream;
  const int me = lcao_zero_async(agg, sizeof(float) * (size_t)n_nodes * V, st);   // nodes without edges stay zero
  if (me != LCAO_OK) return me;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(bw); LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p);
  LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(Wn2); LCAO_CHECK_PTR(bn2);
  // ~10 tiles of 16 edges per chunk of whole nodes: the one partly empty tile per chunk costs ~5 % of the matrix work
  int64_t chunk_nodes = (160 * n_nodes + n_edges - 1) / n_edges;
  if (chunk_nodes < 1) chunk_nodes = 1;
  if (chunk_nodes > 64) chunk_nodes = 64;
  const int64_t n_chunks = (n_nodes + chunk_nodes - 1) / chunk_nodes;
  size_t lds = size_t(V) * (V + 4) * sizeof(float);
  int64_t per_cu = (lds > 80 * 1024) ? 1 : ((lds > 53 * 1024) ? 2 : 3);
  int64_t groups = (n_chunks + 3) / 4, cap = (int64_t)lcao_num_cu() * per_cu;
  if (groups > cap) groups = cap;
  int64_t cgroups = (((n_nodes + 15) >> 4) + 3) / 4;   // column form: one wave per 16 centre nodes
  if (cgroups > cap) cgroups = cap;
#define NMA_CASE(V_)                                                                                             \
  if (V == V_) {                                                                                                 \
    static bool done = false;                                                                                    \
    int rc = set_lds_attr(k_edge_node_message_agg<V_>, lds, done);                                               \
    if (rc != LCAO_OK) return rc;                                                                                \
    hipLaunchKernelGGL(k_edge_node_message_agg<V_>, dim3((unsigned)groups), dim3(256), lds, st, bw, Ps, Pt, src_p,   \
                       dst_p, out_ptr, Wn2, bn2, (int)n_nodes, (int)chunk_nodes, column_hint, pre2, agg);        \
    if (column_hint) {                                                                                           \
      static bool done2 = false;                                                                                 \
      rc = set_lds_attr(k_edge_node_message_cols<V_>, lds, done2);                                               \
      if (rc != LCAO_OK) return rc;                                                                              \
      hipLaunchKernelGGL(k_edge_node_message_cols<V_>, dim3((unsigned)cgroups), dim3(256), lds, st, bw, Ps, Pt,  \
                         dst_p, out_ptr, Wn2, bn2, (int)n_nodes, (int)n_edges, column_hint, pre2, agg);          \
    }                                                                                                            \
    return lcao_launch_status();                                                                                 \
  }
  NMA_CASE(128) NMA_CASE(64) NMA_CASE(32)
#undef NMA_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_edge_node_message_agg(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                          const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                                          int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* agg,
                                          void* stream) {
  return launch_edge_node_message_agg(bw, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, n_nodes, n_edges, V, column_hint, nullptr,
                                      agg, stream);
}

extern "C" int lcao_edge_node_message_agg_saving(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                                 const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2,
                                                 const float* bn2, int64_t n_nodes, int64_t n_edges, int V,
                                                 const int32_t* column_hint, float* pre2, float* agg, void* stream) {
  if (n_edges > 0) LCAO_CHECK_PTR(pre2);
  return launch_edge_node_message_agg(bw, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, n_nodes, n_edges, V, column_hint, pre2,
                                      agg, stream);
}

static int launch_edge_lcao_weight(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges,
                                   int V, float* bw, float* ft_out, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(tbw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(W3); LCAO_CHECK_PTR(Wb); LCAO_CHECK_PTR(bw);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(2) * V * (V + 4) * sizeof(float);
#define LW_CASE(V_)                                                                                          \
  if (V == V_) {                                                                                             \
    /* one workgroup per CU when the two weight matrices fill the LDS: 8 waves keep 2 per SIMD */           \
    constexpr int TH = (2 * V_ * (V_ + 4) * 4 > 80 * 1024) ? 512 : 256;                                      \
    static bool done = false;                                                                                \
    int rc = set_lds_attr(k_edge_lcao_weight<V_, TH>, lds, done);                                            \
    if (rc != LCAO_OK) return rc;                                                                            \
    hipLaunchKernelGGL((k_edge_lcao_weight<V_, TH>), dim3(persistent_groups(n_edges, lds, TH / 64)), dim3(TH), lds, \
                       st, tbw, B, W3, Wb, n_edges, bw, ft_out);                                             \
    return lcao_launch_status();                                                                             \
  }
  LW_CASE(128) LW_CASE(64) LW_CASE(32)
#undef LW_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_edge_lcao_weight(const float* tbw, const float* B, const float* W3, const float* Wb,
                                     int64_t n_edges, int V, float* bw, void* stream) {
  return launch_edge_lcao_weight(tbw, B, W3, Wb, n_edges, V, bw, nullptr, stream);
}

extern "C" int lcao_edge_lcao_weight_saving(const float* tbw, const float* B, const float* W3, const float* Wb,
                                            int64_t n_edges, int V, float* bw, float* ft, void* stream) {
  if (n_edges > 0) LCAO_CHECK_PTR(ft);
  return launch_edge_lcao_weight(tbw, B, W3, Wb, n_edges, V, bw, ft, stream);
}

static int launch_edge_node_message(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                    const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V,
                                    float* msg, float* pre2_out, void* stream) {
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
                       Pt, src_p, dst_p, Wn2, bn2, n_edges, msg, pre2_out);                                    \
    return lcao_launch_status();                                                                               \
  }
  NM_CASE(128) NM_CASE(64) NM_CASE(32)
#undef NM_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_edge_node_message(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                      const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges,
                                      int V, float* msg, void* stream) {
  return launch_edge_node_message(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg, nullptr, stream);
}

extern "C" int lcao_edge_node_message_saving(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                             const int32_t* dst_p, const float* Wn2, const float* bn2,
                                             int64_t n_edges, int V, float* msg, float* pre2, void* stream) {
  if (n_edges > 0) LCAO_CHECK_PTR(pre2);
  return launch_edge_node_message(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg, pre2, stream);
}

// ===================================================================================================
// Reverse sweep of the two-body chain (autograd forces / training), fused like the forward kernels.
//
// lcao_message_bwd_fused — backward of  msg = bw * SiLU(pre2),  pre2 = Wn2 . SiLU(pre1) + b,  pre1 = Ps[s] + Pt[t]:
//   g_msg = g_agg[src_p];  g_bw = g_msg SiLU(pre2);  g_pre2 = g_msg bw SiLU'(pre2)
//   g_pre1 = (Wn2^T . g_pre2) SiLU'(pre1)                                   weights: Wn2T = Wn2^T (V, V) row-major
// lcao_lcao_weight_bwd_fused — backward of  bw = Wb . lw,  lw = u / max(|u|, eps),  u = B (1 + ft),  ft = W3 . tbw:
//   g_lw = Wb^T . g_bw;  g_u = (g_lw - lw (lw . g_lw)) / |u|;  g_B = g_u (1 + ft);  g_ft = g_u B;  g_tbw = W3^T . g_ft
// Same tile formulation as the forward kernels (gradient rows are the MFMA B operand, transposed weights in LDS).
// ===================================================================================================
template <int V>
__global__ __launch_bounds__(256) void k_message_bwd_mfma(
    const float* __restrict__ g_agg, const int32_t* __restrict__ src_p, const int32_t* __restrict__ dst_p,
    const float* __restrict__ bw, const float* __restrict__ pre2, const float* __restrict__ Ps,
    const float* __restrict__ Pt, const float* __restrict__ Wn2T, int64_t E, float* __restrict__ g_bw,
    float* __restrict__ g_pre1, float* __restrict__ g_pre2_out /* or null */) {
  constexpr int MT = V / 16;
  extern __shared__ float lds[];
  stage_weights(lds, Wn2T, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    const int s = src_p[pc], t = dst_p[pc];
    const float* ga = g_agg + (int64_t)s * V + 4 * q;
    f32x4v gp2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const f32x4v g = *(const f32x4v*)(ga + 16 * m);
      const f32x4v b = *(const f32x4v*)(bw + pc * V + 16 * m + 4 * q);
      const f32x4v x = *(const f32x4v*)(pre2 + pc * V + 16 * m + 4 * q);
      f32x4v gb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gb[j] = g[j] * lcao_silu(x[j]);
        gp2[m][j] = g[j] * b[j] * lcao_silu_grad(x[j]);
      }
      if (valid) {
        *(f32x4v*)(g_bw + p * V + 16 * m + 4 * q) = gb;
        if (g_pre2_out) *(f32x4v*)(g_pre2_out + p * V + 16 * m + 4 * q) = gp2[m];
      }
    }
    if (!g_pre1) continue;          // the first layer of a force sweep: its node tables do not depend on positions
    // the node rows of the epilogue (pre1 = Ps[s] + Pt[t]) are requested before the GEMM and land under it
    f32x4v x1[MT];
    {
      const float* ps = Ps + (int64_t)s * V + 4 * q;
      const float* pt = Pt + (int64_t)t * V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) x1[m] = *(const f32x4v*)(ps + 16 * m) + *(const f32x4v*)(pt + 16 * m);
    }
    f32x4v gh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) gh[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<MT, MT>(lds, V + 4, r, q, gp2, gh);
    if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gh[m][j] * lcao_silu_grad(x1[m][j]);
        *(f32x4v*)(g_pre1 + p * V + 16 * m + 4 * q) = o;
      }
    }
  }
}

template <int V, int THREADS>
__global__ __launch_bounds__(THREADS) void k_lcao_weight_bwd_mfma(
    const float* __restrict__ g_bw, const float* __restrict__ B, const float* __restrict__ ft,
    const float* __restrict__ WbT, const float* __restrict__ W3T, int64_t E, float* __restrict__ g_B,
    float* __restrict__ g_tbw, float* __restrict__ g_ft_out /* or null */) {
  constexpr int MT = V / 16, WAVES = THREADS / 64;
  extern __shared__ float lds[];
  float* wbs = lds;
  float* w3s = lds + V * (V + 4);
  stage_weights(wbs, WbT, V, V, V);
  stage_weights(w3s, W3T, V, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
    asm volatile("" ::: "memory");
    const int64_t p = tile * 16 + r;
    const bool valid = p < E;
    const int64_t pc = valid ? p : E - 1;
    f32x4v x[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) x[a] = *(const f32x4v*)(g_bw + pc * V + 16 * a + 4 * q);
    f32x4v gl[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) gl[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<MT, MT>(wbs, V + 4, r, q, x, gl);
    f32x4v b[MT], f[MT];
    float n2 = 0.f, dot = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      b[m] = *(const f32x4v*)(B + pc * V + 16 * m + 4 * q);
      f[m] = *(const f32x4v*)(ft + pc * V + 16 * m + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = b[m][j] + b[m][j] * f[m][j];
        x[m][j] = u;  // x is free after the first GEMM: keep u there
        n2 += u * u;
        dot += u * gl[m][j];
      }
    }
    n2 += __shfl_xor(n2, 16);
    n2 += __shfl_xor(n2, 32);
    dot += __shfl_xor(dot, 16);
    dot += __shfl_xor(dot, 32);
    const float nrm = sqrtf(n2);
    const bool live = nrm > 1e-12f;
    const float inv = live ? 1.0f / nrm : 1e12f;
    const float proj = live ? dot * inv * inv : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4v gB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gu = inv * (gl[m][j] - x[m][j] * proj);
        gB[j] = gu * (1.0f + f[m][j]);
        gl[m][j] = gu * b[m][j];  // g_ft, the operand of the second GEMM
      }
      if (valid) {
        *(f32x4v*)(g_B + p * V + 16 * m + 4 * q) = gB;
        if (g_ft_out) *(f32x4v*)(g_ft_out + p * V + 16 * m + 4 * q) = gl[m];
      }
    }
    f32x4v o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) o[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
    mfma_layer<MT, MT>(w3s, V + 4, r, q, gl, o);
    if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) *(f32x4v*)(g_tbw + p * V + 16 * m + 4 * q) = o[m];
    }
  }
}

extern "C" int lcao_message_bwd_fused(const float* g_agg, const int32_t* src_p, const int32_t* dst_p, const float* bw,
                                      const float* pre2, const float* Ps, const float* Pt, const float* Wn2T,
                                      int64_t n_edges, int V, float* g_bw, float* g_pre1, float* g_pre2,
                                      void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_agg); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(bw); LCAO_CHECK_PTR(pre2);
  LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt); LCAO_CHECK_PTR(Wn2T); LCAO_CHECK_PTR(g_bw);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(V) * (V + 4) * sizeof(float);
#define MB_CASE(V_)                                                                                              \
  if (V == V_) {                                                                                                 \
    static bool done = false;                                                                                    \
    int rc = set_lds_attr(k_message_bwd_mfma<V_>, lds, done);                                                    \
    if (rc != LCAO_OK) return rc;                                                                                \
    hipLaunchKernelGGL(k_message_bwd_mfma<V_>, dim3(persistent_groups(n_edges, lds)), dim3(256), lds, st, g_agg,  \
                       src_p, dst_p, bw, pre2, Ps, Pt, Wn2T, n_edges, g_bw, g_pre1, g_pre2);                     \
    return lcao_launch_status();                                                                                 \
  }
  MB_CASE(128) MB_CASE(64) MB_CASE(32)
#undef MB_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_lcao_weight_bwd_fused(const float* g_bw, const float* B, const float* ft, const float* WbT,
                                          const float* W3T, int64_t n_edges, int V, float* g_B, float* g_tbw,
                                          float* g_ft, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_bw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(ft); LCAO_CHECK_PTR(WbT); LCAO_CHECK_PTR(W3T);
  LCAO_CHECK_PTR(g_B); LCAO_CHECK_PTR(g_tbw);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(2) * V * (V + 4) * sizeof(float);
#define WB_CASE(V_)                                                                                              \
  if (V == V_) {                                                                                                 \
    constexpr int TH = (2 * V_ * (V_ + 4) * 4 > 80 * 1024) ? 512 : 256;                                          \
    static bool done = false;                                                                                    \
    int rc = set_lds_attr(k_lcao_weight_bwd_mfma<V_, TH>, lds, done);                                            \
    if (rc != LCAO_OK) return rc;                                                                                \
    hipLaunchKernelGGL((k_lcao_weight_bwd_mfma<V_, TH>), dim3(persistent_groups(n_edges, lds, TH / 64)), dim3(TH), \
                       lds, st, g_bw, B, ft, WbT, W3T, n_edges, g_B, g_tbw, g_ft);                               \
    return lcao_launch_status();                                                                                 \
  }
  WB_CASE(128) WB_CASE(64) WB_CASE(32)
#undef WB_CASE
  return LCAO_ERR_UNSUPPORTED;
}

// ===================================================================================================
// lcao_triplet_gemm — the three-body term (lcaonet.py:173-189) as one small GEMM per centre node s:
//
//     tbw[Out(s), :] = W_s . Ag[In(s), :, :]        W_s[p, (q,l)] = Y_l(u_p . u_q) / max(|v_pq|, 1e-12)   (q != p)
//
// Ag[q,l,:] = A[q,l,:] * sigmoid(xk)[src(q),:] is written by lcao_coeff_mlp_contract; the per-triplet norm
// |v_pq|^2 = |sum_l Y_l A[q,l,:]|^2 = sum_{l,l'} Y_l Y_l' G_q[l,l'] comes from the per-edge Gram matrix G_q of
// the ungated A (10 floats), so no per-triplet V-long reduction and no T-sized tensor exist.
// One wave per node; the 16-B row pieces of Ag are the MFMA "A" operand straight from global memory (coalesced
// 256-B segments, every row of Ag is read exactly once in the whole launch), W is built per lane from wave-uniform
// scalars (u_q, G_q) and the lane's own out-edge direction and is the "B" operand.  No LDS.
//   k-step = one in-edge q (quarter = l; l >= n_l contributes zero), 32 out-edges per pass (two 16-wide tiles).
//   accumulator tile (h, jj), lane (n, quarter), reg r  <->  out-edge n, channel 64h + 16*quarter + 4r + jj.
//   HBM per edge: read 4*n_l*V + 40 + 12; write 4*V.
// ===================================================================================================
#define gram_idx lcao_tri_idx   // entry (l, l2 >= l) of the per-edge triangular factor (lcao_common.h)

// Per-triplet weights of the two 16-wide out-edge tiles of a step at once (packed fp32: .x = tile A, .y = tile B).
// Monic angular polynomials y0' = 1, y1' = c, y2' = c^2 - Y2B/Y2A, y3' = c (c^2 - 3/5) with Y_l = s_l y_l'; the entries of
// the edge's triangular factor arrive PRE-SCALED, g[i,j] = s_j R[i,j] (one per-lane multiply on the loaded word), so
//     |v|^2 = |R Y|^2 = sum_i t_i^2,   t_i = sum_{j >= i} g[i,j] y_j'
// is a chain of packed FMAs — a sum of squares (lcao_common.h: why not the Gram form) — and the lane's own Y_quarter
// comes from per-lane constants k_l = s_l [l == quarter] instead of a select.  ~21 VALU instructions for both weights.
typedef float f32x2v __attribute__((ext_vector_type(2)));
#define LCAO_Y2_SHIFT (0.31539156525252f / 0.94617469575756f)
__host__ __device__ constexpr float trip_scale(int l) {   // s_l
  return l == 0 ? 0.28209479177387814f : l == 1 ? 0.48860251190292f : l == 2 ? 0.94617469575756f : 5.0f * 0.373176332590115f;
}

template <int NL>
__device__ __forceinline__ f32x2v triplet_weight2(f32x2v ux, f32x2v uy, f32x2v uz, float qx, float qy, float qz,
                                                  const float (&g)[10], const float (&k)[4], bool okA, bool okB) {
  const f32x2v c = ux * qx + uy * qy + uz * qz;
  f32x2v t0 = {g[gram_idx(0, 0)], g[gram_idx(0, 0)]};
  f32x2v yq = {k[0], k[0]};
  f32x2v c2, y2, n2;
  if (NL == 1) n2 = t0 * t0;
  if (NL > 1) {
    c2 = c * c;
    t0 += g[gram_idx(0, 1)] * c;
    yq += k[1] * c;
  }
  if (NL == 2) {
    const f32x2v t1 = g[gram_idx(1, 1)] * c;
    n2 = t0 * t0 + t1 * t1;
  }
  if (NL > 2) {
    y2 = c2 - LCAO_Y2_SHIFT;
    t0 += g[gram_idx(0, 2)] * y2;
    yq += k[2] * y2;
  }
  if (NL == 3) {
    const f32x2v t1 = g[gram_idx(1, 1)] * c + g[gram_idx(1, 2)] * y2;
    const f32x2v t2 = g[gram_idx(2, 2)] * y2;
    n2 = t0 * t0 + t1 * t1 + t2 * t2;
  }
  if (NL > 3) {
    const f32x2v y3 = c * (c2 - 0.6f);
    t0 += g[gram_idx(0, 3)] * y3;
    yq += k[3] * y3;
    const f32x2v t1 = g[gram_idx(1, 1)] * c + g[gram_idx(1, 2)] * y2 + g[gram_idx(1, 3)] * y3;
    const f32x2v t2 = g[gram_idx(2, 2)] * y2 + g[gram_idx(2, 3)] * y3;
    const f32x2v t3 = g[gram_idx(3, 3)] * y3;
    n2 = t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
  }
  // 1 / max(sqrt(n2), 1e-12) = rsqrt(max(n2, 1e-24)): one v_rsq_f32 (1 ulp) instead of the IEEE sqrt + divide sequences
  f32x2v w = {__builtin_amdgcn_rsqf(fmaxf(n2[0], 1e-24f)), __builtin_amdgcn_rsqf(fmaxf(n2[1], 1e-24f))};
  w *= yq;
  w[0] = okA ? w[0] : 0.f;
  w[1] = okB ? w[1] : 0.f;
  return w;
}

template <int V, int NL>
__global__ __launch_bounds__(256) void k_triplet_gemm(const float* __restrict__ Ag, const float* __restrict__ gram,
                                                      const float* __restrict__ uvec_p,
                                                      const int32_t* __restrict__ in_ptr,
                                                      const int32_t* __restrict__ in_pos,
                                                      const int32_t* __restrict__ out_ptr, int n_nodes,
                                                      float* __restrict__ tbw) {
  constexpr int HG = (V + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, quarter = lane >> 4;

  // One step = one in-edge q.  Everything a step needs arrives through the VECTOR memory pipe, PD steps ahead:
  //   F[h]: this lane's 16-B pieces of Ag[q, quarter, :]          (2 x global_load_dwordx4)
  //   R   : lane i < 3 holds u_q[i], lane 3 + i holds entry i of the edge's triangular factor (1 x global_load_dword, then v_readlane)
  // VMEM returns in order, so a COUNTED s_waitcnt vmcnt((PD-1) x 3) leaves the later steps in flight while step j is
  // consumed — but the compiler can only count if every step issues exactly the same loads.  So nothing here is
  // conditional: lanes of quarter >= n_l read the row of the last real l (their weight is forced to zero instead), the
  // u / gram word is one load through a per-lane pointer, and a refill past the node's last in-edge re-reads that last
  // row (an L1 hit).  With any of these under a branch hipcc emitted vmcnt(0) at every step, i.e. a one-step prefetch.
  // Scalar (SMEM) loads would force lgkmcnt(0) on every use and sat on the critical path of each step.
  constexpr int QL = NL - 1;
  const int qrow = quarter < NL ? quarter : QL;
  // per-lane constants: the scale of this lane's loaded word (lanes 3..12 hold factor entry lane - 3 = (l, l') in upper-
  // triangle order, scaled by s_l') and the one-hot Y_quarter coefficients
  float rscale = 1.0f;
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int l2 = l; l2 < 4; ++l2)
      if (lane - 3 == gram_idx(l, l2)) rscale = trip_scale(l2);
  float kq[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) kq[l] = (quarter == l && l < NL) ? trip_scale(l) : 0.f;
  auto load_step = [&](f32x4v(&F)[HG], float& R, int q) {
#pragma unroll
    for (int h = 0; h < HG; ++h) {
      const int ch = (64 * h + 4 * n) < V ? 64 * h + 4 * n : 0;
      F[h] = *(const f32x4v*)(Ag + ((int64_t)q * NL + qrow) * V + ch);
    }
    const float* rp = lane < 3 ? uvec_p + 3 * (int64_t)q + lane : gram + 10 * (int64_t)q + min(lane - 3, 9);
    R = *rp * rscale;
  };
  auto step = [&](f32x4v(&F)[HG], float R, int q0, f32x2v ux, f32x2v uy, f32x2v uz, int pA, int pB, bool vA, bool vB, bool two,
                  f32x4v(&acc)[2][HG][4]) {
    const int r0i = __float_as_int(R);  // readlane moves raw 32-bit lanes
    const float qx = __int_as_float(__builtin_amdgcn_readlane(r0i, 0)),
                qy = __int_as_float(__builtin_amdgcn_readlane(r0i, 1)),
                qz = __int_as_float(__builtin_amdgcn_readlane(r0i, 2));
    float G[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) G[k] = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int l2 = l; l2 < NL; ++l2)
        G[gram_idx(l, l2)] = __int_as_float(__builtin_amdgcn_readlane(r0i, 3 + gram_idx(l, l2)));
    // lcaonet.py:470 drops only the triplet whose two edges coincide (q == p); the back-triplet stays
    const f32x2v w = triplet_weight2<NL>(ux, uy, uz, qx, qy, qz, G, kq, vA && q0 != pA, vB && q0 != pB);
#if LCAO_ABL & 64
    acc[0][0][0] = MFMA16(F[0][0] + F[HG - 1][3] + F[0][1] + F[HG - 1][2], w[0], acc[0][0][0]);  // diagnostic: 1 MFMA per step
    if (two) acc[1][0][0] = MFMA16(F[0][0], w[1], acc[1][0][0]);
#else
#pragma unroll
    for (int h = 0; h < HG; ++h)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[0][h][jj] = MFMA16(F[h][jj], w[0], acc[0][h][jj]);
    if (two) {
#pragma unroll
      for (int h = 0; h < HG; ++h)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[1][h][jj] = MFMA16(F[h][jj], w[1], acc[1][h][jj]);
    }
#endif
  };

  // PD steps stay in flight per wave (PD x 1.5 KB at V=128, n_l=3): with 3 waves per SIMD that is ~100 KB of loads in
  // flight per CU, what a random row gather needs to approach the HBM rate.  Slots are indexed statically (the step loop
  // is unrolled by PD): full groups of PD steps first — every step consumes its slot and refills it —, then the < PD
  // remaining steps without refills.
  constexpr int PD = 5;
  for (int s = blockIdx.x * 4 + wave; s < n_nodes; s += gridDim.x * 4) {
    const int p0 = out_ptr[s], p1 = out_ptr[s + 1];
    const int j0 = in_ptr[s];
    const int nin = in_ptr[s + 1] - j0;
    for (int pb = p0; pb < p1; pb += 32) {
      const int pA = pb + n, pB = pb + 16 + n;
      const bool vA = pA < p1, vB = pB < p1;
      const bool two = (p1 - pb) > 16;  // wave-uniform
      const float* ua = uvec_p + 3 * (int64_t)(vA ? pA : p0);
      const float* ub = uvec_p + 3 * (int64_t)(vB ? pB : p0);
      const f32x2v ux = {ua[0], ub[0]}, uy = {ua[1], ub[1]}, uz = {ua[2], ub[2]};
      f32x4v acc[2][HG][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < HG; ++h)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[t][h][jj] = f32x4v{0.f, 0.f, 0.f, 0.f};

      for (int jc = 0; jc < nin; jc += 64) {   // in-edge ids of this node, 64 at a time: lane i holds in_pos[j0+jc+i]
        const int cn = min(64, nin - jc);
        const int idv = in_pos[j0 + jc + min(lane, cn - 1)];
        f32x4v F[PD][HG];
        float R[PD];
        int Q[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) {
          Q[i] = __builtin_amdgcn_readlane(idv, min(i, cn - 1));
          load_step(F[i], R[i], Q[i]);
        }
        int jb = 0;
        for (; jb + PD <= cn; jb += PD) {
#pragma unroll
          for (int i = 0; i < PD; ++i) {
            step(F[i], R[i], Q[i], ux, uy, uz, pA, pB, vA, vB, two, acc);
            Q[i] = __builtin_amdgcn_readlane(idv, min(jb + i + PD, cn - 1));   // refill this slot with step j + PD
            load_step(F[i], R[i], Q[i]);
          }
        }
        const int rem = cn - jb;
#pragma unroll
        for (int i = 0; i < PD - 1; ++i)
          if (i < rem) step(F[i], R[i], Q[i], ux, uy, uz, pA, pB, vA, vB, two, acc);   // wave-uniform
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int p = t == 0 ? pA : pB;
        if (t == 0 ? vA : vB) {
#pragma unroll
          for (int h = 0; h < HG; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ch = 64 * h + 16 * quarter + 4 * r;
              if (ch < V) {
                f32x4v v = {acc[t][h][0][r], acc[t][h][1][r], acc[t][h][2][r], acc[t][h][3][r]};
                *(f32x4v*)(tbw + (int64_t)p * V + ch) = v;
              }
            }
        }
      }
    }
  }
}

// any-V fallback of k_triplet_gemm (same Gram-form arithmetic on the VALU): one wave per out-edge, lanes over V
template <int R>
__global__ void k_triplet_gram_any(const float* __restrict__ Ag, const float* __restrict__ gram,
                                   const float* __restrict__ uvec_p, const int32_t* __restrict__ in_ptr,
                                   const int32_t* __restrict__ in_pos, const int32_t* __restrict__ out_ptr, int n_nodes,
                                   int V, int NL, float* __restrict__ tbw) {
  int lane = threadIdx.x & 63;
  int64_t wv = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wv; s < n_nodes; s += nwave) {
    const int p0 = out_ptr[s], p1 = out_ptr[s + 1];
    const int j0 = in_ptr[s], j1 = in_ptr[s + 1];
    for (int p = p0; p < p1; ++p) {
      const float ux = uvec_p[3 * (int64_t)p], uy = uvec_p[3 * (int64_t)p + 1], uz = uvec_p[3 * (int64_t)p + 2];
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.f;
      for (int j = j0; j < j1; ++j) {
        const int q = in_pos[j];
        if (q == p) continue;
        const float c = ux * uvec_p[3 * (int64_t)q] + uy * uvec_p[3 * (int64_t)q + 1] + uz * uvec_p[3 * (int64_t)q + 2];
        float y[4], Rq[10], Gy[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) y[l] = l < NL ? lcao_y_l0(c, l) : 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i) Rq[i] = gram[(int64_t)q * 10 + i];
        const float m = lcao_factor_quadratic<4>(Rq, y, Gy);   // rows / columns >= n_l of the factor are zero
        const float inv = 1.0f / fmaxf(sqrtf(fmaxf(m, 0.f)), 1e-12f);
        for (int l = 0; l < NL; ++l) {
          const float w = y[l] * inv;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int ch = lane + 64 * r;
            if (ch < V) acc[r] += w * Ag[((int64_t)q * NL + l) * V + ch];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + 64 * r;
        if (ch < V) tbw[(int64_t)p * V + ch] = acc[r];
      }
    }
  }
}

extern "C" int lcao_triplet_gemm(const float* Ag, const float* gram, const float* uvec_p, const int32_t* in_ptr,
                                 const int32_t* in_pos, const int32_t* out_ptr, int64_t n_nodes, int64_t n_edges, int V,
                                 int n_l, float* tbw, void* stream) {
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0 || n_nodes == 0) return LCAO_OK;
  LCAO_CHECK_PTR(Ag); LCAO_CHECK_PTR(gram); LCAO_CHECK_PTR(uvec_p); LCAO_CHECK_PTR(in_ptr); LCAO_CHECK_PTR(in_pos);
  LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(tbw);
  hipStream_t st = (hipStream_t)stream;
  int64_t groups = (n_nodes + 3) / 4;
  int64_t cap = (int64_t)lcao_num_cu() * 8;
  if (groups > cap) groups = cap;
  // occupancy experiment (tools/k3_ablate.py): LCAO_K3_LDS = bytes of unused dynamic LDS per workgroup caps the waves/SIMD
  static int k3_lds = -1;
  if (k3_lds < 0) { const char* e_ = getenv("LCAO_K3_LDS"); k3_lds = e_ ? atoi(e_) : 0; }
#define TG_CASE(V_, NL_)                                                                                          \
  if (V == V_ && n_l == NL_) {                                                                                    \
    if (k3_lds > 48 * 1024) {                                                                                     \
      static bool done_ = false;                                                                                  \
      int rc_ = set_lds_attr(k_triplet_gemm<V_, NL_>, (size_t)k3_lds, done_);                                     \
      if (rc_ != LCAO_OK) return rc_;                                                                             \
    }                                                                                                             \
    hipLaunchKernelGGL((k_triplet_gemm<V_, NL_>), dim3((unsigned)groups), dim3(256), (size_t)k3_lds, st, Ag, gram, uvec_p, \
                       in_ptr, in_pos, out_ptr, (int)n_nodes, tbw);                                               \
    return lcao_launch_status();                                                                                  \
  }
#define TG_V(V_) TG_CASE(V_, 1) TG_CASE(V_, 2) TG_CASE(V_, 3) TG_CASE(V_, 4)
  TG_V(128) TG_V(64) TG_V(32)
#undef TG_V
#undef TG_CASE
  if (V > 512) return LCAO_ERR_UNSUPPORTED;
  {
    int R = (V + 63) / 64;
    dim3 g2((unsigned)groups), b2(256);
#define TGA(RR)                                                                                                    \
  hipLaunchKernelGGL(k_triplet_gram_any<RR>, g2, b2, 0, st, Ag, gram, uvec_p, in_ptr, in_pos, out_ptr, (int)n_nodes, V, \
                     n_l, tbw)
    if (R == 1) TGA(1);
    else if (R == 2) TGA(2);
    else if (R <= 4) TGA(4);
    else TGA(8);
#undef TGA
  }
  return lcao_launch_status();
}

// ===================================================================================================
// lcao_triplet_bwd (MFMA form) — reverse of lcao_triplet_gemm for one centre node per wave.
//   forward:  tbw[p,:] = sum_{q in In(s), q != p} sum_l w_pql Ag[q,l,:],   w_pql = Y_l(c_pq) / max(sqrt(Y^T G_q Y), eps)
//   reverse:  GW[p,q,l] = g_tbw[p,:] . Ag[q,l,:]                    (GEMM 1: out-edges x (in-edges, l), k = channels)
//             scalar chain per (p,q): g_c = dE/dc_pq, g_gram[q] += ..., w_pql                     (registers)
//             g_Ag[q,l,:] = sum_p w_pql g_tbw[p,:]                   (GEMM 2: channels x in-edges, k = out-edges)
//             g_uq[q] = sum_p g_c u_p   (in-edge side),   g_up[p] = sum_q g_c u_q   (out-edge side)
// Tiles of 16 in-edges (columns, on the lane) x 16 out-edges (rows, in the 4 registers x 4 quarters).  GEMM 1 leaves
// (p = 4*quarter + r, q = lane & 15) in lane (q, quarter) register r for every l — which is the B-operand layout of
// GEMM 2 when its k-step r takes p = 4*quarter + r from quarter `quarter`.  The wave owns every output row it writes
// (each edge is an in-edge of exactly one node, an out-edge of exactly one node): no atomics.
// ===================================================================================================
template <int V, int NL>
__global__ __launch_bounds__(256) void k_triplet_bwd_mfma(
    const float* __restrict__ g_tbw, const float* __restrict__ Ag, const float* __restrict__ gram,
    const float* __restrict__ uvec_p, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_pos,
    const int32_t* __restrict__ out_ptr, int n_nodes, float* __restrict__ g_Ag, float* __restrict__ g_gram,
    float* __restrict__ g_uq, float* __restrict__ g_up) {
  constexpr int KT = V / 16;   // k chunks of GEMM 1 = channel tiles of GEMM 2
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, quarter = lane >> 4;
  for (int s = blockIdx.x * 4 + wave; s < n_nodes; s += gridDim.x * 4) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    // g_up of this node's out-edges accumulates over in-tiles in global memory (this wave is its only writer)
    // (zeroed by the very lanes that later add into it: same-lane program order is the only ordering relied upon)
    for (int it0 = 0; it0 < m_out; it0 += 16)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pi = it0 + 4 * quarter + r;
        if (n == 0 && pi < m_out) {
          float* o = g_up + 3 * (int64_t)(p0 + pi);
          o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
        }
      }
    for (int jt = 0; jt < m_in; jt += 16) {
      const bool qv = (jt + n) < m_in;
      const int q = in_pos[j0 + (qv ? jt + n : 0)];
      const float qx = uvec_p[3 * (int64_t)q], qy = uvec_p[3 * (int64_t)q + 1], qz = uvec_p[3 * (int64_t)q + 2];
      float G[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) G[i] = gram[(int64_t)q * 10 + i];
      f32x4v gA[NL][KT];
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int t = 0; t < KT; ++t) gA[l][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
      float gG[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) gG[i] = 0.f;
      float gqx = 0.f, gqy = 0.f, gqz = 0.f;
      const float* agq = Ag + (int64_t)q * NL * V + 4 * quarter;

      for (int it = 0; it < m_out; it += 16) {
        asm volatile("" ::: "memory");
        // ---- GEMM 1: gw[l][r] = g_tbw[p_(4*quarter+r)] . Ag[q_n, l]   (A operand: row = out-edge n, B: col = in-edge n)
        const int pa = p0 + min(it + n, m_out - 1);
        const float* gta = g_tbw + (int64_t)pa * V + 4 * quarter;
        f32x4v gw[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) gw[l] = f32x4v{0.f, 0.f, 0.f, 0.f};
#if LCAO_ABL & 128
        {  // diagnostic: one k chunk instead of KT
          const f32x4v xa = *(const f32x4v*)(gta);
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            const f32x4v xb = *(const f32x4v*)(agq + l * V);
            gw[l] = MFMA16(xa[0], xb[0], gw[l]);
          }
        }
#else
#pragma unroll
        for (int a = 0; a < KT; ++a) {
          const f32x4v xa = *(const f32x4v*)(gta + 16 * a);
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            const f32x4v xb = *(const f32x4v*)(agq + l * V + 16 * a);
#pragma unroll
            for (int b = 0; b < 4; ++b) gw[l] = MFMA16(xa[b], xb[b], gw[l]);
          }
        }
#endif
        // ---- scalar chain for this lane's 4 pairs (p = it + 4*quarter + r, q)
        float w[NL][4];
        float gpx[4], gpy[4], gpz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pi = it + 4 * quarter + r;
          const int p = p0 + min(pi, m_out - 1);
          const bool ok = qv && pi < m_out && p != q;
          const float px = uvec_p[3 * (int64_t)p], py = uvec_p[3 * (int64_t)p + 1], pz = uvec_p[3 * (int64_t)p + 2];
#if LCAO_ABL & 256
          {  // diagnostic: no per-pair scalar chain
#pragma unroll
            for (int l = 0; l < NL; ++l) w[l][r] = ok ? gw[l][r] * G[0] : 0.f;
            gpx[r] = gpy[r] = gpz[r] = px * qx;
            continue;
          }
#endif
          const float c = px * qx + py * qy + pz * qz;
          float y[4], dy[4], GY[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int l = 0; l < 4; ++l) { y[l] = l < NL ? lcao_y_l0(c, l) : 0.f; dy[l] = l < NL ? lcao_y_l0_grad(c, l) : 0.f; }
          const float m = lcao_factor_quadratic<NL>(G, y, GY);   // G holds the edge's triangular factor: m = |R y|^2, GY = R^T R y
          const float mc = fmaxf(m, 0.f);
          const bool live = mc > 1e-24f;
          const float inv = live ? __builtin_amdgcn_rsqf(mc) : 1e12f;
          float g_inv = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) g_inv += gw[l][r] * y[l];
          const float g_m = (ok && live) ? -0.5f * g_inv * inv * inv * inv : 0.f;
          float gcv = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) gcv += (gw[l][r] * inv + 2.0f * g_m * GY[l]) * dy[l];
          gcv = ok ? gcv : 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            w[l][r] = ok ? y[l] * inv : 0.f;
#pragma unroll
            for (int l2 = l; l2 < NL; ++l2) gG[gram_idx(l, l2)] += g_m * (l == l2 ? y[l] * y[l] : 2.0f * y[l] * y[l2]);
          }
          gqx += gcv * px; gqy += gcv * py; gqz += gcv * pz;
          gpx[r] = gcv * qx; gpy[r] = gcv * qy; gpz[r] = gcv * qz;
        }
        // g_up[p] += sum over the 16 in-edges of this tile (lanes n within each 16-lane row), one writer lane per row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sx = lcao_row16_sum(gpx[r]), sy = lcao_row16_sum(gpy[r]), sz = lcao_row16_sum(gpz[r]);
          const int pi = it + 4 * quarter + r;
          if (n == 0 && pi < m_out) {
            float* o = g_up + 3 * (int64_t)(p0 + pi);
            o[0] += sx; o[1] += sy; o[2] += sz;
          }
        }
        // ---- GEMM 2: gA[l][t] (channel row, in-edge n) += sum_p g_tbw[p][channel] w[l][p, q_n]
        //      k-step r: quarter `quarter` contributes p = it + 4*quarter + r (B operand = w[l][r]).
        //      V >= 64: the A operand is read as float4 (channels 64h + 4n .. +3); its four floats feed the four channel
        //      tiles 4h + j, so tile 4h + j holds channels 64h + 4i + j in its rows i (a permuted tiling that the store
        //      below undoes) — 2 x 16-B loads per k-step instead of 8 x 4-B.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pr = p0 + min(it + 4 * quarter + r, m_out - 1);
          if constexpr (V >= 64) {
            const float* gtr = g_tbw + (int64_t)pr * V + 4 * n;
#if LCAO_ABL & 512
            {
              const f32x4v av = *(const f32x4v*)(gtr);
#pragma unroll
              for (int l = 0; l < NL; ++l) gA[l][0] = MFMA16(av[0], w[l][r], gA[l][0]);
            }
#else
#pragma unroll
            for (int h = 0; h < V / 64; ++h) {
              const f32x4v av = *(const f32x4v*)(gtr + 64 * h);
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < NL; ++l) gA[l][4 * h + j] = MFMA16(av[j], w[l][r], gA[l][4 * h + j]);
            }
#endif
          } else {
            const float* gtr = g_tbw + (int64_t)pr * V + n;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              const float av = gtr[16 * t];
#pragma unroll
              for (int l = 0; l < NL; ++l) gA[l][t] = MFMA16(av, w[l][r], gA[l][t]);
            }
          }
        }
      }
      // ---- this in-tile is complete: reduce the per-lane partials over the four quarters, write the rows of q
#pragma unroll
      for (int i = 0; i < 10; ++i) { gG[i] += __shfl_xor(gG[i], 16); gG[i] += __shfl_xor(gG[i], 32); }
      gqx += __shfl_xor(gqx, 16); gqx += __shfl_xor(gqx, 32);
      gqy += __shfl_xor(gqy, 16); gqy += __shfl_xor(gqy, 32);
      gqz += __shfl_xor(gqz, 16); gqz += __shfl_xor(gqz, 32);
      if (qv) {
        if (quarter == 0) {
#pragma unroll
          for (int i = 0; i < 10; ++i) g_gram[(int64_t)q * 10 + i] = gG[i];
          g_uq[3 * (int64_t)q] = gqx; g_uq[3 * (int64_t)q + 1] = gqy; g_uq[3 * (int64_t)q + 2] = gqz;
        }
        if constexpr (V >= 64) {
          // tile 4h + j, accumulator register rr of lane (q, quarter): channel 64h + 4 (4 quarter + rr) + j
          float* go = g_Ag + (int64_t)q * NL * V + 16 * quarter;
#pragma unroll
          for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int h = 0; h < V / 64; ++h)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const f32x4v v = f32x4v{gA[l][4 * h][rr], gA[l][4 * h + 1][rr], gA[l][4 * h + 2][rr], gA[l][4 * h + 3][rr]};
                *(f32x4v*)(go + l * V + 64 * h + 4 * rr) = v;
              }
        } else {
          float* go = g_Ag + (int64_t)q * NL * V + 4 * quarter;
#pragma unroll
          for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int t = 0; t < KT; ++t) *(f32x4v*)(go + l * V + 16 * t) = gA[l][t];
        }
      }
    }
  }
}

// The same reverse in two phases per in-edge tile, so that neither the Ag tile nor the weights are fetched twice:
//   phase A1 the lane's Ag rows (all l, all channels: NL x KT float4) stay in registers while GEMM 1 runs over every out-edge
//            tile; GW[l][r] of a tile is parked in a lane-private LDS column
//   phase A2 the scalar chain of every (p, q) pair: GW read back, the weights w[l][r] written to the same slots
//   phase B  GEMM 2 over the same out-edge tiles with the weights read back, accumulators live only here
// Register live ranges of the Ag tile (96) and of the g_Ag accumulators (96) no longer overlap (the fused loop above needs
// both plus the chain: 254 registers, 2 waves per SIMD, and re-reads the Ag tile for every out-edge tile).  Out-lists longer
// than 16 NITC edges are processed in chunks; later chunks add into the g_Ag rows this wave has already written.
#ifndef LCAO_TBWD_WAVES
#define LCAO_TBWD_WAVES 2
#endif
// A SHORT in-edge tile (its in-edges x n_l fit 16 MFMA columns: the 2 left over from a degree-18 list) with the pair
// (in-edge q, l) on the MFMA column instead of q alone: both GEMMs then take V / 16 x 4 MFMAs per out-tile instead of
// n_l x that, for the same result (a full tile spends its 16 columns on 16 in-edges; this one had 2 of them valid).
// Lane (column j, quarter): q = in-edge j / NL of the tile, l = j % NL.  GEMM 1 leaves gw[r] = g_tbw[p] . Ag[q, l] for
// p = it + 4 r + quarter; the chain of a pair needs the n_l values of its in-edge, which sit in n_l neighbouring lanes
// (gathered by ds_bpermute); every one of those lanes runs the chain, the lane with l = 0 alone adds to the per-edge sums.
template <int V, int NL, int NITC>
__device__ __forceinline__ void tbwd_short_in_tile(
    const float* __restrict__ g_tbw, const float* __restrict__ Ag, const float* __restrict__ gram,
    const float* __restrict__ uvec_p, const int32_t* __restrict__ in_pos, int j0, int jt, int m_in, int p0, int m_out,
    float* __restrict__ wl, float* __restrict__ g_Ag, float* __restrict__ g_gram, float* __restrict__ g_uq,
    float* __restrict__ g_up) {
  constexpr int KT = V / 16;
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, quarter = lane >> 4;
  const int qi = n / NL, lm = n - qi * NL;                 // this column's in-edge (within the tile) and l
  const bool qv = (jt + qi) < m_in && qi * NL + lm < 16;
  const int q = in_pos[j0 + (qv ? jt + qi : 0)];
  const float qx = uvec_p[3 * (int64_t)q], qy = uvec_p[3 * (int64_t)q + 1], qz = uvec_p[3 * (int64_t)q + 2];
  float G[10], gG[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) { G[i] = gram[(int64_t)q * 10 + i]; gG[i] = 0.f; }
  float gqx = 0.f, gqy = 0.f, gqz = 0.f;
  const float* agq = Ag + ((int64_t)q * NL + lm) * V + 4 * quarter;
  const int lane_l0 = (lane & 48) + qi * NL;               // lane holding l = 0 of this in-edge (same quarter)
  for (int ic = 0; ic < m_out; ic += 16 * NITC) {
    asm volatile("" ::: "memory");
    const int nit = min(NITC, (m_out - ic + 15) >> 4);
    {  // ---- phase A1: GEMM 1, one accumulator per out-tile
      f32x4v xb[KT];
#pragma unroll
      for (int a = 0; a < KT; ++a) xb[a] = *(const f32x4v*)(agq + 16 * a);
      for (int t = 0; t < nit; ++t) {
        asm volatile("" ::: "memory");
        const int it = ic + 16 * t;
        f32x4v gw = f32x4v{0.f, 0.f, 0.f, 0.f};
        const int rem = m_out - it;
        if (rem <= 4) {
          for (int pp = 0; pp < rem; ++pp) {   // wave-uniform trip count <= 4
            const float* gtp = g_tbw + (int64_t)(p0 + it + pp) * V + 4 * quarter;
            f32x4v t4 = *(const f32x4v*)(gtp) * xb[0];
#pragma unroll
            for (int a = 1; a < KT; ++a) t4 += *(const f32x4v*)(gtp + 16 * a) * xb[a];
            float sl = (t4[0] + t4[1]) + (t4[2] + t4[3]);
            sl += __shfl_xor(sl, 16);
            sl += __shfl_xor(sl, 32);
            gw[0] = quarter == pp ? sl : gw[0];
          }
        } else {
          const int pa = p0 + min(it + 4 * (n & 3) + (n >> 2), m_out - 1);
          const float* gta = g_tbw + (int64_t)pa * V + 4 * quarter;
          f32x4v xa[KT];
#pragma unroll
          for (int a = 0; a < KT; ++a) xa[a] = *(const f32x4v*)(gta + 16 * a);
#pragma unroll
          for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) gw = MFMA16(xa[a][b], xb[a][b], gw);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) wl[64 * (t * 4 + r)] = gw[r];
      }
    }
    asm volatile("" ::: "memory");
    // ---- phase A2: the scalar chain; the n_l GW values of an in-edge come from its n_l lanes
    for (int t = 0; t < nit; ++t) {
      asm volatile("" ::: "memory");
      const int it = ic + 16 * t;
      float w[4], gpx[4], gpy[4], gpz[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w[r] = 0.f;
        gpx[r] = gpy[r] = gpz[r] = 0.f;
        if (it + 4 * r >= m_out) continue;
        const float mine = wl[64 * (t * 4 + r)];
        float gwl[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) gwl[l] = __shfl(mine, lane_l0 + l);
        const int pi = it + 4 * r + quarter;
        const int p = p0 + min(pi, m_out - 1);
        const bool ok = qv && pi < m_out && p != q;
        const float px = uvec_p[3 * (int64_t)p], py = uvec_p[3 * (int64_t)p + 1], pz = uvec_p[3 * (int64_t)p + 2];
        const float c = px * qx + py * qy + pz * qz;
        float y[4], dy[4], GY[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) { y[l] = l < NL ? lcao_y_l0(c, l) : 0.f; dy[l] = l < NL ? lcao_y_l0_grad(c, l) : 0.f; }
        const float m = lcao_factor_quadratic<NL>(G, y, GY);
        const float mc = fmaxf(m, 0.f);
        const bool live = mc > 1e-24f;
        const float inv = live ? __builtin_amdgcn_rsqf(mc) : 1e12f;
        float g_inv = 0.f;
#pragma unroll
        for (int l = 0; l < NL; ++l) g_inv += gwl[l] * y[l];
        const bool first = ok && lm == 0;                   // one lane per (p, q) pair feeds the per-edge / per-pair sums
        const float g_m = (first && live) ? -0.5f * g_inv * inv * inv * inv : 0.f;
        float gcv = 0.f;
#pragma unroll
        for (int l = 0; l < NL; ++l) gcv += (gwl[l] * inv + 2.0f * g_m * GY[l]) * dy[l];
        gcv = first ? gcv : 0.f;
        float ysel = y[0];
#pragma unroll
        for (int l = 1; l < NL; ++l) ysel = lm == l ? y[l] : ysel;
        w[r] = ok ? ysel * inv : 0.f;
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int l2 = l; l2 < NL; ++l2) gG[gram_idx(l, l2)] += g_m * (l == l2 ? y[l] * y[l] : 2.0f * y[l] * y[l2]);
        gqx += gcv * px; gqy += gcv * py; gqz += gcv * pz;
        gpx[r] = gcv * qx; gpy[r] = gcv * qy; gpz[r] = gcv * qz;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (it + 4 * r >= m_out) continue;
        const float sx = lcao_row16_sum(gpx[r]), sy = lcao_row16_sum(gpy[r]), sz = lcao_row16_sum(gpz[r]);
        const int pi = it + 4 * r + quarter;
        if (n == 0 && pi < m_out) {
          float* o = g_up + 3 * (int64_t)(p0 + pi);
          o[0] += sx; o[1] += sy; o[2] += sz;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) wl[64 * (t * 4 + r)] = w[r];
    }
    asm volatile("" ::: "memory");
    {  // ---- phase B: GEMM 2 with the weights of this column's (in-edge, l)
      f32x4v gA[KT];
#pragma unroll
      for (int t2 = 0; t2 < KT; ++t2) gA[t2] = f32x4v{0.f, 0.f, 0.f, 0.f};
      for (int t = 0; t < nit; ++t) {
        asm volatile("" ::: "memory");
        const int it = ic + 16 * t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (it + 4 * r >= m_out) continue;
          const float wr = wl[64 * (t * 4 + r)];
          const int pr = p0 + min(it + 4 * r + quarter, m_out - 1);
          if constexpr (TBWD_LINE_STORES) {
            const float* gtr = g_tbw + (int64_t)pr * V + n;
#pragma unroll
            for (int t2 = 0; t2 < KT; ++t2) gA[t2] = MFMA16(gtr[16 * t2], wr, gA[t2]);
          } else {
            const float* gtr = g_tbw + (int64_t)pr * V + 4 * n;
#pragma unroll
            for (int h = 0; h < V / 64; ++h) {
              const f32x4v av = *(const f32x4v*)(gtr + 64 * h);
#pragma unroll
              for (int j = 0; j < 4; ++j) gA[4 * h + j] = MFMA16(av[j], wr, gA[4 * h + j]);
            }
          }
        }
      }
      if (qv) {
        if constexpr (TBWD_LINE_STORES) {
          float* go = g_Ag + ((int64_t)q * NL + lm) * V + 4 * quarter;
#pragma unroll
          for (int t2 = 0; t2 < KT; ++t2) {
            f32x4v v = gA[t2];
            float* dst = go + 16 * t2;
            if (ic > 0) v += *(const f32x4v*)dst;
            *(f32x4v*)dst = v;
          }
        } else {
          float* go = g_Ag + ((int64_t)q * NL + lm) * V + 16 * quarter;
#pragma unroll
          for (int h = 0; h < V / 64; ++h)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              f32x4v v = f32x4v{gA[4 * h][rr], gA[4 * h + 1][rr], gA[4 * h + 2][rr], gA[4 * h + 3][rr]};
              float* dst = go + 64 * h + 4 * rr;
              if (ic > 0) v += *(const f32x4v*)dst;
              *(f32x4v*)dst = v;
            }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) { gG[i] += __shfl_xor(gG[i], 16); gG[i] += __shfl_xor(gG[i], 32); }
  gqx += __shfl_xor(gqx, 16); gqx += __shfl_xor(gqx, 32);
  gqy += __shfl_xor(gqy, 16); gqy += __shfl_xor(gqy, 32);
  gqz += __shfl_xor(gqz, 16); gqz += __shfl_xor(gqz, 32);
  if (qv && quarter == 0 && lm == 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) g_gram[(int64_t)q * 10 + i] = gG[i];
    g_uq[3 * (int64_t)q] = gqx; g_uq[3 * (int64_t)q + 1] = gqy; g_uq[3 * (int64_t)q + 2] = gqz;
  }
}

template <int V, int NL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LCAO_TBWD_WAVES, LCAO_TBWD_WAVES))) void k_triplet_bwd_split(
    const float* __restrict__ g_tbw, const float* __restrict__ Ag, const float* __restrict__ gram,
    const float* __restrict__ uvec_p, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_pos,
    const int32_t* __restrict__ out_ptr, int n_nodes, float* __restrict__ g_Ag, float* __restrict__ g_gram,
    float* __restrict__ g_uq, float* __restrict__ g_up) {
  constexpr int KT = V / 16;
  constexpr int NITC = 16 / NL > 8 ? 8 : 16 / NL;   // out-edge tiles per chunk: NITC x NL x 4 floats per lane parked in LDS
  extern __shared__ float wpark[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, quarter = lane >> 4;
  float* wl = wpark + wave * (NITC * NL * 4 * 64) + lane;   // element k of this lane: wl[64 k]
  for (int s = blockIdx.x * 4 + wave; s < n_nodes; s += gridDim.x * 4) {
    const int p0 = out_ptr[s], m_out = out_ptr[s + 1] - p0;
    const int j0 = in_ptr[s], m_in = in_ptr[s + 1] - j0;
    for (int it0 = 0; it0 < m_out; it0 += 16)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pi = it0 + 4 * quarter + r;
        if (n == 0 && pi < m_out) {
          float* o = g_up + 3 * (int64_t)(p0 + pi);
          o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
        }
      }
    for (int jt = 0; jt < m_in; jt += 16) {
      if (NL > 1 && (m_in - jt) * NL <= 16) {   // wave-uniform: a short last tile, (in-edge, l) pairs on the MFMA columns
        tbwd_short_in_tile<V, NL, NITC>(g_tbw, Ag, gram, uvec_p, in_pos, j0, jt, m_in, p0, m_out, wl, g_Ag, g_gram, g_uq, g_up);
        continue;
      }
      const bool qv = (jt + n) < m_in;
      const int q = in_pos[j0 + (qv ? jt + n : 0)];
      const float qx = uvec_p[3 * (int64_t)q], qy = uvec_p[3 * (int64_t)q + 1], qz = uvec_p[3 * (int64_t)q + 2];
      float G[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) G[i] = gram[(int64_t)q * 10 + i];
      float gG[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) gG[i] = 0.f;
      float gqx = 0.f, gqy = 0.f, gqz = 0.f;
      const float* agq = Ag + (int64_t)q * NL * V + 4 * quarter;
      for (int ic = 0; ic < m_out; ic += 16 * NITC) {
        asm volatile("" ::: "memory");   // the Ag tile is loaded per chunk: hoisted out of this loop it would stay live through phase B
        const int nit = min(NITC, (m_out - ic + 15) >> 4);
        {  // ---------------- phase A1: GEMM 1 over the out-edge tiles of this chunk, the Ag tile resident in registers
          f32x4v xb[NL][KT];
#pragma unroll
          for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int a = 0; a < KT; ++a) xb[l][a] = *(const f32x4v*)(agq + l * V + 16 * a);
          for (int t = 0; t < nit; ++t) {
            asm volatile("" ::: "memory");
            const int it = ic + 16 * t;
            // MFMA row n of this tile is out-edge it + 4 (n % 4) + n / 4: accumulator register r of quarter qu then holds
            // out-edge it + 4 r + qu, i.e. k-step r of GEMM 2 covers four CONSECUTIVE out-edges and a tile with few valid
            // rows (degree 18 = 16 + 2) needs one k-step and one chain pass instead of four
            f32x4v gw[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) gw[l] = f32x4v{0.f, 0.f, 0.f, 0.f};
            const int rem = m_out - it;   // wave-uniform
            if (rem <= 4) {
              // A tile with at most four out-edges (the 2 of a degree-18 list): its 16 x 16 MFMA tile would be 3/4 empty
              // rows at full price (96 MFMAs).  The <= 4 x 16 x NL dot products run on the VALU instead: this lane's 32
              // channels of its in-edge against the same channels of each out-edge row (a broadcast load per 16 lanes),
              // then the four quarters are summed.  rem x 32 NL FMAs per lane, ~1/4 of the MFMA time at rem = 2.
              for (int pp = 0; pp < rem; ++pp) {   // wave-uniform trip count <= 4
                const float* gtp = g_tbw + (int64_t)(p0 + it + pp) * V + 4 * quarter;
                f32x4v ga[KT];
#pragma unroll
                for (int a = 0; a < KT; ++a) ga[a] = *(const f32x4v*)(gtp + 16 * a);
                float sl[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                  f32x4v t4 = ga[0] * xb[l][0];
#pragma unroll
                  for (int a = 1; a < KT; ++a) t4 += ga[a] * xb[l][a];
                  sl[l] = (t4[0] + t4[1]) + (t4[2] + t4[3]);
                  sl[l] += __shfl_xor(sl[l], 16);
                  sl[l] += __shfl_xor(sl[l], 32);
                  gw[l][0] = quarter == pp ? sl[l] : gw[l][0];   // register 0 of quarter qu is out-edge it + qu
                }
              }
            } else {
              const int pa = p0 + min(it + 4 * (n & 3) + (n >> 2), m_out - 1);
              const float* gta = g_tbw + (int64_t)pa * V + 4 * quarter;
              f32x4v xa[KT];
#pragma unroll
              for (int a = 0; a < KT; ++a) xa[a] = *(const f32x4v*)(gta + 16 * a);
#pragma unroll
              for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                  for (int b = 0; b < 4; ++b) gw[l] = MFMA16(xa[a][b], xb[l][a][b], gw[l]);
            }
            // park GW of this tile (lane-private LDS column; the chain below replaces it with the weights)
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
              for (int r = 0; r < 4; ++r) wl[64 * ((t * NL + l) * 4 + r)] = gw[l][r];
          }
        }
        asm volatile("" ::: "memory");
        {  // ---------------- phase A2: the scalar chain of every (p, q) pair; GW in, weights out (same LDS slots)
          for (int t = 0; t < nit; ++t) {
            asm volatile("" ::: "memory");
            const int it = ic + 16 * t;
            f32x4v gw[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
              for (int r = 0; r < 4; ++r) gw[l][r] = wl[64 * ((t * NL + l) * 4 + r)];
        // ---- scalar chain for this lane's 4 pairs (p = it + 4 r + quarter, q); passes r whose four out-edges all lie
        //      past the list are skipped (wave-uniform)
        float w[NL][4];
        float gpx[4], gpy[4], gpz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int l = 0; l < NL; ++l) w[l][r] = 0.f;
          gpx[r] = gpy[r] = gpz[r] = 0.f;
          if (it + 4 * r >= m_out) continue;
          const int pi = it + 4 * r + quarter;
          const int p = p0 + min(pi, m_out - 1);
          const bool ok = qv && pi < m_out && p != q;
          const float px = uvec_p[3 * (int64_t)p], py = uvec_p[3 * (int64_t)p + 1], pz = uvec_p[3 * (int64_t)p + 2];
          const float c = px * qx + py * qy + pz * qz;
          float y[4], dy[4], GY[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int l = 0; l < 4; ++l) { y[l] = l < NL ? lcao_y_l0(c, l) : 0.f; dy[l] = l < NL ? lcao_y_l0_grad(c, l) : 0.f; }
          const float m = lcao_factor_quadratic<NL>(G, y, GY);   // G holds the edge's triangular factor: m = |R y|^2, GY = R^T R y
          const float mc = fmaxf(m, 0.f);
          const bool live = mc > 1e-24f;
          const float inv = live ? __builtin_amdgcn_rsqf(mc) : 1e12f;
          float g_inv = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) g_inv += gw[l][r] * y[l];
          const float g_m = (ok && live) ? -0.5f * g_inv * inv * inv * inv : 0.f;
          float gcv = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) gcv += (gw[l][r] * inv + 2.0f * g_m * GY[l]) * dy[l];
          gcv = ok ? gcv : 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            w[l][r] = ok ? y[l] * inv : 0.f;
#pragma unroll
            for (int l2 = l; l2 < NL; ++l2) gG[gram_idx(l, l2)] += g_m * (l == l2 ? y[l] * y[l] : 2.0f * y[l] * y[l2]);
          }
          gqx += gcv * px; gqy += gcv * py; gqz += gcv * pz;
          gpx[r] = gcv * qx; gpy[r] = gcv * qy; gpz[r] = gcv * qz;
        }
        // g_up[p] += sum over the 16 in-edges of this tile (lanes n within each 16-lane row), one writer lane per row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (it + 4 * r >= m_out) continue;
          const float sx = lcao_row16_sum(gpx[r]), sy = lcao_row16_sum(gpy[r]), sz = lcao_row16_sum(gpz[r]);
          const int pi = it + 4 * r + quarter;
          if (n == 0 && pi < m_out) {
            float* o = g_up + 3 * (int64_t)(p0 + pi);
            o[0] += sx; o[1] += sy; o[2] += sz;
          }
        }
            // park the weights of this tile: lane-private, conflict-free
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
              for (int r = 0; r < 4; ++r) wl[64 * ((t * NL + l) * 4 + r)] = w[l][r];
          }
        }
        asm volatile("" ::: "memory");
        {  // ---------------- phase B: GEMM 2 over the same tiles
          f32x4v gA[NL][KT];
#pragma unroll
          for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int t2 = 0; t2 < KT; ++t2) gA[l][t2] = f32x4v{0.f, 0.f, 0.f, 0.f};
          for (int t = 0; t < nit; ++t) {
            asm volatile("" ::: "memory");
            const int it = ic + 16 * t;
            float w[NL][4];
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
              for (int r = 0; r < 4; ++r) w[l][r] = wl[64 * ((t * NL + l) * 4 + r)];
        // ---- GEMM 2: gA[l][t] (channel row, in-edge n) += sum_p g_tbw[p][channel] w[l][p, q_n]
        //      k-step r: quarter `quarter` contributes p = it + 4 r + quarter (B operand = w[l][r]); k-steps past the list
        //      are skipped.
        //      V >= 64: the A operand is read as float4 (channels 64h + 4n .. +3); its four floats feed the four channel
        //      tiles 4h + j, so tile 4h + j holds channels 64h + 4i + j in its rows i (a permuted tiling that the store
        //      below undoes) — 2 x 16-B loads per k-step instead of 8 x 4-B.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (it + 4 * r >= m_out) continue;
          const int pr = p0 + min(it + 4 * r + quarter, m_out - 1);
          if constexpr (V >= 64 && !TBWD_LINE_STORES) {
            const float* gtr = g_tbw + (int64_t)pr * V + 4 * n;
#pragma unroll
            for (int h = 0; h < V / 64; ++h) {
              const f32x4v av = *(const f32x4v*)(gtr + 64 * h);
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < NL; ++l) gA[l][4 * h + j] = MFMA16(av[j], w[l][r], gA[l][4 * h + j]);
            }
          } else {
            const float* gtr = g_tbw + (int64_t)pr * V + n;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              const float av = gtr[16 * t];
#pragma unroll
              for (int l = 0; l < NL; ++l) gA[l][t] = MFMA16(av, w[l][r], gA[l][t]);
            }
          }
        }
          }
          // rows of q: written by the first chunk, added to by later ones (out-lists longer than 16 NITC edges)
          if (qv) {
            static_assert(V >= 64, "the split kernel serves V in {64, 128}");
            if constexpr (TBWD_LINE_STORES) {
              // tile t <-> channels 16 t + row: the four quarters of an in-edge write one whole 64-byte line per store
              float* go = g_Ag + (int64_t)q * NL * V + 4 * quarter;
#pragma unroll
              for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int t2 = 0; t2 < KT; ++t2) {
                  f32x4v v = gA[l][t2];
                  float* dst = go + l * V + 16 * t2;
                  if (ic > 0) v += *(const f32x4v*)dst;
                  *(f32x4v*)dst = v;
                }
            } else {
            float* go = g_Ag + (int64_t)q * NL * V + 16 * quarter;
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
              for (int h = 0; h < V / 64; ++h)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  f32x4v v = f32x4v{gA[l][4 * h][rr], gA[l][4 * h + 1][rr], gA[l][4 * h + 2][rr], gA[l][4 * h + 3][rr]};
                  float* dst = go + l * V + 64 * h + 4 * rr;
                  if (ic > 0) v += *(const f32x4v*)dst;
                  *(f32x4v*)dst = v;
                }
            }
          }
        }
      }
      // ---- this in-tile is complete: reduce the per-lane partials over the four quarters, write the scalars of q
#pragma unroll
      for (int i = 0; i < 10; ++i) { gG[i] += __shfl_xor(gG[i], 16); gG[i] += __shfl_xor(gG[i], 32); }
      gqx += __shfl_xor(gqx, 16); gqx += __shfl_xor(gqx, 32);
      gqy += __shfl_xor(gqy, 16); gqy += __shfl_xor(gqy, 32);
      gqz += __shfl_xor(gqz, 16); gqz += __shfl_xor(gqz, 32);
      if (qv && quarter == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) g_gram[(int64_t)q * 10 + i] = gG[i];
        g_uq[3 * (int64_t)q] = gqx; g_uq[3 * (int64_t)q + 1] = gqy; g_uq[3 * (int64_t)q + 2] = gqz;
      }
    }
  }
}

extern "C" int lcao_triplet_bwd_mfma(const float* g_tbw, const float* Ag, const float* gram, const float* uvec_p,
                                     const int32_t* in_ptr, const int32_t* in_pos, const int32_t* out_ptr,
                                     int64_t n_nodes, int64_t n_edges, int V, int n_l, float* g_Ag, float* g_gram,
                                     float* g_uq, float* g_up, void* stream) {
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0 || n_nodes == 0) return LCAO_OK;
  LCAO_CHECK_PTR(g_tbw); LCAO_CHECK_PTR(Ag); LCAO_CHECK_PTR(gram); LCAO_CHECK_PTR(uvec_p); LCAO_CHECK_PTR(in_ptr);
  LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(g_Ag); LCAO_CHECK_PTR(g_gram); LCAO_CHECK_PTR(g_uq);
  LCAO_CHECK_PTR(g_up);
  hipStream_t st = (hipStream_t)stream;
  int64_t groups = (n_nodes + 3) / 4;
  int64_t cap = (int64_t)lcao_num_cu() * 8;
  if (groups > cap) groups = cap;
  // V >= 64: the two-phase kernel (weights parked in LDS between the phases); LCAO_TBWD_FUSED=1 keeps the single-loop one (A/B)
  static int fused_only = -1;
  if (fused_only < 0) { const char* e_ = getenv("LCAO_TBWD_FUSED"); fused_only = (e_ && atoi(e_)) ? 1 : 0; }
#define TBM_CASE(V_, NL_)                                                                                          \
  if (V == V_ && n_l == NL_) {                                                                                     \
    if (V_ >= 64 && !fused_only) {                                                                                 \
      constexpr int NITC_ = 16 / NL_ > 8 ? 8 : 16 / NL_;                                                           \
      const size_t lds_ = size_t(4) * NITC_ * NL_ * 4 * 64 * sizeof(float);                                        \
      static bool done_ = false;                                                                                   \
      int rc_ = set_lds_attr(k_triplet_bwd_split<(V_ >= 64 ? V_ : 64), NL_>, lds_, done_);                         \
      if (rc_ != LCAO_OK) return rc_;                                                                              \
      hipLaunchKernelGGL((k_triplet_bwd_split<(V_ >= 64 ? V_ : 64), NL_>), dim3((unsigned)groups), dim3(256), lds_, st, \
                         g_tbw, Ag, gram, uvec_p, in_ptr, in_pos, out_ptr, (int)n_nodes, g_Ag, g_gram, g_uq, g_up);  \
      return lcao_launch_status();                                                                                 \
    }                                                                                                              \
    hipLaunchKernelGGL((k_triplet_bwd_mfma<V_, NL_>), dim3((unsigned)groups), dim3(256), 0, st, g_tbw, Ag, gram, uvec_p, \
                       in_ptr, in_pos, out_ptr, (int)n_nodes, g_Ag, g_gram, g_uq, g_up);                           \
    return lcao_launch_status();                                                                                   \
  }
#define TBM_V(V_) TBM_CASE(V_, 1) TBM_CASE(V_, 2) TBM_CASE(V_, 3) TBM_CASE(V_, 4)
  TBM_V(128) TBM_V(64) TBM_V(32)
#undef TBM_V
#undef TBM_CASE
  return LCAO_ERR_UNSUPPORTED;
}

// ===================================================================================================
// lcao_wgrad — weight gradient of a Dense layer applied to E rows:  dW (M x K) = G^T X,  G (E x M), X (E x K).
// A reduction over ~10^6 rows with a 128 x 128 result: every wave owns one 64 x 64 block of dW for one chunk of rows,
// 16 MFMA accumulators, the 4 rows of a k-step on the four lane quarters.  A lane loads 16 B of G and 16 B of X per step:
// its four floats feed four DIFFERENT feature tiles (tile j <-> features 4i + j, i = lane & 15), a permuted tiling that
// the epilogue undoes — so operands come straight from global memory as float4, no LDS, no transposition.  Chunk
// partials are summed in a fixed order by a second kernel (deterministic, no atomics).
// ===================================================================================================
#define LCAO_WGRAD_CHUNKS 2048

template <int MB>   // a wave owns a (64 MB) x 64 block of dW: MB = 2 halves the X loads per MFMA when M % 128 == 0
__global__ __launch_bounds__(256) void k_wgrad_partial(const float* __restrict__ G, const float* __restrict__ X,
                                                       int64_t E, int M, int K, int n_chunks,
                                                       float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int kb = K >> 6, n_blocks = (M / (64 * MB)) * kb;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;          // one (chunk, block) unit per wave
  const int chunk = (int)(gw / n_blocks), block = (int)(gw - (int64_t)chunk * n_blocks);
  if (chunk >= n_chunks) return;
  const int bm = block / kb, bk = block - bm * kb;
  const int64_t rows = ((E + n_chunks - 1) / n_chunks + 3) & ~int64_t(3);
  const int64_t r0 = chunk * rows, r1 = min(r0 + rows, E);
  f32x4v acc[4 * MB][4];
#pragma unroll
  for (int a = 0; a < 4 * MB; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const float* gp = G + 64 * MB * bm + 4 * i;
  const float* xp = X + 64 * bk + 4 * i;
  const f32x4v zero = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int n_steps = (int)((r1 - r0 + 3) >> 2);
  constexpr int U = MB == 1 ? 8 : 4;          // k-steps per iteration: U (1 + MB) independent 16-B loads in flight
  for (int s0 = 0; s0 < n_steps; s0 += U) {
    f32x4v g[U][MB], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t e = r0 + 4 * (int64_t)(s0 + u) + kq;
      const bool ok = e < r1;
#pragma unroll
      for (int h = 0; h < MB; ++h) g[u][h] = ok ? *(const f32x4v*)(gp + e * M + 64 * h) : zero;
      x[u] = ok ? *(const f32x4v*)(xp + e * K) : zero;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int h = 0; h < MB; ++h)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[4 * h + a][b] = MFMA16(g[u][h][a], x[u][b], acc[4 * h + a][b]);
  }
  // acc[4h + a][b] reg r, lane (i, kq): row index 4*kq + r of feature tile (h, a)  <->  m = 64 (MB bm + h) + 4 (4 kq + r) + a;
  // column i of tile b <-> k = 64 bk + 4 i + b.   The four b's of a lane are 16 contiguous bytes.
  float* out = partial + (int64_t)chunk * M * K;
#pragma unroll
  for (int h = 0; h < MB; ++h)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 64 * (MB * bm + h) + 4 * (4 * kq + r) + a;
        const f32x4v v = f32x4v{acc[4 * h + a][0][r], acc[4 * h + a][1][r], acc[4 * h + a][2][r], acc[4 * h + a][3][r]};
        *(f32x4v*)(out + (int64_t)m * K + 64 * bk + 4 * i) = v;
      }
}

__global__ void k_wgrad_reduce(const float* __restrict__ partial, int n_chunks, int64_t mk, float* __restrict__ dW) {
  const int64_t j = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
  if (j >= mk) return;
  f32x4v s0 = f32x4v{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int c = 0;
  for (; c + 4 <= n_chunks; c += 4) {   // fixed association: four interleaved running sums, then (s0+s1)+(s2+s3)
    s0 += *(const f32x4v*)(partial + (int64_t)c * mk + j);
    s1 += *(const f32x4v*)(partial + (int64_t)(c + 1) * mk + j);
    s2 += *(const f32x4v*)(partial + (int64_t)(c + 2) * mk + j);
    s3 += *(const f32x4v*)(partial + (int64_t)(c + 3) * mk + j);
  }
  for (; c < n_chunks; ++c) s0 += *(const f32x4v*)(partial + (int64_t)c * mk + j);
  *(f32x4v*)(dW + j) = (s0 + s1) + (s2 + s3);
}

static bool wgrad_wide(int M) { return M % 128 == 0 && M >= 256; }   // 128 x 64 blocks per wave pay off from M = 256
static int wgrad_blocks(int M, int K) { return (wgrad_wide(M) ? M / 128 : M / 64) * (K / 64); }

// number of row chunks: ~2048 (chunk, block) wave units, at least 256 rows per chunk
static int wgrad_chunks(int64_t E, int M, int K) {
  int64_t c = 2048 / wgrad_blocks(M, K);
  if (c > LCAO_WGRAD_CHUNKS) c = LCAO_WGRAD_CHUNKS;
  const int64_t by_rows = (E + 255) / 256;
  if (c > by_rows) c = by_rows;
  if (c < 1) c = 1;
  return (int)c;
}

extern "C" size_t lcao_wgrad_workspace_bytes(int64_t n_rows, int M, int K) {
  if (n_rows <= 0 || M <= 0 || K <= 0 || M % 64 || K % 64) return 0;
  return (size_t)wgrad_chunks(n_rows, M, K) * M * K * sizeof(float);
}

extern "C" int lcao_wgrad(const float* G, const float* X, int64_t n_rows, int M, int K, float* dW, void* workspace,
                          size_t ws_bytes, void* stream) {
  LCAO_CHECK_SIZE(n_rows);
  if (M <= 0 || K <= 0) return LCAO_ERR_INVALID;
  if (M % 64 || K % 64) return LCAO_ERR_UNSUPPORTED;
  LCAO_CHECK_PTR(dW);
  hipStream_t st = (hipStream_t)stream;
  if (n_rows == 0) {
    return lcao_zero_async(dW, sizeof(float) * M * K, st);
  }
  LCAO_CHECK_PTR(G); LCAO_CHECK_PTR(X); LCAO_CHECK_PTR(workspace);
  if (((uintptr_t)G | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)workspace) % 16) return LCAO_ERR_INVALID;
  const int chunks = wgrad_chunks(n_rows, M, K);
  if (ws_bytes < (size_t)chunks * M * K * sizeof(float)) return LCAO_ERR_INVALID;
  const unsigned groups = (unsigned)(((int64_t)chunks * wgrad_blocks(M, K) + 3) / 4);
  if (wgrad_wide(M))
    hipLaunchKernelGGL(k_wgrad_partial<2>, dim3(groups), dim3(256), 0, st, G, X, n_rows, M, K, chunks, (float*)workspace);
  else
    hipLaunchKernelGGL(k_wgrad_partial<1>, dim3(groups), dim3(256), 0, st, G, X, n_rows, M, K, chunks, (float*)workspace);
  const int64_t mk = (int64_t)M * K;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(lcao_blocks(mk / 4, 256)), dim3(256), 0, st, (const float*)workspace, chunks, mk,
                     dW);
  return lcao_launch_status();
}
