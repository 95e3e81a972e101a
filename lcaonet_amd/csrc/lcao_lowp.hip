// lcao_lowp.hip — BASELINE config 3: the fused edge kernels on the 16-bit matrix path (v_mfma_f32_16x16x32_{bf16,f16},
// 16x the fp32 matrix rate per instruction), fp32 accumulation.  Two precisions, one template (PREC):
//   PREC_BF16X3  every fp32 operand split into two bf16 terms, three MFMAs per product: ~1e-5 from fp32 (the fp32 bar)
//   PREC_F16     operands rounded once to fp16 (11 significant bits), ONE MFMA per product: ~3e-4, inside config 3's
//                stated tolerance of 1e-3, which a single bf16 MFMA (8 bits: 3.6e-3 on the MLP alone) does not meet
#include <type_traits>

#include "lcao_mfma.h"

enum { PREC_BF16X3 = 0, PREC_F16 = 1 };

// LDS prefetch distance of lowp_layer in steps (one fp16 fragment, or a hi/lo pair of bf16 fragments, per step)
#ifndef LCAO_LOWP_PD_H
#define LCAO_LOWP_PD_H 4
#endif
#ifndef LCAO_LOWP_PD_BF
#define LCAO_LOWP_PD_BF 2
#endif
#ifndef LCAO_LOWP_THREADS
#define LCAO_LOWP_THREADS 512
#endif

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

// Weight planes in LDS, fragment-major: the 16 bytes a lane feeds to one MFMA (weight row rr, k-step s, lane group q) live
// at element ((4 s + q) * rows + rr) * 8, so the 16 lanes of every ds_read_b128 service group (MI355X_MICROARCH.md, LDS
// table) read 16 distinct 16-B slots: conflict-free, no padding.  Round 2's [row][K + 8] layout put (r, q + 1) and
// (r + 1, q) into the same slot: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE (profiles/r03/k2_lowp_pmc_before.txt).
// PERMUTE stores the columns in the order a chained layer delivers its inputs (see header).
template <int PREC, bool PERMUTE>
__device__ __forceinline__ void stage_weights_lowp(typename LowP<PREC>::T* __restrict__ planes, const float* __restrict__ W,
                                                   int rows, int K, float scale = 1.0f) {
  typedef typename LowP<PREC>::T T;
  T* hi = planes;
  T* lo = planes + rows * K;
  for (int i = threadIdx.x; i < rows * K; i += blockDim.x) {
    const int rr = i / K, pos = i - rr * K;
    int col = pos;
    if (PERMUTE) {  // pos = 32 s + 8 g + j  ->  column 32 s + (j < 4 ? 4 g + j : 16 + 4 g + (j - 4))
      const int s = pos >> 5, g = (pos >> 3) & 3, j = pos & 7;
      col = 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
    }
    const float w = W[(int64_t)rr * K + col] * scale;
    const T h = (T)w;
    const int dst = ((pos >> 3) * rows + rr) * 8 + (pos & 7);
    hi[dst] = h;
    if (PREC == PREC_BF16X3) lo[dst] = (T)(w - (float)h);
  }
}

// the B operand of one 32-wide k-step: 8 values per lane, rounded once to fp16 or split into two bf16 terms
template <int PREC> struct LowpOperand;
template <> struct LowpOperand<PREC_F16> {
  f16x8v hi;
  __device__ __forceinline__ void set(const float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (_Float16)v[j];
  }
};
template <> struct LowpOperand<PREC_BF16X3> {
  bf16x8v hi, lo;
  __device__ __forceinline__ void set(const float (&v)[8]) { split8(v, hi, lo); }
};

// one Dense layer on a 16-row tile: out[m] += W[16m.., :] . in, the input given as KS B operands (one per 32-wide k-step).
// The (k-step, feature tile) sequence is flattened and the weight fragment of step t + PD is requested from LDS before the
// MFMA(s) of step t issue (counted lgkmcnt waits): a 16-cycle v_mfma_f32_16x16x32 covers none of a ds_read_b128's latency.
template <int PREC, int KS, int MT>
__device__ __forceinline__ void lowp_layer(const typename LowP<PREC>::T* __restrict__ planes, int rows, int K, int r, int q,
                                           const LowpOperand<PREC> (&in)[KS], f32x4v (&out)[MT]) {
  constexpr int STEPS = KS * MT;
  const int fs = 4 * rows * 8;   // elements between the fragments of consecutive k-steps
  if constexpr (PREC == PREC_BF16X3) {
    constexpr int PD = LCAO_LOWP_PD_BF < STEPS ? LCAO_LOWP_PD_BF : STEPS;
    const __bf16* wh = planes + (q * rows + r) * 8;
    const __bf16* wl = wh + rows * K;
    bf16x8v fh[STEPS], fl[STEPS];
#pragma unroll
    for (int t = 0; t < PD; ++t) {
      fh[t] = *(const bf16x8v*)(wh + (t / MT) * fs + 128 * (t % MT));
      fl[t] = *(const bf16x8v*)(wl + (t / MT) * fs + 128 * (t % MT));
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
      const int s = t / MT, m = t % MT;
      if (t + PD < STEPS) {
        const int t2 = t + PD;
        fh[t2] = *(const bf16x8v*)(wh + (t2 / MT) * fs + 128 * (t2 % MT));
        fl[t2] = *(const bf16x8v*)(wl + (t2 / MT) * fs + 128 * (t2 % MT));
      }
      out[m] = MFMA16BF(fh[t], in[s].lo, out[m]);
      out[m] = MFMA16BF(fl[t], in[s].hi, out[m]);
      out[m] = MFMA16BF(fh[t], in[s].hi, out[m]);
      if (t + PD < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    }
  } else {
    constexpr int PD = LCAO_LOWP_PD_H < STEPS ? LCAO_LOWP_PD_H : STEPS;
    const _Float16* w0 = planes + (q * rows + r) * 8;
    f16x8v fr[STEPS];
#pragma unroll
    for (int t = 0; t < PD; ++t) fr[t] = *(const f16x8v*)(w0 + (t / MT) * fs + 128 * (t % MT));
    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
      const int s = t / MT, m = t % MT;
      if (t + PD < STEPS && !((LCAO_ABL & 4) && t + PD >= 2 * PD))
        fr[t + PD] = *(const f16x8v*)(w0 + ((t + PD) / MT) * fs + 128 * ((t + PD) % MT));
      out[m] = MFMA16H(fr[(LCAO_ABL & 4) ? t % (2 * PD) : t], in[s].hi, out[m]);
      if (t + PD < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Packed coefficient rows: the B operands of layer 1, prepared once per forward instead of once per (layer, edge, orbital)
// — cst is the same tensor in every interaction (lcaonet.py:533-534).  Per group of 8 consecutive values:
//   f16     16 bytes: the 8 values rounded to fp16
//   bf16x3  32 bytes: 8 bf16 high terms, then 8 bf16 low terms (x - float(hi))
// so a lane reads the operand of one k-step with one (two) 16-byte load(s) and issues no conversion.
// ---------------------------------------------------------------------------------------------------
template <int PREC>
__global__ void k_lowp_pack_rows(const float* __restrict__ rows, int64_t n_groups, unsigned char* __restrict__ out) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x) {
    const f32x4v a = *(const f32x4v*)(rows + 8 * g), b = *(const f32x4v*)(rows + 8 * g + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (PREC == PREC_F16) {
      f16x8v h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
      *(f16x8v*)(out + 16 * g) = h;
    } else {
      bf16x8v hi, lo;
      split8(v, hi, lo);
      *(bf16x8v*)(out + 32 * g) = hi;
      *(bf16x8v*)(out + 32 * g + 16) = lo;
    }
  }
}

extern "C" size_t lcao_lowp_packed_bytes(int64_t n_values, int precision) {
  return (size_t)(n_values / 8) * (precision == LCAO_PREC_F16 ? 16 : 32);
}

extern "C" int lcao_lowp_pack_rows(const float* rows, int64_t n_values, int precision, void* out, void* stream) {
  if (n_values < 0 || n_values % 8 || (precision != LCAO_PREC_F16 && precision != LCAO_PREC_BF16X3)) return LCAO_ERR_INVALID;
  if (n_values == 0) return LCAO_OK;
  LCAO_CHECK_PTR(rows); LCAO_CHECK_PTR(out);
  const int64_t groups = n_values / 8;
  hipStream_t st = (hipStream_t)stream;
  if (precision == LCAO_PREC_F16)
    hipLaunchKernelGGL(k_lowp_pack_rows<PREC_F16>, dim3(lcao_blocks(groups, 256, 8192)), dim3(256), 0, st, rows, groups,
                       (unsigned char*)out);
  else
    hipLaunchKernelGGL(k_lowp_pack_rows<PREC_BF16X3>, dim3(lcao_blocks(groups, 256, 8192)), dim3(256), 0, st, rows, groups,
                       (unsigned char*)out);
  return lcao_launch_status();
}

// the operand of k-step s (values 32 s + 8 q ..) of a packed row
template <int PREC>
__device__ __forceinline__ void load_packed(LowpOperand<PREC>& op, const unsigned char* __restrict__ row, int s, int q) {
  if constexpr (PREC == PREC_F16) {
    op.hi = *(const f16x8v*)(row + 16 * (4 * s + q));
  } else {
    op.hi = *(const bf16x8v*)(row + 32 * (4 * s + q));
    op.lo = *(const bf16x8v*)(row + 32 * (4 * s + q) + 16);
  }
}

// Diagnostic build (-DLCAO_LOWP_CLK, tools/k2_lowp_time.py; never in the shipped library): per-wave cycle totals of the
// kernel's phases and the in-kernel clock (s_memtime / s_memrealtime), written to a buffer no kernel reads.
#ifdef LCAO_LOWP_CLK
static unsigned long long* g_lowp_clk = nullptr;
extern "C" void lcao_debug_set_lowp_clock(void* buf) { g_lowp_clk = (unsigned long long*)buf; }
#define LOWP_STAMP(i)                                                  \
  do {                                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
    ph[i] += t_ - tl;                                                  \
    tl = t_;                                                           \
  } while (0)
#define LOWP_CLK_ARG , unsigned long long* __restrict__ clk
#define LOWP_CLK_PASS , g_lowp_clk
#else
#define LOWP_STAMP(i)
#define LOWP_CLK_ARG
#define LOWP_CLK_PASS
#endif

typedef float f32x2v __attribute__((ext_vector_type(2)));

// g(x') = x' / (1 + 2^(-x')) on a pair (lcao_mfma.h) in three stages that the weaving below issues in DIFFERENT steps, so
// no instruction waits on the transcendental before it:  a: two v_exp_f32   b: v_pk_add_f32, two v_rcp_f32   c: v_pk_mul_f32
__device__ __forceinline__ f32x2v lowp_act_a(f32x2v x) {
  return f32x2v{__builtin_amdgcn_exp2f(-x[0]), __builtin_amdgcn_exp2f(-x[1])};
}
__device__ __forceinline__ f32x2v lowp_act_b(f32x2v e) {
  e = e + 1.0f;
  return f32x2v{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}
__device__ __forceinline__ f32x2v lowp_act_c(f32x2v x, f32x2v rcp) {
#if LCAO_ABL & 1
  return x;
#else
  return x * rcp;
#endif
}
__device__ __forceinline__ f32x2v lowp_act2(f32x2v x) { return lowp_act_c(x, lowp_act_b(lowp_act_a(x))); }

// One Dense layer of the coefficient kernel with a second, independent instruction stream woven in: after the MFMA(s) of
// every (k-step, feature tile) step t, `valu(t)` emits a slice of VALU / transcendental work that does not depend on this
// layer's accumulators (activations of the OTHER layer), and a full scheduling fence pins the order, so the wave's own
// stream alternates matrix and vector instructions.  On gfx950 a wave's VALU stream and its SIMD partner's
// v_mfma_f32_16x16x32 stream overlap (profiles/r03/mfma16_valu_overlap.txt: both = max), but only while the two waves are
// in different phases; with the streams mixed inside each wave every issue slot has work of both kinds whatever the
// partner does.  (sched_group_barrier pipelines over blocks of this size were not honoured: the activations were
// emitted in runs ahead of the MFMAs.)
template <int PREC, int KS, int MT, typename InFn, typename ValuFn>
__device__ __forceinline__ void lowp_layer_mix(const typename LowP<PREC>::T* __restrict__ planes, int rows, int K, int r, int q,
                                               InFn in, f32x4v (&out)[MT], ValuFn valu) {
  constexpr int STEPS = KS * MT;
  const int fs = 4 * rows * 8;   // elements between the fragments of consecutive k-steps
  constexpr int PD0 = PREC == PREC_BF16X3 ? LCAO_LOWP_PD_BF : LCAO_LOWP_PD_H;
  constexpr int PD = PD0 < STEPS ? PD0 : STEPS;
  typedef typename LowP<PREC>::T T;
  typedef T t8 __attribute__((ext_vector_type(8)));
  const T* wh = planes + (q * rows + r) * 8;
  const T* wl = wh + rows * K;
  t8 fh[STEPS], fl[PREC == PREC_BF16X3 ? STEPS : 1];
#pragma unroll
  for (int t = 0; t < PD; ++t) {
    fh[t] = *(const t8*)(wh + (t / MT) * fs + 128 * (t % MT));
    if constexpr (PREC == PREC_BF16X3) fl[t] = *(const t8*)(wl + (t / MT) * fs + 128 * (t % MT));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < STEPS; ++t) {
    const int s = t / MT, m = t % MT;
    if (t + PD < STEPS && !((LCAO_ABL & 4) && t + PD >= 2 * PD)) {
      const int t2 = t + PD;
      fh[t2] = *(const t8*)(wh + (t2 / MT) * fs + 128 * (t2 % MT));
      if constexpr (PREC == PREC_BF16X3) fl[t2] = *(const t8*)(wl + (t2 / MT) * fs + 128 * (t2 % MT));
    }
    const LowpOperand<PREC>& b = in(s);
    const int tf = (LCAO_ABL & 4) ? t % (2 * PD) : t;   // ablation: fragments read once per layer call
    if constexpr (PREC == PREC_BF16X3) {
      out[m] = MFMA16BF(fh[tf], b.lo, out[m]);
      out[m] = MFMA16BF(fl[tf], b.hi, out[m]);
      out[m] = MFMA16BF(fh[tf], b.hi, out[m]);
    } else {
      out[m] = MFMA16H(fh[tf], b.hi, out[m]);
    }
    valu(t);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ===================================================================================================
// The persistent coefficient kernel (BASELINE config 3).  Round-3 structure (round 2: waves spent 55 % of their cycles in
// s_waitcnt and the two waves of a SIMD ran the same phase at the same time; profiles/r03/k2_lowp_*.txt):
//   * orbitals are walked grouped by l (a permutation built once per workgroup), so the running accumulator `cur` is one
//     register set and the per-l sums are parked when a group ends — the step body has no branch;
//   * every step is two blocks, each a mix of one layer's MFMAs with the other layer's activations:
//       Y(u): layer 2 of orbital u   woven with  SiLU(layer-1 output of u)        (k-step s + 1 under the MFMAs of k-step s)
//       X(u): layer 1 of the NEXT orbital (of the next tile after a tile's last)  woven with  cur += w SiLU(layer-2 output of u)
//   * the rows of the next orbital are packed 16-bit operands (lcao_lowp_pack_rows), requested at the top of Y(u) and
//     consumed in X(u): straight-line code, so the wait is a counted vmcnt behind younger loads only;
//   * the next tile is drawn from the queue at the top of the current one; its index words are read two orbitals before
//     the end; stores are unconditional (lanes past the last edge recompute and rewrite edge E - 1 bit for bit).
// ===================================================================================================
template <int PREC, int C, int V, int NL, int THREADS, bool GATED>
__global__ __launch_bounds__(THREADS) void k_coeff_mlp_lowp(
    const unsigned char* __restrict__ cst16, const float* __restrict__ rb, const int32_t* __restrict__ order,
    const int32_t* __restrict__ cst_row, const int32_t* __restrict__ l_of_d, const float* __restrict__ W1,
    const float* __restrict__ W2, int64_t E, int D, float* __restrict__ A, float* __restrict__ Bout,
    const float* __restrict__ gate, const int32_t* __restrict__ src_p, float* __restrict__ Ag, float* __restrict__ gram,
    unsigned* __restrict__ tile_counter LOWP_CLK_ARG) {
  static_assert(C % 32 == 0 && V % 32 == 0, "k-steps are 32 wide");
  typedef typename LowP<PREC>::T T;
  constexpr int KS1 = C / 32, KS2 = V / 32, MT = V / 16, WAVES = THREADS / 64;
  constexpr int NP = LowP<PREC>::PLANES;
  constexpr int ROW_BYTES = C * 2 * NP;   // one packed (edge, orbital) row
  extern __shared__ unsigned char lds_raw[];
  T* w1p = (T*)lds_raw;
  T* w2p = w1p + NP * V * C;
#ifdef LCAO_LOWP_CLK
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl = t_start;
#endif
  // log2(e) folded into the first layer (lcao_mfma.h: g(x') = SiLU(x) log2 e for x' = x log2 e, so the second layer's
  // input and output carry the factor and it leaves through the radial multiplier): one VALU multiply less per activation
  stage_weights_lowp<PREC, false>(w1p, W1, V, C, LCAO_LOG2E);
  stage_weights_lowp<PREC, true>(w2p, W2, V, V);
  __syncthreads();
  LOWP_STAMP(0);   // staging
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  // orbitals are walked grouped by l: lmask[l] = the orbitals d with l_of_d[d] == l (scalar registers; D <= 32)
  unsigned lmask[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) lmask[l] = 0u;
  for (int d = 0; d < D; ++d) {
    const int l = l_of_d[d];
#pragma unroll
    for (int ll = 0; ll < NL; ++ll) lmask[ll] |= (l == ll ? 1u : 0u) << d;
  }
  unsigned first_mask = 0u;   // the group the walk starts with
#pragma unroll
  for (int l = NL - 1; l >= 0; --l) first_mask = lmask[l] ? lmask[l] : first_mask;
  const int d_first = first_mask ? __builtin_ctz(first_mask) : 0;
  unsigned all_mask = 0u;
#pragma unroll
  for (int l = 0; l < NL; ++l) all_mask |= lmask[l];
  const int n_steps = __builtin_popcount(all_mask);   // = D for a valid l_of_d
  const int n_tiles = (int)((E + 15) >> 4);
  const int first_dynamic = (int)gridDim.x * WAVES;
  int tile = (int)blockIdx.x * WAVES + wave;

  // a tile's per-lane inputs as 32-bit row numbers (E < 2^31: checked by the host)
  struct Rows { int pc, e, ec, src; };
  auto tile_inputs = [&](int t) {
    Rows n;
    const int p = t * 16 + r;
    n.pc = p < (int)E ? p : (int)E - 1;
    n.e = order ? order[n.pc] : n.pc;
    n.ec = cst_row ? cst_row[n.pc] : n.e;   // gathered coefficient rows (see k_coeff_mlp)
    n.src = GATED ? src_p[n.pc] : 0;
    return n;
  };
  LowpOperand<PREC> x[KS1];
  auto load_rows = [&](const Rows& n, int d) {
    const unsigned char* row = cst16 + ((int64_t)n.ec * D + d) * ROW_BYTES;
#pragma unroll
    for (int s = 0; s < KS1; ++s) load_packed<PREC>(x[s], row, s, q);
  };
  f32x4v h[MT], o[MT];
  auto zero = [](f32x4v (&v)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) v[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
  };
  auto x_of = [&](int s) -> const LowpOperand<PREC>& { return x[s]; };

  Rows nx = {0, 0, 0, 0};
  float w_next = 0.f;
  if (tile < n_tiles) {
    nx = tile_inputs(tile);
    w_next = rb[(int64_t)nx.e * D + d_first];
    load_rows(nx, d_first);
    zero(h);
    lowp_layer_mix<PREC, KS1, MT>(w1p, V, C, r, q, x_of, h, [](int) {});
  }
  while (tile < n_tiles) {
    unsigned drawn = 0;
    if (lane == 0) drawn = atomicAdd(tile_counter, 1u);   // consumed two steps before the tile ends
    int tile_n = n_tiles;
    const Rows cur_rows = nx;
    f32x4v acc[NL][MT];
#pragma unroll
    for (int l = 0; l < NL; ++l) zero(acc[l]);
    int left = n_steps;   // steps of this tile still to run, this one included
    f32x4v gt[GATED ? MT : 1];
    // one step: orbital (group L) whose layer-1 output is in h; dn = the orbital that follows in this tile, or -1
    auto step = [&](auto lc, int dn) {
      constexpr int L = decltype(lc)::value;
      asm volatile("" ::: "memory");  // the LDS fragments are loop-invariant: keep them from being hoisted (spills)
      const float w = w_next * LCAO_INV_LOG2E;
      if (left == (n_steps >= 2 ? 2 : 1)) {
        tile_n = first_dynamic + (int)__builtin_amdgcn_readfirstlane(drawn);
        if (tile_n < n_tiles) nx = tile_inputs(tile_n);
      }
      LOWP_STAMP(1);   // step top / next tile's index words
      // ---- Y: rows + radial weight of the next step requested, then layer 2 woven with the first activation
      if (dn >= 0) {
        w_next = rb[(int64_t)cur_rows.e * D + dn];
        load_rows(cur_rows, dn);
      } else if (tile_n < n_tiles) {
        w_next = rb[(int64_t)nx.e * D + d_first];
        load_rows(nx, d_first);
      }
      // layer 2 runs k-step by k-step on hin[s] = SiLU of feature tiles 2s, 2s + 1 of layer 1; the activations of k-step
      // s + 1 are issued pair by pair between the MFMAs of k-step s
      LowpOperand<PREC> hin[KS2];
      float av[8];
      auto h_pair = [&](int s, int pr) {   // values 2 pr, 2 pr + 1 of the 8 this lane feeds to k-step s
        const int tl_ = 2 * s + (pr >> 1), j = 2 * (pr & 1);
        return f32x2v{h[tl_][j], h[tl_][j + 1]};
      };
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const f32x2v y = lowp_act2(h_pair(0, pr));
        av[2 * pr] = y[0];
        av[2 * pr + 1] = y[1];
      }
      hin[0].set(av);
      zero(o);
      f32x2v ya[4], yb[4];   // stage results of the four pairs of the k-step being prepared
      lowp_layer_mix<PREC, KS2, MT>(
          w2p, V, V, r, q, [&](int s) -> const LowpOperand<PREC>& { return hin[s]; }, o, [&](int t) {
            const int s = t / MT, m = t % MT;
            if (s + 1 >= KS2) return;
            // pair pr: stage a after step 2 pr' , b after the next, c after the one after (MT = 8; fewer steps: stages share)
            constexpr int SPP = MT >= 8 ? 2 : 1;   // steps per pair
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
              const int ta = (pr * SPP * MT) / (4 * SPP), tb = ta + 1, tc = ta + 2;
              if (MT >= 8) {
                if (m == ta) ya[pr] = lowp_act_a(h_pair(s + 1, pr));
                if (m == tb) yb[pr] = lowp_act_b(ya[pr]);
                if (m == (tc < MT ? tc : MT - 1)) {
                  const f32x2v y = lowp_act_c(h_pair(s + 1, pr), yb[pr]);
                  av[2 * pr] = y[0];
                  av[2 * pr + 1] = y[1];
                }
              } else if (pr * MT / 4 == m) {
                const f32x2v y = lowp_act2(h_pair(s + 1, pr));
                av[2 * pr] = y[0];
                av[2 * pr + 1] = y[1];
              }
            }
            if (m == MT - 1) hin[s + 1].set(av);
          });
      LOWP_STAMP(2);   // Y: layer 2 + activation 1
      // ---- X: layer 1 of the next step woven with the second activation and the accumulation into this group's sum
      zero(h);
      const f32x2v w2 = {w, w};
      constexpr int STEPS1 = KS1 * MT, NPR = 2 * MT;   // pair pr: stages after steps pr STEPS1 / NPR + {0, 1, 2}
      f32x2v za[NPR], zb[NPR];
      auto o_pair = [&](int pr) { return f32x2v{o[pr >> 1][2 * (pr & 1)], o[pr >> 1][2 * (pr & 1) + 1]}; };
      auto finish = [&](int pr, f32x2v y) {
        const int m = pr >> 1, j = 2 * (pr & 1);
        const f32x2v c2 = f32x2v{acc[L][m][j], acc[L][m][j + 1]} + w2 * y;
        acc[L][m][j] = c2[0];
        acc[L][m][j + 1] = c2[1];
      };
      lowp_layer_mix<PREC, KS1, MT>(w1p, V, C, r, q, x_of, h, [&](int t) {
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr) {
          const int ta = pr * STEPS1 / NPR;
          if (STEPS1 >= 2 * NPR) {
            const int tc = ta + 2 < STEPS1 ? ta + 2 : STEPS1 - 1;
            if (t == ta) za[pr] = lowp_act_a(o_pair(pr));
            if (t == ta + 1) zb[pr] = lowp_act_b(za[pr]);
            if (t == tc) finish(pr, lowp_act_c(o_pair(pr), zb[pr]));
          } else if (t == ta) {
            finish(pr, lowp_act2(o_pair(pr)));
          }
        }
      });
      LOWP_STAMP(3);   // X: layer 1 + activation 2
      if (GATED && dn < 0) {
        // the centre's gate row for the epilogue, requested here (another basic block) so that B and the Gram matrix are
        // computed while it is in flight: within the epilogue hipcc sinks the loads to their first use
#pragma unroll
        for (int m = 0; m < MT; ++m) gt[m] = *(const f32x4v*)(gate + (int64_t)cur_rows.src * V + 16 * m + 4 * q);
      }
      --left;
    };
    auto group = [&](auto lc) {
      constexpr int L = decltype(lc)::value;
      unsigned later = 0u;   // the next non-empty group after L
#pragma unroll
      for (int l = NL - 1; l > L; --l) later = lmask[l] ? lmask[l] : later;
      unsigned rem = lmask[L];
      while (rem) {
        rem &= rem - 1;   // the step's own orbital: its rows and weight were requested one step earlier
        const int dn = rem ? __builtin_ctz(rem) : (later ? __builtin_ctz(later) : -1);
        step(lc, dn);
      }
    };
    group(std::integral_constant<int, 0>{});
    if constexpr (NL > 1) group(std::integral_constant<int, 1>{});
    if constexpr (NL > 2) group(std::integral_constant<int, 2>{});
    if constexpr (NL > 3) group(std::integral_constant<int, 3>{});
    // epilogue: B, Gram of the ungated A, gated copy or raw A (as k_coeff_mlp)
    if (!((LCAO_ABL & 32) && acc[0][0][0] != 12345.f)) {
      float* Bp = Bout + (int64_t)cur_rows.pc * V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4v bsum = acc[0][m];
#pragma unroll
        for (int l = 1; l < NL; ++l) bsum += acc[l][m];
        *(f32x4v*)(Bp + 16 * m) = bsum;
      }
    }
    LOWP_STAMP(7);   // gate request + B
    if (!((LCAO_ABL & 32) && acc[0][0][0] != 12345.f)) {
      float* Ap = (GATED ? Ag : A) + (int64_t)cur_rows.pc * NL * V + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int l = 0; l < NL; ++l) *(f32x4v*)(Ap + l * V + 16 * m) = GATED ? acc[l][m] * gt[m] : acc[l][m];
    }
    LOWP_STAMP(6);   // A stores
    if (GATED) {
      // triangular factor of the ungated per-l vectors (lcao_common.h: lcao_edge_factor), in place on the accumulators
      // after their last use above
      float g[12], g10[10];
      float(&wv)[NL][4 * MT] = reinterpret_cast<float(&)[NL][4 * MT]>(acc);
      lcao_edge_factor<NL, 4 * MT>(wv, g10, [](float t) {
        t += __shfl_xor(t, 16);
        return t + __shfl_xor(t, 32);
      });
#pragma unroll
      for (int i = 0; i < 10; ++i) g[i] = g10[i];
      g[10] = g[9];   // lane group q stores entries q, q + 4, q + 8 (entries 10, 11 repeat entry 9): three stores, no branch
      g[11] = g[9];
      float* gp = gram + (int64_t)cur_rows.pc * 10;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float v = q == 0 ? g[4 * k] : (q == 1 ? g[4 * k + 1] : (q == 2 ? g[4 * k + 2] : g[4 * k + 3]));
        const int i = 4 * k + q;
        gp[i < 10 ? i : 9] = v;
      }
    }
    LOWP_STAMP(8);   // edge factor
    tile = tile_n;
  }
#ifdef LCAO_LOWP_CLK
  if (clk && lane == 0) {
    unsigned long long* o_ = clk + 13 * ((size_t)blockIdx.x * WAVES + wave);
#pragma unroll
    for (int i = 0; i < 10; ++i) o_[i] = ph[i];
    o_[10] = __builtin_amdgcn_s_memtime() - t_start;
    o_[11] = __builtin_amdgcn_s_memrealtime() - r_start;
    o_[12] = r_start;
  }
#endif
  if (lane == 0) lcao_tile_queue_leave(tile_counter, gridDim.x * (THREADS / 64));   // the last wave re-arms the queue
}

// 512 threads (two waves per SIMD share the LDS weights) while the accumulators fit 256 VGPRs (n_l <= 3 at V = 128)
template <int PREC, int C, int V, int NL>
static int launch_coeff_mlp_lowp(const void* cst16, const float* rb, const int32_t* order, const int32_t* cst_row,
                                 const int32_t* l_of_d, const float* W1, const float* W2, int64_t E, int D, float* A,
                                 float* Bout, const float* gate, const int32_t* src_p, float* Ag, float* gram,
                                 unsigned* counter, hipStream_t st) {
  constexpr int THREADS = (NL * V > 384) ? 256 : LCAO_LOWP_THREADS;
  constexpr int NP = LowP<PREC>::PLANES;
  const size_t lds = (size_t(NP) * V * C + size_t(NP) * V * V) * 2;
  constexpr int WAVES = THREADS / 64;
  int64_t tiles = (E + 15) / 16;
  int64_t groups = (tiles + WAVES - 1) / WAVES;
  int64_t cap = (int64_t)lcao_num_cu() * ((lds > 80 * 1024) ? 1 : 2);
  if (groups > cap) groups = cap;
  const unsigned char* c16 = (const unsigned char*)cst16;
  if (Ag) {
    static bool done = false;
    int rc = set_lds_attr(k_coeff_mlp_lowp<PREC, C, V, NL, THREADS, true>, lds, done);
    if (rc != LCAO_OK) return rc;
    hipLaunchKernelGGL((k_coeff_mlp_lowp<PREC, C, V, NL, THREADS, true>), dim3((unsigned)groups), dim3(THREADS), lds, st, c16,
                       rb, order, cst_row, l_of_d, W1, W2, E, D, A, Bout, gate, src_p, Ag, gram, counter LOWP_CLK_PASS);
  } else {
    static bool done = false;
    int rc = set_lds_attr(k_coeff_mlp_lowp<PREC, C, V, NL, THREADS, false>, lds, done);
    if (rc != LCAO_OK) return rc;
    hipLaunchKernelGGL((k_coeff_mlp_lowp<PREC, C, V, NL, THREADS, false>), dim3((unsigned)groups), dim3(THREADS), lds, st, c16,
                       rb, order, cst_row, l_of_d, W1, W2, E, D, A, Bout, gate, src_p, Ag, gram, counter LOWP_CLK_PASS);
  }
  return lcao_launch_status();
}

template <int PREC>
static int coeff_mlp_contract_lowp(const void* cst16, const float* rb, const int32_t* order, const int32_t* cst_row,
                                   const int32_t* l_of_d, const float* W1, const float* W2, const float* gate,
                                   const int32_t* src_p, int64_t n_edges, int n_orb, int C, int V, int n_l, float* A, float* B,
                                   float* Ag, float* gram, void* workspace, size_t ws_bytes, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  if (n_orb <= 0 || C <= 0 || V <= 0 || n_l < 1 || n_l > LCAO_MAX_L + 1) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(cst16); LCAO_CHECK_PTR(rb); LCAO_CHECK_PTR(l_of_d); LCAO_CHECK_PTR(W1); LCAO_CHECK_PTR(W2);
  LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(workspace);
  if (Ag) { LCAO_CHECK_PTR(gate); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(gram); } else { LCAO_CHECK_PTR(A); }
  if (ws_bytes < 2 * sizeof(unsigned)) return LCAO_ERR_INVALID;   // {drawn, left} of the tile queue
  if (n_orb > 32) return LCAO_ERR_UNSUPPORTED;                    // the orbitals of each l are a 32-bit mask
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctr = (unsigned*)workspace;
#define LP_CASE(C_, V_, NL_)                                                                                       \
  if (C == C_ && V == V_ && n_l == NL_)                                                                            \
    return launch_coeff_mlp_lowp<PREC, C_, V_, NL_>(cst16, rb, order, cst_row, l_of_d, W1, W2, n_edges, n_orb, A, B, gate, \
                                                    src_p, Ag, Ag ? gram : nullptr, ctr, st);
#define LP_CV(C_, V_) LP_CASE(C_, V_, 1) LP_CASE(C_, V_, 2) LP_CASE(C_, V_, 3) LP_CASE(C_, V_, 4)
  LP_CV(128, 128) LP_CV(64, 64) LP_CV(32, 32)
#undef LP_CV
#undef LP_CASE
  return LCAO_ERR_UNSUPPORTED;
}

extern "C" int lcao_coeff_mlp_contract_bf16x3(const void* cst16, const float* rb, const int32_t* order,
                                              const int32_t* cst_row, const int32_t* l_of_d, const float* W1, const float* W2,
                                              const float* gate, const int32_t* src_p, int64_t n_edges, int n_orb, int C,
                                              int V, int n_l, float* A, float* B, float* Ag, float* gram,
                                              void* workspace, size_t ws_bytes, void* stream) {
  return coeff_mlp_contract_lowp<PREC_BF16X3>(cst16, rb, order, cst_row, l_of_d, W1, W2, gate, src_p, n_edges, n_orb, C, V, n_l,
                                              A, B, Ag, gram, workspace, ws_bytes, stream);
}

extern "C" int lcao_coeff_mlp_contract_f16(const void* cst16, const float* rb, const int32_t* order, const int32_t* cst_row,
                                           const int32_t* l_of_d, const float* W1, const float* W2, const float* gate,
                                           const int32_t* src_p, int64_t n_edges, int n_orb, int C, int V, int n_l, float* A,
                                           float* B, float* Ag, float* gram, void* workspace, size_t ws_bytes, void* stream) {
  return coeff_mlp_contract_lowp<PREC_F16>(cst16, rb, order, cst_row, l_of_d, W1, W2, gate, src_p, n_edges, n_orb, C, V, n_l, A,
                                           B, Ag, gram, workspace, ws_bytes, stream);
}

// ===================================================================================================
// low-precision forms of the two-body kernels (same operand treatment; `precision="bf16x3"` / `"f16"`)
// ===================================================================================================
// THREADS = 512 where the weight planes leave room for one workgroup per CU only (bf16x3 at V = 128: 128 KB): two waves per
// SIMD then share them; 256 otherwise (two workgroups per CU).
template <int PREC, int V, int THREADS>
__global__ __launch_bounds__(THREADS) void k_edge_lcao_weight_lowp(const float* __restrict__ tbw, const float* __restrict__ B,
                                                               const float* __restrict__ W3, const float* __restrict__ Wb,
                                                               int64_t E, float* __restrict__ bw,
                                                               const int32_t* __restrict__ skip_if /* nullable: return when it reads non-zero */) {
  typedef typename LowP<PREC>::T T;
  constexpr int MT = V / 16, KS = V / 32, NP = LowP<PREC>::PLANES;
  extern __shared__ unsigned char lds_raw[];
  if (skip_if && *skip_if != 0) return;
  T* w3p = (T*)lds_raw;
  T* wbp = w3p + NP * V * V;
  stage_weights_lowp<PREC, false>(w3p, W3, V, V);
  stage_weights_lowp<PREC, true>(wbp, Wb, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (E + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * (THREADS / 64) + wave; tile < n_tiles; tile += (int64_t)gridDim.x * (THREADS / 64)) {
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
    LowpOperand<PREC> xop[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xop[s].set(xin[s]);
    lowp_layer<PREC, KS, MT>(w3p, V, V, r, q, xop, ft);
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
    LowpOperand<PREC> lop[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) lop[s].set(lin[s]);
    lowp_layer<PREC, KS, MT>(wbp, V, V, r, q, lop, o);
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
  constexpr int MT = V / 16, KS = V / 32;
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
    LowpOperand<PREC> hop[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) hop[s].set(hin[s]);
    lowp_layer<PREC, KS, MT>(wp, V, V, r, q, hop, o);
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
  constexpr int MT = V / 16, KS = V / 32;
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
        LowpOperand<PREC> hop[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) hop[s].set(hin[s]);
    lowp_layer<PREC, KS, MT>(wp, V, V, r, q, hop, o);
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

// ---------------------------------------------------------------------------------------------------------------------
// lcao_edge_two_body_agg_f16 — BOTH two-body kernels of a layer in one launch (lcaonet.py:192-213), fp16 only: W3, Wb and Wn2
// as fp16 fragments are 96 KB of LDS (their fp32 or bf16-pair forms do not fit 160 KB together).  Node-column form as
// k_edge_node_message_cols_lowp — step t takes the t-th out-edge of 16 consecutive centre nodes — with the basis weight
// of that edge computed in place of loading it:
//     ft = W3 . tbw ;  u = B (1 + ft) ;  lw = u / max(|u|, 1e-12) ;  bw = Wb . lw          (k_edge_lcao_weight_lowp)
//     acc[s] += bw * SiLU(Wn2 . SiLU(Ps[s] + Pt[t]) + bn2)                                 (k_edge_node_message_cols_lowp)
// bw (E, V) is neither written nor re-read (1 KB/edge/layer) and one launch per layer goes away.  96 MFMAs per step;
// 512 threads = two waves per SIMD share the weights.
// ---------------------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(512) void k_edge_two_body_cols_f16(
    const float* __restrict__ tbw, const float* __restrict__ B, const float* __restrict__ Ps, const float* __restrict__ Pt,
    const int32_t* __restrict__ dst_p, const int32_t* __restrict__ out_ptr, const float* __restrict__ W3,
    const float* __restrict__ Wb, const float* __restrict__ Wn2, const float* __restrict__ bn2, int n_nodes, int n_edges,
    const int32_t* __restrict__ hint, float* __restrict__ agg) {
  constexpr int PREC = PREC_F16;
  typedef _Float16 T;
  constexpr int MT = V / 16, KS = V / 32, WAVES = 8;
  extern __shared__ unsigned char lds_raw[];
  if (hint && *hint == 0) return;
  const bool by_degree = hint && *hint == 2;
  T* w3p = (T*)lds_raw;
  T* wbp = w3p + V * V;
  T* wnp = wbp + V * V;
  stage_weights_lowp<PREC, false>(w3p, W3, V, V);
  stage_weights_lowp<PREC, true>(wbp, Wb, V, V);
  stage_weights_lowp<PREC, false>(wnp, Wn2, V, V);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n_groups = (n_nodes + 15) >> 4;
  for (int g = blockIdx.x * WAVES + wave; g < n_groups; g += gridDim.x * WAVES) {
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
      // the centre's own row stays in registers for the group; the target row of a step is requested one step ahead and
      // only ADDED to it when the step is over (an add right at the load would make the wave wait for the gather at once)
      f32x4v psr[KS][2];
      {
        const float* pr = Ps + (int64_t)s * V + 8 * q;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          psr[k][0] = *(const f32x4v*)(pr + 32 * k);
          psr[k][1] = *(const f32x4v*)(pr + 32 * k + 4);
        }
      }
      float hin[KS][8];
      {
        const float* pt = Pt + (int64_t)dst_p[edge_at(0)] * V + 8 * q;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const f32x4v a0 = *(const f32x4v*)(pt + 32 * k) + psr[k][0];
          const f32x4v a1 = *(const f32x4v*)(pt + 32 * k + 4) + psr[k][1];
#pragma unroll
          for (int j = 0; j < 4; ++j) { hin[k][j] = lcao_silu(a0[j]); hin[k][4 + j] = lcao_silu(a1[j]); }
        }
      }
      int d_nxt = dst_p[edge_at(1)];
      f32x4v xt[KS][2];     // the three-body row of a step is requested one step ahead (it streams from HBM)
      {
        const int64_t p0w = edge_at(0);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          xt[k][0] = *(const f32x4v*)(tbw + p0w * V + 32 * k + 8 * q);
          xt[k][1] = *(const f32x4v*)(tbw + p0w * V + 32 * k + 8 * q + 4);
        }
      }
      for (int t = 0; t < Tn; ++t) {
        asm volatile("" ::: "memory");
        const float live = t < deg ? 1.0f : 0.0f;
        // this step's B row (needed after two GEMMs) and the NEXT step's rows: requested before the first GEMM
        f32x4v bb[MT], ntv[KS][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) bb[m] = *(const f32x4v*)(B + edge_at(t) * (int64_t)V + 16 * m + 4 * q);
        LowpOperand<PREC> xop[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float v[8] = {xt[k][0][0], xt[k][0][1], xt[k][0][2], xt[k][0][3], xt[k][1][0], xt[k][1][1], xt[k][1][2], xt[k][1][3]};
          xop[k].set(v);
        }
        {
          const int64_t pn = edge_at(t + 1);
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            xt[k][0] = *(const f32x4v*)(tbw + pn * V + 32 * k + 8 * q);
            xt[k][1] = *(const f32x4v*)(tbw + pn * V + 32 * k + 8 * q + 4);
          }
        }
        {
          const float* pt = Pt + (int64_t)d_nxt * V + 8 * q;
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            ntv[k][0] = *(const f32x4v*)(pt + 32 * k);
            ntv[k][1] = *(const f32x4v*)(pt + 32 * k + 4);
          }
        }
        d_nxt = dst_p[edge_at(t + 2)];
        // ---- message factor: SiLU(Wn2 . h1 + bn2), zero for a padded step
        f32x4v o[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) o[m] = *(const f32x4v*)(bn2 + 16 * m + 4 * q);
        {
          LowpOperand<PREC> hop[KS];
#pragma unroll
          for (int k = 0; k < KS; ++k) hop[k].set(hin[k]);
          lowp_layer<PREC, KS, MT>(wnp, V, V, r, q, hop, o);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[m][j] = live * lcao_silu(o[m][j]);
        // ---- basis weight of this edge: f_three, injection, F.normalize, basis_weight
        f32x4v ft[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) ft[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
        lowp_layer<PREC, KS, MT>(w3p, V, V, r, q, xop, ft);
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
        f32x4v wv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) wv[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
        {
          LowpOperand<PREC> lop[KS];
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = ft[2 * k][j] * inv;
              v[4 + j] = ft[2 * k + 1][j] * inv;
            }
            lop[k].set(v);
          }
          lowp_layer<PREC, KS, MT>(wbp, V, V, r, q, lop, wv);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] += wv[m] * o[m];
        // first layer of the next step from the row requested above
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hin[k][j] = lcao_silu(ntv[k][0][j] + psr[k][0][j]);
            hin[k][4 + j] = lcao_silu(ntv[k][1][j] + psr[k][1][j]);
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
                                 float* bw, void* stream, const int32_t* skip_if = nullptr) {
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0) return LCAO_ERR_INVALID;
  if (n_edges == 0) return LCAO_OK;
  LCAO_CHECK_PTR(tbw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(W3); LCAO_CHECK_PTR(Wb); LCAO_CHECK_PTR(bw);
  hipStream_t st = (hipStream_t)stream;
  size_t lds = size_t(2) * LowP<PREC>::PLANES * V * V * 2;
#define LWL_CASE(V_)                                                                                                \
  if (V == V_) {                                                                                                    \
    constexpr int TH = (size_t(2) * LowP<PREC>::PLANES * V_ * V_ * 2 > 80 * 1024) ? 512 : 256;                      \
    static bool done = false;                                                                                       \
    int rc = set_lds_attr(k_edge_lcao_weight_lowp<PREC, V_, TH>, lds, done);                                        \
    if (rc != LCAO_OK) return rc;                                                                                   \
    hipLaunchKernelGGL((k_edge_lcao_weight_lowp<PREC, V_, TH>), dim3(persistent_groups(n_edges, lds, TH / 64)), dim3(TH), lds, \
                       st, tbw, B, W3, Wb, n_edges, bw, skip_if);                                                   \
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
  size_t lds = size_t(LowP<PREC>::PLANES) * V * V * 2;
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
    return lcao_zero_async(agg, sizeof(float) * (size_t)n_nodes * V, st);
  }
  LCAO_CHECK_PTR(msg);
  // uneven degrees: message rows, then the per-node sum (both return at once when the hint reads 1)
  int rc = edge_node_message_lowp<PREC>(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg, stream, column_hint);
  if (rc != LCAO_OK) return rc;
  hipLaunchKernelGGL(k_rows_segment_sum, dim3(lcao_blocks(n_nodes, 4, 65536)), dim3(256), 0, st, msg, out_ptr, (int)n_nodes, V,
                     column_hint, agg);
  if (!column_hint) return lcao_launch_status();
  size_t lds = size_t(LowP<PREC>::PLANES) * V * V * 2;
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

// The two-body half of a layer for precision "f16" (lcaonet.py:192-213): with a column hint reading 1 or 2 ONE fused launch
// (k_edge_two_body_cols_f16); with 0 (uneven degrees) or without a hint the three separate kernels, through the bw / msg
// scratch rows.  Every launch decides on the device from the hint word (no host sync).
extern "C" int lcao_edge_two_body_agg_f16(const float* tbw, const float* B, const float* W3, const float* Wb, const float* Ps,
                                          const float* Pt, const int32_t* src_p, const int32_t* dst_p, const int32_t* out_ptr,
                                          const float* Wn2, const float* bn2, int64_t n_nodes, int64_t n_edges, int V,
                                          const int32_t* column_hint, float* bw_scratch, float* msg_scratch, float* agg,
                                          void* stream) {
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_SIZE(n_edges);
  if (V <= 0 || V % 4) return LCAO_ERR_INVALID;
  if (n_nodes == 0) return LCAO_OK;
  LCAO_CHECK_PTR(agg); LCAO_CHECK_PTR(out_ptr);
  hipStream_t st = (hipStream_t)stream;
  if (n_edges == 0) return lcao_zero_async(agg, sizeof(float) * (size_t)n_nodes * V, st);
  LCAO_CHECK_PTR(tbw); LCAO_CHECK_PTR(B); LCAO_CHECK_PTR(W3); LCAO_CHECK_PTR(Wb); LCAO_CHECK_PTR(Ps); LCAO_CHECK_PTR(Pt);
  LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p); LCAO_CHECK_PTR(Wn2); LCAO_CHECK_PTR(bn2); LCAO_CHECK_PTR(bw_scratch);
  LCAO_CHECK_PTR(msg_scratch);
  if (V != 128 && V != 64 && V != 32) return LCAO_ERR_UNSUPPORTED;
  // uneven degrees (hint word 0) or no hint: basis weights, message rows, per-node sums; each returns at once otherwise
  int rc = edge_lcao_weight_lowp<PREC_F16>(tbw, B, W3, Wb, n_edges, V, bw_scratch, stream, column_hint);
  if (rc != LCAO_OK) return rc;
  rc = edge_node_message_lowp<PREC_F16>(bw_scratch, Ps, Pt, src_p, dst_p, Wn2, bn2, n_edges, V, msg_scratch, stream, column_hint);
  if (rc != LCAO_OK) return rc;
  hipLaunchKernelGGL(k_rows_segment_sum, dim3(lcao_blocks(n_nodes, 4, 65536)), dim3(256), 0, st, msg_scratch, out_ptr,
                     (int)n_nodes, V, column_hint, agg);
  if (!column_hint) return lcao_launch_status();
  const size_t lds = size_t(3) * V * V * 2;
  int64_t cgroups = (((n_nodes + 15) >> 4) + 7) / 8, cap = (int64_t)lcao_num_cu() * (lds > 80 * 1024 ? 1 : 2);
  if (cgroups > cap) cgroups = cap;
#define TB_CASE(V_)                                                                                                   \
  if (V == V_) {                                                                                                      \
    static bool done = false;                                                                                         \
    rc = set_lds_attr(k_edge_two_body_cols_f16<V_>, lds, done);                                                       \
    if (rc != LCAO_OK) return rc;                                                                                     \
    hipLaunchKernelGGL((k_edge_two_body_cols_f16<V_>), dim3((unsigned)cgroups), dim3(512), lds, st, tbw, B, Ps, Pt, dst_p, \
                       out_ptr, W3, Wb, Wn2, bn2, (int)n_nodes, (int)n_edges, column_hint, agg);                      \
    return lcao_launch_status();                                                                                      \
  }
  TB_CASE(128) TB_CASE(64) TB_CASE(32)
#undef TB_CASE
  return LCAO_ERR_UNSUPPORTED;
}
