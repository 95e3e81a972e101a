// lcao_mfma.h — the shared tile formulation of the matrix-core kernels (see the header comment of lcao_mfma.hip):
// weight tile = MFMA A operand from LDS, 16 data rows = B operand from global memory, accumulator layout of one layer
// = B-operand layout of the next.
#pragma once
#include <cstdlib>

#include "lcao_common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Diagnostic ablation builds (tools/k2_ablate.py; never defined in the shipped library):
//   LCAO_ABL & 1: activations replaced by identity     & 2: coefficient rows loaded once per tile, not per orbital
//   LCAO_ABL & 4: weight fragments read once per feature-tile pair instead of once per k chunk (1/KT of the LDS reads)
//   LCAO_ABL & 8: WITHOUT the s_setprio(1) around each layer's MFMA stream     & 16: static s_setprio(1) for waves 4-7
//   LCAO_ABL & 128 / & 256 / & 512 (k_triplet_bwd_mfma): GEMM 1 cut to one k chunk / no scalar chain / GEMM 2 cut to one tile
//   LCAO_ABL & 64 (k_triplet_gemm, tools/k3_ablate.py): one MFMA per in-edge step
//   LCAO_ABL & 1024 / & 2048 / & 4096 (k_edge_node_message_agg, tools/k4b_ablate.py): no segmented reduction / identity
//   activations / target rows gathered from the centre's own row (no random gather)
#ifndef LCAO_ABL
#define LCAO_ABL 0
#endif
#if LCAO_ABL & 1
#define K2_ACT(x) (x)
#define K2_ACT4(x) (x)
#else
#define K2_ACT(x) lcao_silu_log2e(x)
#define K2_ACT4(x) lcao_silu_log2e4(x)
#endif

#define LCAO_LOG2E 1.44269504088896340736f
#define LCAO_INV_LOG2E 0.69314718055994530942f

// g(x') = x' / (1 + 2^(-x')) = SiLU(x) log2 e for x' = x log2 e: v_exp_f32 (negated source), v_add, v_rcp_f32, v_mul
__device__ __forceinline__ float lcao_silu_log2e(float xs) {
  return xs * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-xs));
}

// the same on four accumulator values with the add and the multiply as vector ops (v_pk_add_f32 / v_pk_mul_f32: two
// lanes' worth per instruction): the fp32 MFMA stream and VALU instructions share the SIMD's vector issue
// (profiles/r02/mfma_valu_overlap.txt: t(both) = t(mfma) + t(valu)), so every VALU instruction saved is kernel time.
__device__ __forceinline__ f32x4v lcao_silu_log2e4(f32x4v xs) {
  f32x4v e;
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(-xs[j]);
  e = e + 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
  return xs * e;
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Number of CUs the persistent kernels size their grids for.  LCAO_CU_RESERVE=k leaves k CUs out: the persistent
// coefficient kernel fills every SIMD's register file (2 waves x 256 VGPRs), so a collective's copy kernels launched on
// another stream (the halo exchange of a sharded graph, lcaonet_amd/shard.py) find no slot until it ends unless some CUs
// are kept free.  Default 0 (single-GPU runs); untested on multi-GPU hardware.
static inline int lcao_num_cu() {
  static int g_num_cu = 0;
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cu = prop.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
    const char* e = getenv("LCAO_CU_RESERVE");
    const int reserve = e ? atoi(e) : 0;
    if (reserve > 0 && reserve < g_num_cu) g_num_cu -= reserve;
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

// stage_weights with every element multiplied by `scale` (constants folded into a layer)
__device__ __forceinline__ void stage_weights_scaled(float* __restrict__ dst, const float* __restrict__ W, int rows, int K,
                                                     int ldw, float scale) {
  const int chunks = K >> 2;
  for (int i = threadIdx.x; i < rows * chunks; i += blockDim.x) {
    int rr = i / chunks, cc = i - rr * chunks;
    *(f32x4v*)(dst + rr * (K + 4) + 4 * cc) = *(const f32x4v*)(W + (int64_t)rr * ldw + 4 * cc) * scale;
  }
}

// one dense layer on a 16-row tile held in B-operand layout:  out[m] (+)= W[16m.., :] . in
//   KT = K/16 input chunks (f32x4 each), MT = N/16 output tiles.  Two output tiles are interleaved so that
//   consecutive MFMAs never wait on their own accumulator (40-cycle dependent latency vs 32-cycle issue).
template <int KT, int MT>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ wl, int ldw /*K+4*/, int r, int q,
                                           const f32x4v (&in)[KT], f32x4v (&out)[MT]) {
  static_assert(MT % 2 == 0, "feature tiles are processed in pairs");
  // Flattened (tile pair, k chunk) sequence with the weight fragments of step t+1 requested from LDS before the
  // 8 MFMAs of step t are issued: ~256 cycles of matrix work cover the ds_read_b128 latency (reading a fragment
  // right before its first MFMA stalled the pipe for ~25 % of the kernel).
  constexpr int STEPS = (MT / 2) * KT;
  const float* w0 = wl + r * ldw + 4 * q;
  f32x4v n0 = *(const f32x4v*)(w0);
  f32x4v n1 = *(const f32x4v*)(w0 + 16 * ldw);
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the prologue reads form their own group (see below)
#pragma unroll
  for (int t = 0; t < STEPS; ++t) {
    const int m = 2 * (t / KT), a = t % KT;
    const f32x4v f0 = n0, f1 = n1;
    if (t + 1 < STEPS) {
      const int m2 = 2 * ((t + 1) / KT), a2 = (t + 1) % KT;
      if (!(LCAO_ABL & 4) || a2 == 0) {  // ablation: one fragment pair per tile pair (1/KT of the LDS reads)
        n0 = *(const f32x4v*)(w0 + (16 * m2) * ldw + 16 * a2);
        n1 = *(const f32x4v*)(w0 + (16 * m2 + 16) * ldw + 16 * a2);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      out[m] = MFMA16(f0[b], in[a][b], out[m]);
      out[m + 1] = MFMA16(f1[b], in[a][b], out[m + 1]);
    }
    // pin the issue order: the two ds_read_b128 of step t+1 first, then this step's 8 MFMAs (hipcc otherwise sinks
    // the reads next to their first use and waits on them with one MFMA of cover)
    if (t + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
}

// Tile queue of the persistent coefficient kernels: q[0] = tiles drawn, q[1] = waves that have left.  Both are zero
// before the first launch (the caller allocates the workspace zeroed, once); every wave calls this after its last draw and
// the last of `total_waves` re-arms the queue, so the words are zero again when the kernel ends.  Launches that share a
// workspace must be ordered (same stream).
__device__ __forceinline__ void lcao_tile_queue_leave(unsigned* q, unsigned total_waves) {
#ifdef LCAO_DIAG_MEMSET_REARM   // diagnostic build (tools/probes/tile_queue_suspects.py): round 2's memset node instead, see launch
  return;
#endif
  if (atomicAdd(q + 1, 1u) == total_waves - 1) {
    atomicExch(q, 0u);
    atomicExch(q + 1, 0u);
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

static unsigned persistent_groups(int64_t n_rows, size_t lds, int waves = 4) {
  int64_t tiles = (n_rows + 15) / 16;
  int64_t groups = (tiles + waves - 1) / waves;
  int64_t per_cu = (lds > 80 * 1024) ? 1 : ((lds > 53 * 1024) ? 2 : 3);
  int64_t cap = (int64_t)lcao_num_cu() * per_cu;
  return unsigned(groups > cap ? cap : groups);
}

