// lcao_csr.hip — on-device construction of the per-forward index structures: replaces the
// reference's get_triplets (lcaonet/model/lcaonet.py:439-486, torch_sparse COO sort + row-select, host syncs, three
// T-sized int64 lists) by two stable radix sorts over E int32 keys and two lower-bound passes.  No host sync.
//
//   order   (E)   p -> caller's edge id, edges stably sorted by centre node idx_s  ("p-space")
//   src_p, dst_p (E)  centre / neighbour node of edge p
//   out_ptr (N+1) edges with centre s are p in [out_ptr[s], out_ptr[s+1])
//   in_pos  (E)   p-indices of the edges k->s grouped by s, ordered by (k, caller's edge id)   [stable sort of p by dst]
//   in_ptr  (N+1) in-edge list of s is in_pos[in_ptr[s] : in_ptr[s+1]]
// The triplets of edge p=(s->t) are exactly {(p, q) : q in in-list(s), q != p} (lcaonet.py:470 drops only q == p).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "lcao_common.h"

// int64 -> int32 with a range check: an endpoint outside [0, N) (the reference raises an IndexError for it) sets bit 0 of
// *err and is clamped, so that no later kernel reads out of bounds; the caller turns the flag into an exception.
__global__ void k_csr_convert(const int64_t* __restrict__ idx_s, const int64_t* __restrict__ idx_t, int64_t E, int64_t N,
                              int32_t* __restrict__ src_o, int32_t* __restrict__ dst_o, int32_t* __restrict__ iota,
                              int32_t* __restrict__ err) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = idx_s[e], t = idx_t[e];
  if (s < 0 || s >= N || t < 0 || t >= N) {
    if (err) atomicOr(err, LCAO_FLAG_EDGE_INDEX);
    s = min(max(s, (int64_t)0), N - 1);
    t = min(max(t, (int64_t)0), N - 1);
  }
  src_o[e] = (int32_t)s;
  dst_o[e] = (int32_t)t;
  iota[e] = (int32_t)e;
}

// out[i] = clamp(v[i], lo, hi); a value outside [lo, hi] sets `bit` of *err (atomic numbers: embed.py:41 indexes z - 1)
__global__ void k_checked_index(const int64_t* __restrict__ v, int64_t n, int64_t lo, int64_t hi, int bit,
                                int64_t* __restrict__ out, int32_t* __restrict__ err) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t x = v[i];
  if (x < lo || x > hi) {
    if (err) atomicOr(err, bit);
    x = min(max(x, lo), hi);
  }
  out[i] = x;
}

extern "C" int lcao_checked_index(const int64_t* v, int64_t n, int64_t lo, int64_t hi, int bit, int64_t* out, int32_t* err,
                                  void* stream) {
  LCAO_CHECK_SIZE(n);
  if (lo > hi || bit <= 0) return LCAO_ERR_INVALID;
  if (n == 0) return LCAO_OK;
  LCAO_CHECK_PTR(v); LCAO_CHECK_PTR(out);
  hipLaunchKernelGGL(k_checked_index, dim3(lcao_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, v, n, lo, hi, bit, out,
                     err);
  return lcao_launch_status();
}

__global__ void k_csr_gather(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int64_t E,
                             int32_t* __restrict__ out) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e < E) out[e] = src[idx[e]];
}

// ptr[i] = first position whose (sorted) key is >= i, for i = 0..N
__global__ void k_csr_lower_bound(const int32_t* __restrict__ sorted, int64_t E, int64_t N, int32_t* __restrict__ ptr) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > N) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < (int32_t)i) lo = mid + 1;
    else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

static int key_bits(int64_t n_nodes) {
  int b = 1;
  while ((int64_t(1) << b) < n_nodes) ++b;
  return b;
}

static size_t sort_temp_bytes(int64_t E, int bits) {
  size_t bytes = 0;
  int32_t* np = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, np, np, np, np, (size_t)E, 0, bits, (hipStream_t)0);
  return (bytes + 255) & ~size_t(255);
}

extern "C" size_t lcao_build_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes) {
  if (n_edges <= 0) return 256;
  size_t e4 = ((size_t)n_edges * 4 + 255) & ~size_t(255);
  return 2 * e4 + sort_temp_bytes(n_edges, key_bits(n_nodes));
}

extern "C" int lcao_build_csr(const int64_t* idx_s, const int64_t* idx_t, int64_t n_edges, int64_t n_nodes,
                              int32_t* order, int32_t* src_p, int32_t* dst_p, int32_t* out_ptr, int32_t* in_ptr,
                              int32_t* in_pos, int32_t* src_o, int32_t* dst_o, void* workspace, size_t ws_bytes,
                              int32_t* err_flag, void* stream) {
  LCAO_CHECK_SIZE(n_edges);
  LCAO_CHECK_SIZE(n_nodes);
  LCAO_CHECK_PTR(out_ptr); LCAO_CHECK_PTR(in_ptr);
  hipStream_t st = (hipStream_t)stream;
  if (n_edges == 0) {
    const int e1 = lcao_zero_async(out_ptr, (size_t)(n_nodes + 1) * 4, st);
    const int e2 = lcao_zero_async(in_ptr, (size_t)(n_nodes + 1) * 4, st);
    return e1 != LCAO_OK ? e1 : e2;
  }
  LCAO_CHECK_PTR(idx_s); LCAO_CHECK_PTR(idx_t); LCAO_CHECK_PTR(order); LCAO_CHECK_PTR(src_p); LCAO_CHECK_PTR(dst_p);
  LCAO_CHECK_PTR(in_pos); LCAO_CHECK_PTR(src_o); LCAO_CHECK_PTR(dst_o); LCAO_CHECK_PTR(workspace);
  if (ws_bytes < lcao_build_csr_workspace_bytes(n_edges, n_nodes)) return LCAO_ERR_INVALID;
  const int bits = key_bits(n_nodes);
  const size_t e4 = ((size_t)n_edges * 4 + 255) & ~size_t(255);
  int32_t* iota = (int32_t*)workspace;
  int32_t* dst_sorted = (int32_t*)((char*)workspace + e4);
  void* temp = (char*)workspace + 2 * e4;
  size_t temp_bytes = ws_bytes - 2 * e4;
  const unsigned nb = lcao_blocks(n_edges, 256);
  hipLaunchKernelGGL(k_csr_convert, dim3(nb), dim3(256), 0, st, idx_s, idx_t, n_edges, n_nodes, src_o, dst_o, iota,
                     err_flag);
  // stable sort by centre: keys src_o -> src_p, values iota -> order
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, src_o, src_p, iota, order, (size_t)n_edges, 0, bits, st);
  if (e != hipSuccess) return int(e);
  hipLaunchKernelGGL(k_csr_gather, dim3(nb), dim3(256), 0, st, dst_o, order, n_edges, dst_p);
  // stable sort of p by neighbour: keys dst_p -> dst_sorted, values iota -> in_pos
  e = rocprim::radix_sort_pairs(temp, temp_bytes, dst_p, dst_sorted, iota, in_pos, (size_t)n_edges, 0, bits, st);
  if (e != hipSuccess) return int(e);
  const unsigned nbn = lcao_blocks(n_nodes + 1, 256);
  hipLaunchKernelGGL(k_csr_lower_bound, dim3(nbn), dim3(256), 0, st, src_p, n_edges, n_nodes, out_ptr);
  hipLaunchKernelGGL(k_csr_lower_bound, dim3(nbn), dim3(256), 0, st, dst_sorted, n_edges, n_nodes, in_ptr);
  return lcao_launch_status();
}
