// lcao_neighbors.hip — periodic cell-list neighbour search on the device: the GPU counterpart of
// what the reference does offline on the CPU with ase (lcaonet/data/convert.py:124-129: neighbor_list("ijdS"),
// general triclinic cells, per-axis periodicity).  Output convention as the reference's graphs:
//   edge (i -> j, S):  vec = pos[j] - pos[i] + S @ lattice[graph(i)],  |vec| <= cutoff,  (i, j, S) != (i, i, 0).
// Pipeline (all on the caller's stream, no host sync inside the library):
//   1. lcao_nl_bin:     fractional coordinates, wrap into the cell along periodic axes, global bin id per atom
//   2. (caller) stable sort of atoms by bin id + bin start pointers   (lcao_nl_sort: rocPRIM radix sort + lower bound)
//   3. lcao_nl_search:  one thread per atom walks the bins within reach (with image shifts); COUNT pass, then FILL
//                       pass into compact arrays at offsets = exclusive scan of the counts (caller does the scan)
// Per-graph geometry (inverse lattice, bins per axis, reach in bins) is a tiny host-side/torch computation passed in
// as arrays; the per-centre distance sort and max_neighbors truncation (convert.py:135-148) are done by the caller
// on the compact candidate list.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "lcao_common.h"

struct NlGraph {        // one row per graph, 32 floats
  float lat[9];         // lattice rows a, b, c
  float inv[9];         // inverse lattice: frac = pos @ inv
  int nb[3];            // bins per axis
  int reach[3];         // bins to visit on each side (covers the cutoff)
  int pbc[3];
  int bin_base;         // first global bin id of this graph
  int pad[4];
};
static_assert(sizeof(NlGraph) == 32 * 4, "NlGraph is 32 words");

__global__ void k_nl_bin(const float* __restrict__ pos, const int64_t* __restrict__ batch,
                         const NlGraph* __restrict__ G, int64_t N, float* __restrict__ wpos /* wrapped cartesian */,
                         int32_t* __restrict__ wrap /* (N,3) integer wraps removed */, int32_t* __restrict__ bin,
                         int32_t* __restrict__ iota) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const NlGraph& g = G[batch ? batch[i] : 0];
  float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  float f[3];
  int w[3], b[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f[d] = p[0] * g.inv[d] + p[1] * g.inv[3 + d] + p[2] * g.inv[6 + d];
    w[d] = g.pbc[d] ? (int)floorf(f[d]) : 0;
    f[d] -= (float)w[d];
    int bb = (int)floorf(f[d] * (float)g.nb[d]);
    b[d] = min(max(bb, 0), g.nb[d] - 1);   // non-periodic axes: atoms outside the cell land in the border bins
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    wpos[3 * i + d] = p[d] - ((float)w[0] * g.lat[d] + (float)w[1] * g.lat[3 + d] + (float)w[2] * g.lat[6 + d]);
    wrap[3 * i + d] = w[d];
  }
  bin[i] = g.bin_base + (b[0] * g.nb[1] + b[1]) * g.nb[2] + b[2];
  iota[i] = (int32_t)i;
}

__global__ void k_nl_lower_bound(const int32_t* __restrict__ sorted, int64_t n, int64_t n_keys,
                                 int32_t* __restrict__ ptr) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > n_keys) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < (int32_t)i) lo = mid + 1;
    else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

template <bool FILL>
__global__ void k_nl_search(const float* __restrict__ wpos, const int32_t* __restrict__ wrap,
                            const int64_t* __restrict__ batch, const NlGraph* __restrict__ G,
                            const int32_t* __restrict__ atom_of /* sorted by bin */,
                            const int32_t* __restrict__ bin_ptr, int64_t N, float cutoff,
                            int32_t* __restrict__ count, const int64_t* __restrict__ offset,
                            int64_t* __restrict__ e_src, int64_t* __restrict__ e_dst, float* __restrict__ e_shift,
                            float* __restrict__ e_dist) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= N) return;
  const int i = atom_of[t];   // threads of a wave handle atoms of the same / adjacent bins: coherent bin walks
  const NlGraph& g = G[batch ? batch[i] : 0];
  const float px = wpos[3 * i], py = wpos[3 * i + 1], pz = wpos[3 * i + 2];
  float f[3];
  int b[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f[d] = px * g.inv[d] + py * g.inv[3 + d] + pz * g.inv[6 + d];
    b[d] = min(max((int)floorf(f[d] * (float)g.nb[d]), 0), g.nb[d] - 1);
  }
  const float c2 = cutoff * cutoff;
  int n_found = 0;
  int64_t out = FILL ? offset[i] : 0;
  for (int ox = -g.reach[0]; ox <= g.reach[0]; ++ox) {
    int bx = b[0] + ox, sx = 0;
    if (g.pbc[0]) { sx = (bx >= 0) ? bx / g.nb[0] : -((-bx + g.nb[0] - 1) / g.nb[0]); bx -= sx * g.nb[0]; }
    else if (bx < 0 || bx >= g.nb[0]) continue;
    for (int oy = -g.reach[1]; oy <= g.reach[1]; ++oy) {
      int by = b[1] + oy, sy = 0;
      if (g.pbc[1]) { sy = (by >= 0) ? by / g.nb[1] : -((-by + g.nb[1] - 1) / g.nb[1]); by -= sy * g.nb[1]; }
      else if (by < 0 || by >= g.nb[1]) continue;
      for (int oz = -g.reach[2]; oz <= g.reach[2]; ++oz) {
        int bz = b[2] + oz, sz = 0;
        if (g.pbc[2]) { sz = (bz >= 0) ? bz / g.nb[2] : -((-bz + g.nb[2] - 1) / g.nb[2]); bz -= sz * g.nb[2]; }
        else if (bz < 0 || bz >= g.nb[2]) continue;
        const float shx = (float)sx * g.lat[0] + (float)sy * g.lat[3] + (float)sz * g.lat[6];
        const float shy = (float)sx * g.lat[1] + (float)sy * g.lat[4] + (float)sz * g.lat[7];
        const float shz = (float)sx * g.lat[2] + (float)sy * g.lat[5] + (float)sz * g.lat[8];
        const int gb = g.bin_base + (bx * g.nb[1] + by) * g.nb[2] + bz;
        for (int k = bin_ptr[gb]; k < bin_ptr[gb + 1]; ++k) {
          const int j = atom_of[k];
          if (j == i && sx == 0 && sy == 0 && sz == 0) continue;
          const float dx = wpos[3 * j] + shx - px, dy = wpos[3 * j + 1] + shy - py, dz = wpos[3 * j + 2] + shz - pz;
          const float d2 = dx * dx + dy * dy + dz * dz;
          if (d2 <= c2) {
            if (FILL) {
              e_src[out] = i;
              e_dst[out] = j;
              // shift relative to the caller's (unwrapped) positions: S = S_wrapped + wrap_i - wrap_j
              e_shift[3 * out] = (float)(sx + wrap[3 * i] - wrap[3 * j]);
              e_shift[3 * out + 1] = (float)(sy + wrap[3 * i + 1] - wrap[3 * j + 1]);
              e_shift[3 * out + 2] = (float)(sz + wrap[3 * i + 2] - wrap[3 * j + 2]);
              e_dist[out] = sqrtf(d2);
              ++out;
            }
            ++n_found;
          }
        }
      }
    }
  }
  if (!FILL) count[i] = n_found;
}

extern "C" size_t lcao_nl_workspace_bytes(int64_t n_atoms) {
  if (n_atoms <= 0) return 256;
  size_t bytes = 0;
  int32_t* np = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, np, np, np, np, (size_t)n_atoms, 0, 31, (hipStream_t)0);
  size_t n4 = ((size_t)n_atoms * 4 + 255) & ~size_t(255);
  return ((bytes + 255) & ~size_t(255)) + 3 * n4;
}

// steps 1 + 2: wrapped positions, wraps, atoms sorted by bin (atom_of), bin_ptr (n_bins + 1)
extern "C" int lcao_nl_bin_sort(const float* pos, const int64_t* batch, const void* graphs, int64_t n_atoms,
                                int64_t n_bins, float* wpos, int32_t* wrap, int32_t* atom_of, int32_t* bin_ptr,
                                void* workspace, size_t ws_bytes, void* stream) {
  LCAO_CHECK_SIZE(n_atoms);
  LCAO_CHECK_SIZE(n_bins);
  if (n_atoms == 0) return LCAO_OK;
  LCAO_CHECK_PTR(pos); LCAO_CHECK_PTR(graphs); LCAO_CHECK_PTR(wpos); LCAO_CHECK_PTR(wrap); LCAO_CHECK_PTR(atom_of);
  LCAO_CHECK_PTR(bin_ptr); LCAO_CHECK_PTR(workspace);
  if (ws_bytes < lcao_nl_workspace_bytes(n_atoms)) return LCAO_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const size_t n4 = ((size_t)n_atoms * 4 + 255) & ~size_t(255);
  int32_t* bin = (int32_t*)workspace;
  int32_t* iota = (int32_t*)((char*)workspace + n4);
  int32_t* bin_sorted = (int32_t*)((char*)workspace + 2 * n4);
  void* temp = (char*)workspace + 3 * n4;
  size_t temp_bytes = ws_bytes - 3 * n4;
  const unsigned nb = lcao_blocks(n_atoms, 256);
  hipLaunchKernelGGL(k_nl_bin, dim3(nb), dim3(256), 0, st, pos, batch, (const NlGraph*)graphs, n_atoms, wpos, wrap, bin,
                     iota);
  int bits = 1;
  while ((int64_t(1) << bits) < n_bins + 1) ++bits;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, bin, bin_sorted, iota, atom_of, (size_t)n_atoms, 0, bits, st);
  if (e != hipSuccess) return int(e);
  hipLaunchKernelGGL(k_nl_lower_bound, dim3(lcao_blocks(n_bins + 1, 256)), dim3(256), 0, st, bin_sorted, n_atoms, n_bins,
                     bin_ptr);
  return lcao_launch_status();
}

// step 3: fill == 0: count[i] = candidates of atom i within the cutoff;  fill == 1: write them at offset[i]
extern "C" int lcao_nl_search(const float* wpos, const int32_t* wrap, const int64_t* batch, const void* graphs,
                              const int32_t* atom_of, const int32_t* bin_ptr, int64_t n_atoms, float cutoff, int fill,
                              int32_t* count, const int64_t* offset, int64_t* e_src, int64_t* e_dst, float* e_shift,
                              float* e_dist, void* stream) {
  LCAO_CHECK_SIZE(n_atoms);
  if (!(cutoff > 0.f)) return LCAO_ERR_INVALID;
  if (n_atoms == 0) return LCAO_OK;
  LCAO_CHECK_PTR(wpos); LCAO_CHECK_PTR(wrap); LCAO_CHECK_PTR(graphs); LCAO_CHECK_PTR(atom_of); LCAO_CHECK_PTR(bin_ptr);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = lcao_blocks(n_atoms, 128);
  if (!fill) {
    LCAO_CHECK_PTR(count);
    hipLaunchKernelGGL(k_nl_search<false>, dim3(nb), dim3(128), 0, st, wpos, wrap, batch, (const NlGraph*)graphs, atom_of,
                       bin_ptr, n_atoms, cutoff, count, offset, e_src, e_dst, e_shift, e_dist);
  } else {
    LCAO_CHECK_PTR(offset); LCAO_CHECK_PTR(e_src); LCAO_CHECK_PTR(e_dst); LCAO_CHECK_PTR(e_shift); LCAO_CHECK_PTR(e_dist);
    hipLaunchKernelGGL(k_nl_search<true>, dim3(nb), dim3(128), 0, st, wpos, wrap, batch, (const NlGraph*)graphs, atom_of,
                       bin_ptr, n_atoms, cutoff, count, offset, e_src, e_dst, e_shift, e_dist);
  }
  return lcao_launch_status();
}
