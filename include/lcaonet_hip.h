/* lcaonet_hip.h — C ABI of liblcaonet_hip.so: LCAONet's per-edge interaction hot path on MI355X (gfx950).
 *
 * The reference (nmdl-mizo/lcaonet v0.0.3) is pure Python: it has no FFI.  Its boundary for this path is
 * the nn.Module API; each entry point below replaces the ATen / torch_scatter op group of one reference
 * function (cited as file:line relative to the reference root).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers into caller-owned buffers (PyTorch-ROCm tensors); the library never allocates,
 *     frees, or synchronises; work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - float data is fp32, row-major, contiguous unless a leading dimension is given.
 *   - `int64_t` index arrays are the reference's own (edge_index, batch, z); `int32_t` arrays are the
 *     library's internal CSR built by lcao_build_csr_* (E, N < 2^31, checked).
 *   - return 0 on success; <0 = LCAO_ERR_*; >0 = hipError_t from the launch.  lcao_error_string() decodes.
 *   - re-entrant, no global state, one host thread per device.
 *
 * Edge spaces: "orig" = the caller's edge order; "p" = edges stably sorted by centre node idx_s
 * (`order[p]` = orig edge id).  Per-edge work buffers (A, B, tbw, ...) live in p-space.
 */
#ifndef LCAONET_HIP_H
#define LCAONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: lcao_build_csr gained err_flag and the coefficient kernels cst_row (round 2); the 16-bit coefficient kernels take
 *    packed rows (lcao_lowp_pack_rows); the tile-queue workspace is 16 bytes, caller-zeroed once, persistent. */
#define LCAO_ABI_VERSION 2

#define LCAO_OK 0
#define LCAO_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, size over 2^31) */
#define LCAO_ERR_UNSUPPORTED (-2) /* shape outside what the kernel was built for (caller must not fall back silently) */

enum { LCAO_ACT_NONE = 0, LCAO_ACT_SILU = 1, LCAO_ACT_SIGMOID = 2 };
enum { LCAO_CUTOFF_ENVELOPE = 0, LCAO_CUTOFF_POLYNOMIAL = 1, LCAO_CUTOFF_COSINE = 2, LCAO_CUTOFF_NONE = 3 };
enum { LCAO_RBF_HYDROGEN = 0, LCAO_RBF_BESSEL = 1 };
enum { LCAO_PREC_BF16X3 = 0, LCAO_PREC_F16 = 1 };   /* the two precisions of the 16-bit matrix path (BASELINE config 3) */

/* per-orbital row of the radial table handed to lcao_radial_basis: LCAO_ORB_STRIDE floats
 *   [0] zeta scale 2/(n*r0)   [1] normalisation C_nl   [2] l   [3] polynomial degree k=n-l-1
 *   [4..10] integer polynomial coefficients c_0..c_6 (ascending powers)   [11] bessel frequency n*pi (divided by rc in the kernel) */
#define LCAO_ORB_STRIDE 12
#define LCAO_MAX_L 3

int lcao_abi_version(void);
const char* lcao_error_string(int code);

/* a1  BaseMPNN.calc_atomic_distances (lcaonet/model/base.py:15-43)
 * vec = pos[t]-pos[s] + shift @ lattice[batch[s]];  dist = |vec|;  uvec = vec/dist.   (orig order) */
int lcao_edge_geometry(const float* pos, const int64_t* idx_s, const int64_t* idx_t, const float* shift,
                       const float* lattice, const int64_t* batch /*nullable*/, int64_t n_edges,
                       float* dist, float* uvec, void* stream);

/* a4  {Envelope,Polynomial,Cosine}Cutoff.forward (lcaonet/nn/cutoff.py:32-34,41-43,62-67) */
int lcao_cutoff(const float* r, int64_t n, int kind, float rc, float* out, void* stream);

/* a5/a5'  HydrogenRadialBasis.forward / SphericalBesselRadialBasis.forward (lcaonet/nn/rbf.py:129-142,171-182)
 * r (E) -> rb (E, D).  orb_table: device, D rows of LCAO_ORB_STRIDE floats. */
int lcao_radial_basis(const float* r, int64_t n_edges, int n_orb, const float* orb_table, int rbf_kind,
                      int cutoff_kind, float rc, float* rb, void* stream);

/* a1 + a4 + a5 fused, one thread per edge of p-space (order / src_p / dst_p from lcao_build_csr): the outputs of
 * lcao_edge_geometry (dist, uvec, caller's edge order) and lcao_radial_basis (rb (E, D), caller's order) plus uvec_p, the
 * unit vectors in p-space.  Same arithmetic as the two separate calls. */
int lcao_edge_geometry_basis(const float* pos, const int32_t* order, const int32_t* src_p, const int32_t* dst_p,
                             const float* shift, const float* lattice, const int64_t* batch /*nullable*/, int64_t n_edges,
                             int n_orb, const float* orb_table, int rbf_kind, int cutoff_kind, float rc, float* dist,
                             float* uvec, float* uvec_p, float* rb, void* stream);

/* a6  SphericalHarmonicsBasis.forward (lcaonet/nn/shbf.py:75-87): costheta (T) -> shb (T, D), Y_l^0, l = l_of_d[d] */
int lcao_sph_basis(const float* costheta, int64_t n, int n_orb, const int32_t* l_of_d, float* shb, void* stream);

/* a3  LCAONet.calc_3body_angles (lcaonet/model/lcaonet.py:432-433): cos[t] = uvec[e_st[t]] . uvec[e_ks[t]] */
int lcao_triplet_angles(const float* uvec, const int64_t* e_st, const int64_t* e_ks, int64_t n_trip, float* cosv,
                        void* stream);

/* a2 / SURVEY 8f-1  index structures of one forward, built on the device without a host sync (replaces
 * LCAONet.get_triplets, lcaonet/model/lcaonet.py:439-486, and its torch_sparse sort / row-select):
 *   order, src_p, dst_p (E): edges stably sorted by centre node;  out_ptr (N+1);  in_pos (E), in_ptr (N+1): for each
 *   node s the p-indices of the edges k->s ordered by (k, caller's edge id);  src_o, dst_o (E): idx_s / idx_t as int32.
 * workspace: device, >= lcao_build_csr_workspace_bytes(E, N) bytes. */
size_t lcao_build_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes);
int lcao_build_csr(const int64_t* idx_s, const int64_t* idx_t, int64_t n_edges, int64_t n_nodes, int32_t* order,
                   int32_t* src_p, int32_t* dst_p, int32_t* out_ptr, int32_t* in_ptr, int32_t* in_pos, int32_t* src_o,
                   int32_t* dst_o, void* workspace, size_t ws_bytes,
                   int32_t* err_flag /* device int32, nullable: bit LCAO_FLAG_EDGE_INDEX is OR-ed in when an endpoint is
                                        outside [0, n_nodes) (the offending index is clamped) */,
                   void* stream);

/* Sync-free index validation (the reference raises IndexError from nn.Embedding / tensor indexing): out[i] = v[i] clamped to
 * [lo, hi]; any value outside the range ORs `bit` into *err_flag.  The host reads the flag when it next synchronises. */
#define LCAO_FLAG_EDGE_INDEX 1
#define LCAO_FLAG_ATOMIC_NUMBER 2
int lcao_checked_index(const int64_t* v, int64_t n, int64_t lo, int64_t hi, int bit, int64_t* out, int32_t* err_flag,
                       void* stream);

/* SURVEY 8f-1  periodic cell-list neighbour search on the device: what lcaonet/data/convert.py:124-129 obtains from
 * ase.neighbor_list("ijdS") on the CPU.  `graphs`: one 32-word record per graph {lattice[9], inverse[9], bins[3],
 * reach[3], pbc[3], bin_base, pad[4]} (lcaonet_amd/data/neighbors.py builds it).  Edge convention of the reference:
 * vec = pos[dst] - pos[src] + shift @ lattice.  The caller sorts the candidates per centre by distance and truncates
 * to max_neighbors (convert.py:135-148). */
size_t lcao_nl_workspace_bytes(int64_t n_atoms);
int lcao_nl_bin_sort(const float* pos, const int64_t* batch /*nullable*/, const void* graphs, int64_t n_atoms,
                     int64_t n_bins, float* wpos, int32_t* wrap, int32_t* atom_of, int32_t* bin_ptr, void* workspace,
                     size_t ws_bytes, void* stream);
int lcao_nl_search(const float* wpos, const int32_t* wrap, const int64_t* batch, const void* graphs,
                   const int32_t* atom_of, const int32_t* bin_ptr, int64_t n_atoms, float cutoff, int fill,
                   int32_t* count, const int64_t* offset, int64_t* e_src, int64_t* e_dst, float* e_shift, float* e_dist,
                   void* stream);

/* Dense layers on the path (lcaonet/nn/base.py:71-81 used at lcaonet.py:167,170,192,209,214,290,305):
 * Y[m, n] = act( sum_k X[m,k] * W[n,k] + bias[n] ) + residual[m,n];  bias / residual nullable. */
int lcao_dense(const float* X, int64_t ldx, const float* W, const float* bias, const float* residual, int64_t ldr,
               float* Y, int64_t ldy, int64_t M, int K, int N, int act, void* stream);

/* Node-level tables of one LCAOInteraction in one launch (lcaonet.py:167 node_weight, :187 sigmoid(xk), first f_node
 * layer of :209 split as Dense(cat[x_s,x_t]) = x_s Wa^T + x_t Wb^T):  [gate | Ps | Pt] = x . w_cat^T + b_cat with
 * w_cat (3V, H) = [Wk; Wa Ws; Wb Ws] folded by the caller, sigmoid on the first V columns; three (n_rows, V) outputs
 * (each nullable).  H % 16 == 0, V % 32 == 0. */
int lcao_node_tables(const float* x, const float* w_cat, const float* b_cat, int64_t n_rows, int H, int V, float* gate,
                     float* Ps, float* Pt, void* stream);

/* a8  torch_scatter.scatter(reduce="sum") over CONTIGUOUS segments (index sorted): out[i] = sum_{r in [ptr[i],ptr[i+1])} a[r]*b[r]
 * (b nullable => plain segment sum).  Call sites lcaonet.py:189,208-213,293,307.  Deterministic (no atomics). */
int lcao_segment_sum(const float* a, const float* b, const int32_t* ptr, int64_t n_seg, int width, float* out,
                     void* stream);
/* out[s,:] = sum_{r in [ptr[s], ptr[s+1])} a[idx[r],:] — the same sum over a LIST of rows (e.g. the in-edges of a node,
 * idx = in_pos), i.e. lcao_permute_rows + lcao_segment_sum in one pass. */
int lcao_segment_sum_gather(const float* a, const int32_t* idx, const int32_t* ptr, int64_t n_seg, int width, float* out,
                            void* stream);

/* rows gather: out[i,:] = table[idx[i],:]  (x[idx_s], embedding lookups) */
int lcao_gather_rows(const float* table, const int64_t* idx, int64_t n, int width, float* out, void* stream);
/* out[i,:] = src[perm[i],:] with int32 perm */
int lcao_permute_rows(const float* src, const int32_t* perm, int64_t n, int width, float* out, void* stream);

/* lcaonet.py:173-183 + 200-203 restructured (exact reassociation): per edge p (p-space)
 *   A [p,l,:] = sum_{d: l(d)=l} rb[e,d] * ( c[e,d,:] + m[e,d,:]*cv[e,d,:] )        (E, n_l, V)
 *   B [p,:]   = sum_d rb[e,d] * c[e,d,:]                                          (E, V)
 *   Bv[p,:]   = sum_d rb[e,d] * m[e,d,:]*cv[e,d,:]                                (E, V)  (valence only)
 * with e = order[p] (order nullable = identity), cstf = f_coeffs(cst) of shape (E, D, Vt), Vt = V or 2V
 * ([c | cv]); mask given by pointer + element strides (stride_v = 0 for the reference's expanded view). */
int lcao_coeff_contract(const float* cstf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                        const float* mask, int64_t mask_se, int64_t mask_sd, int64_t mask_sv, int64_t n_edges,
                        int n_orb, int V, int n_l, int valence, float* A, float* B, float* Bv, void* stream);

/* lcaonet.py:173-189 fused (a3 + a6 + contraction + F.normalize + sigmoid gate + segment-sum T->E), no T-sized
 * tensor is ever written:  for edge p = (s->t):
 *   tbw[p,:] = sum_{q in In(s), q != p} normalize( sum_l Y_l(u_p . u_q) A[q,l,:] ) * gate[src(q),:]
 * gate = sigmoid(xk) (N,V); uvec in orig order; In(s) = in_pos[in_ptr[s] .. in_ptr[s+1]) (p-indices). */
int lcao_triplet_sum(const float* A, const float* gate, const float* uvec, const int32_t* order,
                     const int32_t* src_p, const int32_t* in_ptr, const int32_t* in_pos, int64_t n_edges, int V,
                     int n_l, float* tbw, void* stream);

/* lcaonet.py:192-204: lw[p,:] = normalize( (1+ft[p,:V]) * B[p,:] [+ (1+ft[p,V:2V]) * Bv[p,:]] ) */
int lcao_edge_combine(const float* ft, const float* B, const float* Bv, int64_t n_edges, int V, float* lw,
                      void* stream);

/* lcaonet.py:209 first f_node layer, split by linearity: h1[p,:] = SiLU( Ps[src_p[p],:] + Pt[dst_p[p],:] )
 * where Ps = x W[:, :V]^T + b, Pt = x W[:, V:]^T are node-level products. */
int lcao_pair_silu(const float* Ps, const float* Pt, const int32_t* src_p, const int32_t* dst_p, int64_t n_edges,
                   int V, float* h1, void* stream);

/* LCAOEmbedding/EmbedCoeffs (lcaonet.py:72, embed.py:235-249) in eval mode, species-table form:
 *   cst[p,d,c] = scale[d,c] * FE[z_t,d,c] * (1 + U[z_s,c] + Vt[z_t,c]) + shift[d,c]
 * z_s = z[src_p[p]], z_t = z[dst_p[p]] (z int64 atomic numbers, tables indexed by z). */
int lcao_embed_coeffs(const float* FE, const float* U, const float* Vt, const float* scale, const float* shift,
                      const int64_t* z, const int32_t* src_p, const int32_t* dst_p, int64_t n_edges, int n_orb, int C,
                      float* cst, void* stream);

/* Literal LCAOInteraction three-body operator with explicit triplet lists (lcaonet.py:173-188), for the
 * operator-level API:  w[t,:] = normalize( sum_d rb[ks,d]*shb[t,d]*(c[ks,d,:] + m*cv[ks,d,:]) ) * sigmoid(xk[k[t],:]) */
int lcao_triplet_literal(const float* cstf, const float* rb, const float* shb, const float* xk, int64_t ldxk,
                         const int64_t* tri_k, const int64_t* e_ks, const float* mask, int64_t mask_se,
                         int64_t mask_sd, int64_t mask_sv, int64_t n_trip, int n_orb, int V, int valence, float* w,
                         void* stream);

/* ---- MFMA fast paths (V, C multiples of 16; return LCAO_ERR_UNSUPPORTED otherwise) ------------------ */

/* lcaonet.py:170 + the contraction above in ONE kernel: f_coeffs = SiLU(SiLU(cst W1^T) W2^T) is never
 * written to HBM.  cst (E, D, C) orig order; W1 (V, C); W2 (Vt, V).  fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * cst_row (nullable): int32 (E) in p-space — edge p reads row cst_row[p] of `cst` instead of row order[p]: the
 * gather of the coefficient rows fused into the kernel (e.g. cst = the (zdim^2, D, C) species-pair table of
 * LCAOEmbedding and cst_row[p] = z_s * zdim + z_t, so the (E, D, C) tensor is never materialised; every (edge, orbital)
 * row still runs through both layers).
 * Outputs (p-space): B (E,V) [, Bv (E,V) with valence];  and EITHER the raw A (E,n_l,V) (Ag == NULL)
 * OR, for lcao_triplet_gemm, the gated copy Ag[p,l,:] = A[p,l,:] * gate[src_p[p],:] plus the per-edge TRIANGULAR
 * FACTOR gram[p, 10]: R upper triangular over l, l' <= 3 (row-major upper triangle, zero beyond n_l) with R^T R = A A^T of
 * the ungated A, from a modified Gram-Schmidt sweep over the vectors — |sum_l y_l A_l|^2 = |R y|^2 is then a sum of
 * squares (the Gram form y^T (A A^T) y of ABI 1 lost a factor S/|v| of accuracy for correlated A_l).  The argument keeps
 * its name; every consumer below reads the factor.  A is scratch when valence. */
int lcao_coeff_mlp_contract(const float* cst, const float* rb, const int32_t* order, const int32_t* cst_row,
                            const int32_t* l_of_d, const float* valence_ed /* (E,D) orig order, nullable */, const float* W1,
                            const float* W2, const float* gate /* (N,V) sigmoid(xk), nullable */,
                            const int32_t* src_p, int64_t n_edges, int n_orb, int C, int V, int n_l, int valence,
                            float* A, float* B, float* Bv, float* Ag, float* gram,
                            void* workspace /* device, >= 16 bytes, ZERO on first use: the kernels' tile queue; every launch
                                               leaves it zero again (no memset is issued); launches sharing one
                                               workspace must be ordered (same stream) */,
                            size_t ws_bytes, void* stream);

/* cst_row for the call above when cst is LCAOEmbedding's species-pair table: row[p] = z[src_p[p]] * zdim + z[dst_p[p]]
 * (embed.py:246-249: cst[e] depends on the two atomic numbers only). */
int lcao_pair_rows(const int64_t* z, const int32_t* src_p, const int32_t* dst_p, int64_t n_edges, int zdim, int32_t* row,
                   void* stream);

/* BASELINE config 3: the same fused kernel with both Dense layers on bf16 MFMA, every fp32 operand split into two
 * bf16 terms and each product run as three MFMAs with fp32 accumulation ("bf16x3": 6-9e-6 relative error on the MLP,
 * against 3.6e-3 for plain bf16).  No valence.  workspace: as above (16 bytes, zero on first use).
 * cst16: the coefficient rows PACKED by lcao_lowp_pack_rows (same row numbering as cst above: row order[p], or
 * cst_row[p] of a table) — cst is the same tensor in every interaction (lcaonet.py:533-534), so its operands are
 * prepared once per forward instead of once per (layer, edge, orbital).  n_orb <= 32. */
int lcao_coeff_mlp_contract_bf16x3(const void* cst16, const float* rb, const int32_t* order, const int32_t* cst_row,
                                   const int32_t* l_of_d, const float* W1, const float* W2, const float* gate, const int32_t* src_p,
                                   int64_t n_edges, int n_orb, int C, int V, int n_l, float* A, float* B, float* Ag,
                                   float* gram, void* workspace, size_t ws_bytes, void* stream);

/* MFMA operands of n_values fp32 numbers (n_values % 8 == 0; rows of C values, C % 32 == 0), per group of 8 values:
 * LCAO_PREC_F16: 16 bytes, the values rounded to fp16;  LCAO_PREC_BF16X3: 32 bytes, 8 bf16 high terms then 8 bf16 low
 * terms (x - hi).  out: device, lcao_lowp_packed_bytes(n_values, precision) bytes. */
size_t lcao_lowp_packed_bytes(int64_t n_values, int precision);
int lcao_lowp_pack_rows(const float* rows, int64_t n_values, int precision, void* out, void* stream);

/* lcaonet.py:173-189 as one small fp32-MFMA GEMM per centre node (no LDS, no T-sized tensor):
 *   tbw[p,:] = sum_{q in In(s), q != p}  sum_l  Y_l(u_p.u_q) / max(sqrt(Y^T G_q Y), 1e-12) * Ag[q,l,:]
 * Ag / gram from lcao_coeff_mlp_contract; uvec_p (E,3) unit vectors in p-space; CSR as lcao_triplet_sum. */
int lcao_triplet_gemm(const float* Ag, const float* gram, const float* uvec_p, const int32_t* in_ptr,
                      const int32_t* in_pos, const int32_t* out_ptr, int64_t n_nodes, int64_t n_edges, int V, int n_l,
                      float* tbw, void* stream);

/* BASELINE config 3 at its STATED tolerance (1e-3): the same three fused kernels with operands rounded once to fp16 and ONE
 * v_mfma_f32_16x16x32_f16 per product (fp32 accumulation): ~3e-4 from fp32, where a single bf16 MFMA (3.6e-3 on the MLP
 * alone) does not meet the tolerance.  Same arguments as the _bf16x3 forms (cst16 packed with LCAO_PREC_F16). */
int lcao_coeff_mlp_contract_f16(const void* cst16, const float* rb, const int32_t* order, const int32_t* cst_row,
                                const int32_t* l_of_d, const float* W1, const float* W2, const float* gate,
                                const int32_t* src_p, int64_t n_edges, int n_orb, int C, int V, int n_l, float* A, float* B,
                                float* Ag, float* gram, void* workspace, size_t ws_bytes, void* stream);
int lcao_edge_lcao_weight_f16(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges, int V,
                              float* bw, void* stream);
int lcao_edge_node_message_f16(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                               const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V, float* msg,
                               void* stream);

/* bf16x3 forms of the two fused two-body kernels below (same arguments, same split as lcao_coeff_mlp_contract_bf16x3) */
int lcao_edge_lcao_weight_bf16x3(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges,
                                 int V, float* bw, void* stream);
int lcao_edge_node_message_bf16x3(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                  const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V,
                                  float* msg, void* stream);

/* lcao_edge_node_message_agg in the two 16-bit precisions: with column_hint reading 1 the per-node-column kernel; otherwise
 * (or with column_hint == NULL) the message rows go through `msg` (E, V scratch) and a per-node sum.  Same device-side
 * choice as the fp32 form; agg (n_nodes, V) is fully written either way. */
int lcao_edge_node_message_agg_bf16x3(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                      const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                                      int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* msg,
                                      float* agg, void* stream);
int lcao_edge_node_message_agg_f16(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                   const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                                   int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* msg,
                                   float* agg, void* stream);

/* precision "f16": lcao_edge_lcao_weight_f16 + lcao_edge_node_message_agg_f16 as ONE call — with a column hint reading 1 or
 * 2 one fused launch that never writes bw (E, V) (the three fp16 weight matrices share the LDS: 96 KB at V = 128); with a
 * hint of 0, or none, the separate kernels through the bw / msg scratch rows ((n_edges, V) each, caller-owned, contents
 * undefined afterwards).  V in {32, 64, 128}. */
int lcao_edge_two_body_agg_f16(const float* tbw, const float* B, const float* W3, const float* Wb, const float* Ps,
                               const float* Pt, const int32_t* src_p, const int32_t* dst_p, const int32_t* out_ptr,
                               const float* Wn2, const float* bn2, int64_t n_nodes, int64_t n_edges, int V,
                               const int32_t* column_hint, float* bw_scratch, float* msg_scratch, float* agg, void* stream);

/* Species-pair form of lcaonet.py:170-183,200 (eval mode, no valence): with
 * Tf[(z_s*zdim + z_t), d, :] = f_coeffs(cst(z_s, z_t))[d, :] precomputed on the (zdim^2, D, V) pair table,
 *   A[p,l,:] = sum_{d in l} rb[e,d] * Tf[pair(p),d,:];  writes Ag = A * gate[src_p], B = sum_l A, gram = the triangular factor R (R^T R = A A^T, as above)
 * exactly as lcao_coeff_mlp_contract does (same consumers), without the 2*D*(C*V+V*V) flop per edge. */
int lcao_pair_contract(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d, const int64_t* z,
                       const int32_t* src_p, const int32_t* dst_p, const float* gate, int64_t n_edges, int n_orb, int V,
                       int n_l, int zdim, float* Ag, float* B, float* gram, void* stream);

/* lcao_pair_contract walking the edges in the order visit[0..n_edges) (a permutation of p-space, e.g. edges sorted by
 * species pair so that neighbouring waves share table blocks in L1).  Same inputs, same outputs at the same rows. */
int lcao_pair_contract_visit(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d,
                             const int64_t* z, const int32_t* src_p, const int32_t* dst_p, const float* gate,
                             const int32_t* visit, int64_t n_edges, int n_orb, int V, int n_l, int zdim, float* Ag,
                             float* B, float* gram, void* stream);

/* lcaonet.py:192-204 + basis_weight (:209) in ONE kernel (no valence):
 *   bw[p,:] = normalize( (1 + tbw[p,:] W3^T) * B[p,:] ) Wb^T          W3, Wb (V, V) */
int lcao_edge_lcao_weight(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges, int V,
                          float* bw, void* stream);

/* lcaonet.py:209 second f_node layer + message product in ONE kernel:
 *   msg[p,:] = bw[p,:] * SiLU( SiLU(Ps[src_p[p],:] + Pt[dst_p[p],:]) Wn2^T + bn2 ) */
int lcao_edge_node_message(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                           const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V,
                           float* msg, void* stream);

/* The same with the node scatter of lcaonet.py:207-213 fused in: agg[s,:] = sum over the out-edges p of s of the message
 * rows (edges sorted by centre; out_ptr from lcao_build_csr; a wave owns whole centre nodes, DPP segmented sums, no
 * atomics, bitwise reproducible).  msg (E, V) is never written.  agg (n_nodes, V) is fully written (zero for nodes without
 * edges).  n_edges = out_ptr[n_nodes].
 * column_hint (device, (1 + n_nodes) x int32 from lcao_node_column_hint, or NULL): when its first word reads 1 or 2 the sum
 * is done by the "one centre node per MFMA column" form instead (16 nodes per wave, step t = their t-th out-edges, per-lane
 * accumulation: a third of the VALU work).  1: groups of 16 consecutive nodes; 2: groups of 16 nodes of similar degree in
 * the order stored in words 1 .. n_nodes (graphs whose neighbouring nodes have uneven degrees); 0: padding would cost more
 * than 10 % either way and the edge-tiled form runs.  The choice is made on the device — no host synchronisation; all
 * forms give the same bits for a node whatever the grouping. */
int lcao_edge_node_message_agg(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                               const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                               int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* agg,
                               void* stream);
/* ... also writing pre2 (E, V), the pre-activation of the second f_node layer, for the reverse sweep (autograd forces, training) */
int lcao_edge_node_message_agg_saving(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                      const int32_t* dst_p, const int32_t* out_ptr, const float* Wn2, const float* bn2,
                                      int64_t n_nodes, int64_t n_edges, int V, const int32_t* column_hint, float* pre2,
                                      float* agg, void* stream);
int lcao_node_column_hint(const int32_t* out_ptr, int64_t n_nodes, int64_t n_edges, int32_t* hint /* 1 + n_nodes */, void* stream);

/* The same two kernels for a forward pass that will be differentiated (autograd forces, training): they also keep
 * ft = tbw W3^T (E,V) and pre2 = SiLU(Ps[s]+Pt[t]) Wn2^T + bn2 (E,V), the two intermediates the reverse sweep needs. */
int lcao_edge_lcao_weight_saving(const float* tbw, const float* B, const float* W3, const float* Wb, int64_t n_edges,
                                 int V, float* bw, float* ft, void* stream);
int lcao_edge_node_message_saving(const float* bw, const float* Ps, const float* Pt, const int32_t* src_p,
                                  const int32_t* dst_p, const float* Wn2, const float* bn2, int64_t n_edges, int V,
                                  float* msg, float* pre2, void* stream);

/* ---- activation-gradient kernels for autograd forces at inference (lcaonet/model/lcaonet.py:310-317).
 * g_X = dE/dX.  Dense layers are differentiated by lcao_dense on transposed weights; see csrc/lcao_backward.hip. ---- */
int lcao_act_bwd(const float* g, const float* pre, int64_t n, int act, float* out, void* stream);
int lcao_pair_pre(const float* Ps, const float* Pt, const int32_t* src_p, const int32_t* dst_p, int64_t n_edges, int V,
                  float* pre1, void* stream);
int lcao_message_bwd(const float* g_agg, const int32_t* src_p, const float* bw, const float* pre2, int64_t n_edges, int V,
                     float* g_bw, float* g_pre2, void* stream);
int lcao_combine_bwd(const float* g_lw, const float* B, const float* ft, int64_t n_edges, int V, float* g_B, float* g_ft,
                     void* stream);
/* Weight gradient of a Dense layer applied to n_rows rows (training, lcaonet.py loss.backward()):
 *   dW (M, K) = G^T X,   G (n_rows, M) = dE/d(output rows),  X (n_rows, K) = input rows;  M, K multiples of 64.
 * fp32 MFMA, deterministic two-stage sum; workspace of lcao_wgrad_workspace_bytes(n_rows, M, K) bytes. */
size_t lcao_wgrad_workspace_bytes(int64_t n_rows, int M, int K);
int lcao_wgrad(const float* G, const float* X, int64_t n_rows, int M, int K, float* dW, void* workspace, size_t ws_bytes,
               void* stream);
/* add_valence form: ft (E,2V) = [ft_a | ft_b], u = B (1+ft_a) + Bv (1+ft_b);  g_ft (E,2V) = [g_u B | g_u Bv] */
int lcao_combine_bwd_valence(const float* g_lw, const float* B, const float* Bv, const float* ft, int64_t n_edges, int V,
                             float* g_B, float* g_Bv, float* g_ft, void* stream);
/* Fused MFMA forms of the two-body reverse sweep (V in {32, 64, 128}; what torch.autograd does for lcaonet.py:192-213).
 * lcao_message_bwd_fused = lcao_message_bwd + Dense^T(f_node.2) + SiLU'(pre1), pre1 = Ps[src_p] + Pt[dst_p] recomputed:
 *   g_bw = g_agg[src_p] SiLU(pre2);  g_pre2 = g_agg[src_p] bw SiLU'(pre2);  g_pre1 = (g_pre2 Wn2) SiLU'(pre1)
 *   Wn2T = Wn2^T (V,V) contiguous;  g_pre2 (E,V) is written only when non-NULL (weight gradients need it);  g_pre1 NULL:
 *   only g_bw (and g_pre2) — the first layer of a force sweep, whose node tables do not depend on the positions.
 * lcao_lcao_weight_bwd_fused = Dense^T(basis_weight) + lcao_combine_bwd + Dense^T(f_three):
 *   g_lw = g_bw Wb;  (g_B, g_ft) as lcao_combine_bwd;  g_tbw = g_ft W3;   WbT = Wb^T, W3T = W3^T;  g_ft optional. */
int lcao_message_bwd_fused(const float* g_agg, const int32_t* src_p, const int32_t* dst_p, const float* bw,
                           const float* pre2, const float* Ps, const float* Pt, const float* Wn2T, int64_t n_edges,
                           int V, float* g_bw, float* g_pre1, float* g_pre2, void* stream);
int lcao_lcao_weight_bwd_fused(const float* g_bw, const float* B, const float* ft, const float* WbT, const float* W3T,
                               int64_t n_edges, int V, float* g_B, float* g_tbw, float* g_ft, void* stream);
/* three-body backward: g_tbw -> g_Ag, g_gram (per in-edge: dE/d(A A^T), upper triangle with doubled off-diagonals — the
 * GRADIENT stays in Gram coordinates, lcao_pair_contract_bwd turns it into dE/dA; `gram` IN is the triangular factor),
 * g_uq / g_up (dE/d unit vector as in-edge / out-edge);
 * tbase[s] = offset of node s in the (sum_s m_in*m_out) float scratch `gc_scratch` */
int lcao_triplet_bwd(const float* g_tbw, const float* Ag, const float* gram, const float* uvec_p, const int32_t* src_p,
                     const int32_t* dst_p, const int32_t* in_ptr, const int32_t* in_pos, const int32_t* out_ptr,
                     const int64_t* tbase, int64_t n_edges, int64_t e_own, int n_own, int64_t n_in_total, int V, int n_l,
                     float* g_Ag, float* g_gram, float* g_uq, float* g_up, float* gc_scratch, void* stream);
/* the same reverse step as one fp32-MFMA kernel per centre node (V in {32,64,128}; every node a centre): writes g_Ag,
 * g_gram, g_uq for every edge and g_up for every edge; no scratch */
int lcao_triplet_bwd_mfma(const float* g_tbw, const float* Ag, const float* gram, const float* uvec_p,
                          const int32_t* in_ptr, const int32_t* in_pos, const int32_t* out_ptr, int64_t n_nodes,
                          int64_t n_edges, int V, int n_l, float* g_Ag, float* g_gram, float* g_uq, float* g_up,
                          void* stream);
int lcao_pair_contract_bwd(const float* Tf, const float* rb, const int32_t* order, const int32_t* l_of_d, const int64_t* z,
                           const int32_t* src_p, const int32_t* dst_p, const float* gate, const float* g_Ag,
                           const float* g_B, const float* g_gram, int64_t n_edges, int64_t e_own, int n_orb, int V, int n_l,
                           int zdim, float* g_rb /* accumulated */, float* g_gate_part /* (E,V) or NULL: not wanted */,
                           float* g_A_out /* (E,n_l,V) or NULL */, void* stream);
/* training: partial (n_chunks, D, V) sums of rb[e,d] * g_A[p,l(d),:] over chunks of the pair-sorted edge list */
int lcao_pair_table_grad(const float* g_A, const float* rb, const int32_t* order, const int32_t* l_of_d,
                         const int32_t* perm, const int32_t* chunk_ptr, int64_t n_chunks, int n_orb, int V, int n_l,
                         float* partial, void* stream);
int lcao_geometry_bwd(const float* dist, const float* uvec, const float* g_rb, const float* g_u, int64_t n_edges, int n_orb,
                      const float* orb_table, int rbf_kind, int cutoff_kind, float rc, float* g_vec, void* stream);

/* ---- exact double backward of autograd-force training (lcaonet.py:310-317, create_graph=True) ---------------------
 * The three-body sum of lcaonet.py:173-189 with the per-triplet scalar weights w (T, n_l) as an INPUT tensor:
 *   spmm  : out[p,:]   = sum_{q in In(src(p)), l} w[t(p,q),l] * A[q,l,:]           (E, V)
 *   spmm_t: out[q,l,:] = sum_{p in Out(dst(q))}   w[t(p,q),l] * G[p,:]             (E, n_l, V)
 *   sddmm : out[t,l]   = G[p,:] . A[q,l,:]                                          (T, n_l)
 * a bilinear family closed under differentiation (each op's two derivatives are the other two), so autograd can
 * differentiate the backward pass again.  Triplet slot t(p,q) = tbase[s] + rank_out(p) * m_in(s) + rank_in(q) with
 * s = src(p) = dst(q); in_rank[q] = rank of q in its node's in-list; n_centres = number of nodes that own out-edges
 * here (all of them, or the owned nodes of a sharded graph; n_edges counts their out-edges for spmm / sddmm and every
 * local edge for spmm_t, whose rows of non-centre targets are zero; all_targets_are_centres != 0: the caller states that
 * there are none — an unsharded graph — and the zero pass over those rows is skipped).  Any V <= 512 (one wave per centre node on fp32
 * MFMA tiles when V % 4 == 0); deterministic. */
int lcao_trip_spmm(const float* w, const float* A, const int32_t* src_p, const int32_t* in_ptr, const int32_t* in_pos,
                   const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges, int64_t n_centres, int V, int n_l,
                   float* out, void* stream);
int lcao_trip_spmm_t(const float* w, const float* G, const int32_t* dst_p, const int32_t* in_rank, const int32_t* in_ptr,
                     const int32_t* in_pos, const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges,
                     int64_t n_centres, int V, int n_l, int all_targets_are_centres, float* out, void* stream);
int lcao_trip_sddmm(const float* G, const float* A, const int32_t* src_p, const int32_t* in_ptr, const int32_t* in_pos,
                    const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges, int64_t n_centres, int V, int n_l,
                    float* out, void* stream);

/* The per-triplet weights themselves, w[t,l] = [p != q] Y_l(c_t) / sqrt(max(Y^T G_q Y, 1e-24)) with c_t = u_p . u_q
 * (lcaonet.py:175-184 in Gram form; c (T), G (E, n_l, n_l), w (T, n_l)), their reverse sweep (gw -> gc, gG) and the
 * reverse sweep of THAT map for cotangents (hc, hG): ggw = J h, (gc2, gG2) = Hessian-vector product — the three
 * launches behind the exact double backward.  One thread per in-list position; deterministic, no atomics. */
int lcao_trip_weight(const float* c, const float* G, const int32_t* dst_p, const int32_t* in_ptr, const int32_t* in_pos,
                     const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges, int64_t n_centres, int n_l, float* w,
                     void* stream);
int lcao_trip_weight_bwd(const float* gw, const float* c, const float* G, const int32_t* dst_p, const int32_t* in_ptr,
                         const int32_t* in_pos, const int32_t* out_ptr, const int64_t* tbase, int64_t n_edges,
                         int64_t n_centres, int n_l, float* gc, float* gG, void* stream);
int lcao_trip_weight_bwd2(const float* hc, const float* hG, const float* gw, const float* c, const float* G,
                          const int32_t* dst_p, const int32_t* in_ptr, const int32_t* in_pos, const int32_t* out_ptr,
                          const int64_t* tbase, int64_t n_edges, int64_t n_centres, int n_l, float* ggw, float* gc2,
                          float* gG2, void* stream);

/* Radial basis with R' and R'' (second-order jets of the closed forms of lcao_radial_basis): level 0: rb (E, D) -> out_ed;
 * level 1: out_e[e] = sum_d g[e,d] R_d'(r_e);  level 2 (cotangent h (E) of level 1's output): out_ed[e,d] = h[e] R_d'(r_e),
 * out_e[e] = h[e] sum_d g[e,d] R_d''(r_e). */
int lcao_radial_levels(int level, const float* r, const float* g, const float* h, int64_t n_edges, int n_orb,
                       const float* orb_table, int rbf_kind, int cutoff_kind, float rc, float* out_ed, float* out_e,
                       void* stream);

/* SiLU and row normalisation (F.normalize, lcaonet.py:196) with closed-form first and second derivatives; `level` selects
 * 0: y(x) -> out0;  1: reverse sweep for cotangent g -> out0;  2: reverse sweep of level 1 for its cotangent h:
 * out0 = d/dg, out1 = d/dx.  Unused inputs / outputs of a level may be NULL. */
int lcao_silu_levels(int level, const float* x, const float* g, const float* h, int64_t n, float* out0, float* out1,
                     void* stream);
int lcao_rownorm_levels(int level, const float* x, const float* g, const float* k, int64_t n_rows, int V, float eps,
                        float* out0, float* out1, void* stream);

/* f(a, b) = a (c0 + b), elementwise — the message product bw * h2 (c0 = 0) and the injection B (1 + ft) (c0 = 1) — with the
 * same derivative levels: 0: out0 = f;  1 (cotangent g): out0 = d/da, out1 = d/db;  2 (cotangents ha, hb of level 1's
 * outputs): out0 = d/dg, out1 = d/da, out2 = d/db.  16-byte aligned operands. */
int lcao_mulp_levels(int level, const float* a, const float* b, const float* g, const float* ha, const float* hb, float c0,
                     int64_t n, float* out0, float* out1, float* out2, void* stream);

/* The orbital contraction in species-pair form TOGETHER with its consumers (lcaonet.py:170-186, 200-203), as one map with
 * derivative levels — A (E, n_l, V) itself is never stored, every level rebuilds it from rb (E, D) and the table rows
 * Tf (zdim^2, D, V).  p-space rows; gate (N, V) is read at src_p; gG / o_gram are full (E, n_l, n_l) matrices.
 *   level 0: o_big0 = Ag = A gate[src] (E, n_l, V), o_row0 = B = sum_l A (E, V), o_gram = G = A A^T
 *   level 1 (cotangents gAg, gB, gG of level 0): o_rb = dE/drb (E, D), o_row1 = sum_l gAg[l] A[l] (E, V; its segment sum
 *            over the out-edges of a node is dE/dgate), optional o_gA = dE/dA (E, n_l, V) for lcao_pair_table_grad
 *   level 2 (cotangents h_rb (E, D), h_gate (N, V) or NULL of level 1's o_rb and of the segment-summed o_row1):
 *            o_big0 = d/dgAg, o_row0 = d/dgB, o_gram = d/dgG, o_rb = d/drb, o_row1 = the per-edge part of d/dgate,
 *            optional o_gA = dE/dA of level 1 and o_dA = d/dA (both or neither): d/dTf = table_grad(o_gA, h_rb) +
 *            table_grad(o_dA, rb).
 * V in {32, 64, 128} (UNSUPPORTED otherwise), n_l <= 4.  (model/double.py::_ASideFn) */
int lcao_aside_levels(int level, const float* Tf, const float* rb, const int32_t* l_of_d, const int64_t* z,
                      const int32_t* src_p, const int32_t* dst_p, const float* gate, const float* gAg, const float* gB,
                      const float* gG, const float* h_rb, const float* h_gate, int64_t n_edges, int n_orb, int V, int n_l,
                      int zdim, float* o_big0, float* o_row0, float* o_gram, float* o_rb, float* o_row1, float* o_gA,
                      float* o_dA, void* stream);

/* Levels 1 and 2 of lcao_aside_levels TOGETHER with the gradient of the species-pair table: the edges are visited in the
 * pair-sorted chunk lists of lcao_pair_table_grad (perm, chunk_ptr), the per-edge outputs are those of lcao_aside_levels
 * (written at the edges' own rows) and `partial` (n_chunks, D, V) receives each chunk's part of
 *   level 1: dE/dTf = sum_p rb[p,d] gA[p,l(d),:]        level 2: d/dTf = sum_p (h_rb[p,d] gA[p,l(d),:] + rb[p,d] dA[p,l(d),:])
 * (lcao_segment_sum over the pair's chunks finishes it) — gA and dA are never written.  V <= 128, n_orb <= 24.
 * order (nullable): rb, h_rb and o_rb rows are in the caller's edge order, order[p] = that row of p-space edge p (as in
 * lcao_pair_contract_bwd).  gram_tri = 1 (level 1 only): gG is the (E, 10) upper triangle in Gram coordinates that
 * lcao_triplet_bwd* writes — the first-order training sweep (model/train.py) calls this instead of
 * lcao_pair_contract_bwd + lcao_pair_table_grad. */
int lcao_aside_pairs(int level, const float* Tf, const float* rb, const int32_t* l_of_d, const int64_t* z,
                     const int32_t* src_p, const int32_t* dst_p, const float* gate, const float* gAg, const float* gB,
                     const float* gG, const float* h_rb, const float* h_gate, const int32_t* perm, const int32_t* chunk_ptr,
                     const int32_t* order, int gram_tri, int64_t n_chunks, int64_t n_edges, int n_orb, int V, int n_l,
                     int zdim, float* o_big0, float* o_row0, float* o_gram, float* o_rb, float* o_row1, float* partial,
                     void* stream);

/* Two fused maps of the edge chain with the same derivative levels (the tensor between the two fused steps is never
 * stored): injnorm y = x / max(|x|, eps) per row with x = a (1 + b)  (lcaonet.py:192-196: a = B, b = ft) and
 * silumul m = a silu(x)  (lcaonet.py:209-211: a = bw, x = f_node's last pre-activation).  Level 0: out0 = f; level 1
 * (cotangent g): out0 = d/da, out1 = d/d(b or x); level 2 (cotangents ha, hb / hx of level 1's outputs): out0 = d/dg,
 * out1 = d/da, out2 = d/d(b or x).  silumul: n % 4 == 0 and 16-byte aligned operands. */
int lcao_injnorm_levels(int level, const float* a, const float* b, const float* g, const float* ha, const float* hb,
                        int64_t n_rows, int V, float eps, float* out0, float* out1, float* out2, void* stream);
int lcao_silumul_levels(int level, const float* a, const float* x, const float* g, const float* ha, const float* hx,
                        int64_t n, float* out0, float* out1, float* out2, void* stream);

/* h1[p,:] = silu(Ps[src_p[p]] + Pt[dst_p[p]])  (first layer of f_node on the pair, lcaonet.py:142-150) with derivative
 * levels; the (E, V) pre-activation is rebuilt from the node tables Ps, Pt (N, V) at every level.  Level 0: out0 = h1;
 * level 1 (cotangent g (E, V)): out0 = dE/dpre (E, V) — the caller's segment sums over src / dst give dE/dPs, dE/dPt;
 * level 2 (cotangents hPs, hPt (N, V) of those sums): out0 = d/dg, out1 = d/dpre.  V % 4 == 0. */
int lcao_pairsilu_levels(int level, const float* Ps, const float* Pt, const int32_t* src_p, const int32_t* dst_p,
                         const float* g, const float* hPs, const float* hPt, int64_t n_edges, int V, float* out0,
                         float* out1, void* stream);

/* The gated copy Ag = A * gate[src] (lcaonet.py:186, broadcast over l) and B = sum_l A (:200-203) with the terms of their two
 * derivative levels:  bcast_fma: out[e,l,:] = X1[e,l,:] y1[e,:] (+ X2[e,l,:] y2[e,:]) (+ z[e,:]);
 *                     lsum_mul : out[e,:] = sum_l X[e,l,:] (Y[e,l,:] or 1).   X (E, L, V), y / z / out (E, V); V % 4 == 0. */
int lcao_bcast_fma(const float* X1, const float* y1, const float* X2 /* nullable, with y2 */, const float* y2,
                   const float* z /* nullable */, int64_t n_edges, int L, int V, float* out, void* stream);
int lcao_lsum_mul(const float* X, const float* Y /* nullable */, int64_t n_edges, int L, int V, float* out, void* stream);

/* The per-edge Gram matrix of lcaonet.py:179-184 (|sum_l Y_l A[q,l,:]|^2 = Y^T G_q Y, G_q = A_q A_q^T) as a second closed
 * family:  rowgram: out[e,l,m] = sum_v X[e,l,v] Y[e,m,v]   (E, n_l, n_l)
 *          rowmix : out[e,l,v] = sum_m S[e,l,m] A[e,m,v]   (E, n_l, V); trans != 0 reads S[e,m,l]
 * (d rowgram = rowmix of the other operand, d rowmix = rowgram / rowmix with S transposed).  Any V <= 512. */
int lcao_rowgram(const float* X, const float* Y, int64_t n_edges, int V, int n_l, float* out, void* stream);
int lcao_rowmix(const float* S, const float* A, int64_t n_edges, int V, int n_l, int trans, float* out, void* stream);

size_t lcao_lds_bytes_coeff_mlp(int C, int V, int valence);

/* Diagnostics only (tools/clock_probe.py): buffer of 2 x uint64 per wave of lcao_coeff_mlp_contract receiving
 * {shader cycles, 100 MHz ticks} of the wave's tile loop; NULL (default) disables the stamps entirely. */
void lcao_debug_set_clock_probe(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* LCAONET_HIP_H */
